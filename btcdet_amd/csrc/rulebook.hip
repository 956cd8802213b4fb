// Rulebook (neighbour-map) construction for gfx950.
//
// Replaces spconv ops.get_indice_pairs behind SubMConv3d / SparseConv3d /
// SparseConvTranspose3d / SparseMaxPool3d (/root/reference/btcdet/models/backbones_3d/
// spconv_backbone.py:12-29).  Instead of spconv's (2,K,N) pair lists claimed with atomicAdd
// (nondeterministic order) the rulebook is stored as two dense neighbour maps
//     nbr_out (n_out,K): input row gathered by output row i at offset k (or -1)
//     nbr_in  (n_in ,K): output row fed by input row j at offset k (or -1)
// which are a pure function of (indices, geometry) and feed the fused output-stationary conv
// kernels of sparse_conv.hip directly (no atomics in the apply stage).
//
//  SubM   : cell -> row hash table (open addressing, int32 keys, L2 resident), one thread per
//           (row, offset) probes its neighbour; nbr_in is the mirrored map (odd kernels).
//  conv / transpose / pool: the set of reachable output cells is a BITMAP over the output grid
//           (atomicOr), ranked by a popcount prefix sum -> output rows come out ascending in
//           (b,z,y,x) with no sort, and rank(bitmap, cell) is also the cell -> output row lookup.
#include "btc_common.h"

namespace {

__device__ __forceinline__ bool out_cell(const BtcGeom& g, int z, int y, int x, int kk, int* oz, int* oy, int* ox) {
  int kx = kk % g.k[2];
  int ky = (kk / g.k[2]) % g.k[1];
  int kz = kk / (g.k[2] * g.k[1]);
  int c[3] = {z, y, x};
  int kv[3] = {kz, ky, kx};
  int o[3];
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    if (g.mode == BTC_MODE_CONV) {
      int t = c[j] + g.p[j] - kv[j] * g.d[j];
      if (t < 0) return false;
      int q = t / g.s[j];
      if (q * g.s[j] != t) return false;
      o[j] = q;
    } else {
      o[j] = c[j] * g.s[j] - g.p[j] + kv[j] * g.d[j];
    }
    if (o[j] < 0 || o[j] >= g.out_shape[j]) return false;
  }
  *oz = o[0]; *oy = o[1]; *ox = o[2];
  return true;
}

// ------------------------------------------------------------------ SubM
__global__ __launch_bounds__(256) void subm_insert(const int4* __restrict__ idx, int n, BtcGeom g, unsigned mask,
                                                   int32_t* __restrict__ keys, int32_t* __restrict__ vals) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  int4 c = idx[i];
  int key = ((c.x * g.in_shape[0] + c.y) * g.in_shape[1] + c.z) * g.in_shape[2] + c.w;
  unsigned slot = btc_hash32((unsigned)key) & mask;
  while (true) {
    int prev = atomicCAS(&keys[slot], BTC_EMPTY_KEY, key);
    if (prev == BTC_EMPTY_KEY || prev == key) break;
    slot = (slot + 1) & mask;
  }
  vals[slot] = i;
}

__global__ __launch_bounds__(256) void subm_lookup(const int4* __restrict__ idx, int n, BtcGeom g, unsigned mask,
                                                   const int32_t* __restrict__ keys, const int32_t* __restrict__ vals,
                                                   int32_t* __restrict__ nbr_out, int32_t* __restrict__ nbr_in) {
  long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (long long)n * g.K) return;
  int i = (int)(t / g.K), kk = (int)(t % g.K);
  int4 c = idx[i];
  int kx = kk % g.k[2];
  int ky = (kk / g.k[2]) % g.k[1];
  int kz = kk / (g.k[2] * g.k[1]);
  int z = c.y + (kz - g.k[0] / 2) * g.d[0];
  int y = c.z + (ky - g.k[1] / 2) * g.d[1];
  int x = c.w + (kx - g.k[2] / 2) * g.d[2];
  int j = -1;
  if (z >= 0 && z < g.in_shape[0] && y >= 0 && y < g.in_shape[1] && x >= 0 && x < g.in_shape[2]) {
    int key = ((c.x * g.in_shape[0] + z) * g.in_shape[1] + y) * g.in_shape[2] + x;
    unsigned slot = btc_hash32((unsigned)key) & mask;
    while (true) {
      int kq = keys[slot];
      if (kq == key) { j = vals[slot]; break; }
      if (kq == BTC_EMPTY_KEY) break;
      slot = (slot + 1) & mask;
    }
  }
  nbr_out[t] = j;
  // input i feeds, at the mirrored offset K-1-kk, exactly the output row that is its neighbour here
  nbr_in[(size_t)i * g.K + (g.K - 1 - kk)] = j;
}

// ------------------------------------------------------------------ conv / transpose / pool
__global__ __launch_bounds__(256) void conv_mark(const int4* __restrict__ idx, int n, BtcGeom g, int ovol,
                                                 unsigned char* __restrict__ bytemap) {
  long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (long long)n * g.K) return;
  int i = (int)(t / g.K), kk = (int)(t % g.K);
  int4 c = idx[i];
  int oz, oy, ox;
  if (!out_cell(g, c.y, c.z, c.w, kk, &oz, &oy, &ox)) return;
  unsigned cell = (unsigned)(c.x * ovol + (oz * g.out_shape[1] + oy) * g.out_shape[2] + ox);
  bytemap[cell] = 1;  // plain store: every writer stores the same value
}

__global__ __launch_bounds__(256) void conv_fill(const int4* __restrict__ idx, int n, BtcGeom g, int ovol,
                                                 const unsigned* __restrict__ bitmap, const int32_t* __restrict__ prefix,
                                                 int32_t* __restrict__ nbr_out, int32_t* __restrict__ nbr_in) {
  long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (long long)n * g.K) return;
  int i = (int)(t / g.K), kk = (int)(t % g.K);
  int4 c = idx[i];
  int oz, oy, ox;
  int row = -1;
  if (out_cell(g, c.y, c.z, c.w, kk, &oz, &oy, &ox)) {
    unsigned cell = (unsigned)(c.x * ovol + (oz * g.out_shape[1] + oy) * g.out_shape[2] + ox);
    unsigned w = cell >> 5, bit = cell & 31;
    row = prefix[w] + __popc(bitmap[w] & ((1u << bit) - 1u));
    nbr_out[(size_t)row * g.K + kk] = i;
  }
  nbr_in[t] = row;
}

__global__ __launch_bounds__(256) void conv_out_indices(const unsigned* __restrict__ bitmap, const int32_t* __restrict__ prefix,
                                                        long long nwords, BtcGeom g, int ovol, int4* __restrict__ out_idx) {
  long long w = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (w >= nwords) return;
  unsigned bits = bitmap[w];
  if (!bits) return;
  int row = prefix[w];
  const int hw = g.out_shape[1] * g.out_shape[2];
  while (bits) {
    int bit = __ffs(bits) - 1;
    bits &= bits - 1;
    unsigned cell = (unsigned)(w * 32 + bit);
    int b = cell / ovol;
    int rem = cell - b * ovol;
    int z = rem / hw;
    int r2 = rem - z * hw;
    out_idx[row++] = make_int4(b, z, r2 / g.out_shape[2], r2 % g.out_shape[2]);
  }
}

// ------------------------------------------------------------------ spconv-layout pair lists
__global__ __launch_bounds__(256) void pairs_count(const int32_t* __restrict__ nbr_out, int n_out, int K,
                                                   int32_t* __restrict__ flags /* K*(n_out+1), offset-major */) {
  long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (long long)K * (n_out + 1)) return;
  int kk = (int)(t / (n_out + 1)), i = (int)(t % (n_out + 1));
  flags[t] = (i < n_out) ? (nbr_out[(size_t)i * K + kk] >= 0) : 0;
}

__global__ __launch_bounds__(256) void pairs_write(const int32_t* __restrict__ nbr_out, const int32_t* __restrict__ excl,
                                                   int n_out, int K, int n_in, int32_t* __restrict__ pairs,
                                                   int32_t* __restrict__ pair_num) {
  long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (long long)K * (n_out + 1)) return;
  int kk = (int)(t / (n_out + 1)), i = (int)(t % (n_out + 1));
  int base = excl[(size_t)kk * (n_out + 1)];
  int pos = excl[t] - base;
  if (i == n_out) {
    pair_num[kk] = pos;
    return;
  }
  int j = nbr_out[(size_t)i * K + kk];
  if (j >= 0) {
    pairs[(size_t)kk * n_in + pos] = j;                       // pairs[0][k][pos] = input row
    pairs[(size_t)K * n_in + (size_t)kk * n_in + pos] = i;    // pairs[1][k][pos] = output row
  }
}

int fill_geom(BtcGeom* g, const int32_t* in_shape, const int32_t* out_shape, const int32_t* k, const int32_t* s,
              const int32_t* p, const int32_t* d, int mode) {
  for (int j = 0; j < 3; ++j) {
    g->in_shape[j] = in_shape[j];
    g->out_shape[j] = out_shape ? out_shape[j] : in_shape[j];
    g->k[j] = k[j];
    g->s[j] = s ? s[j] : 1;
    g->p[j] = p ? p[j] : 0;
    g->d[j] = d ? d[j] : 1;
    if (g->k[j] < 1 || g->s[j] < 1 || g->d[j] < 1 || g->in_shape[j] < 1 || g->out_shape[j] < 1) {
      btc_set_error("rulebook: bad geometry on axis %d", j);
      return BTC_EINVAL;
    }
  }
  g->K = k[0] * k[1] * k[2];
  g->mode = mode;
  return BTC_OK;
}

}  // namespace

extern "C" int btc_out_shape(const int32_t* in_shape, const int32_t* k, const int32_t* s, const int32_t* p, const int32_t* d,
                             const int32_t* outpad, int mode, int32_t* out_shape) {
  for (int j = 0; j < 3; ++j) {
    if (mode == BTC_MODE_SUBM) out_shape[j] = in_shape[j];
    else if (mode == BTC_MODE_CONV) out_shape[j] = (in_shape[j] + 2 * p[j] - d[j] * (k[j] - 1) - 1) / s[j] + 1;
    else if (mode == BTC_MODE_TRANSPOSE) out_shape[j] = (in_shape[j] - 1) * s[j] - 2 * p[j] + k[j] + (outpad ? outpad[j] : 0);
    else { btc_set_error("btc_out_shape: bad mode %d", mode); return BTC_EINVAL; }
  }
  return BTC_OK;
}

extern "C" size_t btc_rulebook_subm_ws_bytes(int n) {
  unsigned cap = btc_pow2_ge((unsigned long long)(n > 0 ? n : 1) * 2);
  return 2 * btc_align((size_t)cap * sizeof(int32_t));
}

extern "C" int btc_rulebook_subm(const int32_t* indices, int n, int batch, const int32_t* h_shape, const int32_t* h_k,
                                 const int32_t* h_d, int32_t* nbr_out, int32_t* nbr_in, void* ws, size_t ws_bytes,
                                 void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  BtcGeom g;
  int rc = fill_geom(&g, h_shape, nullptr, h_k, nullptr, nullptr, h_d, BTC_MODE_SUBM);
  if (rc) return rc;
  BTC_CHECK_ARG((h_k[0] & 1) && (h_k[1] & 1) && (h_k[2] & 1), "btc_rulebook_subm: kernel sizes must be odd");
  BTC_CHECK_ARG(ws_bytes >= btc_rulebook_subm_ws_bytes(n), "btc_rulebook_subm: workspace too small");
  long long vol = (long long)h_shape[0] * h_shape[1] * h_shape[2];
  if (vol * batch >= 0x7fffffffLL) {
    btc_set_error("btc_rulebook_subm: batch*grid volume %lld exceeds 32-bit cell keys", vol * batch);
    return BTC_ERANGE;
  }
  if (n <= 0) return BTC_OK;
  unsigned cap = btc_pow2_ge((unsigned long long)n * 2);
  BtcCarver cv(ws);
  int32_t* keys = cv.take<int32_t>(cap);
  int32_t* vals = cv.take<int32_t>(cap);
  BTC_HIP(hipMemsetAsync(keys, 0xFF, (size_t)cap * sizeof(int32_t), stream));
  subm_insert<<<btc_cdiv(n, 256), 256, 0, stream>>>((const int4*)indices, n, g, cap - 1, keys, vals);
  BTC_LAUNCH_CHECK();
  subm_lookup<<<btc_cdiv((long long)n * g.K, 256), 256, 0, stream>>>((const int4*)indices, n, g, cap - 1, keys, vals,
                                                                     nbr_out, nbr_in);
  BTC_LAUNCH_CHECK();
  return BTC_OK;
}

static long long conv_nwords(int batch, const int32_t* out_shape) {
  long long cells = (long long)batch * out_shape[0] * out_shape[1] * out_shape[2];
  return (cells + 31) / 32;
}

extern "C" size_t btc_rulebook_conv_ws_bytes(int batch, const int32_t* h_out_shape) {
  long long nw = conv_nwords(batch, h_out_shape);
  return btc_align((size_t)nw * sizeof(unsigned)) + btc_align((size_t)(nw + 1) * sizeof(int32_t)) +
         btc_scan_ws_bytes(nw + 1) + btc_bytemap_bytes(nw);
}

extern "C" int btc_rulebook_conv_count(const int32_t* indices, int n, int batch, const int32_t* h_in_shape,
                                       const int32_t* h_out_shape, const int32_t* h_k, const int32_t* h_s,
                                       const int32_t* h_p, const int32_t* h_d, int mode, int32_t* d_n_out, void* ws,
                                       size_t ws_bytes, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  BTC_CHECK_ARG(mode == BTC_MODE_CONV || mode == BTC_MODE_TRANSPOSE, "btc_rulebook_conv_count: bad mode");
  BtcGeom g;
  int rc = fill_geom(&g, h_in_shape, h_out_shape, h_k, h_s, h_p, h_d, mode);
  if (rc) return rc;
  BTC_CHECK_ARG(ws_bytes >= btc_rulebook_conv_ws_bytes(batch, h_out_shape), "btc_rulebook_conv_count: workspace too small");
  long long ovol = (long long)h_out_shape[0] * h_out_shape[1] * h_out_shape[2];
  if (ovol * batch >= 0x7fffffffLL) {
    btc_set_error("btc_rulebook_conv: batch*out volume %lld exceeds 32-bit cell keys", ovol * batch);
    return BTC_ERANGE;
  }
  long long nw = conv_nwords(batch, h_out_shape);
  BtcCarver cv(ws);
  unsigned* bitmap = cv.take<unsigned>(nw);
  int32_t* prefix = cv.take<int32_t>(nw + 1);
  void* scan_ws = cv.take<char>(btc_scan_ws_bytes(nw + 1));
  unsigned char* bytemap = cv.take<unsigned char>(btc_bytemap_bytes(nw));
  BTC_HIP(hipMemsetAsync(bytemap, 0, btc_bytemap_bytes(nw), stream));
  if (n > 0) {
    conv_mark<<<btc_cdiv((long long)n * g.K, 256), 256, 0, stream>>>((const int4*)indices, n, g, (int)ovol, bytemap);
    BTC_LAUNCH_CHECK();
  }
  return btc_bytemap_to_ranked_bitmap(bytemap, nw, bitmap, prefix, d_n_out, scan_ws, stream);
}

extern "C" int btc_rulebook_conv_fill(const int32_t* indices, int n, int batch, const int32_t* h_in_shape,
                                      const int32_t* h_out_shape, const int32_t* h_k, const int32_t* h_s,
                                      const int32_t* h_p, const int32_t* h_d, int mode, int n_out, int32_t* out_indices,
                                      int32_t* nbr_out, int32_t* nbr_in, void* ws, size_t ws_bytes, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  BTC_CHECK_ARG(mode == BTC_MODE_CONV || mode == BTC_MODE_TRANSPOSE, "btc_rulebook_conv_fill: bad mode");
  BtcGeom g;
  int rc = fill_geom(&g, h_in_shape, h_out_shape, h_k, h_s, h_p, h_d, mode);
  if (rc) return rc;
  BTC_CHECK_ARG(ws_bytes >= btc_rulebook_conv_ws_bytes(batch, h_out_shape), "btc_rulebook_conv_fill: workspace too small");
  long long ovol = (long long)h_out_shape[0] * h_out_shape[1] * h_out_shape[2];
  long long nw = conv_nwords(batch, h_out_shape);
  BtcCarver cv(ws);
  unsigned* bitmap = cv.take<unsigned>(nw);
  int32_t* prefix = cv.take<int32_t>(nw + 1);
  if (n_out > 0) BTC_HIP(hipMemsetAsync(nbr_out, 0xFF, (size_t)n_out * g.K * sizeof(int32_t), stream));
  if (n > 0) {
    conv_fill<<<btc_cdiv((long long)n * g.K, 256), 256, 0, stream>>>((const int4*)indices, n, g, (int)ovol, bitmap, prefix,
                                                                     nbr_out, nbr_in);
    BTC_LAUNCH_CHECK();
  }
  if (n_out > 0) {
    conv_out_indices<<<btc_cdiv(nw, 256), 256, 0, stream>>>(bitmap, prefix, nw, g, (int)ovol, (int4*)out_indices);
    BTC_LAUNCH_CHECK();
  }
  return BTC_OK;
}

// pairs/pair_num view; allocates nothing: uses hipMallocAsync-free path -> caller-provided buffers only.
// Scratch for the K*(n_out+1) flags + scan is taken from the tail of `pairs` is NOT possible (sizes
// differ), so this debugging/inspection entry point allocates its scratch with hipMalloc (it is not on
// the training path; the apply kernels consume nbr_out / nbr_in directly).
extern "C" int btc_pairs_from_nbr(const int32_t* nbr_out, int n_out, int K, int n_in, int32_t* pairs, int32_t* pair_num,
                                  void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  BTC_CHECK_ARG(n_out >= 0 && K >= 1 && n_in >= 0, "btc_pairs_from_nbr: bad sizes");
  if (n_in > 0) BTC_HIP(hipMemsetAsync(pairs, 0xFF, (size_t)2 * K * n_in * sizeof(int32_t), stream));
  long long cnt = (long long)K * (n_out + 1);
  int32_t* flags = nullptr;
  void* scan_ws = nullptr;
  BTC_HIP(hipMalloc((void**)&flags, (size_t)cnt * sizeof(int32_t)));
  BTC_HIP(hipMalloc(&scan_ws, btc_scan_ws_bytes(cnt)));
  pairs_count<<<btc_cdiv(cnt, 256), 256, 0, stream>>>(nbr_out, n_out, K, flags);
  int rc = btc_scan_exclusive_i32(flags, flags, cnt, nullptr, scan_ws, stream);
  if (rc == BTC_OK) {
    pairs_write<<<btc_cdiv(cnt, 256), 256, 0, stream>>>(nbr_out, flags, n_out, K, n_in, pairs, pair_num);
  }
  hipError_t e = hipStreamSynchronize(stream);
  (void)hipFree(flags);
  (void)hipFree(scan_ws);
  if (rc) return rc;
  BTC_HIP(e);
  BTC_LAUNCH_CHECK();
  return BTC_OK;
}
