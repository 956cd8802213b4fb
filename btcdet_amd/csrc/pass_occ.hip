// PassOccVox on gfx950: occupancy probabilities -> added occupancy points -> merged detection voxels.
//
// Replaces /root/reference/btcdet/models/occ_pnt/pass_occ_vox.py:10-59 with add_occ_template.py:78-190,248-268:
// per scene `nonzero` + boolean gathers + `topk` (a sort on ROCm, milliseconds for 50 K candidates), ~40 elementwise
// ops for the cell -> xyz -> detection-cell chain, `mask.nonzero()` + gather of the detection voxels' points,
// concatenation, `torch.unique(dim=0)` + `sort` + scatter-pad and a `.cpu()` sync.  Here:
//   pov_select       one workgroup per scene: exact top-k by a 3-pass LDS radix select on the float bits, then an
//                    ordered compaction (cells ascending; ties at the k-th value resolved towards lower cell ids)
//   pov_make_points  selected cell -> centre (+ predicted residual) -> Cartesian -> detection-grid cell key
//   pov_mark / pov_count / pov_scatter / pov_fill   the bitmap-rank re-voxelization of revoxelize.hip over a VIRTUAL
//                    point list (valid slots of the detection voxels, then the added points) -- nothing is
//                    materialised, concatenated or sorted
// One read-back (M'', Pmax, per-scene counts) separates the two entry points.
#include "btc_common.h"

namespace {

struct PovGeom {
  int B, ncell, max_k;         // occupancy grid cells per scene (nz*ny*nx), selection cap
  int onz, ony, onx;           // occupancy grid
  float o_org[3], o_vs[3];     // occupancy origin / voxel size (x = rho, y = azimuth deg, z)
  int D, H, W;                 // detection grid (z,y,x)
  float d_lo[3], d_vs[3];      // detection range origin / voxel size (x,y,z)
  long long dvol;
  int M, P, C;                 // detection voxels (M,P,C)
  float inten;
  int code_dim;                // appended code channels (2: prob, flag)
};

constexpr float kPiF = 3.14159274101257324f;

// ------------------------------------------------------------------ exact top-k per scene
template <int T>
__device__ __forceinline__ int block_excl_scan_i(int v, int* s_w, int* total) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int incl = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    int t = __shfl_up(incl, o, 64);
    if (lane >= o) incl += t;
  }
  if (lane == 63) s_w[wave] = incl;
  __syncthreads();
  if (threadIdx.x == 0) {
    int run = 0;
    for (int w = 0; w < T / 64; ++w) { int t = s_w[w]; s_w[w] = run; run += t; }
    s_w[T / 64] = run;
  }
  __syncthreads();
  int r = incl - v + s_w[wave];
  *total = s_w[T / 64];
  __syncthreads();
  return r;
}

// find, scanning bins from the TOP, the bin where the running count reaches `want`; returns bin, writes the count above it
__device__ __forceinline__ int find_bin(const int* __restrict__ hist /*2048 LDS*/, int want, int* s_w, int* s_res /*[2]*/) {
  // thread t owns bins 2047-2t and 2046-2t (descending order), exclusive scan of their sums
  const int t = threadIdx.x;
  int h0 = hist[2047 - 2 * t], h1 = hist[2046 - 2 * t];
  int tot;
  int ex = block_excl_scan_i<1024>(h0 + h1, s_w, &tot);
  if (ex < want && want <= ex + h0) { s_res[0] = 2047 - 2 * t; s_res[1] = ex; }
  else if (ex + h0 < want && want <= ex + h0 + h1) { s_res[0] = 2046 - 2 * t; s_res[1] = ex + h0; }
  __syncthreads();
  return tot;
}

__global__ __launch_bounds__(1024) void pov_select(const float* __restrict__ probs, const uint8_t* __restrict__ use, PovGeom G, float thresh,
                                                   int32_t* __restrict__ sel, int32_t* __restrict__ counts) {
  __shared__ int s_hist[2048];
  __shared__ int s_w[1024 / 64 + 1];
  __shared__ int s_res[2];
  __shared__ int s_cnt;
  const int b = blockIdx.x, tid = threadIdx.x;
  const float* p = probs + (size_t)b * G.ncell;
  const unsigned* pk = reinterpret_cast<const unsigned*>(p);
  if (use && !use[b]) {
    if (tid == 0) counts[b] = 0;
    return;
  }
  // number of candidates
  int c = 0;
  for (int i = tid; i < G.ncell; i += 1024) c += (p[i] > thresh);
  if (tid == 0) s_cnt = 0;
  __syncthreads();
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) c += __shfl_down(c, o, 64);
  if ((tid & 63) == 0) atomicAdd(&s_cnt, c);
  __syncthreads();
  const int n_cand = s_cnt;
  unsigned T = 0;      // selection threshold key: keep key > T, plus the first `need_eq` keys == T
  int need_eq = 0;
  const bool all = n_cand <= G.max_k;
  if (!all) {
    // 3-pass radix select of the max_k-th largest key among the candidates (positive floats order like their bits)
    unsigned prefix = 0, pmask = 0;
    int want = G.max_k;
    const int shifts[3] = {21, 10, 0};
    const int widths[3] = {11, 11, 10};
    for (int pass = 0; pass < 3; ++pass) {
      for (int i = tid; i < 2048; i += 1024) s_hist[i] = 0;
      __syncthreads();
      for (int i = tid; i < G.ncell; i += 1024) {
        unsigned k = pk[i];
        if (p[i] > thresh && (k & pmask) == prefix) atomicAdd(&s_hist[(k >> shifts[pass]) & ((1u << widths[pass]) - 1u)], 1);
      }
      __syncthreads();
      find_bin(s_hist, want, s_w, s_res);
      const int bin = s_res[0], above = s_res[1];
      want -= above;
      prefix |= (unsigned)bin << shifts[pass];
      pmask |= ((1u << widths[pass]) - 1u) << shifts[pass];
      __syncthreads();
    }
    T = prefix;
    need_eq = want;  // how many keys equal to T are still needed
  }
  // ordered compaction, 4 consecutive cells per thread and iteration
  int base = 0, eq_seen = 0;
  int32_t* out = sel + (size_t)b * G.max_k;
  for (int start = 0; start < G.ncell; start += 1024 * 4) {
    const int i0 = start + tid * 4;
    int f[4], e[4], ne = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int i = i0 + j;
      bool cand = i < G.ncell && p[i] > thresh;
      unsigned k = cand ? pk[i] : 0u;
      f[j] = cand && (all || k > T);
      e[j] = cand && !all && k == T;
      ne += e[j];
    }
    int tot_e, tot_f;
    const int ex_e = block_excl_scan_i<1024>(ne, s_w, &tot_e) + eq_seen;
    // ties: the j-th equal key of this thread has global tie rank ex_e + (#equal before it in the thread)
    int te = 0, take[4], nt = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      take[j] = f[j] || (e[j] && (ex_e + te) < need_eq);
      te += e[j];
      nt += take[j];
    }
    const int ex_t = block_excl_scan_i<1024>(nt, s_w, &tot_f) + base;
    int pos = ex_t;
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (take[j]) out[pos++] = i0 + j;
    base += tot_f;
    eq_seen += tot_e;
  }
  if (tid == 0) counts[b] = base;
}

// ------------------------------------------------------------------ the same selection across many workgroups
// pov_select walks a scene's 2.8 M cells five times with ONE workgroup (205 us for two KITTI scenes).  Same arithmetic, spread
// over (chunks x scenes) workgroups: three histogram passes into a global histogram (each later pass re-derives the prefix
// of the earlier ones from it), per-chunk counts of kept / tied cells, one scan per scene, and the ordered write.
constexpr int POV_CHUNK = 16384;  // cells per workgroup = 4 iterations of the 1024 x 4 ordered-compaction tile

struct PovSel { unsigned prefix, pmask; int want, all; };

// prefix of the max_k-th largest key after `npass` radix passes (the passes' histograms are complete in ghist_b)
__device__ PovSel pov_resolve(const int* __restrict__ ghist_b, int npass, int max_k, int* s_hist, int* s_w, int* s_res) {
  PovSel r{0u, 0u, max_k, 0};
  const int shifts[3] = {21, 10, 0};
  const int widths[3] = {11, 11, 10};
  for (int q = 0; q < npass; ++q) {
    for (int i = threadIdx.x; i < 2048; i += 1024) s_hist[i] = ghist_b[q * 2048 + i];
    __syncthreads();
    const int tot = find_bin(s_hist, r.want, s_w, s_res);
    if (q == 0 && tot <= max_k) { r.all = 1; return r; }  // fewer candidates than the cap: keep them all
    const int bin = s_res[0], above = s_res[1];
    r.want -= above;
    r.prefix |= (unsigned)bin << shifts[q];
    r.pmask |= ((1u << widths[q]) - 1u) << shifts[q];
    __syncthreads();
  }
  return r;
}

template <int PASS>
__global__ __launch_bounds__(1024) void pov_hist(const float* __restrict__ probs, const uint8_t* __restrict__ use, PovGeom G, float thresh,
                                                 int* __restrict__ ghist) {
  __shared__ int s_hist[2048];
  __shared__ int s_w[1024 / 64 + 1];
  __shared__ int s_res[2];
  const int b = blockIdx.y, tid = threadIdx.x;
  if (use && !use[b]) return;
  int* gh = ghist + (size_t)b * 3 * 2048;
  PovSel r = pov_resolve(gh, PASS, G.max_k, s_hist, s_w, s_res);
  if (r.all) return;
  for (int i = tid; i < 2048; i += 1024) s_hist[i] = 0;
  __syncthreads();
  const float* p = probs + (size_t)b * G.ncell;
  const unsigned* pk = reinterpret_cast<const unsigned*>(p);
  const int shifts[3] = {21, 10, 0};
  const int widths[3] = {11, 11, 10};
  const int lo = blockIdx.x * POV_CHUNK, hi = min(lo + POV_CHUNK, G.ncell);
  for (int i = lo + tid; i < hi; i += 1024) {
    const unsigned k = pk[i];
    if (p[i] > thresh && (k & r.pmask) == r.prefix) atomicAdd(&s_hist[(k >> shifts[PASS]) & ((1u << widths[PASS]) - 1u)], 1);
  }
  __syncthreads();
  for (int i = tid; i < 2048; i += 1024)
    if (s_hist[i]) atomicAdd(&gh[PASS * 2048 + i], s_hist[i]);
}

// per chunk: cells kept outright (key > T, or every candidate) and cells tied at T; chunk 0 publishes (T, need_eq, all)
__global__ __launch_bounds__(1024) void pov_chunk_count(const float* __restrict__ probs, const uint8_t* __restrict__ use, PovGeom G, float thresh,
                                                        const int* __restrict__ ghist, int n_chunk, int32_t* __restrict__ cc,
                                                        int32_t* __restrict__ state) {
  __shared__ int s_hist[2048];
  __shared__ int s_w[1024 / 64 + 1];
  __shared__ int s_res[2];
  __shared__ int s_nf, s_ne;
  const int b = blockIdx.y, tid = threadIdx.x;
  if (use && !use[b]) return;
  PovSel r = pov_resolve(ghist + (size_t)b * 3 * 2048, 3, G.max_k, s_hist, s_w, s_res);
  const unsigned T = r.all ? 0u : r.prefix;
  if (tid == 0) { s_nf = 0; s_ne = 0; }
  __syncthreads();
  const float* p = probs + (size_t)b * G.ncell;
  const unsigned* pk = reinterpret_cast<const unsigned*>(p);
  const int lo = blockIdx.x * POV_CHUNK, hi = min(lo + POV_CHUNK, G.ncell);
  int nf = 0, ne = 0;
  for (int i = lo + tid; i < hi; i += 1024) {
    const bool cand = p[i] > thresh;
    const unsigned k = pk[i];
    nf += cand && (r.all || k > T);
    ne += cand && !r.all && k == T;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { nf += __shfl_down(nf, o, 64); ne += __shfl_down(ne, o, 64); }
  if ((tid & 63) == 0) { atomicAdd(&s_nf, nf); atomicAdd(&s_ne, ne); }
  __syncthreads();
  if (tid == 0) {
    cc[((size_t)b * n_chunk + blockIdx.x) * 2 + 0] = s_nf;
    cc[((size_t)b * n_chunk + blockIdx.x) * 2 + 1] = s_ne;
    if (blockIdx.x == 0) { state[b * 4 + 0] = (int)T; state[b * 4 + 1] = r.all ? 0 : r.want; state[b * 4 + 2] = r.all; }
  }
}

// one workgroup per scene: chunk -> (first output slot, ties seen before the chunk); counts[b] = cells selected
__global__ __launch_bounds__(1024) void pov_chunk_scan(const uint8_t* __restrict__ use, int n_chunk, const int32_t* __restrict__ cc,
                                                       const int32_t* __restrict__ state, int32_t* __restrict__ co,
                                                       int32_t* __restrict__ counts) {
  __shared__ int s_w[1024 / 64 + 1];
  const int b = blockIdx.x, tid = threadIdx.x;
  if (use && !use[b]) {
    if (tid == 0) counts[b] = 0;
    return;
  }
  const int need_eq = state[b * 4 + 1];
  int base_t = 0, base_e = 0;
  for (int c0 = 0; c0 < n_chunk; c0 += 1024) {
    const int c = c0 + tid;
    const int nf = c < n_chunk ? cc[((size_t)b * n_chunk + c) * 2 + 0] : 0, ne = c < n_chunk ? cc[((size_t)b * n_chunk + c) * 2 + 1] : 0;
    int tot_e, tot_t;
    const int ex_e = block_excl_scan_i<1024>(ne, s_w, &tot_e) + base_e;
    int take_e = need_eq - ex_e;
    take_e = take_e < 0 ? 0 : (take_e > ne ? ne : take_e);
    const int ex_t = block_excl_scan_i<1024>(nf + take_e, s_w, &tot_t) + base_t;
    if (c < n_chunk) {
      co[((size_t)b * n_chunk + c) * 2 + 0] = ex_t;
      co[((size_t)b * n_chunk + c) * 2 + 1] = ex_e;
    }
    base_t += tot_t;
    base_e += tot_e;
  }
  if (tid == 0) counts[b] = base_t;
}

// ordered compaction of one chunk (the loop body of pov_select, started at the chunk's offsets)
__global__ __launch_bounds__(1024) void pov_chunk_write(const float* __restrict__ probs, const uint8_t* __restrict__ use, PovGeom G, float thresh,
                                                        int n_chunk, const int32_t* __restrict__ co, const int32_t* __restrict__ state,
                                                        int32_t* __restrict__ sel) {
  __shared__ int s_w[1024 / 64 + 1];
  const int b = blockIdx.y, tid = threadIdx.x;
  if (use && !use[b]) return;
  const unsigned T = (unsigned)state[b * 4 + 0];
  const int need_eq = state[b * 4 + 1];
  const bool all = state[b * 4 + 2] != 0;
  const float* p = probs + (size_t)b * G.ncell;
  const unsigned* pk = reinterpret_cast<const unsigned*>(p);
  int base = co[((size_t)b * n_chunk + blockIdx.x) * 2 + 0], eq_seen = co[((size_t)b * n_chunk + blockIdx.x) * 2 + 1];
  int32_t* out = sel + (size_t)b * G.max_k;
  const int lo = blockIdx.x * POV_CHUNK, hi = min(lo + POV_CHUNK, G.ncell);
  for (int start = lo; start < hi; start += 1024 * 4) {
    const int i0 = start + tid * 4;
    int f[4], e[4], ne = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int i = i0 + j;
      bool cand = i < hi && p[i] > thresh;
      unsigned k = cand ? pk[i] : 0u;
      f[j] = cand && (all || k > T);
      e[j] = cand && !all && k == T;
      ne += e[j];
    }
    int tot_e, tot_f;
    const int ex_e = block_excl_scan_i<1024>(ne, s_w, &tot_e) + eq_seen;
    int te = 0, take[4], nt = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      take[j] = f[j] || (e[j] && (ex_e + te) < need_eq);
      te += e[j];
      nt += take[j];
    }
    const int ex_t = block_excl_scan_i<1024>(nt, s_w, &tot_f) + base;
    int pos = ex_t;
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (take[j]) out[pos++] = i0 + j;
    base += tot_f;
    eq_seen += tot_e;
  }
}

// selected occupancy cell -> added point [x,y,z,prob] and its detection-grid cell key (add_occ_template.py:131-146,78-88)
__global__ __launch_bounds__(256) void pov_make_points(const int32_t* __restrict__ sel, const int32_t* __restrict__ counts,
                                                       const float* __restrict__ probs, const float* __restrict__ res /* B,3,ncell or null */,
                                                       const float* __restrict__ rot_z, PovGeom G, float* __restrict__ occ_xyzp,
                                                       unsigned* __restrict__ occ_key) {
  int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= G.B * G.max_k) return;
  int b = t / G.max_k, q = t % G.max_k;
  if (q >= counts[b]) { occ_key[t] = 0xFFFFFFFFu; return; }
  int cell = sel[t];
  int x = cell % G.onx, y = (cell / G.onx) % G.ony, z = cell / (G.onx * G.ony);
  float cx = __fadd_rn(G.o_org[0], __fmul_rn(__fadd_rn((float)x, 0.5f), G.o_vs[0]));
  float cy = __fadd_rn(G.o_org[1], __fmul_rn(__fadd_rn((float)y, 0.5f), G.o_vs[1]));
  float cz = __fadd_rn(G.o_org[2], __fmul_rn(__fadd_rn((float)z, 0.5f), G.o_vs[2]));
  if (rot_z) cy = __fsub_rn(cy, rot_z[b]);
  float a = __fdiv_rn(__fmul_rn(cy, kPiF), 180.0f);
  float px = __fmul_rn(cx, cosf(a)), py = __fmul_rn(-cx, sinf(a)), pz = cz;
  if (res) {
    const float* r = res + (size_t)b * 3 * G.ncell + cell;
    px = __fadd_rn(px, r[0]);
    py = __fadd_rn(py, r[G.ncell]);
    pz = __fadd_rn(pz, r[2 * (size_t)G.ncell]);
  }
  float* o = occ_xyzp + (size_t)t * 4;
  o[0] = px; o[1] = py; o[2] = pz; o[3] = probs[(size_t)b * G.ncell + cell];
  const float pp[3] = {px, py, pz};
  const int n3[3] = {G.W, G.H, G.D};
  int c3[3];
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    float f = floorf(__fdiv_rn(__fsub_rn(pp[j], G.d_lo[j]), G.d_vs[j]));
    f = fminf(fmaxf(f, 0.f), (float)(n3[j] - 1));
    c3[j] = (int)f;
  }
  occ_key[t] = (unsigned)(b * G.dvol + ((long long)c3[2] * G.H + c3[1]) * G.W + c3[0]);
}

// virtual point i: [0, M*P) = slot (v, s) of the detection voxels; [M*P, M*P + B*max_k) = added point
__device__ __forceinline__ bool vkey(int i, const PovGeom& G, const int32_t* __restrict__ dcoords, const int32_t* __restrict__ dnum,
                                     const unsigned* __restrict__ occ_key, unsigned* key) {
  const int mp = G.M * G.P;
  if (i < mp) {
    int v = i / G.P, s = i % G.P;
    if (s >= dnum[v]) return false;
    int4 c = reinterpret_cast<const int4*>(dcoords)[v];
    if (c.x < 0 || c.x >= G.B || c.y < 0 || c.y >= G.D || c.z < 0 || c.z >= G.H || c.w < 0 || c.w >= G.W) return false;
    *key = (unsigned)(c.x * G.dvol + ((long long)c.y * G.H + c.z) * G.W + c.w);
    return true;
  }
  unsigned k = occ_key[i - mp];
  if (k == 0xFFFFFFFFu) return false;
  *key = k;
  return true;
}

__global__ __launch_bounds__(256) void pov_mark(PovGeom G, int nv, const int32_t* __restrict__ dcoords, const int32_t* __restrict__ dnum,
                                                const unsigned* __restrict__ occ_key, unsigned* __restrict__ bitmap) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nv) return;
  unsigned k;
  if (vkey(i, G, dcoords, dnum, occ_key, &k)) atomicOr(&bitmap[k >> 5], 1u << (k & 31));
}

__global__ __launch_bounds__(256) void pov_popc(const unsigned* __restrict__ bitmap, long long nwords, int32_t* __restrict__ counts) {
  long long w = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (w > nwords) return;
  counts[w] = (w < nwords) ? __popc(bitmap[w]) : 0;
}

__global__ __launch_bounds__(256) void pov_count(PovGeom G, int nv, const int32_t* __restrict__ dcoords, const int32_t* __restrict__ dnum,
                                                 const unsigned* __restrict__ occ_key, const unsigned* __restrict__ bitmap,
                                                 const int32_t* __restrict__ prefix, int32_t* __restrict__ point_row, int32_t* __restrict__ cnt) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nv) return;
  unsigned k;
  int row = -1;
  if (vkey(i, G, dcoords, dnum, occ_key, &k)) {
    unsigned w = k >> 5, bit = k & 31;
    row = prefix[w] + __popc(bitmap[w] & ((1u << bit) - 1u));
    atomicAdd(&cnt[row], 1);
  }
  point_row[i] = row;
}

__global__ __launch_bounds__(256) void pov_info(const int32_t* __restrict__ cnt, const int32_t* __restrict__ d_m, const int32_t* __restrict__ counts,
                                                int B, int32_t* __restrict__ info /* [m, pmax, counts...] */) {
  int r = blockIdx.x * blockDim.x + threadIdx.x;
  int v = (r < *d_m) ? cnt[r] : 0;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = max(v, __shfl_down(v, o, 64));
  if ((threadIdx.x & 63) == 0 && v > 0) atomicMax(&info[1], v);
  if (r == 0) {
    info[0] = *d_m;
    for (int b = 0; b < B; ++b) info[2 + b] = counts[b];
  }
}

__global__ __launch_bounds__(256) void pov_scatter(const int32_t* __restrict__ point_row, int nv, const int32_t* __restrict__ offs,
                                                   int32_t* __restrict__ cursor, int32_t* __restrict__ perm) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nv) return;
  int row = point_row[i];
  if (row < 0) return;
  perm[offs[row] + atomicAdd(&cursor[row], 1)] = i;
}

__global__ __launch_bounds__(256) void pov_fill(PovGeom G, int m, int pmax, const float* __restrict__ dvoxels, const float* __restrict__ occ_xyzp,
                                                const int32_t* __restrict__ offs, const int32_t* __restrict__ perm,
                                                const unsigned* __restrict__ bitmap_unused, float* __restrict__ voxels,
                                                int64_t* __restrict__ vcoords, int64_t* __restrict__ vnum, int32_t* __restrict__ vnum32) {
  long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (long long)m * pmax) return;
  const int row = (int)(t / pmax), slot = (int)(t % pmax);
  const int beg = offs[row], cnt = offs[row + 1] - beg;
  const int CO = G.C + G.code_dim;
  float* o = voxels + (size_t)t * CO;
  if (slot >= cnt) {
    for (int c = 0; c < CO; ++c) o[c] = 0.f;
    return;
  }
  // the slot-th member of the cell in virtual-point order (= input order of the reference's concatenation)
  int mine = -1;
  for (int a = 0; a < cnt; ++a) {
    int ia = perm[beg + a], rank = 0;
    for (int b2 = 0; b2 < cnt; ++b2) rank += (perm[beg + b2] < ia);
    if (rank == slot) { mine = ia; break; }
  }
  const int mp = G.M * G.P;
  if (mine < mp) {
    const float* src = dvoxels + (size_t)mine * G.C;
    for (int c = 0; c < G.C; ++c) o[c] = src[c];
    for (int c = G.C; c < CO; ++c) o[c] = 0.f;  // code channels of real points
  } else {
    const float* src = occ_xyzp + (size_t)(mine - mp) * 4;
    o[0] = src[0]; o[1] = src[1]; o[2] = src[2];
    for (int c = 3; c < G.C; ++c) o[c] = (c == 3) ? G.inten : 0.f;
    o[G.C] = src[3];
    if (G.code_dim > 1) o[G.C + 1] = 1.f;
  }
  if (slot == 0) {
    vnum[row] = cnt;
    if (vnum32) vnum32[row] = cnt;   // (the int32 twin the VFE / backbone read: no conversion launch)
  }
}

// vcoords from the bitmap (cells ascending) -- one thread per set word
__global__ __launch_bounds__(256) void pov_coords(const unsigned* __restrict__ bitmap, const int32_t* __restrict__ prefix, long long nwords,
                                                  PovGeom G, int64_t* __restrict__ vcoords, int32_t* __restrict__ vcoords32) {
  long long w = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (w >= nwords) return;
  unsigned bits = bitmap[w];
  if (!bits) return;
  int row = prefix[w];
  const long long hw = (long long)G.H * G.W;
  while (bits) {
    int bit = __ffs(bits) - 1;
    bits &= bits - 1;
    long long cell = w * 32 + bit;
    long long b = cell / G.dvol, rem = cell % G.dvol;
    int64_t* c = vcoords + (size_t)row * 4;
    c[0] = b; c[1] = rem / hw; c[2] = (rem % hw) / G.W; c[3] = rem % G.W;
    if (vcoords32) *(int4*)(vcoords32 + (size_t)row * 4) = make_int4((int)c[0], (int)c[1], (int)c[2], (int)c[3]);
    ++row;
  }
}

__global__ __launch_bounds__(256) void pov_compact_occ(const float* __restrict__ occ_xyzp, const int32_t* __restrict__ counts, PovGeom G,
                                                       float* __restrict__ occ_pnts, int64_t* __restrict__ occ_b) {
  int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= G.B * G.max_k) return;
  int b = t / G.max_k, q = t % G.max_k;
  if (q >= counts[b]) return;
  int base = 0;
  for (int s = 0; s < b; ++s) base += counts[s];
  const float* src = occ_xyzp + (size_t)t * 4;
  float* d = occ_pnts + (size_t)(base + q) * 4;
  d[0] = src[0]; d[1] = src[1]; d[2] = src[2]; d[3] = src[3];
  occ_b[base + q] = b;
}

struct PovWs {
  int32_t *sel, *counts, *info, *point_row, *cnt, *offs, *cursor, *perm, *prefix, *d_m, *ghist, *cc, *co, *state;
  int n_chunk;
  float* occ_xyzp;
  unsigned *occ_key, *bitmap;
  void* scan_ws;
  long long nw;
  int nv;
};

PovWs pov_carve(void* ws, const PovGeom& G) {
  PovWs w;
  w.nw = ((long long)G.B * G.dvol + 31) / 32;
  w.nv = G.M * G.P + G.B * G.max_k;
  BtcCarver cv(ws);
  w.sel = cv.take<int32_t>((size_t)G.B * G.max_k);
  w.counts = cv.take<int32_t>(G.B);
  w.d_m = cv.take<int32_t>(1);
  w.occ_xyzp = cv.take<float>((size_t)G.B * G.max_k * 4);
  w.occ_key = cv.take<unsigned>((size_t)G.B * G.max_k);
  w.n_chunk = btc_cdiv(G.ncell, POV_CHUNK);
  w.cc = cv.take<int32_t>((size_t)G.B * w.n_chunk * 2);
  w.co = cv.take<int32_t>((size_t)G.B * w.n_chunk * 2);
  w.state = cv.take<int32_t>((size_t)G.B * 4);
  // the arrays that start zeroed are adjacent: one memset covers [info, prefix)
  w.info = cv.take<int32_t>(2 + G.B);
  w.ghist = cv.take<int32_t>((size_t)G.B * 3 * 2048);
  w.cnt = cv.take<int32_t>(w.nv + 1);
  w.cursor = cv.take<int32_t>(w.nv + 1);
  w.bitmap = cv.take<unsigned>(w.nw);
  w.prefix = cv.take<int32_t>(w.nw + 1);
  w.point_row = cv.take<int32_t>(w.nv + 1);
  w.offs = cv.take<int32_t>(w.nv + 1);
  w.perm = cv.take<int32_t>(w.nv + 1);
  long long big = w.nw + 1 > w.nv + 1 ? w.nw + 1 : w.nv + 1;
  w.scan_ws = cv.take<char>(btc_scan_ws_bytes(big));
  return w;
}

int pov_geom(PovGeom* G, const BtcPovConfig* c, int M, int P, int C) {
  G->B = c->batch; G->max_k = c->max_k;
  G->onx = c->occ_grid[0]; G->ony = c->occ_grid[1]; G->onz = c->occ_grid[2];
  G->ncell = G->onx * G->ony * G->onz;
  G->W = c->det_grid[0]; G->H = c->det_grid[1]; G->D = c->det_grid[2];
  G->dvol = (long long)G->W * G->H * G->D;
  for (int j = 0; j < 3; ++j) {
    G->o_org[j] = c->occ_origin[j]; G->o_vs[j] = c->occ_voxel[j];
    G->d_lo[j] = c->det_origin[j]; G->d_vs[j] = c->det_voxel[j];
  }
  G->M = M; G->P = P; G->C = C;
  G->inten = c->inten; G->code_dim = c->code_dim;
  if (G->dvol * G->B >= 0x7fffffffLL) {
    btc_set_error("btc_pass_occ_vox: batch*grid volume exceeds 32-bit cell keys");
    return BTC_ERANGE;
  }
  return BTC_OK;
}

}  // namespace

extern "C" size_t btc_pass_occ_vox_ws_bytes(const BtcPovConfig* c, int M, int P) {
  PovGeom G;
  G.B = c->batch; G.max_k = c->max_k; G.M = M; G.P = P;
  G.dvol = (long long)c->det_grid[0] * c->det_grid[1] * c->det_grid[2];
  long long nw = ((long long)G.B * G.dvol + 31) / 32;
  long long nv = (long long)M * P + (long long)G.B * G.max_k;
  long long big = nw + 1 > nv + 1 ? nw + 1 : nv + 1;
  size_t s = 0;
  s += btc_align((size_t)G.B * G.max_k * 4) * 2 + btc_align((size_t)G.B * 4) + btc_align((size_t)(2 + G.B) * 4) + btc_align(4);
  s += btc_align((size_t)G.B * G.max_k * 16);
  s += btc_align((size_t)nw * 4) + btc_align((size_t)(nw + 1) * 4) + 5 * btc_align((size_t)(nv + 1) * 4);
  s += btc_scan_ws_bytes(big);
  const size_t n_chunk = (size_t)btc_cdiv(c->occ_grid[0] * c->occ_grid[1] * c->occ_grid[2], POV_CHUNK);
  s += 2 * btc_align((size_t)G.B * n_chunk * 2 * 4) + btc_align((size_t)G.B * 4 * 4) + btc_align((size_t)G.B * 3 * 2048 * 4);
  return s;
}

extern "C" int btc_pass_occ_vox_count(const BtcPovConfig* cfg, const float* probs, const float* residuals, const uint8_t* use_occ,
                                      const float* rot_z, const int32_t* det_coords, const int32_t* det_num, int M, int P, int C,
                                      int32_t* d_info, void* ws, size_t ws_bytes, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  BTC_CHECK_ARG(cfg && cfg->batch >= 1 && cfg->max_k >= 1, "btc_pass_occ_vox_count: bad config");
  BTC_CHECK_ARG(ws_bytes >= btc_pass_occ_vox_ws_bytes(cfg, M, P), "btc_pass_occ_vox_count: workspace too small");
  PovGeom G;
  int rc = pov_geom(&G, cfg, M, P, C);
  if (rc) return rc;
  PovWs w = pov_carve(ws, G);
  BTC_HIP(hipMemsetAsync(w.info, 0, (size_t)((char*)w.prefix - (char*)w.info), stream));
  if (btc_tune_get(BTC_TUNE_POV_SELECT) == 1) {  // the single-workgroup selection (kept as the in-tree cross-check of the one below)
    pov_select<<<G.B, 1024, 0, stream>>>(probs, use_occ, G, cfg->occ_thresh, w.sel, w.counts);
    BTC_LAUNCH_CHECK();
  } else {
    const dim3 grid(w.n_chunk, G.B);
    pov_hist<0><<<grid, 1024, 0, stream>>>(probs, use_occ, G, cfg->occ_thresh, w.ghist);
    pov_hist<1><<<grid, 1024, 0, stream>>>(probs, use_occ, G, cfg->occ_thresh, w.ghist);
    pov_hist<2><<<grid, 1024, 0, stream>>>(probs, use_occ, G, cfg->occ_thresh, w.ghist);
    pov_chunk_count<<<grid, 1024, 0, stream>>>(probs, use_occ, G, cfg->occ_thresh, w.ghist, w.n_chunk, w.cc, w.state);
    pov_chunk_scan<<<G.B, 1024, 0, stream>>>(use_occ, w.n_chunk, w.cc, w.state, w.co, w.counts);
    pov_chunk_write<<<grid, 1024, 0, stream>>>(probs, use_occ, G, cfg->occ_thresh, w.n_chunk, w.co, w.state, w.sel);
    BTC_LAUNCH_CHECK();
  }
  pov_make_points<<<btc_cdiv(G.B * G.max_k, 256), 256, 0, stream>>>(w.sel, w.counts, probs, residuals, rot_z, G, w.occ_xyzp, w.occ_key);
  BTC_LAUNCH_CHECK();
  pov_mark<<<btc_cdiv(w.nv, 256), 256, 0, stream>>>(G, w.nv, det_coords, det_num, w.occ_key, w.bitmap);
  BTC_LAUNCH_CHECK();
  pov_popc<<<btc_cdiv(w.nw + 1, 256), 256, 0, stream>>>(w.bitmap, w.nw, w.prefix);
  BTC_LAUNCH_CHECK();
  rc = btc_scan_exclusive_i32(w.prefix, w.prefix, w.nw + 1, w.d_m, w.scan_ws, stream);
  if (rc) return rc;
  pov_count<<<btc_cdiv(w.nv, 256), 256, 0, stream>>>(G, w.nv, det_coords, det_num, w.occ_key, w.bitmap, w.prefix, w.point_row, w.cnt);
  BTC_LAUNCH_CHECK();
  pov_info<<<btc_cdiv(w.nv, 256), 256, 0, stream>>>(w.cnt, w.d_m, w.counts, G.B, w.info);
  BTC_LAUNCH_CHECK();
  rc = btc_scan_exclusive_i32(w.cnt, w.offs, w.nv + 1, nullptr, w.scan_ws, stream);
  if (rc) return rc;
  pov_scatter<<<btc_cdiv(w.nv, 256), 256, 0, stream>>>(w.point_row, w.nv, w.offs, w.cursor, w.perm);
  BTC_LAUNCH_CHECK();
  BTC_HIP(hipMemcpyAsync(d_info, w.info, (size_t)(2 + G.B) * 4, hipMemcpyDeviceToDevice, stream));
  return BTC_OK;
}

extern "C" int btc_pass_occ_vox_fill_i32(const BtcPovConfig* cfg, const float* det_voxels, int M, int P, int C, int m, int pmax, int k_total,
                                         float* voxels, int64_t* vcoords, int64_t* vnum, float* occ_pnts, int64_t* occ_b, int32_t* vcoords32,
                                         int32_t* vnum32, void* ws, size_t ws_bytes, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  BTC_CHECK_ARG(ws_bytes >= btc_pass_occ_vox_ws_bytes(cfg, M, P), "btc_pass_occ_vox_fill: workspace too small");
  PovGeom G;
  int rc = pov_geom(&G, cfg, M, P, C);
  if (rc) return rc;
  PovWs w = pov_carve(ws, G);
  if (m > 0 && pmax > 0) {
    pov_fill<<<btc_cdiv((long long)m * pmax, 256), 256, 0, stream>>>(G, m, pmax, det_voxels, w.occ_xyzp, w.offs, w.perm, w.bitmap, voxels,
                                                                     vcoords, vnum, vnum32);
    BTC_LAUNCH_CHECK();
    pov_coords<<<btc_cdiv(w.nw, 256), 256, 0, stream>>>(w.bitmap, w.prefix, w.nw, G, vcoords, vcoords32);
    BTC_LAUNCH_CHECK();
  }
  if (k_total > 0) {
    pov_compact_occ<<<btc_cdiv(G.B * G.max_k, 256), 256, 0, stream>>>(w.occ_xyzp, w.counts, G, occ_pnts, occ_b);
    BTC_LAUNCH_CHECK();
  }
  return BTC_OK;
}

extern "C" int btc_pass_occ_vox_fill(const BtcPovConfig* cfg, const float* det_voxels, int M, int P, int C, int m, int pmax, int k_total,
                                     float* voxels, int64_t* vcoords, int64_t* vnum, float* occ_pnts, int64_t* occ_b, void* ws,
                                     size_t ws_bytes, void* stream_) {
  return btc_pass_occ_vox_fill_i32(cfg, det_voxels, M, P, C, m, pmax, k_total, voxels, vcoords, vnum, occ_pnts, occ_b, nullptr, nullptr, ws, ws_bytes,
                                   stream_);
}
