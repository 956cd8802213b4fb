// Sorted-unique re-voxelization for gfx950.
//
// Replaces PassOccVox's GPU re-voxelization (/root/reference/btcdet/models/occ_pnt/
// add_occ_template.py:248-268): torch.unique(coords, dim=0, sorted=True, return_inverse,
// return_counts) (a multi-key sort of ~40k x 4 int64) + sort(inverse) + scatter-pad, plus a
// `.cpu()` sync for Pmax.  Here: the occupied cells are a bitmap over the detection grid ranked
// by a popcount prefix sum (cells come out lexicographically ascending in (b,z,y,x) with no
// sort), per-cell member lists are built with atomics and put back into input order by a
// per-cell insertion sort (lists are a few entries long), and one read-back returns (M, Pmax).
#include "btc_common.h"

namespace {

struct RvGeom {
  int D, H, W;
  long long vol;
};

__device__ __forceinline__ bool rv_cell(const int64_t* __restrict__ c, const RvGeom& g, int batch, unsigned* cell) {
  long long b = c[0], z = c[1], y = c[2], x = c[3];
  if (b < 0 || b >= batch || z < 0 || z >= g.D || y < 0 || y >= g.H || x < 0 || x >= g.W) return false;
  *cell = (unsigned)(b * g.vol + (z * g.H + y) * g.W + x);
  return true;
}

__global__ __launch_bounds__(256) void rv_mark(const int64_t* __restrict__ coords, int n, int batch, RvGeom g,
                                               unsigned* __restrict__ bitmap) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  unsigned cell;
  if (rv_cell(coords + (size_t)i * 4, g, batch, &cell)) atomicOr(&bitmap[cell >> 5], 1u << (cell & 31));
}

__global__ __launch_bounds__(256) void rv_popc(const unsigned* __restrict__ bitmap, long long nwords, int32_t* __restrict__ counts) {
  long long w = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (w > nwords) return;
  counts[w] = (w < nwords) ? __popc(bitmap[w]) : 0;
}

__global__ __launch_bounds__(256) void rv_count(const int64_t* __restrict__ coords, int n, int batch, RvGeom g,
                                                const unsigned* __restrict__ bitmap, const int32_t* __restrict__ prefix,
                                                int32_t* __restrict__ point_row, int32_t* __restrict__ cnt) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  unsigned cell;
  int row = -1;
  if (rv_cell(coords + (size_t)i * 4, g, batch, &cell)) {
    unsigned w = cell >> 5, bit = cell & 31;
    row = prefix[w] + __popc(bitmap[w] & ((1u << bit) - 1u));
    atomicAdd(&cnt[row], 1);
  }
  point_row[i] = row;
}

__global__ __launch_bounds__(256) void rv_pmax(const int32_t* __restrict__ cnt, const int32_t* __restrict__ d_m,
                                               int32_t* __restrict__ d_pmax) {
  int r = blockIdx.x * blockDim.x + threadIdx.x;
  int v = (r < *d_m) ? cnt[r] : 0;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = max(v, __shfl_down(v, o, 64));
  if ((threadIdx.x & 63) == 0 && v > 0) atomicMax(d_pmax, v);
}

__global__ __launch_bounds__(256) void rv_scatter(const int32_t* __restrict__ point_row, int n, const int32_t* __restrict__ offs,
                                                  int32_t* __restrict__ cursor, int32_t* __restrict__ perm) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  int row = point_row[i];
  if (row < 0) return;
  int pos = atomicAdd(&cursor[row], 1);
  perm[offs[row] + pos] = i;
}

// one thread per (cell, slot): the cell's member list is first put into ascending point order by
// its slot-0 thread ... simpler and race-free: every thread of the cell finds the slot-th smallest
// member by rank counting (lists are short), then copies that point.
__global__ __launch_bounds__(256) void rv_fill(const float* __restrict__ points, const int64_t* __restrict__ coords, int C,
                                               int m, int pmax, const int32_t* __restrict__ offs, const int32_t* __restrict__ perm,
                                               float* __restrict__ voxels, int64_t* __restrict__ vcoords,
                                               int64_t* __restrict__ vnum) {
  long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (long long)m * pmax) return;
  int row = (int)(t / pmax), slot = (int)(t % pmax);
  int beg = offs[row], cnt = offs[row + 1] - beg;
  float* o = voxels + (size_t)t * C;
  if (slot < cnt) {
    // member with exactly `slot` smaller members = slot-th point of the cell in input order
    int mine = -1;
    for (int a = 0; a < cnt; ++a) {
      int ia = perm[beg + a];
      int rank = 0;
      for (int b = 0; b < cnt; ++b) rank += (perm[beg + b] < ia);
      if (rank == slot) { mine = ia; break; }
    }
    const float* src = points + (size_t)mine * C;
    for (int c = 0; c < C; ++c) o[c] = src[c];
    if (slot == 0) {
      const int64_t* cs = coords + (size_t)mine * 4;
      int64_t* cd = vcoords + (size_t)row * 4;
      cd[0] = cs[0]; cd[1] = cs[1]; cd[2] = cs[2]; cd[3] = cs[3];
      vnum[row] = cnt;
    }
  } else {
    for (int c = 0; c < C; ++c) o[c] = 0.f;
  }
}

struct RvWs {
  unsigned* bitmap;
  int32_t* prefix;
  int32_t* point_row;
  int32_t* cnt;     // n+1
  int32_t* offs;    // n+1
  int32_t* cursor;  // n
  int32_t* perm;    // n
  void* scan_ws;
  long long nw;
};

long long rv_nwords(int batch, const int32_t* shape) {
  long long cells = (long long)batch * shape[0] * shape[1] * shape[2];
  return (cells + 31) / 32;
}

RvWs rv_carve(void* ws, int n, int batch, const int32_t* shape) {
  RvWs w;
  w.nw = rv_nwords(batch, shape);
  BtcCarver cv(ws);
  w.bitmap = cv.take<unsigned>(w.nw);
  w.prefix = cv.take<int32_t>(w.nw + 1);
  w.point_row = cv.take<int32_t>(n + 1);
  w.cnt = cv.take<int32_t>(n + 1);
  w.offs = cv.take<int32_t>(n + 1);
  w.cursor = cv.take<int32_t>(n + 1);
  w.perm = cv.take<int32_t>(n + 1);
  long long big = w.nw + 1 > n + 1 ? w.nw + 1 : n + 1;
  w.scan_ws = cv.take<char>(btc_scan_ws_bytes(big));
  return w;
}

}  // namespace

extern "C" size_t btc_revoxelize_ws_bytes(int n, int batch, const int32_t* h_shape) {
  long long nw = rv_nwords(batch, h_shape);
  long long big = nw + 1 > n + 1 ? nw + 1 : n + 1;
  return btc_align((size_t)nw * 4) + btc_align((size_t)(nw + 1) * 4) + 5 * btc_align((size_t)(n + 1) * 4) +
         btc_scan_ws_bytes(big);
}

extern "C" int btc_revoxelize_count(const int64_t* coords, int n, int batch, const int32_t* h_shape, int32_t* d_m,
                                    int32_t* d_pmax, void* ws, size_t ws_bytes, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  BTC_CHECK_ARG(n >= 0 && batch >= 1, "btc_revoxelize_count: bad sizes");
  BTC_CHECK_ARG(ws_bytes >= btc_revoxelize_ws_bytes(n, batch, h_shape), "btc_revoxelize_count: workspace too small");
  RvGeom g{h_shape[0], h_shape[1], h_shape[2], (long long)h_shape[0] * h_shape[1] * h_shape[2]};
  if (g.vol * batch >= 0x7fffffffLL) {
    btc_set_error("btc_revoxelize: batch*grid volume %lld exceeds 32-bit cell keys", g.vol * batch);
    return BTC_ERANGE;
  }
  RvWs w = rv_carve(ws, n, batch, h_shape);
  BTC_HIP(hipMemsetAsync(w.bitmap, 0, (size_t)w.nw * 4, stream));
  BTC_HIP(hipMemsetAsync(w.cnt, 0, (size_t)(n + 1) * 4, stream));
  BTC_HIP(hipMemsetAsync(w.cursor, 0, (size_t)(n + 1) * 4, stream));
  BTC_HIP(hipMemsetAsync(d_pmax, 0, 4, stream));
  if (n > 0) {
    rv_mark<<<btc_cdiv(n, 256), 256, 0, stream>>>(coords, n, batch, g, w.bitmap);
    BTC_LAUNCH_CHECK();
  }
  rv_popc<<<btc_cdiv(w.nw + 1, 256), 256, 0, stream>>>(w.bitmap, w.nw, w.prefix);
  BTC_LAUNCH_CHECK();
  int rc = btc_scan_exclusive_i32(w.prefix, w.prefix, w.nw + 1, d_m, w.scan_ws, stream);
  if (rc) return rc;
  if (n > 0) {
    rv_count<<<btc_cdiv(n, 256), 256, 0, stream>>>(coords, n, batch, g, w.bitmap, w.prefix, w.point_row, w.cnt);
    BTC_LAUNCH_CHECK();
    rv_pmax<<<btc_cdiv(n, 256), 256, 0, stream>>>(w.cnt, d_m, d_pmax);
    BTC_LAUNCH_CHECK();
    rc = btc_scan_exclusive_i32(w.cnt, w.offs, n + 1, nullptr, w.scan_ws, stream);
    if (rc) return rc;
    rv_scatter<<<btc_cdiv(n, 256), 256, 0, stream>>>(w.point_row, n, w.offs, w.cursor, w.perm);
    BTC_LAUNCH_CHECK();
  }
  return BTC_OK;
}

extern "C" int btc_revoxelize_fill(const float* points, const int64_t* coords, int n, int C, int batch,
                                   const int32_t* h_shape, int m, int pmax, float* voxels, int64_t* vcoords, int64_t* vnum,
                                   void* ws, size_t ws_bytes, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  BTC_CHECK_ARG(ws_bytes >= btc_revoxelize_ws_bytes(n, batch, h_shape), "btc_revoxelize_fill: workspace too small");
  BTC_CHECK_ARG(m <= n, "btc_revoxelize_fill: m > n");
  if (m <= 0 || pmax <= 0) return BTC_OK;
  RvWs w = rv_carve(ws, n, batch, h_shape);
  rv_fill<<<btc_cdiv((long long)m * pmax, 256), 256, 0, stream>>>(points, coords, C, m, pmax, w.offs, w.perm, voxels, vcoords, vnum);
  BTC_LAUNCH_CHECK();
  return BTC_OK;
}
