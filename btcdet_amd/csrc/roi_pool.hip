// ROI head, pooling stage: trilinear read-out of a sparse feature volume at the micro-scenes' lattice points
// (/root/reference/btcdet/models/roi_heads/conv_head.py:509-610 -> common_utils.py:247-311 bilinear / trilinear interpolation with
// normalize = False) without densifying the volume and without touching the 80 % of the points that read nothing but zeros.
//
//   trilinear_corners   one thread per lattice point: world xyz -> fractional cell (z, y, x) of the (strided) grid, the 8 corner cells'
//                       ROW in the sparse tensor (through a cell -> row volume, -1 = empty / out of range) and weight, and whether any
//                       corner can contribute at all (the reference keeps a point iff its interpolated vector has a non-zero entry)
//   trilinear_gather    one wave per KEPT point: out[p] = sum_c w[p][c] * feat[row[p][c]] in corner order (dz, dy, dx) -- the sum the
//                       reference forms term by term
//   trilinear_scatter   backward, DETERMINISTIC: one wave per feature ROW walks the (point, corner) pairs that read it -- the host side
//                       hands them over stably sorted by row -- and sums in that order: no float atomics
#include "btc_common.h"

namespace {

struct TriGeom {
  float lo[3];      // point-cloud range minimum (x, y, z)
  float vs[3];      // voxel size (x, y, z)
  float stride[3];  // stride of the tensor's grid along (z, y, x)
  int D, H, W, B;
};

__global__ __launch_bounds__(256) void trilinear_corners(const float* __restrict__ xyz, long long Q, long long per_batch, TriGeom g,
                                                         const int32_t* __restrict__ cell_row, const uint8_t* __restrict__ row_live,
                                                         int32_t* __restrict__ rows, float* __restrict__ wts, uint8_t* __restrict__ flag) {
  const long long q = (long long)blockIdx.x * 256 + threadIdx.x;
  if (q >= Q) return;
  const int b = (int)(q / per_batch);
  const float px = xyz[q * 3 + 0], py = xyz[q * 3 + 1], pz = xyz[q * 3 + 2];
  // (p - lo) / voxel / stride - 0.5 as torch evaluates it on the device: a division by a host scalar is a multiplication by its fp32
  // reciprocal (ATen div_true_kernel: `a * (1 / b)`), one rounding per operation
  const float fz = __fsub_rn(__fmul_rn(__fmul_rn(__fsub_rn(pz, g.lo[2]), __fdiv_rn(1.f, g.vs[2])), __fdiv_rn(1.f, g.stride[0])), 0.5f);
  const float fy = __fsub_rn(__fmul_rn(__fmul_rn(__fsub_rn(py, g.lo[1]), __fdiv_rn(1.f, g.vs[1])), __fdiv_rn(1.f, g.stride[1])), 0.5f);
  const float fx = __fsub_rn(__fmul_rn(__fmul_rn(__fsub_rn(px, g.lo[0]), __fdiv_rn(1.f, g.vs[0])), __fdiv_rn(1.f, g.stride[2])), 0.5f);
  const float lz = floorf(fz), ly = floorf(fy), lx = floorf(fx);
  const float rz = __fsub_rn(fz, lz), ry = __fsub_rn(fy, ly), rx = __fsub_rn(fx, lx);
  const int iz = (int)lz, iy = (int)ly, ix = (int)lx;
  bool any = false;
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    const int dz = c >> 2, dy = (c >> 1) & 1, dx = c & 1;
    const int cz = iz + dz, cy = iy + dy, cx = ix + dx;
    const bool inside = cz >= 0 && cz < g.D && cy >= 0 && cy < g.H && cx >= 0 && cx < g.W && b < g.B;
    const float w = __fmul_rn(__fmul_rn(dz ? rz : __fsub_rn(1.f, rz), dy ? ry : __fsub_rn(1.f, ry)), dx ? rx : __fsub_rn(1.f, rx));
    int r = -1;
    if (inside) r = cell_row[(((long long)b * g.D + cz) * g.H + cy) * g.W + cx];
    const float wa = r >= 0 ? fabsf(w) : 0.f;
    rows[q * 8 + c] = r;
    wts[q * 8 + c] = wa;
    any = any || (r >= 0 && wa != 0.f && (!row_live || row_live[r]));   // (an all-zero row is read -- and gets its gradient -- but keeps no point alive)
  }
  flag[q] = any ? 1 : 0;
}

// out[p][:] = sum over the 8 corners, in corner order, of w * feat[row]; rows / wts: the KEPT points' (M, 8) tables
__global__ __launch_bounds__(256) void trilinear_gather(const float* __restrict__ feat, int C, long long M, const int32_t* __restrict__ rows,
                                                        const float* __restrict__ wts, float* __restrict__ out) {
  const long long p = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (p >= M) return;
  int r[8];
  float w[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    r[c] = rows[p * 8 + c];
    w[c] = wts[p * 8 + c];
  }
  for (int ch = lane; ch < C; ch += 64) {
    // the reference forms term_0 + term_1 + ... + term_7 with a term for EVERY corner; an absent corner's term is +0 (adding it changes
    // nothing but the sign of an all-zero sum)
    float acc = r[0] >= 0 ? __fmul_rn(feat[(long long)r[0] * C + ch], w[0]) : 0.f;
#pragma unroll
    for (int c = 1; c < 8; ++c)
      acc = __fadd_rn(acc, r[c] >= 0 ? __fmul_rn(feat[(long long)r[c] * C + ch], w[c]) : 0.f);
    out[p * C + ch] = acc;
  }
}

// backward, deterministic: the (point, corner) pairs e = 8 p + c arrive STABLY SORTED by the row they read (seg[row] .. seg[row + 1]).
// One 256-thread workgroup per row: thread (way, channel) adds w * grad_out[p] over the pairs  way, way + WAYS, ...  of the segment in
// four interleaved partial sums (independent chains: the loads of four pairs are in flight at once), the partial sums meet in a
// FIXED order -- no float atomics, run-to-run identical.  (One wave per row walked the longest row -- a few thousand pairs -- alone:
// 1.5 ms for the launch.)
template <int C_T>
__global__ __launch_bounds__(256) void trilinear_scatter(const float* __restrict__ grad_out, int C, const int32_t* __restrict__ pair,
                                                         const int32_t* __restrict__ seg, const float* __restrict__ wts, int n_rows,
                                                         float* __restrict__ grad_feat) {
  constexpr int WAYS = 256 / C_T;
  __shared__ float part[256];
  const int row = blockIdx.x;
  const int ch = threadIdx.x % C_T, way = threadIdx.x / C_T;
  const int s = seg[row], e_end = seg[row + 1];
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  int i = s + way;
  for (; i + 3 * WAYS < e_end; i += 4 * WAYS) {
    const int32_t e0 = pair[i], e1 = pair[i + WAYS], e2 = pair[i + 2 * WAYS], e3 = pair[i + 3 * WAYS];
    a0 = __fadd_rn(a0, __fmul_rn(grad_out[(long long)(e0 >> 3) * C + ch], wts[e0]));
    a1 = __fadd_rn(a1, __fmul_rn(grad_out[(long long)(e1 >> 3) * C + ch], wts[e1]));
    a2 = __fadd_rn(a2, __fmul_rn(grad_out[(long long)(e2 >> 3) * C + ch], wts[e2]));
    a3 = __fadd_rn(a3, __fmul_rn(grad_out[(long long)(e3 >> 3) * C + ch], wts[e3]));
  }
  for (; i < e_end; i += WAYS) {
    const int32_t e0 = pair[i];
    a0 = __fadd_rn(a0, __fmul_rn(grad_out[(long long)(e0 >> 3) * C + ch], wts[e0]));
  }
  part[threadIdx.x] = __fadd_rn(__fadd_rn(a0, a1), __fadd_rn(a2, a3));
  __syncthreads();
  if (way == 0) {
    float acc = part[ch];
#pragma unroll
    for (int w = 1; w < WAYS; ++w) acc = __fadd_rn(acc, part[w * C_T + ch]);
    if (ch < C) grad_feat[(long long)row * C + ch] = acc;
  }
}

// any channel count: one wave per row, lanes stride over the channels
__global__ __launch_bounds__(256) void trilinear_scatter_any(const float* __restrict__ grad_out, int C, const int32_t* __restrict__ pair,
                                                             const int32_t* __restrict__ seg, const float* __restrict__ wts, int n_rows,
                                                             float* __restrict__ grad_feat) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= n_rows) return;
  const int s = seg[row], e_end = seg[row + 1];
  for (int ch = lane; ch < C; ch += 64) {
    float acc = 0.f;
    for (int i = s; i < e_end; ++i) {
      const int32_t e = pair[i];
      acc = __fadd_rn(acc, __fmul_rn(grad_out[(long long)(e >> 3) * C + ch], wts[e]));
    }
    grad_feat[(long long)row * C + ch] = acc;
  }
}

}  // namespace

extern "C" int btc_trilinear_corners(const float* xyz, long long n_points, long long points_per_batch, const float* range_lo, const float* voxel,
                                     const float* stride_zyx, const int32_t* grid_dhw, int batch, const int32_t* cell_row, const uint8_t* row_live,
                                     int32_t* rows, float* weights, uint8_t* flag, void* stream_) {
  BTC_CHECK_ARG(n_points >= 0 && points_per_batch >= 1 && batch >= 1, "btc_trilinear_corners: bad sizes");
  if (n_points == 0) return BTC_OK;
  TriGeom g;
  for (int j = 0; j < 3; ++j) { g.lo[j] = range_lo[j]; g.vs[j] = voxel[j]; g.stride[j] = stride_zyx[j]; }
  g.D = grid_dhw[0]; g.H = grid_dhw[1]; g.W = grid_dhw[2]; g.B = batch;
  trilinear_corners<<<btc_cdiv(n_points, 256), 256, 0, (hipStream_t)stream_>>>(xyz, n_points, points_per_batch, g, cell_row, row_live, rows, weights, flag);
  BTC_LAUNCH_CHECK();
  return BTC_OK;
}

extern "C" int btc_trilinear_gather(const float* feat, int C, long long n_keep, const int32_t* rows, const float* weights, float* out, void* stream_) {
  BTC_CHECK_ARG(C >= 1 && n_keep >= 0, "btc_trilinear_gather: bad sizes");
  if (n_keep == 0) return BTC_OK;
  trilinear_gather<<<btc_cdiv(n_keep, 4), 256, 0, (hipStream_t)stream_>>>(feat, C, n_keep, rows, weights, out);
  BTC_LAUNCH_CHECK();
  return BTC_OK;
}

extern "C" int btc_trilinear_scatter(const float* grad_out, int C, const int32_t* pair_sorted, const int32_t* seg, const float* weights, int n_rows,
                                     float* grad_feat, void* stream_) {
  BTC_CHECK_ARG(C >= 1 && n_rows >= 0, "btc_trilinear_scatter: bad sizes");
  if (n_rows == 0) return BTC_OK;
  hipStream_t stream = (hipStream_t)stream_;
  if (C == 128) trilinear_scatter<128><<<n_rows, 256, 0, stream>>>(grad_out, C, pair_sorted, seg, weights, n_rows, grad_feat);
  else if (C == 64) trilinear_scatter<64><<<n_rows, 256, 0, stream>>>(grad_out, C, pair_sorted, seg, weights, n_rows, grad_feat);
  else if (C == 32) trilinear_scatter<32><<<n_rows, 256, 0, stream>>>(grad_out, C, pair_sorted, seg, weights, n_rows, grad_feat);
  else trilinear_scatter_any<<<btc_cdiv(n_rows, 4), 256, 0, stream>>>(grad_out, C, pair_sorted, seg, weights, n_rows, grad_feat);
  BTC_LAUNCH_CHECK();
  return BTC_OK;
}
