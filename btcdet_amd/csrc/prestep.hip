// Per-scene pre-steps of the hot path on a batch that is already resident in HBM.
//
//   btc_range_mask_compact : DataProcessor.mask_points_and_boxes_outside_range's point half
//                            (/root/reference/btcdet/datasets/processor/data_processor.py:23-29 ->
//                            common_utils.mask_points_by_range, common_utils.py:59-62): keep x in [x_lo, x_hi] and
//                            y in [y_lo, y_hi] (both ends inclusive, z NOT tested), the same mask applied to the
//                            un-rotated copy `pre_rot_points`; numpy boolean indexing keeps input order, so this is a STABLE
//                            compaction: wave ballot + popcount ranks, one prefix sum over workgroup counts, no atomics.
//   btc_gather_rows        : DataProcessor.shuffle_points (data_processor.py:41-51), points = points[shuffle_idx].
//
// Both are HBM streams: 16 n bytes read + 16 n' written per array (points are (n, 4) f32 rows).
#include "btc_common.h"

namespace {

constexpr int PS_T = 256;

struct Range4 {
  float x0, y0, x1, y1;
};

__device__ __forceinline__ bool ps_keep(const float* __restrict__ p, const Range4 r) {
  const float x = p[0], y = p[1];
  return (x >= r.x0) & (x <= r.x1) & (y >= r.y0) & (y <= r.y1);  // NaN compares false, as in numpy
}

__global__ __launch_bounds__(PS_T) void ps_count(const float* __restrict__ pts, int n, int ld, Range4 r, int32_t* __restrict__ block_cnt) {
  const int i = blockIdx.x * PS_T + threadIdx.x;
  const int keep = (i < n) && ps_keep(pts + (size_t)i * ld, r);
  const int c = __syncthreads_count(keep);
  if (threadIdx.x == 0) block_cnt[blockIdx.x] = c;
}

template <bool VEC4>
__device__ __forceinline__ void ps_copy_row(const float* __restrict__ src, float* __restrict__ dst, int ld) {
  if (VEC4) {
    *reinterpret_cast<float4*>(dst) = *reinterpret_cast<const float4*>(src);
  } else {
    for (int c = 0; c < ld; ++c) dst[c] = src[c];
  }
}

template <bool VEC4>
__global__ __launch_bounds__(PS_T) void ps_scatter(const float* __restrict__ a, const float* __restrict__ b, int n, int lda, int ldb, Range4 r,
                                                   const int32_t* __restrict__ block_prefix, float* __restrict__ out_a,
                                                   float* __restrict__ out_b, int32_t* __restrict__ keep_idx) {
  __shared__ int s_wave[PS_T / 64];
  const int i = blockIdx.x * PS_T + threadIdx.x;
  const bool keep = (i < n) && ps_keep(a + (size_t)i * lda, r);
  const unsigned long long m = __ballot(keep);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int in_wave = __popcll(m & ((1ull << lane) - 1ull));
  if (lane == 0) s_wave[wave] = __popcll(m);
  __syncthreads();
  int base = block_prefix[blockIdx.x];
  for (int w = 0; w < wave; ++w) base += s_wave[w];
  if (!keep) return;
  const int dst = base + in_wave;
  ps_copy_row<VEC4>(a + (size_t)i * lda, out_a + (size_t)dst * lda, lda);
  if (b) ps_copy_row<VEC4>(b + (size_t)i * ldb, out_b + (size_t)dst * ldb, ldb);
  if (keep_idx) keep_idx[dst] = i;
}

// out_offsets[s] = number of kept points in front of scene s's first point; one wave per boundary
__global__ __launch_bounds__(64) void ps_offsets(const float* __restrict__ pts, int n, int ld, Range4 r, const int32_t* __restrict__ offsets,
                                                 int batch, const int32_t* __restrict__ block_prefix, const int32_t* __restrict__ total,
                                                 int32_t* __restrict__ out_offsets) {
  const int s = blockIdx.x;
  const int pos = offsets[s];
  if (pos >= n) {
    if (threadIdx.x == 0) out_offsets[s] = *total;
    return;
  }
  const int blk = pos / PS_T, start = blk * PS_T;
  int cnt = 0;
  for (int j = start + (int)threadIdx.x; j < pos; j += 64) cnt += ps_keep(pts + (size_t)j * ld, r) ? 1 : 0;
  for (int o = 32; o > 0; o >>= 1) cnt += __shfl_down(cnt, o);
  if (threadIdx.x == 0) out_offsets[s] = block_prefix[blk] + cnt;
}

template <bool VEC4>
__global__ __launch_bounds__(PS_T) void ps_gather(const float* __restrict__ src, const int32_t* __restrict__ idx, int n_out, int ld, int n_src,
                                                  float* __restrict__ out, int32_t* __restrict__ bad) {
  const int i = blockIdx.x * PS_T + threadIdx.x;
  if (i >= n_out) return;
  const int j = idx[i];
  if (j < 0 || j >= n_src) {  // numpy would raise IndexError; never read out of bounds
    if (bad) atomicAdd(bad, 1);
    for (int c = 0; c < ld; ++c) out[(size_t)i * ld + c] = 0.f;
    return;
  }
  ps_copy_row<VEC4>(src + (size_t)j * ld, out + (size_t)i * ld, ld);
}

bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

}  // namespace

extern "C" size_t btc_range_mask_ws_bytes(int n) {
  const long long nb = btc_cdiv(n > 0 ? n : 1, PS_T);
  return btc_align((size_t)(nb + 1) * sizeof(int32_t)) * 2 + 256 + btc_scan_ws_bytes(nb);
}

extern "C" int btc_range_mask_compact(const float* points, const float* points_b, int n, int ld, int ld_b, const int32_t* scene_offsets, int batch,
                                      const float* h_range_xyxy, float* out, float* out_b, int32_t* out_offsets, int32_t* keep_idx, void* ws,
                                      size_t ws_bytes, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  BTC_CHECK_ARG(n >= 0 && ld >= 2 && batch >= 1, "btc_range_mask_compact: need n >= 0, ld >= 2 (x, y columns), batch >= 1");
  BTC_CHECK_ARG(points_b == nullptr || (ld_b >= 1 && out_b != nullptr), "btc_range_mask_compact: second array needs its output and row length");
  BTC_CHECK_ARG(ws_bytes >= btc_range_mask_ws_bytes(n), "btc_range_mask_compact: workspace too small");
  BTC_CHECK_ARG(h_range_xyxy[0] <= h_range_xyxy[2] && h_range_xyxy[1] <= h_range_xyxy[3], "btc_range_mask_compact: empty range");
  const Range4 r = {h_range_xyxy[0], h_range_xyxy[1], h_range_xyxy[2], h_range_xyxy[3]};
  const int nb = btc_cdiv(n > 0 ? n : 1, PS_T);
  BtcCarver c(ws);
  int32_t* block_cnt = c.take<int32_t>(nb + 1);
  int32_t* block_prefix = c.take<int32_t>(nb + 1);
  int32_t* total = c.take<int32_t>(1);
  void* scan_ws = c.base + c.off;
  ps_count<<<nb, PS_T, 0, stream>>>(points, n, ld, r, block_cnt);
  BTC_LAUNCH_CHECK();
  int rc = btc_scan_exclusive_i32(block_cnt, block_prefix, nb, total, scan_ws, stream);
  if (rc != BTC_OK) return rc;
  const bool vec = ld == 4 && aligned16(points) && aligned16(out) && (!points_b || (ld_b == 4 && aligned16(points_b) && aligned16(out_b)));
  if (n > 0) {
    if (vec) ps_scatter<true><<<nb, PS_T, 0, stream>>>(points, points_b, n, ld, ld_b, r, block_prefix, out, out_b, keep_idx);
    else ps_scatter<false><<<nb, PS_T, 0, stream>>>(points, points_b, n, ld, ld_b, r, block_prefix, out, out_b, keep_idx);
    BTC_LAUNCH_CHECK();
  }
  ps_offsets<<<batch + 1, 64, 0, stream>>>(points, n, ld, r, scene_offsets, batch, block_prefix, total, out_offsets);
  BTC_LAUNCH_CHECK();
  return BTC_OK;
}

extern "C" int btc_gather_rows(const float* src, const int32_t* idx, int n_out, int ld, int n_src, float* out, int32_t* bad_count, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  BTC_CHECK_ARG(n_out >= 0 && ld >= 1 && n_src >= 0, "btc_gather_rows: bad sizes");
  if (n_out == 0) return BTC_OK;
  const int nb = btc_cdiv(n_out, PS_T);
  if (ld == 4 && aligned16(src) && aligned16(out)) ps_gather<true><<<nb, PS_T, 0, stream>>>(src, idx, n_out, ld, n_src, out, bad_count);
  else ps_gather<false><<<nb, PS_T, 0, stream>>>(src, idx, n_out, ld, n_src, out, bad_count);
  BTC_LAUNCH_CHECK();
  return BTC_OK;
}
