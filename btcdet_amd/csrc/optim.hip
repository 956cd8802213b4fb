// One parameter group's optimizer step of the reference's training loop in three launches.
//
// The reference steps each optimizer as clip_grad_norm_(params, 10) -> OptimWrapper.step (true_wd: p *= 1 - wd * lr, then
// torch.optim.Adam with betas = (mom, 0.99)) (tools/train_utils/train_utils.py:121-124, optimization/fastai_optim.py).  With
// torch's multi-tensor ops that is ~10 launches and five Python -> C++ conversions of a ~120-tensor list per group; the
// list handling -- not the kernels -- kept the main stream idle for ~0.4 ms per step while the host worked through it
// (tools/stream_gaps.py).  Here the group's parameters and both Adam moments live in ONE flat buffer each (the module's
// parameters are views into it), the gradients stay where autograd put them and are reached through a pointer table passed BY
// VALUE in the kernel arguments (no host-to-device copy, no staging buffer to recycle), and the step is
//   adam_sqnorm   per-chunk sums of g^2 (double)                      [chunks never straddle two parameters]
//   adam_total    their sum in index order -> total_sq (deterministic)
//   adam_apply    clip coefficient from total_sq, decoupled decay, Adam moments and update, element by element
// Arithmetic per element follows torch's fused Adam (fp32): m = lerp(m, g, 1 - b1); v = b2 v + (1 - b2) g^2;
// p -= (lr / bc1) * m / (sqrt(v) / sqrt(bc2) + eps), with g already divided by max(1, (||g|| + 1e-6) / clip).
#include "btc_common.h"

namespace {

constexpr int OPT_CHUNK = 1024;   // elements per chunk (256 threads x 4)

struct GradPtrs {
  const float* g[BTC_ADAM_MAX_SEGMENTS];
};

// chunk c: segment chunk_seg[c] (index into the pointer table of THIS launch), elements [chunk_off[c], chunk_off[c] + chunk_len[c])
// of that segment, which sit at flat offset chunk_flat[c] of the parameter / moment buffers
__global__ __launch_bounds__(256) void adam_sqnorm(GradPtrs ptrs, const int32_t* __restrict__ chunk_seg, const int32_t* __restrict__ chunk_off,
                                                   const int32_t* __restrict__ chunk_len, int chunk0, double* __restrict__ partial) {
  __shared__ double s_sum[4];
  const int c = chunk0 + blockIdx.x;
  const float* g = ptrs.g[chunk_seg[c]] + chunk_off[c];
  const int len = chunk_len[c];
  double acc = 0.0;
#pragma unroll
  for (int u = 0; u < OPT_CHUNK / 256; ++u) {
    const int e = u * 256 + threadIdx.x;
    if (e < len) {
      const float v = g[e];
      acc += (double)v * v;
    }
  }
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) acc += __shfl_down(acc, d, 64);
  if ((threadIdx.x & 63) == 0) s_sum[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) partial[c] = (s_sum[0] + s_sum[1]) + (s_sum[2] + s_sum[3]);
}

__global__ __launch_bounds__(256) void adam_total(const double* __restrict__ partial, int n_chunks, double* __restrict__ total_sq) {
  __shared__ double s_sum[256];
  double acc = 0.0;
  for (int c = threadIdx.x; c < n_chunks; c += 256) acc += partial[c];   // thread t: chunks t, t + 256, ... in order
  s_sum[threadIdx.x] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0.0;
    for (int i = 0; i < 256; ++i) t += s_sum[i];
    *total_sq = t;
  }
}

__global__ __launch_bounds__(256) void adam_apply(GradPtrs ptrs, const int32_t* __restrict__ chunk_seg, const int32_t* __restrict__ chunk_off,
                                                  const int32_t* __restrict__ chunk_len, const int64_t* __restrict__ chunk_flat, int chunk0,
                                                  const double* __restrict__ total_sq, float clip, float decay, float lr_over_bc1, float beta1,
                                                  float beta2, float inv_sqrt_bc2, float eps, float* __restrict__ p, float* __restrict__ m,
                                                  float* __restrict__ v) {
  const int c = chunk0 + blockIdx.x;
  const float* g = ptrs.g[chunk_seg[c]] + chunk_off[c];
  const int len = chunk_len[c];
  const int64_t base = chunk_flat[c];
  float inv_coef = 1.f;   // gradients are divided by max(1, (norm + 1e-6) / clip)
  if (clip > 0.f) {
    const float norm = (float)sqrt(*total_sq);
    inv_coef = fmaxf((norm + 1e-6f) / clip, 1.f);
  }
#pragma unroll
  for (int u = 0; u < OPT_CHUNK / 256; ++u) {
    const int e = u * 256 + threadIdx.x;
    if (e < len) {
      const float gr = g[e] / inv_coef;
      float pv = p[base + e] * decay;
      float mv = m[base + e], vv = v[base + e];
      mv = mv + (1.f - beta1) * (gr - mv);
      vv = beta2 * vv + (1.f - beta2) * gr * gr;
      const float denom = sqrtf(vv) * inv_sqrt_bc2 + eps;
      pv -= lr_over_bc1 * mv / denom;
      p[base + e] = pv;
      m[base + e] = mv;
      v[base + e] = vv;
    }
  }
}

// flat[chunk_flat[c] + e] = gradient e of chunk c: the pack half of a gradient bucket (btcdet_amd/grad_sync.py), one launch instead
// of a multi-tensor copy over a ~120-tensor list
__global__ __launch_bounds__(256) void grads_pack(GradPtrs ptrs, const int32_t* __restrict__ chunk_seg, const int32_t* __restrict__ chunk_off,
                                                  const int32_t* __restrict__ chunk_len, const int64_t* __restrict__ chunk_flat, int chunk0,
                                                  float* __restrict__ flat) {
  const int c = chunk0 + blockIdx.x;
  const float* g = ptrs.g[chunk_seg[c]] + chunk_off[c];
  const int len = chunk_len[c];
  float* out = flat + chunk_flat[c];
#pragma unroll
  for (int u = 0; u < OPT_CHUNK / 256; ++u) {
    const int e = u * 256 + threadIdx.x;
    if (e < len) out[e] = g[e];
  }
}

}  // namespace

extern "C" int btc_adam_max_segments(void) { return BTC_ADAM_MAX_SEGMENTS; }

extern "C" int btc_grads_pack(const float* const* grads, int n_seg, const int32_t* chunk_seg, const int32_t* chunk_off, const int32_t* chunk_len,
                              const int64_t* chunk_flat, const int32_t* seg_chunk0, float* flat, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  BTC_CHECK_ARG(n_seg >= 0, "btc_grads_pack: bad sizes");
  for (int s0 = 0; s0 < n_seg; s0 += BTC_ADAM_MAX_SEGMENTS) {
    const int s1 = s0 + BTC_ADAM_MAX_SEGMENTS < n_seg ? s0 + BTC_ADAM_MAX_SEGMENTS : n_seg;
    GradPtrs ptrs;
    for (int s = s0; s < s1; ++s) {
      BTC_CHECK_ARG(grads[s] != nullptr, "btc_grads_pack: gradient %d is NULL", s);
      ptrs.g[s - s0] = grads[s];
    }
    const int c0 = seg_chunk0[s0], c1 = seg_chunk0[s1];
    if (c1 <= c0) continue;
    grads_pack<<<c1 - c0, 256, 0, stream>>>(ptrs, chunk_seg, chunk_off, chunk_len, chunk_flat, c0, flat);
    BTC_LAUNCH_CHECK();
  }
  return BTC_OK;
}

extern "C" size_t btc_adam_group_ws_bytes(int n_chunks) { return 256 + btc_align((size_t)(n_chunks > 0 ? n_chunks : 1) * sizeof(double)); }

extern "C" int btc_adam_group_step(const float* const* grads, int n_seg, const int32_t* chunk_seg, const int32_t* chunk_off,
                                   const int32_t* chunk_len, const int64_t* chunk_flat, const int32_t* seg_chunk0, int n_chunks, float* params,
                                   float* exp_avg, float* exp_avg_sq, long long step, float lr, float beta1, float beta2, float eps,
                                   float weight_decay, float clip, void* ws, size_t ws_bytes, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  BTC_CHECK_ARG(n_seg >= 0 && n_chunks >= 0 && step >= 1, "btc_adam_group_step: bad sizes");
  if (n_seg == 0 || n_chunks == 0) return BTC_OK;
  BTC_CHECK_ARG(ws_bytes >= btc_adam_group_ws_bytes(n_chunks), "btc_adam_group_step: workspace too small");
  double* total_sq = (double*)ws;
  double* partial = (double*)((char*)ws + 256);
  const double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
  // the pointer table travels in the kernel arguments, BTC_ADAM_MAX_SEGMENTS parameters per launch; seg_chunk0[s] = first chunk of
  // segment s (seg_chunk0[n_seg] = n_chunks), chunk_seg[] holds the segment index relative to the launch's first segment
  for (int pass = 0; pass < 2; ++pass) {
    if (pass == 1 && clip > 0.f) {
      adam_total<<<1, 256, 0, stream>>>(partial, n_chunks, total_sq);
      BTC_LAUNCH_CHECK();
    }
    for (int s0 = 0; s0 < n_seg; s0 += BTC_ADAM_MAX_SEGMENTS) {
      const int s1 = s0 + BTC_ADAM_MAX_SEGMENTS < n_seg ? s0 + BTC_ADAM_MAX_SEGMENTS : n_seg;
      GradPtrs ptrs;
      for (int s = s0; s < s1; ++s) {
        BTC_CHECK_ARG(grads[s] != nullptr, "btc_adam_group_step: gradient %d is NULL", s);
        ptrs.g[s - s0] = grads[s];
      }
      const int c0 = seg_chunk0[s0], c1 = seg_chunk0[s1];
      if (c1 <= c0) continue;
      if (pass == 0) {
        if (clip > 0.f) adam_sqnorm<<<c1 - c0, 256, 0, stream>>>(ptrs, chunk_seg, chunk_off, chunk_len, c0, partial);
      } else {
        adam_apply<<<c1 - c0, 256, 0, stream>>>(ptrs, chunk_seg, chunk_off, chunk_len, chunk_flat, c0, total_sq, clip, 1.f - weight_decay * lr,
                                               (float)((double)lr / bc1), beta1, beta2, (float)(1.0 / sqrt(bc2)), eps, params, exp_avg, exp_avg_sq);
      }
      BTC_LAUNCH_CHECK();
    }
  }
  return BTC_OK;   // (the squared gradient norm of this step stays in the first 8 bytes of ws, a double)
}
