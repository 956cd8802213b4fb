// Occupancy head losses, fused (forward: one launch, backward: one launch).
//
// Replaces OccHeadTemplate.get_loss -> get_cls_layer_loss / get_res_layer_loss / mean_masked_loss
// (/root/reference/btcdet/models/occ_pnt/occ_dense_heads/occ_head_template.py:52-111) with the loss functions of
// /root/reference/btcdet/utils/loss_utils.py:140-152 (softmax focal, alpha 1, gamma 2, eps 1e-6 added to the softmax)
// and :199-233 (smooth L1, beta = res_beta): there, two `nonzero` syncs, five fancy-index gathers of dense
// [B,C,9,157,209] maps and ~50 elementwise launches plus their autograd twins.
//   cls = fore_cls_weight * sum_{cells in general_cls_loss_mask} w_c * FL(softmax(logit_c), onehot(pos_c)) / max(sum w_c, 1)
//   reg = fore_res_weight * sum_{cells in general_reg_loss_mask} w_r * sum_k smoothL1(res_k - target_k) / max(sum w_r, 1)
#include "btc_common.h"

namespace {

struct LossParams {
  long long ncell;  // cells per scene (nz*ny*nx)
  int B;
  float beta, eps, w_cls, w_res;
};

__device__ __forceinline__ bool last_block_l(int32_t* counter) {
  __shared__ int s_last;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) s_last = (btc_ticket_take(counter) == (int)gridDim.x - 1);   // (the protocol: btc_common.h)
  __syncthreads();
  if (!s_last) return false;
  if (threadIdx.x == 0) btc_ticket_acquire();
  __syncthreads();
  return true;
}

// per-cell terms; returns weighted losses
__device__ __forceinline__ void cell_terms(const float* __restrict__ logit, const float* __restrict__ res, const float* __restrict__ tgt,
                                           const LossParams& P, int b, long long sp, int pos, float* fl, float sl[3]) {
  const float z0 = logit[((size_t)b * 2 + 0) * P.ncell + sp], z1 = logit[((size_t)b * 2 + 1) * P.ncell + sp];
  const float m = fmaxf(z0, z1);
  const float e0 = expf(z0 - m), e1 = expf(z1 - m);
  const float inv = 1.0f / (e0 + e1);
  const float p = (pos ? e1 : e0) * inv + P.eps;
  const float om = 1.0f - p;
  *fl = -(om * om) * logf(p);
  if (res) {
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      float d = fabsf(res[((size_t)b * 3 + k) * P.ncell + sp] - tgt[((size_t)b * 3 + k) * P.ncell + sp]);
      sl[k] = d < P.beta ? 0.5f * d * d / P.beta : d - 0.5f * P.beta;
    }
  }
}

// sums[0] = sum w_c FL, sums[1] = sum w_c, sums[2] = sum w_r SL1, sums[3] = sum w_r ; out[0] = cls, out[1] = reg
__global__ __launch_bounds__(256) void occ_loss_fwd(const float* __restrict__ logit, const float* __restrict__ res, const float* __restrict__ tgt,
                                                    const uint8_t* __restrict__ pos_mask, const uint8_t* __restrict__ cls_mask,
                                                    const float* __restrict__ cls_w, const uint8_t* __restrict__ reg_mask,
                                                    const float* __restrict__ reg_w, LossParams P, double* __restrict__ partial,
                                                    int32_t* __restrict__ counter, float* __restrict__ out, float* __restrict__ norms,
                                                    int with_total) {
  __shared__ double s_red[4][4];
  double acc[4] = {0, 0, 0, 0};
  const long long total = (long long)P.B * P.ncell;
  for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long long)gridDim.x * blockDim.x) {
    const int cm = cls_mask[t], rm = res ? reg_mask[t] : 0;
    if (!cm && !rm) continue;
    const int b = (int)(t / P.ncell);
    const long long sp = t % P.ncell;
    float fl, sl[3] = {0.f, 0.f, 0.f};
    cell_terms(logit, rm ? res : nullptr, tgt, P, b, sp, pos_mask[t], &fl, sl);
    if (cm) { acc[0] += (double)cls_w[t] * fl; acc[1] += cls_w[t]; }
    if (rm) { acc[2] += (double)reg_w[t] * ((double)sl[0] + sl[1] + sl[2]); acc[3] += reg_w[t]; }
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc[q] += __shfl_down(acc[q], o, 64);
    if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6][q] = acc[q];
  }
  __syncthreads();
  if (threadIdx.x < 4) {
    double v = s_red[0][threadIdx.x] + s_red[1][threadIdx.x] + s_red[2][threadIdx.x] + s_red[3][threadIdx.x];
    btc_st_agent(&partial[(size_t)blockIdx.x * 4 + threadIdx.x], v);
  }
  if (!last_block_l(counter)) return;
  // the last workgroup: thread t sums quantity t % 4 over blocks t / 4, t / 4 + 64, ... in order, then a fixed tree over the 64 threads
  // of each quantity (one thread walking every block's four partials was most of this launch)
  {
    const int q = threadIdx.x & 3;
    double v = 0.0;
    for (int g = threadIdx.x >> 2; g < (int)gridDim.x; g += 64) v += btc_ld_agent(&partial[(size_t)g * 4 + q]);
#pragma unroll
    for (int o = 32; o >= 4; o >>= 1) v += __shfl_down(v, o, 64);   // lanes 0..3 of a wave: its 16 threads of quantity 0..3
    __syncthreads();
    if ((threadIdx.x & 63) < 4) s_red[threadIdx.x >> 6][q] = v;
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    double s[4];
    for (int q = 0; q < 4; ++q) s[q] = (s_red[0][q] + s_red[1][q]) + (s_red[2][q] + s_red[3][q]);
    const double nc = s[1] > 1.0 ? s[1] : 1.0, nr = s[3] > 1.0 ? s[3] : 1.0;   // clamp(sum w, min=1)
    out[0] = (float)(s[0] / nc * P.w_cls);
    out[1] = (float)(s[2] / nr * P.w_res);
    if (with_total) out[2] = out[0] + out[1];   // (the fp32 sum `cls + reg` the caller would otherwise make with one more launch)
    norms[0] = (float)(P.w_cls / nc);
    norms[1] = (float)(P.w_res / nr);
    __hip_atomic_store(counter, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// d_logit, d_res dense; g[0], g[g_stride] = upstream gradients of the two scalars (g_stride = 0: one gradient of their sum).
// fill = 0: the caller pre-zeroed both maps, only cells inside a mask are written; fill = 1: every cell is written (zeros outside the masks)
__global__ __launch_bounds__(256) void occ_loss_bwd(const float* __restrict__ logit, const float* __restrict__ res, const float* __restrict__ tgt,
                                                    const uint8_t* __restrict__ pos_mask, const uint8_t* __restrict__ cls_mask,
                                                    const float* __restrict__ cls_w, const uint8_t* __restrict__ reg_mask,
                                                    const float* __restrict__ reg_w, LossParams P, const float* __restrict__ norms,
                                                    const float* __restrict__ g, float* __restrict__ d_logit, float* __restrict__ d_res,
                                                    int g_stride, int fill) {
  const long long total = (long long)P.B * P.ncell;
  long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= total) return;
  const int cm = cls_mask[t], rm = res ? reg_mask[t] : 0;
  if (!cm && !rm && !fill) return;
  const int b = (int)(t / P.ncell);
  const long long sp = t % P.ncell;
  if (fill) {
    if (!cm) {
      d_logit[((size_t)b * 2 + 0) * P.ncell + sp] = 0.f;
      d_logit[((size_t)b * 2 + 1) * P.ncell + sp] = 0.f;
    }
    if (!rm && d_res) {
#pragma unroll
      for (int k = 0; k < 3; ++k) d_res[((size_t)b * 3 + k) * P.ncell + sp] = 0.f;
    }
  }
  if (cm) {
    const size_t i0 = ((size_t)b * 2 + 0) * P.ncell + sp, i1 = ((size_t)b * 2 + 1) * P.ncell + sp;
    const float z0 = logit[i0], z1 = logit[i1];
    const float m = fmaxf(z0, z1);
    const float e0 = expf(z0 - m), e1 = expf(z1 - m);
    const float inv = 1.0f / (e0 + e1);
    const float s0 = e0 * inv, s1 = e1 * inv;
    const int pos = pos_mask[t];
    const float st = pos ? s1 : s0;          // softmax of the target class
    const float p = st + P.eps, om = 1.0f - p;
    const float dfdp = 2.0f * om * logf(p) - om * om / p;   // d/dp [ -(1-p)^2 log p ]
    const float scale = g[0] * norms[0] * cls_w[t] * dfdp;
    // ds_t/dz_t = s_t (1 - s_t), ds_t/dz_other = -s_t s_other
    const float so = pos ? s0 : s1;
    const float dt = scale * st * (1.0f - st), dother = -scale * st * so;
    d_logit[i0] = pos ? dother : dt;
    d_logit[i1] = pos ? dt : dother;
  }
  if (rm) {
    const float scale = g[g_stride] * norms[1] * reg_w[t];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const size_t i = ((size_t)b * 3 + k) * P.ncell + sp;
      float d = res[i] - tgt[i];
      float ad = fabsf(d);
      float gr = ad < P.beta ? d / P.beta : (d > 0.f ? 1.0f : (d < 0.f ? -1.0f : 0.0f));
      d_res[i] = scale * gr;
    }
  }
}

}  // namespace

extern "C" size_t btc_occ_loss_ws_bytes(void) { return 256 + 1024 * 4 * sizeof(double); }

static int occ_loss_fwd_impl(const float* logit, const float* res, const float* res_target, const uint8_t* pos_mask,
                             const uint8_t* cls_mask, const float* cls_w, const uint8_t* reg_mask, const float* reg_w, int B,
                             long long ncell, float beta, float w_cls, float w_res, float* out2, float* norms2, void* ws, size_t ws_bytes,
                             void* stream_, int with_total) {
  hipStream_t stream = (hipStream_t)stream_;
  BTC_CHECK_ARG(ws_bytes >= btc_occ_loss_ws_bytes(), "btc_occ_loss_fwd: workspace too small");
  LossParams P{ncell, B, beta, 1e-6f, w_cls, w_res};
  int32_t* counter = (int32_t*)ws;  // first 256 bytes zero on entry / exit
  double* partial = (double*)((char*)ws + 256);
  long long total = (long long)B * ncell;
  int grid = btc_cdiv(total, 256 * 8);
  if (grid > 1024) grid = 1024;
  if (grid < 1) grid = 1;
  occ_loss_fwd<<<grid, 256, 0, stream>>>(logit, res, res_target, pos_mask, cls_mask, cls_w, reg_mask, reg_w, P, partial, counter, out2, norms2, with_total);
  BTC_LAUNCH_CHECK();
  return BTC_OK;
}

extern "C" int btc_occ_loss_fwd(const float* logit, const float* res, const float* res_target, const uint8_t* pos_mask,
                                const uint8_t* cls_mask, const float* cls_w, const uint8_t* reg_mask, const float* reg_w, int B,
                                long long ncell, float beta, float w_cls, float w_res, float* out2, float* norms2, void* ws, size_t ws_bytes,
                                void* stream_) {
  return occ_loss_fwd_impl(logit, res, res_target, pos_mask, cls_mask, cls_w, reg_mask, reg_w, B, ncell, beta, w_cls, w_res, out2, norms2, ws,
                           ws_bytes, stream_, 0);
}

// the same with out3[2] = out3[0] + out3[1] (the loss the head returns): no separate add launch, and one upstream gradient in backward
extern "C" int btc_occ_loss_fwd_total(const float* logit, const float* res, const float* res_target, const uint8_t* pos_mask,
                                      const uint8_t* cls_mask, const float* cls_w, const uint8_t* reg_mask, const float* reg_w, int B,
                                      long long ncell, float beta, float w_cls, float w_res, float* out3, float* norms2, void* ws,
                                      size_t ws_bytes, void* stream_) {
  return occ_loss_fwd_impl(logit, res, res_target, pos_mask, cls_mask, cls_w, reg_mask, reg_w, B, ncell, beta, w_cls, w_res, out3, norms2, ws,
                           ws_bytes, stream_, 1);
}

extern "C" int btc_occ_loss_bwd(const float* logit, const float* res, const float* res_target, const uint8_t* pos_mask,
                                const uint8_t* cls_mask, const float* cls_w, const uint8_t* reg_mask, const float* reg_w, int B,
                                long long ncell, float beta, const float* norms2, const float* grad2, float* d_logit, float* d_res,
                                void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  LossParams P{ncell, B, beta, 1e-6f, 0.f, 0.f};
  long long total = (long long)B * ncell;
  occ_loss_bwd<<<btc_cdiv(total, 256), 256, 0, stream>>>(logit, res, res_target, pos_mask, cls_mask, cls_w, reg_mask, reg_w, P, norms2, grad2,
                                                         d_logit, d_res, 1, 0);
  BTC_LAUNCH_CHECK();
  return BTC_OK;
}

// backward of btc_occ_loss_fwd_total: ONE upstream gradient (of the sum); every cell of d_logit / d_res is written -- the caller allocates
// them uninitialised (no two fill launches)
extern "C" int btc_occ_loss_bwd_total(const float* logit, const float* res, const float* res_target, const uint8_t* pos_mask,
                                      const uint8_t* cls_mask, const float* cls_w, const uint8_t* reg_mask, const float* reg_w, int B,
                                      long long ncell, float beta, const float* norms2, const float* grad_total, float* d_logit, float* d_res,
                                      void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  LossParams P{ncell, B, beta, 1e-6f, 0.f, 0.f};
  long long total = (long long)B * ncell;
  occ_loss_bwd<<<btc_cdiv(total, 256), 256, 0, stream>>>(logit, res, res_target, pos_mask, cls_mask, cls_w, reg_mask, reg_w, P, norms2, grad_total,
                                                         d_logit, d_res, 0, 1);
  BTC_LAUNCH_CHECK();
  return BTC_OK;
}
