// Small fused helpers of the training step (round 5): each replaces a chain of 3-7 torch elementwise / fill / copy launches between the
// sparse kernels -- the step is bound by its launch count (420 per 4.4 ms in round 4, 108 of them torch's or rocclr's), not by bytes.
//   btc_dense_split_fwd / _bwd : SparseConvTensor.dense() of the occupancy head's MERGED output (N, Ca + Cb) into its two dense maps
//                                (logits, residuals) in one launch each way (was: two column slices + copies, two fills, two scatters;
//                                backward two gathers, two slice_backward fills + copies, one add).  Same values as dense() of each part.
//   btc_cat_pad_fwd / _bwd     : sparse_cat of the detection backbone (spconv_backbone.py:869-873: features | occupancy code) with the
//                                zero channels ops._pad_in_channels appends (34 -> 64) in one launch each way (was cat + fill + copy).
//   btc_sumsq2_fwd / _bwd      : the L2 stand-in loss of the out-of-scope heads (trainer.stand_in_det_loss: ka mean(a^2) + kb mean(b^2))
//                                over both tensors: one reduction launch forward (fp64 partials, last-arriver sum in block order:
//                                deterministic), one elementwise launch backward (was 7 + 6 launches).
#include <type_traits>

#include "btc_common.h"

namespace {

__global__ __launch_bounds__(256) void dense_split_fwd_k(const float* __restrict__ feat, const int4* __restrict__ idx, int n, int Ca, int Cb, int D,
                                                         int H, int Wd, float* __restrict__ da, float* __restrict__ db) {
  const int C = Ca + Cb;
  long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (long long)n * C) return;
  const int c = (int)(t / n), i = (int)(t % n);   // consecutive threads -> consecutive rows of one channel
  const int4 q = idx[i];
  const size_t vol = (size_t)D * H * Wd, cell = ((size_t)q.y * H + q.z) * Wd + q.w;
  const float v = feat[(size_t)i * C + c];
  if (c < Ca) da[((size_t)q.x * Ca + c) * vol + cell] = v;
  else db[((size_t)q.x * Cb + (c - Ca)) * vol + cell] = v;
}

__global__ __launch_bounds__(256) void dense_split_bwd_k(const float* __restrict__ ga, const float* __restrict__ gb, const int4* __restrict__ idx,
                                                         int n, int Ca, int Cb, int D, int H, int Wd, float* __restrict__ dfeat) {
  const int C = Ca + Cb;
  long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (long long)n * C) return;
  const int c = (int)(t / n), i = (int)(t % n);
  const int4 q = idx[i];
  const size_t vol = (size_t)D * H * Wd, cell = ((size_t)q.y * H + q.z) * Wd + q.w;
  float v = 0.f;   // a missing upstream gradient (NULL) is a zero gradient
  if (c < Ca) {
    if (ga) v = ga[((size_t)q.x * Ca + c) * vol + cell];
  } else if (gb) {
    v = gb[((size_t)q.x * Cb + (c - Ca)) * vol + cell];
  }
  dfeat[(size_t)i * C + c] = v;
}

template <typename T>
__global__ __launch_bounds__(256) void cat_pad_fwd_k(const T* __restrict__ a, int ca, const T* __restrict__ b, int cb, long long n, int cout,
                                                     T* __restrict__ out) {
  long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n * cout) return;
  const long long i = t / cout;
  const int c = (int)(t - i * cout);
  out[t] = c < ca ? a[i * ca + c] : (c < ca + cb ? b[i * cb + (c - ca)] : (T)0);   // (the zero bit pattern is 0.0 in fp32 and in bf16)
}

template <typename T>
__global__ __launch_bounds__(256) void cat_pad_bwd_k(const T* __restrict__ g, int cout, long long n, T* __restrict__ da, int ca,
                                                     T* __restrict__ db, int cb) {
  long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const int cc = ca + cb;
  if (t >= n * cc) return;
  const long long i = t / cc;
  const int c = (int)(t - i * cc);
  const T v = g[i * cout + c];
  if (c < ca) da[i * ca + c] = v;
  else db[i * cb + (c - ca)] = v;
}

template <bool BF>
__device__ __forceinline__ float ld_elem(const void* p, long long i) {
  if (BF) return __uint_as_float((unsigned)((const unsigned short*)p)[i] << 16);
  return ((const float*)p)[i];
}

// partial[block] = sum over the block's share of ka a^2 (+ kb b^2); the last block to arrive adds the partials in a fixed order
template <bool BFA, bool BFB>
__global__ __launch_bounds__(256) void sumsq2_fwd_k(const void* __restrict__ a, long long na, double ka, const void* __restrict__ b, long long nb,
                                                    double kb, double* __restrict__ partial, int32_t* __restrict__ counter, float* __restrict__ out) {
  __shared__ double s_red[4];
  __shared__ int s_last;
  double acc_a = 0.0, acc_b = 0.0;
  const long long stride = (long long)gridDim.x * blockDim.x, t0 = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  // 16-byte loads where the tensor allows (the 72 MB BEV map: scalar loads ran at 0.45 TB/s), the tail element by element
  auto sumsq = [&](const void* p, long long n, auto bf) -> double {
    constexpr bool BF = decltype(bf)::value;
    constexpr int EPV = BF ? 8 : 4;
    double acc = 0.0;
    long long done = 0;
    if ((reinterpret_cast<uintptr_t>(p) & 15) == 0) {
      const long long nv = n / EPV;
      const uint4* v4 = (const uint4*)p;
      for (long long t = t0; t < nv; t += stride) {
        const uint4 q = v4[t];
        const unsigned w[4] = {q.x, q.y, q.z, q.w};
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if (BF) {
            const float lo = __uint_as_float(w[j] << 16), hi = __uint_as_float(w[j] & 0xFFFF0000u);
            s += lo * lo + hi * hi;
          } else {
            const float f = __uint_as_float(w[j]);
            s += f * f;
          }
        }
        acc += (double)s;
      }
      done = nv * EPV;
    }
    for (long long t = done + t0; t < n; t += stride) {
      const float v = ld_elem<BF>(p, t);
      acc += (double)v * v;
    }
    return acc;
  };
  acc_a = sumsq(a, na, std::integral_constant<bool, BFA>());
  acc_b = sumsq(b, nb, std::integral_constant<bool, BFB>());
  double acc = acc_a * ka + acc_b * kb;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o, 64);
  if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    // (device-scope store + drained queue instead of a release fence: no write-back of the XCD L2 once per workgroup, see bn_fuse.h)
    btc_st_agent(partial + blockIdx.x, s_red[0] + s_red[1] + s_red[2] + s_red[3]);
    s_last = (btc_ticket_take(counter) == (int)gridDim.x - 1);
  }
  __syncthreads();
  if (!s_last) return;
  // the last workgroup: all 256 threads walk the partials (thread t: blocks t, t + 256, ... in order), then the same fixed tree as
  // above -- one thread reading up to 1024 partials one L2 round trip at a time took ~100 us, four times the pass over the data
  btc_ticket_acquire();
  double s = 0.0;
  for (int g = threadIdx.x; g < (int)gridDim.x; g += 256) s += btc_ld_agent(partial + g);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x != 0) return;
  out[0] = (float)((s_red[0] + s_red[1]) + (s_red[2] + s_red[3]));
  __hip_atomic_store(counter, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // zero on exit
}

// da = a * (g * 2 ka), db = b * (g * 2 kb): the factor is rounded to the tensor's type first, as `x * (g * (2 k)).to(x.dtype)` does
template <bool BFA, bool BFB>
__global__ __launch_bounds__(256) void sumsq2_bwd_k(const void* __restrict__ a, long long na, float ka2, void* __restrict__ da,
                                                    const void* __restrict__ b, long long nb, float kb2, void* __restrict__ db,
                                                    const float* __restrict__ g) {
  const float sa = g[0] * ka2, sb = g[0] * kb2;
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t < na) {
    if (BFA) ((unsigned short*)da)[t] = btc_f32_to_bf16(ld_elem<true>(a, t) * btc_bf16_to_f32(btc_f32_to_bf16(sa)));
    else ((float*)da)[t] = ((const float*)a)[t] * sa;
  } else if (t - na < nb) {
    const long long u = t - na;
    if (BFB) ((unsigned short*)db)[u] = btc_f32_to_bf16(ld_elem<true>(b, u) * btc_bf16_to_f32(btc_f32_to_bf16(sb)));
    else ((float*)db)[u] = ((const float*)b)[u] * sb;
  }
}

// prob[b][cell] = softmax(logit[b][:, cell])[1] * mask[b][cell] for two-class logits (B, 2, cells): torch's Softmax(dim=1), the slice of
// its last channel and the product with the loss mask (occ_head_3D.py:34-38) in one launch -- exp(x - max) / sum, as torch evaluates it
__global__ __launch_bounds__(256) void occ_prob_k(const float* __restrict__ logit, const unsigned char* __restrict__ mask, long long ncell,
                                                  long long total, float* __restrict__ out) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const long long b = i / ncell, c = i - b * ncell;
  const float z0 = logit[(b * 2 + 0) * ncell + c], z1 = logit[(b * 2 + 1) * ncell + c];
  const float m = fmaxf(z0, z1);
  const float e0 = expf(z0 - m), e1 = expf(z1 - m);
  out[i] = (e1 / (e0 + e1)) * (float)mask[i];
}

}  // namespace

extern "C" int btc_dense_split_fwd(const float* feat, const int32_t* indices, int n, int Ca, int Cb, const int32_t* h_shape, float* dense_a,
                                   float* dense_b, void* stream) {
  BTC_CHECK_ARG(Ca >= 1 && Cb >= 1 && n >= 0, "btc_dense_split_fwd: bad sizes");
  if (n <= 0) return BTC_OK;
  dense_split_fwd_k<<<btc_cdiv((long long)n * (Ca + Cb), 256), 256, 0, (hipStream_t)stream>>>(feat, (const int4*)indices, n, Ca, Cb, h_shape[0],
                                                                                           h_shape[1], h_shape[2], dense_a, dense_b);
  BTC_LAUNCH_CHECK();
  return BTC_OK;
}

extern "C" int btc_dense_split_bwd(const float* grad_a, const float* grad_b, const int32_t* indices, int n, int Ca, int Cb, const int32_t* h_shape,
                                   float* dfeat, void* stream) {
  BTC_CHECK_ARG(Ca >= 1 && Cb >= 1 && n >= 0, "btc_dense_split_bwd: bad sizes");
  if (n <= 0) return BTC_OK;
  dense_split_bwd_k<<<btc_cdiv((long long)n * (Ca + Cb), 256), 256, 0, (hipStream_t)stream>>>(grad_a, grad_b, (const int4*)indices, n, Ca, Cb,
                                                                                           h_shape[0], h_shape[1], h_shape[2], dfeat);
  BTC_LAUNCH_CHECK();
  return BTC_OK;
}

extern "C" int btc_cat_pad_fwd(const void* a, int ca, const void* b, int cb, long long n, int cout, int elem_bytes, void* out, void* stream) {
  BTC_CHECK_ARG(ca >= 1 && cb >= 0 && cout >= ca + cb && n >= 0 && (elem_bytes == 4 || elem_bytes == 2), "btc_cat_pad_fwd: bad sizes");
  if (n <= 0) return BTC_OK;
  if (elem_bytes == 4)
    cat_pad_fwd_k<unsigned><<<btc_cdiv(n * cout, 256), 256, 0, (hipStream_t)stream>>>((const unsigned*)a, ca, (const unsigned*)b, cb, n, cout, (unsigned*)out);
  else
    cat_pad_fwd_k<unsigned short><<<btc_cdiv(n * cout, 256), 256, 0, (hipStream_t)stream>>>((const unsigned short*)a, ca, (const unsigned short*)b, cb, n,
                                                                                        cout, (unsigned short*)out);
  BTC_LAUNCH_CHECK();
  return BTC_OK;
}

extern "C" int btc_cat_pad_bwd(const void* grad, int cout, long long n, int elem_bytes, void* da, int ca, void* db, int cb, void* stream) {
  BTC_CHECK_ARG(ca >= 1 && cb >= 0 && cout >= ca + cb && n >= 0 && (elem_bytes == 4 || elem_bytes == 2), "btc_cat_pad_bwd: bad sizes");
  if (n <= 0) return BTC_OK;
  if (elem_bytes == 4)
    cat_pad_bwd_k<unsigned><<<btc_cdiv(n * (ca + cb), 256), 256, 0, (hipStream_t)stream>>>((const unsigned*)grad, cout, n, (unsigned*)da, ca, (unsigned*)db, cb);
  else
    cat_pad_bwd_k<unsigned short><<<btc_cdiv(n * (ca + cb), 256), 256, 0, (hipStream_t)stream>>>((const unsigned short*)grad, cout, n, (unsigned short*)da,
                                                                                             ca, (unsigned short*)db, cb);
  BTC_LAUNCH_CHECK();
  return BTC_OK;
}

extern "C" size_t btc_sumsq2_ws_bytes(void) { return 256 + 1024 * sizeof(double); }

// out[0] = ka sum a^2 + kb sum b^2 (b may be NULL with nb = 0); *_bf16: the tensor is bfloat16
extern "C" int btc_sumsq2_fwd(const void* a, long long na, int a_bf16, double ka, const void* b, long long nb, int b_bf16, double kb, float* out,
                              void* ws, size_t ws_bytes, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  BTC_CHECK_ARG(ws_bytes >= btc_sumsq2_ws_bytes(), "btc_sumsq2_fwd: workspace too small");
  BTC_CHECK_ARG(na >= 0 && nb >= 0 && (nb == 0 || b), "btc_sumsq2_fwd: bad sizes");
  int32_t* counter = (int32_t*)ws;   // first 256 bytes: zero on entry / exit
  double* partial = (double*)((char*)ws + 256);
  const long long m = na > nb ? na : nb;
  int grid = btc_cdiv(m > 0 ? m : 1, 256 * 16);   // (<= 1024 partials: the workspace)
  if (grid > 1024) grid = 1024;
  if (grid < 1) grid = 1;
  if (a_bf16 && b_bf16) sumsq2_fwd_k<true, true><<<grid, 256, 0, stream>>>(a, na, ka, b, nb, kb, partial, counter, out);
  else if (a_bf16) sumsq2_fwd_k<true, false><<<grid, 256, 0, stream>>>(a, na, ka, b, nb, kb, partial, counter, out);
  else if (b_bf16) sumsq2_fwd_k<false, true><<<grid, 256, 0, stream>>>(a, na, ka, b, nb, kb, partial, counter, out);
  else sumsq2_fwd_k<false, false><<<grid, 256, 0, stream>>>(a, na, ka, b, nb, kb, partial, counter, out);
  BTC_LAUNCH_CHECK();
  return BTC_OK;
}

// da = a * (g[0] * ka2), db = b * (g[0] * kb2)   (ka2 = 2 ka: the caller's constant)
extern "C" int btc_sumsq2_bwd(const void* a, long long na, int a_bf16, float ka2, void* da, const void* b, long long nb, int b_bf16, float kb2,
                              void* db, const float* g, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  BTC_CHECK_ARG(na >= 0 && nb >= 0 && g, "btc_sumsq2_bwd: bad arguments");
  if (na + nb <= 0) return BTC_OK;
  const int grid = btc_cdiv(na + nb, 256);
  if (a_bf16 && b_bf16) sumsq2_bwd_k<true, true><<<grid, 256, 0, stream>>>(a, na, ka2, da, b, nb, kb2, db, g);
  else if (a_bf16) sumsq2_bwd_k<true, false><<<grid, 256, 0, stream>>>(a, na, ka2, da, b, nb, kb2, db, g);
  else if (b_bf16) sumsq2_bwd_k<false, true><<<grid, 256, 0, stream>>>(a, na, ka2, da, b, nb, kb2, db, g);
  else sumsq2_bwd_k<false, false><<<grid, 256, 0, stream>>>(a, na, ka2, da, b, nb, kb2, db, g);
  BTC_LAUNCH_CHECK();
  return BTC_OK;
}

// one wave that does nothing for `microseconds` (constant-rate 100 MHz counter): the probe of btcdet_amd/streams.py -- work enqueued on ANOTHER
// stream finishes behind it exactly when the two streams were dealt the same hardware queue
namespace {
__global__ void spin_k(long long ticks) {
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
}
}  // namespace

extern "C" int btc_spin(int microseconds, void* stream) {
  BTC_CHECK_ARG(microseconds >= 0 && microseconds <= 100000, "btc_spin: 0 .. 100000 us");
  spin_k<<<1, 64, 0, (hipStream_t)stream>>>(100LL * microseconds);
  BTC_LAUNCH_CHECK();
  return BTC_OK;
}

extern "C" int btc_occ_prob(const float* logit, const unsigned char* mask, int B, long long ncell, float* prob, void* stream) {
  BTC_CHECK_ARG(B >= 0 && ncell >= 0, "btc_occ_prob: bad sizes");
  const long long total = (long long)B * ncell;
  if (total > 0) occ_prob_k<<<btc_cdiv(total, 256), 256, 0, (hipStream_t)stream>>>(logit, mask, ncell, total, prob);
  BTC_LAUNCH_CHECK();
  return BTC_OK;
}
