// Fused BatchNorm1d (+ReLU) over sparse-tensor features (N rows x C channels, fp32) for gfx950.
//
// The reference applies nn.BatchNorm1d(eps=1e-3, momentum=0.01) and nn.ReLU to `.features` after every sparse conv
// (spconv.SparseSequential, /root/reference/btcdet/models/backbones_3d/spconv_backbone.py:33-43,96,634): on ROCm that is
// ~10 small launches per layer (statistics, running-stat updates, transform, clamp, and their backward twins), all
// latency bound at BtcDet's sizes.  Here: 2 launches forward, 2 backward.
//   bn_stats / bn_bwd_stats : per-workgroup partial channel sums (fp64 accumulation), and the LAST workgroup to
//                             arrive (agent-scope release / acquire around one atomic ticket, cdna guide G16) reduces
//                             the partials in index order (deterministic), updates the running statistics and
//                             num_batches_tracked, and publishes mean / rstd (or dgamma / dbeta)
//   bn_apply / bn_bwd_apply : elementwise, float4
#include "btc_common.h"
#include "bn_fuse.h"

namespace {

constexpr int BN_T = 256;

struct BnShape {
  int N, C, cpow, rpi;  // cpow = pow2 >= min(C,256); rpi = rows per iteration of a workgroup
};

// a workgroup's partial sums are published with device-scope (sc1, write-through) stores: they are at the coherence point once the
// storing wave's vector-memory queue has drained -- no agent-scope release fence, i.e. no write-back of the XCD L2's dirty lines (all
// the activations the previous kernels wrote) once per workgroup (MI355X_MICROARCH.md, handoff-flag: `sc1` payload -> vmcnt(0) -> flag)
__device__ __forceinline__ void st_partial(double* p, double v) { btc_st_agent(p, v); }

// last-arriver election (placement independent: sc1 partials, drained -> ticket -> acquire)
__device__ __forceinline__ bool last_block(int32_t* counter) {
  __shared__ int s_last;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // every wave's partials have left its queue before the workgroup's ticket (btc_common.h)
  __syncthreads();
  if (threadIdx.x == 0) s_last = (btc_ticket_take(counter) == (int)gridDim.x - 1);
  __syncthreads();
  if (!s_last) return false;
  if (threadIdx.x == 0) btc_ticket_acquire();
  __syncthreads();
  return true;
}

// the last workgroup's sum over all workgroups' partial pairs, in a fixed order: thread-row rr takes workgroups rr, rr + rpi, ...
// with 8 pairs in flight (the walk is a chain of L2 round trips: at 4 pairs in flight and 512 workgroups it took longer than the
// pass over the data -- 22 of the 45 us of a 210 K x 32 backward-statistics launch)
__device__ __forceinline__ void reduce_partials(const double* __restrict__ partial, int G, int C, int rpi, int c, int rr, double& a, double& b) {
  int g = rr;
  for (; g + 7 * rpi < G; g += 8 * rpi) {
    double av[8], bv[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      av[u] = btc_ld_agent(&partial[((size_t)(g + u * rpi) * 2 + 0) * C + c]);
      bv[u] = btc_ld_agent(&partial[((size_t)(g + u * rpi) * 2 + 1) * C + c]);
    }
    a += ((av[0] + av[1]) + (av[2] + av[3])) + ((av[4] + av[5]) + (av[6] + av[7]));
    b += ((bv[0] + bv[1]) + (bv[2] + bv[3])) + ((bv[4] + bv[5]) + (bv[6] + bv[7]));
  }
  for (; g < G; g += rpi) {
    a += btc_ld_agent(&partial[((size_t)g * 2 + 0) * C + c]);
    b += btc_ld_agent(&partial[((size_t)g * 2 + 1) * C + c]);
  }
}

// partial[blk][0][c] = sum_x, partial[blk][1][c] = sum_x^2 over the block's rows
// VEC: C % 4 == 0 -- a thread owns 4 adjacent channels and moves one float4 (bf16: 8 bytes) per row: a quarter of the load
// instructions of the scalar walk for the same bytes (SV = the shape over C / 4 channel groups)
template <bool BF, bool VEC>
__global__ __launch_bounds__(BN_T) void bn_stats(const float* __restrict__ x, BnShape S, BnShape SV, double* __restrict__ partial,
                                                 int32_t* __restrict__ counter, float momentum, float eps, int training,
                                                 const float* __restrict__ running_mean_in, float* __restrict__ running_mean,
                                                 float* __restrict__ running_var, long long* __restrict__ num_batches,
                                                 float* __restrict__ mean_out, float* __restrict__ rstd_out) {
  __shared__ double s_a[BN_T * (VEC ? 4 : 1)], s_b[BN_T * (VEC ? 4 : 1)];
  const int tid = threadIdx.x;
  if (VEC) {
    const int CV = S.C >> 2;
    for (int cv0 = 0; cv0 < CV; cv0 += SV.cpow) {
      const int cv = cv0 + (tid % SV.cpow), rr = tid / SV.cpow;
      double a[4] = {0.0, 0.0, 0.0, 0.0}, b[4] = {0.0, 0.0, 0.0, 0.0};
      if (cv < CV) {
        const long long stride = (long long)gridDim.x * SV.rpi;
        long long r = (long long)blockIdx.x * SV.rpi + rr;
        for (; r + 3 * stride < S.N; r += 4 * stride) {  // 4 independent 16-byte loads in flight
          const float4 v0 = btc_ld4<BF>(x, r * S.C + cv * 4), v1 = btc_ld4<BF>(x, (r + stride) * S.C + cv * 4),
                       v2 = btc_ld4<BF>(x, (r + 2 * stride) * S.C + cv * 4), v3 = btc_ld4<BF>(x, (r + 3 * stride) * S.C + cv * 4);
          const float e0[4] = {v0.x, v0.y, v0.z, v0.w}, e1[4] = {v1.x, v1.y, v1.z, v1.w}, e2[4] = {v2.x, v2.y, v2.z, v2.w},
                      e3[4] = {v3.x, v3.y, v3.z, v3.w};
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            a[j] += (double)e0[j] + (double)e1[j] + (double)e2[j] + (double)e3[j];
            b[j] += (double)e0[j] * e0[j] + (double)e1[j] * e1[j] + (double)e2[j] * e2[j] + (double)e3[j] * e3[j];
          }
        }
        for (; r < S.N; r += stride) {
          const float4 v = btc_ld4<BF>(x, r * S.C + cv * 4);
          const float e[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
          for (int j = 0; j < 4; ++j) { a[j] += e[j]; b[j] += (double)e[j] * e[j]; }
        }
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) { s_a[tid * 4 + j] = a[j]; s_b[tid * 4 + j] = b[j]; }
      __syncthreads();
      if (rr == 0 && cv < CV) {
        for (int q = 1; q < SV.rpi; ++q)
#pragma unroll
          for (int j = 0; j < 4; ++j) { a[j] += s_a[(tid + q * SV.cpow) * 4 + j]; b[j] += s_b[(tid + q * SV.cpow) * 4 + j]; }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          st_partial(&partial[((size_t)blockIdx.x * 2 + 0) * S.C + cv * 4 + j], a[j]);
          st_partial(&partial[((size_t)blockIdx.x * 2 + 1) * S.C + cv * 4 + j], b[j]);
        }
      }
      __syncthreads();
    }
  } else
  for (int c0 = 0; c0 < S.C; c0 += S.cpow) {
    const int c = c0 + (tid % S.cpow), rr = tid / S.cpow;
    double a = 0.0, b = 0.0;
    if (c < S.C) {
      const long long stride = (long long)gridDim.x * S.rpi;
      long long r = (long long)blockIdx.x * S.rpi + rr;
      for (; r + 3 * stride < S.N; r += 4 * stride) {  // 4 independent loads in flight
        float v0 = btc_ld1<BF>(x, r * S.C + c), v1 = btc_ld1<BF>(x, (r + stride) * S.C + c), v2 = btc_ld1<BF>(x, (r + 2 * stride) * S.C + c),
              v3 = btc_ld1<BF>(x, (r + 3 * stride) * S.C + c);
        a += (double)v0 + (double)v1 + (double)v2 + (double)v3;
        b += (double)v0 * v0 + (double)v1 * v1 + (double)v2 * v2 + (double)v3 * v3;
      }
      for (; r < S.N; r += stride) {
        float v = btc_ld1<BF>(x, r * S.C + c);
        a += v;
        b += (double)v * v;
      }
    }
    s_a[tid] = a;
    s_b[tid] = b;
    __syncthreads();
    if (rr == 0 && c < S.C) {
      for (int q = 1; q < S.rpi; ++q) {
        a += s_a[tid + q * S.cpow];
        b += s_b[tid + q * S.cpow];
      }
      st_partial(&partial[((size_t)blockIdx.x * 2 + 0) * S.C + c], a);
      st_partial(&partial[((size_t)blockIdx.x * 2 + 1) * S.C + c], b);
    }
    __syncthreads();
  }
  if (!last_block(counter)) return;
  // partials reduced in a fixed order: thread-row rr takes workgroups rr, rr + rpi, ...; rows combined 0..rpi-1
  for (int c0 = 0; c0 < S.C; c0 += S.cpow) {
    const int c = c0 + (tid % S.cpow), rr = tid / S.cpow;
    double a = 0.0, b = 0.0;
    if (c < S.C) reduce_partials(partial, (int)gridDim.x, S.C, S.rpi, c, rr, a, b);
    s_a[tid] = a;
    s_b[tid] = b;
    __syncthreads();
    if (rr == 0 && c < S.C) {
      for (int q = 1; q < S.rpi; ++q) {
        a += s_a[tid + q * S.cpow];
        b += s_b[tid + q * S.cpow];
      }
      double mean = a / S.N;
      double var = b / S.N - mean * mean;  // biased, as F.batch_norm normalises with
      if (var < 0.0) var = 0.0;
      mean_out[c] = (float)mean;
      rstd_out[c] = (float)(1.0 / sqrt(var + (double)eps));
      if (running_mean) {
        double unb = S.N > 1 ? var * ((double)S.N / (double)(S.N - 1)) : var;
        running_mean[c] = (float)((1.0 - momentum) * running_mean[c] + momentum * mean);
        running_var[c] = (float)((1.0 - momentum) * running_var[c] + momentum * unb);
      }
    }
    __syncthreads();
  }
  if (tid == 0) {
    if (num_batches) *num_batches += 1;
    __hip_atomic_store(counter, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // ready for the next call on this stream
  }
}

__global__ __launch_bounds__(BN_T) void bn_eval_stats(const float* __restrict__ running_mean, const float* __restrict__ running_var,
                                                      int C, float eps, float* __restrict__ mean_out, float* __restrict__ rstd_out) {
  int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  mean_out[c] = running_mean[c];
  rstd_out[c] = 1.0f / sqrtf(running_var[c] + eps);
}

template <bool VEC, bool BF>
__global__ __launch_bounds__(BN_T) void bn_apply(const float* __restrict__ x, const float* __restrict__ mean, const float* __restrict__ rstd,
                                                 const float* __restrict__ gamma, const float* __restrict__ beta, long long total, int C,
                                                 int relu, float* __restrict__ y) {
  long long i = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * (VEC ? 4 : 1);
  if (i >= total) return;
  if (VEC) {
    int c = (int)(i % C);
    float4 v = btc_ld4<BF>(x, i);
    float o[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float g = gamma ? gamma[c + j] : 1.f, b = beta ? beta[c + j] : 0.f;
      float t = (o[j] - mean[c + j]) * rstd[c + j] * g + b;
      o[j] = relu ? fmaxf(t, 0.f) : t;
    }
    btc_st4<BF>(y, i, o[0], o[1], o[2], o[3]);
  } else {
    int c = (int)(i % C);
    float g = gamma ? gamma[c] : 1.f, b = beta ? beta[c] : 0.f;
    float t = (btc_ld1<BF>(x, i) - mean[c]) * rstd[c] * g + b;
    btc_st1<BF>(y, i, relu ? fmaxf(t, 0.f) : t);
  }
}

// partial sums of g = dy * (y > 0) and g * xhat; the last block turns them into dbeta / dgamma
template <bool BF, bool VEC>
__global__ __launch_bounds__(BN_T) void bn_bwd_stats(const float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ dy,
                                                     const float* __restrict__ mean, const float* __restrict__ rstd, BnShape S, BnShape SV, int relu,
                                                     double* __restrict__ partial, int32_t* __restrict__ counter,
                                                     float* __restrict__ dgamma, float* __restrict__ dbeta) {
  __shared__ double s_a[BN_T * (VEC ? 4 : 1)], s_b[BN_T * (VEC ? 4 : 1)];
  const int tid = threadIdx.x;
  if (VEC) {
    const int CV = S.C >> 2;
    for (int cv0 = 0; cv0 < CV; cv0 += SV.cpow) {
      const int cv = cv0 + (tid % SV.cpow), rr = tid / SV.cpow;
      double a[4] = {0.0, 0.0, 0.0, 0.0}, b[4] = {0.0, 0.0, 0.0, 0.0};
      if (cv < CV) {
        float m[4], rs[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) { m[j] = mean[cv * 4 + j]; rs[j] = rstd[cv * 4 + j]; }
        const long long stride = (long long)gridDim.x * SV.rpi;
        long long r = (long long)blockIdx.x * SV.rpi + rr;
        for (; r + stride < S.N; r += 2 * stride) {  // 6 independent 16-byte loads in flight
          const long long i0 = r * S.C + cv * 4, i1 = (r + stride) * S.C + cv * 4;
          const float4 g0 = btc_ld4<BF>(dy, i0), y0 = btc_ld4<BF>(y, i0), x0 = btc_ld4<BF>(x, i0);
          const float4 g1 = btc_ld4<BF>(dy, i1), y1 = btc_ld4<BF>(y, i1), x1 = btc_ld4<BF>(x, i1);
          const float gg[2][4] = {{g0.x, g0.y, g0.z, g0.w}, {g1.x, g1.y, g1.z, g1.w}};
          const float yy[2][4] = {{y0.x, y0.y, y0.z, y0.w}, {y1.x, y1.y, y1.z, y1.w}};
          const float xx[2][4] = {{x0.x, x0.y, x0.z, x0.w}, {x1.x, x1.y, x1.z, x1.w}};
#pragma unroll
          for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const float g = (relu && !(yy[u][j] > 0.f)) ? 0.f : gg[u][j];
              a[j] += (double)g;
              b[j] += (double)g * ((xx[u][j] - m[j]) * rs[j]);
            }
        }
        for (; r < S.N; r += stride) {
          const long long i0 = r * S.C + cv * 4;
          const float4 g0 = btc_ld4<BF>(dy, i0), y0 = btc_ld4<BF>(y, i0), x0 = btc_ld4<BF>(x, i0);
          const float gg[4] = {g0.x, g0.y, g0.z, g0.w}, yy[4] = {y0.x, y0.y, y0.z, y0.w}, xx[4] = {x0.x, x0.y, x0.z, x0.w};
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const float g = (relu && !(yy[j] > 0.f)) ? 0.f : gg[j];
            a[j] += (double)g;
            b[j] += (double)g * ((xx[j] - m[j]) * rs[j]);
          }
        }
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) { s_a[tid * 4 + j] = a[j]; s_b[tid * 4 + j] = b[j]; }
      __syncthreads();
      if (rr == 0 && cv < CV) {
        for (int q = 1; q < SV.rpi; ++q)
#pragma unroll
          for (int j = 0; j < 4; ++j) { a[j] += s_a[(tid + q * SV.cpow) * 4 + j]; b[j] += s_b[(tid + q * SV.cpow) * 4 + j]; }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          st_partial(&partial[((size_t)blockIdx.x * 2 + 0) * S.C + cv * 4 + j], a[j]);
          st_partial(&partial[((size_t)blockIdx.x * 2 + 1) * S.C + cv * 4 + j], b[j]);
        }
      }
      __syncthreads();
    }
  } else
  for (int c0 = 0; c0 < S.C; c0 += S.cpow) {
    const int c = c0 + (tid % S.cpow), rr = tid / S.cpow;
    double a = 0.0, b = 0.0;
    if (c < S.C) {
      const float m = mean[c], rs = rstd[c];
      const long long stride = (long long)gridDim.x * S.rpi;
      long long r = (long long)blockIdx.x * S.rpi + rr;
      for (; r + 3 * stride < S.N; r += 4 * stride) {  // 12 independent loads in flight (the kernel is latency bound)
        float g[4], yv[4], xv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const long long i = (r + u * stride) * S.C + c;
          g[u] = btc_ld1<BF>(dy, i);
          yv[u] = btc_ld1<BF>(y, i);
          xv[u] = btc_ld1<BF>(x, i);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          if (relu && !(yv[u] > 0.f)) g[u] = 0.f;
          a += (double)g[u];
          b += (double)g[u] * ((xv[u] - m) * rs);
        }
      }
      for (; r < S.N; r += stride) {
        float g = btc_ld1<BF>(dy, r * S.C + c);
        if (relu && !(btc_ld1<BF>(y, r * S.C + c) > 0.f)) g = 0.f;
        a += g;
        b += (double)g * ((btc_ld1<BF>(x, r * S.C + c) - m) * rs);
      }
    }
    s_a[tid] = a;
    s_b[tid] = b;
    __syncthreads();
    if (rr == 0 && c < S.C) {
      for (int q = 1; q < S.rpi; ++q) {
        a += s_a[tid + q * S.cpow];
        b += s_b[tid + q * S.cpow];
      }
      st_partial(&partial[((size_t)blockIdx.x * 2 + 0) * S.C + c], a);
      st_partial(&partial[((size_t)blockIdx.x * 2 + 1) * S.C + c], b);
    }
    __syncthreads();
  }
  if (!last_block(counter)) return;
  for (int c0 = 0; c0 < S.C; c0 += S.cpow) {
    const int c = c0 + (tid % S.cpow), rr = tid / S.cpow;
    double a = 0.0, b = 0.0;
    if (c < S.C) reduce_partials(partial, (int)gridDim.x, S.C, S.rpi, c, rr, a, b);
    s_a[tid] = a;
    s_b[tid] = b;
    __syncthreads();
    if (rr == 0 && c < S.C) {
      for (int q = 1; q < S.rpi; ++q) {
        a += s_a[tid + q * S.cpow];
        b += s_b[tid + q * S.cpow];
      }
      dbeta[c] = (float)a;
      dgamma[c] = (float)b;
    }
    __syncthreads();
  }
  if (tid == 0) __hip_atomic_store(counter, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <bool VEC, bool BF>
__global__ __launch_bounds__(BN_T) void bn_bwd_apply(const float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ dy,
                                                     const float* __restrict__ mean, const float* __restrict__ rstd,
                                                     const float* __restrict__ gamma, const float* __restrict__ dgamma,
                                                     const float* __restrict__ dbeta, long long total, int C, int N, int relu, int training,
                                                     float* __restrict__ dx) {
  long long i = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * (VEC ? 4 : 1);
  if (i >= total) return;
  const float invn = 1.0f / (float)N;
  constexpr int W = VEC ? 4 : 1;
  float xv[W], yv[W], gv[W];
  if (VEC) {
    float4 a = btc_ld4<BF>(x, i), b = btc_ld4<BF>(y, i), c4 = btc_ld4<BF>(dy, i);
    xv[0] = a.x; xv[W > 1 ? 1 : 0] = a.y; xv[W > 2 ? 2 : 0] = a.z; xv[W > 3 ? 3 : 0] = a.w;
    yv[0] = b.x; yv[W > 1 ? 1 : 0] = b.y; yv[W > 2 ? 2 : 0] = b.z; yv[W > 3 ? 3 : 0] = b.w;
    gv[0] = c4.x; gv[W > 1 ? 1 : 0] = c4.y; gv[W > 2 ? 2 : 0] = c4.z; gv[W > 3 ? 3 : 0] = c4.w;
  } else {
    xv[0] = btc_ld1<BF>(x, i); yv[0] = btc_ld1<BF>(y, i); gv[0] = btc_ld1<BF>(dy, i);
  }
  const int c = (int)(i % C);
  float o[W];
#pragma unroll
  for (int j = 0; j < W; ++j) {
    float g = gv[j];
    if (relu && !(yv[j] > 0.f)) g = 0.f;
    const float gm = gamma ? gamma[c + j] : 1.f, rs = rstd[c + j];
    if (training) {
      float xh = (xv[j] - mean[c + j]) * rs;
      o[j] = gm * rs * (g - dbeta[c + j] * invn - xh * dgamma[c + j] * invn);
    } else {
      o[j] = gm * rs * g;
    }
  }
  if (VEC) btc_st4<BF>(dx, i, o[0], o[W > 1 ? 1 : 0], o[W > 2 ? 2 : 0], o[W > 3 ? 3 : 0]);
  else btc_st1<BF>(dx, i, o[0]);
}

// column sums of an (N, C) matrix (the bias gradient of a sparse conv): same two-level, fixed-order scheme as bn_bwd_stats
template <bool BF>
__global__ __launch_bounds__(BN_T) void col_sum(const float* __restrict__ x, BnShape S, double* __restrict__ partial,
                                                int32_t* __restrict__ counter, float* __restrict__ out) {
  __shared__ double s_a[BN_T];
  const int tid = threadIdx.x;
  for (int c0 = 0; c0 < S.C; c0 += S.cpow) {
    const int c = c0 + (tid % S.cpow), rr = tid / S.cpow;
    double a = 0.0;
    if (c < S.C) {
      const long long stride = (long long)gridDim.x * S.rpi;
      long long r = (long long)blockIdx.x * S.rpi + rr;
      for (; r + 3 * stride < S.N; r += 4 * stride) {
        float v0 = btc_ld1<BF>(x, r * S.C + c), v1 = btc_ld1<BF>(x, (r + stride) * S.C + c), v2 = btc_ld1<BF>(x, (r + 2 * stride) * S.C + c),
              v3 = btc_ld1<BF>(x, (r + 3 * stride) * S.C + c);
        a += ((double)v0 + (double)v1) + ((double)v2 + (double)v3);
      }
      for (; r < S.N; r += stride) a += (double)btc_ld1<BF>(x, r * S.C + c);
    }
    s_a[tid] = a;
    __syncthreads();
    if (rr == 0 && c < S.C) {
      for (int q = 1; q < S.rpi; ++q) a += s_a[tid + q * S.cpow];
      st_partial(&partial[(size_t)blockIdx.x * S.C + c], a);
    }
    __syncthreads();
  }
  if (!last_block(counter)) return;
  for (int c0 = 0; c0 < S.C; c0 += S.cpow) {
    const int c = c0 + (tid % S.cpow), rr = tid / S.cpow;
    double a = 0.0;
    if (c < S.C)
      for (int g = rr; g < (int)gridDim.x; g += S.rpi) a += btc_ld_agent(&partial[(size_t)g * S.C + c]);
    s_a[tid] = a;
    __syncthreads();
    if (rr == 0 && c < S.C) {
      for (int q = 1; q < S.rpi; ++q) a += s_a[tid + q * S.cpow];
      out[c] = (float)a;
    }
    __syncthreads();
  }
  if (tid == 0) __hip_atomic_store(counter, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

BnShape bn_shape(int N, int C) {
  BnShape S;
  S.N = N;
  S.C = C;
  int cp = 1;
  while (cp < C && cp < BN_T) cp <<= 1;
  S.cpow = cp;
  S.rpi = BN_T / cp;
  return S;
}

// bwd: three input streams per row and no dependent second pass behind it -> more, thinner workgroups
int bn_grid_bwd(int N, const BnShape& S) {
  int g = btc_cdiv(N, S.rpi * 16);
  if (g > 512) g = 512;
  if (g < 1) g = 1;
  return g;
}

int bn_grid(int N, const BnShape& S) {
  int g = btc_cdiv(N, S.rpi * 64);  // >= 64 rows per thread-row before adding another workgroup: the last-arriver
  if (g > 256) g = 256;            // reduction walks all partials, so few, fat workgroups win at BtcDet's sizes
  if (g < 1) g = 1;
  return g;
}

// vectorised statistics passes: one workgroup per `kb` KB of input (tuning keys override), at most one per CU.  Few, fat
// workgroups: the last arriver walks every workgroup's partial sums, and that walk -- not the pass over the data -- bounds
// these launches at BtcDet's sizes (tools/bn_bench.py)
int bn_grid_bytes(long long bytes, int tune_key, int kb_default) {
  const int t = btc_tune_get(tune_key);
  const long long per = 1024LL * (t > 0 ? t : kb_default);
  long long g = (bytes + per - 1) / per;
  return (int)(g > 256 ? 256 : (g < 1 ? 1 : g));
}

}  // namespace

extern "C" size_t btc_bn_ws_bytes(int C) { return 256 + btc_align((size_t)512 * 2 * C * sizeof(double)); }

template <bool BF>
static int bn_fwd_impl(const float* x, int N, int C, const float* gamma, const float* beta, float* running_mean,
                               float* running_var, long long* num_batches_tracked, float momentum, float eps, int training, int relu,
                               float* y, float* save_mean, float* save_rstd, void* ws, size_t ws_bytes, void* stream_, bool have_stats = false) {
  // have_stats: save_mean / save_rstd (and the running statistics) were produced by the conv kernel's epilogue (bn_fuse.h): apply only
  hipStream_t stream = (hipStream_t)stream_;
  BTC_CHECK_ARG(N >= 1 && C >= 1, "btc_bn_relu_fwd: empty input");
  BTC_CHECK_ARG(ws_bytes >= btc_bn_ws_bytes(C), "btc_bn_relu_fwd: workspace too small");
  BTC_CHECK_ARG(training || (running_mean && running_var), "btc_bn_relu_fwd: eval mode needs running statistics");
  int32_t* counter = (int32_t*)ws;  // first 256 bytes: arrival counter, zero on entry, zero on exit
  double* partial = (double*)((char*)ws + 256);
  BnShape S = bn_shape(N, C);
  if (have_stats) {
  } else if (training) {
    if ((C & 3) == 0) {
      const BnShape SV = bn_shape(N, C >> 2);
      const int g = bn_grid_bytes((long long)N * C * (BF ? 2 : 4), BTC_TUNE_BN_FWD_KB, 64);
      bn_stats<BF, true><<<g, BN_T, 0, stream>>>(x, S, SV, partial, counter, momentum, eps, training, running_mean, running_mean, running_var,
                                                 num_batches_tracked, save_mean, save_rstd);
    } else {
      bn_stats<BF, false><<<bn_grid(N, S), BN_T, 0, stream>>>(x, S, S, partial, counter, momentum, eps, training, running_mean, running_mean,
                                                          running_var, num_batches_tracked, save_mean, save_rstd);
    }
  } else {
    bn_eval_stats<<<btc_cdiv(C, BN_T), BN_T, 0, stream>>>(running_mean, running_var, C, eps, save_mean, save_rstd);
  }
  BTC_LAUNCH_CHECK();
  long long total = (long long)N * C;
  if ((C & 3) == 0) bn_apply<true, BF><<<btc_cdiv(total / 4, BN_T), BN_T, 0, stream>>>(x, save_mean, save_rstd, gamma, beta, total, C, relu, y);
  else bn_apply<false, BF><<<btc_cdiv(total, BN_T), BN_T, 0, stream>>>(x, save_mean, save_rstd, gamma, beta, total, C, relu, y);
  BTC_LAUNCH_CHECK();
  return BTC_OK;
}

template <bool BF>
static int bn_bwd_impl(const float* x, const float* y, const float* dy, int N, int C, const float* gamma, const float* save_mean,
                               const float* save_rstd, int training, int relu, float* dx, float* dgamma, float* dbeta, void* ws,
                               size_t ws_bytes, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  BTC_CHECK_ARG(N >= 1 && C >= 1, "btc_bn_relu_bwd: empty input");
  BTC_CHECK_ARG(ws_bytes >= btc_bn_ws_bytes(C), "btc_bn_relu_bwd: workspace too small");
  int32_t* counter = (int32_t*)ws;
  double* partial = (double*)((char*)ws + 256);
  BnShape S = bn_shape(N, C);
  if ((C & 3) == 0) {
    const BnShape SV = bn_shape(N, C >> 2);
    const int g = bn_grid_bytes((long long)N * C * (BF ? 6 : 12), BTC_TUNE_BN_BWD_KB, 128);
    bn_bwd_stats<BF, true><<<g, BN_T, 0, stream>>>(x, y, dy, save_mean, save_rstd, S, SV, relu, partial, counter, dgamma, dbeta);
  } else {
    bn_bwd_stats<BF, false><<<bn_grid_bwd(N, S), BN_T, 0, stream>>>(x, y, dy, save_mean, save_rstd, S, S, relu, partial, counter, dgamma, dbeta);
  }
  BTC_LAUNCH_CHECK();
  long long total = (long long)N * C;
  if ((C & 3) == 0)
    bn_bwd_apply<true, BF><<<btc_cdiv(total / 4, BN_T), BN_T, 0, stream>>>(x, y, dy, save_mean, save_rstd, gamma, dgamma, dbeta, total, C, N, relu,
                                                                      training, dx);
  else
    bn_bwd_apply<false, BF><<<btc_cdiv(total, BN_T), BN_T, 0, stream>>>(x, y, dy, save_mean, save_rstd, gamma, dgamma, dbeta, total, C, N, relu,
                                                                   training, dx);
  BTC_LAUNCH_CHECK();
  return BTC_OK;
}

// out[c] = sum over rows of x[r][c] (fp64 accumulation, deterministic).  ws as for btc_bn_relu_fwd.
template <bool BF>
static int col_sum_impl(const float* x, int N, int C, float* out, void* ws, size_t ws_bytes, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  BTC_CHECK_ARG(N >= 1 && C >= 1, "btc_col_sum: empty input");
  BTC_CHECK_ARG(ws_bytes >= btc_bn_ws_bytes(C), "btc_col_sum: workspace too small");
  BnShape S = bn_shape(N, C);
  col_sum<BF><<<bn_grid_bwd(N, S), BN_T, 0, stream>>>(x, S, (double*)((char*)ws + 256), (int32_t*)ws, out);
  BTC_LAUNCH_CHECK();
  return BTC_OK;
}

extern "C" int btc_col_sum(const float* x, int N, int C, float* out, void* ws, size_t ws_bytes, void* stream) {
  return col_sum_impl<false>(x, N, C, out, ws, ws_bytes, stream);
}

extern "C" int btc_col_sum_bf16(const void* x, int N, int C, float* out, void* ws, size_t ws_bytes, void* stream) {
  return col_sum_impl<true>((const float*)x, N, C, out, ws, ws_bytes, stream);
}

extern "C" int btc_bn_relu_fwd(const float* x, int N, int C, const float* gamma, const float* beta, float* running_mean,
                               float* running_var, long long* num_batches_tracked, float momentum, float eps, int training, int relu,
                               float* y, float* save_mean, float* save_rstd, void* ws, size_t ws_bytes, void* stream) {
  return bn_fwd_impl<false>(x, N, C, gamma, beta, running_mean, running_var, num_batches_tracked, momentum, eps, training, relu, y, save_mean,
                            save_rstd, ws, ws_bytes, stream);
}

extern "C" int btc_bn_relu_bwd(const float* x, const float* y, const float* dy, int N, int C, const float* gamma, const float* save_mean,
                               const float* save_rstd, int training, int relu, float* dx, float* dgamma, float* dbeta, void* ws,
                               size_t ws_bytes, void* stream) {
  return bn_bwd_impl<false>(x, y, dy, N, C, gamma, save_mean, save_rstd, training, relu, dx, dgamma, dbeta, ws, ws_bytes, stream);
}

// bfloat16 activations (x, y, dy, dx); parameters, statistics and their gradients stay fp32
extern "C" int btc_bn_relu_fwd_bf16(const void* x, int N, int C, const float* gamma, const float* beta, float* running_mean,
                                    float* running_var, long long* num_batches_tracked, float momentum, float eps, int training, int relu,
                                    void* y, float* save_mean, float* save_rstd, void* ws, size_t ws_bytes, void* stream) {
  return bn_fwd_impl<true>((const float*)x, N, C, gamma, beta, running_mean, running_var, num_batches_tracked, momentum, eps, training, relu,
                           (float*)y, save_mean, save_rstd, ws, ws_bytes, stream);
}

extern "C" int btc_bn_relu_bwd_bf16(const void* x, const void* y, const void* dy, int N, int C, const float* gamma, const float* save_mean,
                                    const float* save_rstd, int training, int relu, void* dx, float* dgamma, float* dbeta, void* ws,
                                    size_t ws_bytes, void* stream) {
  return bn_bwd_impl<true>((const float*)x, (const float*)y, (const float*)dy, N, C, gamma, save_mean, save_rstd, training, relu, (float*)dx,
                           dgamma, dbeta, ws, ws_bytes, stream);
}

// ---- conv -> BatchNorm (training) -> ReLU with the statistics gathered in the conv's epilogue -------------------------------
extern "C" size_t btc_bn_fuse_ws_bytes(void) { return btc_bn_fuse_bytes(); }

extern "C" int btc_conv_bn_relu_fwd(int operands, const void* src, const void* W, const float* bias, const int32_t* nbr, const int32_t* order,
                                    int n_rows, int K, int Cin, int Cout, void* x, const float* gamma, const float* beta, float* running_mean,
                                    float* running_var, long long* num_batches_tracked, float momentum, float eps, int relu, void* y,
                                    float* save_mean, float* save_rstd, void* ws, size_t ws_bytes, void* fuse_ws, void* stream) {
  return btc_conv_bn_relu_fwd_src(operands, src, -1LL, W, bias, nbr, order, n_rows, K, Cin, Cout, x, gamma, beta, running_mean, running_var,
                                  num_batches_tracked, momentum, eps, relu, y, save_mean, save_rstd, ws, ws_bytes, fuse_ws, stream);
}

extern "C" int btc_conv_bn_relu_fwd_src(int operands, const void* src, long long src_rows, const void* W, const float* bias, const int32_t* nbr,
                                        const int32_t* order, int n_rows, int K, int Cin, int Cout, void* x, const float* gamma, const float* beta,
                                        float* running_mean, float* running_var, long long* num_batches_tracked, float momentum, float eps, int relu,
                                        void* y, float* save_mean, float* save_rstd, void* ws, size_t ws_bytes, void* fuse_ws, void* stream) {
  BTC_CHECK_ARG(n_rows >= 1, "btc_conv_bn_relu_fwd: empty input");
  BTC_CHECK_ARG(operands >= BTC_OPERANDS_F32 && operands <= BTC_OPERANDS_F32_SPLIT, "btc_conv_bn_relu_fwd: operands=%d", operands);
  const bool bf = operands == BTC_OPERANDS_BF16_ACT || operands == BTC_OPERANDS_BF16;
  int fused = 0;
  if (fuse_ws && Cout <= BN_FUSE_CMAX && btc_tune_get(BTC_TUNE_BN_FUSE) != 1) {
    BnFuse bn;
    bn.counter = (int32_t*)fuse_ws;
    bn.slots = (double*)((char*)fuse_ws + 256);
    bn.mean_out = save_mean; bn.rstd_out = save_rstd;
    bn.running_mean = running_mean; bn.running_var = running_var; bn.num_batches = num_batches_tracked;
    bn.momentum = momentum; bn.eps = eps; bn.N = n_rows; bn.C = Cout; bn.nslots = btc_bn_fuse_nslots(n_rows);
    int rc = btc_conv_fwd_stats(operands, src, src_rows, (const float*)W, bias, nbr, order, n_rows, K, Cin, Cout, x, bn, (hipStream_t)stream, &fused);
    if (rc) return rc;
  } else {
    int rc = btc_conv_apply_src(BTC_PASS_FWD, operands, src, src_rows, W, bias, nbr, order, n_rows, K, Cin, Cout, x, stream);
    if (rc) return rc;
  }
  if (bf)
    return bn_fwd_impl<true>((const float*)x, n_rows, Cout, gamma, beta, running_mean, running_var, num_batches_tracked, momentum, eps, 1, relu,
                             (float*)y, save_mean, save_rstd, ws, ws_bytes, stream, fused != 0);
  return bn_fwd_impl<false>((const float*)x, n_rows, Cout, gamma, beta, running_mean, running_var, num_batches_tracked, momentum, eps, 1, relu,
                            (float*)y, save_mean, save_rstd, ws, ws_bytes, stream, fused != 0);
}
