// pointnet2_stack on gfx950: ball / shell query, grouping (+grad), furthest point sampling, three-NN, three-interpolate
// (+grad).  Replaces the compiled module `pointnet2_stack_cuda` the reference binds in
// /root/reference/btcdet/ops/pointnet2/pointnet2_stack/pointnet2_utils.py:5,34-36,76,98,208,237,271,289 (kernels:
// src/ball_query_gpu.cu:15-66, shell_query_gpu.cu:15-68, group_points_gpu.cu:14-46,64-97, sampling_gpu.cu:16-142,
// interpolate_gpu.cu:14-69,106-121,141-158).  SURVEY.md §8f row 2: the ROI head's grid pooling (conv_head.py:247-379).
//
// Same results, different work decomposition:
//   ball / shell query  the reference scans a scene's points with ONE THREAD per query (O(M N) dependent loads, 55 K queries
//                       x 28 K points for the ROI grid).  Here one 64-lane WAVE owns a query: 64 points per step, the
//                       in-range lanes are ranked with a ballot + popcount, so hits are appended in index order exactly as
//                       the sequential scan would, and the wave stops once nsample are found.  The points of the scene are
//                       staged through LDS in tiles shared by the 16 waves of a workgroup (one global read per 16 queries).
//   furthest point sampling  one workgroup per scene as in the reference, but the points and running distances of a
//                       thread stay in registers across the npoint iterations (<= 16 points per thread; larger scenes fall
//                       back to global memory) and the argmax is reduced with wave shuffles + one LDS exchange; ties resolve
//                       exactly as the reference's strided scan + tree reduction does: thread t = k mod T keeps its first maximum and the
//                       tree (t, t + half) lets the lower slot win, so among equal maxima the smallest bit-reversed t wins, then
//                       the smallest k; T = the power-of-two thread count the reference would have launched for n points.
//   grouping / interpolation  one thread per output element, coalesced along the sample / channel axis; the gradients use
//                       float atomics like the reference (summation order unspecified there too).
#include "btc_common.h"

namespace {

constexpr int BQ_WAVES = 16;        // queries per workgroup
constexpr int BQ_TILE = 1024;       // points per LDS tile (12 KB)

__device__ __forceinline__ int scene_of(int pt, const int32_t* __restrict__ cnt, int B, int* start_other, const int32_t* __restrict__ other) {
  // scene index of stacked row `pt` (rows of scene b: [sum cnt[:b], sum cnt[:b+1])); *start_other = sum other[:b]
  int bs = 0, acc = cnt[0];
  for (int k = 1; k < B; ++k) {
    if (pt < acc) break;
    acc += cnt[k];
    bs = k;
  }
  int s = 0;
  for (int k = 0; k < bs; ++k) s += other[k];
  *start_other = s;
  return bs;
}

// idx (M, nsample) int32, zero on entry (the reference's .zero_()).  in-range: inner2 <= d2 < outer2 (ball: inner2 = -1)
__global__ __launch_bounds__(BQ_WAVES * 64) void ball_query_k(const float* __restrict__ new_xyz, const int32_t* __restrict__ new_cnt,
                                                               const float* __restrict__ xyz, const int32_t* __restrict__ cnt, int B, int M,
                                                               float inner2, float outer2, int nsample, int32_t* __restrict__ idx) {
  __shared__ float s_xyz[BQ_TILE * 3];
  __shared__ int s_active;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int q = blockIdx.x * BQ_WAVES + wave;
  const bool valid = q < M;
  // all queries of a workgroup that share the first query's scene walk its tiles together; a workgroup that straddles a
  // scene boundary handles the other scene's queries in a second pass
  int start0 = 0, start = 0, bs = -1, n = 0;
  float qx = 0.f, qy = 0.f, qz = 0.f;
  if (valid) {
    bs = scene_of(q, new_cnt, B, &start, cnt);
    n = cnt[bs];
    qx = new_xyz[(size_t)q * 3 + 0]; qy = new_xyz[(size_t)q * 3 + 1]; qz = new_xyz[(size_t)q * 3 + 2];
  }
  int found = 0, first = 0;
  bool done = !valid;
  const int q_first = blockIdx.x * BQ_WAVES, q_last = min(q_first + BQ_WAVES, M) - 1;
  const int bs_lo = scene_of(q_first, new_cnt, B, &start0, cnt);
  int dummy;
  const int bs_hi = scene_of(q_last, new_cnt, B, &dummy, cnt);
  for (int scene = bs_lo; scene <= bs_hi; ++scene) {
    int sc_start = 0;
    for (int k = 0; k < scene; ++k) sc_start += cnt[k];
    const int sc_n = cnt[scene];
    const bool mine = valid && bs == scene;
    for (int t0 = 0; t0 < sc_n; t0 += BQ_TILE) {
      // stop early when every wave of this scene is done
      if (threadIdx.x == 0) s_active = 0;
      __syncthreads();
      if (mine && !done && lane == 0) s_active = 1;
      __syncthreads();
      if (!s_active) break;
      const int tn = min(BQ_TILE, sc_n - t0);
      const float* src = xyz + (size_t)(sc_start + t0) * 3;
      for (int i = threadIdx.x; i < tn * 3; i += BQ_WAVES * 64) s_xyz[i] = src[i];
      __syncthreads();
      if (mine && !done) {
        for (int j0 = 0; j0 < tn && !done; j0 += 64) {
          const int j = j0 + lane;
          bool hit = false;
          if (j < tn) {
            const float x = s_xyz[j * 3 + 0], y = s_xyz[j * 3 + 1], z = s_xyz[j * 3 + 2];
            const float d2 = (qx - x) * (qx - x) + (qy - y) * (qy - y) + (qz - z) * (qz - z);
            hit = d2 >= inner2 && d2 < outer2;
          }
          const unsigned long long m = __ballot(hit);
          if (m) {
            if (found == 0) first = t0 + j0 + (int)__builtin_ctzll(m);
            const int rank = found + (int)__popcll(m & ((1ull << lane) - 1ull));
            if (hit && rank < nsample) idx[(size_t)q * nsample + rank] = t0 + j;
            found += (int)__popcll(m);
            if (found >= nsample) done = true;
          }
        }
      }
      __syncthreads();
    }
  }
  if (valid) {
    if (found == 0) {
      if (lane == 0) idx[(size_t)q * nsample] = -1;
    } else {
      for (int l = min(found, nsample) + lane; l < nsample; l += 64) idx[(size_t)q * nsample + l] = first;  // padded with the first hit
    }
  }
}

__global__ __launch_bounds__(256) void group_points_k(const float* __restrict__ features, const int32_t* __restrict__ f_cnt,
                                                      const int32_t* __restrict__ idx, const int32_t* __restrict__ idx_cnt, int B, int M, int C,
                                                      int nsample, float* __restrict__ out) {
  const long long index = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const int s = (int)(index % nsample), c = (int)((index / nsample) % C);
  const long long pt = index / nsample / C;
  if (pt >= M) return;
  int fstart;
  scene_of((int)pt, idx_cnt, B, &fstart, f_cnt);
  out[index] = features[(size_t)(fstart + idx[pt * nsample + s]) * C + c];
}

__global__ __launch_bounds__(256) void group_points_grad_k(const float* __restrict__ grad_out, const int32_t* __restrict__ idx,
                                                           const int32_t* __restrict__ idx_cnt, const int32_t* __restrict__ f_cnt, int B, int M,
                                                           int C, int nsample, float* __restrict__ grad_features) {
  const long long index = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const int s = (int)(index % nsample), c = (int)((index / nsample) % C);
  const long long pt = index / nsample / C;
  if (pt >= M) return;
  int fstart;
  scene_of((int)pt, idx_cnt, B, &fstart, f_cnt);
  atomicAdd(&grad_features[(size_t)(fstart + idx[pt * nsample + s]) * C + c], grad_out[index]);
}

// ---------------------------------------------------------------------------------------------- furthest point sampling
struct FpsBest {
  float v;
  int key, k;  // key = bit-reversed (k mod T) of the reference launch; ties: smallest (key, k)
  float x, y, z;  // the candidate's coordinates travel with it: the next iteration starts without a global load
};

__device__ __forceinline__ bool fps_wins(const FpsBest& b, const FpsBest& a) {  // does b replace a?
  if (b.v > a.v) return true;
  if (!(b.v == a.v)) return false;  // smaller, or NaN (which never wins a strict '>' in the reference either)
  return b.key < a.key || (b.key == a.key && b.k < a.k);
}

__device__ __forceinline__ FpsBest fps_shfl_down(const FpsBest& b, int o) {
  return FpsBest{__shfl_down(b.v, o, 64), __shfl_down(b.key, o, 64), __shfl_down(b.k, o, 64), __shfl_down(b.x, o, 64),
                 __shfl_down(b.y, o, 64), __shfl_down(b.z, o, 64)};
}

// One barrier per iteration: every wave reduces its candidates with shuffles, lane 0 publishes the wave's winner (value, tie
// key, index, coordinates) in a double-buffered LDS table, and after the barrier EVERY wave reduces the 16 entries itself.
// Register budget: the per-thread point arrays must not spill (a scratch access per iteration costs more than the whole
// reduction): 1024 threads x 4 points (<= 4 K points), or 512 threads x 32 points with 2 waves per SIMD = 256 VGPRs per lane.
template <int PER, bool REG, int THREADS>
__global__ __launch_bounds__(THREADS) __attribute__((amdgpu_waves_per_eu(THREADS / 256, THREADS / 256))) void fps_k(int n, int m, int T, int log2T, const float* __restrict__ dataset, float* __restrict__ temp,
                                              int32_t* __restrict__ idxs) {
  if (m <= 0) return;
  constexpr int NW = THREADS / 64;
  __shared__ float s_v[2][NW], s_x[2][NW], s_y[2][NW], s_z[2][NW];
  __shared__ int s_key[2][NW], s_k[2][NW];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  dataset += (size_t)blockIdx.x * n * 3;
  temp += (size_t)blockIdx.x * n;
  idxs += (size_t)blockIdx.x * m;
  float px[PER], py[PER], pz[PER], pt[PER];
  if (REG) {
#pragma unroll
    for (int u = 0; u < PER; ++u) {
      const int k = tid + u * THREADS;
      if (k < n) { px[u] = dataset[k * 3 + 0]; py[u] = dataset[k * 3 + 1]; pz[u] = dataset[k * 3 + 2]; pt[u] = temp[k]; }
    }
  }
  auto key_of = [&](int k) { return log2T ? (int)(__brev((unsigned)(k & (T - 1))) >> (32 - log2T)) : 0; };
  float x1 = dataset[0], y1 = dataset[1], z1 = dataset[2];
  if (tid == 0) idxs[0] = 0;
  for (int j = 1; j < m; ++j) {
    // a thread without a candidate above the reference's initial best (-1) contributes (v = -1, index 0), like the reference
    FpsBest best{-1.f, 0, 0, 0.f, 0.f, 0.f};
    bool any = false;
    if (REG) {
      // in the loop only (value, slot) are live; key and coordinates are looked up once, after it (register pressure)
      float bv = -1.f;
      int bu = -1;
#pragma unroll
      for (int u = 0; u < PER; ++u) {
        const int k = tid + u * THREADS;
        if (k < n) {
          const float d = (px[u] - x1) * (px[u] - x1) + (py[u] - y1) * (py[u] - y1) + (pz[u] - z1) * (pz[u] - z1);
          const float d2 = fminf(d, pt[u]);
          pt[u] = d2;
          bool take = d2 > bv;
          if (!take && bu >= 0 && d2 == bv) {  // tie inside the thread: smaller (key, k) wins; k grows with u
            take = key_of(k) < key_of(tid + bu * THREADS);
          }
          if (take) { bv = d2; bu = u; }
        }
      }
      if (bu >= 0) {
        any = true;
        best.v = bv;
        best.k = tid + bu * THREADS;
        best.key = key_of(best.k);
#pragma unroll
        for (int u = 0; u < PER; ++u)
          if (u == bu) { best.x = px[u]; best.y = py[u]; best.z = pz[u]; }
      }
    } else {
      for (int k = tid; k < n; k += THREADS) {
        const float x2 = dataset[k * 3 + 0], y2 = dataset[k * 3 + 1], z2 = dataset[k * 3 + 2];
        const float d = (x2 - x1) * (x2 - x1) + (y2 - y1) * (y2 - y1) + (z2 - z1) * (z2 - z1);
        const float d2 = fminf(d, temp[k]);
        temp[k] = d2;
        const FpsBest c{d2, key_of(k), k, x2, y2, z2};
        if (any ? fps_wins(c, best) : d2 > -1.f) { best = c; any = true; }
      }
    }
    // wave winner of (value, tie key, index) by shuffles; its coordinates are published by the lane that owns it
    const int own_k = any ? best.k : -1;
    const float ox = best.x, oy = best.y, oz = best.z;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float ov = __shfl_down(best.v, o, 64);
      const int okey = __shfl_down(best.key, o, 64), ok = __shfl_down(best.k, o, 64);
      if (ov > best.v || (ov == best.v && (okey < best.key || (okey == best.key && ok < best.k)))) { best.v = ov; best.key = okey; best.k = ok; }
    }
    const int buf = j & 1;
    const float wv = __shfl(best.v, 0, 64);
    const int wk = __shfl(best.k, 0, 64);
    if (lane == 0) { s_v[buf][wave] = best.v; s_key[buf][wave] = best.key; s_k[buf][wave] = best.k; }
    if (wv > -1.f) {
      if (own_k == wk) { s_x[buf][wave] = ox; s_y[buf][wave] = oy; s_z[buf][wave] = oz; }
    } else if (lane == 0) {  // no candidate in the whole wave: the reference's (-1, index 0)
      s_x[buf][wave] = dataset[0]; s_y[buf][wave] = dataset[1]; s_z[buf][wave] = dataset[2];
    }
    __syncthreads();
    const int l = lane & (NW - 1);
    float bv = s_v[buf][l];
    int bkey = s_key[buf][l], bk = s_k[buf][l], bw = l;
#pragma unroll
    for (int o = NW / 2; o > 0; o >>= 1) {  // lanes 0..NW-1 hold the waves' winners; the lanes above mirror them and are ignored
      const float ov = __shfl_down(bv, o, 64);
      const int okey = __shfl_down(bkey, o, 64), ok = __shfl_down(bk, o, 64), ow = __shfl_down(bw, o, 64);
      if (l + o < NW && (ov > bv || (ov == bv && (okey < bkey || (okey == bkey && ok < bk))))) { bv = ov; bkey = okey; bk = ok; bw = ow; }
    }
    const int win_w = __shfl(bw, 0, 64), win_k = __shfl(bk, 0, 64);
    x1 = s_x[buf][win_w]; y1 = s_y[buf][win_w]; z1 = s_z[buf][win_w];
    if (tid == 0) idxs[j] = win_k;
  }
  if (REG) {
#pragma unroll
    for (int u = 0; u < PER; ++u) {
      const int k = tid + u * THREADS;
      if (k < n) temp[k] = pt[u];
    }
  }
}

// ---------------------------------------------------------------------------------------------- three-NN / interpolation
constexpr int NN_TILE = 1024;

__global__ __launch_bounds__(256) void three_nn_k(int B, int N, const float* __restrict__ unknown, const int32_t* __restrict__ u_cnt,
                                                  const float* __restrict__ known, const int32_t* __restrict__ k_cnt,
                                                  float* __restrict__ dist2, int32_t* __restrict__ idx) {
  __shared__ float s_xyz[NN_TILE * 3];
  const int pt = blockIdx.x * blockDim.x + threadIdx.x;
  const bool valid = pt < N;
  int kstart = 0, bs = -1;
  float ux = 0.f, uy = 0.f, uz = 0.f;
  if (valid) {
    bs = scene_of(pt, u_cnt, B, &kstart, k_cnt);
    ux = unknown[(size_t)pt * 3 + 0]; uy = unknown[(size_t)pt * 3 + 1]; uz = unknown[(size_t)pt * 3 + 2];
  }
  double best1 = 1e40, best2 = 1e40, best3 = 1e40;
  int b1 = 0, b2 = 0, b3 = 0;
  const int p_first = blockIdx.x * blockDim.x, p_last = min(p_first + (int)blockDim.x, N) - 1;
  int d0, d1;
  const int bs_lo = scene_of(p_first, u_cnt, B, &d0, k_cnt), bs_hi = scene_of(p_last, u_cnt, B, &d1, k_cnt);
  for (int scene = bs_lo; scene <= bs_hi; ++scene) {
    int sc_start = 0;
    for (int k = 0; k < scene; ++k) sc_start += k_cnt[k];
    const int sc_n = k_cnt[scene];
    for (int t0 = 0; t0 < sc_n; t0 += NN_TILE) {
      const int tn = min(NN_TILE, sc_n - t0);
      const float* src = known + (size_t)(sc_start + t0) * 3;
      __syncthreads();
      for (int i = threadIdx.x; i < tn * 3; i += blockDim.x) s_xyz[i] = src[i];
      __syncthreads();
      if (valid && bs == scene) {
        for (int j = 0; j < tn; ++j) {
          const float x = s_xyz[j * 3 + 0], y = s_xyz[j * 3 + 1], z = s_xyz[j * 3 + 2];
          const float d = (ux - x) * (ux - x) + (uy - y) * (uy - y) + (uz - z) * (uz - z);
          const int k = t0 + j;
          if (d < best1) { best3 = best2; b3 = b2; best2 = best1; b2 = b1; best1 = d; b1 = k; }
          else if (d < best2) { best3 = best2; b3 = b2; best2 = d; b2 = k; }
          else if (d < best3) { best3 = d; b3 = k; }
        }
      }
    }
  }
  if (valid) {
    dist2[(size_t)pt * 3 + 0] = (float)best1; dist2[(size_t)pt * 3 + 1] = (float)best2; dist2[(size_t)pt * 3 + 2] = (float)best3;
    idx[(size_t)pt * 3 + 0] = b1 + kstart; idx[(size_t)pt * 3 + 1] = b2 + kstart; idx[(size_t)pt * 3 + 2] = b3 + kstart;
  }
}

__global__ __launch_bounds__(256) void three_interpolate_k(long long total, int C, const float* __restrict__ features, const int32_t* __restrict__ idx,
                                                           const float* __restrict__ weight, float* __restrict__ out) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const long long pt = i / C;
  const int c = (int)(i % C);
  const int32_t* id = idx + pt * 3;
  const float* w = weight + pt * 3;
  out[i] = w[0] * features[(size_t)id[0] * C + c] + w[1] * features[(size_t)id[1] * C + c] + w[2] * features[(size_t)id[2] * C + c];
}

__global__ __launch_bounds__(256) void three_interpolate_grad_k(long long total, int C, const float* __restrict__ grad_out,
                                                                const int32_t* __restrict__ idx, const float* __restrict__ weight,
                                                                float* __restrict__ grad_features) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const long long pt = i / C;
  const int c = (int)(i % C);
  const int32_t* id = idx + pt * 3;
  const float* w = weight + pt * 3;
  const float g = grad_out[i];
  atomicAdd(&grad_features[(size_t)id[0] * C + c], g * w[0]);
  atomicAdd(&grad_features[(size_t)id[1] * C + c], g * w[1]);
  atomicAdd(&grad_features[(size_t)id[2] * C + c], g * w[2]);
}

}  // namespace

extern "C" int btc_ball_query(const float* new_xyz, const int32_t* new_xyz_batch_cnt, const float* xyz, const int32_t* xyz_batch_cnt, int B,
                              int M, float inner_radius, float outer_radius, int nsample, int32_t* idx, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  BTC_CHECK_ARG(B >= 1 && M >= 0 && nsample >= 1, "btc_ball_query: bad sizes");
  if (M == 0) return BTC_OK;
  BTC_HIP(hipMemsetAsync(idx, 0, (size_t)M * nsample * sizeof(int32_t), stream));
  const float inner2 = inner_radius < 0.f ? -1.f : inner_radius * inner_radius;
  ball_query_k<<<btc_cdiv(M, BQ_WAVES), BQ_WAVES * 64, 0, stream>>>(new_xyz, new_xyz_batch_cnt, xyz, xyz_batch_cnt, B, M, inner2,
                                                                     outer_radius * outer_radius, nsample, idx);
  BTC_LAUNCH_CHECK();
  return BTC_OK;
}

extern "C" int btc_group_points(const float* features, const int32_t* features_batch_cnt, const int32_t* idx, const int32_t* idx_batch_cnt,
                                int B, int M, int C, int nsample, float* out, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  BTC_CHECK_ARG(B >= 1 && M >= 0 && C >= 1 && nsample >= 1, "btc_group_points: bad sizes");
  const long long total = (long long)M * C * nsample;
  if (total == 0) return BTC_OK;
  group_points_k<<<(unsigned)((total + 255) / 256), 256, 0, stream>>>(features, features_batch_cnt, idx, idx_batch_cnt, B, M, C, nsample, out);
  BTC_LAUNCH_CHECK();
  return BTC_OK;
}

extern "C" int btc_group_points_grad(const float* grad_out, const int32_t* idx, const int32_t* idx_batch_cnt,
                                     const int32_t* features_batch_cnt, int B, int M, int C, int N, int nsample, float* grad_features,
                                     void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  BTC_CHECK_ARG(B >= 1 && M >= 0 && C >= 1 && N >= 0 && nsample >= 1, "btc_group_points_grad: bad sizes");
  if (N > 0) BTC_HIP(hipMemsetAsync(grad_features, 0, (size_t)N * C * sizeof(float), stream));
  const long long total = (long long)M * C * nsample;
  if (total == 0) return BTC_OK;
  group_points_grad_k<<<(unsigned)((total + 255) / 256), 256, 0, stream>>>(grad_out, idx, idx_batch_cnt, features_batch_cnt, B, M, C, nsample,
                                                                          grad_features);
  BTC_LAUNCH_CHECK();
  return BTC_OK;
}

extern "C" int btc_furthest_point_sampling(const float* xyz, int B, int N, int npoint, float* temp, int32_t* idx, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  BTC_CHECK_ARG(B >= 0 && N >= 1 && npoint >= 0, "btc_furthest_point_sampling: bad sizes");
  if (B == 0 || npoint == 0) return BTC_OK;
  int T = 1, log2T = 0;  // threads the reference launches for n points: the largest power of two <= n, capped at 1024 (opt_n_threads)
  while (T * 2 <= N && T < 1024) { T *= 2; ++log2T; }
  if (N <= 4 * 1024) fps_k<4, true, 1024><<<B, 1024, 0, stream>>>(N, npoint, T, log2T, xyz, temp, idx);
  else if (N <= 16 * 1024) fps_k<32, true, 512><<<B, 512, 0, stream>>>(N, npoint, T, log2T, xyz, temp, idx);
  else fps_k<1, false, 1024><<<B, 1024, 0, stream>>>(N, npoint, T, log2T, xyz, temp, idx);
  BTC_LAUNCH_CHECK();
  return BTC_OK;
}

extern "C" int btc_three_nn(const float* unknown, const int32_t* unknown_batch_cnt, const float* known, const int32_t* known_batch_cnt, int B,
                            int N, float* dist2, int32_t* idx, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  BTC_CHECK_ARG(B >= 1 && N >= 0, "btc_three_nn: bad sizes");
  if (N == 0) return BTC_OK;
  three_nn_k<<<btc_cdiv(N, 256), 256, 0, stream>>>(B, N, unknown, unknown_batch_cnt, known, known_batch_cnt, dist2, idx);
  BTC_LAUNCH_CHECK();
  return BTC_OK;
}

extern "C" int btc_three_interpolate(const float* features, const int32_t* idx, const float* weight, int N, int C, float* out, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  BTC_CHECK_ARG(N >= 0 && C >= 1, "btc_three_interpolate: bad sizes");
  const long long total = (long long)N * C;
  if (total == 0) return BTC_OK;
  three_interpolate_k<<<(unsigned)((total + 255) / 256), 256, 0, stream>>>(total, C, features, idx, weight, out);
  BTC_LAUNCH_CHECK();
  return BTC_OK;
}

extern "C" int btc_three_interpolate_grad(const float* grad_out, const int32_t* idx, const float* weight, int N, int C, int M,
                                          float* grad_features, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  BTC_CHECK_ARG(N >= 0 && C >= 1 && M >= 0, "btc_three_interpolate_grad: bad sizes");
  if (M > 0) BTC_HIP(hipMemsetAsync(grad_features, 0, (size_t)M * C * sizeof(float), stream));
  const long long total = (long long)N * C;
  if (total == 0) return BTC_OK;
  three_interpolate_grad_k<<<(unsigned)((total + 255) / 256), 256, 0, stream>>>(total, C, grad_out, idx, weight, grad_features);
  BTC_LAUNCH_CHECK();
  return BTC_OK;
}
