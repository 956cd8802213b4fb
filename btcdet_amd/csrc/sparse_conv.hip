// Sparse convolution apply for gfx950: fused, output-stationary implicit GEMM on the fp32 MFMA pipe.
//
// Replaces spconv's indice_conv / indice_subm_conv / indice_inverse_conv (one gather + mm +
// scatter-add launch triple PER KERNEL OFFSET) with ONE launch per layer:
// a workgroup owns 64 output rows, walks the K offsets of the neighbour map, gathers the rows
// that exist into LDS, and accumulates  acc += A_k (64 x Cin) * W[k] (Cin x Cout)  with
// v_mfma_f32_16x16x4_f32.  No atomics, no per-offset temporaries; the summation order is fixed
// (offset ascending, channel ascending) and the f32 MFMA is an exact fmaf chain, so results are
// bit-reproducible and equal to the CPU oracle's fmaf chain.
//
// Roofline: HBM/L2 bound at BtcDet's channel widths: per output row the kernel
// moves (pairs * Cin + Cout) * 4 B and does 2 * pairs * Cin * Cout flop.
#include "btc_common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int TM = 64;       // output rows per workgroup
constexpr int KC = 32;       // reduction-channel chunk staged in LDS
constexpr int LDA = KC + 2;  // bank-conflict-free A fragment reads (ds_read_b32, 2 x 32-lane groups)

__host__ __device__ constexpr int ldb_of(int nt) { return nt * 16 + (((nt * 16) % 32 == 0) ? 16 : 0); }

// out[i] = bias + sum_k feat[nbr[i][k]] @ Wk     (TRANS_W = false: Wk = W[k]      (Cin x Cout), fwd)
// din[j] =        sum_k dout[nbr[j][k]] @ Wk     (TRANS_W = true : Wk = W[k]^T    (Cout x Cin), dgrad)
// Cred = reduction channels, Cres = result channels; W is always stored [K][Cin][Cout].
template <int NT, bool TRANS_W>
__global__ __launch_bounds__(256) void conv_apply(const float* __restrict__ feat, const float* __restrict__ W,
                                                  const float* __restrict__ bias, const int32_t* __restrict__ nbr,
                                                  int n_rows, int K, int Cred, int Cres, float* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int LDB = ldb_of(NT);
  float* As = (float*)smem;                    // [TM][LDA]
  float* Bs = As + TM * LDA;                   // [KC][LDB]
  int32_t* s_nbr = (int32_t*)(Bs + KC * LDB);  // [TM][K]
  int32_t* s_kact = s_nbr + TM * K;            // [K]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int row0 = blockIdx.x * TM;
  const int n0 = blockIdx.y * (NT * 16);

  for (int e = tid; e < K; e += 256) s_kact[e] = 0;
  __syncthreads();
  {
    const long long gbase = (long long)row0 * K;
    const long long gend = (long long)n_rows * K;
    for (int e = tid; e < TM * K; e += 256) {
      int v = (gbase + e < gend) ? nbr[gbase + e] : -1;
      s_nbr[e] = v;
      if (v >= 0) s_kact[e % K] = 1;
    }
  }
  __syncthreads();

  f32x4 acc[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) acc[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int arow = wave * 16 + (lane & 15);
  const int kq = lane >> 4;
  const bool vec4 = (Cred & 3) == 0;

  for (int k = 0; k < K; ++k) {
    if (!s_kact[k]) continue;  // block-uniform
    const float* Wk = W + (size_t)k * Cred * Cres;  // same element count either orientation
    for (int cc = 0; cc < Cred; cc += KC) {
      const int kc = min(KC, Cred - cc);
      // ---- gather A tile: TM rows x KC channels (zeros for missing rows / channels)
      if (vec4) {
        for (int e = tid; e < TM * (KC / 4); e += 256) {
          int r = e / (KC / 4), c = (e % (KC / 4)) * 4;
          int j = s_nbr[r * K + k];
          float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
          if (j >= 0 && c < kc) v = *reinterpret_cast<const float4*>(feat + (size_t)j * Cred + cc + c);
          float* d = As + r * LDA + c;
          d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
        }
      } else {
        for (int e = tid; e < TM * KC; e += 256) {
          int r = e / KC, c = e % KC;
          int j = s_nbr[r * K + k];
          As[r * LDA + c] = (j >= 0 && c < kc) ? feat[(size_t)j * Cred + cc + c] : 0.f;
        }
      }
      // ---- stage B tile: KC reduction channels x NT*16 result channels of W[k]
      if (!TRANS_W) {
        for (int e = tid; e < KC * (NT * 16); e += 256) {
          int r = e / (NT * 16), c = e % (NT * 16);
          Bs[r * LDB + c] = (r < kc && n0 + c < Cres) ? Wk[(size_t)(cc + r) * Cres + n0 + c] : 0.f;
        }
      } else {
        // Wk^T[r = co][c = ci] = W[k][ci][co]; read along co (contiguous), write transposed
        for (int e = tid; e < KC * (NT * 16); e += 256) {
          int c = e / KC, r = e % KC;
          Bs[r * LDB + c] = (r < kc && n0 + c < Cres) ? Wk[(size_t)(n0 + c) * Cred + cc + r] : 0.f;
        }
      }
      __syncthreads();
      const int steps = (kc + 3) >> 2;
      for (int q = 0; q < steps; ++q) {
        float a = As[arow * LDA + q * 4 + kq];
        const float* bp = Bs + (q * 4 + kq) * LDB + (lane & 15);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bp[nt * 16], acc[nt], 0, 0, 0);
      }
      __syncthreads();
    }
  }

  // ---- epilogue: C/D layout of 16x16: col = lane&15, row = (lane>>4)*4 + reg
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int col = n0 + nt * 16 + (lane & 15);
    if (col >= Cres) continue;
    const float bv = bias ? bias[col] : 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      int row = row0 + wave * 16 + kq * 4 + r;
      if (row < n_rows) out[(size_t)row * Cres + col] = bias ? (acc[nt][r] + bv) : acc[nt][r];
    }
  }
}

// dW partial: part[s][k][ci][co] = sum over the split's rows of feat[nbr[i][k]][ci] * dout[i][co]
// block = (k, split, tile of 64 Cin x NT*16 Cout); wave w owns dW rows [m0 + 16w, m0 + 16w + 16)
constexpr int WG_LDA = 64 + 16;
template <int NT>
__global__ __launch_bounds__(256) void conv_wgrad_partial(const float* __restrict__ feat, const float* __restrict__ dout,
                                                          const int32_t* __restrict__ nbr, int n_out, int K, int Cin,
                                                          int Cout, int tiles_per_split, int n_cblk, float* __restrict__ part) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int LDB = ldb_of(NT);
  float* As = (float*)smem;       // [TM][WG_LDA]  gathered input rows, 64 channels
  float* Ds = As + TM * WG_LDA;   // [TM][LDB]     dout rows, NT*16 channels
  int32_t* s_j = (int32_t*)(Ds + TM * LDB);  // [TM] (kept inside the one dynamic LDS array)

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int k = blockIdx.x, split = blockIdx.y;
  const int m0 = (blockIdx.z / n_cblk) * 64;
  const int n0 = (blockIdx.z % n_cblk) * (NT * 16);
  const int n_tiles = (n_out + TM - 1) / TM;
  const int t_begin = split * tiles_per_split;
  const int t_end = min(n_tiles, t_begin + tiles_per_split);

  f32x4 acc[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) acc[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const int kq = lane >> 4;
  const bool wave_live = (m0 + wave * 16) < Cin;

  for (int t = t_begin; t < t_end; ++t) {
    const int row0 = t * TM;
    int j = -1;
    if (tid < TM && row0 + tid < n_out) j = nbr[(size_t)(row0 + tid) * K + k];
    if (tid < TM) s_j[tid] = j;
    if (!__syncthreads_or(j >= 0)) continue;
    for (int e = tid; e < TM * 64; e += 256) {
      int r = e >> 6, c = e & 63;
      int jj = s_j[r];
      As[r * WG_LDA + c] = (jj >= 0 && m0 + c < Cin) ? feat[(size_t)jj * Cin + m0 + c] : 0.f;
    }
    for (int e = tid; e < TM * (NT * 16); e += 256) {
      int r = e / (NT * 16), c = e % (NT * 16);
      Ds[r * LDB + c] = (s_j[r] >= 0 && n0 + c < Cout) ? dout[(size_t)(row0 + r) * Cout + n0 + c] : 0.f;
    }
    __syncthreads();
    if (wave_live) {
#pragma unroll 4
      for (int q = 0; q < TM / 4; ++q) {
        float a = As[(q * 4 + kq) * WG_LDA + wave * 16 + (lane & 15)];
        const float* bp = Ds + (q * 4 + kq) * LDB + (lane & 15);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bp[nt * 16], acc[nt], 0, 0, 0);
      }
    }
    __syncthreads();
  }
  float* P = part + ((size_t)split * K + k) * Cin * Cout;
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int col = n0 + nt * 16 + (lane & 15);
    if (col >= Cout) continue;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      int ci = m0 + wave * 16 + kq * 4 + r;
      if (ci < Cin) P[(size_t)ci * Cout + col] = acc[nt][r];
    }
  }
}

__global__ __launch_bounds__(256) void wgrad_reduce(const float* __restrict__ part, int S, long long count,
                                                    float* __restrict__ dW) {
  long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= count) return;
  float s = 0.f;
  for (int q = 0; q < S; ++q) s += part[(size_t)q * count + e];
  dW[e] = s;
}

__global__ __launch_bounds__(256) void maxpool_fwd_k(const float* __restrict__ feat, const int32_t* __restrict__ nbr,
                                                     int n_out, int K, int C, float* __restrict__ out) {
  long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (long long)n_out * C) return;
  int i = (int)(t / C), c = (int)(t % C);
  float m = 0.f;
  for (int k = 0; k < K; ++k) {
    int j = nbr[(size_t)i * K + k];
    if (j >= 0) m = fmaxf(m, feat[(size_t)j * C + c]);
  }
  out[t] = m;
}

__global__ __launch_bounds__(256) void maxpool_bwd_k(const float* __restrict__ feat, const float* __restrict__ out,
                                                     const float* __restrict__ dout, const int32_t* __restrict__ nbr_in,
                                                     int n_in, int K, int C, float* __restrict__ din) {
  long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (long long)n_in * C) return;
  int j = (int)(t / C), c = (int)(t % C);
  float v = feat[t], g = 0.f;
  for (int k = 0; k < K; ++k) {
    int i = nbr_in[(size_t)j * K + k];
    if (i >= 0 && out[(size_t)i * C + c] == v) g += dout[(size_t)i * C + c];
  }
  din[t] = g;
}

// dense[b][c][z][y][x] = feat[row][c]; consecutive threads take consecutive rows of one channel so the
// dense writes coalesce for (b,z,y,x)-sorted tensors (feature reads are strided but L2 resident)
__global__ __launch_bounds__(256) void dense_fwd_k(const float* __restrict__ feat, const int4* __restrict__ idx, int n, int C,
                                                   int D, int H, int Wd, float* __restrict__ dense) {
  long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (long long)n * C) return;
  int c = (int)(t / n), i = (int)(t % n);  // consecutive threads -> consecutive rows of one channel
  int4 q = idx[i];
  size_t vol = (size_t)D * H * Wd;
  dense[((size_t)q.x * C + c) * vol + ((size_t)q.y * H + q.z) * Wd + q.w] = feat[(size_t)i * C + c];
}

__global__ __launch_bounds__(256) void dense_bwd_k(const float* __restrict__ ddense, const int4* __restrict__ idx, int n, int C,
                                                   int D, int H, int Wd, float* __restrict__ dfeat) {
  long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (long long)n * C) return;
  int c = (int)(t / n), i = (int)(t % n);
  int4 q = idx[i];
  size_t vol = (size_t)D * H * Wd;
  dfeat[(size_t)i * C + c] = ddense[((size_t)q.x * C + c) * vol + ((size_t)q.y * H + q.z) * Wd + q.w];
}

template <bool TRANS_W>
int launch_apply(const float* feat, const float* W, const float* bias, const int32_t* nbr, int n_rows, int K, int Cred,
                 int Cres, float* out, hipStream_t stream) {
  if (n_rows <= 0) return BTC_OK;
  int nt = Cres <= 16 ? 1 : (Cres <= 32 ? 2 : (Cres <= 64 ? 4 : 8));
  dim3 grid(btc_cdiv(n_rows, TM), btc_cdiv(Cres, nt * 16));
  size_t lds = (size_t)(TM * LDA + KC * ldb_of(nt)) * sizeof(float) + (size_t)(TM * K + K) * sizeof(int32_t);
  switch (nt) {
    case 1: conv_apply<1, TRANS_W><<<grid, 256, lds, stream>>>(feat, W, bias, nbr, n_rows, K, Cred, Cres, out); break;
    case 2: conv_apply<2, TRANS_W><<<grid, 256, lds, stream>>>(feat, W, bias, nbr, n_rows, K, Cred, Cres, out); break;
    case 4: conv_apply<4, TRANS_W><<<grid, 256, lds, stream>>>(feat, W, bias, nbr, n_rows, K, Cred, Cres, out); break;
    default: conv_apply<8, TRANS_W><<<grid, 256, lds, stream>>>(feat, W, bias, nbr, n_rows, K, Cred, Cres, out); break;
  }
  BTC_LAUNCH_CHECK();
  return BTC_OK;
}

struct WgradPlan {
  int nt, n_cblk, n_mblk, S, tiles_per_split;
};

WgradPlan wgrad_plan(int n_out, int K, int Cin, int Cout) {
  WgradPlan p;
  p.nt = Cout <= 16 ? 1 : (Cout <= 32 ? 2 : (Cout <= 64 ? 4 : 8));
  p.n_cblk = btc_cdiv(Cout, p.nt * 16);
  p.n_mblk = btc_cdiv(Cin, 64);
  int n_tiles = btc_cdiv(n_out > 0 ? n_out : 1, TM);
  int S = 1536 / (K * p.n_cblk * p.n_mblk);
  if (S > 32) S = 32;
  if (S < 1) S = 1;
  if (S > n_tiles) S = n_tiles;
  p.tiles_per_split = btc_cdiv(n_tiles, S);
  p.S = btc_cdiv(n_tiles, p.tiles_per_split);
  return p;
}

}  // namespace

extern "C" int btc_conv_fwd(const float* feat, const float* W, const float* bias, const int32_t* nbr_out, int n_out, int K,
                            int Cin, int Cout, float* out, void* stream) {
  BTC_CHECK_ARG(K >= 1 && Cin >= 1 && Cout >= 1 && n_out >= 0, "btc_conv_fwd: bad sizes");
  BTC_CHECK_ARG(K <= 512, "btc_conv_fwd: K=%d too large for the LDS neighbour tile", K);
  return launch_apply<false>(feat, W, bias, nbr_out, n_out, K, Cin, Cout, out, (hipStream_t)stream);
}

extern "C" int btc_conv_dgrad(const float* dout, const float* W, const int32_t* nbr_in, int n_in, int K, int Cin, int Cout,
                              float* din, void* stream) {
  BTC_CHECK_ARG(K >= 1 && Cin >= 1 && Cout >= 1 && n_in >= 0, "btc_conv_dgrad: bad sizes");
  BTC_CHECK_ARG(K <= 512, "btc_conv_dgrad: K=%d too large for the LDS neighbour tile", K);
  return launch_apply<true>(dout, W, nullptr, nbr_in, n_in, K, /*Cred=*/Cout, /*Cres=*/Cin, din, (hipStream_t)stream);
}

extern "C" size_t btc_conv_wgrad_ws_bytes(int n_out, int K, int Cin, int Cout) {
  WgradPlan p = wgrad_plan(n_out, K, Cin, Cout);
  return btc_align((size_t)p.S * K * Cin * Cout * sizeof(float));
}

extern "C" int btc_conv_wgrad(const float* feat, const float* dout, const int32_t* nbr_out, int n_out, int K, int Cin,
                              int Cout, float* dW, void* ws, size_t ws_bytes, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  BTC_CHECK_ARG(K >= 1 && Cin >= 1 && Cout >= 1 && n_out >= 0, "btc_conv_wgrad: bad sizes");
  BTC_CHECK_ARG(ws_bytes >= btc_conv_wgrad_ws_bytes(n_out, K, Cin, Cout), "btc_conv_wgrad: workspace too small");
  long long count = (long long)K * Cin * Cout;
  if (n_out <= 0) {
    BTC_HIP(hipMemsetAsync(dW, 0, (size_t)count * sizeof(float), stream));
    return BTC_OK;
  }
  WgradPlan p = wgrad_plan(n_out, K, Cin, Cout);
  dim3 grid(K, p.S, p.n_mblk * p.n_cblk);
  size_t lds = (size_t)(TM * WG_LDA + TM * ldb_of(p.nt)) * sizeof(float) + TM * sizeof(int32_t);
  float* part = (float*)ws;
  switch (p.nt) {
    case 1: conv_wgrad_partial<1><<<grid, 256, lds, stream>>>(feat, dout, nbr_out, n_out, K, Cin, Cout, p.tiles_per_split, p.n_cblk, part); break;
    case 2: conv_wgrad_partial<2><<<grid, 256, lds, stream>>>(feat, dout, nbr_out, n_out, K, Cin, Cout, p.tiles_per_split, p.n_cblk, part); break;
    case 4: conv_wgrad_partial<4><<<grid, 256, lds, stream>>>(feat, dout, nbr_out, n_out, K, Cin, Cout, p.tiles_per_split, p.n_cblk, part); break;
    default: conv_wgrad_partial<8><<<grid, 256, lds, stream>>>(feat, dout, nbr_out, n_out, K, Cin, Cout, p.tiles_per_split, p.n_cblk, part); break;
  }
  BTC_LAUNCH_CHECK();
  wgrad_reduce<<<btc_cdiv(count, 256), 256, 0, stream>>>(part, p.S, count, dW);
  BTC_LAUNCH_CHECK();
  return BTC_OK;
}

extern "C" int btc_maxpool_fwd(const float* feat, const int32_t* nbr_out, int n_out, int K, int C, float* out, void* stream) {
  if (n_out <= 0) return BTC_OK;
  maxpool_fwd_k<<<btc_cdiv((long long)n_out * C, 256), 256, 0, (hipStream_t)stream>>>(feat, nbr_out, n_out, K, C, out);
  BTC_LAUNCH_CHECK();
  return BTC_OK;
}

extern "C" int btc_maxpool_bwd(const float* feat, const float* out, const float* dout, const int32_t* nbr_in, int n_in, int K,
                               int C, float* din, void* stream) {
  if (n_in <= 0) return BTC_OK;
  maxpool_bwd_k<<<btc_cdiv((long long)n_in * C, 256), 256, 0, (hipStream_t)stream>>>(feat, out, dout, nbr_in, n_in, K, C, din);
  BTC_LAUNCH_CHECK();
  return BTC_OK;
}

extern "C" int btc_dense_fwd(const float* feat, const int32_t* indices, int n, int C, const int32_t* h_shape, float* dense,
                             void* stream) {
  if (n <= 0) return BTC_OK;
  dense_fwd_k<<<btc_cdiv((long long)n * C, 256), 256, 0, (hipStream_t)stream>>>(feat, (const int4*)indices, n, C, h_shape[0],
                                                                               h_shape[1], h_shape[2], dense);
  BTC_LAUNCH_CHECK();
  return BTC_OK;
}

extern "C" int btc_dense_bwd(const float* ddense, const int32_t* indices, int n, int C, const int32_t* h_shape, float* dfeat,
                             void* stream) {
  if (n <= 0) return BTC_OK;
  dense_bwd_k<<<btc_cdiv((long long)n * C, 256), 256, 0, (hipStream_t)stream>>>(ddense, (const int4*)indices, n, C, h_shape[0],
                                                                               h_shape[1], h_shape[2], dfeat);
  BTC_LAUNCH_CHECK();
  return BTC_OK;
}
