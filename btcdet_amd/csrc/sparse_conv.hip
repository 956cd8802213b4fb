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
#include <mutex>

#include "btc_common.h"
#include "bn_fuse.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int TM = 64;       // output rows per workgroup
constexpr int KC = 32;       // reduction-channel chunk staged in LDS
constexpr int LDA = KC + 2;  // bank-conflict-free A fragment reads (ds_read_b32, 2 x 32-lane groups)

__host__ __device__ constexpr int ldb_of(int nt) { return nt * 16 + (((nt * 16) % 32 == 0) ? 16 : 0); }

// out[i] = bias + sum_k feat[nbr[i][k]] @ Wk     (TRANS_W = false: Wk = W[k]      (Cin x Cout), fwd)
// din[j] =        sum_k dout[nbr[j][k]] @ Wk     (TRANS_W = true : Wk = W[k]^T    (Cout x Cin), dgrad)
// Cred = reduction channels, Cres = result channels; W is always stored [K][Cin][Cout].
//
// Work items = (active offset k, 32-channel chunk).  Register-staged software pipeline: the global loads of item
// i+1 (gathered rows + weight panel) are issued before the MFMAs of item i and land in registers while the matrix
// pipe runs; all loads of an item are issued back to back (one memory latency per item, not one per element).
template <int NT, bool TRANS_W, bool VEC, int KB, int THREADS>
__global__ __launch_bounds__(THREADS) void conv_apply(const float* __restrict__ feat, const float* __restrict__ W,
                                                  const float* __restrict__ bias, const int32_t* __restrict__ nbr,
                                                  int n_rows, int K, int Cred, int Cres, float* __restrict__ out, int mirror, const BnFuse bn) {
  // mirror: `nbr` is a submanifold layer's forward map read as its backward map -- column K-1-k holds offset k (rulebook.hip)
  // bn: batch statistics of the result for the BatchNorm behind this layer, gathered in the epilogue (bn_fuse.h)
  // KB > 1 (narrow layers, Cred <= 32: one chunk per offset): KB active offsets are staged per phase, which divides the
  // number of barrier-separated phases by KB and multiplies the loads in flight per workgroup by KB.
  // THREADS = 256 / 128 / 64 -> 64 / 32 / 16 output rows per workgroup (one 16-row MFMA slab per wave): small and
  // mid-size layers do not have enough 64-row tiles to hide memory latency by occupancy, so they get smaller workgroups.
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int TM = THREADS / 4;                   // rows per workgroup (shadows the file-level TM)
  constexpr int LDB = ldb_of(NT);
  constexpr int NB = NT * 512 / THREADS;            // weight-panel floats per thread and offset (KC * NT*16 / THREADS)
  float* As = (float*)smem;                         // [KB][TM][LDA]
  float* Bs = As + KB * TM * LDA;                   // [KB][KC][LDB]
  int32_t* s_nbr = (int32_t*)(Bs + KB * KC * LDB);  // [TM][K]
  int32_t* s_kact = s_nbr + TM * K;                 // [K] flags, then the compact list of active offsets
  int32_t* s_nact = s_kact + K;                     // [1]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int row0 = blockIdx.x * TM;
  const int n0 = blockIdx.y * (NT * 16);

  for (int e = tid; e < K; e += THREADS) s_kact[e] = 0;
  __syncthreads();
  {
    const long long gbase = (long long)row0 * K;
    const long long gend = (long long)n_rows * K;
    for (int e = tid; e < TM * K; e += THREADS) {
      const int kk = e % K;
      int v = (gbase + e < gend) ? nbr[mirror ? gbase + e - kk + (K - 1 - kk) : gbase + e] : -1;
      s_nbr[e] = v;
      if (v >= 0) s_kact[kk] = 1;
    }
  }
  __syncthreads();
  if (tid == 0) {
    int n = 0;
    for (int k = 0; k < K; ++k)
      if (s_kact[k]) s_kact[n++] = k;  // in-place compaction (n <= k)
    *s_nact = n;
  }
  __syncthreads();
  const int n_act = *s_nact;
  const int n_chunks = (KB > 1) ? 1 : (Cred + KC - 1) / KC;
  const int n_items = (KB > 1) ? (n_act + KB - 1) / KB : n_act * n_chunks;

  f32x4 acc[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) acc[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const int arow = wave * 16 + (lane & 15);
  const int kq = lane >> 4;

  float4 av[KB][2];   // VEC : 2 float4 of each A tile
  float as_[KB][8];   // !VEC: 8 scalars
  float bv[KB][NB];

  // cursors of the prefetch walk (one item ahead) and of the compute walk over the (offset, chunk) items: no integer division by
  // the runtime n_chunks inside the loop (conv_apply_split.hip)
  int pq = 0, pr = 0, cr = 0;
  auto prefetch = [&](int item) {
    const int pq_now = pq, pr_now = pr;
    if (KB == 1 && ++pr == n_chunks) { pr = 0; ++pq; }
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) {
      const int ai = (KB > 1) ? item * KB + kb : pq_now;
      const bool live = ai < n_act;
      const int k = live ? s_kact[ai] : 0;
      const int cc = (KB > 1) ? 0 : pr_now * KC;
      const int kc = min(KC, Cred - cc);
      const float* Wk = W + (size_t)k * Cred * Cres;
      if (VEC) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          int e = i * THREADS + tid, r = e / (KC / 4), c = (e % (KC / 4)) * 4;
          int j = live ? s_nbr[r * K + k] : -1;
          av[kb][i] = (j >= 0 && c < kc) ? *reinterpret_cast<const float4*>(feat + (size_t)j * Cred + cc + c) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
      } else {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          int e = i * THREADS + tid, r = e / KC, c = e % KC;
          int j = live ? s_nbr[r * K + k] : -1;
          as_[kb][i] = (j >= 0 && c < kc) ? feat[(size_t)j * Cred + cc + c] : 0.f;
        }
      }
#pragma unroll
      for (int i = 0; i < NB; ++i) {
        int e = i * THREADS + tid;
        if (!TRANS_W) {
          int r = e / (NT * 16), c = e % (NT * 16);
          bv[kb][i] = (live && r < kc && n0 + c < Cres) ? Wk[(size_t)(cc + r) * Cres + n0 + c] : 0.f;
        } else {  // Wk^T[r = co][c = ci] = W[k][ci][co]; consecutive threads read along co (contiguous)
          int c = e / KC, r = e % KC;
          bv[kb][i] = (live && r < kc && n0 + c < Cres) ? Wk[(size_t)(n0 + c) * Cred + cc + r] : 0.f;
        }
      }
    }
  };
  auto commit = [&]() {
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) {
      float* A = As + kb * TM * LDA;
      float* B = Bs + kb * KC * LDB;
      if (VEC) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          int e = i * THREADS + tid, r = e / (KC / 4), c = (e % (KC / 4)) * 4;
          float* d = A + r * LDA + c;
          d[0] = av[kb][i].x; d[1] = av[kb][i].y; d[2] = av[kb][i].z; d[3] = av[kb][i].w;
        }
      } else {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          int e = i * THREADS + tid, r = e / KC, c = e % KC;
          A[r * LDA + c] = as_[kb][i];
        }
      }
#pragma unroll
      for (int i = 0; i < NB; ++i) {
        int e = i * THREADS + tid;
        if (!TRANS_W) {
          int r = e / (NT * 16), c = e % (NT * 16);
          B[r * LDB + c] = bv[kb][i];
        } else {
          int c = e / KC, r = e % KC;
          B[r * LDB + c] = bv[kb][i];
        }
      }
    }
  };

  if (n_items > 0) prefetch(0);
  for (int item = 0; item < n_items; ++item) {
    const int cc = (KB > 1) ? 0 : cr * KC;
    if (KB == 1 && ++cr == n_chunks) cr = 0;
    const int kc = min(KC, Cred - cc);
    __syncthreads();  // the previous item's fragment reads are done
    commit();
    __syncthreads();
    if (item + 1 < n_items) prefetch(item + 1);  // in flight during the MFMAs below
    const int steps = (kc + 3) >> 2;
    const int cnt = (KB > 1) ? min(KB, n_act - item * KB) : 1;
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) {
      if (kb >= cnt) break;  // block-uniform; padded offsets hold zeros anyway
      const float* A = As + kb * TM * LDA;
      const float* B = Bs + kb * KC * LDB;
      for (int q = 0; q < steps; ++q) {
        float a = A[arow * LDA + q * 4 + kq];
        const float* bp = B + (q * 4 + kq) * LDB + (lane & 15);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bp[nt * 16], acc[nt], 0, 0, 0);
      }
    }
  }

  // ---- epilogue: C/D layout of 16x16: col = lane&15, row = (lane>>4)*4 + reg
  float vals[NT][4];
  bool valid[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) valid[r] = row0 + wave * 16 + kq * 4 + r < n_rows;
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int col = n0 + nt * 16 + (lane & 15);
    const float bv0 = (bias && col < Cres) ? bias[col] : 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = row0 + wave * 16 + kq * 4 + r;
      vals[nt][r] = bias ? (acc[nt][r] + bv0) : acc[nt][r];
      if (col < Cres && row < n_rows) out[(size_t)row * Cres + col] = vals[nt][r];
    }
  }
  if (bn.slots) {
    bn_fuse_wave<NT>(bn, vals, valid, n0, (int)((blockIdx.x * (THREADS / 64) + wave) & (bn.nslots - 1)));
    bn_fuse_finish(bn, (int*)smem, (double*)(smem + 16));
  }
}

// ------------------------------------------------------------------------------------------------------------
// Weight-stationary variant for narrow layers (Cred, Cres <= 32: K*Cred*Cres*4 <= 110 KB).  The profile of
// conv_apply on these layers is dominated by L2->LDS re-reads of the K weight panels by every 64-row workgroup and
// by the serial chain of K barrier-separated gather->MFMA phases per tile.  Here one persistent 16-wave workgroup
// per CU keeps ALL K panels resident in LDS (160 KB per CU on gfx950); every wave walks its own 16-row tiles with
// no workgroup barrier after the weight load; the A fragments are gathered STRAIGHT INTO REGISTERS in MFMA layout
// (lane (row, kq) loads channels q*4+kq), WS_KB offsets at a time, one group ahead of the MFMAs.
// Same summation order as conv_apply (offset ascending, channel ascending) -> same bits.
// ------------------------------------------------------------------------------------------------------------
constexpr int WS_WAVES = 16;
constexpr int WS_KB = 8;    // offsets per group of gathers (round 6: 4 -> 8 with WS_Q below: a 27-offset tile is 4 dependent round trips, not 7;
                            // 5 -> 32 at 210 K rows 80 -> 72 us, 4 -> 16 18.5 -> 16.4, same bits; 14 offsets per group measured 77)
constexpr int WS_Q = 2;     // k-steps of a row: the kernel is launched for reductions of <= 8 channels only (it sized its fragment registers for 32)

__host__ __device__ inline size_t ws_lds_bytes(int K, int Cred, int nt) {
  int crp = (Cred + 3) & ~3;
  return (size_t)K * crp * (nt * 16) * sizeof(float) + (size_t)WS_WAVES * 16 * K * sizeof(int32_t);
}

template <int NT, bool TRANS_W>
__global__ __launch_bounds__(WS_WAVES * 64) void conv_apply_ws(const float* __restrict__ feat, const float* __restrict__ W,
                                                              const float* __restrict__ bias, const int32_t* __restrict__ nbr,
                                                              int n_rows, int K, int Cred, int Cres, float* __restrict__ out, int mirror,
                                                              const BnFuse bn) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int LDW = NT * 16;
  constexpr int QMAX = WS_Q;
  const int crp = (Cred + 3) & ~3;
  float* Ws = (float*)smem;                                     // [K][crp][LDW], column XOR-swizzled by (row & 1) << 4 when NT == 2
  int32_t* nb_all = (int32_t*)(Ws + (size_t)K * crp * LDW);     // [WS_WAVES][16][K]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, kq = lane >> 4, lrow = lane & 15;
  int32_t* nb = nb_all + wave * 16 * K;

  // ---- resident weights: Ws[k][r][c] = Wk[r][c]  (Wk = W[k] or W[k]^T), zero padded; loads batched 8 deep
  {
    const int total = K * crp * LDW;
    for (int base = 0; base < total; base += WS_WAVES * 64 * 8) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        int e = base + u * WS_WAVES * 64 + tid;
        int c = e % LDW, r = (e / LDW) % crp, k = e / (LDW * crp);
        v[u] = 0.f;
        if (e < total && r < Cred && c < Cres) v[u] = TRANS_W ? W[((size_t)k * Cres + c) * Cred + r] : W[((size_t)k * Cred + r) * Cres + c];
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        int e = base + u * WS_WAVES * 64 + tid;
        if (e < total) {
          int c = e % LDW, r = (e / LDW) % crp, k = e / (LDW * crp);
          int cs = (NT == 2) ? (c ^ ((r & 1) << 4)) : c;
          Ws[((size_t)k * crp + r) * LDW + cs] = v[u];
        }
      }
    }
  }
  __syncthreads();

  const int n_tiles = (n_rows + 15) >> 4;
  const int steps = crp >> 2;
  for (int tile = blockIdx.x * WS_WAVES + wave; tile < n_tiles; tile += gridDim.x * WS_WAVES) {
    const int row0 = tile << 4;
    // neighbour rows of the tile (16 x K ints, contiguous) and the set of offsets that occur in it
    unsigned long long kmask = 0;
    {
      const long long gbase = (long long)row0 * K, gend = (long long)n_rows * K;
      for (int e = lane; e < 16 * K; e += 64) {
        const int kk = e % K;
        int v = (gbase + e < gend) ? nbr[mirror ? gbase + e - kk + (K - 1 - kk) : gbase + e] : -1;
        nb[e] = v;
        if (v >= 0) kmask |= 1ull << kk;
      }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) kmask |= __shfl_xor(kmask, o, 64);
    }
    __builtin_amdgcn_wave_barrier();
    f32x4 acc[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};

    float va[WS_KB][QMAX], vb[WS_KB][QMAX];  // current / next group of A fragments
    int ka[WS_KB], kb_[WS_KB];
    unsigned long long rem = kmask;
    // take the next WS_KB active offsets off the mask and issue their gathers into `v`
#define WS_FETCH(v, kk)                                                                             \
  _Pragma("unroll") for (int g = 0; g < WS_KB; ++g) {                                               \
    kk[g] = rem ? (__ffsll((long long)rem) - 1) : -1;                                               \
    rem &= rem - 1;                                                                                 \
    const int j = (kk[g] >= 0) ? nb[lrow * K + kk[g]] : -1;                                         \
    const float* src = feat + (size_t)(j >= 0 ? j : 0) * Cred + kq;                                 \
    _Pragma("unroll") for (int q = 0; q < QMAX; ++q)                                                \
        v[g][q] = (j >= 0 && q * 4 + kq < Cred && q < steps) ? src[q * 4] : 0.f;                    \
  }
#define WS_MATH(v, kk)                                                                              \
  _Pragma("unroll") for (int g = 0; g < WS_KB; ++g) {                                               \
    if (kk[g] < 0) break;                                                                           \
    const float* Wk = Ws + (size_t)kk[g] * crp * LDW;                                               \
    _Pragma("unroll") for (int q = 0; q < QMAX; ++q) {                                              \
      if (q >= steps) break;                                                                        \
      const int r = q * 4 + kq;                                                                     \
      _Pragma("unroll") for (int nt = 0; nt < NT; ++nt) {                                           \
        int c = nt * 16 + lrow;                                                                     \
        if (NT == 2) c ^= (r & 1) << 4;                                                             \
        acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(v[g][q], Wk[r * LDW + c], acc[nt], 0, 0, 0); \
      }                                                                                             \
    }                                                                                               \
  }
    WS_FETCH(va, ka)
    while (ka[0] >= 0) {
      WS_FETCH(vb, kb_)   // next group in flight during this group's MFMAs
      WS_MATH(va, ka)
      if (kb_[0] < 0) break;
      WS_FETCH(va, ka)
      WS_MATH(vb, kb_)
    }
#undef WS_FETCH
#undef WS_MATH
    float vals[NT][4];
    bool valid[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) valid[r] = row0 + kq * 4 + r < n_rows;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int col = nt * 16 + lrow;
      const float bv0 = (bias && col < Cres) ? bias[col] : 0.f;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = row0 + kq * 4 + r;
        vals[nt][r] = bias ? (acc[nt][r] + bv0) : acc[nt][r];
        if (col < Cres && row < n_rows) out[(size_t)row * Cres + col] = vals[nt][r];
      }
    }
    if (bn.slots) bn_fuse_wave<NT>(bn, vals, valid, 0, (int)(tile & (bn.nslots - 1)));
    __builtin_amdgcn_wave_barrier();
  }
  if (bn.slots) bn_fuse_finish(bn, (int*)smem, (double*)(smem + 16));
}

// dW partial: part[s][k][ci][co] = sum over the split's rows of feat[nbr[i][k]][ci] * dout[i][co]
// block = (k, split, tile of 64 Cin x NT*16 Cout); wave w owns dW rows [m0 + 16w, m0 + 16w + 16)
constexpr int WG_LDA = 64 + 16;
template <int NT, bool BF>
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
    // all loads of a thread are issued before the first LDS store (one memory latency per tile)
    if (((Cin | Cout) & 3) == 0) {
      float4 va[4], vd[NT];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        int e = i * 256 + tid, r = e >> 4, c = (e & 15) * 4;
        int jj = s_j[r];
        va[i] = (jj >= 0 && m0 + c < Cin) ? btc_ld4<BF>(feat, (size_t)jj * Cin + m0 + c) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
#pragma unroll
      for (int i = 0; i < NT; ++i) {
        int e = i * 256 + tid, r = e / (NT * 4), c = (e % (NT * 4)) * 4;
        vd[i] = (s_j[r] >= 0 && n0 + c < Cout) ? btc_ld4<BF>(dout, (size_t)(row0 + r) * Cout + n0 + c)
                                               : make_float4(0.f, 0.f, 0.f, 0.f);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        int e = i * 256 + tid, r = e >> 4, c = (e & 15) * 4;
        float* d = As + r * WG_LDA + c;
        d[0] = va[i].x; d[1] = va[i].y; d[2] = va[i].z; d[3] = va[i].w;
      }
#pragma unroll
      for (int i = 0; i < NT; ++i) {
        int e = i * 256 + tid, r = e / (NT * 4), c = (e % (NT * 4)) * 4;
        float* d = Ds + r * LDB + c;
        d[0] = vd[i].x; d[1] = vd[i].y; d[2] = vd[i].z; d[3] = vd[i].w;
      }
    } else {
      float va[16], vd[NT * 4];
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        int e = i * 256 + tid, r = e >> 6, c = e & 63;
        int jj = s_j[r];
        va[i] = (jj >= 0 && m0 + c < Cin) ? btc_ld1<BF>(feat, (size_t)jj * Cin + m0 + c) : 0.f;
      }
#pragma unroll
      for (int i = 0; i < NT * 4; ++i) {
        int e = i * 256 + tid, r = e / (NT * 16), c = e % (NT * 16);
        vd[i] = (s_j[r] >= 0 && n0 + c < Cout) ? btc_ld1<BF>(dout, (size_t)(row0 + r) * Cout + n0 + c) : 0.f;
      }
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        int e = i * 256 + tid, r = e >> 6, c = e & 63;
        As[r * WG_LDA + c] = va[i];
      }
#pragma unroll
      for (int i = 0; i < NT * 4; ++i) {
        int e = i * 256 + tid, r = e / (NT * 16), c = e % (NT * 16);
        Ds[r * LDB + c] = vd[i];
      }
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

// conv_wgrad_partial for channel counts that are multiples of 4, with
//  * the next tile's loads in flight during this tile's MFMAs: the map column entry of tile t + 1 is read while tile t is being
//    staged, its gathered rows and dOut rows are requested right after tile t's tiles are visible in LDS and sit in registers
//    until tile t's MFMAs are done (same LDS footprint, same two barriers per tile);
//  * per-tile packing: only the rows of the tile that HAVE the offset are staged, packed to the front of the LDS tiles (a wave
//    ballot over the column gives every live row its slot), and the reduction runs over ceil(m / 4) 4-row steps instead of 16.
//    Both operands are gathered per offset here anyway, so packing costs no indirection in the MFMA loop.  At the wide layers
//    that land on this kernel (128 -> 128, 256 -> 128 on the 8x-downsampled level) 55 % of the (row, offset) slots are live.
// Skipped terms are exact zeros; dW differs from the unpacked sum only in how rows group into 4-row MFMA steps.
template <int NT, bool BF>
__global__ __launch_bounds__(256) void conv_wgrad_partial_p(const float* __restrict__ feat, const float* __restrict__ dout,
                                                            const int32_t* __restrict__ nbr, int n_out, int K, int Cin, int Cout,
                                                            int tiles_per_split, int n_cblk, float* __restrict__ part) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int LDB = ldb_of(NT);
  float* As = (float*)smem;       // [TM][WG_LDA]  gathered input rows (packed), 64 channels
  float* Ds = As + TM * WG_LDA;   // [TM][LDB]     dout rows (packed), NT*16 channels
  int32_t* s_src = (int32_t*)(Ds + TM * LDB);  // [2][TM] input row of packed slot t (-1: padding of the last 4-row step)
  int32_t* s_dst = s_src + 2 * TM;             // [2][TM] output row of packed slot t
  int32_t* s_m = s_dst + 2 * TM;               // [2] live rows of the tile

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

  float4 va[4], vd[NT];
  auto load_j = [&](int t) {   // wave 0: map column entry of row `lane` of tile t
    const int row = t * TM + lane;
    return (wave == 0 && t < t_end && row < n_out) ? nbr[(size_t)row * K + k] : -1;
  };
  auto pack = [&](int t, int j, int b) {   // wave 0: packed slots of tile t into buffer b
    if (wave != 0) return;
    const unsigned long long live = __ballot(j >= 0);
    const int m = __popcll(live);
    if (j >= 0) {
      const int slot = __popcll(live & ((1ull << lane) - 1ull));
      s_src[b * TM + slot] = j;
      s_dst[b * TM + slot] = t * TM + lane;
    }
    if (lane >= m && lane < ((m + 3) & ~3)) {
      s_src[b * TM + lane] = -1;
      s_dst[b * TM + lane] = -1;
    }
    if (lane == 0) s_m[b] = m;
  };
  auto load_tile = [&](int b) {
    const int m4 = (s_m[b] + 3) & ~3;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int e = i * 256 + tid, r = e >> 4, c = (e & 15) * 4;
      const int jj = r < m4 ? s_src[b * TM + r] : -1;
      va[i] = (jj >= 0 && m0 + c < Cin) ? btc_ld4<BF>(feat, (size_t)jj * Cin + m0 + c) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int i = 0; i < NT; ++i) {
      const int e = i * 256 + tid, r = e / (NT * 4), c = (e % (NT * 4)) * 4;
      const int ro = r < m4 ? s_dst[b * TM + r] : -1;
      vd[i] = (ro >= 0 && n0 + c < Cout) ? btc_ld4<BF>(dout, (size_t)ro * Cout + n0 + c) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  auto store_tile = [&](int m4) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int e = i * 256 + tid, r = e >> 4, c = (e & 15) * 4;
      if (r < m4) {
        float* d = As + r * WG_LDA + c;
        d[0] = va[i].x; d[1] = va[i].y; d[2] = va[i].z; d[3] = va[i].w;
      }
    }
#pragma unroll
    for (int i = 0; i < NT; ++i) {
      const int e = i * 256 + tid, r = e / (NT * 4), c = (e % (NT * 4)) * 4;
      if (r < m4) {
        float* d = Ds + r * LDB + c;
        d[0] = vd[i].x; d[1] = vd[i].y; d[2] = vd[i].z; d[3] = vd[i].w;
      }
    }
  };

  // prologue: the first tile's column and loads, the second tile's column
  int m4 = 0, jn = -1;
  if (t_begin < t_end) {
    pack(t_begin, load_j(t_begin), 0);
    __syncthreads();
    m4 = (s_m[0] + 3) & ~3;
    if (m4) load_tile(0);
    jn = load_j(t_begin + 1);
  }
  for (int t = t_begin; t < t_end; ++t) {
    const int cur = (t - t_begin) & 1;
    if (m4) store_tile(m4);            // tile t: registers -> LDS (the previous tile's MFMAs ended at the barrier below)
    pack(t + 1, jn, cur ^ 1);          // tile t + 1's packed slots
    __syncthreads();                   // tile t in LDS; everyone sees tile t + 1's slots
    const int m4_next = (s_m[cur ^ 1] + 3) & ~3;
    if (m4_next) load_tile(cur ^ 1);   // in flight during the MFMAs below
    jn = load_j(t + 2);
    if (m4 && wave_live) {
      const int steps = m4 >> 2;
      const float* ap = As + kq * WG_LDA + wave * 16 + (lane & 15);
      const float* bp = Ds + kq * LDB + (lane & 15);
#pragma unroll 4
      for (int q = 0; q < steps; ++q) {
        const float a = ap[q * 4 * WG_LDA];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bp[q * 4 * LDB + nt * 16], acc[nt], 0, 0, 0);
      }
    }
    __syncthreads();   // MFMAs of tile t done: the LDS tiles may be overwritten
    m4 = m4_next;
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

// ------------------------------------------------------------------------------------------------------------
// Row-stationary weight gradient for the large-N / small-C layers (the occupancy branch: up to 210 K rows at 32
// channels).  A persistent workgroup walks row tiles; per tile the dOut rows and the neighbour-map rows are loaded
// ONCE (coalesced, row-major) and the K offsets are processed in phases of KB gathered input tiles; the whole
// dW slab of the workgroup's offset group (PH*KB offsets x Cin x Cout) lives in MFMA accumulators for the entire
// walk and is written out once.  (The offset-major kernel below re-reads dOut K times and reads the map column-wise.)
//   MT, NT : 16-wide tiles of Cin / Cout;  KB : offsets per LDS phase;  PH : phases per offset group
// ------------------------------------------------------------------------------------------------------------
template <int MT, int NT, int KB, int PH, bool BF>
__global__ __launch_bounds__(256) void conv_wgrad_rows(const float* __restrict__ feat, const float* __restrict__ dout,
                                                       const int32_t* __restrict__ nbr, const int32_t* __restrict__ order, int n_out, int K,
                                                       int Cin, int Cout, float* __restrict__ part, int swap, int dbg) {
  // order (optional, row_order.hip): tile slot t works on map row order[t]; rows with the same offsets share tiles, so fewer
  // offset phases per tile are live.  dW is the fp32 sum over rows in walk order.
  // Naming follows the un-swapped case: `feat` = gathered operand (Cin channels, via the map), `dout` = contiguous
  // operand (Cout channels), one tile per 64 map rows.  swap = 1: the walk is over the INPUT rows instead (map =
  // nbr_in, gathered = dOut, contiguous = features) -- used when the layer has far fewer input than output rows
  // (transposed / dilating convs) -- and the slab is written transposed so that dW keeps the [K][Cin][Cout] layout.
  constexpr int TPP = KB * MT * NT / 4;  // accumulator tiles per wave per phase
  static_assert(KB * MT * NT % 4 == 0, "phase tiles must split evenly over the 4 waves");
  constexpr int LDA = ldb_of(MT), LDB = ldb_of(NT);
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* As = (float*)smem;                       // [KB][TM][LDA]
  float* Ds = As + KB * TM * LDA;                 // [TM][LDB]
  int32_t* s_nbr = (int32_t*)(Ds + TM * LDB);     // [TM][K]
  int32_t* s_kact = s_nbr + TM * K;               // [K]
  int32_t* s_row = s_kact + K;                    // [TM] map row of each tile slot, -1 past the end

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, kq = lane >> 4;
  const int kg0 = blockIdx.y * (PH * KB);         // first offset of this workgroup's group
  const int n_tiles = (n_out + TM - 1) / TM;

  f32x4 acc[PH * TPP];
#pragma unroll
  for (int t = 0; t < PH * TPP; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};

  for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const int row0 = tile * TM;
    for (int e = tid; e < K; e += 256) s_kact[e] = 0;
    if (tid < TM) s_row[tid] = (row0 + tid < n_out) ? (order ? order[row0 + tid] : row0 + tid) : -1;
    __syncthreads();
    for (int e = tid; e < TM * K; e += 256) {
      const int rloc = e / K, kk = e - rloc * K;
      const int gr = s_row[rloc];
      const int v = gr >= 0 ? nbr[(long long)gr * K + kk] : -1;
      s_nbr[e] = v;
      if (v >= 0) s_kact[kk] = 1;
    }
    // dOut tile: all loads of a thread are issued before the first LDS store (one latency, not one per element)
    if ((Cout & 3) == 0) {
      float4 v[NT];
#pragma unroll
      for (int i = 0; i < NT; ++i) {
        int e = i * 256 + tid, r = e / (NT * 4), c = (e % (NT * 4)) * 4;
        const int gr = s_row[r];
        v[i] = (gr >= 0 && c < Cout) ? btc_ld4<BF>(dout, (size_t)gr * Cout + c) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
#pragma unroll
      for (int i = 0; i < NT; ++i) {
        int e = i * 256 + tid, r = e / (NT * 4), c = (e % (NT * 4)) * 4;
        float* d = Ds + r * LDB + c;
        d[0] = v[i].x; d[1] = v[i].y; d[2] = v[i].z; d[3] = v[i].w;
      }
    } else {
      float v[NT * 4];
#pragma unroll
      for (int i = 0; i < NT * 4; ++i) {
        int e = i * 256 + tid, r = e / (NT * 16), c = e % (NT * 16);
        const int gr = s_row[r];
        v[i] = (gr >= 0 && c < Cout) ? btc_ld1<BF>(dout, (size_t)gr * Cout + c) : 0.f;
      }
#pragma unroll
      for (int i = 0; i < NT * 4; ++i) {
        int e = i * 256 + tid, r = e / (NT * 16), c = e % (NT * 16);
        Ds[r * LDB + c] = v[i];
      }
    }
    __syncthreads();
#pragma unroll
    for (int p = 0; p < PH; ++p) {
      const int k0 = kg0 + p * KB;
      int any = 0;
#pragma unroll
      for (int kb = 0; kb < KB; ++kb) any |= (k0 + kb < K) ? s_kact[k0 + kb] : 0;
      if (!any) continue;  // block-uniform
      // gather KB input tiles; loads batched in registers as above
      if ((Cin & 3) == 0) {
        float4 v[KB * MT];
#pragma unroll
        for (int i = 0; i < KB * MT; ++i) {
          int e = i * 256 + tid, c = (e % (MT * 4)) * 4, r = (e / (MT * 4)) % TM, kb = e / (MT * 4 * TM);
          int j = (k0 + kb < K) ? s_nbr[r * K + k0 + kb] : -1;
          if (dbg & 8) j = -1;  // timing experiments only (BTC_TUNE_APPLY_DEBUG, tools/wgrad_bench.py): no gathers
          v[i] = (j >= 0 && c < Cin) ? btc_ld4<BF>(feat, (size_t)j * Cin + c) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int i = 0; i < KB * MT; ++i) {
          int e = i * 256 + tid, c = (e % (MT * 4)) * 4, r = (e / (MT * 4)) % TM, kb = e / (MT * 4 * TM);
          float* d = As + (kb * TM + r) * LDA + c;
          d[0] = v[i].x; d[1] = v[i].y; d[2] = v[i].z; d[3] = v[i].w;
        }
      } else {
        float v[KB * MT * 4];
#pragma unroll
        for (int i = 0; i < KB * MT * 4; ++i) {
          int e = i * 256 + tid, c = e % (MT * 16), r = (e / (MT * 16)) % TM, kb = e / (MT * 16 * TM);
          int j = (k0 + kb < K) ? s_nbr[r * K + k0 + kb] : -1;
          v[i] = (j >= 0 && c < Cin) ? btc_ld1<BF>(feat, (size_t)j * Cin + c) : 0.f;
        }
#pragma unroll
        for (int i = 0; i < KB * MT * 4; ++i) {
          int e = i * 256 + tid, c = e % (MT * 16), r = (e / (MT * 16)) % TM, kb = e / (MT * 16 * TM);
          As[(kb * TM + r) * LDA + c] = v[i];
        }
      }
      __syncthreads();
#pragma unroll
      for (int q = 0; q < TPP; ++q) {
        if (dbg & 4) continue;       // timing experiments only: no MFMA phase
        const int l = q * 4 + wave;  // phase-local tile: (kb, mt, nt)
        const int nt = l % NT, mt = (l / NT) % MT, kb = l / (NT * MT);
        const float* ap = As + (size_t)kb * TM * LDA + mt * 16 + (lane & 15);
        const float* bp = Ds + nt * 16 + (lane & 15);
        f32x4 a4 = acc[p * TPP + q];
#pragma unroll 4
        for (int s = 0; s < TM / 4; ++s)
          a4 = __builtin_amdgcn_mfma_f32_16x16x4f32(ap[(s * 4 + kq) * LDA], bp[(s * 4 + kq) * LDB], a4, 0, 0, 0);
        acc[p * TPP + q] = a4;
      }
      __syncthreads();
    }
  }
  // write this workgroup's slab: part[blockIdx.x][k][ci][co]
  float* P = part + (size_t)blockIdx.x * K * Cin * Cout;
#pragma unroll
  for (int p = 0; p < PH; ++p)
#pragma unroll
    for (int q = 0; q < TPP; ++q) {
      const int l = q * 4 + wave;
      const int nt = l % NT, mt = (l / NT) % MT, kb = l / (NT * MT);
      const int k = kg0 + p * KB + kb;
      const int co = nt * 16 + (lane & 15);
      if (k >= K || co >= Cout) continue;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        int ci = mt * 16 + kq * 4 + r;
        if (ci < Cin) {
          if (!swap) P[((size_t)k * Cin + ci) * Cout + co] = acc[p * TPP + q][r];
          else P[((size_t)k * Cout + co) * Cin + ci] = acc[p * TPP + q][r];  // here ci indexes dOut channels, co feature channels
        }
      }
    }
}

// ------------------------------------------------------------------------------------------------------------
// conv_wgrad_rows, software-pipelined (gathered-operand channel counts that are multiples of 4).  tools/wgrad_bench.py on the
// kernel above: removing the MFMA phase halves its time, removing the gathers changes nothing -- a workgroup alternates
// between a staging round (issue the loads, wait one L2 / HBM latency, store to LDS, barrier: ~1.8 us) and an MFMA phase of
// about the same length, and only the other workgroup of the CU fills the holes.  Here
//  * the loads of item g + 1 (the gathered rows of the next phase; the map rows of the tile after next) are issued right
//    after item g's barrier and land in registers while item g's MFMAs run; they are stored to the OTHER LDS buffer at the
//    top of item g + 1: one barrier per item, no exposed load latency;
//  * the contiguous operand never goes through LDS: 4 % NT == 0, so a wave's accumulator tiles all share ONE 16-column block
//    (nt = wave % NT), and its MFMA B fragments for the 16 4-row steps of a tile are 16 registers, loaded once per tile
//    (prefetched during the previous tile's last phase) and reused by every offset of the group.  Half the LDS reads of the
//    MFMA loop, and the LDS footprint drops to the double-buffered gather tile: 41-64 KB, two to three workgroups per CU.
// Same tiles, same 4-row MFMA steps, same order over rows as the kernel above.  Items are all (tile, phase) pairs: a phase
// none of whose offsets occurs in the tile costs zeros -- at 64-row tiles that is < 10 % of the phases of the layers this
// kernel takes.
//   LDS: As[2][KB][TM][LDA] | s_nbr[3][TM][NOFF] (the group's offsets only) | s_row[3][TM]
// ------------------------------------------------------------------------------------------------------------
template <int MT, int NT, int KB, int PH, bool BF>
__global__ __launch_bounds__(256) void conv_wgrad_rows_p(const float* __restrict__ feat, const float* __restrict__ dout,
                                                         const int32_t* __restrict__ nbr, const int32_t* __restrict__ order, int n_out, int K,
                                                         int Cin, int Cout, float* __restrict__ part, int swap) {
  constexpr int TPP = KB * MT * NT / 4;
  static_assert(KB * MT * NT % 4 == 0, "phase tiles must split evenly over the 4 waves");
  static_assert(4 % NT == 0, "a wave's tiles must share one column block");
  constexpr int LDA = ldb_of(MT);
  constexpr int NOFF = PH * KB;
  constexpr int NV = (TM * NOFF + 255) / 256;   // map entries per thread and tile
  constexpr int NS = TM / 4;                    // 4-row MFMA steps per tile
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* As = (float*)smem;                              // [2][KB][TM][LDA]
  int32_t* s_nbr = (int32_t*)(As + 2 * KB * TM * LDA);   // [3][TM][NOFF]
  int32_t* s_row = s_nbr + 3 * TM * NOFF;                // [3][TM]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, kq = lane >> 4;
  const int kg0 = blockIdx.y * NOFF;
  const int n_tiles = (n_out + TM - 1) / TM;
  const int nt_wg = ((int)blockIdx.x < n_tiles) ? (n_tiles - 1 - (int)blockIdx.x) / (int)gridDim.x + 1 : 0;   // tiles of this workgroup
  const int bcol = (wave % NT) * 16 + (lane & 15);   // this lane's column of the contiguous operand

  f32x4 acc[PH * TPP];
#pragma unroll
  for (int t = 0; t < PH * TPP; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};

  int nv[NV], nrow = -1;            // the map rows (this group's offsets) and row ids of a tile, in flight
  float4 gv[KB * MT];               // the gathered rows of an item, in flight
  float bcur[NS], bnext[NS];        // B fragments of the tile / of the next tile (in flight)

  auto load_map = [&](int i) {      // tile i of this workgroup -> registers
    const int row0 = (blockIdx.x + i * gridDim.x) * TM;
    if (tid < TM) nrow = (row0 + tid < n_out) ? (order ? order[row0 + tid] : row0 + tid) : -1;
#pragma unroll
    for (int u = 0; u < NV; ++u) {
      const int e = u * 256 + tid, r = e / NOFF, o = e - r * NOFF;
      int v = -1;
      if (e < TM * NOFF && row0 + r < n_out && kg0 + o < K) {
        const int gr = order ? order[row0 + r] : row0 + r;
        v = nbr[(long long)gr * K + kg0 + o];
      }
      nv[u] = v;
    }
  };
  auto store_map = [&](int i) {
    int32_t* dn = s_nbr + (i % 3) * TM * NOFF;
#pragma unroll
    for (int u = 0; u < NV; ++u) {
      const int e = u * 256 + tid;
      if (e < TM * NOFF) dn[e] = nv[u];
    }
    if (tid < TM) s_row[(i % 3) * TM + tid] = nrow;
  };
  auto load_b = [&](int i) {        // B fragments of tile i (its row ids are in LDS): row 4 s + kq, column bcol
    const int32_t* rows = s_row + (i % 3) * TM + kq;
#pragma unroll
    for (int s2 = 0; s2 < NS; ++s2) {
      const int gr = rows[s2 * 4];
      bnext[s2] = (gr >= 0 && bcol < Cout) ? btc_ld1<BF>(dout, (size_t)gr * Cout + bcol) : 0.f;
    }
  };
  // bf16 activations (the launcher guarantees Cin % 8 == 0 for these instances): 16-byte loads of 8 channels -- half the load
  // instructions of the 4-channel walk and half its staging registers (216 -> 152 VGPRs for the 64 x 64 shape: a third workgroup
  // per CU) -- widened to fp32 on the way into LDS
  constexpr int UPR8 = MT * 2;                             // 8-channel units per gathered row
  constexpr int NU8 = BF ? (KB * TM * UPR8 + 255) / 256 : 1;   // units per thread and item
  uint4 gq[NU8];
  constexpr bool wide = BF;
  auto load_g = [&](int i, int p) { // the gathered rows of phase p of tile i
    const int32_t* mp = s_nbr + (i % 3) * TM * NOFF + p * KB;
    if (wide) {
#pragma unroll
      for (int u = 0; u < NU8; ++u) {
        const int e = u * 256 + tid, c8 = e % UPR8, r = (e / UPR8) % TM, kb = e / (UPR8 * TM);
        uint4 v = make_uint4(0u, 0u, 0u, 0u);
        if (e < KB * TM * UPR8) {
          const int j = mp[r * NOFF + kb];
          if (j >= 0 && c8 * 8 < Cin) v = *reinterpret_cast<const uint4*>(reinterpret_cast<const unsigned short*>(feat) + (size_t)j * Cin + c8 * 8);
        }
        gq[u] = v;
      }
      return;
    }
#pragma unroll
    for (int u = 0; u < KB * MT; ++u) {
      const int e = u * 256 + tid, c = (e % (MT * 4)) * 4, r = (e / (MT * 4)) % TM, kb = e / (MT * 4 * TM);
      const int j = mp[r * NOFF + kb];
      gv[u] = (j >= 0 && c < Cin) ? btc_ld4<BF>(feat, (size_t)j * Cin + c) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  auto store_g = [&](int buf) {
    float* A = As + buf * KB * TM * LDA;
    if (wide) {
#pragma unroll
      for (int u = 0; u < NU8; ++u) {
        const int e = u * 256 + tid, c8 = e % UPR8, r = (e / UPR8) % TM, kb = e / (UPR8 * TM);
        if (e < KB * TM * UPR8) {
          float* d = A + (kb * TM + r) * LDA + c8 * 8;
          const unsigned w[4] = {gq[u].x, gq[u].y, gq[u].z, gq[u].w};
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            d[2 * q] = __uint_as_float(w[q] << 16);
            d[2 * q + 1] = __uint_as_float(w[q] & 0xffff0000u);
          }
        }
      }
      return;
    }
#pragma unroll
    for (int u = 0; u < KB * MT; ++u) {
      const int e = u * 256 + tid, c = (e % (MT * 4)) * 4, r = (e / (MT * 4)) % TM, kb = e / (MT * 4 * TM);
      float* d = A + (kb * TM + r) * LDA + c;
      d[0] = gv[u].x; d[1] = gv[u].y; d[2] = gv[u].z; d[3] = gv[u].w;
    }
  };

  if (nt_wg > 0) {
    load_map(0);
    store_map(0);
    if (nt_wg > 1) {
      load_map(1);
      store_map(1);
    }
    __syncthreads();
    load_b(0);
    load_g(0, 0);
  }
  int buf = 0;
  for (int i = 0; i < nt_wg; ++i) {
#pragma unroll
    for (int p = 0; p < PH; ++p) {
      store_g(buf);
      if (p == 0) {
#pragma unroll
        for (int s2 = 0; s2 < NS; ++s2) bcur[s2] = bnext[s2];
      }
      // the map rows that were loaded during the previous item: tile i + 2 (PH > 1: loaded at this tile's phase 0) or
      // tile i + 1 (PH == 1: loaded during tile i - 1); tiles 0 and 1 come from the prologue
      if (PH > 1 ? (p == 1 && i + 2 < nt_wg) : (i >= 1 && i + 1 < nt_wg)) store_map(PH > 1 ? i + 2 : i + 1);
      __syncthreads();
      // ---- loads for the next item, in flight during this item's MFMAs
      if (p + 1 < PH) {
        load_g(i, p + 1);
      } else if (i + 1 < nt_wg) {
        load_b(i + 1);
        load_g(i + 1, 0);
      }
      if (PH > 1 ? (p == 0 && i + 2 < nt_wg) : (i + 2 < nt_wg)) load_map(i + 2);
      // ---- MFMAs of item (i, p)
      const float* A = As + buf * KB * TM * LDA;
#pragma unroll
      for (int q = 0; q < TPP; ++q) {
        const int l = q * 4 + wave;  // phase-local tile: (kb, mt, nt), nt == wave % NT
        const int mt = (l / NT) % MT, kb = l / (NT * MT);
        const float* ap = A + (size_t)kb * TM * LDA + mt * 16 + (lane & 15) + kq * LDA;
        f32x4 a4 = acc[p * TPP + q];
#pragma unroll
        for (int s2 = 0; s2 < NS; ++s2) a4 = __builtin_amdgcn_mfma_f32_16x16x4f32(ap[s2 * 4 * LDA], bcur[s2], a4, 0, 0, 0);
        acc[p * TPP + q] = a4;
      }
      buf ^= 1;
    }
  }
  float* P = part + (size_t)blockIdx.x * K * Cin * Cout;
#pragma unroll
  for (int p = 0; p < PH; ++p)
#pragma unroll
    for (int q = 0; q < TPP; ++q) {
      const int l = q * 4 + wave;
      const int nt = l % NT, mt = (l / NT) % MT, kb = l / (NT * MT);
      const int k = kg0 + p * KB + kb;
      const int co = nt * 16 + (lane & 15);
      if (k >= K || co >= Cout) continue;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        int ci = mt * 16 + kq * 4 + r;
        if (ci < Cin) {
          if (!swap) P[((size_t)k * Cin + ci) * Cout + co] = acc[p * TPP + q][r];
          else P[((size_t)k * Cout + co) * Cin + ci] = acc[p * TPP + q][r];
        }
      }
    }
}

// dW[e] = sum_s part[s][e] in slab order (deterministic); 8 loads in flight per thread
__global__ __launch_bounds__(256) void wgrad_reduce(const float* __restrict__ part, int S, long long count,
                                                    float* __restrict__ dW) {
  long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= count) return;
  float s = 0.f;
  int q = 0;
  // (a chain of S dependent additions per element, bound by the round trips of its loads: 16 in flight, then 8; the order of the
  // additions is the slab order either way)
  for (; q + 16 <= S; q += 16) {
    float v[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) v[u] = part[(size_t)(q + u) * count + e];
#pragma unroll
    for (int u = 0; u < 16; ++u) s += v[u];
  }
  for (; q + 8 <= S; q += 8) {
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = part[(size_t)(q + u) * count + e];
#pragma unroll
    for (int u = 0; u < 8; ++u) s += v[u];
  }
  for (; q < S; ++q) s += part[(size_t)q * count + e];
  dW[e] = s;
}

// the slab reductions of MANY layers in one launch (btc_wgrad_reduce_multi: every weight gradient of a backward pass whose dW nobody
// reads before the side stream's join): block b works on job j with block0[j] <= b < block0[j + 1]; same sums, same order as wgrad_reduce
struct ReduceJobs {
  const float* part[BTC_WGRAD_MULTI_MAX];
  float* dW[BTC_WGRAD_MULTI_MAX];
  long long count[BTC_WGRAD_MULTI_MAX];
  int S[BTC_WGRAD_MULTI_MAX];
  int block0[BTC_WGRAD_MULTI_MAX + 1];
  int n;
};

__global__ __launch_bounds__(256) void wgrad_reduce_multi(const ReduceJobs jobs) {
  int j = 0;
  while (j + 1 < jobs.n && (int)blockIdx.x >= jobs.block0[j + 1]) ++j;   // (uniform: scalar loop over <= 64 entries)
  const long long e = (long long)((int)blockIdx.x - jobs.block0[j]) * 256 + threadIdx.x;
  const long long count = jobs.count[j];
  if (e >= count) return;
  const float* __restrict__ part = jobs.part[j];
  const int S = jobs.S[j];
  float s = 0.f;
  int q = 0;
  // (a chain of S dependent additions per element, bound by the round trips of its loads: 16 in flight, then 8; the order of the
  // additions is the slab order either way)
  for (; q + 16 <= S; q += 16) {
    float v[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) v[u] = part[(size_t)(q + u) * count + e];
#pragma unroll
    for (int u = 0; u < 16; ++u) s += v[u];
  }
  for (; q + 8 <= S; q += 8) {
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = part[(size_t)(q + u) * count + e];
#pragma unroll
    for (int u = 0; u < 8; ++u) s += v[u];
  }
  for (; q < S; ++q) s += part[(size_t)q * count + e];
  jobs.dW[j][e] = s;
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

template <int NT, bool TRANS_W, int THREADS>
void launch_apply_t(dim3 grid, size_t lds, hipStream_t stream, bool vec, const float* feat, const float* W, const float* bias,
                    const int32_t* nbr, int n_rows, int K, int Cred, int Cres, float* out, int mirror, const BnFuse& bn) {
  if (vec) conv_apply<NT, TRANS_W, true, 1, THREADS><<<grid, THREADS, lds, stream>>>(feat, W, bias, nbr, n_rows, K, Cred, Cres, out, mirror, bn);
  else conv_apply<NT, TRANS_W, false, 1, THREADS><<<grid, THREADS, lds, stream>>>(feat, W, bias, nbr, n_rows, K, Cred, Cres, out, mirror, bn);
}

template <bool TRANS_W>
int launch_apply(const float* feat, const float* W, const float* bias, const int32_t* nbr, int n_rows, int K, int Cred,
                 int Cres, float* out, hipStream_t stream, bool bf = false, const int32_t* order = nullptr, int mirror = 0,
                 const BnFuse* bn_ = nullptr) {
  const BnFuse bn = bn_ ? *bn_ : btc_bn_fuse_none();   // batch statistics of the result in the epilogue (every kernel family below has it)
  // order: optional row-order hint (row_order.hip); only the LDS-DMA kernel tiles by it, the others ignore it (same results)
  // bf: feat / out are bfloat16 (passed through the float* parameters); only the LDS-DMA kernel has that variant
  if (n_rows <= 0) return BTC_OK;
  BTC_CHECK_ARG(!bf || btc_apply_glds_supported(K, Cred, Cres),
                "bf16 activations need channel counts that are multiples of 16 (K=%d, %d -> %d): convert to fp32", K, Cred, Cres);
  int nt = Cres <= 16 ? 1 : (Cres <= 32 ? 2 : (Cres <= 64 ? 4 : 8));
  const int t_kernel = btc_tune_get(BTC_TUNE_APPLY_KERNEL), t_nt = btc_tune_get(BTC_TUNE_APPLY_NT),
            t_xcd = btc_tune_get(BTC_TUNE_APPLY_XCD);
  // LDS-DMA pipelined kernel (conv_apply_glds.hip) for every layer whose channel counts are multiples of 16, except the
  // 200 K-row occupancy-branch layers where the register-staged kernel below measures 5-10 % faster.  Wave shapes from
  // tools/conv_bench.py on MI355X (us per launch, register-staged -> LDS-DMA): 14 K rows 64->64: 80 -> 53, 128->128: 225 -> 158,
  // 256->128: 456 -> 309; 3 K rows 64->64: 65 -> 40; 30 K rows 64->64: 119 -> 99.
  // (the 100 K-row exception is about the 32-channel occupancy-branch layers; wide layers -- the ROI head's 128-channel
  // micro-scene pyramid at 130 K rows -- stay on the LDS-DMA kernel at any row count)
  if (bf || (t_kernel != 1 && btc_apply_glds_supported(K, Cred, Cres) && (t_kernel == 2 || n_rows < 100000 || (Cred >= 64 && Cres >= 64)))) {
    int shape, kc = (Cred % 64 == 0) ? 64 : ((Cred % 32 == 0) ? 32 : 16);
    // Wave shapes from a row-count sweep on MI355X (fp32, K = 27, a 7.7-pairs-per-row SubM map cut to n rows; us per launch).
    // The time of a shape is a staircase in n with a step at every multiple of 256 CUs x TM rows, so the best TM depends on
    // where n falls: 256 -> 128 at 4 K rows 296 (424) vs 148 (222), at 10 K rows 301 vs 196 (242), at 16 K rows 306 (424, one
    // full round) vs 353, at 18 K rows 467 (424: a second, nearly empty round) vs 383 (242); 64 -> 64 at 10 K rows 52 (422)
    // vs 37 (141), from 18 K rows on 80-133 (422) vs 67-122 (241); 32 -> 64 from 18 K rows on 50-76 (422) vs 41-71 (222).
    if (bf) {
      if (Cres % 128 == 0) shape = 424;                       // 64 rows x 128 columns, 8 waves
      else if (Cres % 64 == 0) shape = n_rows < 8192 ? 141 : 422;  // few rows: 16-row workgroups, 4 waves across the columns
      else if (Cres % 32 == 0) shape = 221;
      else shape = 411;
    } else if (Cres % 128 == 0) {
      shape = n_rows < 7000 ? 222 : (n_rows < 13000 ? 242 : (n_rows <= 16384 ? 424 : (n_rows < 19500 ? 242 : 424)));
    } else if (Cres % 64 == 0) {
      if (kc == 64) shape = n_rows < 11000 ? 141 : 241;
      else shape = n_rows < 13000 ? 141 : 222;
    } else if (Cres % 32 == 0) {
      shape = 221;
    } else {
      shape = 411;
    }
    if (t_nt > 8 && !bf) {  // tuning run: BTC_TUNE_APPLY_NT carries the wave shape WR*100 + WC*10 + NTW
      int wc = (t_nt / 10) % 10, ntw = t_nt % 10;
      while (ntw > 1 && Cres % (16 * wc * ntw)) ntw >>= 1;
      while (wc > 1 && Cres % (16 * wc * ntw)) wc >>= 1;
      const int cand = (t_nt / 100) * 100 + wc * 10 + ntw;
      if (btc_apply_glds_has_shape(cand)) shape = cand;
    }
    const int t_kc = btc_tune_get(BTC_TUNE_APPLY_KC);
    if (t_kc && Cred % t_kc == 0) kc = t_kc;
    while (kc > 16 && btc_apply_glds_lds_bytes(shape, kc, K, bf) > 160 * 1024) kc >>= 1;  // 3-stage ring + map tile
    return btc_launch_apply_glds(TRANS_W, shape, kc, (t_xcd == 2 ? 1 : 0) | (mirror ? 2 : 0), bf, feat, W, bias, nbr, order, n_rows, K, Cred, Cres, out, stream, &bn);
  }
  // weight-stationary persistent kernel (one 16-wave workgroup per CU).  Measured on MI355X: its dword-granular register
  // gather wins 2.5x for Cred <= 8 (the dgrad of the 2/3-channel occupancy heads, the 4/6-channel input layers) and loses
  // 2x at Cred = 32 (load-issue bound), so wider layers stay on the LDS-staged float4 kernel below.
  if (t_kernel != 1 && Cred <= 4 * WS_Q && Cres <= 32 && K <= 64 && n_rows >= 2048 && ws_lds_bytes(K, Cred, nt) <= 160 * 1024) {
    const int n_tiles16 = btc_cdiv(n_rows, 16);
    int wgs = btc_cdiv(n_tiles16, WS_WAVES);
    if (wgs > 256) wgs = 256;
    size_t lds = ws_lds_bytes(K, Cred, nt);
    // the statistics epilogue's last arriver shares the slot sums through 16 + 16 bytes x threads of (by then dead) LDS
    // (bn_fuse_finish's s_part): few offsets x few channels -- K = 8 with 4 input channels is 10 KB -- would leave it short
    if (bn.slots && lds < 16 + (size_t)16 * WS_WAVES * 64) lds = 16 + (size_t)16 * WS_WAVES * 64;
    static BtcPerDeviceOnce once;
    btc_once_per_device(once, [] {
      (void)hipFuncSetAttribute((const void*)conv_apply_ws<1, TRANS_W>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      (void)hipFuncSetAttribute((const void*)conv_apply_ws<2, TRANS_W>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    });
    if (nt == 1) conv_apply_ws<1, TRANS_W><<<wgs, WS_WAVES * 64, lds, stream>>>(feat, W, bias, nbr, n_rows, K, Cred, Cres, out, mirror, bn);
    else conv_apply_ws<2, TRANS_W><<<wgs, WS_WAVES * 64, lds, stream>>>(feat, W, bias, nbr, n_rows, K, Cred, Cres, out, mirror, bn);
    BTC_LAUNCH_CHECK();
    return BTC_OK;
  }
  // 64 rows per workgroup (256 threads).  Measured: 32- and 16-row workgroups of this register-staged kernel are ~2x SLOWER on
  // every BtcDet layer -- each workgroup re-reads all K weight panels, so L2->LDS weight traffic scales with the number of
  // workgroups (the template keeps the THREADS parameter; only the 256-thread instances are built).
  const int tm = 64;
  const int n_tiles = btc_cdiv(n_rows, tm);
  // still too few workgroups (deep, narrow levels): also split the result channels
  while (nt > 2 && (long long)n_tiles * btc_cdiv(Cres, nt * 16) < 512) nt >>= 1;
  if (t_nt && t_nt <= 8) nt = t_nt;
  dim3 grid(n_tiles, btc_cdiv(Cres, nt * 16));
  size_t lds = (size_t)(tm * LDA + KC * ldb_of(nt)) * sizeof(float) + (size_t)(tm * K + K + 1) * sizeof(int32_t);
  const bool vec = (Cred & 3) == 0;
#define BTC_APPLY(NT_) launch_apply_t<NT_, TRANS_W, 256>(grid, lds, stream, vec, feat, W, bias, nbr, n_rows, K, Cred, Cres, out, mirror, bn)
  switch (nt) {
    case 1: BTC_APPLY(1); break;
    case 2: BTC_APPLY(2); break;
    case 4: BTC_APPLY(4); break;
    default: BTC_APPLY(8); break;
  }
#undef BTC_APPLY
  BTC_LAUNCH_CHECK();
  return BTC_OK;
}

template <int MT, int NT, int KB, int PH, bool BF>
void launch_wgrad_rows_p(dim3 grid, size_t lds, hipStream_t stream, const float* g, const float* c, const int32_t* map, const int32_t* ord, int rows,
                         int K, int Cg, int Cc, float* part, int swap) {
  static BtcPerDeviceOnce once;   // launches come from the training thread, the autograd thread and the prefetch thread
  btc_once_per_device(once, [] {
    (void)hipFuncSetAttribute((const void*)conv_wgrad_rows_p<MT, NT, KB, PH, BF>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  });
  conv_wgrad_rows_p<MT, NT, KB, PH, BF><<<grid, 256, lds, stream>>>(g, c, map, ord, rows, K, Cg, Cc, part, swap);
}

struct WgradPlan {
  int nt, n_cblk, n_mblk, S, tiles_per_split;
  int rows_kernel;  // 1 = conv_wgrad_rows (row-stationary), 0 = offset-major conv_wgrad_partial
  int mt, kb, ph, groups, swap, rows;
  int pipe;         // rows kernel: 1 = conv_wgrad_rows_p (software-pipelined)
  size_t lds;       // rows kernel: dynamic LDS bytes
};

size_t wgrad_rows_lds(int mt, int nt, int kb, int ph_built, int K, bool pipe) {
  const int noff = kb * ph_built;
  if (pipe) return (size_t)(2 * kb * TM * ldb_of(mt)) * sizeof(float) + (size_t)(3 * TM * noff + 3 * TM) * sizeof(int32_t);
  return (size_t)(kb * TM * ldb_of(mt) + TM * ldb_of(nt)) * sizeof(float) + (size_t)(TM * K + K + TM) * sizeof(int32_t);
}

WgradPlan wgrad_plan(int n_out, int K, int Cin, int Cout, int n_in = -1, bool bf = false) {
  WgradPlan p;
  p.pipe = 0;
  p.lds = 0;
  p.rows_kernel = 0;
  p.swap = 0;
  p.rows = n_out;
  {
    // row-stationary kernel: supported (MT,NT) tile shapes and enough rows to amortise the persistent walk
    const bool swap = n_in > 0 && 2 * n_in < n_out;  // walk the smaller side of the rulebook
    const int rows = swap ? n_in : n_out;
    int mt = btc_cdiv(swap ? Cout : Cin, 16), ntt = btc_cdiv(swap ? Cin : Cout, 16), kb = 0, ph = 7;
    if (mt == 1 && ntt == 1) { kb = 8; ph = 4; }
    else if (mt * ntt == 2) kb = 4;
    else if (mt == 2 && ntt == 2) kb = 4;
    else if (mt == 3 && ntt == 2) kb = 2;
    else if (mt * ntt == 8 && (mt == 2 || mt == 4)) kb = 2;
    else if (mt == 4 && ntt == 4) kb = 1;
    if (kb && rows >= 4096 && K <= 64) {
      // Work split (tools/conv_bench.py, MI355X): two workgroups per CU (512 in all) = row splits x offset groups.  More
      // phases per group = fewer groups re-reading the dOut tile but a larger accumulator slab per workgroup (PH = 7 no
      // longer fits two workgroups per CU) and more slab traffic; the largest PH <= 4 that still leaves >= 3 row tiles
      // per workgroup measured best from 12 K to 210 K rows (e.g. 32->32 at 210 K rows 307 -> 219 us, at 12 K rows 63 -> 30 us).
      const int t_ph = btc_tune_get(BTC_TUNE_WGRAD_PH), t_wgs = btc_tune_get(BTC_TUNE_WGRAD_WGS);
      // software-pipelined variant (conv_wgrad_rows_p) when the gathered operand's channel count is a multiple of 4; it has its
      // own (KB, PH) per tile shape (the B fragments live in registers: LDS holds the double-buffered gather tile only)
      p.pipe = ((swap ? Cout : Cin) & (bf ? 7 : 3)) == 0 && btc_tune_get(BTC_TUNE_WGRAD_PIPE) != 1;   // bf16: 8-channel gathers
      const int wgs = t_wgs ? t_wgs : 512;
      const int n_tiles = btc_cdiv(rows, TM);
      if (p.pipe) {
        if (mt == 1 && ntt == 1) { kb = 4; ph = 4; }
        else if (mt == 2 && ntt == 1) { kb = 2; ph = 8; }
        else if (mt == 1 && ntt == 2) { kb = 4; ph = 4; }
        else if (mt == 2 && ntt == 2) { kb = 2; ph = 8; }
        else if (mt == 3 && ntt == 2) { kb = 2; ph = 4; }
        else if (mt == 2 && ntt == 4) { kb = 2; ph = 4; }
        else if (mt == 4 && ntt == 2) { kb = 1; ph = 8; }
        else { kb = 1; ph = 4; }
        // few tiles: half the phases per workgroup = twice the offset groups = twice the workgroups
        if ((long long)n_tiles * btc_cdiv(K, kb * ph) < 3LL * wgs) ph >>= 1;
        if (t_ph == ph * 2 || t_ph * 2 == ph) ph = t_ph;   // tuning runs: the other variant
        p.lds = wgrad_rows_lds(mt, ntt, kb, ph, K, true);
      } else {
        if (!(mt == 1 && ntt == 1)) {
          ph = 1;
          for (int cand = 4; cand > 1; cand >>= 1)
            if ((long long)n_tiles * btc_cdiv(K, kb * cand) >= 3LL * wgs) { ph = cand; break; }
          if (t_ph) ph = t_ph;
        }
        const int ph_built = (mt == 1 && ntt == 1) ? 4 : ((ph == 1 || ph == 2 || ph == 4) ? ph : 7);   // the PH the launch macros instantiate
        p.lds = wgrad_rows_lds(mt, ntt, kb, ph_built, K, false);
      }
      p.rows_kernel = 1;
      p.swap = swap; p.rows = rows;
      p.mt = mt; p.nt = ntt; p.kb = kb; p.ph = ph;
      p.groups = btc_cdiv(K, kb * ph);
      int S = wgs / p.groups;
      if (S > n_tiles / 2) S = n_tiles / 2;  // at least two row tiles per persistent workgroup
      if (S > n_tiles) S = n_tiles;
      if (S < 1) S = 1;
      p.S = S;
      p.n_cblk = p.n_mblk = 1;
      p.tiles_per_split = 0;
      return p;
    }
  }
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

extern "C" int btc_conv_fwd_bf16(const void* feat, const float* W, const float* bias, const int32_t* nbr_out, int n_out, int K,
                                 int Cin, int Cout, void* out, void* stream) {
  BTC_CHECK_ARG(K >= 1 && Cin >= 1 && Cout >= 1 && n_out >= 0, "btc_conv_fwd_bf16: bad sizes");
  return launch_apply<false>((const float*)feat, W, bias, nbr_out, n_out, K, Cin, Cout, (float*)out, (hipStream_t)stream, true);
}

extern "C" int btc_conv_dgrad_bf16(const void* dout, const float* W, const int32_t* nbr_in, int n_in, int K, int Cin, int Cout,
                                   void* din, void* stream) {
  BTC_CHECK_ARG(K >= 1 && Cin >= 1 && Cout >= 1 && n_in >= 0, "btc_conv_dgrad_bf16: bad sizes");
  return launch_apply<true>((const float*)dout, W, nullptr, nbr_in, n_in, K, /*Cred=*/Cout, /*Cres=*/Cin, (float*)din,
                            (hipStream_t)stream, true);
}

extern "C" int btc_conv_dgrad(const float* dout, const float* W, const int32_t* nbr_in, int n_in, int K, int Cin, int Cout,
                              float* din, void* stream) {
  BTC_CHECK_ARG(K >= 1 && Cin >= 1 && Cout >= 1 && n_in >= 0, "btc_conv_dgrad: bad sizes");
  BTC_CHECK_ARG(K <= 512, "btc_conv_dgrad: K=%d too large for the LDS neighbour tile", K);
  return launch_apply<true>(dout, W, nullptr, nbr_in, n_in, K, /*Cred=*/Cout, /*Cres=*/Cin, din, (hipStream_t)stream);
}

// split operands: the kernel gathers through 32-bit byte offsets -- the HOST refuses a source it cannot reach (or whose size it was not
// told), nothing traps on the device
static int split_source_ok(const char* who, long long src_rows, int Cred) {
  BTC_CHECK_ARG(src_rows >= 0, "%s: BTC_OPERANDS_F32_SPLIT needs the row count of src (btc_conv_apply_src / btc_conv_bn_relu_fwd_src)", who);
  BTC_CHECK_ARG(src_rows * Cred * 4 < 0xFFFFFF00LL, "%s: a source of %lld rows x %d channels is past the 32-bit gather offsets of the split-operand "
                "kernel (4 GB): use BTC_OPERANDS_F32", who, src_rows, Cred);
  return BTC_OK;
}

extern "C" int btc_conv_apply_ordered(int pass, int operands, const void* src, const void* W, const float* bias, const int32_t* nbr,
                                      const int32_t* order, int n_rows, int K, int Cin, int Cout, void* dst, void* stream) {
  // (a submanifold layer's source has as many rows as its result; any other source's size is the caller's to state)
  return btc_conv_apply_src(pass, operands, src, pass == BTC_PASS_DGRAD_MIRROR ? (long long)n_rows : -1LL, W, bias, nbr, order, n_rows, K, Cin, Cout,
                            dst, stream);
}

extern "C" int btc_conv_apply_src(int pass, int operands, const void* src, long long src_rows, const void* W, const float* bias, const int32_t* nbr,
                                  const int32_t* order, int n_rows, int K, int Cin, int Cout, void* dst, void* stream) {
  BTC_CHECK_ARG((pass == BTC_PASS_FWD || pass == BTC_PASS_DGRAD || pass == BTC_PASS_DGRAD_MIRROR) && operands >= BTC_OPERANDS_F32 &&
                    operands <= BTC_OPERANDS_F32_SPLIT, "btc_conv_apply_ordered: pass=%d operands=%d", pass, operands);
  // BTC_PASS_DGRAD_MIRROR: dgrad of a submanifold layer through its FORWARD map -- nbr_in[j][k] == nbr_out[j][K-1-k] there, so the
  // kernels read column K-1-k for offset k and the backward map never exists (same bits as the explicit map: tests)
  const int mirror = pass == BTC_PASS_DGRAD_MIRROR;
  if (mirror) pass = BTC_PASS_DGRAD;
  BTC_CHECK_ARG(K >= 1 && K <= 512 && Cin >= 1 && Cout >= 1 && n_rows >= 0, "btc_conv_apply_ordered: bad sizes");
  BTC_CHECK_ARG(pass == BTC_PASS_FWD || bias == nullptr, "btc_conv_apply_ordered: dgrad takes no bias");
  const int Cred = pass == BTC_PASS_FWD ? Cin : Cout, Cres = pass == BTC_PASS_FWD ? Cout : Cin;
  if (operands == BTC_OPERANDS_BF16) {
    BTC_CHECK_ARG(btc_conv_bf16w_supported(K, Cred, Cres), "btc_conv_apply_ordered: bf16 operands need K <= 64, Cred %% 32 == 0, Cres %% 16 == 0 (K=%d, %d -> %d)",
                  K, Cin, Cout);
    return btc_apply_bf16w(src, W, bias, nbr, order, n_rows, K, Cred, Cres, dst, (hipStream_t)stream, mirror);
  }
  if (operands == BTC_OPERANDS_F32_SPLIT) {
    if (n_rows > 0) {
      const int rc = split_source_ok("btc_conv_apply_src", src_rows, Cred);
      if (rc) return rc;
    }
    return btc_apply_split((const float*)src, W, bias, nbr, order, n_rows, K, Cred, Cres, (float*)dst, (hipStream_t)stream, mirror, nullptr);
  }
  const bool bf = operands == BTC_OPERANDS_BF16_ACT;
  if (pass == BTC_PASS_FWD)
    return launch_apply<false>((const float*)src, (const float*)W, bias, nbr, n_rows, K, Cred, Cres, (float*)dst, (hipStream_t)stream, bf, order);
  return launch_apply<true>((const float*)src, (const float*)W, nullptr, nbr, n_rows, K, Cred, Cres, (float*)dst, (hipStream_t)stream, bf, order, mirror);
}

// forward conv whose epilogue also gathers the batch statistics of its result (bn_fuse.h); operands F32 / BF16_ACT only.
// -> BTC_OK, *fused = 1 when the statistics were taken (always, for these operand kinds and n_rows > 0)
int btc_conv_fwd_stats(int operands, const void* src, long long src_rows, const float* W, const float* bias, const int32_t* nbr, const int32_t* order,
                       int n_rows, int K, int Cin, int Cout, void* dst, const BnFuse& bn, hipStream_t stream, int* fused) {
  *fused = 0;
  BTC_CHECK_ARG(K >= 1 && K <= 512 && Cin >= 1 && Cout >= 1 && n_rows >= 0, "btc_conv_fwd_stats: bad sizes");
  BTC_CHECK_ARG(operands >= BTC_OPERANDS_F32 && operands <= BTC_OPERANDS_F32_SPLIT, "btc_conv_fwd_stats: operands=%d", operands);
  BTC_CHECK_ARG(Cout <= BN_FUSE_CMAX, "btc_conv_fwd_stats: more than %d channels", BN_FUSE_CMAX);
  if (n_rows <= 0) return BTC_OK;
  if (operands == BTC_OPERANDS_BF16) {   // W = wt_bf16 (the forward copy of btc_weights_to_bf16)
    BTC_CHECK_ARG(btc_conv_bf16w_supported(K, Cin, Cout), "btc_conv_fwd_stats: bf16 operands need K <= 64, Cin %% 32 == 0, Cout %% 16 == 0");
    *fused = 1;
    return btc_apply_bf16w(src, W, bias, nbr, order, n_rows, K, Cin, Cout, dst, stream, 0, &bn);
  }
  if (operands == BTC_OPERANDS_F32_SPLIT) {
    const int rc = split_source_ok("btc_conv_bn_relu_fwd_src", src_rows, Cin);
    if (rc) return rc;
    *fused = 1;
    return btc_apply_split((const float*)src, W, bias, nbr, order, n_rows, K, Cin, Cout, (float*)dst, stream, 0, &bn);
  }
  const bool bf = operands == BTC_OPERANDS_BF16_ACT;
  *fused = 1;
  return launch_apply<false>((const float*)src, W, bias, nbr, n_rows, K, Cin, Cout, (float*)dst, stream, bf, order, 0, &bn);
}

// does a weight-gradient launch take the bf16-pipe kernel?  n_feat: rows of `feat` if known (>= 0); have_bwd: the backward map was given
// (the walk may then run over the smaller side of the rulebook)
static bool wgrad_x_wanted(bool bf, int n_out, int n_feat, bool have_bwd, int K, int Cin, int Cout, int* mode, int* swap, int* rows) {
  if (btc_tune_get(BTC_TUNE_WGRAD_X) == 1 || (!bf && btc_tune_get(BTC_TUNE_SPLIT) == 1) || n_feat < 0) return false;
  *mode = bf ? 0 : 1;
  *swap = have_bwd && n_feat > 0 && 2 * (long long)n_feat < n_out;
  *rows = *swap ? n_feat : n_out;
  const int cg = *swap ? Cout : Cin, cc = *swap ? Cin : Cout;
  const long long esz = bf ? 2 : 4;
  if (*rows < 2048 || !btc_wgrad_x_supported(*mode, K, cg, cc)) return false;
  return (long long)n_feat * Cin * esz < 0xFFFFFF00LL && (long long)n_out * Cout * esz < 0xFFFFFF00LL;
}

extern "C" size_t btc_conv_wgrad_ws_bytes(int n_out, int K, int Cin, int Cout, int n_in) {
  // one size for both activation types (the bf16 instances of the pipelined kernel want Cin % 8 == 0, so the two plans can differ)
  const WgradPlan p = wgrad_plan(n_out, K, Cin, Cout, n_in, false), q = wgrad_plan(n_out, K, Cin, Cout, n_in, true);
  int S = p.S > q.S ? p.S : q.S;
  // ... and for the bf16-pipe kernel's split of the walk (conv_wgrad_x.hip), over either side, either activation type
  for (int swap = 0; swap < 2; ++swap) {
    if (swap && !(n_in > 0 && 2 * (long long)n_in < n_out)) continue;
    const int cg = swap ? Cout : Cin, cc = swap ? Cin : Cout, rows = swap ? n_in : n_out;
    for (int mode = 0; mode < 2 && rows >= 2048; ++mode)
      if (btc_wgrad_x_supported(mode, K, cg, cc)) {
        int sx = 1, ph = 1;
        btc_wgrad_x_plan(mode, rows, K, cg, cc, &sx, &ph);
        if (sx > S) S = sx;
      }
  }
  // ... and for the narrow-result walk over the input rows (conv_wgrad_n.hip)
  if (btc_wgrad_n_kind(K, Cin, Cout)) {
    const int sn = btc_wgrad_n_plan(n_out), sm = n_in > 0 ? btc_wgrad_n_plan(n_in) : 0;
    if (sn > S) S = sn;
    if (sm > S) S = sm;
  }
  return btc_align((size_t)S * K * Cin * Cout * sizeof(float));
}

template <bool BF>
static int wgrad_impl(const float* feat, const float* dout, const int32_t* nbr_out, int n_out, const int32_t* nbr_in,
                      int n_in, int K, int Cin, int Cout, float* dW, void* ws, size_t ws_bytes, void* stream_,
                      const int32_t* order_out = nullptr, const int32_t* order_in = nullptr, int* slabs_out = nullptr) {
  // slabs_out: the caller adds the slabs up itself, later (btc_wgrad_reduce_multi): *slabs_out = S > 0 slabs of K Cin Cout floats
  // in `ws`, dW untouched -- or 0: dW is complete (no rows: zeros)
  hipStream_t stream = (hipStream_t)stream_;
  BTC_CHECK_ARG(K >= 1 && Cin >= 1 && Cout >= 1 && n_out >= 0, "btc_conv_wgrad: bad sizes");
  // rows of `feat`: n_in when the backward map comes with it, or when a caller without one states a positive count; a legacy call that
  // passes NULL and 0 (the argument used to be ignored without a map) leaves it unknown -> the fp32-pipe kernels, which need no bound
  // nbr_in == nbr_out (the same pointer, n_in == n_out): a submanifold layer -- its backward map is the forward map with the offset index
  // mirrored, nothing else is stored (rulebook.hip); the kernels that walk the output rows take it as "no backward map"
  const bool mirror = nbr_in != nullptr && nbr_in == nbr_out && n_in == n_out;
  if (mirror) nbr_in = nullptr;
  const int n_feat = (nbr_in || n_in > 0) ? n_in : -1;
  if (!nbr_in) n_in = -1;
  BTC_CHECK_ARG(ws_bytes >= btc_conv_wgrad_ws_bytes(n_out, K, Cin, Cout, n_in), "btc_conv_wgrad: workspace too small");
  long long count = (long long)K * Cin * Cout;
  if (slabs_out) *slabs_out = 0;
  if (n_out <= 0) {
    BTC_HIP(hipMemsetAsync(dW, 0, (size_t)count * sizeof(float), stream));
    return BTC_OK;
  }
  WgradPlan p = wgrad_plan(n_out, K, Cin, Cout, n_in, BF);
  float* part = (float*)ws;
  // every path below ends with the same reduction of p.S slabs
#define BTC_WGRAD_FINISH()                                                             \
  do {                                                                                 \
    BTC_LAUNCH_CHECK();                                                                \
    if (slabs_out) {                                                                   \
      *slabs_out = p.S;                                                                \
    } else {                                                                           \
      wgrad_reduce<<<btc_cdiv(count, 256), 256, 0, stream>>>(part, p.S, count, dW);    \
      BTC_LAUNCH_CHECK();                                                              \
    }                                                                                  \
    return BTC_OK;                                                                     \
  } while (0)
  const int n_kind = btc_tune_get(BTC_TUNE_WGRAD_NARROW) != 1 ? btc_wgrad_n_kind(K, Cin, Cout) : 0;
  if (n_kind == 1 && (mirror || nbr_in)) {
    // narrow result side (the 5-channel occupancy head): walk the layer's INPUT rows -- x read once, dy gathered (conv_wgrad_n.hip).
    // 32-bit byte offsets: map and both operands under 4 GB.
    const long long rows = mirror ? n_out : n_in, esz = BF ? 2 : 4;
    if (rows >= 2048 && rows * K * 4 < 0xFFFFFF00LL && rows * Cin * esz < 0xFFFFFF00LL && (long long)n_out * Cout * esz < 0xFFFFFF00LL) {
      p.S = btc_wgrad_n_plan((int)rows);
      const int rc = btc_launch_wgrad_n(BF, feat, dout, mirror ? nbr_out : nbr_in, (int)rows, K, Cin, Cout, part, mirror ? 1 : 0, stream);
      if (rc != BTC_OK) return rc;
      BTC_WGRAD_FINISH();
    }
  }
  if (n_kind == 2 && n_feat >= 0 && !(nbr_in && 2 * (long long)n_feat < n_out)) {
    // narrow input side (the 4- / 6-channel first layers): walk the OUTPUT rows -- dOut read once, the features gathered through nbr_out
    // (not where the rulebook's input side is less than half the output side: the kernels below walk that side -- 4 -> 16 from 8.4 K to
    // 40 K rows: 13.6 us there, 23.8 here)
    const long long esz = BF ? 2 : 4;
    if (n_out >= 2048 && (long long)n_out * K * 4 < 0xFFFFFF00LL && (long long)n_out * Cout * esz < 0xFFFFFF00LL && (long long)n_feat * Cin * esz < 0xFFFFFF00LL) {
      p.S = btc_wgrad_n_plan(n_out);
      const int rc = btc_launch_wgrad_n(BF, dout, feat, nbr_out, n_out, K, Cout, Cin, part, 2, stream);
      if (rc != BTC_OK) return rc;
      BTC_WGRAD_FINISH();
    }
  }
  {
    // the row-stationary walk on the bf16 matrix pipe (conv_wgrad_x.hip): bf16 activations as they are, fp32 activations as three exact
    // bf16 pieces; any channel counts whose gathered side is a multiple of 16 (a workgroup owns a <= 64 x 64 block of every dW[k]).
    // Its gathers use 32-bit byte offsets: both operands must stay under 4 GB, and the row count of `feat` must be known.
    int x_mode, x_swap, x_rows;
    if (wgrad_x_wanted(BF, n_out, n_feat, nbr_in != nullptr, K, Cin, Cout, &x_mode, &x_swap, &x_rows)) {
      const int cg = x_swap ? Cout : Cin, cc = x_swap ? Cin : Cout;
      int ph = 1;
      btc_wgrad_x_plan(x_mode, x_rows, K, cg, cc, &p.S, &ph);
      const int rc = btc_launch_wgrad_x(x_mode, x_swap ? (const void*)dout : (const void*)feat, x_swap ? (const void*)feat : (const void*)dout,
                                        x_swap ? nbr_in : nbr_out, x_swap ? order_in : order_out, x_rows, K, cg, cc, part, x_swap, stream);
      if (rc != BTC_OK) return rc;
      BTC_WGRAD_FINISH();
    }
  }
  if (p.rows_kernel) {
    // operands of the walk: gathered rows (via the map) and contiguous rows, see conv_wgrad_rows
    const float* g_ = p.swap ? dout : feat;
    const float* c_ = p.swap ? feat : dout;
    const int32_t* map_ = p.swap ? nbr_in : nbr_out;
    const int32_t* ord_ = p.swap ? order_in : order_out;
    const int Cg = p.swap ? Cout : Cin, Cc = p.swap ? Cin : Cout;
    dim3 grid(p.S, p.groups);
    const size_t lds = p.lds;
    if (p.pipe) {
#define BTC_WGP2(MT_, NT_, KB_, PH_) launch_wgrad_rows_p<MT_, NT_, KB_, PH_, BF>(grid, lds, stream, g_, c_, map_, ord_, p.rows, K, Cg, Cc, part, p.swap)
#define BTC_WGP(MT_, NT_, KB_, PH_)                  \
  do {                                               \
    if (p.ph == PH_) BTC_WGP2(MT_, NT_, KB_, PH_);   \
    else BTC_WGP2(MT_, NT_, KB_, (PH_ / 2));         \
  } while (0)
      if (p.mt == 1 && p.nt == 1) BTC_WGP(1, 1, 4, 4);
      else if (p.mt == 2 && p.nt == 1) BTC_WGP(2, 1, 2, 8);
      else if (p.mt == 1 && p.nt == 2) BTC_WGP(1, 2, 4, 4);
      else if (p.mt == 2 && p.nt == 2) BTC_WGP(2, 2, 2, 8);
      else if (p.mt == 3 && p.nt == 2) BTC_WGP(3, 2, 2, 4);
      else if (p.mt == 2 && p.nt == 4) BTC_WGP(2, 4, 2, 4);
      else if (p.mt == 4 && p.nt == 2) BTC_WGP(4, 2, 1, 8);
      else BTC_WGP(4, 4, 1, 4);
#undef BTC_WGP
#undef BTC_WGP2
      BTC_WGRAD_FINISH();
    }
#define BTC_WG_ROWS(MT_, NT_, KB_, PH_) \
  conv_wgrad_rows<MT_, NT_, KB_, PH_, BF><<<grid, 256, lds, stream>>>(g_, c_, map_, ord_, p.rows, K, Cg, Cc, part, p.swap, btc_tune_get(BTC_TUNE_APPLY_DEBUG))
#define BTC_WG_ROWS_PH(MT_, NT_, KB_)               \
  do {                                              \
    if (p.ph == 1) BTC_WG_ROWS(MT_, NT_, KB_, 1);   \
    else if (p.ph == 2) BTC_WG_ROWS(MT_, NT_, KB_, 2); \
    else if (p.ph == 4) BTC_WG_ROWS(MT_, NT_, KB_, 4); \
    else BTC_WG_ROWS(MT_, NT_, KB_, 7);             \
  } while (0)
    if (p.mt == 1 && p.nt == 1) BTC_WG_ROWS(1, 1, 8, 4);
    else if (p.mt == 2 && p.nt == 1) BTC_WG_ROWS_PH(2, 1, 4);
    else if (p.mt == 1 && p.nt == 2) BTC_WG_ROWS_PH(1, 2, 4);
    else if (p.mt == 2 && p.nt == 2) BTC_WG_ROWS_PH(2, 2, 4);
    else if (p.mt == 3 && p.nt == 2) BTC_WG_ROWS_PH(3, 2, 2);
    else if (p.mt == 2 && p.nt == 4) BTC_WG_ROWS_PH(2, 4, 2);
    else if (p.mt == 4 && p.nt == 2) BTC_WG_ROWS_PH(4, 2, 2);
    else BTC_WG_ROWS_PH(4, 4, 1);
#undef BTC_WG_ROWS_PH
#undef BTC_WG_ROWS
    BTC_WGRAD_FINISH();
  }
  dim3 grid(K, p.S, p.n_mblk * p.n_cblk);
  size_t lds = (size_t)(TM * WG_LDA + TM * ldb_of(p.nt)) * sizeof(float) + (4 * TM + 2) * sizeof(int32_t);
  if (((Cin | Cout) & 3) == 0 && btc_tune_get(BTC_TUNE_WGRAD_PIPE) != 1) {
    switch (p.nt) {
      case 1: conv_wgrad_partial_p<1, BF><<<grid, 256, lds, stream>>>(feat, dout, nbr_out, n_out, K, Cin, Cout, p.tiles_per_split, p.n_cblk, part); break;
      case 2: conv_wgrad_partial_p<2, BF><<<grid, 256, lds, stream>>>(feat, dout, nbr_out, n_out, K, Cin, Cout, p.tiles_per_split, p.n_cblk, part); break;
      case 4: conv_wgrad_partial_p<4, BF><<<grid, 256, lds, stream>>>(feat, dout, nbr_out, n_out, K, Cin, Cout, p.tiles_per_split, p.n_cblk, part); break;
      default: conv_wgrad_partial_p<8, BF><<<grid, 256, lds, stream>>>(feat, dout, nbr_out, n_out, K, Cin, Cout, p.tiles_per_split, p.n_cblk, part); break;
    }
    BTC_WGRAD_FINISH();
  }
  switch (p.nt) {
    case 1: conv_wgrad_partial<1, BF><<<grid, 256, lds, stream>>>(feat, dout, nbr_out, n_out, K, Cin, Cout, p.tiles_per_split, p.n_cblk, part); break;
    case 2: conv_wgrad_partial<2, BF><<<grid, 256, lds, stream>>>(feat, dout, nbr_out, n_out, K, Cin, Cout, p.tiles_per_split, p.n_cblk, part); break;
    case 4: conv_wgrad_partial<4, BF><<<grid, 256, lds, stream>>>(feat, dout, nbr_out, n_out, K, Cin, Cout, p.tiles_per_split, p.n_cblk, part); break;
    default: conv_wgrad_partial<8, BF><<<grid, 256, lds, stream>>>(feat, dout, nbr_out, n_out, K, Cin, Cout, p.tiles_per_split, p.n_cblk, part); break;
  }
  BTC_WGRAD_FINISH();
#undef BTC_WGRAD_FINISH
}

extern "C" int btc_conv_wgrad(const float* feat, const float* dout, const int32_t* nbr_out, int n_out, const int32_t* nbr_in,
                              int n_in, int K, int Cin, int Cout, float* dW, void* ws, size_t ws_bytes, void* stream) {
  return wgrad_impl<false>(feat, dout, nbr_out, n_out, nbr_in, n_in, K, Cin, Cout, dW, ws, ws_bytes, stream);
}

extern "C" int btc_conv_wgrad_bf16(const void* feat, const void* dout, const int32_t* nbr_out, int n_out, const int32_t* nbr_in,
                                   int n_in, int K, int Cin, int Cout, float* dW, void* ws, size_t ws_bytes, void* stream) {
  return wgrad_impl<true>((const float*)feat, (const float*)dout, nbr_out, n_out, nbr_in, n_in, K, Cin, Cout, dW, ws, ws_bytes, stream);
}

extern "C" int btc_conv_wgrad_ordered(int bf16_act, const void* feat, const void* dout, const int32_t* nbr_out, int n_out, const int32_t* nbr_in,
                                      int n_in, const int32_t* order_out, const int32_t* order_in, int K, int Cin, int Cout, float* dW, void* ws,
                                      size_t ws_bytes, void* stream) {
  if (bf16_act)
    return wgrad_impl<true>((const float*)feat, (const float*)dout, nbr_out, n_out, nbr_in, n_in, K, Cin, Cout, dW, ws, ws_bytes, stream, order_out,
                            order_in);
  return wgrad_impl<false>((const float*)feat, (const float*)dout, nbr_out, n_out, nbr_in, n_in, K, Cin, Cout, dW, ws, ws_bytes, stream, order_out,
                           order_in);
}

extern "C" int btc_conv_wgrad_slabs(int bf16_act, const void* feat, const void* dout, const int32_t* nbr_out, int n_out, const int32_t* nbr_in,
                                    int n_in, const int32_t* order_out, const int32_t* order_in, int K, int Cin, int Cout, float* dW, void* ws,
                                    size_t ws_bytes, int* n_slabs, void* stream) {
  BTC_CHECK_ARG(n_slabs != nullptr, "btc_conv_wgrad_slabs: n_slabs is NULL");
  if (bf16_act)
    return wgrad_impl<true>((const float*)feat, (const float*)dout, nbr_out, n_out, nbr_in, n_in, K, Cin, Cout, dW, ws, ws_bytes, stream, order_out,
                            order_in, n_slabs);
  return wgrad_impl<false>((const float*)feat, (const float*)dout, nbr_out, n_out, nbr_in, n_in, K, Cin, Cout, dW, ws, ws_bytes, stream, order_out,
                           order_in, n_slabs);
}

extern "C" int btc_wgrad_reduce_multi(const float* const* parts, float* const* dWs, const int* n_slabs, const long long* counts, int n_jobs,
                                      void* stream) {
  BTC_CHECK_ARG(n_jobs >= 0 && (n_jobs == 0 || (parts && dWs && n_slabs && counts)), "btc_wgrad_reduce_multi: bad arguments");
  for (int base = 0; base < n_jobs; base += BTC_WGRAD_MULTI_MAX) {
    ReduceJobs jobs;
    jobs.n = n_jobs - base < BTC_WGRAD_MULTI_MAX ? n_jobs - base : BTC_WGRAD_MULTI_MAX;
    long long blocks = 0;
    for (int j = 0; j < jobs.n; ++j) {
      BTC_CHECK_ARG(n_slabs[base + j] >= 1 && counts[base + j] >= 1 && parts[base + j] && dWs[base + j], "btc_wgrad_reduce_multi: bad job %d", base + j);
      jobs.part[j] = parts[base + j];
      jobs.dW[j] = dWs[base + j];
      jobs.S[j] = n_slabs[base + j];
      jobs.count[j] = counts[base + j];
      jobs.block0[j] = (int)blocks;
      blocks += (counts[base + j] + 255) / 256;
    }
    jobs.block0[jobs.n] = (int)blocks;
    BTC_CHECK_ARG(blocks < (1LL << 31), "btc_wgrad_reduce_multi: too many elements");
    wgrad_reduce_multi<<<(unsigned)blocks, 256, 0, (hipStream_t)stream>>>(jobs);
    BTC_LAUNCH_CHECK();
  }
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
