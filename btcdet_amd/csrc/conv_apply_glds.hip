// Sparse convolution apply, LDS-DMA pipelined variant (gfx950 `global_load_lds_dwordx4`).
//
// Same contract, same summation order (offset ascending, channel ascending, exact fp32 MFMA fmaf chain) and therefore the
// same bits as conv_apply in sparse_conv.hip.  What changes is how the operands reach LDS: conv_apply stages every
// (offset, 32-channel chunk) item through registers with two workgroup barriers per item, so a workgroup alternates
// between a store phase and an MFMA phase; here the gathered rows and the weight panel are written into a 3-deep LDS
// ring by the LDS-DMA path (no staging VGPRs, no ds_write pass), two items stay in flight across ONE raw s_barrier per
// item (counted `s_waitcnt vmcnt(N)`, never a drain), and a wave skips the MFMAs of an offset none of its 16 rows has.
//
// LDS images (the DMA destination is wave-uniform base + lane * 16 B, so swizzles are applied to the per-lane SOURCE):
//   A (per wave, 16 rows x KC):  slot p of row r holds channel unit  p ^ (r & (KC/4 - 1))
//   B fwd   [k][TN]           :  16-column group g of row k holds columns of group  g ^ (k & 1)       (conflict-free reads)
//   B dgrad [c][KC] (= W^T)   :  slot p of column c holds reduction unit  p ^ (c & (KC/4 - 1))
// Rows without a neighbour at the offset read a zeroed device row (the DMA has no per-lane predicated zero fill).
#include "btc_common.h"
#include "bn_fuse.h"

#include <mutex>

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

__device__ float g_zero_row[64];  // zero-initialised; the source of gathers for absent neighbours

constexpr int G_STAGES_DEFAULT = 3;   // ring depth when the launcher does not say (flags bits 4..7)
constexpr int G_STAGES_MAX = 8;

__device__ __forceinline__ void glds16(const float* g, float* l) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)l, 16,
                                   0, 0);
}

// 4 x 4 transpose across the four 16-lane rows of a wave: on entry lane (r, i) -- row r = lane >> 4 -- holds x[e] = M[r][e],
// on exit y[e] = M[e][r].  Lane (i, kq) of an MFMA operand needs channel 4 q + kq for reduction step q: it reads the 16-byte
// unit kq of a 16-channel block with ONE ds_read_b128 (conflict-free for the swizzled images above) and the transpose hands
// every lane its four steps -- the same values in the same MFMA slots as four ds_read_b32, which for the row-major A / B^T
// images are 2-way bank conflicts each (16 rows x 2 slots of a 32-lane group fall on 16 of the 32 banks).
__device__ __forceinline__ void transpose4(const f32x4 v, float& y0, float& y1, float& y2, float& y3) {
  unsigned x0 = __float_as_uint(v[0]), x1 = __float_as_uint(v[1]), x2 = __float_as_uint(v[2]), x3 = __float_as_uint(v[3]);
  auto a = __builtin_amdgcn_permlane32_swap(x0, x2, false, false);   // rows 2, 3 of x0 <-> rows 0, 1 of x2
  x0 = a[0]; x2 = a[1];
  auto b = __builtin_amdgcn_permlane32_swap(x1, x3, false, false);
  x1 = b[0]; x3 = b[1];
  auto c = __builtin_amdgcn_permlane16_swap(x0, x1, false, false);   // odd rows of x0 <-> even rows of x1
  x0 = c[0]; x1 = c[1];
  auto d = __builtin_amdgcn_permlane16_swap(x2, x3, false, false);
  x2 = d[0]; x3 = d[1];
  y0 = __uint_as_float(x0); y1 = __uint_as_float(x1); y2 = __uint_as_float(x2); y3 = __uint_as_float(x3);
}

template <int N>
__device__ __forceinline__ void wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// Workgroup = WR x WC waves: WR 16-row groups (TM = 16 WR output rows) x WC column groups of NTW 16-column tiles
// (TN = 16 NTW WC result channels).  KC: reduction channels per pipeline item (16, 32 or 64).
// Small layers get small TM (more workgroups than CUs), wide layers split the columns over waves (more waves per SIMD
// to cover each other's LDS round trips); the price of a smaller TM is that every workgroup streams all K weight panels.
// BF: activations (the gathered operand and the result) are bfloat16 in HBM / LDS -- "bf16 features" of BASELINE.json
// configs[2]; weights, bias and the accumulation stay fp32, so a result is the bf16 rounding (RNE) of exactly the fp32
// fmaf chain the fp32 kernel computes on the same (bf16-representable) inputs.
template <int WR, int WC, int NTW, bool TRANS_W, int KC, bool BF>
__global__ __launch_bounds__(64 * WR * WC) void conv_apply_g(const void* __restrict__ feat_, const float* __restrict__ W,
                                                             const float* __restrict__ bias, const int32_t* __restrict__ nbr,
                                                             const int32_t* __restrict__ order, int n_rows, int K, int Cred, int Cres,
                                                             void* __restrict__ out_, int xcd_swizzle, int dbg, const BnFuse bn) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // ring depth (flags bits 4..7, set by the launcher which sized the LDS for it): S - 1 items are in flight while one is multiplied
  const int S = ((xcd_swizzle >> 4) & 15) ? ((xcd_swizzle >> 4) & 15) : G_STAGES_DEFAULT;
  constexpr int NW = WR * WC, THREADS = 64 * NW;
  constexpr int TM = 16 * WR, TN = 16 * NTW * WC, NG = NTW * WC;  // NG: 16-column groups of the B image
  constexpr int ES = BF ? 2 : 4;                   // bytes per activation element
  constexpr int UPR = KC * ES / 16;                // 16-byte units per A row
  constexpr int UPB = KC / 4;                      // 16-byte units per B^T column (weights are fp32)
  constexpr int RPB = (128 / (KC * ES)) < 1 ? 1 : (128 / (KC * ES));  // A rows per 128-byte bank row
  constexpr int A_UNITS = TM * UPR;
  constexpr int NAI_TOTAL = (A_UNITS + 63) / 64;   // A DMA instructions per item (64 units each), whole workgroup
  constexpr int NAI = (NAI_TOTAL + NW - 1) / NW;   // ... per wave (duplicates when not divisible: same data, same place)
  constexpr int NBI_TOTAL = KC * TN / 256;         // B DMA instructions per item, whole workgroup
  constexpr int NBI = (NBI_TOTAL + NW - 1) / NW;
  constexpr int NPI = NAI + NBI;                   // DMA instructions per wave and item
  constexpr int A_BYTES = (TM * KC * ES + 1023) / 1024 * 1024;
  constexpr int STAGE = A_BYTES + KC * TN * 4;     // bytes
  char* ring = smem;                               // [G_STAGES][ A: TM x KC activations | B: KC x TN fp32 ]
  int32_t* s_nbr = (int32_t*)(ring + S * STAGE);  // [TM][K]
  int32_t* s_kact = s_nbr + TM * K;                // [K] flags, then the compact list of active offsets
  int32_t* s_nact = s_kact + K;                    // [1]
  int32_t* s_row = s_nact + 1;                     // [TM] the row each tile slot works on (order[] or identity), -1 past the end
  const char* feat = (const char*)feat_;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave / WC, wc = wave % WC;
  // bit 1 of the flags: the map is a submanifold layer's FORWARD map read as its backward map -- column K-1-k of nbr is
  // offset k of the transposed map (rulebook.hip: the two are mirror images, so nbr_in is never materialised)
  const int mirror = (xcd_swizzle >> 1) & 1;
  xcd_swizzle &= 1;
  int bx = blockIdx.x;
  if (xcd_swizzle) {  // workgroups are dealt round-robin to the 8 XCDs: give XCD x the contiguous tile range x
    const int nb = gridDim.x, per = nb >> 3, main = per << 3;
    if (bx < main) bx = (bx & 7) * per + (bx >> 3);
  }
  const int row0 = bx * TM;
  const int n0 = blockIdx.y * TN;

  for (int e = tid; e < K; e += THREADS) s_kact[e] = 0;
  for (int e = tid; e < TM; e += THREADS) s_row[e] = (row0 + e < n_rows) ? (order ? order[row0 + e] : row0 + e) : -1;
  __syncthreads();
  for (int e = tid; e < TM * K; e += THREADS) {
    const int rloc = e / K, kk = e - rloc * K;
    const int gr = s_row[rloc];
    const int v = gr >= 0 ? nbr[(long long)gr * K + (mirror ? K - 1 - kk : kk)] : -1;
    s_nbr[e] = v;
    if (v >= 0) s_kact[kk] = 1;
  }
  __syncthreads();
  // offsets this wave's 16-row group touches: lane k scans its column of the map (K <= 64)
  unsigned long long wave_act;
  {
    bool any = false;
    if (lane < K)
      for (int r = 0; r < 16; ++r) any |= s_nbr[(wr * 16 + r) * K + lane] >= 0;
    wave_act = __ballot(any);
  }
  const int kflag = (lane < K) ? s_kact[lane] : 0;
  __syncthreads();
  if (wave == 0) {  // compact list of the workgroup's active offsets, ascending
    const unsigned long long m = __ballot(kflag != 0);
    if (kflag) s_kact[__popcll(m & ((1ull << lane) - 1ull))] = lane;
    if (lane == 0) *s_nact = __popcll(m);
  }
  __syncthreads();
  const int n_act = *s_nact;
  const int n_chunks = Cred / KC;
  const int n_items = (dbg & 32) ? 0 : n_act * n_chunks;   // timing experiments: 32 = no item loop (prologue + epilogue only)
  if (dbg & 16) return;                                    //                     16 = prologue only

  f32x4 acc[NTW];
#pragma unroll
  for (int nt = 0; nt < NTW; ++nt) acc[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // cursors of the issue walk (S - 1 items ahead) and of the compute walk: one item per call, no division by n_chunks in the loop
  int iq = 0, ir = 0, cq = 0, cr = 0;
  auto issue = [&](int st) {
    const int k = s_kact[iq];
    const int cc = ir * KC;
    if (++ir == n_chunks) { ir = 0; ++iq; }
    char* As = ring + st * STAGE;
    float* Bs = (float*)(As + A_BYTES);
#pragma unroll
    for (int t = 0; t < NAI; ++t) {
      if (dbg & 8) break;   // timing experiments: no A gathers
      const int ai = (wave + NW * t) % NAI_TOTAL;  // instruction ai covers units [64 ai, 64 ai + 64) of the A image
      const int U = ai * 64 + lane;
      if (A_UNITS % 64 == 0 || U < A_UNITS) {      // a partial last instruction is exec-masked (inactive lanes write nothing)
        const int rloc = U / UPR;
        const int u = (U % UPR) ^ ((rloc / RPB) & (UPR - 1));
        const int nb = s_nbr[rloc * K + k];
        const char* src = nb >= 0 ? feat + ((size_t)nb * Cred + cc) * ES + u * 16 : (const char*)g_zero_row;
        glds16((const float*)src, (float*)(As + ai * 1024));
      }
    }
    const float* Wk = W + (size_t)k * Cred * Cres;
#pragma unroll
    for (int t = 0; t < NBI; ++t) {
      if (dbg & 4) break;   // timing experiments: no weight panels
      const int ii = (wave + NW * t) % NBI_TOTAL;
      const int U = ii * 64 + lane;
      const float* src;
      if (!TRANS_W) {
        const int kr = U / (TN / 4), pu = U % (TN / 4);
        const int g = (NG >= 2) ? ((pu >> 2) ^ (kr & 1)) : (pu >> 2);
        src = Wk + (size_t)(cc + kr) * Cres + n0 + g * 16 + (pu & 3) * 4;
      } else {  // B^T[c][r] = W[k][ci = n0 + c][co = cc + r]: contiguous along r
        const int c = U / UPB, pu = U % UPB;
        const int ru = pu ^ (c & (UPB - 1));
        src = Wk + (size_t)(n0 + c) * Cred + cc + ru * 4;
      }
      glds16(src, Bs + ii * 256);
    }
  };

  const int arow = lane & 15, kq = lane >> 4;
  // vmcnt is a 6-bit counter: the launcher keeps (S - 2) * NPI <= 63 (btc_apply_glds_stages), the clamped cases below are never taken
#define WAIT_ITEMS(n) wait_vm<((n) * NPI > 63 ? 63 : (n) * NPI)>()
  for (int i = 0; i < S - 1 && i < n_items; ++i) issue(i);
  int st = 0;
  for (int item = 0; item < n_items; ++item) {
    // this item has landed; the min(S - 2, items left) issued behind it stay in flight
    const int left = n_items - 1 - item, fl = left < S - 2 ? left : S - 2;
    switch (fl) {
      case 0: wait_vm<0>(); break;
      case 1: WAIT_ITEMS(1); break;
      case 2: WAIT_ITEMS(2); break;
      case 3: WAIT_ITEMS(3); break;
      case 4: WAIT_ITEMS(4); break;
      case 5: WAIT_ITEMS(5); break;
      default: WAIT_ITEMS(6); break;
    }
#undef WAIT_ITEMS
    __builtin_amdgcn_s_barrier();  // everyone's share of this item is in LDS, and everyone is done reading item - 1
    asm volatile("" ::: "memory");
    if (item + S - 1 < n_items && !(dbg & 2)) issue(st == 0 ? S - 1 : st - 1);  // into the stage item - 1 occupied
    const int k = s_kact[cq];
    if (++cr == n_chunks) { cr = 0; ++cq; }
    if (((wave_act >> k) & 1ull) && !(dbg & 1)) {
      // fragment reads are software-pipelined by hand in groups of G reduction steps: the reads of group g + 1 are issued
      // before the MFMAs of group g and pinned there (sched_barrier) -- left alone, hipcc sinks every ds_read next to its
      // MFMA and drains lgkmcnt(0) there, one exposed LDS round trip per step
      const char* A = ring + st * STAGE + (wr * 16 + arow) * (KC * ES);
      const float* B = (const float*)(ring + st * STAGE + A_BYTES);
      const int aswz = ((wr * 16 + arow) / RPB) & (UPR - 1);
      constexpr int Q = KC / 4;
      constexpr int G = (16 / NTW) < 1 ? 1 : ((16 / NTW) > Q ? Q : (16 / NTW));
      constexpr int NGRP = Q / G;
      float a[2][G], b[2][G][NTW];
      constexpr bool A128 = !BF && (G % 4 == 0);               // fp32 A fragments: one b128 read + a lane transpose per 4 steps
      constexpr bool B128 = TRANS_W && (G % 4 == 0) && UPB >= 8 && NTW == 1;  // the same for the B^T image of dgrad (NTW x the swaps: measured +7 % at NTW = 4)
      auto load_group = [&](int g, int buf) {
        if (A128) {
#pragma unroll
          for (int jb = 0; jb < G / 4; ++jb) {
            const int u = (g * (G / 4) + jb) * 4 + kq;
            const f32x4 v = *(const f32x4*)(A + ((u ^ aswz) * 16));
            transpose4(v, a[buf][4 * jb], a[buf][4 * jb + 1], a[buf][4 * jb + 2], a[buf][4 * jb + 3]);
          }
        }
        if (B128) {
#pragma unroll
          for (int nt = 0; nt < NTW; ++nt) {
            const float* colp = B + ((wc * NTW + nt) * 16 + arow) * KC;
#pragma unroll
            for (int jb = 0; jb < G / 4; ++jb) {
              const int u = (g * (G / 4) + jb) * 4 + kq;
              const f32x4 v = *(const f32x4*)(colp + ((u ^ (arow & (UPB - 1))) * 4));
              transpose4(v, b[buf][4 * jb][nt], b[buf][4 * jb + 1][nt], b[buf][4 * jb + 2][nt], b[buf][4 * jb + 3][nt]);
            }
          }
        }
#pragma unroll
        for (int i = 0; i < G; ++i) {
          const int q = g * G + i;
          if (A128) {
          } else if (!BF) {  // channel q*4 + kq: unit q, element kq
            a[buf][i] = *(const float*)(A + ((q ^ aswz) * 16) + kq * 4);
          } else {    // unit q/2, element (q&1)*4 + kq; bf16 -> fp32 is a 16-bit shift
            const unsigned short h = *(const unsigned short*)(A + (((q >> 1) ^ aswz) * 16) + (((q & 1) * 4 + kq) * 2));
            a[buf][i] = __uint_as_float((unsigned)h << 16);
          }
          if (!TRANS_W) {
            const float* bp = B + (q * 4 + kq) * TN + arow;
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt) {
              const int cg = wc * NTW + nt;
              b[buf][i][nt] = bp[((NG >= 2) ? (cg ^ (kq & 1)) : cg) * 16];
            }
          } else if (!B128) {
            const float* bp = B + (wc * NTW * 16 + arow) * KC + (q ^ (arow & (UPB - 1))) * 4 + kq;
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt) b[buf][i][nt] = bp[nt * 16 * KC];
          }
        }
      };
      load_group(0, 0);
#pragma unroll
      for (int g = 0; g < NGRP; ++g) {
        if (g + 1 < NGRP) load_group(g + 1, (g + 1) & 1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < G; ++i)
#pragma unroll
          for (int nt = 0; nt < NTW; ++nt)
            acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[g & 1][i], b[g & 1][i][nt], acc[nt], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    st = (st == S - 1) ? 0 : st + 1;
  }

  // ---- epilogue: C/D layout of 16x16: col = lane&15, row = (lane>>4)*4 + reg
  float vals[NTW][4];
  bool valid[4];
#pragma unroll
  for (int nt = 0; nt < NTW; ++nt) {
    const int col = n0 + (wc * NTW + nt) * 16 + (lane & 15);
    const float bv0 = bias ? bias[col] : 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = s_row[wr * 16 + kq * 4 + r];
      valid[r] = row >= 0;
      float v = bias ? (acc[nt][r] + bv0) : acc[nt][r];
      if (BF) {
        const unsigned short h = btc_f32_to_bf16(v);
        if (row >= 0) ((unsigned short*)out_)[(size_t)row * Cres + col] = h;
        v = btc_bf16_to_f32(h);   // the statistics below are those of the tensor as stored
      } else if (row >= 0) {
        ((float*)out_)[(size_t)row * Cres + col] = v;
      }
      vals[nt][r] = v;
    }
  }
  if (bn.slots) {   // batch statistics for the BatchNorm behind this layer (bn_fuse.h)
    bn_fuse_wave<NTW>(bn, vals, valid, n0 + wc * NTW * 16, (int)((bx * WR + wr) & (bn.nslots - 1)));
    bn_fuse_finish(bn, (int*)smem, (double*)(smem + 16));
  }
}

template <int WR, int WC, int NTW, bool TRANS_W, int KC, bool BF>
int launch_g(const void* feat, const float* W, const float* bias, const int32_t* nbr, const int32_t* order, int n_rows, int K, int Cred,
             int Cres, void* out, int xcd, hipStream_t stream, const BnFuse& bn) {
  constexpr int TM = 16 * WR, TN = 16 * NTW * WC;
  const int stages = btc_apply_glds_stages(WR * 100 + WC * 10 + NTW, KC, K, BF, (xcd >> 4) & 15);
  xcd = (xcd & 15) | (stages << 4);
  const size_t lds = btc_apply_glds_lds_bytes(WR * 100 + WC * 10 + NTW, KC, K, BF, stages);
  BTC_CHECK_ARG(lds <= 160 * 1024, "conv_apply_g: tile does not fit the LDS");
  static BtcPerDeviceOnce once;   // launches come from the training thread, the autograd thread and the prefetch thread
  btc_once_per_device(once, [] {
    (void)hipFuncSetAttribute((const void*)conv_apply_g<WR, WC, NTW, TRANS_W, KC, BF>, hipFuncAttributeMaxDynamicSharedMemorySize,
                              160 * 1024);
  });
  dim3 grid(btc_cdiv(n_rows, TM), Cres / TN);
  conv_apply_g<WR, WC, NTW, TRANS_W, KC, BF><<<grid, 64 * WR * WC, lds, stream>>>(feat, W, bias, nbr, order, n_rows, K, Cred, Cres, out, xcd,
                                                                                btc_tune_get(BTC_TUNE_APPLY_DEBUG), bn);
  BTC_LAUNCH_CHECK();
  return BTC_OK;
}

#define G_ARGS feat, W, bias, nbr, order, n_rows, K, Cred, Cres, out, xcd, stream, bn
#define G_PARAMS                                                                                                             \
  const void *feat, const float *W, const float *bias, const int32_t *nbr, const int32_t *order, int n_rows, int K, int Cred, int Cres, \
      void *out, int xcd, hipStream_t stream, const BnFuse &bn

template <int WR, int WC, int NTW, bool TRANS_W, bool BF>
int launch_g_kc(int kc, G_PARAMS) {
  switch (kc) {
    case 16: return launch_g<WR, WC, NTW, TRANS_W, 16, BF>(G_ARGS);
    case 32: return launch_g<WR, WC, NTW, TRANS_W, 32, BF>(G_ARGS);
    default: return launch_g<WR, WC, NTW, TRANS_W, 64, BF>(G_ARGS);
  }
}

template <bool TRANS_W>
int launch_g_shape(int wr, int wc, int ntw, int kc, bool bf, G_PARAMS) {
  const int code = wr * 100 + wc * 10 + ntw;
  switch (code) {
    // the shapes the built-in policy picks exist for both activation types, the rest (tuning runs) for fp32 only
#define G_CASE2(WR_, WC_, NTW_)                                                      \
  case WR_ * 100 + WC_ * 10 + NTW_:                                                  \
    return bf ? launch_g_kc<WR_, WC_, NTW_, TRANS_W, true>(kc, G_ARGS) : launch_g_kc<WR_, WC_, NTW_, TRANS_W, false>(kc, G_ARGS)
#define G_CASE(WR_, WC_, NTW_)      \
  case WR_ * 100 + WC_ * 10 + NTW_: \
    if (bf) break;                  \
    return launch_g_kc<WR_, WC_, NTW_, TRANS_W, false>(kc, G_ARGS)
    G_CASE2(4, 1, 1); G_CASE(4, 1, 2); G_CASE(4, 1, 4); G_CASE(4, 1, 8);
    G_CASE(4, 2, 1); G_CASE2(4, 2, 2); G_CASE2(4, 2, 4);
    G_CASE2(2, 2, 1); G_CASE(2, 2, 2); G_CASE(2, 2, 4);
    G_CASE(2, 4, 1); G_CASE(2, 4, 2);
    G_CASE2(1, 4, 1); G_CASE(1, 4, 2);
#undef G_CASE
#undef G_CASE2
    default: break;
  }
  btc_set_error("conv_apply_g: no instance for WR=%d WC=%d NTW=%d bf16=%d", wr, wc, ntw, (int)bf);
  return BTC_EINVAL;
}

}  // namespace

bool btc_apply_glds_has_shape(int shape, bool bf) {
  static const int shapes[] = {411, 412, 414, 418, 421, 422, 424, 221, 222, 224, 241, 242, 141, 142};
  static const int shapes_bf[] = {411, 422, 424, 221, 141};
  if (bf) {
    for (int v : shapes_bf)
      if (v == shape) return true;
    return false;
  }
  for (int v : shapes)
    if (v == shape) return true;
  return false;
}

size_t btc_apply_glds_lds_bytes(int shape, int kc, int K, bool bf, int stages) {
  const int tm = 16 * (shape / 100), tn = 16 * ((shape / 10) % 10) * (shape % 10);
  const size_t a_bytes = ((size_t)tm * kc * (bf ? 2 : 4) + 1023) / 1024 * 1024;
  return (size_t)(stages > 0 ? stages : G_STAGES_DEFAULT) * (a_bytes + (size_t)kc * tn * 4) + (size_t)(tm * K + K + 1 + tm) * sizeof(int32_t);
}

// Ring depth of a launch.  An item of a narrow layer is 8-16 MFMAs a wave (0.1-0.25 us) against a DMA round trip of 1-2 us,
// so with two items in flight a workgroup's walk over its (offset, chunk) items is a chain of exposed latencies; the depth
// is therefore as large as the LDS allows while `want_wgs` workgroups still share a CU (64 KB for the ring at most).
int btc_apply_glds_stages(int shape, int kc, int K, bool bf, int asked) {
  const int tm = 16 * (shape / 100), tn = 16 * ((shape / 10) % 10) * (shape % 10), nw = (shape / 100) * ((shape / 10) % 10);
  const size_t a_bytes = ((size_t)tm * kc * (bf ? 2 : 4) + 1023) / 1024 * 1024, stage = a_bytes + (size_t)kc * tn * 4;
  const int npi = (int)(((a_bytes / 1024) + nw - 1) / nw + ((size_t)kc * tn / 256 + nw - 1) / nw);   // DMA instructions per wave and item
  int s = asked ? asked : btc_tune_get(BTC_TUNE_APPLY_STAGES);
  if (!s) {
    s = (int)((size_t)48 * 1024 / stage);
    if (s > 6) s = 6;
  }
  if (s < 3) s = 3;
  if (s > G_STAGES_MAX) s = G_STAGES_MAX;
  while (s > 3 && ((s - 2) * npi > 63 || btc_apply_glds_lds_bytes(shape, kc, K, bf, s) > 160 * 1024)) --s;
  return s;
}

bool btc_apply_glds_supported(int K, int Cred, int Cres) { return K <= 64 && Cred % 16 == 0 && Cres % 16 == 0 && Cred >= 16; }

// shape = WR*100 + WC*10 + NTW (waves: WR row groups x WC column groups of NTW 16-column tiles), kc = reduction chunk,
// bf = activations (feat, out) are bfloat16
int btc_launch_apply_glds(bool trans_w, int shape, int kc, int xcd, bool bf, const void* feat, const float* W, const float* bias,
                          const int32_t* nbr, const int32_t* order, int n_rows, int K, int Cred, int Cres, void* out, hipStream_t stream,
                          const BnFuse* bn_) {
  if (n_rows <= 0) return BTC_OK;
  const BnFuse bn = bn_ ? *bn_ : btc_bn_fuse_none();
  const int wr = shape / 100, wc = (shape / 10) % 10, ntw = shape % 10;
  BTC_CHECK_ARG(btc_apply_glds_supported(K, Cred, Cres) && wc * ntw > 0 && Cres % (16 * wc * ntw) == 0 && Cred % kc == 0 &&
                    (kc == 16 || kc == 32 || kc == 64),
                "conv_apply_g: unsupported K=%d Cred=%d Cres=%d shape=%d kc=%d", K, Cred, Cres, shape, kc);
  return trans_w ? launch_g_shape<true>(wr, wc, ntw, kc, bf, G_ARGS) : launch_g_shape<false>(wr, wc, ntw, kc, bf, G_ARGS);
}
