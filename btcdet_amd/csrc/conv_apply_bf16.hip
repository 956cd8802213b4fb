// Sparse convolution apply with bfloat16 OPERANDS on the bf16 matrix pipe (gfx950 `v_mfma_f32_16x16x32_bf16`, fp32 accumulate):
// BASELINE.json configs[2] / [4] ("bf16 features", "mixed bf16").
//
// conv_apply_g (conv_apply_glds.hip) takes bf16 activations too, but widens them and multiplies by fp32 weights on the fp32
// MFMA (v_mfma_f32_16x16x4_f32, 1/16 of the bf16 rate): on the >= 64-channel layers its matrix phase is the long pole
// (DESIGN.md section 5), so bf16 storage bought nothing there.  Here BOTH operands are bf16 -- the activations as they are
// stored, the weights as a bf16 copy made once per optimizer step (btc_weights_to_bf16) -- and one MFMA retires 32
// reduction channels of a 16 x 16 tile: 16x fewer matrix instructions per item.  Accumulation, bias and the BatchNorm
// statistics downstream stay fp32; the result is rounded to bf16 once (RNE).
//
// Numerics (tests/test_hip_bf16_mfma.py): products of two bf16 values are exact in fp32; the sum over (offset, channel) is
// accumulated in fp32 in the MFMA's internal order, so a result differs from the fp32 fmaf chain over the SAME bf16-rounded
// operands by fp32 accumulation-order effects only (<= 1 bf16 ulp after the final rounding), and from the fp32-weight
// computation by the weight rounding (2^-9 relative per weight, random sign: ~2^-9 / sqrt(K Cin) of the result's scale).
//
// Both passes use ONE operand layout: the weight panel is read as B^T rows contiguous along the reduction axis,
//     forward : Wq = W^T  bf16 [K][Cout][Cin]   (Cred = Cin,  Cres = Cout)
//     dgrad   : Wq = W    bf16 [K][Cin][Cout]   (Cred = Cout, Cres = Cin)
// i.e. Wq[k][n0 + c][cc + r] in both cases.  LDS images (per pipeline stage; the LDS-DMA destination is lane-linear, so the
// bank swizzle is applied to the per-lane SOURCE address, as in conv_apply_g):
//     A   : TM gathered rows x KC bf16        unit p of row r holds source unit  p ^ ((r / RPB) & (UPR - 1))
//     B^T : TN weight rows  x KC bf16         same rule
// A lane's MFMA fragment (8 consecutive reduction channels of its row / column) is one 16-byte LDS read for each operand.
#include <mutex>

#include "btc_common.h"
#include "bn_fuse.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

namespace {

__device__ unsigned short g_zero_row_b[128];  // zero-initialised source of gathers for absent neighbours (>= KC bf16)

constexpr int B_STAGES = 3;

__device__ __forceinline__ void glds16b(const void* g, void* l) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}

template <int N>
__device__ __forceinline__ void wait_vm_b() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// PAIR (KC = 64 over a 32-channel reduction), as in conv_apply_s: an item is TWO active offsets -- units 0..3 of a 64-channel A row are
// the row gathered for the first, 4..7 for the second; the weight panel is the two offsets' panels side by side -- so a tile walks
// ceil(n_act / 2) items.  The accumulators see the same products in the same order: bit-identical results.
template <int WR, int WC, int NTW, int KC, bool PAIR = false>
__global__ __launch_bounds__(64 * WR * WC) void conv_apply_b(const unsigned short* __restrict__ feat, const unsigned short* __restrict__ Wq,
                                                             const float* __restrict__ bias, const int32_t* __restrict__ nbr,
                                                             const int32_t* __restrict__ order, int n_rows, int K, int Cred, int Cres,
                                                             unsigned short* __restrict__ out, int xcd_swizzle, const BnFuse bn) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int NW = WR * WC, THREADS = 64 * NW;
  constexpr int TM = 16 * WR, TN = 16 * NTW * WC;
  constexpr int UPR = KC / 8;                       // 16-byte units per row of either image
  constexpr int RPB = (128 / (KC * 2)) < 1 ? 1 : (128 / (KC * 2));
  constexpr int A_UNITS = TM * UPR, B_UNITS = TN * UPR;
  constexpr int NAI_TOTAL = (A_UNITS + 63) / 64, NAI = (NAI_TOTAL + NW - 1) / NW;
  constexpr int NBI_TOTAL = (B_UNITS + 63) / 64, NBI = (NBI_TOTAL + NW - 1) / NW;
  constexpr int NPI = NAI + NBI;
  constexpr int A_BYTES = (TM * KC * 2 + 1023) / 1024 * 1024;
  constexpr int B_BYTES = (TN * KC * 2 + 1023) / 1024 * 1024;
  constexpr int STAGE = A_BYTES + B_BYTES;
  char* ring = smem;
  int32_t* s_nbr = (int32_t*)(ring + B_STAGES * STAGE);  // [TM][K]
  int32_t* s_kact = s_nbr + TM * K;
  int32_t* s_nact = s_kact + K;
  int32_t* s_row = s_nact + 1;                           // [TM] row of each tile slot (order[] or identity), -1 past the end

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave / WC, wc = wave % WC;
  const int mirror = (xcd_swizzle >> 1) & 1;   // bit 1: a submanifold FORWARD map read as the backward map (column K-1-k), see conv_apply_g
  xcd_swizzle &= 1;
  int bx = blockIdx.x;
  if (xcd_swizzle) {
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
  unsigned long long wave_act;
  {
    bool any = false;
    if (lane < K)
      for (int r = 0; r < 16; ++r) any |= s_nbr[(wr * 16 + r) * K + lane] >= 0;
    wave_act = __ballot(any);
  }
  const int kflag = (lane < K) ? s_kact[lane] : 0;
  __syncthreads();
  if (wave == 0) {
    const unsigned long long m = __ballot(kflag != 0);
    if (kflag) s_kact[__popcll(m & ((1ull << lane) - 1ull))] = lane;
    if (lane == 0) *s_nact = __popcll(m);
  }
  __syncthreads();
  const int n_act = *s_nact;
  static_assert(!PAIR || KC == 64, "PAIR: two 32-channel offsets per 64-channel item");
  const int n_chunks = PAIR ? 1 : Cred / KC;
  const int n_items = PAIR ? (n_act + 1) >> 1 : n_act * n_chunks;

  f32x4 acc[NTW];
#pragma unroll
  for (int nt = 0; nt < NTW; ++nt) acc[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // cursors of the issue walk (two items ahead) and of the compute walk: one item per call, no division by n_chunks in the loop
  int iq = 0, ir = 0, cq = 0, cr = 0;
  auto issue = [&](int st) {
    const int k = s_kact[PAIR ? 2 * iq : iq];
    const int k1 = (PAIR && 2 * iq + 1 < n_act) ? s_kact[2 * iq + 1] : -1;
    const int cc = PAIR ? 0 : ir * KC;
    if (++ir == n_chunks) { ir = 0; ++iq; }
    char* As = ring + st * STAGE;
    char* Bs = As + A_BYTES;
#pragma unroll
    for (int t = 0; t < NAI; ++t) {
      const int ai = (wave + NW * t) % NAI_TOTAL;
      const int U = ai * 64 + lane;
      if (A_UNITS % 64 == 0 || U < A_UNITS) {
        const int rloc = U / UPR;
        const int u = (U % UPR) ^ ((rloc / RPB) & (UPR - 1));
        const int ku = (PAIR && (u & 4)) ? k1 : k;
        const int nb = ku >= 0 ? s_nbr[rloc * K + ku] : -1;
        const unsigned short* src = nb >= 0 ? feat + (size_t)nb * Cred + cc + (PAIR ? (u & 3) : u) * 8 : g_zero_row_b;
        glds16b(src, As + ai * 1024);
      }
    }
    const unsigned short* Wk = Wq + ((size_t)k * Cres + n0) * Cred + cc;
    // (a pair without a second offset reads the first one's panel twice: its half of the gathered rows is zeros)
    const unsigned short* Wk1 = PAIR ? Wq + ((size_t)(k1 >= 0 ? k1 : k) * Cres + n0) * Cred : Wk;
#pragma unroll
    for (int t = 0; t < NBI; ++t) {
      const int bi = (wave + NW * t) % NBI_TOTAL;
      const int U = bi * 64 + lane;
      if (B_UNITS % 64 == 0 || U < B_UNITS) {
        const int c = U / UPR;
        const int u = (U % UPR) ^ ((c / RPB) & (UPR - 1));
        glds16b(((PAIR && (u & 4)) ? Wk1 : Wk) + (size_t)c * Cred + (PAIR ? (u & 3) : u) * 8, Bs + bi * 1024);
      }
    }
  };

  const int arow = lane & 15, kq = lane >> 4;
  if (n_items > 0) issue(0);
  if (n_items > 1) issue(1);
  int st = 0;
  for (int item = 0; item < n_items; ++item) {
    if (item + 1 < n_items) wait_vm_b<NPI>();
    else wait_vm_b<0>();
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (item + 2 < n_items) issue(st == 0 ? 2 : st - 1);
    const int k = s_kact[PAIR ? 2 * cq : cq];
    const int k2 = (PAIR && 2 * cq + 1 < n_act) ? s_kact[2 * cq + 1] : -1;
    if (++cr == n_chunks) { cr = 0; ++cq; }
    const bool act0 = (wave_act >> k) & 1ull, act1 = PAIR && k2 >= 0 && ((wave_act >> k2) & 1ull);
    if (act0 || act1) {
      const char* A = ring + st * STAGE + (wr * 16 + arow) * (KC * 2);
      const char* B = ring + st * STAGE + A_BYTES;
      const int aswz = ((wr * 16 + arow) / RPB) & (UPR - 1);
      constexpr int STEPS = KC / 32;
      bf16x8 a[STEPS], b[STEPS][NTW];   // typed vector loads: an untyped (uint4) LDS read draws a compiler-inserted vmcnt(0), see conv_apply_split.hip
#pragma unroll
      for (int s = 0; s < STEPS; ++s) {
        const int q = s * 4 + kq;
        a[s] = *(const bf16x8*)(A + ((q ^ aswz) * 16));
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt) {
          const int col = (wc * NTW + nt) * 16 + arow;
          b[s][nt] = *(const bf16x8*)(B + col * (KC * 2) + ((q ^ ((col / RPB) & (UPR - 1))) * 16));
        }
      }
#pragma unroll
      for (int s = 0; s < STEPS; ++s) {
        if (PAIR && !(s ? act1 : act0)) continue;   // (this wave's 16 rows have nothing under that offset: as an item skipped)
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt)
          acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[s], b[s][nt], acc[nt], 0, 0, 0);
      }
    }
    st = (st == B_STAGES - 1) ? 0 : st + 1;
  }

  // C/D layout of 16x16: col = lane & 15, row = (lane >> 4) * 4 + reg
  float vals[NTW][4];
  bool valid[4];
#pragma unroll
  for (int nt = 0; nt < NTW; ++nt) {
    const int col = n0 + (wc * NTW + nt) * 16 + (lane & 15);
    const float bv0 = bias ? bias[col] : 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = s_row[wr * 16 + kq * 4 + r];
      const unsigned short h = btc_f32_to_bf16(bias ? (acc[nt][r] + bv0) : acc[nt][r]);
      if (row >= 0) out[(size_t)row * Cres + col] = h;
      valid[r] = row >= 0;
      vals[nt][r] = btc_bf16_to_f32(h);   // the value as STORED: what the BatchNorm behind this layer reads
    }
  }
  if (bn.slots) {   // batch statistics for the BatchNorm behind this layer (bn_fuse.h), round 5: the bf16 step carried 23 bn_stats launches
    bn_fuse_wave<NTW>(bn, vals, valid, n0 + wc * NTW * 16, (int)((bx * WR + wr) & (bn.nslots - 1)));
    bn_fuse_finish(bn, (int*)smem, (double*)(smem + 16));
  }
}

size_t lds_bytes_b(int tm, int tn, int kc, int K) {
  const size_t a = ((size_t)tm * kc * 2 + 1023) / 1024 * 1024, b = ((size_t)tn * kc * 2 + 1023) / 1024 * 1024;
  return (size_t)B_STAGES * (a + b) + (size_t)(tm * K + K + 1 + tm) * sizeof(int32_t);
}

template <int WR, int WC, int NTW, int KC, bool PAIR = false>
int launch_b(const unsigned short* feat, const unsigned short* Wq, const float* bias, const int32_t* nbr, const int32_t* order, int n_rows, int K, int Cred,
             int Cres, unsigned short* out, int xcd, hipStream_t stream, const BnFuse& bn) {
  constexpr int TM = 16 * WR, TN = 16 * NTW * WC;
  const size_t lds = lds_bytes_b(TM, TN, KC, K);
  BTC_CHECK_ARG(lds <= 160 * 1024, "conv_apply_b: tile does not fit the LDS");
  static BtcPerDeviceOnce once;   // launches come from the training thread, the autograd thread and the prefetch thread
  btc_once_per_device(once, [] {
    (void)hipFuncSetAttribute((const void*)conv_apply_b<WR, WC, NTW, KC, PAIR>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  });
  dim3 grid(btc_cdiv(n_rows, TM), Cres / TN);
  conv_apply_b<WR, WC, NTW, KC, PAIR><<<grid, 64 * WR * WC, lds, stream>>>(feat, Wq, bias, nbr, order, n_rows, K, Cred, Cres, out, xcd, bn);
  BTC_LAUNCH_CHECK();
  return BTC_OK;
}

template <int WR, int WC, int NTW>
int launch_b_kc(int kc, const unsigned short* feat, const unsigned short* Wq, const float* bias, const int32_t* nbr, const int32_t* order, int n_rows,
                int K, int Cred, int Cres, unsigned short* out, int xcd, hipStream_t stream, const BnFuse& bn) {
  if (kc == 128) return launch_b<WR, WC, NTW, 64, true>(feat, Wq, bias, nbr, order, n_rows, K, Cred, Cres, out, xcd, stream, bn);   // (code for PAIR)
  if (kc == 64) return launch_b<WR, WC, NTW, 64>(feat, Wq, bias, nbr, order, n_rows, K, Cred, Cres, out, xcd, stream, bn);
  return launch_b<WR, WC, NTW, 32>(feat, Wq, bias, nbr, order, n_rows, K, Cred, Cres, out, xcd, stream, bn);
}

int apply_b(const void* feat_, const void* Wq_, const float* bias, const int32_t* nbr, const int32_t* order, int n_rows, int K, int Cred, int Cres,
            void* out_, hipStream_t stream, int mirror = 0, const BnFuse* bn_ = nullptr) {
  if (n_rows <= 0) return BTC_OK;
  const BnFuse bn = bn_ ? *bn_ : btc_bn_fuse_none();
  const unsigned short* feat = (const unsigned short*)feat_;
  const unsigned short* Wq = (const unsigned short*)Wq_;
  unsigned short* out = (unsigned short*)out_;
  int kc = (Cred % 64 == 0) ? 64 : 32;
  // 32-channel reductions: two offsets per 64-channel item (PAIR; BTC_TUNE_SPLIT_PAIR = 1 switches it off, same bits)
  if (Cred == 32 && K >= 8 && btc_tune_get(BTC_TUNE_SPLIT_PAIR) != 1 && (btc_tune_get(BTC_TUNE_SPLIT_PAIR) == 2 || n_rows >= 5000)) kc = 128;
  const int xcd = (btc_tune_get(BTC_TUNE_APPLY_XCD) == 2 ? 1 : 0) | (mirror ? 2 : 0);   // kernel flags: bit 0 XCD mapping, bit 1 mirrored map
  // wave shapes as conv_apply_g's policy (sparse_conv.hip): 64 rows x 128 columns on 8 waves for wide results, 16-row
  // workgroups with 4 waves across the columns when there are few rows
  if (Cres % 128 == 0) return launch_b_kc<4, 2, 4>(kc, feat, Wq, bias, nbr, order, n_rows, K, Cred, Cres, out, xcd, stream, bn);
  if (Cres % 64 == 0) {
    if (n_rows < 8192) return launch_b_kc<1, 4, 1>(kc, feat, Wq, bias, nbr, order, n_rows, K, Cred, Cres, out, xcd, stream, bn);
    return launch_b_kc<4, 2, 2>(kc, feat, Wq, bias, nbr, order, n_rows, K, Cred, Cres, out, xcd, stream, bn);
  }
  if (Cres % 32 == 0) return launch_b_kc<2, 2, 1>(kc, feat, Wq, bias, nbr, order, n_rows, K, Cred, Cres, out, xcd, stream, bn);
  return launch_b_kc<4, 1, 1>(kc, feat, Wq, bias, nbr, order, n_rows, K, Cred, Cres, out, xcd, stream, bn);
}

__global__ __launch_bounds__(256) void weights_to_bf16(const float* __restrict__ W, int K, int Cin, int Cout, unsigned short* __restrict__ w_b,
                                                       unsigned short* __restrict__ wt_b) {
  const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long per = (long long)Cin * Cout;
  if (e >= (long long)K * per) return;
  const unsigned short h = btc_f32_to_bf16(W[e]);
  w_b[e] = h;
  const int k = (int)(e / per);
  const int rem = (int)(e - (long long)k * per);
  const int ci = rem / Cout, co = rem - ci * Cout;
  wt_b[(long long)k * per + (long long)co * Cin + ci] = h;
}

// the same for up to BF16_MULTI_MAX weights in ONE launch (a parameter group's layers right behind its optimizer step: ~20 launches of 5 us
// per step become two) -- the table travels in the kernel arguments, a block finds its weight by its first-block entry
constexpr int BF16_MULTI_MAX = 32;
struct Bf16Table {
  const float* W[BF16_MULTI_MAX];
  unsigned short* w_b[BF16_MULTI_MAX];
  unsigned short* wt_b[BF16_MULTI_MAX];
  int K[BF16_MULTI_MAX], Cin[BF16_MULTI_MAX], Cout[BF16_MULTI_MAX], first_block[BF16_MULTI_MAX + 1];
  int n;
};

__global__ __launch_bounds__(256) void weights_to_bf16_multi(const Bf16Table t) {
  int s = 0;
  while (s + 1 < t.n && (int)blockIdx.x >= t.first_block[s + 1]) ++s;
  const long long e = (long long)((int)blockIdx.x - t.first_block[s]) * 256 + threadIdx.x;
  const int Cin = t.Cin[s], Cout = t.Cout[s];
  const long long per = (long long)Cin * Cout;
  if (e >= (long long)t.K[s] * per) return;
  const unsigned short h = btc_f32_to_bf16(t.W[s][e]);
  t.w_b[s][e] = h;
  const int k = (int)(e / per);
  const int rem = (int)(e - (long long)k * per);
  const int ci = rem / Cout, co = rem - ci * Cout;
  t.wt_b[s][(long long)k * per + (long long)co * Cin + ci] = h;
}

}  // namespace

extern "C" int btc_weights_to_bf16_multi(const float* const* W, void* const* w_bf16, void* const* wt_bf16, const int32_t* K, const int32_t* Cin,
                                         const int32_t* Cout, int n, void* stream) {
  BTC_CHECK_ARG(n >= 0, "btc_weights_to_bf16_multi: bad count");
  for (int base = 0; base < n; base += BF16_MULTI_MAX) {
    Bf16Table t;
    t.n = n - base < BF16_MULTI_MAX ? n - base : BF16_MULTI_MAX;
    int blocks = 0;
    for (int i = 0; i < t.n; ++i) {
      BTC_CHECK_ARG(K[base + i] >= 1 && Cin[base + i] >= 1 && Cout[base + i] >= 1, "btc_weights_to_bf16_multi: bad sizes");
      t.W[i] = W[base + i];
      t.w_b[i] = (unsigned short*)w_bf16[base + i];
      t.wt_b[i] = (unsigned short*)wt_bf16[base + i];
      t.K[i] = K[base + i]; t.Cin[i] = Cin[base + i]; t.Cout[i] = Cout[base + i];
      t.first_block[i] = blocks;
      blocks += (int)btc_cdiv((long long)K[base + i] * Cin[base + i] * Cout[base + i], 256);
    }
    t.first_block[t.n] = blocks;
    if (blocks > 0) weights_to_bf16_multi<<<blocks, 256, 0, (hipStream_t)stream>>>(t);
    BTC_LAUNCH_CHECK();
  }
  return BTC_OK;
}

int btc_apply_bf16w(const void* src, const void* Wq, const float* bias, const int32_t* nbr, const int32_t* order, int n_rows, int K, int Cred,
                    int Cres, void* dst, hipStream_t stream, int mirror, const BnFuse* bn) {
  return apply_b(src, Wq, bias, nbr, order, n_rows, K, Cred, Cres, dst, stream, mirror, bn);
}

extern "C" int btc_conv_bf16w_supported(int K, int Cred, int Cres) { return K >= 1 && K <= 64 && Cred >= 32 && Cred % 32 == 0 && Cres % 16 == 0; }

extern "C" int btc_weights_to_bf16(const float* W, int K, int Cin, int Cout, void* w_bf16, void* wt_bf16, void* stream) {
  BTC_CHECK_ARG(K >= 1 && Cin >= 1 && Cout >= 1, "btc_weights_to_bf16: bad sizes");
  const long long n = (long long)K * Cin * Cout;
  weights_to_bf16<<<btc_cdiv(n, 256), 256, 0, (hipStream_t)stream>>>(W, K, Cin, Cout, (unsigned short*)w_bf16, (unsigned short*)wt_bf16);
  BTC_LAUNCH_CHECK();
  return BTC_OK;
}

extern "C" int btc_conv_fwd_bf16w(const void* feat, const void* wt_bf16, const float* bias, const int32_t* nbr_out, int n_out, int K, int Cin,
                                  int Cout, void* out, void* stream) {
  BTC_CHECK_ARG(n_out >= 0 && btc_conv_bf16w_supported(K, Cin, Cout), "btc_conv_fwd_bf16w: needs K <= 64, Cin %% 32 == 0, Cout %% 16 == 0 (K=%d, %d -> %d)",
                K, Cin, Cout);
  return apply_b(feat, wt_bf16, bias, nbr_out, nullptr, n_out, K, /*Cred=*/Cin, /*Cres=*/Cout, out, (hipStream_t)stream);
}

extern "C" int btc_conv_dgrad_bf16w(const void* dout, const void* w_bf16, const int32_t* nbr_in, int n_in, int K, int Cin, int Cout, void* din,
                                    void* stream) {
  BTC_CHECK_ARG(n_in >= 0 && btc_conv_bf16w_supported(K, Cout, Cin), "btc_conv_dgrad_bf16w: needs K <= 64, Cout %% 32 == 0, Cin %% 16 == 0 (K=%d, %d -> %d)",
                K, Cin, Cout);
  return apply_b(dout, w_bf16, nullptr, nbr_in, nullptr, n_in, K, /*Cred=*/Cout, /*Cres=*/Cin, din, (hipStream_t)stream);
}
