// Sparse convolution apply, fp32 in / fp32 out, with the products on the bf16 matrix pipe: every fp32 operand is split into
// three bfloat16 pieces  x = hi + mid + lo  (8 + 8 + 8 significant bits, by truncation: the split is EXACT), and
//     a * b  ~=  hi*hi + hi*mid + mid*hi + mid*mid + hi*lo + lo*hi          (the three dropped terms are <= 2^-24 |a b| each)
// is six `v_mfma_f32_16x16x32_bf16` (fp32 accumulate) instead of eight `v_mfma_f32_16x16x4_f32` per 32 reduction channels of a
// 16 x 16 tile: 6 x 16 cycles of the matrix pipe against 8 x 32.  Products of bf16 pairs are exact in fp32, so what differs from
// the fp32 fmaf chain of conv_apply_g is (a) the dropped terms, ~1e-7 of |a b|, and (b) the fp32 accumulation order inside
// the MFMA -- both of the order of the fp32 rounding the exact chain itself carries.  tests/test_hip_split.py states the bound
// (<= 2e-6 of the result's scale against the exact kernel and against the fp64 product) and checks run-to-run bit-identity.
// conv_apply_g stays the parity reference (BTC_TUNE_SPLIT = 1 selects it everywhere); this kernel takes the layers whose matrix
// phase is the long pole (on the >= 64-channel layers conv_apply_g spends 75-85 % of its time there).
//
// Operands: activations fp32 as stored (gathered rows through the LDS-DMA ring, split in registers after the fragment read:
// 8 consecutive channels = two ds_read_b128, 11 VALU operations per pair of values); weights as three bf16 planes made once per
// optimizer step (btc_weights_split3), read as B^T rows contiguous along the reduction axis like conv_apply_b's:
//     forward : Ws = planes of W^T  [3][K][Cout][Cin]   (Cred = Cin,  Cres = Cout)
//     dgrad   : Ws = planes of W    [3][K][Cin][Cout]   (Cred = Cout, Cres = Cin)
// LDS images per pipeline stage (bank swizzles on the per-lane SOURCE address; checked conflict-free for ds_read_b128's four
// 16-lane groups by brute force):
//     A    : TM rows x KC fp32          16-byte unit p of row r holds source unit  p ^ ((r ^ (r >> 3)) & (KC/4 - 1))
//     B^T  : 3 planes x TN rows x KC bf16   unit p of row c holds source unit  p ^ ((c >> 1) & (KC/8 - 1))
#include <mutex>

#include "btc_common.h"
#include "bn_fuse.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

namespace {

// gathered activation rows travel through a raw buffer descriptor over the feature tensor: a lane whose offset lies past num_records
// reads nothing and WRITES ZEROS to its 16 bytes of LDS (tools/lds_dma_oob.hip checks exactly that on the device) -- the lanes of an
// absent neighbour (submanifold levels: 17 of 27 on average) cost no L2 -> LDS traffic at all, where they used to fetch a zero row.
// Offsets are 32-bit: the host entry points refuse a source tensor of 4 GB or more (btc_conv_apply_src; conv_apply_g takes any size).
#define BTC_RSRC_RECORDS 0xFFFFFF00u
#define BTC_RSRC_ABSENT 0xFFFFFFF0u
__device__ __forceinline__ void blds16s(__amdgpu_buffer_rsrc_t rsrc, unsigned voffset, void* l) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)l, 16, voffset, 0, 0, 0);
}

template <int N>
__device__ __forceinline__ void wait_vm_s() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// two fp32 values -> one dword of each plane (low half = a's piece, high half = b's piece)
__device__ __forceinline__ void split2(float a, float b, unsigned& hi, unsigned& mid, unsigned& lo) {
  const unsigned ab = __float_as_uint(a), bb = __float_as_uint(b);
  hi = __builtin_amdgcn_perm(bb, ab, 0x07060302u);
  const float a1 = a - __uint_as_float(ab & 0xFFFF0000u), b1 = b - __uint_as_float(bb & 0xFFFF0000u);   // exact
  const unsigned a1b = __float_as_uint(a1), b1b = __float_as_uint(b1);
  mid = __builtin_amdgcn_perm(b1b, a1b, 0x07060302u);
  const float a2 = a1 - __uint_as_float(a1b & 0xFFFF0000u), b2 = b1 - __uint_as_float(b1b & 0xFFFF0000u);   // exact
  lo = __builtin_amdgcn_perm(__float_as_uint(b2), __float_as_uint(a2), 0x07060302u);
}

// LW > 0: LW extra LOADER waves per workgroup issue every LDS-DMA piece and wait for it; the WR x WC product waves never touch the
// vector-memory queue.  An LDS-DMA instruction holds its wave until the CU's address unit has taken it (~20 cycles a piece behind
// 40-64 pieces per item): with the product waves issuing their own pieces, every wave of the workgroup sat in that queue at the top
// of every item and the matrix pipe idled -- loads and products ADDED (ablation: DESIGN.md section 5, round 4).
// PAIR (KC = 64 over a 32-channel reduction): an item is TWO active offsets -- channels 0..31 of the item's 64 are the rows gathered
// for the first, 32..63 the rows gathered for the second, the weight panel is the two offsets' panels side by side -- so a tile walks
// ceil(n_act / 2) items instead of n_act: half the barriers and waits of a 32-channel layer (a dense level is bound by its item loop,
// not by what it fetches: profiles/r06_rejected_experiments.txt).  Every accumulator sees the same products in the same order as
// with one offset per item: the results are bit-identical.
template <int WR, int WC, int NTW, int KC, int S_STAGES, int LW, bool PAIR = false>
__global__ __launch_bounds__(64 * (WR * WC + LW)) void conv_apply_s(const float* __restrict__ feat, const unsigned short* __restrict__ Ws,
                                                             const float* __restrict__ bias, const int32_t* __restrict__ nbr,
                                                             const int32_t* __restrict__ order, int n_rows, int K, int Cred, int Cres,
                                                             float* __restrict__ out, int flags, const BnFuse bn) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int NW = WR * WC, THREADS = 64 * (NW + LW);
  constexpr int NLD = LW ? LW : NW;                  // waves that issue the DMA pieces
  constexpr int TM = 16 * WR, TN = 16 * NTW * WC;
  constexpr int UPA = KC / 4, UPB = KC / 8;          // 16-byte units per A row (fp32) / per B^T row of one plane (bf16)
  constexpr int A_UNITS = TM * UPA, B_UNITS = 3 * TN * UPB;
  static_assert(A_UNITS % 64 == 0 && B_UNITS % 64 == 0, "whole DMA instructions");
  constexpr int NAI_TOTAL = A_UNITS / 64, NAI = (NAI_TOTAL + NLD - 1) / NLD;
  constexpr int NBI_TOTAL = B_UNITS / 64, NBI = (NBI_TOTAL + NLD - 1) / NLD;
  static_assert(LW == 0 || (NAI_TOTAL % LW == 0 && NBI_TOTAL % LW == 0), "the loader waves share the pieces evenly");
  constexpr int NPI = NAI + NBI;
  constexpr int A_BYTES = TM * KC * 4, P_BYTES = TN * KC * 2;   // one plane of the weight panel
  constexpr int STAGE = A_BYTES + 3 * P_BYTES;
  char* ring = smem;
  int32_t* s_nbr = (int32_t*)(ring + S_STAGES * STAGE);  // [TM][K]
  int32_t* s_kact = s_nbr + TM * K;
  int32_t* s_nact = s_kact + K;
  int32_t* s_row = s_nact + 1;                           // [TM] row of each tile slot (order[] or identity), -1 past the end

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const bool computes = LW == 0 || wave < NW, loads = LW == 0 || wave >= NW;
  const int wr = computes ? wave / WC : 0, wc = wave % WC;
  const int lid = LW ? wave - NW : wave;   // this wave's place among the issuing waves
  const int mirror = (flags >> 1) & 1;   // a submanifold FORWARD map read as the backward map (column K-1-k), see conv_apply_g
  const int dbg = flags >> 8;            // timing experiments (BTC_TUNE_APPLY_DEBUG, wrong results): 1 no products, 2 no loads inside the loop, 4 no weight panels, 8 no row gathers
  int bx = blockIdx.x;
  if (flags & 1) {
    const int nb = gridDim.x, per = nb >> 3, main = per << 3;
    if (bx < main) bx = (bx & 7) * per + (bx >> 3);
  }
  const int row0 = bx * TM;
  const int n0 = blockIdx.y * TN;
  const size_t plane = (size_t)K * Cred * Cres;   // elements between two planes of Ws

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
  const int n_act = __builtin_amdgcn_readfirstlane(*s_nact);   // (uniform: keeps the item cursors in scalar registers)
  static_assert(!PAIR || KC == 64, "PAIR: two 32-channel offsets per 64-channel item");
  const int n_chunks = PAIR ? 1 : Cred / KC;
  const int n_pairs = PAIR ? (n_act + 1) >> 1 : n_act;   // entries of the item walk's outer index
  // gridDim.z > 1: the workgroups z = 0 .. Z-1 of a tile share its (offset, chunk) items -- contiguous ranges, in order -- and each
  // writes its partial sums to slab z of `out` (n_rows x Cres floats each; split_reduce adds them up in z order).  For levels
  // of a few thousand rows: one workgroup per tile walks 54-108 items one after the other while most CUs have nothing to do.
  const int n_all = n_pairs * n_chunks;
  const int i0 = (int)((long long)blockIdx.z * n_all / gridDim.z);
  const int n_items = (dbg & 32) ? i0 : (int)((long long)(blockIdx.z + 1) * n_all / gridDim.z);   // (32: no item loop)
  out += (size_t)blockIdx.z * n_rows * Cres;

  // cursors of the two walks over the tile's (offset, chunk) items -- issue runs S_STAGES - 1 items ahead of compute; both advance by
  // one item per call: no integer division by n_chunks inside the loop (a runtime divisor costs ~100 VALU instructions per wave, twice
  // per item and side: an EMPTY item loop measured 0.5 us per item with three workgroups per CU, additive to the loads and MFMAs)
  int iq = i0 / n_chunks, ir = i0 - iq * n_chunks;
  int cq = iq, cr = ir;
  const __amdgpu_buffer_rsrc_t rfeat = __builtin_amdgcn_make_buffer_rsrc((void*)feat, 0, BTC_RSRC_RECORDS, 0x00020000);
  // No LDS round trip on the way to a DMA: the tile's active offsets sit in one register across the lanes (lane a = the a-th active
  // offset; v_readlane with the cursor), and the neighbours of the rows this wave gathers are read one item AHEAD, behind the issue of
  // the item before (an item used to open with s_kact -> wait -> s_nbr -> wait -> address, per DMA instruction, in every wave at once).
  const int kvec = lane < n_act ? s_kact[lane] : 0;
  int nbq[NAI];
  auto load_nbq = [&]() {
    if (PAIR) {
      // (iq may run one pair past the end, as it runs one offset past the end without PAIR: lanes >= n_act of kvec hold 0)
      const int k0 = __builtin_amdgcn_readlane(kvec, (2 * iq) & 63);
      const int k1 = (2 * iq + 1 < n_act) ? __builtin_amdgcn_readlane(kvec, (2 * iq + 1) & 63) : -1;
#pragma unroll
      for (int t = 0; t < NAI; ++t) {
        const int U = ((lid + NLD * t) % NAI_TOTAL) * 64 + lane;
        const int rloc = U / UPA;
        const int u = (U % UPA) ^ ((rloc ^ (rloc >> 3)) & (UPA - 1));   // the SOURCE unit this lane moves: units 8..15 belong to the second offset
        const int k = (u & 8) ? k1 : k0;
        nbq[t] = k >= 0 ? s_nbr[rloc * K + k] : -1;
      }
      return;
    }
    const int k = __builtin_amdgcn_readlane(kvec, iq);
#pragma unroll
    for (int t = 0; t < NAI; ++t) {
      const int U = ((lid + NLD * t) % NAI_TOTAL) * 64 + lane;
      nbq[t] = s_nbr[(U / UPA) * K + k];
    }
  };
  if (loads) load_nbq();
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)Ws, 0, BTC_RSRC_RECORDS, 0x00020000);
  static_assert(NLD % 2 == 0 && (TN * UPB) % 64 == 0 && 64 % UPB == 0, "one lane offset per issuing wave");
  const int w_c = lane / UPB;   // panel row of the lane within a piece; (row >> 1) & (UPB - 1) of the piece's first row: 4 (lid & 1) at UPB = 8, 0 at UPB = 4
  const int w_su = (lane % UPB) ^ (((w_c >> 1) + (UPB == 8 ? 4 * (lid & 1) : 0)) & (UPB - 1));   // the source unit of the panel row this lane moves
  // PAIR: a panel row is 4 units of the first offset's row + 4 of the second's (each offset's B^T row is 32 channels = 4 units long)
  const unsigned w_lane = (unsigned)(w_c * Cred + (PAIR ? (w_su & 3) : w_su) * 8) * 2u;
  auto issue = [&](int st) {
    const int k = __builtin_amdgcn_readlane(kvec, PAIR ? ((2 * iq) & 63) : iq);
    // (a pair without a second offset loads the first one's panel twice: its half of the gathered rows is zeros, and zeros times
    // finite weights add nothing)
    const int k1 = PAIR ? ((2 * iq + 1 < n_act) ? __builtin_amdgcn_readlane(kvec, (2 * iq + 1) & 63) : k) : k;
    const unsigned w_lane_i = PAIR ? w_lane + ((w_su & 4) ? (unsigned)((size_t)(k1 - k) * Cres * Cred) * 2u : 0u) : w_lane;
    const int cc = ir * KC;
    char* As = ring + st * STAGE;
    char* Bs = As + A_BYTES;
#pragma unroll
    for (int t = 0; t < NAI; ++t) {
      if (dbg & 8) break;
      const int ai = (lid + NLD * t) % NAI_TOTAL;
      const int U = ai * 64 + lane;
      const int rloc = U / UPA;
      const int u = (U % UPA) ^ ((rloc ^ (rloc >> 3)) & (UPA - 1));
      const int nb = nbq[t];
      blds16s(rfeat, nb >= 0 ? ((unsigned)nb * (unsigned)Cred + (unsigned)(PAIR ? (u & 7) * 4 : cc + u * 4)) * 4u : BTC_RSRC_ABSENT, As + ai * 1024);
    }
    // weight panel: piece bi of the issuing wave covers 64 / UPB panel rows (all three planes are 64-unit aligned); the lane's share
    // of the address -- its row within the piece and its swizzled unit -- is the same for every piece of a wave (w_lane, below), the
    // rest is scalar: one buffer load with a register offset + a scalar offset per piece, no per-piece address registers
    const unsigned w_item = (unsigned)(((size_t)k * Cres + n0) * Cred + (PAIR ? 0 : cc)) * 2u;
#pragma unroll
    for (int t = 0; t < NBI; ++t) {
      if (dbg & 4) break;
      const int bi = (lid + NLD * t) % NBI_TOTAL;
      const int pl = (bi * 64) / (TN * UPB), c0 = ((bi * 64) % (TN * UPB)) / UPB;
      const unsigned w_piece = (unsigned)((size_t)pl * plane + (size_t)c0 * Cred) * 2u;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (__attribute__((address_space(3))) void*)(Bs + bi * 1024), 16, w_lane_i, w_item + w_piece, 0, 0);
    }
    if (++ir == n_chunks) {
      ir = 0;
      ++iq;
      load_nbq();   // (past the last active offset: lane n_act of kvec holds 0, a harmless read)
    }
  };

  const int arow = lane & 15, kg = lane >> 4;
  static_assert((S_STAGES - 2) * NPI <= 63, "vmcnt is a 6-bit counter");
  float vals[NTW][4];
  bool valid[4];
  if (LW > 0 && wave >= NW) {
    // ---- loader waves: wait for an item's pieces, meet the product waves at the item's barrier, issue the item S - 1 ahead
    // (two separate branches, so that the loaders' address registers and the product waves' accumulators share the register file)
#pragma unroll
    for (int i = 0; i < S_STAGES - 1; ++i)
      if (i0 + i < n_items) issue(i);
    int st = 0;
    for (int item = i0; item < n_items; ++item) {
      const int left = n_items - 1 - item;
      if (S_STAGES >= 4 && left >= 2) wait_vm_s<(S_STAGES >= 4 ? 2 : 0) * NPI>();
      else if (S_STAGES >= 3 && left >= 1) wait_vm_s<(S_STAGES >= 3 ? 1 : 0) * NPI>();
      else wait_vm_s<0>();
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      if (item + S_STAGES - 1 < n_items && !(dbg & 2)) issue(st == 0 ? S_STAGES - 1 : st - 1);
      st = (st == S_STAGES - 1) ? 0 : st + 1;
    }
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
      for (int r = 0; r < 4; ++r) vals[nt][r] = 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) valid[r] = false;   // (loader waves own no rows)
  } else {
    // three accumulators per tile, one per magnitude class of the piece products (1, 2^-8, 2^-16 of |a b|): the matrix pipe aligns
    // the 32 products of an instruction to the accumulator it adds them to, so small products added to a large running sum lose
    // their low bits one by one (measured: 9x the exact chain's error on the 6912-term sums of the 256 -> 128 layer); summed among
    // themselves they keep them, and the classes meet once, in the epilogue
    f32x4 acc[NTW], accm[NTW], accs[NTW];
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt) acc[nt] = accm[nt] = accs[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < S_STAGES - 1; ++i)
      if (LW == 0 && i0 + i < n_items) issue(i);
    int st = 0;
    for (int item = i0; item < n_items; ++item) {
      // this item has landed; the min(S_STAGES - 2, items left) issued behind it stay in flight
      const int left = n_items - 1 - item;
      if (LW == 0) {
        if (S_STAGES >= 4 && left >= 2) wait_vm_s<(S_STAGES >= 4 ? 2 : 0) * NPI>();
        else if (S_STAGES >= 3 && left >= 1) wait_vm_s<(S_STAGES >= 3 ? 1 : 0) * NPI>();
        else wait_vm_s<0>();
      }
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      if (LW == 0 && item + S_STAGES - 1 < n_items && !(dbg & 2)) issue(st == 0 ? S_STAGES - 1 : st - 1);
      // (pinned: hipcc is free to sink the DMA issue below the products -- it did, in the 128-row instances -- and with two stages the
      // loop's next wait then meets loads that were issued a moment ago)
      asm volatile("" ::: "memory");
      const int k = __builtin_amdgcn_readlane(kvec, PAIR ? ((2 * cq) & 63) : cq);
      const int k2 = (PAIR && 2 * cq + 1 < n_act) ? __builtin_amdgcn_readlane(kvec, (2 * cq + 1) & 63) : -1;
      if (++cr == n_chunks) { cr = 0; ++cq; }
      const bool act0 = (wave_act >> k) & 1ull, act1 = PAIR && k2 >= 0 && ((wave_act >> k2) & 1ull);
      if ((act0 || act1) && !(dbg & 1)) {
        const int r = wr * 16 + arow;
        const char* A = ring + st * STAGE + r * (KC * 4);
        const char* B = ring + st * STAGE + A_BYTES;
        const int aswz = (r ^ (r >> 3)) & (UPA - 1);
        constexpr int STEPS = KC / 32;
        // every fragment of the item is requested before the first product: one exposed LDS round trip per item instead of two per
        // 32-channel step (left alone, hipcc reads A, waits, splits, reads B, waits, multiplies -- step after step)
        // (typed vector loads: hipcc's waitcnt pass puts an `s_waitcnt vmcnt(0)` in front of an LDS read whose memory operand carries no
        // type information -- a uint4 struct copy -- because it cannot order it against the LDS-DMA in flight: that wait sat between the
        // A and the B reads of EVERY instance until round 4 and serialised the loads of item i + S - 1 with the products of item i;
        // tools/isa_waits.py lists the waits inside the loop)
        // (wide tiles: one 32-channel step's fragments at a time -- 56 registers of fragments beside 48 of accumulators)
        constexpr int GS = (NTW >= 4) ? 1 : STEPS;   // steps whose fragments are in registers together
#pragma unroll
        for (int g = 0; g < STEPS / GS; ++g) {
          f32x4 av[GS][2];
          bf16x8 bh[GS][NTW], bm[GS][NTW], bl[GS][NTW];
#pragma unroll
          for (int j = 0; j < GS; ++j) {
            const int u0 = 8 * (g * GS + j) + 2 * kg;   // channels 32 s + 8 kg .. + 7 of the lane's row
            av[j][0] = *(const f32x4*)(A + ((u0 ^ aswz) * 16));
            av[j][1] = *(const f32x4*)(A + (((u0 + 1) ^ aswz) * 16));
          }
#pragma unroll
          for (int j = 0; j < GS; ++j)
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt) {
              const int col = (wc * NTW + nt) * 16 + arow;
              const char* bp = B + col * (KC * 2) + (((4 * (g * GS + j) + kg) ^ ((col >> 1) & (UPB - 1))) * 16);
              bh[j][nt] = *(const bf16x8*)bp;
              bm[j][nt] = *(const bf16x8*)(bp + P_BYTES);
              bl[j][nt] = *(const bf16x8*)(bp + 2 * P_BYTES);
            }
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int j = 0; j < GS; ++j) {
            if (PAIR && !((g * GS + j) ? act1 : act0)) continue;   // (this wave's 16 rows have nothing under that offset: as an item skipped)
            const f32x4 v0 = av[j][0], v1 = av[j][1];
            uint4 ah, am, al;
            split2(v0[0], v0[1], ah.x, am.x, al.x);
            split2(v0[2], v0[3], ah.y, am.y, al.y);
            split2(v1[0], v1[1], ah.z, am.z, al.z);
            split2(v1[2], v1[3], ah.w, am.w, al.w);
            const bf16x8 Ah = __builtin_bit_cast(bf16x8, ah), Am = __builtin_bit_cast(bf16x8, am), Al = __builtin_bit_cast(bf16x8, al);
#define S_MFMA(ACC, X, Y)                 \
    _Pragma("unroll") for (int nt = 0; nt < NTW; ++nt) ACC[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(X, Y[j][nt], ACC[nt], 0, 0, 0)
            S_MFMA(accs, Al, bh);
            S_MFMA(accs, Ah, bl);
            S_MFMA(accs, Am, bm);
            S_MFMA(accm, Am, bh);
            S_MFMA(accm, Ah, bm);
            S_MFMA(acc, Ah, bh);
#undef S_MFMA
          }
        }
      }
      st = (st == S_STAGES - 1) ? 0 : st + 1;
    }

    // epilogue as conv_apply_g's: C/D layout of 16x16: col = lane & 15, row = (lane >> 4) * 4 + reg
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt) {
      const int col = n0 + (wc * NTW + nt) * 16 + (lane & 15);
      const float bv0 = bias ? bias[col] : 0.f;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = s_row[wr * 16 + kg * 4 + r];
        valid[r] = row >= 0;
        const float sum = acc[nt][r] + (accm[nt][r] + accs[nt][r]);
        const float v = bias ? (sum + bv0) : sum;
        if (row >= 0) out[(size_t)row * Cres + col] = v;
        vals[nt][r] = v;
      }
    }
  }
  if (bn.slots) {   // batch statistics for the BatchNorm behind this layer (bn_fuse.h)
    if (computes) bn_fuse_wave<NTW>(bn, vals, valid, n0 + wc * NTW * 16, (int)((bx * WR + wr) & (bn.nslots - 1)));
    bn_fuse_finish(bn, (int*)smem, (double*)(smem + 16));
  }
}

size_t lds_bytes_s(int tm, int tn, int kc, int K, int stages) {
  return (size_t)stages * ((size_t)tm * kc * 4 + (size_t)3 * tn * kc * 2) + (size_t)(tm * K + K + 1 + tm) * sizeof(int32_t);
}

template <int WR, int WC, int NTW, int KC, int S_STAGES, int LW = 0, bool PAIR = false>
int launch_s(const float* feat, const unsigned short* Ws, const float* bias, const int32_t* nbr, const int32_t* order, int n_rows, int K, int Cred,
             int Cres, float* out, int flags, hipStream_t stream, const BnFuse& bn, int zsplit) {
  constexpr int TM = 16 * WR, TN = 16 * NTW * WC;
  const size_t lds = lds_bytes_s(TM, TN, KC, K, S_STAGES);
  BTC_CHECK_ARG(lds <= 160 * 1024, "conv_apply_s: tile does not fit the LDS");
  static BtcPerDeviceOnce once;   // launches come from the training thread, the autograd thread and the prefetch thread
  btc_once_per_device(once, [] {
    (void)hipFuncSetAttribute((const void*)conv_apply_s<WR, WC, NTW, KC, S_STAGES, LW, PAIR>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  });
  dim3 grid(btc_cdiv(n_rows, TM), Cres / TN, zsplit);
  conv_apply_s<WR, WC, NTW, KC, S_STAGES, LW, PAIR><<<grid, 64 * (WR * WC + LW), lds, stream>>>(feat, Ws, bias, nbr, order, n_rows, K, Cred, Cres, out, flags, bn);
  BTC_LAUNCH_CHECK();
  return BTC_OK;
}

// out[r][c] = bias[c] + slab_0[r][c] + slab_1[r][c] + ... (in that order: deterministic) of a z-split launch, with the BatchNorm
// statistics the conv kernels' epilogues gather (bn_fuse.h slots, the same last-arriver finish).  A workgroup owns 64 rows x 64
// columns: a lane reads float4s (16 lanes = one 256-byte row segment), walks 4 of its wave's 16 rows and keeps the column sums of
// its 4 columns; the four lanes of a column group meet through two shuffles.  (Cres % 64 == 0 for every shape that splits.)
__global__ __launch_bounds__(256) void split_reduce(const float* __restrict__ slabs, int Z, const float* __restrict__ bias, int n_rows, int Cres,
                                                    float* __restrict__ out, const BnFuse bn) {
  __shared__ int s_flag;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int cq = lane & 15, rq = lane >> 4;
  const int col = blockIdx.y * 64 + cq * 4;
  const int row0 = (blockIdx.x * 4 + wave) * 16;
  const size_t slab = (size_t)n_rows * Cres;
  const f32x4 b = bias ? *(const f32x4*)(bias + col) : (f32x4){0.f, 0.f, 0.f, 0.f};
  double s1[4] = {0.0, 0.0, 0.0, 0.0}, s2[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = row0 + i * 4 + rq;
    if (row < n_rows) {
      const float* p = slabs + (size_t)row * Cres + col;
      f32x4 v = *(const f32x4*)p;
      for (int z = 1; z < Z; ++z) v += *(const f32x4*)(p + z * slab);
      if (bias) v += b;
      *(f32x4*)(out + (size_t)row * Cres + col) = v;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        s1[e] += (double)v[e];
        s2[e] += (double)v[e] * (double)v[e];
      }
    }
  }
  if (bn.slots) {
    const int slot = (int)((blockIdx.x * 4 + wave) & (bn.nslots - 1));
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      s1[e] += __shfl_xor(s1[e], 16, 64);
      s2[e] += __shfl_xor(s2[e], 16, 64);
      s1[e] += __shfl_xor(s1[e], 32, 64);
      s2[e] += __shfl_xor(s2[e], 32, 64);
      if (lane < 16 && col + e < bn.C) {
        double* p = bn.slots + ((size_t)slot * 2) * BN_FUSE_CMAX + col + e;
        unsafeAtomicAdd(p, s1[e]);
        unsafeAtomicAdd(p + BN_FUSE_CMAX, s2[e]);
      }
    }
    bn_fuse_finish(bn, &s_flag);
  }
}

__global__ __launch_bounds__(256) void weights_split3(const float* __restrict__ W, int K, int Cin, int Cout, unsigned short* __restrict__ w_s,
                                                      unsigned short* __restrict__ wt_s) {
  const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long per = (long long)Cin * Cout, n = (long long)K * per;
  if (e >= n) return;
  const float x = W[e];
  const unsigned xb = __float_as_uint(x);
  const float x1 = x - __uint_as_float(xb & 0xFFFF0000u);
  const unsigned x1b = __float_as_uint(x1);
  const float x2 = x1 - __uint_as_float(x1b & 0xFFFF0000u);
  const unsigned short p[3] = {(unsigned short)(xb >> 16), (unsigned short)(x1b >> 16), (unsigned short)(__float_as_uint(x2) >> 16)};
  const int k = (int)(e / per);
  const int rem = (int)(e - (long long)k * per);
  const int ci = rem / Cout, co = rem - ci * Cout;
  const long long et = (long long)k * per + (long long)co * Cin + ci;
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    w_s[j * n + e] = p[j];
    wt_s[j * n + et] = p[j];
  }
}

// the same for up to SPLIT_MULTI_MAX weights in ONE launch: the table travels in the kernel arguments, a block finds its weight by
// its first-block entry (one optimizer group's eligible layers after its step: ~10 launches of 6 us become one)
constexpr int SPLIT_MULTI_MAX = 32;
struct SplitTable {
  const float* W[SPLIT_MULTI_MAX];
  unsigned short* w_s[SPLIT_MULTI_MAX];
  unsigned short* wt_s[SPLIT_MULTI_MAX];
  int K[SPLIT_MULTI_MAX], Cin[SPLIT_MULTI_MAX], Cout[SPLIT_MULTI_MAX], first_block[SPLIT_MULTI_MAX + 1];
  int n;
};

__global__ __launch_bounds__(256) void weights_split3_multi(const SplitTable t) {
  int s = 0;
  while (s + 1 < t.n && (int)blockIdx.x >= t.first_block[s + 1]) ++s;
  const long long e = (long long)((int)blockIdx.x - t.first_block[s]) * 256 + threadIdx.x;
  const int Cin = t.Cin[s], Cout = t.Cout[s];
  const long long per = (long long)Cin * Cout, n = (long long)t.K[s] * per;
  if (e >= n) return;
  const float x = t.W[s][e];
  const unsigned xb = __float_as_uint(x);
  const float x1 = x - __uint_as_float(xb & 0xFFFF0000u);
  const unsigned x1b = __float_as_uint(x1);
  const float x2 = x1 - __uint_as_float(x1b & 0xFFFF0000u);
  const unsigned short p[3] = {(unsigned short)(xb >> 16), (unsigned short)(x1b >> 16), (unsigned short)(__float_as_uint(x2) >> 16)};
  const int k = (int)(e / per);
  const int rem = (int)(e - (long long)k * per);
  const int ci = rem / Cout, co = rem - ci * Cout;
  const long long et = (long long)k * per + (long long)co * Cin + ci;
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    t.w_s[s][j * n + e] = p[j];
    t.wt_s[s][j * n + et] = p[j];
  }
}

}  // namespace

extern "C" int btc_weights_split3_multi(const float* const* W, void* const* w_split, void* const* wt_split, const int32_t* K, const int32_t* Cin,
                                        const int32_t* Cout, int n, void* stream) {
  BTC_CHECK_ARG(n >= 0, "btc_weights_split3_multi: bad count");
  for (int base = 0; base < n; base += SPLIT_MULTI_MAX) {
    SplitTable t;
    t.n = n - base < SPLIT_MULTI_MAX ? n - base : SPLIT_MULTI_MAX;
    int blocks = 0;
    for (int i = 0; i < t.n; ++i) {
      BTC_CHECK_ARG(K[base + i] >= 1 && Cin[base + i] >= 1 && Cout[base + i] >= 1, "btc_weights_split3_multi: bad sizes");
      t.W[i] = W[base + i];
      t.w_s[i] = (unsigned short*)w_split[base + i];
      t.wt_s[i] = (unsigned short*)wt_split[base + i];
      t.K[i] = K[base + i]; t.Cin[i] = Cin[base + i]; t.Cout[i] = Cout[base + i];
      t.first_block[i] = blocks;
      blocks += (int)btc_cdiv((long long)K[base + i] * Cin[base + i] * Cout[base + i], 256);
    }
    t.first_block[t.n] = blocks;
    if (blocks > 0) weights_split3_multi<<<blocks, 256, 0, (hipStream_t)stream>>>(t);
    BTC_LAUNCH_CHECK();
  }
  return BTC_OK;
}

extern "C" int btc_conv_split_supported(int K, int Cred, int Cres) {
  return K >= 1 && K <= 64 && Cred >= 32 && Cred % 32 == 0 && Cres >= 32 && Cres % 32 == 0;
}

// the built-in policy of the host bindings (which register a scratch buffer per stream, so small levels run z-split): take this
// kernel for an fp32 launch of n_rows rows?
// two offsets per item (PAIR) from this many rows on -- tools/conv_bench.py `split split:21=2`
// (us per launch, one offset per item -> two: 32 -> 32 at 210 K rows 122.8 / 128.3 -> 111.7 / 112.4 (forward / dgrad), at 12 K rows 20.1 -> 16.0,
// at 26-29 K rows 24.1 -> 22.6, 32 -> 64 at 14 K rows 23.5 -> 18.6; the 64 x 64 tiles lose -- dgrad of 64 -> 32 at 29 K rows 32.1 -> 38.1 --
// and so do the layers with few pairs per row)
static bool btc_split_pair_wanted(int K, int n_rows, int shape) { return K >= 8 && shape != 422 && n_rows >= 5000; }

extern "C" int btc_conv_split_wanted(int K, int Cred, int Cres, int n_rows) {
  return btc_tune_get(BTC_TUNE_SPLIT) != 1 && btc_conv_split_supported(K, Cred, Cres) &&
         n_rows >= ((Cred >= 128 || Cres >= 128) ? 2500 : (Cres % 64 == 0 ? (Cred >= 64 ? 2500 : 6000) : 20000));
}

// Ws: the planes btc_weights_split3 made for this pass (wt_split for forward, w_split for dgrad)
int btc_apply_split(const float* src, const void* Ws_, const float* bias_, const int32_t* nbr, const int32_t* order, int n_rows, int K, int Cred,
                    int Cres, float* dst_, hipStream_t stream, int mirror, const BnFuse* bn_) {
  if (n_rows <= 0) return BTC_OK;
  const float* bias = bias_;
  float* dst = dst_;
  BTC_CHECK_ARG(btc_conv_split_supported(K, Cred, Cres), "conv_apply_s: needs K <= 64, Cred %% 32 == 0, Cres %% 64 == 0 (K=%d, %d -> %d)", K, Cred, Cres);
  const BnFuse bn = bn_ ? *bn_ : btc_bn_fuse_none();
  const unsigned short* Ws = (const unsigned short*)Ws_;
  const int flags = (btc_tune_get(BTC_TUNE_APPLY_XCD) == 2 ? 1 : 0) | (mirror ? 2 : 0) | (btc_tune_get(BTC_TUNE_APPLY_DEBUG) << 8);
  const int t_nt = btc_tune_get(BTC_TUNE_APPLY_NT);
  // shapes / chunk from tools/conv_bench.py on MI355X (us per launch, exact fp32 chain -> this kernel): 256 -> 128 at 14 K rows 333 -> 200
  // (64 x 128 tile, 64-channel items, double buffer; 32-channel items with three stages 241), 128 -> 128 164 -> 102, 64 -> 64 at 14 K rows
  // 57 -> 37.5 (64-channel items; 32-channel items 46.7), at 30 K rows 110 -> 68 (32-channel items, three stages; 64-channel items 75),
  // 32 -> 64 at 30 K rows 65 -> 36.  A launch is bound by the issue of its LDS-DMA pieces (the three weight planes are 6 bytes a
  // weight), so the fewer, larger items win until the second workgroup per CU is lost.
  // few rows (under 10 K): with the stream's scratch buffer the 64-row tiles stay and up to four workgroups share a tile's items
  // (z-split, below): 256 -> 128 at 6.4 K rows 192 us (exact chain) -> 146 (32 x 64 tiles, unsplit) -> 99 (64 x 128 tiles, Z = 4);
  // 128 -> 128 96 -> 72 -> 59; 64 -> 64 at 6.4 K rows 37 -> 34 -> 26 (64 x 64, Z = 2), at 3 K rows 31 -> 34 -> 21 (Z = 4).
  // Without scratch: 32 x 64 tiles, so that there are workgroups enough.
  size_t scratch_bytes = 0;
  float* scratch = (float*)btc_scratch(stream, &scratch_bytes);
  const int t_z = btc_tune_get(BTC_TUNE_SPLIT_Z);
  const bool few = n_rows < 10000;
  if (scratch_bytes > BTC_SCRATCH_HEAD) {   // (the head of a registered scratch buffer is reserved)
    scratch = (float*)((char*)scratch + BTC_SCRATCH_HEAD);
    scratch_bytes -= BTC_SCRATCH_HEAD;
  } else {
    scratch = nullptr;
    scratch_bytes = 0;
  }
  const bool can_z = few && t_z != 1 && scratch_bytes >= (size_t)2 * n_rows * Cres * sizeof(float);
  int shape = (Cres % 128 == 0) ? ((!few || can_z || Cres >= 256) ? 424 : 222) : ((!few || can_z) ? 422 : 222);
  // 32 result columns: 128 x 32 / 64 x 32 tiles, the waves split the rows.  32 -> 32 at 26-29 K rows 46 / 40 us (exact chain) -> 29 / 28
  // (128-row tiles; 64-row tiles 32 / 31), at 210 K rows 165 / 196 (forward / dgrad) -> 147 / 146 with 64-row tiles (128-row tiles 183)
  if (Cres % 64 != 0) shape = n_rows >= 100000 ? 412 : 812;
  // 5-22 K rows with up to 64 result columns, and the 2- / 3-offset strided layers: 64 x 32 tiles (two workgroups of four product waves +
  // loaders per CU, every tile resident at once; the gathered rows are fetched once per column half).  With the loader waves, us per
  // launch against the 64 x 64 / 128 x 32 tiles: 64 -> 64 at 14 K rows 32.1 -> 28.6 (dgrad of the strided 64 -> 64 at 6.4 K 29.7 -> 25.3),
  // 32 -> 32 at 12 K rows 24.4 -> 20.8, the 3-offset 64 -> 128 at 5.3 K rows 10.0 -> 7.9; at 3 K rows (z-split wins, 19.2 against 22.6) and
  // from 25 K rows up (25.0 against 26.3) the larger tiles stay
  if ((Cres % 128 != 0 && n_rows >= 5000 && n_rows < 22000) || (K <= 4 && n_rows < 22000)) shape = 412;
  // two offsets per item (PAIR, below) change the balance for 32 -> 32: 64-row tiles at every size (26-29 K rows 23.0 -> 20.9 / 19.9 us)
  if (Cres % 64 != 0 && Cred == 32 && K >= 8 && n_rows >= 5000 && btc_tune_get(BTC_TUNE_SPLIT_PAIR) != 1 && btc_tune_get(BTC_TUNE_APPLY_KC) != 32) shape = 412;
  if ((t_nt == 812 || t_nt == 412) && Cres % 32 == 0) shape = t_nt;
  if ((t_nt == 224 || t_nt == 424) && Cres % 128 == 0) shape = t_nt;   // tuning runs
  if ((t_nt == 222 || t_nt == 422 || t_nt == 414 || t_nt == 814) && Cres % 64 == 0) shape = t_nt;
  if ((t_nt == 418 || t_nt == 818) && Cres % 128 == 0) shape = t_nt;
  const int t_kc = btc_tune_get(BTC_TUNE_APPLY_KC), t_st = btc_tune_get(BTC_TUNE_APPLY_STAGES);
  int kc = (Cred % 64 == 0 && (shape % 10 == 4 || n_rows < 22000) && shape % 100 != 12) ? 64 : 32;
  if (t_kc == 32 || (t_kc == 64 && Cred % 64 == 0)) kc = t_kc;
  // 32-column tiles: 64-channel items where the reduction allows (round 4: half the barriers -- the empty item loop is 0.2-0.5 us per
  // item, DESIGN.md section 5 --: 64 -> 32 at 28.8 K rows 47.6 -> 39.8 us, the dgrad of 32 -> 64 at 14.2 K rows 45.3 -> 37.4, at 3.1 K
  // rows 43.6 -> 36.2; `tools/conv_bench.py split split:4=32` is the comparison)
  if (shape % 100 == 12) kc = (Cred % 64 == 0 && t_kc != 32) ? 64 : 32;
  if (shape == 818) kc = 32;   // (two 80 KB stages do not fit)
  // 32-channel reductions: two offsets per 64-channel item (PAIR, kernel header) on the 32- and 64-column tiles
  const int t_pair = btc_tune_get(BTC_TUNE_SPLIT_PAIR);
  const bool pair = Cred == 32 && t_pair != 1 && t_kc != 32 && (shape == 412 || shape == 812 || shape == 422) && (t_pair == 2 || btc_split_pair_wanted(K, n_rows, shape));
  if (pair) kc = 64;
  int stages = t_st ? t_st : 3;
  if (kc == 64) stages = (t_st == 3 && shape == 422) ? 3 : 2;
  // z-split (conv_apply_s header): few rows -> few tiles -> most CUs idle while each workgroup walks its tile's 27-108 items alone.
  // The partial slabs live in the stream's scratch buffer (btc_set_scratch); without one, or when it is too small, no split.
  const int items = K * (Cred / kc);
  int Z = 1;
  if (can_z && shape % 100 != 12 && items >= 12) {
    Z = t_z > 1 ? t_z : ((Cres % 128 == 0 || n_rows < 4000) ? 4 : 2);
    if (Z > items / 6) Z = items / 6;
    while (Z > 1 && (size_t)Z * n_rows * Cres * sizeof(float) > scratch_bytes) --Z;
    if (Z > 1) dst = scratch;
  }
  // (The slabs are added by a second launch, split_reduce.  Adding them inside the launch -- the tile's last workgroup, a ticket in the
  // scratch head -- was measured in round 4 and is gone: the last arriver's serial tail costs more than the launch it saves, 256 -> 128 at
  // 6.4 K rows 112.8 us against 101.8, 128 -> 128 70.5 / 58.8, the step's conv launches 2296 us against 2219.)
  const BnFuse bn_kernel = Z > 1 ? btc_bn_fuse_none() : bn;
  if (Z > 1) bias = nullptr;
#define S_ARGS src, Ws, bias, nbr, order, n_rows, K, Cred, Cres, dst, flags, stream, bn_kernel, Z
  int rc = BTC_EINVAL;
  // BTC_TUNE_SPLIT_LOADERS: 0 = built-in policy, 1 = the product waves issue their own pieces, 2 / 4 = that many loader waves per workgroup
  const int t_lw = btc_tune_get(BTC_TUNE_SPLIT_LOADERS);
  // built-in (tools/conv_bench.py `split:17=1 split:17=2 split:17=4`, MI355X): four loaders with 64-channel items (64 -> 64 at 14 K rows
  // 34.7 -> 32.1 us, 256 -> 128 at 6.4 K rows 91.5 -> 88.1, 64 -> 32 at 29 K rows 37.6 -> 33.4), two for the 64 x 32 tiles of the 210 K-row
  // level (134.5 -> 125.9); the three-stage 32-channel items are faster when the product waves issue their own pieces (32 -> 32 at
  // 12-29 K rows 24.5 against 27.8): two loaders spend longer on an item's pieces than its products take
  int lw = t_lw == 2 ? 2 : (t_lw == 4 ? 4 : (t_lw == 1 ? 0 : (kc == 64 ? 4 : (shape == 412 ? 2 : 0))));
  {
    const int tm = (shape / 100) * 16, tn = 16 * ((shape / 10) % 10) * (shape % 10);
    if (lw == 4 && ((tm * kc / 256) % 4 != 0 || (3 * tn * kc / 512) % 4 != 0)) lw = 2;   // the loaders share the pieces evenly
  }
#define S_CASE(code, WR_, WC_, NTW_, KC_, ST_)                                                                          \
  case code:                                                                                                           \
    if constexpr ((16 * WR_ * KC_ / 256) % 4 == 0 && (3 * 16 * NTW_ * WC_ * KC_ / 512) % 4 == 0) {                       \
      if (lw == 4) { rc = launch_s<WR_, WC_, NTW_, KC_, ST_, 4>(S_ARGS); break; }                                       \
    }                                                                                                                  \
    rc = lw ? launch_s<WR_, WC_, NTW_, KC_, ST_, 2>(S_ARGS) : launch_s<WR_, WC_, NTW_, KC_, ST_, 0>(S_ARGS);            \
    break
#define S_PAIR(code, WR_, WC_, NTW_)                                                                                     \
  case code:                                                                                                           \
    rc = lw == 4 ? launch_s<WR_, WC_, NTW_, 64, 2, 4, true>(S_ARGS)                                                      \
                 : (lw ? launch_s<WR_, WC_, NTW_, 64, 2, 2, true>(S_ARGS) : launch_s<WR_, WC_, NTW_, 64, 2, 0, true>(S_ARGS)); \
    break
  if (pair) {
    switch (shape) {
      S_PAIR(412, 4, 1, 2);
      S_PAIR(812, 8, 1, 2);
      S_PAIR(422, 4, 2, 2);
    }
#undef S_PAIR
    return rc;
  }
  switch (shape * 10 + stages + (kc == 64 ? 5 : 0)) {   // ...7 / ...8: KC = 64 with 2 / 3 stages
    S_CASE(4243, 4, 2, 4, 32, 3);
    S_CASE(4244, 4, 2, 4, 32, 4);
    S_CASE(4247, 4, 2, 4, 64, 2);
    S_CASE(2243, 2, 2, 4, 32, 3);
    S_CASE(2247, 2, 2, 4, 64, 2);
    S_CASE(2227, 2, 2, 2, 64, 2);
    S_CASE(4228, 4, 2, 2, 64, 3);
    S_CASE(2244, 2, 2, 4, 32, 4);
    S_CASE(4223, 4, 2, 2, 32, 3);
    S_CASE(4224, 4, 2, 2, 32, 4);
    S_CASE(4227, 4, 2, 2, 64, 2);
    S_CASE(4127, 4, 1, 2, 64, 2);
    S_CASE(8127, 8, 1, 2, 64, 2);
    S_CASE(4123, 4, 1, 2, 32, 3);
    S_CASE(4124, 4, 1, 2, 32, 4);
    S_CASE(8123, 8, 1, 2, 32, 3);
    S_CASE(8124, 8, 1, 2, 32, 4);
    S_CASE(4147, 4, 1, 4, 64, 2);
    S_CASE(4143, 4, 1, 4, 32, 3);
    S_CASE(8147, 8, 1, 4, 64, 2);
    S_CASE(8143, 8, 1, 4, 32, 3);
    S_CASE(4187, 4, 1, 8, 64, 2);
    S_CASE(4183, 4, 1, 8, 32, 3);
    case 8183: rc = launch_s<8, 1, 8, 32, 3, 0>(S_ARGS); break;   // (with loader waves the 168-register budget of 10-12 waves spills)
    S_CASE(2223, 2, 2, 2, 32, 3);
    S_CASE(2224, 2, 2, 2, 32, 4);
    default:
      btc_set_error("conv_apply_s: no instance for shape %d, %d stages, kc %d", shape, stages, kc);
      return BTC_EINVAL;
  }
#undef S_CASE
#undef S_ARGS
  if (rc != BTC_OK || Z == 1) return rc;
  split_reduce<<<dim3(btc_cdiv(n_rows, 64), Cres / 64), 256, 0, stream>>>(dst, Z, bias_, n_rows, Cres, dst_, bn);
  BTC_LAUNCH_CHECK();
  return BTC_OK;
}

extern "C" int btc_weights_split3(const float* W, int K, int Cin, int Cout, void* w_split, void* wt_split, void* stream) {
  BTC_CHECK_ARG(K >= 1 && Cin >= 1 && Cout >= 1, "btc_weights_split3: bad sizes");
  const long long n = (long long)K * Cin * Cout;
  weights_split3<<<btc_cdiv(n, 256), 256, 0, (hipStream_t)stream>>>(W, K, Cin, Cout, (unsigned short*)w_split, (unsigned short*)wt_split);
  BTC_LAUNCH_CHECK();
  return BTC_OK;
}
