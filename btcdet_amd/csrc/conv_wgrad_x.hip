// Weight gradient of a sparse convolution on the bf16 matrix pipe (round 5).
//
//     dW[k][ci][co] = sum_rows  g[nbr[row][k]][ci] * c[row][co]
//
// is a contraction over ROWS: for `v_mfma_f32_16x16x32_bf16` both operands must hold 8 consecutive reduction indices -- 8 rows of ONE
// channel -- per lane, while the activations are stored row-major (a row's channels are contiguous).  conv_wgrad_rows_p (sparse_conv.hip)
// sidesteps that with `v_mfma_f32_16x16x4_f32`, whose operands are one value per lane, at 32 cycles of the matrix pipe per 4 rows; this
// kernel keeps the row-major image in LDS and reads the gathered operand with gfx950's transposing LDS read (`ds_read_b64_tr_b16`: a
// 16-lane group reads a 4-row x 16-channel block, lane t comes back with column t), two reads per 32-row step:
//     MODE 0  bf16 activations: the gathered rows go to LDS as they are, ONE product per (16 x 16 block, 32 rows) where the fp32 chain
//             needed eight 16x16x4 instructions on widened values (the bf16 weight gradient was slower than the fp32 one: VERDICT r4 #5);
//     MODE 1  fp32 activations, split EXACTLY into three bf16 planes (x = hi + mid + lo, conv_apply_split.hip's scheme) on the way into
//             LDS (gathered operand) / in registers (contiguous operand); a * b as the six largest piece products, one accumulator per
//             magnitude class: 6 x 16 cycles per 32 rows against 8 x 32.
// Structure of the walk = conv_wgrad_rows_p: a persistent workgroup owns an offset group (PH phases of KB offsets) and a share of the
// 64-row tiles; its dW slab lives in MFMA accumulators for the whole walk and is written once (slab reduction: wgrad_reduce, or ONE
// batched reduction for all layers of a backward pass -- btc_wgrad_reduce_multi); the gathered rows of item g + 1 and the map rows of
// the tile after next are in flight during item g's products (one barrier per item); the contiguous operand never touches LDS (a wave's
// accumulator tiles share one 16-column block, its fragments are 16 element loads per tile and lane).
// Row slots of a 32-row step s as the fragments see them: lane group g = lane >> 4, read hh in {0, 1}, element j in 0..3 -> tile row
// 32 s + 16 hh + 4 g + j, for BOTH operands (any permutation of the reduction index is fine as long as the two agree).
// Results: fp32 sums in walk order per slab, slabs added in index order: deterministic; not the fp32 kernel's bit pattern (tolerances in
// tests/test_hip_wgrad_x.py: MODE 1 against float64 no worse than the fp32 MFMA chain; MODE 0 exact products of the bf16 inputs).
#include <type_traits>
#include <utility>

#include "btc_common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int TMX = 64;   // rows per tile: two 32-row MFMA steps
// both operands travel through raw buffer descriptors with 32-bit byte offsets: the host side refuses tensors of 4 GB or more
// (btc_wgrad_x_supported's callers fall back to conv_wgrad_rows_p), an offset past num_records returns zeros without a fetch
#define X_RECORDS 0xFFFFFF00u
#define X_ABSENT 0xFFFFFFF0u

// two fp32 values -> one dword of each plane (low half = a's piece, high half = b's piece); exact (conv_apply_split.hip)
__device__ __forceinline__ void split2(float a, float b, unsigned& hi, unsigned& mid, unsigned& lo) {
  const unsigned ab = __float_as_uint(a), bb = __float_as_uint(b);
  hi = __builtin_amdgcn_perm(bb, ab, 0x07060302u);
  const float a1 = a - __uint_as_float(ab & 0xFFFF0000u), b1 = b - __uint_as_float(bb & 0xFFFF0000u);
  const unsigned a1b = __float_as_uint(a1), b1b = __float_as_uint(b1);
  mid = __builtin_amdgcn_perm(b1b, a1b, 0x07060302u);
  const float a2 = a1 - __uint_as_float(a1b & 0xFFFF0000u), b2 = b1 - __uint_as_float(b1b & 0xFFFF0000u);
  lo = __builtin_amdgcn_perm(__float_as_uint(b2), __float_as_uint(a2), 0x07060302u);
}

// the lane's 8 reduction rows of one channel: two transposing reads 16 image rows apart
template <int RS>
__device__ __forceinline__ bf16x8 frag_tr(const char* p) {
  typedef __attribute__((address_space(3))) s16x4* lds_p;
  const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)p);
  const s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(p + 16 * RS));
  const s16x8 v = __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
  return __builtin_bit_cast(bf16x8, v);
}

template <int N, typename F, int... I>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
  static_for_impl<N>(f, std::make_integer_sequence<int, N>{});
}

// DEPTH: items of gathered rows in flight ahead of the products (register sets).  1: item t + 1 is loaded during item t's products, and
// the top of item t + 1 waits for it.  2: item t + 2 is issued during item t (a second register set, NU more uint4), a gather has two
// items to land.  Measured per layer (tools/wgrad_bench.py ALT=20=..): MODE 0 is 4 % faster over the step's 28 layers with two in flight
// (the 210 K-row layers 10-17 %), MODE 1 is not (+1 %): its waves spend their issue slots on the split (SQ counters: the two waves of a
// SIMD are issuing 92 % of the time, the matrix pipe is busy 26 %) -- halving the products, dropping two thirds of the fragment reads,
// the contiguous operand's fetches or the gather's bytes each moved its total by < 10 % (profiles/r05_wgrad_x_experiments.txt).
template <int MT, int NT, int KB, int PH, int MODE, int DEPTH>
__global__ __launch_bounds__(256) void conv_wgrad_x(const void* __restrict__ gsrc, const void* __restrict__ csrc,
                                                    const int32_t* __restrict__ nbr, const int32_t* __restrict__ order, int n_rows, int K,
                                                    int Cg_all, int Cc_all, float* __restrict__ part, int swap) {
  // gsrc: gathered operand (rows of Cg_all channels, via the map), csrc: contiguous operand (rows of Cc_all channels); swap: see
  // conv_wgrad_rows.  A workgroup owns a Cg x Cc block of every dW[k] of its offset group: blockIdx.z = (block of the gathered operand's
  // channels) * n_cblk + (block of the contiguous operand's); Cg_all is a multiple of Cg, Cc_all need not be one of Cc (the 5-channel
  // occupancy head: columns past the end are zeros and are not written)
  constexpr int Cg = MT * 16, Cc = NT * 16;
  const int n_cblk = (Cc_all + Cc - 1) / Cc;
  const int cg0 = ((int)blockIdx.z / n_cblk) * Cg, cc0 = ((int)blockIdx.z % n_cblk) * Cc;
  constexpr int NPL = MODE ? 3 : 1;
  constexpr int RS = Cg * 2 + 16;          // bytes per image row (16-byte aligned; the pad staggers the rows over the banks)
  constexpr int IMG = TMX * RS;            // one plane of one offset's gathered tile
  constexpr int ITEM = KB * NPL * IMG;
  constexpr int TPP = KB * MT * NT / 4;    // accumulator tiles per wave and phase
  static_assert(KB * MT * NT % 4 == 0, "phase tiles must split evenly over the 4 waves");
  static_assert(4 % NT == 0, "a wave's tiles must share one column block");
  constexpr int NOFF = PH * KB;
  constexpr int NV = (TMX * NOFF + 255) / 256;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* As = smem;                                     // [2][KB][NPL][TMX] rows of RS bytes
  constexpr int RING = (DEPTH == 2 && PH == 1) ? 4 : 3;   // map rows of tiles in LDS (DEPTH 2, one phase a tile: item t + 2 is tile i + 2)
  int32_t* s_nbr = (int32_t*)(As + 2 * ITEM);          // [RING][TMX][NOFF]
  constexpr int RROW = DEPTH == 2 ? RING + 1 : RING;   // rows of tiles (through `order`) in LDS: DEPTH 2 resolves them one tile ahead of the map
  int32_t* s_row = s_nbr + RING * TMX * NOFF;          // [RROW][TMX]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g4 = lane >> 4, t16 = lane & 15;
  const int kg0 = blockIdx.y * NOFF;
  const int n_tiles = (n_rows + TMX - 1) / TMX;
  const int nt_wg = ((int)blockIdx.x < n_tiles) ? (n_tiles - 1 - (int)blockIdx.x) / (int)gridDim.x + 1 : 0;
  const int bcol = cc0 + (wave % NT) * 16 + t16;       // this lane's column of the contiguous operand
  const bool bvalid = bcol < Cc_all;
  const int lane_off = (4 * g4 + (t16 >> 2)) * RS + (t16 & 3) * 8;   // transposing read: row 4 g + t / 4 of the block, 8-byte chunk t % 4

  constexpr int NC = MODE ? 3 : 1;          // accumulators per tile (magnitude classes)
  f32x4 acc[NC][PH * TPP];
#pragma unroll
  for (int c = 0; c < NC; ++c)
#pragma unroll
    for (int t = 0; t < PH * TPP; ++t) acc[c][t] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const __amdgpu_buffer_rsrc_t rg = __builtin_amdgcn_make_buffer_rsrc((void*)gsrc, 0, X_RECORDS, 0x00020000);
  const __amdgpu_buffer_rsrc_t rc = __builtin_amdgcn_make_buffer_rsrc((void*)csrc, 0, X_RECORDS, 0x00020000);
  int nv[NV], nrow = -1;
  auto load_map = [&](int i) {
    const int row0 = (blockIdx.x + i * gridDim.x) * TMX;
    if (tid < TMX) nrow = (row0 + tid < n_rows) ? (order ? order[row0 + tid] : row0 + tid) : -1;
#pragma unroll
    for (int u = 0; u < NV; ++u) {
      const int e = u * 256 + tid, r = e / NOFF, o = e - r * NOFF;
      int v = -1;
      if (e < TMX * NOFF && row0 + r < n_rows && kg0 + o < K) {
        const int gr = order ? order[row0 + r] : row0 + r;
        v = nbr[(long long)gr * K + kg0 + o];
      }
      nv[u] = v;
    }
  };
  auto store_map = [&](int i) {
    int32_t* dn = s_nbr + (i % RING) * TMX * NOFF;
#pragma unroll
    for (int u = 0; u < NV; ++u) {
      const int e = u * 256 + tid;
      if (e < TMX * NOFF) dn[e] = nv[u];
    }
    if (tid < TMX) s_row[(i % RROW) * TMX + tid] = nrow;
  };

  // DEPTH 2: the same in two stages a barrier apart -- rows through `order` first, the map rows read them from LDS.  load_map() above
  // follows order[] with a dependent load, i.e. a wait for EVERY load in flight (vector-memory loads return in order), the gathers
  // issued ahead included: that wait would bring the walk back to one item in flight once a tile
  auto load_rows = [&](int i) {
    const int row0 = (blockIdx.x + i * gridDim.x) * TMX;
    if (tid < TMX) nrow = (row0 + tid < n_rows) ? (order ? order[row0 + tid] : row0 + tid) : -1;
  };
  auto store_rows = [&](int i) {
    if (tid < TMX) s_row[(i % RROW) * TMX + tid] = nrow;
  };
  auto load_map2 = [&](int i) {
    const int32_t* rw = s_row + (i % RROW) * TMX;
#pragma unroll
    for (int u = 0; u < NV; ++u) {
      const int e = u * 256 + tid, r = e / NOFF, o = e - r * NOFF;
      int v = -1;
      if (e < TMX * NOFF && kg0 + o < K) {
        const int gr = rw[r];
        if (gr >= 0) v = nbr[(long long)gr * K + kg0 + o];
      }
      nv[u] = v;
    }
  };
  auto store_map2 = [&](int i) {
    int32_t* dn = s_nbr + (i % RING) * TMX * NOFF;
#pragma unroll
    for (int u = 0; u < NV; ++u) {
      const int e = u * 256 + tid;
      if (e < TMX * NOFF) dn[e] = nv[u];
    }
  };

  // ---- contiguous operand: 16 elements per tile and lane (slot order s, hh, j), in flight as loaded, split / packed at the tile's top
  float bnf[MODE ? 16 : 1];
  unsigned short bnh[MODE ? 1 : 16];
  bf16x8 bh[2], bm[MODE ? 2 : 1], bl[MODE ? 2 : 1];
  auto load_b = [&](int i) {
    const int32_t* rows = s_row + (i % RROW) * TMX + 4 * g4;
    int gr[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) gr[u] = rows[(u >> 2) * 16 + (u & 3)];   // u = 8 s + 4 hh + j -> tile row 32 s + 16 hh + 4 g + j
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      // (buffer loads: a row past the tile's end is an out-of-range offset -- zeros come back, nothing is fetched, no branch)
      const unsigned off = (gr[u] >= 0 && bvalid) ? ((unsigned)gr[u] * (unsigned)Cc_all + (unsigned)bcol) * (MODE ? 4u : 2u) : X_ABSENT;
      if (MODE) bnf[u] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rc, off, 0, 0));
      else bnh[u] = __builtin_amdgcn_raw_buffer_load_b16(rc, off, 0, 0);
    }
  };
  auto take_b = [&]() {
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      uint4 h, m, l;
      if (MODE) {
        split2(bnf[8 * s + 0], bnf[8 * s + 1], h.x, m.x, l.x);
        split2(bnf[8 * s + 2], bnf[8 * s + 3], h.y, m.y, l.y);
        split2(bnf[8 * s + 4], bnf[8 * s + 5], h.z, m.z, l.z);
        split2(bnf[8 * s + 6], bnf[8 * s + 7], h.w, m.w, l.w);
        bm[s] = __builtin_bit_cast(bf16x8, m);
        bl[s] = __builtin_bit_cast(bf16x8, l);
      } else {
        h.x = (unsigned)bnh[8 * s + 0] | ((unsigned)bnh[8 * s + 1] << 16);
        h.y = (unsigned)bnh[8 * s + 2] | ((unsigned)bnh[8 * s + 3] << 16);
        h.z = (unsigned)bnh[8 * s + 4] | ((unsigned)bnh[8 * s + 5] << 16);
        h.w = (unsigned)bnh[8 * s + 6] | ((unsigned)bnh[8 * s + 7] << 16);
      }
      bh[s] = __builtin_bit_cast(bf16x8, h);
    }
  };

  // ---- gathered operand: registers -> (split ->) row-major bf16 image(s) in LDS
  constexpr int UPR = MODE ? MT * 4 : MT * 2;                  // 16-byte units per gathered row
  constexpr int NU = (KB * TMX * UPR + 255) / 256;             // units per thread and item
  uint4 gq[DEPTH][NU];
  auto load_g = [&](int i, int p, auto set_c) {
    constexpr int SET = decltype(set_c)::value;
    const int32_t* mp = s_nbr + (i % RING) * TMX * NOFF + p * KB;
    int jj[NU];
#pragma unroll
    for (int u = 0; u < NU; ++u) {
      const int e = u * 256 + tid, r = (e / UPR) % TMX, kb = e / (UPR * TMX);
      jj[u] = (e < KB * TMX * UPR) ? mp[r * NOFF + kb] : -1;
    }
#pragma unroll
    for (int u = 0; u < NU; ++u) {
      const int c = (u * 256 + tid) % UPR;
      // absent neighbour (17 of 27 on submanifold levels): out-of-range offset, zeros, no traffic
      const unsigned off = jj[u] >= 0 ? ((unsigned)jj[u] * (unsigned)Cg_all + (unsigned)cg0) * (MODE ? 4u : 2u) + (unsigned)c * 16u : X_ABSENT;
      const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rg, off, 0, 0);
      gq[SET][u] = make_uint4(v.x, v.y, v.z, v.w);
    }
  };
  auto store_g = [&](int buf, auto set_c) {
    constexpr int SET = decltype(set_c)::value;
    char* A = As + buf * ITEM;
#pragma unroll
    for (int u = 0; u < NU; ++u) {
      const int e = u * 256 + tid, c = e % UPR, r = (e / UPR) % TMX, kb = e / (UPR * TMX);
      if (e < KB * TMX * UPR) {
        char* d = A + kb * NPL * IMG + r * RS;
        if (MODE) {
          uint2 h, m, l;
          split2(__uint_as_float(gq[SET][u].x), __uint_as_float(gq[SET][u].y), h.x, m.x, l.x);
          split2(__uint_as_float(gq[SET][u].z), __uint_as_float(gq[SET][u].w), h.y, m.y, l.y);
          *reinterpret_cast<uint2*>(d + c * 8) = h;
          *reinterpret_cast<uint2*>(d + IMG + c * 8) = m;
          *reinterpret_cast<uint2*>(d + 2 * IMG + c * 8) = l;
        } else {
          *reinterpret_cast<uint4*>(d + c * 16) = gq[SET][u];
        }
      }
    }
  };

  typedef std::integral_constant<int, 0> Set0;
  typedef std::integral_constant<int, DEPTH - 1> Set1;
  if (nt_wg > 0 && DEPTH == 1) {
    load_map(0);
    store_map(0);
    if (nt_wg > 1) {
      load_map(1);
      store_map(1);
    }
    __syncthreads();
    load_b(0);
    load_g(0, 0, Set0{});
  }
  if (nt_wg > 0 && DEPTH == 2) {
    for (int j = 0; j < RROW - 1 && j < nt_wg; ++j) {
      load_rows(j);
      store_rows(j);
    }
    __syncthreads();
    for (int j = 0; j < RING - 1 && j < nt_wg; ++j) {
      load_map2(j);
      store_map2(j);
    }
    __syncthreads();
    load_b(0);
    load_g(0, 0, Set0{});
    if (PH > 1) load_g(0, 1, Set1{});
    else if (nt_wg > 1) load_g(1, 0, Set1{});
  }
  int buf = 0;
  // ---- products of item (i, p) from LDS buffer `buf`
  // The fragments of step t + 1 (a step = 32 rows of one accumulator tile) are read while step t's products issue: with one or two
  // waves a SIMD nothing else covers the LDS latency (read -> wait -> six products -> read ... left the matrix pipe idle for most of
  // an item: 3 full LDS waits per 12 products in the ISA)
  // The fragments of step t + 1 (a step = 32 rows of one accumulator tile) are read before step t's products issue
  auto products = [&](auto p_c) {
    constexpr int p = decltype(p_c)::value;
    constexpr int STEPS = 2 * TPP;
    const char* A = As + buf * ITEM + lane_off;
    bf16x8 fh[2], fm[MODE ? 2 : 1], fl[MODE ? 2 : 1];
    auto read_step = [&](int t, int w) {
      const int q = t >> 1, s2 = t & 1, l = q * 4 + wave;      // phase-local tile (kb, mt, nt), nt == wave % NT
      const int mt = (l / NT) % MT, kb = l / (NT * MT);
      const char* ap = A + kb * NPL * IMG + mt * 32 + 32 * s2 * RS;
      fh[w] = frag_tr<RS>(ap);
      if (MODE) {
        fm[MODE ? w : 0] = frag_tr<RS>(ap + IMG);
        fl[MODE ? w : 0] = frag_tr<RS>(ap + 2 * IMG);
      }
    };
    read_step(0, 0);
#pragma unroll
    for (int t = 0; t < STEPS; ++t) {
      const int q = t >> 1, s2 = t & 1, w = t & 1;
      if (t + 1 < STEPS) read_step(t + 1, w ^ 1);
      const bf16x8 ah = fh[w];
      if (MODE) {
        const bf16x8 am = fm[MODE ? w : 0], al = fl[MODE ? w : 0];
        // (issue order: no two consecutive products add to the same accumulator)
        f32x4 c2 = acc[NC - 1][p * TPP + q], c1 = acc[NC > 1 ? 1 : 0][p * TPP + q], c0 = acc[0][p * TPP + q];
        c2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh[s2], c2, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bh[s2], c1, 0, 0, 0);
        c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh[s2], c0, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl[MODE ? s2 : 0], c2, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bm[MODE ? s2 : 0], c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bm[MODE ? s2 : 0], c2, 0, 0, 0);
        acc[NC - 1][p * TPP + q] = c2;
        acc[NC > 1 ? 1 : 0][p * TPP + q] = c1;
        acc[0][p * TPP + q] = c0;
      } else {
        acc[0][p * TPP + q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh[s2], acc[0][p * TPP + q], 0, 0, 0);
      }
    }
  };
  if (DEPTH == 1) {
    for (int i = 0; i < nt_wg; ++i) {
      static_for<PH>([&](auto p_c) {
        constexpr int p = decltype(p_c)::value;
        store_g(buf, Set0{});
        if (p == 0) take_b();
        if (PH > 1 ? (p == 1 && i + 2 < nt_wg) : (i >= 1 && i + 1 < nt_wg)) store_map(PH > 1 ? i + 2 : i + 1);
        __syncthreads();
        // ---- loads for the next item, in flight during this item's products
        if (p + 1 < PH) {
          load_g(i, p + 1, Set0{});
        } else if (i + 1 < nt_wg) {
          load_b(i + 1);
          load_g(i + 1, 0, Set0{});
        }
        if (PH > 1 ? (p == 0 && i + 2 < nt_wg) : (i + 2 < nt_wg)) load_map(i + 2);
        products(p_c);
        buf ^= 1;
      });
    }
  } else if (PH > 1) {
    // register set of item (i, p): p & 1 (PH is even).  Issue order after the barrier = the order the top of the next items waits in
    // (vector-memory loads return in order): contiguous operand of the next tile, map rows, then the gather two items ahead
    for (int i = 0; i < nt_wg; ++i) {
      static_for<PH>([&](auto p_c) {
        constexpr int p = decltype(p_c)::value;
        typedef std::integral_constant<int, (p & 1) ? DEPTH - 1 : 0> Set;
        store_g(buf, Set{});
        if (p == 0) take_b();
        if (p == 1 && i + 2 < nt_wg) store_map2(i + 2);
        if (p == 1 && i + 3 < nt_wg) store_rows(i + 3);
        __syncthreads();
        if (p == PH - 2 && i + 1 < nt_wg) load_b(i + 1);
        if (p == 0 && i + 3 < nt_wg) load_rows(i + 3);
        if (p == 0 && i + 2 < nt_wg) load_map2(i + 2);
        if (p + 2 < PH) load_g(i, p + 2, Set{});
        else if (i + 1 < nt_wg) load_g(i + 1, p + 2 - PH, Set{});
        products(p_c);
        buf ^= 1;
      });
    }
  } else {
    // one phase a tile: item = tile, register set = i & 1 (the loop is unrolled by two), map rows of four tiles (rows of five) in LDS
    auto tile = [&](int i, auto set_c) {
      store_g(buf, set_c);
      take_b();
      if (i >= 1 && i + 2 < nt_wg) store_map2(i + 2);
      if (i >= 1 && i + 3 < nt_wg) store_rows(i + 3);
      __syncthreads();
      if (i + 1 < nt_wg) load_b(i + 1);
      if (i + 4 < nt_wg) load_rows(i + 4);
      if (i + 3 < nt_wg) load_map2(i + 3);
      if (i + 2 < nt_wg) load_g(i + 2, 0, set_c);
      products(std::integral_constant<int, 0>{});
      buf ^= 1;
    };
    for (int i = 0; i < nt_wg; i += 2) {
      tile(i, Set0{});
      if (i + 1 < nt_wg) tile(i + 1, Set1{});
    }
  }
  // slab of this workgroup: part[blockIdx.x][k][ci][co]; D layout of 16x16: col = lane & 15 (contiguous operand's channel),
  // row = (lane >> 4) * 4 + reg (gathered operand's channel)
  float* P = part + (size_t)blockIdx.x * K * Cg_all * Cc_all;
#pragma unroll
  for (int p = 0; p < PH; ++p)
#pragma unroll
    for (int q = 0; q < TPP; ++q) {
      const int l = q * 4 + wave;
      const int nt = l % NT, mt = (l / NT) % MT, kb = l / (NT * MT);
      const int k = kg0 + p * KB + kb;
      const int co = cc0 + nt * 16 + t16;
      if (k >= K || co >= Cc_all) continue;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int ci = cg0 + mt * 16 + g4 * 4 + r;
        float v = acc[0][p * TPP + q][r];
        if (MODE) v = v + (acc[NC > 1 ? 1 : 0][p * TPP + q][r] + acc[NC - 1][p * TPP + q][r]);
        if (!swap) P[((size_t)k * Cg_all + ci) * Cc_all + co] = v;
        else P[((size_t)k * Cc_all + co) * Cg_all + ci] = v;   // the walk is over the layer's INPUT rows: gathered = dOut, contiguous = features
      }
    }
}

template <int MT, int NT, int KB, int PH, int MODE, int DEPTH>
size_t lds_x() {
  constexpr int NPL = MODE ? 3 : 1;
  constexpr int RING = (DEPTH == 2 && PH == 1) ? 4 : 3;
  constexpr int RROW = DEPTH == 2 ? RING + 1 : RING;
  return (size_t)2 * KB * NPL * TMX * (MT * 32 + 16) + (size_t)(RING * TMX * PH * KB + RROW * TMX) * sizeof(int32_t);
}

template <int MT, int NT, int KB, int PH, int MODE, int DEPTH>
void launch_x(dim3 grid, hipStream_t stream, const void* g, const void* c, const int32_t* map, const int32_t* ord, int rows, int K, int cg_all,
              int cc_all, float* part, int swap) {
  static BtcPerDeviceOnce once;
  btc_once_per_device(once, [] {
    (void)hipFuncSetAttribute((const void*)conv_wgrad_x<MT, NT, KB, PH, MODE, DEPTH>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  });
  conv_wgrad_x<MT, NT, KB, PH, MODE, DEPTH><<<grid, 256, lds_x<MT, NT, KB, PH, MODE, DEPTH>(), stream>>>(g, c, map, ord, rows, K, cg_all, cc_all, part, swap);
}

// (KB, PH) per tile shape; PH is halved when the launch has too few (tile, group, block) triples to fill the machine
struct XShape {
  int mt, nt, kb, ph;
};
const XShape X_BF16[] = {{4, 4, 1, 4}, {4, 2, 1, 8}, {2, 4, 2, 4}, {3, 2, 2, 4}, {2, 2, 2, 8}, {2, 1, 2, 8}, {1, 2, 4, 4}, {1, 1, 4, 4}};
const XShape X_SPLIT[] = {{4, 4, 1, 2}, {4, 2, 1, 4}, {2, 4, 2, 2}, {3, 2, 2, 2}, {2, 2, 2, 4}, {2, 1, 2, 8}, {1, 2, 2, 8}};

// the tile shape of a launch: the gathered operand's channels in whole blocks of 16 MT (48 channels: MT = 3, else the largest of 4, 2, 1
// that divides), the contiguous operand's in blocks of the smallest 16 NT that covers them (at most 64)
const XShape* find_shape(int mode, int cg_all, int cc_all) {
  if (cg_all <= 0 || cc_all <= 0 || (cg_all & 15)) return nullptr;
  const XShape* tab = mode ? X_SPLIT : X_BF16;
  const int n = mode ? (int)(sizeof(X_SPLIT) / sizeof(XShape)) : (int)(sizeof(X_BF16) / sizeof(XShape));
  const int nt_want = cc_all <= 16 ? 1 : (cc_all <= 32 ? 2 : 4);
  for (int nt = nt_want; nt >= 1; nt >>= 1)
    for (int i = 0; i < n; ++i) {   // (the tables list the larger gathered blocks first)
      if (tab[i].nt != nt) continue;
      const int cg = tab[i].mt * 16;
      if (cg_all % cg != 0 || (tab[i].mt == 3 && cg_all != 48)) continue;
      return tab + i;
    }
  return nullptr;
}

}  // namespace

// mode: 0 = bf16 activations, 1 = fp32 activations (split operands); cg / cc: channels of the gathered / contiguous operand
bool btc_wgrad_x_supported(int mode, int K, int cg, int cc) {
  if (K > 64 || K < 1) return false;
  return find_shape(mode, cg, cc) != nullptr;
}

// the work split of a launch: -> offset groups, *S = row splits (slabs), *ph = phases per group actually used, *z = channel blocks
int btc_wgrad_x_plan(int mode, int rows, int K, int cg, int cc, int* S, int* ph, int* z) {
  const XShape* sh = find_shape(mode, cg, cc);
  const int t_wgs = btc_tune_get(BTC_TUNE_WGRAD_WGS);
  const int wgs = t_wgs ? t_wgs : 512;
  const int n_tiles = btc_cdiv(rows, TMX);
  const int nz = (cg / (sh->mt * 16)) * btc_cdiv(cc, sh->nt * 16);
  int p = sh->ph;
  if (p > 1 && (long long)n_tiles * btc_cdiv(K, sh->kb * p) * nz < 3LL * wgs) p >>= 1;
  const int groups = btc_cdiv(K, sh->kb * p);
  int s = wgs / (groups * nz);
  if (s > n_tiles / 2) s = n_tiles / 2;
  if (s < 1) s = 1;
  *S = s;
  *ph = p;
  if (z) *z = nz;
  return groups;
}

int btc_launch_wgrad_x(int mode, const void* g, const void* c, const int32_t* map, const int32_t* ord, int rows, int K, int cg, int cc, float* part,
                       int swap, hipStream_t stream) {
  const XShape* sh = find_shape(mode, cg, cc);
  BTC_CHECK_ARG(sh != nullptr && K <= 64, "btc_launch_wgrad_x: unsupported shape %d x %d (mode %d)", cg, cc, mode);
  int S = 1, ph = 1, nz = 1;
  const int groups = btc_wgrad_x_plan(mode, rows, K, cg, cc, &S, &ph, &nz);
  dim3 grid(S, groups, nz);
  const int t_depth = btc_tune_get(BTC_TUNE_WGRAD_X_DEPTH);
  const bool shallow = t_depth ? t_depth == 1 : mode == 1;   // default: two items in flight for bf16 activations, one for split fp32
#define X2(MT_, NT_, KB_, PH_, MODE_)                                                                            \
  do {                                                                                                           \
    if (shallow) launch_x<MT_, NT_, KB_, PH_, MODE_, 1>(grid, stream, g, c, map, ord, rows, K, cg, cc, part, swap);  \
    else launch_x<MT_, NT_, KB_, PH_, MODE_, 2>(grid, stream, g, c, map, ord, rows, K, cg, cc, part, swap);          \
  } while (0)
#define X(MT_, NT_, KB_, PH_, MODE_)                                \
  do {                                                              \
    if (ph == PH_) X2(MT_, NT_, KB_, PH_, MODE_);                   \
    else X2(MT_, NT_, KB_, (PH_ / 2), MODE_);                       \
  } while (0)
  const int mt = sh->mt, nt = sh->nt;
  if (mode == 0) {
    if (mt == 1 && nt == 1) X(1, 1, 4, 4, 0);
    else if (mt == 2 && nt == 1) X(2, 1, 2, 8, 0);
    else if (mt == 1 && nt == 2) X(1, 2, 4, 4, 0);
    else if (mt == 2 && nt == 2) X(2, 2, 2, 8, 0);
    else if (mt == 3 && nt == 2) X(3, 2, 2, 4, 0);
    else if (mt == 2 && nt == 4) X(2, 4, 2, 4, 0);
    else if (mt == 4 && nt == 2) X(4, 2, 1, 8, 0);
    else X(4, 4, 1, 4, 0);
  } else {
    if (mt == 2 && nt == 1) X(2, 1, 2, 8, 1);
    else if (mt == 1 && nt == 2) X(1, 2, 2, 8, 1);
    else if (mt == 2 && nt == 2) X(2, 2, 2, 4, 1);
    else if (mt == 3 && nt == 2) X(3, 2, 2, 2, 1);
    else if (mt == 2 && nt == 4) X(2, 4, 2, 2, 1);
    else if (mt == 4 && nt == 2) X(4, 2, 1, 4, 1);
    else X(4, 4, 1, 2, 1);
  }
#undef X
#undef X2
  BTC_LAUNCH_CHECK();
  return BTC_OK;
}
