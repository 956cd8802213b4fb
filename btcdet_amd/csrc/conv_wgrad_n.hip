// Weight gradient of a sparse convolution whose RESULT side is narrow (the occupancy head: 32 -> 5 channels at 210-260 K rows), round 6.
//
//     dW[k][ci][co] = sum_i  x[nbr_out[i][k]][ci] * dy[i][co]  =  sum_j  x[j][ci] * dy[nbr_in[j][k]][co]
//
// The general kernels (conv_wgrad_x.hip, conv_wgrad_rows_p) walk the first form: every source row x[j] (Cin channels: 128 bytes and up) is
// gathered once per offset that reaches it -- 24 of 27 on the dense occupancy level -- and, on the bf16 pipe, split into three planes
// every time; the 5 result channels ride in a 16-column block that is two thirds padding.  That launch was the longest of the whole step
// (305 us; the same layer's forward is 113 us, its data gradient 90).  This kernel walks the SECOND form over the layer's input rows:
//   * x is read ONCE, contiguously (a lane's value of the A operand is one coalesced element load);
//   * what is gathered is dy -- rows of <= 8 channels, 4.5 MB for the whole level: resident in the L2s;
//   * the K offsets and the Cout result channels are ONE matrix dimension: column c = k' * Cout + co of a (rows x K Cout) operand
//     G[j][c] = dy[map[j][k']][co] that exists only in registers -- 135 columns for 27 x 5, nine 16-column tiles, nothing padded per
//     offset -- so the whole dW is a (Cin x rows) . (rows x K Cout) product on `v_mfma_f32_16x16x4_f32` (exact fp32 products, fp32
//     accumulation: the arithmetic of the fp32-pipe kernels, no split);
//   * no LDS and no barrier in the walk: a wave owns every (4 S)-th group of 4 rows, keeps its 16 MT x 16 NTL block of dW in accumulators,
//     has the map values of the groups up to five steps ahead and the operand values of the next three groups in flight during a group's
//     products (51 loads in flight per wave: one wave a SIMD measured faster than two).
// A layer with a narrow INPUT (the 4- / 6-channel first layers: 52 us for 36 K rows on the fp32-pipe kernel) is the same walk with the roles
// exchanged: over its output rows (dOut contiguous), the features gathered through nbr_out, columns (k, ci), slab written [k][ci][co].
// The backward map of a submanifold layer is its forward map with the offset index mirrored (rulebook.hip): `mirror` reads column k' of
// nbr_out as nbr_in's column K-1-k', i.e. writes dW[K-1-k'].  The four waves of a workgroup add their blocks in wave order through LDS,
// one slab per workgroup, slabs added in index order by wgrad_reduce / btc_wgrad_reduce_multi: deterministic.
#include "btc_common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int NTL_MAX = 11;                 // 16-column tiles of the (k', narrow channel) dimension: 9 (K Cn <= 144) or 11 (<= 176) per instance
constexpr int NW = 4;                       // waves per workgroup; one slab per workgroup (8 waves on 256 workgroups measured 57 against 48 us)
#define N_RECORDS 0xFFFFFF00u

template <int MT, int NTL, bool BF>
__global__ __launch_bounds__(NW * 64, 2) void conv_wgrad_n(const void* __restrict__ xsrc, const void* __restrict__ dysrc, const int32_t* __restrict__ map,
                                                    int n_rows, int K, int Cx_all, int Cy, float* __restrict__ part, int flags) {
  // xsrc: the layer's input rows (n_rows x Cx_all), walked contiguously; dysrc: the result side's gradient rows (Cy channels), gathered
  // through map (n_rows x K: for input row j and column k' the result row, or -1).  blockIdx.y: block of 16 MT input channels.
  // flags bit 0: mirrored map (above); bit 1: the roles are the other way round -- a layer with a narrow INPUT (the 4- / 6-channel first
  // layers): the walk is over its OUTPUT rows (xsrc = dOut, contiguous), the gathered rows are its input features through nbr_out, the
  // columns are (k, ci) and the slab is written as [k][narrow][walked].
  const bool mirror = flags & 1, narrow_in = flags & 2;
  constexpr unsigned ESZ = BF ? 2u : 4u;
  const int tid = threadIdx.x, lane = tid & 63, g4 = lane >> 4, t16 = lane & 15;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // (wave-uniform: the walk's loop counter lives in scalar registers)
  const int cx0 = blockIdx.y * (16 * MT);
  const int n_cols = K * Cy;
  const unsigned yrow = (unsigned)Cy * ESZ;
  const int n_groups = (n_rows + 3) >> 2;
  const int stride = gridDim.x * NW;
  const int w = blockIdx.x * NW + wave;

  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)xsrc, 0, N_RECORDS, 0x00020000);
  const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc((void*)dysrc, 0, N_RECORDS, 0x00020000);
  const __amdgpu_buffer_rsrc_t rm = __builtin_amdgcn_make_buffer_rsrc((void*)map, 0, N_RECORDS, 0x00020000);

  // this lane's column of every tile: byte offset of its map column inside a map row, byte offset of its channel inside a dy row.
  // A column past K Cy (the last tile's tail) walks as column 0 does: the columns of a product are independent and the slab write
  // below leaves those out -- no mask in the loop.
  unsigned mcol[NTL], ycol[NTL];
#pragma unroll
  for (int nt = 0; nt < NTL; ++nt) {
    const int c = nt * 16 + t16 < n_cols ? nt * 16 + t16 : 0;
    const int kk = c / Cy;
    mcol[nt] = (unsigned)kk * 4u;
    ycol[nt] = (unsigned)(c - kk * Cy) * ESZ;
  }

  f32x4 acc[MT][NTL];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NTL; ++nt) acc[mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // a group = 4 consecutive rows; lane group g4 holds row 4 g + g4 of both operands (the reduction index of the 16x16x4 product).
  // Every load is a raw buffer load and every "not there" an out-of-range offset (zeros come back, nothing is fetched, no branch):
  //   a row past the end  -> its map row starts at N_RECORDS (K <= 64: every column of it is out of range), its x and dy offsets are
  //                          or-ed with all ones;
  //   an absent neighbour -> map value -1: its sign, smeared over the word, is or-ed into the dy offset.
  // So the walk has no tail handling: a group past the end multiplies zeros.
  // Pipeline: the map values of group i are requested PM steps ahead of its products, its operand values PO steps ahead (from the map
  // values, which have had PM - PO steps to arrive).  Rings of RM = PM - PO map sets and RO = PO + 1 operand sets; the loop is unrolled
  // over the rings, so every set index is a compile-time constant and the compiler's vmcnt waits are exact.  Loads in flight per wave:
  // PO (NTL + MT) + (PM - PO) NTL = 51 with MT = 2 -- the counter holds 63.
  constexpr int PO = 3, PM = 5, RM = PM - PO, RO = PO + 1, UN = RO % RM == 0 ? RO : RO * RM;
  int mv[RM][NTL];
  float yv[RO][NTL], xv[RO][MT];
  auto load_m = [&](int g, int set) {
    const int j = 4 * g + g4;
    const unsigned row = (g < n_groups && j < n_rows) ? (unsigned)j * (unsigned)K * 4u : N_RECORDS;
#pragma unroll
    for (int nt = 0; nt < NTL; ++nt) mv[set][nt] = __builtin_amdgcn_raw_buffer_load_b32(rm, row + mcol[nt], 0, 0);
  };
  auto load_yx = [&](int g, int mset, int set) {
    const int j = 4 * g + g4;
    const bool live = g < n_groups && j < n_rows;
    const unsigned dead = live ? 0u : 0xFFFFFFFFu;
#pragma unroll
    for (int nt = 0; nt < NTL; ++nt) {
      const int nb = mv[mset][nt];
      const unsigned off = ((unsigned)nb * yrow + ycol[nt]) | (unsigned)(nb >> 31) | dead;
      if (BF) yv[set][nt] = __uint_as_float((unsigned)__builtin_amdgcn_raw_buffer_load_b16(ry, off, 0, 0) << 16);
      else yv[set][nt] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(ry, off, 0, 0));
    }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const unsigned off = (((unsigned)j * (unsigned)Cx_all + (unsigned)(cx0 + mt * 16 + t16)) * ESZ) | dead;
      if (BF) xv[set][mt] = __uint_as_float((unsigned)__builtin_amdgcn_raw_buffer_load_b16(rx, off, 0, 0) << 16);
      else xv[set][mt] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rx, off, 0, 0));
    }
  };
  auto products = [&](int set) {
#pragma unroll
    for (int nt = 0; nt < NTL; ++nt)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(xv[set][mt], yv[set][nt], acc[mt][nt], 0, 0, 0);
  };

  // group of step i: w + i * stride; its map set i % RM, its operand set i % RO
  int g = w;
#pragma unroll
  for (int i = 0; i < PO; ++i) {           // steps -PO .. -1 of the walk: operands of groups 0 .. PO-1 (their map values first)
    load_m(g + i * stride, i % RM);
    load_yx(g + i * stride, i % RM, i % RO);
  }
#pragma unroll
  for (int i = PO; i < PM; ++i) load_m(g + i * stride, i % RM);
  if (g < n_groups) {
    do {   // (bottom-tested: with the test at the top the compiler copies all 72 accumulator registers out for the exit edge every step)
#pragma unroll
      for (int u = 0; u < UN; ++u) {
        load_yx(g + (u + PO) * stride, (u + PO) % RM, (u + PO) % RO);
        load_m(g + (u + PM) * stride, (u + PM) % RM);
        products(u % RO);
      }
      g += UN * stride;
    } while (g < n_groups);
  }

  // ---- the workgroup's slab: wave 0's block, + wave 1's, ... + wave NW-1's (that order), through LDS; then every thread writes a
  // contiguous piece of it (layout [k][ci][co], this workgroup's 16 MT input channels)
  __shared__ float red[MT * NTL * 4 * 64];
#pragma unroll 1
  for (int s = 0; s < NW; ++s) {
    if (wave == s) {
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NTL; ++nt)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float* p = red + ((mt * NTL + nt) * 4 + r) * 64 + lane;
            *p = s > 0 ? *p + acc[mt][nt][r] : acc[mt][nt][r];
          }
    }
    __syncthreads();
  }
  // D layout of a 16 x 16 tile: column = lane & 15 (the (k', narrow channel) column), row = (lane >> 4) * 4 + reg (the walked side's channel)
  float* P = part + (size_t)blockIdx.x * K * Cx_all * Cy;
  const int per_k = 16 * MT * Cy;           // this workgroup's elements of one dW[k] ([walked][narrow]: contiguous in the slab)
  for (int e = tid; e < K * per_k; e += NW * 64) {
    const int k = e / per_k, f = e - k * per_k, cil = f / Cy, cn = f - cil * Cy;
    const int c = (mirror ? K - 1 - k : k) * Cy + cn;
    const int mt = cil >> 4, r16 = cil & 15;
    const float v = red[((mt * NTL + (c >> 4)) * 4 + (r16 & 3)) * 64 + (r16 >> 2) * 16 + (c & 15)];
    if (!narrow_in) P[((size_t)k * Cx_all + cx0) * Cy + f] = v;
    else P[((size_t)k * Cy + cn) * Cx_all + cx0 + cil] = v;
  }
}

}  // namespace

// a layer with a narrow RESULT (Cout <= 8: walked over its input rows, Cin a multiple of 16) or a narrow INPUT (Cin <= 8: walked over its
// output rows, Cout a multiple of 16): -> 1 / 2, or 0
int btc_wgrad_n_kind(int K, int Cin, int Cout) {
  if (K < 1 || K > 64 || Cin < 1 || Cout < 1) return 0;   // (K <= 64: a dead row's map offsets, N_RECORDS + 4 k', must stay below 2^32)
  if (Cout <= 8 && K * Cout <= 16 * NTL_MAX && Cin >= 16 && (Cin & 15) == 0) return 1;
  if (Cin <= 8 && K * Cin <= 16 * NTL_MAX && Cout >= 16 && (Cout & 15) == 0) return 2;
  return 0;
}
bool btc_wgrad_n_supported(int K, int Cin, int Cout) { return btc_wgrad_n_kind(K, Cin, Cout) == 1; }

// slabs (= workgroups along x) of a launch over `rows` walked rows
int btc_wgrad_n_plan(int rows) {
  const int t_wgs = btc_tune_get(BTC_TUNE_WGRAD_WGS);
  // one workgroup per CU: every slab is 4 K Cin Cout bytes written and read again, and one more term of the reduction's chain
  // (256 / 512 / 768 slabs: 48 / 61 / 77 us at 210 K rows)
  int s = t_wgs ? t_wgs : 256;
  const int n_groups = btc_cdiv(rows, 4);
  if (s > n_groups / (8 * NW)) s = n_groups / (8 * NW);   // at least 8 groups a wave
  return s < 1 ? 1 : s;
}

// walked: rows of Cw channels (a multiple of 16), read once; gathered: rows of Cn <= 8 channels through map (rows x K); flags as the kernel's
int btc_launch_wgrad_n(bool bf, const void* walked, const void* gathered, const int32_t* map, int rows, int K, int Cw, int Cn, float* part, int flags,
                       hipStream_t stream) {
  BTC_CHECK_ARG(K >= 1 && K <= 64 && Cn >= 1 && Cn <= 8 && K * Cn <= 16 * NTL_MAX && Cw >= 16 && (Cw & 15) == 0,
                "btc_launch_wgrad_n: unsupported shape %d x %d, K = %d", Cw, Cn, K);
  const int S = btc_wgrad_n_plan(rows);
  const bool wide = K * Cn > 144, two = (Cw & 31) == 0;
  dim3 grid(S, Cw / (two ? 32 : 16));
#define N_LAUNCH(MT_, NTL_)                                                                                                            \
  do {                                                                                                                                 \
    if (bf) conv_wgrad_n<MT_, NTL_, true><<<grid, NW * 64, 0, stream>>>(walked, gathered, map, rows, K, Cw, Cn, part, flags);          \
    else conv_wgrad_n<MT_, NTL_, false><<<grid, NW * 64, 0, stream>>>(walked, gathered, map, rows, K, Cw, Cn, part, flags);            \
  } while (0)
  if (two && !wide) N_LAUNCH(2, 9);
  else if (two) N_LAUNCH(2, 11);
  else if (!wide) N_LAUNCH(1, 9);
  else N_LAUNCH(1, 11);
#undef N_LAUNCH
  BTC_LAUNCH_CHECK();
  return BTC_OK;
}
