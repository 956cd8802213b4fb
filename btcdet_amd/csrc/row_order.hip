// Row-order hints for the sparse-conv apply / weight-gradient kernels.
//
// The output-stationary kernels (conv_apply_g, conv_apply_b, conv_wgrad_rows) work on tiles of 16 / 64 consecutive rows of a
// neighbour map and must process every kernel offset that ANY row of the tile has.  In map order (rows sorted by voxel
// coordinate) a 16-row tile of a stride-2 layer's dgrad map uses 28-30 % of the (row, offset) slots it pays for: an input row
// at coordinate c reaches output (c + p - k) / s only through offsets k = (c + p) mod s, so consecutive rows alternate between
// the 8 residue classes, each of which sees a different 1/8 of the 27 offsets.  A tile of the forward map uses 17-25 %, a
// 3x3x3 SubM tile 38-64 % (full-size golden scene, tests/golden/btc_full_a.npz).  Which rows share a tile does not change any
// result (a row's sum only involves its own map row), so the kernels accept a permutation `order` of the rows and tile THAT.
//
// The permutation: inside blocks of ORDER_BLK = 2048 consecutive rows, a STABLE counting sort of the rows by their first
// present offset (the lowest k with nbr[row][k] >= 0; K for a row without neighbours).  Rows of different residue classes
// have disjoint offset sets, hence different first offsets: every group is class-pure, and a strided dgrad tile is 96 % full
// (a device-wide sort by the full offset mask reaches 100 % but cost 95 us per chain, and a block-local bitonic sort of the
// masks 106 us -- half of what either bought; this kernel reads the maps once and does three block-wide passes).  Strided
// forward maps go from 17-25 % to 35-38 %, SubM maps gain little (38-64 % -> 46-69 %) and are not ordered by the callers.
// Block-local also keeps the rows of a tile within 2048 rows of each other in coordinate order (the gathers' L2 locality).
//
// The order is a pure function of the map (stable sort): same map -> same order, so a weight gradient walked in this order
// stays run-to-run deterministic.
#include "btc_common.h"

namespace {

constexpr int MAX_MAPS = BTC_ROW_ORDER_MAX_MAPS;
constexpr int ORDER_BLK = 2048;   // rows per sort block
constexpr int ORDER_T = 512;      // 8 waves, each owns 256 consecutive rows of the block
constexpr int ORDER_W = ORDER_T / 64;
constexpr int ORDER_BINS = 65;    // first offsets 0 .. 63, and "no neighbour"

struct OrderJobs {
  const int32_t* nbr[MAX_MAPS];
  const int32_t* first[MAX_MAPS];   // per map: the rows' first present offsets when the rulebook fill produced them (rulebook.hip), else NULL

  int n[MAX_MAPS];
  int K[MAX_MAPS];
  int off[MAX_MAPS];       // first element of the map's order in the output
  int blk0[MAX_MAPS + 1];  // first workgroup of the map
  int n_maps;
};

__global__ __launch_bounds__(ORDER_T) void order_local(OrderJobs jobs, int32_t* __restrict__ order) {
  __shared__ int s_first[ORDER_BLK];              // first present offset of each row
  __shared__ int s_cnt[ORDER_BINS * ORDER_W];     // [bin][wave]: rows of the wave's 256 in the bin, then their first output slot
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int j = 0;
  while (j + 1 < jobs.n_maps && (int)blockIdx.x >= jobs.blk0[j + 1]) ++j;   // workgroup-uniform
  const int chunk = blockIdx.x - jobs.blk0[j];
  const int n = jobs.n[j], K = jobs.K[j];
  const int row0 = chunk * ORDER_BLK;
  const int rows = min(ORDER_BLK, n - row0);
  for (int r = tid; r < ORDER_BLK; r += ORDER_T) s_first[r] = K;
  for (int e = tid; e < ORDER_BINS * ORDER_W; e += ORDER_T) s_cnt[e] = 0;
  __syncthreads();
  if (jobs.first[j]) {   // 4 bytes a row instead of 4 K: the keys came out of the rulebook fill
    const int32_t* f = jobs.first[j] + row0;
    for (int r = tid; r < rows; r += ORDER_T) s_first[r] = min(f[r], K);   // (a scattered map's untouched keys hold a large value)
  } else {
    const int32_t* base = jobs.nbr[j] + (size_t)row0 * K;   // rows * K consecutive ints: coalesced
    const int total = rows * K;
    for (int e0 = tid; e0 < total; e0 += 8 * ORDER_T) {     // 8 loads in flight per thread (the walk is latency bound otherwise)
      int v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = (e0 + u * ORDER_T < total) ? base[e0 + u * ORDER_T] : -1;
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (v[u] >= 0) {
          const int e = e0 + u * ORDER_T, r = e / K;
          atomicMin(&s_first[r], e - r * K);
        }
    }
  }
  __syncthreads();
  // wave w owns rows [256 w, 256 w + 256) of the block, 64 at a time in row order (ORDER_BLK / ORDER_W == 256)
  for (int it = 0; it < ORDER_BLK / ORDER_T; ++it) {
    const int r = wave * (ORDER_BLK / ORDER_W) + it * 64 + lane;
    if (r < rows) atomicAdd(&s_cnt[s_first[r] * ORDER_W + wave], 1);
  }
  __syncthreads();
  if (wave == 0) {   // exclusive scan of s_cnt in (bin, wave) order: lane l takes PER consecutive entries
    constexpr int PER = (ORDER_BINS * ORDER_W + 63) / 64;
    int loc[PER], sum = 0;
#pragma unroll
    for (int q = 0; q < PER; ++q) {
      const int e = lane * PER + q;
      loc[q] = e < ORDER_BINS * ORDER_W ? s_cnt[e] : 0;
      sum += loc[q];
    }
    int incl = sum;   // inclusive wave scan of the per-lane sums
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const int v = __shfl_up(incl, d, 64);
      if (lane >= d) incl += v;
    }
    int run = incl - sum;
#pragma unroll
    for (int q = 0; q < PER; ++q) {
      const int e = lane * PER + q;
      if (e < ORDER_BINS * ORDER_W) s_cnt[e] = run;
      run += loc[q];
    }
  }
  __syncthreads();
  int32_t* out = order + (size_t)jobs.off[j] + row0;
  for (int it = 0; it < ORDER_BLK / ORDER_T; ++it) {
    const int r = wave * (ORDER_BLK / ORDER_W) + it * 64 + lane;
    const bool live = r < rows;
    const int bin = live ? s_first[r] : -1;
    unsigned long long todo = __ballot(live);
    while (todo) {   // one round per distinct bin among the wave's 64 rows, lowest lane first (wave-uniform loop)
      const int leader = __ffsll((long long)todo) - 1;
      const int b = __shfl(bin, leader, 64);
      const unsigned long long same = __ballot(bin == b);
      if (bin == b) {
        const int rank = __popcll(same & ((1ull << lane) - 1ull));
        out[s_cnt[b * ORDER_W + wave] + rank] = row0 + r;
      }
      __builtin_amdgcn_wave_barrier();   // every lane has read the slot base before the leader moves it
      if (lane == leader) s_cnt[b * ORDER_W + wave] += __popcll(same);
      todo &= ~same;
    }
  }
}

}  // namespace

extern "C" int btc_row_orders(const int32_t* const* nbrs, const int32_t* n_rows, const int32_t* Ks, int n_maps, int32_t* order, void* stream) {
  return btc_row_orders_keyed(nbrs, nullptr, n_rows, Ks, n_maps, order, stream);
}

extern "C" int btc_row_orders_keyed(const int32_t* const* nbrs, const int32_t* const* firsts, const int32_t* n_rows, const int32_t* Ks, int n_maps,
                                    int32_t* order, void* stream) {
  BTC_CHECK_ARG(n_maps >= 1 && n_maps <= MAX_MAPS, "btc_row_orders: 1..%d maps per call (got %d)", MAX_MAPS, n_maps);
  OrderJobs jobs;
  long long total = 0;
  int blocks = 0;
  for (int j = 0; j < n_maps; ++j) {
    BTC_CHECK_ARG(n_rows[j] >= 0 && Ks[j] >= 1 && Ks[j] <= ORDER_BINS - 1, "btc_row_orders: map %d: n=%d K=%d (K <= %d)", j, n_rows[j], Ks[j],
                  ORDER_BINS - 1);
    BTC_CHECK_ARG(n_rows[j] == 0 || nbrs[j] != nullptr || (firsts && firsts[j]), "btc_row_orders: map %d is NULL", j);
    jobs.nbr[j] = nbrs[j];
    jobs.first[j] = firsts ? firsts[j] : nullptr;
    jobs.n[j] = n_rows[j];
    jobs.K[j] = Ks[j];
    jobs.off[j] = (int)total;
    jobs.blk0[j] = blocks;
    total += n_rows[j];
    blocks += btc_cdiv(n_rows[j], ORDER_BLK);
    BTC_CHECK_ARG(total < (1LL << 31), "btc_row_orders: %lld rows in one call", total);
  }
  jobs.blk0[n_maps] = blocks;
  jobs.n_maps = n_maps;
  if (blocks == 0) return BTC_OK;
  BTC_CHECK_ARG(order != nullptr, "btc_row_orders: order is NULL");
  order_local<<<blocks, ORDER_T, 0, (hipStream_t)stream>>>(jobs, order);
  BTC_LAUNCH_CHECK();
  return BTC_OK;
}
