// Rulebook (neighbour-map) construction for gfx950.
//
// Replaces spconv ops.get_indice_pairs behind SubMConv3d / SparseConv3d /
// SparseConvTranspose3d / SparseMaxPool3d (/root/reference/btcdet/models/backbones_3d/
// spconv_backbone.py:12-29).  Instead of spconv's (2,K,N) pair lists claimed with atomicAdd
// (nondeterministic order) the rulebook is stored as two dense neighbour maps
//     nbr_out (n_out,K): input row gathered by output row i at offset k (or -1)
//     nbr_in  (n_in ,K): output row fed by input row j at offset k (or -1)
// which are a pure function of (indices, geometry) and feed the output-stationary conv kernels directly.
//
// Data structure: a LEVEL = the active set of one resolution as a RANKED BITMAP over its grid
//     words   : 1 bit per cell (cell = b * vol + (z * H + y) * W + x), 32-byte blocks of 8 words
//     bprefix : per block, the number of set bits in front of it inside its 2048-word chunk
//     cprefix : per chunk, the number of set bits in front of it (cprefix[nchunks] = total)
// rank(cell) = cprefix[chunk] + bprefix[block] + popcount inside the block = the ROW of the cell, because every level
// built here emits its rows in ascending cell order ((b,z,y,x)-sorted, what spconv's sort-unique yields).  One structure
// answers every question of the stage: the sorted unique output set of a strided / transposed conv or pool (mark the
// reachable cells, rank them -- no sort), cell -> output row for nbr_in, and cell -> row for the 27 neighbour probes of
// every submanifold layer on that level (a bit test that mostly fails, then one 32-byte block + two prefix words).
// Cost per level: grid volume / 8 bytes cleared and scanned once (det level [21,800,704] x 2: 2.9 MB), everything else is
// proportional to the active rows.  The arbitrary (unsorted, un-ranked) input of a chain -- voxelizer order -- is served by
// an open-addressing hash with 64-bit cell keys instead.
//
// A whole chain of layers (an encoder / decoder branch) is built in two phases around ONE read-back (btc_chain_levels /
// btc_chain_maps): phase A builds every level on the device, each level's row count staying in device memory and the rows
// of level l marking level l+1 with their count read from device memory; the host reads all counts at once, sizes the maps,
// and phase B fills every neighbour map of the chain in one multi-job launch.
#include "btc_common.h"

namespace {

constexpr int RB_T = 256;
constexpr int RB_BLK = 8;           // words per rank block (one 32-byte sector)
constexpr int RB_CHUNK = 256;       // blocks per chunk: one thread per block in rb_scan
constexpr int RB_MAX_JOBS = 17;      // (a chain of the hot path is 13-14 jobs: one launch; the table travels in the kernel arguments, 4 KB at most)

struct Level {
  unsigned* words;
  int32_t* bprefix;
  int32_t* cprefix;
  int shape[3];
  int vol;             // cells per scene
  long long nblk;      // 8-word blocks
};

__host__ __device__ __forceinline__ long long lvl_cell(const Level& L, int b, int z, int y, int x) {
  return (long long)b * L.vol + ((long long)z * L.shape[1] + y) * L.shape[2] + x;
}

// rank of a cell of a scanned level, -1 if the cell is not active
__device__ __forceinline__ int lvl_rank(const Level& L, long long cell) {
  const long long w = cell >> 5;
  const unsigned bit = (unsigned)cell & 31u;
  const unsigned word = L.words[w];
  if (!((word >> bit) & 1u)) return -1;
  const long long blk = w >> 3;
  const int wi = (int)(w & 7);
  const uint4* p = reinterpret_cast<const uint4*>(L.words + blk * RB_BLK);
  const uint4 a = p[0], b = p[1];
  const unsigned ws[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
  int r = L.cprefix[blk / RB_CHUNK] + L.bprefix[blk] + __popc(word & ((1u << bit) - 1u));
#pragma unroll
  for (int j = 0; j < 7; ++j) r += (j < wi) ? __popc(ws[j]) : 0;
  return r;
}

// integer division by the small runtime constants of a geometry is ~40 instructions on this ISA: the common values
// (stride 1 / 2, 3x3x3 kernels) take shift / multiply paths
__device__ __forceinline__ bool div_stride(int t, int s, int* q) {
  if (s == 1) { *q = t; return true; }
  if (s == 2) { *q = t >> 1; return !(t & 1); }
  const int v = t / s;
  *q = v;
  return v * s == t;
}

__device__ __forceinline__ void split_offset(const BtcGeom& g, int kk, int* kz, int* ky, int* kx) {
  if (g.k[2] == 3 && g.k[1] == 3) {
    *kx = kk % 3; *ky = (kk / 3) % 3; *kz = kk / 9;
  } else if (g.k[2] == 1 && g.k[1] == 1) {
    *kx = 0; *ky = 0; *kz = kk;
  } else {
    *kx = kk % g.k[2]; *ky = (kk / g.k[2]) % g.k[1]; *kz = kk / (g.k[2] * g.k[1]);
  }
}

__device__ __forceinline__ void split_item(long long t, int K, int* i, int* kk) {
  if (K == 27) { const long long q = t / 27; *i = (int)q; *kk = (int)(t - q * 27); }
  else if (K == 1) { *i = (int)t; *kk = 0; }
  else { const long long q = t / K; *i = (int)q; *kk = (int)(t - q * K); }
}

// one axis of the forward map: input coordinate + kernel offset -> output coordinate (CONV: divisibility; TRANSPOSE: integral)
__device__ __forceinline__ bool fwd_axis(const BtcGeom& g, int j, int c, int kv, int* o) {
  int q;
  if (g.mode == BTC_MODE_CONV) {
    const int t = c + g.p[j] - kv * g.d[j];
    if (t < 0 || !div_stride(t, g.s[j], &q)) return false;
  } else {
    q = c * g.s[j] - g.p[j] + kv * g.d[j];
  }
  *o = q;
  return q >= 0 && q < g.out_shape[j];
}

__device__ __forceinline__ bool fwd_cell(const BtcGeom& g, int z, int y, int x, int kk, int* oz, int* oy, int* ox) {
  int kz, ky, kx;
  split_offset(g, kk, &kz, &ky, &kx);
  return fwd_axis(g, 0, z, kz, oz) && fwd_axis(g, 1, y, ky, oy) && fwd_axis(g, 2, x, kx, ox);
}

__device__ __forceinline__ void or_word(unsigned* words, long long w, unsigned bits) {
  if ((__hip_atomic_load(&words[w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & bits) != bits) atomicOr(&words[w], bits);
}

// every cell an input row reaches along one (kz, ky) line of the kernel, OR-ed into the output level's bitmap; the bits of
// one word are merged in registers, and a word that already holds them is not touched again (stride 2: ~8 inputs share an
// output cell).  The axes are resolved once each (no per-offset divisions).
__device__ __forceinline__ void mark_line(const BtcGeom& g, const Level& out, int b, int z, int y, int x, int kz, int ky) {
  int oz, oy;
  if (!fwd_axis(g, 0, z, kz, &oz) || !fwd_axis(g, 1, y, ky, &oy)) return;
  const long long line = lvl_cell(out, b, oz, oy, 0);
  long long cur_w = -1;
  unsigned cur_bits = 0;
  for (int kx = 0; kx < g.k[2]; ++kx) {
    int ox;
    if (!fwd_axis(g, 2, x, kx, &ox)) continue;
    const long long cell = line + ox;
    const long long w = cell >> 5;
    if (w != cur_w) {
      if (cur_bits) or_word(out.words, cur_w, cur_bits);
      cur_w = w;
      cur_bits = 0;
    }
    cur_bits |= 1u << ((unsigned)cell & 31u);
  }
  if (cur_bits) or_word(out.words, cur_w, cur_bits);
}

// n rows: from d_n (device, the count of the producing level) when given, else n_host.  A workgroup takes MARK_ROWS consecutive
// rows.  Rows of a level built here are sorted by cell, i.e. neighbours in space, and under a stride-2 layer ~8 of them reach the
// same output cell: marked straight into the global bitmap, the 9 x rows device-scope atomic ORs of such a workgroup pile up on a
// few dozen words (the 40 K-row occupancy level: 53 us for this launch).  So when the workgroup's rows lie in one (batch, z) plane
// it ORs into an LDS window of the <= 3 output planes x the y-range the rows can reach, and flushes the non-zero words with one
// global atomic each.  `sorted` = 0 (the chain's arbitrary-order input level) and windows that do not fit mark directly.
constexpr int MARK_ROWS = 32;    // x (kz, ky) lines = 288 items for 256 threads; small enough that a 3 K-row level still spreads over ~100 workgroups
constexpr int MARK_WIN = 384;   // bitmap words per staged output plane

__global__ __launch_bounds__(RB_T) void rb_mark(const int4* __restrict__ idx, int n_host, const int32_t* __restrict__ d_n, BtcGeom g, Level out,
                                                int sorted) {
  __shared__ unsigned s_win[3][MARK_WIN];
  __shared__ long long s_w0[3];
  __shared__ int s_cnt[3];
  __shared__ int s_stage;
  const int n = d_n ? *d_n : n_host;
  const int tid = threadIdx.x;
  // (the grid is sized from a host-side BOUND on the level's rows, capped: the workgroups stride over the real row tiles)
  for (int r0 = blockIdx.x * MARK_ROWS; r0 < n; r0 += gridDim.x * MARK_ROWS) {
  const int rows = n - r0 < MARK_ROWS ? n - r0 : MARK_ROWS;
  __syncthreads();   // the previous tile's flush has read the window
  if (tid == 0) {
    int stage = 0;
    if (sorted && g.k[0] <= 3 && rows >= 8) {
      const int4 a = idx[r0], b = idx[r0 + rows - 1];
      if (a.x == b.x && a.y == b.y) {
        // y-range of the output cells the rows can reach (fwd_axis over y in [a.z, b.z])
        int ylo, yhi;
        if (g.mode == BTC_MODE_CONV) {
          const int t0 = a.z + g.p[1] - (g.k[1] - 1) * g.d[1];
          ylo = t0 <= 0 ? 0 : t0 / g.s[1];
          yhi = (b.z + g.p[1]) / g.s[1];
        } else {
          ylo = a.z * g.s[1] - g.p[1];
          yhi = b.z * g.s[1] - g.p[1] + (g.k[1] - 1) * g.d[1];
        }
        ylo = ylo < 0 ? 0 : ylo;
        yhi = yhi >= g.out_shape[1] ? g.out_shape[1] - 1 : yhi;
        stage = 1;
        for (int kz = 0; kz < 3; ++kz) {
          int oz;
          s_cnt[kz] = 0;
          s_w0[kz] = 0;
          if (kz >= g.k[0] || ylo > yhi || !fwd_axis(g, 0, a.y, kz, &oz)) continue;
          const long long w_lo = lvl_cell(out, a.x, oz, ylo, 0) >> 5, w_hi = lvl_cell(out, a.x, oz, yhi, out.shape[2] - 1) >> 5;
          if (w_hi - w_lo + 1 > MARK_WIN) { stage = 0; break; }
          s_w0[kz] = w_lo;
          s_cnt[kz] = (int)(w_hi - w_lo + 1);
        }
      }
    }
    s_stage = stage;
  }
  for (int e = tid; e < 3 * MARK_WIN; e += RB_T) (&s_win[0][0])[e] = 0u;
  __syncthreads();
  const bool staged = s_stage != 0;
  const int lines = g.k[0] * g.k[1];
  if (!staged) {
    for (int e = tid; e < rows * lines; e += RB_T) {
      const int il = e / lines, l = e - il * lines;
      const int kz = l / g.k[1], ky = l - kz * g.k[1];
      const int4 c = idx[r0 + il];
      mark_line(g, out, c.x, c.y, c.z, c.w, kz, ky);
    }
    continue;
  }
  for (int e = tid; e < rows * lines; e += RB_T) {
    const int il = e / lines, l = e - il * lines;
    const int kz = l / g.k[1], ky = l - kz * g.k[1];
    const int4 c = idx[r0 + il];
    int oz, oy;
    if (!fwd_axis(g, 0, c.y, kz, &oz) || !fwd_axis(g, 1, c.z, ky, &oy)) continue;
    const long long line = lvl_cell(out, c.x, oz, oy, 0);
    for (int kx = 0; kx < g.k[2]; ++kx) {
      int ox;
      if (!fwd_axis(g, 2, c.w, kx, &ox)) continue;
      const long long cell = line + ox;
      atomicOr(&s_win[kz][(int)((cell >> 5) - s_w0[kz])], 1u << ((unsigned)cell & 31u));
    }
  }
  __syncthreads();
  for (int e = tid; e < 3 * MARK_WIN; e += RB_T) {
    const int pz = e / MARK_WIN, j = e - pz * MARK_WIN;
    if (j >= s_cnt[pz]) continue;
    const unsigned bits = s_win[pz][j];
    if (bits) or_word(out.words, s_w0[pz] + j, bits);
  }
  }
}

// rb_mark for a level that was BUILT here (a ranked-bitmap level): walk the input level's BITMAP instead of its row list -- a set bit is
// a row, its coordinates follow from the cell index -- so marking level l + 1 needs neither level l's rows (rb_emit) nor its ranks
// (rb_scan): a chain's levels are marked back to back and then scanned and emitted by ONE launch each (rb_scan_all, rb_emit_all): n + 2
// dependent launches in front of the chain's read-back instead of 3 n (round 5; the detection branch 18 -> 8).  A workgroup takes
// MARKB_WORDS consecutive words of the input bitmap (4 threads a word); spans without a set bit leave at once.  Same LDS window as
// rb_mark when the span lies in one (batch, z) plane -- cells of a span are neighbours in space and reach the same few output words.
// lw = log2 of the span (words per workgroup), 3..8, chosen per level by the host (markb_span): large sparse levels take 256-word spans
// (few workgroups, most of them not empty), small dense ones 8-word spans (enough workgroups to fill the chip -- a [3, 40, 53] x 2 level
// is 398 words).  Two phases, because the set bits of a span are unevenly spread over its words (a BEV level near the sensor is dense,
// the rest empty): the threads first push the span's active cells into an LDS list, then ALL threads share the (cell, kernel line)
// items of that list as rb_mark shares its (row, line) items -- one thread walking the 27 offsets of all 8 bits of a dense word was the
// critical path of the first version (15 us for a 69-workgroup launch).
__global__ __launch_bounds__(RB_T) void rb_mark_b(Level in, long long in_ncell, BtcGeom g, Level out, int lw) {
  __shared__ unsigned s_win[3][MARK_WIN];
  __shared__ unsigned short s_list[8192];       // active cells of the span: offset from the span's first cell
  __shared__ long long s_w0[3];
  __shared__ int s_cnt[3];
  __shared__ int s_stage, s_n, s_b0, s_z0, s_rp0;
  const int tid = threadIdx.x;
  const int tshift = 8 - lw;                    // log2 threads per word
  const int span = 1 << lw;                     // words per workgroup
  const int wl = tid >> tshift;                 // word of the span this thread looks at
  const long long w = (long long)blockIdx.x * span + wl;
  unsigned bits = w < in.nblk * RB_BLK ? in.words[w] : 0u;
  {
    const int nb = 32 >> tshift;                // bits per thread
    const unsigned m = nb >= 32 ? 0xFFFFFFFFu : ((1u << nb) - 1u);
    bits &= m << ((tid & ((1 << tshift) - 1)) * nb);
  }
  if (!__syncthreads_or(bits != 0u)) return;
  const int hw = in.shape[1] * in.shape[2];
  const long long c0 = (long long)blockIdx.x * span * 32;
  if (tid == 0) {
    int stage = 0;
    long long c1 = c0 + (long long)span * 32 - 1;
    if (c1 >= in_ncell) c1 = in_ncell - 1;
    const int b0 = (int)(c0 / in.vol), b1 = (int)(c1 / in.vol);
    const int r0 = (int)(c0 - (long long)b0 * in.vol), r1 = (int)(c1 - (long long)b1 * in.vol);
    const int z0 = r0 / hw, z1 = r1 / hw;
    s_cnt[0] = s_cnt[1] = s_cnt[2] = 0;
    s_n = 0;
    s_b0 = b0; s_z0 = z0; s_rp0 = r0 - z0 * hw;
    if (g.k[0] <= 3 && b0 == b1 && z0 == z1) {
      const int y0 = (r0 - z0 * hw) / in.shape[2], y1 = (r1 - z1 * hw) / in.shape[2];
      int ylo, yhi;
      if (g.mode == BTC_MODE_CONV) {
        const int t0 = y0 + g.p[1] - (g.k[1] - 1) * g.d[1];
        ylo = t0 <= 0 ? 0 : t0 / g.s[1];
        yhi = (y1 + g.p[1]) / g.s[1];
      } else {
        ylo = y0 * g.s[1] - g.p[1];
        yhi = y1 * g.s[1] - g.p[1] + (g.k[1] - 1) * g.d[1];
      }
      ylo = ylo < 0 ? 0 : ylo;
      yhi = yhi >= g.out_shape[1] ? g.out_shape[1] - 1 : yhi;
      stage = 1;
      for (int kz = 0; kz < 3; ++kz) {
        int oz;
        s_cnt[kz] = 0;
        s_w0[kz] = 0;
        if (kz >= g.k[0] || ylo > yhi || !fwd_axis(g, 0, z0, kz, &oz)) continue;
        const long long w_lo = lvl_cell(out, b0, oz, ylo, 0) >> 5, w_hi = lvl_cell(out, b0, oz, yhi, out.shape[2] - 1) >> 5;
        if (w_hi - w_lo + 1 > MARK_WIN) { stage = 0; break; }
        s_w0[kz] = w_lo;
        s_cnt[kz] = (int)(w_hi - w_lo + 1);
      }
    }
    s_stage = stage;
  }
  __syncthreads();
  const bool staged = s_stage != 0;
  if (staged)
    for (int pz = 0; pz < 3; ++pz)
      for (int j = tid; j < s_cnt[pz]; j += RB_T) s_win[pz][j] = 0u;
  while (bits) {   // phase 1: the span's active cells
    const int bit = __ffs(bits) - 1;
    bits &= bits - 1;
    s_list[atomicAdd(&s_n, 1)] = (unsigned short)(wl * 32 + bit);
  }
  __syncthreads();
  const int n = s_n, lines = g.k[0] * g.k[1];
  const int b0 = s_b0, z0 = s_z0, rp0 = s_rp0;
  for (int e = tid; e < n * lines; e += RB_T) {   // phase 2: (cell, kernel line) items over all threads
    const int ci = e / lines, l = e - ci * lines;
    const int kz = l / g.k[1], ky = l - kz * g.k[1];
    const int off = s_list[ci];
    int bb, z, y, x;
    if (staged) {   // one (batch, z) plane: the cell's row and column follow from its offset in the plane
      const int rp = rp0 + off;
      bb = b0; z = z0;
      y = rp / in.shape[2];
      x = rp - y * in.shape[2];
    } else {
      const long long cell = c0 + off;
      bb = (int)(cell / in.vol);
      const int rem = (int)(cell - (long long)bb * in.vol);
      z = rem / hw;
      const int r2 = rem - z * hw;
      y = r2 / in.shape[2];
      x = r2 - y * in.shape[2];
      mark_line(g, out, bb, z, y, x, kz, ky);
      continue;
    }
    int oz, oy;
    if (!fwd_axis(g, 0, z, kz, &oz) || !fwd_axis(g, 1, y, ky, &oy)) continue;
    const long long line = lvl_cell(out, bb, oz, oy, 0);
    for (int kx = 0; kx < g.k[2]; ++kx) {
      int ox;
      if (!fwd_axis(g, 2, x, kx, &ox)) continue;
      const long long oc = line + ox;
      atomicOr(&s_win[kz][(int)((oc >> 5) - s_w0[kz])], 1u << ((unsigned)oc & 31u));
    }
  }
  if (!staged) return;   // (block-uniform)
  __syncthreads();
  for (int pz = 0; pz < 3; ++pz)
    for (int j = tid; j < s_cnt[pz]; j += RB_T) {
      const unsigned wb = s_win[pz][j];
      if (wb) or_word(out.words, s_w0[pz] + j, wb);
    }
}

// Marks of SEVERAL levels from the chain's input rows in ONE launch (round 5): a run of strided CONV-mode layers, each reading the level
// the one before built (an encoder: the detection branch's conv2 .. conv_out).  Marking is monotone -- a level is a set, bits are only
// ever set -- so the levels need no barrier between them, only a rule for WHO carries a cell on to the next level: the thread whose
// atomicOr set the cell's bit (the returned word says which of its bits were new).  Every active cell of level j then has exactly one
// owner, which marks the cells IT reaches at level j + 1 (one axis of one cell: the contiguous range [ceil((c + p - K + 1) / s),
// floor((c + p) / s)] clipped to the grid, dilation 1) and descends into those it set first: the work per level is cells x box, what the
// level-by-level launches do, without their launch boundaries.  (Marking every level from every input ROW instead -- boxes composed per
// axis -- is correct too and was 158 us a launch: tens of thousands of threads test and OR the same few words of the deep levels.)
// Transposed layers are left to rb_mark_b (a decoder level is 8 x its input: the queues would not hold a tile's share).
__device__ __forceinline__ unsigned long long rb_hash64(unsigned long long k) {
  k ^= k >> 33;
  k *= 0xff51afd7ed558ccdULL;
  k ^= k >> 29;
  return k;
}

constexpr int MARKM_MAX = 6;      // levels per launch

struct MarkMulti {
  int n;
  int lines[MARKM_MAX];   // bound on the (z, y) lines of one cell's box at that level
  BtcGeom g[MARKM_MAX];
  Level out[MARKM_MAX];
  // the chain's input level is probed through a hash (its rows come in voxelizer order): filled here too when keys != NULL (rb_hash_insert's job)
  unsigned long long* keys;
  int32_t* vals;
  unsigned long long mask;
  Level l0;               // shape / vol of level 0 (cell index of a row)
};

__device__ __forceinline__ bool box_axis(const BtcGeom& g, int j, int c, int* lo, int* hi) {   // a cell of the input level -> its range at the output level
  const int a = c + g.p[j] - g.k[j] + 1, s = g.s[j];
  const int olo = a <= 0 ? 0 : (s == 1 ? a : (s == 2 ? (a + 1) >> 1 : (a + s - 1) / s));
  int ohi = s == 1 ? c + g.p[j] : (s == 2 ? (c + g.p[j]) >> 1 : (c + g.p[j]) / s);
  if (ohi >= g.out_shape[j]) ohi = g.out_shape[j] - 1;
  *lo = olo;
  *hi = ohi;
  return olo <= ohi;
}

// The walk of a workgroup's MARKM_TILE input rows is breadth first: generation j = the cells of level j the workgroup's threads own, in
// an LDS queue; all threads share a generation's (cell, line of its box) units, newly owned cells go to the next generation's queue.
// (One thread walking its row's subtree depth first -- two dependent L2 round trips per word, level after level -- took 203 us a launch.)
// A generation cannot outgrow its queue: what MARKM_TILE rows reach at a level is at most MARKM_TILE x the box of one ROW there, and
// the host ends the run where that bound passes MARKM_QCAP (stride-2 encoders keep 2 cells an axis: 512 entries).
constexpr int MARKM_TILE = 64;    // (32 and 64 rows a workgroup measure the same, 256 is 2.4x slower)
constexpr int MARKM_QCAP = 2048;

__device__ __forceinline__ unsigned long long mq_pack(int b, int z, int y, int x) {
  return ((unsigned long long)(unsigned)b << 56) | ((unsigned long long)(unsigned)z << 40) | ((unsigned long long)(unsigned)y << 20) | (unsigned long long)(unsigned)x;
}

__global__ __launch_bounds__(RB_T) void rb_mark_multi(const int4* __restrict__ idx, int n, MarkMulti M) {
  __shared__ unsigned long long s_q[2][MARKM_QCAP];
  __shared__ int s_cnt[2];
  const int tid = threadIdx.x;
  const int r0 = blockIdx.x * MARKM_TILE;
  if (tid == 0) {
    s_cnt[0] = n - r0 < MARKM_TILE ? n - r0 : MARKM_TILE;
    s_cnt[1] = 0;
  }
  if (tid < MARKM_TILE && r0 + tid < n) {
    const int4 c = idx[r0 + tid];
    s_q[0][tid] = mq_pack(c.x, c.y, c.z, c.w);
  }
  __syncthreads();
  for (int lv = 0; lv < M.n; ++lv) {
    const int cur = lv & 1, nxt = cur ^ 1;
    const int cnt = s_cnt[cur];
    if (cnt == 0) break;   // (block-uniform)
    const BtcGeom& g = M.g[lv];
    const Level& L = M.out[lv];
    const int lb = M.lines[lv];
    for (int u = tid; u < cnt * lb; u += RB_T) {
      const int it = u / lb, ln = u - it * lb;
      const unsigned long long q = s_q[cur][it];
      const int b = (int)(q >> 56), z = (int)((q >> 40) & 0xFFFF), y = (int)((q >> 20) & 0xFFFFF), x = (int)(q & 0xFFFFF);
      int zl, zh, yl, yh, xl, xh;
      if (!box_axis(g, 0, z, &zl, &zh) || !box_axis(g, 1, y, &yl, &yh) || !box_axis(g, 2, x, &xl, &xh)) continue;
      const int ny = yh - yl + 1, lz = ln / ny, ly = ln - lz * ny;
      if (lz > zh - zl) continue;
      const int oz = zl + lz, oy = yl + ly;
      const long long line = lvl_cell(L, b, oz, oy, 0);
      const long long first = line + xl, last = line + xh;
      for (long long w = first >> 5; w <= (last >> 5); ++w) {
        const int b0 = w == (first >> 5) ? (int)(first & 31) : 0, b1 = w == (last >> 5) ? (int)(last & 31) : 31;
        const unsigned bits = (b1 == 31 ? 0xFFFFFFFFu : ((1u << (b1 + 1)) - 1u)) & ~((1u << b0) - 1u);
        // (no read in front of the atomic: a device-scope access is a round trip past the XCD's L2 either way, and this one is on the
        // generation's critical path -- 47 -> 43 us a launch)
        unsigned mine = bits & ~atomicOr(&L.words[w], bits);
        if (lv + 1 >= M.n) continue;   // the last level of the run: nobody to carry it on to
        while (mine) {
          const int bit = __ffs(mine) - 1;
          mine &= mine - 1;
          const int ox = (int)((w << 5) + bit - line);
          const int pos = atomicAdd(&s_cnt[nxt], 1);
          if (pos < MARKM_QCAP) s_q[nxt][pos] = mq_pack(b, oz, oy, ox);   // (always: the host's bound)
        }
      }
    }
    __syncthreads();
    if (tid == 0) {
      s_cnt[cur] = 0;
      if (s_cnt[nxt] > MARKM_QCAP) s_cnt[nxt] = MARKM_QCAP;
    }
    __syncthreads();
  }
  if (M.keys && tid < MARKM_TILE && r0 + tid < n) {   // cell -> row of the input level (as rb_hash_insert)
    const int4 c = idx[r0 + tid];
    const unsigned long long key = (unsigned long long)lvl_cell(M.l0, c.x, c.y, c.z, c.w) + 1ull;
    unsigned long long slot = rb_hash64(key) & M.mask;
    while (true) {
      const unsigned long long prev = atomicCAS(&M.keys[slot], 0ull, key);
      if (prev == 0ull || prev == key) break;
      slot = (slot + 1) & M.mask;
    }
    M.vals[slot] = r0 + tid;
  }
}

__device__ __forceinline__ int rb_wave_incl_scan(int v) {
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int t = __shfl_up(v, o, 64);
    if (lane >= o) v += t;
  }
  return v;
}

// one thread per 32-byte block: chunk-relative block prefixes; the LAST workgroup to arrive turns the chunk sums into
// chunk prefixes and publishes the level's row count (device, and a pinned host word when given)
__device__ __forceinline__ void scan_chunk(const Level& L, int32_t* __restrict__ chunk_sums, int32_t* __restrict__ counter, int nchunks,
                                           int32_t* __restrict__ d_total, int chunk) {
  __shared__ int s_wave[RB_T / 64 + 1];
  __shared__ int s_last;
  const long long blk = (long long)chunk * RB_CHUNK + threadIdx.x;
  int cnt = 0;
  if (blk < L.nblk) {
    const uint4* p = reinterpret_cast<const uint4*>(L.words + blk * RB_BLK);
    const uint4 a = p[0], b = p[1];
    cnt = __popc(a.x) + __popc(a.y) + __popc(a.z) + __popc(a.w) + __popc(b.x) + __popc(b.y) + __popc(b.z) + __popc(b.w);
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int incl = rb_wave_incl_scan(cnt);
  if (lane == 63) s_wave[wave] = incl;
  __syncthreads();
  int base = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < RB_T / 64; ++w) {
    base += (w < wave) ? s_wave[w] : 0;
    tot += s_wave[w];
  }
  if (blk < L.nblk) L.bprefix[blk] = base + incl - cnt;
  if (threadIdx.x == 0) {
    // (device-scope store + drained queue instead of a release fence -- which writes back the XCD L2 once per workgroup, bn_fuse.h)
    btc_st_agent(&chunk_sums[chunk], tot);
    s_last = (btc_ticket_take(counter) == nchunks - 1);
    if (s_last) btc_ticket_acquire();
  }
  __syncthreads();
  if (!s_last) return;
  // exclusive scan of the chunk sums in index order (deterministic), 256 at a time
  int carry = 0;
  for (int c0 = 0; c0 < nchunks; c0 += RB_T) {
    const int c = c0 + threadIdx.x;
    const int v = c < nchunks ? btc_ld_agent(&chunk_sums[c]) : 0;
    const int inc = rb_wave_incl_scan(v);
    __syncthreads();
    if (lane == 63) s_wave[wave] = inc;
    __syncthreads();
    int b2 = 0, t2 = 0;
#pragma unroll
    for (int w = 0; w < RB_T / 64; ++w) {
      b2 += (w < wave) ? s_wave[w] : 0;
      t2 += s_wave[w];
    }
    if (c < nchunks) L.cprefix[c] = carry + b2 + inc - v;
    carry += t2;
  }
  if (threadIdx.x == 0) {
    L.cprefix[nchunks] = carry;
    *d_total = carry;
    __hip_atomic_store(counter, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

__global__ __launch_bounds__(RB_T) void rb_scan(Level L, int32_t* __restrict__ chunk_sums, int32_t* __restrict__ counter, int nchunks,
                                                int32_t* __restrict__ d_total) {
  scan_chunk(L, chunk_sums, counter, nchunks, d_total, (int)blockIdx.x);
}

// every level of a chain in one launch: workgroup b works on level j with block0[j] <= b < block0[j + 1] (scan: one chunk, its level's
// own last-arriver; emit: 256 words)
struct LevelSet {
  Level lv[BTC_CHAIN_MAX_LAYERS];
  int32_t* chunk_sums[BTC_CHAIN_MAX_LAYERS];
  int32_t* counter[BTC_CHAIN_MAX_LAYERS];
  int32_t* d_total[BTC_CHAIN_MAX_LAYERS];
  int4* out_idx[BTC_CHAIN_MAX_LAYERS];
  long long cap[BTC_CHAIN_MAX_LAYERS];
  int nchunks[BTC_CHAIN_MAX_LAYERS];
  int block0[BTC_CHAIN_MAX_LAYERS + 1];
  int n;
};

__device__ __forceinline__ int level_of_block(const LevelSet& S) {
  int j = 0;
  while (j + 1 < S.n && (int)blockIdx.x >= S.block0[j + 1]) ++j;
  return j;
}

__global__ __launch_bounds__(RB_T) void rb_scan_all(const LevelSet S) {
  const int j = level_of_block(S);
  scan_chunk(S.lv[j], S.chunk_sums[j], S.counter[j], S.nchunks[j], S.d_total[j], (int)blockIdx.x - S.block0[j]);
}

// rows of a scanned level in ascending cell order: one thread per bitmap word (its rank base = chunk prefix + block prefix +
// the words in front of it inside the 32-byte block)
__device__ __forceinline__ void emit_word(const Level& L, long long w, int4* __restrict__ out_idx, long long cap) {
  unsigned bits = L.words[w];
  if (!bits) return;
  const long long blk = w >> 3;
  const int wi = (int)(w & 7);
  const uint4* p = reinterpret_cast<const uint4*>(L.words + blk * RB_BLK);
  const uint4 a = p[0], b = p[1];
  const unsigned ws[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
  long long row = (long long)L.cprefix[blk / RB_CHUNK] + L.bprefix[blk];
#pragma unroll
  for (int j = 0; j < 7; ++j) row += (j < wi) ? __popc(ws[j]) : 0;
  const int hw = L.shape[1] * L.shape[2];
  while (bits) {
    const int bit = __ffs(bits) - 1;
    bits &= bits - 1;
    const long long cell = w * 32 + bit;
    const int bb = (int)(cell / L.vol);
    const int rem = (int)(cell - (long long)bb * L.vol);
    const int z = rem / hw;
    const int r2 = rem - z * hw;
    const int y = r2 / L.shape[2], x = r2 - y * L.shape[2];
    if (row < cap) out_idx[row] = make_int4(bb, z, y, x);
    ++row;
  }
}

__global__ __launch_bounds__(RB_T) void rb_emit(Level L, int4* __restrict__ out_idx, long long cap) {
  const long long w = (long long)blockIdx.x * RB_T + threadIdx.x;
  if (w >= L.nblk * RB_BLK) return;
  emit_word(L, w, out_idx, cap);
}

__global__ __launch_bounds__(RB_T) void rb_emit_all(const LevelSet S) {
  const int j = level_of_block(S);
  const long long w = (long long)((int)blockIdx.x - S.block0[j]) * RB_T + threadIdx.x;
  if (w >= S.lv[j].nblk * RB_BLK) return;
  emit_word(S.lv[j], w, S.out_idx[j], S.cap[j]);
}

// ---- 64-bit-key hash of an arbitrary (unsorted) input level: key = cell + 1, 0 = empty (one memset clears bitmaps and hash)

__global__ __launch_bounds__(RB_T) void rb_hash_insert(const int4* __restrict__ idx, int n, Level L, unsigned long long mask,
                                                       unsigned long long* __restrict__ keys, int32_t* __restrict__ vals) {
  const int i = blockIdx.x * RB_T + threadIdx.x;
  if (i >= n) return;
  const int4 c = idx[i];
  const unsigned long long key = (unsigned long long)lvl_cell(L, c.x, c.y, c.z, c.w) + 1ull;
  unsigned long long slot = rb_hash64(key) & mask;
  while (true) {
    const unsigned long long prev = atomicCAS(&keys[slot], 0ull, key);
    if (prev == 0ull || prev == key) break;
    slot = (slot + 1) & mask;
  }
  vals[slot] = i;
}

__device__ __forceinline__ int rb_hash_find(unsigned long long key, unsigned long long mask, const unsigned long long* __restrict__ keys,
                                            const int32_t* __restrict__ vals) {
  unsigned long long slot = rb_hash64(key) & mask;
  while (true) {
    const unsigned long long kq = keys[slot];
    if (kq == key) return vals[slot];
    if (kq == 0ull) return -1;
    slot = (slot + 1) & mask;
  }
}

// ---- neighbour maps: every job of a chain in one launch
// A job walks the rows of one level and writes one (rows, K) map, coalesced:
//   JOB_FWD   input rows  -> nbr_in [i][k] = rank, in the OUTPUT level, of the cell input i reaches at offset k
//             (+ scatter, for an input level in arbitrary order that cannot be probed: nbr_out[row][k] = i into a map pre-filled with -1)
//   JOB_BWD   output rows -> nbr_out[o][k] = rank, in the INPUT level, of the cell that reaches output o at offset k -- a gather: no
//             -1 fill and no scattered 4-byte stores (the fill + scatter wrote every line of the map twice: profiles/r02h_pmc.json)
//   JOB_SUBM  rows of a level -> nbr_out[i][k] = rank of the neighbour cell; nbr_in is its mirror image (nbr_in[i][K-1-k] ==
//             nbr_out[i][k]) and is only written on request -- the apply kernels read the forward map mirrored instead
//             (BTC_PASS_DGRAD_MIRROR), half of a submanifold rulebook's bytes
// A workgroup takes RB_ROWS consecutive rows.  Levels built here are sorted by cell, so those rows are neighbours in space: when they
// lie in one (batch, z) plane the workgroup STAGES the bitmap words of the <= 3 target planes x the y-range its rows can reach,
// each with the rank of its first bit, in LDS, and the rows x K probes become LDS reads (one global 32-byte block + two prefix
// words per staged WORD instead of per hit).  Windows that do not fit (sparse levels: 64 rows spread over many grid lines) and
// hashed levels are probed directly.
enum { JOB_FWD = 0, JOB_BWD = 1, JOB_SUBM = 2 };
constexpr int RB_ROWS = 32;          // rows per workgroup of rb_fill (x 27 offsets = 864 probes for 256 threads)
constexpr int RB_WIN = 224;          // bitmap words per staged plane window

struct Job {
  int type;
  int n;                        // rows walked
  int ranked;                   // the probed level is a ranked bitmap (else the hash of the chain's input level)
  int sorted;                   // the walked rows are in ascending cell order (a level built here): staging is possible
  long long first_block;        // first workgroup of the job
  BtcGeom g;
  Level lvl;                    // the PROBED level: FWD the output level, BWD the input level, SUBM the level itself (hash: shape only)
  const int4* idx;              // the walked rows
  int32_t* map;                 // (n, K), written coalesced: FWD nbr_in, BWD nbr_out, SUBM nbr_out
  int32_t* map2;                // FWD: nbr_out to scatter into, or NULL; SUBM: nbr_in (mirror image), or NULL
  int32_t* first;               // (n) or NULL: lowest offset k with map[i][k] >= 0, K if none -- the sort key of the row-order hints
                                // (row_order.hip), produced here so that order_local need not read the map again
  int32_t* first2;              // FWD + scatter: the same keys for the scattered map (rows of the OUTPUT level), pre-filled with a
                                // large value and lowered with atomicMin; or NULL
  const unsigned long long* keys;
  const int32_t* vals;
  unsigned long long mask;
};

struct Jobs {
  int count;
  Job j[RB_MAX_JOBS];
};
static_assert(sizeof(Jobs) <= 4096, "rb_fill's job table must fit the kernel argument segment");

// one axis of a probe: coordinate c of the walked row + kernel offset kv -> coordinate in the probed level
__device__ __forceinline__ bool probe_axis(const Job& J, int j, int c, int kv, int* o) {
  const BtcGeom& g = J.g;
  int q;
  if (J.type == JOB_SUBM) {
    q = c + (kv - g.k[j] / 2) * g.d[j];
  } else if ((J.type == JOB_FWD) == (g.mode == BTC_MODE_CONV)) {   // FWD of a conv / BWD of a transposed conv: (c + p - kv d) / s
    const int t = c + g.p[j] - kv * g.d[j];
    if (t < 0 || !div_stride(t, g.s[j], &q)) return false;
  } else {                                                         // FWD of a transposed conv / BWD of a conv: c s - p + kv d
    q = c * g.s[j] - g.p[j] + kv * g.d[j];
  }
  *o = q;
  return q >= 0 && q < J.lvl.shape[j];
}

// range of probed coordinates along axis j for walked coordinates c0 <= c1, clamped to the level (lo > hi: nothing)
__device__ __forceinline__ void probe_range(const Job& J, int j, int c0, int c1, int* lo, int* hi) {
  const BtcGeom& g = J.g;
  int a, b;
  if (J.type == JOB_SUBM) {
    a = c0 - (g.k[j] / 2) * g.d[j];
    b = c1 + (g.k[j] - 1 - g.k[j] / 2) * g.d[j];
  } else if ((J.type == JOB_FWD) == (g.mode == BTC_MODE_CONV)) {
    const int t0 = c0 + g.p[j] - (g.k[j] - 1) * g.d[j];
    a = t0 <= 0 ? 0 : t0 / g.s[j];
    b = (c1 + g.p[j]) / g.s[j];
  } else {
    a = c0 * g.s[j] - g.p[j];
    b = c1 * g.s[j] - g.p[j] + (g.k[j] - 1) * g.d[j];
  }
  *lo = a < 0 ? 0 : a;
  *hi = b >= J.lvl.shape[j] ? J.lvl.shape[j] - 1 : b;
}

__global__ __launch_bounds__(RB_T) void rb_fill(Jobs jobs) {
  __shared__ unsigned s_word[3][RB_WIN];
  __shared__ int32_t s_base[3][RB_WIN];
  __shared__ long long s_w0[3];
  __shared__ int s_cnt[3];
  __shared__ int s_stage;
  __shared__ int s_first[RB_ROWS];
  int ji = 0;
#pragma unroll
  for (int q = 1; q < RB_MAX_JOBS; ++q)
    if (q < jobs.count && (long long)blockIdx.x >= jobs.j[q].first_block) ji = q;
  const Job& J = jobs.j[ji];
  const int K = J.g.K;
  const int r0 = (int)((long long)blockIdx.x - J.first_block) * RB_ROWS;
  if (r0 >= J.n) return;
  const int rows = J.n - r0 < RB_ROWS ? J.n - r0 : RB_ROWS;
  const Level& L = J.lvl;
  const int tid = threadIdx.x;

  // ---- staging decision (thread 0): one plane, <= 3 kernel planes, every window fits
  if (tid == 0) {
    int stage = 0;
    if (J.ranked && J.sorted && J.g.k[0] <= 3 && rows >= 8) {
      const int4 a = J.idx[r0], b = J.idx[r0 + rows - 1];
      if (a.x == b.x && a.y == b.y) {
        int ylo, yhi;
        probe_range(J, 1, a.z, b.z, &ylo, &yhi);
        stage = 1;
        for (int kz = 0; kz < 3; ++kz) {
          int nz;
          s_cnt[kz] = 0;
          s_w0[kz] = 0;
          if (kz >= J.g.k[0] || ylo > yhi || !probe_axis(J, 0, a.y, kz, &nz)) continue;
          const long long w_lo = lvl_cell(L, a.x, nz, ylo, 0) >> 5, w_hi = lvl_cell(L, a.x, nz, yhi, L.shape[2] - 1) >> 5;
          if (w_hi - w_lo + 1 > RB_WIN) { stage = 0; break; }
          s_w0[kz] = w_lo;
          s_cnt[kz] = (int)(w_hi - w_lo + 1);
        }
      }
    }
    s_stage = stage;
  }
  if (tid < RB_ROWS) s_first[tid] = K;
  __syncthreads();
  const bool staged = s_stage != 0;
  if (staged) {
    for (int e = tid; e < 3 * RB_WIN; e += RB_T) {
      const int pz = e / RB_WIN, j = e - pz * RB_WIN;
      if (j >= s_cnt[pz]) continue;
      const long long w = s_w0[pz] + j;
      const long long blk = w >> 3;
      const int wi = (int)(w & 7);
      const uint4* p = reinterpret_cast<const uint4*>(L.words + blk * RB_BLK);
      const uint4 a = p[0], b = p[1];
      const unsigned ws[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
      int r = L.cprefix[blk / RB_CHUNK] + L.bprefix[blk];
      unsigned word = 0;
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        r += (q < wi) ? __popc(ws[q]) : 0;
        word = (q == wi) ? ws[q] : word;
      }
      s_word[pz][j] = word;
      s_base[pz][j] = r;
    }
    __syncthreads();
  }

  for (int e = tid; e < rows * K; e += RB_T) {
    int il, kk;
    split_item((long long)e, K, &il, &kk);
    const int i = r0 + il;
    const int4 c = J.idx[i];
    int kz, ky, kx, z, y, x;
    split_offset(J.g, kk, &kz, &ky, &kx);
    int r = -1;
    if (probe_axis(J, 0, c.y, kz, &z) && probe_axis(J, 1, c.z, ky, &y) && probe_axis(J, 2, c.w, kx, &x)) {
      const long long cell = lvl_cell(L, c.x, z, y, x);
      if (staged) {
        const int j = (int)((cell >> 5) - s_w0[kz]);
        const unsigned word = s_word[kz][j];
        const unsigned bit = (unsigned)cell & 31u;
        if ((word >> bit) & 1u) r = s_base[kz][j] + __popc(word & ((1u << bit) - 1u));
      } else if (J.ranked) {
        r = lvl_rank(L, cell);
      } else {
        r = rb_hash_find((unsigned long long)cell + 1ull, J.mask, J.keys, J.vals);
      }
    }
    J.map[(size_t)i * K + kk] = r;
    if (J.first && r >= 0) atomicMin(&s_first[il], kk);
    if (J.map2) {
      if (J.type == JOB_SUBM) J.map2[(size_t)i * K + (K - 1 - kk)] = r;   // input i feeds, at the mirrored offset, exactly the row that is its neighbour here
      else if (r >= 0) {                                                   // JOB_FWD scatter (the output level is the union of what is reachable: r >= 0 whenever the cell is in range)
        J.map2[(size_t)r * K + kk] = i;
        if (J.first2) atomicMin(&J.first2[r], kk);
      }
    }
  }
  if (J.first) {
    __syncthreads();
    if (tid < rows) J.first[r0 + tid] = s_first[tid];
  }
}

// ---- spconv-layout pair lists
__global__ __launch_bounds__(256) void pairs_count(const int32_t* __restrict__ nbr_out, int n_out, int K,
                                                   int32_t* __restrict__ flags /* K*(n_out+1), offset-major */) {
  long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (long long)K * (n_out + 1)) return;
  int kk = (int)(t / (n_out + 1)), i = (int)(t % (n_out + 1));
  flags[t] = (i < n_out) ? (nbr_out[(size_t)i * K + kk] >= 0) : 0;
}

__global__ __launch_bounds__(256) void pairs_write(const int32_t* __restrict__ nbr_out, const int32_t* __restrict__ excl,
                                                   int n_out, int K, int n_in, int32_t* __restrict__ pairs,
                                                   int32_t* __restrict__ pair_num) {
  long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (long long)K * (n_out + 1)) return;
  int kk = (int)(t / (n_out + 1)), i = (int)(t % (n_out + 1));
  int base = excl[(size_t)kk * (n_out + 1)];
  int pos = excl[t] - base;
  if (i == n_out) {
    pair_num[kk] = pos;
    return;
  }
  int j = nbr_out[(size_t)i * K + kk];
  if (j >= 0) {
    pairs[(size_t)kk * n_in + pos] = j;                       // pairs[0][k][pos] = input row
    pairs[(size_t)K * n_in + (size_t)kk * n_in + pos] = i;    // pairs[1][k][pos] = output row
  }
}

int fill_geom(BtcGeom* g, const int32_t* in_shape, const int32_t* out_shape, const int32_t* k, const int32_t* s,
              const int32_t* p, const int32_t* d, int mode) {
  for (int j = 0; j < 3; ++j) {
    g->in_shape[j] = in_shape[j];
    g->out_shape[j] = out_shape ? out_shape[j] : in_shape[j];
    g->k[j] = k[j];
    g->s[j] = s ? s[j] : 1;
    g->p[j] = p ? p[j] : 0;
    g->d[j] = d ? d[j] : 1;
    if (g->k[j] < 1 || g->s[j] < 1 || g->d[j] < 1 || g->in_shape[j] < 1 || g->out_shape[j] < 1) {
      btc_set_error("rulebook: bad geometry on axis %d", j);
      return BTC_EINVAL;
    }
  }
  g->K = k[0] * k[1] * k[2];
  g->mode = mode;
  return BTC_OK;
}

// ---- host-side layout of a level inside a workspace
struct LevelLayout {
  long long ncell, nw, nblk;
  int nchunks;
  size_t words_bytes, bprefix_bytes, chunk_bytes;  // chunk_bytes covers chunk_sums (nchunks) and cprefix (nchunks + 1)
};

int level_layout(int batch, const int32_t* shape, LevelLayout* o) {
  const long long vol = (long long)shape[0] * shape[1] * shape[2];
  if (vol >= 0x7fffffffLL) {
    btc_set_error("rulebook: a single scene's grid of %lld cells exceeds the 31-bit per-scene cell index", vol);
    return BTC_ERANGE;
  }
  o->ncell = vol * batch;
  o->nw = (o->ncell + 31) / 32;
  o->nblk = (o->nw + RB_BLK - 1) / RB_BLK;
  const long long nch = (o->nblk + RB_CHUNK - 1) / RB_CHUNK;
  if (nch >= 0x7fffffffLL || o->ncell / 8 > (1ll << 40)) {
    btc_set_error("rulebook: batch * grid of %lld cells is too large", o->ncell);
    return BTC_ERANGE;
  }
  o->nchunks = (int)nch;
  o->words_bytes = btc_align((size_t)o->nblk * RB_BLK * sizeof(unsigned));
  o->bprefix_bytes = btc_align((size_t)o->nblk * sizeof(int32_t));
  o->chunk_bytes = btc_align((size_t)(2 * o->nchunks + 1) * sizeof(int32_t));
  return BTC_OK;
}

Level make_level(const LevelLayout& lo, const int32_t* shape, unsigned* words, int32_t* bprefix, int32_t* chunk) {
  Level L;
  L.words = words;
  L.bprefix = bprefix;
  L.cprefix = chunk + lo.nchunks;   // chunk_sums first, then cprefix
  L.shape[0] = shape[0]; L.shape[1] = shape[1]; L.shape[2] = shape[2];
  L.vol = shape[0] * shape[1] * shape[2];
  L.nblk = lo.nblk;
  return L;
}

int markb_span(long long nw) {   // log2 words per workgroup of rb_mark_b: about 2 K workgroups, 8..256 words each
  int lw = 3;
  while (lw < 8 && (nw >> lw) > 2048) ++lw;
  return lw;
}

int mark_grid(long long n_rows) {   // one workgroup per MARK_ROWS rows of the host-side bound, at most 4096 (they stride over the real tiles)
  long long g = (n_rows + MARK_ROWS - 1) / MARK_ROWS;
  return (int)(g < 1 ? 1 : (g > 4096 ? 4096 : g));
}

}  // namespace

extern "C" int btc_out_shape(const int32_t* in_shape, const int32_t* k, const int32_t* s, const int32_t* p, const int32_t* d,
                             const int32_t* outpad, int mode, int32_t* out_shape) {
  for (int j = 0; j < 3; ++j) {
    if (mode == BTC_MODE_SUBM) out_shape[j] = in_shape[j];
    else if (mode == BTC_MODE_CONV) out_shape[j] = (in_shape[j] + 2 * p[j] - d[j] * (k[j] - 1) - 1) / s[j] + 1;
    else if (mode == BTC_MODE_TRANSPOSE) out_shape[j] = (in_shape[j] - 1) * s[j] - 2 * p[j] + k[j] + (outpad ? outpad[j] : 0);
    else { btc_set_error("btc_out_shape: bad mode %d", mode); return BTC_EINVAL; }
  }
  return BTC_OK;
}

// ------------------------------------------------------------------------------------------------ single SubM rulebook
// (arbitrary input order: hash)
static unsigned long long subm_hash_cap(int n) { return btc_pow2_ge((unsigned long long)(n > 0 ? n : 1) * 2); }

extern "C" size_t btc_rulebook_subm_ws_bytes(int n) {
  const unsigned long long cap = subm_hash_cap(n);
  return btc_align((size_t)cap * sizeof(unsigned long long)) + btc_align((size_t)cap * sizeof(int32_t));
}

extern "C" int btc_rulebook_subm(const int32_t* indices, int n, int batch, const int32_t* h_shape, const int32_t* h_k,
                                 const int32_t* h_d, int32_t* nbr_out, int32_t* nbr_in, void* ws, size_t ws_bytes,
                                 void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  BtcGeom g;
  int rc = fill_geom(&g, h_shape, nullptr, h_k, nullptr, nullptr, h_d, BTC_MODE_SUBM);
  if (rc) return rc;
  BTC_CHECK_ARG((h_k[0] & 1) && (h_k[1] & 1) && (h_k[2] & 1), "btc_rulebook_subm: kernel sizes must be odd");
  BTC_CHECK_ARG(ws_bytes >= btc_rulebook_subm_ws_bytes(n), "btc_rulebook_subm: workspace too small");
  LevelLayout lo;
  rc = level_layout(batch, h_shape, &lo);
  if (rc) return rc;
  if (n <= 0) return BTC_OK;
  const unsigned long long cap = subm_hash_cap(n);
  BtcCarver cv(ws);
  unsigned long long* keys = cv.take<unsigned long long>(cap);
  int32_t* vals = cv.take<int32_t>(cap);
  BTC_HIP(hipMemsetAsync(keys, 0, (size_t)cap * sizeof(unsigned long long), stream));
  Level L = make_level(lo, h_shape, nullptr, nullptr, nullptr);
  rb_hash_insert<<<btc_cdiv(n, RB_T), RB_T, 0, stream>>>((const int4*)indices, n, L, cap - 1, keys, vals);
  BTC_LAUNCH_CHECK();
  Jobs jobs;
  jobs.count = 1;
  Job& J = jobs.j[0];
  J.type = JOB_SUBM; J.n = n; J.ranked = 0; J.sorted = 0; J.first_block = 0; J.g = g; J.lvl = L; J.idx = (const int4*)indices;
  J.map = nbr_out; J.map2 = nbr_in; J.first = nullptr; J.first2 = nullptr; J.keys = keys; J.vals = vals; J.mask = cap - 1;
  rb_fill<<<btc_cdiv(n, RB_ROWS), RB_T, 0, stream>>>(jobs);
  BTC_LAUNCH_CHECK();
  return BTC_OK;
}

// ------------------------------------------------------------------------------------------------ single strided rulebook
extern "C" size_t btc_rulebook_conv_ws_bytes(int batch, const int32_t* h_out_shape) {
  LevelLayout lo;
  if (level_layout(batch, h_out_shape, &lo) != BTC_OK) return 0;
  return lo.words_bytes + lo.bprefix_bytes + lo.chunk_bytes + 256;
}

static int conv_ws_carve(int batch, const int32_t* h_out_shape, void* ws, size_t ws_bytes, LevelLayout* lo, Level* L, int32_t** chunk_sums,
                         int32_t** counter) {
  int rc = level_layout(batch, h_out_shape, lo);
  if (rc) return rc;
  BTC_CHECK_ARG(ws_bytes >= lo->words_bytes + lo->bprefix_bytes + lo->chunk_bytes + 256, "btc_rulebook_conv: workspace too small");
  char* base = (char*)ws;
  unsigned* words = (unsigned*)base;
  *counter = (int32_t*)(base + lo->words_bytes);                     // cleared together with the words
  int32_t* bprefix = (int32_t*)(base + lo->words_bytes + 256);
  int32_t* chunk = (int32_t*)(base + lo->words_bytes + 256 + lo->bprefix_bytes);
  *chunk_sums = chunk;
  *L = make_level(*lo, h_out_shape, words, bprefix, chunk);
  return BTC_OK;
}

extern "C" int btc_rulebook_conv_count(const int32_t* indices, int n, int batch, const int32_t* h_in_shape,
                                       const int32_t* h_out_shape, const int32_t* h_k, const int32_t* h_s,
                                       const int32_t* h_p, const int32_t* h_d, int mode, int32_t* d_n_out, void* ws,
                                       size_t ws_bytes, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  BTC_CHECK_ARG(mode == BTC_MODE_CONV || mode == BTC_MODE_TRANSPOSE, "btc_rulebook_conv_count: bad mode");
  BtcGeom g;
  int rc = fill_geom(&g, h_in_shape, h_out_shape, h_k, h_s, h_p, h_d, mode);
  if (rc) return rc;
  LevelLayout lo;
  Level L;
  int32_t *chunk_sums, *counter;
  rc = conv_ws_carve(batch, h_out_shape, ws, ws_bytes, &lo, &L, &chunk_sums, &counter);
  if (rc) return rc;
  BTC_HIP(hipMemsetAsync(L.words, 0, lo.words_bytes + 256, stream));
  if (n > 0) {
    rb_mark<<<mark_grid(n), RB_T, 0, stream>>>((const int4*)indices, n, nullptr, g, L, 0);
    BTC_LAUNCH_CHECK();
  }
  rb_scan<<<lo.nchunks, RB_T, 0, stream>>>(L, chunk_sums, counter, lo.nchunks, d_n_out);
  BTC_LAUNCH_CHECK();
  return BTC_OK;
}

extern "C" int btc_rulebook_conv_fill(const int32_t* indices, int n, int batch, const int32_t* h_in_shape,
                                      const int32_t* h_out_shape, const int32_t* h_k, const int32_t* h_s,
                                      const int32_t* h_p, const int32_t* h_d, int mode, int n_out, int32_t* out_indices,
                                      int32_t* nbr_out, int32_t* nbr_in, void* ws, size_t ws_bytes, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  BTC_CHECK_ARG(mode == BTC_MODE_CONV || mode == BTC_MODE_TRANSPOSE, "btc_rulebook_conv_fill: bad mode");
  BtcGeom g;
  int rc = fill_geom(&g, h_in_shape, h_out_shape, h_k, h_s, h_p, h_d, mode);
  if (rc) return rc;
  LevelLayout lo;
  Level L;
  int32_t *chunk_sums, *counter;
  rc = conv_ws_carve(batch, h_out_shape, ws, ws_bytes, &lo, &L, &chunk_sums, &counter);
  if (rc) return rc;
  if (n_out > 0) {
    BTC_HIP(hipMemsetAsync(nbr_out, 0xFF, (size_t)n_out * g.K * sizeof(int32_t), stream));
    rb_emit<<<btc_cdiv(lo.nblk * RB_BLK, RB_T), RB_T, 0, stream>>>(L, (int4*)out_indices, (long long)n_out);
    BTC_LAUNCH_CHECK();
  }
  if (n > 0) {
    Jobs jobs;
    jobs.count = 1;
    Job& J = jobs.j[0];
    J.type = JOB_FWD; J.n = n; J.ranked = 1; J.sorted = 0; J.first_block = 0; J.g = g; J.lvl = L; J.idx = (const int4*)indices;
    J.map = nbr_in; J.map2 = nbr_out; J.first = nullptr; J.first2 = nullptr; J.keys = nullptr; J.vals = nullptr; J.mask = 0;
    rb_fill<<<btc_cdiv(n, RB_ROWS), RB_T, 0, stream>>>(jobs);
    BTC_LAUNCH_CHECK();
  }
  return BTC_OK;
}

// ------------------------------------------------------------------------------------------------ a chain of layers
// Levels: 0 = the chain's input rows (arbitrary order, hash); every kind-1 layer creates a new level.
namespace {

struct ChainPlan {
  int n_layers;
  int lvl_in[BTC_CHAIN_MAX_LAYERS], lvl_out[BTC_CHAIN_MAX_LAYERS];   // level ids per layer
  int n_levels;                                                      // including level 0
  int producer[BTC_CHAIN_MAX_LAYERS + 1];                            // level -> layer that builds it (-1 for level 0)
  bool need_hash;
};

int chain_plan(const BtcChainLayer* layers, int n_layers, ChainPlan* P) {
  BTC_CHECK_ARG(n_layers >= 1 && n_layers <= BTC_CHAIN_MAX_LAYERS, "chain: 1..%d layers", BTC_CHAIN_MAX_LAYERS);
  P->n_layers = n_layers;
  P->n_levels = 1;
  P->producer[0] = -1;
  P->need_hash = false;
  int cur = 0;
  for (int i = 0; i < n_layers; ++i) {
    const BtcChainLayer& l = layers[i];
    if (l.kind == 0) {
      BTC_CHECK_ARG((l.k[0] & 1) && (l.k[1] & 1) && (l.k[2] & 1), "chain: submanifold kernel sizes must be odd");
      P->lvl_in[i] = P->lvl_out[i] = cur;
      if (cur == 0) P->need_hash = true;
    } else if (l.kind == 1) {
      BTC_CHECK_ARG(l.mode == BTC_MODE_CONV || l.mode == BTC_MODE_TRANSPOSE, "chain: layer %d: bad mode", i);
      P->lvl_in[i] = cur;
      cur = P->n_levels++;
      P->producer[cur] = i;
      P->lvl_out[i] = cur;
    } else if (l.kind == 2 || l.kind == 3) {
      BTC_CHECK_ARG(l.ref >= 0 && l.ref < i, "chain: layer %d: bad reference", i);
      const int r = l.ref;
      P->lvl_in[i] = l.kind == 2 ? P->lvl_out[r] : P->lvl_in[r];
      P->lvl_out[i] = l.kind == 2 ? P->lvl_in[r] : P->lvl_out[r];
      cur = P->lvl_out[i];
    } else {
      btc_set_error("chain: layer %d: bad kind %d", i, l.kind);
      return BTC_EINVAL;
    }
  }
  return BTC_OK;
}

struct ChainWs {
  LevelLayout lo[BTC_CHAIN_MAX_LAYERS + 1];
  Level lv[BTC_CHAIN_MAX_LAYERS + 1];
  int32_t* chunk_sums[BTC_CHAIN_MAX_LAYERS + 1];
  int32_t* counters;            // one per level
  unsigned long long* keys;
  int32_t* vals;
  unsigned long long hash_cap;
  size_t zero_bytes;            // leading region cleared by one memset: all bitmaps, the counters, the hash keys
  size_t total_bytes;
};

int chain_ws(const BtcChainLayer* layers, const ChainPlan& P, int batch, int n0, void* ws, ChainWs* W) {
  size_t off = 0;
  char* base = (char*)ws;
  for (int lv = 1; lv < P.n_levels; ++lv) {   // bitmaps first (zeroed region)
    const BtcChainLayer& l = layers[P.producer[lv]];
    int rc = level_layout(batch, l.out_shape, &W->lo[lv]);
    if (rc) return rc;
    W->lv[lv].words = (unsigned*)(base + off);
    off += W->lo[lv].words_bytes;
  }
  W->counters = (int32_t*)(base + off);
  off += btc_align((size_t)(BTC_CHAIN_MAX_LAYERS + 1) * sizeof(int32_t));
  W->hash_cap = P.need_hash ? subm_hash_cap(n0) : 0;
  W->keys = (unsigned long long*)(base + off);
  off += btc_align((size_t)W->hash_cap * sizeof(unsigned long long));
  W->zero_bytes = off;
  W->vals = (int32_t*)(base + off);
  off += btc_align((size_t)W->hash_cap * sizeof(int32_t));
  for (int lv = 1; lv < P.n_levels; ++lv) {
    const BtcChainLayer& l = layers[P.producer[lv]];
    unsigned* words = W->lv[lv].words;
    int32_t* bprefix = (int32_t*)(base + off);
    off += W->lo[lv].bprefix_bytes;
    int32_t* chunk = (int32_t*)(base + off);
    off += W->lo[lv].chunk_bytes;
    W->chunk_sums[lv] = chunk;
    W->lv[lv] = make_level(W->lo[lv], l.out_shape, words, bprefix, chunk);
  }
  W->total_bytes = off;
  return BTC_OK;
}

BtcGeom geom_of(const BtcChainLayer& l) {
  BtcGeom g;
  for (int j = 0; j < 3; ++j) {
    g.in_shape[j] = l.in_shape[j]; g.out_shape[j] = l.out_shape[j]; g.k[j] = l.k[j];
    g.s[j] = l.s[j]; g.p[j] = l.p[j]; g.d[j] = l.d[j];
  }
  g.K = l.k[0] * l.k[1] * l.k[2];
  g.mode = l.mode;
  return g;
}

}  // namespace

extern "C" size_t btc_chain_ws_bytes(const BtcChainLayer* layers, int n_layers, int batch, int n0) {
  ChainPlan P;
  if (chain_plan(layers, n_layers, &P) != BTC_OK) return 0;
  ChainWs W;
  if (chain_ws(layers, P, batch, n0, nullptr, &W) != BTC_OK) return 0;
  return W.total_bytes + 256;
}

extern "C" int btc_chain_caps(const BtcChainLayer* layers, int n_layers, int batch, int n0, int64_t* h_cap) {
  ChainPlan P;
  int rc = chain_plan(layers, n_layers, &P);
  if (rc) return rc;
  long long cap_of[BTC_CHAIN_MAX_LAYERS + 1];
  cap_of[0] = n0;
  for (int i = 0; i < n_layers; ++i) {
    h_cap[i] = 0;
    if (layers[i].kind != 1) continue;
    const long long K = (long long)layers[i].k[0] * layers[i].k[1] * layers[i].k[2];
    const long long cells = (long long)batch * layers[i].out_shape[0] * layers[i].out_shape[1] * layers[i].out_shape[2];
    long long c = cap_of[P.lvl_in[i]] * K;
    if (c > cells) c = cells;
    if (c < 1) c = 1;
    cap_of[P.lvl_out[i]] = c;
    h_cap[i] = c;
  }
  return BTC_OK;
}

extern "C" int btc_chain_levels(const int32_t* indices, int n0, int batch, const BtcChainLayer* layers, int n_layers,
                                int32_t* const* out_indices, const int64_t* h_cap, int32_t* d_counts, void* ws, size_t ws_bytes, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  ChainPlan P;
  int rc = chain_plan(layers, n_layers, &P);
  if (rc) return rc;
  ChainWs W;
  rc = chain_ws(layers, P, batch, n0, ws, &W);
  if (rc) return rc;
  BTC_CHECK_ARG(ws_bytes >= W.total_bytes, "btc_chain_levels: workspace too small");
  BTC_HIP(hipMemsetAsync(ws, 0, W.zero_bytes, stream));
  // the run of CONV-mode strided layers that starts at level 0 is marked by ONE launch from the input rows (rb_mark_multi), which fills
  // the input level's hash as well; `composed[i]` = layer i's level is covered by it
  bool composed[BTC_CHAIN_MAX_LAYERS] = {false};
  MarkMulti M;
  M.n = 0;
  M.keys = nullptr;
  M.vals = nullptr;
  M.mask = 0;
  if (n0 > 0 && btc_tune_get(BTC_TUNE_RB_MARK_MULTI) != 1) {
    int at = 0;   // the level the run has reached
    int size[3] = {1, 1, 1};   // bound on what ONE input row reaches at that level, per axis (a range of `size` cells maps onto
                               // at most floor((size - 1 + K - 1) / s) + 1 cells)
    int members[MARKM_MAX];
    for (int i = 0; i < n_layers && M.n < MARKM_MAX; ++i) {
      const BtcChainLayer& l = layers[i];
      if (l.kind != 1) continue;
      if (P.lvl_in[i] != at) continue;          // (a layer that reads an earlier level: not part of the run, but does not end it either)
      if (l.mode != BTC_MODE_CONV || l.d[0] != 1 || l.d[1] != 1 || l.d[2] != 1 || l.s[0] < 1 || l.s[1] < 1 || l.s[2] < 1) break;
      // a queue entry packs (batch, z, y, x) into 8 + 16 + 20 + 20 bits: grids or batches beyond that keep the launch-per-level marks
      if (batch > 255 || l.in_shape[0] >= (1 << 16) || l.in_shape[1] >= (1 << 20) || l.in_shape[2] >= (1 << 20) ||
          l.out_shape[0] >= (1 << 16) || l.out_shape[1] >= (1 << 20) || l.out_shape[2] >= (1 << 20)) break;
      int nsz[3];
      for (int j = 0; j < 3; ++j) nsz[j] = (size[j] - 1 + l.k[j] - 1) / l.s[j] + 1;
      if ((long long)MARKM_TILE * nsz[0] * nsz[1] * nsz[2] > MARKM_QCAP) break;
      for (int j = 0; j < 3; ++j) size[j] = nsz[j];
      M.g[M.n] = geom_of(l);
      M.out[M.n] = W.lv[P.lvl_out[i]];
      M.lines[M.n] = ((l.k[0] - 1) / l.s[0] + 1) * ((l.k[1] - 1) / l.s[1] + 1);
      members[M.n++] = i;
      at = P.lvl_out[i];
    }
    if (M.n >= 2) {   // (a single level: rb_mark's per-tile path is the same work)
      for (int q = 0; q < M.n; ++q) composed[members[q]] = true;
    } else {
      M.n = 0;
    }
  }
  if (P.need_hash && n0 > 0) {
    // level 0's grid is the input shape of the first layer that runs on it
    int first = -1;
    for (int i = 0; i < n_layers && first < 0; ++i)
      if (P.lvl_in[i] == 0 && layers[i].kind <= 1) first = i;
    LevelLayout l0;
    rc = level_layout(batch, layers[first].in_shape, &l0);
    if (rc) return rc;
    Level L0 = make_level(l0, layers[first].in_shape, nullptr, nullptr, nullptr);
    if (M.n > 0) {
      M.keys = W.keys;
      M.vals = W.vals;
      M.mask = W.hash_cap - 1;
      M.l0 = L0;
    } else {
      rb_hash_insert<<<btc_cdiv(n0, RB_T), RB_T, 0, stream>>>((const int4*)indices, n0, L0, W.hash_cap - 1, W.keys, W.vals);
      BTC_LAUNCH_CHECK();
    }
  }
  if (M.n > 0) {
    rb_mark_multi<<<btc_cdiv(n0, MARKM_TILE), RB_T, 0, stream>>>((const int4*)indices, n0, M);
    BTC_LAUNCH_CHECK();
  }
  LevelSet S;
  S.n = 0;
  long long scan_blocks = 0;
  for (int i = 0; i < n_layers; ++i) {
    if (layers[i].kind != 1) continue;
    const int li = P.lvl_in[i], lo = P.lvl_out[i];
    const BtcGeom g = geom_of(layers[i]);
    if (composed[i]) {
      // marked above
    } else if (li == 0) {
      if (n0 > 0) {
        rb_mark<<<mark_grid(n0), RB_T, 0, stream>>>((const int4*)indices, n0, nullptr, g, W.lv[lo], 0);
        BTC_LAUNCH_CHECK();
      }
    } else {
      const long long nw = W.lo[li].nblk * RB_BLK;
      const int lw = markb_span(nw);
      rb_mark_b<<<btc_cdiv(nw, 1LL << lw), RB_T, 0, stream>>>(W.lv[li], W.lo[li].ncell, g, W.lv[lo], lw);
      BTC_LAUNCH_CHECK();
    }
    BTC_CHECK_ARG(S.n < BTC_CHAIN_MAX_LAYERS, "btc_chain_levels: too many levels");
    S.lv[S.n] = W.lv[lo];
    S.chunk_sums[S.n] = W.chunk_sums[lo];
    S.counter[S.n] = W.counters + lo;
    S.d_total[S.n] = d_counts + i;
    S.out_idx[S.n] = (int4*)out_indices[i];
    S.cap[S.n] = (long long)h_cap[i];
    S.nchunks[S.n] = W.lo[lo].nchunks;
    S.block0[S.n] = (int)scan_blocks;
    scan_blocks += W.lo[lo].nchunks;
    ++S.n;
  }
  if (S.n == 0) return BTC_OK;
  S.block0[S.n] = (int)scan_blocks;
  BTC_CHECK_ARG(scan_blocks < (1LL << 31), "btc_chain_levels: grids too large");
  rb_scan_all<<<(unsigned)scan_blocks, RB_T, 0, stream>>>(S);
  BTC_LAUNCH_CHECK();
  long long emit_blocks = 0;
  for (int q = 0; q < S.n; ++q) {
    S.block0[q] = (int)emit_blocks;
    emit_blocks += btc_cdiv(S.lv[q].nblk * RB_BLK, RB_T);
  }
  S.block0[S.n] = (int)emit_blocks;
  BTC_CHECK_ARG(emit_blocks < (1LL << 31), "btc_chain_levels: grids too large");
  rb_emit_all<<<(unsigned)emit_blocks, RB_T, 0, stream>>>(S);
  BTC_LAUNCH_CHECK();
  return BTC_OK;
}

extern "C" int btc_chain_maps(const int32_t* indices, int n0, int batch, const BtcChainLayer* layers, int n_layers, const int32_t* h_counts,
                              int32_t* const* out_indices, int32_t* const* nbr_out, int32_t* const* nbr_in, int32_t* const* first_out,
                              int32_t* const* first_in, void* ws, size_t ws_bytes, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  ChainPlan P;
  int rc = chain_plan(layers, n_layers, &P);
  if (rc) return rc;
  ChainWs W;
  rc = chain_ws(layers, P, batch, n0, ws, &W);
  if (rc) return rc;
  BTC_CHECK_ARG(ws_bytes >= W.total_bytes, "btc_chain_maps: workspace too small");
  int rows_of[BTC_CHAIN_MAX_LAYERS + 1];
  const int32_t* idx_of[BTC_CHAIN_MAX_LAYERS + 1];
  rows_of[0] = n0;
  idx_of[0] = indices;
  for (int lv = 1; lv < P.n_levels; ++lv) {
    rows_of[lv] = h_counts[P.producer[lv]];
    idx_of[lv] = out_indices[P.producer[lv]];
  }
  // nbr_out of a strided layer whose input is the chain's arbitrary-order level 0 is scattered into (it cannot be gathered: that
  // level has no ranks) and starts as -1; adjacent buffers are cleared by one memset.  Every other strided nbr_out is gathered.
  {
    char* run_begin = nullptr;
    size_t run_bytes = 0;
    for (int i = 0; i <= n_layers; ++i) {
      char* p = nullptr;
      size_t bytes = 0;
      if (i < n_layers && layers[i].kind == 1 && P.lvl_in[i] == 0 && h_counts[i] > 0) {
        p = (char*)nbr_out[i];
        bytes = (size_t)h_counts[i] * layers[i].k[0] * layers[i].k[1] * layers[i].k[2] * sizeof(int32_t);
      } else if (i < n_layers) {
        continue;
      }
      if (p && run_begin && p >= run_begin + run_bytes && (size_t)(p - (run_begin + run_bytes)) < 256) {
        run_bytes = (size_t)(p - run_begin) + bytes;   // contiguous up to alignment padding
      } else {
        if (run_begin) BTC_HIP(hipMemsetAsync(run_begin, 0xFF, run_bytes, stream));
        run_begin = p;
        run_bytes = bytes;
      }
    }
  }
  Jobs jobs;
  jobs.count = 0;
  long long blocks = 0;
  auto flush = [&]() -> int {
    if (jobs.count == 0) return BTC_OK;
    rb_fill<<<(unsigned)blocks, RB_T, 0, stream>>>(jobs);
    BTC_LAUNCH_CHECK();
    jobs.count = 0;
    blocks = 0;
    return BTC_OK;
  };
  auto add = [&](int type, int n, const BtcGeom& g, const Level& lvl, int ranked, int sorted, const int32_t* idx, int32_t* map, int32_t* map2,
                 int32_t* first, int32_t* first2 = nullptr) -> int {
    if (n <= 0) return BTC_OK;
    const long long nb = btc_cdiv(n, RB_ROWS);
    if (jobs.count == RB_MAX_JOBS || blocks + nb > 0x7fffffffLL) {
      int rc2 = flush();
      if (rc2) return rc2;
    }
    Job& J = jobs.j[jobs.count++];
    J.type = type; J.n = n; J.ranked = ranked; J.sorted = sorted; J.first_block = blocks; J.g = g; J.lvl = lvl; J.idx = (const int4*)idx;
    J.map = map; J.map2 = map2; J.first = first; J.first2 = first2; J.keys = W.keys; J.vals = W.vals; J.mask = W.hash_cap ? W.hash_cap - 1 : 0;
    blocks += nb;
    return BTC_OK;
  };
  for (int i = 0; i < n_layers; ++i) {
    if (layers[i].kind > 1) continue;
    if (layers[i].kind == 0 && !nbr_out[i] && !nbr_in[i]) continue;   // the caller has this submanifold layer's maps already
    BTC_CHECK_ARG(nbr_out[i] && (layers[i].kind == 0 || nbr_in[i]), "btc_chain_maps: layer %d: nbr_out (and, for a strided layer, nbr_in) required", i);
    const int li = P.lvl_in[i];
    const BtcGeom g = geom_of(layers[i]);
    if (layers[i].kind == 1) {
      const int lo = P.lvl_out[i];
      // nbr_in: the input rows probe the output level; level-0 inputs also scatter nbr_out (see above)
      int32_t* keys2 = (li == 0 && first_out) ? first_out[i] : nullptr;
      if (keys2 && h_counts[i] > 0)   // scattered keys start large (0x7f7f7f7f) and are lowered with atomicMin; readers clamp to K
        BTC_HIP(hipMemsetAsync(keys2, 0x7F, (size_t)h_counts[i] * sizeof(int32_t), stream));
      rc = add(JOB_FWD, rows_of[li], g, W.lv[lo], 1, li != 0, idx_of[li], nbr_in[i], li == 0 ? nbr_out[i] : nullptr, first_in ? first_in[i] : nullptr, keys2);
      if (rc) return rc;
      if (li != 0) {   // nbr_out: the output rows probe the (ranked) input level
        rc = add(JOB_BWD, rows_of[lo], g, W.lv[li], 1, 1, idx_of[lo], nbr_out[i], nullptr, first_out ? first_out[i] : nullptr);
        if (rc) return rc;
      }
    } else if (li == 0) {
      LevelLayout l0;
      rc = level_layout(batch, layers[i].in_shape, &l0);
      if (rc) return rc;
      rc = add(JOB_SUBM, rows_of[li], g, make_level(l0, layers[i].in_shape, nullptr, nullptr, nullptr), 0, 0, idx_of[li], nbr_out[i], nbr_in[i], nullptr);
      if (rc) return rc;
    } else {
      rc = add(JOB_SUBM, rows_of[li], g, W.lv[li], 1, 1, idx_of[li], nbr_out[i], nbr_in[i], nullptr);
      if (rc) return rc;
    }
  }
  return flush();
}

// ------------------------------------------------------------------------------------------------ spconv-layout view
extern "C" size_t btc_pairs_from_nbr_ws_bytes(int n_out, int K) {
  const long long cnt = (long long)K * (n_out + 1);
  return btc_align((size_t)cnt * sizeof(int32_t)) + btc_scan_ws_bytes(cnt);
}

extern "C" int btc_pairs_from_nbr(const int32_t* nbr_out, int n_out, int K, int n_in, int32_t* pairs, int32_t* pair_num, void* ws,
                                  size_t ws_bytes, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  BTC_CHECK_ARG(n_out >= 0 && K >= 1 && n_in >= 0, "btc_pairs_from_nbr: bad sizes");
  BTC_CHECK_ARG(ws_bytes >= btc_pairs_from_nbr_ws_bytes(n_out, K), "btc_pairs_from_nbr: workspace too small");
  if (n_in > 0) BTC_HIP(hipMemsetAsync(pairs, 0xFF, (size_t)2 * K * n_in * sizeof(int32_t), stream));
  const long long cnt = (long long)K * (n_out + 1);
  BtcCarver cv(ws);
  int32_t* flags = cv.take<int32_t>(cnt);
  void* scan_ws = cv.base + cv.off;
  pairs_count<<<btc_cdiv(cnt, 256), 256, 0, stream>>>(nbr_out, n_out, K, flags);
  BTC_LAUNCH_CHECK();
  int rc = btc_scan_exclusive_i32(flags, flags, cnt, nullptr, scan_ws, stream);
  if (rc) return rc;
  pairs_write<<<btc_cdiv(cnt, 256), 256, 0, stream>>>(nbr_out, flags, n_out, K, n_in, pairs, pair_num);
  BTC_LAUNCH_CHECK();
  return BTC_OK;
}
