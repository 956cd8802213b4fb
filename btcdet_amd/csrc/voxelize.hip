// Batched first-come voxelizer for gfx950.
//
// Reference semantics: spconv.utils.VoxelGeneratorV2.generate (sequential C++ loop, SURVEY.md
// App. B.1), called at /root/reference/btcdet/datasets/processor/data_processor.py:85,136,177.
// The sequential "first come" order is reproduced in parallel:
//   1. every point CAS-inserts its cell key into an open-addressing hash table (L2 resident) and
//      pushes its row index through a cascade of atomicMin's so that each cell ends up holding its
//      `max_points` SMALLEST point indices in ascending order  (= the first points in input order);
//   2. a point is "first of its voxel" iff it is list[0] of its cell; an ordered prefix sum of those
//      flags over the points gives the voxel's first-appearance rank; ranks >= max_voxels (per
//      scene) are dropped, exactly the voxels the sequential loop would refuse;
//   3. one thread per (voxel, slot) gathers the kept points -> fully coalesced writes, zero padding.
// HBM traffic: points are read twice (insert, gather), the table stays in L2.
#include "btc_common.h"

namespace {

#define VOX_EMPTY_KEY 0x7F7F7F7F7F7F7F7FLL   // an empty hash slot: the byte pattern of the point lists' sentinel (one fill for both)


struct VoxParams {
  float lo[3];
  float vs[3];
  int grid[3];  // x,y,z
  int vol;
  int ld, xyz_col, feat_col, C;
  int n, batch, max_points, max_voxels;
  unsigned mask;
};

__device__ __forceinline__ int scene_of(const int32_t* __restrict__ off, int batch, int i) {
  int b = 0;
  while (b + 1 < batch && i >= off[b + 1]) ++b;
  return b;
}

__global__ __launch_bounds__(256) void vox_insert(const float* __restrict__ pts, const int32_t* __restrict__ off,
                                                  VoxParams P, long long* __restrict__ keys, int32_t* __restrict__ lists,
                                                  int32_t* __restrict__ cellslot) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= P.n) return;
  const float* p = pts + (size_t)i * P.ld + P.xyz_col;
  int c[3];
  bool ok = true;
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    float q = (p[j] - P.lo[j]) / P.vs[j];  // IEEE fp32 divide, as the reference's float division
    int cj = (int)floorf(q);
    ok = ok && (cj >= 0) && (cj < P.grid[j]);
    c[j] = cj;
  }
  if (!ok) {
    cellslot[i] = -1;
    return;
  }
  int b = scene_of(off, P.batch, i);
  // 64-bit cell keys: batch * grid volume may exceed 2^31 (the KITTI detection grid does from 24 scenes per batch on)
  const int lin = (c[2] * P.grid[1] + c[1]) * P.grid[0] + c[0];
  const long long key = (long long)b * P.vol + lin;
  unsigned slot = btc_hash32((unsigned)lin ^ ((unsigned)b * 0x9E3779B9u)) & P.mask;
  while (true) {
    const long long prev = (long long)atomicCAS((unsigned long long*)&keys[slot], (unsigned long long)VOX_EMPTY_KEY, (unsigned long long)key);
    if (prev == VOX_EMPTY_KEY || prev == key) break;
    slot = (slot + 1) & P.mask;
  }
  cellslot[i] = (int)slot;
  // cascade insert: slot s keeps the (s+1)-th smallest index seen so far
  int v = i;
  int32_t* l = lists + (size_t)slot * P.max_points;
  for (int s = 0; s < P.max_points; ++s) {
    int old = atomicMin(&l[s], v);
    if (old > v) v = old;
    if (v == BTC_EMPTY_IDX) break;
  }
}

// flags[i] = 1 iff point i is the first point of its voxel; flags has n+1 entries (last = 0)
__global__ __launch_bounds__(256) void vox_flags(const int32_t* __restrict__ cellslot, const int32_t* __restrict__ lists,
                                                 int n, int max_points, int32_t* __restrict__ flags) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i > n) return;
  int f = 0;
  if (i < n) {
    int s = cellslot[i];
    if (s >= 0) f = (lists[(size_t)s * max_points] == i);
  }
  flags[i] = f;
}

// scene_info[b] = {first-appearance rank of the scene's first voxel, output row base}; total -> d_total
__global__ void vox_scene_info(const int32_t* __restrict__ excl, const int32_t* __restrict__ off, int batch,
                               int max_voxels, int32_t* __restrict__ scene_info, int32_t* __restrict__ d_total) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  int base = 0;
  for (int b = 0; b < batch; ++b) {
    int r0 = excl[off[b]], r1 = excl[off[b + 1]];
    int kept = min(r1 - r0, max_voxels);
    scene_info[2 * b] = r0;
    scene_info[2 * b + 1] = base;
    base += kept;
  }
  *d_total = base;
}

__global__ __launch_bounds__(256) void vox_assign(const int32_t* __restrict__ cellslot, const int32_t* __restrict__ flags,
                                                  const int32_t* __restrict__ excl, const int32_t* __restrict__ off,
                                                  const int32_t* __restrict__ scene_info, const long long* __restrict__ keys,
                                                  VoxParams P, int32_t* __restrict__ vox_slot, int32_t* __restrict__ coords) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= P.n || !flags[i]) return;
  int b = scene_of(off, P.batch, i);
  int r = excl[i] - scene_info[2 * b];
  if (r >= P.max_voxels) return;  // voxel refused by the cap: its slot never gets an output row
  int vid = scene_info[2 * b + 1] + r;
  int slot = cellslot[i];
  vox_slot[vid] = slot;
  int lin = (int)(keys[slot] - (long long)b * P.vol);
  int x = lin % P.grid[0];
  int y = (lin / P.grid[0]) % P.grid[1];
  int z = lin / (P.grid[0] * P.grid[1]);
  reinterpret_cast<int4*>(coords)[vid] = make_int4(b, z, y, x);
}

__global__ __launch_bounds__(256) void vox_fill(const float* __restrict__ pts, const int32_t* __restrict__ vox_slot,
                                                const int32_t* __restrict__ lists, const int32_t* __restrict__ d_total,
                                                VoxParams P, float* __restrict__ voxels, int32_t* __restrict__ num) {
  long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  int total = *d_total;
  if (t >= (long long)total * P.max_points) return;
  int vid = (int)(t / P.max_points), s = (int)(t % P.max_points);
  const int32_t* l = lists + (size_t)vox_slot[vid] * P.max_points;
  int pi = l[s];
  float* o = voxels + (size_t)t * P.C;
  if (pi != BTC_EMPTY_IDX) {
    const float* src = pts + (size_t)pi * P.ld + P.feat_col;
    for (int c = 0; c < P.C; ++c) o[c] = src[c];
  } else {
    for (int c = 0; c < P.C; ++c) o[c] = 0.0f;
  }
  if (s == 0) {
    int cnt = 0;
    for (int q = 0; q < P.max_points; ++q) cnt += (l[q] != BTC_EMPTY_IDX);
    num[vid] = cnt;
  }
}

__global__ __launch_bounds__(256) void cart_to_occ(const float* __restrict__ in, float* __restrict__ out, int n, int ld,
                                                   int mode) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float* p = in + (size_t)i * ld;
  float* o = out + (size_t)i * ld;
  float x = p[0], y = p[1], z = p[2];
  // numpy: np.linalg.norm(points[:, :2], axis=1) = sqrt(x*x + y*y) in fp32 (no fused multiply-add)
  float xy2 = __fadd_rn(__fmul_rn(x, x), __fmul_rn(y, y));
  float xyd = sqrtf(xy2);
  const float rad2deg_num = 180.0f;
  const float pi_f = 3.14159274101257324f;  // float32(np.pi)
  float az = __fdiv_rn(__fmul_rn(atan2f(-y, x), rad2deg_num), pi_f);
  float o0, o2;
  if (mode == 1) {
    o0 = xyd;
    o2 = z;
  } else {
    o0 = sqrtf(__fadd_rn(xy2, __fmul_rn(z, z)));
    o2 = __fdiv_rn(__fmul_rn(atan2f(z, xyd), rad2deg_num), pi_f);
  }
  o[0] = o0;
  o[1] = az;
  o[2] = o2;
  for (int c = 3; c < ld; ++c) o[c] = p[c];
}

__global__ __launch_bounds__(256) void voxel_shift_col(float* __restrict__ voxels, const int32_t* __restrict__ coords, int m,
                                                       int max_points, int C, int col, const float* __restrict__ rot,
                                                       float sign) {
  long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (long long)m * max_points) return;
  int vid = (int)(t / max_points);
  int b = coords[(size_t)vid * 4];
  float* v = voxels + (size_t)t * C + col;
  *v = __fadd_rn(*v, __fmul_rn(sign, rot[b]));
}

}  // namespace

extern "C" size_t btc_voxelize_ws_bytes(int n, int batch, int max_points) {
  unsigned cap = btc_pow2_ge((unsigned long long)(n > 0 ? n : 1) * 2);
  size_t s = 0;
  s += btc_align((size_t)cap * sizeof(long long));             // keys (64-bit: batch * grid volume may exceed 2^31)
  s += btc_align((size_t)cap * max_points * sizeof(int32_t));  // lists
  s += btc_align((size_t)(n + 1) * sizeof(int32_t));           // cellslot
  s += btc_align((size_t)(n + 1) * sizeof(int32_t));           // flags
  s += btc_align((size_t)(n + 1) * sizeof(int32_t));           // excl
  s += btc_align((size_t)(n + 1) * sizeof(int32_t));           // vox_slot (<= n voxels)
  s += btc_align((size_t)batch * 2 * sizeof(int32_t));         // scene_info
  s += btc_scan_ws_bytes(n + 1);
  return s;
}

extern "C" int btc_voxelize(const float* points, int n, int ld, int xyz_col, int feat_col, int C,
                            const int32_t* scene_offsets, int batch, const float* h_range, const float* h_vsize,
                            const int32_t* h_grid, int max_points, int max_voxels, float* voxels, int32_t* coords,
                            int32_t* num, int32_t* d_total, void* ws, size_t ws_bytes, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  BTC_CHECK_ARG(n >= 0 && batch >= 1 && max_points >= 1 && max_voxels >= 1, "btc_voxelize: bad sizes");
  BTC_CHECK_ARG(xyz_col >= 0 && xyz_col + 3 <= ld && feat_col >= 0 && feat_col + C <= ld, "btc_voxelize: bad columns");
  BTC_CHECK_ARG(ws_bytes >= btc_voxelize_ws_bytes(n, batch, max_points), "btc_voxelize: workspace too small");
  long long vol = (long long)h_grid[0] * h_grid[1] * h_grid[2];
  if (vol >= 0x7fffffffLL) {
    btc_set_error("btc_voxelize: one scene's grid of %lld cells exceeds the 31-bit per-scene cell index", vol);
    return BTC_ERANGE;
  }
  if (n == 0) {
    BTC_HIP(hipMemsetAsync(d_total, 0, sizeof(int32_t), stream));
    return BTC_OK;
  }
  VoxParams P;
  for (int j = 0; j < 3; ++j) {
    P.lo[j] = h_range[j];
    P.vs[j] = h_vsize[j];
    P.grid[j] = h_grid[j];
  }
  P.vol = (int)vol;
  P.ld = ld; P.xyz_col = xyz_col; P.feat_col = feat_col; P.C = C;
  P.n = n; P.batch = batch; P.max_points = max_points; P.max_voxels = max_voxels;
  unsigned cap = btc_pow2_ge((unsigned long long)n * 2);
  P.mask = cap - 1;

  BtcCarver cv(ws);
  long long* keys = cv.take<long long>(cap);
  int32_t* lists = cv.take<int32_t>((size_t)cap * max_points);
  int32_t* cellslot = cv.take<int32_t>(n + 1);
  int32_t* flags = cv.take<int32_t>(n + 1);
  int32_t* excl = cv.take<int32_t>(n + 1);
  int32_t* vox_slot = cv.take<int32_t>(n + 1);
  int32_t* scene_info = cv.take<int32_t>((size_t)batch * 2);
  void* scan_ws = cv.take<char>(btc_scan_ws_bytes(n + 1));

  // keys | lists are carved back to back and share one byte pattern (an empty key is 0x7F7F.., above every real key: keys are below
  // batch x 2^31; an empty list entry is BTC_EMPTY_IDX = 0x7F7F7F7F): ONE fill
  BTC_HIP(hipMemsetAsync(keys, 0x7F, (size_t)((char*)(lists + (size_t)cap * max_points) - (char*)keys), stream));
  const int T = 256;
  vox_insert<<<btc_cdiv(n, T), T, 0, stream>>>(points, scene_offsets, P, keys, lists, cellslot);
  BTC_LAUNCH_CHECK();
  vox_flags<<<btc_cdiv(n + 1, T), T, 0, stream>>>(cellslot, lists, n, max_points, flags);
  BTC_LAUNCH_CHECK();
  int rc = btc_scan_exclusive_i32(flags, excl, n + 1, nullptr, scan_ws, stream);
  if (rc) return rc;
  vox_scene_info<<<1, 64, 0, stream>>>(excl, scene_offsets, batch, max_voxels, scene_info, d_total);
  BTC_LAUNCH_CHECK();
  vox_assign<<<btc_cdiv(n, T), T, 0, stream>>>(cellslot, flags, excl, scene_offsets, scene_info, keys, P, vox_slot, coords);
  BTC_LAUNCH_CHECK();
  // at most min(n, batch*max_voxels) voxels exist; threads beyond *d_total exit
  long long cap_vox = (long long)batch * max_voxels;
  if (cap_vox > n) cap_vox = n;
  vox_fill<<<btc_cdiv(cap_vox * max_points, T), T, 0, stream>>>(points, vox_slot, lists, d_total, P, voxels, num);
  BTC_LAUNCH_CHECK();
  return BTC_OK;
}

extern "C" int btc_cart_to_occ_coords(const float* in, float* out, int n, int ld, int mode, void* stream) {
  BTC_CHECK_ARG(mode == 1 || mode == 2, "btc_cart_to_occ_coords: mode must be 1 (cylinder) or 2 (sphere)");
  BTC_CHECK_ARG(ld >= 3, "btc_cart_to_occ_coords: ld < 3");
  if (n <= 0) return BTC_OK;
  cart_to_occ<<<btc_cdiv(n, 256), 256, 0, (hipStream_t)stream>>>(in, out, n, ld, mode);
  BTC_LAUNCH_CHECK();
  return BTC_OK;
}

extern "C" int btc_voxel_shift_col(float* voxels, const int32_t* coords, int m, int max_points, int C, int col,
                                   const float* rot_by_batch, float sign, void* stream) {
  BTC_CHECK_ARG(col >= 0 && col < C, "btc_voxel_shift_col: bad column");
  if (m <= 0) return BTC_OK;
  voxel_shift_col<<<btc_cdiv((long long)m * max_points, 256), 256, 0, (hipStream_t)stream>>>(voxels, coords, m, max_points,
                                                                                             C, col, rot_by_batch, sign);
  BTC_LAUNCH_CHECK();
  return BTC_OK;
}
