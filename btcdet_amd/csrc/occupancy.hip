// Occupancy / occlusion target generator for gfx950.
//
// Replaces OccTargets3D.forward -> create_voxel_res_label
// (/root/reference/btcdet/models/occ_pnt/occ_training_targets/occ_targets_3d.py:18-171 and
// occ_targets_template.py:82-255,330-447): there, hundreds of small torch ops, a 225x index
// expansion (86 MB of int64 for the vcc dilation), a python loop over the batch and ~17 nonzero()
// host syncs.  Here: seven launches, no host sync, one pass over the points and one over the
// 590 K-cell occupancy grid.
//
//   occ_point_pass   per (voxel, slot): cylinder payload -> absolute xyz (written back), voxel
//                    mask, spherical support-map scatter, points-in-boxes, foreground / mirrored
//                    point accumulation (float atomics into per-cell sums)
//   occ_vcc_dilate   per (voxel, offset): the 5x9x5 "visible cell context" dilation as byte stores
//   occ_ray_count / occ_ray_project   per spherical ray (b, elevation, azimuth): hit count, the
//                    EMPT_SUR_THRESH empty-ray fix, prefix-OR along range (cumsum > 0.9) with a
//                    wave64 ballot, and back-projection of every occluded sphere cell corner into
//                    the cylinder grid
//   occ_bm_pass      per best-match template point: in-box test, cell accumulation
//   occ_colmin       per (b, x): lowest occupied cell centre (height gating of filter_occ)
//   occ_finalize     per cell: forebox label, every loss mask / weight map and res_mtrx
//
// Floating point follows the reference's operation order in fp32; transcendental results can differ
// from the CPU libm by an ulp, which moves a few boundary cells (see tests/test_hip_occupancy.py).
#include "btc_common.h"

namespace {

struct OccParams {
  int B, nz, ny, nx;        // cylinder grid
  int snz, sny, snx;        // spherical support grid
  float origin[3], pmax[3], vs[3];        // cylinder range / voxel size (x = rho, y = azimuth deg, z)
  float s_origin[3], s_max[3], s_vs[3];   // sphere range (r, azimuth deg, elevation deg)
  float det_zmin, det_zmax;
  int kz, ky, kx, x0;       // dilation kernel and the first x offset (-(kx/2) + concede_x)
  int empt_thresh;          // EMPT_SUR_THRESH, < 0 = disabled
  int G;                    // padded boxes per scene
  float w_fore_cls, w_mirr_cls, w_bm_cls, w_neg_cls, w_fore_res, w_mirr_res, w_bm_res, box_weight;
  int use_box_weight;
  int reverse_vis;          // REVERSE_VIS: 0 NOTHING, 1 VCC, 2 BACK_TRACK (occ_targets_template.py:110-134)
  int vis_half;             // VCC: cells in front of a hit that stay visible = (DIST_KERN[2] + 1) / 2
};

constexpr float kPi = 3.14159274101257324f;        // float32(np.pi)
constexpr float kRad2Deg = 57.2957801818847656f;   // float32(180. / np.pi)

// Correctly-rounded fp32 transcendentals (evaluate in fp64, round once).  The CPU reference (torch / sleef,
// <= 1 ulp, correctly rounded for the vast majority of arguments) quantises these values on cell boundaries,
// so every ulp of disagreement moves boundary cells; the ocml fp32 routines disagree far more often.
__device__ __forceinline__ float cr_cos(float v) { return (float)cos((double)v); }
__device__ __forceinline__ float cr_sin(float v) { return (float)sin((double)v); }
__device__ __forceinline__ float cr_atan2(float y, float x) { return (float)atan2((double)y, (double)x); }

__device__ __forceinline__ float deg2rad(float d) { return __fdiv_rn(__fmul_rn(d, kPi), 180.0f); }  // v * pi / 180.

// (p - lo) / vs truncated toward zero, clamped; inclusive range test (occ_targets_template.py:82-90)
__device__ __forceinline__ bool cell_inrange(const float p[3], const float lo[3], const float hi[3], const float vs[3],
                                             const int n[3], int c[3]) {
  bool ok = true;
#pragma unroll
  for (int j = 0; j < 3; ++j) ok = ok && (p[j] >= lo[j]) && (p[j] <= hi[j]);
  if (!ok) return false;
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    int q = (int)__fdiv_rn(__fsub_rn(p[j], lo[j]), vs[j]);
    c[j] = min(max(q, 0), n[j] - 1);
  }
  return true;
}

__device__ __forceinline__ void cart_to_cyl(float x, float y, float z, float out[3]) {
  out[0] = sqrtf(__fadd_rn(__fmul_rn(x, x), __fmul_rn(y, y)));
  out[1] = __fmul_rn(cr_atan2(-y, x), kRad2Deg);
  out[2] = z;
}

// in-box test in the box frame; local = R^T (p - c)
__device__ __forceinline__ bool in_box(const float* __restrict__ box, float x, float y, float z, float loc[3]) {
  float dx = x - box[0], dy = y - box[1], dz = z - box[2];
  float c = cr_cos(box[6]), s = cr_sin(box[6]);
  loc[0] = c * dx + s * dy;
  loc[1] = -s * dx + c * dy;
  loc[2] = dz;
  return fabsf(loc[0]) <= box[3] * 0.5f && fabsf(loc[1]) <= box[4] * 0.5f && fabsf(loc[2]) <= box[5] * 0.5f;
}

__device__ __forceinline__ size_t cell_index(const OccParams& P, int b, int z, int y, int x) {
  return (((size_t)b * P.nz + z) * P.ny + y) * P.nx + x;
}

__device__ __forceinline__ void accumulate(float* __restrict__ sums, int32_t* __restrict__ cnt, const OccParams& P, int b, int z,
                                           int y, int x, float px, float py, float pz) {
  size_t vol = (size_t)P.nz * P.ny * P.nx;
  size_t sp = ((size_t)z * P.ny + y) * P.nx + x;
  float* s = sums + (size_t)b * 3 * vol + sp;
  atomicAdd(s, px);
  atomicAdd(s + vol, py);
  atomicAdd(s + 2 * vol, pz);
  atomicAdd(cnt + (size_t)b * vol + sp, 1);
}

__global__ __launch_bounds__(256) void occ_point_pass(float* __restrict__ voxels, const int32_t* __restrict__ coords,
                                                      const int32_t* __restrict__ num, int M, int Pn, int C,
                                                      const float* __restrict__ gt, const int32_t* __restrict__ gt_num,
                                                      const float* __restrict__ mirr_flag, const float* __restrict__ rot_z, OccParams P,
                                                      uint8_t* __restrict__ voxelwise, uint8_t* __restrict__ smap,
                                                      uint8_t* __restrict__ fore_mask, uint8_t* __restrict__ mirr_mask,
                                                      float* __restrict__ fore_sum, int32_t* __restrict__ fore_cnt,
                                                      float* __restrict__ mirr_sum, int32_t* __restrict__ mirr_cnt) {
  long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (long long)M * Pn) return;
  int v = (int)(t / Pn), slot = (int)(t % Pn);
  float* p = voxels + (size_t)t * C;
  // cylinder_uvd2absxyz on every slot, padded ones included (occ_targets_3d.py:45-47)
  float rho = p[0], th = p[1], z = p[2];
  float a = deg2rad(th);
  float x = __fmul_rn(rho, cr_cos(a)), y = __fmul_rn(-rho, cr_sin(a));
  p[0] = x;
  p[1] = y;
  int4 c = reinterpret_cast<const int4*>(coords)[v];
  if (slot == 0) voxelwise[cell_index(P, c.x, c.y, c.z, c.w)] = 1;
  if (slot >= num[v]) return;
  const int b = c.x;
  const float rz = rot_z[b];
  // ---- spherical support map (occ_targets_template.py:137-144)
  {
    float xy2 = __fadd_rn(__fmul_rn(x, x), __fmul_rn(y, y));
    float sp[3];
    sp[0] = sqrtf(__fadd_rn(xy2, __fmul_rn(z, z)));
    sp[1] = __fadd_rn(__fmul_rn(cr_atan2(-y, x), kRad2Deg), rz);
    sp[2] = __fmul_rn(cr_atan2(z, sqrtf(xy2)), kRad2Deg);
    int sc[3];
    const int sn[3] = {P.snx, P.sny, P.snz};
    if (cell_inrange(sp, P.s_origin, P.s_max, P.s_vs, sn, sc))
      smap[(((size_t)b * P.snz + sc[2]) * P.sny + sc[1]) * P.snx + sc[0]] = 1;
  }
  // ---- foreground / mirrored points (point_box_utils.py:252-306)
  const int ng = gt_num[b];
  bool fore = false;
  for (int g = 0; g < ng; ++g) {
    const float* box = gt + ((size_t)b * P.G + g) * 8;
    float loc[3];
    if (!in_box(box, x, y, z, loc)) continue;
    if (box[7] > 1e-2f) fore = true;
    if (mirr_flag[(size_t)b * P.G + g] > 0.5f) {
      float cs = cr_cos(box[6]), sn = cr_sin(box[6]);
      float ly = -loc[1];
      float mx = cs * loc[0] - sn * ly + box[0];
      float my = sn * loc[0] + cs * ly + box[1];
      float mz = loc[2] + box[2];
      float cyl[3];
      cart_to_cyl(mx, my, mz, cyl);
      cyl[1] = __fadd_rn(cyl[1], rz);
      int cc[3];
      const int n3[3] = {P.nx, P.ny, P.nz};
      if (cell_inrange(cyl, P.origin, P.pmax, P.vs, n3, cc)) {
        mirr_mask[cell_index(P, b, cc[2], cc[1], cc[0])] = 1;
        accumulate(mirr_sum, mirr_cnt, P, b, cc[2], cc[1], cc[0], mx, my, mz);
      }
    }
  }
  if (fore) {
    fore_mask[cell_index(P, b, c.y, c.z, c.w)] = 1;
    accumulate(fore_sum, fore_cnt, P, b, c.y, c.z, c.w, x, y, z);
  }
}

__global__ __launch_bounds__(256) void occ_vcc_dilate(const int32_t* __restrict__ coords, int M, OccParams P,
                                                      uint8_t* __restrict__ vcc) {
  const int KK = P.kz * P.ky * P.kx;
  long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (long long)M * KK) return;
  int v = (int)(t / KK), o = (int)(t % KK);
  int4 c = reinterpret_cast<const int4*>(coords)[v];
  int ox = o % P.kx, oy = (o / P.kx) % P.ky, oz = o / (P.kx * P.ky);
  int z = c.y + oz - P.kz / 2, y = c.z + oy - P.ky / 2, x = c.w + ox + P.x0;
  // the reference clamps instead of dropping; the clamped targets are a subset of the in-range ones
  if (z < 0 || z >= P.nz || y < 0 || y >= P.ny || x < 0 || x >= P.nx) return;
  vcc[cell_index(P, c.x, z, y, x)] = 1;
}

// hits per ray (b, el, az): one wave per ray
__global__ __launch_bounds__(256) void occ_ray_count(const uint8_t* __restrict__ smap, OccParams P, int32_t* __restrict__ ray_cnt) {
  int ray = blockIdx.x * 4 + (threadIdx.x >> 6);
  int lane = threadIdx.x & 63;
  int nrays = P.B * P.snz * P.sny;
  if (ray >= nrays) return;
  const uint8_t* r = smap + (size_t)ray * P.snx;
  int c = 0;
  for (int x = lane; x < P.snx; x += 64) c += r[x];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) c += __shfl_down(c, o, 64);
  if (lane == 0) ray_cnt[ray] = c;
}

// per ray: empty-ray fix (occ_targets_template.py:128-130,186-191), prefix OR along range
// (cumsum > 0.9, :133-134), back-projection of the occluded cell corners (:145-154)
// back-projection of ONE sphere cell corner (sz, sy, x) into the cylinder grid -> linear cell z*ny*nx + y*nx + x, or -1 out of range
// (occ_targets_template.py:146-152: idx * voxel + origin, sphere_uvd2absxyz, cartesian_cylinder_coords, point2coords_inrange);
// (ce, se, ca, sa) = cos / sin of the corner's elevation and azimuth
__device__ __forceinline__ int backproject_cell(const OccParams& P, int x, float ce, float se, float ca, float sa) {
  const float rr = __fadd_rn(__fmul_rn((float)x, P.s_vs[0]), P.s_origin[0]);
  const float xyd = __fmul_rn(rr, ce);
  const float px = __fmul_rn(xyd, ca), py = __fmul_rn(-xyd, sa), pz = __fmul_rn(rr, se);
  float cyl[3];
  cart_to_cyl(px, py, pz, cyl);
  int cc[3];
  const int n3[3] = {P.nx, P.ny, P.nz};
  if (!cell_inrange(cyl, P.origin, P.pmax, P.vs, n3, cc)) return -1;
  return (cc[2] * P.ny + cc[1]) * P.nx + cc[0];
}

// The back-projection is a function of the two grids alone -- the corner lattice of the sphere grid does not move with the batch
// (rot_z only enters the FORWARD projection of the points) -- so it is a static table: lut[sz][sy][x] = cylinder cell or -1.
// This kernel fills it with the device's correctly-rounded transcendentals (the same arithmetic occ_ray_project evaluates inline
// when no table is given); the host side may instead supply a table evaluated with the arithmetic of whatever platform the
// reference is to be reproduced on (btcdet_amd/occ_targets.py: torch's own CPU kernels).
__global__ __launch_bounds__(256) void occ_backproject_lut(OccParams P, int32_t* __restrict__ lut) {
  const int ray = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (ray >= P.snz * P.sny) return;
  const int sy = ray % P.sny, sz = ray / P.sny;
  const float el = __fadd_rn(__fmul_rn((float)sz, P.s_vs[2]), P.s_origin[2]);
  const float az = __fadd_rn(__fmul_rn((float)sy, P.s_vs[1]), P.s_origin[1]);
  const float ce = cr_cos(deg2rad(el)), se = cr_sin(deg2rad(el)), ca = cr_cos(deg2rad(az)), sa = cr_sin(deg2rad(az));
  for (int x = lane; x < P.snx; x += 64) lut[(size_t)ray * P.snx + x] = backproject_cell(P, x, ce, se, ca, sa);
}

__global__ __launch_bounds__(256) void occ_ray_project(const uint8_t* __restrict__ smap, const int32_t* __restrict__ ray_cnt,
                                                       OccParams P, const int32_t* __restrict__ lut, uint8_t* __restrict__ occ_raw) {
  int ray = blockIdx.x * 4 + (threadIdx.x >> 6);
  int lane = threadIdx.x & 63;
  int nrays = P.B * P.snz * P.sny;
  if (ray >= nrays) return;
  int sy = ray % P.sny, sz = (ray / P.sny) % P.snz, b = ray / (P.sny * P.snz);
  const uint8_t* r = smap + (size_t)ray * P.snx;
  // hits of the ray as bit masks, 64 range bins a word (snx <= 512); column 0 carries the EMPT_SUR_THRESH fix in the configured mode only
  // (the reference applies it in the NOTHING branch of occ_from_sphere_ocp alone, occ_targets_template.py:128-130)
  int col0 = r[0];
  if (P.reverse_vis == 0 && P.empt_thresh >= 0) {
    int neigh = 0;
    for (int dz = -1; dz <= 1; ++dz)
      for (int dy = -1; dy <= 1; ++dy) {
        int zz = sz + dz, yy = sy + dy;
        if (zz >= 0 && zz < P.snz && yy >= 0 && yy < P.sny) neigh += ray_cnt[((size_t)b * P.snz + zz) * P.sny + yy];
      }
    col0 = (ray_cnt[ray] == 0) && (neigh > P.empt_thresh);
  }
  constexpr int MAXW = 8;
  unsigned long long hit[MAXW + 1], sel[MAXW];
  const int nw = (P.snx + 63) >> 6;
  int first = P.snx;
#pragma unroll
  for (int w = 0; w < MAXW; ++w) {
    hit[w] = 0ull;
    if (w < nw) {
      const int x = w * 64 + lane;
      int bit = 0;
      if (x < P.snx) bit = (x == 0) ? col0 : r[x];
      hit[w] = __ballot(bit != 0);
      if (hit[w] && first == P.snx) first = w * 64 + __ffsll((long long)hit[w]) - 1;
    }
  }
  hit[MAXW] = 0ull;
  if (P.reverse_vis == 1) {
    // VCC: every cell counts except the vis_half cells IN FRONT of a hit (x + 1 .. x + vis_half holds a hit), unless hit itself
#pragma unroll
    for (int w = 0; w < MAXW; ++w) {
      unsigned long long front = 0ull;
      for (int d = 1; d <= P.vis_half && d < 64; ++d) front |= (hit[w] >> d) | (hit[w + 1] << (64 - d));
      sel[w] = hit[w] | ~front;
    }
  } else {
    // NOTHING: at or behind the first hit (none: nothing).  BACK_TRACK: behind the last hit or at / behind the first = the same set on a
    // ray with a hit, and EVERY cell of a ray without one
    if (first >= P.snx) {
      if (P.reverse_vis != 2) return;
      first = 0;
    }
#pragma unroll
    for (int w = 0; w < MAXW; ++w) {
      const int lo = first - w * 64;
      sel[w] = lo <= 0 ? ~0ull : (lo >= 64 ? 0ull : (~0ull << lo));
    }
  }
  uint8_t* occ_b = occ_raw + (size_t)b * P.nz * P.ny * P.nx;
  if (lut) {   // static table of the corner lattice (coalesced 4-byte reads, no transcendental in the step)
    const int32_t* l = lut + ((size_t)sz * P.sny + sy) * P.snx;
#pragma unroll
    for (int w = 0; w < MAXW; ++w) {
      const int x = w * 64 + lane;
      if (w < nw && x < P.snx && ((sel[w] >> lane) & 1ull)) {
        const int c = l[x];
        if (c >= 0) occ_b[c] = 1;
      }
    }
    return;
  }
  // corner (no +0.5) of the sphere cell: idx * voxel + origin (occ_targets_template.py:147)
  float el = __fadd_rn(__fmul_rn((float)sz, P.s_vs[2]), P.s_origin[2]);
  float az = __fadd_rn(__fmul_rn((float)sy, P.s_vs[1]), P.s_origin[1]);
  float ce = cr_cos(deg2rad(el)), se = cr_sin(deg2rad(el)), ca = cr_cos(deg2rad(az)), sa = cr_sin(deg2rad(az));
#pragma unroll
  for (int w = 0; w < MAXW; ++w) {
    const int x = w * 64 + lane;
    if (w < nw && x < P.snx && ((sel[w] >> lane) & 1ull)) {
      const int c = backproject_cell(P, x, ce, se, ca, sa);
      if (c >= 0) occ_b[c] = 1;
    }
  }
}

__global__ __launch_bounds__(256) void occ_bm_pass(const float* __restrict__ bm, int n, const float* __restrict__ gt,
                                                   const int32_t* __restrict__ gt_num, const float* __restrict__ rot_z, OccParams P,
                                                   uint8_t* __restrict__ bm_mask, float* __restrict__ bm_sum,
                                                   int32_t* __restrict__ bm_cnt) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float* q = bm + (size_t)i * 4;
  int b = (int)q[0];
  if (b < 0 || b >= P.B) return;
  float x = q[1], y = q[2], z = q[3];
  bool in = false;
  const int ng = gt_num[b];
  for (int g = 0; g < ng && !in; ++g) {
    const float* box = gt + ((size_t)b * P.G + g) * 8;
    float loc[3];
    in = in_box(box, x, y, z, loc) && box[7] > 1e-2f;
  }
  if (!in) return;
  float cyl[3];
  cart_to_cyl(x, y, z, cyl);
  cyl[1] = __fadd_rn(cyl[1], rot_z[b]);
  int cc[3];
  const int n3[3] = {P.nx, P.ny, P.nz};
  if (cell_inrange(cyl, P.origin, P.pmax, P.vs, n3, cc)) {
    bm_mask[cell_index(P, b, cc[2], cc[1], cc[0])] = 1;
    accumulate(bm_sum, bm_cnt, P, b, cc[2], cc[1], cc[0], x, y, z);
  }
}

// colmin[b][x] = min over (z,y) of (occupied ? centre_z : 100 + centre_z)  (occ_targets_template.py:251-252)
// block = 64 x-columns x 16 slices of the (z,y) plane, LDS min-reduce over the slices
__global__ __launch_bounds__(1024) void occ_colmin(const uint8_t* __restrict__ voxelwise, OccParams P, float* __restrict__ colmin) {
  __shared__ float s_min[16][64];
  const int lx = threadIdx.x & 63, part = threadIdx.x >> 6;
  const int xblocks = (P.nx + 63) / 64;
  const int b = blockIdx.x / xblocks, x = (blockIdx.x % xblocks) * 64 + lx;
  float m = 3.0e38f;
  if (x < P.nx) {
    const int zy = P.nz * P.ny;
    for (int q = part; q < zy; q += 16) {
      int z = q / P.ny;
      float cz = __fadd_rn(__fmul_rn(__fadd_rn(0.5f, (float)z), P.vs[2]), P.origin[2]);
      float v = voxelwise[((size_t)b * zy + q) * P.nx + x] ? cz : __fadd_rn(100.0f, cz);
      m = fminf(m, v);
    }
  }
  s_min[part][lx] = m;
  __syncthreads();
  if (part == 0 && x < P.nx) {
#pragma unroll
    for (int q = 1; q < 16; ++q) m = fminf(m, s_min[q][lx]);
    colmin[b * P.nx + x] = m;
  }
}

struct OccOut {
  uint8_t *vcc, *voxelwise, *bm_mask, *occ, *fore_mask, *pos, *gen, *f_cls, *m_cls, *b_cls, *reg_m;
  int8_t* forebox;
  float *cls_w, *reg_w, *res;
  int32_t* pos_all_num;
};

__device__ __forceinline__ void mean_res(const float* __restrict__ sums, const int32_t* __restrict__ cnt, const OccParams& P, int b,
                                         size_t sp, size_t vol, float cx, float cy, float cz, float out[3]) {
  int n = cnt[(size_t)b * vol + sp];
  if (n <= 0) {
    out[0] = out[1] = out[2] = 0.f;
    return;
  }
  const float* s = sums + (size_t)b * 3 * vol + sp;
  float fn = (float)n;
  out[0] = __fsub_rn(__fdiv_rn(s[0], fn), cx);
  out[1] = __fsub_rn(__fdiv_rn(s[vol], fn), cy);
  out[2] = __fsub_rn(__fdiv_rn(s[2 * vol], fn), cz);
}

__global__ __launch_bounds__(256) void occ_finalize(const float* __restrict__ gt, const int32_t* __restrict__ gt_num,
                                                    const float* __restrict__ rot_z, const float* __restrict__ centers /* nz,ny,nx,3 */,
                                                    OccParams P, const uint8_t* __restrict__ occ_raw, const float* __restrict__ colmin,
                                                    const uint8_t* __restrict__ mirr_raw, const float* __restrict__ fore_sum,
                                                    const int32_t* __restrict__ fore_cnt, const float* __restrict__ mirr_sum,
                                                    const int32_t* __restrict__ mirr_cnt, const float* __restrict__ bm_sum,
                                                    const int32_t* __restrict__ bm_cnt, OccOut O) {
  const size_t vol = (size_t)P.nz * P.ny * P.nx;
  size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  int pos_any = 0;
  if (t < (size_t)P.B * vol) {
    int b = (int)(t / vol);
    size_t sp = t % vol;
    int x = (int)(sp % P.nx), y = (int)((sp / P.nx) % P.ny), z = (int)(sp / ((size_t)P.nx * P.ny));
    const int vw = O.voxelwise[t];
    // ---- filter_occ (occ_targets_template.py:249-255)
    float cz = __fadd_rn(__fmul_rn(__fadd_rn(0.5f, (float)z), P.vs[2]), P.origin[2]);
    float vz = colmin[b * P.nx + x];
    if (vz > 20.0f) vz = __fsub_rn(vz, 200.0f);
    vz = fmaxf(vz, P.det_zmin);
    const int occ = occ_raw[t] && (cz > vz) && (cz < P.det_zmax);
    // ---- exclusions (occ_targets_3d.py:60-66)
    const int fore = O.fore_mask[t];
    const int mirr = mirr_raw[t] && !vw;
    const int bmm = O.bm_mask[t] && !vw && !mirr;
    // ---- cells inside GT boxes (occ_targets_3d.py:70-86): centre rotated by rot_z into the box frame of reference
    int fb = 0;
    if (P.use_box_weight) {
      const float* ctr = centers + sp * 3;
      float ang = deg2rad(rot_z[b]);
      float cr = cr_cos(ang), sr = cr_sin(ang);
      // rotatez: points @ transpose(yaw_rotation)^T ... = [c*x + s*y? ] -> (x*c - ... ) see point_box_utils.py:241-249
      float rx = ctr[0] * cr + ctr[1] * (-sr);
      float ry = ctr[0] * sr + ctr[1] * cr;
      float rz = ctr[2];
      const int ng = gt_num[b];
      for (int g = 0; g < ng && !fb; ++g) {
        const float* box = gt + ((size_t)b * P.G + g) * 8;
        float loc[3];
        fb = in_box(box, rx, ry, rz, loc) && box[7] > 1e-2f;
      }
    }
    // ---- loss maps (occ_targets_template.py:330-401)
    const int vcc = O.vcc[t];
    const int gen = vcc && occ;
    const int f_cls = fore && gen, m_cls = mirr && gen, b_cls = bmm && gen;
    const int pos = f_cls || m_cls || b_cls;
    const int neg = gen && !pos;
    float cw = (float)f_cls * P.w_fore_cls + (float)m_cls * P.w_mirr_cls + (float)b_cls * P.w_bm_cls + (float)neg * P.w_neg_cls;
    if (P.use_box_weight) cw += (float)(neg && fb) * (P.box_weight - P.w_neg_cls);
    float rw = (float)f_cls * P.w_fore_res + (float)m_cls * P.w_mirr_res + (float)b_cls * P.w_bm_res;
    const int reg_m = rw > 0.f;
    // ---- residual targets: per-cell mean point minus cell centre (occ_targets_3d.py:122-145)
    float res[3] = {0.f, 0.f, 0.f};
    if (reg_m) {
      float u = __fadd_rn(__fmul_rn(__fadd_rn((float)x, 0.5f), P.vs[0]), P.origin[0]);
      float th = __fsub_rn(__fadd_rn(__fmul_rn(__fadd_rn((float)y, 0.5f), P.vs[1]), P.origin[1]), rot_z[b]);
      float a = deg2rad(th);
      float cx = __fmul_rn(u, cr_cos(a)), cy = __fmul_rn(-u, cr_sin(a));
      float cz2 = __fadd_rn(__fmul_rn(__fadd_rn((float)z, 0.5f), P.vs[2]), P.origin[2]);
      float r[3];
      mean_res(fore_sum, fore_cnt, P, b, sp, vol, cx, cy, cz2, r);
      res[0] += r[0]; res[1] += r[1]; res[2] += r[2];
      if (!vw) {
        mean_res(mirr_sum, mirr_cnt, P, b, sp, vol, cx, cy, cz2, r);
        res[0] += r[0]; res[1] += r[1]; res[2] += r[2];
        if (!mirr) {
          mean_res(bm_sum, bm_cnt, P, b, sp, vol, cx, cy, cz2, r);
          res[0] += r[0]; res[1] += r[1]; res[2] += r[2];
        }
      }
    }
    O.occ[t] = occ;
    O.bm_mask[t] = bmm;
    O.pos[t] = pos;
    O.gen[t] = gen;
    O.f_cls[t] = f_cls;
    O.m_cls[t] = m_cls;
    O.b_cls[t] = b_cls;
    O.reg_m[t] = reg_m;
    O.forebox[t] = (int8_t)fb;
    O.cls_w[t] = cw;
    O.reg_w[t] = rw;
    float* rp = O.res + (size_t)b * 3 * vol + sp;
    rp[0] = res[0];
    rp[vol] = res[1];
    rp[2 * vol] = res[2];
    pos_any = fore || mirr || bmm;
  }
  // pos_all_num = sum(fore | mirr | bm) (occ_targets_template.py:376)
  unsigned long long m = __ballot(pos_any);
  if ((threadIdx.x & 63) == 0 && m) atomicAdd(O.pos_all_num, __popcll(m));
}

}  // namespace

static size_t occ_ws_layout(const BtcOccConfig* c, void* ws, uint8_t** smap, uint8_t** occ_raw, uint8_t** mirr_raw, float** sums,
                            int32_t** cnts, int32_t** ray_cnt, float** colmin) {
  size_t vol = (size_t)c->grid[0] * c->grid[1] * c->grid[2] * c->batch;
  size_t svol = (size_t)c->sphere_grid[0] * c->sphere_grid[1] * c->sphere_grid[2] * c->batch;
  size_t rays = (size_t)c->sphere_grid[1] * c->sphere_grid[2] * c->batch;
  BtcCarver cv(ws);
  uint8_t* a = cv.take<uint8_t>(svol);
  uint8_t* b = cv.take<uint8_t>(vol);
  uint8_t* m = cv.take<uint8_t>(vol);
  float* s = cv.take<float>(vol * 9);
  int32_t* n = cv.take<int32_t>(vol * 3);
  int32_t* r = cv.take<int32_t>(rays);
  float* cm = cv.take<float>((size_t)c->batch * c->grid[0]);
  if (smap) { *smap = a; *occ_raw = b; *mirr_raw = m; *sums = s; *cnts = n; *ray_cnt = r; *colmin = cm; }
  return cv.off;
}

extern "C" size_t btc_occ_targets_ws_bytes(const BtcOccConfig* cfg) {
  return occ_ws_layout(cfg, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr);
}

static void occ_params_from(const BtcOccConfig* cfg, OccParams& P) {
  P.B = cfg->batch;
  P.nx = cfg->grid[0]; P.ny = cfg->grid[1]; P.nz = cfg->grid[2];
  P.snx = cfg->sphere_grid[0]; P.sny = cfg->sphere_grid[1]; P.snz = cfg->sphere_grid[2];
  for (int j = 0; j < 3; ++j) {
    P.origin[j] = cfg->occ_range[j]; P.pmax[j] = cfg->occ_range[3 + j]; P.vs[j] = cfg->occ_voxel[j];
    P.s_origin[j] = cfg->sphere_range[j]; P.s_max[j] = cfg->sphere_range[3 + j]; P.s_vs[j] = cfg->sphere_voxel[j];
  }
  P.det_zmin = cfg->det_zmin; P.det_zmax = cfg->det_zmax;
  P.kz = cfg->dist_kern[0]; P.ky = cfg->dist_kern[1]; P.kx = cfg->dist_kern[2];
  P.x0 = -(P.kx / 2) + cfg->concede_x;
  P.empt_thresh = cfg->empt_sur_thresh;
  P.G = cfg->max_boxes;
  P.w_fore_cls = cfg->w_fore_cls; P.w_mirr_cls = cfg->w_mirr_cls; P.w_bm_cls = cfg->w_bm_cls; P.w_neg_cls = cfg->w_neg_cls;
  P.w_fore_res = cfg->w_fore_res; P.w_mirr_res = cfg->w_mirr_res; P.w_bm_res = cfg->w_bm_res; P.box_weight = cfg->box_weight;
  P.use_box_weight = cfg->use_box_weight;
  P.reverse_vis = cfg->reverse_vis;
  P.vis_half = cfg->vis_half;
}

extern "C" int btc_occ_backproject_lut(const BtcOccConfig* cfg, int32_t* lut, void* stream_) {
  BTC_CHECK_ARG(cfg && lut, "btc_occ_backproject_lut: null config / table");
  OccParams P;
  occ_params_from(cfg, P);
  BTC_CHECK_ARG(P.snx > 0 && P.sny > 0 && P.snz > 0, "btc_occ_backproject_lut: empty sphere grid");
  occ_backproject_lut<<<btc_cdiv(P.snz * P.sny, 4), 256, 0, (hipStream_t)stream_>>>(P, lut);
  BTC_LAUNCH_CHECK();
  return BTC_OK;
}

extern "C" int btc_occ_targets(const BtcOccConfig* cfg, float* voxels, const int32_t* voxel_coords, const int32_t* voxel_num,
                               int M, int max_points, int C, const float* gt_boxes, const int32_t* gt_num,
                               const float* mirr_flag, const float* bm_points, int n_bm, const float* rot_z,
                               const float* centers, const BtcOccBuffers* out, void* ws, size_t ws_bytes, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  BTC_CHECK_ARG(cfg && out, "btc_occ_targets: null config / buffers");
  BTC_CHECK_ARG(C >= 3 && M >= 0 && max_points >= 1, "btc_occ_targets: bad voxel layout");
  BTC_CHECK_ARG(ws_bytes >= btc_occ_targets_ws_bytes(cfg), "btc_occ_targets: workspace too small");
  BTC_CHECK_ARG(cfg->sphere_grid[0] <= 512 && cfg->reverse_vis >= 0 && cfg->reverse_vis <= 2 && cfg->vis_half >= 0 && cfg->vis_half < 64,
                "btc_occ_targets: support sphere wider than 512 range bins, or bad REVERSE_VIS settings");
  OccParams P;
  occ_params_from(cfg, P);

  const size_t vol = (size_t)P.nx * P.ny * P.nz * P.B;
  uint8_t *smap, *occ_raw, *mirr_raw;
  float *sums, *colmin;
  int32_t *cnts, *ray_cnt;
  occ_ws_layout(cfg, ws, &smap, &occ_raw, &mirr_raw, &sums, &cnts, &ray_cnt, &colmin);
  float *fore_sum = sums, *mirr_sum = sums + vol * 3, *bm_sum = sums + vol * 6;
  int32_t *fore_cnt = cnts, *mirr_cnt = cnts + vol, *bm_cnt = cnts + vol * 2;

  // What the kernels accumulate into starts zeroed: the workspace's smap | occ_raw | mirr_raw | sums | cnts (carved back to back,
  // occ_ws_layout), the byte masks, the positive count.  A caller that lays them out as ONE arena -- pos_all_num, 256 bytes on the five
  // masks as a (5, vol) block, rounded up to 256 bytes the workspace (btcdet_amd/occ_targets.py) -- gets one fill for all of it
  // (separate tensors of the caching allocator sit on 512-byte blocks and cannot meet these offsets by accident).
  uint8_t* m0 = (uint8_t*)out->vcc_mask;
  const bool block5 = (uint8_t*)out->voxelwise_mask == m0 + vol && (uint8_t*)out->bm_voxelwise_mask == m0 + 2 * vol &&
                      (uint8_t*)out->occ_voxelwise_mask == m0 + 3 * vol && (uint8_t*)out->fore_voxelwise_mask == m0 + 4 * vol;
  if (block5 && (uint8_t*)out->pos_all_num + 256 == m0 && m0 + ((5 * vol + 255) & ~(size_t)255) == (uint8_t*)ws && smap == (uint8_t*)ws) {
    BTC_HIP(hipMemsetAsync(out->pos_all_num, 0, (size_t)((char*)ray_cnt - (char*)out->pos_all_num), stream));
  } else {
    BTC_HIP(hipMemsetAsync(smap, 0, (size_t)((char*)ray_cnt - (char*)smap), stream));
    if (block5) {   // (occ_voxelwise_mask is overwritten by occ_finalize)
      BTC_HIP(hipMemsetAsync(m0, 0, 5 * vol, stream));
    } else {
      BTC_HIP(hipMemsetAsync(out->vcc_mask, 0, vol, stream));
      BTC_HIP(hipMemsetAsync(out->voxelwise_mask, 0, vol, stream));
      BTC_HIP(hipMemsetAsync(out->fore_voxelwise_mask, 0, vol, stream));
      BTC_HIP(hipMemsetAsync(out->bm_voxelwise_mask, 0, vol, stream));
    }
    BTC_HIP(hipMemsetAsync(out->pos_all_num, 0, sizeof(int32_t), stream));
  }

  const int T = 256;
  if (M > 0) {
    occ_point_pass<<<btc_cdiv((long long)M * max_points, T), T, 0, stream>>>(
        voxels, voxel_coords, voxel_num, M, max_points, C, gt_boxes, gt_num, mirr_flag, rot_z, P, out->voxelwise_mask, smap,
        out->fore_voxelwise_mask, mirr_raw, fore_sum, fore_cnt, mirr_sum, mirr_cnt);
    BTC_LAUNCH_CHECK();
    occ_vcc_dilate<<<btc_cdiv((long long)M * P.kz * P.ky * P.kx, T), T, 0, stream>>>(voxel_coords, M, P, out->vcc_mask);
    BTC_LAUNCH_CHECK();
  }
  const int nrays = P.B * P.snz * P.sny;
  occ_ray_count<<<btc_cdiv(nrays, 4), T, 0, stream>>>(smap, P, ray_cnt);
  BTC_LAUNCH_CHECK();
  occ_ray_project<<<btc_cdiv(nrays, 4), T, 0, stream>>>(smap, ray_cnt, P, cfg->backproject_lut, occ_raw);
  BTC_LAUNCH_CHECK();
  if (n_bm > 0) {
    occ_bm_pass<<<btc_cdiv(n_bm, T), T, 0, stream>>>(bm_points, n_bm, gt_boxes, gt_num, rot_z, P, out->bm_voxelwise_mask, bm_sum, bm_cnt);
    BTC_LAUNCH_CHECK();
  }
  occ_colmin<<<P.B * ((P.nx + 63) / 64), 1024, 0, stream>>>(out->voxelwise_mask, P, colmin);
  BTC_LAUNCH_CHECK();
  OccOut O;
  O.vcc = out->vcc_mask; O.voxelwise = out->voxelwise_mask; O.bm_mask = out->bm_voxelwise_mask; O.occ = out->occ_voxelwise_mask;
  O.fore_mask = out->fore_voxelwise_mask; O.pos = out->pos_mask; O.gen = out->general_cls_loss_mask;
  O.f_cls = out->occ_fore_cls_mask; O.m_cls = out->occ_mirr_cls_mask; O.b_cls = out->occ_bm_cls_mask;
  O.reg_m = out->general_reg_loss_mask; O.forebox = out->forebox_label; O.cls_w = out->general_cls_loss_mask_float;
  O.reg_w = out->general_reg_loss_mask_float; O.res = out->res_mtrx; O.pos_all_num = out->pos_all_num;
  occ_finalize<<<btc_cdiv((long long)vol, T), T, 0, stream>>>(gt_boxes, gt_num, rot_z, centers, P, occ_raw, colmin, mirr_raw, fore_sum,
                                                             fore_cnt, mirr_sum, mirr_cnt, bm_sum, bm_cnt, O);
  BTC_LAUNCH_CHECK();
  return BTC_OK;
}
