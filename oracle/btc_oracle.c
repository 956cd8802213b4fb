/*
 * btc_oracle.c -- CPU ORACLE (TEST INFRASTRUCTURE ONLY, never shipped, never benchmarked as
 * the product).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.
 *
 * Plain-C restatement of the spconv-side arithmetic of BtcDet's hot path.
 *
 * PARITY STATUS: *** parity unpinned ***  The algorithms restated here live in the third-party
 * dependency spconv v1.2.1 (commit fad3000249d27ca918f2655ff73c41f39b0f3127), pinned by the
 * reference only in prose (/root/reference/README.md:42, setup.py:41).  Its source is absent from
 * /root/reference, not installed, and the reference has no tests or golden vectors at this boundary
 * (SURVEY.md §4, §8c).  What is restated is spconv's published algorithm as SURVEY.md App. B
 * records it; what anchors it are (i) the reference's call sites cited per function and (ii)
 * dense-equivalence tests against torch.nn.functional.conv3d / conv_transpose3d / max_pool3d
 * (tests/test_oracle_dense_equiv.py).
 *
 * Conventions
 *   indices        (N,4) int32 rows [b,z,y,x]                     (spconv_backbone.py:150,952)
 *   kernel offset  k = (kz*KH + ky)*KW + kx, weight viewed as W[K][Cin][Cout]
 *                  (weight parameter layout [kD,kH,kW,Cin,Cout])
 *   orientation    PyTorch cross-correlation: in = out*s - p + k*d (conv), out = in*s - p + k*d (transpose)
 *   neighbour maps nbr_out (n_out,K): input row feeding output row i at offset k, or -1
 *                  nbr_in  (n_in ,K): output row fed by input row j at offset k, or -1
 *   canonical order: output rows ascending in (b,z,y,x) for conv/transpose/pool, input order for SubM
 *                  (SURVEY.md App. B.4).
 *
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off -shared -fPIC).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------ */
/* Voxelizer: spconv.utils.VoxelGeneratorV2.generate -> points_to_voxel_3d_np */
/* call sites: /root/reference/btcdet/datasets/processor/data_processor.py:63-73,107-117,136,   */
/*             160-170,177.                                                                    */
/* Sequential first-come semantics: voxel order = order of first appearance, within-voxel      */
/* order = input order, points beyond max_points and voxels beyond max_voxels are dropped,      */
/* padding is 0.  coords are written zyx.  Division and floor are float32.                      */
/* scratch: coor_to_voxelidx, grid[2]*grid[1]*grid[0] int32, must be all -1 on entry and is     */
/* restored to all -1 on exit.  Returns the number of voxels.                                  */
/* ------------------------------------------------------------------------------------------ */
int orc_voxelize(const float* pts, int n, int ld, int C, const float* range, const float* vsize,
                 const int* grid /* x,y,z */, int max_pts, int max_vox, float* voxels /* max_vox*max_pts*C, pre-zeroed */,
                 int* coords /* max_vox*3 zyx */, int* num /* max_vox, pre-zeroed */, int* scratch) {
  int voxel_num = 0;
  for (int i = 0; i < n; ++i) {
    const float* p = pts + (size_t)i * ld;
    int c[3];
    int failed = 0;
    for (int j = 0; j < 3; ++j) {
      float q = (p[j] - range[j]) / vsize[j];
      int cj = (int)floorf(q);
      if (cj < 0 || cj >= grid[j]) { failed = 1; break; }
      c[j] = cj;
    }
    if (failed) continue;
    size_t lin = ((size_t)c[2] * grid[1] + c[1]) * grid[0] + c[0];
    int vid = scratch[lin];
    if (vid == -1) {
      if (voxel_num >= max_vox) continue;
      vid = voxel_num++;
      scratch[lin] = vid;
      coords[vid * 3 + 0] = c[2];
      coords[vid * 3 + 1] = c[1];
      coords[vid * 3 + 2] = c[0];
    }
    int m = num[vid];
    if (m < max_pts) {
      memcpy(voxels + ((size_t)vid * max_pts + m) * C, p, sizeof(float) * C);
      num[vid] = m + 1;
    }
  }
  for (int v = 0; v < voxel_num; ++v) {
    size_t lin = ((size_t)coords[v * 3] * grid[1] + coords[v * 3 + 1]) * grid[0] + coords[v * 3 + 2];
    scratch[lin] = -1;
  }
  return voxel_num;
}

/* ------------------------------------------------------------------------------------------ */
/* Rulebook: spconv ops.get_indice_pairs; call sites are every sparse    */
/* layer of /root/reference/btcdet/models/backbones_3d/spconv_backbone.py:106-128,656-700,      */
/* 732-767,831-847 and occ_head_3D.py:25-31.                                                   */
/* mode 0 = SubM, 1 = regular conv / max-pool geometry, 2 = transposed conv.                   */
/* ------------------------------------------------------------------------------------------ */
static int cmp_i64(const void* a, const void* b) {
  int64_t x = *(const int64_t*)a, y = *(const int64_t*)b;
  return (x > y) - (x < y);
}

void orc_out_shape(const int* in_shape, const int* k, const int* s, const int* p, const int* d,
                   const int* outpad, int mode, int* out_shape) {
  for (int j = 0; j < 3; ++j) {
    if (mode == 0) out_shape[j] = in_shape[j];
    else if (mode == 1) out_shape[j] = (in_shape[j] + 2 * p[j] - d[j] * (k[j] - 1) - 1) / s[j] + 1;
    else out_shape[j] = (in_shape[j] - 1) * s[j] - 2 * p[j] + k[j] + outpad[j];
  }
}

/* returns n_out; out_indices capacity must be >= n*K rows (worst case), nbr_out >= n*K*K ints is
 * never needed: n_out <= n*K, caller passes cap_out rows for out_indices / nbr_out. */
int orc_rulebook(const int* indices, int n, const int* in_shape, const int* out_shape, const int* k,
                 const int* s, const int* p, const int* d, int mode, int cap_out, int* out_indices,
                 int* nbr_out, int* nbr_in) {
  const int K = k[0] * k[1] * k[2];
  const int64_t ovol = (int64_t)out_shape[0] * out_shape[1] * out_shape[2];
  for (size_t t = 0; t < (size_t)n * K; ++t) nbr_in[t] = -1;

  if (mode == 0) {
    /* SubM: outputs = inputs in input order; pair iff indices[j] == indices[i] + (off - centre)*d */
    if (n > cap_out) return -1;
    int64_t* keys = (int64_t*)malloc(sizeof(int64_t) * 2 * (size_t)(n > 0 ? n : 1));
    for (int i = 0; i < n; ++i) {
      const int* c = indices + 4 * i;
      keys[2 * i] = (((int64_t)c[0] * in_shape[0] + c[1]) * in_shape[1] + c[2]) * in_shape[2] + c[3];
      keys[2 * i + 1] = i;
      memcpy(out_indices + 4 * i, c, 4 * sizeof(int));
    }
    qsort(keys, n, 2 * sizeof(int64_t), cmp_i64);
    for (size_t t = 0; t < (size_t)n * K; ++t) nbr_out[t] = -1;
    for (int i = 0; i < n; ++i) {
      const int* c = indices + 4 * i;
      for (int kz = 0; kz < k[0]; ++kz)
        for (int ky = 0; ky < k[1]; ++ky)
          for (int kx = 0; kx < k[2]; ++kx) {
            int kk = (kz * k[1] + ky) * k[2] + kx;
            int z = c[1] + (kz - k[0] / 2) * d[0];
            int y = c[2] + (ky - k[1] / 2) * d[1];
            int x = c[3] + (kx - k[2] / 2) * d[2];
            if (z < 0 || z >= in_shape[0] || y < 0 || y >= in_shape[1] || x < 0 || x >= in_shape[2]) continue;
            int64_t key = (((int64_t)c[0] * in_shape[0] + z) * in_shape[1] + y) * in_shape[2] + x;
            int lo = 0, hi = n;
            while (lo < hi) { int mid = (lo + hi) / 2; if (keys[2 * mid] < key) lo = mid + 1; else hi = mid; }
            if (lo < n && keys[2 * lo] == key) {
              int j = (int)keys[2 * lo + 1];
              nbr_out[(size_t)i * K + kk] = j;   /* output i gathers input j at offset kk   */
              nbr_in[(size_t)j * K + kk] = i;    /* input j scatters to output i at offset kk */
            }
          }
    }
    free(keys);
    return n;
  }

  /* regular / transposed: enumerate candidate outputs, sort-unique */
  int64_t* cand = (int64_t*)malloc(sizeof(int64_t) * (size_t)(n > 0 ? n : 1) * K);
  size_t nc = 0;
  for (int i = 0; i < n; ++i) {
    const int* c = indices + 4 * i;
    for (int kz = 0; kz < k[0]; ++kz)
      for (int ky = 0; ky < k[1]; ++ky)
        for (int kx = 0; kx < k[2]; ++kx) {
          int kv[3] = {kz, ky, kx};
          int o[3];
          int ok = 1;
          for (int j = 0; j < 3; ++j) {
            if (mode == 1) {
              int t = c[1 + j] + p[j] - kv[j] * d[j];
              if (t < 0 || t % s[j] != 0) { ok = 0; break; }
              o[j] = t / s[j];
            } else {
              o[j] = c[1 + j] * s[j] - p[j] + kv[j] * d[j];
            }
            if (o[j] < 0 || o[j] >= out_shape[j]) { ok = 0; break; }
          }
          if (!ok) continue;
          cand[nc++] = (int64_t)c[0] * ovol + ((int64_t)o[0] * out_shape[1] + o[1]) * out_shape[2] + o[2];
        }
  }
  int64_t* uniq = (int64_t*)malloc(sizeof(int64_t) * (nc > 0 ? nc : 1));
  memcpy(uniq, cand, sizeof(int64_t) * nc);
  qsort(uniq, nc, sizeof(int64_t), cmp_i64);
  int n_out = 0;
  for (size_t t = 0; t < nc; ++t)
    if (t == 0 || uniq[t] != uniq[t - 1]) uniq[n_out++] = uniq[t];
  if (n_out > cap_out) { free(cand); free(uniq); return -1; }
  for (int r = 0; r < n_out; ++r) {
    int64_t key = uniq[r];
    int b = (int)(key / ovol);
    int64_t rem = key % ovol;
    out_indices[4 * r + 0] = b;
    out_indices[4 * r + 1] = (int)(rem / ((int64_t)out_shape[1] * out_shape[2]));
    out_indices[4 * r + 2] = (int)((rem / out_shape[2]) % out_shape[1]);
    out_indices[4 * r + 3] = (int)(rem % out_shape[2]);
  }
  for (size_t t = 0; t < (size_t)n_out * K; ++t) nbr_out[t] = -1;
  /* second pass: fill maps */
  for (int i = 0; i < n; ++i) {
    const int* c = indices + 4 * i;
    for (int kz = 0; kz < k[0]; ++kz)
      for (int ky = 0; ky < k[1]; ++ky)
        for (int kx = 0; kx < k[2]; ++kx) {
          int kv[3] = {kz, ky, kx};
          int kk = (kz * k[1] + ky) * k[2] + kx;
          int o[3];
          int ok = 1;
          for (int j = 0; j < 3; ++j) {
            if (mode == 1) {
              int t = c[1 + j] + p[j] - kv[j] * d[j];
              if (t < 0 || t % s[j] != 0) { ok = 0; break; }
              o[j] = t / s[j];
            } else {
              o[j] = c[1 + j] * s[j] - p[j] + kv[j] * d[j];
            }
            if (o[j] < 0 || o[j] >= out_shape[j]) { ok = 0; break; }
          }
          if (!ok) continue;
          int64_t key = (int64_t)c[0] * ovol + ((int64_t)o[0] * out_shape[1] + o[1]) * out_shape[2] + o[2];
          int lo = 0, hi = n_out;
          while (lo < hi) { int mid = (lo + hi) / 2; if (uniq[mid] < key) lo = mid + 1; else hi = mid; }
          nbr_out[(size_t)lo * K + kk] = i;
          nbr_in[(size_t)i * K + kk] = lo;
        }
  }
  free(cand);
  free(uniq);
  return n_out;
}

/* ------------------------------------------------------------------------------------------ */
/* Sparse conv apply (spconv indice_conv / indice_subm_conv / indice_inverse_conv, SURVEY §3.4, */
/* App. B.6).  Output-stationary restatement with a FIXED summation order: for each output row, */
/* offsets k ascending, input channels ascending, one fused multiply-add per term starting from */
/* 0, bias added last.  (spconv's scatter-add order is unspecified; any order is within fp32     */
/* roundoff of this one.)                                                                      */
/* ------------------------------------------------------------------------------------------ */
void orc_conv_fwd(const float* feat, const float* W, const float* bias, const int* nbr_out, int n_out,
                  int K, int Cin, int Cout, float* out) {
  for (int i = 0; i < n_out; ++i) {
    float* o = out + (size_t)i * Cout;
    for (int co = 0; co < Cout; ++co) o[co] = 0.0f;
    for (int k = 0; k < K; ++k) {
      int j = nbr_out[(size_t)i * K + k];
      if (j < 0) continue;
      const float* f = feat + (size_t)j * Cin;
      const float* w = W + (size_t)k * Cin * Cout;
      for (int ci = 0; ci < Cin; ++ci) {
        float a = f[ci];
        for (int co = 0; co < Cout; ++co) o[co] = fmaf(a, w[(size_t)ci * Cout + co], o[co]);
      }
    }
    if (bias) for (int co = 0; co < Cout; ++co) o[co] += bias[co];
  }
}

/* dgrad: dIn[j] = sum_k dOut[nbr_in[j][k]] @ W[k]^T  (k ascending, co ascending, fmaf chain) */
void orc_conv_dgrad(const float* dout, const float* W, const int* nbr_in, int n_in, int K, int Cin,
                    int Cout, float* din) {
  for (int j = 0; j < n_in; ++j) {
    float* g = din + (size_t)j * Cin;
    for (int ci = 0; ci < Cin; ++ci) g[ci] = 0.0f;
    for (int k = 0; k < K; ++k) {
      int i = nbr_in[(size_t)j * K + k];
      if (i < 0) continue;
      const float* dy = dout + (size_t)i * Cout;
      const float* w = W + (size_t)k * Cin * Cout;
      for (int co = 0; co < Cout; ++co) {
        float a = dy[co];
        for (int ci = 0; ci < Cin; ++ci) g[ci] = fmaf(a, w[(size_t)ci * Cout + co], g[ci]);
      }
    }
  }
}

/* wgrad: dW[k][ci][co] = sum_i feat[nbr_out[i][k]][ci] * dOut[i][co], accumulated in double  */
void orc_conv_wgrad(const float* feat, const float* dout, const int* nbr_out, int n_out, int K, int Cin,
                    int Cout, float* dW) {
  double* acc = (double*)calloc((size_t)K * Cin * Cout, sizeof(double));
  for (int i = 0; i < n_out; ++i) {
    const float* dy = dout + (size_t)i * Cout;
    for (int k = 0; k < K; ++k) {
      int j = nbr_out[(size_t)i * K + k];
      if (j < 0) continue;
      const float* f = feat + (size_t)j * Cin;
      double* a = acc + (size_t)k * Cin * Cout;
      for (int ci = 0; ci < Cin; ++ci)
        for (int co = 0; co < Cout; ++co) a[(size_t)ci * Cout + co] += (double)f[ci] * (double)dy[co];
    }
  }
  for (size_t t = 0; t < (size_t)K * Cin * Cout; ++t) dW[t] = (float)acc[t];
  free(acc);
}

/* ------------------------------------------------------------------------------------------ */
/* Sparse max-pool (spconv indice_maxpool): out starts at 0, out = max(out,in) */
/* over pairs; backward routes dOut to every input equal to the output.                        */
/* call site: spconv_backbone.py:29,831-847.                                                   */
/* ------------------------------------------------------------------------------------------ */
void orc_maxpool_fwd(const float* feat, const int* nbr_out, int n_out, int K, int C, float* out) {
  for (int i = 0; i < n_out; ++i)
    for (int c = 0; c < C; ++c) {
      float m = 0.0f;
      for (int k = 0; k < K; ++k) {
        int j = nbr_out[(size_t)i * K + k];
        if (j < 0) continue;
        float v = feat[(size_t)j * C + c];
        if (v > m) m = v;
      }
      out[(size_t)i * C + c] = m;
    }
}

void orc_maxpool_bwd(const float* feat, const float* out, const float* dout, const int* nbr_in, int n_in,
                     int K, int C, float* din) {
  for (int j = 0; j < n_in; ++j)
    for (int c = 0; c < C; ++c) {
      float g = 0.0f;
      float v = feat[(size_t)j * C + c];
      for (int k = 0; k < K; ++k) {
        int i = nbr_in[(size_t)j * K + k];
        if (i < 0) continue;
        if (out[(size_t)i * C + c] == v) g += dout[(size_t)i * C + c];
      }
      din[(size_t)j * C + c] = g;
    }
}

/* ------------------------------------------------------------------------------------------ */
/* SparseConvTensor.dense(): scatter rows into zeros[B,C,D,H,W]               */
/* call sites: height_compression.py:21, occ_head_3D.py:46,51, spconv_backbone.py:890,930.      */
/* ------------------------------------------------------------------------------------------ */
void orc_dense(const float* feat, const int* indices, int n, int C, int B, const int* shape, float* out) {
  size_t vol = (size_t)shape[0] * shape[1] * shape[2];
  memset(out, 0, sizeof(float) * (size_t)B * C * vol);
  for (int i = 0; i < n; ++i) {
    const int* c = indices + 4 * i;
    size_t sp = ((size_t)c[1] * shape[1] + c[2]) * shape[2] + c[3];
    for (int ch = 0; ch < C; ++ch) out[((size_t)c[0] * C + ch) * vol + sp] = feat[(size_t)i * C + ch];
  }
}

/* ------------------------------------------------------------------------------------------ */
/* Rotated BEV overlap / IoU and NMS.  Restates, in float32 like the      */
/* reference's kernels, /root/reference/btcdet/ops/iou3d_nms/src/iou3d_nms_kernel.cu:           */
/*   box_overlap :107-233 (edge intersections :66-98 + corners inside the other box :55-64,     */
/*   ordered by atan2 about their centroid :101-103 with a bubble sort :198-208, fan area),     */
/*   iou_bev :235-243, nms_kernel :268-309 + the host reduction iou3d_nms.cpp:104-136,          */
/*   iou_normal / nms_normal_kernel :312-362.                                                   */
/* PARITY UNPINNED: the reference's own CPU twin (iou3d_cpu.cpp) includes cuda.h and torch      */
/* headers and cannot be built here, and the reference holds no test vectors for these ops;     */
/* tests/test_oracle_iou3d.py pins this restatement against an independent float64 polygon      */
/* clipper and closed-form cases instead.  boxes are (N,7) [x, y, z, dx, dy, dz, heading].      */
/* ------------------------------------------------------------------------------------------ */
typedef struct { float x, y; } orc_pt;

static float orc_cross3(orc_pt p1, orc_pt p2, orc_pt p0) { return (p1.x - p0.x) * (p2.y - p0.y) - (p2.x - p0.x) * (p1.y - p0.y); }

static int orc_in_box2d(const float* box, orc_pt p) {
  const float margin = 1e-2f;
  float c = cosf(-box[6]), s = sinf(-box[6]);
  float rx = (p.x - box[0]) * c + (p.y - box[1]) * (-s);
  float ry = (p.x - box[0]) * s + (p.y - box[1]) * c;
  return fabsf(rx) < box[3] / 2 + margin && fabsf(ry) < box[4] / 2 + margin;
}

static int orc_seg_intersection(orc_pt p1, orc_pt p0, orc_pt q1, orc_pt q0, orc_pt* ans) {
  if (!(fminf(p0.x, p1.x) <= fmaxf(q0.x, q1.x) && fminf(q0.x, q1.x) <= fmaxf(p0.x, p1.x) &&
        fminf(p0.y, p1.y) <= fmaxf(q0.y, q1.y) && fminf(q0.y, q1.y) <= fmaxf(p0.y, p1.y)))
    return 0;
  float s1 = orc_cross3(q0, p1, p0), s2 = orc_cross3(p1, q1, p0), s3 = orc_cross3(p0, q1, q0), s4 = orc_cross3(q1, p1, q0);
  if (!(s1 * s2 > 0 && s3 * s4 > 0)) return 0;
  float s5 = orc_cross3(q1, p1, p0);
  if (fabsf(s5 - s1) > 1e-8f) {
    ans->x = (s5 * q0.x - s1 * q1.x) / (s5 - s1);
    ans->y = (s5 * q0.y - s1 * q1.y) / (s5 - s1);
  } else {
    float a0 = p0.y - p1.y, b0 = p1.x - p0.x, c0 = p0.x * p1.y - p1.x * p0.y;
    float a1 = q0.y - q1.y, b1 = q1.x - q0.x, c1 = q0.x * q1.y - q1.x * q0.y;
    float D = a0 * b1 - a1 * b0;
    ans->x = (b0 * c1 - b1 * c0) / D;
    ans->y = (a1 * c0 - a0 * c1) / D;
  }
  return 1;
}

static void orc_corners(const float* box, orc_pt* c /* 5 */) {
  float hx = box[3] / 2, hy = box[4] / 2, co = cosf(box[6]), si = sinf(box[6]);
  const float lx[4] = {box[0] - hx, box[0] + hx, box[0] + hx, box[0] - hx};
  const float ly[4] = {box[1] - hy, box[1] - hy, box[1] + hy, box[1] + hy};
  for (int k = 0; k < 4; ++k) {
    c[k].x = (lx[k] - box[0]) * co + (ly[k] - box[1]) * (-si) + box[0];
    c[k].y = (lx[k] - box[0]) * si + (ly[k] - box[1]) * co + box[1];
  }
  c[4] = c[0];
}

float orc_box_overlap_bev(const float* a, const float* b) {
  orc_pt ca[5], cb[5], pts[16], ctr = {0.f, 0.f};
  orc_corners(a, ca);
  orc_corners(b, cb);
  int cnt = 0;
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j)
      if (orc_seg_intersection(ca[i + 1], ca[i], cb[j + 1], cb[j], &pts[cnt])) {
        ctr.x += pts[cnt].x; ctr.y += pts[cnt].y;
        ++cnt;
      }
  for (int k = 0; k < 4; ++k) {
    if (orc_in_box2d(a, cb[k])) { ctr.x += cb[k].x; ctr.y += cb[k].y; pts[cnt++] = cb[k]; }
    if (orc_in_box2d(b, ca[k])) { ctr.x += ca[k].x; ctr.y += ca[k].y; pts[cnt++] = ca[k]; }
  }
  ctr.x /= cnt; ctr.y /= cnt;  /* cnt == 0: the loops below do not run (as in the reference) */
  for (int j = 0; j < cnt - 1; ++j)
    for (int i = 0; i < cnt - j - 1; ++i)
      if (atan2f(pts[i].y - ctr.y, pts[i].x - ctr.x) > atan2f(pts[i + 1].y - ctr.y, pts[i + 1].x - ctr.x)) {
        orc_pt t = pts[i]; pts[i] = pts[i + 1]; pts[i + 1] = t;
      }
  float area = 0.f;
  for (int k = 0; k < cnt - 1; ++k) {
    float ax = pts[k].x - pts[0].x, ay = pts[k].y - pts[0].y, bx = pts[k + 1].x - pts[0].x, by = pts[k + 1].y - pts[0].y;
    area += ax * by - ay * bx;
  }
  return fabsf(area) / 2.0f;
}

float orc_iou_bev(const float* a, const float* b) {
  float sa = a[3] * a[4], sb = b[3] * b[4], so = orc_box_overlap_bev(a, b);
  return so / fmaxf(sa + sb - so, 1e-8f);
}

static float orc_iou_normal(const float* a, const float* b) {
  float left = fmaxf(a[0] - a[3] / 2, b[0] - b[3] / 2), right = fminf(a[0] + a[3] / 2, b[0] + b[3] / 2);
  float top = fmaxf(a[1] - a[4] / 2, b[1] - b[4] / 2), bottom = fminf(a[1] + a[4] / 2, b[1] + b[4] / 2);
  float w = fmaxf(right - left, 0.f), h = fmaxf(bottom - top, 0.f), inter = w * h;
  return inter / fmaxf(a[3] * a[4] + b[3] * b[4] - inter, 1e-8f);
}

/* mode 0: overlap area, 1: BEV IoU; out (na, nb) */
void orc_boxes_pairwise_bev(const float* boxes_a, int na, const float* boxes_b, int nb, int mode, float* out) {
  for (int i = 0; i < na; ++i)
    for (int j = 0; j < nb; ++j)
      out[(size_t)i * nb + j] = mode ? orc_iou_bev(boxes_a + 7 * i, boxes_b + 7 * j) : orc_box_overlap_bev(boxes_a + 7 * i, boxes_b + 7 * j);
}

/* greedy NMS over boxes already sorted by descending score: box j > i is suppressed by a kept box i when iou > thresh.    */
/* rotated = 0: axis-aligned IoU (nms_normal).  keep gets the kept positions in ascending order; returns their number.     */
int orc_nms(const float* boxes, int n, float thresh, int rotated, long long* keep) {
  unsigned char* removed = (unsigned char*)calloc((size_t)(n > 0 ? n : 1), 1);
  int nk = 0;
  for (int i = 0; i < n; ++i) {
    if (removed[i]) continue;
    keep[nk++] = i;
    for (int j = i + 1; j < n; ++j) {
      if (removed[j]) continue;
      float v = rotated ? orc_iou_bev(boxes + 7 * i, boxes + 7 * j) : orc_iou_normal(boxes + 7 * i, boxes + 7 * j);
      if (v > thresh) removed[j] = 1;
    }
  }
  free(removed);
  return nk;
}

/* ======================================================================================================================
 * pointnet2_stack: sequential restatement of the reference's CUDA kernels, one loop body per
 * CUDA thread.  PARITY: unpinned against an execution of the reference (the ops exist only as CUDA sources there and the
 * reference has no tests for them); anchored on the kernels cited per function and cross-checked against independent numpy
 * formulations (tests/test_oracle_pointnet2.py).
 * ====================================================================================================================== */
static int orc_scene_of(int pt, const int32_t* cnt, int B, const int32_t* other, int* start_other) {
  int bs = 0, acc = cnt[0];
  for (int k = 1; k < B; ++k) {
    if (pt < acc) break;
    acc += cnt[k];
    bs = k;
  }
  int s = 0;
  for (int k = 0; k < bs; ++k) s += other[k];
  *start_other = s;
  return bs;
}

/* ball_query_gpu.cu:15-66 / shell_query_gpu.cu:15-68 (inner_radius < 0: ball).  idx (M,nsample) must be zero on entry
 * (pointnet2_utils.py:32). */
void orc_ball_query(const float* new_xyz, const int32_t* new_cnt, const float* xyz, const int32_t* cnt, int B, int M,
                    float inner_radius, float outer_radius, int nsample, int32_t* idx) {
  const float outer2 = outer_radius * outer_radius, inner2 = inner_radius * inner_radius;
  for (int pt = 0; pt < M; ++pt) {
    int start;
    const int bs = orc_scene_of(pt, new_cnt, B, cnt, &start);
    const float* q = new_xyz + (size_t)pt * 3;
    const float* p = xyz + (size_t)start * 3;
    int32_t* out = idx + (size_t)pt * nsample;
    const int n = cnt[bs];
    int c = 0;
    for (int k = 0; k < n; ++k) {
      const float x = p[k * 3 + 0], y = p[k * 3 + 1], z = p[k * 3 + 2];
      const float d2 = (q[0] - x) * (q[0] - x) + (q[1] - y) * (q[1] - y) + (q[2] - z) * (q[2] - z);
      if ((inner_radius < 0.f || d2 >= inner2) && d2 < outer2) {
        if (c == 0)
          for (int l = 0; l < nsample; ++l) out[l] = k;
        out[c] = k;
        if (++c >= nsample) break;
      }
    }
    if (c == 0) out[0] = -1;
  }
}

/* group_points_gpu.cu:64-97 */
void orc_group_points(const float* features, const int32_t* f_cnt, const int32_t* idx, const int32_t* idx_cnt, int B, int M, int C,
                      int nsample, float* out) {
  for (int pt = 0; pt < M; ++pt) {
    int fstart;
    orc_scene_of(pt, idx_cnt, B, f_cnt, &fstart);
    for (int c = 0; c < C; ++c)
      for (int s = 0; s < nsample; ++s)
        out[((size_t)pt * C + c) * nsample + s] = features[(size_t)(fstart + idx[(size_t)pt * nsample + s]) * C + c];
  }
}

/* group_points_gpu.cu:14-46 (atomicAdd there: summation order unspecified; here ascending (pt, c, sample)) */
void orc_group_points_grad(const float* grad_out, const int32_t* idx, const int32_t* idx_cnt, const int32_t* f_cnt, int B, int M, int C,
                           int N, int nsample, float* grad_features) {
  for (size_t i = 0; i < (size_t)N * C; ++i) grad_features[i] = 0.f;
  for (int pt = 0; pt < M; ++pt) {
    int fstart;
    orc_scene_of(pt, idx_cnt, B, f_cnt, &fstart);
    for (int c = 0; c < C; ++c)
      for (int s = 0; s < nsample; ++s)
        grad_features[(size_t)(fstart + idx[(size_t)pt * nsample + s]) * C + c] += grad_out[((size_t)pt * C + c) * nsample + s];
  }
}

/* sampling_gpu.cu:16-142: block_size = opt_n_threads(n) threads, thread t scans k = t, t + T, ... keeping its first
 * strict maximum, then the pairwise tree (tid, tid + half) where the lower tid wins ties.  temp (B,n) is 1e10 on entry. */
void orc_furthest_point_sampling(const float* xyz, int B, int n, int m, float* temp, int32_t* idxs) {
  if (m <= 0) return;
  int T = 1;
  while (T * 2 <= n && T < 1024) T *= 2;
  float* dv = (float*)malloc(sizeof(float) * (size_t)T);
  int* di = (int*)malloc(sizeof(int) * (size_t)T);
  for (int b = 0; b < B; ++b) {
    const float* d = xyz + (size_t)b * n * 3;
    float* tp = temp + (size_t)b * n;
    int32_t* out = idxs + (size_t)b * m;
    int old = 0;
    out[0] = 0;
    for (int j = 1; j < m; ++j) {
      const float x1 = d[old * 3 + 0], y1 = d[old * 3 + 1], z1 = d[old * 3 + 2];
      for (int t = 0; t < T; ++t) {
        int besti = 0;
        float best = -1.f;
        for (int k = t; k < n; k += T) {
          const float x2 = d[k * 3 + 0], y2 = d[k * 3 + 1], z2 = d[k * 3 + 2];
          const float dd = (x2 - x1) * (x2 - x1) + (y2 - y1) * (y2 - y1) + (z2 - z1) * (z2 - z1);
          const float d2 = fminf(dd, tp[k]);
          tp[k] = d2;
          besti = d2 > best ? k : besti;
          best = d2 > best ? d2 : best;
        }
        dv[t] = best;
        di[t] = besti;
      }
      for (int half = T / 2; half >= 1; half /= 2)
        for (int t = 0; t < half; ++t) {
          const float v1 = dv[t], v2 = dv[t + half];
          const int i1 = di[t], i2 = di[t + half];
          dv[t] = v1 > v2 ? v1 : v2;  /* max(v1, v2) */
          di[t] = v2 > v1 ? i2 : i1;
        }
      old = di[0];
      out[j] = old;
    }
  }
  free(dv);
  free(di);
}

/* interpolate_gpu.cu:14-69 */
void orc_three_nn(const float* unknown, const int32_t* u_cnt, const float* known, const int32_t* k_cnt, int B, int N, float* dist2,
                  int32_t* idx) {
  for (int pt = 0; pt < N; ++pt) {
    int kstart;
    const int bs = orc_scene_of(pt, u_cnt, B, k_cnt, &kstart);
    const float* kn = known + (size_t)kstart * 3;
    const float ux = unknown[(size_t)pt * 3 + 0], uy = unknown[(size_t)pt * 3 + 1], uz = unknown[(size_t)pt * 3 + 2];
    double best1 = 1e40, best2 = 1e40, best3 = 1e40;
    int b1 = 0, b2 = 0, b3 = 0;
    for (int k = 0; k < k_cnt[bs]; ++k) {
      const float x = kn[k * 3 + 0], y = kn[k * 3 + 1], z = kn[k * 3 + 2];
      const float d = (ux - x) * (ux - x) + (uy - y) * (uy - y) + (uz - z) * (uz - z);
      if (d < best1) { best3 = best2; b3 = b2; best2 = best1; b2 = b1; best1 = d; b1 = k; }
      else if (d < best2) { best3 = best2; b3 = b2; best2 = d; b2 = k; }
      else if (d < best3) { best3 = d; b3 = k; }
    }
    dist2[(size_t)pt * 3 + 0] = (float)best1; dist2[(size_t)pt * 3 + 1] = (float)best2; dist2[(size_t)pt * 3 + 2] = (float)best3;
    idx[(size_t)pt * 3 + 0] = b1 + kstart; idx[(size_t)pt * 3 + 1] = b2 + kstart; idx[(size_t)pt * 3 + 2] = b3 + kstart;
  }
}

/* interpolate_gpu.cu:106-121 */
void orc_three_interpolate(const float* features, const int32_t* idx, const float* weight, int N, int C, float* out) {
  for (int pt = 0; pt < N; ++pt)
    for (int c = 0; c < C; ++c) {
      const int32_t* id = idx + (size_t)pt * 3;
      const float* w = weight + (size_t)pt * 3;
      out[(size_t)pt * C + c] = w[0] * features[(size_t)id[0] * C + c] + w[1] * features[(size_t)id[1] * C + c] + w[2] * features[(size_t)id[2] * C + c];
    }
}

/* interpolate_gpu.cu:141-158 (atomicAdd there) */
void orc_three_interpolate_grad(const float* grad_out, const int32_t* idx, const float* weight, int N, int C, int M, float* grad_features) {
  for (size_t i = 0; i < (size_t)M * C; ++i) grad_features[i] = 0.f;
  for (int pt = 0; pt < N; ++pt)
    for (int c = 0; c < C; ++c) {
      const int32_t* id = idx + (size_t)pt * 3;
      const float* w = weight + (size_t)pt * 3;
      const float g = grad_out[(size_t)pt * C + c];
      grad_features[(size_t)id[0] * C + c] += g * w[0];
      grad_features[(size_t)id[1] * C + c] += g * w[1];
      grad_features[(size_t)id[2] * C + c] += g * w[2];
    }
}
