// Voxel feature encoders as single launches (the torch formulation is ~20 small ops per call; the forward pass is bound
// by the host's launch rate).  Replaces the arithmetic of
//   /root/reference/btcdet/models/backbones_3d/vfe/mean_vfe.py:27-38   (MeanVFE, maxprob = False: sum over slots / max(count, 1))
//   /root/reference/btcdet/models/backbones_3d/vfe/occ_vfe.py:24-55    (OccVFE: slots with code < 0.05 are raw points, the others
//     occupancy points; mean of the raw channels over raw slots, or over occupancy slots for voxels that hold only occupancy
//     points; max of the code channels over ALL slots, padding included)
// Sums run over the slots in order (the reference's torch.sum over 5 / 12 slots does the same).
#include "btc_common.h"

namespace {

// voxels (M, P, C) -> out (M, C): sum over all P slots (padding slots are zero) / max(count, 1)
__global__ __launch_bounds__(256) void mean_vfe_k(const float* __restrict__ vox, const float* __restrict__ num_f, const int32_t* __restrict__ num_i,
                                                  int M, int P, int C, float* __restrict__ out) {
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (long long)M * C) return;
  const int m = (int)(t / C), c = (int)(t % C);
  float s = 0.f;
  for (int p = 0; p < P; ++p) s += vox[((size_t)m * P + p) * C + c];
  const float n = num_f ? num_f[m] : (float)num_i[m];
  out[t] = s / fmaxf(n, 1.0f);
}

// voxels (M, P, F), F = R raw channels + (F - R) code channels; feat (M, F), occ (M, F - R)
__global__ __launch_bounds__(256) void occ_vfe_k(const float* __restrict__ vox, const float* __restrict__ num_f, const int32_t* __restrict__ num_i,
                                                 int M, int P, int F, int R, float* __restrict__ feat, float* __restrict__ occ) {
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (long long)M * F) return;
  const int m = (int)(t / F), c = (int)(t % F);
  const int n = num_f ? (int)num_f[m] : num_i[m];
  const float* v = vox + (size_t)m * P * F;
  if (c >= R) {  // code channel: max over all slots
    float mx = v[c];
    for (int p = 1; p < P; ++p) mx = fmaxf(mx, v[(size_t)p * F + c]);
    feat[t] = mx;
    occ[(size_t)m * (F - R) + (c - R)] = mx;
    return;
  }
  float raw = 0.f, oc = 0.f;
  int raw_n = 0, occ_n = 0;
  for (int p = 0; p < P; ++p) {
    const bool valid = p < n;
    const bool is_occ = v[(size_t)p * F + F - 1] >= 0.05f;
    const float x = v[(size_t)p * F + c];
    if (valid && !is_occ) { raw += x; ++raw_n; }
    if (valid && is_occ) { oc += x; ++occ_n; }
  }
  const float rf = raw / fmaxf((float)raw_n, 1.0f), of = oc / fmaxf((float)occ_n, 1.0f);
  feat[t] = rf + ((occ_n > 0 && raw_n == 0) ? of : 0.f);
}

}  // namespace

extern "C" int btc_mean_vfe(const float* voxels, const void* num_points, int num_is_float, int M, int P, int C, float* out, void* stream) {
  BTC_CHECK_ARG(M >= 0 && P >= 1 && C >= 1, "btc_mean_vfe: bad sizes");
  if (M == 0) return BTC_OK;
  mean_vfe_k<<<btc_cdiv((long long)M * C, 256), 256, 0, (hipStream_t)stream>>>(voxels, num_is_float ? (const float*)num_points : nullptr,
                                                                             num_is_float ? nullptr : (const int32_t*)num_points, M, P, C, out);
  BTC_LAUNCH_CHECK();
  return BTC_OK;
}

extern "C" int btc_occ_vfe(const float* voxels, const void* num_points, int num_is_float, int M, int P, int F, int R, float* feat, float* occ,
                           void* stream) {
  BTC_CHECK_ARG(M >= 0 && P >= 1 && R >= 1 && F > R, "btc_occ_vfe: bad sizes");
  if (M == 0) return BTC_OK;
  occ_vfe_k<<<btc_cdiv((long long)M * F, 256), 256, 0, (hipStream_t)stream>>>(voxels, num_is_float ? (const float*)num_points : nullptr,
                                                                            num_is_float ? nullptr : (const int32_t*)num_points, M, P, F, R, feat,
                                                                            occ);
  BTC_LAUNCH_CHECK();
  return BTC_OK;
}
