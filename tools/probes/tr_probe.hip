// What does ds_read_b64_tr_b16 return?  (csrc/conv_wgrad_x.hip relies on: within a 16-lane group, lane t supplies the address of 4
// contiguous 16-bit elements = row t / 4, 8-byte chunk t % 4 of a 4-row x 16-column block, and comes back with COLUMN t: rows 0..3.)
// hipcc --offload-arch=gfx950 tools/probes/tr_probe.hip -o /tmp/tr_probe && /tmp/tr_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef short s16x4 __attribute__((ext_vector_type(4)));
constexpr int RS = 48;   // bytes per image row (16 columns + pad), as the kernel's RS for 16 channels
__global__ void k(short* out) {
  __shared__ __attribute__((aligned(16))) char lds[64 * RS];
  for (int i = threadIdx.x; i < 64 * 16; i += 64) *(short*)(lds + (i / 16) * RS + (i % 16) * 2) = (short)((i / 16) * 100 + (i % 16));   // row * 100 + col
  __syncthreads();
  const int lane = threadIdx.x, g = lane >> 4, t = lane & 15;
  const char* p = lds + (4 * g + (t >> 2)) * RS + (t & 3) * 8;
  s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p);
  for (int j = 0; j < 4; ++j) out[lane * 4 + j] = v[j];
}
int main() {
  short* d; short h[256];
  hipMalloc(&d, sizeof(h));
  k<<<1, 64>>>(d);
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  int bad = 0;
  for (int l = 0; l < 64; ++l)
    for (int j = 0; j < 4; ++j) bad += h[l * 4 + j] != (short)((4 * (l >> 4) + j) * 100 + (l & 15));
  printf("tr_probe: %d of 256 elements differ from [row 4 g + j][col t]\n", bad);
  if (bad)
    for (int l = 0; l < 64; ++l) printf("lane %2d: %5d %5d %5d %5d\n", l, h[l * 4], h[l * 4 + 1], h[l * 4 + 2], h[l * 4 + 3]);
  return bad != 0;
}
