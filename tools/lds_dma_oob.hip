// Does an out-of-range lane of `buffer_load_dwordx4 ... lds` write ZEROS into the LDS (raw buffer, stride 0: a lane whose offset + 16
// exceeds num_records)?  The apply kernels rely on it for absent neighbours (conv_apply_split.hip `issue`): no L2 traffic, no stale bytes.
//   hipcc --offload-arch=gfx950 -O3 tools/lds_dma_oob.hip -o /tmp/lds_dma_oob && /tmp/lds_dma_oob
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void k(const float* feat, float* out) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* s = (float*)smem;
  for (int i = threadIdx.x; i < 256; i += 64) s[i] = -7.f;   // garbage that must be overwritten
  __syncthreads();
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)feat, 0, 0xFFFFFF00, 0x00020000);
  const unsigned vo = (threadIdx.x % 3 == 1) ? 0xFFFFFFF0u : (unsigned)(threadIdx.x * 16);
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)smem, 16, vo, 0, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = threadIdx.x; i < 256; i += 64) out[i] = s[i];
}

int main() {
  float *f, *o;
  hipMalloc(&f, 1024); hipMalloc(&o, 1024);
  std::vector<float> h(256), g(256);
  for (int i = 0; i < 256; ++i) h[i] = 1.f + i;
  hipMemcpy(f, h.data(), 1024, hipMemcpyHostToDevice);
  k<<<1, 64, 1024>>>(f, o);
  if (hipMemcpy(g.data(), o, 1024, hipMemcpyDeviceToHost) != hipSuccess) { printf("launch failed\n"); return 2; }
  int bad = 0;
  for (int l = 0; l < 64; ++l)
    for (int j = 0; j < 4; ++j) {
      const float want = (l % 3 == 1) ? 0.f : h[l * 4 + j];
      if (g[l * 4 + j] != want) { if (bad < 8) printf("lane %d word %d: %g, want %g\n", l, j, g[l * 4 + j], want); ++bad; }
    }
  printf("out-of-range lanes of buffer_load ... lds: %s (%d mismatches)\n", bad ? "DO NOT write zeros" : "write zeros", bad);
  return bad ? 1 : 0;
}
