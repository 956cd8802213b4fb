// Shared host/device helpers for libbtcdet_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/btcdet_hip.h"

#define BTC_EMPTY_KEY (-1)            // hash key sentinel (memset 0xFF)
#define BTC_EMPTY_IDX 0x7F7F7F7F      // point-index sentinel (memset 0x7F), larger than any row index

void btc_set_error(const char* fmt, ...);

#define BTC_CHECK_ARG(cond, ...)            \
  do {                                      \
    if (!(cond)) {                          \
      btc_set_error(__VA_ARGS__);           \
      return BTC_EINVAL;                    \
    }                                       \
  } while (0)

#define BTC_HIP(call)                                                             \
  do {                                                                            \
    hipError_t e_ = (call);                                                       \
    if (e_ != hipSuccess) {                                                       \
      btc_set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
      return BTC_ELAUNCH;                                                         \
    }                                                                             \
  } while (0)

#define BTC_LAUNCH_CHECK()                                                        \
  do {                                                                            \
    hipError_t e_ = hipGetLastError();                                            \
    if (e_ != hipSuccess) {                                                       \
      btc_set_error("kernel launch failed: %s (%s:%d)", hipGetErrorString(e_), __FILE__, __LINE__); \
      return BTC_ELAUNCH;                                                         \
    }                                                                             \
  } while (0)

// ---- last-arriver reductions: ONE statement of the protocol (bn.hip, bn_fuse.h, rulebook.hip rb_scan, occ_loss.hip, glue.hip sumsq2) ----
// producers : publish their partial results with agent-scope relaxed stores / atomics (btc_st_agent, unsafeAtomicAdd: `sc1` operations,
//             performed at the device's coherence point, never left dirty in one XCD's L2), then btc_ticket_take(): drain the wave's
//             vector-memory queue (s_waitcnt vmcnt(0)) and take a relaxed agent-scope ticket;
// consumer  : the workgroup that draws the last ticket calls btc_ticket_acquire() (agent-scope acquire fence) and reads every partial
//             with btc_ld_agent (agent-scope atomic load: never served from a stale line of its own XCD's L2).
// There is deliberately NO agent-scope release fence on the producer side: on gfx950 it is `buffer_wbl2 sc1`, a write-back of the XCD
// L2's dirty lines -- every activation the previous kernels wrote -- once per workgroup (2-6 us each).
// That the payload is visible once vmcnt has drained rests on the sc1 write-through behaviour MI355X_MICROARCH.md documents for this
// target ("handoff-flag: sc1 payload -> vmcnt(0) -> sc1 flag"), not on the HSA memory model -- hence the guard: another target must
// re-derive it (or put the release fence back).  tests/test_hip_core.py::test_last_arriver_reductions_stress repeats every such
// reduction on grids that span all eight XCDs, behind kernels that leave the L2s dirty, against float64 sums made by torch.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "the fence-free last-arriver protocol (btc_ticket_take) is derived for gfx950's sc1 write-through stores only"
#endif
#if defined(__HIPCC__)
template <class T>
__device__ __forceinline__ void btc_st_agent(T* p, T v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
template <class T>
__device__ __forceinline__ T btc_ld_agent(const T* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// the calling thread's ticket (0 .. total-1)
__device__ __forceinline__ int btc_ticket_take(int32_t* counter) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  return __hip_atomic_fetch_add(counter, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void btc_ticket_acquire() { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); }
#endif

// tuning overrides (btc_tune_set): 0 = built-in policy
#define BTC_TUNE_KEYS 24
int btc_tune_get(int key);

// conv_apply_glds.hip: LDS-DMA pipelined sparse-conv apply (same results as conv_apply)
bool btc_apply_glds_supported(int K, int Cred, int Cres);
bool btc_apply_glds_has_shape(int shape, bool bf = false);
size_t btc_apply_glds_lds_bytes(int shape, int kc, int K, bool bf, int stages = 3);
int btc_apply_glds_stages(int shape, int kc, int K, bool bf, int asked);
// xcd: kernel flags -- bit 0 = XCD-contiguous tile mapping, bit 1 = `nbr` is a submanifold layer's forward map read as its backward
// map (column K-1-k holds offset k: the two maps are mirror images, rulebook.hip)
int btc_launch_apply_glds(bool trans_w, int shape, int kc, int xcd, bool bf, const void* feat, const float* W, const float* bias,
                          const int32_t* nbr, const int32_t* order /* row order hint or NULL */, int n_rows, int K, int Cred, int Cres, void* out,
                          hipStream_t stream, const struct BnFuse* bn = nullptr /* bn_fuse.h: batch statistics of the result in the epilogue */);

// sparse_conv.hip: forward conv + batch statistics of its result in the epilogue (bn_fuse.h), fp32 weights
int btc_conv_fwd_stats(int operands, const void* src, long long src_rows, const float* W, const float* bias, const int32_t* nbr, const int32_t* order,
                       int n_rows, int K, int Cin, int Cout, void* dst, const struct BnFuse& bn, hipStream_t stream, int* fused);

// conv_apply_bf16.hip: bf16 operands on the bf16 matrix pipe; Wq[k][Cres][Cred] bf16
int btc_apply_bf16w(const void* src, const void* Wq, const float* bias, const int32_t* nbr, const int32_t* order, int n_rows, int K, int Cred,
                    int Cres, void* dst, hipStream_t stream, int mirror = 0, const struct BnFuse* bn = nullptr);
// conv_wgrad_x.hip: weight gradient on the bf16 matrix pipe (mode 0: bf16 activations, 1: fp32 activations as three exact bf16 pieces);
// cg / cc = channels of the gathered / contiguous operand of the row walk
bool btc_wgrad_x_supported(int mode, int K, int cg, int cc);
int btc_wgrad_x_plan(int mode, int rows, int K, int cg, int cc, int* S, int* ph, int* z = nullptr);   // -> offset groups; *S = slabs, *z = channel blocks
int btc_launch_wgrad_x(int mode, const void* g, const void* c, const int32_t* map, const int32_t* ord, int rows, int K, int cg, int cc, float* part,
                       int swap, hipStream_t stream);
// conv_wgrad_n.hip: weight gradient of a layer with a narrow result side (<= 8 channels), walked over the layer's INPUT rows: x read once,
// dy gathered through the backward map (mirror: a submanifold layer's forward map, column k' = offset K-1-k'); fp32 matrix pipe
// ... or with a narrow input (<= 8 channels: the first layers), walked over its OUTPUT rows with the features gathered through nbr_out
int btc_wgrad_n_kind(int K, int Cin, int Cout);   // 1 narrow result, 2 narrow input, 0 neither
bool btc_wgrad_n_supported(int K, int Cin, int Cout);   // kind 1
int btc_wgrad_n_plan(int rows);   // -> slabs
int btc_launch_wgrad_n(bool bf, const void* walked, const void* gathered, const int32_t* map, int rows, int K, int Cw, int Cn, float* part,
                       int flags /* 1 mirrored map, 2 narrow input (slab written [k][narrow][walked]) */, hipStream_t stream);
constexpr size_t BTC_SCRATCH_HEAD = 64 * 1024;              // head of a registered scratch buffer: zeroed at registration, zero between launches
constexpr long long BTC_SCRATCH_TICKETS = BTC_SCRATCH_HEAD / 4;   // (the z-split launches' per-tile tickets live there)
void* btc_scratch(hipStream_t stream, size_t* bytes);   // the stream's registered scratch buffer (btc_set_scratch) or NULL
struct BnFuse;
int btc_apply_split(const float* src, const void* Ws, const float* bias, const int32_t* nbr, const int32_t* order, int n_rows, int K, int Cred,
                    int Cres, float* dst, hipStream_t stream, int mirror, const BnFuse* bn);

// bfloat16 <-> fp32 (round to nearest even; NaN stays NaN)
__host__ __device__ __forceinline__ unsigned short btc_f32_to_bf16(float f) {
  union { float f; unsigned u; } c;
  c.f = f;
  unsigned u = c.u;
  if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (unsigned short)(u >> 16);
}
__host__ __device__ __forceinline__ float btc_bf16_to_f32(unsigned short h) {
  union { float f; unsigned u; } c;
  c.u = (unsigned)h << 16;
  return c.f;
}

// activation loads / stores: fp32, or bfloat16 widened to / rounded from fp32 (BF); `i` is an element index
template <bool BF>
__device__ __forceinline__ float4 btc_ld4(const float* base, size_t i) {
  if (!BF) return *reinterpret_cast<const float4*>(base + i);
  const uint2 v = *reinterpret_cast<const uint2*>(reinterpret_cast<const unsigned short*>(base) + i);
  return make_float4(__uint_as_float(v.x << 16), __uint_as_float(v.x & 0xffff0000u), __uint_as_float(v.y << 16),
                     __uint_as_float(v.y & 0xffff0000u));
}
template <bool BF>
__device__ __forceinline__ float btc_ld1(const float* base, size_t i) {
  if (!BF) return base[i];
  return __uint_as_float((unsigned)reinterpret_cast<const unsigned short*>(base)[i] << 16);
}
template <bool BF>
__device__ __forceinline__ void btc_st4(float* base, size_t i, float a, float b, float c, float d) {
  if (!BF) {
    *reinterpret_cast<float4*>(base + i) = make_float4(a, b, c, d);
  } else {
    uint2 v;
    v.x = (unsigned)btc_f32_to_bf16(a) | ((unsigned)btc_f32_to_bf16(b) << 16);
    v.y = (unsigned)btc_f32_to_bf16(c) | ((unsigned)btc_f32_to_bf16(d) << 16);
    *reinterpret_cast<uint2*>(reinterpret_cast<unsigned short*>(base) + i) = v;
  }
}
template <bool BF>
__device__ __forceinline__ void btc_st1(float* base, size_t i, float a) {
  if (!BF) base[i] = a;
  else reinterpret_cast<unsigned short*>(base)[i] = btc_f32_to_bf16(a);
}

// hipFuncSetAttribute (dynamic LDS size) is a per-DEVICE attribute: one flag per device the process touches, not one per process
// (a process that drives two GPUs would otherwise launch with the default 64 KB limit on the second one)
#ifdef __cplusplus
#include <mutex>
constexpr int BTC_MAX_DEVICES = 16;
struct BtcPerDeviceOnce {
  std::once_flag flag[BTC_MAX_DEVICES];
};
template <class F>
static inline void btc_once_per_device(BtcPerDeviceOnce& o, F&& fn) {
  int dev = 0;
  (void)hipGetDevice(&dev);
  std::call_once(o.flag[dev >= 0 && dev < BTC_MAX_DEVICES ? dev : 0], fn);
}
#endif

static inline int btc_cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }
static inline size_t btc_align(size_t x, size_t a = 256) { return (x + a - 1) / a * a; }
static inline unsigned btc_pow2_ge(unsigned long long v) {
  unsigned long long p = 1;
  while (p < v) p <<= 1;
  return (unsigned)p;
}

// Workspace carving: every sub-buffer 256-byte aligned.
struct BtcCarver {
  char* base;
  size_t off;
  explicit BtcCarver(void* p) : base((char*)p), off(0) {}
  template <typename T>
  T* take(size_t count) {
    T* r = (T*)(base + off);
    off += btc_align(count * sizeof(T));
    return r;
  }
};

// Device-wide exclusive scan of int32 (scan.hip).  out may alias in.  n >= 0.
// ws: btc_scan_ws_bytes(n).  out[i] = sum_{j<i} in[j]; if total != nullptr, *total = sum of all.
size_t btc_scan_ws_bytes(long long n);
int btc_scan_exclusive_i32(const int32_t* in, int32_t* out, long long n, int32_t* total, void* ws, hipStream_t stream);

// byte map (one byte per cell, 0 / 1) -> bitmap + exclusive popcount prefix (scan.hip); see rulebook.hip
size_t btc_bytemap_bytes(long long nwords);
int btc_bytemap_to_ranked_bitmap(const unsigned char* bytemap, long long nwords, unsigned* bitmap, int32_t* prefix, int32_t* total,
                                 void* ws, hipStream_t stream);

// Geometry passed by value to kernels.
struct BtcGeom {
  int in_shape[3];
  int out_shape[3];
  int k[3];
  int s[3];
  int p[3];
  int d[3];
  int K;
  int mode;
};

__device__ __forceinline__ unsigned btc_hash32(unsigned key) {
  key *= 2654435761u;
  key ^= key >> 15;
  return key;
}
