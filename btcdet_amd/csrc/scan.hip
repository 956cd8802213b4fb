// Device-wide exclusive prefix sum of int32 (three launches: block reduce, scan of block sums,
// block scan + offset).  Wave64 shuffles for the in-wave scan, LDS for the 4 waves of a block.
#include <stdarg.h>

#include <cstdlib>

#include "btc_common.h"

static thread_local char g_err[512] = "";

void btc_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* btc_last_error(void) { return g_err; }
extern "C" int btc_version(void) { return 1; }

// kernel-selection overrides for tuning runs (tools/conv_bench.py); 0 = the built-in policy
static int g_tune[BTC_TUNE_KEYS] = {0};
int btc_tune_get(int key) { return (key >= 0 && key < BTC_TUNE_KEYS) ? g_tune[key] : 0; }
extern "C" int btc_tune_value(int key) { return btc_tune_get(key); }
extern "C" int btc_tune_set(int key, int value) {
  BTC_CHECK_ARG(key >= 0 && key < BTC_TUNE_KEYS, "btc_tune_set: unknown key %d", key);
  // the one key that changes RESULTS (kernel phases switched off for timing experiments) needs an explicit unlock in the
  // environment of the process: a stray BTC_TUNE entry must not be able to make a production run compute garbage
  BTC_CHECK_ARG(key != BTC_TUNE_APPLY_DEBUG || value == 0 || getenv("BTC_ALLOW_WRONG_RESULTS") != nullptr,
                "btc_tune_set: BTC_TUNE_APPLY_DEBUG produces wrong results; set BTC_ALLOW_WRONG_RESULTS=1 for timing experiments");
  g_tune[key] = value;
  return BTC_OK;
}

// per-stream scratch buffers (btc_set_scratch): caller-allocated device memory the library may use for the lifetime of a launch on
// that stream -- today the partial slabs of z-split sparse-conv launches (conv_apply_split.hip).  Launches on one stream are
// serialised, so one buffer per stream is enough; the caller keeps it alive until it registers another (or NULL) for the stream.
#include <mutex>
#include <unordered_map>
// Keyed by (current device, stream handle): the default stream's handle is 0 on EVERY device, so a process that drives two GPUs
// must not find device 1's buffer when it launches on device 0.
#include <map>
static std::mutex g_scratch_mu;
static std::map<std::pair<int, void*>, std::pair<void*, size_t>> g_scratch;
static std::pair<int, void*> scratch_key(void* stream) {
  int dev = 0;
  (void)hipGetDevice(&dev);
  return std::make_pair(dev, stream);
}
extern "C" int btc_set_scratch(void* stream, void* ptr, size_t bytes) {
  auto key = scratch_key(stream);
  hipPointerAttribute_t attr;
  if (ptr && hipPointerGetAttributes(&attr, ptr) == hipSuccess) key.first = attr.device;   // the buffer's device, whatever is current
  else (void)hipGetLastError();
  if (ptr && bytes >= BTC_SCRATCH_HEAD)   // the head holds counters the kernels expect zero (and leave zero): cleared once, on the buffer's stream
    BTC_HIP(hipMemsetAsync(ptr, 0, BTC_SCRATCH_HEAD, (hipStream_t)stream));
  std::lock_guard<std::mutex> lock(g_scratch_mu);
  if (!ptr || !bytes) g_scratch.erase(key);
  else g_scratch[key] = std::make_pair(ptr, bytes);
  return BTC_OK;
}
void* btc_scratch(hipStream_t stream, size_t* bytes) {
  const auto key = scratch_key((void*)stream);
  std::lock_guard<std::mutex> lock(g_scratch_mu);
  auto it = g_scratch.find(key);
  if (it == g_scratch.end()) { *bytes = 0; return nullptr; }
  *bytes = it->second.second;
  return it->second.first;
}

namespace {

constexpr int SCAN_THREADS = 256;
constexpr int SCAN_ITEMS = 8;
constexpr int SCAN_TILE = SCAN_THREADS * SCAN_ITEMS;

__device__ __forceinline__ int wave_incl_scan(int v) {
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    int t = __shfl_up(v, o, 64);
    if (lane >= o) v += t;
  }
  return v;
}

// exclusive scan across the block of per-thread values; returns exclusive prefix, total in *block_total
template <int THREADS>
__device__ __forceinline__ int block_excl_scan(int v, int* s_wave /* THREADS/64 + 1 */, int* block_total) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int incl = wave_incl_scan(v);
  if (lane == 63) s_wave[wave] = incl;
  __syncthreads();
  if (threadIdx.x == 0) {
    int run = 0;
#pragma unroll
    for (int w = 0; w < THREADS / 64; ++w) {
      int t = s_wave[w];
      s_wave[w] = run;
      run += t;
    }
    s_wave[THREADS / 64] = run;
  }
  __syncthreads();
  int r = incl - v + s_wave[wave];
  *block_total = s_wave[THREADS / 64];
  __syncthreads();
  return r;
}

__global__ __launch_bounds__(SCAN_THREADS) void scan_block_reduce(const int32_t* __restrict__ in, long long n,
                                                                  int32_t* __restrict__ block_sums) {
  __shared__ int s_wave[SCAN_THREADS / 64];
  long long base = (long long)blockIdx.x * SCAN_TILE + (long long)threadIdx.x * SCAN_ITEMS;
  int sum = 0;
#pragma unroll
  for (int j = 0; j < SCAN_ITEMS; ++j)
    if (base + j < n) sum += in[base + j];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) sum += __shfl_down(sum, o, 64);
  if ((threadIdx.x & 63) == 0) s_wave[threadIdx.x >> 6] = sum;
  __syncthreads();
  if (threadIdx.x == 0) {
    int t = 0;
#pragma unroll
    for (int w = 0; w < SCAN_THREADS / 64; ++w) t += s_wave[w];
    block_sums[blockIdx.x] = t;
  }
}

__global__ __launch_bounds__(1024) void scan_block_sums(int32_t* __restrict__ block_sums, int nblocks,
                                                        int32_t* __restrict__ total) {
  __shared__ int s_wave[1024 / 64 + 1];
  int carry = 0;
  for (int base = 0; base < nblocks; base += 1024) {
    int i = base + threadIdx.x;
    int v = i < nblocks ? block_sums[i] : 0;
    int tot;
    int ex = block_excl_scan<1024>(v, s_wave, &tot);
    if (i < nblocks) block_sums[i] = ex + carry;
    carry += tot;
  }
  if (threadIdx.x == 0 && total) *total = carry;
}

__global__ __launch_bounds__(SCAN_THREADS) void scan_final(const int32_t* __restrict__ in, int32_t* __restrict__ out,
                                                           long long n, const int32_t* __restrict__ block_sums) {
  __shared__ int s_wave[SCAN_THREADS / 64 + 1];
  long long base = (long long)blockIdx.x * SCAN_TILE + (long long)threadIdx.x * SCAN_ITEMS;
  int v[SCAN_ITEMS];
  int sum = 0;
#pragma unroll
  for (int j = 0; j < SCAN_ITEMS; ++j) {
    v[j] = (base + j < n) ? in[base + j] : 0;
    sum += v[j];
  }
  int tot;
  int ex = block_excl_scan<SCAN_THREADS>(sum, s_wave, &tot) + block_sums[blockIdx.x];
#pragma unroll
  for (int j = 0; j < SCAN_ITEMS; ++j) {
    if (base + j < n) out[base + j] = ex;
    ex += v[j];
  }
}

// scan_block_sums + scan_final in one launch (round 5: the step is bound by its launch count): every block adds up the block sums in front
// of it itself -- at most SCAN_FUSE_BLOCKS values, read from L2 -- instead of waiting for a one-workgroup launch that scans them
constexpr int SCAN_FUSE_BLOCKS = 4096;

__global__ __launch_bounds__(SCAN_THREADS) void scan_final_fused(const int32_t* __restrict__ in, int32_t* __restrict__ out, long long n,
                                                                 const int32_t* __restrict__ block_sums, int32_t* __restrict__ total) {
  __shared__ int s_wave[SCAN_THREADS / 64 + 1];
  __shared__ int s_off[SCAN_THREADS / 64];
  int part = 0;
  for (int j = threadIdx.x; j < (int)blockIdx.x; j += SCAN_THREADS) part += block_sums[j];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) part += __shfl_down(part, o, 64);
  if ((threadIdx.x & 63) == 0) s_off[threadIdx.x >> 6] = part;
  __syncthreads();
  int offset = 0;
#pragma unroll
  for (int w = 0; w < SCAN_THREADS / 64; ++w) offset += s_off[w];
  long long base = (long long)blockIdx.x * SCAN_TILE + (long long)threadIdx.x * SCAN_ITEMS;
  int v[SCAN_ITEMS];
  int sum = 0;
#pragma unroll
  for (int j = 0; j < SCAN_ITEMS; ++j) {
    v[j] = (base + j < n) ? in[base + j] : 0;
    sum += v[j];
  }
  int tot;
  int ex = block_excl_scan<SCAN_THREADS>(sum, s_wave, &tot) + offset;
#pragma unroll
  for (int j = 0; j < SCAN_ITEMS; ++j) {
    if (base + j < n) out[base + j] = ex;
    ex += v[j];
  }
  if (total && blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) *total = offset + tot;
}

// ---- byte map -> ranked bitmap (rulebook of strided / transposed convs and pools) ------------------------------
// The reachable output cells are marked with plain byte stores (many writers, one value: no atomics -- device-scope
// atomicOr on a shared bitmap word runs at the memory side on this multi-XCD part and measured 40-80 us per rulebook).
// bytes_to_bits packs 32 bytes into a bitmap word and leaves the per-block popcount sums for the scan.
__global__ __launch_bounds__(SCAN_THREADS) void bytes_to_bits(const uint4* __restrict__ bytemap, long long nwords,
                                                              unsigned* __restrict__ bitmap, int32_t* __restrict__ block_sums) {
  __shared__ int s_wave[SCAN_THREADS / 64];
  const long long wbase = (long long)blockIdx.x * SCAN_TILE;
  int sum = 0;
#pragma unroll
  for (int j = 0; j < SCAN_ITEMS; ++j) {
    const long long w = wbase + j * SCAN_THREADS + threadIdx.x;  // coalesced: consecutive lanes take consecutive words
    if (w < nwords) {
      const uint4 a = bytemap[2 * w], b = bytemap[2 * w + 1];
      const unsigned d[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
      unsigned bits = 0;
#pragma unroll
      for (int q = 0; q < 8; ++q) bits |= (((d[q] & 0x01010101u) * 0x01020408u) >> 24) << (4 * q);  // bytes are 0 / 1
      bitmap[w] = bits;
      sum += __popc(bits);
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) sum += __shfl_down(sum, o, 64);
  if ((threadIdx.x & 63) == 0) s_wave[threadIdx.x >> 6] = sum;
  __syncthreads();
  if (threadIdx.x == 0) {
    int t = 0;
#pragma unroll
    for (int w = 0; w < SCAN_THREADS / 64; ++w) t += s_wave[w];
    block_sums[blockIdx.x] = t;
  }
}

// prefix[w] = number of set bits in words [0, w), w in [0, nwords] (block_sums already scanned)
__global__ __launch_bounds__(SCAN_THREADS) void bitmap_prefix(const unsigned* __restrict__ bitmap, long long nwords,
                                                              const int32_t* __restrict__ block_sums, int32_t* __restrict__ prefix) {
  __shared__ int s_wave[SCAN_THREADS / 64 + 1];
  long long base = (long long)blockIdx.x * SCAN_TILE + (long long)threadIdx.x * SCAN_ITEMS;
  int v[SCAN_ITEMS];
  int sum = 0;
#pragma unroll
  for (int j = 0; j < SCAN_ITEMS; ++j) {
    v[j] = (base + j < nwords) ? __popc(bitmap[base + j]) : 0;
    sum += v[j];
  }
  int tot;
  int ex = block_excl_scan<SCAN_THREADS>(sum, s_wave, &tot) + block_sums[blockIdx.x];
#pragma unroll
  for (int j = 0; j < SCAN_ITEMS; ++j) {
    if (base + j <= nwords) prefix[base + j] = ex;
    ex += v[j];
  }
}

}  // namespace

size_t btc_scan_ws_bytes(long long n) {
  long long nblocks = (n + SCAN_TILE - 1) / SCAN_TILE;
  if (nblocks < 1) nblocks = 1;
  return btc_align((size_t)nblocks * sizeof(int32_t));
}

int btc_scan_exclusive_i32(const int32_t* in, int32_t* out, long long n, int32_t* total, void* ws, hipStream_t stream) {
  if (n <= 0) {
    if (total) BTC_HIP(hipMemsetAsync(total, 0, sizeof(int32_t), stream));
    return BTC_OK;
  }
  int nblocks = (int)((n + SCAN_TILE - 1) / SCAN_TILE);
  int32_t* block_sums = (int32_t*)ws;
  if (nblocks == 1) {   // one tile: the fused kernel's block 0 reads no block sums at all
    scan_final_fused<<<1, SCAN_THREADS, 0, stream>>>(in, out, n, block_sums, total);
    BTC_LAUNCH_CHECK();
    return BTC_OK;
  }
  scan_block_reduce<<<nblocks, SCAN_THREADS, 0, stream>>>(in, n, block_sums);
  BTC_LAUNCH_CHECK();
  if (nblocks <= SCAN_FUSE_BLOCKS) {   // (in may alias out: a block reads its own tile before it writes it, and block_sums is a separate buffer)
    scan_final_fused<<<nblocks, SCAN_THREADS, 0, stream>>>(in, out, n, block_sums, total);
    BTC_LAUNCH_CHECK();
    return BTC_OK;
  }
  scan_block_sums<<<1, 1024, 0, stream>>>(block_sums, nblocks, total);
  BTC_LAUNCH_CHECK();
  scan_final<<<nblocks, SCAN_THREADS, 0, stream>>>(in, out, n, block_sums);
  BTC_LAUNCH_CHECK();
  return BTC_OK;
}

// bytemap: one byte per cell (0 / 1), padded with zeros to btc_bytemap_bytes(nwords); bitmap: nwords words;
// prefix: nwords + 1 entries (prefix[nwords] = *total = number of marked cells); ws: btc_scan_ws_bytes(nwords + 1)
size_t btc_bytemap_bytes(long long nwords) { return btc_align((size_t)nwords * 32); }

int btc_bytemap_to_ranked_bitmap(const unsigned char* bytemap, long long nwords, unsigned* bitmap, int32_t* prefix, int32_t* total,
                                 void* ws, hipStream_t stream) {
  if (nwords <= 0) {
    BTC_HIP(hipMemsetAsync(prefix, 0, sizeof(int32_t), stream));
    if (total) BTC_HIP(hipMemsetAsync(total, 0, sizeof(int32_t), stream));
    return BTC_OK;
  }
  const int nblocks = (int)((nwords + 1 + SCAN_TILE - 1) / SCAN_TILE);  // the prefix has nwords + 1 entries
  int32_t* block_sums = (int32_t*)ws;
  bytes_to_bits<<<nblocks, SCAN_THREADS, 0, stream>>>((const uint4*)bytemap, nwords, bitmap, block_sums);
  BTC_LAUNCH_CHECK();
  scan_block_sums<<<1, 1024, 0, stream>>>(block_sums, nblocks, total);
  BTC_LAUNCH_CHECK();
  bitmap_prefix<<<nblocks, SCAN_THREADS, 0, stream>>>(bitmap, nwords, block_sums, prefix);
  BTC_LAUNCH_CHECK();
  return BTC_OK;
}
