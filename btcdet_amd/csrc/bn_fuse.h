// Batch statistics of a sparse conv's result for the training-mode BatchNorm1d that follows it (the reference's post_act_block:
// conv -> BatchNorm1d(eps 1e-3, momentum 0.01) -> ReLU, spconv_backbone.py:33-43), gathered in the conv kernel's EPILOGUE: the
// 16 x 16 result tiles are still in registers there, so the separate statistics pass over the freshly written (N, C) tensor -- one
// launch per layer at its latency floor and a second read of every activation -- disappears (27 launches per training step).
//
// Each wave reduces its tile columns (fp64), adds them to one of nslots slot rows with device-scope fp64 atomics, and the
// LAST workgroup to arrive (release -> ticket -> acquire, as bn.hip) sums the slots in index order, publishes mean / rstd, updates
// the running statistics and num_batches_tracked and leaves slots and counter zeroed for the next layer on that stream.  The slot
// stride is fixed (BN_FUSE_CMAX channels), so whatever channel count used the buffer last has cleaned exactly what it touched.
// Sums of fp32 values (and of their exact fp64 squares) in fp64: the order the atomics land in moves the sums by ~1e-16 relative,
// i.e. mean / rstd are run-to-run identical after their rounding to fp32 except on a rounding boundary.
#pragma once
#include "btc_common.h"

// Slots: a wave adds its column sums to slot (its tile index) mod nslots.  Atomics on ONE address are served one after the other at the
// L2 (~150 ns each): a 210 K-row layer issues ~840 K of them, with 32 slots x 64 channel sums that was ~400 per address = 60 us of
// a 100 us launch (found in round 5 when the bf16-operand kernel got this epilogue).  nslots therefore grows with the row count
// (btc_bn_fuse_nslots: 32 .. BN_FUSE_SLOTS_MAX), and the last workgroup sums the slots with all of its threads.
constexpr int BN_FUSE_SLOTS_MAX = 512;
constexpr int BN_FUSE_CMAX = 1024;

struct BnFuse {
  double* slots;        // [nslots <= BN_FUSE_SLOTS_MAX][2][BN_FUSE_CMAX]: sum, sum of squares; zero on entry, zero on exit.  nullptr: no statistics
  int32_t* counter;     // arrival counter, zero on entry / exit
  float* mean_out;      // [C]
  float* rstd_out;      // [C]
  float* running_mean;  // [C] or nullptr
  float* running_var;
  long long* num_batches;
  float momentum, eps;
  int N, C;
  int nslots;           // power of two, 32 .. BN_FUSE_SLOTS_MAX
};

static inline size_t btc_bn_fuse_bytes() { return 256 + (size_t)BN_FUSE_SLOTS_MAX * 2 * BN_FUSE_CMAX * sizeof(double); }

static inline int btc_bn_fuse_nslots(long long n_rows) {
  int s = 32;
  while (s < BN_FUSE_SLOTS_MAX && (long long)s * 1024 < n_rows) s <<= 1;
  return s;
}

static inline BnFuse btc_bn_fuse_none() {
  BnFuse b;
  b.slots = nullptr; b.counter = nullptr; b.mean_out = b.rstd_out = b.running_mean = b.running_var = nullptr; b.num_batches = nullptr;
  b.momentum = b.eps = 0.f; b.N = b.C = 0; b.nslots = 32;
  return b;
}

// one wave's NT 16 x 16 tiles in the MFMA C/D layout (column = lane & 15, rows (lane >> 4) * 4 + r): v[nt][r] the values as STORED
// (bias added, rounded to bf16 where the tensor is bf16), valid[r] whether row r exists, col0 the column of tile 0, lane 0
template <int NT>
__device__ __forceinline__ void bn_fuse_wave(const BnFuse& bn, const float (&v)[NT][4], const bool (&valid)[4], int col0, int slot) {
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    double s1 = 0.0, s2 = 0.0;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const double x = valid[r] ? (double)v[nt][r] : 0.0;
      s1 += x;
      s2 += x * x;
    }
    s1 += __shfl_xor(s1, 16, 64);
    s2 += __shfl_xor(s2, 16, 64);
    s1 += __shfl_xor(s1, 32, 64);
    s2 += __shfl_xor(s2, 32, 64);
    const int col = col0 + nt * 16 + lane;
    if (lane < 16 && col < bn.C) {
      double* p = bn.slots + ((size_t)slot * 2) * BN_FUSE_CMAX + col;
      unsafeAtomicAdd(p, s1);
      unsafeAtomicAdd(p + BN_FUSE_CMAX, s2);
    }
  }
}

// end of the kernel, every thread of every workgroup: the last workgroup to arrive turns the slots into mean / rstd.
// s_flag: one int of the workgroup's LDS that nobody needs any more (the kernels run at the 160 KB dynamic limit: no static LDS here);
// s_part: blockDim.x * 2 doubles of dead LDS (or nullptr): with it the slot sums of a channel are shared by blockDim.x / C threads
__device__ __forceinline__ void bn_fuse_finish(const BnFuse& bn, int* s_flag, double* s_part = nullptr) {
  const int tid = threadIdx.x;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (tid == 0) {
    // NO agent-scope release fence here (round 5).  What the last arriver reads are the slot sums, and those are written by device-scope
    // ATOMICS only -- sc1 operations, performed at the coherence point, never left dirty in this XCD's L2 -- so "every wave drained its
    // vector-memory queue (the s_waitcnt above), then the ticket" is the whole protocol (MI355X_MICROARCH.md, handoff-flag: `sc1`
    // payload -> vmcnt(0) -> `sc1` flag).  The fence that stood here is `buffer_wbl2 sc1`: a write-back of the XCD L2's dirty lines --
    // i.e. of every conv result tile written a moment ago -- ONCE PER WORKGROUP: 2-6 us each, ~430 us of the 13 K-workgroup launches
    // of the bf16-operand kernel on the 210 K-row level, which is how it was found.
    const int total = (int)(gridDim.x * gridDim.y * gridDim.z);
    *s_flag = (btc_ticket_take(bn.counter) == total - 1);   // (the protocol in one place: btc_common.h)
  }
  __syncthreads();
  if (!*s_flag) return;
  if (tid == 0) btc_ticket_acquire();
  __syncthreads();
  auto publish = [&](int c, double a, double b) {
    const double mean = a / bn.N;
    double var = b / bn.N - mean * mean;  // biased, as F.batch_norm normalises with
    if (var < 0.0) var = 0.0;
    bn.mean_out[c] = (float)mean;
    bn.rstd_out[c] = (float)(1.0 / sqrt(var + (double)bn.eps));
    if (bn.running_mean) {
      const double unb = bn.N > 1 ? var * ((double)bn.N / (double)(bn.N - 1)) : var;
      bn.running_mean[c] = (float)((1.0 - bn.momentum) * bn.running_mean[c] + bn.momentum * mean);
      bn.running_var[c] = (float)((1.0 - bn.momentum) * bn.running_var[c] + bn.momentum * unb);
    }
  };
  const int T = (int)blockDim.x;
  if (s_part && bn.C <= T) {
    const int G = T / bn.C;                 // threads per channel
    const int c = tid % bn.C, g = tid / bn.C;
    double a = 0.0, b = 0.0;
    if (g < G)
      for (int s = g; s < bn.nslots; s += G) {
        double* p = bn.slots + ((size_t)s * 2) * BN_FUSE_CMAX + c;
        a += btc_ld_agent(p);
        b += btc_ld_agent(p + BN_FUSE_CMAX);
        p[0] = 0.0;
        p[BN_FUSE_CMAX] = 0.0;
      }
    s_part[2 * tid] = a;
    s_part[2 * tid + 1] = b;
    __syncthreads();
    if (tid < bn.C) {
      a = b = 0.0;
      for (int q = 0; q < G; ++q) {         // (fixed order: deterministic given the slot contents)
        a += s_part[2 * (q * bn.C + tid)];
        b += s_part[2 * (q * bn.C + tid) + 1];
      }
      publish(tid, a, b);
    }
  } else {
    for (int c = tid; c < bn.C; c += T) {
      double a = 0.0, b = 0.0;
#pragma unroll 8
      for (int s = 0; s < bn.nslots; ++s) {
        double* p = bn.slots + ((size_t)s * 2) * BN_FUSE_CMAX + c;
        a += btc_ld_agent(p);
        b += btc_ld_agent(p + BN_FUSE_CMAX);
        p[0] = 0.0;
        p[BN_FUSE_CMAX] = 0.0;
      }
      publish(c, a, b);
    }
  }
  if (tid == 0) {
    if (bn.num_batches) *bn.num_batches += 1;
    __hip_atomic_store(bn.counter, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}
