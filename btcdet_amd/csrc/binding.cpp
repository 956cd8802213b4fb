// Compiled binding of the hot entry points of libbtcdet_hip.so for PyTorch (CPython extension `btcdet_amd._btcfast`).
//
// spconv reaches its native code through a pybind11 layer (spconv/ops.py -> torch.ops.spconv.*); this is the same layer for
// this library.  It exists because the forward pass is bound by the host's launch rate (tools/host_phases.py,
// tools/layer_host_split.py): per sparse layer the ctypes route spends ~30 us of Python on output allocation, argument
// marshalling and attribute lookups around ~12 us of launches.  Here one call allocates the outputs with at::empty and
// calls the C ABI (include/btcdet_hip.h) directly; memory comes from torch's caching allocator, streams from c10::hip.  The
// conv -> BatchNorm -> ReLU triple is additionally a C++ autograd node (no Python Function.apply per layer), with wgrad on a
// side stream beside dgrad for mid-size layers.  The ctypes route (btcdet_amd/_lib.py) stays the reference binding
//  and computes the same thing; tests run both.
#include <torch/extension.h>
#include <c10/hip/HIPCachingAllocator.h>
#include <c10/hip/HIPStream.h>
#include <c10/hip/HIPGuard.h>
#include <hip/hip_runtime_api.h>
#include <torch/csrc/autograd/engine.h>

#include <atomic>
#include <mutex>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <cstdio>
#include <cstdlib>
#include <memory>
#include <tuple>
#include <vector>

#include "../../include/btcdet_hip.h"

namespace {

using at::Tensor;
using OptTensor = c10::optional<at::Tensor>;

inline void chk(int rc, const char* what) {
  if (rc != 0) throw std::runtime_error(std::string(what) + " failed (code " + std::to_string(rc) + "): " + btc_last_error());
}

inline const float* fptr(const OptTensor& t) { return (t.has_value() && t->defined()) ? (const float*)t->data_ptr() : nullptr; }
inline void* vptr(const OptTensor& t) { return (t.has_value() && t->defined()) ? t->data_ptr() : nullptr; }
inline void* st(int64_t stream) { return reinterpret_cast<void*>(stream); }
inline const int32_t* ip(int64_t p) { return reinterpret_cast<const int32_t*>(p); }


inline void need(bool ok, const char* msg) {
  if (!ok) throw std::runtime_error(msg);
}

// bf16 copies of a layer's weights for the bf16-operand kernels (btc_conv_*_bf16w): row 0 = W [K][Cin][Cout] (dgrad operand),
// row 1 = W^T [K][Cout][Cin] (forward operand).  Converted when first needed after the parameter changed (its version counter
// moves with every in-place optimizer update / load_state_dict), i.e. once per optimizer step: the forward pass converts,
// the backward pass of the same step finds the copy.  The weak reference keeps a freed parameter's slot from being taken
// for a live one.
struct WqEntry {
  c10::weak_intrusive_ptr<c10::TensorImpl> weak;
  uint32_t version;
  Tensor q;
  WqEntry(c10::weak_intrusive_ptr<c10::TensorImpl> w, uint32_t v, Tensor t) : weak(std::move(w)), version(v), q(std::move(t)) {}
};
std::mutex g_wq_mu;
std::unordered_map<const void*, WqEntry> g_wq;

bool bf16_operands(const Tensor& features, int64_t K, int64_t cred, int64_t cres) {
  return features.scalar_type() == at::kBFloat16 && btc_conv_bf16w_supported((int)K, (int)cred, (int)cres) &&
         btc_tune_value(BTC_TUNE_BF16_OPERANDS) != 1;
}

// planes = 1: the bf16 copies; planes = 3: the hi / mid / lo planes of the split-operand kernel (btc_weights_split3) -- same rows
std::unordered_map<const void*, WqEntry> g_ws;

void convert_weights(const Tensor& w, int64_t K, int64_t cin, int64_t cout, int planes, Tensor& q, int64_t stream) {
  char* row1 = (char*)q.data_ptr() + 2 * planes * w.numel();
  if (planes == 1)
    chk(btc_weights_to_bf16((const float*)w.data_ptr(), (int)K, (int)cin, (int)cout, q.data_ptr(), row1, st(stream)), "btc_weights_to_bf16");
  else
    chk(btc_weights_split3((const float*)w.data_ptr(), (int)K, (int)cin, (int)cout, q.data_ptr(), row1, st(stream)), "btc_weights_split3");
}

Tensor weights_q(const Tensor& w, int64_t K, int64_t cin, int64_t cout, int64_t stream, int planes) {
  const void* key = w.unsafeGetTensorImpl();
  const uint32_t version = (uint32_t)w._version();
  if (!w.is_leaf()) {   // a temporary (the zero-padded 34 -> 48 channel weight is a fresh tensor every step): convert, do not cache
    Tensor q = at::empty({2, planes * w.numel()}, w.options().dtype(at::kBFloat16));
    convert_weights(w, K, cin, cout, planes, q, stream);
    return q;
  }
  std::lock_guard<std::mutex> lock(g_wq_mu);
  auto& tab = planes == 1 ? g_wq : g_ws;
  auto it = tab.find(key);
  if (it != tab.end() && !it->second.weak.expired() && it->second.version == version && it->second.q.get_device() == w.get_device())
    return it->second.q;
  Tensor q = (it != tab.end() && !it->second.weak.expired() && it->second.q.numel() == 2 * planes * w.numel() && it->second.q.get_device() == w.get_device())
                 ? it->second.q : at::empty({2, planes * w.numel()}, w.options().dtype(at::kBFloat16));
  convert_weights(w, K, cin, cout, planes, q, stream);
  if (it != tab.end()) tab.erase(it);
  tab.emplace(key, WqEntry(c10::weak_intrusive_ptr<c10::TensorImpl>(w.getIntrusivePtr()), version, q));
  // models come and go in a long-lived process (tests): every miss drops the entries of freed parameters (a few dozen entries)
  for (auto e = tab.begin(); e != tab.end();) e = e->second.weak.expired() ? tab.erase(e) : std::next(e);
  return q;
}

Tensor weights_bf16(const Tensor& w, int64_t K, int64_t cin, int64_t cout, int64_t stream) { return weights_q(w, K, cin, cout, stream, 1); }

// the operand copies of MANY leaf weights in one launch (a parameter group's eligible layers right after its optimizer step, on the
// stream its next forward runs on): every cache entry is brought to the weight's current version, later weights_q calls hit.
// planes = 3: the hi / mid / lo planes of the split-operand kernel (fp32 features); planes = 1: the bf16 copies (bf16 features)
void refresh_weights(const std::vector<Tensor>& weights, int64_t stream, int planes) {
  const size_t n = weights.size();
  if (n == 0) return;
  std::vector<const float*> W(n);
  std::vector<void*> ws(n), wts(n);
  std::vector<int32_t> K(n), Cin(n), Cout(n);
  std::lock_guard<std::mutex> lock(g_wq_mu);
  auto& tab = planes == 1 ? g_wq : g_ws;
  for (size_t i = 0; i < n; ++i) {
    const Tensor& w = weights[i];
    need(w.is_leaf() && w.is_contiguous() && w.scalar_type() == at::kFloat && w.dim() >= 3, "refresh_weights: contiguous fp32 leaf weights [.., Cin, Cout] expected");
    const int64_t cin = w.size(-2), cout = w.size(-1);
    const void* key = w.unsafeGetTensorImpl();
    auto it = tab.find(key);
    Tensor q = (it != tab.end() && !it->second.weak.expired() && it->second.q.numel() == 2 * planes * w.numel() && it->second.q.get_device() == w.get_device())
                   ? it->second.q : at::empty({2, planes * w.numel()}, w.options().dtype(at::kBFloat16));
    if (it != tab.end()) tab.erase(it);
    tab.emplace(key, WqEntry(c10::weak_intrusive_ptr<c10::TensorImpl>(w.getIntrusivePtr()), (uint32_t)w._version(), q));
    W[i] = (const float*)w.data_ptr();
    ws[i] = q.data_ptr();
    wts[i] = (char*)q.data_ptr() + 2 * planes * w.numel();
    K[i] = (int32_t)(w.numel() / (cin * cout)); Cin[i] = (int32_t)cin; Cout[i] = (int32_t)cout;
  }
  if (planes == 1)
    chk(btc_weights_to_bf16_multi(W.data(), ws.data(), wts.data(), K.data(), Cin.data(), Cout.data(), (int)n, st(stream)), "btc_weights_to_bf16_multi");
  else
    chk(btc_weights_split3_multi(W.data(), ws.data(), wts.data(), K.data(), Cin.data(), Cout.data(), (int)n, st(stream)), "btc_weights_split3_multi");
}

void split_weights(const std::vector<Tensor>& weights, int64_t stream) { refresh_weights(weights, stream, 3); }
void bf16_weights(const std::vector<Tensor>& weights, int64_t stream) { refresh_weights(weights, stream, 1); }

// the stream's scratch buffer for z-split launches of the split-operand kernel (btc_set_scratch): 48 MB, allocated on first use
// from the stream's own pool and kept for the life of the process
void ensure_scratch(const Tensor& like, int64_t stream) {
  static std::mutex mu;
  static std::unordered_map<uint64_t, Tensor> tab;
  const uint64_t key = ((uint64_t)(uint8_t)like.get_device() << 56) ^ (uint64_t)stream;
  std::lock_guard<std::mutex> lock(mu);
  if (tab.find(key) != tab.end()) return;
  Tensor t = at::empty({(int64_t)48 << 20}, like.options().dtype(at::kByte));
  chk(btc_set_scratch(st(stream), t.data_ptr(), (size_t)t.numel()), "btc_set_scratch");
  tab.emplace(key, t);
}

// fp32 launch of n_rows rows on the split-operand kernel (csrc/conv_apply_split.hip)?  The library's policy; BTC_TUNE_SPLIT = 1: never
bool split_operands(const Tensor& src, int64_t K, int64_t cred, int64_t cres, int64_t n_rows, int64_t stream) {
  if (!(src.scalar_type() == at::kFloat && btc_conv_split_wanted((int)K, (int)cred, (int)cres, (int)n_rows))) return false;
  if (src.numel() * 4 >= (int64_t)0xFFFFFF00LL) return false;   // the kernel's gathers use 32-bit byte offsets (the C entry refuses past them): exact kernels
  ensure_scratch(src, stream);
  return true;
}

// out = conv(features) ; features (n_src, Cin) fp32 | bf16 contiguous, w [K.., Cin, Cout] fp32, map_fwd (n_res, K) int32
// a row-order hint of a map (csrc/row_order.hip): int32 (n_rows,), or nothing
const int32_t* order_ptr(const OptTensor& o, int64_t n_rows, const char* what) {
  if (!o.has_value() || !o->defined()) return nullptr;
  need(o->scalar_type() == at::kInt && o->is_contiguous() && o->numel() == n_rows, what);
  return (const int32_t*)o->data_ptr();
}

Tensor conv_fwd(const Tensor& features, const Tensor& w, const OptTensor& bias, const Tensor& map_fwd, const OptTensor& order_fwd, int64_t stream) {
  const int64_t cin = w.size(-2), cout = w.size(-1), K = map_fwd.size(1), n_res = map_fwd.size(0);
  need(features.is_contiguous() && w.is_contiguous() && map_fwd.is_contiguous(), "conv_fwd: contiguous tensors expected");
  need(w.numel() == K * cin * cout && features.size(1) == cin, "conv_fwd: weight does not match the rulebook / features");
  const int32_t* order = order_ptr(order_fwd, n_res, "conv_fwd: the row order does not match the map");
  Tensor out = at::empty({n_res, cout}, features.options());
  if (bf16_operands(features, K, cin, cout)) {
    Tensor q = weights_bf16(w, K, cin, cout, stream);
    chk(btc_conv_apply_ordered(BTC_PASS_FWD, BTC_OPERANDS_BF16, features.data_ptr(), (const char*)q.data_ptr() + 2 * w.numel(), fptr(bias),
                               (const int32_t*)map_fwd.data_ptr(), order, (int)n_res, (int)K, (int)cin, (int)cout, out.data_ptr(), st(stream)),
        "btc_conv_apply_ordered (fwd, bf16 operands)");
  } else if (split_operands(features, K, cin, cout, n_res, stream)) {
    Tensor q = weights_q(w, K, cin, cout, stream, 3);
    chk(btc_conv_apply_src(BTC_PASS_FWD, BTC_OPERANDS_F32_SPLIT, features.data_ptr(), (long long)features.size(0), (const char*)q.data_ptr() + 6 * w.numel(), fptr(bias),
                               (const int32_t*)map_fwd.data_ptr(), order, (int)n_res, (int)K, (int)cin, (int)cout, out.data_ptr(), st(stream)),
        "btc_conv_apply_src (fwd, split operands)");
  } else {
    const int operands = features.scalar_type() == at::kBFloat16 ? BTC_OPERANDS_BF16_ACT : BTC_OPERANDS_F32;
    chk(btc_conv_apply_ordered(BTC_PASS_FWD, operands, features.data_ptr(), w.data_ptr(), fptr(bias), (const int32_t*)map_fwd.data_ptr(), order,
                               (int)n_res, (int)K, (int)cin, (int)cout, out.data_ptr(), st(stream)), "btc_conv_apply_ordered (fwd)");
  }
  return out;
}

// y = [relu](batchnorm(x)); stats (2, C) = mean | rstd
std::tuple<Tensor, Tensor> bn_fwd(const Tensor& x, const OptTensor& gamma, const OptTensor& beta, const OptTensor& rm, const OptTensor& rv,
                                  const OptTensor& nbt, bool use_batch, double momentum, double eps, bool relu, const Tensor& ws, int64_t ws_bytes,
                                  int64_t stream) {
  const int64_t N = x.size(0), C = x.size(1);
  Tensor y = at::empty_like(x);
  Tensor stats = at::empty({2, C}, x.options().dtype(at::kFloat));
  float* mean = (float*)stats.data_ptr();
  float* rstd = mean + C;
  long long* nb = (nbt.has_value() && nbt->defined()) ? (long long*)nbt->data_ptr() : nullptr;
  if (x.scalar_type() == at::kBFloat16)
    chk(btc_bn_relu_fwd_bf16(x.data_ptr(), (int)N, (int)C, fptr(gamma), fptr(beta), (float*)vptr(rm), (float*)vptr(rv), nb, (float)momentum,
                             (float)eps, (int)use_batch, (int)relu, y.data_ptr(), mean, rstd, ws.data_ptr(), (size_t)ws_bytes, st(stream)),
        "btc_bn_relu_fwd_bf16");
  else
    chk(btc_bn_relu_fwd((const float*)x.data_ptr(), (int)N, (int)C, fptr(gamma), fptr(beta), (float*)vptr(rm), (float*)vptr(rv), nb,
                        (float)momentum, (float)eps, (int)use_batch, (int)relu, (float*)y.data_ptr(), mean, rstd, ws.data_ptr(), (size_t)ws_bytes,
                        st(stream)), "btc_bn_relu_fwd");
  return std::make_tuple(y, stats);
}

// zero-initialised slot buffer of the fused conv + BatchNorm statistics (csrc/bn_fuse.h), one per (device, stream): the kernels
// leave it zeroed, and launches on one stream are serialised
Tensor fuse_ws_of(const Tensor& like, int64_t stream) {
  static std::mutex mu;
  static std::unordered_map<uint64_t, Tensor> tab;
  const uint64_t key = ((uint64_t)(uint8_t)like.get_device() << 56) ^ (uint64_t)stream;
  std::lock_guard<std::mutex> lock(mu);
  auto it = tab.find(key);
  if (it != tab.end()) return it->second;
  Tensor t = at::zeros({(int64_t)btc_bn_fuse_ws_bytes()}, like.options().dtype(at::kByte));
  tab.emplace(key, t);
  return t;
}

std::tuple<Tensor, Tensor, Tensor> conv_bn_fwd(const Tensor& features, const Tensor& w, const OptTensor& bias, const Tensor& map_fwd,
                                               const OptTensor& order_fwd, const OptTensor& gamma, const OptTensor& beta, const OptTensor& rm, const OptTensor& rv,
                                               const OptTensor& nbt, bool use_batch, double momentum, double eps, bool relu, const Tensor& ws,
                                               int64_t ws_bytes, int64_t stream) {
  const int64_t cin = w.size(-2), cout = w.size(-1), K = map_fwd.size(1), n_res = map_fwd.size(0);
  if (use_batch && n_res >= 1) {
    // training-mode BatchNorm: its batch statistics come out of the conv kernel's epilogue (btc_conv_bn_relu_fwd, csrc/bn_fuse.h)
    need(features.is_contiguous() && w.is_contiguous() && map_fwd.is_contiguous(), "conv_bn_fwd: contiguous tensors expected");
    need(w.numel() == K * cin * cout && features.size(1) == cin, "conv_bn_fwd: weight does not match the rulebook / features");
    const int32_t* order = order_ptr(order_fwd, n_res, "conv_bn_fwd: the row order does not match the map");
    Tensor x = at::empty({n_res, cout}, features.options());
    Tensor y = at::empty_like(x);
    Tensor stats = at::empty({2, cout}, features.options().dtype(at::kFloat));
    Tensor fw = fuse_ws_of(features, stream);
    float* mean = (float*)stats.data_ptr();
    long long* nb = (nbt.has_value() && nbt->defined()) ? (long long*)nbt->data_ptr() : nullptr;
    int operands = features.scalar_type() == at::kBFloat16 ? BTC_OPERANDS_BF16_ACT : BTC_OPERANDS_F32;
    const void* wp = w.data_ptr();
    Tensor q;
    if (bf16_operands(features, K, cin, cout)) {          // (round 5: the bf16-operand kernel gathers the statistics too)
      q = weights_bf16(w, K, cin, cout, stream);
      operands = BTC_OPERANDS_BF16;
      wp = (const char*)q.data_ptr() + 2 * w.numel();
    } else if (split_operands(features, K, cin, cout, n_res, stream)) {
      q = weights_q(w, K, cin, cout, stream, 3);
      operands = BTC_OPERANDS_F32_SPLIT;
      wp = (const char*)q.data_ptr() + 6 * w.numel();
    }
    chk(btc_conv_bn_relu_fwd_src(operands, features.data_ptr(), (long long)features.size(0), wp, fptr(bias), (const int32_t*)map_fwd.data_ptr(), order, (int)n_res, (int)K,
                             (int)cin, (int)cout, x.data_ptr(), fptr(gamma), fptr(beta), (float*)vptr(rm), (float*)vptr(rv), nb, (float)momentum,
                             (float)eps, (int)relu, y.data_ptr(), mean, mean + cout, ws.data_ptr(), (size_t)ws_bytes, fw.data_ptr(), st(stream)),
        "btc_conv_bn_relu_fwd");
    return std::make_tuple(x, y, stats);
  }
  Tensor x = conv_fwd(features, w, bias, map_fwd, order_fwd, stream);
  auto ys = bn_fwd(x, gamma, beta, rm, rv, nbt, use_batch, momentum, eps, relu, ws, ws_bytes, stream);
  return std::make_tuple(x, std::get<0>(ys), std::get<1>(ys));
}

// dx, dparam (2, C) = dgamma | dbeta
std::tuple<Tensor, Tensor> bn_bwd(const Tensor& x, const Tensor& y, const Tensor& dy, const OptTensor& gamma, const Tensor& stats, bool use_batch,
                                  bool relu, const Tensor& ws, int64_t ws_bytes, int64_t stream) {
  const int64_t N = x.size(0), C = x.size(1);
  need(dy.is_contiguous() && dy.scalar_type() == x.scalar_type(), "bn_bwd: dy must be contiguous and of the activation type");
  Tensor dx = at::empty_like(x);
  Tensor dparam = at::empty({2, C}, x.options().dtype(at::kFloat));
  const float* mean = (const float*)stats.data_ptr();
  float* dgamma = (float*)dparam.data_ptr();
  if (x.scalar_type() == at::kBFloat16)
    chk(btc_bn_relu_bwd_bf16(x.data_ptr(), y.data_ptr(), dy.data_ptr(), (int)N, (int)C, fptr(gamma), mean, mean + C, (int)use_batch, (int)relu,
                             dx.data_ptr(), dgamma, dgamma + C, ws.data_ptr(), (size_t)ws_bytes, st(stream)), "btc_bn_relu_bwd_bf16");
  else
    chk(btc_bn_relu_bwd((const float*)x.data_ptr(), (const float*)y.data_ptr(), (const float*)dy.data_ptr(), (int)N, (int)C, fptr(gamma), mean,
                        mean + C, (int)use_batch, (int)relu, (float*)dx.data_ptr(), dgamma, dgamma + C, ws.data_ptr(), (size_t)ws_bytes, st(stream)),
        "btc_bn_relu_bwd");
  return std::make_tuple(dx, dparam);
}

// fork / join of a side stream around wgrad (one pair of events and one pooled stream per device, reused in stream order)
// a weight gradient whose partial slabs wait for the batched reduction at the side stream's join (btc_wgrad_reduce_multi)
struct SlabJob {
  Tensor ws;       // held until the reduction has been ENQUEUED: a block released earlier could be handed out again while it is still owed a read
  // dW is NOT held: AccumulateGrad adopts a gradient as `.grad` only when nobody else references the tensor (it clones otherwise -- and
  // would clone a buffer that is written later).  A weak reference to its storage says at reduction time whether anybody still has it.
  c10::weak_intrusive_ptr<c10::StorageImpl> dw_storage;
  float* dw;
  long long count;
  int S;
};

struct SideStream {
  hipStream_t side = nullptr;
  hipEvent_t fork = nullptr, join = nullptr;
  c10::hip::HIPStream* c10side = nullptr;
  std::mutex slab_mu;
  std::vector<SlabJob> slabs;             // deferred weight gradients in flight on `side`, not yet reduced
  std::atomic<bool> pending{false};     // deferred mode: wgrads are in flight on the side stream, the join is still owed
  std::atomic<int> queued_task{-1};     // id of the backward pass (autograd graph task) whose end-of-pass callback will pay it;
                                        // a pass that aborted never runs its callback -- the next pass has another id and queues anew
};

// deferred join (set from Python): weight gradients are off the backward pass's critical path -- only dgrad feeds the next
// node -- so the whole chain of wgrads runs on the side stream beside the main chain (BatchNorm backward + dgrad) and is
// joined ONCE: by an autograd-engine callback at the end of backward, or earlier by join_wgrad() (a gradient reducer that
// reads dW in mid-backward).  Everything a deferred wgrad touches is handed to the allocator with recordStream, and a layer
// is deferred only if its weight is a leaf parameter (allow_defer): a dW that another autograd node consumes during
// backward -- the occupancy head's merged weight goes through CatBackward -- must be complete when its node returns.
std::atomic<bool> g_defer_join{false};
// ... and their slab reductions are batched into the join (false: one reduction per layer, as before round 5)
const bool g_batch_reduce = true;

// flags of events that only order streams of one device among themselves (never waited for by the host to read host memory)
unsigned sync_event_flags() {
  return hipEventDisableTiming | hipEventDisableSystemFence;
}

SideStream& side_of(int device) {
  static SideStream tab[64];
  static std::mutex mu;
  SideStream& s = tab[device & 63];
  std::lock_guard<std::mutex> lock(mu);
  if (!s.side) {
    static std::vector<c10::hip::HIPStream> keep;  // keeps the pooled stream objects alive
    keep.reserve(64);
    keep.push_back(c10::hip::getStreamFromPool(false, (c10::DeviceIndex)device));
    s.c10side = &keep.back();
    s.side = keep.back().stream();
    // stream-to-stream ordering on ONE device: a device-scope release is all these events need.  HIP's default is a SYSTEM-scope
    // release at every record -- a write-back + invalidate of the caches (hip_runtime_api.h, hipEventDisableSystemFence: "avoiding
    // the cost of cache writeback and invalidation, and the performance impact of those actions on the execution of following
    // work") -- ~28 times per backward pass here, each one emptying the L2 the gathers of the next kernels live on.
    const unsigned flags = sync_event_flags();
    if (hipEventCreateWithFlags(&s.fork, flags) != hipSuccess || hipEventCreateWithFlags(&s.join, flags) != hipSuccess)
      throw std::runtime_error("hipEventCreateWithFlags failed");
  }
  return s;
}

// every weight gradient of the backward pass so far: ONE reduction launch (per 64 layers) on the side stream instead of one per layer
void reduce_slabs(SideStream& s) {
  std::vector<SlabJob> jobs;
  {
    std::lock_guard<std::mutex> lock(s.slab_mu);
    jobs.swap(s.slabs);
  }
  if (jobs.empty()) return;
  std::vector<const float*> parts;
  std::vector<float*> dws;
  std::vector<int> S;
  std::vector<long long> counts;
  std::vector<c10::intrusive_ptr<c10::StorageImpl>> alive;   // until the launch is enqueued (the storages are registered with the side stream)
  for (size_t i = 0; i < jobs.size(); ++i) {
    auto st = jobs[i].dw_storage.lock();
    if (!st) continue;                     // the gradient was dropped: nobody will read it, and its block may belong to somebody else
    alive.push_back(std::move(st));
    parts.push_back((const float*)jobs[i].ws.data_ptr());
    dws.push_back(jobs[i].dw);
    S.push_back(jobs[i].S);
    counts.push_back(jobs[i].count);
  }
  if (parts.empty()) return;
  chk(btc_wgrad_reduce_multi(parts.data(), dws.data(), S.data(), counts.data(), (int)parts.size(), (void*)s.side), "btc_wgrad_reduce_multi");
}

void join_side(SideStream& s, hipStream_t main) {
  if (!s.pending.load()) return;
  reduce_slabs(s);
  if (hipEventRecord(s.join, s.side) != hipSuccess || hipStreamWaitEvent(main, s.join, 0) != hipSuccess)
    throw std::runtime_error("side-stream join failed");
  s.pending = false;
}

// make the current stream wait for every weight gradient still in flight on the side stream (no-op when nothing is owed)
void join_wgrad() {
  const int dev = (int)c10::hip::current_device();
  join_side(side_of(dev), c10::hip::getCurrentHIPStream().stream());
}

void set_defer_wgrad_join(bool on) { g_defer_join = on; }

// the weight-gradient side stream of `device_index`: a stream the caller has chosen (btcdet_amd/streams.py: one that does not share a hardware
// queue with the streams the step's chains run on) instead of the next one of torch's pool.  Call it while no weight gradient is in flight.
void set_side_stream(int64_t raw_stream, int64_t device_index) {
  SideStream& s = side_of((int)device_index);
  if (s.pending.load()) throw std::runtime_error("set_side_stream: weight gradients are in flight on the current side stream");
  static std::mutex mu;
  std::lock_guard<std::mutex> lock(mu);
  static std::vector<c10::hip::HIPStream> keep;
  keep.reserve(256);
  keep.push_back(c10::hip::getStreamFromExternal(reinterpret_cast<hipStream_t>(raw_stream), (c10::DeviceIndex)device_index));
  s.c10side = &keep.back();
  s.side = keep.back().stream();
}

// din (n_src, Cin), dw (shape of w); either may come back undefined (None) when not needed.  overlap: wgrad runs on a side
// stream beside dgrad (fork / join with events, no host sync); every temporary is released after the join has been enqueued,
// so the caching allocator's stream-ordered reuse stays valid without recordStream.
std::tuple<OptTensor, OptTensor> conv_bwd(const Tensor& features, const Tensor& w, const Tensor& map_fwd, const Tensor& map_bwd,
                                          const OptTensor& order_bwd, const Tensor& grad_out, bool need_din, bool need_dw, bool overlap,
                                          bool allow_defer, int64_t stream) {
  const int64_t cin = w.size(-2), cout = w.size(-1), K = map_fwd.size(1), n_res = map_fwd.size(0), n_src = map_bwd.size(0);
  need(grad_out.is_contiguous() && grad_out.scalar_type() == features.scalar_type(), "conv_bwd: grad must be contiguous and of the activation type");
  const bool bf = features.scalar_type() == at::kBFloat16;
  const int32_t* order = order_ptr(order_bwd, n_src, "conv_bwd: the row order does not match the map");
  // a submanifold rulebook has ONE map: its backward map is the forward map with the offset index mirrored, and the caller hands
  // in the same tensor for both (ops.Rulebook.map_bwd) -- the dgrad kernels then read column K-1-k for offset k
  const bool mirror = map_fwd.numel() > 0 && map_bwd.data_ptr() == map_fwd.data_ptr();
  const int pass_dgrad = mirror ? BTC_PASS_DGRAD_MIRROR : BTC_PASS_DGRAD;
  const int32_t* wg_map_bwd = (const int32_t*)map_bwd.data_ptr();   // (weight gradient: the same pointer twice = "one map, mirrored")
  OptTensor din, dw;
  Tensor ws;
  hipStream_t main = (hipStream_t)st(stream);
  // ... and only when AccumulateGrad will adopt dW as .grad without touching it (an existing .grad means a read-modify-write
  // on the main stream in mid-backward)
  // on the main stream in mid-backward); for a non-leaf weight the caller vouches that its consumer joins first (allow_defer)
  const bool defer = g_defer_join && need_dw && allow_defer && (!w.is_leaf() || !w.grad().defined());
  SideStream* ss = ((overlap && need_din && need_dw) || defer) ? &side_of(features.get_device()) : nullptr;
  void* wstream = st(stream);
  if (ss) {
    if (hipEventRecord(ss->fork, main) != hipSuccess || hipStreamWaitEvent(ss->side, ss->fork, 0) != hipSuccess)
      throw std::runtime_error("side-stream fork failed");
    wstream = (void*)ss->side;
  }
  // Launch order of the two kernels (they run on different streams; the fork recorded above already marks where the side stream
  // may start, so neither waits for the other).  Whichever is enqueued first gets the compute units first.  fp32: dgrad first --
  // it feeds the next backward node, the critical path (6.57 -> 6.42 ms per step).  bf16 operands: the weight gradient (still
  // fp32-accumulated from widened activations) is the long pole of the backward pass, 2-3x the dgrad beside it; started late
  // it leaves an exposed tail at the join (7.4 ms per step dgrad-first, 6.1 wgrad-first).
  const bool wgrad_first = bf;
  // The operand copy of a NON-leaf weight (the zero-padded 34 -> 64-channel one) is a temporary of this call.  It must outlive the
  // allocations of run_wgrad: this layer's weight gradient runs on the side stream BESIDE the dgrad (its fork was recorded before
  // either launch), so a dW / workspace block carved out of the just-released planes would be written while the dgrad still
  // reads them -- non-finite dX now and then, found by the two-ranks-on-one-GPU test.  Released when the function returns: what is
  // allocated after that is used behind this dgrad on its own stream, or behind a later fork on the side stream.
  Tensor q_hold;
  auto run_dgrad = [&]() {
  if (need_din) {
    Tensor d = at::empty({n_src, cin}, features.options());
    if (bf16_operands(grad_out, K, cout, cin)) {
      Tensor q = q_hold = weights_bf16(w, K, cin, cout, stream);
      chk(btc_conv_apply_ordered(pass_dgrad, BTC_OPERANDS_BF16, grad_out.data_ptr(), q.data_ptr(), nullptr, (const int32_t*)map_bwd.data_ptr(),
                                 order, (int)n_src, (int)K, (int)cin, (int)cout, d.data_ptr(), st(stream)), "btc_conv_apply_ordered (dgrad, bf16 operands)");
    } else if (split_operands(grad_out, K, cout, cin, n_src, stream)) {
      Tensor q = q_hold = weights_q(w, K, cin, cout, stream, 3);
      chk(btc_conv_apply_src(pass_dgrad, BTC_OPERANDS_F32_SPLIT, grad_out.data_ptr(), (long long)grad_out.size(0), q.data_ptr(), nullptr, (const int32_t*)map_bwd.data_ptr(),
                                 order, (int)n_src, (int)K, (int)cin, (int)cout, d.data_ptr(), st(stream)), "btc_conv_apply_src (dgrad, split operands)");
    } else
      chk(btc_conv_apply_ordered(pass_dgrad, bf ? BTC_OPERANDS_BF16_ACT : BTC_OPERANDS_F32, grad_out.data_ptr(), w.data_ptr(), nullptr,
                                 (const int32_t*)map_bwd.data_ptr(), order, (int)n_src, (int)K, (int)cin, (int)cout, d.data_ptr(), st(stream)),
          "btc_conv_apply_ordered (dgrad)");
    din = d;
  }
  };
  auto run_wgrad = [&]() {
  if (need_dw) {
    Tensor g = at::empty(w.sizes(), w.options());
    const size_t ws_bytes = btc_conv_wgrad_ws_bytes((int)n_res, (int)K, (int)cin, (int)cout, (int)n_src);
    ws = at::empty({(int64_t)(ws_bytes > 256 ? ws_bytes : 256)}, features.options().dtype(at::kByte));
    if (defer && g_batch_reduce) {
      // deferred: nobody reads dW before the side stream's join, so the partial slabs stay in `ws` and the join adds up every layer's in
      // one launch (28 reduction launches per step -> 2)
      int n_slabs = 0;
      chk(btc_conv_wgrad_slabs((int)bf, features.data_ptr(), grad_out.data_ptr(), (const int32_t*)map_fwd.data_ptr(), (int)n_res, wg_map_bwd, (int)n_src,
                               nullptr, nullptr, (int)K, (int)cin, (int)cout, (float*)g.data_ptr(), ws.data_ptr(), ws_bytes, &n_slabs, wstream),
          "btc_conv_wgrad_slabs");
      if (n_slabs > 0) {
        std::lock_guard<std::mutex> lock(ss->slab_mu);
        ss->slabs.push_back(SlabJob{ws, g.storage().getWeakStorageImpl(), (float*)g.data_ptr(), (long long)g.numel(), n_slabs});
      }
    } else if (bf)
      chk(btc_conv_wgrad_bf16(features.data_ptr(), grad_out.data_ptr(), (const int32_t*)map_fwd.data_ptr(), (int)n_res,
                              wg_map_bwd, (int)n_src, (int)K, (int)cin, (int)cout, (float*)g.data_ptr(), ws.data_ptr(),
                              ws_bytes, wstream), "btc_conv_wgrad_bf16");
    else
      chk(btc_conv_wgrad((const float*)features.data_ptr(), (const float*)grad_out.data_ptr(), (const int32_t*)map_fwd.data_ptr(), (int)n_res,
                         wg_map_bwd, (int)n_src, (int)K, (int)cin, (int)cout, (float*)g.data_ptr(), ws.data_ptr(),
                         ws_bytes, wstream), "btc_conv_wgrad");
    dw = g;
  }
  };
  if (wgrad_first) { run_wgrad(); run_dgrad(); } else { run_dgrad(); run_wgrad(); }
  if (ss && !defer) {  // join: dW (and the release of ws / grad_out by the caller) is ordered after wgrad on the main stream
    ss->pending = true;
    join_side(*ss, main);
  } else if (ss) {     // deferred: the temporaries outlive this call on the side stream; one join at the end of backward
    ss->pending = true;
    c10::hip::HIPCachingAllocator::recordStream(grad_out.storage().data_ptr(), *ss->c10side);
    c10::hip::HIPCachingAllocator::recordStream(ws.storage().data_ptr(), *ss->c10side);
    c10::hip::HIPCachingAllocator::recordStream(features.storage().data_ptr(), *ss->c10side);
    c10::hip::HIPCachingAllocator::recordStream(dw->storage().data_ptr(), *ss->c10side);
    // the rulebook is kept alive by this node's saved tensors only: once the node has run it may be freed on the main stream
    c10::hip::HIPCachingAllocator::recordStream(map_fwd.storage().data_ptr(), *ss->c10side);
    c10::hip::HIPCachingAllocator::recordStream(map_bwd.storage().data_ptr(), *ss->c10side);
    const int task = torch::autograd::get_current_graph_task_id();
    if (task < 0) {                        // not inside a backward pass: nobody would pay the join later
      join_side(*ss, main);
    } else if (ss->queued_task.load() != task) {
      SideStream* sp = ss;
      try {
        torch::autograd::Engine::get_default_engine().queue_callback([sp]() {
          sp->queued_task = -1;
          join_side(*sp, c10::hip::getCurrentHIPStream().stream());
        });
        ss->queued_task = task;
      } catch (const std::exception&) {
        join_side(*ss, main);
      }
    }
  }
  return std::make_tuple(din, dw);
}

// row-order hints (csrc/row_order.hip) of several neighbour maps from ONE launch: order[j] is a slice of one buffer.  The apply
// kernels tile the rows in that order (rows with the same offsets share a tile); results do not depend on it.
std::vector<Tensor> row_orders_keyed(const std::vector<Tensor>& maps, const std::vector<Tensor>& keys, int64_t stream);
std::vector<Tensor> row_orders(const std::vector<Tensor>& maps, int64_t stream) { return row_orders_keyed(maps, {}, stream); }

// keys: nothing, or per map an undefined tensor / the (n) first-offset keys the rulebook fill wrote (btc_row_orders_keyed)
std::vector<Tensor> row_orders_keyed(const std::vector<Tensor>& maps, const std::vector<Tensor>& keys, int64_t stream) {
  std::vector<Tensor> out(maps.size());
  need(keys.empty() || keys.size() == maps.size(), "row_orders: one key tensor (or an undefined one) per map");
  for (size_t base = 0; base < maps.size(); base += BTC_ROW_ORDER_MAX_MAPS) {
    const size_t m = std::min(maps.size() - base, (size_t)BTC_ROW_ORDER_MAX_MAPS);
    std::vector<const int32_t*> ptrs(m), kptrs(m, nullptr);
    std::vector<int32_t> ns(m), ks(m);
    int64_t total = 0;
    for (size_t j = 0; j < m; ++j) {
      const Tensor& t = maps[base + j];
      need(t.dim() == 2 && t.scalar_type() == at::kInt && t.is_contiguous(), "row_orders: (n, K) int32 contiguous maps expected");
      ptrs[j] = (const int32_t*)t.data_ptr();
      ns[j] = (int32_t)t.size(0);
      ks[j] = (int32_t)t.size(1);
      total += t.size(0);
      if (!keys.empty() && keys[base + j].defined()) {
        const Tensor& k = keys[base + j];
        need(k.scalar_type() == at::kInt && k.is_contiguous() && k.numel() == t.size(0), "row_orders: keys must be (n) int32 contiguous");
        kptrs[j] = (const int32_t*)k.data_ptr();
      }
    }
    Tensor order = at::empty({total > 0 ? total : 1}, maps[base].options());
    chk(btc_row_orders_keyed(ptrs.data(), kptrs.data(), ns.data(), ks.data(), (int)m, (int32_t*)order.data_ptr(), st(stream)), "btc_row_orders_keyed");
    int64_t off = 0;
    for (size_t j = 0; j < m; ++j) {
      out[base + j] = order.narrow(0, off, ns[j]);
      off += ns[j];
    }
  }
  return out;
}

// submanifold rulebook: returns nbr_out (n, K); nbr_in is its mirror image and is not built (BTC_PASS_DGRAD_MIRROR).  p_* are the
// addresses of int32[3] host arrays.
Tensor rulebook_subm(const Tensor& indices, int64_t batch, int64_t p_in, int64_t p_k, int64_t p_d, int64_t K, int64_t stream) {
  const int64_t n = indices.size(0);
  Tensor nbr = at::empty({n, K}, indices.options());
  const size_t ws_bytes = btc_rulebook_subm_ws_bytes((int)n);
  Tensor ws = at::empty({(int64_t)(ws_bytes > 256 ? ws_bytes : 256)}, indices.options().dtype(at::kByte));
  chk(btc_rulebook_subm((const int32_t*)indices.data_ptr(), (int)n, (int)batch, ip(p_in), ip(p_k), ip(p_d), (int32_t*)nbr.data_ptr(), nullptr,
                        ws.data_ptr(), ws_bytes, st(stream)), "btc_rulebook_subm");
  return nbr;
}

// strided / transposed rulebook, synchronous: count, one blocking 4-byte read-back, fill
std::tuple<Tensor, Tensor, Tensor, Tensor, Tensor> rulebook_conv(const Tensor& indices, int64_t batch, int64_t p_in, int64_t p_out, int64_t p_k, int64_t p_s,
                                                 int64_t p_p, int64_t p_d, int64_t mode, int64_t K, int64_t ws_bytes, int64_t stream) {
  const int64_t n = indices.size(0);
  Tensor ws = at::empty({ws_bytes > 256 ? ws_bytes : 256}, indices.options().dtype(at::kByte));
  Tensor d_n = at::empty({1}, indices.options());
  chk(btc_rulebook_conv_count((const int32_t*)indices.data_ptr(), (int)n, (int)batch, ip(p_in), ip(p_out), ip(p_k), ip(p_s), ip(p_p), ip(p_d),
                              (int)mode, (int32_t*)d_n.data_ptr(), ws.data_ptr(), (size_t)ws_bytes, st(stream)), "btc_rulebook_conv_count");
  const int64_t n_out = d_n.item<int32_t>();
  Tensor out_indices = at::empty({n_out, 4}, indices.options());
  Tensor nbr_out = at::empty({n_out, K}, indices.options());
  Tensor nbr_in = at::empty({n, K}, indices.options());
  chk(btc_rulebook_conv_fill((const int32_t*)indices.data_ptr(), (int)n, (int)batch, ip(p_in), ip(p_out), ip(p_k), ip(p_s), ip(p_p), ip(p_d),
                             (int)mode, (int)n_out, (int32_t*)out_indices.data_ptr(), (int32_t*)nbr_out.data_ptr(), (int32_t*)nbr_in.data_ptr(),
                             ws.data_ptr(), (size_t)ws_bytes, st(stream)), "btc_rulebook_conv_fill");
  auto ord = row_orders({nbr_out, nbr_in}, stream);
  return std::make_tuple(out_indices, nbr_out, nbr_in, ord[0], ord[1]);
}

inline int64_t current_stream() { return reinterpret_cast<int64_t>(c10::hip::getCurrentHIPStream().stream()); }

// ---- strided / transposed rulebook in two halves (ops.py LOOKAHEAD): the count half runs on a side stream as soon as the
// input level exists and writes n_out straight into pinned host memory; the fill half runs in the consumer's forward and
// waits -- on the host -- for the count's event only, never for the main stream.
// (BtcHotPath.prepare may run on a worker thread while the training thread is in forward: slots are handed out atomically, every
// slot has its own fork / done events, and the per-device state is created under a mutex)
struct RbLookahead {
  hipStream_t side = nullptr;
  hipEvent_t fork[64] = {};
  hipEvent_t done[64] = {};
  int32_t* host_n = nullptr;  // 64 pinned ints, device-visible
  std::atomic<unsigned> next{0};
};

RbLookahead& rb_of(int device) {
  static RbLookahead tab[64];
  static std::mutex mu;
  RbLookahead& r = tab[device & 63];
  std::lock_guard<std::mutex> lock(mu);
  if (!r.side) {
    static std::vector<c10::hip::HIPStream> keep;
    keep.reserve(64);
    keep.push_back(c10::hip::getStreamFromPool(false, (c10::DeviceIndex)device));
    hipStream_t side = keep.back().stream();
    bool ok = hipHostMalloc((void**)&r.host_n, 64 * sizeof(int32_t), hipHostMallocDefault) == hipSuccess;
    for (int i = 0; ok && i < 64; ++i)
      ok = hipEventCreateWithFlags(&r.done[i], hipEventDisableTiming) == hipSuccess &&
           hipEventCreateWithFlags(&r.fork[i], sync_event_flags()) == hipSuccess;
    if (!ok) throw std::runtime_error("rulebook lookahead: event / pinned-memory setup failed");
    r.side = side;
  }
  return r;
}

struct PendingRb {
  Tensor indices, ws;
  int64_t batch, p_in, p_out, p_k, p_s, p_p, p_d, mode, K, ws_bytes;
  hipEvent_t done;
  volatile int32_t* host_n;
};

std::shared_ptr<PendingRb> rulebook_conv_start(const Tensor& indices, int64_t batch, int64_t p_in, int64_t p_out, int64_t p_k, int64_t p_s,
                                               int64_t p_p, int64_t p_d, int64_t mode, int64_t K, int64_t ws_bytes) {
  auto p = std::make_shared<PendingRb>();
  p->indices = indices; p->batch = batch; p->p_in = p_in; p->p_out = p_out; p->p_k = p_k; p->p_s = p_s; p->p_p = p_p; p->p_d = p_d;
  p->mode = mode; p->K = K; p->ws_bytes = ws_bytes;
  RbLookahead& r = rb_of(indices.get_device());
  const int slot = (int)(r.next.fetch_add(1) & 63u);
  p->done = r.done[slot];
  p->host_n = r.host_n + slot;
  p->ws = at::empty({ws_bytes > 256 ? ws_bytes : 256}, indices.options().dtype(at::kByte));
  hipStream_t main = (hipStream_t)st(current_stream());
  // fork after the allocation: the side stream is ordered behind every earlier user of that block and behind the kernels
  // that produce `indices`
  if (hipEventRecord(r.fork[slot], main) != hipSuccess || hipStreamWaitEvent(r.side, r.fork[slot], 0) != hipSuccess)
    throw std::runtime_error("rulebook lookahead: fork failed");
  chk(btc_rulebook_conv_count((const int32_t*)indices.data_ptr(), (int)indices.size(0), (int)batch, ip(p_in), ip(p_out), ip(p_k), ip(p_s),
                              ip(p_p), ip(p_d), (int)mode, (int32_t*)p->host_n, p->ws.data_ptr(), (size_t)ws_bytes, (void*)r.side),
      "btc_rulebook_conv_count");
  if (hipEventRecord(p->done, r.side) != hipSuccess) throw std::runtime_error("rulebook lookahead: event record failed");
  return p;
}

std::tuple<Tensor, Tensor, Tensor, Tensor, Tensor> rulebook_conv_finish(const std::shared_ptr<PendingRb>& p) {
  if (hipEventSynchronize(p->done) != hipSuccess) throw std::runtime_error("rulebook lookahead: event synchronize failed");
  const int64_t n_out = *p->host_n, n = p->indices.size(0), K = p->K;
  hipStream_t main = (hipStream_t)st(current_stream());
  if (hipStreamWaitEvent(main, p->done, 0) != hipSuccess) throw std::runtime_error("rulebook lookahead: join failed");
  Tensor out_indices = at::empty({n_out, 4}, p->indices.options());
  Tensor nbr_out = at::empty({n_out, K}, p->indices.options());
  Tensor nbr_in = at::empty({n, K}, p->indices.options());
  chk(btc_rulebook_conv_fill((const int32_t*)p->indices.data_ptr(), (int)n, (int)p->batch, ip(p->p_in), ip(p->p_out), ip(p->p_k), ip(p->p_s),
                             ip(p->p_p), ip(p->p_d), (int)p->mode, (int)n_out, (int32_t*)out_indices.data_ptr(), (int32_t*)nbr_out.data_ptr(),
                             (int32_t*)nbr_in.data_ptr(), p->ws.data_ptr(), (size_t)p->ws_bytes, (void*)main), "btc_rulebook_conv_fill");
  auto ord = row_orders({nbr_out, nbr_in}, (int64_t)(intptr_t)main);
  return std::make_tuple(out_indices, nbr_out, nbr_in, ord[0], ord[1]);
}

// conv -> BatchNorm1d (-> ReLU) as a C++ autograd node: the same three launches as ops.SparseConvBNReLUFunction without the
// Python Function.apply / ctx bookkeeping per layer (the forward pass is bound by the host's launch rate).  wgrad runs on the
// backward stream right before dgrad (no side-stream overlap here; Python keeps that variant for the 20 K - 100 K-row layers).
struct ConvBNReLUNode : public torch::autograd::Function<ConvBNReLUNode> {
  static Tensor forward(torch::autograd::AutogradContext* ctx, const Tensor& features, const Tensor& weight, const OptTensor& bias,
                        const Tensor& map_fwd, const Tensor& map_bwd, const OptTensor& order_fwd, const OptTensor& order_bwd,
                        const OptTensor& gamma, const OptTensor& beta, const OptTensor& rm, const OptTensor& rv, const OptTensor& nbt,
                        bool use_batch, double momentum, double eps, bool relu, const Tensor& ws, int64_t ws_bytes, bool overlap,
                        bool allow_defer) {
    const int64_t stream = current_stream();
    auto r = conv_bn_fwd(features, weight, bias, map_fwd, order_fwd, gamma, beta, rm, rv, nbt, use_batch, momentum, eps, relu, ws, ws_bytes, stream);
    const Tensor g = (gamma.has_value() && gamma->defined()) ? *gamma : Tensor();
    const Tensor ob = (order_bwd.has_value() && order_bwd->defined()) ? *order_bwd : Tensor();
    ctx->save_for_backward({features, weight, map_fwd, map_bwd, std::get<0>(r), std::get<1>(r), g, std::get<2>(r), ws, ob});
    ctx->saved_data["use_batch"] = use_batch;
    ctx->saved_data["relu"] = relu;
    ctx->saved_data["ws_bytes"] = ws_bytes;
    ctx->saved_data["overlap"] = overlap;
    ctx->saved_data["allow_defer"] = allow_defer;
    ctx->saved_data["has_bias"] = bias.has_value() && bias->defined();
    return std::get<1>(r);
  }

  static torch::autograd::variable_list backward(torch::autograd::AutogradContext* ctx, torch::autograd::variable_list grads) {
    const auto saved = ctx->get_saved_variables();
    const Tensor &features = saved[0], &w = saved[1], &map_fwd = saved[2], &map_bwd = saved[3], &x = saved[4], &y = saved[5], &gamma = saved[6],
                 &stats = saved[7], &ws = saved[8];
    const bool use_batch = ctx->saved_data["use_batch"].toBool(), relu = ctx->saved_data["relu"].toBool();
    const int64_t ws_bytes = ctx->saved_data["ws_bytes"].toInt(), stream = current_stream();
    Tensor dy = grads[0];
    if (dy.scalar_type() != x.scalar_type()) dy = dy.to(x.scalar_type());
    dy = dy.contiguous();
    OptTensor og;
    if (gamma.defined()) og = gamma;
    auto b = bn_bwd(x, y, dy, og, stats, use_batch, relu, ws, ws_bytes, stream);
    const Tensor& dx = std::get<0>(b);
    const Tensor& dparam = std::get<1>(b);
    OptTensor order_bwd;
    if (saved[9].defined()) order_bwd = saved[9];
    auto cb = conv_bwd(features, w, map_fwd, map_bwd, order_bwd, dx, ctx->needs_input_grad(0), ctx->needs_input_grad(1),
                       ctx->saved_data["overlap"].toBool(), ctx->saved_data["allow_defer"].toBool(), stream);
    Tensor din = std::get<0>(cb).has_value() ? *std::get<0>(cb) : Tensor();
    Tensor dw = std::get<1>(cb).has_value() ? *std::get<1>(cb) : Tensor();
    Tensor db;
    if (ctx->saved_data["has_bias"].toBool() && ctx->needs_input_grad(2)) {  // the same column-sum kernel as ops._bias_grad
      db = at::empty({dx.size(1)}, dx.options().dtype(at::kFloat));
      if (dx.scalar_type() == at::kBFloat16)
        chk(btc_col_sum_bf16(dx.data_ptr(), (int)dx.size(0), (int)dx.size(1), (float*)db.data_ptr(), ws.data_ptr(), (size_t)ws_bytes, st(stream)),
            "btc_col_sum_bf16");
      else
        chk(btc_col_sum((const float*)dx.data_ptr(), (int)dx.size(0), (int)dx.size(1), (float*)db.data_ptr(), ws.data_ptr(), (size_t)ws_bytes,
                        st(stream)), "btc_col_sum");
    }
    Tensor dgamma, dbeta;
    if (gamma.defined()) {
      dgamma = dparam[0];
      dbeta = dparam[1];
    }
    return {din, dw, db, Tensor(), Tensor(), Tensor(), Tensor(), dgamma, dbeta, Tensor(), Tensor(), Tensor(), Tensor(), Tensor(), Tensor(), Tensor(), Tensor(), Tensor(),
            Tensor(), Tensor()};
  }
};

Tensor conv_bn_relu(const Tensor& features, const Tensor& weight, const OptTensor& bias, const Tensor& map_fwd, const Tensor& map_bwd,
                    const OptTensor& order_fwd, const OptTensor& order_bwd, const OptTensor& gamma, const OptTensor& beta, const OptTensor& rm, const OptTensor& rv, const OptTensor& nbt, bool use_batch,
                    double momentum, double eps, bool relu, const Tensor& ws, int64_t ws_bytes, bool overlap, bool allow_defer) {
  return ConvBNReLUNode::apply(features, weight, bias, map_fwd, map_bwd, order_fwd, order_bwd, gamma, beta, rm, rv, nbt, use_batch, momentum, eps, relu, ws, ws_bytes,
                               overlap, allow_defer);
}

// every rulebook of a chain of sparse layers from the input coordinates alone (BtcHotPath.prepare runs this in a worker thread
// beside the backward pass: one call, GIL released, instead of a Python walk over the layers that competes with the autograd
// thread for the GIL).  kind: 0 = build submanifold, 1 = build strided / transposed, 2 = inverse of layer ref (continue on ITS
// input level), 3 = reuse layer ref's rulebook (continue on its output level).  Returns per built layer
// {in_indices, out_indices, nbr_out, nbr_in} (+ {order_out, order_in}, the row-order hints, for strided / transposed layers), an
// empty list for the others.
// Two phases of the C ABI around ONE read-back for the whole chain (btc_chain_levels / btc_chain_maps, csrc/rulebook.hip): the
// levels are built on the device with their row counts left there and their rows written into capacity-sized buffers (16
// bytes a row -- HBM is 288 GB, the untouched tail costs nothing); the counts come back together; the maps are sized exactly.
// The walk in two halves.  start: phase A (the levels) and the read-back of the row counts, either on the current stream with
// a blocking read-back, or -- `side` -- forked onto a side stream with the counts copied into pinned host memory, so that the
// caller can keep launching on its own stream (the detection backbone runs its first stage, which only needs the level-0
// submanifold rulebook, while the strided levels are being built).  finish: waits for the counts (host + current stream), sizes
// the maps exactly and runs phase B (every map of the chain in one multi-job launch) and the row-order launch on the current
// stream.  skip_maps: built submanifold layers whose maps the caller already has (no buffers, no fill job, an empty list back).
struct WalkSide {
  hipStream_t side = nullptr;
  c10::hip::HIPStream* c10side = nullptr;
  hipEvent_t fork[8] = {}, done[8] = {};
  int32_t* host_counts = nullptr;   // 8 slots x BTC_CHAIN_MAX_LAYERS pinned ints
  std::atomic<unsigned> next{0};
};

WalkSide& walk_of(int device) {
  static WalkSide tab[64];
  static std::mutex mu;
  WalkSide& w = tab[device & 63];
  std::lock_guard<std::mutex> lock(mu);
  if (!w.side) {
    static std::vector<c10::hip::HIPStream> keep;
    keep.reserve(64);
    keep.push_back(c10::hip::getStreamFromPool(false, (c10::DeviceIndex)device));
    bool ok = hipHostMalloc((void**)&w.host_counts, 8 * BTC_CHAIN_MAX_LAYERS * sizeof(int32_t), hipHostMallocDefault) == hipSuccess;
    for (int i = 0; ok && i < 8; ++i)
      ok = hipEventCreateWithFlags(&w.fork[i], sync_event_flags()) == hipSuccess &&
           hipEventCreateWithFlags(&w.done[i], hipEventDisableTiming) == hipSuccess;
    if (!ok) throw std::runtime_error("geometry walk: event / pinned-memory setup failed");
    w.c10side = &keep.back();
    w.side = keep.back().stream();
  }
  return w;
}

struct PendingWalk {
  Tensor indices, ws, d_counts, h_counts;
  std::vector<Tensor> out_idx;
  std::vector<BtcChainLayer> layers;
  std::vector<int64_t> kind, K, ref;
  size_t wsb = 0;
  int n0 = 0;
  int64_t batch = 0;
  bool side = false;
  hipEvent_t done = nullptr;
  const int32_t* counts = nullptr;   // host: valid after the wait in finish
};

std::shared_ptr<PendingWalk> geometry_walk_start(const Tensor& indices, int64_t batch, const std::vector<int64_t>& kind,
                                                 const std::vector<int64_t>& a_in, const std::vector<int64_t>& a_out,
                                                 const std::vector<int64_t>& a_k, const std::vector<int64_t>& a_s,
                                                 const std::vector<int64_t>& a_p, const std::vector<int64_t>& a_d,
                                                 const std::vector<int64_t>& mode, const std::vector<int64_t>& K,
                                                 const std::vector<int64_t>& ws_bytes, const std::vector<int64_t>& ref, int64_t side_mode) {
  // side_mode: 0 = current stream, blocking read-back; 1 = forked onto the walk's side stream, the counts copied to pinned memory
  // asynchronously.  (Round 4 measured three more -- the asynchronous copy on the current stream, the read-back deferred to finish(), a
  // hand-over between threads through a copy stream: none faster, DESIGN.md section 5 -- and they are gone.)
  need(side_mode == 0 || side_mode == 1, "geometry_walk_start: side_mode must be 0 or 1");
  const bool side = side_mode == 1;
  const size_t n = kind.size();
  need(a_in.size() == n && a_out.size() == n && a_k.size() == n && a_s.size() == n && a_p.size() == n && a_d.size() == n && mode.size() == n &&
           K.size() == n && ws_bytes.size() == n && ref.size() == n, "geometry_walk: per-layer argument lists differ in length");
  need(n >= 1 && n <= BTC_CHAIN_MAX_LAYERS, "geometry_walk: too many layers for one chain");
  auto p = std::make_shared<PendingWalk>();
  p->indices = indices; p->batch = batch; p->kind = kind; p->K = K; p->ref = ref; p->side = side;
  p->n0 = (int)indices.size(0);
  p->layers.resize(n);
  for (size_t i = 0; i < n; ++i) {
    BtcChainLayer& l = p->layers[i];
    l.kind = (int32_t)kind[i];
    l.ref = (int32_t)ref[i];
    l.mode = (int32_t)mode[i];
    for (int j = 0; j < 3; ++j) {
      const bool built = kind[i] < 2;
      l.in_shape[j] = built ? ip(a_in[i])[j] : 1;
      l.out_shape[j] = built ? (kind[i] == 1 ? ip(a_out[i])[j] : ip(a_in[i])[j]) : 1;
      l.k[j] = built ? ip(a_k[i])[j] : 1;
      l.s[j] = (built && kind[i] == 1) ? ip(a_s[i])[j] : 1;
      l.p[j] = (built && kind[i] == 1) ? ip(a_p[i])[j] : 0;
      l.d[j] = built ? ip(a_d[i])[j] : 1;
    }
  }
  std::vector<int64_t> cap(n, 0);
  chk(btc_chain_caps(p->layers.data(), (int)n, (int)batch, p->n0, cap.data()), "btc_chain_caps");
  p->wsb = btc_chain_ws_bytes(p->layers.data(), (int)n, (int)batch, p->n0);
  need(p->wsb > 0, "geometry_walk: btc_chain_ws_bytes failed");
  p->ws = at::empty({(int64_t)p->wsb}, indices.options().dtype(at::kByte));
  p->d_counts = at::empty({(int64_t)n}, indices.options());   // (every entry that is read -- the strided layers' -- is written by its level's scan)
  p->out_idx.resize(n);
  std::vector<int32_t*> p_out_idx(n, nullptr);
  for (size_t i = 0; i < n; ++i)
    if (kind[i] == 1) {
      p->out_idx[i] = at::empty({cap[i], 4}, indices.options());
      p_out_idx[i] = (int32_t*)p->out_idx[i].data_ptr();
    }
  hipStream_t main = (hipStream_t)st(current_stream());
  hipStream_t run = main;
  WalkSide* w = nullptr;
  int slot = 0;
  if (side) {
    w = &walk_of(indices.get_device());
    slot = (int)(w->next.fetch_add(1) & 7u);
  }
  if (side) {
    run = w->side;
    // fork after the allocations and the zero fill of d_counts: the side stream is ordered behind them and behind `indices`
    if (hipEventRecord(w->fork[slot], main) != hipSuccess || hipStreamWaitEvent(run, w->fork[slot], 0) != hipSuccess)
      throw std::runtime_error("geometry walk: fork failed");
    for (const Tensor* t : {&p->ws, &p->d_counts, &p->indices}) c10::hip::HIPCachingAllocator::recordStream(t->storage().data_ptr(), *w->c10side);
    for (const Tensor& t : p->out_idx)
      if (t.defined()) c10::hip::HIPCachingAllocator::recordStream(t.storage().data_ptr(), *w->c10side);
  }
  chk(btc_chain_levels((const int32_t*)indices.data_ptr(), p->n0, (int)batch, p->layers.data(), (int)n, p_out_idx.data(), cap.data(),
                       (int32_t*)p->d_counts.data_ptr(), p->ws.data_ptr(), p->wsb, (void*)run), "btc_chain_levels");
  if (side) {
    int32_t* host = w->host_counts + slot * BTC_CHAIN_MAX_LAYERS;
    if (hipMemcpyAsync(host, p->d_counts.data_ptr(), n * sizeof(int32_t), hipMemcpyDeviceToHost, run) != hipSuccess ||
        hipEventRecord(w->done[slot], run) != hipSuccess)
      throw std::runtime_error("geometry walk: read-back enqueue failed");
    p->done = w->done[slot];
    p->counts = host;
  } else {
    p->h_counts = p->d_counts.to(at::kCPU);  // the one read-back of the chain (current stream only)
    p->counts = (const int32_t*)p->h_counts.data_ptr();
  }
  return p;
}

std::vector<std::vector<Tensor>> geometry_walk_finish(const std::shared_ptr<PendingWalk>& p, const std::vector<int64_t>& skip_maps) {
  const size_t n = p->kind.size();
  const std::vector<int64_t>&kind = p->kind, &K = p->K, &ref = p->ref;
  const Tensor& indices = p->indices;
  const int64_t stream = current_stream();
  if (p->side) {
    if (hipEventSynchronize(p->done) != hipSuccess || hipStreamWaitEvent((hipStream_t)st(stream), p->done, 0) != hipSuccess)
      throw std::runtime_error("geometry walk: join failed");
    // finish() may run on another stream than start() did: the buffers start() allocated are used by the fill below on THIS stream
    const c10::hip::HIPStream cur = c10::hip::getCurrentHIPStream();
    for (const Tensor* t : {&p->ws, &p->d_counts, &p->indices}) c10::hip::HIPCachingAllocator::recordStream(t->storage().data_ptr(), cur);
    for (const Tensor& t : p->out_idx)
      if (t.defined()) c10::hip::HIPCachingAllocator::recordStream(t.storage().data_ptr(), cur);
  }
  int32_t hc_copy[BTC_CHAIN_MAX_LAYERS];
  for (size_t i = 0; i < n; ++i) hc_copy[i] = p->counts[i];   // the pinned slot is recycled 8 walks later
  const int32_t* hc = hc_copy;
  std::vector<char> skip(n, 0);
  for (int64_t i : skip_maps) {
    need(i >= 0 && (size_t)i < n && kind[i] == 0, "geometry_walk: only built submanifold layers can skip their maps");
    skip[i] = 1;
  }
  // levels as the walk sees them
  std::vector<Tensor> level_in(n), level_out(n);
  std::vector<int32_t*> p_out_idx(n, nullptr), p_nbr_out(n, nullptr), p_nbr_in(n, nullptr);
  Tensor cur = indices;
  int64_t strided_elems = 0, other_elems = 0;
  for (size_t i = 0; i < n; ++i) {
    if (kind[i] == 0) {
      level_in[i] = level_out[i] = cur;
      if (!skip[i]) other_elems += cur.size(0) * K[i];   // one map: nbr_in is its mirror image (BTC_PASS_DGRAD_MIRROR)
    } else if (kind[i] == 1) {
      level_in[i] = cur;
      p_out_idx[i] = (int32_t*)p->out_idx[i].data_ptr();
      cur = p->out_idx[i].narrow(0, 0, hc[i]);
      level_out[i] = cur;
      strided_elems += (int64_t)hc[i] * K[i];
      other_elems += level_in[i].size(0) * K[i];
    } else {
      need(ref[i] >= 0 && (size_t)ref[i] < i, "geometry_walk: bad layer reference");
      cur = kind[i] == 2 ? level_in[ref[i]] : level_out[ref[i]];
      level_in[i] = kind[i] == 2 ? level_out[ref[i]] : level_in[ref[i]];
      level_out[i] = cur;
    }
  }
  // the strided layers' nbr_out in ONE buffer (one -1 fill for those that need it), everything else in another
  Tensor buf_s = at::empty({strided_elems > 0 ? strided_elems : 1}, indices.options());
  Tensor buf_o = at::empty({other_elems > 0 ? other_elems : 1}, indices.options());
  int64_t off_s = 0, off_o = 0;
  std::vector<std::vector<Tensor>> out(n);
  // row-order hints of the strided / transposed layers' maps (both directions), one launch for the chain: these are the maps
  // whose 16-row tiles are mostly empty in coordinate order (csrc/row_order.hip); order_mode 2 would order the SubM maps too (measured:
  // within noise at KITTI sizes), 0 none.  The sort keys (first present offset of every row) come out of the fill
  // itself (btc_chain_maps first_out / first_in).
  constexpr int order_mode = 1;
  auto wants_order = [&](size_t i) {
    return !skip[i] && K[i] <= 64 && ((kind[i] == 1 && order_mode >= 1) || (kind[i] == 0 && order_mode >= 2));
  };
  int64_t key_elems = 0;
  for (size_t i = 0; i < n; ++i)
    if (kind[i] == 1 && wants_order(i)) key_elems += level_in[i].size(0) + level_out[i].size(0);
  Tensor buf_k = at::empty({key_elems > 0 ? key_elems : 1}, indices.options());
  int64_t off_k = 0;
  std::vector<int32_t*> p_first_out(n, nullptr), p_first_in(n, nullptr);
  std::vector<Tensor> key_out(n), key_in(n);
  for (size_t i = 0; i < n; ++i) {
    if (kind[i] > 1 || skip[i]) continue;
    const int64_t rows_in = level_in[i].size(0), rows_out = level_out[i].size(0);
    Tensor nbr_out, nbr_in;
    if (kind[i] == 1) {
      nbr_out = buf_s.narrow(0, off_s, rows_out * K[i]).view({rows_out, K[i]});
      off_s += rows_out * K[i];
    } else {
      nbr_out = buf_o.narrow(0, off_o, rows_out * K[i]).view({rows_out, K[i]});
      off_o += rows_out * K[i];
    }
    p_nbr_out[i] = (int32_t*)nbr_out.data_ptr();
    if (kind[i] == 1) {
      nbr_in = buf_o.narrow(0, off_o, rows_in * K[i]).view({rows_in, K[i]});
      off_o += rows_in * K[i];
      p_nbr_in[i] = (int32_t*)nbr_in.data_ptr();
      if (wants_order(i)) {
        key_in[i] = buf_k.narrow(0, off_k, rows_in);
        off_k += rows_in;
        p_first_in[i] = (int32_t*)key_in[i].data_ptr();
        key_out[i] = buf_k.narrow(0, off_k, rows_out);
        off_k += rows_out;
        p_first_out[i] = (int32_t*)key_out[i].data_ptr();
      }
    } else {
      nbr_in = nbr_out;   // the SAME tensor: conv_bwd recognises the submanifold rulebook by that and reads it mirrored
    }
    out[i] = {level_in[i], level_out[i], nbr_out, nbr_in};
  }
  chk(btc_chain_maps((const int32_t*)indices.data_ptr(), p->n0, (int)p->batch, p->layers.data(), (int)n, hc, p_out_idx.data(), p_nbr_out.data(),
                     p_nbr_in.data(), p_first_out.data(), p_first_in.data(), p->ws.data_ptr(), p->wsb, st(stream)), "btc_chain_maps");
  std::vector<Tensor> to_order, to_order_keys;
  for (size_t i = 0; i < n; ++i)
    if (wants_order(i)) {
      to_order.push_back(out[i][2]);
      to_order_keys.push_back(key_out[i]);
      to_order.push_back(out[i][3]);
      to_order_keys.push_back(key_in[i]);
    }
  if (!to_order.empty()) {
    auto ord = row_orders_keyed(to_order, to_order_keys, stream);
    size_t q = 0;
    for (size_t i = 0; i < n; ++i)
      if (wants_order(i)) {
        out[i].push_back(ord[q++]);
        out[i].push_back(ord[q++]);
      }
  }
  return out;
}

std::vector<std::vector<Tensor>> geometry_walk(const Tensor& indices, int64_t batch, const std::vector<int64_t>& kind,
                                               const std::vector<int64_t>& a_in, const std::vector<int64_t>& a_out,
                                               const std::vector<int64_t>& a_k, const std::vector<int64_t>& a_s,
                                               const std::vector<int64_t>& a_p, const std::vector<int64_t>& a_d,
                                               const std::vector<int64_t>& mode, const std::vector<int64_t>& K,
                                               const std::vector<int64_t>& ws_bytes, const std::vector<int64_t>& ref) {
  return geometry_walk_finish(geometry_walk_start(indices, batch, kind, a_in, a_out, a_k, a_s, a_p, a_d, mode, K, ws_bytes, ref, 0), {});
}

// a SparseSequential of conv -> BatchNorm -> ReLU layers whose rulebooks all exist already (the occupancy branch after
// BtcHotPath.prepare): the same autograd nodes as one conv_bn_relu call per layer, entered from Python ONCE -- the per-layer
// Python (module call, SparseConvolution.forward, argument marshalling: ~60 us a layer) is what bounds the forward pass
Tensor conv_bn_relu_chain(const Tensor& features, const std::vector<Tensor>& weights, const std::vector<OptTensor>& biases,
                          const std::vector<Tensor>& map_fwd, const std::vector<Tensor>& map_bwd, const std::vector<OptTensor>& order_fwd,
                          const std::vector<OptTensor>& order_bwd, const std::vector<OptTensor>& gammas,
                          const std::vector<OptTensor>& betas, const std::vector<OptTensor>& rms, const std::vector<OptTensor>& rvs,
                          const std::vector<OptTensor>& nbts, const std::vector<bool>& use_batch, const std::vector<double>& momenta,
                          const std::vector<double>& epss, const std::vector<bool>& relus, const Tensor& ws, const std::vector<int64_t>& ws_bytes,
                          const std::vector<bool>& overlaps, const std::vector<bool>& allow_defers) {
  const size_t L = weights.size();
  need(L >= 1 && biases.size() == L && map_fwd.size() == L && map_bwd.size() == L && order_fwd.size() == L && order_bwd.size() == L && gammas.size() == L && betas.size() == L && rms.size() == L &&
           rvs.size() == L && nbts.size() == L && use_batch.size() == L && momenta.size() == L && epss.size() == L && relus.size() == L &&
           ws_bytes.size() == L && overlaps.size() == L && allow_defers.size() == L,
       "conv_bn_relu_chain: per-layer argument lists differ in length");
  Tensor x = features;
  for (size_t i = 0; i < L; ++i)
    x = ConvBNReLUNode::apply(x, weights[i], biases[i], map_fwd[i], map_bwd[i], order_fwd[i], order_bwd[i], gammas[i], betas[i], rms[i], rvs[i], nbts[i], use_batch[i],
                              momenta[i], epss[i], relus[i], ws, ws_bytes[i], overlaps[i], allow_defers[i]);
  return x;
}

// one parameter group's optimizer step (csrc/optim.hip): norm clip + decoupled decay + Adam over the group's flat buffers, the
// gradients read where autograd left them.  The parameters are written through raw pointers, so their version counters are
// bumped here (the bf16 weight copies above are keyed on them).
void adam_group_step(const std::vector<Tensor>& params, const std::vector<Tensor>& grads, const Tensor& chunk_seg, const Tensor& chunk_off,
                     const Tensor& chunk_len, const Tensor& chunk_flat, const std::vector<int64_t>& seg_chunk0, const Tensor& flat_p,
                     const Tensor& flat_m, const Tensor& flat_v, int64_t step, double lr, double beta1, double beta2, double eps, double weight_decay,
                     double clip, const Tensor& ws, int64_t stream) {
  const size_t n = params.size();
  need(grads.size() == n && seg_chunk0.size() == n + 1, "adam_group_step: list lengths differ");
  std::vector<const float*> ptrs(n);
  std::vector<int32_t> sc(n + 1);
  for (size_t i = 0; i < n; ++i) {
    const Tensor& g = grads[i];
    need(g.defined() && g.is_contiguous() && g.scalar_type() == at::kFloat && g.numel() == params[i].numel() && g.get_device() == flat_p.get_device(),
         "adam_group_step: every gradient must be a contiguous fp32 tensor of its parameter's size on the parameters' device");
    ptrs[i] = (const float*)g.data_ptr();
    sc[i] = (int32_t)seg_chunk0[i];
  }
  sc[n] = (int32_t)seg_chunk0[n];
  chk(btc_adam_group_step(ptrs.data(), (int)n, (const int32_t*)chunk_seg.data_ptr(), (const int32_t*)chunk_off.data_ptr(),
                          (const int32_t*)chunk_len.data_ptr(), (const int64_t*)chunk_flat.data_ptr(), sc.data(), (int)chunk_seg.numel(),
                          (float*)flat_p.data_ptr(), (float*)flat_m.data_ptr(), (float*)flat_v.data_ptr(), (long long)step, (float)lr, (float)beta1,
                          (float)beta2, (float)eps, (float)weight_decay, (float)clip, ws.data_ptr(), (size_t)ws.numel(), st(stream)),
      "btc_adam_group_step");
  for (size_t i = 0; i < n; ++i) params[i].unsafeGetTensorImpl()->bump_version();
}

// A parameter group as an object: the list of its tensors crosses the Python boundary ONCE (converting a 300-tensor list costs ~30 us per
// call, reading 300 `.grad` attributes in Python another ~60 us -- per group and step, on a path that is bound by its host threads).
struct ParamList {
  std::vector<Tensor> p;
};

std::shared_ptr<ParamList> make_param_list(const std::vector<Tensor>& params) {
  auto pl = std::make_shared<ParamList>();
  pl->p = params;
  return pl;
}

// adam_group_step with the gradients read from the parameters' .grad here; -> false (nothing launched) when a gradient is missing or is
// not a contiguous fp32 tensor of its parameter's size on the parameters' device: the caller takes its general path
bool adam_group_step_pl(const std::shared_ptr<ParamList>& pl, const Tensor& chunk_seg, const Tensor& chunk_off, const Tensor& chunk_len,
                        const Tensor& chunk_flat, const std::vector<int64_t>& seg_chunk0, const Tensor& flat_p, const Tensor& flat_m,
                        const Tensor& flat_v, int64_t step, double lr, double beta1, double beta2, double eps, double weight_decay, double clip,
                        const Tensor& ws, int64_t stream) {
  const size_t n = pl->p.size();
  std::vector<Tensor> grads(n);
  for (size_t i = 0; i < n; ++i) {
    const Tensor& g = pl->p[i].grad();
    if (!(g.defined() && g.is_contiguous() && g.scalar_type() == at::kFloat && g.numel() == pl->p[i].numel() && g.get_device() == flat_p.get_device()))
      return false;
    grads[i] = g;
  }
  adam_group_step(pl->p, grads, chunk_seg, chunk_off, chunk_len, chunk_flat, seg_chunk0, flat_p, flat_m, flat_v, step, lr, beta1, beta2, eps,
                  weight_decay, clip, ws, stream);
  return true;
}

// p.grad = None for every parameter of the list (optimizer.zero_grad(set_to_none=True))
void clear_grads_pl(const std::shared_ptr<ParamList>& pl) {
  for (Tensor& t : pl->p) t.mutable_grad().reset();
}

// pack half of a gradient bucket: flat <- the gradients, one launch (csrc/optim.hip grads_pack); chunk tables as adam_group_step
void pack_grads(const std::vector<Tensor>& grads, const Tensor& chunk_seg, const Tensor& chunk_off, const Tensor& chunk_len, const Tensor& chunk_flat,
                const std::vector<int64_t>& seg_chunk0, const std::vector<int64_t>& sizes, const Tensor& flat, int64_t stream) {
  const size_t n = grads.size();
  need(seg_chunk0.size() == n + 1 && sizes.size() == n, "pack_grads: list lengths differ");
  std::vector<const float*> ptrs(n);
  std::vector<int32_t> sc(n + 1);
  for (size_t i = 0; i < n; ++i) {
    const Tensor& g = grads[i];
    need(g.defined() && g.is_contiguous() && g.scalar_type() == at::kFloat && g.numel() == sizes[i] && g.get_device() == flat.get_device(),
         "pack_grads: every gradient must be a contiguous fp32 tensor of its parameter's size on the bucket's device");
    ptrs[i] = (const float*)g.data_ptr();
    sc[i] = (int32_t)seg_chunk0[i];
  }
  sc[n] = (int32_t)seg_chunk0[n];
  chk(btc_grads_pack(ptrs.data(), (int)n, (const int32_t*)chunk_seg.data_ptr(), (const int32_t*)chunk_off.data_ptr(), (const int32_t*)chunk_len.data_ptr(),
                     (const int64_t*)chunk_flat.data_ptr(), sc.data(), (float*)flat.data_ptr(), st(stream)), "btc_grads_pack");
}

}  // namespace

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {
  m.doc() = "compiled PyTorch binding of libbtcdet_hip.so's hot entry points";
  m.def("split_weights", &split_weights, py::call_guard<py::gil_scoped_release>());
  m.def("bf16_weights", &bf16_weights, py::call_guard<py::gil_scoped_release>());
  m.def("conv_fwd", &conv_fwd, py::call_guard<py::gil_scoped_release>());
  m.def("bn_fwd", &bn_fwd, py::call_guard<py::gil_scoped_release>());
  m.def("conv_bn_fwd", &conv_bn_fwd, py::call_guard<py::gil_scoped_release>());
  m.def("bn_bwd", &bn_bwd, py::call_guard<py::gil_scoped_release>());
  m.def("conv_bwd", &conv_bwd, py::call_guard<py::gil_scoped_release>());
  m.def("conv_bn_relu", &conv_bn_relu, py::call_guard<py::gil_scoped_release>());
  m.def("row_orders", &row_orders, py::call_guard<py::gil_scoped_release>());
  m.def("geometry_walk", &geometry_walk, py::call_guard<py::gil_scoped_release>());
  py::class_<PendingWalk, std::shared_ptr<PendingWalk>>(m, "PendingWalk");
  m.def("geometry_walk_start", &geometry_walk_start, py::call_guard<py::gil_scoped_release>());
  m.def("geometry_walk_finish", &geometry_walk_finish, py::call_guard<py::gil_scoped_release>());
  m.def("conv_bn_relu_chain", &conv_bn_relu_chain, py::call_guard<py::gil_scoped_release>());
  m.def("pack_grads", &pack_grads, py::call_guard<py::gil_scoped_release>());
  m.def("adam_group_step", &adam_group_step, py::call_guard<py::gil_scoped_release>());
  py::class_<ParamList, std::shared_ptr<ParamList>>(m, "ParamList");
  m.def("make_param_list", &make_param_list);
  m.def("adam_group_step_pl", &adam_group_step_pl, py::call_guard<py::gil_scoped_release>());
  m.def("clear_grads_pl", &clear_grads_pl);
  m.def("join_wgrad", &join_wgrad, py::call_guard<py::gil_scoped_release>());
  m.def("set_defer_wgrad_join", &set_defer_wgrad_join);
  m.def("set_side_stream", &set_side_stream);
  m.def("rulebook_subm", &rulebook_subm, py::call_guard<py::gil_scoped_release>());
  m.def("rulebook_conv", &rulebook_conv, py::call_guard<py::gil_scoped_release>());
  py::class_<PendingRb, std::shared_ptr<PendingRb>>(m, "PendingRb");
  m.def("rulebook_conv_start", &rulebook_conv_start, py::call_guard<py::gil_scoped_release>());
  m.def("rulebook_conv_finish", &rulebook_conv_finish, py::call_guard<py::gil_scoped_release>());
  m.def("abi_version", []() { return btc_version(); });
}
