// Compiled binding of the hot entry points of libbtcdet_hip.so for PyTorch (CPython extension `btcdet_amd._btcfast`).
//
// spconv reaches its native code through a pybind11 layer (spconv/ops.py -> torch.ops.spconv.*); this is the same layer for
// this library.  It exists because the forward pass is bound by the host's launch rate (tools/host_phases.py,
// tools/layer_host_split.py): per sparse layer the ctypes route spends ~30 us of Python on output allocation, argument
// marshalling and attribute lookups around ~12 us of launches.  Here one call allocates the outputs with at::empty and
// calls the C ABI (include/btcdet_hip.h) directly.  No HIP headers are needed: the stream comes in as an integer handle
// (torch._C._cuda_getCurrentRawStream), memory comes from torch's caching allocator.  The ctypes route
// (btcdet_amd/_lib.py) stays the reference binding and computes the same thing; tests run both.
#include <torch/extension.h>

#include <stdexcept>
#include <string>
#include <tuple>

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

// out = conv(features) ; features (n_src, Cin) fp32 | bf16 contiguous, w [K.., Cin, Cout] fp32, map_fwd (n_res, K) int32
Tensor conv_fwd(const Tensor& features, const Tensor& w, const OptTensor& bias, const Tensor& map_fwd, int64_t stream) {
  const int64_t cin = w.size(-2), cout = w.size(-1), K = map_fwd.size(1), n_res = map_fwd.size(0);
  need(features.is_contiguous() && w.is_contiguous() && map_fwd.is_contiguous(), "conv_fwd: contiguous tensors expected");
  need(w.numel() == K * cin * cout && features.size(1) == cin, "conv_fwd: weight does not match the rulebook / features");
  Tensor out = at::empty({n_res, cout}, features.options());
  if (features.scalar_type() == at::kBFloat16)
    chk(btc_conv_fwd_bf16(features.data_ptr(), (const float*)w.data_ptr(), fptr(bias), (const int32_t*)map_fwd.data_ptr(), (int)n_res, (int)K,
                          (int)cin, (int)cout, out.data_ptr(), st(stream)), "btc_conv_fwd_bf16");
  else
    chk(btc_conv_fwd((const float*)features.data_ptr(), (const float*)w.data_ptr(), fptr(bias), (const int32_t*)map_fwd.data_ptr(), (int)n_res,
                     (int)K, (int)cin, (int)cout, (float*)out.data_ptr(), st(stream)), "btc_conv_fwd");
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

std::tuple<Tensor, Tensor, Tensor> conv_bn_fwd(const Tensor& features, const Tensor& w, const OptTensor& bias, const Tensor& map_fwd,
                                               const OptTensor& gamma, const OptTensor& beta, const OptTensor& rm, const OptTensor& rv,
                                               const OptTensor& nbt, bool use_batch, double momentum, double eps, bool relu, const Tensor& ws,
                                               int64_t ws_bytes, int64_t stream) {
  Tensor x = conv_fwd(features, w, bias, map_fwd, stream);
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

// din (n_src, Cin), dw (shape of w); either may come back undefined (None) when not needed.  Same stream for both.
std::tuple<OptTensor, OptTensor> conv_bwd(const Tensor& features, const Tensor& w, const Tensor& map_fwd, const Tensor& map_bwd,
                                          const Tensor& grad_out, bool need_din, bool need_dw, int64_t stream) {
  const int64_t cin = w.size(-2), cout = w.size(-1), K = map_fwd.size(1), n_res = map_fwd.size(0), n_src = map_bwd.size(0);
  need(grad_out.is_contiguous() && grad_out.scalar_type() == features.scalar_type(), "conv_bwd: grad must be contiguous and of the activation type");
  const bool bf = features.scalar_type() == at::kBFloat16;
  OptTensor din, dw;
  if (need_dw) {
    Tensor g = at::empty(w.sizes(), w.options());
    const size_t ws_bytes = btc_conv_wgrad_ws_bytes((int)n_res, (int)K, (int)cin, (int)cout, (int)n_src);
    Tensor ws = at::empty({(int64_t)(ws_bytes > 256 ? ws_bytes : 256)}, features.options().dtype(at::kByte));
    if (bf)
      chk(btc_conv_wgrad_bf16(features.data_ptr(), grad_out.data_ptr(), (const int32_t*)map_fwd.data_ptr(), (int)n_res,
                              (const int32_t*)map_bwd.data_ptr(), (int)n_src, (int)K, (int)cin, (int)cout, (float*)g.data_ptr(), ws.data_ptr(),
                              ws_bytes, st(stream)), "btc_conv_wgrad_bf16");
    else
      chk(btc_conv_wgrad((const float*)features.data_ptr(), (const float*)grad_out.data_ptr(), (const int32_t*)map_fwd.data_ptr(), (int)n_res,
                         (const int32_t*)map_bwd.data_ptr(), (int)n_src, (int)K, (int)cin, (int)cout, (float*)g.data_ptr(), ws.data_ptr(),
                         ws_bytes, st(stream)), "btc_conv_wgrad");
    dw = g;
  }
  if (need_din) {
    Tensor d = at::empty({n_src, cin}, features.options());
    if (bf)
      chk(btc_conv_dgrad_bf16(grad_out.data_ptr(), (const float*)w.data_ptr(), (const int32_t*)map_bwd.data_ptr(), (int)n_src, (int)K, (int)cin,
                              (int)cout, d.data_ptr(), st(stream)), "btc_conv_dgrad_bf16");
    else
      chk(btc_conv_dgrad((const float*)grad_out.data_ptr(), (const float*)w.data_ptr(), (const int32_t*)map_bwd.data_ptr(), (int)n_src, (int)K,
                         (int)cin, (int)cout, (float*)d.data_ptr(), st(stream)), "btc_conv_dgrad");
    din = d;
  }
  return std::make_tuple(din, dw);
}

// submanifold rulebook: returns nbr (2, n, K) = nbr_out | nbr_in.  p_* are the addresses of int32[3] host arrays.
Tensor rulebook_subm(const Tensor& indices, int64_t batch, int64_t p_in, int64_t p_k, int64_t p_d, int64_t K, int64_t stream) {
  const int64_t n = indices.size(0);
  Tensor nbr = at::empty({2, n, K}, indices.options());
  const size_t ws_bytes = btc_rulebook_subm_ws_bytes((int)n);
  Tensor ws = at::empty({(int64_t)(ws_bytes > 256 ? ws_bytes : 256)}, indices.options().dtype(at::kByte));
  int32_t* out = (int32_t*)nbr.data_ptr();
  chk(btc_rulebook_subm((const int32_t*)indices.data_ptr(), (int)n, (int)batch, ip(p_in), ip(p_k), ip(p_d), out, out + n * K, ws.data_ptr(),
                        ws_bytes, st(stream)), "btc_rulebook_subm");
  return nbr;
}

// strided / transposed rulebook, synchronous: count, one blocking 4-byte read-back, fill
std::tuple<Tensor, Tensor, Tensor> rulebook_conv(const Tensor& indices, int64_t batch, int64_t p_in, int64_t p_out, int64_t p_k, int64_t p_s,
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
  return std::make_tuple(out_indices, nbr_out, nbr_in);
}

}  // namespace

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {
  m.doc() = "compiled PyTorch binding of libbtcdet_hip.so's hot entry points";
  m.def("conv_fwd", &conv_fwd);
  m.def("bn_fwd", &bn_fwd);
  m.def("conv_bn_fwd", &conv_bn_fwd);
  m.def("bn_bwd", &bn_bwd);
  m.def("conv_bwd", &conv_bwd);
  m.def("rulebook_subm", &rulebook_subm);
  m.def("rulebook_conv", &rulebook_conv);
  m.def("abi_version", []() { return btc_version(); });
}
