// torch.ops.adalog.* -- PyTorch custom-op registration (TORCH_LIBRARY, HIP dispatch key only) over the C ABI of
// libadalog_hip.so (include/adalog_hip.h).  north_star / SURVEY 8(b): "the build exposes torch.ops.adalog.* custom ops and
// the Python classes call them".  Thin shims: argument checks, output allocation and the current HIP stream come from
// torch; every byte of arithmetic happens in the hand-written kernels behind the C ABI.  There is NO CPU kernel: calling
// an op with CPU tensors raises torch's "no kernel for backend CPU" error (the product has no CPU fallback).
//
// Built by adalog_amd/csrc/build_torch_ops.py into adalog_amd/csrc/libadalog_torch.so (in-tree), loaded by
// adalog_amd/_torch_ops.py with torch.ops.load_library; adalog_amd/ops.py routes through torch.ops.adalog when it is there
// and through ctypes (INTEGRATION.md level 2) otherwise.
#include <ATen/ATen.h>
#include <c10/hip/HIPStream.h>
#include <torch/library.h>

#include <string>

extern "C" {
const char* adalog_last_error(void);
int adalog_uniform_fake_quant_f32(const float* x, float* y, uint8_t* bins, int64_t n, const float* scale, const float* zp,
                                  int64_t n_ch, int64_t inner, int n_bits, int sym, void* stream);
int adalog_log_fake_quant_f32(const float* x, float* y, uint8_t* bins, int64_t n, const float* scale, const int64_t* q,
                              const float* table1, const float* table2, int n_bits, const float* shift, int sub_shift,
                              int train_form, void* stream);
int adalog_log2_shift(const float* x, float* out, int64_t n, float shift, void* stream);
int adalog_score_act_fused_ok(int M, int64_t T, int K, int64_t Kp, int P, int n_bits);
int64_t adalog_score_act_fused_workspace_bytes(int64_t T, int64_t Kp);
int adalog_score_act_fused(const void* Wp, int M, int64_t Kp, const float* x, const float* Lx, int64_t T, int K,
                           const float* ref, const float* row_scale, const float* row_bias, const float* scale,
                           const float* qv, int P, int n_bits, const float* mant37, float shift, int clamp_u, float sa_mul,
                           double norm, void* workspace, int64_t workspace_bytes, float* scores, void* stream);
int adalog_topk(const float* scores, int P, int cols, int k, int32_t* idx, void* stream);
}

namespace {

void* cur_stream() { return (void*)c10::hip::getCurrentHIPStream().stream(); }

void check(int rc, const char* what) {
    TORCH_CHECK(rc == 0, what, " failed (rc=", rc, "): ", adalog_last_error());
}

const float* fptr(const at::Tensor& t, const char* name) {
    TORCH_CHECK(t.is_cuda() && t.scalar_type() == at::kFloat && t.is_contiguous(), name, ": expected a contiguous float32 HIP tensor");
    return t.data_ptr<float>();
}

// uniform.py:25-36 (eval form): y = (clamp(round(x/s) + round(zp), 0, 2L-1) - round(zp)) * s ; channel(i) = (i / inner) % n_ch
at::Tensor uniform_fake_quant(const at::Tensor& x, const at::Tensor& scale, const c10::optional<at::Tensor>& zero_point,
                              int64_t n_ch, int64_t inner, int64_t n_bits, bool sym) {
    at::Tensor y = at::empty_like(x);
    check(adalog_uniform_fake_quant_f32(fptr(x, "x"), y.data_ptr<float>(), nullptr, x.numel(), fptr(scale, "scale"),
                                        zero_point.has_value() ? fptr(*zero_point, "zero_point") : nullptr, n_ch, inner,
                                        (int)n_bits, sym ? 1 : 0, cur_stream()),
          "adalog::uniform_fake_quant");
    return y;
}

// logarithm.py:83-99 / 127-135 (eval form)
at::Tensor log_fake_quant(const at::Tensor& x, const at::Tensor& scale, const at::Tensor& q, const at::Tensor& table1,
                          const at::Tensor& table2, int64_t n_bits, const c10::optional<at::Tensor>& shift, bool sub_shift) {
    TORCH_CHECK(q.is_cuda() && q.scalar_type() == at::kLong, "q: expected an int64 HIP tensor");
    at::Tensor y = at::empty_like(x);
    check(adalog_log_fake_quant_f32(fptr(x, "x"), y.data_ptr<float>(), nullptr, x.numel(), fptr(scale, "scale"),
                                    q.data_ptr<int64_t>(), fptr(table1, "table1"), fptr(table2, "table2"), (int)n_bits,
                                    shift.has_value() ? fptr(*shift, "shift") : nullptr, sub_shift ? 1 : 0, 0, cur_stream()),
          "adalog::log_fake_quant");
    return y;
}

at::Tensor log2_shift(const at::Tensor& x, double shift) {
    at::Tensor out = at::empty_like(x);
    check(adalog_log2_shift(fptr(x, "x"), out.data_ptr<float>(), x.numel(), (float)shift, cur_stream()), "adalog::log2_shift");
    return out;
}

// linear.py:816-931: one post-GELU activation-candidate scoring call, quantisation fused into the GEMM's loader
at::Tensor score_act_fused(const at::Tensor& wp, const at::Tensor& x2, const at::Tensor& lx2, const at::Tensor& ref2,
                           const at::Tensor& row_scale, const c10::optional<at::Tensor>& row_bias, const at::Tensor& scale,
                           const at::Tensor& qv, int64_t n_bits, const at::Tensor& mant37, double shift, bool clamp_u,
                           double sa_mul, double norm) {
    TORCH_CHECK(wp.is_cuda() && wp.scalar_type() == at::kBFloat16 && wp.is_contiguous(), "wp: expected a contiguous bf16 HIP tensor");
    const int64_t M = wp.size(-2), Kp = wp.size(-1), T = x2.size(0), K = x2.size(1), P = scale.numel();
    TORCH_CHECK(adalog_score_act_fused_ok((int)M, T, (int)K, Kp, (int)P, (int)n_bits), "adalog::score_act_fused: shape not supported");
    const int64_t wsb = adalog_score_act_fused_workspace_bytes(T, Kp);
    at::Tensor ws = at::empty({(wsb + 7) / 8}, x2.options().dtype(at::kDouble));
    at::Tensor scores = at::empty({P, 1}, x2.options());
    check(adalog_score_act_fused(wp.data_ptr(), (int)M, Kp, fptr(x2, "x"), fptr(lx2, "log2 x"), T, (int)K, fptr(ref2, "ref"),
                                 fptr(row_scale, "row_scale"), row_bias.has_value() ? fptr(*row_bias, "row_bias") : nullptr,
                                 fptr(scale, "scale"), fptr(qv, "qv"), (int)P, (int)n_bits, fptr(mant37, "mant37"), (float)shift,
                                 clamp_u ? 1 : 0, (float)sa_mul, norm, ws.data_ptr(), wsb, scores.data_ptr<float>(), cur_stream()),
          "adalog::score_act_fused");
    return scores;
}

// linear.py:483-523 (torch.topk with ties made deterministic): idx [k, cols] int32
at::Tensor topk(const at::Tensor& scores, int64_t k) {
    TORCH_CHECK(scores.dim() == 2, "scores: expected [P, cols]");
    at::Tensor idx = at::empty({k, scores.size(1)}, scores.options().dtype(at::kInt));
    check(adalog_topk(fptr(scores, "scores"), (int)scores.size(0), (int)scores.size(1), (int)k, idx.data_ptr<int32_t>(), cur_stream()),
          "adalog::topk");
    return idx;
}

}  // namespace

TORCH_LIBRARY(adalog, m) {
    m.def("uniform_fake_quant(Tensor x, Tensor scale, Tensor? zero_point, int n_ch, int inner, int n_bits, bool sym) -> Tensor");
    m.def("log_fake_quant(Tensor x, Tensor scale, Tensor q, Tensor table1, Tensor table2, int n_bits, Tensor? shift, bool sub_shift) -> Tensor");
    m.def("log2_shift(Tensor x, float shift) -> Tensor");
    m.def("score_act_fused(Tensor wp, Tensor x2, Tensor lx2, Tensor ref2, Tensor row_scale, Tensor? row_bias, Tensor scale, Tensor qv, "
          "int n_bits, Tensor mant37, float shift, bool clamp_u, float sa_mul, float norm) -> Tensor");
    m.def("topk(Tensor scores, int k) -> Tensor");
}

// HIP dispatch key only ("CUDA" is the HIP key on ROCm builds of PyTorch): there is deliberately no CPU implementation
TORCH_LIBRARY_IMPL(adalog, CUDA, m) {
    m.impl("uniform_fake_quant", &uniform_fake_quant);
    m.impl("log_fake_quant", &log_fake_quant);
    m.impl("log2_shift", &log2_shift);
    m.impl("score_act_fused", &score_act_fused);
    m.impl("topk", &topk);
}
