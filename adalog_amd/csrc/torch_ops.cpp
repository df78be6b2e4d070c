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
#include <tuple>

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
int adalog_pack_uniform(const float* x, int64_t G, int64_t R, int64_t K, int64_t sxg, int64_t sxr, int64_t sxk,
                        const float* scale, const float* zero_point, int64_t C, int64_t pc, int64_t gmod, int64_t pg,
                        int64_t pr, int n_bits, int out_dtype, void* out, int64_t Kp, int32_t* rowsum, int c_inner,
                        void* stream);
int adalog_pack_adalog_bf16(const float* x, int64_t G, int64_t R, int64_t K, int64_t sxg, int64_t sxr, int64_t sxk,
                            const float* scale, const float* qv, int64_t C, int64_t pc, int64_t gmod, int64_t pg,
                            int n_bits, const float* mant37, const float* shift, int clamp_u, void* out, int64_t Kp,
                            int c_inner, void* stream);
int64_t adalog_gemm_score_layout(int M, int N, int C, int G, int gmod, int ref_div, int reduce_cols, int dtype, int64_t Kp,
                                 int64_t k_valid, int ref_transposed, int* MT, int* Npad, int* mode);
int adalog_gemm_score(int dtype, const void* A, const void* B, int64_t sAc, int64_t sAg, int64_t sBc, int64_t sBg, int M, int N,
                      int64_t Kp, int64_t k_valid, int C, int G, int gmod, const float* ref, int64_t ldr, int64_t sRg,
                      int64_t ref_cs, int ref_div, const float* sa, int64_t sa_c, int64_t sa_g, float sa_mul, const float* sb,
                      int64_t sb_c, int64_t sb_g, int64_t sb_n, const float* bias, int64_t bi_c, int64_t bi_g, int64_t bi_n,
                      const float* row_scale, const float* row_bias, float* partial, int64_t partial_elems, float* out,
                      int64_t ldo, int64_t sOc, int64_t sOg, int order, int reduce_cols, void* stream);
int adalog_finish_scores(const float* partial, float* scores, int MT, int N, int Npad, int C, int G, int gmod, int keep_h,
                         int keep_n, int cand_inner, double norm, void* workspace, int64_t workspace_bytes, void* stream);
int64_t adalog_finish_workspace_bytes(int MT, int N, int C, int G, int keep_n, int cand_inner);
int adalog_finish_topk_next(const float* partial, float* scores, int MT, int N, int Npad, int C, int G, int gmod, int keep_h,
                            int keep_n, int cand_inner, double norm, void* workspace, int64_t workspace_bytes, int k,
                            const float* scale, const float* zp, const float* third, int new_cnt, const float* lin, float* delta,
                            int has_clamp, float clamp_min, float* out_scale, float* out_zp, float* out_third, void* stream);
int adalog_topk_next(const float* scores, int P, int cols, int k, const float* scale, const float* zp, const float* third,
                     int new_cnt, const float* lin, float* delta, int has_clamp, float clamp_min, float* out_scale,
                     float* out_zp, float* out_third, int* idx_out, void* stream);
int adalog_score_w_self(const float* w, int rows, int I, const float* scale, const float* zp, int P, int n_bits,
                        float* scores, void* stream);
int adalog_score_a_self(const float* x, int64_t rows, int I, const float* scale, const float* zp, int P, int channel_wise,
                        int n_bits, double norm, float* partial, int64_t partial_elems, float* scores, void* stream);
int64_t adalog_score_a_self_partial_elems(int64_t rows, int I, int P);
int adalog_score_act_gen_ok(int dtype, int M, int64_t T, int K, int64_t Kp, int P);
int64_t adalog_score_act_gen_workspace_bytes(int dtype, int M, int64_t T, int K, int64_t Kp, int P);
int adalog_score_act_gen(int dtype, const void* Wp, int M, int64_t Kp, const float* x, int64_t T, int K, int64_t ldx,
                         const float* scale, const float* zp, int P, int n_bits, const float* ref, const float* row_scale,
                         const float* row_bias, double norm, void* workspace, int64_t workspace_bytes, float* scores,
                         void* stream);
int64_t adalog_sorted_prefix_workspace_bytes(int64_t S, int64_t n);
int adalog_sorted_prefix_build(const float* x, int64_t S, int64_t n, float* sorted, double* prefix, void* workspace,
                               int64_t workspace_bytes, void* stream);
int adalog_score_self_sorted(const float* sorted, const double* prefix, int64_t S, int64_t n, const float* scale,
                             const float* zp, int P, int n_bits, double norm, float* scores, void* stream);
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

at::ScalarType pack_dtype(int64_t dtype) {
    TORCH_CHECK(dtype >= 0 && dtype <= 3, "dtype: 0 (i8), 1 (bf16), 2 (f32) or 3 (fp8 e4m3)");
    return dtype == 0 ? at::kChar : dtype == 1 ? at::kBFloat16 : dtype == 2 ? at::kFloat : at::kFloat8_e4m3fn;
}
const float* optf(const c10::optional<at::Tensor>& t, const char* name) { return t.has_value() ? fptr(*t, name) : nullptr; }
void check_view3(const at::Tensor& x3) {
    TORCH_CHECK(x3.is_cuda() && x3.dim() == 3 && x3.scalar_type() == at::kFloat, "operand must be a 3-D float32 HIP view [G, R, K]");
}

// operand packing of a scoring call (reference linear.py:376,413 / matmul.py:170-207 fake-quant of the candidate operand):
// -> (packed [C, G, R, Kp] or [1, G, R*C, Kp], int32 rowsum [C, G, R] or an empty tensor)
std::tuple<at::Tensor, at::Tensor> pack_uniform(const at::Tensor& x3, const at::Tensor& scale, const at::Tensor& zero_point,
                                                int64_t C, int64_t pc, int64_t gmod, int64_t pg, int64_t pr, int64_t n_bits,
                                                int64_t dtype, int64_t Kp, bool want_rowsum, bool c_inner) {
    check_view3(x3);
    const int64_t G = x3.size(0), R = x3.size(1), K = x3.size(2);
    at::Tensor out = at::empty(c_inner ? at::IntArrayRef({1, G, R * C, Kp}) : at::IntArrayRef({C, G, R, Kp}),
                               x3.options().dtype(pack_dtype(dtype)));
    at::Tensor rowsum = want_rowsum ? at::empty({C, G, R}, x3.options().dtype(at::kInt)) : at::empty({0}, x3.options().dtype(at::kInt));
    check(adalog_pack_uniform(x3.data_ptr<float>(), G, R, K, x3.stride(0), x3.stride(1), x3.stride(2), fptr(scale, "scale"),
                              fptr(zero_point, "zero_point"), C, pc, gmod, pg, pr, (int)n_bits, (int)dtype, out.data_ptr(), Kp,
                              want_rowsum ? rowsum.data_ptr<int32_t>() : nullptr, c_inner ? 1 : 0, cur_stream()),
          "adalog::pack_uniform");
    return {out, rowsum};
}

at::Tensor pack_adalog(const at::Tensor& x3, const at::Tensor& scale, const at::Tensor& qv, int64_t C, int64_t pc, int64_t gmod,
                       int64_t pg, int64_t n_bits, const at::Tensor& mant37, const c10::optional<at::Tensor>& shift, bool clamp_u,
                       int64_t Kp, bool c_inner) {
    check_view3(x3);
    const int64_t G = x3.size(0), R = x3.size(1), K = x3.size(2);
    at::Tensor out = at::empty(c_inner ? at::IntArrayRef({1, G, R * C, Kp}) : at::IntArrayRef({C, G, R, Kp}),
                               x3.options().dtype(at::kBFloat16));
    check(adalog_pack_adalog_bf16(x3.data_ptr<float>(), G, R, K, x3.stride(0), x3.stride(1), x3.stride(2), fptr(scale, "scale"),
                                  fptr(qv, "qv"), C, pc, gmod, pg, (int)n_bits, fptr(mant37, "mant37"), optf(shift, "shift"),
                                  clamp_u ? 1 : 0, out.data_ptr(), Kp, c_inner ? 1 : 0, cur_stream()),
          "adalog::pack_adalog");
    return out;
}

struct GemmLaunch { at::Tensor partial; int MT, Npad, mode, n_last; };

GemmLaunch gemm_score_launch(int64_t dtype, const at::Tensor& A, const at::Tensor& B, int64_t M, int64_t N, int64_t C, int64_t G,
                             int64_t gmod, int64_t k_valid, const at::Tensor& ref, const at::Tensor& sa, int64_t sa_c, int64_t sa_g,
                             double sa_mul, const at::Tensor& sb, int64_t sb_c, int64_t sb_g, int64_t sb_n,
                             const c10::optional<at::Tensor>& bias, int64_t bi_c, int64_t bi_g, int64_t bi_n, bool keep_n,
                             int64_t ref_div, int64_t order, bool ref_transposed, const c10::optional<at::Tensor>& row_scale,
                             const c10::optional<at::Tensor>& row_bias) {
    TORCH_CHECK(A.is_cuda() && B.is_cuda() && A.is_contiguous() && B.is_contiguous() && A.dim() == 4 && B.dim() == 4, "A / B: contiguous 4-D HIP tensors");
    // dtype 4: bf16 rows x fp8 candidate columns (adalog_gemm_mixed_ok)
    TORCH_CHECK(A.scalar_type() == pack_dtype(dtype == 4 ? 1 : dtype) && B.scalar_type() == pack_dtype(dtype == 4 ? 3 : dtype),
                "A / B: dtype does not match");
    const int64_t Kp = A.size(3), n_cols = N * ref_div, c_grid = ref_div > 1 ? 1 : C;
    TORCH_CHECK(B.size(3) == Kp && A.size(2) == M && B.size(2) == n_cols, "A / B: shapes do not match M, N, Kp");
    const int64_t sAc = A.size(0) == 1 ? 0 : A.stride(0), sAg = (A.size(1) == 1 && G > 1) ? 0 : A.stride(1);
    const int64_t sBc = B.size(0) == 1 ? 0 : B.stride(0), sBg = (B.size(1) == 1 && G > 1) ? 0 : B.stride(1);
    fptr(ref, "ref");
    int64_t ldr, ref_cs;
    if (ref_transposed) { TORCH_CHECK(ref.size(-1) == M && ref.size(-2) == N, "ref: expected [G, N, M]"); ldr = 1; ref_cs = M; }
    else { ldr = ref.size(-1); ref_cs = 1; }
    const int64_t sRg = G == 1 ? 0 : ref.size(-1) * ref.size(-2);
    const int reduce_cols = keep_n ? 0 : 1;
    GemmLaunch L;
    const int64_t n_part = adalog_gemm_score_layout((int)M, (int)n_cols, (int)c_grid, (int)G, (int)gmod, (int)ref_div, reduce_cols,
                                                    (int)dtype, Kp, k_valid, ref_transposed ? 1 : 0, &L.MT, &L.Npad, &L.mode);
    L.partial = at::empty({(n_part + 1) / 2}, A.options().dtype(at::kDouble));              // 8-byte aligned
    check(adalog_gemm_score((int)dtype, A.data_ptr(), B.data_ptr(), sAc, sAg, sBc, sBg, (int)M, (int)n_cols, Kp, k_valid, (int)c_grid,
                            (int)G, (int)gmod, ref.data_ptr<float>(), ldr, sRg, ref_cs, (int)ref_div, fptr(sa, "sa"), sa_c, sa_g,
                            (float)sa_mul, fptr(sb, "sb"), sb_c, sb_g, sb_n, optf(bias, "bias"), bi_c, bi_g, bi_n,
                            optf(row_scale, "row_scale"), optf(row_bias, "row_bias"), (float*)L.partial.data_ptr(), n_part, nullptr, 0,
                            0, 0, (int)order, reduce_cols, cur_stream()),
          "adalog::gemm_score");
    L.n_last = (reduce_cols && L.mode != 1) ? L.Npad : (int)N;
    return L;
}

// One scoring call: the MFMA GEMM with the squared-error epilogue and its fixed-order finish (reference linear.py:378-385,
// 415-424; matmul.py:154-164, 345-352).  A: [C|1, G|1, M, Kp], B: [C|1, G|1, N*ref_div, Kp], ref: [G, M, N] (or [G, N, M]).
at::Tensor gemm_score(int64_t dtype, const at::Tensor& A, const at::Tensor& B, int64_t M, int64_t N, int64_t C, int64_t G,
                      int64_t gmod, int64_t k_valid, const at::Tensor& ref, const at::Tensor& sa, int64_t sa_c, int64_t sa_g,
                      double sa_mul, const at::Tensor& sb, int64_t sb_c, int64_t sb_g, int64_t sb_n,
                      const c10::optional<at::Tensor>& bias, int64_t bi_c, int64_t bi_g, int64_t bi_n, bool keep_h, bool keep_n,
                      double norm, int64_t ref_div, int64_t order, bool ref_transposed, const c10::optional<at::Tensor>& row_scale,
                      const c10::optional<at::Tensor>& row_bias) {
    const GemmLaunch L = gemm_score_launch(dtype, A, B, M, N, C, G, gmod, k_valid, ref, sa, sa_c, sa_g, sa_mul, sb, sb_c, sb_g, sb_n,
                                           bias, bi_c, bi_g, bi_n, keep_n, ref_div, order, ref_transposed, row_scale, row_bias);
    const int64_t cols = (keep_h ? gmod : 1) * (keep_n ? N : 1);
    at::Tensor scores = at::empty({C, cols}, A.options().dtype(at::kFloat));
    const int64_t wsb = adalog_finish_workspace_bytes(L.MT, L.n_last, (int)C, (int)G, keep_n ? 1 : 0, L.mode);
    at::Tensor ws = at::empty({wsb / 8}, A.options().dtype(at::kDouble));
    check(adalog_finish_scores((const float*)L.partial.data_ptr(), scores.data_ptr<float>(), L.MT, L.n_last, L.Npad, (int)C, (int)G,
                               (int)gmod, keep_h ? 1 : 0, keep_n ? 1 : 0, L.mode, norm, wsb ? ws.data_ptr() : nullptr, wsb, cur_stream()),
          "adalog::gemm_score (finish)");
    return scores;
}

// the same launch WITHOUT the finish: returns the partial sums (layout: adalog_gemm_score_layout) for adalog::finish_topk_next
at::Tensor gemm_score_partial(int64_t dtype, const at::Tensor& A, const at::Tensor& B, int64_t M, int64_t N, int64_t C, int64_t G,
                              int64_t gmod, int64_t k_valid, const at::Tensor& ref, const at::Tensor& sa, int64_t sa_c, int64_t sa_g,
                              double sa_mul, const at::Tensor& sb, int64_t sb_c, int64_t sb_g, int64_t sb_n,
                              const c10::optional<at::Tensor>& bias, int64_t bi_c, int64_t bi_g, int64_t bi_n, bool keep_h, bool keep_n,
                              double norm, int64_t ref_div, int64_t order, bool ref_transposed,
                              const c10::optional<at::Tensor>& row_scale, const c10::optional<at::Tensor>& row_bias) {
    (void)keep_h; (void)norm;
    return gemm_score_launch(dtype, A, B, M, N, C, G, gmod, k_valid, ref, sa, sa_c, sa_g, sa_mul, sb, sb_c, sb_g, sb_n, bias, bi_c, bi_g,
                             bi_n, keep_n, ref_div, order, ref_transposed, row_scale, row_bias).partial;
}

// finish + top-k + next grid (linear.py:483-523 after a scoring call) in one launch where the layout allows it
std::tuple<at::Tensor, at::Tensor, at::Tensor, at::Tensor> finish_topk_next(
        const at::Tensor& partial, int64_t MT, int64_t n_last, int64_t Npad, int64_t C, int64_t G, int64_t gmod, bool keep_h, bool keep_n,
        int64_t mode, double norm, int64_t k, const at::Tensor& scale, const c10::optional<at::Tensor>& zp,
        const c10::optional<at::Tensor>& third, int64_t new_cnt, const c10::optional<at::Tensor>& lin,
        const c10::optional<at::Tensor>& delta, bool has_clamp, double clamp_min) {
    TORCH_CHECK(partial.is_cuda() && partial.is_contiguous(), "partial: contiguous HIP tensor");
    const int64_t cols = scale.size(1), rows = new_cnt > 0 ? k * new_cnt : 1;
    at::Tensor scores = at::empty({C, cols}, scale.options());
    at::Tensor o_s = at::empty({rows, cols}, scale.options());
    at::Tensor o_z = zp.has_value() ? at::empty({rows, cols}, scale.options()) : at::empty({0}, scale.options());
    at::Tensor o_t = third.has_value() ? at::empty({rows, cols}, scale.options()) : at::empty({0}, scale.options());
    const int64_t wsb = adalog_finish_workspace_bytes((int)MT, (int)n_last, (int)C, (int)G, keep_n ? 1 : 0, (int)mode);
    at::Tensor ws = at::empty({wsb / 8}, scale.options().dtype(at::kDouble));
    check(adalog_finish_topk_next((const float*)partial.data_ptr(), scores.data_ptr<float>(), (int)MT, (int)n_last, (int)Npad, (int)C,
                                  (int)G, (int)gmod, keep_h ? 1 : 0, keep_n ? 1 : 0, (int)mode, norm, wsb ? ws.data_ptr() : nullptr, wsb,
                                  (int)k, fptr(scale, "scale"), optf(zp, "zp"), optf(third, "third"), (int)new_cnt, optf(lin, "lin"),
                                  delta.has_value() ? const_cast<float*>(fptr(*delta, "delta")) : nullptr, has_clamp ? 1 : 0,
                                  (float)clamp_min, o_s.data_ptr<float>(), zp.has_value() ? o_z.data_ptr<float>() : nullptr,
                                  third.has_value() ? o_t.data_ptr<float>() : nullptr, cur_stream()),
          "adalog::finish_topk_next");
    return {scores, o_s, o_z, o_t};
}

// linear.py:483-523: top-k of the scores and the next candidate grid around the winners (or the committed winner)
std::tuple<at::Tensor, at::Tensor, at::Tensor> topk_next(const at::Tensor& scores, const at::Tensor& scale,
                                                         const c10::optional<at::Tensor>& zp, const c10::optional<at::Tensor>& third,
                                                         int64_t k, int64_t new_cnt, const c10::optional<at::Tensor>& lin,
                                                         const c10::optional<at::Tensor>& delta, bool has_clamp, double clamp_min) {
    const int64_t P = scores.size(0), cols = scores.size(1), rows = new_cnt > 0 ? k * new_cnt : 1;
    at::Tensor o_s = at::empty({rows, cols}, scale.options());
    at::Tensor o_z = zp.has_value() ? at::empty({rows, cols}, scale.options()) : at::empty({0}, scale.options());
    at::Tensor o_t = third.has_value() ? at::empty({rows, cols}, scale.options()) : at::empty({0}, scale.options());
    check(adalog_topk_next(fptr(scores, "scores"), (int)P, (int)cols, (int)k, fptr(scale, "scale"), optf(zp, "zp"), optf(third, "third"),
                           (int)new_cnt, optf(lin, "lin"), delta.has_value() ? const_cast<float*>(fptr(*delta, "delta")) : nullptr,
                           has_clamp ? 1 : 0, (float)clamp_min, o_s.data_ptr<float>(), zp.has_value() ? o_z.data_ptr<float>() : nullptr,
                           third.has_value() ? o_t.data_ptr<float>() : nullptr, nullptr, cur_stream()),
          "adalog::topk_next");
    return {o_s, o_z, o_t};
}

// linear.py:296-309
at::Tensor score_w_self(const at::Tensor& w2, const at::Tensor& scale, const at::Tensor& zp, int64_t n_bits) {
    const int64_t rows = w2.size(0), I = w2.size(1), P = scale.size(0);
    at::Tensor scores = at::empty({P, rows}, w2.options());
    check(adalog_score_w_self(fptr(w2, "weight"), (int)rows, (int)I, fptr(scale, "scale"), fptr(zp, "zp"), (int)P, (int)n_bits,
                              scores.data_ptr<float>(), cur_stream()),
          "adalog::score_w_self");
    return scores;
}

// linear.py:320-345
at::Tensor score_a_self(const at::Tensor& x2, const at::Tensor& scale, const at::Tensor& zp, bool channel_wise, int64_t n_bits,
                        double norm) {
    const int64_t rows = x2.size(0), I = x2.size(1), P = scale.size(0);
    const int64_t n_part = adalog_score_a_self_partial_elems(rows, (int)I, (int)P);
    at::Tensor partial = at::empty({n_part}, x2.options());
    at::Tensor scores = at::empty({P, channel_wise ? I : 1}, x2.options());
    check(adalog_score_a_self(fptr(x2, "x"), rows, (int)I, fptr(scale, "scale"), fptr(zp, "zp"), (int)P, channel_wise ? 1 : 0,
                              (int)n_bits, norm, partial.data_ptr<float>(), n_part, scores.data_ptr<float>(), cur_stream()),
          "adalog::score_a_self");
    return scores;
}

// linear.py:394-430: one activation-candidate scoring call, the candidate operand generated inside the slab kernel
at::Tensor score_act_gen(int64_t dtype, const at::Tensor& wp, const at::Tensor& x2, const at::Tensor& scale, const at::Tensor& zp,
                         int64_t n_bits, const at::Tensor& ref2, const at::Tensor& row_scale, const c10::optional<at::Tensor>& row_bias,
                         double norm) {
    TORCH_CHECK(wp.is_cuda() && wp.is_contiguous() && wp.scalar_type() == pack_dtype(dtype), "wp: contiguous packed weight image of the given dtype");
    const int64_t M = wp.size(-2), Kp = wp.size(-1), T = x2.size(0), K = x2.size(1), P = scale.numel();
    TORCH_CHECK(ref2.dim() == 2 && ref2.size(0) == T && ref2.size(1) == M, "ref: expected [T, M]");
    const int64_t wsb = adalog_score_act_gen_workspace_bytes((int)dtype, (int)M, T, (int)K, Kp, (int)P);
    TORCH_CHECK(wsb >= 0, "adalog::score_act_gen: shape not supported (adalog_score_act_gen_ok)");
    at::Tensor ws = at::empty({(wsb + 15) / 16 * 2}, x2.options().dtype(at::kDouble));
    at::Tensor scores = at::empty({P, 1}, x2.options());
    check(adalog_score_act_gen((int)dtype, wp.data_ptr(), (int)M, Kp, fptr(x2, "x"), T, (int)K, K, fptr(scale, "scale"), fptr(zp, "zp"),
                               (int)P, (int)n_bits, fptr(ref2, "ref"), fptr(row_scale, "row_scale"), optf(row_bias, "row_bias"), norm,
                               ws.data_ptr(), ws.numel() * 8, scores.data_ptr<float>(), cur_stream()),
          "adalog::score_act_gen");
    return scores;
}

// ... WITHOUT the finish: returns the workspace whose head holds the fp64 accumulators [wgs][1][256] (for adalog::finish_topk_next)
at::Tensor score_act_gen_partial(int64_t dtype, const at::Tensor& wp, const at::Tensor& x2, const at::Tensor& scale, const at::Tensor& zp,
                                 int64_t n_bits, const at::Tensor& ref2, const at::Tensor& row_scale,
                                 const c10::optional<at::Tensor>& row_bias) {
    TORCH_CHECK(wp.is_cuda() && wp.is_contiguous() && wp.scalar_type() == pack_dtype(dtype), "wp: contiguous packed weight image of the given dtype");
    const int64_t M = wp.size(-2), Kp = wp.size(-1), T = x2.size(0), K = x2.size(1), P = scale.numel();
    TORCH_CHECK(ref2.dim() == 2 && ref2.size(0) == T && ref2.size(1) == M, "ref: expected [T, M]");
    const int64_t wsb = adalog_score_act_gen_workspace_bytes((int)dtype, (int)M, T, (int)K, Kp, (int)P);
    TORCH_CHECK(wsb >= 0, "adalog::score_act_gen_partial: shape not supported (adalog_score_act_gen_ok)");
    at::Tensor ws = at::empty({(wsb + 15) / 16 * 2}, x2.options().dtype(at::kDouble));
    check(adalog_score_act_gen((int)dtype, wp.data_ptr(), (int)M, Kp, fptr(x2, "x"), T, (int)K, K, fptr(scale, "scale"), fptr(zp, "zp"),
                               (int)P, (int)n_bits, fptr(ref2, "ref"), fptr(row_scale, "row_scale"), optf(row_bias, "row_bias"), 1.0,
                               ws.data_ptr(), ws.numel() * 8, nullptr, cur_stream()),
          "adalog::score_act_gen_partial");
    return ws;
}

// linear.py:296-318 / 320-353 in sorted-prefix form (csrc/sorted_score.hip): x2 [S, n] -> (sorted [S, n], prefix [S, n + 1, 2] f64)
std::tuple<at::Tensor, at::Tensor> sorted_prefix(const at::Tensor& x2) {
    TORCH_CHECK(x2.dim() == 2, "x2: expected [S, n]");
    const int64_t S = x2.size(0), n = x2.size(1);
    const int64_t wsb = adalog_sorted_prefix_workspace_bytes(S, n);
    TORCH_CHECK(wsb >= 0, "adalog::sorted_prefix: unsupported size");
    at::Tensor ws = at::empty({(wsb + 15) / 16 * 2}, x2.options().dtype(at::kDouble));
    at::Tensor sorted = at::empty_like(x2);
    at::Tensor prefix = at::empty({S, n + 1, 2}, x2.options().dtype(at::kDouble));
    check(adalog_sorted_prefix_build(fptr(x2, "x"), S, n, sorted.data_ptr<float>(), prefix.data_ptr<double>(), ws.data_ptr(),
                                     ws.numel() * 8, cur_stream()),
          "adalog::sorted_prefix");
    return {sorted, prefix};
}

at::Tensor score_self_sorted(const at::Tensor& sorted, const at::Tensor& prefix, const at::Tensor& scale, const at::Tensor& zp,
                             int64_t n_bits, double norm) {
    const int64_t S = sorted.size(0), n = sorted.size(1), P = scale.size(0);
    TORCH_CHECK(prefix.is_cuda() && prefix.scalar_type() == at::kDouble && prefix.is_contiguous() && prefix.numel() == S * (n + 1) * 2,
                "prefix: expected a contiguous float64 HIP tensor [S, n + 1, 2]");
    TORCH_CHECK(scale.numel() == P * S && zp.numel() == P * S, "scale / zp: expected [P, S]");
    at::Tensor scores = at::empty({P, S}, sorted.options());
    check(adalog_score_self_sorted(fptr(sorted, "sorted"), prefix.data_ptr<double>(), S, n, fptr(scale, "scale"), fptr(zp, "zp"), (int)P,
                                   (int)n_bits, norm, scores.data_ptr<float>(), cur_stream()),
          "adalog::score_self_sorted");
    return scores;
}

}  // namespace

TORCH_LIBRARY(adalog, m) {
    m.def("uniform_fake_quant(Tensor x, Tensor scale, Tensor? zero_point, int n_ch, int inner, int n_bits, bool sym) -> Tensor");
    m.def("log_fake_quant(Tensor x, Tensor scale, Tensor q, Tensor table1, Tensor table2, int n_bits, Tensor? shift, bool sub_shift) -> Tensor");
    m.def("log2_shift(Tensor x, float shift) -> Tensor");
    m.def("score_act_fused(Tensor wp, Tensor x2, Tensor lx2, Tensor ref2, Tensor row_scale, Tensor? row_bias, Tensor scale, Tensor qv, "
          "int n_bits, Tensor mant37, float shift, bool clamp_u, float sa_mul, float norm) -> Tensor");
    m.def("topk(Tensor scores, int k) -> Tensor");
    m.def("pack_uniform(Tensor x3, Tensor scale, Tensor zero_point, int C, int pc, int gmod, int pg, int pr, int n_bits, int dtype, "
          "int Kp, bool want_rowsum, bool c_inner) -> (Tensor, Tensor)");
    m.def("pack_adalog(Tensor x3, Tensor scale, Tensor qv, int C, int pc, int gmod, int pg, int n_bits, Tensor mant37, Tensor? shift, "
          "bool clamp_u, int Kp, bool c_inner) -> Tensor");
    m.def("gemm_score(int dtype, Tensor A, Tensor B, int M, int N, int C, int G, int gmod, int k_valid, Tensor ref, Tensor sa, int sa_c, "
          "int sa_g, float sa_mul, Tensor sb, int sb_c, int sb_g, int sb_n, Tensor? bias, int bi_c, int bi_g, int bi_n, bool keep_h, "
          "bool keep_n, float norm, int ref_div, int order, bool ref_transposed, Tensor? row_scale, Tensor? row_bias) -> Tensor");
    m.def("gemm_score_partial(int dtype, Tensor A, Tensor B, int M, int N, int C, int G, int gmod, int k_valid, Tensor ref, Tensor sa, "
          "int sa_c, int sa_g, float sa_mul, Tensor sb, int sb_c, int sb_g, int sb_n, Tensor? bias, int bi_c, int bi_g, int bi_n, "
          "bool keep_h, bool keep_n, float norm, int ref_div, int order, bool ref_transposed, Tensor? row_scale, Tensor? row_bias) -> Tensor");
    m.def("finish_topk_next(Tensor partial, int MT, int n_last, int Npad, int C, int G, int gmod, bool keep_h, bool keep_n, int mode, "
          "float norm, int k, Tensor scale, Tensor? zp, Tensor? third, int new_cnt, Tensor? lin, Tensor(a!)? delta, bool has_clamp, "
          "float clamp_min) -> (Tensor, Tensor, Tensor, Tensor)");
    m.def("topk_next(Tensor scores, Tensor scale, Tensor? zp, Tensor? third, int k, int new_cnt, Tensor? lin, Tensor(a!)? delta, "
          "bool has_clamp, float clamp_min) -> (Tensor, Tensor, Tensor)");
    m.def("score_w_self(Tensor w2, Tensor scale, Tensor zp, int n_bits) -> Tensor");
    m.def("score_a_self(Tensor x2, Tensor scale, Tensor zp, bool channel_wise, int n_bits, float norm) -> Tensor");
    m.def("score_act_gen(int dtype, Tensor wp, Tensor x2, Tensor scale, Tensor zp, int n_bits, Tensor ref2, Tensor row_scale, "
          "Tensor? row_bias, float norm) -> Tensor");
    m.def("score_act_gen_partial(int dtype, Tensor wp, Tensor x2, Tensor scale, Tensor zp, int n_bits, Tensor ref2, Tensor row_scale, "
          "Tensor? row_bias) -> Tensor");
    m.def("sorted_prefix(Tensor x2) -> (Tensor, Tensor)");
    m.def("score_self_sorted(Tensor sorted, Tensor prefix, Tensor scale, Tensor zp, int n_bits, float norm) -> Tensor");
}

// HIP dispatch key only ("CUDA" is the HIP key on ROCm builds of PyTorch): there is deliberately no CPU implementation
TORCH_LIBRARY_IMPL(adalog, CUDA, m) {
    m.impl("uniform_fake_quant", &uniform_fake_quant);
    m.impl("log_fake_quant", &log_fake_quant);
    m.impl("log2_shift", &log2_shift);
    m.impl("score_act_fused", &score_act_fused);
    m.impl("topk", &topk);
    m.impl("pack_uniform", &pack_uniform);
    m.impl("pack_adalog", &pack_adalog);
    m.impl("gemm_score", &gemm_score);
    m.impl("gemm_score_partial", &gemm_score_partial);
    m.impl("finish_topk_next", &finish_topk_next);
    m.impl("topk_next", &topk_next);
    m.impl("score_w_self", &score_w_self);
    m.impl("score_a_self", &score_a_self);
    m.impl("score_act_gen", &score_act_gen);
    m.impl("score_act_gen_partial", &score_act_gen_partial);
    m.impl("sorted_prefix", &sorted_prefix);
    m.impl("score_self_sorted", &score_self_sorted);
}
