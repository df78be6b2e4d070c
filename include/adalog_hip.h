/* adalog_hip.h -- C ABI of the MI355X (gfx950) AdaLog calibration kernels  (libadalog_hip.so)
 *
 * The reference (GoatWu/AdaLog) has no FFI: its hot path is Python calling ATen ops.  The drop-in boundary is
 * therefore the Python module API (quantizers / quant_layers / utils.calibrator, mirrored in adalog_amd/); THIS header
 * is the layer directly beneath it -- what a maintainer of the reference would bind with ctypes to replace the bodies
 * of the functions cited below (INTEGRATION.md shows the stubs).  Plain pointers and sizes only, no torch types.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer (HIP, same GPU as `stream`) unless the name ends in _host;
 *   - `stream` is a hipStream_t passed as void*; all work is enqueued on it, nothing synchronises the host;
 *   - return value: 0 = enqueued; -1 = rejected arguments; otherwise the hipError_t of the failed call.
 *     adalog_last_error() returns a thread-local description;
 *   - candidate tensors are fp32 [P][cols], candidate-major (P = eq_n = 128 in the shipped configs);
 *   - "n_bits" is the quantiser bit-width b; L = 2^(b-1) levels per sign, the integer grid is [0, 2L-1].
 */
#ifndef ADALOG_HIP_H
#define ADALOG_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

int adalog_abi_version(void);
const char* adalog_last_error(void);
/* measurement aid: name of the scoring kernel the last adalog_gemm_score / adalog_score_act_fused call of this thread launched */
const char* adalog_last_kernel(void);

/* ---- K1  UniformQuantizer.forward, eval form                      reference quantizers/uniform.py:25-36
 * y = (clamp(rne(x/s) + rne(zp), 0, 2L-1) - rne(zp)) * s   (asymmetric), or clamp(rne(x/s), -L, L-1) * s (symmetric).
 * Broadcast of scale/zero_point: channel(i) = (i / inner) % n_channels
 *   per-tensor            n_channels = 1
 *   per-row weights       x = W.view(rows, I):      n_channels = rows, inner = I      (linear.py:90-92, conv.py:115-120)
 *   per-head [N,H,S,C]    n_channels = H, inner = S*C                                 (matmul.py:129-133)
 *   per-channel [.., I]   n_channels = I, inner = 1                                   (linear.py:566-568)
 * y and bins (uint8 integer bin index, asymmetric only) are each optional (NULL). */
int adalog_uniform_fake_quant_f32(const float* x, float* y, uint8_t* bins, int64_t n, const float* scale,
                                  const float* zero_point, int64_t n_channels, int64_t inner, int n_bits, int symmetric,
                                  void* stream);

/* ---- K2/K3  AdaLogQuantizer.forward / ShiftAdaLogQuantizer.forward   reference quantizers/logarithm.py:83-99,127-135
 * k = rne(-log2(clamp((x+shift)/s, 1e-15, 1)) * 37 / q);  y = 2^-table1[k] * table2[k] * s * [k < 2L]  [- shift]
 * scale: [1]; q: int64 [1] (the quantiser's buffer); table1/table2: fp32 [2L] (update_table, logarithm.py:77-81);
 * shift: [1] or NULL; sub_shift: subtract the shift again (bias not yet re-parameterised);
 * train_form: y = 2^(-k*q/37) * s (logarithm.py:88-92, no LUT).  bins: k, or 255 where masked. */
int adalog_log_fake_quant_f32(const float* x, float* y, uint8_t* bins, int64_t n, const float* scale, const int64_t* q,
                              const float* table1, const float* table2, int n_bits, const float* shift, int sub_shift,
                              int train_form, void* stream);

/* ---- operand packing for the scoring GEMMs (prologue of K7/K8/K11-K15)
 * Input: fp32 view x[g][r][k] with element strides (sxg, sxr, sxk), g < G, r < R, k < K.
 * Output: out[c][g][r][Kp], K-contiguous, zero padded to Kp (Kp * sizeof(elem) a multiple of 128 bytes = one cache line);
 *   adalog_pack_uniform / adalog_pack_adalog_bf16 with c_inner = 1 write out[g][r][c][Kp] instead (candidates innermost, see ref_div below).
 * Parameter addressing for candidate c: idx = c*pc + (g % gmod)*pg + r*pr.
 * out_dtype: 0 = int8, 1 = bf16, 2 = fp32.
 *
 * adalog_pack_uniform: value = clamp(rne(x/s)+rne(zp), 0, 2L-1) - rne(zp)  (an exact small integer)
 *   replaces the per-candidate fp32 copies at linear.py:369-371,409-411; matmul.py:150-151,188-189; conv.py:240-242
 *   rowsum (optional int32 [C][G][R]) = sum_k value, used to fold the post-GELU shift into the bias (linear.py:1002-1005).
 * adalog_pack_adalog_bf16: value = m * 2^-t with k = rne(-log2(u)*37/q_c), t = floor(k*q_c/37), m = mant37[(k*q_c) % 37],
 *   u = (x + shift)/s_c [clamped to [1e-15,1] when clamp_u];  masked bins (k >= 2L) -> 0
 *   replaces linear.py:830-836,872-878,913-919 and matmul.py:337-342.  mant37: fp32 [37] integer numerators
 *   round(2^(-j/37) * (4L-2))  (linear.py:750-752).
 * adalog_pack_raw_f32: zero-padded fp32 copy (conv input with qconv_a_bit = 8, conv.py:55-58).
 * adalog_pack_split3_bf16: the same unquantised operand as three bf16 terms, out[row] = [hi(Kt) | mid(Kt) | lo(Kt)] with
 *   hi + mid + lo == x exactly (8 + 8 + 8 mantissa bits; zero past K).  Scored against an exact-integer bf16 candidate operand
 *   repeated three times along K, the bf16 MFMA accumulates the products the fp32 MFMA would, at 5x its rate. */
int adalog_pack_uniform(const float* x, int64_t G, int64_t R, int64_t K, int64_t sxg, int64_t sxr, int64_t sxk,
                        const float* scale, const float* zero_point, int64_t C, int64_t pc, int64_t gmod, int64_t pg,
                        int64_t pr, int n_bits, int out_dtype, void* out, int64_t Kp, int32_t* rowsum, int c_inner,
                        void* stream);
int adalog_pack_adalog_bf16(const float* x, int64_t G, int64_t R, int64_t K, int64_t sxg, int64_t sxr, int64_t sxk,
                            const float* scale, const float* qv, int64_t C, int64_t pc, int64_t gmod, int64_t pg,
                            int n_bits, const float* mant37, const float* shift, int clamp_u, void* out, int64_t Kp,
                            int c_inner, void* stream);
int adalog_pack_raw_f32(const float* x, int64_t G, int64_t R, int64_t K, int64_t sxg, int64_t sxr, int64_t sxk, void* out,
                        int64_t Kp, void* stream);
int adalog_pack_split3_bf16(const float* x, int64_t G, int64_t R, int64_t K, int64_t sxg, int64_t sxr, int64_t sxk, void* out,
                            int64_t Kt, void* stream);

/* ---- K7/K8/K11-K15  scoring GEMM with fused squared-error epilogue (MFMA)
 * For candidate c < C and group g < G (gh = g % gmod):
 *     D = A[c][g] (M x Kp) . B[c][g]^T (N x Kp)          A/B strides sAc,sAg,sBc,sBg in elements, 0 = shared
 *     out[m][n] = D[m][n] * (sa[c*sa_c + gh*sa_g] * sa_mul * sb[c*sb_c + gh*sb_g + n*sb_n]) + bias[c*bi_c + gh*bi_g + n*bi_n]
 *     partial[c][g][m_tile][n] = sum over the tile's rows of (ref[g*sRg + m*ldr + (n/ref_div)*ref_cs] - out[m][n])^2
 *       (ldr = N, ref_cs = 1 for a row-major reference; ldr = 1, ref_cs = M' reads it transposed)
 * and/or `out` is stored (quant_forward, linear.py:46-51 / matmul.py:43-45 / conv.py:60-65).
 * Replaces F.linear / @ / F.conv2d + _get_similarity + mean/sum in
 *   linear.py:355-384, 394-423, 816-848, 856-890, 898-931; matmul.py:135-163, 173-201, 321-351; conv.py:226-255.
 * ref_div > 1 (weight searches; needs C = 1): GEMM column j encodes (output channel n = j / ref_div, candidate
 *   c = j % ref_div) -- B packed with c_inner = 1 -- so the 128 candidates of a channel share one reference column;
 *   sa/sb/bias are then addressed with (c, n) and partial is laid out [ref_div][G][m_tile][N/ref_div padded].
 * reduce_cols = 1: the tile's column sums are added up in-kernel (fixed order) and partial holds one value per tile,
 *   [C][G][m_tile][n_tile] -- for searches whose score does not keep the column axis.
 * order: workgroup -> tile order, fastest index first (L2 reuse): 0 = n,m,g,c   1 = n,c,m,g   2 = m,n,c,g.
 * dtype: 0 = int8 (exact integer dot products), 1 = bf16, 2 = fp32, 3 = fp8 e4m3 holding exact integers of <= 4-bit operands,
 *   4 = A bf16 rows x B fp8 columns, converted to bf16 in registers (the shapes adalog_gemm_mixed_ok accepts; sAc / sBc ... in
 *   elements of the respective operand).  bias may be NULL.  partial and ref go together.
 * Kp: row stride of both packed operands in elements (a multiple of 128 bytes); k_valid: leading elements of a row that
 *   can be non-zero (the pack kernels zero-fill [K, Kp)); 0 means Kp.  The streaming kernel skips whole 64-byte
 *   K-steps of padding (q.k^T with head_dim 64: half of the padded row).
 * partial must hold adalog_gemm_score_layout(M, N, C, G, ...) floats. */
/* adalog_gemm_mixed_ok: 1 when adalog_gemm_score takes dtype 4 -- A: bf16 rows [..][Kp], B: fp8 e4m3 candidate columns
 *   [..][Kp] (q - z of a <= 4-bit quantiser: exact) -- for this shape (N includes the candidate factor ref_div; C = 1,
 *   transposed reference).  Three families, all searches whose FIXED operand holds AdaLog values (bf16) and whose candidates are
 *   uniform:  k_valid 193..256, 129..224 rows, >= 8 groups (softmax.v weight search of a 197-token ViT, reference matmul.py:173-201;
 *   Kp = 256);  k_valid <= 64, <= 64 rows, >= 256 groups (the same search over swin windows; Kp = 64);  k_valid > 256 on the wide
 *   streaming form (weight search of the post-GELU layer, reference linear.py:355-392; Kp = any multiple of 64).  The kernels convert
 *   the fp8 fragments with v_cvt_scalef32_pk_bf16_fp8, so the streamed candidate operand has half the bytes of a bf16 one. */
int adalog_gemm_mixed_ok(int M, int N, int G, int gmod, int ref_div, int64_t k_valid);
int adalog_gemm_score(int dtype, const void* A, const void* B, int64_t sAc, int64_t sAg, int64_t sBc, int64_t sBg, int M,
                      int N, int64_t Kp, int64_t k_valid, int C, int G, int gmod, const float* ref, int64_t ldr, int64_t sRg, int64_t ref_cs,
                      int ref_div, const float* sa, int64_t sa_c, int64_t sa_g, float sa_mul, const float* sb, int64_t sb_c,
                      int64_t sb_g, int64_t sb_n, const float* bias, int64_t bi_c, int64_t bi_g, int64_t bi_n,
                      const float* row_scale, const float* row_bias, float* partial, int64_t partial_elems, float* out,
                      int64_t ldo, int64_t sOc, int64_t sOg, int order, int reduce_cols, void* stream);
/* row_scale / row_bias (optional, C = 1): out[m][n] = (D*alpha) * row_scale[m] + row_bias[m] + bias -- the activation
 * searches run transposed (rows = output channels, columns = (token, candidate)), so the per-channel weight scale and
 * the layer bias are per-ROW there.
 * Layout of `partial`: [C][G][MT][Npad] for C > 1 launches, [G][MT][Npad][ref_div] (candidate innermost: coalesced
 * stores) for ref_div > 1 launches -- pass cand_inner = 1 to adalog_finish_scores for the latter; adalog_gemm_score_layout returns its size and (MT, Npad) for the given
 * problem (tilings: 128x128 for C > 1; (64..256)x256 when C = 1 -- the row count of a tile depends on M and, for the
 * streaming kernel, on the dtype / k_valid / reference orientation of the launch, so pass the same values here; for
 * the slab kernel -- int8 / fp8 storage, K <= 384 bytes, one group -- MT is the number of pieces a 256-column slab can be
 * cut into by the workgroups' ranges, and the launch clears `partial` itself). */
int64_t adalog_gemm_score_layout(int M, int N, int C, int G, int gmod, int ref_div, int reduce_cols, int dtype, int64_t Kp,
                                 int64_t k_valid, int ref_transposed, int* MT, int* Npad, int* mode);
/* -> number of floats `partial` must hold (8-byte aligned); *mode is what to pass as `cand_inner` to adalog_finish_scores:
 *    0 = [C][G][MT][Npad]; 1 = [G][MT][Npad][ref_div] (candidate innermost); 2 = per-workgroup fp64 accumulators
 *    [MT = workgroups][gmod][256] (reduce_cols = 1 with ref_div > 1 on the streaming kernel: the column axis is summed
 *    inside the launch, in a fixed order, and no per-tile partial is written). */

/* scores[c][h?][n?] = -norm * sum_{image = g/gmod, (h), m_tile, (n < N)} partial[c][g][m_tile][n] (layout MT, Npad from
 * adalog_gemm_score_layout; C = c_eff), accumulated in fp64 in a
 * fixed order (deterministic).  keep_h / keep_n select which axes survive:
 *   Linear weight search  keep_n=1 -> [P][O]   (linear.py:378-385, norm = 1/T)
 *   Linear act. search    keep_n=0 -> [P]      (linear.py:415-424, norm = 1/(T*O))
 *   MatMul per head       keep_h=1 -> [P][H]   (matmul.py:154-164, norm = 1/(S*S'))
 *   post-softmax base     none     -> [P]      (matmul.py:345-352, norm = 1/(H*S*S')) */
/* 1 when a scoring launch of this shape (C = 1, reduce_cols = 1, transposed reference) runs on the window kernel -- many
 * small groups of <= 64 rows with one K-step, e.g. swin's 49 x 49 x 32 attention windows (reference matmul.py:135-209 on
 * window attention) -- in which case int8 / fp8 operands of K <= 32 may be packed with 32-byte rows (Kp = 32). */
int adalog_gemm_win_ok(int dtype, int M, int N, int G, int gmod, int ref_div, int64_t k_valid);
/* Attention searches with uniform candidates (reference quant_layers/matmul.py:135-163 for operand A, :173-201 for B), GEN form:
 *   what adalog_gemm_score computes from the packed candidate operand [G][N][Kp] (N = source rows x ref_div candidates,
 *   candidates innermost) without that operand -- the window / wave-private group kernels quantise the fp32 tensor
 *   x [G][N / ref_div][K = k_valid] (rows ldx, groups sg apart; 16-byte aligned, multiples of 4) in registers with the candidates'
 *   (sb, zp)[c * sb_c + head * sb_g] and n_bits, bit for bit what adalog_pack_uniform would have written (int8: dtype 0; fp8:
 *   dtype 3, n_bits <= 4).  A: the packed fixed operand [G][M][Kp] (groups sAg elements apart); ref: [G][N / ref_div][M]
 *   (transposed, groups sRg apart); partial: the per-workgroup fp64 accumulators of adalog_gemm_score_layout (reduce_cols = 1).
 *   adalog_gemm_score_gen_ok: 1 for the shapes taken (K = 16, 32, 48 or 64; the shapes of adalog_gemm_win_ok and of the
 *   wave-private q.k^T kernel). */
int adalog_gemm_score_gen(int dtype, const void* A, int64_t sAg, int M, int N, int64_t Kp, int64_t k_valid, int G, int gmod,
                          const float* x, int64_t ldx, int64_t sg, const float* zp, int n_bits, const float* ref, int64_t sRg,
                          int ref_div, const float* sa, int64_t sa_c, int64_t sa_g, float sa_mul, const float* sb, int64_t sb_c,
                          int64_t sb_g, float* partial, int64_t partial_elems, void* stream);
int adalog_gemm_score_gen_ok(int dtype, int M, int N, int G, int gmod, int ref_div, int64_t k_valid, int64_t Kp);
/* softmax.v, log-base search of the post-softmax AdaLog quantiser (reference quant_layers/matmul.py:321-351): scores of the P = 128
 *   candidate bases q[P] with the candidate operand -- P AdaLog quantisations (scale 1, no clamp: logarithm.py:83-99 as the search
 *   applies it) of the probabilities x [G][N][K] (fp32, rows ldx, groups sg apart) -- generated inside the kernel instead of
 *   packed by adalog_pack_adalog_bf16 and streamed (2.2 GB per launch for deit_small).  A: the fixed operand v^T as packed bf16
 *   [G][M <= 64][Kp] (groups sAg elements apart); lut: dword table [2^n_bits + 1][P], entry [k][c] = bf16 bits of bin k's value
 *   under base q[c] (numerator * 2^-t, exactly what the packer writes), last row 0 (the masked code); ref [G][N][M] (groups sRg
 *   apart); sa / sa_mul / sb: the epilogue factors of adalog_gemm_score; partial: the per-workgroup fp64 accumulators of
 *   adalog_gemm_score_layout(M, N * P, 1, G, gmod, P, 1, dtype 1, Kp, k_valid, 1).  K <= 208 (197-token ViTs: 13 K-steps; windows: 4). */
int adalog_gemm_score_avq(const void* A, int64_t sAg, int M, int N, int64_t Kp, int64_t k_valid, int G, int gmod, const float* x,
                          int64_t ldx, int64_t sg, const float* q, const uint32_t* lut, int n_bits, const float* ref, int64_t sRg,
                          int P, const float* sa, int64_t sa_c, int64_t sa_g, float sa_mul, const float* sb, int64_t sb_c,
                          int64_t sb_g, float* partial, int64_t partial_elems, void* stream);
int adalog_gemm_score_avq_ok(int M, int N, int G, int gmod, int P, int64_t k_valid, int64_t Kp, int n_bits);
int adalog_finish_scores(const float* partial, float* scores, int MT, int N, int Npad, int C, int G, int gmod, int keep_h,
                         int keep_n, int cand_inner, double norm, void* workspace, int64_t workspace_bytes, void* stream);
/* Scratch for the two-stage form used when cand_inner = 1 and keep_n = 0 (sums of 10^4..10^5 terms per candidate): bytes
 * to pass as `workspace` (0: not needed; a NULL / short workspace falls back to the one-pass kernels). */
int64_t adalog_finish_workspace_bytes(int MT, int N, int C, int G, int keep_n, int cand_inner);
/* adalog_finish_scores followed by adalog_topk_next (reference linear.py:483-523 after a scoring call) -- in ONE launch where the
 * partial layout allows it (per-workgroup accumulators: the last block to finish ranks; candidate-innermost per-column partials: a
 * block holds all candidates of its columns), otherwise as the two launches.  Single-GPU form (with several ranks the scores are
 * all-reduced between the two steps).  `scores` [C][cols] is still written. */
int adalog_finish_topk_next(const float* partial, float* scores, int MT, int N, int Npad, int C, int G, int gmod, int keep_h,
                            int keep_n, int cand_inner, double norm, void* workspace, int64_t workspace_bytes, int k,
                            const float* scale, const float* zp, const float* third, int new_cnt, const float* lin, float* delta,
                            int has_clamp, float clamp_min, float* out_scale, float* out_zp, float* out_third, void* stream);

/* ---- K11 fused  post-GELU activation-candidate search with the AdaLog quantisation inside the GEMM's loader
 *                                               reference linear.py:816-848, :856-890, :898-931 (one scoring call)
 * scores[p] = -norm * sum_{t < T, o < M} ( (ref[t][o] - row_bias[o]) - row_scale[o] * (scale[p] * sa_mul) *
 *                                         sum_k Wp[o][k] * m_p(x[t][k]) )^2            for the P = 128 candidates (scale[p], qv[p]),
 * m_p(x) = integer-numerator form of the search-time AdaLog value (linear.py:831-836):
 *          k = rne(-log2(clamp((x + shift) / scale[p], 1e-15, 1)) * 37 / qv[p]);  0 if k >= 2^n_bits;
 *          else 2^-floor(k qv / 37) * mant37[(k qv) mod 37].
 * Same result as adalog_pack_adalog_bf16(c_inner) + adalog_gemm_score + adalog_finish_scores (bins are exact: near-ties
 * of the fast evaluation are re-evaluated with the IEEE divide / correctly rounded log2 pipeline and the accumulators
 * corrected), but the [T*P][K] candidate operand is never written: HBM traffic = Lx + ref + Wp.
 *   Wp: bf16 image of the quantised weight, [M][Kp] (adalog_pack_uniform, out_dtype 1);  x: [T][K] fp32;
 *   Lx = adalog_log2_shift(x, shift): [T][K] fp32, correctly rounded log2(x + shift), -inf where x + shift <= 0;
 *   ref: raw_out [T][M];  workspace: adalog_score_act_fused_workspace_bytes(T, Kp) bytes, 8-byte aligned (partial sums and
 *   the threshold table).
 * adalog_score_act_fused_ok: 1 when the shape is taken (P = 128, n_bits <= 6, >= 6 K-steps of 32, LDS budget), else the
 * caller uses the packed path. */
int adalog_log2_shift(const float* x, float* out, int64_t n, float shift, void* stream);
int adalog_score_act_fused_ok(int M, int64_t T, int K, int64_t Kp, int P, int n_bits);
int64_t adalog_score_act_fused_workspace_bytes(int64_t T, int64_t Kp);
int adalog_score_act_fused(const void* Wp, int M, int64_t Kp, const float* x, const float* Lx, int64_t T, int K,
                           const float* ref, const float* row_scale, const float* row_bias, const float* scale,
                           const float* qv, int P, int n_bits, const float* mant37, float shift, int clamp_u, float sa_mul,
                           double norm, void* workspace, int64_t workspace_bytes, float* scores, void* stream);

/* ---- K16  FPCS driver pieces                 reference linear.py:483-523, matmul.py:243-262, conv.py:292-311
 * adalog_topk: idx[j][col] = candidate with the j-th best score of column col, j < k; order (score desc, index asc),
 *   i.e. torch.topk(sorted=True) with ties made deterministic.  scores: [P][cols], P <= 256.
 * adalog_fpcs_next: gathers the survivors and
 *   new_cnt > 0: writes the next survivor-major grid  out[(j*new_cnt+i)][col] = top_scale[j][col] + (lin[i]-0.5)*delta[col]
 *                (clamped below when has_clamp), copies zero points / third plane, then delta[col] /= (new_cnt - 0.5);
 *   new_cnt = 0: commits the winner (k = 1) to out_scale/out_zp/out_third [cols]          (linear.py:387-391).
 * adalog_candidate_grid: the initial percentile grid                    (linear.py:442-451,472-481; matmul.py:231-240)
 *   quant4 = [4][cols] {Q_hi0, Q_hi1, Q_lo0, Q_lo1};  scale[(zi*num_scale+si)][col] = (dmin + lin[si]*(dmax-dmin))/(2L-1),
 *   zp = zp_min + zi, delta[col] = scale[1][col] - scale[0][col]. */
int adalog_topk(const float* scores, int P, int cols, int k, int* idx, void* stream);
/* The tail of one FPCS step as ARGUMENTS (round 6; csrc/fpcs_tail.h holds the device code): rank the P scores of every column, then
 * write the next k x new_cnt grid around the survivors (new_cnt > 0) or commit the winner (new_cnt == 0, k == 1) -- reference
 * linear.py:483-523.  Every kernel that produces FINAL scores takes it and runs it in its own launch (the last workgroup / wave to
 * finish a column draws the column's ticket and ranks it): adalog_gram_score_w_tail, adalog_gram_act_score_tail,
 * adalog_score_self_sorted_tail; adalog_topk_next_tail / adalog_finish_topk_next_tail are the stand-alone forms.
 *   scale / zp / third: the grid that was scored, [P][cols] (zp, third may be null; the in / out planes must match);
 *   lin: linspace(0, 1, new_cnt); delta_in [cols] is read, delta_out [cols] receives delta_in / (new_cnt - 0.5) (they may alias);
 *   out_*: [k * new_cnt][cols], or [cols] for the committed winner -- which may be the quantiser's own parameter storage. */
typedef struct adalog_fpcs_tail {
    int32_t k, new_cnt, has_clamp;
    float clamp_min;
    const float* scale; const float* zp; const float* third;
    const float* lin;
    const float* delta_in; float* delta_out;
    float* out_scale; float* out_zp; float* out_third;
} adalog_fpcs_tail;
int adalog_topk_next_tail(const float* scores, int P, int cols, const adalog_fpcs_tail* tail, int* idx_out, void* stream);
/* adalog_finish_topk_next with the tail as a struct (delta_in / delta_out apart, commit into caller-owned storage) */
int adalog_finish_topk_next_tail(const float* partial, float* scores, int MT, int N, int Npad, int C, int G, int gmod, int keep_h,
                                 int keep_n, int cand_inner, double norm, void* workspace, int64_t workspace_bytes,
                                 const adalog_fpcs_tail* tail, void* stream);
/* adalog_topk_next = adalog_topk followed by adalog_fpcs_next in one launch (one workgroup per column); idx_out (optional)
 * receives the top-k indices [k][cols]. */
int adalog_topk_next(const float* scores, int P, int cols, int k, const float* scale, const float* zp, const float* third,
                     int new_cnt, const float* lin, float* delta, int has_clamp, float clamp_min, float* out_scale,
                     float* out_zp, float* out_third, int* idx_out, void* stream);
int adalog_fpcs_next(const float* scale, const float* zp, const float* third, int cols, const int* idx, int k, int new_cnt,
                     const float* lin, float* delta, int has_clamp, float clamp_min, float* out_scale, float* out_zp,
                     float* out_third, void* stream);
int adalog_candidate_grid(const float* quant4, int cols, int num_scale, int num_zp, int zp_min, int n_bits, const float* lin,
                          int has_clamp, float clamp_min, float* scale, float* zp, float* delta, void* stream);

/* ---- K8 with the candidate operand generated in the kernel            reference linear.py:394-430 (_search_best_a_scale)
 * One activation-candidate scoring call of a uniformly quantised Linear layer WITHOUT a packed candidate operand: the slab
 * kernel quantises the two (or one / four) activation rows of a slab for all P candidates straight from the fp32 activation.
 *   scores[p] = -norm * sum_{t, o} (ref[t][o] - row_bias[o] - row_scale[o] * s_p * sum_k Wq[o][k] * xq_p[t][k])^2,
 *   xq_p = clamp(rne(x / s_p) + z_p, 0, 2^bits - 1) - z_p.
 * Wp: packed weight image [M][Kp] (adalog_pack_uniform, dtype 0 = int8 or 3 = fp8 e4m3 for <= 4 bit); x: fp32 [T][ldx], K valid
 * (K % 16 == 0); ref: fp32 [T][M]; scale / zp: [P], P in {64, 128, 256}; row_bias may be null.  adalog_score_act_gen_ok says
 * whether the shape is taken (else: adalog_pack_uniform + adalog_gemm_score).  workspace: 16-byte aligned,
 * adalog_score_act_gen_workspace_bytes(...) bytes.  scores == NULL: the fp64 accumulators [wgs][1][256] are left at the start of
 * the workspace for adalog_finish_scores / adalog_finish_topk_next (cand_inner = 2, MT = adalog_score_act_gen_wgs, N = Npad = 256). */
int adalog_score_act_gen_ok(int dtype, int M, int64_t T, int K, int64_t Kp, int P);
int64_t adalog_score_act_gen_workspace_bytes(int dtype, int M, int64_t T, int K, int64_t Kp, int P);
int adalog_score_act_gen_wgs(int dtype, int M, int64_t T, int K, int64_t Kp, int P);   /* workgroups = MT of the accumulators; -1: not taken */
int adalog_score_act_gen(int dtype, const void* Wp, int M, int64_t Kp, const float* x, int64_t T, int K, int64_t ldx,
                         const float* scale, const float* zp, int P, int n_bits, const float* ref, const float* row_scale,
                         const float* row_bias, double norm, void* workspace, int64_t workspace_bytes, float* scores,
                         void* stream);
/* The weight-candidate scoring call of a uniformly quantised Linear (reference quant_layers/linear.py:355-392) with the candidate
 *   operand GENERATED inside the slab kernel: score[p][o] = -norm * sum_t (raw_out[t][o] - bias[o] - s_a * s_w[p][o] *
 *   sum_k (q_a(x) - z_a)[t][k] * (clamp(rne(W[o][k] / s_w[p][o]) + z_w[p][o], 0, 2^bits - 1) - z_w[p][o]))^2.
 *   Xp: packed activation image [T][Kp] (dtype 0 int8 / 3 fp8); W fp32 [O][ldw]; scale / zp [P][O]; ref = raw_out transposed
 *   [O][T]; sa: device scalar s_a; bias [O] or null.  No packed [O*P][Kp] candidate operand is written or read.  partial: the
 *   buffer adalog_gemm_score_layout(T, O*P, 1, 1, 1, P, 0, dtype, Kp, K, 1) describes (finish with adalog_finish_scores /
 *   adalog_finish_topk_next, keep_n = 1).  adalog_score_w_gen_ok: whether the shape is taken (K % 16 == 0, T % 32 == 0, ...). */
int adalog_score_w_gen_ok(int dtype, int T, int O, int K, int64_t Kp, int P);
int adalog_score_w_gen(int dtype, const void* Xp, int T, int64_t Kp, const float* W, int O, int K, int64_t ldw, const float* scale,
                       const float* zp, int P, int n_bits, const float* ref, const float* sa, const float* bias, float* partial,
                       int64_t partial_elems, void* stream);

/* ---- K7, Gram form   _search_best_w_scale scored from the Gram matrix       reference quant_layers/linear.py:355-392,483-503
 * The activation quantiser is fixed for the six scoring calls of a weight_fpcs call, and the score of candidate p for output row o
 *   sum_t (r[t][o] - sigma x_int[t][:] . w[:])^2,  r = raw_out - bias, sigma = s_a s_w[p][o], w = clamp(rne(W[o]/s_w)+z_w, 0, 2^b-1) - z_w
 * is the quadratic form  S0[o] - 2 sigma (w . c[o]) + sigma^2 (w^T G w)  with G = X_int^T X_int [K][K], c[o] = X_int^T r[:, o],
 * S0[o] = sum_t r^2.  adalog_gram_build computes G, c, S0 ONCE per weight_fpcs call into `workspace` (exact integers: G in balanced
 * int8 limbs, r rounded once per column to 30-bit fixed point; csrc/gram.hip has the error analysis); adalog_gram_score_w then scores
 * one FPCS step's P candidates of every output row from it: K^2 multiply-adds per candidate row and limb instead of T K.
 *   x fp32 [T][ldx] (K valid); sa / za: activation scale / zero point (device scalars); ref_t = raw_out TRANSPOSED [O][T];
 *   bias [O] or null; W fp32 [O][ldw]; scale / zp [P][O]; scores [P][O] = -norm * (the sum above) -- final scores, no partials.
 * adalog_gram_supported: the shape can be scored this way (K % 32 == 0 and instantiated, <= 7-bit operands, P % 32 == 0);
 * adalog_gram_ok: ... and it pays (limbs * K <= T / 2: at most half the token form's multiply-adds).
 * The (T, O, K, a_bits) of a score call must be those of the build that filled the workspace (256-byte aligned). */
int adalog_gram_supported(int T, int O, int K, int a_bits, int w_bits, int P);
int adalog_gram_ok(int T, int O, int K, int a_bits, int w_bits, int P);
int64_t adalog_gram_workspace_bytes(int T, int O, int K, int a_bits);
int adalog_gram_limbs(int T, int a_bits);   /* int8 limbs of G: a score call issues limbs * K^2 + 32 K multiply-adds per candidate row */
int adalog_gram_build(const float* x, int T, int K, int64_t ldx, const float* sa, const float* za, int a_bits, const float* ref_t,
                      int O, const float* bias, void* workspace, int64_t workspace_bytes, void* stream);
int adalog_gram_score_w(const float* W, int O, int K, int64_t ldw, const float* scale, const float* zp, int P, int w_bits,
                        const void* workspace, int T, int a_bits, const float* sa, double norm, float* scores, void* stream);
/* The build with the calibration images SHARDED over ranks (reference: scores are sums over images, linear.py:384), in
 * three calls with the collectives between them:  adalog_gram_amax (this rank's column maxima, as float bits: all-reduce MAX as
 * int32), adalog_gram_build_sums (with the GLOBAL amax: gsum [K][K] and csum [O][K] as int64, s0 [O] fp64, of this rank's tokens;
 * workspace = adalog_gram_workspace_bytes of the LOCAL shape: all-reduce SUM of the three), adalog_gram_build_from_sums (-> the
 * workspace adalog_gram_score_w reads, for T_total = the global token count).  Every rank then holds the same state and computes
 * the same FINAL scores: an FPCS step of the search needs no collective. */
int adalog_gram_amax(const float* ref_t, int T, int O, const float* bias, unsigned int* amax, void* stream);
int adalog_gram_build_sums(const float* x, int T, int K, int64_t ldx, const float* sa, const float* za, int a_bits, const float* ref_t,
                           int O, const float* bias, const unsigned int* amax, long long* gsum, long long* csum, double* s0,
                           void* workspace, int64_t workspace_bytes, void* stream);
int adalog_gram_build_from_sums(const long long* gsum, const long long* csum, const double* s0, const unsigned int* amax, int T_total,
                                int O, int K, int a_bits, void* workspace, int64_t workspace_bytes, void* stream);
/* adalog_gram_score_w followed by the FPCS step's tail (tail may be null; its grid is (scale, zp)): two launches (csrc/gram.hip says why) */
int adalog_gram_score_w_tail(const float* W, int O, int K, int64_t ldw, const float* scale, const float* zp, int P, int w_bits,
                             const void* workspace, int T, int a_bits, const float* sa, double norm, float* scores,
                             const adalog_fpcs_tail* tail, void* stream);

/* ---- K8, Gram form   _search_best_a_scale scored from the candidates' Gram matrices   reference quant_layers/linear.py:394-430,505-523
 * The weight quantiser is fixed for the six scoring calls of an activation_fpcs call, and the score of the per-tensor candidate
 * p = (s_p, z_p),  sum_{t,o} (r[t][o] - s_p sum_k Wq[o][k] x_p[t][k])^2  with r = raw_out - bias, Wq = diag(s_w) W_int and
 * x_p = clamp(rne(x / s_p) + z_p, 0, 2^b - 1) - z_p, expands to  S0 - 2 s_p <X_p, C> + s_p^2 <H, X_p^T X_p>  with C = r . Wq [T][K],
 * H = Wq^T Wq [K][K].  Per candidate only G_p = X_p^T X_p is a product: K x K x T / 2 multiply-adds (symmetric) instead of O x K x T,
 * exact on the int8 MFMA; <X_p, C> comes from prefix sums of C along the SORTED activation (a uniform quantiser's levels are runs of
 * it): 2^b + 2 bisections per candidate.  csrc/gram_act.hip has the error analysis.
 *   adalog_gram_act_prepare   once per captured activation x [T][K] (contiguous): xt fp32 [K][Tp] (Tp = T rounded up to 128), the
 *                             sorted values [T K] and the sorting permutation (uint32 flat indices); sort_ws of
 *                             adalog_gram_act_sort_bytes(T K) bytes
 *   adalog_gram_act_build     once per activation_fpcs call: raw_out [T][O], bias [O] | null, W [O][ldw] + its quantiser (s_w, z_w) [O]
 *                             -> workspace (adalog_gram_act_workspace_bytes, 256-byte aligned): H, prefix sums of C, S0
 *   adalog_gram_act_score     one FPCS step: scores [P] = -norm * (the sum above) for candidates (scale, zp) [P]; qpart: scratch of
 *                             P * adalog_gram_act_splits(T, O, K, P) doubles
 * adalog_gram_act_supported: per-tensor candidates, K % 32 == 0 and <= 384 or K = 512 / 768 (instantiated), <= 7-bit operands; adalog_gram_act_ok: ... and
 * enough tokens for it to pay. */
int adalog_gram_act_supported(int T, int O, int K, int a_bits, int w_bits, int P);
int adalog_gram_act_ok(int T, int O, int K, int a_bits, int w_bits, int P);
int64_t adalog_gram_act_workspace_bytes(int T, int O, int K, int P);
int64_t adalog_gram_act_sort_bytes(int64_t n);
int adalog_gram_act_splits(int T, int O, int K, int P);
int adalog_gram_act_prepare(const float* x, int T, int K, int64_t ldx, float* xt, float* sorted, unsigned int* perm, void* sort_ws,
                            int64_t sort_ws_bytes, void* stream);
int adalog_gram_act_build(const float* raw_out, int T, int O, const float* bias, const float* W, int K, int64_t ldw, const float* sw,
                          const float* zw, int w_bits, const unsigned int* perm, int P, void* workspace, int64_t workspace_bytes,
                          void* stream);
int adalog_gram_act_score(const float* xt, const float* sorted, int T, int O, int K, const float* scale, const float* zp, int P, int a_bits,
                          const void* workspace, double norm, double* qpart, float* scores, void* stream);
/* ... and the FPCS step's tail in the finish kernel's last block (tail may be null; its grid is (scale, zp) as [P][1]) */
int adalog_gram_act_score_tail(const float* xt, const float* sorted, int T, int O, int K, const float* scale, const float* zp, int P,
                               int a_bits, const void* workspace, double norm, double* qpart, float* scores,
                               const adalog_fpcs_tail* tail, void* stream);

/* ---- K9   _search_best_w_scale_self                                   reference linear.py:296-309
 * scores[p][row] = -mean_i (w[row][i] - fq_p(w[row][i]))^2,  w: [rows][I], scale/zp: [P][rows]. */
int adalog_score_w_self(const float* w, int rows, int I, const float* scale, const float* zp, int P, int n_bits,
                        float* scores, void* stream);

/* ---- K10  _search_best_a_scale_self                                   reference linear.py:320-345
 * x: [rows][I] read ONCE for all P candidates.  channel_wise: scale/zp [P][I] -> scores [P][I], norm = 1/T;
 * else scale/zp [P][1] -> scores [P][1], norm = 1/(T*I).  partial: scratch of adalog_score_a_self_partial_elems floats. */
int adalog_score_a_self(const float* x, int64_t rows, int I, const float* scale, const float* zp, int P, int channel_wise,
                        int n_bits, double norm, float* partial, int64_t partial_elems, float* scores, void* stream);
int64_t adalog_score_a_self_partial_elems(int64_t rows, int I, int P);

/* ---- K9 / K10 in sorted-prefix form                                   reference linear.py:296-318, 320-353
 * A uniform quantiser is a monotone step function, so the elements a candidate maps to one level are one contiguous run of
 * the SORTED tensor: sort each segment once per captured tensor (per-tensor search: S = 1; per-channel / per-weight-row:
 * one segment each), keep fp64 exclusive prefix sums of x and x^2 along the sorted order, and score a candidate from
 * 2^bits bisections (exact predicate rne(x / s) >= k) + prefix differences instead of quantising every element.
 *   adalog_sorted_prefix_build: x [S][n] -> sorted [S][n], prefix [S][n + 1][2] (double); workspace of
 *     adalog_sorted_prefix_workspace_bytes(S, n) bytes (-1: unsupported size), 16-byte aligned.
 *   adalog_score_self_sorted: scale / zp [P][S] -> scores[p][seg] = -norm * sum_seg (x - fq_p(x))^2. */
int64_t adalog_sorted_prefix_workspace_bytes(int64_t S, int64_t n);
int adalog_sorted_prefix_build(const float* x, int64_t S, int64_t n, float* sorted, double* prefix, void* workspace,
                               int64_t workspace_bytes, void* stream);
int adalog_score_self_sorted(const float* sorted, const double* prefix, int64_t S, int64_t n, const float* scale,
                             const float* zp, int P, int n_bits, double norm, float* scores, void* stream);
/* ... and the FPCS step's tail in the same launch: a ticket per segment (tail may be null; its grid is (scale, zp); S <= 65536) */
int adalog_score_self_sorted_tail(const float* sorted, const double* prefix, int64_t S, int64_t n, const float* scale,
                                  const float* zp, int P, int n_bits, double norm, float* scores, const adalog_fpcs_tail* tail,
                                  void* stream);

/* ---- K15 with the A-side fake quantisation in the GEMM's loader (round 6)      reference linear.py:46-51, matmul.py:43-45
 *   out[g][m][n] = sa[gh * sa_g] * sa_mul * sb[gh * sb_g + n * sb_n] * sum_k (q_a(x[g][m][k]) - z_a) * B[g][n][k] + bias[gh * bi_g + n * bi_n]
 * x: fp32 [G][M][ldx] (groups sxg elements apart; K valid, K % 16 == 0); (a_scale, a_zp)[gh * a_pg]: its uniform quantiser, per tensor
 * (a_pg = 0) or per head (gh = g % gmod); B: packed int8 [G][N][Kp] (adalog_pack_uniform; sBg bytes between groups, 0 = shared);
 * out fp32 [G][M][ldo] (groups sOg apart).  Bit for bit adalog_pack_uniform(x) + adalog_gemm_score(out = ...), in one launch and
 * without the int8 image of the activation. */
int adalog_gemm_out_gen(const float* x, int64_t ldx, int64_t sxg, int K, const float* a_scale, const float* a_zp, int64_t a_pg, int n_bits,
                        const void* B, int64_t sBg, int M, int N, int64_t Kp, int G, int gmod, const float* sa, int64_t sa_g,
                        float sa_mul, const float* sb, int64_t sb_g, int64_t sb_n, const float* bias, int64_t bi_g, int64_t bi_n,
                        float* out, int64_t ldo, int64_t sOg, void* stream);

/* ... + addend[g][m][n] (may be null; indexed like out): the residual stream of a transformer block added in the epilogue */
int adalog_gemm_out_gen_ex(const float* x, int64_t ldx, int64_t sxg, int K, const float* a_scale, const float* a_zp, int64_t a_pg, int n_bits,
                           const void* B, int64_t sBg, int M, int N, int64_t Kp, int G, int gmod, const float* sa, int64_t sa_g,
                           float sa_mul, const float* sb, int64_t sb_g, int64_t sb_n, const float* bias, int64_t bi_g, int64_t bi_n,
                           const float* addend, float* out, int64_t ldo, int64_t sOg, void* stream);
/* K15 from packed operands (dtype 0 = int8, 1 = bf16; A [G][M][Kp], B [G][N][Kp], group strides in elements, 0 = shared) with the
 * same epilogue extras:  out = sa * sa_mul * sb[n] * (A . B^T) + bias[n] + addend.  out_gi > 0: two-level output groups, group g
 * written at (g % out_gi) * sOg + (g / out_gi) * sOo -- softmax . v with out_gi = H, sOg = D, sOo = N * H * D, ldo = H * D writes
 * [B][N][H][D] storage: the transpose(1, 2).reshape(B, N, C) of reference utils/wrap_net.py:31 becomes a view. */
int adalog_gemm_out_ex(int dtype, const void* A, const void* B, int64_t sAg, int64_t sBg, int M, int N, int64_t Kp, int G, int gmod,
                       const float* sa, int64_t sa_g, float sa_mul, const float* sb, int64_t sb_g, int64_t sb_n, const float* bias,
                       int64_t bi_g, int64_t bi_n, const float* addend, float* out, int64_t ldo, int64_t sOg, int out_gi, int64_t sOo,
                       void* stream);

/* ---- quant_forward prologues of a transformer block (round 6): the passes between two products folded into the packers
 * adalog_pack_adalog_bf16_pre: adalog_pack_adalog_bf16 of GELU(x) (pre = 1: x * 0.5 * (1 + erf(x / sqrt 2)), ATen's fp32 expression) --
 *   the activation between fc1 and fc2 (timm Mlp; reference linear.py:770-796 reads GELU's output).  Per-tensor scale, sxk = 1.
 * adalog_softmax_adalog_pack_bf16: (x * mul).softmax(-1) (reference utils/wrap_net.py:26-27), quantised by the post-softmax AdaLog
 *   quantiser (matmul.py:337-343: per-tensor scale, log base 2^(-q/37), u clamped to [1e-15, 1]) into the bf16 operand [rows][Kp]
 *   of softmax . v; the probabilities are never stored.  x fp32 [rows][S] contiguous, S <= Kp <= 256.  ATen's softmax arithmetic,
 *   operation for operation.
 * adalog_attn_split_pack: qkv fp32 [B][N][3][H][64] -> the packed operands of both attention products through their three per-head
 *   (pg = 1) or per-tensor (pg = 0) uniform quantisers: qp, kp int8 [B*H][N][128]; vp bf16 [B*H][64][Np] (v transposed, zero beyond
 *   N; Np a multiple of 64).  Replaces the split / permute copies of wrap_net.py:20-22 and three adalog_pack_uniform launches. */
int adalog_pack_adalog_bf16_pre(const float* x, int64_t G, int64_t R, int64_t K, int64_t sxg, int64_t sxr, int64_t sxk,
                                const float* scale, const float* qv, int64_t C, int64_t pc, int64_t gmod, int64_t pg,
                                int n_bits, const float* mant37, const float* shift, int clamp_u, void* out, int64_t Kp,
                                int c_inner, int pre, void* stream);
/* the same prologue in the training-form quantiser of a BRECQ iteration (block_recon.py:116-121 through fc2's input quantiser,
 * logarithm.py:88-92): y = q(GELU(x)) and its backward (x = the GELU's input; gx through the STE and the GELU's derivative) */
int adalog_log_fake_quant_f32_pre(const float* x, float* y, uint8_t* bins, int64_t n, const float* scale, const int64_t* q,
                                  const float* table1, const float* table2, int n_bits, const float* shift, int sub_shift,
                                  int train_form, int pre, void* stream);
int adalog_log_fq_backward_pre(const float* gy, const float* x, const float* y, float* gx, int64_t n, const float* scale,
                               const int64_t* q, int n_bits, const float* shift, int sub_shift, float* gscale, float* workspace,
                               int pre, void* stream);
/* what precedes a captured BRECQ iteration's replay (block_recon.py:114-117) in one launch: dst_in[b] = src_in[idx[b]],
 * dst_out[b] = src_out[idx[b]] (rows of row_in / row_out floats, multiples of 4), sched_dev[0..n_sched) = sched_row[0..n_sched) */
int adalog_brecq_prepare(const float* src_in, const float* src_out, const int64_t* idx, float* dst_in, float* dst_out, int64_t bs,
                         int64_t row_in, int64_t row_out, const float* sched_row, float* sched_dev, int n_sched, void* stream);
int adalog_softmax_adalog_pack_bf16(const float* x, int64_t rows, int S, float mul, const float* scale, const float* qv, int n_bits,
                                    const float* mant37, void* out, int64_t Kp, void* stream);
int adalog_attn_split_pack(const float* qkv, int B, int N, int H, const float* q_scale, const float* q_zp, int q_bits,
                           const float* k_scale, const float* k_zp, int k_bits, const float* v_scale, const float* v_zp, int v_bits,
                           int pg, void* qp, void* kp, void* vp, int64_t Np, void* stream);

/* ---- stable LSD radix sort of fp32 keys, per segment (csrc/radix_sort.hip; hipCUB until round 5): what the sorted forms above and
 * adalog_gram_act_prepare sort with.  x [S][n] contiguous -> sorted [S][n] (ascending per segment; -0 before +0), perm (may be null)
 * [S][n]: perm[s][i] = index within segment s of its i-th smallest value, equal values in input order.  n <= 8192: one launch (a
 * workgroup per segment, four passes in LDS, no workspace needed); longer segments: 12 launches over 8192-key tiles.  workspace:
 * adalog_sort_workspace_bytes(S, n, perm != NULL) bytes, 256-byte aligned; S <= 65535, S n < 2^31; x and sorted must not overlap. */
int64_t adalog_sort_workspace_bytes(int64_t S, int64_t n, int with_perm);
int adalog_sort_f32(const float* x, int64_t S, int64_t n, float* sorted, unsigned int* perm, void* workspace, int64_t workspace_bytes,
                    void* stream);

/* ---- K5/K6  exact order statistics by radix select
 * adalog_quantile_rows: torch.quantile(x.view(S, n), q, dim=-1, interpolation='linear') for nq <= 4 quantiles, then the mean
 *   over each group of `mbs` consecutive rows (the reference's chunked quantile, linear.py:465-471, matmul.py:219-230).
 *   ranks_lo_hi: int64 [2*nq] = {floor(pos_j), ceil(pos_j)}, weights: fp32 [nq] = pos_j - floor(pos_j), pos_j = q_j*(n-1)
 *   (computed by the host exactly as ATen does).  out: [nq][S/mbs].
 * adalog_positive_percentile_rows: linear.py:763-798 -- value of rank ceil(count*q)-1 among the entries > 0 (0 if none).
 *   qfrac: fp32 [nq].  out: [nq][S].
 * workspace: adalog_select_workspace_bytes(S, R) bytes, R = 2*nq resp. nq. */
int adalog_quantile_rows(const float* x, int64_t S, int64_t n, int nq, const int64_t* ranks_lo_hi, const float* weights,
                         int mbs, float* out, void* workspace, int64_t workspace_bytes, void* stream);
int adalog_positive_percentile_rows(const float* x, int64_t S, int64_t n, int nq, const float* qfrac, float* out,
                                    void* workspace, int64_t workspace_bytes, void* stream);
int64_t adalog_select_workspace_bytes(int64_t S, int R);
/* Sharded form (image-sharded ranks; SURVEY 8e "distributed radix-select with histogram all-reduce"): the same four radix
 * passes, split so the caller can sum the histograms of all ranks (the first S*R*256 uint32 of the workspace) between a
 * pass's counting and its pick.  S, R describe the GLOBAL segments; a rank counts only the rows it holds: local row s
 * belongs to global segment first + (s / inner) * outer + s % inner.  ranks: int64 [R] (NULL with positive_only, where
 * the rank is ceil(count * qfrac) - 1 of the GLOBAL positive count).  Outputs as the fused entry points above. */
int adalog_select_init(void* workspace, int64_t workspace_bytes, int64_t S, int R, const int64_t* ranks, void* stream);
int adalog_select_hist(const float* x, int64_t S_local, int64_t n_local, int first, int inner, int outer, int64_t S, int R,
                       int pass, int positive_only, void* workspace, void* stream);
int adalog_select_pick(void* workspace, int64_t S, int R, int pass, const float* qfrac, int positive_only, void* stream);
int adalog_select_quantile_out(void* workspace, int64_t S, int nq, const float* weights, int mbs, float* out, void* stream);
int adalog_select_value_out(void* workspace, int64_t S, int R, float* out, void* stream);

/* ---- small vector kernels
 * adalog_shift_fold: out[c][o] = bias[o] - shift[0] * (w_scale[c][o] * rowsum[c][o])      reference linear.py:999-1006
 *   (rowsum from adalog_pack_uniform): folds the post-GELU "- shift" operand term into the bias, for reparam_bias and
 *   for every post-GELU scoring call (linear.py:837,879,920).  bias may be NULL.
 * adalog_minmax_rows: K4, per-row (min, max) [of |w| when use_abs] of w[rows][I]             reference linear.py:265-274
 * adalog_absminmax_cols: K4, (min |x|, max |x|) per tensor ([1]) or per column ([I])          reference linear.py:276-294 */
int adalog_shift_fold(const int32_t* rowsum, const float* w_scale, const float* shift, const float* bias, int C, int O,
                      float* out, void* stream);
int adalog_minmax_rows(const float* w, int rows, int I, int use_abs, float* mn, float* mx, void* stream);
int adalog_absminmax_cols(const float* x, int64_t rows, int I, int per_channel, float* mn, float* mx, void* stream);

/* ---- K17  BRECQ / AdaRound block reconstruction: fused straight-through backward passes and AdaRound kernels
 * adalog_uniform_fq_backward: gradients of the training form y = (clamp(round_ste(x/s) + round_ste(zp), 0, 2L-1) - round_ste(zp)) * s
 *   (reference quantizers/uniform.py:29-35 + _ste.py:5-6):  gx = gy*[inside];  gscale[ch] = sum gy*((q - z) - [inside]*x/s);
 *   gzp[ch] = sum gy*([inside] ? 0 : -s).  Broadcast layout as adalog_uniform_fake_quant_f32 with inner >= 1 rows
 *   (per-tensor / per-row / per-head).  gx, gscale, gzp are each optional.  workspace: 2*(n/inner)*blocks floats with
 *   blocks = adalog_uniform_fq_backward_blocks(n, n_channels, inner).
 * adalog_log_fq_backward: gradients of AdaLog's training form (reference quantizers/logarithm.py:88-92,133-135):
 *   gx = gy * y'/(u*s) inside both clamps, gscale = sum gy*(y'/s - dy/dx * (x+shift)/s), y' = y before the shift
 *   subtraction.  workspace: 1024 floats.
 * adalog_adaround: forward (backward = 0): out = (clamp(floor(w/s) + h + zp, 0, 2L-1) - zp) * s with
 *   h = clamp(sigmoid(alpha)*1.2 - 0.1, 0, 1) (soft) or [alpha >= 0] (hard)   (reference quantizers/adaround.py:43-60);
 *   backward = 1: out = d loss / d alpha = gy * s * h'(alpha) * [inside clamp].  w/alpha: [rows][inner], scale/zp: [rows].
 * adalog_round_loss: loss[0] = sum (1 - |2h(alpha)-1|^b)  and, if galpha, with g = gscale * (gmul ? gmul[0] : 1) * d/d alpha:
 *   galpha = g (overwrite != 0) or galpha += g   (reference utils/block_recon.py:205-210).  gmul (optional, device fp32
 *   [1]) is the upstream gradient of the scalar loss.  b_dev (optional, device fp32 [1]) overrides b: the exponent is then
 *   read on the device, so a captured HIP graph of a BRECQ iteration follows the decaying b.  workspace: 1024 floats.
 * The per-tensor parameter gradients and the loss value are reduced inside the producing kernel: every block leaves an
 *   fp32 partial, and the last block to arrive (device ticket counter) sums them in fp64 in a fixed order. */
/* y = clamp(rne(x/s) + rne(zp), 0, 2L-1) - rne(zp) as fp32 (per-tensor s, zp): the exact integer part of reference
 *   quantizers/uniform.py:29-35, i.e. the fake-quantised activation before its scale -- the integer operand of adalog_gemm_f32x3. */
int adalog_uniform_int_f32(const float* x, float* y, int64_t n, const float* scale, const float* zero_point, int n_bits,
                           void* stream);
/* q, k, v of an attention block split AND fake-quantised in one pass, and the straight-through gradients the other way (a BRECQ
 *   iteration; reference utils/wrap_net.py:21-22 followed by the input quantisers of the two attention products,
 *   quant_layers/matmul.py:43-47: asymmetric uniform, per tensor or per head).  src [B][N][3][H][D] -> y0, y1, y2 [B][H][N][D];
 *   part p uses scales[p] / zps[p] (H values when per_head[p], else one) and n_bits[p]; HOST arrays of three.  D = 32 or 64.
 *   backward: g0, g1, g2 (null = zeros) -> gx [B][N][3][H][D] (optional) and gscales[p] (optional; [H] or [1]);
 *   workspace: 3 * H * adalog_qkv_quant_chunks(B, N, D) floats.  Per element the operations of adalog_uniform_fake_quant_f32 /
 *   adalog_uniform_fq_backward. */
int adalog_qkv_quant_chunks(int64_t B, int64_t N, int D);
int adalog_qkv_split_quant(const float* src, float* y0, float* y1, float* y2, int64_t B, int64_t N, int H, int D,
                           const float* const* scales, const float* const* zps, const int* per_head, const int* n_bits, void* stream);
int adalog_qkv_merge_quant_backward(const float* g0, const float* g1, const float* g2, const float* src, float* gx, int64_t B,
                                    int64_t N, int H, int D, const float* const* scales, const float* const* zps, const int* per_head,
                                    const int* n_bits, float* const* gscales, float* workspace, void* stream);
/* y = softmax(x * scale) over rows of n <= 1024 values, and gx = scale * y * (gy - sum_j gy_j y_j): the `attn * scale` +
 *   softmax of an attention block (reference utils/wrap_net.py:26-27) and its autograd transposes as one pass each way inside a
 *   BRECQ iteration (ATen's order of operations: fl(x * scale), max, exp, fp32 sum, divide). */
int adalog_scaled_softmax(const float* x, float* y, int64_t rows, int n, float scale, void* stream);
int adalog_scaled_softmax_backward(const float* gy, const float* y, float* gx, int64_t rows, int n, float scale, void* stream);
/* Head split of an attention block in one pass: src [B][N][P][H][D] (the qkv Linear's output, P = 3) -> dst [P][B][H][N][D]
 *   (contiguous q, k, v); inverse != 0: the other way (the gradient).  Replaces the reshape / permute / unbind copies of the
 *   reference's attention forward (utils/wrap_net.py:19-33) and their autograd transposes in a BRECQ iteration.  D % 4 == 0. */
int adalog_permute_heads(const float* src, float* dst, int64_t B, int64_t N, int P, int H, int D, int inverse, void* stream);
/* The inverse with the P <= 4 parts as separate tensors [B][H][N][D] (the gradients of q, k, v as autograd delivers them);
 *   a null part counts as zeros.  dst [B][N][P][H][D]. */
int adalog_merge_heads(const float* p0, const float* p1, const float* p2, const float* p3, float* dst, int64_t B, int64_t N, int P,
                       int H, int D, void* stream);
int adalog_uniform_fq_backward_blocks(int64_t n, int64_t n_channels, int64_t inner);
int adalog_uniform_fq_backward(const float* gy, const float* x, float* gx, int64_t n, const float* scale,
                               const float* zero_point, int64_t n_channels, int64_t inner, int n_bits, int symmetric,
                               float* gscale, float* gzp, float* workspace, void* stream);
int adalog_log_fq_backward(const float* gy, const float* x, const float* y, float* gx, int64_t n, const float* scale,
                           const int64_t* q, int n_bits, const float* shift, int sub_shift, float* gscale, float* workspace,
                           void* stream);
int adalog_adaround(const float* w, const float* alpha, const float* gy, float* out, int64_t rows, int64_t inner,
                    const float* scale, const float* zero_point, int n_bits, int soft, int backward, void* stream);
/* adalog_adaround's forward writing both orientations: out [rows][inner] and out_t [inner][rows] (the K-major image of the
 *   soft-rounded weights that the forward product of a BRECQ iteration reads; adaround.py:38-57 otherwise unchanged). */
int adalog_adaround_t(const float* w, const float* alpha, float* out, float* out_t, int64_t rows, int64_t inner, const float* scale,
                      const float* zero_point, int n_bits, int soft, void* stream);
int adalog_round_loss(const float* alpha, int64_t n, float b, const float* b_dev, float* loss, float* galpha, float gscale,
                      const float* gmul, int overwrite, float* workspace, void* stream);
/* The regulariser of a whole block in one launch: loss[0] = weight * sum_t sum_i (1 - |2h(alpha_t[i])-1|^b) over `count`
 *   (<= 16) tensors and grads[t][i] = weight * d/d alpha_t[i] (the loop of reference utils/block_recon.py:205-210).
 *   alphas / grads / ns are HOST arrays (device pointers / element counts).  workspace:
 *   adalog_round_loss_multi_workspace(ns, count) floats. */
int64_t adalog_round_loss_multi_workspace(const int64_t* ns, int count);
int adalog_round_loss_multi(const float* const* alphas, float* const* grads, const int64_t* ns, int count, float b,
                            const float* b_dev, float weight, float* loss, float* workspace, const float* gate_dev, void* stream);
/* Reconstruction loss of a BRECQ iteration, LossFunction.lp_loss with p = 2 (reference utils/block_recon.py:186-199):
 *   loss[0] = scale * sum (pred - tgt)^2, the caller folding 1/(batch*channels) of `.sum(1).mean()` and the /10 into scale;
 *   backward: gpred = 2 * scale * gmul[0] * (pred - tgt).  workspace: 2048 floats. */
int adalog_rec_loss(const float* pred, const float* tgt, int64_t n, float scale, float* loss, float* workspace, void* stream);
int adalog_rec_loss_backward(const float* pred, const float* tgt, int64_t n, float scale, const float* gmul, float* gpred,
                             void* stream);
/* AdaRound's alpha of <= 16 layers, gradient AND Adam step in one launch (a captured BRECQ iteration on one GPU): per element
 *   g = [inside] gw * s * h'(alpha)  +  gmul[0] * weight * d/d alpha (1 - |2 h(alpha) - 1|^b)      (quantizers/adaround.py:38-57 under
 *   autograd; utils/block_recon.py:205-210), then torch.optim.Adam's default update (utils/block_recon.py:108-109,122-125) -- the
 *   operations of adalog_adaround (backward form), adalog_round_loss_multi's gradients and adalog_adam_multi, in their order.
 *   HOST arrays of device pointers; gws[t] = dL/dw_sim of layer t or null; ns[t] = rows * inners[t]; gmul null = no regulariser;
 *   gate (optional): a device 0 / 1 factor on gmul (the regulariser's warm-up switch).  adalog_round_loss_multi's gate_dev likewise
 *   multiplies its value. */
int adalog_alpha_step_multi(float* const* alphas, const float* const* ws, const float* const* gws, const float* const* scales,
                            const float* const* zps, float* const* exp_avg, float* const* exp_avg_sq, const int64_t* ns,
                            const int64_t* inners, const int* n_bits, int count, float lr, const float* lr_dev, float beta1, float beta2,
                            float eps, float* step_dev, float b, const float* b_dev, float weight, const float* gmul, const float* gate, void* stream);
/* One Adam step (torch.optim.Adam defaults; reference utils/block_recon.py:108-109,122-125) for `count` <= 16 fp32 tensors in ONE
 *   launch:  m = m + (1-b1)(g - m);  v = b2 v + (1-b2) g^2;  p -= lr / (1 - b1^t) * m / (sqrt(v) / sqrt(1 - b2^t) + eps), t =
 *   step_dev[0] + 1; step_dev[0] is advanced by one.  params / grads / exp_avg / exp_avg_sq / ns are HOST arrays (device
 *   pointers / element counts); lr_dev (optional device fp32 [1]) overrides lr (the cosine schedule writes it). */
int adalog_adam_multi(float* const* params, const float* const* grads, float* const* exp_avg, float* const* exp_avg_sq,
                      const int64_t* ns, int count, float lr, const float* lr_dev, float beta1, float beta2, float eps,
                      float* step_dev, void* stream);
/* Allocates the current device's ticket counters of the BRECQ kernels' in-kernel reductions (idempotent, one ring per
 *   device).  The allocation synchronises the device: call it before capturing BRECQ launches into a HIP graph
 *   (the training loop of reference utils/block_recon.py:114-127 is replayed from one). */
int adalog_brecq_init(void);


/* ---- K17b  BRECQ's training-mode contractions (csrc/brecq_gemm.hip).  Replaces the fp32 library GEMMs of one iteration of
 *   reference utils/block_recon.py:116-121 -- F.linear(x_sim, w_sim, bias) (quant_layers/linear.py:46-50) and A_sim @ B_sim
 *   (quant_layers/matmul.py:41-44), forward and both backward products each:
 *     C[g][m][n] = alpha * (alpha_dev ? alpha_dev[0] : 1) * sum_k opA(A[g])[m][k] * opB(B[g])[n][k]  + bias[n]
 *     opA(A)[m][k] = transA ? A[k*lda + m] : A[m*lda + k],   opB(B)[n][k] = transB ? B[k*ldb + n] : B[n*ldb + k]
 *   fp32 operands, fp32 result, fp32-class accuracy: each operand element is split in registers into three bf16 terms
 *   (exact: 24 significand bits) and the six bf16 MFMA products of weight >= 2^-24 are accumulated in fp32.
 *   Requirements: base pointers 16-byte aligned; with a bias N % 16 == 0; an operand matrix below 2 GiB.  Rows need not be
 *   16-byte aligned (the attention products have 197 tokens); N, ldc or sCg off a multiple of 4 take an element-store epilogue.  allow_split: the library may split K into
 *   fixed ranges (few-tile, long-K products such as dL/dw) and add the partial tiles in a fixed order; workspace then holds
 *   adalog_gemm_f32x3_workspace_bytes(...) bytes (0 = none needed).  exactA / exactB: the caller guarantees that the operand's
 *   values are exact in bf16 (integers |v| <= 256: the integer part q - z of a uniformly fake-quantised activation, whose
 *   trained scale is then handed over as alpha_dev): the forward (exactA, both operands K-contiguous) and dL/dw (exactB, both
 *   K-major) forms then take 3 products instead of 6; the result is the same either way. */
int64_t adalog_gemm_f32x3_workspace_bytes(int M, int N, int K, int G, int allow_split, int exactA, int exactB, int transA,
                                          int transB);
int adalog_gemm_f32x3(const float* A, int64_t lda, int transA, const float* B, int64_t ldb, int transB, float* C, int64_t ldc,
                      int M, int N, int K, int G, int64_t sAg, int64_t sBg, int64_t sCg, const float* bias, float alpha,
                      const float* alpha_dev, int allow_split, int exactA, int exactB, float* workspace, void* stream);
/* ... + addend[g][m][n] (laid out like C, may be null): added in the split product's reduction pass, or by a pass of its own when the
 * product is not split -- the residual stream (x + fc2(...)) inside a BRECQ iteration */
int adalog_gemm_f32x3_add(const float* A, int64_t lda, int transA, const float* B, int64_t ldb, int transB, float* C, int64_t ldc, int M,
                          int N, int K, int G, int64_t sAg, int64_t sBg, int64_t sCg, const float* bias, float alpha,
                          const float* alpha_dev, int allow_split, int exactA, int exactB, const float* addend, float* workspace,
                          void* stream);
/* The same product over TWO-LEVEL groups: group g = go * Gi + gi sits at gi * s?g + go * s?o in each operand (a [B][H] batch whose
 *   strides do not collapse into one: q / k / v read in place from the qkv output, or softmax.v writing [B][N][H][D] directly, so
 *   that the transpose(1, 2).reshape of reference utils/wrap_net.py:31 is a view).  Gi <= 0 or >= G: one level. */
int adalog_gemm_f32x3_g2(const float* A, int64_t lda, int transA, const float* B, int64_t ldb, int transB, float* C,
                         int64_t ldc, int M, int N, int K, int G, int64_t sAg, int64_t sBg, int64_t sCg, int Gi, int64_t sAo,
                         int64_t sBo, int64_t sCo, const float* bias, float alpha, const float* alpha_dev, int allow_split,
                         int exactA, int exactB, float* workspace, void* stream);
/* The same product with B ALREADY SPLIT into three bf16 planes by adalog_pack_split3_bf16 (row n of group g at
 *   Bp + (g*N + n)*3*Kt: hi | mid | lo of Kt >= K elements each, zero beyond K, Kt % 32 == 0) -- for an operand that every row
 *   tile re-reads (the soft-rounded weights w_sim / w_sim^T of reference quant_layers/linear.py:46-50): the split then runs
 *   once per iteration in the packer instead of once per row tile in the GEMM's registers.  A is K-contiguous. */
int64_t adalog_gemm_f32x3_planes_workspace_bytes(int M, int N, int K, int G, int allow_split, int exactA);
int adalog_gemm_f32x3_planes(const float* A, int64_t lda, const void* Bp, int64_t Kt, float* C, int64_t ldc, int M, int N, int K,
                             int G, int64_t sAg, int64_t sCg, const float* bias, float alpha, const float* alpha_dev,
                             int allow_split, int exactA, float* workspace, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* ADALOG_HIP_H */
