// K7/K8/K11-K15 -- candidate-scoring GEMM with a fused squared-error epilogue, on the CDNA4 matrix cores.
//
// Replaces, for every scoring call of the reference's searches
//   quant_layers/linear.py:355-384 (_search_best_w_scale), :394-423 (_search_best_a_scale),
//   :816-848/:856-890/:898-931 (post-GELU AdaLog searches), matmul.py:135-163/:173-201/:321-351, conv.py:226-255,
// the sequence  F.linear / @ / F.conv2d  ->  out_sim[.., P, ..] in HBM  ->  (raw_out - out_sim)**2  ->  mean/sum,
// by ONE kernel per call: D = A.B^T on MFMA from packed operands (operand.hip), then in registers
//   out = D * (sa * sb[col]) + bias[col];   e = ref - out;   column sums of e*e over the tile's rows,
// so only per-tile score partials ever reach HBM (the reference materialises out_sim: 3.7 GB for deit_small qkv).
// A second tiny kernel adds the partials in fp64 in a fixed order (deterministic, SURVEY "hard parts").
//
// Data types:  0 = int8  (v_mfma_i32_32x32x32_i8,  exact integer dot products)
//              3 = fp8   (v_mfma_scale_f32_32x32x64_f8f6f4, e4m3: operands q - z in [-15, 15] of <= 4-bit layers are exact,
//                         sums < 2^24 exact in the fp32 accumulator; same rate as int8, no cvt in the epilogue)
//              1 = bf16  (v_mfma_f32_32x32x16_bf16, AdaLog operand m*2^-t and integer operand exact in bf16)
//              2 = fp32  (v_mfma_f32_32x32x2_f32,   conv patch-embed with unquantised 8-bit input)
// Five kernels live here (DESIGN.md section 4 has the measurements that led from one to the next):
//   k_gemm_slab    -- int8 searches with K <= 384 bytes: 256 candidate columns resident in LDS, the fixed operand
//                     streamed wave-privately past them, column sums in registers, no barrier in the main loop;
//   k_gemm_grp     -- attention q.k^T searches (one K-step, many small groups): 7 consumer waves with register-resident
//                     row fragments + 1 LDS-DMA loader wave;
//   k_gemm_stream  -- every other search (candidates in the GEMM columns, reference rows contiguous): persistent
//                     workgroups, LDS-DMA ring streaming across tiles, packed-fp32 epilogue, per-workgroup fp64 sums;
//   k_gemm_cand    -- quant_forward (stores the product) and the launches k_gemm_stream does not take
//                     (k_gemm_cand_glds is its LDS-DMA variant): (64..256) x 256 tile, 128-byte K-steps;
//   k_gemm_score   -- candidates in a grid dimension (C > 1): 128 x 128 tile, 64-byte K-steps.
// All stage operands through LDS with a 16-byte-slot XOR swizzle (0 bank conflicts measured) and order workgroups so that
// tiles sharing an operand run on one XCD (its L2).
#include "common.h"
#include "fpcs_tail.h"
#include <type_traits>
#include <stdlib.h>

// The quantisation kernels are compiled without FMA contraction (bin indices must round like the reference); this file
// holds no bin-defining arithmetic, so the epilogues may fuse multiply-adds.
#pragma clang fp contract(fast)

namespace {

#include "gemm_types.inc"
#include "gemm_k_basic.inc"
#include "gemm_k_stream.inc"
#include "gemm_k_slab.inc"
#include "gemm_k_grp.inc"
#include "gemm_finish.inc"

}  // namespace

static int device_cus() {
    static int n_cu = 0;
    if (!n_cu) {
        int dev = 0; hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) n_cu = prop.multiProcessorCount;
        if (n_cu <= 0) n_cu = 256;
    }
    return n_cu;
}

// ---- tile selection shared by launch, layout query and finish
// quant_forward (store form): the products of a forward pass are small -- fc2 of deit_small is 25 x 2 tiles of 256 x 256 on 256 CUs.
// Largest row tile that still gives every CU its two workgroups; 64-row tiles otherwise.  Measured per product of a deit_small
// forward (32 images, us per launch at 256- / 128- / 64-row tiles): fc2 66 / 39 / 26, softmax.v 29 / 28 / 16, proj - / 22 / 13,
// fc1 - / 31 / 21, qkv - / 20 / 17, q.k^T 13 / 15 / 12 (profiles/r06_notes.md).
static int pick_tm_out(int M, int64_t col_tiles_x_groups) {
    if (const char* e = getenv("ADALOG_GEMM_TM")) { const int v = atoi(e); if (v == 1 || v == 2 || v == 4) return v; }
    const int64_t want = 2 * (int64_t)device_cus();
    const int tms[3] = {4, 2, 1};
    for (int i = 0; i < 3; ++i)
        if ((int64_t)cdiv(M, 64 * tms[i]) * col_tiles_x_groups >= want) return tms[i];
    return 1;
}

static int pick_tm(int M, bool scoring) {
    // largest row tile whose padding waste stays within 10 % of the best achievable.  The scoring epilogue keeps more
    // state than the store epilogue: with TM = 4 (128 accumulator VGPRs) it spills, so scoring launches use TM <= 2.
    if (const char* e = getenv("ADALOG_GEMM_TM")) {             // tuning knob for experiments (1, 2 or 4)
        const int v = atoi(e);
        if (v == 1 || v == 2 || (v == 4 && !scoring)) return v;
    }
    double best = 0.0;
    int tms[3] = {scoring ? 2 : 4, 2, 1};
    double util[3];
    for (int i = 0; i < 3; ++i) {
        const int bm = 64 * tms[i];
        util[i] = (double)M / ((double)cdiv(M, bm) * bm);
        if (util[i] > best) best = util[i];
    }
    for (int i = 0; i < 3; ++i)
        if (util[i] >= 0.9 * best) return tms[i];
    return 1;
}

struct Layout { int big, tm, wide, MT, NT, Npad, c_eff, n_eff, stream, acc, wgs, slab, slab_U, slab_R, slab_nb; int64_t elems; };

// Wide (one workgroup per CU, 192/256-row tile) form of the streaming kernel: from 8 K-steps on, where the L2 -> LDS path
// bounds the main loop (measured: K = 768 int8 -- vit_base / deit_base -- 0.39 -> 0.44 of peak, a calibration 4.5 % shorter;
// K = 512 -- swin stage 2 -- +4 %; below that the un-overlapped epilogue of the larger tile costs more than it saves).
static int pick_wide(int M, int64_t kvalid_bytes) {
    static const int use_wide = getenv("ADALOG_GEMM_WIDE") ? atoi(getenv("ADALOG_GEMM_WIDE")) : 1;
    static const int min_k = getenv("ADALOG_GEMM_WIDE_MINK") ? atoi(getenv("ADALOG_GEMM_WIDE_MINK")) : 512;
    if (!use_wide || kvalid_bytes < min_k || M < 192) return 0;
    const int64_t pad4 = (int64_t)cdiv(M, 256) * 256, pad3 = (int64_t)cdiv(M, 192) * 192, pad2 = (int64_t)cdiv(M, 128) * 128;
    const int ri = pad4 <= pad3 + pad3 / 32 ? 4 : 3;                      // 256 rows unless 192 pads > 3 % less
    const int64_t padw = ri == 4 ? pad4 : pad3;
    return padw <= pad2 + pad2 / 8 ? ri : 0;                              // not if it pads > 12 % more than 128-row tiles
}

// reduce_cols with ref_div > 1 asks for per-workgroup accumulation, which only the streaming kernel provides: when that
// kernel is not eligible the launch falls back to per-tile partials (candidate innermost), and the layout says so.
static Layout layout_of(int M, int N, int C, int G, int gmod, int ref_div, int reduce_cols, bool scoring = true,
                        int64_t kvalid_bytes = 0, int64_t kb = 0, bool ref_transposed = false, int dtype = -1, bool gen = false) {
    static const int use_stream = getenv("ADALOG_GEMM_STREAM") ? atoi(getenv("ADALOG_GEMM_STREAM")) : 1;
    static const int use_wgacc = getenv("ADALOG_GEMM_WGACC") ? atoi(getenv("ADALOG_GEMM_WGACC")) : 1;
    Layout L{};
    L.big = (C == 1);
    L.tm = L.big ? (scoring ? pick_tm(M, scoring) : pick_tm_out(M, (int64_t)cdiv(N, 256) * G)) : 2;
    const bool cand_cols = L.big && scoring && ref_transposed && (ref_div == 64 || ref_div == 128 || ref_div == 256);
    if (cand_cols && use_stream) {
        L.wide = pick_wide(M, kvalid_bytes);
        if (L.wide) L.tm = L.wide;
    }
    L.stream = cand_cols && use_stream && (L.tm <= 2 || L.wide) && (int64_t)(64 * L.tm + BN2) * kb < ((int64_t)1 << 31);
    if (!L.stream && L.wide) { L.wide = 0; L.tm = pick_tm(M, scoring); }
    // Slab kernel: int8 or fp8 storage, one group, 2..6 K-steps (the 256-column slab is <= 96 KiB), whole 32-row units, at least three
    // units per wave and slab, and a streamed operand small enough for the caches.
    static const int use_slab = getenv("ADALOG_GEMM_SLAB") ? atoi(getenv("ADALOG_GEMM_SLAB")) : 1;
    // (the GEN form -- no packed operand, no packer launch -- pays from one unit per wave and slab on: attn.proj's 384 rows)
    static const int slab_min_m_env = getenv("ADALOG_GEMM_SLAB_MINM") ? atoi(getenv("ADALOG_GEMM_SLAB_MINM")) : 768;
    const int slab_min_m = gen ? 256 : slab_min_m_env;
    // streamed (fixed) operand: every workgroup walks all of it past its resident slab, from L2 or the Infinity Cache (16 MiB:
    // swin stage 0's 100 352 tokens x 128 B -- those launches ran on the streaming kernel at 0.26 of peak, 0.43 here)
    static const int64_t slab_max_bytes = getenv("ADALOG_GEMM_SLAB_MAXB") ? atoll(getenv("ADALOG_GEMM_SLAB_MAXB")) : ((int64_t)16 << 20);
    // 256-column slabs up to 6 K-steps; 128-column slabs up to 12 (K = 512 / 768: swin stage 2, vit_base) when a slab still holds
    // whole reference columns (64 or 128 candidates)
    static const int use_slab128 = getenv("ADALOG_GEMM_SLAB128") ? atoi(getenv("ADALOG_GEMM_SLAB128")) : 1;
    const bool ok128 = use_slab128 && kb <= 12 * BK3 && (ref_div == 64 || ref_div == 128);
    const int slab_nb = kb <= 6 * BK3 ? 8 : (ok128 ? 4 : 0);       // (128-column slabs at K = 384 were measured: +57 ms per calibration)
    if (L.stream && (g_slab_override >= 0 ? g_slab_override : use_slab) && (dtype == 0 || dtype == 3) && G == 1 && slab_nb != 0 && kvalid_bytes > BK3 && M % 32 == 0 && M >= slab_min_m &&
        (int64_t)M * kb <= slab_max_bytes && (int64_t)cdiv(N, 32 * slab_nb) * (M / 32) < ((int64_t)1 << 30)) {
        const int SBN = 32 * slab_nb;
        L.slab_nb = slab_nb;
        L.slab = 1;
        L.slab_U = M / 32;
        L.NT = cdiv(N, SBN);
        const int64_t units = (int64_t)L.NT * L.slab_U;
        L.slab_R = (int)(cdiv(cdiv(units, (int64_t)device_cus()), (int64_t)8) * 8);
        L.wgs = (int)cdiv(units, (int64_t)L.slab_R);
        L.MT = cdiv(L.slab_U, L.slab_R) + 1;                 // pieces a slab can be cut into by the range boundaries
        L.n_eff = N / ref_div;
        L.c_eff = ref_div;
        L.acc = reduce_cols && use_wgacc;
        L.Npad = cdiv(L.n_eff, 64) * 64;
        L.elems = L.acc ? (int64_t)2 * L.wgs * gmod * BN2 : (int64_t)L.c_eff * G * L.MT * L.Npad;
        L.wide = 0; L.tm = 2;
        return L;
    }
    const int bm = L.big ? 64 * L.tm : BM, bn = L.big ? BN2 : BN;
    L.MT = cdiv(M, bm);
    L.NT = cdiv(N, bn);
    L.n_eff = ref_div > 1 ? N / ref_div : N;
    L.c_eff = ref_div > 1 ? ref_div : C;
    const int64_t tiles = (int64_t)L.MT * L.NT * G * C;
    const int64_t want = (int64_t)(L.wide ? 1 : 2) * device_cus();
    L.wgs = (int)(tiles < want ? tiles : want);
    L.acc = L.stream && reduce_cols && ref_div > 1 && use_wgacc;
    const int red = reduce_cols && ref_div == 1;
    L.Npad = red ? L.NT : (ref_div > 1 ? cdiv(L.n_eff, 64) * 64 : L.NT * bn);
    L.elems = L.acc ? (int64_t)2 * L.wgs * gmod * BN2 : (int64_t)L.c_eff * G * L.MT * L.Npad;
    return L;
}

// Launch-time choice of the group kernel (it shares the streaming kernel's accumulator layout, so the layout query does
// not need to know): int8 or fp8, one K-step, 5..7 row blocks, many groups (up to 16 heads per image: the per-head fp64
// sums live in LDS), plain column factors.
static bool grp_ok(int dtype, int M, int N, int G, int gmod, int ref_div, int64_t kvalid_bytes, const float* bias,
                   const float* row_scale, int64_t sb_n, int64_t ref_cs) {
    static const int use_grp = getenv("ADALOG_GEMM_GRP") ? atoi(getenv("ADALOG_GEMM_GRP")) : 1;
    return use_grp && (dtype == 0 || dtype == 3) && kvalid_bytes <= BK3 && M > 128 && M <= 224 && G >= 8 && gmod <= 16 && !bias && !row_scale &&
           sb_n == 0 && (ref_div == 64 || ref_div == 128 || ref_div == 256) && N % ref_div == 0 &&
           (int64_t)(N / ref_div) * ref_cs * 4 < ((int64_t)1 << 31);
}

static bool grpw_on() {
    static const int v = getenv("ADALOG_GEMM_GRPW") ? atoi(getenv("ADALOG_GEMM_GRPW")) : 1;     // wave-private form of the q.k^T kernel
    return v != 0;
}

// ... and of its several-K-steps form: bf16, exactly 7 K-steps (K = 193..224 elements: the 197 tokens of a 224 x 224 ViT).
static bool grpk_ok(int dtype, int M, int N, int G, int gmod, int ref_div, int64_t kvalid_bytes, const float* bias,
                    const float* row_scale, int64_t sb_n, int64_t ref_cs) {
    static const int use_grp = getenv("ADALOG_GEMM_GRP") ? atoi(getenv("ADALOG_GEMM_GRP")) : 1;
    return use_grp && dtype == 1 && kvalid_bytes > 6 * BK3 && kvalid_bytes <= 7 * BK3 && M > 128 && M <= 224 && G >= 8 && gmod <= 16 &&
           !bias && !row_scale && sb_n == 0 && (ref_div == 64 || ref_div == 128 || ref_div == 256) && N % ref_div == 0 &&
           (int64_t)(N / ref_div) * ref_cs * 4 < ((int64_t)1 << 31);
}

// ... and of its mixed form (dtype 4: bf16 rows x fp8 columns): exactly 4 K-steps of 64 elements (K = 193..256).
static bool grpk8_ok(int M, int N, int G, int gmod, int ref_div, int64_t k_valid, const float* bias, const float* row_scale,
                     int64_t sb_n, int64_t ref_cs) {
    static const int use_grp = getenv("ADALOG_GEMM_GRP") ? atoi(getenv("ADALOG_GEMM_GRP")) : 1;
    return use_grp && k_valid > 192 && k_valid <= 256 && M > 128 && M <= 224 && G >= 8 && gmod <= 16 && !bias && !row_scale &&
           sb_n == 0 && (ref_div == 64 || ref_div == 128 || ref_div == 256) && N % ref_div == 0 &&
           (int64_t)(N / ref_div) * ref_cs * 4 < ((int64_t)1 << 31);
}

// ... mixed operands, window family: K <= 64 elements (one 64-byte fp8 K-step against 128-byte bf16 rows), at most 64 rows.
static bool winb_ok(int M, int N, int G, int gmod, int ref_div, int64_t k_valid, const float* bias, const float* row_scale, int64_t sb_n,
                    int64_t ref_cs, int wgs) {
    static const int use_win = getenv("ADALOG_GEMM_WIN") ? atoi(getenv("ADALOG_GEMM_WIN")) : 1;
    const int n_eff = ref_div > 0 ? N / ref_div : 0;
    return use_win && k_valid >= 1 && k_valid <= 64 && M >= 4 && M <= 64 && G >= 256 && gmod <= 32 && !bias && !row_scale && sb_n == 0 &&
           (ref_div == 64 || ref_div == 128 || ref_div == 256) && N % ref_div == 0 && n_eff <= 64 && wgs * 4 >= gmod && ref_cs >= M;
}

// ... and of the window kernel: int8 or fp8, one K-step, at most 64 rows, hundreds of groups or more, at least as many waves as
// heads per image (every participating wave owns one head).
static bool win_ok(int dtype, int M, int N, int G, int gmod, int ref_div, int64_t kvalid_bytes, const float* bias,
                   const float* row_scale, int64_t sb_n, int64_t ref_cs, int wgs) {
    static const int use_win = getenv("ADALOG_GEMM_WIN") ? atoi(getenv("ADALOG_GEMM_WIN")) : 1;
    const int n_eff = ref_div > 0 ? N / ref_div : 0;
    return use_win && (dtype == 0 || dtype == 3) && kvalid_bytes <= BK3 && M >= 4 && M <= 64 && G >= 256 && gmod <= 32 && !bias &&
           !row_scale && sb_n == 0 && (ref_div == 64 || ref_div == 128 || ref_div == 256) && N % ref_div == 0 && n_eff <= 64 &&
           wgs * 4 >= gmod && ref_cs >= M;
}

// M, N: GEMM rows / columns (N includes the candidate factor when ref_div > 1).  Outputs the partial-buffer layout
// [c_eff][G][MT][Npad] the kernel will write, for allocation and for adalog_finish_scores.
extern "C" int64_t adalog_gemm_score_layout(int M, int N, int C, int G, int gmod, int ref_div, int reduce_cols, int dtype,
                                            int64_t Kp, int64_t k_valid, int ref_transposed, int* MT, int* Npad, int* mode) {
    if (dtype == 4) dtype = 1;                          // mixed operands (bf16 rows x fp8 columns): laid out like the bf16 launch
    const int esz = (dtype == 0 || dtype == 3) ? 1 : dtype == 1 ? 2 : 4;
    const Layout L = layout_of(M, N, C, G, gmod, ref_div, reduce_cols, true, (k_valid > 0 ? k_valid : Kp) * esz, Kp * esz,
                               ref_transposed != 0, dtype);
    if (MT) *MT = L.acc ? L.wgs : L.MT;
    if (Npad) *Npad = L.acc ? BN2 : L.Npad;
    if (mode) *mode = L.acc ? 2 : (ref_div > 1 ? 1 : 0);
    return L.elems;
}

// 1 when a scoring launch of this shape runs on the window kernel, i.e. when int8 / fp8 operands of K <= 32 may be packed
// with 32-byte rows (half the operand bytes of the 64-byte K-step padding).  C = 1, reduce_cols = 1, transposed reference.
extern "C" int adalog_gemm_win_ok(int dtype, int M, int N, int G, int gmod, int ref_div, int64_t k_valid) {
    const int esz = 1;
    if (!(dtype == 0 || dtype == 3) || k_valid > 32 || ref_div < 1 || N % ref_div != 0) return 0;
    const Layout L = layout_of(M, N, 1, G, gmod, ref_div, 1, true, k_valid * esz, 32 * esz, true, dtype);
    return (L.stream && L.acc && win_ok(dtype, M, N, G, gmod, ref_div, k_valid * esz, nullptr, nullptr, 0, M, L.wgs)) ? 1 : 0;
}

// 1 when adalog_gemm_score takes dtype 4 (A: bf16 rows, B: fp8 e4m3 candidate columns, both [..][Kp] with Kp = 256 elements) for
// this shape: the softmax.v weight search of a 197-token ViT (M = 197 attention rows, K = 197 keys) with <= 4-bit candidates.
// C = 1, reduce_cols = 1, transposed reference.
extern "C" int adalog_gemm_mixed_ok(int M, int N, int G, int gmod, int ref_div, int64_t k_valid) {
    if (ref_div < 1 || N % ref_div != 0 || k_valid < 1) return 0;
    if (k_valid > 256) {
        // third family: the wide streaming kernel (one group or many), rows of any multiple of 64 elements: taken when the all-bf16
        // launch of this shape would run on the wide form (K >= 256 elements, M >= 192)
        static const int use_mx = getenv("ADALOG_GEMM_STREAM_MX") ? atoi(getenv("ADALOG_GEMM_STREAM_MX")) : 1;
        const int64_t Kp = (k_valid + 63) / 64 * 64;
        const Layout L = layout_of(M, N, 1, G, gmod, ref_div, 0, true, k_valid * 2, Kp * 2, true, 1);
        return (use_mx && L.stream && L.wide && !L.slab) ? 1 : 0;
    }
    const int64_t Kp = k_valid <= 64 ? 64 : 256;                        // windows / 197-token groups
    const Layout L = layout_of(M, N, 1, G, gmod, ref_div, 1, true, k_valid * 2, Kp * 2, true, 1);
    if (!(L.stream && L.acc)) return 0;
    if (k_valid <= 64) return winb_ok(M, N, G, gmod, ref_div, k_valid, nullptr, nullptr, 0, M, L.wgs) ? 1 : 0;
    return grpk8_ok(M, N, G, gmod, ref_div, k_valid, nullptr, nullptr, 0, M) ? 1 : 0;
}

// GEN form of the attention searches (adalog_gemm_score_gen): the candidate operand B is not read but generated in the kernel
// from the fp32 tensor x [G][N / ref_div][K] with the candidates' (sb, zp) pairs
struct MmGen { const float* x; int64_t ldx, sg; const float* zp; int n_bits; };
// extras of the STORE epilogue (GemmArgs: addend, out_gi, sOo)
struct MmOutEx { const float* addend; int out_gi; int64_t sOo; };

static int gemm_score_impl(int dtype, const void* A, const void* B, int64_t sAc, int64_t sAg, int64_t sBc,
                           int64_t sBg, int M, int N, int64_t Kp, int64_t k_valid, int C, int G, int gmod, const float* ref,
                           int64_t ldr, int64_t sRg, int64_t ref_cs, int ref_div, const float* sa, int64_t sa_c,
                           int64_t sa_g, float sa_mul, const float* sb, int64_t sb_c, int64_t sb_g, int64_t sb_n,
                           const float* bias, int64_t bi_c, int64_t bi_g, int64_t bi_n, const float* row_scale,
                           const float* row_bias, float* partial, int64_t partial_elems, float* out, int64_t ldo,
                           int64_t sOc, int64_t sOg, int order, int reduce_cols, void* stream, const MmGen* gen,
                           const MmOutEx* ox = nullptr) {
    ADALOG_ARG_CHECK(A && (B || gen) && sa && sb, "gemm_score: null operand/scale pointer");
    ADALOG_ARG_CHECK(!ox || (out && C == 1 && !row_scale && (ox->out_gi == 0 || (ox->out_gi > 0 && G % ox->out_gi == 0))),
                     "gemm_out_ex: the epilogue extras need out, C == 1 and an inner group count that divides G");
    ADALOG_ARG_CHECK(!gen || ((dtype == 0 || dtype == 3) && gen->x && gen->zp && k_valid > 0 && k_valid % 16 == 0 && k_valid <= 64 &&
                              gen->ldx % 4 == 0 && gen->sg % 4 == 0 && (((uintptr_t)gen->x) & 15) == 0),
                     "gemm_score_gen: int8 / fp8 candidates of K = 16, 32, 48 or 64 from a 16-byte aligned fp32 tensor");
    ADALOG_ARG_CHECK(dtype >= 0 && dtype <= 4, "gemm_score: dtype must be 0 (i8), 1 (bf16), 2 (f32), 3 (fp8 e4m3) or 4 (bf16 rows x fp8 columns)");
    ADALOG_ARG_CHECK(M >= 1 && N >= 1 && C >= 1 && G >= 1 && gmod >= 1 && G % gmod == 0 && ref_div >= 1, "gemm_score: bad sizes");
    if (dtype == 4 && k_valid > 256) {
        // mixed operands, streaming family: the wide persistent kernel with two 64-byte planes of bf16 rows per fp8 K-step
        ADALOG_ARG_CHECK(Kp % 64 == 0 && k_valid <= Kp && C == 1 && partial && ref && !out && ldr == 1 && !row_scale &&
                         adalog_gemm_mixed_ok(M, N, G, gmod, ref_div, k_valid),
                         "gemm_score: bf16 x fp8 operands, streaming family: Kp a multiple of 64, C = 1, transposed reference, a shape adalog_gemm_mixed_ok accepts");
        ADALOG_ARG_CHECK(order >= 0 && order <= 2, "gemm_score: order must be 0, 1 or 2");
        const Layout L = layout_of(M, N, C, G, gmod, ref_div, reduce_cols, true, k_valid * 2, Kp * 2, true, 1);
        ADALOG_ARG_CHECK(L.stream && L.wide && !L.slab, "gemm_score: bf16 x fp8 operands: not a wide streaming shape");
        GemmArgs p{};
        p.A = (const uint8_t*)A; p.B = (const uint8_t*)B;
        p.sAc = sAc * 2; p.sAg = sAg * 2; p.sBc = sBc; p.sBg = sBg;
        p.M = M; p.N = N; p.Kb = Kp; p.KbA = Kp * 2; p.Kvb = k_valid; p.C = C; p.G = G; p.gmod = gmod;
        p.ref = ref; p.ldr = ldr; p.sRg = sRg; p.ref_cs = ref_cs; p.ref_div = ref_div;
        ADALOG_ARG_CHECK(((int64_t)(M - 1) * ldr + (int64_t)(L.n_eff - 1) * (ref_cs > 0 ? ref_cs : 1) < ((int64_t)1 << 31)),
                         "gemm_score: reference group exceeds 32-bit addressing");
        p.sa = sa; p.sa_c = sa_c; p.sa_g = sa_g; p.sa_mul = sa_mul;
        p.sb = sb; p.sb_c = sb_c; p.sb_g = sb_g; p.sb_n = sb_n;
        p.bias = bias; p.bi_c = bi_c; p.bi_g = bi_g; p.bi_n = bi_n;
        p.row_bias = row_bias;
        p.MT = L.MT; p.NT = L.NT; p.Npad = L.Npad;
        p.order = order; p.reduce_cols = 0; p.partial = partial;
        ADALOG_ARG_CHECK(partial_elems >= L.elems, "gemm_score: partial buffer too small");
        if (L.acc) { ADALOG_ARG_CHECK(((uintptr_t)partial & 7) == 0, "gemm_score: accumulator buffer must be 8-byte aligned"); p.wg_acc = (double*)partial; }
        {
            static const int64_t grp_bytes = getenv("ADALOG_GEMM_GM_BYTES") ? atoll(getenv("ADALOG_GEMM_GM_BYTES")) : ((int64_t)8 << 20);
            int64_t gm = grp_bytes / ((int64_t)64 * L.tm * p.KbA);
            if (const char* e = getenv("ADALOG_GEMM_GM")) gm = atoi(e);
            p.gm = (int)(gm < 1 ? 1 : gm > L.MT ? L.MT : gm);
        }
        hipStream_t st = (hipStream_t)stream;
        const size_t shm = (size_t)3 * (2 * 64 * L.tm + BN2) * BK3;
#define LAUNCH_STREAM_MX(RIV)                                                                                     \
        do {                                                                                                      \
            static unsigned long long attr_dev = 0; \
            { hipError_t ea__ = adalog_max_lds(reinterpret_cast<const void*>(&k_gemm_stream<1, RIV, 8, 3, true>), (int)(150 * 1024), &attr_dev); \
              if (ea__ != hipSuccess) { adalog_set_error("hipFuncSetAttribute", ea__); return (int)ea__; } } \
            adalog_note_kernel("k_gemm_stream<bf16xfp8>");                                                        \
            hipLaunchKernelGGL((k_gemm_stream<1, RIV, 8, 3, true>), dim3((unsigned)L.wgs), dim3(512), shm, st, p); \
        } while (0)
        if (L.wide == 4) LAUNCH_STREAM_MX(4); else LAUNCH_STREAM_MX(3);
#undef LAUNCH_STREAM_MX
        ADALOG_LAUNCH_CHECK("adalog_gemm_score (bf16 x fp8, streaming)");
        return 0;
    }
    if (dtype == 4) {
        // mixed operands: one kernel, one shape family (adalog_gemm_mixed_ok)
        const bool window = k_valid > 0 && k_valid <= 64;
        ADALOG_ARG_CHECK(Kp == (window ? 64 : 256) && k_valid > 0 && C == 1 && partial && ref && !out && ldr == 1 && reduce_cols == 1 &&
                         adalog_gemm_mixed_ok(M, N, G, gmod, ref_div, k_valid),
                         "gemm_score: bf16 x fp8 operands are taken for the shapes adalog_gemm_mixed_ok accepts only (Kp = 64 or 256, C = 1, transposed reference)");
        const Layout L = layout_of(M, N, C, G, gmod, ref_div, reduce_cols, true, k_valid * 2, Kp * 2, true, 1);
        ADALOG_ARG_CHECK(window ? winb_ok(M, N, G, gmod, ref_div, k_valid, bias, row_scale, sb_n, ref_cs, L.wgs)
                                : grpk8_ok(M, N, G, gmod, ref_div, k_valid, bias, row_scale, sb_n, ref_cs),
                         "gemm_score: bf16 x fp8 operands: epilogue options not supported for this shape");
        GemmArgs p{};
        p.A = (const uint8_t*)A; p.B = (const uint8_t*)B;
        p.sAc = sAc * 2; p.sAg = sAg * 2; p.sBc = sBc; p.sBg = sBg;
        p.M = M; p.N = N; p.Kb = Kp; p.KbA = Kp * 2; p.Kvb = k_valid; p.C = C; p.G = G; p.gmod = gmod;
        p.ref = ref; p.ldr = ldr; p.sRg = sRg; p.ref_cs = ref_cs; p.ref_div = ref_div;
        p.sa = sa; p.sa_c = sa_c; p.sa_g = sa_g; p.sa_mul = sa_mul;
        p.sb = sb; p.sb_c = sb_c; p.sb_g = sb_g; p.sb_n = sb_n;
        p.MT = L.MT; p.NT = L.NT; p.Npad = L.Npad; p.order = order; p.reduce_cols = 0; p.partial = partial;
        ADALOG_ARG_CHECK(partial_elems >= L.elems && ((uintptr_t)partial & 7) == 0, "gemm_score: accumulator buffer too small or misaligned");
        p.wg_acc = (double*)partial;
        hipStream_t st = (hipStream_t)stream;
        if (window) {
            const int n_eff = N / ref_div;
            const size_t ref_lds = (size_t)4 * n_eff * 64 * 4, acc_lds = (size_t)gmod * 256 * 8;
            const size_t shm_w = ref_lds > acc_lds ? ref_lds : acc_lds;
#define LAUNCH_WINB(NJV)                                                                                          \
            do {                                                                                                  \
                static unsigned long long attr_dev = 0; \
                { hipError_t ea__ = adalog_max_lds(reinterpret_cast<const void*>(&k_gemm_winb<NJV>), (int)(80 * 1024), &attr_dev); \
                  if (ea__ != hipSuccess) { adalog_set_error("hipFuncSetAttribute", ea__); return (int)ea__; } } \
                adalog_note_kernel("k_gemm_winb<bf16xfp8>");                                                      \
                hipLaunchKernelGGL((k_gemm_winb<NJV>), dim3((unsigned)L.wgs), dim3(256), shm_w, st, p);           \
            } while (0)
            if (ref_div == 64) LAUNCH_WINB(2); else if (ref_div == 128) LAUNCH_WINB(4); else LAUNCH_WINB(8);
#undef LAUNCH_WINB
            ADALOG_LAUNCH_CHECK("adalog_gemm_score (bf16 x fp8, windows)");
            return 0;
        }
        const int NB = N / 32;
        const int nch0 = cdiv((int64_t)3 * L.wgs, G);
        const int CB = cdiv(cdiv(NB, nch0 < 1 ? 1 : nch0), 8) * 8;
        p.slab_R = CB; p.slab_U = cdiv(NB, CB);
        const size_t shm = (size_t)3 * 4 * 4 * 32 * BK3 + (size_t)7 * ref_div * 4 + (size_t)gmod * 256 * 8;   // 3 stages of 4 K-steps x 4 blocks
#define LAUNCH_GRPK8(NJV)                                                                                         \
        do {                                                                                                      \
            static unsigned long long attr_dev = 0; \
            { hipError_t ea__ = adalog_max_lds(reinterpret_cast<const void*>(&k_gemm_grpk8<NJV, 4>), (int)(160 * 1024), &attr_dev); \
              if (ea__ != hipSuccess) { adalog_set_error("hipFuncSetAttribute", ea__); return (int)ea__; } } \
            adalog_note_kernel("k_gemm_grpk8<bf16xfp8>");                                                         \
            hipLaunchKernelGGL((k_gemm_grpk8<NJV, 4>), dim3((unsigned)L.wgs), dim3(512), shm, st, p);             \
        } while (0)
        if (ref_div == 64) LAUNCH_GRPK8(2); else if (ref_div == 128) LAUNCH_GRPK8(4); else LAUNCH_GRPK8(8);
#undef LAUNCH_GRPK8
        ADALOG_LAUNCH_CHECK("adalog_gemm_score (bf16 x fp8)");
        return 0;
    }
    const int esz = (dtype == 0 || dtype == 3) ? 1 : dtype == 1 ? 2 : 4;
    ADALOG_ARG_CHECK(Kp > 0 && ((Kp * esz) % BK3 == 0 || (Kp * esz == 32 && (dtype == 0 || dtype == 3))),
                     "gemm_score: padded K must be a multiple of 64 bytes (32-byte rows: int8 / fp8, window kernel only)");
    ADALOG_ARG_CHECK((partial != nullptr) == (ref != nullptr), "gemm_score: partial and ref go together");
    ADALOG_ARG_CHECK(partial || out, "gemm_score: nothing to produce");
    ADALOG_ARG_CHECK(order >= 0 && order <= 2, "gemm_score: order must be 0, 1 or 2");
    ADALOG_ARG_CHECK(ref_div == 1 || (C == 1 && N % ref_div == 0 && !out), "gemm_score: ref_div > 1 needs C == 1, N % ref_div == 0, no out");
    ADALOG_ARG_CHECK(!row_scale || (C == 1 && row_bias), "gemm_score: per-row scale needs C == 1 and a row_bias vector");
    ADALOG_ARG_CHECK(!(partial && out), "gemm_score: either score against ref or store out, not both");
    const Layout L = layout_of(M, N, C, G, gmod, ref_div, reduce_cols, out == nullptr, (k_valid > 0 ? k_valid : Kp) * esz, Kp * esz,
                               ldr == 1 && ref != nullptr, dtype);
    GemmArgs p{};
    p.A = (const uint8_t*)A; p.B = (const uint8_t*)B;
    p.sAc = sAc * esz; p.sAg = sAg * esz; p.sBc = sBc * esz; p.sBg = sBg * esz;
    ADALOG_ARG_CHECK(k_valid >= 0 && k_valid <= Kp, "gemm_score: k_valid must be in [0, Kp]");
    p.M = M; p.N = N; p.Kb = Kp * esz; p.Kvb = (k_valid > 0 ? k_valid : Kp) * esz; p.C = C; p.G = G; p.gmod = gmod;
    p.ref = ref; p.ldr = ldr; p.sRg = sRg; p.ref_cs = ref_cs; p.ref_div = ref_div;
    ADALOG_ARG_CHECK(!ref || ((int64_t)(M - 1) * ldr + (int64_t)(L.n_eff - 1) * (ref_cs > 0 ? ref_cs : 1) < ((int64_t)1 << 31)),
                     "gemm_score: reference group exceeds 32-bit addressing");
    p.sa = sa; p.sa_c = sa_c; p.sa_g = sa_g; p.sa_mul = sa_mul;
    p.sb = sb; p.sb_c = sb_c; p.sb_g = sb_g; p.sb_n = sb_n;
    p.bias = bias; p.bi_c = bi_c; p.bi_g = bi_g; p.bi_n = bi_n;
    p.row_scale = row_scale; p.row_bias = row_bias;
    p.MT = L.MT; p.NT = L.NT; p.Npad = L.Npad;
    p.order = order; p.reduce_cols = reduce_cols && ref_div == 1; p.timeline = g_timeline;
    p.partial = partial; p.out = out; p.ldo = ldo; p.sOc = sOc; p.sOg = sOg;
    if (ox) { p.addend = ox->addend; p.out_gi = ox->out_gi; p.sOo = ox->sOo; }
    if (partial) ADALOG_ARG_CHECK(partial_elems >= L.elems, "gemm_score: partial buffer too small");
    if (L.acc) { ADALOG_ARG_CHECK(((uintptr_t)partial & 7) == 0, "gemm_score: accumulator buffer must be 8-byte aligned"); p.wg_acc = (double*)partial; }
    if (gen) {
        const bool win = L.stream && !out && L.acc && !L.slab && win_ok(dtype, M, N, G, gmod, ref_div, p.Kvb, bias, row_scale, sb_n, ref_cs, L.wgs);
        const bool grpw = L.stream && !out && L.acc && !L.slab && grp_ok(dtype, M, N, G, gmod, ref_div, p.Kvb, bias, row_scale, sb_n, ref_cs) &&
                          grpw_on() && L.wgs * 4 >= gmod && ref_cs >= M;
        ADALOG_ARG_CHECK(win || grpw, "gemm_score_gen: not a shape of the window / wave-private group kernels (adalog_gemm_score_gen_ok)");
        p.gen_x = gen->x; p.gen_ldx = gen->ldx; p.gen_sg = gen->sg; p.gen_K = (int)k_valid; p.gen_zp = gen->zp;
        p.gen_qmax = (float)((1 << gen->n_bits) - 1);
        const float tie = 6e-7f * (float)(1 << gen->n_bits);
        p.gen_tie = 0.5f - (tie > 1e-5f ? tie : 1e-5f);
    }
    const int64_t nwg = (int64_t)L.MT * L.NT * G * C;
    ADALOG_ARG_CHECK(nwg < (int64_t)1 << 31, "gemm_score: grid too large");
    hipStream_t st = (hipStream_t)stream;
    dim3 grid((unsigned)nwg);
    static const int use_glds = getenv("ADALOG_GEMM_GLDS") ? atoi(getenv("ADALOG_GEMM_GLDS")) : 1;   // LDS-DMA pipeline (default on)
    ADALOG_ARG_CHECK((Kp * esz) % BK2 == 0 || (L.stream && !out),
                     "gemm_score: rows padded to 64 (not 128) bytes are taken by the streaming search kernel only");
    ADALOG_ARG_CHECK((Kp * esz) % BK3 == 0 || (L.stream && !out && L.acc && win_ok(dtype, M, N, G, gmod, ref_div, p.Kvb, bias, row_scale, sb_n, ref_cs, L.wgs)),
                     "gemm_score: 32-byte rows are taken by the window kernel only (adalog_gemm_win_ok)");
    ADALOG_ARG_CHECK(dtype != 3 || (L.stream && !out), "gemm_score: fp8 operands are taken by the streaming search kernel only (ref_div 64/128/256, transposed reference)");
    if (L.slab && !out) {
        // slab kernel: one workgroup per CU, each takes a contiguous range of (slab, unit) pairs
        p.MT = L.MT; p.NT = L.NT; p.slab_U = L.slab_U; p.slab_R = L.slab_R;
        const int nk = (int)((p.Kvb + BK3 - 1) / BK3);
        const int SBN = 32 * L.slab_nb;
        const size_t shm = (size_t)nk * SBN * BK3 + 8 * 3 * 32 * BK3 + 8 * 192 * 4 + 8 * SBN * 4;
        // a slab that is not cut has unused pieces: they must read as zero
        if (!L.acc) {
            const hipError_t me = hipMemsetAsync(partial, 0, (size_t)L.elems * sizeof(float), st);
            if (me != hipSuccess) { adalog_set_error("adalog_gemm_score (clear partials)", me); return (int)me; }
        }
        const int nref = SBN / ref_div;
#define LAUNCH_SLAB(NREFV, ROWSV, DTV, NBV)                                                                       \
        do {                                                                                                      \
            static unsigned long long attr_dev = 0; \
            { hipError_t ea__ = adalog_max_lds(reinterpret_cast<const void*>(&k_gemm_slab<NREFV, ROWSV, DTV, NBV>), (int)(160 * 1024), &attr_dev); \
              if (ea__ != hipSuccess) { adalog_set_error("hipFuncSetAttribute", ea__); return (int)ea__; } } \
            adalog_note_kernel(DTV == 3 ? (NBV == 8 ? "k_gemm_slab<fp8>" : "k_gemm_slab128<fp8>") : (NBV == 8 ? "k_gemm_slab<i8>" : "k_gemm_slab128<i8>")); \
            hipLaunchKernelGGL((k_gemm_slab<NREFV, ROWSV, DTV, NBV>), dim3((unsigned)L.wgs), dim3(512), shm, st, p); \
        } while (0)
#define LAUNCH_SLAB_DT(DTV)                                                                                       \
        do {                                                                                                      \
            if (L.slab_nb == 8) {                                                                                 \
                if (row_scale) { if (nref == 1) LAUNCH_SLAB(1, true, DTV, 8); else if (nref == 2) LAUNCH_SLAB(2, true, DTV, 8); else LAUNCH_SLAB(4, true, DTV, 8); } \
                else { if (nref == 1) LAUNCH_SLAB(1, false, DTV, 8); else if (nref == 2) LAUNCH_SLAB(2, false, DTV, 8); else LAUNCH_SLAB(4, false, DTV, 8); } \
            } else {                                                                                              \
                if (row_scale) { if (nref == 1) LAUNCH_SLAB(1, true, DTV, 4); else LAUNCH_SLAB(2, true, DTV, 4); } \
                else { if (nref == 1) LAUNCH_SLAB(1, false, DTV, 4); else LAUNCH_SLAB(2, false, DTV, 4); }        \
            }                                                                                                     \
        } while (0)
        if (dtype == 3) LAUNCH_SLAB_DT(3); else LAUNCH_SLAB_DT(0);
#undef LAUNCH_SLAB_DT
#undef LAUNCH_SLAB
    } else if (L.stream && !out && L.acc && win_ok(dtype, M, N, G, gmod, ref_div, p.Kvb, bias, row_scale, sb_n, ref_cs, L.wgs)) {
        // window kernel (swin attention searches): a wave per group, same accumulator layout and workgroup count
        const int n_eff = N / ref_div;
        const size_t ref_lds = (size_t)4 * n_eff * 64 * 4, acc_lds = (size_t)gmod * 256 * 8;
        const size_t shm = ref_lds > acc_lds ? ref_lds : acc_lds;
#define LAUNCH_WIN(NJV, DTV)                                                                                      \
        do {                                                                                                      \
            static unsigned long long attr_dev = 0; \
            { hipError_t ea__ = adalog_max_lds(reinterpret_cast<const void*>(&k_gemm_win<NJV, DTV>), (int)(80 * 1024), &attr_dev); \
              if (ea__ != hipSuccess) { adalog_set_error("hipFuncSetAttribute", ea__); return (int)ea__; } } \
            if (gen) {                                                                                            \
                static unsigned long long attr_gen = 0; \
                { hipError_t ea__ = adalog_max_lds(reinterpret_cast<const void*>(&k_gemm_win<NJV, DTV, true>), (int)(80 * 1024), &attr_gen); \
                  if (ea__ != hipSuccess) { adalog_set_error("hipFuncSetAttribute", ea__); return (int)ea__; } } \
                adalog_note_kernel(DTV == 3 ? "k_gemm_win_gen<fp8>" : "k_gemm_win_gen<i8>");                       \
                hipLaunchKernelGGL((k_gemm_win<NJV, DTV, true>), dim3((unsigned)L.wgs), dim3(256), shm, st, p);   \
            } else {                                                                                              \
                adalog_note_kernel(DTV == 3 ? "k_gemm_win<fp8>" : "k_gemm_win<i8>");                               \
                hipLaunchKernelGGL((k_gemm_win<NJV, DTV>), dim3((unsigned)L.wgs), dim3(256), shm, st, p);         \
            }                                                                                                     \
        } while (0)
        if (dtype == 3) { if (ref_div == 64) LAUNCH_WIN(2, 3); else if (ref_div == 128) LAUNCH_WIN(4, 3); else LAUNCH_WIN(8, 3); }
        else { if (ref_div == 64) LAUNCH_WIN(2, 0); else if (ref_div == 128) LAUNCH_WIN(4, 0); else LAUNCH_WIN(8, 0); }
#undef LAUNCH_WIN
    } else if (L.stream && !out && L.acc && grp_ok(dtype, M, N, G, gmod, ref_div, p.Kvb, bias, row_scale, sb_n, ref_cs) &&
               grpw_on() && L.wgs * 4 >= gmod && ref_cs >= M) {
        // wave-private group kernel (q.k^T searches): a wave per (group, chunk of reference columns), no barrier in the loop
        const int n_eff = N / ref_div;
        const int64_t waves = (int64_t)L.wgs * 4;
        int nch = (int)cdiv(3 * waves, G);
        nch = nch < 1 ? 1 : nch > n_eff ? n_eff : nch;
        const int cbc = cdiv(n_eff, nch);
        p.slab_R = cbc; p.slab_U = cdiv(n_eff, cbc);
        const size_t ref_lds = (size_t)4 * 2 * 224 * 4, acc_lds = (size_t)gmod * 256 * 8;
        const size_t shm = ref_lds > acc_lds ? ref_lds : acc_lds;
#define LAUNCH_GRPW(NJV, DTV)                                                                                     \
        do {                                                                                                      \
            if (gen) {                                                                                            \
                adalog_note_kernel(DTV == 3 ? "k_gemm_grpw_gen<fp8>" : "k_gemm_grpw_gen<i8>");                     \
                hipLaunchKernelGGL((k_gemm_grpw<NJV, DTV, true>), dim3((unsigned)L.wgs), dim3(256), shm, st, p);  \
            } else {                                                                                              \
                adalog_note_kernel(DTV == 3 ? "k_gemm_grpw<fp8>" : "k_gemm_grpw<i8>");                             \
                hipLaunchKernelGGL((k_gemm_grpw<NJV, DTV>), dim3((unsigned)L.wgs), dim3(256), shm, st, p);        \
            }                                                                                                     \
        } while (0)
        if (dtype == 3) { if (ref_div == 64) LAUNCH_GRPW(2, 3); else if (ref_div == 128) LAUNCH_GRPW(4, 3); else LAUNCH_GRPW(8, 3); }
        else { if (ref_div == 64) LAUNCH_GRPW(2, 0); else if (ref_div == 128) LAUNCH_GRPW(4, 0); else LAUNCH_GRPW(8, 0); }
#undef LAUNCH_GRPW
    } else if (L.stream && !out && L.acc && grp_ok(dtype, M, N, G, gmod, ref_div, p.Kvb, bias, row_scale, sb_n, ref_cs)) {
        // group kernel (q.k^T searches): same accumulator layout and workgroup count as the streaming kernel
        const int NB = N / 32;
        const int nch0 = cdiv((int64_t)3 * L.wgs, G);
        const int CB = cdiv(cdiv(NB, nch0 < 1 ? 1 : nch0), 8) * 8;
        p.slab_R = CB; p.slab_U = cdiv(NB, CB);
        const size_t shm = (size_t)3 * 8 * 32 * BK3 + 7 * 256 * 4 + (size_t)gmod * 256 * 8;
#define LAUNCH_GRP(NJV, DTV)                                                                                      \
        do {                                                                                                      \
            static unsigned long long attr_dev = 0; \
            { hipError_t ea__ = adalog_max_lds(reinterpret_cast<const void*>(&k_gemm_grp<NJV, DTV>), (int)(96 * 1024), &attr_dev); \
              if (ea__ != hipSuccess) { adalog_set_error("hipFuncSetAttribute", ea__); return (int)ea__; } } \
            adalog_note_kernel(DTV == 3 ? "k_gemm_grp<fp8>" : "k_gemm_grp<i8>");                                   \
            hipLaunchKernelGGL((k_gemm_grp<NJV, DTV>), dim3((unsigned)L.wgs), dim3(512), shm, st, p);             \
        } while (0)
        if (dtype == 3) { if (ref_div == 64) LAUNCH_GRP(2, 3); else if (ref_div == 128) LAUNCH_GRP(4, 3); else LAUNCH_GRP(8, 3); }
        else { if (ref_div == 64) LAUNCH_GRP(2, 0); else if (ref_div == 128) LAUNCH_GRP(4, 0); else LAUNCH_GRP(8, 0); }
#undef LAUNCH_GRP
    } else if (L.stream && !out && L.acc && grpk_ok(dtype, M, N, G, gmod, ref_div, p.Kvb, bias, row_scale, sb_n, ref_cs)) {
        // group kernel, 7 K-steps (softmax.v weight search)
        const int NB = N / 32;
        const int nch0 = cdiv((int64_t)3 * L.wgs, G);
        const int CB = cdiv(cdiv(NB, nch0 < 1 ? 1 : nch0), 8) * 8;
        p.slab_R = CB; p.slab_U = cdiv(NB, CB);
        const size_t shm = (size_t)3 * 7 * 2 * 32 * BK3 + (size_t)7 * ref_div * 4 + (size_t)gmod * 256 * 8;
#define LAUNCH_GRPK(NJV)                                                                                          \
        do {                                                                                                      \
            static unsigned long long attr_dev = 0; \
            { hipError_t ea__ = adalog_max_lds(reinterpret_cast<const void*>(&k_gemm_grpk<NJV, 7>), (int)(160 * 1024), &attr_dev); \
              if (ea__ != hipSuccess) { adalog_set_error("hipFuncSetAttribute", ea__); return (int)ea__; } } \
            adalog_note_kernel("k_gemm_grpk<bf16>");                                                              \
            hipLaunchKernelGGL((k_gemm_grpk<NJV, 7>), dim3((unsigned)L.wgs), dim3(512), shm, st, p);              \
        } while (0)
        if (ref_div == 64) LAUNCH_GRPK(2); else if (ref_div == 128) LAUNCH_GRPK(4); else LAUNCH_GRPK(8);
#undef LAUNCH_GRPK
    } else if (L.stream && !out) {
        // persistent streaming kernel: two (wide form: one) workgroups per CU walk the tile list
        {   // m-tiles per group: the group's A rows are re-read once per n-tile (from L2 / the 256 MiB Infinity Cache), the B
            // tiles stream from HBM once per GROUP -- with the candidate operand at 150..600 MB per launch that stream is
            // what must not repeat: groups of <= 8 MiB of A rows (2 MiB, half of an XCD's L2, re-read vit_base's fc2
            // candidates 25 times: 4.16 -> 4.05 s per calibration, 3.99 at 64 MiB; deit_small -- its operand fits the
            // Infinity Cache -- is indifferent up to 16 MiB and 2 % slower at 64)
            static const int64_t grp_bytes = getenv("ADALOG_GEMM_GM_BYTES") ? atoll(getenv("ADALOG_GEMM_GM_BYTES")) : ((int64_t)8 << 20);
            const int64_t a_tile = (int64_t)64 * L.tm * p.Kb;
            int64_t gm = grp_bytes / a_tile;
            if (const char* e = getenv("ADALOG_GEMM_GM")) gm = atoi(e);
            p.gm = (int)(gm < 1 ? 1 : gm > L.MT ? L.MT : gm);
        }
        dim3 pgrid((unsigned)L.wgs);
        const size_t shm = (size_t)(L.wide ? 4 : 3) * (64 * L.tm + BN2) * BK3;
#define LAUNCH_STREAM(DT, RIV, NWV, NSV)                                                                          \
        do {                                                                                                      \
            static unsigned long long attr_dev = 0; \
            { hipError_t ea__ = adalog_max_lds(reinterpret_cast<const void*>(&k_gemm_stream<DT, RIV, NWV, NSV>), (int)((NSV == 4 ? 128 : 80) * 1024), &attr_dev); \
              if (ea__ != hipSuccess) { adalog_set_error("hipFuncSetAttribute", ea__); return (int)ea__; } } \
            adalog_note_kernel(DT == 0 ? "k_gemm_stream<i8>" : DT == 1 ? "k_gemm_stream<bf16>" : DT == 2 ? "k_gemm_stream<f32>" : "k_gemm_stream<fp8>"); \
            hipLaunchKernelGGL((k_gemm_stream<DT, RIV, NWV, NSV>), pgrid, dim3(64 * NWV), shm, st, p);            \
        } while (0)
#define LAUNCH_STREAM_DT(DT)                                                                                      \
        do {                                                                                                      \
            if (L.wide == 4) LAUNCH_STREAM(DT, 4, 8, 4); else if (L.wide == 3) LAUNCH_STREAM(DT, 3, 8, 4);        \
            else if (L.tm == 2) LAUNCH_STREAM(DT, 2, 4, 3); else LAUNCH_STREAM(DT, 1, 4, 3);                      \
        } while (0)
        if (dtype == 0) LAUNCH_STREAM_DT(0); else if (dtype == 1) LAUNCH_STREAM_DT(1); else if (dtype == 2) LAUNCH_STREAM_DT(2);
        else LAUNCH_STREAM_DT(3);
#undef LAUNCH_STREAM_DT
#undef LAUNCH_STREAM
    } else if (L.big && use_glds && !out && L.tm <= 2) {
        const size_t shm = (size_t)3 * (64 * L.tm + BN2) * BK2 + (512 + 256) * sizeof(float);
#define LAUNCH_GLDS(DT, TMV)                                                                                      \
        do {                                                                                                      \
            static unsigned long long attr_dev = 0; \
            { hipError_t ea__ = adalog_max_lds(reinterpret_cast<const void*>(&k_gemm_cand_glds<DT, TMV>), (int)(160 * 1024), &attr_dev); \
              if (ea__ != hipSuccess) { adalog_set_error("hipFuncSetAttribute", ea__); return (int)ea__; } } \
            adalog_note_kernel("k_gemm_cand_glds");                                                               \
            hipLaunchKernelGGL((k_gemm_cand_glds<DT, TMV>), grid, dim3(512), shm, st, p);                         \
        } while (0)
        if (dtype == 0) { if (L.tm == 2) LAUNCH_GLDS(0, 2); else LAUNCH_GLDS(0, 1); }
        else if (dtype == 1) { if (L.tm == 2) LAUNCH_GLDS(1, 2); else LAUNCH_GLDS(1, 1); }
        else { if (L.tm == 2) LAUNCH_GLDS(2, 2); else LAUNCH_GLDS(2, 1); }
#undef LAUNCH_GLDS
    } else if (L.big) {
        const size_t shm = (size_t)(64 * L.tm + BN2) * BK2 + (512 + 256) * sizeof(float);
#define LAUNCH_BIG(DT, TMV, ST)                                                                                   \
        do {                                                                                                      \
            static unsigned long long attr_dev = 0; \
            { hipError_t ea__ = adalog_max_lds(reinterpret_cast<const void*>(&k_gemm_cand<DT, TMV, ST>), (int)(72 * 1024), &attr_dev); \
              if (ea__ != hipSuccess) { adalog_set_error("hipFuncSetAttribute", ea__); return (int)ea__; } } \
            adalog_note_kernel("k_gemm_cand");                                                                    \
            hipLaunchKernelGGL((k_gemm_cand<DT, TMV, ST>), grid, dim3(512), shm, st, p);                          \
        } while (0)
#define LAUNCH_BIG_TM(DT, ST)                                                                                     \
        do {                                                                                                      \
            if (L.tm == 4) LAUNCH_BIG(DT, 4, ST); else if (L.tm == 2) LAUNCH_BIG(DT, 2, ST); else LAUNCH_BIG(DT, 1, ST); \
        } while (0)
#define LAUNCH_BIG_DT(ST)                                                                                         \
        do {                                                                                                      \
            if (dtype == 0) LAUNCH_BIG_TM(0, ST); else if (dtype == 1) LAUNCH_BIG_TM(1, ST); else LAUNCH_BIG_TM(2, ST); \
        } while (0)
        if (out && ox) {
            // the STORE form with the epilogue extras (int8 / bf16 operands)
#define LAUNCH_ADD(DT, TMV)                                                                                       \
        do {                                                                                                      \
            static unsigned long long attr_dev = 0;                                                               \
            { hipError_t ea__ = adalog_max_lds(reinterpret_cast<const void*>(&k_gemm_cand<DT, TMV, true, false, true>), (int)(72 * 1024), &attr_dev); \
              if (ea__ != hipSuccess) { adalog_set_error("hipFuncSetAttribute", ea__); return (int)ea__; } }     \
            adalog_note_kernel("k_gemm_cand_ex");                                                                 \
            hipLaunchKernelGGL((k_gemm_cand<DT, TMV, true, false, true>), grid, dim3(512), shm, st, p);           \
        } while (0)
            ADALOG_ARG_CHECK(dtype == 0 || dtype == 1, "gemm_out_ex: int8 or bf16 operands");
            if (dtype == 0) { if (L.tm == 4) LAUNCH_ADD(0, 4); else if (L.tm == 2) LAUNCH_ADD(0, 2); else LAUNCH_ADD(0, 1); }
            else { if (L.tm == 4) LAUNCH_ADD(1, 4); else if (L.tm == 2) LAUNCH_ADD(1, 2); else LAUNCH_ADD(1, 1); }
#undef LAUNCH_ADD
        } else if (out) LAUNCH_BIG_DT(true); else LAUNCH_BIG_DT(false);
#undef LAUNCH_BIG_DT
#undef LAUNCH_BIG_TM
#undef LAUNCH_BIG
    } else {
        ADALOG_ARG_CHECK(!row_scale, "gemm_score: per-row scale is only available with C == 1");
        dim3 block(256);
#define LAUNCH(DT)                                                                                   \
        do {                                                                                         \
            adalog_note_kernel("k_gemm_score");                                                      \
            if (out) hipLaunchKernelGGL((k_gemm_score<DT, true>), grid, block, 0, st, p);            \
            else hipLaunchKernelGGL((k_gemm_score<DT, false>), grid, block, 0, st, p);               \
        } while (0)
        if (dtype == 0) LAUNCH(0); else if (dtype == 1) LAUNCH(1); else LAUNCH(2);
#undef LAUNCH
    }
    ADALOG_LAUNCH_CHECK("adalog_gemm_score");
    return 0;
}

extern "C" int adalog_gemm_score(int dtype, const void* A, const void* B, int64_t sAc, int64_t sAg, int64_t sBc,
                                 int64_t sBg, int M, int N, int64_t Kp, int64_t k_valid, int C, int G, int gmod, const float* ref,
                                 int64_t ldr, int64_t sRg, int64_t ref_cs, int ref_div, const float* sa, int64_t sa_c,
                                 int64_t sa_g, float sa_mul, const float* sb, int64_t sb_c, int64_t sb_g, int64_t sb_n,
                                 const float* bias, int64_t bi_c, int64_t bi_g, int64_t bi_n, const float* row_scale,
                                 const float* row_bias, float* partial, int64_t partial_elems, float* out, int64_t ldo,
                                 int64_t sOc, int64_t sOg, int order, int reduce_cols, void* stream) {
    return gemm_score_impl(dtype, A, B, sAc, sAg, sBc, sBg, M, N, Kp, k_valid, C, G, gmod, ref, ldr, sRg, ref_cs, ref_div, sa, sa_c, sa_g,
                           sa_mul, sb, sb_c, sb_g, sb_n, bias, bi_c, bi_g, bi_n, row_scale, row_bias, partial, partial_elems, out, ldo,
                           sOc, sOg, order, reduce_cols, stream, nullptr);
}

// quant_forward product from packed operands with the epilogue extras (reference quant_layers/linear.py:46-51, matmul.py:43-45,
// utils/wrap_net.py:30-31):  out[g][m][n] = sa[gh * sa_g] * sa_mul * sb[gh * sb_g + n * sb_n] * (A[g] . B[g]^T)[m][n]
//                                            + bias[gh * bi_g + n * bi_n] + addend[g][m][n]
// A [G][M][Kp], B [G][N][Kp] packed int8 (dtype 0) / bf16 (1) operands (group strides in elements; 0 = shared), gh = g % gmod.
// addend (may be null): the residual stream, indexed like out.  out_gi > 0: two-level output groups -- group g is written at
// (g % out_gi) * sOg + (g / out_gi) * sOo, e.g. softmax . v with out_gi = H, sOg = D, sOo = N * H * D, ldo = H * D writes
// [B][N][H][D] storage, so that the transpose(1, 2).reshape(B, N, C) in front of the projection layer is a view.
extern "C" int adalog_gemm_out_ex(int dtype, const void* A, const void* B, int64_t sAg, int64_t sBg, int M, int N, int64_t Kp, int G,
                                  int gmod, const float* sa, int64_t sa_g, float sa_mul, const float* sb, int64_t sb_g, int64_t sb_n,
                                  const float* bias, int64_t bi_g, int64_t bi_n, const float* addend, float* out, int64_t ldo,
                                  int64_t sOg, int out_gi, int64_t sOo, void* stream) {
    ADALOG_ARG_CHECK(out && (dtype == 0 || dtype == 1), "gemm_out_ex: int8 or bf16 operands, out required");
    const MmOutEx ox{addend, out_gi, sOo};
    return gemm_score_impl(dtype, A, B, 0, sAg, 0, sBg, M, N, Kp, 0, 1, G, gmod, nullptr, 0, 0, 1, 1, sa, 0, sa_g, sa_mul, sb, 0, sb_g, sb_n,
                           bias, 0, bi_g, bi_n, nullptr, nullptr, nullptr, 0, out, ldo, 0, sOg, 0, 0, stream, nullptr, &ox);
}

// quant_forward of a uniformly quantised Linear layer / q.k^T product (reference quant_layers/linear.py:46-51, matmul.py:43-45) with
// the A-side fake quantisation INSIDE the GEMM's loader (k_gemm_cand<.., GENA>):
//   out[g][m][n] = sa[gh * sa_g] * sa_mul * sb[gh * sb_g + n * sb_n] * sum_k (q_a(x[g][m][k]) - z_a) * B[g][n][k] + bias[gh * bi_g + n * bi_n]
// x fp32 [G][M][ldx] (groups sxg apart, K valid, K % 16 == 0), (a_scale, a_zp)[gh * a_pg] its per-tensor (a_pg = 0) / per-head
// quantiser (gh = g % gmod), B the packed int8 operand [G][N][Kp] (adalog_pack_uniform).  Same result, bit for bit, as
// adalog_pack_uniform(x) + adalog_gemm_score(out): the activation is read once as fp32 instead of written and re-read as int8, and
// one launch per layer disappears.
extern "C" int adalog_gemm_out_gen_ex(const float* x, int64_t ldx, int64_t sxg, int K, const float* a_scale, const float* a_zp, int64_t a_pg,
                                      int n_bits, const void* B, int64_t sBg, int M, int N, int64_t Kp, int G, int gmod, const float* sa,
                                      int64_t sa_g, float sa_mul, const float* sb, int64_t sb_g, int64_t sb_n, const float* bias,
                                      int64_t bi_g, int64_t bi_n, const float* addend, float* out, int64_t ldo, int64_t sOg, void* stream);
extern "C" int adalog_gemm_out_gen(const float* x, int64_t ldx, int64_t sxg, int K, const float* a_scale, const float* a_zp, int64_t a_pg,
                                   int n_bits, const void* B, int64_t sBg, int M, int N, int64_t Kp, int G, int gmod, const float* sa,
                                   int64_t sa_g, float sa_mul, const float* sb, int64_t sb_g, int64_t sb_n, const float* bias,
                                   int64_t bi_g, int64_t bi_n, float* out, int64_t ldo, int64_t sOg, void* stream) {
    return adalog_gemm_out_gen_ex(x, ldx, sxg, K, a_scale, a_zp, a_pg, n_bits, B, sBg, M, N, Kp, G, gmod, sa, sa_g, sa_mul, sb, sb_g, sb_n,
                                  bias, bi_g, bi_n, nullptr, out, ldo, sOg, stream);
}
// ... + addend[g][m][n] (same strides as out): the residual stream added in the epilogue (x + proj(...) of a transformer block)
extern "C" int adalog_gemm_out_gen_ex(const float* x, int64_t ldx, int64_t sxg, int K, const float* a_scale, const float* a_zp, int64_t a_pg,
                                      int n_bits, const void* B, int64_t sBg, int M, int N, int64_t Kp, int G, int gmod, const float* sa,
                                      int64_t sa_g, float sa_mul, const float* sb, int64_t sb_g, int64_t sb_n, const float* bias,
                                      int64_t bi_g, int64_t bi_n, const float* addend, float* out, int64_t ldo, int64_t sOg, void* stream) {
    ADALOG_ARG_CHECK(x && a_scale && a_zp && B && sa && sb && out, "gemm_out_gen: null pointer");
    ADALOG_ARG_CHECK(M >= 1 && N >= 1 && G >= 1 && gmod >= 1 && G % gmod == 0 && K >= 16 && K % 16 == 0 && Kp >= K && Kp % BK2 == 0,
                     "gemm_out_gen: K must be a multiple of 16, Kp a multiple of 128 covering it");
    ADALOG_ARG_CHECK(n_bits >= 2 && n_bits <= 7 && ldx >= K && ldx % 4 == 0 && sxg % 4 == 0 && (((uintptr_t)x) & 15) == 0 &&
                     (((uintptr_t)B) & 15) == 0, "gemm_out_gen: <= 7-bit quantiser, 16-byte aligned fp32 rows");
    ADALOG_ARG_CHECK((int64_t)M * ldx < ((int64_t)1 << 31) && (int64_t)N * Kp < ((int64_t)1 << 31), "gemm_out_gen: operand exceeds 32-bit addressing");
    GemmArgs p{};
    p.A = nullptr; p.B = (const uint8_t*)B; p.sAc = 0; p.sAg = 0; p.sBc = 0; p.sBg = sBg;
    p.M = M; p.N = N; p.Kb = Kp; p.KbA = Kp; p.Kvb = K; p.C = 1; p.G = G; p.gmod = gmod;
    p.ref = nullptr; p.ldr = 0; p.sRg = 0; p.ref_cs = 1; p.ref_div = 1;
    p.sa = sa; p.sa_c = 0; p.sa_g = sa_g; p.sa_mul = sa_mul;
    p.sb = sb; p.sb_c = 0; p.sb_g = sb_g; p.sb_n = sb_n;
    p.bias = bias; p.bi_c = 0; p.bi_g = bi_g; p.bi_n = bi_n;
    p.out = out; p.ldo = ldo; p.sOc = 0; p.sOg = sOg; p.addend = addend;
    p.order = 0; p.reduce_cols = 0;
    p.gen_x = x; p.gen_ldx = ldx; p.gen_sg = sxg; p.gen_K = K; p.gen_scale = a_scale; p.gen_zp = a_zp; p.gen_sn = a_pg;
    p.gen_qmax = (float)((1 << n_bits) - 1);
    // 128-row tiles while they give every CU one, else 64-row tiles (attn.proj of deit_small: 50 x 2 -> 99 x 2 tiles)
    int tmv = pick_tm_out(M, (int64_t)cdiv(N, BN2) * G);
    if (tmv > 2) tmv = 2;
    p.MT = cdiv(M, 64 * tmv); p.NT = cdiv(N, BN2); p.Npad = p.NT * BN2;
    const int64_t tiles = (int64_t)p.MT * p.NT * G;
    ADALOG_ARG_CHECK(tiles < ((int64_t)1 << 31), "gemm_out_gen: grid too large");
    const size_t shm = (size_t)(64 * tmv + BN2) * BK2 + (512 + 256) * sizeof(float);
#define LAUNCH_GENA(TMV, ADDV, LABEL)                                                                             \
    do {                                                                                                          \
        static unsigned long long attr_dev = 0;                                                                   \
        { hipError_t ea__ = adalog_max_lds(reinterpret_cast<const void*>(&k_gemm_cand<0, TMV, true, true, ADDV>), (int)(72 * 1024), &attr_dev); \
          if (ea__ != hipSuccess) { adalog_set_error("hipFuncSetAttribute", ea__); return (int)ea__; } }         \
        adalog_note_kernel(LABEL);                                                                                \
        hipLaunchKernelGGL((k_gemm_cand<0, TMV, true, true, ADDV>), dim3((unsigned)tiles), dim3(512), shm, (hipStream_t)stream, p); \
    } while (0)
    if (addend) { if (tmv == 2) LAUNCH_GENA(2, true, "k_gemm_cand_gen_ex"); else LAUNCH_GENA(1, true, "k_gemm_cand_gen_ex"); }
    else { if (tmv == 2) LAUNCH_GENA(2, false, "k_gemm_cand_gen"); else LAUNCH_GENA(1, false, "k_gemm_cand_gen"); }
#undef LAUNCH_GENA
    ADALOG_LAUNCH_CHECK("adalog_gemm_out_gen");
    return 0;
}

// Attention searches with uniform candidates (reference quant_layers/matmul.py:135-163 / 173-201), GEN form: scores of the
// ref_div candidates (sb, zp)[c * sb_c + head * sb_g] of the operand x [G][N / ref_div][K = k_valid] (fp32, rows ldx apart, groups
// sg apart) against the packed fixed operand A [G][M][Kp] -- what adalog_gemm_score computes from the packed candidate operand
// [G][N][Kp] (candidates innermost), without that operand: the kernels quantise x in registers (k_gemm_win / k_gemm_grpw, GEN).
// Transposed reference (ref_cs = M), per-workgroup fp64 accumulators: the partial-buffer layout of adalog_gemm_score_layout.
// Shapes: adalog_gemm_score_gen_ok.
extern "C" int adalog_gemm_score_gen(int dtype, const void* A, int64_t sAg, int M, int N, int64_t Kp, int64_t k_valid, int G, int gmod,
                                     const float* x, int64_t ldx, int64_t sg, const float* zp, int n_bits, const float* ref,
                                     int64_t sRg, int ref_div, const float* sa, int64_t sa_c, int64_t sa_g, float sa_mul,
                                     const float* sb, int64_t sb_c, int64_t sb_g, float* partial, int64_t partial_elems,
                                     void* stream) {
    ADALOG_ARG_CHECK(n_bits >= 1 && n_bits <= 8 && (dtype != 3 || n_bits <= 4), "gemm_score_gen: fp8 candidates hold <= 4-bit values");
    const MmGen gen{x, ldx, sg, zp, n_bits};
    return gemm_score_impl(dtype, A, nullptr, 0, sAg, 0, 0, M, N, Kp, k_valid, 1, G, gmod, ref, 1, sRg, M, ref_div, sa, sa_c, sa_g, sa_mul,
                           sb, sb_c, sb_g, 0, nullptr, 0, 0, 0, nullptr, nullptr, partial, partial_elems, nullptr, 0, 0, 0, 2, 1, stream,
                           &gen);
}

// softmax.v, log-base search of the post-softmax AdaLog quantiser (reference quant_layers/matmul.py:321-351): scores of the P = 128
// candidate bases q against the reference, with the candidate operand (P AdaLog quantisations of the probabilities x [G][N][K]:
// scale 1, no clamp) generated inside the kernel (k_gemm_avq) instead of packed by adalog_pack_adalog_bf16 and streamed.
//   A: the fixed operand v^T, bf16 [G][M <= 64][Kp] (K-contiguous, zero past K);  q: the bases [P];
//   lut: dword table [2^n_bits + 1][P], entry [k][c] = bf16 bits of the value of bin k under base q[c] (numerator * 2^-t, what
//        the packer writes), row 2^n_bits = 0 (the masked code);  ref [G][N][M];  sa / sb / sa_mul / partial as adalog_gemm_score
//        (C = 1, ref_div = P, reduce_cols = 1: the per-workgroup fp64 accumulators of adalog_gemm_score_layout with dtype 1).
static bool avq_ok(int M, int N, int G, int gmod, int P, int64_t k_valid, int64_t Kp, int n_bits) {
    static const int use_avq = getenv("ADALOG_GEMM_AVQ") ? atoi(getenv("ADALOG_GEMM_AVQ")) : 1;
    if (!use_avq || P != 128 || M < 1 || M > 64 || N < 1 || k_valid < 1 || k_valid > 208 || n_bits < 1 || n_bits > 6 || gmod < 1 || gmod > 16 ||
        G % gmod || G < 8)
        return false;
    const int nks = k_valid <= 64 ? 4 : 13;
    if (Kp < nks * 16 || (Kp * 2) % 16) return false;
    const Layout L = layout_of(M, N * P, 1, G, gmod, P, 1, true, k_valid * 2, Kp * 2, true, 1);
    return L.stream && L.acc && !L.slab && L.wgs * 4 >= gmod;
}
extern "C" int adalog_gemm_score_avq_ok(int M, int N, int G, int gmod, int P, int64_t k_valid, int64_t Kp, int n_bits) {
    return avq_ok(M, N, G, gmod, P, k_valid, Kp, n_bits) ? 1 : 0;
}
extern "C" int adalog_gemm_score_avq(const void* A, int64_t sAg, int M, int N, int64_t Kp, int64_t k_valid, int G, int gmod,
                                     const float* x, int64_t ldx, int64_t sg, const float* q, const uint32_t* lut, int n_bits,
                                     const float* ref, int64_t sRg, int P, const float* sa, int64_t sa_c, int64_t sa_g, float sa_mul,
                                     const float* sb, int64_t sb_c, int64_t sb_g, float* partial, int64_t partial_elems,
                                     void* stream) {
    ADALOG_ARG_CHECK(A && x && q && lut && ref && sa && sb && partial, "gemm_score_avq: null pointer");
    ADALOG_ARG_CHECK(avq_ok(M, N, G, gmod, P, k_valid, Kp, n_bits), "gemm_score_avq: shape not taken (adalog_gemm_score_avq_ok)");
    ADALOG_ARG_CHECK((((uintptr_t)A) & 15) == 0 && (((uintptr_t)partial) & 7) == 0, "gemm_score_avq: operand / accumulator alignment");
    const Layout L = layout_of(M, N * P, 1, G, gmod, P, 1, true, k_valid * 2, Kp * 2, true, 1);
    ADALOG_ARG_CHECK(partial_elems >= L.elems, "gemm_score_avq: partial buffer too small");
    GemmArgs p{};
    p.A = (const uint8_t*)A; p.sAg = sAg * 2; p.M = M; p.N = N * P; p.Kb = Kp * 2; p.Kvb = k_valid * 2; p.C = 1; p.G = G; p.gmod = gmod;
    p.ref = ref; p.ldr = 1; p.sRg = sRg; p.ref_cs = M; p.ref_div = P;
    p.sa = sa; p.sa_c = sa_c; p.sa_g = sa_g; p.sa_mul = sa_mul; p.sb = sb; p.sb_c = sb_c; p.sb_g = sb_g;
    p.wg_acc = (double*)partial; p.partial = partial;
    p.gen_x = x; p.gen_ldx = ldx; p.gen_sg = sg; p.gen_K = (int)k_valid; p.gen_q = q; p.gen_lut = lut; p.gen_nb = 1 << n_bits;
    const int64_t waves = (int64_t)L.wgs * 4;
    int nch = (int)cdiv(3 * waves, G);
    if (nch < cdiv(N, AVQ_ROWS)) nch = cdiv(N, AVQ_ROWS);           // <= AVQ_ROWS attention rows per item (they are staged in LDS)
    nch = nch < 1 ? 1 : nch > N ? N : nch;
    const int cbc = cdiv(N, nch);
    p.slab_R = cbc; p.slab_U = cdiv(N, cbc);
    const int nks = k_valid <= 64 ? 4 : 13;
    const size_t lut_b = (((size_t)(p.gen_nb + 1) * P + 3) & ~(size_t)3) * 4;
    const size_t rows_b = (size_t)4 * ((size_t)AVQ_ROWS * (nks * 16 + 64) + 4 * 64 * 2) * 4;
    const size_t acc_b = (size_t)gmod * 256 * 8;
    const size_t shm = lut_b + rows_b > acc_b ? lut_b + rows_b : acc_b;
    hipStream_t st = (hipStream_t)stream;
#define LAUNCH_AVQ(NKSV, RBV)                                                                                     \
    do {                                                                                                          \
        static unsigned long long attr_dev = 0; \
        { hipError_t ea__ = adalog_max_lds(reinterpret_cast<const void*>(&k_gemm_avq<4, NKSV, RBV>), (int)(80 * 1024), &attr_dev); \
          if (ea__ != hipSuccess) { adalog_set_error("hipFuncSetAttribute", ea__); return (int)ea__; } } \
        adalog_note_kernel(NKSV == 13 ? "k_gemm_avq<13,bf16>" : "k_gemm_avq<4,bf16>");                              \
        hipLaunchKernelGGL((k_gemm_avq<4, NKSV, RBV>), dim3((unsigned)L.wgs), dim3(256), shm, st, p);             \
    } while (0)
    if (nks == 13) { if (M > 32) LAUNCH_AVQ(13, 2); else LAUNCH_AVQ(13, 1); }
    else { if (M > 32) LAUNCH_AVQ(4, 2); else LAUNCH_AVQ(4, 1); }
#undef LAUNCH_AVQ
    ADALOG_LAUNCH_CHECK("adalog_gemm_score_avq");
    return 0;
}

// 1 when adalog_gemm_score_gen takes this shape (M rows of the fixed operand, N = source rows x ref_div candidate columns).
extern "C" int adalog_gemm_score_gen_ok(int dtype, int M, int N, int G, int gmod, int ref_div, int64_t k_valid, int64_t Kp) {
    if (!(dtype == 0 || dtype == 3) || k_valid < 16 || k_valid > 64 || k_valid % 16 || ref_div < 1 || N % ref_div != 0 || G % gmod) return 0;
    const Layout L = layout_of(M, N, 1, G, gmod, ref_div, 1, true, k_valid, Kp, true, dtype);
    if (!(L.stream && L.acc) || L.slab) return 0;
    if (win_ok(dtype, M, N, G, gmod, ref_div, k_valid, nullptr, nullptr, 0, M, L.wgs)) return 1;
    return (grp_ok(dtype, M, N, G, gmod, ref_div, k_valid, nullptr, nullptr, 0, M) && grpw_on() && L.wgs * 4 >= gmod) ? 1 : 0;
}

// ---- activation-candidate scoring call with the candidate operand GENERATED inside the slab kernel (k_gemm_slab<.., GEN>)
// reference linear.py:394-430 (_search_best_a_scale): scores[p] = -norm * sum_{t, o} (raw_out[t][o] - bias[o] -
//   s_w[o] * s_p * sum_k Wq[o][k] * (clamp(rne(x[t][k] / s_p) + z_p, 0, 2^bits - 1) - z_p))^2
// Wp: packed weight image (q_w - z_w) [M][Kp] int8 (dtype 0) or fp8 e4m3 (dtype 3); x: fp32 [T][ldx] (K valid); ref: raw_out
// [T][M]; scale / zp: the P (64, 128 or 256) per-tensor candidates; row_scale [M] = s_w, row_bias [M] = bias (or null).
extern "C" int adalog_finish_scores(const float* partial, float* scores, int MT, int N, int Npad, int C, int G, int gmod,
                                    int keep_h, int keep_n, int cand_inner, double norm, void* workspace,
                                    int64_t workspace_bytes, void* stream);

static Layout gen_layout(int dtype, int M, int64_t T, int K, int64_t Kp, int P) {
    Layout L{};
    if (!(dtype == 0 || dtype == 3) || !(P == 64 || P == 128 || P == 256) || T < 1 || T * P >= ((int64_t)1 << 31) || K < 16 || K % 16 != 0 ||
        Kp < K) return L;
    return layout_of(M, (int)(T * P), 1, 1, 1, P, 1, true, (int64_t)K, Kp, true, dtype, true);
}

extern "C" int adalog_score_act_gen_ok(int dtype, int M, int64_t T, int K, int64_t Kp, int P) {
    static const int use_gen = getenv("ADALOG_SLAB_GEN") ? atoi(getenv("ADALOG_SLAB_GEN")) : 1;
    if (!use_gen || Kp % BK3 != 0) return 0;
    const Layout L = gen_layout(dtype, M, T, K, Kp, P);
    return (L.slab && L.acc) ? 1 : 0;
}

extern "C" int adalog_score_act_gen_wgs(int dtype, int M, int64_t T, int K, int64_t Kp, int P) {
    const Layout L = gen_layout(dtype, M, T, K, Kp, P);
    return (L.slab && L.acc) ? L.wgs : -1;
}

extern "C" int64_t adalog_score_act_gen_workspace_bytes(int dtype, int M, int64_t T, int K, int64_t Kp, int P) {
    const Layout L = gen_layout(dtype, M, T, K, Kp, P);
    if (!(L.slab && L.acc)) return -1;
    return L.elems * (int64_t)sizeof(float) + ((int64_t)M * 4 + 15) / 16 * 16;          // accumulators + a zero row-bias vector
}

extern "C" int adalog_score_act_gen(int dtype, const void* Wp, int M, int64_t Kp, const float* x, int64_t T, int K, int64_t ldx,
                                    const float* scale, const float* zp, int P, int n_bits, const float* ref,
                                    const float* row_scale, const float* row_bias, double norm, void* workspace,
                                    int64_t workspace_bytes, float* scores, void* stream) {
    ADALOG_ARG_CHECK(Wp && x && scale && zp && ref && row_scale && workspace, "score_act_gen: null pointer");
    ADALOG_ARG_CHECK(n_bits >= 2 && n_bits <= 7 && (dtype != 3 || n_bits <= 4), "score_act_gen: bad bit width for the operand type");
    ADALOG_ARG_CHECK(ldx >= K && (ldx % 4 == 0) && ((uintptr_t)x & 15) == 0, "score_act_gen: activation rows must be 16-byte aligned");
    const Layout L = gen_layout(dtype, M, T, K, Kp, P);
    ADALOG_ARG_CHECK(L.slab && L.acc && Kp % BK3 == 0, "score_act_gen: shape not taken by the slab kernel (adalog_score_act_gen_ok)");
    ADALOG_ARG_CHECK(workspace_bytes >= adalog_score_act_gen_workspace_bytes(dtype, M, T, K, Kp, P) && ((uintptr_t)workspace & 15) == 0,
                     "score_act_gen: workspace too small or misaligned");
    hipStream_t st = (hipStream_t)stream;
    float* partial = (float*)workspace;
    float* zero_bias = (float*)((uint8_t*)workspace + L.elems * sizeof(float));
    if (!row_bias) {
        const hipError_t me = hipMemsetAsync(zero_bias, 0, (size_t)M * 4, st);
        if (me != hipSuccess) { adalog_set_error("adalog_score_act_gen (zero bias)", me); return (int)me; }
        row_bias = zero_bias;
    }
    GemmArgs p{};
    p.A = (const uint8_t*)Wp; p.B = nullptr;
    p.M = M; p.N = (int)(T * P); p.Kb = Kp; p.Kvb = K; p.C = 1; p.G = 1; p.gmod = 1;
    p.ref = ref; p.ldr = 1; p.sRg = 0; p.ref_cs = M; p.ref_div = P;
    ADALOG_ARG_CHECK((int64_t)(T - 1) * M + M < ((int64_t)1 << 31), "score_act_gen: reference exceeds 32-bit addressing");
    p.sa = scale; p.sa_c = 0; p.sa_mul = 1.0f; p.sb = scale; p.sb_c = 1; p.sb_n = 0;
    p.row_scale = row_scale; p.row_bias = row_bias;
    p.MT = L.MT; p.NT = L.NT; p.Npad = L.Npad; p.order = 2; p.reduce_cols = 0;
    p.partial = partial; p.wg_acc = (double*)partial; p.timeline = g_timeline;
    p.slab_U = L.slab_U; p.slab_R = L.slab_R;
    p.gen_x = x; p.gen_ldx = ldx; p.gen_K = K; p.gen_scale = scale; p.gen_zp = zp; p.gen_sc = 1; p.gen_sn = 0; p.gen_sa = nullptr;
    p.gen_qmax = (float)((1 << n_bits) - 1);
    // tie zone: the reciprocal-multiply quotient is within ~2 ulp of the IEEE one; only |quotient| <= 2^bits matters (beyond it
    // both clamp alike), so 6e-7 * 2^bits bounds the difference with a factor of 2.5 to spare; never narrower than 1e-5
    const float zone = 6e-7f * (float)(1 << n_bits);
    p.gen_tie = 0.5f - (zone > 1e-5f ? zone : 1e-5f);
    const int nk = (int)((p.Kvb + BK3 - 1) / BK3);
    const int SBN = 32 * L.slab_nb;
    const size_t shm = (size_t)nk * SBN * BK3 + 8 * 3 * 32 * BK3 + 8 * 192 * 4 + 8 * SBN * 4;
    const int nref = SBN / P;
#define LAUNCH_GEN(NREFV, DTV, NBV)                                                                               \
    do {                                                                                                          \
        static unsigned long long attr_dev = 0; \
        { hipError_t ea__ = adalog_max_lds(reinterpret_cast<const void*>(&k_gemm_slab<NREFV, true, DTV, NBV, true>), (int)(160 * 1024), &attr_dev); \
          if (ea__ != hipSuccess) { adalog_set_error("hipFuncSetAttribute", ea__); return (int)ea__; } } \
        adalog_note_kernel(DTV == 3 ? (NBV == 8 ? "k_gemm_slab_gen<fp8>" : "k_gemm_slab128_gen<fp8>") : (NBV == 8 ? "k_gemm_slab_gen<i8>" : "k_gemm_slab128_gen<i8>")); \
        hipLaunchKernelGGL((k_gemm_slab<NREFV, true, DTV, NBV, true>), dim3((unsigned)L.wgs), dim3(512), shm, st, p); \
    } while (0)
#define LAUNCH_GEN_DT(DTV)                                                                                        \
    do {                                                                                                          \
        if (L.slab_nb == 8) { if (nref == 1) LAUNCH_GEN(1, DTV, 8); else if (nref == 2) LAUNCH_GEN(2, DTV, 8); else LAUNCH_GEN(4, DTV, 8); } \
        else { if (nref == 1) LAUNCH_GEN(1, DTV, 4); else LAUNCH_GEN(2, DTV, 4); }                                \
    } while (0)
    if (dtype == 3) LAUNCH_GEN_DT(3); else LAUNCH_GEN_DT(0);
#undef LAUNCH_GEN_DT
#undef LAUNCH_GEN
    ADALOG_LAUNCH_CHECK("adalog_score_act_gen");
    // fixed-order fp64 finish of the per-workgroup accumulators [wgs][1][256] (scores == null: left to the caller, who hands the
    // accumulators -- the start of the workspace -- to adalog_finish_scores / adalog_finish_topk_next with MT = adalog_score_act_gen_wgs)
    if (!scores) return 0;
    return adalog_finish_scores(partial, scores, L.wgs, BN2, BN2, P, 1, 1, 0, 0, 2, norm, nullptr, 0, stream);
}

// ---- weight-candidate scoring call with the candidate operand GENERATED inside the slab kernel
// reference linear.py:355-392 (_search_best_w_scale): score[p][o] = -norm * sum_t (raw_out[t][o] - bias[o] -
//   s_a * s_w[p][o] * sum_k (q_a(x) - z_a)[t][k] * (clamp(rne(W[o][k] / s_w[p][o]) + z_w[p][o], 0, 2^bits - 1) - z_w[p][o]))^2
// Xp: packed activation image (q_a - z_a) [T][Kp] int8 (dtype 0) or fp8 e4m3 (dtype 3); W: fp32 [O][ldw] (K valid); scale / zp:
// the P (64, 128 or 256) candidates of every output row, [P][O]; ref: raw_out TRANSPOSED [O][T]; sa: device scalar s_a; bias [O]
// or null.  The packed [O * P][Kp] candidate operand (56-117 MB per call) is neither written nor read.  partial: the layout of
// adalog_gemm_score_layout(T, O * P, 1, 1, 1, P, 0, dtype, Kp, K, 1) (per-column partial sums, kept axis = (o, p)).
static Layout wgen_layout(int dtype, int T, int O, int K, int64_t Kp, int P) {
    Layout L{};
    if (!(dtype == 0 || dtype == 3) || !(P == 64 || P == 128 || P == 256) || O < 1 || (int64_t)O * P >= ((int64_t)1 << 31) || K < 16 || K % 16 != 0 ||
        Kp < K) return L;
    return layout_of(T, O * P, 1, 1, 1, P, 0, true, (int64_t)K, Kp, true, dtype, false);
}

extern "C" int adalog_score_w_gen_ok(int dtype, int T, int O, int K, int64_t Kp, int P) {
    static const int use_gen = getenv("ADALOG_SLAB_WGEN") ? atoi(getenv("ADALOG_SLAB_WGEN")) : 1;
    if (!use_gen || Kp % BK3 != 0) return 0;
    const Layout L = wgen_layout(dtype, T, O, K, Kp, P);
    return (L.slab && !L.acc) ? 1 : 0;
}

extern "C" int adalog_score_w_gen(int dtype, const void* Xp, int T, int64_t Kp, const float* W, int O, int K, int64_t ldw,
                                  const float* scale, const float* zp, int P, int n_bits, const float* ref, const float* sa,
                                  const float* bias, float* partial, int64_t partial_elems, void* stream) {
    ADALOG_ARG_CHECK(Xp && W && scale && zp && ref && sa && partial, "score_w_gen: null pointer");
    ADALOG_ARG_CHECK(n_bits >= 2 && n_bits <= 7 && (dtype != 3 || n_bits <= 4), "score_w_gen: bad bit width for the operand type");
    ADALOG_ARG_CHECK(ldw >= K && (ldw % 4 == 0) && ((uintptr_t)W & 15) == 0, "score_w_gen: weight rows must be 16-byte aligned");
    const Layout L = wgen_layout(dtype, T, O, K, Kp, P);
    ADALOG_ARG_CHECK(L.slab && !L.acc && Kp % BK3 == 0, "score_w_gen: shape not taken by the slab kernel (adalog_score_w_gen_ok)");
    ADALOG_ARG_CHECK(partial_elems >= L.elems, "score_w_gen: partial buffer too small");
    hipStream_t st = (hipStream_t)stream;
    GemmArgs p{};
    p.A = (const uint8_t*)Xp; p.B = nullptr;
    p.M = T; p.N = O * P; p.Kb = Kp; p.Kvb = K; p.C = 1; p.G = 1; p.gmod = 1;
    p.ref = ref; p.ldr = 1; p.sRg = 0; p.ref_cs = T; p.ref_div = P;
    ADALOG_ARG_CHECK((int64_t)(O - 1) * T + T < ((int64_t)1 << 31), "score_w_gen: reference exceeds 32-bit addressing");
    p.sa = sa; p.sa_c = 0; p.sa_mul = 1.0f; p.sb = scale; p.sb_c = O; p.sb_n = 1;
    p.bias = bias; p.bi_c = 0; p.bi_g = 0; p.bi_n = 1;
    p.MT = L.MT; p.NT = L.NT; p.Npad = L.Npad; p.order = 2; p.reduce_cols = 0;
    p.partial = partial; p.timeline = g_timeline;
    p.slab_U = L.slab_U; p.slab_R = L.slab_R;
    p.gen_x = W; p.gen_ldx = ldw; p.gen_K = K; p.gen_scale = scale; p.gen_zp = zp; p.gen_sc = O; p.gen_sn = 1; p.gen_sa = sa;
    p.gen_qmax = (float)((1 << n_bits) - 1);
    const float zone = 6e-7f * (float)(1 << n_bits);
    p.gen_tie = 0.5f - (zone > 1e-5f ? zone : 1e-5f);
    const int nk = (int)((p.Kvb + BK3 - 1) / BK3);
    const int SBN = 32 * L.slab_nb;
    const size_t shm = (size_t)nk * SBN * BK3 + 8 * 3 * 32 * BK3 + 8 * 192 * 4 + 8 * SBN * 4;
    const int nref = SBN / P;
    {   // a slab that is not cut has unused pieces: they must read as zero
        const hipError_t me = hipMemsetAsync(partial, 0, (size_t)L.elems * sizeof(float), st);
        if (me != hipSuccess) { adalog_set_error("adalog_score_w_gen (clear partials)", me); return (int)me; }
    }
#define LAUNCH_WGEN(NREFV, DTV, NBV)                                                                              \
    do {                                                                                                          \
        static unsigned long long attr_dev = 0; \
        { hipError_t ea__ = adalog_max_lds(reinterpret_cast<const void*>(&k_gemm_slab<NREFV, false, DTV, NBV, true>), (int)(160 * 1024), &attr_dev); \
          if (ea__ != hipSuccess) { adalog_set_error("hipFuncSetAttribute", ea__); return (int)ea__; } } \
        adalog_note_kernel(DTV == 3 ? (NBV == 8 ? "k_gemm_slab_wgen<fp8>" : "k_gemm_slab128_wgen<fp8>") : (NBV == 8 ? "k_gemm_slab_wgen<i8>" : "k_gemm_slab128_wgen<i8>")); \
        hipLaunchKernelGGL((k_gemm_slab<NREFV, false, DTV, NBV, true>), dim3((unsigned)L.wgs), dim3(512), shm, st, p); \
    } while (0)
#define LAUNCH_WGEN_DT(DTV)                                                                                       \
    do {                                                                                                          \
        if (L.slab_nb == 8) { if (nref == 1) LAUNCH_WGEN(1, DTV, 8); else if (nref == 2) LAUNCH_WGEN(2, DTV, 8); else LAUNCH_WGEN(4, DTV, 8); } \
        else { if (nref == 1) LAUNCH_WGEN(1, DTV, 4); else LAUNCH_WGEN(2, DTV, 4); }                              \
    } while (0)
    if (dtype == 3) LAUNCH_WGEN_DT(3); else LAUNCH_WGEN_DT(0);
#undef LAUNCH_WGEN_DT
#undef LAUNCH_WGEN
    ADALOG_LAUNCH_CHECK("adalog_score_w_gen");
    return 0;
}

// scores[c][h?][n?] = -norm * sum over (image, [h], m_tile, [n]) of partial[c][g][m_tile][n] with the layout returned by
// adalog_gemm_score_layout (MT, Npad); N = number of valid entries along the last axis (n_eff, or NT when reduced).
extern "C" int64_t adalog_finish_workspace_bytes(int MT, int N, int C, int G, int keep_n, int cand_inner) {
    if (cand_inner != 1 || keep_n || !(C == 64 || C == 128 || C == 256)) return 0;
    return (int64_t)G * MT * cdiv(N, FSEG) * C * (int64_t)sizeof(double);
}

extern "C" int adalog_finish_scores(const float* partial, float* scores, int MT, int N, int Npad, int C, int G, int gmod,
                                    int keep_h, int keep_n, int cand_inner, double norm, void* workspace,
                                    int64_t workspace_bytes, void* stream) {
    ADALOG_ARG_CHECK(partial && scores && MT >= 1 && N >= 1 && Npad >= N && C >= 1 && G >= 1 && gmod >= 1 && G % gmod == 0,
                     "finish_scores: bad arguments");
    FinishArgs p{};
    p.partial = partial; p.scores = scores; p.C = C; p.G = G; p.gmod = gmod; p.MT = MT; p.N = N; p.Npad = Npad;
    p.keep_h = keep_h; p.keep_n = keep_n; p.norm = norm; p.cin = cand_inner ? C : 0;
    if (cand_inner == 2) {                  // per-workgroup accumulators: partial = double [MT = workgroups][gmod][256]
        ADALOG_ARG_CHECK(!keep_n && (C == 64 || C == 128 || C == 256) && Npad == 256, "finish_scores: bad accumulator layout");
        const int64_t nout2 = (int64_t)C * (keep_h ? gmod : 1);
        hipLaunchKernelGGL(k_finish_wgacc, dim3((unsigned)nout2), dim3(256), 0, (hipStream_t)stream, p,
                           (const double*)partial, MT);
        ADALOG_LAUNCH_CHECK("adalog_finish_scores");
        return 0;
    }
    const int64_t nout = (int64_t)C * (keep_h ? gmod : 1) * (keep_n ? N : 1);
    const int64_t per_out = (int64_t)(G / gmod) * (keep_h ? 1 : gmod) * MT * (keep_n ? 1 : N);
    const int64_t need = adalog_finish_workspace_bytes(MT, N, C, G, keep_n, cand_inner);
    if (need > 0 && workspace && workspace_bytes >= need && per_out >= 1024) {
        const int nseg = cdiv(N, FSEG);
        hipLaunchKernelGGL(k_finish_rows, dim3((unsigned)nseg, (unsigned)(G * MT)), dim3(256), 0, (hipStream_t)stream, p,
                           (double*)workspace, nseg);
        hipLaunchKernelGGL(k_finish_stage2, dim3((unsigned)((nout + 3) / 4)), dim3(256), 0, (hipStream_t)stream, p,
                           (const double*)workspace, nseg);
    } else if (p.cin > 0 && per_out <= 512 && nout >= 4096)
        hipLaunchKernelGGL(k_finish_tpo, dim3((unsigned)((nout + 255) / 256)), dim3(256), 0, (hipStream_t)stream, p);
    else if (per_out >= 2048)
        hipLaunchKernelGGL(k_finish<false>, dim3((unsigned)nout), dim3(256), 0, (hipStream_t)stream, p);
    else
        hipLaunchKernelGGL(k_finish<true>, dim3((unsigned)((nout + 3) / 4)), dim3(256), 0, (hipStream_t)stream, p);
    ADALOG_LAUNCH_CHECK("adalog_finish_scores");
    return 0;
}

// finish + top-k + next grid in one launch where the layout allows it (see gemm_finish.inc); otherwise adalog_finish_scores
// followed by adalog_topk_next.  Arguments: those of adalog_finish_scores, then those of adalog_topk_next (scores [C][cols],
// cols = (keep_h ? gmod : 1) * (keep_n ? N : 1)).  Single-GPU only: with several ranks the scores are all-reduced between the two.
extern "C" int adalog_topk_next_tail(const float* scores, int P, int cols, const adalog_fpcs_tail* tail, int* idx_out, void* stream);

extern "C" int adalog_finish_topk_next_tail(const float* partial, float* scores, int MT, int N, int Npad, int C, int G, int gmod,
                                            int keep_h, int keep_n, int cand_inner, double norm, void* workspace, int64_t workspace_bytes,
                                            const adalog_fpcs_tail* tail, void* stream) {
    ADALOG_ARG_CHECK(partial && scores && tail && MT >= 1 && N >= 1 && Npad >= N && C >= 1 && C <= 256 && G >= 1 &&
                     gmod >= 1 && G % gmod == 0, "finish_topk_next: bad arguments");
    const char* why = fpcs::tail_problem(tail, C);
    ADALOG_ARG_CHECK(why == nullptr, why);
    static const int use_fused = getenv("ADALOG_FINISH_TOPK") ? atoi(getenv("ADALOG_FINISH_TOPK")) : 1;
    const int nh = keep_h ? gmod : 1, nn = keep_n ? N : 1, cols = nh * nn;
    FinishArgs p{};
    p.partial = partial; p.scores = scores; p.C = C; p.G = G; p.gmod = gmod; p.MT = MT; p.N = N; p.Npad = Npad;
    p.keep_h = keep_h; p.keep_n = keep_n; p.norm = norm; p.cin = cand_inner ? C : 0;
    const TopkArgs t = *tail;
    hipStream_t st = (hipStream_t)stream;
    const bool c_ok = (C == 64 || C == 128 || C == 256);
    if (use_fused && cand_inner == 2 && !keep_n && c_ok && Npad == 256 && nh <= 64) {
        unsigned int* ticket = adalog_ticket_slots_on(nh, stream);
        if (ticket) {
            hipLaunchKernelGGL(k_finish_wgacc_topk, dim3((unsigned)(C * nh)), dim3(256), 0, st, p, (const double*)partial, MT, t, ticket);
            ADALOG_LAUNCH_CHECK("adalog_finish_topk_next");
            return 0;
        }
    }
    const int64_t nout = (int64_t)C * cols;
    const int64_t per_out = (int64_t)(G / gmod) * (keep_h ? 1 : gmod) * MT * (keep_n ? 1 : N);
    if (use_fused && cand_inner == 1 && c_ok && per_out <= 512 && nout >= 4096) {
        const int gpb = 256 / C;
        hipLaunchKernelGGL(k_finish_tpo_topk, dim3((unsigned)((cols + gpb - 1) / gpb)), dim3(256), 0, st, p, t);
        ADALOG_LAUNCH_CHECK("adalog_finish_topk_next");
        return 0;
    }
    const int rc = adalog_finish_scores(partial, scores, MT, N, Npad, C, G, gmod, keep_h, keep_n, cand_inner, norm, workspace,
                                        workspace_bytes, stream);
    if (rc) return rc;
    return adalog_topk_next_tail(scores, C, cols, tail, nullptr, stream);
}

extern "C" int adalog_finish_topk_next(const float* partial, float* scores, int MT, int N, int Npad, int C, int G, int gmod,
                                       int keep_h, int keep_n, int cand_inner, double norm, void* workspace, int64_t workspace_bytes,
                                       int k, const float* scale, const float* zp, const float* third, int new_cnt,
                                       const float* lin, float* delta, int has_clamp, float clamp_min, float* out_scale,
                                       float* out_zp, float* out_third, void* stream) {
    const adalog_fpcs_tail t{k, new_cnt, has_clamp, clamp_min, scale, zp, third, lin, delta, delta, out_scale, out_zp, out_third};
    return adalog_finish_topk_next_tail(partial, scores, MT, N, Npad, C, G, gmod, keep_h, keep_n, cand_inner, norm, workspace,
                                        workspace_bytes, &t, stream);
}
