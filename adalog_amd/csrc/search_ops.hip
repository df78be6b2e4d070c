// Search-side kernels of the Fast Progressive Combining Search (FPCS), all tiny or HBM-bound:
//   K16  adalog_topk / adalog_fpcs_next / adalog_candidate_grid   <- linear.py:432-451,483-523; matmul.py:231-262;
//                                                                     conv.py:281-311  (topk + gather + next 16x8 grid)
//   K9   adalog_score_w_self                                       <- linear.py:296-309
//   K10  adalog_score_a_self                                       <- linear.py:320-345
// Candidate tensors live on the device as fp32 [P][cols] (candidate-major), so one FPCS call is a chain of
// launches on one stream with no host synchronisation; the winner is committed by the kernel itself.
#include "common.h"
#include "fpcs_tail.h"

namespace {

// ---------------------------------------------------------------------------------------------- top-k (K16)
// Deterministic: order by (score desc, candidate index asc); NaN ranks first like torch.topk.
__global__ __launch_bounds__(256) void k_topk(const float* __restrict__ scores, int P, int cols, int k,
                                              int* __restrict__ idx) {
    __shared__ float s[256];
    const int col = blockIdx.x, p = threadIdx.x;
    if (p < P) s[p] = scores[(int64_t)p * cols + col];
    __syncthreads();
    if (p >= P) return;
    const float me = s[p];
    const bool me_nan = me != me;
    int rank = 0;
    for (int j = 0; j < P; ++j) {
        const float o = s[j];
        const bool o_nan = o != o;
        bool before;
        if (o_nan || me_nan) before = (o_nan && !me_nan) || (o_nan && me_nan && j < p);
        else before = (o > me) || (o == me && j < p);
        rank += before ? 1 : 0;
    }
    if (rank < k) idx[(int64_t)rank * cols + col] = p;
}

// gather survivors, then either commit the winner (k == 1) or emit the next survivor-major grid.
// A block is CL column lanes x OL output lanes (CL = min(64, cols rounded up to a power of two), CL * OL = 256): the
// k * new_cnt grid points of a column are spread over the output lanes (a per-tensor search has ONE column: a thread
// per column walked its 128 points alone for 14 us), and neighbouring threads write neighbouring columns.
__global__ __launch_bounds__(256) void k_fpcs_next(const float* __restrict__ scale, const float* __restrict__ zp,
                                                   const float* __restrict__ third, int cols, const int* __restrict__ idx,
                                                   int k, int new_cnt, const float* __restrict__ lin,
                                                   float* __restrict__ delta, float clamp_min, int has_clamp,
                                                   float* __restrict__ o_scale, float* __restrict__ o_zp,
                                                   float* __restrict__ o_third, int CL) {
    const int cl = threadIdx.x % CL, ol = threadIdx.x / CL, OL = 256 / CL;
    const int col = blockIdx.x * CL + cl;
    const bool live = col < cols;
    if (new_cnt == 0) {                                   // commit (topk == 1 branch, linear.py:387-391)
        if (live && ol == 0) {
            const int p = idx[col];
            o_scale[col] = scale[(int64_t)p * cols + col];
            if (zp) o_zp[col] = zp[(int64_t)p * cols + col];
            if (third) o_third[col] = third[(int64_t)p * cols + col];
        }
        return;
    }
    const float d = live ? delta[col] : 0.0f;
    if (live) {
        const int total = k * new_cnt;
        for (int o = ol; o < total; o += OL) {
            const int j = o / new_cnt, i = o - j * new_cnt;
            const int p = idx[(int64_t)j * cols + col];
            const float ts = scale[(int64_t)p * cols + col];
            float v = ts + (lin[i] - 0.5f) * d;                // linear.py:492-495
            if (has_clamp) v = fmaxf(v, clamp_min);            // linear.py:516
            const int64_t oo = (int64_t)o * cols + col;
            o_scale[oo] = v;
            if (zp) o_zp[oo] = zp[(int64_t)p * cols + col];
            if (third) o_third[oo] = third[(int64_t)p * cols + col];
        }
    }
    __syncthreads();                                          // every lane of this column has read delta
    if (live && ol == 0) delta[col] = d / ((float)new_cnt - 0.5f);   // linear.py:493
}

// top-k and the next grid in ONE launch (every FPCS step needs both, and each is a 5-10 us launch for microseconds of
// work): one workgroup per column ranks its P <= 256 candidates in LDS (deterministic order of k_topk), then the same workgroup
// gathers the k survivors and writes the k * new_cnt grid points of its column (or commits the winner) -- fpcs_tail.h, the
// code the scoring kernels run themselves where they produce final scores.
__global__ __launch_bounds__(256) void k_topk_next(const float* __restrict__ scores, int P, int cols, fpcs::Tail t,
                                                   int* __restrict__ idx_out) {
    __shared__ float s[256];
    __shared__ int top[256];
    const int col = blockIdx.x;
    fpcs::column<256, false, false>(scores, P, cols, col, (int)threadIdx.x, t, s, top);
    if (idx_out && (int)threadIdx.x < t.k) idx_out[(int64_t)threadIdx.x * cols + col] = top[threadIdx.x];
}

// initial percentile grid: quant = [4][cols] = {Q_hi0, Q_hi1, Q_lo0, Q_lo1} (e.g. Q.9, Q1.0, Q.1, Q0)
__global__ __launch_bounds__(256) void k_candidate_grid(const float* __restrict__ quant, int cols, int num_scale,
                                                        int num_zp, int zp_min, float denom,
                                                        const float* __restrict__ lin, float clamp_min, int has_clamp,
                                                        float* __restrict__ scale, float* __restrict__ zp,
                                                        float* __restrict__ delta) {
    const int col = blockIdx.x * blockDim.x + threadIdx.x;
    if (col >= cols) return;
    const float dmin = quant[col] - quant[2 * cols + col];
    const float dmax = quant[cols + col] - quant[3 * cols + col];
    const float span = dmax - dmin;
    float s0 = 0.f, s1 = 0.f;
    for (int si = 0; si < num_scale; ++si) {
        float v = (dmin + lin[si] * span) / denom;
        if (has_clamp) v = fmaxf(v, clamp_min);
        if (si == 0) s0 = v;
        if (si == 1) s1 = v;
        for (int zi = 0; zi < num_zp; ++zi) {
            const int64_t o = (int64_t)(zi * num_scale + si) * cols + col;
            scale[o] = v;
            zp[o] = (float)(zp_min + zi);
        }
    }
    delta[col] = s1 - s0;
}

// ---------------------------------------------------------------------------------------------- K9
// scores[p][row] = -mean_i (w - fq_p(w))^2 ; one workgroup per weight row, one thread per candidate.
__global__ __launch_bounds__(256) void k_score_w_self(const float* __restrict__ w, int rows, int I,
                                                      const float* __restrict__ scale, const float* __restrict__ zp,
                                                      int P, float qmax, float* __restrict__ scores) {
    extern __shared__ float wrow[];
    const int row = blockIdx.x;
    for (int i = threadIdx.x; i < I; i += blockDim.x) wrow[i] = w[(int64_t)row * I + i];
    __syncthreads();
    const int p = threadIdx.x;
    if (p >= P) return;
    const float s = scale[(int64_t)p * rows + row], z = zp[(int64_t)p * rows + row];
    float acc = 0.0f;
    for (int i = 0; i < I; ++i) {
        const float v = wrow[i];
        const float dq = (fminf(fmaxf(rintf(v / s) + z, 0.0f), qmax) - z) * s;
        const float e = v - dq;
        acc += e * e;
    }
    scores[(int64_t)p * rows + row] = -(acc / (float)I);
}

// ---------------------------------------------------------------------------------------------- K10
// partial[p][slab][ch] = sum over the slab's rows of (x - fq_p(x))^2 ; x is read from HBM once for all P candidates.
constexpr int SLAB_ROWS_PER_THREAD = 32, SLAB_ROWS = 4 * SLAB_ROWS_PER_THREAD;
__global__ __launch_bounds__(256) void k_score_a_self(const float* __restrict__ x, int64_t rows, int I,
                                                      const float* __restrict__ scale, const float* __restrict__ zp,
                                                      int P, int pstride_ch, float qmax, float* __restrict__ partial,
                                                      int n_slab, int Cpad) {
    __shared__ float red[2][4][64];              // double-buffered by candidate parity: one barrier per candidate
    const int chl = threadIdx.x & 63, rg = threadIdx.x >> 6;
    const int ch = blockIdx.x * 64 + chl;
    const int slab = blockIdx.y;
    const bool cv = ch < I;
    float xv[SLAB_ROWS_PER_THREAD];
    const int64_t r0 = (int64_t)slab * SLAB_ROWS + rg;
#pragma unroll
    for (int r = 0; r < SLAB_ROWS_PER_THREAD; ++r) {
        const int64_t row = r0 + 4 * r;
        xv[r] = (cv && row < rows) ? x[row * I + ch] : __builtin_nanf("");
    }
    const int cols = pstride_ch ? I : 1;
    const bool slab_full = cv && (r0 + 4 * (SLAB_ROWS_PER_THREAD - 1) < rows);
    const int64_t pcol = pstride_ch ? (cv ? ch : 0) : 0;
    float s_n = scale[pcol], z_n = zp[pcol];     // the next candidate's parameters are fetched under the current one's math
    for (int p = 0; p < P; ++p) {
        const float s = s_n, z = z_n;
        if (p + 1 < P) { s_n = scale[(int64_t)(p + 1) * cols + pcol]; z_n = zp[(int64_t)(p + 1) * cols + pcol]; }
        float acc = 0.0f;
        if (slab_full && rintf(z) == z) {
            // reciprocal fast path; inside the tie zone (|frac - 0.5| < 1e-4) the exact IEEE quotient decides, so the bin
            // is the one rintf(v / s) gives.  clamp(k + z, 0, qmax) - z == med3(k, -z, qmax - z) for integral z.
            // Two rows per step on packed fp32 math (v_pk_mul / v_pk_add; no contraction in this file): 6 VALU per
            // element-candidate instead of 9.
            typedef float v2f __attribute__((ext_vector_type(2)));
            const float inv_s = __builtin_amdgcn_rcpf(s), lo = -z, hi = qmax - z;
            const v2f inv2 = {inv_s, inv_s}, s2 = {s, s};
            v2f acc2 = {0.0f, 0.0f};
#pragma unroll
            for (int r = 0; r < SLAB_ROWS_PER_THREAD; r += 2) {
                const v2f v = {xv[r], xv[r + 1]};
                const v2f t = v * inv2;
                v2f k = {rintf(t.x), rintf(t.y)};
                const v2f d = t - k;
                if (__builtin_expect(fmaxf(fabsf(d.x), fabsf(d.y)) > 0.4999f, 0)) {      // one branch per pair
                    k.x = rintf(v.x / s);
                    k.y = rintf(v.y / s);
                }
                const v2f kq = {__builtin_amdgcn_fmed3f(k.x, lo, hi), __builtin_amdgcn_fmed3f(k.y, lo, hi)};
                const v2f e = v - kq * s2;
                acc2 += e * e;
            }
            acc = acc2.x + acc2.y;
        } else {
#pragma unroll
            for (int r = 0; r < SLAB_ROWS_PER_THREAD; ++r) {
                const float v = xv[r];
                const float dq = (fminf(fmaxf(rintf(v / s) + z, 0.0f), qmax) - z) * s;
                const float e = v - dq;
                acc += (v == v) ? e * e : 0.0f;
            }
        }
        float (*rb)[64] = red[p & 1];
        rb[rg][chl] = acc;
        __syncthreads();                          // the buffer of candidate p - 1 is free again after this barrier
        if (rg == 0 && cv)
            partial[((int64_t)p * n_slab + slab) * Cpad + ch] = (rb[0][chl] + rb[1][chl]) + (rb[2][chl] + rb[3][chl]);
    }
}

// generic fp64 finishing reduction over partial[c][mt][Npad]: scores[c][n?] = -norm * sum
__global__ __launch_bounds__(256) void k_finish_simple(const float* __restrict__ partial, float* __restrict__ scores,
                                                       int MT, int N, int Npad, int keep_n, double norm) {
    __shared__ double sm[256];
    const int nn = keep_n ? N : 1;
    const int n = blockIdx.x % nn, c = blockIdx.x / nn;
    const int n_cnt = keep_n ? 1 : N;
    const int64_t total = (int64_t)MT * n_cnt;
    double acc = 0.0;
    for (int64_t i = threadIdx.x; i < total; i += 256) {
        const int ni = (int)(i % n_cnt), mt = (int)(i / n_cnt);
        acc += (double)partial[((int64_t)c * MT + mt) * Npad + (keep_n ? n : ni)];
    }
    sm[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) sm[threadIdx.x] += sm[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) scores[blockIdx.x] = (float)(-norm * sm[0]);
}

}  // namespace

extern "C" int adalog_topk(const float* scores, int P, int cols, int k, int* idx, void* stream) {
    ADALOG_ARG_CHECK(scores && idx && P >= 1 && P <= 256 && cols >= 1 && k >= 1 && k <= P, "topk: bad arguments");
    hipLaunchKernelGGL(k_topk, dim3(cols), dim3(256), 0, (hipStream_t)stream, scores, P, cols, k, idx);
    ADALOG_LAUNCH_CHECK("adalog_topk");
    return 0;
}

extern "C" int adalog_fpcs_next(const float* scale, const float* zp, const float* third, int cols, const int* idx, int k,
                                int new_cnt, const float* lin, float* delta, int has_clamp, float clamp_min,
                                float* out_scale, float* out_zp, float* out_third, void* stream) {
    ADALOG_ARG_CHECK(scale && idx && out_scale && cols >= 1 && k >= 1, "fpcs_next: bad arguments");
    ADALOG_ARG_CHECK(new_cnt == 0 || (lin && delta), "fpcs_next: expansion needs lin and delta");
    ADALOG_ARG_CHECK((zp == nullptr) == (out_zp == nullptr) && (third == nullptr) == (out_third == nullptr),
                     "fpcs_next: in/out parameter planes must match");
    int CL = 1;
    while (CL < 64 && CL < cols) CL <<= 1;
    hipLaunchKernelGGL(k_fpcs_next, dim3(cdiv(cols, CL)), dim3(256), 0, (hipStream_t)stream, scale, zp, third, cols, idx, k,
                       new_cnt, lin, delta, clamp_min, has_clamp, out_scale, out_zp, out_third, CL);
    ADALOG_LAUNCH_CHECK("adalog_fpcs_next");
    return 0;
}

extern "C" int adalog_topk_next_tail(const float* scores, int P, int cols, const adalog_fpcs_tail* tail, int* idx_out, void* stream) {
    ADALOG_ARG_CHECK(scores && tail && P >= 1 && P <= 256 && cols >= 1, "topk_next: bad arguments (P <= 256)");
    const char* why = fpcs::tail_problem(tail, P);
    ADALOG_ARG_CHECK(why == nullptr, why);
    hipLaunchKernelGGL(k_topk_next, dim3(cols), dim3(256), 0, (hipStream_t)stream, scores, P, cols, *tail, idx_out);
    ADALOG_LAUNCH_CHECK("adalog_topk_next");
    return 0;
}

extern "C" int adalog_topk_next(const float* scores, int P, int cols, int k, const float* scale, const float* zp,
                                const float* third, int new_cnt, const float* lin, float* delta, int has_clamp,
                                float clamp_min, float* out_scale, float* out_zp, float* out_third, int* idx_out, void* stream) {
    const adalog_fpcs_tail t{k, new_cnt, has_clamp, clamp_min, scale, zp, third, lin, delta, delta, out_scale, out_zp, out_third};
    return adalog_topk_next_tail(scores, P, cols, &t, idx_out, stream);
}

extern "C" int adalog_candidate_grid(const float* quant4, int cols, int num_scale, int num_zp, int zp_min, int n_bits,
                                     const float* lin, int has_clamp, float clamp_min, float* scale, float* zp,
                                     float* delta, void* stream) {
    ADALOG_ARG_CHECK(quant4 && lin && scale && zp && delta && cols >= 1 && num_scale >= 2 && num_zp >= 1,
                     "candidate_grid: bad arguments");
    const float denom = (float)((1 << n_bits) - 1);
    hipLaunchKernelGGL(k_candidate_grid, dim3(cdiv(cols, 256)), dim3(256), 0, (hipStream_t)stream, quant4, cols, num_scale,
                       num_zp, zp_min, denom, lin, clamp_min, has_clamp, scale, zp, delta);
    ADALOG_LAUNCH_CHECK("adalog_candidate_grid");
    return 0;
}

extern "C" int adalog_score_w_self(const float* w, int rows, int I, const float* scale, const float* zp, int P, int n_bits,
                                   float* scores, void* stream) {
    ADALOG_ARG_CHECK(w && scale && zp && scores && rows >= 1 && I >= 1 && P >= 1 && P <= 256, "score_w_self: bad arguments");
    ADALOG_ARG_CHECK((size_t)I * 4 <= 160 * 1024 - 1024, "score_w_self: row too long for LDS staging");
    const float qmax = (float)((1 << n_bits) - 1);
    hipLaunchKernelGGL(k_score_w_self, dim3(rows), dim3(256), (size_t)I * 4, (hipStream_t)stream, w, rows, I, scale, zp, P,
                       qmax, scores);
    ADALOG_LAUNCH_CHECK("adalog_score_w_self");
    return 0;
}

extern "C" int64_t adalog_score_a_self_partial_elems(int64_t rows, int I, int P) {
    const int64_t n_slab = (rows + SLAB_ROWS - 1) / SLAB_ROWS;
    const int64_t Cpad = ((I + 63) / 64) * 64;
    return (int64_t)P * n_slab * Cpad;
}

// scores: [P][I] when channel_wise, else [P][1].  norm = 1/T (channel-wise) or 1/(T*I) (per tensor).
extern "C" int adalog_score_a_self(const float* x, int64_t rows, int I, const float* scale, const float* zp, int P,
                                   int channel_wise, int n_bits, double norm, float* partial, int64_t partial_elems,
                                   float* scores, void* stream) {
    ADALOG_ARG_CHECK(x && scale && zp && partial && scores && rows >= 1 && I >= 1 && P >= 1, "score_a_self: bad arguments");
    const int n_slab = (int)((rows + SLAB_ROWS - 1) / SLAB_ROWS);
    const int Cpad = ((I + 63) / 64) * 64;
    ADALOG_ARG_CHECK(partial_elems >= (int64_t)P * n_slab * Cpad, "score_a_self: partial buffer too small");
    ADALOG_ARG_CHECK(n_slab <= 65535, "score_a_self: too many rows for one launch");
    const float qmax = (float)((1 << n_bits) - 1);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(k_score_a_self, dim3(Cpad / 64, n_slab), dim3(256), 0, st, x, rows, I, scale, zp, P,
                       channel_wise ? 1 : 0, qmax, partial, n_slab, Cpad);
    ADALOG_LAUNCH_CHECK("adalog_score_a_self");
    const int nout = P * (channel_wise ? I : 1);
    hipLaunchKernelGGL(k_finish_simple, dim3(nout), dim3(256), 0, st, partial, scores, n_slab, I, Cpad, channel_wise ? 1 : 0, norm);
    ADALOG_LAUNCH_CHECK("adalog_score_a_self/finish");
    return 0;
}
