// K9 / K10 in sorted-prefix form -- the self-MSE searches
//   quant_layers/linear.py:296-318 (_search_best_w_scale_self)   scores[p][row] = -mean_i (W - fq_p(W))^2
//   quant_layers/linear.py:320-353 (_search_best_a_scale_self)   scores[p][col] = -sum_images mean_tokens (x - fq_p(x))^2
// without touching every element once per candidate.
//
// The reference evaluates `fq_p` on the whole tensor for each of the 128 candidates of each of the 6 FPCS steps (768
// passes); the first HIP form (search_ops.hip k_score_a_self) read x once per step but still spent 128 x ~6 VALU per element
// and sat at its VALU floor (0.01 of the HBM roofline, 6.4 % of a calibration).  A uniform quantiser is a monotone step
// function: for a candidate (s, z) the elements that land on level k form ONE contiguous run of the SORTED tensor.  So
//   1. sort every segment once per captured tensor (per tensor: one segment; per channel / per weight row: one each) --
//      csrc/radix_sort.hip (hipCUB until round 5);
//   2. exclusive prefix sums of x and x^2 along the sorted order, in fp64 (x^2 of an fp32 is exact in fp64);
//   3. per (segment, candidate): one thread per level finds the first element of its run by bisection WITH THE EXACT
//      PREDICATE rne(fl32(x / s)) >= k (IEEE divide, round-half-even: the reference's own op sequence, uniform.py:29) and
//      takes  sum (x - c)^2 = S2 - 2 c S1 + n c^2  over the run from the prefix sums, c = fl32((q - z) * s).
// Per step that is (levels x candidates x segments) bisections instead of (elements x candidates) quantisations: deit_small's
// 6304 x 384 activation costs 1 920 bisections per step instead of 310 M element-candidates.  Steps 1 and 2 are memoised
// per tensor by the caller (ops.SortedPrefix), so the six FPCS steps of a search share them.
// Exactness: the run boundaries are exact (same predicate as the reference); the sums are fp64 where the reference
// rounds every (x - c) and every square to fp32 -- agreement ~1e-7 relative, inside the 1e-4 bar of the parity tests.
#include "common.h"
#include "fpcs_tail.h"
#include <stdlib.h>

extern "C" int64_t adalog_sort_workspace_bytes(int64_t S, int64_t n, int with_perm);
extern "C" int adalog_sort_f32(const float* x, int64_t S, int64_t n, float* sorted, unsigned int* perm, void* workspace,
                               int64_t workspace_bytes, void* stream);
extern "C" int adalog_topk_next_tail(const float* scores, int P, int cols, const adalog_fpcs_tail* tail, int* idx_out, void* stream);

namespace {

typedef double d2 __attribute__((ext_vector_type(2)));
constexpr int PB = 1024;                 // elements per prefix block (256 threads x 4)

__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// pass 1: bsum[seg][b] = (sum x, sum x^2) of block b
__global__ __launch_bounds__(256) void k_sp_blocksum(const float* __restrict__ sorted, int64_t n, int nb, d2* __restrict__ bsum) {
    __shared__ double sm[2][4];
    const int64_t seg = blockIdx.y, b = blockIdx.x;
    const float* x = sorted + seg * n;
    const int64_t i0 = b * PB + (int64_t)threadIdx.x * 4;
    double s1 = 0.0, s2 = 0.0;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        if (i0 + e < n) { const double v = (double)x[i0 + e]; s1 += v; s2 += v * v; }
    }
    s1 = wave_sum_d(s1); s2 = wave_sum_d(s2);
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { sm[0][w] = s1; sm[1][w] = s2; }
    __syncthreads();
    if (threadIdx.x == 0) {
        d2 o = {(sm[0][0] + sm[0][1]) + (sm[0][2] + sm[0][3]), (sm[1][0] + sm[1][1]) + (sm[1][2] + sm[1][3])};
        bsum[seg * nb + b] = o;
    }
}

// pass 2: exclusive scan of a segment's block sums, in place (one workgroup per segment, chunks of 256 with a carry)
__global__ __launch_bounds__(256) void k_sp_blockscan(d2* __restrict__ bsum, int nb) {
    __shared__ double sm[2][4];
    d2* p = bsum + (int64_t)blockIdx.x * nb;
    double c1 = 0.0, c2 = 0.0;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int base = 0; base < nb; base += 256) {
        const int i = base + threadIdx.x;
        d2 v = {0.0, 0.0};
        if (i < nb) v = p[i];
        double a1 = v.x, a2 = v.y;                                    // inclusive wave scan
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const double u1 = __shfl_up(a1, o), u2 = __shfl_up(a2, o);
            if (lane >= o) { a1 += u1; a2 += u2; }
        }
        if (lane == 63) { sm[0][w] = a1; sm[1][w] = a2; }
        __syncthreads();
        double o1 = c1, o2 = c2;
        for (int j = 0; j < w; ++j) { o1 += sm[0][j]; o2 += sm[1][j]; }
        const double t1 = ((sm[0][0] + sm[0][1]) + sm[0][2]) + sm[0][3], t2 = ((sm[1][0] + sm[1][1]) + sm[1][2]) + sm[1][3];
        if (i < nb) { d2 e = {o1 + a1 - v.x, o2 + a2 - v.y}; p[i] = e; }
        c1 += t1; c2 += t2;
        __syncthreads();
    }
}

// pass 3: prefix[seg][i] = (sum_{j<i} x_j, sum_{j<i} x_j^2), i = 0..n
__global__ __launch_bounds__(256) void k_sp_prefix(const float* __restrict__ sorted, int64_t n, int nb, const d2* __restrict__ bofs,
                                                   d2* __restrict__ prefix) {
    __shared__ double sm[2][4];
    const int64_t seg = blockIdx.y, b = blockIdx.x;
    const float* x = sorted + seg * n;
    d2* out = prefix + seg * (n + 1);
    const int64_t i0 = b * PB + (int64_t)threadIdx.x * 4;
    double v1[4], v2[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const double v = (i0 + e < n) ? (double)x[i0 + e] : 0.0;
        v1[e] = v; v2[e] = v * v;
    }
    const double t1 = (v1[0] + v1[1]) + (v1[2] + v1[3]), t2 = (v2[0] + v2[1]) + (v2[2] + v2[3]);
    double a1 = t1, a2 = t2;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const double u1 = __shfl_up(a1, o), u2 = __shfl_up(a2, o);
        if (lane >= o) { a1 += u1; a2 += u2; }
    }
    if (lane == 63) { sm[0][w] = a1; sm[1][w] = a2; }
    __syncthreads();
    const d2 base = bofs[seg * nb + b];
    double o1 = base.x + (a1 - t1), o2 = base.y + (a2 - t2);
    for (int j = 0; j < w; ++j) { o1 += sm[0][j]; o2 += sm[1][j]; }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        if (i0 + e <= n) { d2 o = {o1, o2}; out[i0 + e] = o; }       // (i == n: the segment totals)
        o1 += v1[e]; o2 += v2[e];
    }
    if (i0 + 4 == n) { d2 o = {o1, o2}; out[n] = o; }                // n a multiple of 4: the totals sit past the last element
}

// One group of G = 2^bits threads per (segment, candidate); a workgroup holds 256 / G groups.
//   level(x) = rne(fl32(x / s));  q = clamp(level + z, 0, qmax);  value = fl32(fl32(q - z) * s)      (uniform.py:29-36)
// klo = ceil(-z), khi = floor(qmax - z): levels klo..khi are unclamped.  Thread t owns the run of level klo + t (its lower
// boundary B[t] = first sorted index with level >= min(klo + t, khi + 1)); thread 0 also owns the two clamped runs
// [0, B[0]) and [B[G], n).
template <int G>
__global__ __launch_bounds__(256) void k_score_sorted(const float* __restrict__ sorted, const d2* __restrict__ prefix, int64_t S,
                                                      int64_t n, const float* __restrict__ scale, const float* __restrict__ zp,
                                                      int P, float qmax, double norm, float* __restrict__ scores) {
    constexpr int GPB = 256 / G;
    __shared__ int64_t bnd[GPB][G + 1];
    __shared__ double red[256];
    const int gi = threadIdx.x / G, t = threadIdx.x % G;
    const int64_t pair = (int64_t)blockIdx.x * GPB + gi;               // = p * S + seg
    const bool live = pair < (int64_t)P * S;
    const int64_t seg = live ? pair % S : 0;
    const float s = live ? scale[pair] : 1.0f, z = live ? zp[pair] : 0.0f;
    const float klo = ceilf(-z), khi = floorf(qmax - z);
    const float* x = sorted + seg * n;
    const d2* pf = prefix + seg * (n + 1);
    auto lower = [&](float target) {                                   // first i with rne(x[i] / s) >= target
        int64_t lo = 0, hi = n;
        while (lo < hi) {
            const int64_t mid = (lo + hi) >> 1;
            if (rintf(x[mid] / s) >= target) hi = mid; else lo = mid + 1;
        }
        return lo;
    };
    if (live) {
        bnd[gi][t] = lower(fminf(klo + (float)t, khi + 1.0f));
        if (t == 0) bnd[gi][G] = lower(khi + 1.0f);
    }
    __syncthreads();
    double acc = 0.0;
    if (live) {
        auto run = [&](int64_t a, int64_t b, float level) {
            if (b <= a) return 0.0;
            const float q = fminf(fmaxf(level + z, 0.0f), qmax);
            const double c = (double)((q - z) * s);
            const d2 pa = pf[a], pb = pf[b];
            return (pb.y - pa.y) - 2.0 * c * (pb.x - pa.x) + (double)(b - a) * c * c;
        };
        acc = run(bnd[gi][t], bnd[gi][t + 1], klo + (float)t);
        if (t == 0) acc += run(0, bnd[gi][0], klo - 1.0f) + run(bnd[gi][G], n, khi + 1.0f);
    }
    red[threadIdx.x] = acc;
    __syncthreads();
#pragma unroll
    for (int o = G / 2; o > 0; o >>= 1) {                               // fixed order within the group
        if (t < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (live && t == 0) scores[pair] = (float)(-norm * red[threadIdx.x]);
}

size_t sort_temp_bytes(int64_t S, int64_t n) {
    const int64_t b = adalog_sort_workspace_bytes(S, n, 0);            // csrc/radix_sort.hip
    return (size_t)((b < 256 ? 256 : b) + 255) / 256 * 256;
}

}  // namespace

// Workspace of adalog_sorted_prefix_build: radix-sort temporaries + segment offsets + per-block sums.
extern "C" int64_t adalog_sorted_prefix_workspace_bytes(int64_t S, int64_t n) {
    if (S < 1 || n < 1 || S * n >= ((int64_t)1 << 31)) return -1;
    const int64_t nb = (n + PB - 1) / PB;
    return (int64_t)sort_temp_bytes(S, n) + ((S + 1) * 4 + 255) / 256 * 256 + S * nb * 16;
}

// x: [S][n] fp32 (contiguous segments) -> sorted [S][n] (ascending per segment) and prefix [S][n + 1][2] fp64
// (exclusive prefix sums of x and x^2 along the sorted order; entry n = the segment totals).
extern "C" int adalog_sorted_prefix_build(const float* x, int64_t S, int64_t n, float* sorted, double* prefix, void* workspace,
                                          int64_t workspace_bytes, void* stream) {
    ADALOG_ARG_CHECK(x && sorted && prefix && workspace && S >= 1 && n >= 1, "sorted_prefix_build: bad arguments");
    ADALOG_ARG_CHECK(S * n < ((int64_t)1 << 31), "sorted_prefix_build: more than 2^31 - 1 elements");
    ADALOG_ARG_CHECK(workspace_bytes >= adalog_sorted_prefix_workspace_bytes(S, n), "sorted_prefix_build: workspace too small");
    ADALOG_ARG_CHECK((((uintptr_t)workspace | (uintptr_t)prefix) & 15) == 0, "sorted_prefix_build: workspace / prefix must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    size_t tb = sort_temp_bytes(S, n);
    uint8_t* ws = (uint8_t*)workspace;
    int* offs = (int*)(ws + tb);
    d2* bsum = (d2*)(ws + tb + ((S + 1) * 4 + 255) / 256 * 256);
    ADALOG_ARG_CHECK(((uintptr_t)workspace & 255) == 0 || n <= 8192, "sorted_prefix_build: workspace must be 256-byte aligned");
    {
        const int rc = adalog_sort_f32(x, S, n, sorted, nullptr, ws, (int64_t)tb, stream);      // hand-written LSD radix sort, per segment
        if (rc) return rc;
    }
    const int nb = cdiv(n, PB);
    ADALOG_ARG_CHECK(S <= 65535, "sorted_prefix_build: more than 65535 segments");
    hipLaunchKernelGGL(k_sp_blocksum, dim3(nb, (unsigned)S), dim3(256), 0, st, sorted, n, nb, bsum);
    hipLaunchKernelGGL(k_sp_blockscan, dim3((unsigned)S), dim3(256), 0, st, bsum, nb);
    hipLaunchKernelGGL(k_sp_prefix, dim3(nb, (unsigned)S), dim3(256), 0, st, sorted, n, nb, bsum, (d2*)prefix);
    ADALOG_LAUNCH_CHECK("adalog_sorted_prefix_build");
    return 0;
}

// scores[p][seg] = -norm * sum over the segment of (x - fq_{p,seg}(x))^2 ; scale / zp: [P][S] (candidate-major).
// `tail` (may be null): the FPCS step's ranking + next grid / commit inside the same launch (fpcs_tail.h); its grid is (scale, zp).
extern "C" int adalog_score_self_sorted_tail(const float* sorted, const double* prefix, int64_t S, int64_t n, const float* scale,
                                             const float* zp, int P, int n_bits, double norm, float* scores,
                                             const adalog_fpcs_tail* tail, void* stream) {
    ADALOG_ARG_CHECK(sorted && prefix && scale && zp && scores && S >= 1 && n >= 1 && P >= 1, "score_self_sorted: bad arguments");
    ADALOG_ARG_CHECK(n_bits >= 1 && n_bits <= 8, "score_self_sorted: 1..8 bits");
    const char* why = fpcs::tail_problem(tail, P);
    ADALOG_ARG_CHECK(why == nullptr, why);
    const float qmax = (float)((1 << n_bits) - 1);
    const int G = 1 << n_bits;
    hipStream_t st = (hipStream_t)stream;
    adalog_note_kernel("k_score_sorted");
    const int64_t pairs = (int64_t)P * S;
    const int gpb = 256 / G;
    const int64_t blocks = (pairs + gpb - 1) / gpb;
    ADALOG_ARG_CHECK(blocks < ((int64_t)1 << 31), "score_self_sorted: grid too large");
#define LAUNCH_SS(GV) hipLaunchKernelGGL((k_score_sorted<GV>), dim3((unsigned)blocks), dim3(256), 0, st, sorted, (const d2*)prefix, S, n, scale, zp, P, qmax, norm, scores)
    switch (n_bits) {
        case 1: LAUNCH_SS(2); break;
        case 2: LAUNCH_SS(4); break;
        case 3: LAUNCH_SS(8); break;
        case 4: LAUNCH_SS(16); break;
        case 5: LAUNCH_SS(32); break;
        case 6: LAUNCH_SS(64); break;
        case 7: LAUNCH_SS(128); break;
        default: LAUNCH_SS(256); break;
    }
#undef LAUNCH_SS
    ADALOG_LAUNCH_CHECK("adalog_score_self_sorted");
    // the FPCS step's tail: a second launch.  Measured and not kept (round 6, same-box A/Bs): a ticket per segment with the last
    // arrival ranking it (126 us against 32 + 11: the last blocks serialise the columns' tails), and a workgroup of 1 024 threads per
    // segment holding all P x G bisections and ranking in LDS (60 us against 40 + 11 for the per-channel searches, 51 against 16 + 11
    // for the per-tensor ones, where one CU's address path serialises the divergent loads)
    if (tail) return adalog_topk_next_tail(scores, P, (int)S, tail, nullptr, stream);
    return 0;
}

extern "C" int adalog_score_self_sorted(const float* sorted, const double* prefix, int64_t S, int64_t n, const float* scale,
                                        const float* zp, int P, int n_bits, double norm, float* scores, void* stream) {
    return adalog_score_self_sorted_tail(sorted, prefix, S, n, scale, zp, P, n_bits, norm, scores, nullptr, stream);
}
