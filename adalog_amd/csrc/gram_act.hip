// K8 (Gram form) -- output-MSE ACTIVATION searches of the uniformly quantised Linear layers (per-tensor candidates) scored from
// the candidates' own Gram matrices instead of from the layer's outputs.
//
// Reference: quant_layers/linear.py:394-430 (_search_best_a_scale).  Candidate p = (s_p, z_p) quantises the whole activation,
// x_p = clamp(rne(x / s_p) + z_p, 0, 2^b - 1) - z_p, and is scored by  sum_{t,o} (r[t,o] - s_p sum_k Wq[o,k] x_p[t,k])^2  with
// r = raw_out - bias and Wq = diag(s_w) W_int the FIXED quantised weight (fixed for the six scoring calls of an activation_fpcs,
// linear.py:505-523).  Expanding the square,
//     S0 - 2 s_p <X_p, C> + s_p^2 <H, X_p^T X_p>,      C = r . Wq  [T,K],   H = Wq^T Wq  [K,K],   S0 = sum r^2,
// only G_p = X_p^T X_p depends on the candidate through a product -- K x K x T multiply-adds, symmetric (half of them needed),
// INSTEAD OF the O x K x T of W . X_p^T: a sixth of the matrix work for qkv (O = 3 K), an eighth for fc1 (O = 4 K), exact on the
// int8 MFMA (x_p are small integers).  The linear term needs no pass over the tensor at all: a per-tensor uniform quantiser is a
// monotone step function, so the elements of one level are ONE run of the activation sorted by value, and <X_p, C> =
// sum_levels v (P[end] - P[begin]) with P the prefix sums of C in that order (built once per activation_fpcs call; the sort once
// per captured tensor) -- 2^b + 2 bisections per candidate with the exact predicate rne(fl32(x / s)) >= v.
//
// Exactness (the three terms are each ~ |r|^2 and cancel to the score):
//   * G_p exact (int8 MFMA, int32 accumulators over <= 2^17 tokens per task), <H, G_p> in fp64 with H = sum_o s_w[o]^2 w w^T in fp64;
//   * C exact up to ONE rounding of r s_w to 30-bit fixed point per token row (four int8 limbs, int8 MFMA against W_int^T, int64
//     recombination), then fp64; prefix sums fp64 in a fixed order;  S0 fp64;
//   * out = s_p s_w[o] D here, s_p fl32(s_w[o] D) in the token-form kernel (gemm_k_slab.inc): 6e-8 relative per output, random.
// Against the token form and the fp32 ATen evaluation the scores agree to ~1e-6 (tests/test_gpu_kernels.py).
//
// Kernels (build, once per activation_fpcs call):
//   k_ga_fragorder   x [T,K] fp32 -> the score kernel's fragment order (once per captured tensor)
//   k_ga_pack_wt     W [O,K] fp32, (s_w, z_w) per row -> W_int^T int8 [K][Op]
//   k_ga_rmax/rfix   raw_out [T,O], bias, s_w -> per-token exponent, four int8 limb planes [4][T][Op], S0
//   k_i8mm           (gram_mm.inc: LDS-tiled int8 product, shared with gram.hip) C_fix = limbs . W_int
//   k_ga_fin_c       limbs recombined -> C fp64 [T][K]
//   k_ga_gather_sum / k_ga_scan / k_ga_prefix   prefix sums of C along the sorted order of x (three passes, fixed order)
//   k_ga_h           H = Wq^T Wq in fp64 (register-tiled over the fp64 weight image), written in the accumulator layout of the
//                    score kernels, role by role (off-diagonal blocks doubled)
// Per FPCS step:
//   k_ga_quad        one workgroup (4 waves x 512 registers: the whole upper triangle of G_p lives in the CU's registers) per
//                    (candidate, token range): generates the candidate's int8 operand fragments from x_t with the exact-bin
//                    arithmetic of the slab GEN form, 32 tokens per step, shares them through LDS, one MFMA per owned block;
//                    at the end contracts its accumulators with H
//   k_ga_finish      per candidate: run boundaries by bisection on the sorted activation -> linear term; score
#include "common.h"
#include "fpcs_tail.h"
#include <stdlib.h>
#include <type_traits>
#include <utility>

extern "C" int64_t adalog_sort_workspace_bytes(int64_t S, int64_t n, int with_perm);
extern "C" int adalog_sort_f32(const float* x, int64_t S, int64_t n, float* sorted, unsigned int* perm, void* workspace,
                               int64_t workspace_bytes, void* stream);
extern "C" int adalog_topk_next_tail(const float* scores, int P, int cols, const adalog_fpcs_tail* tail, int* idx_out, void* stream);

namespace {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

template <typename F, int... I> __device__ __forceinline__ void sfor_impl(F&& f, std::integer_sequence<int, I...>) {
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, typename F> __device__ __forceinline__ void sfor(F&& f) { sfor_impl(f, std::make_integer_sequence<int, N>{}); }

// The fast path of the score kernels' generator for two neighbouring elements (xa, xb) of a fragment: bin by reciprocal multiply, track
// the slot's largest distance from an integer (dm: the tie test; the pair's two distances go into one v_max3), clamp the QUOTIENT, and
// let ONE SDWA add of 1.5 * 2^23 round it to nearest-even, convert it and write the signed byte into position e0 / e0 + 1 of the packed
// dword pk (the low byte of the sum's bit pattern is rne(t) in two's complement): 5.5 VALU instructions per element.  Same-box A/B at
// the qkv / swin stage-1 shapes (us per step): element by element through a biased cvt_pk_u8 (add 128, med3, convert, one xor per
// dword: 7.25 per element) 183 / 391; the same in pairs with v_max3 184 / 397; the bias folded into the multiply (fma(x, 1/s, 128),
// 5.75) 174 / 370 -- but the sum's rounding at 128..256 costs 2^-17 of the tie zone, and with the zone widened for it the calibration
// was no faster (same box 868 / 870 against 878 / 873 ms); this form 175 / 372 with the original zone.
#define GA_GEN_PAIR(xa, xb, dm, pk, e0)                                                                                           \
    do {                                                                                                                          \
        const float ta_ = (xa) * ginv, ka_ = rintf(ta_), tb_ = (xb) * ginv, kb_ = rintf(tb_);                                     \
        dm = fmaxf(dm, fmaxf(fabsf(ta_ - ka_), fabsf(tb_ - kb_)));                                                                \
        pk = sdwa_rne_byte(pk, __builtin_amdgcn_fmed3f(ta_, glo - 128.0f, ghi - 128.0f), gmagic, (e0));                            \
        pk = sdwa_rne_byte(pk, __builtin_amdgcn_fmed3f(tb_, glo - 128.0f, ghi - 128.0f), gmagic, (e0) + 1);                        \
    } while (0)

constexpr int RLIMBS = 4;
constexpr int RFIX_BITS = 29;
constexpr int PBLK = 1024;             // elements per prefix block

// ------------------------------------------------------------------------------------------------ transposes / packs
// x [T][K] fp32 -> "fragment order"  xf[chunk c][k-block b][piece q][lane][4 floats]: lane (r = lane & 31, h = lane >> 5) of the
// score kernel generates row k = 32 b + r over the 16 tokens 32 c + 16 h .. of a chunk; its piece q holds tokens 32 c + 16 h + 4 q + 0..3.
// Every load instruction of the score kernel then reads 1 KiB of consecutive memory (a plain [K][T] transpose has each lane on a
// different cache line: the loads took longer to ISSUE than the MFMAs they feed).  Tokens past T are zero.
__global__ __launch_bounds__(256) void k_ga_fragorder(const float* __restrict__ x, int T, int K, int64_t ldx, float* __restrict__ xf, int nchunk) {
    __shared__ float tile[32][33];                          // [token][k] of one (chunk, k-block)
    const int c = blockIdx.x, b = blockIdx.y, NJ = K >> 5;
    const int tid = threadIdx.x;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int tk = tid + 256 * i, tl = tk >> 5, kl = tk & 31;                 // coalesced along k
        const int t = 32 * c + tl;
        tile[tl][kl] = t < T ? x[(int64_t)t * ldx + 32 * b + kl] : 0.0f;
    }
    __syncthreads();
    // thread = (q, lane): writes its 4 floats (16 bytes): consecutive threads -> consecutive 16-byte pieces
    const int q = tid >> 6, lane = tid & 63, r = lane & 31, h = lane >> 5;
    float4 v;
    v.x = tile[16 * h + 4 * q + 0][r]; v.y = tile[16 * h + 4 * q + 1][r]; v.z = tile[16 * h + 4 * q + 2][r]; v.w = tile[16 * h + 4 * q + 3][r];
    *reinterpret_cast<float4*>(xf + ((((int64_t)c * NJ + b) * 4 + q) * 64 + lane) * 4) = v;
}

// W_int^T [K][Op] int8: column o of row k = clamp(rne(W[o][k] / s_w[o]) + z_w[o], 0, qmax) - z_w[o]   (exact bins: uni_bin_fast);
// wsd [O][K] fp64 = s_w[o] * that value (the quantised weight Wq, exact)
__global__ __launch_bounds__(256) void k_ga_pack_wt(const float* __restrict__ W, int O, int K, int64_t ldw, const float* __restrict__ sw,
                                                    const float* __restrict__ zw, float qmax, int8_t* __restrict__ wt, int64_t Op,
                                                    double* __restrict__ wsd) {
#pragma clang fp contract(off)
    __shared__ int8_t tile[64][65];
    const int tid = threadIdx.x, tx = tid & 63, ty = tid >> 6;
    const int o0 = blockIdx.x * 64, k0 = blockIdx.y * 64;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int o = o0 + ty + 4 * i, k = k0 + tx;
        int q = 0;
        if (o < O && k < K) {
            const float s = sw[o], z = rintf(zw[o]);
            q = (int)(uni_bin_fast(W[(int64_t)o * ldw + k], s, 1.0f / s, z, qmax) - z);
            wsd[(int64_t)o * K + k] = (double)s * (double)q;           // Wq[o][k] exactly (24 + 8 bits): the operand of k_ga_h
        }
        tile[ty + 4 * i][tx] = (int8_t)q;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int k = k0 + ty + 4 * i, o = o0 + tx;
        if (k < K && o < Op) wt[(int64_t)k * Op + o] = tile[tx][ty + 4 * i];
    }
}

__device__ __forceinline__ double ga_block_sum(double v, double* sm) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = v;
    __syncthreads();
    return (sm[0] + sm[1]) + (sm[2] + sm[3]);
}

// one workgroup per token row t: r'' = fl32(fl32(raw_out - bias) * s_w[o]); e_t with max|r''| * 2^e_t in [2^29, 2^30); four balanced
// int8 limbs of rne(r'' * 2^e_t) into planes [l][t][Op]; s0p[t] = sum_o (raw_out - bias)^2 (the un-scaled reference: S0)
__global__ __launch_bounds__(256) void k_ga_rfix(const float* __restrict__ ref, int T, int O, int64_t Op, const float* __restrict__ bias,
                                                 const float* __restrict__ sw, int8_t* __restrict__ rl, double* __restrict__ s0p,
                                                 double* __restrict__ cscl) {
#pragma clang fp contract(off)
    __shared__ double smd[4];
    __shared__ float smf[4];
    const int t = blockIdx.x, tid = threadIdx.x;
    const float* r = ref + (int64_t)t * O;
    float am = 0.0f;
    for (int o = tid; o < O; o += 256) am = fmaxf(am, fabsf((r[o] - (bias ? bias[o] : 0.0f)) * sw[o]));
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) am = fmaxf(am, __shfl_xor(am, d));
    if ((tid & 63) == 0) smf[tid >> 6] = am;
    __syncthreads();
    am = fmaxf(fmaxf(smf[0], smf[1]), fmaxf(smf[2], smf[3]));
    int e = 0;
    if (am > 0.0f && am < 3.0e38f) e = RFIX_BITS - ilogbf(am);
    const double sc = ldexp(1.0, e);
    double acc = 0.0;
    int8_t* l0 = rl + (int64_t)t * Op;
    const int64_t plane = (int64_t)T * Op;
    for (int o = tid; o < Op; o += 256) {
        int v = 0;
        if (o < O) {
            const float rb = r[o] - (bias ? bias[o] : 0.0f);
            acc += (double)rb * (double)rb;
            v = (int)rint((double)(rb * sw[o]) * sc);
        }
        int rest = v;
#pragma unroll
        for (int l = 0; l < RLIMBS; ++l) {
            const int d = (int)(int8_t)(rest & 0xff);
            l0[l * plane + o] = (int8_t)d;
            rest = (rest - d) >> 8;
        }
    }
    const double tot = ga_block_sum(acc, smd);
    if (tid == 0) { s0p[t] = tot; cscl[t] = ldexp(1.0, -e); }
}

#include "gram_mm.inc"          // k_i8mm: the LDS-tiled int8 product shared with gram.hip

// C[t][k] = (sum_l part[l T + t][k] << 8 l) * 2^-e_t   (fp64; |C_fix| < 2^53 for O < 2^15: 2^29 * 2^6..7 * O)
__global__ __launch_bounds__(256) void k_ga_fin_c(const int* __restrict__ part, int T, int K, const double* __restrict__ cscl, double* __restrict__ C) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (int64_t)T * K) return;
    const int t = (int)(idx / K);
    long long c = 0;
#pragma unroll
    for (int l = 0; l < RLIMBS; ++l) c += ((long long)part[(int64_t)l * T * K + idx]) << (8 * l);
    C[idx] = (double)c * cscl[t];
}

// ------------------------------------------------------------------------------------------------ prefix sums of C in sorted order

// Cs[i] = C[perm[i]] (the one random pass over C) and the block sums of Cs
__global__ __launch_bounds__(256) void k_ga_gather_sum(const double* __restrict__ C, const unsigned int* __restrict__ perm, int64_t n,
                                                       double* __restrict__ Cs, double* __restrict__ bsum) {
    __shared__ double smd[4];
    const int64_t i0 = (int64_t)blockIdx.x * PBLK + (int64_t)threadIdx.x * 4;
    double a = 0.0;
#pragma unroll
    for (int e = 0; e < 4; ++e)
        if (i0 + e < n) { const double v = C[perm[i0 + e]]; Cs[i0 + e] = v; a += v; }
    const double tot = ga_block_sum(a, smd);
    if (threadIdx.x == 0) bsum[blockIdx.x] = tot;
}

// exclusive scan of the block sums, in place (one workgroup, chunks of 256 with a carry)
// (workgroup 1 of the same launch: S0 = sum_t s0p[t] in a fixed order)
__global__ __launch_bounds__(256) void k_ga_scan(double* __restrict__ bsum, int nb, const double* __restrict__ s0p, int T, double* __restrict__ s0) {
    __shared__ double sm[4];
    if (blockIdx.x == 1) {
        double a = 0.0;
        for (int t = threadIdx.x; t < T; t += 256) a += s0p[t];
        const double tot = ga_block_sum(a, sm);
        if (threadIdx.x == 0) s0[0] = tot;
        return;
    }
    double carry = 0.0;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    double nxt = (int)threadIdx.x < nb ? bsum[threadIdx.x] : 0.0;
    for (int base = 0; base < nb; base += 256) {
        const int i = base + threadIdx.x;
        const double v = nxt;
        nxt = i + 256 < nb ? bsum[i + 256] : 0.0;               // the next chunk's load is in flight under this chunk's scan
        double a = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const double u = __shfl_up(a, o);
            if (lane >= o) a += u;
        }
        if (lane == 63) sm[w] = a;
        __syncthreads();
        double off = carry;
        for (int j = 0; j < w; ++j) off += sm[j];
        const double tot = ((sm[0] + sm[1]) + sm[2]) + sm[3];
        if (i < nb) bsum[i] = off + a - v;
        carry += tot;
        __syncthreads();
    }
}

// prefix[i] = sum_{j < i} C[perm[j]], i = 0..n
__global__ __launch_bounds__(256) void k_ga_prefix(const double* __restrict__ Cs, int64_t n, const double* __restrict__ bofs,
                                                   double* __restrict__ prefix) {
    __shared__ double sm[4];
    const int64_t i0 = (int64_t)blockIdx.x * PBLK + (int64_t)threadIdx.x * 4;
    double v[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = (i0 + e < n) ? Cs[i0 + e] : 0.0;
    const double t = (v[0] + v[1]) + (v[2] + v[3]);
    double a = t;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const double u = __shfl_up(a, o);
        if (lane >= o) a += u;
    }
    if (lane == 63) sm[w] = a;
    __syncthreads();
    double o1 = bofs[blockIdx.x] + (a - t);
    for (int j = 0; j < w; ++j) o1 += sm[j];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        if (i0 + e <= n) prefix[i0 + e] = o1;
        o1 += v[e];
    }
    if (i0 + 4 == n) prefix[n] = o1;
}

// ------------------------------------------------------------------------------------------------ H in the accumulator layout
// upper-triangle blocks (i <= j) in row-major order: n(i, j) = i NJ - i (i - 1) / 2 + (j - i)
__host__ __device__ constexpr int tri_index(int i, int j, int NJ) { return i * NJ - (i * (i - 1)) / 2 + (j - i); }

// hfrag[n][lane][e] = (i == j ? 1 : 2) * H[32 i + 8 (e >> 2) + 4 (lane >> 5) + (e & 3)][32 j + (lane & 31)],  H = Wq^T Wq = sum_o Wq[o][k1] Wq[o][k2]
// for the blocks of one ROLE of the score kernels: tri != 0: the upper triangle of the nj k-blocks from i0 (row-major, i <= j);
// else the ni x nj rectangle of rows i0 .. against columns j0 .. (row-major).
// One workgroup of eight waves per 32 x 32 block: a wave takes an eighth of the output channels, a lane a 4 x 4 patch of the block
// (16 fp64 FMAs per two 32-byte loads of the fp64 weight image; the round's first form looped one thread per entry over all of O with
// int8 loads and a conversion per product: 40 us at the qkv shape, 4 x 65 us at vit_base's); the eight partial blocks are summed in a
// fixed order through LDS.
constexpr int GA_H_WAVES = 8;
__global__ __launch_bounds__(64 * GA_H_WAVES) void k_ga_h(const double* __restrict__ wsd, int O, int K, double* __restrict__ hfrag, int i0, int j0,
                                                         int ni, int nj, int tri) {
    __shared__ double red[GA_H_WAVES - 1][64][17];
    const int n = blockIdx.x, lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    int i, j;
    if (tri) {
        int li = 0, rem = n;
        while (rem >= nj - li) { rem -= nj - li; ++li; }
        i = i0 + li; j = i0 + li + rem;
    } else {
        i = i0 + n / nj; j = j0 + n % nj;
    }
    const int a = lane >> 3, b = lane & 7;                  // rows 4 a .. 4 a + 3 of the block, columns 4 b .. 4 b + 3
    const int per = (O + GA_H_WAVES - 1) / GA_H_WAVES;
    const int o0 = w * per, o1 = min(O, o0 + per);
    const double* pa = wsd + 32 * i + 4 * a;
    const double* pb = wsd + 32 * j + 4 * b;
    double acc[4][4];
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int v = 0; v < 4; ++v) acc[u][v] = 0.0;
#pragma unroll 4
    for (int o = o0; o < o1; ++o) {
        const double2 a0 = *reinterpret_cast<const double2*>(pa + (int64_t)o * K), a1 = *reinterpret_cast<const double2*>(pa + (int64_t)o * K + 2);
        const double2 b0 = *reinterpret_cast<const double2*>(pb + (int64_t)o * K), b1 = *reinterpret_cast<const double2*>(pb + (int64_t)o * K + 2);
        const double av[4] = {a0.x, a0.y, a1.x, a1.y}, bv[4] = {b0.x, b0.y, b1.x, b1.y};
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int v = 0; v < 4; ++v) acc[u][v] = fma(av[u], bv[v], acc[u][v]);
    }
    if (w > 0) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int v = 0; v < 4; ++v) red[w - 1][lane][4 * u + v] = acc[u][v];
    }
    __syncthreads();
    if (w == 0) {
        const double f = i == j ? 1.0 : 2.0;
        double* out = hfrag + (int64_t)n * 1024;
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                double h = acc[u][v];
#pragma unroll
                for (int x = 0; x < GA_H_WAVES - 1; ++x) h += red[x][lane][4 * u + v];
                // block row rr = 4 a + u = 8 (e >> 2) + 4 (lane' >> 5) + (e & 3), column 4 b + v = lane' & 31
                const int lane2 = (a & 1) * 32 + 4 * b + v, e = 4 * (a >> 1) + u;
                out[lane2 * 16 + e] = f * h;
            }
    }
}

// ------------------------------------------------------------------------------------------------ the per-step kernel
struct GaQuadArgs {
    const float* xt; int64_t Tp; int K;            // x in fragment order [chunk][k-block][piece][lane][4] (k_ga_fragorder)
    const float* scale; const float* zp; int P;    // candidates [P]
    const double* hfrag;                           // this launch's first role: [tiles of a role][64][16], roles back to back
    double* qpart;                                 // [P][QS]: this launch writes columns q0 + role * S + split
    int S, chunks_per_split, nchunk;               // 32-token chunks
    int NJT;                                       // k-blocks of the whole tensor (K / 32): the chunk stride of xt
    int R;                                         // roles of this launch (workgroups per (candidate, split)): parts of the K x K triangle
    int QS, q0;                                    // partial sums per candidate over all launches / this launch's first column
    int kb_a, kb_b;                                // triangle role r: k-blocks kb_a + r NJ ..;  rectangle role r: rows kb_a .., columns kb_b + r NJC ..
    float qmax, tie;
    long long* timeline;                           // lab only: per workgroup [4 waves][8] cycle sums (mfma phase, generation, barrier, total)
};

// upper-triangle block n (row-major, i <= j) -> (i, j), at compile time
template <int NJ> __host__ __device__ constexpr int tri_row(int n) { int i = 0, rem = n; while (rem >= NJ - i) { rem -= NJ - i; ++i; } return i; }
template <int NJ> __host__ __device__ constexpr int tri_col(int n) { const int i = tri_row<NJ>(n); return n - tri_index(i, i, NJ) + i; }

template <int NJ, int W>
__device__ __forceinline__ void ga_quad_wave(const GaQuadArgs& p, uint8_t* lds, int lane, int cand, int split, int role) {
    constexpr int NBT = NJ * (NJ + 1) / 2;                 // upper-triangle blocks
    const int kb0 = p.kb_a + role * NJ;                    // this role's first k-block
    // this wave's blocks: a CONTIGUOUS range of the row-major list (runs of one row i: A fragment i stays in registers, the B
    // fragments j stream through two registers' worth -- all NJ fragments resident cost 48 registers the accumulators need)
    constexpr int N0 = (NBT * W) / 4, N1 = (NBT * (W + 1)) / 4;
    constexpr int NOWN = N1 - N0;
    constexpr int NGEN = (NJ - W + 3) / 4;                 // k-blocks W, W + 4, ... this wave generates
    const float gs = p.scale[cand], gz = rintf(p.zp[cand]);
    const float ginv = __builtin_amdgcn_rcpf(gs);
    const float glo = 128.0f - gz, ghi = 128.0f + (p.qmax - gz);
    const float gmagic = 12582912.0f;                      // 1.5 * 2^23 (GA_GEN_PAIR)
    const int c0 = split * p.chunks_per_split, c1 = min(c0 + p.chunks_per_split, p.nchunk);

    v16i acc[NOWN > 0 ? NOWN : 1];
#pragma unroll
    for (int b = 0; b < NOWN; ++b)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[b][e] = 0;

    // this lane's 16 tokens of row (32 b + r) of k-block b, chunk c: x_t[32 b + r][32 c + 16 h ..]
    float4 xr[NGEN > 0 ? NGEN : 1][4];
    auto load_x = [&](int c) {
        const int cc = min(c, p.nchunk - 1);
        sfor<NGEN>([&](auto gc) {
            constexpr int g = decltype(gc)::value;
            const float* src = p.xt + (((int64_t)cc * p.NJT + (kb0 + W + 4 * g)) * 4) * 256 + lane * 4;
#pragma unroll
            for (int j = 0; j < 4; ++j) xr[g][j] = *reinterpret_cast<const float4*>(src + j * 256);
        });
    };
    // one fragment's share of load_x: issued inside the interleaved block as soon as the generator has consumed the fragment's
    // registers instead of 12 vector loads back to back in front of the barrier (same box: K = 384 176 -> 175 us, K = 512 384 -> 373,
    // K = 768 633 -> 628 us per step)
    auto load_x_one = [&](int c, auto gc) {
        constexpr int g = decltype(gc)::value;
        const int cc = min(c, p.nchunk - 1);
        const float* src = p.xt + (((int64_t)cc * p.NJT + (kb0 + W + 4 * g)) * 4) * 256 + lane * 4;
#pragma unroll
        for (int j = 0; j < 4; ++j) xr[g][j] = *reinterpret_cast<const float4*>(src + j * 256);
    };
    // fast path of the generator for fragment g: bins by reciprocal multiply; returns the fragment and the slot's largest distance
    // from an integer (the tie test of the slab GEN form: above p.tie the IEEE quotient decides -- done afterwards, off the hot block)
    auto gen_fast = [&](auto gc, v4i& pk, float& dm) {
#pragma clang fp contract(off)
        constexpr int g = decltype(gc)::value;
        float xv[16];
#pragma unroll
        for (int j = 0; j < 4; ++j) { xv[4 * j] = xr[g][j].x; xv[4 * j + 1] = xr[g][j].y; xv[4 * j + 2] = xr[g][j].z; xv[4 * j + 3] = xr[g][j].w; }
        float kq[16];
        dm = 0.0f;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const float t = xv[e] * ginv;
            kq[e] = rintf(t);
            dm = fmaxf(dm, fabsf(t - kq[e]));
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            unsigned u = 0;
#pragma unroll
            for (int e = 0; e < 4; ++e)
                u = __builtin_amdgcn_cvt_pk_u8_f32(__builtin_amdgcn_fmed3f(kq[4 * j + e] + 128.0f, glo, ghi), e, u);
            pk[j] = (int)(u ^ 0x80808080u);
        }
    };
    auto gen_exact = [&](auto gc, v4i& pk) {                // the same with the IEEE quotient (a near-tie somewhere in the slot)
#pragma clang fp contract(off)
        constexpr int g = decltype(gc)::value;
        float xv[16];
#pragma unroll
        for (int j = 0; j < 4; ++j) { xv[4 * j] = xr[g][j].x; xv[4 * j + 1] = xr[g][j].y; xv[4 * j + 2] = xr[g][j].z; xv[4 * j + 3] = xr[g][j].w; }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            unsigned u = 0;
#pragma unroll
            for (int e = 0; e < 4; ++e)
                u = __builtin_amdgcn_cvt_pk_u8_f32(__builtin_amdgcn_fmed3f(rintf(xv[4 * j + e] / gs) + 128.0f, glo, ghi), e, u);
            pk[j] = (int)(u ^ 0x80808080u);
        }
    };
    // ... for a fragment of chunk c whose registers already hold the chunk after it: the (rare) slot re-reads its x
    auto gen_exact_at = [&](int c, auto gc, v4i& pk) {
        float4 keep[4];
        constexpr int g = decltype(gc)::value;
#pragma unroll
        for (int j = 0; j < 4; ++j) keep[j] = xr[g][j];
        load_x_one(c, gc);
        gen_exact(gc, pk);
#pragma unroll
        for (int j = 0; j < 4; ++j) xr[g][j] = keep[j];
    };
    auto frag_addr = [&](int buf, int g) { return lds + (buf * NJ + (W + 4 * g)) * 1024 + lane * 16; };

    if (c0 < c1) {
        load_x(c0);
        sfor<NGEN>([&](auto gc) {
            constexpr int g = decltype(gc)::value;
            v4i pk; float dm;
            gen_fast(gc, pk, dm);
            if (dm > p.tie) gen_exact(gc, pk);
            *reinterpret_cast<v4i*>(frag_addr(c0 & 1, g)) = pk;
        });
        load_x(c0 + 1);
    }
    __syncthreads();
    long long tl0 = 0, tl1 = 0, tl2 = 0, tlA = 0;
#define GA_T() ((long long)__builtin_readcyclecounter())
    if (p.timeline) tlA = GA_T();
    for (int c = c0; c < c1; ++c) {
        long long ta = 0;
        if (p.timeline) ta = GA_T();
        const uint8_t* fb = lds + ((c & 1) * NJ) * 1024 + lane * 16;
        // all NJ fragments in one LDS round trip (read just in time, every MFMA waited out its own: ~2 k cycles per chunk); the
        // VGPRs are there now that sixteen accumulator tiles sit in AGPRs
        v4i fr[NJ];
        sfor<NJ>([&](auto jc) { fr[decltype(jc)::value] = *reinterpret_cast<const v4i*>(fb + decltype(jc)::value * 1024); });
        // ONE basic block: this chunk's MFMAs and the NEXT chunk's fragments (its x is in registers since the last iteration; past
        // the last chunk the clamped re-read generates a fragment nobody uses).  The generator's VALU work is ~16 instructions per
        // MFMA: issued between the MFMAs it runs while the matrix pipe works (sched_group_barrier pins the interleave; back to back
        // the two phases took 1.4 k + 1.85 k cycles per chunk).
        v4i pkn[NGEN > 0 ? NGEN : 1];
        float dmn[NGEN > 0 ? NGEN : 1];
#pragma unroll
        for (int g = 0; g < NGEN; ++g) { pkn[g] = v4i{0, 0, 0, 0}; dmn[g] = 0.0f; }
        constexpr int NEL = NGEN * 16;                          // generated elements of this wave per chunk
        sfor<NOWN>([&](auto bc) {
            constexpr int b = decltype(bc)::value;
            constexpr int n = N0 + b;
            constexpr int i = tri_row<NJ>(n), j = tri_col<NJ>(n);
            const v4i fi = fr[i];
            const v4i fj = fr[j];
            // 20 accumulator tiles are 320 registers, the AGPR file holds 16.  The compiler picks ONE form of the MFMA per function
            // (accumulators in AGPRs) and copied the other tiles in and out of VGPRs around every MFMA (3 600 v_accvgpr moves in the
            // listing): tiles 16.. are issued in the VGPR-accumulator form by hand.  (Their next use is 20 MFMAs or a barrier away:
            // no MFMA -> MFMA / MFMA -> VALU hazard window is open.)
            // Every MFMA is a volatile asm statement: volatile statements keep their order, and the element slices between them are
            // tied to that order by the empty asm below (as builtins the MFMAs were clustered and all generator arithmetic sunk
            // behind them, sched_barrier or not).
            if constexpr (b < 16) asm volatile("v_mfma_i32_32x32x32_i8 %0, %1, %2, %0" : "+a"(acc[b]) : "v"(fi), "v"(fj));
            else asm volatile("v_mfma_i32_32x32x32_i8 %0, %1, %2, %0" : "+v"(acc[b]) : "v"(fi), "v"(fj));
            // ... and behind each MFMA its share of the generator's element pairs (fast path, GA_GEN_PAIR), pinned there.  (Tried and
            // dropped, same box: packed fp32 operations -- 196 against 182 us: packed fp32 issues badly beside MFMAs, as the guide's
            // filler table says.)
            {
#pragma clang fp contract(off)
                constexpr int NPR = NEL / 2;                    // the generator works on PAIRS of elements (GA_GEN_PAIR)
                constexpr int lo = (b * NPR) / NOWN, hi = ((b + 1) * NPR) / NOWN;
                sfor<hi - lo>([&](auto kc) {
                    constexpr int pi = lo + decltype(kc)::value, g = pi >> 3, e = (pi & 7) * 2;
                    const float4 q4 = xr[g][e >> 2];
                    const float xa = (e & 2) ? q4.z : q4.x, xb = (e & 2) ? q4.w : q4.y;
                    GA_GEN_PAIR(xa, xb, dmn[g], pkn[g][e >> 2], e & 3);
                    if constexpr (decltype(kc)::value == hi - lo - 1) asm volatile("" : "+v"(dmn[g]), "+v"(pkn[g]));
                    if constexpr ((pi & 7) == 7) load_x_one(c + 2, std::integral_constant<int, g>{});    // fragment g consumed
                });
            }
            __builtin_amdgcn_sched_barrier(0);
        });
        if constexpr (NOWN == 0 && NGEN > 0) {             // (tiny K: a wave that generates but owns no block -- nothing to hide behind)
#pragma clang fp contract(off)
            sfor<NEL / 2>([&](auto kc) {
                constexpr int pi = decltype(kc)::value, g = pi >> 3, e = (pi & 7) * 2;
                const float4 q4 = xr[g][e >> 2];
                const float xa = (e & 2) ? q4.z : q4.x, xb = (e & 2) ? q4.w : q4.y;
                GA_GEN_PAIR(xa, xb, dmn[g], pkn[g][e >> 2], e & 3);
            });
            load_x(c + 2);
        }
        long long tb = 0, tc = 0;
        if (p.timeline) { asm volatile("s_nop 0" ::: "memory"); tb = GA_T(); }
        sfor<NGEN>([&](auto gc) {
            constexpr int g = decltype(gc)::value;
            if (__builtin_expect(dmn[g] > p.tie, 0)) gen_exact_at(c + 1, gc, pkn[g]);
            *reinterpret_cast<v4i*>(frag_addr((c + 1) & 1, g)) = pkn[g];
        });
        if (p.timeline) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); tc = GA_T(); }
        __syncthreads();
        if (p.timeline) { const long long td = GA_T(); tl0 += tb - ta; tl1 += tc - tb; tl2 += td - tc; }
    }
    if (p.timeline && lane == 0) {
        long long* o = p.timeline + ((int64_t)blockIdx.x * 4 + W) * 8;
        o[0] = tl0; o[1] = tl1; o[2] = tl2; o[3] = GA_T() - tlA; o[4] = c1 - c0;
    }
#undef GA_T

    // ---- <H, G_p> over this wave's blocks
    double q = 0.0;
    sfor<NOWN>([&](auto bc) {
        constexpr int b = decltype(bc)::value;
        constexpr int n = N0 + b;
        const double* hp = p.hfrag + (((int64_t)role * NBT + n) * 64 + lane) * 16;
#pragma unroll
        for (int e = 0; e < 16; ++e) q += hp[e] * (double)acc[b][e];
    });
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o);
    double* red = reinterpret_cast<double*>(lds);
    __syncthreads();
    if (lane == 0) red[W] = q;
    __syncthreads();
    if (W == 0 && lane == 0) p.qpart[(int64_t)cand * p.QS + p.q0 + role * p.S + split] = (red[0] + red[1]) + (red[2] + red[3]);
}

template <int NJ>
__global__ __launch_bounds__(256, 1) void k_ga_quad(GaQuadArgs p) {
#if defined(__HIP_DEVICE_COMPILE__)
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];      // [2][NJ][64][16]
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int cand = blockIdx.x / (p.S * p.R), rem = blockIdx.x % (p.S * p.R), role = rem / p.S, split = rem % p.S;
    if (w == 0) ga_quad_wave<NJ, 0>(p, lds, lane, cand, split, role);
    else if (w == 1) ga_quad_wave<NJ, 1>(p, lds, lane, cand, split, role);
    else if (w == 2) ga_quad_wave<NJ, 2>(p, lds, lane, cand, split, role);
    else ga_quad_wave<NJ, 3>(p, lds, lane, cand, split, role);
#endif
}

// ------------------------------------------------------------------------------------------------ K > 384: the triangle in parts
// The upper triangle of K = 512 (136 blocks) / 768 (300 blocks) does not fit one CU's registers.  The k-blocks are cut in two halves
// A | B:  the triangles A x A and B x B are two ROLES of k_ga_quad<NJ / 2> (each generates only its own half of the fragments), and the
// square A x B is scored by k_ga_rect: one role (8 x 8 blocks, K = 512) or two (12 x 6 each, K = 768); a wave owns NIW rows x NJC columns
// of blocks and reads NIW + NJC fragments per chunk instead of all of them.  Generated fragments per candidate: 2 x K / 32 (K = 512),
// 2.5 x K / 32 (K = 768) instead of K / 32 -- the generator is what bounds these kernels -- against O / 32 MFMAs per fragment of the
// token form: the split forms pay for O >= 2 K (qkv, fc1).
template <int NIW, int NJC, int W>
__device__ __forceinline__ void ga_rect_wave(const GaQuadArgs& p, uint8_t* lds, int lane, int cand, int split, int role) {
    constexpr int NF = 4 * NIW + NJC;                      // fragments per chunk in LDS: slots 0 .. 4 NIW - 1 the rows, then the columns
    constexpr int NOWN = NIW * NJC;                        // this wave's blocks: rows NIW W .., all NJC columns
    constexpr int NGEN = (NF - W + 3) / 4;                 // slots W, W + 4, ... this wave generates
    constexpr int NRD = NIW + NJC;
    static_assert(NOWN <= 20 && NGEN >= 1, "accumulator budget");
    const int kbA = p.kb_a, kbB = p.kb_b + role * NJC;
    const float gs = p.scale[cand], gz = rintf(p.zp[cand]);
    const float ginv = __builtin_amdgcn_rcpf(gs);
    const float glo = 128.0f - gz, ghi = 128.0f + (p.qmax - gz);
    const float gmagic = 12582912.0f;                      // 1.5 * 2^23 (GA_GEN_PAIR)
    const int c0 = split * p.chunks_per_split, c1 = min(c0 + p.chunks_per_split, p.nchunk);

    v16i acc[NOWN];
#pragma unroll
    for (int b = 0; b < NOWN; ++b)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[b][e] = 0;

    float4 xr[NGEN][4];
    auto load_x = [&](int c) {
        const int cc = min(c, p.nchunk - 1);
        sfor<NGEN>([&](auto gc) {
            constexpr int g = decltype(gc)::value, f = W + 4 * g;
            const int kb = f < 4 * NIW ? kbA + f : kbB + (f - 4 * NIW);
            const float* src = p.xt + (((int64_t)cc * p.NJT + kb) * 4) * 256 + lane * 4;
#pragma unroll
            for (int j = 0; j < 4; ++j) xr[g][j] = *reinterpret_cast<const float4*>(src + j * 256);
        });
    };
    auto load_x_one = [&](int c, auto gc) {
        constexpr int g = decltype(gc)::value, f = W + 4 * g;
        const int cc = min(c, p.nchunk - 1);
        const int kb = f < 4 * NIW ? kbA + f : kbB + (f - 4 * NIW);
        const float* src = p.xt + (((int64_t)cc * p.NJT + kb) * 4) * 256 + lane * 4;
#pragma unroll
        for (int j = 0; j < 4; ++j) xr[g][j] = *reinterpret_cast<const float4*>(src + j * 256);
    };
    auto gen_fast = [&](auto gc, v4i& pk, float& dm) {
#pragma clang fp contract(off)
        constexpr int g = decltype(gc)::value;
        float xv[16];
#pragma unroll
        for (int j = 0; j < 4; ++j) { xv[4 * j] = xr[g][j].x; xv[4 * j + 1] = xr[g][j].y; xv[4 * j + 2] = xr[g][j].z; xv[4 * j + 3] = xr[g][j].w; }
        float kq[16];
        dm = 0.0f;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const float t = xv[e] * ginv;
            kq[e] = rintf(t);
            dm = fmaxf(dm, fabsf(t - kq[e]));
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            unsigned u = 0;
#pragma unroll
            for (int e = 0; e < 4; ++e)
                u = __builtin_amdgcn_cvt_pk_u8_f32(__builtin_amdgcn_fmed3f(kq[4 * j + e] + 128.0f, glo, ghi), e, u);
            pk[j] = (int)(u ^ 0x80808080u);
        }
    };
    auto gen_exact = [&](auto gc, v4i& pk) {
#pragma clang fp contract(off)
        constexpr int g = decltype(gc)::value;
        float xv[16];
#pragma unroll
        for (int j = 0; j < 4; ++j) { xv[4 * j] = xr[g][j].x; xv[4 * j + 1] = xr[g][j].y; xv[4 * j + 2] = xr[g][j].z; xv[4 * j + 3] = xr[g][j].w; }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            unsigned u = 0;
#pragma unroll
            for (int e = 0; e < 4; ++e)
                u = __builtin_amdgcn_cvt_pk_u8_f32(__builtin_amdgcn_fmed3f(rintf(xv[4 * j + e] / gs) + 128.0f, glo, ghi), e, u);
            pk[j] = (int)(u ^ 0x80808080u);
        }
    };
    auto gen_exact_at = [&](int c, auto gc, v4i& pk) {     // (the fragment's registers already hold the chunk after c: re-read its x)
        float4 keep[4];
        constexpr int g = decltype(gc)::value;
#pragma unroll
        for (int j = 0; j < 4; ++j) keep[j] = xr[g][j];
        load_x_one(c, gc);
        gen_exact(gc, pk);
#pragma unroll
        for (int j = 0; j < 4; ++j) xr[g][j] = keep[j];
    };
    auto frag_addr = [&](int buf, int g) { return lds + (buf * NF + (W + 4 * g)) * 1024 + lane * 16; };

    if (c0 < c1) {
        load_x(c0);
        sfor<NGEN>([&](auto gc) {
            constexpr int g = decltype(gc)::value;
            v4i pk; float dm;
            gen_fast(gc, pk, dm);
            if (dm > p.tie) gen_exact(gc, pk);
            *reinterpret_cast<v4i*>(frag_addr(c0 & 1, g)) = pk;
        });
        load_x(c0 + 1);
    }
    __syncthreads();
    for (int c = c0; c < c1; ++c) {
        const uint8_t* fb = lds + ((c & 1) * NF) * 1024 + lane * 16;
        v4i fr[NRD];                                        // the wave's NIW row fragments, then the NJC column fragments
        sfor<NRD>([&](auto kc) {
            constexpr int k = decltype(kc)::value, slot = k < NIW ? NIW * W + k : 4 * NIW + (k - NIW);
            fr[k] = *reinterpret_cast<const v4i*>(fb + slot * 1024);
        });
        v4i pkn[NGEN];
        float dmn[NGEN];
#pragma unroll
        for (int g = 0; g < NGEN; ++g) { pkn[g] = v4i{0, 0, 0, 0}; dmn[g] = 0.0f; }
        constexpr int NEL = NGEN * 16;
        sfor<NOWN>([&](auto bc) {                            // (the interleave of k_ga_quad: an MFMA, then its share of the generator)
            constexpr int b = decltype(bc)::value;
            const v4i fi = fr[b / NJC];
            const v4i fj = fr[NIW + b % NJC];
            if constexpr (b < 16) asm volatile("v_mfma_i32_32x32x32_i8 %0, %1, %2, %0" : "+a"(acc[b]) : "v"(fi), "v"(fj));
            else asm volatile("v_mfma_i32_32x32x32_i8 %0, %1, %2, %0" : "+v"(acc[b]) : "v"(fi), "v"(fj));
            {
#pragma clang fp contract(off)
                constexpr int NPR = NEL / 2;                    // the generator works on PAIRS of elements (GA_GEN_PAIR)
                constexpr int lo = (b * NPR) / NOWN, hi = ((b + 1) * NPR) / NOWN;
                sfor<hi - lo>([&](auto kc) {
                    constexpr int pi = lo + decltype(kc)::value, g = pi >> 3, e = (pi & 7) * 2;
                    const float4 q4 = xr[g][e >> 2];
                    const float xa = (e & 2) ? q4.z : q4.x, xb = (e & 2) ? q4.w : q4.y;
                    GA_GEN_PAIR(xa, xb, dmn[g], pkn[g][e >> 2], e & 3);
                    if constexpr (decltype(kc)::value == hi - lo - 1) asm volatile("" : "+v"(dmn[g]), "+v"(pkn[g]));
                    if constexpr ((pi & 7) == 7) load_x_one(c + 2, std::integral_constant<int, g>{});    // fragment g consumed
                });
            }
            __builtin_amdgcn_sched_barrier(0);
        });
        sfor<NGEN>([&](auto gc) {
            constexpr int g = decltype(gc)::value;
            if (__builtin_expect(dmn[g] > p.tie, 0)) gen_exact_at(c + 1, gc, pkn[g]);
            *reinterpret_cast<v4i*>(frag_addr((c + 1) & 1, g)) = pkn[g];
        });
        __syncthreads();
    }

    // ---- <H, G_p> over this wave's blocks (row-major over the role's 4 NIW x NJC blocks; every one off the diagonal: H doubled)
    double q = 0.0;
    sfor<NOWN>([&](auto bc) {
        constexpr int b = decltype(bc)::value;
        constexpr int n = (NIW * W + b / NJC) * NJC + b % NJC;
        const double* hp = p.hfrag + (((int64_t)role * (4 * NIW * NJC) + n) * 64 + lane) * 16;
#pragma unroll
        for (int e = 0; e < 16; ++e) q += hp[e] * (double)acc[b][e];
    });
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o);
    double* red = reinterpret_cast<double*>(lds);
    __syncthreads();
    if (lane == 0) red[W] = q;
    __syncthreads();
    if (W == 0 && lane == 0) p.qpart[(int64_t)cand * p.QS + p.q0 + role * p.S + split] = (red[0] + red[1]) + (red[2] + red[3]);
}

template <int NIW, int NJC>
__global__ __launch_bounds__(256, 1) void k_ga_rect(GaQuadArgs p) {
#if defined(__HIP_DEVICE_COMPILE__)
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];      // [2][4 NIW + NJC][64][16]
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int cand = blockIdx.x / (p.S * p.R), rem = blockIdx.x % (p.S * p.R), role = rem / p.S, split = rem % p.S;
    if (w == 0) ga_rect_wave<NIW, NJC, 0>(p, lds, lane, cand, split, role);
    else if (w == 1) ga_rect_wave<NIW, NJC, 1>(p, lds, lane, cand, split, role);
    else if (w == 2) ga_rect_wave<NIW, NJC, 2>(p, lds, lane, cand, split, role);
    else ga_rect_wave<NIW, NJC, 3>(p, lds, lane, cand, split, role);
#endif
}

// ------------------------------------------------------------------------------------------------ finish: linear term + score
// One group of G = 2^bits threads per candidate (k_score_sorted's scheme): thread t owns the run of level klo + t.
template <int G>
__global__ __launch_bounds__(256) void k_ga_finish(const float* __restrict__ sorted, const double* __restrict__ prefix, int64_t n,
                                                   const float* __restrict__ scale, const float* __restrict__ zp, int P, float qmax,
                                                   const double* __restrict__ qpart, int S, const double* __restrict__ s0, double norm,
                                                   float* __restrict__ scores) {
    constexpr int GPB = 256 / G;
    __shared__ int64_t bnd[GPB][G + 1];
    __shared__ double red[256];
    const int gi = threadIdx.x / G, t = threadIdx.x % G;
    const int cand = blockIdx.x * GPB + gi;
    const bool live = cand < P;
    const float s = live ? scale[cand] : 1.0f, z = live ? rintf(zp[cand]) : 0.0f;
    const float klo = ceilf(-z), khi = floorf(qmax - z);
    // (everything that does not depend on the boundaries is requested now)
    double quad = 0.0;
    if (live && t == 0)
        for (int sp = 0; sp < S; ++sp) quad += qpart[(int64_t)cand * S + sp];
    const double s0v = s0[0];
    // (tried on the same box and dropped: the boundary as a threshold float searched 16-ary with 15 loads in flight, a coarse table of
    // the sorted tensor in LDS, both searches of thread 0 in lockstep -- 17.9 -> 19.3 us: whatever the search, the launch is a chain of
    // ~8 dependent far-memory round trips on eight CUs, and the first ten steps of a bisection share their cache lines)
    auto lower = [&](float target) {                                   // first i with rne(x[i] / s) >= target
        int64_t lo = 0, hi = n;
        while (lo < hi) {
            const int64_t mid = (lo + hi) >> 1;
            if (rintf(sorted[mid] / s) >= target) hi = mid; else lo = mid + 1;
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
            return (double)(q - z) * (prefix[b] - prefix[a]);          // the integer level value times the run's sum of C
        };
        acc = run(bnd[gi][t], bnd[gi][t + 1], klo + (float)t);
        if (t == 0) acc += run(0, bnd[gi][0], klo - 1.0f) + run(bnd[gi][G], n, khi + 1.0f);
    }
    red[threadIdx.x] = acc;
    __syncthreads();
#pragma unroll
    for (int o = G / 2; o > 0; o >>= 1) {
        if (t < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (live && t == 0) {
        const double sd = (double)s;
        scores[cand] = (float)(-norm * (s0v - 2.0 * sd * red[threadIdx.x] + sd * sd * quad));
    }
}

// ------------------------------------------------------------------------------------------------ host side
struct GaPlan {
    int T, O, K, NJ; int64_t Tp, Op, n;
    int nchunk, S, cps;                            // the triangle launch: token splits, chunks per split
    int NJH, RT;                                   // k-blocks per triangle role, triangle roles (1: the whole triangle; 2: A x A and B x B)
    int RR, NIW, NJC, SR, cpsr;                    // rectangle roles (0: none), their shape, token splits, chunks per split
    int QS;                                        // partial sums per candidate: RT * S + RR * SR
    int64_t off_hfrag, off_prefix, off_s0, off_wt, off_rl, off_s0p, off_cscl, off_cpart, off_C, off_Cs, off_wsd, off_bsum, total;
    bool ok;
};

static int64_t al256(int64_t v) { return (v + 255) / 256 * 256; }
static bool ga_nj_ok(int nj) { return nj == 1 || nj == 2 || nj == 3 || nj == 4 || nj == 6 || nj == 8 || nj == 12 || nj == 16 || nj == 24; }

static int ga_cus() { return adalog_device_cus(); }   // per device ordinal (common.h)

static GaPlan ga_plan(int T, int O, int K, int P) {
    GaPlan g{};
    g.T = T; g.O = O; g.K = K;
    g.ok = T >= 1 && O >= 1 && K >= 32 && K % 32 == 0 && (int64_t)T * K < ((int64_t)1 << 31) && (int64_t)RLIMBS * T < ((int64_t)1 << 30) && P >= 1;
    if (!g.ok) return g;
    g.NJ = K / 32;
    g.Tp = ((int64_t)T + 127) / 128 * 128;
    g.Op = ((int64_t)O + 127) / 128 * 128;
    g.n = (int64_t)T * K;
    g.nchunk = (int)(g.Tp / 32);
    // token splits: P * S workgroups, one per CU at a time -- the smallest S that fills the chip, chunks <= 4096 (int32 accumulators:
    // 127^2 * 32 * 4096 < 2^31)
    auto splits = [&](int wgs_per_split) {
        int S = (ga_cus() + wgs_per_split - 1) / wgs_per_split;
        if (S < 1) S = 1;
        while ((g.nchunk + S - 1) / S > 4096) ++S;
        if (S > g.nchunk) S = g.nchunk;
        return S;
    };
    // K <= 384: one role, the whole triangle.  K = 512 / 768: halves A | B -- two triangle roles and the A x B square as one 8 x 8
    // rectangle (a wave 2 x 8 blocks) / two 12 x 6 rectangles (a wave 3 x 6 blocks)
    g.NJH = g.NJ; g.RT = 1; g.RR = 0; g.NIW = g.NJC = 0; g.SR = 0; g.cpsr = 0;
    if (g.NJ == 16) { g.NJH = 8; g.RT = 2; g.RR = 1; g.NIW = 2; g.NJC = 8; }
    if (g.NJ == 24) { g.NJH = 12; g.RT = 2; g.RR = 2; g.NIW = 3; g.NJC = 6; }
    g.S = splits(P * g.RT);
    g.cps = (g.nchunk + g.S - 1) / g.S;
    if (g.RR) { g.SR = splits(P * g.RR); g.cpsr = (g.nchunk + g.SR - 1) / g.SR; }
    g.QS = g.RT * g.S + g.RR * g.SR;
    const int nbt = g.NJ * (g.NJ + 1) / 2;
    int64_t off = 0;
    g.off_hfrag = off; off += al256((int64_t)nbt * 1024 * 8);
    g.off_prefix = off; off += al256((g.n + 1) * 8);
    g.off_s0 = off; off += 256;
    g.off_wt = off; off += al256((int64_t)K * g.Op);
    g.off_rl = off; off += al256((int64_t)RLIMBS * T * g.Op);
    g.off_s0p = off; off += al256((int64_t)T * 8);
    g.off_cscl = off; off += al256((int64_t)T * 8);
    g.off_cpart = off; off += al256((int64_t)RLIMBS * T * K * 4);
    g.off_C = off; off += al256(g.n * 8);
    g.off_Cs = off; off += al256(g.n * 8);
    g.off_wsd = off; off += al256((int64_t)O * K * 8);
    g.off_bsum = off; off += al256(((g.n + PBLK - 1) / PBLK) * 8);
    g.total = off;
    return g;
}

}  // namespace

// The Gram form of an output-MSE activation search is supported for a per-tensor uniform activation quantiser with <= 7-bit
// operands, K % 32 == 0 up to 384 (the candidate's whole K x K upper triangle lives in one CU's registers) or K = 512 / 768 (the
// triangle in three / four parts, k_ga_rect) and an instantiated K.
extern "C" int adalog_gram_act_supported(int T, int O, int K, int a_bits, int w_bits, int P) {
    const GaPlan g = ga_plan(T, O, K, P);
    if (!g.ok || !ga_nj_ok(g.NJ) || a_bits < 2 || a_bits > 7 || w_bits < 2 || w_bits > 7 || P > 65535) return 0;
    if (O >= (1 << 15)) return 0;                          // |C_fix| < 2^53
    return 1;
}

// ... and it pays: the candidate Gram matrices cost K / 2 multiply-adds per generated element against the token form's O
extern "C" int adalog_gram_act_ok(int T, int O, int K, int a_bits, int w_bits, int P) {
    if (!adalog_gram_act_supported(T, O, K, a_bits, w_bits, P) || T < 256) return 0;
    return K <= 384 || O >= 2 * K ? 1 : 0;                 // the split forms generate every fragment 2 - 2.5 times
}

extern "C" int64_t adalog_gram_act_workspace_bytes(int T, int O, int K, int P) {
    const GaPlan g = ga_plan(T, O, K, P);
    return g.ok ? g.total : -1;
}

/* once per captured activation: x [T][ldx] fp32 -> xt [K][Tp] fp32 (Tp = T rounded up to 128, zero padded), sorted copy of the T K
 * values, and the permutation that sorts them (perm[i] = flat index t K + k of the i-th smallest).  sort_ws: adalog_gram_act_sort_bytes. */
extern "C" int64_t adalog_gram_act_sort_bytes(int64_t n) {
    if (n < 1 || n >= ((int64_t)1 << 31)) return -1;
    return al256(adalog_sort_workspace_bytes(1, n, 1));                 // csrc/radix_sort.hip
}

extern "C" int adalog_gram_act_prepare(const float* x, int T, int K, int64_t ldx, float* xt, float* sorted, unsigned int* perm,
                                       void* sort_ws, int64_t sort_ws_bytes, void* stream) {
    ADALOG_ARG_CHECK(x && xt && sorted && perm && sort_ws && T >= 1 && K >= 1 && ldx == K, "gram_act_prepare: bad arguments (x must be contiguous)");
    const int64_t n = (int64_t)T * K;
    ADALOG_ARG_CHECK(n < ((int64_t)1 << 31) && sort_ws_bytes >= adalog_gram_act_sort_bytes(n), "gram_act_prepare: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    const int64_t Tp = ((int64_t)T + 127) / 128 * 128;
    ADALOG_ARG_CHECK(K % 32 == 0, "gram_act_prepare: K must be a multiple of 32");
    hipLaunchKernelGGL(k_ga_fragorder, dim3((unsigned)(Tp / 32), (unsigned)(K / 32)), dim3(256), 0, st, x, T, K, ldx, xt, (int)(Tp / 32));
    // the values sorted, and the permutation that sorts them (flat indices t K + k): one segment of the hand-written radix sort
    const int rc = adalog_sort_f32(x, 1, n, sorted, perm, sort_ws, sort_ws_bytes, stream);
    if (rc) return rc;
    ADALOG_LAUNCH_CHECK("adalog_gram_act_prepare");
    return 0;
}

/* once per activation_fpcs call: the weight quantiser is fixed.  raw_out [T][O], bias [O] or null, W [O][ldw] with its (s_w, z_w) [O],
 * perm from adalog_gram_act_prepare.  Leaves hfrag, the prefix sums of C and S0 in the workspace. */
extern "C" int adalog_gram_act_build(const float* raw_out, int T, int O, const float* bias, const float* W, int K, int64_t ldw,
                                     const float* sw, const float* zw, int w_bits, const unsigned int* perm, int P, void* ws,
                                     int64_t ws_bytes, void* stream) {
    ADALOG_ARG_CHECK(raw_out && W && sw && zw && perm && ws, "gram_act_build: null pointer");
    const GaPlan g = ga_plan(T, O, K, P);
    ADALOG_ARG_CHECK(g.ok && ga_nj_ok(g.NJ) && w_bits >= 2 && w_bits <= 7, "gram_act_build: shape not supported (adalog_gram_act_supported)");
    ADALOG_ARG_CHECK(ws_bytes >= g.total && ((uintptr_t)ws & 255) == 0, "gram_act_build: workspace too small / unaligned");
    hipStream_t st = (hipStream_t)stream;
    uint8_t* base = (uint8_t*)ws;
    int8_t* wt = (int8_t*)(base + g.off_wt);
    int8_t* rl = (int8_t*)(base + g.off_rl);
    double* s0p = (double*)(base + g.off_s0p);
    double* cscl = (double*)(base + g.off_cscl);
    int* cpart = (int*)(base + g.off_cpart);
    double* C = (double*)(base + g.off_C);
    double* bsum = (double*)(base + g.off_bsum);
    double* wsd = (double*)(base + g.off_wsd);
    hipLaunchKernelGGL(k_ga_pack_wt, dim3((unsigned)(g.Op / 64), (unsigned)((K + 63) / 64)), dim3(256), 0, st, W, O, K, ldw, sw, zw,
                       (float)((1 << w_bits) - 1), wt, g.Op, wsd);
    hipLaunchKernelGGL(k_ga_rfix, dim3((unsigned)T), dim3(256), 0, st, raw_out, T, O, g.Op, bias, sw, rl, s0p, cscl);
    const int RA = RLIMBS * T;
    hipLaunchKernelGGL(k_i8mm, dim3((unsigned)((RA + 127) / 128), (unsigned)((K + 127) / 128), 1), dim3(256), 0, st, rl, wt, RA, K, g.Op,
                       (int)(g.Op / 128), (int)(g.Op / 128), cpart);
    hipLaunchKernelGGL(k_ga_fin_c, dim3((unsigned)((g.n + 255) / 256)), dim3(256), 0, st, cpart, T, K, cscl, C);
    const int nb = (int)((g.n + PBLK - 1) / PBLK);
    double* Cs = (double*)(base + g.off_Cs);
    hipLaunchKernelGGL(k_ga_gather_sum, dim3((unsigned)nb), dim3(256), 0, st, C, perm, g.n, Cs, bsum);
    hipLaunchKernelGGL(k_ga_scan, dim3(2), dim3(256), 0, st, bsum, nb, s0p, T, (double*)(base + g.off_s0));
    hipLaunchKernelGGL(k_ga_prefix, dim3((unsigned)nb), dim3(256), 0, st, Cs, g.n, bsum, (double*)(base + g.off_prefix));
    const int nbt = g.NJ * (g.NJ + 1) / 2;
    double* hfrag = (double*)(base + g.off_hfrag);
    const int ntt = g.NJH * (g.NJH + 1) / 2;
    for (int r = 0; r < g.RT; ++r)                             // role order of the score launches: triangles, then rectangles
        hipLaunchKernelGGL(k_ga_h, dim3((unsigned)ntt), dim3(64 * GA_H_WAVES), 0, st, wsd, O, K, hfrag + (int64_t)r * ntt * 1024, r * g.NJH, 0,
                           g.NJH, g.NJH, 1);
    for (int r = 0; r < g.RR; ++r) {
        const int ntr = 4 * g.NIW * g.NJC;
        hipLaunchKernelGGL(k_ga_h, dim3((unsigned)ntr), dim3(64 * GA_H_WAVES), 0, st, wsd, O, K,
                           hfrag + ((int64_t)g.RT * ntt + (int64_t)r * ntr) * 1024, 0, g.NJH + r * g.NJC, 4 * g.NIW, g.NJC, 0);
    }
    (void)nbt;
    ADALOG_LAUNCH_CHECK("adalog_gram_act_build");
    return 0;
}

/* one FPCS step: scores [P] = -norm * sum_{t,o} (raw_out - bias - s_p Wq . x_p)^2 for the P per-tensor candidates (scale, zp).
 * xt / sorted from adalog_gram_act_prepare, ws from adalog_gram_act_build (same T, O, K, P); qpart: P * adalog_gram_act_splits doubles. */
extern "C" int adalog_gram_act_splits(int T, int O, int K, int P) {
    const GaPlan g = ga_plan(T, O, K, P);
    return g.ok ? g.QS : -1;
}

static long long* g_ga_timeline = nullptr;
extern "C" void adalog_gram_act_set_timeline(long long* buf) { g_ga_timeline = buf; }   // lab only

// `tail` (may be null): the FPCS step's ranking + next grid / commit in the finish kernel's last block (fpcs_tail.h); scale / zp of the
// tail are the candidates scored here ([P][1]).
extern "C" int adalog_gram_act_score_tail(const float* xt, const float* sorted, int T, int O, int K, const float* scale, const float* zp, int P,
                                          int a_bits, const void* ws, double norm, double* qpart, float* scores,
                                          const adalog_fpcs_tail* tail, void* stream) {
    ADALOG_ARG_CHECK(xt && sorted && scale && zp && ws && qpart && scores, "gram_act_score: null pointer");
    const char* why = fpcs::tail_problem(tail, P);
    ADALOG_ARG_CHECK(why == nullptr, why);

    ADALOG_ARG_CHECK(adalog_gram_act_supported(T, O, K, a_bits, a_bits, P), "gram_act_score: shape not supported (adalog_gram_act_supported)");
    const GaPlan g = ga_plan(T, O, K, P);
    const uint8_t* base = (const uint8_t*)ws;
    GaQuadArgs a{};
    a.xt = xt; a.Tp = g.Tp; a.K = K; a.scale = scale; a.zp = zp; a.P = P;
    a.hfrag = (const double*)(base + g.off_hfrag); a.qpart = qpart;
    a.S = g.S; a.chunks_per_split = g.cps; a.nchunk = g.nchunk;
    a.NJT = g.NJ; a.R = g.RT; a.QS = g.QS; a.q0 = 0; a.kb_a = 0; a.kb_b = 0;
    a.qmax = (float)((1 << a_bits) - 1);
    const float zone = 6e-7f * (float)(1 << a_bits);
    a.tie = 0.5f - (zone > 1e-5f ? zone : 1e-5f);
    a.timeline = g_ga_timeline;
    hipStream_t st = (hipStream_t)stream;
#define GA_LAUNCH(NJV)                                                                                            \
    do {                                                                                                          \
        const size_t shm = (size_t)2 * NJV * 1024 < 64 ? 64 : (size_t)2 * NJV * 1024;                             \
        adalog_note_kernel("k_gram_act<i8>");                                                                     \
        hipLaunchKernelGGL((k_ga_quad<NJV>), dim3((unsigned)(P * g.S * g.RT)), dim3(256), shm, st, a);            \
    } while (0)
    switch (g.NJH) {
        case 1: GA_LAUNCH(1); break;
        case 2: GA_LAUNCH(2); break;
        case 3: GA_LAUNCH(3); break;
        case 4: GA_LAUNCH(4); break;
        case 6: GA_LAUNCH(6); break;
        case 8: GA_LAUNCH(8); break;
        case 12: GA_LAUNCH(12); break;
        default: ADALOG_ARG_CHECK(false, "gram_act_score: K not instantiated");
    }
#undef GA_LAUNCH
    if (g.RR) {                                                // the A x B square
        GaQuadArgs b = a;
        const int ntt = g.NJH * (g.NJH + 1) / 2;
        b.hfrag = a.hfrag + (int64_t)g.RT * ntt * 1024;
        b.S = g.SR; b.chunks_per_split = g.cpsr; b.R = g.RR; b.q0 = g.RT * g.S; b.kb_a = 0; b.kb_b = g.NJH;
        b.timeline = nullptr;
        const size_t shm = (size_t)2 * (4 * g.NIW + g.NJC) * 1024;
        const dim3 grid((unsigned)(P * g.SR * g.RR));
        if (g.NIW == 2) hipLaunchKernelGGL((k_ga_rect<2, 8>), grid, dim3(256), shm, st, b);
        else hipLaunchKernelGGL((k_ga_rect<3, 6>), grid, dim3(256), shm, st, b);
    }
    const double* prefix = (const double*)(base + g.off_prefix);
    const double* s0 = (const double*)(base + g.off_s0);
    const int G = 1 << a_bits;
    const int gpb = 256 / G;
    const unsigned blocks = (unsigned)((P + gpb - 1) / gpb);
#define GA_FIN(GV) hipLaunchKernelGGL((k_ga_finish<GV>), dim3(blocks), dim3(256), 0, st, sorted, prefix, g.n, scale, zp, P, a.qmax, qpart, g.QS, s0, norm, scores)
    switch (a_bits) {
        case 2: GA_FIN(4); break;
        case 3: GA_FIN(8); break;
        case 4: GA_FIN(16); break;
        case 5: GA_FIN(32); break;
        case 6: GA_FIN(64); break;
        default: GA_FIN(128); break;
    }
#undef GA_FIN
    ADALOG_LAUNCH_CHECK("adalog_gram_act_score");
    // the FPCS step's tail: a second launch.  Both in-launch forms were measured and lost (round 6): the last of the finish kernel's
    // eight workgroups ranking by ticket (26.9 us against 17.9 + 10.8: the agent-scope ticket and score reads are ~2 us round trips
    // each), and ONE workgroup of 1 024 threads taking all 2 176 bisections (36.6 us: a single CU's address path serialises the
    // divergent loads)
    if (tail) return adalog_topk_next_tail(scores, P, 1, tail, nullptr, stream);
    return 0;
}

extern "C" int adalog_gram_act_score(const float* xt, const float* sorted, int T, int O, int K, const float* scale, const float* zp, int P,
                                     int a_bits, const void* ws, double norm, double* qpart, float* scores, void* stream) {
    return adalog_gram_act_score_tail(xt, sorted, T, O, K, scale, zp, P, a_bits, ws, norm, qpart, scores, nullptr, stream);
}
