// K7 (Gram form) -- output-MSE weight searches of the uniformly quantised Linear layers scored from the Gram matrix of the
// quantised activation instead of from its tokens.
//
// Reference: quant_layers/linear.py:355-392 (_search_best_w_scale).  For candidate p and output row o it scores
//     sum_t ( r[t,o] - sigma * x_int[t,:] . w_int_p[o,:] )^2,   r = raw_out - bias,  sigma = s_a * s_w[p,o],
// with the activation quantiser FIXED for the whole weight_fpcs call (linear.py:483-503: six such calls).  That is a
// quadratic form in the candidate's integer row w = w_int_p[o,:]:
//     S0[o] - 2 sigma (w . c[o,:]) + sigma^2 (w^T G w),    G = X_int^T X_int  [K,K],  c[o,:] = X_int^T r[:,o],  S0 = sum_t r^2.
// G, c and S0 are built ONCE per weight_fpcs call (adalog_gram_build: 2 T K^2 + 2 T K O multiply-adds); every candidate row
// then costs K^2 multiply-adds per limb of G instead of T K: (limbs K / T) of the work -- 0.18 for deit_small qkv, 0.005 for
// swin_base stage 0.
//
// Exactness.  The three terms are each ~ |r|^2 and cancel to the score (1e-2 ... 1e-4 of it), so they are carried exactly:
//   * x_int = q - z (|.| <= 2^bits - 1) is int8; G = X^T X is an exact integer, accumulated in int32 per token chunk and int64
//     across chunks, then split into balanced int8 limbs (3 for 4-bit activations of 6 304 tokens): w^T G w runs on
//     v_mfma_i32_32x32x32_i8 once per limb with exact int32 accumulators; the row dot with w and the limb recombination are exact
//     integers folded into fp64 below 2^53 ... 2^56 (relative rounding 1e-16);
//   * r is rounded ONCE per output column to 30-bit fixed point relative to the column's largest magnitude (four balanced int8
//     limbs; |delta r| <= 2^-30 max|r|: far below fp32's own 2^-24 relative spacing for every element within 2^6 of the
//     maximum), after which c = X^T r_fix and S0 = sum r_fix^2 are exact integers too (int8 MFMA, int64); the score returned
//     is the EXACT score of the rounded reference up to the final fp64 combination;
//   * sigma = fl32(s_a * s_w) as in the token-form kernels' epilogue (gemm_k_slab.inc), the combination in fp64.
// Against the fp32 ATen evaluation (~5e-6 relative noise of its own) the difference is <= 1e-6 (tests/test_gpu_kernels.py).
//
// Kernels:
//   k_gram_pack_xt   x [T,K] fp32 -> (q - z) int8, TRANSPOSED [K][Tp] (token-contiguous: the contraction runs over tokens)
//   k_gram_rfix      raw_out^T [O,T] fp32, bias -> four int8 limb planes [4][O][Tp], S0[o], 2^-e_o
//   k_i8mm           (gram_mm.inc) C[i,j] = sum_t A[i,t] B[j,t]  (int8 MFMA, int32 per chunk -> int64 partial per token split); A optionally in
//                    limb planes (recombined with shifts): G = xt.xt^T and c = rfix.xt^T
//   k_gram_fin_g     sum of the split partials -> balanced limbs in MFMA A-fragment order (rows permuted so that a lane's
//                    accumulator rows are the k of its own B fragment)
//   k_gram_fin_c     sum of the split partials -> 8 balanced limbs [O][8][K] (the rows of a 9th "panel": w . c on the MFMA too)
//   k_gram_score     per FPCS step: a wave owns up to CB blocks of 32 candidates; a lane IS a candidate: it quantises its weight
//                    row into the B fragments (registers, all of K) with the slab GEN form's exact-bin arithmetic, then streams
//                    the (row tile, limb) panels of G through an LDS ring shared by the workgroup's four waves:
//                    12 MFMAs per panel and block, then sum_e acc[e] * w[e] over the lane's own 16 rows (v_mad_i32_i24).
#include "common.h"
#include "fpcs_tail.h"
#include <stdlib.h>
#include <type_traits>
#include <utility>

namespace {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

// compile-time loop: f(integral_constant<int, 0>), ..., f(integral_constant<int, N - 1>).  (#pragma unroll is a request the optimiser
// declines for bodies this large; an index that stays dynamic sends the register-resident fragment array to scratch.)
template <typename F, int... I> __device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>) {
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, typename F> __device__ __forceinline__ void static_for(F&& f) { static_for_impl(f, std::make_integer_sequence<int, N>{}); }

constexpr int RLIMBS = 4;          // int8 limbs of the fixed-point reference (30 bits + sign)
constexpr int CLIMBS = 8;          // int8 limbs of c = X^T r_fix (<= 2^62)
constexpr int RFIX_BITS = 29;      // |r_fix| < 2^30

// ------------------------------------------------------------------------------------------------ pack x^T
// bins exactly as uni_bin_fast (common.h), value = bin - z
__global__ __launch_bounds__(256) void k_gram_pack_xt(const float* __restrict__ x, int T, int K, int64_t ldx, const float* __restrict__ sa,
                                                      const float* __restrict__ za, float qmax, int8_t* __restrict__ xt, int64_t Tp) {
#pragma clang fp contract(off)
    __shared__ uint32_t tile[64][17];                       // [token][16 dwords of 4 channels] (+1: bank spread)
    const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
    const int m0 = blockIdx.x * 64, k0 = blockIdx.y * 64;
    const float s = sa[0], z = rintf(za[0]), inv = 1.0f / s;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = m0 + ty + 16 * i, k = k0 + 4 * tx;
        uint32_t pk = 0;
        if (m < T && k < K) {
            const float4 v = *reinterpret_cast<const float4*>(x + (int64_t)m * ldx + k);
            const float e[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int q = (int)(uni_bin_fast(e[j], s, inv, z, qmax) - z);
                pk |= ((uint32_t)(q & 0xff)) << (8 * j);
            }
        }
        tile[ty + 16 * i][tx] = pk;
    }
    __syncthreads();
    const int c = tid >> 2, part = tid & 3;                 // channel within the tile, 16-token part
    if (k0 + c < K) {
        uint32_t o[4];
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            uint32_t u = 0;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const uint32_t d = tile[part * 16 + w * 4 + j][c >> 2];
                u |= ((d >> (8 * (c & 3))) & 0xffu) << (8 * j);
            }
            o[w] = u;
        }
        *reinterpret_cast<uint4*>(xt + (int64_t)(k0 + c) * Tp + m0 + part * 16) = make_uint4(o[0], o[1], o[2], o[3]);
    }
}

// ------------------------------------------------------------------------------------------------ fixed-point reference
__device__ __forceinline__ double block_sum_d(double v, double* sm) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    const int w = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sm[w] = v;
    __syncthreads();
    return (sm[0] + sm[1]) + (sm[2] + sm[3]);
}

// pass 1: amax[o] = max_t |raw_out[t][o] - bias[o]| as the bits of a non-negative float (integer order = float order)
__global__ __launch_bounds__(256) void k_gram_rmax(const float* __restrict__ ref_t, int T, int chunk, const float* __restrict__ bias,
                                                   unsigned int* __restrict__ amax) {
#pragma clang fp contract(off)
    __shared__ float smf[4];
    const int o = blockIdx.x, tid = threadIdx.x;
    const int m0 = blockIdx.y * chunk, m1 = min(T, m0 + chunk);
    const float* r = ref_t + (int64_t)o * T;
    const float b = bias ? bias[o] : 0.0f;
    float am = 0.0f;
    for (int m = m0 + tid; m < m1; m += 256) am = fmaxf(am, fabsf(r[m] - b));
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) am = fmaxf(am, __shfl_xor(am, d));
    if ((tid & 63) == 0) smf[tid >> 6] = am;
    __syncthreads();
    if (tid == 0) {
        am = fmaxf(fmaxf(smf[0], smf[1]), fmaxf(smf[2], smf[3]));
        if (am == am) atomicMax(amax + o, __float_as_uint(am));          // (NaN: left out, as fmaxf does)
    }
}

// pass 2: the limb planes of chunk blockIdx.y of row o, and that chunk's share of S0 (s0p[o][chunk]: summed in a fixed order later)
__global__ __launch_bounds__(256) void k_gram_rfix(const float* __restrict__ ref_t, int T, int64_t Tp, int O, int chunk, int nchunk,
                                                   const float* __restrict__ bias, const unsigned int* __restrict__ amax,
                                                   int8_t* __restrict__ rl, double* __restrict__ s0p, double* __restrict__ cscl) {
#pragma clang fp contract(off)
    __shared__ double smd[4];
    const int o = blockIdx.x, tid = threadIdx.x;
    const int m0 = blockIdx.y * chunk, m1 = (int)min((int64_t)Tp, (int64_t)m0 + chunk);       // chunk is a multiple of 128: the last one pads
    const float* r = ref_t + (int64_t)o * T;
    const float b = bias ? bias[o] : 0.0f;
    const float am = __uint_as_float(amax[o]);
    // e: am * 2^e in [2^29, 2^30)   (am == 0 or non-finite: e = 0, every limb 0 / garbage in, garbage out as in the token form)
    int e = 0;
    if (am > 0.0f && am < 3.0e38f) e = RFIX_BITS - ilogbf(am);
    const double sc = ldexp(1.0, e);
    double acc = 0.0;
    int8_t* l0 = rl + (int64_t)o * Tp;
    const int64_t plane = (int64_t)O * Tp;
    for (int m = m0 + tid; m < m1; m += 256) {
        int v = 0;
        if (m < T) v = (int)rint((double)(r[m] - b) * sc);               // exact product; one rounding to the 2^-e grid
        acc += (double)v * (double)v;
        int rest = v;
#pragma unroll
        for (int l = 0; l < RLIMBS; ++l) {
            const int d = (int)(int8_t)(rest & 0xff);
            l0[l * plane + m] = (int8_t)d;
            rest = (rest - d) >> 8;
        }
    }
    const double tot = block_sum_d(acc, smd);
    if (tid == 0) {
        s0p[(int64_t)o * nchunk + blockIdx.y] = ldexp(tot, -2 * e);
        if (blockIdx.y == 0) cscl[o] = ldexp(1.0, -e);
    }
}

// ------------------------------------------------------------------------------------------------ C = A . B^T over tokens (int8)
// A [RA][ld], B [RB][ld] int8, token-contiguous; part[z][RA][RB] int32 = the sum over the z-th token range (<= 512 steps of 128
// tokens: |sum| < 2^31): gram_mm.inc.  The reference's limb planes are simply more rows of A (RA = 4 O); the limbs are recombined
// when the splits are summed (k_gram_fin_c).
#include "gram_mm.inc"          // k_i8mm

// ------------------------------------------------------------------------------------------------ finalise G and c
// gfrag[l][kt][jt][lane][16]: lane = (tile row r', half h); tile row r' = 8 i + 4 h' + j holds G row 32 kt + 16 h' + 4 i + j, so
// that accumulator element e = 4 i + j of lane (., h') is G row 32 kt + 16 h' + e -- the k of byte e of that lane's B fragment kt.
__global__ __launch_bounds__(256) void k_gram_fin_g(const int* __restrict__ part, int splits, int K, int NL, int8_t* __restrict__ gfrag) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;          // one thread per (tile row slot, column) of G
    if (idx >= (int64_t)K * K) return;
    const int col = (int)(idx % K), slot = (int)(idx / K);                // slot = 32 kt + r'
    const int kt = slot >> 5, rp = slot & 31;
    const int row = 32 * kt + 16 * ((rp >> 2) & 1) + 4 * (rp >> 3) + (rp & 3);
    long long g = 0;
    for (int z = 0; z < splits; ++z) g += (long long)part[((int64_t)z * K + row) * K + col];
    const int nj = K >> 5, jt = col >> 5, h = (col >> 4) & 1, e = col & 15;
    long long rest = g;
    for (int l = 0; l < NL; ++l) {
        const int d = (int)(int8_t)(rest & 0xff);
        gfrag[((((int64_t)l * nj + kt) * nj + jt) * 64 + (rp + 32 * h)) * 16 + e] = (int8_t)d;
        rest = (rest - d) >> 8;
    }
}

__global__ __launch_bounds__(256) void k_gram_fin_c(const int* __restrict__ part, int splits, int O, int K, int8_t* __restrict__ clim,
                                                    const double* __restrict__ s0p, int nchunk, double* __restrict__ s0) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx < O) {                                          // S0[o]: the chunks' shares in a fixed order
        double t = 0.0;
        for (int ch = 0; ch < nchunk; ++ch) t += s0p[idx * nchunk + ch];
        s0[idx] = t;
    }
    if (idx >= (int64_t)O * K) return;
    const int o = (int)(idx / K), k = (int)(idx % K);
    long long c = 0;
    for (int z = 0; z < splits; ++z)
#pragma unroll
        for (int l = 0; l < RLIMBS; ++l) c += ((long long)part[((int64_t)z * RLIMBS * O + (int64_t)l * O + o) * K + k]) << (8 * l);
    long long rest = c;
#pragma unroll
    for (int l = 0; l < CLIMBS; ++l) {
        const int d = (int)(int8_t)(rest & 0xff);
        clim[((int64_t)o * CLIMBS + l) * K + k] = (int8_t)d;
        rest = (rest - d) >> 8;
    }
}

// LDS fragment reads of the panel pipeline as volatile asm (the compiler keeps their order and places no waits of its own);
// gram_wait<N> ties a fragment to the counted wait that makes it valid; gram_rd_after names the accumulators an earlier MFMA
// writes as a nominal input, so the read cannot be moved above the MFMAs that still use its destination registers.
template <int OFF> __device__ __forceinline__ v4i gram_rd(uint32_t addr) {
    v4i r;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(OFF));
    return r;
}
template <int OFF> __device__ __forceinline__ v4i gram_rd_after(uint32_t addr, v16i& tie) {
    v4i r;
    asm volatile("ds_read_b128 %0, %2 offset:%3" : "=v"(r), "+v"(tie) : "v"(addr), "n"(OFF));
    return r;
}
template <int N> __device__ __forceinline__ void gram_wait(v4i& f) {
    if constexpr (N >= 2) asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(f));
    else if constexpr (N == 1) asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(f));
    else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f));
}

struct GramScoreArgs {
    const float* W; int64_t ldw; int O, K, P;
    const float* scale; const float* zp;       // [P][O]
    const int8_t* gfrag; int NL;               // [NL][NJ][NJ][64][16]
    const int8_t* clim;                        // [O][8][K]
    const double* s0; const double* cscl;      // [O]
    const float* sa;                           // device scalar
    double norm;
    float qmax, tie;
    float* scores;                             // [P][O]
    int nblk;                                  // O * P / 32
    long long* timeline;                       // lab only (tools/lab/gram_check.py): cycle stamps of wave 0 of each workgroup, else null
};

// One pass of a wave over NB (<= CB) blocks of 32 candidates; every wave of the workgroup runs the same panel sequence (NB = 0: it
// only helps moving the panels).  Two workgroups share a CU and run unsynchronised: one's row dots (VALU) issue under the other's MFMAs.
// COOP >= 0 (= the wave's index): the thin tail of a workgroup's range -- at most CB blocks are left, three waves would idle through
// all NJ NL panels -- is run COOPERATIVELY: all four waves take the SAME blocks and split the contraction index, wave COOP owning
// the K chunks [COOP NJ/4, (COOP + 1) NJ/4).  The row dot is linear in the accumulators, so each wave applies it to its partial
// products and only the per-candidate sums (quad, lin) are added across the waves at the end.
template <int NJ, int NB, bool BIG, bool TIE, int COOP = -1>
__device__ __forceinline__ void gram_pass(const GramScoreArgs& p, uint8_t* lds, int blk0, int w, int lane, float sa, int stamp_slot) {
    constexpr bool CO = COOP >= 0;
    constexpr int QN = CO ? NJ / 4 : NJ;                   // K chunks this wave multiplies
    constexpr int J0 = CO ? COOP * (NJ / 4) : 0;           // ... starting here
    static_assert(!CO || (NJ % 4 == 0 && NB >= 1), "the cooperative pass splits NJ four ways");
    constexpr int NPW = (NJ + 3) / 4;                      // 1 KiB pieces of a panel per wave
    constexpr int NRING = 4;                               // panels in the ring; the same LDS is the waves' staging area while generating
    constexpr int NBA = NB > 0 ? NB : 1;
    const int c = lane & 31, h = lane >> 5;
    const int PB = p.P >> 5;
    const int npanel = NJ * p.NL;
#define GRAM_STAMP(i_) do { if (p.timeline && stamp_slot >= 0 && w == 0 && lane == 0) p.timeline[stamp_slot * 8 + (i_)] = (long long)__builtin_readcyclecounter(); } while (0)
    GRAM_STAMP(0);

    // ---- B fragments: this lane's candidate of each block, all of K (int8 q - z; exact bins, gemm_k_slab.inc GEN form).  The
    // generating loop is rolled (its index is dynamic) and hands the fragments over through the wave's NJ KiB of LDS, from where they
    // are read back into statically named registers.
    v4i bf[NBA][NJ];
    int orow[NBA];
    float sig[NBA];
    // staging: a private NJ KiB per wave, or (cooperative) NJ KiB per BLOCK shared by the four waves, each writing its chunks
    uint8_t* stage = lds + (CO ? 0 : w * (NJ * 1024)) + lane * 16;
    __syncthreads();                                       // the previous pass's ring is dead (all waves): it is the staging area now
    {
#pragma clang fp contract(off)
        static_for<NB>([&](auto bc) {
            constexpr int b = decltype(bc)::value;
            const int blk = blk0 + b;
            const int o = blk / PB, cand = (blk - o * PB) * 32 + c;
            orow[b] = o;
            const int64_t gpi = (int64_t)cand * p.O + o;
            const float gs = p.scale[gpi], gz = rintf(p.zp[gpi]);
            sig[b] = sa * gs;
            const float ginv = __builtin_amdgcn_rcpf(gs);
            const float glo = 128.0f - gz, ghi = 128.0f + (p.qmax - gz);
            const float* __restrict__ wr = p.W + (int64_t)o * p.ldw + 16 * h;
            uint8_t* stb = stage + (CO ? b * (NJ * 1024) : 0);
            float4 nx[4];                                  // the row one 16-value slot ahead of the arithmetic
#pragma unroll
            for (int j = 0; j < 4; ++j) nx[j] = *reinterpret_cast<const float4*>(wr + 32 * J0 + 4 * j);
#pragma unroll 1
            for (int jt = J0; jt < J0 + QN; ++jt) {
                float xv[16];
#pragma unroll
                for (int j = 0; j < 4; ++j) { xv[4 * j] = nx[j].x; xv[4 * j + 1] = nx[j].y; xv[4 * j + 2] = nx[j].z; xv[4 * j + 3] = nx[j].w; }
                {
                    const int jn = min(jt + 1, NJ - 1);
#pragma unroll
                    for (int j = 0; j < 4; ++j) nx[j] = *reinterpret_cast<const float4*>(wr + 32 * jn + 4 * j);
                }
                float kq[16], dm = 0.0f;
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const float t = xv[e] * ginv;
                    kq[e] = rintf(t);
                    dm = fmaxf(dm, fabsf(t - kq[e]));
                }
                if (__builtin_expect(dm > p.tie, 0)) {
#pragma unroll
                    for (int e = 0; e < 16; ++e) kq[e] = rintf(xv[e] / gs);
                }
                v4i pk;
#pragma unroll
                for (int j = 0; j < 4; ++j) {                  // clamp, then one SDWA add converts and inserts the signed byte (common.h)
                    int u = 0;
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        u = sdwa_rne_byte(u, __builtin_amdgcn_fmed3f(kq[4 * j + e], glo - 128.0f, ghi - 128.0f), 12582912.0f, e);
                    pk[j] = u;
                }
                *reinterpret_cast<v4i*>(stb + jt * 1024) = pk;
            }
            if constexpr (!CO)
                static_for<NJ>([&](auto jtc) { bf[b][decltype(jtc)::value] = *reinterpret_cast<const v4i*>(stb + decltype(jtc)::value * 1024); });
        });
        if constexpr (CO) {                                // every wave needs every chunk (the row dot walks all of K)
            __syncthreads();
            static_for<NB>([&](auto bc) {
                constexpr int b = decltype(bc)::value;
                static_for<NJ>([&](auto jtc) {
                    bf[b][decltype(jtc)::value] = *reinterpret_cast<const v4i*>(stage + (b * NJ + decltype(jtc)::value) * 1024);
                });
            });
        }
    }
    GRAM_STAMP(1);

    // ---- w . c on the MFMA: the 8 limb rows of c[o,:] are rows 0..7 of a 32-row A tile (straight from global memory)
    double lin[NBA];
    static_for<NB>([&](auto bc) {
        constexpr int b = decltype(bc)::value;
        v16i acc;
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] = 0;
        const int8_t* cr = p.clim + ((int64_t)orow[b] * CLIMBS + (c & 7)) * p.K + 16 * h;
        static_for<QN>([&](auto jtc) {
            constexpr int jt = J0 + decltype(jtc)::value;
            v4i a = *reinterpret_cast<const v4i*>(cr + 32 * jt);
            if (c >= CLIMBS) a = v4i{0, 0, 0, 0};
            acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, bf[b][jt], acc, 0, 0, 0);
        });
        // rows 4 h + j (j < 4) = limbs 4 h + j of this lane's candidate
        double v = 0.0;
#pragma unroll
        for (int j = 0; j < 4; ++j) v += (double)acc[j] * (h ? 4294967296.0 : 1.0) * (double)(1 << (8 * j));
        v += __shfl_xor(v, 32);
        lin[b] = v;
    });
    GRAM_STAMP(2);

    // ---- w^T G w: stream the (l, kt) panels of G through the LDS ring
    double quad[NBA];
#pragma unroll
    for (int b = 0; b < NBA; ++b) quad[b] = 0.0;
    v4i stg[NPW];
    auto fetch = [&](int pn) {                            // this wave's pieces of panel pn -> registers
        const int pc = min(pn, npanel - 1);
        const int8_t* src = p.gfrag + (int64_t)pc * NJ * 1024;            // panel (l, kt) = gfrag[l][kt]: pn = l * NJ + kt
#pragma unroll
        for (int i = 0; i < NPW; ++i) {
            const int piece = min(w + 4 * i, NJ - 1);
            stg[i] = *reinterpret_cast<const v4i*>(src + (int64_t)piece * 1024 + lane * 16);
        }
    };
    auto stash = [&](int pn) {                            // registers -> ring slot of panel pn
        uint8_t* dst = lds + (pn % NRING) * (NJ * 1024);
#pragma unroll
        for (int i = 0; i < NPW; ++i) {
            const int piece = min(w + 4 * i, NJ - 1);
            *reinterpret_cast<v4i*>(dst + piece * 1024 + lane * 16) = stg[i];
        }
    };
    __syncthreads();                                       // every wave has read its fragments back: the staging area becomes the ring
    fetch(0); stash(0);
    fetch(1); stash(1);
    fetch(2);
    int kt = 0;
    double sh = 1.0;
#pragma unroll 1
    for (int pn = 0; pn < npanel; ++pn) {
        __syncthreads();                                   // panel pn is complete; panel pn - 2's slot is free
        // panel pn + 2 was fetched a whole panel ago (its L2 round trip is over); panel pn + 3's fetch gets the same lead
        if (pn + 2 < npanel) stash(pn + 2);
        fetch(pn + 3);
        if constexpr (NB > 0) {
            const uint8_t* pan = lds + (pn % NRING) * (NJ * 1024) + lane * 16;
            v16i acc[NB];
#pragma unroll
            for (int b = 0; b < NB; ++b)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[b][e] = 0;
            // A fragments three reads ahead of their MFMAs (left to itself the compiler reads each one into the same registers right
            // before its use: an LDS round trip per pair of MFMAs)
            constexpr int AD = QN < 3 ? QN : 3;
            v4i af[AD];
            const uint32_t pa = (uint32_t)(uintptr_t)pan;
            static_for<AD>([&](auto ic) { af[decltype(ic)::value] = gram_rd<(J0 + decltype(ic)::value) * 1024>(pa); });
            static_for<QN>([&](auto jtc) {
                constexpr int ji = decltype(jtc)::value, jt = J0 + ji;
                constexpr int left = (QN - 1 - ji) < (AD - 1) ? (QN - 1 - ji) : (AD - 1);     // reads younger than fragment ji
                gram_wait<left>(af[ji % AD]);
                static_for<NB>([&](auto bc) {
                    constexpr int b = decltype(bc)::value;
                    acc[b] = __builtin_amdgcn_mfma_i32_32x32x32_i8(af[ji % AD], bf[b][jt], acc[b], 0, 0, 0);
                });
                if constexpr (ji + AD < QN) {
                    // (TIE: accumulators in VGPRs -- naming them keeps the read behind the MFMAs that use its destination; with the
                    // accumulators in AGPRs the tie would cost a copy out and back per read)
                    if constexpr (TIE) af[ji % AD] = gram_rd_after<(jt + AD) * 1024>(pa, acc[NB - 1]);
                    else af[ji % AD] = gram_rd<(jt + AD) * 1024>(pa);
                }
            });
            // row dot: accumulator element e = G row 32 kt + 16 h + e = byte e of this lane's own fragment kt (kt is wave-uniform:
            // a branch picks the registers)
            static_for<NB>([&](auto bc) {
                constexpr int b = decltype(bc)::value;
                v4i wv = bf[b][0];
                switch (kt) {
#define GRAM_CASE(i_) case i_: if constexpr ((i_) < NJ) wv = bf[b][(i_) < NJ ? (i_) : 0]; break;
                    GRAM_CASE(1) GRAM_CASE(2) GRAM_CASE(3) GRAM_CASE(4) GRAM_CASE(5) GRAM_CASE(6) GRAM_CASE(7) GRAM_CASE(8)
                    GRAM_CASE(9) GRAM_CASE(10) GRAM_CASE(11) GRAM_CASE(12) GRAM_CASE(13) GRAM_CASE(14) GRAM_CASE(15) GRAM_CASE(16)
                    GRAM_CASE(17) GRAM_CASE(18) GRAM_CASE(19) GRAM_CASE(20) GRAM_CASE(21) GRAM_CASE(22) GRAM_CASE(23)
#undef GRAM_CASE
                    default: break;
                }
                constexpr int GRP = BIG ? 4 : 16;
#pragma unroll
                for (int g0 = 0; g0 < 16; g0 += GRP) {
                    int ts = 0;
#pragma unroll
                    for (int e = g0; e < g0 + GRP; ++e) {
                        const int we = __builtin_amdgcn_sbfe(wv[e >> 2], 8 * (e & 3), 8);
                        ts += __mul24(acc[b][e], we);
                    }
                    quad[b] += (double)ts * sh;
                }
            });
        }
        if (++kt == NJ) { kt = 0; sh *= 256.0; }
    }
    GRAM_STAMP(3);

    // ---- scores
    if constexpr (CO) {
        // the four waves' shares of (quad, lin) of the same candidates: added in wave order by wave 0 (fixed order: reproducible)
        double* red = reinterpret_cast<double*>(lds);       // [wave][block][2][64 lanes]
        __syncthreads();                                    // the ring is dead
        static_for<NB>([&](auto bc) {
            constexpr int b = decltype(bc)::value;
            red[((w * NB + b) * 2 + 0) * 64 + lane] = quad[b];
            red[((w * NB + b) * 2 + 1) * 64 + lane] = lin[b];
        });
        __syncthreads();
        if (w == 0) {
            static_for<NB>([&](auto bc) {
                constexpr int b = decltype(bc)::value;
                double q = 0.0, l = 0.0;
#pragma unroll
                for (int ww = 0; ww < 4; ++ww) { q += red[((ww * NB + b) * 2 + 0) * 64 + lane]; l += red[((ww * NB + b) * 2 + 1) * 64 + lane]; }
                quad[b] = q; lin[b] = l;
            });
        }
    }
    if (!CO || w == 0) {
        static_for<NB>([&](auto bc) {
            constexpr int b = decltype(bc)::value;
            const double q = quad[b] + __shfl_xor(quad[b], 32);
            const int o = orow[b];
            const double s = (double)sig[b];
            const double tot = p.s0[o] - 2.0 * s * (lin[b] * p.cscl[o]) + s * s * q;
            const int blk = blk0 + b;
            const int cand = (blk - o * PB) * 32 + c;
            if (h == 0) p.scores[(int64_t)cand * p.O + o] = (float)(-p.norm * tot);
        });
    }
    GRAM_STAMP(4);
#undef GRAM_STAMP
}

template <int NJ, int CB, bool BIG, int WGS_PER_CU>
__global__ __launch_bounds__(256, WGS_PER_CU) void k_gram_score(GramScoreArgs p) {
#if defined(__HIP_DEVICE_COMPILE__)
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];      // [4][NJ][64][16]
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nwg = gridDim.x;
    const int b_begin = (int)(((int64_t)p.nblk * blockIdx.x) / nwg), b_end = (int)(((int64_t)p.nblk * (blockIdx.x + 1)) / nwg);
    const float sa = p.sa[0];
    int pass_no = 0;
    for (int pass0 = b_begin; pass0 < b_end; pass0 += 4 * CB, ++pass_no) {
        const int n = min(4 * CB, b_end - pass0);
        const int slot = pass_no < 2 ? (int)blockIdx.x * 2 + pass_no : -1;
        // (measured: pays from NJ = 8 -- below, a panel is so little work that the per-panel barrier dominates either way and the final
        // cross-wave sum only adds to it: swin stage 0, K = 128, 12.4 -> 17.7 us; NJ = 16 sits at the register limit of two
        // workgroups per CU already)
        if constexpr (NJ % 4 == 0 && NJ >= 8 && NJ != 16) {
            if (n <= CB) {                                 // the thin tail: all four waves on the same n blocks, K split four ways
#define GRAM_COOP(NBV)                                                                                            \
                do {                                                                                              \
                    if (w == 0) gram_pass<NJ, NBV, BIG, WGS_PER_CU == 2, 0>(p, lds, pass0, w, lane, sa, slot);     \
                    else if (w == 1) gram_pass<NJ, NBV, BIG, WGS_PER_CU == 2, 1>(p, lds, pass0, w, lane, sa, slot); \
                    else if (w == 2) gram_pass<NJ, NBV, BIG, WGS_PER_CU == 2, 2>(p, lds, pass0, w, lane, sa, slot); \
                    else gram_pass<NJ, NBV, BIG, WGS_PER_CU == 2, 3>(p, lds, pass0, w, lane, sa, slot);            \
                } while (0)
                if (n == 1) GRAM_COOP(1);
                else if (CB >= 2 && n == 2) GRAM_COOP((CB >= 2 ? 2 : 1));
                else if (CB >= 3 && n == 3) GRAM_COOP((CB >= 3 ? 3 : 1));
                else if (CB >= 4 && n == 4) GRAM_COOP((CB >= 4 ? 4 : 1));
#undef GRAM_COOP
                continue;
            }
        }
        const int nbw = n / 4 + (w < (n & 3) ? 1 : 0);     // this wave's blocks (wave-uniform)
        const int blk0 = pass0 + w * (n / 4) + min(w, n & 3);
        if (nbw == CB) gram_pass<NJ, CB, BIG, WGS_PER_CU == 2>(p, lds, blk0, w, lane, sa, slot);
        else if (nbw == 0) gram_pass<NJ, 0, BIG, WGS_PER_CU == 2>(p, lds, blk0, w, lane, sa, slot);
        else if (nbw == 1) gram_pass<NJ, 1, BIG, WGS_PER_CU == 2>(p, lds, blk0, w, lane, sa, slot);
        else if (CB > 2 && nbw == 2) gram_pass<NJ, (CB > 2 ? 2 : 1), BIG, WGS_PER_CU == 2>(p, lds, blk0, w, lane, sa, slot);
        else if (CB > 3 && nbw == 3) gram_pass<NJ, (CB > 3 ? 3 : 1), BIG, WGS_PER_CU == 2>(p, lds, blk0, w, lane, sa, slot);
    }
#endif
}

// ------------------------------------------------------------------------------------------------ host side
struct GramPlan {
    int T, O, K, NJ, NL;
    int64_t Tp;
    int steps_total, sg_steps, sg_splits, sc_steps, sc_splits, r_chunk, r_nchunk;
    // byte offsets into the workspace
    int64_t off_gfrag, off_clim, off_s0, off_cscl, off_xt, off_rl, off_gpart, off_cpart, off_amax, off_s0p, total;
    bool ok;
};

static int g_limbs(int64_t T, int a_bits) {
    const double gmax = (double)T * (double)((1 << a_bits) - 1) * (double)((1 << a_bits) - 1);
    double cover = 127.0;                                   // largest magnitude of L balanced limbs: 127 * (256^L - 1) / 255
    for (int L = 1; L <= 7; ++L) {
        if (gmax <= cover) return L;
        cover = cover * 256.0 + 127.0;
    }
    return 8;
}

// ---- image-sharded build (several ranks hold disjoint token ranges): the integer sums are additive over tokens, so every rank forms
// its own G and c as int64, the ranks all-reduce them (exact), and the limbs are cut from the GLOBAL sums.
// gsum [K][K] = sum of the split partials (plain row-major)
__global__ __launch_bounds__(256) void k_gram_sum_g(const int* __restrict__ part, int splits, int K, long long* __restrict__ gsum) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (int64_t)K * K) return;
    long long g = 0;
    for (int z = 0; z < splits; ++z) g += (long long)part[(int64_t)z * K * K + idx];
    gsum[idx] = g;
}

// csum [O][K] = the recombined limb products; s0 [O] = the chunks' shares in a fixed order
__global__ __launch_bounds__(256) void k_gram_sum_c(const int* __restrict__ part, int splits, int O, int K, long long* __restrict__ csum,
                                                    const double* __restrict__ s0p, int nchunk, double* __restrict__ s0) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx < O) {
        double t = 0.0;
        for (int ch = 0; ch < nchunk; ++ch) t += s0p[idx * nchunk + ch];
        s0[idx] = t;
    }
    if (idx >= (int64_t)O * K) return;
    const int o = (int)(idx / K), k = (int)(idx % K);
    long long c = 0;
    for (int z = 0; z < splits; ++z)
#pragma unroll
        for (int l = 0; l < RLIMBS; ++l) c += ((long long)part[((int64_t)z * RLIMBS * O + (int64_t)l * O + o) * K + k]) << (8 * l);
    csum[idx] = c;
}

// global sums -> the score kernel's operands: G limbs in fragment order (as k_gram_fin_g), c limbs (as k_gram_fin_c), S0, 2^-e
__global__ __launch_bounds__(256) void k_gram_limbs_g(const long long* __restrict__ gsum, int K, int NL, int8_t* __restrict__ gfrag) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (int64_t)K * K) return;
    const int col = (int)(idx % K), slot = (int)(idx / K);
    const int kt = slot >> 5, rp = slot & 31;
    const int row = 32 * kt + 16 * ((rp >> 2) & 1) + 4 * (rp >> 3) + (rp & 3);
    const int nj = K >> 5, jt = col >> 5, h = (col >> 4) & 1, e = col & 15;
    long long rest = gsum[(int64_t)row * K + col];
    for (int l = 0; l < NL; ++l) {
        const int d = (int)(int8_t)(rest & 0xff);
        gfrag[((((int64_t)l * nj + kt) * nj + jt) * 64 + (rp + 32 * h)) * 16 + e] = (int8_t)d;
        rest = (rest - d) >> 8;
    }
}

__global__ __launch_bounds__(256) void k_gram_limbs_c(const long long* __restrict__ csum, int O, int K, int8_t* __restrict__ clim,
                                                      const double* __restrict__ s0_in, const unsigned int* __restrict__ amax,
                                                      double* __restrict__ s0, double* __restrict__ cscl) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx < O) {
        s0[idx] = s0_in[idx];
        const float am = __uint_as_float(amax[idx]);
        int e = 0;
        if (am > 0.0f && am < 3.0e38f) e = RFIX_BITS - ilogbf(am);           // (the exponent k_gram_rfix used)
        cscl[idx] = ldexp(1.0, -e);
    }
    if (idx >= (int64_t)O * K) return;
    const int o = (int)(idx / K), k = (int)(idx % K);
    long long rest = csum[idx];
#pragma unroll
    for (int l = 0; l < CLIMBS; ++l) {
        const int d = (int)(int8_t)(rest & 0xff);
        clim[((int64_t)o * CLIMBS + l) * K + k] = (int8_t)d;
        rest = (rest - d) >> 8;
    }
}

static int64_t al256(int64_t v) { return (v + 255) / 256 * 256; }

static int pick_steps(int steps_total, int tiles, int target_waves) {
    // token steps (of 128) per split: enough splits to fill the chip with 128 x 128 workgroup tiles, at least 4 steps per split (the partials are
    // summed afterwards), at most 512 (int32 accumulators)
    int splits = (target_waves + tiles - 1) / tiles;
    if (splits < 1) splits = 1;
    int per = (steps_total + splits - 1) / splits;
    if (per < 4) per = 4;
    if (per > 512) per = 512;
    if (per > steps_total) per = steps_total;
    return per;
}

constexpr int MM_BI = 4, MM_BJ = 4;                         // k_i8mm workgroup tile: 128 x 128

static GramPlan gram_plan(int T, int O, int K, int a_bits) {
    GramPlan g{};
    g.T = T; g.O = O; g.K = K;
    g.ok = T >= 1 && O >= 1 && K >= 32 && K % 32 == 0 && a_bits >= 2 && a_bits <= 7 && (int64_t)RLIMBS * O < ((int64_t)1 << 30);
    if (!g.ok) return g;
    g.NJ = K / 32;
    g.NL = g_limbs(T, a_bits);
    g.Tp = ((int64_t)T + 127) / 128 * 128;
    g.steps_total = (int)(g.Tp / 128);
    const int tk = (K + 32 * MM_BJ - 1) / (32 * MM_BJ);
    const int tg = ((K + 32 * MM_BI - 1) / (32 * MM_BI)) * tk, tc = ((RLIMBS * O + 32 * MM_BI - 1) / (32 * MM_BI)) * tk;
    g.sg_steps = pick_steps(g.steps_total, tg, 512);
    g.sg_splits = (g.steps_total + g.sg_steps - 1) / g.sg_steps;
    g.sc_steps = pick_steps(g.steps_total, tc, 512);
    g.sc_splits = (g.steps_total + g.sc_steps - 1) / g.sc_steps;
    g.r_chunk = 8192;                                       // tokens per k_gram_rfix workgroup (a multiple of 128)
    g.r_nchunk = (int)((g.Tp + g.r_chunk - 1) / g.r_chunk);
    int64_t off = 0;
    g.off_gfrag = off; off += al256((int64_t)g.NL * K * K);
    g.off_clim = off; off += al256((int64_t)O * CLIMBS * K);
    g.off_s0 = off; off += al256((int64_t)O * 8);
    g.off_cscl = off; off += al256((int64_t)O * 8);
    g.off_amax = off; off += al256((int64_t)O * 4);
    g.off_s0p = off; off += al256((int64_t)O * g.r_nchunk * 8);
    g.off_xt = off; off += al256((int64_t)K * g.Tp);
    g.off_rl = off; off += al256((int64_t)RLIMBS * O * g.Tp);
    g.off_gpart = off; off += al256((int64_t)g.sg_splits * K * K * 4);
    g.off_cpart = off; off += al256((int64_t)g.sc_splits * RLIMBS * O * K * 4);
    g.total = off;
    return g;
}

static bool nj_supported(int nj) { return nj == 1 || nj == 2 || nj == 3 || nj == 4 || nj == 6 || nj == 8 || nj == 12 || nj == 16 || nj == 24; }

}  // namespace

// adalog_gram_supported: the Gram form can score this weight search (K a multiple of 32 with an instantiated kernel, <= 7-bit
// operands, P a multiple of 32).  adalog_gram_ok: ... and it pays: at most half the multiply-adds of the token form
// (limbs * K <= T / 2).
extern "C" int adalog_gram_supported(int T, int O, int K, int a_bits, int w_bits, int P) {
    const GramPlan g = gram_plan(T, O, K, a_bits);
    if (!g.ok || !nj_supported(g.NJ) || w_bits < 2 || w_bits > 7 || P < 32 || P % 32 != 0) return 0;
    if ((int64_t)128 * ((1 << w_bits) - 1) * K >= (1 << 23)) return 0;           // accumulators must fit 24 bits (v_mul_i32_i24)
    if ((int64_t)O * P / 32 >= ((int64_t)1 << 30)) return 0;
    return 1;
}

extern "C" int adalog_gram_ok(int T, int O, int K, int a_bits, int w_bits, int P) {
    if (!adalog_gram_supported(T, O, K, a_bits, w_bits, P)) return 0;
    return (int64_t)g_limbs(T, a_bits) * K * 2 <= T ? 1 : 0;
}

/* int8 limbs of G = X_int^T X_int for T tokens of an a_bits-bit activation (what adalog_gram_score_w issues per candidate row:
 * limbs * K^2 + 32 K multiply-adds -- measurement aid for the roofline accounting) */
extern "C" int adalog_gram_limbs(int T, int a_bits) { return (T >= 1 && a_bits >= 2 && a_bits <= 7) ? g_limbs(T, a_bits) : -1; }

extern "C" int64_t adalog_gram_workspace_bytes(int T, int O, int K, int a_bits) {
    const GramPlan g = gram_plan(T, O, K, a_bits);
    return g.ok ? g.total : -1;
}

/* Built once per weight_fpcs call (reference linear.py:483-503 re-enters _search_best_w_scale six times with the same activation
 * quantiser).  x [T][ldx] fp32 (K valid), sa / za: the activation quantiser's scale / zero point (device scalars), ref_t = raw_out
 * TRANSPOSED [O][T], bias [O] or null. */
extern "C" int adalog_gram_build(const float* x, int T, int K, int64_t ldx, const float* sa, const float* za, int a_bits,
                                 const float* ref_t, int O, const float* bias, void* ws, int64_t ws_bytes, void* stream) {
    ADALOG_ARG_CHECK(x && sa && za && ref_t && ws, "gram_build: null pointer");
    const GramPlan g = gram_plan(T, O, K, a_bits);
    ADALOG_ARG_CHECK(g.ok && nj_supported(g.NJ), "gram_build: shape not supported (adalog_gram_supported)");
    ADALOG_ARG_CHECK(ws_bytes >= g.total, "gram_build: workspace too small");
    ADALOG_ARG_CHECK(ldx >= K && ldx % 4 == 0 && ((uintptr_t)x & 15) == 0 && ((uintptr_t)ws & 255) == 0, "gram_build: x rows / workspace must be aligned");
    hipStream_t st = (hipStream_t)stream;
    uint8_t* base = (uint8_t*)ws;
    int8_t* xt = (int8_t*)(base + g.off_xt);
    int8_t* rl = (int8_t*)(base + g.off_rl);
    int* gpart = (int*)(base + g.off_gpart);
    int* cpart = (int*)(base + g.off_cpart);
    unsigned int* amax = (unsigned int*)(base + g.off_amax);
    double* s0p = (double*)(base + g.off_s0p);
    {
        const hipError_t me = hipMemsetAsync(amax, 0, (size_t)O * 4, st);
        if (me != hipSuccess) { adalog_set_error("adalog_gram_build (clear)", me); return (int)me; }
    }
    hipLaunchKernelGGL(k_gram_pack_xt, dim3((unsigned)(g.Tp / 64), (unsigned)((K + 63) / 64)), dim3(256), 0, st, x, T, K, ldx, sa, za,
                       (float)((1 << a_bits) - 1), xt, g.Tp);
    hipLaunchKernelGGL(k_gram_rmax, dim3((unsigned)O, (unsigned)g.r_nchunk), dim3(256), 0, st, ref_t, T, g.r_chunk, bias, amax);
    hipLaunchKernelGGL(k_gram_rfix, dim3((unsigned)O, (unsigned)g.r_nchunk), dim3(256), 0, st, ref_t, T, g.Tp, O, g.r_chunk, g.r_nchunk, bias,
                       amax, rl, s0p, (double*)(base + g.off_cscl));
    const unsigned tk = (unsigned)((K + 32 * MM_BJ - 1) / (32 * MM_BJ));
    hipLaunchKernelGGL(k_i8mm, dim3((unsigned)((K + 32 * MM_BI - 1) / (32 * MM_BI)), tk, (unsigned)g.sg_splits), dim3(256), 0, st,
                       xt, xt, K, K, g.Tp, g.sg_steps, g.steps_total, gpart);
    hipLaunchKernelGGL(k_i8mm, dim3((unsigned)((RLIMBS * O + 32 * MM_BI - 1) / (32 * MM_BI)), tk, (unsigned)g.sc_splits), dim3(256), 0,
                       st, rl, xt, RLIMBS * O, K, g.Tp, g.sc_steps, g.steps_total, cpart);
    hipLaunchKernelGGL(k_gram_fin_g, dim3((unsigned)(((int64_t)K * K + 255) / 256)), dim3(256), 0, st, gpart, g.sg_splits, K, g.NL,
                       (int8_t*)(base + g.off_gfrag));
    hipLaunchKernelGGL(k_gram_fin_c, dim3((unsigned)(((int64_t)O * K + 255) / 256)), dim3(256), 0, st, cpart, g.sc_splits, O, K,
                       (int8_t*)(base + g.off_clim), s0p, g.r_nchunk, (double*)(base + g.off_s0));
    ADALOG_LAUNCH_CHECK("adalog_gram_build");
    return 0;
}

/* ---- the same build for calibration images SHARDED over ranks, in three calls with the collectives between them:
 *   adalog_gram_amax        amax[o] = bits of max_t |raw_out[t][o] - bias[o]| over THIS rank's tokens  -> all-reduce MAX (as int32)
 *   adalog_gram_build_sums  with the GLOBAL amax: gsum [K][K], csum [O][K] (int64) and s0 [O] (fp64) of this rank's tokens
 *                           -> all-reduce SUM of the three (integers: exact; s0: the same bits on every rank)
 *   adalog_gram_build_from_sums  -> the workspace adalog_gram_score_w reads, for T = the GLOBAL token count (limbs of G sized for it)
 * After that every rank scores the SAME final scores from the same state: an FPCS step needs no collective at all (six score
 * all-reduces per weight_fpcs call before), and adalog_gram_ok is asked with the global token count. */
extern "C" int adalog_gram_amax(const float* ref_t, int T, int O, const float* bias, unsigned int* amax, void* stream) {
    ADALOG_ARG_CHECK(ref_t && amax && T >= 1 && O >= 1, "gram_amax: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    const hipError_t me = hipMemsetAsync(amax, 0, (size_t)O * 4, st);
    if (me != hipSuccess) { adalog_set_error("adalog_gram_amax (clear)", me); return (int)me; }
    const int chunk = 8192, nchunk = (T + chunk - 1) / chunk;
    hipLaunchKernelGGL(k_gram_rmax, dim3((unsigned)O, (unsigned)nchunk), dim3(256), 0, st, ref_t, T, chunk, bias, amax);
    ADALOG_LAUNCH_CHECK("adalog_gram_amax");
    return 0;
}

extern "C" int adalog_gram_build_sums(const float* x, int T, int K, int64_t ldx, const float* sa, const float* za, int a_bits,
                                      const float* ref_t, int O, const float* bias, const unsigned int* amax, long long* gsum,
                                      long long* csum, double* s0, void* ws, int64_t ws_bytes, void* stream) {
    ADALOG_ARG_CHECK(x && sa && za && ref_t && amax && gsum && csum && s0 && ws, "gram_build_sums: null pointer");
    const GramPlan g = gram_plan(T, O, K, a_bits);
    ADALOG_ARG_CHECK(g.ok, "gram_build_sums: shape not supported");
    ADALOG_ARG_CHECK(ws_bytes >= g.total, "gram_build_sums: workspace too small (adalog_gram_workspace_bytes of the LOCAL shape)");
    ADALOG_ARG_CHECK(ldx >= K && ldx % 4 == 0 && ((uintptr_t)x & 15) == 0 && ((uintptr_t)ws & 255) == 0, "gram_build_sums: x rows / workspace must be aligned");
    hipStream_t st = (hipStream_t)stream;
    uint8_t* base = (uint8_t*)ws;
    int8_t* xt = (int8_t*)(base + g.off_xt);
    int8_t* rl = (int8_t*)(base + g.off_rl);
    int* gpart = (int*)(base + g.off_gpart);
    int* cpart = (int*)(base + g.off_cpart);
    double* s0p = (double*)(base + g.off_s0p);
    hipLaunchKernelGGL(k_gram_pack_xt, dim3((unsigned)(g.Tp / 64), (unsigned)((K + 63) / 64)), dim3(256), 0, st, x, T, K, ldx, sa, za,
                       (float)((1 << a_bits) - 1), xt, g.Tp);
    hipLaunchKernelGGL(k_gram_rfix, dim3((unsigned)O, (unsigned)g.r_nchunk), dim3(256), 0, st, ref_t, T, g.Tp, O, g.r_chunk, g.r_nchunk, bias,
                       amax, rl, s0p, (double*)(base + g.off_cscl));
    const unsigned tk = (unsigned)((K + 32 * MM_BJ - 1) / (32 * MM_BJ));
    hipLaunchKernelGGL(k_i8mm, dim3((unsigned)((K + 32 * MM_BI - 1) / (32 * MM_BI)), tk, (unsigned)g.sg_splits), dim3(256), 0, st,
                       xt, xt, K, K, g.Tp, g.sg_steps, g.steps_total, gpart);
    hipLaunchKernelGGL(k_i8mm, dim3((unsigned)((RLIMBS * O + 32 * MM_BI - 1) / (32 * MM_BI)), tk, (unsigned)g.sc_splits), dim3(256), 0,
                       st, rl, xt, RLIMBS * O, K, g.Tp, g.sc_steps, g.steps_total, cpart);
    hipLaunchKernelGGL(k_gram_sum_g, dim3((unsigned)(((int64_t)K * K + 255) / 256)), dim3(256), 0, st, gpart, g.sg_splits, K, gsum);
    hipLaunchKernelGGL(k_gram_sum_c, dim3((unsigned)(((int64_t)O * K + 255) / 256)), dim3(256), 0, st, cpart, g.sc_splits, O, K, csum,
                       s0p, g.r_nchunk, s0);
    ADALOG_LAUNCH_CHECK("adalog_gram_build_sums");
    return 0;
}

extern "C" int adalog_gram_build_from_sums(const long long* gsum, const long long* csum, const double* s0, const unsigned int* amax,
                                           int T_total, int O, int K, int a_bits, void* ws, int64_t ws_bytes, void* stream) {
    ADALOG_ARG_CHECK(gsum && csum && s0 && amax && ws, "gram_build_from_sums: null pointer");
    const GramPlan g = gram_plan(T_total, O, K, a_bits);
    ADALOG_ARG_CHECK(g.ok && nj_supported(g.NJ), "gram_build_from_sums: shape not supported (adalog_gram_supported)");
    ADALOG_ARG_CHECK(ws_bytes >= g.total && ((uintptr_t)ws & 255) == 0, "gram_build_from_sums: workspace too small / misaligned");
    hipStream_t st = (hipStream_t)stream;
    uint8_t* base = (uint8_t*)ws;
    hipLaunchKernelGGL(k_gram_limbs_g, dim3((unsigned)(((int64_t)K * K + 255) / 256)), dim3(256), 0, st, gsum, K, g.NL,
                       (int8_t*)(base + g.off_gfrag));
    hipLaunchKernelGGL(k_gram_limbs_c, dim3((unsigned)(((int64_t)O * K + 255) / 256)), dim3(256), 0, st, csum, O, K,
                       (int8_t*)(base + g.off_clim), s0, amax, (double*)(base + g.off_s0), (double*)(base + g.off_cscl));
    ADALOG_LAUNCH_CHECK("adalog_gram_build_from_sums");
    return 0;
}

static long long* g_gram_timeline = nullptr;
// lab only: cycle stamps [workgroup][pass < 2][8] of the next adalog_gram_score_w launches (null switches them off)
extern "C" void adalog_gram_set_timeline(long long* buf) { g_gram_timeline = buf; }

static int device_cus_gram() { return adalog_device_cus(); }   // per device ordinal (common.h)

/* One FPCS step: scores [P][O] = -norm * sum_t (raw_out - b - q_a(x) . fq_p(W)^T)^2 for the P candidates (scale, zp) [P][O] of every
 * output row, from the workspace adalog_gram_build left.  W fp32 [O][ldw].  (T, O, K, a_bits) must be those of the build. */
extern "C" int adalog_gram_score_w(const float* W, int O, int K, int64_t ldw, const float* scale, const float* zp, int P, int w_bits,
                                   const void* ws, int T, int a_bits, const float* sa, double norm, float* scores, void* stream) {
    ADALOG_ARG_CHECK(W && scale && zp && ws && sa && scores, "gram_score_w: null pointer");
    ADALOG_ARG_CHECK(adalog_gram_supported(T, O, K, a_bits, w_bits, P), "gram_score_w: shape not supported (adalog_gram_supported)");
    ADALOG_ARG_CHECK(ldw >= K && ldw % 4 == 0 && ((uintptr_t)W & 15) == 0, "gram_score_w: weight rows must be 16-byte aligned");
    const GramPlan g = gram_plan(T, O, K, a_bits);
    ADALOG_ARG_CHECK(g.ok && nj_supported(g.NJ) && P % 32 == 0, "gram_score_w: shape not supported");
    const uint8_t* base = (const uint8_t*)ws;
    GramScoreArgs a{};
    a.W = W; a.ldw = ldw; a.O = O; a.K = K; a.P = P; a.scale = scale; a.zp = zp;
    a.gfrag = (const int8_t*)(base + g.off_gfrag); a.NL = g.NL; a.clim = (const int8_t*)(base + g.off_clim);
    a.s0 = (const double*)(base + g.off_s0); a.cscl = (const double*)(base + g.off_cscl);
    a.sa = sa; a.norm = norm; a.scores = scores;
    a.qmax = (float)((1 << w_bits) - 1);
    const float zone = 6e-7f * (float)(1 << w_bits);
    a.tie = 0.5f - (zone > 1e-5f ? zone : 1e-5f);
    a.nblk = (int)((int64_t)O * P / 32);
    a.timeline = g_gram_timeline;
    const bool big = w_bits > 4;
    hipStream_t st = (hipStream_t)stream;
    const int ncu = device_cus_gram();
#define GRAM_LAUNCH(NJV, CBV, BIGV, WPC)                                                                          \
    do {                                                                                                          \
        /* every workgroup slot of the chip gets blocks (a workgroup with fewer than 4 CBV blocks runs one thin pass) */ \
        int wgs = a.nblk < WPC * ncu ? a.nblk : WPC * ncu;                                                        \
        const size_t shm = (size_t)4 * NJV * 1024;                                                                \
        static unsigned long long attr_dev = 0; \
        { hipError_t ea__ = adalog_max_lds(reinterpret_cast<const void*>(&k_gram_score<NJV, CBV, BIGV, WPC>), (int)(160 * 1024), &attr_dev); \
          if (ea__ != hipSuccess) { adalog_set_error("hipFuncSetAttribute", ea__); return (int)ea__; } } \
        adalog_note_kernel("k_gram_score<i8>");                                                                   \
        hipLaunchKernelGGL((k_gram_score<NJV, CBV, BIGV, WPC>), dim3((unsigned)wgs), dim3(256), shm, st, a);      \
    } while (0)
#define GRAM_LAUNCH_B(NJV, CBV, WPC) do { if (big) GRAM_LAUNCH(NJV, CBV, true, WPC); else GRAM_LAUNCH(NJV, CBV, false, WPC); } while (0)
    switch (g.NJ) {
        case 1: GRAM_LAUNCH_B(1, 4, 2); break;
        case 2: GRAM_LAUNCH_B(2, 4, 2); break;
        case 3: GRAM_LAUNCH_B(3, 4, 2); break;
        case 4: GRAM_LAUNCH_B(4, 4, 2); break;
        case 6: GRAM_LAUNCH_B(6, 4, 2); break;
        case 8: GRAM_LAUNCH_B(8, 2, 2); break;
        case 12: GRAM_LAUNCH_B(12, 2, 2); break;
        case 16: GRAM_LAUNCH_B(16, 2, 2); break;
        case 24: GRAM_LAUNCH_B(24, 2, 1); break;
        default: ADALOG_ARG_CHECK(false, "gram_score_w: K not instantiated");
    }
#undef GRAM_LAUNCH_B
#undef GRAM_LAUNCH
    ADALOG_LAUNCH_CHECK("adalog_gram_score_w");
    return 0;
}

extern "C" int adalog_topk_next_tail(const float* scores, int P, int cols, const adalog_fpcs_tail* tail, int* idx_out, void* stream);

// adalog_gram_score_w followed by the FPCS step's tail (`tail` may be null).  The tail is a SECOND launch here (adalog_topk_next_tail):
// run inside k_gram_score -- a ticket per output row, the wave / workgroup storing a row's last block ranks it -- it cost 25 us per
// launch (agent-scope ticket round trips of ~2 us per pass, then ~2 rows of tails at the end of every workgroup) against the 11 us of
// the separate launch (round 6, same-box A/B 875 against 852 ms per calibration); a workgroup per complete row would unbalance the
// 9-blocks-per-workgroup split.
extern "C" int adalog_gram_score_w_tail(const float* W, int O, int K, int64_t ldw, const float* scale, const float* zp, int P, int w_bits,
                                        const void* ws, int T, int a_bits, const float* sa, double norm, float* scores,
                                        const adalog_fpcs_tail* tail, void* stream) {
    const char* why = fpcs::tail_problem(tail, P);
    ADALOG_ARG_CHECK(why == nullptr, why);
    const int rc = adalog_gram_score_w(W, O, K, ldw, scale, zp, P, w_bits, ws, T, a_bits, sa, norm, scores, stream);
    if (rc || !tail) return rc;
    return adalog_topk_next_tail(scores, P, O, tail, nullptr, stream);
}
