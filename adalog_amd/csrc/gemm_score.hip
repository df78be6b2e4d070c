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
//              1 = bf16  (v_mfma_f32_32x32x16_bf16, AdaLog operand m*2^-t and integer operand exact in bf16)
//              2 = fp32  (v_mfma_f32_32x32x2_f32,   conv patch-embed with unquantised 8-bit input)
// Tiling: 128x128 output tile per 256-thread workgroup (4 waves as 2x2, each 64x64 = 2x2 MFMA 32x32 tiles),
// K-step 64 bytes, LDS double-buffered with a 16-byte-slot XOR swizzle (conflict-free ds_read_b128),
// XCD-aware block order so the workgroups sharing an A tile sit on one XCD's L2.
#include "common.h"
#include <stdlib.h>

// The quantisation kernels are compiled without FMA contraction (bin indices must round like the reference); this file
// holds no bin-defining arithmetic, so the epilogues may fuse multiply-adds.
#pragma clang fp contract(fast)

namespace {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef __bf16 v8bf __attribute__((ext_vector_type(8)));

constexpr int BM = 128, BN = 128, BKB = 64;   // BKB: K-step in bytes

struct GemmArgs {
    const uint8_t* A; const uint8_t* B;
    int64_t sAc, sAg, sBc, sBg;      // byte strides between candidates / groups (0 = shared operand)
    int M, N; int64_t Kb;            // Kb = padded K in bytes (multiple of 64)
    int C, G, gmod;
    const float* ref; int64_t ldr, sRg, ref_cs; int ref_div;
    const float* sa; int64_t sa_c, sa_g;
    const float* sb; int64_t sb_c, sb_g, sb_n;
    const float* bias; int64_t bi_c, bi_g, bi_n;
    const float* row_scale; const float* row_bias;   // optional per-ROW factor / offset (transposed activation searches)
    float* partial; int MT, NT, Npad;
    float* out; int64_t ldo, sOc, sOg;
    float sa_mul;                    // constant folded into sa (e.g. 1/(4L-2) for AdaLog numerators)
    int reduce_cols;                 // 1: one partial per tile (sum over its columns) instead of one per column
    int order;                       // tile order (fastest index first): 0 = nt,mt,g,c  1 = nt,c,mt,g  2 = mt,nt,c,g
};

__device__ __forceinline__ int swz(int row, int slot) { return row * BKB + ((slot ^ ((row >> 2) & 3)) << 4); }

template <int DT> struct Acc { typedef v16f type; };
template <> struct Acc<0> { typedef v16i type; };

template <int DT>
__device__ __forceinline__ void mma(const uint4& a, const uint4& b, typename Acc<DT>::type& c) {
    if constexpr (DT == 0) {
        c = __builtin_amdgcn_mfma_i32_32x32x32_i8(*reinterpret_cast<const v4i*>(&a), *reinterpret_cast<const v4i*>(&b), c, 0, 0, 0);
    } else if constexpr (DT == 1) {
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const v8bf*>(&a), *reinterpret_cast<const v8bf*>(&b), c, 0, 0, 0);
    } else {
        const float* af = reinterpret_cast<const float*>(&a);
        const float* bf = reinterpret_cast<const float*>(&b);
#pragma unroll
        for (int s = 0; s < 4; ++s) c = __builtin_amdgcn_mfma_f32_32x32x2f32(af[s], bf[s], c, 0, 0, 0);
    }
}

template <int DT, bool STORE>
__global__ __launch_bounds__(256, 2) void k_gemm_score(GemmArgs p) {
    __shared__ __attribute__((aligned(16))) uint8_t lds[2 * (BM + BN) * BKB];
    __shared__ float red[2][2][2][32];
    __shared__ float colv[128];
    uint8_t* As = lds;
    uint8_t* Bs = lds + 2 * BM * BKB;

    // ---- XCD-aware bijective block remap: consecutive logical tiles (same A tile, neighbouring n) share an XCD/L2
    const unsigned nwg = gridDim.x, bid = blockIdx.x;
    const unsigned q8 = nwg >> 3, r8 = nwg & 7, xcd = bid & 7;
    const unsigned lid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
    // Tile order = which operand stays hot in the XCD's 4 MiB L2 while its neighbours run:
    //   1 (activation / matmul searches): the n-tiles of one (candidate, m-tile) share the candidate's A tile, and all
    //     candidates of one m-tile share the fp32 reference rows and the fixed operand;
    //   2 (weight searches, columns = (out-channel, candidate)): all m-tiles of one column tile share its packed weights.
    unsigned t = lid;
    int nt, mt, g, c;
    if (p.order == 1) { nt = t % p.NT; t /= p.NT; c = t % p.C; t /= p.C; mt = t % p.MT; g = t / p.MT; }
    else if (p.order == 2) { mt = t % p.MT; t /= p.MT; nt = t % p.NT; t /= p.NT; c = t % p.C; g = t / p.C; }
    else { nt = t % p.NT; t /= p.NT; mt = t % p.MT; t /= p.MT; g = t % p.G; c = t / p.G; }
    const int gh = g % p.gmod;
    const int m0 = mt * BM, n0 = nt * BN;

    const uint8_t* Ag = p.A + c * p.sAc + g * p.sAg;
    const uint8_t* Bg = p.B + c * p.sBc + g * p.sBg;

    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, wr = w >> 1, wc = w & 1;
    // global->LDS staging: thread loads rows (tid>>2) and (tid>>2)+64, 16-byte slot tid&3
    const int lrow = tid >> 2, lslot = tid & 3;
    int ar0 = m0 + lrow, ar1 = m0 + lrow + 64, br0 = n0 + lrow, br1 = n0 + lrow + 64;
    ar0 = ar0 < p.M ? ar0 : p.M - 1; ar1 = ar1 < p.M ? ar1 : p.M - 1;      // edge rows: clamp, masked in epilogue
    br0 = br0 < p.N ? br0 : p.N - 1; br1 = br1 < p.N ? br1 : p.N - 1;
    const uint4* ga0 = reinterpret_cast<const uint4*>(Ag + (int64_t)ar0 * p.Kb) + lslot;
    const uint4* ga1 = reinterpret_cast<const uint4*>(Ag + (int64_t)ar1 * p.Kb) + lslot;
    const uint4* gb0 = reinterpret_cast<const uint4*>(Bg + (int64_t)br0 * p.Kb) + lslot;
    const uint4* gb1 = reinterpret_cast<const uint4*>(Bg + (int64_t)br1 * p.Kb) + lslot;
    const int so0 = swz(lrow, lslot), so1 = swz(lrow + 64, lslot);

    typename Acc<DT>::type acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0;

    const int nk = (int)(p.Kb / BKB);
    uint4 ra0 = ga0[0], ra1 = ga1[0], rb0 = gb0[0], rb1 = gb1[0];
    *reinterpret_cast<uint4*>(As + so0) = ra0; *reinterpret_cast<uint4*>(As + so1) = ra1;
    *reinterpret_cast<uint4*>(Bs + so0) = rb0; *reinterpret_cast<uint4*>(Bs + so1) = rb1;
    __syncthreads();

    const int frow = lane & 31, fkg = lane >> 5;
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) {
            const int o = (kt + 1) * (BKB / 16);
            ra0 = ga0[o]; ra1 = ga1[o]; rb0 = gb0[o]; rb1 = gb1[o];
        }
        const uint8_t* Ac = As + cur * BM * BKB;
        const uint8_t* Bc = Bs + cur * BN * BKB;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            uint4 af[2], bf[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) af[i] = *reinterpret_cast<const uint4*>(Ac + swz(wr * 64 + i * 32 + frow, ks * 2 + fkg));
#pragma unroll
            for (int j = 0; j < 2; ++j) bf[j] = *reinterpret_cast<const uint4*>(Bc + swz(wc * 64 + j * 32 + frow, ks * 2 + fkg));
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) mma<DT>(af[i], bf[j], acc[i][j]);
        }
        if (kt + 1 < nk) {
            uint8_t* An = As + (cur ^ 1) * BM * BKB;
            uint8_t* Bn = Bs + (cur ^ 1) * BN * BKB;
            *reinterpret_cast<uint4*>(An + so0) = ra0; *reinterpret_cast<uint4*>(An + so1) = ra1;
            *reinterpret_cast<uint4*>(Bn + so0) = rb0; *reinterpret_cast<uint4*>(Bn + so1) = rb1;
        }
        __syncthreads();
    }

    // ---- epilogue: out = acc * (sa*sb) + bias; squared error against ref; column sums over the tile's rows.
    // With ref_div > 1 a GEMM column encodes (output channel n = col / ref_div, candidate = col % ref_div): the 128
    // candidates of one channel sit in one tile and share ONE reference column (weight searches).
    // Branch-free: edge rows/columns are clamped for addressing and weighted 0; 32-bit offsets inside a group.
    const float* refg = p.ref ? p.ref + (int64_t)g * p.sRg : nullptr;
    float* outg = STORE ? p.out + (int64_t)c * p.sOc + (int64_t)g * p.sOg : nullptr;
    const int ldr = (int)p.ldr, rcs = (int)p.ref_cs, ldo = (int)p.ldo;
    const bool interior = (m0 + BM <= p.M) && (n0 + BN <= p.N);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int col = n0 + wc * 64 + j * 32 + frow;
        const bool cv = col < p.N;
        const int colc = cv ? col : p.N - 1;
        const int ci = p.ref_div > 1 ? colc % p.ref_div : c;
        const int ni = p.ref_div > 1 ? colc / p.ref_div : colc;
        const float alpha = p.sa[ci * p.sa_c + gh * p.sa_g] * p.sa_mul * p.sb[ci * p.sb_c + gh * p.sb_g + ni * p.sb_n];
        const float beta = p.bias ? p.bias[ci * p.bi_c + gh * p.bi_g + ni * p.bi_n] : 0.0f;
        const int rc0 = ni * rcs;
        float csum = 0.0f;
        if (interior) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int rb = m0 + wr * 64 + i * 32 + 4 * fkg;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = rb + (r & 3) + 8 * (r >> 2);
                    const float o = (float)acc[i][j][r] * alpha + beta;
                    if (STORE) outg[row * ldo + col] = o;
                    if (refg) {
                        const float e = refg[row * ldr + rc0] - o;
                        csum += e * e;
                    }
                }
            }
        } else {
            const float cm = cv ? 1.0f : 0.0f;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int rb = m0 + wr * 64 + i * 32 + 4 * fkg;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = rb + (r & 3) + 8 * (r >> 2);
                    const bool rv = row < p.M;
                    const int rowc = rv ? row : p.M - 1;
                    const float o = (float)acc[i][j][r] * alpha + beta;
                    if (STORE) { if (rv && cv) outg[rowc * ldo + col] = o; }
                    if (refg) {
                        const float e = refg[rowc * ldr + rc0] - o;
                        csum += (e * e) * (rv ? cm : 0.0f);
                    }
                }
            }
        }
        csum += __shfl_xor(csum, 32);
        if (fkg == 0) red[wr][wc][j][frow] = csum;
    }
    if (p.partial) {
        __syncthreads();
        float v = 0.0f;
        int col = 0;
        if (tid < 128) {
            const int cwc = tid >> 6, cj = (tid >> 5) & 1, cl = tid & 31;
            col = n0 + cwc * 64 + cj * 32 + cl;
            v = red[0][cwc][cj][cl] + red[1][cwc][cj][cl];
        }
        if (p.reduce_cols) {                       // fixed-order tile total: LDS, then one wave's xor tree
            if (tid < 128) colv[tid] = v;
            __syncthreads();
            if (tid < 64) {
                float t2 = colv[tid] + colv[tid + 64];
#pragma unroll
                for (int sft = 32; sft > 0; sft >>= 1) t2 += __shfl_xor(t2, sft);
                if (tid == 0) p.partial[(((int64_t)c * p.G + g) * p.MT + mt) * p.Npad + nt] = t2;
            }
        } else if (tid < 128) {
            if (p.ref_div > 1) {
                if (col < p.N)
                    p.partial[((((int64_t)g) * p.MT + mt) * p.Npad + col / p.ref_div) * p.ref_div + col % p.ref_div] = v;
            } else if (col < p.Npad) {
                p.partial[(((int64_t)c * p.G + g) * p.MT + mt) * p.Npad + col] = v;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------ large-tile kernel
// Second-generation scoring kernel, used whenever the candidates sit in the GEMM's column axis (every search) or there
// is a single problem per group (quant_forward).  Per 512-thread workgroup (8 waves as 2 x 4): (64*TM) x 256 output tile,
// K-step = 128 BYTES so that every staged row is one full 128-byte cache line (the 64-byte steps of the first kernel
// fetched each line twice and left it latency-bound: MFMA 13 % busy, 64 % of wave cycles parked, profiles/r01_pmc_*).
// Per wave (32*TM) x 64 = TM x 2 MFMA 32x32 tiles; operand bytes per MAC are half those of the 128 x 128 tile.
// One LDS stage (<= 64 KiB -> 2 workgroups per CU) + register prefetch of the next K-step.
constexpr int BN2 = 256, BK2 = 128;

__device__ __forceinline__ int swz2(int row, int slot) { return row * BK2 + ((slot ^ ((row >> 1) & 7)) << 4); }

// Epilogue of the large-tile kernel, specialised at compile time so the unrolled body is branch-free:
//   STORE: write out (quant_forward) / else: squared error against ref;  EDGE: tile touches the M or N boundary;
//   ROWS: per-row scale and bias present.
template <int DT, int TM, bool STORE, bool EDGE, bool ROWS>
__device__ __forceinline__ void epilogue2(const GemmArgs& p, typename Acc<DT>::type (&acc)[TM][2], int g, int gh, int m0,
                                          int n0, int wr, int wc, int frow, int fkg, float* red) {
    constexpr int BM2 = 64 * TM;
    const float* refg = STORE ? nullptr : p.ref + (int64_t)g * p.sRg;
    float* outg = STORE ? p.out + (int64_t)g * p.sOg : nullptr;
    const int ldr = (int)p.ldr, rcs = (int)p.ref_cs, ldo = (int)p.ldo;
    float alpha[2], beta[2], cm[2], csum[2] = {0.0f, 0.0f};
    int rc0[2], colj[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int col = n0 + wc * 64 + j * 32 + frow;
        const bool cv = !EDGE || col < p.N;
        colj[j] = col;
        const int colc = cv ? col : p.N - 1;
        const int ci = p.ref_div > 1 ? colc % p.ref_div : 0;
        const int ni = p.ref_div > 1 ? colc / p.ref_div : colc;
        alpha[j] = p.sa[ci * p.sa_c + gh * p.sa_g] * p.sa_mul * p.sb[ci * p.sb_c + gh * p.sb_g + ni * p.sb_n];
        beta[j] = p.bias ? p.bias[ci * p.bi_c + gh * p.bi_g + ni * p.bi_n] : 0.0f;
        rc0[j] = ni * rcs;
        cm[j] = cv ? 1.0f : 0.0f;
    }
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int rb0 = m0 + wr * (BM2 / 2) + i * 32 + 4 * fkg;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = rb0 + (r & 3) + 8 * (r >> 2);
            if (!EDGE || row < p.M) {                   // edge tiles (rare) predicate whole rows; interior tiles have no branch
                float rs = 1.0f, rbv = 0.0f;
                if (ROWS) { rs = p.row_scale[row]; rbv = p.row_bias[row]; }
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    float o = (float)acc[i][j][r] * alpha[j];
                    if (ROWS) o = o * rs + rbv;
                    o += beta[j];
                    if (STORE) {
                        if (!EDGE || cm[j] != 0.0f) outg[row * ldo + colj[j]] = o;
                    } else {
                        const float e = refg[row * ldr + rc0[j]] - o;
                        csum[j] += EDGE ? (e * e) * cm[j] : e * e;
                    }
                }
            }
            if ((r & 7) == 7) __builtin_amdgcn_sched_barrier(0);
        }
    }
    if (!STORE) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            float cs = csum[j];
            cs += __shfl_xor(cs, 32);
            if (fkg == 0) red[((wr * 4 + wc) * 2 + j) * 32 + frow] = cs;
        }
    }
}

// Scoring epilogue with the reference slice staged through LDS (candidates-in-columns layout, rows contiguous in ref:
// ldr == 1).  The tile needs only (256 / ref_div) reference columns x BM2 rows (1-4 KiB instead of BM2 x 256 values):
// they are fetched with one coalesced pass, and each thread then reads four consecutive rows per ds_read_b128.
template <int DT, int TM, bool EDGE, bool ROWS>
__device__ __forceinline__ void epilogue_lds(const GemmArgs& p, typename Acc<DT>::type (&acc)[TM][2], int g, int gh, int m0,
                                             int n0, int wr, int wc, int frow, int fkg, float* red, float* stage) {
    constexpr int BM2 = 64 * TM;
    const float* refg = p.ref + (int64_t)g * p.sRg;
    const int rcs = (int)p.ref_cs;
    const int nref = BN2 / p.ref_div;                       // reference columns touched by this tile (<= 8)
    const int ni0 = n0 / p.ref_div;
    const int nvalid = (p.N / p.ref_div) - ni0;              // reference columns that exist
    float* refs = stage;                                     // [nref][BM2]
    float* rsc = stage + 8 * BM2;                            // [BM2]
    float* rbi = rsc + BM2;
    const int tid = threadIdx.x;
    for (int e = tid; e < nref * BM2; e += 512) {
        const int nl = e / BM2, rl = e - nl * BM2;
        const int row = m0 + rl;
        float v = 0.0f;
        if ((!EDGE || row < p.M) && nl < nvalid) v = refg[row + (ni0 + nl) * rcs];
        refs[e] = v;
    }
    if (ROWS) {
        for (int e = tid; e < BM2; e += 512) {
            const int row = m0 + e;
            const bool ok = !EDGE || row < p.M;
            rsc[e] = ok ? p.row_scale[row] : 0.0f;
            rbi[e] = ok ? p.row_bias[row] : 0.0f;
        }
    }
    __syncthreads();
    float alpha[2], beta[2], cm[2], csum[2] = {0.0f, 0.0f};
    const float* rj[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int col = n0 + wc * 64 + j * 32 + frow;
        const bool cv = !EDGE || col < p.N;
        const int colc = cv ? col : p.N - 1;
        const int ci = colc % p.ref_div, ni = colc / p.ref_div;
        alpha[j] = p.sa[ci * p.sa_c + gh * p.sa_g] * p.sa_mul * p.sb[ci * p.sb_c + gh * p.sb_g + ni * p.sb_n];
        beta[j] = p.bias ? p.bias[ci * p.bi_c + gh * p.bi_g + ni * p.bi_n] : 0.0f;
        cm[j] = cv ? 1.0f : 0.0f;
        rj[j] = refs + (ni - ni0) * BM2;
    }
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) {
            const int rl = wr * (BM2 / 2) + i * 32 + 4 * fkg + 8 * q4;          // 4 consecutive rows rl .. rl+3
            const float4 r0 = *reinterpret_cast<const float4*>(rj[0] + rl);
            const float4 r1 = *reinterpret_cast<const float4*>(rj[1] + rl);
            float4 s4 = make_float4(1.f, 1.f, 1.f, 1.f), b4 = make_float4(0.f, 0.f, 0.f, 0.f);
            if (ROWS) { s4 = *reinterpret_cast<const float4*>(rsc + rl); b4 = *reinterpret_cast<const float4*>(rbi + rl); }
            const float rr0[4] = {r0.x, r0.y, r0.z, r0.w}, rr1[4] = {r1.x, r1.y, r1.z, r1.w};
            const float ss[4] = {s4.x, s4.y, s4.z, s4.w}, bb[4] = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int r = q4 * 4 + k;
                float w = 1.0f;
                if (EDGE) w = (m0 + rl + k < p.M) ? 1.0f : 0.0f;
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    float o = (float)acc[i][j][r] * alpha[j];
                    if (ROWS) o = o * ss[k] + bb[k];
                    o += beta[j];
                    const float e = (j == 0 ? rr0[k] : rr1[k]) - o;
                    csum[j] += EDGE ? (e * e) * (w * cm[j]) : e * e;
                }
            }
        }
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        float cs = csum[j];
        cs += __shfl_xor(cs, 32);
        if (fkg == 0) red[((wr * 4 + wc) * 2 + j) * 32 + frow] = cs;
    }
}

template <int DT, int TM, bool STORE>
__global__ __launch_bounds__(512, 2) void k_gemm_cand(GemmArgs p) {
    constexpr int BM2 = 64 * TM;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    uint8_t* As = smem;
    uint8_t* Bs = smem + BM2 * BK2;
    float* red = reinterpret_cast<float*>(smem + (BM2 + BN2) * BK2);       // [2][4][2][32] then colv[256]
    float* colv = red + 512;

    const unsigned nwg = gridDim.x, bid = blockIdx.x;
    const unsigned q8 = nwg >> 3, r8 = nwg & 7, xcd = bid & 7;
    const unsigned lid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
    unsigned t = lid;
    int nt, mt, g;
    if (p.order == 2) { mt = t % p.MT; t /= p.MT; nt = t % p.NT; g = t / p.NT; }
    else { nt = t % p.NT; t /= p.NT; mt = t % p.MT; g = t / p.MT; }
    const int gh = g % p.gmod;
    const int m0 = mt * BM2, n0 = nt * BN2;

    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, wr = w >> 2, wc = w & 3;
    const int lrow = tid >> 3, lslot = tid & 7;
    // per-thread global source addresses (clamped rows at the edges) and swizzled LDS destinations
    const uint8_t* Ab = p.A + g * p.sAg + lslot * 16;
    const uint8_t* Bb = p.B + g * p.sBg + lslot * 16;
    int64_t oa0, oa1, oa2, oa3, ob0, ob1, ob2, ob3;
    {
        auto rowoff = [&](int base, int lim) { int r = base < lim ? base : lim - 1; return (int64_t)r * p.Kb; };
        oa0 = rowoff(m0 + lrow, p.M); oa1 = rowoff(m0 + lrow + 64, p.M);
        oa2 = rowoff(m0 + lrow + 128, p.M); oa3 = rowoff(m0 + lrow + 192, p.M);
        ob0 = rowoff(n0 + lrow, p.N); ob1 = rowoff(n0 + lrow + 64, p.N);
        ob2 = rowoff(n0 + lrow + 128, p.N); ob3 = rowoff(n0 + lrow + 192, p.N);
    }
    const int s0 = swz2(lrow, lslot), s1 = swz2(lrow + 64, lslot), s2 = swz2(lrow + 128, lslot), s3 = swz2(lrow + 192, lslot);

    typename Acc<DT>::type acc[TM][2];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0;

    const int nk = (int)(p.Kb / BK2);
    uint4 ra0, ra1, ra2, ra3, rb0, rb1, rb2, rb3;
    ra1 = ra2 = ra3 = make_uint4(0, 0, 0, 0);
#define GLOAD(KT)                                                                              \
    do {                                                                                       \
        const int64_t ko = (int64_t)(KT) * BK2;                                                \
        ra0 = *reinterpret_cast<const uint4*>(Ab + oa0 + ko);                                  \
        if (TM >= 2) ra1 = *reinterpret_cast<const uint4*>(Ab + oa1 + ko);                     \
        if (TM >= 4) { ra2 = *reinterpret_cast<const uint4*>(Ab + oa2 + ko);                   \
                       ra3 = *reinterpret_cast<const uint4*>(Ab + oa3 + ko); }                 \
        rb0 = *reinterpret_cast<const uint4*>(Bb + ob0 + ko);                                  \
        rb1 = *reinterpret_cast<const uint4*>(Bb + ob1 + ko);                                  \
        rb2 = *reinterpret_cast<const uint4*>(Bb + ob2 + ko);                                  \
        rb3 = *reinterpret_cast<const uint4*>(Bb + ob3 + ko);                                  \
    } while (0)
#define SSTORE()                                                                               \
    do {                                                                                       \
        *reinterpret_cast<uint4*>(As + s0) = ra0;                                              \
        if (TM >= 2) *reinterpret_cast<uint4*>(As + s1) = ra1;                                 \
        if (TM >= 4) { *reinterpret_cast<uint4*>(As + s2) = ra2;                               \
                       *reinterpret_cast<uint4*>(As + s3) = ra3; }                             \
        *reinterpret_cast<uint4*>(Bs + s0) = rb0; *reinterpret_cast<uint4*>(Bs + s1) = rb1;    \
        *reinterpret_cast<uint4*>(Bs + s2) = rb2; *reinterpret_cast<uint4*>(Bs + s3) = rb3;    \
    } while (0)
    GLOAD(0);
    SSTORE();
    __syncthreads();

    const int frow = lane & 31, fkg = lane >> 5;
    const int arow = wr * (BM2 / 2) + frow, brow = wc * 64 + frow;
    for (int kt = 0; kt < nk; ++kt) {
        if (kt + 1 < nk) GLOAD(kt + 1);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            uint4 af[TM], bf[2];
#pragma unroll
            for (int i = 0; i < TM; ++i) af[i] = *reinterpret_cast<const uint4*>(As + swz2(arow + i * 32, ks * 2 + fkg));
#pragma unroll
            for (int j = 0; j < 2; ++j) bf[j] = *reinterpret_cast<const uint4*>(Bs + swz2(brow + j * 32, ks * 2 + fkg));
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) mma<DT>(af[i], bf[j], acc[i][j]);
        }
        if (kt + 1 < nk) {
            __syncthreads();
            SSTORE();
            __syncthreads();
        }
    }
#undef GLOAD
#undef SSTORE

    const bool edge = (m0 + BM2 > p.M) || (n0 + BN2 > p.N);
    const bool rows = p.row_scale != nullptr;
    const bool lds_ref = !STORE && p.ldr == 1 && p.ref_div >= 32 && (BN2 % p.ref_div) == 0;
    if (lds_ref) {
        __syncthreads();                                   // every wave is done with the operand tiles: reuse As as staging
        float* stage = reinterpret_cast<float*>(smem);
        if (edge) {
            if (rows) epilogue_lds<DT, TM, true, true>(p, acc, g, gh, m0, n0, wr, wc, frow, fkg, red, stage);
            else epilogue_lds<DT, TM, true, false>(p, acc, g, gh, m0, n0, wr, wc, frow, fkg, red, stage);
        } else {
            if (rows) epilogue_lds<DT, TM, false, true>(p, acc, g, gh, m0, n0, wr, wc, frow, fkg, red, stage);
            else epilogue_lds<DT, TM, false, false>(p, acc, g, gh, m0, n0, wr, wc, frow, fkg, red, stage);
        }
    } else if (edge) {
        if (rows) epilogue2<DT, TM, STORE, true, true>(p, acc, g, gh, m0, n0, wr, wc, frow, fkg, red);
        else epilogue2<DT, TM, STORE, true, false>(p, acc, g, gh, m0, n0, wr, wc, frow, fkg, red);
    } else {
        if (rows) epilogue2<DT, TM, STORE, false, true>(p, acc, g, gh, m0, n0, wr, wc, frow, fkg, red);
        else epilogue2<DT, TM, STORE, false, false>(p, acc, g, gh, m0, n0, wr, wc, frow, fkg, red);
    }
    if (!STORE && p.partial) {
        __syncthreads();
        float v = 0.0f;
        const int col = n0 + tid;                                      // tid < 256: column tid of the tile
        if (tid < 256) v = red[tid] + red[256 + tid];                  // wr = 0 plus wr = 1 (index = wc*64 + j*32 + lane)
        if (p.reduce_cols) {
            if (tid < 256) colv[tid] = v;
            __syncthreads();
            if (tid < 64) {
                float t2 = (colv[tid] + colv[tid + 64]) + (colv[tid + 128] + colv[tid + 192]);
#pragma unroll
                for (int sft = 32; sft > 0; sft >>= 1) t2 += __shfl_xor(t2, sft);
                if (tid == 0) p.partial[(((int64_t)g) * p.MT + mt) * p.Npad + nt] = t2;
            }
        } else if (tid < 256) {
            if (p.ref_div > 1) {
                if (col < p.N)
                    p.partial[((((int64_t)g) * p.MT + mt) * p.Npad + col / p.ref_div) * p.ref_div + col % p.ref_div] = v;
            } else if (col < p.Npad) {
                p.partial[(((int64_t)g) * p.MT + mt) * p.Npad + col] = v;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------ LDS-DMA pipelined variant
// Same tile and epilogues as k_gemm_cand, but the operand tiles go global -> LDS directly (global_load_lds_dwordx4, no
// VGPR round trip, no ds_write pass) into a ring of NS = 3 stages, with counted vmcnt and raw s_barrier so that two
// K-steps of loads stay in flight across the barrier (cdna guide section 5: "glds span barrier").  The LDS destination of
// an LDS-DMA is lane-linear (wave base + lane*16), so the bank swizzle is applied to the per-lane GLOBAL source address
// (rule 21): lane l of a request covering 8 rows fetches logical slot (l & 7) ^ ((row >> 1) & 7) of row (l >> 3).
typedef const void __attribute__((address_space(1)))* gas_ptr;
typedef void __attribute__((address_space(3)))* las_ptr;

// One K-step of the LDS-DMA pipeline.  `cur` (stage being read) and `nxt` (ring slot being refilled) are __restrict__ so
// that, after inlining, the DMA stores and the fragment reads carry disjoint alias scopes: without them the waitcnt
// insertion pass must assume the ds_reads alias the in-flight LDS-DMA and drains vmcnt(0) every step.
template <int DT, int TM, int NA>
__device__ __forceinline__ void glds_step(const uint8_t* __restrict__ cur, uint8_t* __restrict__ nxt, bool do_issue,
                                          const uint8_t* Ab, const uint8_t* Bb, const int64_t (&oa)[NA], const int64_t (&ob)[4],
                                          const int (&la)[NA], const int (&lb)[4], int64_t ko, int arow, int brow, int fkg,
                                          typename Acc<DT>::type (&acc)[TM][2]) {
    constexpr int BM2 = 64 * TM;
    if (do_issue) {
#pragma unroll
        for (int i = 0; i < NA; ++i)
            __builtin_amdgcn_global_load_lds((gas_ptr)(Ab + oa[i] + ko), (las_ptr)(nxt + la[i]), 16, 0, 0);
#pragma unroll
        for (int i = 0; i < 4; ++i)
            __builtin_amdgcn_global_load_lds((gas_ptr)(Bb + ob[i] + ko), (las_ptr)(nxt + lb[i]), 16, 0, 0);
    }
    const uint8_t* As = cur;
    const uint8_t* Bs = cur + BM2 * BK2;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        uint4 af[TM], bf[2];
#pragma unroll
        for (int i = 0; i < TM; ++i) af[i] = *reinterpret_cast<const uint4*>(As + swz2(arow + i * 32, ks * 2 + fkg));
#pragma unroll
        for (int j = 0; j < 2; ++j) bf[j] = *reinterpret_cast<const uint4*>(Bs + swz2(brow + j * 32, ks * 2 + fkg));
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) mma<DT>(af[i], bf[j], acc[i][j]);
    }
}

template <int DT, int TM>
__global__ __launch_bounds__(512, 2) void k_gemm_cand_glds(GemmArgs p) {
    constexpr int BM2 = 64 * TM;
    constexpr int STAGE = (BM2 + BN2) * BK2;
    constexpr int NS = 3;
    constexpr int NA = BM2 / 64;                   // 8-row requests per wave for the A tile (BM2/8 requests over 8 waves)
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    float* red = reinterpret_cast<float*>(smem + NS * STAGE);
    float* colv = red + 512;

    const unsigned nwg = gridDim.x, bid = blockIdx.x;
    const unsigned q8 = nwg >> 3, r8 = nwg & 7, xcd = bid & 7;
    const unsigned lid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
    unsigned t = lid;
    int nt, mt, g;
    if (p.order == 2) { mt = t % p.MT; t /= p.MT; nt = t % p.NT; g = t / p.NT; }
    else { nt = t % p.NT; t /= p.NT; mt = t % p.MT; g = t / p.MT; }
    const int gh = g % p.gmod;
    const int m0 = mt * BM2, n0 = nt * BN2;

    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = w >> 2, wc = w & 3;
    const int lr = lane >> 3, lslot = lane & 7;
    // per-lane global sources: A requests rbA = w*NA + i (i < NA), B requests rbB = w*4 + i (i < 4); 8 rows per request
    const uint8_t* Ab = p.A + g * p.sAg;
    const uint8_t* Bb = p.B + g * p.sBg;
    int64_t oa[NA], ob[4];
    int la[NA], lb[4];                              // wave-uniform LDS byte offsets of the requests inside a stage
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        const int rl = (w * NA + i) * 8 + lr;       // row inside the tile
        int r = m0 + rl; r = r < p.M ? r : p.M - 1;
        oa[i] = (int64_t)r * p.Kb + ((lslot ^ ((rl >> 1) & 7)) << 4);
        la[i] = (w * NA + i) * 1024;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int rl = (w * 4 + i) * 8 + lr;
        int r = n0 + rl; r = r < p.N ? r : p.N - 1;
        ob[i] = (int64_t)r * p.Kb + ((lslot ^ ((rl >> 1) & 7)) << 4);
        lb[i] = BM2 * BK2 + (w * 4 + i) * 1024;
    }
    auto issue = [&](int kt) {
        uint8_t* st = smem + (kt % NS) * STAGE;
        const int64_t ko = (int64_t)kt * BK2;
#pragma unroll
        for (int i = 0; i < NA; ++i)
            __builtin_amdgcn_global_load_lds((gas_ptr)(Ab + oa[i] + ko), (las_ptr)(st + la[i]), 16, 0, 0);
#pragma unroll
        for (int i = 0; i < 4; ++i)
            __builtin_amdgcn_global_load_lds((gas_ptr)(Bb + ob[i] + ko), (las_ptr)(st + lb[i]), 16, 0, 0);
    };

    typename Acc<DT>::type acc[TM][2];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0;

    const int nk = (int)(p.Kb / BK2);
    issue(0);
    if (nk > 1) issue(1);
    const int frow = lane & 31, fkg = lane >> 5;
    const int arow = wr * (BM2 / 2) + frow, brow = wc * 64 + frow;
    for (int kt = 0; kt < nk; ++kt) {
        // stage kt has landed once at most one later stage (NA + 4 requests of this wave) is still outstanding
        if (kt + 1 < nk) {
            if (NA == 2) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();               // everyone's part of stage kt landed; everyone finished stage kt-1
        asm volatile("" ::: "memory");
        glds_step<DT, TM, NA>(smem + (kt % NS) * STAGE, smem + ((kt + 2) % NS) * STAGE, kt + 2 < nk, Ab, Bb, oa, ob, la, lb,
                              (int64_t)(kt + 2) * BK2, arow, brow, fkg, acc);
    }

    const bool edge = (m0 + BM2 > p.M) || (n0 + BN2 > p.N);
    const bool rows = p.row_scale != nullptr;
    const bool lds_ref = p.ldr == 1 && p.ref_div >= 32 && (BN2 % p.ref_div) == 0;
    __syncthreads();
    if (lds_ref) {
        float* stage = reinterpret_cast<float*>(smem);
        if (edge) {
            if (rows) epilogue_lds<DT, TM, true, true>(p, acc, g, gh, m0, n0, wr, wc, frow, fkg, red, stage);
            else epilogue_lds<DT, TM, true, false>(p, acc, g, gh, m0, n0, wr, wc, frow, fkg, red, stage);
        } else {
            if (rows) epilogue_lds<DT, TM, false, true>(p, acc, g, gh, m0, n0, wr, wc, frow, fkg, red, stage);
            else epilogue_lds<DT, TM, false, false>(p, acc, g, gh, m0, n0, wr, wc, frow, fkg, red, stage);
        }
    } else if (edge) {
        if (rows) epilogue2<DT, TM, false, true, true>(p, acc, g, gh, m0, n0, wr, wc, frow, fkg, red);
        else epilogue2<DT, TM, false, true, false>(p, acc, g, gh, m0, n0, wr, wc, frow, fkg, red);
    } else {
        if (rows) epilogue2<DT, TM, false, false, true>(p, acc, g, gh, m0, n0, wr, wc, frow, fkg, red);
        else epilogue2<DT, TM, false, false, false>(p, acc, g, gh, m0, n0, wr, wc, frow, fkg, red);
    }
    if (p.partial) {
        __syncthreads();
        float v = 0.0f;
        const int col = n0 + tid;
        if (tid < 256) v = red[tid] + red[256 + tid];
        if (p.reduce_cols) {
            if (tid < 256) colv[tid] = v;
            __syncthreads();
            if (tid < 64) {
                float t2 = (colv[tid] + colv[tid + 64]) + (colv[tid + 128] + colv[tid + 192]);
#pragma unroll
                for (int sft = 32; sft > 0; sft >>= 1) t2 += __shfl_xor(t2, sft);
                if (tid == 0) p.partial[(((int64_t)g) * p.MT + mt) * p.Npad + nt] = t2;
            }
        } else if (tid < 256) {
            if (p.ref_div > 1) {
                if (col < p.N)
                    p.partial[((((int64_t)g) * p.MT + mt) * p.Npad + col / p.ref_div) * p.ref_div + col % p.ref_div] = v;
            } else if (col < p.Npad) {
                p.partial[(((int64_t)g) * p.MT + mt) * p.Npad + col] = v;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------ finish
// scores[c][h?][n?] = -norm * sum over (image = g / gmod, [h], m-tile, [n]) of partial[c][g][mt][n]   in fp64,
// fixed summation order: each thread takes a strided subset, then a fixed LDS tree.
struct FinishArgs {
    const float* partial; float* scores;
    int C, G, gmod, MT, N, Npad;
    int keep_h, keep_n;
    int cin;                 // > 0: partial is [G][MT][Npad][cin] (candidate innermost, written by ref_div launches)
    double norm;
};

// WPO = 1: one wavefront per output (4 per block) -- many outputs with short sums (weight searches);
// WPO = 0: one 256-thread block per output -- few outputs with long sums (activation / attention searches).
// Either way each thread takes a fixed strided subset and the combine is a fixed tree in fp64: bit-reproducible.
template <bool WPO>
__global__ __launch_bounds__(256) void k_finish(FinishArgs p) {
    __shared__ double sm[4];
    const int nh = p.keep_h ? p.gmod : 1, nn = p.keep_n ? p.N : 1;
    const int64_t nout = (int64_t)p.C * nh * nn;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int64_t oid = WPO ? (int64_t)blockIdx.x * 4 + wv : (int64_t)blockIdx.x;
    if (oid >= nout) return;
    int64_t o = oid;
    const int n = (int)(o % nn); o /= nn;
    const int h = (int)(o % nh);
    const int c = (int)(o / nh);
    const int n_lo = p.keep_n ? n : 0, n_cnt = p.keep_n ? 1 : p.N;
    const int imgs = p.G / p.gmod;
    const int h_lo = p.keep_h ? h : 0, h_cnt = p.keep_h ? 1 : p.gmod;
    const int64_t total = (int64_t)imgs * h_cnt * p.MT * n_cnt;
    double acc = 0.0;
    for (int64_t i = WPO ? lane : threadIdx.x; i < total; i += WPO ? 64 : 256) {
        int64_t t = i;
        const int nn_i = (int)(t % n_cnt); t /= n_cnt;
        const int mt = (int)(t % p.MT); t /= p.MT;
        const int hh = (int)(t % h_cnt); t /= h_cnt;
        const int img = (int)t;
        const int g = img * p.gmod + h_lo + hh;
        const int64_t pi = p.cin > 0 ? ((((int64_t)g) * p.MT + mt) * p.Npad + n_lo + nn_i) * p.cin + c
                                     : (((int64_t)c * p.G + g) * p.MT + mt) * p.Npad + n_lo + nn_i;
        acc += (double)p.partial[pi];
    }
#pragma unroll
    for (int s = 32; s > 0; s >>= 1) acc += __shfl_xor(acc, s);
    if (WPO) {
        if (lane == 0) p.scores[oid] = (float)(-p.norm * acc);
    } else {
        if (lane == 0) sm[wv] = acc;
        __syncthreads();
        if (threadIdx.x == 0) p.scores[oid] = (float)(-p.norm * ((sm[0] + sm[1]) + (sm[2] + sm[3])));
    }
}

// Thread-per-output finish for the candidate-innermost layout with short sums (weight searches: [P][O] outputs, MT terms
// each): adjacent threads take adjacent candidates, so every step of the sequential fp64 sum is a coalesced read.
__global__ __launch_bounds__(256) void k_finish_tpo(FinishArgs p) {
    const int nh = p.keep_h ? p.gmod : 1, nn = p.keep_n ? p.N : 1;
    const int64_t nout = (int64_t)p.C * nh * nn;
    const int64_t tid = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (tid >= nout) return;
    const int c = (int)(tid % p.C);
    int64_t o = tid / p.C;
    const int n = (int)(o % nn);
    const int h = (int)(o / nn);
    const int n_lo = p.keep_n ? n : 0, n_cnt = p.keep_n ? 1 : p.N;
    const int imgs = p.G / p.gmod;
    const int h_lo = p.keep_h ? h : 0, h_cnt = p.keep_h ? 1 : p.gmod;
    double acc = 0.0;
    for (int img = 0; img < imgs; ++img)
        for (int hh = 0; hh < h_cnt; ++hh) {
            const int g = img * p.gmod + h_lo + hh;
            for (int mt = 0; mt < p.MT; ++mt)
                for (int ni = 0; ni < n_cnt; ++ni)
                    acc += (double)p.partial[((((int64_t)g) * p.MT + mt) * p.Npad + n_lo + ni) * p.cin + c];
        }
    p.scores[((int64_t)c * nh + h) * nn + n] = (float)(-p.norm * acc);
}

}  // namespace

// ---- tile selection shared by launch, layout query and finish
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

struct Layout { int big, tm, MT, NT, Npad, c_eff, n_eff; int64_t elems; };

static Layout layout_of(int M, int N, int C, int ref_div, int reduce_cols, bool scoring = true) {
    Layout L{};
    L.big = (C == 1);
    L.tm = L.big ? pick_tm(M, scoring) : 2;
    const int bm = L.big ? 64 * L.tm : BM, bn = L.big ? BN2 : BN;
    L.MT = cdiv(M, bm);
    L.NT = cdiv(N, bn);
    L.n_eff = ref_div > 1 ? N / ref_div : N;
    L.c_eff = ref_div > 1 ? ref_div : C;
    L.Npad = reduce_cols ? L.NT : (ref_div > 1 ? cdiv(L.n_eff, 64) * 64 : L.NT * bn);
    return L;
}

// M, N: GEMM rows / columns (N includes the candidate factor when ref_div > 1).  Outputs the partial-buffer layout
// [c_eff][G][MT][Npad] the kernel will write, for allocation and for adalog_finish_scores.
extern "C" int64_t adalog_gemm_score_layout(int M, int N, int C, int G, int ref_div, int reduce_cols, int* MT, int* Npad) {
    const Layout L = layout_of(M, N, C, ref_div, reduce_cols);
    if (MT) *MT = L.MT;
    if (Npad) *Npad = L.Npad;
    return (int64_t)L.c_eff * G * L.MT * L.Npad;
}

extern "C" int adalog_gemm_score(int dtype, const void* A, const void* B, int64_t sAc, int64_t sAg, int64_t sBc,
                                 int64_t sBg, int M, int N, int64_t Kp, int C, int G, int gmod, const float* ref,
                                 int64_t ldr, int64_t sRg, int64_t ref_cs, int ref_div, const float* sa, int64_t sa_c,
                                 int64_t sa_g, float sa_mul, const float* sb, int64_t sb_c, int64_t sb_g, int64_t sb_n,
                                 const float* bias, int64_t bi_c, int64_t bi_g, int64_t bi_n, const float* row_scale,
                                 const float* row_bias, float* partial, int64_t partial_elems, float* out, int64_t ldo,
                                 int64_t sOc, int64_t sOg, int order, int reduce_cols, void* stream) {
    ADALOG_ARG_CHECK(A && B && sa && sb, "gemm_score: null operand/scale pointer");
    ADALOG_ARG_CHECK(dtype >= 0 && dtype <= 2, "gemm_score: dtype must be 0 (i8), 1 (bf16) or 2 (f32)");
    ADALOG_ARG_CHECK(M >= 1 && N >= 1 && C >= 1 && G >= 1 && gmod >= 1 && G % gmod == 0 && ref_div >= 1, "gemm_score: bad sizes");
    const int esz = dtype == 0 ? 1 : dtype == 1 ? 2 : 4;
    ADALOG_ARG_CHECK((Kp * esz) % BK2 == 0 && Kp > 0, "gemm_score: padded K must be a multiple of 128 bytes");
    ADALOG_ARG_CHECK((partial != nullptr) == (ref != nullptr), "gemm_score: partial and ref go together");
    ADALOG_ARG_CHECK(partial || out, "gemm_score: nothing to produce");
    ADALOG_ARG_CHECK(order >= 0 && order <= 2, "gemm_score: order must be 0, 1 or 2");
    ADALOG_ARG_CHECK(ref_div == 1 || (C == 1 && N % ref_div == 0 && !out), "gemm_score: ref_div > 1 needs C == 1, N % ref_div == 0, no out");
    ADALOG_ARG_CHECK(!(reduce_cols && ref_div > 1), "gemm_score: reduce_cols and ref_div > 1 are exclusive");
    ADALOG_ARG_CHECK(!row_scale || (C == 1 && row_bias), "gemm_score: per-row scale needs C == 1 and a row_bias vector");
    ADALOG_ARG_CHECK(!(partial && out), "gemm_score: either score against ref or store out, not both");
    const Layout L = layout_of(M, N, C, ref_div, reduce_cols, out == nullptr);
    GemmArgs p{};
    p.A = (const uint8_t*)A; p.B = (const uint8_t*)B;
    p.sAc = sAc * esz; p.sAg = sAg * esz; p.sBc = sBc * esz; p.sBg = sBg * esz;
    p.M = M; p.N = N; p.Kb = Kp * esz; p.C = C; p.G = G; p.gmod = gmod;
    p.ref = ref; p.ldr = ldr; p.sRg = sRg; p.ref_cs = ref_cs; p.ref_div = ref_div;
    ADALOG_ARG_CHECK(!ref || ((int64_t)(M - 1) * ldr + (int64_t)(L.n_eff - 1) * (ref_cs > 0 ? ref_cs : 1) < ((int64_t)1 << 31)),
                     "gemm_score: reference group exceeds 32-bit addressing");
    p.sa = sa; p.sa_c = sa_c; p.sa_g = sa_g; p.sa_mul = sa_mul;
    p.sb = sb; p.sb_c = sb_c; p.sb_g = sb_g; p.sb_n = sb_n;
    p.bias = bias; p.bi_c = bi_c; p.bi_g = bi_g; p.bi_n = bi_n;
    p.row_scale = row_scale; p.row_bias = row_bias;
    p.MT = L.MT; p.NT = L.NT; p.Npad = L.Npad;
    p.order = order; p.reduce_cols = reduce_cols;
    p.partial = partial; p.out = out; p.ldo = ldo; p.sOc = sOc; p.sOg = sOg;
    if (partial) ADALOG_ARG_CHECK(partial_elems >= (int64_t)L.c_eff * G * L.MT * L.Npad, "gemm_score: partial buffer too small");
    const int64_t nwg = (int64_t)L.MT * L.NT * G * C;
    ADALOG_ARG_CHECK(nwg < (int64_t)1 << 31, "gemm_score: grid too large");
    hipStream_t st = (hipStream_t)stream;
    dim3 grid((unsigned)nwg);
    static const int use_glds = getenv("ADALOG_GEMM_GLDS") ? atoi(getenv("ADALOG_GEMM_GLDS")) : 1;   // LDS-DMA pipeline (default on)
    if (L.big && use_glds && !out && L.tm <= 2) {
        const size_t shm = (size_t)3 * (64 * L.tm + BN2) * BK2 + (512 + 256) * sizeof(float);
#define LAUNCH_GLDS(DT, TMV)                                                                                      \
        do {                                                                                                      \
            static bool attr_set = false;                                                                         \
            if (!attr_set) {                                                                                      \
                (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gemm_cand_glds<DT, TMV>),              \
                                          hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);                \
                attr_set = true;                                                                                  \
            }                                                                                                     \
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
            static bool attr_set = false;                                                                         \
            if (!attr_set) {                                                                                      \
                (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gemm_cand<DT, TMV, ST>),                   \
                                    hipFuncAttributeMaxDynamicSharedMemorySize, 72 * 1024);                       \
                attr_set = true;                                                                                  \
            }                                                                                                     \
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
        if (out) LAUNCH_BIG_DT(true); else LAUNCH_BIG_DT(false);
#undef LAUNCH_BIG_DT
#undef LAUNCH_BIG_TM
#undef LAUNCH_BIG
    } else {
        ADALOG_ARG_CHECK(!row_scale, "gemm_score: per-row scale is only available with C == 1");
        dim3 block(256);
#define LAUNCH(DT)                                                                                   \
        do {                                                                                         \
            if (out) hipLaunchKernelGGL((k_gemm_score<DT, true>), grid, block, 0, st, p);            \
            else hipLaunchKernelGGL((k_gemm_score<DT, false>), grid, block, 0, st, p);               \
        } while (0)
        if (dtype == 0) LAUNCH(0); else if (dtype == 1) LAUNCH(1); else LAUNCH(2);
#undef LAUNCH
    }
    ADALOG_LAUNCH_CHECK("adalog_gemm_score");
    return 0;
}

// scores[c][h?][n?] = -norm * sum over (image, [h], m_tile, [n]) of partial[c][g][m_tile][n] with the layout returned by
// adalog_gemm_score_layout (MT, Npad); N = number of valid entries along the last axis (n_eff, or NT when reduced).
extern "C" int adalog_finish_scores(const float* partial, float* scores, int MT, int N, int Npad, int C, int G, int gmod,
                                    int keep_h, int keep_n, int cand_inner, double norm, void* stream) {
    ADALOG_ARG_CHECK(partial && scores && MT >= 1 && N >= 1 && Npad >= N && C >= 1 && G >= 1 && gmod >= 1 && G % gmod == 0,
                     "finish_scores: bad arguments");
    FinishArgs p{};
    p.partial = partial; p.scores = scores; p.C = C; p.G = G; p.gmod = gmod; p.MT = MT; p.N = N; p.Npad = Npad;
    p.keep_h = keep_h; p.keep_n = keep_n; p.norm = norm; p.cin = cand_inner ? C : 0;
    const int64_t nout = (int64_t)C * (keep_h ? gmod : 1) * (keep_n ? N : 1);
    const int64_t per_out = (int64_t)(G / gmod) * (keep_h ? 1 : gmod) * MT * (keep_n ? 1 : N);
    if (p.cin > 0 && per_out <= 512 && nout >= 4096)
        hipLaunchKernelGGL(k_finish_tpo, dim3((unsigned)((nout + 255) / 256)), dim3(256), 0, (hipStream_t)stream, p);
    else if (per_out >= 2048)
        hipLaunchKernelGGL(k_finish<false>, dim3((unsigned)nout), dim3(256), 0, (hipStream_t)stream, p);
    else
        hipLaunchKernelGGL(k_finish<true>, dim3((unsigned)((nout + 3) / 4)), dim3(256), 0, (hipStream_t)stream, p);
    ADALOG_LAUNCH_CHECK("adalog_finish_scores");
    return 0;
}
