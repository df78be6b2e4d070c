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
#include <type_traits>
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
    int M, N; int64_t Kb;            // Kb = padded K in bytes: the row stride (multiple of 128; 64 for the streaming kernel)
    int64_t Kvb;                     // bytes of a row that can be non-zero (<= Kb)
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
    double* wg_acc;                  // streaming kernel: per-workgroup fp64 column sums [workgroup][gmod][256] instead of
                                     // per-tile partials (searches that do not keep the column axis)
    int gm;                          // streaming kernel, order 2: m-tiles per L2 group (rows of A kept hot while n advances)
    int slab_U, slab_R;              // slab kernel: 32-row units per slab, units per workgroup (NT = slabs, MT = pieces);
                                     // group kernel: chunks per group, 32-column blocks per chunk
    long long* timeline;             // profiling only (tools/gemm_lab.hip): 8 cycle stamps per workgroup, else nullptr
};
#if defined(GEMM_LAB_TIMELINE)   // tools/lab only: the stamp stores would otherwise cost waits in the production kernel
#define TL_STAMP(i) do { if (p.timeline && threadIdx.x == 0) p.timeline[(size_t)lid * 8 + (i)] = (long long)__builtin_readcyclecounter(); } while (0)
#else
#define TL_STAMP(i) do { } while (0)
#endif
static long long* g_timeline = nullptr;   // set only by the lab harness, which includes this file
static int g_slab_override = -1;          // lab harness: force the slab kernel off (0) / on (1) per call

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

// fp8 (e4m3) operands, 64 K-bytes per instruction: lane = (row, K half) holds 32 bytes.  Any fixed permutation of K is
// fine as long as A and B share it, so the two 16-byte fragments a lane reads for the int8 path are simply concatenated.
typedef int v8i __attribute__((ext_vector_type(8)));
__device__ __forceinline__ v16f mma_fp8x64(const uint4& a0, const uint4& a1, const uint4& b0, const uint4& b1, v16f c) {
    const v8i a = {(int)a0.x, (int)a0.y, (int)a0.z, (int)a0.w, (int)a1.x, (int)a1.y, (int)a1.z, (int)a1.w};
    const v8i b = {(int)b0.x, (int)b0.y, (int)b0.z, (int)b0.w, (int)b1.x, (int)b1.y, (int)b1.z, (int)b1.w};
    return __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);   // scales 2^0
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
    TL_STAMP(0);
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
    TL_STAMP(1);
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
        if (kt == 0) TL_STAMP(2);
        glds_step<DT, TM, NA>(smem + (kt % NS) * STAGE, smem + ((kt + 2) % NS) * STAGE, kt + 2 < nk, Ab, Bb, oa, ob, la, lb,
                              (int64_t)(kt + 2) * BK2, arow, brow, fkg, acc);
    }

    const bool edge = (m0 + BM2 > p.M) || (n0 + BN2 > p.N);
    const bool rows = p.row_scale != nullptr;
    const bool lds_ref = p.ldr == 1 && p.ref_div >= 32 && (BN2 % p.ref_div) == 0;
    TL_STAMP(3);
    __syncthreads();
    TL_STAMP(4);
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
    TL_STAMP(5);
    if (p.partial) {
        __syncthreads();
        TL_STAMP(6);
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
    TL_STAMP(7);
}

// ------------------------------------------------------------------------------------------------ streaming kernel
// Third-generation scoring kernel (every search: candidates in the column axis, reference rows contiguous).
// What the per-workgroup cycle stamps of tools/lab/gemm_lab.hip showed for the kernel above at K = 384 (three 128-byte
// steps per tile): of 14.2k cycles per tile only 3.1k were matrix work -- 1.6k set-up (index divisions, 64-bit address
// arithmetic), 1.7-3.2k waiting for the first stage, 4.8k main loop (every wave issues its six DMA requests back to
// back at the top of a step while the matrix pipe idles), 4.0-5.4k epilogue (six VALU per output), 2k barriers/stores,
// and nothing overlaps anything because one 147 KiB workgroup owns the CU.  Hence:
//   * PERSISTENT workgroups: each walks a strided list of tiles and its LDS-DMA ring (3 stages of 64-byte K-steps)
//     keeps streaming across tile boundaries, so set-up and first-stage latency are paid once per launch, not per tile;
//   * 256 threads, (64*TM) x 256 tile, wave tile (32*TM) x 128 (fewer ds_reads per MFMA), 77 KiB of LDS: TWO workgroups
//     per CU, un-synchronised, so one's epilogue (VALU) runs under the other's main loop (MFMA);
//   * buffer_load ... lds with a per-tile resource: one VGPR offset per request, k offset in an SGPR, and the requests
//     are issued between the MFMAs of a step instead of in front of them;
//   * epilogue on packed fp32 math: reference slice minus row/column bias, row scale and column factors are staged in
//     LDS during the tile's first K-step; per pair of outputs 2 cvt + 2..4 v_pk_* instead of 12 scalar VALU.
constexpr int BK3 = 64;
// tools/lab builds this file with GEMM_LAB_NO_DMA / GEMM_LAB_NO_MFMA to time the two halves of the main loop separately
#if defined(GEMM_LAB_NO_DMA)
#define STREAM_DMA(rsrc, dst, voff, soff) do { } while (0)
#else
#define STREAM_DMA(rsrc, dst, voff, soff) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (las_ptr)(dst), 16, voff, soff, 0, 0)
#endif
#define STREAM_DMA4(rsrc, dst, voff) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (las_ptr)(dst), 4, (int)(voff), 0, 0, 0)
typedef float v2f __attribute__((ext_vector_type(2)));

__device__ __forceinline__ int swz3(int row, int slot) { return row * BK3 + ((slot ^ ((row >> 2) & 3)) << 4); }

template <int DT> __device__ __forceinline__ typename Acc<DT>::type mma0(const uint4& a, const uint4& b) {
    typename Acc<DT>::type z;
#pragma unroll
    for (int r = 0; r < 16; ++r) z[r] = 0;
    mma<DT>(a, b, z);
    return z;
}

struct StreamTile { int mt, nt, g; };
__device__ __forceinline__ float4 lds_f4(const float* __restrict__ base, int off) { return *reinterpret_cast<const float4*>(base + off); }

// Fragment read through a __restrict__ stage pointer: the load carries alias-scope metadata, which keeps the compiler's
// waitcnt pass from ordering it behind the (untagged) in-flight LDS-DMA with a vmcnt(0); the hand-written counted vmcnt
// in front of each step's barrier is what orders them.
__device__ __forceinline__ uint4 lds_frag(const uint8_t* __restrict__ stage, int off) {
    return *reinterpret_cast<const uint4*>(stage + off);
}

// Two shapes of the same kernel:
//   NW = 4 waves, tile (64*RI) x 256, RI <= 2, 3-stage ring, 79 KiB of LDS -> TWO workgroups per CU (small K: the other
//          workgroup's main loop covers this one's epilogue);
//          (eight waves x 128 x 256 at two per CU -- four waves per SIMD, 128 VGPRs -- measured 8-12 % slower);
//   NW = 8 waves, tile (64*RI) x 256, RI = 3 or 4, 4-stage ring, <= 138 KiB -> one workgroup per CU (large K: the main
//          loop is bound by the L2 -> LDS DMA path, measured ~30 B/clk/CU, and the 192/256-row tile moves 23/33 % fewer
//          operand bytes per MAC; the epilogue is < 10 % of such a tile).
// Waves form a 2 x (NW/2) grid; each owns RI x CJ MFMA tiles (CJ = 16/NW), 128 accumulator registers at most.
template <int DT, int RI, int NW, int NS = (NW == 4 ? 3 : 4)>
__global__ __launch_bounds__(64 * NW, NS == 3 ? (NW == 8 ? 4 : 2) : 1) void k_gemm_stream(GemmArgs p) {
#if defined(__HIP_DEVICE_COMPILE__)   // the buffer-resource type and builtins exist only in the device pass
    constexpr int NT3 = 64 * NW;                     // threads
    constexpr int CJ = 16 / NW;                      // column MFMA tiles per wave
    constexpr int WCOLS = CJ * 32;                   // columns per wave
    constexpr int BM3 = 64 * RI;
    constexpr int STAGE3 = (BM3 + BN2) * BK3;
    constexpr int AP = BM3 / 16;                     // 16-row DMA requests of the A tile per K-step; B has 16
    constexpr int PT = AP + 16;
    constexpr int MAXQ = (PT + NW - 1) / NW;         // requests per wave per K-step (the first PT % NW waves own MAXQ)
    extern __shared__ __attribute__((aligned(16))) uint8_t ring[];   // NS * STAGE3 bytes (dynamic: keeps the DMA untagged)
    __shared__ __attribute__((aligned(16))) float s_ref[4 * BM3];     // [nref <= 4][BM3]: ref - row_bias (- column bias if shared)
    __shared__ __attribute__((aligned(16))) float s_rs[BM3];          // row scale, 0 past M
    __shared__ __attribute__((aligned(16))) float s_w[BM3];           // 1 for rows < M else 0
    __shared__ float s_alpha[BN2], s_beta[BN2];                       // per column: -(sa*sb), column bias; 0 past N
    __shared__ float s_red[2][BN2];

    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = w / (NW / 2), wc = w % (NW / 2);
    const int frow = lane & 31, fkg = lane >> 5;
    const int arow = wr * (BM3 / 2) + frow, brow = wc * WCOLS + frow;

    // ---- tile list of this workgroup: the XCD it runs on owns a contiguous range of tiles (neighbours share operand
    // tiles in that XCD's L2); its workgroups take them round-robin.
    const unsigned T = (unsigned)p.MT * p.NT * p.G;
    const unsigned nwg = gridDim.x, bid = blockIdx.x, xcd = bid & 7;
    const unsigned q8 = T >> 3, r8 = T & 7;
    const unsigned t_lo = xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8;
    const unsigned t_cnt = q8 + (xcd < r8 ? 1u : 0u);
    const unsigned nj = (nwg >> 3) + (xcd < (nwg & 7) ? 1u : 0u);
    const unsigned j0 = bid >> 3;
    auto decode = [&](unsigned local) {
        unsigned t = t_lo + local;
        StreamTile r;
        if (p.order == 2) {
            // grouped: within a group of gm m-tiles the m index runs fastest, then n; the group's A rows (<= ~2 MiB)
            // stay in the XCD's L2 while the B tiles stream through once per group
            const unsigned per_g = (unsigned)p.MT * p.NT;
            r.g = t / per_g; t -= r.g * per_g;
            const unsigned grp = t / ((unsigned)p.gm * p.NT), first = grp * p.gm;
            const unsigned gsz = min((unsigned)p.gm, (unsigned)p.MT - first);
            t -= grp * p.gm * p.NT;
            r.nt = t / gsz; r.mt = first + (t - r.nt * gsz);
        } else { r.nt = t % p.NT; t /= p.NT; r.mt = t % p.MT; r.g = t / p.MT; }
        return r;
    };
    const int nk = (int)((p.Kvb + BK3 - 1) / BK3);          // whole 64-byte steps of zero padding are skipped
    const int Kb = (int)p.Kb;

    // ---- issue cursor: runs NS - 1 K-steps ahead of the compute cursor, across tile boundaries.  Request r of a step
    // covers 16 rows x 64 bytes and lands at stage + r * 1024 (A rows first, then B rows); wave w owns r = w + q * NW.
    // Per request the lane's source is  row * Kb + 16 * (logical slot);  LDS destinations are lane-linear, so the bank
    // swizzle goes into the source slot: lane l lands in (row l>>2, physical slot l&3) = logical slot (l&3) ^ ((l>>4)&3).
    const int lrow = lane >> 2, lslot16 = ((lane & 3) ^ ((lane >> 4) & 3)) << 4;
    __amdgpu_buffer_rsrc_t ra, rb;
    int vo[MAXQ];
    unsigned i_local = j0;
    int i_k = 0;
    auto issue_tile = [&](unsigned local) {
        const StreamTile t = decode(local);
        const int m0 = t.mt * BM3, n0 = t.nt * BN2;
        ra = __builtin_amdgcn_make_buffer_rsrc((void*)(p.A + (int64_t)t.g * p.sAg + (int64_t)m0 * Kb), 0, 0x7ffffffe, 0x00020000);
        rb = __builtin_amdgcn_make_buffer_rsrc((void*)(p.B + (int64_t)t.g * p.sBg + (int64_t)n0 * Kb), 0, 0x7ffffffe, 0x00020000);
#pragma unroll
        for (int q = 0; q < MAXQ; ++q) {
            const int r = w + q * NW;                       // wave-uniform
            const bool isA = r < AP;
            int row = (isA ? r : r - AP) * 16 + lrow;
            row = min(row, isA ? p.M - 1 - m0 : p.N - 1 - n0);   // edge rows: re-read the last valid row (masked in the epilogue)
            vo[q] = row * Kb + lslot16;
        }
    };
    auto issue_advance = [&]() {
        if (++i_k == nk) {
            i_k = 0;
            if (i_local + nj < t_cnt) { i_local += nj; issue_tile(i_local); }   // past the last tile: harmless re-fetch
        }
    };
    auto issue_slot = [&](int q, uint8_t* st, int ko) {
        const int r = w + q * NW;
        if (PT % NW == 0 || q + 1 < MAXQ || r < PT) {
            if (r < AP) STREAM_DMA(ra, st + r * 1024, vo[q], ko);
            else STREAM_DMA(rb, st + r * 1024, vo[q], ko);
        }
    };
    issue_tile(i_local);
#pragma unroll
    for (int s0 = 0; s0 < NS - 1; ++s0) {                   // prologue: the first NS - 1 steps
#pragma unroll
        for (int q = 0; q < MAXQ; ++q) issue_slot(q, ring + s0 * STAGE3, i_k * BK3);
        issue_advance();
    }

    const int rsh = __ffs(p.ref_div) - 1, rmask = p.ref_div - 1;   // ref_div is 64, 128 or 256 here
    const int nref = BN2 >> rsh;                            // reference columns per tile (<= 4)
    const int n_eff = p.N >> rsh;
    const bool beta_staged = p.bias && p.bi_c == 0;         // column bias independent of the candidate: fold into s_ref
    const bool beta_cols = p.bias && p.bi_c != 0;
    const bool rows = p.row_scale != nullptr;
    int st = 0;                                             // ring slot of the current step
    typename Acc<DT>::type acc[RI][CJ];
    // per-workgroup accumulation (wg_acc): thread t < 256 owns tile column t; its running fp64 sum covers this workgroup's
    // tiles of one head in their fixed order and is flushed into [workgroup][head][t] when the head changes
    double run = 0.0;
    int run_h = -1;
    if (p.wg_acc && tid < BN2)
        for (int h = 0; h < p.gmod; ++h) p.wg_acc[((int64_t)bid * p.gmod + h) * BN2 + tid] = 0.0;

    // One 64-byte K-step = 2 sub-steps of (RI + CJ) fragment reads and RI x CJ MFMAs.  Both sub-steps' fragments are read
    // up front; this wave's DMA requests for the step NS - 1 ahead are issued between the MFMA groups.
#define STREAM_STEP(FIRST, cur, nxt, ko)                                                                   \
    do {                                                                                                   \
        const uint8_t* Bs_ = (cur) + BM3 * BK3;                                                            \
        uint4 a0[RI], b0[CJ], a1[RI], b1[CJ];                                                              \
        _Pragma("unroll") for (int i = 0; i < RI; ++i) a0[i] = lds_frag((cur), swz3(arow + i * 32, fkg));  \
        _Pragma("unroll") for (int j = 0; j < CJ; ++j) b0[j] = lds_frag(Bs_, swz3(brow + j * 32, fkg));    \
        _Pragma("unroll") for (int i = 0; i < RI; ++i) a1[i] = lds_frag((cur), swz3(arow + i * 32, 2 + fkg)); \
        _Pragma("unroll") for (int j = 0; j < CJ; ++j) b1[j] = lds_frag(Bs_, swz3(brow + j * 32, 2 + fkg)); \
        if constexpr (DT == 3) {                                                                           \
            _Pragma("unroll") for (int j = 0; j < CJ; ++j) {                                               \
                _Pragma("unroll") for (int i = 0; i < RI; ++i) {                                           \
                    v16f z_;                                                                               \
                    _Pragma("unroll") for (int r = 0; r < 16; ++r) z_[r] = 0.0f;                           \
                    acc[i][j] = mma_fp8x64(a0[i], a1[i], b0[j], b1[j], FIRST ? z_ : acc[i][j]);            \
                }                                                                                          \
                _Pragma("unroll") for (int q = j; q < MAXQ; q += CJ) issue_slot(q, (nxt), (ko));           \
            }                                                                                              \
        } else {                                                                                           \
            _Pragma("unroll") for (int j = 0; j < CJ; ++j) {                                               \
                _Pragma("unroll") for (int i = 0; i < RI; ++i) {                                           \
                    if (FIRST) acc[i][j] = mma0<DT == 3 ? 1 : DT>(a0[i], b0[j]);                           \
                    else mma<DT == 3 ? 1 : DT>(a0[i], b0[j], acc[i][j]);                                   \
                }                                                                                          \
                _Pragma("unroll") for (int q = j; q < MAXQ; q += 2 * CJ) issue_slot(q, (nxt), (ko));       \
            }                                                                                              \
            _Pragma("unroll") for (int j = 0; j < CJ; ++j) {                                               \
                _Pragma("unroll") for (int i = 0; i < RI; ++i) mma<DT == 3 ? 1 : DT>(a1[i], b1[j], acc[i][j]); \
                _Pragma("unroll") for (int q = CJ + j; q < MAXQ; q += 2 * CJ) issue_slot(q, (nxt), (ko));  \
            }                                                                                              \
        }                                                                                                  \
    } while (0)

    for (unsigned local = j0; local < t_cnt; local += nj) {
        const StreamTile tl = decode(local);
        const int g = tl.g, gh = g % p.gmod, m0 = tl.mt * BM3, n0 = tl.nt * BN2;
        const int ni0 = n0 >> rsh;
        const bool edge = (m0 + BM3 > p.M) || (n0 + BN2 > p.N);
        [[maybe_unused]] const unsigned lid = t_lo + local;
        TL_STAMP(0);
        for (int kt = 0; kt < nk; ++kt) {
            // stage `st` has landed once only the newest NS - 2 steps' requests of this wave are still outstanding
            if (PT % NW != 0 && w >= PT % NW) {
                if ((NS - 2) * (MAXQ - 1) == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
                else if ((NS - 2) * (MAXQ - 1) == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                else if ((NS - 2) * (MAXQ - 1) == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            } else {
                if ((NS - 2) * MAXQ == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
                else if ((NS - 2) * MAXQ == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
                else if ((NS - 2) * MAXQ == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            const int stn = st == 0 ? NS - 1 : st - 1;       // (st + NS - 1) % NS: the slot read in the previous step
            const int ko = i_k * BK3;
            if (kt == 0) {
                TL_STAMP(1);
                // ---- epilogue operands of this tile: plain loads now (clamped addresses, no dependent arithmetic, so
                // nothing waits before the MFMAs), masks + arithmetic + LDS staging writes after the step
                const float* refg = p.ref + (int64_t)g * p.sRg;
                const int rcs = (int)p.ref_cs;
                constexpr int EU = (4 * BM3 + NT3 - 1) / NT3;
                float e_ref[EU], e_rb[EU], e_cb[EU];
#pragma unroll
                for (int u = 0; u < EU; ++u) {
                    const int e = min(tid + u * NT3, 4 * BM3 - 1);
                    const int nl = e / BM3, rl = e - nl * BM3;
                    const int rowc = min(m0 + rl, p.M - 1), nic = min(ni0 + nl, n_eff - 1);
#if defined(GEMM_LAB_ELOAD_HOT)   // lab: always-cached address, to tell load latency from instruction overhead
                    e_ref[u] = refg[(rowc + nic * rcs) & 1023];
#else
                    e_ref[u] = refg[rowc + nic * rcs];
#endif
                    e_rb[u] = p.row_bias ? p.row_bias[rowc] : 0.0f;
                    e_cb[u] = beta_staged ? p.bias[gh * p.bi_g + nic * p.bi_n] : 0.0f;
                }
                const int rrow = min(m0 + min(tid, BM3 - 1), p.M - 1);
                const float e_rs = rows ? p.row_scale[rrow] : 1.0f;
                const int colc = min(n0 + min(tid, BN2 - 1), p.N - 1);
                const int cci = colc & rmask, cni = colc >> rsh;
                const float e_sa = p.sa[cci * p.sa_c + gh * p.sa_g];
                const float e_sb = p.sb[cci * p.sb_c + gh * p.sb_g + cni * p.sb_n];
                const float e_be = beta_cols ? p.bias[cci * p.bi_c + gh * p.bi_g + cni * p.bi_n] : 0.0f;
                STREAM_STEP(true, ring + st * STAGE3, ring + stn * STAGE3, ko);
#pragma unroll
                for (int u = 0; u < EU; ++u) {
                    const int e = tid + u * NT3;
                    const int nl = e / BM3, rl = e - nl * BM3;
                    const bool ok = (m0 + rl < p.M) && (ni0 + nl < n_eff);
                    if (e < nref * BM3) s_ref[e] = ok ? (e_ref[u] - e_rb[u]) - e_cb[u] : 0.0f;
                }
                if (tid < BM3) {
                    const bool ok = m0 + tid < p.M;
                    s_rs[tid] = ok ? e_rs : 0.0f;
                    s_w[tid] = ok ? 1.0f : 0.0f;
                }
                if (tid < BN2) {
                    const bool ok = n0 + tid < p.N;
                    s_alpha[tid] = ok ? -(e_sa * p.sa_mul * e_sb) : 0.0f;
                    s_beta[tid] = ok ? e_be : 0.0f;
                }
                TL_STAMP(2);
            } else {
                STREAM_STEP(false, ring + st * STAGE3, ring + stn * STAGE3, ko);
            }
            issue_advance();
            st = st == NS - 1 ? 0 : st + 1;
        }
        TL_STAMP(3);

        // ---- epilogue.  The staging written after the first step must be visible: with nk >= 2 a later step's barrier
        // already separates them.
        if (nk == 1) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        v2f cs2[CJ];
        float nal[CJ], bet[CJ], cm[CJ];
        const float* rj[CJ];
#pragma unroll
        for (int j = 0; j < CJ; ++j) {
            const int cl = wc * WCOLS + j * 32 + frow;
            nal[j] = s_alpha[cl]; bet[j] = s_beta[cl];
            cm[j] = (n0 + cl < p.N) ? 1.0f : 0.0f;
            rj[j] = s_ref + (cl >> rsh) * BM3;
            cs2[j] = (v2f){0.0f, 0.0f};
        }
        const bool full = edge || beta_cols;                // generic body: row weights and column bias applied
#define STREAM_EPILOGUE(ROWS_, FULL_)                                                                          \
        _Pragma("unroll") for (int i = 0; i < RI; ++i) {                                                       \
            _Pragma("unroll") for (int q4 = 0; q4 < 4; ++q4) {                                                 \
                const int rl = wr * (BM3 / 2) + i * 32 + 4 * fkg + 8 * q4;                                     \
                float4 s4 = make_float4(1.f, 1.f, 1.f, 1.f), w4 = s4;                                          \
                if (ROWS_) s4 = *reinterpret_cast<const float4*>(s_rs + rl);                                   \
                if (FULL_) w4 = *reinterpret_cast<const float4*>(s_w + rl);                                    \
                const v2f sA = {s4.x, s4.y}, sB = {s4.z, s4.w}, wA = {w4.x, w4.y}, wB = {w4.z, w4.w};          \
                _Pragma("unroll") for (int j = 0; j < CJ; ++j) {                                               \
                    const float4 r4 = *reinterpret_cast<const float4*>(rj[j] + rl);                            \
                    v2f rA = {r4.x, r4.y}, rB = {r4.z, r4.w};                                                  \
                    v2f tA = {(float)acc[i][j][q4 * 4 + 0], (float)acc[i][j][q4 * 4 + 1]};                     \
                    v2f tB = {(float)acc[i][j][q4 * 4 + 2], (float)acc[i][j][q4 * 4 + 3]};                     \
                    const v2f na = {nal[j], nal[j]};                                                           \
                    if (ROWS_) { tA *= sA; tB *= sB; }                                                         \
                    if (FULL_) { const v2f b2 = {bet[j], bet[j]}; rA -= b2; rB -= b2; }                        \
                    v2f dA = tA * na + rA, dB = tB * na + rB;                                                  \
                    if (FULL_) { dA *= wA; dB *= wB; }                                                         \
                    cs2[j] += dA * dA; cs2[j] += dB * dB;                                                      \
                }                                                                                              \
            }                                                                                                  \
        }
        // ref_div >= 128: the wave's CJ * 32 columns share ONE reference column, so a row group needs one float4 of the
        // staged reference (not CJ); the next group's LDS values are fetched before the current group's arithmetic (the
        // epilogue was LDS-latency-bound: ~4.2 k cycles for ~300 VALU, with or without the int->float conversions).
#define STREAM_EPILOGUE_SAME(ROWS_, FULL_)                                                                     \
        {                                                                                                      \
            constexpr int NG_ = RI * 4;                                                                        \
            const float* rbase_ = rj[0] + wr * (BM3 / 2) + 4 * fkg;                                            \
            const float* sbase_ = s_rs + wr * (BM3 / 2) + 4 * fkg;                                             \
            const float* wbase_ = s_w + wr * (BM3 / 2) + 4 * fkg;                                              \
            float4 r_n = *reinterpret_cast<const float4*>(rbase_), s_n = make_float4(1.f, 1.f, 1.f, 1.f), w_n = s_n; \
            if (ROWS_) s_n = *reinterpret_cast<const float4*>(sbase_);                                         \
            if (FULL_) w_n = *reinterpret_cast<const float4*>(wbase_);                                         \
            _Pragma("unroll") for (int gi = 0; gi < NG_; ++gi) {                                               \
                const int i = gi >> 2, q4 = gi & 3;                                                            \
                const float4 r4 = r_n, s4 = s_n, w4 = w_n;                                                     \
                if (gi + 1 < NG_) {                                                                            \
                    const int o_ = ((gi + 1) >> 2) * 32 + 8 * ((gi + 1) & 3);                                  \
                    r_n = *reinterpret_cast<const float4*>(rbase_ + o_);                                       \
                    if (ROWS_) s_n = *reinterpret_cast<const float4*>(sbase_ + o_);                            \
                    if (FULL_) w_n = *reinterpret_cast<const float4*>(wbase_ + o_);                            \
                }                                                                                              \
                const v2f sA = {s4.x, s4.y}, sB = {s4.z, s4.w}, wA = {w4.x, w4.y}, wB = {w4.z, w4.w};          \
                _Pragma("unroll") for (int j = 0; j < CJ; ++j) {                                               \
                    v2f rA = {r4.x, r4.y}, rB = {r4.z, r4.w};                                                  \
                    v2f tA = {(float)acc[i][j][q4 * 4 + 0], (float)acc[i][j][q4 * 4 + 1]};                     \
                    v2f tB = {(float)acc[i][j][q4 * 4 + 2], (float)acc[i][j][q4 * 4 + 3]};                     \
                    const v2f na = {nal[j], nal[j]};                                                           \
                    if (ROWS_) { tA *= sA; tB *= sB; }                                                         \
                    if (FULL_) { const v2f b2 = {bet[j], bet[j]}; rA -= b2; rB -= b2; }                        \
                    v2f dA = tA * na + rA, dB = tB * na + rB;                                                  \
                    if (FULL_) { dA *= wA; dB *= wB; }                                                         \
                    cs2[j] += dA * dA; cs2[j] += dB * dB;                                                      \
                }                                                                                              \
            }                                                                                                  \
        }
        if (p.ref_div >= WCOLS) {
            if (full) STREAM_EPILOGUE_SAME(true, true)
            else if (rows) STREAM_EPILOGUE_SAME(true, false)
            else STREAM_EPILOGUE_SAME(false, false)
        } else {
            if (full) { STREAM_EPILOGUE(true, true) }
            else if (rows) { STREAM_EPILOGUE(true, false) }
            else { STREAM_EPILOGUE(false, false) }
        }
#undef STREAM_EPILOGUE_SAME
#undef STREAM_EPILOGUE
#pragma unroll
        for (int j = 0; j < CJ; ++j) {
            float cs = (cs2[j].x + cs2[j].y) * cm[j];
            cs += __shfl_xor(cs, 32);
            if (fkg == 0) s_red[wr][wc * WCOLS + j * 32 + frow] = cs;
        }
        TL_STAMP(4);
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // not __syncthreads(): its fence would drain the DMA
        TL_STAMP(5);
        if (tid < BN2) {
            const int col = n0 + tid;
            const float v = s_red[0][tid] + s_red[1][tid];         // 0 for columns past N (masked above)
            if (p.wg_acc) {
                if (gh != run_h) {
                    if (run_h >= 0) p.wg_acc[((int64_t)bid * p.gmod + run_h) * BN2 + tid] += run;
                    run = 0.0; run_h = gh;
                }
                run += (double)v;
            } else if (col < p.N) {
                p.partial[((((int64_t)g) * p.MT + tl.mt) * p.Npad + (col >> rsh)) * p.ref_div + (col & rmask)] = v;
            }
        }
        TL_STAMP(6);
        TL_STAMP(7);
    }
    if (p.wg_acc && tid < BN2 && run_h >= 0) p.wg_acc[((int64_t)bid * p.gmod + run_h) * BN2 + tid] += run;
#undef STREAM_STEP
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // drain the run-ahead steps before the LDS is released
#endif
}

// ------------------------------------------------------------------------------------------------ slab kernel
// Fourth form, for int8 searches with short K (a 256-column slab of the candidate operand, all of K, fits in 96 KiB):
//   * the workgroup (8 waves, one per CU) keeps 256 candidate COLUMNS resident in LDS and streams the other operand
//     (the fixed one: <= 3 MiB, so it stays in every XCD's L2) past them in units of 32 rows;
//   * each wave streams its OWN units through a private 3-stage LDS-DMA ring: the main loop has no workgroup barrier,
//     the waves drift apart and one wave's epilogue (VALU) runs under its SIMD partner's MFMAs;
//   * a lane owns 8 candidate columns (one per 32-column block) and keeps their squared-error sums in registers across
//     all the units it sees, so nothing is staged or reduced per unit: per slab the 8 waves combine once through LDS;
//   * the candidate operand is read from HBM exactly once, and the L2 -> LDS path carries 32 x K bytes per
//     32 x 256 outputs instead of (128 + 256) x K per 128 x 256.
// Work split: the (slab, unit) list is cut into equal contiguous ranges, one per workgroup; a slab cut by a range
// boundary gets one partial row per piece (MT = pieces), or, with wg_acc, everything a workgroup sees goes into its
// fp64 column sums.  Measured on the deit_small shapes (tools/lab): 1.8-2.0 PFLOP/s against 1.35 for k_gemm_stream.
template <int NREF, bool ROWS, int DT>
__global__ __launch_bounds__(512, 2) void k_gemm_slab(GemmArgs p) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int NSL = 3, UST = 32 * BK3, PW = 192;       // ring stages, bytes per stage, floats of per-wave operands
    constexpr int NP = 1 + (NREF == 4 ? 1 : 0) + (ROWS ? 1 : 0);   // operand requests per unit
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int frow = lane & 31, fkg = lane >> 5;
    const int nk = (int)((p.Kvb + BK3 - 1) / BK3), Kb = (int)p.Kb;
    uint8_t* slab = lds;                                                   // [nk][256 columns][64 bytes]
    uint8_t* ringw = lds + nk * BN2 * BK3 + w * (NSL * UST);               // this wave's ring
    uint8_t* tail = lds + nk * BN2 * BK3 + 8 * NSL * UST;
    float* parw = reinterpret_cast<float*>(tail) + w * PW;                 // [NREF <= 4][32] reference, [32] row scale, [32] row bias
    float* red = reinterpret_cast<float*>(tail + 8 * PW * 4);              // [8 waves][256 columns]

    const int rsh = __ffs(p.ref_div) - 1, rmask = p.ref_div - 1;
    const int n_eff = p.N >> rsh;
    const int U = p.slab_U, R = p.slab_R;
    const int u0 = (int)blockIdx.x * R, u1 = min(u0 + R, p.NT * U);
    if (u0 >= u1) return;
    const int s_first = u0 / U, s_last = (u1 - 1) / U;
    const int lrow = lane >> 2, lslot16 = ((lane & 3) ^ ((lane >> 4) & 3)) << 4;
    const int vo0 = lrow * Kb + lslot16, vo1 = (16 + lrow) * Kb + lslot16;

    // ---- this wave's units: in every slab segment [a, b) of the range it takes a + w, a + w + 8, ...
    auto seg_end = [&](int s) { return min(U, u1 - s * U); };
    const int a_first = u0 - s_first * U;
    // (only the range's first segment can start past 0 and only its last can end before U >= 24, so two probes suffice)
    auto probe = [&](int s, int& r) { r = (s == s_first ? a_first : 0) + w; return s <= s_last && r < seg_end(s); };
    auto first_from = [&](int s, int& os, int& orr) {
        int r;
        if (probe(s, r)) { os = s; orr = r; return true; }
        if (probe(s + 1, r)) { os = s + 1; orr = r; return true; }
        return false;
    };
    auto next_unit = [&](int& s, int& r) {
        if (r + 8 < seg_end(s)) { r += 8; return true; }
        return first_from(s + 1, s, r);
    };

    // ---- issue cursor: two K-steps ahead of the compute cursor, across units and slabs (the stream does not depend on
    // the slab).  Rows past M do not occur (M % 32 == 0 is a launch condition).
    __amdgpu_buffer_rsrc_t ra;
    int i_s = 0, i_r = 0, i_k = 0, i_slot = 0;
    bool i_more = first_from(s_first, i_s, i_r);
    auto issue_unit = [&]() {
        ra = __builtin_amdgcn_make_buffer_rsrc((void*)(p.A + (int64_t)i_r * 32 * Kb), 0, 0x7ffffffe, 0x00020000);
    };
    auto issue_step = [&]() {
        uint8_t* st = ringw + i_slot * UST;
        STREAM_DMA(ra, st, vo0, i_k * BK3);
        STREAM_DMA(ra, st + 1024, vo1, i_k * BK3);
        i_slot = i_slot == NSL - 1 ? 0 : i_slot + 1;
        if (++i_k == nk) {
            i_k = 0;
            if (i_more) { i_more = next_unit(i_s, i_r); if (i_more) issue_unit(); }   // past the last unit: harmless re-fetch
        }
    };
    // per-unit epilogue operands: reference slice [NREF][32 rows] (+ row scale | row bias), NP requests
    const uint32_t ref_bytes = (uint32_t)(((int64_t)(n_eff - 1) * p.ref_cs + p.M) * 4);
    const __amdgpu_buffer_rsrc_t rref = __builtin_amdgcn_make_buffer_rsrc((void*)p.ref, 0, (int)ref_bytes, 0x00020000);
    auto issue_params = [&](int s, int r) {
        const int m0 = r * 32, nj = s * NREF + fkg;
        const uint32_t oob = 0xfffffff0u;
        STREAM_DMA4(rref, parw, nj < n_eff ? (uint32_t)((nj * (int)p.ref_cs + m0 + frow) * 4) : oob);
        if (NREF == 4) STREAM_DMA4(rref, parw + 64, nj + 2 < n_eff ? (uint32_t)(((nj + 2) * (int)p.ref_cs + m0 + frow) * 4) : oob);
        if (ROWS) __builtin_amdgcn_global_load_lds((gas_ptr)((fkg ? p.row_bias : p.row_scale) + m0 + frow), (las_ptr)(parw + 128), 4, 0, 0);
    };

    if (i_more) {
        issue_unit();
        const int s0 = i_s, r0 = i_r;
        issue_step(); issue_step();
        issue_params(s0, r0);
    }

    int c_s = 0, c_r = 0;
    bool c_more = first_from(s_first, c_s, c_r);
    int st = 0;
    typename Acc<DT == 3 ? 1 : 0>::type acc[8];              // fp8 storage (DT = 3) accumulates in fp32: no cvt in the epilogue
    v2f cs2[8];                                            // running squared-error sums of this lane's 8 columns
#pragma unroll
    for (int b = 0; b < 8; ++b) cs2[b] = (v2f){0.0f, 0.0f};
    const bool bcols = p.bias && p.bi_c != 0;
    bool parked = false;

    for (int s = s_first; s <= s_last; ++s) {
        // ---- switch to slab s (workgroup-uniform): everyone is done with the old slab and with `red`
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        const int col_s = s * BN2;
        {
            const int64_t left = (int64_t)(p.N - col_s) * Kb;              // columns past N read as zero
            __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc((void*)(p.B + (int64_t)col_s * Kb), 0, (int)min(left, (int64_t)0x7ffffffe), 0x00020000);
            const int nreq = nk * 16;                                      // 16 columns x 64 bytes each
            for (int q = w; q < nreq; q += 8) {
                const int kt = q >> 4, c16 = q & 15;
                STREAM_DMA(rb, slab + kt * BN2 * BK3 + c16 * 1024, (c16 * 16 + lrow) * Kb + lslot16, kt * BK3);
            }
        }
        if (p.wg_acc && col_s + BN2 > p.N) {
            // partial last slab under per-workgroup accumulation: its padding columns must not count, so the running
            // sums are parked in `red` (unused in this mode) and this slab is summed on its own
#pragma unroll
            for (int b = 0; b < 8; ++b) {
                float v = cs2[b].x + cs2[b].y;
                v += __shfl_xor(v, 32);
                if (fkg == 0) red[w * BN2 + b * 32 + frow] = v;
                cs2[b] = (v2f){0.0f, 0.0f};
            }
            parked = true;
        }
        float nal[8], bn[NREF];
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            const int col = col_s + b * 32 + frow;
            const bool ok = col < p.N;
            const int colc = ok ? col : p.N - 1, ci = colc & rmask, ni = colc >> rsh;
            const float e_sa = p.sa[ci * p.sa_c], e_sb = p.sb[ci * p.sb_c + ni * p.sb_n];
            nal[b] = ok ? -(e_sa * p.sa_mul * e_sb) : 0.0f;
        }
#pragma unroll
        for (int j = 0; j < NREF; ++j) {
            const int nj = min(s * NREF + j, n_eff - 1);
            bn[j] = (p.bias && !bcols) ? p.bias[nj * p.bi_n] : 0.0f;
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");

        while (c_more && c_s == s) {
            // One K-step: may stay outstanding behind its two requests: the next step's two and, in a unit's first two
            // steps, the unit's NP operand requests (stores are not counted: stricter if one is still in flight).
#define SLAB_BODY(FIRST_)                                                                                       \
                uint4 b0n = lds_frag(Bs, swz3(frow, fkg)), b1n = lds_frag(Bs, swz3(frow, 2 + fkg));             \
                _Pragma("unroll") for (int b = 0; b < 8; ++b) {                                                 \
                    const uint4 b0 = b0n, b1 = b1n;                                                             \
                    if (b + 1 < 8) { b0n = lds_frag(Bs, swz3((b + 1) * 32 + frow, fkg)); b1n = lds_frag(Bs, swz3((b + 1) * 32 + frow, 2 + fkg)); } \
                    if constexpr (DT == 3) {                                                                    \
                        if (FIRST_) { _Pragma("unroll") for (int e_ = 0; e_ < 16; ++e_) acc[b][e_] = 0.0f; }    \
                        acc[b] = mma_fp8x64(a0, a1, b0, b1, acc[b]);                                            \
                    } else {                                                                                    \
                        if (FIRST_) acc[b] = mma0<0>(a0, b0); else mma<0>(a0, b0, acc[b]);                      \
                        mma<0>(a1, b1, acc[b]);                                                                 \
                    }                                                                                           \
                    if (b == 3) issue_step();                                                                   \
                    /* keep the one-block look-ahead: hoisting more reads costs the accumulators their VGPRs */ \
                    __builtin_amdgcn_sched_barrier(0);                                                          \
                }
#define SLAB_STEP(FIRST_, EARLY_, kt_)                                                                          \
            do {                                                                                                \
                if (EARLY_) {                                                                                   \
                    if (NP == 1) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");                               \
                    else if (NP == 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");                          \
                    else asm volatile("s_waitcnt vmcnt(5)" ::: "memory");                                       \
                } else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");                                         \
                const uint8_t* cur = ringw + st * UST;                                                          \
                const uint8_t* Bs = slab + (kt_) * BN2 * BK3;                                                   \
                const uint4 a0 = lds_frag(cur, swz3(frow, fkg)), a1 = lds_frag(cur, swz3(frow, 2 + fkg));       \
                SLAB_BODY(FIRST_)                                                                               \
                st = st == NSL - 1 ? 0 : st + 1;                                                                \
            } while (0)
            SLAB_STEP(true, true, 0);
            SLAB_STEP(false, true, 1);                                     // nk >= 2 is a launch condition
            for (int kt = 2; kt < nk; ++kt) SLAB_STEP(false, false, kt);
#undef SLAB_STEP
#undef SLAB_BODY
            // ---- epilogue: lane column = block b, lane & 31; rows 8*q4 + 4*fkg + e of the unit
            asm volatile("s_waitcnt vmcnt(4)" ::: "memory");               // all but the two run-ahead steps have landed
            // (a per-column bias is rare -- no int8 search of the calibrator has one -- so its values are re-read per
            // unit instead of living in eight more registers)
#define SLAB_EPILOGUE(BCOLS_)                                                                                   \
            float bco[BCOLS_ ? 8 : 1];                                                                          \
            if (BCOLS_) {                                                                                       \
                _Pragma("unroll") for (int b = 0; b < 8; ++b) {                                                 \
                    const int colc = min(col_s + b * 32 + frow, p.N - 1);                                       \
                    bco[b] = p.bias[(colc & rmask) * p.bi_c + (colc >> rsh) * p.bi_n];                          \
                }                                                                                               \
            }                                                                                                   \
            _Pragma("unroll") for (int q4 = 0; q4 < 4; ++q4) {                                                  \
                const int ro = 8 * q4 + 4 * fkg;                                                                \
                v2f sA = {1.f, 1.f}, sB = {1.f, 1.f};                                                           \
                float4 rb4 = make_float4(0.f, 0.f, 0.f, 0.f);                                                   \
                if (ROWS) { const float4 s4 = lds_f4(parw, 128 + ro); sA = (v2f){s4.x, s4.y}; sB = (v2f){s4.z, s4.w}; rb4 = lds_f4(parw, 160 + ro); } \
                v2f rA[NREF], rB[NREF];                                                                         \
                _Pragma("unroll") for (int j = 0; j < NREF; ++j) {                                              \
                    const float4 r4 = lds_f4(parw, j * 32 + ro);                                                \
                    rA[j] = (v2f){r4.x - rb4.x - bn[j], r4.y - rb4.y - bn[j]};                                  \
                    rB[j] = (v2f){r4.z - rb4.z - bn[j], r4.w - rb4.w - bn[j]};                                  \
                }                                                                                               \
                _Pragma("unroll") for (int b = 0; b < 8; ++b) {                                                 \
                    const int j = (b * NREF) >> 3;                                                              \
                    v2f tA = {(float)acc[b][q4 * 4 + 0], (float)acc[b][q4 * 4 + 1]};                            \
                    v2f tB = {(float)acc[b][q4 * 4 + 2], (float)acc[b][q4 * 4 + 3]};                            \
                    const v2f na = {nal[b], nal[b]};                                                            \
                    if (ROWS) { tA *= sA; tB *= sB; }                                                           \
                    v2f dA = tA * na + rA[j], dB = tB * na + rB[j];                                             \
                    if (BCOLS_) { const v2f b2 = {bco[b], bco[b]}; dA -= b2; dB -= b2; }                        \
                    cs2[b] += dA * dA; cs2[b] += dB * dB;                                                       \
                }                                                                                               \
            }
#if defined(GEMM_LAB_NO_EPI)   // tools/lab: time the main loop alone
            _Pragma("unroll") for (int b = 0; b < 8; ++b) cs2[b] += (v2f){(float)acc[b][0], (float)acc[b][15]};
#else
            if (bcols) { SLAB_EPILOGUE(true) } else { SLAB_EPILOGUE(false) }
#endif
#undef SLAB_EPILOGUE
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");             // parw fully read before the next unit's operands land
            c_more = next_unit(c_s, c_r);
            if (c_more) issue_params(c_s, c_r);
        }

        // ---- end of this workgroup's share of slab s
        // columns past N (last slab only): operand and reference read as zero, but a folded bias does not -- mask them
        if (p.wg_acc) {
            if (col_s + BN2 > p.N) {
#pragma unroll
                for (int b = 0; b < 8; ++b) if (col_s + b * 32 + frow >= p.N) cs2[b] = (v2f){0.0f, 0.0f};
            }
        } else {
#pragma unroll
            for (int b = 0; b < 8; ++b) {
                float cs = (col_s + b * 32 + frow < p.N) ? cs2[b].x + cs2[b].y : 0.0f;
                cs += __shfl_xor(cs, 32);
                if (fkg == 0) red[w * BN2 + b * 32 + frow] = cs;
                cs2[b] = (v2f){0.0f, 0.0f};
            }
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            if (tid < BN2) {
                float v = 0.0f;
#pragma unroll
                for (int ww = 0; ww < 8; ++ww) v += red[ww * BN2 + tid];
                const int col = col_s + tid;
                const int piece = (int)blockIdx.x - (s * U) / R;           // 0 for the workgroup that holds the slab's first unit
                if (col < p.N) p.partial[(((int64_t)piece) * p.Npad + (col >> rsh)) * p.ref_div + (col & rmask)] = v;
            }
        }
    }
    if (p.wg_acc) {
        // column sums of everything this workgroup saw (fp32 per lane: <= a few thousand terms; fp64 from here on):
        // lanes pair up, then a fixed wave order through LDS
        double* redd = reinterpret_cast<double*>(slab);                    // the slab is dead now
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            double v = (double)cs2[b].x + (double)cs2[b].y;
            v += __shfl_xor(v, 32);
            if (fkg == 0) redd[w * BN2 + b * 32 + frow] = v + (parked ? (double)red[w * BN2 + b * 32 + frow] : 0.0);
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (tid < BN2) {
            double v = 0.0;
#pragma unroll
            for (int ww = 0; ww < 8; ++ww) v += redd[ww * BN2 + tid];
            for (int h = 0; h < p.gmod; ++h) p.wg_acc[((int64_t)blockIdx.x * p.gmod + h) * BN2 + tid] = h == 0 ? v : 0.0;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
}

// ------------------------------------------------------------------------------------------------ group kernel
// Fifth form, for the attention q.k^T searches: many small groups (image x head), one 64-byte K-step, 129..224 rows.
// k_gemm_stream spends such a tile on its epilogue and its per-tile set-up (2 MFMAs per 32 x 32 outputs against ~32
// VALU), at 0.4 PFLOP/s.  Here a workgroup is 7 consumer waves + 1 loader wave:
//   * consumer w keeps the fragments of row block w of the current group in REGISTERS (8 VGPRs) for a whole item
//     (= group x chunk of candidate columns) together with the 16 reference values per reference column it needs;
//   * the loader wave alone issues the LDS-DMA of the candidate columns (stages of 256 columns = 16 KiB, 3-stage ring)
//     and is the only wave that counts vmcnt; one workgroup barrier per stage hands a stage over;
//   * per 32 x 32 block a consumer does 2 ds_read_b128, 2 MFMAs and the 32-VALU epilogue; column sums stay in registers
//     for the item, then go through LDS into per-workgroup fp64 sums per head (fixed order: no atomics).
// Output: the wg_acc layout of k_gemm_stream ([workgroup][head][256] fp64), same number of workgroups.
typedef unsigned v4u __attribute__((ext_vector_type(4)));

template <int NJ, int DT>
__global__ __launch_bounds__(512, NJ == 8 ? 2 : 4) void k_gemm_grp(GemmArgs p) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int NSG = 3, SB = 8, SBYTES = SB * 32 * BK3;     // ring stages, blocks per stage, bytes per stage
    constexpr int P = NJ * 32;
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    uint8_t* ring = lds;                                                        // [NSG][256 columns][64 bytes]
    float* red = reinterpret_cast<float*>(lds + NSG * SBYTES);                  // [7 waves][256]
    double* accl = reinterpret_cast<double*>(lds + NSG * SBYTES + 7 * 256 * 4); // [gmod][256]
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int frow = lane & 31, fkg = lane >> 5;
    const int Kb = (int)p.Kb;
    const int NB = p.N >> 5;                                  // 32-column blocks per group
    const int CB = p.slab_R, NCH = p.slab_U;                  // blocks per chunk, chunks per group
    const int items = p.G * NCH;
    const int n_eff = p.N / P;
    for (int i = tid; i < p.gmod * 256; i += 512) accl[i] = 0.0;
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");

    // ---------------- issue side (every wave): the stages of this workgroup's items, two ahead of the compute side.
    // A stage is 16 requests of 16 columns x 64 bytes.  Issuing one costs a wave ~100 cycles, so a single loader wave
    // (1 600 cycles per stage against ~900 of consumer work) was the bottleneck: the loader keeps requests 7..15 and
    // consumer w issues request w (an even split, two per wave, measured 5 % slower).
    const int lrow = lane >> 2, lslot16 = ((lane & 3) ^ ((lane >> 4) & 3)) << 4;
    int li = blockIdx.x, lst = 0, slot = 0;                   // item / stage cursor of the issue side
    int ahead = 0;                                             // stages issued and not yet handed over
    auto stages_of = [&](int item) { const int c = item % NCH; return (min(CB, NB - c * CB) + SB - 1) / SB; };
    auto issue = [&]() {
        const int g = li / NCH, c = li - g * NCH;
        const int col0 = (c * CB + lst * SB) * 32;
        const int64_t left = (int64_t)(p.N - col0) * Kb;
        const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc((void*)(p.B + (int64_t)g * p.sBg + (int64_t)col0 * Kb), 0,
                                                                            (int)(left < 0 ? 0 : min(left, (int64_t)0x7ffffffe)), 0x00020000);
        uint8_t* st = ring + slot * SBYTES;
        if (w == 7) {
#pragma unroll
            for (int q = 7; q < 16; ++q) STREAM_DMA(rb, st + q * 1024, (q * 16 + lrow) * Kb + lslot16, 0);
        } else STREAM_DMA(rb, st + w * 1024, (w * 16 + lrow) * Kb + lslot16, 0);
        slot = slot == NSG - 1 ? 0 : slot + 1;
        if (++lst == stages_of(li)) { lst = 0; li += gridDim.x; }
        ++ahead;
    };
    if (li < items) issue();
    if (li < items) issue();

    if (w == 7) {
        // ---------------- loader wave: nothing but its share of the requests
        for (int item = blockIdx.x; item < items; item += gridDim.x) {
            const int ns = stages_of(item);
            for (int t = 0; t < ns; ++t) {
                if (ahead >= 2) asm volatile("s_waitcnt vmcnt(9)" ::: "memory");    // the following stage's nine may be in flight
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                asm volatile("s_barrier" ::: "memory");
                --ahead;
                if (li < items) issue();
            }
            asm volatile("s_barrier" ::: "memory");                        // item end: column sums are in `red`
        }
    } else {
        // ---------------- consumers: row block w
        const int row0 = w * 32;
        const bool rowblock_live = row0 < p.M;                 // (always true for the launches this kernel takes)
        int st = 0;                                            // ring slot of the next stage (runs on across items)
        for (int item = blockIdx.x; item < items; item += gridDim.x) {
            const int g = item / NCH, c = item - g * NCH, gh = g % p.gmod;
            const int blk0 = c * CB, nblk = min(CB, NB - blk0), ns = (nblk + SB - 1) / SB;
            // fragments of this row block (rows past M are zero), column factors of the group's head
            uint4 a0, a1;
            {
                const bool ok = row0 + frow < p.M;
                const uint8_t* ar = p.A + (int64_t)g * p.sAg + (int64_t)min(row0 + frow, p.M - 1) * Kb;
                a0 = *reinterpret_cast<const uint4*>(ar + fkg * 16);
                a1 = *reinterpret_cast<const uint4*>(ar + 32 + fkg * 16);
                if (!ok) { a0 = make_uint4(0, 0, 0, 0); a1 = a0; }
            }
            float nal[NJ];
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const int ci = j * 32 + frow;
                nal[j] = -(p.sa[ci * p.sa_c + gh * p.sa_g] * p.sa_mul * p.sb[ci * p.sb_c + gh * p.sb_g]);
            }
            // reference slice of one reference column: rows row0 + 8*q + 4*fkg + {0..3}, zero past M.  One 16-byte load
            // per q from a uniform column base + a per-lane row offset clamped to M - 4; the ragged (last) row block then
            // shifts the elements it still owns into place.
            // (buffer loads: uniform column offset in an SGPR + four loop-invariant lane offsets -- with flat pointers the
            // compiler kept one 64-bit address per call site alive and spilled them)
            const uint32_t ref_bytes = (uint32_t)(((int64_t)(n_eff - 1) * p.ref_cs + p.M) * 4);   // past it a load returns 0
            const __amdgpu_buffer_rsrc_t rrg = __builtin_amdgcn_make_buffer_rsrc((void*)(p.ref + (int64_t)g * p.sRg), 0, (int)ref_bytes, 0x00020000);
            const bool full_rows = row0 + 32 <= p.M;          // wave-uniform
            int roff[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) roff[q] = min(row0 + 8 * q + 4 * fkg, p.M - 4) * 4;
            v2f cs2[NJ];
#pragma unroll
            for (int j = 0; j < NJ; ++j) cs2[j] = (v2f){0.0f, 0.0f};
            // The stage loop exists twice (whole / ragged row block, chosen per wave) so that its body has no branch: with
            // one, the compiler's waitcnt pass drains the reference prefetch it has just issued.
            auto run = [&](auto full_tag) {
                constexpr bool FULL = decltype(full_tag)::value;
                auto load_ref = [&](int n, float4 (&r)[4]) {
                    const int coff = n < n_eff ? n * (int)p.ref_cs * 4 : 0x7ffffff0;   // (uniform) past the last column: zeros
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const v4u uu = __builtin_amdgcn_raw_buffer_load_b128(rrg, roff[q], coff, 0);
                        r[q] = make_float4(__uint_as_float(uu.x), __uint_as_float(uu.y), __uint_as_float(uu.z), __uint_as_float(uu.w));
                    }
                };
                // ragged row block: the loaded group of four starts at min(row, M - 4); move the elements this lane owns
                // into place and zero the rows past M.  Done when a set becomes the current one, NOT at the load: there
                // it would wait for the load it has just issued (and the whole workgroup waits for this wave at the barrier).
                auto fix_ref = [&](float4 (&r)[4]) {
                    if (FULL) return;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int rr = row0 + 8 * q + 4 * fkg;
                        const float4 u = r[q];
                        const int sh = rr * 4 - roff[q];       // 0 unless the group of four crosses M (then 4, 8 or 12 bytes)
                        const float e0 = sh == 0 ? u.x : sh == 4 ? u.y : sh == 8 ? u.z : u.w;
                        const float e1 = sh == 0 ? u.y : sh == 4 ? u.z : u.w;
                        const float e2 = sh == 0 ? u.z : u.w;
                        r[q] = make_float4(rr < p.M ? e0 : 0.f, rr + 1 < p.M ? e1 : 0.f, rr + 2 < p.M ? e2 : 0.f, rr + 3 < p.M ? u.w : 0.f);
                    }
                };
                int n = blk0 / NJ;                             // chunks start on a reference column (CB % 8 == 0, NJ | 8)
                // two register sets take turns (a stage holds an even number of reference columns when NJ <= 4, so which
                // set a block reads is known at compile time; with NJ = 8 the sets are swapped by copying)
                float4 rr2[2][4];
                load_ref(n, rr2[0]);
                load_ref(n + 1, rr2[1]);
                fix_ref(rr2[0]);
                for (int t = 0; t < ns; ++t) {
                    // this wave's request of the stage has landed once only what was issued after it can be outstanding:
                    // the following stage's request and the R reference loads of the previous stage (the first stage of
                    // an item comes after the item's set-up loads: drain)
                    constexpr int R = (SB / NJ) * 4;
                    if (t == 0 || ahead < 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    else if (R == 4) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
                    else if (R == 8) asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
                    else asm volatile("s_waitcnt vmcnt(17)" ::: "memory");
                    asm volatile("s_barrier" ::: "memory");
                    --ahead;
                    if (li < items) issue();
                    const uint8_t* cur = ring + st * SBYTES;
                    uint4 b0n = lds_frag(cur, swz3(frow, fkg)), b1n = lds_frag(cur, swz3(frow, 2 + fkg));
#pragma unroll
                    for (int b = 0; b < SB; ++b) {
                        const int j = b % NJ;
                        const int set = NJ == 8 ? 0 : (b / NJ) & 1;
                        const uint4 b0 = b0n, b1 = b1n;
                        if (b + 1 < SB) { b0n = lds_frag(cur, swz3((b + 1) * 32 + frow, fkg)); b1n = lds_frag(cur, swz3((b + 1) * 32 + frow, 2 + fkg)); }
                        typename Acc<DT == 3 ? 1 : 0>::type acc;      // fp8 operands accumulate in fp32: no conversion below
                        if constexpr (DT == 3) {
                            v16f z_;
#pragma unroll
                            for (int e = 0; e < 16; ++e) z_[e] = 0.0f;
                            acc = mma_fp8x64(a0, a1, b0, b1, z_);
                        } else {
                            acc = mma0<0>(a0, b0);
                            mma<0>(a1, b1, acc);
                        }
                        const v2f na = {nal[j], nal[j]};
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const v2f tA = {(float)acc[q * 4 + 0], (float)acc[q * 4 + 1]}, tB = {(float)acc[q * 4 + 2], (float)acc[q * 4 + 3]};
                            const v2f rA = {rr2[set][q].x, rr2[set][q].y}, rB = {rr2[set][q].z, rr2[set][q].w};
                            const v2f dA = tA * na + rA, dB = tB * na + rB;
                            cs2[j] += dA * dA; cs2[j] += dB * dB;
                        }
                        if (j == NJ - 1) {                     // next block starts the next reference column
                            ++n;
                            if (NJ == 8) {
#pragma unroll
                                for (int q = 0; q < 4; ++q) rr2[0][q] = rr2[1][q];
                                fix_ref(rr2[0]);
                                load_ref(n + 1, rr2[1]);
                            } else {
                                fix_ref(rr2[set ^ 1]);         // the other set holds column n: it becomes the current one
                                load_ref(n + 1, rr2[set]);     // this set is free now
                            }
                        }
                    }
                    st = st == NSG - 1 ? 0 : st + 1;
                }
            };
            if (!rowblock_live) {                              // (never for the launches this kernel takes)
                for (int t = 0; t < ns; ++t) {
                    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
                    --ahead;
                    if (li < items) issue();
                    st = st == NSG - 1 ? 0 : st + 1;
                }
            } else if (full_rows) run(std::true_type{});
            else run(std::false_type{});
            // item end: pair up the lanes, park the sums, one thread per candidate adds them in a fixed wave order
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                float cs = cs2[j].x + cs2[j].y;
                cs += __shfl_xor(cs, 32);
                if (fkg == 0) red[w * 256 + j * 32 + frow] = cs;
            }
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            if (tid < P) {
                double v = 0.0;
                const int nw = min(7, (p.M + 31) / 32);
                for (int ww = 0; ww < nw; ++ww) v += (double)red[ww * 256 + tid];
                accl[gh * 256 + tid] += v;
            }
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    for (int i = tid; i < p.gmod * 256; i += 512) p.wg_acc[(int64_t)blockIdx.x * p.gmod * 256 + i] = accl[i];
#endif
}

// The same organisation for groups with several K-steps (the softmax.v weight search: bf16, K = 197 keys = 7 steps, 197
// attention rows per group, 64 x P candidate columns): a consumer keeps the NK x 2 fragments of its row block in
// registers for an item, a stage is 2 column blocks x all of K (NK x 4 requests of 16 columns x 64 bytes; consumer w
// issues two of them, the loader the rest), and a block is 2 NK MFMAs against the same 16-value epilogue.
template <int NJ, int NK>
__global__ __launch_bounds__(512, 2) void k_gemm_grpk(GemmArgs p) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int NSG = 3, SB = 2, SBYTES = NK * SB * 32 * BK3;   // ring stages, blocks per stage, bytes per stage
    constexpr int RQ = NK * SB * 2, CQ = 2;                    // requests per stage, requests per consumer
    constexpr int STG = NJ / SB > 0 ? NJ / SB : 1;             // stages per reference column
    constexpr int P = NJ * 32;
    static_assert(RQ > 7 * CQ, "the loader needs a share");
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    uint8_t* ring = lds;                                                        // [NSG][NK][64 columns][64 bytes]
    float* red = reinterpret_cast<float*>(lds + NSG * SBYTES);                  // [7 waves][P]
    double* accl = reinterpret_cast<double*>(lds + NSG * SBYTES + 7 * P * 4);   // [gmod][256]
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int frow = lane & 31, fkg = lane >> 5;
    const int Kb = (int)p.Kb;
    const int NB = p.N >> 5;
    const int CB = p.slab_R, NCH = p.slab_U;
    const int items = p.G * NCH;
    const int n_eff = p.N / P;
    for (int i = tid; i < p.gmod * 256; i += 512) accl[i] = 0.0;
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");

    const int lrow = lane >> 2, lslot16 = ((lane & 3) ^ ((lane >> 4) & 3)) << 4;
    int li = blockIdx.x, lst = 0, slot = 0, ahead = 0;
    auto stages_of = [&](int item) { const int c = item % NCH; return (min(CB, NB - c * CB) + SB - 1) / SB; };
    auto issue = [&]() {
        const int g = li / NCH, c = li - g * NCH;
        const int col0 = (c * CB + lst * SB) * 32;
        const int64_t left = (int64_t)(p.N - col0) * Kb;
        const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc((void*)(p.B + (int64_t)g * p.sBg + (int64_t)col0 * Kb), 0,
                                                                            (int)(left < 0 ? 0 : min(left, (int64_t)0x7ffffffe)), 0x00020000);
        uint8_t* st = ring + slot * SBYTES;
        // request r: K-step r / 4, columns 16 * (r % 4) .. + 15
        if (w == 7) {
#pragma unroll
            for (int r = 7 * CQ; r < RQ; ++r) STREAM_DMA(rb, st + r * 1024, ((r & 3) * 16 + lrow) * Kb + lslot16, (r >> 2) * BK3);
        } else {
#pragma unroll
            for (int i = 0; i < CQ; ++i) {
                const int r = w * CQ + i;
                STREAM_DMA(rb, st + r * 1024, ((r & 3) * 16 + lrow) * Kb + lslot16, (r >> 2) * BK3);
            }
        }
        slot = slot == NSG - 1 ? 0 : slot + 1;
        if (++lst == stages_of(li)) { lst = 0; li += gridDim.x; }
        ++ahead;
    };
    if (li < items) issue();
    if (li < items) issue();

    if (w == 7) {
        for (int item = blockIdx.x; item < items; item += gridDim.x) {
            const int ns = stages_of(item);
            for (int t = 0; t < ns; ++t) {
                if (ahead >= 2) {
                    if (RQ - 7 * CQ == 14) asm volatile("s_waitcnt vmcnt(14)" ::: "memory");
                    else if (RQ - 7 * CQ == 7) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
                    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                } else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                asm volatile("s_barrier" ::: "memory");
                --ahead;
                if (li < items) issue();
            }
            asm volatile("s_barrier" ::: "memory");
        }
    } else {
        const int row0 = w * 32;
        const bool rowblock_live = row0 < p.M;
        int st = 0;
        for (int item = blockIdx.x; item < items; item += gridDim.x) {
            const int g = item / NCH, c = item - g * NCH, gh = g % p.gmod;
            const int blk0 = c * CB, nblk = min(CB, NB - blk0), ns = (nblk + SB - 1) / SB;
            uint4 af[NK][2];
            {
                const bool ok = row0 + frow < p.M;
                const uint8_t* ar = p.A + (int64_t)g * p.sAg + (int64_t)min(row0 + frow, p.M - 1) * Kb;
#pragma unroll
                for (int kt = 0; kt < NK; ++kt) {
                    af[kt][0] = *reinterpret_cast<const uint4*>(ar + kt * BK3 + fkg * 16);
                    af[kt][1] = *reinterpret_cast<const uint4*>(ar + kt * BK3 + 32 + fkg * 16);
                    if (!ok) { af[kt][0] = make_uint4(0, 0, 0, 0); af[kt][1] = af[kt][0]; }
                }
            }
            float nal[NJ];
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const int ci = j * 32 + frow;
                nal[j] = -(p.sa[ci * p.sa_c + gh * p.sa_g] * p.sa_mul * p.sb[ci * p.sb_c + gh * p.sb_g]);
            }
            const uint32_t ref_bytes = (uint32_t)(((int64_t)(n_eff - 1) * p.ref_cs + p.M) * 4);
            const __amdgpu_buffer_rsrc_t rrg = __builtin_amdgcn_make_buffer_rsrc((void*)(p.ref + (int64_t)g * p.sRg), 0, (int)ref_bytes, 0x00020000);
            const bool full_rows = row0 + 32 <= p.M;
            int roff[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) roff[q] = min(row0 + 8 * q + 4 * fkg, p.M - 4) * 4;
            v2f cs2[NJ];
#pragma unroll
            for (int j = 0; j < NJ; ++j) cs2[j] = (v2f){0.0f, 0.0f};
            auto run = [&](auto full_tag) {
                constexpr bool FULL = decltype(full_tag)::value;
                auto load_ref = [&](int n, float4 (&r)[4]) {
                    const int coff = n < n_eff ? n * (int)p.ref_cs * 4 : 0x7ffffff0;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const v4u uu = __builtin_amdgcn_raw_buffer_load_b128(rrg, roff[q], coff, 0);
                        r[q] = make_float4(__uint_as_float(uu.x), __uint_as_float(uu.y), __uint_as_float(uu.z), __uint_as_float(uu.w));
                    }
                };
                auto fix_ref = [&](float4 (&r)[4]) {
                    if (FULL) return;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int rr = row0 + 8 * q + 4 * fkg;
                        const float4 u = r[q];
                        const int sh = rr * 4 - roff[q];
                        const float e0 = sh == 0 ? u.x : sh == 4 ? u.y : sh == 8 ? u.z : u.w;
                        const float e1 = sh == 0 ? u.y : sh == 4 ? u.z : u.w;
                        const float e2 = sh == 0 ? u.z : u.w;
                        r[q] = make_float4(rr < p.M ? e0 : 0.f, rr + 1 < p.M ? e1 : 0.f, rr + 2 < p.M ? e2 : 0.f, rr + 3 < p.M ? u.w : 0.f);
                    }
                };
                int n = blk0 / NJ;
                float4 rc[4], rn[4];                           // current / next reference column
                load_ref(n, rc);
                load_ref(n + 1, rn);
                fix_ref(rc);
                for (int tg = 0; tg < ns; tg += STG) {
#pragma unroll
                    for (int s = 0; s < STG; ++s) {
                        // behind this wave's CQ requests of the stage: the next stage's CQ and the reference loads of
                        // the two stages in between (at least 8 / 4 / 0 of them for NJ = 2 / 4 / 8)
                        if ((tg == 0 && s == 0) || ahead < 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                        else if (NJ == 2) { if (CQ == 2) asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(11)" ::: "memory"); }
                        else if (NJ == 4) { if (CQ == 2) asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); }
                        else { if (CQ == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); }
                        asm volatile("s_barrier" ::: "memory");
                        --ahead;
                        if (li < items) issue();
                        const uint8_t* cur = ring + st * SBYTES;
#pragma unroll
                        for (int b = 0; b < SB; ++b) {
                            const int j = (s * SB + b) % NJ;
                            v16f acc;
#pragma unroll
                            for (int kt = 0; kt < NK; ++kt) {
                                const uint8_t* ks = cur + kt * (SB * 32 * BK3);
                                const uint4 b0 = lds_frag(ks, swz3(b * 32 + frow, fkg)), b1 = lds_frag(ks, swz3(b * 32 + frow, 2 + fkg));
                                if (kt == 0) acc = mma0<1>(af[0][0], b0); else mma<1>(af[kt][0], b0, acc);
                                mma<1>(af[kt][1], b1, acc);
                            }
                            const v2f na = {nal[j], nal[j]};
#pragma unroll
                            for (int q = 0; q < 4; ++q) {
                                const v2f tA = {acc[q * 4 + 0], acc[q * 4 + 1]}, tB = {acc[q * 4 + 2], acc[q * 4 + 3]};
                                const v2f rA = {rc[q].x, rc[q].y}, rB = {rc[q].z, rc[q].w};
                                const v2f dA = tA * na + rA, dB = tB * na + rB;
                                cs2[j] += dA * dA; cs2[j] += dB * dB;
                            }
                            if (j == NJ - 1) {                 // next block starts the next reference column
                                ++n;
#pragma unroll
                                for (int q = 0; q < 4; ++q) rc[q] = rn[q];
                                fix_ref(rc);
                                load_ref(n + 1, rn);
                            }
                        }
                        st = st == NSG - 1 ? 0 : st + 1;
                    }
                }
            };
            if (!rowblock_live) {
                for (int t = 0; t < ns; ++t) {
                    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
                    --ahead;
                    if (li < items) issue();
                    st = st == NSG - 1 ? 0 : st + 1;
                }
            } else if (full_rows) run(std::true_type{});
            else run(std::false_type{});
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                float cs = cs2[j].x + cs2[j].y;
                cs += __shfl_xor(cs, 32);
                if (fkg == 0) red[w * P + j * 32 + frow] = cs;
            }
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            if (tid < P) {
                double v = 0.0;
                const int nw = min(7, (p.M + 31) / 32);
                for (int ww = 0; ww < nw; ++ww) v += (double)red[ww * P + tid];
                accl[gh * 256 + tid] += v;
            }
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    for (int i = tid; i < p.gmod * 256; i += 512) p.wg_acc[(int64_t)blockIdx.x * p.gmod * 256 + i] = accl[i];
#endif
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

// Two-stage finish for the candidate-innermost layout when the score does not keep the column axis (activation and
// attention searches: 10^4..10^5 terms per output).  Stage 1 walks [rows][cin] with the candidates across adjacent
// threads (every load instruction reads whole 256..1024-byte rows; the one-block-per-output form above strides by cin
// floats and reached 0.3 TB/s), 128 rows per block in fp64; stage 2 adds the per-block sums in a fixed order.
constexpr int FSEG = 128;
__global__ __launch_bounds__(256) void k_finish_rows(FinishArgs p, double* part2, int nseg) {
    __shared__ double sm[256];
    const int cin = p.cin, lanes = 256 / cin;                 // cin is 64, 128 or 256
    const int c = threadIdx.x % cin, rl = threadIdx.x / cin;
    const int seg = blockIdx.x;
    const int64_t combo = blockIdx.y;                         // g * MT + mt
    const float* base = p.partial + (combo * p.Npad) * cin + c;
    const int n_hi = min(p.N, (seg + 1) * FSEG);
    double acc = 0.0;
    for (int n = seg * FSEG + rl; n < n_hi; n += lanes) acc += (double)base[(int64_t)n * cin];
    sm[threadIdx.x] = acc;
    __syncthreads();
    if (rl == 0) {
        for (int l = 1; l < lanes; ++l) acc += sm[l * cin + c];
        part2[(combo * nseg + seg) * cin + c] = acc;
    }
}

// one wavefront per output (c, h): terms = (image, [head], m-tile, segment) in a fixed lane-strided order
__global__ __launch_bounds__(256) void k_finish_stage2(FinishArgs p, const double* part2, int nseg) {
    const int nh = p.keep_h ? p.gmod : 1;
    const int lane = threadIdx.x & 63;
    const int oid = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (oid >= p.C * nh) return;
    const int h = oid % nh, c = oid / nh;
    const int imgs = p.G / p.gmod, h_lo = p.keep_h ? h : 0, h_cnt = p.keep_h ? 1 : p.gmod;
    const int per_g = p.MT * nseg;
    const int64_t total = (int64_t)imgs * h_cnt * per_g;
    double acc = 0.0;
    for (int64_t i = lane; i < total; i += 64) {
        const int r = (int)(i % per_g);
        const int64_t t = i / per_g;
        const int g = (int)(t / h_cnt) * p.gmod + h_lo + (int)(t % h_cnt);
        acc += part2[((int64_t)g * per_g + r) * p.cin + c];
    }
#pragma unroll
    for (int s = 32; s > 0; s >>= 1) acc += __shfl_xor(acc, s);
    if (lane == 0) p.scores[(int64_t)c * nh + h] = (float)(-p.norm * acc);
}

// Finish for per-workgroup accumulators acc[wg][head][256] (fp64): one 256-thread block per output (candidate, head);
// terms = (workgroup, [head], column replica with col % cin == candidate), thread-strided, then a fixed LDS tree.
// (One wavefront per output looped 16 times over dependent loads: 10 us; this form is bound by the launch itself.)
__global__ __launch_bounds__(256) void k_finish_wgacc(FinishArgs p, const double* acc, int nwg) {
    __shared__ double red[4];
    const int nh = p.keep_h ? p.gmod : 1;
    const int oid = blockIdx.x;
    const int h = oid % nh, c = oid / nh;
    const int reps = 256 / p.cin, h_lo = p.keep_h ? h : 0, h_cnt = p.keep_h ? 1 : p.gmod;
    const int per_wg = h_cnt * reps;
    const int64_t total = (int64_t)nwg * per_wg;
    double sum = 0.0;
    for (int64_t i = threadIdx.x; i < total; i += 256) {
        const int r = (int)(i % per_wg);
        const int64_t wg = i / per_wg;
        const int hh = h_lo + r / reps, rep = r % reps;
        sum += acc[(wg * p.gmod + hh) * 256 + rep * p.cin + c];
    }
#pragma unroll
    for (int s = 32; s > 0; s >>= 1) sum += __shfl_xor(sum, s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = sum;
    __syncthreads();
    if (threadIdx.x == 0) p.scores[(int64_t)c * nh + h] = (float)(-p.norm * ((red[0] + red[1]) + (red[2] + red[3])));
}

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

struct Layout { int big, tm, wide, MT, NT, Npad, c_eff, n_eff, stream, acc, wgs, slab, slab_U, slab_R; int64_t elems; };

// Wide (one workgroup per CU, 192/256-row tile) form of the streaming kernel: long K only -- 16+ K-steps, where the
// L2 -> LDS path bounds the main loop and the un-overlapped epilogue is < 10 % of a tile.
static int pick_wide(int M, int64_t kvalid_bytes) {
    static const int use_wide = getenv("ADALOG_GEMM_WIDE") ? atoi(getenv("ADALOG_GEMM_WIDE")) : 1;
    static const int min_k = getenv("ADALOG_GEMM_WIDE_MINK") ? atoi(getenv("ADALOG_GEMM_WIDE_MINK")) : 1024;
    if (!use_wide || kvalid_bytes < min_k || M < 192) return 0;
    const int64_t pad4 = (int64_t)cdiv(M, 256) * 256, pad3 = (int64_t)cdiv(M, 192) * 192, pad2 = (int64_t)cdiv(M, 128) * 128;
    const int ri = pad4 <= pad3 + pad3 / 32 ? 4 : 3;                      // 256 rows unless 192 pads > 3 % less
    const int64_t padw = ri == 4 ? pad4 : pad3;
    return padw <= pad2 + pad2 / 8 ? ri : 0;                              // not if it pads > 12 % more than 128-row tiles
}

// reduce_cols with ref_div > 1 asks for per-workgroup accumulation, which only the streaming kernel provides: when that
// kernel is not eligible the launch falls back to per-tile partials (candidate innermost), and the layout says so.
static Layout layout_of(int M, int N, int C, int G, int gmod, int ref_div, int reduce_cols, bool scoring = true,
                        int64_t kvalid_bytes = 0, int64_t kb = 0, bool ref_transposed = false, int dtype = -1) {
    static const int use_stream = getenv("ADALOG_GEMM_STREAM") ? atoi(getenv("ADALOG_GEMM_STREAM")) : 1;
    static const int use_wgacc = getenv("ADALOG_GEMM_WGACC") ? atoi(getenv("ADALOG_GEMM_WGACC")) : 1;
    Layout L{};
    L.big = (C == 1);
    L.tm = L.big ? pick_tm(M, scoring) : 2;
    const bool cand_cols = L.big && scoring && ref_transposed && (ref_div == 64 || ref_div == 128 || ref_div == 256);
    if (cand_cols && use_stream) {
        L.wide = pick_wide(M, kvalid_bytes);
        if (L.wide) L.tm = L.wide;
    }
    L.stream = cand_cols && use_stream && (L.tm <= 2 || L.wide) && (int64_t)(64 * L.tm + BN2) * kb < ((int64_t)1 << 31);
    if (!L.stream && L.wide) { L.wide = 0; L.tm = pick_tm(M, scoring); }
    // Slab kernel: int8 or fp8 storage, one group, 2..6 K-steps (the 256-column slab is <= 96 KiB), whole 32-row units, at least three
    // units per wave and slab, and a streamed operand that stays in an XCD's L2.
    static const int use_slab = getenv("ADALOG_GEMM_SLAB") ? atoi(getenv("ADALOG_GEMM_SLAB")) : 1;
    static const int slab_min_m = getenv("ADALOG_GEMM_SLAB_MINM") ? atoi(getenv("ADALOG_GEMM_SLAB_MINM")) : 768;
    if (L.stream && (g_slab_override >= 0 ? g_slab_override : use_slab) && (dtype == 0 || dtype == 3) && G == 1 && kb <= 6 * BK3 && kvalid_bytes > BK3 && M % 32 == 0 && M >= slab_min_m &&
        (int64_t)M * kb <= ((int64_t)3 << 20) && (int64_t)cdiv(N, BN2) * (M / 32) < ((int64_t)1 << 30)) {
        L.slab = 1;
        L.slab_U = M / 32;
        L.NT = cdiv(N, BN2);
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
// not need to know): int8 or fp8, one K-step, 5..7 row blocks, many groups, plain column factors.
static bool grp_ok(int dtype, int M, int N, int G, int gmod, int ref_div, int64_t kvalid_bytes, const float* bias,
                   const float* row_scale, int64_t sb_n, int64_t ref_cs) {
    static const int use_grp = getenv("ADALOG_GEMM_GRP") ? atoi(getenv("ADALOG_GEMM_GRP")) : 1;
    return use_grp && (dtype == 0 || dtype == 3) && kvalid_bytes <= BK3 && M > 128 && M <= 224 && G >= 8 && gmod <= 8 && !bias && !row_scale &&
           sb_n == 0 && (ref_div == 64 || ref_div == 128 || ref_div == 256) && N % ref_div == 0 &&
           (int64_t)(N / ref_div) * ref_cs * 4 < ((int64_t)1 << 31);
}

// ... and of its several-K-steps form: bf16, exactly 7 K-steps (K = 193..224 elements: the 197 tokens of a 224 x 224 ViT).
static bool grpk_ok(int dtype, int M, int N, int G, int gmod, int ref_div, int64_t kvalid_bytes, const float* bias,
                    const float* row_scale, int64_t sb_n, int64_t ref_cs) {
    static const int use_grp = getenv("ADALOG_GEMM_GRP") ? atoi(getenv("ADALOG_GEMM_GRP")) : 1;
    return use_grp && dtype == 1 && kvalid_bytes > 6 * BK3 && kvalid_bytes <= 7 * BK3 && M > 128 && M <= 224 && G >= 8 && gmod <= 8 &&
           !bias && !row_scale && sb_n == 0 && (ref_div == 64 || ref_div == 128 || ref_div == 256) && N % ref_div == 0 &&
           (int64_t)(N / ref_div) * ref_cs * 4 < ((int64_t)1 << 31);
}

// M, N: GEMM rows / columns (N includes the candidate factor when ref_div > 1).  Outputs the partial-buffer layout
// [c_eff][G][MT][Npad] the kernel will write, for allocation and for adalog_finish_scores.
extern "C" int64_t adalog_gemm_score_layout(int M, int N, int C, int G, int gmod, int ref_div, int reduce_cols, int dtype,
                                            int64_t Kp, int64_t k_valid, int ref_transposed, int* MT, int* Npad, int* mode) {
    const int esz = (dtype == 0 || dtype == 3) ? 1 : dtype == 1 ? 2 : 4;
    const Layout L = layout_of(M, N, C, G, gmod, ref_div, reduce_cols, true, (k_valid > 0 ? k_valid : Kp) * esz, Kp * esz,
                               ref_transposed != 0, dtype);
    if (MT) *MT = L.acc ? L.wgs : L.MT;
    if (Npad) *Npad = L.acc ? BN2 : L.Npad;
    if (mode) *mode = L.acc ? 2 : (ref_div > 1 ? 1 : 0);
    return L.elems;
}

extern "C" int adalog_gemm_score(int dtype, const void* A, const void* B, int64_t sAc, int64_t sAg, int64_t sBc,
                                 int64_t sBg, int M, int N, int64_t Kp, int64_t k_valid, int C, int G, int gmod, const float* ref,
                                 int64_t ldr, int64_t sRg, int64_t ref_cs, int ref_div, const float* sa, int64_t sa_c,
                                 int64_t sa_g, float sa_mul, const float* sb, int64_t sb_c, int64_t sb_g, int64_t sb_n,
                                 const float* bias, int64_t bi_c, int64_t bi_g, int64_t bi_n, const float* row_scale,
                                 const float* row_bias, float* partial, int64_t partial_elems, float* out, int64_t ldo,
                                 int64_t sOc, int64_t sOg, int order, int reduce_cols, void* stream) {
    ADALOG_ARG_CHECK(A && B && sa && sb, "gemm_score: null operand/scale pointer");
    ADALOG_ARG_CHECK(dtype >= 0 && dtype <= 3, "gemm_score: dtype must be 0 (i8), 1 (bf16), 2 (f32) or 3 (fp8 e4m3)");
    ADALOG_ARG_CHECK(M >= 1 && N >= 1 && C >= 1 && G >= 1 && gmod >= 1 && G % gmod == 0 && ref_div >= 1, "gemm_score: bad sizes");
    const int esz = (dtype == 0 || dtype == 3) ? 1 : dtype == 1 ? 2 : 4;
    ADALOG_ARG_CHECK((Kp * esz) % BK3 == 0 && Kp > 0, "gemm_score: padded K must be a multiple of 64 bytes");
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
    if (partial) ADALOG_ARG_CHECK(partial_elems >= L.elems, "gemm_score: partial buffer too small");
    if (L.acc) { ADALOG_ARG_CHECK(((uintptr_t)partial & 7) == 0, "gemm_score: accumulator buffer must be 8-byte aligned"); p.wg_acc = (double*)partial; }
    const int64_t nwg = (int64_t)L.MT * L.NT * G * C;
    ADALOG_ARG_CHECK(nwg < (int64_t)1 << 31, "gemm_score: grid too large");
    hipStream_t st = (hipStream_t)stream;
    dim3 grid((unsigned)nwg);
    static const int use_glds = getenv("ADALOG_GEMM_GLDS") ? atoi(getenv("ADALOG_GEMM_GLDS")) : 1;   // LDS-DMA pipeline (default on)
    ADALOG_ARG_CHECK((Kp * esz) % BK2 == 0 || (L.stream && !out),
                     "gemm_score: rows padded to 64 (not 128) bytes are taken by the streaming search kernel only");
    ADALOG_ARG_CHECK(dtype != 3 || (L.stream && !out), "gemm_score: fp8 operands are taken by the streaming search kernel only (ref_div 64/128/256, transposed reference)");
    if (L.slab && !out) {
        // slab kernel: one workgroup per CU, each takes a contiguous range of (slab, unit) pairs
        p.MT = L.MT; p.NT = L.NT; p.slab_U = L.slab_U; p.slab_R = L.slab_R;
        const int nk = (int)((p.Kvb + BK3 - 1) / BK3);
        const size_t shm = (size_t)nk * BN2 * BK3 + 8 * 3 * 32 * BK3 + 8 * 192 * 4 + 8 * BN2 * 4;
        // a slab that is not cut has unused pieces: they must read as zero
        if (!L.acc) {
            const hipError_t me = hipMemsetAsync(partial, 0, (size_t)L.elems * sizeof(float), st);
            if (me != hipSuccess) { adalog_set_error("adalog_gemm_score (clear partials)", me); return (int)me; }
        }
        const int nref = BN2 / ref_div;
#define LAUNCH_SLAB(NREFV, ROWSV, DTV)                                                                            \
        do {                                                                                                      \
            static bool attr_set = false;                                                                         \
            if (!attr_set) {                                                                                      \
                (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gemm_slab<NREFV, ROWSV, DTV>),         \
                                          hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);                \
                attr_set = true;                                                                                  \
            }                                                                                                     \
            adalog_note_kernel(DTV == 3 ? "k_gemm_slab<fp8>" : "k_gemm_slab<i8>");                                 \
            hipLaunchKernelGGL((k_gemm_slab<NREFV, ROWSV, DTV>), dim3((unsigned)L.wgs), dim3(512), shm, st, p);   \
        } while (0)
#define LAUNCH_SLAB_DT(DTV)                                                                                       \
        do {                                                                                                      \
            if (row_scale) { if (nref == 1) LAUNCH_SLAB(1, true, DTV); else if (nref == 2) LAUNCH_SLAB(2, true, DTV); else LAUNCH_SLAB(4, true, DTV); } \
            else { if (nref == 1) LAUNCH_SLAB(1, false, DTV); else if (nref == 2) LAUNCH_SLAB(2, false, DTV); else LAUNCH_SLAB(4, false, DTV); } \
        } while (0)
        if (dtype == 3) LAUNCH_SLAB_DT(3); else LAUNCH_SLAB_DT(0);
#undef LAUNCH_SLAB_DT
#undef LAUNCH_SLAB
    } else if (L.stream && !out && L.acc && grp_ok(dtype, M, N, G, gmod, ref_div, p.Kvb, bias, row_scale, sb_n, ref_cs)) {
        // group kernel (q.k^T searches): same accumulator layout and workgroup count as the streaming kernel
        const int NB = N / 32;
        const int nch0 = cdiv((int64_t)3 * L.wgs, G);
        const int CB = cdiv(cdiv(NB, nch0 < 1 ? 1 : nch0), 8) * 8;
        p.slab_R = CB; p.slab_U = cdiv(NB, CB);
        const size_t shm = (size_t)3 * 8 * 32 * BK3 + 7 * 256 * 4 + (size_t)gmod * 256 * 8;
#define LAUNCH_GRP(NJV, DTV)                                                                                      \
        do {                                                                                                      \
            static bool attr_set = false;                                                                         \
            if (!attr_set) {                                                                                      \
                (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gemm_grp<NJV, DTV>),                   \
                                          hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);                 \
                attr_set = true;                                                                                  \
            }                                                                                                     \
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
            static bool attr_set = false;                                                                         \
            if (!attr_set) {                                                                                      \
                (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gemm_grpk<NJV, 7>),                    \
                                          hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);                \
                attr_set = true;                                                                                  \
            }                                                                                                     \
            adalog_note_kernel("k_gemm_grpk<bf16>");                                                              \
            hipLaunchKernelGGL((k_gemm_grpk<NJV, 7>), dim3((unsigned)L.wgs), dim3(512), shm, st, p);              \
        } while (0)
        if (ref_div == 64) LAUNCH_GRPK(2); else if (ref_div == 128) LAUNCH_GRPK(4); else LAUNCH_GRPK(8);
#undef LAUNCH_GRPK
    } else if (L.stream && !out) {
        // persistent streaming kernel: two (wide form: one) workgroups per CU walk the tile list
        {   // m-tiles per L2 group: A rows of one group <= 2 MiB (half of an XCD's L2)
            const int64_t a_tile = (int64_t)64 * L.tm * p.Kb;
            int64_t gm = ((int64_t)2 << 20) / a_tile;
            if (const char* e = getenv("ADALOG_GEMM_GM")) gm = atoi(e);
            p.gm = (int)(gm < 1 ? 1 : gm > L.MT ? L.MT : gm);
        }
        dim3 pgrid((unsigned)L.wgs);
        const size_t shm = (size_t)(L.wide ? 4 : 3) * (64 * L.tm + BN2) * BK3;
#define LAUNCH_STREAM(DT, RIV, NWV, NSV)                                                                          \
        do {                                                                                                      \
            static bool attr_set = false;                                                                         \
            if (!attr_set) {                                                                                      \
                (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gemm_stream<DT, RIV, NWV, NSV>),       \
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (NSV == 4 ? 128 : 80) * 1024); \
                attr_set = true;                                                                                  \
            }                                                                                                     \
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
            static bool attr_set = false;                                                                         \
            if (!attr_set) {                                                                                      \
                (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gemm_cand_glds<DT, TMV>),              \
                                          hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);                \
                attr_set = true;                                                                                  \
            }                                                                                                     \
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
            static bool attr_set = false;                                                                         \
            if (!attr_set) {                                                                                      \
                (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gemm_cand<DT, TMV, ST>),                   \
                                    hipFuncAttributeMaxDynamicSharedMemorySize, 72 * 1024);                       \
                attr_set = true;                                                                                  \
            }                                                                                                     \
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
        if (out) LAUNCH_BIG_DT(true); else LAUNCH_BIG_DT(false);
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
