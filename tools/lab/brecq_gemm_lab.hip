// EXPERIMENT, not built into libadalog_hip.so (tools/lab/bgemm_bench.py builds and times it; outcome in profiles/r03_notes.md).
// K18 candidate -- fp32 matrix products of the BRECQ block reconstruction on the fp32 MFMA pipe (v_mfma_f32_32x32x2_f32).
//   forward of a quantised Linear      y  = Xq Wq^T + b         (reference quant_layers/linear.py:95-101 in training mode)
//   its two backward products          dX = dY Wq,  dW = dY^T Xq (what autograd derives for block_recon.py:137 `err.backward()`)
//   the attention products q k^T / softmax v and their backward products, batched over (image, head) (matmul.py:58-68)
// One kernel computes C[b] = opA(A[b]) opB(B[b])^T (+ bias) for operands stored either K-contiguous ([rows][K]) or K-major
// ([K][rows]), so none of the three products of a Linear needs a transposed copy of X, dY or W.
//
// The fp32 MFMA does 32x32x2 per 64 cycles and takes ONE float per lane per operand: at that rate LDS and the L2 are far
// from their limits (0.1 of the LDS bandwidth at a 128 x 128 tile), so the kernel is plain -- register-staged double
// buffer, one barrier per 16-wide K tile, 2 workgroups per CU -- and what matters is filling the 256 CUs: a Linear of a
// deit_small block has 150..600 output tiles forward but 9..36 in dW (K = tokens = 6 304), hence split-K with a fixed-order
// second pass (no atomics: the result does not depend on the launch's timing).
#include "common.h"

#include <stdlib.h>

namespace {

typedef float v16f __attribute__((ext_vector_type(16)));

constexpr int BKG = 16;          // K tile (floats)
constexpr int BNG = 128;         // columns of C per workgroup

struct BGemmArgs {
    const float* A; const float* B; const float* bias; float* C;
    int64_t lda, ldb, ldc, sA, sB, sC;     // leading dimensions; strides between batch entries
    int M, N, K;
    int MT, NT, splits, kt_per_split;      // tiles; K tiles per split
    int64_t sSplit;                        // floats between the partial results of two splits (0: written to C directly)
};

// Operand tile [BKG][ROWS] in LDS, K-major whatever the operand's layout in memory:
//   K-contiguous source: thread -> (row, 4 consecutive k): one global float4, four transposing ds_write_b32;
//                        row stride ROWS + 2 (= 2 mod 32 banks): a half-wave's 8 rows x 4 chunks hit 32 different banks
//   K-major source:      thread -> (k, 4 consecutive rows): one global float4, one ds_write_b128; row stride ROWS + 4
template <bool KMAJOR, int ROWS> struct Tile {
    static constexpr int STRIDE = KMAJOR ? ROWS + 4 : ROWS + 2;
    static constexpr int CH = BKG / 4;                            // float4 chunks of a K-contiguous row per K tile
    static constexpr int NLD = ROWS * BKG / 4 / 256;              // float4 loads per thread per K tile
    static constexpr int FLOATS = BKG * STRIDE;

    // rows: valid rows of the operand from this tile's first row on;  kleft: valid k from this K tile's first k on.
    // Branch-free: out-of-range requests read the nearest valid float4 instead (a guarded load becomes a branch with its own
    // vmcnt(0), which serialises the loads with the MFMAs) and `ok` tells store() to write zeros in their place.
    __device__ static __forceinline__ unsigned load(float4 (&v)[NLD], const float* __restrict__ src, int64_t ld, int rows, int kleft, int tid) {
        unsigned ok = 0;
        if constexpr (KMAJOR) {
            const int m4 = (tid % (ROWS / 4)) * 4, kr = tid / (ROWS / 4);
            constexpr int KSTEP = 256 / (ROWS / 4);
            const int mc = m4 < rows ? m4 : rows - 4;
#pragma unroll
            for (int q = 0; q < NLD; ++q) {
                const int k = q * KSTEP + kr;
                const int kc = k < kleft ? k : kleft - 1;
                v[q] = *reinterpret_cast<const float4*>(src + (int64_t)kc * ld + mc);
                ok |= (k < kleft && m4 < rows) ? 1u << q : 0u;
            }
        } else {
            const int c = tid % CH, r = tid / CH;
            const int cc = 4 * c < kleft ? 4 * c : kleft - 4;
#pragma unroll
            for (int q = 0; q < NLD; ++q) {
                const int row = q * (256 / CH) + r;
                const int rc = row < rows ? row : rows - 1;
                v[q] = *reinterpret_cast<const float4*>(src + (int64_t)rc * ld + cc);
                ok |= (row < rows && 4 * c < kleft) ? 1u << q : 0u;
            }
        }
        return ok;
    }
    __device__ static __forceinline__ void store(const float4 (&v)[NLD], unsigned ok, float* __restrict__ s, int tid) {
        if constexpr (KMAJOR) {
            const int m4 = (tid % (ROWS / 4)) * 4, kr = tid / (ROWS / 4);
            constexpr int KSTEP = 256 / (ROWS / 4);
#pragma unroll
            for (int q = 0; q < NLD; ++q)
                *reinterpret_cast<float4*>(s + (q * KSTEP + kr) * STRIDE + m4) = (ok >> q & 1u) ? v[q] : make_float4(0.f, 0.f, 0.f, 0.f);
        } else {
            const int c = tid % CH, r = tid / CH;
#pragma unroll
            for (int q = 0; q < NLD; ++q) {
                float* d = s + (4 * c) * STRIDE + q * (256 / CH) + r;
                const float4 x = (ok >> q & 1u) ? v[q] : make_float4(0.f, 0.f, 0.f, 0.f);
                d[0] = x.x; d[STRIDE] = x.y; d[2 * STRIDE] = x.z; d[3 * STRIDE] = x.w;
            }
        }
    }
};

// C tile (64*TM) x 128, 4 waves as 2 x 2, each TM x 2 MFMA blocks of 32 x 32.
template <bool AK, bool BK_, int TM>
__global__ __launch_bounds__(256, 2) void k_bgemm(BGemmArgs p) {
    constexpr int BMG = 64 * TM;
    typedef Tile<AK, BMG> TA;
    typedef Tile<BK_, BNG> TB;
    __shared__ __attribute__((aligned(16))) float sA[2][TA::FLOATS];
    __shared__ __attribute__((aligned(16))) float sB[2][TB::FLOATS];

    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int wr = w >> 1, wc = w & 1, fi = lane & 31, fk = lane >> 5;

    // tile of this workgroup: every XCD (block id mod 8) takes a contiguous range of (m tile, n tile) pairs, n fastest, so
    // the workgroups that share rows of A run on one XCD and find them in its L2
    const unsigned per_z = (unsigned)p.MT * p.NT, bid = blockIdx.x, xcd = bid & 7, nb = gridDim.x;
    const unsigned q8 = nb >> 3, r8 = nb & 7;
    const unsigned lin = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
    const int nt = lin % p.NT, mt = (lin / p.NT) % p.MT;
    const int z = lin / per_z, split = z % p.splits, b = z / p.splits;
    const int m0 = mt * BMG, n0 = nt * BNG;
    const int kt0 = split * p.kt_per_split;
    int nkt = (p.K + BKG - 1) / BKG - kt0;
    if (nkt > p.kt_per_split) nkt = p.kt_per_split;

    const float* Ab = p.A + (int64_t)b * p.sA + (AK ? (int64_t)m0 : (int64_t)m0 * p.lda);
    const float* Bb = p.B + (int64_t)b * p.sB + (BK_ ? (int64_t)n0 : (int64_t)n0 * p.ldb);
    const int64_t akstep = AK ? p.lda : 1, bkstep = BK_ ? p.ldb : 1;
    const int arows = p.M - m0, brows = p.N - n0;

    v16f acc[TM][2];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    float4 ra[TA::NLD], rb[TB::NLD];
    unsigned oka = 0, okb = 0;
    if (nkt > 0) {
        const int k0 = kt0 * BKG;
        oka = TA::load(ra, Ab + (int64_t)k0 * akstep, p.lda, arows, p.K - k0, tid);
        okb = TB::load(rb, Bb + (int64_t)k0 * bkstep, p.ldb, brows, p.K - k0, tid);
        TA::store(ra, oka, sA[0], tid);
        TB::store(rb, okb, sB[0], tid);
    }
    __syncthreads();
    for (int t = 0; t < nkt; ++t) {
#if defined(BG_LAB_NO_LOADS)     // tools/lab only: the main loop without its global loads / LDS stores
        const bool more = false;
#else
        const bool more = t + 1 < nkt;
#endif
        if (more) {
            const int k0 = (kt0 + t + 1) * BKG;
            oka = TA::load(ra, Ab + (int64_t)k0 * akstep, p.lda, arows, p.K - k0, tid);
            okb = TB::load(rb, Bb + (int64_t)k0 * bkstep, p.ldb, brows, p.K - k0, tid);
        }
        const float* a = sA[t & 1] + wr * (32 * TM) + fi + fk * TA::STRIDE;
        const float* bq = sB[t & 1] + wc * 64 + fi + fk * TB::STRIDE;
        // all fragments of the K tile first (16 ds_read2), then the MFMAs behind counted lgkmcnt waits: read -> wait -> 4 MFMAs
        // per K pair left every read's latency exposed
        float av[BKG / 2][TM], bv[BKG / 2][2];
#pragma unroll
        for (int kk = 0; kk < BKG / 2; ++kk) {
#pragma unroll
            for (int i = 0; i < TM; ++i) av[kk][i] = a[kk * 2 * TA::STRIDE + i * 32];
#pragma unroll
            for (int j = 0; j < 2; ++j) bv[kk][j] = bq[kk * 2 * TB::STRIDE + j * 32];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int kk = 0; kk < BKG / 2; ++kk) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
#if defined(BG_LAB_NO_MFMA)      // tools/lab only: loads, LDS traffic and barriers without the matrix work
                for (int j = 0; j < 2; ++j) acc[i][j][kk] += av[kk][i] + bv[kk][j];
#else
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[kk][i], bv[kk][j], acc[i][j], 0, 0, 0);
#endif
        }
        __builtin_amdgcn_sched_barrier(0);
        if (more) {
            TA::store(ra, oka, sA[(t + 1) & 1], tid);
            TB::store(rb, okb, sB[(t + 1) & 1], tid);
        }
        __syncthreads();
    }

    // accumulator register r of a 32 x 32 block: row 8 * (r / 4) + 4 * (lane / 32) + r % 4, column lane % 32
    float* Cb = p.C + (int64_t)b * p.sC + (int64_t)split * p.sSplit;
    const bool add_bias = p.bias != nullptr && p.splits == 1;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int col = n0 + wc * 64 + j * 32 + fi;
        if (col >= p.N) continue;
        const float bs = add_bias ? p.bias[col] : 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int rbase = m0 + wr * (32 * TM) + i * 32 + 4 * fk;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = rbase + 8 * (r >> 2) + (r & 3);
                if (row < p.M) Cb[(int64_t)row * p.ldc + col] = acc[i][j][r] + bs;
            }
        }
    }
}

// C[b][m][n] = bias[n] + sum over splits (fixed order) of the partial products
__global__ __launch_bounds__(256) void k_bgemm_reduce(const float* __restrict__ part, int64_t sSplit, int splits, const float* __restrict__ bias,
                                                     float* __restrict__ C, int64_t ldc, int64_t sC, int M, int N, int batch) {
    const int n4 = N >> 2;
    const int64_t total = (int64_t)batch * M * n4;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int c = (int)(i % n4) * 4;
        const int64_t rm = i / n4;
        const int m = (int)(rm % M), b = (int)(rm / M);
        const float* src = part + ((int64_t)b * M + m) * N + c;
        float4 s = *reinterpret_cast<const float4*>(src);
        for (int k = 1; k < splits; ++k) {
            const float4 v = *reinterpret_cast<const float4*>(src + (int64_t)k * sSplit);
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
        if (bias) { s.x += bias[c]; s.y += bias[c + 1]; s.z += bias[c + 2]; s.w += bias[c + 3]; }
        *reinterpret_cast<float4*>(C + (int64_t)b * sC + (int64_t)m * ldc + c) = s;
    }
}

int bgemm_env(const char* name, int dflt) {
    const char* v = getenv(name);
    return v ? atoi(v) : dflt;
}

// rows per workgroup (64 or 128) and the split of K: enough workgroups for two per CU on all 256 CUs, K pieces of at least
// four tiles, partial results of at most a few tens of MB
void bgemm_plan(int M, int N, int K, int batch, int& tm, int& splits, int& ktps) {
    const int nt = cdiv(N, BNG), nkt = cdiv(K, BKG);
    tm = 2;
    if ((int64_t)cdiv(M, 128) * nt * batch < 384) tm = 1;
    static const int force_tm = bgemm_env("ADALOG_BGEMM_TM", 0);
    if (force_tm == 1 || force_tm == 2) tm = force_tm;
    const int64_t tiles = (int64_t)cdiv(M, 64 * tm) * nt * batch;
    int s = 1;
    if (tiles < 256) {
        s = (int)((512 + tiles - 1) / tiles);
        if (s > nkt / 4) s = nkt / 4;
        if (s < 1) s = 1;
    }
    static const int force_s = bgemm_env("ADALOG_BGEMM_SPLITS", 0);
    if (force_s >= 1) s = force_s < nkt ? force_s : nkt;
    ktps = cdiv(nkt, s);
    splits = cdiv(nkt, ktps);
}

}  // namespace

// 1 when adalog_brecq_gemm takes these operands (16-byte aligned rows: every leading dimension and, for K-major operands,
// the row counts are multiples of 4 floats; K a multiple of 4 for K-contiguous operands)
extern "C" int adalog_brecq_gemm_ok(int M, int N, int K, int64_t lda, int a_kmajor, int64_t ldb, int b_kmajor, int64_t ldc) {
    if (M < 1 || N < 1 || K < 1 || (lda & 3) || (ldb & 3) || (ldc & 3) || (N & 3)) return 0;
    if (a_kmajor ? (M & 3) : (K & 3)) return 0;
    if (b_kmajor ? (N & 3) : (K & 3)) return 0;
    return 1;
}

extern "C" int64_t adalog_brecq_gemm_workspace_bytes(int M, int N, int K, int batch) {
    int tm, splits, ktps;
    bgemm_plan(M, N, K, batch, tm, splits, ktps);
    return splits > 1 ? (int64_t)splits * batch * M * N * 4 : 0;
}

// C[b] (M x N, row stride ldc) = opA(A[b]) opB(B[b])^T (+ bias[n]),  b < batch, fp32 throughout (fp32 MFMA accumulate).
//   a_kmajor = 0: A[b] is [M][K] with row stride lda;   1: A[b] is [K][M] with row stride lda.   Same for B with N.
//   Linear forward   y[T, O]  = x[T, I] w[O, I]^T + bias : A = x (0), B = w (0)
//   Linear backward  dx[T, I] = dy[T, O] w[O, I]         : A = dy (0), B = w (1, K = O)
//                    dw[O, I] = dy[T, O]^T x[T, I]       : A = dy (1), B = x (1), K = T
extern "C" int adalog_brecq_gemm(const float* A, int64_t lda, int a_kmajor, const float* B, int64_t ldb, int b_kmajor,
                                 const float* bias, float* C, int64_t ldc, int M, int N, int K, int batch, int64_t sA, int64_t sB,
                                 int64_t sC, void* workspace, int64_t workspace_bytes, void* stream) {
    ADALOG_ARG_CHECK(A && B && C && batch >= 1, "brecq_gemm: null pointer / empty batch");
    ADALOG_ARG_CHECK(adalog_brecq_gemm_ok(M, N, K, lda, a_kmajor, ldb, b_kmajor, ldc), "brecq_gemm: operands not 16-byte aligned by rows (ask adalog_brecq_gemm_ok first)");
    ADALOG_ARG_CHECK(((((uintptr_t)A | (uintptr_t)B | (uintptr_t)C) & 15) == 0) && ((sA | sB | sC) & 3) == 0, "brecq_gemm: pointers / batch strides must be 16-byte aligned");
    int tm, splits, ktps;
    bgemm_plan(M, N, K, batch, tm, splits, ktps);
    ADALOG_ARG_CHECK(splits == 1 || (workspace && workspace_bytes >= (int64_t)splits * batch * M * N * 4 && ((uintptr_t)workspace & 15) == 0),
                     "brecq_gemm: workspace too small or misaligned (adalog_brecq_gemm_workspace_bytes)");
    BGemmArgs p{};
    p.A = A; p.B = B; p.bias = bias; p.lda = lda; p.ldb = ldb; p.sA = sA; p.sB = sB;
    p.M = M; p.N = N; p.K = K; p.MT = cdiv(M, 64 * tm); p.NT = cdiv(N, BNG); p.splits = splits; p.kt_per_split = ktps;
    if (splits == 1) { p.C = C; p.ldc = ldc; p.sC = sC; p.sSplit = 0; }
    else { p.C = (float*)workspace; p.ldc = N; p.sC = (int64_t)M * N; p.sSplit = (int64_t)batch * M * N; }
    const int64_t nwg = (int64_t)p.MT * p.NT * splits * batch;
    ADALOG_ARG_CHECK(nwg < ((int64_t)1 << 31), "brecq_gemm: too many tiles");
    hipStream_t st = (hipStream_t)stream;
#define BG_LAUNCH(AKV, BKV, TMV) hipLaunchKernelGGL((k_bgemm<AKV, BKV, TMV>), dim3((unsigned)nwg), dim3(256), 0, st, p)
#define BG_TM(AKV, BKV) do { if (tm == 2) BG_LAUNCH(AKV, BKV, 2); else BG_LAUNCH(AKV, BKV, 1); } while (0)
    adalog_note_kernel("k_bgemm<f32>");
    if (a_kmajor) { if (b_kmajor) BG_TM(true, true); else BG_TM(true, false); }
    else { if (b_kmajor) BG_TM(false, true); else BG_TM(false, false); }
#undef BG_TM
#undef BG_LAUNCH
    ADALOG_LAUNCH_CHECK("adalog_brecq_gemm");
    if (splits > 1) {
        const int64_t total = (int64_t)batch * M * (N >> 2);
        const int blocks = (int)(total / 256 + 1 < 2048 ? total / 256 + 1 : 2048);
        hipLaunchKernelGGL(k_bgemm_reduce, dim3((unsigned)blocks), dim3(256), 0, st, (const float*)workspace, p.sSplit, splits, bias,
                           C, ldc, sC, M, N, batch);
        ADALOG_LAUNCH_CHECK("adalog_brecq_gemm (reduce)");
    }
    return 0;
}
