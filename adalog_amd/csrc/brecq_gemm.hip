// K17b -- the training-mode contractions of BRECQ / AdaRound block reconstruction on the bf16 matrix cores, at fp32 accuracy.
//
// Replaces the fp32 library GEMMs behind utils/block_recon.py:116-121 (`out_quant = block(cur_inp)`, `err.backward()`):
//   quant_layers/linear.py:46-50   F.linear(x_sim, w_sim, bias)   -> forward, dL/dx_sim = dL/dy . w_sim, dL/dw_sim = dL/dy^T . x_sim
//   quant_layers/matmul.py:41-44   A_sim @ B_sim                  -> forward and both backward products, batched over (image, head)
// All operands are general fp32 tensors (fake-quantised activations times a trained scale, soft-rounded weights, upstream
// gradients), and the reference computes the products in fp32.  gfx950 has no reduced-precision fp32 matrix path (no xf32); its
// fp32 MFMA runs at 1/16 of the bf16 rate.  Here every fp32 operand element is split IN REGISTERS into three bf16 terms
//   x = hi + mid + lo   (hi = bf16(x), mid = bf16(x - hi), lo = bf16(x - hi - mid): exact, 3 x 8 = 24 significand bits)
// and a product is accumulated as the six bf16 MFMA products whose weight is >= 2^-24 of the full one
//   hi.hi + hi.mid + mid.hi + hi.lo + mid.mid + lo.hi        (each bf16 x bf16 product is exact in the fp32 accumulator)
// i.e. the same operand values and the same accumulation precision as an fp32 FMA chain, at 6/16 of its matrix-pipe time.
// Operands are read from HBM as fp32 exactly once per tile pass (no packed copies, no transposed copies): a tile of either
// operand may be K-contiguous ("N": rows of 16 k-values, LDS rows of 64 bytes with the 16-byte-slot XOR swizzle) or K-major
// ("T": the backward products read dL/dy and x_sim with the token index as K; LDS holds [k][row] and a fragment is eight
// conflict-free ds_read_b32).  Staging is LDS-DMA (buffer_load ... lds) in a 3-stage ring with counted vmcnt, persistent
// workgroups walking an XCD-contiguous tile list, two workgroups per CU.  Long-K / few-tile products (dL/dw: K = tokens) are
// split along K into fixed ranges whose partial tiles a second kernel adds in a fixed order (bit-reproducible).
#include "common.h"
#include <stdlib.h>

#pragma clang fp contract(off)      // the residuals x - hi, (x - hi) - mid must stay exact subtractions

namespace {

typedef float v16f __attribute__((ext_vector_type(16)));
typedef __bf16 v8bf __attribute__((ext_vector_type(8)));
typedef __bf16 v2bf __attribute__((ext_vector_type(2)));
typedef void __attribute__((address_space(3)))* las_ptr;

struct BqArgs {
    const float* A; const float* B; float* C;
    int64_t lda, ldb, ldc, sAg, sBg, sCg;      // element strides
    int Gi; int64_t sAo, sBo, sCo;             // two-level groups: group g sits at (g % Gi) * s?g + (g / Gi) * s?o  (Gi = G: one level)
    int M, N, K, G, S;                         // S = K splits (S > 1: C is the partial buffer [S][G][M][N], ldc = N)
    int MT, NT, nk;                            // tiles, 16-element K-steps
    const float* bias;                         // [N] or null (S == 1 only)
    float alpha; const float* alpha_dev;       // C = alpha * alpha_dev[0] * (A.B^T) + bias
    int kb[65];                                // K-step range of split s: [kb[s], kb[s + 1])  (host-computed: keeps them scalar)
    int64_t bplane;                            // PB == 0 (B = three bf16 planes, ldb = row stride in bf16 elements): plane stride
};

#if defined(BQ_LAB_NS)                           // tools/lab only
constexpr int BQ_NS = BQ_LAB_NS;
#else
constexpr int BQ_NS = 4;                        // ring stages: compute s | fragments of s + 1 | two steps of requests in flight
#endif
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
constexpr int BQ_KS = 16;                       // fp32 elements per K-step (64 bytes of an N row)

__device__ __forceinline__ int swz64(int row, int slot) { return row * 64 + ((slot ^ ((row >> 2) & 3)) << 4); }

__device__ __forceinline__ uint4 lds_u4(const uint8_t* __restrict__ stage, int off) { return *reinterpret_cast<const uint4*>(stage + off); }
__device__ __forceinline__ float lds_f1(const uint8_t* __restrict__ stage, int off) { return *reinterpret_cast<const float*>(stage + off); }

// x[0..7] -> three bf16x8 fragments with hi + mid + lo == x exactly (round-to-nearest-even conversions, exact residuals)
struct Frag3 { uint4 hi, mid, lo; };
// One fp32 subtraction as ONE plain VALU instruction: the SLP vectoriser would pair two of them into v_pk_add_f32, and packed
// fp32 math does not issue beside MFMAs (profiles/r02_probe_wave_pair_overlap.txt) -- the split must run in their shadow.
__device__ __forceinline__ float sub1(float a, float b) {
    float d;
    asm("v_sub_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
__device__ __forceinline__ void split_pair(float a, float b, uint32_t& H, uint32_t& Mi, uint32_t& L) {
    v2bf h = {(__bf16)a, (__bf16)b};
    H = *reinterpret_cast<uint32_t*>(&h);
    const float ra = sub1(a, __uint_as_float(H << 16)), rb = sub1(b, __uint_as_float(H & 0xffff0000u));
    v2bf m = {(__bf16)ra, (__bf16)rb};
    Mi = *reinterpret_cast<uint32_t*>(&m);
    const float sa = sub1(ra, __uint_as_float(Mi << 16)), sb = sub1(rb, __uint_as_float(Mi & 0xffff0000u));
    v2bf l = {(__bf16)sa, (__bf16)sb};
    L = *reinterpret_cast<uint32_t*>(&l);
}
// two-term split (hi + mid: relative error 2^-16): the gradient contractions of a BRECQ iteration (round 6)
__device__ __forceinline__ void split_pair2(float a, float b, uint32_t& H, uint32_t& Mi) {
    v2bf h = {(__bf16)a, (__bf16)b};
    H = *reinterpret_cast<uint32_t*>(&h);
    const float ra = sub1(a, __uint_as_float(H << 16)), rb = sub1(b, __uint_as_float(H & 0xffff0000u));
    v2bf m = {(__bf16)ra, (__bf16)rb};
    Mi = *reinterpret_cast<uint32_t*>(&m);
}
// values that are exact in bf16 (small integers): one conversion, mid = lo = 0 (never read)
__device__ __forceinline__ Frag3 cvt8(const float (&x)[8]) {
    Frag3 f;
    v2bf h0 = {(__bf16)x[0], (__bf16)x[1]}, h1 = {(__bf16)x[2], (__bf16)x[3]}, h2 = {(__bf16)x[4], (__bf16)x[5]}, h3 = {(__bf16)x[6], (__bf16)x[7]};
    f.hi = make_uint4(*reinterpret_cast<uint32_t*>(&h0), *reinterpret_cast<uint32_t*>(&h1), *reinterpret_cast<uint32_t*>(&h2), *reinterpret_cast<uint32_t*>(&h3));
    f.mid = make_uint4(0, 0, 0, 0); f.lo = f.mid;
    return f;
}
__device__ __forceinline__ Frag3 split8(const float (&x)[8]) {
    Frag3 f;
#if defined(BQ_LAB_NOSPLIT)   // tools/lab only: one conversion instead of the three-term split (wrong numerics, for timing)
    v2bf h0 = {(__bf16)x[0], (__bf16)x[1]}, h1 = {(__bf16)x[2], (__bf16)x[3]}, h2 = {(__bf16)x[4], (__bf16)x[5]}, h3 = {(__bf16)x[6], (__bf16)x[7]};
    f.hi = make_uint4(*reinterpret_cast<uint32_t*>(&h0), *reinterpret_cast<uint32_t*>(&h1), *reinterpret_cast<uint32_t*>(&h2), *reinterpret_cast<uint32_t*>(&h3));
    f.mid = f.hi; f.lo = f.hi;
    return f;
#endif
    split_pair(x[0], x[1], f.hi.x, f.mid.x, f.lo.x);
    split_pair(x[2], x[3], f.hi.y, f.mid.y, f.lo.y);
    split_pair(x[4], x[5], f.hi.z, f.mid.z, f.lo.z);
    split_pair(x[6], x[7], f.hi.w, f.mid.w, f.lo.w);
    return f;
}
// 16 consecutive floats at a wave-uniform address through the scalar cache (read-only data; waits for them itself)
__device__ __forceinline__ v16f sload16(const float* base) {
    const uint64_t a = (uint64_t)base;
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)a), hi = __builtin_amdgcn_readfirstlane((uint32_t)(a >> 32));
    const uint64_t u = ((uint64_t)hi << 32) | lo;
    v16f r;
    asm volatile("s_load_dwordx16 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(r) : "s"(u) : "memory");
    return r;
}
__device__ __forceinline__ v16f mm(const uint4& a, const uint4& b, v16f c) {
#if defined(BQ_LAB_NOMFMA)
    c[0] += __uint_as_float(a.x ^ b.x);
    return c;
#endif
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const v8bf*>(&a), *reinterpret_cast<const v8bf*>(&b), c, 0, 0, 0);
}

// C tile (64 RI) x (32 CJ WN), 2 WN waves as 2 x WN, wave tile (32 RI) x (32 CJ).  TA / TB: the operand is K-major in memory.
// PA / PB = 1: the operand's values are exactly representable in bf16 (small integers: the integer part of a uniformly
// fake-quantised activation, its scale handed over as alpha_dev) -- one conversion instead of the split, 3 products instead of
// 6.  PA / PB = 2: a two-term split (hi + mid, 2^-16 relative) -- 3 products for two general operands (hi.hi, hi.mid, mid.hi),
// 2 against an exact one: the gradient contractions (adalog_gemm_f32x3: exact = 2).
// KT: K % 16 != 0 with a K-contiguous operand (the tail of the last step is zeroed in registers).
// The MFMA's A operand is the N side: a lane then owns 4 consecutive n of one m in each accumulator quad -> 16-byte stores.
template <int RI, int CJ, int WN, bool TA, bool TB, int PA, int PB, bool KT>
__global__ __launch_bounds__(128 * WN, (RI * CJ >= 8 ? 1 : 2)) void k_bq_gemm(BqArgs p) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int NW = 2 * WN;
    constexpr int BM = 64 * RI, BN = 32 * CJ * WN;
    // PB == 0: B arrives already split -- three bf16 planes, K-contiguous rows (the soft-rounded weights: small, re-read by every
    // row tile, split once per iteration by the packer instead of once per row tile here).  A stage holds 32 bytes per row and
    // plane; a 1 KiB request covers 32 rows of one plane, the 16-byte half XOR-ed with bit 3 of the row (conflict-free b128 reads)
    constexpr bool BP = PB == 0;
    static_assert(!BP || !TB, "pre-split B is K-contiguous");
    constexpr int BREQ = BP ? 3 * BN / 32 : BN / 16;         // 1 KiB requests of the B part
    static_assert(BM % (16 * NW) == 0 && BREQ % NW == 0, "requests must divide evenly over the waves");
    constexpr int QA = BM / (16 * NW), QB = BREQ / NW;        // DMA requests (1 KiB) per wave per stage, A part / B part
    constexpr int MAXQ = QA + QB;
    constexpr int STAGE = BM * 64 + (BP ? BN * 96 : BN * 64);
    extern __shared__ __attribute__((aligned(16))) uint8_t ring[];     // BQ_NS * STAGE bytes (the only LDS object)

    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = w / WN, wc = w % WN;
    const int frow = lane & 31, fkg = lane >> 5;

    // ---- tile list: XCD x owns a contiguous range (neighbours share operand rows in its L2), its workgroups take them round-robin
    const unsigned T = (unsigned)p.MT * p.NT * p.G * p.S;
    const unsigned nwg = gridDim.x, bid = blockIdx.x, xcd = bid & 7;
    const unsigned q8 = T >> 3, r8 = T & 7;
    const unsigned t_lo = xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8;
    const unsigned t_cnt = q8 + (xcd < r8 ? 1u : 0u);
    const unsigned nj = (nwg >> 3) + (xcd < (nwg & 7) ? 1u : 0u);
    const unsigned j0 = bid >> 3;
    struct Tile { int mt, nt, s, g; };
    auto decode = [&](unsigned local) {
        unsigned t = t_lo + local;
        Tile r;
        r.nt = t % p.NT; t /= p.NT;
        r.mt = t % p.MT; t /= p.MT;
        r.s = t % p.S; r.g = t / p.S;
        // (the division expansions run on the VALU: tell the compiler the results are wave-uniform, or every DMA request
        // whose scalar offset derives from them is wrapped in a waterfall loop)
        r.nt = __builtin_amdgcn_readfirstlane(r.nt); r.mt = __builtin_amdgcn_readfirstlane(r.mt);
        r.s = __builtin_amdgcn_readfirstlane(r.s); r.g = __builtin_amdgcn_readfirstlane(r.g);
        return r;
    };
    auto k_range = [&](int s, int& lo, int& hi) {   // K-steps of split s: equal contiguous ranges
        lo = p.kb[s];
        hi = p.kb[s + 1];
    };

    // ---- issue cursor, BQ_NS - 1 steps ahead of the compute cursor, across tile boundaries
    const int lrow = lane >> 2, lslot16 = ((lane & 3) ^ ((lane >> 4) & 3)) << 4;
    __amdgpu_buffer_rsrc_t ra, rb;
    int voa[QA], vob[QB];
    int sa_step, sb_step;                           // bytes the scalar offset advances per K-step
    unsigned i_local = j0;
    int i_k = 0, i_hi = 0;
    auto issue_tile = [&](unsigned local) {
        const Tile t = decode(local);
        const int m0 = t.mt * BM, n0 = t.nt * BN;
        int lo;
        k_range(t.s, lo, i_hi);
        i_k = lo;
        const int go = __builtin_amdgcn_readfirstlane((unsigned)t.g / (unsigned)p.Gi), gi = t.g - go * p.Gi;
        const float* Ab = p.A + (int64_t)gi * p.sAg + (int64_t)go * p.sAo;
        [[maybe_unused]] const float* Bb = p.B + (int64_t)gi * p.sBg + (int64_t)go * p.sBo;
        if constexpr (TA) {
            // rows k >= K are out of range of the resource: they read as zero
            const int64_t bytes = ((int64_t)(p.K - 1) * p.lda + p.M - m0) * 4;     // ends with the last valid element
            ra = __builtin_amdgcn_make_buffer_rsrc((void*)(Ab + m0), 0, (int)(bytes > 0x7ffffffe ? 0x7ffffffe : bytes), 0x00020000);
#pragma unroll
            for (int q = 0; q < QA; ++q) {
                const int e = (w + NW * q) * 256 + lane * 4;            // float index inside the [16][BM] stage part
                voa[q] = ((e / BM) * (int)p.lda + (e % BM)) * 4;
            }
            sa_step = BQ_KS * (int)p.lda * 4;
        } else {
            // the resource ends with the last valid element of this group's matrix: with K % 16 != 0 the last K-step of the LAST row
            // would otherwise fetch up to 60 bytes past the tensor (masked by ktail, but still an out-of-bounds address); dwords
            // past the end read as zero, like the T form's
            const int64_t abytes = ((int64_t)(p.M - 1 - m0) * p.lda + p.K) * 4;
            ra = __builtin_amdgcn_make_buffer_rsrc((void*)(Ab + (int64_t)m0 * p.lda), 0, (int)(abytes > 0x7ffffffe ? 0x7ffffffe : abytes), 0x00020000);
#pragma unroll
            for (int q = 0; q < QA; ++q) {
                const int row = min((w + NW * q) * 16 + lrow, p.M - 1 - m0);  // edge rows: re-read the last valid row (never stored)
                voa[q] = row * (int)p.lda * 4 + lslot16;
            }
            sa_step = BQ_KS * 4;
        }
        if constexpr (BP) {
            const uint16_t* Bp = reinterpret_cast<const uint16_t*>(p.B) + (int64_t)gi * p.sBg + (int64_t)go * p.sBo;
            rb = __builtin_amdgcn_make_buffer_rsrc((void*)(Bp + (int64_t)n0 * p.ldb), 0, 0x7ffffffe, 0x00020000);
#pragma unroll
            for (int q = 0; q < QB; ++q) {
                const int rq = w + NW * q, plane = rq / (BN / 32), chunk = rq % (BN / 32);
                const int row = min(chunk * 32 + (lane >> 1), p.N - 1 - n0);
                vob[q] = (int)((row * p.ldb + plane * p.bplane) * 2) + (((lane & 1) ^ ((lane >> 4) & 1)) << 4);
            }
            sb_step = BQ_KS * 2;
        } else if constexpr (TB) {
            const int64_t bytes = ((int64_t)(p.K - 1) * p.ldb + p.N - n0) * 4;
            rb = __builtin_amdgcn_make_buffer_rsrc((void*)(Bb + n0), 0, (int)(bytes > 0x7ffffffe ? 0x7ffffffe : bytes), 0x00020000);
#pragma unroll
            for (int q = 0; q < QB; ++q) {
                const int e = (w + NW * q) * 256 + lane * 4;
                vob[q] = ((e / BN) * (int)p.ldb + (e % BN)) * 4;
            }
            sb_step = BQ_KS * (int)p.ldb * 4;
        } else {
            const int64_t bbytes = ((int64_t)(p.N - 1 - n0) * p.ldb + p.K) * 4;           // (as for A: ends with the last valid element)
            rb = __builtin_amdgcn_make_buffer_rsrc((void*)(Bb + (int64_t)n0 * p.ldb), 0, (int)(bbytes > 0x7ffffffe ? 0x7ffffffe : bbytes), 0x00020000);
#pragma unroll
            for (int q = 0; q < QB; ++q) {
                const int row = min((w + NW * q) * 16 + lrow, p.N - 1 - n0);
                vob[q] = row * (int)p.ldb * 4 + lslot16;
            }
            sb_step = BQ_KS * 4;
        }
    };
    auto issue_advance = [&]() {
        if (++i_k >= i_hi) {
            if (i_local + nj < t_cnt) { i_local += nj; issue_tile(i_local); }   // past the last tile: harmless re-fetch
            else { int lo; k_range(decode(i_local).s, lo, i_hi); i_k = lo; }
        }
    };
    auto issue_slot = [&](int q, uint8_t* st, int ik) {
#if defined(BQ_LAB_NODMA)
        return;
#endif
        if (q < QA) __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (las_ptr)(st + (w + NW * q) * 1024), 16, voa[q], ik * sa_step, 0, 0);
        else __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (las_ptr)(st + BM * 64 + (w + NW * (q - QA)) * 1024), 16, vob[q - QA], ik * sb_step, 0, 0);
    };
    if (j0 >= t_cnt) return;                        // (more workgroups than tiles on this XCD)
    issue_tile(i_local);
#pragma unroll
    for (int s0 = 0; s0 < BQ_NS - 1; ++s0) {
#pragma unroll
        for (int q = 0; q < MAXQ; ++q) issue_slot(q, ring + s0 * STAGE, i_k);
        issue_advance();
    }

    const float alpha = p.alpha * (p.alpha_dev ? p.alpha_dev[0] : 1.0f);
    const int ktail = p.K & (BQ_KS - 1);            // valid elements of the last K-step (0 = all 16); N operands only

    // ---- The K-step, software-pipelined inside the wave and ACROSS the step boundary.
    // A wave issues in order: back to back, the split of a fragment (~52 VALU) and the MFMAs that use it do not overlap (PMC
    // on the first version: 41 % of a wave's cycles issuing, 44 % stalled on the matrix pipe, 16 % parked at the barrier /
    // waiting for its first fragments).  So (1) every split is interleaved, one pair of elements at a time, with the MFMAs of a
    // product group that does not need it (order pinned with sched_barrier: the machine scheduler clusters MFMAs otherwise);
    // (2) the fragments a step STARTS with -- all m fragments and n0 -- are read and split during the previous step, from the
    // next ring stage (the barrier at the top of step s publishes stage s + 1), so nothing is exposed after a barrier.
    // Jobs of step s, one per product group g = j RI + i:  g % RI == 0 -> n[j+1] of stage s;  the others, in order -> the carry
    // m0', m1', .., n0' of stage s + 1;  the groups left over issue this wave's DMA requests for stage s + BQ_NS - 1.
    constexpr int PBE = BP ? 3 : PB;
    constexpr int NP = (PA == 3 && PBE == 3) ? 6 : (PA == 1 && PBE == 1) ? 1 : (PA == 2 && PBE == 2) ? 3
                     : ((PA == 2 && PBE == 1) || (PA == 1 && PBE == 2)) ? 2 : 3;      // MFMAs per product group
    constexpr int NG = RI * CJ;
    constexpr int NCARRY = RI + 1;
    Frag3 mf[RI], nf[CJ], mfn[RI], nf0n;
    float raw[2][8];

    auto read_raw = [&](const uint8_t* stg, bool ism, int fi, bool lastk, float (&x)[8]) {
        const int row = ism ? wr * (BM / 2) + fi * 32 + frow : wc * (32 * CJ) + fi * 32 + frow;
        const uint8_t* base = ism ? stg : stg + BM * 64;
        const bool tr = ism ? TA : TB;
        const int ldt = ism ? BM : BN;
        if (tr) {
#pragma unroll
            for (int e = 0; e < 8; ++e) x[e] = lds_f1(base, ((8 * fkg + e) * ldt + row) * 4);
        } else {
            const uint4 u0 = lds_u4(base, swz64(row, 2 * fkg)), u1 = lds_u4(base, swz64(row, 2 * fkg + 1));
            x[0] = __uint_as_float(u0.x); x[1] = __uint_as_float(u0.y); x[2] = __uint_as_float(u0.z); x[3] = __uint_as_float(u0.w);
            x[4] = __uint_as_float(u1.x); x[5] = __uint_as_float(u1.y); x[6] = __uint_as_float(u1.z); x[7] = __uint_as_float(u1.w);
            if constexpr (KT) {
                if (lastk) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) if (8 * fkg + e >= ktail) x[e] = 0.0f;
                }
            }
        }
    };
    auto read_planes = [&](const uint8_t* stg, int j, Frag3& f) {
        const int row = wc * (32 * CJ) + j * 32 + frow;
        const uint8_t* pb = stg + BM * 64 + row * 32 + ((fkg ^ ((row >> 3) & 1)) << 4);
        f.hi = lds_u4(pb, 0); f.mid = lds_u4(pb, BN * 32); f.lo = lds_u4(pb, 2 * BN * 32);
    };
    // pair pr (0..3) of a fragment from raw x (elements 2 pr, 2 pr + 1); planes = 1: exact in bf16
    auto split_pair_into = [&](Frag3& f, int planes, const float (&x)[8], int pr) {
        uint32_t H, Mi = 0, L = 0;
        if (planes == 1) {
            v2bf h = {(__bf16)x[2 * pr], (__bf16)x[2 * pr + 1]};
            H = *reinterpret_cast<uint32_t*>(&h);
        } else {
#if defined(BQ_LAB_NOSPLIT)
            v2bf h = {(__bf16)x[2 * pr], (__bf16)x[2 * pr + 1]};
            H = *reinterpret_cast<uint32_t*>(&h); Mi = H; L = H;
#else
            if (planes == 2) split_pair2(x[2 * pr], x[2 * pr + 1], H, Mi);
            else split_pair(x[2 * pr], x[2 * pr + 1], H, Mi, L);
#endif
        }
        if (pr == 0) { f.hi.x = H; f.mid.x = Mi; f.lo.x = L; }
        else if (pr == 1) { f.hi.y = H; f.mid.y = Mi; f.lo.y = L; }
        else if (pr == 2) { f.hi.z = H; f.mid.z = Mi; f.lo.z = L; }
        else { f.hi.w = H; f.mid.w = Mi; f.lo.w = L; }
    };
    // job table in closed form (constant after unrolling).  kind: 0 = none, 1 = n[idx] of the current stage, 2 = carry m'[idx],
    // 3 = carry n0'.  Group g holds n[g / RI + 1] when g % RI == 0 and that fragment exists; the other groups take the carry
    // jobs in order; with RI == 1 one carry is left over and rides as a second job of the last group.
    auto job_kind = [](int g, int which) {
        const bool isn = (g % RI == 0) && (g / RI + 1 < CJ);
        const int nn = ((g + RI - 1) / RI) < (CJ - 1) ? ((g + RI - 1) / RI) : (CJ - 1);     // n jobs in groups before g
        const int c = g - nn;                                                                 // carries placed before g
        if (which == 0) {
            if (isn) return 16 + (g / RI + 1);
            if (c < NCARRY) return (c < RI ? 2 : 3) * 16 + c;
            return 0;
        }
        const int placed = (NG - (CJ - 1)) < NCARRY ? (NG - (CJ - 1)) : NCARRY;               // carries in primary slots
        if (g == NG - 1 && placed < NCARRY) return (placed < RI ? 2 : 3) * 16 + placed;       // (at most one left over)
        return 0;
    };
    static_assert(NG >= 2, "at least two product groups per step");

    // ---- prologue: the carry of the first step, from stage 0 (exposed once per workgroup)
    {
        wait_vm<MAXQ * (BQ_NS - 2)>();
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        const Tile t0 = decode(j0);
        const bool lk = KT && (p.kb[t0.s] == p.nk - 1);
#pragma unroll
        for (int i = 0; i < RI; ++i) {
            read_raw(ring, true, i, lk, raw[0]);
#pragma unroll
            for (int pr = 0; pr < 4; ++pr) split_pair_into(mf[i], PA, raw[0], pr);
        }
        if constexpr (BP) read_planes(ring, 0, nf[0]);
        else {
            read_raw(ring, false, 0, lk, raw[0]);
#pragma unroll
            for (int pr = 0; pr < 4; ++pr) split_pair_into(nf[0], PB, raw[0], pr);
        }
    }

    int st = 0;
    for (unsigned local = j0; local < t_cnt; local += nj) {
        const Tile tl = decode(local);
        const int m0 = tl.mt * BM, n0 = tl.nt * BN;
        int k_lo, k_hi;
        k_range(tl.s, k_lo, k_hi);
        // first K-step of the tile after this one (for the carry of this tile's last step)
        int kt_after = 0;
        if constexpr (KT) kt_after = local + nj < t_cnt ? p.kb[decode(local + nj).s] : 0;
        v16f acc[RI][CJ];
#pragma unroll
        for (int i = 0; i < RI; ++i)
#pragma unroll
            for (int j = 0; j < CJ; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

        for (int kt = k_lo; kt < k_hi; ++kt) {
            // stage st + 1 has landed once only this wave's newest BQ_NS - 3 steps of requests are outstanding
            wait_vm<MAXQ * (BQ_NS - 3)>();
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            const uint8_t* cur = ring + st * STAGE;
            const uint8_t* nx = ring + (st == BQ_NS - 1 ? 0 : st + 1) * STAGE;
            uint8_t* dst = ring + (st == 0 ? BQ_NS - 1 : st - 1) * STAGE;     // the slot read in the previous step
            const int ik = i_k;
            [[maybe_unused]] const bool last = KT && (kt == p.nk - 1);
            [[maybe_unused]] const bool last_n = KT && ((kt + 1 < k_hi ? kt + 1 : kt_after) == p.nk - 1);

            // raw read / plain read of a job
            auto job_read = [&](int code, float (&x)[8]) {
                const int kind = code >> 4, idx = code & 15;
                if (kind == 1) { if constexpr (BP) read_planes(cur, idx, nf[idx]); else read_raw(cur, false, idx, last, x); }
                else if (kind == 2) read_raw(nx, true, idx, last_n, x);
                else if (kind == 3) { if constexpr (BP) read_planes(nx, 0, nf0n); else read_raw(nx, false, 0, last_n, x); }
            };
            auto job_pair = [&](int code, const float (&x)[8], int pr) {
                const int kind = code >> 4, idx = code & 15;
                if (kind == 1) { if constexpr (!BP) split_pair_into(nf[idx], PB, x, pr); }
                else if (kind == 2) split_pair_into(mfn[idx], PA, x, pr);
                else if (kind == 3) { if constexpr (!BP) split_pair_into(nf0n, PB, x, pr); }
            };
            auto group_mfma = [&](int i, int j, int u) {
                v16f c = acc[i][j];
                const Frag3& n_ = nf[j]; const Frag3& m_ = mf[i];
                if constexpr (PA == 3 && PBE == 3) {
                    c = u == 0 ? mm(n_.lo, m_.hi, c) : u == 1 ? mm(n_.hi, m_.lo, c) : u == 2 ? mm(n_.mid, m_.mid, c)
                      : u == 3 ? mm(n_.mid, m_.hi, c) : u == 4 ? mm(n_.hi, m_.mid, c) : mm(n_.hi, m_.hi, c);
                } else if constexpr (PA == 2 && PBE == 2) {
                    c = u == 0 ? mm(n_.mid, m_.hi, c) : u == 1 ? mm(n_.hi, m_.mid, c) : mm(n_.hi, m_.hi, c);
                } else if constexpr (PA == 1 && PBE == 2) {
                    c = u == 0 ? mm(n_.mid, m_.hi, c) : mm(n_.hi, m_.hi, c);
                } else if constexpr (PA == 2 && PBE == 1) {
                    c = u == 0 ? mm(n_.hi, m_.mid, c) : mm(n_.hi, m_.hi, c);
                } else if constexpr (PA == 1 && PBE == 3) {
                    c = u == 0 ? mm(n_.lo, m_.hi, c) : u == 1 ? mm(n_.mid, m_.hi, c) : mm(n_.hi, m_.hi, c);
                } else if constexpr (PA == 3 && PBE == 1) {
                    c = u == 0 ? mm(n_.hi, m_.lo, c) : u == 1 ? mm(n_.hi, m_.mid, c) : mm(n_.hi, m_.hi, c);
                } else {
                    c = mm(n_.hi, m_.hi, c);
                }
                acc[i][j] = c;
            };

            job_read(job_kind(0, 0), raw[0]);                // the first group's job: its read is the only exposed one
            __builtin_amdgcn_sched_barrier(0);
            int qd = 0;                                      // DMA requests issued so far (compile-time after unrolling)
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                const int j = g / RI, i = g % RI;
                const int c0 = job_kind(g, 0), c1 = job_kind(g, 1);
                // reads of the jobs that follow, one group ahead (raw slots alternate; a second job of the last group uses the other)
                if (c1) job_read(c1, raw[(g + 1) & 1]);
                else if (g + 1 < NG) job_read(job_kind(g + 1, 0), raw[(g + 1) & 1]);
                __builtin_amdgcn_sched_barrier(0);
                constexpr int PER = (MAXQ + NG - 1) / NG;    // this wave's DMA requests, spread over the groups
#pragma unroll
                for (int u = 0; u < NP; ++u) {
                    group_mfma(i, j, u);
                    if (c0) {                                 // 4 pair-splits spread over the group's MFMAs
                        if (NP >= 4) { if (u < 4) job_pair(c0, raw[g & 1], u); }
                        else if (NP == 3) { job_pair(c0, raw[g & 1], u); if (u == 2) job_pair(c0, raw[g & 1], 3); }
                        else if (NP == 2) { job_pair(c0, raw[g & 1], 2 * u); job_pair(c0, raw[g & 1], 2 * u + 1); }
                        else { job_pair(c0, raw[g & 1], 0); job_pair(c0, raw[g & 1], 1); job_pair(c0, raw[g & 1], 2); job_pair(c0, raw[g & 1], 3); }
                    }
                    if (u == NP - 1) {
                        if (c1) {
#pragma unroll
                            for (int pr = 0; pr < 4; ++pr) job_pair(c1, raw[(g + 1) & 1], pr);
                        }
#pragma unroll
                        for (int q = 0; q < PER; ++q) if (qd + q < MAXQ) issue_slot(qd + q, dst, ik);
                        qd += PER;
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            // the carry becomes the next step's fragments
#pragma unroll
            for (int i = 0; i < RI; ++i) mf[i] = mfn[i];
            nf[0] = nf0n;
            issue_advance();
            st = st == BQ_NS - 1 ? 0 : st + 1;
        }

        // ---- epilogue: a lane owns, per accumulator quad, 4 consecutive n of one m.  The bias values of the wave's columns are
        // fetched first, then the stores issue back to back.
        const int cgo = __builtin_amdgcn_readfirstlane((unsigned)tl.g / (unsigned)p.Gi), cgi = tl.g - cgo * p.Gi;
        float* Cb = p.C + (p.S > 1 ? ((int64_t)tl.s * p.G + tl.g) * (int64_t)p.M * p.N : (int64_t)cgi * p.sCg + (int64_t)cgo * p.sCo);
        const int64_t ldc = p.S > 1 ? p.N : p.ldc;
        const float al = p.S > 1 ? 1.0f : alpha;
        const bool vec_store = ((p.N | (int)ldc | (int)(p.S > 1 ? 0 : (p.sCg | p.sCo))) & 3) == 0;
        // The bias of the wave's columns depends on the lane only through fkg: it comes through the SCALAR cache (two 16-float
        // loads per 32 columns, then a select).  A vector load here would do: but any VGPR-destination VMEM load inside the
        // persistent loop makes hipcc's waitcnt pass put a vmcnt(0) at the head of the K loop, which drains the DMA ring.
        float4 bv[CJ][4];
#pragma unroll
        for (int j = 0; j < CJ; ++j)
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) bv[j][q4] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (p.bias && p.S == 1) {                       // (host: N % 16 == 0 on this path)
#pragma unroll
            for (int j = 0; j < CJ; ++j) {
#pragma unroll
                for (int hh = 0; hh < 2; ++hh) {
                    const int nb = min(n0 + wc * (32 * CJ) + j * 32 + 16 * hh, p.N - 16);
                    const v16f sb = sload16(p.bias + nb);
#pragma unroll
                    for (int q2 = 0; q2 < 2; ++q2) {
                        float4& d = bv[j][2 * hh + q2];
                        d.x = fkg ? sb[8 * q2 + 4] : sb[8 * q2 + 0];
                        d.y = fkg ? sb[8 * q2 + 5] : sb[8 * q2 + 1];
                        d.z = fkg ? sb[8 * q2 + 6] : sb[8 * q2 + 2];
                        d.w = fkg ? sb[8 * q2 + 7] : sb[8 * q2 + 3];
                    }
                }
            }
        }
#pragma unroll
        for (int i = 0; i < RI; ++i) {
            const int m = m0 + wr * (BM / 2) + i * 32 + frow;
            float* Crow = Cb + (int64_t)min(m, p.M - 1) * ldc;
#pragma unroll
            for (int j = 0; j < CJ; ++j) {
#pragma unroll
                for (int q4 = 0; q4 < 4; ++q4) {
                    const int n = n0 + wc * (32 * CJ) + j * 32 + 8 * q4 + 4 * fkg;
                    float4 v;
                    v.x = acc[i][j][4 * q4] * al + bv[j][q4].x;
                    v.y = acc[i][j][4 * q4 + 1] * al + bv[j][q4].y;
                    v.z = acc[i][j][4 * q4 + 2] * al + bv[j][q4].z;
                    v.w = acc[i][j][4 * q4 + 3] * al + bv[j][q4].w;
#if defined(BQ_LAB_NOEPI)
                    if (v.x == 123.456f)
#endif
                    if (vec_store) {
                        if (m < p.M && n < p.N) *reinterpret_cast<float4*>(Crow + n) = v;
                    } else if (m < p.M) {                 // N or ldc not a multiple of 4 (attention: 197 keys): element stores
                        if (n < p.N) Crow[n] = v.x;
                        if (n + 1 < p.N) Crow[n + 1] = v.y;
                        if (n + 2 < p.N) Crow[n + 2] = v.z;
                        if (n + 3 < p.N) Crow[n + 3] = v.w;
                    }
                }
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // drain the run-ahead requests before the LDS is released
#endif
}

// C[g][m][n] = alpha * sum_s part[s][g][m][n] + bias[n]   (fixed order: bit-reproducible)
__global__ __launch_bounds__(256) void k_bq_reduce(const float* __restrict__ part, int S, int64_t gmn, int M, int N,
                                                   float* __restrict__ C, int64_t ldc, int64_t sCg, int Gi, int64_t sCo,
                                                   const float* __restrict__ bias, float alpha, const float* __restrict__ alpha_dev,
                                                   const float* __restrict__ addend) {
    // addend (optional): laid out like C, added last -- the residual stream of a transformer block (x + fc2(...)) rides in this pass
    const float a = alpha * (alpha_dev ? alpha_dev[0] : 1.0f);
    if (((N | (int)ldc | (int)(sCg | sCo)) & 3) != 0) {          // element form (N, ldc or the group stride not a multiple of 4)
        for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < gmn; e += (int64_t)gridDim.x * 256) {
            float acc = part[e];
            for (int s = 1; s < S; ++s) acc += part[(int64_t)s * gmn + e];
            const int64_t g = e / ((int64_t)M * N), r = e - g * (int64_t)M * N;
            const int m = (int)(r / N), n = (int)(r - (int64_t)m * N);
            const int64_t o = (g % Gi) * sCg + (g / Gi) * sCo + (int64_t)m * ldc + n;
            const float v = acc * a + (bias ? bias[n] : 0.0f);
            C[o] = addend ? v + addend[o] : v;
        }
        return;
    }
    const int64_t n4 = gmn >> 2;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        float4 acc = reinterpret_cast<const float4*>(part)[i];
        for (int s = 1; s < S; ++s) {
            const float4 v = reinterpret_cast<const float4*>(part + (int64_t)s * gmn)[i];
            acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        }
        const int64_t e = i << 2;
        const int64_t g = e / ((int64_t)M * N), r = e - g * (int64_t)M * N;
        const int m = (int)(r / N), n = (int)(r - (int64_t)m * N);
        acc.x *= a; acc.y *= a; acc.z *= a; acc.w *= a;
        if (bias) {
            const float4 b = *reinterpret_cast<const float4*>(bias + n);
            acc.x += b.x; acc.y += b.y; acc.z += b.z; acc.w += b.w;
        }
        const int64_t o = (g % Gi) * sCg + (g / Gi) * sCo + (int64_t)m * ldc + n;
        if (addend) {
            const float4 r = *reinterpret_cast<const float4*>(addend + o);
            acc.x += r.x; acc.y += r.y; acc.z += r.z; acc.w += r.w;
        }
        *reinterpret_cast<float4*>(C + o) = acc;
    }
}

// C += addend over the same [G][M][N] layout: the products that are NOT split along K take the residual in a pass of their own
__global__ __launch_bounds__(256) void k_bq_addend(float* __restrict__ C, const float* __restrict__ addend, int64_t gmn, int M, int N,
                                                   int64_t ldc, int64_t sCg, int Gi, int64_t sCo) {
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < gmn; e += (int64_t)gridDim.x * 256) {
        const int64_t g = e / ((int64_t)M * N), r = e - g * (int64_t)M * N;
        const int m = (int)(r / N), n = (int)(r - (int64_t)m * N);
        const int64_t o = (g % Gi) * sCg + (g / Gi) * sCo + (int64_t)m * ldc + n;
        C[o] += addend[o];
    }
}

struct BqPlan { int shape, S, MT, NT, nk, wgs; };

int bq_cus() {
    static int n_cu = 0;
    if (!n_cu) {
        int dev = 0; hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) n_cu = prop.multiProcessorCount;
        if (n_cu <= 0) n_cu = 256;
    }
    return n_cu;
}

// Tile shapes: (RI, CJ, WN) -> rows x columns, workgroups that fit a CU, relative rate of a full tile
struct BqShape { int ri, cj, wn, bm, bn, per_cu; double speed; const char* name; };
static const BqShape BQ_SHAPES[5] = {
    {2, 2, 4, 128, 256, 1, 1.00, "bq_gemm<128x256,8w>"},    // 8 waves: two per SIMD in ONE workgroup
    {2, 4, 2, 128, 256, 1, 1.00, "bq_gemm<128x256,4w>"},    // one wave per SIMD, 128 accumulator registers (fewest splits per MFMA)
    {2, 2, 2, 128, 128, 2, 0.80, "bq_gemm<128x128,4w>"},
    {1, 2, 2, 64, 128, 3, 0.60, "bq_gemm<64x128,4w>"},
    {3, 2, 2, 192, 128, 2, 0.90, "bq_gemm<192x128,4w>"},    // for tile counts that 128 x 128 leaves just over one round of the chip
};

// workgroups of a shape that fit the chip at once
int bq_slots(int shape, bool bp) {
    const BqShape& sh = BQ_SHAPES[shape];
    int per_cu = (160 * 1024) / (BQ_NS * (sh.bm * 64 + sh.bn * (bp ? 96 : 64)));
    if (per_cu > sh.per_cu) per_cu = sh.per_cu;
    if (per_cu < 1) per_cu = 1;
    return bq_cus() * per_cu;
}

// Tile shape and K split.  Rules from the MI355X measurements of tools/lab/bq_lab (profiles/r04_notes.md): the products of a
// BRECQ iteration are small (5-45 GFLOP of bf16 work), so what decides is how evenly the work items cover the chip: 128 x 128
// tiles, two workgroups per CU, whenever that gives >= ~400 items; otherwise the K range is cut so that it does (the partial tiles
// are added in a fixed order), provided every range keeps >= 24 K-steps; few-tile products (dL/dw) take 64 x 128 tiles.
BqPlan bq_plan(int M, int N, int K, int G, int allow_split, int products, int bp = 0) {
    (void)products;
    const int nk = cdiv(K, BQ_KS);
    int shape = 2;
    if (const char* e = getenv("ADALOG_BQ_SHAPE")) { const int v = atoi(e); if (v >= 0 && v < 5) shape = v; else shape = -1; }
    else {
        const int64_t t2 = (int64_t)cdiv(M, 128) * cdiv(N, 128) * G;
        shape = t2 >= 64 ? 2 : 3;
        // a tile count just over a whole number of rounds of the chip (600 tiles on 512 slots: the second round runs at 17 %):
        // 192 x 128 tiles when they take fewer rounds (a 192-row tile costs ~1.6 of a 128-row one; measured fc1 forward
        // 53 -> 44 us, tools/lab/bq_lab)
        static const int use_s4 = getenv("ADALOG_BQ_S4") ? atoi(getenv("ADALOG_BQ_S4")) : 1;
        if (t2 >= 400 && !bp && use_s4) {
            const int64_t t4 = (int64_t)cdiv(M, 192) * cdiv(N, 128) * G;
            const int s2 = bq_slots(2, false), s4 = bq_slots(4, false);
            const double c2 = (double)cdiv(t2, s2), c4 = 1.6 * (double)cdiv(t4, s4);
            if (c4 < 0.95 * c2) shape = 4;
        }
    }
    if (shape < 0) shape = 2;
    const BqShape& sh = BQ_SHAPES[shape];
    const int MT = cdiv(M, sh.bm), NT = cdiv(N, sh.bn);
    const int64_t tiles = (int64_t)MT * NT * G;
    const int slots = bq_slots(shape, bp != 0);
    int S = 1;
    if (allow_split && tiles < 400) {
        S = (int)((480 + tiles / 2) / tiles);
        int smax = nk / 24;
        if (smax > 16) smax = 16;
        if (S > smax) S = smax;
        if (S < 1) S = 1;
    }
    if (const char* e = getenv("ADALOG_BQ_SPLIT")) { const int v = atoi(e); if (v >= 1 && allow_split) S = v > nk ? nk : v; }
    const int64_t items = tiles * S;
    return {shape, S, MT, NT, nk, (int)(items < slots ? items : slots)};
}

template <int RI, int CJ, int WN, bool TA, bool TB, int PA, int PB, bool KT>
int bq_go(const BqArgs& a, int wgs, hipStream_t st) {
    const size_t lds = (size_t)BQ_NS * (64 * RI * 64 + 32 * CJ * WN * (PB == 0 ? 96 : 64));
    static unsigned long long attr_done = 0;                 // one bit per device: the attribute is per device, not per process
    int dev = 0;
    (void)hipGetDevice(&dev);
    const unsigned long long bit = 1ull << (dev & 63);
    if (!(attr_done & bit)) {
        if (hipFuncSetAttribute((const void*)k_bq_gemm<RI, CJ, WN, TA, TB, PA, PB, KT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return -2;
        attr_done |= bit;
    }
    hipLaunchKernelGGL((k_bq_gemm<RI, CJ, WN, TA, TB, PA, PB, KT>), dim3(wgs), dim3(128 * WN), lds, st, a);
    return 0;
}

// general fp32 operands (6 products): every orientation, with and without a K tail;
// integer operand forms (3 products): forward (A = x_int, K-contiguous both) and dL/dw (B = x_int, K-major both)
template <int RI, int CJ, int WN>
int bq_launch(const BqArgs& a, int transA, int transB, int pa, int pb, bool kt, int wgs, hipStream_t st) {
    if (pb == 0) {                                   // B = pre-split planes (K-contiguous); A K-contiguous
        if (transA || transB) return -3;
        if (pa == 1) return kt ? bq_go<RI, CJ, WN, false, false, 1, 0, true>(a, wgs, st) : bq_go<RI, CJ, WN, false, false, 1, 0, false>(a, wgs, st);
        return kt ? bq_go<RI, CJ, WN, false, false, 3, 0, true>(a, wgs, st) : bq_go<RI, CJ, WN, false, false, 3, 0, false>(a, wgs, st);
    }
    if (pa == 1 && pb == 2 && !transA && !transB && !kt) return bq_go<RI, CJ, WN, false, false, 1, 2, false>(a, wgs, st);
    if (pa == 1 && pb == 2 && !transA && transB && !kt) return bq_go<RI, CJ, WN, false, true, 1, 2, false>(a, wgs, st);
    if (pa == 1 && pb == 3 && !transA && !transB && !kt) return bq_go<RI, CJ, WN, false, false, 1, 3, false>(a, wgs, st);
    if (pa == 1 && pb == 3 && !transA && transB && !kt) return bq_go<RI, CJ, WN, false, true, 1, 3, false>(a, wgs, st);
    if (pa == 3 && pb == 1 && transA && transB) return bq_go<RI, CJ, WN, true, true, 3, 1, false>(a, wgs, st);
    // two-term operands (gradient contractions): every orientation for the general pair, the dL/dw form against an exact operand
    if (pa == 2 && pb == 1 && transA && transB) return bq_go<RI, CJ, WN, true, true, 2, 1, false>(a, wgs, st);
    if (pa == 2 && pb == 2) {
        if (transA && transB) return bq_go<RI, CJ, WN, true, true, 2, 2, false>(a, wgs, st);
        if (kt) {
            if (transA) return bq_go<RI, CJ, WN, true, false, 2, 2, true>(a, wgs, st);
            if (transB) return bq_go<RI, CJ, WN, false, true, 2, 2, true>(a, wgs, st);
            return bq_go<RI, CJ, WN, false, false, 2, 2, true>(a, wgs, st);
        }
        if (transA) return bq_go<RI, CJ, WN, true, false, 2, 2, false>(a, wgs, st);
        if (transB) return bq_go<RI, CJ, WN, false, true, 2, 2, false>(a, wgs, st);
        return bq_go<RI, CJ, WN, false, false, 2, 2, false>(a, wgs, st);
    }
    if (pa != 3 || pb != 3) return -3;
    if (transA && transB) return bq_go<RI, CJ, WN, true, true, 3, 3, false>(a, wgs, st);
    if (kt) {
        if (transA) return bq_go<RI, CJ, WN, true, false, 3, 3, true>(a, wgs, st);
        if (transB) return bq_go<RI, CJ, WN, false, true, 3, 3, true>(a, wgs, st);
        return bq_go<RI, CJ, WN, false, false, 3, 3, true>(a, wgs, st);
    }
    if (transA) return bq_go<RI, CJ, WN, true, false, 3, 3, false>(a, wgs, st);
    if (transB) return bq_go<RI, CJ, WN, false, true, 3, 3, false>(a, wgs, st);
    return bq_go<RI, CJ, WN, false, false, 3, 3, false>(a, wgs, st);
}

}  // namespace

// exact: 0 = general fp32 operand (three-term split), 1 = exactly representable in bf16, 2 = two-term split (2^-16 relative)
static void bq_forms(int exactA, int exactB, int transA, int transB, int& pa, int& pb) {
    pa = pb = 3;
    if (exactA == 1 && exactB != 1 && !transA) { pa = 1; pb = exactB == 2 ? 2 : 3; return; }   // forward: A = x_int K-contiguous, B either orientation
    if (exactB == 1 && exactA != 1 && transA && transB) { pb = 1; pa = exactA == 2 ? 2 : 3; return; }   // dL/dw: B = x_int, both K-major
    if (exactA == 2 && exactB == 2) { pa = pb = 2; return; }
    // (a two-term request next to a general operand, or an exact operand in an orientation without a dedicated form, runs as the
    // general product: the split of an exact value is (x, 0, 0), the result the same or better)
}
static int bq_products(int exactA, int exactB, int transA, int transB) {
    int pa, pb;
    bq_forms(exactA, exactB, transA, transB, pa, pb);
    return (pa == 3 && pb == 3) ? 6 : ((pa == 2 && pb == 1) || (pa == 1 && pb == 2)) ? 2 : 3;
}

// Bytes of workspace adalog_gemm_f32x3 needs for this shape (0 when the product is not split along K).
extern "C" int64_t adalog_gemm_f32x3_workspace_bytes(int M, int N, int K, int G, int allow_split, int exactA, int exactB,
                                                     int transA, int transB) {
    if (M < 1 || N < 1 || K < 1 || G < 1) return 0;
    const BqPlan pl = bq_plan(M, N, K, G, allow_split, bq_products(exactA, exactB, transA, transB));
    return pl.S > 1 ? (int64_t)pl.S * G * M * N * 4 : 0;
}

static int bq_run(const float* A, int64_t lda, int transA, const void* B, int64_t ldb, int transB, int64_t bplane, float* C,
                  int64_t ldc, int M, int N, int K, int G, int64_t sAg, int64_t sBg, int64_t sCg, const float* bias, float alpha,
                  const float* alpha_dev, int allow_split, int pa, int pb, float* workspace, void* stream, int Gi = 0,
                  int64_t sAo = 0, int64_t sBo = 0, int64_t sCo = 0, const float* addend = nullptr) {
    // the integer forward form is built without the K-tail masking: with K % 16 != 0 it runs as a general product (same result)
    if (pa == 1 && pb == 3 && (K & (BQ_KS - 1)) != 0) pa = 3;
    if (pa == 1 && pb == 2 && (K & (BQ_KS - 1)) != 0) pa = 2;
    const int products = (pa == 3 && pb != 1) ? 6 : (pa == 1 && pb == 1) ? 1 : ((pa == 2 && pb == 1) || (pa == 1 && pb == 2)) ? 2 : 3;
    const BqPlan pl = bq_plan(M, N, K, G, allow_split, products, pb == 0);
    ADALOG_ARG_CHECK(pl.S == 1 || workspace, "gemm_f32x3: the split product needs its workspace");
    ADALOG_ARG_CHECK((((uintptr_t)workspace) & 15) == 0, "gemm_f32x3: workspace must be 16-byte aligned");
    BqArgs a;
    a.A = A; a.B = (const float*)B; a.C = pl.S > 1 ? workspace : C;
    a.lda = lda; a.ldb = ldb; a.ldc = ldc; a.sAg = sAg; a.sBg = sBg; a.sCg = sCg; a.bplane = bplane;
    a.Gi = (Gi > 0 && Gi < G) ? Gi : G; a.sAo = sAo; a.sBo = sBo; a.sCo = sCo;
    a.M = M; a.N = N; a.K = K; a.G = G; a.S = pl.S; a.MT = pl.MT; a.NT = pl.NT; a.nk = pl.nk;
    a.bias = pl.S > 1 ? nullptr : bias; a.alpha = alpha; a.alpha_dev = alpha_dev;
    for (int s = 0; s <= pl.S; ++s) a.kb[s] = (int)(((int64_t)pl.nk * s) / pl.S);
    hipStream_t st = (hipStream_t)stream;
    const bool kt = (K & (BQ_KS - 1)) != 0 && !(transA && transB);
    int rc = 0;
    switch (pl.shape) {
        case 0: rc = bq_launch<2, 2, 4>(a, transA, transB, pa, pb, kt, pl.wgs, st); break;
        case 1: rc = bq_launch<2, 4, 2>(a, transA, transB, pa, pb, kt, pl.wgs, st); break;
        case 2: rc = bq_launch<2, 2, 2>(a, transA, transB, pa, pb, kt, pl.wgs, st); break;
        case 4: rc = bq_launch<3, 2, 2>(a, transA, transB, pa, pb, kt, pl.wgs, st); break;
        default: rc = bq_launch<1, 2, 2>(a, transA, transB, pa, pb, kt, pl.wgs, st); break;
    }
    ADALOG_ARG_CHECK(rc == 0, "gemm_f32x3: cannot launch (LDS size attribute / unsupported operand form)");
    ADALOG_LAUNCH_CHECK("adalog_gemm_f32x3");
    adalog_note_kernel(BQ_SHAPES[pl.shape].name);
    if (pl.S > 1) {
        const int64_t gmn = (int64_t)G * M * N;
        int blocks = (int)((gmn / 4 + 255) / 256);
        if (blocks > 2048) blocks = 2048;
        hipLaunchKernelGGL(k_bq_reduce, dim3(blocks), dim3(256), 0, st, workspace, pl.S, gmn, M, N, C, ldc, sCg, a.Gi, sCo, bias, alpha, alpha_dev,
                           addend);
        ADALOG_LAUNCH_CHECK("adalog_gemm_f32x3/reduce");
    } else if (addend) {
        const int64_t gmn = (int64_t)G * M * N;
        int blocks = (int)((gmn + 1023) / 1024);
        if (blocks > 4096) blocks = 4096;
        hipLaunchKernelGGL(k_bq_addend, dim3(blocks), dim3(256), 0, st, C, addend, gmn, M, N, ldc, sCg, a.Gi, sCo);
        ADALOG_LAUNCH_CHECK("adalog_gemm_f32x3/addend");
    }
    return 0;
}

// C[g] = alpha * alpha_dev[0] * opA(A[g]) . opB(B[g])^T + bias        (fp32 in, fp32 out, fp32-class accuracy: see the header)
//   opA(A)[m][k] = transA ? A[k * lda + m] : A[m * lda + k]         opB(B)[n][k] = transB ? B[k * ldb + n] : B[n * ldb + k]
// exactA / exactB: the caller guarantees that operand's values are exactly representable in bf16 (integers of magnitude
//   <= 256: the integer part of a uniformly fake-quantised activation); honoured for the forward (exactA, A K-contiguous)
//   and the dL/dw (exactB, both K-major) forms -- 3 products instead of 6 -- and ignored otherwise (the split of such a value is
//   (x, 0, 0), so the result is the same).
// Requirements: A, B, C, bias 16-byte aligned; lda, ldb, ldc, sAg, sBg, sCg multiples of 4 elements; N a multiple of 4.
// workspace: adalog_gemm_f32x3_workspace_bytes(...) bytes (may be null when that is 0).
// Two-level groups (adalog_gemm_f32x3_g2): group g = go * Gi + gi sits at gi * s?g + go * s?o in each operand -- a [B][H] batch whose
//   two strides do not collapse, e.g. the softmax.v product writing its [B][N][H][D] result in place (no transposed copy before
//   the projection layer).  Gi <= 0 or Gi >= G: one level (the outer strides are ignored).
extern "C" int adalog_gemm_f32x3_g2(const float* A, int64_t lda, int transA, const float* B, int64_t ldb, int transB, float* C,
                                    int64_t ldc, int M, int N, int K, int G, int64_t sAg, int64_t sBg, int64_t sCg, int Gi,
                                    int64_t sAo, int64_t sBo, int64_t sCo, const float* bias, float alpha, const float* alpha_dev,
                                    int allow_split, int exactA, int exactB, float* workspace, void* stream);
// ... + addend[g][m][n] (laid out like C; may be null): added in the split product's reduction pass (no extra launch) or, for a product
// that is not split, by a pass of its own -- x + fc2(...) of a transformer block inside a BRECQ iteration.
extern "C" int adalog_gemm_f32x3_add(const float* A, int64_t lda, int transA, const float* B, int64_t ldb, int transB, float* C,
                                     int64_t ldc, int M, int N, int K, int G, int64_t sAg, int64_t sBg, int64_t sCg,
                                     const float* bias, float alpha, const float* alpha_dev, int allow_split, int exactA, int exactB,
                                     const float* addend, float* workspace, void* stream) {
    if (M == 0 || N == 0 || G == 0) return 0;
    ADALOG_ARG_CHECK(A && B && C && M > 0 && N > 0 && K > 0 && G > 0, "gemm_f32x3_add: bad arguments");
    ADALOG_ARG_CHECK((((uintptr_t)A | (uintptr_t)B | (uintptr_t)C | (uintptr_t)bias | (uintptr_t)addend) & 15) == 0, "gemm_f32x3_add: operands must be 16-byte aligned");
    ADALOG_ARG_CHECK(!bias || (N & 15) == 0, "gemm_f32x3_add: a fused bias needs N to be a multiple of 16");
    ADALOG_ARG_CHECK(lda >= (transA ? M : K) && ldb >= (transB ? N : K) && ldc >= N, "gemm_f32x3_add: leading dimensions too small");
    ADALOG_ARG_CHECK((int64_t)(transA ? K : M) * lda * 4 < 0x7fffffffLL && (int64_t)(transB ? K : N) * ldb * 4 < 0x7fffffffLL,
                     "gemm_f32x3_add: an operand matrix exceeds 2 GiB");
    int pa, pb;
    bq_forms(exactA, exactB, transA, transB, pa, pb);
    return bq_run(A, lda, transA, B, ldb, transB, 0, C, ldc, M, N, K, G, sAg, sBg, sCg, bias, alpha, alpha_dev, allow_split, pa, pb,
                  workspace, stream, 0, 0, 0, 0, addend);
}
extern "C" int adalog_gemm_f32x3(const float* A, int64_t lda, int transA, const float* B, int64_t ldb, int transB, float* C,
                                 int64_t ldc, int M, int N, int K, int G, int64_t sAg, int64_t sBg, int64_t sCg,
                                 const float* bias, float alpha, const float* alpha_dev, int allow_split, int exactA, int exactB,
                                 float* workspace, void* stream) {
    return adalog_gemm_f32x3_g2(A, lda, transA, B, ldb, transB, C, ldc, M, N, K, G, sAg, sBg, sCg, 0, 0, 0, 0, bias, alpha, alpha_dev,
                                allow_split, exactA, exactB, workspace, stream);
}
extern "C" int adalog_gemm_f32x3_g2(const float* A, int64_t lda, int transA, const float* B, int64_t ldb, int transB, float* C,
                                    int64_t ldc, int M, int N, int K, int G, int64_t sAg, int64_t sBg, int64_t sCg, int Gi,
                                    int64_t sAo, int64_t sBo, int64_t sCo, const float* bias, float alpha, const float* alpha_dev,
                                    int allow_split, int exactA, int exactB, float* workspace, void* stream) {
    if (M == 0 || N == 0 || G == 0) return 0;
    ADALOG_ARG_CHECK(Gi <= 0 || Gi >= G || G % Gi == 0, "gemm_f32x3: the inner group count must divide G");
    ADALOG_ARG_CHECK(A && B && C && M > 0 && N > 0 && K > 0 && G > 0, "gemm_f32x3: bad arguments");
    // (rows need not be 16-byte aligned: the LDS-DMA requests are dword-aligned 16-byte loads; N, ldc or sCg off a multiple
    // of 4 -- attention products with 197 tokens -- take the element-store epilogue)
    ADALOG_ARG_CHECK((((uintptr_t)A | (uintptr_t)B | (uintptr_t)C | (uintptr_t)bias) & 15) == 0, "gemm_f32x3: operands must be 16-byte aligned");
    ADALOG_ARG_CHECK(!bias || (N & 15) == 0, "gemm_f32x3: a fused bias needs N to be a multiple of 16");
    ADALOG_ARG_CHECK(lda >= (transA ? M : K) && ldb >= (transB ? N : K) && ldc >= N, "gemm_f32x3: leading dimensions too small");
    ADALOG_ARG_CHECK((int64_t)(transA ? K : M) * lda * 4 < 0x7fffffffLL && (int64_t)(transB ? K : N) * ldb * 4 < 0x7fffffffLL,
                     "gemm_f32x3: an operand matrix exceeds 2 GiB");
    int pa, pb;
    bq_forms(exactA, exactB, transA, transB, pa, pb);
    return bq_run(A, lda, transA, B, ldb, transB, 0, C, ldc, M, N, K, G, sAg, sBg, sCg, bias, alpha, alpha_dev, allow_split, pa, pb,
                  workspace, stream, Gi, sAo, sBo, sCo);
}

// The same product with B handed over ALREADY SPLIT into three bf16 planes (adalog_pack_split3_bf16's layout: row n of group g
// at Bp + (g * N + n) * 3 * Kt, its planes hi | mid | lo of Kt >= K elements each, zero beyond K; Kt % 32 == 0):
//   C[g][m][n] = alpha * alpha_dev[0] * sum_k A[g][m * lda + k] * (hi + mid + lo)[g][n][k] + bias[n]
// For an operand that many row tiles re-read (the soft-rounded weights w_sim and w_sim^T of a Linear layer): the split runs once
// per iteration in the packer instead of once per row tile in the GEMM's registers.  A is K-contiguous; exactA as above.
extern "C" int64_t adalog_gemm_f32x3_planes_workspace_bytes(int M, int N, int K, int G, int allow_split, int exactA) {
    if (M < 1 || N < 1 || K < 1 || G < 1) return 0;
    const BqPlan pl = bq_plan(M, N, K, G, allow_split, exactA ? 3 : 6, 1);
    return pl.S > 1 ? (int64_t)pl.S * G * M * N * 4 : 0;
}
extern "C" int adalog_gemm_f32x3_planes(const float* A, int64_t lda, const void* Bp, int64_t Kt, float* C, int64_t ldc, int M, int N,
                                        int K, int G, int64_t sAg, int64_t sCg, const float* bias, float alpha,
                                        const float* alpha_dev, int allow_split, int exactA, float* workspace, void* stream) {
    if (M == 0 || N == 0 || G == 0) return 0;
    ADALOG_ARG_CHECK(A && Bp && C && M > 0 && N > 0 && K > 0 && G > 0, "gemm_f32x3_planes: bad arguments");
    ADALOG_ARG_CHECK((((uintptr_t)A | (uintptr_t)Bp | (uintptr_t)C | (uintptr_t)bias) & 15) == 0, "gemm_f32x3_planes: operands must be 16-byte aligned");
    ADALOG_ARG_CHECK(((lda | ldc | sAg | sCg) & 3) == 0 && (N & 3) == 0, "gemm_f32x3_planes: strides and N must be multiples of 4");
    ADALOG_ARG_CHECK(Kt >= K && (Kt & 31) == 0, "gemm_f32x3_planes: Kt must cover K and be a multiple of 32");
    ADALOG_ARG_CHECK(!bias || (N & 15) == 0, "gemm_f32x3_planes: a fused bias needs N to be a multiple of 16");
    ADALOG_ARG_CHECK(lda >= K && ldc >= N, "gemm_f32x3_planes: leading dimensions too small");
    ADALOG_ARG_CHECK((int64_t)M * lda * 4 < 0x7fffffffLL && (int64_t)N * 3 * Kt * 2 < 0x7fffffffLL, "gemm_f32x3_planes: an operand matrix exceeds 2 GiB");
    return bq_run(A, lda, 0, Bp, 3 * Kt, 0, Kt, C, ldc, M, N, K, G, sAg, (int64_t)N * 3 * Kt, sCg, bias, alpha, alpha_dev, allow_split,
                  exactA ? 1 : 3, 0, workspace, stream);
}
