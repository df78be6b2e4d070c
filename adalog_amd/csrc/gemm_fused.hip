// K11 -- post-GELU activation-candidate search with the AdaLog quantisation FUSED INTO THE GEMM'S LOADER.
//
// Replaces, for one scoring call of reference quant_layers/linear.py:816-848 / :856-890 / :898-931
//   x_sim[N,T,I,P] = AdaLog_p(x + shift) * s_p - shift   (P = 128 candidates (s_p, q_p), ~12 elementwise ATen ops)
//   out_sim        = F.linear(x_sim, q_w(W), b)          ->  -(raw_out - out_sim)^2  ->  mean / sum  ->  scores [P]
// by ONE kernel that never materialises the candidate operand: round 1 wrote it to HBM as bf16 (2.5 GB per call for
// deit_small fc2: k_pack_adalog_fast 0.4 ms) and read it back in the scoring GEMM (0.99 ms, 2.7 GB: 55x the
// algorithmic bytes).  Here the only HBM streams are x and log2(x + shift) (fp32, 2 x 38.7 MB, read once), raw_out
// (9.7 MB) and the bf16 weight image (1.2 MB, L2-resident).
//
// Evaluated transposed, like the streaming kernel's activation searches:  D[o, (t, p)] = sum_k Wq[o, k] * v_p(x[t, k]),
// GEMM rows = output channels, GEMM columns = (token, candidate), tile = (32 * NRB) rows x 2 tokens x 128 candidates.
// Two forms of the same algorithm (same LDS tables, same results):
//   - the shipped one, k_act_fused_asm_* below: EIGHT waves per workgroup (two per SIMD), wave = (token of the pair,
//     candidate block of 32), NRB accumulator tiles each, hand-scheduled (tools/gen_fused_asm.py, whose header holds the
//     measured machine model behind that shape);
//   - the compiler form k_act_fused<NRB, FNS> (ADALOG_FUSED_ASM=0; 4 waves, one per SIMD, 2 x NRB tiles each), described
//     in the rest of this comment: it is the readable statement of the algorithm and the first correct implementation.
//   * wave w owns token (w >> 1) and candidate blocks 2 * (w & 1), 2 * (w & 1) + 1: NRB x 2 accumulator tiles of 32 x 32
//     (NRB = 12: 384 registers).  A lane IS a candidate column: its (37/q_p, log2(s_p) * 37/q_p, clamp) live in registers;
//   * the B fragments (candidate operand) are produced in registers, straight into MFMA operand layout: a lane needs 8
//     consecutive k of its (token, candidate) per v_mfma_f32_32x32x16_bf16, i.e. per element-candidate
//         kf = med3(fma(L, -37/q, c), lo, hi);  t = kf + 1.5*2^23;  d = kf - (t - 1.5*2^23);
//         value = LUT[(bits(t) << 9) + lane const]   (ds_read_b32: dword table [bin][candidate], bank = candidate: no
//         conflicts);  two values pack into one dword;  max3 over |d| flags near-ties          -- ~7 VALU + 1 LDS read;
//     L = log2(x + shift) is precomputed once per layer (adalog_log2_shift, correctly rounded) and reaches the wave through
//     LDS together with x (one 256-byte DMA each per K-step and wave, broadcast reads): no transcendental, no division
//     and no fp64 in this kernel;
//   * EXACT bins through a THRESHOLD TABLE.  The reference's fp32 pipeline  xs = x + shift;  u = clamp(xs / s, 1e-15, 1);
//     k = rne(fl(fl(-log2(u) * 37) / q))  is a monotone non-increasing step function of xs for a fixed candidate (every
//     step is a correctly rounded monotone operation), so it is fully described by its break points: thr[b][p] = the
//     smallest float xs with k <= b.  adalog_tie_thresholds finds them per scoring call by bisection over the float
//     ordering with the exact pipeline (IEEE divide, correctly rounded log2 through fp64: common.h's rule) -- 128 x 2^bits
//     searches of 31 steps, a few microseconds.  The fast kf above is within ~2e-5 of that pipeline (error budget in
//     DESIGN.md); when a chunk of 8 element-candidates holds one within 1e-4 of a rounding tie, a cold block re-derives the
//     chunk's bins as  fast bin + [xs < thr[bin]] - [xs >= thr[bin - 1]]  (two LDS reads and two compares) and re-reads
//     the LUT.  The operand is therefore bit-identical to what k_pack_adalog_fast (operand.hip) writes;
//   * the weight tile (32 * NRB rows x 64 bytes per K-step) streams through an LDS-DMA ring shared by the four waves
//     (buffer_load ... lds, swizzled source slots, counted vmcnt + one barrier per K-step) -- 16 B/clk/CU at the matrix
//     pipe's full rate; an A fragment read feeds two MFMAs.  The ring and the fragment generation run ahead across tile
//     boundaries (persistent workgroups);
//   * epilogue in registers: e = (ref - row_bias) - D * (s_p * ts) * s_w[o], the lane adds e^2 over its rows, lanes l and
//     l + 32 combine, per-wave fp64 running sums per candidate; one [workgroup][128] fp64 row leaves the kernel and a
//     fixed-order finish turns the rows into scores (bit-reproducible).
//
// MFMA roofline: 2 * M * (T * 128) * K flops per launch at the bf16 rate (2.5 PFLOP/s dense).
#include "common.h"
#include <stdlib.h>

namespace {

typedef float v16f __attribute__((ext_vector_type(16)));
typedef __bf16 v8bf __attribute__((ext_vector_type(8)));
typedef unsigned u4v __attribute__((ext_vector_type(4)));
typedef void __attribute__((address_space(3)))* las_ptr;
typedef const uint32_t __attribute__((address_space(3)))* lds_u32p;
typedef const float __attribute__((address_space(3)))* lds_f32p;

struct FusedArgs {
    const uint8_t* W;          // [M][Kb] bf16 image of q_w(W) - z (integers), rows zero-padded to Kb bytes
    const float* L;            // [T][K]  log2(x + shift)  (-inf where x + shift <= 0)
    const float* x;            // [T][K]  the captured activation
    const float* ref;          // [T][M]  raw_out
    const float* row_scale;    // [M]     weight scale s_w[o]
    const float* row_bias;     // [M]     bias with the -shift term folded in (may be null)
    const float* scale;        // [128]   candidate scales s_p
    const float* qv;           // [128]   candidate bases q_p (as floats)
    const float* mant;         // [37]    integer numerators of the search-time mantissa table (linear.py:750-752)
    const float* thr;          // [levels2][128] break points of the exact pipeline (k_tie_thresholds)
    double* wg_acc;            // [gridDim.x][128]
    int M, T, K;
    int Kb;                    // row pitch of W in bytes (multiple of 128)
    int levels2;               // 2^bits
    int n_rt;                  // row tiles of 32 * NRB rows
    int nk;                    // 64-byte K-steps
    float shift, sa_mul;
};

constexpr float MAGIC = 12582912.0f;         // 1.5 * 2^23: fl32(kf + MAGIC) carries rne(kf) in its low mantissa bits
constexpr unsigned MAGIC_BITS = 0x4B400000u;
constexpr float TIE = 0.4999f;               // |kf - rne(kf)| above this: the threshold table decides

// Fragment / parameter reads through __restrict__ helpers: the loads carry alias scopes, so the waitcnt pass does not
// order them behind the (untagged) in-flight LDS-DMA with a vmcnt(0) -- the counted vmcnt before each barrier does that.
__device__ __forceinline__ uint4 lds_frag(const uint8_t* __restrict__ stage, int off) {
    return *reinterpret_cast<const uint4*>(stage + off);
}
__device__ __forceinline__ float4 lds_f4(const uint8_t* __restrict__ base, int off) {
    return *reinterpret_cast<const float4*>(base + off);
}
__device__ __forceinline__ v8bf frag8(const uint32_t (&d)[4]) {
    const u4v v = {d[0], d[1], d[2], d[3]};
    return __builtin_bit_cast(v8bf, v);
}

// The compiler selects ONE form of MFMA per function (accumulators in AGPRs when the wave may use 512 registers), so it
// cannot keep more than 256 accumulator registers without copying tiles in and out around every MFMA.  Hand-placed
// classes: the tiles of the first NA row blocks use the compiler's form (AGPRs), the rest an asm form with VGPR accumulators.
#define MFMA_BF16_V(ACC, A, B) asm("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(ACC) : "v"(A), "v"(B))
template <int NRB>
__device__ __forceinline__ void mfma_tile(v16f& acc, int rb, const v8bf& a, const v8bf& b) {
    constexpr int NA = NRB * 2 <= 16 ? NRB : 8;          // row blocks whose two tiles sit in AGPRs (16 tiles = 256 registers)
    if (rb < NA) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
    else MFMA_BF16_V(acc, a, b);
}

// the reference's pipeline for one value (clamped form, linear.py:829-833): bin in [0, levels2] (levels2 = masked)
__device__ __forceinline__ float exact_kc(float xs, float s, float qf, int levels2) {
    float ue = xs / s;
    ue = fminf(fmaxf(ue, 1e-15f), 1.0f);
    const float le = (float)log2((double)ue);            // correctly rounded log2 (common.h: adalog_k)
    const float t = (-le) * 37.0f / qf;
    const float kk = rintf(t);
    return (kk == kk) ? fminf(fmaxf(kk, 0.0f), (float)levels2) : (float)levels2;
}

// thr[b][p] = smallest positive float xs whose bin is <= b (the bin never increases with xs); -inf when even the clamped
// end of the range (u = 1e-15) stays at or below b.  One thread per (bin boundary, candidate), bisection over float bits.
__global__ __launch_bounds__(128) void k_tie_thresholds(const float* __restrict__ scale, const float* __restrict__ qv,
                                                        int levels2, float* __restrict__ thr) {
    const int b = blockIdx.x, pc = threadIdx.x;
    const float s = scale[pc], qf = qv[pc];
    unsigned lo = 0x00800000u, hi = 0x7F7FFFFFu;         // FLT_MIN .. FLT_MAX: bin(lo) is the largest, bin(hi) = 0
    float out;
    if (exact_kc(__uint_as_float(lo), s, qf, levels2) <= (float)b) {
        out = -__builtin_inff();
    } else {
        while (hi - lo > 1u) {
            const unsigned mid = lo + ((hi - lo) >> 1);
            if (exact_kc(__uint_as_float(mid), s, qf, levels2) <= (float)b) hi = mid; else lo = mid;
        }
        out = __uint_as_float(hi);
    }
    thr[b * 128 + pc] = out;
}

template <int NRB, int FNS>
__global__ __launch_bounds__(256, 1) void k_act_fused(FusedArgs p) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int AT = NRB * 2048;               // weight tile bytes per stage (32 * NRB rows x 64 B)
    constexpr int RQ = NRB / 2;                  // weight DMA requests per wave per K-step (16 rows x 64 B each)
    constexpr int RW = RQ + 2;                   // + the wave's x and log2 requests
    constexpr int ROWS = 32 * NRB;
    constexpr int XS = FNS + 1;                  // slots of the x / log2 ring (read one step earlier AND during the step)
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    uint8_t* ring = smem;                                                   // FNS * AT
    uint8_t* xring = ring + FNS * AT;                                       // XS * 4 waves * {256 B x, 256 B log2}
    uint32_t* s_lut = reinterpret_cast<uint32_t*>(xring + XS * 2048);       // [levels2 + 2][128] bf16 bits (low half)
    const int lut_rows = p.levels2 + 2;
    float* s_thr = reinterpret_cast<float*>(s_lut + lut_rows * 128);        // [levels2][128]
    float4* s_par = reinterpret_cast<float4*>(s_thr + p.levels2 * 128);     // [128] {-37/q, log2(s)*37/q, hi, s * sa_mul}
    float* s_refb = reinterpret_cast<float*>(s_par + 128);                  // [2][ROWS] ref - row_bias
    float* s_rs = s_refb + 2 * ROWS;                                        // [ROWS]    row scale (0 past M)
    double* s_fin = reinterpret_cast<double*>(s_rs + ROWS);                 // [4][64]

    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int frow = lane & 31, fkg = lane >> 5;
    const int wtok = w >> 1, cbp = w & 1;

    // ---- per-launch tables
    for (int e = tid; e < lut_rows * 128; e += 256) {
        const int bin = e >> 7, c = e & 127;
        const int kq = bin * (int)p.qv[c];
        const int t = kq / ADALOG_R, j = kq - t * ADALOG_R;
        const float v = (bin >= p.levels2 || t > 100) ? 0.0f : ldexpf(p.mant[j], -t);
        s_lut[e] = __float_as_uint(v) >> 16;                                // exact: <= 8 significant bits
    }
    for (int e = tid; e < p.levels2 * 128; e += 256) s_thr[e] = p.thr[e];
    const float top = (float)p.levels2 + 0.75f;                             // rounds to 2L + 1: a zero entry
    if (tid < 128) {
        const float s = p.scale[tid], qf = p.qv[tid];
        const float rq37 = 37.0f / qf;
        const float NL15 = 49.828921f;                                      // -fl32(log2(1e-15f))
        s_par[tid] = make_float4(-rq37, __log2f(s) * rq37, fminf(NL15 * rq37, top), s * p.sa_mul);
    }
    __syncthreads();
    float ca[2], cc[2], chi[2], calpha[2];
    unsigned lutc[2], thrc[2];
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) {
        const int c = 64 * cbp + 32 * cb + frow;
        const float4 pr = s_par[c];
        ca[cb] = pr.x; cc[cb] = pr.y; chi[cb] = pr.z; calpha[cb] = pr.w;
        // byte address of LUT[bin][c] = lut_base + bin * 512 + c * 4, with bin = bits(t) - MAGIC_BITS folded in (mod 2^32)
        lutc[cb] = (unsigned)(uintptr_t)(lds_u32p)s_lut + (unsigned)c * 4u - (MAGIC_BITS << 9);
        thrc[cb] = (unsigned)(uintptr_t)(lds_f32p)s_thr + (unsigned)c * 4u;
    }

    // ---- tiles: (token pair, row tile), row tile fastest; static stride over the persistent workgroups
    const unsigned npair = (unsigned)(p.T + 1) >> 1;
    const unsigned ntile = npair * (unsigned)p.n_rt;
    const unsigned nwg = gridDim.x, bid = blockIdx.x;
    const int nk = p.nk;
    if (bid >= ntile) return;

    // ---- issue cursors.  Weights run FNS - 1 steps ahead of the compute cursor; the x / log2 runs of step n are first
    // consumed half a step earlier than its weights (the B fragments of a K half are produced while the previous half
    // multiplies), so their cursor runs one step further ahead and their requests travel with the weights of step n - 1.
    // A cursor is (token pair, row tile, K-step); all of it is wave-uniform and advances without branches.
    const int lrow = lane >> 2, lslot16 = ((lane & 3) ^ ((lane >> 4) & 3)) << 4;
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)p.W, 0, p.M * p.Kb, 0x00020000);
    const unsigned d_pair = nwg / (unsigned)p.n_rt, d_rt = nwg - d_pair * (unsigned)p.n_rt;
    struct Cur { unsigned tile, pair, rt; int k; };
    auto cur_init = [&]() { Cur c; c.tile = bid; c.pair = bid / (unsigned)p.n_rt; c.rt = bid - c.pair * (unsigned)p.n_rt; c.k = 0; return c; };
    auto cur_step = [&](Cur& c) __attribute__((always_inline)) {
        const int k1 = c.k + 1;
        const bool wrap = k1 == nk;
        const bool adv = wrap && (c.tile + nwg < ntile);       // past the last tile: stay (harmless re-fetch)
        c.k = wrap ? 0 : k1;
        const unsigned rt1 = c.rt + d_rt, carry = rt1 >= (unsigned)p.n_rt ? 1u : 0u;
        c.tile = adv ? c.tile + nwg : c.tile;
        c.pair = adv ? c.pair + d_pair + carry : c.pair;
        c.rt = adv ? rt1 - carry * (unsigned)p.n_rt : c.rt;
    };
    Cur ca_ = cur_init(), cl_ = cur_init();
    const int vl = (wtok * p.K + lane) * 4;          // the wave's token: 64 floats from the step's k0 (32 are used)
    auto issue_x = [&](int slot) __attribute__((always_inline)) {
        const int tok0 = (int)cl_.pair * 2;
        const int nrec = min(2, p.T - tok0) * p.K * 4;
        const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)(p.x + (int64_t)tok0 * p.K), 0, nrec, 0x00020000);
        const __amdgpu_buffer_rsrc_t rl = __builtin_amdgcn_make_buffer_rsrc((void*)(p.L + (int64_t)tok0 * p.K), 0, nrec, 0x00020000);
        uint8_t* dst = xring + slot * 2048 + w * 512;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (las_ptr)dst, 4, vl, cl_.k * 128, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rl, (las_ptr)(dst + 256), 4, vl, cl_.k * 128, 0, 0);
        cur_step(cl_);
    };
    auto issue_a = [&](int slot) __attribute__((always_inline)) {
        uint8_t* st_ = ring + slot * AT;
        const int row0 = (int)ca_.rt * ROWS + w * 16 + lrow;
#pragma unroll
        for (int q = 0; q < RQ; ++q) {
            const int row = min(row0 + q * 64, p.M - 1);       // past M: re-read the last row (masked by s_rs = 0)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (las_ptr)(st_ + (w + 4 * q) * 1024), 16, row * p.Kb + lslot16, ca_.k * 64, 0, 0);
        }
        cur_step(ca_);
    };
    issue_x(0);                                      // x / log2 runs of global step 0
#pragma unroll
    for (int s0 = 0; s0 < FNS - 1; ++s0) { issue_a(s0); issue_x(s0 + 1); }

    v16f acc[NRB][2];
    uint32_t bA[2][4], bB[2][4];                     // B fragments of one K half [candidate block][dword]
    double run[2] = {0.0, 0.0};
    int st = 0, sx = 0;                              // ring slots of the current global step (weights; x / log2)

    // ---- B fragments of one (candidate block, K half) chunk: 8 element-candidates of this lane.
    // Returns max |kf - rne(kf)| over the chunk (the tie detector).
    auto gen_chunk = [&](const float4& l0, const float4& l1, int cb, uint32_t (&out)[4]) __attribute__((always_inline)) -> float {
        const float lv[8] = {l0.x, l0.y, l0.z, l0.w, l1.x, l1.y, l1.z, l1.w};
        uint32_t v[8];
        float dm = 0.0f;
#pragma unroll
        for (int e = 0; e < 8; e += 2) {
            const float k0 = __builtin_amdgcn_fmed3f(__builtin_fmaf(lv[e], ca[cb], cc[cb]), 0.0f, chi[cb]);
            const float k1 = __builtin_amdgcn_fmed3f(__builtin_fmaf(lv[e + 1], ca[cb], cc[cb]), 0.0f, chi[cb]);
            const float t0 = k0 + MAGIC, t1 = k1 + MAGIC;
            const float d0 = k0 - (t0 - MAGIC), d1 = k1 - (t1 - MAGIC);
            dm = fmaxf(fmaxf(dm, fabsf(d0)), fabsf(d1));                     // v_max3_f32 with |.| modifiers
            v[e] = *(lds_u32p)(uintptr_t)((__float_as_uint(t0) << 9) + lutc[cb]);
            v[e + 1] = *(lds_u32p)(uintptr_t)((__float_as_uint(t1) << 9) + lutc[cb]);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) out[i] = v[2 * i] | (v[2 * i + 1] << 16);
        return dm;
    };

    // ---- a chunk with a near-tie (cold): every bin of the chunk from the threshold table.  The fast bin f is within one of
    // the exact one, which is  f + [xs < thr[f]] - [xs >= thr[f - 1]]  (bins count the break points above xs).
    auto fix_chunk = [&](const float4& l0, const float4& l1, const float4& x0, const float4& x1, int cb, uint32_t (&out)[4])
        __attribute__((always_inline)) {
        const float lv[8] = {l0.x, l0.y, l0.z, l0.w, l1.x, l1.y, l1.z, l1.w};
        const float xv[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
        uint32_t v[8];
        const int L2 = p.levels2;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float kf = __builtin_amdgcn_fmed3f(__builtin_fmaf(lv[e], ca[cb], cc[cb]), 0.0f, chi[cb]);
            const int f = min((int)(__float_as_uint(kf + MAGIC) & 0xFFu), L2);
            const float xs = xv[e] + p.shift;
            const float tu = *(lds_f32p)(uintptr_t)(thrc[cb] + (unsigned)min(f, L2 - 1) * 512u);
            const float td = *(lds_f32p)(uintptr_t)(thrc[cb] + (unsigned)max(f - 1, 0) * 512u);
            const int up = (f < L2 && xs < tu) ? 1 : 0, dn = (f > 0 && !(xs < td)) ? 1 : 0;
            v[e] = s_lut[(f + up - dn) * 128 + 64 * cbp + 32 * cb + frow];
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) out[i] = v[2 * i] | (v[2 * i + 1] << 16);
    };

#if defined(FUSED_IGLP)
#define FUSED_SCHED_HINT __builtin_amdgcn_iglp_opt(FUSED_IGLP);
#else
#define FUSED_SCHED_HINT
#endif
    // ---- half a K-step: 2 groups, each = one chunk of the NEXT half's B fragments (BN) + NRB MFMAs of this half (BC);
    // a scheduling barrier per group keeps the fragment look-ahead (and the register pressure) to one group.
    // XN / LN: the x / log2 runs the next half reads (8 floats per lane at element offset EO)
#define FUSED_HALF(H, BC, BN, XN, EO)                                                                             \
    do {                                                                                                          \
        const float4 l0_ = lds_f4((XN), 256 + ((EO) + 8 * fkg) * 4), l1_ = lds_f4((XN), 256 + ((EO) + 8 * fkg) * 4 + 16); \
        _Pragma("unroll") for (int cbg_ = 0; cbg_ < 2; ++cbg_) {                                                  \
            const float dm_ = gen_chunk(l0_, l1_, cbg_, BN[cbg_]);                                                \
            _Pragma("unroll") for (int r2 = 0; r2 < NRB / 2; ++r2) {                                              \
                const int rb = cbg_ * (NRB / 2) + r2;                                                             \
                const uint4 a_ = lds_frag(As_, (rb * 32 + frow) * 64 + (((2 * (H) + fkg) ^ ((frow >> 2) & 3)) << 4)); \
                mfma_tile<NRB>(acc[rb][0], rb, __builtin_bit_cast(v8bf, a_), frag8(BC[0]));                       \
                mfma_tile<NRB>(acc[rb][1], rb, __builtin_bit_cast(v8bf, a_), frag8(BC[1]));                       \
            }                                                                                                     \
            FUSED_SCHED_HINT                                                                                      \
            __builtin_amdgcn_sched_barrier(0);                                                                    \
            if (__builtin_expect(__any(dm_ > TIE), 0)) {                                                          \
                const float4 x0_ = lds_f4((XN), ((EO) + 8 * fkg) * 4), x1_ = lds_f4((XN), ((EO) + 8 * fkg) * 4 + 16); \
                fix_chunk(l0_, l1_, x0_, x1_, cbg_, BN[cbg_]);                                                    \
            }                                                                                                     \
        }                                                                                                         \
    } while (0)

    // B fragments of the very first half (the runs of global step 0 were issued first)
    {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const uint8_t* X0 = xring + w * 512;
        const float4 l0_ = lds_f4(X0, 256 + 8 * fkg * 4), l1_ = lds_f4(X0, 256 + 8 * fkg * 4 + 16);
        const float4 x0_ = lds_f4(X0, 8 * fkg * 4), x1_ = lds_f4(X0, 8 * fkg * 4 + 16);
#pragma unroll
        for (int cbg_ = 0; cbg_ < 2; ++cbg_) {
            const float dm_ = gen_chunk(l0_, l1_, cbg_, bA[cbg_]);
            if (__any(dm_ > TIE)) fix_chunk(l0_, l1_, x0_, x1_, cbg_, bA[cbg_]);
        }
    }

    for (unsigned tile = bid; tile < ntile; tile += nwg) {
        const unsigned pair = tile / (unsigned)p.n_rt;
        const int m0 = (int)(tile - pair * (unsigned)p.n_rt) * ROWS;
        const int tok0 = (int)pair * 2;
        const bool tok_ok = tok0 + wtok < p.T;
#pragma unroll
        for (int rb = 0; rb < NRB; ++rb)
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc[rb][0][r] = 0.0f; acc[rb][1][r] = 0.0f; }
        // epilogue operands of this tile: loaded and staged here (the barrier orders the stores behind the previous
        // tile's epilogue reads of every wave; the vmcnt(0) costs one ring refill per tile of nk steps)
        {
            constexpr int EU = (2 * ROWS + 255) / 256, RU = (ROWS + 255) / 256;
            float e_ref[EU], e_rs[RU];
#pragma unroll
            for (int u = 0; u < EU; ++u) {
                const int e = tid + u * 256, ts_ = e >= ROWS ? 1 : 0, r = e - ts_ * ROWS;
                const int row = m0 + r, tk = tok0 + ts_;
                const bool ok = e < 2 * ROWS && row < p.M && tk < p.T;
                e_ref[u] = ok ? p.ref[(int64_t)tk * p.M + row] - (p.row_bias ? p.row_bias[row] : 0.0f) : 0.0f;
            }
#pragma unroll
            for (int u = 0; u < RU; ++u) {
                const int r = tid + u * 256, row = m0 + r;
                e_rs[u] = (r < ROWS && row < p.M) ? p.row_scale[row] : 0.0f;
            }
            __builtin_amdgcn_s_barrier();
#pragma unroll
            for (int u = 0; u < EU; ++u) if (tid + u * 256 < 2 * ROWS) s_refb[tid + u * 256] = e_ref[u];
#pragma unroll
            for (int u = 0; u < RU; ++u) if (tid + u * 256 < ROWS) s_rs[tid + u * 256] = e_rs[u];
        }
        for (int kt = 0; kt < nk; ++kt) {
            // this step's weights (and the x / log2 runs of the next) have landed once only the newest FNS - 2 batches of
            // this wave are outstanding
            if ((FNS - 2) * RW == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
            else if ((FNS - 2) * RW == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
            else if ((FNS - 2) * RW == 10) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
            else if ((FNS - 2) * RW == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else if ((FNS - 2) * RW == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            else if ((FNS - 2) * RW == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
            else if ((FNS - 2) * RW == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            const uint8_t* As_ = ring + st * AT;
            const int sxn = sx == XS - 1 ? 0 : sx + 1;
            const uint8_t* Xc_ = xring + sx * 2048 + w * 512;       // runs of this step (second half still to be produced)
            const uint8_t* Xn_ = xring + sxn * 2048 + w * 512;      // runs of the next step
            issue_a(st == 0 ? FNS - 1 : st - 1);
            issue_x(sx == 0 ? XS - 1 : sx - 1);                     // step n + FNS -> the slot step n - 1 has left
            FUSED_HALF(0, bA, bB, Xc_, 16);
            FUSED_HALF(1, bB, bA, Xn_, 0);
            st = st == FNS - 1 ? 0 : st + 1;
            sx = sxn;
        }
        // ---- epilogue (one row block at a time: the scheduling barrier keeps the staged reads from being hoisted)
        float s0 = 0.0f, s1 = 0.0f;
        const uint8_t* rb_ = reinterpret_cast<const uint8_t*>(s_refb + wtok * ROWS);
        const uint8_t* rs_ = reinterpret_cast<const uint8_t*>(s_rs);
#pragma unroll
        for (int rb = 0; rb < NRB; ++rb) {
#pragma unroll
            for (int i4 = 0; i4 < 4; ++i4) {
                const int r = rb * 32 + 8 * i4 + 4 * fkg;
                const float4 rf = lds_f4(rb_, r * 4), rs = lds_f4(rs_, r * 4);
                const float rfa[4] = {rf.x, rf.y, rf.z, rf.w}, rsa[4] = {rs.x, rs.y, rs.z, rs.w};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float e0 = __builtin_fmaf(-(acc[rb][0][4 * i4 + j] * calpha[0]), rsa[j], rfa[j]);
                    const float e1 = __builtin_fmaf(-(acc[rb][1][4 * i4 + j] * calpha[1]), rsa[j], rfa[j]);
                    s0 = __builtin_fmaf(e0, e0, s0);
                    s1 = __builtin_fmaf(e1, e1, s1);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        s0 += __shfl_xor(s0, 32);
        s1 += __shfl_xor(s1, 32);
        if (tok_ok) { run[0] += (double)s0; run[1] += (double)s1; }
    }
#undef FUSED_HALF
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // drain the run-ahead requests before the LDS is released
    if (fkg == 0) { s_fin[w * 64 + frow] = run[0]; s_fin[w * 64 + 32 + frow] = run[1]; }
    __syncthreads();
    if (tid < 128) {
        const int cw = tid >> 6, wi = tid & 63;
        p.wg_acc[(int64_t)bid * 128 + tid] = s_fin[cw * 64 + wi] + s_fin[(2 + cw) * 64 + wi];
    }
#endif
}

// ---------------------------------------------------------------------------------------------- hand-scheduled form
// Same algorithm, same LDS tables, same results as k_act_fused<NRB, FNS>; the persistent loop (DMA ring, fragment
// generation, MFMA stream, threshold fix-ups, epilogue) is the generated asm of tools/gen_fused_asm.py
// (fused_loop_nrb*.inc): EIGHT waves per workgroup, two per SIMD (256 registers each: accumulator tiles in a0..a127
// and v64..v127, 64 working VGPRs), wave = (token of the pair, candidate block of 32); see the generator's header for the machine model behind
// that shape.  The HIP part builds the tables, passes the scalars through an LDS config array (the block clobbers
// v0..v127, a0..a127 and s8..s99) and writes the workgroup's row of sums.
#define FUSED_ASM_CLOBBERS \
    "v0", "v1", "v2", "v3", "v4", "v5", "v6", "v7", "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15", "v16", \
    "v17", "v18", "v19", "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32", \
    "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", \
    "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63", "v64", \
    "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73", "v74", "v75", "v76", "v77", "v78", "v79", "v80", \
    "v81", "v82", "v83", "v84", "v85", "v86", "v87", "v88", "v89", "v90", "v91", "v92", "v93", "v94", "v95", "v96", \
    "v97", "v98", "v99", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", \
    "v111", "v112", "v113", "v114", "v115", "v116", "v117", "v118", "v119", "v120", "v121", "v122", "v123", "v124", \
    "v125", "v126", "v127", "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", \
    "a14", "a15", "a16", "a17", "a18", "a19", "a20", "a21", "a22", "a23", "a24", "a25", "a26", "a27", "a28", "a29", \
    "a30", "a31", "a32", "a33", "a34", "a35", "a36", "a37", "a38", "a39", "a40", "a41", "a42", "a43", "a44", "a45", \
    "a46", "a47", "a48", "a49", "a50", "a51", "a52", "a53", "a54", "a55", "a56", "a57", "a58", "a59", "a60", "a61", \
    "a62", "a63", "a64", "a65", "a66", "a67", "a68", "a69", "a70", "a71", "a72", "a73", "a74", "a75", "a76", "a77", \
    "a78", "a79", "a80", "a81", "a82", "a83", "a84", "a85", "a86", "a87", "a88", "a89", "a90", "a91", "a92", "a93", \
    "a94", "a95", "a96", "a97", "a98", "a99", "a100", "a101", "a102", "a103", "a104", "a105", "a106", "a107", \
    "a108", "a109", "a110", "a111", "a112", "a113", "a114", "a115", "a116", "a117", "a118", "a119", "a120", "a121", \
    "a122", "a123", "a124", "a125", "a126", "a127", "s8", "s9", "s10", "s11", "s12", "s13", "s14", "s15", "s16", \
    "s17", "s18", "s19", "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27", "s28", "s29", "s30", "s31", "s34", \
    "s35", "s36", "s37", "s39", "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s48", "s49", "s50", "s51", \
    "s52", "s53", "s54", "s55", "s56", "s57", "s58", "s59", "s60", "s61", "s62", "s63", "s64", "s65", "s66", "s67", \
    "s68", "s69", "s70", "s71", "s72", "s73", "s74", "s75", "s76", "s77", "s78", "s79", "s80", "s81", "s82", "s83", \
    "s84", "s85", "s86", "s87", "s88", "s89", "s90", "s91", "s92", "s93", "s94", "s95", "s96", "s97", "s98", "s99", \
    "vcc", "scc", "m0", "memory"

#define KNAME k_act_fused_asm_12_4
#define KNRB 12
#define KFNS 4
#define KINC "fused_loop_nrb12_s4.inc"
#include "fused_asm_kernel.inc"
#undef KNAME
#undef KNRB
#undef KFNS
#undef KINC
#define KNAME k_act_fused_asm_12_3
#define KNRB 12
#define KFNS 3
#define KINC "fused_loop_nrb12_s3.inc"
#include "fused_asm_kernel.inc"
#undef KNAME
#undef KNRB
#undef KFNS
#undef KINC
#define KNAME k_act_fused_asm_8_4
#define KNRB 8
#define KFNS 4
#define KINC "fused_loop_nrb8_s4.inc"
#include "fused_asm_kernel.inc"
#undef KNAME
#undef KNRB
#undef KFNS
#undef KINC
#define KNAME k_act_fused_asm_4_4
#define KNRB 4
#define KFNS 4
#define KINC "fused_loop_nrb4_s4.inc"
#include "fused_asm_kernel.inc"
#undef KNAME
#undef KNRB
#undef KFNS
#undef KINC

// L = log2(x + shift), correctly rounded (fp64 log2 rounded once): -inf where x + shift <= 0 (the u-clamp / the bin mask
// then take over, exactly as for log2 of a non-positive number in k_pack_adalog_fast)
__global__ __launch_bounds__(256) void k_log2_shift(const float* __restrict__ x, float* __restrict__ out, int64_t n, float shift) {
    const int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i >= n) return;
    if (i + 4 <= n) {
        const float4 v = *reinterpret_cast<const float4*>(x + i);
        const float a[4] = {v.x + shift, v.y + shift, v.z + shift, v.w + shift};
        float r[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) r[e] = a[e] > 0.0f ? (float)log2((double)a[e]) : -__builtin_inff();
        *reinterpret_cast<float4*>(out + i) = make_float4(r[0], r[1], r[2], r[3]);
    } else {
        for (int64_t j = i; j < n; ++j) {
            const float a = x[j] + shift;
            out[j] = a > 0.0f ? (float)log2((double)a) : -__builtin_inff();
        }
    }
}

// scores[c] = -norm * sum over workgroups of acc[wg][c], fixed order (lane-strided, then a shuffle tree)
__global__ __launch_bounds__(64) void k_fused_finish(const double* __restrict__ acc, int nwg, double norm, float* __restrict__ scores) {
    const int c = blockIdx.x, lane = threadIdx.x;
    double s = 0.0;
    for (int i = lane; i < nwg; i += 64) s += acc[(int64_t)i * 128 + c];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (lane == 0) scores[c] = (float)(-norm * s);
}

static int fused_cus() {
    static int n_cu = 0;
    if (!n_cu) {
        int dev = 0; hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) n_cu = prop.multiProcessorCount;
        if (n_cu <= 0) n_cu = 256;
    }
    return n_cu;
}

static bool fused_use_asm() {
    static const int use_asm = getenv("ADALOG_FUSED_ASM") ? atoi(getenv("ADALOG_FUSED_ASM")) : 1;
    return use_asm != 0;
}

// row blocks per tile: least padding of M; the hand-scheduled loops exist for 12, 8 and 4 (its eight waves split the
// weight requests four ways), the compiler form also for 6
static int pick_nrb(int M) {
    const int cand[4] = {12, 8, 6, 4};
    int best = 12;
    int64_t best_pad = -1;
    for (int i = 0; i < 4; ++i) {
        if (cand[i] == 6 && fused_use_asm()) continue;
        const int rows = 32 * cand[i];
        const int64_t pad = (int64_t)((M + rows - 1) / rows) * rows;
        if (best_pad < 0 || pad < best_pad) { best_pad = pad; best = cand[i]; }
    }
    return best;
}

static size_t fused_lds_bytes(int nrb, int fns, int levels2) {
    return (size_t)fns * nrb * 2048 + (size_t)(fns + 1) * 2048 + (size_t)(levels2 + 2) * 512 + (size_t)levels2 * 512 + 128 * 16 +
           (size_t)3 * 32 * nrb * 4 + 256 * 8;
}
static int pick_fns(int nrb, int levels2) {
    return fused_lds_bytes(nrb, 4, levels2) <= 160 * 1024 ? 4 : 3;
}

}  // namespace

extern "C" int adalog_log2_shift(const float* x, float* out, int64_t n, float shift, void* stream) {
    ADALOG_ARG_CHECK(x && out && n >= 1, "log2_shift: bad arguments");
    ADALOG_ARG_CHECK((((uintptr_t)x | (uintptr_t)out) & 15) == 0, "log2_shift: pointers must be 16-byte aligned");
    hipLaunchKernelGGL(k_log2_shift, dim3((unsigned)((n + 1023) / 1024)), dim3(256), 0, (hipStream_t)stream, x, out, n, shift);
    ADALOG_LAUNCH_CHECK("adalog_log2_shift");
    return 0;
}

// 1 when adalog_score_act_fused takes this shape (else the caller packs the candidate operand and uses adalog_gemm_score)
extern "C" int adalog_score_act_fused_ok(int M, int64_t T, int K, int64_t Kp, int P, int n_bits) {
    if (P != 128 || n_bits < 2 || n_bits > 6 || M < 1 || T < 1 || K < 1) return 0;
    if (Kp < K || (Kp * 2) % 64 != 0 || Kp / 32 < 6) return 0;                       // >= 6 K-steps of 32 elements
    if ((int64_t)M * Kp * 2 >= ((int64_t)1 << 31) || T >= ((int64_t)1 << 30)) return 0;
    const int nrb = pick_nrb(M);
    // the B fragments are regenerated per row tile: beyond two tiles (M > 768) the generation outweighs the operand
    // traffic it saves (measured: swin_base stage 3, 4096 -> 1024, is 13 % faster on the packed path)
    static const int max_rt = getenv("ADALOG_FUSED_MAX_RT") ? atoi(getenv("ADALOG_FUSED_MAX_RT")) : 2;
    if ((M + 32 * nrb - 1) / (32 * nrb) > max_rt) return 0;
    return fused_lds_bytes(nrb, pick_fns(nrb, 1 << n_bits), 1 << n_bits) <= 160 * 1024 ? 1 : 0;
}

// [workgroups][128] fp64 partial sums, then the [2^6][128] fp32 threshold table (T, Kp: the call's shape; reserved)
extern "C" int64_t adalog_score_act_fused_workspace_bytes(int64_t T, int64_t Kp) {
    (void)T; (void)Kp;
    return (int64_t)fused_cus() * 128 * 8 + 64 * 128 * 4;
}

// scores[p] = -norm * sum_{t, o} ( (ref[t, o] - row_bias[o]) - s_w[o] * (s_p * sa_mul) * sum_k Wq[o, k] * m_p(x[t, k]) )^2
// for the 128 AdaLog candidates (s_p, q_p); m_p = integer-numerator form of the search-time AdaLog value
// (linear.py:831-836): 2^-floor(k q / 37) * mant37[(k q) mod 37], k = rne(-log2(clamp((x + shift) / s_p)) * 37 / q_p).
// Wp: bf16 image of the quantised weight [M][Kp] (adalog_pack_uniform, bf16);  Lx = adalog_log2_shift(x, shift).
extern "C" int adalog_score_act_fused(const void* Wp, int M, int64_t Kp, const float* x, const float* Lx, int64_t T, int K,
                                      const float* ref, const float* row_scale, const float* row_bias, const float* scale,
                                      const float* qv, int P, int n_bits, const float* mant37, float shift, int clamp_u,
                                      float sa_mul, double norm, void* workspace, int64_t workspace_bytes, float* scores,
                                      void* stream) {
    ADALOG_ARG_CHECK(Wp && x && Lx && ref && row_scale && scale && qv && mant37 && workspace && scores, "score_act_fused: null pointer");
    ADALOG_ARG_CHECK(adalog_score_act_fused_ok(M, T, K, Kp, P, n_bits), "score_act_fused: shape not supported (ask adalog_score_act_fused_ok first)");
    ADALOG_ARG_CHECK(clamp_u, "score_act_fused: only the clamped form (post-GELU searches, linear.py:829) is implemented");
    ADALOG_ARG_CHECK(workspace_bytes >= adalog_score_act_fused_workspace_bytes(T, Kp) && ((uintptr_t)workspace & 7) == 0,
                     "score_act_fused: workspace too small or misaligned");
    FusedArgs a{};
    a.W = (const uint8_t*)Wp; a.L = Lx; a.x = x; a.ref = ref; a.row_scale = row_scale; a.row_bias = row_bias;
    a.scale = scale; a.qv = qv; a.mant = mant37; a.wg_acc = (double*)workspace;
    a.M = M; a.T = (int)T; a.K = K; a.Kb = (int)(Kp * 2); a.levels2 = 1 << n_bits;
    a.nk = (int)(Kp * 2 / 64); a.shift = shift; a.sa_mul = sa_mul;
    float* thr = reinterpret_cast<float*>((uint8_t*)workspace + (size_t)fused_cus() * 128 * 8);
    a.thr = thr;
    const int nrb = pick_nrb(M);
    const int fns = pick_fns(nrb, a.levels2);
    a.n_rt = (M + 32 * nrb - 1) / (32 * nrb);
    const int64_t ntile = ((T + 1) / 2) * a.n_rt;
    const int nwg = (int)(ntile < fused_cus() ? ntile : fused_cus());
    const size_t shm = fused_lds_bytes(nrb, fns, a.levels2);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(k_tie_thresholds, dim3((unsigned)a.levels2), dim3(128), 0, st, scale, qv, a.levels2, thr);
    ADALOG_LAUNCH_CHECK("adalog_score_act_fused (thresholds)");
#define LAUNCH_FUSED(NRBV, FNSV)                                                                               \
    do {                                                                                                       \
        static unsigned long long attr_dev = 0; \
        { hipError_t ea__ = adalog_max_lds(reinterpret_cast<const void*>(&k_act_fused<NRBV, FNSV>), (int)(160 * 1024), &attr_dev); \
          if (ea__ != hipSuccess) { adalog_set_error("hipFuncSetAttribute", ea__); return (int)ea__; } } \
        adalog_note_kernel("k_act_fused<bf16>");                                                             \
        hipLaunchKernelGGL((k_act_fused<NRBV, FNSV>), dim3((unsigned)nwg), dim3(256), shm, st, a);             \
    } while (0)
#define LAUNCH_FUSED_N(NRBV) do { if (fns == 4) LAUNCH_FUSED(NRBV, 4); else LAUNCH_FUSED(NRBV, 3); } while (0)
    const bool use_asm = fused_use_asm();
    const size_t shm_asm = 256 + shm;
#define LAUNCH_ASM(KERNEL, TAG)                                                                                 \
    do {                                                                                                       \
        static unsigned long long attr_dev = 0; \
        { hipError_t ea__ = adalog_max_lds(reinterpret_cast<const void*>(&KERNEL), (int)(160 * 1024), &attr_dev); \
          if (ea__ != hipSuccess) { adalog_set_error("hipFuncSetAttribute", ea__); return (int)ea__; } } \
        adalog_note_kernel(TAG);                                                                               \
        hipLaunchKernelGGL(KERNEL, dim3((unsigned)nwg), dim3(512), shm_asm, st, a);                            \
    } while (0)
    if (use_asm && shm_asm <= 160 * 1024) {
        if (nrb == 12 && fns == 4) LAUNCH_ASM(k_act_fused_asm_12_4, "k_act_fused_asm<12,4,bf16>");
        else if (nrb == 12) LAUNCH_ASM(k_act_fused_asm_12_3, "k_act_fused_asm<12,3,bf16>");
        else if (nrb == 8) LAUNCH_ASM(k_act_fused_asm_8_4, "k_act_fused_asm<8,4,bf16>");
        else LAUNCH_ASM(k_act_fused_asm_4_4, "k_act_fused_asm<4,4,bf16>");
    } else if (nrb == 12) LAUNCH_FUSED_N(12);
    else if (nrb == 8) LAUNCH_FUSED_N(8);
    else if (nrb == 6) LAUNCH_FUSED_N(6);
    else LAUNCH_FUSED_N(4);
#undef LAUNCH_ASM
#undef LAUNCH_FUSED_N
#undef LAUNCH_FUSED
    ADALOG_LAUNCH_CHECK("adalog_score_act_fused");
    hipLaunchKernelGGL(k_fused_finish, dim3(128), dim3(64), 0, st, (const double*)workspace, nwg, norm, scores);
    ADALOG_LAUNCH_CHECK("adalog_score_act_fused (finish)");
    return 0;
}
