// K11 -- post-GELU activation-candidate search with the AdaLog quantisation FUSED INTO THE GEMM'S LOADER.
//
// Replaces, for one scoring call of reference quant_layers/linear.py:816-848 / :856-890 / :898-931
//   x_sim[N,T,I,P] = AdaLog_p(x + shift) * s_p - shift   (P = 128 candidates (s_p, q_p), ~12 elementwise ATen ops)
//   out_sim        = F.linear(x_sim, q_w(W), b)          ->  -(raw_out - out_sim)^2  ->  mean / sum  ->  scores [P]
// by ONE kernel that never materialises the candidate operand: round 1 wrote it to HBM as bf16 (2.5 GB per call for
// deit_small fc2: k_pack_adalog_fast 0.4 ms) and read it back in the scoring GEMM (0.99 ms, 2.7 GB: 55x the
// algorithmic bytes).  Here the only HBM streams are log2(x + shift) (fp32, 38.7 MB, read once), raw_out (9.7 MB) and the
// bf16 weight image (1.2 MB, L2-resident).
//
// Evaluated transposed, like the streaming kernel's activation searches:  D[o, (t, p)] = sum_k Wq[o, k] * v_p(x[t, k]),
// GEMM rows = output channels, GEMM columns = (token, candidate).  One workgroup = 4 waves, ONE PER SIMD (512 registers
// each), tile = (32 * NRB) rows x 2 tokens x 128 candidates:
//   * wave w owns token (w >> 1) and candidate blocks 2 * (w & 1), 2 * (w & 1) + 1: NRB x 2 accumulator tiles of 32 x 32
//     (NRB = 12: 384 registers).  A lane IS a candidate column: its (37/q_p, log2(s_p) * 37/q_p, clamp) live in registers;
//   * the B fragments (candidate operand) are produced in registers, straight into MFMA operand layout: a lane needs 8
//     consecutive k of its (token, candidate) per v_mfma_f32_32x32x16_bf16, i.e. per element-candidate
//         kf = med3(fma(L, -37/q, c), lo, hi);  t = kf + 1.5*2^23;  d = kf - (t - 1.5*2^23);
//         value = LUT[(bits(t) << 9) + lane const]   (ds_read_b32: dword table [bin][candidate], bank = candidate: no
//         conflicts);  two values pack into one dword;  max3 over |d| flags near-ties          -- ~7 VALU + 1 LDS read;
//     L = log2(x + shift) is precomputed once per layer (adalog_log2_shift, correctly rounded) and reaches the wave through
//     LDS (one 256-byte DMA per K-step and wave, broadcast reads): no transcendental runs in this kernel;
//   * EXACT bins, deferred: the fast kf is within ~2e-5 of the reference's fp32 pipeline  x/s -> clamp -> log2 -> *37 -> /q
//     (error budget in DESIGN.md).  A lane whose |d| lies within 1e-4 of a rounding tie pushes (k, candidate, fast bin)
//     into its wave's LDS queue -- ~12 instructions in a cold block.  At the end of the tile the wave resolves the queue
//     64 events at a time: exact pipeline (IEEE divide, correctly rounded log2 through fp64: common.h's rule), and where
//     the exact bin differs from the fast one (~1e-5 of element-candidates) the two lanes that own the candidate's column
//     apply the rank-1 correction  acc[o] += Wq[o, k] * (v_exact - v_fast)  to their accumulators.  The scores are
//     those of the exact bins; nothing heavy sits in the MFMA loop;
//   * the weight tile (32 * NRB rows x 64 bytes per K-step) streams through a 4-stage LDS-DMA ring shared by the four
//     waves (buffer_load ... lds, swizzled source slots, counted vmcnt + one barrier per K-step) -- 16 B/clk/CU at the
//     matrix pipe's full rate; an A fragment read feeds two MFMAs;
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
typedef void __attribute__((address_space(3)))* las_ptr;
typedef const uint32_t __attribute__((address_space(3)))* lds_u32p;

struct FusedArgs {
    const uint8_t* W;          // [M][Kb] bf16 image of q_w(W) - z (integers), rows zero-padded to Kb bytes
    const float* L;            // [T][K]  log2(x + shift)  (-inf where x + shift <= 0)
    const float* x;            // [T][K]  the captured activation (exact path only)
    const float* ref;          // [T][M]  raw_out
    const float* row_scale;    // [M]     weight scale s_w[o]
    const float* row_bias;     // [M]     bias with the -shift term folded in (may be null)
    const float* scale;        // [128]   candidate scales s_p
    const float* qv;           // [128]   candidate bases q_p (as floats)
    const float* mant;         // [37]    integer numerators of the search-time mantissa table (linear.py:750-752)
    double* wg_acc;            // [gridDim.x][128]
    int M, T, K;
    int Kb;                    // row pitch of W in bytes (multiple of 128)
    int levels2;               // 2^bits
    int clamp_u;
    int n_rt;                  // row tiles of 32 * NRB rows
    int nk;                    // 64-byte K-steps (even: Kb is a multiple of 128)
    float shift, sa_mul;
};

constexpr int FNS = 4;                       // ring stages
constexpr int QCAP = 1024;                   // tie-queue entries per wave (a chunk can push 512)
constexpr float MAGIC = 12582912.0f;         // 1.5 * 2^23: fl32(kf + MAGIC) carries rne(kf) in its low mantissa bits
constexpr unsigned MAGIC_BITS = 0x4B400000u;
constexpr float TIE = 0.4999f;               // |kf - rne(kf)| above this: the exact pipeline decides

// Fragment / parameter reads through __restrict__ helpers: the loads carry alias scopes, so the waitcnt pass does not
// order them behind the (untagged) in-flight LDS-DMA with a vmcnt(0) -- the counted vmcnt before each barrier does that.
__device__ __forceinline__ uint4 lds_frag(const uint8_t* __restrict__ stage, int off) {
    return *reinterpret_cast<const uint4*>(stage + off);
}
__device__ __forceinline__ float4 lds_f4(const uint8_t* __restrict__ base, int off) {
    return *reinterpret_cast<const float4*>(base + off);
}

typedef unsigned u4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ v8bf frag8(const uint32_t (&d)[4]) {
    const u4v v = {d[0], d[1], d[2], d[3]};
    return __builtin_bit_cast(v8bf, v);
}

// The compiler selects ONE form of MFMA per function (accumulators in AGPRs when the wave may use 512 registers), so it
// cannot keep more than 256 accumulator registers without copying tiles in and out around every MFMA.  Hand-placed
// classes: the first NA accumulator tiles live in AGPRs ("a"), the rest in VGPRs ("v").
#define MFMA_BF16_A(ACC, A, B) asm("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(ACC) : "v"(A), "v"(B))
#define MFMA_BF16_V(ACC, A, B) asm("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(ACC) : "v"(A), "v"(B))
template <int NRB>
__device__ __forceinline__ void mfma_tile(v16f& acc, int rb, const v8bf& a, const v8bf& b) {
    constexpr int NA = NRB * 2 <= 16 ? NRB : 8;          // row blocks whose two tiles sit in AGPRs (16 tiles = 256 registers)
    if (rb < NA) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);   // compiler form: AGPR accumulators
    else MFMA_BF16_V(acc, a, b);
}

template <int NRB>
__global__ __launch_bounds__(256, 1) void k_act_fused(FusedArgs p) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int AT = NRB * 2048;               // weight tile bytes per stage (32 * NRB rows x 64 B)
    constexpr int STG = AT + 1024;               // + 4 x 256 B of log2 values (one 64-float run per wave)
    constexpr int RQ = NRB / 2;                  // weight DMA requests per wave per K-step (16 rows x 64 B each)
    constexpr int RW = RQ + 1;                   // + the wave's log2 request
    constexpr int ROWS = 32 * NRB;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    uint8_t* ring = smem;                                                   // FNS * STG
    uint32_t* s_lut = reinterpret_cast<uint32_t*>(smem + FNS * STG);        // [levels2 + 2][128] bf16 bits (low half)
    const int lut_rows = p.levels2 + 2;
    float4* s_par = reinterpret_cast<float4*>(s_lut + lut_rows * 128);      // [128] {-37/q, log2(s)*37/q, hi, s * sa_mul}
    float2* s_sq = reinterpret_cast<float2*>(s_par + 128);                  // [128] {s, q}           (exact path)
    float* s_refb = reinterpret_cast<float*>(s_sq + 128);                   // [2][ROWS] ref - row_bias
    float* s_rs = s_refb + 2 * ROWS;                                        // [ROWS]    row scale (0 past M)
    double* s_fin = reinterpret_cast<double*>(s_rs + ROWS);                 // [4][64]
    uint32_t* s_queue = reinterpret_cast<uint32_t*>(s_fin + 256);           // [4][QCAP]

    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int frow = lane & 31, fkg = lane >> 5;
    const int wtok = w >> 1, cbp = w & 1;
    uint32_t* myq = s_queue + w * QCAP;

    // ---- per-launch tables
    for (int e = tid; e < lut_rows * 128; e += 256) {
        const int bin = e >> 7, c = e & 127;
        const int kq = bin * (int)p.qv[c];
        const int t = kq / ADALOG_R, j = kq - t * ADALOG_R;
        const float v = (bin >= p.levels2 || t > 100) ? 0.0f : ldexpf(p.mant[j], -t);
        s_lut[e] = __float_as_uint(v) >> 16;                                // exact: <= 8 significant bits
    }
    const float lo = p.clamp_u ? 0.0f : -0.25f;
    const float top = (float)p.levels2 + 0.75f;                             // rounds to 2L + 1: a zero entry
    if (tid < 128) {
        const float s = p.scale[tid], qf = p.qv[tid];
        const float rq37 = 37.0f / qf;
        const float NL15 = 49.828921f;                                      // -fl32(log2(1e-15f))
        s_par[tid] = make_float4(-rq37, __log2f(s) * rq37, p.clamp_u ? fminf(NL15 * rq37, top) : top, s * p.sa_mul);
        s_sq[tid] = make_float2(s, qf);
    }
    __syncthreads();
    float ca[2], cc[2], chi[2], calpha[2];
    unsigned lutc[2];
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) {
        const int c = 64 * cbp + 32 * cb + frow;
        const float4 pr = s_par[c];
        ca[cb] = pr.x; cc[cb] = pr.y; chi[cb] = pr.z; calpha[cb] = pr.w;
        // byte address of LUT[bin][c] = lut_base + bin * 512 + c * 4, with bin = bits(t) - MAGIC_BITS folded in (mod 2^32)
        lutc[cb] = (unsigned)(uintptr_t)(lds_u32p)s_lut + (unsigned)c * 4u - (MAGIC_BITS << 9);
    }

    // ---- tiles: (token pair, row tile), row tile fastest; static stride over the persistent workgroups
    const unsigned npair = (unsigned)(p.T + 1) >> 1;
    const unsigned ntile = npair * (unsigned)p.n_rt;
    const unsigned nwg = gridDim.x, bid = blockIdx.x;
    const int nk = p.nk;
    if (bid >= ntile) return;

    // ---- issue cursors.  Weights run FNS - 1 steps ahead of the compute cursor; the log2 run of step n is consumed one
    // step earlier than the weights of step n (the B fragments of step n are produced while step n - 1 multiplies), so its
    // cursor runs one step further ahead and its request travels with the weights of step n - 1.  A cursor is
    // (token pair, row tile, K-step); all of it is wave-uniform and advances without branches.
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
    auto issue_l = [&](int slot) __attribute__((always_inline)) {
        const int tok0 = (int)cl_.pair * 2;
        const __amdgpu_buffer_rsrc_t rl = __builtin_amdgcn_make_buffer_rsrc((void*)(p.L + (int64_t)tok0 * p.K), 0,
                                                                             min(2, p.T - tok0) * p.K * 4, 0x00020000);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rl, (las_ptr)(ring + slot * STG + AT + w * 256), 4, vl, cl_.k * 128, 0, 0);
        cur_step(cl_);
    };
    auto issue_a = [&](int slot) __attribute__((always_inline)) {
        uint8_t* st_ = ring + slot * STG;
        const int row0 = (int)ca_.rt * ROWS + w * 16 + lrow;
#pragma unroll
        for (int q = 0; q < RQ; ++q) {
            const int row = min(row0 + q * 64, p.M - 1);       // past M: re-read the last row (masked by s_rs = 0)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (las_ptr)(st_ + (w + 4 * q) * 1024), 16, row * p.Kb + lslot16, ca_.k * 64, 0, 0);
        }
        cur_step(ca_);
    };
    issue_l(0);                                      // log2 run of global step 0
#pragma unroll
    for (int s0 = 0; s0 < FNS - 1; ++s0) { issue_a(s0); issue_l((s0 + 1) % FNS); }

    v16f acc[NRB][2];
    uint32_t bA[2][2][4], bB[2][2][4];               // B fragments [candidate block][K half][dword]
    double run[2] = {0.0, 0.0};
    int qn = 0;                                      // tie-queue fill (wave-uniform)
    int st = 0;                                      // ring slot of the current global step
    int tok_abs = 0, m0 = 0;                         // this wave's token and the row-tile origin of the current tile

    // ---- B fragments of one (candidate block, K half) chunk: 8 element-candidates of this lane.
    // Returns max |kf - rne(kf)| over the chunk (the tie detector).
    auto gen_chunk = [&](const float4& l0, const float4& l1, int cb, uint32_t (&out)[4]) __attribute__((always_inline)) -> float {
        const float lv[8] = {l0.x, l0.y, l0.z, l0.w, l1.x, l1.y, l1.z, l1.w};
        uint32_t v[8];
        float dm = 0.0f;
#pragma unroll
        for (int e = 0; e < 8; e += 2) {
            const float k0 = __builtin_amdgcn_fmed3f(__builtin_fmaf(lv[e], ca[cb], cc[cb]), lo, chi[cb]);
            const float k1 = __builtin_amdgcn_fmed3f(__builtin_fmaf(lv[e + 1], ca[cb], cc[cb]), lo, chi[cb]);
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

    // ---- rank-1 correction of the accumulator column of candidate (cb, fr):  acc[o] += Wq[o, kabs] * (v_exact - v_fast),
    // as two MFMAs per row block with one live K slot: A = the weight column (lane (row, K group 0), element 0),
    // B = +v_exact / -v_fast in the owner column (both exact in bf16; their difference is not)
    auto apply = [&](int kabs, int cb, int fr, unsigned ve_bits, unsigned vf_bits) __attribute__((always_inline)) {
        const bool own = frow == fr && fkg == 0;
        const u4v b1 = {own ? ve_bits : 0u, 0u, 0u, 0u}, b2 = {own ? (vf_bits ^ 0x8000u) : 0u, 0u, 0u, 0u};
        const uint16_t* wcol = reinterpret_cast<const uint16_t*>(p.W) + kabs;
#pragma unroll
        for (int rb = 0; rb < NRB; ++rb) {
            const int row = m0 + rb * 32 + frow;
            const unsigned wv = (fkg == 0 && row < p.M) ? (unsigned)wcol[(int64_t)row * (p.Kb >> 1)] : 0u;
            const u4v a = {wv, 0u, 0u, 0u};
            if (cb == 0) {
                mfma_tile<NRB>(acc[rb][0], rb, __builtin_bit_cast(v8bf, a), __builtin_bit_cast(v8bf, b1));
                mfma_tile<NRB>(acc[rb][0], rb, __builtin_bit_cast(v8bf, a), __builtin_bit_cast(v8bf, b2));
            } else {
                mfma_tile<NRB>(acc[rb][1], rb, __builtin_bit_cast(v8bf, a), __builtin_bit_cast(v8bf, b1));
                mfma_tile<NRB>(acc[rb][1], rb, __builtin_bit_cast(v8bf, a), __builtin_bit_cast(v8bf, b2));
            }
        }
    };

    // ---- resolve the queued near-ties of the current tile, 64 at a time: exact pipeline, fix-up where it differs
    auto resolve = [&]() __attribute__((always_inline)) {
        for (int base = 0; base < qn; base += 64) {
            const int i = base + lane;
            const bool act = i < qn;
            const uint32_t rec = act ? myq[i] : 0u;
            const int kabs = rec & 0xFFFF, cbq = (rec >> 16) & 1, fr = (rec >> 17) & 31, fb = rec >> 24;
            const int c = 64 * cbp + 32 * cbq + fr;
            float kk = (float)fb;
            if (act) {
                const float xs = p.x[(int64_t)tok_abs * p.K + min(kabs, p.K - 1)] + p.shift;
                const float2 sq = s_sq[c];
                float ue = xs / sq.x;
                if (p.clamp_u) ue = fminf(fmaxf(ue, 1e-15f), 1.0f);
                const float le = (float)log2((double)ue);                    // correctly rounded log2 (common.h: adalog_k)
                const float t = (-le) * 37.0f / sq.y;
                kk = rintf(t);
                kk = (kk == kk) ? fminf(fmaxf(kk, 0.0f), (float)(p.levels2 + 1)) : (float)(p.levels2 + 1);
            }
            const uint32_t ve = s_lut[(int)kk * 128 + c], vf = s_lut[fb * 128 + c];
            unsigned long long mm = __ballot(act && ve != vf && kabs < p.K);
            while (mm) {                                                      // ~1e-5 of element-candidates
                const int j = __ffsll(mm) - 1;
                mm &= mm - 1;
                const uint32_t rj = __builtin_amdgcn_readlane(rec, j);
                apply(rj & 0xFFFF, (rj >> 16) & 1, (rj >> 17) & 31, __builtin_amdgcn_readlane(ve, j), __builtin_amdgcn_readlane(vf, j));
            }
        }
        qn = 0;
    };

    // ---- queue the near-ties of a chunk (cold): recompute the 8 fast values, push (k, candidate, fast bin) per hit
    auto push_chunk = [&](const float4& l0, const float4& l1, int cb, int kbase) __attribute__((always_inline)) {
        const float lv[8] = {l0.x, l0.y, l0.z, l0.w, l1.x, l1.y, l1.z, l1.w};
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float kf = __builtin_amdgcn_fmed3f(__builtin_fmaf(lv[e], ca[cb], cc[cb]), lo, chi[cb]);
            const float t = kf + MAGIC;
            const bool f = fabsf(kf - (t - MAGIC)) > TIE;
            const unsigned long long m = __ballot(f);
            if (m) {
                const int idx = qn + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
                if (f) myq[idx] = (unsigned)(kbase + e) | ((unsigned)cb << 16) | ((unsigned)frow << 17) | ((__float_as_uint(t) & 0xFFu) << 24);
                qn += (int)__popcll(m);
            }
        }
        if (qn > QCAP - 512) resolve();                                       // room for a whole chunk at the next push
    };

    // ---- one 64-byte K-step: MFMAs of step n from BC, B fragments of step n + 1 into BN (GEN).  Four groups, each =
    // one chunk of generation + (half the row blocks of one K half) x 2 candidate blocks = NRB MFMAs; a scheduling
    // barrier per group keeps the fragment look-ahead (and with it the register pressure) to one group.
#define FUSED_STEP(BC, BN, GEN, KT)                                                                               \
    do {                                                                                                          \
        if (RW == 7) asm volatile("s_waitcnt vmcnt(14)" ::: "memory");                                            \
        else if (RW == 5) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");                                       \
        else if (RW == 4) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");                                        \
        else if (RW == 3) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");                                        \
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                     \
        __builtin_amdgcn_s_barrier();                                                                             \
        asm volatile("" ::: "memory");                                                                            \
        const uint8_t* As_ = ring + st * STG;                                                                     \
        const int stn_ = st == FNS - 1 ? 0 : st + 1;                                                              \
        const uint8_t* Ls_ = ring + stn_ * STG + AT + w * 256;                                                    \
        issue_a(st == 0 ? FNS - 1 : st - 1);                                                                      \
        issue_l(st);                                                                                              \
        float4 l0_ = make_float4(0.f, 0.f, 0.f, 0.f), l1_ = l0_;                                                  \
        _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                           \
            const int h_ = j >> 1, cbg_ = j & 1;                                                                  \
            float dm_ = 0.0f;                                                                                     \
            if (GEN) {                                                                                            \
                if (cbg_ == 0) { l0_ = lds_f4(Ls_, (16 * h_ + 8 * fkg) * 4); l1_ = lds_f4(Ls_, (16 * h_ + 8 * fkg) * 4 + 16); } \
                dm_ = gen_chunk(l0_, l1_, cbg_, BN[cbg_][h_]);                                                    \
            }                                                                                                     \
            _Pragma("unroll") for (int r2 = 0; r2 < NRB / 2; ++r2) {                                              \
                const int rb = cbg_ * (NRB / 2) + r2;                                                             \
                const uint4 a_ = lds_frag(As_, (rb * 32 + frow) * 64 + (((2 * h_ + fkg) ^ ((frow >> 2) & 3)) << 4)); \
                mfma_tile<NRB>(acc[rb][0], rb, __builtin_bit_cast(v8bf, a_), frag8(BC[0][h_]));                   \
                mfma_tile<NRB>(acc[rb][1], rb, __builtin_bit_cast(v8bf, a_), frag8(BC[1][h_]));                   \
            }                                                                                                     \
            __builtin_amdgcn_sched_barrier(0);                                                                    \
            if (GEN && __builtin_expect(__any(dm_ > TIE), 0)) push_chunk(l0_, l1_, cbg_, ((KT) + 1) * 32 + 16 * h_ + 8 * fkg); \
        }                                                                                                         \
        st = stn_;                                                                                                \
    } while (0)

    Cur cc_ = cur_init();                            // compute cursor (tile granularity)
    for (; cc_.tile < ntile; cc_.tile += nwg) {
        const unsigned pair = cc_.tile / (unsigned)p.n_rt;
        m0 = (int)(cc_.tile - pair * (unsigned)p.n_rt) * ROWS;
        const int tok0 = (int)pair * 2;
        tok_abs = min(tok0 + wtok, p.T - 1);
        const bool tok_ok = tok0 + wtok < p.T;
#pragma unroll
        for (int rb = 0; rb < NRB; ++rb)
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc[rb][0][r] = 0.0f; acc[rb][1][r] = 0.0f; }
        // epilogue operands: plain loads now, staged into LDS after the second step's barrier
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
        // B fragments of the tile's first step (its log2 run travelled with the previous tile's last weights)
        {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // once per tile: also covers the very first step
            const uint8_t* Ls0 = ring + st * STG + AT + w * 256;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int h_ = j >> 1, cbg_ = j & 1;
                const float4 l0_ = lds_f4(Ls0, (16 * h_ + 8 * fkg) * 4), l1_ = lds_f4(Ls0, (16 * h_ + 8 * fkg) * 4 + 16);
                const float dm_ = gen_chunk(l0_, l1_, cbg_, bA[cbg_][h_]);
                if (__builtin_expect(__any(dm_ > TIE), 0)) push_chunk(l0_, l1_, cbg_, 16 * h_ + 8 * fkg);
            }
        }
        for (int kt = 0; kt < nk; kt += 2) {
            FUSED_STEP(bA, bB, true, kt);
            if (kt == 2) {                                          // stage the epilogue operands (read after >= 1 more barrier)
#pragma unroll
                for (int u = 0; u < EU; ++u) if (tid + u * 256 < 2 * ROWS) s_refb[tid + u * 256] = e_ref[u];
#pragma unroll
                for (int u = 0; u < RU; ++u) if (tid + u * 256 < ROWS) s_rs[tid + u * 256] = e_rs[u];
            }
            if (kt + 2 < nk) FUSED_STEP(bB, bA, true, kt + 1);
            else FUSED_STEP(bB, bA, false, kt + 1);
        }
        if (qn) resolve();
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
#undef FUSED_STEP
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // drain the run-ahead requests before the LDS is released
    if (fkg == 0) { s_fin[w * 64 + frow] = run[0]; s_fin[w * 64 + 32 + frow] = run[1]; }
    __syncthreads();
    if (tid < 128) {
        const int cw = tid >> 6, wi = tid & 63;
        p.wg_acc[(int64_t)bid * 128 + tid] = s_fin[cw * 64 + wi] + s_fin[(2 + cw) * 64 + wi];
    }
#endif
}

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

static int pick_nrb(int M) {
    const int cand[4] = {12, 8, 6, 4};
    int best = 12;
    int64_t best_pad = -1;
    for (int i = 0; i < 4; ++i) {
        const int rows = 32 * cand[i];
        const int64_t pad = (int64_t)((M + rows - 1) / rows) * rows;
        if (best_pad < 0 || pad < best_pad) { best_pad = pad; best = cand[i]; }
    }
    return best;
}

static size_t fused_lds_bytes(int nrb, int levels2) {
    return (size_t)FNS * (nrb * 2048 + 1024) + (size_t)(levels2 + 2) * 512 + 128 * 16 + 128 * 8 + (size_t)3 * 32 * nrb * 4 +
           256 * 8 + (size_t)4 * QCAP * 4;
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
    if (Kp < K || (Kp * 2) % 128 != 0 || Kp / 32 < 6 || Kp >= 65536) return 0;     // >= 6 K-steps; k fits the queue record
    if ((int64_t)M * Kp * 2 >= ((int64_t)1 << 31) || T >= ((int64_t)1 << 30)) return 0;
    return fused_lds_bytes(pick_nrb(M), 1 << n_bits) <= 160 * 1024 ? 1 : 0;
}

extern "C" int64_t adalog_score_act_fused_workspace_bytes(void) { return (int64_t)fused_cus() * 128 * 8; }

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
    ADALOG_ARG_CHECK(workspace_bytes >= adalog_score_act_fused_workspace_bytes() && ((uintptr_t)workspace & 7) == 0,
                     "score_act_fused: workspace too small or misaligned");
    FusedArgs a{};
    a.W = (const uint8_t*)Wp; a.L = Lx; a.x = x; a.ref = ref; a.row_scale = row_scale; a.row_bias = row_bias;
    a.scale = scale; a.qv = qv; a.mant = mant37; a.wg_acc = (double*)workspace;
    a.M = M; a.T = (int)T; a.K = K; a.Kb = (int)(Kp * 2); a.levels2 = 1 << n_bits; a.clamp_u = clamp_u;
    a.nk = (int)(Kp * 2 / 64); a.shift = shift; a.sa_mul = sa_mul;
    const int nrb = pick_nrb(M);
    a.n_rt = (M + 32 * nrb - 1) / (32 * nrb);
    const int64_t ntile = ((T + 1) / 2) * a.n_rt;
    const int nwg = (int)(ntile < fused_cus() ? ntile : fused_cus());
    const size_t shm = fused_lds_bytes(nrb, a.levels2);
    hipStream_t st = (hipStream_t)stream;
#define LAUNCH_FUSED(NRBV)                                                                                     \
    do {                                                                                                       \
        static bool attr_set = false;                                                                          \
        if (!attr_set) {                                                                                       \
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_act_fused<NRBV>),                       \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);                 \
            attr_set = true;                                                                                   \
        }                                                                                                      \
        hipLaunchKernelGGL((k_act_fused<NRBV>), dim3((unsigned)nwg), dim3(256), shm, st, a);                   \
    } while (0)
    if (nrb == 12) LAUNCH_FUSED(12);
    else if (nrb == 8) LAUNCH_FUSED(8);
    else if (nrb == 6) LAUNCH_FUSED(6);
    else LAUNCH_FUSED(4);
#undef LAUNCH_FUSED
    ADALOG_LAUNCH_CHECK("adalog_score_act_fused");
    hipLaunchKernelGGL(k_fused_finish, dim3(128), dim3(64), 0, st, (const double*)workspace, nwg, norm, scores);
    ADALOG_LAUNCH_CHECK("adalog_score_act_fused (finish)");
    return 0;
}
