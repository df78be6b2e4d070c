// Operand packing for the candidate-scoring GEMMs (K7/K8/K11-K14 prologue).
//
// The reference re-materialises an fp32 fake-quantised copy of the searched operand for every candidate
// (quant_layers/linear.py:369-371,409-411,830-837; matmul.py:150-151,188-189,337-342; conv.py:240-242).  Here each
// operand is packed ONCE per scoring call into an MFMA-ready, K-contiguous compact image:
//   * uniformly quantised operands  -> int8   (q - z)                exact small integers  
//   * AdaLog operands               -> bf16   m * 2^-t               m = LUT numerator <= 4L-2, exact in bf16
//   * unquantised conv input        -> fp32   zero-padded copy
// so a candidate costs 1 B (2 B) per element instead of 4 B x ~8 temporaries, and the de-quantisation scales are
// applied once per output in the GEMM epilogue.  Output layout: out[c][g][r][Kp], Kp = K rounded up to 64 bytes,
// zero padded (zeros contribute nothing to the integer dot product).
#include "common.h"
#include <hip/hip_bf16.h>
#include <type_traits>
#include <stdlib.h>

namespace {

enum { KIND_UNIFORM = 0, KIND_ADALOG = 1, KIND_RAW = 2 };

struct PackArgs {
    const float* x;
    int64_t G, R, K, sxg, sxr, sxk;
    // parameter addressing: idx = c*pc + (g % gmod)*pg + r*pr
    const float* scale;
    const float* zp;      // uniform: zero point (rounded in-kernel); adalog: unused
    const float* qv;      // adalog: log-base numerator q per candidate (same addressing, pr ignored)
    int64_t C, pc, gmod, pg, pr;
    float qmax;           // 2L-1
    int levels2;          // 2L
    const float* mant;    // adalog: 37 integer numerators of the search table (linear.py:750-752)
    const float* shift;   // adalog: optional input shift (post-GELU), device scalar
    int clamp_u;          // adalog: clamp((x+shift)/s, 1e-15, 1) (linear.py:830) or raw log2 (matmul.py:337)
    void* out;
    int64_t Kp;
    int32_t* rowsum;      // optional [C][G][R] sum_k (q - z)
    int c_inner;          // 0: out[c][g][r][Kp]   1: out[g][r][c][Kp] (candidates innermost: GEMM columns = (row, candidate))
    int pre;              // adalog, fast kernel: 1 = the operand is GELU(x) (erf form, ATen's expression): quant_forward of fc2 reads fc1's output
};

// x * 0.5 * (1 + erf(x / sqrt 2)) as ATen's GeluCUDAKernelImpl evaluates it in fp32 ("none" approximation)
__device__ __forceinline__ float gelu_erf(float x) { return (x * 0.5f) * (1.0f + erff(x * 0.70710678118654752440f)); }

struct fp8_t { uint8_t b; };   // e4m3 byte as the hardware converts it (v_cvt_pk_fp8_f32); integers up to 16 are exact

template <typename T> struct Out;
template <> struct Out<int8_t> { static constexpr int EPT = 16; };
template <> struct Out<fp8_t> { static constexpr int EPT = 16; };
template <> struct Out<__hip_bfloat16> { static constexpr int EPT = 8; };
template <> struct Out<float> { static constexpr int EPT = 4; };

template <typename T> __device__ __forceinline__ T cvt(float v);
template <> __device__ __forceinline__ int8_t cvt<int8_t>(float v) { return (int8_t)(int)v; }
template <> __device__ __forceinline__ fp8_t cvt<fp8_t>(float v) {
    fp8_t r; r.b = (uint8_t)(__builtin_amdgcn_cvt_pk_fp8_f32(v, 0.0f, 0, false) & 0xff); return r;
}
template <> __device__ __forceinline__ __hip_bfloat16 cvt<__hip_bfloat16>(float v) { return __float2bfloat16(v); }
template <> __device__ __forceinline__ float cvt<float>(float v) { return v; }

template <typename T, int KIND, bool RFAST>
__global__ __launch_bounds__(256) void k_pack(PackArgs a) {
    constexpr int EPT = Out<T>::EPT;
    const int64_t nchunk = a.Kp / EPT;
    const int64_t per_c = a.G * a.R * nchunk;
    const int64_t c = blockIdx.y;
    __shared__ float s_mant[ADALOG_R];
    if (KIND == KIND_ADALOG) {
        if (threadIdx.x < ADALOG_R) s_mant[threadIdx.x] = a.mant[threadIdx.x];
        __syncthreads();
    }
    const float sh = (KIND == KIND_ADALOG && a.shift) ? a.shift[0] : 0.0f;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < per_c; idx += (int64_t)gridDim.x * blockDim.x) {
        int64_t g, r, ch;
        if (RFAST) {
            r = idx % a.R;
            int64_t t = idx / a.R;
            ch = t % nchunk;
            g = t / nchunk;
        } else {
            ch = idx % nchunk;
            int64_t t = idx / nchunk;
            r = t % a.R;
            g = t / a.R;
        }
        const int64_t pidx = c * a.pc + (g % a.gmod) * a.pg + r * a.pr;
        float s = 1.0f, z = 0.0f, qf = 37.0f;
        if (KIND != KIND_RAW) s = a.scale[pidx];
        if (KIND == KIND_UNIFORM) z = rintf(a.zp[pidx]);
        if (KIND == KIND_ADALOG) qf = a.qv[c * a.pc + (g % a.gmod) * a.pg];
        const float* xp = a.x + g * a.sxg + r * a.sxr;
        alignas(16) T vals[EPT];
        int isum = 0;
#pragma unroll
        for (int e = 0; e < EPT; ++e) {
            const int64_t k = ch * EPT + e;
            float v = 0.0f;
            if (k < a.K) {
                const float xv = xp[k * a.sxk];
                if (KIND == KIND_UNIFORM) {
                    v = fminf(fmaxf(rintf(xv / s) + z, 0.0f), a.qmax) - z;
                    isum += (int)v;
                } else if (KIND == KIND_ADALOG) {
                    float u = (a.shift ? xv + sh : xv) / s;
                    if (a.clamp_u) u = fminf(fmaxf(u, 1e-15f), 1.0f);
                    float kk = adalog_k(u, qf);
                    if (kk < (float)a.levels2 && kk == kk) {        // masked bins (and NaN) -> 0
                        kk = fmaxf(kk, 0.0f);
                        const int kq = (int)kk * (int)qf;             // exact: k <= 255, q <= 137
                        const int t = kq / ADALOG_R, j = kq - t * ADALOG_R;
                        v = (t > 100) ? 0.0f : ldexpf(s_mant[j], -t);  // < 2^-100: below fp32 resolution of the sums
                    }
                } else {
                    v = xv;
                }
            }
            vals[e] = cvt<T>(v);
        }
        const int64_t orow = a.c_inner ? (g * a.R + r) * a.C + c : (c * a.G + g) * a.R + r;
        T* op = reinterpret_cast<T*>(a.out) + orow * a.Kp + ch * EPT;
        *reinterpret_cast<uint4*>(op) = *reinterpret_cast<const uint4*>(vals);
        if (KIND == KIND_UNIFORM && a.rowsum) atomicAdd(a.rowsum + (c * a.G + g) * a.R + r, isum);
    }
}

// K-contiguous sources (activations, weights, q@k^T operands): one thread owns 4 consecutive k of one row and loops over
// ALL candidates, so the fp32 source is read once (coalesced float4) instead of once per candidate, and every store
// instruction of a wave writes one contiguous 256 B / 512 B / 1 KiB run.
template <typename T, int KIND>
__global__ __launch_bounds__(256) void k_pack_kfast(PackArgs a) {
    constexpr bool FP8OUT = sizeof(T) == 1 && !std::is_same<T, int8_t>::value;
    const int64_t nq = a.Kp >> 2;
    const int64_t total = a.G * a.R * nq;
    // AdaLog: per-candidate value LUT  lut[c][k] = m[(k q_c) mod 37] * 2^-floor(k q_c / 37)  (bf16-exact), built once per
    // block, so a candidate costs ~8 VALU + one LDS read per element; log2(x + shift) is taken once per element and
    // -log2(u) = log2(s_c) - log2(x + shift) per candidate (exact path re-evaluated inside the tie zone).
    extern __shared__ unsigned short s_lut[];
    const float sh = (KIND == KIND_ADALOG && a.shift) ? a.shift[0] : 0.0f;
    if (KIND == KIND_ADALOG) {
        const int lw = a.levels2 + 1;                                        // entry [2L] = 0: masked bins
        for (int e = threadIdx.x; e < (int)a.C * lw; e += blockDim.x) {
            const int c = e / lw, k = e - c * lw;
            const int kqv = k * (int)a.qv[c * a.pc];
            const int t = kqv / ADALOG_R, j = kqv - t * ADALOG_R;
            const float v = (k == a.levels2 || t > 100) ? 0.0f : ldexpf(a.mant[j], -t);
            s_lut[e] = (unsigned short)(__float_as_uint(v) >> 16);          // exact: <= 8 significant bits
        }
        __syncthreads();
    }
    const float NL15 = 49.828921f;                                           // -fl32(log2(1e-15f))
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int64_t kq = idx % nq;
        const int64_t t0 = idx / nq;
        const int64_t r = t0 % a.R, g = t0 / a.R;
        const int64_t k0 = kq << 2;
        const float* xp = a.x + g * a.sxg + r * a.sxr + k0;
        float xv[4];
        if (k0 + 3 < a.K && ((uintptr_t)xp & 15) == 0) {
            const float4 v = *reinterpret_cast<const float4*>(xp);
            xv[0] = v.x; xv[1] = v.y; xv[2] = v.z; xv[3] = v.w;
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) xv[e] = (k0 + e < a.K) ? xp[e] : 0.0f;
        }
        float lx[4] = {0.f, 0.f, 0.f, 0.f};
        if (KIND == KIND_ADALOG) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (a.shift) xv[e] += sh;
                lx[e] = __log2f(xv[e]);                                       // -inf for 0, NaN for negatives
                if (a.clamp_u && !(lx[e] == lx[e])) lx[e] = -__builtin_inff(); // negatives clamp to u = 1e-15 like zeros
            }
        }
        const bool full = k0 + 3 < a.K;
        const int64_t pbase = (g % a.gmod) * a.pg + r * a.pr;
        // (uniform: the next candidate's scale and zero point are fetched under the current one's arithmetic -- with per-row
        // parameters each iteration otherwise starts with an L2 round trip: 1.8 TB/s written for the fc2 weight candidates)
        float s_nx = 1.0f, z_nx = 0.0f;
        if (KIND == KIND_UNIFORM && (int64_t)blockIdx.y < a.C) { s_nx = a.scale[blockIdx.y * a.pc + pbase]; z_nx = a.zp[blockIdx.y * a.pc + pbase]; }
        for (int64_t c = blockIdx.y; c < a.C; c += gridDim.y) {
            alignas(16) T vals[4];
            int isum = 0;
            if (KIND == KIND_RAW) {
#pragma unroll
                for (int e = 0; e < 4; ++e) vals[e] = cvt<T>(xv[e]);
            } else if (KIND == KIND_UNIFORM) {
                // per element: mul, rint, sub, cmp (tie zone -> exact divide, rare), med3 [, add, cvt_pk_u8]
                const float s = s_nx, z = rintf(z_nx);
                if (c + gridDim.y < a.C) { s_nx = a.scale[(c + gridDim.y) * a.pc + pbase]; z_nx = a.zp[(c + gridDim.y) * a.pc + pbase]; }
                const float inv_s = __builtin_amdgcn_rcpf(s);
                const float lo = -z, hi = a.qmax - z;                     // clamp(k + z, 0, qmax) - z == med3(k, -z, qmax - z)
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float t = xv[e] * inv_s;
                    float k = rintf(t);
                    // tie zone: the reciprocal path is within 1.8e-7 * |t| of x/s, i.e. < 1e-4 wherever the clamp does not
                    // decide anyway (|t| < 555), so only |frac - 0.5| < 1e-4 needs the exact quotient
                    if (__builtin_expect(fabsf(t - k) > 0.4999f, 0)) k = rintf(xv[e] / s);
                    v[e] = __builtin_amdgcn_fmed3f(k, lo, hi);
                    if (!full && !(k0 + e < a.K)) v[e] = 0.0f;
                }
                if (a.rowsum) isum = (int)((v[0] + v[1]) + (v[2] + v[3]));    // small integers: exact in fp32
                if (FP8OUT) {
                    int pk = __builtin_amdgcn_cvt_pk_fp8_f32(v[0], v[1], 0, false);
                    pk = __builtin_amdgcn_cvt_pk_fp8_f32(v[2], v[3], pk, true);
                    *reinterpret_cast<int*>(vals) = pk;
                } else if (sizeof(T) == 1) {
                    // (v + 128) saturates into a byte lane per instruction; xor 0x80 turns it into two's complement
                    unsigned pk = 0;
#pragma unroll
                    for (int e = 0; e < 4; ++e) pk = __builtin_amdgcn_cvt_pk_u8_f32(v[e] + 128.0f, e, pk);
                    *reinterpret_cast<unsigned*>(vals) = pk ^ 0x80808080u;
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) vals[e] = cvt<T>(v[e]);
                }
            } else {
                // per element: fma, med3, rint, sub, cmp (tie zone -> exact path, rare), min, cvt, LDS read
                const float s = a.scale[c * a.pc + pbase];
                const float qf = a.qv[c * a.pc];
                const float rq37 = 37.0f / qf, lsr = __log2f(s) * rq37;
                const float tmax = NL15 * rq37, lvl1 = (float)a.levels2 + 1.0f;
                const unsigned short* lut = s_lut + c * (a.levels2 + 1);
                unsigned short hv[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float t = lsr - lx[e] * rq37;                             // (log2 s - log2 x) * 37 / q
                    if (a.clamp_u) t = __builtin_amdgcn_fmed3f(t, 0.0f, tmax); // u clamped to [1e-15, 1]  (lx = -inf for x <= 0)
                    float kk = rintf(t);
                    // tie zone: for t < 2L + 1 (larger bins are masked to 0 whatever they round to) the fast t is within ~1e-5
                    // of the reference's fp32 pipeline, so |frac - 0.5| < 1e-4 is where the exact path (fp64 log2) decides.
                    // Any lane taking it stalls its whole wave: the margin is kept as narrow as the error bound allows.
                    if (__builtin_expect(fabsf(t - kk) > 0.4999f && t < lvl1 && t > -1.0f, 0)) {
                        float ue = xv[e] / s;
                        if (a.clamp_u) ue = fminf(fmaxf(ue, 1e-15f), 1.0f);
                        kk = adalog_k(ue, qf);
                    }
                    // bins >= 2L are masked: LUT entry [2L] is 0.  NaN (x < 0 without the clamp) -> 0 as well.
                    const float kc = fminf(fmaxf(kk, 0.0f), (float)a.levels2);
                    unsigned short h = lut[(int)kc];
                    if (!a.clamp_u && !(kk == kk)) h = 0;
                    if (!full && !(k0 + e < a.K)) h = 0;
                    hv[e] = h;
                }
                *reinterpret_cast<uint2*>(vals) = make_uint2((unsigned)hv[0] | ((unsigned)hv[1] << 16),
                                                              (unsigned)hv[2] | ((unsigned)hv[3] << 16));
            }
            const int64_t orow = a.c_inner ? (g * a.R + r) * a.C + c : (c * a.G + g) * a.R + r;
            T* op = reinterpret_cast<T*>(a.out) + orow * a.Kp + k0;
            if (sizeof(T) == 1) *reinterpret_cast<uint32_t*>(op) = *reinterpret_cast<const uint32_t*>(vals);
            else if (sizeof(T) == 2) *reinterpret_cast<uint2*>(op) = *reinterpret_cast<const uint2*>(vals);
            else *reinterpret_cast<uint4*>(op) = *reinterpret_cast<const uint4*>(vals);
            if (KIND == KIND_UNIFORM && a.rowsum) {
                // one atomic per wavefront when all 64 lanes work on the same row (the common case: Kp/4 >= 64)
                const int64_t ridx = (c * a.G + g) * a.R + r;
                const int64_t first = __builtin_amdgcn_readfirstlane((int)ridx);
                if (__all((int)ridx == (int)first) && __popcll(__ballot(1)) == 64) {
                    int v = isum;
#pragma unroll
                    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
                    if ((threadIdx.x & 63) == 0 && v != 0) atomicAdd(a.rowsum + ridx, v);
                } else if (isum != 0) {
                    atomicAdd(a.rowsum + ridx, isum);
                }
            }
        }
    }
}

// AdaLog packing, per-tensor scale (pg == 0), K-contiguous source: the hot form (post-GELU and post-softmax
// searches write 2.5 GB of bf16 per call).  Everything that depends only on the candidate -- the value LUT, 37/q,
// log2(s)*37/q, the clamp bound -- is built once per block in LDS, so a candidate costs per element
//   fma, med3, rndne, sub, cmp(tie zone), cvt, address, ds_read_u16   (+ 1/2 v_perm, 1/4 store).
// Bins >= 2L and x <= 0 without the u-clamp read one of two zero entries behind each LUT row, so no select is needed.
__global__ __launch_bounds__(256) void k_pack_adalog_fast(PackArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char s_raw[];
    const int lw = a.levels2 + 2;
    float4* s_par = reinterpret_cast<float4*>(s_raw);                       // [C] {37/q, log2(s)*37/q, lower, upper}
    unsigned short* s_lut = reinterpret_cast<unsigned short*>(s_raw + (size_t)a.C * sizeof(float4));   // [C][2L + 2]
    const float NL15 = 49.828921f;                                           // -fl32(log2(1e-15f))
    for (int e = threadIdx.x; e < (int)a.C * lw; e += blockDim.x) {
        const int c = e / lw, k = e - c * lw;
        const int kqv = k * (int)a.qv[c * a.pc];
        const int t = kqv / ADALOG_R, j = kqv - t * ADALOG_R;
        const float v = (k >= a.levels2 || t > 100) ? 0.0f : ldexpf(a.mant[j], -t);
        s_lut[e] = (unsigned short)(__float_as_uint(v) >> 16);              // exact: <= 8 significant bits
    }
    for (int c = threadIdx.x; c < (int)a.C; c += blockDim.x) {
        const float s = a.scale[c * a.pc], qf = a.qv[c * a.pc];
        const float rq37 = 37.0f / qf;
        const float top = (float)a.levels2 + 0.75f;                          // rounds to 2L + 1: a zero entry
        s_par[c] = make_float4(rq37, __log2f(s) * rq37, a.clamp_u ? 0.0f : -0.25f,
                               a.clamp_u ? fminf(NL15 * rq37, top) : top);
    }
    __syncthreads();
    const float sh = a.shift ? a.shift[0] : 0.0f;
    const int64_t nq = a.Kp >> 2;
    const int64_t total = a.G * a.R * nq;
    const int64_t cstep = gridDim.y;
    const bool ragged = (a.K & 3) != 0;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int64_t kq = idx % nq;
        const int64_t t0 = idx / nq;
        const int64_t r = t0 % a.R, g = t0 / a.R;
        const int64_t k0 = kq << 2;
        float xv[4] = {0.f, 0.f, 0.f, 0.f};
        const int nlive = (int)min((int64_t)4, max((int64_t)0, a.K - k0));   // data elements of this quad (rest: zero padding)
        const bool live = nlive > 0;
        if (live) {
            const float* xp = a.x + g * a.sxg + r * a.sxr + k0;
            if (nlive == 4 && ((uintptr_t)xp & 15) == 0) {
                const float4 v = *reinterpret_cast<const float4*>(xp);
                xv[0] = v.x; xv[1] = v.y; xv[2] = v.z; xv[3] = v.w;
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) if (e < nlive) xv[e] = xp[e];
            }
            if (a.pre == 1) {
#pragma unroll
                for (int e = 0; e < 4; ++e) xv[e] = gelu_erf(xv[e]);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) if (e < nlive) xv[e] += sh;
        }
        float lx[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            lx[e] = __log2f(xv[e]);                                          // -inf for 0, NaN for negatives
            if (!(lx[e] == lx[e])) lx[e] = -__builtin_inff();                // x < 0: u clamps to 1e-15 / bin masked, like x = 0
        }
        const int64_t c0 = blockIdx.y;
        const int64_t orow = a.c_inner ? (g * a.R + r) * a.C + c0 : (c0 * a.G + g) * a.R + r;
        const int64_t ostep = (a.c_inner ? cstep : cstep * a.G * a.R) * a.Kp;   // elements between this thread's candidates
        __hip_bfloat16* op = reinterpret_cast<__hip_bfloat16*>(a.out) + orow * a.Kp + k0;
        for (int64_t c = c0; c < a.C; c += cstep, op += ostep) {
            uint2 pk = make_uint2(0u, 0u);
            if (live) {
                const float4 pr = s_par[c];
                const unsigned short* lut = s_lut + c * lw;
                unsigned short hv[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    // (log2 s - log2 x) * 37/q, one explicit fma: this value only steers the fast path (the tie zone covers its error)
                    const float t = __builtin_amdgcn_fmed3f(__builtin_fmaf(-lx[e], pr.x, pr.y), pr.z, pr.w);
                    float kk = rintf(t);
                    // within 1e-4 of a rounding tie the fast t (error ~1e-5 below 2L + 1) does not decide: exact path.
                    // One lane taking it stalls the wave, so the zone is as narrow as the error bound allows.
                    if (__builtin_expect(fabsf(t - kk) > 0.4999f, 0)) {
                        const float s = a.scale[c * a.pc], qf = a.qv[c * a.pc];
                        float ue = xv[e] / s;
                        if (a.clamp_u) ue = fminf(fmaxf(ue, 1e-15f), 1.0f);
                        kk = adalog_k(ue, qf);
                        kk = (kk == kk) ? fminf(fmaxf(kk, 0.0f), (float)(a.levels2 + 1)) : (float)(a.levels2 + 1);
                    }
                    hv[e] = lut[(int)kk];
                }
                if (ragged) {                                                // K % 4 != 0: the row's last quad is part padding
#pragma unroll
                    for (int e = 1; e < 4; ++e) if (e >= nlive) hv[e] = 0;
                }
                pk = make_uint2((unsigned)hv[0] | ((unsigned)hv[1] << 16), (unsigned)hv[2] | ((unsigned)hv[3] << 16));
            }
            *reinterpret_cast<uint2*>(op) = pk;
        }
    }
}

// Uniform int8 packing with per-tensor candidates (pg == pr == 0: the activation searches, 0.3-1.2 GB per call): the
// candidate's reciprocal scale and clamp bounds come from LDS; per element  mul, rndne, sub, cmp(tie zone), med3, add,
// cvt_pk_u8;  the output pointer advances by a constant per candidate.
template <bool FP8>
__global__ __launch_bounds__(256) void k_pack_uniform_i8_fast(PackArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char s_raw[];
    float4* s_par = reinterpret_cast<float4*>(s_raw);                       // [C] {1/s, s, -z + bias, qmax - z + bias}
    const float vbias = FP8 ? 0.0f : 128.0f;                                 // int8 goes through a biased u8 conversion
    // per-group parameters (pg != 0: one (scale, zp) per candidate and head): grid.z walks the groups, so a block's
    // parameters are still one LDS table
    const bool per_group = a.pg != 0;
    const int64_t pgo = per_group ? ((int64_t)blockIdx.z % a.gmod) * a.pg : 0;
    for (int c = threadIdx.x; c < (int)a.C; c += blockDim.x) {
        const float s = a.scale[c * a.pc + pgo], z = rintf(a.zp[c * a.pc + pgo]);
        s_par[c] = make_float4(__builtin_amdgcn_rcpf(s), s, vbias - z, vbias + (a.qmax - z));
    }
    __syncthreads();
    const int64_t nq = a.Kp >> 2;
    const int64_t total = (per_group ? 1 : a.G) * a.R * nq;
    const int64_t cstep = gridDim.y;
    const bool ragged = (a.K & 3) != 0;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int64_t kq = idx % nq;
        const int64_t t0 = idx / nq;
        const int64_t r = t0 % a.R, g = per_group ? (int64_t)blockIdx.z : t0 / a.R;
        const int64_t k0 = kq << 2;
        float xv[4] = {0.f, 0.f, 0.f, 0.f};
        const int nlive = (int)min((int64_t)4, max((int64_t)0, a.K - k0));
        const bool live = nlive > 0;
        if (live) {
            const float* xp = a.x + g * a.sxg + r * a.sxr + k0;
            if (nlive == 4 && ((uintptr_t)xp & 15) == 0) {
                const float4 v = *reinterpret_cast<const float4*>(xp);
                xv[0] = v.x; xv[1] = v.y; xv[2] = v.z; xv[3] = v.w;
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) if (e < nlive) xv[e] = xp[e];
            }
        }
        const int64_t c0 = blockIdx.y;
        const int64_t orow = a.c_inner ? (g * a.R + r) * a.C + c0 : (c0 * a.G + g) * a.R + r;
        const int64_t ostep = (a.c_inner ? cstep : cstep * a.G * a.R) * a.Kp;
        int8_t* op = reinterpret_cast<int8_t*>(a.out) + orow * a.Kp + k0;
        for (int64_t c = c0; c < a.C; c += cstep, op += ostep) {
            unsigned pk = FP8 ? 0u : 0x80808080u;                             // (biased) zeros
            if (live) {
                const float4 pr = s_par[c];
                float vb[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float t = xv[e] * pr.x;
                    float k = rintf(t);
                    if (__builtin_expect(fabsf(t - k) > 0.4999f, 0)) k = rintf(xv[e] / pr.y);   // see k_pack_kfast
                    vb[e] = __builtin_amdgcn_fmed3f(k + vbias, pr.z, pr.w);                     // int8: (q - z) + 128 in [1, 255]
                    if (ragged && e >= nlive) vb[e] = vbias;
                }
                if (FP8) {
                    int q = __builtin_amdgcn_cvt_pk_fp8_f32(vb[0], vb[1], 0, false);
                    pk = (unsigned)__builtin_amdgcn_cvt_pk_fp8_f32(vb[2], vb[3], q, true);
                } else {
                    pk = 0;
#pragma unroll
                    for (int e = 0; e < 4; ++e) pk = __builtin_amdgcn_cvt_pk_u8_f32(vb[e], e, pk);
                }
            }
            *reinterpret_cast<unsigned*>(op) = FP8 ? pk : pk ^ 0x80808080u;   // int8: un-bias to two's complement
        }
    }
}

// Uniform packing with per-ROW candidate parameters (pr != 0: the weight candidates of every weight search -- a (scale, zero
// point) per candidate and output channel).  k_pack_kfast fetches the two parameters of every (candidate, row) from global
// memory inside the candidate loop; even fetched one candidate ahead that is an L2 round trip per iteration, and the fc2
// weight candidates (128 x 384 x 1536 bf16 = 151 MB) were written at 1.05 TB/s.  Here a block owns 256 consecutive k-quads
// (at most RT rows), stages {1/s, s, -z, qmax - z} of those rows for ALL candidates in LDS once, and its loop over the
// candidates touches global memory only to store.  The tie zone is tested once per quad, and bf16 comes from the top half of
// the fp32 (the values are small integers: exact).
constexpr int TAB_ROWS = 6;
template <typename T>
__global__ __launch_bounds__(256) void k_pack_uniform_tab(PackArgs a) {
    constexpr bool FP8OUT = std::is_same<T, fp8_t>::value;
    extern __shared__ __attribute__((aligned(16))) unsigned char s_raw[];
    float4* s_tab = reinterpret_cast<float4*>(s_raw);                       // [rows of this block][C]
    const int64_t nq = a.Kp >> 2;
    const int64_t total = a.G * a.R * nq;
    const int64_t q0 = (int64_t)blockIdx.x * 256;
    const int64_t row0 = q0 / nq;                                            // flattened (g, r) row
    const int64_t rowN = min((q0 + 255) / nq, a.G * a.R - 1);
    const int nrows = (int)(rowN - row0 + 1);
    const int C = (int)a.C;
    for (int e = threadIdx.x; e < nrows * C; e += 256) {
        const int rl = e / C, c = e - rl * C;
        const int64_t gr = row0 + rl, g = gr / a.R, r = gr - g * a.R;
        const int64_t pidx = c * a.pc + (g % a.gmod) * a.pg + r * a.pr;
        const float sc = a.scale[pidx], z = rintf(a.zp[pidx]);
        s_tab[e] = make_float4(__builtin_amdgcn_rcpf(sc), sc, -z, a.qmax - z);
    }
    __syncthreads();
    const int64_t idx = q0 + threadIdx.x;
    if (idx >= total) return;
    const int64_t gr = idx / nq, kq = idx - gr * nq;
    const int64_t g = gr / a.R, r = gr - g * a.R;
    const int64_t k0 = kq << 2;
    const float* xp = a.x + g * a.sxg + r * a.sxr + k0;
    float xv[4];
    const bool full = k0 + 3 < a.K;
    if (full && ((uintptr_t)xp & 15) == 0) {
        const float4 v = *reinterpret_cast<const float4*>(xp);
        xv[0] = v.x; xv[1] = v.y; xv[2] = v.z; xv[3] = v.w;
    } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) xv[e] = (k0 + e < a.K) ? xp[e] : 0.0f;
    }
    const float4* tab = s_tab + (int)(gr - row0) * C;
    for (int64_t c = blockIdx.y; c < a.C; c += gridDim.y) {
        const float4 pr = tab[c];
        float t[4], k[4], dm = 0.0f;
#pragma unroll
        for (int e = 0; e < 4; ++e) { t[e] = xv[e] * pr.x; k[e] = rintf(t[e]); dm = fmaxf(dm, fabsf(t[e] - k[e])); }
        if (__builtin_expect(dm > 0.4999f, 0)) {                             // tie zone (see k_pack_kfast): the IEEE quotient decides
#pragma unroll
            for (int e = 0; e < 4; ++e) k[e] = rintf(xv[e] / pr.y);
        }
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            v[e] = __builtin_amdgcn_fmed3f(k[e], pr.z, pr.w);                // clamp(k + z, 0, qmax) - z
            if (!full && !(k0 + e < a.K)) v[e] = 0.0f;
        }
        const int64_t orow = a.c_inner ? (g * a.R + r) * a.C + c : (c * a.G + g) * a.R + r;
        T* op = reinterpret_cast<T*>(a.out) + orow * a.Kp + k0;
        if (FP8OUT) {
            int pk = __builtin_amdgcn_cvt_pk_fp8_f32(v[0], v[1], 0, false);
            pk = __builtin_amdgcn_cvt_pk_fp8_f32(v[2], v[3], pk, true);
            *reinterpret_cast<int*>(op) = pk;
        } else if (sizeof(T) == 1) {
            unsigned pk = 0;
#pragma unroll
            for (int e = 0; e < 4; ++e) pk = __builtin_amdgcn_cvt_pk_u8_f32(v[e] + 128.0f, e, pk);
            *reinterpret_cast<unsigned*>(op) = pk ^ 0x80808080u;
        } else if (sizeof(T) == 2) {
            const unsigned b0 = __float_as_uint(v[0]), b1 = __float_as_uint(v[1]), b2 = __float_as_uint(v[2]), b3 = __float_as_uint(v[3]);
            *reinterpret_cast<uint2*>(op) = make_uint2((b0 >> 16) | (b1 & 0xffff0000u), (b2 >> 16) | (b3 & 0xffff0000u));
        } else {
            *reinterpret_cast<float4*>(op) = make_float4(v[0], v[1], v[2], v[3]);
        }
        if (a.rowsum) {
            // one atomic per wavefront when all 64 lanes work on the same row (the common case: Kp/4 >= 64)
            int isum = (int)((v[0] + v[1]) + (v[2] + v[3]));                 // small integers: exact in fp32
            const int64_t ridx = (c * a.G + g) * a.R + r;
            const int64_t first = __builtin_amdgcn_readfirstlane((int)ridx);
            if (__all((int)ridx == (int)first) && __popcll(__ballot(1)) == 64) {
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) isum += __shfl_xor(isum, o);
                if ((threadIdx.x & 63) == 0 && isum != 0) atomicAdd(a.rowsum + ridx, isum);
            } else if (isum != 0) {
                atomicAdd(a.rowsum + ridx, isum);
            }
        }
    }
}

// The same packer for ONE-BYTE outputs (fp8 / int8) with 16 consecutive k per thread: one 16-byte store per lane and candidate
// (1 KiB per wave-store; the 4-byte stores of k_pack_uniform_tab wrote the fc2 weight candidates -- 75 MB of fp8 per launch -- at
// 1.4 TB/s).  Needs K % 16 == 0 == Kp % 16, 16-byte aligned source rows and Kp / 16 >= 64 (a wave then spans at most two rows: the
// row sums are reduced per row inside the wave).
template <typename T>
__global__ __launch_bounds__(256) void k_pack_uniform_tab16(PackArgs a) {
    constexpr bool FP8OUT = std::is_same<T, fp8_t>::value;
    static_assert(sizeof(T) == 1, "one-byte outputs");
    extern __shared__ __attribute__((aligned(16))) unsigned char s_raw[];
    float4* s_tab = reinterpret_cast<float4*>(s_raw);                       // [rows of this block][C]
    const int64_t nq = a.Kp >> 4;                                            // 16-element groups per row
    const int64_t total = a.G * a.R * nq;
    const int64_t q0 = (int64_t)blockIdx.x * 256;
    const int64_t row0 = q0 / nq;
    const int64_t rowN = min((q0 + 255) / nq, a.G * a.R - 1);
    const int nrows = (int)(rowN - row0 + 1);
    const int C = (int)a.C;
    for (int e = threadIdx.x; e < nrows * C; e += 256) {
        const int rl = e / C, c = e - rl * C;
        const int64_t gr = row0 + rl, g = gr / a.R, r = gr - g * a.R;
        const int64_t pidx = c * a.pc + (g % a.gmod) * a.pg + r * a.pr;
        const float sc = a.scale[pidx], z = rintf(a.zp[pidx]);
        s_tab[e] = make_float4(__builtin_amdgcn_rcpf(sc), sc, -z, a.qmax - z);
    }
    __syncthreads();
    const int64_t idx = q0 + threadIdx.x;
    const bool live = idx < total;
    const int64_t gr = live ? idx / nq : row0, kq = live ? idx - gr * nq : 0;
    const int64_t g = gr / a.R, r = gr - g * a.R;
    const int64_t k0 = kq << 4;
    float xv[16];
    const bool in_k = live && k0 < a.K;                                      // (K % 16 == 0: a group is all valid or all padding)
    if (in_k) {
        const float4* xp = reinterpret_cast<const float4*>(a.x + g * a.sxg + r * a.sxr + k0);
#pragma unroll
        for (int j = 0; j < 4; ++j) { const float4 v = xp[j]; xv[4 * j] = v.x; xv[4 * j + 1] = v.y; xv[4 * j + 2] = v.z; xv[4 * j + 3] = v.w; }
    } else {
#pragma unroll
        for (int e = 0; e < 16; ++e) xv[e] = 0.0f;
    }
    const float4* tab = s_tab + (int)(gr - row0) * C;
    for (int64_t c = blockIdx.y; c < a.C; c += gridDim.y) {
        const float4 pr = tab[c];
        float k[16], dm = 0.0f;
#pragma unroll
        for (int e = 0; e < 16; ++e) { const float t = xv[e] * pr.x; k[e] = rintf(t); dm = fmaxf(dm, fabsf(t - k[e])); }
        if (__builtin_expect(dm > 0.4999f, 0)) {                             // tie zone (see k_pack_kfast): the IEEE quotient decides
#pragma unroll
            for (int e = 0; e < 16; ++e) k[e] = rintf(xv[e] / pr.y);
        }
        float v[16];
        float fs = 0.0f;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            v[e] = in_k ? __builtin_amdgcn_fmed3f(k[e], pr.z, pr.w) : 0.0f;   // clamp(k + z, 0, qmax) - z
            fs += v[e];                                                      // small integers: exact in fp32
        }
        if (live) {
            const int64_t orow = a.c_inner ? (g * a.R + r) * a.C + c : (c * a.G + g) * a.R + r;
            T* op = reinterpret_cast<T*>(a.out) + orow * a.Kp + k0;
            uint4 o;
            unsigned* ow = reinterpret_cast<unsigned*>(&o);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (FP8OUT) {
                    int pk = __builtin_amdgcn_cvt_pk_fp8_f32(v[4 * j], v[4 * j + 1], 0, false);
                    pk = __builtin_amdgcn_cvt_pk_fp8_f32(v[4 * j + 2], v[4 * j + 3], pk, true);
                    ow[j] = (unsigned)pk;
                } else {
                    unsigned pk = 0;
#pragma unroll
                    for (int e = 0; e < 4; ++e) pk = __builtin_amdgcn_cvt_pk_u8_f32(v[4 * j + e] + 128.0f, e, pk);
                    ow[j] = pk ^ 0x80808080u;
                }
            }
            *reinterpret_cast<uint4*>(op) = o;
        }
        if (a.rowsum) {
            // a wave spans at most two rows (Kp / 16 >= 64): reduce each row's lanes, one atomic per (row, wave)
            const int ridx = live ? (int)((c * a.G + g) * a.R + r) : -1;
            const int first = __builtin_amdgcn_readfirstlane(ridx);
            const bool mine = ridx == first;
            int s1 = (live && mine) ? (int)fs : 0, s2 = (live && !mine) ? (int)fs : 0;
#pragma unroll
            for (int o2 = 32; o2 > 0; o2 >>= 1) { s1 += __shfl_xor(s1, o2); s2 += __shfl_xor(s2, o2); }
            const unsigned long long other = __ballot(live && !mine);
            if ((threadIdx.x & 63) == 0 && first >= 0 && s1 != 0) atomicAdd(a.rowsum + first, s1);
            if (other) {
                const int lane = threadIdx.x & 63, ol = __ffsll((long long)other) - 1;
                const int r2 = __shfl(ridx, ol);
                if (lane == ol && s2 != 0) atomicAdd(a.rowsum + r2, s2);
            }
        }
    }
}

// quant_forward of softmax . v (reference utils/wrap_net.py:26-30 + quant_layers/matmul.py:43-45): (scores * scale).softmax(-1) and the
// AdaLog quantisation of the probabilities in ONE pass -- the probabilities are never written; the output is the packed bf16
// operand of the product (what k_pack_adalog_fast writes for C = 1).  A wavefront owns a row of S <= 256 scores.  The arithmetic is
// ATen's softmax_warp_forward, operation for operation (element k = lane + 64 it; max; e = exp(x - max) summed per lane in `it`
// order, then the xor butterfly 32, 16, .., 1; e / sum), so that the composed route (torch softmax, then the packer) and this
// kernel quantise the same fp32 probabilities.
struct SoftmaxPackArgs {
    const float* x; int64_t rows; int S; float mul;      // scores [rows][S] (contiguous), multiplied by `mul` first
    const float* scale; const float* qv; const float* mant; int levels2;   // the AdaLog quantiser: device scalars (scale, q), 37 numerators
    unsigned short* out; int64_t Kp;                     // bf16 bits [rows][Kp], zero beyond S
};
constexpr int SM_ROWS = 4;
__global__ __launch_bounds__(256) void k_softmax_adalog_pack(SoftmaxPackArgs a) {
    __shared__ unsigned short s_lut[258];
    const int lw = a.levels2 + 2;
    const float qf = a.qv[0], sc = a.scale[0];
    for (int k = threadIdx.x; k < lw; k += blockDim.x) {
        const int kqv = k * (int)qf;
        const int t = kqv / ADALOG_R, j = kqv - t * ADALOG_R;
        const float v = (k >= a.levels2 || t > 100) ? 0.0f : ldexpf(a.mant[j], -t);
        s_lut[k] = (unsigned short)(__float_as_uint(v) >> 16);
    }
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const float inv_s = __builtin_amdgcn_rcpf(sc), rq37 = 37.0f / qf;
    // a wavefront takes SM_ROWS consecutive rows (the table above is built once per 4 SM_ROWS rows); their loads are issued together
    const int64_t row0 = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * SM_ROWS;
    float raw[SM_ROWS][4];
#pragma unroll
    for (int rr = 0; rr < SM_ROWS; ++rr) {
        const bool rv = row0 + rr < a.rows;
        const float* xr = a.x + (rv ? row0 + rr : 0) * a.S;
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int k = lane + 64 * it;
            raw[rr][it] = (rv && k < a.S) ? xr[k] : 0.0f;
        }
    }
#pragma unroll
    for (int rr = 0; rr < SM_ROWS; ++rr) {
        const int64_t row = row0 + rr;
        if (row >= a.rows) return;
        float el[4];
#pragma unroll
        for (int it = 0; it < 4; ++it) el[it] = lane + 64 * it < a.S ? raw[rr][it] * a.mul : -__builtin_inff();
        float mx = el[0];
#pragma unroll
        for (int it = 1; it < 4; ++it) mx = mx < el[it] ? el[it] : mx;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { const float b = __shfl_xor(mx, o); mx = mx < b ? b : mx; }
        float sum = 0.0f;
#pragma unroll
        for (int it = 0; it < 4; ++it) { el[it] = expf(el[it] - mx); sum += el[it]; }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) sum = sum + __shfl_xor(sum, o);
        unsigned short* orow = a.out + row * a.Kp;
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int k = lane + 64 * it;
            if (k >= a.Kp) break;
            unsigned short hv = 0;
            if (k < a.S) {
                const float pr = el[it] / sum;
                float kk = adalog_k_fast(pr, sc, inv_s, qf, rq37, true);
                kk = (kk == kk) ? fminf(fmaxf(kk, 0.0f), (float)(a.levels2 + 1)) : (float)(a.levels2 + 1);
                hv = s_lut[(int)kk];
            }
            orow[k] = hv;
        }
    }
}

// quant_forward of an attention block (reference utils/wrap_net.py:19-31 + quant_layers/matmul.py:43-45): the qkv projection's output
// [B][N][3][H][64] is split into heads, passed through the three per-head uniform input quantisers (q and k of q . k^T, v of
// softmax . v) and written as the packed operands of the two products in ONE pass -- no permuted fp32 copies of q / k / v, no three
// packer launches:  qp, kp int8 [B*H][N][128] (64 codes q - z, 64 zero bytes), vp bf16 [B*H][64][Np] (v transposed: a row per
// channel, a column per token, zero beyond N).  Codes as adalog_pack_uniform writes them: clamp(rne(x / s) + rne(z), 0, qmax) - rne(z).
struct AttnSplitArgs {
    const float* qkv; int B, N, H;
    const float* qs; const float* qz; const float* ks; const float* kz; const float* vs; const float* vz;    // [H] each (pg = 1) or [1] (pg = 0)
    int pg; float q_qmax, k_qmax, v_qmax;
    int8_t* qp; int8_t* kp; unsigned short* vp; int64_t Np;
};
__global__ __launch_bounds__(256) void k_attn_split_pack(AttnSplitArgs a) {
    __shared__ unsigned short vt[64][72];                       // [channel][token of the tile], rows padded against bank conflicts
    const int h = blockIdx.y, b = blockIdx.z, n0 = blockIdx.x * 64;
    const int t = threadIdx.x, row = t >> 2, qt = t & 3;       // token of the tile, 16-channel quarter
    const int n = n0 + row;
    const int HD = a.H * 64;
    const int64_t g = (int64_t)b * a.H + h;
    const int pi = a.pg ? h : 0;
    const bool live = n < a.N;
    const float* src = a.qkv + ((int64_t)b * a.N + (live ? n : 0)) * (3 * HD) + h * 64 + qt * 16;
    auto load16 = [&](const float* p_, float (&x)[16]) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float4 v = *reinterpret_cast<const float4*>(p_ + 4 * j);
            x[4 * j] = v.x; x[4 * j + 1] = v.y; x[4 * j + 2] = v.z; x[4 * j + 3] = v.w;
        }
    };
    auto codes = [&](const float (&x)[16], float s, float z, float qmax, float (&c)[16]) {
#pragma unroll
        for (int e = 0; e < 16; ++e) c[e] = fminf(fmaxf(rintf(x[e] / s) + z, 0.0f), qmax) - z;
    };
    float x[16], c[16];
    // q and k: one 16-byte store of codes and one of padding per thread and operand
    for (int which = 0; which < 2; ++which) {
        const float s = which ? a.ks[pi] : a.qs[pi], z = rintf(which ? a.kz[pi] : a.qz[pi]), qmax = which ? a.k_qmax : a.q_qmax;
        if (live) {
            load16(src + which * HD, x);
            codes(x, s, z, qmax, c);
            uint4 o;
            unsigned* ow = reinterpret_cast<unsigned*>(&o);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                unsigned pk = 0;
#pragma unroll
                for (int e = 0; e < 4; ++e) pk |= ((unsigned)(int)c[4 * j + e] & 0xffu) << (8 * e);
                ow[j] = pk;
            }
            int8_t* dst = (which ? a.kp : a.qp) + (g * a.N + n) * 128 + qt * 16;
            *reinterpret_cast<uint4*>(dst) = o;
            *reinterpret_cast<uint4*>(dst + 64) = make_uint4(0, 0, 0, 0);
        }
    }
    // v: quantise, transpose through LDS, rows of 64 tokens (128 bytes) out
    {
        const float s = a.vs[pi], z = rintf(a.vz[pi]);
        if (live) { load16(src + 2 * HD, x); codes(x, s, z, a.v_qmax, c); }
#pragma unroll
        for (int e = 0; e < 16; ++e) vt[qt * 16 + e][row] = live ? (unsigned short)(__float_as_uint(c[e]) >> 16) : (unsigned short)0;   // small integers: exact in bf16
        __syncthreads();
        const int ch = t >> 2, tq = t & 3;                       // channel, 16-token quarter
        if (n0 + tq * 16 < a.Np) {
            const uint4 lo = *reinterpret_cast<const uint4*>(&vt[ch][tq * 16]), hi = *reinterpret_cast<const uint4*>(&vt[ch][tq * 16 + 8]);
            unsigned short* dst = a.vp + (g * 64 + ch) * a.Np + n0 + tq * 16;
            *reinterpret_cast<uint4*>(dst) = lo;
            *reinterpret_cast<uint4*>(dst + 8) = hi;
        }
    }
}

template <typename T, int KIND>
int launch_pack(const PackArgs& a, hipStream_t st) {
    constexpr int EPT = Out<T>::EPT;
    if (a.sxk == 1) {
        const int64_t total = a.G * a.R * (a.Kp >> 2);
        int64_t gx = (total + 255) / 256;
        if (gx > 16384) gx = 16384;
        if (gx < 1) gx = 1;
        // enough candidate groups to fill the chip when the source is small (weights), all candidates per thread otherwise
        int64_t gy = 1;
        while (gx * gy < 2048 && gy < a.C) gy *= 2;
        static const int use_tab = getenv("ADALOG_PACK_TAB") ? atoi(getenv("ADALOG_PACK_TAB")) : 1;
        if (KIND == KIND_UNIFORM && a.pr != 0 && use_tab && (a.Kp >> 2) >= 64 && a.C * TAB_ROWS * sizeof(float4) <= 64 * 1024 &&
            !getenv("ADALOG_PACK_GENERIC")) {
            // per-row parameters (weight candidates): parameters staged in LDS per block, exact grid (no grid-stride loop)
            if constexpr (sizeof(T) == 1) {
                // one-byte outputs, long aligned rows: 16 k per thread, 16-byte stores
                static const int use16 = getenv("ADALOG_PACK_TAB16") ? atoi(getenv("ADALOG_PACK_TAB16")) : 1;
                if (use16 && (a.K & 15) == 0 && (a.Kp & 15) == 0 && (a.Kp >> 4) >= 64 && (a.sxr & 3) == 0 && (a.sxg & 3) == 0 &&
                    ((uintptr_t)a.x & 15) == 0 && ((uintptr_t)a.out & 15) == 0) {
                    const int64_t t16 = a.G * a.R * (a.Kp >> 4);
                    const int64_t bx16 = (t16 + 255) / 256;
                    int64_t by16 = 1;
                    while (bx16 * by16 < 1024 && by16 < a.C) by16 *= 2;
                    // a block of 256 threads covers 256 * 16 k: at most 256 / (Kp / 16) + 2 <= TAB_ROWS rows
                    if (bx16 < ((int64_t)1 << 31)) {
                        hipLaunchKernelGGL((k_pack_uniform_tab16<T>), dim3((unsigned)bx16, (unsigned)by16), dim3(256),
                                           (size_t)a.C * TAB_ROWS * sizeof(float4), st, a);
                        return 0;
                    }
                }
            }
            const int64_t bx = (total + 255) / 256;
            int64_t by = 1;
            while (bx * by < 1024 && by < a.C) by *= 2;
            if (bx < ((int64_t)1 << 31)) {
                hipLaunchKernelGGL((k_pack_uniform_tab<T>), dim3((unsigned)bx, (unsigned)by), dim3(256),
                                   (size_t)a.C * TAB_ROWS * sizeof(float4), st, a);
                return 0;
            }
        }
        if (KIND == KIND_UNIFORM && sizeof(T) == 1 && a.pr == 0 && (a.pg == 0 || a.G <= 65535) && !a.rowsum && a.C <= 2048 &&
            !getenv("ADALOG_PACK_GENERIC")) {
            // per-tensor parameters: one grid over all groups; per-group (per-head) parameters: grid.z = groups
            int64_t fx = gx, fy = gy, fz = 1;
            if (a.pg != 0) {
                fz = a.G;
                fx = (a.R * (a.Kp >> 2) + 255) / 256;
                if (fx > 16384) fx = 16384;
                fy = 1;
                while (fx * fy * fz < 2048 && fy < a.C) fy *= 2;
            }
            if (std::is_same<T, int8_t>::value)
                hipLaunchKernelGGL(k_pack_uniform_i8_fast<false>, dim3((unsigned)fx, (unsigned)fy, (unsigned)fz), dim3(256),
                                   (size_t)a.C * sizeof(float4), st, a);
            else
                hipLaunchKernelGGL(k_pack_uniform_i8_fast<true>, dim3((unsigned)fx, (unsigned)fy, (unsigned)fz), dim3(256),
                                   (size_t)a.C * sizeof(float4), st, a);
            return 0;
        }
        if (KIND == KIND_ADALOG && a.pg == 0 && !getenv("ADALOG_PACK_GENERIC")) {
            const size_t shm2 = (size_t)a.C * (sizeof(float4) + (size_t)(a.levels2 + 2) * sizeof(unsigned short));
            if (shm2 <= 64 * 1024) {
                hipLaunchKernelGGL(k_pack_adalog_fast, dim3((unsigned)gx, (unsigned)gy), dim3(256), shm2, st, a);
                return 0;
            }
        }
        const size_t shm = (KIND == KIND_ADALOG) ? (size_t)a.C * (a.levels2 + 1) * sizeof(unsigned short) : 0;
        if (KIND != KIND_ADALOG || (a.pg == 0 && shm <= 64 * 1024)) {
            hipLaunchKernelGGL((k_pack_kfast<T, KIND>), dim3((unsigned)gx, (unsigned)gy), dim3(256), shm, st, a);
            return 0;
        }
    }
    {
    const int64_t per_c = a.G * a.R * (a.Kp / EPT);
    int64_t gx = (per_c + 255) / 256;
    if (gx > 4096) gx = 4096;
    if (gx < 1) gx = 1;
    dim3 grid((unsigned)gx, (unsigned)a.C);
    const bool rfast = (a.sxr == 1 && a.sxk != 1);
    if (rfast)
        hipLaunchKernelGGL((k_pack<T, KIND, true>), grid, dim3(256), 0, st, a);
    else
        hipLaunchKernelGGL((k_pack<T, KIND, false>), grid, dim3(256), 0, st, a);
    }
    return 0;
}

}  // namespace

// out dtype codes shared with gemm_score: 0 = int8, 1 = bf16, 2 = fp32
extern "C" int adalog_pack_uniform(const float* x, int64_t G, int64_t R, int64_t K, int64_t sxg, int64_t sxr, int64_t sxk,
                                   const float* scale, const float* zero_point, int64_t C, int64_t pc, int64_t gmod,
                                   int64_t pg, int64_t pr, int n_bits, int out_dtype, void* out, int64_t Kp,
                                   int32_t* rowsum, int c_inner, void* stream) {
    ADALOG_ARG_CHECK(x && scale && zero_point && out, "pack_uniform: null pointer");
    ADALOG_ARG_CHECK(G >= 1 && R >= 1 && K >= 1 && C >= 1 && C <= 65535 && gmod >= 1, "pack_uniform: bad sizes");
    ADALOG_ARG_CHECK(out_dtype != 3 || n_bits <= 4, "pack_uniform: fp8 output holds q - z exactly only for n_bits <= 4");
    {
        const int64_t row_bytes = Kp * ((out_dtype == 0 || out_dtype == 3) ? 1 : out_dtype == 1 ? 2 : 4);
        // 32-byte rows: int8 / fp8 operands of K <= 32 for the window kernel only (adalog_gemm_score checks its side)
        ADALOG_ARG_CHECK(Kp >= K && (row_bytes % 64 == 0 || (row_bytes == 32 && (out_dtype == 0 || out_dtype == 3))),
                         "pack_uniform: Kp must cover K and be a multiple of 64 bytes (128 for anything but the search kernels)");
    }
    ADALOG_ARG_CHECK(n_bits >= 2 && n_bits <= 7, "pack_uniform: n_bits must be in [2,7] (q - z must fit int8)");
    hipStream_t st = (hipStream_t)stream;
    PackArgs a{};
    a.x = x; a.G = G; a.R = R; a.K = K; a.sxg = sxg; a.sxr = sxr; a.sxk = sxk;
    a.scale = scale; a.zp = zero_point; a.C = C; a.pc = pc; a.gmod = gmod; a.pg = pg; a.pr = pr;
    a.qmax = (float)((1 << n_bits) - 1); a.levels2 = 1 << n_bits; a.out = out; a.Kp = Kp; a.rowsum = rowsum; a.c_inner = c_inner;
    if (rowsum) {
        hipError_t e = hipMemsetAsync(rowsum, 0, sizeof(int32_t) * C * G * R, st);
        if (e != hipSuccess) { adalog_set_error("pack_uniform/memset", e); return (int)e; }
    }
    if (out_dtype == 0) launch_pack<int8_t, KIND_UNIFORM>(a, st);
    else if (out_dtype == 1) launch_pack<__hip_bfloat16, KIND_UNIFORM>(a, st);
    else if (out_dtype == 2) launch_pack<float, KIND_UNIFORM>(a, st);
    else if (out_dtype == 3) launch_pack<fp8_t, KIND_UNIFORM>(a, st);
    else { adalog_set_error_msg("pack_uniform: unknown out_dtype"); return -1; }
    ADALOG_LAUNCH_CHECK("adalog_pack_uniform");
    return 0;
}

extern "C" int adalog_pack_adalog_bf16_pre(const float* x, int64_t G, int64_t R, int64_t K, int64_t sxg, int64_t sxr,
                                           int64_t sxk, const float* scale, const float* qv, int64_t C, int64_t pc,
                                           int64_t gmod, int64_t pg, int n_bits, const float* mant37, const float* shift,
                                           int clamp_u, void* out, int64_t Kp, int c_inner, int pre, void* stream);
extern "C" int adalog_pack_adalog_bf16(const float* x, int64_t G, int64_t R, int64_t K, int64_t sxg, int64_t sxr,
                                       int64_t sxk, const float* scale, const float* qv, int64_t C, int64_t pc,
                                       int64_t gmod, int64_t pg, int n_bits, const float* mant37, const float* shift,
                                       int clamp_u, void* out, int64_t Kp, int c_inner, void* stream) {
    return adalog_pack_adalog_bf16_pre(x, G, R, K, sxg, sxr, sxk, scale, qv, C, pc, gmod, pg, n_bits, mant37, shift, clamp_u, out, Kp,
                                       c_inner, 0, stream);
}
// pre = 1: the operand is GELU(x) -- the activation function between fc1 and fc2 (reference timm Mlp / quant_layers/linear.py:770-796
// quant_forward) applied in the packer's loader, so that quant_forward of the MLP runs no separate GELU pass.  Per-tensor scale, unit
// stride along K only (the fast kernel); anything else is refused.
extern "C" int adalog_pack_adalog_bf16_pre(const float* x, int64_t G, int64_t R, int64_t K, int64_t sxg, int64_t sxr,
                                           int64_t sxk, const float* scale, const float* qv, int64_t C, int64_t pc,
                                           int64_t gmod, int64_t pg, int n_bits, const float* mant37, const float* shift,
                                           int clamp_u, void* out, int64_t Kp, int c_inner, int pre, void* stream) {
    ADALOG_ARG_CHECK(pre == 0 || (pre == 1 && pg == 0 && sxk == 1 && !getenv("ADALOG_PACK_GENERIC")), "pack_adalog: the GELU prologue needs a per-tensor scale and unit K stride");
    ADALOG_ARG_CHECK(x && scale && qv && mant37 && out, "pack_adalog: null pointer");
    ADALOG_ARG_CHECK(G >= 1 && R >= 1 && K >= 1 && C >= 1 && C <= 65535 && gmod >= 1, "pack_adalog: bad sizes");
    ADALOG_ARG_CHECK(Kp >= K && (Kp * 2) % 64 == 0, "pack_adalog: Kp must cover K and be a multiple of 32 elements");
    ADALOG_ARG_CHECK(n_bits >= 2 && n_bits <= 7, "pack_adalog: n_bits must be in [2,7] (numerators must fit bf16)");
    PackArgs a{};
    a.x = x; a.G = G; a.R = R; a.K = K; a.sxg = sxg; a.sxr = sxr; a.sxk = sxk;
    a.scale = scale; a.qv = qv; a.C = C; a.pc = pc; a.gmod = gmod; a.pg = pg; a.pr = 0;
    a.levels2 = 1 << n_bits; a.mant = mant37; a.shift = shift; a.clamp_u = clamp_u; a.out = out; a.Kp = Kp;
    a.c_inner = c_inner; a.pre = pre;
    ADALOG_ARG_CHECK(pre == 0 || (size_t)C * (sizeof(float4) + (size_t)(a.levels2 + 2) * sizeof(unsigned short)) <= 64 * 1024, "pack_adalog: the GELU prologue runs on the fast kernel only");
    launch_pack<__hip_bfloat16, KIND_ADALOG>(a, (hipStream_t)stream);
    ADALOG_LAUNCH_CHECK("adalog_pack_adalog_bf16");
    return 0;
}

extern "C" int adalog_pack_raw_f32(const float* x, int64_t G, int64_t R, int64_t K, int64_t sxg, int64_t sxr, int64_t sxk,
                                   void* out, int64_t Kp, void* stream) {
    ADALOG_ARG_CHECK(x && out && G >= 1 && R >= 1 && K >= 1, "pack_raw: bad arguments");
    ADALOG_ARG_CHECK(Kp >= K && Kp % 32 == 0, "pack_raw: Kp must cover K and be a multiple of 32 elements");
    PackArgs a{};
    a.x = x; a.G = G; a.R = R; a.K = K; a.sxg = sxg; a.sxr = sxr; a.sxk = sxk; a.C = 1; a.gmod = 1;
    a.out = out; a.Kp = Kp;
    launch_pack<float, KIND_RAW>(a, (hipStream_t)stream);
    ADALOG_LAUNCH_CHECK("adalog_pack_raw_f32");
    return 0;
}

// ---- three-term bf16 image of an UNQUANTISED fp32 operand (patch-embedding weight search: the input keeps 8 bits or more and is
// scored as it is, conv.py:226-255).  x = hi + mid + lo exactly (8 + 8 + 8 mantissa bits, each residual is exact in fp32), stored
// as [row][hi(0..Kt) | mid(0..Kt) | lo(0..Kt)]; against a candidate operand of exact small integers repeated three times along K
// the bf16 MFMA then accumulates the same fp32 products as the fp32 MFMA does, at 5x its rate.
namespace {
__device__ __forceinline__ unsigned bf16_rne(float v) {
    unsigned u = __float_as_uint(v);
    u += 0x7fffu + ((u >> 16) & 1u);
    return u >> 16;
}
__global__ __launch_bounds__(256) void k_pack_split3(const float* __restrict__ x, int64_t R, int64_t K, int64_t sxg, int64_t sxr, int64_t sxk,
                                                    uint16_t* __restrict__ out, int64_t Kt) {
    const int64_t g = blockIdx.z;
    for (int64_t row = blockIdx.y; row < R; row += gridDim.y) {
    uint16_t* o = out + (g * R + row) * 3 * Kt;
    for (int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x; k < Kt; k += (int64_t)gridDim.x * 256) {
        unsigned h = 0, m = 0, l = 0;
        if (k < K) {
            const float v = x[g * sxg + row * sxr + k * sxk];
            h = bf16_rne(v);
            const float r1 = v - __uint_as_float(h << 16);
            m = bf16_rne(r1);
            const float r2 = r1 - __uint_as_float(m << 16);
            l = bf16_rne(r2);
        }
        o[k] = (uint16_t)h; o[Kt + k] = (uint16_t)m; o[2 * Kt + k] = (uint16_t)l;
    }
    }
}
}  // namespace

extern "C" int adalog_pack_split3_bf16(const float* x, int64_t G, int64_t R, int64_t K, int64_t sxg, int64_t sxr, int64_t sxk,
                                       void* out, int64_t Kt, void* stream) {
    ADALOG_ARG_CHECK(x && out && G >= 1 && R >= 1 && K >= 1, "pack_split3: bad arguments");
    ADALOG_ARG_CHECK(Kt >= K && Kt % 32 == 0, "pack_split3: Kt must cover K and be a multiple of 32 elements");
    ADALOG_ARG_CHECK(G < 65536, "pack_split3: too many groups for one launch");
    const unsigned bx = (unsigned)((Kt + 255) / 256);
    hipLaunchKernelGGL(k_pack_split3, dim3(bx, (unsigned)(R < 65535 ? R : 65535), (unsigned)G), dim3(256), 0, (hipStream_t)stream, x, R, K, sxg, sxr, sxk,
                       (uint16_t*)out, Kt);
    ADALOG_LAUNCH_CHECK("adalog_pack_split3_bf16");
    return 0;
}

// (x * mul).softmax(-1) quantised by the post-softmax AdaLog quantiser (per-tensor scale, log base 2^(-q/37), u clamped to
// [1e-15, 1]: reference quant_layers/matmul.py:337-343 with the searched q) straight into the bf16 operand image [rows][Kp] of
// the softmax . v product.  x: fp32 [rows][S] contiguous, S <= 256; Kp: a multiple of 32 elements covering S, <= 256.
extern "C" int adalog_softmax_adalog_pack_bf16(const float* x, int64_t rows, int S, float mul, const float* scale, const float* qv,
                                               int n_bits, const float* mant37, void* out, int64_t Kp, void* stream) {
    if (rows == 0) return 0;
    ADALOG_ARG_CHECK(x && scale && qv && mant37 && out && rows > 0, "softmax_adalog_pack: null pointer");
    ADALOG_ARG_CHECK(S >= 1 && S <= 256 && Kp >= S && Kp <= 256 && (Kp * 2) % 64 == 0, "softmax_adalog_pack: 1 <= S <= Kp <= 256, Kp a multiple of 32");
    ADALOG_ARG_CHECK(n_bits >= 2 && n_bits <= 7, "softmax_adalog_pack: n_bits must be in [2,7]");
    SoftmaxPackArgs a{x, rows, S, mul, scale, qv, mant37, 1 << n_bits, reinterpret_cast<unsigned short*>(out), Kp};
    adalog_note_kernel("k_softmax_adalog_pack");
    hipLaunchKernelGGL(k_softmax_adalog_pack, dim3((unsigned)((rows + 4 * SM_ROWS - 1) / (4 * SM_ROWS))), dim3(256), 0, (hipStream_t)stream, a);
    ADALOG_LAUNCH_CHECK("adalog_softmax_adalog_pack_bf16");
    return 0;
}

// The three operand packs of an attention block's quant_forward in one launch (k_attn_split_pack): qkv fp32 [B][N][3][H][64]
// (contiguous, 16-byte aligned); (scale, zero point) of the three uniform quantisers per head (pg = 1: [H]) or per tensor (pg = 0);
// qp, kp int8 [B*H][N][128]; vp bf16 [B*H][64][Np], Np a multiple of 64 covering N.  Head dimension 64 only.
extern "C" int adalog_attn_split_pack(const float* qkv, int B, int N, int H, const float* q_scale, const float* q_zp, int q_bits,
                                      const float* k_scale, const float* k_zp, int k_bits, const float* v_scale, const float* v_zp,
                                      int v_bits, int pg, void* qp, void* kp, void* vp, int64_t Np, void* stream) {
    if (B == 0 || N == 0) return 0;
    ADALOG_ARG_CHECK(qkv && q_scale && q_zp && k_scale && k_zp && v_scale && v_zp && qp && kp && vp, "attn_split_pack: null pointer");
    ADALOG_ARG_CHECK(B >= 1 && N >= 1 && H >= 1 && H <= 65535 && B <= 65535 && Np >= N && Np % 64 == 0, "attn_split_pack: bad sizes (Np: a multiple of 64 covering N)");
    ADALOG_ARG_CHECK(q_bits >= 2 && q_bits <= 7 && k_bits >= 2 && k_bits <= 7 && v_bits >= 2 && v_bits <= 7, "attn_split_pack: n_bits must be in [2,7]");
    ADALOG_ARG_CHECK(((((uintptr_t)qkv) | ((uintptr_t)qp) | ((uintptr_t)kp) | ((uintptr_t)vp)) & 15) == 0, "attn_split_pack: 16-byte aligned buffers");
    AttnSplitArgs a{qkv, B, N, H, q_scale, q_zp, k_scale, k_zp, v_scale, v_zp, pg ? 1 : 0, (float)((1 << q_bits) - 1), (float)((1 << k_bits) - 1),
                    (float)((1 << v_bits) - 1), reinterpret_cast<int8_t*>(qp), reinterpret_cast<int8_t*>(kp),
                    reinterpret_cast<unsigned short*>(vp), Np};
    adalog_note_kernel("k_attn_split_pack");
    hipLaunchKernelGGL(k_attn_split_pack, dim3((unsigned)(Np / 64), (unsigned)H, (unsigned)B), dim3(256), 0, (hipStream_t)stream, a);
    ADALOG_LAUNCH_CHECK("adalog_attn_split_pack");
    return 0;
}
