// Shared device/host helpers for the AdaLog MI355X (gfx950) kernels.
// Compiled with hipcc --offload-arch=gfx950 -ffp-contract=off: every fp32 step below must round exactly
// like the reference's ATen CPU op (IEEE divide, round-half-even, no FMA contraction).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#define ADALOG_R 37            // AdaLog fixed denominator r (reference quantizers/logarithm.py:71)

extern "C" void adalog_set_error(const char* where, hipError_t e);
extern "C" void adalog_set_error_msg(const char* msg);
extern "C" void adalog_note_kernel(const char* name);
extern "C" unsigned int* adalog_ticket_slot(void);      // brecq.hip: a zeroed device word for "last block finishes" kernels
extern "C" unsigned int* adalog_ticket_slots(int n);    // ... n <= 64 consecutive zeroed words
extern "C" unsigned int* adalog_ticket_slots_on(int n, void* stream);   // ... from the sub-ring of that stream

#define ADALOG_LAUNCH_CHECK(name)                                   \
    do {                                                            \
        hipError_t e__ = hipGetLastError();                         \
        if (e__ != hipSuccess) { adalog_set_error(name, e__); return (int)e__; } \
    } while (0)

#define ADALOG_ARG_CHECK(cond, msg)                                 \
    do { if (!(cond)) { adalog_set_error_msg(msg); return -1; } } while (0)

static inline int cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

// Raise a kernel's dynamic-LDS limit once PER DEVICE (the attribute belongs to the function's code object on one device: a
// per-process "done" flag would leave the second device of a process without it).  ``done_mask``: a static per call site, one bit
// per device ordinal.  Returns the error of hipFuncSetAttribute so that the caller fails instead of launching with 64 KiB.
static inline hipError_t adalog_max_lds(const void* fn, int bytes, unsigned long long* done_mask) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    const unsigned long long bit = 1ull << (dev & 63);
    if (*done_mask & bit) return hipSuccess;
    e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e == hipSuccess) *done_mask |= bit;
    return e;
}

// compute units of the CURRENT device (cached per device ordinal, not per process)
static inline int adalog_device_cus() {
    static int cus[64] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 256;
    int& c = cus[dev & 63];
    if (c == 0) {
        hipDeviceProp_t pr;
        c = (hipGetDeviceProperties(&pr, dev) == hipSuccess && pr.multiProcessorCount > 0) ? pr.multiProcessorCount : 256;
    }
    return c;
}

// ---------------------------------------------------------------------------------------------- device math
// asymmetric uniform bin: clamp(rne(x / s) + z, 0, qmax)       (uniform.py:29-35; z already rounded)
__device__ __forceinline__ float uni_bin(float x, float s, float z, float qmax) {
    float xi = rintf(x / s);
    return fminf(fmaxf(xi + z, 0.0f), qmax);
}

// k = rne( (-log2(u)) * 37 / q ) with u a positive float, evaluated so that it equals the fp32 pipeline
//   L = fl32(log2 u) correctly rounded;  t = fl32(fl32(-L * 37) / q);  k = rne(t)
// (logarithm.py:94 with a correctly-rounded log2; torch's SLEEF log2 differs from correct rounding on <1e-3 of
// inputs by one ulp, which moves k only for t within one ulp of a tie: ~1e-9 of elements).
// Fast path: hardware v_log_f32 (1 ulp) unless t lands within 1e-3 of a tie; then the exact path in fp64.
__device__ __forceinline__ float adalog_k(float u, float qf) {
    float lf = __log2f(u);
    float t = (-lf) * 37.0f / qf;
    float k = rintf(t);
    float fr = fabsf(t - k);
    if (__builtin_expect(fr > 0.499f || !(t < 3.0e38f), 0)) {
        float le = (float)log2((double)u);
        t = (-le) * 37.0f / qf;
        k = rintf(t);
    }
    return k;
}

// pk.byte[pos] = rne(tc) as a two's-complement int8 (|tc| <= 127; pos a compile-time constant after inlining): ONE SDWA add of
// 1.5 * 2^23 rounds to nearest-even (the sum's ulp is 1), and the low byte of the sum's bit pattern IS the integer -- rounding,
// conversion and the byte insert of a generated int8 MFMA operand in one VALU instruction (round 5; gram.hip, gram_act.hip).
__device__ __forceinline__ int sdwa_rne_byte(int pk, float tc, float magic /* 12582912.0f */, int pos) {
    if (pos == 0) asm("v_add_f32_sdwa %0, %1, %2 dst_sel:BYTE_0 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD" : "+v"(pk) : "v"(tc), "v"(magic));
    else if (pos == 1) asm("v_add_f32_sdwa %0, %1, %2 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD" : "+v"(pk) : "v"(tc), "v"(magic));
    else if (pos == 2) asm("v_add_f32_sdwa %0, %1, %2 dst_sel:BYTE_2 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD" : "+v"(pk) : "v"(tc), "v"(magic));
    else asm("v_add_f32_sdwa %0, %1, %2 dst_sel:BYTE_3 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD" : "+v"(pk) : "v"(tc), "v"(magic));
    return pk;
}

// ---- fast forms used by the operand-packing kernels (hundreds of millions of elements per scoring call).
// Both evaluate with a reciprocal multiply first and fall back to the exact IEEE sequence above only when the result
// lands within 1e-3 of a rounding tie, so they return EXACTLY the same integer as the exact forms (the reciprocal path
// is accurate to ~3 ulp, i.e. < 1e-4 absolute at the magnitudes that survive the clamp).
__device__ __forceinline__ float uni_bin_fast(float x, float s, float inv_s, float z, float qmax) {
    float t = x * inv_s;
    float k = rintf(t);
    if (__builtin_expect(fabsf(t - k) > 0.499f && fabsf(t) < 4096.0f, 0)) k = rintf(x / s);
    return fminf(fmaxf(k + z, 0.0f), qmax);
}

// xs = x (+ shift); returns k = rne(-log2(clamp(xs/s)) * 37 / q) exactly as adalog_k(clamp(xs / s), q)
__device__ __forceinline__ float adalog_k_fast(float xs, float s, float inv_s, float qf, float rq37, bool clamp_u) {
    float u = xs * inv_s;
    if (clamp_u) u = fminf(fmaxf(u, 1e-15f), 1.0f);
    float t = -__log2f(u) * rq37;
    float k = rintf(t);
    if (__builtin_expect(fabsf(t - k) > 0.499f || !(t < 3.0e38f), 0)) {
        float ue = xs / s;
        if (clamp_u) ue = fminf(fmaxf(ue, 1e-15f), 1.0f);
        float le = (float)log2((double)ue);
        t = (-le) * 37.0f / qf;
        k = rintf(t);
    }
    return k;
}
