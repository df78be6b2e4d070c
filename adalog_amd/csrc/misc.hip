// Small vector kernels around the searches.
//   adalog_shift_fold      bias' = bias - shift * s_w * sum_i (q_w - z_w)     <- linear.py:999-1006 (reparam_bias) and
//                          the "- shift" term of every post-GELU search operand (linear.py:837,879,920)
//   adalog_minmax_rows     K4 per-row min/max of the weight                    <- linear.py:265-274
//   adalog_absminmax_cols  K4 min/max of |x| per tensor or per channel         <- linear.py:276-294
// Wavefront (64-lane) shuffles do the reductions; min/max are order-independent, so results are deterministic.
#include "common.h"
#include <float.h>

namespace {

__global__ __launch_bounds__(256) void k_shift_fold(const int32_t* __restrict__ rowsum, const float* __restrict__ ws,
                                                    const float* __restrict__ shift, const float* __restrict__ bias,
                                                    int C, int O, float* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= C * O) return;
    const float t = ws[i] * (float)rowsum[i];
    const float f = shift[0] * t;
    out[i] = (bias ? bias[i % O] : 0.0f) - f;
}

__device__ __forceinline__ float wave_min(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o));
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}

// one wavefront per row
__global__ __launch_bounds__(256) void k_minmax_rows(const float* __restrict__ w, int rows, int I, int use_abs,
                                                     float* __restrict__ mn, float* __restrict__ mx) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= rows) return;
    float lo = FLT_MAX, hi = -FLT_MAX;
    for (int i = lane; i < I; i += 64) {
        float v = w[(int64_t)row * I + i];
        if (use_abs) v = fabsf(v);
        lo = fminf(lo, v); hi = fmaxf(hi, v);
    }
    lo = wave_min(lo); hi = wave_max(hi);
    if (lane == 0) { mn[row] = lo; mx[row] = hi; }
}

// column-wise over |x|: block = 64 columns x 4 row groups; grid.y = row slabs; atomics on the bit pattern of
// non-negative floats (monotone as unsigned ints) keep the result order-independent
__global__ __launch_bounds__(256) void k_absminmax_cols(const float* __restrict__ x, int64_t rows, int I, int per_channel,
                                                        unsigned* __restrict__ mn, unsigned* __restrict__ mx) {
    const int col = blockIdx.x * 64 + (threadIdx.x & 63), rg = threadIdx.x >> 6;
    float lo = FLT_MAX, hi = 0.0f;
    if (col < I) {
        for (int64_t r = (int64_t)blockIdx.y * 4 + rg; r < rows; r += (int64_t)gridDim.y * 4) {
            const float v = fabsf(x[r * I + col]);
            lo = fminf(lo, v); hi = fmaxf(hi, v);
        }
    }
    if (!per_channel) { lo = wave_min(lo); hi = wave_max(hi); }
    if (per_channel ? (col < I) : ((threadIdx.x & 63) == 0)) {
        const int o = per_channel ? col : 0;
        atomicMin(mn + o, __float_as_uint(lo));
        atomicMax(mx + o, __float_as_uint(hi));
    }
}

}  // namespace

extern "C" int adalog_shift_fold(const int32_t* rowsum, const float* w_scale, const float* shift, const float* bias, int C,
                                 int O, float* out, void* stream) {
    ADALOG_ARG_CHECK(rowsum && w_scale && shift && out && C >= 1 && O >= 1, "shift_fold: bad arguments");
    hipLaunchKernelGGL(k_shift_fold, dim3(cdiv((int64_t)C * O, 256)), dim3(256), 0, (hipStream_t)stream, rowsum, w_scale, shift,
                       bias, C, O, out);
    ADALOG_LAUNCH_CHECK("adalog_shift_fold");
    return 0;
}

extern "C" int adalog_minmax_rows(const float* w, int rows, int I, int use_abs, float* mn, float* mx, void* stream) {
    ADALOG_ARG_CHECK(w && mn && mx && rows >= 1 && I >= 1, "minmax_rows: bad arguments");
    hipLaunchKernelGGL(k_minmax_rows, dim3(cdiv(rows, 4)), dim3(256), 0, (hipStream_t)stream, w, rows, I, use_abs, mn, mx);
    ADALOG_LAUNCH_CHECK("adalog_minmax_rows");
    return 0;
}

extern "C" int adalog_absminmax_cols(const float* x, int64_t rows, int I, int per_channel, float* mn, float* mx,
                                     void* stream) {
    ADALOG_ARG_CHECK(x && mn && mx && rows >= 1 && I >= 1, "absminmax_cols: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    const int n = per_channel ? I : 1;
    hipError_t e = hipMemsetAsync(mn, 0x7f, sizeof(float) * n, st);       // 0x7f7f7f7f: a huge positive float
    if (e == hipSuccess) e = hipMemsetAsync(mx, 0, sizeof(float) * n, st);
    if (e != hipSuccess) { adalog_set_error("absminmax_cols/memset", e); return (int)e; }
    int gy = (int)((rows + 255) / 256);
    if (gy > 1024) gy = 1024;
    if (gy < 1) gy = 1;
    hipLaunchKernelGGL(k_absminmax_cols, dim3(cdiv(I, 64), gy), dim3(256), 0, st, x, rows, I, per_channel, (unsigned*)mn,
                       (unsigned*)mx);
    ADALOG_LAUNCH_CHECK("adalog_absminmax_cols");
    return 0;
}
