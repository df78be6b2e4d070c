// K1/K2/K3 -- elementwise fake-quant kernels (HBM-bound: 4 B read + 4 B write [+1 B bin] per element).
//   adalog_uniform_fake_quant_f32   <- quantizers/uniform.py:25-36   (UniformQuantizer.forward, eval form)
//   adalog_log_fake_quant_f32       <- quantizers/logarithm.py:83-99 (AdaLogQuantizer.forward) and :127-135 (Shift*)
// One kernel per call instead of the reference's 5..12 ATen passes; scale / zero-point / LUTs are read from
// device memory (they are nn.Parameters / buffers), so no host sync is needed to launch.
#include "common.h"

namespace {

// ------------------------------------------------------------------------------------------------ uniform
// Layout A ("row"): x viewed as [rows][inner]; channel of a row = row % n_ch   (per-tensor: rows=1, n_ch=1;
//                   per-row weights: n_ch=rows; per-head [N,H,S,C]: rows=N*H, n_ch=H, inner=S*C)
template <bool VEC>
__global__ __launch_bounds__(256) void k_uniform_rows(const float* __restrict__ x, float* __restrict__ y,
                                                      uint8_t* __restrict__ bins, int64_t rows, int64_t inner,
                                                      const float* __restrict__ scale, const float* __restrict__ zp,
                                                      int64_t n_ch, float qmin, float qmax) {
    for (int64_t row = blockIdx.y; row < rows; row += gridDim.y) {
        const int64_t ch = row % n_ch;
        const float s = scale[ch];
        const float z = zp ? rintf(zp[ch]) : 0.0f;
        const float* xr = x + row * inner;
        float* yr = y ? y + row * inner : nullptr;
        uint8_t* br = bins ? bins + row * inner : nullptr;
        if (VEC) {
            const int64_t n4 = inner >> 2;
            for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
                float4 v = reinterpret_cast<const float4*>(xr)[i];
                float q0 = fminf(fmaxf(rintf(v.x / s) + z, qmin), qmax);
                float q1 = fminf(fmaxf(rintf(v.y / s) + z, qmin), qmax);
                float q2 = fminf(fmaxf(rintf(v.z / s) + z, qmin), qmax);
                float q3 = fminf(fmaxf(rintf(v.w / s) + z, qmin), qmax);
                if (yr) reinterpret_cast<float4*>(yr)[i] = make_float4((q0 - z) * s, (q1 - z) * s, (q2 - z) * s, (q3 - z) * s);
                if (br) {
                    uchar4 b = make_uchar4((uint8_t)(int)q0, (uint8_t)(int)q1, (uint8_t)(int)q2, (uint8_t)(int)q3);
                    reinterpret_cast<uchar4*>(br)[i] = b;
                }
            }
        } else {
            for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < inner; i += (int64_t)gridDim.x * blockDim.x) {
                float q = fminf(fmaxf(rintf(xr[i] / s) + z, qmin), qmax);
                if (yr) yr[i] = (q - z) * s;
                if (br) br[i] = (uint8_t)(int)q;
            }
        }
    }
}

// Layout B ("col"): x viewed as [rows][n_ch], channel = column (per-channel activations, scale shape [I])
__global__ __launch_bounds__(256) void k_uniform_cols(const float* __restrict__ x, float* __restrict__ y,
                                                      uint8_t* __restrict__ bins, int64_t rows, int64_t n_ch,
                                                      const float* __restrict__ scale, const float* __restrict__ zp,
                                                      float qmin, float qmax) {
    const int64_t total = rows * n_ch;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t ch = i % n_ch;
        const float s = scale[ch];
        const float z = zp ? rintf(zp[ch]) : 0.0f;
        float q = fminf(fmaxf(rintf(x[i] / s) + z, qmin), qmax);
        if (y) y[i] = (q - z) * s;
        if (bins) bins[i] = (uint8_t)(int)q;
    }
}

// ------------------------------------------------------------------------------------------------ AdaLog
// y = 2^-T1[k] * T2[k] * s * [k < 2L],  k = rne(-log2(clamp((x+shift)/s, 1e-15, 1)) * 37 / q)   [- shift]
template <bool VEC>
__global__ __launch_bounds__(256) void k_adalog(const float* __restrict__ x, float* __restrict__ y,
                                                uint8_t* __restrict__ bins, int64_t n,
                                                const float* __restrict__ scale, const int64_t* __restrict__ q,
                                                const float* __restrict__ t1, const float* __restrict__ t2,
                                                int levels2, const float* __restrict__ shift, int sub_shift,
                                                int train_form, int pre) {
    __shared__ float lut[256];
    const float s = scale[0];
    const float qf = (float)q[0];
    const float sh = shift ? shift[0] : 0.0f;
    for (int i = threadIdx.x; i < levels2; i += blockDim.x) {
        // (2 ** -T1[k]) * T2[k]: exact power-of-two scaling, same value torch computes (logarithm.py:97)
        lut[i] = train_form ? exp2f(-1.0f * (float)i * qf / 37.0f) : ldexpf(t2[i], -(int)t1[i]);
    }
    __syncthreads();
    const float kmax = (float)(levels2 - 1);
    // the bin through the reciprocal / v_log_f32 fast path of the operand packers (common.h adalog_k_fast: the exact IEEE sequence --
    // x / s, correctly rounded log2, / q -- decides whenever the fast value lands within 1e-3 of a rounding tie, so k is EXACTLY
    // adalog_k(clamp(xs / s), q)): the two IEEE divisions per element kept this kernel at 0.45 of the HBM rate (round 5)
    const float inv_s = 1.0f / s, rq37 = 37.0f / qf;
    auto one = [&](float v, float& out, uint8_t& b) {
        if (pre) v = (v * 0.5f) * (1.0f + erff(v * 0.70710678118654752440f));   // GELU (ATen's fp32 expression): fc2 reads fc1's output
        float xs = shift ? v + sh : v;
        float k = adalog_k_fast(xs, s, inv_s, qf, rq37, true);
        bool keep = k < (float)levels2;
        k = fminf(fmaxf(k, 0.0f), kmax);
        float r = lut[(int)k] * s;
        r = keep ? r : r * 0.0f;
        out = sub_shift ? r - sh : r;
        b = keep ? (uint8_t)(int)k : (uint8_t)255;
    };
    if (VEC) {
        const int64_t n4 = n >> 2;
        for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
            float4 v = reinterpret_cast<const float4*>(x)[i];
            float4 o; uchar4 b;
            one(v.x, o.x, b.x); one(v.y, o.y, b.y); one(v.z, o.z, b.z); one(v.w, o.w, b.w);
            if (y) reinterpret_cast<float4*>(y)[i] = o;
            if (bins) reinterpret_cast<uchar4*>(bins)[i] = b;
        }
    } else {
        for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
            float o; uint8_t b;
            one(x[i], o, b);
            if (y) y[i] = o;
            if (bins) bins[i] = b;
        }
    }
}

inline int grid_for(int64_t work_items) {
    int64_t b = (work_items + 255) / 256;
    if (b < 1) b = 1;
    if (b > 8192) b = 8192;   // 256 CUs x 32 resident blocks; grid-stride the rest
    return (int)b;
}

inline bool aligned16(const void* p) { return ((uintptr_t)p & 15) == 0; }

}  // namespace

extern "C" int adalog_uniform_fake_quant_f32(const float* x, float* y, uint8_t* bins, int64_t n, const float* scale,
                                             const float* zero_point, int64_t n_channels, int64_t inner, int n_bits,
                                             int symmetric, void* stream) {
    if (n == 0) return 0;
    ADALOG_ARG_CHECK(x && scale && n >= 0 && n_channels >= 1 && inner >= 1, "uniform_fake_quant: bad arguments");
    ADALOG_ARG_CHECK(symmetric || zero_point, "uniform_fake_quant: asymmetric needs zero_point");
    ADALOG_ARG_CHECK(n_bits >= 2 && n_bits <= 8, "uniform_fake_quant: n_bits must be in [2,8]");
    if (n == 0) return 0;
    const float L = (float)(1 << (n_bits - 1));
    // bins are stored biased to [0, 2L-1] in both modes (symmetric: q + L)
    const float qmin = symmetric ? -L : 0.0f, qmax = symmetric ? L - 1.0f : 2.0f * L - 1.0f;
    hipStream_t st = (hipStream_t)stream;
    if (symmetric) {
        ADALOG_ARG_CHECK(bins == nullptr, "uniform_fake_quant: bins output is defined for the asymmetric form only");
    }
    if (inner == 1 && n_channels > 1) {
        ADALOG_ARG_CHECK(n % n_channels == 0, "uniform_fake_quant: n not a multiple of n_channels");
        hipLaunchKernelGGL(k_uniform_cols, dim3(grid_for(n)), dim3(256), 0, st, x, y, bins, n / n_channels, n_channels,
                           scale, symmetric ? nullptr : zero_point, qmin, qmax);
    } else {
        ADALOG_ARG_CHECK(n % inner == 0, "uniform_fake_quant: n not a multiple of inner");
        const int64_t rows = n / inner;
        const bool vec = (inner % 4 == 0) && aligned16(x) && (!y || aligned16(y)) && (!bins || ((uintptr_t)bins & 3) == 0);
        int gx = grid_for(vec ? inner / 4 : inner);
        int gy = (int)(rows < 2048 ? rows : 2048);
        while ((int64_t)gx * gy > 16384 && gx > 1) gx = (gx + 1) / 2;
        if (vec)
            hipLaunchKernelGGL(k_uniform_rows<true>, dim3(gx, gy), dim3(256), 0, st, x, y, bins, rows, inner, scale,
                               symmetric ? nullptr : zero_point, n_channels, qmin, qmax);
        else
            hipLaunchKernelGGL(k_uniform_rows<false>, dim3(gx, gy), dim3(256), 0, st, x, y, bins, rows, inner, scale,
                               symmetric ? nullptr : zero_point, n_channels, qmin, qmax);
    }
    ADALOG_LAUNCH_CHECK("adalog_uniform_fake_quant_f32");
    return 0;
}

extern "C" int adalog_log_fake_quant_f32_pre(const float* x, float* y, uint8_t* bins, int64_t n, const float* scale,
                                             const int64_t* q, const float* table1, const float* table2, int n_bits,
                                             const float* shift, int sub_shift, int train_form, int pre, void* stream);
extern "C" int adalog_log_fake_quant_f32(const float* x, float* y, uint8_t* bins, int64_t n, const float* scale,
                                         const int64_t* q, const float* table1, const float* table2, int n_bits,
                                         const float* shift, int sub_shift, int train_form, void* stream) {
    return adalog_log_fake_quant_f32_pre(x, y, bins, n, scale, q, table1, table2, n_bits, shift, sub_shift, train_form, 0, stream);
}
// pre = 1: the quantiser's input is GELU(x) (erf form), applied on the fly: a BRECQ iteration's fc2 input quantiser reads fc1's output
// (the GELU pass and its stored result disappear; adalog_log_fq_backward_pre is the matching backward)
extern "C" int adalog_log_fake_quant_f32_pre(const float* x, float* y, uint8_t* bins, int64_t n, const float* scale,
                                             const int64_t* q, const float* table1, const float* table2, int n_bits,
                                             const float* shift, int sub_shift, int train_form, int pre, void* stream) {
    if (n == 0) return 0;
    ADALOG_ARG_CHECK(x && scale && q && n >= 0, "log_fake_quant: bad arguments");
    ADALOG_ARG_CHECK(train_form || (table1 && table2), "log_fake_quant: eval form needs table1/table2");
    ADALOG_ARG_CHECK(n_bits >= 2 && n_bits <= 8, "log_fake_quant: n_bits must be in [2,8]");
    if (n == 0) return 0;
    const int levels2 = 1 << n_bits;
    hipStream_t st = (hipStream_t)stream;
    const bool vec = (n % 4 == 0) && aligned16(x) && (!y || aligned16(y)) && (!bins || ((uintptr_t)bins & 3) == 0);
    if (vec)
        hipLaunchKernelGGL(k_adalog<true>, dim3(grid_for(n / 4)), dim3(256), 0, st, x, y, bins, n, scale, q, table1,
                           table2, levels2, shift, sub_shift, train_form, pre ? 1 : 0);
    else
        hipLaunchKernelGGL(k_adalog<false>, dim3(grid_for(n)), dim3(256), 0, st, x, y, bins, n, scale, q, table1, table2,
                           levels2, shift, sub_shift, train_form, pre ? 1 : 0);
    ADALOG_LAUNCH_CHECK("adalog_log_fake_quant_f32");
    return 0;
}
