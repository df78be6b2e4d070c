// K17 -- fused forward/backward kernels for the BRECQ / AdaRound block-reconstruction stage.
//   adalog_uniform_fq_backward   <- quantizers/uniform.py:29-35 with round_ste (_ste.py:5-6): STE gradients
//   adalog_log_fq_backward       <- quantizers/logarithm.py:88-92 (training form of AdaLog) STE gradients
//   adalog_adaround_forward/_backward <- quantizers/adaround.py:43-60 (learned hard-sigmoid rounding)
//   adalog_round_loss            <- utils/block_recon.py:205-210 (regulariser sum(1 - |2h-1|^b)) value + gradient
// The reference builds these from ~10 autograd nodes each; here each is one elementwise kernel, with the per-channel
// parameter gradients reduced deterministically (per-block partials, then a fixed-order fp64 finish).
#include "common.h"

namespace {

__device__ __forceinline__ float block_sum(float v, float* sm) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0) sm[w] = v;
    __syncthreads();
    float r = (sm[0] + sm[1]) + (sm[2] + sm[3]);
    __syncthreads();
    return r;
}

// x viewed [rows][inner], channel = row % n_ch (per-tensor: rows = 1).  part_* : [rows][gridDim.x]
__global__ __launch_bounds__(256) void k_uniform_bwd(const float* __restrict__ gy, const float* __restrict__ x,
                                                     float* __restrict__ gx, int64_t rows, int64_t inner,
                                                     const float* __restrict__ scale, const float* __restrict__ zp,
                                                     int64_t n_ch, float qmin, float qmax, float* __restrict__ part_s,
                                                     float* __restrict__ part_z) {
    __shared__ float sm[4];
    for (int64_t row = blockIdx.y; row < rows; row += gridDim.y) {
        const int64_t ch = row % n_ch;
        const float s = scale[ch];
        const float z = zp ? rintf(zp[ch]) : 0.0f;
        float as = 0.0f, az = 0.0f;
        for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < inner; i += (int64_t)gridDim.x * blockDim.x) {
            const int64_t o = row * inner + i;
            const float xv = x[o], g = gy[o];
            const float t = rintf(xv / s) + z;
            const bool inside = (t >= qmin) && (t <= qmax);
            const float q = fminf(fmaxf(t, qmin), qmax);
            if (gx) gx[o] = inside ? g : 0.0f;
            as += g * ((q - z) - (inside ? xv / s : 0.0f));
            az += inside ? 0.0f : -g * s;
        }
        if (part_s) {
            const float ts = block_sum(as, sm), tz = block_sum(az, sm);
            if (threadIdx.x == 0) {
                part_s[row * gridDim.x + blockIdx.x] = ts;
                if (part_z) part_z[row * gridDim.x + blockIdx.x] = tz;
            }
        }
    }
}

// out[ch] = sum over rows with row % n_ch == ch and over the nb block partials (fp64, fixed order)
__global__ __launch_bounds__(64) void k_param_grad_finish(const float* __restrict__ part, int64_t rows, int nb, int64_t n_ch,
                                                          float* __restrict__ out) {
    const int64_t ch = blockIdx.x;
    const int lane = threadIdx.x;
    double acc = 0.0;
    const int64_t per = ((rows - ch) + n_ch - 1) / n_ch;       // rows ch, ch+n_ch, ...
    for (int64_t i = lane; i < per * nb; i += 64) {
        const int64_t row = ch + (i / nb) * n_ch;
        acc += (double)part[row * nb + (i % nb)];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    if (lane == 0) out[ch] = (float)acc;
}

__global__ __launch_bounds__(256) void k_adalog_bwd(const float* __restrict__ gy, const float* __restrict__ x,
                                                    const float* __restrict__ y, float* __restrict__ gx, int64_t n,
                                                    const float* __restrict__ scale, const int64_t* __restrict__ q,
                                                    int levels2, const float* __restrict__ shift, int sub_shift,
                                                    float* __restrict__ part_s) {
    __shared__ float sm[4];
    const float s = scale[0], qf = (float)q[0];
    const float sh = shift ? shift[0] : 0.0f;
    float as = 0.0f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float xs = shift ? x[i] + sh : x[i];
        const float ur = xs / s;
        const bool iu = (ur >= 1e-15f) && (ur <= 1.0f);
        const float u = fminf(fmaxf(ur, 1e-15f), 1.0f);
        const float k = adalog_k(u, qf);
        const bool ik = (k >= 0.0f) && (k <= (float)(levels2 - 1));
        // y before the "- shift", recomputed from k (adding the shift back to the stored output would cancel small y)
        const float yv = (k < (float)levels2) ? exp2f(-1.0f * fminf(fmaxf(k, 0.0f), (float)(levels2 - 1)) * qf / 37.0f) * s : 0.0f;
        const float g = gy[i];
        const float dydx = (iu && ik) ? yv / (u * s) : 0.0f;
        if (gx) gx[i] = g * dydx;
        as += g * (yv / s - dydx * xs / s);
    }
    if (part_s) {
        const float ts = block_sum(as, sm);
        if (threadIdx.x == 0) part_s[blockIdx.x] = ts;
    }
}

// AdaRound: w viewed [rows][inner] with per-row scale/zp; alpha same shape as w
__device__ __forceinline__ float soft_h(float a, float& dh) {
    const float sg = 1.0f / (1.0f + __expf(-a));
    const float v = sg * 1.2f - 0.1f;                  // sigmoid * (zeta - gamma) + gamma, adaround.py:35-36,60
    const bool lin = (v >= 0.0f) && (v <= 1.0f);
    dh = lin ? 1.2f * sg * (1.0f - sg) : 0.0f;
    return fminf(fmaxf(v, 0.0f), 1.0f);
}

__global__ __launch_bounds__(256) void k_adaround(const float* __restrict__ w, const float* __restrict__ alpha,
                                                  const float* __restrict__ gy, float* __restrict__ out, int64_t rows,
                                                  int64_t inner, const float* __restrict__ scale,
                                                  const float* __restrict__ zp, float qmax, int soft, int backward) {
    const int64_t n = rows * inner;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = i / inner;
        const float s = scale[row], z = zp[row];
        const float fl = floorf(w[i] / s);
        float dh = 0.0f;
        const float a = alpha[i];
        const float h = soft ? soft_h(a, dh) : (a >= 0.0f ? 1.0f : 0.0f);
        const float t = fl + h + z;
        if (!backward) {
            out[i] = (fminf(fmaxf(t, 0.0f), qmax) - z) * s;
        } else {
            const bool inside = (t >= 0.0f) && (t <= qmax);
            out[i] = (soft && inside) ? gy[i] * s * dh : 0.0f;          // d/d alpha
        }
    }
}

// round loss value (block partials) and, when galpha != null, galpha[i] += gscale * d/d alpha
__global__ __launch_bounds__(256) void k_round_loss(const float* __restrict__ alpha, int64_t n, float b,
                                                    const float* __restrict__ b_dev, float* __restrict__ part,
                                                    float* __restrict__ galpha, float gscale) {
    __shared__ float sm[4];
    if (b_dev) b = b_dev[0];                 // exponent read on the device: lets a captured HIP graph follow the decaying b
    float acc = 0.0f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        float dh;
        const float h = soft_h(alpha[i], dh);
        const float d = 2.0f * (h - 0.5f);
        const float ad = fabsf(d);
        acc += 1.0f - powf(ad, b);
        if (galpha) {
            const float sgn = d > 0.0f ? 1.0f : (d < 0.0f ? -1.0f : 0.0f);
            const float dl = ad > 0.0f ? -b * powf(ad, b - 1.0f) * sgn * 2.0f * dh : 0.0f;
            galpha[i] += gscale * dl;
        }
    }
    if (part) {
        const float t = block_sum(acc, sm);
        if (threadIdx.x == 0) part[blockIdx.x] = t;
    }
}

inline int grid1(int64_t n, int cap = 2048) {
    int64_t b = (n + 255) / 256;
    if (b < 1) b = 1;
    if (b > cap) b = cap;
    return (int)b;
}

}  // namespace

extern "C" int adalog_uniform_fq_backward_blocks(int64_t n, int64_t n_channels, int64_t inner) {
    (void)n_channels;
    return grid1(inner, 64);
}

// gscale / gzp: [n_channels] (each optional).  workspace: 2 * rows * adalog_uniform_fq_backward_blocks floats.
extern "C" int adalog_uniform_fq_backward(const float* gy, const float* x, float* gx, int64_t n, const float* scale,
                                          const float* zero_point, int64_t n_channels, int64_t inner, int n_bits,
                                          int symmetric, float* gscale, float* gzp, float* workspace, void* stream) {
    if (n == 0) return 0;
    ADALOG_ARG_CHECK(gy && x && scale && n_channels >= 1 && inner >= 1 && n % inner == 0, "uniform_fq_backward: bad arguments");
    ADALOG_ARG_CHECK(symmetric || zero_point, "uniform_fq_backward: asymmetric needs zero_point");
    ADALOG_ARG_CHECK(!(gscale || gzp) || workspace, "uniform_fq_backward: parameter gradients need a workspace");
    const float L = (float)(1 << (n_bits - 1));
    const float qmin = symmetric ? -L : 0.0f, qmax = symmetric ? L - 1.0f : 2.0f * L - 1.0f;
    const int64_t rows = n / inner;
    const int nb = grid1(inner, 64);
    int gy_ = (int)(rows < 4096 ? rows : 4096);
    hipStream_t st = (hipStream_t)stream;
    float* ps = (gscale || gzp) ? workspace : nullptr;
    float* pz = (gzp && !symmetric) ? workspace + rows * nb : nullptr;
    hipLaunchKernelGGL(k_uniform_bwd, dim3(nb, gy_), dim3(256), 0, st, gy, x, gx, rows, inner, scale,
                       symmetric ? nullptr : zero_point, n_channels, qmin, qmax, ps, pz);
    ADALOG_LAUNCH_CHECK("adalog_uniform_fq_backward");
    if (gscale) hipLaunchKernelGGL(k_param_grad_finish, dim3((unsigned)n_channels), dim3(64), 0, st, ps, rows, nb, n_channels, gscale);
    if (pz) hipLaunchKernelGGL(k_param_grad_finish, dim3((unsigned)n_channels), dim3(64), 0, st, pz, rows, nb, n_channels, gzp);
    ADALOG_LAUNCH_CHECK("adalog_uniform_fq_backward/finish");
    return 0;
}

// gscale: [1] optional; workspace: 1024 floats
extern "C" int adalog_log_fq_backward(const float* gy, const float* x, const float* y, float* gx, int64_t n,
                                      const float* scale, const int64_t* q, int n_bits, const float* shift, int sub_shift,
                                      float* gscale, float* workspace, void* stream) {
    if (n == 0) return 0;
    ADALOG_ARG_CHECK(gy && x && y && scale && q, "log_fq_backward: bad arguments");
    ADALOG_ARG_CHECK(!gscale || workspace, "log_fq_backward: the scale gradient needs a workspace");
    const int nb = grid1(n, 1024);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(k_adalog_bwd, dim3(nb), dim3(256), 0, st, gy, x, y, gx, n, scale, q, 1 << n_bits, shift, sub_shift,
                       gscale ? workspace : nullptr);
    ADALOG_LAUNCH_CHECK("adalog_log_fq_backward");
    if (gscale) hipLaunchKernelGGL(k_param_grad_finish, dim3(1), dim3(64), 0, st, workspace, (int64_t)1, nb, (int64_t)1, gscale);
    ADALOG_LAUNCH_CHECK("adalog_log_fq_backward/finish");
    return 0;
}

extern "C" int adalog_adaround(const float* w, const float* alpha, const float* gy, float* out, int64_t rows, int64_t inner,
                               const float* scale, const float* zero_point, int n_bits, int soft, int backward,
                               void* stream) {
    if (rows * inner == 0) return 0;
    ADALOG_ARG_CHECK(w && alpha && out && scale && zero_point && (!backward || gy), "adaround: bad arguments");
    hipLaunchKernelGGL(k_adaround, dim3(grid1(rows * inner, 8192)), dim3(256), 0, (hipStream_t)stream, w, alpha, gy, out, rows,
                       inner, scale, zero_point, (float)((1 << n_bits) - 1), soft, backward);
    ADALOG_LAUNCH_CHECK("adalog_adaround");
    return 0;
}

// loss[0] = sum_i (1 - |2 h(alpha_i) - 1|^b); if galpha: galpha += gscale * dloss/dalpha.  workspace: 1024 floats
extern "C" int adalog_round_loss(const float* alpha, int64_t n, float b, const float* b_dev, float* loss, float* galpha,
                                 float gscale, float* workspace, void* stream) {
    ADALOG_ARG_CHECK(alpha && n >= 1 && (loss == nullptr || workspace), "round_loss: bad arguments");
    const int nb = grid1(n, 1024);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(k_round_loss, dim3(nb), dim3(256), 0, st, alpha, n, b, b_dev, loss ? workspace : nullptr, galpha, gscale);
    ADALOG_LAUNCH_CHECK("adalog_round_loss");
    if (loss) hipLaunchKernelGGL(k_param_grad_finish, dim3(1), dim3(64), 0, st, workspace, (int64_t)1, nb, (int64_t)1, loss);
    ADALOG_LAUNCH_CHECK("adalog_round_loss/finish");
    return 0;
}
