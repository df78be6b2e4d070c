// K17 -- fused forward/backward kernels for the BRECQ / AdaRound block-reconstruction stage.
//   adalog_uniform_fq_backward   <- quantizers/uniform.py:29-35 with round_ste (_ste.py:5-6): STE gradients
//   adalog_log_fq_backward       <- quantizers/logarithm.py:88-92 (training form of AdaLog) STE gradients
//   adalog_adaround_forward/_backward <- quantizers/adaround.py:43-60 (learned hard-sigmoid rounding)
//   adalog_round_loss            <- utils/block_recon.py:205-210 (regulariser sum(1 - |2h-1|^b)) value + gradient
// The reference builds these from ~10 autograd nodes each; here each is one elementwise kernel, with the per-channel
// parameter gradients reduced deterministically (per-block partials, then a fixed-order fp64 finish).
#include "common.h"

#include <atomic>
#include <mutex>

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

// "Last block finishes" reductions.  Every block leaves an fp32 partial and takes a ticket; the block that draws the
// last ticket sums the partials in fp64 in a fixed order.  The partials and the ticket are device-scope atomics (they go
// straight to the coherence point, across the eight XCDs' L2s), ordered by a wait on the store: no L2 write-back, which
// a device-scope fence in every block would cost (measured: 3x the kernel time).
__device__ __forceinline__ void store_agent(float* p, float v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ float load_agent(const float* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// Call after thread 0 issued its store_agent()s.  True in exactly one block; the counters are back at zero afterwards.
// Device-scope atomics on ONE word serialise (~10 ns each, measured: a 4 096-block launch spent 41 us of 62 drawing tickets), so
// the ticket has two levels: a block draws from one of TW_GROUPS first-level words (by block index, so that blocks finishing
// together hit different words; the words lie TW_STRIDE apart -- another 4 KiB page and another 256-byte slot, whichever the
// memory channels interleave on), and the last block of a group draws from the top word.  Every draw waits for the value of the
// one before it, and a block's partial stores are acknowledged before its first draw: when the last top ticket is drawn, all
// partials are visible.
constexpr int TW_GROUPS = 16;
constexpr int TW_STRIDE = 1088;                         // words
constexpr int TW_WORDS = (TW_GROUPS + 1) * TW_STRIDE;   // one wide ticket
__device__ __forceinline__ bool last_block_arrives(unsigned int* tw, unsigned total) {
    __shared__ int is_last;
    if (threadIdx.x == 0) {
        // the partial stores must be acknowledged before the ticket is drawn.  A workgroup-scope fence compiles to
        // `s_waitcnt lgkmcnt(0)` only -- the stores are VMEM operations to other addresses (possibly other channels) than the
        // ticket, so without the explicit vmcnt wait the ticket could become visible first (tests/test_abi.py greps the ISA)
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned lin = blockIdx.y * gridDim.x + blockIdx.x;
        const unsigned ng = total < (unsigned)TW_GROUPS ? total : (unsigned)TW_GROUPS;
        const unsigned gi = lin % ng;
        const unsigned cnt = total / ng + (gi < total % ng ? 1u : 0u);
        unsigned int* c1 = tw + gi * TW_STRIDE;
        unsigned int* top = tw + TW_GROUPS * TW_STRIDE;
        int last = 0;
        const unsigned t = __hip_atomic_fetch_add(c1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (t == cnt - 1) {
            const unsigned t2 = __hip_atomic_fetch_add(top, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(c1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (t2 == ng - 1) {
                __hip_atomic_store(top, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                last = 1;
            }
        }
        is_last = last;
    }
    __syncthreads();
    return is_last != 0;
}

// fixed-order fp64 sum of part[0..count) by the 256 threads of one block (the same order whichever block runs it)
__device__ __forceinline__ float block_total_f64(const float* part, int64_t count, double* smd, double mul = 1.0) {
    double acc = 0.0;
    for (int64_t i = threadIdx.x; i < count; i += 256) acc += (double)load_agent(part + i);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    __syncthreads();
    if (lane == 0) smd[w] = acc;
    __syncthreads();
    return (float)(((smd[0] + smd[1]) + (smd[2] + smd[3])) * mul);
}

__device__ __forceinline__ void uniform_bwd_one(float xv, float g, float s, float z, float qmin, float qmax, float& gxo,
                                                float& as, float& az) {
    const float r = xv / s;
    const float t = rintf(r) + z;
    const bool inside = (t >= qmin) && (t <= qmax);
    const float q = fminf(fmaxf(t, qmin), qmax);
    gxo = inside ? g : 0.0f;
    as += g * ((q - z) - (inside ? r : 0.0f));
    az += inside ? 0.0f : -g * s;
}

// x viewed [rows][inner], channel = row % n_ch (per-tensor: rows = 1).  part_* : [rows][gridDim.x].
// counter != null (per-tensor form only): the last block to finish reduces the partials into gscale / gzp itself.
__global__ __launch_bounds__(256) void k_uniform_bwd(const float* __restrict__ gy, const float* __restrict__ x,
                                                     float* __restrict__ gx, int64_t rows, int64_t inner,
                                                     const float* __restrict__ scale, const float* __restrict__ zp,
                                                     int64_t n_ch, float qmin, float qmax, float* __restrict__ part_s,
                                                     float* __restrict__ part_z, unsigned int* counter,
                                                     float* __restrict__ gscale, float* __restrict__ gzp) {
    __shared__ float sm[4];
    __shared__ double smd[4];
    const bool vec = (inner & 3) == 0;
    for (int64_t row = blockIdx.y; row < rows; row += gridDim.y) {
        const int64_t ch = row % n_ch;
        const float s = scale[ch];
        const float z = zp ? rintf(zp[ch]) : 0.0f;
        float as = 0.0f, az = 0.0f;
        if (vec) {
            const float4* x4 = reinterpret_cast<const float4*>(x + row * inner);
            const float4* g4 = reinterpret_cast<const float4*>(gy + row * inner);
            float4* o4 = gx ? reinterpret_cast<float4*>(gx + row * inner) : nullptr;
            for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < (inner >> 2); i += (int64_t)gridDim.x * blockDim.x) {
                const float4 xv = x4[i], g = g4[i];
                float4 o;
                uniform_bwd_one(xv.x, g.x, s, z, qmin, qmax, o.x, as, az);
                uniform_bwd_one(xv.y, g.y, s, z, qmin, qmax, o.y, as, az);
                uniform_bwd_one(xv.z, g.z, s, z, qmin, qmax, o.z, as, az);
                uniform_bwd_one(xv.w, g.w, s, z, qmin, qmax, o.w, as, az);
                if (o4) o4[i] = o;
            }
        } else {
            for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < inner; i += (int64_t)gridDim.x * blockDim.x) {
                const int64_t o = row * inner + i;
                float go;
                uniform_bwd_one(x[o], gy[o], s, z, qmin, qmax, go, as, az);
                if (gx) gx[o] = go;
            }
        }
        if (part_s) {
            const float ts = block_sum(as, sm), tz = block_sum(az, sm);
            if (threadIdx.x == 0) {
                store_agent(part_s + row * gridDim.x + blockIdx.x, ts);
                if (part_z) store_agent(part_z + row * gridDim.x + blockIdx.x, tz);
            }
        }
    }
    if (counter && part_s && last_block_arrives(counter, gridDim.x * gridDim.y)) {
        const int64_t cnt = rows * gridDim.x;
        if (gscale) { const float v = block_total_f64(part_s, cnt, smd); if (threadIdx.x == 0) gscale[0] = v; }
        if (gzp && part_z) { const float v = block_total_f64(part_z, cnt, smd); if (threadIdx.x == 0) gzp[0] = v; }
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
                                                    float* __restrict__ part_s, unsigned int* counter,
                                                    float* __restrict__ gscale, int pre) {
    __shared__ float sm[4];
    __shared__ double smd[4];
    const float s = scale[0], qf = (float)q[0];
    const float sh = shift ? shift[0] : 0.0f;
    const float inv_s = 1.0f / s, rq37 = 37.0f / qf, q37 = qf / 37.0f, kmax = (float)(levels2 - 1);
    float as = 0.0f;
    // k is the forward's bin (same exact-with-fallback evaluation); the gradient factors use reciprocals: they are smooth
    // in their inputs, so a last-ulp difference from the IEEE quotients moves the gradient by ~1e-7 relative.
    auto one = [&](float xv, float g, float& gxo) {
        // pre: the quantiser saw GELU(xv); d GELU / dx = cdf + x * pdf as ATen's GeluBackward evaluates it
        float dg = 1.0f;
        if (pre) {
            const float cdf = 0.5f * (1.0f + erff(xv * 0.70710678118654752440f));
            const float pdf = expf(-0.5f * xv * xv) * 0.39894228040143267794f;
            dg = cdf + xv * pdf;
            xv = (xv * 0.5f) * (1.0f + erff(xv * 0.70710678118654752440f));
        }
        const float xs = shift ? xv + sh : xv;
        const float ur = xs / s;
        const bool iu = (ur >= 1e-15f) && (ur <= 1.0f);
        const float u = fminf(fmaxf(ur, 1e-15f), 1.0f);
        const float k = adalog_k_fast(u, 1.0f, 1.0f, qf, rq37, false);
        const bool ik = (k >= 0.0f) && (k <= kmax);
        // y before the "- shift", recomputed from k (adding the shift back to the stored output would cancel small y)
        const float yv = (k <= kmax) ? __builtin_amdgcn_exp2f(-fminf(fmaxf(k, 0.0f), kmax) * q37) * s : 0.0f;
        const float dydx = (iu && ik) ? yv * __builtin_amdgcn_rcpf(u * s) : 0.0f;
        gxo = g * dydx * dg;
        as += g * inv_s * (yv - dydx * xs);
    };
    // four elements per thread and turn (16-byte loads; x, gy, gx come from the allocator: 16-byte aligned)
    const int64_t n4 = ((((uintptr_t)gy | (uintptr_t)x | (uintptr_t)gx) & 15) == 0) ? (n >> 2) : 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        const float4 xv = reinterpret_cast<const float4*>(x)[i], g = reinterpret_cast<const float4*>(gy)[i];
        float4 o;
        one(xv.x, g.x, o.x); one(xv.y, g.y, o.y); one(xv.z, g.z, o.z); one(xv.w, g.w, o.w);
        if (gx) reinterpret_cast<float4*>(gx)[i] = o;
    }
    for (int64_t i = (n4 << 2) + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        float o;
        one(x[i], gy[i], o);
        if (gx) gx[i] = o;
    }
    if (part_s) {
        const float ts = block_sum(as, sm);
        if (threadIdx.x == 0) store_agent(part_s + blockIdx.x, ts);
        if (last_block_arrives(counter, gridDim.x)) {
            const float v = block_total_f64(part_s, gridDim.x, smd);
            if (threadIdx.x == 0) gscale[0] = v;
        }
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

// The AdaRound forward writing its result in BOTH orientations: out [rows][inner] and out_t [inner][rows] (the K-major image the
// forward product of a BRECQ iteration reads: brecq_gemm.hip fetches a K-contiguous operand as half cache lines).  A block moves a
// 32 x 32 tile through LDS so that both stores are coalesced.
__global__ __launch_bounds__(256) void k_adaround_t(const float* __restrict__ w, const float* __restrict__ alpha,
                                                    float* __restrict__ out, float* __restrict__ out_t, int64_t rows, int64_t inner,
                                                    const float* __restrict__ scale, const float* __restrict__ zp, float qmax,
                                                    int soft) {
    __shared__ float tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int64_t c0 = (int64_t)blockIdx.x * 32, r0 = (int64_t)blockIdx.y * 32;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int64_t r = r0 + ty + 8 * i, c = c0 + tx;
        if (r < rows && c < inner) {
            const int64_t idx = r * inner + c;
            const float s = scale[r], z = zp[r];
            const float fl = floorf(w[idx] / s);
            float dh = 0.0f;
            const float a = alpha[idx];
            const float h = soft ? soft_h(a, dh) : (a >= 0.0f ? 1.0f : 0.0f);
            const float v = (fminf(fmaxf(fl + h + z, 0.0f), qmax) - z) * s;
            out[idx] = v;
            tile[ty + 8 * i][tx] = v;
        }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int64_t c = c0 + ty + 8 * i, r = r0 + tx;
        if (r < rows && c < inner) out_t[c * rows + r] = tile[tx][ty + 8 * i];
    }
}

// x^e for x in (0, 1] through the hardware log2 / exp2 (~1e-5 relative at the largest exponent the schedule uses, b = 20; the
// library powf is ~100 instructions and made the regulariser's kernels compute-bound: 47 us for the 7 M alphas of a vit_base
// block whose bytes take 8).  Every kernel of the regulariser uses this one form, so the routes agree bit for bit.
__device__ __forceinline__ float pow01(float x, float e) { return __builtin_amdgcn_exp2f(e * __builtin_amdgcn_logf(x)); }

// round loss value (block partials) and, when galpha != null, galpha[i] += gscale * d/d alpha
__global__ __launch_bounds__(256) void k_round_loss(const float* __restrict__ alpha, int64_t n, float b,
                                                    const float* __restrict__ b_dev, float* __restrict__ part,
                                                    float* __restrict__ galpha, float gscale, const float* __restrict__ gmul,
                                                    int overwrite, unsigned int* counter, float* __restrict__ loss) {
    __shared__ float sm[4];
    __shared__ double smd[4];
    if (gmul) gscale *= gmul[0];             // upstream gradient of the (scalar) loss, read on the device
    if (b_dev) b = b_dev[0];                 // exponent read on the device: lets a captured HIP graph follow the decaying b
    float acc = 0.0f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        float dh;
        const float h = soft_h(alpha[i], dh);
        const float d = 2.0f * (h - 0.5f);
        const float ad = fabsf(d);
        acc += 1.0f - (ad > 0.0f ? pow01(ad, b) : 0.0f);
        if (galpha) {
            const float sgn = d > 0.0f ? 1.0f : (d < 0.0f ? -1.0f : 0.0f);
            const float dl = ad > 0.0f ? -b * pow01(ad, b - 1.0f) * sgn * 2.0f * dh : 0.0f;
            galpha[i] = overwrite ? gscale * dl : galpha[i] + gscale * dl;
        }
    }
    if (part) {
        const float t = block_sum(acc, sm);
        if (threadIdx.x == 0) store_agent(part + blockIdx.x, t);
        if (last_block_arrives(counter, gridDim.x)) {
            const float v = block_total_f64(part, gridDim.x, smd);
            if (threadIdx.x == 0) loss[0] = v;
        }
    }
}

// The rounding regulariser of a whole block in one launch: up to RL_MAX alpha tensors, value and gradient together.
constexpr int RL_MAX = 16;
struct RoundLossMulti {
    const float* alpha[RL_MAX];
    float* grad[RL_MAX];
    int64_t n[RL_MAX];
    int first_block[RL_MAX + 1];          // tensor t owns blocks [first_block[t], first_block[t+1])
    int count;
};

__global__ __launch_bounds__(256) void k_round_loss_multi(RoundLossMulti a, float b, const float* __restrict__ b_dev,
                                                          float weight, float* __restrict__ part, unsigned int* counter,
                                                          float* __restrict__ loss, const float* __restrict__ gate) {
    __shared__ float sm[4];
    __shared__ double smd[4];
    if (b_dev) b = b_dev[0];
    int t = 0;
    while (t + 1 < a.count && (int)blockIdx.x >= a.first_block[t + 1]) ++t;
    const float* __restrict__ al = a.alpha[t];
    float* __restrict__ gr = a.grad[t];
    const int64_t n = a.n[t];
    const int nblk = a.first_block[t + 1] - a.first_block[t];
    float acc = 0.0f;
    for (int64_t i = (int64_t)(blockIdx.x - a.first_block[t]) * 256 + threadIdx.x; i < n; i += (int64_t)nblk * 256) {
        float dh;
        const float h = soft_h(al[i], dh);
        const float d = 2.0f * (h - 0.5f);
        const float ad = fabsf(d);
        const float pm1 = ad > 0.0f ? pow01(ad, b - 1.0f) : 0.0f;
        acc += 1.0f - pm1 * ad;
        const float sgn = d > 0.0f ? 1.0f : (d < 0.0f ? -1.0f : 0.0f);
        if (gr) gr[i] = weight * (-b * pm1 * sgn * 2.0f * dh);
    }
    const float ts = block_sum(acc, sm);
    if (threadIdx.x == 0) store_agent(part + blockIdx.x, ts);
    if (last_block_arrives(counter, gridDim.x)) {
        const float v = block_total_f64(part, gridDim.x, smd, (double)weight);
        if (threadIdx.x == 0) loss[0] = gate ? v * gate[0] : v;      // (gate: the 0 / 1 switch of the warm-up, read on the device)
    }
}

// One Adam step (torch.optim.Adam defaults: no weight decay, no amsgrad) for up to ADAM_MAX tensors in ONE launch.  The
// reference's two optimisers (utils/block_recon.py:108-109) step ~10 tensors per iteration; torch's fused multi-tensor kernel
// gives a 65 536-element chunk to a block, i.e. ~40 blocks for a block's 2.4 M trained values (52 us); here every block takes
// 1024 elements.  step_dev holds the number of steps taken so far (the kernel uses step_dev[0] + 1 and its last workgroup to
// arrive stores that back, so a captured HIP graph keeps counting); lr_dev (optional) overrides lr: the cosine schedule writes it on the host.
constexpr int ADAM_MAX = 16;
struct AdamMulti {
    float* param[ADAM_MAX];
    const float* grad[ADAM_MAX];
    float* m[ADAM_MAX];
    float* v[ADAM_MAX];
    int64_t n[ADAM_MAX];
    int first_block[ADAM_MAX + 1];
    int count;
};

__global__ __launch_bounds__(256) void k_adam_multi(AdamMulti a, float lr, const float* __restrict__ lr_dev, float beta1, float beta2,
                                                    float eps, float* step_dev, unsigned int* ticket) {
    int t = 0;
    while (t + 1 < a.count && (int)blockIdx.x >= a.first_block[t + 1]) ++t;
    if (lr_dev) lr = lr_dev[0];
    const float step = step_dev[0] + 1.0f;
    const float bc1 = 1.0f - powf(beta1, step), bc2 = 1.0f - powf(beta2, step);
    const float step_size = lr / bc1, bc2s = sqrtf(bc2);
    float* __restrict__ pp = a.param[t];
    const float* __restrict__ gg = a.grad[t];
    float* __restrict__ mm = a.m[t];
    float* __restrict__ vv = a.v[t];
    const int64_t n = a.n[t];
    const int64_t base = (int64_t)(blockIdx.x - a.first_block[t]) * 1024;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int64_t i = base + u * 256 + threadIdx.x;
        if (i < n) {
            const float g = gg[i];
            const float m_ = mm[i] + (1.0f - beta1) * (g - mm[i]);            // exp_avg.lerp_(grad, 1 - beta1)
            const float v_ = beta2 * vv[i] + (1.0f - beta2) * g * g;          // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, 1 - beta2)
            const float denom = sqrtf(v_) / bc2s + eps;
            mm[i] = m_; vv[i] = v_;
            pp[i] = pp[i] - step_size * (m_ / denom);                         // param.addcdiv_(exp_avg, denom, value=-step_size)
        }
    }
    // the step counter advances in this launch: every workgroup has read it by the time the last one arrives (round 6; a one-thread
    // k_adam_count launch per optimiser and iteration before)
    if (last_block_arrives(ticket, gridDim.x) && threadIdx.x == 0) step_dev[0] = step;
}

// The whole update of AdaRound's alpha in one launch (a captured BRECQ iteration, one GPU): the gradient through the soft-rounded
// weights (k_adaround's backward form, from dL/dw_sim), the gradient of the rounding regulariser (k_round_loss_multi's, times its
// upstream factor) and the Adam step (k_adam_multi's), element by element with the same operations in the same order -- what
// autograd does with four backward launches, a multiply, four accumulations and the optimiser's launch.  gw[t] may be null (a
// layer whose weights took no gradient this iteration).
struct AlphaStepMulti {
    float* alpha[ADAM_MAX];
    const float* w[ADAM_MAX];
    const float* gw[ADAM_MAX];
    const float* scale[ADAM_MAX];
    const float* zp[ADAM_MAX];
    float* m[ADAM_MAX];
    float* v[ADAM_MAX];
    int64_t n[ADAM_MAX];
    int64_t inner[ADAM_MAX];
    float qmax[ADAM_MAX];
    int first_block[ADAM_MAX + 1];
    int count;
};

__global__ __launch_bounds__(256) void k_alpha_step_multi(AlphaStepMulti a, float lr, const float* __restrict__ lr_dev, float beta1,
                                                          float beta2, float eps, float* step_dev, float b,
                                                          const float* __restrict__ b_dev, float weight,
                                                          const float* __restrict__ gmul, const float* __restrict__ gate, int epb,
                                                          unsigned int* ticket) {
    int t = 0;
    while (t + 1 < a.count && (int)blockIdx.x >= a.first_block[t + 1]) ++t;
    if (lr_dev) lr = lr_dev[0];
    if (b_dev) b = b_dev[0];
    const float g_rl = gmul ? (gate ? gmul[0] * gate[0] : gmul[0]) : 0.0f;      // upstream gradient of the regulariser (times its 0 / 1 gate)
    const float step = step_dev[0] + 1.0f;
    const float bc1 = 1.0f - powf(beta1, step), bc2 = 1.0f - powf(beta2, step);
    const float step_size = lr / bc1, bc2s = sqrtf(bc2);
    float* __restrict__ al = a.alpha[t];
    const float* __restrict__ ww = a.w[t];
    const float* __restrict__ gw = a.gw[t];
    const float* __restrict__ sc = a.scale[t];
    const float* __restrict__ zz = a.zp[t];
    float* __restrict__ mm = a.m[t];
    float* __restrict__ vv = a.v[t];
    const int64_t n = a.n[t], inner = a.inner[t];
    const float qmax = a.qmax[t];
    // (epb = 1 024 or 4 096 values per block: the two powf of the bias corrections above are per-wave work worth ~4 values, so
    // large blocks pay -- once there are enough of them to fill the chip)
    const int64_t base = (int64_t)(blockIdx.x - a.first_block[t]) * epb;
    for (int u = 0; u < epb / 256; ++u) {
        const int64_t i = base + u * 256 + threadIdx.x;
        if (i < n) {
            const int64_t row = i / inner;
            const float s = sc[row], z = zz[row];
            const float av = al[i];
            float dh;
            const float h = soft_h(av, dh);
            // d/d alpha through w_sim (k_adaround, backward form, soft targets)
            const float tt = floorf(ww[i] / s) + h + z;
            const bool inside = (tt >= 0.0f) && (tt <= qmax);
            const float ga = (gw && inside) ? gw[i] * s * dh : 0.0f;
            // d/d alpha of the regulariser (k_round_loss_multi) times its upstream factor
            const float d = 2.0f * (h - 0.5f);
            const float ad = fabsf(d);
            const float pm1 = ad > 0.0f ? pow01(ad, b - 1.0f) : 0.0f;
            const float sgn = d > 0.0f ? 1.0f : (d < 0.0f ? -1.0f : 0.0f);
            const float grl = weight * (-b * pm1 * sgn * 2.0f * dh);
            const float g = ga + grl * g_rl;
            // Adam (k_adam_multi)
            const float m_ = mm[i] + (1.0f - beta1) * (g - mm[i]);
            const float v_ = beta2 * vv[i] + (1.0f - beta2) * g * g;
            const float denom = sqrtf(v_) / bc2s + eps;
            mm[i] = m_; vv[i] = v_;
            al[i] = av - step_size * (m_ / denom);
        }
    }
    if (last_block_arrives(ticket, gridDim.x) && threadIdx.x == 0) step_dev[0] = step;     // (as in k_adam_multi)
}

// reconstruction loss  scale * sum_i (pred_i - tgt_i)^2  (block_recon.py:186-199 with p = 2) and its gradient
__global__ __launch_bounds__(256) void k_rec_loss(const float* __restrict__ pred, const float* __restrict__ tgt, int64_t n,
                                                  float scale, float* __restrict__ part, unsigned int* counter,
                                                  float* __restrict__ loss) {
    __shared__ float sm[4];
    __shared__ double smd[4];
    float acc = 0.0f;
    const int64_t n4 = n >> 2;
    const float4* p4 = reinterpret_cast<const float4*>(pred);
    const float4* t4 = reinterpret_cast<const float4*>(tgt);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        const float4 a = p4[i], b = t4[i];
        const float d0 = a.x - b.x, d1 = a.y - b.y, d2 = a.z - b.z, d3 = a.w - b.w;
        acc += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        const float d = pred[(n4 << 2) + threadIdx.x] - tgt[(n4 << 2) + threadIdx.x];
        acc += d * d;
    }
    const float t = block_sum(acc, sm);
    if (threadIdx.x == 0) store_agent(part + blockIdx.x, t);
    if (last_block_arrives(counter, gridDim.x)) {
        const float v = block_total_f64(part, gridDim.x, smd, (double)scale);
        if (threadIdx.x == 0) loss[0] = v;
    }
}

__global__ __launch_bounds__(256) void k_rec_loss_bwd(const float* __restrict__ pred, const float* __restrict__ tgt, int64_t n,
                                                      float scale2, const float* __restrict__ gmul, float* __restrict__ gpred) {
    const float f = scale2 * gmul[0];
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        gpred[i] = (pred[i] - tgt[i]) * f;
}

// What precedes a captured BRECQ iteration's replay (utils/block_recon.py:114-117: cur_inp, cur_out = block.raw_input[idx], raw_out[idx];
// the iteration's b / gate / learning rate) as ONE launch: the mini-batch rows of the block's stored inputs and outputs gathered into
// the graph's static tensors and the iteration's schedule row copied into its device scalars -- two index_select launches and a
// device-to-device copy before.  rows of rin4 / rout4 float4 each; idx: int64 [bs].
__global__ __launch_bounds__(256) void k_brecq_prepare(const float4* __restrict__ src_in, const float4* __restrict__ src_out,
                                                       const int64_t* __restrict__ idx, float4* __restrict__ dst_in,
                                                       float4* __restrict__ dst_out, int64_t bs, int64_t rin4, int64_t rout4,
                                                       const float* __restrict__ sched_row, float* __restrict__ sched_dev, int n_sched) {
    if (blockIdx.x == 0 && (int)threadIdx.x < n_sched) sched_dev[threadIdx.x] = sched_row[threadIdx.x];
    const int64_t n_in = bs * rin4, total = n_in + bs * rout4;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        if (i < n_in) {
            const int64_t b = i / rin4, e = i - b * rin4;
            dst_in[i] = src_in[idx[b] * rin4 + e];
        } else {
            const int64_t j = i - n_in, b = j / rout4, e = j - b * rout4;
            dst_out[j] = src_out[idx[b] * rout4 + e];
        }
    }
}

// q - z of the per-tensor asymmetric quantiser, as fp32: the exact integer operand of the training-mode GEMM
// (brecq_gemm.hip), whose epilogue multiplies by the trained scale -- (q - z) * s is the fake-quantised activation.
__global__ __launch_bounds__(256) void k_uniform_int(const float* __restrict__ x, float* __restrict__ y, int64_t n,
                                                     const float* __restrict__ scale, const float* __restrict__ zp, float qmax) {
    const float s = scale[0], z = rintf(zp[0]);
    const int64_t n4 = n >> 2;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        const float4 v = reinterpret_cast<const float4*>(x)[i];
        float4 o;
        o.x = fminf(fmaxf(rintf(v.x / s) + z, 0.0f), qmax) - z;
        o.y = fminf(fmaxf(rintf(v.y / s) + z, 0.0f), qmax) - z;
        o.z = fminf(fmaxf(rintf(v.z / s) + z, 0.0f), qmax) - z;
        o.w = fminf(fmaxf(rintf(v.w / s) + z, 0.0f), qmax) - z;
        reinterpret_cast<float4*>(y)[i] = o;
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        const int64_t i = (n4 << 2) + threadIdx.x;
        y[i] = fminf(fmaxf(rintf(x[i] / s) + z, 0.0f), qmax) - z;
    }
}

// softmax(x * scale) over the last dim and its gradient, one pass each (an attention block's `attn * scale` + softmax under
// autograd is four passes: reference utils/wrap_net.py:26-27).  A wave owns a row of n <= 1024 values (lane l: elements l, l + 64,
// ...).  Forward as ATen evaluates it: t = fl(x * scale), m = max t, e = exp(t - m), y = e / sum e (fp32 sums, accurate expf).
// Backward: gx = scale * y * (gy - sum_j gy_j y_j)   (ATen: (gy - sum(gy * y)) * y, then the multiply's gradient).
constexpr int SM_MAX = 16;                                     // elements per lane at most
template <int NPL>                                          // elements per lane of this instantiation: ceil(n / 64) rounded up
__global__ __launch_bounds__(256) void k_scaled_softmax(const float* __restrict__ x, float* __restrict__ y, int64_t rows, int n,
                                                        float scale) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* xr = x + row * n;
    float t[NPL];
    float m = -__builtin_inff();
#pragma unroll
    for (int i = 0; i < NPL; ++i) {
        const int e = lane + 64 * i;
        t[i] = e < n ? xr[e] * scale : -__builtin_inff();
        m = fmaxf(m, t[i]);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    float sum = 0.0f;
#pragma unroll
    for (int i = 0; i < NPL; ++i) {
        t[i] = (lane + 64 * i) < n ? expf(t[i] - m) : 0.0f;
        sum += t[i];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
    float* yr = y + row * n;
#pragma unroll
    for (int i = 0; i < NPL; ++i) {
        const int e = lane + 64 * i;
        if (e < n) yr[e] = t[i] / sum;
    }
}

template <int NPL>
__global__ __launch_bounds__(256) void k_scaled_softmax_bwd(const float* __restrict__ gy, const float* __restrict__ y,
                                                            float* __restrict__ gx, int64_t rows, int n, float scale) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* gr = gy + row * n;
    const float* yr = y + row * n;
    float g[NPL], p[NPL];
    float dot = 0.0f;
#pragma unroll
    for (int i = 0; i < NPL; ++i) {
        const int e = lane + 64 * i;
        g[i] = e < n ? gr[e] : 0.0f;
        p[i] = e < n ? yr[e] : 0.0f;
        dot += g[i] * p[i];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) dot += __shfl_xor(dot, o);
    float* xr = gx + row * n;
#pragma unroll
    for (int i = 0; i < NPL; ++i) {
        const int e = lane + 64 * i;
        if (e < n) xr[e] = ((g[i] - dot) * p[i]) * scale;
    }
}

// The head split of an attention block as ONE pass: x [B][N][P][H][D] (the qkv Linear's output, P = 3) -> y [P][B][H][N][D]
// (contiguous q, k, v), and the inverse for the gradient.  Autograd's own route is three strided copies forward and a stack +
// copies backward.  D % 4 == 0; a thread moves one float4; consecutive threads walk D, then H (forward: reads of H * D contiguous
// floats, writes of D-float runs).
__global__ __launch_bounds__(256) void k_permute_heads(const float* __restrict__ src, float* __restrict__ dst, int64_t B, int64_t N,
                                                       int P, int H, int D4, int inverse) {
    const int64_t total = B * N * P * H * D4;
    const float4* s4 = reinterpret_cast<const float4*>(src);
    float4* d4 = reinterpret_cast<float4*>(dst);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        // i indexes the [B][N][P][H][D4] side
        int64_t r = i;
        const int d = (int)(r % D4); r /= D4;
        const int h = (int)(r % H); r /= H;
        const int pp = (int)(r % P); r /= P;
        const int64_t n = r % N, b = r / N;
        const int64_t j = ((((int64_t)pp * B + b) * H + h) * N + n) * D4 + d;       // the [P][B][H][N][D4] side
        if (inverse) d4[i] = s4[j]; else d4[j] = s4[i];
    }
}

// The inverse of the head split with the P parts given as separate tensors [B][H][N][D] (the gradients of q, k, v as autograd
// hands them over: no stack in front); a null part counts as zeros.
struct HeadParts { const float* p[4]; };
__global__ __launch_bounds__(256) void k_merge_heads(HeadParts src, float* __restrict__ dst, int64_t B, int64_t N, int P, int H,
                                                     int D4) {
    const int64_t total = B * N * P * H * D4;
    float4* d4 = reinterpret_cast<float4*>(dst);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        int64_t r = i;
        const int d = (int)(r % D4); r /= D4;
        const int h = (int)(r % H); r /= H;
        const int pp = (int)(r % P); r /= P;
        const int64_t n = r % N, b = r / N;
        const float* part = src.p[pp];
        const int64_t j = ((b * H + h) * N + n) * D4 + d;
        d4[i] = part ? reinterpret_cast<const float4*>(part)[j] : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    }
}

// q, k, v of an attention block split AND fake-quantised in one pass (a BRECQ iteration: the three tensors go straight into the
// quantisers of the two attention products -- q.k^T's A and B quantiser, softmax.v's B quantiser: asymmetric uniform, per tensor or
// per head; reference utils/wrap_net.py:21-22 + quant_layers/matmul.py:43-47), and the straight-through gradients the other way:
//   forward:  y_p[b][h][n][d] = (clamp(rne(x / s) + z, 0, qmax) - z) * s      x = src[b][n][p][h][d], (s, z) = part p, head h
//   backward: gx = inside ? gy_p : 0  into [B][N][3][H][D];  gscale_p[h] = sum gy_p * ((q - z) - inside * x / s)
// Same IEEE operations per element as k_uniform_rows / k_uniform_bwd.  grid = (row chunks, 3 * H): a block owns one (part, head)
// for a chunk of (b, n) rows, 16 rows x D / 4 float4 per pass; D in {32, 64}.  Block partials of the scale gradients go to
// part[(p * H + h) * nchunks + chunk]; k_param_grad_finish sums them (fp64, fixed order).
struct QkvQuant {
    const float* scale[3]; const float* zp[3];
    int sstride[3];                                            // 1: per head, 0: per tensor
    float qmax[3];
};
template <int D>
__global__ __launch_bounds__(256) void k_qkv_split_quant(const float* __restrict__ src, float* __restrict__ y0, float* __restrict__ y1,
                                                         float* __restrict__ y2, int64_t B, int64_t N, int H, QkvQuant q) {
    constexpr int D4 = D / 4, RPP = 256 / D4;                  // float4 per row, rows per pass
    const int ph = blockIdx.y, pp = ph / H, h = ph % H;
    const float s = q.scale[pp][h * q.sstride[pp]], z = rintf(q.zp[pp][h * q.sstride[pp]]), qmax = q.qmax[pp];
    float* __restrict__ y = pp == 0 ? y0 : (pp == 1 ? y1 : y2);
    const int d4 = threadIdx.x % D4, rl = threadIdx.x / D4;
    const int64_t rows = B * N;
    for (int64_t r = (int64_t)blockIdx.x * RPP + rl; r < rows; r += (int64_t)gridDim.x * RPP) {
        const int64_t b = r / N, n = r - b * N;
        const float4 v = reinterpret_cast<const float4*>(src + ((r * 3 + pp) * H + h) * D)[d4];
        const float q0 = fminf(fmaxf(rintf(v.x / s) + z, 0.0f), qmax), q1 = fminf(fmaxf(rintf(v.y / s) + z, 0.0f), qmax);
        const float q2 = fminf(fmaxf(rintf(v.z / s) + z, 0.0f), qmax), q3 = fminf(fmaxf(rintf(v.w / s) + z, 0.0f), qmax);
        reinterpret_cast<float4*>(y + ((b * H + h) * N + n) * D)[d4] = make_float4((q0 - z) * s, (q1 - z) * s, (q2 - z) * s, (q3 - z) * s);
    }
}

template <int D>
__global__ __launch_bounds__(256) void k_qkv_merge_quant_bwd(const float* __restrict__ g0, const float* __restrict__ g1,
                                                             const float* __restrict__ g2, const float* __restrict__ src,
                                                             float* __restrict__ gx, int64_t B, int64_t N, int H, QkvQuant q,
                                                             float* __restrict__ part) {
    __shared__ float sm[4];
    constexpr int D4 = D / 4, RPP = 256 / D4;
    const int ph = blockIdx.y, pp = ph / H, h = ph % H;
    const float s = q.scale[pp][h * q.sstride[pp]], z = rintf(q.zp[pp][h * q.sstride[pp]]), qmax = q.qmax[pp];
    const float* __restrict__ gy = pp == 0 ? g0 : (pp == 1 ? g1 : g2);
    const int d4 = threadIdx.x % D4, rl = threadIdx.x / D4;
    const int64_t rows = B * N;
    float as = 0.0f, az = 0.0f;
    for (int64_t r = (int64_t)blockIdx.x * RPP + rl; r < rows; r += (int64_t)gridDim.x * RPP) {
        const int64_t b = r / N, n = r - b * N;
        const int64_t xo = ((r * 3 + pp) * H + h) * D;
        const float4 xv = reinterpret_cast<const float4*>(src + xo)[d4];
        const float4 g = gy ? reinterpret_cast<const float4*>(gy + ((b * H + h) * N + n) * D)[d4] : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        float4 o;
        uniform_bwd_one(xv.x, g.x, s, z, 0.0f, qmax, o.x, as, az);
        uniform_bwd_one(xv.y, g.y, s, z, 0.0f, qmax, o.y, as, az);
        uniform_bwd_one(xv.z, g.z, s, z, 0.0f, qmax, o.z, as, az);
        uniform_bwd_one(xv.w, g.w, s, z, 0.0f, qmax, o.w, as, az);
        if (gx) reinterpret_cast<float4*>(gx + xo)[d4] = o;
    }
    if (part) {
        const float ts = block_sum(as, sm);
        if (threadIdx.x == 0) part[(int64_t)ph * gridDim.x + blockIdx.x] = ts;
    }
}

// scale gradients of the three parts from the block partials of k_qkv_merge_quant_bwd (fp64, fixed order): block = (part, head)
struct QkvGrads { float* g[3]; int per_head[3]; };
__global__ __launch_bounds__(64) void k_qkv_grad_finish(const float* __restrict__ part, int H, int nch, QkvGrads o) {
    const int pp = blockIdx.x / H, h = blockIdx.x % H, lane = threadIdx.x;
    if (!o.g[pp] || (!o.per_head[pp] && h != 0)) return;
    const float* src = part + ((int64_t)pp * H + (o.per_head[pp] ? h : 0)) * nch;
    const int64_t cnt = o.per_head[pp] ? nch : (int64_t)H * nch;
    double acc = 0.0;
    for (int64_t i = lane; i < cnt; i += 64) acc += (double)src[i];
#pragma unroll
    for (int sft = 32; sft > 0; sft >>= 1) acc += __shfl_xor(acc, sft);
    if (lane == 0) o.g[pp][o.per_head[pp] ? h : 0] = (float)acc;
}

// Ticket counters for the "last block finishes" reductions: zeroed device words, handed out round-robin at enqueue time (the
// launch's last block puts its words back to zero).  A word may be re-issued only to a launch that runs AFTER the one holding
// it, which stream order guarantees on ONE stream; launches enqueued far ahead on two streams (the calibrator's lanes) could
// otherwise hold the same words at the same time.  So the ring of a device is cut into SUB sub-rings and every stream gets its
// own (first come, first served; streams beyond SUB share the last one, as all streams did before).
constexpr int RING = 1024;
constexpr int SUB = 4;                      // sub-rings per device (streams that enqueue "last block finishes" kernels concurrently)
constexpr int SUBLEN = RING / SUB;          // 256 words: >= 4 requests of the largest size (64) before a word comes round again
constexpr int MAX_DEV = 16;
// One ring per device (a ring allocated on device 0 would be a foreign pointer for a kernel on device 1); the table is
// guarded by a mutex on every call (a few ns against a kernel launch).  The first call for a device allocates and
// synchronises, which a stream capture does not survive: adalog_brecq_init() makes that call ahead of any capture.
unsigned int* ticket_slots(int n, void* stream) {
    static unsigned int* rings[MAX_DEV] = {};
    static unsigned next[MAX_DEV][SUB] = {};
    static void* owner[MAX_DEV][SUB] = {};
    static int owners[MAX_DEV] = {};
    static std::mutex mu;
    int dev = 0;
    if (n < 1 || n > 64 || hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAX_DEV) return nullptr;
    std::lock_guard<std::mutex> lk(mu);
    if (!rings[dev]) {
        unsigned int* p = nullptr;
        if (hipMalloc(&p, RING * sizeof(unsigned int)) != hipSuccess || hipMemset(p, 0, RING * sizeof(unsigned int)) != hipSuccess
            || hipDeviceSynchronize() != hipSuccess)
            return nullptr;
        rings[dev] = p;
    }
    int sub = -1;
    for (int i = 0; i < owners[dev]; ++i)
        if (owner[dev][i] == stream) { sub = i; break; }
    if (sub < 0) {
        if (owners[dev] < SUB) { sub = owners[dev]++; owner[dev][sub] = stream; }
        else sub = SUB - 1;
    }
    unsigned& nx = next[dev][sub];
    if (nx % SUBLEN + n > SUBLEN) nx += SUBLEN - nx % SUBLEN;      // n consecutive words: do not wrap inside a request
    unsigned int* r = rings[dev] + sub * SUBLEN + nx % SUBLEN;
    nx += n;
    return r;
}
unsigned int* ticket_slot(void* stream) { return ticket_slots(1, stream); }

// Wide (two-level) tickets of this file's own reductions (last_block_arrives): TW_WORDS zeroed words each, WIDE per sub-ring.
// Launches of one stream run one after the other and every launch leaves its ticket zeroed, so a stream could reuse one ticket
// for ever; the sub-rings keep concurrently enqueuing streams (the calibrator's lanes) apart, as above.
constexpr int WIDE = 4;
unsigned int* ticket_wide(void* stream) {
    static unsigned int* rings[MAX_DEV] = {};
    static unsigned next[MAX_DEV][SUB] = {};
    static void* owner[MAX_DEV][SUB] = {};
    static int owners[MAX_DEV] = {};
    static std::mutex mu;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAX_DEV) return nullptr;
    std::lock_guard<std::mutex> lk(mu);
    if (!rings[dev]) {
        unsigned int* p = nullptr;
        const size_t bytes = (size_t)SUB * WIDE * TW_WORDS * sizeof(unsigned int);
        if (hipMalloc(&p, bytes) != hipSuccess || hipMemset(p, 0, bytes) != hipSuccess || hipDeviceSynchronize() != hipSuccess)
            return nullptr;
        rings[dev] = p;
    }
    int sub = -1;
    for (int i = 0; i < owners[dev]; ++i)
        if (owner[dev][i] == stream) { sub = i; break; }
    if (sub < 0) {
        if (owners[dev] < SUB) { sub = owners[dev]++; owner[dev][sub] = stream; }
        else sub = SUB - 1;
    }
    return rings[dev] + ((size_t)sub * WIDE + (next[dev][sub]++ % WIDE)) * TW_WORDS;
}

// Per-column tickets of the fused FPCS tails (fpcs_tail.h: one counter per output row / segment, up to 65 536 of them): a pool of
// POOL_SLABS x 65 536 zeroed words per sub-ring, handed out round-robin like the small ring -- a launch's last arrivals put their
// words back to zero, and a range comes round again only after POOL_SLABS later requests of the same stream, which run after it.
constexpr int POOL_WORDS = 65536, POOL_SLABS = 4;
unsigned int* ticket_pool(int n, void* stream) {
    static unsigned int* pools[MAX_DEV] = {};
    static unsigned next[MAX_DEV][SUB] = {};
    static void* owner[MAX_DEV][SUB] = {};
    static int owners[MAX_DEV] = {};
    static std::mutex mu;
    int dev = 0;
    if (n < 1 || n > POOL_WORDS || hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAX_DEV) return nullptr;
    std::lock_guard<std::mutex> lk(mu);
    if (!pools[dev]) {
        unsigned int* p = nullptr;
        const size_t bytes = (size_t)SUB * POOL_SLABS * POOL_WORDS * sizeof(unsigned int);
        if (hipMalloc(&p, bytes) != hipSuccess || hipMemset(p, 0, bytes) != hipSuccess || hipDeviceSynchronize() != hipSuccess)
            return nullptr;
        pools[dev] = p;
    }
    int sub = -1;
    for (int i = 0; i < owners[dev]; ++i)
        if (owner[dev][i] == stream) { sub = i; break; }
    if (sub < 0) {
        if (owners[dev] < SUB) { sub = owners[dev]++; owner[dev][sub] = stream; }
        else sub = SUB - 1;
    }
    return pools[dev] + ((size_t)sub * POOL_SLABS + (next[dev][sub]++ % POOL_SLABS)) * POOL_WORDS;
}

// lab switch (tools/lab/brecq_small_bench.py): cap on the blocks of the "last block finishes" reductions
inline int red_cap(int dflt) {
    static const int v = getenv("ADALOG_RED_CAP") ? atoi(getenv("ADALOG_RED_CAP")) : 0;
    return v > 0 && v < dflt ? v : dflt;
}

inline int grid1(int64_t n, int cap = 2048) {
    int64_t b = (n + 255) / 256;
    if (b < 1) b = 1;
    if (b > cap) b = cap;
    return (int)b;
}

}  // namespace

// One zeroed device word of the current device's ring (the launch's last block puts it back to zero): shared with the fused
// finish + top-k kernel of gemm_score.hip.  nullptr when the ring cannot be allocated.
extern "C" unsigned int* adalog_ticket_slot(void) { return ticket_slot(nullptr); }
extern "C" unsigned int* adalog_ticket_slots(int n) { return ticket_slots(n, nullptr); }
extern "C" unsigned int* adalog_ticket_slots_on(int n, void* stream) { return ticket_slots(n, stream); }
extern "C" unsigned int* adalog_ticket_pool_on(int n, void* stream) { return n <= 64 ? ticket_slots(n, stream) : ticket_pool(n, stream); }

// Allocates the current device's ticket ring (idempotent).  Call once per device before capturing BRECQ launches into a
// HIP graph: the allocation synchronises the device, which would invalidate a capture in progress.
extern "C" int adalog_brecq_init(void) {
    ADALOG_ARG_CHECK(ticket_slot(nullptr) != nullptr && ticket_wide(nullptr) != nullptr && ticket_pool(65, nullptr) != nullptr,
                     "brecq_init: cannot allocate the ticket counters");
    return 0;
}

// x blocks per row of the [rows][inner] view: enough blocks in total to fill the chip, at least 2048 elements per block
extern "C" int adalog_uniform_fq_backward_blocks(int64_t n, int64_t n_channels, int64_t inner) {
    (void)n_channels;
    const int64_t rows = inner > 0 ? n / inner : 1;
    const int64_t want = rows >= 4096 ? 1 : (4096 + rows - 1) / rows;      // blocks per row so that rows * nb >= 4096
    int64_t nb = (inner + 2047) / 2048;
    if (nb > want) nb = want;
    if (rows == 1 && nb > red_cap(4096)) nb = red_cap(4096);
    if (nb < 1) nb = 1;
    return (int)nb;
}

// y[i] = clamp(rne(x[i] / s) + rne(zp), 0, 2^bits - 1) - rne(zp)   (per-tensor s, zp: device scalars; x, y 16-byte aligned)
extern "C" int adalog_uniform_int_f32(const float* x, float* y, int64_t n, const float* scale, const float* zero_point,
                                      int n_bits, void* stream) {
    if (n == 0) return 0;
    ADALOG_ARG_CHECK(x && y && scale && zero_point && n_bits >= 2 && n_bits <= 8, "uniform_int: bad arguments");
    ADALOG_ARG_CHECK((((uintptr_t)x | (uintptr_t)y) & 15) == 0, "uniform_int: x / y must be 16-byte aligned");
    hipLaunchKernelGGL(k_uniform_int, dim3(grid1(n / 4 + 1, 4096)), dim3(256), 0, (hipStream_t)stream, x, y, n, scale, zero_point,
                       (float)((1 << n_bits) - 1));
    ADALOG_LAUNCH_CHECK("adalog_uniform_int_f32");
    return 0;
}

extern "C" int adalog_permute_heads(const float* src, float* dst, int64_t B, int64_t N, int P, int H, int D, int inverse,
                                    void* stream) {
    if (B * N * P * H * D == 0) return 0;
    ADALOG_ARG_CHECK(src && dst && B > 0 && N > 0 && P > 0 && H > 0 && D > 0 && D % 4 == 0, "permute_heads: bad arguments");
    hipLaunchKernelGGL(k_permute_heads, dim3(grid1(B * N * P * H * (D / 4), 8192)), dim3(256), 0, (hipStream_t)stream, src, dst, B,
                       N, P, H, D / 4, inverse);
    ADALOG_LAUNCH_CHECK("adalog_permute_heads");
    return 0;
}

static int qkv_fill(QkvQuant& q, const float* const* scales, const float* const* zps, const int* per_head, const int* n_bits) {
    for (int p = 0; p < 3; ++p) {
        if (!scales[p] || !zps[p] || n_bits[p] < 1 || n_bits[p] > 8) return 1;
        q.scale[p] = scales[p]; q.zp[p] = zps[p]; q.sstride[p] = per_head[p] ? 1 : 0; q.qmax[p] = (float)((1 << n_bits[p]) - 1);
    }
    return 0;
}
static int qkv_chunks(int64_t rows, int D) {
    const int rpp = 256 / (D / 4);
    int64_t c = (rows + (int64_t)rpp * 4 - 1) / ((int64_t)rpp * 4);          // >= 4 passes per block
    return (int)(c < 1 ? 1 : c > 512 ? 512 : c);
}
// Blocks per (part, head) of adalog_qkv_merge_quant_backward: its workspace holds 3 * H * that many floats.
extern "C" int adalog_qkv_quant_chunks(int64_t B, int64_t N, int D) { return (D == 32 || D == 64) ? qkv_chunks(B * N, D) : 0; }

// src [B][N][3][H][D] -> y0, y1, y2 [B][H][N][D]: split and asymmetric uniform fake-quant (part p with scales[p] / zps[p]:
// H values when per_head[p], else one).  HOST arrays of 3 pointers / flags.  D = 32 or 64.
extern "C" int adalog_qkv_split_quant(const float* src, float* y0, float* y1, float* y2, int64_t B, int64_t N, int H, int D,
                                      const float* const* scales, const float* const* zps, const int* per_head, const int* n_bits,
                                      void* stream) {
    if (B * N * H == 0) return 0;
    ADALOG_ARG_CHECK(src && y0 && y1 && y2 && scales && zps && per_head && n_bits && H >= 1 && (D == 32 || D == 64), "qkv_split_quant: bad arguments");
    QkvQuant q;
    ADALOG_ARG_CHECK(qkv_fill(q, scales, zps, per_head, n_bits) == 0, "qkv_split_quant: bad quantiser parameters");
    const dim3 grid((unsigned)qkv_chunks(B * N, D), (unsigned)(3 * H));
    if (D == 64) hipLaunchKernelGGL(k_qkv_split_quant<64>, grid, dim3(256), 0, (hipStream_t)stream, src, y0, y1, y2, B, N, H, q);
    else hipLaunchKernelGGL(k_qkv_split_quant<32>, grid, dim3(256), 0, (hipStream_t)stream, src, y0, y1, y2, B, N, H, q);
    ADALOG_LAUNCH_CHECK("adalog_qkv_split_quant");
    return 0;
}

// The gradients of adalog_qkv_split_quant: g0, g1, g2 [B][H][N][D] (null = zeros), src as above -> gx [B][N][3][H][D] (optional) and
// gscales[p] ([H] or [1], optional per part).  workspace: 3 * H * adalog_qkv_quant_chunks(B, N, D) floats.
extern "C" int adalog_qkv_merge_quant_backward(const float* g0, const float* g1, const float* g2, const float* src, float* gx,
                                               int64_t B, int64_t N, int H, int D, const float* const* scales,
                                               const float* const* zps, const int* per_head, const int* n_bits,
                                               float* const* gscales, float* workspace, void* stream) {
    if (B * N * H == 0) return 0;
    ADALOG_ARG_CHECK(src && scales && zps && per_head && n_bits && gscales && workspace && H >= 1 && (D == 32 || D == 64),
                     "qkv_merge_quant_backward: bad arguments");
    QkvQuant q;
    ADALOG_ARG_CHECK(qkv_fill(q, scales, zps, per_head, n_bits) == 0, "qkv_merge_quant_backward: bad quantiser parameters");
    const int nch = qkv_chunks(B * N, D);
    const dim3 grid((unsigned)nch, (unsigned)(3 * H));
    hipStream_t st = (hipStream_t)stream;
    if (D == 64) hipLaunchKernelGGL(k_qkv_merge_quant_bwd<64>, grid, dim3(256), 0, st, g0, g1, g2, src, gx, B, N, H, q, workspace);
    else hipLaunchKernelGGL(k_qkv_merge_quant_bwd<32>, grid, dim3(256), 0, st, g0, g1, g2, src, gx, B, N, H, q, workspace);
    // the scale gradients: part p per head -> H sums of nch partials; per tensor -> one sum of H * nch partials
    QkvGrads og;
    for (int p = 0; p < 3; ++p) { og.g[p] = gscales[p]; og.per_head[p] = per_head[p]; }
    hipLaunchKernelGGL(k_qkv_grad_finish, dim3((unsigned)(3 * H)), dim3(64), 0, st, workspace, H, nch, og);
    ADALOG_LAUNCH_CHECK("adalog_qkv_merge_quant_backward");
    return 0;
}

extern "C" int adalog_scaled_softmax(const float* x, float* y, int64_t rows, int n, float scale, void* stream) {
    if (rows == 0 || n == 0) return 0;
    ADALOG_ARG_CHECK(x && y && rows > 0 && n >= 1 && n <= 64 * SM_MAX, "scaled_softmax: rows of 1 .. 1024 values");
    const dim3 grid((unsigned)((rows + 3) / 4));
    hipStream_t st = (hipStream_t)stream;
    if (n <= 64) hipLaunchKernelGGL(k_scaled_softmax<1>, grid, dim3(256), 0, st, x, y, rows, n, scale);
    else if (n <= 128) hipLaunchKernelGGL(k_scaled_softmax<2>, grid, dim3(256), 0, st, x, y, rows, n, scale);
    else if (n <= 256) hipLaunchKernelGGL(k_scaled_softmax<4>, grid, dim3(256), 0, st, x, y, rows, n, scale);
    else if (n <= 512) hipLaunchKernelGGL(k_scaled_softmax<8>, grid, dim3(256), 0, st, x, y, rows, n, scale);
    else hipLaunchKernelGGL(k_scaled_softmax<16>, grid, dim3(256), 0, st, x, y, rows, n, scale);
    ADALOG_LAUNCH_CHECK("adalog_scaled_softmax");
    return 0;
}

extern "C" int adalog_scaled_softmax_backward(const float* gy, const float* y, float* gx, int64_t rows, int n, float scale,
                                              void* stream) {
    if (rows == 0 || n == 0) return 0;
    ADALOG_ARG_CHECK(gy && y && gx && rows > 0 && n >= 1 && n <= 64 * SM_MAX, "scaled_softmax_backward: rows of 1 .. 1024 values");
    const dim3 grid((unsigned)((rows + 3) / 4));
    hipStream_t st = (hipStream_t)stream;
    if (n <= 64) hipLaunchKernelGGL(k_scaled_softmax_bwd<1>, grid, dim3(256), 0, st, gy, y, gx, rows, n, scale);
    else if (n <= 128) hipLaunchKernelGGL(k_scaled_softmax_bwd<2>, grid, dim3(256), 0, st, gy, y, gx, rows, n, scale);
    else if (n <= 256) hipLaunchKernelGGL(k_scaled_softmax_bwd<4>, grid, dim3(256), 0, st, gy, y, gx, rows, n, scale);
    else if (n <= 512) hipLaunchKernelGGL(k_scaled_softmax_bwd<8>, grid, dim3(256), 0, st, gy, y, gx, rows, n, scale);
    else hipLaunchKernelGGL(k_scaled_softmax_bwd<16>, grid, dim3(256), 0, st, gy, y, gx, rows, n, scale);
    ADALOG_LAUNCH_CHECK("adalog_scaled_softmax_backward");
    return 0;
}

extern "C" int adalog_merge_heads(const float* p0, const float* p1, const float* p2, const float* p3, float* dst, int64_t B,
                                  int64_t N, int P, int H, int D, void* stream) {
    if (B * N * P * H * D == 0) return 0;
    ADALOG_ARG_CHECK(dst && B > 0 && N > 0 && P > 0 && P <= 4 && H > 0 && D > 0 && D % 4 == 0, "merge_heads: bad arguments");
    ADALOG_ARG_CHECK((((uintptr_t)p0 | (uintptr_t)p1 | (uintptr_t)p2 | (uintptr_t)p3 | (uintptr_t)dst) & 15) == 0,
                     "merge_heads: tensors must be 16-byte aligned");
    HeadParts hp = {{p0, p1, p2, p3}};
    hipLaunchKernelGGL(k_merge_heads, dim3(grid1(B * N * P * H * (D / 4), 8192)), dim3(256), 0, (hipStream_t)stream, hp, dst, B, N, P,
                       H, D / 4);
    ADALOG_LAUNCH_CHECK("adalog_merge_heads");
    return 0;
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
    const int nb = adalog_uniform_fq_backward_blocks(n, n_channels, inner);
    int gy_ = (int)(rows < 4096 ? rows : 4096);
    hipStream_t st = (hipStream_t)stream;
    float* ps = (gscale || gzp) ? workspace : nullptr;
    float* pz = (gzp && !symmetric) ? workspace + rows * nb : nullptr;
    // per-tensor parameters: the kernel's last block reduces the partials itself (no finish launches)
    unsigned int* ticket = (ps && n_channels == 1) ? ticket_wide(stream) : nullptr;
    ADALOG_ARG_CHECK(!(ps && n_channels == 1) || ticket, "uniform_fq_backward: cannot allocate the ticket counters");
    hipLaunchKernelGGL(k_uniform_bwd, dim3(nb, gy_), dim3(256), 0, st, gy, x, gx, rows, inner, scale,
                       symmetric ? nullptr : zero_point, n_channels, qmin, qmax, ps, pz, ticket, gscale, gzp);
    ADALOG_LAUNCH_CHECK("adalog_uniform_fq_backward");
    if (!ticket) {
        if (gscale) hipLaunchKernelGGL(k_param_grad_finish, dim3((unsigned)n_channels), dim3(64), 0, st, ps, rows, nb, n_channels, gscale);
        if (pz) hipLaunchKernelGGL(k_param_grad_finish, dim3((unsigned)n_channels), dim3(64), 0, st, pz, rows, nb, n_channels, gzp);
        ADALOG_LAUNCH_CHECK("adalog_uniform_fq_backward/finish");
    }
    return 0;
}

// gscale: [1] optional; workspace: 1024 floats
extern "C" int adalog_log_fq_backward_pre(const float* gy, const float* x, const float* y, float* gx, int64_t n,
                                          const float* scale, const int64_t* q, int n_bits, const float* shift, int sub_shift,
                                          float* gscale, float* workspace, int pre, void* stream);
extern "C" int adalog_log_fq_backward(const float* gy, const float* x, const float* y, float* gx, int64_t n,
                                      const float* scale, const int64_t* q, int n_bits, const float* shift, int sub_shift,
                                      float* gscale, float* workspace, void* stream) {
    return adalog_log_fq_backward_pre(gy, x, y, gx, n, scale, q, n_bits, shift, sub_shift, gscale, workspace, 0, stream);
}
// pre = 1: backward of adalog_log_fake_quant_f32_pre -- x is the GELU's INPUT; gx = dL/dx through the quantiser (STE) and the GELU
extern "C" int adalog_log_fq_backward_pre(const float* gy, const float* x, const float* y, float* gx, int64_t n,
                                          const float* scale, const int64_t* q, int n_bits, const float* shift, int sub_shift,
                                          float* gscale, float* workspace, int pre, void* stream) {
    if (n == 0) return 0;
    ADALOG_ARG_CHECK(gy && x && y && scale && q, "log_fq_backward: bad arguments");
    ADALOG_ARG_CHECK(!gscale || workspace, "log_fq_backward: the scale gradient needs a workspace");
    const int nb = grid1(n, red_cap(1024));
    hipStream_t st = (hipStream_t)stream;
    unsigned int* ticket = gscale ? ticket_wide(stream) : nullptr;
    ADALOG_ARG_CHECK(!gscale || ticket, "log_fq_backward: cannot allocate the ticket counters");
    hipLaunchKernelGGL(k_adalog_bwd, dim3(nb), dim3(256), 0, st, gy, x, y, gx, n, scale, q, 1 << n_bits, shift, sub_shift,
                       gscale ? workspace : nullptr, ticket, gscale, pre ? 1 : 0);
    ADALOG_LAUNCH_CHECK("adalog_log_fq_backward");
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

// loss[0] = sum_i (1 - |2 h(alpha_i) - 1|^b) (when loss != null).  When galpha != null, with
// g_i = gscale * (gmul ? gmul[0] : 1) * dloss/dalpha_i:  galpha[i] = g_i (overwrite) or galpha[i] += g_i.
// workspace: 1024 floats (needed for the loss value only)
extern "C" int adalog_adaround_t(const float* w, const float* alpha, float* out, float* out_t, int64_t rows, int64_t inner,
                                 const float* scale, const float* zero_point, int n_bits, int soft, void* stream) {
    if (rows * inner == 0) return 0;
    ADALOG_ARG_CHECK(w && alpha && out && out_t && scale && zero_point, "adaround_t: bad arguments");
    hipLaunchKernelGGL(k_adaround_t, dim3((unsigned)((inner + 31) / 32), (unsigned)((rows + 31) / 32)), dim3(256), 0, (hipStream_t)stream,
                       w, alpha, out, out_t, rows, inner, scale, zero_point, (float)((1 << n_bits) - 1), soft);
    ADALOG_LAUNCH_CHECK("adalog_adaround_t");
    return 0;
}

extern "C" int adalog_round_loss(const float* alpha, int64_t n, float b, const float* b_dev, float* loss, float* galpha,
                                 float gscale, const float* gmul, int overwrite, float* workspace, void* stream) {
    ADALOG_ARG_CHECK(alpha && n >= 1 && (loss == nullptr || workspace), "round_loss: bad arguments");
    const int nb = grid1(n, 1024);
    hipStream_t st = (hipStream_t)stream;
    unsigned int* ticket = loss ? ticket_wide(stream) : nullptr;
    ADALOG_ARG_CHECK(!loss || ticket, "round_loss: cannot allocate the ticket counters");
    hipLaunchKernelGGL(k_round_loss, dim3(nb), dim3(256), 0, st, alpha, n, b, b_dev, loss ? workspace : nullptr, galpha, gscale,
                       gmul, overwrite, ticket, loss);
    ADALOG_LAUNCH_CHECK("adalog_round_loss");
    return 0;
}

// loss[0] = scale * sum (pred - tgt)^2   (LossFunction.lp_loss with p = 2, utils/block_recon.py:186-199; the caller folds
// the 1/(batch*channels) of .sum(1).mean() and the /10 into `scale`).  workspace: 2048 floats.
extern "C" int adalog_rec_loss(const float* pred, const float* tgt, int64_t n, float scale, float* loss, float* workspace,
                               void* stream) {
    ADALOG_ARG_CHECK(pred && tgt && loss && workspace && n >= 1, "rec_loss: bad arguments");
    ADALOG_ARG_CHECK((((uintptr_t)pred | (uintptr_t)tgt) & 15) == 0, "rec_loss: pred / tgt must be 16-byte aligned");
    unsigned int* ticket = ticket_wide(stream);
    ADALOG_ARG_CHECK(ticket, "rec_loss: cannot allocate the ticket counters");
    hipLaunchKernelGGL(k_rec_loss, dim3(grid1(n / 4 + 1, red_cap(2048))), dim3(256), 0, (hipStream_t)stream, pred, tgt, n, scale, workspace,
                       ticket, loss);
    ADALOG_LAUNCH_CHECK("adalog_rec_loss");
    return 0;
}

// gpred = 2 * scale * gmul[0] * (pred - tgt);  gmul: device fp32 [1], the upstream gradient of the scalar loss
extern "C" int adalog_rec_loss_backward(const float* pred, const float* tgt, int64_t n, float scale, const float* gmul,
                                        float* gpred, void* stream) {
    ADALOG_ARG_CHECK(pred && tgt && gmul && gpred && n >= 1, "rec_loss_backward: bad arguments");
    hipLaunchKernelGGL(k_rec_loss_bwd, dim3(grid1(n, 4096)), dim3(256), 0, (hipStream_t)stream, pred, tgt, n, 2.0f * scale, gmul,
                       gpred);
    ADALOG_LAUNCH_CHECK("adalog_rec_loss_backward");
    return 0;
}

// dst_in[b] = src_in[idx[b]], dst_out[b] = src_out[idx[b]] (rows of row_in / row_out floats, multiples of 4, 16-byte aligned tensors),
// and sched_dev[0..n_sched) = sched_row[0..n_sched) (n_sched <= 8; sched_row may be null with n_sched = 0), in one launch.
extern "C" int adalog_brecq_prepare(const float* src_in, const float* src_out, const int64_t* idx, float* dst_in, float* dst_out,
                                    int64_t bs, int64_t row_in, int64_t row_out, const float* sched_row, float* sched_dev, int n_sched,
                                    void* stream) {
    ADALOG_ARG_CHECK(src_in && src_out && idx && dst_in && dst_out && bs >= 1 && row_in >= 4 && row_out >= 4 && row_in % 4 == 0 &&
                     row_out % 4 == 0, "brecq_prepare: rows must be multiples of 4 floats");
    ADALOG_ARG_CHECK(((((uintptr_t)src_in) | ((uintptr_t)src_out) | ((uintptr_t)dst_in) | ((uintptr_t)dst_out)) & 15) == 0,
                     "brecq_prepare: 16-byte aligned tensors");
    ADALOG_ARG_CHECK(n_sched >= 0 && n_sched <= 8 && (n_sched == 0 || (sched_row && sched_dev)), "brecq_prepare: bad schedule row");
    const int64_t total = bs * (row_in / 4 + row_out / 4);
    hipLaunchKernelGGL(k_brecq_prepare, dim3(grid1(total, 8192)), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const float4*>(src_in), reinterpret_cast<const float4*>(src_out), idx,
                       reinterpret_cast<float4*>(dst_in), reinterpret_cast<float4*>(dst_out), bs, row_in / 4, row_out / 4, sched_row,
                       sched_dev, n_sched);
    ADALOG_LAUNCH_CHECK("adalog_brecq_prepare");
    return 0;
}

// loss[0] = weight * sum over the `count` (<= 16) alpha tensors of sum_i (1 - |2 h(alpha_i) - 1|^b)  and
// grads[t][i] = weight * d/d alpha_t[i] of it, in one launch (LossFunction.__call__'s loop over the block's AdaRound
// quantisers, utils/block_recon.py:205-210).  alphas / grads / ns: HOST arrays of device pointers / lengths.
// workspace: adalog_round_loss_multi_workspace(ns, count) floats.
extern "C" int64_t adalog_round_loss_multi_workspace(const int64_t* ns, int count) {
    int64_t blocks = 0;
    for (int t = 0; t < count; ++t) blocks += grid1(ns[t], 256);
    return blocks;
}

extern "C" int adalog_round_loss_multi(const float* const* alphas, float* const* grads, const int64_t* ns, int count, float b,
                                       const float* b_dev, float weight, float* loss, float* workspace, const float* gate_dev,
                                       void* stream) {
    ADALOG_ARG_CHECK(alphas && ns && loss && workspace && count >= 1 && count <= RL_MAX, "round_loss_multi: bad arguments");
    RoundLossMulti a;
    int blocks = 0;
    for (int t = 0; t < count; ++t) {
        ADALOG_ARG_CHECK(alphas[t] && ns[t] >= 1, "round_loss_multi: bad tensor");
        a.alpha[t] = alphas[t];
        a.grad[t] = grads ? grads[t] : nullptr;                 // no gradients wanted (adalog_alpha_step_multi computes them itself)
        a.n[t] = ns[t];
        a.first_block[t] = blocks;
        blocks += grid1(ns[t], 256);
    }
    a.first_block[count] = blocks;
    a.count = count;
    unsigned int* ticket = ticket_wide(stream);
    ADALOG_ARG_CHECK(ticket, "round_loss_multi: cannot allocate the ticket counters");
    hipLaunchKernelGGL(k_round_loss_multi, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a, b, b_dev, weight, workspace, ticket,
                       loss, gate_dev);
    ADALOG_LAUNCH_CHECK("adalog_round_loss_multi");
    return 0;
}

// AdaRound's alpha of `count` (<= 16) layers: gradient (through w_sim from gws[t] = dL/dw_sim, plus the rounding regulariser's,
// scaled by the device scalar gmul[0]) and Adam step in one launch (reference quantizers/adaround.py:38-57 under autograd,
// utils/block_recon.py:108-125,205-210).  HOST arrays of device pointers; ns[t] = rows * inners[t] elements; gws[t] may be null.
// gmul may be null (regulariser not in the loss).  step_dev: steps taken so far (advanced here).  soft targets only.
extern "C" int adalog_alpha_step_multi(float* const* alphas, const float* const* ws, const float* const* gws, const float* const* scales,
                                       const float* const* zps, float* const* exp_avg, float* const* exp_avg_sq, const int64_t* ns,
                                       const int64_t* inners, const int* n_bits, int count, float lr, const float* lr_dev, float beta1,
                                       float beta2, float eps, float* step_dev, float b, const float* b_dev, float weight,
                                       const float* gmul, const float* gate, void* stream) {
    ADALOG_ARG_CHECK(alphas && ws && gws && scales && zps && exp_avg && exp_avg_sq && ns && inners && n_bits && step_dev && count >= 1 &&
                     count <= ADAM_MAX, "alpha_step_multi: bad arguments");
    AlphaStepMulti a;
    int blocks = 0;
    int64_t total = 0;
    for (int t = 0; t < count; ++t) total += ns[t] > 0 ? ns[t] : 0;
    const int epb = total >= ((int64_t)4 << 20) ? 4096 : 1024;
    for (int t = 0; t < count; ++t) {
        ADALOG_ARG_CHECK(alphas[t] && ws[t] && scales[t] && zps[t] && exp_avg[t] && exp_avg_sq[t] && ns[t] >= 1 && inners[t] >= 1 &&
                         ns[t] % inners[t] == 0, "alpha_step_multi: bad tensor");
        a.alpha[t] = alphas[t]; a.w[t] = ws[t]; a.gw[t] = gws[t]; a.scale[t] = scales[t]; a.zp[t] = zps[t];
        a.m[t] = exp_avg[t]; a.v[t] = exp_avg_sq[t]; a.n[t] = ns[t]; a.inner[t] = inners[t];
        a.qmax[t] = (float)((1 << n_bits[t]) - 1);
        a.first_block[t] = blocks;
        blocks += (int)((ns[t] + epb - 1) / epb);
    }
    a.first_block[count] = blocks;
    a.count = count;
    hipStream_t st = (hipStream_t)stream;
    unsigned int* ticket = ticket_wide(stream);
    ADALOG_ARG_CHECK(ticket, "alpha_step_multi: cannot allocate the ticket counters");
    hipLaunchKernelGGL(k_alpha_step_multi, dim3(blocks), dim3(256), 0, st, a, lr, lr_dev, beta1, beta2, eps, step_dev, b, b_dev, weight,
                       gmul, gate, epb, ticket);
    ADALOG_LAUNCH_CHECK("adalog_alpha_step_multi");
    return 0;
}

// One Adam step for `count` (<= 16) fp32 tensors in one launch (torch.optim.Adam with default options; reference
// utils/block_recon.py:108-109,122-125).  params / grads / exp_avg / exp_avg_sq / ns: HOST arrays of device pointers / lengths.
// step_dev: device fp32 [1], steps taken so far (advanced by one here).  lr_dev: optional device fp32 [1] overriding lr.
extern "C" int adalog_adam_multi(float* const* params, const float* const* grads, float* const* exp_avg, float* const* exp_avg_sq,
                                 const int64_t* ns, int count, float lr, const float* lr_dev, float beta1, float beta2, float eps,
                                 float* step_dev, void* stream) {
    ADALOG_ARG_CHECK(params && grads && exp_avg && exp_avg_sq && ns && step_dev && count >= 1 && count <= ADAM_MAX, "adam_multi: bad arguments");
    AdamMulti a;
    int blocks = 0;
    for (int t = 0; t < count; ++t) {
        ADALOG_ARG_CHECK(params[t] && grads[t] && exp_avg[t] && exp_avg_sq[t] && ns[t] >= 1, "adam_multi: bad tensor");
        a.param[t] = params[t]; a.grad[t] = grads[t]; a.m[t] = exp_avg[t]; a.v[t] = exp_avg_sq[t]; a.n[t] = ns[t];
        a.first_block[t] = blocks;
        blocks += (int)((ns[t] + 1023) / 1024);
    }
    a.first_block[count] = blocks;
    a.count = count;
    hipStream_t st = (hipStream_t)stream;
    unsigned int* ticket = ticket_wide(stream);
    ADALOG_ARG_CHECK(ticket, "adam_multi: cannot allocate the ticket counters");
    hipLaunchKernelGGL(k_adam_multi, dim3(blocks), dim3(256), 0, st, a, lr, lr_dev, beta1, beta2, eps, step_dev, ticket);
    ADALOG_LAUNCH_CHECK("adalog_adam_multi");
    return 0;
}
