// Hand-written LSD radix sort of fp32 keys (ascending, stable), per segment, with an optional permutation output -- what the sorted
// forms of the self-MSE searches (sorted_score.hip; reference quant_layers/linear.py:296-353) and of the Gram activation search
// (gram_act.hip; linear.py:394-430) need once per captured tensor.  Rounds 3-5 called hipCUB (a CUB-compatibility layer over
// rocPRIM) here: 510 library launches per deit_small calibration.
//
//   adalog_sort_f32(x [S][n]) -> sorted [S][n] (each segment ascending; -0 before +0, NaNs last: the order of the bit transform
//                                below), perm [S][n] (optional): perm[s][i] = index within segment s of its i-th smallest value;
//                                equal values keep their input order.
//
// Four passes of 8 bits over order-preserving uint32 keys.  The stable rank of a key inside a workgroup's tile comes from wavefront
// ballots: a wave takes 64 consecutive keys at a time ("row"); eight ballots -- one per digit bit -- give every lane the mask of the
// row's lanes that hold ITS digit, i.e. its rank among them (popcount below the lane) and the row's count of that digit; the first
// lane of each digit adds the count to the wave's running per-digit counter in LDS after every lane of the group has read it
// (the wave's LDS operations execute in program order).  A tile is laid out (wave, row, lane), so that order IS the key order.
//   * short segments (n <= 8192: per-channel activations of a ViT, weight rows): ONE launch, a workgroup per segment, the four
//     passes ping-pong between two LDS images (k_rs_small);
//   * long segments: per pass  k_rs_hist (digit counts per 8192-key tile, LDS atomics) -> k_rs_scan (a workgroup per (segment,
//     digit) turns its tiles' counts into exclusive offsets and leaves the digit's total) -> k_rs_scatter (ranks as above, digit bases
//     from the 256 totals; the tile is first sorted by digit in LDS so that the global stores of neighbouring lanes are neighbours):
//     12 launches, 48 bytes of traffic per key.
#include "common.h"

namespace {

__device__ __forceinline__ uint32_t rs_key(float f) {
    const uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float rs_unkey(uint32_t k) {
    const uint32_t u = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
    return __uint_as_float(u);
}

// the lanes of this wave (among those with `valid`) whose 8-bit digit equals this lane's
__device__ __forceinline__ unsigned long long rs_match8(unsigned d, bool valid) {
    unsigned long long m = __ballot(valid);
#pragma unroll
    for (int b = 0; b < 8; ++b) {
        const bool bit = (d >> b) & 1u;
        const unsigned long long bb = __ballot(valid && bit);
        m &= bit ? bb : ~bb;
    }
    return m;
}
__device__ __forceinline__ unsigned long long rs_lanemask_lt(int lane) { return lane == 0 ? 0ull : (~0ull >> (64 - lane)); }

// Ranks of one wave's ROWS rows of keys (row r of the wave = elements first + r * 64 + lane of the tile, count valid keys in the
// tile): pos[r] = this key's rank among the wave's keys of the same digit (earlier rows first); leaves cnt[256] (the wave's LDS
// counters, zero on entry) holding the wave's count per digit.
// (cnt is volatile: the counters carry values from one lane to another between rows, which a per-thread view of the code cannot see.)
template <int ROWS>
__device__ __forceinline__ void rs_wave_rank(const uint32_t (&key)[ROWS], int first, int count, int shift, int lane,
                                             volatile unsigned* cnt, unsigned (&pos)[ROWS]) {
    const unsigned long long lt = rs_lanemask_lt(lane);
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
        pos[r] = 0;
        if (first + r * 64 >= count) continue;                         // (wave-uniform: the row lies past the tile's keys)
        const bool valid = first + r * 64 + lane < count;
        const unsigned d = (key[r] >> shift) & 255u;
        const unsigned long long m = rs_match8(d, valid);
        const unsigned below = (unsigned)__popcll(m & lt);
        unsigned base = 0;
        if (valid) base = cnt[d];                                       // every lane of the digit's group reads ...
        if (valid && below == 0) cnt[d] = base + (unsigned)__popcll(m);   // ... before its first lane adds the row's count
        pos[r] = base + below;
    }
}

constexpr int RS_SMALL_MAX = 8192, RS_SMALL_ROWS = RS_SMALL_MAX / 256;    // rows of 64 keys per wave (4 waves)

// One workgroup (4 waves) per segment, all four passes in LDS.  LDS: kbuf [2][n] (+ ibuf [2][n] with PERM), cnt [4][256], dig [256].
template <bool PERM>
__global__ __launch_bounds__(256) void k_rs_small(const float* __restrict__ x, int n, float* __restrict__ sorted, unsigned int* __restrict__ perm) {
    extern __shared__ __attribute__((aligned(16))) unsigned char rs_lds[];
    uint32_t* kbuf = reinterpret_cast<uint32_t*>(rs_lds);
    uint32_t* ibuf = kbuf + 2 * (size_t)n;
    unsigned* cnt = reinterpret_cast<unsigned*>(ibuf + (PERM ? 2 * (size_t)n : 0));   // [4][256]
    unsigned* dig = cnt + 4 * 256;                                                     // [256]: exclusive digit offsets
    const int seg = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const float* xs = x + (int64_t)seg * n;
    const int per_wave = ((n + 255) / 256) * 64;                      // keys per wave, a multiple of 64
    const int first = w * per_wave;
    for (int i = tid; i < n; i += 256) {
        kbuf[i] = rs_key(xs[i]);
        if (PERM) ibuf[i] = (uint32_t)i;
    }
    __syncthreads();
    int cur = 0;
    for (int pass = 0; pass < 4; ++pass) {
        const int shift = 8 * pass;
        const uint32_t* kin = kbuf + (size_t)cur * n;
        uint32_t* kout = kbuf + (size_t)(cur ^ 1) * n;
        const uint32_t* iin = ibuf + (size_t)cur * n;
        uint32_t* iout = ibuf + (size_t)(cur ^ 1) * n;
        for (int i = tid; i < 4 * 256; i += 256) cnt[i] = 0;
        uint32_t key[RS_SMALL_ROWS];
        unsigned pos[RS_SMALL_ROWS];
#pragma unroll
        for (int r = 0; r < RS_SMALL_ROWS; ++r) {
            const int e = first + r * 64 + lane;
            key[r] = (r * 64 < per_wave && e < n) ? kin[e] : 0xffffffffu;
        }
        __syncthreads();
        // (rows past the wave's share are invalid: count = min(n, first + per_wave) bounds them)
        rs_wave_rank<RS_SMALL_ROWS>(key, first, min(n, first + per_wave), shift, lane, cnt + w * 256, pos);
        __syncthreads();
        {   // thread d: exclusive prefix over the waves, the digit's total
            const unsigned c0 = cnt[tid], c1 = cnt[256 + tid], c2 = cnt[512 + tid], c3 = cnt[768 + tid];
            cnt[tid] = 0; cnt[256 + tid] = c0; cnt[512 + tid] = c0 + c1; cnt[768 + tid] = c0 + c1 + c2;
            dig[tid] = c0 + c1 + c2 + c3;
        }
        __syncthreads();
        if (w == 0) {                                                  // exclusive scan of the 256 totals: four per lane
            const unsigned a0 = dig[4 * lane], a1 = dig[4 * lane + 1], a2 = dig[4 * lane + 2], a3 = dig[4 * lane + 3];
            const unsigned own = a0 + a1 + a2 + a3;
            unsigned incl = own;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const unsigned u = __shfl_up(incl, o);
                if (lane >= o) incl += u;
            }
            const unsigned ex = incl - own;
            dig[4 * lane] = ex; dig[4 * lane + 1] = ex + a0; dig[4 * lane + 2] = ex + a0 + a1; dig[4 * lane + 3] = ex + a0 + a1 + a2;
        }
        __syncthreads();
        const bool last = pass == 3;
#pragma unroll
        for (int r = 0; r < RS_SMALL_ROWS; ++r) {
            const int e = first + r * 64 + lane;
            if (r * 64 < per_wave && e < n) {
                const unsigned d = (key[r] >> shift) & 255u;
                const unsigned dst = dig[d] + cnt[w * 256 + d] + pos[r];
                if (last) {
                    sorted[(int64_t)seg * n + dst] = rs_unkey(key[r]);
                    if (PERM) perm[(int64_t)seg * n + dst] = iin[e];
                } else {
                    kout[dst] = key[r];
                    if (PERM) iout[dst] = iin[e];
                }
            }
        }
        __syncthreads();
        cur ^= 1;
    }
}

// ---- long segments: tiles of RS_TILE keys, 8 waves per workgroup
constexpr int RS_WAVES = 8, RS_ROWS = 16, RS_TILE = RS_WAVES * RS_ROWS * 64;     // 8192

// hist[(seg * 256 + d) * ntiles + tile] = keys of the tile with digit d
template <bool FIRST>
__global__ __launch_bounds__(512) void k_rs_hist(const void* __restrict__ in, int64_t n, int ntiles, int shift, unsigned* __restrict__ hist) {
    __shared__ unsigned h[256];
    const int tile = blockIdx.x, seg = blockIdx.y, tid = threadIdx.x;
    if (tid < 256) h[tid] = 0;
    __syncthreads();
    const int64_t base = (int64_t)seg * n + (int64_t)tile * RS_TILE;
    const int count = (int)min((int64_t)RS_TILE, n - (int64_t)tile * RS_TILE);
    for (int i = tid; i < count; i += 512) {
        const uint32_t k = FIRST ? rs_key(reinterpret_cast<const float*>(in)[base + i]) : reinterpret_cast<const uint32_t*>(in)[base + i];
        atomicAdd(&h[(k >> shift) & 255u], 1u);
    }
    __syncthreads();
    if (tid < 256) hist[((int64_t)seg * 256 + tid) * ntiles + tile] = h[tid];
}

// one workgroup per (digit, segment): exclusive scan of its tiles' counts in place, the digit's total to totals[seg][d]
__global__ __launch_bounds__(256) void k_rs_scan(unsigned* __restrict__ hist, int ntiles, unsigned* __restrict__ totals) {
    __shared__ unsigned sm[4];
    const int d = blockIdx.x, seg = blockIdx.y, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    unsigned* p = hist + ((int64_t)seg * 256 + d) * ntiles;
    unsigned carry = 0;
    for (int b = 0; b < ntiles; b += 256) {
        const int i = b + tid;
        const unsigned v = i < ntiles ? p[i] : 0u;
        unsigned incl = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const unsigned u = __shfl_up(incl, o);
            if (lane >= o) incl += u;
        }
        if (lane == 63) sm[w] = incl;
        __syncthreads();
        unsigned off = carry;
        for (int j = 0; j < w; ++j) off += sm[j];
        const unsigned tot = sm[0] + sm[1] + sm[2] + sm[3];
        if (i < ntiles) p[i] = off + incl - v;
        carry += tot;
        __syncthreads();
    }
    if (tid == 0) totals[seg * 256 + d] = carry;
}

// The keys of a tile are first moved to their place INSIDE the tile (sorted by this pass's digit, LDS), then written out by
// consecutive threads: a digit's keys of one tile go to consecutive global addresses, so neighbouring lanes store neighbouring
// words (the direct scatter stored 64 different lines per wave instruction: 56 us per pass for 2.4 M pairs, 0.7 TB/s).
// LDS (dynamic): cnt [8][256], goff [256], ldig [256], kbuf [RS_TILE] (+ ibuf [RS_TILE] with PERM).
template <bool FIRST, bool LAST, bool PERM>
__global__ __launch_bounds__(512) void k_rs_scatter(const void* __restrict__ in, const uint32_t* __restrict__ iin, int64_t n, int ntiles,
                                                    int shift, const unsigned* __restrict__ hist, const unsigned* __restrict__ totals,
                                                    void* __restrict__ out, uint32_t* __restrict__ iout) {
    extern __shared__ __attribute__((aligned(16))) unsigned char rs_lds2[];
    unsigned* cnt = reinterpret_cast<unsigned*>(rs_lds2);              // [RS_WAVES][256]
    unsigned* goff = cnt + RS_WAVES * 256;                             // [256] global offset of (digit, this tile) within the segment
    unsigned* ldig = goff + 256;                                       // [256] first position of the digit inside the sorted tile
    uint32_t* kbuf = ldig + 256;                                       // [RS_TILE]
    uint32_t* ibuf = kbuf + RS_TILE;                                   // [RS_TILE] (PERM)
    const int tile = blockIdx.x, seg = blockIdx.y, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    for (int i = tid; i < RS_WAVES * 256; i += 512) cnt[i] = 0;
    const int64_t sbase = (int64_t)seg * n;
    const int64_t base = sbase + (int64_t)tile * RS_TILE;
    const int count = (int)min((int64_t)RS_TILE, n - (int64_t)tile * RS_TILE);
    const int first = w * (RS_ROWS * 64);
    uint32_t key[RS_ROWS];
    unsigned pos[RS_ROWS];
#pragma unroll
    for (int r = 0; r < RS_ROWS; ++r) {
        const int e = first + r * 64 + lane;
        uint32_t k = 0xffffffffu;
        if (e < count) k = FIRST ? rs_key(reinterpret_cast<const float*>(in)[base + e]) : reinterpret_cast<const uint32_t*>(in)[base + e];
        key[r] = k;
    }
    __syncthreads();
    rs_wave_rank<RS_ROWS>(key, first, count, shift, lane, cnt + w * 256, pos);
    __syncthreads();
    if (tid < 256) {                                                   // exclusive prefix over the waves; the digit's count in this tile
        unsigned run = 0;
#pragma unroll
        for (int ww = 0; ww < RS_WAVES; ++ww) { const unsigned c = cnt[ww * 256 + tid]; cnt[ww * 256 + tid] = run; run += c; }
        ldig[tid] = run;
        goff[tid] = totals[seg * 256 + tid];
    }
    __syncthreads();
    if (w < 2) {                                                       // exclusive scans of 256 values, four per lane: wave 0 the digit
        unsigned* a = w == 0 ? goff : ldig;                            // totals of the segment, wave 1 the digit counts of the tile
        const unsigned a0 = a[4 * lane], a1 = a[4 * lane + 1], a2 = a[4 * lane + 2], a3 = a[4 * lane + 3];
        const unsigned own = a0 + a1 + a2 + a3;
        unsigned incl = own;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const unsigned u = __shfl_up(incl, o);
            if (lane >= o) incl += u;
        }
        const unsigned ex = incl - own;
        a[4 * lane] = ex; a[4 * lane + 1] = ex + a0; a[4 * lane + 2] = ex + a0 + a1; a[4 * lane + 3] = ex + a0 + a1 + a2;
    }
    __syncthreads();
    if (tid < 256) goff[tid] += hist[((int64_t)seg * 256 + tid) * ntiles + tile];
    // the tile sorted by digit, in LDS
#pragma unroll
    for (int r = 0; r < RS_ROWS; ++r) {
        const int e = first + r * 64 + lane;
        if (e < count) {
            const unsigned d = (key[r] >> shift) & 255u;
            const unsigned lp = ldig[d] + cnt[w * 256 + d] + pos[r];
            kbuf[lp] = key[r];
            if (PERM) ibuf[lp] = FIRST ? (uint32_t)((int64_t)tile * RS_TILE + e) : iin[base + e];
        }
    }
    __syncthreads();
    for (int i = tid; i < count; i += 512) {
        const uint32_t k = kbuf[i];
        const unsigned d = (k >> shift) & 255u;
        const int64_t dst = sbase + goff[d] + ((unsigned)i - ldig[d]);
        if (LAST) reinterpret_cast<float*>(out)[dst] = rs_unkey(k);
        else reinterpret_cast<uint32_t*>(out)[dst] = k;
        if (PERM) iout[dst] = ibuf[i];
    }
}

int64_t rs_al256(int64_t v) { return (v + 255) / 256 * 256; }

struct RsPlan { bool small; int ntiles; int64_t off_keys, off_idx, off_hist, off_tot, total; };

RsPlan rs_plan(int64_t S, int64_t n, int with_perm) {
    RsPlan p{};
    p.small = n <= RS_SMALL_MAX;
    p.ntiles = (int)((n + RS_TILE - 1) / RS_TILE);
    int64_t off = 0;
    if (!p.small) {
        p.off_keys = off; off += rs_al256(S * n * 4);
        p.off_idx = off; off += with_perm ? rs_al256(S * n * 4) : 0;
        p.off_hist = off; off += rs_al256(S * 256 * (int64_t)p.ntiles * 4);
        p.off_tot = off; off += rs_al256(S * 256 * 4);
    }
    p.total = off > 256 ? off : 256;
    return p;
}

}  // namespace

extern "C" int64_t adalog_sort_workspace_bytes(int64_t S, int64_t n, int with_perm) {
    if (S < 1 || n < 1 || S > 65535 || S * n >= ((int64_t)1 << 31)) return -1;
    return rs_plan(S, n, with_perm).total;
}

// x [S][n] fp32 (contiguous) -> sorted [S][n]; perm (may be null) [S][n] uint32: index within the segment.  workspace:
// adalog_sort_workspace_bytes(S, n, perm != null) bytes, 256-byte aligned.  x and sorted must not overlap.
extern "C" int adalog_sort_f32(const float* x, int64_t S, int64_t n, float* sorted, unsigned int* perm, void* workspace,
                               int64_t workspace_bytes, void* stream) {
    ADALOG_ARG_CHECK(x && sorted && S >= 1 && n >= 1 && S <= 65535 && S * n < ((int64_t)1 << 31), "sort_f32: bad arguments");
    const RsPlan p = rs_plan(S, n, perm != nullptr);
    hipStream_t st = (hipStream_t)stream;
    if (p.small) {
        const size_t lds = (size_t)n * 8 * (perm ? 2 : 1) + 5 * 256 * 4;
        static unsigned long long attr_a = 0, attr_b = 0;
        hipError_t e = perm ? adalog_max_lds(reinterpret_cast<const void*>(&k_rs_small<true>), 150 * 1024, &attr_a)
                            : adalog_max_lds(reinterpret_cast<const void*>(&k_rs_small<false>), 150 * 1024, &attr_b);
        if (e != hipSuccess) { adalog_set_error("hipFuncSetAttribute", e); return (int)e; }
        if (perm) hipLaunchKernelGGL(k_rs_small<true>, dim3((unsigned)S), dim3(256), lds, st, x, (int)n, sorted, perm);
        else hipLaunchKernelGGL(k_rs_small<false>, dim3((unsigned)S), dim3(256), lds, st, x, (int)n, sorted, perm);
        ADALOG_LAUNCH_CHECK("adalog_sort_f32");
        return 0;
    }
    ADALOG_ARG_CHECK(workspace && workspace_bytes >= p.total && ((uintptr_t)workspace & 255) == 0, "sort_f32: workspace too small / unaligned");
    uint8_t* base = (uint8_t*)workspace;
    uint32_t* kA = (uint32_t*)(base + p.off_keys);
    uint32_t* iA = perm ? (uint32_t*)(base + p.off_idx) : nullptr;
    unsigned* hist = (unsigned*)(base + p.off_hist);
    unsigned* tot = (unsigned*)(base + p.off_tot);
    uint32_t* kB = reinterpret_cast<uint32_t*>(sorted);                  // the output buffer is the second key image
    const dim3 gt((unsigned)p.ntiles, (unsigned)S), gs(256, (unsigned)S);
    // pass 0: x -> A;  1: A -> B;  2: B -> A;  3: A -> sorted (as floats)        (perm alike: iota -> iA -> perm -> iA -> perm)
    const size_t sl = (size_t)(RS_WAVES * 256 + 512) * 4 + (size_t)RS_TILE * 4 * (perm ? 2 : 1);
#define RS_PASS(FIRSTV, LASTV, SRC, ISRC, DST, IDST, SHIFT)                                                                       \
    do {                                                                                                                          \
        hipLaunchKernelGGL((k_rs_hist<FIRSTV>), gt, dim3(512), 0, st, (const void*)(SRC), n, p.ntiles, SHIFT, hist);              \
        hipLaunchKernelGGL(k_rs_scan, gs, dim3(256), 0, st, hist, p.ntiles, tot);                                                 \
        static unsigned long long attr_p = 0, attr_k = 0;                                                                         \
        hipError_t ea__ = perm ? adalog_max_lds(reinterpret_cast<const void*>(&k_rs_scatter<FIRSTV, LASTV, true>), 80 * 1024, &attr_p) \
                               : adalog_max_lds(reinterpret_cast<const void*>(&k_rs_scatter<FIRSTV, LASTV, false>), 80 * 1024, &attr_k); \
        if (ea__ != hipSuccess) { adalog_set_error("hipFuncSetAttribute", ea__); return (int)ea__; }                              \
        if (perm) hipLaunchKernelGGL((k_rs_scatter<FIRSTV, LASTV, true>), gt, dim3(512), sl, st, (const void*)(SRC), ISRC, n, p.ntiles, \
                                     SHIFT, hist, tot, (void*)(DST), IDST);                                                      \
        else hipLaunchKernelGGL((k_rs_scatter<FIRSTV, LASTV, false>), gt, dim3(512), sl, st, (const void*)(SRC), ISRC, n, p.ntiles, \
                                SHIFT, hist, tot, (void*)(DST), IDST);                                                           \
    } while (0)
    RS_PASS(true, false, x, (const uint32_t*)nullptr, kA, iA, 0);
    RS_PASS(false, false, kA, iA, kB, perm, 8);
    RS_PASS(false, false, kB, perm, kA, iA, 16);
    RS_PASS(false, true, kA, iA, sorted, perm, 24);
#undef RS_PASS
    ADALOG_LAUNCH_CHECK("adalog_sort_f32");
    return 0;
}
