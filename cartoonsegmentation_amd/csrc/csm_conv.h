// csm_conv.h -- device / host helpers shared by the convolution translation units of libcsm355 (nets.hip, wino.hip):
// scalar math of the numerical contract, NHWC views, the conv argument block, XCD-aware tile order, LDS-DMA.
#pragma once
#include "csm_common.h"
#include <mutex>

namespace csmconv {

typedef float f32x16 __attribute__((ext_vector_type(16)));

// ---- shared scalar math (restated independently in oracle/nets_oracle.c) -----------------------
__device__ __forceinline__ float csm_expf(float x) {
    x = fminf(fmaxf(x, -87.0f), 88.0f);
    float n = rintf(x * 1.44269504088896341f);
    float r = fmaf(n, -0.693145751953125f, x);
    r = fmaf(n, -1.42860682030941723212e-6f, r);
    float p = 1.9875691500e-4f;
    p = fmaf(p, r, 1.3981999507e-3f);
    p = fmaf(p, r, 8.3334519073e-3f);
    p = fmaf(p, r, 4.1665795894e-2f);
    p = fmaf(p, r, 1.6666665459e-1f);
    p = fmaf(p, r, 5.0000001201e-1f);
    float e = fmaf(p, r * r, r) + 1.0f;
    return e * __int_as_float(((int)n + 127) << 23);
}

// natural logarithm of a positive normal float (Cephes logf: mantissa in [sqrt(1/2), sqrt(2)), degree-8 polynomial); restated in the oracle
__device__ __forceinline__ float csm_logf(float x) {
    int bits = __float_as_int(x);
    int e = ((bits >> 23) & 0xff) - 126;
    float m = __int_as_float((bits & 0x007fffff) | 0x3f000000);        // [0.5, 1)
    if (m < 0.707106781186547524f) { e -= 1; m = (m + m) - 1.0f; } else { m = m - 1.0f; }
    const float z = m * m;
    float y = 7.0376836292e-2f;
    y = fmaf(y, m, -1.1514610310e-1f); y = fmaf(y, m, 1.1676998740e-1f); y = fmaf(y, m, -1.2420140846e-1f);
    y = fmaf(y, m, 1.4249322787e-1f); y = fmaf(y, m, -1.6668057665e-1f); y = fmaf(y, m, 2.0000714765e-1f);
    y = fmaf(y, m, -2.4999993993e-1f); y = fmaf(y, m, 3.3333331174e-1f);
    y = y * m * z;
    const float fe = (float)e;
    y = fmaf(fe, -2.12194440e-4f, y);
    y = fmaf(z, -0.5f, y);
    return fmaf(fe, 0.693359375f, m + y);
}
// erf, Abramowitz-Stegun 7.1.26 (|error| < 1.5e-7), on the same expf
__device__ __forceinline__ float csm_erff(float x) {
    const float ax = fabsf(x);
    const float t = 1.0f / fmaf(0.3275911f, ax, 1.0f);
    float p = 1.061405429f;
    p = fmaf(p, t, -1.453152027f); p = fmaf(p, t, 1.421413741f); p = fmaf(p, t, -0.284496736f); p = fmaf(p, t, 0.254829592f);
    const float r = 1.0f - (p * t) * csm_expf(-(ax * ax));
    return x < 0.0f ? -r : r;
}

__device__ __forceinline__ float apply_act(float v, int act, float slope) {
    switch (act) {
        case CSM_ACT_SOFTPLUS: return v > 20.0f ? v : csm_logf(1.0f + csm_expf(v));
        case CSM_ACT_GELU: return (0.5f * v) * (1.0f + csm_erff(v * 0.707106781186547524f));
        case CSM_ACT_RELU: return fmaxf(v, 0.0f);
        case CSM_ACT_SILU: return v / (1.0f + csm_expf(-v));
        case CSM_ACT_PRELU: return v >= 0.0f ? v : v * slope;
        case CSM_ACT_HSIGMOID: return fminf(fmaxf(v + 3.0f, 0.0f), 6.0f) / 6.0f;
        case CSM_ACT_SIGMOID: return 1.0f / (1.0f + csm_expf(-v));
        default: return v;
    }
}

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) and the occupancy query are per DEVICE, and FrameLanes drives run_ops from several host
// threads: the check, the preparation and its publication happen under one mutex, so a launch can never see "prepared" before the
// attribute has been set (a > 64 KB dynamic-LDS launch would fail), and the blocks-per-CU figure is kept per device.
struct KernelPrep {
    std::mutex m;
    unsigned done = 0;                 // bit d: prepared on device d
    int blocks_per_cu[32] = {};
    template <class F> int ensure(F &&prepare /* () -> resident blocks per CU (<= 0: unknown) */) {
        int d = 0;
        (void)hipGetDevice(&d);
        d &= 31;
        std::lock_guard<std::mutex> lk(m);
        if (!(done & (1u << d))) {
            const int nb = prepare();
            blocks_per_cu[d] = nb > 0 ? nb : 1;
            done |= 1u << d;
        }
        return blocks_per_cu[d];
    }
};
template <class K> static int prepare_kernel(K kernel, int threads, size_t lds) {
    const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess)            // the launch that follows fails and is reported by its own check; say WHY here (once per kernel and device)
        fprintf(stderr, "csm355: hipFuncSetAttribute(MaxDynamicSharedMemorySize = %zu) failed: %s\n", lds, hipGetErrorString(e));
    int nb = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, reinterpret_cast<const void *>(kernel), threads, lds) != hipSuccess) nb = 0;
    return nb;
}

struct View {           // NHWC view
    float *p;
    int n, h, w, c, ld;
};

struct ConvArgs {
    View in, out, res;
    const float *w, *bias, *slope;
    int kh, kw, stride, pad, dil;
    int groups, cin_g, cout_g, npad;   // npad = cout_g rounded up to 32 (packed weight rows)
    int act, res_mode;
    int M;                             // n*ho*wo
    int ncb;                           // ceil(cin_g / 32)
    int m_tiles;
    int ksplit;                        // >1: blockIdx.z = g*ksplit + s, raw partial sums go to `partial`
    float *partial;                    // [M][ksplit*cout] (groups == 1 only)
    int serial;                        // ksplit > 1 only: 1 = one block walks all runs and combines them in registers (SER kernels)
    int m_begin;                       // k_conv_dma: first output row of this launch (rows [m_begin, M)); 0 unless the launch is split
    int split;                         // launcher hint: cover the last partial round of the grid with small tiles (see launch_conv_dma_t)
    int dbg;                           // tuning aid: 1 = no global loads, 2 = no MFMA, 4 = no LDS stores, 8 = no epilogue stores
    int ngroup;                        // tile order (speed only): N tiles per group, 0 = one group (see rem_to_tile)
    unsigned dv_hw_mul, dv_hw_shr, dv_w_mul, dv_w_shr;   // magic numbers of m / (ho * wo) and rem / wo (fast_div; filled by set_fast_div)
};

// Row set-up of the LDS-DMA loaders: output row m -> (sample, oy, ox) with two divisions by run-time constants, and the validity mask of
// its kh x kw taps.  The compiler's 32-bit division is ~35 vector instructions and the tap double loop ~9 per tap; on short-K tiles (K = 288:
// 144 MFMAs per wave) the loader's set-up was a third of the 1 337 vector instructions a wave executes (profiles/r04_grouped_conv_pmc.txt).
// fast_div: q = (mulhi(n, mul) + n) >> shr with shr = ceil(log2 d), mul = floor(2^32 (2^shr - d) / d) + 1 -- exact for n < 2^31.
__device__ __forceinline__ unsigned fast_div(unsigned n, unsigned mul, unsigned shr) { return (__umulhi(n, mul) + n) >> shr; }
static void set_fast_div(unsigned d, unsigned &mul, unsigned &shr) {
    unsigned l = 0;
    while ((1ull << l) < d) ++l;
    shr = l;
    mul = (unsigned)((((1ull << l) - d) << 32) / d + 1);
}
struct RowSetup { int n, iy0, ix0; unsigned vm; };
__device__ __forceinline__ RowSetup row_setup(const ConvArgs &a, int mm, bool rv) {
    const int wo = a.out.w, howo = a.out.h * wo;
    RowSetup r;
    r.n = (int)fast_div((unsigned)mm, a.dv_hw_mul, a.dv_hw_shr);
    const int rem = mm - r.n * howo;
    const int oy = (int)fast_div((unsigned)rem, a.dv_w_mul, a.dv_w_shr), ox = rem - oy * wo;
    r.iy0 = oy * a.stride - a.pad; r.ix0 = ox * a.stride - a.pad;
    // taps: valid rows x valid columns (bit kh * kw_count + kw)
    unsigned colm = 0u, vm = 0u;
    for (int kw = 0; kw < a.kw; ++kw) { const int ix = r.ix0 + kw * a.dil; colm |= (ix >= 0 && ix < a.in.w) ? 1u << kw : 0u; }
    for (int kh = 0; kh < a.kh; ++kh) { const int iy = r.iy0 + kh * a.dil; vm |= (iy >= 0 && iy < a.in.h) ? colm << (kh * a.kw) : 0u; }
    r.vm = rv ? vm : 0u;
    return r;
}

// Block -> tile map (speed only).  Workgroups are handed to the 8 XCDs round-robin in launch order, and each XCD has its own L2:
// give XCD x a CONTIGUOUS run of tiles in the order (z, m-tile, n-tile) with n fastest, so the blocks resident on one XCD
// share a few activation tiles (read from HBM once, all their N tiles hit L2) instead of streaming the whole activation
// tensor once per N tile.  L -> (x = L%8, i = L/8) -> j = start(x) + i is a bijection because both sides split `total`
// into 8 runs whose lengths differ by at most one, longer runs first.
// Order of the tiles inside one z slice, n fastest.  With `ngroup` (a divisor of the N-tile count, chosen by the host when the layer's
// weights exceed an XCD's L2) the order is (N group, m tile, n inside the group): an XCD's contiguous run then stays inside ONE group of
// N tiles whose weight slices fit its 4 MB L2 together with the activation tiles in flight, instead of cycling through the whole weight
// tensor once per M tile (K = 1024 -> 1024 1x1 at batch 8: 4 MB of weights + 2 MB of activation tiles thrash the L2 -- 243 MB fetched
// for 56 MB of inputs, profiles/r04_conv_pmc.txt).
__device__ __forceinline__ void rem_to_tile(unsigned rem, unsigned nm, unsigned nn, int ngroup, int &mt, int &nt) {
    if (ngroup > 0) {
        const unsigned per_g = nm * (unsigned)ngroup, gi = rem / per_g, r2 = rem - gi * per_g, m = r2 / (unsigned)ngroup;
        mt = (int)m; nt = (int)(gi * (unsigned)ngroup + (r2 - m * (unsigned)ngroup));
    } else {
        mt = (int)(rem / nn); nt = (int)(rem - (rem / nn) * nn);
    }
}
__device__ __forceinline__ void block_to_tile(int &mt, int &nt, int &z, int ngroup = 0) {
    const unsigned nm = gridDim.x, nn = gridDim.y, total = nm * nn * gridDim.z;
    const unsigned L = blockIdx.x + nm * (blockIdx.y + nn * blockIdx.z);
    const unsigned x = L & 7u, i = L >> 3, q = total >> 3, r = total & 7u;
    const unsigned j = x * q + (x < r ? x : r) + i;
    const unsigned per_z = nm * nn, zz = j / per_z, rem = j - zz * per_z;
    z = (int)zz;
    rem_to_tile(rem, nm, nn, ngroup, mt, nt);
}


typedef int i32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void dma16(unsigned voff, i32x4 rsrc, unsigned lds_byte_addr) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(rsrc), "s"(lds_byte_addr) : "memory");
}

// Winograd F(2x2, 3x3) convolution (wino.hip): launches k_conv_wino for an op that carries CSM_CONV_FLAG_WINOGRAD
bool wino_eligible(const ConvArgs &a);
int launch_conv_wino(const ConvArgs &a, hipStream_t st);
// Winograd F(4x4, 3x3) convolution (wino4.hip): k_conv_wino4 for an op that carries CSM_CONV_FLAG_WINOGRAD4
bool wino4_eligible(const ConvArgs &a);
int64_t wino4_scratch_floats(int n, int h, int w, int cout);      // scratch of the row-split forms: 24 floats per 4x4 tile and channel
int launch_conv_wino4(const ConvArgs &a, hipStream_t st);
// narrow grouped 3x3 convolutions on the vector pipe (grouped.hip): k_conv_grouped for an op that carries CSM_CONV_FLAG_GROUPED
// (direct arithmetic, own weight image; csm_op.groups / cin_g / cout_g are the REAL groups there)
bool grouped_eligible(const ConvArgs &a);
int launch_conv_grouped(const ConvArgs &a, hipStream_t st);

}  // namespace csmconv
