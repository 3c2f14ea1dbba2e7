// frametail.hip -- the depth-of-field tail of the Ken Burns frame loop without host round trips (kenburns_effect.py:1042-1067).
//
// Per output frame the reference (and this repo's round-1 code) sorts the 1 M depth values to read two percentiles, reads four
// scalars back, then reads three more for the bokeh depth map: ~230 us of sort kernels plus seven host syncs per frame, 75 frames
// per video.  Here:
//   csm_percentile_pair   EXACT order statistics by a 3-pass radix select on the order-preserving integer image of the floats
//                         (11 + 11 + 10 bits; per-block LDS histograms -> per-block partials -> one small pick kernel per pass;
//                         the four ranks of the two percentiles travel together) + numpy's linear interpolation rule; the results
//                         stay in device memory.
//   csm_colorize_gray_r_dev  colorize(value, cmap='gray_r') reading vmin / vmax from device memory, LUT applied in the kernel.
//   csm_bokeh_depth_auto  the depth map of bokeh_blur (utils/effects.py:146-163): its three scalar reductions run over the 256-bin
//                         histogram of the uint8 depth (max d; min and max of dmax - |d - focal| only depend on which values occur).
// All integer / order work: bit-exact with the sorted formulation (same elements selected), asserted in tests/test_gpu_kenburns.py.
#include "csm_common.h"

namespace {

constexpr int kBlock = 256;
constexpr int kSelBlocks = 64;        // blocks of the histogram passes (partials: kSelBlocks x 4 ranks x 2048 bins; the pick kernel
                                      // reads them all: with 256 blocks it took 69 us per pass, 3x the histogram itself)
constexpr int kBins = 2048;

__device__ __forceinline__ unsigned ordered_key(float f) {      // monotone float -> unsigned map (total order incl. negatives)
    unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float key_to_float(unsigned k) {
    return __uint_as_float((k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k);
}

struct SelState {            // device-resident state of the 4 concurrent selections
    unsigned prefix[4];      // key bits fixed so far (high bits)
    unsigned rank[4];        // rank still to find among the elements matching the prefix
    unsigned ticket;         // blocks of the last pick pass that are done (the last one computes the result); 0 between calls
};
struct SelInit { unsigned rank[4]; };                     // the four ranks of a call (arguments of the first histogram pass)

// pass p: shift / bits of the digit, mask of the already fixed bits
__device__ __forceinline__ void pass_geom(int p, int &shift, int &bits) {
    if (p == 0) { shift = 21; bits = 11; } else if (p == 1) { shift = 10; bits = 11; } else { shift = 0; bits = 10; }
}

__global__ __launch_bounds__(kBlock) void k_sel_hist(const float *__restrict__ v, int64_t n, int pass, SelState *__restrict__ st, SelInit init,
                                                      unsigned *__restrict__ partial /* [4][kSelBlocks][kBins] */) {
    __shared__ unsigned h[4][kBins];
    int shift, bits; pass_geom(pass, shift, bits);
    const int nb = 1 << bits;
    unsigned pre[4];
    // pass 0 does not read the state (no bits are fixed yet): block 0 INITIALISES it for the pick kernel that follows -- what
    // used to be a separate one-thread launch (a launch costs ~4.8 us on this stack whatever it does)
#pragma unroll
    for (int r = 0; r < 4; ++r) pre[r] = pass == 0 ? 0u : st->prefix[r];
    if (pass == 0 && blockIdx.x == 0 && threadIdx.x < 4) { st->prefix[threadIdx.x] = 0u; st->rank[threadIdx.x] = init.rank[threadIdx.x]; }
    if (pass == 0 && blockIdx.x == 0 && threadIdx.x == 4) st->ticket = 0u;
    // ranks that share a prefix share a histogram (pass 0: all four)
    int owner[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) { owner[r] = r; for (int q = 0; q < r; ++q) if (pass == 0 || pre[q] == pre[r]) { owner[r] = owner[q]; break; } }
    for (int i = threadIdx.x; i < 4 * kBins; i += kBlock) (&h[0][0])[i] = 0u;
    __syncthreads();
    const unsigned himask = pass == 0 ? 0u : (0xFFFFFFFFu << (shift + bits));
    // kUnroll independent loads per trip (one load per trip left the pass latency bound: 64 dependent round trips per thread;
    // a ballot-aggregated LDS add -- one atomic per distinct bin per wave -- measured SLOWER than the plain atomic, 31 vs 25 us)
    constexpr int kUnroll = 8;
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t base = (int64_t)blockIdx.x * kBlock + threadIdx.x; base - threadIdx.x < n; base += stride * kUnroll) {
        float val[kUnroll];
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) { const int64_t i = base + u * stride; val[u] = i < n ? v[i] : 0.0f; }
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) {
            const bool live = base + u * stride < n;
            const unsigned k = ordered_key(val[u]);
            const unsigned d = (k >> shift) & (unsigned)(nb - 1);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (owner[r] != r) continue;                                   // block-uniform
                const bool hit = live && (k & himask) == (pre[r] & himask);
                // a wave reads 64 neighbouring pixels: on a depth map they mostly share the digit (always in pass 0, whose
                // digit is sign + exponent + 2 mantissa bits), and 64 LDS atomics on one address serialise.  One ballot decides:
                // every hit lane has the first hit lane's digit -> that lane adds the count; otherwise plain atomics.
                const unsigned long long hits = __ballot(hit);
                if (hits == 0ull) continue;                                    // wave-uniform
                const int first = __builtin_amdgcn_readfirstlane(__ffsll((long long)hits) - 1);
                const unsigned d0 = (unsigned)__builtin_amdgcn_readlane((int)d, first);
                if (__ballot(hit && d == d0) == hits) { if ((int)(threadIdx.x & 63) == first) atomicAdd(&h[r][d0], (unsigned)__popcll(hits)); }
                else if (hit) atomicAdd(&h[r][d], 1u);
            }
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 4 * kBins; i += kBlock) {
        const int r = i / kBins, d = i % kBins;
        if (d < nb) partial[((int64_t)r * gridDim.x + blockIdx.x) * kBins + d] = h[owner[r]][d];
    }
}

// one block per rank: column sums of the partials, scan, pick the digit that holds the rank, advance prefix / rank
__device__ __forceinline__ void sel_finish(const volatile SelState *st, double t_lo, double t_hi, float *out2);

__global__ __launch_bounds__(1024) void k_sel_pick(const unsigned *__restrict__ partial, int nblocks, int pass, SelState *__restrict__ st,
                                                    double t_lo, double t_hi, float *__restrict__ out2) {
    __shared__ unsigned col[kBins];
    __shared__ unsigned scan[1024];
    int shift, bits; pass_geom(pass, shift, bits);
    const int nb = 1 << bits, r = blockIdx.x, tid = threadIdx.x;
    const unsigned rank = st->rank[r];                          // read by everyone BEFORE the barriers; one thread rewrites it at the end
    for (int d = tid; d < nb; d += 1024) {
        const unsigned *P = partial + (int64_t)r * nblocks * kBins + d;
        unsigned acc16[16];                                      // sixteen independent load streams (integer sums: any order)
#pragma unroll
        for (int q = 0; q < 16; ++q) acc16[q] = 0u;
        int b = 0;
        for (; b + 15 < nblocks; b += 16) {
#pragma unroll
            for (int q = 0; q < 16; ++q) acc16[q] += P[(int64_t)(b + q) * kBins];
        }
        for (; b < nblocks; ++b) acc16[0] += P[(int64_t)b * kBins];
        unsigned tot = 0u;
#pragma unroll
        for (int q = 0; q < 16; ++q) tot += acc16[q];
        col[d] = tot;
    }
    __syncthreads();
    // inclusive scan over nb (<= 2048) counts: two per thread + Hillis-Steele over 1024
    const unsigned a = 2 * tid < nb ? col[2 * tid] : 0u, b2 = 2 * tid + 1 < nb ? col[2 * tid + 1] : 0u;
    scan[tid] = a + b2;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        const unsigned t = tid >= off ? scan[tid - off] : 0u;
        __syncthreads();
        scan[tid] += t;
        __syncthreads();
    }
    const unsigned before = tid ? scan[tid - 1] : 0u;           // elements in digits < 2 tid
    // the digit d with  count(< d) <= rank < count(<= d)
    if (rank >= before && rank < before + a && 2 * tid < nb) { st->prefix[r] |= (unsigned)(2 * tid) << shift; st->rank[r] = rank - before; __threadfence(); }
    else if (rank >= before + a && rank < before + a + b2 && 2 * tid + 1 < nb) { st->prefix[r] |= (unsigned)(2 * tid + 1) << shift; st->rank[r] = rank - before - a; __threadfence(); }
    if (pass != 2) return;
    // last pass: the block that finishes last turns the four order statistics into the two percentiles (was a fourth launch)
    __syncthreads();
    if (tid == 0) {
        __threadfence();
        if (atomicAdd(&st->ticket, 1u) == gridDim.x - 1) {
            __threadfence();
            sel_finish(st, t_lo, t_hi, out2);
            st->ticket = 0u;
        }
    }
}

// numpy percentile, method 'linear' (numpy 1.26 _lerp): a + (b - a) t for t < 0.5, else b - (b - a)(1 - t); arithmetic in float64 on
// float32 samples, result rounded to float32 -- what depth_modules/zoedepth/utils/misc.py:118-119 gets from np.percentile
__device__ __forceinline__ void sel_finish(const volatile SelState *st, double t_lo, double t_hi, float *out2) {
    const double a0 = (double)key_to_float(st->prefix[0]), b0 = (double)key_to_float(st->prefix[1]);
    const double a1 = (double)key_to_float(st->prefix[2]), b1 = (double)key_to_float(st->prefix[3]);
    const double r0 = t_lo < 0.5 ? a0 + (b0 - a0) * t_lo : b0 - (b0 - a0) * (1.0 - t_lo);
    const double r1 = t_hi < 0.5 ? a1 + (b1 - a1) * t_hi : b1 - (b1 - a1) * (1.0 - t_hi);
    out2[0] = (float)r0; out2[1] = (float)r1;
}

struct Lut256 { uint8_t v[256]; };

__global__ __launch_bounds__(kBlock) void k_colorize_dev(const float *__restrict__ v, uint8_t *__restrict__ out, int64_t n,
                                                          const float *__restrict__ vmm, Lut256 lut) {
    int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const float vmin = vmm[0], vmax = vmm[1];
    float x = vmin != vmax ? (v[i] - vmin) / (vmax - vmin) : 0.0f;
    // matplotlib Colormap.__call__: xa = x*256; xa==256 -> 255; clip to [-1, 256]; int(); <0 -> under (lut[0]), >255 -> over (lut[255])
    float xa = x * 256.0f;
    if (xa == 256.0f) xa = 255.0f;
    xa = fminf(fmaxf(xa, -1.0f), 256.0f);
    int k = (int)xa;
    k = k < 0 ? 0 : (k > 255 ? 255 : k);
    out[i] = lut.v[k];
}

// ---- bokeh depth: scalars from the histogram of the uint8 depth --------------------------------------------------------------
// histogram of the uint8 depth in per-block partials.  (Folding the statistics into this kernel through a last-block ticket was
// measured: 31 us against 10 + 11 us for the two launches -- the one block that reads all partials serialises the tail.)
__global__ __launch_bounds__(kBlock) void k_u8_hist(const uint8_t *__restrict__ d, int64_t n, unsigned *__restrict__ partial /* [blocks][256] */) {
    __shared__ unsigned h[256];
    h[threadIdx.x] = 0u;
    __syncthreads();
    // 16 bytes per lane and trip when the plane is 16-B aligned (it is: a torch allocation), bytes otherwise
    const int64_t n16 = (((uintptr_t)d & 15) == 0) ? n / 16 : 0;
    const uint4 *d16 = reinterpret_cast<const uint4 *>(d);
    for (int64_t base = (int64_t)blockIdx.x * kBlock; base < n16; base += (int64_t)gridDim.x * kBlock) {
        const int64_t i = base + threadIdx.x;
        const bool live = i < n16;
        const uint4 q = live ? d16[i] : uint4{0u, 0u, 0u, 0u};
        const unsigned w4[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) if (live) atomicAdd(&h[(w4[a] >> (8 * b)) & 255u], 1u);
    }
    for (int64_t base = n16 * 16 + (int64_t)blockIdx.x * kBlock; base < n; base += (int64_t)gridDim.x * kBlock) {
        const int64_t i = base + threadIdx.x;
        if (i < n) atomicAdd(&h[d[i]], 1u);
    }
    __syncthreads();
    partial[(int64_t)blockIdx.x * 256 + threadIdx.x] = h[threadIdx.x];
}

// stats3 = {dmax, mn, mx2} exactly as utils/effects.py:146-153 computes them with float32 numpy reductions
__global__ __launch_bounds__(256) void k_bokeh_stats(const unsigned *__restrict__ partial, int nblocks, float focal, float *__restrict__ stats3) {
    __shared__ float red[256];
    __shared__ int present[256];
    const int tid = threadIdx.x;
    unsigned c = 0;
    for (int b = 0; b < nblocks; ++b) c += partial[(int64_t)b * 256 + tid];
    present[tid] = c != 0u;
    __syncthreads();
    red[tid] = present[tid] ? (float)tid : -1.0f;
    __syncthreads();
    for (int s = 128; s >= 1; s >>= 1) { if (tid < s) red[tid] = fmaxf(red[tid], red[tid + s]); __syncthreads(); }
    const float dmax = red[0];
    __syncthreads();
    const float t = dmax - fabsf((float)tid - focal);           // depth = depth.max() - |depth - focal_plane|
    red[tid] = present[tid] ? t : INFINITY;
    __syncthreads();
    for (int s = 128; s >= 1; s >>= 1) { if (tid < s) red[tid] = fminf(red[tid], red[tid + s]); __syncthreads(); }
    const float mn = red[0];
    __syncthreads();
    red[tid] = present[tid] ? t - mn : -INFINITY;               // depth -= depth.min(); depth.max()
    __syncthreads();
    for (int s = 128; s >= 1; s >>= 1) { if (tid < s) red[tid] = fmaxf(red[tid], red[tid + s]); __syncthreads(); }
    if (tid == 0) { stats3[0] = dmax; stats3[1] = mn; stats3[2] = red[0]; }
}

__global__ __launch_bounds__(kBlock) void k_bokeh_depth_dev(const uint8_t *__restrict__ d8, float *__restrict__ out, int64_t n, float focal,
                                                             const float *__restrict__ stats3) {
    int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const float dmax = stats3[0], mn = stats3[1], mx2 = stats3[2];
    float v = dmax - fabsf((float)d8[i] - focal);
    v = v - mn;
    v = v / mx2;
    v = 1.0f - v;
    out[i] = v * 0.0005f;
}

}  // namespace

extern "C" size_t csm_percentile_scratch_bytes(void) { return sizeof(SelState) + 64 + sizeof(unsigned) * 4 * (size_t)kSelBlocks * kBins; }

extern "C" int csm_percentile_pair(const float *value, int64_t n, double q_lo, double q_hi, float *out2, void *scratch, void *stream) {
    CSM_REQUIRE(value && out2 && scratch && n > 0 && n < (1ll << 32) && q_lo >= 0.0 && q_lo <= 100.0 && q_hi >= 0.0 && q_hi <= 100.0);
    hipStream_t st = (hipStream_t)stream;
    SelState *state = (SelState *)scratch;
    unsigned *partial = (unsigned *)((char *)scratch + 64);
    // numpy: virtual index (n - 1) q / 100, lower / upper neighbours, interpolation weight
    const double v0 = (double)(n - 1) * (q_lo / 100.0), v1 = (double)(n - 1) * (q_hi / 100.0);
    const int64_t l0 = (int64_t)v0, l1 = (int64_t)v1;
    const int64_t h0 = l0 + 1 < n ? l0 + 1 : n - 1, h1 = l1 + 1 < n ? l1 + 1 : n - 1;
    SelInit init; init.rank[0] = (unsigned)l0; init.rank[1] = (unsigned)h0; init.rank[2] = (unsigned)l1; init.rank[3] = (unsigned)h1;
    int rc;
    for (int pass = 0; pass < 3; ++pass) {
        k_sel_hist<<<kSelBlocks, kBlock, 0, st>>>(value, n, pass, state, init, partial);
        rc = csm::check_launch("k_sel_hist"); if (rc) return rc;
        k_sel_pick<<<4, 1024, 0, st>>>(partial, kSelBlocks, pass, state, v0 - (double)l0, v1 - (double)l1, out2);
        rc = csm::check_launch("k_sel_pick"); if (rc) return rc;
    }
    return CSM_OK;
}

extern "C" int csm_colorize_gray_r_dev(const float *value, uint8_t *out, int64_t n, const float *vmin_vmax_dev, const uint8_t *lut256_host,
                                       void *stream) {
    CSM_REQUIRE(value && out && vmin_vmax_dev && lut256_host && n > 0);
    Lut256 lut;
    for (int i = 0; i < 256; ++i) lut.v[i] = lut256_host[i];
    k_colorize_dev<<<csm::cdiv(n, kBlock), kBlock, 0, (hipStream_t)stream>>>(value, out, n, vmin_vmax_dev, lut);
    return csm::check_launch("k_colorize_dev");
}

extern "C" size_t csm_bokeh_depth_scratch_bytes(void) { return 64 + sizeof(unsigned) * 64 * 256; }

extern "C" int csm_bokeh_depth_auto(const uint8_t *depth_u8, float *out, int64_t n, float focal_plane, void *scratch, void *stream) {
    CSM_REQUIRE(depth_u8 && out && scratch && n > 0);
    hipStream_t st = (hipStream_t)stream;
    float *stats = (float *)scratch;
    unsigned *partial = (unsigned *)((char *)scratch + 64);
    k_u8_hist<<<64, kBlock, 0, st>>>(depth_u8, n, partial);
    int rc = csm::check_launch("k_u8_hist"); if (rc) return rc;
    k_bokeh_stats<<<1, 256, 0, st>>>(partial, 64, focal_plane, stats);
    rc = csm::check_launch("k_bokeh_stats"); if (rc) return rc;
    k_bokeh_depth_dev<<<csm::cdiv(n, kBlock), kBlock, 0, st>>>(depth_u8, out, n, focal_plane, stats);
    return csm::check_launch("k_bokeh_depth_dev");
}

// ---- focal plane of the depth of field: the largest per-instance MEDIAN of the colourised depth (kenburns_effect.py:1045-1056) ----
// The reference gathers depth_rendered[mask] on the host and calls np.median per instance.  Values are uint8, so a 256-bin histogram
// per instance holds everything: the two middle order statistics come from its prefix sums, np.median's even-count rule (mean of
// the two) is exact in float, empty masks are skipped like the reference's `nan > x == False`.  No gather, no sort, and one
// scalar for the host to read instead of two reads per instance.
namespace {
__global__ __launch_bounds__(256) void k_masked_hist(const uint8_t *__restrict__ v, const uint8_t *__restrict__ masks, int64_t n,
                                                      unsigned *__restrict__ hist /* [inst][256], zeroed */) {
    __shared__ unsigned h[256];
    h[threadIdx.x] = 0u;
    __syncthreads();
    const uint8_t *m = masks + (int64_t)blockIdx.y * n;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
        if (m[i]) atomicAdd(&h[v[i]], 1u);
    __syncthreads();
    if (h[threadIdx.x]) atomicAdd(&hist[(int64_t)blockIdx.y * 256 + threadIdx.x], h[threadIdx.x]);
}
__global__ __launch_bounds__(256) void k_masked_median_max(const unsigned *__restrict__ hist, int n_inst, float *__restrict__ out /* [n_inst + 1] */) {
    __shared__ unsigned cum[256];
    __shared__ float best;
    if (threadIdx.x == 0) best = -1.0f;                      // focalplane_end = -1
    __syncthreads();
    for (int k = 0; k < n_inst; ++k) {
        cum[threadIdx.x] = hist[k * 256 + threadIdx.x];
        __syncthreads();
        if (threadIdx.x == 0) {
            unsigned run = 0;
            for (int b = 0; b < 256; ++b) { run += cum[b]; cum[b] = run; }          // inclusive prefix (256 steps, once per instance)
            const unsigned cnt = run;
            float med = __uint_as_float(0x7FC00000u);        // np.median of an empty selection: nan
            if (cnt > 0) {
                const unsigned r1 = (cnt - 1) / 2, r2 = cnt / 2;                     // 0-based ranks of the two middle values
                int v1 = 0, v2 = 0;
                while (cum[v1] <= r1) ++v1;
                while (cum[v2] <= r2) ++v2;
                med = ((float)v1 + (float)v2) / 2.0f;
                if (med > best) best = med;
            }
            out[k] = med;
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) out[n_inst] = best;
}
}  // namespace

extern "C" int csm_masked_u8_median_max(const uint8_t *values, const uint8_t *masks, int n_inst, int64_t n, unsigned *hist, float *out,
                                        void *stream) {
    CSM_REQUIRE(values && masks && hist && out && n_inst > 0 && n > 0);
    hipStream_t st = (hipStream_t)stream;
    CSM_HIP(hipMemsetAsync(hist, 0, sizeof(unsigned) * 256 * (size_t)n_inst, st));
    const unsigned blocks = (unsigned)((n + 256 * 16 - 1) / (256 * 16) < 256 ? ((n + 256 * 16 - 1) / (256 * 16) > 0 ? (n + 256 * 16 - 1) / (256 * 16) : 1) : 256);
    k_masked_hist<<<dim3(blocks, (unsigned)n_inst), 256, 0, st>>>(values, masks, n, hist);
    int rc = csm::check_launch("k_masked_hist"); if (rc) return rc;
    k_masked_median_max<<<1, 256, 0, st>>>(hist, n_inst, out);
    return csm::check_launch("k_masked_median_max");
}

// ---- one output frame of the Ken Burns loop as ONE call (kenburns_effect.py:1027-1072) ----------------------------------------------
// warp (shift -> splat -> fill -> uint8) [-> colourised depth -> depth-of-field blur] -> crop + resize into the video buffer.  The
// kernels are the ones behind the separate entry points, launched in the same order with the same arguments; what goes away is
// the host side of ~12 calls and ~8 allocations per frame, which had become the limit of the frame loop once consecutive frames ran
// on several streams (round 3: 75 bokeh frames were host bound at ~275 us each).  All scratch is caller-owned.
namespace {
inline size_t a256(size_t v) { return (v + 255) & ~(size_t)255; }
struct FrameScratch { float *vmm; void *sel; void *bdepth; uint8_t *depth_u8; float *dm, *hi, *pa, *pb; uint8_t *blurred; size_t total; };
inline FrameScratch carve_frame_scratch(void *base, int H, int W) {
    const size_t P = (size_t)H * W;
    FrameScratch s; char *p = (char *)base; size_t off = 0;
    auto take = [&](size_t bytes) { char *q = p ? p + off : nullptr; off += a256(bytes); return (void *)q; };
    s.vmm = (float *)take(64);
    s.sel = take(csm_percentile_scratch_bytes());
    s.bdepth = take(csm_bokeh_depth_scratch_bytes());
    s.depth_u8 = (uint8_t *)take(P);
    s.dm = (float *)take(P * 4);
    s.hi = (float *)take(P * 12); s.pa = (float *)take(P * 12); s.pb = (float *)take(P * 12);
    s.blurred = (uint8_t *)take(P * 3);
    s.total = off;
    return s;
}
}  // namespace

extern "C" size_t csm_kenburns_frame_scratch_bytes(int H, int W) { return (H <= 0 || W <= 0) ? 0 : carve_frame_scratch(nullptr, H, W).total; }

extern "C" int csm_kenburns_frame(const float *pts, const float *rgb, const float *depth, int64_t N, int H, int W, double focal, double baseline,
                                  float sx, float sy, float sz, void *warp_scratch, float *render /* [4,H,W]; required when dof */,
                                  uint8_t *frame_u8 /* [H,W,3] warp output (scratch of the caller's lane) */,
                                  int dof, float focal_plane, int num_samples, float lightness, const uint8_t *gray_r_lut256_host,
                                  void *tail_scratch /* csm_kenburns_frame_scratch_bytes; its bokeh-depth part zeroed once */,
                                  int patch_h, int patch_w, float center_x, float center_y, uint8_t *out_hwc, void *stream) {
    CSM_REQUIRE(frame_u8 && out_hwc && H > 0 && W > 0);
    CSM_REQUIRE(!dof || (render && tail_scratch && gray_r_lut256_host && num_samples > 0));
    int rc = csm_warp_frame_tiled(pts, rgb, depth, N, H, W, focal, baseline, sx, sy, sz, warp_scratch, render, frame_u8, stream);
    if (rc) return rc;
    const uint8_t *src = frame_u8;
    if (dof) {
        const FrameScratch s = carve_frame_scratch(tail_scratch, H, W);
        const int64_t P = (int64_t)H * W;
        const float *rdepth = render + 3 * P;                                              // tenRender[0, 3]
        // colorize(depth, cmap='gray_r')[..., 0] with the 2nd / 85th percentiles (zoedepth/utils/misc.py:97-135)
        rc = csm_percentile_pair(rdepth, P, 2.0, 85.0, s.vmm, s.sel, stream); if (rc) return rc;
        rc = csm_colorize_gray_r_dev(rdepth, s.depth_u8, P, s.vmm, gray_r_lut256_host, stream); if (rc) return rc;
        // bokeh_blur(frame, depth_u8, num_samples, lightness, focal_plane=..., depth_factor=1)  (utils/effects.py:143-181)
        rc = csm_bokeh_depth_auto(s.depth_u8, s.dm, P, focal_plane, s.bdepth, stream); if (rc) return rc;
        rc = csm_bokeh_highlight(frame_u8, s.hi, P * 3, lightness, stream); if (rc) return rc;
        const double PI = 3.14159265358979323846;
        rc = csm_bokeh_pass(s.hi, s.dm, s.pa, H, W, num_samples, 0.0f, 1.0f, stream); if (rc) return rc;
        rc = csm_bokeh_pass(s.pa, s.dm, s.pb, H, W, num_samples, (float)cos(-PI / 6), (float)sin(-PI / 6), stream); if (rc) return rc;
        rc = csm_bokeh_pass_finish(s.pb, s.dm, s.blurred, H, W, num_samples, (float)cos(-PI * 5 / 6), (float)sin(-PI * 5 / 6), lightness, stream);
        if (rc) return rc;
        src = s.blurred;
    }
    return csm_crop_resize_u8(src, H, W, patch_h, patch_w, center_x, center_y, out_hwc, stream);
}
