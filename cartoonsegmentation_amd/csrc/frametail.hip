// frametail.hip -- the depth-of-field tail of the Ken Burns frame loop without host round trips (kenburns_effect.py:1042-1067).
//
// Per output frame the reference (and this repo's round-1 code) sorts the 1 M depth values to read two percentiles, reads four
// scalars back, then reads three more for the bokeh depth map: ~230 us of sort kernels plus seven host syncs per frame, 75 frames
// per video.  Here:
//   csm_percentile_pair   EXACT order statistics by a 3-pass radix select on the order-preserving integer image of the floats
//                         (16 + 8 + 8 bits; wave-aggregated counts in LDS, one table copy per XCD behind them, the last block of a
//                         pass picks the buckets: THREE launches; the four ranks of the two percentiles travel together) + numpy's
//                         linear interpolation rule; the results stay in device memory.
//                         (Round 2: 11 + 11 + 10 bits with per-block partials and separate pick kernels: six launches, 117 us.)
//   csm_colorize_gray_r_dev  colorize(value, cmap='gray_r') reading vmin / vmax from device memory, LUT applied in the kernel.
//   csm_bokeh_depth_auto  the depth map of bokeh_blur (utils/effects.py:146-163): its three scalar reductions run over the 256-bin
//                         histogram of the uint8 depth (max d; min and max of dmax - |d - focal| only depend on which values occur).
// All integer / order work: bit-exact with the sorted formulation (same elements selected), asserted in tests/test_gpu_kenburns.py.
#include "csm_common.h"

namespace {

constexpr int kBlock = 256;
constexpr int kSelGrid = 256;         // blocks of a selection pass
constexpr int kCoarse = 256;          // a 16-bit digit is counted twice: 256 coarse bins (its top 8 bits) and 65536 fine bins
constexpr int kFine = 65536;
constexpr int kWin = 4096;            // fine bins a block keeps in LDS (a window around its first element; the rest goes to HBM)
constexpr int kHistWords = kCoarse + kFine;
constexpr int kXcd = 8;               // one copy of every counting table per XCD: a line that all eight L2s update ping-pongs between
                                      // them (measured: 1.4 ns per atomic, 730 us for a plane that misses the LDS window); a line
                                      // that only one XCD's blocks touch stays in that L2

__device__ __forceinline__ unsigned ordered_key(float f) {      // monotone float -> unsigned map (total order incl. negatives)
    unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float key_to_float(unsigned k) {
    return __uint_as_float((k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k);
}

struct SelState {            // device-resident state of the 4 concurrent selections (64 bytes)
    unsigned prefix[4];      // top 16 key bits of each order statistic (written by the last block of pass 1)
    unsigned rank[4];        // its rank among the elements that share those bits
    unsigned ticket;         // blocks of the running pass that are done (the last one finishes the pass); 0 between passes
    unsigned poisoned;       // sticky: a pass found its tables / ticket not in the all-zero state the protocol starts from (see k_sel_top16)
    unsigned pad[6];
};
struct SelInit { unsigned rank[4]; };                     // the four ranks of a call

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ unsigned ld_agent(const unsigned *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// the XCD this workgroup runs on (which L2 its atomics execute in); only a placement hint -- any value gives the same counts
__device__ __forceinline__ unsigned xcd_id() {
    unsigned x;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
    return x & (unsigned)(kXcd - 1);
}
// Before a block takes its ticket its counts must have been PERFORMED (they are atomic read-modify-writes, read back by the last block
// with coherent atomic loads: single-location coherence needs no cache maintenance, only completion -- on gfx9 vmcnt covers atomics
// and stores).  A full __threadfence() here is a buffer_wbl2 per block: 256 write-backs of the L2 made each pass 35 us long.
// This hand-off is an ARCHITECTURE contract, not a language one: it holds on gfx9-family targets where device-scope atomic
// read-modify-writes execute at the memory side of the XCD L2s and are counted by vmcnt; the library is built for gfx950 only and refuses
// anything else at compile time.  The per-stream tables must be all-zero on entry: the last block re-zeroes them with plain stores, and
// every public entry point that uses them runs to completion or reports a launch error before touching them again (a device fault
// aborts the context anyway).
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "frametail.hip's last-block hand-off (s_waitcnt vmcnt(0) before a relaxed ticket atomic, HW_REG_XCC_ID) is validated for gfx950 only"
#endif
__device__ __forceinline__ void counts_done() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
// sum of one bin over the XCD copies of a table (`tab` = copy 0, copies kXcdStride words apart)
__device__ __forceinline__ unsigned ld_bin(const unsigned *tab, int64_t stride, unsigned bin) {
    unsigned c = 0u;
#pragma unroll
    for (int x = 0; x < kXcd; ++x) c += ld_agent(tab + x * stride + bin);
    return c;
}

// inclusive scan over the 256 threads of a block (wave shuffles + 4 wave totals through LDS)
__device__ __forceinline__ unsigned block_scan_256(unsigned x, unsigned *wsum /* LDS[4] */) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    unsigned v = x;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const unsigned t = __shfl_up(v, o, 64); if (lane >= o) v += t; }
    __syncthreads();                                       // wsum may still be read from the previous scan
    if (lane == 63) wsum[w] = v;
    __syncthreads();
    unsigned add = 0u;
    for (int q = 0; q < w; ++q) add += wsum[q];
    return v + add;
}

// One digit of one element per lane into a two-level histogram.  A wave reads 64 (x4) neighbouring pixels; on a depth map they
// mostly share the digit, and 64 atomics on one address serialise -- so a wave whose hit lanes all share the digit adds once.  Fine bins inside the block's LDS window and all coarse
// bins are LDS atomics (flushed once per block); fine bins outside the window go to the table in HBM directly.
struct HistTarget { unsigned *lds_fine; unsigned *lds_coarse; unsigned *g_fine; unsigned win_base; };
__device__ __forceinline__ void hist_count(const HistTarget &t, unsigned d, unsigned cnt) {
    atomicAdd(&t.lds_coarse[d >> 8], cnt);
    const unsigned w = d - t.win_base;
    if (w < (unsigned)kWin) atomicAdd(&t.lds_fine[w], cnt); else atomicAdd(&t.g_fine[d], cnt);
}
__device__ __forceinline__ void wave_hist_add(const HistTarget &t, bool hit, unsigned d) {
    const unsigned long long hits = __ballot(hit);
    if (hits == 0ull) return;                                                      // wave-uniform
    const int first = __builtin_amdgcn_readfirstlane(__ffsll((long long)hits) - 1);
    const unsigned d0 = (unsigned)__builtin_amdgcn_readlane((int)d, first);
    // ONE ballot decides: every hit lane has the first hit lane's digit -> that lane adds the count; otherwise plain atomics (a
    // leader loop over the distinct digits of a wave was measured slower: LDS same-address atomics cost less than its scalar trips)
    const int lane = threadIdx.x & 63;
    if (__ballot(hit && d == d0) == hits) { if (lane == first) hist_count(t, d0, (unsigned)__popcll(hits)); return; }
    const bool inside = d - t.win_base < (unsigned)kWin;
    if (hit && inside) hist_count(t, d, 1u);
    // digits outside the LDS window cost an L2 atomic each: there the distinct values of the wave are counted one by one
    unsigned long long todo = __ballot(hit && !inside);
#pragma unroll 1
    for (int it = 0; it < 8 && todo != 0ull; ++it) {
        const int leader = __builtin_amdgcn_readfirstlane(__ffsll((long long)todo) - 1);
        const unsigned dl = (unsigned)__builtin_amdgcn_readlane((int)d, leader);
        const unsigned long long same = __ballot(hit && d == dl) & todo;
        if (lane == leader) hist_count(t, dl, (unsigned)__popcll(same));
        todo &= ~same;
    }
    if ((todo >> lane) & 1ull) hist_count(t, d, 1u);
}

// flush of a block's LDS histograms into the tables in HBM (non-zero bins only)
__device__ __forceinline__ void hist_flush(const HistTarget &t, unsigned *g_coarse) {
    for (int i = threadIdx.x; i < kWin; i += kBlock) { const unsigned c = t.lds_fine[i]; if (c && t.win_base + i < (unsigned)kFine) atomicAdd(&t.g_fine[t.win_base + i], c); }
    for (int i = threadIdx.x; i < kCoarse; i += kBlock) { const unsigned c = t.lds_coarse[i]; if (c) atomicAdd(&g_coarse[i], c); }
}

// which of the 256 bins (one count per thread, already loaded) holds `rank`: the bin and the rank inside it, through LDS.
// All 256 threads call it; the loads of all four ranks are issued BEFORE the first call (they are coherent reads that miss every
// cache, ~2 us each: issued one after the other they made the tail of a pass 15 us long).
__device__ __forceinline__ void pick_bin(unsigned count, unsigned rank, unsigned *wsum, unsigned *res2 /* LDS[2] */) {
    const unsigned incl = block_scan_256(count, wsum), excl = incl - count;
    if (rank >= excl && rank < incl) { res2[0] = threadIdx.x; res2[1] = rank - excl; }
    __syncthreads();
}

// element loop shared by the two passes: 4 float4 loads in flight per lane, wave-converged calls of `body(live, value)`
template <class F>
__device__ __forceinline__ void for_each_value(const float *__restrict__ v, int64_t n, F body) {
    const int64_t n4 = (((uintptr_t)v & 15) == 0) ? n / 4 : 0;
    const float4 *v4 = reinterpret_cast<const float4 *>(v);
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    constexpr int kUnroll = 4;
    for (int64_t base = (int64_t)blockIdx.x * kBlock; base < n4; base += stride * kUnroll) {
        float4 q[kUnroll];
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) { const int64_t i = base + u * stride + threadIdx.x; q[u] = i < n4 ? v4[i] : float4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) {
            const bool live = base + u * stride + threadIdx.x < n4;
            body(live, q[u].x); body(live, q[u].y); body(live, q[u].z); body(live, q[u].w);
        }
    }
    for (int64_t base = n4 * 4 + (int64_t)blockIdx.x * kBlock; base < n; base += stride) {
        const int64_t i = base + threadIdx.x;
        const bool live = i < n;
        body(live, live ? v[i] : 0.0f);
    }
}

// pass 1: histogram of the TOP 16 key bits of every element; the last block to finish finds, for each of the four ranks, the
// 16-bit bucket that holds it.
__global__ __launch_bounds__(kBlock) void k_sel_top16(const float *__restrict__ v, int64_t n, SelState *__restrict__ st, SelInit init,
                                                       unsigned *__restrict__ g /* [kXcd][kHistWords], zero on entry (cleared by pass 2 of the previous call) */) {
    __shared__ unsigned hf[kWin];
    __shared__ unsigned hc[kCoarse];
    __shared__ unsigned wsum[4], sbin[4], sbefore[4], res2[2];
    __shared__ bool last;
    const int tid = threadIdx.x;
    for (int i = tid; i < kWin; i += kBlock) hf[i] = 0u;
    unsigned *gx = g + (int64_t)xcd_id() * kHistWords;                    // this XCD's copy of the pass-1 tables
    hc[tid] = 0u;
    // window: centred on the block's first element
    const int64_t first = (int64_t)blockIdx.x * kBlock * 4;
    const unsigned d_first = ordered_key(v[first < n ? first : 0]) >> 16;
    HistTarget t{hf, hc, gx + kCoarse, d_first > (unsigned)(kWin / 2) ? d_first - (unsigned)(kWin / 2) : 0u};
    __syncthreads();
    for_each_value(v, n, [&](bool live, float x) { wave_hist_add(t, live, ordered_key(x) >> 16); });
    __syncthreads();
    hist_flush(t, gx);
    counts_done();
    __syncthreads();
    if (tid == 0) last = atomicAdd(&st->ticket, 1u) == gridDim.x - 1;
    __syncthreads();
    if (!last) return;                                   // (the reads below are coherent atomic loads of atomically written counts: no fence)
    const unsigned c = ld_bin(g, kHistWords, (unsigned)tid);
    const unsigned incl = block_scan_256(c, wsum), excl = incl - c;
    // Poisoned-state check (ADVICE r03 / VERDICT r04): the hand-off assumes tables and ticket all-zero on entry.  If an earlier launch on
    // these buffers was aborted between its counts and its clean-up, stale counts or a stale ticket survive: either way the coarse bins
    // seen by the block that believes it is last do not add up to n (every element is counted exactly once).  That is detected here and
    // made LOUD: the sticky flag turns the outputs of this and every later selection on the state into NaN until the caller re-zeroes the
    // scratch buffer (csm_percentile_scratch_bytes documents the all-zero start).
    if (tid == kBlock - 1 && (int64_t)incl != n) st->poisoned = 1u;
#pragma unroll
    for (int r = 0; r < 4; ++r)
        if (init.rank[r] >= excl && init.rank[r] < incl) { sbin[r] = (unsigned)tid; sbefore[r] = excl; }
    __syncthreads();
    unsigned f[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) f[r] = ld_bin(g + kCoarse, kHistWords, sbin[r] * 256u + (unsigned)tid);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        pick_bin(f[r], init.rank[r] - sbefore[r], wsum, res2);
        if (tid == 0) { st->prefix[r] = sbin[r] * 256u + res2[0]; st->rank[r] = res2[1]; }
        __syncthreads();
    }
    if (tid == 0) st->ticket = 0u;
}

// numpy percentile, method 'linear' (numpy 1.26 _lerp): a + (b - a) t for t < 0.5, else b - (b - a)(1 - t); arithmetic in float64 on
// float32 samples, result rounded to float32 -- what depth_modules/zoedepth/utils/misc.py:118-119 gets from np.percentile
__device__ __forceinline__ void sel_finish(const unsigned *key4, double t_lo, double t_hi, float *out2) {
    const double a0 = (double)key_to_float(key4[0]), b0 = (double)key_to_float(key4[1]);
    const double a1 = (double)key_to_float(key4[2]), b1 = (double)key_to_float(key4[3]);
    const double r0 = t_lo < 0.5 ? a0 + (b0 - a0) * t_lo : b0 - (b0 - a0) * (1.0 - t_lo);
    const double r1 = t_hi < 0.5 ? a1 + (b1 - a1) * t_hi : b1 - (b1 - a1) * (1.0 - t_hi);
    out2[0] = (float)r0; out2[1] = (float)r1;
}

// one byte per lane into a block's 256-bin LDS histogram; lanes of a wave that share the byte (neighbouring pixels) are counted by one
__device__ __forceinline__ void wave_hist256_add(unsigned *h, bool live, unsigned b) {
    const unsigned long long hits = __ballot(live);
    if (hits == 0ull) return;
    const int first = __builtin_amdgcn_readfirstlane(__ffsll((long long)hits) - 1);
    const unsigned b0 = (unsigned)__builtin_amdgcn_readlane((int)b, first);
    if (__ballot(live && b == b0) == hits) { if ((int)(threadIdx.x & 63) == first) atomicAdd(&h[b0], (unsigned)__popcll(hits)); }
    else if (live) atomicAdd(&h[b], 1u);
}

// passes 2 and 3: the next 8 key bits (SHIFT = 8, then 0) of the elements that match a rank's prefix, counted in LDS only (256 bins per
// distinct prefix) whatever the distribution; the last block advances prefix / rank and clears the small tables; the final pass
// interpolates the two percentiles.  Pass 2 also clears the pass-1 tables for the next call.
// (A single 16-bit second pass was tried first: the matching elements of a real render depth plane are many and their low bits are
// spread over all 65536 bins -- 35 ... 300 us of L2 atomics per frame.)
template <int SHIFT>
__global__ __launch_bounds__(kBlock) void k_sel_digit8(const float *__restrict__ v, int64_t n, SelState *__restrict__ st, double t_lo, double t_hi,
                                                        float *__restrict__ out2, unsigned *__restrict__ g /* pass-1 tables */,
                                                        unsigned *__restrict__ g8 /* [kXcd][4][256], zero on entry, left zero */) {
    __shared__ unsigned hc[4][256];
    __shared__ unsigned wsum[4], res2[2], key4[4];
    __shared__ bool last;
    const int tid = threadIdx.x;
    if (SHIFT == 8)
        for (int i = blockIdx.x * kBlock + tid; i < kXcd * kHistWords / 4; i += gridDim.x * kBlock)
            __builtin_nontemporal_store(u32x4{0u, 0u, 0u, 0u}, &reinterpret_cast<u32x4 *>(g)[i]);
    unsigned pre[4]; int owner[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) pre[r] = st->prefix[r];
#pragma unroll
    for (int r = 0; r < 4; ++r) { owner[r] = r; for (int q = 0; q < r; ++q) if (pre[q] == pre[r]) { owner[r] = owner[q]; break; } }   // ranks with one prefix share a table
#pragma unroll
    for (int r = 0; r < 4; ++r) hc[r][tid] = 0u;
    __syncthreads();
    for_each_value(v, n, [&](bool live, float x) {
        const unsigned k = ordered_key(x), hi = k >> (SHIFT + 8), d = (k >> SHIFT) & 255u;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (owner[r] != r) continue;                                          // block-uniform
            const bool hit = live && hi == pre[r];
            if (__ballot(hit) == 0ull) continue;                                  // wave-uniform
            wave_hist256_add(hc[r], hit, d);
        }
    });
    __syncthreads();
    unsigned *g8x = g8 + xcd_id() * 1024u;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        if (owner[r] != r) continue;
        const unsigned c = hc[r][tid];
        if (c) atomicAdd(&g8x[r * 256 + tid], c);
    }
    counts_done();
    __syncthreads();
    if (tid == 0) last = atomicAdd(&st->ticket, 1u) == gridDim.x - 1;
    __syncthreads();
    if (!last) return;                                   // (the reads below are coherent atomic loads of atomically written counts: no fence)
    unsigned c[4], rk[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) { c[r] = ld_bin(g8 + owner[r] * 256, 1024, (unsigned)tid); rk[r] = st->rank[r]; }
#pragma unroll
    for (int x = 0; x < kXcd * 4; ++x) g8[x * 256 + tid] = 0u;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        pick_bin(c[r], rk[r], wsum, res2);
        if (tid == 0) { key4[r] = (pre[r] << 8) | res2[0]; st->prefix[r] = key4[r]; st->rank[r] = res2[1]; }
        __syncthreads();
    }
    if (tid == 0) {
        if (SHIFT == 0) {
            sel_finish(key4, t_lo, t_hi, out2);
            if (st->poisoned) { out2[0] = __uint_as_float(0x7fc00000u); out2[1] = __uint_as_float(0x7fc00000u); }      // (see k_sel_top16)
        }
        st->ticket = 0u;
    }
}

struct Lut256 { uint8_t v[256]; };

__global__ __launch_bounds__(kBlock) void k_colorize_dev(const float *__restrict__ v, uint8_t *__restrict__ out, int64_t n,
                                                          const float *__restrict__ vmm, Lut256 lut);
// the matplotlib index of one value (Colormap.__call__: xa = x*256; xa==256 -> 255; clip to [-1, 256]; int(); under / over -> ends)
__device__ __forceinline__ int cmap_index(float v, float vmin, float vmax) {
    const float x = vmin != vmax ? (v - vmin) / (vmax - vmin) : 0.0f;
    float xa = x * 256.0f;
    if (xa == 256.0f) xa = 255.0f;
    xa = fminf(fmaxf(xa, -1.0f), 256.0f);
    const int k = (int)xa;
    return k < 0 ? 0 : (k > 255 ? 255 : k);
}

// stats3 = {dmax, mn, mx2} exactly as utils/effects.py:146-153 computes them with float32 numpy reductions, from "which uint8 values
// occur" (present[256] in LDS); all 256 threads of a block call it
__device__ __forceinline__ void bokeh_stats_block(const int *present, float *red /* LDS[256] */, float focal, float *__restrict__ stats3) {
    const int tid = threadIdx.x;
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

// frame loop: colorize + the histogram of its uint8 output + (last block) the three scalars of the bokeh depth map -- what were
// k_colorize_dev, k_u8_hist and k_bokeh_stats (3 launches, 29 us).  hist256 ([kXcd][256], one copy per XCD: see kXcd) / ticket: zero
// on entry, left zero.  (First version: 1024 blocks flushing into ONE table = 30 K contended atomics, 71 us.)
__global__ __launch_bounds__(kBlock) void k_colorize_stats(const float *__restrict__ v, uint8_t *__restrict__ out, int64_t n,
                                                            const float *__restrict__ vmm, Lut256 lut, unsigned *__restrict__ hist256,
                                                            unsigned *__restrict__ ticket, float focal, float *__restrict__ stats3) {
    __shared__ unsigned h[256];
    __shared__ int present[256];
    __shared__ float red[256];
    __shared__ bool last;
    const int tid = threadIdx.x;
    h[tid] = 0u;
    __syncthreads();
    const float vmin = vmm[0], vmax = vmm[1];
    const bool vec = (((uintptr_t)v & 15) == 0) && (((uintptr_t)out & 3) == 0);
    const int64_t n4 = vec ? n / 4 : 0;
    for (int64_t base = (int64_t)blockIdx.x * kBlock; base < n4; base += (int64_t)gridDim.x * kBlock) {     // wave-converged trips
        const int64_t i = base + tid;
        const bool live = i < n4;
        const float4 q = live ? reinterpret_cast<const float4 *>(v)[i] : float4{0.f, 0.f, 0.f, 0.f};
        const unsigned b0 = lut.v[cmap_index(q.x, vmin, vmax)], b1 = lut.v[cmap_index(q.y, vmin, vmax)];
        const unsigned b2 = lut.v[cmap_index(q.z, vmin, vmax)], b3 = lut.v[cmap_index(q.w, vmin, vmax)];
        if (live) reinterpret_cast<unsigned *>(out)[i] = b0 | (b1 << 8) | (b2 << 16) | (b3 << 24);
        wave_hist256_add(h, live, b0); wave_hist256_add(h, live, b1); wave_hist256_add(h, live, b2); wave_hist256_add(h, live, b3);
    }
    for (int64_t base = n4 * 4 + (int64_t)blockIdx.x * kBlock; base < n; base += (int64_t)gridDim.x * kBlock) {
        const int64_t i = base + tid;
        const bool live = i < n;
        const unsigned b = lut.v[cmap_index(live ? v[i] : 0.0f, vmin, vmax)];
        if (live) out[i] = (uint8_t)b;
        wave_hist256_add(h, live, b);
    }
    __syncthreads();
    if (h[tid]) atomicAdd(&hist256[xcd_id() * 256u + tid], h[tid]);
    counts_done();
    __syncthreads();
    if (tid == 0) last = atomicAdd(ticket, 1u) == gridDim.x - 1;
    __syncthreads();
    if (!last) return;                                   // (the reads below are coherent atomic loads of atomically written counts: no fence)
    present[tid] = ld_bin(hist256, 256, (unsigned)tid) != 0u;
#pragma unroll
    for (int x = 0; x < kXcd; ++x) hist256[x * 256 + tid] = 0u;
    __syncthreads();
    bokeh_stats_block(present, red, focal, stats3);
    if (tid == 0) *ticket = 0u;
}

// frame loop: the bokeh depth plane (utils/effects.py:146-153, :162-163) and the highlighted image (img / 255)^lightness
// (:155-156) in one pass (were k_bokeh_depth_dev + k_bokeh_highlight).  Both are functions of one byte: a block tabulates the 2 x 256
// values once (one powf per thread instead of twelve), then every lane turns 4-byte words into float4s -- loads and stores of a
// wave are contiguous.
__global__ __launch_bounds__(kBlock) void k_bokeh_prep(const uint8_t *__restrict__ d8, const uint8_t *__restrict__ img, float *__restrict__ dm,
                                                        float *__restrict__ hi, int64_t n, float focal, const float *__restrict__ stats3, float lf) {
    __shared__ float dl[256], hl[256];
    const int tid = threadIdx.x;
    {
        const float dmax = stats3[0], mn = stats3[1], mx2 = stats3[2];
        float t = dmax - fabsf((float)tid - focal); t = t - mn; t = t / mx2; t = 1.0f - t;
        dl[tid] = t * 0.0005f;
        hl[tid] = powf((float)tid / 255.0f, lf);
    }
    __syncthreads();
    const bool vec = (((uintptr_t)d8 | (uintptr_t)img) & 3) == 0 && (((uintptr_t)dm | (uintptr_t)hi) & 15) == 0;
    const int64_t nd = vec ? n / 4 : 0, ni = vec ? (n * 3) / 4 : 0;          // whole 4-byte words of the depth plane / of the image
    for (int64_t j = (int64_t)blockIdx.x * kBlock + tid; j < nd; j += (int64_t)gridDim.x * kBlock) {
        const unsigned w = reinterpret_cast<const unsigned *>(d8)[j];
        reinterpret_cast<float4 *>(dm)[j] = float4{dl[w & 255u], dl[(w >> 8) & 255u], dl[(w >> 16) & 255u], dl[w >> 24]};
    }
    for (int64_t j = (int64_t)blockIdx.x * kBlock + tid; j < ni; j += (int64_t)gridDim.x * kBlock) {
        const unsigned w = reinterpret_cast<const unsigned *>(img)[j];
        reinterpret_cast<float4 *>(hi)[j] = float4{hl[w & 255u], hl[(w >> 8) & 255u], hl[(w >> 16) & 255u], hl[w >> 24]};
    }
    if (blockIdx.x == 0) {
        for (int64_t j = nd * 4 + tid; j < n; j += kBlock) dm[j] = dl[d8[j]];
        for (int64_t j = ni * 4 + tid; j < n * 3; j += kBlock) hi[j] = hl[img[j]];
    }
}

__global__ __launch_bounds__(kBlock) void k_colorize_dev(const float *__restrict__ v, uint8_t *__restrict__ out, int64_t n,
                                                          const float *__restrict__ vmm, Lut256 lut) {
    int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    out[i] = lut.v[cmap_index(v[i], vmm[0], vmm[1])];
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

__global__ __launch_bounds__(256) void k_bokeh_stats(const unsigned *__restrict__ partial, int nblocks, float focal, float *__restrict__ stats3) {
    __shared__ float red[256];
    __shared__ int present[256];
    const int tid = threadIdx.x;
    unsigned c = 0;
    for (int b = 0; b < nblocks; ++b) c += partial[(int64_t)b * 256 + tid];
    present[tid] = c != 0u;
    __syncthreads();
    bokeh_stats_block(present, red, focal, stats3);
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

extern "C" size_t csm_percentile_scratch_bytes(void) { return sizeof(SelState) + sizeof(unsigned) * ((size_t)kXcd * kHistWords + (size_t)kXcd * 4 * 256); }

extern "C" int csm_percentile_pair(const float *value, int64_t n, double q_lo, double q_hi, float *out2, void *scratch, void *stream) {
    CSM_REQUIRE(value && out2 && scratch && n > 0 && n < (1ll << 32) && q_lo >= 0.0 && q_lo <= 100.0 && q_hi >= 0.0 && q_hi <= 100.0);
    hipStream_t st = (hipStream_t)stream;
    SelState *state = (SelState *)scratch;
    unsigned *tables = (unsigned *)((char *)scratch + sizeof(SelState)), *tables8 = tables + (size_t)kXcd * kHistWords;
    // numpy: virtual index (n - 1) q / 100, lower / upper neighbours, interpolation weight
    const double v0 = (double)(n - 1) * (q_lo / 100.0), v1 = (double)(n - 1) * (q_hi / 100.0);
    const int64_t l0 = (int64_t)v0, l1 = (int64_t)v1;
    const int64_t h0 = l0 + 1 < n ? l0 + 1 : n - 1, h1 = l1 + 1 < n ? l1 + 1 : n - 1;
    SelInit init; init.rank[0] = (unsigned)l0; init.rank[1] = (unsigned)h0; init.rank[2] = (unsigned)l1; init.rank[3] = (unsigned)h1;
    const int64_t want = (n + kBlock * 16 - 1) / (kBlock * 16);
    const unsigned grid = (unsigned)(want < 1 ? 1 : (want > kSelGrid ? kSelGrid : want));
    const double t_lo = v0 - (double)l0, t_hi = v1 - (double)l1;
    k_sel_top16<<<grid, kBlock, 0, st>>>(value, n, state, init, tables);
    int rc = csm::check_launch("k_sel_top16"); if (rc) return rc;
    k_sel_digit8<8><<<grid, kBlock, 0, st>>>(value, n, state, t_lo, t_hi, out2, tables, tables8);
    rc = csm::check_launch("k_sel_digit8"); if (rc) return rc;
    k_sel_digit8<0><<<grid, kBlock, 0, st>>>(value, n, state, t_lo, t_hi, out2, tables, tables8);
    return csm::check_launch("k_sel_digit8");
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
                                  void *tail_scratch /* csm_kenburns_frame_scratch_bytes; zeroed once by the caller */,
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
        // colourised depth + its histogram + the three scalars of the bokeh depth map (one launch), then depth plane + highlights (one launch)
        {
            Lut256 lut;
            for (int i = 0; i < 256; ++i) lut.v[i] = gray_r_lut256_host[i];
            float *stats3 = (float *)s.bdepth;
            unsigned *ticket = (unsigned *)((char *)s.bdepth + 48), *hist256 = (unsigned *)((char *)s.bdepth + 64);   // [kXcd][256]
            const unsigned blocks = (unsigned)csm::cdiv(csm::cdiv(P, 4), kBlock);
            k_colorize_stats<<<blocks < 256u ? blocks : 256u, kBlock, 0, (hipStream_t)stream>>>(rdepth, s.depth_u8, P, s.vmm, lut, hist256, ticket, focal_plane, stats3);
            rc = csm::check_launch("k_colorize_stats"); if (rc) return rc;
            k_bokeh_prep<<<blocks < 1024u ? blocks : 1024u, kBlock, 0, (hipStream_t)stream>>>(s.depth_u8, frame_u8, s.dm, s.hi, P, focal_plane, stats3, lightness);
            rc = csm::check_launch("k_bokeh_prep"); if (rc) return rc;
        }
        const double PI = 3.14159265358979323846;
        rc = csm_bokeh_pass(s.hi, s.dm, s.pa, H, W, num_samples, 0.0f, 1.0f, stream); if (rc) return rc;
        rc = csm_bokeh_pass(s.pa, s.dm, s.pb, H, W, num_samples, (float)cos(-PI / 6), (float)sin(-PI / 6), stream); if (rc) return rc;
        rc = csm_bokeh_pass_finish(s.pb, s.dm, s.blurred, H, W, num_samples, (float)cos(-PI * 5 / 6), (float)sin(-PI * 5 / 6), lightness, stream);
        if (rc) return rc;
        src = s.blurred;
    }
    return csm_crop_resize_u8(src, H, W, patch_h, patch_w, center_x, center_y, out_hwc, stream);
}
