// warptile.hip -- one Ken Burns output frame (kenburns_effect.py:1027-1040) with the splat done per DESTINATION TILE in LDS.
//
// Why: the round-1 chain (warp.hip: fill, updateZee, degrid, updateOutput, finalize, fill-holes) spends half of its 141 us in
// updateOutput, whose 20 fp32 atomics per point go to the L2 atomic units (~0.25 T lane-atomics/s on gfx950), 50x below what the
// LDS of 256 CUs sustains, and a wave of consecutive source pixels lands in one or two destination tiles anyway.  Three kernels:
//
//   k_tile_bin      ONE pass over the points: process_shift + projection (warp_device.h, the reference's statements), the <= 4
//                   32 x 16 tiles whose 1-px-expanded area the 2 x 2 footprint touches; block-local LDS histogram whose atomic return
//                   value is the entry's rank inside (block, tile); the rank-0 lane of every touched tile reserves the block's run
//                   in the tile's FIXED-CAPACITY segment with one global atomic; 16-B entries {fx, fy, fltError, point index} go
//                   to segment + run + rank.  Overflow goes to a spill list (exact for any cloud).
//   k_tile_render   one 256-thread block per tile, everything else in LDS: z-buffer of the tile + 1-px ring (ds_min_i32 / ds_max_u32:
//                   float min through the sign-split integer trick), Jacobi degrid, z-test + bilinear splat of rgb / depth / weight
//                   into 64-bit FIXED-POINT accumulators (ds_add_u64: ~2x the rate of ds_add_f32 and order free, so a frame is
//                   bit-reproducible), normalise, depth mask, uint8 frame, masked-depth plane, row / column valid BITMAPS (ballot);
//                   the tile's holes are appended to the frame's flat hole list with one atomic; re-arms the tile's counter.
//   k_tile_holes    32 lanes per hole = 16 directions x {from, to} of fill_disocclusion (common.py:145-248); the four axis rays are
//                   word scans of the row / column bitmaps, the oblique rays probe the bitmap four steps per round trip and stop
//                   once they have walked further than the best complete direction found so far.
//
// No float atomic reaches L2, the accumulator planes never exist in HBM, and the z-buffer decisions are the ones of the round-1
// chain bit for bit (min is order free; the degrid is the same Jacobi form; every comparison is the reference's expression).
// Measured numbers per round: DESIGN.md section 4.1 and profiles/.
#include "warp_device.h"
#include <mutex>

namespace {
using namespace csmwarp;

constexpr int kBlock = 256;
constexpr int TW = 32, TH = 16, TPIX = TW * TH;      // destination tile (one bitmap word wide; 16 rows: 2048 blocks at 1024^2, 6 per CU)
constexpr int ZW = TW + 2, ZH = TH + 2;              // z-buffer window: tile + 1-px ring (the degrid stencil)
struct TileGeom { int ntx, nty, nt; };
__host__ __device__ inline TileGeom tile_geom(int H, int W) {
    TileGeom g; g.ntx = (W + TW - 1) / TW; g.nty = (H + TH - 1) / TH; g.nt = g.ntx * g.nty; return g;
}
__device__ __forceinline__ int floor_div(int a, int b) { return a >= 0 ? a / b : -((-a + b - 1) / b); }

// tiles whose expanded area [t*T - 1, t*T + T] meets the footprint [v0, v0 + 1]:  ceil(v0 / T) - 1 <= t <= floor((v0 + 2) / T)
__device__ __forceinline__ void tile_range(int v0, int T, int nt, int &lo, int &hi) {
    lo = floor_div(v0 + T - 1, T) - 1;
    hi = floor_div(v0 + 2, T);
    lo = lo < 0 ? 0 : lo;
    hi = hi > nt - 1 ? nt - 1 : hi;
}

struct Entry { float fx, fy, err; int idx; };        // 16 B

// Binning = counting sort without hot global counters.  (The first version used one global atomicAdd per distinct tile per wave:
// ~20 k atomics on 1024 adjacent counters = 32 cache lines; same-line device-scope atomics serialise at the memory side and the
// count pass alone took 49 us.)  Each block owns kPPB consecutive points and histograms their tiles in LDS; per tile it touches
// it reserves a run in the tile's segment with ONE global atomic (blocks x ~6 tiles, spread over all counters) and remembers
// the run's start in blockbase[block][tile].  After the scan the scatter pass seeds LDS cursors from tile_off + blockbase and
// hands out slots with LDS atomics.
#ifndef KPPT
#define KPPT 2       // measured at 1024^2 (whole chain): 1 -> 57.9 us, 2 -> 57.6, 4 -> 61.1, 8 -> 68.1 (more, shorter latency chains)
#endif
constexpr int kPPT = KPPT, kPPB = kBlock * kPPT;      // points per thread / per block
constexpr int kPatchH = kPPB / 32;                    // a block's patch of the source grid: 32 columns x kPatchH rows

struct PointBins { float fx, fy, err; int txl, txh, tyl, tyh; bool ok; };

template <bool SHIFT>
__device__ __forceinline__ PointBins bins_of_point(const float *__restrict__ pts, int64_t N, int64_t p, const ProjConst &pc, Shift s,
                                                   const TileGeom &g) {
    PointBins b; b.fx = b.fy = b.err = 0.f; b.txl = b.tyl = 0; b.txh = b.tyh = -1;
    b.ok = p < N;
    if (b.ok) {
        float x, y, z;
        load_point<SHIFT>(pts, N, p, s, x, y, z);
        b.ok = project(x, y, z, pc, b.fx, b.fy, b.err);
    }
    if (b.ok) {
        // a footprint that far out cannot touch the image; the range test also keeps (int)floorf() away from overflow (NaN fails)
        b.ok = b.fx > -4.0f && b.fy > -4.0f && b.fx < (float)(pc.W + 4) && b.fy < (float)(pc.H + 4);
        if (b.ok) {
            const int x0 = (int)floorf(b.fx), y0 = (int)floorf(b.fy);
            tile_range(x0, TW, g.ntx, b.txl, b.txh);
            tile_range(y0, TH, g.nty, b.tyl, b.tyh);
            b.ok = b.txl <= b.txh && b.tyl <= b.tyh;
        }
    }
    return b;
}

// Binning in ONE pass over the points: every tile owns a fixed-capacity segment entries[t * cap .. t * cap + cap).  A block
// histograms its kPPB points in LDS (the LDS atomic's return value is the entry's rank inside (block, tile)), the rank-0 lane of
// every touched tile reserves the block's run in the tile's segment with ONE global atomic, then every lane writes its entries at
// segment + run start + rank.  No counting pre-pass, no scan, no per-block offset table.  (Round-2 history: a separate count
// kernel + per-block scan of the totals in the scatter kernel cost 8.7 + 10 us at 1024^2; the first design with one global
// atomic per distinct tile per wave cost 49 us -- same-line device-scope atomics serialise at the memory side.)
// Capacity is ~2.5x the expected load (tile_cap below).  Entries beyond it go to a global spill list that every render block
// filters by tile id: exact for any input, slow only for pathological clouds (thousands of points in one 32 x 16 tile).
// Which point does (block, slot q of kPPB) handle?  The cloud's first H * W points are the frame's own pixels in row-major order
// (kenburns_effect.py:928-933; inpainting appends the rest), so when W is a multiple of 32 a block takes a 32-column x kPatchH-row PATCH
// of that grid instead of kPPB consecutive points: its footprints then fall into ~4 tiles instead of ~20-40 (fewer, longer runs per tile and
// 8x fewer global reservations; measured 16.9 -> 15.4 us for the bin pass, 36.2 -> 35.3 us for the render pass).  Purely a locality choice: any cloud gives the same frame under either mapping.
struct PointMap { int64_t patched; int patches_x, W; };   // points below `patched` are visited patch-wise
__host__ __device__ inline PointMap make_point_map(int H, int W, int64_t N) {
    PointMap m; m.W = W; m.patches_x = W / 32; m.patched = 0;
    if (W % 32 == 0 && W >= 32) {
        const int64_t grid = N < (int64_t)H * W ? N : (int64_t)H * W;
        m.patched = grid / (kPatchH * (int64_t)W) * (kPatchH * (int64_t)W);   // whole kPatchH-row bands
    }
    return m;
}
__device__ __forceinline__ int64_t point_of(const PointMap &m, int64_t block, int q) {
    const int64_t lin = block * kPPB + q;
    if (lin >= m.patched) return lin;                                         // (patched is a multiple of kPatchH W = patches_x * kPPB)
    const int64_t band = block / m.patches_x; const int px = (int)(block - band * m.patches_x);
    return (band * kPatchH + (q >> 5)) * m.W + px * 32 + (q & 31);
}

constexpr int kTotalStride = 32;       // ints between two tiles' global counters: one 128-B line each (same-line atomics serialise)

template <bool SHIFT>
__global__ __launch_bounds__(kBlock) void k_tile_bin(const float *__restrict__ pts, int64_t N, ProjConst pc, Shift s, TileGeom g, int cap,
                                                      PointMap pm, int *__restrict__ tile_total, Entry *__restrict__ entries,
                                                      Entry *__restrict__ spill, int *__restrict__ spill_tile, int *__restrict__ spill_count,
                                                      int *__restrict__ hole_total) {
    extern __shared__ int lds_bin[];                  // hist[nt] | base[nt]
    if (blockIdx.x == 0 && threadIdx.x == 0) *hole_total = 0;   // the previous frame's k_tile_holes is done (stream order); k_tile_render re-fills it
    int *hist = lds_bin, *base = lds_bin + g.nt;
    for (int t = threadIdx.x; t < g.nt; t += kBlock) hist[t] = 0;
    __syncthreads();
    PointBins b[kPPT];
    int64_t pidx[kPPT];
    int rank[kPPT][4];
#pragma unroll
    for (int i = 0; i < kPPT; ++i) {
        pidx[i] = point_of(pm, blockIdx.x, i * kBlock + threadIdx.x);
        b[i] = bins_of_point<SHIFT>(pts, N, pidx[i], pc, s, g);
    }
#pragma unroll
    for (int i = 0; i < kPPT; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int ty = b[i].tyl + (j >> 1), tx = b[i].txl + (j & 1);
            // (a ballot-aggregated rank -- one LDS atomic per distinct tile per wave -- measured 3 us SLOWER than the plain atomic)
            rank[i][j] = (b[i].ok && ty <= b[i].tyh && tx <= b[i].txh) ? atomicAdd(&hist[ty * g.ntx + tx], 1) : -1;
        }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < kPPT; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (rank[i][j] == 0) {
                const int t = (b[i].tyl + (j >> 1)) * g.ntx + b[i].txl + (j & 1);
                base[t] = atomicAdd(tile_total + (int64_t)t * kTotalStride, hist[t]);
            }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < kPPT; ++i) {
        Entry e; e.fx = b[i].fx; e.fy = b[i].fy; e.err = b[i].err; e.idx = (int)pidx[i];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (rank[i][j] < 0) continue;
            const int t = (b[i].tyl + (j >> 1)) * g.ntx + b[i].txl + (j & 1);
            const int slot = base[t] + rank[i][j];
            if (slot < cap) entries[(int64_t)t * cap + slot] = e;
            else { const int q = atomicAdd(spill_count, 1); spill[q] = e; spill_tile[q] = t; }
        }
    }
}

struct FrameOut {
    uint8_t *frame;          // [H,W,3]
    unsigned *vbits;         // [H][ntx]   bit x%32 of word (y, x/32) = masked depth > 0 (fill_disocclusion's validity, common.py:160)
    unsigned short *cbits;   // [W][cpitch] the same map transposed, 16 rows per half-word: read as 32-bit words (x, y/32)
    float *mdepth;           // [P]   render[3] * (existing > 0)   (kenburns_effect.py:1039)
    float *render;           // [4,P] or null
    unsigned *holes;         // flat list of all holes of the frame, (y << 16) | x, in tile-completion order (capacity H * W)
    int *hole_count;         // [1] running total: every tile with holes reserves its run with one atomic; re-armed by the next frame
    int *totals;             // [nt] bin counters of k_tile_bin (zeroed here for the next frame)
    int cpitch;              // half-words per column of cbits (nty rounded up to even)
    const Entry *spill;      // entries that did not fit their tile's segment, with their tile ids; *spill_count of them
    const int *spill_tile;
    int *spill_count;        // re-armed by k_tile_holes (the kernel after the last reader)
};

__device__ __forceinline__ void lds_min_f32(float *addr, float v) {        // warp_device.h::atomic_min_f32 on LDS
    if (v >= 0.0f) atomicMin(reinterpret_cast<int *>(addr), __float_as_int(v));
    else atomicMax(reinterpret_cast<unsigned int *>(addr), __float_as_uint(v));
}

// Accumulators are 64-bit FIXED POINT in LDS.  Measured on gfx950 (phase ablation of this kernel at 1024^2): splatting the five
// channels with ds_add_f32 costs 88 us per frame (~0.2 lane-atomics per clock and CU), with ds_add_u64 19 us, and a
// sort-by-pixel + pull variant without float atomics 37 us (instruction bound).  Every product v * w is formed in fp32 exactly as
// the reference forms it (VALUE(data) * fltNorthwest, models/utils.py:270-310), converted to a 64-bit integer (2^-40 units for
// colour and weight, 2^-20 for depth: |v w| < 2^22 resp. 2^42) and added with an integer atomic: the sum is exact, order
// free -- deterministic, unlike the reference's fp32 atomicAdd -- and rounded to fp32 once at the end, i.e. within one ulp of the
// exact sum every fp32 summation order approximates.  A positive product never converts to 0, so `existing > 0` and
// `depth mask > 0` (the decisions fill_disocclusion depends on) are exactly the reference's.
constexpr float kScaleC = 1099511627776.0f;           // 2^40
constexpr float kScaleD = 1048576.0f;                 // 2^20
__device__ __forceinline__ unsigned long long to_fixed(float prod, float scale) {
    long long q = (long long)(prod * scale);
    if (q == 0) q = prod > 0.0f ? 1 : (prod < 0.0f ? -1 : 0);
    return (unsigned long long)q;
}
// the operator path accumulates ARBITRARY feature channels (Inpaint.forward's 64 context channels) at 2^-32 units: a product beyond
// +-2^31 would overflow the conversion (undefined behaviour) -- saturate instead (+-2^62 units, i.e. |v w| <= 2^30 is exact to 2^-32;
// NaN contributions count as the negative bound).  Supported range stated in INTEGRATION.md; ops.render_pointcloud(path='atomics')
// keeps IEEE semantics for non-finite inputs.
__device__ __forceinline__ unsigned long long to_fixed_sat(float prod, float scale) {
    const float f = fminf(fmaxf(prod * scale, -4.611686018427387904e18f), 4.611686018427387904e18f);
    long long q = (long long)f;
    if (q == 0) q = prod > 0.0f ? 1 : (prod < 0.0f ? -1 : 0);
    return (unsigned long long)q;
}
__device__ __forceinline__ float from_fixed(unsigned long long q, double inv_scale) { return (float)((double)(long long)q * inv_scale); }

__global__ __launch_bounds__(kBlock) void k_tile_render(const Entry *__restrict__ entries, int cap,
                                                         const float *__restrict__ rgb, const float *__restrict__ depth, int64_t N,
                                                         int H, int W, TileGeom g, FrameOut out) {
    __shared__ float zee[ZH * ZW];
    __shared__ float zd[TPIX];
    __shared__ unsigned long long acc[5 * TPIX];
    __shared__ unsigned rowbits[TH];
    __shared__ int nholes, hole_base;
    __shared__ unsigned short hole_px[TPIX];
    const int t = blockIdx.x, tid = threadIdx.x;
    const int tx0 = (t % g.ntx) * TW, ty0 = (t / g.ntx) * TH;
    for (int i = tid; i < ZH * ZW; i += kBlock) zee[i] = 1000000.0f;        // models/utils.py:59
    for (int i = tid; i < 5 * TPIX; i += kBlock) acc[i] = 0ull;
    if (tid == 0) nholes = 0;
    __syncthreads();
    const int total = out.totals[(int64_t)t * kTotalStride];
    const int e0 = (int)0, e1 = total < cap ? total : cap;
    entries += (int64_t)t * cap;
    const int nspill = total > cap ? *out.spill_count : 0;                     // only an overflowing tile has entries on the spill list
    // ---- updateZee (models/utils.py:101-147) on the window ------------------------------------------------------------------
    auto zee_entry = [&](const Entry &en) {
        int x0, y0, cx, cy; float w[4];
        corner_weights(en.fx, en.fy, x0, y0, w);
        if (!argmax_corner(w, x0, y0, cx, cy)) return;
        if (cx < 0 || cx >= W || cy < 0 || cy >= H) return;
        const int lx = cx - (tx0 - 1), ly = cy - (ty0 - 1);
        if (lx < 0 || lx >= ZW || ly < 0 || ly >= ZH) return;
        lds_min_f32(&zee[ly * ZW + lx], en.err);
    };
    // The first kReg * kBlock entries of the tile (all of them unless the tile is crowded) stay in registers for both passes, and
    // their colours are requested NOW: the gather's round trip runs under the z pass, the degrid and two barriers instead of in
    // front of the splat (all blocks of a CU reach the same phase together, so exposed latency is not hidden by other blocks).
    constexpr int kReg = 4;
    Entry en[kReg]; bool has[kReg]; float c0[kReg], c1[kReg], c2[kReg], c3[kReg];
#pragma unroll
    for (int k = 0; k < kReg; ++k) {
        const int e = e0 + tid + k * kBlock;
        has[k] = e < e1;
        en[k] = has[k] ? entries[e] : Entry{0.0f, 0.0f, 0.0f, 0};
    }
#pragma unroll
    for (int k = 0; k < kReg; ++k) {
        c0[k] = c1[k] = c2[k] = c3[k] = 0.0f;
        if (has[k]) { const int64_t p = en[k].idx; c0[k] = rgb[p]; c1[k] = rgb[N + p]; c2[k] = rgb[2 * N + p]; c3[k] = depth[p]; }
    }
#pragma unroll
    for (int k = 0; k < kReg; ++k) if (has[k]) zee_entry(en[k]);
    for (int e = e0 + tid + kReg * kBlock; e < e1; e += kBlock) zee_entry(entries[e]);
    for (int e = tid; e < nspill; e += kBlock) if (out.spill_tile[e] == t) zee_entry(out.spill[e]);
    __syncthreads();
    // ---- updateDegrid (models/utils.py:152-212), Jacobi form ------------------------------------------------------------------
    for (int i = tid; i < TPIX; i += kBlock) {
        const int lx = i % TW, ly = i / TW;
        const int x = tx0 + lx, y = ty0 + ly;
        const float c = zee[(ly + 1) * ZW + lx + 1];
        float r = c;
        if (x < W && y < H) {
            int cnt = 0; float sum = 0.0f;
            const int ox[4] = {1, 0, 1, 1}, oy[4] = {0, 1, 1, -1};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int x1 = x + ox[k], y1 = y + oy[k], x2 = x - ox[k], y2 = y - oy[k];
                if (x1 < 0 || x1 >= W || y1 < 0 || y1 >= H) continue;
                if (x2 < 0 || x2 >= W || y2 < 0 || y2 >= H) continue;
                const float a = zee[(ly + 1 + oy[k]) * ZW + lx + 1 + ox[k]], d = zee[(ly + 1 - oy[k]) * ZW + lx + 1 - ox[k]];
                if ((double)c >= (double)a + 1.0 && (double)c >= (double)d + 1.0) { cnt += 2; sum += a; sum += d; }
            }
            if (cnt > 0) r = fminf(c, sum / (float)cnt);
        }
        zd[i] = r;
    }
    __syncthreads();
    // ---- updateOutput (models/utils.py:215-313): z-test + bilinear splat, C = rgb + depth, + the ones channel -------------------
    auto splat = [&](const Entry &e, bool gather, float v0, float v1, float v2, float v3) {
        int x0, y0; float w[4];
        corner_weights(e.fx, e.fy, x0, y0, w);
        int li[4]; bool pass[4]; bool any = false;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int cx = x0 + (k & 1), cy = y0 + (k >> 1);
            const int lx = cx - tx0, ly = cy - ty0;
            const bool in = lx >= 0 && lx < TW && ly >= 0 && ly < TH && cx < W && cy < H;      // cx, cy >= 0 follows from lx, ly >= 0
            li[k] = in ? ly * TW + lx : 0;
            pass[k] = in && ((double)e.err <= (double)zd[li[k]] + 1.0);
            any = any || pass[k];
        }
        if (!any) return;
        if (gather) { const int64_t p = e.idx; v0 = rgb[p]; v1 = rgb[N + p]; v2 = rgb[2 * N + p]; v3 = depth[p]; }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (!pass[k]) continue;
            const float wk = w[k];
            atomicAdd(&acc[li[k]], to_fixed(v0 * wk, kScaleC));
            atomicAdd(&acc[TPIX + li[k]], to_fixed(v1 * wk, kScaleC));
            atomicAdd(&acc[2 * TPIX + li[k]], to_fixed(v2 * wk, kScaleC));
            atomicAdd(&acc[3 * TPIX + li[k]], to_fixed(v3 * wk, kScaleD));
            atomicAdd(&acc[4 * TPIX + li[k]], to_fixed(1.0f * wk, kScaleC));
        }
    };
#pragma unroll
    for (int k = 0; k < kReg; ++k) if (has[k]) splat(en[k], false, c0[k], c1[k], c2[k], c3[k]);
    for (int e = e0 + tid + kReg * kBlock; e < e1; e += kBlock) splat(entries[e], true, 0.0f, 0.0f, 0.0f, 0.0f);
    for (int e = tid; e < nspill; e += kBlock) if (out.spill_tile[e] == t) splat(out.spill[e], true, 0.0f, 0.0f, 0.0f, 0.0f);
    __syncthreads();
    // ---- models/utils.py:315 normalise; kenburns_effect.py:1039-1040 depth mask + uint8; hole list for fill_disocclusion --------
    const int64_t plane = (int64_t)H * W;
    for (int i = tid; i < TPIX; i += kBlock) {                                // a wave = two 32-px rows of the tile
        const int lx = i % TW, ly = i / TW;
        const int x = tx0 + lx, y = ty0 + ly;
        const bool inside = x < W && y < H;
        const float e = from_fixed(acc[4 * TPIX + i], 1.0 / 1099511627776.0);
        const float den = e + 0.0000001f;
        const float r0 = from_fixed(acc[i], 1.0 / 1099511627776.0) / den, r1 = from_fixed(acc[TPIX + i], 1.0 / 1099511627776.0) / den;
        const float r2 = from_fixed(acc[2 * TPIX + i], 1.0 / 1099511627776.0) / den, r3 = from_fixed(acc[3 * TPIX + i], 1.0 / 1048576.0) / den;
        const float m = r3 * (e > 0.0f ? 1.0f : 0.0f);
        const bool okp = inside && (double)m > 0.0;                             // common.py:160
        const unsigned long long bits = __ballot(okp);
        if ((tid & 31) == 0) {
            rowbits[ly] = (unsigned)(bits >> (tid & 32));
            if (y < H) out.vbits[(int64_t)y * g.ntx + (t % g.ntx)] = (unsigned)(bits >> (tid & 32));
        }
        if (!inside) continue;
        const int64_t o = (int64_t)y * W + x;
        out.mdepth[o] = m;
        if (out.render) { out.render[o] = r0; out.render[plane + o] = r1; out.render[2 * plane + o] = r2; out.render[3 * plane + o] = r3; }
        out.frame[o * 3 + 0] = to_u8(r0); out.frame[o * 3 + 1] = to_u8(r1); out.frame[o * 3 + 2] = to_u8(r2);
        if (!okp) hole_px[atomicAdd(&nholes, 1)] = (unsigned short)i;
    }
    __syncthreads();
    // the tile's holes join the frame's flat list: one global atomic per tile that has holes (a few hundred per frame, spread over
    // the kernel) instead of per-tile lists + a 2048-entry prefix scan in every block of k_tile_holes (5.8 of its 16.9 us)
    if (tid == 0) {
        hole_base = nholes > 0 ? atomicAdd(out.hole_count, nholes) : 0;
        out.totals[(int64_t)t * kTotalStride] = 0;                              // the bin counter is re-armed for the next frame
    }
    __syncthreads();
    for (int i = tid; i < nholes; i += kBlock) {
        const int li = hole_px[i];
        out.holes[hole_base + i] = ((unsigned)(ty0 + li / TW) << 16) | (unsigned)(tx0 + li % TW);
    }
    if (tid < TW && tx0 + tid < W) {                                            // 16 x 32 bit transpose: a half-word per column
        unsigned c = 0;
#pragma unroll
        for (int r = 0; r < TH; ++r) c |= ((rowbits[r] >> tid) & 1u) << r;
        out.cbits[(int64_t)(tx0 + tid) * out.cpitch + (t / g.ntx)] = (unsigned short)c;
        if ((g.nty & 1) && t / g.ntx == g.nty - 1) out.cbits[(int64_t)(tx0 + tid) * out.cpitch + g.nty] = 0;   // pad to whole words
    }
}

// ---- render_pointcloud for ANY channel count on the tile path (Inpaint.forward splats 68 feature channels, pointcloud_inpainting.py:135) ----
// Same binning, same z-buffer / degrid / z-test as k_tile_render; the channels are splatted in GROUPS of kGroup through 64-bit fixed-point
// LDS accumulators (2^-32 units: |value x weight| < 2^31; the weight plane keeps 2^-40 so that a positive weight never becomes 0 and
// `existing > 0` is the reference's decision), each group normalised by the weight plane and written as whole row segments.  The
// per-entry geometry (corner pixels, weights, z-test) of the first kReg x 256 entries of a tile is computed once and kept in registers
// for all groups.  Integer sums: deterministic (the L2 float-atomic path k_update_output<.., 0> is not) and ~4x faster at C = 68
// (276 L2 atomics per point there).
constexpr int kGroup = 8;
constexpr float kScaleF = 4294967296.0f;             // 2^32
struct EntryGeom { unsigned short li[4]; float w[4]; unsigned mask; };       // corner pixel in the tile, weight, z-test bits

__global__ __launch_bounds__(kBlock) void k_tile_render_c(const Entry *__restrict__ entries, int cap, const float *__restrict__ data, int C,
                                                           int64_t N, int H, int W, TileGeom g, int *__restrict__ totals,
                                                           const Entry *__restrict__ spill, const int *__restrict__ spill_tile,
                                                           const int *__restrict__ spill_count, float *__restrict__ render,
                                                           float *__restrict__ existing) {
    __shared__ float zee[ZH * ZW];
    __shared__ float zd[TPIX];
    __shared__ float den[TPIX];
    __shared__ unsigned long long wacc[TPIX];
    __shared__ unsigned long long acc[kGroup * TPIX];
    const int t = blockIdx.x, tid = threadIdx.x;
    const int tx0 = (t % g.ntx) * TW, ty0 = (t / g.ntx) * TH;
    for (int i = tid; i < ZH * ZW; i += kBlock) zee[i] = 1000000.0f;        // models/utils.py:59
    for (int i = tid; i < TPIX; i += kBlock) wacc[i] = 0ull;
    __syncthreads();
    const int total = totals[(int64_t)t * kTotalStride];
    const int e1 = total < cap ? total : cap;
    entries += (int64_t)t * cap;
    const int nspill = total > cap ? *spill_count : 0;
    auto zee_entry = [&](const Entry &en) {                                 // updateZee, models/utils.py:101-147
        int x0, y0, cx, cy; float w[4];
        corner_weights(en.fx, en.fy, x0, y0, w);
        if (!argmax_corner(w, x0, y0, cx, cy)) return;
        if (cx < 0 || cx >= W || cy < 0 || cy >= H) return;
        const int lx = cx - (tx0 - 1), ly = cy - (ty0 - 1);
        if (lx < 0 || lx >= ZW || ly < 0 || ly >= ZH) return;
        lds_min_f32(&zee[ly * ZW + lx], en.err);
    };
    constexpr int kReg = 4;
    Entry en[kReg]; bool has[kReg];
#pragma unroll
    for (int k = 0; k < kReg; ++k) {
        const int e = tid + k * kBlock;
        has[k] = e < e1;
        en[k] = has[k] ? entries[e] : Entry{0.0f, 0.0f, 0.0f, 0};
    }
#pragma unroll
    for (int k = 0; k < kReg; ++k) if (has[k]) zee_entry(en[k]);
    for (int e = tid + kReg * kBlock; e < e1; e += kBlock) zee_entry(entries[e]);
    for (int e = tid; e < nspill; e += kBlock) if (spill_tile[e] == t) zee_entry(spill[e]);
    __syncthreads();
    for (int i = tid; i < TPIX; i += kBlock) {                              // updateDegrid (models/utils.py:152-212), Jacobi form
        const int lx = i % TW, ly = i / TW;
        const int x = tx0 + lx, y = ty0 + ly;
        const float c = zee[(ly + 1) * ZW + lx + 1];
        float r = c;
        if (x < W && y < H) {
            int cnt = 0; float sum = 0.0f;
            const int ox[4] = {1, 0, 1, 1}, oy[4] = {0, 1, 1, -1};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int x1 = x + ox[k], y1 = y + oy[k], x2 = x - ox[k], y2 = y - oy[k];
                if (x1 < 0 || x1 >= W || y1 < 0 || y1 >= H) continue;
                if (x2 < 0 || x2 >= W || y2 < 0 || y2 >= H) continue;
                const float a = zee[(ly + 1 + oy[k]) * ZW + lx + 1 + ox[k]], d = zee[(ly + 1 - oy[k]) * ZW + lx + 1 - ox[k]];
                if ((double)c >= (double)a + 1.0 && (double)c >= (double)d + 1.0) { cnt += 2; sum += a; sum += d; }
            }
            if (cnt > 0) r = fminf(c, sum / (float)cnt);
        }
        zd[i] = r;
    }
    __syncthreads();
    auto geom_of = [&](const Entry &e) {                                    // updateOutput's z-test + bilinear weights (models/utils.py:215-313)
        EntryGeom q; int x0, y0;
        corner_weights(e.fx, e.fy, x0, y0, q.w);
        q.mask = 0u;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int cx = x0 + (k & 1), cy = y0 + (k >> 1);
            const int lx = cx - tx0, ly = cy - ty0;
            const bool in = lx >= 0 && lx < TW && ly >= 0 && ly < TH && cx < W && cy < H;
            q.li[k] = (unsigned short)(in ? ly * TW + lx : 0);
            if (in && ((double)e.err <= (double)zd[q.li[k]] + 1.0)) q.mask |= 1u << k;
        }
        return q;
    };
    EntryGeom gq[kReg];
#pragma unroll
    for (int k = 0; k < kReg; ++k) { gq[k] = geom_of(en[k]); if (!has[k]) gq[k].mask = 0u; }
    // the ones channel (tenOutput[:, -1]): weights only
    auto splat_w = [&](const EntryGeom &q) {
#pragma unroll
        for (int k = 0; k < 4; ++k) if (q.mask & (1u << k)) atomicAdd(&wacc[q.li[k]], to_fixed(1.0f * q.w[k], kScaleC));
    };
#pragma unroll
    for (int k = 0; k < kReg; ++k) splat_w(gq[k]);
    for (int e = tid + kReg * kBlock; e < e1; e += kBlock) splat_w(geom_of(entries[e]));
    for (int e = tid; e < nspill; e += kBlock) if (spill_tile[e] == t) splat_w(geom_of(spill[e]));
    __syncthreads();
    const int64_t plane = (int64_t)H * W;
    for (int i = tid; i < TPIX; i += kBlock) {
        const float e = from_fixed(wacc[i], 1.0 / 1099511627776.0);
        den[i] = e + 0.0000001f;                                            // models/utils.py:315
        const int x = tx0 + i % TW, y = ty0 + i / TW;
        if (x < W && y < H) existing[(int64_t)y * W + x] = e;
    }
    for (int g0 = 0; g0 < C; g0 += kGroup) {
        const int ng = C - g0 < kGroup ? C - g0 : kGroup;
        for (int i = tid; i < kGroup * TPIX; i += kBlock) acc[i] = 0ull;
        __syncthreads();                                                    // (also orders den[] before its first use)
        const float *dg = data + (int64_t)g0 * N;
        auto splat_g = [&](const EntryGeom &q, int idx) {
            if (!q.mask) return;
            float v[kGroup];
#pragma unroll
            for (int j = 0; j < kGroup; ++j) v[j] = j < ng ? dg[(int64_t)j * N + idx] : 0.0f;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (!(q.mask & (1u << k))) continue;
                const float wk = q.w[k];
#pragma unroll
                for (int j = 0; j < kGroup; ++j) if (j < ng) atomicAdd(&acc[j * TPIX + q.li[k]], to_fixed_sat(v[j] * wk, kScaleF));
            }
        };
#pragma unroll
        for (int k = 0; k < kReg; ++k) splat_g(gq[k], en[k].idx);
        for (int e = tid + kReg * kBlock; e < e1; e += kBlock) { const Entry x = entries[e]; splat_g(geom_of(x), x.idx); }
        for (int e = tid; e < nspill; e += kBlock) if (spill_tile[e] == t) { const Entry x = spill[e]; splat_g(geom_of(x), x.idx); }
        __syncthreads();
        for (int i = tid; i < ng * TPIX; i += kBlock) {                     // a wave = two 32-px rows of one channel
            const int j = i / TPIX, px = i - j * TPIX;
            const int x = tx0 + px % TW, y = ty0 + px / TW;
            if (x < W && y < H) render[(int64_t)(g0 + j) * plane + (int64_t)y * W + x] = from_fixed(acc[i], 1.0 / 4294967296.0) / den[px];
        }
        __syncthreads();
    }
    if (tid == 0) totals[(int64_t)t * kTotalStride] = 0;                    // the bin counter is re-armed for the next call
}

__constant__ float kDirX[16] = {-1, 0, 1, 1, -1, 1, 2, 2, -2, -1, 1, 2, 3, 3, 3, 3};   // common.py:168
__constant__ float kDirY[16] = {1, 1, 1, 0, 2, 2, 1, -1, 3, 3, 3, 3, 2, 1, -1, -2};    // common.py:169

// fill_disocclusion (common.py:145-248).  Holes come from the frame's flat list (k_tile_render appends every tile's holes with one
// atomic per tile): all groups of 32 lanes carry the same load wherever the holes are.  32 lanes per hole = 16 directions x {from, to}; every lane marches
// ONE ray through the valid BITMAP (128 KB at 1024^2: cache resident) with 4 speculative steps per round trip; the two axis
// directions, whose rays run the length of a disoccluded border strip, scan the row / column bitmap a word (32 px) at a time --
// their steps are exact integers in fp32, so the visited pixels are the same.  A 16-lane lexicographic (distance, direction)
// minimum then picks the direction exactly like the reference's sequential loop (shortest distance, first direction wins ties).
__global__ __launch_bounds__(kBlock) void k_tile_holes(int H, int W, TileGeom g, FrameOut out) {
    const int tid = threadIdx.x;
    const int total = *out.hole_count;                                          // the frame's flat hole list (k_tile_render)
    if (blockIdx.x == 0 && tid == 0) *out.spill_count = 0;                      // every reader (k_tile_render) has finished
    const int lane32 = tid & 31, k = lane32 & 15;
    const bool to = lane32 >= 16;
    float dx = kDirX[k], dy = kDirY[k];
    const float nrm = sqrtf((dx * dx) + (dy * dy));                             // common.py:172-175
    dx /= nrm; dy /= nrm;
    const float sx = to ? dx : -dx, sy = to ? dy : -dy;                         // a - d == a + (-d) exactly
    const int64_t plane = (int64_t)H * W;
    const int groups = gridDim.x * (kBlock >> 5);
    for (int h = blockIdx.x * (kBlock >> 5) + (tid >> 5); h < total; h += groups) {
        const unsigned packed = out.holes[h];
        const int x = (int)(packed & 0xFFFFu), y = (int)(packed >> 16);
        float fx = (float)x, fy = (float)y;
        int ix = 0, iy = 0;
        bool ok = false, oblique = false;
        if (k != 1 && k != 3) { oblique = true; }
        else if (k == 3) {
            // direction (1, 0): the ray visits (x +- j, y) exactly (integer steps are exact in fp32) -> scan the row bitmap by words
            const unsigned *R = out.vbits + (int64_t)y * g.ntx;
            if (to) {
                int q = (x + 1) >> 5;
                unsigned bits = (x + 1 < W) ? (R[q] & (0xFFFFFFFFu << ((x + 1) & 31))) : 0u;
                while (bits == 0u && ++q < g.ntx) bits = R[q];
                ok = bits != 0u; ix = ok ? q * 32 + __ffs((int)bits) - 1 : W; iy = y;
            } else {
                int q = (x - 1) >> 5;
                unsigned bits = (x - 1 >= 0) ? (R[q] & (0xFFFFFFFFu >> (31 - ((x - 1) & 31)))) : 0u;
                while (bits == 0u && --q >= 0) bits = R[q];
                ok = bits != 0u; ix = ok ? q * 32 + 31 - __clz((int)bits) : -1; iy = y;
            }
        } else if (k == 1) {
            // direction (0, 1): the same on the transposed bitmap
            const unsigned *C = reinterpret_cast<const unsigned *>(out.cbits + (int64_t)x * out.cpitch);
            const int nq = out.cpitch >> 1;                                      // whole 32-row words per column
            if (to) {
                int q = (y + 1) >> 5;
                unsigned bits = (y + 1 < H) ? (C[q] & (0xFFFFFFFFu << ((y + 1) & 31))) : 0u;
                while (bits == 0u && ++q < nq) bits = C[q];
                ok = bits != 0u; iy = ok ? q * 32 + __ffs((int)bits) - 1 : H; ix = x;
            } else {
                int q = (y - 1) >> 5;
                unsigned bits = (y - 1 >= 0) ? (C[q] & (0xFFFFFFFFu >> (31 - ((y - 1) & 31)))) : 0u;
                while (bits == 0u && --q >= 0) bits = C[q];
                ok = bits != 0u; iy = ok ? q * 32 + 31 - __clz((int)bits) : -1; ix = x;
            }
        }
        // Oblique rays (common.py:186-193 / :197-204) march in rounds of kAhead speculative steps.  Between rounds the 16 directions
        // of the hole share the shortest COMPLETE from-to distance found so far (the axis rays are complete before the first
        // round): a ray that has already walked further than that cannot belong to the winning direction -- its endpoints are
        // within 0.71 px of the unrounded positions and at least steps + 2 unit steps apart -- and gives up.  The winner
        // (strict lexicographic (dist, k) minimum below) is unchanged; rays along a disocclusion band no longer run to the border.
        bool done = !oblique;
        float best_so_far = INFINITY;
        for (int steps = 0;; steps += 4) {
            constexpr int kAhead = 4;            // speculative steps per trip: the positions do not depend on what is read
            {
                const int ox = __shfl_xor(ix, 16), oy = __shfl_xor(iy, 16);
                const bool pair_ok = done && ok && (__shfl_xor((int)(done && ok), 16) != 0);
                const float ddx = (float)(ox - ix), ddy = (float)(oy - iy);
                float m = pair_ok ? sqrtf(ddx * ddx + ddy * ddy) : INFINITY;
#pragma unroll
                for (int off = 8; off >= 1; off >>= 1) m = fminf(m, __shfl_xor(m, off));
                best_so_far = fminf(best_so_far, m);
            }
            if (!done && (float)steps >= best_so_far) { done = true; ok = false; }   // dist >= steps + 2 - 1.42 > best
            const unsigned long long waiting = __ballot(!done);
            if (((waiting >> (tid & 32)) & 0xFFFFFFFFull) == 0ull) break;
            if (!done) {
                int jx[kAhead], jy[kAhead], v[kAhead];
#pragma unroll
                for (int j = 0; j < kAhead; ++j) { fx += sx; fy += sy; jx[j] = (int)roundf(fx); jy[j] = (int)roundf(fy); }
#pragma unroll
                for (int j = 0; j < kAhead; ++j) {
                    const bool inb = jx[j] >= 0 && jx[j] < W && jy[j] >= 0 && jy[j] < H;
                    const unsigned word = inb ? out.vbits[(int64_t)jy[j] * g.ntx + (jx[j] >> 5)] : 0u;
                    v[j] = inb ? (int)((word >> (jx[j] & 31)) & 1u) : 2;        // 2 = left the image: the ray stops, nothing found
                }
                int stop = -1;
#pragma unroll
                for (int j = kAhead - 1; j >= 0; --j) if (v[j] != 0) stop = j;
                if (stop >= 0) {
#pragma unroll
                    for (int j = 0; j < kAhead; ++j) if (j == stop) { ix = jx[j]; iy = jy[j]; ok = v[j] == 1; }
                    done = true;
                }
            }
        }
        // pair the from/to rays of one direction (lanes k and k+16)
        const int ox = __shfl_xor(ix, 16), oy = __shfl_xor(iy, 16);
        const bool ook = __shfl_xor((int)ok, 16) != 0;
        const int fromx = to ? ox : ix, fromy = to ? oy : iy, tox = to ? ix : ox, toy = to ? iy : oy;
        const float ddx = (float)(tox - fromx), ddy = (float)(toy - fromy);
        const float dist = sqrtf(ddx * ddx + ddy * ddy);                        // common.py:208
        const bool cand = ok && ook && (1000000.0f > dist);                     // fltShortest starts at 1e6, strict >
        float best = cand ? dist : INFINITY;
        int bestk = cand ? k : 16;
#pragma unroll
        for (int off = 8; off >= 1; off >>= 1) {                                // 16-lane lexicographic (dist, k) min
            const float od = __shfl_xor(best, off); const int okk = __shfl_xor(bestk, off);
            if (od < best || (od == best && okk < bestk)) { best = od; bestk = okk; }
        }
        if (bestk < 16) {
            const int srcl = (tid & 32) + bestk;                                // the "from" lane of the winning direction
            const int wfx = __shfl(fromx, srcl), wfy = __shfl(fromy, srcl), wtx = __shfl(tox, srcl), wty = __shfl(toy, srcl);
            const int64_t of = (int64_t)wfy * W + wfx, ot = (int64_t)wty * W + wtx;
            const float dfrom = out.mdepth[of], dto = out.mdepth[ot];
            const int64_t so = dfrom < dto ? ot : of;                           // common.py:214-217
            const int64_t o = (int64_t)y * W + x;
            if (lane32 < 3) out.frame[o * 3 + lane32] = out.frame[so * 3 + lane32];   // uint8 of the same render value
            if (out.render && lane32 < 4) out.render[(int64_t)lane32 * plane + o] = out.render[(int64_t)lane32 * plane + so];
        }
    }
}

// scratch layout (bytes, 16-B aligned sections):
//   header ints : totals[nt * kTotalStride] (one 128-B line per tile; zero between frames) | hole total | spill_count (zero between frames)
//   vbits[H * ntx] (u32) | cbits[W * cpitch] (u16) | mdepth[P] (f32) | holes[P] (u32, flat list) | entries[nt * cap] (16 B) |
//   spill entries[4 N] (16 B) | spill tiles[4 N] (int)
struct TileScratch { int *totals, *hole_count, *spill_count; unsigned *vbits; unsigned short *cbits; float *mdepth; unsigned *holes;
                     Entry *entries, *spill; int *spill_tile; int cpitch, cap; };
inline size_t align16(size_t v) { return (v + 15) & ~(size_t)15; }
inline size_t header_bytes(int nt) { return align16(sizeof(int) * ((size_t)nt * kTotalStride + 64)); }
inline size_t bin_blocks(int64_t N) { return (size_t)((N + kPPB - 1) / kPPB); }
// segment capacity per tile: 2.5x the load of a uniform cloud (N / P points per pixel, x 1.21 for the shared 1-px borders), at
// least 1024, a multiple of 64.  Monotone in N, so a scratch sized for a larger N serves every smaller one.
inline int tile_cap(int H, int W, int64_t N) {
    const double per_tile = (double)N * TPIX / ((double)H * W) * 1.21;
    int64_t c = (int64_t)(per_tile * 2.5) + 63;
    c -= c % 64;
    return (int)(c < 1024 ? 1024 : (c > (1 << 24) ? (1 << 24) : c));
}
struct Sizes { size_t vbits, cbits, mdepth, holes, entries, spill, spill_tile; int cpitch, cap; };
inline Sizes section_sizes(int H, int W, int64_t N) {
    const TileGeom g = tile_geom(H, W);
    Sizes z; z.cpitch = g.nty + (g.nty & 1);          // half-words per column, padded to whole 32-row words
    z.cap = tile_cap(H, W, N);
    z.vbits = align16(4 * (size_t)H * g.ntx);
    z.cbits = align16(2 * (size_t)W * z.cpitch);
    z.mdepth = align16(4 * (size_t)H * W);
    z.holes = align16(4 * (size_t)H * W);
    z.entries = align16(16 * (size_t)g.nt * z.cap);
    z.spill = align16(16 * 4 * (size_t)N + 16);
    z.spill_tile = align16(4 * 4 * (size_t)N + 16);
    return z;
}
inline TileScratch carve(void *scratch, int H, int W, int nt, int64_t N) {
    const Sizes z = section_sizes(H, W, N);
    TileScratch s; char *p = (char *)scratch;
    s.totals = (int *)p; s.hole_count = s.totals + (size_t)nt * kTotalStride; s.spill_count = s.hole_count + 32;   // own 128-B lines
    p += header_bytes(nt);
    s.vbits = (unsigned *)p; p += z.vbits;
    s.cbits = (unsigned short *)p; p += z.cbits;
    s.mdepth = (float *)p; p += z.mdepth;
    s.holes = (unsigned *)p; p += z.holes;
    s.entries = (Entry *)p; p += z.entries;
    s.spill = (Entry *)p; p += z.spill;
    s.spill_tile = (int *)p;
    s.cpitch = z.cpitch; s.cap = z.cap;
    return s;
}

}  // namespace

extern "C" size_t csm_warp_tile_scratch_bytes(int H, int W, int64_t N) {
    if (H <= 0 || W <= 0 || N < 0) return 0;
    const Sizes z = section_sizes(H, W, N);
    return header_bytes(tile_geom(H, W).nt) + z.vbits + z.cbits + z.mdepth + z.holes + z.entries + z.spill + z.spill_tile;
}

extern "C" size_t csm_warp_tile_header_bytes(int H, int W) { return (H <= 0 || W <= 0) ? 0 : header_bytes(tile_geom(H, W).nt); }

// largest tile count the block-local histograms / prefix tables support (LDS: 4 B per tile); larger frames use csm_warp_frame
extern "C" int csm_warp_tile_supported(int H, int W) { return H > 0 && W > 0 && tile_geom(H, W).nt <= 8192; }

// ---- K frames of one cloud in ONE call: frame k + 1's binning and frame k - 1's hole fill run under frame k's render ------------------
// The three kernels of a frame are a dependent chain (bin 12 us -> render 35 us -> holes 11 us at 1024^2) and each leaves most of the
// chip idle part of the time (binning is atomic-latency bound, the hole fill walks bitmaps): one frame in isolation runs at 0.34 of the
// HBM roof.  Frames of a video are independent, so the call deals them round-robin onto `lanes` internal streams, each with its own
// scratch, forked from and joined to the caller's stream with events: the caller sees one
// asynchronous call on one stream, the GPU sees up to three chains in different phases (0.5 of the roof, profiles/).
namespace {
struct LaneSet { hipStream_t aux[3] = {nullptr, nullptr, nullptr}; hipEvent_t fork = nullptr, join[3] = {nullptr, nullptr, nullptr}; bool ready = false; };
LaneSet g_lanes[32];
std::mutex g_lanes_mutex;
int get_lanes(LaneSet *&ls) {
    int d = 0;
    CSM_HIP(hipGetDevice(&d));
    std::lock_guard<std::mutex> lk(g_lanes_mutex);
    ls = &g_lanes[d & 31];
    if (!ls->ready) {
        for (int i = 0; i < 3; ++i) {
            CSM_HIP(hipStreamCreateWithFlags(&ls->aux[i], hipStreamNonBlocking));
            CSM_HIP(hipEventCreateWithFlags(&ls->join[i], hipEventDisableTiming));
        }
        CSM_HIP(hipEventCreateWithFlags(&ls->fork, hipEventDisableTiming));
        ls->ready = true;
    }
    return CSM_OK;
}
}  // namespace

extern "C" size_t csm_warp_frames_scratch_bytes(int H, int W, int64_t N, int lanes) {
    const size_t one = (csm_warp_tile_scratch_bytes(H, W, N) + 255) & ~(size_t)255;
    return one * (size_t)(lanes < 1 ? 1 : (lanes > 3 ? 3 : lanes));
}

extern "C" int csm_warp_frames_tiled(const float *pts, const float *rgb, const float *depth, int64_t N, int H, int W, double focal,
                                     double baseline, const float *shifts_host, int K, int lanes, void *scratch, float *render_filled,
                                     uint8_t *frames_u8, void *stream) {
    CSM_REQUIRE(K >= 0 && lanes >= 1 && lanes <= 3);
    if (K == 0) return CSM_OK;
    CSM_REQUIRE(shifts_host && scratch && frames_u8);
    CSM_REQUIRE((((uintptr_t)scratch) & 15) == 0);
    const size_t one = (csm_warp_tile_scratch_bytes(H, W, N) + 255) & ~(size_t)255;
    const size_t frame_bytes = (size_t)H * W * 3, render_floats = (size_t)4 * H * W;
    hipStream_t main = (hipStream_t)stream;
    const int nl = lanes < K ? lanes : K;
    LaneSet *ls = nullptr;
    std::unique_lock<std::mutex> lk(g_lanes_mutex, std::defer_lock);       // (released on every return path)
    if (nl > 1) {
        int rc = get_lanes(ls); if (rc) return rc;
        // the lane streams are shared by every caller on this device: one multi-frame call at a time forks from / joins to its own stream
        lk.lock();
        // every lane is an internal stream; the caller's stream only forks and joins (frames issued on the caller's stream itself -- the
        // NULL stream in a plain torch program -- overlapped worse: 47.6 us per frame against 39 for three side streams)
        CSM_HIP(hipEventRecord(ls->fork, main));
        for (int i = 0; i < nl; ++i) CSM_HIP(hipStreamWaitEvent(ls->aux[i], ls->fork, 0));
    }
    int rc = CSM_OK;
    for (int k = 0; k < K && rc == CSM_OK; ++k) {
        const int lane = k % nl;
        hipStream_t st = nl == 1 ? main : ls->aux[lane];
        rc = csm_warp_frame_tiled(pts, rgb, depth, N, H, W, focal, baseline, shifts_host[3 * k], shifts_host[3 * k + 1], shifts_host[3 * k + 2],
                                  (char *)scratch + (size_t)lane * one, render_filled ? render_filled + (size_t)k * render_floats : nullptr,
                                  frames_u8 + (size_t)k * frame_bytes, (void *)st);
    }
    if (nl > 1) {
        for (int i = 0; i < nl; ++i) {
            hipError_t e = hipEventRecord(ls->join[i], ls->aux[i]);
            if (e == hipSuccess) e = hipStreamWaitEvent(main, ls->join[i], 0);
            if (e != hipSuccess && rc == CSM_OK) { csm::set_error("csm_warp_frames_tiled join: %s", hipGetErrorString(e)); rc = CSM_ERR_HIP; }
        }
    }
    return rc;
}

extern "C" int csm_warp_frame_tiled(const float *pts, const float *rgb, const float *depth, int64_t N, int H, int W, double focal,
                                    double baseline, float sx, float sy, float sz, void *scratch, float *render_filled,
                                    uint8_t *frame_u8, void *stream) {
    CSM_REQUIRE(scratch && frame_u8 && N >= 0 && H > 0 && W > 0 && N < (1ll << 29));
    CSM_REQUIRE(N == 0 || (pts && rgb && depth));
    CSM_REQUIRE((((uintptr_t)scratch) & 15) == 0);
    if (!csm_warp_tile_supported(H, W)) return csm::fail_arg("frame too large for the tiled path (more than 8192 tiles): use csm_warp_frame");
    hipStream_t st = (hipStream_t)stream;
    const TileGeom g = tile_geom(H, W);
    const TileScratch ts = carve(scratch, H, W, g.nt, N);
    const ProjConst pc = make_proj(H, W, focal, baseline);
    const Shift s{sx, sy, sz};
    const unsigned nb = (unsigned)bin_blocks(N);
    int rc;
    if (N > 0) {
        k_tile_bin<true><<<nb, kBlock, 2 * sizeof(int) * (size_t)g.nt, st>>>(pts, N, pc, s, g, ts.cap, make_point_map(H, W, N), ts.totals,
                                                                              ts.entries, ts.spill, ts.spill_tile, ts.spill_count, ts.hole_count);
        rc = csm::check_launch("k_tile_bin"); if (rc) return rc;
    } else {
        CSM_HIP(hipMemsetAsync(ts.hole_count, 0, sizeof(int), st));
    }
    FrameOut out{frame_u8, ts.vbits, ts.cbits, ts.mdepth, render_filled, ts.holes, ts.hole_count, ts.totals, ts.cpitch,
                 ts.spill, ts.spill_tile, ts.spill_count};
    k_tile_render<<<g.nt, kBlock, 0, st>>>(ts.entries, ts.cap, rgb, depth, N, H, W, g, out);
    rc = csm::check_launch("k_tile_render"); if (rc) return rc;
    k_tile_holes<<<1024, kBlock, 0, st>>>(H, W, g, out);
    return csm::check_launch("k_tile_holes");
}

// render_pointcloud (models/utils.py:56-315) for one cloud and any channel count through the tile path: bin -> per-tile z-buffer /
// degrid / splat in channel groups.  scratch: csm_warp_tile_scratch_bytes(H, W, N) bytes whose first csm_warp_tile_header_bytes are zero
// between calls (zeroed once by the caller; every call leaves them zero) -- a csm_warp_frame_tiled scratch of the same size serves.
extern "C" int csm_render_pointcloud_tiled(const float *pts, const float *data, int C, int64_t N, int W, int H, double focal,
                                           double baseline, void *scratch, float *render, float *existing, void *stream) {
    CSM_REQUIRE(scratch && render && existing && C > 0 && N >= 0 && H > 0 && W > 0 && N < (1ll << 29));
    CSM_REQUIRE(N == 0 || (pts && data));
    CSM_REQUIRE((((uintptr_t)scratch) & 15) == 0);
    if (!csm_warp_tile_supported(H, W)) return csm::fail_arg("frame too large for the tiled path (more than 8192 tiles): use csm_render_pointcloud");
    hipStream_t st = (hipStream_t)stream;
    const TileGeom g = tile_geom(H, W);
    const TileScratch ts = carve(scratch, H, W, g.nt, N);
    int rc;
    if (N > 0) {
        k_tile_bin<false><<<(unsigned)bin_blocks(N), kBlock, 2 * sizeof(int) * (size_t)g.nt, st>>>(pts, N, make_proj(H, W, focal, baseline),
                                                                                                   Shift{0, 0, 0}, g, ts.cap, make_point_map(H, W, N), ts.totals,
                                                                                                   ts.entries, ts.spill, ts.spill_tile, ts.spill_count, ts.hole_count);
        rc = csm::check_launch("k_tile_bin"); if (rc) return rc;
    }
    k_tile_render_c<<<g.nt, kBlock, 0, st>>>(ts.entries, ts.cap, data, C, N, H, W, g, ts.totals, ts.spill, ts.spill_tile, ts.spill_count,
                                             render, existing);
    rc = csm::check_launch("k_tile_render_c"); if (rc) return rc;
    CSM_HIP(hipMemsetAsync(ts.spill_count, 0, sizeof(int), st));            // (the frame path's k_tile_holes re-arms it; here nothing runs after the reader)
    return CSM_OK;
}
