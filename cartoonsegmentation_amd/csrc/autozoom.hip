// autozoom.hip -- batched coverage search of process_autozoom (anime_3dkenburns/common.py:86-142) for gfx950.
//
// The reference renders the point cloud once per candidate shift (<= 256 of them: 3 kernel launches, 4 torch ops and one
// `.item()` host sync each) only to count the pixels with `tenExisting > 0`.  Coverage needs the z-buffer (updateZee), its
// degrid pass and the z-test of updateOutput -- but none of the colour accumulation.  Here a CHUNK of candidates is processed by
// four launches, every candidate with its own z-buffer plane, and the per-candidate counts are read by the host ONCE:
//
//   k_az_fill_count  zeeA[k] <- 1e6 for the chunk (and, fused, counts the covered marks the previous chunk left in zeeA)
//   k_az_zee         one thread per POINT, loop over the chunk's candidates: the point is loaded once, everything that only
//                    depends on z (z-ratio of process_shift, ray distance, fltError with its fp64 divide) is computed once --
//                    all candidates of one autozoom share the z shift -- and each candidate costs a few flops + one native
//                    integer atomic (float min, see warp_device.h)
//   k_degrid_batch   Jacobi degrid of all planes of the chunk (zeeA -> zeeB)
//   k_az_cover       one thread per point, loop over candidates: the four z-tests of updateOutput against zeeB; a passing
//                    corner with a positive bilinear weight marks its pixel in zeeA (dead after the degrid) with a NaN
//                    pattern.  Plain stores: every writer writes the same value.
//
// `existing > 0`  <=>  some accumulated weight is positive  <=>  some passing corner has weight > 0 (weights are products of
// non-negative factors), so the marks are exactly the reference's coverage under the Jacobi degrid order; the arithmetic per
// candidate is the statement sequence of load_point<true> / project / corner_weights (warp_device.h), unchanged.
// HBM-bound: 24 B per pixel and candidate (fill 4, atomic RMW 4, degrid 4+4, z-test reads 4, marks + count 4) on top of the
// atomics; a chunk of 32 planes at 1024^2 (256 MB for A and B) stays inside the 256 MB Infinity Cache.
#include "warp_device.h"
#include <algorithm>
#include <cstdlib>
#include <vector>

namespace {
using namespace csmwarp;

constexpr int kBlock = 256;
constexpr int kMaxChunk = 32;
constexpr unsigned kMark = 0xFFFFFFFFu;          // a NaN: never a z value (zee holds finite floats only)
// block-level counts go to kSlots counters per plane, each on its own 128-B line: same-line device-scope atomics serialise at
// the memory side (a single counter per plane made the degrid kernel 2x slower)
constexpr int kSlots = 64, kSlotPitch = 32;

struct Cands { float sx[kMaxChunk], sy[kMaxChunk]; float sz; int n; };

__global__ __launch_bounds__(kBlock) void k_az_fill_count(float *__restrict__ zee, int64_t plane, int n_fill, int n_count,
                                                           int *__restrict__ counts, int *__restrict__ slots) {
    // grid (blocks, max(n_fill, n_count)); plane % 4 == 0 is guaranteed by the host (the scratch pitch is rounded up).
    // Four independent 16-B loads per lane per trip: a streaming pass needs ~10 MB in flight chip-wide to approach the HBM rate
    // (one load per lane = 32 waves x 64 x 16 B x 256 CUs = 8 MB at best, and the count's dependency serialises the trips).
    const int k = blockIdx.y;
    uint4 *Z = reinterpret_cast<uint4 *>(zee + (int64_t)k * plane);
    const int64_t n4 = plane >> 2;
    const uint32_t big = __float_as_uint(1000000.0f);
    const uint4 fill = make_uint4(big, big, big, big);
    const bool do_count = k < n_count, do_fill = k < n_fill;
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    int c = 0;
    if (do_count && blockIdx.x == 0 && threadIdx.x < kSlots) {         // the "certain" pixels k_degrid_batch* counted for this plane
        int *sl = slots + ((int64_t)k * kSlots + threadIdx.x) * kSlotPitch;
        c = *sl; *sl = 0;
    }
    int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    for (; i + 3 * stride < n4; i += 4 * stride) {
        if (do_count) {
            uint4 v0 = Z[i], v1 = Z[i + stride], v2 = Z[i + 2 * stride], v3 = Z[i + 3 * stride];
            c += (v0.x == kMark) + (v0.y == kMark) + (v0.z == kMark) + (v0.w == kMark);
            c += (v1.x == kMark) + (v1.y == kMark) + (v1.z == kMark) + (v1.w == kMark);
            c += (v2.x == kMark) + (v2.y == kMark) + (v2.z == kMark) + (v2.w == kMark);
            c += (v3.x == kMark) + (v3.y == kMark) + (v3.z == kMark) + (v3.w == kMark);
        }
        if (do_fill) { Z[i] = fill; Z[i + stride] = fill; Z[i + 2 * stride] = fill; Z[i + 3 * stride] = fill; }
    }
    for (; i < n4; i += stride) {
        if (do_count) {
            uint4 v = Z[i];
            c += (v.x == kMark) + (v.y == kMark) + (v.z == kMark) + (v.w == kMark);
        }
        if (do_fill) Z[i] = fill;
    }
    if (do_count) {
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) c += __shfl_xor(c, off);
        __shared__ int part[kBlock / 64];
        if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = c;
        __syncthreads();
        if (threadIdx.x == 0) {
            int t = 0;
#pragma unroll
            for (int i2 = 0; i2 < kBlock / 64; ++i2) t += part[i2];
            if (t) atomicAdd(counts + k, t);
        }
    }
}

__global__ __launch_bounds__(kBlock) void k_az_zee(const float *__restrict__ pts, int64_t N, ProjConst pc, Cands cd,
                                                    float *__restrict__ zee, int64_t pitch) {
    const int64_t p = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (p >= N) return;
    for (int k = 0; k < cd.n; ++k) {
        float x, y, z;
        load_point<true>(pts, N, p, Shift{cd.sx[k], cd.sy[k], cd.sz}, x, y, z);
        float fx, fy, err, w[4];
        if (!project(x, y, z, pc, fx, fy, err)) return;      // the rejection tests only involve z: the same for every candidate
        int x0, y0, cx, cy;
        corner_weights(fx, fy, x0, y0, w);
        if (!argmax_corner(w, x0, y0, cx, cy)) continue;
        if (cx >= 0 && cx < pc.W && cy >= 0 && cy < pc.H)
            atomic_min_f32(zee + (int64_t)k * pitch + (int64_t)cy * pc.W + cx, err);
    }
}

// kernel_pointrender_updateDegrid (models/utils.py:152-212), Jacobi form, for `gridDim.z` planes of pitch `pitch`.
// One lane = 4 consecutive pixels of a row (W % 4 == 0 and 16-B aligned planes: the vector path; otherwise the scalar kernel):
// three 16-B row loads + six edge dwords per lane.  A one-pixel-per-lane stencil has 4 unique bytes in flight per lane --
// about 2 MB chip-wide -- and runs at ~1.3 TB/s (latency bound, measured); this form quadruples the bytes in flight.
__device__ __forceinline__ float degrid_one(float c, const float n[8], bool use[4]) {
    // n: neighbours in the order E, W, S, N, SE, NW, NE(x+1,y-1), SW(x-1,y+1); pairs (0,1) (2,3) (4,5) (6,7) = the four lines
    int cnt = 0; float sum = 0.0f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (!use[k]) continue;
        const float a = n[2 * k], d = n[2 * k + 1];
        if ((double)c >= (double)a + 1.0 && (double)c >= (double)d + 1.0) { cnt += 2; sum += a; sum += d; }
    }
    return cnt > 0 ? fminf(c, sum / (float)cnt) : c;
}

// Pixels whose own z-buffer winner survives the degrid (zee < 1e6 and zee <= degridded + 1) are covered for certain: that
// point passes the z-test at its largest-weight corner (weight >= 1/4 > 0).  They are counted here and their degridded value is
// replaced by NaN, which fails every `fltError <= z + 1` of k_az_cover -- about two thirds of its stores (the L2-op-bound part of
// the search) disappear, and marked + certain pixels stay disjoint.
__device__ __forceinline__ float certain_or(float zee, float zdg, int &n) {
    const bool certain = zee < 1000000.0f && ((double)zee <= (double)zdg + 1.0);
    n += certain ? 1 : 0;
    return certain ? __uint_as_float(0x7FC00000u) : zdg;
}

__device__ __forceinline__ void block_count_add(int c, int *__restrict__ dst) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) c += __shfl_xor(c, off);
    __shared__ int part[kBlock / 64];
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) {
        int t = 0;
#pragma unroll
        for (int i = 0; i < kBlock / 64; ++i) t += part[i];
        if (t) atomicAdd(dst, t);
    }
}

__global__ __launch_bounds__(kBlock) void k_degrid_batch4(const float *__restrict__ zin, float *__restrict__ zout, int H, int W,
                                                           int64_t pitch, int *__restrict__ slots) {
    const int x = (blockIdx.x * 64 + (threadIdx.x & 63)) * 4;
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    int ncert = 0;
    if (x < W && y < H) {
    const float *Z = zin + (int64_t)blockIdx.z * pitch;
    const bool up = y > 0, dn = y < H - 1, lf = x > 0, rt = x + 4 < W;
    const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
    const float4 rc = *reinterpret_cast<const float4 *>(Z + (int64_t)y * W + x);
    const float4 ru = up ? *reinterpret_cast<const float4 *>(Z + (int64_t)(y - 1) * W + x) : zero;
    const float4 rd = dn ? *reinterpret_cast<const float4 *>(Z + (int64_t)(y + 1) * W + x) : zero;
    const float cl = lf ? Z[(int64_t)y * W + x - 1] : 0.f, cr = rt ? Z[(int64_t)y * W + x + 4] : 0.f;
    const float ul = (up && lf) ? Z[(int64_t)(y - 1) * W + x - 1] : 0.f, ur = (up && rt) ? Z[(int64_t)(y - 1) * W + x + 4] : 0.f;
    const float dl = (dn && lf) ? Z[(int64_t)(y + 1) * W + x - 1] : 0.f, dr = (dn && rt) ? Z[(int64_t)(y + 1) * W + x + 4] : 0.f;
    const float C[6] = {cl, rc.x, rc.y, rc.z, rc.w, cr};
    const float U[6] = {ul, ru.x, ru.y, ru.z, ru.w, ur};
    const float D[6] = {dl, rd.x, rd.y, rd.z, rd.w, dr};
    float out[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const bool hasl = (j > 0) || lf, hasr = (j < 3) || rt;
        // a line is used only when BOTH of its ends are inside the image (the reference `continue`s otherwise); loop order of the
        // reference: (1,0) (0,1) (1,1) (1,-1) with the +offset end first
        const float n[8] = {C[j + 2], C[j], D[j + 1], U[j + 1], D[j + 2], U[j], U[j + 2], D[j]};
        bool use[4] = {hasl && hasr, up && dn, hasl && hasr && up && dn, hasl && hasr && up && dn};
        out[j] = certain_or(C[j + 1], degrid_one(C[j + 1], n, use), ncert);
    }
    *reinterpret_cast<float4 *>(zout + (int64_t)blockIdx.z * pitch + (int64_t)y * W + x) = make_float4(out[0], out[1], out[2], out[3]);
    }
    block_count_add(ncert, slots + ((int64_t)blockIdx.z * kSlots + ((blockIdx.x + blockIdx.y * gridDim.x) & (kSlots - 1))) * kSlotPitch);
}

__global__ __launch_bounds__(kBlock) void k_degrid_batch(const float *__restrict__ zin, float *__restrict__ zout, int H, int W,
                                                          int64_t pitch, int *__restrict__ slots) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    int ncert = 0;
    if (x < W && y < H) {
    const float *Z = zin + (int64_t)blockIdx.z * pitch;
    const float c = Z[(int64_t)y * W + x];
    int cnt = 0; float sum = 0.0f;
    const int ox[4] = {1, 0, 1, 1}, oy[4] = {0, 1, 1, -1};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        int x1 = x + ox[k], y1 = y + oy[k], x2 = x - ox[k], y2 = y - oy[k];
        if (x1 < 0 || x1 >= W || y1 < 0 || y1 >= H) continue;
        if (x2 < 0 || x2 >= W || y2 < 0 || y2 >= H) continue;
        float a = Z[(int64_t)y1 * W + x1], d = Z[(int64_t)y2 * W + x2];
        if ((double)c >= (double)a + 1.0 && (double)c >= (double)d + 1.0) { cnt += 2; sum += a; sum += d; }
    }
    float r = c;
    if (cnt > 0) r = fminf(c, sum / (float)cnt);
    zout[(int64_t)blockIdx.z * pitch + (int64_t)y * W + x] = certain_or(c, r, ncert);
    }
    block_count_add(ncert, slots + ((int64_t)blockIdx.z * kSlots + ((blockIdx.x + blockIdx.y * gridDim.x) & (kSlots - 1))) * kSlotPitch);
}

__global__ __launch_bounds__(kBlock) void k_az_cover(const float *__restrict__ pts, int64_t N, ProjConst pc, Cands cd,
                                                      const float *__restrict__ zeeB, float *__restrict__ zeeA, int64_t pitch) {
    const int64_t p = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (p >= N) return;
    constexpr int U = 4;                                        // candidates per trip: 4 x 4 scattered z reads in flight per lane
    for (int k0 = 0; k0 < cd.n; k0 += U) {
        int64_t o[U][4]; float zb[U][4], w[U][4], err[U]; bool in[U][4];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int k = k0 + u;
            const bool live = k < cd.n;
            float x, y, z, fx = 0.f, fy = 0.f;
            load_point<true>(pts, N, p, Shift{live ? cd.sx[k] : 0.f, live ? cd.sy[k] : 0.f, cd.sz}, x, y, z);
            err[u] = 0.f;
            if (!project(x, y, z, pc, fx, fy, err[u])) return;  // rejection only involves z: the same for every candidate
            int x0, y0;
            corner_weights(fx, fy, x0, y0, w[u]);
#pragma unroll
            for (int c = 0; c < 4; ++c) {                       // models/utils.py:268-310 without the colour channels
                const int cx = x0 + (c & 1), cy = y0 + (c >> 1);
                in[u][c] = live && cx >= 0 && cx < pc.W && cy >= 0 && cy < pc.H;
                o[u][c] = (int64_t)k * pitch + (in[u][c] ? (int64_t)cy * pc.W + cx : 0);
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int c = 0; c < 4; ++c) zb[u][c] = in[u][c] ? zeeB[o[u][c]] : 0.f;
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                if (!in[u][c]) continue;
                if (!((double)err[u] <= (double)zb[u][c] + 1.0)) continue;
                // the ones channel: atomicAdd(existing, 1.0 * w); `existing > 0` <=> some passing weight is positive
                if (1.0f * w[u][c] > 0.0f) reinterpret_cast<unsigned *>(zeeA)[o[u][c]] = kMark;
            }
    }
}


// ---- band path -------------------------------------------------------------------------------------------------------------------
// All candidates of one search share the z shift, and the 16 x 16 grid of common.py:96-108 has 16 distinct y shifts: candidates
// with the same (sy, sz) differ only in fx (warp_device.h::project_yz / project_x) -- a point lands in the same ROWS for all of
// them.  So: group the candidates by sy; per group bin the points ONCE by destination row band (BR rows, full width); then one
// 1024-thread block per (band, group) keeps the band's z-buffer (+ 1-row halo) and its degridded copy in LDS, holds the band's
// entries {x r, dist, fy, err} in registers, and runs z pass / Jacobi degrid / z-test marks / count for each of the group's
// candidates without touching HBM: per candidate the chip moves nothing but a 4-byte partial count per band.  (The plane path
// above streams 24 B per pixel and candidate through L2 atomics and is bound by L2 lane operations: 22.8 us per candidate at
// 1024^2.)  A band segment that overflows its capacity raises a flag the caller reads together with the counts and falls back
// to the plane path, so the result is exact for every cloud.
constexpr int kGroupMax = 16, kPerGroup = 16;          // groups per launch, candidates per group
constexpr int kBandThreads = 1024;
static size_t kBandLds = 150 * 1024;                   // LDS per block: window + degridded band + as many cached entries as fit
                                                        // (CSM_AZ_LDS_KB overrides: a measurement knob)
constexpr int kCountStride = 32;                        // ints between two segment counters (one 128-B line each)
struct Groups { int ng; float sz; float sy[kGroupMax]; int n[kGroupMax]; float sx[kGroupMax][kPerGroup]; int out[kGroupMax][kPerGroup]; };
struct BandEntry { float xr, dist, fy, err; };          // x * z/(z + 1e-7), ray factor, projected row coordinate, fltError
struct BandGeom { int br, nbands, cap, ncache; };       // ncache: entries of a band kept in LDS (the rest is re-read per candidate)

// largest float t with (double)t <= d (NaN stays NaN): turns the reference's mixed-precision tests `(double)e <= (double)z + 1.0`
// into ONE fp32 compare against a per-pixel threshold -- e <= t  <=>  (double)e <= d, because the floats <= d are exactly those <= t
__device__ __forceinline__ float round_down_f32(double d) {
    float t = (float)d;
    if ((double)t > d) {
        const unsigned u = __float_as_uint(t);
        t = t > 0.0f ? __uint_as_float(u - 1u) : (t < 0.0f ? __uint_as_float(u + 1u) : __uint_as_float(0x80000001u));
    }
    return t;
}

// bands whose window rows [b*br - 1, b*br + br] meet the footprint rows [y0, y0 + 1]
__device__ __forceinline__ void band_range(int y0, int br, int nb, int &lo, int &hi) {
    const int a = y0 - br;                                                     // b >= ceil((y0 - br) / br)
    lo = a >= 0 ? (a + br - 1) / br : -((-a) / br);
    const int c = y0 + 2;                                                      // b <= floor((y0 + 2) / br)
    hi = c >= 0 ? c / br : -((-c + br - 1) / br);
    lo = lo < 0 ? 0 : lo;
    hi = hi > nb - 1 ? nb - 1 : hi;
}

// (Groups travels through device memory: a 2 KB by-value kernel argument indexed by blockIdx is copied to scratch by hipcc --
// 588 B of private memory per lane and 6 ms per search in the first version of k_band_cover)
__global__ __launch_bounds__(kBlock) void k_band_bin(const float *__restrict__ pts, int64_t N, ProjConst pc, const Groups *__restrict__ grp,
                                                      BandGeom bg, int *__restrict__ seg_count, BandEntry *__restrict__ entries,
                                                      int *__restrict__ overflow) {
    extern __shared__ int lds_i[];                     // hist[nbands] | base[nbands]
    int *hist = lds_i, *base = lds_i + bg.nbands;
    const int g = blockIdx.y;
    const float gsy = grp->sy[g], gsz = grp->sz;
    for (int t = threadIdx.x; t < bg.nbands; t += kBlock) hist[t] = 0;
    __syncthreads();
    constexpr int kPPT = 4;
    BandEntry e[kPPT]; int lo[kPPT], hi[kPPT], rank[kPPT][3];
#pragma unroll
    for (int i = 0; i < kPPT; ++i) {
        const int64_t p = ((int64_t)blockIdx.x * kPPT + i) * kBlock + threadIdx.x;
        lo[i] = 0; hi[i] = -1;
        e[i] = BandEntry{0.f, 0.f, 0.f, 0.f};
        if (p < N) {
            float x = pts[p], y = pts[N + p], z = pts[2 * N + p];
            const float r = z / (z + 0.0000001f);                               // common.py:78-81, the x shift is added per candidate
            x = x * r; y = y * r + gsy; z = z + gsz;
            float dist, fy, err;
            if (project_yz(y, z, pc, dist, fy, err) && fy > -4.0f && fy < (float)(pc.H + 4)) {
                e[i] = BandEntry{x, dist, fy, err};
                band_range((int)floorf(fy), bg.br, bg.nbands, lo[i], hi[i]);
            }
        }
#pragma unroll
        for (int j = 0; j < 3; ++j) rank[i][j] = lo[i] + j <= hi[i] ? atomicAdd(&hist[lo[i] + j], 1) : -1;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < kPPT; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j)
            if (rank[i][j] == 0) {
                const int b = lo[i] + j;
                base[b] = atomicAdd(seg_count + ((int64_t)g * bg.nbands + b) * kCountStride, hist[b]);
            }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < kPPT; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            if (rank[i][j] < 0) continue;
            const int b = lo[i] + j, slot = base[b] + rank[i][j];
            if (slot < bg.cap) entries[((int64_t)g * bg.nbands + b) * bg.cap + slot] = e[i];
            else *overflow = 1;
        }
}

__global__ __launch_bounds__(kBandThreads) void k_band_cover(ProjConst pc, const Groups *__restrict__ grp, BandGeom bg, int *__restrict__ seg_count,
                                                              const BandEntry *__restrict__ entries, int *__restrict__ partial /* [ng][kPerGroup][nbands] */) {
    extern __shared__ __attribute__((aligned(16))) float lds_f[];   // zee[(br + 2) * W] | zd[br * W] | cached entries
    const int W = pc.W, H = pc.H, br = bg.br;
    float *zee = lds_f, *zd = lds_f + (size_t)(br + 2) * W;
    __shared__ int red[kBandThreads / 64];
    const int band = blockIdx.x, g = blockIdx.y, tid = threadIdx.x;
    const int r0 = band * br;                          // first interior row; window row 0 is image row r0 - 1
    int *cnt = seg_count + ((int64_t)g * bg.nbands + band) * kCountStride;
    const int total = *cnt < bg.cap ? *cnt : bg.cap;
    const BandEntry *E = entries + ((int64_t)g * bg.nbands + band) * bg.cap;
    BandEntry *ce = reinterpret_cast<BandEntry *>(zd + (size_t)br * W);         // LDS cache of the band's entries (16-B aligned: W % 2 == 0 checked by the host)
    const int ncache = total < bg.ncache ? total : bg.ncache;
    for (int i = tid; i < ncache; i += kBandThreads) ce[i] = E[i];
    const int nwin = (br + 2) * W, nint = br * W;
    const int ncand = grp->n[g];
    for (int i = tid; i < nwin; i += kBandThreads) zee[i] = 1000000.0f;         // models/utils.py:59 (later trips: the count sweep)
    for (int c = 0; c < ncand; ++c) {
        const float sx = grp->sx[g][c];
        __syncthreads();                                                        // window armed (first trip: the entry cache too)
        // ---- updateZee (models/utils.py:101-147)
        auto zpass = [&](const BandEntry &e) {
            const float fx = project_x(e.xr + sx, e.dist, pc);
            int x0, y0, cx, cy; float w[4];
            corner_weights(fx, e.fy, x0, y0, w);
            if (!argmax_corner(w, x0, y0, cx, cy)) return;
            const int ly = cy - (r0 - 1);
            if (cx >= 0 && cx < W && cy >= 0 && cy < H && ly >= 0 && ly < br + 2) {
                float *a = &zee[ly * W + cx];
                if (e.err >= 0.0f) atomicMin(reinterpret_cast<int *>(a), __float_as_int(e.err));
                else atomicMax(reinterpret_cast<unsigned int *>(a), __float_as_uint(e.err));
            }
        };
        for (int i = tid; i < total; i += kBandThreads) zpass(i < ncache ? ce[i] : E[i]);
        __syncthreads();
        // ---- updateDegrid (models/utils.py:152-212), Jacobi form, + the "certain" pixels (see certain_or)
        int n = 0;
        for (int i = tid; i < nint; i += kBandThreads) {
            const int ly = i / W, x = i - ly * W, y = r0 + ly;
            if (y >= H) { zd[i] = __uint_as_float(0x7FC00000u); continue; }
            const float *Zc = zee + (ly + 1) * W + x;
            const float cz = Zc[0];
            const bool up = y > 0, dn = y < H - 1, lf = x > 0, rt = x < W - 1;
            int cnt2 = 0; float sum = 0.0f;
            // The reference tests `(double)cz >= (double)a + 1.0` per neighbour.  For |a| >= 2^-28 (or a == 0) the sum a + 1.0 is
            // exact in double, and so is cz - 1.0 for |cz| >= 2^-28 (or 0), hence the test equals (double)a <= (double)cz - 1.0
            // = a <= thr with thr the largest float <= cz - 1.0: one threshold per pixel, fp32 compares per neighbour.
            const float thr = round_down_f32((double)cz - 1.0);
            // a value for which the threshold form is not exact: non-zero and below 2^-28 (0x31800000).  Such pixels (none in a real
            // cloud) are recomputed with the reference's expression below; the common path carries three integer ops per value read.
            // (Selecting per neighbour between the fp32 compare and the double expression cost 0.95 ms per 256-candidate search.)
            auto tiny = [](float v) { return ((__float_as_uint(v) & 0x7FFFFFFFu) - 1u) < 0x317FFFFFu; };
            bool slow = tiny(cz);
            auto line = [&](float a, float d, auto below) { if (below(a) && below(d)) { cnt2 += 2; sum += a; sum += d; } };
            auto degrid = [&](auto below) {
                // loop order of the reference: (1,0) (0,1) (1,1) (1,-1), the +offset end first; a line counts only if both ends are inside
                cnt2 = 0; sum = 0.0f;
                if (lf && rt) line(Zc[1], Zc[-1], below);
                if (up && dn) line(Zc[W], Zc[-W], below);
                if (lf && rt && up && dn) { line(Zc[W + 1], Zc[-W - 1], below); line(Zc[-W + 1], Zc[W - 1], below); }
            };
            degrid([&](float a) { slow = slow || tiny(a); return a <= thr; });
            if (slow) degrid([&](float a) { return (double)cz >= (double)a + 1.0; });
            const float r = cnt2 > 0 ? fminf(cz, sum / (float)cnt2) : cz;
            // zd holds the z-test THRESHOLD of the pixel, not the degridded value: err passes  <=>  (double)err <= (double)r + 1.0
            zd[i] = round_down_f32((double)certain_or(cz, r, n) + 1.0);
        }
        __syncthreads();
        // ---- the z-tests of updateOutput (models/utils.py:268-310): a passing corner with positive weight marks its pixel
        auto cover = [&](const BandEntry &e) {
            const float fx = project_x(e.xr + sx, e.dist, pc);
            int x0, y0; float w[4];
            corner_weights(fx, e.fy, x0, y0, w);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int cx = x0 + (q & 1), cy = y0 + (q >> 1), ly = cy - r0;
                // inside the image and the band's interior rows (cy >= 0 follows from ly >= 0)
                if ((unsigned)cx >= (unsigned)W || (unsigned)ly >= (unsigned)br || cy >= H) continue;
                if (!(e.err <= zd[ly * W + cx])) continue;                     // the per-pixel threshold of the degrid pass
                if (1.0f * w[q] > 0.0f) reinterpret_cast<unsigned *>(zee)[(ly + 1) * W + cx] = kMark;
            }
        };
        for (int i = tid; i < total; i += kBandThreads) cover(i < ncache ? ce[i] : E[i]);
        __syncthreads();
        // count the marks and re-arm the window for the next candidate in one sweep (mark rows are the interior rows 1 .. br)
        for (int i = tid; i < nwin; i += kBandThreads) {
            n += reinterpret_cast<const unsigned *>(zee)[i] == kMark ? 1 : 0;   // halo rows never hold a mark
            zee[i] = 1000000.0f;
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) n += __shfl_xor(n, off);
        if ((tid & 63) == 0) red[tid >> 6] = n;
        __syncthreads();
        if (tid == 0) {
            int t = 0;
#pragma unroll
            for (int i = 0; i < kBandThreads / 64; ++i) t += red[i];
            partial[((int64_t)g * kPerGroup + c) * bg.nbands + band] = t;
        }
    }
    if (tid == 0) *cnt = 0;                            // the segment counter is re-armed for the next launch
}

__global__ __launch_bounds__(kBlock) void k_band_sum(const Groups *__restrict__ grp, int nbands, const int *__restrict__ partial,
                                                      int *__restrict__ counts) {
    const int g = blockIdx.y, c = blockIdx.x;
    if (c >= grp->n[g]) return;
    int t = 0;
    for (int b = threadIdx.x; b < nbands; b += kBlock) t += partial[((int64_t)g * kPerGroup + c) * nbands + b];
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) t += __shfl_xor(t, off);
    __shared__ int part[kBlock / 64];
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = t;
    __syncthreads();
    if (threadIdx.x == 0) { int s2 = 0; for (int i = 0; i < kBlock / 64; ++i) s2 += part[i]; counts[grp->out[g][c]] = s2; }
}

inline BandGeom band_geom(int H, int W, int64_t N) {
    static bool env_read = false;
    if (!env_read) { const char *e = getenv("CSM_AZ_LDS_KB"); if (e && atoi(e) >= 64 && atoi(e) <= 156) kBandLds = (size_t)atoi(e) * 1024; env_read = true; }
    BandGeom bg; bg.br = 0; bg.nbands = 0; bg.cap = 0; bg.ncache = 0;
    // narrow bands leave LDS for the entry cache: 4 rows up to W = 1228, 2 rows up to W = 3072 (window + degridded band <= 48 / 72 KB)
    if (W % 2) return bg;                                                     // the entry cache must start 16-B aligned
    if ((size_t)(2 * 8 + 2) * W * 4 <= 40 * 1024) bg.br = 8;
    else if ((size_t)(2 * 4 + 2) * W * 4 <= 48 * 1024) bg.br = 4;
    else if ((size_t)(2 * 2 + 2) * W * 4 <= 72 * 1024) bg.br = 2;
    else return bg;
    bg.nbands = (H + bg.br - 1) / bg.br;
    const double per_band = (double)N * (bg.br + 3) / (double)H;               // uniform cloud: rows y0 in [b br - 2, b br + br]
    int64_t cap = (int64_t)(per_band * 3.0) + 1023;
    cap -= cap % 1024;
    bg.cap = (int)(cap < 4096 ? 4096 : (cap > (1 << 26) ? (1 << 26) : cap));
    const size_t window = (size_t)(2 * bg.br + 2) * W * 4;
    const size_t room = (kBandLds - window) / 16;
    bg.ncache = (int)(room < (size_t)bg.cap ? room : (size_t)bg.cap);
    return bg;
}
inline size_t band_align(size_t v) { return (v + 255) & ~(size_t)255; }

}  // namespace

extern "C" int csm_autozoom_max_chunk(void) { return kMaxChunk; }

extern "C" size_t csm_autozoom_scratch_floats(int H, int W, int chunk) {
    if (chunk < 1) chunk = 1;
    if (chunk > kMaxChunk) chunk = kMaxChunk;
    size_t pitch = (((size_t)H * (size_t)W + 3) / 4) * 4;
    return 2 * pitch * (size_t)chunk + (size_t)chunk * kSlots * kSlotPitch;
}

extern "C" int csm_autozoom_coverage(const float *pts, int64_t N, int H, int W, double focal, double baseline,
                                     const float *shifts_xy, float shift_z, int K, int chunk, float *scratch, int *counts,
                                     void *stream) {
    CSM_REQUIRE(scratch && counts && H > 0 && W > 0 && N >= 0 && K >= 0 && (K == 0 || shifts_xy) && (N == 0 || pts));
    hipStream_t st = (hipStream_t)stream;
    if (K == 0) return CSM_OK;
    if (chunk < 1) chunk = 1;
    if (chunk > kMaxChunk) chunk = kMaxChunk;
    CSM_HIP(hipMemsetAsync(counts, 0, sizeof(int) * (size_t)K, st));
    const int64_t plane = (int64_t)H * W, pitch = ((plane + 3) / 4) * 4;
    float *zeeA = scratch, *zeeB = scratch + pitch * chunk;
    int *slots = reinterpret_cast<int *>(scratch + 2 * pitch * chunk);
    CSM_HIP(hipMemsetAsync(slots, 0, sizeof(int) * (size_t)chunk * kSlots * kSlotPitch, st));
    const ProjConst pc = make_proj(H, W, focal, baseline);
    const unsigned fill_blocks = csm::cdiv(pitch / 4, kBlock * 4) < 1 ? 1 : csm::cdiv(pitch / 4, kBlock * 4);
    int prev = 0;                                              // candidates of the previous chunk whose marks are still in zeeA
    for (int k0 = 0; k0 < K || prev > 0; k0 += chunk) {
        const int n = k0 < K ? (K - k0 < chunk ? K - k0 : chunk) : 0;
        const int ny = n > prev ? n : prev;
        k_az_fill_count<<<dim3(fill_blocks, ny), kBlock, 0, st>>>(zeeA, pitch, n, prev, counts + (k0 - chunk < 0 ? 0 : k0 - chunk), slots);
        int rc = csm::check_launch("k_az_fill_count"); if (rc) return rc;
        prev = n;
        if (n == 0) break;
        Cands cd;
        for (int i = 0; i < n; ++i) { cd.sx[i] = shifts_xy[2 * (k0 + i)]; cd.sy[i] = shifts_xy[2 * (k0 + i) + 1]; }
        cd.sz = shift_z; cd.n = n;
        if (N > 0) {
            k_az_zee<<<csm::cdiv(N, kBlock), kBlock, 0, st>>>(pts, N, pc, cd, zeeA, pitch);
            rc = csm::check_launch("k_az_zee"); if (rc) return rc;
        }
        if ((W & 3) == 0 && (((uintptr_t)scratch) & 15) == 0)
            k_degrid_batch4<<<dim3(csm::cdiv(W, 256), csm::cdiv(H, 4), n), kBlock, 0, st>>>(zeeA, zeeB, H, W, pitch, slots);
        else
            k_degrid_batch<<<dim3(csm::cdiv(W, 64), csm::cdiv(H, 4), n), kBlock, 0, st>>>(zeeA, zeeB, H, W, pitch, slots);
        rc = csm::check_launch("k_degrid_batch"); if (rc) return rc;
        if (N > 0) {
            k_az_cover<<<csm::cdiv(N, kBlock), kBlock, 0, st>>>(pts, N, pc, cd, zeeB, zeeA, pitch);
            rc = csm::check_launch("k_az_cover"); if (rc) return rc;
        }
    }
    return CSM_OK;
}

extern "C" int csm_autozoom_band_supported(int H, int W) {
    if (H <= 0 || W <= 0) return 0;
    const BandGeom bg = band_geom(H, W, 1);
    return bg.br > 0 && bg.nbands <= 4096;
}

// scratch: segment counters [kGroupMax * nbands * kCountStride] (ZEROED by the first launch of every call) | partial counts
// [kGroupMax * kPerGroup * nbands] | entries [kGroupMax * nbands * cap] (16 B)
extern "C" size_t csm_autozoom_band_scratch_bytes(int H, int W, int64_t N) {
    if (!csm_autozoom_band_supported(H, W) || N < 0) return 0;
    const BandGeom bg = band_geom(H, W, N);
    return band_align(4 * (size_t)kGroupMax * bg.nbands * kCountStride) + band_align(4 * (size_t)kGroupMax * kPerGroup * bg.nbands) +
           band_align(sizeof(Groups)) * 64 + 16 * (size_t)kGroupMax * bg.nbands * bg.cap + 256;
}

extern "C" int csm_autozoom_coverage_bands(const float *pts, int64_t N, int H, int W, double focal, double baseline,
                                           const float *shifts_xy, float shift_z, int K, void *scratch, int *counts, int *overflow,
                                           void *stream) {
    CSM_REQUIRE(scratch && counts && overflow && H > 0 && W > 0 && N >= 0 && K >= 0 && (K == 0 || shifts_xy) && (N == 0 || pts));
    CSM_REQUIRE((((uintptr_t)scratch) & 15) == 0);
    if (!csm_autozoom_band_supported(H, W)) return csm::fail_arg("frame too wide for the band path: use csm_autozoom_coverage");
    hipStream_t st = (hipStream_t)stream;
    CSM_HIP(hipMemsetAsync(overflow, 0, sizeof(int), st));
    if (K == 0) return CSM_OK;
    const BandGeom bg = band_geom(H, W, N);
    const ProjConst pc = make_proj(H, W, focal, baseline);
    char *p = (char *)scratch;
    int *seg_count = (int *)p; const size_t seg_bytes = band_align(4 * (size_t)kGroupMax * bg.nbands * kCountStride); p += seg_bytes;
    int *partial = (int *)p; p += band_align(4 * (size_t)kGroupMax * kPerGroup * bg.nbands);
    char *grp_slots = p; p += band_align(sizeof(Groups)) * 64;                  // one slot per launch batch (64 x 256 candidates)
    BandEntry *entries = (BandEntry *)p;
    CSM_HIP(hipMemsetAsync(seg_count, 0, seg_bytes, st));
    // group the candidates by their y shift (exact float equality), at most kPerGroup per group, kGroupMax groups per launch
    std::vector<int> order(K);
    for (int i = 0; i < K; ++i) order[i] = i;
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return shifts_xy[2 * a + 1] < shifts_xy[2 * b + 1]; });
    const size_t lds = (size_t)(2 * bg.br + 2) * W * 4 + (size_t)bg.ncache * 16;
    // per device and per call (a search is one call: the microsecond does not matter, and no cross-thread / cross-device state is kept)
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_band_cover), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kBandLds);
    int i = 0, batch = 0;
    while (i < K) {
        if (batch >= 64) return csm::fail_arg("autozoom band path: more than 64 launch batches (K too large): use csm_autozoom_coverage");
        Groups gr; gr.ng = 0; gr.sz = shift_z;
        while (i < K && gr.ng < kGroupMax) {
            const int g = gr.ng++;
            gr.sy[g] = shifts_xy[2 * order[i] + 1]; gr.n[g] = 0;
            while (i < K && gr.n[g] < kPerGroup && shifts_xy[2 * order[i] + 1] == gr.sy[g]) {
                gr.sx[g][gr.n[g]] = shifts_xy[2 * order[i]]; gr.out[g][gr.n[g]] = order[i]; ++gr.n[g]; ++i;
            }
        }
        Groups *grp = reinterpret_cast<Groups *>(grp_slots + (size_t)batch * band_align(sizeof(Groups)));
        ++batch;
        CSM_HIP(hipMemcpyAsync(grp, &gr, sizeof(Groups), hipMemcpyHostToDevice, st));   // pageable source: staged before the call returns
        if (N > 0) {
            k_band_bin<<<dim3((unsigned)csm::cdiv(N, kBlock * 4), gr.ng), kBlock, 2 * sizeof(int) * (size_t)bg.nbands, st>>>(
                pts, N, pc, grp, bg, seg_count, entries, overflow);
            int rc = csm::check_launch("k_band_bin"); if (rc) return rc;
        }
        k_band_cover<<<dim3(bg.nbands, gr.ng), kBandThreads, lds, st>>>(pc, grp, bg, seg_count, entries, partial);
        int rc = csm::check_launch("k_band_cover"); if (rc) return rc;
        k_band_sum<<<dim3(kPerGroup, gr.ng), kBlock, 0, st>>>(grp, bg.nbands, partial, counts);
        rc = csm::check_launch("k_band_sum"); if (rc) return rc;
    }
    return CSM_OK;
}
