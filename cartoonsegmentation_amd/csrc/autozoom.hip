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
