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

struct Cands { float sx[kMaxChunk], sy[kMaxChunk]; float sz; int n; };

__global__ __launch_bounds__(kBlock) void k_az_fill_count(float *__restrict__ zee, int64_t plane, int n_fill, int n_count,
                                                           int *__restrict__ counts) {
    // grid (blocks, max(n_fill, n_count)); plane % 4 == 0 is guaranteed by the host (the scratch pitch is rounded up)
    const int k = blockIdx.y;
    uint4 *Z = reinterpret_cast<uint4 *>(zee + (int64_t)k * plane);
    const int64_t n4 = plane >> 2;
    const uint32_t big = __float_as_uint(1000000.0f);
    int c = 0;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n4; i += (int64_t)gridDim.x * kBlock) {
        if (k < n_count) {
            uint4 v = Z[i];
            c += (v.x == kMark) + (v.y == kMark) + (v.z == kMark) + (v.w == kMark);
        }
        if (k < n_fill) Z[i] = make_uint4(big, big, big, big);
    }
    if (k < n_count) {
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) c += __shfl_xor(c, off);
        __shared__ int part[kBlock / 64];
        if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = c;
        __syncthreads();
        if (threadIdx.x == 0) {
            int t = 0;
#pragma unroll
            for (int i = 0; i < kBlock / 64; ++i) t += part[i];
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

// kernel_pointrender_updateDegrid (models/utils.py:152-212), Jacobi form, for `gridDim.z` planes of pitch `pitch`
__global__ __launch_bounds__(kBlock) void k_degrid_batch(const float *__restrict__ zin, float *__restrict__ zout, int H, int W,
                                                          int64_t pitch) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= W || y >= H) return;
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
    zout[(int64_t)blockIdx.z * pitch + (int64_t)y * W + x] = r;
}

__global__ __launch_bounds__(kBlock) void k_az_cover(const float *__restrict__ pts, int64_t N, ProjConst pc, Cands cd,
                                                      const float *__restrict__ zeeB, float *__restrict__ zeeA, int64_t pitch) {
    const int64_t p = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (p >= N) return;
    for (int k = 0; k < cd.n; ++k) {
        float x, y, z;
        load_point<true>(pts, N, p, Shift{cd.sx[k], cd.sy[k], cd.sz}, x, y, z);
        float fx, fy, err, w[4];
        if (!project(x, y, z, pc, fx, fy, err)) return;
        int x0, y0;
        corner_weights(fx, fy, x0, y0, w);
        const float *ZB = zeeB + (int64_t)k * pitch;
        unsigned *ZA = reinterpret_cast<unsigned *>(zeeA + (int64_t)k * pitch);
#pragma unroll
        for (int c = 0; c < 4; ++c) {                           // models/utils.py:268-310 without the colour channels
            const int cx = x0 + (c & 1), cy = y0 + (c >> 1);
            if (cx < 0 || cx >= pc.W || cy < 0 || cy >= pc.H) continue;
            const int64_t o = (int64_t)cy * pc.W + cx;
            if (!((double)err <= (double)ZB[o] + 1.0)) continue;
            if (1.0f * w[c] > 0.0f) ZA[o] = kMark;             // the ones channel: atomicAdd(existing, 1.0 * w) ; existing > 0
        }
    }
}

}  // namespace

extern "C" int csm_autozoom_max_chunk(void) { return kMaxChunk; }

extern "C" size_t csm_autozoom_scratch_floats(int H, int W, int chunk) {
    if (chunk < 1) chunk = 1;
    if (chunk > kMaxChunk) chunk = kMaxChunk;
    size_t pitch = (((size_t)H * (size_t)W + 3) / 4) * 4;
    return 2 * pitch * (size_t)chunk;
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
    const ProjConst pc = make_proj(H, W, focal, baseline);
    const unsigned fill_blocks = csm::cdiv(pitch / 4, kBlock * 4) < 1 ? 1 : csm::cdiv(pitch / 4, kBlock * 4);
    int prev = 0;                                              // candidates of the previous chunk whose marks are still in zeeA
    for (int k0 = 0; k0 < K || prev > 0; k0 += chunk) {
        const int n = k0 < K ? (K - k0 < chunk ? K - k0 : chunk) : 0;
        const int ny = n > prev ? n : prev;
        k_az_fill_count<<<dim3(fill_blocks, ny), kBlock, 0, st>>>(zeeA, pitch, n, prev, counts + (k0 - chunk < 0 ? 0 : k0 - chunk));
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
        k_degrid_batch<<<dim3(csm::cdiv(W, 64), csm::cdiv(H, 4), n), kBlock, 0, st>>>(zeeA, zeeB, H, W, pitch);
        rc = csm::check_launch("k_degrid_batch"); if (rc) return rc;
        if (N > 0) {
            k_az_cover<<<csm::cdiv(N, kBlock), kBlock, 0, st>>>(pts, N, pc, cd, zeeB, zeeA, pitch);
            rc = csm::check_launch("k_az_cover"); if (rc) return rc;
        }
    }
    return CSM_OK;
}
