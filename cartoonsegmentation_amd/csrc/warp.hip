// warp.hip -- Ken Burns point-cloud render / disocclusion fill / point-wise kernels for gfx950.
//
// What is computed follows the reference's kernels statement by statement (citations are
// relative to /root/reference); how it is computed is CDNA4-native:
//   * AOT-compiled, shapes are runtime arguments (the reference re-JITs per shape through NVRTC),
//   * float atomicMin via one native integer atomic (no CAS loop),
//   * deterministic Jacobi degrid (the reference pass is an in-place race),
//   * process_shift / ones-channel / divide / depth-mask / uint8 conversion fused into the
//     neighbouring kernels, so a frame is 6 launches and never materialises the shifted cloud.
// All of these are HBM/L2-atomic bound scatter/stencil kernels: 256-thread blocks, one point or
// pixel per lane, planar (channel-major) accumulators so that the 64 lanes of a wave hit 2-4
// cache lines per atomic instruction.
//
// Built with -ffp-contract=off: every fp32/fp64 operation rounds exactly once, as written, so
// integer/decision results are bit-identical to oracle/warp_oracle.c.
#include "warp_device.h"

namespace {
using namespace csmwarp;

constexpr int kBlock = 256;

__global__ __launch_bounds__(kBlock) void k_fill(float *__restrict__ a, int64_t na, float va,
                                                  float *__restrict__ b, int64_t nb, float vb) {
    // vectorised fill of two ranges (zee <- 1e6, accumulators <- 0); na, nb multiples of 4 not required
    int64_t i = ((int64_t)blockIdx.x * kBlock + threadIdx.x) * 4;
    int64_t stride = (int64_t)gridDim.x * kBlock * 4;
    for (int64_t j = i; j < na; j += stride) {
        if (j + 3 < na && (((uintptr_t)(a + j)) & 15) == 0) *reinterpret_cast<float4 *>(a + j) = make_float4(va, va, va, va);
        else for (int k = 0; k < 4 && j + k < na; ++k) a[j + k] = va;
    }
    for (int64_t j = i; j < nb; j += stride) {
        if (j + 3 < nb && (((uintptr_t)(b + j)) & 15) == 0) *reinterpret_cast<float4 *>(b + j) = make_float4(vb, vb, vb, vb);
        else for (int k = 0; k < 4 && j + k < nb; ++k) b[j + k] = vb;
    }
}

// kernel_pointrender_updateZee  (models/utils.py:63-149)
template <bool SHIFT>
__global__ __launch_bounds__(kBlock) void k_update_zee(const float *__restrict__ pts, int64_t N, int64_t total,
                                                        ProjConst pc, Shift s, float *__restrict__ zee) {
    int64_t idx = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (idx >= total) return;
    int64_t b = idx / N, p = idx - b * N;
    float x, y, z;
    load_point<SHIFT>(pts + b * 3 * N, N, p, s, x, y, z);
    float fx, fy, err, w[4];
    if (!project(x, y, z, pc, fx, fy, err)) return;
    int x0, y0;
    corner_weights(fx, fy, x0, y0, w);
    float nw = w[0], ne = w[1], sw = w[2], se = w[3];
    int cx, cy;
    if (nw >= ne && nw >= sw && nw >= se) { cx = x0; cy = y0; }
    else if (ne >= nw && ne >= sw && ne >= se) { cx = x0 + 1; cy = y0; }
    else if (sw >= nw && sw >= ne && sw >= se) { cx = x0; cy = y0 + 1; }
    else if (se >= nw && se >= ne && se >= sw) { cx = x0 + 1; cy = y0 + 1; }
    else return;
    if (cx >= 0 && cx < pc.W && cy >= 0 && cy < pc.H)
        atomic_min_f32(zee + (b * pc.H + cy) * (int64_t)pc.W + cx, err);
}

// kernel_pointrender_updateDegrid  (models/utils.py:152-212), Jacobi form
__global__ __launch_bounds__(kBlock) void k_degrid(const float *__restrict__ zin, float *__restrict__ zout, int B,
                                                    int H, int W) {
    int x = blockIdx.x * 64 + (threadIdx.x & 63);
    int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    int b = blockIdx.z;
    if (x >= W || y >= H) return;
    const float *Z = zin + (int64_t)b * H * W;
    float c = Z[(int64_t)y * W + x];
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
    zout[((int64_t)b * H + y) * W + x] = r;
}

// kernel_pointrender_updateOutput  (models/utils.py:215-313)
// data = two channel segments (d0: C0 channels, d1: C1n channels) so the frame path needs no torch.cat;
// the reference's appended ones channel (models/utils.py:57) is the implicit last channel.
template <bool SHIFT, int CT>
__global__ __launch_bounds__(kBlock) void k_update_output(const float *__restrict__ pts,
                                                           const float *__restrict__ d0, int C0,
                                                           const float *__restrict__ d1, int C1n, int64_t N,
                                                           int64_t total, ProjConst pc, Shift s,
                                                           const float *__restrict__ zee,
                                                           float *__restrict__ accum) {
    int64_t idx = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    const bool live = idx < total;
    const int64_t b = live ? idx / N : 0, p = live ? idx - b * N : 0;
    float x = 0.f, y = 0.f, z = 0.f;
    if (live) load_point<SHIFT>(pts + b * 3 * N, N, p, s, x, y, z);
    float fx = 0.f, fy = 0.f, err = 0.f, w[4] = {0.f, 0.f, 0.f, 0.f};
    const bool proj = live && project(x, y, z, pc, fx, fy, err);
    int x0 = 0, y0 = 0;
    if (proj) corner_weights(fx, fy, x0, y0, w);
    const int C = CT > 0 ? CT : (C0 + C1n);
    const int64_t plane = (int64_t)pc.H * pc.W;
    const float *D0 = d0 + b * C0 * N;
    const float *D1 = d1 ? d1 + b * C1n * N : nullptr;
    float *A = accum + b * (C + 1) * plane;
    const float *Z = zee + b * plane;
    if constexpr (CT > 0) {
        // Fixed channel count (frame loop, C = 4; autozoom, C = 3).  The float atomics bound this kernel (20 per point at
        // C = 4), and a wave's 64 lanes are 64 consecutive source pixels: after the camera shift lane i's east corners are
        // very often lane i+1's west corners.  Merge them in registers (two cross-lane moves per channel) and issue one
        // atomic instead of two -- the sum of a pixel is order-free up to fp32 rounding either way.
        float v[CT];
#pragma unroll
        for (int c = 0; c < CT; ++c) v[c] = proj ? (c < C0 ? D0[(int64_t)c * N + p] : D1[(int64_t)(c - C0) * N + p]) : 0.f;
        int o[4]; bool ok[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            int cx = x0 + (k & 1), cy = y0 + (k >> 1);
            bool in = proj && cx >= 0 && cx < pc.W && cy >= 0 && cy < pc.H;
            o[k] = in ? cy * pc.W + cx : -1;
            ok[k] = in && ((double)err <= (double)Z[o[k]] + 1.0);
        }
        float cw[4][CT + 1];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
#pragma unroll
            for (int c = 0; c < CT; ++c) cw[k][c] = v[c] * w[k];
            cw[k][CT] = 1.0f * w[k];
        }
        const int lane = threadIdx.x & 63;
        const int bl = (int)b, b_left = __shfl_up(bl, 1);
#pragma unroll
        for (int h = 0; h < 2; ++h) {                            // h = 0: north pair (NE -> NW), h = 1: south pair (SE -> SW)
            const int dst = 2 * h, src = 2 * h + 1;
            const int o_left = __shfl_up(ok[src] ? o[src] : -2, 1);
            const bool merge = lane > 0 && ok[dst] && o_left == o[dst] && b_left == bl;
#pragma unroll
            for (int c = 0; c <= CT; ++c) {
                float t = __shfl_up(cw[src][c], 1);
                if (merge) cw[dst][c] += t;
            }
            const int absorbed = __shfl_down(merge ? 1 : 0, 1);
            if (lane < 63 && absorbed) ok[src] = false;
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (!ok[k]) continue;
#pragma unroll
            for (int c = 0; c <= CT; ++c) atomicAdd(A + (int64_t)c * plane + o[k], cw[k][c]);
        }
        return;
    }
    if (!proj) return;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        int cx = x0 + (k & 1), cy = y0 + (k >> 1);
        if (cx < 0 || cx >= pc.W || cy < 0 || cy >= pc.H) continue;
        int64_t o = (int64_t)cy * pc.W + cx;
        if (!((double)err <= (double)Z[o] + 1.0)) continue;
        float wk = w[k];
        for (int c = 0; c < C0; ++c) atomicAdd(A + c * plane + o, D0[(int64_t)c * N + p] * wk);
        for (int c = 0; c < C1n; ++c) atomicAdd(A + (C0 + c) * plane + o, D1[(int64_t)c * N + p] * wk);
        atomicAdd(A + (int64_t)C * plane + o, 1.0f * wk);
    }
}

// models/utils.py:315  render = acc[:C]/(acc[C]+1e-7), existing = acc[C].clone()
__global__ __launch_bounds__(kBlock) void k_finalize(const float *__restrict__ accum, int B, int C, int64_t plane,
                                                      float *__restrict__ render, float *__restrict__ existing) {
    int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= plane) return;
    int b = blockIdx.y;
    const float *A = accum + (int64_t)b * (C + 1) * plane;
    float e = A[(int64_t)C * plane + i];
    float den = e + 0.0000001f;
    for (int c = 0; c < C; ++c) render[((int64_t)b * C + c) * plane + i] = A[(int64_t)c * plane + i] / den;
    existing[(int64_t)b * plane + i] = e;
}


// ---- disocclusion fill (common.py:145-248), restructured for CDNA4 ---------------------------
// The reference runs one thread per pixel and lets hole pixels march 32 rays sequentially; on a
// 64-wide machine that serialises ~32 x (ray length) dependent L2 round trips behind a handful of
// active lanes.  Here pass 1 streams every pixel once (copy / normalise / uint8 / valid-byte) and
// compacts hole pixels into a list; pass 2 gives every hole 32 lanes (16 directions x {from,to}),
// each lane marching ONE ray with 4 speculative loads in flight, then a 16-lane argmin picks the
// direction exactly like the sequential loop (shortest distance, first direction wins ties).

struct HoleWork {
    uint8_t *valid;   // [B*P] 1 = depth > 0
    int *holes;       // [B*P] flat pixel ids
    int *count;       // 1
};

// pass 1, generic operator: out <- in, valid <- depth > 0, holes <- pixels with depth <= 0
__global__ __launch_bounds__(kBlock) void k_fill_prepare(const float *__restrict__ in, const float *__restrict__ depth,
                                                          float *__restrict__ out, int C, int64_t plane, HoleWork hw) {
    int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= plane) return;
    int b = blockIdx.y;
    bool ok = (double)depth[(int64_t)b * plane + i] > 0.0;                           // common.py:160
    hw.valid[(int64_t)b * plane + i] = ok ? 1 : 0;
    for (int c = 0; c < C; ++c) out[((int64_t)b * C + c) * plane + i] = in[((int64_t)b * C + c) * plane + i];
    if (!ok) hw.holes[atomicAdd(hw.count, 1)] = (int)((int64_t)b * plane + i);
}

// pass 1, frame path: normalise the accumulators (models/utils.py:315), depth mask
// render[3]*(existing>0) (kenburns_effect.py:1039), uint8 frame (:1040), valid byte, hole list.
__global__ __launch_bounds__(kBlock) void k_finalize_frame(const float *__restrict__ accum, int64_t plane,
                                                            float *__restrict__ render, uint8_t *__restrict__ frame,
                                                            HoleWork hw) {
    // one pixel per lane (measured: the 4-pixel/float4 variant was 2x slower on gfx950: 27 us vs 13.5 us @1024^2)
    int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= plane) return;
    float e = accum[4 * plane + i];
    float den = e + 0.0000001f;
    float r0 = accum[i] / den, r1 = accum[plane + i] / den, r2 = accum[2 * plane + i] / den, r3 = accum[3 * plane + i] / den;
    float m = r3 * (e > 0.0f ? 1.0f : 0.0f);
    bool ok = (double)m > 0.0;
    hw.valid[i] = ok ? 1 : 0;
    if (render) { render[i] = r0; render[plane + i] = r1; render[2 * plane + i] = r2; render[3 * plane + i] = r3; }
    frame[i * 3 + 0] = to_u8(r0); frame[i * 3 + 1] = to_u8(r1); frame[i * 3 + 2] = to_u8(r2);
    if (!ok) hw.holes[atomicAdd(hw.count, 1)] = (int)i;
}

__constant__ float kDirX[16] = {-1, 0, 1, 1, -1, 1, 2, 2, -2, -1, 1, 2, 3, 3, 3, 3};   // common.py:168
__constant__ float kDirY[16] = {1, 1, 1, 0, 2, 2, 1, -1, 3, 3, 3, 3, 2, 1, -1, -2};    // common.py:169

// pass 2.  FRAME=false: values from `in`/`depth` (generic operator, C channels).
//          FRAME=true : values recomputed from the accumulators (bit-identical to pass 1), C = 4, also patches the
//                       uint8 frame.
template <bool FRAME>
__global__ __launch_bounds__(kBlock) void k_fill_holes(const float *__restrict__ src, const float *__restrict__ depth,
                                                        float *__restrict__ out, uint8_t *__restrict__ frame, int C,
                                                        int H, int W, HoleWork hw) {
    const int lane32 = threadIdx.x & 31;
    const int k = lane32 & 15;
    const bool to = lane32 >= 16;
    float dx = kDirX[k], dy = kDirY[k];
    float nrm = sqrtf((dx * dx) + (dy * dy));                                        // common.py:172-175
    dx /= nrm; dy /= nrm;
    const float sx = to ? dx : -dx, sy = to ? dy : -dy;                              // a - d == a + (-d) exactly
    const int slot = (blockIdx.x * kBlock + threadIdx.x) >> 5;
    const int nslots = (gridDim.x * kBlock) >> 5;
    const int count = *hw.count;
    const int64_t plane = (int64_t)H * W;
    for (int h = slot; h < count; h += nslots) {
        const int g = hw.holes[h];
        const int b = (int)(g / plane);
        const int i = (int)(g - (int64_t)b * plane);
        const int y = i / W, x = i - y * W;
        const uint8_t *V = hw.valid + (int64_t)b * plane;
        float fx = (float)x, fy = (float)y;
        int ix = 0, iy = 0;
        bool ok = false;
        for (;;) {                                                                   // common.py:186-193 / :197-204
            constexpr int kAhead = 4;    // speculative steps per round trip: positions do not depend on the loads
            int jx[kAhead], jy[kAhead], v[kAhead];
#pragma unroll
            for (int j = 0; j < kAhead; ++j) {
                fx += sx; fy += sy;
                jx[j] = (int)roundf(fx); jy[j] = (int)roundf(fy);
            }
#pragma unroll
            for (int j = 0; j < kAhead; ++j) {
                bool inb = jx[j] >= 0 && jx[j] < W && jy[j] >= 0 && jy[j] < H;
                v[j] = inb ? (int)V[(int64_t)jy[j] * W + jx[j]] : 2;
            }
            int stop = -1;
#pragma unroll
            for (int j = kAhead - 1; j >= 0; --j) if (v[j] != 0) stop = j;
            if (stop >= 0) {
#pragma unroll
                for (int j = 0; j < kAhead; ++j) if (j == stop) { ix = jx[j]; iy = jy[j]; ok = v[j] == 1; }
                break;
            }
        }
        // pair the from/to rays of one direction (lanes k and k+16)
        int ox = __shfl_xor(ix, 16), oy = __shfl_xor(iy, 16);
        bool ook = __shfl_xor((int)ok, 16) != 0;
        int fromx = to ? ox : ix, fromy = to ? oy : iy, tox = to ? ix : ox, toy = to ? iy : oy;
        float ddx = (float)(tox - fromx), ddy = (float)(toy - fromy);
        float dist = sqrtf(ddx * ddx + ddy * ddy);                                    // common.py:208
        bool cand = ok && ook && (1000000.0f > dist);                                 // fltShortest starts at 1e6, strict >
        float best = cand ? dist : INFINITY;
        int bestk = cand ? k : 16;
#pragma unroll
        for (int off = 8; off >= 1; off >>= 1) {                                     // 16-lane lexicographic (dist, k) min
            float od = __shfl_xor(best, off); int okk = __shfl_xor(bestk, off);
            if (od < best || (od == best && okk < bestk)) { best = od; bestk = okk; }
        }
        if (bestk < 16) {
            const int base = threadIdx.x & 32 & 63;  // 0 or 32 within the wave
            const int srcl = base + bestk;            // the "from" lane of the winning direction
            int wfx = __shfl(fromx, srcl), wfy = __shfl(fromy, srcl), wtx = __shfl(tox, srcl), wty = __shfl(toy, srcl);
            const int64_t of = (int64_t)wfy * W + wfx, ot = (int64_t)wty * W + wtx;
            float dfrom, dto;
            if (FRAME) {
                float ef = src[4 * plane + of], et = src[4 * plane + ot];
                dfrom = (src[3 * plane + of] / (ef + 0.0000001f)) * (ef > 0.0f ? 1.0f : 0.0f);
                dto = (src[3 * plane + ot] / (et + 0.0000001f)) * (et > 0.0f ? 1.0f : 0.0f);
            } else {
                dfrom = depth[(int64_t)b * plane + of]; dto = depth[(int64_t)b * plane + ot];
            }
            const int64_t so = dfrom < dto ? ot : of;                                 // common.py:214-217
            if (FRAME) {
                if (lane32 < 4) {
                    float es = src[4 * plane + so];
                    float v = src[(int64_t)lane32 * plane + so] / (es + 0.0000001f);
                    if (out) out[(int64_t)lane32 * plane + i] = v;
                    if (lane32 < 3) frame[(int64_t)i * 3 + lane32] = to_u8(v);
                }
            } else {
                for (int c = lane32; c < C; c += 32)
                    out[((int64_t)b * C + c) * plane + i] = src[((int64_t)b * C + c) * plane + so];
            }
        }
    }
}

// spatial_filter 'laplacian' (models/utils.py:12-24): replicate pad + asymmetric 3x3
__device__ __forceinline__ float laplacian_at(const float *__restrict__ I, int x, int y, int H, int W, float scale_div) {
    int ym = y > 0 ? y - 1 : 0, yp = y < H - 1 ? y + 1 : H - 1;
    int xm = x > 0 ? x - 1 : 0, xp = x < W - 1 ? x + 1 : W - 1;
    float acc = 0.0f;
    if (scale_div != 0.0f) {
        acc += -1.0f * (I[(int64_t)ym * W + x] / scale_div);
        acc += -1.0f * (I[(int64_t)ym * W + xp] / scale_div);
        acc += -1.0f * (I[(int64_t)y * W + xm] / scale_div);
        acc += 4.0f * (I[(int64_t)y * W + x] / scale_div);
        acc += -1.0f * (I[(int64_t)yp * W + xm] / scale_div);
    } else {
        acc += -1.0f * I[(int64_t)ym * W + x];
        acc += -1.0f * I[(int64_t)ym * W + xp];
        acc += -1.0f * I[(int64_t)y * W + xm];
        acc += 4.0f * I[(int64_t)y * W + x];
        acc += -1.0f * I[(int64_t)yp * W + xm];
    }
    return acc;
}

__global__ __launch_bounds__(kBlock) void k_laplacian(const float *__restrict__ in, float *__restrict__ out, int H,
                                                       int W) {
    int x = blockIdx.x * 64 + (threadIdx.x & 63);
    int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= W || y >= H) return;
    const float *I = in + (int64_t)blockIdx.z * H * W;
    out[((int64_t)blockIdx.z * H + y) * W + x] = laplacian_at(I, x, y, H, W, 0.0f);
}


// spatial_filter 'median-3' / 'median-5' (models/utils.py:26-36): reflect pad K/2, K x K window, torch.median = lower median (5th of 9,
// 13th of 25)
template <int K>
__global__ __launch_bounds__(kBlock) void k_median(const float *__restrict__ in, float *__restrict__ out, int H, int W) {
    constexpr int R = K / 2, NV = K * K;
    int x = blockIdx.x * 64 + (threadIdx.x & 63);
    int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= W || y >= H) return;
    const float *I = in + (int64_t)blockIdx.z * H * W;
    float v[NV];
#pragma unroll
    for (int dy = -R; dy <= R; ++dy) {
        int yy = y + dy; yy = yy < 0 ? -yy : (yy >= H ? 2 * H - 2 - yy : yy);
#pragma unroll
        for (int dx = -R; dx <= R; ++dx) {
            int xx = x + dx; xx = xx < 0 ? -xx : (xx >= W ? 2 * W - 2 - xx : xx);
            v[(dy + R) * K + (dx + R)] = I[(int64_t)yy * W + xx];
        }
    }
    // rank of each element (ties broken by index) -> the element of rank (NV - 1) / 2 is the lower median
    // a NaN anywhere in the window gives NaN, as torch.median does (every comparison with a NaN is false: the ranks would all tie)
    float med = v[0];
    bool any_nan = false;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        int rank = 0;
        any_nan |= v[i] != v[i];
#pragma unroll
        for (int j = 0; j < NV; ++j) rank += (v[j] < v[i] || (v[j] == v[i] && j < i)) ? 1 : 0;
        if (rank == (NV - 1) / 2) med = v[i];
    }
    if (any_nan) med = __int_as_float(0x7fc00000);
    out[((int64_t)blockIdx.z * H + y) * W + x] = med;
}

// depth_to_points (models/utils.py:43-50)
__global__ __launch_bounds__(kBlock) void k_depth_to_points(const float *__restrict__ depth, float *__restrict__ pts,
                                                             int H, int W, float invf, float x_start, float y_start) {
    int x = blockIdx.x * 64 + (threadIdx.x & 63);
    int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= W || y >= H) return;
    const int64_t plane = (int64_t)H * W, o = (int64_t)y * W + x;
    int b = blockIdx.z;
    float d = depth[b * plane + o];
    float hx = (x_start + (float)x) * invf, vy = (y_start + (float)y) * invf;
    float *P = pts + (int64_t)b * 3 * plane;
    P[o] = d * hx; P[plane + o] = d * vy; P[2 * plane + o] = d;
}

// kenburns_effect.py:928-933 fused into one pass over the disparity map
__global__ __launch_bounds__(kBlock) void k_disparity_to_points(const float *__restrict__ disp, const float *__restrict__ disp_max_p, int H,
                                                                 int W, float fb, float eps, float invf, float x_start,
                                                                 float y_start, float *__restrict__ depth,
                                                                 float *__restrict__ valid, float *__restrict__ pts,
                                                                 float *__restrict__ unaltered) {
    int x = blockIdx.x * 64 + (threadIdx.x & 63);
    int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= W || y >= H) return;
    const int64_t plane = (int64_t)H * W, o = (int64_t)y * W + x;
    float d = (1.0f / (disp[o] + eps)) * fb;  // float / Tensor == reciprocal()*float in torch
    float lap = laplacian_at(disp, x, y, H, W, disp_max_p[0]);
    float v = fabsf(lap) < 0.03f ? 1.0f : 0.0f;
    float hx = (x_start + (float)x) * invf, vy = (y_start + (float)y) * invf;
    depth[o] = d; valid[o] = v;
    float dv = d * v;
    pts[o] = dv * hx; pts[plane + o] = dv * vy; pts[2 * plane + o] = dv;
    unaltered[o] = d * hx; unaltered[plane + o] = d * vy; unaltered[2 * plane + o] = d;
}

__global__ __launch_bounds__(kBlock) void k_process_shift(const float *__restrict__ pts, float *__restrict__ out,
                                                           int64_t N, int64_t total, Shift s) {
    int64_t idx = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (idx >= total) return;
    int64_t b = idx / N, p = idx - b * N;
    float x, y, z;
    load_point<true>(pts + b * 3 * N, N, p, s, x, y, z);
    float *O = out + b * 3 * N;
    O[p] = x; O[N + p] = y; O[2 * N + p] = z;
}


inline dim3 grid2d(int W, int H, int B, int bx, int by) { return dim3(csm::cdiv(W, bx), csm::cdiv(H, by), B); }

}  // namespace

// ------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------
extern "C" int csm_pointrender_update_zee(const float *pts, int B, int64_t N, int H, int W, double focal,
                                          double baseline, float *zee, void *stream) {
    CSM_REQUIRE(zee && (pts || N == 0) && B > 0 && N >= 0 && H > 0 && W > 0);
    if (N == 0) return CSM_OK;
    int64_t total = (int64_t)B * N;
    k_update_zee<false><<<csm::cdiv(total, kBlock), kBlock, 0, (hipStream_t)stream>>>(
        pts, N, total, make_proj(H, W, focal, baseline), Shift{0, 0, 0}, zee);
    return csm::check_launch("k_update_zee");
}

extern "C" int csm_pointrender_degrid(const float *zee_in, float *zee_out, int B, int H, int W, void *stream) {
    CSM_REQUIRE(zee_in && zee_out && zee_in != zee_out && B > 0 && H > 0 && W > 0);
    k_degrid<<<grid2d(W, H, B, 64, 4), kBlock, 0, (hipStream_t)stream>>>(zee_in, zee_out, B, H, W);
    return csm::check_launch("k_degrid");
}

static int launch_update_output(const float *pts, const float *d0, int C0, const float *d1, int C1n, int B,
                                int64_t N, int H, int W, double focal, double baseline, bool shift, Shift s,
                                const float *zee, float *accum, hipStream_t st) {
    int64_t total = (int64_t)B * N;
    if (total == 0) return CSM_OK;
    dim3 g(csm::cdiv(total, kBlock));
    ProjConst pc = make_proj(H, W, focal, baseline);
    int C = C0 + C1n;
#define CSM_LAUNCH_UO(SH, CT) k_update_output<SH, CT><<<g, kBlock, 0, st>>>(pts, d0, C0, d1, C1n, N, total, pc, s, zee, accum)
    if (shift) { if (C == 4) CSM_LAUNCH_UO(true, 4); else if (C == 3) CSM_LAUNCH_UO(true, 3); else CSM_LAUNCH_UO(true, 0); }
    else { if (C == 4) CSM_LAUNCH_UO(false, 4); else if (C == 3) CSM_LAUNCH_UO(false, 3); else CSM_LAUNCH_UO(false, 0); }
#undef CSM_LAUNCH_UO
    return csm::check_launch("k_update_output");
}

extern "C" int csm_pointrender_update_output(const float *pts, const float *data, const float *zee, int B, int C,
                                             int64_t N, int H, int W, double focal, double baseline, float *accum,
                                             void *stream) {
    CSM_REQUIRE(zee && accum && ((pts && data) || N == 0) && B > 0 && C > 0 && N >= 0 && H > 0 && W > 0);
    return launch_update_output(pts, data, C, nullptr, 0, B, N, H, W, focal, baseline, false, Shift{0, 0, 0}, zee,
                                accum, (hipStream_t)stream);
}

extern "C" int csm_render_pointcloud(const float *pts, const float *data, int B, int C, int64_t N, int W, int H,
                                     double focal, double baseline, float *zee_scratch, float *accum_scratch,
                                     float *render, float *existing, void *stream) {
    CSM_REQUIRE(zee_scratch && accum_scratch && render && existing && ((pts && data) || N == 0));
    CSM_REQUIRE(B > 0 && C > 0 && N >= 0 && H > 0 && W > 0);
    hipStream_t st = (hipStream_t)stream;
    const int64_t plane = (int64_t)H * W;
    float *zeeA = zee_scratch, *zeeB = zee_scratch + B * plane;
    k_fill<<<1024, kBlock, 0, st>>>(zeeA, B * plane, 1000000.0f, accum_scratch, (int64_t)B * (C + 1) * plane, 0.0f);
    int rc = csm::check_launch("k_fill"); if (rc) return rc;
    if (N > 0) {
        int64_t total = (int64_t)B * N;
        k_update_zee<false><<<csm::cdiv(total, kBlock), kBlock, 0, st>>>(pts, N, total, make_proj(H, W, focal, baseline),
                                                                          Shift{0, 0, 0}, zeeA);
        rc = csm::check_launch("k_update_zee"); if (rc) return rc;
    }
    k_degrid<<<grid2d(W, H, B, 64, 4), kBlock, 0, st>>>(zeeA, zeeB, B, H, W);
    rc = csm::check_launch("k_degrid"); if (rc) return rc;
    rc = launch_update_output(pts, data, C, nullptr, 0, B, N, H, W, focal, baseline, false, Shift{0, 0, 0}, zeeB,
                              accum_scratch, st);
    if (rc) return rc;
    k_finalize<<<dim3(csm::cdiv(plane, kBlock), B), kBlock, 0, st>>>(accum_scratch, B, C, plane, render, existing);
    return csm::check_launch("k_finalize");
}

static HoleWork carve_holework(void *scratch, int64_t npix) {
    HoleWork hw;
    char *p = (char *)scratch;
    hw.count = (int *)p;                       // 16 bytes reserved
    hw.holes = (int *)(p + 16);
    hw.valid = (uint8_t *)(p + 16 + 4 * npix);
    return hw;
}

extern "C" size_t csm_fill_disocclusion_scratch_bytes(int B, int H, int W) {
    int64_t n = (int64_t)B * H * W;
    return (size_t)(16 + 4 * n + ((n + 15) / 16) * 16);
}

extern "C" int csm_fill_disocclusion(const float *in, const float *depth, float *out, int B, int C, int H, int W,
                                     void *scratch, void *stream) {
    CSM_REQUIRE(in && depth && out && scratch && in != out && B > 0 && C > 0 && H > 0 && W > 0);
    hipStream_t st = (hipStream_t)stream;
    const int64_t plane = (int64_t)H * W;
    HoleWork hw = carve_holework(scratch, (int64_t)B * plane);
    CSM_HIP(hipMemsetAsync(hw.count, 0, 16, st));
    k_fill_prepare<<<dim3(csm::cdiv(plane, kBlock), B), kBlock, 0, st>>>(in, depth, out, C, plane, hw);
    int rc = csm::check_launch("k_fill_prepare"); if (rc) return rc;
    k_fill_holes<false><<<1024, kBlock, 0, st>>>(in, depth, out, nullptr, C, H, W, hw);
    return csm::check_launch("k_fill_holes");
}

extern "C" int csm_spatial_filter_laplacian(const float *in, float *out, int BC, int H, int W, void *stream) {
    CSM_REQUIRE(in && out && BC > 0 && H > 0 && W > 0);
    k_laplacian<<<grid2d(W, H, BC, 64, 4), kBlock, 0, (hipStream_t)stream>>>(in, out, H, W);
    return csm::check_launch("k_laplacian");
}

extern "C" int csm_spatial_filter_median5(const float *in, float *out, int BC, int H, int W, void *stream) {
    CSM_REQUIRE(in && out && in != out && BC > 0 && H > 2 && W > 2);
    k_median<5><<<grid2d(W, H, BC, 64, 4), kBlock, 0, (hipStream_t)stream>>>(in, out, H, W);
    return csm::check_launch("k_median5");
}

extern "C" int csm_spatial_filter_median3(const float *in, float *out, int BC, int H, int W, void *stream) {
    CSM_REQUIRE(in && out && in != out && BC > 0 && H > 1 && W > 1);
    k_median<3><<<grid2d(W, H, BC, 64, 4), kBlock, 0, (hipStream_t)stream>>>(in, out, H, W);
    return csm::check_launch("k_median3");
}

extern "C" int csm_depth_to_points(const float *depth, float *pts, int B, int H, int W, double focal, void *stream) {
    CSM_REQUIRE(depth && pts && B > 0 && H > 0 && W > 0 && focal != 0.0);
    k_depth_to_points<<<grid2d(W, H, B, 64, 4), kBlock, 0, (hipStream_t)stream>>>(
        depth, pts, H, W, (float)(1.0 / focal), (float)(-0.5 * W + 0.5), (float)(-0.5 * H + 0.5));
    return csm::check_launch("k_depth_to_points");
}

extern "C" int csm_disparity_to_points(const float *disp, const float *disp_max, int H, int W, double focal, double baseline,
                                       float eps, float *depth, float *valid, float *pts, float *unaltered, void *stream) {
    CSM_REQUIRE(disp && disp_max && depth && valid && pts && unaltered && H > 0 && W > 0 && focal != 0.0);
    k_disparity_to_points<<<grid2d(W, H, 1, 64, 4), kBlock, 0, (hipStream_t)stream>>>(
        disp, disp_max, H, W, (float)(focal * baseline), eps, (float)(1.0 / focal), (float)(-0.5 * W + 0.5),
        (float)(-0.5 * H + 0.5), depth, valid, pts, unaltered);
    return csm::check_launch("k_disparity_to_points");
}

extern "C" int csm_process_shift(const float *pts, float *out, int B, int64_t N, float sx, float sy, float sz,
                                 void *stream) {
    CSM_REQUIRE(((pts && out) || N == 0) && B > 0 && N >= 0);
    int64_t total = (int64_t)B * N;
    if (total == 0) return CSM_OK;
    k_process_shift<<<csm::cdiv(total, kBlock), kBlock, 0, (hipStream_t)stream>>>(pts, out, N, total, Shift{sx, sy, sz});
    return csm::check_launch("k_process_shift");
}

// scratch layout (floats): zeeA[P] | zeeB[P] | accum[5P] | hole work (16 B + 4P + P bytes)
extern "C" size_t csm_warp_frame_scratch_floats(int H, int W) {
    size_t P = (size_t)H * (size_t)W;
    return 7 * P + (csm_fill_disocclusion_scratch_bytes(1, H, W) + 3) / 4 + 4;
}

extern "C" int csm_warp_frame(const float *pts, const float *rgb, const float *depth, int64_t N, int H, int W,
                              double focal, double baseline, float sx, float sy, float sz, float *scratch,
                              float *render_filled, uint8_t *frame_u8, void *stream) {
    CSM_REQUIRE(scratch && frame_u8 && N >= 0 && H > 0 && W > 0);
    CSM_REQUIRE(N == 0 || (pts && rgb && depth));
    hipStream_t st = (hipStream_t)stream;
    const int64_t plane = (int64_t)H * W;
    float *zeeA = scratch, *zeeB = scratch + plane, *accum = scratch + 2 * plane;
    HoleWork hw = carve_holework(scratch + 7 * plane, plane);
    Shift s{sx, sy, sz};
    ProjConst pc = make_proj(H, W, focal, baseline);
    // one fill covers zee (1e6) and accum (0); the hole counter sits right behind accum and is zeroed with it
    k_fill<<<2048, kBlock, 0, st>>>(zeeA, plane, 1000000.0f, accum, 5 * plane + 4, 0.0f);
    int rc = csm::check_launch("k_fill"); if (rc) return rc;
    if (N > 0) {
        k_update_zee<true><<<csm::cdiv(N, kBlock), kBlock, 0, st>>>(pts, N, N, pc, s, zeeA);
        rc = csm::check_launch("k_update_zee"); if (rc) return rc;
    }
    k_degrid<<<grid2d(W, H, 1, 64, 4), kBlock, 0, st>>>(zeeA, zeeB, 1, H, W);
    rc = csm::check_launch("k_degrid"); if (rc) return rc;
    rc = launch_update_output(pts, rgb, 3, depth, 1, 1, N, H, W, focal, baseline, true, s, zeeB, accum, st);
    if (rc) return rc;
    k_finalize_frame<<<csm::cdiv(plane, kBlock), kBlock, 0, st>>>(accum, plane, render_filled, frame_u8, hw);
    rc = csm::check_launch("k_finalize_frame"); if (rc) return rc;
    k_fill_holes<true><<<1024, kBlock, 0, st>>>(accum, nullptr, render_filled, frame_u8, 4, H, W, hw);
    return csm::check_launch("k_fill_holes");
}
