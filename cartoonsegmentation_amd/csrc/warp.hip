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
#include "csm_common.h"

namespace {

constexpr int kBlock = 256;

struct ProjConst {
    double focal_baseline;  // focal*baseline, folded in double like the literal in the CUDA text
    double half_w, half_h;  // 0.5*W, 0.5*H
    float focal_f;          // make_float3(0,0,focal).z
    int W, H;
};
struct Shift { float x, y, z; };

template <bool SHIFT>
__device__ __forceinline__ void load_point(const float *__restrict__ P, int64_t N, int64_t p, Shift s,
                                           float &x, float &y, float &z) {
    x = P[p]; y = P[N + p]; z = P[2 * N + p];
    if (SHIFT) {  // common.py:78-81
        float r = z / (z + 0.0000001f);
        x = x * r + s.x; y = y * r + s.y; z = z + s.z;
    }
}

// models/utils.py:76-99  (mixed fp32/fp64 exactly as the untyped CUDA literals evaluate)
__device__ __forceinline__ bool project(float x, float y, float z, const ProjConst &pc, float &fx, float &fy,
                                        float &err) {
    if ((double)z < 0.001) return false;
    float lvx = 0.0f - x, lvy = 0.0f - y, lvz = 0.0f - z;
    float ax = 0.0f - x, ay = 0.0f - y, az = pc.focal_f - z;
    float num = ax * 0.0f + ay * 0.0f + az * 1.0f;
    float den = lvx * 0.0f + lvy * 0.0f + lvz * 1.0f;
    float dist = num / den;
    if ((double)fabsf(den) < 0.001) return false;
    float ix = x + dist * lvx;
    float iy = y + dist * lvy;
    fx = (float)(((double)ix + pc.half_w) - 0.5);
    fy = (float)(((double)iy + pc.half_h) - 0.5);
    err = (float)(1000000.0 - (pc.focal_baseline / ((double)z + 0.0000001)));
    return true;
}

__device__ __forceinline__ void corner_weights(float fx, float fy, int &x0, int &y0, float w[4]) {
    x0 = (int)floorf(fx); y0 = (int)floorf(fy);
    float x1 = (float)(x0 + 1), y1 = (float)(y0 + 1), xf = (float)x0, yf = (float)y0;
    w[0] = (x1 - fx) * (y1 - fy);  // NW
    w[1] = (fx - xf) * (y1 - fy);  // NE
    w[2] = (x1 - fx) * (fy - yf);  // SW
    w[3] = (fx - xf) * (fy - yf);  // SE
}

// float min through native integer atomics (replaces the CAS loop of utils/cupy_utils.py:21-29).
// Non-negative floats order like signed ints; negative floats order inversely like unsigned ints.
__device__ __forceinline__ void atomic_min_f32(float *addr, float v) {
    if (v >= 0.0f) atomicMin(reinterpret_cast<int *>(addr), __float_as_int(v));
    else atomicMax(reinterpret_cast<unsigned int *>(addr), __float_as_uint(v));
}

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
    if (idx >= total) return;
    int64_t b = idx / N, p = idx - b * N;
    float x, y, z;
    load_point<SHIFT>(pts + b * 3 * N, N, p, s, x, y, z);
    float fx, fy, err, w[4];
    if (!project(x, y, z, pc, fx, fy, err)) return;
    int x0, y0;
    corner_weights(fx, fy, x0, y0, w);
    const int C = CT > 0 ? CT : (C0 + C1n);
    const int64_t plane = (int64_t)pc.H * pc.W;
    const float *D0 = d0 + b * C0 * N;
    const float *D1 = d1 ? d1 + b * C1n * N : nullptr;
    float *A = accum + b * (C + 1) * plane;
    const float *Z = zee + b * plane;
    float v[CT > 0 ? CT : 1];
    if (CT > 0) {
#pragma unroll
        for (int c = 0; c < CT; ++c) v[c] = c < C0 ? D0[(int64_t)c * N + p] : D1[(int64_t)(c - C0) * N + p];
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        int cx = x0 + (k & 1), cy = y0 + (k >> 1);
        if (cx < 0 || cx >= pc.W || cy < 0 || cy >= pc.H) continue;
        int64_t o = (int64_t)cy * pc.W + cx;
        if (!((double)err <= (double)Z[o] + 1.0)) continue;
        float wk = w[k];
        if (CT > 0) {
#pragma unroll
            for (int c = 0; c < CT; ++c) atomicAdd(A + c * plane + o, v[c] * wk);
        } else {
            for (int c = 0; c < C0; ++c) atomicAdd(A + c * plane + o, D0[(int64_t)c * N + p] * wk);
            for (int c = 0; c < C1n; ++c) atomicAdd(A + (C0 + c) * plane + o, D1[(int64_t)c * N + p] * wk);
        }
        atomicAdd(A + (int64_t)C * plane + o, 1.0f * wk);
    }
}

// models/utils.py:315  render = acc[:C]/(acc[C]+1e-7), existing = acc[C].clone()
// MASK: also emit kenburns_effect.py:1039's  render[3]*(existing>0)  (C==4 only)
template <bool MASK>
__global__ __launch_bounds__(kBlock) void k_finalize(const float *__restrict__ accum, int B, int C, int64_t plane,
                                                      float *__restrict__ render, float *__restrict__ existing,
                                                      float *__restrict__ mask) {
    int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= plane) return;
    int b = blockIdx.y;
    const float *A = accum + (int64_t)b * (C + 1) * plane;
    float e = A[(int64_t)C * plane + i];
    float den = e + 0.0000001f;
    float last = 0.0f;
    for (int c = 0; c < C; ++c) {
        last = A[(int64_t)c * plane + i] / den;
        render[((int64_t)b * C + c) * plane + i] = last;
    }
    if (existing) existing[(int64_t)b * plane + i] = e;
    if (MASK) mask[(int64_t)b * plane + i] = last * (e > 0.0f ? 1.0f : 0.0f);
}

struct Dirs { float x[16], y[16]; };

// kernel_discfill_updateOutput  (common.py:149-245).  U8: also write the uint8 HWC frame
// (kenburns_effect.py:1040) from channels 0..2.
template <bool U8>
__global__ __launch_bounds__(kBlock) void k_discfill(const float *__restrict__ in, const float *__restrict__ depth,
                                                      float *__restrict__ out, uint8_t *__restrict__ frame, int C,
                                                      int H, int W, Dirs dirs) {
    int x = blockIdx.x * 32 + (threadIdx.x & 31);
    int y = blockIdx.y * 8 + (threadIdx.x >> 5);
    int b = blockIdx.z;
    if (x >= W || y >= H) return;
    const int64_t plane = (int64_t)H * W;
    const float *D = depth + (int64_t)b * plane;
    const float *I = in + (int64_t)b * C * plane;
    int srcx = x, srcy = y;
    if (!((double)D[(int64_t)y * W + x] > 0.0)) {
        float shortest = 1000000.0f;
        int fillx = -1, filly = -1;
        for (int k = 0; k < 16; ++k) {
            const float dx = dirs.x[k], dy = dirs.y[k];
            float ffx = (float)x, ffy = (float)y; int ifx = 0, ify = 0;
            for (;;) {
                ffx -= dx; ifx = (int)roundf(ffx);
                ffy -= dy; ify = (int)roundf(ffy);
                if (ifx < 0 || ifx >= W) break;
                if (ify < 0 || ify >= H) break;
                if ((double)D[(int64_t)ify * W + ifx] > 0.0) break;
            }
            if (ifx < 0 || ifx >= W || ify < 0 || ify >= H) continue;
            float ftx = (float)x, fty = (float)y; int itx = 0, ity = 0;
            for (;;) {
                ftx += dx; itx = (int)roundf(ftx);
                fty += dy; ity = (int)roundf(fty);
                if (itx < 0 || itx >= W) break;
                if (ity < 0 || ity >= H) break;
                if ((double)D[(int64_t)ity * W + itx] > 0.0) break;
            }
            if (itx < 0 || itx >= W || ity < 0 || ity >= H) continue;
            float ddx = (float)(itx - ifx), ddy = (float)(ity - ify);
            float dist = sqrtf(ddx * ddx + ddy * ddy);
            if (shortest > dist) {
                fillx = ifx; filly = ify;
                if (D[(int64_t)ify * W + ifx] < D[(int64_t)ity * W + itx]) { fillx = itx; filly = ity; }
                shortest = dist;
            }
        }
        if (fillx != -1 && filly != -1) { srcx = fillx; srcy = filly; }
    }
    const int64_t so = (int64_t)srcy * W + srcx, o = (int64_t)y * W + x;
    for (int c = 0; c < C; ++c) {
        float v = I[(int64_t)c * plane + so];
        if (out) out[((int64_t)b * C + c) * plane + o] = v;
        if (U8 && c < 3) {
            float u = v * 255.0f;
            u = u < 0.0f ? 0.0f : (u > 255.0f ? 255.0f : u);
            frame[o * 3 + c] = (uint8_t)u;
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
__global__ __launch_bounds__(kBlock) void k_disparity_to_points(const float *__restrict__ disp, float disp_max, int H,
                                                                 int W, float fb, float invf, float x_start,
                                                                 float y_start, float *__restrict__ depth,
                                                                 float *__restrict__ valid, float *__restrict__ pts,
                                                                 float *__restrict__ unaltered) {
    int x = blockIdx.x * 64 + (threadIdx.x & 63);
    int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= W || y >= H) return;
    const int64_t plane = (int64_t)H * W, o = (int64_t)y * W + x;
    float d = (1.0f / (disp[o] + 0.00001f)) * fb;  // float / Tensor == reciprocal()*float in torch
    float lap = laplacian_at(disp, x, y, H, W, disp_max);
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

ProjConst make_proj(int H, int W, double focal, double baseline) {
    ProjConst pc;
    pc.focal_baseline = focal * baseline;
    pc.half_w = 0.5 * W; pc.half_h = 0.5 * H;
    pc.focal_f = (float)focal; pc.W = W; pc.H = H;
    return pc;
}

Dirs make_dirs() {  // common.py:168-176 (host IEEE fp32 == device IEEE fp32)
    const float dx[16] = {-1, 0, 1, 1, -1, 1, 2, 2, -2, -1, 1, 2, 3, 3, 3, 3};
    const float dy[16] = {1, 1, 1, 0, 2, 2, 1, -1, 3, 3, 3, 3, 2, 1, -1, -2};
    Dirs d;
    for (int k = 0; k < 16; ++k) {
        volatile float n = sqrtf((dx[k] * dx[k]) + (dy[k] * dy[k]));
        volatile float qx = dx[k] / n, qy = dy[k] / n;
        d.x[k] = qx; d.y[k] = qy;
    }
    return d;
}

inline dim3 grid2d(int W, int H, int B, int bx, int by) { return dim3(csm::cdiv(W, bx), csm::cdiv(H, by), B); }

}  // namespace

// ------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------
extern "C" int csm_pointrender_update_zee(const float *pts, int B, int64_t N, int H, int W, double focal,
                                          double baseline, float *zee, void *stream) {
    CSM_REQUIRE(pts && zee && B > 0 && N >= 0 && H > 0 && W > 0);
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
    CSM_REQUIRE(pts && data && zee && accum && B > 0 && C > 0 && N >= 0 && H > 0 && W > 0);
    return launch_update_output(pts, data, C, nullptr, 0, B, N, H, W, focal, baseline, false, Shift{0, 0, 0}, zee,
                                accum, (hipStream_t)stream);
}

extern "C" int csm_render_pointcloud(const float *pts, const float *data, int B, int C, int64_t N, int W, int H,
                                     double focal, double baseline, float *zee_scratch, float *accum_scratch,
                                     float *render, float *existing, void *stream) {
    CSM_REQUIRE(pts && data && zee_scratch && accum_scratch && render && existing);
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
    k_finalize<false><<<dim3(csm::cdiv(plane, kBlock), B), kBlock, 0, st>>>(accum_scratch, B, C, plane, render,
                                                                             existing, nullptr);
    return csm::check_launch("k_finalize");
}

extern "C" int csm_fill_disocclusion(const float *in, const float *depth, float *out, int B, int C, int H, int W,
                                     void *stream) {
    CSM_REQUIRE(in && depth && out && in != out && B > 0 && C > 0 && H > 0 && W > 0);
    static const Dirs dirs = make_dirs();
    k_discfill<false><<<grid2d(W, H, B, 32, 8), kBlock, 0, (hipStream_t)stream>>>(in, depth, out, nullptr, C, H, W, dirs);
    return csm::check_launch("k_discfill");
}

extern "C" int csm_spatial_filter_laplacian(const float *in, float *out, int BC, int H, int W, void *stream) {
    CSM_REQUIRE(in && out && BC > 0 && H > 0 && W > 0);
    k_laplacian<<<grid2d(W, H, BC, 64, 4), kBlock, 0, (hipStream_t)stream>>>(in, out, H, W);
    return csm::check_launch("k_laplacian");
}

extern "C" int csm_depth_to_points(const float *depth, float *pts, int B, int H, int W, double focal, void *stream) {
    CSM_REQUIRE(depth && pts && B > 0 && H > 0 && W > 0 && focal != 0.0);
    k_depth_to_points<<<grid2d(W, H, B, 64, 4), kBlock, 0, (hipStream_t)stream>>>(
        depth, pts, H, W, (float)(1.0 / focal), (float)(-0.5 * W + 0.5), (float)(-0.5 * H + 0.5));
    return csm::check_launch("k_depth_to_points");
}

extern "C" int csm_disparity_to_points(const float *disp, float disp_max, int H, int W, double focal, double baseline,
                                       float *depth, float *valid, float *pts, float *unaltered, void *stream) {
    CSM_REQUIRE(disp && depth && valid && pts && unaltered && H > 0 && W > 0 && focal != 0.0);
    k_disparity_to_points<<<grid2d(W, H, 1, 64, 4), kBlock, 0, (hipStream_t)stream>>>(
        disp, disp_max, H, W, (float)(focal * baseline), (float)(1.0 / focal), (float)(-0.5 * W + 0.5),
        (float)(-0.5 * H + 0.5), depth, valid, pts, unaltered);
    return csm::check_launch("k_disparity_to_points");
}

extern "C" int csm_process_shift(const float *pts, float *out, int B, int64_t N, float sx, float sy, float sz,
                                 void *stream) {
    CSM_REQUIRE(pts && out && B > 0 && N >= 0);
    int64_t total = (int64_t)B * N;
    if (total == 0) return CSM_OK;
    k_process_shift<<<csm::cdiv(total, kBlock), kBlock, 0, (hipStream_t)stream>>>(pts, out, N, total, Shift{sx, sy, sz});
    return csm::check_launch("k_process_shift");
}

extern "C" size_t csm_warp_frame_scratch_floats(int H, int W) { return (size_t)12 * (size_t)H * (size_t)W; }

extern "C" int csm_warp_frame(const float *pts, const float *rgb, const float *depth, int64_t N, int H, int W,
                              double focal, double baseline, float sx, float sy, float sz, float *scratch,
                              float *render_filled, uint8_t *frame_u8, void *stream) {
    CSM_REQUIRE(pts && rgb && depth && scratch && frame_u8 && N >= 0 && H > 0 && W > 0);
    hipStream_t st = (hipStream_t)stream;
    static const Dirs dirs = make_dirs();
    const int64_t plane = (int64_t)H * W;
    float *zeeA = scratch, *zeeB = scratch + plane, *accum = scratch + 2 * plane;  // zeeA|zeeB|accum[5]|render[4]|mask
    float *render = scratch + 7 * plane, *mask = scratch + 11 * plane;
    Shift s{sx, sy, sz};
    ProjConst pc = make_proj(H, W, focal, baseline);
    k_fill<<<1024, kBlock, 0, st>>>(zeeA, plane, 1000000.0f, accum, 5 * plane, 0.0f);
    int rc = csm::check_launch("k_fill"); if (rc) return rc;
    if (N > 0) {
        k_update_zee<true><<<csm::cdiv(N, kBlock), kBlock, 0, st>>>(pts, N, N, pc, s, zeeA);
        rc = csm::check_launch("k_update_zee"); if (rc) return rc;
    }
    k_degrid<<<grid2d(W, H, 1, 64, 4), kBlock, 0, st>>>(zeeA, zeeB, 1, H, W);
    rc = csm::check_launch("k_degrid"); if (rc) return rc;
    rc = launch_update_output(pts, rgb, 3, depth, 1, 1, N, H, W, focal, baseline, true, s, zeeB, accum, st);
    if (rc) return rc;
    k_finalize<true><<<dim3(csm::cdiv(plane, kBlock), 1), kBlock, 0, st>>>(accum, 1, 4, plane, render, nullptr, mask);
    rc = csm::check_launch("k_finalize"); if (rc) return rc;
    k_discfill<true><<<grid2d(W, H, 1, 32, 8), kBlock, 0, st>>>(render, mask, render_filled, frame_u8, 4, H, W, dirs);
    return csm::check_launch("k_discfill");
}
