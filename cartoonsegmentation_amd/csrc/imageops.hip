// imageops.hip -- uint8 image plumbing around the nets, restated from OpenCV 4.10 semantics [EXT: cv2 is not under
// /root/reference and not installed here => parity unpinned for these resamplers]:
//   * LeReS input : scaledown_maxsize (cv2 INTER_LINEAR u8) -> /255 -> BGR->RGB -> ToTensor+Normalize
//                   (kenburns_effect.py:563-571, depth_modules/leres/leres/depthmap.py:16-38)
//   * LeReS output: min-max -> uint16 -> convertScaleAbs -> bitwise_not (depth_modules/leres/__init__.py:121-145),
//                   then cv2.resize(INTER_AREA) back to the frame size and zero-fix (kenburns_effect.py:572-578)
//   * frame tail  : cv2.getRectSubPix + cv2.resize(INTER_LINEAR) (kenburns_effect.py:1069-1070)
#include "csm_common.h"

namespace {

__device__ __forceinline__ void cv_src(int d, int in_size, double scale, int &i0, int &i1, float &f) {
    float fx = (float)((d + 0.5) * scale - 0.5);
    int sx = (int)floorf(fx);
    fx -= (float)sx;
    if (sx < 0) { fx = 0.0f; sx = 0; }
    if (sx >= in_size - 1) { fx = 0.0f; sx = in_size - 1; }
    i0 = sx; i1 = min(sx + 1, in_size - 1); f = fx;
}
// INTER_AREA when up-sampling: linear taps with "area" fractions (resize.cpp, area_mode branch)
__device__ __forceinline__ void cv_src_area(int d, int in_size, double scale, int &i0, int &i1, float &f) {
    int sx = (int)floor(d * scale);
    float fx = (float)((d + 1) - (sx + 1) * (1.0 / scale));
    fx = fx <= 0.0f ? 0.0f : fx - floorf(fx);
    if (sx < 0) { fx = 0.0f; sx = 0; }
    if (sx >= in_size - 1) { fx = 0.0f; sx = in_size - 1; }
    i0 = sx; i1 = min(sx + 1, in_size - 1); f = fx;
}
__device__ __forceinline__ int cv_lin_u8(int p00, int p01, int p10, int p11, float fx, float fy) {
    const int a0 = (int)rintf((1.0f - fx) * 2048.0f), a1 = (int)rintf(fx * 2048.0f);
    const int b0 = (int)rintf((1.0f - fy) * 2048.0f), b1 = (int)rintf(fy * 2048.0f);
    int r0 = p00 * a0 + p01 * a1, r1 = p10 * a0 + p11 * a1;
    int q = (((b0 * (r0 >> 4)) >> 16) + ((b1 * (r1 >> 4)) >> 16) + 2) >> 2;
    return q < 0 ? 0 : (q > 255 ? 255 : q);
}

struct Norm3 { float mean[3], stdv[3]; };

__global__ __launch_bounds__(256) void k_leres_input(const uint8_t *__restrict__ img, int H, int W, int h, int w, Norm3 nm,
                                                      float *__restrict__ out) {
    const int y = blockIdx.y, x = blockIdx.x * 256 + threadIdx.x;
    if (x >= w) return;
    const int64_t plane = (int64_t)h * w;
    double sy = (double)H / h, sx = (double)W / w;
    int y0 = y, y1 = y, x0 = x, x1 = x; float fy = 0.0f, fx = 0.0f;
    const bool same = (h == H && w == W);
    if (!same) { cv_src(y, H, sy, y0, y1, fy); cv_src(x, W, sx, x0, x1, fx); }
    for (int c = 0; c < 3; ++c) {      // output channel c = RGB <- BGR channel 2-c
        int sc = 2 - c;
        int q = same ? img[((int64_t)y * W + x) * 3 + sc]
                     : cv_lin_u8(img[((int64_t)y0 * W + x0) * 3 + sc], img[((int64_t)y0 * W + x1) * 3 + sc],
                                 img[((int64_t)y1 * W + x0) * 3 + sc], img[((int64_t)y1 * W + x1) * 3 + sc], fx, fy);
        float v = (float)q / 255.0f;
        out[c * plane + (int64_t)y * w + x] = (v - nm.mean[c]) / nm.stdv[c];
    }
}

// depth fp32 [h,w] -> u8: 65535*(d-min)/(max-min) -> uint16 (trunc) -> cvRound(x*255/65535) -> 255 - v
__global__ __launch_bounds__(256) void k_leres_quantize(const float *__restrict__ d, int64_t n, const float *__restrict__ mnmx,
                                                         uint8_t *__restrict__ out) {
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float mn = mnmx[0], mx = mnmx[1];
    float o = 0.0f;
    if ((double)(mx - mn) > 2.220446049250313e-16) o = 65535.0f * (d[i] - mn) / (mx - mn);
    uint16_t u16 = (uint16_t)o;
    float s = (float)u16 * (float)(255.0 / 65535.0);
    int v = (int)rintf(fabsf(s));
    v = v > 255 ? 255 : v;
    out[i] = (uint8_t)(255 - v);
}

// u8 [h,w] -> fp32 [H,W] with cv2.resize: INTER_AREA when enlarging (linear taps, area fractions), identity if same size
__global__ __launch_bounds__(256) void k_resize_u8_to_f32(const uint8_t *__restrict__ src, int h, int w, int H, int W,
                                                           float *__restrict__ out) {
    const int y = blockIdx.y, x = blockIdx.x * 256 + threadIdx.x;
    if (x >= W) return;
    int q;
    if (h == H && w == W) q = src[(int64_t)y * w + x];
    else {
        int y0, y1, x0, x1; float fy, fx;
        cv_src_area(y, h, (double)h / H, y0, y1, fy); cv_src_area(x, w, (double)w / W, x0, x1, fx);
        q = cv_lin_u8(src[(int64_t)y0 * w + x0], src[(int64_t)y0 * w + x1], src[(int64_t)y1 * w + x0], src[(int64_t)y1 * w + x1], fx, fy);
    }
    out[(int64_t)y * W + x] = (float)q;
}

// cv2.getRectSubPix(frame, (pw,ph), center) followed by cv2.resize(..., (W,H), INTER_LINEAR); u8 HWC 3 channels.
// getRectSubPix (8u): bilinear at constant sub-pixel offset, float weights, cvRound (imgwarp: getRectSubPix_8u32f + convert)
__device__ __forceinline__ int subpix(const uint8_t *__restrict__ f, int H, int W, int c, int px, int py, int ix, int iy, float a,
                                      float b) {
    int x0 = ix + px, y0 = iy + py;
    auto at = [&](int yy, int xx) { yy = yy < 0 ? 0 : (yy >= H ? H - 1 : yy); xx = xx < 0 ? 0 : (xx >= W ? W - 1 : xx);
                                    return (float)f[((int64_t)yy * W + xx) * 3 + c]; };
    float a11 = (1.0f - a) * (1.0f - b), a12 = a * (1.0f - b), a21 = (1.0f - a) * b, a22 = a * b;
    float v = at(y0, x0) * a11 + at(y0, x0 + 1) * a12 + at(y0 + 1, x0) * a21 + at(y0 + 1, x0 + 1) * a22;
    int q = (int)rintf(v);
    return q < 0 ? 0 : (q > 255 ? 255 : q);
}
__global__ __launch_bounds__(256) void k_crop_resize(const uint8_t *__restrict__ frame, int H, int W, int ph, int pw,
                                                      float cx, float cy, uint8_t *__restrict__ out) {
    const int y = blockIdx.y, x = blockIdx.x * 256 + threadIdx.x;
    if (x >= W) return;
    // patch origin (sub-pixel): center - (patch-1)/2
    float ox = cx - (float)(pw - 1) * 0.5f, oy = cy - (float)(ph - 1) * 0.5f;
    int ix = (int)floorf(ox), iy = (int)floorf(oy);
    float a = ox - (float)ix, b = oy - (float)iy;
    int y0 = y, y1 = y, x0 = x, x1 = x; float fy = 0.0f, fx = 0.0f;
    const bool same = (ph == H && pw == W);
    if (!same) { cv_src(y, ph, (double)ph / H, y0, y1, fy); cv_src(x, pw, (double)pw / W, x0, x1, fx); }
    for (int c = 0; c < 3; ++c) {
        int q = same ? subpix(frame, H, W, c, x, y, ix, iy, a, b)
                     : cv_lin_u8(subpix(frame, H, W, c, x0, y0, ix, iy, a, b), subpix(frame, H, W, c, x1, y0, ix, iy, a, b),
                                 subpix(frame, H, W, c, x0, y1, ix, iy, a, b), subpix(frame, H, W, c, x1, y1, ix, iy, a, b), fx, fy);
        out[((int64_t)y * W + x) * 3 + c] = (uint8_t)q;
    }
}

// The same arithmetic on a 64 x 4 output tile whose source pixels (<= 8 rows x 80 columns when the patch is not larger than the
// output, the Ken Burns case) are staged once in LDS as packed BGR words: 48 clamped byte gathers per output pixel become 16 LDS
// reads (25 -> ~6 us per 1024^2 frame).  A tile whose window does not fit uses the direct path of k_crop_resize.
__global__ __launch_bounds__(256) void k_crop_resize_tile(const uint8_t *__restrict__ frame, int H, int W, int ph, int pw,
                                                           float cx, float cy, uint8_t *__restrict__ out) {
    constexpr int TX = 64, TY = 4, RW = 80, RH = 8;
    __shared__ unsigned win[RH * RW];
    const int tid = threadIdx.x;
    const int bx = blockIdx.x * TX, by = blockIdx.y * TY;
    const int x = bx + (tid & 63), y = by + (tid >> 6);
    float ox = cx - (float)(pw - 1) * 0.5f, oy = cy - (float)(ph - 1) * 0.5f;
    const int ix = (int)floorf(ox), iy = (int)floorf(oy);
    const float a = ox - (float)ix, b = oy - (float)iy;
    const bool same = (ph == H && pw == W);
    const int xl = min(bx + TX - 1, W - 1), yl = min(by + TY - 1, H - 1);       // last output pixel of the tile
    int xmin = bx, xmax = xl, ymin = by, ymax = yl, t0, t1; float tf;
    if (!same) {
        cv_src(bx, pw, (double)pw / W, xmin, t1, tf); cv_src(xl, pw, (double)pw / W, t0, xmax, tf);
        cv_src(by, ph, (double)ph / H, ymin, t1, tf); cv_src(yl, ph, (double)ph / H, t0, ymax, tf);
    }
    const int rw = xmax - xmin + 2, rh = ymax - ymin + 2;                       // +1: the second tap of getRectSubPix
    const bool fits = rw <= RW && rh <= RH;                                     // block-uniform
    if (fits) {
        for (int i = tid; i < rh * rw; i += 256) {
            const int r = i / rw, c = i - r * rw;
            int yy = iy + ymin + r, xx = ix + xmin + c;
            yy = yy < 0 ? 0 : (yy >= H ? H - 1 : yy); xx = xx < 0 ? 0 : (xx >= W ? W - 1 : xx);
            const uint8_t *f = frame + ((int64_t)yy * W + xx) * 3;
            win[r * RW + c] = (unsigned)f[0] | ((unsigned)f[1] << 8) | ((unsigned)f[2] << 16);
        }
        __syncthreads();
    }
    if (x >= W || y >= H) return;
    int y0 = y, y1 = y, x0 = x, x1 = x; float fy = 0.0f, fx = 0.0f;
    if (!same) { cv_src(y, ph, (double)ph / H, y0, y1, fy); cv_src(x, pw, (double)pw / W, x0, x1, fx); }
    if (!fits) {
        for (int c = 0; c < 3; ++c) {
            int q = same ? subpix(frame, H, W, c, x, y, ix, iy, a, b)
                         : cv_lin_u8(subpix(frame, H, W, c, x0, y0, ix, iy, a, b), subpix(frame, H, W, c, x1, y0, ix, iy, a, b),
                                     subpix(frame, H, W, c, x0, y1, ix, iy, a, b), subpix(frame, H, W, c, x1, y1, ix, iy, a, b), fx, fy);
            out[((int64_t)y * W + x) * 3 + c] = (uint8_t)q;
        }
        return;
    }
    const float a11 = (1.0f - a) * (1.0f - b), a12 = a * (1.0f - b), a21 = (1.0f - a) * b, a22 = a * b;
    auto sub3 = [&](int px, int py, int q[3]) {                                // subpix() of the three channels from the window
        const unsigned *wp = win + (py - ymin) * RW + (px - xmin);
        const unsigned w00 = wp[0], w01 = wp[1], w10 = wp[RW], w11 = wp[RW + 1];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float v = (float)((w00 >> (8 * c)) & 255u) * a11 + (float)((w01 >> (8 * c)) & 255u) * a12 +
                            (float)((w10 >> (8 * c)) & 255u) * a21 + (float)((w11 >> (8 * c)) & 255u) * a22;
            const int r = (int)rintf(v);
            q[c] = r < 0 ? 0 : (r > 255 ? 255 : r);
        }
    };
    int o[3];
    if (same) sub3(x, y, o);
    else {
        int q00[3], q01[3], q10[3], q11[3];
        sub3(x0, y0, q00); sub3(x1, y0, q01); sub3(x0, y1, q10); sub3(x1, y1, q11);
#pragma unroll
        for (int c = 0; c < 3; ++c) o[c] = cv_lin_u8(q00[c], q01[c], q10[c], q11[c], fx, fy);
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) out[((int64_t)y * W + x) * 3 + c] = (uint8_t)o[c];
}

// ---- small fused reductions of the per-frame depth glue (replace ~25 torch kernels per frame; min/max are order-free) ---------
__device__ __forceinline__ unsigned f2ord(float f) { unsigned u = __float_as_uint(f); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); }
__device__ __forceinline__ float ord2f(unsigned o) { return __uint_as_float((o & 0x80000000u) ? (o & 0x7fffffffu) : ~o); }

// out[0] = min, out[1] = max of x[0..n): 256 blocks write partial {min, max} pairs, one block folds them (min / max are
// order-free).  A single-block version measured 150 us on a 1024^2 map -- one CU cannot pull 4 MB fast enough.
__global__ __launch_bounds__(256) void k_minmax_partial(const float *__restrict__ x, int64_t n, float *__restrict__ part) {
    __shared__ float smn[256], smx[256];
    float mn = INFINITY, mx = -INFINITY;
    const int64_t n4 = n >> 2;
    const float4 *x4 = reinterpret_cast<const float4 *>(x);
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        float4 v = x4[i];
        mn = fminf(fminf(mn, v.x), fminf(v.y, fminf(v.z, v.w)));
        mx = fmaxf(fmaxf(mx, v.x), fmaxf(v.y, fmaxf(v.z, v.w)));
    }
    if (blockIdx.x == 0) for (int64_t i = (n4 << 2) + threadIdx.x; i < n; i += 256) { float v = x[i]; mn = fminf(mn, v); mx = fmaxf(mx, v); }
    smn[threadIdx.x] = mn; smx[threadIdx.x] = mx;
    __syncthreads();
    for (int st = 128; st >= 1; st >>= 1) {
        if ((int)threadIdx.x < st) {
            smn[threadIdx.x] = fminf(smn[threadIdx.x], smn[threadIdx.x + st]);
            smx[threadIdx.x] = fmaxf(smx[threadIdx.x], smx[threadIdx.x + st]);
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) { part[2 * blockIdx.x] = smn[0]; part[2 * blockIdx.x + 1] = smx[0]; }
}
__global__ __launch_bounds__(256) void k_minmax_final(const float *__restrict__ part, int nparts, float *__restrict__ out) {
    __shared__ float smn[256], smx[256];
    float mn = INFINITY, mx = -INFINITY;
    for (int i = threadIdx.x; i < nparts; i += 256) { mn = fminf(mn, part[2 * i]); mx = fmaxf(mx, part[2 * i + 1]); }
    smn[threadIdx.x] = mn; smx[threadIdx.x] = mx;
    __syncthreads();
    for (int st = 128; st >= 1; st >>= 1) {
        if ((int)threadIdx.x < st) {
            smn[threadIdx.x] = fminf(smn[threadIdx.x], smn[threadIdx.x + st]);
            smx[threadIdx.x] = fmaxf(smx[threadIdx.x], smx[threadIdx.x + st]);
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) { out[0] = smn[0]; out[1] = smx[0]; }
}

// leres/__init__.py:143-145 `depth[depth == 0] = depth[depth > 0].min()`: pass 1 finds the smallest positive value (as ordered
// uint, atomicMin) and whether a zero exists; pass 2 rewrites only if both hold.  st[0] = ordered min positive, st[1] = zero seen.
__global__ __launch_bounds__(256) void k_minpos_scan(const float *__restrict__ x, int64_t n, unsigned *__restrict__ st) {
    __shared__ unsigned smn[256]; __shared__ int sz[256];
    unsigned mn = 0xffffffffu; int z = 0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        float v = x[i];
        if (v > 0.0f) mn = min(mn, __float_as_uint(v));      // positive floats order like their bit patterns
        z |= v == 0.0f;
    }
    smn[threadIdx.x] = mn; sz[threadIdx.x] = z;
    __syncthreads();
    for (int s2 = 128; s2 >= 1; s2 >>= 1) {
        if ((int)threadIdx.x < s2) { smn[threadIdx.x] = min(smn[threadIdx.x], smn[threadIdx.x + s2]); sz[threadIdx.x] |= sz[threadIdx.x + s2]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) { atomicMin(&st[0], smn[0]); if (sz[0]) atomicOr(&st[1], 1u); }
}
__global__ __launch_bounds__(256) void k_minpos_apply(float *__restrict__ x, int64_t n, const unsigned *__restrict__ st) {
    if (st[1] == 0u || st[0] == 0xffffffffu) return;         // no zero, or nothing positive: unchanged
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n && x[i] == 0.0f) x[i] = __uint_as_float(st[0]);
}

// kenburns_effect.py:928 `disparity / disparity.max() * baseline` (two roundings, like torch) with the maximum read on the device
__global__ __launch_bounds__(256) void k_normalise(const float *__restrict__ x, int64_t n, const float *__restrict__ minmax, float scale,
                                                    float *__restrict__ out, float *__restrict__ norm_max) {
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = (x[i] / minmax[1]) * scale;
    if (i == 0 && norm_max) norm_max[0] = (minmax[1] / minmax[1]) * scale;     // = max of `out` (the map is monotonic)
}

// cv2.minMaxLoc(depth[y0:y1, x0:x1]) (kenburns_effect.py:935): value and FIRST row-major position of the minimum and of the
// maximum.  Keys = (ordered value << 32) | index (min) and (ordered value << 32) | ~index (max) through 64-bit atomics.
__global__ __launch_bounds__(256) void k_crop_minmaxloc(const float *__restrict__ d, int W, int y0, int x0, int ch, int cw,
                                                         unsigned long long *__restrict__ keys) {
    __shared__ unsigned long long kmn[256], kmx[256];
    unsigned long long mn = ~0ull, mx = 0ull;
    const int64_t n = (int64_t)ch * cw;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        int y = (int)(i / cw), x = (int)(i - (int64_t)y * cw);
        unsigned o = f2ord(d[(int64_t)(y0 + y) * W + x0 + x]);
        unsigned long long a = ((unsigned long long)o << 32) | (unsigned)i, b = ((unsigned long long)o << 32) | (unsigned)(~(unsigned)i);
        mn = a < mn ? a : mn; mx = b > mx ? b : mx;
    }
    kmn[threadIdx.x] = mn; kmx[threadIdx.x] = mx;
    __syncthreads();
    for (int s2 = 128; s2 >= 1; s2 >>= 1) {
        if ((int)threadIdx.x < s2) {
            kmn[threadIdx.x] = kmn[threadIdx.x + s2] < kmn[threadIdx.x] ? kmn[threadIdx.x + s2] : kmn[threadIdx.x];
            kmx[threadIdx.x] = kmx[threadIdx.x + s2] > kmx[threadIdx.x] ? kmx[threadIdx.x + s2] : kmx[threadIdx.x];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) { atomicMin(&keys[0], kmn[0]); atomicMax(&keys[1], kmx[0]); }
}
// out[0..5] (float64) = {raw_min_normalised, raw_max_normalised, crop min, crop max, crop argmin, crop argmax}
__global__ void k_stats_pack(const float *__restrict__ minmax_raw, float scale, const unsigned long long *__restrict__ keys,
                             double *__restrict__ out) {
    out[0] = (double)((minmax_raw[0] / minmax_raw[1]) * scale);     // min / max of the normalised map: x -> (x/m)*s is monotonic
    out[1] = (double)((minmax_raw[1] / minmax_raw[1]) * scale);
    out[2] = (double)ord2f((unsigned)(keys[0] >> 32)); out[3] = (double)ord2f((unsigned)(keys[1] >> 32));
    out[4] = (double)(unsigned)(keys[0] & 0xffffffffull); out[5] = (double)(unsigned)(~(unsigned)(keys[1] & 0xffffffffull));
}

}  // namespace

// cv2.resize(u8 HWC [H,W,C], (w,h), INTER_LINEAR): utils/io_utils.py:254-274 scaledown_maxsize (frame: kenburns_effect.py:917)
__global__ __launch_bounds__(256) void k_resize_u8_linear(const uint8_t *__restrict__ src, int H, int W, int C, int h, int w,
                                                           uint8_t *__restrict__ dst) {
    const int y = blockIdx.y, x = blockIdx.x * 256 + threadIdx.x;
    if (x >= w) return;
    int y0, y1, x0, x1; float fy, fx;
    cv_src(y, H, (double)H / h, y0, y1, fy); cv_src(x, W, (double)W / w, x0, x1, fx);
    for (int c = 0; c < C; ++c)
        dst[((int64_t)y * w + x) * C + c] = (uint8_t)cv_lin_u8(src[((int64_t)y0 * W + x0) * C + c], src[((int64_t)y0 * W + x1) * C + c],
                                                              src[((int64_t)y1 * W + x0) * C + c], src[((int64_t)y1 * W + x1) * C + c], fx, fy);
}

extern "C" int csm_resize_u8_linear(const uint8_t *src_hwc, int H, int W, int C, int h, int w, uint8_t *dst_hwc, void *stream) {
    CSM_REQUIRE(src_hwc && dst_hwc && H > 0 && W > 0 && h > 0 && w > 0 && C > 0 && C <= 4);
    if (h == H && w == W) {
        CSM_HIP(hipMemcpyAsync(dst_hwc, src_hwc, (size_t)H * W * C, hipMemcpyDeviceToDevice, (hipStream_t)stream));
        return CSM_OK;
    }
    k_resize_u8_linear<<<dim3(csm::cdiv(w, 256), h), 256, 0, (hipStream_t)stream>>>(src_hwc, H, W, C, h, w, dst_hwc);
    return csm::check_launch("k_resize_u8_linear");
}

// cv2.resize(float32 HWC, (w,h), INTER_LINEAR): scaledown_maxsize / resize_pad on float masks (utils/io_utils.py:254-292 as called by
// prepare_refine_batch, animeinsseg/__init__.py:47).  [EXT: OpenCV resize.cpp, float path restated: the same source index / fraction
// as the uint8 path, HResizeLinear (row: s0 * (1 - fx) + s1 * fx) then VResizeLinear (r0 * (1 - fy) + r1 * fy), all in fp32.]
__global__ __launch_bounds__(256) void k_resize_f32_linear(const float *__restrict__ src, int H, int W, int C, int h, int w,
                                                            float *__restrict__ dst) {
    const int y = blockIdx.y, x = blockIdx.x * 256 + threadIdx.x;
    if (x >= w) return;
    int y0, y1, x0, x1; float fy, fx;
    cv_src(y, H, (double)H / h, y0, y1, fy); cv_src(x, W, (double)W / w, x0, x1, fx);
    const float a0 = 1.0f - fx, a1 = fx, b0 = 1.0f - fy, b1 = fy;
    for (int c = 0; c < C; ++c) {
        const float r0 = src[((int64_t)y0 * W + x0) * C + c] * a0 + src[((int64_t)y0 * W + x1) * C + c] * a1;
        const float r1 = src[((int64_t)y1 * W + x0) * C + c] * a0 + src[((int64_t)y1 * W + x1) * C + c] * a1;
        dst[((int64_t)y * w + x) * C + c] = r0 * b0 + r1 * b1;
    }
}

extern "C" int csm_resize_f32_linear(const float *src_hwc, int H, int W, int C, int h, int w, float *dst_hwc, void *stream) {
    CSM_REQUIRE(src_hwc && dst_hwc && H > 0 && W > 0 && h > 0 && w > 0 && C > 0 && C <= 4);
    if (h == H && w == W) {
        CSM_HIP(hipMemcpyAsync(dst_hwc, src_hwc, (size_t)H * W * C * sizeof(float), hipMemcpyDeviceToDevice, (hipStream_t)stream));
        return CSM_OK;
    }
    k_resize_f32_linear<<<dim3(csm::cdiv(w, 256), h), 256, 0, (hipStream_t)stream>>>(src_hwc, H, W, C, h, w, dst_hwc);
    return csm::check_launch("k_resize_f32_linear");
}

// cv2.resize(u8 [h,w], (W,H), INTER_LANCZOS4) -> float32 (kenburns_effect.py:572-575 when the 32-aligned LeReS map is LARGER than the
// frame, k > 1).  [EXT: OpenCV 4.10 resize.cpp restated: source coordinate (d + 0.5) scale - 0.5, 8 taps sx-3..sx+4 with replicate
// clamping per tap, interpolateLanczos4 coefficients in float -> short Q11 (cvRound), horizontal pass to int, vertical pass, result
// (v + 2^21) >> 22 saturated.]
__device__ __forceinline__ void lanczos4_q11(float x, int c[8]) {
    const double s45 = 0.70710678118654752440084436210485;
    const double cs[8][2] = {{1, 0}, {-s45, -s45}, {0, 1}, {s45, -s45}, {-1, 0}, {s45, s45}, {0, -1}, {-s45, s45}};
    float coeffs[8], sum = 0.0f;
    const double y0 = -(x + 3) * 3.14159265358979323846 * 0.25, s0 = sin(y0), c0 = cos(y0);
    for (int i = 0; i < 8; ++i) {
        const float y0_ = (x + 3 - i);
        if (fabsf(y0_) >= 1e-6f) {
            const double y = -y0_ * 3.14159265358979323846 * 0.25;
            coeffs[i] = (float)((cs[i][0] * s0 + cs[i][1] * c0) / (y * y));
        } else coeffs[i] = 1e30f;
        sum += coeffs[i];
    }
    sum = 1.0f / sum;
    for (int i = 0; i < 8; ++i) {
        const float v = coeffs[i] * sum * 2048.0f;
        int q = (int)rintf(v);
        c[i] = q > 32767 ? 32767 : (q < -32768 ? -32768 : q);
    }
}

__global__ __launch_bounds__(256) void k_resize_u8_lanczos4(const uint8_t *__restrict__ src, int h, int w, int H, int W,
                                                             float *__restrict__ out) {
    const int y = blockIdx.y, x = blockIdx.x * 256 + threadIdx.x;
    if (x >= W) return;
    float fx = (float)((x + 0.5) * ((double)w / W) - 0.5), fy = (float)((y + 0.5) * ((double)h / H) - 0.5);
    int sx = (int)floorf(fx), sy = (int)floorf(fy);
    fx -= (float)sx; fy -= (float)sy;
    int cx[8], cy[8];
    lanczos4_q11(fx, cx); lanczos4_q11(fy, cy);
    int acc = 0;
    for (int j = 0; j < 8; ++j) {
        int yy = sy - 3 + j; yy = yy < 0 ? 0 : (yy > h - 1 ? h - 1 : yy);
        int row = 0;
        for (int i = 0; i < 8; ++i) {
            int xx = sx - 3 + i; xx = xx < 0 ? 0 : (xx > w - 1 ? w - 1 : xx);
            row += (int)src[(int64_t)yy * w + xx] * cx[i];
        }
        acc += row * cy[j];
    }
    int v = (acc + (1 << 21)) >> 22;
    v = v < 0 ? 0 : (v > 255 ? 255 : v);
    out[(int64_t)y * W + x] = (float)v;
}

extern "C" int csm_resize_u8_lanczos4_to_f32(const uint8_t *src, int h, int w, int H, int W, float *out, void *stream) {
    CSM_REQUIRE(src && out && h > 0 && w > 0 && H > 0 && W > 0);
    k_resize_u8_lanczos4<<<dim3(csm::cdiv(W, 256), H), 256, 0, (hipStream_t)stream>>>(src, h, w, H, W, out);
    return csm::check_launch("k_resize_u8_lanczos4");
}

extern "C" int csm_leres_input(const uint8_t *img_hwc, int H, int W, int h, int w, float *out, void *stream) {
    CSM_REQUIRE(img_hwc && out && H > 0 && W > 0 && h > 0 && w > 0);
    Norm3 nm{{0.485f, 0.456f, 0.406f}, {0.229f, 0.224f, 0.225f}};
    k_leres_input<<<dim3(csm::cdiv(w, 256), h), 256, 0, (hipStream_t)stream>>>(img_hwc, H, W, h, w, nm, out);
    return csm::check_launch("k_leres_input");
}

extern "C" int csm_leres_quantize(const float *depth, int64_t n, const float *min_max_dev, uint8_t *out, void *stream) {
    CSM_REQUIRE(depth && min_max_dev && out && n > 0);
    k_leres_quantize<<<csm::cdiv(n, 256), 256, 0, (hipStream_t)stream>>>(depth, n, min_max_dev, out);
    return csm::check_launch("k_leres_quantize");
}

// ---- depth_adjustment_animesseg, one instance (kenburns_effect.py:68-78), in place ------------------------------------------
// plane = disp * mask.  Rows of the instance = rows whose plane has a positive entry (the reference tests `plane.sum(3) > 0`;
// disparities are non-negative, so "sum > 0" == "any > 0"); r0 = round_half_even(top + 0.97 (bottom - top)) in float64;
// val = max of the plane over rows >= r0 (zeros outside the mask included, like the reference's slice); pixels of the mask
// become val.  Skipped when the plane is empty (`plane.sum() == 0`).  scratch: H row maxima, H row flags, {val, apply}.
__global__ __launch_bounds__(256) void k_adjust_rows(const float *__restrict__ disp, const uint8_t *__restrict__ mask, int W,
                                                       float *__restrict__ rowmax, float *__restrict__ rowflag) {
    __shared__ float smax[256];
    __shared__ int sflag[256];
    const int r = blockIdx.x;
    float mx = -INFINITY; int fl = 0;           // bit 0: a positive entry, bit 1: a non-zero entry
    for (int x = threadIdx.x; x < W; x += 256) {
        float p = disp[(int64_t)r * W + x] * (mask[(int64_t)r * W + x] ? 1.0f : 0.0f);
        mx = fmaxf(mx, p);
        fl |= (p > 0.0f ? 1 : 0) | (p != 0.0f ? 2 : 0);
    }
    smax[threadIdx.x] = mx; sflag[threadIdx.x] = fl;
    __syncthreads();
    for (int st = 128; st >= 1; st >>= 1) {
        if ((int)threadIdx.x < st) {
            smax[threadIdx.x] = fmaxf(smax[threadIdx.x], smax[threadIdx.x + st]);
            sflag[threadIdx.x] |= sflag[threadIdx.x + st];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) { rowmax[r] = smax[0]; rowflag[r] = (float)sflag[0]; }
}

__global__ __launch_bounds__(256) void k_adjust_pick(const float *__restrict__ rowmax, const float *__restrict__ rowflag, int H,
                                                       float *__restrict__ out2) {
    __shared__ int stop[256], sbot[256], snz[256];
    __shared__ float smax[256];
    int top = H, bot = -1, nz = 0;
    for (int r = threadIdx.x; r < H; r += 256) {
        int f = (int)rowflag[r];
        if (f & 1) { top = min(top, r); bot = max(bot, r); }
        nz |= f & 2;
    }
    stop[threadIdx.x] = top; sbot[threadIdx.x] = bot; snz[threadIdx.x] = nz;
    __syncthreads();
    for (int st = 128; st >= 1; st >>= 1) {
        if ((int)threadIdx.x < st) {
            stop[threadIdx.x] = min(stop[threadIdx.x], stop[threadIdx.x + st]);
            sbot[threadIdx.x] = max(sbot[threadIdx.x], sbot[threadIdx.x + st]);
            snz[threadIdx.x] |= snz[threadIdx.x + st];
        }
        __syncthreads();
    }
    top = stop[0]; bot = sbot[0]; nz = snz[0];
    const bool apply = bot >= 0 && nz != 0;
    const int r0 = apply ? (int)rint((double)top + (0.97 * (double)(bot - top))) : H;
    float mx = -INFINITY;
    for (int r = threadIdx.x; r < H; r += 256)
        if (r >= r0) mx = fmaxf(mx, rowmax[r]);
    smax[threadIdx.x] = mx;
    __syncthreads();
    for (int st = 128; st >= 1; st >>= 1) {
        if ((int)threadIdx.x < st) smax[threadIdx.x] = fmaxf(smax[threadIdx.x], smax[threadIdx.x + st]);
        __syncthreads();
    }
    if (threadIdx.x == 0) { out2[0] = smax[0]; out2[1] = apply ? 1.0f : 0.0f; }
}

__global__ __launch_bounds__(256) void k_adjust_apply(float *__restrict__ disp, const uint8_t *__restrict__ mask, int64_t n,
                                                        const float *__restrict__ out2) {
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n || out2[1] == 0.0f) return;
    const float m = mask[i] ? 1.0f : 0.0f;
    disp[i] = ((1.0f - m) * disp[i]) + (m * out2[0]);          // kenburns_effect.py:78, literally
}

extern "C" int csm_depth_adjust_instance(float *disp, const uint8_t *mask, int H, int W, float *scratch, void *stream) {
    CSM_REQUIRE(disp && mask && scratch && H > 0 && W > 0);
    hipStream_t st = (hipStream_t)stream;
    k_adjust_rows<<<H, 256, 0, st>>>(disp, mask, W, scratch, scratch + H);
    k_adjust_pick<<<1, 256, 0, st>>>(scratch, scratch + H, H, scratch + 2 * H);
    k_adjust_apply<<<csm::cdiv((int64_t)H * W, 256), 256, 0, st>>>(disp, mask, (int64_t)H * W, scratch + 2 * H);
    return csm::check_launch("k_adjust_*");
}

extern "C" int csm_minmax(const float *x, int64_t n, float *out2, float *scratch512, void *stream) {
    CSM_REQUIRE(x && out2 && scratch512 && n > 0 && !(((uintptr_t)x) & 15));
    const int nparts = (int)(n >= (1 << 18) ? 256 : (n + 1023) / 1024 > 0 ? (n + 1023) / 1024 : 1);
    k_minmax_partial<<<nparts, 256, 0, (hipStream_t)stream>>>(x, n, scratch512);
    k_minmax_final<<<1, 256, 0, (hipStream_t)stream>>>(scratch512, nparts, out2);
    return csm::check_launch("k_minmax");
}

extern "C" int csm_fill_zero_min_positive(float *x, int64_t n, unsigned *scratch2, void *stream) {
    CSM_REQUIRE(x && scratch2 && n > 0);
    hipStream_t st = (hipStream_t)stream;
    CSM_HIP(hipMemsetAsync(scratch2, 0xff, 4, st));
    CSM_HIP(hipMemsetAsync(scratch2 + 1, 0, 4, st));
    k_minpos_scan<<<512, 256, 0, st>>>(x, n, scratch2);
    k_minpos_apply<<<csm::cdiv(n, 256), 256, 0, st>>>(x, n, scratch2);
    return csm::check_launch("k_minpos_*");
}

extern "C" int csm_normalise_disparity(const float *x, int64_t n, const float *minmax_dev, float scale, float *out, float *norm_max_out,
                                       void *stream) {
    CSM_REQUIRE(x && minmax_dev && out && n > 0);
    k_normalise<<<csm::cdiv(n, 256), 256, 0, (hipStream_t)stream>>>(x, n, minmax_dev, scale, out, norm_max_out);
    return csm::check_launch("k_normalise");
}

extern "C" int csm_depth_range_stats(const float *minmax_raw_dev, float scale, const float *depth, int H, int W, int y0, int x0,
                                     int crop_h, int crop_w, unsigned long long *scratch2, double *out6, void *stream) {
    CSM_REQUIRE(minmax_raw_dev && depth && scratch2 && out6 && crop_h > 0 && crop_w > 0 && y0 >= 0 && x0 >= 0 &&
                y0 + crop_h <= H && x0 + crop_w <= W && (int64_t)crop_h * crop_w < (1ll << 32));
    hipStream_t st = (hipStream_t)stream;
    CSM_HIP(hipMemsetAsync(scratch2, 0xff, 8, st));
    CSM_HIP(hipMemsetAsync(scratch2 + 1, 0, 8, st));
    k_crop_minmaxloc<<<256, 256, 0, st>>>(depth, W, y0, x0, crop_h, crop_w, scratch2);
    k_stats_pack<<<1, 1, 0, st>>>(minmax_raw_dev, scale, scratch2, out6);
    return csm::check_launch("k_crop_minmaxloc");
}

// uint8 HWC image -> float32 CHW in [0, 1]: `img.permute(2, 0, 1)[None].float() * (1.0 / 255.0)` (kenburns_effect.py:878-880 feeds every
// net input from this tensor) -- one pass instead of torch's convert, multiply and contiguous copy; the same fp32 product
namespace {
__global__ __launch_bounds__(256) void k_u8_hwc_to_f32_chw(const uint8_t *__restrict__ src, int64_t plane, float *__restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= plane) return;
    const float s = (float)(1.0 / 255.0);
    out[i] = (float)src[i * 3] * s; out[plane + i] = (float)src[i * 3 + 1] * s; out[2 * plane + i] = (float)src[i * 3 + 2] * s;
}
}  // namespace
extern "C" int csm_u8_hwc_to_f32_chw(const uint8_t *src_hwc, int H, int W, float *out_chw, void *stream) {
    CSM_REQUIRE(src_hwc && out_chw && H > 0 && W > 0);
    const int64_t plane = (int64_t)H * W;
    k_u8_hwc_to_f32_chw<<<csm::cdiv(plane, 256), 256, 0, (hipStream_t)stream>>>(src_hwc, plane, out_chw);
    return csm::check_launch("k_u8_hwc_to_f32_chw");
}

extern "C" int csm_resize_u8_to_f32(const uint8_t *src, int h, int w, int H, int W, float *out, void *stream) {
    CSM_REQUIRE(src && out && h > 0 && w > 0 && H >= h && W >= w);
    k_resize_u8_to_f32<<<dim3(csm::cdiv(W, 256), H), 256, 0, (hipStream_t)stream>>>(src, h, w, H, W, out);
    return csm::check_launch("k_resize_u8_to_f32");
}

extern "C" int csm_crop_resize_u8(const uint8_t *frame_hwc, int H, int W, int patch_h, int patch_w, float center_x,
                                  float center_y, uint8_t *out_hwc, void *stream) {
    CSM_REQUIRE(frame_hwc && out_hwc && frame_hwc != out_hwc && H > 0 && W > 0 && patch_h > 0 && patch_w > 0);
    k_crop_resize_tile<<<dim3(csm::cdiv(W, 64), csm::cdiv(H, 4)), 256, 0, (hipStream_t)stream>>>(frame_hwc, H, W, patch_h, patch_w,
                                                                                                   center_x, center_y, out_hwc);
    return csm::check_launch("k_crop_resize");
}

// ---- bokeh depth-of-field (utils/effects.py:12-181) --------------------------------------------------------------
namespace {

// kernel_bokeh (utils/effects.py:16-74): 32 depth-weighted samples along (dx, dy) per (pixel, channel); HWC-interleaved fp32
// image (the reference kernel indexes raw memory as (y*W+x)*3+c).
//
// One PIXEL per lane (its three channels share the sample positions and weights; each channel keeps its own accumulation chain,
// so the bits are those of the one-(pixel, channel)-per-thread reference).  A block owns a 32 x 8 pixel tile and stages the
// tile + R-px halo as {r, g, b, depth} texels in LDS: a sample is one ds_read_b128 instead of two dependent global gathers per
// channel (the round-2 kernel: 192 scattered 4-B loads per pixel, 103 us per pass at 1024^2).  The sample offsets are
// round(d * dir * (s - n/2) * min(H, W)) with d = the caller's depth plane; bokeh_blur feeds d <= 0.0005 (utils/effects.py:153),
// i.e. |offset| <= 0.008 min(H, W), and R is picked for that.  A sample that falls outside the staged window anyway (any other
// depth plane) is fetched from global memory: same result for every input, only slower.
// FINISH: the third pass of bokeh_blur also applies utils/effects.py:172,179-180 -- uint8(pow((diag + rhom) / 2, 1 / lightness) * 255)
// with diag = this pass's INPUT at the pixel (the window centre) and rhom = its result -- and writes the uint8 frame instead of
// the float plane: one launch, one 12 MB write and two 12 MB reads less per frame.
template <int R, bool FINISH>
__global__ __launch_bounds__(256) void k_bokeh_pass_tile(const float *__restrict__ img, const float *__restrict__ depth,
                                                          float *__restrict__ out, uint8_t *__restrict__ out_u8, float inv_lf,
                                                          int H, int W, int nsamples, float dx, float dy) {
    constexpr int TX = 32, TY = 8, WW = TX + 2 * R, WH = TY + 2 * R;
    __shared__ float4 win[WH * WW];
    const int tid = threadIdx.x;
    const int bx = blockIdx.x * TX, by = blockIdx.y * TY;
    const int x0 = bx - R, y0 = by - R;
    for (int i = tid; i < WH * WW; i += 256) {
        const int wy = i / WW, wx = i - wy * WW;
        const int gx = x0 + wx, gy = y0 + wy;
        float4 t = float4{0.0f, 0.0f, 0.0f, 0.0f};
        if (gx >= 0 && gx < W && gy >= 0 && gy < H) {
            const int64_t o = (int64_t)gy * W + gx;
            t.x = img[o * 3]; t.y = img[o * 3 + 1]; t.z = img[o * 3 + 2]; t.w = depth[o];
        }
        win[i] = t;
    }
    __syncthreads();
    const int lx = tid & 31, ly = tid >> 5;
    const int x = bx + lx, y = by + ly;
    float r0 = 0.0f, r1 = 0.0f, r2 = 0.0f;
    float4 ctr_keep = float4{0.0f, 0.0f, 0.0f, 0.0f};
    if (x < W && y < H) {
        const float4 ctr = win[(ly + R) * WW + lx + R];
        ctr_keep = ctr;
        const int im_size = min(H, W), off = nsamples / 2;
        const float ddx = dx * ctr.w, ddy = dy * ctr.w;
        float weight = 0.0f, c0 = 0.0f, c1 = 0.0f, c2 = 0.0f;
        // a block whose staged window lies inside the image (all but the border tiles) needs no image-bounds test for a sample that
        // falls inside the window: one unsigned compare per axis instead of four signed ones and two branches
        const bool interior = x0 >= 0 && y0 >= 0 && x0 + WW <= W && y0 + WH <= H;        // block-uniform
        if (interior) {
            const int wx0 = lx + R, wy0 = ly + R;
            for (int s = 0; s < nsamples; ++s) {
                const int sp = (s - off) * im_size;
                const int ox = (int)roundf(ddx * (float)sp), oy = (int)roundf(ddy * (float)sp);
                const int wx = wx0 + ox, wy = wy0 + oy;
                float4 t;
                if ((unsigned)wx < (unsigned)WW && (unsigned)wy < (unsigned)WH) t = win[wy * WW + wx];
                else {
                    const int x_ = x + ox, y_ = y + oy;
                    if (x_ >= W || y_ >= H || x_ < 0 || y_ < 0) continue;
                    const int64_t o = (int64_t)y_ * W + x_; t = float4{img[o * 3], img[o * 3 + 1], img[o * 3 + 2], depth[o]};
                }
                weight += t.w;
                c0 += t.x * t.w; c1 += t.y * t.w; c2 += t.z * t.w;
            }
        } else {
            for (int s = 0; s < nsamples; ++s) {
                const int sp = (s - off) * im_size;
                const int x_ = x + (int)roundf(ddx * (float)sp), y_ = y + (int)roundf(ddy * (float)sp);
                if (x_ >= W || y_ >= H || x_ < 0 || y_ < 0) continue;
                const int wx = x_ - x0, wy = y_ - y0;
                float4 t;
                if (wx >= 0 && wx < WW && wy >= 0 && wy < WH) t = win[wy * WW + wx];
                else { const int64_t o = (int64_t)y_ * W + x_; t = float4{img[o * 3], img[o * 3 + 1], img[o * 3 + 2], depth[o]}; }
                weight += t.w;
                c0 += t.x * t.w; c1 += t.y * t.w; c2 += t.z * t.w;
            }
        }
        r0 = weight != 0.0f ? c0 / weight : ctr.x;
        r1 = weight != 0.0f ? c1 / weight : ctr.y;
        r2 = weight != 0.0f ? c2 / weight : ctr.z;
    }
    __syncthreads();                                   // the window is dead: reuse it to write whole row segments
    const int row_vals = (W - bx < TX ? W - bx : TX) * 3;
    if (FINISH) {
        uint8_t *stage8 = reinterpret_cast<uint8_t *>(win);
        if (x < W && y < H) {
            const float4 c = ctr_keep;
            stage8[tid * 3] = (uint8_t)(powf((c.x + r0) / 2.0f, inv_lf) * 255.0f);
            stage8[tid * 3 + 1] = (uint8_t)(powf((c.y + r1) / 2.0f, inv_lf) * 255.0f);
            stage8[tid * 3 + 2] = (uint8_t)(powf((c.z + r2) / 2.0f, inv_lf) * 255.0f);
        }
        __syncthreads();
        for (int i = tid; i < TY * TX * 3; i += 256) {
            const int ry = i / (TX * 3), rx = i - ry * (TX * 3);
            if (by + ry < H && rx < row_vals) out_u8[((int64_t)(by + ry) * W + bx) * 3 + rx] = stage8[i];
        }
    } else {
        float *stage = reinterpret_cast<float *>(win);
        stage[tid * 3] = r0; stage[tid * 3 + 1] = r1; stage[tid * 3 + 2] = r2;
        __syncthreads();
        for (int i = tid; i < TY * TX * 3; i += 256) {
            const int ry = i / (TX * 3), rx = i - ry * (TX * 3);
            if (by + ry < H && rx < row_vals) out[((int64_t)(by + ry) * W + bx) * 3 + rx] = stage[i];
        }
    }
}

// img u8 HWC -> (img/255)^lightness  (utils/effects.py:155-156)
__global__ __launch_bounds__(256) void k_bokeh_highlight(const uint8_t *__restrict__ img, float *__restrict__ out, int64_t n, float lf) {
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    out[i] = powf((float)img[i] / 255.0f, lf);
}
// ((diag + rhom)/2)^(1/lightness) * 255 -> u8  (utils/effects.py:172, :179-180)
__global__ __launch_bounds__(256) void k_bokeh_finish(const float *__restrict__ a, const float *__restrict__ b, uint8_t *__restrict__ out,
                                                       int64_t n, float inv_lf) {
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float v = powf((a[i] + b[i]) / 2.0f, inv_lf) * 255.0f;
    out[i] = (uint8_t)v;
}
// depth u8 -> bokeh depth map (utils/effects.py:146-153, :162-163): mx - |d - focal|, minus min, / max, 1 - x, * 0.0005
// stats = {max(d), min(mx-|d-f|), max(after minus min)} are computed by the caller (scalar reductions)
__global__ __launch_bounds__(256) void k_bokeh_depth(const uint8_t *__restrict__ d8, float *__restrict__ out, int64_t n, float dmax,
                                                      float focal, float mn, float mx2) {
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float v = dmax - fabsf((float)d8[i] - focal);
    v = v - mn;
    v = v / mx2;
    v = 1.0f - v;
    out[i] = v * 0.0005f;
}
// colorize(depth, cmap='gray_r')[...,0] (zoedepth/utils/misc.py:97-135): (v - vmin)/(vmax - vmin) -> matplotlib LUT index
__global__ __launch_bounds__(256) void k_colorize_gray_r(const float *__restrict__ v, uint8_t *__restrict__ out, int64_t n, float vmin,
                                                          float vmax) {
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float x = vmin != vmax ? (v[i] - vmin) / (vmax - vmin) : 0.0f;
    // matplotlib Colormap.__call__: xa = x*256; xa==256 -> 255; clip to [-1, 256]; int(); <0 -> under (lut[0]), >255 -> over (lut[255])
    float xa = x * 256.0f;
    if (xa == 256.0f) xa = 255.0f;
    xa = fminf(fmaxf(xa, -1.0f), 256.0f);
    int k = (int)xa;
    k = k < 0 ? 0 : (k > 255 ? 255 : k);
    out[i] = (uint8_t)(255 - k);          // gray_r LUT: uint8((1 - k/255) * 255) == 255 - k  (checked against matplotlib in tests)
}

}  // namespace

static int bokeh_pass_launch(const float *img, const float *depth, float *out, uint8_t *out_u8, float inv_lf, int H, int W, int nsamples,
                             float dx, float dy, hipStream_t st) {
    // halo for bokeh_blur's depth scale (see the kernel): |offset| <= round(0.0005 * (nsamples / 2) * min(H, W)), +1 for safety
    const int reach = (int)(0.0005 * (double)((nsamples + 1) / 2) * (double)(H < W ? H : W) + 0.5) + 1;
    const dim3 grid((unsigned)csm::cdiv(W, 32), (unsigned)csm::cdiv(H, 8));
    if (out_u8) {
        if (reach <= 9) k_bokeh_pass_tile<9, true><<<grid, 256, 0, st>>>(img, depth, out, out_u8, inv_lf, H, W, nsamples, dx, dy);
        else if (reach <= 16) k_bokeh_pass_tile<16, true><<<grid, 256, 0, st>>>(img, depth, out, out_u8, inv_lf, H, W, nsamples, dx, dy);
        else k_bokeh_pass_tile<20, true><<<grid, 256, 0, st>>>(img, depth, out, out_u8, inv_lf, H, W, nsamples, dx, dy);
    } else {
        if (reach <= 9) k_bokeh_pass_tile<9, false><<<grid, 256, 0, st>>>(img, depth, out, out_u8, inv_lf, H, W, nsamples, dx, dy);
        else if (reach <= 16) k_bokeh_pass_tile<16, false><<<grid, 256, 0, st>>>(img, depth, out, out_u8, inv_lf, H, W, nsamples, dx, dy);
        else k_bokeh_pass_tile<20, false><<<grid, 256, 0, st>>>(img, depth, out, out_u8, inv_lf, H, W, nsamples, dx, dy);
    }
    return csm::check_launch("k_bokeh_pass");
}

extern "C" int csm_bokeh_pass(const float *img_hwc, const float *depth, float *out_hwc, int H, int W, int nsamples, float dx, float dy,
                              void *stream) {
    CSM_REQUIRE(img_hwc && depth && out_hwc && img_hwc != out_hwc && H > 0 && W > 0 && nsamples > 0);
    return bokeh_pass_launch(img_hwc, depth, out_hwc, nullptr, 1.0f, H, W, nsamples, dx, dy, (hipStream_t)stream);
}
extern "C" int csm_bokeh_pass_finish(const float *diag_hwc, const float *depth, uint8_t *out_hwc_u8, int H, int W, int nsamples, float dx, float dy,
                                     float lightness, void *stream) {
    CSM_REQUIRE(diag_hwc && depth && out_hwc_u8 && H > 0 && W > 0 && nsamples > 0 && lightness != 0.0f);
    return bokeh_pass_launch(diag_hwc, depth, nullptr, out_hwc_u8, (float)(1.0 / (double)lightness), H, W, nsamples, dx, dy, (hipStream_t)stream);
}
extern "C" int csm_bokeh_highlight(const uint8_t *img_hwc, float *out_hwc, int64_t n, float lightness, void *stream) {
    CSM_REQUIRE(img_hwc && out_hwc && n > 0);
    k_bokeh_highlight<<<csm::cdiv(n, 256), 256, 0, (hipStream_t)stream>>>(img_hwc, out_hwc, n, lightness);
    return csm::check_launch("k_bokeh_highlight");
}
extern "C" int csm_bokeh_finish(const float *diag_hwc, const float *rhom_hwc, uint8_t *out_hwc, int64_t n, float lightness, void *stream) {
    CSM_REQUIRE(diag_hwc && rhom_hwc && out_hwc && n > 0 && lightness != 0.0f);
    k_bokeh_finish<<<csm::cdiv(n, 256), 256, 0, (hipStream_t)stream>>>(diag_hwc, rhom_hwc, out_hwc, n, (float)(1.0 / (double)lightness));
    return csm::check_launch("k_bokeh_finish");
}
extern "C" int csm_bokeh_depth(const uint8_t *depth_u8, float *out, int64_t n, float dmax, float focal_plane, float mn, float mx2,
                               void *stream) {
    CSM_REQUIRE(depth_u8 && out && n > 0);
    k_bokeh_depth<<<csm::cdiv(n, 256), 256, 0, (hipStream_t)stream>>>(depth_u8, out, n, dmax, focal_plane, mn, mx2);
    return csm::check_launch("k_bokeh_depth");
}
// General form of the depth map of bokeh_blur (utils/effects.py:146-153, :162-163) for the reference's own defaults -- float or uint8
// depth, focal_plane optional, any depth_factor:  d' = has_focal ? max(d) - |d - focal| : d;  d'' = depth_factor != 1 ? d' ^ factor : d';
// out = (1 - (d'' - min d'') / max(d'' - min d'')) * 0.0005.  np.power(float32, 2) is a square (numpy's fast path), other exponents
// go through powf.  Every step is one fp32 operation per element as numpy performs it; x -> x - mn is monotone under rounding, so
// max(d'' - mn) = fl(mx - mn) exactly and two min/max reductions (of d and of d'') are all the statistics needed.
__global__ __launch_bounds__(256) void k_bokeh_depth_pre(const void *__restrict__ depth, int is_u8, int64_t n, const float *__restrict__ dmm,
                                                          int has_focal, float focal, float factor, float *__restrict__ tmp) {
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float v = is_u8 ? (float)reinterpret_cast<const uint8_t *>(depth)[i] : reinterpret_cast<const float *>(depth)[i];
    if (has_focal) v = dmm[1] - fabsf(v - focal);
    if (factor == 2.0f) v = v * v;
    else if (factor != 1.0f) v = powf(v, factor);
    tmp[i] = v;
}
__global__ __launch_bounds__(256) void k_bokeh_depth_post(const float *__restrict__ tmp, int64_t n, const float *__restrict__ tmm,
                                                           float *__restrict__ out) {
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float mn = tmm[0], mx2 = tmm[1] - tmm[0];
    float v = tmp[i] - mn;
    v = v / mx2;
    v = 1.0f - v;
    out[i] = v * 0.0005f;
}
extern "C" int csm_bokeh_depth_general(const void *depth, int is_u8, int64_t n, int has_focal, float focal_plane, float depth_factor,
                                       float *tmp, float *mm4, float *scratch512, float *out, void *stream) {
    CSM_REQUIRE(depth && tmp && mm4 && scratch512 && out && n > 0 && !(((uintptr_t)tmp) & 15));
    hipStream_t st = (hipStream_t)stream;
    if (has_focal) {
        // max(depth): through the float reduction (a uint8 plane is widened first -- into `out`, which is rewritten at the end)
        const float *src = reinterpret_cast<const float *>(depth);
        if (is_u8) {
            k_bokeh_depth_pre<<<csm::cdiv(n, 256), 256, 0, st>>>(depth, 1, n, mm4, 0, 0.0f, 1.0f, out);
            src = out;
        }
        CSM_REQUIRE(!(((uintptr_t)src) & 15));
        int rc = csm_minmax(src, n, mm4, scratch512, stream); if (rc) return rc;
    }
    k_bokeh_depth_pre<<<csm::cdiv(n, 256), 256, 0, st>>>(depth, is_u8, n, mm4, has_focal, focal_plane, depth_factor, tmp);
    int rc = csm_minmax(tmp, n, mm4 + 2, scratch512, stream); if (rc) return rc;
    k_bokeh_depth_post<<<csm::cdiv(n, 256), 256, 0, st>>>(tmp, n, mm4 + 2, out);
    return csm::check_launch("k_bokeh_depth_general");
}
extern "C" int csm_colorize_gray_r(const float *value, uint8_t *out, int64_t n, float vmin, float vmax, void *stream) {
    CSM_REQUIRE(value && out && n > 0);
    k_colorize_gray_r<<<csm::cdiv(n, 256), 256, 0, (hipStream_t)stream>>>(value, out, n, vmin, vmax);
    return csm::check_launch("k_colorize_gray_r");
}

// ---- whole-tensor mean / std and the normalise / de-normalise steps around the Inpaint and Refine nets ------------------------------
// pointcloud_inpainting.py:116-131, :196-203 and disparity_refinement.py:97-107, :121-126: x.mean(), x.std(unbiased=False), (x - mean) /
// (std + 1e-7), y * (std + 1e-7) + mean, then clip(0, 1) (image) or threshold(0) (disparity).  The reductions accumulate in double
// (two passes: mean, then the centred second moment) -- at least as accurate as any fp32 summation order torch may use; results are
// compared with the reference modules at fp32 tolerance.  Statistics stay on the device.
namespace {
__global__ __launch_bounds__(256) void k_sum_partial(const float *__restrict__ x, int64_t n, const float *__restrict__ centre, double *__restrict__ part) {
    __shared__ double red[256];
    const double c = centre ? (double)centre[0] : 0.0;
    double s = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const double v = (double)x[i] - c;
        s += centre ? v * v : v;
    }
    red[threadIdx.x] = s;
    __syncthreads();
    for (int st = 128; st >= 1; st >>= 1) { if ((int)threadIdx.x < st) red[threadIdx.x] += red[threadIdx.x + st]; __syncthreads(); }
    if (threadIdx.x == 0) part[blockIdx.x] = red[0];
}
__global__ __launch_bounds__(256) void k_sum_final(const double *__restrict__ part, int nparts, int64_t n, int is_var, float *__restrict__ out) {
    __shared__ double red[256];
    double s = 0.0;
    for (int i = threadIdx.x; i < nparts; i += 256) s += part[i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int st = 128; st >= 1; st >>= 1) { if ((int)threadIdx.x < st) red[threadIdx.x] += red[threadIdx.x + st]; __syncthreads(); }
    if (threadIdx.x == 0) out[0] = is_var ? (float)sqrt(red[0] / (double)n) : (float)(red[0] / (double)n);
}
__global__ __launch_bounds__(256) void k_normalise_ms(const float *__restrict__ x, int64_t n, const float *__restrict__ ms, float *__restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    out[i] = (x[i] - ms[0]) / (ms[1] + 0.0000001f);
}
__global__ __launch_bounds__(256) void k_denormalise_ms(const float *__restrict__ x, int64_t n, const float *__restrict__ ms, int mode, float *__restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float v = x[i] * (ms[1] + 0.0000001f) + ms[0];
    if (mode == 1) v = fminf(fmaxf(v, 0.0f), 1.0f);            // tensor.clip(0.0, 1.0)
    else if (mode == 2) v = v > 0.0f ? v : 0.0f;              // torch.nn.functional.threshold(v, 0.0, 0.0)
    out[i] = v;
}

// torch.nn.functional.interpolate(mode='bilinear') of `planes` independent [H, W] planes (aten upsample_bilinear2d: UpSample.h
// area_pixel_compute_source_index; identity when the sizes match) -- the resize branches of depth_adjustment_animesseg
// (kenburns_effect.py:49-52, :89-90) and of disparity_estimation (models/__init__.py:46-49)
__device__ __forceinline__ void aten_src(int dst, int in_size, int out_size, float scale, bool align, int &i0, int &i1, float &l0, float &l1) {
    if (in_size == out_size) { i0 = i1 = dst; l0 = 1.0f; l1 = 0.0f; return; }
    float real;
    if (align) real = scale * (float)dst;
    else { real = scale * ((float)dst + 0.5f) - 0.5f; if (real < 0.0f) real = 0.0f; }
    i0 = min((int)real, in_size - 1);
    i1 = i0 + (i0 < in_size - 1 ? 1 : 0);
    l1 = fminf(fmaxf(real - (float)i0, 0.0f), 1.0f);
    l0 = 1.0f - l1;
}
__global__ __launch_bounds__(256) void k_bilinear_planes(const float *__restrict__ in, int planes, int H, int W, int h, int w, int align,
                                                          float sh, float sw, float *__restrict__ out) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x, total = (int64_t)planes * h * w;
    if (idx >= total) return;
    const int x = (int)(idx % w); int64_t t = idx / w; const int y = (int)(t % h); const int64_t p = t / h;
    int y0, y1, x0, x1; float hl0, hl1, wl0, wl1;
    aten_src(y, H, h, sh, align != 0, y0, y1, hl0, hl1); aten_src(x, W, w, sw, align != 0, x0, x1, wl0, wl1);
    const float *P = in + p * (int64_t)H * W;
    out[idx] = hl0 * (wl0 * P[(int64_t)y0 * W + x0] + wl1 * P[(int64_t)y0 * W + x1]) + hl1 * (wl0 * P[(int64_t)y1 * W + x0] + wl1 * P[(int64_t)y1 * W + x1]);
}

// AnimeInstances.resize (animeinsseg/anime_instances.py:268-280): interpolate(masks.float(), (h, w), mode='area') > 0.3 = adaptive
// average pooling of 0 / 1 values.  Window [floor(o I / O), ceil((o + 1) I / O)) per axis and `sum / kH / kW` as aten's device kernel
// writes it (AdaptiveAveragePooling: START_IND / END_IND in float); the sum of a 0 / 1 window is an exact integer in any order.
__global__ __launch_bounds__(256) void k_mask_area_threshold(const uint8_t *__restrict__ m, int n, int H, int W, int h, int w, float thr,
                                                              uint8_t *__restrict__ out) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x, total = (int64_t)n * h * w;
    if (idx >= total) return;
    const int x = (int)(idx % w); int64_t t = idx / w; const int y = (int)(t % h); const int64_t k = t / h;
    const int ys = (int)floorf((float)(y * H) / (float)h), ye = (int)ceilf((float)((y + 1) * H) / (float)h);
    const int xs = (int)floorf((float)(x * W) / (float)w), xe = (int)ceilf((float)((x + 1) * W) / (float)w);
    const uint8_t *P = m + k * (int64_t)H * W;
    float sum = 0.0f;
    for (int yy = ys; yy < ye; ++yy)
        for (int xx = xs; xx < xe; ++xx) sum += P[(int64_t)yy * W + xx] ? 1.0f : 0.0f;
    out[idx] = (sum / (float)(ye - ys) / (float)(xe - xs)) > thr ? 1 : 0;
}
}  // namespace

extern "C" int csm_mean_std(const float *x, int64_t n, float *out2, void *scratch, void *stream) {
    CSM_REQUIRE(x && out2 && scratch && n > 0);
    hipStream_t st = (hipStream_t)stream;
    double *part = (double *)scratch;
    const int nb = (int)(n >= (1 << 18) ? 512 : (n + 1023) / 1024 > 0 ? (n + 1023) / 1024 : 1);
    k_sum_partial<<<nb, 256, 0, st>>>(x, n, nullptr, part);
    k_sum_final<<<1, 256, 0, st>>>(part, nb, n, 0, out2);
    k_sum_partial<<<nb, 256, 0, st>>>(x, n, out2, part);
    k_sum_final<<<1, 256, 0, st>>>(part, nb, n, 1, out2 + 1);
    return csm::check_launch("k_mean_std");
}
extern "C" size_t csm_mean_std_scratch_bytes(void) { return 512 * sizeof(double); }

extern "C" int csm_normalise_mean_std(const float *x, int64_t n, const float *mean_std_dev, float *out, void *stream) {
    CSM_REQUIRE(x && mean_std_dev && out && n > 0);
    k_normalise_ms<<<csm::cdiv(n, 256), 256, 0, (hipStream_t)stream>>>(x, n, mean_std_dev, out);
    return csm::check_launch("k_normalise_ms");
}
extern "C" int csm_denormalise_mean_std(const float *x, int64_t n, const float *mean_std_dev, int mode, float *out, void *stream) {
    CSM_REQUIRE(x && mean_std_dev && out && n > 0 && mode >= 0 && mode <= 2);
    k_denormalise_ms<<<csm::cdiv(n, 256), 256, 0, (hipStream_t)stream>>>(x, n, mean_std_dev, mode, out);
    return csm::check_launch("k_denormalise_ms");
}

extern "C" int csm_resize_bilinear_planes(const float *in, int planes, int H, int W, int h, int w, int align_corners, float *out, void *stream) {
    CSM_REQUIRE(in && out && planes > 0 && H > 0 && W > 0 && h > 0 && w > 0);
    float sh, sw;
    if (align_corners) { sh = h > 1 ? (float)(H - 1) / (float)(h - 1) : 0.0f; sw = w > 1 ? (float)(W - 1) / (float)(w - 1) : 0.0f; }
    else { sh = (float)H / (float)h; sw = (float)W / (float)w; }
    k_bilinear_planes<<<csm::cdiv((int64_t)planes * h * w, 256), 256, 0, (hipStream_t)stream>>>(in, planes, H, W, h, w, align_corners, sh, sw, out);
    return csm::check_launch("k_bilinear_planes");
}

extern "C" int csm_mask_area_resize_threshold(const uint8_t *masks, int n, int H, int W, int h, int w, float thr, uint8_t *out, void *stream) {
    CSM_REQUIRE(masks && out && n > 0 && H > 0 && W > 0 && h > 0 && w > 0 && (int64_t)H * h < (1 << 24) && (int64_t)W * w < (1 << 24));
    k_mask_area_threshold<<<csm::cdiv((int64_t)n * h * w, 256), 256, 0, (hipStream_t)stream>>>(masks, n, H, W, h, w, thr, out);
    return csm::check_launch("k_mask_area_threshold");
}
