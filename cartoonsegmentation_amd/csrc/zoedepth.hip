// zoedepth.hip -- the parts of `depth_est: 'zoe'` (anime_3dkenburns/kenburns_effect.py:540-544, :812-818) that surround the metric-bins
// head (nets.hip, CSM_OP_ATTRACTOR / CSM_OP_LOGBINOM) and the un-vendored MiDaS core:
//   * DepthModel._infer_with_pad_aug / infer_with_flip_aug (depth_modules/zoedepth/models/depth_model.py:57-113): reflect padding
//     by sqrt(size / 2) * 3, the horizontal-flip pass, bicubic resize back to the padded size, crop, average of the two passes;
//   * MidasCore.forward's PrepForMidas (models/base_models/midas.py:49-187): bilinear align_corners=True resize to the
//     multiple-of-32 size chosen by Resize.get_size, Normalize(mean 0.5, std 0.5);
//   * the depth -> disparity step of _depth_est_zoe (kenburns_effect.py:815-817).
// All HBM-bound elementwise / gather kernels (3 to 12 floats per output element); padding, flip, resize and normalisation are ONE
// pass (the padded image never exists in memory), resize-back, crop, un-flip and the average are one pass per TTA branch.
#include "csm_common.h"

namespace {

// torch.nn.functional.pad(mode='reflect') index: -1 -> 1, n -> n - 2 (pad < n, asserted by the host)
__device__ __forceinline__ int reflect(int i, int n) {
    if (i < 0) i = -i;
    if (i >= n) i = 2 * (n - 1) - i;
    return i;
}

// aten upsample_bilinear2d, align_corners = true (UpSample.h: area_pixel_compute_source_index)
__device__ __forceinline__ void src_ac(int dst, int in_size, float scale, int &i0, int &i1, float &l0, float &l1) {
    const float real = scale * (float)dst;
    i0 = min((int)real, in_size - 1);
    i1 = i0 + (i0 < in_size - 1 ? 1 : 0);
    l1 = fminf(fmaxf(real - (float)i0, 0.0f), 1.0f);
    l0 = 1.0f - l1;
}

// out[b, c, y, x] = (bilinear_ac(pad_reflect(flip?(img)))[y, x] - 0.5) / 0.5
__global__ __launch_bounds__(256) void k_zoe_pad_prep(const float *__restrict__ img, int B, int H, int W, int pad_h, int pad_w, int flip,
                                                       int nh, int nw, float sh, float sw, int same, float *__restrict__ out) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x, total = (int64_t)B * 3 * nh * nw;
    if (idx >= total) return;
    const int x = (int)(idx % nw); int64_t t = idx / nw; const int y = (int)(t % nh); const int64_t bc = t / nh;
    const int Hp = H + 2 * pad_h, Wp = W + 2 * pad_w;
    const float *P = img + bc * (int64_t)H * W;
    auto at = [&](int py, int px) {                      // padded-image sample -> source pixel
        int sy = reflect(py - pad_h, H), sx = reflect(px - pad_w, W);
        if (flip) sx = W - 1 - sx;                       // flip(x) then symmetric reflect padding == padding then flip
        return P[(int64_t)sy * W + sx];
    };
    float v;
    if (same) v = at(y, x);                              // nn.functional.interpolate returns the input values when sizes match
    else {
        int y0, y1, x0, x1; float hl0, hl1, wl0, wl1;
        src_ac(y, Hp, sh, y0, y1, hl0, hl1); src_ac(x, Wp, sw, x0, x1, wl0, wl1);
        v = hl0 * (wl0 * at(y0, x0) + wl1 * at(y0, x1)) + hl1 * (wl0 * at(y1, x0) + wl1 * at(y1, x1));
    }
    out[idx] = (v - 0.5f) / 0.5f;                        // torchvision Normalize: sub mean, div std
}

// aten upsample_bicubic2d (A = -0.75, align_corners = false, border-clamped taps)
__device__ __forceinline__ float cc1(float x, float A) { return ((A + 2.0f) * x - (A + 3.0f)) * x * x + 1.0f; }
__device__ __forceinline__ float cc2(float x, float A) { return ((A * x - 5.0f * A) * x + 8.0f * A) * x - 4.0f * A; }
__device__ __forceinline__ void cubic_coeffs(float t, float c[4]) {
    const float A = -0.75f;
    c[0] = cc2(t + 1.0f, A); c[1] = cc1(t, A);
    const float u = 1.0f - t;
    c[2] = cc1(u, A); c[3] = cc2(u + 1.0f, A);
}

// d [B, 1, h, w] (the head's metric depth at the core's resolution) -> out [B, 1, H, W]: bicubic to the padded size (Hp, Wp) evaluated
// only inside the crop window, optionally mirrored back; mode 0: out = v, mode 1: out = (out + v) / 2 (second TTA branch)
__global__ __launch_bounds__(256) void k_zoe_resize_crop(const float *__restrict__ d, int B, int h, int w, int Hp, int Wp, int pad_h, int pad_w,
                                                          int H, int W, int unflip, int mode, float sh, float sw, float *__restrict__ out) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x, total = (int64_t)B * H * W;
    if (idx >= total) return;
    const int x = (int)(idx % W); int64_t t = idx / W; const int y = (int)(t % H); const int64_t b = t / H;
    const int px = (unflip ? W - 1 - x : x) + pad_w, py = y + pad_h;        // position in the padded (still flipped) prediction
    const float *P = d + b * (int64_t)h * w;
    float v;
    if (h == Hp && w == Wp) v = P[(int64_t)py * w + px];
    else {
        const float rx = sw * ((float)px + 0.5f) - 0.5f, ry = sh * ((float)py + 0.5f) - 0.5f;     // cubic: no clamp at 0
        const int ix = (int)floorf(rx), iy = (int)floorf(ry);
        float cx[4], cy[4];
        cubic_coeffs(rx - (float)ix, cx); cubic_coeffs(ry - (float)iy, cy);
        float rows[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int yy = min(max(iy - 1 + i, 0), h - 1);
            const float *R = P + (int64_t)yy * w;
            const float x0 = R[min(max(ix - 1, 0), w - 1)], x1 = R[min(max(ix, 0), w - 1)];
            const float x2 = R[min(max(ix + 1, 0), w - 1)], x3 = R[min(max(ix + 2, 0), w - 1)];
            rows[i] = x0 * cx[0] + x1 * cx[1] + x2 * cx[2] + x3 * cx[3];
        }
        v = rows[0] * cy[0] + rows[1] * cy[1] + rows[2] * cy[2] + rows[3] * cy[3];
    }
    out[idx] = mode ? (out[idx] + v) / 2.0f : v;
}

// kenburns_effect.py:816-817: disparity = (focal * baseline) / (depth + 0.00001) [python float / Tensor = reciprocal * float];
// nan_to_num_(0, 0, 0)
__global__ __launch_bounds__(256) void k_zoe_disparity(const float *__restrict__ depth, int64_t n, float fb, float *__restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float v = (1.0f / (depth[i] + 0.00001f)) * fb;
    if (isnan(v) || isinf(v)) v = 0.0f;
    out[i] = v;
}

}  // namespace

extern "C" int csm_zoe_pad_prep(const float *img, int B, int H, int W, int pad_h, int pad_w, int flip, int nh, int nw, float *out,
                                void *stream) {
    CSM_REQUIRE(img && out && B > 0 && H > 1 && W > 1 && nh > 0 && nw > 0 && pad_h >= 0 && pad_w >= 0 && pad_h < H && pad_w < W);
    const int Hp = H + 2 * pad_h, Wp = W + 2 * pad_w;
    const float sh = nh > 1 ? (float)(Hp - 1) / (float)(nh - 1) : 0.0f, sw = nw > 1 ? (float)(Wp - 1) / (float)(nw - 1) : 0.0f;
    k_zoe_pad_prep<<<csm::cdiv((int64_t)B * 3 * nh * nw, 256), 256, 0, (hipStream_t)stream>>>(img, B, H, W, pad_h, pad_w, flip, nh, nw, sh, sw,
                                                                                              (nh == Hp && nw == Wp) ? 1 : 0, out);
    return csm::check_launch("k_zoe_pad_prep");
}

extern "C" int csm_zoe_resize_crop(const float *d, int B, int h, int w, int pad_h, int pad_w, int H, int W, int unflip, int mode, float *out,
                                   void *stream) {
    CSM_REQUIRE(d && out && B > 0 && h > 0 && w > 0 && H > 0 && W > 0 && pad_h >= 0 && pad_w >= 0);
    const int Hp = H + 2 * pad_h, Wp = W + 2 * pad_w;
    k_zoe_resize_crop<<<csm::cdiv((int64_t)B * H * W, 256), 256, 0, (hipStream_t)stream>>>(d, B, h, w, Hp, Wp, pad_h, pad_w, H, W, unflip, mode,
                                                                                        (float)h / (float)Hp, (float)w / (float)Wp, out);
    return csm::check_launch("k_zoe_resize_crop");
}

extern "C" int csm_zoe_depth_to_disparity(const float *depth, int64_t n, float focal_times_baseline, float *out, void *stream) {
    CSM_REQUIRE(depth && out && n > 0);
    k_zoe_disparity<<<csm::cdiv(n, 256), 256, 0, (hipStream_t)stream>>>(depth, n, focal_times_baseline, out);
    return csm::check_launch("k_zoe_disparity");
}
