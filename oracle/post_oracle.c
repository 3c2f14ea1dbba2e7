/*
 * oracle/post_oracle.c  --  TEST INFRASTRUCTURE ONLY.
 * Sequential CPU restatement of the instance-segmentation post-processing:
 *   nms          : mmcv nms semantics [EXT mmcv 2.1.0 ops/csrc nms: devIoU inter > thr*(Sa+Sb-inter)], greedy in score order
 *   maskhead     : animeinsseg/models/rtmdet_inshead_custom.py:253-303 (+ mmdet parse_dynamic_params split 80/64/8 | 8/8/1)
 *   mask_resize  : mmdet _bbox_mask_post_process, mirrored at animeinsseg/__init__.py:361-370
 *   refine_batch : animeinsseg/__init__.py:37-55, utils/io_utils.py:254-292 (cv2.resize INTER_LINEAR restated [EXT OpenCV 4.10])
 *   refine_thr   : animeinsseg/__init__.py:653-662
 * Parity pin: the bilinear index rules are aten's (verified against torch.nn.functional.interpolate in
 * tests/test_oracle_post.py); the mask head against a torch restatement of the vendored reference function in the
 * same test.  mmdet/mmcv/cv2 themselves are absent => those parts are "parity unpinned" (DESIGN.md).
 */
#include <math.h>
#include <stdint.h>
#include <string.h>

static float orc_expf(float x)
{
    x = fminf(fmaxf(x, -87.0f), 88.0f);
    float n = rintf(x * 1.44269504088896341f);
    float r = fmaf(n, -0.693145751953125f, x);
    r = fmaf(n, -1.42860682030941723212e-6f, r);
    float p = 1.9875691500e-4f;
    p = fmaf(p, r, 1.3981999507e-3f);
    p = fmaf(p, r, 8.3334519073e-3f);
    p = fmaf(p, r, 4.1665795894e-2f);
    p = fmaf(p, r, 1.6666665459e-1f);
    p = fmaf(p, r, 5.0000001201e-1f);
    float e = fmaf(p, r * r, r) + 1.0f;
    int32_t bits = ((int32_t)n + 127) << 23; float s; memcpy(&s, &bits, 4);
    return e * s;
}
static float orc_sigmoid(float v) { return 1.0f / (1.0f + orc_expf(-v)); }
float orc_sigmoid_scalar(float v) { return orc_sigmoid(v); }

static void src_index(int dst, int in_size, int out_size, float scale, int align, int *i0, int *i1, float *l0, float *l1)
{
    if (in_size == out_size) { *i0 = *i1 = dst; *l0 = 1.0f; *l1 = 0.0f; return; }
    float real;
    if (align) real = scale * (float)dst;
    else { real = scale * ((float)dst + 0.5f) - 0.5f; if (real < 0.0f) real = 0.0f; }
    *i0 = (int)real < in_size - 1 ? (int)real : in_size - 1;
    *i1 = *i0 + (*i0 < in_size - 1 ? 1 : 0);
    *l1 = fminf(fmaxf(real - (float)*i0, 0.0f), 1.0f);
    *l0 = 1.0f - *l1;
}

int orc_nms(const float *boxes, const float *off, int n, float thr, int max_keep, int *keep)
{
    static uint8_t dead[4096];
    memset(dead, 0, sizeof(dead));
    int cnt = 0;
    for (int i = 0; i < n; ++i) {
        if (dead[i]) continue;
        if (cnt < max_keep) keep[cnt] = i;
        ++cnt;
        float ao = off ? off[i] : 0.0f;
        float ax1 = boxes[i * 4] + ao, ay1 = boxes[i * 4 + 1] + ao, ax2 = boxes[i * 4 + 2] + ao, ay2 = boxes[i * 4 + 3] + ao;
        for (int j = i + 1; j < n; ++j) {
            float bo = off ? off[j] : 0.0f;
            float bx1 = boxes[j * 4] + bo, by1 = boxes[j * 4 + 1] + bo, bx2 = boxes[j * 4 + 2] + bo, by2 = boxes[j * 4 + 3] + bo;
            float left = fmaxf(ax1, bx1), right = fminf(ax2, bx2), top = fmaxf(ay1, by1), bottom = fminf(ay2, by2);
            float width = fmaxf(right - left, 0.0f), height = fmaxf(bottom - top, 0.0f);
            float inter = width * height;
            float sa = (ax2 - ax1) * (ay2 - ay1), sb = (bx2 - bx1) * (by2 - by1);
            if (inter > thr * (sa + sb - inter)) dead[j] = 1;
        }
    }
    return cnt < max_keep ? cnt : max_keep;
}

void orc_maskhead_logits(const float *feat, int ld, int h, int w, const float *params, const float *priors, int n,
                         int feat_stride, float *logits)
{
    enum { P = 8, D = 8, G = 169 };
    for (int inst = 0; inst < n; ++inst) {
        const float *sp = params + (int64_t)inst * G;
        const float *w0 = sp, *w1 = sp + (P + 2) * D, *w2 = w1 + D * D, *b0 = w2 + D, *b1 = b0 + D, *b2 = b1 + D;
        float px = priors[inst * 4], py = priors[inst * 4 + 1], den = priors[inst * 4 + 2] * 8.0f;
        for (int y = 0; y < h; ++y)
            for (int x = 0; x < w; ++x) {
                float in[P + 2], h0[D], h1[D];
                in[0] = (px - (float)(x * feat_stride)) / den;             /* :268-275 */
                in[1] = (py - (float)(y * feat_stride)) / den;
                for (int c = 0; c < P; ++c) in[2 + c] = feat[((int64_t)y * w + x) * ld + c];
                for (int o = 0; o < D; ++o) { float a = b0[o]; for (int c = 0; c < P + 2; ++c) a = fmaf(in[c], w0[o * (P + 2) + c], a); h0[o] = fmaxf(a, 0.0f); }
                for (int o = 0; o < D; ++o) { float a = b1[o]; for (int c = 0; c < D; ++c) a = fmaf(h0[c], w1[o * D + c], a); h1[o] = fmaxf(a, 0.0f); }
                float a = b2[0]; for (int c = 0; c < D; ++c) a = fmaf(h1[c], w2[c], a);
                logits[((int64_t)inst * h + y) * w + x] = a;
            }
    }
}

static float bilerp(const float *L, int w, int y0, int y1, int x0, int x1, float hl0, float hl1, float wl0, float wl1)
{
    return hl0 * (wl0 * L[y0 * w + x0] + wl1 * L[y0 * w + x1]) + hl1 * (wl0 * L[y1 * w + x0] + wl1 * L[y1 * w + x1]);
}

void orc_mask_resize_threshold(const float *logits, int n, int h, int w, int s, int rh, int rw, int oh, int ow, float thr,
                               uint8_t *out)
{
    int Sh = h * s, Sw = w * s;
    float sc1 = 1.0f / (float)s, sh2 = (float)Sh / (float)rh, sw2 = (float)Sw / (float)rw;
#pragma omp parallel for collapse(2)
    for (int inst = 0; inst < n; ++inst)
        for (int oy = 0; oy < oh; ++oy) {
            const float *L = logits + (int64_t)inst * h * w;
            int Y0, Y1, ya0, ya1, yb0, yb1; float HL0, HL1, hla0, hla1, hlb0, hlb1;
            src_index(oy, Sh, rh, sh2, 0, &Y0, &Y1, &HL0, &HL1);
            src_index(Y0, h, Sh, sc1, 0, &ya0, &ya1, &hla0, &hla1);
            src_index(Y1, h, Sh, sc1, 0, &yb0, &yb1, &hlb0, &hlb1);
            for (int ox = 0; ox < ow; ++ox) {
                int X0, X1, xa0, xa1, xb0, xb1; float WL0, WL1, wla0, wla1, wlb0, wlb1;
                src_index(ox, Sw, rw, sw2, 0, &X0, &X1, &WL0, &WL1);
                src_index(X0, w, Sw, sc1, 0, &xa0, &xa1, &wla0, &wla1);
                src_index(X1, w, Sw, sc1, 0, &xb0, &xb1, &wlb0, &wlb1);
                float v00 = bilerp(L, w, ya0, ya1, xa0, xa1, hla0, hla1, wla0, wla1);
                float v01 = bilerp(L, w, ya0, ya1, xb0, xb1, hla0, hla1, wlb0, wlb1);
                float v10 = bilerp(L, w, yb0, yb1, xa0, xa1, hlb0, hlb1, wla0, wla1);
                float v11 = bilerp(L, w, yb0, yb1, xb0, xb1, hlb0, hlb1, wlb0, wlb1);
                float v = HL0 * (WL0 * v00 + WL1 * v01) + HL1 * (WL0 * v10 + WL1 * v11);
                out[((int64_t)inst * oh + oy) * ow + ox] = orc_sigmoid(v) > thr ? 1 : 0;
            }
        }
}

void orc_refine_threshold(const float *logits, int n, int S_h, int S_w, int ch, int cw, int oh, int ow, float thr, uint8_t *out)
{
    float sh = oh > 1 ? (float)(ch - 1) / (float)(oh - 1) : 0.0f, sw = ow > 1 ? (float)(cw - 1) / (float)(ow - 1) : 0.0f;
#pragma omp parallel for collapse(2)
    for (int inst = 0; inst < n; ++inst)
        for (int oy = 0; oy < oh; ++oy) {
            const float *L = logits + (int64_t)inst * S_h * S_w;
            int y0, y1; float hl0, hl1;
            src_index(oy, ch, oh, sh, 1, &y0, &y1, &hl0, &hl1);
            for (int ox = 0; ox < ow; ++ox) {
                int x0, x1; float wl0, wl1;
                src_index(ox, cw, ow, sw, 1, &x0, &x1, &wl0, &wl1);
                float p00 = orc_sigmoid(L[y0 * S_w + x0]), p01 = orc_sigmoid(L[y0 * S_w + x1]);
                float p10 = orc_sigmoid(L[y1 * S_w + x0]), p11 = orc_sigmoid(L[y1 * S_w + x1]);
                float v = hl0 * (wl0 * p00 + wl1 * p01) + hl1 * (wl0 * p10 + wl1 * p11);
                out[((int64_t)inst * oh + oy) * ow + ox] = v > thr ? 1 : 0;
            }
        }
}

static void cv_src(int d, int in_size, double scale, int *i0, int *i1, float *f)
{
    float fx = (float)((d + 0.5) * scale - 0.5);
    int sx = (int)floorf(fx);
    fx -= (float)sx;
    if (sx < 0) { fx = 0.0f; sx = 0; }
    if (sx >= in_size - 1) { fx = 0.0f; sx = in_size - 1; }
    *i0 = sx; *i1 = sx + 1 < in_size - 1 ? sx + 1 : in_size - 1; *f = fx;
}

void orc_refine_prepare_batch(const uint8_t *img, const uint8_t *masks, int n, int H, int W, int rh, int rw, int Hm, int Wm,
                              int rhm, int rwm, int T, float *out)
{
    /* image channels: resize_pad(img) ; mask channel: resize_pad(seg) with the seg's own shape (animeinsseg/__init__.py:39,47) */
    int64_t plane = (int64_t)T * T;
    double sy = (double)H / rh, sx = (double)W / rw;
    double sym = (double)Hm / rhm, sxm = (double)Wm / rwm;
    for (int inst = 0; inst < n; ++inst)
        for (int y = 0; y < T; ++y)
            for (int x = 0; x < T; ++x) {
                float *O = out + (int64_t)inst * 4 * plane + (int64_t)y * T + x;
                const uint8_t *M = masks + (int64_t)inst * Hm * Wm;
                int y0, y1, x0, x1; float fy, fx;
                if (y >= rh || x >= rw) { O[0] = O[plane] = O[2 * plane] = 0.0f; }
                else if (rh == H && rw == W) {
                    for (int c = 0; c < 3; ++c) O[c * plane] = (float)img[((int64_t)y * W + x) * 3 + c] / 255.0f;
                } else {
                    cv_src(y, H, sy, &y0, &y1, &fy); cv_src(x, W, sx, &x0, &x1, &fx);
                    int a0 = (int)rintf((1.0f - fx) * 2048.0f), a1 = (int)rintf(fx * 2048.0f);
                    int b0 = (int)rintf((1.0f - fy) * 2048.0f), b1 = (int)rintf(fy * 2048.0f);
                    for (int c = 0; c < 3; ++c) {
                        int r0 = img[((int64_t)y0 * W + x0) * 3 + c] * a0 + img[((int64_t)y0 * W + x1) * 3 + c] * a1;
                        int r1 = img[((int64_t)y1 * W + x0) * 3 + c] * a0 + img[((int64_t)y1 * W + x1) * 3 + c] * a1;
                        int v = (((b0 * (r0 >> 4)) >> 16) + ((b1 * (r1 >> 4)) >> 16) + 2) >> 2;
                        v = v < 0 ? 0 : (v > 255 ? 255 : v);
                        O[c * plane] = (float)v / 255.0f;
                    }
                }
                if (y >= rhm || x >= rwm) { O[3 * plane] = 0.0f; continue; }
                if (rhm == Hm && rwm == Wm) { O[3 * plane] = (float)M[(int64_t)y * Wm + x]; continue; }
                cv_src(y, Hm, sym, &y0, &y1, &fy); cv_src(x, Wm, sxm, &x0, &x1, &fx);
                float m00 = M[(int64_t)y0 * Wm + x0], m01 = M[(int64_t)y0 * Wm + x1], m10 = M[(int64_t)y1 * Wm + x0], m11 = M[(int64_t)y1 * Wm + x1];
                float r0 = m00 * (1.0f - fx) + m01 * fx, r1 = m10 * (1.0f - fx) + m11 * fx;
                O[3 * plane] = r0 * (1.0f - fy) + r1 * fy;
            }
}

/* mmdet Resize(keep_ratio, cv2 INTER_LINEAR) + Pad(114) + DetDataPreprocessor normalise [EXT]; see maskhead.hip */
void orc_det_preprocess(const uint8_t *img, int H, int W, int rh, int rw, int S_h, int S_w, const float *mean,
                        const float *stdv, float pad, float *out)
{
    int64_t plane = (int64_t)S_h * S_w;
    double sy = (double)H / rh, sx = (double)W / rw;
    for (int y = 0; y < S_h; ++y)
        for (int x = 0; x < S_w; ++x) {
            float v[3];
            if (y >= rh || x >= rw) { v[0] = v[1] = v[2] = pad; }
            else if (rh == H && rw == W) { for (int c = 0; c < 3; ++c) v[c] = (float)img[((int64_t)y * W + x) * 3 + c]; }
            else {
                int y0, y1, x0, x1; float fy, fx;
                cv_src(y, H, sy, &y0, &y1, &fy); cv_src(x, W, sx, &x0, &x1, &fx);
                int a0 = (int)rintf((1.0f - fx) * 2048.0f), a1 = (int)rintf(fx * 2048.0f);
                int b0 = (int)rintf((1.0f - fy) * 2048.0f), b1 = (int)rintf(fy * 2048.0f);
                for (int c = 0; c < 3; ++c) {
                    int r0 = img[((int64_t)y0 * W + x0) * 3 + c] * a0 + img[((int64_t)y0 * W + x1) * 3 + c] * a1;
                    int r1 = img[((int64_t)y1 * W + x0) * 3 + c] * a0 + img[((int64_t)y1 * W + x1) * 3 + c] * a1;
                    int q = (((b0 * (r0 >> 4)) >> 16) + ((b1 * (r1 >> 4)) >> 16) + 2) >> 2;
                    v[c] = (float)(q < 0 ? 0 : (q > 255 ? 255 : q));
                }
            }
            for (int c = 0; c < 3; ++c) out[c * plane + (int64_t)y * S_w + x] = (v[c] - mean[c]) / stdv[c];
        }
}

/* ---- image plumbing around LeReS and the frame tail (mirrors imageops.hip; OpenCV semantics restated [EXT]) ---- */
static void cv_src_area(int d, int in_size, double scale, int *i0, int *i1, float *f)
{
    int sx = (int)floor(d * scale);
    float fx = (float)((d + 1) - (sx + 1) * (1.0 / scale));
    fx = fx <= 0.0f ? 0.0f : fx - floorf(fx);
    if (sx < 0) { fx = 0.0f; sx = 0; }
    if (sx >= in_size - 1) { fx = 0.0f; sx = in_size - 1; }
    *i0 = sx; *i1 = sx + 1 < in_size - 1 ? sx + 1 : in_size - 1; *f = fx;
}
static int cv_lin_u8(int p00, int p01, int p10, int p11, float fx, float fy)
{
    int a0 = (int)rintf((1.0f - fx) * 2048.0f), a1 = (int)rintf(fx * 2048.0f);
    int b0 = (int)rintf((1.0f - fy) * 2048.0f), b1 = (int)rintf(fy * 2048.0f);
    int r0 = p00 * a0 + p01 * a1, r1 = p10 * a0 + p11 * a1;
    int q = (((b0 * (r0 >> 4)) >> 16) + ((b1 * (r1 >> 4)) >> 16) + 2) >> 2;
    return q < 0 ? 0 : (q > 255 ? 255 : q);
}

/* kenburns_effect.py:563-571 + leres/depthmap.py:16-38 */
/* cv2.resize(u8 HWC, INTER_LINEAR)  -- utils/io_utils.py:254-274 */
void orc_resize_u8_linear(const uint8_t *src, int H, int W, int C, int h, int w, uint8_t *dst)
{
    double sy = (double)H / h, sx = (double)W / w;
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            int y0, y1, x0, x1; float fy, fx;
            cv_src(y, H, sy, &y0, &y1, &fy); cv_src(x, W, sx, &x0, &x1, &fx);
            for (int c = 0; c < C; ++c)
                dst[((int64_t)y * w + x) * C + c] = (uint8_t)cv_lin_u8(src[((int64_t)y0 * W + x0) * C + c], src[((int64_t)y0 * W + x1) * C + c],
                                                                      src[((int64_t)y1 * W + x0) * C + c], src[((int64_t)y1 * W + x1) * C + c], fx, fy);
        }
}

/* cv2.resize(float32 HWC, INTER_LINEAR) -- utils/io_utils.py:254-292 on float masks (animeinsseg/__init__.py:47) [EXT: OpenCV
 * resize.cpp float path restated: HResizeLinear then VResizeLinear in fp32] */
void orc_resize_f32_linear(const float *src, int H, int W, int C, int h, int w, float *dst)
{
    double sy = (double)H / h, sx = (double)W / w;
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            int y0, y1, x0, x1; float fy, fx;
            cv_src(y, H, sy, &y0, &y1, &fy); cv_src(x, W, sx, &x0, &x1, &fx);
            float a0 = 1.0f - fx, a1 = fx, b0 = 1.0f - fy, b1 = fy;
            for (int c = 0; c < C; ++c) {
                float r0 = src[((int64_t)y0 * W + x0) * C + c] * a0 + src[((int64_t)y0 * W + x1) * C + c] * a1;
                float r1 = src[((int64_t)y1 * W + x0) * C + c] * a0 + src[((int64_t)y1 * W + x1) * C + c] * a1;
                dst[((int64_t)y * w + x) * C + c] = r0 * b0 + r1 * b1;
            }
        }
}

/* cv2.resize(u8, INTER_LANCZOS4) -> float32 [EXT: OpenCV 4.10 resize.cpp restated, see imageops.hip] */
static void orc_lanczos4_q11(float x, int c[8])
{
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

void orc_resize_u8_lanczos4_to_f32(const uint8_t *src, int h, int w, int H, int W, float *out)
{
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            float fx = (float)((x + 0.5) * ((double)w / W) - 0.5), fy = (float)((y + 0.5) * ((double)h / H) - 0.5);
            int sx = (int)floorf(fx), sy = (int)floorf(fy);
            fx -= (float)sx; fy -= (float)sy;
            int cx[8], cy[8];
            orc_lanczos4_q11(fx, cx); orc_lanczos4_q11(fy, cy);
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
            out[(int64_t)y * W + x] = (float)(v < 0 ? 0 : (v > 255 ? 255 : v));
        }
}

void orc_leres_input(const uint8_t *img, int H, int W, int h, int w, float *out)
{
    const float mean[3] = {0.485f, 0.456f, 0.406f}, stdv[3] = {0.229f, 0.224f, 0.225f};
    int64_t plane = (int64_t)h * w;
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            int y0 = y, y1 = y, x0 = x, x1 = x; float fy = 0.0f, fx = 0.0f;
            int same = (h == H && w == W);
            if (!same) { cv_src(y, H, (double)H / h, &y0, &y1, &fy); cv_src(x, W, (double)W / w, &x0, &x1, &fx); }
            for (int c = 0; c < 3; ++c) {
                int sc = 2 - c;
                int q = same ? img[((int64_t)y * W + x) * 3 + sc]
                             : cv_lin_u8(img[((int64_t)y0 * W + x0) * 3 + sc], img[((int64_t)y0 * W + x1) * 3 + sc],
                                         img[((int64_t)y1 * W + x0) * 3 + sc], img[((int64_t)y1 * W + x1) * 3 + sc], fx, fy);
                float v = (float)q / 255.0f;
                out[c * plane + (int64_t)y * w + x] = (v - mean[c]) / stdv[c];
            }
        }
}

/* depth_modules/leres/__init__.py:121-145 */
void orc_leres_quantize(const float *d, int64_t n, float mn, float mx, uint8_t *out)
{
    for (int64_t i = 0; i < n; ++i) {
        float o = 0.0f;
        if ((double)(mx - mn) > 2.220446049250313e-16) o = 65535.0f * (d[i] - mn) / (mx - mn);
        uint16_t u16 = (uint16_t)o;
        float s = (float)u16 * (float)(255.0 / 65535.0);
        int v = (int)rintf(fabsf(s));
        v = v > 255 ? 255 : v;
        out[i] = (uint8_t)(255 - v);
    }
}

/* kenburns_effect.py:572-575 (cv2.resize INTER_AREA, enlarging) */
void orc_resize_u8_to_f32(const uint8_t *src, int h, int w, int H, int W, float *out)
{
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            int q;
            if (h == H && w == W) q = src[(int64_t)y * w + x];
            else {
                int y0, y1, x0, x1; float fy, fx;
                cv_src_area(y, h, (double)h / H, &y0, &y1, &fy); cv_src_area(x, w, (double)w / W, &x0, &x1, &fx);
                q = cv_lin_u8(src[(int64_t)y0 * w + x0], src[(int64_t)y0 * w + x1], src[(int64_t)y1 * w + x0], src[(int64_t)y1 * w + x1], fx, fy);
            }
            out[(int64_t)y * W + x] = (float)q;
        }
}

static int subpix(const uint8_t *f, int H, int W, int c, int px, int py, int ix, int iy, float a, float b)
{
    int x0 = ix + px, y0 = iy + py;
    float v[4]; int k = 0;
    for (int dy = 0; dy < 2; ++dy)
        for (int dx = 0; dx < 2; ++dx) {
            int yy = y0 + dy, xx = x0 + dx;
            yy = yy < 0 ? 0 : (yy >= H ? H - 1 : yy); xx = xx < 0 ? 0 : (xx >= W ? W - 1 : xx);
            v[k++] = (float)f[((int64_t)yy * W + xx) * 3 + c];
        }
    float a11 = (1.0f - a) * (1.0f - b), a12 = a * (1.0f - b), a21 = (1.0f - a) * b, a22 = a * b;
    float r = v[0] * a11 + v[1] * a12 + v[2] * a21 + v[3] * a22;
    int q = (int)rintf(r);
    return q < 0 ? 0 : (q > 255 ? 255 : q);
}

/* kenburns_effect.py:1069-1070 */
void orc_crop_resize_u8(const uint8_t *frame, int H, int W, int ph, int pw, float cx, float cy, uint8_t *out)
{
    float ox = cx - (float)(pw - 1) * 0.5f, oy = cy - (float)(ph - 1) * 0.5f;
    int ix = (int)floorf(ox), iy = (int)floorf(oy);
    float a = ox - (float)ix, b = oy - (float)iy;
    int same = (ph == H && pw == W);
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            int y0 = y, y1 = y, x0 = x, x1 = x; float fy = 0.0f, fx = 0.0f;
            if (!same) { cv_src(y, ph, (double)ph / H, &y0, &y1, &fy); cv_src(x, pw, (double)pw / W, &x0, &x1, &fx); }
            for (int c = 0; c < 3; ++c) {
                int q = same ? subpix(frame, H, W, c, x, y, ix, iy, a, b)
                             : cv_lin_u8(subpix(frame, H, W, c, x0, y0, ix, iy, a, b), subpix(frame, H, W, c, x1, y0, ix, iy, a, b),
                                         subpix(frame, H, W, c, x0, y1, ix, iy, a, b), subpix(frame, H, W, c, x1, y1, ix, iy, a, b), fx, fy);
                out[((int64_t)y * W + x) * 3 + c] = (uint8_t)q;
            }
        }
}

/* kernel_bokeh  (utils/effects.py:16-74), sequential */
void orc_bokeh_pass(const float *img, const float *depth, float *out, int H, int W, int nsamples, float dx, float dy)
{
    int im_size = H < W ? H : W, off = nsamples / 2;
    for (int64_t pix = 0; pix < (int64_t)H * W; ++pix) {
        int y = (int)(pix / W), x = (int)(pix - (int64_t)y * W);
        float d = depth[pix], ddx = dx * d, ddy = dy * d;
        for (int c = 0; c < 3; ++c) {
            float weight = 0.0f, color = 0.0f;
            for (int s = 0; s < nsamples; ++s) {
                int sp = (s - off) * im_size;
                int x_ = x + (int)roundf(ddx * (float)sp), y_ = y + (int)roundf(ddy * (float)sp);
                if (x_ >= W || y_ >= H || x_ < 0 || y_ < 0) continue;
                float w_ = depth[(int64_t)y_ * W + x_];
                weight += w_;
                color += img[((int64_t)y_ * W + x_) * 3 + c] * w_;
            }
            out[pix * 3 + c] = weight != 0.0f ? color / weight : img[pix * 3 + c];
        }
    }
}
