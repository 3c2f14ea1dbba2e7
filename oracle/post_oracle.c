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

void orc_refine_prepare_batch(const uint8_t *img, const uint8_t *masks, int n, int H, int W, int rh, int rw, int T, float *out)
{
    int64_t plane = (int64_t)T * T;
    double sy = (double)H / rh, sx = (double)W / rw;
    for (int inst = 0; inst < n; ++inst)
        for (int y = 0; y < T; ++y)
            for (int x = 0; x < T; ++x) {
                float *O = out + (int64_t)inst * 4 * plane + (int64_t)y * T + x;
                const uint8_t *M = masks + (int64_t)inst * H * W;
                if (y >= rh || x >= rw) { O[0] = O[plane] = O[2 * plane] = O[3 * plane] = 0.0f; continue; }
                if (rh == H && rw == W) {
                    for (int c = 0; c < 3; ++c) O[c * plane] = (float)img[((int64_t)y * W + x) * 3 + c] / 255.0f;
                    O[3 * plane] = (float)M[(int64_t)y * W + x];
                    continue;
                }
                int y0, y1, x0, x1; float fy, fx;
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
                float m00 = M[(int64_t)y0 * W + x0], m01 = M[(int64_t)y0 * W + x1], m10 = M[(int64_t)y1 * W + x0], m11 = M[(int64_t)y1 * W + x1];
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
