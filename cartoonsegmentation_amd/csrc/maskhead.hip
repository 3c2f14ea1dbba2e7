// maskhead.hip -- instance-segmentation post-processing kernels for gfx950:
//   * class-aware NMS (replaces mmcv._ext nms),
//   * RTMDet-Ins dynamic-conv mask head (animeinsseg/models/rtmdet_inshead_custom.py:253-303),
//   * fused  x8 bilinear -> rescale bilinear -> crop -> sigmoid -> threshold  writing 1 byte/pixel
//     (mmdet _bbox_mask_post_process, mirrored in-repo at animeinsseg/__init__.py:361-370),
//   * ISNet refine glue: image/mask -> [n,4,720,720] batch (animeinsseg/__init__.py:37-55) and
//     sigmoid -> crop -> bilinear(align_corners=True) -> threshold (:653-662).
// All HBM-bound: the only large traffic is the n x H x W byte masks, written once, coalesced.
#include "csm_common.h"

namespace {

__device__ __forceinline__ float csm_expf(float x) {   // same polynomial as nets.hip / DESIGN.md
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
    return e * __int_as_float(((int)n + 127) << 23);
}
__device__ __forceinline__ float csm_sigmoid(float v) { return 1.0f / (1.0f + csm_expf(-v)); }

// aten upsample_bilinear2d source index (same rule as nets.hip)
__device__ __forceinline__ void src_index(int dst, int in_size, int out_size, float scale, bool align, int &i0, int &i1,
                                          float &l0, float &l1) {
    if (in_size == out_size) { i0 = i1 = dst; l0 = 1.0f; l1 = 0.0f; return; }
    float real;
    if (align) real = scale * (float)dst;
    else { real = scale * ((float)dst + 0.5f) - 0.5f; if (real < 0.0f) real = 0.0f; }
    i0 = min((int)real, in_size - 1);
    i1 = i0 + (i0 < in_size - 1 ? 1 : 0);
    l1 = fminf(fmaxf(real - (float)i0, 0.0f), 1.0f);
    l0 = 1.0f - l1;
}

// ---- NMS ------------------------------------------------------------------------------------
// boxes sorted by descending score.  mask[i][w] bit j: box (64w+j) is suppressed by box i (j > i).
// mmcv devIoU: suppress iff inter > thr * (Sa + Sb - inter), box width = x2 - x1 (offset 0).
__global__ __launch_bounds__(64) void k_nms_mask(const float *__restrict__ boxes, const float *__restrict__ cls_off,
                                                  int n, float thr, unsigned long long *__restrict__ mask, int words) {
    int i = blockIdx.x, w = blockIdx.y, j = w * 64 + threadIdx.x;
    float ax1 = boxes[i * 4], ay1 = boxes[i * 4 + 1], ax2 = boxes[i * 4 + 2], ay2 = boxes[i * 4 + 3];
    float ao = cls_off ? cls_off[i] : 0.0f;
    ax1 += ao; ay1 += ao; ax2 += ao; ay2 += ao;
    bool sup = false;
    if (j < n && j > i) {
        float bo = cls_off ? cls_off[j] : 0.0f;
        float bx1 = boxes[j * 4] + bo, by1 = boxes[j * 4 + 1] + bo, bx2 = boxes[j * 4 + 2] + bo, by2 = boxes[j * 4 + 3] + bo;
        float left = fmaxf(ax1, bx1), right = fminf(ax2, bx2), top = fmaxf(ay1, by1), bottom = fminf(ay2, by2);
        float width = fmaxf(right - left, 0.0f), height = fmaxf(bottom - top, 0.0f);
        float inter = width * height;
        float sa = (ax2 - ax1) * (ay2 - ay1), sb = (bx2 - bx1) * (by2 - by1);
        sup = inter > thr * (sa + sb - inter);
    }
    unsigned long long b = __ballot(sup);
    if (threadIdx.x == 0) mask[(int64_t)i * words + w] = b;
}

// one wave scans the boxes in score order; lane w owns word w of the removed-set (n <= 4096)
__global__ __launch_bounds__(64) void k_nms_scan(const unsigned long long *__restrict__ mask, int n, int words,
                                                  int *__restrict__ keep, int *__restrict__ n_keep, int max_keep) {
    unsigned long long removed = 0ull;
    int lane = threadIdx.x, cnt = 0;
    for (int i = 0; i < n && cnt < max_keep; ++i) {   // results[:max_per_img]: later boxes cannot change earlier keeps
        unsigned long long wv = __shfl(removed, i >> 6);
        bool dead = (wv >> (i & 63)) & 1ull;
        if (!dead) {
            if (lane == 0 && cnt < max_keep) keep[cnt] = i;
            ++cnt;
            if (lane < words) removed |= mask[(int64_t)i * words + lane];
        }
    }
    if (lane == 0) *n_keep = cnt < max_keep ? cnt : max_keep;
}

// ---- dynamic-conv mask head ------------------------------------------------------------------
// per instance: x = [rel_x, rel_y, mask_feat(P)] -> (P+2 -> D) relu -> (D -> D) relu -> (D -> 1)
// params [n, G] laid out as mmdet parse_dynamic_params: weights [ (P+2)*D | D*D | D ] then biases [ D | D | 1 ].
template <int P, int D>
__global__ __launch_bounds__(256) void k_maskhead(const float *__restrict__ feat, int ld, int h, int w,
                                                   const float *__restrict__ params, int G,
                                                   const float *__restrict__ priors, int feat_stride,
                                                   float *__restrict__ logits) {
    __shared__ float sp[(P + 2) * D + D * D + D + D + D + 1];
    const int inst = blockIdx.y;
    for (int i = threadIdx.x; i < G; i += 256) sp[i] = params[(int64_t)inst * G + i];
    __syncthreads();
    int pix = blockIdx.x * 256 + threadIdx.x;
    if (pix >= h * w) return;
    int y = pix / w, x = pix - y * w;
    const float px = priors[inst * 4], py = priors[inst * 4 + 1], ps = priors[inst * 4 + 2];
    // coord = MlvlPointGenerator(offset=0) level-0 grid: (x*stride, y*stride); (points - coord) / (stride_inst * 8)
    float in[P + 2];
    float den = ps * 8.0f;
    in[0] = (px - (float)(x * feat_stride)) / den;
    in[1] = (py - (float)(y * feat_stride)) / den;
    const float *F = feat + (int64_t)pix * ld;
#pragma unroll
    for (int c = 0; c < P; ++c) in[2 + c] = F[c];
    const float *w0 = sp, *w1 = sp + (P + 2) * D, *w2 = w1 + D * D;
    const float *b0 = w2 + D, *b1 = b0 + D, *b2 = b1 + D;
    float h0[D], h1[D];
#pragma unroll
    for (int o = 0; o < D; ++o) {
        float a = b0[o];
#pragma unroll
        for (int c = 0; c < P + 2; ++c) a = fmaf(in[c], w0[o * (P + 2) + c], a);
        h0[o] = fmaxf(a, 0.0f);
    }
#pragma unroll
    for (int o = 0; o < D; ++o) {
        float a = b1[o];
#pragma unroll
        for (int c = 0; c < D; ++c) a = fmaf(h0[c], w1[o * D + c], a);
        h1[o] = fmaxf(a, 0.0f);
    }
    float a = b2[0];
#pragma unroll
    for (int c = 0; c < D; ++c) a = fmaf(h1[c], w2[c], a);
    logits[(int64_t)inst * h * w + pix] = a;
}

__device__ __forceinline__ float bilerp(const float *__restrict__ L, int w, int y0, int y1, int x0, int x1, float hl0,
                                        float hl1, float wl0, float wl1) {
    return hl0 * (wl0 * L[y0 * w + x0] + wl1 * L[y0 * w + x1]) + hl1 * (wl0 * L[y1 * w + x0] + wl1 * L[y1 * w + x1]);
}

// logits [n,h,w] -> interpolate(scale_factor=s) [n,S_h,S_w] -> interpolate(size=(rh,rw)) -> [:oh,:ow] -> sigmoid > thr
// out: uint8 [n,oh,ow].  4 pixels per lane, one 32-bit store.
__global__ __launch_bounds__(256) void k_mask_resize_threshold(const float *__restrict__ logits, int h, int w, int s,
                                                                int rh, int rw, int oh, int ow, float thr,
                                                                uint8_t *__restrict__ out) {
    const int inst = blockIdx.z;
    const int oy = blockIdx.y;
    const int ox0 = (blockIdx.x * 256 + threadIdx.x) * 4;
    if (ox0 >= ow) return;
    const float *L = logits + (int64_t)inst * h * w;
    const int Sh = h * s, Sw = w * s;
    const float sc1 = 1.0f / (float)s;                       // scale_factor given -> aten uses 1/scale_factor
    const float sh2 = (float)Sh / (float)rh, sw2 = (float)Sw / (float)rw;
    int Y0, Y1; float HL0, HL1;
    src_index(oy, Sh, rh, sh2, false, Y0, Y1, HL0, HL1);
    // the two S-grid rows each need their own stride-grid rows
    int ya0, ya1, yb0, yb1; float hla0, hla1, hlb0, hlb1;
    src_index(Y0, h, Sh, sc1, false, ya0, ya1, hla0, hla1);
    src_index(Y1, h, Sh, sc1, false, yb0, yb1, hlb0, hlb1);
    uint8_t r[4] = {0, 0, 0, 0};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        int ox = ox0 + j;
        if (ox >= ow) break;
        int X0, X1; float WL0, WL1;
        src_index(ox, Sw, rw, sw2, false, X0, X1, WL0, WL1);
        int xa0, xa1, xb0, xb1; float wla0, wla1, wlb0, wlb1;
        src_index(X0, w, Sw, sc1, false, xa0, xa1, wla0, wla1);
        src_index(X1, w, Sw, sc1, false, xb0, xb1, wlb0, wlb1);
        float v00 = bilerp(L, w, ya0, ya1, xa0, xa1, hla0, hla1, wla0, wla1);   // S-grid (Y0,X0)
        float v01 = bilerp(L, w, ya0, ya1, xb0, xb1, hla0, hla1, wlb0, wlb1);   // (Y0,X1)
        float v10 = bilerp(L, w, yb0, yb1, xa0, xa1, hlb0, hlb1, wla0, wla1);   // (Y1,X0)
        float v11 = bilerp(L, w, yb0, yb1, xb0, xb1, hlb0, hlb1, wlb0, wlb1);   // (Y1,X1)
        float v = HL0 * (WL0 * v00 + WL1 * v01) + HL1 * (WL0 * v10 + WL1 * v11);
        r[j] = csm_sigmoid(v) > thr ? 1 : 0;
    }
    uint8_t *O = out + ((int64_t)inst * oh + oy) * ow + ox0;
    if (ox0 + 3 < ow && ((ow & 3) == 0)) *reinterpret_cast<uint32_t *>(O) = r[0] | (r[1] << 8) | (r[2] << 16) | ((uint32_t)r[3] << 24);
    else for (int j = 0; j < 4 && ox0 + j < ow; ++j) O[j] = r[j];
}

// ISNet refine output: logits [n,1,S,S] -> sigmoid -> crop [:ch,:cw] -> bilinear(align_corners=True) to (oh,ow) -> > thr
__global__ __launch_bounds__(256) void k_refine_threshold(const float *__restrict__ logits, int S_h, int S_w, int ch,
                                                           int cw, int oh, int ow, float thr, uint8_t *__restrict__ out) {
    const int inst = blockIdx.z, oy = blockIdx.y;
    const int ox = blockIdx.x * 256 + threadIdx.x;
    if (ox >= ow) return;
    const float *L = logits + (int64_t)inst * S_h * S_w;
    float sh = oh > 1 ? (float)(ch - 1) / (float)(oh - 1) : 0.0f, sw = ow > 1 ? (float)(cw - 1) / (float)(ow - 1) : 0.0f;
    int y0, y1, x0, x1; float hl0, hl1, wl0, wl1;
    src_index(oy, ch, oh, sh, true, y0, y1, hl0, hl1);
    src_index(ox, cw, ow, sw, true, x0, x1, wl0, wl1);
    float p00 = csm_sigmoid(L[y0 * S_w + x0]), p01 = csm_sigmoid(L[y0 * S_w + x1]);
    float p10 = csm_sigmoid(L[y1 * S_w + x0]), p11 = csm_sigmoid(L[y1 * S_w + x1]);
    float v = hl0 * (wl0 * p00 + wl1 * p01) + hl1 * (wl0 * p10 + wl1 * p11);
    out[((int64_t)inst * oh + oy) * ow + ox] = v > thr ? 1 : 0;
}

// cv2.resize(INTER_LINEAR) source coordinate (half-pixel centres, clamped)  [EXT: OpenCV 4.10 resize.cpp]
__device__ __forceinline__ void cv_src(int d, int in_size, double scale, int &i0, int &i1, float &f) {
    float fx = (float)((d + 0.5) * scale - 0.5);
    int sx = (int)floorf(fx);
    fx -= (float)sx;
    if (sx < 0) { fx = 0.0f; sx = 0; }
    if (sx >= in_size - 1) { fx = 0.0f; sx = in_size - 1; }
    i0 = sx; i1 = min(sx + 1, in_size - 1); f = fx;
}

// prepare_refine_batch (animeinsseg/__init__.py:37-55): img u8 HWC [H,W,3] -> resize_pad to T (keep ratio, pad
// bottom/right with 0) -> /255 -> channels 0..2 ; mask u8 [n,H,W] -> float -> resize (float bilinear) -> channel 3.
// out: [n,4,T,T] NCHW fp32.  rh, rw = resized size (<= T).  uint8 path = cv2's 11-bit fixed point.
// The mask plane has its OWN size (Hm,Wm) and resized extent (rhm,rwm): mmdet's `[..., :ori_h, :ori_w]` slice can leave the
// detector masks 1-2 px smaller than the image, and the reference resize_pad()s each seg by its own shape.
__global__ __launch_bounds__(256) void k_refine_batch(const uint8_t *__restrict__ img, const uint8_t *__restrict__ masks,
                                                       int H, int W, int rh, int rw, int Hm, int Wm, int rhm, int rwm, int T,
                                                       float *__restrict__ out) {
    const int inst = blockIdx.z, y = blockIdx.y, x = blockIdx.x * 256 + threadIdx.x;
    if (x >= T) return;
    float *O = out + (int64_t)inst * 4 * T * T + (int64_t)y * T + x;
    const int64_t plane = (int64_t)T * T;
    if (y >= rh || x >= rw) { O[0] = 0.0f; O[plane] = 0.0f; O[2 * plane] = 0.0f; }
    else if (rh == H && rw == W) {
        for (int c = 0; c < 3; ++c) O[c * plane] = (float)img[((int64_t)y * W + x) * 3 + c] / 255.0f;
    } else {
        double sy = (double)H / rh, sx = (double)W / rw;
        int y0, y1, x0, x1; float fy, fx;
        cv_src(y, H, sy, y0, y1, fy); cv_src(x, W, sx, x0, x1, fx);
        // 8-bit: coefficients in Q11, horizontal pass to int, vertical pass with the >>4 / >>16 / +2 >>2 rounding
        const int a0 = (int)rintf((1.0f - fx) * 2048.0f), a1 = (int)rintf(fx * 2048.0f);
        const int b0 = (int)rintf((1.0f - fy) * 2048.0f), b1 = (int)rintf(fy * 2048.0f);
        for (int c = 0; c < 3; ++c) {
            int r0 = img[((int64_t)y0 * W + x0) * 3 + c] * a0 + img[((int64_t)y0 * W + x1) * 3 + c] * a1;
            int r1 = img[((int64_t)y1 * W + x0) * 3 + c] * a0 + img[((int64_t)y1 * W + x1) * 3 + c] * a1;
            int v = (((b0 * (r0 >> 4)) >> 16) + ((b1 * (r1 >> 4)) >> 16) + 2) >> 2;
            v = v < 0 ? 0 : (v > 255 ? 255 : v);
            O[c * plane] = (float)v / 255.0f;
        }
    }
    const uint8_t *M = masks + (int64_t)inst * Hm * Wm;
    if (y >= rhm || x >= rwm) { O[3 * plane] = 0.0f; return; }
    if (rhm == Hm && rwm == Wm) { O[3 * plane] = (float)M[(int64_t)y * Wm + x]; return; }
    // float32 path: plain fp32 bilinear
    double sy = (double)Hm / rhm, sx = (double)Wm / rwm;
    int y0, y1, x0, x1; float fy, fx;
    cv_src(y, Hm, sy, y0, y1, fy); cv_src(x, Wm, sx, x0, x1, fx);
    float m00 = M[(int64_t)y0 * Wm + x0], m01 = M[(int64_t)y0 * Wm + x1], m10 = M[(int64_t)y1 * Wm + x0], m11 = M[(int64_t)y1 * Wm + x1];
    float r0 = m00 * (1.0f - fx) + m01 * fx, r1 = m10 * (1.0f - fx) + m11 * fx;
    O[3 * plane] = r0 * (1.0f - fy) + r1 * fy;
}

// mmdet test pipeline + DetDataPreprocessor [EXT mmdet 3.3.0 / mmcv 2.1.0; call sites animeinsseg/__init__.py:63-76,
// :212-215]: Resize(keep_ratio, cv2 INTER_LINEAR u8) -> Pad(bottom/right, 114) -> (x - mean) / std, BGR kept.
// img u8 HWC [H,W,3] -> out fp32 NCHW [1,3,S_h,S_w]; (rh,rw) = resized extent.
struct Norm3 { float mean[3], stdv[3]; };
__global__ __launch_bounds__(256) void k_det_preprocess(const uint8_t *__restrict__ img, int H, int W, int rh, int rw,
                                                         int S_h, int S_w, Norm3 nm, float pad, float *__restrict__ out) {
    const int y = blockIdx.y, x = blockIdx.x * 256 + threadIdx.x;
    if (x >= S_w) return;
    const int64_t plane = (int64_t)S_h * S_w;
    float *O = out + (int64_t)y * S_w + x;
    float v[3];
    if (y >= rh || x >= rw) { v[0] = v[1] = v[2] = pad; }
    else if (rh == H && rw == W) { for (int c = 0; c < 3; ++c) v[c] = (float)img[((int64_t)y * W + x) * 3 + c]; }
    else {
        double sy = (double)H / rh, sx = (double)W / rw;
        int y0, y1, x0, x1; float fy, fx;
        cv_src(y, H, sy, y0, y1, fy); cv_src(x, W, sx, x0, x1, fx);
        const int a0 = (int)rintf((1.0f - fx) * 2048.0f), a1 = (int)rintf(fx * 2048.0f);
        const int b0 = (int)rintf((1.0f - fy) * 2048.0f), b1 = (int)rintf(fy * 2048.0f);
        for (int c = 0; c < 3; ++c) {
            int r0 = img[((int64_t)y0 * W + x0) * 3 + c] * a0 + img[((int64_t)y0 * W + x1) * 3 + c] * a1;
            int r1 = img[((int64_t)y1 * W + x0) * 3 + c] * a0 + img[((int64_t)y1 * W + x1) * 3 + c] * a1;
            int q = (((b0 * (r0 >> 4)) >> 16) + ((b1 * (r1 >> 4)) >> 16) + 2) >> 2;
            v[c] = (float)(q < 0 ? 0 : (q > 255 ? 255 : q));
        }
    }
    for (int c = 0; c < 3; ++c) O[c * plane] = (v[c] - nm.mean[c]) / nm.stdv[c];
}

}  // namespace

// np.packbits(mask != 0, bitorder='little') of a byte plane: the form instance masks travel in when a rank ships its outputs to
// rank 0 (3 MB frame + bit-packed masks per frame).  One thread per output byte, 8-byte loads when aligned.
namespace {
__global__ __launch_bounds__(256) void k_pack_bits(const uint8_t *__restrict__ m, int64_t n, uint8_t *__restrict__ out, int aligned) {
    const int64_t o = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (o >= ((n + 7) >> 3)) return;
    const int64_t base = o << 3;
    unsigned v = 0u;
    if (aligned && base + 8 <= n) {
        const unsigned long long w = *reinterpret_cast<const unsigned long long *>(m + base);
#pragma unroll
        for (int b = 0; b < 8; ++b) v |= (((w >> (8 * b)) & 0xffull) ? 1u : 0u) << b;
    } else {
        for (int b = 0; b < 8 && base + b < n; ++b) v |= (m[base + b] ? 1u : 0u) << b;
    }
    out[o] = (uint8_t)v;
}
}  // namespace
extern "C" int csm_pack_mask_bits(const uint8_t *mask, int64_t n, uint8_t *out, void *stream) {
    CSM_REQUIRE(mask && out && n > 0);
    k_pack_bits<<<csm::cdiv((n + 7) >> 3, 256), 256, 0, (hipStream_t)stream>>>(mask, n, out, (((uintptr_t)mask) & 7) == 0 ? 1 : 0);
    return csm::check_launch("k_pack_bits");
}

extern "C" size_t csm_nms_scratch_bytes(int n) { return (size_t)n * ((n + 63) / 64) * 8 + 64; }

extern "C" int csm_nms(const float *boxes, const float *class_offsets, int n, float iou_thr, int max_keep, int *keep,
                       int *n_keep, void *scratch, void *stream) {
    CSM_REQUIRE(keep && n_keep && n >= 0 && n <= 4096 && max_keep > 0);
    hipStream_t st = (hipStream_t)stream;
    if (n == 0) { CSM_HIP(hipMemsetAsync(n_keep, 0, sizeof(int), st)); return CSM_OK; }
    CSM_REQUIRE(boxes && scratch);
    int words = (n + 63) / 64;
    unsigned long long *mask = (unsigned long long *)scratch;
    k_nms_mask<<<dim3(n, words), 64, 0, st>>>(boxes, class_offsets, n, iou_thr, mask, words);
    int rc = csm::check_launch("k_nms_mask"); if (rc) return rc;
    k_nms_scan<<<1, 64, 0, st>>>(mask, n, words, keep, n_keep, max_keep);
    return csm::check_launch("k_nms_scan");
}

extern "C" int csm_maskhead_logits(const float *mask_feat, int ld, int h, int w, int num_prototypes,
                                   int dyconv_channels, const float *kernels, const float *priors, int n,
                                   int feat_stride, float *logits, void *stream) {
    CSM_REQUIRE(mask_feat && logits && n >= 0 && h > 0 && w > 0);
    if (n == 0) return CSM_OK;
    CSM_REQUIRE(kernels && priors);
    if (num_prototypes != 8 || dyconv_channels != 8) return csm::fail_arg("mask head compiled for 8 prototypes x 8 dyconv channels");
    k_maskhead<8, 8><<<dim3(csm::cdiv((int64_t)h * w, 256), n), 256, 0, (hipStream_t)stream>>>(
        mask_feat, ld, h, w, kernels, 169, priors, feat_stride, logits);
    return csm::check_launch("k_maskhead");
}

extern "C" int csm_mask_resize_threshold(const float *logits, int n, int h, int w, int up, int rh, int rw, int oh, int ow,
                                         float thr, uint8_t *masks, void *stream) {
    // oh/ow = min(resized, original): mmdet's `[..., :ori_h, :ori_w]` is a slice, so when ceil(S/scale) lands 1-2 px below the
    // original size the mask is simply that much smaller (the caller clips; ISNet refine resizes it back)
    CSM_REQUIRE(masks && n >= 0 && h > 0 && w > 0 && up > 0 && rh >= oh && rw >= ow && oh > 0 && ow > 0);
    if (n == 0) return CSM_OK;
    CSM_REQUIRE(logits);
    k_mask_resize_threshold<<<dim3(csm::cdiv((ow + 3) / 4, 256), oh, n), 256, 0, (hipStream_t)stream>>>(
        logits, h, w, up, rh, rw, oh, ow, thr, masks);
    return csm::check_launch("k_mask_resize_threshold");
}

extern "C" int csm_refine_prepare_batch(const uint8_t *img_hwc, const uint8_t *masks, int n, int H, int W, int rh, int rw,
                                        int Hm, int Wm, int rhm, int rwm, int T, float *batch, void *stream) {
    CSM_REQUIRE(img_hwc && masks && batch && n > 0 && H > 0 && W > 0 && rh <= T && rw <= T);
    CSM_REQUIRE(Hm > 0 && Wm > 0 && rhm > 0 && rwm > 0 && rhm <= T && rwm <= T && rh > 0 && rw > 0);
    k_refine_batch<<<dim3(csm::cdiv(T, 256), T, n), 256, 0, (hipStream_t)stream>>>(img_hwc, masks, H, W, rh, rw, Hm, Wm, rhm,
                                                                                     rwm, T, batch);
    return csm::check_launch("k_refine_batch");
}

extern "C" int csm_refine_threshold(const float *logits, int n, int S_h, int S_w, int crop_h, int crop_w, int oh, int ow,
                                    float thr, uint8_t *masks, void *stream) {
    CSM_REQUIRE(logits && masks && n > 0 && crop_h <= S_h && crop_w <= S_w && oh > 0 && ow > 0);
    k_refine_threshold<<<dim3(csm::cdiv(ow, 256), oh, n), 256, 0, (hipStream_t)stream>>>(logits, S_h, S_w, crop_h, crop_w,
                                                                                            oh, ow, thr, masks);
    return csm::check_launch("k_refine_threshold");
}

extern "C" int csm_det_preprocess(const uint8_t *img_hwc, int H, int W, int rh, int rw, int S_h, int S_w, const float *mean3,
                                  const float *std3, float pad_value, float *out, void *stream) {
    CSM_REQUIRE(img_hwc && out && mean3 && std3 && H > 0 && W > 0 && rh <= S_h && rw <= S_w && rh > 0 && rw > 0);
    Norm3 nm;
    for (int c = 0; c < 3; ++c) { nm.mean[c] = mean3[c]; nm.stdv[c] = std3[c]; }
    k_det_preprocess<<<dim3(csm::cdiv(S_w, 256), S_h), 256, 0, (hipStream_t)stream>>>(img_hwc, H, W, rh, rw, S_h, S_w, nm,
                                                                                        pad_value, out);
    return csm::check_launch("k_det_preprocess");
}
