/*
 * oracle/warp_oracle.c  --  TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement (plain C, sequential) of the Ken Burns warp hot path of the
 * reference.  Only tests/, __graft_entry__.smoke() and bench.py's
 * `cpu_baseline` leg may load this; the product path (libcsm355.so) never does.
 *
 * Build: gcc -O2 -ffp-contract=off -fno-fast-math -shared -fPIC  (oracle/Makefile)
 * -ffp-contract=off is part of the contract: every float op below rounds once,
 * exactly as written, so the HIP kernels (also built with -ffp-contract=off)
 * can be compared bit-for-bit where the algorithm is order-independent.
 *
 * Parity pin: tests/golden/warp_*.npz are produced by
 * tests/golden/make_golden_warp.py, which loads the reference's own
 * anime_3dkenburns/models/utils.py + common.py by path, captures the expanded
 * CUDA kernel text from utils/cupy_utils.py::preprocess_kernel and executes
 * that text sequentially on the CPU.  tests/test_oracle_warp.py checks every
 * function here against those fixtures.
 *
 * Reference citations are relative to /root/reference.
 *
 * Mixed-precision notes: the CUDA text uses untyped literals, so
 * several sub-expressions are evaluated in double and then rounded to float.
 * They are restated literally below; do not "simplify" them.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* ---- projection shared by updateZee / updateOutput ------------------------
 * anime_3dkenburns/models/utils.py:76-99 and :229-252 (identical text).
 * returns 0 if the point is skipped. */
static int orc_project(float x, float y, float z, double focal, double baseline,
                       int W, int H, float *ox, float *oy, float *err)
{
    /* float3 fltPlanePoint = make_float3(0.0, 0.0, focal); normal = (0,0,1) */
    float ppz = (float)focal;
    /* fltLineVector = make_float3(0,0,0) - fltLinePoint */
    float lvx = 0.0f - x, lvy = 0.0f - y, lvz = 0.0f - z;
    if ((double)z < 0.001) return 0;                               /* :82 */
    /* dot(a,b) = a.x*b.x + a.y*b.y + a.z*b.z   (helper_math.h dot(float3,float3)) */
    float ax = 0.0f - x, ay = 0.0f - y, az = ppz - z;
    float num = ax * 0.0f + ay * 0.0f + az * 1.0f;                 /* :86 */
    float den = lvx * 0.0f + lvy * 0.0f + lvz * 1.0f;              /* :87 */
    float dist = num / den;                                        /* :88 */
    if ((double)fabsf(den) < 0.001) return 0;                      /* :90 */
    float ix = x + dist * lvx;                                     /* :94 */
    float iy = y + dist * lvy;
    *ox = (float)(((double)ix + (0.5 * W)) - 0.5);                 /* :96 double expr */
    *oy = (float)(((double)iy + (0.5 * H)) - 0.5);                 /* :97 */
    *err = (float)(1000000.0 - ((focal * baseline) / ((double)z + 0.0000001))); /* :99 */
    return 1;
}

static void orc_corners(float fx, float fy, int *nwx, int *nwy, float w[4])
{
    int x0 = (int)floorf(fx), y0 = (int)floorf(fy);                /* :101-102 */
    int nex = x0 + 1, ney = y0, swx = x0, swy = y0 + 1, sex = x0 + 1, sey = y0 + 1;
    w[0] = ((float)sex - fx) * ((float)sey - fy);                  /* NW :110 */
    w[1] = (fx - (float)swx) * ((float)swy - fy);                  /* NE :111 */
    w[2] = ((float)nex - fx) * (fy - (float)ney);                  /* SW :112 */
    w[3] = (fx - (float)x0) * (fy - (float)y0);                    /* SE :113 */
    *nwx = x0; *nwy = y0;
}

/* kernel_pointrender_updateZee  (models/utils.py:63-149)
 * pts [B,3,N]; zee [B,1,H,W] must be pre-filled with 1e6 (:59). */
void orc_pointrender_update_zee(int B, int64_t N, int H, int W, double focal,
                                double baseline, const float *pts, float *zee)
{
    for (int b = 0; b < B; ++b)
        for (int64_t p = 0; p < N; ++p) {
            const float *P = pts + (int64_t)b * 3 * N;
            float fx, fy, err, w[4];
            if (!orc_project(P[p], P[N + p], P[2 * N + p], focal, baseline, W, H, &fx, &fy, &err)) continue;
            int x0, y0;
            orc_corners(fx, fy, &x0, &y0, w);
            float nw = w[0], ne = w[1], sw = w[2], se = w[3];
            int cx, cy;
            if (nw >= ne && nw >= sw && nw >= se) { cx = x0; cy = y0; }              /* :115 */
            else if (ne >= nw && ne >= sw && ne >= se) { cx = x0 + 1; cy = y0; }     /* :120 */
            else if (sw >= nw && sw >= ne && sw >= se) { cx = x0; cy = y0 + 1; }     /* :125 */
            else if (se >= nw && se >= ne && se >= sw) { cx = x0 + 1; cy = y0 + 1; } /* :130 */
            else continue; /* NaN weights */
            if (cx >= 0 && cx < W && cy >= 0 && cy < H) {
                float *z = zee + ((int64_t)b * H + cy) * W + cx;
                if (*z > err) *z = err;          /* float atomicMin, cupy_utils.py:21-29 */
            }
        }
}

/* kernel_pointrender_updateDegrid  (models/utils.py:152-212)
 * mode 0: in-place, raster order (ONE legal interleaving of the racy reference)
 * mode 1: Jacobi -- all reads from a snapshot taken before the pass.  This is
 *         the deterministic semantics the HIP build adopts. */
static int orc_mt = 0;     /* set only inside orc_warp_frame_mt (the timed all-cores baseline); 0 = the sequential checker */

void orc_pointrender_degrid(int B, int H, int W, float *zee, int mode)
{
    static const int ox[4] = {1, 0, 1, 1}, oy[4] = {0, 1, 1, -1};   /* :170-171 */
    int64_t n = (int64_t)B * H * W;
    float *src = zee;
    if (mode == 1) { src = (float *)malloc(n * sizeof(float)); memcpy(src, zee, n * sizeof(float)); }
    for (int b = 0; b < B; ++b)
#pragma omp parallel for schedule(static) if (orc_mt && mode == 1)
        for (int y = 0; y < H; ++y)
            for (int x = 0; x < W; ++x) {
                const float *Z = src + (int64_t)b * H * W;
                int cnt = 0; float sum = 0.0f;
                float c = Z[(int64_t)y * W + x];
                for (int k = 0; k < 4; ++k) {
                    int x1 = x + ox[k], y1 = y + oy[k], x2 = x - ox[k], y2 = y - oy[k];
                    if (x1 < 0 || x1 >= W || y1 < 0 || y1 >= H) continue;     /* :179 */
                    if (x2 < 0 || x2 >= W || y2 < 0 || y2 >= H) continue;     /* :182 */
                    float a = Z[(int64_t)y1 * W + x1], d = Z[(int64_t)y2 * W + x2];
                    if ((double)c >= (double)a + 1.0)                         /* :187 double */
                        if ((double)c >= (double)d + 1.0) {                   /* :188 */
                            cnt += 2; sum += a; sum += d;                     /* :189-191 */
                        }
                }
                if (cnt > 0) {
                    float m = sum / (float)cnt;                               /* :197 float/int */
                    zee[((int64_t)b * H + y) * W + x] = fminf(c, m);
                }
            }
    if (mode == 1) free(src);
}

/* kernel_pointrender_updateOutput  (models/utils.py:215-313)
 * data [B,C1,N] (C1 = C+1, ones channel appended by the caller, :57),
 * out [B,C1,H,W] zero-initialised (:60).  Sequential atomicAdd order = point order. */
void orc_pointrender_update_output(int B, int64_t N, int C1, int H, int W, double focal,
                                   double baseline, const float *pts, const float *data,
                                   const float *zee, float *out)
{
    for (int b = 0; b < B; ++b)
        for (int64_t p = 0; p < N; ++p) {
            const float *P = pts + (int64_t)b * 3 * N;
            const float *D = data + (int64_t)b * C1 * N;
            float fx, fy, err, w[4];
            if (!orc_project(P[p], P[N + p], P[2 * N + p], focal, baseline, W, H, &fx, &fy, &err)) continue;
            int x0, y0;
            orc_corners(fx, fy, &x0, &y0, w);
            const int dx[4] = {0, 1, 0, 1}, dy[4] = {0, 0, 1, 1};  /* NW, NE, SW, SE :267-297 */
            for (int k = 0; k < 4; ++k) {
                int cx = x0 + dx[k], cy = y0 + dy[k];
                if (cx < 0 || cx >= W || cy < 0 || cy >= H) continue;
                float zc = zee[((int64_t)b * H + cy) * W + cx];
                if (!((double)err <= (double)zc + 1.0)) continue;              /* :269 double */
                for (int c = 0; c < C1; ++c)
                    out[(((int64_t)b * C1 + c) * H + cy) * W + cx] += D[(int64_t)c * N + p] * w[k];
            }
        }
}

/* render_pointcloud  (models/utils.py:56-315) -- glue + the three kernels.
 * data [B,C,N]; render [B,C,H,W]; existing [B,1,H,W]. degrid_mode as above. */
void orc_render_pointcloud(int B, int C, int64_t N, int H, int W, double focal, double baseline,
                           const float *pts, const float *data, int degrid_mode,
                           float *render, float *existing, float *zee_out /* may be NULL */)
{
    int C1 = C + 1;
    int64_t P = (int64_t)H * W;
    float *d1 = (float *)malloc((size_t)B * C1 * N * sizeof(float));
    float *zee = (float *)malloc((size_t)B * P * sizeof(float));
    float *acc = (float *)calloc((size_t)B * C1 * P, sizeof(float));
    for (int b = 0; b < B; ++b) {
        memcpy(d1 + (int64_t)b * C1 * N, data + (int64_t)b * C * N, (size_t)C * N * sizeof(float));
        for (int64_t p = 0; p < N; ++p) d1[((int64_t)b * C1 + C) * N + p] = 1.0f;      /* :57 */
    }
    for (int64_t i = 0; i < B * P; ++i) zee[i] = 1000000.0f;                             /* :59 */
    orc_pointrender_update_zee(B, N, H, W, focal, baseline, pts, zee);
    orc_pointrender_degrid(B, H, W, zee, degrid_mode);
    orc_pointrender_update_output(B, N, C1, H, W, focal, baseline, pts, d1, zee, acc);
    for (int b = 0; b < B; ++b)
        for (int64_t i = 0; i < P; ++i) {
            float e = acc[((int64_t)b * C1 + C) * P + i];
            float den = e + 0.0000001f;                                                  /* :315 */
            for (int c = 0; c < C; ++c)
                render[((int64_t)b * C + c) * P + i] = acc[((int64_t)b * C1 + c) * P + i] / den;
            existing[(int64_t)b * P + i] = e;
        }
    if (zee_out) memcpy(zee_out, zee, (size_t)B * P * sizeof(float));
    free(d1); free(zee); free(acc);
}

/* kernel_discfill_updateOutput  (anime_3dkenburns/common.py:149-245)
 * in [B,C,H,W], depth [B,1,H,W]; out must be a copy of in (common.py:146). */
void orc_fill_disocclusion(int B, int C, int H, int W, const float *in, const float *depth, float *out)
{
    float dirx[16] = {-1, 0, 1, 1, -1, 1, 2, 2, -2, -1, 1, 2, 3, 3, 3, 3};              /* :168 */
    float diry[16] = {1, 1, 1, 0, 2, 2, 1, -1, 3, 3, 3, 3, 2, 1, -1, -2};               /* :169 */
    for (int k = 0; k < 16; ++k) {
        float nrm = sqrtf((dirx[k] * dirx[k]) + (diry[k] * diry[k]));                    /* :172 */
        dirx[k] /= nrm; diry[k] /= nrm;
    }
    int64_t P = (int64_t)H * W;
    for (int b = 0; b < B; ++b) {
        const float *D = depth + (int64_t)b * P;
#pragma omp parallel for schedule(dynamic, 8) if (orc_mt)
        for (int y = 0; y < H; ++y)
            for (int x = 0; x < W; ++x) {
                if ((double)D[(int64_t)y * W + x] > 0.0) continue;                       /* :160 */
                float shortest = 1000000.0f;
                int fillx = -1, filly = -1;
                for (int k = 0; k < 16; ++k) {
                    float ffx = (float)x, ffy = (float)y; int ifx = 0, ify = 0;
                    float ftx = (float)x, fty = (float)y; int itx = 0, ity = 0;
                    for (;;) {                                                            /* :186-193 */
                        ffx -= dirx[k]; ifx = (int)roundf(ffx);
                        ffy -= diry[k]; ify = (int)roundf(ffy);
                        if (ifx < 0 || ifx >= W) break;
                        if (ify < 0 || ify >= H) break;
                        if ((double)D[(int64_t)ify * W + ifx] > 0.0) break;
                    }
                    if (ifx < 0 || ifx >= W) continue;
                    if (ify < 0 || ify >= H) continue;
                    for (;;) {                                                            /* :197-204 */
                        ftx += dirx[k]; itx = (int)roundf(ftx);
                        fty += diry[k]; ity = (int)roundf(fty);
                        if (itx < 0 || itx >= W) break;
                        if (ity < 0 || ity >= H) break;
                        if ((double)D[(int64_t)ity * W + itx] > 0.0) break;
                    }
                    if (itx < 0 || itx >= W) continue;
                    if (ity < 0 || ity >= H) continue;
                    /* sqrt(powf(dx,2)+powf(dy,2)) :208 -- integer squares are exact in fp32 here */
                    float ddx = (float)(itx - ifx), ddy = (float)(ity - ify);
                    float dist = sqrtf(ddx * ddx + ddy * ddy);
                    if (shortest > dist) {                                                /* :210 */
                        fillx = ifx; filly = ify;
                        if (D[(int64_t)ify * W + ifx] < D[(int64_t)ity * W + itx]) { fillx = itx; filly = ity; }
                        shortest = dist;
                    }
                }
                if (fillx == -1 || filly == -1) continue;                                 /* :224-230 */
                for (int c = 0; c < C; ++c)
                    out[(((int64_t)b * C + c) * H + y) * W + x] =
                        in[(((int64_t)b * C + c) * H + filly) * W + fillx];
            }
    }
}

/* spatial_filter(x,'laplacian')  (models/utils.py:12-24): replicate pad, then a
 * cross-correlation with taps k[0][1]=k[0][2]=k[1][0]=k[2][0]=-1, k[1][1]=4.
 * Summation order here: row-major over the 3x3 kernel, fmaf-free. */
void orc_spatial_filter_laplacian(int B, int H, int W, const float *in, float *out)
{
    for (int b = 0; b < B; ++b) {
        const float *I = in + (int64_t)b * H * W;
        for (int y = 0; y < H; ++y)
            for (int x = 0; x < W; ++x) {
                int ym = y > 0 ? y - 1 : 0, yp = y < H - 1 ? y + 1 : H - 1;
                int xm = x > 0 ? x - 1 : 0, xp = x < W - 1 ? x + 1 : W - 1;
                float acc = 0.0f;
                acc += -1.0f * I[(int64_t)ym * W + x];    /* k[0][1] */
                acc += -1.0f * I[(int64_t)ym * W + xp];   /* k[0][2] */
                acc += -1.0f * I[(int64_t)y * W + xm];    /* k[1][0] */
                acc += 4.0f * I[(int64_t)y * W + x];      /* k[1][1] */
                acc += -1.0f * I[(int64_t)yp * W + xm];   /* k[2][0] */
                out[((int64_t)b * H + y) * W + x] = acc;
            }
    }
}

/* spatial_filter(x,'median-5')  (models/utils.py:32-36): reflect pad 2, lower median of the 25 window values */
static int cmp_f(const void *a, const void *b) { float x = *(const float *)a, y = *(const float *)b; return (x > y) - (x < y); }
void orc_spatial_filter_median5(int B, int H, int W, const float *in, float *out)
{
    for (int b = 0; b < B; ++b)
        for (int y = 0; y < H; ++y)
            for (int x = 0; x < W; ++x) {
                float v[25]; int k = 0;
                for (int dy = -2; dy <= 2; ++dy)
                    for (int dx = -2; dx <= 2; ++dx) {
                        int yy = y + dy, xx = x + dx;
                        yy = yy < 0 ? -yy : (yy >= H ? 2 * H - 2 - yy : yy);
                        xx = xx < 0 ? -xx : (xx >= W ? 2 * W - 2 - xx : xx);
                        v[k++] = in[((int64_t)b * H + yy) * W + xx];
                    }
                qsort(v, 25, sizeof(float), cmp_f);
                out[((int64_t)b * H + y) * W + x] = v[12];
            }
}

/* spatial_filter(x,'median-3')  (models/utils.py:26-30): reflect pad 1, lower median (5th) of the 9 window values */
void orc_spatial_filter_median3(int B, int H, int W, const float *in, float *out)
{
    for (int b = 0; b < B; ++b)
        for (int y = 0; y < H; ++y)
            for (int x = 0; x < W; ++x) {
                float v[9]; int k = 0;
                for (int dy = -1; dy <= 1; ++dy)
                    for (int dx = -1; dx <= 1; ++dx) {
                        int yy = y + dy, xx = x + dx;
                        yy = yy < 0 ? -yy : (yy >= H ? 2 * H - 2 - yy : yy);
                        xx = xx < 0 ? -xx : (xx >= W ? 2 * W - 2 - xx : xx);
                        v[k++] = in[((int64_t)b * H + yy) * W + xx];
                    }
                qsort(v, 9, sizeof(float), cmp_f);
                out[((int64_t)b * H + y) * W + x] = v[4];
            }
}

/* depth_to_points  (models/utils.py:43-50).  linspace(-W/2+.5, W/2-.5, W) has
 * step exactly 1.0f, so h[i] = -0.5*W + 0.5 + i exactly. */
void orc_depth_to_points(int B, int H, int W, double focal, const float *depth, float *pts)
{
    float invf = (float)(1.0 / focal);
    int64_t P = (int64_t)H * W;
    for (int b = 0; b < B; ++b)
        for (int y = 0; y < H; ++y)
            for (int x = 0; x < W; ++x) {
                float d = depth[(int64_t)b * P + (int64_t)y * W + x];
                float hx = ((float)(-0.5 * W + 0.5) + (float)x) * invf;
                float vy = ((float)(-0.5 * H + 0.5) + (float)y) * invf;
                pts[((int64_t)b * 3 + 0) * P + (int64_t)y * W + x] = d * hx;
                pts[((int64_t)b * 3 + 1) * P + (int64_t)y * W + x] = d * vy;
                pts[((int64_t)b * 3 + 2) * P + (int64_t)y * W + x] = d;
            }
}

/* disparity -> (depth, valid, points, unaltered)  (kenburns_effect.py:928-933)
 *   disparity = disparity / max * baseline ; depth = focal*baseline/(disparity+1e-5)
 *   valid = |laplacian(disparity / max(disparity))| < 0.03 */
void orc_disparity_to_points(int H, int W, double focal, double baseline, const float *disp_in,
                             float *disp, float *depth, float *valid, float *pts, float *unaltered)
{
    int64_t P = (int64_t)H * W;
    float mx = -INFINITY;
    for (int64_t i = 0; i < P; ++i) if (disp_in[i] > mx) mx = disp_in[i];
    for (int64_t i = 0; i < P; ++i) disp[i] = disp_in[i] / mx * (float)baseline;
    float fb = (float)(focal * baseline);
    /* python: float / Tensor  ==  Tensor.__rtruediv__  ==  tensor.reciprocal() * float  (two roundings) */
    for (int64_t i = 0; i < P; ++i) depth[i] = (1.0f / (disp[i] + 0.00001f)) * fb;
    float mx2 = -INFINITY;
    for (int64_t i = 0; i < P; ++i) if (disp[i] > mx2) mx2 = disp[i];
    float *nd = (float *)malloc(P * sizeof(float)), *lap = (float *)malloc(P * sizeof(float));
    for (int64_t i = 0; i < P; ++i) nd[i] = disp[i] / mx2;
    orc_spatial_filter_laplacian(1, H, W, nd, lap);
    for (int64_t i = 0; i < P; ++i) valid[i] = fabsf(lap[i]) < 0.03f ? 1.0f : 0.0f;
    for (int64_t i = 0; i < P; ++i) nd[i] = depth[i] * valid[i];
    orc_depth_to_points(1, H, W, focal, nd, pts);
    orc_depth_to_points(1, H, W, focal, depth, unaltered);
    free(nd); free(lap);
}

/* point part of process_shift  (common.py:74-81); shift = FloatTensor([sx,sy,sz]) */
void orc_process_shift(int B, int64_t N, float sx, float sy, float sz, const float *pts, float *out)
{
    for (int b = 0; b < B; ++b) {
        const float *P = pts + (int64_t)b * 3 * N; float *O = out + (int64_t)b * 3 * N;
        for (int64_t p = 0; p < N; ++p) {
            float z = P[2 * N + p];
            float r = z / (z + 0.0000001f);                 /* :78-79 */
            O[p] = P[p] * r + sx;                           /* :81 */
            O[N + p] = P[N + p] * r + sy;
            O[2 * N + p] = z + sz;
        }
    }
}

/* frame = (render[0,0:3]*255).clip(0,255).astype(uint8), HWC  (kenburns_effect.py:1040) */
void orc_frame_to_u8(int C_total, int H, int W, const float *render, uint8_t *frame)
{
    int64_t P = (int64_t)H * W; (void)C_total;
    for (int64_t i = 0; i < P; ++i)
        for (int c = 0; c < 3; ++c) {
            float v = render[(int64_t)c * P + i] * 255.0f;
            v = v < 0.0f ? 0.0f : (v > 255.0f ? 255.0f : v);
            frame[i * 3 + c] = (uint8_t)v;
        }
}

/* One full warp "frame": process_shift -> render_pointcloud(cat[rgb,depth]) ->
 * fill_disocclusion(render, render[3]*(existing>0)) -> uint8   (kenburns_effect.py:1027-1040)
 * pts [1,3,N], rgbd [1,4,N]; outputs: render_filled [1,4,H,W], frame u8 [H,W,3] */
void orc_warp_frame(int64_t N, int H, int W, double focal, double baseline, float sx, float sy, float sz,
                    const float *pts, const float *rgbd, int degrid_mode,
                    float *render_filled, float *existing, uint8_t *frame)
{
    int64_t P = (int64_t)H * W;
    float *sp = (float *)malloc((size_t)3 * N * sizeof(float));
    float *rnd = (float *)malloc((size_t)4 * P * sizeof(float));
    float *dm = (float *)malloc((size_t)P * sizeof(float));
    orc_process_shift(1, N, sx, sy, sz, pts, sp);
    orc_render_pointcloud(1, 4, N, H, W, focal, baseline, sp, rgbd, degrid_mode, rnd, existing, NULL);
    for (int64_t i = 0; i < P; ++i) dm[i] = rnd[3 * P + i] * (existing[i] > 0.0f ? 1.0f : 0.0f);
    memcpy(render_filled, rnd, (size_t)4 * P * sizeof(float));
    orc_fill_disocclusion(1, 4, H, W, rnd, dm, render_filled);
    orc_frame_to_u8(4, H, W, render_filled, frame);
    free(sp); free(rnd); free(dm);
}


/* ---- all-cores variant of orc_warp_frame: TIMING ONLY (bench.py cpu_baseline), never used as a checker ------------------------
 * The reference's kernels are one-thread-per-point with atomics (models/utils.py:63-313); this is the same formulation with OpenMP
 * threads instead of GPU lanes: float atomicMin as a compare-exchange loop on the bit pattern, atomicAdd as `omp atomic`.  Like the
 * reference on a GPU, the accumulation ORDER is then unspecified, so results equal the sequential checker's only up to fp32
 * summation order (tests/test_oracle_warp.py compares them with a tolerance). */
static void atomic_min_float(float *addr, float v)
{
    int *ia = (int *)addr;
    int old = __atomic_load_n(ia, __ATOMIC_RELAXED);
    for (;;) {
        float cur; memcpy(&cur, &old, 4);
        if (!(cur > v)) return;
        int nv; memcpy(&nv, &v, 4);
        if (__atomic_compare_exchange_n(ia, &old, nv, 1, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) return;
    }
}

void orc_warp_frame_mt(int64_t N, int H, int W, double focal, double baseline, float sx, float sy, float sz,
                       const float *pts, const float *rgbd, float *render_filled, float *existing, uint8_t *frame)
{
    const int C = 4, C1 = 5;
    int64_t P = (int64_t)H * W;
    float *sp = (float *)malloc((size_t)3 * N * sizeof(float));
    float *zee = (float *)malloc((size_t)P * sizeof(float));
    float *acc = (float *)calloc((size_t)C1 * P, sizeof(float));
    float *rnd = (float *)malloc((size_t)C * P * sizeof(float));
    float *dm = (float *)malloc((size_t)P * sizeof(float));
    orc_mt = 1;
#pragma omp parallel for schedule(static)
    for (int64_t p = 0; p < N; ++p) {                                  /* process_shift, common.py:74-81 */
        float z = pts[2 * N + p];
        float r = z / (z + 0.0000001f);
        sp[p] = pts[p] * r + sx; sp[N + p] = pts[N + p] * r + sy; sp[2 * N + p] = z + sz;
    }
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < P; ++i) zee[i] = 1000000.0f;
#pragma omp parallel for schedule(static)
    for (int64_t p = 0; p < N; ++p) {                                  /* updateZee */
        float fx, fy, err, w[4];
        if (!orc_project(sp[p], sp[N + p], sp[2 * N + p], focal, baseline, W, H, &fx, &fy, &err)) continue;
        int x0, y0;
        orc_corners(fx, fy, &x0, &y0, w);
        float nw = w[0], ne = w[1], sw = w[2], se = w[3];
        int cx, cy;
        if (nw >= ne && nw >= sw && nw >= se) { cx = x0; cy = y0; }
        else if (ne >= nw && ne >= sw && ne >= se) { cx = x0 + 1; cy = y0; }
        else if (sw >= nw && sw >= ne && sw >= se) { cx = x0; cy = y0 + 1; }
        else if (se >= nw && se >= ne && se >= sw) { cx = x0 + 1; cy = y0 + 1; }
        else continue;
        if (cx >= 0 && cx < W && cy >= 0 && cy < H) atomic_min_float(zee + (int64_t)cy * W + cx, err);
    }
    orc_pointrender_degrid(1, H, W, zee, 1);
#pragma omp parallel for schedule(static)
    for (int64_t p = 0; p < N; ++p) {                                  /* updateOutput */
        float fx, fy, err, w[4];
        if (!orc_project(sp[p], sp[N + p], sp[2 * N + p], focal, baseline, W, H, &fx, &fy, &err)) continue;
        int x0, y0;
        orc_corners(fx, fy, &x0, &y0, w);
        const int dx[4] = {0, 1, 0, 1}, dy[4] = {0, 0, 1, 1};
        for (int k = 0; k < 4; ++k) {
            int cx = x0 + dx[k], cy = y0 + dy[k];
            if (cx < 0 || cx >= W || cy < 0 || cy >= H) continue;
            float zc = zee[(int64_t)cy * W + cx];
            if (!((double)err <= (double)zc + 1.0)) continue;
            for (int c = 0; c < C1; ++c) {
                float v = (c < C ? rgbd[(int64_t)c * N + p] : 1.0f) * w[k];
#pragma omp atomic
                acc[((int64_t)c * H + cy) * W + cx] += v;
            }
        }
    }
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < P; ++i) {
        float e = acc[(int64_t)C * P + i], den = e + 0.0000001f;
        for (int c = 0; c < C; ++c) rnd[(int64_t)c * P + i] = acc[(int64_t)c * P + i] / den;
        existing[i] = e;
        dm[i] = rnd[3 * P + i] * (e > 0.0f ? 1.0f : 0.0f);
    }
    memcpy(render_filled, rnd, (size_t)C * P * sizeof(float));
    orc_fill_disocclusion(1, C, H, W, rnd, dm, render_filled);
    orc_frame_to_u8(C, H, W, render_filled, frame);
    orc_mt = 0;
    free(sp); free(zee); free(acc); free(rnd); free(dm);
}
