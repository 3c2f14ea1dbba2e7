/*
 * oracle/nets_oracle.c  --  TEST INFRASTRUCTURE ONLY.
 *
 * CPU interpreter of the layer program (csm_op records, include/csm355.h) that the product
 * executes with HIP kernels.  It is the bit-exact checker of the dense nets: every convolution
 * output is one fp32 fmaf chain in the contract's order (bias first; 32-channel blocks outer, taps row-major inner; aligned
 * blocks of 8 input channels in the order 0,4,1,5,2,6,3,7), activations use the same polynomial
 * expf, pooling / resize follow aten's index rules.
 *
 * Independence from the product: this file reads NATURAL weight layouts
 * ([cout][cin_g][kh][kw] for convs, [c][kh][kw] for depthwise) -- the product's packed/padded
 * MFMA layout is never seen here, so a packing bug cannot cancel out.
 *
 * Parity pin: oracle/nets (this interpreter run on programs lowered from the build's net
 * definitions) is checked in tests/test_oracle_nets.py against fixtures produced by the
 * reference's own torch modules (tests/golden/make_golden_nets.py: ISNetDIS isnet.py:524-645,
 * LeReS network_auxi.py / Resnext_torch.py) -- tolerance 2e-4 relative (BN folding + summation
 * order differ from torch's kernels).  RTMDet (mmdet 3.3.0, not in /root/reference): parity
 * unpinned, see DESIGN.md.
 *
 * Build: gcc -O2 -ffp-contract=off -mfma -fopenmp (oracle/Makefile).  fmaf() is explicit, so
 * -ffp-contract=off does not change it and -mfma only makes it one instruction.
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "../include/csm355.h"

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
    int32_t bits = ((int32_t)n + 127) << 23;
    float s; memcpy(&s, &bits, 4);
    return e * s;
}

static float orc_logf(float x)
{
    int32_t bits; memcpy(&bits, &x, 4);
    int e = ((bits >> 23) & 0xff) - 126;
    int32_t mb = (bits & 0x007fffff) | 0x3f000000;
    float m; memcpy(&m, &mb, 4);
    if (m < 0.707106781186547524f) { e -= 1; m = (m + m) - 1.0f; } else { m = m - 1.0f; }
    const float z = m * m;
    float y = 7.0376836292e-2f;
    y = fmaf(y, m, -1.1514610310e-1f); y = fmaf(y, m, 1.1676998740e-1f); y = fmaf(y, m, -1.2420140846e-1f);
    y = fmaf(y, m, 1.4249322787e-1f); y = fmaf(y, m, -1.6668057665e-1f); y = fmaf(y, m, 2.0000714765e-1f);
    y = fmaf(y, m, -2.4999993993e-1f); y = fmaf(y, m, 3.3333331174e-1f);
    y = y * m * z;
    const float fe = (float)e;
    y = fmaf(fe, -2.12194440e-4f, y);
    y = fmaf(z, -0.5f, y);
    return fmaf(fe, 0.693359375f, m + y);
}

static float orc_erff(float x)
{
    const float ax = fabsf(x);
    const float t = 1.0f / fmaf(0.3275911f, ax, 1.0f);
    float p = 1.061405429f;
    p = fmaf(p, t, -1.453152027f); p = fmaf(p, t, 1.421413741f); p = fmaf(p, t, -0.284496736f); p = fmaf(p, t, 0.254829592f);
    const float r = 1.0f - (p * t) * orc_expf(-(ax * ax));
    return x < 0.0f ? -r : r;
}

static float orc_act(float v, int act, float slope)
{
    switch (act) {
        case CSM_ACT_SOFTPLUS: return v > 20.0f ? v : orc_logf(1.0f + orc_expf(v));
        case CSM_ACT_GELU: return (0.5f * v) * (1.0f + orc_erff(v * 0.707106781186547524f));
        case CSM_ACT_RELU: return fmaxf(v, 0.0f);
        case CSM_ACT_SILU: return v / (1.0f + orc_expf(-v));
        case CSM_ACT_PRELU: return v >= 0.0f ? v : v * slope;
        case CSM_ACT_HSIGMOID: return fminf(fmaxf(v + 3.0f, 0.0f), 6.0f) / 6.0f;
        case CSM_ACT_SIGMOID: return 1.0f / (1.0f + orc_expf(-v));
        default: return v;
    }
}

typedef struct { float *p; int n, h, w, c, ld; } view_t;

static int get_view(const csm_tensor_desc *t, int id, float *ws, void *const *ext, view_t *v)
{
    if (id < 0) { memset(v, 0, sizeof(*v)); return 0; }
    float *base = t[id].ext >= 0 ? (float *)ext[t[id].ext] : ws;
    v->p = base + t[id].offset; v->n = t[id].n; v->h = t[id].h; v->w = t[id].w; v->c = t[id].c; v->ld = t[id].ld;
    return 0;
}

/* conv: weights natural [groups*cout_g][cin_g][kh][kw]; op->groups = REAL groups */
static void orc_conv(const csm_op *op, view_t in, view_t res, view_t out, const float *W, const float *bias,
                     const float *slope)
{
    const int kh = op->kh, kw = op->kw, cin = op->cin_g, cout = op->cout_g, G = op->groups;
    const int64_t M = (int64_t)out.n * out.h * out.w;
#pragma omp parallel for schedule(static)
    for (int64_t m = 0; m < M; ++m) {
        int n = (int)(m / ((int64_t)out.h * out.w));
        int rem = (int)(m - (int64_t)n * out.h * out.w);
        int oy = rem / out.w, ox = rem - oy * out.w;
        for (int g = 0; g < G; ++g)
            for (int co = 0; co < cout; ++co) {
                int oc = g * cout + co;
                /* K runs (csm_op.ksplit): chunk = (32-channel block, tap), block-major; run s = chunks [s*T/S, (s+1)*T/S) */
                const int S = op->ksplit > 1 ? op->ksplit : 1, ncb = (cin + 31) / 32, Tall = kh * kw * ncb;
                float part[16];
                for (int s_ = 0; s_ < S; ++s_) part[s_] = 0.0f;
                part[0] = bias ? bias[oc] : 0.0f;
                /* chain order: 32-channel block (outer), taps row-major, 8-channel sub-blocks, channels 0,4,1,5,2,6,3,7 */
                for (int cb = 0; cb < ncb; ++cb)
                    for (int ky = 0; ky < kh; ++ky) {
                        int iy = oy * op->stride - op->pad + ky * op->dil;
                        for (int kx = 0; kx < kw; ++kx) {
                            int ix = ox * op->stride - op->pad + kx * op->dil;
                            if (iy < 0 || iy >= in.h || ix < 0 || ix >= in.w) continue;
                            const float *x = in.p + ((int64_t)(n * in.h + iy) * in.w + ix) * in.ld + g * cin;
                            const float *w = W + ((int64_t)oc * cin) * kh * kw + ky * kw + kx;
                            int chunk = cb * kh * kw + ky * kw + kx, run = 0;
                            if (S > 1) { run = (int)(((int64_t)(chunk + 1) * S - 1) / Tall); while ((int64_t)run * Tall / S > chunk) --run; while ((int64_t)(run + 1) * Tall / S <= chunk) ++run; }
                            float a_ = part[run];
                            for (int kb = cb * 32; kb < cb * 32 + 32 && kb < cin; kb += 8)
                                for (int t = 0; t < 4; ++t)
                                    for (int h = 0; h < 2; ++h) {
                                        int c = kb + 4 * h + t;
                                        if (c < cin) a_ = fmaf(x[c], w[(int64_t)c * kh * kw], a_);
                                    }
                            part[run] = a_;
                        }
                    }
                float acc = part[0];
                for (int s_ = 1; s_ < S; ++s_) acc += part[s_];
                if (op->res_mode == 1 && res.p) acc += res.p[m * res.ld + oc];
                acc = orc_act(acc, op->act, slope ? slope[oc] : 0.0f);
                if (op->res_mode == 2 && res.p) acc += res.p[m * res.ld + oc];
                out.p[m * out.ld + oc] = acc;
            }
    }
}

/* The same convolution, organised for speed (bench.py's cpu_baseline times the full-size nets with it): identical fmaf chains --
 * same operands, same order, taps outside the image skipped, channels >= cin skipped -- but the weights are first re-laid in CHAIN
 * order per output channel (unit-stride reads instead of a kh*kw stride), and four output channels advance together so that four
 * independent chains hide the FMA latency.  tests/test_oracle_nets.py asserts bit-equality with orc_conv on every feature
 * (groups, stride, dilation, ragged cin / cout, split-K).  ORC_CONV_REFERENCE=1 selects the plain loop nest. */
static void orc_conv_fast(const csm_op *op, view_t in, view_t res, view_t out, const float *W, const float *bias,
                          const float *slope)
{
    const int kh = op->kh, kw = op->kw, cin = op->cin_g, cout = op->cout_g, G = op->groups, taps = kh * kw;
    const int ncb = (cin + 31) / 32, S = op->ksplit > 1 ? op->ksplit : 1, Tall = taps * ncb;
    const int64_t M = (int64_t)out.n * out.h * out.w;
    int *cidx = (int *)malloc(sizeof(int) * (size_t)ncb * 32), *ncnt = (int *)malloc(sizeof(int) * ncb);
    int *koff = (int *)malloc(sizeof(int) * (ncb + 1)), *run_of = (int *)malloc(sizeof(int) * Tall);
    int Ktot = 0;
    for (int cb = 0; cb < ncb; ++cb) {            /* channels of the block in chain order: 8-blocks, then 0,4,1,5,2,6,3,7 */
        int n = 0;
        for (int kb = cb * 32; kb < cb * 32 + 32 && kb < cin; kb += 8)
            for (int t = 0; t < 4; ++t)
                for (int h = 0; h < 2; ++h) { int c = kb + 4 * h + t; if (c < cin) cidx[cb * 32 + n++] = c; }
        ncnt[cb] = n; koff[cb] = Ktot; Ktot += n * taps;
    }
    koff[ncb] = Ktot;
    for (int chunk = 0; chunk < Tall; ++chunk) {
        int run = 0;
        if (S > 1) { run = (int)(((int64_t)(chunk + 1) * S - 1) / Tall); while ((int64_t)run * Tall / S > chunk) --run; while ((int64_t)(run + 1) * Tall / S <= chunk) ++run; }
        run_of[chunk] = run;
    }
    float *Wt = (float *)malloc(sizeof(float) * (size_t)G * cout * Ktot);
#pragma omp parallel for schedule(static)
    for (int oc = 0; oc < G * cout; ++oc)
        for (int cb = 0; cb < ncb; ++cb)
            for (int tp = 0; tp < taps; ++tp)
                for (int i = 0; i < ncnt[cb]; ++i)
                    Wt[(int64_t)oc * Ktot + koff[cb] + tp * ncnt[cb] + i] = W[((int64_t)oc * cin + cidx[cb * 32 + i]) * taps + tp];
#pragma omp parallel for schedule(dynamic, 64)
    for (int64_t m = 0; m < M; ++m) {
        int n = (int)(m / ((int64_t)out.h * out.w));
        int rem = (int)(m - (int64_t)n * out.h * out.w);
        int oy = rem / out.w, ox = rem - oy * out.w;
        const float *xp[taps];                      /* (a VLA: the 16 x 16 patch embedding of the BEiT core has 256 taps) */
        for (int ky = 0; ky < kh; ++ky)
            for (int kx = 0; kx < kw; ++kx) {
                int iy = oy * op->stride - op->pad + ky * op->dil, ix = ox * op->stride - op->pad + kx * op->dil;
                xp[ky * kw + kx] = (iy < 0 || iy >= in.h || ix < 0 || ix >= in.w) ? NULL
                                   : in.p + ((int64_t)(n * in.h + iy) * in.w + ix) * in.ld;
            }
        for (int g = 0; g < G; ++g)
            for (int co = 0; co < cout; co += 4) {
                const int nco = cout - co < 4 ? cout - co : 4, oc0 = g * cout + co;
                float part[16][4];
                for (int s_ = 0; s_ < S; ++s_) part[s_][0] = part[s_][1] = part[s_][2] = part[s_][3] = 0.0f;
                for (int j = 0; j < nco; ++j) part[0][j] = bias ? bias[oc0 + j] : 0.0f;
                const float *w0 = Wt + (int64_t)oc0 * Ktot, *w1 = w0 + (nco > 1 ? Ktot : 0), *w2 = w0 + (nco > 2 ? 2 * (int64_t)Ktot : 0),
                            *w3 = w0 + (nco > 3 ? 3 * (int64_t)Ktot : 0);
                for (int cb = 0; cb < ncb; ++cb) {
                    const int nc = ncnt[cb];
                    const int *ci = cidx + cb * 32;
                    for (int tp = 0; tp < taps; ++tp) {
                        if (!xp[tp]) continue;
                        const float *x = xp[tp] + g * cin;
                        const int run = run_of[cb * taps + tp], k0 = koff[cb] + tp * nc;
                        float a0 = part[run][0], a1 = part[run][1], a2 = part[run][2], a3 = part[run][3];
                        for (int i = 0; i < nc; ++i) {
                            const float xv = x[ci[i]];
                            a0 = fmaf(xv, w0[k0 + i], a0); a1 = fmaf(xv, w1[k0 + i], a1);
                            a2 = fmaf(xv, w2[k0 + i], a2); a3 = fmaf(xv, w3[k0 + i], a3);
                        }
                        part[run][0] = a0; part[run][1] = a1; part[run][2] = a2; part[run][3] = a3;
                    }
                }
                for (int j = 0; j < nco; ++j) {
                    const int oc = oc0 + j;
                    float acc = part[0][j];
                    for (int s_ = 1; s_ < S; ++s_) acc += part[s_][j];
                    if (op->res_mode == 1 && res.p) acc += res.p[m * res.ld + oc];
                    acc = orc_act(acc, op->act, slope ? slope[oc] : 0.0f);
                    if (op->res_mode == 2 && res.p) acc += res.p[m * res.ld + oc];
                    out.p[m * out.ld + oc] = acc;
                }
            }
    }
    free(Wt); free(cidx); free(ncnt); free(koff); free(run_of);
}

/* ---- Winograd F(2x2, 3x3) convolution (csm_op.flags bit 2; 3x3, stride 1, dilation 1, pad 1, groups 1) -----------------------------
 * Same result as a direct 3x3 convolution up to fp32 rounding (2.25x fewer multiplications), with its OWN fixed arithmetic -- the
 * contract the HIP kernel k_conv_wino executes bit for bit:
 *   weights   U[f][co][c] = fp32( G g G^T ), f = 4 i + j, evaluated in DOUBLE in a fixed order: rows of G applied to the kernel's rows
 *             first (r0 = g0, r1 = ((g0 + g1) + g2) * 0.5, r2 = ((g0 - g1) + g2) * 0.5, r3 = g2), then the same to the columns.  This
 *             function derives U from the NATURAL [cout][cin][3][3] weights itself; the product's host code packs its own copy.
 *   input     d[i][j] = x[2 ty - 1 + i][2 tx - 1 + j] (0 outside the image); t_0 = d_0 - d_2, t_1 = d_1 + d_2, t_2 = d_2 - d_1,
 *             t_3 = d_1 - d_3 over the ROW index i first, then the same over the column index j: V[i][j], one fp32 operation each.
 *   products  M[f] = fmaf chain over the input channels starting at 0.0f, in the direct convolution's channel order (32-channel blocks
 *             ascending, aligned 8-blocks in the order 0,4,1,5,2,6,3,7).
 *   output    over j first: s[i][0] = (M[i][0] + M[i][1]) + M[i][2], s[i][1] = (M[i][1] - M[i][2]) - M[i][3]; then over i:
 *             Y[0][b] = (s[0][b] + s[1][b]) + s[2][b], Y[1][b] = (s[1][b] - s[2][b]) - s[3][b]; y = Y + bias; residual / activation as in
 *             orc_conv.  Output pixel (2 ty + a, 2 tx + b); tiles cover ceil(h / 2) x ceil(w / 2), the excess is dropped.
 * The transforms contain only 0, +-1, +-0.5: |error| vs the direct chain is a few fp32 ulps of the largest partial sum. */
static void orc_wino_weights(const float *W, int cout, int cin, const int *cidx, float *U /* [cout][cin in CHAIN order][16] */)
{
#pragma omp parallel for schedule(static)
    for (int co = 0; co < cout; ++co)
        for (int k = 0; k < cin; ++k) {
            const float *g = W + ((int64_t)co * cin + cidx[k]) * 9;
            double r[4][3], u[4][4];
            for (int x = 0; x < 3; ++x) {
                const double g0 = g[x], g1 = g[3 + x], g2 = g[6 + x];
                r[0][x] = g0; r[1][x] = ((g0 + g1) + g2) * 0.5; r[2][x] = ((g0 - g1) + g2) * 0.5; r[3][x] = g2;
            }
            for (int i = 0; i < 4; ++i) {
                const double g0 = r[i][0], g1 = r[i][1], g2 = r[i][2];
                u[i][0] = g0; u[i][1] = ((g0 + g1) + g2) * 0.5; u[i][2] = ((g0 - g1) + g2) * 0.5; u[i][3] = g2;
            }
            for (int f = 0; f < 16; ++f) U[((int64_t)co * cin + k) * 16 + f] = (float)u[f >> 2][f & 3];
        }
}

static void orc_conv_wino(const csm_op *op, view_t in, view_t res, view_t out, const float *W, const float *bias, const float *slope)
{
    const int cin = op->cin_g, cout = op->cout_g, ncb = (cin + 31) / 32;
    const int ty_n = (out.h + 1) / 2, tx_n = (out.w + 1) / 2;
    int *cidx = (int *)malloc(sizeof(int) * (size_t)ncb * 32);
    int K = 0;
    for (int cb = 0; cb < ncb; ++cb)
        for (int kb = cb * 32; kb < cb * 32 + 32 && kb < cin; kb += 8)
            for (int t = 0; t < 4; ++t)
                for (int h = 0; h < 2; ++h) { int c = kb + 4 * h + t; if (c < cin) cidx[K++] = c; }
    float *U = (float *)malloc(sizeof(float) * (size_t)16 * cout * cin);
    orc_wino_weights(W, cout, cin, cidx, U);
    const int64_t T = (int64_t)out.n * ty_n * tx_n;
#pragma omp parallel
    {
        float *V = (float *)malloc(sizeof(float) * (size_t)16 * cin);
#pragma omp for schedule(dynamic, 16)
        for (int64_t tile = 0; tile < T; ++tile) {
            const int n = (int)(tile / ((int64_t)ty_n * tx_n)), rem = (int)(tile - (int64_t)n * ty_n * tx_n);
            const int ty = rem / tx_n, tx = rem - ty * tx_n;
            const float *xp[16];
            for (int i = 0; i < 4; ++i)
                for (int j = 0; j < 4; ++j) {
                    const int iy = 2 * ty - 1 + i, ix = 2 * tx - 1 + j;
                    xp[4 * i + j] = (iy < 0 || iy >= in.h || ix < 0 || ix >= in.w) ? NULL : in.p + ((int64_t)(n * in.h + iy) * in.w + ix) * in.ld;
                }
            for (int k = 0; k < cin; ++k) {
                const int c = cidx[k];
                float d[4][4], t[4][4];
                for (int q = 0; q < 16; ++q) d[q >> 2][q & 3] = xp[q] ? xp[q][c] : 0.0f;
                for (int j = 0; j < 4; ++j) {
                    t[0][j] = d[0][j] - d[2][j]; t[1][j] = d[1][j] + d[2][j]; t[2][j] = d[2][j] - d[1][j]; t[3][j] = d[1][j] - d[3][j];
                }
                for (int i = 0; i < 4; ++i) {
                    V[k * 16 + 4 * i + 0] = t[i][0] - t[i][2]; V[k * 16 + 4 * i + 1] = t[i][1] + t[i][2];
                    V[k * 16 + 4 * i + 2] = t[i][2] - t[i][1]; V[k * 16 + 4 * i + 3] = t[i][1] - t[i][3];
                }
            }
            for (int co0 = 0; co0 < cout; co0 += 4) {
              /* four output channels x 16 frequencies = 64 independent chains (vector lanes = frequencies): same chain per (f, co) */
              float m4[4][16] __attribute__((aligned(32)));
              const int nco = cout - co0 < 4 ? cout - co0 : 4;
              for (int q = 0; q < 4; ++q) for (int f = 0; f < 16; ++f) m4[q][f] = 0.0f;
              const float *u0 = U + (int64_t)co0 * cin * 16, *u1 = u0 + (nco > 1 ? (int64_t)cin * 16 : 0),
                          *u2 = u0 + (nco > 2 ? 2 * (int64_t)cin * 16 : 0), *u3 = u0 + (nco > 3 ? 3 * (int64_t)cin * 16 : 0);
              for (int k = 0; k < cin; ++k) {
                  const float *v = V + k * 16;
#pragma omp simd
                  for (int f = 0; f < 16; ++f) {
                      m4[0][f] = fmaf(v[f], u0[k * 16 + f], m4[0][f]); m4[1][f] = fmaf(v[f], u1[k * 16 + f], m4[1][f]);
                      m4[2][f] = fmaf(v[f], u2[k * 16 + f], m4[2][f]); m4[3][f] = fmaf(v[f], u3[k * 16 + f], m4[3][f]);
                  }
              }
              for (int q = 0; q < nco; ++q) {
                const int co = co0 + q;
                const float *m = m4[q];
                float s[4][2], Y[2][2];
                for (int i = 0; i < 4; ++i) {
                    s[i][0] = (m[4 * i] + m[4 * i + 1]) + m[4 * i + 2];
                    s[i][1] = (m[4 * i + 1] - m[4 * i + 2]) - m[4 * i + 3];
                }
                for (int b = 0; b < 2; ++b) { Y[0][b] = (s[0][b] + s[1][b]) + s[2][b]; Y[1][b] = (s[1][b] - s[2][b]) - s[3][b]; }
                for (int a_ = 0; a_ < 2; ++a_)
                    for (int b = 0; b < 2; ++b) {
                        const int oy = 2 * ty + a_, ox = 2 * tx + b;
                        if (oy >= out.h || ox >= out.w) continue;
                        const int64_t mrow = ((int64_t)n * out.h + oy) * out.w + ox;
                        float acc = Y[a_][b] + (bias ? bias[co] : 0.0f);
                        if (op->res_mode == 1 && res.p) acc += res.p[mrow * res.ld + co];
                        acc = orc_act(acc, op->act, slope ? slope[co] : 0.0f);
                        if (op->res_mode == 2 && res.p) acc += res.p[mrow * res.ld + co];
                        out.p[mrow * out.ld + co] = acc;
                    }
              }
            }
        }
        free(V);
    }
    free(U); free(cidx);
}

/* ---- Winograd F(4x4, 3x3) convolution (csm_op.flags bit 3, CSM_CONV_FLAG_WINOGRAD4; same layer class as F(2x2)) -------------------------
 * 36 multiplications per 4x4 output tile and channel pair (direct: 144, F(2x2): 64).  Lavin & Gray's matrices for the points
 * 0, +-1, +-2, inf; its OWN fixed arithmetic -- the contract csrc/wino4.hip::k_conv_wino4 executes bit for bit (include/csm355.h
 * "Winograd F(4x4) contract"):
 *   weights   U[f = 6 i + j][co][c] = fp32( G g G^T ) in DOUBLE, rows of G applied to the kernel's rows first, then to the columns:
 *             r0 = g0 * 0.25, r1 = -((g0 + g1) + g2) / 6, r2 = -((g0 - g1) + g2) / 6, r3 = ((g0 + 2 g1) + 4 g2) / 24,
 *             r4 = ((g0 - 2 g1) + 4 g2) / 24, r5 = g2  (IEEE double division by 6.0 / 24.0).
 *   input     d[i][j] = x[4 ty - 1 + i][4 tx - 1 + j] (0 outside); the 1-D transform T6 (B^T), fp32, fmaf where written:
 *                 t0 = fmaf(4, d0, fmaf(-5, d2, d4));            t5 = fmaf(4, d1, fmaf(-5, d3, d5));
 *                 a = fmaf(-4, d2, d4); b = fmaf(-4, d1, d3);    t1 = a + b;  t2 = a - b;
 *                 c = d4 - d2;          e = d3 - d1;             t3 = fmaf(2, e, c);  t4 = fmaf(-2, e, c);
 *             over the ROW index first, then over the column index: V[i][j].
 *   products  M[f] = one fmaf chain per (f, tile, co) over the input channels from 0.0f, channel order of the direct contract.
 *   output    the 1-D transform O6 (A^T):  p = m1 + m2; q = m1 - m2; r = m3 + m4; t = m3 - m4;
 *                 s0 = (m0 + p) + r;  s1 = fmaf(2, t, q);  s2 = fmaf(4, r, p);  s3 = fmaf(8, t, q) + m5
 *             over j first (s[i][b]), then over i (Y[a][b]); y = Y + bias; residual / activation as in orc_conv.  Output pixel
 *             (4 ty + a, 4 tx + b); tiles cover ceil(h / 4) x ceil(w / 4), the excess is dropped.
 * Error: the transforms amplify (entries up to 8, 1/24): measured 4-5x the direct chain's fp32 error. */
static inline void orc_t6(const float d0, const float d1, const float d2, const float d3, const float d4, const float d5, float *t, int st)
{
    t[0] = fmaf(4.0f, d0, fmaf(-5.0f, d2, d4));
    t[5 * st] = fmaf(4.0f, d1, fmaf(-5.0f, d3, d5));
    const float a = fmaf(-4.0f, d2, d4), b = fmaf(-4.0f, d1, d3);
    t[1 * st] = a + b; t[2 * st] = a - b;
    const float c = d4 - d2, e = d3 - d1;
    t[3 * st] = fmaf(2.0f, e, c); t[4 * st] = fmaf(-2.0f, e, c);
}
static inline void orc_o6(const float m0, const float m1, const float m2, const float m3, const float m4, const float m5, float *s, int st)
{
    const float p = m1 + m2, q = m1 - m2, r = m3 + m4, t = m3 - m4;
    s[0] = (m0 + p) + r;
    s[1 * st] = fmaf(2.0f, t, q);
    s[2 * st] = fmaf(4.0f, r, p);
    s[3 * st] = fmaf(8.0f, t, q) + m5;
}
static inline void orc_g6(const double g0, const double g1, const double g2, double *r, int st)
{
    r[0] = g0 * 0.25;
    r[1 * st] = -((g0 + g1) + g2) / 6.0;
    r[2 * st] = -((g0 - g1) + g2) / 6.0;
    r[3 * st] = ((g0 + 2.0 * g1) + 4.0 * g2) / 24.0;
    r[4 * st] = ((g0 - 2.0 * g1) + 4.0 * g2) / 24.0;
    r[5 * st] = g2;
}
static void orc_wino4_weights(const float *W, int cout, int cin, const int *cidx, float *U /* [cout][cin in CHAIN order][36] */)
{
#pragma omp parallel for schedule(static)
    for (int co = 0; co < cout; ++co)
        for (int k = 0; k < cin; ++k) {
            const float *g = W + ((int64_t)co * cin + cidx[k]) * 9;
            double r[6][3], u[6][6];
            for (int x = 0; x < 3; ++x) orc_g6(g[x], g[3 + x], g[6 + x], &r[0][x], 3);
            for (int i = 0; i < 6; ++i) orc_g6(r[i][0], r[i][1], r[i][2], &u[i][0], 1);
            for (int f = 0; f < 36; ++f) U[((int64_t)co * cin + k) * 36 + f] = (float)u[f / 6][f % 6];
        }
}

static void orc_conv_wino4(const csm_op *op, view_t in, view_t res, view_t out, const float *W, const float *bias, const float *slope)
{
    const int cin = op->cin_g, cout = op->cout_g, ncb = (cin + 31) / 32;
    const int ty_n = (out.h + 3) / 4, tx_n = (out.w + 3) / 4;
    int *cidx = (int *)malloc(sizeof(int) * (size_t)ncb * 32);
    int K = 0;
    for (int cb = 0; cb < ncb; ++cb)
        for (int kb = cb * 32; kb < cb * 32 + 32 && kb < cin; kb += 8)
            for (int t = 0; t < 4; ++t)
                for (int h = 0; h < 2; ++h) { int c = kb + 4 * h + t; if (c < cin) cidx[K++] = c; }
    float *U = (float *)malloc(sizeof(float) * (size_t)36 * cout * cin);
    orc_wino4_weights(W, cout, cin, cidx, U);
    const int64_t T = (int64_t)out.n * ty_n * tx_n;
#pragma omp parallel
    {
        float *V = (float *)malloc(sizeof(float) * (size_t)36 * cin);
#pragma omp for schedule(dynamic, 8)
        for (int64_t tile = 0; tile < T; ++tile) {
            const int n = (int)(tile / ((int64_t)ty_n * tx_n)), rem = (int)(tile - (int64_t)n * ty_n * tx_n);
            const int ty = rem / tx_n, tx = rem - ty * tx_n;
            const float *xp[36];
            for (int i = 0; i < 6; ++i)
                for (int j = 0; j < 6; ++j) {
                    const int iy = 4 * ty - 1 + i, ix = 4 * tx - 1 + j;
                    xp[6 * i + j] = (iy < 0 || iy >= in.h || ix < 0 || ix >= in.w) ? NULL : in.p + ((int64_t)(n * in.h + iy) * in.w + ix) * in.ld;
                }
            for (int k = 0; k < cin; ++k) {
                const int c = cidx[k];
                float d[6][6], t[6][6];
                for (int q = 0; q < 36; ++q) d[q / 6][q % 6] = xp[q] ? xp[q][c] : 0.0f;
                for (int j = 0; j < 6; ++j) orc_t6(d[0][j], d[1][j], d[2][j], d[3][j], d[4][j], d[5][j], &t[0][j], 6);      /* rows */
                for (int i = 0; i < 6; ++i) orc_t6(t[i][0], t[i][1], t[i][2], t[i][3], t[i][4], t[i][5], V + k * 36 + 6 * i, 1);   /* columns */
            }
            for (int co0 = 0; co0 < cout; co0 += 4) {
              float m4[4][40] __attribute__((aligned(32)));
              const int nco = cout - co0 < 4 ? cout - co0 : 4;
              for (int q = 0; q < 4; ++q) for (int f = 0; f < 36; ++f) m4[q][f] = 0.0f;
              const float *u0 = U + (int64_t)co0 * cin * 36, *u1 = u0 + (nco > 1 ? (int64_t)cin * 36 : 0),
                          *u2 = u0 + (nco > 2 ? 2 * (int64_t)cin * 36 : 0), *u3 = u0 + (nco > 3 ? 3 * (int64_t)cin * 36 : 0);
              for (int k = 0; k < cin; ++k) {
                  const float *v = V + k * 36;
#pragma omp simd
                  for (int f = 0; f < 36; ++f) {
                      m4[0][f] = fmaf(v[f], u0[k * 36 + f], m4[0][f]); m4[1][f] = fmaf(v[f], u1[k * 36 + f], m4[1][f]);
                      m4[2][f] = fmaf(v[f], u2[k * 36 + f], m4[2][f]); m4[3][f] = fmaf(v[f], u3[k * 36 + f], m4[3][f]);
                  }
              }
              for (int q = 0; q < nco; ++q) {
                const int co = co0 + q;
                const float *m = m4[q];
                float s[6][4], Y[4][4];
                for (int i = 0; i < 6; ++i) orc_o6(m[6 * i], m[6 * i + 1], m[6 * i + 2], m[6 * i + 3], m[6 * i + 4], m[6 * i + 5], &s[i][0], 1);
                for (int b = 0; b < 4; ++b) orc_o6(s[0][b], s[1][b], s[2][b], s[3][b], s[4][b], s[5][b], &Y[0][b], 4);
                for (int a_ = 0; a_ < 4; ++a_)
                    for (int b = 0; b < 4; ++b) {
                        const int oy = 4 * ty + a_, ox = 4 * tx + b;
                        if (oy >= out.h || ox >= out.w) continue;
                        const int64_t mrow = ((int64_t)n * out.h + oy) * out.w + ox;
                        float acc = Y[a_][b] + (bias ? bias[co] : 0.0f);
                        if (op->res_mode == 1 && res.p) acc += res.p[mrow * res.ld + co];
                        acc = orc_act(acc, op->act, slope ? slope[co] : 0.0f);
                        if (op->res_mode == 2 && res.p) acc += res.p[mrow * res.ld + co];
                        out.p[mrow * out.ld + co] = acc;
                    }
              }
            }
        }
        free(V);
    }
    free(U); free(cidx);
}

/* depthwise: weights natural [c][kh][kw] */
static void orc_dwconv(const csm_op *op, view_t in, view_t out, const float *W, const float *bias, const float *slope)
{
    const int64_t M = (int64_t)out.n * out.h * out.w;
#pragma omp parallel for schedule(static)
    for (int64_t m = 0; m < M; ++m) {
        int n = (int)(m / ((int64_t)out.h * out.w));
        int rem = (int)(m - (int64_t)n * out.h * out.w);
        int oy = rem / out.w, ox = rem - oy * out.w;
        for (int c = 0; c < out.c; ++c) {
            float acc = bias ? bias[c] : 0.0f;
            for (int ky = 0; ky < op->kh; ++ky) {
                int iy = oy * op->stride - op->pad + ky * op->dil;
                if (iy < 0 || iy >= in.h) continue;
                for (int kx = 0; kx < op->kw; ++kx) {
                    int ix = ox * op->stride - op->pad + kx * op->dil;
                    if (ix < 0 || ix >= in.w) continue;
                    acc = fmaf(in.p[((int64_t)(n * in.h + iy) * in.w + ix) * in.ld + c],
                               W[((int64_t)c * op->kh + ky) * op->kw + kx], acc);
                }
            }
            out.p[m * out.ld + c] = orc_act(acc, op->act, slope ? slope[c] : 0.0f);
        }
    }
}

static void orc_maxpool(const csm_op *op, view_t in, view_t out)
{
    const int64_t M = (int64_t)out.n * out.h * out.w;
#pragma omp parallel for schedule(static)
    for (int64_t m = 0; m < M; ++m) {
        int n = (int)(m / ((int64_t)out.h * out.w));
        int rem = (int)(m - (int64_t)n * out.h * out.w);
        int oy = rem / out.w, ox = rem - oy * out.w;
        for (int c = 0; c < out.c; ++c) {
            float mx = -INFINITY;
            for (int ky = 0; ky < op->kh; ++ky) {
                int iy = oy * op->stride - op->pad + ky;
                if (iy < 0 || iy >= in.h) continue;
                for (int kx = 0; kx < op->kh; ++kx) {
                    int ix = ox * op->stride - op->pad + kx;
                    if (ix < 0 || ix >= in.w) continue;
                    mx = fmaxf(mx, in.p[((int64_t)(n * in.h + iy) * in.w + ix) * in.ld + c]);
                }
            }
            out.p[m * out.ld + c] = mx;
        }
    }
}

/* aten/src/ATen/native/UpSample.h: area_pixel_compute_source_index + guard_index_and_lambda */
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

static void orc_bilinear(const csm_op *op, view_t in, view_t out, const float *slope)
{
    int align = op->flags & 1;
    float sh, sw;
    if (align) { sh = out.h > 1 ? (float)(in.h - 1) / (float)(out.h - 1) : 0.0f; sw = out.w > 1 ? (float)(in.w - 1) / (float)(out.w - 1) : 0.0f; }
    else { sh = (float)in.h / (float)out.h; sw = (float)in.w / (float)out.w; }
    const int64_t M = (int64_t)out.n * out.h * out.w;
#pragma omp parallel for schedule(static)
    for (int64_t m = 0; m < M; ++m) {
        int n = (int)(m / ((int64_t)out.h * out.w));
        int rem = (int)(m - (int64_t)n * out.h * out.w);
        int oy = rem / out.w, ox = rem - oy * out.w;
        int y0, y1, x0, x1; float hl0, hl1, wl0, wl1;
        src_index(oy, in.h, out.h, sh, align, &y0, &y1, &hl0, &hl1);
        src_index(ox, in.w, out.w, sw, align, &x0, &x1, &wl0, &wl1);
        const float *P = in.p + (int64_t)n * in.h * in.w * in.ld;
        for (int c = 0; c < out.c; ++c) {
            float p00 = P[((int64_t)y0 * in.w + x0) * in.ld + c], p01 = P[((int64_t)y0 * in.w + x1) * in.ld + c];
            float p10 = P[((int64_t)y1 * in.w + x0) * in.ld + c], p11 = P[((int64_t)y1 * in.w + x1) * in.ld + c];
            out.p[m * out.ld + c] = orc_act(hl0 * (wl0 * p00 + wl1 * p01) + hl1 * (wl0 * p10 + wl1 * p11), op->act, slope ? slope[c] : 0.0f);
        }
    }
}

static void orc_nearest(view_t in, view_t out)
{
    int fy = out.h / in.h, fx = out.w / in.w;
    const int64_t M = (int64_t)out.n * out.h * out.w;
    for (int64_t m = 0; m < M; ++m) {
        int n = (int)(m / ((int64_t)out.h * out.w));
        int rem = (int)(m - (int64_t)n * out.h * out.w);
        int oy = rem / out.w, ox = rem - oy * out.w;
        memcpy(out.p + m * out.ld, in.p + ((int64_t)(n * in.h + oy / fy) * in.w + ox / fx) * in.ld, (size_t)out.c * 4);
    }
}

static void orc_eltwise(view_t a, view_t b, view_t out, int act, int mode, const float *slope)
{
    const int64_t M = (int64_t)out.n * out.h * out.w;
    for (int64_t m = 0; m < M; ++m) {
        int64_t n = m / ((int64_t)out.h * out.w);
        /* the first operand of an add may be up to one row / column larger than the output (torch's negative pad crop) */
        const int64_t r = m - n * (int64_t)out.h * out.w;
        const int64_t ma = (n * a.h + r / out.w) * a.w + r % out.w;
        for (int c = 0; c < out.c; ++c) {
            float v = a.p[(mode == 1 ? ma : m) * a.ld + c];
            if (mode == 1) v = v + b.p[m * b.ld + c];
            else if (mode == 2) v = v * b.p[n * b.ld + c];
            out.p[m * out.ld + c] = orc_act(v, act, slope ? slope[c] : 0.0f);
        }
    }
}

/* ZoeDepth attractor update (depth_modules/zoedepth/models/layers/attractor.py:117-208, the memory_efficient loop) */
static void orc_attractor(const csm_op *op, view_t A, view_t b, view_t out, const float *par)
{
    const float alpha = par[0];
    const int64_t M = (int64_t)out.n * out.h * out.w;
#pragma omp parallel for schedule(static)
    for (int64_t m = 0; m < M; ++m)
        for (int k = 0; k < out.c; ++k) {
            const float c = b.p[m * b.ld + k];
            float delta = 0.0f;
            for (int i = 0; i < A.c; ++i) {
                const float dx = A.p[m * A.ld + i] - c;
                float d;
                if (op->flags & 1) d = orc_expf(-alpha * (fabsf(dx) * fabsf(dx))) * dx;
                else d = dx / (1.0f + alpha * (dx * dx));
                delta += d;
            }
            if (op->flags & 2) delta = delta / (float)A.c;
            out.p[m * out.ld + k] = c + delta;
        }
}

/* ConditionalLogBinomial tail + weighted sum (dist_layers.py:46-121, zoedepth_v1.py:196-199) */
static void orc_logbinom(view_t pt, view_t cen, view_t out, const float *par)
{
    const float p_eps = par[0], min_temp = par[1], max_temp = par[2];
    const float *lb = par + 3;
    const int K = cen.c;
    const int64_t M = (int64_t)out.n * out.h * out.w;
#pragma omp parallel for schedule(static)
    for (int64_t m = 0; m < M; ++m) {
        const float *q = pt.p + m * pt.ld;
        const float p0 = q[0] + p_eps, p1 = q[1] + p_eps, t0 = q[2] + p_eps, t1 = q[3] + p_eps;
        const float p = p0 / (p0 + p1);
        float t = t0 / (t0 + t1);
        t = (max_temp - min_temp) * t + min_temp;
        const float eps = 1e-4f;
        const float omx = fminf(fmaxf(1.0f - p, eps), 1.0f), x = fminf(fmaxf(p, eps), 1.0f);
        const float lx = orc_logf(x), lo = orc_logf(omx);
        const float *c = cen.p + m * cen.ld;
        float mx = -INFINITY;
        for (int k = 0; k < K; ++k) {
            const float y = (lb[k] + (float)k * lx + (float)(K - 1 - k) * lo) / t;
            mx = fmaxf(mx, y);
        }
        float den = 0.0f, num = 0.0f;
        for (int k = 0; k < K; ++k) {
            const float y = (lb[k] + (float)k * lx + (float)(K - 1 - k) * lo) / t;
            const float e = orc_expf(y - mx);
            den += e; num += e * c[k];
        }
        out.p[m * out.ld] = num / den;
    }
}

/* same fixed reduction tree as k_gavgpool: 256 strided sequential partials, then 128,64,..,1 */
static void orc_gavgpool(view_t in, view_t out)
{
    int hw = in.h * in.w;
    for (int n = 0; n < in.n; ++n)
        for (int c = 0; c < in.c; ++c) {
            float part[256];
            const float *P = in.p + (int64_t)n * hw * in.ld + c;
            for (int t = 0; t < 256; ++t) {
                float s = 0.0f;
                for (int i = t; i < hw; i += 256) s += P[(int64_t)i * in.ld];
                part[t] = s;
            }
            for (int st = 128; st >= 1; st >>= 1)
                for (int t = 0; t < st; ++t) part[t] += part[t + st];
            out.p[(int64_t)n * out.ld + c] = part[0] / (float)hw;
        }
}

/* ---- transformer ops of the MiDaS DPT-BEiT core (timm 0.6.x models/beit.py Attention / Block, MiDaS 3.1 midas/backbones/beit.py and
 * utils.py; restated from the published definitions -- the reference pulls them through torch.hub, base_models/midas.py:341).  Reductions
 * in double: this is the high-precision side of a tolerance-level comparison. */
static void orc_layernorm(view_t in, view_t out, const float *gamma, const float *beta, float eps)
{
    int64_t rows = (int64_t)in.n * in.h * in.w;
#pragma omp parallel for
    for (int64_t r = 0; r < rows; ++r) {
        const float *x = in.p + r * in.ld;
        double s = 0.0, q = 0.0;
        for (int c = 0; c < in.c; ++c) s += x[c];
        double mean = s / in.c;
        for (int c = 0; c < in.c; ++c) { double d = x[c] - mean; q += d * d; }
        double rstd = 1.0 / sqrt(q / in.c + (double)eps);
        for (int c = 0; c < in.c; ++c) out.p[r * out.ld + c] = (float)((x[c] - mean) * rstd * gamma[c] + beta[c]);
    }
}

/* timm Attention.forward after the qkv projection (q pre-scaled by the lowering): attn = q k^T + relative_position_bias -> softmax -> attn v.
 * bias index = timm gen_relative_position_index for a (gh, gw) window with the class token first. */
static void orc_attention(const csm_op *op, view_t in, view_t out, const float *table)
{
    const int heads = op->groups, d = op->cin_g, N = in.h, C = heads * d, gh = op->kh, gw = op->kw;
    const int T = (2 * gh - 1) * (2 * gw - 1) + 3;
#pragma omp parallel for collapse(2)
    for (int b = 0; b < in.n; ++b)
        for (int h = 0; h < heads; ++h) {
            double *sc = (double *)malloc(sizeof(double) * (size_t)N);
            for (int i = 0; i < N; ++i) {
                const float *q = in.p + ((int64_t)b * N + i) * in.ld + h * d;
                double mx = -1e300;
                for (int j = 0; j < N; ++j) {
                    const float *k = in.p + ((int64_t)b * N + j) * in.ld + C + h * d;
                    double s = 0.0;
                    for (int e = 0; e < d; ++e) s += (double)q[e] * (double)k[e];
                    if (table) {
                        int idx;
                        if (i == 0) idx = j == 0 ? T - 1 : T - 3;
                        else if (j == 0) idx = T - 2;
                        else {
                            int yi = (i - 1) / gw, xi = (i - 1) % gw, yj = (j - 1) / gw, xj = (j - 1) % gw;
                            idx = (yi - yj + gh - 1) * (2 * gw - 1) + (xi - xj + gw - 1);
                        }
                        s += table[(int64_t)idx * heads + h];
                    }
                    sc[j] = s;
                    if (s > mx) mx = s;
                }
                double sum = 0.0;
                for (int j = 0; j < N; ++j) { sc[j] = exp(sc[j] - mx); sum += sc[j]; }
                for (int e = 0; e < d; ++e) {
                    double o = 0.0;
                    for (int j = 0; j < N; ++j) o += sc[j] * (double)in.p[((int64_t)b * N + j) * in.ld + 2 * C + h * d + e];
                    out.p[((int64_t)b * N + i) * out.ld + h * d + e] = (float)(o / sum);
                }
            }
            free(sc);
        }
}

static void orc_tokens(const csm_op *op, view_t in, view_t out, const float *cls)
{
    const int mode = op->flags;
    if (mode == 0) {
        int np = in.h * in.w;
        for (int b = 0; b < in.n; ++b)
            for (int t = 0; t <= np; ++t)
                for (int c = 0; c < in.c; ++c)
                    out.p[((int64_t)b * (np + 1) + t) * out.ld + c] = t == 0 ? cls[c] : in.p[((int64_t)b * np + t - 1) * in.ld + c];
    } else {
        int np = out.h * out.w;
        for (int b = 0; b < in.n; ++b)
            for (int i = 0; i < np; ++i) {
                float *o = out.p + ((int64_t)b * np + i) * out.ld;
                for (int c = 0; c < in.c; ++c) o[c] = in.p[((int64_t)b * (np + 1) + 1 + i) * in.ld + c];
                if (mode == 1)
                    for (int c = 0; c < in.c; ++c) o[in.c + c] = in.p[((int64_t)b * (np + 1)) * in.ld + c];
            }
    }
}

static void orc_depth_to_space(const csm_op *op, view_t in, view_t out)
{
    const int k = op->stride, C = out.c;
    for (int b = 0; b < in.n; ++b)
        for (int y = 0; y < in.h; ++y)
            for (int x = 0; x < in.w; ++x)
                for (int ky = 0; ky < k; ++ky)
                    for (int kx = 0; kx < k; ++kx)
                        for (int c = 0; c < C; ++c)
                            out.p[(((int64_t)b * out.h + y * k + ky) * out.w + x * k + kx) * out.ld + c] =
                                in.p[(((int64_t)b * in.h + y) * in.w + x) * in.ld + (ky * k + kx) * C + c];
}

int orc_run_program(const csm_op *ops, int n_ops, const csm_tensor_desc *tensors, int n_tensors, const float *weights,
                    float *workspace, void *const *ext, int n_ext)
{
    (void)n_tensors; (void)n_ext;
    for (int i = 0; i < n_ops; ++i) {
        const csm_op *op = &ops[i];
        view_t in, in1, out;
        get_view(tensors, op->in0, workspace, ext, &in);
        get_view(tensors, op->in1, workspace, ext, &in1);
        get_view(tensors, op->out, workspace, ext, &out);
        const float *W = op->w_off >= 0 ? weights + op->w_off : NULL;
        const float *B = op->b_off >= 0 ? weights + op->b_off : NULL;
        const float *S = op->aux_off >= 0 ? weights + op->aux_off : NULL;
        switch (op->kind) {
            case CSM_OP_CONV: {
                const char *e = getenv("ORC_CONV_REFERENCE");
                if (op->flags & CSM_CONV_FLAG_WINOGRAD) {
                    if (op->kh != 3 || op->kw != 3 || op->stride != 1 || op->dil != 1 || op->pad != 1 || op->groups != 1 || op->ksplit > 1) {
                        fprintf(stderr, "orc_run_program: op %d: Winograd flag on an ineligible convolution\n", i); return 1;
                    }
                    orc_conv_wino(op, in, in1, out, W, B, S);
                } else if (op->flags & CSM_CONV_FLAG_WINOGRAD4) {
                    if (op->kh != 3 || op->kw != 3 || op->stride != 1 || op->dil != 1 || op->pad != 1 || op->groups != 1 || op->ksplit > 1) {
                        fprintf(stderr, "orc_run_program: op %d: Winograd F(4x4) flag on an ineligible convolution\n", i); return 1;
                    }
                    orc_conv_wino4(op, in, in1, out, W, B, S);
                } else if (e && e[0] == '1') orc_conv(op, in, in1, out, W, B, S); else orc_conv_fast(op, in, in1, out, W, B, S);
                break;
            }
            case CSM_OP_DWCONV: orc_dwconv(op, in, out, W, B, S); break;
            case CSM_OP_MAXPOOL: orc_maxpool(op, in, out); break;
            case CSM_OP_BILINEAR: orc_bilinear(op, in, out, S); break;
            case CSM_OP_NEAREST: orc_nearest(in, out); break;
            case CSM_OP_ADD: orc_eltwise(in, in1, out, op->act, 1, NULL); break;
            case CSM_OP_SCALE: orc_eltwise(in, in1, out, op->act, 2, NULL); break;
            case CSM_OP_ACT: orc_eltwise(in, in1, out, op->act, 0, S); break;
            case CSM_OP_COPY: orc_eltwise(in, in1, out, 0, 0, NULL); break;
            case CSM_OP_GAVGPOOL: orc_gavgpool(in, out); break;
            case CSM_OP_ATTRACTOR: orc_attractor(op, in, in1, out, S); break;
            case CSM_OP_LOGBINOM: orc_logbinom(in, in1, out, S); break;
            case CSM_OP_LAYERNORM: orc_layernorm(in, out, W, B, S[0]); break;
            case CSM_OP_ATTENTION: orc_attention(op, in, out, S); break;
            case CSM_OP_TOKENS: orc_tokens(op, in, out, S); break;
            case CSM_OP_DEPTH_TO_SPACE: orc_depth_to_space(op, in, out); break;
            case CSM_OP_NCHW_TO_NHWC: {
                int64_t hw = (int64_t)out.h * out.w;
                for (int64_t n = 0; n < out.n; ++n)
                    for (int64_t p = 0; p < hw; ++p)
                        for (int c = 0; c < out.c; ++c)
                            out.p[(n * hw + p) * out.ld + c] = c < in.c ? in.p[(n * in.c + c) * hw + p] : 0.0f;
                break;
            }
            case CSM_OP_NHWC_TO_NCHW: {
                int64_t hw = (int64_t)in.h * in.w;
                for (int64_t n = 0; n < in.n; ++n)
                    for (int c = 0; c < in.c; ++c)
                        for (int64_t p = 0; p < hw; ++p) out.p[(n * in.c + c) * hw + p] = in.p[(n * hw + p) * in.ld + c];
                break;
            }
            default: fprintf(stderr, "orc_run_program: unknown op kind %d\n", op->kind); return 1;
        }
    }
    return 0;
}

/* test hook: the oracle's own U = G g G^T for natural-order channels, [cout][cin][16] (tests/test_oracle_winograd.py compares the
 * product's host packing with it bit for bit) */
void orc_wino_transform_weights(const float *W, int cout, int cin, float *U)
{
    int *cidx = (int *)malloc(sizeof(int) * (size_t)cin);
    for (int c = 0; c < cin; ++c) cidx[c] = c;
    orc_wino_weights(W, cout, cin, cidx, U);
    free(cidx);
}

/* the same hook for F(4x4): [cout][cin][36] */
void orc_wino4_transform_weights(const float *W, int cout, int cin, float *U)
{
    int *cidx = (int *)malloc(sizeof(int) * (size_t)cin);
    for (int c = 0; c < cin; ++c) cidx[c] = c;
    orc_wino4_weights(W, cout, cin, cidx, U);
    free(cidx);
}

size_t orc_sizeof_op(void) { return sizeof(csm_op); }
size_t orc_sizeof_tensor(void) { return sizeof(csm_tensor_desc); }
