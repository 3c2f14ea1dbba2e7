/*
 * csm355.h -- C ABI of libcsm355.so: the MI355X (gfx950) hot path of CartoonSegmentation.
 *
 * Drop-in boundary: the reference has no plugin ABI for this path; its
 * operator boundary is utils/cupy_utils.py::launch_kernel(name, src)(grid, block,
 * args=[int32 n, raw device pointers...]) (utils/cupy_utils.py:7-13) -- caller-owned,
 * pre-allocated torch buffers, raw pointers, nothing returned.  This header keeps that
 * convention: plain pointers + sizes, caller-allocated outputs, an explicit HIP stream,
 * int status (0 = ok; message via csm_last_error()).  No torch types.
 *
 * All tensors are contiguous fp32 unless stated; layouts are the reference's
 * (NCHW / [B,C,N]).  Every function is asynchronous on `stream` (a hipStream_t cast
 * to void*; NULL = the null stream) and re-entrant per stream.
 *
 * Reference citations are relative to /root/reference.
 */
#ifndef CSM355_H
#define CSM355_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define CSM_OK 0
#define CSM_ERR_ARG 1
#define CSM_ERR_HIP 2
#define CSM_ERR_NAN 3

/* thread-local message of the last non-zero status */
const char *csm_last_error(void);
/* library/ABI version (major*1000 + minor) and build info string */
int csm_version(void);
const char *csm_build_info(void);

/* ------------------------------------------------------------------------------------
 * Ken Burns warp operators
 * ---------------------------------------------------------------------------------- */

/* kernel_pointrender_updateZee   anime_3dkenburns/models/utils.py:63-149
 * pts [B,3,N]; zee [B,1,H,W] must hold 1e6 on entry (models/utils.py:59). */
int csm_pointrender_update_zee(const float *pts, int B, int64_t N, int H, int W, double focal,
                               double baseline, float *zee, void *stream);

/* kernel_pointrender_updateDegrid   models/utils.py:152-212
 * Deterministic (Jacobi) form: reads zee_in, writes zee_out (must not alias). */
int csm_pointrender_degrid(const float *zee_in, float *zee_out, int B, int H, int W, void *stream);

/* kernel_pointrender_updateOutput   models/utils.py:215-313
 * data [B,C,N] WITHOUT the ones channel; accum [B,C+1,H,W] must be zero on entry;
 * channel C of accum receives the bilinear weight itself (the reference's appended
 * ones channel, models/utils.py:57). */
int csm_pointrender_update_output(const float *pts, const float *data, const float *zee, int B, int C,
                                  int64_t N, int H, int W, double focal, double baseline, float *accum,
                                  void *stream);

/* render_pointcloud   models/utils.py:56-315  (fill + 3 kernels + divide)
 * scratch: zee_scratch 2*B*H*W floats, accum_scratch B*(C+1)*H*W floats.
 * outputs: render [B,C,H,W], existing [B,1,H,W]. */
int csm_render_pointcloud(const float *pts, const float *data, int B, int C, int64_t N, int W, int H,
                          double focal, double baseline, float *zee_scratch, float *accum_scratch,
                          float *render, float *existing, void *stream);

/* The same operator for ONE cloud (B = 1) and any channel count on the destination-tile path (warptile.hip): binning, per-tile
 * z-buffer / degrid / z-test in LDS, the channels splatted in groups of 8 through 64-bit fixed-point LDS accumulators (|data| < 2^30):
 * deterministic sums, ~4x faster than the L2 float atomics at the 68 channels Inpaint.forward splats (pointcloud_inpainting.py:135).
 * scratch: csm_warp_tile_scratch_bytes(H, W, N) bytes (a csm_warp_frame_tiled scratch serves) whose first csm_warp_tile_header_bytes(H, W)
 * are zeroed once by the caller. */
int csm_render_pointcloud_tiled(const float *pts, const float *data, int C, int64_t N, int W, int H, double focal, double baseline,
                                void *scratch, float *render, float *existing, void *stream);

/* fill_disocclusion   anime_3dkenburns/common.py:145-248
 * in [B,C,H,W], depth [B,1,H,W] -> out [B,C,H,W] (out is fully written; no pre-clone needed).
 * scratch: csm_fill_disocclusion_scratch_bytes(B,H,W) bytes of device memory (hole list + valid map). */
size_t csm_fill_disocclusion_scratch_bytes(int B, int H, int W);
int csm_fill_disocclusion(const float *in, const float *depth, float *out, int B, int C, int H, int W,
                          void *scratch, void *stream);

/* spatial_filter(x,'laplacian')   models/utils.py:12-24 ; x,out [BC,H,W] */
int csm_spatial_filter_laplacian(const float *in, float *out, int BC, int H, int W, void *stream);

/* spatial_filter(x,'median-5')   models/utils.py:32-36 (reflect pad, lower median of 25) ; x,out [BC,H,W]
 * A window that contains a NaN yields NaN (torch.median's rule); the same holds for median-3. */
int csm_spatial_filter_median5(const float *in, float *out, int BC, int H, int W, void *stream);

/* spatial_filter(x,'median-3')   models/utils.py:26-30 (reflect pad 1, lower median of 9) ; x,out [BC,H,W] */
int csm_spatial_filter_median3(const float *in, float *out, int BC, int H, int W, void *stream);

/* depth_to_points   models/utils.py:43-50 ; depth [B,1,H,W] -> pts [B,3,H,W] */
int csm_depth_to_points(const float *depth, float *pts, int B, int H, int W, double focal, void *stream);

/* kenburns_effect.py:928-933 fused: normalised disparity [1,1,H,W] (already /max*baseline) ->
 * depth = (1/(disp+eps))*focal*baseline, valid, points (depth*valid), unaltered points.  disp_max -> ONE float in device
 * memory holding max(disparity) (a device pointer so the caller needs no host sync);
 * eps = 1e-5 at kenburns_effect.py:929, 1e-7 at :458 and pointcloud_inpainting.py:117. */
int csm_disparity_to_points(const float *disp, const float *disp_max, int H, int W, double focal, double baseline, float eps,
                            float *depth, float *valid, float *pts, float *unaltered, void *stream);

/* tensor part of process_shift   common.py:74-81 ; shift = float32(sx,sy,sz) */
int csm_process_shift(const float *pts, float *out, int B, int64_t N, float sx, float sy, float sz,
                      void *stream);

/* One output frame of KenBurnsPipeline.process_kenburns (kenburns_effect.py:1027-1040), fused:
 * process_shift -> render_pointcloud(cat[rgb, depth]) -> fill_disocclusion(render,
 * render[3]*(existing>0)) -> uint8 HWC frame.
 * pts [1,3,N], rgb [1,3,N], depth [1,1,N]; scratch >= csm_warp_frame_scratch_floats(H,W) floats.
 * outputs: render_filled [1,4,H,W] (may be NULL), frame_u8 [H,W,3]. */
size_t csm_warp_frame_scratch_floats(int H, int W);
int csm_warp_frame(const float *pts, const float *rgb, const float *depth, int64_t N, int H, int W,
                   double focal, double baseline, float sx, float sy, float sz, float *scratch,
                   float *render_filled, uint8_t *frame_u8, void *stream);

/* The same frame as csm_warp_frame with the splat done per destination tile in LDS (warptile.hip): points are binned by the
 * 32 x 16 tile(s) their footprint touches (integer atomics only), then one block per tile builds the z-buffer window, degrids,
 * z-tests, accumulates into 64-bit fixed-point LDS accumulators (integer atomics: order free, a frame is bit-reproducible),
 * normalises and writes the uint8 frame; holes are filled from row / column validity bitmaps.  Same decisions (z-buffer, coverage,
 * fill sources) as csm_warp_frame; colours within 2^-20 absolute of the exact sum that every fp32 summation order approximates
 * (the reference's own order is unspecified).
 * csm_warp_tile_supported(H, W): 1 when the frame has at most 8192 tiles (up to ~2048 x 2048); larger frames use csm_warp_frame.
 * scratch: csm_warp_tile_scratch_bytes(H, W, N) bytes, 16-B aligned; its first csm_warp_tile_header_bytes(H, W) bytes must be
 * ZERO before the first call (hipMemset once after allocation) -- every call leaves them re-armed for the next frame. */
size_t csm_warp_tile_scratch_bytes(int H, int W, int64_t N);
size_t csm_warp_tile_header_bytes(int H, int W);
int csm_warp_tile_supported(int H, int W);
int csm_warp_frame_tiled(const float *pts, const float *rgb, const float *depth, int64_t N, int H, int W, double focal,
                         double baseline, float sx, float sy, float sz, void *scratch, float *render_filled, uint8_t *frame_u8,
                         void *stream);

/* K frames of ONE cloud under K camera shifts (the frame loop of KenBurnsPipeline.process_kenburns, kenburns_effect.py:1027-1040) in one
 * asynchronous call on `stream`: the frames are dealt round-robin onto `lanes` (1..3) internal streams (lanes = 1: `stream` itself) with one scratch
 * each, forked from and joined to `stream` by events, so that frame k + 1's binning and frame k - 1's hole fill run under frame k's
 * render.  Every frame is bit-identical to csm_warp_frame_tiled with the same arguments.
 * shifts_host: K x (sx, sy, sz) HOST floats (read before the call returns); scratch: csm_warp_frames_scratch_bytes(H, W, N, lanes) device
 * bytes = `lanes` consecutive csm_warp_frame_tiled scratches (each rounded up to 256 B), every one with its header zeroed once by the
 * caller -- hipMemset the whole buffer once; frames_u8 [K][H][W][3]; render_filled NULL or [K][4][H][W]. */
size_t csm_warp_frames_scratch_bytes(int H, int W, int64_t N, int lanes);
int csm_warp_frames_tiled(const float *pts, const float *rgb, const float *depth, int64_t N, int H, int W, double focal, double baseline,
                          const float *shifts_host, int K, int lanes, void *scratch, float *render_filled, uint8_t *frames_u8, void *stream);

/* Coverage search of process_autozoom   anime_3dkenburns/common.py:86-142, batched.
 * For each of K candidate camera shifts (sx_k, sy_k, shift_z) -- the float32 tenShift of process_shift (common.py:74) --
 * counts[k] = number of pixels with tenExisting > 0 after process_shift + render_pointcloud of pts [1,3,N]
 * (the `(tenExisting > 0.0).float().sum()` of common.py:126), without rendering any colour: z-buffer, degrid (Jacobi form) and
 * the z-test only.  Candidates are processed `chunk` (<= csm_autozoom_max_chunk()) at a time, each with its own z-buffer plane.
 * shifts_xy: HOST pointer to K x {sx, sy}; counts: DEVICE int32 [K] (zeroed by the call; read them once after the stream
 * is synchronised); scratch: csm_autozoom_scratch_floats(H, W, chunk) device floats. */
int csm_autozoom_max_chunk(void);
size_t csm_autozoom_scratch_floats(int H, int W, int chunk);
int csm_autozoom_coverage(const float *pts, int64_t N, int H, int W, double focal, double baseline, const float *shifts_xy,
                          float shift_z, int K, int chunk, float *scratch, int *counts, void *stream);

/* The same counts through the BAND path (autozoom.hip): candidates are grouped by their y shift (host side, exact float
 * equality), the points of a group are binned once by destination row band, and one block per (band, group) evaluates all the
 * group's candidates with the band's z-buffer in LDS -- no per-candidate HBM planes.  Same arithmetic, same counts.
 * counts: DEVICE int32 [K]; overflow: DEVICE int32 [1], set non-zero when a band segment was too small for this cloud -- the
 * counts are then INVALID and the caller re-runs csm_autozoom_coverage (read it together with the counts, one transfer).
 * csm_autozoom_band_supported: 0 for frames too wide for the LDS band (then only csm_autozoom_coverage applies).
 * scratch: csm_autozoom_band_scratch_bytes(H, W, N) device bytes, 16-B aligned (no initialisation needed). */
int csm_autozoom_band_supported(int H, int W);
size_t csm_autozoom_band_scratch_bytes(int H, int W, int64_t N);
int csm_autozoom_coverage_bands(const float *pts, int64_t N, int H, int W, double focal, double baseline, const float *shifts_xy,
                                float shift_z, int K, void *scratch, int *counts, int *overflow, void *stream);

/* ------------------------------------------------------------------------------------
 * Dense networks: a flat "layer program" executed on one stream (no allocation, graph-capturable)
 *
 * The reference runs these nets through torch nn.Modules (cuDNN); the replacement boundary is the
 * module call: animeinsseg/models/animeseg_refine/isnet.py:578 (ISNetDIS.forward),
 * depth_modules/leres/leres/multi_depth_model_woauxi.py:30 (RelDepthModel), mmdet RTMDet
 * (call site animeinsseg/__init__.py:450).  The host (Python, cartoonsegmentation_amd/program.py)
 * lowers a net into csm_op records + one packed fp32 weight buffer (BN folded); this library
 * executes them with hand-written gfx950 kernels.  Activations are NHWC fp32 inside a caller-
 * provided workspace; a tensor may be a channel slice of a wider buffer (ld = channel pitch), which
 * is how torch.cat is realised without copies.
 *
 * Numerical contract (so the CPU oracle can be bit-exact): every convolution output is ONE fp32
 * fmaf chain  acc = bias; for each aligned block of 32 input channels (outer); for tap (kh,kw) row-major; for each
 * aligned sub-block of 8 channels, channel order 0,4,1,5,2,6,3,7 (the v_mfma_f32_32x32x2_f32 lane order):
 * acc = fmaf(x, w, acc)
 * -- or `ksplit` such chains over consecutive K runs, summed in order (see csm_op.ksplit; the host picks
 * ksplit > 1 for small feature maps so that all 256 CUs get work at batch 1).  ksplit follows the PER-SAMPLE shape only, so
 * a batch-n program computes, for every sample, the bits of the batch-1 program.  How the runs are executed (S blocks +
 * a reduce kernel, or one block that walks the runs and adds them in registers) is a speed choice with identical results.
 * Epilogue order: (+residual if res_mode==1) -> activation -> (+residual if res_mode==2).
 * SiLU/sigmoid use the polynomial expf documented in DESIGN.md.
 * ---------------------------------------------------------------------------------- */
enum csm_op_kind {
    CSM_OP_CONV = 1,        /* dense or grouped conv on fp32 MFMA (implicit GEMM) */
    CSM_OP_DWCONV = 2,      /* depthwise conv, direct */
    CSM_OP_MAXPOOL = 3,     /* p: k, stride, pad (ceil_mode is implied by the output size) */
    CSM_OP_BILINEAR = 4,    /* p[0]=align_corners; output size from the out tensor */
    CSM_OP_NEAREST = 5,     /* nearest upsample, integer factor from sizes */
    CSM_OP_ADD = 6,         /* out = act(in0 + in1) */
    CSM_OP_GAVGPOOL = 7,    /* global average pool -> [n,1,1,c] */
    CSM_OP_SCALE = 8,       /* out = in0 * in1[n,0,0,c] */
    CSM_OP_NCHW_TO_NHWC = 9,/* in0 = ext NCHW tensor (c real channels) -> NHWC padded to out.c (zeros) */
    CSM_OP_NHWC_TO_NCHW = 10,
    CSM_OP_ACT = 11,        /* out = act(in0) */
    CSM_OP_COPY = 12,       /* out = in0 (slice copy) */
    /* ZoeDepth metric-bins head (depth_modules/zoedepth/models/layers/attractor.py, dist_layers.py) */
    CSM_OP_ATTRACTOR = 13,  /* in0 = attractor points A [n,h,w,na], in1 = bin centers b [n,h,w,nb] -> out = b + agg_i dist(A_i - b):
                               dist = dx / (1 + alpha dx^2) (flags bit 0 = 0, "inv") or exp(-alpha dx^2) dx (bit 0 = 1, "exp");
                               agg = sum over i in order, / na when flags bit 1 (kind "mean"); aux_off -> {alpha}  (gamma = 2) */
    CSM_OP_LOGBINOM = 14,   /* in0 = pt [n,h,w,4] (softplus'ed p0,p1,t0,t1), in1 = bin centers [n,h,w,nb] -> out [n,h,w,1] =
                               sum_k softmax_k(logbinomial_k(p) / t) * centers_k  (ConditionalLogBinomial.forward tail + the
                               weighted sum of zoedepth_v1.py:199); aux_off -> {p_eps, min_temp, max_temp, lb[0..nb)} with
                               lb[k] = log_binom(nb-1, k) tabulated by the host */
    /* MiDaS DPT-BEiT core of ZoeDepth (timm 0.6.x beit_large_patch16_384 + MiDaS 3.1 dpt_depth.py / backbones/beit.py; the reference
     * fetches it with torch.hub, depth_modules/zoedepth/models/base_models/midas.py:341).  A token sequence [B, N, C] is the NHWC tensor
     * n = B, h = N, w = 1, c = C, so every nn.Linear is a 1x1 CSM_OP_CONV (exact fmaf chains, as above).  The ops below hold reductions
     * over channels / keys: their order is implementation-defined and parity with the oracle is tolerance-level (1e-5 relative), not
     * bit-exact. */
    CSM_OP_LAYERNORM = 15,  /* out[p, :] = (in0[p, :] - mean) * rsqrt(var + eps) * gamma + beta over the c channels of every pixel p
                               (biased variance); w_off -> gamma[c], b_off -> beta[c], aux_off -> {eps} */
    CSM_OP_ATTENTION = 16,  /* multi-head self-attention core.  in0 = qkv [n, N, 1, 3 * heads * d] (channels: q | k | v, each head-major;
                               q already carries the 1/sqrt(d) scale), out = [n, N, 1, heads * d]; groups = heads, cin_g = d.
                               out[i, h, :] = sum_j softmax_j(q_i . k_j + bias_h(i, j)) v_j.  BEiT's relative position bias: token 0 is the
                               class token, token 1 + y * kw + x the patch (y, x) of the kh x kw grid (N = kh * kw + 1); aux_off ->
                               table [(2 kh - 1) * (2 kw - 1) + 3][heads]: bias_h(i, j) = table[(yi - yj + kh - 1) * (2 kw - 1) + (xi - xj +
                               kw - 1)][h] between patches, rows T-3 / T-2 / T-1 for cls->patch / patch->cls / cls->cls (timm
                               gen_relative_position_index), stored HEAD-MAJOR ([heads][T]) in the device weight buffer; aux_off < 0:
                               no bias.  Keys are streamed with a running max / sum: any sequence length */
    CSM_OP_TOKENS = 17,     /* token plumbing, mode = flags: 0 "assemble" in0 = patch embedding [n, gh, gw, c] -> out [n, gh*gw + 1, 1, c],
                               row 0 = class token (aux_off -> c floats); 1 "readout project input" in0 = tokens [n, N, 1, c] ->
                               out [n, kh, kw, 2c] = (token 1 + i | class token) (MiDaS ProjectReadout's concat); 2 "readout ignore"
                               -> out [n, kh, kw, c] = token 1 + i (MiDaS Slice) */
    CSM_OP_DEPTH_TO_SPACE = 18 /* out[n, y*k + ky, x*k + kx, c] = in0[n, y, x, (ky*k + kx) * out.c + c], k = stride: the scatter half of a
                               ConvTranspose2d(kernel = stride = k) whose GEMM half is a 1x1 CSM_OP_CONV to k*k*cout channels */
};
enum csm_act { CSM_ACT_NONE = 0, CSM_ACT_RELU = 1, CSM_ACT_SILU = 2, CSM_ACT_PRELU = 3, CSM_ACT_HSIGMOID = 4,
               CSM_ACT_SIGMOID = 5, CSM_ACT_SOFTPLUS = 6 /* torch.nn.Softplus(): x > 20 ? x : log(1 + exp(x)) */,
               CSM_ACT_GELU = 7 /* torch.nn.GELU(): 0.5 x (1 + erf(x / sqrt 2)) */ };

typedef struct csm_tensor_desc {
    int64_t offset;      /* floats from the workspace base (ext < 0) or from ext pointer slot `ext` */
    int32_t ext;         /* -1: workspace; >=0: index into the ext pointer table of csm_run_program */
    int32_t n, h, w, c;  /* logical NHWC shape (c = channels of this view) */
    int32_t ld;          /* channel pitch in floats (>= c); NCHW ext tensors: ld is ignored */
} csm_tensor_desc;

/* Winograd contract (csm_op.flags & CSM_CONV_FLAG_WINOGRAD; restated in oracle/nets_oracle.c::orc_conv_wino, executed by
 * csrc/nets.hip::k_conv_wino): F(2x2, 3x3) with the transform matrices of Lavin & Gray (0, +-1, +-1/2 only).
 *   U[f = 4i + j][co][c] = fp32(G g G^T) evaluated in double, rows first: r0 = g0, r1 = ((g0 + g1) + g2) / 2, r2 = ((g0 - g1) + g2) / 2,
 *                          r3 = g2, then the same over the columns;
 *   V = B^T d B on the 4 x 4 input window d (origin (2 ty - 1, 2 tx - 1), zeros outside): t0 = d0 - d2, t1 = d1 + d2, t2 = d2 - d1,
 *                          t3 = d1 - d3 over the row index first, then over the column index -- one fp32 operation per value;
 *   M[f] = one fmaf chain per (f, tile, co) over the input channels from 0.0f, channel order of the direct contract;
 *   Y = A^T M A over j first: s0 = (m0 + m1) + m2, s1 = (m1 - m2) - m3, then over i; y = Y + bias; residual / activation as usual.
 * Packed device weights (cout tile of 64, 32-channel block cb, 8-channel step q): [co / 64][cb][q][f][h][co % 64][4] with channel
 * c = 32 cb + 8 q + 4 h + e at element e -- exactly the LDS image of one pipeline step (32 KB), moved by a linear LDS-DMA copy. */
/* Winograd F(4x4, 3x3) contract (csm_op.flags & CSM_CONV_FLAG_WINOGRAD4; restated in oracle/nets_oracle.c::orc_conv_wino4, executed by
 * csrc/wino4.hip::k_conv_wino4): Lavin & Gray's matrices for the points 0, +-1, +-2, inf -- 36 products per 4x4 output tile and channel
 * pair (F(2x2): 64, direct: 144).  Same layer class as F(2x2); which of the two a layer takes is the lowering's per-sample rule.
 *   U[f = 6i + j][co][c] = fp32(G g G^T) in double, rows first: r0 = g0 * 0.25, r1 = -((g0 + g1) + g2) / 6, r2 = -((g0 - g1) + g2) / 6,
 *                          r3 = ((g0 + 2 g1) + 4 g2) / 24, r4 = ((g0 - 2 g1) + 4 g2) / 24, r5 = g2 (IEEE double division), then the columns;
 *   V = B^T d B on the 6 x 6 window d (origin (4 ty - 1, 4 tx - 1), zeros outside), 1-D transform in fp32 with fmaf where written:
 *                          t0 = fmaf(4, d0, fmaf(-5, d2, d4)), t5 = fmaf(4, d1, fmaf(-5, d3, d5)),
 *                          a = fmaf(-4, d2, d4), b = fmaf(-4, d1, d3), t1 = a + b, t2 = a - b,
 *                          c = d4 - d2, e = d3 - d1, t3 = fmaf(2, e, c), t4 = fmaf(-2, e, c)   -- row index first, then column index;
 *   M[f] = one fmaf chain per (f, tile, co) over the input channels from 0.0f, channel order of the direct contract;
 *   Y = A^T M A over j first, then i, 1-D transform p = m1 + m2, q = m1 - m2, r = m3 + m4, t = m3 - m4,
 *                          s0 = (m0 + p) + r, s1 = fmaf(2, t, q), s2 = fmaf(4, r, p), s3 = fmaf(8, t, q) + m5;
 *   y = Y + bias; residual / activation as usual.  Output pixel (4 ty + a, 4 tx + b).
 * Packed device weights (cout tile of 64, step s = 4 input channels: 8-block s >> 1, half s & 1):
 *   [co / 64][s][wave = 6 nh + i][piece p][lh][li][jj][t] = U[6 i + 2 p + jj][64 (co / 64) + 32 nh + li][8 (s >> 1) + 4 lh + 2 (s & 1) + t]
 *   -- 36 KB per (cout tile, step); a wave's 3 KB are its three LDS-DMA pieces, lane 32 lh + li reads its four values as one 16-B word. */
#define CSM_CONV_FLAG_STEM 2
#define CSM_CONV_FLAG_WINOGRAD 4
#define CSM_CONV_FLAG_WINOGRAD4 8
/* Narrow grouped 3x3 convolutions on the vector pipe (csm_op.flags & CSM_CONV_FLAG_GROUPED; csrc/grouped.hip::k_conv_grouped): the DIRECT
 * arithmetic above, unchanged (same fmaf chain, same bits as the block-diagonal matrix-pipe form) -- the flag only says which weight image
 * the op carries.  3x3, stride 1, dilation 1, pad 1, cin_g == cout_g in {8, 16, 32}, channels % 32 == 0, ksplit 1; groups / cin_g /
 * cout_g are the REAL groups (not super-groups).  Packed device weights, one 32-float half unit per (tap, 8-channel block kb, half h):
 *   [group][octet][tap = 3 ky + kx][kb][h][chain position i][t] = w[group cin_g + 8 octet + 4 h + t][8 kb + 4 (i & 1) + (i >> 1)][ky][kx] */
#define CSM_CONV_FLAG_GROUPED 16

typedef struct csm_op {
    int32_t kind;
    int32_t in0, in1, out;   /* tensor ids; in1 = residual / second operand, -1 if none */
    int32_t kh, kw, stride, pad, dil;
    int32_t groups;          /* CONV: number of super-groups (1 = dense); each maps cin_g -> cout_g channels */
    int32_t cin_g, cout_g;
    int32_t act, res_mode;   /* res_mode: 0 none, 1 add before act, 2 add after act */
    int64_t w_off, b_off, aux_off;   /* float offsets into the weight buffer (aux = PReLU slopes); -1 = none */
    int32_t flags;           /* BILINEAR: bit 0 = align_corners.  CONV: bit 1 = stem (cin padded to 4): weights are packed with K =
                                (tap, channel), 8 taps x 4 channels per 32-wide chunk, for k_conv_stem; the chain is unchanged.
                                CONV bit 2 (CSM_CONV_FLAG_WINOGRAD): exact-fp32 Winograd F(2x2, 3x3) arithmetic (3x3, stride 1, dilation
                                1, pad 1, dense, cin % 32 == 0, cout % 64 == 0, ksplit 1): weights are the transformed panels
                                U = G g G^T packed for k_conv_wino; part of the NUMERICAL contract (set by the lowering from the layer's
                                per-sample shape, never by the tuner) -- see "Winograd contract" below.
                                CONV bit 3 (CSM_CONV_FLAG_WINOGRAD4): the same layer class in the F(4x4, 3x3) arithmetic (weights = its own
                                36-frequency panels packed for k_conv_wino4); bits 2 and 3 are exclusive.
                                CONV bit 4 (CSM_CONV_FLAG_GROUPED): narrow groups on the vector pipe -- direct arithmetic, its own
                                weight image, REAL groups in groups / cin_g / cout_g (see above) */
    int32_t ksplit;          /* CONV: K is cut into `ksplit` runs of (32-channel block, tap) chunks (block-major), run s = chunks
                                [s*T/ksplit, (s+1)*T/ksplit); each run is its own fmaf chain (run 0 starts at the bias,
                                the others at 0) and the runs are added in order ((p0+p1)+p2)...  1 = single chain */
    int32_t scratch;         /* tensor id of the [n,h,w,ksplit*cout] partial-sum buffer when ksplit > 1, else -1.  CONV with
                                CSM_CONV_FLAG_WINOGRAD4: optional (-1 = none) contiguous tensor of >= n * ceil(h/4) * ceil(w/4) * 24 * cout
                                floats; with it the library may run launches of few block tiles in a row-split execution form of
                                the same arithmetic (speed only: same bits) */
    int32_t tile;            /* CONV: 0 = built-in tile rule; low 6 bits k > 0 = tile configuration k-1, bit 6 (64) = split-K runs
                                walked serially by one block instead of ksplit blocks + reduce, bit 7 (128) = mixed-tile launch (the
                                configuration's tiles cover whole rounds of the grid, 64 x 64 tiles the remaining rows).  Speed only:
                                every configuration and every launch form produces the same bits; filled in by csm_conv_autotune */
} csm_op;

/* Measure every eligible tile configuration of every CONV op on the device (HIP events on `stream`, `reps` timed launches
 * each, minimum taken) and store the fastest in ops[i].tile.  Results are unchanged by construction (same fmaf chains);
 * the workspace contents are clobbered.  Returns the number of ops tuned (>= 0) or a negative status. */
int csm_conv_autotune(csm_op *ops, int n_ops, const csm_tensor_desc *tensors, int n_tensors, const float *weights,
                      float *workspace, void *const *ext, int n_ext, void *stream, int reps);

/* Persist / restore the tuned tiles (text file: layer signature -> tile).  load returns the number of entries merged (0 if the
 * file does not exist); csm_conv_autotune consults the table before measuring. */
int csm_conv_tile_cache_save(const char *path);
int csm_conv_tile_cache_load(const char *path);

/* Execute ops[0..n_ops) in order on `stream`.  `weights` and `workspace` are device pointers;
 * ext[i] are device pointers of external (caller-owned) tensors. */
int csm_run_program(const csm_op *ops, int n_ops, const csm_tensor_desc *tensors, int n_tensors,
                    const float *weights, float *workspace, void *const *ext, int n_ext, void *stream);
/* Tuning aid (not stable ABI): force the conv tile configuration index, -1 = built-in cost model. */
int csm_debug_force_conv_cfg(int cfg);
/* Test aid (not stable ABI): how ksplit > 1 layers are executed: -1 = tuned / built-in rule, 0 = parallel, 1 = serial. */
int csm_debug_force_splitk_serial(int mode);
/* Measurement aid (not stable ABI): bit 0 = the autotuner may choose mixed-tile launches (default on); bit 1 = N-grouped tile order
 * (layers whose weights exceed an XCD's L2) OFF (default on).  Speed only. */
int csm_debug_conv_tuner_options(int options);
/* Test aid (not stable ABI): bit 0 = CSM_OP_ATTENTION gathers the relative position bias from the table in global memory even when the
 * tile pair's window of the table fits the LDS budget (the path token grids wider than ~340 take).  Same results at the op's tolerance. */
int csm_debug_attention_options(int options);
/* Measurement aid: same execution, each op bracketed by HIP events on `stream`; synchronises and returns ms per op. */
int csm_run_program_profile(const csm_op *ops, int n_ops, const csm_tensor_desc *tensors, int n_tensors,
                            const float *weights, float *workspace, void *const *ext, int n_ext, void *stream,
                            float *op_ms);

/* ------------------------------------------------------------------------------------
 * Glue around the Inpaint / Refine nets, the depth-adjustment resize branch and AnimeInstances.resize (rounds 1-2 used torch ops here)
 * ---------------------------------------------------------------------------------- */

/* out2 = {x.mean(), x.std(unbiased=False)} over all n elements (pointcloud_inpainting.py:116-119, disparity_refinement.py:97-98);
 * double accumulation, two passes; scratch: csm_mean_std_scratch_bytes() bytes. */
size_t csm_mean_std_scratch_bytes(void);
int csm_mean_std(const float *x, int64_t n, float *out2, void *scratch, void *stream);
/* out = (x - mean) / (std + 1e-7) with {mean, std} read from device memory (:121-131 / :100-107) */
int csm_normalise_mean_std(const float *x, int64_t n, const float *mean_std_dev, float *out, void *stream);
/* out = x * (std + 1e-7) + mean, then mode 0: nothing, 1: clip(0, 1) (tenImage), 2: threshold(0.0, 0.0) (tenDisparity) */
int csm_denormalise_mean_std(const float *x, int64_t n, const float *mean_std_dev, int mode, float *out, void *stream);
/* torch.nn.functional.interpolate(mode='bilinear', align_corners=...) of `planes` [H,W] planes (NCHW with N*C = planes) to [h,w]:
 * depth_adjustment_animesseg's resize round trip (kenburns_effect.py:49-52, :89-90), disparity_estimation's input resize */
int csm_resize_bilinear_planes(const float *in, int planes, int H, int W, int h, int w, int align_corners, float *out, void *stream);
/* AnimeInstances.resize (animeinsseg/anime_instances.py:268-280): interpolate(masks.float(), (h, w), mode='area') > thr on boolean
 * masks [n,H,W] (1 B per pixel) -> [n,h,w] */
int csm_mask_area_resize_threshold(const uint8_t *masks, int n, int H, int W, int h, int w, float thr, uint8_t *out, void *stream);

/* ------------------------------------------------------------------------------------
 * ZoeDepth plumbing around the metric-bins head (CSM_OP_ATTRACTOR / CSM_OP_LOGBINOM) and the pluggable MiDaS core
 * replaces DepthModel.infer / _infer_with_pad_aug / infer_with_flip_aug (depth_modules/zoedepth/models/depth_model.py:57-129),
 * PrepForMidas + Resize (models/base_models/midas.py:49-187) and the tail of _depth_est_zoe (kenburns_effect.py:812-818)
 * ---------------------------------------------------------------------------------- */

/* img [B,3,H,W] (0..1) -> out [B,3,nh,nw] = Normalize(0.5, 0.5)(bilinear_align_corners(reflect_pad(flip ? hflip(img) : img, pad_w,
 * pad_h) -> (nh, nw))) in ONE pass (the padded image never exists); pad < size. */
int csm_zoe_pad_prep(const float *img, int B, int H, int W, int pad_h, int pad_w, int flip, int nh, int nw, float *out, void *stream);
/* d [B,1,h,w] (prediction at the core's resolution for the padded input) -> out [B,1,H,W]: bicubic (aten, A = -0.75,
 * align_corners = false) to the padded size (H + 2 pad_h, W + 2 pad_w), cropped to the original view, mirrored back when unflip;
 * mode 0: out = v; mode 1: out = (out + v) / 2 (second pass of infer_with_flip_aug). */
int csm_zoe_resize_crop(const float *d, int B, int h, int w, int pad_h, int pad_w, int H, int W, int unflip, int mode, float *out,
                        void *stream);
/* disparity = reciprocal(depth + 1e-5) * (focal * baseline); nan / +-inf -> 0   (kenburns_effect.py:816-817) */
int csm_zoe_depth_to_disparity(const float *depth, int64_t n, float focal_times_baseline, float *out, void *stream);

/* ------------------------------------------------------------------------------------
 * Instance-segmentation post-processing
 * ---------------------------------------------------------------------------------- */

/* np.packbits(mask != 0, bitorder='little'): boolean instance masks (1 B per pixel, as AnimeInstances holds them) -> 1 bit per pixel,
 * the wire format of the per-rank output gather.  out: ceil(n / 8) bytes. */
int csm_pack_mask_bits(const uint8_t *mask, int64_t n, uint8_t *out, void *stream);

/* RTMDet-Ins box decode between the head's raw maps and NMS -- mmdet RTMDetInsHead.predict_by_feat / _predict_by_feat_single /
 * _bbox_mask_post_process + filter_scores_and_topk [EXT mmdet 3.3.0; call site animeinsseg/__init__.py:450 model.test_step], with
 * FIXED shapes and no host sync: per level the `nms_pre` highest scores above score_thr (stable descending order), distance2bbox
 * against the resized-image bounds, rescale to the original image, min_bbox_size filter, then all levels merged in score order.
 * Slots without a valid candidate carry score -1 and sort last (they can never suppress a valid box in the greedy NMS).
 *   cls / reg: HOST arrays of n_levels DEVICE pointers to NHWC maps [nb, h_l, w_l, *] (cls: num_classes sigmoid scores per prior,
 *   reg: 4 relu'd distances in stride units); level_hw {h0, w0, h1, w1, ...}; strides; lds3 {cls, reg, kernel channel pitch} per level.
 *   outputs per image, in NMS order: scores [nb,K], boxes [nb,K,4] xyxy, src [nb,K] (global prior index, level-major), labels [nb,K],
 *   class_offsets [nb,K] = label * (max box coordinate + 1) (written when num_classes > 1; may be NULL otherwise).
 *   K <= csm_det_decode_slots(...) = sum_l min(nms_pre, h_l w_l num_classes) (<= 4096; nms_pre <= 1024; -1 = unsupported).
 *   scratch: csm_det_decode_scratch_bytes(nb, slots). */
int csm_det_decode_slots(const int *level_hw, int n_levels, int num_classes, int nms_pre);
size_t csm_det_decode_scratch_bytes(int nb, int slots);
int csm_det_decode(const float *const *cls, const float *const *reg, const int *level_hw, const int *strides, const int *lds3,
                   int n_levels, int nb, int num_classes, float score_thr, int nms_pre, float clamp_w, float clamp_h,
                   float scale_x, float scale_y, float min_bbox_size, int K, float *scores, float *boxes, int *src, int *labels,
                   float *class_offsets, void *scratch, void *stream);
/* After NMS: for the max_keep slots of keep [nb, max_keep] (indices into the K candidates; slots past the kept count are ignored by
 * the caller) gather scores, boxes, labels, the priors (x, y, stride, stride: MlvlPointGenerator offset 0) and the num_gen_params
 * dynamic-conv parameters from the per-level kernel maps. */
int csm_det_gather(const float *const *kern, const int *level_hw, const int *strides, const int *lds3, int n_levels, int nb, int K,
                   int max_keep, int num_gen_params, const int *keep, const float *scores, const float *boxes, const int *src,
                   const int *labels, float *kept_scores, float *kept_boxes, int *kept_labels, float *kept_priors,
                   float *kept_kernels, void *stream);

/* Greedy NMS, replaces mmcv.ops.batched_nms -> nms (C++/CUDA ext; call site: mmdet head, imported at
 * animeinsseg/models/rtmdet_inshead_custom.py:10).  boxes [n,4] xyxy sorted by descending score;
 * class_offsets [n] (label * (max_coord+1)) or NULL for class-agnostic / single class;
 * suppress iff inter > iou_thr * (Sa + Sb - inter).  keep[<=max_keep] = indices in score order.  n <= 4096. */
size_t csm_nms_scratch_bytes(int n);
int csm_nms(const float *boxes, const float *class_offsets, int n, float iou_thr, int max_keep, int *keep,
            int *n_keep, void *scratch, void *stream);

/* RTMDetInsSepBNHeadCustom._mask_predict_by_feat_single   animeinsseg/models/rtmdet_inshead_custom.py:253-303
 * mask_feat NHWC [h,w,num_prototypes] with channel pitch ld; kernels [n,169]; priors [n,4] = (x,y,stride,stride);
 * logits [n,h,w]. */
int csm_maskhead_logits(const float *mask_feat, int ld, int h, int w, int num_prototypes, int dyconv_channels,
                        const float *kernels, const float *priors, int n, int feat_stride, float *logits,
                        void *stream);

/* mmdet _bbox_mask_post_process tail (mirrored at animeinsseg/__init__.py:361-370), fused:
 * interpolate(scale_factor=up) -> interpolate(size=(rh,rw)) -> [..., :oh, :ow] -> sigmoid() > thr.
 * masks: uint8 [n,oh,ow] (0/1).  The slice cannot enlarge: the caller passes oh = min(rh, ori_h), ow = min(rw, ori_w)
 * (mmdet's ceil(S / scale) lands 1-2 px under the original size for ~20 % of image shapes). */
int csm_mask_resize_threshold(const float *logits, int n, int h, int w, int up, int rh, int rw, int oh, int ow,
                              float thr, uint8_t *masks, void *stream);

/* prepare_refine_batch   animeinsseg/__init__.py:37-55 (+ utils/io_utils.py:254-292 resize_pad):
 * img u8 HWC [H,W,3], masks u8 [n,Hm,Wm] -> batch fp32 NCHW [n,4,T,T]; (rh,rw) = keep-ratio size of the image inside T x T,
 * (rhm,rwm) = the same rule applied to the mask's own shape (the reference resize_pad()s every seg by its own size; detector
 * masks can be 1-2 px smaller than the image, see csm_mask_resize_threshold). */
int csm_refine_prepare_batch(const uint8_t *img_hwc, const uint8_t *masks, int n, int H, int W, int rh, int rw, int Hm, int Wm,
                             int rhm, int rwm, int T, float *batch, void *stream);

/* _postprocess_refine tail   animeinsseg/__init__.py:653-662:
 * sigmoid -> crop [:crop_h,:crop_w] -> bilinear(align_corners=True) to (oh,ow) -> > thr ; masks u8 [n,oh,ow]. */
int csm_refine_threshold(const float *logits, int n, int S_h, int S_w, int crop_h, int crop_w, int oh, int ow, float thr,
                         uint8_t *masks, void *stream);

/* Detector input: mmdet test pipeline Resize(keep_ratio) + Pad(pad_value) + DetDataPreprocessor normalise
 * (call sites animeinsseg/__init__.py:63-76, :212-215, :395-399).  img u8 HWC [H,W,3] (BGR) -> fp32 NCHW
 * [1,3,S_h,S_w]; (rh,rw) resized extent (host computes mmcv rescale_size); mean3/std3 are HOST pointers. */
int csm_det_preprocess(const uint8_t *img_hwc, int H, int W, int rh, int rw, int S_h, int S_w, const float *mean3,
                       const float *std3, float pad_value, float *out, void *stream);

/* ------------------------------------------------------------------------------------
 * uint8 image plumbing around the depth net and the frame loop (OpenCV semantics restated, [EXT])
 * ---------------------------------------------------------------------------------- */

/* utils/io_utils.py:254-274 scaledown_maxsize (the frame itself: kenburns_effect.py:917): cv2.resize(INTER_LINEAR) of a uint8
 * HWC image [H,W,C] (C <= 4) to [h,w,C]; the host applies the size rule. */
int csm_resize_u8_linear(const uint8_t *src_hwc, int H, int W, int C, int h, int w, uint8_t *dst_hwc, void *stream);
/* the same for float32 images / masks (resize_pad(seg, ...) of prepare_refine_batch, animeinsseg/__init__.py:47): cv2's float
 * INTER_LINEAR -- horizontal then vertical linear pass in fp32 */
int csm_resize_f32_linear(const float *src_hwc, int H, int W, int C, int h, int w, float *dst_hwc, void *stream);
/* kenburns_effect.py:563-571 + depth_modules/leres/leres/depthmap.py:16-38: BGR u8 HWC [H,W,3] -> cv2 INTER_LINEAR to
 * (h,w) -> /255 -> RGB -> (x-mean)/std (ImageNet) -> fp32 NCHW [1,3,h,w] */
int csm_leres_input(const uint8_t *img_hwc, int H, int W, int h, int w, float *out, void *stream);
/* depth_modules/leres/__init__.py:121-145: min-max -> uint16 -> convertScaleAbs(255/65535) -> bitwise_not.
 * min_max_dev: DEVICE pointer to {min, max} of depth. */
int csm_leres_quantize(const float *depth, int64_t n, const float *min_max_dev, uint8_t *out, void *stream);
/* Small device-side reductions of the per-frame depth glue (no host sync; min / max are order-free, so results equal torch's).
 * csm_minmax: out2 = {min, max} of x[0..n) (x 16-byte aligned); scratch512 = 512 device floats (block partials).
 * csm_fill_zero_min_positive: leres/__init__.py:143-145 `depth[depth == 0] = depth[depth > 0].min()` in place (unchanged when
 *   there is no zero or no positive value); scratch2 = 2 uint32.
 * csm_normalise_disparity: out = (x / minmax[1]) * scale   (kenburns_effect.py:928: disparity / disparity.max() * baseline);
 *   norm_max_out (1 device float, may be NULL) receives max(out).
 * csm_depth_range_stats: out6 (float64, device) = {min, max of the NORMALISED disparity (from the raw {min,max} and scale),
 *   cv2.minMaxLoc(depth[y0:y0+crop_h, x0:x0+crop_w]) = min, max, first row-major argmin, argmax}; scratch2 = 2 uint64. */
int csm_minmax(const float *x, int64_t n, float *out2, float *scratch512, void *stream);
int csm_fill_zero_min_positive(float *x, int64_t n, unsigned *scratch2, void *stream);
int csm_normalise_disparity(const float *x, int64_t n, const float *minmax_dev, float scale, float *out, float *norm_max_out,
                            void *stream);
int csm_depth_range_stats(const float *minmax_raw_dev, float scale, const float *depth, int H, int W, int y0, int x0, int crop_h,
                          int crop_w, unsigned long long *scratch2, double *out6, void *stream);

/* depth_adjustment_animesseg for ONE instance, in place (kenburns_effect.py:68-78, the non-median branch): pixels of `mask`
 * (bool/uint8 [H,W]) take the maximum of disp*mask over the rows from round(top + 0.97 (bottom - top)) down; untouched when the
 * instance plane is empty.  scratch: 2*H + 2 floats (device).  No host sync. */
int csm_depth_adjust_instance(float *disp, const uint8_t *mask, int H, int W, float *scratch, void *stream);
/* kenburns_effect.py:572-575: cv2.resize(u8 depth, (W,H), INTER_AREA) (enlarging) -> float32 */
int csm_resize_u8_to_f32(const uint8_t *src, int h, int w, int H, int W, float *out, void *stream);
/* uint8 [H,W,3] -> float32 [3,H,W] * (1/255): the image tensor of kenburns_effect.py:878-880 (`permute(2,0,1)[None].float() * (1.0 / 255.0)`) */
int csm_u8_hwc_to_f32_chw(const uint8_t *src_hwc, int H, int W, float *out_chw, void *stream);
/* the same line when the 32-aligned LeReS map is larger than the frame (k > 1): cv2.resize(..., INTER_LANCZOS4) -> float32 */
int csm_resize_u8_lanczos4_to_f32(const uint8_t *src, int h, int w, int H, int W, float *out, void *stream);
/* kenburns_effect.py:1069-1070: cv2.getRectSubPix(frame,(patch_w,patch_h),center) + cv2.resize(INTER_LINEAR) to (W,H) */
int csm_crop_resize_u8(const uint8_t *frame_hwc, int H, int W, int patch_h, int patch_w, float center_x, float center_y,
                       uint8_t *out_hwc, void *stream);

/* ------------------------------------------------------------------------------------
 * Bokeh depth-of-field (utils/effects.py:12-181; kenburns_effect.py:1042-1067)
 * ---------------------------------------------------------------------------------- */
/* kernel_bokeh   utils/effects.py:16-74 : one depth-weighted `nsamples`-tap line blur along (dx,dy); img/out fp32 HWC
 * [H,W,3] (the reference kernel indexes raw HWC memory), depth fp32 [H,W] (already x 0.0005). */
int csm_bokeh_pass(const float *img_hwc, const float *depth, float *out_hwc, int H, int W, int nsamples, float dx, float dy,
                   void *stream);
/* (img/255)^lightness  utils/effects.py:155-156 ; n = H*W*3 */
int csm_bokeh_highlight(const uint8_t *img_hwc, float *out_hwc, int64_t n, float lightness, void *stream);
/* the third pass of bokeh_blur fused with its finish: out = uint8(((diag + pass(diag)) / 2)^(1/lightness) * 255), the same
 * operations in the same order as csm_bokeh_pass followed by csm_bokeh_finish (no float plane is written) */
int csm_bokeh_pass_finish(const float *diag_hwc, const float *depth, uint8_t *out_hwc_u8, int H, int W, int nsamples, float dx, float dy,
                          float lightness, void *stream);
/* ((diag+rhom)/2)^(1/lightness)*255 -> uint8  utils/effects.py:172,179-180 */
int csm_bokeh_finish(const float *diag_hwc, const float *rhom_hwc, uint8_t *out_hwc, int64_t n, float lightness, void *stream);
/* depth map of bokeh_blur  utils/effects.py:146-153,162-163: out = (1 - ((dmax - |d - focal|) - mn) / mx2) * 0.0005 ;
 * dmax = max(d), mn = min(dmax - |d-focal|), mx2 = max(that - mn) are scalar reductions supplied by the caller. */
int csm_bokeh_depth(const uint8_t *depth_u8, float *out, int64_t n, float dmax, float focal_plane, float mn, float mx2, void *stream);
/* The same map for the reference's own call forms (utils/effects.py:143-153 defaults: float depth, depth_factor = 2, focal_plane =
 * None): depth is float32 (is_u8 = 0, 16-B aligned) or uint8 (is_u8 = 1); has_focal selects `max(d) - |d - focal_plane|`;
 * depth_factor != 1 applies np.power (2 -> square, else powf).  tmp [n] floats, mm4 [4] floats and scratch512 [512] floats are
 * device scratch; all reductions stay on the device (no host sync). */
int csm_bokeh_depth_general(const void *depth, int is_u8, int64_t n, int has_focal, float focal_plane, float depth_factor,
                            float *tmp, float *mm4, float *scratch512, float *out, void *stream);
/* Focal plane of the depth of field (kenburns_effect.py:1045-1056): out[k] = np.median(values[masks[k] != 0]) for each of the n_inst
 * boolean masks [n_inst, n] (nan when the mask is empty), out[n_inst] = the largest of them, -1 if every mask is empty.  values
 * uint8 [n]; hist: n_inst * 256 uint32 of device scratch (zeroed by the call). */
int csm_masked_u8_median_max(const uint8_t *values, const uint8_t *masks, int n_inst, int64_t n, unsigned *hist, float *out, void *stream);
/* colorize(value, cmap='gray_r')[...,0]  depth_modules/zoedepth/utils/misc.py:97-135 (vmin/vmax = 2nd/85th percentile) */
int csm_colorize_gray_r(const float *value, uint8_t *out, int64_t n, float vmin, float vmax, void *stream);

/* The same three steps without host round trips (frametail.hip) -- the per-frame tail of kenburns_effect.py:1042-1067:
 * csm_percentile_pair: out2 (DEVICE) = {np.percentile(value, q_lo), np.percentile(value, q_hi)} (method 'linear'), exact, by a
 *   3-pass (16 + 8 + 8 bit) radix select instead of a sort, three launches; scratch = csm_percentile_scratch_bytes() device bytes, ZEROED
 *   ONCE by the caller (every call leaves the counting tables it needs next time cleared) and used by one stream at a time.  If a call
 *   finds the tables NOT in that state (e.g. after an aborted launch) its counts do not add up to n: out2 becomes NaN, for that call and
 *   every later one, until the caller zeroes the scratch again -- never a plausible wrong percentile.
 * csm_colorize_gray_r_dev: csm_colorize_gray_r with vmin / vmax read from device memory and the matplotlib byte LUT (256 HOST
 *   bytes, index -> grey level) applied in the kernel.
 * csm_bokeh_depth_auto: csm_bokeh_depth with dmax / mn / mx2 derived on the device from the histogram of depth_u8; scratch =
 *   csm_bokeh_depth_scratch_bytes() device bytes. */
size_t csm_percentile_scratch_bytes(void);
int csm_percentile_pair(const float *value, int64_t n, double q_lo, double q_hi, float *out2, void *scratch, void *stream);
int csm_colorize_gray_r_dev(const float *value, uint8_t *out, int64_t n, const float *vmin_vmax_dev, const uint8_t *lut256_host,
                            void *stream);
size_t csm_bokeh_depth_scratch_bytes(void);
int csm_bokeh_depth_auto(const uint8_t *depth_u8, float *out, int64_t n, float focal_plane, void *scratch, void *stream);

/* One output frame of KenBurnsPipeline.process_kenburns (kenburns_effect.py:1027-1072) in ONE call: csm_warp_frame_tiled
 * [-> csm_percentile_pair(2, 85) -> csm_colorize_gray_r_dev -> csm_bokeh_depth_auto -> csm_bokeh_highlight -> 2 x csm_bokeh_pass ->
 * csm_bokeh_pass_finish, when dof != 0 (depth_field, depth_factor 1)] -> csm_crop_resize_u8 into out_hwc.  Same kernels, order and
 * arguments as the separate calls (bit-identical frames); it exists because the host side of a dozen calls per frame was the limit
 * of the frame loop.  warp_scratch as for csm_warp_frame_tiled; render [4,H,W] is required when dof; frame_u8 [H,W,3] receives the
 * un-cropped warp; tail_scratch: csm_kenburns_frame_scratch_bytes(H, W) bytes, zeroed ONCE by the caller (tickets and counting tables: every call leaves them cleared). */
size_t csm_kenburns_frame_scratch_bytes(int H, int W);
int csm_kenburns_frame(const float *pts, const float *rgb, const float *depth, int64_t N, int H, int W, double focal, double baseline,
                       float sx, float sy, float sz, void *warp_scratch, float *render, uint8_t *frame_u8, int dof, float focal_plane,
                       int num_samples, float lightness, const uint8_t *gray_r_lut256_host, void *tail_scratch, int patch_h, int patch_w,
                       float center_x, float center_y, uint8_t *out_hwc, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* CSM355_H */
