"""ZoeDepth metric-bins head (everything of ZoeDepth.forward after `self.core(...)`) -> layer program.

Reference: depth_modules/zoedepth/models/zoedepth/zoedepth_v1.py:124-202 with the layers of models/layers/{localbins_layers,
attractor,dist_layers}.py, configured as the shipped ZoeD_M12_N model (config_zoedepth.json): 64 bins, softplus (unbounded) bin
centres, attractors [16, 8, 4, 1] of type "inv" / kind "mean" with alpha 1000, gamma 2, log-binomial output with temperature in
[0.0212, 50].  All 1x1 convolutions run on the MFMA engine (softplus / GELU fused in the epilogue), the attractor update and the
log-binomial expectation are the two dedicated ops CSM_OP_ATTRACTOR / CSM_OP_LOGBINOM.

NOT included: `self.core` -- the MiDaS DPT-BEiT-L backbone comes from torch.hub (intel-isl/MiDaS) + timm, neither vendored nor
installed.  The program therefore takes the core's six feature maps and its relative depth as inputs; with a real core
in front it IS the rest of ZoeDepth.forward.
"""
import numpy as np

from ..program import Program
from ..weights import conv_plain


def log_binom_table(n_bins):
    """dist_layers.py:29-34 log_binom(K - 1, k) for k = 0..K-1, evaluated in float32 like the registered buffers"""
    f = np.float32
    eps = f(1e-7)
    n = f(n_bins - 1) + eps
    k = np.arange(n_bins).astype(np.float32) + eps
    return (n * np.log(n) - k * np.log(k) - (n - k) * np.log(n - k + eps)).astype(np.float32)


def build_zoe_head(ws, n, H, W, feat_sizes, n_bins=64, bin_embedding_dim=128, n_attractors=(16, 8, 4, 1), attractor_alpha=1000.0,
                   attractor_kind='mean', attractor_type='inv', min_temp=0.0212, max_temp=50.0, btlnck_features=256,
                   num_out_features=(256, 256, 256, 256), n_midas_out=32):
    """ext tensors (NCHW): [0] rel_depth [n,1,H,W], [1] out_conv activation [n,32,H,W], [2] bottleneck [n,256,h0,w0],
    [3..6] decoder blocks r4..r1 [n,256,h_i,w_i] with feat_sizes = [(h0,w0), (h1,w1), ...(h4,w4)], [7] metric depth out [n,1,H,W]"""
    assert len(feat_sizes) == 1 + len(num_out_features)
    p = Program("zoe_head")
    rel_ext = p.ext_nchw(n, 1, H, W)
    oc_ext = p.ext_nchw(n, n_midas_out, H, W)
    bt_ext = p.ext_nchw(n, btlnck_features, *feat_sizes[0])
    blk_ext = [p.ext_nchw(n, c, *feat_sizes[i + 1]) for i, c in enumerate(num_out_features)]
    out_ext = p.ext_nchw(n, 1, H, W)

    def mlp(name, x, hidden, cout, act_out=None):
        w0, b0 = conv_plain(ws, name + '.0', hidden, x.c, 1)
        t = p.conv(x, w0, b0, act='relu')
        w1, b1 = conv_plain(ws, name + '.2', cout, hidden, 1)
        return p.conv(t, w1, b1, act=act_out)

    btl = p.to_nhwc(bt_ext)
    wc, bc = conv_plain(ws, 'conv2', btlnck_features, btlnck_features, 1)
    x = p.conv(btl, wc, bc)                                                        # x_d0
    b_prev = mlp('seed_bin_regressor._net', x, 256, n_bins, act_out='softplus')    # SeedBinRegressorUnnormed
    prev_emb = mlp('seed_projector._net', x, 128, bin_embedding_dim)
    emb = prev_emb
    for i, (xe, na) in enumerate(zip(blk_ext, n_attractors)):
        xb = p.to_nhwc(xe)
        emb = mlp('projectors.%d._net' % i, xb, 128, bin_embedding_dim)
        # AttractorLayerUnnormed.forward: x + interpolate(prev_b_embedding) -> _net -> A ; b_prev interpolated ; b + mean_i inv(A_i - b)
        pe = p.bilinear(prev_emb, (xb.h, xb.w), align_corners=True) if (prev_emb.h, prev_emb.w) != (xb.h, xb.w) else prev_emb
        xa = p.add(emb, pe)
        A = mlp('attractors.%d._net' % i, xa, 128, na, act_out='softplus')
        bu = p.bilinear(b_prev, (xb.h, xb.w), align_corners=True) if (b_prev.h, b_prev.w) != (xb.h, xb.w) else b_prev
        # reference quirk kept: both attractor layers call `dist(A - b)` WITHOUT alpha / gamma (attractor.py:105-106, :188-189), so
        # the jit functions' defaults alpha = 300, gamma = 2 apply whatever `attractor_alpha` the config carries (1000 in the shipped one)
        b_prev = p.attractor(A, bu, 300.0, attractor_type, attractor_kind)
        prev_emb = emb
    # last = cat([outconv_activation, rel_cond]) ; cat with the up-sampled embedding -> ConditionalLogBinomial.mlp.
    # Buffer channel order here: [out_conv | embedding | rel_depth + zero pad]; the first conv's input columns are permuted to match.
    cin = n_midas_out + 1 + bin_embedding_dim
    L = p.buffer(n, H, W, n_midas_out + bin_embedding_dim + 4)
    p.to_nhwc_into(oc_ext, L.slice(0, n_midas_out))
    p.bilinear(emb, (H, W), align_corners=True, out=L.slice(n_midas_out, n_midas_out + bin_embedding_dim))
    p.to_nhwc_into(rel_ext, L.slice(n_midas_out + bin_embedding_dim, n_midas_out + bin_embedding_dim + 4))
    bottleneck = cin // 2
    w0, b0 = conv_plain(ws, 'conditional_log_binomial.mlp.0', bottleneck, cin, 1)
    w0p = np.zeros((bottleneck, L.c, 1, 1), np.float32)
    w0p[:, :n_midas_out] = w0[:, :n_midas_out]
    w0p[:, n_midas_out:n_midas_out + bin_embedding_dim] = w0[:, n_midas_out + 1:]
    w0p[:, n_midas_out + bin_embedding_dim] = w0[:, n_midas_out]
    t = p.conv(L, w0p, b0, act='gelu')
    w1, b1 = conv_plain(ws, 'conditional_log_binomial.mlp.2', 4, bottleneck, 1)
    pt = p.conv(t, w1, b1, act='softplus')
    cen = p.bilinear(b_prev, (H, W), align_corners=True)
    d = p.logbinom(pt, cen, 1e-4, min_temp, max_temp, log_binom_table(n_bins))
    p.to_nchw(d, out_ext)
    p.plan()
    return p
