"""MiDaS DPT-BEiT-L-384 -- the core of ZoeDepth (`self.core.core`, depth_modules/zoedepth/models/base_models/midas.py:341:
torch.hub.load("intel-isl/MiDaS", "DPT_BEiT_L_384")) -> layer program.

The network is NOT under /root/reference (torch.hub + timm): it is lowered here from the PUBLISHED definitions [EXT, unpinned
against MiDaS itself], with the parameter names of the published checkpoint `dpt_beit_large_384.pt`, i.e.

  * timm 0.6.x `beit_large_patch16_384` (models/beit.py: PatchEmbed 16 x 16, class token, no absolute position embedding, 24 x Block =
    x + gamma_1 * Attention(LayerNorm(x)); x + gamma_2 * Mlp(LayerNorm(x)), LayerNorm eps 1e-6, Attention with q_bias / v_bias (no k bias),
    q scaled by head_dim^-0.5 and a per-block relative-position-bias table of (2*24-1)^2 + 3 rows);
  * MiDaS 3.1 midas/backbones/beit.py: the table is bilinearly re-sampled to the (2 gh - 1) x (2 gw - 1) window of the actual input in
    every forward (`_get_rel_pos_bias`, including its reshape(1, old_width, old_height, -1) axis order) and indexed with timm's
    gen_relative_position_index of that window; hooks on blocks 5 / 11 / 17 / 23;
  * midas/backbones/utils.py make_backbone_default: readout "project" (ProjectReadout: Linear(2C -> C) + GELU on (token | class token)),
    1x1 convolutions to 256 / 512 / 1024 / 1024 channels, ConvTranspose2d(k = s = 4) / (k = s = 2) / identity / Conv2d(3x3, stride 2);
  * midas/dpt_depth.py + blocks.py: scratch.layerN_rn (3x3, no bias) -> FeatureFusionBlock_custom x 4 (ResidualConvUnit_custom with
    ReLU(False), bilinear align_corners=True, 1x1 out_conv) -> head Conv3x3(256 -> 128), x2 bilinear, Conv3x3(128 -> 32) + ReLU,
    Conv1x1(32 -> 1) + ReLU.

Lowering choices (exact or tolerance-level, never silent): every nn.Linear is a 1x1 convolution on the fp32 MFMA engine; the attention
scale 1/8 (a power of two) is folded into the q rows of the qkv weight and bias (bit-identical); the layer scales gamma_1 / gamma_2 are
folded into proj / fc2 (one rounding of w * gamma instead of one of y * gamma: tolerance-level) so that the residual add is the
convolution's epilogue; the relative-position table is re-sampled on the host when the program is built (it depends on the input size
only) and the N x N bias is never materialised (CSM_OP_ATTENTION indexes the table arithmetically).

What ZoeDepth's MidasCore hooks deliver (midas.py:189 layer_names): (out_conv, l4_rn, r4, r3, r2, r1) + the relative depth.
"""
import numpy as np

from ..program import Program


class DPTBeitConfig:
    def __init__(self, embed=1024, depth=24, heads=16, mlp_ratio=4, patch=16, base_grid=(24, 24), hooks=(5, 11, 17, 23), features=256,
                 neck=(256, 512, 1024, 1024), readout='project', ln_eps=1e-6, head_features_2=32):
        self.embed, self.depth, self.heads, self.mlp_ratio, self.patch = embed, depth, heads, mlp_ratio, patch
        self.base_grid, self.hooks, self.features, self.neck = tuple(base_grid), tuple(hooks), features, tuple(neck)
        self.readout, self.ln_eps, self.head_features_2 = readout, ln_eps, head_features_2
        assert readout in ('project', 'ignore') and embed % heads == 0 and len(hooks) == 4 and len(neck) == 4


def _src_index(dst, in_size, out_size):
    """aten area_pixel_compute_source_index for bilinear, align_corners=False -> (i0, i1, l0, l1) in float32 arithmetic"""
    scale = np.float32(in_size) / np.float32(out_size)
    src = np.maximum(scale * (dst.astype(np.float32) + np.float32(0.5)) - np.float32(0.5), np.float32(0.0)).astype(np.float32)
    i0 = np.minimum(src.astype(np.int64), in_size - 1)
    i1 = np.minimum(i0 + 1, in_size - 1)
    l1 = (src - i0.astype(np.float32)).astype(np.float32)
    return i0, i1, (np.float32(1.0) - l1).astype(np.float32), l1


def resample_rel_table(table, base_grid, grid):
    """MiDaS `_get_rel_pos_bias` up to the index gather: table [(2b0-1)(2b1-1)+3, heads] of the pre-training window `base_grid` ->
    [(2gh-1)(2gw-1)+3, heads] for the window `grid`.  F.interpolate(mode="bilinear") (align_corners=False) restated in numpy float32;
    MiDaS views the old table as [1, heads, old_WIDTH, old_HEIGHT] (its reshape order) and asks for size (new_height, new_width)."""
    table = np.asarray(table, np.float32)
    heads = table.shape[1]
    old_h, old_w = 2 * base_grid[0] - 1, 2 * base_grid[1] - 1
    new_h, new_w = 2 * grid[0] - 1, 2 * grid[1] - 1
    assert table.shape[0] == old_h * old_w + 3
    sub = table[:old_h * old_w].reshape(old_w, old_h, heads)             # axes as MiDaS's reshape names them
    if (new_h, new_w) != (old_w, old_h):
        r0, r1, a0, a1 = _src_index(np.arange(new_h), old_w, new_h)
        c0, c1, b0, b1 = _src_index(np.arange(new_w), old_h, new_w)
        top = sub[r0][:, c0] * b0[None, :, None] + sub[r0][:, c1] * b1[None, :, None]
        bot = sub[r1][:, c0] * b0[None, :, None] + sub[r1][:, c1] * b1[None, :, None]
        sub = (top * a0[:, None, None] + bot * a1[:, None, None]).astype(np.float32)
    return np.concatenate([sub.reshape(new_h * new_w, heads), table[old_h * old_w:]]).astype(np.float32)


def build_dpt_beit(ws, n, H, W, cfg=None):
    """ext tensors (NCHW): [0] the prepared input [n,3,H,W] (PrepForMidas output, H and W multiples of 32), then the outputs
    [1] relative depth [n,1,H,W], [2] out_conv activation [n,32,H,W], [3] l4_rn [n,F,H/32,W/32], [4..7] r4, r3, r2, r1 [n,F,H/16 .. H/2]"""
    cfg = cfg or DPTBeitConfig()
    E, heads, P = cfg.embed, cfg.heads, cfg.patch
    assert H % (2 * P) == 0 and W % (2 * P) == 0, "input must be a multiple of 32 (PrepForMidas ensures it)"
    gh, gw = H // P, W // P
    d = E // heads
    F = cfg.features
    p = Program("dpt_beit")
    x_ext = p.ext_nchw(n, 3, H, W)
    rel_ext = p.ext_nchw(n, 1, H, W)
    oc_ext = p.ext_nchw(n, cfg.head_features_2, H, W)
    l4_ext = p.ext_nchw(n, F, gh // 2, gw // 2)
    r_ext = [p.ext_nchw(n, F, gh << k, gw << k) for k in range(4)]        # r4 (H/16), r3, r2, r1 (H/2)

    def lin(name, cout, cin, bias=True):
        w = ws.get(name + '.weight', (cout, cin), 'lin_w')
        return w, (ws.get(name + '.bias', (cout,), 'conv_b') if bias else None)

    def conv(name, cout, cin, k, bias=True):
        w = ws.get(name + '.weight', (cout, cin, k, k), 'conv_w')
        return w, (ws.get(name + '.bias', (cout,), 'conv_b') if bias else None)

    # ---- BEiT encoder (timm beit.py forward_features as patched by MiDaS beit.py) ---------------------------------------------------
    x = p.to_nhwc(x_ext)
    pw, pb = conv('pretrained.model.patch_embed.proj', E, 3, P)
    t = p.tokens_assemble(p.conv(x, pw, pb, stride=P), ws.get('pretrained.model.cls_token', (1, 1, E), 'token').reshape(-1))
    scale = np.float32(d) ** np.float32(-0.5)
    T0 = (2 * cfg.base_grid[0] - 1) * (2 * cfg.base_grid[1] - 1) + 3
    taken = {}
    for i in range(cfg.depth):
        pre = 'pretrained.model.blocks.%d.' % i
        y = p.layernorm(t, ws.get(pre + 'norm1.weight', (E,), 'bn_gamma'), ws.get(pre + 'norm1.bias', (E,), 'bn_beta'), cfg.ln_eps)
        qkv_w = ws.get(pre + 'attn.qkv.weight', (3 * E, E), 'lin_w').copy()
        qkv_b = np.concatenate([ws.get(pre + 'attn.q_bias', (E,), 'conv_b'), np.zeros(E, np.float32), ws.get(pre + 'attn.v_bias', (E,), 'conv_b')])
        qkv_w[:E] *= scale; qkv_b[:E] *= scale                              # q = q * self.scale (timm Attention.forward)
        table = resample_rel_table(ws.get(pre + 'attn.relative_position_bias_table', (T0, heads), 'rel_bias'), cfg.base_grid, (gh, gw))
        a = p.attention(p.linear(y, qkv_w, qkv_b), heads, (gh, gw), table)
        g1 = ws.get(pre + 'gamma_1', (E,), 'layer_scale')
        ow, ob = lin(pre + 'attn.proj', E, E)
        t = p.linear(a, ow * g1[:, None], ob * g1, res=t)                   # x + gamma_1 * proj(attn)
        y = p.layernorm(t, ws.get(pre + 'norm2.weight', (E,), 'bn_gamma'), ws.get(pre + 'norm2.bias', (E,), 'bn_beta'), cfg.ln_eps)
        w1, b1 = lin(pre + 'mlp.fc1', cfg.mlp_ratio * E, E)
        w2, b2 = lin(pre + 'mlp.fc2', E, cfg.mlp_ratio * E)
        g2 = ws.get(pre + 'gamma_2', (E,), 'layer_scale')
        t = p.linear(p.linear(y, w1, b1, act='gelu'), w2 * g2[:, None], b2 * g2, res=t)
        if i in cfg.hooks:
            taken[i] = t
    hooked = [taken[h] for h in cfg.hooks]                                 # activations "1".."4" = blocks hooks[0]..hooks[3] (make_backbone_default)

    # ---- reassemble (act_postprocess1..4) + scratch.layerN_rn ---------------------------------------------------------------------
    layers_rn = []
    for k, tk in enumerate(hooked):
        pre = 'pretrained.act_postprocess%d.' % (k + 1)
        if cfg.readout == 'project':
            rw, rb = lin(pre + '0.project.0', E, 2 * E)
            y = p.linear(p.tokens_readout(tk, (gh, gw), project=True), rw, rb, act='gelu')
        else:
            y = p.tokens_readout(tk, (gh, gw), project=False)
        cw, cb = conv(pre + '3', cfg.neck[k], E, 1)
        y = p.conv(y, cw, cb)
        if k == 0:
            y = p.conv_transpose_nonoverlap(y, ws.get(pre + '4.weight', (cfg.neck[0], cfg.neck[0], 4, 4), 'conv_w'),
                                            ws.get(pre + '4.bias', (cfg.neck[0],), 'conv_b'), 4)
        elif k == 1:
            y = p.conv_transpose_nonoverlap(y, ws.get(pre + '4.weight', (cfg.neck[1], cfg.neck[1], 2, 2), 'conv_w'),
                                            ws.get(pre + '4.bias', (cfg.neck[1],), 'conv_b'), 2)
        elif k == 3:
            dw, db = conv(pre + '4', cfg.neck[3], cfg.neck[3], 3)
            y = p.conv(y, dw, db, stride=2, pad=1)
        rw_, _ = conv('scratch.layer%d_rn' % (k + 1), F, cfg.neck[k], 3, bias=False)
        layers_rn.append(p.conv(y, rw_, None, pad=1))
    l1, l2, l3, l4 = layers_rn

    # ---- fusion blocks (blocks.py FeatureFusionBlock_custom / ResidualConvUnit_custom, activation ReLU(False), no batch norm) ------
    def rcu(name, xin):
        w1, b1 = conv(name + '.conv1', F, F, 3)
        w2, b2 = conv(name + '.conv2', F, F, 3)
        return p.conv(p.conv(p.act(xin, 'relu'), w1, b1, pad=1, act='relu'), w2, b2, pad=1, res=xin, res_mode=1)

    def fusion(name, path, skip, size):
        out = path
        if skip is not None:
            out = p.add(out, rcu(name + '.resConfUnit1', skip))
        out = rcu(name + '.resConfUnit2', out)
        out = p.bilinear(out, size, align_corners=True)
        ow, ob = conv(name + '.out_conv', F, F, 1)
        return p.conv(out, ow, ob)

    path4 = fusion('scratch.refinenet4', l4, None, (l3.h, l3.w))
    path3 = fusion('scratch.refinenet3', path4, l3, (l2.h, l2.w))
    path2 = fusion('scratch.refinenet2', path3, l2, (l1.h, l1.w))
    path1 = fusion('scratch.refinenet1', path2, l1, (2 * l1.h, 2 * l1.w))

    # ---- head (dpt_depth.py DPTDepthModel, non_negative=True) -------------------------------------------------------------------------
    w0, b0 = conv('scratch.output_conv.0', F // 2, F, 3)
    y = p.bilinear(p.conv(path1, w0, b0, pad=1), (H, W), align_corners=True)
    w2, b2 = conv('scratch.output_conv.2', cfg.head_features_2, F // 2, 3)
    oc = p.conv(y, w2, b2, pad=1, act='relu')                               # MidasCore's "out_conv" hook: output_conv[3] (the ReLU)
    w4, b4 = conv('scratch.output_conv.4', 1, cfg.head_features_2, 1)
    rel = p.conv(oc, w4, b4, act='relu')
    for src, dst in ((rel, rel_ext), (oc, oc_ext), (l4, l4_ext), (path4, r_ext[0]), (path3, r_ext[1]), (path2, r_ext[2]), (path1, r_ext[3])):
        p.to_nchw(src, dst)
    p.plan()
    return p
