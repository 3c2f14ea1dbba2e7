"""RTMDet-Ins-L (CSPNeXt-L backbone + CSPNeXtPAFPN + RTMDetInsSepBNHead) -> layer program.

The reference builds this model through the mmdet registry from the cfg string stored in the checkpoint
(animeinsseg/__init__.py:196-210); mmdet 3.3.0 / mmcv 2.1.0 are NOT under /root/reference, so
this file restates the mmdet 3.3.0 modules from their published definitions -- PARITY UNPINNED:
  mmdet/models/backbones/cspnext.py (CSPNeXt, arch P5), backbones/csp_darknet.py (SPPBottleneck),
  layers/csp_layer.py (CSPLayer, CSPNeXtBlock, ChannelAttention), necks/cspnext_pafpn.py,
  dense_heads/rtmdet_ins_head.py (RTMDetInsSepBNHead, MaskFeatModule).
The head consumed by the reference's own subclass is pinned where it is vendored:
animeinsseg/models/rtmdet_inshead_custom.py:253-303 (dynamic-conv mask head, see maskhead.hip).

The architecture is config-driven (RTMDetConfig); defaults = rtmdet-ins_l with num_classes=1.
Parameter names follow mmdet's state_dict so a real rtmdetl_e60.ckpt imports unchanged.
"""
from dataclasses import dataclass, field
from typing import Tuple

import numpy as np

from ..program import Program
from ..weights import conv_bn, conv_plain


@dataclass
class RTMDetConfig:
    deepen_factor: float = 1.0
    widen_factor: float = 1.0
    expand_ratio: float = 0.5
    num_classes: int = 1
    feat_channels: int = 256
    stacked_convs: int = 2
    num_prototypes: int = 8
    dyconv_channels: int = 8
    num_dyconvs: int = 3
    strides: Tuple[int, ...] = (8, 16, 32)
    share_conv: bool = True
    # BatchNorm eps per sub-module: the rtmdet configs pass norm_cfg=dict(type='SyncBN') to backbone / neck / head, i.e. torch's
    # 1e-5; a cfg that leaves norm_cfg out gets the class defaults of CSPNeXt / CSPNeXtPAFPN (eps=1e-3) -- config_from_ckpt_cfg
    # reads them from the checkpoint's cfg text
    bn_eps: float = 1e-5
    bn_eps_backbone: float = None
    bn_eps_neck: float = None
    bn_eps_head: float = None
    # DetDataPreprocessor (BGR kept: bgr_to_rgb=False)
    mean: Tuple[float, ...] = (103.53, 116.28, 123.675)
    std: Tuple[float, ...] = (57.375, 57.12, 58.395)
    pad_value: int = 114
    # test_cfg
    nms_pre: int = 1000
    score_thr: float = 0.05
    nms_iou: float = 0.6
    max_per_img: int = 100
    mask_thr_binary: float = 0.5
    min_bbox_size: float = 0.0
    arch: tuple = field(default_factory=lambda: ((64, 128, 3, True, False), (128, 256, 6, True, False),
                                                 (256, 512, 6, True, False), (512, 1024, 3, False, True)))

    @property
    def num_gen_params(self):
        d, p = self.dyconv_channels, self.num_prototypes
        return (p + 2) * d + d * d * (self.num_dyconvs - 2) + d + d * (self.num_dyconvs - 1) + 1


class _B:
    def __init__(self, p, ws, cfg):
        self.p, self.ws, self.cfg = p, ws, cfg

    def eps(self, name):
        c = self.cfg
        v = c.bn_eps_backbone if name.startswith('backbone.') else (c.bn_eps_neck if name.startswith('neck.') else c.bn_eps_head)
        return c.bn_eps if v is None else v

    def cm_params(self, name, cin, cout, k, wname=None, bnname=None):
        """folded (w, b) of an mmcv ConvModule conv(bias=False) + BN; wname / bnname: parameter-name aliases (shared modules)"""
        return conv_bn(_Alias(self.ws, name, wname, bnname), name + '.conv', name + '.bn', cout, cin, k, eps=self.eps(name))

    def cm(self, name, x, cout, k, stride=1, act='silu', out=None, res=None, res_mode=0, wname=None, bnname=None, params=None, cin=None):
        """mmcv ConvModule: conv(bias=False) + BN + act.  cin: the checkpoint's input-channel count when the NHWC tensor carries zero
        padding channels (the 3-channel image sits in a 4-channel buffer; Program.conv pads the weights to match)"""
        w, b = params if params is not None else self.cm_params(name, x.c if cin is None else cin, cout, k, wname, bnname)
        return self.p.conv(x, w, b, stride=stride, pad=k // 2, act=act, out=out, res=res, res_mode=res_mode)

    def dwcm(self, name, x, k, act='silu'):
        from ..program import fold_bn
        ws, c = self.ws, x.c
        w = ws.get(name + '.conv.weight', (c, 1, k, k), 'conv_w')
        g = ws.get(name + '.bn.weight', (c,), 'bn_gamma'); be = ws.get(name + '.bn.bias', (c,), 'bn_beta')
        m = ws.get(name + '.bn.running_mean', (c,), 'bn_mean'); v = ws.get(name + '.bn.running_var', (c,), 'bn_var')
        wf, bf = fold_bn(w, None, g, be, m, v, self.eps(name))
        return self.p.dwconv(x, wf, bf, pad=k // 2, act=act)

    def cspnext_block(self, name, x, add_identity, out=None):
        t = self.cm(name + '.conv1', x, x.c, 3)
        t = self.dwcm(name + '.conv2.depthwise_conv', t, 5)
        return self.cm(name + '.conv2.pointwise_conv', t, x.c, 1, out=out, res=x if add_identity else None, res_mode=2)

    def csp_layer(self, name, x, cout, num_blocks, add_identity, channel_attention, out=None):
        p, mid = self.p, int(cout * self.cfg.expand_ratio)
        F = p.buffer(x.n, x.h, x.w, 2 * mid)                    # cat((x_main, x_short), 1)
        self.cm(name + '.short_conv', x, mid, 1, out=F.slice(mid, 2 * mid))
        t = self.cm(name + '.main_conv', x, mid, 1, out=F.slice(0, mid) if num_blocks == 0 else None)
        for i in range(num_blocks):
            t = self.cspnext_block('%s.blocks.%d' % (name, i), t, add_identity, out=F.slice(0, mid) if i == num_blocks - 1 else None)
        if channel_attention:
            wfc, bfc = conv_plain(self.ws, name + '.attention.fc', 2 * mid, 2 * mid, 1)
            s = p.conv(p.gavgpool(F), wfc, bfc, act='hsigmoid')
            F = p.scale(F, s, out=F)
        return self.cm(name + '.final_conv', F, cout, 1, out=out)

    def spp(self, name, x, cout):
        p, mid = self.p, x.c // 2
        S = p.buffer(x.n, x.h, x.w, 4 * mid)
        t = self.cm(name + '.conv1', x, mid, 1, out=S.slice(0, mid))
        for i in range(1, 4):                                   # maxpool 5, 9 (=5o5), 13 (=5o5o5): exact for max
            t = p.maxpool(t, 5, 1, 2, out=S.slice(i * mid, (i + 1) * mid))
        return self.cm(name + '.conv2', S, cout, 1)


class _Alias:
    """parameter-name aliases of shared modules: conv weights under `wname`, BN parameters under `bnname`"""
    def __init__(self, ws, name, wname, bnname=None):
        self.ws, self.name, self.wname, self.bnname = ws, name, wname, bnname

    def get(self, n, shape, kind):
        if self.wname is not None and kind == 'conv_w':
            n = self.wname + n[len(self.name):]
        elif self.bnname is not None and kind.startswith('bn_'):
            n = self.bnname + n[len(self.name):]
        return self.ws.get(n, shape, kind)


class RTMDetProgram:
    def __init__(self, prog, cfg, n, h, w, outs):
        self.prog, self.cfg, self.n, self.h, self.w = prog, cfg, n, h, w
        self.cls, self.reg, self.kern, self.mask_feat = outs

    def example_ext(self, dev):
        import torch
        return [torch.randn(self.n, 3, self.h, self.w, device=dev)]


def build_rtmdet(ws, n, h, w, cfg=None):
    """ext tensor [0]: normalised input NCHW [n,3,h,w] (BGR, (x-mean)/std).  Outputs stay in the workspace
    (NHWC): per level cls [n,hl,wl,nc] (sigmoid applied), reg [n,hl,wl,4] (already relu'd, NOT yet x stride), kernels
    [n,hl,wl,169]; mask_feat [n,h/8,w/8,8].  returns (RTMDetProgram, cfg)"""
    cfg = cfg or RTMDetConfig()
    assert h % 32 == 0 and w % 32 == 0
    p = Program("rtmdet")
    B = _B(p, ws, cfg)
    x_ext = p.ext_nchw(n, 3, h, w)
    x = p.to_nhwc(x_ext)
    wf = cfg.widen_factor
    arch = [(int(a * wf), int(b * wf), max(round(c * cfg.deepen_factor), 1), d, e) for a, b, c, d, e in cfg.arch]
    c3, c4, c5 = arch[1][1], arch[2][1], arch[3][1]
    # neck concat buffers first, so backbone outputs land in place (cat([upsample_feat, feat_low], 1))
    TD1 = p.buffer(n, h // 16, w // 16, 2 * c4)
    TD0 = p.buffer(n, h // 8, w // 8, 2 * c3)
    BU0 = p.buffer(n, h // 16, w // 16, 2 * c3)                 # cat([downsample_feat, feat_height], 1)
    BU1 = p.buffer(n, h // 32, w // 32, 2 * c4)
    oc = cfg.feat_channels
    MF = p.buffer(n, h // 8, w // 8, 3 * oc)                    # MaskFeatModule fusion input

    t = B.cm('backbone.stem.0', x, arch[0][0] // 2, 3, stride=2, cin=3)
    t = B.cm('backbone.stem.1', t, arch[0][0] // 2, 3)
    t = B.cm('backbone.stem.2', t, arch[0][0], 3)
    outs_to = {1: TD0.slice(c3, 2 * c3), 2: TD1.slice(c4, 2 * c4)}
    feats = []
    for i, (cin, cout, nb, add_id, use_spp) in enumerate(arch):
        s = 'backbone.stage%d' % (i + 1)
        t = B.cm(s + '.0', t, cout, 3, stride=2)
        k = 1
        if use_spp:
            t = B.spp(s + '.1', t, cout); k = 2
        t = B.csp_layer('%s.%d' % (s, k), t, cout, nb, add_id, True, out=outs_to.get(i))
        feats.append(t)
    C3, C4, C5 = feats[1], feats[2], feats[3]
    nb = max(round(3 * cfg.deepen_factor), 1)
    fh5 = B.cm('neck.reduce_layers.0', C5, c4, 1, out=BU1.slice(c4, 2 * c4))
    p.nearest(fh5, 2, out=TD1.slice(0, c4))
    inner1 = B.csp_layer('neck.top_down_blocks.0', TD1, c4, nb, False, False)
    fh4 = B.cm('neck.reduce_layers.1', inner1, c3, 1, out=BU0.slice(c3, 2 * c3))
    p.nearest(fh4, 2, out=TD0.slice(0, c3))
    o0 = B.csp_layer('neck.top_down_blocks.1', TD0, c3, nb, False, False)
    B.cm('neck.downsamples.0', o0, c3, 3, stride=2, out=BU0.slice(0, c3))
    o1 = B.csp_layer('neck.bottom_up_blocks.0', BU0, c4, nb, False, False)
    B.cm('neck.downsamples.1', o1, c4, 3, stride=2, out=BU1.slice(0, c4))
    o2 = B.csp_layer('neck.bottom_up_blocks.1', BU1, c5, nb, False, False)
    P3 = B.cm('neck.out_convs.0', o0, oc, 3, out=MF.slice(0, oc))
    P4 = B.cm('neck.out_convs.1', o1, oc, 3)
    P5 = B.cm('neck.out_convs.2', o2, oc, 3)
    # MaskFeatModule (rtmdet_ins_head.py): bilinear (align_corners=False) to P3 size, cat, 1x1 fuse, 4x conv, 1x1 proj
    p.bilinear(P4, (P3.h, P3.w), out=MF.slice(oc, 2 * oc))
    p.bilinear(P5, (P3.h, P3.w), out=MF.slice(2 * oc, 3 * oc))
    H = 'bbox_head.'
    wfu, bfu = conv_plain(ws, H + 'mask_head.fusion_conv', oc, 3 * oc, 1)
    m = p.conv(MF, wfu, bfu)
    for i in range(4):
        m = B.cm('%smask_head.stacked_convs.%d' % (H, i), m, oc, 3)
    wpj, bpj = conv_plain(ws, H + 'mask_head.projection', cfg.num_prototypes, oc, 1)
    mask_feat = p.keep(p.conv(m, wpj, bpj))
    # RTMDetInsSepBNHead._init_layers (mmdet 3.x), as the published parameter counts of all five model sizes pin it down
    # (tools/rtmdet_params.py: tiny 5.6 / s 10.18 / m 27.58 / l 57.37 / x 102.7 M are reproduced exactly by this rule and by no
    # simpler one):
    #   * `self.reg_convs.append(cls_convs)`: the reg tower IS the cls tower (the same ConvModule objects, so the state_dict
    #     holds bbox_head.reg_convs.* as aliases of bbox_head.cls_convs.*);
    #   * share_conv ties only cls_convs[n][i].conv (= reg_convs) to level 0; kernel_convs stay per level.
    # A state_dict is always read by its per-level names (aliases included), so either structure imports correctly; the
    # closed-form weights follow the rule above through name aliases.  When the reg tower's folded parameters equal the cls
    # tower's (always, under the rule) its convolutions are the same computation on the same input and are issued once.
    synth = not hasattr(ws, 'sd')
    cls, reg, kern = [], [], []
    for lvl, f in enumerate((P3, P4, P5)):
        def tower_params(kind):
            out = []
            for i in range(cfg.stacked_convs):
                nm, wname, bnname = '%s%s.%d.%d' % (H, kind, lvl, i), None, None
                if synth:
                    if kind == 'reg_convs':
                        wname, bnname = '%scls_convs.%d.%d' % (H, 0 if cfg.share_conv else lvl, i), '%scls_convs.%d.%d' % (H, lvl, i)
                    elif kind == 'cls_convs' and cfg.share_conv:
                        wname = '%scls_convs.0.%d' % (H, i)
                out.append(B.cm_params(nm, oc, oc, 3, wname, bnname))
            return out

        def run_tower(kind, params):
            t_ = f
            for i, wb in enumerate(params):
                t_ = B.cm('%s%s.%d.%d' % (H, kind, lvl, i), t_, oc, 3, params=wb)
            return t_

        def head(t_, name, cout, act=None):
            wh, bh = conv_plain(ws, '%s%s.%d' % (H, name, lvl), cout, oc, 1)
            return p.keep(p.conv(t_, wh, bh, act=act))
        pc, pr = tower_params('cls_convs'), tower_params('reg_convs')
        cls_feat = run_tower('cls_convs', pc)
        cls.append(head(cls_feat, 'rtm_cls', cfg.num_classes, act='sigmoid'))        # cls_score.sigmoid() fused
        kern.append(head(run_tower('kernel_convs', tower_params('kernel_convs')), 'rtm_kernel', cfg.num_gen_params))
        same = all(np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) for a, b in zip(pc, pr))
        reg_feat = cls_feat if same else run_tower('reg_convs', pr)
        reg.append(head(reg_feat, 'rtm_reg', 4, act='relu'))              # F.relu(rtm_reg(.)) * stride  (x stride in decode)
    p.plan()
    return RTMDetProgram(p, cfg, n, h, w, (cls, reg, kern, mask_feat)), cfg
