"""CPU: the RTMDet-Ins graph of the product (nets/rtmdet.py -> program.py, executed by the oracle interpreter) against an
INDEPENDENT torch.nn restatement of mmdet 3.3.0's modules (oracle/rtmdet_torch.py: plain nn.Modules with mmdet's attribute names, no
lowering, parallel SPP pools, real torch.cat).  The two share only a state_dict with mmdet's parameter names -- what a real
rtmdetl_e60.ckpt hands the product -- so a wiring error in the lowering (concat order, slice offsets, which tower feeds rtm_reg,
top-down / bottom-up order, MaskFeat fusion order) cannot cancel.  Reference call sites: animeinsseg/__init__.py:196-215, :450.

mmdet itself is not under /root/reference, so this pins the product's wiring against a second statement of the published model, not
against mmdet's own code.
"""
import numpy as np
import pytest

torch = pytest.importorskip("torch")

from cartoonsegmentation_amd import synth  # noqa: E402
from cartoonsegmentation_amd.nets import build_rtmdet  # noqa: E402
from cartoonsegmentation_amd.nets.rtmdet import RTMDetConfig  # noqa: E402
from cartoonsegmentation_amd.weights import StateDictWeights  # noqa: E402
from oracle import nets as onets, rtmdet_torch as rt, segment as oseg  # noqa: E402


def _rel(a, b):
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-12))


def _build(cfg, S):
    model = rt.fill_closed_form(rt.RTMDetIns(cfg.deepen_factor, cfg.widen_factor, cfg.expand_ratio, cfg.num_classes, cfg.feat_channels,
                                             cfg.stacked_convs, cfg.share_conv, eps_backbone=1e-5, eps_neck=1e-3, eps_head=1e-5))
    cfg.bn_eps_backbone, cfg.bn_eps_neck, cfg.bn_eps_head = 1e-5, 1e-3, 1e-5       # one eps per sub-module, as a ckpt cfg may carry
    sd = {k: v for k, v in model.state_dict().items() if not k.endswith('num_batches_tracked')}
    rp, _ = build_rtmdet(StateDictWeights(sd), 1, S, S, cfg)
    return model, rp, sd


@pytest.mark.parametrize("name,cfg,S", [
    ("l", RTMDetConfig(), 96),                                                        # RTMDet-Ins-L, the shipped detector
    ("tiny", RTMDetConfig(deepen_factor=0.167, widen_factor=0.375, feat_channels=96), 128),
])
def test_lowered_graph_matches_independent_torch_modules(name, cfg, S):
    model, rp, sd = _build(cfg, S)
    # mmdet's module sharing is visible in the state_dict: reg tower = cls tower, cls convs tied across levels, BNs per level
    assert torch.equal(sd['bbox_head.reg_convs.1.0.conv.weight'], sd['bbox_head.cls_convs.0.0.conv.weight'])
    assert torch.equal(sd['bbox_head.reg_convs.2.1.bn.weight'], sd['bbox_head.cls_convs.2.1.bn.weight'])
    assert not torch.equal(sd['bbox_head.cls_convs.1.0.bn.weight'], sd['bbox_head.cls_convs.0.0.bn.weight'])
    assert not torch.equal(sd['bbox_head.kernel_convs.1.0.conv.weight'], sd['bbox_head.kernel_convs.0.0.conv.weight'])
    img = synth.image_u8(S, S, 77)
    x = oseg.det_preprocess(img, S, cfg, S, S)
    views = onets.run_program(rp.prog, [x], want_views=rp.cls + rp.reg + rp.kern + [rp.mask_feat])
    with torch.no_grad():
        cls_t, reg_t, kern_t, mf_t = model(torch.from_numpy(x))
    nhwc = lambda t: t[0].permute(1, 2, 0).numpy()                                    # noqa: E731
    for lvl, stride in enumerate(cfg.strides):
        assert _rel(views[rp.cls[lvl]][0], nhwc(cls_t[lvl].sigmoid())) < 1e-4, (name, 'cls', lvl)
        assert _rel(views[rp.reg[lvl]][0] * np.float32(stride), nhwc(reg_t[lvl])) < 1e-4, (name, 'reg', lvl)
        assert _rel(views[rp.kern[lvl]][0], nhwc(kern_t[lvl])) < 1e-4, (name, 'kernel', lvl)
    assert _rel(views[rp.mask_feat][0][..., :cfg.num_prototypes], nhwc(mf_t)) < 1e-4, (name, 'mask_feat')
    # the outputs must depend on every input region and not be degenerate (a dead branch would also "match")
    assert all(np.std(views[v]) > 1e-4 for v in rp.cls + rp.reg + rp.kern)


def test_decode_nms_and_masks_match_independent_restatement():
    """predict_by_feat / _bbox_mask_post_process: oracle/segment.py::detect (numpy + post_oracle.c) vs oracle/rtmdet_torch.py::decode
    (torch ops, written separately) on the SAME raw head outputs -> same kept set, order, boxes, scores; masks identical except
    where the mask probability sits within float noise of the threshold."""
    cfg, S = RTMDetConfig(deepen_factor=0.167, widen_factor=0.375, feat_channels=96), 128
    model, rp, _ = _build(cfg, S)
    H, W = 100, 120
    img = synth.image_u8(H, W, 78)
    f = min(S / max(H, W), S / min(H, W))
    rh, rw = int(H * f + 0.5), int(W * f + 0.5)
    cfg.max_per_img = 5
    d = oseg.detect(img, rp, cfg, S, pred_score_thr=0.0)
    assert d['n'] > 1
    x = oseg.det_preprocess(img, S, cfg, rh, rw)
    views = onets.run_program(rp.prog, [x], want_views=rp.cls + rp.reg + rp.kern + [rp.mask_feat])
    nchw = lambda a: torch.from_numpy(np.ascontiguousarray(a.transpose(0, 3, 1, 2)))   # noqa: E731
    inv_sig = lambda p: torch.log(p / (1 - p))                                          # noqa: E731  (decode applies sigmoid itself)
    raw = ([inv_sig(nchw(views[v]).double()).float() for v in rp.cls],
           [nchw(views[v] * np.float32(s)) for v, s in zip(rp.reg, cfg.strides)], [nchw(views[v]) for v in rp.kern],
           nchw(views[rp.mask_feat][..., :cfg.num_prototypes]))
    t = rt.decode(model.bbox_head, raw, (rh, rw), (H, W), (rw / W, rh / H), cfg.score_thr, cfg.nms_pre, cfg.nms_iou, cfg.max_per_img,
                  cfg.min_bbox_size, cfg.mask_thr_binary)
    assert t['n'] == d['n']
    assert np.allclose(t['scores'], d['scores'], atol=2e-6)                             # sigmoid(logit(p)) round trip
    assert np.allclose(t['bboxes'], d['boxes_f'], rtol=1e-6, atol=1e-4)
    assert _rel(t['logits'], d['logits']) < 1e-5
    sure = np.abs(t['mask_prob'] - cfg.mask_thr_binary) > 1e-4
    assert t['masks'].shape == d['masks'].shape and np.array_equal(t['masks'][sure], d['masks'].astype(bool)[sure]) and sure.mean() > 0.99
