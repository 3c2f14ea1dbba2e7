"""CPU: real-checkpoint import paths (no checkpoint exists offline, so the files are synthesised with the REFERENCE's names).

  * rtmdetl_e60.ckpt  = {'meta': {'cfg': <mmengine pretty_text>}, 'state_dict': {mmdet parameter names}}
    (animeinsseg/__init__.py:195-208) -> config_from_ckpt_cfg + StateDictWeights -> the same packed weights as the closed-form
    source the state_dict was filled from;
  * res101.pth        = {'depth_model': {'module.' + names}}  (depth_modules/leres/__init__.py:83-89);
  * the RTMDet-Ins parameter counts of all five published sizes (mmdet model zoo) are reproduced by the lowered architecture.
"""
import os
import sys

import numpy as np
import pytest

torch = pytest.importorskip("torch")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

# what mmengine's Config.pretty_text stores in ckpt['meta']['cfg'] for an RTMDet-Ins-L run: plain assignments, dict(...) calls,
# lists, nested dicts, `file_client_args` (renamed by the reference before parsing, animeinsseg/__init__.py:197)
CFG_TEXT = '''
default_scope = 'mmdet'
file_client_args = dict(backend='disk')
img_scales = [(640, 640), (320, 320), (960, 960)]
model = dict(
    type='RTMDet',
    data_preprocessor=dict(
        type='DetDataPreprocessor',
        mean=[103.53, 116.28, 123.675],
        std=[57.375, 57.12, 58.395],
        bgr_to_rgb=False,
        batch_augments=None),
    backbone=dict(
        type='CSPNeXt',
        arch='P5',
        expand_ratio=0.5,
        deepen_factor=1,
        widen_factor=1,
        channel_attention=True,
        norm_cfg=dict(type='SyncBN'),
        act_cfg=dict(type='SiLU', inplace=True)),
    neck=dict(
        type='CSPNeXtPAFPN',
        in_channels=[256, 512, 1024],
        out_channels=256,
        num_csp_blocks=3,
        expand_ratio=0.5,
        norm_cfg=dict(type='SyncBN', eps=0.001),
        act_cfg=dict(type='SiLU', inplace=True)),
    bbox_head=dict(
        type='RTMDetInsSepBNHeadCustom',
        num_classes=1,
        in_channels=256,
        stacked_convs=2,
        share_conv=True,
        pred_kernel_size=1,
        feat_channels=256,
        act_cfg=dict(type='SiLU', inplace=True),
        norm_cfg=dict(type='SyncBN', requires_grad=True),
        anchor_generator=dict(type='MlvlPointGenerator', offset=0, strides=[8, 16, 32]),
        bbox_coder=dict(type='DistancePointBBoxCoder'),
        loss_cls=dict(type='QualityFocalLoss', use_sigmoid=True, beta=2.0, loss_weight=1.0),
        loss_bbox=dict(type='GIoULoss', loss_weight=2.0),
        loss_mask=dict(type='DiceLoss', loss_weight=2.0, eps=5e-06, reduction='mean')),
    train_cfg=dict(assigner=dict(type='DynamicSoftLabelAssigner', topk=13), allowed_border=-1, pos_weight=-1, debug=False),
    test_cfg=dict(nms_pre=1000, min_bbox_size=0, score_thr=0.05, nms=dict(type='nms', iou_threshold=0.6), max_per_img=100,
                  mask_thr_binary=0.5))
test_pipeline = [
    dict(type='LoadImageFromFile', file_client_args=dict(backend='disk')),
    dict(type='Resize', scale=(640, 640), keep_ratio=True),
    dict(type='Pad', size=(640, 640), pad_val=dict(img=(114, 114, 114))),
    dict(type='PackDetInputs', meta_keys=('img_id', 'img_path', 'ori_shape', 'img_shape', 'scale_factor'))]
'''


class _Recorder:
    """walks a builder like a weight source and records every parameter it asks for (name -> tensor), under the PER-LEVEL names a
    torch state_dict holds (no aliases: a state_dict lists shared modules under every name)"""

    def __init__(self, src):
        self.src, self.sd = src, {}

    def get(self, name, shape, kind):
        a = self.src.get(name, shape, kind)
        self.sd[name] = torch.from_numpy(np.array(a))
        return a


def test_rtmdet_checkpoint_roundtrip(tmp_path):
    from cartoonsegmentation_amd.nets import build_rtmdet
    from cartoonsegmentation_amd.segmentation import config_from_ckpt_cfg
    from cartoonsegmentation_amd.weights import StateDictWeights, SynthWeights
    cfg = config_from_ckpt_cfg(CFG_TEXT)
    assert (cfg.deepen_factor, cfg.widen_factor, cfg.num_classes, cfg.feat_channels) == (1, 1, 1, 256)
    assert cfg.mean == (103.53, 116.28, 123.675) and cfg.nms_iou == 0.6 and cfg.max_per_img == 100 and cfg.strides == (8, 16, 32)
    assert (cfg.bn_eps_backbone, cfg.bn_eps_neck, cfg.bn_eps_head) == (1e-5, 1e-3, 1e-5)       # norm_cfg eps per sub-module
    # a state_dict with mmdet's names: the closed-form weights, recorded under the names the builder reads from a state_dict
    rec = _Recorder(SynthWeights('rtmdet.'))                    # has .sd -> the builder reads per-level names, like a state_dict
    rp_ref, _ = build_rtmdet(rec, 1, 64, 64, cfg)
    sd = dict(rec.sd)
    assert 'bbox_head.kernel_convs.2.1.conv.weight' in sd and 'bbox_head.reg_convs.1.0.bn.running_var' in sd
    assert 'backbone.stage4.1.conv2.bn.weight' in sd and 'neck.bottom_up_blocks.1.blocks.2.conv2.depthwise_conv.conv.weight' in sd
    sd['bbox_head.cls_convs.0.0.conv.weight_dummy_extra'] = torch.zeros(1)            # strict=False in the reference: extras ignored
    path = str(tmp_path / "rtmdetl_e60.ckpt")
    torch.save({'meta': {'cfg': CFG_TEXT}, 'state_dict': sd}, path)
    blob = torch.load(path, map_location='cpu', weights_only=False)
    rp, _ = build_rtmdet(StateDictWeights(blob['state_dict']), 1, 64, 64, config_from_ckpt_cfg(blob['meta']['cfg']))
    _, _, w_ref = rp_ref.prog.serialise(oracle=False)
    _, _, w = rp.prog.serialise(oracle=False)
    assert w.shape == w_ref.shape and np.array_equal(w, w_ref)
    assert len(rp.prog.ops) == len(rp_ref.prog.ops)


def test_shared_head_towers_follow_the_state_dict():
    """mmdet's RTMDetInsSepBNHead registers the cls tower a second time as the reg tower and ties the cls convs of all levels:
    a state_dict with that structure gets ONE tower computation per level for cls + reg (6 convs fewer); a state_dict whose reg
    tower differs gets its own convolutions"""
    from cartoonsegmentation_amd.nets import build_rtmdet
    from cartoonsegmentation_amd.nets.rtmdet import RTMDetConfig
    from cartoonsegmentation_amd.weights import StateDictWeights, SynthWeights
    cfg = RTMDetConfig(deepen_factor=0.167, widen_factor=0.375, feat_channels=96)
    rec = _Recorder(SynthWeights('rtmdet.'))                    # has .sd -> per-level names, every tower with its own values
    rp_sep, _ = build_rtmdet(rec, 1, 64, 64, cfg)
    n_sep = sum(1 for o in rp_sep.prog.ops if o['kind'] == 1)
    sd = {k: v.clone() for k, v in rec.sd.items()}
    for k in list(sd):                                          # the aliases mmdet's module sharing produces
        if k.startswith('bbox_head.cls_convs.') and k.endswith('conv.weight'):
            lvl, i = k.split('.')[2:4]
            sd[k] = sd['bbox_head.cls_convs.0.%s.conv.weight' % i].clone()
    for k in list(sd):
        if k.startswith('bbox_head.reg_convs.'):
            sd[k] = sd[k.replace('reg_convs', 'cls_convs')].clone()
    rp_mm, _ = build_rtmdet(StateDictWeights(sd), 1, 64, 64, cfg)
    n_mm = sum(1 for o in rp_mm.prog.ops if o['kind'] == 1)
    assert n_mm == n_sep - 3 * cfg.stacked_convs
    # the closed-form weights follow the same structure through name aliases
    rp_syn, _ = build_rtmdet(SynthWeights('rtmdet.'), 1, 64, 64, cfg)
    assert sum(1 for o in rp_syn.prog.ops if o['kind'] == 1) == n_mm


def test_leres_checkpoint_names(tmp_path):
    """res101.pth layout: {'depth_model': {'module.<name>': tensor}} -> strip 'module.' (leres/__init__.py:83-89)"""
    from cartoonsegmentation_amd.nets import build_leres
    from cartoonsegmentation_amd.weights import StateDictWeights, SynthWeights
    rec = _Recorder(SynthWeights('leres.'))
    p_ref = build_leres(rec, 1, 32, 32)
    names = list(rec.sd)
    assert all(n.startswith('depth_model.') for n in names)
    raw = {'module.' + n[len('depth_model.'):]: v for n, v in rec.sd.items()}
    path = str(tmp_path / "res101.pth")
    torch.save({'depth_model': raw}, path)
    sd = torch.load(path, map_location='cpu', weights_only=False)['depth_model']
    ws = StateDictWeights({'depth_model.' + k.replace('module.', '', 1): v for k, v in sd.items()})       # kenburns.py:set_depth_estimation
    p = build_leres(ws, 1, 32, 32)
    assert np.array_equal(p.serialise(oracle=False)[2], p_ref.serialise(oracle=False)[2])


def test_rtmdet_parameter_counts_match_the_published_model_zoo():
    """mmdet configs/rtmdet/README.md, RTMDet-Ins (80 classes): tiny 5.6 M, s 10.18 M, m 27.58 M, l 57.37 M, x 102.7 M"""
    import rtmdet_params as rp
    for size, (_, _, pub) in rp.SIZES.items():
        n = sum(rp.count(size).values()) / 1e6
        digits = 1 if size in ('tiny', 'x') else 2
        assert round(n, digits) == pub, (size, n, pub)
