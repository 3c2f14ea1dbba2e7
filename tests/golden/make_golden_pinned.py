#!/usr/bin/env python3
"""Generate tests/golden/pin_*.npz by EXECUTING the reference's own text for the rows that round 1 only restated
(VERDICT r01 "next" 1a): the dynamic-conv mask head, the mask up-sampling / threshold sequence, the ISNet refine tail,
AnimeInstances.resize / remove_duplicated / compose_masks, depth_adjustment_animesseg and process_autozoom.

Build container only (needs /root/reference).  Files that import mmdet / mmcv / cv2 / omegaconf at module level are not
imported: single definitions are compiled from the file's text at run time (ref_loader.extract_def) and run on CPU tensors.
The only helpers supplied from outside the reference tree are the three mmdet/torchvision functions the vendored text calls
and does not define -- MlvlPointGenerator.single_level_grid_priors(offset=0), RTMDetInsHead.parse_dynamic_params
(mmdet 3.3.0) and torchvision.ops.box_iou -- restated below and marked [EXT].  Stored: inputs and expected outputs only.
"""
import os
import sys
import types

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE); sys.path.insert(0, ROOT)
import ref_loader  # noqa: E402
from cartoonsegmentation_amd import synth  # noqa: E402

ref_loader.install_stubs()
AnimeInstances = ref_loader.load_anime_instances()
AnimeInstances.draw_instances = lambda self, *a, **k: None      # visualisation only (cv2 drawing), not part of any result


# ---- [EXT] helpers the vendored text calls -------------------------------------------------------------------------------------
class PointGenerator:
    """mmdet MlvlPointGenerator(offset=0, strides=[8,16,32]).single_level_grid_priors(with_stride=False)"""
    def __init__(self, strides=(8, 16, 32)):
        self.strides = [(s, s) for s in strides]

    def single_level_grid_priors(self, featmap_size, level_idx, dtype=torch.float32, device='cpu', with_stride=False):
        fh, fw = featmap_size
        sw, sh = self.strides[level_idx]
        sx = (torch.arange(0, fw, device=device) + 0) * sw
        sy = (torch.arange(0, fh, device=device) + 0) * sh
        yy, xx = torch.meshgrid(sy.to(dtype), sx.to(dtype), indexing='ij')
        return torch.stack([xx.reshape(-1), yy.reshape(-1)], dim=-1)


class FakeHead:
    """what RTMDetInsSepBNHeadCustom._mask_predict_by_feat_single reads from `self`"""
    num_prototypes, dyconv_channels, num_dyconvs = 8, 8, 3

    def __init__(self):
        self.prior_generator = PointGenerator()
        d, p = self.dyconv_channels, self.num_prototypes
        self.weight_nums = [(p + 2) * d, d * d, d]
        self.bias_nums = [d, d, 1]

    def parse_dynamic_params(self, flatten_kernels):            # mmdet 3.3.0 RTMDetInsHead.parse_dynamic_params
        n_inst = flatten_kernels.size(0)
        n_layers = len(self.weight_nums)
        splits = list(torch.split_with_sizes(flatten_kernels, self.weight_nums + self.bias_nums, dim=1))
        ws, bs = splits[:n_layers], splits[n_layers:]
        for i in range(n_layers):
            if i < n_layers - 1:
                ws[i] = ws[i].reshape(n_inst * self.dyconv_channels, -1, 1, 1)
                bs[i] = bs[i].reshape(n_inst * self.dyconv_channels)
            else:
                ws[i] = ws[i].reshape(n_inst, -1, 1, 1)
                bs[i] = bs[i].reshape(n_inst)
        return ws, bs


def box_iou(a, b):                                             # torchvision.ops.boxes.box_iou
    area = lambda t: (t[:, 2] - t[:, 0]) * (t[:, 3] - t[:, 1])  # noqa: E731
    lt, rb = torch.max(a[:, None, :2], b[:, :2]), torch.min(a[:, None, 2:], b[:, 2:])
    wh = (rb - lt).clamp(min=0)
    inter = wh[:, :, 0] * wh[:, :, 1]
    return inter / (area(a)[:, None] + area(b) - inter)


HEAD_NS = dict(torch=torch, F=F, Tensor=torch.Tensor,
               sthgoeswrong=lambda x: bool(torch.any(torch.isnan(x)) or torch.any(torch.isinf(x))))
mask_predict = ref_loader.extract_def("animeinsseg/models/rtmdet_inshead_custom.py",
                                      "RTMDetInsSepBNHeadCustom._mask_predict_by_feat_single", HEAD_NS)
FakeHead._mask_predict_by_feat_single = mask_predict


def _detections(g, n, S, hw):
    """plausible detector outputs: priors on the stride grids, kernels ~ N(0, 0.6), boxes in image coordinates"""
    strides = g.choice([8, 16, 32], n)
    px = np.array([g.integers(0, S // s) * s for s in strides], np.float32)
    py = np.array([g.integers(0, S // s) * s for s in strides], np.float32)
    priors = np.stack([px, py, strides.astype(np.float32), strides.astype(np.float32)], 1)
    kernels = g.normal(0, 0.6, (n, 169)).astype(np.float32)
    H, W = hw
    x1 = g.uniform(0, W * 0.6, n); y1 = g.uniform(0, H * 0.6, n)
    boxes = np.stack([x1, y1, x1 + g.uniform(8, W * 0.4, n), y1 + g.uniform(8, H * 0.4, n)], 1).astype(np.float32)
    scores = np.sort(g.uniform(0.3, 0.99, n).astype(np.float32))[::-1].copy()
    return priors, kernels, boxes, scores


def maskhead_case(name, h, w, n, seed):
    """a5: rtmdet_inshead_custom.py:253-303"""
    g = np.random.default_rng(seed)
    feat = g.normal(0, 1, (1, 8, h, w)).astype(np.float32)
    priors, kernels, _, _ = _detections(g, n, 8 * max(h, w), (8 * h, 8 * w))
    with torch.no_grad():
        logits = FakeHead()._mask_predict_by_feat_single(torch.from_numpy(feat), torch.from_numpy(kernels), torch.from_numpy(priors))
    np.savez_compressed(os.path.join(HERE, name + '.npz'), mask_feat=feat, kernels=kernels, priors=priors, logits=logits.numpy())
    print(name, tuple(logits.shape), float(logits.mean()), float(logits.std()))


def boxprompt_case(name, H, W, S, nq, seed):
    """a6 (+a5): AnimeInsSeg.segment_with_bboxes, animeinsseg/__init__.py:339-393 -- x8 bilinear, resize to
    [long_side, long_side], crop, sigmoid > 0.5, xyxy -> xywh int32.  The refine step is pinned separately."""
    g = np.random.default_rng(seed)
    h, w = S // 8, S // 8
    feat = g.normal(0, 1, (1, 8, h, w)).astype(np.float32)
    n = 7
    priors, kernels, boxes, scores = _detections(g, n, S, (H, W))
    q = boxes[g.permutation(n)[:nq]] + g.uniform(-3, 3, (nq, 4)).astype(np.float32)
    ns = dict(torch=torch, np=np, F=F, box_iou=box_iou, AnimeInstances=AnimeInstances)
    seg = ref_loader.extract_def("animeinsseg/__init__.py", "AnimeInsSeg.segment_with_bboxes", ns)
    fake = types.SimpleNamespace(model=types.SimpleNamespace(bbox_head=FakeHead()), _postprocess_refine=lambda *a, **k: None)
    data = types.SimpleNamespace(scores=torch.from_numpy(scores), bboxes=torch.from_numpy(boxes), priors=torch.from_numpy(priors),
                                 kernels=torch.from_numpy(kernels))
    img = synth.image_u8(H, W, seed)
    with torch.no_grad():
        inst = seg(fake, img, [b for b in q], data, torch.from_numpy(feat))
    np.savez_compressed(os.path.join(HERE, name + '.npz'), H=H, W=W, S=S, mask_feat=feat, kernels=kernels, priors=priors, boxes=boxes,
                        scores=scores, query=q, masks=np.asarray(inst.masks), out_bboxes=np.asarray(inst.bboxes),
                        out_scores=np.asarray(inst.scores, np.float32))
    print(name, np.asarray(inst.masks).shape, float(np.asarray(inst.masks).mean()))


def refine_case(name, H, W, T, n, seed, as_tensor):
    """a7: AnimeInsSeg._postprocess_refine + prepare_refine_batch + resize_pad (animeinsseg/__init__.py:37-55, :638-665,
    utils/io_utils.py:254-292) around the reference ISNetDIS filled with the closed-form weights.  max(H,W) <= T, so the
    un-vendored cv2.resize is never reached (the stub raises)."""
    from make_golden_nets import fill_synthetic
    ref_loader._bare("animeinsseg.models"); ref_loader._bare("animeinsseg.models.animeseg_refine")
    isn = ref_loader.load_by_path("animeinsseg.models.animeseg_refine.isnet", "animeinsseg/models/animeseg_refine/isnet.py")
    net = fill_synthetic(isn.ISNetDIS(in_ch=4), 'isnet.')
    cv2 = ref_loader.cv2_stub()
    io_ns = dict(np=np, cv2=cv2, Tuple=tuple)
    scaledown = ref_loader.extract_def("utils/io_utils.py", "scaledown_maxsize", io_ns)
    resize_pad = ref_loader.extract_def("utils/io_utils.py", "resize_pad", io_ns)
    ns = dict(torch=torch, np=np, F=F, AnimeInstances=AnimeInstances, resize_pad=resize_pad)
    ref_loader.extract_def("animeinsseg/__init__.py", "prepare_refine_batch", ns)
    refine = ref_loader.extract_def("animeinsseg/__init__.py", "AnimeInsSeg._postprocess_refine", ns)
    g = np.random.default_rng(seed)
    img = synth.image_u8(H, W, seed)
    masks = np.zeros((n, H, W), bool)
    for k in range(n):
        cy, cx, ry, rx = g.uniform(0.2, 0.8) * H, g.uniform(0.2, 0.8) * W, g.uniform(0.15, 0.4) * H, g.uniform(0.15, 0.4) * W
        yy, xx = np.mgrid[0:H, 0:W]
        masks[k] = ((yy - cy) / ry) ** 2 + ((xx - cx) / rx) ** 2 < 1
    boxes = np.tile(np.array([[0, 0, W, H]], np.int32), (n, 1))
    inst = AnimeInstances(torch.from_numpy(masks.copy()) if as_tensor else masks.copy(), torch.from_numpy(boxes) if as_tensor else boxes,
                          torch.ones(n) if as_tensor else np.ones(n, np.float32))
    # The closed-form weights drive ISNet's d1 to +16 +- 6, where sigmoid() > 0.3 is true everywhere.  The refine tail treats
    # `self.refinenet` as an opaque callable, so the fixture runs it with the logits re-centred, (d1 - c) / s with c, s stored in
    # the fixture: the threshold then cuts through the map and the crop / align_corners resize / threshold sequence is exercised.
    with torch.no_grad():
        batches = [b for b, _ in ns['prepare_refine_batch'](masks.astype(np.float32), img, 4, 'cpu', T)]
        raw = torch.cat([net(b)[0][0] for b in batches])
    c, sdev = float(raw.median()), float(raw.std())

    def centred(batch):
        return [[(net(batch)[0][0] - c) / sdev]]
    fake = types.SimpleNamespace(refinenet=centred, device='cpu', mask_thr=0.3)
    refine(fake, inst, img, refine_size=T, max_refine_batch=4)
    out = inst.masks.numpy() if as_tensor else inst.masks
    # the float probabilities as well, so a test can tell borderline pixels (|p - thr| tiny) from real differences
    probs = []
    with torch.no_grad():
        for batch, (pt, pb, pl, pr) in ns['prepare_refine_batch'](masks.astype(np.float32), img, 4, 'cpu', T):
            p = centred(batch)[0][0].sigmoid()
            p = p[..., pt: -(pb if pb else -H), pl: -(pr if pr else -W)]
            probs.append(F.interpolate(p, (H, W), mode='bilinear', align_corners=True)[:, 0])
    probs = torch.cat(probs).numpy()
    assert np.array_equal(probs > 0.3, out)
    np.savez_compressed(os.path.join(HERE, name + '.npz'), img=img, masks_in=masks, T=T, masks_out=out, probs=probs.astype(np.float32),
                        batch=torch.cat(batches).numpy(), logits_raw=raw.numpy(), centre=c, scale=sdev)
    print(name, out.shape, float(out.mean()), 'flipped vs input %.4f' % float((out != masks).mean()))


def instances_case(name, seed):
    """a9: AnimeInstances.resize / remove_duplicated / compose_masks (anime_instances.py:84-127, :268-298)"""
    g = np.random.default_rng(seed)
    H, W, n = 60, 84, 6
    masks = np.zeros((n, H, W), bool)
    yy, xx = np.mgrid[0:H, 0:W]
    for k in range(n):
        cy, cx, r = g.uniform(10, H - 10), g.uniform(10, W - 10), g.uniform(6, 22)
        masks[k] = (yy - cy) ** 2 + (xx - cx) ** 2 < r * r
    masks[4] = masks[1] & (xx < W // 2 + 20)          # mostly covered by a larger one -> removed as duplicate
    masks[5] = masks[0]                               # exact duplicate
    boxes = g.integers(0, 40, (n, 4)).astype(np.int32)
    scores = g.uniform(0.3, 1, n).astype(np.float32)
    out = dict(masks=masks, bboxes=boxes, scores=scores)
    for tag, (h, w) in (('down', (33, 50)), ('up', (90, 100)), ('same', (H, W))):
        a = AnimeInstances(torch.from_numpy(masks.copy()), torch.from_numpy(boxes.copy()), torch.from_numpy(scores.copy()))
        a.resize(h, w)
        out['resize_%s_hw' % tag] = np.array([h, w]); out['resize_%s_masks' % tag] = a.masks.numpy()
        out['resize_%s_bboxes' % tag] = a.bboxes.numpy()
    a = AnimeInstances(torch.from_numpy(masks.copy()), torch.from_numpy(boxes.copy()), torch.from_numpy(scores.copy()))
    out['compose'] = a.compose_masks().numpy()
    a.remove_duplicated()
    out['dedup_masks'], out['dedup_bboxes'], out['dedup_scores'] = a.masks.numpy(), a.bboxes.numpy(), a.scores.numpy()
    b = AnimeInstances(masks.copy(), boxes.copy(), scores.copy())
    b.remove_duplicated()
    assert b.is_numpy and np.array_equal(b.masks, out['dedup_masks'])
    np.savez_compressed(os.path.join(HERE, name + '.npz'), **out)
    print(name, 'kept', len(a), 'of', n)


def depth_adjust_case(name, seed):
    """a10: depth_adjustment_animesseg (kenburns_effect.py:39-91), both branches, with and without the resize round trip"""
    ns = dict(torch=torch, np=np, AnimeInstances=AnimeInstances)
    fn = ref_loader.extract_def("anime_3dkenburns/kenburns_effect.py", "depth_adjustment_animesseg", ns)
    g = np.random.default_rng(seed)
    H, W = 72, 96
    sc = synth.warp_scene(H, W, seed)
    disp = torch.from_numpy(sc['disp'].reshape(1, 1, H, W).copy())
    yy, xx = np.mgrid[0:H, 0:W]
    masks = np.stack([((yy - 30) / 18.0) ** 2 + ((xx - 30) / 14.0) ** 2 < 1, ((yy - 50) / 15.0) ** 2 + ((xx - 64) / 20.0) ** 2 < 1,
                      np.zeros((H, W), bool), (yy > H - 6) & (xx > 70)])
    inst = AnimeInstances(torch.from_numpy(masks), torch.zeros((4, 4), dtype=torch.int32), torch.ones(4))
    img = torch.zeros(1, 3, H, W)
    out = dict(disp=disp.numpy(), masks=masks)
    out['adjusted'] = fn(inst, disp.clone(), img, False).numpy()
    out['adjusted_median'] = fn(inst, disp.clone(), img, True).numpy()
    small = F.interpolate(disp, size=(H // 2, W // 2), mode='bilinear', align_corners=False)
    out['disp_small'] = small.numpy()
    out['adjusted_resized'] = fn(inst, small.clone(), img, False).numpy()
    out['adjusted_empty'] = fn(AnimeInstances(), disp.clone(), img, False).numpy()
    np.savez_compressed(os.path.join(HERE, name + '.npz'), **out)
    print(name, float(np.abs(out['adjusted'] - out['disp']).max()), float(np.abs(out['adjusted_median'] - out['disp']).max()))


def autozoom_case(name, H, W, seed):
    """a16: process_autozoom (common.py:86-142) through the reference's own render_pointcloud (CUDA text on the CPU shim);
    the per-candidate coverage counts are recorded by wrapping render_pointcloud."""
    mu, co, _ = ref_loader.load_warp_modules()
    sc = synth.warp_scene(H, W, seed)
    focal, baseline = sc['focal'], sc['baseline']
    disp = torch.from_numpy(sc['disp'].reshape(1, 1, H, W).copy())
    disp = disp / disp.max() * baseline
    depth = (focal * baseline) / (disp + 0.00001)
    valid = (mu.spatial_filter(disp / disp.max(), 'laplacian').abs() < 0.03).float()
    pts = mu.depth_to_points(depth * valid, focal).view(1, 3, -1)
    b = max(2, min(H, W) // 8)                                   # the reference crops 128 px at 1024^2
    crop = depth[0, 0, b:-b, b:-b]
    dmin = float(crop.min()); loc = int(crop.argmin())
    common = {'objDepthrange': (dmin, float(crop.max()), (loc % crop.shape[1], loc // crop.shape[1]), (0, 0)),
              'intWidth': W, 'intHeight': H, 'fltFocal': focal, 'fltBaseline': baseline,
              'tenRawPoints': pts.contiguous(), 'tenRawImage': torch.from_numpy(sc['rgb'].reshape(1, 3, H, W).copy())}
    objFrom = {'fltCenterU': W / 2.0, 'fltCenterV': H / 2.0, 'intCropWidth': int(np.floor(0.97 * W)), 'intCropHeight': int(np.floor(0.97 * H))}
    counts = []
    orig = co.render_pointcloud

    def recording(*a, **k):
        r, e = orig(*a, **k)
        counts.append(float((e > 0.0).float().sum().item()))
        return r, e
    co.render_pointcloud = recording
    shift = 100.0 * W / 1024.0
    objTo = co.process_autozoom({'fltShift': shift, 'fltZoom': 1.25, 'objFrom': objFrom}, common)
    co.render_pointcloud = orig
    np.savez_compressed(os.path.join(HERE, name + '.npz'), H=H, W=W, focal=focal, baseline=baseline, disp_raw=sc['disp'], rgb=sc['rgb'],
                        pts=pts.numpy(), depth=depth.numpy(), depthrange=np.array([common['objDepthrange'][0], common['objDepthrange'][1],
                                                                                     common['objDepthrange'][2][0], common['objDepthrange'][2][1]], np.float64),
                        shift=shift, counts=np.asarray(counts, np.float64),
                        objTo=np.array([objTo['fltCenterU'], objTo['fltCenterV'], objTo['intCropWidth'], objTo['intCropHeight']], np.float64))
    print(name, 'candidates', len(counts), 'objTo', objTo)


def median_case(name, seed):
    """a12: spatial_filter(x, 'median-3' / 'median-5') (models/utils.py:26-36) executed: a smooth map, a binary mask (heavy ties), B = 2"""
    mu, _, _ = ref_loader.load_warp_modules()
    g = np.random.default_rng(seed)
    x = g.normal(0, 1, (2, 2, 23, 31)).astype(np.float32)
    x[1] = (g.uniform(0, 1, (2, 23, 31)) > 0.45).astype(np.float32)
    t = torch.from_numpy(x)
    np.savez_compressed(os.path.join(HERE, name + '.npz'), x=x, median3=mu.spatial_filter(t, 'median-3').numpy(),
                        median5=mu.spatial_filter(t, 'median-5').numpy())
    print(name, x.shape)


if __name__ == '__main__':
    torch.manual_seed(0)
    which = sys.argv[1:] or ['maskhead', 'boxprompt', 'refine', 'instances', 'depth', 'autozoom', 'median']
    if 'median' in which:
        median_case('pin_spatial_filter_median', 50)
    if 'maskhead' in which:
        maskhead_case('pin_maskhead_20x20', 20, 20, 5, 41)
        maskhead_case('pin_maskhead_12x28', 12, 28, 3, 42)
    if 'boxprompt' in which:
        boxprompt_case('pin_boxprompt_100x140', 100, 140, 160, 3, 43)
        boxprompt_case('pin_boxprompt_152x96', 152, 96, 128, 2, 44)
    if 'refine' in which:
        refine_case('pin_refine_90x74_T96', 90, 74, 96, 5, 45, True)
        refine_case('pin_refine_64x64_T64', 64, 64, 64, 2, 46, False)
    if 'instances' in which:
        instances_case('pin_instances', 47)
    if 'depth' in which:
        depth_adjust_case('pin_depth_adjust', 48)
    if 'autozoom' in which:
        autozoom_case('pin_autozoom_96x128', 96, 128, 49)
