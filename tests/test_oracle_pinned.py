"""CPU: the oracle restatements (and the product's pure-host classes) against fixtures produced by EXECUTING the reference's
own text for these rows (tests/golden/make_golden_pinned.py): dynamic-conv mask head (a5), mask up-sample / threshold (a6),
ISNet refine glue (a7), AnimeInstances (a9), depth_adjustment_animesseg (a10), process_autozoom (a16).
The `-m gpu` twins (tests/test_gpu_pinned.py) compare the HIP path with the same fixtures."""
import os

import numpy as np
import pytest

from oracle import kenburns as okb, segment as oseg

torch = pytest.importorskip("torch")


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name + '.npz'))


@pytest.mark.parametrize("name", ["pin_maskhead_20x20", "pin_maskhead_12x28"])
def test_maskhead_logits(golden_dir, name):
    d = _load(golden_dir, name)
    feat = np.ascontiguousarray(d['mask_feat'][0].transpose(1, 2, 0))              # the engine keeps activations NHWC
    got = oseg.maskhead_logits(feat, d['kernels'], d['priors'], 8)
    ref = d['logits']
    # torch's grouped conv2d sums the 10 / 8 / 8 products in its own order: fp32 reassociation only
    assert np.abs(got - ref).max() <= 2e-5 * max(1.0, float(np.abs(ref).max()))


@pytest.mark.parametrize("name", ["pin_boxprompt_100x140", "pin_boxprompt_152x96"])
def test_box_prompted_masks(golden_dir, name):
    """segment_with_bboxes: best-IoU match, x8 bilinear, resize to [long,long], crop, sigmoid > 0.5, xywh int32"""
    d = _load(golden_dir, name)
    H, W = int(d['H']), int(d['W'])
    q, t = d['query'], d['boxes']
    lt, rb = np.maximum(q[:, None, :2], t[None, :, :2]), np.minimum(q[:, None, 2:], t[None, :, 2:])
    inter = np.clip(rb - lt, 0, None).prod(2)
    area = lambda b: (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])                       # noqa: E731
    idx = (inter / (area(q)[:, None] + area(t)[None] - inter)).argmax(1)
    feat = np.ascontiguousarray(d['mask_feat'][0].transpose(1, 2, 0))
    logits = oseg.maskhead_logits(feat, d['kernels'][idx], d['priors'][idx], 8)
    long_side = max(H, W)
    masks = oseg.mask_resize_threshold(logits, 8, long_side, long_side, H, W, 0.5)
    assert masks.shape == d['masks'].shape
    assert (masks.astype(bool) != d['masks']).mean() <= 2e-4                          # pixels whose sigmoid sits on 0.5 +- 1e-6
    bb = t[idx].astype(np.int32); bb[:, 2:] -= bb[:, :2]
    assert np.array_equal(bb, d['out_bboxes']) and np.array_equal(d['scores'][idx], d['out_scores'])


@pytest.mark.parametrize("name", ["pin_refine_90x74_T96", "pin_refine_64x64_T64"])
def test_refine_glue(golden_dir, name):
    """prepare_refine_batch layout (bit-exact) and the sigmoid / crop / align_corners resize / threshold tail"""
    d = _load(golden_dir, name)
    img, T = d['img'], int(d['T'])
    H, W = img.shape[:2]
    batch = oseg.refine_prepare_batch(img, d['masks_in'].astype(np.uint8), H, W, T)
    assert np.array_equal(batch, d['batch'])
    logits = ((d['logits_raw'] - np.float32(d['centre'])) / np.float32(d['scale'])).astype(np.float32)
    out = oseg.refine_threshold(logits, H, W, H, W, 0.3)
    diff = out.astype(bool) != d['masks_out']
    assert diff.mean() <= 1e-4
    assert np.all(np.abs(d['probs'][diff] - 0.3) < 1e-5)                               # only borderline pixels may flip


def test_anime_instances_against_the_reference_class(golden_dir):
    from cartoonsegmentation_amd.anime_instances import AnimeInstances
    d = _load(golden_dir, "pin_instances")
    mk = lambda: AnimeInstances(torch.from_numpy(d['masks'].copy()), torch.from_numpy(d['bboxes'].copy()),   # noqa: E731
                                torch.from_numpy(d['scores'].copy()))
    for tag in ('down', 'up', 'same'):
        a = mk()
        h, w = (int(v) for v in d['resize_%s_hw' % tag])
        a.resize(h, w)
        assert np.array_equal(a.masks.numpy(), d['resize_%s_masks' % tag]), tag
        assert np.array_equal(a.bboxes.numpy(), d['resize_%s_bboxes' % tag]), tag
    a = mk()
    assert np.array_equal(a.compose_masks().numpy(), d['compose'])
    a.remove_duplicated()
    assert np.array_equal(a.masks.numpy(), d['dedup_masks']) and np.array_equal(a.bboxes.numpy(), d['dedup_bboxes'])
    assert np.array_equal(a.scores.numpy(), d['dedup_scores'])
    b = AnimeInstances(d['masks'].copy(), d['bboxes'].copy(), d['scores'].copy())
    b.remove_duplicated()
    assert b.is_numpy and np.array_equal(b.masks, d['dedup_masks'])


def test_depth_adjustment(golden_dir):
    d = _load(golden_dir, "pin_depth_adjust")
    assert np.array_equal(okb.depth_adjustment(d['masks'], d['disp']), d['adjusted'])
    assert np.array_equal(okb.depth_adjustment(d['masks'], d['disp'], use_medium=True), d['adjusted_median'])
    assert np.array_equal(okb.depth_adjustment([], d['disp']), d['adjusted_empty'])


def test_autozoom_search(golden_dir):
    """process_autozoom: the sequential (raster in-place degrid) execution reproduces the reference's coverage count of every
    candidate; the Jacobi degrid of the HIP build picks the same target"""
    d = _load(golden_dir, "pin_autozoom_96x128")
    H, W = int(d['H']), int(d['W'])
    dr = d['depthrange']
    kc = dict(depthrange=(float(dr[0]), float(dr[1]), (int(dr[2]), int(dr[3]))), pts=d['pts'])
    counts = []
    to, _ = okb.autozoom_target(kc, d['rgb'], W, H, float(d['focal']), float(d['baseline']), shift=float(d['shift']), degrid_mode=0,
                                counts_out=counts)
    assert np.array_equal(np.asarray(counts), d['counts'])
    assert [to['fltCenterU'], to['fltCenterV'], to['intCropWidth'], to['intCropHeight']] == list(d['objTo'])
    counts_j = []
    to_j, _ = okb.autozoom_target(kc, d['rgb'], W, H, float(d['focal']), float(d['baseline']), shift=float(d['shift']), degrid_mode=1,
                                  counts_out=counts_j)
    # updateDegrid is an in-place read/write race in the reference (models/utils.py:152-212): on a GPU its coverage counts are
    # neither the raster nor the Jacobi numbers.  The two orders differ by < 0.5 % of the pixels per candidate here, and the
    # candidate the Jacobi order picks is (by the REFERENCE's own counts) within one pixel of the reference's best.
    cj, c = np.asarray(counts_j), d['counts']
    assert np.abs(cj - c).max() <= 0.005 * H * W
    assert c[int(np.argmax(cj))] >= c.max() - 0.001 * H * W
    assert to_j['intCropWidth'] == to['intCropWidth'] and to_j['intCropHeight'] == to['intCropHeight']
