"""GPU parity of AnimeInsSeg.infer() (HIP path through the drop-in import surface) vs the CPU oracle pipeline:
instance indices, boxes, scores and masks after threshold must be IDENTICAL (north_star: bit-exact masks)."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu


def _img(h, w, seed):
    from cartoonsegmentation_amd import synth
    return synth.image_u8(h, w, seed)


# (64,130) and (134,66): mmdet's ceil(S/scale) lands below W resp. H, so the detector masks are 2 px smaller than the image
# (the `[..., :ori_h, :ori_w]` slice cannot enlarge) and the ISNet refine resizes them back (this used to raise)
@pytest.mark.parametrize("H,W,S,T", [(96, 128, 64, 48), (64, 64, 64, 64), (130, 100, 96, 80), (64, 130, 64, 48), (134, 66, 64, 80)])
def test_infer_matches_oracle(H, W, S, T):
    from animeinsseg import AnimeInsSeg, AnimeInstances
    from cartoonsegmentation_amd.nets import build_isnet, build_rtmdet
    from cartoonsegmentation_amd.weights import SynthWeights
    from oracle import segment as oseg
    img = _img(H, W, 5)
    net = AnimeInsSeg('synthetic', default_det_size=S, refine_kwargs={'refine_method': 'refinenet_isnet', 'refine_size': T})
    inst = net.infer(img, pred_score_thr=0.3, max_instances=3, output_type='numpy')
    assert isinstance(inst, AnimeInstances)
    # oracle: same lowered programs (built independently from the same closed-form weights), CPU execution
    rp, cfg = build_rtmdet(SynthWeights('rtmdet.'), 1, S, S)
    cfg.max_per_img = 3
    d = oseg.detect(img, rp, cfg, S, pred_score_thr=0.3)
    assert d['n'] == len(inst) and d['n'] > 0
    assert np.array_equal(d['scores'], inst.scores)
    assert np.array_equal(d['bboxes'], inst.bboxes)
    progs = {}

    def isnet_for(b):
        if b not in progs:
            progs[b] = build_isnet(SynthWeights('isnet.'), b, T, T)
        return progs[b]
    refined = oseg.refine(img, d['masks'], isnet_for, T, 0.3)
    assert inst.masks.dtype == np.bool_ and inst.masks.shape == (d['n'], H, W)
    assert np.array_equal(refined.astype(bool), inst.masks)
    # unrefined detector masks as well
    net2 = AnimeInsSeg('synthetic', default_det_size=S, refine_kwargs={'refine_method': 'none'})
    raw = net2.infer(img, pred_score_thr=0.3, max_instances=3, output_type='numpy')
    assert np.array_equal(d['masks'].astype(bool), raw.masks)
    if (H, W) == (64, 130):
        assert raw.masks.shape[1:] == (64, 128)
    if (H, W) == (134, 66):
        assert raw.masks.shape[1:] == (132, 66)


def test_api_surface_and_empty_result():
    from animeinsseg import AnimeInsSeg
    from animeinsseg.anime_instances import get_color
    assert len(get_color(3)) == 3
    net = AnimeInsSeg('synthetic', default_det_size=64, refine_kwargs={'refine_method': 'none'})
    out = net.infer([_img(64, 64, 1), _img(64, 96, 2)], pred_score_thr=0.999, max_instances=2)
    assert isinstance(out, list) and len(out) == 2 and all(o.is_empty for o in out)
    t = net.infer(_img(64, 64, 3), pred_score_thr=0.3, max_instances=2)
    assert t.is_tensor and t.is_cuda and t.masks.dtype == torch.bool and t.bboxes.dtype == torch.int32
    m = t.compose_masks()
    assert m.shape == (64, 64)
    t.resize(32, 32)
    assert t.masks.shape[1:] == (32, 32)


def test_embeddings_and_box_prompted_masks():
    from animeinsseg import AnimeInsSeg
    net = AnimeInsSeg('synthetic', default_det_size=64, refine_kwargs={'refine_method': 'none'})
    net.set_max_instance(4)
    img = _img(96, 80, 9)
    im, data, mask_feat = net.infer_embeddings(img)
    assert data['n'] > 0 and data['bboxes'].shape == (data['n'], 4) and mask_feat.shape[-1] == 8
    full = net.infer(img, pred_score_thr=0.0, max_instances=4)
    q = data['bboxes'][:2].cpu().numpy()
    inst = net.segment_with_bboxes(img, q, data, mask_feat)
    assert len(inst) == 2 and inst.masks.shape[1:] == (96, 80)
    # mask_feat is an owned copy: detecting another image in between must not change the prompted masks
    net.infer(_img(96, 80, 10), pred_score_thr=0.0, max_instances=4)
    again = net.segment_with_bboxes(img, q, data, mask_feat)
    assert torch.equal(again.masks, inst.masks)
    # an exact box match returns that detection's own mask (square image-sized rescale == det rescale when H,W <= long side)
    assert inst.bboxes.shape == (2, 4)


def test_long_lists_are_chunked_and_match_single_frames(monkeypatch):
    """infer(list) runs the detector in chunks of CSM_DET_BATCH frames (bounded workspace / program cache); scores, boxes
    and masks are BITWISE those of the per-frame calls: split-K follows the per-sample shape, so a sample's fmaf chains do not
    depend on the batch it runs in"""
    monkeypatch.setenv('CSM_DET_BATCH', '2')
    from animeinsseg import AnimeInsSeg
    net = AnimeInsSeg('synthetic', default_det_size=64, refine_kwargs={'refine_method': 'none'})
    assert net.det_batch == 2
    imgs = [_img(64, 96, 20 + k) for k in range(5)]
    outs = net.infer(imgs, pred_score_thr=0.3, max_instances=2, output_type='numpy')
    assert len(outs) == 5 and set(k[1] for k in net._det_programs) <= {1, 2}
    for im, o in zip(imgs, outs):
        one = net.infer(im, pred_score_thr=0.3, max_instances=2, output_type='numpy')
        assert len(one) == len(o)
        if len(o):
            assert np.array_equal(one.scores, o.scores) and np.array_equal(one.bboxes, o.bboxes)
            assert np.array_equal(one.masks, o.masks)
    # with the ISNet refine in shared batches as well
    net = AnimeInsSeg('synthetic', default_det_size=64, refine_kwargs={'refine_method': 'refinenet_isnet', 'refine_size': 48})
    outs = net.infer(imgs, pred_score_thr=0.3, max_instances=2, output_type='numpy')
    for im, o in zip(imgs, outs):
        one = net.infer(im, pred_score_thr=0.3, max_instances=2, output_type='numpy')
        assert len(one) == len(o) and (len(o) == 0 or (np.array_equal(one.masks, o.masks) and np.array_equal(one.scores, o.scores)))


def test_infer_accepts_a_path_and_a_directory(tmp_path):
    """prepare_data_pipeline (animeinsseg/__init__.py:667-693): `imgs` may be one image path or a directory of images"""
    from PIL import Image
    from animeinsseg import AnimeInsSeg
    imgs = [_img(64, 96, 40 + k) for k in range(3)]
    names = ['b.png', 'a.png', 'c.jpg.png']
    for im, nm in zip(imgs, names):
        Image.fromarray(im[..., ::-1]).save(str(tmp_path / nm))                 # files hold RGB; imread returns BGR like mmcv / cv2
    (tmp_path / 'notes.txt').write_text('not an image')
    net = AnimeInsSeg('synthetic', default_det_size=64, refine_kwargs={'refine_method': 'none'})
    one = net.infer(str(tmp_path / 'a.png'), pred_score_thr=0.3, max_instances=2, output_type='numpy')
    ref = net.infer(imgs[1], pred_score_thr=0.3, max_instances=2, output_type='numpy')
    assert len(one) == len(ref) > 0 and np.array_equal(one.masks, ref.masks) and np.array_equal(one.bboxes, ref.bboxes)
    outs = net.infer(str(tmp_path), pred_score_thr=0.3, max_instances=2, output_type='numpy')
    assert isinstance(outs, list) and len(outs) == 3
    from utils.io_utils import find_all_imgs
    order = [p.split('/')[-1] for p in find_all_imgs(str(tmp_path), abs_path=True)]
    for nm, o in zip(order, outs):
        r = net.infer(imgs[names.index(nm)], pred_score_thr=0.3, max_instances=2, output_type='numpy')
        assert len(o) == len(r) and (len(r) == 0 or np.array_equal(o.masks, r.masks))
