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


def test_device_decode_matches_filter_then_topk_with_ties_and_classes():
    """csm_det_decode + csm_det_gather (hand-written top-k / merge sort / box decode, csrc/detdecode.hip) against a numpy statement of
    mmdet's filter_scores_and_topk -> distance2bbox -> rescale -> min_bbox_size filter -> score sort [EXT], on random head maps with
    HEAVY score ties (stable order: score descending, then prior * nc + class ascending, then level), 3 classes, 2 images, a level
    with fewer candidates than nms_pre and one with more, and boxes below min_bbox_size"""
    import ctypes
    from cartoonsegmentation_amd import _lib
    from cartoonsegmentation_amd._lib import check, f32, i32, ptr, stream_ptr
    L = _lib.load()
    rng = np.random.default_rng(8)
    nb, nc, G, nms_pre, thr = 2, 3, 7, 40, 0.25
    hw, strides = [(12, 10), (6, 5), (3, 3)], [8, 16, 32]
    clamp_w, clamp_h, sx, sy, min_box = 75.0, 90.0, np.float32(1.37), np.float32(0.81), 2.5
    cls = [(rng.integers(0, 12, (nb, h, w, nc)) / 12.0).astype(np.float32) * 0.9 + 0.05 for h, w in hw]        # 12 distinct values: ties everywhere
    reg = [rng.uniform(0, 3.0, (nb, h, w, 4)).astype(np.float32) for h, w in hw]
    reg[0][:, :3] *= 0.02                                                                                         # tiny boxes -> min_bbox_size filter
    kern = [rng.normal(0, 1, (nb, h, w, G)).astype(np.float32) for h, w in hw]
    d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()                                               # noqa: E731
    cls_d, reg_d, kern_d = [d(a) for a in cls], [d(a) for a in reg], [d(a) for a in kern]
    nl = 3
    vp = ctypes.c_void_p * nl
    cp_, rp_, kp_ = vp(*[t.data_ptr() for t in cls_d]), vp(*[t.data_ptr() for t in reg_d]), vp(*[t.data_ptr() for t in kern_d])
    level_hw = (ctypes.c_int * 6)(*[v for h, w in hw for v in (h, w)])
    st = (ctypes.c_int * 3)(*strides)
    lds3 = (ctypes.c_int * 9)(*[v for _ in range(3) for v in (nc, 4, G)])
    slots = L.csm_det_decode_slots(level_hw, i32(nl), i32(nc), i32(nms_pre))
    assert slots == 40 + 40 + 27
    K = slots
    scores = torch.empty((nb, K), device='cuda'); boxes = torch.empty((nb, K, 4), device='cuda')
    src = torch.empty((nb, K), dtype=torch.int32, device='cuda'); labels = torch.empty((nb, K), dtype=torch.int32, device='cuda')
    offs = torch.empty((nb, K), device='cuda')
    scratch = torch.empty(L.csm_det_decode_scratch_bytes(i32(nb), i32(slots)), dtype=torch.uint8, device='cuda')
    check(L.csm_det_decode(cp_, rp_, level_hw, st, lds3, i32(nl), i32(nb), i32(nc), f32(thr), i32(nms_pre), f32(clamp_w), f32(clamp_h),
                           f32(float(sx)), f32(float(sy)), f32(min_box), i32(K), ptr(scores), ptr(boxes), ptr(src), ptr(labels), ptr(offs),
                           ptr(scratch), stream_ptr()))
    prior0 = [0, 120, 150]
    for b in range(nb):
        sc_l, box_l, src_l, lab_l = [], [], [], []
        for l, ((h, w), s) in enumerate(zip(hw, strides)):
            flat = cls[l][b].reshape(-1)
            idx = np.nonzero(flat > np.float32(thr))[0]
            order = np.argsort(-flat[idx], kind='stable')[:nms_pre]
            idx = idx[order]
            p, lab = idx // nc, idx % nc
            px, py = ((p % w) * s).astype(np.float32), ((p // w) * s).astype(np.float32)
            dist = reg[l][b].reshape(-1, 4)[p] * np.float32(s)
            x1 = np.clip(px - dist[:, 0], 0, np.float32(clamp_w)) * sx; y1 = np.clip(py - dist[:, 1], 0, np.float32(clamp_h)) * sy
            x2 = np.clip(px + dist[:, 2], 0, np.float32(clamp_w)) * sx; y2 = np.clip(py + dist[:, 3], 0, np.float32(clamp_h)) * sy
            sc_l.append(flat[idx]); box_l.append(np.stack([x1, y1, x2, y2], 1).astype(np.float32)); src_l.append(prior0[l] + p); lab_l.append(lab)
        sc, bx, sr, lb = np.concatenate(sc_l), np.concatenate(box_l), np.concatenate(src_l), np.concatenate(lab_l)
        ok = ((bx[:, 2] - bx[:, 0]) > np.float32(min_box)) & ((bx[:, 3] - bx[:, 1]) > np.float32(min_box))
        assert 0 < ok.sum() < len(ok)
        sc, bx, sr, lb = sc[ok], bx[ok], sr[ok], lb[ok]
        order = np.argsort(-sc, kind='stable')
        sc, bx, sr, lb = sc[order], bx[order], sr[order], lb[order]
        n = len(sc)
        got_s = scores[b].cpu().numpy()
        assert np.array_equal(got_s[:n], sc) and (got_s[n:] == -1.0).all()
        assert np.array_equal(boxes[b, :n].cpu().numpy(), bx) and np.array_equal(src[b, :n].cpu().numpy(), sr)
        assert np.array_equal(labels[b, :n].cpu().numpy(), lb)
        assert np.array_equal(offs[b, :n].cpu().numpy(), (lb.astype(np.float32) * (np.float32(bx.max()) + np.float32(1))).astype(np.float32))
        # gather of an arbitrary kept list: priors and dynamic-conv parameters of the chosen candidates
        M = 6
        keep = torch.tensor([[0, 5, n - 1, 3, 1, 2]] * nb, dtype=torch.int32, device='cuda')
        ks, kb, kl = torch.empty((nb, M), device='cuda'), torch.empty((nb, M, 4), device='cuda'), torch.empty((nb, M), dtype=torch.int32, device='cuda')
        kp, kk = torch.empty((nb, M, 4), device='cuda'), torch.empty((nb, M, G), device='cuda')
        check(L.csm_det_gather(kp_, level_hw, st, lds3, i32(nl), i32(nb), i32(K), i32(M), i32(G), ptr(keep), ptr(scores), ptr(boxes), ptr(src),
                               ptr(labels), ptr(ks), ptr(kb), ptr(kl), ptr(kp), ptr(kk), stream_ptr()))
        for j, r in enumerate([0, 5, n - 1, 3, 1, 2]):
            g = int(sr[r]); l = 2 if g >= 150 else (1 if g >= 120 else 0); p = g - prior0[l]
            assert np.array_equal(kk[b, j].cpu().numpy(), kern[l][b].reshape(-1, G)[p])
            assert kp[b, j].cpu().numpy().tolist() == [float((p % hw[l][1]) * strides[l]), float((p // hw[l][1]) * strides[l]), float(strides[l]), float(strides[l])]
            assert float(ks[b, j]) == float(sc[r]) and np.array_equal(kb[b, j].cpu().numpy(), bx[r]) and int(kl[b, j]) == int(lb[r])


def test_detector_programs_of_different_sizes_do_not_share_a_mismatched_weight_image():
    """the packed weight image is part of the lowering (a layer's weights are Winograd panels or direct tiles depending on its map size):
    det 640 and det 1024 lower stride-32 layers differently (20 x 20 direct, 32 x 32 Winograd F(4x4)), so the second program must not run
    on the first one's image (round 6: a shared image made the det-1024 program read panels that were never packed -- memory fault).
    Both sizes against the oracle pipeline."""
    import os
    os.environ["CSM_SYNTHETIC_WEIGHTS"] = "1"
    from animeinsseg import AnimeInsSeg
    from cartoonsegmentation_amd import synth
    from cartoonsegmentation_amd.nets import build_rtmdet
    from cartoonsegmentation_amd.weights import SynthWeights
    from oracle import segment as oseg
    img = synth.image_u8(200, 264, 17)
    net = AnimeInsSeg('synthetic', default_det_size=640, refine_kwargs={'refine_method': 'none'})
    for S in (640, 1024, 640):
        inst = net.infer(img, pred_score_thr=0.3, max_instances=2, output_type='numpy', det_size=S)
        rp, cfg = build_rtmdet(SynthWeights('rtmdet.'), 1, S, S)
        cfg.max_per_img = 2
        d = oseg.detect(img, rp, cfg, S, 0.3)
        assert d['n'] == len(inst) and np.array_equal(d['bboxes'], inst.bboxes) and np.array_equal(d['scores'], inst.scores), S
    packings = {id(cp.weights) for _, cp in net._det_programs.values()}
    flags = {S: tuple(o['flags'] for o in rp_.prog.ops if o['kind'] == 1) for (S, _), (rp_, _) in net._det_programs.items()}
    assert (len(packings) == 2) == (flags[640] != flags[1024])
