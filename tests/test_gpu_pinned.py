"""GPU: the HIP path (through the C ABI / the drop-in classes) against the fixtures produced by EXECUTING the reference's own
text (tests/golden/make_golden_pinned.py): mask head (a5), mask up-sample / threshold (a6), refine glue (a7), depth adjustment
(a10), autozoom search (a16).  CPU twins (oracle vs the same fixtures): tests/test_oracle_pinned.py."""
import os

import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def _load(name):
    return np.load(os.path.join(GOLDEN, name + '.npz'))


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


@pytest.mark.parametrize("name", ["pin_maskhead_20x20", "pin_maskhead_12x28"])
def test_maskhead_logits_vs_reference_text(name):
    from cartoonsegmentation_amd import _lib
    from cartoonsegmentation_amd._lib import check, i32, ptr, stream_ptr
    d = _load(name)
    feat = dev(d['mask_feat'][0].transpose(1, 2, 0))                               # NHWC like the engine's activations
    h, w = feat.shape[:2]
    n = len(d['priors'])
    logits = torch.empty((n, h, w), device='cuda')
    ker, pri = dev(d['kernels']), dev(d['priors'])                                # kept alive across the launch
    check(_lib.load().csm_maskhead_logits(ptr(feat), i32(8), i32(h), i32(w), i32(8), i32(8), ptr(ker), ptr(pri),
                                          i32(n), i32(8), ptr(logits), stream_ptr()), "maskhead")
    ref = d['logits']
    assert np.abs(logits.cpu().numpy() - ref).max() <= 2e-5 * max(1.0, float(np.abs(ref).max()))


@pytest.mark.parametrize("name", ["pin_boxprompt_100x140", "pin_boxprompt_152x96"])
def test_segment_with_bboxes_vs_reference_text(name):
    """AnimeInsSeg.segment_with_bboxes (reference :339-393) fed with the fixture's detections: masks, boxes, scores"""
    from animeinsseg import AnimeInsSeg
    d = _load(name)
    H, W = int(d['H']), int(d['W'])
    net = AnimeInsSeg('synthetic', default_det_size=int(d['S']), refine_kwargs={'refine_method': 'none'})
    data = dict(n=len(d['boxes']), boxes=dev(d['boxes']), scores=dev(d['scores']), priors=dev(d['priors']), kernels=dev(d['kernels']),
                H=H, W=W)
    feat = dev(d['mask_feat'][0].transpose(1, 2, 0))
    inst = net.segment_with_bboxes(np.zeros((H, W, 3), np.uint8), d['query'], data, feat)
    masks = inst.masks.cpu().numpy()
    assert masks.shape == d['masks'].shape and (masks != d['masks']).mean() <= 2e-4
    assert np.array_equal(inst.bboxes.cpu().numpy(), d['out_bboxes'])
    assert np.array_equal(inst.scores.cpu().numpy(), d['out_scores'])


@pytest.mark.parametrize("name", ["pin_refine_90x74_T96", "pin_refine_64x64_T64"])
def test_refine_glue_vs_reference_text(name):
    """prepare_refine_batch (bit-exact), ISNet on that batch (vs the reference module's logits), sigmoid / crop / align_corners
    resize / threshold on the fixture's (re-centred) logits"""
    from cartoonsegmentation_amd import _lib
    from cartoonsegmentation_amd._lib import check, f32, i32, ptr, stream_ptr
    from cartoonsegmentation_amd.nets import build_isnet
    from cartoonsegmentation_amd.runtime import CompiledProgram
    from cartoonsegmentation_amd.weights import SynthWeights
    L = _lib.load()
    d = _load(name)
    img, T = d['img'], int(d['T'])
    H, W = img.shape[:2]
    n = d['masks_in'].shape[0]
    batch = torch.empty((n, 4, T, T), device='cuda')
    img_d, masks_d = dev(img), dev(d['masks_in'].astype(np.uint8))               # kept alive across the launch
    check(L.csm_refine_prepare_batch(ptr(img_d), ptr(masks_d), i32(n), i32(H), i32(W), i32(H), i32(W),
                                     i32(H), i32(W), i32(H), i32(W), i32(T), ptr(batch), stream_ptr()), "prepare")
    assert np.array_equal(batch.cpu().numpy(), d['batch'])
    raw = []
    for k0 in range(0, n, 4):                                                   # the reference's batches of <= 4
        b = min(4, n - k0)
        cp = CompiledProgram(build_isnet(SynthWeights('isnet.'), b, T, T), torch.device('cuda'))
        out = torch.empty((b, 1, T, T), device='cuda')
        cp.run(batch[k0:k0 + b].contiguous(), out)
        raw.append(out)
    raw = torch.cat(raw).cpu().numpy()
    assert np.abs(raw - d['logits_raw']).max() <= 2e-4 * float(np.abs(d['logits_raw']).max())
    logits = dev(((d['logits_raw'] - np.float32(d['centre'])) / np.float32(d['scale'])).astype(np.float32))
    out = torch.empty((n, H, W), dtype=torch.uint8, device='cuda')
    check(L.csm_refine_threshold(ptr(logits), i32(n), i32(T), i32(T), i32(H), i32(W), i32(H), i32(W), f32(0.3), ptr(out), stream_ptr()),
          "threshold")
    diff = out.cpu().numpy().astype(bool) != d['masks_out']
    assert diff.mean() <= 1e-4 and np.all(np.abs(d['probs'][diff] - 0.3) < 1e-5)


def test_depth_adjustment_vs_reference_text():
    from animeinsseg import AnimeInstances
    from cartoonsegmentation_amd.kenburns import depth_adjustment_animesseg
    d = _load("pin_depth_adjust")
    H, W = d['disp'].shape[2:]
    inst = AnimeInstances(dev(d['masks']), torch.zeros((4, 4), dtype=torch.int32, device='cuda'), torch.ones(4, device='cuda'))
    img = torch.zeros(1, 3, H, W, device='cuda')
    assert np.array_equal(depth_adjustment_animesseg(inst, dev(d['disp']), img, False).cpu().numpy(), d['adjusted'])
    assert np.array_equal(depth_adjustment_animesseg(inst, dev(d['disp']), img, True).cpu().numpy(), d['adjusted_median'])
    assert np.array_equal(depth_adjustment_animesseg(AnimeInstances(), dev(d['disp']), img, False).cpu().numpy(), d['adjusted_empty'])
    got = depth_adjustment_animesseg(inst, dev(d['disp_small']), img, False).cpu().numpy()       # bilinear round trip: fp32 rounding
    assert np.abs(got - d['adjusted_resized']).max() <= 1e-4 * float(d['adjusted_resized'].max())


def test_batched_autozoom_vs_reference_text_and_oracle():
    """csm_autozoom_coverage: coverage counts of every candidate == the oracle's per-candidate render (Jacobi degrid), and the
    chosen target is, by the REFERENCE's own counts, within one pixel of the reference's best (its in-place degrid is a race)"""
    from cartoonsegmentation_amd import ops
    from oracle import kenburns as okb
    d = _load("pin_autozoom_96x128")
    H, W = int(d['H']), int(d['W'])
    dr = d['depthrange']
    common = {'objDepthrange': (float(dr[0]), float(dr[1]), (int(dr[2]), int(dr[3])), (0, 0)), 'intWidth': W, 'intHeight': H,
              'fltFocal': float(d['focal']), 'fltBaseline': float(d['baseline']), 'tenRawPoints': dev(d['pts'])}
    objFrom = {'fltCenterU': W / 2.0, 'fltCenterV': H / 2.0, 'intCropWidth': int(np.floor(0.97 * W)), 'intCropHeight': int(np.floor(0.97 * H))}
    settings = {'fltShift': float(d['shift']), 'fltZoom': 1.25, 'objFrom': objFrom}
    kc = dict(depthrange=common['objDepthrange'][:3], pts=d['pts'])
    cj = []
    to_j, _ = okb.autozoom_target(kc, d['rgb'], W, H, common['fltFocal'], common['fltBaseline'], shift=float(d['shift']), degrid_mode=1,
                                  counts_out=cj)
    for path, chunk in (('bands', None), ('planes', None), ('planes', 5)):        # LDS band path; plane path, default and ragged chunk
        os.environ['CSM_AUTOZOOM_PATH'] = path
        if chunk:
            os.environ['CSM_AUTOZOOM_CHUNK'] = str(chunk)
        try:
            to, cands, counts = ops.process_autozoom(settings, common, return_counts=True)
        finally:
            os.environ.pop('CSM_AUTOZOOM_CHUNK', None); os.environ.pop('CSM_AUTOZOOM_PATH', None)
        assert len(counts) == len(d['counts']) and counts == [int(c) for c in cj], (path, chunk)
        assert to == to_j
    c_ref = d['counts']
    assert c_ref[int(np.argmax(counts))] >= c_ref.max() - 0.001 * H * W
    assert np.abs(np.asarray(counts) - c_ref).max() <= 0.005 * H * W
    # band path on awkward clouds == plane path: N > P with scattered extra points, odd sizes, a pile-up that overflows the band
    # segments (the flag sends the call to the plane path), and an empty cloud
    g = np.random.default_rng(11)
    for (Hh, Ww, n, pile) in ((45, 70, 9000, False), (250, 333, 120000, False), (96, 128, 40000, True), (64, 64, 0, False)):
        z = g.uniform(30, 80, n).astype(np.float32)
        spread = 0.02 if pile else 1.2
        xy = g.uniform(-spread, spread, (2, n)).astype(np.float32) * z * np.array([[Ww / 70.0], [Hh / 70.0]], np.float32)
        pts = torch.from_numpy(np.stack([xy[0], xy[1], z])[None].astype(np.float32)).cuda()
        shifts = [(float(a), float(b), -3.5) for b in (-2.0, 0.0, 1.5) for a in (-4.0, -1.0, 0.5, 2.0, 3.0)]
        res = {}
        for path in ('bands', 'planes'):
            os.environ['CSM_AUTOZOOM_PATH'] = path
            try:
                res[path] = ops.autozoom_coverage(pts, shifts, Ww, Hh, 35.0, 40.0)
            finally:
                os.environ.pop('CSM_AUTOZOOM_PATH', None)
        assert res['bands'] == res['planes'], (Hh, Ww, n, pile)
        assert n == 0 or max(res['bands']) > 0
    # z-buffer entries NEAR ZERO: depths around focal * baseline / 1e6 make fltError = 1e6 - fb / z land within a few
    # units of 0 in steps of ~0.08, so the degrid's `c >= a + 1.0` tests run on small, mixed-sign operands where the band path's
    # one-threshold-per-pixel form must still decide like the reference expression (plane path)
    n = 30000
    z0 = np.float32(35.0 * 40.0 / 1e6)
    z = (z0 + g.integers(-40, 41, n).astype(np.float32) * np.spacing(z0)).astype(np.float32)
    xy = g.uniform(-1.0, 1.0, (2, n)).astype(np.float32) * z * np.array([[96 / 70.0], [64 / 70.0]], np.float32)
    pts = torch.from_numpy(np.stack([xy[0], xy[1], z])[None].astype(np.float32)).cuda()
    shifts = [(float(a) * 1e-5, float(b) * 1e-5, 0.0) for b in (-2.0, 0.0, 1.5) for a in (-4.0, -1.0, 0.5, 2.0)]
    res = {}
    for path in ('bands', 'planes'):
        os.environ['CSM_AUTOZOOM_PATH'] = path
        try:
            res[path] = ops.autozoom_coverage(pts, shifts, 96, 64, 35.0, 40.0)
        finally:
            os.environ.pop('CSM_AUTOZOOM_PATH', None)
    assert res['bands'] == res['planes'] and max(res['bands']) > 0
    # BASELINE's full size: the 1024 x 1024 frame cloud (N = P), a 6 x 6 sub-grid of the search's shifts, both paths, equal counts
    from cartoonsegmentation_amd import synth
    S = 1024
    sc = synth.warp_scene(S, S, 7)
    disp = torch.from_numpy(sc['disp']).cuda()
    disp = disp / disp.max() * sc['baseline']
    _, _, pts, _ = ops.disparity_to_points(disp, sc['focal'], sc['baseline'])
    pts = pts.view(1, 3, -1).contiguous()
    lin = np.linspace(-9.0, 9.0, 6)
    shifts = [(float(a), float(b), -6.5) for b in lin for a in lin]
    res = {}
    for path in ('bands', 'planes'):
        os.environ['CSM_AUTOZOOM_PATH'] = path
        try:
            res[path] = ops.autozoom_coverage(pts, shifts, S, S, sc['focal'], sc['baseline'])
        finally:
            os.environ.pop('CSM_AUTOZOOM_PATH', None)
    assert res['bands'] == res['planes'] and min(res['bands']) > 0.9 * S * S


def test_frame_scaledown_and_path_input(tmp_path):
    """generate_kenburns_config on an image larger than max_size (reference :917 scaledown_maxsize) given as a FILE PATH (:909):
    the frame is the cv2-INTER_LINEAR restatement of the oracle, instances are resized to it, the rest of the pipeline runs"""
    os.environ["CSM_SYNTHETIC_WEIGHTS"] = "1"
    import ctypes
    from PIL import Image
    from anime_3dkenburns import KenBurnsConfig, KenBurnsPipeline
    from cartoonsegmentation_amd import synth
    from oracle import segment as oseg
    H, W = 640, 960
    img = synth.image_u8(H, W, 21)
    p = str(tmp_path / "in.png")
    Image.fromarray(img[:, :, ::-1]).save(p)
    cfg = KenBurnsConfig(det_ckpt='synthetic', depth_est='leres', depth_est_size=96, max_size=600, refine_crf=False, focal=300.0,
                         num_frame=2, mask_refine_kwargs={'refine_method': 'none'})
    pipe = KenBurnsPipeline(cfg)
    pipe.animeinsseg.set_detect_size(96)
    kc = pipe.generate_kenburns_config(p)
    h, w = 400, 600
    assert (kc.int_height, kc.int_width) == (h, w) and kc['tenRawImage'].shape == (1, 3, h, w)
    small = np.empty((h, w, 3), np.uint8)
    oseg.lib().orc_resize_u8_linear(oseg._p(img), ctypes.c_int(H), ctypes.c_int(W), ctypes.c_int(3), ctypes.c_int(h), ctypes.c_int(w),
                                    oseg._p(small))
    got = (kc['tenRawImage'][0].permute(1, 2, 0) * 255.0).round().to(torch.uint8).cpu().numpy()
    assert np.array_equal(got, small)
    assert isinstance(kc.original_img_nparray, np.ndarray) and kc.original_img_nparray.shape == (h, w, 3)
    if not kc.instances.is_empty:
        assert kc.instances.masks.shape[1:] == (h, w)
    frames = pipe.autozoom(kc, inpaint=False)
    assert len(frames) == 2 and frames[0].shape == (h, w, 3)


def test_run_kenburns_call_sequence_on_1920x1080_with_the_shipped_yaml(tmp_path, monkeypatch):
    """run_kenburns.py:19-42 verbatim on a 1920 x 1080 image file with configs/3dkenburns.yaml's keys (max_size 1024 < image:
    the frame is scaled down, BASELINE configs[0]'s example has this shape): KenBurnsPipeline(cfg path) ->
    generate_kenburns_config(img) -> autozoom -> frames.  Only num_frame is cut (75 -> 3) and the checkpoints are closed-form."""
    import yaml
    from PIL import Image
    monkeypatch.setenv("CSM_SYNTHETIC_WEIGHTS", "1")
    monkeypatch.setenv("CSM_AUTOTUNE", "0")                      # built-in tile rule: this test is about function, not speed
    from anime_3dkenburns import KenBurnsPipeline
    from cartoonsegmentation_amd import synth
    from utils.io_utils import imread
    cfg = {'inpaint_type': 'default', 'detector': 'animeinsseg', 'num_frame': 3, 'playback': True, 'dof_speed': 50, 'depth_field': True,
           'max_size': 1024, 'ldm_inpaint_size': 1024, 'sd_img2img_url': 'http://127.0.0.1:7860/sdapi/v1/img2img',
           'mask_refine_kwargs': {'refine_method': 'refinenet_isnet', 'refine_size': 720}, 'depth_est': 'leres', 'depth_est_size': 640,
           'det_ckpt': 'synthetic', 'det_size': 640, 'pred_score_thr': 0.3, 'refine_crf': False, 'depth_factor': 1}
    cfgp = tmp_path / "3dkenburns.yaml"
    cfgp.write_text(yaml.safe_dump(cfg))
    imgp = str(tmp_path / "kenburns_lion.png")
    Image.fromarray(synth.image_u8(1080, 1920, 3)[:, :, ::-1]).save(imgp)
    kpipe = KenBurnsPipeline(str(cfgp))
    kpipe.max_instances = 2                                      # closed-form weights score every prior alike: cap like infer(max_instances=)
    img = imread(imgp)                                           # mmcv.imread stand-in
    kcfg = kpipe.generate_kenburns_config(img, verbose=False, savep=str(tmp_path / "out.mp4"))
    assert (kcfg.int_height, kcfg.int_width) == (576, 1024) and kcfg['tenRawPoints'].shape == (1, 3, 576 * 1024)
    assert kcfg.instances.is_empty or kcfg.instances.masks.shape[1:] == (576, 1024)
    frames = kpipe.autozoom(kcfg, verbose=False)
    assert len(frames) == 3 and all(f.shape == (576, 1024, 3) and f.dtype == np.uint8 for f in frames)
    assert kcfg.playback is True
