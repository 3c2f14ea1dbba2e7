"""AnimeInsSeg -- host-side mirror of animeinsseg/__init__.py::AnimeInsSeg (reference :185-708) on libcsm355.

Same constructor / infer() signature and result type (AnimeInstances), so run_segmentation.ipynb drops in.
Flow per image (reference :464-504, :447-462, :638-665), everything on the MI355X, masks never leave HBM:
  uint8 image --csm_det_preprocess--> RTMDet-Ins layer program --csm_det_decode--> csm_nms --csm_det_gather-->
  csm_maskhead_logits --> csm_mask_resize_threshold --> [ISNet refine: csm_refine_prepare_batch --> ISNet layer
  program --> csm_refine_threshold] --> AnimeInstances(masks bool [n,H,W], bboxes xywh int32, scores).
The reference makes 5 host<->device crossings per image; here the only syncs are the data-dependent counts.

Checkpoints: a real `rtmdetl_e60.ckpt` (mmdet state_dict + cfg text) and `refine_last.ckpt` import through
StateDictWeights; `ckpt="synthetic"` uses closed-form weights (no checkpoint exists in this container).
mmdet's pieces are restated from mmdet 3.3.0 (not vendored in the reference => parity unpinned).
"""
import ctypes
import math
import os
from typing import List, Union

import numpy as np
import torch

from . import _lib
from ._lib import check, f32, i32, ptr, stream_ptr
from .anime_instances import AnimeInstances
from .nets import RTMDetConfig, build_isnet, build_rtmdet
from .runtime import CompiledProgram
from .weights import StateDictWeights, SynthWeights

VALID_REFINEMETHODS = {'refinenet_isnet', 'none'}


def rescale_size(h, w, scale):
    """mmcv.image.rescale_size for scale=(S,S): factor = min(long/max, short/min); int(x*f+0.5)"""
    long_e, short_e = max(scale), min(scale)
    f = min(long_e / max(h, w), short_e / min(h, w))
    return int(h * float(f) + 0.5), int(w * float(f) + 0.5)


def scaledown_size(h, w, max_size):
    """utils/io_utils.py:254-274 scaledown_maxsize size rule (never upsamples)"""
    r = max_size / max(h, w)
    if r < 1:
        if h > w:
            h, w = max_size, max(1, int(round(w * r)))
        else:
            w, h = max_size, max(1, int(round(h * r)))
    return h, w


def _parse_mm_cfg(text):
    """evaluate the mmengine config text stored in the checkpoint (animeinsseg/__init__.py:196-197) without
    mmengine: it is plain Python made of literals and dict(...) calls."""
    env = {'__builtins__': {}, 'dict': dict, 'list': list, 'tuple': tuple, 'True': True, 'False': False, 'None': None}
    loc = {}
    exec(text.replace('file_client_args', 'backend_args'), env, loc)   # noqa: S102 (checkpoint is as trusted as torch.load)
    return loc


def config_from_ckpt_cfg(cfg_text):
    loc = _parse_mm_cfg(cfg_text)
    m = loc['model']
    c = RTMDetConfig()
    bb, head = m.get('backbone', {}), m.get('bbox_head', {})
    c.deepen_factor, c.widen_factor = bb.get('deepen_factor', 1.0), bb.get('widen_factor', 1.0)
    c.expand_ratio = bb.get('expand_ratio', c.expand_ratio)
    c.num_classes = head.get('num_classes', 1)
    c.feat_channels = head.get('feat_channels', 256)
    c.stacked_convs = head.get('stacked_convs', 2)
    c.share_conv = head.get('share_conv', True)
    c.num_prototypes = head.get('num_prototypes', c.num_prototypes)
    c.dyconv_channels = head.get('dyconv_channels', c.dyconv_channels)
    c.num_dyconvs = head.get('num_dyconvs', c.num_dyconvs)
    ag = head.get('anchor_generator', {})
    c.strides = tuple(ag.get('strides', c.strides))
    # BatchNorm eps: a norm_cfg without `eps` means torch's default 1e-5; no norm_cfg at all means the mmdet class default
    # (CSPNeXt / CSPNeXtPAFPN: dict(type='BN', momentum=0.03, eps=0.001); RTMDetHead: dict(type='BN') -> 1e-5)
    c.bn_eps_backbone = bb['norm_cfg'].get('eps', 1e-5) if 'norm_cfg' in bb else 1e-3
    nk = m.get('neck', {})
    c.bn_eps_neck = nk['norm_cfg'].get('eps', 1e-5) if 'norm_cfg' in nk else 1e-3
    c.bn_eps_head = head['norm_cfg'].get('eps', 1e-5) if 'norm_cfg' in head else 1e-5
    dp = m.get('data_preprocessor', {})
    c.mean, c.std = tuple(dp.get('mean', c.mean)), tuple(dp.get('std', c.std))
    t = m.get('test_cfg', {})
    c.nms_pre, c.score_thr = t.get('nms_pre', c.nms_pre), t.get('score_thr', c.score_thr)
    c.nms_iou = t.get('nms', {}).get('iou_threshold', c.nms_iou)
    c.max_per_img, c.mask_thr_binary = t.get('max_per_img', c.max_per_img), t.get('mask_thr_binary', c.mask_thr_binary)
    c.min_bbox_size = t.get('min_bbox_size', c.min_bbox_size)
    return c


class AnimeInsSeg:
    def __init__(self, ckpt: str, default_det_size: int = 640, device: str = None,
                 refine_kwargs: dict = {'refine_method': 'refinenet_isnet'},
                 tagger_path: str = 'models/wd-v1-4-swinv2-tagger-v2/model.onnx', mask_thr=0.3):
        if device is None:
            device = 'cuda'
        if not torch.cuda.is_available():
            raise _lib.CsmError("AnimeInsSeg needs an MI355X: libcsm355 has no CPU path")
        _lib.load()
        self.device = torch.device(device if device != 'cuda' else 'cuda:%d' % torch.cuda.current_device())
        self.ckpt, self.default_det_size, self.mask_thr = ckpt, default_det_size, mask_thr
        self.tagger, self.tagger_path = None, tagger_path
        if ckpt is None or str(ckpt).startswith('synthetic'):
            self.cfg, self._det_ws = RTMDetConfig(), SynthWeights('rtmdet.')
        else:
            blob = torch.load(ckpt, map_location='cpu', weights_only=False)
            self.cfg = config_from_ckpt_cfg(blob['meta']['cfg'])
            self._det_ws = StateDictWeights(blob['state_dict'])
        self._det_programs, self._det_weights = {}, {}          # (weights: packed images on the device by packing signature)
        self._refine_programs, self._refine_weights, self._refine_ws = {}, {}, None
        self.refine_method = None
        self.refine_batch = int(os.environ.get('CSM_REFINE_BATCH', '16'))   # instances per ISNet run when frames are batched
        self.det_batch = max(1, int(os.environ.get('CSM_DET_BATCH', '16')))  # frames per detector run (longer lists are chunked)
        self.set_refine_method(**(refine_kwargs or {'refine_method': 'none'}))

    # ---- configuration (reference :395-399, :623-636, :704-708) --------------------------------
    def set_detect_size(self, det_size: Union[int, tuple]):
        self.default_det_size = det_size if isinstance(det_size, int) else max(det_size)

    def set_refine_method(self, refine_method: str = 'none', refine_size: int = 720, refinenet_ckpt: str = None, **kw):
        if refine_method == 'animeseg':
            raise NotImplementedError("refine_method 'animeseg' is out of the hot-path scope (SURVEY 2.1)")
        if refine_method not in VALID_REFINEMETHODS:
            raise NotImplementedError('Invalid refine method: %s' % refine_method)
        self.refine_method, self.refine_size = refine_method, refine_size
        if refine_method == 'refinenet_isnet' and self._refine_ws is None:
            refinenet_ckpt = refinenet_ckpt or 'models/AnimeInstanceSegmentation/refine_last.ckpt'   # utils/constants.py:80
            synthetic = str(self.ckpt).startswith('synthetic') or os.environ.get('CSM_SYNTHETIC_WEIGHTS', '0') == '1'
            if not os.path.exists(refinenet_ckpt) and not synthetic:
                raise FileNotFoundError(refinenet_ckpt)
            if os.path.exists(refinenet_ckpt):
                sd = torch.load(refinenet_ckpt, map_location='cpu', weights_only=False)
                sd = sd.get('state_dict', sd)
                self._refine_ws = StateDictWeights({k.replace('net.', '', 1) if k.startswith('net.') else k: v for k, v in sd.items()})
            else:
                self._refine_ws = SynthWeights('isnet.')

    def set_mask_threshold(self, mask_thr: float):
        self.cfg.mask_thr_binary = mask_thr

    def set_max_instance(self, num_ins):
        self.cfg.max_per_img = num_ins

    # ---- compiled programs ----------------------------------------------------------------------
    def _detector(self, S, n=1):
        if (S, n) not in self._det_programs:
            rp, _ = build_rtmdet(self._det_ws, n, S, S, self.cfg)
            cp = CompiledProgram(rp.prog, self.device, shared=self._det_weights)     # (shared between shapes only where the packing is the same)
            self._det_programs[(S, n)] = (rp, cp)
        return self._det_programs[(S, n)]

    def _refiner(self, n, T):
        if (n, T) not in self._refine_programs:
            prog = build_isnet(self._refine_ws, n, T, T)
            cp = CompiledProgram(prog, self.device, shared=self._refine_weights)
            self._refine_programs[(n, T)] = cp
        return self._refine_programs[(n, T)]

    # ---- public entry (reference :401-445) --------------------------------------------------------
    def infer(self, imgs, pred_score_thr: float = 0.3, refine_kwargs: dict = None, output_type: str = "tensor",
              det_size: int = None, save_dir: str = '', save_visualization: bool = False, save_annotation: str = '',
              infer_tags: bool = False, obj_id_start: int = -1, img_id_start: int = -1, verbose: bool = False,
              infer_grey: bool = False, save_mask_only: bool = False, val_dir=None, max_instances: int = 100, **kw):
        if det_size is not None:
            self.set_detect_size(det_size)
        if refine_kwargs is not None:
            self.set_refine_method(**refine_kwargs)
        self.set_max_instance(max_instances)
        if save_annotation or save_visualization or infer_tags:
            raise NotImplementedError("annotation export / tagging are outside the hot-path scope (SURVEY 2.1)")
        assert output_type in {'tensor', 'numpy'}
        return_list = isinstance(imgs, list)
        if isinstance(imgs, str):                                 # prepare_data_pipeline, reference :667-693: a directory or one file
            if os.path.isdir(imgs):
                from utils.io_utils import find_all_imgs
                imgs, return_list = find_all_imgs(imgs, abs_path=True), True
            elif imgs.endswith('.txt') or imgs.endswith('.json'):
                raise NotImplementedError("image lists / COCO files belong to the annotation tooling (SURVEY 2.1)")
        imgs = imgs if return_list else [imgs]
        if any(isinstance(im, str) for im in imgs):               # single_image_preprocess, reference :62-64: mmcv.imread(path)
            from utils.io_utils import imread
            imgs = [imread(im) if isinstance(im, str) else im for im in imgs]
        same = len(imgs) > 1 and all(tuple(im.shape) == tuple(imgs[0].shape) for im in imgs)
        if same:            # equally sized frames: one batched detector run + refine batches shared across frames
            insts = [self._instances_from(d, pred_score_thr) for d in self.detect_raw_batch(imgs)]
            if self.refine_method == 'refinenet_isnet':
                self._refine_many(list(zip(insts, imgs)), self.refine_size)
        else:
            insts = []
            for img in imgs:
                inst = self._det_forward(img, pred_score_thr)
                if self.refine_method == 'refinenet_isnet':
                    self._postprocess_refine(inst, img, refine_size=self.refine_size)
                insts.append(inst)
        if output_type == 'numpy':
            for inst in insts:
                inst.to_numpy()
        return insts if return_list else insts[0]

    # ---- detector forward + post-process (reference :447-462 + mmdet predict_by_feat) ---------------
    def _upload(self, img):
        if isinstance(img, torch.Tensor):
            t = img.to(self.device)
        else:
            t = torch.from_numpy(np.ascontiguousarray(img)).to(self.device)
        assert t.dtype == torch.uint8 and t.dim() == 3 and t.shape[2] == 3, "expected uint8 HxWx3 (BGR)"
        return t.contiguous()

    def detect_raw(self, img):
        """returns dict with kept boxes/scores/kernels/priors (score-sorted, after NMS) + scale info"""
        return self.detect_raw_batch([img])[0]

    def detect_raw_batch(self, imgs):
        """MI355X addition: frames of equal size run through ONE detector program (batch n) -- the small feature maps
        of the head/neck then fill the chip.  Every op is per-sample; the only batch-size dependence is the host's split-K
        choice (csm_op.ksplit follows the total pixel count), i.e. fp32 summation grouping at the ulp level."""
        L, cfg = _lib.load(), self.cfg
        imgs_d = [self._upload(im) for im in imgs]
        H, W = int(imgs_d[0].shape[0]), int(imgs_d[0].shape[1])
        assert all(tuple(t.shape) == (H, W, 3) for t in imgs_d), "detect_raw_batch needs equally sized images"
        if len(imgs_d) > self.det_batch:
            # bounded workspace and a bounded set of compiled programs: full chunks of det_batch frames plus one remainder
            # program.  The prototype maps of earlier chunks are copied out (the next chunk overwrites the workspace).
            outs = []
            for c0 in range(0, len(imgs_d), self.det_batch):
                part = self.detect_raw_batch(imgs_d[c0:c0 + self.det_batch])
                if c0 + self.det_batch < len(imgs_d):
                    for d in part:
                        if d['n']:
                            d['mask_feat'] = d['cp'].view(d['rp'].mask_feat)[d['bi']][..., :cfg.num_prototypes].clone()
                outs += part
            return outs
        nb = len(imgs_d)
        S = self.default_det_size
        rh, rw = rescale_size(H, W, (S, S))
        w_scale, h_scale = rw / W, rh / H
        rp, cp = self._detector(S, nb)
        x = torch.empty((nb, 3, S, S), dtype=torch.float32, device=self.device)
        mean = (ctypes.c_float * 3)(*cfg.mean); std = (ctypes.c_float * 3)(*cfg.std)
        for bi, img_d in enumerate(imgs_d):
            check(L.csm_det_preprocess(ptr(img_d), i32(H), i32(W), i32(rh), i32(rw), i32(S), i32(S), mean, std,
                                       f32(cfg.pad_value), ptr(x[bi]), stream_ptr()), "det_preprocess")
        cp.run(x)
        return self._decode_batch(rp, cp, nb, H, W, S, rh, rw, w_scale, h_scale)

    def _decode_batch(self, rp, cp, nb, H, W, S, rh, rw, w_scale, h_scale):
        """mmdet RTMDetInsHead.predict_by_feat / _bbox_mask_post_process up to and including NMS, for all images of the batch, on the
        device with FIXED shapes and one host sync (csm_det_decode -> csm_nms -> csm_det_gather, csrc/detdecode.hip): instead of
        filtering (`scores > score_thr`, data-dependent sizes) invalid candidates carry score -1, which the stable descending sorts
        push behind every valid one; they cannot suppress a valid box in the greedy NMS (a box only suppresses lower-ranked ones) and
        are dropped at the end.  Same kept set and order as the filter-then-topk of the reference."""
        L, cfg, dev = _lib.load(), self.cfg, self.device
        nc, nl, G, M = cfg.num_classes, len(cfg.strides), cfg.num_gen_params, cfg.max_per_img
        if cfg.score_thr < 0:
            raise _lib.CsmError("score_thr must be >= 0 (scores are sigmoids; the device top-k orders their bit patterns)")

        def base(v):                                       # device address of the first element of an NHWC view in the workspace
            return cp.workspace.data_ptr() + 4 * (v.buf.offset + v.coff)
        vp = ctypes.c_void_p * nl
        cls_p, reg_p, ker_p = (vp(*[base(t[l]) for l in range(nl)]) for t in (rp.cls, rp.reg, rp.kern))
        level_hw = (ctypes.c_int * (2 * nl))(*[v for l in range(nl) for v in (rp.cls[l].h, rp.cls[l].w)])
        strides = (ctypes.c_int * nl)(*[int(s) for s in cfg.strides])
        lds3 = (ctypes.c_int * (3 * nl))(*[v for l in range(nl) for v in (rp.cls[l].buf.c, rp.reg[l].buf.c, rp.kern[l].buf.c)])
        slots = L.csm_det_decode_slots(level_hw, i32(nl), i32(nc), i32(cfg.nms_pre))
        if slots < 0:
            raise _lib.CsmError("detector decode: nms_pre <= 1024 and at most 4096 candidates per image are supported")
        K = min(slots, 4096)
        scores = torch.empty((nb, K), dtype=torch.float32, device=dev)
        boxes = torch.empty((nb, K, 4), dtype=torch.float32, device=dev)
        src = torch.empty((nb, K), dtype=torch.int32, device=dev)
        labels = torch.empty((nb, K), dtype=torch.int32, device=dev)
        offs = torch.empty((nb, K), dtype=torch.float32, device=dev) if nc > 1 else None
        scratch = torch.empty(L.csm_det_decode_scratch_bytes(i32(nb), i32(slots)), dtype=torch.uint8, device=dev)
        sfx, sfy = float(np.float32(1 / w_scale)), float(np.float32(1 / h_scale))                                      # rescale=True
        check(L.csm_det_decode(cls_p, reg_p, level_hw, strides, lds3, i32(nl), i32(nb), i32(nc), f32(cfg.score_thr), i32(cfg.nms_pre),
                               f32(rw), f32(rh), f32(sfx), f32(sfy), f32(cfg.min_bbox_size), i32(K), ptr(scores), ptr(boxes), ptr(src),
                               ptr(labels), ptr(offs), ptr(scratch), stream_ptr()), "det_decode")
        keep = torch.zeros((nb, M), dtype=torch.int32, device=dev)
        nk = torch.zeros(nb, dtype=torch.int32, device=dev)
        nscr = torch.empty(L.csm_nms_scratch_bytes(i32(K)), dtype=torch.uint8, device=dev)
        for bi in range(nb):
            check(L.csm_nms(ptr(boxes[bi]), ptr(None if offs is None else offs[bi]), i32(K), f32(cfg.nms_iou), i32(M),
                            ptr(keep[bi]), ptr(nk[bi:bi + 1]), ptr(nscr), stream_ptr()), "nms")
        k_scores = torch.empty((nb, M), dtype=torch.float32, device=dev)
        k_boxes = torch.empty((nb, M, 4), dtype=torch.float32, device=dev)
        k_labels = torch.empty((nb, M), dtype=torch.int32, device=dev)
        k_priors = torch.empty((nb, M, 4), dtype=torch.float32, device=dev)
        k_kernels = torch.empty((nb, M, G), dtype=torch.float32, device=dev)
        check(L.csm_det_gather(ker_p, level_hw, strides, lds3, i32(nl), i32(nb), i32(K), i32(M), i32(G), ptr(keep), ptr(scores), ptr(boxes),
                               ptr(src), ptr(labels), ptr(k_scores), ptr(k_boxes), ptr(k_labels), ptr(k_priors), ptr(k_kernels),
                               stream_ptr()), "det_gather")
        nk_h, ks_h = nk.tolist(), k_scores.tolist()                                # the one host sync of the decode
        outs = []
        for bi in range(nb):
            n = sum(1 for j in range(nk_h[bi]) if ks_h[bi][j] > cfg.score_thr)    # valid ones come first in keep[]
            out = dict(H=H, W=W, S=S, rh=rh, rw=rw, w_scale=w_scale, h_scale=h_scale, rp=rp, cp=cp, bi=bi, n=n)
            if n:
                out.update(boxes=k_boxes[bi, :n], scores=k_scores[bi, :n], priors=k_priors[bi, :n], kernels=k_kernels[bi, :n],
                           labels=k_labels[bi, :n], scores_host=ks_h[bi][:n])
            outs.append(out)
        return outs

    def _masks_from(self, d, sel=None):
        """dynamic-conv mask head + resize + sigmoid + threshold -> uint8 [n,H,W] on device"""
        L, cfg = _lib.load(), self.cfg
        pri, ker = (d['priors'], d['kernels']) if sel is None else (d['priors'][sel].contiguous(), d['kernels'][sel].contiguous())
        n = int(pri.shape[0])
        mf, ld, h, w = self._mask_feat_of(d)
        logits = torch.empty((n, h, w), dtype=torch.float32, device=self.device)
        check(L.csm_maskhead_logits(ptr(mf), i32(ld), i32(h), i32(w), i32(cfg.num_prototypes), i32(cfg.dyconv_channels),
                                    ptr(ker), ptr(pri), i32(n), i32(cfg.strides[0]), ptr(logits), stream_ptr()), "maskhead")
        up = cfg.strides[0]
        # mmdet quirk kept: scale_factor = [1/w_scale, 1/h_scale] is applied as (height, width)
        rh2 = math.ceil(h * up * (1 / d['w_scale'])); rw2 = math.ceil(w * up * (1 / d['h_scale']))
        # `[..., :ori_h, :ori_w]` is a slice: for ~20 % of image shapes ceil(S / scale) is 1-2 px short of the original size and
        # mmdet returns masks that much smaller (the ISNet refine brings them back to (H, W); AnimeInstances.resize handles the rest)
        oh, ow = min(rh2, d['H']), min(rw2, d['W'])
        masks = torch.empty((n, oh, ow), dtype=torch.uint8, device=self.device)
        check(L.csm_mask_resize_threshold(ptr(logits), i32(n), i32(h), i32(w), i32(up), i32(rh2), i32(rw2), i32(oh),
                                          i32(ow), f32(cfg.mask_thr_binary), ptr(masks), stream_ptr()), "mask_resize")
        return masks

    @staticmethod
    def _mask_feat_of(d, mask_feat=None):
        """(tensor, channel pitch, h, w) of the prototype map a detection dict refers to: an owned NHWC copy when one was taken
        (infer_embeddings / chunked batches), else the live view inside the detector workspace (valid until the next run)"""
        mf = mask_feat if mask_feat is not None else d.get('mask_feat')
        if mf is not None:
            mf = mf.contiguous()
            return mf, int(mf.shape[2]), int(mf.shape[0]), int(mf.shape[1])
        v = d['rp'].mask_feat
        b = v.buf
        return d['cp'].workspace[b.offset + d.get('bi', 0) * v.h * v.w * b.c:], b.c, v.h, v.w

    def _det_forward(self, img, pred_score_thr: float = 0.3) -> AnimeInstances:
        return self._instances_from(self.detect_raw(img), pred_score_thr)

    def _instances_from(self, d, pred_score_thr: float = 0.3) -> AnimeInstances:
        if d['n'] == 0:
            return AnimeInstances()
        hs = d.get('scores_host')
        if hs is not None:                                                         # reference :452, decided from the synced copy
            idx = [j for j, v in enumerate(hs) if np.float32(v) > np.float32(pred_score_thr)]
            if not idx:
                return AnimeInstances()
            sel = torch.tensor(idx, dtype=torch.long, device=self.device)
        else:
            sel = (d['scores'] > pred_score_thr).nonzero()[:, 0]
            if sel.numel() < 1:
                return AnimeInstances()
        masks = self._masks_from(d, sel).bool()
        bboxes = d['boxes'][sel].to(torch.int32)                                   # :458-459 xyxy -> xywh (truncation)
        bboxes[:, 2:] -= bboxes[:, :2]
        return AnimeInstances(masks, bboxes, d['scores'][sel])

    # ---- Web-UI helpers (reference :241-393): detector embeddings + box-prompted masks ---------------------
    def infer_embeddings(self, imgs, det_size=None):
        """reference :241-337: returns (img, instance_data, mask_feat) with the NMS'ed detections *before* mask
        generation; instance_data is a dict with bboxes [n,4] xyxy, scores, priors, kernels (+ internal handles)."""
        if det_size is not None:
            self.set_detect_size(det_size)
        img = imgs[0] if isinstance(imgs, list) else imgs
        d = self.detect_raw(img)
        if d['n'] == 0:
            d.update(boxes=torch.zeros((0, 4), device=self.device), scores=torch.zeros(0, device=self.device),
                     priors=torch.zeros((0, 4), device=self.device), kernels=torch.zeros((0, self.cfg.num_gen_params), device=self.device))
        d['bboxes'] = d['boxes']
        # an OWNED copy (the reference returns an owned tensor, animeinsseg/__init__.py:339-360): the detector workspace is
        # overwritten by the next infer()/detect_raw() of any image
        mask_feat = d['cp'].view(d['rp'].mask_feat)[d.get('bi', 0)][..., :self.cfg.num_prototypes].clone()
        d['mask_feat'] = mask_feat
        return img, d, mask_feat

    def segment_with_bboxes(self, img, bboxes, instance_data, mask_feat=None):
        """reference :339-393: for every query box (xyxy) pick the detection with the highest IoU, build its mask
        (x8 bilinear -> resize to [long_side, long_side] -> crop -> sigmoid > 0.5) and refine."""
        L, cfg, d = _lib.load(), self.cfg, instance_data
        if d['n'] == 0 or len(bboxes) == 0:
            return AnimeInstances()
        q = torch.as_tensor(np.asarray(bboxes), dtype=d['boxes'].dtype, device=self.device).view(-1, 4)
        t = d['boxes']
        lt, rb = torch.max(q[:, None, :2], t[None, :, :2]), torch.min(q[:, None, 2:], t[None, :, 2:])
        inter = (rb - lt).clamp(min=0).prod(2)
        area = lambda b: (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])          # noqa: E731  torchvision box_iou
        iou = inter / (area(q)[:, None] + area(t)[None, :] - inter)
        idx = iou.argmax(1)
        H, W = d['H'], d['W']
        long_side = max(H, W)
        feat, ld, fh, fw = self._mask_feat_of(d, mask_feat)          # the caller's tensor, as in the reference
        n = int(idx.numel())
        logits = torch.empty((n, fh, fw), dtype=torch.float32, device=self.device)
        # named, so both gathers are alive when the kernel is enqueued: two temporaries in one argument list can be handed the SAME
        # block by the caching allocator (the first is released before the second is allocated)
        ker_sel, pri_sel = d['kernels'][idx].contiguous(), d['priors'][idx].contiguous()
        check(L.csm_maskhead_logits(ptr(feat), i32(ld), i32(fh), i32(fw), i32(cfg.num_prototypes), i32(cfg.dyconv_channels),
                                    ptr(ker_sel), ptr(pri_sel), i32(n), i32(cfg.strides[0]), ptr(logits), stream_ptr()), "maskhead")
        masks = torch.empty((n, H, W), dtype=torch.uint8, device=self.device)
        check(L.csm_mask_resize_threshold(ptr(logits), i32(n), i32(fh), i32(fw), i32(cfg.strides[0]), i32(long_side),
                                          i32(long_side), i32(H), i32(W), f32(0.5), ptr(masks), stream_ptr()), "mask_resize")
        bb = t[idx].to(torch.int32)
        bb[:, 2:] -= bb[:, :2]
        inst = AnimeInstances(masks.bool(), bb, d['scores'][idx])
        if self.refine_method == 'refinenet_isnet':
            self._postprocess_refine(inst, img, refine_size=self.refine_size)
        return inst

    def _refine_many(self, pairs, refine_size=720, max_batch=None):
        """ISNet refine of several (instances, image) pairs of equal image size with shared batches (per-sample results are
        those of _postprocess_refine; the reference's per-image limit of 4 only bounds its memory use)."""
        L = _lib.load()
        jobs = []
        for inst, img in pairs:
            if inst.is_empty:
                continue
            img_d = self._upload(img)
            segs = inst.masks.to(self.device).to(torch.uint8).contiguous()
            jobs.append((inst, img_d, segs))
        if not jobs:
            return
        H, W = int(jobs[0][1].shape[0]), int(jobs[0][1].shape[1])
        T = refine_size
        rh, rw = scaledown_size(H, W, T)
        Hm, Wm = int(jobs[0][2].shape[1]), int(jobs[0][2].shape[2])       # equal image sizes => equal detector mask sizes
        rhm, rwm = scaledown_size(Hm, Wm, T)
        flat = [(j, k) for j, (_, _, segs) in enumerate(jobs) for k in range(segs.shape[0])]
        outs = [torch.empty((segs.shape[0], H, W), dtype=torch.uint8, device=self.device) for _, _, segs in jobs]
        max_batch = max_batch or self.refine_batch
        for c0 in range(0, len(flat), max_batch):
            chunk = flat[c0:c0 + max_batch]
            b = len(chunk)
            cp = self._refiner(b, T)
            batch = torch.empty((b, 4, T, T), dtype=torch.float32, device=self.device)
            for i, (j, k) in enumerate(chunk):
                check(L.csm_refine_prepare_batch(ptr(jobs[j][1]), ptr(jobs[j][2][k:k + 1]), i32(1), i32(H), i32(W), i32(rh), i32(rw),
                                                 i32(Hm), i32(Wm), i32(rhm), i32(rwm), i32(T), ptr(batch[i:i + 1]), stream_ptr()),
                      "refine_prepare")
            logits = torch.empty((b, 1, T, T), dtype=torch.float32, device=self.device)
            cp.run(batch, logits)
            for i, (j, k) in enumerate(chunk):
                check(L.csm_refine_threshold(ptr(logits[i:i + 1]), i32(1), i32(T), i32(T), i32(rh), i32(rw), i32(H), i32(W),
                                             f32(self.mask_thr), ptr(outs[j][k:k + 1]), stream_ptr()), "refine_threshold")
        for (inst, _, _), o in zip(jobs, outs):
            inst.masks = o.bool()

    # ---- ISNet refine (reference :638-665, :37-55) ---------------------------------------------------
    def _postprocess_refine(self, instances: AnimeInstances, img, refine_size: int = 720, max_refine_batch: int = 4, **kw):
        if instances.is_empty:
            return
        L = _lib.load()
        img_d = self._upload(img)
        H, W = int(img_d.shape[0]), int(img_d.shape[1])
        was_numpy = instances.is_numpy
        segs = (torch.from_numpy(instances.masks) if was_numpy else instances.masks).to(self.device).to(torch.uint8).contiguous()
        n, T = int(segs.shape[0]), refine_size
        rh, rw = scaledown_size(H, W, T)
        Hm, Wm = int(segs.shape[1]), int(segs.shape[2])
        rhm, rwm = scaledown_size(Hm, Wm, T)                       # resize_pad(seg): the seg's own shape (reference :47)
        out = torch.empty((n, H, W), dtype=torch.uint8, device=self.device)
        for k0 in range(0, n, max_refine_batch):
            b = min(max_refine_batch, n - k0)
            cp = self._refiner(b, T)
            batch = torch.empty((b, 4, T, T), dtype=torch.float32, device=self.device)
            check(L.csm_refine_prepare_batch(ptr(img_d), ptr(segs[k0:k0 + b]), i32(b), i32(H), i32(W), i32(rh), i32(rw), i32(Hm),
                                             i32(Wm), i32(rhm), i32(rwm), i32(T), ptr(batch), stream_ptr()), "refine_prepare")
            logits = torch.empty((b, 1, T, T), dtype=torch.float32, device=self.device)
            cp.run(batch, logits)
            check(L.csm_refine_threshold(ptr(logits), i32(b), i32(T), i32(T), i32(rh), i32(rw), i32(H), i32(W),
                                         f32(self.mask_thr), ptr(out[k0:k0 + b]), stream_ptr()), "refine_threshold")
        masks = out.bool()
        instances.masks = masks.cpu().numpy() if was_numpy else masks
