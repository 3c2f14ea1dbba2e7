"""CPU oracle of AnimeInsSeg.infer() (TEST INFRASTRUCTURE ONLY): numpy + oracle/post_oracle.c + the oracle
program interpreter.  Restates the flow of animeinsseg/__init__.py:401-504,638-665 and mmdet's predict_by_feat
independently of cartoonsegmentation_amd/segmentation.py (only the lowered Program objects are shared)."""
import ctypes
import math
import os
import subprocess

import numpy as np

from . import nets as onets

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def lib():
    global _LIB
    if _LIB is None:
        so, src = os.path.join(_HERE, "liboracle_post.so"), os.path.join(_HERE, "post_oracle.c")
        if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
            subprocess.check_call(["make", "-C", _HERE, "liboracle_post.so"], stdout=subprocess.DEVNULL)
        _LIB = ctypes.CDLL(so)
        _LIB.orc_sigmoid_scalar.restype = ctypes.c_float
        _LIB.orc_sigmoid_scalar.argtypes = [ctypes.c_float]
    return _LIB


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


ci, cf = ctypes.c_int, ctypes.c_float


def det_preprocess(img, S, cfg, rh, rw):
    H, W = img.shape[:2]
    x = np.empty((1, 3, S, S), np.float32)
    mean, std = np.asarray(cfg.mean, np.float32), np.asarray(cfg.std, np.float32)
    lib().orc_det_preprocess(_p(np.ascontiguousarray(img)), ci(H), ci(W), ci(rh), ci(rw), ci(S), ci(S), _p(mean), _p(std),
                             cf(cfg.pad_value), _p(x))
    return x


def nms(boxes, offs, thr, max_keep):
    keep = np.zeros(max_keep, np.int32)
    n = lib().orc_nms(_p(np.ascontiguousarray(boxes, np.float32)), None if offs is None else _p(offs), ci(len(boxes)), cf(thr),
                      ci(max_keep), _p(keep))
    return keep[:n]


def maskhead_logits(mask_feat_hwc, kernels, priors, feat_stride):
    h, w, c = mask_feat_hwc.shape
    n = len(priors)
    out = np.empty((n, h, w), np.float32)
    lib().orc_maskhead_logits(_p(np.ascontiguousarray(mask_feat_hwc)), ci(c), ci(h), ci(w), _p(np.ascontiguousarray(kernels, np.float32)),
                              _p(np.ascontiguousarray(priors, np.float32)), ci(n), ci(feat_stride), _p(out))
    return out


def mask_resize_threshold(logits, up, rh, rw, oh, ow, thr):
    n, h, w = logits.shape
    out = np.empty((n, oh, ow), np.uint8)
    lib().orc_mask_resize_threshold(_p(np.ascontiguousarray(logits)), ci(n), ci(h), ci(w), ci(up), ci(rh), ci(rw), ci(oh), ci(ow),
                                    cf(thr), _p(out))
    return out


def scaledown_size(h, w, max_size):
    """utils/io_utils.py:254-266"""
    r = max_size / max(h, w)
    if r < 1:
        if h > w:
            h, w = max_size, max(1, int(round(w * r)))
        else:
            w, h = max_size, max(1, int(round(h * r)))
    return h, w


def refine_prepare_batch(img, masks_u8, rh, rw, T):
    n, Hm, Wm = masks_u8.shape
    H, W = img.shape[:2]
    rhm, rwm = scaledown_size(Hm, Wm, T)
    out = np.empty((n, 4, T, T), np.float32)
    lib().orc_refine_prepare_batch(_p(np.ascontiguousarray(img)), _p(np.ascontiguousarray(masks_u8)), ci(n), ci(H), ci(W), ci(rh),
                                   ci(rw), ci(Hm), ci(Wm), ci(rhm), ci(rwm), ci(T), _p(out))
    return out


def refine_threshold(logits, ch, cw, oh, ow, thr):
    n, _, S_h, S_w = logits.shape
    out = np.empty((n, oh, ow), np.uint8)
    lib().orc_refine_threshold(_p(np.ascontiguousarray(logits)), ci(n), ci(S_h), ci(S_w), ci(ch), ci(cw), ci(oh), ci(ow), cf(thr), _p(out))
    return out


def detect(img, rp, cfg, S, pred_score_thr=0.3):
    """RTMDet forward through the oracle interpreter + mmdet predict_by_feat restated in numpy"""
    H, W = img.shape[:2]
    f = min(S / max(H, W), S / min(H, W))
    rh, rw = int(H * float(f) + 0.5), int(W * float(f) + 0.5)
    w_scale, h_scale = rw / W, rh / H
    x = det_preprocess(img, S, cfg, rh, rw)
    views = onets.run_program(rp.prog, [x], want_views=rp.cls + rp.reg + rp.kern + [rp.mask_feat])
    sc_l, box_l, pri_l, ker_l, lab_l = [], [], [], [], []
    for lvl, stride in enumerate(cfg.strides):
        cls = views[rp.cls[lvl]].reshape(-1, cfg.num_classes)
        reg = views[rp.reg[lvl]].reshape(-1, 4) * np.float32(stride)
        ker = views[rp.kern[lvl]].reshape(-1, cfg.num_gen_params)
        hl, wl = rp.cls[lvl].h, rp.cls[lvl].w
        ys, xs = np.meshgrid(np.arange(hl), np.arange(wl), indexing='ij')
        pri = np.stack([xs.reshape(-1) * stride, ys.reshape(-1) * stride, np.full(hl * wl, stride), np.full(hl * wl, stride)], 1).astype(np.float32)
        idx = np.argwhere(cls > np.float32(cfg.score_thr))
        sc = cls[idx[:, 0], idx[:, 1]]
        order = np.argsort(-sc, kind='stable')[:cfg.nms_pre]
        idx, sc = idx[order], sc[order]
        sc_l.append(sc); lab_l.append(idx[:, 1]); box_l.append(reg[idx[:, 0]]); pri_l.append(pri[idx[:, 0]]); ker_l.append(ker[idx[:, 0]])
    scores, labels = np.concatenate(sc_l), np.concatenate(lab_l)
    dist, priors, kernels = np.concatenate(box_l), np.concatenate(pri_l), np.concatenate(ker_l)
    x1 = np.clip(priors[:, 0] - dist[:, 0], 0, np.float32(rw)); y1 = np.clip(priors[:, 1] - dist[:, 1], 0, np.float32(rh))
    x2 = np.clip(priors[:, 0] + dist[:, 2], 0, np.float32(rw)); y2 = np.clip(priors[:, 1] + dist[:, 3], 0, np.float32(rh))
    sf = np.array([1 / w_scale, 1 / h_scale] * 2, np.float32)
    boxes = (np.stack([x1, y1, x2, y2], 1) * sf).astype(np.float32)
    if cfg.min_bbox_size >= 0:
        ok = ((boxes[:, 2] - boxes[:, 0]) > cfg.min_bbox_size) & ((boxes[:, 3] - boxes[:, 1]) > cfg.min_bbox_size)
        scores, labels, boxes, priors, kernels = scores[ok], labels[ok], boxes[ok], priors[ok], kernels[ok]
    if len(scores) == 0:
        return dict(n=0, H=H, W=W)
    order = np.argsort(-scores, kind='stable')[:4096]
    scores, labels, boxes, priors, kernels = scores[order], labels[order], boxes[order], priors[order], kernels[order]
    offs = (labels.astype(np.float32) * (boxes.max() + 1)).astype(np.float32) if cfg.num_classes > 1 else None
    keep = nms(boxes, offs, cfg.nms_iou, cfg.max_per_img)
    scores, boxes, priors, kernels = scores[keep], boxes[keep], priors[keep], kernels[keep]
    sel = np.nonzero(scores > np.float32(pred_score_thr))[0]
    if len(sel) == 0:
        return dict(n=0, H=H, W=W)
    mf = views[rp.mask_feat][0]
    logits = maskhead_logits(mf, kernels[sel], priors[sel], cfg.strides[0])
    up = cfg.strides[0]
    rh2 = math.ceil(mf.shape[0] * up * (1 / w_scale)); rw2 = math.ceil(mf.shape[1] * up * (1 / h_scale))
    masks = mask_resize_threshold(logits, up, rh2, rw2, min(rh2, H), min(rw2, W), cfg.mask_thr_binary)   # [..., :ori_h, :ori_w] is a slice
    bb = boxes[sel].astype(np.int32)
    bb[:, 2:] -= bb[:, :2]
    return dict(n=len(sel), H=H, W=W, masks=masks, bboxes=bb, scores=scores[sel], logits=logits, boxes_f=boxes[sel])


def refine(img, masks_u8, isnet_prog_for, T, mask_thr, max_batch=4):
    """_postprocess_refine (reference :638-665); isnet_prog_for(b) -> oracle-runnable Program for batch b"""
    n = masks_u8.shape[0]
    H, W = img.shape[:2]
    rh, rw = scaledown_size(H, W, T)
    out = np.empty((n, H, W), np.uint8)
    for k0 in range(0, n, max_batch):
        b = min(max_batch, n - k0)
        batch = refine_prepare_batch(img, masks_u8[k0:k0 + b], rh, rw, T)
        logits = np.zeros((b, 1, T, T), np.float32)
        onets.run_program(isnet_prog_for(b), [batch, logits])
        out[k0:k0 + b] = refine_threshold(logits, rh, rw, H, W, mask_thr)
    return out
