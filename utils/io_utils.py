"""reference utils/io_utils.py -- the image helpers on the hot path's edge: find_all_imgs (:92-103), scaledown_maxsize
(:254-274), resize_pad (:277-292), plus imread (mmcv.imread stand-in: the reference's callers decode with mmcv / cv2, which this
image does not have).  The resamplers run on the MI355X (cv2's uint8 INTER_LINEAR arithmetic restated in imageops.hip)."""
import os
import os.path as osp
from pathlib import Path

import numpy as np

IMG_EXT = {'.bmp', '.jpg', '.png', '.jpeg'}


def find_all_imgs(img_dir, abs_path=False):
    out = []
    for filename in os.listdir(img_dir):
        if Path(filename).suffix.lower() not in IMG_EXT:
            continue
        out.append(osp.join(img_dir, filename) if abs_path else filename)
    return out


def imread(path):
    """uint8 BGR HxWx3 like mmcv.imread / cv2.imread(IMREAD_COLOR) (EXIF orientation applied, alpha dropped)"""
    from PIL import Image, ImageOps
    with Image.open(path) as im:
        im = ImageOps.exif_transpose(im).convert('RGB')
        return np.ascontiguousarray(np.asarray(im)[:, :, ::-1])


def scaledown_size(im_h, im_w, max_size, divisior=None):
    """the size rule of scaledown_maxsize (utils/io_utils.py:256-270)"""
    r = max_size / max(im_h, im_w)
    if r < 1:
        if im_h > im_w:
            im_h, im_w = max_size, max(1, int(round(im_w * r)))
        else:
            im_w, im_h = max_size, max(1, int(round(im_h * r)))
    if divisior is not None:
        im_w = int(np.ceil(im_w / divisior) * divisior)
        im_h = int(np.ceil(im_h / divisior) * divisior)
    return im_h, im_w


def scaledown_maxsize(img, max_size: int, divisior: int = None):
    """utils/io_utils.py:254-274: cv2.resize(INTER_LINEAR) so that max(h, w) <= max_size (never enlarges, except for the
    `divisior` round-up).  numpy in -> numpy out, device tensor in -> device tensor out."""
    import torch
    from cartoonsegmentation_amd import ops
    h0, w0 = img.shape[:2]
    h, w = scaledown_size(h0, w0, max_size, divisior)
    if (h, w) == (h0, w0):
        return img
    if isinstance(img, torch.Tensor):
        if img.dtype == torch.uint8:
            return ops.resize_u8_linear(img, h, w)
        return ops.resize_f32_linear(img.float(), h, w).to(img.dtype if img.is_floating_point() else torch.float32)
    if img.dtype == np.uint8:
        return ops.resize_u8_linear(torch.from_numpy(np.ascontiguousarray(img)).cuda(), h, w).cpu().numpy()
    if img.dtype == np.bool_:
        raise TypeError("scaledown_maxsize: cv2.resize does not take bool arrays either -- pass uint8 or float32 masks")
    # float masks / images (prepare_refine_batch calls resize_pad(seg, ...), animeinsseg/__init__.py:47): cv2's float INTER_LINEAR
    out = ops.resize_f32_linear(torch.from_numpy(np.ascontiguousarray(img, dtype=np.float32)).cuda(), h, w).cpu().numpy()
    return out.astype(img.dtype) if img.dtype in (np.float64, np.float16) else out


def resize_pad(img, tgt_size: int, pad_value=(0, 0, 0)):
    """utils/io_utils.py:277-292: scaledown_maxsize, then pad bottom / right to tgt_size x tgt_size; returns (img, (pt, pb, pl, pr))"""
    img = scaledown_maxsize(img, tgt_size)
    h, w = img.shape[:2]
    pb, pr = tgt_size - h, tgt_size - w
    if pb + pr > 0:
        import torch
        v = pad_value[0] if isinstance(pad_value, (tuple, list)) else pad_value
        if isinstance(img, torch.Tensor):
            out = img.new_full((tgt_size, tgt_size) + tuple(img.shape[2:]), v)
            out[:h, :w] = img
            img = out
        else:
            pads = [(0, pb), (0, pr)] + [(0, 0)] * (img.ndim - 2)
            if isinstance(pad_value, (tuple, list)) and img.ndim == 3 and len(set(pad_value)) > 1:
                out = np.empty((tgt_size, tgt_size, img.shape[2]), img.dtype)
                out[:] = np.asarray(pad_value, img.dtype)
                out[:h, :w] = img
                img = out
            else:
                img = np.pad(img, pads, mode='constant', constant_values=v)
    return img, (0, pb, 0, pr)
