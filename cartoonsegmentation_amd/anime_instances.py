"""AnimeInstances result container -- API mirror of animeinsseg/anime_instances.py:31-298 (reference).

masks: bool [n,H,W]; bboxes: int32 [n,4] xywh; scores: float [n]; numpy or torch (cpu/cuda).
Written from the reference's observable behaviour (properties, conversions, resize, compose_masks,
remove_duplicated, get_instance, draw_instances); drawing uses numpy only (cv2 is not a dependency).
"""
from typing import List

import numpy as np
import torch

class Colors:
    """instance colours of the notebook / draw_instances: utils/constants.py:44-57 (hex table -> RGB, returned as BGR by default)"""
    _HEX = ('FF1010', '10FF10', 'FFF010', '100FFF', '0018EC', 'FF3838', 'FF9D97', 'FF701F', 'FFB21D', 'CFD231', '48F90A', '92CC17',
            '3DDB86', '1A9334', '00D4BB', '2C99A8', '00C2FF', '344593', '6473FF', '0018EC', '8438FF', '520085', 'CB38FF', 'FF95C8',
            'FF37C7')

    def __init__(self):
        self.palette = [tuple(int(h[i:i + 2], 16) for i in (0, 2, 4)) for h in self._HEX]
        self.n = len(self.palette)

    def __call__(self, i, bgr=True):
        r, g, b = self.palette[int(i) % self.n]
        return (b, g, r) if bgr else (r, g, b)


colors = Colors()


def get_color(idx):
    """utils/constants.py:59-63"""
    return 255 if idx == -1 else colors(idx)


class AnimeInstances:
    def __init__(self, masks=None, bboxes=None, scores=None, tags: List[str] = None, character_tags: List[str] = None):
        self.masks, self.bboxes = masks, bboxes
        if scores is None:
            scores = [1.0] * len(self)
            scores = np.array(scores) if self.is_numpy else (torch.tensor(scores) if self.is_tensor else scores)
        self.scores = scores
        if tags is None:
            self.tags, self.character_tags = [''] * len(self), [''] * len(self)
        else:
            self.tags, self.character_tags = tags, character_tags

    # ---- state -------------------------------------------------------------------------------
    @property
    def is_empty(self):
        return self.masks is None or len(self.masks) == 0

    @property
    def is_tensor(self):
        return (not self.is_empty) and isinstance(self.masks, torch.Tensor)

    @property
    def is_numpy(self):
        return self.is_empty or isinstance(self.masks, np.ndarray)

    @property
    def is_cuda(self):
        return self.is_tensor and self.masks.is_cuda

    def __len__(self):
        return 0 if self.is_empty else len(self.masks)

    # ---- conversions --------------------------------------------------------------------------
    def to_tensor(self, device='cpu'):
        if self.is_empty:
            return self
        if self.is_tensor:
            self.masks, self.bboxes, self.scores = (t.to(device) for t in (self.masks, self.bboxes, self.scores))
            return self
        self.masks = torch.from_numpy(self.masks).to(device)
        self.bboxes = torch.from_numpy(self.bboxes).to(device)
        self.scores = torch.from_numpy(np.asarray(self.scores)).to(device)
        return self

    def cuda(self):
        return self if self.is_empty else self.to_tensor('cuda')

    def cpu(self):
        if self.is_cuda:
            self.masks, self.bboxes, self.scores = self.masks.cpu(), self.bboxes.cpu(), self.scores.cpu()
        return self

    def to_numpy(self):
        if not self.is_numpy:
            self.masks, self.bboxes, self.scores = (t.detach().cpu().numpy() for t in (self.masks, self.bboxes, self.scores))
        return self

    def get_instance(self, ins_idx, out_type=None, device=None):
        mask, bbox, score = self.masks[ins_idx], self.bboxes[ins_idx], self.scores[ins_idx]
        if out_type == 'numpy' and not self.is_numpy:
            mask, bbox, score = mask.cpu().numpy(), bbox.cpu().numpy(), score.cpu().numpy()
        if out_type == 'tensor' and not self.is_tensor:
            mask, bbox, score = torch.from_numpy(mask), torch.from_numpy(bbox), torch.from_numpy(np.asarray(score))
        if isinstance(mask, torch.Tensor) and device is not None:
            mask, bbox, score = mask.to(device), bbox.to(device), score.to(device)
        return {'mask': mask, 'tags': self.tags[ins_idx], 'character_tags': self.character_tags[ins_idx], 'bbox': bbox,
                'score': score}

    # ---- geometry -----------------------------------------------------------------------------
    def resize(self, h, w, mode='area'):
        """anime_instances.py:268-280 (tensor instances only, like the reference; note the reference scales
        bbox columns 0,2 by the HEIGHT ratio and 1,3 by the WIDTH ratio -- kept)"""
        if self.is_empty or not self.is_tensor:
            return
        oh, ow = int(self.masks.shape[1]), int(self.masks.shape[2])
        hs, ws = h / oh, w / ow
        bboxes = self.bboxes.float()
        bboxes[:, ::2] *= hs
        bboxes[:, 1::2] *= ws
        self.bboxes = torch.round(bboxes).int()
        if (oh, ow) == (h, w):        # identity resize: interpolate(area) of a 0/1 mask then > 0.3 returns the mask itself
            self.masks = self.masks.to(torch.float) > 0.3
        elif self.masks.is_cuda and mode == 'area' and self.masks.dtype == torch.bool:
            # device instances (the pipeline's case): adaptive-average resize + threshold in one hand-written pass over the 1-B masks
            from . import _lib
            from ._lib import check, f32, i32, ptr, stream_ptr
            src = self.masks.contiguous().view(torch.uint8)
            out = torch.empty((src.shape[0], h, w), dtype=torch.uint8, device=src.device)
            check(_lib.load().csm_mask_area_resize_threshold(ptr(src), i32(src.shape[0]), i32(oh), i32(ow), i32(h), i32(w), f32(0.3), ptr(out),
                                                             stream_ptr(src.device)), "mask_area_resize")
            self.masks = out.view(torch.bool)
        else:
            self.masks = torch.nn.functional.interpolate(self.masks.to(torch.float).unsqueeze(1), (h, w), mode=mode).squeeze(1) > 0.3

    def compose_masks(self, output_type=None):
        if self.is_empty:
            return None
        mask = self.masks[0]
        for m in self.masks[1:]:
            mask = np.logical_or(mask, m) if self.is_numpy else torch.logical_or(mask, m)
        if output_type == 'numpy' and not self.is_numpy:
            mask = mask.cpu().numpy()
        if output_type == 'tensor' and not self.is_tensor:
            mask = torch.from_numpy(mask)
        return mask

    def remove_duplicated(self):
        """drop instances whose mask is >80% covered by the union of larger ones (anime_instances.py:84-127)"""
        n = len(self)
        if n < 2:
            return
        back = self.is_numpy
        if back:
            self.to_tensor()
        areas = torch.tensor([float(m.sum()) for m in self.masks])
        order = torch.argsort(areas, descending=True).tolist()
        areas = areas[order]
        masks, bboxes, scores = self.masks[order], self.bboxes[order], self.scores[order]
        tags = [self.tags[i] for i in order]
        canvas, valid = masks[0], list(range(n))
        for k in range(1, n):
            inter = torch.bitwise_and(canvas, masks[k]).sum()
            if inter / areas[k] > 0.8:
                valid.remove(k)
            elif k != n - 1:
                canvas = torch.bitwise_or(canvas, masks[k])
        self.masks, self.bboxes, self.scores = masks[valid], bboxes[valid], scores[valid]
        self.tags = [tags[i] for i in valid]
        if back:
            self.to_numpy()

    def draw_instances(self, img, draw_bbox=True, draw_ins_mask=True, draw_ins_contour=False, draw_tags=False,
                       draw_indices=None, mask_alpha=0.4):
        """overlay masks (alpha blend) and box outlines on a BGR image; returns uint8 HxWx3"""
        canvas = np.array(img, dtype=np.float32, copy=True)
        if self.is_empty:
            return canvas.astype(np.uint8)
        masks = self.masks.cpu().numpy() if self.is_tensor else self.masks
        bboxes = self.bboxes.cpu().numpy() if self.is_tensor else self.bboxes
        for i in (range(len(masks)) if draw_indices is None else draw_indices):
            col = np.array(get_color(i), np.float32)
            if draw_ins_mask:
                m = masks[i].astype(bool)
                canvas[m] = canvas[m] * (1 - mask_alpha) + col * mask_alpha
            if draw_bbox:
                x, y, w, h = [int(v) for v in bboxes[i]]
                x2, y2 = min(x + w, canvas.shape[1] - 1), min(y + h, canvas.shape[0] - 1)
                canvas[y:y2 + 1, [x, x2]] = col
                canvas[[y, y2], x:x2 + 1] = col
        return canvas.clip(0, 255).astype(np.uint8)
