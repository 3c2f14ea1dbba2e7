"""ZoeDepth on libcsm355 -- host-side mirror of the reference's `depth_est: 'zoe'` path (anime_3dkenburns/kenburns_effect.py:540-544,
:812-818 -> depth_modules/__init__.py:40-47 load_zoe -> depth_modules/zoedepth/models/{depth_model.py:47-129, zoedepth/zoedepth_v1.py:
124-202, base_models/midas.py:49-350}).

Built here, all on the device:
  * DepthModel.infer: reflect padding by int(sqrt(size / 2) * 3), horizontal-flip test-time augmentation (the mirrored pass runs as extra
    samples of the same core / head run), bicubic resize of the prediction back to the padded size, crop, average of the two passes
    (csrc/zoedepth.hip);
  * MidasCore.forward's PrepForMidas: Resize.get_size ("minimal", multiple of 32, aspect ratio kept: config "infer": force_keep_ar),
    bilinear align_corners=True, Normalize(0.5, 0.5) -- fused with the padding / flip into one pass;
  * ZoeDepth.forward after `self.core(...)`: the metric-bins head as a layer program (nets/zoedepth_head.py);
  * the feature order MidasCore's hooks deliver: (out_conv, l4_rn, r4, r3, r2, r1) = layer_names of midas.py:189, with rel_depth.

The core itself -- `self.core.core`, the MiDaS DPT-BEiT-L network the reference fetches with torch.hub "intel-isl/MiDaS" + timm
(midas.py:341, not vendored) -- is built in (round 4): `DPTBeitCore` runs nets/dpt_beit.py's layer program (24 BEiT blocks on the fp32
MFMA engine + CSM_OP_LAYERNORM / ATTENTION / TOKENS, the DPT reassemble / fusion / head), lowered from the published timm / MiDaS
definitions with the checkpoint's parameter names [EXT: unpinned against MiDaS' own code, pinned against HuggingFace's DPT-BEiT
implementation by tests/test_oracle_dpt_beit.py].  `core=` still accepts any callable with the same contract: core(x [B,3,h,w]) ->
(rel_depth [B,h,w], [out_conv [B,32,h,w], bottleneck [B,256,h/32,w/32], r4 .. r1 [B,256,h/16 .. h/2, ...]]) as device tensors.
"""
import math

import numpy as np
import torch

from . import _lib
from ._lib import check, f32, i32, i64, ptr, stream_ptr
from .nets import build_zoe_head
from .runtime import CompiledProgram


class PrefixedWeights:
    """a weight source seen through a name prefix (ZoeD_M12_N.pt keeps the MiDaS network under `core.core.`)"""

    def __init__(self, ws, prefix):
        self.ws, self.prefix = ws, prefix

    def get(self, name, shape, kind):
        return self.ws.get(self.prefix + name, shape, kind)


class DPTBeitCore:
    """the MiDaS DPT-BEiT core as a callable: one compiled layer program per (batch, height, width) of the prepared input.

    Cost of a NEW shape (keep_aspect_ratio=True makes the prepared size follow the input's aspect ratio; the batch changes with the
    number of frames and with the flip TTA): a host re-pack of the 345 M BEiT-L parameters plus 24 re-sampled relative-position tables
    (seconds) and a device buffer of ~1.3 GB of packed weights + the workspace.  The packed image depends on the shape (the tables sit
    between the layers' panels; which 3x3 layers take the Winograd panels follows the per-sample map size), so programs cannot share one
    buffer.  The cache is therefore BOUNDED: the `max_programs` most recently used shapes stay resident (default 2, env
    CSM_ZOE_PROGRAM_CACHE), older ones are dropped and their HBM released -- a service fed arbitrary image sizes pays the re-pack again
    instead of leaking a gigabyte per shape.  INTEGRATION.md states the same."""

    def __init__(self, ws, cfg=None, device=None, max_programs=None):
        import collections
        import os
        from .nets import DPTBeitConfig
        self.ws, self.cfg = ws, cfg or DPTBeitConfig()
        self.device = torch.device('cuda', torch.cuda.current_device()) if device in (None, 'cuda') else torch.device(device)
        self.max_programs = max(1, int(max_programs if max_programs is not None else os.environ.get('CSM_ZOE_PROGRAM_CACHE', '2')))
        self._progs, self._weights = collections.OrderedDict(), None
        self.evictions = 0

    def program(self, n, h, w):
        from .nets import build_dpt_beit
        key = (n, h, w)
        if key in self._progs:
            self._progs.move_to_end(key)
            return self._progs[key]
        while len(self._progs) >= self.max_programs:         # least recently used first; its weights / workspace tensors die with it
            self._progs.popitem(last=False)
            self.evictions += 1
        if self.evictions:
            torch.cuda.current_stream(self.device).synchronize()     # (the evicted program may still be executing on this stream)
            torch.cuda.empty_cache()
        self._progs[key] = CompiledProgram(build_dpt_beit(self.ws, n, h, w, self.cfg), self.device)
        return self._progs[key]

    def __call__(self, xp):
        if not xp.is_cuda or xp.dtype != torch.float32 or xp.dim() != 4 or xp.shape[1] != 3:
            raise _lib.CsmError("DPTBeitCore: float32 device tensor [B,3,h,w] expected")
        n, _, h, w = (int(v) for v in xp.shape)
        if h % 32 or w % 32:
            raise _lib.CsmError("DPTBeitCore: the prepared input must be a multiple of 32 (PrepForMidas), got %dx%d" % (h, w))
        F, gh, gw = self.cfg.features, h // 16, w // 16
        dev = self.device
        rel = torch.empty((n, 1, h, w), dtype=torch.float32, device=dev)
        oc = torch.empty((n, self.cfg.head_features_2, h, w), dtype=torch.float32, device=dev)
        l4 = torch.empty((n, F, gh // 2, gw // 2), dtype=torch.float32, device=dev)
        rs = [torch.empty((n, F, gh << k, gw << k), dtype=torch.float32, device=dev) for k in range(4)]
        self.program(n, h, w).run(xp.contiguous(), rel, oc, l4, *rs)
        return rel.view(n, h, w), [oc, l4] + rs


def midas_size(width, height, net_w, net_h, keep_aspect_ratio=True, multiple_of=32):
    """Resize.get_size (midas.py:108-160) for resize_method 'minimal' (PrepForMidas default); returns (new_width, new_height)"""
    scale_h, scale_w = net_h / height, net_w / width
    if keep_aspect_ratio:
        if abs(1 - scale_w) < abs(1 - scale_h):
            scale_h = scale_w
        else:
            scale_w = scale_h
    cm = lambda v: int(np.round(v / multiple_of) * multiple_of)        # noqa: E731  constrain_to_multiple_of without bounds
    return cm(scale_w * width), cm(scale_h * height)


class ZoeDepth:
    def __init__(self, ws, core=None, img_size=(384, 512), keep_aspect_ratio=True, device=None, **head_kw):
        if not torch.cuda.is_available():
            raise _lib.CsmError("ZoeDepth needs an MI355X: libcsm355 has no CPU path")
        _lib.load()
        self.device = torch.device('cuda', torch.cuda.current_device()) if device in (None, 'cuda') else torch.device(device)
        self.ws, self.head_kw = ws, head_kw
        self._builtin = None
        self.core = core or self._builtin_core()
        self.net_h, self.net_w = (img_size, img_size) if isinstance(img_size, int) else tuple(img_size)
        self.keep_aspect_ratio = keep_aspect_ratio
        self._heads = {}

    def _builtin_core(self):
        """the MiDaS network of the same checkpoint: ZoeDepth's state_dict keeps it under `core.core.` (ZoeDepth.core = MidasCore, .core = DPT)"""
        if self._builtin is None:
            self._builtin = DPTBeitCore(PrefixedWeights(self.ws, 'core.core.'), device=self.device)
        return self._builtin

    def set_core(self, core):
        """core = None restores the built-in DPT-BEiT-L program"""
        self.core = core or self._builtin_core()

    # ---- ZoeDepth.forward (zoedepth_v1.py:124-202) on an already prepared input: core -> metric-bins head --------------------
    def _head(self, n, h, w, feat_sizes):
        key = (n, h, w, tuple(feat_sizes))
        if key not in self._heads:
            self._heads[key] = CompiledProgram(build_zoe_head(self.ws, n, h, w, list(feat_sizes), **self.head_kw), self.device)
        return self._heads[key]

    def forward_prepared(self, xp):
        """xp: what PrepForMidas hands the core [B,3,h,w] -> metric depth [B,1,h,w] (the size of the core's out_conv activation)"""
        rel, feats = self.core(xp)
        out_conv, btl, blocks = feats[0], feats[1], list(feats[2:])
        n, _, h, w = out_conv.shape
        ext = [rel.reshape(n, 1, h, w).float().contiguous(), out_conv.float().contiguous(), btl.float().contiguous()] + \
              [b.float().contiguous() for b in blocks]
        out = torch.empty((n, 1, h, w), dtype=torch.float32, device=self.device)
        self._head(n, h, w, [tuple(btl.shape[2:])] + [tuple(b.shape[2:]) for b in blocks]).run(*ext, out)
        return out

    # ---- DepthModel.infer (depth_model.py:47-129) --------------------------------------------------------------------------------
    def infer(self, x, pad_input=True, with_flip_aug=True):
        """x [B,3,H,W] float in [0,1] (device) -> metric depth [B,1,H,W]"""
        L = _lib.load()
        if not x.is_cuda or x.dtype != torch.float32 or x.dim() != 4 or x.shape[1] != 3:
            raise _lib.CsmError("ZoeDepth.infer: float32 device tensor [B,3,H,W] expected")
        x = x.contiguous()
        B, _, H, W = (int(v) for v in x.shape)
        pad_h = int(np.sqrt(H / 2) * 3) if pad_input else 0
        pad_w = int(np.sqrt(W / 2) * 3) if pad_input else 0
        Hp, Wp = H + 2 * pad_h, W + 2 * pad_w
        nw, nh = midas_size(Wp, Hp, self.net_w, self.net_h, self.keep_aspect_ratio)
        out = torch.empty((B, 1, H, W), dtype=torch.float32, device=self.device)
        # the flipped pass of infer_with_flip_aug (depth_model.py:113-129) rides in the SAME core / head run as the plain one, as samples
        # B .. 2B - 1: the layer programs are batch invariant (every sample's bits are those of a run by itself), the matrix pipe sees
        # twice the rows per weight tile and the attention twice the blocks
        flips = (0, 1) if with_flip_aug else (0,)
        xp = torch.empty((len(flips) * B, 3, nh, nw), dtype=torch.float32, device=self.device)
        for k, flip in enumerate(flips):
            check(L.csm_zoe_pad_prep(ptr(x), i32(B), i32(H), i32(W), i32(pad_h), i32(pad_w), i32(flip), i32(nh), i32(nw), ptr(xp[k * B:]),
                                     stream_ptr()), "zoe_pad_prep")
        d = self.forward_prepared(xp)
        for k, flip in enumerate(flips):
            check(L.csm_zoe_resize_crop(ptr(d[k * B:]), i32(B), i32(d.shape[2]), i32(d.shape[3]), i32(pad_h), i32(pad_w), i32(H), i32(W),
                                        i32(flip), i32(k), ptr(out), stream_ptr()), "zoe_resize_crop")
        return out


def depth_to_disparity(depth, focal, baseline):
    """_depth_est_zoe tail (kenburns_effect.py:815-817): zero fill with the smallest positive depth, focal * baseline / (depth + 1e-5),
    nan / inf -> 0"""
    L = _lib.load()
    depth = depth.contiguous()
    scratch = torch.empty(2, dtype=torch.int32, device=depth.device)
    check(L.csm_fill_zero_min_positive(ptr(depth), i64(depth.numel()), ptr(scratch), stream_ptr()), "fill_zero")
    out = torch.empty_like(depth)
    check(L.csm_zoe_depth_to_disparity(ptr(depth), i64(depth.numel()), f32(float(np.float32(focal * baseline))), ptr(out), stream_ptr()),
          "zoe_disparity")
    return out
