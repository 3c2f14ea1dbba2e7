"""KenBurnsPipeline -- host-side mirror of anime_3dkenburns/kenburns_effect.py (reference :207-1090) on libcsm355.

Call surface kept: KenBurnsConfig (dataclass + dict-style aliases), KenBurnsPipeline(cfg, device)
.generate_kenburns_config / .autozoom / .process_kenburns, npyframes2video; module-level operators live in
cartoonsegmentation_amd.ops (render_pointcloud, fill_disocclusion, process_shift, ...).

MI355X-first differences (results unchanged):
  * LeReS runs as one layer program; its uint8 pre/post-processing stays on the device (imageops.hip);
  * process_autozoom (common.py:86-142) evaluates the <=256 candidate shifts in batched launches that only compute what
    coverage needs (z-buffer, degrid, z-test: autozoom.hip); the counts are read back once;
  * the 75-frame loop (kenburns_effect.py:1015-1072) is the fused csm_warp_frame + csm_crop_resize_u8; frames are
    copied to the host once at the end (the reference does a 12 MB D2H per frame).
Point-cloud inpainting (Inpaint GridNet, :441-512), bokeh depth-of-field (depth_field=True, :1042-1067), the `Refine` depth
refinement (:619-622) and the sniklaus `default` estimator are built; out of scope: ldm/patchmatch inpainting, CRF refinement, the
un-vendored BEiT core of ZoeDepth and Marigold.
"""
import math
import os
from copy import deepcopy
from dataclasses import dataclass, field, fields
from typing import Union

import numpy as np
import torch

from . import _lib, ops
from ._lib import check, f32, i32, i64, ptr, stream_ptr
from .anime_instances import AnimeInstances
from .nets import build_disparity, build_inpaint_context, build_inpaint_grid, build_leres, build_refine, build_semantics
from .runtime import CompiledProgram
from .segmentation import AnimeInsSeg, scaledown_size
from .weights import StateDictWeights, SynthWeights

_ALIASES = {'fltFocal': 'focal', 'fltBaseline': 'baseline', 'intWidth': 'int_width', 'intHeight': 'int_height',
            'fltDispmin': 'disparity_min', 'fltDispmax': 'disparity_max', 'objDepthrange': 'depth_range',
            'tenRawImage': 'tensor_raw_image', 'tenRawDisparity': 'raw_disparity', 'tenRawDepth': 'raw_depth',
            'tenRawPoints': 'raw_point', 'tenRawUnaltered': 'raw_unaltered', 'tenInpaImage': 'inpainted_img',
            'tenInpaDisparity': 'inpainted_disparity', 'tenInpaDepth': 'inpainted_depth', 'tenInpaPoints': 'inpainted_points'}


def _synthetic_ok():
    return os.environ.get('CSM_SYNTHETIC_WEIGHTS', '0') == '1'


@dataclass
class KenBurnsConfig:
    """field names / defaults of the reference dataclass (kenburns_effect.py:207-290); yaml keys map 1:1"""
    detector: str = 'animeinsseg'
    det_ckpt: str = 'models/AnimeInstanceSegmentation/rtmdetl_e60.ckpt'
    det_size: int = 640
    scale_depth: bool = False
    depth_field: bool = False
    mask_refine_kwargs: dict = field(default_factory=dict)
    marigold_kwargs: dict = field(default_factory=dict)
    pred_score_thr: float = 0.3
    depth_est: str = 'zoe'
    depth_est_device: str = ''
    depth_refinement: str = 'default'
    depthest_use_medium: bool = False
    inpaint_type: str = 'default'
    num_frame: int = 75
    playback: bool = True
    auto_zoom: bool = True
    focal: float = 1024 / 2.0
    baseline: float = 40.0
    dof_speed: float = 50.
    depth_factor: int = 1
    lightness_factor: int = 13
    max_size: int = 720
    int_height: int = 1024
    int_width: int = 1024
    default_depth_refine: bool = False
    refine_crf: bool = True
    depth_est_size: int = 640
    sd_img2img_url: str = 'http://127.0.0.1:7860/sdapi/v1/img2img'
    ldm_inpaint_options: dict = field(default_factory=dict)
    ldm_inpaint_size: int = 0
    instances: AnimeInstances = None
    # run-time state (not constructor fields in the reference either)
    disparity_min = 0
    disparity_max = 0
    depth_range = None
    tensor_raw_image = None
    original_img_nparray = None
    raw_disparity = None
    raw_depth = None
    raw_point = None
    raw_unaltered = None
    inpainted_img = None
    inpainted_disparity = None
    inpainted_depth = None
    inpainted_points = None
    save_path = r''
    stage_inpainted_imgs = None
    stage_inpainted_masks = None
    stage_depth_coarse = None
    stage_depth_adjusted = None
    stage_depth_final = None

    def __getitem__(self, item):
        return getattr(self, _ALIASES.get(item, item))

    def __setitem__(self, item, value):
        setattr(self, _ALIASES.get(item, item), value)

    def copy(self):
        return deepcopy(self)


def build_kenburns_cfg(tgt_cfg: Union[str, dict]):
    """kenburns_effect.py:369-374 (OmegaConf replaced by PyYAML: same keys)"""
    if isinstance(tgt_cfg, str):
        import yaml
        with open(tgt_cfg) as f:
            tgt_cfg = yaml.safe_load(f)
    names = {f.name for f in fields(KenBurnsConfig) if f.init}
    return KenBurnsConfig(**{k: v for k, v in dict(tgt_cfg).items() if k in names})


def colorize_depth(depth, inverse=False, rgb2bgr=False, cmap='magma_r'):
    """debug visualisation of kenburns_effect.py:382-389 / zoedepth colorize (2nd..85th percentile) -- only used by
    --verbose callers; needs matplotlib, otherwise falls back to a grey ramp."""
    d = np.asarray(depth, np.float32).squeeze()
    if inverse:
        d = 1 / (d + 1e-5)
    lo, hi = np.percentile(d, 2), np.percentile(d, 85)
    v = (d - lo) / (hi - lo) if hi != lo else d * 0
    try:
        import matplotlib
        col = matplotlib.colormaps[cmap](v, bytes=True)[..., :3]
    except Exception:
        g = (np.clip(v, 0, 1) * 255).astype(np.uint8)
        col = np.stack([g, g, g], -1)
    return col[..., ::-1].copy() if rgb2bgr else col


def _mm_scratch(t):
    """512 floats of block partials for csm_minmax"""
    return torch.empty(512, dtype=torch.float32, device=t.device)


def _fill_zero_with_min_positive(depth):
    """`depth[depth == 0] = depth[depth > 0].min()` (leres/__init__.py:143-145 semantics: skipped when nothing is positive),
    in place on the device, no host sync"""
    scratch = torch.empty(2, dtype=torch.int32, device=depth.device)
    check(_lib.load().csm_fill_zero_min_positive(ptr(depth), i64(depth.numel()), ptr(scratch), stream_ptr()), "fill_zero")
    return depth


def depth_adjustment_animesseg(instances, tenDisparity, tenImage, use_medium=False):
    """kenburns_effect.py:39-91: flatten every instance to the disparity at the bottom 3% of its rows"""
    assert tenDisparity.shape[0] == 1
    masks = [] if instances.is_empty else ([instances.masks[i].float() for i in range(instances.masks.shape[0])] if use_medium else [True])
    resized = tenDisparity.shape[2:] != tenImage.shape[2:]
    adj = ops.resize_bilinear(tenDisparity, int(tenImage.shape[2]), int(tenImage.shape[3])) if resized else tenDisparity
    if use_medium:
        for m in masks:
            plane = adj * m
            if plane.sum().item() == 0:
                continue
            adj[plane > 0] = adj[plane > 0].median()
    elif masks:
        # kenburns_effect.py:68-78 per instance, fused: row maxima / row flags -> (top, bottom, r0, value) -> apply; 3 small
        # kernels per instance, in place on a private copy, no host sync (the reference syncs 4x per instance)
        H, W = int(adj.shape[2]), int(adj.shape[3])
        adj = adj.clone() if not resized else adj.contiguous()
        scratch = torch.empty(2 * H + 2, dtype=torch.float32, device=adj.device)
        mk = instances.masks if isinstance(instances.masks, torch.Tensor) else torch.as_tensor(instances.masks)
        mk = mk.to(adj.device)
        mk = mk if mk.dtype in (torch.bool, torch.uint8) else (mk > 0)
        mk = mk.contiguous().view(torch.uint8)
        for i in range(mk.shape[0]):
            check(_lib.load().csm_depth_adjust_instance(ptr(adj), ptr(mk[i]), i32(H), i32(W), ptr(scratch), stream_ptr()), "depth_adjust")
    if resized:
        return ops.resize_bilinear(adj, int(tenDisparity.shape[2]), int(tenDisparity.shape[3]))
    return adj


class KenBurnsPipeline:
    def __init__(self, cfg: Union[KenBurnsConfig, str, dict] = None, device: str = None) -> None:
        if cfg is None:
            cfg = KenBurnsConfig()
        elif isinstance(cfg, (str, dict)):
            cfg = build_kenburns_cfg(cfg)
        elif not isinstance(cfg, KenBurnsConfig):
            raise NotImplementedError
        self.cfg = cfg
        if not torch.cuda.is_available():
            raise _lib.CsmError("KenBurnsPipeline needs an MI355X: libcsm355 has no CPU path")
        self.device = torch.device('cuda:%d' % torch.cuda.current_device()) if device in (None, 'cuda') else torch.device(device)
        self.animeinsseg = None
        self._leres, self._leres_weights, self._leres_ws = {}, {}, None
        self._refine_ws, self._refine_progs = None, {}
        self.max_instances = 100                 # AnimeInsSeg.infer default (animeinsseg/__init__.py:417)
        self.overlap_depth = True                # MI355X: LeReS runs on a second HIP stream next to the segmentation nets
        self._side_stream = None
        self._side_streams = []
        self.depth_streams = int(os.environ.get('CSM_DEPTH_STREAMS', '1'))   # batched path: LeReS sub-batches on this many side streams
        self.frame_streams = int(os.environ.get('CSM_FRAME_STREAMS', '3'))   # process_kenburns: output frames in flight (1 = the serial loop)
        self._frame_streams = []
        self._lane_wf = {}
        self.set_detector(cfg.detector)
        self.set_depth_estimation(cfg.depth_est)
        if self.cfg.default_depth_refine:
            self.set_depth_refinement(cfg.depth_refinement)
        self.set_inpainting(cfg.inpaint_type)

    # ---- component selection (kenburns_effect.py:425-440, :514-546) ---------------------------------
    def set_detector(self, detector: str):
        if detector != 'animeinsseg':
            raise NotImplementedError("detector %r: only 'animeinsseg' is on the hot path" % detector)
        if self.animeinsseg is None:
            ckpt = self.cfg.det_ckpt
            if not os.path.exists(ckpt):
                if not _synthetic_ok() and not str(ckpt).startswith('synthetic'):
                    raise FileNotFoundError("%s (set det_ckpt='synthetic' or CSM_SYNTHETIC_WEIGHTS=1 for closed-form weights)" % ckpt)
                ckpt = 'synthetic'
            # kenburns_effect.py:832-837: AnimeInsSeg(cfg.det_ckpt, device) -- the reference never applies cfg.det_size
            # (SURVEY F8); callers that want another detector size use animeinsseg.set_detect_size / infer(det_size=)
            self.animeinsseg = AnimeInsSeg(ckpt, device=str(self.device))

    def set_depth_estimation(self, depth_est: str):
        if depth_est == 'default':                       # kenburns_effect.py:547-548: the original 3D-Ken-Burns estimator
            self._set_default_estimator()
            return
        if depth_est == 'zoe':                           # kenburns_effect.py:541-544
            self._set_zoe_estimator()
            return
        if depth_est != 'leres':
            raise NotImplementedError("depth_est %r: 'leres' (the shipped yaml), 'default' (sniklaus Disparity + VGG19-BN) and 'zoe' (with "
                                      "a MiDaS core plugged in) are built; marigold needs the un-vendored diffusers pipeline (SURVEY F4)" % depth_est)
        if self._leres_ws is None:
            p = os.environ.get('CSM_LERES_CKPT', 'models/leres/res101.pth')
            if not os.path.exists(p) and not _synthetic_ok():
                raise FileNotFoundError("%s (set CSM_SYNTHETIC_WEIGHTS=1 for closed-form weights)" % p)
            if os.path.exists(p):
                sd = torch.load(p, map_location='cpu', weights_only=False)['depth_model']
                self._leres_ws = StateDictWeights({'depth_model.' + k.replace('module.', '', 1): v for k, v in sd.items()})
            else:
                self._leres_ws = SynthWeights('leres.')
        self._depth_est = self._depth_est_leres

    def _set_zoe_estimator(self):
        """depth_modules/__init__.py:40-47 load_zoe(DEPTH_ZOE_CKPT, img_size=[672, 672]): the head's and the MiDaS DPT-BEiT-L core's weights
        come from the checkpoint (`core.core.*`; or closed-form).  The core is the built-in layer program (zoedepth.DPTBeitCore: the
        reference downloads that network with torch.hub and does not vendor it) unless self.set_zoe_core plugged another callable"""
        from .zoedepth import ZoeDepth
        if getattr(self, 'depth_zoe', None) is None:
            p = 'models/AnimeInstanceSegmentation/ZoeD_M12_N.pt'                     # utils/constants.py:82
            if os.path.exists(p):
                sd = torch.load(p, map_location='cpu', weights_only=False)
                ws = StateDictWeights(sd.get('model', sd))
            elif _synthetic_ok() or str(self.cfg.det_ckpt).startswith('synthetic'):
                ws = SynthWeights('zoe.')
            else:
                raise FileNotFoundError("%s (set CSM_SYNTHETIC_WEIGHTS=1 for closed-form head weights)" % p)
            self.depth_zoe = ZoeDepth(ws, core=getattr(self, '_zoe_core', None), img_size=(672, 672), keep_aspect_ratio=True, device=self.device)
        self._depth_est = self._depth_est_zoe

    def set_zoe_core(self, core):
        """plug a MiDaS core in place of the built-in one: core(x_prepared [B,3,h,w]) -> (rel_depth [B,h,w], [out_conv, bottleneck, r4, r3,
        r2, r1]); B = 2 x frames when the flip TTA is on (the mirrored pass rides in the same call).  None restores the built-in program"""
        self._zoe_core = core
        if getattr(self, 'depth_zoe', None) is not None:
            self.depth_zoe.set_core(core)

    def _depth_est_zoe(self, img_tensor, img_d):
        """kenburns_effect.py:812-818"""
        from .zoedepth import depth_to_disparity
        if img_tensor is None:
            img_tensor = ops.image_tensor(img_d)                                    # permute(2, 0, 1)[None].float() * (1.0 / 255.0)
        depth = self.depth_zoe.infer(img_tensor, with_flip_aug=True, pad_input=True)
        return depth_to_disparity(depth, self.cfg.focal, self.cfg.baseline)

    def _depth_est_zoe_batch(self, frames_d):
        """_depth_est_zoe for several frames of ONE size: a single DepthModel.infer over the stack (B frames + their mirrored passes = 2 B
        samples of one core run: the layer programs are batch invariant, every frame's bits are those of a run by itself)"""
        from .zoedepth import depth_to_disparity
        n = len(frames_d)
        # The core keeps one compiled program per (2 B, prepared size) and a new one costs a host re-pack of 345 M parameters: a tail
        # group (fewer frames than the group before it, same frame size) is PADDED to the resident batch with copies of its last frame
        # instead of building a second program -- the layer programs are batch invariant, so the kept frames' bits do not change
        key = tuple(frames_d[0].shape)
        resident = getattr(self, '_zoe_group', None)
        pad = resident[1] - n if resident is not None and resident[0] == key and resident[1] > n else 0
        x = torch.cat([ops.image_tensor(f) for f in frames_d] + [ops.image_tensor(frames_d[-1])] * pad, 0)
        if pad == 0:
            self._zoe_group = (key, n)
        depth = self.depth_zoe.infer(x, with_flip_aug=True, pad_input=True)
        return [depth_to_disparity(depth[k:k + 1].contiguous(), self.cfg.focal, self.cfg.baseline) for k in range(n)]

    def _set_default_estimator(self):
        """anime_3dkenburns/models/__init__.py:33-52: Semantics (torchvision vgg19_bn) + Disparity (network-disparity.pytorch)"""
        if getattr(self, '_disp_ws', None) is None:
            pd_, pv = 'models/kenburns/network-disparity.pytorch', 'models/kenburns/vgg19_bn.pth'
            if os.path.exists(pd_) and os.path.exists(pv):
                sd = torch.load(pd_, map_location='cpu', weights_only=False)
                self._disp_ws = StateDictWeights({k.replace('module', 'net'): v for k, v in sd.items()})      # models/__init__.py:42
                vg = torch.load(pv, map_location='cpu', weights_only=False)
                from .nets.disparity import VGG
                ren = {}
                for e in VGG:                                 # torchvision names features.<i>.* -> the reference module's netVgg.<slice>.<i>.*
                    if e != 'M':
                        for i in (e[1], e[1] + 1):
                            for k, v in vg.items():
                                if k.startswith('features.%d.' % i):
                                    ren['netVgg.%d.%d.%s' % (e[0], i, k.split('.', 2)[2])] = v
                self._sem_ws = StateDictWeights(ren)
            elif _synthetic_ok() or str(self.cfg.det_ckpt).startswith('synthetic'):
                self._disp_ws, self._sem_ws = SynthWeights('disparity.'), SynthWeights('semantics.')
            else:
                raise FileNotFoundError("%s / %s (the reference downloads them with torch.hub / torchvision; set "
                                        "CSM_SYNTHETIC_WEIGHTS=1 for closed-form weights)" % (pd_, pv))
            self._disp_progs = {}
        self._depth_est = self._depth_est_default

    def _depth_est_default(self, img_tensor, img_d):
        """disparity_estimation (anime_3dkenburns/models/__init__.py:43-52): bilinear resize to <= 512, VGG19-BN semantics, GridNet;
        returns the disparity at HALF that resolution (depth_adjustment / Refine bring it back, kenburns_effect.py:49-52, :619-622)"""
        if img_tensor is None:
            img_tensor = ops.image_tensor(img_d)                                    # permute(2, 0, 1)[None].float() * (1.0 / 255.0)
        H, W = int(img_tensor.shape[2]), int(img_tensor.shape[3])
        ratio = float(W) / float(H)
        w, h = min(int(512 * ratio), 512), min(int(512 / ratio), 512)
        x = ops.resize_bilinear(img_tensor, h, w)
        if (h, w) not in self._disp_progs:
            self._disp_progs[(h, w)] = (CompiledProgram(build_semantics(self._sem_ws, h, w), self.device),
                                        CompiledProgram(build_disparity(self._disp_ws, h, w), self.device))
        sp, dp = self._disp_progs[(h, w)]
        sb = next(b for b in sp.prog.bufs if b.ext == 1)
        sem = torch.empty((1, 512, sb.h, sb.w), dtype=torch.float32, device=self.device)
        sp.run(x, sem)
        out = torch.empty((1, 1, (h + 1) // 2, (w + 1) // 2), dtype=torch.float32, device=self.device)
        dp.run(x, sem, out)
        return out

    def set_depth_refinement(self, depth_refinement: str):
        """kenburns_effect.py:820-829: 'default' = the Ken Burns `Refine` net"""
        if depth_refinement != 'default':
            raise NotImplementedError('Invalid depth refinement: %s' % depth_refinement)
        if self._refine_ws is None:
            p = 'models/AnimeInstanceSegmentation/kenburns_depth_refinenet.ckpt'     # utils/constants.py:81
            if os.path.exists(p):
                self._refine_ws = StateDictWeights(torch.load(p, map_location='cpu', weights_only=False))
            elif _synthetic_ok() or str(self.cfg.det_ckpt).startswith('synthetic'):
                self._refine_ws = SynthWeights('refine.')
            else:
                raise FileNotFoundError(p)

    def refine_depth(self, img: torch.Tensor, disparity: torch.Tensor):
        """Refine.forward (anime_3dkenburns/models/disparity_refinement.py:97-126): normalise, net, de-normalise, threshold(0)"""
        if self._refine_ws is None:
            self.set_depth_refinement('default')
        H, W, h, w = img.shape[2], img.shape[3], disparity.shape[2], disparity.shape[3]
        if (H, W, h, w) not in self._refine_progs:
            self._refine_progs[(H, W, h, w)] = CompiledProgram(build_refine(self._refine_ws, H, W, h, w), self.device)
        ms_i, ms_d = ops.mean_std(img), ops.mean_std(disparity)              # statistics stay on the device
        ni, nd = ops.normalise(img, ms_i), ops.normalise(disparity, ms_d)
        out = torch.empty((1, 1, H, W), dtype=torch.float32, device=self.device)
        self._refine_progs[(H, W, h, w)].run(ni, nd, out)
        return ops.denormalise(out, ms_d, 2)                                  # * (std + 1e-7) + mean, threshold(0)

    def set_inpainting(self, inpainting: str):
        """kenburns_effect.py:425-440: 'default' = the Inpaint GridNet; 'ldm'/'patchmatch' call external services"""
        if inpainting != 'default':
            raise NotImplementedError("inpaint_type %r needs stable-diffusion-webui / libpatchmatch (out of scope, SURVEY 2.1)" % inpainting)
        self.inpaint_type = inpainting
        if getattr(self, '_inpaint_ws', None) is None:
            p = 'models/AnimeInstanceSegmentation/kenburns_inpaintnet.ckpt'          # utils/constants.py:82
            if os.path.exists(p):
                self._inpaint_ws = StateDictWeights(torch.load(p, map_location='cpu', weights_only=False))
            elif _synthetic_ok() or str(self.cfg.det_ckpt).startswith('synthetic'):
                self._inpaint_ws = SynthWeights('inpaint.')
            else:
                self._inpaint_ws = None         # raised lazily: inpainting is only needed by process_kenburns(inpaint=True)
            self._inpaint_progs = {}

    def _inpaint_programs(self, H, W):
        if self._inpaint_ws is None:
            raise FileNotFoundError('models/AnimeInstanceSegmentation/kenburns_inpaintnet.ckpt')
        if (H, W) not in self._inpaint_progs:
            self._inpaint_progs[(H, W)] = (CompiledProgram(build_inpaint_context(self._inpaint_ws, H, W), self.device),
                                           CompiledProgram(build_inpaint_grid(self._inpaint_ws, H, W), self.device))
        return self._inpaint_progs[(H, W)]

    def _inpaint(self, tenImage, tenDisparity, tenShift, objCommon, segmasks=None):
        """Inpaint.forward (anime_3dkenburns/models/pointcloud_inpainting.py:116-203): context convs -> forward splat of
        68 channels -> median-5 clean-up -> GridNet -> colour + disparity.  mean/std are scalar torch reductions."""
        W, H, f, b = objCommon['intWidth'], objCommon['intHeight'], objCommon['fltFocal'], objCommon['fltBaseline']
        ctx_p, grid_p = self._inpaint_programs(H, W)
        # Everything up to the splat depends on the RAW image / disparity only, and process_kenburns inpaints the same pair at two
        # shifts (kenburns_effect.py:441-453, :1003-1013): the point cloud, the normalisation and the context features (two 1024^2
        # convolutions + a 64-channel layout change) are computed once per (image, disparity) pair and reused -- same program,
        # same inputs, same bits as recomputing them.
        key = (tenImage.data_ptr(), tenDisparity.data_ptr(), tenImage._version, tenDisparity._version, H, W)
        shared = getattr(objCommon, '_inpaint_shared', None)
        if shared is None or shared[0] != key:
            _, _, pts, _ = ops.disparity_to_points(tenDisparity, f, b, eps=0.0000001)
            pts = pts.view(1, 3, -1)
            ms_i, ms_d = ops.mean_std(tenImage), ops.mean_std(tenDisparity)
            ni, nd = ops.normalise(tenImage, ms_i), ops.normalise(tenDisparity, ms_d)
            x = torch.cat([ni, nd], 1).contiguous()
            ctx = torch.empty((1, 64, H, W), dtype=torch.float32, device=self.device)
            ctx_p.run(x, ctx)
            feat = torch.cat([ni, nd, ctx], 1).view(1, 68, -1)
            shared = (key, pts, ms_i, ms_d, nd, feat)
            try:
                objCommon._inpaint_shared = shared
            except AttributeError:                                              # a plain dict config: no caching
                pass
        _, pts, ms_i, ms_d, nd, feat = shared
        ps = (pts + tenShift).contiguous()
        render, existing = ops.render_pointcloud(ps, feat, W, H, f, b)
        if segmasks is not None:
            s = torch.cat([segmasks, nd], 1).view(1, segmasks.shape[1] + 1, -1)
            segmasks, _ = ops.render_pointcloud(ps, s, W, H, f, b)
        existing = (existing > 0.0).float()
        existing = existing * ops.spatial_filter(existing, 'median-5')
        render = render * existing
        gin = torch.cat([render, existing], 1).contiguous()
        img = torch.empty((1, 3, H, W), dtype=torch.float32, device=self.device)
        dsp = torch.empty((1, 1, H, W), dtype=torch.float32, device=self.device)
        grid_p.run(gin, img, dsp)
        return {'tenExisting': existing, 'tenImage': ops.denormalise(img, ms_i, 1),          # * (std + 1e-7) + mean, clip(0, 1)
                'tenDisparity': ops.denormalise(dsp, ms_d, 2), 'segmasks': segmasks}       # ..., threshold(0)

    def inpaint(self, tenShift, tenPoints, objCommon: KenBurnsConfig, verbose: bool = False):
        """kenburns_effect.py:441-512 (inpaint_type 'default'): inpaint the view at `tenShift` and append the points that
        fill its holes to the cloud (N grows, data dependent)."""
        ins = objCommon.instances
        mask_with_ins = None
        if ins is not None and not ins.is_empty:
            m = ins.masks[0]
            for k in ins.masks[1:]:
                m = torch.logical_or(m, k)
            mask_with_ins = m.to(torch.float32).repeat(3, 1, 1).unsqueeze(0)
        f, b = objCommon['fltFocal'], objCommon['fltBaseline']
        o = self._inpaint(objCommon['tenRawImage'], objCommon['tenRawDisparity'], tenShift, objCommon, mask_with_ins)
        depth_t, _, pts, _ = ops.disparity_to_points(o['tenDisparity'], f, b, eps=0.0000001)      # :457-460
        pts = pts.view(1, 3, -1) - tenShift
        tenMask = (o['tenExisting'] == 0.0).view(1, 1, -1)
        m1, m3 = tenMask[0, 0], tenMask.repeat(1, 3, 1)
        objCommon.inpainted_img = torch.cat([objCommon.inpainted_img, o['tenImage'].view(1, 3, -1)[m3].view(1, 3, -1)], 2)
        objCommon['tenInpaDisparity'] = torch.cat([objCommon['tenInpaDisparity'], o['tenDisparity'].view(1, 1, -1)[:, :, m1]], 2)
        objCommon['tenInpaDepth'] = torch.cat([objCommon['tenInpaDepth'], depth_t.view(1, 1, -1)[:, :, m1]], 2)
        objCommon['tenInpaPoints'] = torch.cat([objCommon['tenInpaPoints'], pts[m3].view(1, 3, -1)], 2)
        if not hasattr(objCommon, 'stage_inpainted_imgs') or objCommon.stage_inpainted_imgs is None:
            objCommon.stage_inpainted_imgs, objCommon.stage_inpainted_masks = [], []
        if verbose:
            objCommon.stage_inpainted_imgs.append((o['tenImage'][0] * 255).to(torch.uint8).permute(1, 2, 0).cpu().numpy())
            objCommon.stage_inpainted_masks.append((tenMask.view(objCommon.int_height, objCommon.int_width).to(torch.uint8) * 255).cpu().numpy())
        return o

    # ---- depth (kenburns_effect.py:563-581) ------------------------------------------------------------
    def _leres_prog(self, h, w, n=1, slot=0):
        """slot: programs that may run concurrently on different streams need their own workspace (weights are shared)"""
        if (h, w, n, slot) not in self._leres:
            cp = CompiledProgram(build_leres(self._leres_ws, n, h, w), self.device, shared=self._leres_weights)
            self._leres[(h, w, n, slot)] = cp
        return self._leres[(h, w, n, slot)]

    def _depth_est_leres_batch(self, imgs_d, slot=0):
        """LeReS on several equally sized frames in one program run (per-sample results as _depth_est_leres)"""
        L = _lib.load()
        nb = len(imgs_d)
        H, W = int(imgs_d[0].shape[0]), int(imgs_d[0].shape[1])
        h, w = scaledown_size(H, W, self.cfg.depth_est_size)
        h, w = int(math.ceil(h / 32) * 32), int(math.ceil(w / 32) * 32)
        x = torch.empty((nb, 3, h, w), dtype=torch.float32, device=self.device)
        for bi, im in enumerate(imgs_d):
            check(L.csm_leres_input(ptr(im), i32(H), i32(W), i32(h), i32(w), ptr(x[bi]), stream_ptr()), "leres_input")
        y = torch.empty((nb, 1, h, w), dtype=torch.float32, device=self.device)
        self._leres_prog(h, w, nb, slot).run(x, y)
        outs = []
        for bi in range(nb):
            yb = y[bi]
            mnmx = torch.empty(2, dtype=torch.float32, device=self.device)
            check(L.csm_minmax(ptr(yb), i64(h * w), ptr(mnmx), ptr(_mm_scratch(yb)), stream_ptr()), "minmax")
            q = torch.empty((h, w), dtype=torch.uint8, device=self.device)
            check(L.csm_leres_quantize(ptr(yb), i64(h * w), ptr(mnmx), ptr(q), stream_ptr()), "leres_quantize")
            depth = torch.empty((1, 1, H, W), dtype=torch.float32, device=self.device)
            self._leres_resize_back(q, h, w, H, W, depth)
            outs.append(_fill_zero_with_min_positive(depth))
        return outs

    def _depth_est_leres(self, img_tensor, img_d):
        """img_d: uint8 BGR HWC device tensor -> 'depth' (inverse-depth like, 1..255) fp32 [1,1,H,W]"""
        L = _lib.load()
        H, W = int(img_d.shape[0]), int(img_d.shape[1])
        h, w = scaledown_size(H, W, self.cfg.depth_est_size)
        h, w = int(math.ceil(h / 32) * 32), int(math.ceil(w / 32) * 32)
        x = torch.empty((1, 3, h, w), dtype=torch.float32, device=self.device)
        check(L.csm_leres_input(ptr(img_d), i32(H), i32(W), i32(h), i32(w), ptr(x), stream_ptr()), "leres_input")
        y = torch.empty((1, 1, h, w), dtype=torch.float32, device=self.device)
        self._leres_prog(h, w).run(x, y)
        mnmx = torch.empty(2, dtype=torch.float32, device=self.device)
        check(L.csm_minmax(ptr(y), i64(h * w), ptr(mnmx), ptr(_mm_scratch(y)), stream_ptr()), "minmax")
        q = torch.empty((h, w), dtype=torch.uint8, device=self.device)
        check(L.csm_leres_quantize(ptr(y), i64(h * w), ptr(mnmx), ptr(q), stream_ptr()), "leres_quantize")
        depth = torch.empty((1, 1, H, W), dtype=torch.float32, device=self.device)
        self._leres_resize_back(q, h, w, H, W, depth)
        return _fill_zero_with_min_positive(depth)

    @staticmethod
    def _leres_resize_back(q, h, w, H, W, depth):
        """kenburns_effect.py:571-573: k = depth.shape[0] / ori_h; cv2.resize(depth, (ori_w, ori_h), INTER_LANCZOS4 if k > 1 else
        INTER_AREA).  k > 1 happens when the 32-aligned LeReS size exceeds the frame (e.g. 600 x 400 -> 608 x 416)."""
        L = _lib.load()
        if h / H > 1:
            check(L.csm_resize_u8_lanczos4_to_f32(ptr(q), i32(h), i32(w), i32(H), i32(W), ptr(depth), stream_ptr()), "resize_lanczos4")
        else:
            check(L.csm_resize_u8_to_f32(ptr(q), i32(h), i32(w), i32(H), i32(W), ptr(depth), stream_ptr()), "resize_u8")

    def run_instance_segmentation(self, img, scale_down_to_maxsize=True):
        if scale_down_to_maxsize:                                                                                    # :862-863
            from utils.io_utils import scaledown_maxsize
            img = scaledown_maxsize(img, self.cfg.max_size)
        inst = self.animeinsseg.infer(img, self.cfg.pred_score_thr, self.cfg.mask_refine_kwargs or None, output_type='tensor',
                                      max_instances=self.max_instances)                                              # :869-872
        return inst, img

    def infer_disparity(self, img, instances=None, img_tensor=None, kcfg=None, coarse=None, **kw):
        img_d = self.animeinsseg._upload(img)
        if instances is None:
            instances, _ = self.run_instance_segmentation(img, scale_down_to_maxsize=False)
        if img_tensor is None:
            img_tensor = ops.image_tensor(img_d)                                    # permute(2, 0, 1)[None].float() * (1.0 / 255.0)
        disparity = self._depth_est(img_tensor, img_d) if coarse is None else coarse
        verbose = kw.get('verbose', False) and kcfg is not None
        if verbose:
            kcfg.stage_depth_coarse = colorize_depth(disparity.cpu().numpy(), inverse=True, rgb2bgr=True)
        disparity = depth_adjustment_animesseg(instances, disparity, img_tensor, self.cfg.depthest_use_medium)
        if verbose:
            kcfg.stage_depth_adjusted = colorize_depth(disparity.cpu().numpy(), inverse=True, rgb2bgr=True)
        if self.cfg.default_depth_refine:                                        # kenburns_effect.py:619-622
            disparity = self.refine_depth(img_tensor, disparity)
        elif self.cfg.refine_crf:
            raise NotImplementedError("refine_crf=True needs cv2 / pydensecrf CPU heuristics (out of scope, SURVEY 2.1; off in the shipped yaml)")
        if verbose:
            kcfg.stage_depth_final = colorize_depth(disparity.cpu().numpy(), inverse=True, rgb2bgr=True)
        return disparity

    # ---- generate_kenburns_config (kenburns_effect.py:898-951) -----------------------------------------------
    def _scaled_frame(self, img_d):
        """kenburns_effect.py:917 `img = scaledown_maxsize(img, self.cfg.max_size)` on the device (cv2 INTER_LINEAR u8 restated)"""
        H, W = int(img_d.shape[0]), int(img_d.shape[1])
        h, w = scaledown_size(H, W, self.cfg.max_size)
        return img_d if (h, w) == (H, W) else ops.resize_u8_linear(img_d, h, w)

    def generate_kenburns_config(self, img, instances: AnimeInstances = None, verbose: bool = False, savep=None):
        if isinstance(img, str):                                   # kenburns_effect.py:909-910 mmcv.imread(img)
            from utils.io_utils import imread
            img = imread(img)
        with torch.no_grad():
            img_dev = self.animeinsseg._upload(img)
            # the frame the depth / warp stages work on (reference :917); the detector sees the full-size image (:914-915)
            frame_dev = self._scaled_frame(img_dev)
            coarse = None
            if instances is None and self.overlap_depth:
                # Segmentation (RTMDet + ISNet) and the depth CNN only share the input image: run LeReS on a second HIP stream
                # so its large GEMM-like layers fill the CUs that the detector's small feature maps leave idle.  Results are
                # identical to the sequential order of the reference (kenburns_effect.py:914-923).
                main = torch.cuda.current_stream(self.device)
                if self._side_stream is None:
                    self._side_stream = torch.cuda.Stream(self.device)
                side = self._side_stream
                side.wait_stream(main)
                with torch.cuda.stream(side):
                    coarse = self._depth_est(None, frame_dev)
                instances, _ = self.run_instance_segmentation(img_dev, scale_down_to_maxsize=False)
                main.wait_stream(side)
                coarse.record_stream(main)
            elif instances is None:
                instances, _ = self.run_instance_segmentation(img_dev, scale_down_to_maxsize=False)
            return self._config_from(img, instances, coarse, verbose, frame_dev=frame_dev)

    def generate_kenburns_configs(self, imgs, verbose: bool = False):
        """MI355X addition (the reference loops image by image, run_kenburns_batch.py:36-62): equally sized frames share one
        batched detector run, shared ISNet refine batches and one batched LeReS run; the per-frame glue is unchanged."""
        with torch.no_grad():
            imgs_d = [self.animeinsseg._upload(im) for im in imgs]
            frames_d = [self._scaled_frame(t) for t in imgs_d]
            seg = lambda: self.animeinsseg.infer(list(imgs_d), self.cfg.pred_score_thr, self.cfg.mask_refine_kwargs or None,
                                                 output_type='tensor', max_instances=self.max_instances)
            # the batched estimator exists for LeReS only; any other selected estimator runs frame by frame (same results as
            # generate_kenburns_config), still on the side stream when overlap is on
            batched_leres = self._depth_est == self._depth_est_leres
            def depth_of(group, slot):
                if batched_leres:
                    return self._depth_est_leres_batch(group, slot=slot)
                if self._depth_est == self._depth_est_zoe and len(group) > 1 and len({tuple(f.shape) for f in group}) == 1:
                    return self._depth_est_zoe_batch(group)
                return [self._depth_est(None, f) for f in group]
            if self.overlap_depth:
                # the depth CNN only needs the images: it runs on a second HIP stream while the detector / ISNet batches (and the
                # detector's one host sync) occupy the main stream, so kernel tails of one net are filled by the other.
                main = torch.cuda.current_stream(self.device)
                k = max(1, min(self.depth_streams, len(imgs_d)))
                while len(self._side_streams) < k:
                    self._side_streams.append(torch.cuda.Stream(self.device))
                per = (len(imgs_d) + k - 1) // k
                coarse = []
                for si in range(k):
                    grp = frames_d[si * per:(si + 1) * per]
                    if not grp:
                        continue
                    side = self._side_streams[si]
                    side.wait_stream(main)
                    with torch.cuda.stream(side):
                        coarse += depth_of(grp, si)
                insts = seg()
                for side in self._side_streams[:k]:
                    main.wait_stream(side)
                for c in coarse:
                    c.record_stream(main)
            else:
                insts = seg()
                coarse = depth_of(frames_d, 0)
            return [self._config_from(im, inst, c, verbose, frame_dev=f) for im, inst, c, f in zip(imgs, insts, coarse, frames_d)]

    def _config_from(self, img, instances, coarse, verbose=False, frame_dev=None):
        """kenburns_effect.py:917-951 from the scaled frame on: instances.resize, depth glue, point cloud"""
        if frame_dev is None:
            frame_dev = self._scaled_frame(self.animeinsseg._upload(img))
        H, W = int(frame_dev.shape[0]), int(frame_dev.shape[1])
        instances.resize(H, W)
        self.cfg.int_height, self.cfg.int_width = H, W
        img_tensor = ops.image_tensor(frame_dev)                                    # permute(2, 0, 1)[None].float() * (1.0 / 255.0)
        cfg = self.cfg.copy()
        disparity = self.infer_disparity(frame_dev, instances, img_tensor, kcfg=cfg, coarse=coarse, verbose=verbose)
        if tuple(frame_dev.shape) == tuple(img.shape):
            kept = img                                             # cfg.original_img_nparray: the caller's own array / tensor
        else:
            kept = frame_dev if isinstance(img, torch.Tensor) else frame_dev.cpu().numpy()
        return self._finish_config(cfg, kept, img_tensor, instances, disparity)

    def _finish_config(self, cfg, img, img_tensor, instances, disparity):
        """kenburns_effect.py:928-951"""
        # kenburns_effect.py:928-935 with the reductions on the device: raw {min,max} -> normalise -> points -> minMaxLoc of the
        # depth crop; ONE host read of six scalars (the reference syncs per scalar)
        L, dev = _lib.load(), disparity.device
        raw = disparity.contiguous()
        H, W = int(raw.shape[2]), int(raw.shape[3])
        mm = torch.empty(2, dtype=torch.float32, device=dev)
        nmax = torch.empty(1, dtype=torch.float32, device=dev)
        check(L.csm_minmax(ptr(raw), i64(raw.numel()), ptr(mm), ptr(_mm_scratch(raw)), stream_ptr()), "minmax")
        disparity = torch.empty_like(raw)
        check(L.csm_normalise_disparity(ptr(raw), i64(raw.numel()), ptr(mm), f32(self.cfg.baseline), ptr(disparity), ptr(nmax),
                                        stream_ptr()), "normalise")
        depth, valid, pts, unaltered = ops.disparity_to_points(disparity, cfg.focal, cfg.baseline, dmax=nmax)
        crop = depth[0, 0, 128:-128, 128:-128]                      # cv2.minMaxLoc(depth[128:-128,128:-128])
        keys = torch.empty(2, dtype=torch.int64, device=dev)
        out6 = torch.empty(6, dtype=torch.float64, device=dev)
        check(L.csm_depth_range_stats(ptr(mm), f32(self.cfg.baseline), ptr(depth), i32(H), i32(W), i32(128), i32(128),
                                      i32(H - 256), i32(W - 256), ptr(keys), ptr(out6), stream_ptr()), "depth_range_stats")
        st = out6.tolist()                                          # the one host sync of this stage
        cfg['fltDispmin'], cfg['fltDispmax'] = st[0], st[1]
        amin, amax, cw = int(st[4]), int(st[5]), crop.shape[1]
        cfg['objDepthrange'] = (st[2], st[3], (amin % cw, amin // cw), (amax % cw, amax // cw))
        cfg['tenRawImage'], cfg['tenRawDisparity'], cfg['tenRawDepth'] = img_tensor, disparity, depth
        cfg['tenRawPoints'], cfg['tenRawUnaltered'] = pts.view(1, 3, -1), unaltered.view(1, 3, -1)
        cfg.inpainted_img = img_tensor.view(1, 3, -1)
        cfg['tenInpaDisparity'], cfg['tenInpaDepth'] = disparity.view(1, 1, -1), depth.view(1, 1, -1)
        cfg['tenInpaPoints'] = cfg['tenRawPoints']
        cfg.instances, cfg.original_img_nparray = instances, img
        return cfg

    # ---- autozoom (kenburns_effect.py:953-977, common.py:86-142) ------------------------------------------------
    def process_autozoom(self, objSettings, objCommon):
        """common.py:86-142 through the batched coverage kernels (ops.process_autozoom)"""
        return ops.process_autozoom(objSettings, objCommon)

    def autozoom(self, cfg: KenBurnsConfig, verbose: bool = False, inpaint: bool = True):
        with torch.no_grad():
            objFrom = {'fltCenterU': cfg.int_width / 2.0, 'fltCenterV': cfg.int_height / 2.0,
                       'intCropWidth': int(math.floor(0.97 * cfg.int_width)), 'intCropHeight': int(math.floor(0.97 * cfg.int_height))}
            objTo = self.process_autozoom({'fltShift': 100.0, 'fltZoom': 1.25, 'objFrom': objFrom}, cfg)
            frames, _ = self.process_kenburns({'fltSteps': np.linspace(0.0, 1.0, cfg.num_frame).tolist(), 'objFrom': objFrom,
                                               'objTo': objTo, 'boolInpaint': True}, cfg, inpaint, verbose)
            return frames

    def _focal_end(self, depth_u8, ins):
        """kenburns_effect.py:1045-1056: the largest per-instance MEDIAN of the colourised depth (np.median: mean of the two middle
        order statistics for even counts), the plane the depth of field settles on.  Per-instance 256-bin histograms on the device
        (csm_masked_u8_median_max): no gather, no sort, ONE scalar read (the reference syncs twice per instance)."""
        L = _lib.load()
        masks = ins.masks if isinstance(ins.masks, torch.Tensor) else torch.as_tensor(ins.masks)
        masks = masks.to(self.device)
        masks = (masks if masks.dtype == torch.bool else masks > 0).contiguous().view(torch.uint8)
        n = int(masks.shape[0])
        d8 = depth_u8.contiguous()
        hist = torch.empty(n * 256, dtype=torch.int32, device=self.device)
        out = torch.empty(n + 1, dtype=torch.float32, device=self.device)
        check(L.csm_masked_u8_median_max(ptr(d8), ptr(masks), i32(n), i64(d8.numel()), ptr(hist), ptr(out), stream_ptr()), "masked_median")
        return float(out[n].item())

    # ---- frame loop (kenburns_effect.py:979-1081) -----------------------------------------------------------------
    def process_kenburns(self, objSettings, objCommon: KenBurnsConfig, inpaint: bool = True, verbose: bool = False,
                         to_numpy: bool = True):
        L = _lib.load()
        with torch.no_grad():
            W, H = objCommon['intWidth'], objCommon['intHeight']
            oF, oT = objSettings['objFrom'], objSettings['objTo']
            wf = ops.WarpFrame(H, W, self.device, keep_render=bool(objCommon.depth_field))
            focal_start, focal_end = 0, 255
            steps = objSettings['fltSteps']
            out = torch.empty((len(steps), H, W, 3), dtype=torch.uint8, device=self.device)
            pw, ph = max(oF['intCropWidth'], oT['intCropWidth']), max(oF['intCropHeight'], oT['intCropHeight'])
            if inpaint:                                                   # kenburns_effect.py:984-1012
                objCommon.inpainted_img = objCommon['tenRawImage'].view(1, 3, -1)
                objCommon['tenInpaDisparity'] = objCommon['tenRawDisparity'].view(1, 1, -1)
                objCommon['tenInpaDepth'] = objCommon['tenRawDepth'].view(1, 1, -1)
                objCommon['tenInpaPoints'] = objCommon['tenRawPoints'].view(1, 3, -1)
                for fltStep in [0.0, 1.0]:
                    fltFrom = 1.0 - fltStep
                    fltTo = 1.0 - fltFrom
                    su = ((fltFrom * oF['fltCenterU']) + (fltTo * oT['fltCenterU'])) - (W / 2.0)
                    sv = ((fltFrom * oF['fltCenterV']) + (fltTo * oT['fltCenterV'])) - (H / 2.0)
                    cwid = (fltFrom * oF['intCropWidth']) + (fltTo * oT['intCropWidth'])
                    d_from = objCommon['objDepthrange'][0]
                    d_to = d_from * (cwid / max(oF['intCropWidth'], oT['intCropWidth']))
                    shift = ops.shift_vector({'fltShiftU': su, 'fltShiftV': sv, 'fltDepthFrom': d_from, 'fltDepthTo': d_to}, objCommon)
                    tenShift = torch.tensor(shift, dtype=torch.float32).view(1, 3, 1).to(self.device)
                    self.inpaint(1.1 * tenShift, None, objCommon, verbose)
                objCommon._inpaint_shared = None                   # the features shared by the two passes (285 MB at 1024^2) are dead now
            pts, rgb, dep = objCommon['tenInpaPoints'].contiguous(), objCommon.inpainted_img.contiguous(), objCommon['tenInpaDepth'].contiguous()
            # MI355X: the frames of a video are independent (same cloud, different shift) and a frame is a chain of ~20 short,
            # latency-bound kernels (bin -> render -> holes -> percentiles -> bokeh passes -> crop), so consecutive frames go to
            # `frame_streams` HIP streams round-robin, each with its own warp scratch: frame k + 1's binning runs under frame k's hole
            # fill / bokeh tail.  Every frame's kernels and results are those of the one-stream loop (kenburns_effect.py:1015-1072).
            main = torch.cuda.current_stream(self.device)
            ns = max(1, min(self.frame_streams, len(steps)))
            while len(self._frame_streams) < ns:
                self._frame_streams.append(torch.cuda.Stream(self.device))
            def lane_wf(i):                                                      # warp scratch per lane, kept across videos of one size
                key = (H, W, bool(objCommon.depth_field), i)
                if key not in self._lane_wf:
                    self._lane_wf = {kk: v for kk, v in self._lane_wf.items() if kk[:2] == key[:2]}    # another frame size: drop the old sets
                    self._lane_wf[key] = ops.WarpFrame(H, W, self.device, keep_render=bool(objCommon.depth_field))
                return self._lane_wf[key]
            lanes = [(main if ns == 1 else self._frame_streams[i], wf if i == 0 else lane_wf(i)) for i in range(ns)]
            for k, fltStep in enumerate(steps):
                fltFrom = 1.0 - fltStep
                fltTo = 1.0 - fltFrom
                su = ((fltFrom * oF['fltCenterU']) + (fltTo * oT['fltCenterU'])) - (W / 2.0)
                sv = ((fltFrom * oF['fltCenterV']) + (fltTo * oT['fltCenterV'])) - (H / 2.0)
                cwid = (fltFrom * oF['intCropWidth']) + (fltTo * oT['intCropWidth'])
                d_from = objCommon['objDepthrange'][0]
                d_to = d_from * (cwid / max(oF['intCropWidth'], oT['intCropWidth']))
                shift = ops.shift_vector({'fltShiftU': su, 'fltShiftV': sv, 'fltDepthFrom': d_from, 'fltDepthTo': d_to}, objCommon)
                if k == 1 and ns > 1:
                    for st_, _ in lanes:                                         # frame 0 (focal-plane statistics) ran on the caller's stream
                        st_.wait_stream(main)
                st_k, wf_k = (main, lanes[0][1]) if (k == 0 or ns == 1) else lanes[k % ns]
                with torch.cuda.stream(st_k):
                    fused = wf_k.path == 'tiled' and objCommon.depth_factor == 1 and (k > 0 or not objCommon.depth_field)
                    if fused:
                        # the whole frame in one library call (csm_kenburns_frame): same kernels, a twelfth of the host work
                        dof = None
                        if objCommon.depth_field:                                     # kenburns_effect.py:1058-1067
                            focal_int = 1 / (1 + np.exp((0.5 - fltStep) * objCommon.dof_speed))
                            dof = (focal_int * focal_end + (1 - focal_int) * focal_start, 32, objCommon.lightness_factor)
                        wf_k.frame_into(out[k], pts, rgb, dep, objCommon['fltFocal'], objCommon['fltBaseline'], shift, ph, pw, W / 2.0, H / 2.0, dof)
                        continue
                    frame, render = wf_k(pts, rgb, dep, objCommon['fltFocal'], objCommon['fltBaseline'], shift)
                    if objCommon.depth_field:                                         # kenburns_effect.py:1042-1067
                        depth_u8 = ops.colorize_gray_r(render[0, 3])
                        if k == 0:
                            ins = objCommon.instances
                            if ins is not None and not ins.is_empty:
                                focal_end = self._focal_end(depth_u8, ins)
                                focal_start = 255 if abs(255 - focal_end) > abs(0 - focal_end) else 0
                        focal_int = 1 / (1 + np.exp((0.5 - fltStep) * objCommon.dof_speed))
                        focal_plane = focal_int * focal_end + (1 - focal_int) * focal_start
                        frame = ops.bokeh_blur(frame, depth_u8, 32, objCommon.lightness_factor, focal_plane=focal_plane, use_cuda=True,
                                               depth_factor=objCommon.depth_factor)
                    check(L.csm_crop_resize_u8(ptr(frame), i32(H), i32(W), i32(ph), i32(pw), f32(W / 2.0), f32(H / 2.0),
                                               ptr(out[k]), stream_ptr()), "crop_resize")
            if ns > 1:
                for st_, _ in lanes:
                    main.wait_stream(st_)
            frames = [f for f in out.cpu().numpy()] if to_numpy else out
            return [frames, objCommon]


def npyframes2video(npy_frame_list, video_save_path: str, playback: bool = False):
    """kenburns_effect.py:1086-1090 (BGR->RGB, optional ping-pong, 25 fps mp4 through moviepy)"""
    sequence = [f[:, :, ::-1] for f in npy_frame_list]
    if playback:
        sequence += sequence[::-1][1:-1]
    try:
        import moviepy.editor
    except ImportError as e:
        raise RuntimeError("npyframes2video needs moviepy/ffmpeg (video encoding is outside the hot path)") from e
    moviepy.editor.ImageSequenceClip(sequence=sequence, fps=25).write_videofile(video_save_path, preset="veryslow")
