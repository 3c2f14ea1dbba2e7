"""Torch-facing operators with the reference's signatures, backed by libcsm355.so.

Each function mirrors the reference operator cited in its docstring (paths relative to
/root/reference): same argument meaning, same returned tensors, Python exceptions on error.
Tensors must live on the MI355X; a CPU tensor raises (the reference hard-codes CUDA as well,
SURVEY F5).
"""
import torch

from . import _lib
from ._lib import check, f32, f64, i32, i64, ptr, stream_ptr


def _dev(t, name):
    if not t.is_cuda:
        raise _lib.CsmError("%s must be a device tensor (got %s); libcsm355 has no CPU path" % (name, t.device))
    if t.dtype != torch.float32:
        raise _lib.CsmError("%s must be float32" % name)
    return t.contiguous()


def pointrender_update_zee(tenInput, intWidth, intHeight, fltFocal, fltBaseline, tenZee=None):
    """kernel_pointrender_updateZee -- anime_3dkenburns/models/utils.py:63-149"""
    tenInput = _dev(tenInput, "tenInput")
    B, _, N = tenInput.shape
    if tenZee is None:
        tenZee = tenInput.new_full([B, 1, intHeight, intWidth], 1000000.0)
    check(_lib.load().csm_pointrender_update_zee(ptr(tenInput), i32(B), i64(N), i32(intHeight), i32(intWidth),
                                                 f64(fltFocal), f64(fltBaseline), ptr(tenZee), stream_ptr()), "update_zee")
    return tenZee


def pointrender_degrid(tenZee):
    """kernel_pointrender_updateDegrid (Jacobi form) -- models/utils.py:152-212"""
    tenZee = _dev(tenZee, "tenZee")
    B, _, H, W = tenZee.shape
    out = torch.empty_like(tenZee)
    check(_lib.load().csm_pointrender_degrid(ptr(tenZee), ptr(out), i32(B), i32(H), i32(W), stream_ptr()), "degrid")
    return out


def pointrender_update_output(tenInput, tenData, tenZee, fltFocal, fltBaseline):
    """kernel_pointrender_updateOutput -- models/utils.py:215-313; returns accum [B,C+1,H,W]"""
    tenInput, tenData, tenZee = _dev(tenInput, "tenInput"), _dev(tenData, "tenData"), _dev(tenZee, "tenZee")
    B, C, N = tenData.shape
    _, _, H, W = tenZee.shape
    acc = tenInput.new_zeros([B, C + 1, H, W])
    check(_lib.load().csm_pointrender_update_output(ptr(tenInput), ptr(tenData), ptr(tenZee), i32(B), i32(C), i64(N),
                                                    i32(H), i32(W), f64(fltFocal), f64(fltBaseline), ptr(acc),
                                                    stream_ptr()), "update_output")
    return acc


_RENDER_SCRATCH = {}


def _render_tile_scratch(dev, H, W, N):
    """scratch of the tiled render_pointcloud, one per (device, stream, frame size), grown with the cloud; header zeroed once"""
    key = (dev, torch.cuda.current_stream(dev).cuda_stream)            # one set per stream: another frame size replaces it
    have = _RENDER_SCRATCH.get(key)
    if have is None or have[1] < N or have[2] != (H, W):
        L = _lib.load()
        cap = int(N * 1.25) + 1024
        buf = torch.empty((L.csm_warp_tile_scratch_bytes(i32(H), i32(W), i64(cap)) + 3) // 4, dtype=torch.float32, device=dev)
        buf[:(L.csm_warp_tile_header_bytes(i32(H), i32(W)) + 3) // 4].zero_()
        _RENDER_SCRATCH[key] = have = (buf, cap, (H, W))
    return have[0]


def render_pointcloud(tenInput, tenData, intWidth, intHeight, fltFocal, fltBaseline, path=None):
    """render_pointcloud -- anime_3dkenburns/models/utils.py:56-315
    tenInput [B,3,N], tenData [B,C,N] -> (tenRender [B,C,H,W], tenExisting [B,1,H,W]).
    path 'tiled' (default for one cloud on frames of at most 8192 tiles): destination-tile binning + LDS splat in channel groups
    (csm_render_pointcloud_tiled, deterministic); 'atomics': the global-atomic chain (csm_render_pointcloud; CSM_RENDER_PATH selects)."""
    import os
    tenInput, tenData = _dev(tenInput, "tenInput"), _dev(tenData, "tenData")
    B, C, N = tenData.shape
    if tenInput.shape[0] != B or tenInput.shape[1] != 3 or tenInput.shape[2] != N:
        raise _lib.CsmError("render_pointcloud: tenInput must be [B,3,N] matching tenData [B,C,N]")
    L = _lib.load()
    path = path or os.environ.get('CSM_RENDER_PATH', 'tiled')
    assert path in ('tiled', 'atomics')
    render = tenInput.new_empty([B, C, intHeight, intWidth])
    existing = tenInput.new_empty([B, 1, intHeight, intWidth])
    if path == 'tiled' and B == 1 and L.csm_warp_tile_supported(i32(intHeight), i32(intWidth)):
        tenInput, tenData = tenInput.contiguous(), tenData.contiguous()
        check(L.csm_render_pointcloud_tiled(ptr(tenInput), ptr(tenData), i32(C), i64(N), i32(intWidth), i32(intHeight), f64(fltFocal),
                                            f64(fltBaseline), ptr(_render_tile_scratch(tenInput.device, intHeight, intWidth, N)),
                                            ptr(render), ptr(existing), stream_ptr()), "render_pointcloud_tiled")
        return render, existing
    zee = tenInput.new_empty([2, B, intHeight, intWidth])
    acc = tenInput.new_empty([B, C + 1, intHeight, intWidth])
    check(L.csm_render_pointcloud(ptr(tenInput), ptr(tenData), i32(B), i32(C), i64(N), i32(intWidth),
                                  i32(intHeight), f64(fltFocal), f64(fltBaseline), ptr(zee), ptr(acc),
                                  ptr(render), ptr(existing), stream_ptr()), "render_pointcloud")
    return render, existing


def fill_disocclusion(tenInput, tenDepth):
    """fill_disocclusion -- anime_3dkenburns/common.py:145-248"""
    tenInput, tenDepth = _dev(tenInput, "tenInput"), _dev(tenDepth, "tenDepth")
    B, C, H, W = tenInput.shape
    out = torch.empty_like(tenInput)
    nbytes = _lib.load().csm_fill_disocclusion_scratch_bytes(i32(B), i32(H), i32(W))
    scratch = torch.empty(nbytes, dtype=torch.uint8, device=tenInput.device)
    check(_lib.load().csm_fill_disocclusion(ptr(tenInput), ptr(tenDepth), ptr(out), i32(B), i32(C), i32(H), i32(W),
                                            ptr(scratch), stream_ptr()), "fill_disocclusion")
    return out


def spatial_filter(tenInput, strType):
    """spatial_filter -- anime_3dkenburns/models/utils.py:9-40: 'laplacian' (the hot-path mode), 'median-3', 'median-5'"""
    L = _lib.load()
    fns = {'laplacian': L.csm_spatial_filter_laplacian, 'median-3': L.csm_spatial_filter_median3, 'median-5': L.csm_spatial_filter_median5}
    if strType not in fns:
        raise ValueError("spatial_filter(%r): the reference knows 'laplacian', 'median-3' and 'median-5' (any other type leaves its "
                         "tenOutput None)" % (strType,))
    tenInput = _dev(tenInput, "tenInput")
    B, C, H, W = tenInput.shape
    out = torch.empty_like(tenInput)
    fn = fns[strType]
    check(fn(ptr(tenInput), ptr(out), i32(B * C), i32(H), i32(W), stream_ptr()), "spatial_filter")
    return out


def depth_to_points(tenDepth, fltFocal):
    """depth_to_points -- anime_3dkenburns/models/utils.py:43-50"""
    tenDepth = _dev(tenDepth, "tenDepth")
    B, _, H, W = tenDepth.shape
    pts = tenDepth.new_empty([B, 3, H, W])
    check(_lib.load().csm_depth_to_points(ptr(tenDepth), ptr(pts), i32(B), i32(H), i32(W), f64(fltFocal), stream_ptr()),
          "depth_to_points")
    return pts


def disparity_to_points(tenDisparity, fltFocal, fltBaseline, eps=0.00001, dmax=None):
    """kenburns_effect.py:929-933 fused: normalised disparity -> depth, valid, points, unaltered.
    dmax: optional 1-element device tensor holding max(tenDisparity) when the caller already has it"""
    d = _dev(tenDisparity, "tenDisparity")
    H, W = d.shape[-2:]
    depth, valid = torch.empty_like(d), torch.empty_like(d)
    pts, un = d.new_empty([1, 3, H, W]), d.new_empty([1, 3, H, W])
    if dmax is None:
        dmax = d.max().reshape(1)                                              # stays on the device: no host sync
    check(_lib.load().csm_disparity_to_points(ptr(d), ptr(dmax), i32(H), i32(W), f64(fltFocal),
                                              f64(fltBaseline), f32(eps), ptr(depth), ptr(valid), ptr(pts), ptr(un), stream_ptr()),
          "disparity_to_points")
    return depth, valid, pts, un


def shift_vector(objSettings, objCommon):
    """scalar part of process_shift -- anime_3dkenburns/common.py:60-72 (python floats, like the reference)"""
    cd = objCommon['objDepthrange'][0] + (objSettings['fltDepthTo'] - objSettings['fltDepthFrom'])
    fu, fv = objCommon['objDepthrange'][2][0], objCommon['objDepthrange'][2][1]
    tu, tv = fu + objSettings['fltShiftU'], fv + objSettings['fltShiftV']
    w2, h2, f = objCommon['intWidth'] / 2.0, objCommon['intHeight'] / 2.0, objCommon['fltFocal']
    fx, fy = ((fu - w2) * cd) / f, ((fv - h2) * cd) / f
    tx, ty = ((tu - w2) * cd) / f, ((tv - h2) * cd) / f
    return [fx - tx, fy - ty, objSettings['fltDepthTo'] - objSettings['fltDepthFrom']]


def _f32x3(shift):
    # FloatTensor rounding of the python floats (common.py:74): float64 -> float32, round to nearest even == numpy's conversion
    return f32(float(_np.float32(shift[0]))), f32(float(_np.float32(shift[1]))), f32(float(_np.float32(shift[2])))


def shift_points(tenPoints, shift):
    """tensor part of process_shift -- common.py:74-81; shift = 3 python floats"""
    tenPoints = _dev(tenPoints, "tenPoints")
    B, _, N = tenPoints.shape
    sx, sy, sz = _f32x3(shift)
    out = torch.empty_like(tenPoints)
    check(_lib.load().csm_process_shift(ptr(tenPoints), ptr(out), i32(B), i64(N), sx, sy, sz, stream_ptr()), "process_shift")
    return out


def process_shift(objSettings, objCommon):
    """process_shift -- anime_3dkenburns/common.py:59-84 -> (tenPoints, tenShift)"""
    shift = shift_vector(objSettings, objCommon)
    pts = objSettings['tenPoints']
    tenShift = torch.tensor(shift, dtype=torch.float32).view(1, 3, 1).to(pts.device)
    return shift_points(pts, shift), tenShift


def resize_u8_linear(img, h, w):
    """cv2.resize(img, (w, h), interpolation=cv2.INTER_LINEAR) of a uint8 HxWxC (or HxW) device tensor -- the resampler of
    utils/io_utils.py:254-274 scaledown_maxsize"""
    if not img.is_cuda or img.dtype != torch.uint8:
        raise _lib.CsmError("resize_u8_linear: uint8 device tensor expected")
    img = img.contiguous()
    C = 1 if img.dim() == 2 else int(img.shape[2])
    out = torch.empty((h, w) if img.dim() == 2 else (h, w, C), dtype=torch.uint8, device=img.device)
    check(_lib.load().csm_resize_u8_linear(ptr(img), i32(img.shape[0]), i32(img.shape[1]), i32(C), i32(h), i32(w), ptr(out),
                                           stream_ptr()), "resize_u8_linear")
    return out


def resize_f32_linear(img, h, w):
    """cv2.resize(img, (w, h), interpolation=cv2.INTER_LINEAR) of a float32 HxWxC (or HxW) device tensor (float masks through
    utils/io_utils.py:254-292, animeinsseg/__init__.py:47)"""
    if not img.is_cuda or img.dtype != torch.float32:
        raise _lib.CsmError("resize_f32_linear: float32 device tensor expected")
    img = img.contiguous()
    C = 1 if img.dim() == 2 else int(img.shape[2])
    out = torch.empty((h, w) if img.dim() == 2 else (h, w, C), dtype=torch.float32, device=img.device)
    check(_lib.load().csm_resize_f32_linear(ptr(img), i32(img.shape[0]), i32(img.shape[1]), i32(C), i32(h), i32(w), ptr(out),
                                            stream_ptr()), "resize_f32_linear")
    return out


def resize_bilinear(x, h, w, align_corners=False):
    """torch.nn.functional.interpolate(x, size=(h, w), mode='bilinear', align_corners=...) of a float32 NCHW device tensor (aten
    upsample_bilinear2d restated: csm_resize_bilinear_planes)"""
    x = _dev(x, "x")
    N, C, H, W = x.shape
    out = torch.empty((N, C, h, w), dtype=torch.float32, device=x.device)
    check(_lib.load().csm_resize_bilinear_planes(ptr(x), i32(N * C), i32(H), i32(W), i32(h), i32(w), i32(1 if align_corners else 0), ptr(out),
                                                 stream_ptr()), "resize_bilinear_planes")
    return out


def mean_std(x):
    """device tensor {x.mean(), x.std(unbiased=False)} over all elements (the statistics of Inpaint.forward / Refine.forward)"""
    x = _dev(x, "x")
    out = torch.empty(2, dtype=torch.float32, device=x.device)
    scratch = torch.empty(_lib.load().csm_mean_std_scratch_bytes(), dtype=torch.uint8, device=x.device)
    check(_lib.load().csm_mean_std(ptr(x), i64(x.numel()), ptr(out), ptr(scratch), stream_ptr()), "mean_std")
    return out


def normalise(x, ms):
    """(x - mean) / (std + 1e-7) with ms = mean_std(...)"""
    x = _dev(x, "x")
    out = torch.empty_like(x)
    check(_lib.load().csm_normalise_mean_std(ptr(x), i64(x.numel()), ptr(ms), ptr(out), stream_ptr()), "normalise_mean_std")
    return out


def denormalise(x, ms, mode=0):
    """x * (std + 1e-7) + mean; mode 1: .clip(0, 1), mode 2: threshold(0, 0)"""
    x = _dev(x, "x")
    out = torch.empty_like(x)
    check(_lib.load().csm_denormalise_mean_std(ptr(x), i64(x.numel()), ptr(ms), i32(mode), ptr(out), stream_ptr()), "denormalise_mean_std")
    return out


def autozoom_coverage(tenPoints, shifts, intWidth, intHeight, fltFocal, fltBaseline, chunk=None, host=True):
    """coverage counts `(tenExisting > 0.0).float().sum()` of render_pointcloud(process_shift(tenPoints, shift_k)) for every
    candidate shift_k = (sx, sy, sz) -- common.py:110-126 -- in batched launches (csm_autozoom_coverage), no colour rendered,
    ONE host read.  All candidates of one search share sz (common.py:92-93).  Returns the K counts as a Python list (host=True,
    the default: the read that replaces the reference's <= 256 `.item()` syncs), or a device tensor (host=False; on the band path
    it has K + 1 entries, the last one being the overflow flag of csm_autozoom_coverage_bands)."""
    import ctypes
    import os
    tenPoints = _dev(tenPoints, "tenPoints")
    assert tenPoints.shape[0] == 1 and tenPoints.shape[1] == 3
    L, K = _lib.load(), len(shifts)
    counts = torch.zeros(max(K, 1), dtype=torch.int32, device=tenPoints.device)
    if K == 0:
        return [] if host else counts[:0]
    # FloatTensor rounding of the python-float shifts (common.py:74): float64 -> float32, round to nearest even == numpy's astype
    s32 = _np.asarray([[float(s[0]), float(s[1]), float(s[2])] for s in shifts], dtype=_np.float64).astype(_np.float32)
    sz = {float(v) for v in s32[:, 2]}
    assert len(sz) == 1, "the candidates of one autozoom search share the z shift"
    xy = (ctypes.c_float * (2 * K)).from_buffer_copy(_np.ascontiguousarray(s32[:, :2]).tobytes())
    path = os.environ.get('CSM_AUTOZOOM_PATH', 'bands')
    if chunk is None and path == 'bands' and L.csm_autozoom_band_supported(i32(intHeight), i32(intWidth)):
        # band path: z-buffers in LDS, candidates grouped by y shift.  The overflow flag travels with the counts (one transfer).
        N = tenPoints.shape[2]
        out = torch.empty(K + 1, dtype=torch.int32, device=tenPoints.device)
        scratch = torch.empty((L.csm_autozoom_band_scratch_bytes(i32(intHeight), i32(intWidth), i64(N)) + 3) // 4, dtype=torch.float32,
                              device=tenPoints.device)
        zs = f32(sz.pop())
        check(L.csm_autozoom_coverage_bands(ptr(tenPoints), i64(N), i32(intHeight), i32(intWidth), f64(fltFocal), f64(fltBaseline), xy,
                                            zs, i32(K), ptr(scratch), ptr(out), ctypes.c_void_p(out.data_ptr() + 4 * K), stream_ptr()),
              "autozoom_coverage_bands")
        if not host:
            return out                                      # [K + 1]: counts, overflow flag (callers that stay on the device check it)
        vals = out.tolist()
        if vals[K] == 0:
            return vals[:K]
        sz = {zs.value}                                     # a band segment overflowed (pathological cloud): exact plane path below
    chunk = int(chunk or os.environ.get('CSM_AUTOZOOM_CHUNK', L.csm_autozoom_max_chunk()))
    chunk = max(1, min(chunk, L.csm_autozoom_max_chunk(), K))
    scratch = torch.empty(L.csm_autozoom_scratch_floats(i32(intHeight), i32(intWidth), i32(chunk)), dtype=torch.float32,
                          device=tenPoints.device)
    check(L.csm_autozoom_coverage(ptr(tenPoints), i64(tenPoints.shape[2]), i32(intHeight), i32(intWidth), f64(fltFocal),
                                  f64(fltBaseline), xy, f32(sz.pop()), i32(K), i32(chunk), ptr(scratch), ptr(counts), stream_ptr()),
          "autozoom_coverage")
    return counts[:K].tolist() if host else counts[:K]


def process_autozoom(objSettings, objCommon, return_counts=False):
    """process_autozoom -- anime_3dkenburns/common.py:86-142: the 16 x 16 grid of candidate shifts whose crop stays inside the
    image, the one with the largest rendered coverage wins (first strictly-better candidate, like the reference's `<`).
    MI355X: all candidates in batched launches + ONE host read instead of <= 256 x (3 kernels + `.item()`)."""
    import numpy as np
    shift = objSettings['fltShift']
    lin = np.linspace(-shift, shift, 16)
    oF = objSettings['objFrom']
    cw, ch = oF['intCropWidth'] / objSettings['fltZoom'], oF['intCropHeight'] / objSettings['fltZoom']
    d_from = objCommon['objDepthrange'][0]
    d_to = objCommon['objDepthrange'][0] * (cw / oF['intCropWidth'])
    cu, cv = oF['fltCenterU'], oF['fltCenterV']
    W, H = objCommon['intWidth'], objCommon['intHeight']
    cands = []
    for iu in range(16):
        for iv in range(16):
            su, sv = lin[iv].item(), lin[iu].item()        # npyShiftU[intU, intV] = lin[intV]; npyShiftV[intU, intV] = lin[intU]
            if cu + su < cw / 2.0 or cu + su > W - (cw / 2.0) or cv + sv < ch / 2.0 or cv + sv > H - (ch / 2.0):
                continue
            cands.append((su, sv))
    shifts = [shift_vector({'fltShiftU': su, 'fltShiftV': sv, 'fltDepthFrom': d_from, 'fltDepthTo': d_to}, objCommon) for su, sv in cands]
    counts = autozoom_coverage(objCommon['tenRawPoints'], shifts, W, H, objCommon['fltFocal'], objCommon['fltBaseline'])
    best, bu, bv = 0.0, None, None
    for (su, sv), c in zip(cands, counts):
        if best < c:
            best, bu, bv = float(c), su, sv
    out = {'fltCenterU': cu + bu, 'fltCenterV': cv + bv,
           'intCropWidth': int(round(oF['intCropWidth'] / objSettings['fltZoom'])),
           'intCropHeight': int(round(oF['intCropHeight'] / objSettings['fltZoom']))}
    return (out, cands, counts) if return_counts else out


class WarpFrame:
    """Fused per-frame warp of KenBurnsPipeline.process_kenburns (kenburns_effect.py:1027-1040):
    process_shift -> render_pointcloud(cat[rgb,depth]) -> fill_disocclusion -> uint8 HWC.
    Owns the scratch so the frame loop allocates nothing.  path 'tiled' (default): destination-tile binning + LDS splat
    (csm_warp_frame_tiled); 'atomics': the global-atomic chain of csm_warp_frame (CSM_WARP_PATH selects)."""

    def __init__(self, H, W, device, keep_render=False, path=None):
        import os
        self.H, self.W, self.device = H, W, device
        self.path = path or os.environ.get('CSM_WARP_PATH', 'tiled')
        assert self.path in ('tiled', 'atomics')
        if self.path == 'tiled' and not _lib.load().csm_warp_tile_supported(i32(H), i32(W)):
            self.path = 'atomics'                          # more than 8192 tiles (e.g. 3840 x 2160): the global-atomic chain has no limit
        self.frame = torch.empty((H, W, 3), dtype=torch.uint8, device=device)
        self.render = torch.empty((1, 4, H, W), dtype=torch.float32, device=device) if keep_render else None
        self.scratch, self._cap = None, -1
        if self.path == 'atomics':
            n = _lib.load().csm_warp_frame_scratch_floats(i32(H), i32(W))
            self.scratch = torch.empty(n, dtype=torch.float32, device=device)

    def _tile_scratch(self, N):
        if N > self._cap:                                  # the point cloud grows when inpainting appends points
            L = _lib.load()
            cap = int(N * 1.25) + 1024
            nbytes = L.csm_warp_tile_scratch_bytes(i32(self.H), i32(self.W), i64(cap))
            self.scratch = torch.empty((nbytes + 3) // 4, dtype=torch.float32, device=self.device)
            hdr = (L.csm_warp_tile_header_bytes(i32(self.H), i32(self.W)) + 3) // 4
            self.scratch[:hdr].zero_()                     # the bin counters must start at zero; every frame re-arms them
            self._cap = cap
        return self.scratch

    def __call__(self, tenPoints, tenImage, tenDepth, fltFocal, fltBaseline, shift, stream=None):
        N = tenPoints.shape[2]
        sx, sy, sz = _f32x3(shift)
        st = stream_ptr() if stream is None else stream
        if self.path == 'tiled':
            check(_lib.load().csm_warp_frame_tiled(ptr(tenPoints), ptr(tenImage), ptr(tenDepth), i64(N), i32(self.H), i32(self.W),
                                                   f64(fltFocal), f64(fltBaseline), sx, sy, sz, ptr(self._tile_scratch(N)),
                                                   ptr(self.render), ptr(self.frame), st), "warp_frame_tiled")
        else:
            check(_lib.load().csm_warp_frame(ptr(tenPoints), ptr(tenImage), ptr(tenDepth), i64(N), i32(self.H), i32(self.W),
                                             f64(fltFocal), f64(fltBaseline), sx, sy, sz, ptr(self.scratch),
                                             ptr(self.render), ptr(self.frame), st), "warp_frame")
        return self.frame, self.render

    def frames(self, tenPoints, tenImage, tenDepth, fltFocal, fltBaseline, shifts, lanes=3, out=None, stream=None):
        """K frames of one cloud in ONE call (csm_warp_frames_tiled): shifts = K x (sx, sy, sz); returns uint8 [K, H, W, 3].  The frames are
        bit-identical to K __call__s; frame k + 1's binning and frame k - 1's hole fill overlap frame k's render on internal streams."""
        import ctypes
        assert self.path == 'tiled'
        L = _lib.load()
        N = tenPoints.shape[2]
        K = len(shifts)
        flat = []
        for s in shifts:
            flat.extend(float(v.value) for v in _f32x3(s))
        arr = (ctypes.c_float * (3 * max(K, 1)))(*flat)
        st = stream_ptr() if stream is None else stream
        if getattr(self, '_multi_key', None) is None or self._multi_key[0] < N or self._multi_key[1] != lanes:
            cap = int(N * 1.25) + 1024                         # (the point cloud grows when inpainting appends points)
            nbytes = L.csm_warp_frames_scratch_bytes(i32(self.H), i32(self.W), i64(cap), i32(lanes))
            self._multi = torch.zeros((nbytes + 3) // 4, dtype=torch.float32, device=self.device)     # (headers must start at zero)
            self._multi_key = (cap, lanes)
            self._multi_n = N
        elif self._multi_n != N:
            # the library carves lane i at i * round256(csm_warp_tile_scratch_bytes(H, W, N)) with the CURRENT N: when N changes, the
            # headers of lanes 1.. fall into bytes the previous call used as depth / entry storage, and the tile protocol needs every
            # header zero on entry -> clear the buffer on the stream the call runs on
            if stream is None:
                self._multi.zero_()
            else:
                with torch.cuda.stream(torch.cuda.ExternalStream(int(stream.value or 0), device=self.device)):
                    self._multi.zero_()
            self._multi_n = N
        if out is None:
            out = torch.empty((K, self.H, self.W, 3), dtype=torch.uint8, device=self.device)
        check(L.csm_warp_frames_tiled(ptr(tenPoints), ptr(tenImage), ptr(tenDepth), i64(N), i32(self.H), i32(self.W), f64(fltFocal),
                                      f64(fltBaseline), arr, i32(K), i32(lanes), ptr(self._multi), ptr(None), ptr(out), st), "warp_frames_tiled")
        return out

    def frame_into(self, out_hwc, tenPoints, tenImage, tenDepth, fltFocal, fltBaseline, shift, patch_h, patch_w, center_x, center_y,
                   dof=None):
        """One output frame of the video loop in ONE library call (csm_kenburns_frame, kenburns_effect.py:1027-1072): warp
        [-> colourised depth -> depth-of-field blur with dof = (focal_plane, num_samples, lightness_factor)] -> crop + resize into
        `out_hwc`.  Same kernels and bits as __call__ + colorize_gray_r + bokeh_blur + csm_crop_resize_u8; tiled path only."""
        import ctypes
        global _GRAY_R_LUT
        assert self.path == 'tiled'
        L = _lib.load()
        if dof is not None:
            assert self.render is not None, "depth of field needs WarpFrame(keep_render=True)"
            if _GRAY_R_LUT is None:
                lut = ((1.0 - _np.linspace(0.0, 1.0, 256)) * 255).astype(_np.uint8)
                _GRAY_R_LUT = (ctypes.c_uint8 * 256)(*[int(x) for x in lut])
            if getattr(self, '_tail', None) is None:
                self._tail = torch.zeros(L.csm_kenburns_frame_scratch_bytes(i32(self.H), i32(self.W)), dtype=torch.uint8, device=self.device)
        N = tenPoints.shape[2]
        sx, sy, sz = _f32x3(shift)
        fp, ns, lf = (0.0, 32, 1.0) if dof is None else (float(_np.float32(dof[0])), int(dof[1]), float(dof[2]))
        check(L.csm_kenburns_frame(ptr(tenPoints), ptr(tenImage), ptr(tenDepth), i64(N), i32(self.H), i32(self.W), f64(fltFocal),
                                   f64(fltBaseline), sx, sy, sz, ptr(self._tile_scratch(N)), ptr(self.render), ptr(self.frame),
                                   i32(0 if dof is None else 1), f32(fp), i32(ns), f32(lf), _GRAY_R_LUT if dof is not None else None,
                                   ptr(getattr(self, '_tail', None)), i32(patch_h), i32(patch_w), f32(center_x), f32(center_y),
                                   ptr(out_hwc), stream_ptr()), "kenburns_frame")


# ---- bokeh depth-of-field (utils/effects.py:143-181, depth_modules/zoedepth/utils/misc.py:97-135) ---------------------
import math as _math

import numpy as _np

_GRAY_R_LUT = None


def _percentile_linear(sorted_vals, q):
    """np.percentile(..., method='linear') on an ascending device tensor (two element reads)"""
    n = sorted_vals.numel()
    vi = (n - 1) * (q / 100.0)
    lo = int(_math.floor(vi)); hi = min(lo + 1, n - 1)
    a, b = float(sorted_vals[lo].item()), float(sorted_vals[hi].item())
    t = vi - lo
    r = a + (b - a) * t if t < 0.5 else b - (b - a) * (1 - t)          # numpy _lerp
    return float(_np.float32(r))


_TAIL_SCRATCH = {}


def image_tensor(img_hwc_u8):
    """uint8 HxWx3 device image -> float32 [1,3,H,W] in [0,1] (`img.permute(2, 0, 1)[None].float() * (1.0 / 255.0)`), one kernel"""
    img = img_hwc_u8
    if not (isinstance(img, torch.Tensor) and img.is_cuda and img.dtype == torch.uint8 and img.dim() == 3 and img.shape[2] == 3):
        raise _lib.CsmError("image_tensor: a uint8 HxWx3 device tensor is required; libcsm355 has no CPU path")
    img = img.contiguous()
    H, W = int(img.shape[0]), int(img.shape[1])
    out = torch.empty((1, 3, H, W), dtype=torch.float32, device=img.device)
    check(_lib.load().csm_u8_hwc_to_f32_chw(ptr(img), i32(H), i32(W), ptr(out), stream_ptr()), "u8_hwc_to_f32_chw")
    return out


def ctypes_ptr(t, offset_elems):
    """device pointer `offset_elems` elements into tensor t"""
    import ctypes
    return ctypes.c_void_p(t.data_ptr() + offset_elems * t.element_size())


def _tail_scratch(dev):
    """scratch of the sync-free frame tail (percentile select state + partial histograms, bokeh stats), one set per (device,
    stream): calls on one stream are ordered, calls on different streams / threads of a device get their own state"""
    key = (dev, torch.cuda.current_stream(dev).cuda_stream)
    if key not in _TAIL_SCRATCH:
        L = _lib.load()
        _TAIL_SCRATCH[key] = (torch.zeros(L.csm_percentile_scratch_bytes(), dtype=torch.uint8, device=dev),          # counting tables start cleared
                              torch.zeros(L.csm_bokeh_depth_scratch_bytes(), dtype=torch.uint8, device=dev),   # completion counter starts at 0
                              torch.empty(2, dtype=torch.float32, device=dev))
    return _TAIL_SCRATCH[key]


def colorize_gray_r(tenValue):
    """colorize(value, cmap='gray_r')[..., 0] -> uint8 tensor (same shape, squeezed); vmin/vmax = 2nd / 85th percentile
    (depth_modules/zoedepth/utils/misc.py:97-135).  No sort and no host sync: the percentiles are selected on the device
    (csm_percentile_pair) and consumed from device memory."""
    global _GRAY_R_LUT
    import ctypes
    v = _dev(tenValue.reshape(-1), "tenValue")
    L = _lib.load()
    if _GRAY_R_LUT is None:
        # matplotlib: lut = 1 - linspace(0,1,256) (float64); bytes=True -> (lut*255).astype(uint8)  [truncation, not 255-k]
        lut = ((1.0 - _np.linspace(0.0, 1.0, 256)) * 255).astype(_np.uint8)
        _GRAY_R_LUT = (ctypes.c_uint8 * 256)(*[int(x) for x in lut])
    sel, _, vmm = _tail_scratch(v.device)
    check(L.csm_percentile_pair(ptr(v), i64(v.numel()), f64(2.0), f64(85.0), ptr(vmm), ptr(sel), stream_ptr()), "percentile_pair")
    out = torch.empty(v.numel(), dtype=torch.uint8, device=v.device)
    check(L.csm_colorize_gray_r_dev(ptr(v), ptr(out), i64(v.numel()), ptr(vmm), _GRAY_R_LUT, stream_ptr()), "colorize")
    return out.reshape(tenValue.squeeze().shape)


def bokeh_blur(img, depth, num_samples=32, lightness_factor=10, depth_factor=2, use_cuda=False, focal_plane=None):
    """bokeh_blur -- utils/effects.py:143-181.  img uint8 HxWx3 and depth uint8/float HxW as device tensors (numpy inputs are
    uploaded); returns a uint8 HxWx3 DEVICE tensor.  `use_cuda` is accepted for signature compatibility (always device)."""
    L = _lib.load()
    dev = img.device if isinstance(img, torch.Tensor) else torch.device('cuda', torch.cuda.current_device())
    img_d = (img if isinstance(img, torch.Tensor) else torch.from_numpy(_np.ascontiguousarray(img))).to(dev).contiguous()
    H, W = int(img_d.shape[0]), int(img_d.shape[1])
    n = H * W
    d8 = (depth if isinstance(depth, torch.Tensor) else torch.from_numpy(_np.ascontiguousarray(depth))).to(dev)
    if d8.dtype not in (torch.uint8, torch.float32):
        d8 = d8.float()                                    # `depth.astype(np.float32)`, utils/effects.py:147
    d8 = d8.contiguous()
    dm = torch.empty((H, W), dtype=torch.float32, device=dev)
    if d8.dtype == torch.uint8 and depth_factor == 1 and focal_plane is not None:
        # the pipeline's call (uint8 colorized depth, configs/3dkenburns.yaml:47): depth.max(), min / max of depth.max() - |depth - focal|
        # (utils/effects.py:146-153) in closed form from the uint8 histogram, on the device
        fp = float(_np.float32(focal_plane))
        check(L.csm_bokeh_depth_auto(ptr(d8), ptr(dm), i64(n), f32(fp), ptr(_tail_scratch(dev)[1]), stream_ptr()), "bokeh_depth")
    else:
        # the reference's general form, incl. its own defaults (float depth, depth_factor = 2, focal_plane = None)
        tmp = torch.empty(n + 4 + 512, dtype=torch.float32, device=dev)
        check(L.csm_bokeh_depth_general(ptr(d8), i32(1 if d8.dtype == torch.uint8 else 0), i64(n), i32(0 if focal_plane is None else 1),
                                        f32(0.0 if focal_plane is None else float(_np.float32(focal_plane))), f32(float(depth_factor)),
                                        ptr(tmp), ctypes_ptr(tmp, n), ctypes_ptr(tmp, n + 4), ptr(dm), stream_ptr()), "bokeh_depth_general")
    hi = torch.empty((H, W, 3), dtype=torch.float32, device=dev)
    check(L.csm_bokeh_highlight(ptr(img_d), ptr(hi), i64(n * 3), f32(lightness_factor), stream_ptr()), "bokeh_highlight")
    a, b = torch.empty_like(hi), torch.empty_like(hi)
    PI = _math.pi
    for src, dst, (dx, dy) in ((hi, a, (0, 1)), (a, b, (_math.cos(-PI / 6), _math.sin(-PI / 6)))):
        check(L.csm_bokeh_pass(ptr(src), ptr(dm), ptr(dst), i32(H), i32(W), i32(num_samples), f32(dx), f32(dy), stream_ptr()), "bokeh_pass")
    out = torch.empty((H, W, 3), dtype=torch.uint8, device=dev)
    # third pass + ((diag + rhom) / 2) ** (1 / lightness) * 255 -> uint8 (utils/effects.py:172,179-180) in one kernel
    check(L.csm_bokeh_pass_finish(ptr(b), ptr(dm), ptr(out), i32(H), i32(W), i32(num_samples), f32(_math.cos(-PI * 5 / 6)),
                                  f32(_math.sin(-PI * 5 / 6)), f32(lightness_factor), stream_ptr()), "bokeh_pass_finish")
    return out
