"""Lowering of a dense net into the layer program of libcsm355 (csm_op records, include/csm355.h).

Host-side only (numpy + ctypes): weight import (BN folding, MFMA packing), tensor/liveness planning,
op emission.  The same Program object yields
  * the HIP program  (packed weights, super-grouped convs)      -> csm_run_program
  * the oracle program (natural weights, real groups)           -> tests only (oracle/nets.py)
so that the two executions share nothing but the op order and tensor plan.
"""
import ctypes
import os

import numpy as np

# ---- C structs (must match include/csm355.h) -------------------------------------------------
OP_CONV, OP_DWCONV, OP_MAXPOOL, OP_BILINEAR, OP_NEAREST, OP_ADD, OP_GAVGPOOL, OP_SCALE = 1, 2, 3, 4, 5, 6, 7, 8
OP_NCHW_TO_NHWC, OP_NHWC_TO_NCHW, OP_ACT, OP_COPY, OP_ATTRACTOR, OP_LOGBINOM = 9, 10, 11, 12, 13, 14
OP_LAYERNORM, OP_ATTENTION, OP_TOKENS, OP_DEPTH_TO_SPACE = 15, 16, 17, 18
ACT = {None: 0, 'none': 0, 'relu': 1, 'silu': 2, 'prelu': 3, 'hsigmoid': 4, 'sigmoid': 5, 'softplus': 6, 'gelu': 7}


class CsmTensorDesc(ctypes.Structure):
    _fields_ = [("offset", ctypes.c_int64), ("ext", ctypes.c_int32), ("n", ctypes.c_int32), ("h", ctypes.c_int32),
                ("w", ctypes.c_int32), ("c", ctypes.c_int32), ("ld", ctypes.c_int32)]


class CsmOp(ctypes.Structure):
    _fields_ = [("kind", ctypes.c_int32), ("in0", ctypes.c_int32), ("in1", ctypes.c_int32), ("out", ctypes.c_int32),
                ("kh", ctypes.c_int32), ("kw", ctypes.c_int32), ("stride", ctypes.c_int32), ("pad", ctypes.c_int32),
                ("dil", ctypes.c_int32), ("groups", ctypes.c_int32), ("cin_g", ctypes.c_int32),
                ("cout_g", ctypes.c_int32), ("act", ctypes.c_int32), ("res_mode", ctypes.c_int32),
                ("w_off", ctypes.c_int64), ("b_off", ctypes.c_int64), ("aux_off", ctypes.c_int64),
                ("flags", ctypes.c_int32), ("ksplit", ctypes.c_int32), ("scratch", ctypes.c_int32),
                ("tile", ctypes.c_int32)]


def fold_bn(w, b, gamma, beta, mean, var, eps):
    """conv+BN(eval) -> conv: w' = w*s, b' = (b-mean)*s+beta, s = gamma/sqrt(var+eps); float64 math, fp32 result"""
    w = np.asarray(w, np.float64)
    s = np.asarray(gamma, np.float64) / np.sqrt(np.asarray(var, np.float64) + eps)
    b0 = np.zeros(w.shape[0]) if b is None else np.asarray(b, np.float64)
    return (w * s.reshape(-1, 1, 1, 1)).astype(np.float32), ((b0 - np.asarray(mean, np.float64)) * s
                                                             + np.asarray(beta, np.float64)).astype(np.float32)


# channels of a grouped convolution's super-group: narrow groups run block-diagonally inside one 32x32x2 MFMA tile (exact: fmaf(x, 0,
# acc) == acc; 4x / 2x zero work for 8- / 16-channel groups).  Measured round 3: 16-wide super-groups on the 16x16x4 tile (2x / none)
# are SLOWER (ResNeXt g8 layer 22.0 -> 16.0 TF/s, g16 42.8 -> 31.9: the register-staged kernel's per-block overhead dominates).
GROUP_PACK = 32


def pack_conv_weights(w, groups):
    """w [cout, cin_g, kh, kw] (fp32, folded) -> (packed fp32 1-D, super_groups, cin_sg, cout_sg)
    packed layout: [sg][cb][tap][npad][32]  (cb = 32-channel block, OUTER = the chain order; npad = cout_sg rounded to 32; zero padded)"""
    cout, cin_g, kh, kw = w.shape
    if groups == 1:
        sg, cin_sg, cout_sg = 1, cin_g, cout
        wsg = w.reshape(1, cout, cin_g, kh, kw)
    else:
        cout_g = cout // groups
        s = max(1, GROUP_PACK // cin_g)
        while groups % s:
            s -= 1
        sg, cin_sg, cout_sg = groups // s, s * cin_g, s * cout_g
        wsg = np.zeros((sg, cout_sg, cin_sg, kh, kw), np.float32)
        wg = w.reshape(groups, cout_g, cin_g, kh, kw)
        for g in range(groups):
            q, r = divmod(g, s)
            wsg[q, r * cout_g:(r + 1) * cout_g, r * cin_g:(r + 1) * cin_g] = wg[g]
    ncb, npad = (cin_sg + 31) // 32, (cout_sg + 31) // 32 * 32
    full = np.zeros((sg, npad, ncb * 32, kh, kw), np.float32)
    full[:, :cout_sg, :cin_sg] = wsg
    packed = full.reshape(sg, npad, ncb, 32, kh, kw).transpose(0, 2, 4, 5, 1, 3)     # sg,cb,kh,kw,npad,32
    return np.ascontiguousarray(packed).reshape(-1), sg, cin_sg, cout_sg


# split-K rule (part of the numerical contract of a lowered program): below KSPLIT_BELOW 64x64 output tiles PER SAMPLE, cut K so that
# about KSPLIT_TARGET blocks exist at batch 1.  The rule sees one sample's shape only, so a sample's bits do not depend on the batch it
# runs in; whether the runs become separate blocks or are walked by one block is decided on the device (csm_op.tile, speed only).
KSPLIT_BELOW = int(os.environ.get('CSM_KSPLIT_BELOW', '512'))
KSPLIT_TARGET = int(os.environ.get('CSM_KSPLIT_TARGET', '768'))


# Narrow grouped 3x3 convolutions (ResNeXt conv2: 8 / 16 / 32 channels per group): csrc/grouped.hip runs them on the vector pipe with the
# DIRECT chain (same bits as the block-diagonal matrix-pipe form above) from its own weight image.  Speed only; CSM_GROUPED_VALU=0 keeps
# the super-group form, CSM_GROUPED_VALU_MAX_CG bounds the channels per group that take it (default 16: at 32 channels per group the
# vector kernel is 6 % ahead at batch 8 -- 81 against 87 us at 8 x 40^2 x 1024 -- and a third BEHIND on one sample, 29 against 21 us,
# where a launch is 160 blocks of one wave per SIMD; profiles/r06_grouped_conv.txt).
GROUPED_VALU = os.environ.get('CSM_GROUPED_VALU', '1') != '0'
GROUPED_VALU_MAX_CG = int(os.environ.get('CSM_GROUPED_VALU_MAX_CG', '16'))
CONV_FLAG_GROUPED = 16
_CHAIN8 = (0, 4, 1, 5, 2, 6, 3, 7)


def grouped_valu_eligible(kh, kw, stride, pad, dil, groups, cin_g, cout, ld=None, h=0, w=0):
    """(mirrors csrc/grouped.hip::grouped_eligible, including its 32-bit byte-offset bound on ONE sample's input view: the weights of a
    flagged op exist in the vector kernel's image only, so a layer the kernel would refuse must not be flagged)"""
    c = groups * cin_g
    return (GROUPED_VALU and groups > 1 and kh == 3 and kw == 3 and stride == 1 and pad == 1 and dil == 1 and cout == c
            and cin_g in (8, 16, 32) and cin_g <= GROUPED_VALU_MAX_CG and c % 32 == 0 and (ld is None or ld % 4 == 0)
            and h * w * (ld or c) * 4 < 2 ** 31)


def pack_grouped_weights(w, groups):
    """w [cout, cin_g, 3, 3] with cout_g == cin_g in {8, 16, 32} -> the image of include/csm355.h "CSM_CONV_FLAG_GROUPED":
    [group][octet][tap][kb][h][chain position i][t] = w[group cin_g + 8 octet + 4 h + t][8 kb + 4 (i & 1) + (i >> 1)][ky][kx]"""
    cout, cg, kh, kw = w.shape
    assert kh == 3 and kw == 3 and cout == groups * cg and cg % 8 == 0
    og = kb = cg // 8
    wg = w.reshape(groups, og, 2, 4, kb, 8, 9)[:, :, :, :, :, _CHAIN8, :]          # [g][octet][h][t][kb][i][tap]
    return np.ascontiguousarray(wg.transpose(0, 1, 6, 4, 2, 5, 3)).reshape(-1)      # [g][octet][tap][kb][h][i][t]


def pack_stem_weights(w):
    """stem convs (cin padded to 4, groups 1): K is packed as (tap, channel) -- 8 taps x 4 channels per 32-wide chunk instead of
    one chunk per tap with 4 of 32 channels used.  w [cout, 4, kh, kw] -> packed [chunk][npad][32]; within each 8-block the positions
    hold (tap 2j: c0, c2 | tap 2j+1: c0, c2 | tap 2j: c1, c3 | tap 2j+1: c1, c3), so that the MFMA lane order 0,4,1,5,2,6,3,7
    multiplies tap 2j's channels 0..3, then tap 2j+1's: the contract's chain (taps row-major, channels ascending)."""
    cout, cin, kh, kw = w.shape
    assert cin == 4
    ntaps = kh * kw
    nck, npad = (ntaps + 7) // 8, (cout + 31) // 32 * 32
    wt = w.reshape(cout, 4, ntaps)
    packed = np.zeros((nck, npad, 32), np.float32)
    pos = ((0, 0), (0, 2), (1, 0), (1, 2), (0, 1), (0, 3), (1, 1), (1, 3))        # position in the 8-block -> (tap parity, channel)
    for c in range(nck):
        for jb in range(4):
            for r, (s_, ch) in enumerate(pos):
                tap = 8 * c + 2 * jb + s_
                if tap < ntaps:
                    packed[c, :cout, 8 * jb + r] = wt[:, ch, tap]
    return packed.reshape(-1)


# ---- Winograd F(2x2, 3x3) (include/csm355.h "Winograd contract") ------------------------------------------------------------------
# Which layers take it is part of the NUMERICAL contract of a lowered program (the result differs from the direct chain by fp32
# rounding), so the rule sees one sample's shape only -- never the batch, never the tuner: 3x3 / stride 1 / dilation 1 / pad 1 / dense,
# cin % 32 == 0, cout % 64 == 0 and at least WINO_MIN_PIXELS output pixels per sample.  CSM_CONV_EXACT_DIRECT=1 lowers every layer to
# the direct chain (the round-1..4 arithmetic).
CONV_FLAG_STEM, CONV_FLAG_WINOGRAD = 2, 4
WINO_ENABLE = os.environ.get('CSM_CONV_EXACT_DIRECT', '0') != '1'
WINO_MIN_PIXELS = int(os.environ.get('CSM_WINO_MIN_PIXELS', '6400'))
WINO_BN = 64


def wino_eligible(kh, kw, stride, pad, dil, groups, cin_g, cout, ho, wo, ld=None):
    """ld = channel pitch of the input view (None: cin_g).  The byte bounds mirror csrc/wino.hip::wino_eligible (32-bit buffer
    descriptors: one sample's input view and the U panel under 2 GiB, pitch % 4 == 0); they depend on the shape only, so a layer
    beyond them lowers to the direct chain instead of failing at run time"""
    ld = cin_g if ld is None else ld
    return (kh == 3 and kw == 3 and stride == 1 and pad == 1 and dil == 1 and groups == 1 and cin_g % 32 == 0 and cout % WINO_BN == 0
            and ho * wo >= WINO_MIN_PIXELS and ld % 4 == 0 and ((ho * wo - 1) * ld + cin_g) * 4 < 2 ** 31
            and (cout // WINO_BN) * (cin_g // 32) * 4 * 32768 < 2 ** 31)


def wino_transform(w):
    """w [cout, cin, 3, 3] fp32 -> U [16, cout, cin] fp32 = G g G^T, float64 arithmetic in the contract's fixed order (rows, then columns;
    each row (g0, ((g0 + g1) + g2) / 2, ((g0 - g1) + g2) / 2, g2)) -- elementwise numpy operations only, so every value is the IEEE
    result of the same expression the oracle evaluates in C"""
    g = np.asarray(w, np.float64)

    def rows(g0, g1, g2):
        return [g0, ((g0 + g1) + g2) * 0.5, ((g0 - g1) + g2) * 0.5, g2]
    r = rows(g[:, :, 0, :], g[:, :, 1, :], g[:, :, 2, :])                 # 4 x [cout, cin, 3]
    u = [rows(ri[:, :, 0], ri[:, :, 1], ri[:, :, 2]) for ri in r]         # u[i][j] [cout, cin]
    return np.stack([u[i][j] for i in range(4) for j in range(4)]).astype(np.float32)


def pack_wino_weights(w):
    """-> packed fp32 1-D in the LDS image order of k_conv_wino: [cout / 64][cb][q][f][h][64][4], channel 32 cb + 8 q + 4 h + e"""
    cout, cin = w.shape[:2]
    assert cout % WINO_BN == 0 and cin % 32 == 0
    U = wino_transform(w)                                                  # [16, cout, cin]
    U = U.reshape(16, cout // WINO_BN, WINO_BN, cin // 32, 4, 2, 4)       # f, nt, co, cb, q, h, e
    return np.ascontiguousarray(U.transpose(1, 3, 4, 0, 5, 2, 6)).reshape(-1)   # nt, cb, q, f, h, co, e


# ---- Winograd F(4x4, 3x3) (include/csm355.h "Winograd F(4x4) contract") ----------------------------------------------------------
# Same layer class as F(2x2); 36 products per 4x4 output tile and channel pair instead of 64.  A layer takes it when it has at least
# WINO4_MIN_PIXELS output pixels per sample (per-sample rule: batch invariant); CSM_WINO4=0 keeps every Winograd layer on F(2x2).
CONV_FLAG_WINOGRAD4 = 8
WINO4_ENABLE = os.environ.get('CSM_WINO4', '1') != '0'
WINO4_MIN_PIXELS = int(os.environ.get('CSM_WINO4_MIN_PIXELS', '6400'))
WINO4_SPLIT_BELOW = int(os.environ.get('CSM_WINO4_SPLIT_BELOW', '1024'))      # block tiles of a launch below which the scratch for the row-split forms is planned (speed only)


WINO4_SMALL_MIN_PIXELS = int(os.environ.get('CSM_WINO4_SMALL_MIN_PIXELS', '500'))
WINO4_SMALL_MAX_CIN = int(os.environ.get('CSM_WINO4_SMALL_MAX_CIN', '512'))
WINO4_SMALL_MIN_CH = int(os.environ.get('CSM_WINO4_SMALL_MIN_CH', '128'))      # narrower layers on small maps gain nothing at any batch


def wino4_eligible(kh, kw, stride, pad, dil, groups, cin_g, cout, ho, wo, ld=None):
    """the F(2x2) layer class (3x3 / stride 1 / pad 1 / dense, cin % 32 == 0, cout % 64 == 0, 32-bit descriptor ranges) on maps of at
    least WINO4_MIN_PIXELS output pixels per sample -- and, round 6b, on SMALLER maps down to WINO4_SMALL_MIN_PIXELS (23 x 23) when
    128 <= cin <= WINO4_SMALL_MAX_CIN and cout >= 128: at batch 8 those layers run 1.4-1.8 x the direct kernels, and the K loop of one block tile
    (cin / 4 steps) stays short enough for the row-split forms at batch 1 (profiles/r06_wino4_forms.txt).  Per-sample: batch invariant."""
    ld = cin_g if ld is None else ld
    px = ho * wo
    return (kh == 3 and kw == 3 and stride == 1 and pad == 1 and dil == 1 and groups == 1 and cin_g % 32 == 0 and cout % WINO_BN == 0
            and (px >= WINO4_MIN_PIXELS or (px >= WINO4_SMALL_MIN_PIXELS and WINO4_SMALL_MIN_CH <= cin_g <= WINO4_SMALL_MAX_CIN
                                            and cout >= WINO4_SMALL_MIN_CH))
            and ld % 4 == 0 and ((px - 1) * ld + cin_g) * 4 < 2 ** 31 and (cout // WINO_BN) * (cin_g // 4) * 36864 < 2 ** 31)


def wino4_transform(w):
    """w [cout, cin, 3, 3] fp32 -> U [36, cout, cin] fp32 = G g G^T (Lavin & Gray, points 0, +-1, +-2, inf), float64 arithmetic in the
    contract's fixed order -- elementwise numpy operations only (IEEE double: the same values the oracle's C computes)"""
    g = np.asarray(w, np.float64)

    def rows(g0, g1, g2):
        return [g0 * 0.25, -((g0 + g1) + g2) / 6.0, -((g0 - g1) + g2) / 6.0, ((g0 + 2.0 * g1) + 4.0 * g2) / 24.0,
                ((g0 - 2.0 * g1) + 4.0 * g2) / 24.0, g2]
    r = rows(g[:, :, 0, :], g[:, :, 1, :], g[:, :, 2, :])                 # 6 x [cout, cin, 3]
    u = [rows(ri[:, :, 0], ri[:, :, 1], ri[:, :, 2]) for ri in r]         # u[i][j] [cout, cin]
    return np.stack([u[i][j] for i in range(6) for j in range(6)]).astype(np.float32)


def pack_wino4_weights(w):
    """-> packed fp32 1-D for k_conv_wino4: [cout / 64][step s][wave 6 nh + i][piece p][lh][li][jj][t], value
    U[6 i + 2 p + jj][64 nt + 32 nh + li][8 (s >> 1) + 4 lh + 2 (s & 1) + t]"""
    cout, cin = w.shape[:2]
    assert cout % WINO_BN == 0 and cin % 8 == 0
    U = wino4_transform(w)                                                 # [36, cout, cin]
    U = U.reshape(6, 3, 2, cout // WINO_BN, 2, 32, cin // 8, 2, 2, 2)      # i, p, jj, nt, nh, li, q8, lh, hf, t
    return np.ascontiguousarray(U.transpose(3, 6, 8, 4, 0, 1, 7, 5, 2, 9)).reshape(-1)   # nt, q8, hf, nh, i, p, lh, li, jj, t


class Buf:
    def __init__(self, n, h, w, c, ext=-1, nchw=False):
        self.n, self.h, self.w, self.c, self.ext, self.nchw = n, h, w, c, ext, nchw
        self.offset = None
        self.first, self.last = None, None
        self.keep = False          # outputs / persistent tensors are never recycled
        self.alias = None          # crop_rows(): a view of another buffer's first rows (shares its storage and lifetime)


class T:
    """view of `c` channels starting at channel `coff` of a buffer"""
    def __init__(self, prog, buf, coff, c):
        self.prog, self.buf, self.coff, self.c = prog, buf, coff, c
        self.id = len(prog.views)
        prog.views.append(self)

    n = property(lambda s: s.buf.n)
    h = property(lambda s: s.buf.h)
    w = property(lambda s: s.buf.w)

    def slice(self, c0, c1):
        return T(self.prog, self.buf, self.coff + c0, c1 - c0)

    @property
    def shape(self):
        return (self.n, self.h, self.w, self.c)


def _round4(c):
    return (c + 3) // 4 * 4


class Program:
    def __init__(self, name=""):
        self.name = name
        self.views, self.bufs, self.ops = [], [], []      # ops: list of dict
        self.w_hip, self.w_nat = [], []
        self.n_hip = self.n_nat = 0
        self.n_ext = 0
        self.flops = 0                # natural (direct-convolution) FLOPs of the net: 2 MAC
        self.flops_exec = 0           # FLOPs the matrix pipe executes: Winograd layers count 16 / 36 of their natural work
        self.conv_bytes = 0

    # ---- tensors --------------------------------------------------------------------------
    def buffer(self, n, h, w, c):
        b = Buf(n, h, w, c)
        self.bufs.append(b)
        return T(self, b, 0, c)

    def ext_nchw(self, n, c, h, w):
        """external (caller-owned) NCHW tensor, e.g. the net input / output"""
        b = Buf(n, h, w, c, ext=self.n_ext, nchw=True)
        self.n_ext += 1
        self.bufs.append(b)
        return T(self, b, 0, c)

    def _w(self, hip, nat):
        oh, on = self.n_hip, self.n_nat
        hip, nat = np.ascontiguousarray(hip, np.float32).reshape(-1), np.ascontiguousarray(nat, np.float32).reshape(-1)
        pad = (-hip.size) % 4
        if pad:
            hip = np.concatenate([hip, np.zeros(pad, np.float32)])
        self.w_hip.append(hip); self.w_nat.append(nat)
        self.n_hip += hip.size; self.n_nat += nat.size
        return oh, on

    def _emit(self, kind, in0, in1, out, **kw):
        op = dict(kind=kind, in0=in0.id, in1=-1 if in1 is None else in1.id, out=out.id, kh=0, kw=0, stride=1, pad=0,
                  dil=1, groups=1, cin_g=0, cout_g=0, act=0, res_mode=0, w_off=-1, b_off=-1, aux_off=-1, flags=0,
                  ksplit=1, scratch=-1, nat=None)
        op.update(kw)
        i = len(self.ops)
        scr = kw.get('scratch_view')
        op.pop('scratch_view', None)
        for t in (in0, in1, out, scr):
            if t is None:
                continue
            b = t.buf
            while b is not None:                       # a row-crop view keeps the buffer it aliases alive
                b.first = i if b.first is None else b.first
                b.last = i
                b = b.alias
        self.ops.append(op)
        return out

    # ---- ops ------------------------------------------------------------------------------
    def to_nhwc(self, x_ext, c_pad=None):
        c = _round4(x_ext.c) if c_pad is None else c_pad
        out = self.buffer(x_ext.n, x_ext.h, x_ext.w, c)
        return self._emit(OP_NCHW_TO_NHWC, x_ext, None, out)

    def to_nchw(self, x, out_ext):
        return self._emit(OP_NHWC_TO_NCHW, x, None, out_ext)

    def conv(self, x, w, b=None, stride=1, pad=0, dil=1, groups=1, act=None, slope=None, res=None, res_mode=0, out=None):
        """w [cout, cin/groups, kh, kw] fp32 (BN already folded); x channels may exceed w's cin only by zero padding"""
        w = np.asarray(w, np.float32)
        cout, cin_g, kh, kw = w.shape
        if groups == 1 and x.c != cin_g:
            assert x.c > cin_g and x.c == _round4(cin_g), (x.c, cin_g)
            w = np.concatenate([w, np.zeros((cout, x.c - cin_g, kh, kw), np.float32)], 1)
            cin_g = x.c
        assert cin_g * groups == x.c, (cin_g, groups, x.c)
        ho = (x.h + 2 * pad - dil * (kh - 1) - 1) // stride + 1
        wo = (x.w + 2 * pad - dil * (kw - 1) - 1) // stride + 1
        if out is None:
            out = self.buffer(x.n, ho, wo, cout)
        assert out.shape == (x.n, ho, wo, cout), (out.shape, (x.n, ho, wo, cout))
        stem = groups == 1 and cin_g == 4 and cout > 4        # k_conv_stem: (tap, channel)-packed K, csm_op.flags bit 1
        wino4 = self.winograd and self.winograd4 and wino4_eligible(kh, kw, stride, pad, dil, groups, cin_g, cout, ho, wo, ld=x.buf.c)
        wino = wino4 or (self.winograd and wino_eligible(kh, kw, stride, pad, dil, groups, cin_g, cout, ho, wo, ld=x.buf.c))
        gvalu = self.grouped_valu and grouped_valu_eligible(kh, kw, stride, pad, dil, groups, cin_g, cout, ld=x.buf.c, h=x.h, w=x.w)
        if stem:
            packed, sg, cin_sg, cout_sg = pack_stem_weights(w), 1, 4, cout
        elif gvalu:
            packed, sg, cin_sg, cout_sg = pack_grouped_weights(w, groups), groups, cin_g, cout // groups
        elif wino4:
            packed, sg, cin_sg, cout_sg = pack_wino4_weights(w), 1, cin_g, cout
        elif wino:
            packed, sg, cin_sg, cout_sg = pack_wino_weights(w), 1, cin_g, cout
        else:
            packed, sg, cin_sg, cout_sg = pack_conv_weights(w, groups)
        w_h, w_n = self._w(packed, w)
        b_h = b_n = a_h = a_n = -1
        if b is not None:
            b_h, b_n = self._w(b, b)
        if slope is not None:
            a_h, a_n = self._w(slope, slope)
        self.flops += 2 * x.n * ho * wo * cout * cin_g * kh * kw
        self.flops_exec += ((2 * x.n * ((ho + 3) // 4) * ((wo + 3) // 4) * 36 * cout * cin_g) if wino4 else
                            (2 * x.n * ((ho + 1) // 2) * ((wo + 1) // 2) * 16 * cout * cin_g) if wino else (2 * x.n * ho * wo * cout * cin_g * kh * kw))
        self.conv_bytes += 4 * (x.n * x.h * x.w * x.c + x.n * ho * wo * cout + w.size)
        ksplit, scr = (1 if stem or wino else self.choose_ksplit(ho * wo, cout, kh * kw * ((cin_sg + 31) // 32), groups)), None
        if ksplit > 1:
            scr = self.buffer(x.n, ho, wo, ksplit * cout)
        elif wino4 and (x.n * ((ho + 15) // 16) * ((wo + 15) // 16) + 1) // 2 * (cout // WINO_BN) < WINO4_SPLIT_BELOW:
            # fewer block tiles (two 16 x 16-pixel squares x 64 channels each) than would fill the chip: the library may run a ROW-SPLIT form of the same arithmetic
            # (csrc/wino4.hip, speed only) and needs 24 floats of scratch per 4x4 tile and channel for it
            scr = self.buffer(x.n, (ho + 3) // 4, (wo + 3) // 4, 24 * cout)
        return self._emit(OP_CONV, x, res, out, kh=kh, kw=kw, stride=stride, pad=pad, dil=dil, groups=sg, cin_g=cin_sg,
                          cout_g=cout_sg, act=ACT[act], res_mode=res_mode if res is not None else 0, w_off=w_h, b_off=b_h,
                          aux_off=a_h, ksplit=ksplit, scratch=-1 if scr is None else scr.id, scratch_view=scr,
                          flags=CONV_FLAG_STEM if stem else (CONV_FLAG_GROUPED if gvalu else CONV_FLAG_WINOGRAD4 if wino4 else (CONV_FLAG_WINOGRAD if wino else 0)),
                          nat=dict(groups=groups, cin_g=cin_g, cout_g=cout // groups, w_off=w_n, b_off=b_n, aux_off=a_n))

    split_k = True
    winograd = WINO_ENABLE          # (class default; a test may lower one program with / without it)
    grouped_valu = GROUPED_VALU     # narrow grouped 3x3 layers on the vector pipe (csrc/grouped.hip; same bits as the super-group form)
    winograd4 = WINO4_ENABLE        # F(4x4) for the Winograd layers of at least WINO4_MIN_PIXELS pixels (else F(2x2))

    def choose_ksplit(self, M, N, T, groups):
        """small feature maps (M = output pixels of ONE sample): not enough 64x64 output tiles to fill 256 CUs -> cut K
        (part of the numerical contract, hence independent of the batch size)"""
        if not self.split_k or groups != 1 or N <= 4:      # N <= 4: k_conv_narrow, one output pixel per lane, needs no split
            return 1
        tiles = ((M + 63) // 64) * ((N + 63) // 64)
        if tiles >= KSPLIT_BELOW or T < 8:      # enough blocks per CU: the extra reduce launch (~6 us) costs more than it buys
            return 1
        s = min(-(-KSPLIT_TARGET // tiles), T // 4, 16)
        return s if s >= 2 else 1

    def dwconv(self, x, w, b=None, stride=1, pad=0, dil=1, act=None, out=None):
        """depthwise: w [c,1,kh,kw]"""
        w = np.asarray(w, np.float32)
        c, _, kh, kw = w.shape
        assert c == x.c and c % 4 == 0
        ho = (x.h + 2 * pad - dil * (kh - 1) - 1) // stride + 1
        wo = (x.w + 2 * pad - dil * (kw - 1) - 1) // stride + 1
        if out is None:
            out = self.buffer(x.n, ho, wo, c)
        w_h, w_n = self._w(w.reshape(c, kh * kw).T, w)          # hip: [tap][c]
        b_h = b_n = -1
        if b is not None:
            b_h, b_n = self._w(b, b)
        self.flops += 2 * x.n * ho * wo * c * kh * kw
        self.flops_exec += 2 * x.n * ho * wo * c * kh * kw
        return self._emit(OP_DWCONV, x, None, out, kh=kh, kw=kw, stride=stride, pad=pad, dil=dil, act=ACT[act], w_off=w_h,
                          b_off=b_h, nat=dict(w_off=w_n, b_off=b_n))

    def maxpool(self, x, k, stride, pad=0, ceil_mode=False, out=None):
        def osz(i):
            num = i + 2 * pad - k
            o = (-(-num // stride) if ceil_mode else num // stride) + 1
            if ceil_mode and (o - 1) * stride >= i + pad:        # torch: last window must start inside input+left pad
                o -= 1
            return o
        if out is None:
            out = self.buffer(x.n, osz(x.h), osz(x.w), x.c)
        return self._emit(OP_MAXPOOL, x, None, out, kh=k, kw=k, stride=stride, pad=pad)

    def bilinear(self, x, size, align_corners=False, out=None, act=None, slope=None):
        """F.interpolate(mode='bilinear') [followed by an activation applied to the interpolated value in the same kernel: the
        Upsample blocks of the Ken Burns GridNets are Upsample -> PReLU -> conv and nothing else reads the un-activated map]"""
        if out is None:
            out = self.buffer(x.n, size[0], size[1], x.c)
        assert (out.h, out.w) == tuple(size)
        a_h = a_n = -1
        if slope is not None:
            slope = np.asarray(slope, np.float32)
            if slope.size < x.c:                       # zero-padded channels keep slope 0
                slope = np.concatenate([slope, np.zeros(x.c - slope.size, np.float32)])
            a_h, a_n = self._w(slope, slope)
        return self._emit(OP_BILINEAR, x, None, out, flags=1 if align_corners else 0, act=ACT[act], aux_off=a_h, nat=dict(aux_off=a_n))

    def nearest(self, x, factor, out=None):
        if out is None:
            out = self.buffer(x.n, x.h * factor, x.w * factor, x.c)
        return self._emit(OP_NEAREST, x, None, out)

    def add(self, a, b, act=None, out=None):
        """a + b.  `a` may be one row and / or one column LARGER than b: its first b.h x b.w pixels are used (the reference's
        `pad(tenUp, [0, -1, 0, -1])` crops, disparity_estimation.py:172-173) -- the kernel indexes a with its own pitch"""
        assert a.n == b.n and a.c == b.c and b.h <= a.h <= b.h + 1 and b.w <= a.w <= b.w + 1, (a.shape, b.shape)
        if out is None:
            out = self.buffer(b.n, b.h, b.w, b.c)
        return self._emit(OP_ADD, a, b, out, act=ACT[act])

    def act(self, x, act, out=None, slope=None):
        if out is None:
            out = self.buffer(x.n, x.h, x.w, x.c)
        a_h = a_n = -1
        if slope is not None:
            slope = np.asarray(slope, np.float32)
            if slope.size < x.c:                       # zero-padded channels keep slope 0
                slope = np.concatenate([slope, np.zeros(x.c - slope.size, np.float32)])
            a_h, a_n = self._w(slope, slope)
        return self._emit(OP_ACT, x, None, out, act=ACT[act], aux_off=a_h, nat=dict(aux_off=a_n))

    def copy(self, x, out):
        return self._emit(OP_COPY, x, None, out)

    def to_nhwc_into(self, x_ext, out):
        """NCHW ext tensor -> a channel slice of an NHWC buffer (channels beyond the source's are written as zeros)"""
        return self._emit(OP_NCHW_TO_NHWC, x_ext, None, out)

    def attractor(self, A, b, alpha, attractor_type='inv', kind='mean'):
        """ZoeDepth bin-centre attractor update (attractor.py): out = b + agg_i dist(A_i - b), gamma = 2"""
        out = self.buffer(b.n, b.h, b.w, b.c)
        par = np.array([alpha, 0, 0, 0], np.float32)
        a_h, a_n = self._w(par, par)
        flags = (1 if attractor_type == 'exp' else 0) | (2 if kind == 'mean' else 0)
        return self._emit(OP_ATTRACTOR, A, b, out, aux_off=a_h, flags=flags, nat=dict(aux_off=a_n))

    def logbinom(self, pt, centers, p_eps, min_temp, max_temp, log_binom_table):
        """ConditionalLogBinomial tail + expectation over the bins -> [n,h,w,1] (stored in a 4-channel buffer)"""
        out = self.buffer(pt.n, pt.h, pt.w, 4).slice(0, 1)
        par = np.concatenate([np.array([p_eps, min_temp, max_temp], np.float32), np.asarray(log_binom_table, np.float32)])
        a_h, a_n = self._w(par, par)
        return self._emit(OP_LOGBINOM, pt, centers, out, aux_off=a_h, nat=dict(aux_off=a_n))

    # ---- transformer ops (MiDaS DPT-BEiT core; a token sequence [B, N, C] is the tensor n = B, h = N, w = 1, c = C) ---------------
    def linear(self, x, w, b=None, act=None, res=None, out=None):
        """nn.Linear over the channels of every pixel / token: w [cout, cin] -> a 1x1 convolution on the MFMA engine"""
        w = np.asarray(w, np.float32)
        return self.conv(x, w.reshape(w.shape[0], w.shape[1], 1, 1), b, act=act, res=res, res_mode=1 if res is not None else 0, out=out)

    def layernorm(self, x, gamma, beta, eps, out=None):
        if out is None:
            out = self.buffer(x.n, x.h, x.w, x.c)
        assert x.c % 4 == 0 and gamma.size == x.c and beta.size == x.c
        w_h, w_n = self._w(gamma, gamma)
        b_h, b_n = self._w(beta, beta)
        a_h, a_n = self._w(np.array([eps, 0, 0, 0], np.float32), np.array([eps, 0, 0, 0], np.float32))
        return self._emit(OP_LAYERNORM, x, None, out, w_off=w_h, b_off=b_h, aux_off=a_h, nat=dict(w_off=w_n, b_off=b_n, aux_off=a_n))

    def attention(self, qkv, heads, grid=None, rel_table=None):
        """softmax(q k^T + bias) v per head; qkv [n, N, 1, 3 * heads * d] with q pre-scaled; rel_table [(2gh-1)(2gw-1)+3, heads] (BEiT) with
        grid = (gh, gw) and N == gh * gw + 1 (token 0 = class token), or None"""
        d = qkv.c // (3 * heads)
        assert qkv.w == 1 and qkv.c == 3 * heads * d and d in (32, 64, 128), (qkv.shape, heads)
        out = self.buffer(qkv.n, qkv.h, 1, heads * d)
        a_h = a_n = -1
        gh, gw = grid if grid is not None else (0, 0)
        if rel_table is not None:
            rel_table = np.asarray(rel_table, np.float32)
            assert rel_table.shape == ((2 * gh - 1) * (2 * gw - 1) + 3, heads) and qkv.h == gh * gw + 1, (rel_table.shape, grid, qkv.h)
            a_h, a_n = self._w(np.ascontiguousarray(rel_table.T), rel_table)     # device: [heads][T] (a block reads one head's row)
        self.flops += 4 * qkv.n * heads * qkv.h * qkv.h * d
        self.flops_exec += 4 * qkv.n * heads * qkv.h * qkv.h * d
        return self._emit(OP_ATTENTION, qkv, None, out, groups=heads, cin_g=d, kh=gh, kw=gw, aux_off=a_h, nat=dict(aux_off=a_n))

    def tokens_assemble(self, patches, cls):
        """[n, gh, gw, c] patch embedding -> [n, gh*gw + 1, 1, c] with the class token in row 0"""
        out = self.buffer(patches.n, patches.h * patches.w + 1, 1, patches.c)
        a_h, a_n = self._w(cls, cls)
        return self._emit(OP_TOKENS, patches, None, out, flags=0, aux_off=a_h, nat=dict(aux_off=a_n))

    def tokens_readout(self, tokens, grid, project=True):
        """MiDaS readout: drop the class token (Slice) or concatenate it to every patch token (ProjectReadout's input) -> [n, gh, gw, c or 2c]"""
        gh, gw = grid
        assert tokens.w == 1 and tokens.h == gh * gw + 1
        out = self.buffer(tokens.n, gh, gw, tokens.c * (2 if project else 1))
        return self._emit(OP_TOKENS, tokens, None, out, flags=1 if project else 2, kh=gh, kw=gw)

    def depth_to_space(self, x, k):
        assert x.c % (k * k) == 0 and (x.c // (k * k)) % 4 == 0
        out = self.buffer(x.n, x.h * k, x.w * k, x.c // (k * k))
        return self._emit(OP_DEPTH_TO_SPACE, x, None, out, stride=k)

    def conv_transpose_nonoverlap(self, x, w, b, k):
        """nn.ConvTranspose2d(cin, cout, kernel_size = stride = k): w [cin, cout, k, k].  Non-overlapping, so it is a 1x1 convolution to
        k*k*cout channels (channel (ky*k + kx)*cout + co <- w[:, co, ky, kx]) followed by the depth-to-space scatter"""
        w = np.asarray(w, np.float32)
        cin, cout = w.shape[0], w.shape[1]
        assert w.shape[2:] == (k, k) and cin == x.c
        w1 = w.transpose(2, 3, 1, 0).reshape(k * k * cout, cin, 1, 1)
        b1 = None if b is None else np.tile(np.asarray(b, np.float32), k * k)
        return self.depth_to_space(self.conv(x, w1, b1), k)

    def crop_rows(self, x, h):
        """the first h rows of a single-image map as a VIEW (torch's negative bottom pad): same buffer, smaller height"""
        assert x.n == 1 and 0 < h <= x.h and x.coff == 0 and x.c == x.buf.c
        b = Buf(1, h, x.w, x.buf.c)
        b.alias = x.buf
        b.first, b.last = x.buf.first, x.buf.last
        self.bufs.append(b)
        return T(self, b, 0, x.c)

    def gavgpool(self, x):
        out = self.buffer(x.n, 1, 1, x.c)
        return self._emit(OP_GAVGPOOL, x, None, out)

    def scale(self, x, s, out=None):
        if out is None:
            out = self.buffer(x.n, x.h, x.w, x.c)
        return self._emit(OP_SCALE, x, s, out)

    def keep(self, t):
        t.buf.keep = True
        return t

    # ---- planning / serialisation -----------------------------------------------------------
    def plan(self):
        """greedy first-fit workspace plan over buffer lifetimes; returns workspace floats"""
        ALIGN = 64
        free, top = [], 0            # free: list of (offset, size)
        by_first, by_last = {}, {}
        for b in self.bufs:
            if b.ext >= 0 or b.first is None or b.alias is not None:
                continue
            b.size = (b.n * b.h * b.w * b.c + ALIGN - 1) // ALIGN * ALIGN
            by_first.setdefault(b.first, []).append(b)
            if not b.keep:
                by_last.setdefault(b.last, []).append(b)
        for i in range(len(self.ops)):
            for b in by_first.get(i, []):
                best = None
                for k, (off, sz) in enumerate(free):
                    if sz >= b.size and (best is None or sz < free[best][1]):
                        best = k
                if best is None:
                    b.offset = top; top += b.size
                else:
                    off, sz = free.pop(best)
                    b.offset = off
                    if sz > b.size:
                        free.append((off + b.size, sz - b.size))
            for b in by_last.get(i, []):
                free.append((b.offset, b.size))
                free.sort()
                merged = []
                for off, sz in free:
                    if merged and merged[-1][0] + merged[-1][1] == off:
                        merged[-1] = (merged[-1][0], merged[-1][1] + sz)
                    else:
                        merged.append((off, sz))
                free = merged
        for b in self.bufs:
            if b.alias is not None:
                b.offset = b.alias.offset
        self.workspace_floats = top
        return top

    def serialise(self, oracle=False):
        """-> (ops ctypes array, tensors ctypes array, weights fp32 ndarray)"""
        if getattr(self, "workspace_floats", None) is None:
            self.plan()
        tens = (CsmTensorDesc * len(self.views))()
        for i, v in enumerate(self.views):
            b = v.buf
            if b.ext >= 0:
                tens[i] = CsmTensorDesc(0, b.ext, b.n, b.h, b.w, v.c, b.c)
            else:
                tens[i] = CsmTensorDesc(b.offset + v.coff, -1, b.n, b.h, b.w, v.c, b.c)
        ops = (CsmOp * len(self.ops))()
        for i, o in enumerate(self.ops):
            d = dict(o)
            nat = d.pop('nat')
            if oracle and nat:
                d.update(nat)
            ops[i] = CsmOp(**{k: int(v) for k, v in d.items()}, tile=0)
        ws = self.w_nat if oracle else self.w_hip
        weights = np.concatenate(ws) if ws else np.zeros(4, np.float32)
        return ops, tens, weights
