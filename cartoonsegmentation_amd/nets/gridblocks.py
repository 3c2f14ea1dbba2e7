"""PReLU conv blocks shared by the Ken Burns `Refine` and `Inpaint` nets
(anime_3dkenburns/models/disparity_refinement.py:5-79 == pointcloud_inpainting.py:5-79: Basic / Downsample / Upsample).
nn.Sequential index names are kept so the reference state_dicts import unchanged."""
from ..weights import conv_plain


def _prelu(ws, name, c):
    return ws.get(name + '.weight', (c,), 'prelu')


def basic(p, ws, name, kind, ch, x, out=None, res_extra=None):
    """Basic(strType, [c0,c1,c2]): main(x) + (x | shortcut(x))   [+ res_extra is NOT fused; callers add explicitly]"""
    c0, c1, c2 = ch
    if c0 != c2:
        wsx, bsx = conv_plain(ws, name + '.netShortcut', c2, c0, 1)
        short = p.conv(x, wsx, bsx)
    else:
        short = x
    if kind == 'relu-conv-relu-conv':
        a = p.act(x, 'prelu', slope=_prelu(ws, name + '.netMain.0', c0))
        w1, b1 = conv_plain(ws, name + '.netMain.1', c1, c0, 3)
        t = p.conv(a, w1, b1, pad=1, act='prelu', slope=_prelu(ws, name + '.netMain.2', c1))
        w2, b2 = conv_plain(ws, name + '.netMain.3', c2, c1, 3)
    else:   # 'conv-relu-conv'
        w1, b1 = conv_plain(ws, name + '.netMain.0', c1, c0, 3)
        t = p.conv(x, w1, b1, pad=1, act='prelu', slope=_prelu(ws, name + '.netMain.1', c1))
        w2, b2 = conv_plain(ws, name + '.netMain.2', c2, c1, 3)
    return p.conv(t, w2, b2, pad=1, res=short, res_mode=2, out=out)


def downsample(p, ws, name, ch, x, res=None, out=None):
    c0, c1, c2 = ch
    a = p.act(x, 'prelu', slope=_prelu(ws, name + '.netMain.0', c0))
    w1, b1 = conv_plain(ws, name + '.netMain.1', c1, c0, 3)
    t = p.conv(a, w1, b1, stride=2, pad=1, act='prelu', slope=_prelu(ws, name + '.netMain.2', c1))
    w2, b2 = conv_plain(ws, name + '.netMain.3', c2, c1, 3)
    return p.conv(t, w2, b2, pad=1, res=res, res_mode=2, out=out)


def upsample(p, ws, name, ch, x, res=None, out=None):
    c0, c1, c2 = ch
    a = p.bilinear(x, (x.h * 2, x.w * 2), align_corners=False, act='prelu', slope=_prelu(ws, name + '.netMain.1', c0))   # Upsample + PReLU, one pass
    w1, b1 = conv_plain(ws, name + '.netMain.2', c1, c0, 3)
    t = p.conv(a, w1, b1, pad=1, act='prelu', slope=_prelu(ws, name + '.netMain.3', c1))
    w2, b2 = conv_plain(ws, name + '.netMain.4', c2, c1, 3)
    return p.conv(t, w2, b2, pad=1, res=res, res_mode=2, out=out)
