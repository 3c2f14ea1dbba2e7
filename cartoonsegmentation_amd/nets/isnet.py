"""ISNetDIS(in_ch=4) mask refiner -> layer program.

Mirrors animeinsseg/models/animeseg_refine/isnet.py (reference): REBNCONV :95-108, RSU7 :118-198,
RSU6/5/4 (same pattern, :201-365), RSU4F :368-407, ISNetDIS :524-645.  Only d1 (side1 of stage1d,
upsampled to the input size) is built: the caller uses `refinenet(batch)[0][0]`
(animeinsseg/__init__.py:653); side2..6 and pool_in are dead for inference.
torch.cat is realised by writing producers into channel slices of one buffer.
"""
from ..program import Program
from ..weights import conv_bn, conv_plain


def _rebn(p, ws, name, x, cout, dil=1, out=None, res=None, res_mode=0):
    w, b = conv_bn(ws, name + '.conv_s1', name + '.bn_s1', cout, x.c, 3, conv_bias=True)
    return p.conv(x, w, b, stride=1, pad=dil, dil=dil, act='relu', out=out, res=res, res_mode=res_mode)


def _rsu(p, ws, name, x, L, mid, out_ch, out=None):
    """RSU-L with pooling (L = 7, 6, 5, 4)"""
    hxin = _rebn(p, ws, name + '.rebnconvin', x, out_ch)
    cats, h = {}, hxin
    for k in range(1, L):
        B = p.buffer(h.n, h.h, h.w, 2 * mid)              # [ upsampled deeper | hx_k ]
        hxk = _rebn(p, ws, '%s.rebnconv%d' % (name, k), h, mid, out=B.slice(mid, 2 * mid))
        cats[k] = B
        h = p.maxpool(hxk, 2, 2, ceil_mode=True) if k < L - 1 else hxk
    _rebn(p, ws, '%s.rebnconv%d' % (name, L), h, mid, dil=2, out=cats[L - 1].slice(0, mid))
    d = _rebn(p, ws, '%s.rebnconv%dd' % (name, L - 1), cats[L - 1], mid)
    for k in range(L - 2, 0, -1):
        B = cats[k]
        p.bilinear(d, (B.h, B.w), align_corners=False, out=B.slice(0, mid))
        if k > 1:
            d = _rebn(p, ws, '%s.rebnconv%dd' % (name, k), B, mid)
        else:
            d = _rebn(p, ws, name + '.rebnconv1d', B, out_ch, out=out, res=hxin, res_mode=2)   # hx1d + hxin
    return d


def _rsu4f(p, ws, name, x, mid, out_ch, out=None):
    hxin = _rebn(p, ws, name + '.rebnconvin', x, out_ch)
    B1, B2, B3 = (p.buffer(x.n, x.h, x.w, 2 * mid) for _ in range(3))
    hx1 = _rebn(p, ws, name + '.rebnconv1', hxin, mid, 1, out=B1.slice(mid, 2 * mid))
    hx2 = _rebn(p, ws, name + '.rebnconv2', hx1, mid, 2, out=B2.slice(mid, 2 * mid))
    hx3 = _rebn(p, ws, name + '.rebnconv3', hx2, mid, 4, out=B3.slice(mid, 2 * mid))
    _rebn(p, ws, name + '.rebnconv4', hx3, mid, 8, out=B3.slice(0, mid))
    _rebn(p, ws, name + '.rebnconv3d', B3, mid, 4, out=B2.slice(0, mid))
    _rebn(p, ws, name + '.rebnconv2d', B2, mid, 2, out=B1.slice(0, mid))
    return _rebn(p, ws, name + '.rebnconv1d', B1, out_ch, 1, out=out, res=hxin, res_mode=2)


def build_isnet(ws, n, h, w, in_ch=4):
    """returns Program with ext tensors: [0] input NCHW [n,in_ch,h,w], [1] output d1 logits NCHW [n,1,h,w]"""
    p = Program("isnet")
    x_ext = p.ext_nchw(n, in_ch, h, w)
    y_ext = p.ext_nchw(n, 1, h, w)
    x = p.to_nhwc(x_ext)
    wi, bi = conv_plain(ws, 'conv_in', 64, in_ch, 3)
    hxin = p.conv(x, wi, bi, stride=2, pad=1)

    def pool(t):
        return p.maxpool(t, 2, 2, ceil_mode=True)

    def osz(v):
        return -(-v // 2)
    s1 = (hxin.h, hxin.w)
    sizes = [s1]
    for _ in range(5):
        sizes.append((osz(sizes[-1][0]), osz(sizes[-1][1])))
    C1 = p.buffer(n, *sizes[0], 128); C2 = p.buffer(n, *sizes[1], 256); C3 = p.buffer(n, *sizes[2], 512)
    C4 = p.buffer(n, *sizes[3], 1024); C5 = p.buffer(n, *sizes[4], 1024)
    hx1 = _rsu(p, ws, 'stage1', hxin, 7, 32, 64, out=C1.slice(64, 128))
    hx2 = _rsu(p, ws, 'stage2', pool(hx1), 6, 32, 128, out=C2.slice(128, 256))
    hx3 = _rsu(p, ws, 'stage3', pool(hx2), 5, 64, 256, out=C3.slice(256, 512))
    hx4 = _rsu(p, ws, 'stage4', pool(hx3), 4, 128, 512, out=C4.slice(512, 1024))
    hx5 = _rsu4f(p, ws, 'stage5', pool(hx4), 256, 512, out=C5.slice(512, 1024))
    hx6 = _rsu4f(p, ws, 'stage6', pool(hx5), 256, 512)
    p.bilinear(hx6, sizes[4], out=C5.slice(0, 512))
    hx5d = _rsu4f(p, ws, 'stage5d', C5, 256, 512)
    p.bilinear(hx5d, sizes[3], out=C4.slice(0, 512))
    hx4d = _rsu(p, ws, 'stage4d', C4, 4, 128, 256)
    p.bilinear(hx4d, sizes[2], out=C3.slice(0, 256))
    hx3d = _rsu(p, ws, 'stage3d', C3, 5, 64, 128)
    p.bilinear(hx3d, sizes[1], out=C2.slice(0, 128))
    hx2d = _rsu(p, ws, 'stage2d', C2, 6, 32, 64)
    p.bilinear(hx2d, sizes[0], out=C1.slice(0, 64))
    hx1d = _rsu(p, ws, 'stage1d', C1, 7, 16, 64)
    wsd, bsd = conv_plain(ws, 'side1', 1, 64, 3)
    d1 = p.conv(hx1d, wsd, bsd, stride=1, pad=1)
    p.bilinear(d1, (h, w), out=y_ext)           # NHWC with c == 1 is NCHW
    p.plan()
    return p
