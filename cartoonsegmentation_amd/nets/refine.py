"""Ken Burns disparity `Refine` net -> layer program (anime_3dkenburns/models/disparity_refinement.py:81-127).
Program inputs are the ALREADY mean/std-normalised image [1,3,H,W] and disparity [1,1,h,w] (:98-107); the output is
netRefine's raw map [1,1,H,W]; de-normalisation and threshold(0) (:122-126) are two scalar ops on the host side."""
from ..program import Program
from .gridblocks import basic, downsample, upsample


def build_refine(ws, H, W, h, w):
    p = Program("refine")
    img_ext = p.ext_nchw(1, 3, H, W)
    dsp_ext = p.ext_nchw(1, 1, h, w)
    out_ext = p.ext_nchw(1, 1, H, W)
    img, dsp = p.to_nhwc(img_ext), p.to_nhwc(dsp_ext)
    s2 = ((H + 1) // 2, (W + 1) // 2)
    s3 = ((s2[0] + 1) // 2, (s2[1] + 1) // 2)
    C3 = p.buffer(1, s3[0], s3[1], 192)      # cat([tenImageThr, tenUpsample], 1)
    C2 = p.buffer(1, s2[0], s2[1], 144)      # cat([tenImageTwo, tenUpsample], 1)
    C1 = p.buffer(1, H, W, 72)               # cat([tenImageOne, tenUpsample], 1)
    one = basic(p, ws, 'netImageOne', 'conv-relu-conv', (3, 24, 24), img, out=C1.slice(0, 24))
    two = downsample(p, ws, 'netImageTwo', (24, 48, 48), one, out=C2.slice(0, 48))
    downsample(p, ws, 'netImageThr', (48, 96, 96), two, out=C3.slice(0, 96))
    d1 = basic(p, ws, 'netDisparityOne', 'conv-relu-conv', (1, 96, 96), dsp)
    if (d1.h, d1.w) != s3:
        p.bilinear(d1, s3, out=C3.slice(96, 192))
    else:
        p.copy(d1, C3.slice(96, 192))
    u = upsample(p, ws, 'netDisparityTwo', (192, 96, 96), C3)
    if (u.h, u.w) != s2:
        p.bilinear(u, s2, out=C2.slice(48, 144))
    else:
        p.copy(u, C2.slice(48, 144))
    u = upsample(p, ws, 'netDisparityThr', (144, 48, 48), C2)
    if (u.h, u.w) != (H, W):
        p.bilinear(u, (H, W), out=C1.slice(24, 72))
    else:
        p.copy(u, C1.slice(24, 72))
    f = basic(p, ws, 'netDisparityFou', 'conv-relu-conv', (72, 24, 24), C1)
    r = basic(p, ws, 'netRefine', 'conv-relu-conv', (24, 24, 1), f)
    p.to_nchw(r, out_ext)
    p.plan()
    return p
