"""sniklaus 3D-Ken-Burns disparity estimator (`depth_est: default`) -> two layer programs.

  semantics : `Semantics` (anime_3dkenburns/models/disparity_estimation.py:80-116): channel flip + ImageNet normalisation, then
              torchvision's vgg19_bn.features[0:39] with its four max-pools replaced by ceil_mode ones.  The layer list is
              torchvision's cfg 'E' with batch norm (conv3x3 + BN + ReLU per entry) [EXT: torchvision is not installed; the slice
              indices 0:3 ... 36:39 of the reference text fix which entries are used].  Parameter names follow the reference module:
              netVgg.<slice>.<torchvision index>.{weight,bias,running_mean,running_var}.
  disparity : `Disparity` GridNet (:118-193): 6 rows (32, 48, 64, 512, 512, 512 channels) x 4 columns of pre-activation PReLU blocks
              (the Basic / Downsample / Upsample of gridblocks.py), image stem 7x7 stride 2, semantics injected at row 3, output
              Basic(32,32,1) + threshold(0) at HALF the input resolution.
The caller (kenburns.py::_depth_est_default) resizes the image to <= 512 (models/__init__.py:43-52) first.
Odd feature-map sizes (the reference crops the x2-up-sampled map with a negative pad, :172-173) are supported along H only.
"""
from ..program import Program
from ..weights import conv_bn, conv_plain
from .gridblocks import basic, downsample, upsample

ROWS = (32, 48, 64, 512, 512, 512)
# torchvision vgg19_bn.features: (slice index in the reference's Sequential, torchvision conv index, out channels) ; 'M' = ceil-mode pool
VGG = ((0, 0, 64), (1, 3, 64), 'M', (3, 7, 128), (4, 10, 128), 'M', (6, 14, 256), (7, 17, 256), (8, 20, 256), (9, 23, 256), 'M',
       (11, 27, 512), (12, 30, 512), (13, 33, 512), (14, 36, 512), 'M')
MEAN, STD = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)


def build_semantics(ws, H, W):
    """ext [0]: image NCHW [1,3,H,W] as the pipeline holds it (BGR, 0..1); ext [1]: features NCHW [1,512,ceil(H/16),ceil(W/16)]"""
    import numpy as np
    p = Program("semantics")
    x_ext = p.ext_nchw(1, 3, H, W)
    x = p.to_nhwc(x_ext)
    # tenInput.flip([1]) - mean, * (1/std)  (:106-108) as a 1x1 conv: weight = permutation / std, bias = -mean / std.
    # ((x - m) * (1/s) and x * (1/s) - m/s differ in the last ulp; the fixture tolerance covers it)
    wn = np.zeros((4, 4, 1, 1), np.float32); bn = np.zeros(4, np.float32)
    for c in range(3):
        wn[c, 2 - c, 0, 0] = np.float32(1.0 / STD[c]); bn[c] = np.float32(-MEAN[c] * np.float32(1.0 / STD[c]))
    t = p.conv(x, wn, bn)
    first = True
    for e in VGG:
        if e == 'M':
            t = p.maxpool(t, 2, 2, 0, ceil_mode=True)
            continue
        si, li, cout = e
        name = 'netVgg.%d.%d' % (si, li)
        w, b = conv_bn(ws, name, 'netVgg.%d.%d' % (si, li + 1), cout, 3 if first else t.c, 3, conv_bias=True)
        t = p.conv(t, w, b, pad=1, act='relu')
        first = False
    y_ext = p.ext_nchw(1, 512, t.h, t.w)
    p.to_nchw(t, y_ext)
    p.plan()
    return p


def build_disparity(ws, H, W):
    """ext [0]: image NCHW [1,3,H,W]; ext [1]: semantics NCHW [1,512,hs,ws]; ext [2]: disparity NCHW [1,1,ceil(H/2),ceil(W/2)]
    (threshold(0) applied)"""
    p = Program("disparity")
    img_ext = p.ext_nchw(1, 3, H, W)
    h2, w2 = (H + 1) // 2, (W + 1) // 2                            # 7x7 stride 2 pad 3
    sizes = [(h2, w2)]
    for _ in range(5):
        sizes.append(((sizes[-1][0] + 1) // 2, (sizes[-1][1] + 1) // 2))
    sem_ext = p.ext_nchw(1, 512, sizes[3][0], sizes[3][1])
    out_ext = p.ext_nchw(1, 1, h2, w2)
    x, sem = p.to_nhwc(img_ext), p.to_nhwc(sem_ext)
    wi, bi = conv_plain(ws, 'netImage', 32, 3, 7)
    col = [p.conv(x, wi, bi, stride=2, pad=3)]
    wsm, bsm = conv_plain(ws, 'netSemantics', 512, 512, 3)
    semf = p.conv(sem, wsm, bsm, pad=1)
    for r in range(1, 6):                                          # column 0: down the rows; row 3 += netSemantics(semantics)
        col.append(downsample(p, ws, '%dx0 - %dx0' % (r - 1, r), (ROWS[r - 1], ROWS[r], ROWS[r]), col[r - 1], res=semf if r == 3 else None))
    for r in range(6):                                             # column 1: lateral + the down-sampled row above
        lat = basic(p, ws, '%dx0 - %dx1' % (r, r), 'relu-conv-relu-conv', (ROWS[r],) * 3, col[r])
        col[r] = lat if r == 0 else downsample(p, ws, '%dx1 - %dx1' % (r - 1, r), (ROWS[r - 1], ROWS[r], ROWS[r]), col[r - 1], res=lat)
    for c in (2, 3):                                               # columns 2, 3: lateral + the up-sampled row below
        for r in range(5, -1, -1):
            lat = basic(p, ws, '%dx%d - %dx%d' % (r, c - 1, r, c), 'relu-conv-relu-conv', (ROWS[r],) * 3, col[r])
            if r == 5:
                col[r] = lat
                continue
            below = col[r + 1]
            if below.h * 2 == lat.h and below.w * 2 == lat.w:
                col[r] = upsample(p, ws, '%dx%d - %dx%d' % (r + 1, c, r, c), (ROWS[r + 1], ROWS[r], ROWS[r]), below, res=lat)
            else:                                                  # pad [0,0,0,-1] / [0,-1,0,0]: drop the last row / column of the up-sampled map
                up = upsample(p, ws, '%dx%d - %dx%d' % (r + 1, c, r, c), (ROWS[r + 1], ROWS[r], ROWS[r]), below)
                col[r] = p.add(up, lat)
    d = basic(p, ws, 'netDisparity', 'conv-relu-conv', (32, 32, 1), col[0])
    d = p.act(d, 'relu')                                           # threshold(input, 0.0, 0.0)
    p.to_nchw(d, out_ext)
    p.plan()
    return p
