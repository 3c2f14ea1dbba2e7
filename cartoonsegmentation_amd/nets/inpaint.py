"""Ken Burns point-cloud `Inpaint` GridNet -> two layer programs (anime_3dkenburns/models/pointcloud_inpainting.py:81-203).

  context : netContext on cat(normalised image, normalised disparity)            (:82-87, :133)
  grid    : netInput + 4x4 GridNet + netImage / netDisparity on cat(render, existing)  (:89-110, :147-190)
The forward splat between them (render_pointcloud with C=68, :135), the median-5 mask clean-up (:141) and the
mean/std (de)normalisation are operators / scalar glue in kenburns.py.  H and W must be multiples of 8 (the
reference crops odd up-sampled maps with a negative pad, :170-171; not needed at the benchmark sizes)."""
from ..program import Program
from ..weights import conv_plain
from .gridblocks import basic, downsample, upsample

ROWS = (32, 64, 128, 256)


def build_inpaint_context(ws, H, W):
    p = Program("inpaint_context")
    x_ext = p.ext_nchw(1, 4, H, W)
    y_ext = p.ext_nchw(1, 64, H, W)
    x = p.to_nhwc(x_ext)
    w0, b0 = conv_plain(ws, 'netContext.0', 64, 4, 3)
    t = p.conv(x, w0, b0, pad=1, act='prelu', slope=ws.get('netContext.1.weight', (64,), 'prelu'))
    w1, b1 = conv_plain(ws, 'netContext.2', 64, 64, 3)
    t = p.conv(t, w1, b1, pad=1, act='prelu', slope=ws.get('netContext.3.weight', (64,), 'prelu'))
    p.to_nchw(t, y_ext)
    p.plan()
    return p


def build_inpaint_grid(ws, H, W):
    if H % 8 or W % 8:
        raise NotImplementedError("Inpaint GridNet: H and W must be multiples of 8 (odd-size crop path not built)")
    p = Program("inpaint_grid")
    x_ext = p.ext_nchw(1, 69, H, W)
    img_ext = p.ext_nchw(1, 3, H, W)
    dsp_ext = p.ext_nchw(1, 1, H, W)
    x = p.to_nhwc(x_ext)
    col = [basic(p, ws, 'netInput', 'conv-relu-conv', (69, 32, 32), x)]
    for r in range(1, 4):
        col.append(downsample(p, ws, '%dx0 - %dx0' % (r - 1, r), (ROWS[r - 1], ROWS[r], ROWS[r]), col[r - 1]))
    # column 1: lateral Basic, plus the down-sampled row above (tenColumn[row] += Downsample(tenColumn[row-1]))
    for r in range(4):
        lat = basic(p, ws, '%dx0 - %dx1' % (r, r), 'relu-conv-relu-conv', (ROWS[r],) * 3, col[r])
        col[r] = lat if r == 0 else downsample(p, ws, '%dx1 - %dx1' % (r - 1, r), (ROWS[r - 1], ROWS[r], ROWS[r]), col[r - 1], res=lat)
    for c in (2, 3):
        for r in range(3, -1, -1):
            lat = basic(p, ws, '%dx%d - %dx%d' % (r, c - 1, r, c), 'relu-conv-relu-conv', (ROWS[r],) * 3, col[r])
            col[r] = lat if r == 3 else upsample(p, ws, '%dx%d - %dx%d' % (r + 1, c, r, c), (ROWS[r + 1], ROWS[r], ROWS[r]), col[r + 1], res=lat)
    im = basic(p, ws, 'netImage', 'conv-relu-conv', (32, 32, 3), col[0])
    ds = basic(p, ws, 'netDisparity', 'conv-relu-conv', (32, 32, 1), col[0])
    p.to_nchw(im, img_ext)
    p.to_nchw(ds, dsp_ext)
    p.plan()
    return p
