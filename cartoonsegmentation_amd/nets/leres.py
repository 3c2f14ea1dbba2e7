"""LeReS relative-depth CNN (ResNeXt101-32x8d encoder + FTB/FFM/AO decoder) -> layer program.

Mirrors depth_modules/leres/leres (reference): Resnext_torch.py Bottleneck :70-117, ResNet._forward_impl
:196-221 (features = layer1..4 outputs), network_auxi.py Decoder :15-62, FTB :100-125, FFM :195-218,
AO :245-266, wrapper multi_depth_model_woauxi.py:22-33.

Two reference subtleties kept on purpose:
  * FTB's conv_branch starts with nn.ReLU(inplace=True) (network_auxi.py:108), which overwrites x in
    place, so FTB computes relu(t + branch(relu(t))) with t := relu(conv1(x))  -- conv1 gets a fused ReLU
    and the skip is the post-ReLU tensor.
  * nn.Upsample(scale_factor=2, align_corners=True) everywhere in the decoder.
Grouped 3x3 convs (32 groups) run on the MFMA kernel as 32-channel block-diagonal super-groups.
"""
from ..program import Program
from ..weights import conv_bn, conv_plain

LAYERS = (3, 4, 23, 3)


def _bottleneck(p, ws, name, x, planes, stride, downsample):
    width, out_ch = planes * 4, planes * 4          # int(planes*(8/64))*32 ; expansion 4
    w1, b1 = conv_bn(ws, name + '.conv1', name + '.bn1', width, x.c, 1)
    t = p.conv(x, w1, b1, act='relu')
    w2, b2 = conv_bn(ws, name + '.conv2', name + '.bn2', width, width // 32, 3)
    t = p.conv(t, w2, b2, stride=stride, pad=1, groups=32, act='relu')
    if downsample:
        wd, bd = conv_bn(ws, name + '.downsample.0', name + '.downsample.1', out_ch, x.c, 1)
        idt = p.conv(x, wd, bd, stride=stride)
    else:
        idt = x
    w3, b3 = conv_bn(ws, name + '.conv3', name + '.bn3', out_ch, width, 1)
    return p.conv(t, w3, b3, act='relu', res=idt, res_mode=1)        # relu(bn3(conv3) + identity)


def _ftb(p, ws, name, x, mid):
    w1, b1 = conv_plain(ws, name + '.conv1', mid, x.c, 3)
    t = p.conv(x, w1, b1, pad=1, act='relu')                          # inplace ReLU of conv_branch[0] hits x too
    wa, ba = conv_bn(ws, name + '.conv_branch.1', name + '.conv_branch.2', mid, mid, 3, conv_bias=True)
    u = p.conv(t, wa, ba, pad=1, act='relu')
    wb, bb = conv_plain(ws, name + '.conv_branch.4', mid, mid, 3)
    return p.conv(u, wb, bb, pad=1, act='relu', res=t, res_mode=1)


def _ffm(p, ws, name, low, high, mid, out_ch):
    x = _ftb(p, ws, name + '.ftb1', low, mid)
    x = p.add(x, high)
    x = _ftb(p, ws, name + '.ftb2', x, out_ch)
    return p.bilinear(x, (x.h * 2, x.w * 2), align_corners=True)


def build_leres(ws, n, h, w):
    """ext tensors: [0] input NCHW [n,3,h,w] (RGB, ImageNet-normalised), [1] output depth NCHW [n,1,h,w]"""
    assert h % 32 == 0 and w % 32 == 0, "LeReS input must be a multiple of 32 (reference: scaledown_maxsize(..., divisior=32))"
    p = Program("leres")
    x_ext = p.ext_nchw(n, 3, h, w)
    y_ext = p.ext_nchw(n, 1, h, w)
    x = p.to_nhwc(x_ext)
    E = 'depth_model.encoder_modules.encoder.'
    w0, b0 = conv_bn(ws, E + 'conv1', E + 'bn1', 64, 3, 7)
    t = p.conv(x, w0, b0, stride=2, pad=3, act='relu')
    t = p.maxpool(t, 3, 2, 1)
    feats = []
    for li, (planes, blocks) in enumerate(zip((64, 128, 256, 512), LAYERS)):
        for bi in range(blocks):
            stride = 2 if (bi == 0 and li > 0) else 1
            t = _bottleneck(p, ws, '%slayer%d.%d' % (E, li + 1, bi), t, planes, stride, downsample=(bi == 0))
        feats.append(t)
    D = 'depth_model.decoder_modules.'
    x32 = _ftb(p, ws, D + 'conv', feats[3], 512)
    wc, bc = conv_plain(ws, D + 'conv1', 256, 512, 3)
    x32 = p.conv(x32, wc, bc, pad=1)
    x16 = p.bilinear(x32, (x32.h * 2, x32.w * 2), align_corners=True)
    x8 = _ffm(p, ws, D + 'ffm2', feats[2], x16, 256, 256)
    x4 = _ffm(p, ws, D + 'ffm1', feats[1], x8, 256, 256)
    x2 = _ffm(p, ws, D + 'ffm0', feats[0], x4, 256, 256)
    wa, ba = conv_bn(ws, D + 'outconv.adapt_conv.0', D + 'outconv.adapt_conv.1', 128, 256, 3, conv_bias=True)
    a = p.conv(x2, wa, ba, pad=1, act='relu')
    wo, bo = conv_plain(ws, D + 'outconv.adapt_conv.3', 1, 128, 3)
    o = p.conv(a, wo, bo, pad=1)
    p.bilinear(o, (h, w), align_corners=True, out=y_ext)
    p.plan()
    return p
