#!/usr/bin/env python3
"""Generate tests/golden/net_*.npz from the REFERENCE's own torch modules (build container only).

The reference files are imported by path (ref_loader.py); their parameters/buffers are filled with
cartoonsegmentation_amd.weights.synth_tensor(<reference parameter name>, ...) -- the same closed-form
weights the build's own lowering uses -- then the reference module runs on CPU (eval mode).
Stored: input, output.  No weights and no reference code travel.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE); sys.path.insert(0, ROOT)
import ref_loader  # noqa: E402
from cartoonsegmentation_amd.weights import synth_tensor  # noqa: E402

ref_loader.install_stubs()


def fill_synthetic(module, prefix):
    """assign every parameter / buffer of a reference nn.Module from synth_tensor(prefix+name)"""
    kinds = {}
    for mname, m in module.named_modules():
        pre = mname + '.' if mname else ''
        if isinstance(m, (torch.nn.Conv2d, torch.nn.ConvTranspose2d)):
            kinds[pre + 'weight'] = 'conv_w'; kinds[pre + 'bias'] = 'conv_b'
        elif isinstance(m, torch.nn.BatchNorm2d):
            kinds[pre + 'weight'] = 'bn_gamma'; kinds[pre + 'bias'] = 'bn_beta'
            kinds[pre + 'running_mean'] = 'bn_mean'; kinds[pre + 'running_var'] = 'bn_var'
        elif isinstance(m, torch.nn.PReLU):
            kinds[pre + 'weight'] = 'prelu'
    sd = module.state_dict()
    for name, t in sd.items():
        if name.endswith('num_batches_tracked'):
            continue
        t.copy_(torch.from_numpy(synth_tensor(prefix + name, tuple(t.shape), kinds[name])))
    module.load_state_dict(sd)
    return module.eval()


def isnet_cases():
    ref_loader._bare("animeinsseg"); ref_loader._bare("animeinsseg.models"); ref_loader._bare("animeinsseg.models.animeseg_refine")
    m = ref_loader.load_by_path("animeinsseg.models.animeseg_refine.isnet", "animeinsseg/models/animeseg_refine/isnet.py")
    net = fill_synthetic(m.ISNetDIS(in_ch=4), 'isnet.')
    for tag, (h, w) in (('64x64', (64, 64)), ('90x74', (90, 74))):
        g = np.random.default_rng(100 + h)
        x = g.uniform(0, 1, (1, 4, h, w)).astype(np.float32)
        with torch.no_grad():
            d1 = net(torch.from_numpy(x))[0][0]           # animeinsseg/__init__.py:653 uses [0][0]
        np.savez_compressed(os.path.join(HERE, 'net_isnet_%s.npz' % tag), x=x, d1=d1.numpy())
        print('isnet', tag, float(d1.mean()), float(d1.std()))


def leres_cases():
    for n in ("depth_modules", "depth_modules.leres", "depth_modules.leres.leres"):
        ref_loader._bare(n)
    ref_loader.load_by_path("depth_modules.leres.leres.Resnet", "depth_modules/leres/leres/Resnet.py")
    ref_loader.load_by_path("depth_modules.leres.leres.Resnext_torch", "depth_modules/leres/leres/Resnext_torch.py")
    ref_loader.load_by_path("depth_modules.leres.leres.network_auxi", "depth_modules/leres/leres/network_auxi.py")
    ref_loader.load_by_path("depth_modules.leres.leres.net_tools", "depth_modules/leres/leres/net_tools.py")
    m = ref_loader.load_by_path("depth_modules.leres.leres.multi_depth_model_woauxi",
                                "depth_modules/leres/leres/multi_depth_model_woauxi.py")
    net = fill_synthetic(m.RelDepthModel(backbone='resnext101'), 'leres.')
    for tag, (h, w) in (('64x64', (64, 64)), ('96x64', (96, 64))):
        g = np.random.default_rng(200 + h)
        x = g.normal(0, 1, (1, 3, h, w)).astype(np.float32)
        with torch.no_grad():
            y = net.depth_model(torch.from_numpy(x))
        np.savez_compressed(os.path.join(HERE, 'net_leres_%s.npz' % tag), x=x, y=y.numpy())
        print('leres', tag, float(y.mean()), float(y.std()))


def inpaint_case():
    """whole Inpaint.forward of the reference (incl. its render_pointcloud CUDA text through cuda_on_cpu.h)"""
    mu, co, cu = ref_loader.load_warp_modules()
    m = ref_loader.load_by_path("anime_3dkenburns.models.pointcloud_inpainting", "anime_3dkenburns/models/pointcloud_inpainting.py")
    net = fill_synthetic(m.Inpaint(), 'inpaint.')
    H, W = 32, 40
    g = np.random.default_rng(77)
    img = g.uniform(0, 1, (1, 3, H, W)).astype(np.float32)
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float32)
    disp = (10 + 8 * yy / H + 14.0 / (1.0 + np.exp(np.clip((np.hypot(xx - 18, yy - 14) - 8) / 0.75, -60, 60)))).astype(np.float32)[None, None]
    disp = disp / disp.max() * 40.0
    seg = (np.hypot(xx - 18, yy - 14) < 8).astype(np.float32)[None, None].repeat(3, 1)
    shift = np.array([2.5, -1.5, -3.0], np.float32).reshape(1, 3, 1)
    common = {'fltFocal': W / 2.0, 'fltBaseline': 40.0, 'intWidth': W, 'intHeight': H}
    with torch.no_grad():
        out = net(torch.from_numpy(img), torch.from_numpy(disp.astype(np.float32)), torch.from_numpy(shift), common, torch.from_numpy(seg))
    np.savez_compressed(os.path.join(HERE, 'net_inpaint_32x40.npz'), img=img, disp=disp.astype(np.float32), shift=shift, seg=seg,
                        existing=out['tenExisting'].numpy(), image=out['tenImage'].numpy(), disparity=out['tenDisparity'].numpy(),
                        segmasks=out['segmasks'].numpy())
    print('inpaint', float(out['tenExisting'].mean()), float(out['tenImage'].mean()), float(out['tenDisparity'].mean()))


def disparity_case():
    """Semantics (VGG19-BN slices) + Disparity GridNet of anime_3dkenburns/models/disparity_estimation.py.  torchvision is absent:
    `torchvision.models.vgg19_bn` is supplied with torchvision's published cfg 'E' + batch norm ([EXT]); which of its entries are
    used, the ceil-mode pools, the input flip / normalisation and the whole GridNet are the reference's own text."""
    import types
    nn = torch.nn

    def vgg19_bn(pretrained=False, **kw):
        layers, c = [], 3
        for v in [64, 64, 'M', 128, 128, 'M', 256, 256, 256, 256, 'M', 512, 512, 512, 512, 'M', 512, 512, 512, 512, 'M']:
            if v == 'M':
                layers.append(nn.MaxPool2d(kernel_size=2, stride=2))
            else:
                layers += [nn.Conv2d(c, v, kernel_size=3, padding=1), nn.BatchNorm2d(v), nn.ReLU(inplace=True)]
                c = v
        return types.SimpleNamespace(features=nn.Sequential(*layers))
    sys.modules['torchvision.models'].vgg19_bn = vgg19_bn
    sys.modules['torchvision'].models = sys.modules['torchvision.models']
    ref_loader._bare("anime_3dkenburns"); ref_loader._bare("anime_3dkenburns.models")
    m = ref_loader.load_by_path("anime_3dkenburns.models.disparity_estimation", "anime_3dkenburns/models/disparity_estimation.py")
    sem = fill_synthetic(m.Semantics(), 'semantics.')
    dis = fill_synthetic(m.Disparity(), 'disparity.')
    for tag, (h, w) in (('96x64', (96, 64)), ('64x128', (64, 128))):
        g = np.random.default_rng(300 + h)
        x = g.uniform(0, 1, (1, 3, h, w)).astype(np.float32)
        with torch.no_grad():
            s_ = sem(torch.from_numpy(x))
            d = dis(torch.from_numpy(x), s_)
        np.savez_compressed(os.path.join(HERE, 'net_disparity_%s.npz' % tag), x=x, sem=s_.numpy(), disp=d.numpy())
        print('disparity', tag, tuple(s_.shape), float(s_.mean()), tuple(d.shape), float(d.mean()), float(d.std()))


if __name__ == '__main__':
    which = sys.argv[1:] or ['isnet', 'leres']
    if 'disparity' in which:
        disparity_case()
    if 'isnet' in which:
        isnet_cases()
    if 'leres' in which:
        leres_cases()
    if 'inpaint' in which:
        inpaint_case()

