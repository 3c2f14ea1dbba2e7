"""A deterministic STAND-IN for the MiDaS DPT-BEiT-L network inside MidasCore (depth_modules/zoedepth/models/base_models/midas.py:341
loads the real one with torch.hub; it is not vendored).  Pure elementwise torch ops on the prepared input, shaped like the
tensors MidasCore's hooks deliver (rel_depth [B,h,w]; out_conv [B,32,h,w]; l4_rn [B,256,h/32,w/32]; r4..r1 [B,256,h/16..h/2,...]).
Used by the fixture generator (wrapped in nn.Modules so the REFERENCE's MidasCore hooks capture them) and by the GPU test (as the
plugged `core`), so both sides see the same stand-in features and everything AROUND the core is what gets compared."""
import torch
import torch.nn.functional as F


def _feat(gray, scale, channels, phase):
    g = F.avg_pool2d(gray, scale) if scale > 1 else gray
    c = torch.arange(channels, dtype=torch.float32, device=gray.device).view(1, channels, 1, 1)
    return torch.sin(g * (0.5 + 0.01 * c) + (phase + 0.1 * c))


def layer4_rn(xp):
    return _feat(xp.mean(1, keepdim=True), 32, 256, 0.1)


def refinenet(xp, level):                      # level 4..1 -> 1/16 .. 1/2 resolution
    return _feat(xp.mean(1, keepdim=True), 2 ** level, 256, 0.1 + 0.1 * (5 - level))


def out_conv(xp):
    return torch.relu(_feat(xp.mean(1, keepdim=True), 1, 32, 0.3))


def rel_depth(xp):
    return torch.relu(xp.mean(1) * 2.0 + 3.0)


def core(xp):
    """the plug for cartoonsegmentation_amd.zoedepth.ZoeDepth: (rel_depth, [out_conv, l4_rn, r4, r3, r2, r1])"""
    return rel_depth(xp), [out_conv(xp), layer4_rn(xp)] + [refinenet(xp, lv) for lv in (4, 3, 2, 1)]
