"""Weight sources for the net builders.

StateDictWeights : a torch / numpy state_dict with the reference's parameter names (real checkpoints:
                   refine_last.ckpt, res101.pth, rtmdetl_e60.ckpt -> SURVEY.md 5 "checkpoint / resume").
SynthWeights     : closed-form, seed-free-file deterministic weights (no checkpoint ships in this
                   container): a splitmix64 hash of (crc32(name), flat index) -> uniform(-1,1), scaled per kind.
                   The golden generator fills the reference's own modules with the SAME function, so
                   fixtures pin the build against the reference without any weight file travelling.
"""
import zlib

import numpy as np

_M = np.uint64(0xFFFFFFFFFFFFFFFF)


def hash_uniform(name, n):
    """n floats in [-1, 1), deterministic in (name, index)"""
    with np.errstate(over='ignore'):
        seed = np.uint64(zlib.crc32(name.encode()) + 1) * np.uint64(0x9E3779B97F4A7C15)
        x = np.arange(n, dtype=np.uint64) * np.uint64(0xD1342543DE82EF95) + seed
        x ^= x >> np.uint64(30); x *= np.uint64(0xBF58476D1CE4E5B9)
        x ^= x >> np.uint64(27); x *= np.uint64(0x94D049BB133111EB)
        x ^= x >> np.uint64(31)
    return ((x >> np.uint64(11)).astype(np.float64) * (2.0 / 9007199254740992.0) - 1.0)


def synth_tensor(name, shape, kind):
    n = int(np.prod(shape))
    u = hash_uniform(name, n)
    if kind == 'conv_w':      # He-uniform: keeps activations O(1) through ReLU/SiLU stacks
        fan_in = int(np.prod(shape[1:]))
        v = u * np.sqrt(6.0 / fan_in)
    elif kind == 'conv_b':
        v = 0.05 * u
    elif kind == 'bn_gamma':
        v = 1.0 + 0.1 * u
        # the last BN of a ResNe(X)t bottleneck: with gamma ~ 1 every `x + branch(x)` doubles the variance and LeReS's 33 blocks
        # end at |y| ~ 1e5 (the decoder was being verified in a numerically odd regime).  A quarter-size
        # gamma -- the usual "small last gamma" initialisation -- keeps the trunk O(1).
        if '.layer' in name and name.endswith('.bn3.weight'):
            v = 0.25 * v
    elif kind == 'bn_beta':
        v = 0.1 * u
    elif kind == 'bn_mean':
        v = 0.05 * u
    elif kind == 'bn_var':
        v = 1.0 + 0.1 * u * u
    elif kind == 'prelu':
        v = 0.25 + 0.05 * u
    elif kind == 'lin_w':        # transformer linears behind a LayerNorm: unit-variance outputs for unit-variance inputs
        v = u * np.sqrt(3.0 / shape[-1])
    elif kind == 'token':        # class token
        v = 0.5 * u
    elif kind == 'rel_bias':     # relative-position-bias table: logits of order one, so that the bias visibly shapes the softmax
        v = 1.5 * u
    elif kind == 'layer_scale':  # BEiT gamma_1 / gamma_2 (trained values are O(0.1 .. 1))
        v = 0.3 + 0.2 * u
    else:
        raise KeyError(kind)
    return v.astype(np.float32).reshape(shape)


class SynthWeights:
    """closed-form weights.  With CSM_WEIGHTS_PLACEHOLDER=1 (ranks != 0 of a multi-GPU job) only shapes are produced
    (zeros): the packed device buffers are then filled by the RCCL broadcast from rank 0 (shard.broadcast_weights)."""

    def __init__(self, prefix=""):
        import os
        self.prefix = prefix
        self.placeholder = os.environ.get("CSM_WEIGHTS_PLACEHOLDER", "0") == "1"

    def get(self, name, shape, kind):
        if self.placeholder:
            return np.zeros(tuple(shape), np.float32)
        return synth_tensor(self.prefix + name, tuple(shape), kind)


class StateDictWeights:
    def __init__(self, sd, prefix=""):
        self.sd, self.prefix = sd, prefix

    def get(self, name, shape, kind):
        t = self.sd[self.prefix + name]
        a = t.detach().cpu().numpy() if hasattr(t, 'detach') else np.asarray(t)
        a = a.astype(np.float32)
        assert tuple(a.shape) == tuple(shape), (name, a.shape, shape)
        return a


def conv_bn(ws, conv, bn, cout, cin_g, k, conv_bias=False, eps=1e-5):
    """fetch conv (+optional bias) and BN params by reference names and fold them"""
    from .program import fold_bn
    kh, kw = (k, k) if isinstance(k, int) else k
    w = ws.get(conv + '.weight', (cout, cin_g, kh, kw), 'conv_w')
    b = ws.get(conv + '.bias', (cout,), 'conv_b') if conv_bias else None
    g = ws.get(bn + '.weight', (cout,), 'bn_gamma'); be = ws.get(bn + '.bias', (cout,), 'bn_beta')
    m = ws.get(bn + '.running_mean', (cout,), 'bn_mean'); v = ws.get(bn + '.running_var', (cout,), 'bn_var')
    return fold_bn(w, b, g, be, m, v, eps)


def conv_plain(ws, conv, cout, cin_g, k, bias=True):
    kh, kw = (k, k) if isinstance(k, int) else k
    w = ws.get(conv + '.weight', (cout, cin_g, kh, kw), 'conv_w')
    b = ws.get(conv + '.bias', (cout,), 'conv_b') if bias else None
    return w, b
