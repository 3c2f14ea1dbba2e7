"""torch-CPU execution of a lowered Program (TEST INFRASTRUCTURE / bench.py cpu_baseline leg ONLY).

The reference runs its nets through torch.nn (oneDNN on a CPU device, animeinsseg/__init__.py:187-215, depth_modules/leres/__init__.py:
83-147); this module executes the same layers -- natural-layout weights of the Program, BN folded -- with torch.nn.functional on all
host cores.  Two uses:
  * bench.py's `cpu_baseline`: the torch-CPU time of RTMDet-Ins-L / ISNet / LeReS at the benchmark's own sizes, next to
    the fmaf-chain oracle's time;
  * a second, tolerance-level check of the C oracle interpreter (different summation order, different kernels).
Never imported by the product.
"""
import numpy as np
import torch
import torch.nn.functional as F

from cartoonsegmentation_amd import program as P

_ACT = {v: k for k, v in P.ACT.items() if k not in (None,)}


def _act(x, act, slope=None):
    name = _ACT.get(act, 'none')
    if name == 'none':
        return x
    if name == 'relu':
        return F.relu(x)
    if name == 'silu':
        return F.silu(x)
    if name == 'prelu':
        return F.prelu(x, slope)
    if name == 'hsigmoid':
        return F.hardsigmoid(x)
    if name == 'sigmoid':
        return torch.sigmoid(x)
    if name == 'softplus':
        return F.softplus(x)
    if name == 'gelu':
        return F.gelu(x)
    raise KeyError(name)


def _rel_index(gh, gw):
    """timm 0.6.x models/beit.py gen_relative_position_index for a (gh, gw) window (restated): [N, N] indices into the bias table,
    class token first; built with meshgrid / broadcasting (the HIP kernel and the C oracle compute the same index arithmetically)"""
    T = (2 * gh - 1) * (2 * gw - 1) + 3
    coords = torch.stack(torch.meshgrid(torch.arange(gh), torch.arange(gw), indexing='ij')).flatten(1)     # 2, gh*gw
    rel = (coords[:, :, None] - coords[:, None, :]).permute(1, 2, 0).contiguous()
    rel[:, :, 0] += gh - 1
    rel[:, :, 1] += gw - 1
    rel[:, :, 0] *= 2 * gw - 1
    idx = torch.zeros((gh * gw + 1,) * 2, dtype=torch.long)
    idx[1:, 1:] = rel.sum(-1)
    idx[0, 0:] = T - 3
    idx[0:, 0] = T - 2
    idx[0, 0] = T - 1
    return idx


def run_program(prog, ext_arrays, want_views=(), threads=None):
    """ext_arrays: list of float32 numpy arrays NCHW (inputs read, outputs written in place).  Returns {view: ndarray [n,h,w,c]}."""
    if threads:
        torch.set_num_threads(int(threads))
    if getattr(prog, "workspace_floats", None) is None:
        prog.plan()
    w_nat = torch.from_numpy(np.concatenate(prog.w_nat) if prog.w_nat else np.zeros(4, np.float32))
    store = {}                                             # Buf -> NCHW tensor
    last_use = {}
    for i, o in enumerate(prog.ops):
        for k in ('in0', 'in1', 'out'):
            if o[k] >= 0:
                b = prog.views[o[k]].buf
                last_use[b.alias or b] = i
    keep = {(t.buf.alias or t.buf) for t in want_views}

    def buf_tensor(b):
        if b.alias is not None:
            return buf_tensor(b.alias)[:, :, :b.h]
        if b not in store:
            if b.ext >= 0:
                store[b] = torch.from_numpy(ext_arrays[b.ext]).view(b.n, b.c, b.h, b.w)
            else:
                store[b] = torch.zeros((b.n, b.c, b.h, b.w), dtype=torch.float32)
        return store[b]

    def rd(vid):
        v = prog.views[vid]
        return buf_tensor(v.buf)[:, v.coff:v.coff + v.c]

    def wr(vid, val):
        v = prog.views[vid]
        buf_tensor(v.buf)[:, v.coff:v.coff + v.c] = val

    def wt(off, shape):
        n = int(np.prod(shape))
        return w_nat[off:off + n].view(*shape)

    with torch.no_grad():
        for i, o in enumerate(prog.ops):
            kind, nat = o['kind'], o['nat'] or {}
            x = rd(o['in0'])
            vo = prog.views[o['out']]
            if kind == P.OP_CONV:
                g, cin_g, cout_g = nat['groups'], nat['cin_g'], nat['cout_g']
                w = wt(nat['w_off'], (g * cout_g, cin_g, o['kh'], o['kw']))
                b = wt(nat['b_off'], (g * cout_g,)) if nat['b_off'] >= 0 else None
                y = F.conv2d(x, w, b, stride=o['stride'], padding=o['pad'], dilation=o['dil'], groups=g)
                slope = wt(nat['aux_off'], (g * cout_g,)) if nat['aux_off'] >= 0 else None
                if o['res_mode'] == 1:
                    y = y + rd(o['in1'])
                y = _act(y, o['act'], slope)
                if o['res_mode'] == 2:
                    y = y + rd(o['in1'])
                wr(o['out'], y)
            elif kind == P.OP_DWCONV:
                c = vo.c
                w = wt(nat['w_off'], (c, 1, o['kh'], o['kw']))
                b = wt(nat['b_off'], (c,)) if nat['b_off'] >= 0 else None
                wr(o['out'], _act(F.conv2d(x, w, b, stride=o['stride'], padding=o['pad'], dilation=o['dil'], groups=c), o['act']))
            elif kind == P.OP_MAXPOOL:
                k, s, pd = o['kh'], o['stride'], o['pad']
                y = F.max_pool2d(x, k, s, pd, ceil_mode=False)
                if tuple(y.shape[2:]) != (vo.h, vo.w):
                    y = F.max_pool2d(x, k, s, pd, ceil_mode=True)
                assert tuple(y.shape[2:]) == (vo.h, vo.w)
                wr(o['out'], y)
            elif kind == P.OP_BILINEAR:
                slope = wt(nat['aux_off'], (vo.c,)) if nat and nat.get('aux_off', -1) >= 0 else None
                wr(o['out'], _act(F.interpolate(x, size=(vo.h, vo.w), mode='bilinear', align_corners=bool(o['flags'] & 1)), o['act'], slope))
            elif kind == P.OP_NEAREST:
                wr(o['out'], F.interpolate(x, size=(vo.h, vo.w), mode='nearest'))
            elif kind == P.OP_ADD:
                wr(o['out'], _act(x[:, :, :vo.h, :vo.w] + rd(o['in1']), o['act']))
            elif kind == P.OP_GAVGPOOL:
                wr(o['out'], x.mean((2, 3), keepdim=True))
            elif kind == P.OP_SCALE:
                wr(o['out'], x * rd(o['in1']))
            elif kind in (P.OP_NCHW_TO_NHWC, P.OP_NHWC_TO_NCHW, P.OP_COPY):
                c = min(x.shape[1], vo.c)
                y = torch.zeros((vo.n, vo.c, vo.h, vo.w), dtype=torch.float32)
                y[:, :c] = x[:, :c]
                wr(o['out'], y)
            elif kind == P.OP_ACT:
                slope = wt(nat['aux_off'], (vo.c,)) if nat.get('aux_off', -1) >= 0 else None
                wr(o['out'], _act(x, o['act'], slope))
            elif kind == P.OP_LAYERNORM:
                c = vo.c
                g, b = wt(nat['w_off'], (c,)), wt(nat['b_off'], (c,))
                eps = float(w_nat[nat['aux_off']])
                wr(o['out'], F.layer_norm(x.permute(0, 2, 3, 1), (c,), g, b, eps).permute(0, 3, 1, 2))
            elif kind == P.OP_ATTENTION:
                heads, d = o['groups'], o['cin_g']
                n_, N = x.shape[0], x.shape[2]
                qkv = x[:, :, :, 0].permute(0, 2, 1).reshape(n_, N, 3, heads, d).permute(2, 0, 3, 1, 4)
                attn = qkv[0] @ qkv[1].transpose(-2, -1)                       # q arrives pre-scaled
                if nat.get('aux_off', -1) >= 0:
                    gh, gw = o['kh'], o['kw']
                    table = wt(nat['aux_off'], ((2 * gh - 1) * (2 * gw - 1) + 3, heads))
                    attn = attn + table[_rel_index(gh, gw).view(-1)].view(N, N, heads).permute(2, 0, 1).unsqueeze(0)
                y = (attn.softmax(dim=-1) @ qkv[2]).transpose(1, 2).reshape(n_, N, heads * d)
                wr(o['out'], y.permute(0, 2, 1).unsqueeze(-1))
            elif kind == P.OP_TOKENS:
                mode = o['flags']
                if mode == 0:
                    c = vo.c
                    cls = wt(nat['aux_off'], (c,)).view(1, c, 1, 1).expand(x.shape[0], c, 1, 1)
                    wr(o['out'], torch.cat([cls, x.flatten(2).unsqueeze(-1)], 2))
                else:
                    tok = x[:, :, 1:, 0]                                       # [n, c, np]
                    if mode == 1:
                        tok = torch.cat([tok, x[:, :, :1, 0].expand_as(tok)], 1)
                    wr(o['out'], tok.reshape(tok.shape[0], tok.shape[1], vo.h, vo.w))
            elif kind == P.OP_DEPTH_TO_SPACE:
                k, c = o['stride'], vo.c
                n_, _, h_, w_ = x.shape
                wr(o['out'], x.reshape(n_, k, k, c, h_, w_).permute(0, 3, 4, 1, 5, 2).reshape(n_, c, h_ * k, w_ * k))
            else:
                raise NotImplementedError("op kind %d has no torch-CPU restatement here" % kind)
            for b, l in list(last_use.items()):            # release dead intermediates (LeReS @640 would otherwise hold ~6 GB)
                if l == i and b in store and b.ext < 0 and b not in keep and not b.keep:
                    del store[b]
    out = {}
    for t in want_views:
        out[t] = rd(t.id).permute(0, 2, 3, 1).contiguous().numpy()
    return out
