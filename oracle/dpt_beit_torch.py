"""MiDaS DPT-BEiT as plain torch.nn modules (TEST INFRASTRUCTURE ONLY -- never imported by the product).

An INDEPENDENT statement of the network `torch.hub.load("intel-isl/MiDaS", "DPT_BEiT_L_384")` builds -- the core of ZoeDepth, which the
reference does not vendor (depth_modules/zoedepth/models/base_models/midas.py:341) -- written from the published definitions
(timm 0.6.x models/beit.py; MiDaS 3.1 midas/backbones/beit.py, backbones/utils.py, blocks.py, dpt_depth.py) with THEIR attribute names,
so that `state_dict()` has the names and shapes of the published checkpoint dpt_beit_large_384.pt.  It shares nothing with the lowering
(cartoonsegmentation_amd/nets/dpt_beit.py): real nn.Linear / nn.ConvTranspose2d / F.interpolate / softmax, the relative-position bias
gathered through an index tensor every forward, layer scales applied to activations.  Three uses:

  * tests/test_oracle_dpt_beit.py: the lowered program (C oracle) against these modules fed with the same state_dict -- wiring check;
  * the same modules against HuggingFace transformers' DPTForDepthEstimation + BeitBackbone (an unrelated third implementation that is
    installed in this image), weights mapped name by name -- pins THIS restatement against an external one;
  * golden-free: no fixture travels, both checks run wherever torch + transformers are installed.

Parity status: against MiDaS' own code [EXT, unpinned] (not under /root/reference, torch.hub has no network here).
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F


def gen_relative_position_index(window_size):
    """timm models/beit.py (0.6.x) gen_relative_position_index; MiDaS calls it per input window"""
    num_relative_distance = (2 * window_size[0] - 1) * (2 * window_size[1] - 1) + 3
    window_area = window_size[0] * window_size[1]
    coords = torch.stack(torch.meshgrid([torch.arange(window_size[0]), torch.arange(window_size[1])], indexing='ij'))
    coords_flatten = torch.flatten(coords, 1)
    relative_coords = coords_flatten[:, :, None] - coords_flatten[:, None, :]
    relative_coords = relative_coords.permute(1, 2, 0).contiguous()
    relative_coords[:, :, 0] += window_size[0] - 1
    relative_coords[:, :, 1] += window_size[1] - 1
    relative_coords[:, :, 0] *= 2 * window_size[1] - 1
    relative_position_index = torch.zeros(size=(window_area + 1,) * 2, dtype=relative_coords.dtype)
    relative_position_index[1:, 1:] = relative_coords.sum(-1)
    relative_position_index[0, 0:] = num_relative_distance - 3
    relative_position_index[0:, 0] = num_relative_distance - 2
    relative_position_index[0, 0] = num_relative_distance - 1
    return relative_position_index


class Attention(nn.Module):
    """timm beit.py Attention with MiDaS beit.py attention_forward / _get_rel_pos_bias"""

    def __init__(self, dim, num_heads, window_size):
        super().__init__()
        self.num_heads = num_heads
        head_dim = dim // num_heads
        self.scale = head_dim ** -0.5
        self.qkv = nn.Linear(dim, dim * 3, bias=False)
        self.q_bias = nn.Parameter(torch.zeros(dim))
        self.v_bias = nn.Parameter(torch.zeros(dim))
        self.window_size = window_size
        self.num_relative_distance = (2 * window_size[0] - 1) * (2 * window_size[1] - 1) + 3
        self.relative_position_bias_table = nn.Parameter(torch.zeros(self.num_relative_distance, num_heads))
        self.proj = nn.Linear(dim, dim)

    def _get_rel_pos_bias(self, window_size):
        old_height = 2 * self.window_size[0] - 1
        old_width = 2 * self.window_size[1] - 1
        new_height = 2 * window_size[0] - 1
        new_width = 2 * window_size[1] - 1
        table = self.relative_position_bias_table
        old_sub_table = table[:self.num_relative_distance - 3]
        old_sub_table = old_sub_table.reshape(1, old_width, old_height, -1).permute(0, 3, 1, 2)
        new_sub_table = F.interpolate(old_sub_table, size=(new_height, new_width), mode="bilinear")
        new_sub_table = new_sub_table.permute(0, 2, 3, 1).reshape(new_height * new_width, -1)
        new_table = torch.cat([new_sub_table, table[self.num_relative_distance - 3:]])
        index = gen_relative_position_index(window_size)
        n = window_size[0] * window_size[1] + 1
        bias = new_table[index.view(-1)].view(n, n, -1)
        return bias.permute(2, 0, 1).contiguous().unsqueeze(0)

    def forward(self, x, resolution):
        B, N, C = x.shape
        qkv_bias = torch.cat((self.q_bias, torch.zeros_like(self.v_bias), self.v_bias))
        qkv = F.linear(x, self.qkv.weight, qkv_bias)
        qkv = qkv.reshape(B, N, 3, self.num_heads, -1).permute(2, 0, 3, 1, 4)
        q, k, v = qkv.unbind(0)
        q = q * self.scale
        attn = q @ k.transpose(-2, -1)
        window_size = (resolution[0] // 16, resolution[1] // 16)
        attn = attn + self._get_rel_pos_bias(window_size)
        attn = attn.softmax(dim=-1)
        x = (attn @ v).transpose(1, 2).reshape(B, N, -1)
        return self.proj(x)


class Mlp(nn.Module):
    def __init__(self, dim, hidden):
        super().__init__()
        self.fc1 = nn.Linear(dim, hidden)
        self.act = nn.GELU()
        self.fc2 = nn.Linear(hidden, dim)

    def forward(self, x):
        return self.fc2(self.act(self.fc1(x)))


class Block(nn.Module):
    def __init__(self, dim, num_heads, mlp_ratio, window_size, eps):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim, eps=eps)
        self.attn = Attention(dim, num_heads, window_size)
        self.norm2 = nn.LayerNorm(dim, eps=eps)
        self.mlp = Mlp(dim, int(dim * mlp_ratio))
        self.gamma_1 = nn.Parameter(torch.ones(dim))
        self.gamma_2 = nn.Parameter(torch.ones(dim))

    def forward(self, x, resolution):
        x = x + self.gamma_1 * self.attn(self.norm1(x), resolution)
        x = x + self.gamma_2 * self.mlp(self.norm2(x))
        return x


class PatchEmbed(nn.Module):
    def __init__(self, patch, dim):
        super().__init__()
        self.proj = nn.Conv2d(3, dim, kernel_size=patch, stride=patch)

    def forward(self, x):
        return self.proj(x).flatten(2).transpose(1, 2)


class Beit(nn.Module):
    """timm Beit without absolute position embedding, head and final norm use (MiDaS hooks the blocks)"""

    def __init__(self, embed, depth, heads, mlp_ratio, patch, base_grid, eps):
        super().__init__()
        self.patch_embed = PatchEmbed(patch, embed)
        self.cls_token = nn.Parameter(torch.zeros(1, 1, embed))
        self.blocks = nn.ModuleList([Block(embed, heads, mlp_ratio, base_grid, eps) for _ in range(depth)])

    def forward_features(self, x, hooks):
        resolution = x.shape[2:]
        x = self.patch_embed(x)
        x = torch.cat((self.cls_token.expand(x.shape[0], -1, -1), x), dim=1)
        taken = []
        for i, blk in enumerate(self.blocks):
            x = blk(x, resolution)
            if i in hooks:
                taken.append(x)
        return taken


class ProjectReadout(nn.Module):
    def __init__(self, in_features, start_index=1):
        super().__init__()
        self.start_index = start_index
        self.project = nn.Sequential(nn.Linear(2 * in_features, in_features), nn.GELU())

    def forward(self, x):
        readout = x[:, 0].unsqueeze(1).expand_as(x[:, self.start_index:])
        return self.project(torch.cat((x[:, self.start_index:], readout), -1))


class Slice(nn.Module):
    def forward(self, x):
        return x[:, 1:]


class _Stub(nn.Module):
    """the parameter-free positions 1, 2 (Transpose, Unflatten) of MiDaS's act_postprocess Sequentials, kept so that the numbered
    children 0 / 3 / 4 carry the checkpoint's names"""
    def forward(self, x):
        return x


class ResidualConvUnit_custom(nn.Module):
    def __init__(self, features):
        super().__init__()
        self.conv1 = nn.Conv2d(features, features, 3, 1, 1, bias=True)
        self.conv2 = nn.Conv2d(features, features, 3, 1, 1, bias=True)

    def forward(self, x):
        out = self.conv1(F.relu(x))
        out = self.conv2(F.relu(out))
        return out + x


class FeatureFusionBlock_custom(nn.Module):
    def __init__(self, features):
        super().__init__()
        self.out_conv = nn.Conv2d(features, features, 1, 1, 0, bias=True)
        self.resConfUnit1 = ResidualConvUnit_custom(features)
        self.resConfUnit2 = ResidualConvUnit_custom(features)

    def forward(self, *xs, size=None):
        output = xs[0]
        if len(xs) == 2:
            output = output + self.resConfUnit1(xs[1])
        output = self.resConfUnit2(output)
        modifier = {"scale_factor": 2} if size is None else {"size": size}
        output = F.interpolate(output, **modifier, mode="bilinear", align_corners=True)
        return self.out_conv(output)


class Interpolate(nn.Module):
    def forward(self, x):
        return F.interpolate(x, scale_factor=2, mode="bilinear", align_corners=True)


class DPTBeit(nn.Module):
    """DPTDepthModel(backbone="beitl16_384", non_negative=True): .pretrained.model (Beit), .pretrained.act_postprocess1..4, .scratch.*"""

    def __init__(self, embed=1024, depth=24, heads=16, mlp_ratio=4, patch=16, base_grid=(24, 24), hooks=(5, 11, 17, 23), features=256,
                 neck=(256, 512, 1024, 1024), readout='project', ln_eps=1e-6, head_features_2=32):
        super().__init__()
        self.hooks, self.patch = tuple(hooks), patch
        self.pretrained = nn.Module()
        self.pretrained.model = Beit(embed, depth, heads, mlp_ratio, patch, tuple(base_grid), ln_eps)

        def readout_op():
            return ProjectReadout(embed) if readout == 'project' else Slice()
        self.pretrained.act_postprocess1 = nn.Sequential(readout_op(), _Stub(), _Stub(), nn.Conv2d(embed, neck[0], 1),
                                                         nn.ConvTranspose2d(neck[0], neck[0], kernel_size=4, stride=4, padding=0))
        self.pretrained.act_postprocess2 = nn.Sequential(readout_op(), _Stub(), _Stub(), nn.Conv2d(embed, neck[1], 1),
                                                         nn.ConvTranspose2d(neck[1], neck[1], kernel_size=2, stride=2, padding=0))
        self.pretrained.act_postprocess3 = nn.Sequential(readout_op(), _Stub(), _Stub(), nn.Conv2d(embed, neck[2], 1))
        self.pretrained.act_postprocess4 = nn.Sequential(readout_op(), _Stub(), _Stub(), nn.Conv2d(embed, neck[3], 1),
                                                         nn.Conv2d(neck[3], neck[3], kernel_size=3, stride=2, padding=1))
        self.scratch = nn.Module()
        for k in range(4):
            setattr(self.scratch, 'layer%d_rn' % (k + 1), nn.Conv2d(neck[k], features, 3, 1, 1, bias=False))
            setattr(self.scratch, 'refinenet%d' % (k + 1), FeatureFusionBlock_custom(features))
        self.scratch.output_conv = nn.Sequential(nn.Conv2d(features, features // 2, 3, 1, 1), Interpolate(),
                                                 nn.Conv2d(features // 2, head_features_2, 3, 1, 1), nn.ReLU(True),
                                                 nn.Conv2d(head_features_2, 1, 1, 1, 0), nn.ReLU(True), nn.Identity())

    def forward(self, x):
        """-> (relative depth [B,H,W], {'out_conv','l4_rn','r4','r3','r2','r1'}: what ZoeDepth's MidasCore hooks collect)"""
        b, _, h, w = x.shape
        gh, gw = h // self.patch, w // self.patch
        layers = []
        for k, tok in enumerate(self.pretrained.model.forward_features(x, self.hooks)):
            pp = getattr(self.pretrained, 'act_postprocess%d' % (k + 1))
            y = pp[0](tok).transpose(1, 2).unflatten(2, (gh, gw))
            for m in list(pp)[3:]:
                y = m(y)
            layers.append(y)
        s = self.scratch
        l1, l2, l3, l4 = (getattr(s, 'layer%d_rn' % (k + 1))(layers[k]) for k in range(4))
        p4 = s.refinenet4(l4, size=l3.shape[2:])
        p3 = s.refinenet3(p4, l3, size=l2.shape[2:])
        p2 = s.refinenet2(p3, l2, size=l1.shape[2:])
        p1 = s.refinenet1(p2, l1)
        oc = s.output_conv[3](s.output_conv[2](s.output_conv[1](s.output_conv[0](p1))))
        rel = s.output_conv[5](s.output_conv[4](oc))
        return rel.squeeze(1), {'out_conv': oc, 'l4_rn': l4, 'r4': p4, 'r3': p3, 'r2': p2, 'r1': p1}


def fill_deterministic(module, seed=0):
    """deterministic O(1)-scale parameters (no checkpoint exists offline): uniform fan-in scaling for weights, small biases, layer scales
    and LayerNorm gammas away from their trivial initial values, a relative-position table with logits of order one"""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for name, p in module.named_parameters():
            u = torch.rand(p.shape, generator=g) * 2 - 1
            if name.endswith('relative_position_bias_table'):
                p.copy_(1.5 * u)
            elif 'gamma_' in name:
                p.copy_(0.3 + 0.2 * u)
            elif name.endswith('cls_token'):
                p.copy_(0.5 * u)
            elif 'norm' in name and name.endswith('weight'):
                p.copy_(1.0 + 0.1 * u)
            elif p.dim() >= 2:
                fan_in = p[0].numel() if not isinstance(module.get_submodule(name.rsplit('.', 1)[0]), nn.ConvTranspose2d) else p.shape[0] * p[0, 0].numel()
                p.copy_(u * math.sqrt(3.0 / fan_in) * (1.4 if p.dim() == 4 else 1.0))
            else:
                p.copy_(0.05 * u)
    return module
