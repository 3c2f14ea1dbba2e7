"""Independent torch.nn restatement of RTMDet-Ins (TEST INFRASTRUCTURE ONLY -- never imported by the product).

Why it exists: the reference builds its detector through the mmdet registry (animeinsseg/__init__.py:196-215, :450) and mmdet 3.3.0 /
mmcv 2.1.0 are not under /root/reference, so the product's graph builder (cartoonsegmentation_amd/nets/rtmdet.py -> program.py) and the
oracle interpreter that executes the SAME lowered program could share a wiring error (concat order, slice offset, which tower feeds
rtm_reg) without any "HIP == oracle" assertion noticing.  This file shares NOTHING with that lowering: it is written as plain
nn.Modules with mmdet's attribute names -- mmdet/models/backbones/cspnext.py (CSPNeXt), backbones/csp_darknet.py (SPPBottleneck),
layers/csp_layer.py (CSPLayer / CSPNeXtBlock / ChannelAttention), necks/cspnext_pafpn.py (CSPNeXtPAFPN),
dense_heads/rtmdet_ins_head.py (RTMDetInsSepBNHead / MaskFeatModule / predict_by_feat / _bbox_mask_post_process), mmcv ConvModule /
DepthwiseSeparableConvModule -- restated from their published definitions (PARITY UNPINNED against mmdet itself; what it pins is the
product's wiring against a second, structurally different statement of the same model).  Its state_dict() has mmdet's parameter names,
so the product consumes exactly what a real `rtmdetl_e60.ckpt` would hand it.

Also used by bench.py's cpu_baseline leg as the torch-CPU (oneDNN) execution of the detector.
"""
import math

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F


class ConvModule(nn.Module):
    """mmcv.cnn.ConvModule with norm_cfg=BN, act_cfg=SiLU: conv(bias=False) -> bn -> act"""

    def __init__(self, cin, cout, k, stride=1, padding=0, groups=1, eps=1e-5, act=True):
        super().__init__()
        self.conv = nn.Conv2d(cin, cout, k, stride=stride, padding=padding, groups=groups, bias=False)
        self.bn = nn.BatchNorm2d(cout, eps=eps)
        self.activate = nn.SiLU() if act else nn.Identity()

    def forward(self, x):
        return self.activate(self.bn(self.conv(x)))


class DepthwiseSeparableConvModule(nn.Module):
    """mmcv: depthwise ConvModule (groups = channels) + pointwise ConvModule, both with norm + act"""

    def __init__(self, cin, cout, k, padding, eps):
        super().__init__()
        self.depthwise_conv = ConvModule(cin, cin, k, padding=padding, groups=cin, eps=eps)
        self.pointwise_conv = ConvModule(cin, cout, 1, eps=eps)

    def forward(self, x):
        return self.pointwise_conv(self.depthwise_conv(x))


class CSPNeXtBlock(nn.Module):
    def __init__(self, cin, cout, expansion, add_identity, eps, kernel_size=5):
        super().__init__()
        hidden = int(cout * expansion)
        self.conv1 = ConvModule(cin, hidden, 3, padding=1, eps=eps)
        self.conv2 = DepthwiseSeparableConvModule(hidden, cout, kernel_size, kernel_size // 2, eps)
        self.add_identity = add_identity and cin == cout

    def forward(self, x):
        out = self.conv2(self.conv1(x))
        return out + x if self.add_identity else out


class ChannelAttention(nn.Module):
    def __init__(self, channels):
        super().__init__()
        self.global_avgpool = nn.AdaptiveAvgPool2d(1)
        self.fc = nn.Conv2d(channels, channels, 1, 1, 0, bias=True)
        self.act = nn.Hardsigmoid()

    def forward(self, x):
        return x * self.act(self.fc(self.global_avgpool(x)))


class CSPLayer(nn.Module):
    def __init__(self, cin, cout, expand_ratio, num_blocks, add_identity, channel_attention, eps):
        super().__init__()
        mid = int(cout * expand_ratio)
        self.main_conv = ConvModule(cin, mid, 1, eps=eps)
        self.short_conv = ConvModule(cin, mid, 1, eps=eps)
        self.final_conv = ConvModule(2 * mid, cout, 1, eps=eps)
        self.blocks = nn.Sequential(*[CSPNeXtBlock(mid, mid, 1.0, add_identity, eps) for _ in range(num_blocks)])
        self.channel_attention = channel_attention
        if channel_attention:
            self.attention = ChannelAttention(2 * mid)

    def forward(self, x):
        x_short = self.short_conv(x)
        x_main = self.blocks(self.main_conv(x))
        x_final = torch.cat((x_main, x_short), dim=1)
        if self.channel_attention:
            x_final = self.attention(x_final)
        return self.final_conv(x_final)


class SPPBottleneck(nn.Module):
    def __init__(self, cin, cout, eps, kernel_sizes=(5, 9, 13)):
        super().__init__()
        mid = cin // 2
        self.conv1 = ConvModule(cin, mid, 1, eps=eps)
        self.poolings = nn.ModuleList([nn.MaxPool2d(kernel_size=ks, stride=1, padding=ks // 2) for ks in kernel_sizes])
        self.conv2 = ConvModule(mid * (len(kernel_sizes) + 1), cout, 1, eps=eps)

    def forward(self, x):
        x = self.conv1(x)
        return self.conv2(torch.cat([x] + [p(x) for p in self.poolings], dim=1))


class CSPNeXt(nn.Module):
    ARCH_P5 = [[64, 128, 3, True, False], [128, 256, 6, True, False], [256, 512, 6, True, False], [512, 1024, 3, False, True]]

    def __init__(self, deepen_factor=1.0, widen_factor=1.0, expand_ratio=0.5, eps=1e-5, out_indices=(2, 3, 4)):
        super().__init__()
        a = self.ARCH_P5
        c0 = int(a[0][0] * widen_factor // 2)
        self.stem = nn.Sequential(ConvModule(3, c0, 3, stride=2, padding=1, eps=eps), ConvModule(c0, c0, 3, padding=1, eps=eps),
                                  ConvModule(c0, int(a[0][0] * widen_factor), 3, padding=1, eps=eps))
        self.out_indices = out_indices
        for i, (cin, cout, nb, add_identity, use_spp) in enumerate(a):
            cin, cout = int(cin * widen_factor), int(cout * widen_factor)
            nb = max(round(nb * deepen_factor), 1)
            stage = [ConvModule(cin, cout, 3, stride=2, padding=1, eps=eps)]
            if use_spp:
                stage.append(SPPBottleneck(cout, cout, eps))
            stage.append(CSPLayer(cout, cout, expand_ratio, nb, add_identity, True, eps))
            self.add_module('stage%d' % (i + 1), nn.Sequential(*stage))

    def forward(self, x):
        outs = []
        for i, name in enumerate(['stem', 'stage1', 'stage2', 'stage3', 'stage4']):
            x = getattr(self, name)(x)
            if i in self.out_indices:
                outs.append(x)
        return tuple(outs)


class CSPNeXtPAFPN(nn.Module):
    def __init__(self, in_channels, out_channels, num_csp_blocks, expand_ratio, eps):
        super().__init__()
        self.in_channels = in_channels
        self.upsample = nn.Upsample(scale_factor=2, mode='nearest')
        self.reduce_layers, self.top_down_blocks = nn.ModuleList(), nn.ModuleList()
        for idx in range(len(in_channels) - 1, 0, -1):
            self.reduce_layers.append(ConvModule(in_channels[idx], in_channels[idx - 1], 1, eps=eps))
            self.top_down_blocks.append(CSPLayer(in_channels[idx - 1] * 2, in_channels[idx - 1], expand_ratio, num_csp_blocks, False, False, eps))
        self.downsamples, self.bottom_up_blocks = nn.ModuleList(), nn.ModuleList()
        for idx in range(len(in_channels) - 1):
            self.downsamples.append(ConvModule(in_channels[idx], in_channels[idx], 3, stride=2, padding=1, eps=eps))
            self.bottom_up_blocks.append(CSPLayer(in_channels[idx] * 2, in_channels[idx + 1], expand_ratio, num_csp_blocks, False, False, eps))
        self.out_convs = nn.ModuleList([ConvModule(c, out_channels, 3, padding=1, eps=eps) for c in in_channels])

    def forward(self, inputs):
        n = len(self.in_channels)
        inner_outs = [inputs[-1]]
        for idx in range(n - 1, 0, -1):
            feat_high = self.reduce_layers[n - 1 - idx](inner_outs[0])
            inner_outs[0] = feat_high
            inner = self.top_down_blocks[n - 1 - idx](torch.cat([self.upsample(feat_high), inputs[idx - 1]], 1))
            inner_outs.insert(0, inner)
        outs = [inner_outs[0]]
        for idx in range(n - 1):
            down = self.downsamples[idx](outs[-1])
            outs.append(self.bottom_up_blocks[idx](torch.cat([down, inner_outs[idx + 1]], 1)))
        return tuple(conv(o) for conv, o in zip(self.out_convs, outs))


class MaskFeatModule(nn.Module):
    def __init__(self, in_channels, feat_channels, stacked_convs, num_levels, num_prototypes, eps):
        super().__init__()
        self.num_levels = num_levels
        self.fusion_conv = nn.Conv2d(num_levels * in_channels, in_channels, 1)
        self.stacked_convs = nn.Sequential(*[ConvModule(in_channels if i == 0 else feat_channels, feat_channels, 3, padding=1, eps=eps)
                                             for i in range(stacked_convs)])
        self.projection = nn.Conv2d(feat_channels, num_prototypes, kernel_size=1)

    def forward(self, features):
        size = features[0].shape[-2:]
        fusion = [features[0]] + [F.interpolate(features[i], size=size, mode='bilinear') for i in range(1, self.num_levels)]
        return self.projection(self.stacked_convs(self.fusion_conv(torch.cat(fusion, dim=1))))


class RTMDetInsSepBNHead(nn.Module):
    def __init__(self, num_classes=1, in_channels=256, feat_channels=256, stacked_convs=2, share_conv=True, num_prototypes=8,
                 dyconv_channels=8, num_dyconvs=3, strides=(8, 16, 32), eps=1e-5):
        super().__init__()
        self.num_classes, self.strides, self.share_conv = num_classes, strides, share_conv
        self.num_prototypes, self.dyconv_channels, self.num_dyconvs = num_prototypes, dyconv_channels, num_dyconvs
        self.cls_convs, self.reg_convs, self.kernel_convs = nn.ModuleList(), nn.ModuleList(), nn.ModuleList()
        self.rtm_cls, self.rtm_reg, self.rtm_kernel = nn.ModuleList(), nn.ModuleList(), nn.ModuleList()
        weight_nums, bias_nums = [], []
        for i in range(num_dyconvs):
            if i == 0:
                weight_nums.append((num_prototypes + 2) * dyconv_channels); bias_nums.append(dyconv_channels)
            elif i == num_dyconvs - 1:
                weight_nums.append(dyconv_channels); bias_nums.append(1)
            else:
                weight_nums.append(dyconv_channels * dyconv_channels); bias_nums.append(dyconv_channels)
        self.weight_nums, self.bias_nums = weight_nums, bias_nums
        self.num_gen_params = sum(weight_nums) + sum(bias_nums)
        for _ in strides:
            cls_convs, reg_convs, kernel_convs = nn.ModuleList(), nn.ModuleList(), nn.ModuleList()
            for i in range(stacked_convs):
                chn = in_channels if i == 0 else feat_channels
                cls_convs.append(ConvModule(chn, feat_channels, 3, padding=1, eps=eps))
                reg_convs.append(ConvModule(chn, feat_channels, 3, padding=1, eps=eps))        # built, then dropped (as in mmdet)
                kernel_convs.append(ConvModule(chn, feat_channels, 3, padding=1, eps=eps))
            self.cls_convs.append(cls_convs)
            self.reg_convs.append(cls_convs)                     # mmdet 3.x registers the cls tower as the reg tower
            self.kernel_convs.append(kernel_convs)
            self.rtm_cls.append(nn.Conv2d(feat_channels, num_classes, 1))
            self.rtm_reg.append(nn.Conv2d(feat_channels, 4, 1))
            self.rtm_kernel.append(nn.Conv2d(feat_channels, self.num_gen_params, 1))
        if share_conv:
            for n in range(len(strides)):
                for i in range(stacked_convs):
                    self.cls_convs[n][i].conv = self.cls_convs[0][i].conv
                    self.reg_convs[n][i].conv = self.reg_convs[0][i].conv
        self.mask_head = MaskFeatModule(in_channels, feat_channels, 4, len(strides), num_prototypes, eps)

    def forward(self, feats):
        mask_feat = self.mask_head(feats)
        cls_scores, bbox_preds, kernel_preds = [], [], []
        for idx, (x, stride) in enumerate(zip(feats, self.strides)):
            cls_feat = reg_feat = kernel_feat = x
            for layer in self.cls_convs[idx]:
                cls_feat = layer(cls_feat)
            cls_score = self.rtm_cls[idx](cls_feat)
            for layer in self.kernel_convs[idx]:
                kernel_feat = layer(kernel_feat)
            kernel_pred = self.rtm_kernel[idx](kernel_feat)
            for layer in self.reg_convs[idx]:
                reg_feat = layer(reg_feat)
            reg_dist = F.relu(self.rtm_reg[idx](reg_feat)) * stride
            cls_scores.append(cls_score); bbox_preds.append(reg_dist); kernel_preds.append(kernel_pred)
        return cls_scores, bbox_preds, kernel_preds, mask_feat

    # ---- rtmdet_ins_head.py::_mask_predict_by_feat_single / parse_dynamic_params (restated; the reference's own copy of this
    # function is pinned separately by tests/golden/pin_maskhead_*.npz) ----
    def mask_logits(self, mask_feat, kernels, priors):
        num_inst = priors.shape[0]
        h, w = mask_feat.shape[-2:]
        if num_inst < 1:
            return torch.empty((0, h, w), dtype=mask_feat.dtype)
        stride = self.strides[0]
        ys, xs = torch.meshgrid(torch.arange(h, dtype=torch.float32) * stride, torch.arange(w, dtype=torch.float32) * stride, indexing='ij')
        coord = torch.stack([xs.reshape(-1), ys.reshape(-1)], 1)                                # MlvlPointGenerator(offset=0)
        rel = (priors[:, None, :2] - coord[None]).permute(0, 2, 1) / (priors[:, 2, None, None] * 8)
        rel = rel.reshape(num_inst, 2, h, w)
        x = torch.cat([rel, mask_feat.repeat(num_inst, 1, 1, 1)], dim=1)
        params = list(torch.split(kernels, self.weight_nums + self.bias_nums, dim=1))
        ws, bs = params[:self.num_dyconvs], params[self.num_dyconvs:]
        x = x.reshape(1, -1, h, w)
        for i, (wt, b) in enumerate(zip(ws, bs)):
            last = i == self.num_dyconvs - 1
            wt = wt.reshape(num_inst * (1 if last else self.dyconv_channels), -1, 1, 1)
            b = b.reshape(num_inst * (1 if last else self.dyconv_channels))
            x = F.conv2d(x, wt, bias=b, stride=1, padding=0, groups=num_inst)
            if not last:
                x = F.relu(x)
        return x.reshape(num_inst, h, w)


class RTMDetIns(nn.Module):
    """detector = backbone + neck + bbox_head, with the state_dict names of mmdet's RTMDet"""

    def __init__(self, deepen_factor=1.0, widen_factor=1.0, expand_ratio=0.5, num_classes=1, feat_channels=256, stacked_convs=2,
                 share_conv=True, eps_backbone=1e-5, eps_neck=1e-5, eps_head=1e-5):
        super().__init__()
        chans = [int(c * widen_factor) for c in (256, 512, 1024)]
        self.backbone = CSPNeXt(deepen_factor, widen_factor, expand_ratio, eps_backbone)
        self.neck = CSPNeXtPAFPN(chans, feat_channels, max(round(3 * deepen_factor), 1), expand_ratio, eps_neck)
        self.bbox_head = RTMDetInsSepBNHead(num_classes, feat_channels, feat_channels, stacked_convs, share_conv, eps=eps_head)

    def forward(self, x):
        return self.bbox_head(self.neck(self.backbone(x)))


def fill_closed_form(model, prefix='rtmdet.'):
    """deterministic closed-form parameters (cartoonsegmentation_amd.weights.synth_tensor by parameter kind); shared modules are filled once"""
    from cartoonsegmentation_amd.weights import synth_tensor
    seen = set()
    with torch.no_grad():
        for name, t in list(model.named_parameters(remove_duplicate=False)) + list(model.named_buffers(remove_duplicate=False)):
            if id(t) in seen or name.endswith('num_batches_tracked'):
                continue
            seen.add(id(t))
            leaf = name.rsplit('.', 1)[1]
            is_bn = '.bn.' in name
            kind = {'weight': 'bn_gamma', 'bias': 'bn_beta', 'running_mean': 'bn_mean', 'running_var': 'bn_var'}[leaf] if is_bn else \
                ('conv_w' if leaf == 'weight' else 'conv_b')
            t.copy_(torch.from_numpy(synth_tensor(prefix + name, tuple(t.shape), kind)))
    return model.eval()


def preprocess(img_bgr_u8_resized_padded, mean, std):
    """DetDataPreprocessor on an already resized + padded uint8 BGR image [S,S,3] -> [1,3,S,S]"""
    x = torch.from_numpy(np.ascontiguousarray(img_bgr_u8_resized_padded)).permute(2, 0, 1).float()
    m, s = torch.tensor(mean).view(3, 1, 1), torch.tensor(std).view(3, 1, 1)
    return ((x - m) / s)[None]


def _nms(boxes, thr):
    """mmcv.ops.nms, CUDA criterion: j is suppressed by a kept i when inter > thr * (Sa + Sb - inter); boxes sorted by score"""
    n = boxes.shape[0]
    b = boxes.numpy()
    area = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    sup = np.zeros(n, bool)
    keep = []
    for i in range(n):
        if sup[i]:
            continue
        keep.append(i)
        lt = np.maximum(b[i, :2], b[i + 1:, :2]); rb = np.minimum(b[i, 2:], b[i + 1:, 2:])
        wh = np.maximum(rb - lt, np.float32(0))
        inter = wh[:, 0] * wh[:, 1]
        sup[i + 1:] |= inter > np.float32(thr) * (area[i] + area[i + 1:] - inter)
    return torch.tensor(keep, dtype=torch.long)


def predict(model, x, img_shape, ori_shape, scale_factor, score_thr=0.05, nms_pre=1000, nms_iou=0.6, max_per_img=100, min_bbox_size=0,
            mask_thr_binary=0.5):
    """RTMDetInsHead.predict_by_feat + _bbox_mask_post_process(rescale=True, with_nms=True) for ONE image.
    img_shape = (rh, rw) of the resized image inside the padded input, scale_factor = (w_scale, h_scale)"""
    with torch.no_grad():
        raw = model(x)
    return decode(model.bbox_head, raw, img_shape, ori_shape, scale_factor, score_thr, nms_pre, nms_iou, max_per_img, min_bbox_size,
                  mask_thr_binary)


def decode(head, raw, img_shape, ori_shape, scale_factor, score_thr=0.05, nms_pre=1000, nms_iou=0.6, max_per_img=100, min_bbox_size=0,
           mask_thr_binary=0.5):
    """the post-processing half of predict(): raw = (cls_scores, bbox_preds (already x stride), kernel_preds, mask_feat), NCHW tensors"""
    cls_scores, bbox_preds, kernel_preds, mask_feat = raw
    mlvl = dict(scores=[], labels=[], bbox=[], priors=[], kernels=[])
    for cls_score, bbox_pred, kernel_pred, stride in zip(cls_scores, bbox_preds, kernel_preds, head.strides):
        h, w = cls_score.shape[-2:]
        ys, xs = torch.meshgrid(torch.arange(h, dtype=torch.float32) * stride, torch.arange(w, dtype=torch.float32) * stride, indexing='ij')
        priors = torch.stack([xs.reshape(-1), ys.reshape(-1), torch.full((h * w,), float(stride)), torch.full((h * w,), float(stride))], 1)
        bbox_pred = bbox_pred[0].permute(1, 2, 0).reshape(-1, 4)
        scores = cls_score[0].permute(1, 2, 0).reshape(-1, head.num_classes).sigmoid()
        kernel_pred = kernel_pred[0].permute(1, 2, 0).reshape(-1, head.num_gen_params)
        # filter_scores_and_topk
        valid = scores > score_thr
        sc = scores[valid]
        idxs = valid.nonzero()
        k = min(nms_pre, idxs.shape[0])
        sc, order = sc.sort(descending=True, stable=True)
        sc, order = sc[:k], order[:k]
        keep, labels = idxs[order, 0], idxs[order, 1]
        mlvl['scores'].append(sc); mlvl['labels'].append(labels); mlvl['bbox'].append(bbox_pred[keep])
        mlvl['priors'].append(priors[keep]); mlvl['kernels'].append(kernel_pred[keep])
    scores, labels = torch.cat(mlvl['scores']), torch.cat(mlvl['labels'])
    dist, priors, kernels = torch.cat(mlvl['bbox']), torch.cat(mlvl['priors']), torch.cat(mlvl['kernels'])
    # DistancePointBBoxCoder.decode(max_shape=img_shape)
    x1 = (priors[:, 0] - dist[:, 0]).clamp(0, img_shape[1]); y1 = (priors[:, 1] - dist[:, 1]).clamp(0, img_shape[0])
    x2 = (priors[:, 0] + dist[:, 2]).clamp(0, img_shape[1]); y2 = (priors[:, 1] + dist[:, 3]).clamp(0, img_shape[0])
    sf = [1 / s for s in scale_factor]
    boxes = torch.stack([x1, y1, x2, y2], 1) * torch.tensor(sf * 2, dtype=torch.float32)
    if min_bbox_size >= 0:
        ok = ((boxes[:, 2] - boxes[:, 0]) > min_bbox_size) & ((boxes[:, 3] - boxes[:, 1]) > min_bbox_size)
        scores, labels, boxes, priors, kernels = scores[ok], labels[ok], boxes[ok], priors[ok], kernels[ok]
    if boxes.numel() == 0:
        return dict(n=0)
    # batched_nms (class offsets) -> nms -> max_per_img
    off = labels.to(boxes) * (boxes.max() + 1)
    scores, order = scores.sort(descending=True, stable=True)
    boxes, labels, priors, kernels, off = boxes[order], labels[order], priors[order], kernels[order], off[order]
    keep = _nms(boxes + off[:, None], nms_iou)[:max_per_img]
    scores, boxes, labels, priors, kernels = scores[keep], boxes[keep], labels[keep], priors[keep], kernels[keep]
    logits = head.mask_logits(mask_feat, kernels, priors)
    stride = head.strides[0]
    up = F.interpolate(logits.unsqueeze(0), scale_factor=stride, mode='bilinear')
    up = F.interpolate(up, size=[math.ceil(up.shape[-2] * sf[0]), math.ceil(up.shape[-1] * sf[1])], mode='bilinear',
                       align_corners=False)[..., :ori_shape[0], :ori_shape[1]]
    masks = up.sigmoid().squeeze(0) > mask_thr_binary
    return dict(n=int(boxes.shape[0]), scores=scores.numpy(), bboxes=boxes.numpy(), labels=labels.numpy(), masks=masks.numpy(),
                logits=logits.numpy(), mask_prob=up.sigmoid().squeeze(0).numpy(),
                raw=dict(cls=[c.sigmoid().numpy() for c in cls_scores], reg=[b.numpy() for b in bbox_preds],
                         kern=[k.numpy() for k in kernel_preds], mask_feat=mask_feat.numpy()))
