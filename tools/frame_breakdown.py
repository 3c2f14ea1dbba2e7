#!/usr/bin/env python3
"""wall-clock breakdown of one bench frame (sync after each phase)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["CSM_SYNTHETIC_WEIGHTS"] = "1"
import torch
from cartoonsegmentation_amd import synth, ops
from cartoonsegmentation_amd.kenburns import KenBurnsConfig, KenBurnsPipeline
dev = torch.device('cuda', 0)
cfg = KenBurnsConfig(det_ckpt='synthetic', depth_est='leres', depth_est_size=640, max_size=1024, refine_crf=False, depth_field=False, focal=512.0,
                     mask_refine_kwargs={'refine_method': 'refinenet_isnet', 'refine_size': 720})
pipe = KenBurnsPipeline(cfg); pipe.max_instances = 2; pipe.overlap_depth = False
img = torch.from_numpy(synth.image_u8(1024, 1024, 1234)).to(dev)
a = pipe.animeinsseg


def T(fn, n=5):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        r = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3, r


a.set_max_instance(2)
t_det, d = T(lambda: a.detect_raw(img))
print("detect_raw (preprocess+RTMDet+decode+nms)  %.2f ms" % t_det)
sel = torch.arange(d['n'], device=dev)
t_mask, masks = T(lambda: a._masks_from(d, sel))
print("mask head + resize + threshold             %.2f ms" % t_mask)
from cartoonsegmentation_amd.anime_instances import AnimeInstances
def refine():
    inst = AnimeInstances(masks.bool(), d['boxes'].int(), d['scores'])
    a._postprocess_refine(inst, img, refine_size=720); return inst
t_ref, inst = T(refine)
print("ISNet refine (prepare + net + threshold)   %.2f ms" % t_ref)
t_inf, inst = T(lambda: a.infer(img, 0.3, cfg.mask_refine_kwargs, max_instances=2))
print("AnimeInsSeg.infer total                    %.2f ms" % t_inf)
img_t = (img.permute(2, 0, 1)[None].float() * (1.0 / 255.0)).contiguous()
t_dep, depth = T(lambda: pipe._depth_est(img_t, img))
print("LeReS depth (input + net + quantize+resize) %.2f ms" % t_dep)
t_cfg, kc = T(lambda: pipe.generate_kenburns_config(img, instances=inst))
print("generate_kenburns_config(instances given)  %.2f ms  (glue = %.2f)" % (t_cfg, t_cfg - t_dep))
t_all, kc = T(lambda: pipe.generate_kenburns_config(img))
print("generate_kenburns_config (all)             %.2f ms" % t_all)
