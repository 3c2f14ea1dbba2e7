import os, sys, time
import numpy as np, torch
sys.path.insert(0, '/root/repo')
os.environ["CSM_SYNTHETIC_WEIGHTS"] = "1"
from cartoonsegmentation_amd import ops, synth
from cartoonsegmentation_amd.kenburns import KenBurnsConfig, KenBurnsPipeline
size = 1024
cfg = KenBurnsConfig(det_ckpt='synthetic', depth_est='leres', depth_est_size=640, max_size=size, refine_crf=False, depth_field=True,
                     focal=size / 2.0, mask_refine_kwargs={'refine_method': 'refinenet_isnet', 'refine_size': 720})
pipe = KenBurnsPipeline(cfg); pipe.max_instances = 2
img = torch.from_numpy(synth.image_u8(size, size, 1234)).cuda()
def sync():
    torch.cuda.synchronize(); return time.perf_counter()
for rep, ns in enumerate((1, 1, 3, 3, 3, 1)):
    pipe.frame_streams = ns
    kc = pipe.generate_kenburns_config(img)
    objFrom = {'fltCenterU': size / 2.0, 'fltCenterV': size / 2.0, 'intCropWidth': int(np.floor(0.97 * size)), 'intCropHeight': int(np.floor(0.97 * size))}
    objTo = pipe.process_autozoom({'fltShift': 100.0, 'fltZoom': 1.25, 'objFrom': objFrom}, kc)
    settings = {'fltSteps': np.linspace(0.0, 1.0, 75).tolist(), 'objFrom': objFrom, 'objTo': objTo, 'boolInpaint': True}
    t0 = sync()
    pipe.process_kenburns(settings, kc, True, False, to_numpy=False); t1 = sync()
    pipe.process_kenburns(settings, kc, False, False, to_numpy=False); t2 = sync()
    print("rep %d ns %d: inpaint+75 dof frames %.1f ms, 75 dof frames alone %.1f ms -> inpaint %.1f ms" % (rep, ns, (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t1 - t0 - (t2 - t1)) * 1e3), flush=True)
