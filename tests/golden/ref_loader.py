"""Load single files of the reference (/root/reference) by path, with inert stubs
for the third-party packages that are absent in this container.

Fixture-generation tooling only (runs in the build container, never on the GPU
box: /root/reference does not exist there).  Nothing from the reference is
copied; its Python is *imported* and its CUDA kernel text is *captured at run
time* and executed on the CPU through tests/golden/cuda_on_cpu.h.
"""
import ctypes
import hashlib
import importlib.util
import os
import re
import subprocess
import sys
import tempfile
import types

REF = os.environ.get("CSM_REFERENCE", "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))
_BUILD = os.path.join(tempfile.gettempdir(), "csm_golden_build")
FP_CONTRACT = os.environ.get("CSM_GOLDEN_FP_CONTRACT", "off")


def _bare(name):
    if name not in sys.modules:
        m = types.ModuleType(name)
        m.__path__ = []
        sys.modules[name] = m
    return sys.modules[name]


def load_by_path(modname, relpath):
    spec = importlib.util.spec_from_file_location(modname, os.path.join(REF, relpath))
    mod = importlib.util.module_from_spec(spec)
    sys.modules[modname] = mod
    spec.loader.exec_module(mod)
    return mod


class _CpuRawKernel:
    """Stands in for cupy.RawKernel: compiles the kernel text for the host and
    runs the launch grid sequentially."""

    def __init__(self, src, name):
        self.src, self.name = src, name
        self._fn = None

    def _build(self):
        os.makedirs(_BUILD, exist_ok=True)
        src = re.sub(r"#include <[^>]*helper_math.h>", "", self.src)
        m = re.search(r'extern "C" __global__ void ' + self.name + r"\s*\(([^)]*)\)", src)
        params = [p.strip() for p in m.group(1).split(",")]
        names = [p.split()[-1].replace("*", "") for p in params]
        self.ctypes_args = []
        for p in params:
            if "*" in p:
                self.ctypes_args.append(ctypes.c_void_p)
            elif "float" in p:
                self.ctypes_args.append(ctypes.c_float)
            else:
                self.ctypes_args.append(ctypes.c_int)
        driver = (
            '\nextern "C" void run_%s(int gx, int bx, %s) {\n'
            "  gridDim.x = gx; blockDim.x = bx;\n"
            "  for (int b = 0; b < gx; ++b) for (int t = 0; t < bx; ++t) {\n"
            "    blockIdx.x = b; threadIdx.x = t; %s(%s); }\n}\n"
        ) % (self.name, ", ".join(params), self.name, ", ".join(names))
        full = '#include "cuda_on_cpu.h"\n' + src + driver
        tag = hashlib.sha1((full + FP_CONTRACT + ("-O2" if FP_CONTRACT == "fast" else "")).encode()).hexdigest()[:16]
        cpp, so = os.path.join(_BUILD, tag + ".cpp"), os.path.join(_BUILD, tag + ".so")
        if not os.path.exists(so):
            with open(cpp, "w") as f:
                f.write(full)
            cmd = ["g++", "-O1", "-ffp-contract=" + FP_CONTRACT, "-fno-fast-math", "-shared", "-fPIC",
                   "-I", HERE, cpp, "-o", so]
            if FP_CONTRACT == "fast":
                # gcc forms FMAs in the widening-mul pass, which only runs from -O2 (at -O1 the "fast" build contains no
                # vfmadd at all and the bracket would be vacuous); -O2 also inlines the float3 helpers so a*b+c is visible
                cmd[1] = "-O2"
                cmd.insert(1, "-mfma")
            subprocess.check_call(cmd)
        lib = ctypes.CDLL(so)
        self._fn = getattr(lib, "run_" + self.name)
        self._fn.argtypes = [ctypes.c_int, ctypes.c_int] + self.ctypes_args
        self._fn.restype = None

    def __call__(self, grid, block, args):
        if self._fn is None:
            self._build()
        self._fn(int(grid[0]), int(block[0]), *[a for a in args])


def install_stubs():
    import torch
    cupy = types.ModuleType("cupy")
    cupy.memoize = lambda **kw: (lambda f: f)
    cupy.RawKernel = _CpuRawKernel
    cupy.int32 = int
    cupy.float32 = float
    sys.modules["cupy"] = cupy
    tv = _bare("torchvision")
    tvm = _bare("torchvision.models")
    tv.models = tvm
    tvt = _bare("torchvision.transforms")
    tv.transforms = tvt
    torch.Tensor.cuda = lambda self, *a, **k: self   # reference hard-codes .cuda()


def load_warp_modules():
    """returns (models_utils, common, cupy_utils) modules of the reference"""
    install_stubs()
    _bare("utils")
    cu = load_by_path("utils.cupy_utils", "utils/cupy_utils.py")
    _bare("anime_3dkenburns")
    _bare("anime_3dkenburns.models")
    mu = load_by_path("anime_3dkenburns.models.utils", "anime_3dkenburns/models/utils.py")
    co = load_by_path("anime_3dkenburns.common", "anime_3dkenburns/common.py")
    return mu, co, cu


# ---- executing single definitions of files whose module-level imports cannot be satisfied here -----------------------------
def extract_def(relpath, qualname, namespace):
    """Compile ONE function / method of a reference file (its own text, taken from the file at run time through `ast`) and
    define it in `namespace` (which supplies the globals the body needs).  Used for files whose module level imports mmdet /
    mmcv / cv2 / omegaconf...: the definition is executed, the module is not imported.  Returns the function object."""
    import ast
    src = open(os.path.join(REF, relpath)).read()
    tree = ast.parse(src)
    body = tree.body
    parts = qualname.split('.')
    node = None
    for i, part in enumerate(parts):
        node = next((n for n in body if isinstance(n, (ast.FunctionDef, ast.ClassDef)) and n.name == part), None)
        if node is None:
            raise KeyError("%s not found in %s" % (qualname, relpath))
        body = getattr(node, 'body', [])
    if isinstance(node, ast.FunctionDef):
        node.decorator_list = []           # e.g. @torch.no_grad(), @AvoidCUDAOOM.retry_if_cuda_oom: wrappers, not semantics
    mod = ast.Module(body=[node], type_ignores=[])
    ast.fix_missing_locations(mod)
    exec(compile(mod, os.path.join(REF, relpath), 'exec'), namespace)
    return namespace[parts[-1]]


def cv2_stub():
    """the only two cv2 calls the pinned paths reach when no image has to be scaled: copyMakeBorder (constant) ; resize raises,
    so a fixture can never silently depend on an un-vendored OpenCV kernel"""
    import numpy as np
    cv2 = types.ModuleType("cv2")
    cv2.BORDER_CONSTANT, cv2.INTER_LINEAR, cv2.INTER_AREA, cv2.LINE_AA = 0, 1, 3, 16

    def copyMakeBorder(img, top, bottom, left, right, borderType, value=0):
        pads = [(top, bottom), (left, right)] + [(0, 0)] * (img.ndim - 2)
        v = value[0] if isinstance(value, (tuple, list)) else value
        assert not isinstance(value, (tuple, list)) or all(x == v for x in value)
        return np.pad(img, pads, mode='constant', constant_values=v)

    def resize(*a, **k):
        raise RuntimeError("cv2.resize is not vendored: pinned fixtures must not need it")
    cv2.copyMakeBorder, cv2.resize = copyMakeBorder, resize
    return cv2


def load_anime_instances():
    """the reference's AnimeInstances class (animeinsseg/anime_instances.py) with utils.constants loaded by path"""
    install_stubs()
    sys.modules.setdefault("cv2", cv2_stub())
    _bare("utils")
    load_by_path("utils.constants", "utils/constants.py")
    _bare("animeinsseg")
    return load_by_path("animeinsseg.anime_instances", "animeinsseg/anime_instances.py").AnimeInstances
