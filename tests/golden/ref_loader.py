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
        tag = hashlib.sha1((full + FP_CONTRACT).encode()).hexdigest()[:16]
        cpp, so = os.path.join(_BUILD, tag + ".cpp"), os.path.join(_BUILD, tag + ".so")
        if not os.path.exists(so):
            with open(cpp, "w") as f:
                f.write(full)
            cmd = ["g++", "-O1", "-ffp-contract=" + FP_CONTRACT, "-fno-fast-math", "-shared", "-fPIC",
                   "-I", HERE, cpp, "-o", so]
            if FP_CONTRACT == "fast":
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
