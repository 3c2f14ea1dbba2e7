"""Device-side execution of a lowered Program through libcsm355's csm_run_program."""
import ctypes
import os

import torch

from . import _lib
from ._lib import check, stream_ptr


class CompiledProgram:
    """weights + workspace resident in HBM; run() enqueues the whole net on the current stream
    (no allocation, no host sync -> capturable in a hipGraph via torch.cuda.graph; the first run() tunes the conv tiles)."""
    _cache_loaded = False

    def __init__(self, prog, device, weights=None, shared=None):
        """`shared`: a dict owned by the caller in which programs of ONE net at different shapes / batch sizes share their packed weight
        image on the device -- but only when the images are the same: the packing is part of the lowering (a layer's weights are
        Winograd panels or direct tiles depending on its map size), so two shapes may pack the same checkpoint differently.  The key is
        the packed image's own signature (length + per-op packing); a program with another packing gets its own copy.
        (`weights`: an already uploaded image, trusted as is.)"""
        self.prog = prog
        self.ops, self.tensors, w = prog.serialise(oracle=False)
        self.device = torch.device(device)
        if weights is not None:
            self.weights = weights
        elif shared is not None:
            sig = (int(w.size),) + tuple((o['kind'], o['flags'], o['w_off'], o['b_off'], o['aux_off']) for o in prog.ops)
            if sig not in shared:
                shared[sig] = torch.from_numpy(w).to(self.device)
            self.weights = shared[sig]
        else:
            self.weights = torch.from_numpy(w).to(self.device)
        self.workspace = torch.empty(max(prog.workspace_floats, 64), dtype=torch.float32, device=self.device)
        self.n_ext = prog.n_ext
        self._ext = (ctypes.c_void_p * max(self.n_ext, 1))()
        self._tuned = os.environ.get("CSM_AUTOTUNE", "1") != "1"
        self.runs = 0                              # number of run() calls (bench.py weights its per-op profile by it)

    def _bind(self, ext_tensors):
        assert len(ext_tensors) == self.n_ext, (len(ext_tensors), self.n_ext)
        for i, t in enumerate(ext_tensors):
            if not t.is_cuda or t.dtype != torch.float32 or not t.is_contiguous():
                raise _lib.CsmError("program ext tensor %d must be a contiguous float32 device tensor" % i)
            self._ext[i] = t.data_ptr()

    def autotune(self, *ext_tensors, reps=3):
        """time every eligible tile configuration of every conv op once (csm_conv_autotune) and keep the fastest in
        ops[i].tile.  Speed only -- all configurations give the same bits.  Clobbers the workspace and ext outputs."""
        self._bind(ext_tensors)
        cache = os.environ.get("CSM_TUNE_CACHE")          # optional file shared between processes / runs
        if cache and not CompiledProgram._cache_loaded:
            _lib.load().csm_conv_tile_cache_load(cache.encode())
            CompiledProgram._cache_loaded = True
        n = _lib.load().csm_conv_autotune(self.ops, ctypes.c_int(len(self.ops)), self.tensors, ctypes.c_int(len(self.tensors)),
                                          ctypes.c_void_p(self.weights.data_ptr()), ctypes.c_void_p(self.workspace.data_ptr()),
                                          self._ext, ctypes.c_int(self.n_ext), stream_ptr(), ctypes.c_int(reps))
        if n < 0:
            check(-n, "conv_autotune(%s)" % self.prog.name)
        self._tuned = True
        if cache:
            check(_lib.load().csm_conv_tile_cache_save(cache.encode()), "tile_cache_save")
        return n

    def run(self, *ext_tensors):
        if not self._tuned:                       # first call: pick tiles on the real buffers (CSM_AUTOTUNE=0 disables)
            self.autotune(*ext_tensors)
        self._bind(ext_tensors)
        self.runs += 1
        check(_lib.load().csm_run_program(self.ops, ctypes.c_int(len(self.ops)), self.tensors,
                                          ctypes.c_int(len(self.tensors)), ctypes.c_void_p(self.weights.data_ptr()),
                                          ctypes.c_void_p(self.workspace.data_ptr()), self._ext, ctypes.c_int(self.n_ext),
                                          stream_ptr()), "run_program(%s)" % self.prog.name)

    def profile(self, *ext_tensors):
        """per-op durations in ms (HIP events on the launch stream); returns list aligned with prog.ops"""
        if ext_tensors:
            if not self._tuned:
                self.autotune(*ext_tensors)
            self._bind(ext_tensors)
        ms = (ctypes.c_float * len(self.ops))()
        check(_lib.load().csm_run_program_profile(self.ops, ctypes.c_int(len(self.ops)), self.tensors,
                                                  ctypes.c_int(len(self.tensors)), ctypes.c_void_p(self.weights.data_ptr()),
                                                  ctypes.c_void_p(self.workspace.data_ptr()), self._ext, ctypes.c_int(self.n_ext),
                                                  stream_ptr(), ms), "run_program_profile(%s)" % self.prog.name)
        return list(ms)

    def view(self, t):
        """zero-copy NHWC view [n,h,w,c] of a planned tensor whose buffer holds exactly its channels (e.g. net outputs)"""
        b = t.buf
        assert t.coff == 0 and t.c == b.c
        return self.workspace[b.offset:b.offset + b.n * b.h * b.w * b.c].view(b.n, b.h, b.w, b.c)

    def read_view(self, t):
        """debug: copy a planned NHWC view out of the workspace as [n,h,w,c]"""
        b = t.buf
        full = self.workspace[b.offset:b.offset + b.n * b.h * b.w * b.c].view(b.n, b.h, b.w, b.c)
        return full[..., t.coff:t.coff + t.c].clone()
