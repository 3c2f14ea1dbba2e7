#!/usr/bin/env python3
"""ISA-level check of the LDS-DMA kernels of libcsm355.so: the barrier / LDS-read hazard must not be able to come back.

The convolution kernels refill an LDS stage by DMA (`buffer_load_dwordx4 ... lds`) right behind a raw `s_barrier`; the stage being
refilled is the one the PREVIOUS compute phase read its MFMA fragments from.  hipcc may sink the last MFMAs of that phase -- and with
them the `s_waitcnt lgkmcnt` for the `ds_read_b128`s that feed them -- below the barrier, so a wave could arrive at the barrier with
fragment reads still in flight while another wave's DMA into the same stage is issued behind it (seen once in round 3: a few dozen
wrong values in 13 M).  The kernels therefore carry an explicit `s_waitcnt ... lgkmcnt(0)` in front of every barrier.  This tool
disassembles the gfx950 code objects of the shared library and fails unless, in EVERY kernel that contains an LDS-DMA load, EVERY
`s_barrier` is preceded -- in the same basic block, with no LDS instruction and no branch in between -- by an `s_waitcnt` whose
lgkmcnt field is 0.

usage: tools/check_isa_barriers.py [path/to/libcsm355.so]      exit status 0 = all barriers guarded
"""
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OBJDUMP_CANDIDATES = ["/opt/rocm/lib/llvm/bin/llvm-objdump", "/opt/rocm/llvm/bin/llvm-objdump", shutil.which("llvm-objdump") or ""]


def find_objdump():
    for c in OBJDUMP_CANDIDATES:
        if c and os.path.exists(c):
            return c
    return None


def extract_code_objects(so_path, objdump, tmp):
    """llvm-objdump --offloading writes the bundles next to its input: work on a copy inside `tmp`"""
    local = os.path.join(tmp, os.path.basename(so_path))
    shutil.copy(so_path, local)
    subprocess.run([objdump, "--offloading", local], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, cwd=tmp)
    return sorted(os.path.join(tmp, f) for f in os.listdir(tmp) if "gfx950" in f and f != os.path.basename(so_path))


FUNC_RE = re.compile(r"^([0-9a-f]+) <(.+)>:$")
INSN_RE = re.compile(r"^\s+(\S+)\s*(.*?)\s*//\s*([0-9A-Fa-f]+):\s*((?:[0-9A-Fa-f]{8}\s*)+)(?:<.*)?$")


def parse_functions(text):
    """-> {name: [(addr, mnemonic, operands, first encoding dword)]}"""
    funcs, cur = {}, None
    for line in text.splitlines():
        m = FUNC_RE.match(line)
        if m:
            cur = funcs.setdefault(m.group(2), [])
            continue
        if cur is None:
            continue
        m = INSN_RE.match(line)
        if m:
            enc = m.group(4).split()
            cur.append((int(m.group(3), 16), m.group(1), m.group(2), int(enc[0], 16) if enc else 0))
    return funcs


def branch_targets(insns):
    """addresses that start a basic block because a branch lands there (SOPP simm16 = dword offset from the next instruction)"""
    t = set()
    for addr, mn, _ops, enc in insns:
        if mn.startswith("s_cbranch") or mn == "s_branch":
            simm = enc & 0xffff
            if simm & 0x8000:
                simm -= 0x10000
            t.add(addr + 4 + 4 * simm)
    return t


def lgkm_zero(ops):
    return "lgkmcnt(0)" in ops.replace(" ", "")


def check_kernel(name, insns):
    """-> list of problems (strings) for one kernel"""
    problems = []
    leaders = branch_targets(insns)
    n_bar = 0
    for i, (addr, mn, ops, _enc) in enumerate(insns):
        if mn != "s_barrier":
            continue
        n_bar += 1
        ok, why = False, "start of the kernel reached"
        j = i
        while j > 0:
            if insns[j][0] in leaders:                       # a branch lands here: what precedes does not dominate the barrier
                why = "basic block boundary at 0x%x in front of the barrier" % insns[j][0]
                break
            j -= 1
            a2, m2, o2, _ = insns[j]
            if m2 == "s_waitcnt" and lgkm_zero(o2):
                ok = True
                break
            if m2.startswith("ds_"):
                why = "LDS instruction %s at 0x%x between the last lgkmcnt(0) wait and the barrier" % (m2, a2)
                break
            if m2.startswith("s_cbranch") or m2 in ("s_branch", "s_endpgm", "s_setpc_b64"):
                why = "branch %s at 0x%x in front of the barrier" % (m2, a2)
                break
        if not ok:
            problems.append("%s: s_barrier at 0x%x is not dominated by s_waitcnt lgkmcnt(0) in its block (%s)" % (name, addr, why))
    return n_bar, problems


def check_library(so_path, verbose=True):
    objdump = find_objdump()
    if not objdump:
        raise RuntimeError("llvm-objdump not found")
    tmp = tempfile.mkdtemp(prefix="csm_isa_")
    try:
        kernels = barriers = 0
        problems = []
        for co in extract_code_objects(so_path, objdump, tmp):
            text = subprocess.run([objdump, "-d", co], check=True, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL).stdout.decode()
            for name, insns in parse_functions(text).items():
                if not any(mn.startswith("buffer_load") and " lds" in (" " + ops) for _a, mn, ops, _e in insns):
                    continue                                  # only kernels that fill LDS by DMA have the hazard
                n_bar, pr = check_kernel(name, insns)
                kernels += 1
                barriers += n_bar
                problems += pr
        if verbose:
            print("%s: %d LDS-DMA kernels, %d barriers checked, %d unguarded" % (os.path.basename(so_path), kernels, barriers, len(problems)))
            for p in problems[:40]:
                print("  " + p)
        return kernels, barriers, problems
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    so = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "cartoonsegmentation_amd", "libcsm355.so")
    k, b, pr = check_library(so)
    if k == 0 or b == 0:
        print("no LDS-DMA kernels / barriers found: the check did not see what it is meant to see")
        sys.exit(2)
    sys.exit(1 if pr else 0)
