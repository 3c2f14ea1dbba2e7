"""static half of the barrier / LDS-read hazard check (the dynamic half is tests/test_gpu_nets.py::test_dma_kernels_repeated_runs_are_bitwise_stable):
every s_barrier of every LDS-DMA kernel in the built libcsm355.so must be preceded, in its basic block, by s_waitcnt lgkmcnt(0)"""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import check_isa_barriers as isa  # noqa: E402

SO = os.path.join(ROOT, "cartoonsegmentation_amd", "libcsm355.so")


def test_checker_sees_an_unguarded_barrier():
    """the rule itself, on hand-written listings"""
    def ins(lines):
        return [(0x100 + 4 * i, mn, ops, enc) for i, (mn, ops, enc) in enumerate(lines)]
    good = ins([("ds_read_b128", "v[0:3], v4", 0), ("s_waitcnt", "vmcnt(0) lgkmcnt(0)", 0), ("s_add_i32", "s1, s1, 1", 0), ("s_barrier", "", 0)])
    assert isa.check_kernel("good", good) == (1, [])
    late_read = ins([("s_waitcnt", "lgkmcnt(0)", 0), ("ds_read_b128", "v[0:3], v4", 0), ("s_waitcnt", "vmcnt(0)", 0), ("s_barrier", "", 0)])
    assert len(isa.check_kernel("late", late_read)[1]) == 1
    vm_only = ins([("v_mfma_f32_32x32x2_f32", "a[0:15], v0, v1, a[0:15]", 0), ("s_waitcnt", "vmcnt(0)", 0), ("s_barrier", "", 0)])
    assert len(isa.check_kernel("vm", vm_only)[1]) == 1
    # a branch lands between the wait and the barrier (loop head): the wait does not dominate the barrier on the back edge
    loop = ins([("s_waitcnt", "lgkmcnt(0)", 0), ("s_nop", "0", 0), ("s_barrier", "", 0), ("s_cbranch_scc1", "65533", 0xbf85fffd)])
    assert len(isa.check_kernel("loop", loop)[1]) == 1


@pytest.mark.skipif(not os.path.exists(SO) or isa.find_objdump() is None, reason="needs the built library and llvm-objdump")
def test_every_barrier_of_the_lds_dma_kernels_is_guarded():
    kernels, barriers, problems = isa.check_library(SO, verbose=False)
    assert kernels >= 20 and barriers >= 60, (kernels, barriers)      # the check must see the conv kernels it is meant for
    assert not problems, "\n".join(problems[:10])
