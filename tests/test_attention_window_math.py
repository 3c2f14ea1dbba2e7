"""The attention kernel (csrc/tokens.hip k_attention) reads the BEiT relative position bias of a (32 queries) x (32 keys) tile pair from a
WINDOW of the head's table: (2R - 1)(2gw - 1) consecutive entries starting at table row yq0 - yk0 - R + gh, R = 31 // gw + 2, addressed as
base(query) - kterm(key).  This is the index arithmetic of the kernel restated in numpy and checked against timm's
gen_relative_position_index (oracle/dpt_beit_torch.py) for every patch pair of several grids: the window always contains the entry, the
local index is inside it, and it names the same table entry.  (The kernel itself is checked against the oracle on the GPU:
tests/test_gpu_dpt_beit.py; this test pins the derivation the kernel's comments state.)"""
import numpy as np
import pytest

torch = pytest.importorskip("torch")

from oracle.dpt_beit_torch import gen_relative_position_index  # noqa: E402


@pytest.mark.parametrize("gh,gw", [(42, 42), (24, 32), (4, 6), (36, 36), (7, 2), (3, 50), (12, 31)])
def test_window_covers_every_patch_pair(gh, gw):
    N = gh * gw + 1
    idx = gen_relative_position_index((gh, gw)).numpy()
    W2, R = 2 * gw - 1, 31 // gw + 2
    WN, T = (2 * R - 1) * W2, (2 * gh - 1) * (2 * gw - 1) + 3
    tok = np.arange(N)
    y = np.where(tok >= 1, (tok - 1) // gw, 0)
    x = np.where(tok >= 1, (tok - 1) % gw, 0)
    kterm = np.where(tok >= 1, y * W2 + x, 0)                         # staged per key tile by the block
    NT = (N + 31) // 32
    for qt in range(NT):
        q0 = qt * 32
        yq0 = (q0 - 1) // gw if q0 >= 1 else 0
        qs = np.arange(max(q0, 1), min(q0 + 32, N))                   # patch queries of the tile (the class token takes the scalar entries)
        for kt in range(NT):
            j0 = kt * 32
            yk0 = (j0 - 1) // gw if j0 >= 1 else 0
            g0 = (yq0 - yk0 - R + gh) * W2                            # first table entry of the window
            ks = np.arange(max(j0, 1), min(j0 + 32, N))
            if len(qs) == 0 or len(ks) == 0:
                continue
            wbase = (y[qs] - yq0 + yk0 + R - 1) * W2 + x[qs] + gw - 1
            local = wbase[:, None] - kterm[ks][None, :]
            assert local.min() >= 0 and local.max() < WN, (qt, kt, local.min(), local.max(), WN)
            glob = g0 + local
            assert glob.min() >= 0 and glob.max() < T - 3             # a real table entry (the DMA's range check never zeroes a needed one)
            assert np.array_equal(glob, idx[np.ix_(qs, ks)])


def test_class_token_entries():
    gh, gw = 5, 7
    idx = gen_relative_position_index((gh, gw)).numpy()
    T = (2 * gh - 1) * (2 * gw - 1) + 3
    assert idx[0, 0] == T - 1 and (idx[0, 1:] == T - 3).all() and (idx[1:, 0] == T - 2).all()
