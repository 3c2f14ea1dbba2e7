"""The LDS-DMA conv kernels turn an output row m into (sample, oy, ox) with two divisions by run-time constants done as
q = (mulhi(n, mul) + n) >> shr,  shr = ceil(log2 d),  mul = floor(2^32 (2^shr - d) / d) + 1     (csrc/nets.hip: fast_div / set_fast_div).
Exact for every dividend below 2^31 (the kernels' rows are below that: views < 2 GiB) -- checked here in numpy for divisors of every
size, including the ho * wo and wo of the three nets; the sum mulhi + n stays below 2^32 (no carry lost in the 32-bit add)."""
import numpy as np
import pytest


def magic(d):
    shr = 0
    while (1 << shr) < d:
        shr += 1
    mul = (((1 << shr) - d) << 32) // d + 1
    assert 0 < mul < 2 ** 32
    return mul, shr


@pytest.mark.parametrize("seed", [0, 1])
def test_fast_div_is_exact_below_2_31(seed):
    rng = np.random.default_rng(seed)
    ds = list(range(1, 1500)) + [int(x) for x in rng.integers(1, 2 ** 31 - 1, 1500)]
    ds += [40 * 40, 80 * 80, 160 * 160, 320 * 320, 360 * 360, 23 * 23, 45 * 45, 90 * 90, 180 * 180, 20 * 20, 640 * 640, 1024 * 1024,
           2 ** 30, 2 ** 30 + 1, 2 ** 31 - 1, 517, 333]
    for d in ds:
        mul, shr = magic(d)
        n = np.concatenate([rng.integers(0, 2 ** 31, 300, dtype=np.int64),
                            np.array([0, 1, d - 1, d, d + 1, 2 * d - 1, 2 * d, 2 ** 31 - 1], dtype=np.int64)])
        n = n[(n >= 0) & (n < 2 ** 31)]
        t = (n * mul) >> 32
        assert ((t + n) < 2 ** 32).all(), d
        assert np.array_equal((t + n) >> shr, n // d), d
