#!/usr/bin/env python3
"""CPU side of tools/ubench/mfma_bf16_probe.hip: which arithmetic does v_mfma_f32_32x32x16_bf16 execute?  Candidate models are evaluated
in exact rational arithmetic on the probe's operands and compared with the hardware's result BIT FOR BIT.
usage: mfma_bf16_model.py probe.bin [cases per family]"""
import struct
import sys
from fractions import Fraction

import numpy as np

TWO = Fraction(2)


def frac_of_f32(bits):
    """exact Fraction of an fp32 bit pattern (finite)"""
    s = -1 if bits >> 31 else 1
    e = (bits >> 23) & 0xff
    m = bits & 0x7fffff
    if e == 0:
        return Fraction(s * m, 1 << 149)
    return Fraction(s * (m | 0x800000)) * (TWO ** (e - 150))


def exp_of(x):
    if x == 0:
        return -10 ** 6
    a = abs(x)
    e = a.numerator.bit_length() - a.denominator.bit_length()
    if TWO ** e > a:
        e -= 1
    return e


def round_f32(x, mode='rn'):
    """Fraction -> fp32 bits, round to nearest even ('rn') or toward zero ('rz'); no overflow handling (the probe stays in range)"""
    if x == 0:
        return 0
    s = 0x80000000 if x < 0 else 0
    a = -x if x < 0 else x
    e = max(exp_of(a), -126)
    scaled = a / (TWO ** (e - 23))                   # in [2^23, 2^24) for normals
    n = scaled.numerator // scaled.denominator
    rem = scaled - n
    if mode == 'rn' and (rem > Fraction(1, 2) or (rem == Fraction(1, 2) and (n & 1))):
        n += 1
    if n >= (1 << 24):
        n >>= 1
        e += 1
    if n < (1 << 23):                                # subnormal
        return s | n
    return s | ((e + 127) << 23) | (n & 0x7fffff)


def trunc_to_grid(x, lsb_exp):
    """truncate (toward zero) a Fraction to a multiple of 2^lsb_exp"""
    q = x / (TWO ** lsb_exp)
    n = abs(q.numerator) // q.denominator
    return (n if q >= 0 else -n) * (TWO ** lsb_exp)


def fl(x, mode='rn'):
    return frac_of_f32(round_f32(x, mode))


def models(c, p):
    """c: Fraction accumulator input, p: 16 exact products (Fractions) in k order -> {name: fp32 bits}"""
    out = {}
    tot = sum(p)
    out['exact: RN(c + sum p)'] = round_f32(c + tot)
    out['exact: RZ(c + sum p)'] = round_f32(c + tot, 'rz')
    acc = c
    for x in p:
        acc = fl(acc + x)
    out['fmaf chain k = 0..15'] = round_f32(acc)
    out['RN(RN(sum p) + c)'] = round_f32(fl(tot) + c)
    for g in (2, 4, 8):
        for mode in ('rn', 'rz'):
            acc = c
            for i in range(0, 16, g):
                acc = fl(acc + sum(p[i:i + g]), mode)
            out['groups of %d exact, %s chain' % (g, mode.upper())] = round_f32(acc)
    acc = c
    for i in range(8):
        acc = fl(acc + p[i] + p[i + 8])
    out['pairs (k, k + 8) exact, RN chain'] = round_f32(acc)
    # aligned-truncation models: every addend (c and the products) is truncated to W bits below the largest exponent, summed exactly, rounded
    terms = [c] + list(p)
    emax = max(exp_of(t) for t in terms)
    for W in (24, 25, 26, 27, 28, 30, 32, 36, 40, 48):
        ssum = sum(trunc_to_grid(t, emax - W) for t in terms)
        out['align to max exp, keep %d bits, RN' % W] = round_f32(ssum)
        out['align to max exp, keep %d bits, RZ' % W] = round_f32(ssum, 'rz')
    for g in (4, 8):
        for W in (24, 26, 28, 32, 40):
            for mode in ('rn', 'rz'):
                acc = c
                for i in range(0, 16, g):
                    ts = [acc] + list(p[i:i + g])
                    em = max(exp_of(t) for t in ts)
                    acc = fl(sum(trunc_to_grid(t, em - W) for t in ts), mode)
                out['groups of %d, aligned %d bits, %s chain' % (g, W, mode.upper())] = round_f32(acc)
    return out


def main(path, limit_per_family=400):
    raw = open(path, 'rb').read()
    T = struct.unpack_from('<i', raw, 0)[0]
    off = 8
    A = np.frombuffer(raw, np.uint16, T * 32 * 16, off).reshape(T, 32, 16); off += A.nbytes
    B = np.frombuffer(raw, np.uint16, T * 16 * 32, off).reshape(T, 16, 32); off += B.nbytes
    C = np.frombuffer(raw, np.uint32, T * 1024, off).reshape(T, 32, 32); off += C.nbytes
    D = np.frombuffer(raw, np.uint32, T * 1024, off).reshape(T, 32, 32)
    fam_names = ['narrow exponents', 'wide exponents', 'cancellation', 'one huge + tiny', 'huge C', 'small integers (exact: layout check)']
    rng = np.random.default_rng(0)
    overall = {}
    total = 0
    for fam in range(6):
        hits, n = {}, 0
        ts = [t for t in range(T) if t % 6 == fam]
        per = limit_per_family // len(ts) + 1
        picks = [(t, int(i), int(j)) for t in ts for i, j in zip(rng.integers(0, 32, per), rng.integers(0, 32, per))]
        for t, i, j in picks[:limit_per_family]:
            a = [frac_of_f32(int(A[t, i, k]) << 16) for k in range(16)]
            b = [frac_of_f32(int(B[t, k, j]) << 16) for k in range(16)]
            p = [x * y for x, y in zip(a, b)]
            c = frac_of_f32(int(C[t, i, j]))
            got = int(D[t, i, j])
            for name, bits in models(c, p).items():
                ok = 1 if bits == got else 0
                hits[name] = hits.get(name, 0) + ok
                overall[name] = overall.get(name, 0) + ok
            n += 1
        total += n
        print("family %d (%s): %d cases" % (fam, fam_names[fam], n))
        for name, h in sorted(hits.items(), key=lambda kv: -kv[1])[:6]:
            print("    %-46s %5d / %d  (%.1f %%)" % (name, h, n, 100.0 * h / n))
    print("all families: %d cases" % total)
    for name, h in sorted(overall.items(), key=lambda kv: -kv[1])[:10]:
        print("    %-46s %5d / %d  (%.2f %%)" % (name, h, total, 100.0 * h / total))
    sys.stdout.flush()


if __name__ == '__main__':
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 400)
