#!/usr/bin/env python3
"""Generates the double-double constants of rpg_monocular_pose_estimator_amd/csrc/mpe_ddmath.h (printed to stdout; the
header holds a pasted copy — this script is how they were made and how a reader can check them).  Exact rational
arithmetic (fractions), series carried to > 330 bits; float(Fraction) rounds to nearest."""
from fractions import Fraction as F

BITS = 340


def ln2():
    # ln 2 = sum_{k>=1} 1 / (k 2^k)
    s, k = F(0), 1
    while k < BITS + 10:
        s += F(1, k << k)
        k += 1
    return s


def atan_inv(n):
    # atan(1/n) = sum (-1)^k / ((2k+1) n^(2k+1))
    s, k = F(0), 0
    while True:
        t = F(1, (2 * k + 1) * n ** (2 * k + 1))
        if t.denominator.bit_length() - t.numerator.bit_length() > BITS + 10:
            break
        s += t if k % 2 == 0 else -t
        k += 1
    return s


def pi():
    return 16 * atan_inv(5) - 4 * atan_inv(239)  # Machin


def split(v, n=3, first_bits=None):
    out = []
    for i in range(n):
        if i == 0 and first_bits:
            import math
            e = v.numerator.bit_length() - v.denominator.bit_length()
            sc = first_bits - e
            x = F(round(v * F(2) ** sc), 1) / F(2) ** sc
            d = float(x)
            assert F(d) == x
        else:
            d = float(v)
        out.append(d)
        v -= F(d)
    return out


def cdd(name, parts):
    return "MPE_DD_CONST double %s[%d] = {%s};" % (name, len(parts), ", ".join(float.hex(p) for p in parts))


if __name__ == "__main__":
    L2, PI = ln2(), pi()
    print(cdd("kLn2", split(L2, 3, first_bits=42)), " // ln 2: 42 bits (k * hi is exact for |k| < 2^11), then two doubles")
    print(cdd("kInvLn2", split(1 / L2, 1)))
    print(cdd("kPi", split(PI, 3)))
    print(cdd("kPi2", split(PI / 2, 3)))
    rec = []
    for k in range(0, 26):
        rec += split(F(1, 2 * k + 1), 2)
    print("// 1 / (2k + 1), k = 0 .. 25, as (hi, lo) pairs")
    print("MPE_DD_CONST double kOddRec[52] = {%s};" % ", ".join(float.hex(p) for p in rec))
    fac, f = [], 1
    for n in range(0, 24):
        if n:
            f *= n
        fac += split(F(1, f), 2)
    print("// 1 / n!, n = 0 .. 23, as (hi, lo) pairs")
    print("MPE_DD_CONST double kFacRec[48] = {%s};" % ", ".join(float.hex(p) for p in fac))
