#!/usr/bin/env python3
"""Exact constants of cray_math.h's correctly rounded sincos_cr (double-double evaluation):
pi/2 split into exact 33-bit parts, and sin(k/64), cos(k/64), k = 0..51, as (hi, lo) double pairs.
All values come from exact rational / 70-digit decimal arithmetic; prints C initialisers."""
from decimal import Decimal, getcontext
from fractions import Fraction
import math

getcontext().prec = 90
PI = Decimal("3.14159265358979323846264338327950288419716939937510582097494459230781640628620899862803482534211706798214808651")


def dsin(x):
    x = Decimal(x); term = x; s = x; n = 1
    while abs(term) > Decimal(10) ** -85:
        term = -term * x * x / ((2 * n) * (2 * n + 1)); s += term; n += 1
    return s


def dcos(x):
    x = Decimal(x); term = Decimal(1); s = Decimal(1); n = 1
    while abs(term) > Decimal(10) ** -85:
        term = -term * x * x / ((2 * n - 1) * (2 * n)); s += term; n += 1
    return s


def dd(dec):
    fr = Fraction(dec)
    hi = float(fr); lo = float(fr - Fraction(hi))
    return hi, lo


def top_bits(fr, nbits):
    e = math.floor(math.log2(float(fr)))
    scale = Fraction(2) ** (nbits - 1 - e)
    return Fraction(math.floor(fr * scale)) / scale


if __name__ == '__main__':
    half_pi = Fraction(PI) / 2
    P1 = top_bits(half_pi, 33); r = half_pi - P1
    P2 = top_bits(r, 33); r -= P2
    P3 = top_bits(r, 33); r -= P3
    P4 = Fraction(float(r))
    print('P1..P4 =', ', '.join(float(p).hex() for p in (P1, P2, P3, P4)), ' 2/pi =', float(2 / Fraction(PI)).hex())
    rows = []
    for k in range(52):
        h = Decimal(k) / 64
        sh, sl = dd(dsin(h)); ch, cl = dd(dcos(h))
        rows.append('{%s, %s, %s, %s}' % (sh.hex(), sl.hex(), ch.hex(), cl.hex()))
    print('const double T[52][4] = {\n    ' + ',\n    '.join(rows) + '};')
    for name, fr in [('1/6', Fraction(1, 6)), ('1/120', Fraction(1, 120)), ('1/2', Fraction(1, 2)), ('1/24', Fraction(1, 24))]:
        h, l = float(fr), float(fr - Fraction(float(fr)))
        print(name, h.hex(), l.hex())
