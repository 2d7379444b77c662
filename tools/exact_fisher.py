#!/usr/bin/env python3
"""Two-sided Fisher exact p-value of one 2x2 table in exact rational arithmetic (minutes for margins of ~1e5):
the sum of pmf(k) over the support with pmf(k) <= pmf(a) (1 + 1e-12), every term a Fraction, one rounding at the end.
Referee for the cases where scipy itself is unreliable (p-values near the underflow limit: tests/test_gpu_parity.py
test_fisher_p_near_underflow_is_the_exact_sum).

    python tools/exact_fisher.py 41976 5113 372553 78120      ->  6.575064524545e-310
    python tools/exact_fisher.py 58002 90233 44143 91836      ->  3.450966061159082e-300
"""
import sys
from fractions import Fraction
from math import comb


def exact_two_sided(a, b, c, d, terms=4000):
    n1, n2, nn = a + b, c + d, a + c
    M = n1 + n2
    den = comb(M, nn)
    lo_k, hi_k = max(0, nn - n2), min(n1, nn)

    def pmf(k):
        return Fraction(comb(n1, k) * comb(n2, nn - k), den)
    pa = pmf(a)
    thr = pa * (1 + Fraction(1, 10 ** 12))
    mode = int((nn + 1) * (n1 + 1) / (M + 2))
    up = a >= mode
    total, p, k = Fraction(0), pa, a
    for _ in range(terms):                       # the tail a lies in, outwards
        total += p
        if up:
            if k + 1 > hi_k:
                break
            p *= Fraction((n1 - k) * (nn - k), (k + 1) * (n2 - nn + k + 1)); k += 1
        else:
            if k - 1 < lo_k:
                break
            p *= Fraction(k * (n2 - nn + k), (n1 - k + 1) * (nn - k + 1)); k -= 1
    # the other side: the outermost-from-the-mode run with pmf <= thr
    if up:
        lo, hi = lo_k, mode
        if pmf(lo) > thr:
            return float(total)
        while lo < hi:
            mid = (lo + hi + 1) // 2
            if pmf(mid) <= thr:
                lo = mid
            else:
                hi = mid - 1
        p, k = pmf(lo), lo
        for _ in range(terms):
            total += p
            if k - 1 < lo_k:
                break
            p *= Fraction(k * (n2 - nn + k), (n1 - k + 1) * (nn - k + 1)); k -= 1
    else:
        lo, hi = mode, hi_k
        if pmf(hi) > thr:
            return float(total)
        while lo < hi:
            mid = (lo + hi) // 2
            if pmf(mid) <= thr:
                hi = mid
            else:
                lo = mid + 1
        p, k = pmf(lo), lo
        for _ in range(terms):
            total += p
            if k + 1 > hi_k:
                break
            p *= Fraction((n1 - k) * (nn - k), (k + 1) * (n2 - nn + k + 1)); k += 1
    return float(min(total, Fraction(1)))


if __name__ == "__main__":
    print(repr(exact_two_sided(*[int(x) for x in sys.argv[1:5]])))
