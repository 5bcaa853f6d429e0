"""CPU suite: the number formatting of the executable's output (csrc/host/cnf2_format.h) must spell what the reference's
fprintf spells -- "%.5lf" for the per-locus rows (cnF2freq.cpp:6183-6188), "%f" / "%lf" / "%d" for the dump
(cnF2freq.cpp:8157-8192).  fmt_fixed takes a short cut only where it can prove the digits; everything here is compared with
snprintf character by character, the decimal ties and their neighbours (where a short cut could go wrong) first of all."""
import ctypes as C

import numpy as np
import pytest

from conftest import build_host_shim


@pytest.fixture(scope="module")
def shim():
    s = build_host_shim()
    s.shim_format_check.restype = C.c_int
    s.shim_format_check.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    s.shim_format_int.restype = C.c_int
    s.shim_format_int.argtypes = [C.c_int, C.c_char_p]
    return s


def check(shim, v, decimals):
    v = np.ascontiguousarray(v, np.float64)
    first, fb = C.c_int(), C.c_int()
    bad = shim.shim_format_check(v.ctypes.data, len(v), decimals, C.byref(first), C.byref(fb))
    assert bad == 0, "%d of %d differ from printf, first %r (%%.%df)" % (bad, len(v), float(v[first.value]), decimals)
    return fb.value


@pytest.mark.parametrize("decimals", [5, 6])
def test_random_values_print_like_printf(shim, decimals):
    rng = np.random.default_rng(decimals)
    n = 2_000_000
    v = np.concatenate([
        rng.random(n),                                   # what rows and certainties are: [0, 1)
        rng.random(n // 4) * 1e-4,                       # small probabilities
        10.0 ** rng.uniform(-320, 12, n // 4) * rng.choice([-1.0, 1.0], n // 4),
        rng.integers(0, 10 ** 6, n // 4) / 10.0 ** decimals,      # values that ARE decimal numbers of N digits (as nearly as a double can)
        np.array([0.0, -0.0, 0.5, 1.0, 0.02, 0.98, 1e-300, 5e-324, np.inf, -np.inf, np.nan, 1e9, 999999999.9999999, 1e15, 1e22,
                  0.999995, 0.9999995, 0.99999949999999, 2.5e-6, 5e-7, -1e-7, 123456789.123456]),
    ])
    fb = check(shim, v, decimals)
    assert fb < len(v) // 50                             # the short cut is the rule, not the exception


@pytest.mark.parametrize("decimals", [5, 6])
def test_decimal_ties_and_their_neighbours(shim, decimals):
    """(k + 1/2) 10^-N is where the rounded digit flips.  Such a number is a double only when k + 1/2 carries enough factors
    of two (k + 1/2 = j 2^-1: exact ties exist for 10^-N = 2^-N 5^-N only when 5^N divides ... never, except through
    rounding): so take the doubles nearest to the ties and walk a few ulp to either side, where the product a 10^N lands
    within rounding of k + 1/2 -- printf decides by the exact binary value, and so must fmt_fixed."""
    rng = np.random.default_rng(100 + decimals)
    k = np.concatenate([np.arange(0, 3000), rng.integers(0, 10 ** 9, 200000)]).astype(np.float64)
    ties = (k + 0.5) / 10.0 ** decimals
    vals = [ties]
    up, dn = ties.copy(), ties.copy()
    for _ in range(4):
        up = np.nextafter(up, np.inf)
        dn = np.nextafter(dn, -np.inf)
        vals += [up.copy(), dn.copy()]
    v = np.concatenate(vals)
    fb = check(shim, np.concatenate([v, -v]), decimals)
    assert fb > len(ties)                                # these are the values the short cut must refuse
    # true ties: multiples of 2^-k with few bits (0.5, 0.25, 0.125 ... at 1-3 decimals are ties of coarser formats; at N
    # decimals x = j / 2^(N+1) with odd j 5^... is a tie only if it has exactly N + 1 decimals ending in 5)
    j = np.arange(1, 200001, 2, dtype=np.float64)
    true_ties = j / 2.0 ** (decimals + 1)                # exact doubles whose decimal expansion has N + 1 digits ending in 5
    check(shim, np.concatenate([true_ties, -true_ties]), decimals)


def test_integers(shim):
    buf = C.create_string_buffer(32)
    for v in [0, 1, 9, 10, 99, 100, 12345, 2 ** 31 - 1, -1, -10, -(2 ** 31)]:
        n = shim.shim_format_int(v, buf)
        assert buf.value[:n].decode() == "%d" % v


@pytest.mark.parametrize("has_prior", [0, 1])
def test_rows_and_dump_text_equal_the_fprintf_rendering(shim, has_prior):
    """csrc/host/cnf2_text.h forms an individual's rows and its part of the dump in memory; the shim renders the same data
    with the reference's format strings (cnF2freq.cpp:6183-6188, 8157-8192) into a memory stream and compares the bytes."""
    shim.shim_text_check.restype = C.c_int
    shim.shim_text_check.argtypes = [C.c_int] + [C.c_void_p] * 6 + [C.c_int, C.c_int, C.c_char_p]
    rng = np.random.default_rng(7 + has_prior)
    M = 5000
    dos = rng.dirichlet([0.3, 0.3, 0.3], M)
    dos[::7] = [1.0, 0.0, 0.0]
    dos[3::11] = [0.999995, 0.000005, 0.0]
    hw = rng.random(M)
    hw[::5] = 0.5
    hw[1::9] = 1e-9
    allele = rng.choice([0, 1, 2, 9], (M, 2)).astype(np.uint8)
    sure = rng.random((M, 2)) * 0.5
    sure[::3] = 0.02
    sure[1::13] = 0.0
    pa = rng.choice([0, 1, 2], (M, 2)).astype(np.uint8)
    ps = np.where(rng.random((M, 2)) < 0.5, 0.02, rng.random((M, 2)) * 0.5)
    arrs = [np.ascontiguousarray(a) for a in (dos, hw, allele, sure, pa, ps)]
    rc = shim.shim_text_check(M, *[a.ctypes.data for a in arrs], has_prior, 12345, b"ind_12345_aux_realf")
    assert rc == 0, {1: "rows differ", 2: "dump differs", 3: "rows and dump differ"}[rc]


def test_dump_lines_parse_like_sscanf(shim):
    """Engine::deserialize reads a marker's line of a dump with parse_dump_line (cnf2_format.h) where the reference uses
    sscanf("%lf %d %d %lf %lf %lf") (cnF2freq.cpp:7994-8062): same return value, same bits in every field -- for what a dump
    holds ("%f" text), for other spellings of numbers, and for lines that are not a marker's line at all."""
    shim.shim_parse_check.restype = C.c_int
    shim.shim_parse_check.argtypes = [C.c_char_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    rng = np.random.default_rng(11)
    lines = []
    for _ in range(200000):
        hw, s1, s2 = rng.random(3)
        a, b = rng.choice([0, 1, 2, 9], 2)
        lines.append("%f\t%d\t%d\t\t%f\t%f %f %f\t%d\t%d\t%f\t%f" % (hw, a, b, 0.0, s1, s2, 0.5, 1, 2, 0.02, 0.02))   # a dump's line
    n_plain = len(lines)
    def num():
        k = rng.integers(0, 12)
        x = rng.random() * 10.0 ** rng.integers(-8, 8)
        return ["%.17g" % x, "%e" % x, "%.3f" % x, "%d" % int(x), "-%f" % x, "+%.2f" % x, ".5", "5.", "nan", "inf", "0x1p-3",
                "%.25f" % x][k]
    for _ in range(100000):
        sep = [" ", "\t", "  ", " \t"][rng.integers(0, 4)]
        f = [num(), str(rng.integers(-3, 12)), str(rng.integers(0, 3)), num(), num(), num()]
        if rng.random() < 0.1:
            f[rng.integers(0, 6)] = ["x", "1.5.2", "", "12abc", "3e", "99999999999", "1234567890123456789012"][rng.integers(0, 7)]
        if rng.random() < 0.1:
            f = f[:rng.integers(0, 6)]
        lines.append(sep.join(f))
    lines += ["", " ", "12 G0_1", "abc", "0.5 1 2 0.0 0.02 0.02", "9007199254740993 1 1 0.1 0.2 0.3", "0.000000\t1\t2\t\t0.000000\t0.020000 0.020000 0.500000"]
    blob = b"\0".join(l.encode() for l in lines) + b"\0"
    first, fast = C.c_int(), C.c_int()
    bad = shim.shim_parse_check(blob, len(lines), C.byref(first), C.byref(fast))
    assert bad == 0, "%d lines differ from sscanf, first: %r" % (bad, lines[first.value])
    assert fast.value >= n_plain                      # every line of a real dump takes the short cut
