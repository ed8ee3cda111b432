"""CPU checks of the plan-time recipe logic (jit.hip / plan.hip) through ndfft_explain_plan: no GPU is needed, nothing is launched.
Every recipe the planner hands to a specialised kernel must be self-consistent -- a wrong one only shows up on a GPU as a wrong answer or a failed compile."""
import math
import os
import re
import subprocess

import pytest

from ndrustfft_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LDS = 160 * 1024
BFLY = set(range(2, 17)) | {17, 19, 23, 29, 31}


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(_lib.LIB_PATH):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "ndrustfft_amd", "csrc"), "-s", "-j4"])
    return _lib.Library()


def _fields(line):
    return dict(kv.split("=", 1) for kv in line.split())


def _prod(radix):
    return math.prod(int(r) for r in radix.split(".")) if radix != "-" else 1


def _largest_prime(n):
    p, m, f = 1, n, 2
    while f * f <= m:
        while m % f == 0:
            p, m = f, m // f
        f += 1
    return m if m > 1 else p


def _check_line(d, dtype):
    F = int(d["F"])
    esz = 8 if dtype == _lib.F32 else 16
    route = d.get("route", d.get("fallback"))
    if d.get("route") == "rader":
        p, M, tpl, e = int(d["p"]), int(d["M"]), int(d["tpl"]), int(d["e"])
        mc1, mc2 = (int(x) for x in d["mc"].split("x"))
        mc = mc1 * mc2
        assert p == _largest_prime(F) and mc * p == F and math.gcd(mc, p) == 1, d
        assert M == ((p - 1) // 2 if "half_conv" in d else p - 1) and _prod(d["radix"]) == M, d
        assert mc1 in BFLY | {1} and mc2 in BFLY | {1} and mc <= (48 if dtype == _lib.F32 else (33 if "sym_rows" in d else 32)), d
        # sym_rows (round 5, DCT-I slot with an odd cofactor > 1): only (mc + 1) / 2 of the mc Rader transforms run, on that many groups of tpl threads
        rows = int(d["sym_rows"]) if "sym_rows" in d else mc
        if "sym_rows" in d:
            assert d["slot"] == "DCT1" and mc % 2 == 1 and rows == (mc + 1) // 2 and (("half_conv" in d) == (mc == 1)), d
        assert 1 <= tpl and tpl * rows <= 1024 and int(d["lanes"]) >= 1 and int(d["lanes"]) * tpl * rows <= 1024, d
        radices = [int(r) for r in d["radix"].split(".")]
        assert all(r in BFLY for r in radices), d
        # f64: at most 21 elements per thread, except where one pass of radix 23 / 29 / 31 needs that many (round 6: the wide radices are open to f64 too)
        wide = max([r for r in radices if r > 21], default=0)
        assert e == max(-(-(M // r) // tpl) * r for r in radices) and e <= (32 if dtype == _lib.F32 else max(21, wide)), d
        zlen = ((p + 1) // 2) * mc if "sym_rows" in d else F
        zraw = max(zlen + (zlen >> 4) + 3, (F + 2) // 2 if "sym_rows" in d else 0)
        lane = max(rows * (M + (M >> 4) + 2), zraw) | 1
        assert int(d["lanes"]) * lane * esz <= LDS, d
    if d.get("route") == "jit":
        tpl, e = int(d["tpl"]), int(d["e"])
        radices = [int(r) for r in d["radix"].split(".")]
        assert _prod(d["radix"]) == F and all(2 <= r <= 16 for r in radices) and tpl >= 1, d
        assert e == max(-(-(F // r) // tpl) * r for r in radices) and e <= 32, d
    if "fs_tpl" in d:      # round 6: the recipe this length runs as a FACTOR of a row four-step (hiprtc passes, jit.hip: jit_fourstep_choose)
        tpl, e = int(d["fs_tpl"]), int(d["fs_e"])
        radices = [int(r) for r in d["fs_radix"].split(".")]
        assert _prod(d["fs_radix"]) == F and all(2 <= r <= 16 for r in radices) and 48 <= F <= 2048, d
        assert e == max(-(-(F // r) // tpl) * r for r in radices) and e <= (32 if dtype == _lib.F32 else 24), d
        lanes = 16 if (dtype == _lib.F32 and 16 * tpl <= 1024) else 8
        assert lanes * tpl <= 1024, d
        e0 = -(-(F // radices[0]) // tpl) * radices[0]
        assert lanes * (((F + (F >> 4) + 2) | 1) + e0) * esz <= LDS, d
    if "blueM" in d:
        M = int(d["blueM"])
        assert M >= 2 * F - 1, d
        p2 = 1 << (2 * F - 2).bit_length()
        if M & (M - 1):      # a smooth length instead of the power of two: 13-smooth and clearly shorter
            m = M
            for f in (2, 3, 5, 7, 11, 13):
                while m % f == 0:
                    m //= f
            assert m == 1 and 100 * M <= 65 * p2, d
            assert _prod(d["blue_radix"]) == M, d
        else:
            assert M == p2, d
    assert route is not None, d


@pytest.mark.parametrize("dtype", [_lib.F32, _lib.F64])
def test_recipes_are_self_consistent(lib, dtype):
    seen = set()
    sizes = list(range(2, 700)) + list(range(701, 4200, 7)) + [1009, 2017, 3027, 4001, 4093, 4099, 7001, 8191, 8402, 10007, 16001, 65537]
    for kind in (_lib.KIND_C2C, _lib.KIND_R2C, _lib.KIND_DCT):
        for n in sizes:
            text = lib.explain_plan(kind, dtype, n)
            assert text.strip(), (kind, n)
            for line in text.strip().splitlines():
                d = _fields(line)
                _check_line(d, dtype)
                seen.add(d.get("route"))
    assert {"rader", "jit", "pow2", "four_step"} <= seen, seen
    # a long smooth lane whose factors are not powers of two plans the two-pass form with hiprtc passes
    assert "jit_passes=2" in lib.explain_plan(_lib.KIND_C2C, dtype, 196608) and "jit_passes=12" in lib.explain_plan(_lib.KIND_C2C, dtype, 200000)
    # ... and its real ops plan the real four-step on even factors N1 x N2 = n (f64: biased to a power-of-two N2); odd lengths and lengths not divisible by 4 do not
    m = re.search(r"real_four_step=(\d+)x(\d+) ops=(\d+)", lib.explain_plan(_lib.KIND_R2C, dtype, 196608))
    assert m and int(m.group(1)) * int(m.group(2)) == 196608 and int(m.group(1)) % 2 == 0 and int(m.group(2)) % 2 == 0 and int(m.group(3)) & 1, m
    if dtype == _lib.F64:
        assert m.group(2) == "512", m.group(0)
    assert "real_four_step" not in lib.explain_plan(_lib.KIND_R2C, dtype, 196610) and "real_four_step" not in lib.explain_plan(_lib.KIND_DCT, dtype, 3 ** 11)
    assert seen & {"blue_reg", "blue_lds", "blue_global"}, seen


def test_known_recipes(lib):
    """A few recipes whose choice was measured: they should not drift silently."""
    d = _fields(lib.explain_plan(_lib.KIND_C2C, _lib.F64, 1009).splitlines()[0])
    assert d["route"] == "rader" and d["p"] == "1009" and d["mc"] == "1x1" and d["radix"] == "12.12.7" and d["tpl"] == "84"
    d = _fields(lib.explain_plan(_lib.KIND_C2C, _lib.F64, 511).splitlines()[0])
    assert d["route"] == "rader" and d["p"] == "73" and d["mc"] == "7x1" and d["radix"] == "9.8" and d["tpl"] == "9"
    d = _fields(lib.explain_plan(_lib.KIND_C2C, _lib.F64, 513).splitlines()[0])
    assert d["route"] == "rader" and d["p"] == "19" and d["mc"] == "9x3"
    d = _fields(lib.explain_plan(_lib.KIND_C2C, _lib.F64, 263).splitlines()[0])          # 262 = 2 x 131: Bluestein, on a smooth length
    assert d["route"] == "blue_reg" and int(d["blueM"]) < 1024
    d = _fields(lib.explain_plan(_lib.KIND_C2C, _lib.F64, 2039).splitlines()[0])         # 2038 = 2 x 1019: Bluestein, power of two
    assert d["route"] == "blue_reg" and d["blueM"] == "4096"
    dct = lib.explain_plan(_lib.KIND_DCT, _lib.F64, 512).splitlines()
    assert any("slot=DCT1 F=511 route=rader" in l for l in dct), dct
    # round 4 (profiles/r07/r07f_col_recipe_sweep_c64.jsonl): column tiles of short f32 C2C lanes keep the default recipe where it holds <= 18
    # elements per thread in no more passes than the rows' re-planned one -- exactly these ten lengths below 256
    alt = [n for n in range(97, 256) if "col_tpl" in lib.explain_plan(_lib.KIND_C2C, _lib.F32, n)]
    assert alt == [98, 99, 100, 110, 121, 143, 144, 156, 162, 220], alt
    d = _fields(lib.explain_plan(_lib.KIND_C2C, _lib.F32, 100).splitlines()[0])
    assert d["radix"] == "5.5.4" and d["col_radix"] == "10.10" and d["col_tpl"] == "10"
    assert not any("col_tpl" in lib.explain_plan(_lib.KIND_C2C, _lib.F64, n) for n in (100, 144, 220))
    # round 6: f32 C2C rows whose default recipe has one butterfly per thread in the first or last pass run the same radix list on half the threads
    # (16-byte accesses): exactly these 14 lengths up to 8192, never f64, never a partial-round or re-planned recipe
    rv = [n for n in range(256, 8193) if "rowvec_tpl" in lib.explain_plan(_lib.KIND_C2C, _lib.F32, n).splitlines()[0]]
    assert rv == [432, 500, 576, 648, 864, 1000, 1296, 2000, 2500, 2592, 3456, 3888, 5000, 5184], rv
    d = _fields(lib.explain_plan(_lib.KIND_C2C, _lib.F32, 1000).splitlines()[0])
    assert (d["tpl"], d["e"], d["rowvec_tpl"], d["rowvec_e"], d["radix"]) == ("100", "10", "50", "20", "10.10.10"), d
    assert "rowvec" not in lib.explain_plan(_lib.KIND_C2C, _lib.F64, 1000) and "rowvec" not in lib.explain_plan(_lib.KIND_C2C, _lib.F32, 264)
