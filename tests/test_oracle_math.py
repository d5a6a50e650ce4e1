"""Oracle checks that need no GPU: the reference's own known-answer tests for helpers on this path
(src/common/math.rs:260-300), the deterministic math substrate, Sobol identities, filter table."""
import ctypes as C
import math
import os

import numpy as np

from conftest import ROOT


def test_log2_int_reference_kat(orc):
    """math.rs:264-274 test_log2_int"""
    L = orc.lib()
    for i in range(63):
        assert L.orc_log2_int(1 << i) == i
    for i in range(1, 63):
        assert L.orc_log2_int((1 << i) + 1) == i


def _solve(L, a, b):
    a = (C.c_float * 4)(*a)
    b = (C.c_float * 2)(*b)
    x = (C.c_float * 2)()
    ok = L.orc_solve_2x2(a, b, x)
    return ok, (x[0], x[1])


def test_solve_linear_system_2x2_reference_kat(orc):
    """math.rs:276-299 test_solve_linear_system_2x2 (row-major a)"""
    L = orc.lib()
    assert _solve(L, [0, 1, 1, 0], [2, 4]) == (1, (4.0, 2.0))
    assert _solve(L, [0, 0, 0, 0], [2, 4])[0] == 0
    assert _solve(L, [1, 1, -1, 1], [2, 2]) == (1, (0.0, 2.0))


def test_next_float_quirk(orc):
    """Q32: next_float_down moves positive values UP (math.rs:98-103 has the branches swapped)."""
    L = orc.lib()
    one = np.float32(1.0)
    assert L.orc_next_float_up(1.0) == float(np.nextafter(one, np.float32(2)))
    assert L.orc_next_float_down(1.0) == float(np.nextafter(one, np.float32(2)))
    assert L.orc_next_float_down(-1.0) == float(np.nextafter(np.float32(-1), np.float32(0)))
    assert math.isnan(L.orc_next_float_down(0.0))


def test_detmath_is_correctly_rounded_on_samples(orc):
    """include/ptrs_detmath.h against binary64 libm rounded once to binary32."""
    L = orc.lib()
    rng = np.random.default_rng(0)
    cases = [(0, np.sin, -20, 20), (1, np.cos, -20, 20), (2, np.log, 1e-6, 50), (3, np.log2, 1e-8, 8), (4, np.exp, -60, 60), (7, np.arccos, -1, 1), (8, np.tan, -1.4, 1.4)]
    for fn, ref, lo, hi in cases:
        xs = rng.uniform(lo, hi, 20000).astype(np.float32)
        got = np.array([L.orc_detmath(fn, float(x), 0.0) for x in xs], dtype=np.float32)
        exp = ref(xs.astype(np.float64)).astype(np.float32)
        assert np.array_equal(got, exp), fn
    ys, xs = rng.uniform(-4, 4, 20000).astype(np.float32), rng.uniform(-4, 4, 20000).astype(np.float32)
    got = np.array([L.orc_detmath(6, float(y), float(x)) for y, x in zip(ys, xs)], dtype=np.float32)
    assert np.array_equal(got, np.arctan2(ys.astype(np.float64), xs.astype(np.float64)).astype(np.float32))
    assert L.orc_detmath(6, 0.0, -1.0) == float(np.float32(math.pi))


def test_fused_sincos_equals_sin_and_cos(orc):
    """pt_sincosf (one reduction, both polynomials once; what the device's cosine-hemisphere and environment sampling call)
    returns bit for bit what pt_sinf and pt_cosf return -- the oracle keeps calling those two."""
    L = orc.lib()
    rng = np.random.default_rng(1)
    xs = np.concatenate([rng.uniform(-8, 8, 40000), rng.uniform(-1e4, 1e4, 5000), [0.0, -0.0, np.pi / 4, np.pi / 2, np.pi, 3 * np.pi / 2, 1e-30, np.inf, np.nan]]).astype(np.float32)
    for x in xs:
        s0, c0 = np.float32(L.orc_detmath(0, float(x), 0.0)), np.float32(L.orc_detmath(1, float(x), 0.0))
        s1, c1 = np.float32(L.orc_detmath(9, float(x), 0.0)), np.float32(L.orc_detmath(10, float(x), 0.0))
        assert s0.view(np.uint32) == s1.view(np.uint32) or (np.isnan(s0) and np.isnan(s1)), x
        assert c0.view(np.uint32) == c1.view(np.uint32) or (np.isnan(c0) and np.isnan(c1)), x


def _tables():
    raw = open(os.path.join(ROOT, "data", "sobol_tables.bin"), "rb").read()
    assert raw[:8] == b"PTRSSOB1"
    hdr = np.frombuffer(raw, dtype="<u4", count=6, offset=8)
    mats = np.frombuffer(raw, dtype="<u4", count=1024 * 52, offset=32).reshape(1024, 52)
    return hdr, mats


def test_sobol_table_identities():
    hdr, mats = _tables()
    assert list(hdr[:5]) == [1024, 52, 25, 26, 52]
    assert [int(v) for v in mats[0, :32]] == [1 << (31 - i) for i in range(32)]  # dim 0 = van der Corput
    assert not mats[0, 32:].any()


def _radical_inverse_dims01(index, mats):
    v0 = v1 = 0
    i = 0
    while index:
        if index & 1:
            v0 ^= int(mats[0, i])
            v1 ^= int(mats[1, i])
        index >>= 1
        i += 1
    return v0, v1


def test_sobol_interval_index_lands_in_pixel(orc):
    """Independent check of sobol_interval_to_index (lowdiscrepancy.rs:9-39): the UNscrambled first
    two dimensions of the returned index fall in the requested cell of the 2^m grid."""
    _, mats = _tables()
    rng = np.random.default_rng(1)
    for (w, h, spp) in [(256, 256, 16), (1024, 1024, 256), (60, 28, 4)]:
        p = orc.make_params(w, h, spp, 4)
        res = orc.round_up_pow2(max(w + 4, h + 4))
        m = res.bit_length() - 1
        px, py = rng.integers(-2, w + 2, 300), rng.integers(-2, h + 2, 300)
        sn = rng.integers(0, spp, 300)
        _, idx = orc.sobol_samples(p, px, py, sn, np.zeros(300, dtype=np.uint32))
        for x, y, s, i in zip(px, py, sn, idx):
            v0, v1 = _radical_inverse_dims01(int(i), mats)
            assert (v0 >> (32 - m), v1 >> (32 - m)) == (x + 2, y + 2)
            assert int(i) >> (2 * m) == s


def test_camera_sample_offsets_Q1_checkerboard(orc):
    """Q1: the per-pixel scramble (low 32 bits of the Cantor pairing, sobol.rs:83-86) is also
    XOR-ed into dimensions 0/1.  Its top bit is set exactly when x+y is even, which throws the point
    out of the pixel and clamps the film offset to 0 or 1-eps (sobol.rs:185-190); for x+y odd the
    high bits are zero and the offset is a genuine in-pixel jitter.  (SURVEY.md Q1 states the clamp
    for every pixel; its probes were all even-sum pixels.)"""
    p = orc.make_params(1024, 1024, 256, 15)
    rng = np.random.default_rng(2)
    px, py = rng.integers(-2, 1026, 20000), rng.integers(-2, 1026, 20000)
    sn = rng.integers(0, 256, 20000)
    lo, hi = np.float32(0.0), np.float32(float.fromhex("0x1.fffffep-1"))
    even = (px + py) % 2 == 0
    for d in (0, 1):
        v, _ = orc.sobol_samples(p, px, py, sn, np.full(20000, d, dtype=np.uint32))
        clamped = (v == lo) | (v == hi)
        assert clamped[even].all()
        assert clamped[~even].mean() < 0.01
        assert ((v >= lo) & (v <= hi)).all()


def test_filter_table(orc):
    """film.rs:133-144 with GuassianFilter::new(2.0) (filter.rs:61-90)."""
    t = orc.filter_table()
    c = (np.arange(16, dtype=np.float64) + 0.5) * 2.0 / 16.0
    g = np.maximum(0.0, np.exp(-2.0 * c * c) - math.exp(-8.0))
    assert np.allclose(t, np.outer(g, g), rtol=2e-6, atol=1e-9)
    assert np.array_equal(t, t.T) and t[0, 0] == t.max() and (t >= 0).all()
