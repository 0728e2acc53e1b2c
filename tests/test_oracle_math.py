"""The oracle's deterministic exp/log ("dmath") are faithful restatements of WGSL exp/log (WGSL allows 3+2|x| and 3 ULP);
its fp16 conversions equal IEEE round-to-nearest-even (numpy)."""
import ctypes

import numpy as np


def _call(orc, name, x, out_dtype):
    out = np.zeros(x.shape[0], out_dtype)
    getattr(orc.lib(), name)(ctypes.c_uint32(x.shape[0]), x.ctypes.data_as(ctypes.c_void_p), out.ctypes.data_as(ctypes.c_void_p))
    return out


def _ulp_err(got, ref64):
    ref32 = ref64.astype(np.float32)
    ulp = np.spacing(np.abs(ref32)).astype(np.float64)
    return np.abs(got.astype(np.float64) - ref64) / ulp


def test_exp_within_one_ulp(orc):
    rng = np.random.default_rng(1)
    x = np.concatenate([rng.uniform(-86, 88, 2_000_000), rng.uniform(-1, 1, 500_000), np.linspace(-86, 88, 100_001)]).astype(np.float32)
    got = _call(orc, "orc_test_exp", x, np.float32)
    err = _ulp_err(got, np.exp(x.astype(np.float64)))
    assert err.max() < 1.0, err.max()


def test_exp_edges(orc):
    x = np.array([0.0, -0.0, -86.0, -86.00001, -1e30, 88.0, 88.00001, 1e30, np.nan, -np.inf, np.inf], np.float32)
    got = _call(orc, "orc_test_exp", x, np.float32)
    assert got[0] == 1.0 and got[1] == 1.0
    assert got[2] > 0 and got[3] == 0.0 and got[4] == 0.0 and got[9] == 0.0
    assert np.isfinite(got[5]) and np.isinf(got[6]) and np.isinf(got[7]) and np.isinf(got[10])
    assert np.isnan(got[8])


def test_log_within_one_ulp(orc):
    rng = np.random.default_rng(2)
    bits = rng.integers(1, 0x7F800000, 2_000_000, dtype=np.uint32)
    x = np.concatenate([bits.view(np.float32), rng.uniform(0.5, 2.0, 500_000).astype(np.float32), rng.uniform(1e-3, 128, 500_000).astype(np.float32)])
    got = _call(orc, "orc_test_log", x, np.float32)
    err = _ulp_err(got, np.log(x.astype(np.float64)))
    assert err.max() < 1.0, err.max()
    edge = _call(orc, "orc_test_log", np.array([0.0, -1.0, np.inf, 1.0], np.float32), np.float32)
    assert np.isneginf(edge[0]) and np.isnan(edge[1]) and np.isposinf(edge[2]) and edge[3] == 0.0


def test_fp16_conversions_match_ieee(orc):
    rng = np.random.default_rng(3)
    bits = rng.integers(0, 2**32, 3_000_000, dtype=np.uint64).astype(np.uint32)
    x = bits.view(np.float32)
    x = x[~np.isnan(x)]
    # values around every fp16 rounding boundary, subnormals, overflow threshold
    h = np.arange(0, 0x7C00, dtype=np.uint16).view(np.float16).astype(np.float32)
    mid = (h[:-1] + h[1:]) * 0.5
    x = np.concatenate([x, h, mid, np.nextafter(mid, np.inf), np.nextafter(mid, -np.inf), -mid, np.array([65504, 65519.99, 65520, 65536, 1e10, 2**-24, 2**-25, 2**-25 * 1.0001], np.float32)]).astype(np.float32)
    got = _call(orc, "orc_test_f32_to_f16", x, np.uint16)
    with np.errstate(over="ignore"):
        ref = x.astype(np.float16).view(np.uint16)
    assert np.array_equal(got, ref)
    allh = np.arange(0, 65536, dtype=np.uint32).astype(np.uint16)
    back = _call(orc, "orc_test_f16_to_f32", allh, np.float32)
    refb = allh.view(np.float16).astype(np.float32)
    ok = (back.view(np.uint32) == refb.view(np.uint32)) | (np.isnan(back) & np.isnan(refb))
    assert ok.all()
