"""GPU: the kernels' pinned arithmetic primitives ("dmath", webdgs_amd/csrc/dmath.h) against the oracle's independent
implementations (oracle/wgsl_shim.hpp), bit for bit, over edge cases and millions of random bit patterns -- the foundation of
every `==` comparison in the other parity tests."""
import ctypes

import numpy as np
import pytest

from webdgs_amd import _lib

pytestmark = pytest.mark.gpu


def _device(hip_device, which, bits):
    src = hip_device.bufferFrom(bits.astype(np.uint32))
    dst = hip_device.createBuffer(4 * bits.size)
    _lib.check(hip_device.lib.wdgs_debug_eval_math(hip_device.handle, which, bits.size, src.ptr, dst.ptr))
    return dst.read(np.uint32, bits.size)


def _oracle(orc, name, arr, out_dtype):
    out = np.zeros(arr.shape[0], out_dtype)
    getattr(orc.lib(), name)(ctypes.c_uint32(arr.shape[0]), arr.ctypes.data_as(ctypes.c_void_p), out.ctypes.data_as(ctypes.c_void_p))
    return out


def _patterns(seed, n=3_000_000):
    rng = np.random.default_rng(seed)
    special = np.array([0x00000000, 0x80000000, 0x00000001, 0x80000001, 0x007FFFFF, 0x00800000, 0x3F800000, 0xBF800000, 0x7F7FFFFF, 0xFF7FFFFF,
                        0x7F800000, 0xFF800000, 0x7FC00000, 0xFFC00001, 0x4F000000, 0xCF000000, 0x4F800000, 0x4EFFFFFF, 0xC2AC0000, 0x42B00000,
                        0xC2AC0001, 0x42B00001, 0x33800000, 0x477FE000, 0x477FF000, 0x38800000, 0x387FC000, 0x33000000, 0x33000001], np.uint32)
    return np.concatenate([special, rng.integers(0, 1 << 32, n, dtype=np.uint64).astype(np.uint32),
                           rng.uniform(-100, 100, n // 4).astype(np.float32).view(np.uint32), rng.uniform(-2, 2, n // 4).astype(np.float32).view(np.uint32)])


def _same(got, ref_bits, what, bits):
    ref_bits = ref_bits.view(np.uint32)
    nan_ok = np.ones(got.shape, bool)
    bad = np.flatnonzero(got != ref_bits)
    assert bad.size == 0, f"{what}: {bad.size} of {got.size} differ; first input bits {int(bits[bad[0]]):#010x}: device {int(got[bad[0]]):#010x} oracle {int(ref_bits[bad[0]]):#010x}"


def test_exp_log_sqrt_rcp_bit_exact(hip_device, orc):
    bits = _patterns(11)
    x = bits.view(np.float32)
    for which, name in ((0, "orc_test_exp"), (1, "orc_test_log"), (6, "orc_test_sqrt"), (7, "orc_test_rcp")):
        got = _device(hip_device, which, bits)
        ref = _oracle(orc, name, x, np.float32).view(np.uint32)
        # NaN results: any NaN payload is a NaN (the two sides may differ in sign/payload of a produced NaN)
        both_nan = np.isnan(got.view(np.float32)) & np.isnan(ref.view(np.float32))
        got, ref = np.where(both_nan, 0x7FC00000, got).astype(np.uint32), np.where(both_nan, 0x7FC00000, ref).astype(np.uint32)
        _same(got, ref, name, bits)


def test_f16_conversions_bit_exact(hip_device, orc):
    bits = _patterns(12)
    got = _device(hip_device, 2, bits)
    ref = _oracle(orc, "orc_test_f32_to_f16", bits.view(np.float32), np.uint16).astype(np.uint32)
    nan = np.isnan(bits.view(np.float32))
    assert np.array_equal(got[~nan], ref[~nan]), "f32 -> f16 (RNE)"
    assert ((got[nan] & 0x7C00) == 0x7C00).all() and ((got[nan] & 0x03FF) != 0).all(), "NaN must stay NaN in f16"
    halves = np.arange(1 << 16, dtype=np.uint32)
    got = _device(hip_device, 3, halves)
    ref = _oracle(orc, "orc_test_f16_to_f32", halves.astype(np.uint16), np.float32).view(np.uint32)
    nanh = np.isnan(ref.view(np.float32))
    assert np.array_equal(got[~nanh], ref[~nanh]), "f16 -> f32"
    assert np.isnan(got[nanh].view(np.float32)).all()


def test_saturating_casts_bit_exact(hip_device, orc):
    bits = _patterns(13)
    x = bits.view(np.float32)
    _same(_device(hip_device, 4, bits), _oracle(orc, "orc_test_to_i32", x, np.int32), "f32 -> i32 (truncate, saturate, NaN -> 0)", bits)
    _same(_device(hip_device, 5, bits), _oracle(orc, "orc_test_to_u32", x, np.uint32), "f32 -> u32 (truncate, saturate, NaN -> 0)", bits)


def test_inrange_forms_equal_the_full_forms_where_the_kernels_use_them(hip_device):
    """backward_rasterize / rasterize / metric_count evaluate exp and the division with the range handling taken out, on arguments
    they have brought into range (dmath.h: wd_exp_inrange, wd_div_inrange).  On those ranges the short forms must return the full
    forms' bits: twelve million arguments over [-86, 87] for exp, and tens of millions of operand
    pairs T / (1 - alpha) over and well beyond the kernels' range for the division."""
    rng = np.random.default_rng(21)
    # exp on [-86, 87]: uniform samples, the interval the kernels' alpha test lives in, arguments near 0, and the end points
    xs = np.concatenate([rng.uniform(-86.0, 87.0, 6_000_000), rng.uniform(-8.0, 0.0, 4_000_000), -np.exp(rng.uniform(-30.0, 4.45, 2_000_000)),
                         np.array([-86.0, -85.99999, -80.0, 87.0, 0.0, -0.0, -1e-30, 1e-30, -5.5, -5.541, 86.999])]).astype(np.float32)
    xs = xs[(xs >= -86.0) & (xs <= 87.0)]
    bits = xs.view(np.uint32)
    _same(_device(hip_device, 8, bits), _device(hip_device, 0, bits), "wd_exp_inrange vs wd_exp on [-86, 87]", bits)
    # division: a = T in [1e-6, 2], b = 1 - alpha in [0.005, 1.5] (the kernels: [1e-4, 1] and [0.01, 0.997]); pairs are neighbours (i, i ^ 1)
    n = 20_000_000
    a = np.exp(rng.uniform(np.log(1e-6), np.log(2.0), n)).astype(np.float32)
    b = np.where(rng.random(n) < 0.5, rng.uniform(0.005, 1.5, n), 1.0 - rng.uniform(1.0 / 255.0, 0.99, n)).astype(np.float32)
    pairs = np.empty(2 * n, np.float32)
    pairs[0::2], pairs[1::2] = a, b
    pb = pairs.view(np.uint32)
    short, full = _device(hip_device, 9, pb)[0::2], _device(hip_device, 10, pb)[0::2]
    _same(short, full, "wd_div_inrange vs wd_div", pb[0::2])
    # the loss kernel's operands: numerators of either sign from 1e-20 to 1e3 and exact zeros, denominators 25 and [1e-8, 1e3]
    n = 10_000_000
    a = (np.exp(rng.uniform(np.log(1e-20), np.log(1e3), n)) * rng.choice([-1.0, 1.0], n)).astype(np.float32)
    a[rng.random(n) < 0.02] = 0.0
    # (not -0: the short form returns +0 for -0 / b; the kernel's numerators are sums that start from +0 and a product with a positive factor)
    b = np.where(rng.random(n) < 0.4, 25.0, np.exp(rng.uniform(np.log(1e-8), np.log(1e3), n))).astype(np.float32)
    pairs = np.empty(2 * n, np.float32)
    pairs[0::2], pairs[1::2] = a, b
    pb = pairs.view(np.uint32)
    short, full = _device(hip_device, 9, pb)[0::2], _device(hip_device, 10, pb)[0::2]
    _same(short, full, "wd_div_inrange vs wd_div on the loss kernel's operand range", pb[0::2])
