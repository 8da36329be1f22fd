"""The ray set-up's shared-reciprocal divisions (volym_amd/csrc/raymarch_device.h: div_pixel, rcp_refined / div_by; the
divisions of shaders/importance_driven_volume_rendering.wgsl:221-241) restated on the CPU with fmaf and checked against
IEEE division (oracle/volym_oracle.c vo_check_*).  CPU only: this pins the arithmetic argument -- the sequence the product
issues is the one a correctly rounded quotient comes out of --; on the device the two forms are compared bit for bit by
volym_selftest_ray_setup (tests/test_gpu_parity.py::test_ray_setup_selftest), and the frames by every oracle parity test."""
import ctypes as C

import pytest


@pytest.fixture(scope="module")
def olib(oracle):
    L = oracle.lib()
    L.vo_check_pixel_quotients.restype = C.c_long
    L.vo_check_pixel_quotients.argtypes = [C.c_int]
    L.vo_check_shared_division.restype = C.c_long
    L.vo_check_shared_division.argtypes = [C.c_uint64, C.c_long, C.c_int, C.POINTER(C.c_long)]
    return L


def test_pixel_quotients_exhaustive(olib):
    """g / W for every 0 <= g < W <= 16384 (134 M pairs; volym_update vouches for exactly that range): r = RN(1 / W) from
    the host and ONE correction step give the bits of the division."""
    assert olib.vo_check_pixel_quotients(16384) == 0


@pytest.mark.parametrize("skew", [0, -1, 1])
def test_shared_reciprocal_sequence(olib, skew):
    """The hardware's division sequence without its scaling steps, operands in [2^-40, 2^40): equal to a / d, also when the
    reciprocal estimate is off by one unit in the last place.  Denominators with a significand of all ones are counted
    apart: with a skewed estimate they are the known exception of such sequences (1 / d sits just above a rounding boundary);
    the product does not depend on it -- it issues the instructions `/` compiles to, so it shares whatever the hardware does."""
    ones = C.c_long(0)
    for seed in (1, 0x9E3779B97F4A7C15):
        assert olib.vo_check_shared_division(seed, 5_000_000, skew, C.byref(ones)) == 0
        if skew == 0:
            assert ones.value == 0
