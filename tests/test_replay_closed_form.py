"""The closed-form replay of saturated empty-space steps (volym_amd/csrc/raymarch_device.h, replay_saturated)
restated in NumPy float32/uint32 and checked against the plain recurrence t += base of
shaders/importance_driven_volume_rendering.wgsl:263-274 (cur == base).  CPU only: it pins the arithmetic argument
(n steps inside a binade are one integer multiply-add), the GPU parity tests pin the kernel."""
import numpy as np
import pytest

f32 = np.float32
u32 = np.uint32


def bits(x):
    return int(np.array(x, f32).view(u32))


def from_bits(b):
    return np.array(b & 0xFFFFFFFF, u32).view(f32)[()]


def replay_plain(t, t_stop, base):
    n = 0
    while t < t_stop:
        t = f32(t + base)
        n += 1
    return t, n


def replay_closed(t, t_stop, base, rcp_skew=0):
    trips = 0
    while t < t_stop:
        trips += 1
        tb = bits(t)
        ex = tb >> 23
        inv_u = from_bits(((277 - ex) & 0xFF) << 23) if 23 <= 277 - ex <= 254 else f32(0)
        q = f32(base * inv_u)
        kf = f32(np.rint(q))
        if ex < 100 or not (q < f32(4194304.0)) or abs(f32(q - kf)) == f32(0.5) or not (kf >= f32(1.0)):
            t = f32(t + base)
            continue
        m = (tb & 0x7FFFFF) | 0x800000
        K = int(kf)
        xs = f32(t_stop * inv_u)
        lim = int(np.ceil(xs)) if xs < f32(16777216.0) else 0x1000000
        need = lim - m
        assert need >= 1
        r = f32(f32(1.0) / kf)
        r = from_bits(bits(r) + rcp_skew)              # v_rcp_f32 is good to 1 ulp: the fix-ups must absorb that
        n = int(f32(f32(need) * r))
        if n * K < need:
            n += 1
        if n * K < need:
            n += 1
        assert n * K >= need and (n - 1) * K < need, (n, K, need)
        if m + n * K >= 0x1000000:
            m2 = m + (n - 1) * K
            t = from_bits((tb & 0xFF800000) | (m2 & 0x7FFFFF))
            t = f32(t + base)
        else:
            m2 = m + n * K
            t = from_bits((tb & 0xFF800000) | (m2 & 0x7FFFFF))
    return t, trips


@pytest.mark.parametrize("skew", [-1, 0, 1])
def test_closed_form_equals_recurrence(skew):
    rng = np.random.default_rng(20260410 + skew)
    bases = [f32(0.01), f32(0.02), f32(0.005), f32(0.0025), f32(0.1), f32(1.0e-4), f32(0.37), f32(1.0)]
    bases += [f32(x) for x in rng.uniform(1.0e-4, 0.2, 24)]
    # steps whose low mantissa bits make base / ulp(t) an exact half in some binade (ties go to even: the slow path)
    bases += [from_bits(0x3C23D700 | 0x40), from_bits(0x3C23D700 | 0x20), from_bits(0x3C000000 | 0x1)]
    cases = 0
    for base in bases:
        for _ in range(40):
            t0 = f32(rng.uniform(0.0, 3.0)) if rng.random() < 0.9 else f32(rng.choice([0.0, 0.5, 1.0, 2.0, 0.99999994, 1.9999999]))
            span = f32(rng.uniform(0.0, 1.2))
            t_stop = f32(t0 + span)
            if float(span) / float(base) > 20000:      # keep the plain loop short
                t_stop = f32(t0 + f32(20000) * base)
            want, n = replay_plain(t0, t_stop, base)
            got, trips = replay_closed(t0, t_stop, base, skew)
            assert bits(want) == bits(got), (float(t0), float(t_stop), float(base), float(want), float(got))
            cases += 1
    assert cases > 1000


def test_closed_form_needs_few_trips():
    # the benchmark's case: a binade crossing costs one extra trip, nothing else does
    t, trips = replay_closed(f32(0.6180339), f32(2.3), f32(0.01))
    want, n = replay_plain(f32(0.6180339), f32(2.3), f32(0.01))
    assert bits(t) == bits(want) and n > 150 and trips <= 4
