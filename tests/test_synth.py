"""The synthetic stand-ins for the reference's missing .raw assets are pinned by hash."""
import numpy as np

from volym_amd import synth


def test_lowbias32_known_values():
    """lowbias32 (x ^= x>>16; x *= 0x7feb352d; x ^= x>>15; x *= 0x846ca68b; x ^= x>>16) in plain ints."""
    got = [int(v) for v in synth.lowbias32(np.array([0, 1, 0xFFFFFFFF, 20250310], np.uint32))]
    assert got == [0, 1753845952, 1734902346, 3120490440]


def test_bonsai_hashes_and_density():
    v = synth.synth_bonsai(64)
    assert synth.sha256(v) == "bf0bb35048c1cbfe09c9d3333551ef7c60197a41e68005418490d55f9814a80f"
    v, l = synth.synth_bonsai(64, with_labels=True)
    assert synth.sha256(v) == "bf0bb35048c1cbfe09c9d3333551ef7c60197a41e68005418490d55f9814a80f"
    assert set(np.unique(l)) <= {0, 2, 3, 4}
    assert np.all(v[l == 0] <= 7) and np.all(v[l == 4] == 230)


def test_ball_and_vessels_hashes():
    """the two extra scenes of the scheduling rows (scripts/scene_rows.py): one long dense run per ray / sparse thin tubes"""
    assert synth.sha256(synth.synth_ball(64)) == "2a7d0d413dc232f34f9535c8d91d4afc42902f9e55b9f43043f796e87e371f36"
    v = synth.synth_vessels(64)
    assert synth.sha256(v) == "d79f1153b5a134a9b86aea78f571f1879e1e2e4a3fb0d2ebc1f821b8df7b1965"
    assert 0.02 < float((v >= 39).mean()) < 0.10


def test_bonsai256_and_teapot_hashes():
    v = synth.synth_bonsai(256)
    assert synth.sha256(v) == "e0be5753326650e825c59a8f0fc756aac1bf2ee1d5e7088bcccda74e2f361718"
    frac = float((v >= 39).mean())           # density_threshold 0.15 -> byte 39
    assert 0.08 < frac < 0.15
    d, l = synth.synth_teapot()
    assert d.size == 256 * 256 * 178         # shorter than 256^3: exercises the zero-pad path
    assert synth.sha256(d) == "8b9ca080da554f5a0800e8a8c5ed42caedc593ae14fa3fff8abdd54945db4761"
    assert synth.sha256(l) == "c50a55185c5abfcc2036a1626b9fb463f74a8f635dad4bb81ceadbb568756a3a"
    assert set(np.unique(l)) == {0, 2, 3, 4}


def test_cpp_generators_match_python(volym_lib):
    """The C++ generators in libvolym_hip.so (used by the `volym` binary) produce the same bytes."""
    import ctypes as C
    u8 = lambda a: a.ctypes.data_as(C.POINTER(C.c_uint8))
    d = np.empty(64 ** 3, np.uint8)
    l = np.empty(64 ** 3, np.uint8)
    assert volym_lib.volym_synth_bonsai(64, synth.DEFAULT_SEED, u8(d), u8(l)) == 0
    pv, pl = synth.synth_bonsai(64, with_labels=True)
    assert np.array_equal(d, pv) and np.array_equal(l, pl)
    n = 256 * 256 * 178
    d, l = np.empty(n, np.uint8), np.empty(n, np.uint8)
    assert volym_lib.volym_synth_teapot(256, 256, 178, synth.DEFAULT_SEED, u8(d), u8(l)) == 0
    assert synth.sha256(d) == "8b9ca080da554f5a0800e8a8c5ed42caedc593ae14fa3fff8abdd54945db4761"
    assert synth.sha256(l) == "c50a55185c5abfcc2036a1626b9fb463f74a8f635dad4bb81ceadbb568756a3a"
