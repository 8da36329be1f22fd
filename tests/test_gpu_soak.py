"""A few seconds of the soak test of the asynchronous host path (scripts/soak.py): random view and parameter changes, option
toggles, settles, throttles and read-backs; every frame read back equals the reference frame of its view bit for bit, and
nothing hangs (the cost-feedback thread once lost a wake-up: DESIGN.md section 6)."""
import importlib.util
import os

import pytest

pytestmark = pytest.mark.gpu


def test_soak_8_seconds(volym_lib):
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts", "soak.py")
    spec = importlib.util.spec_from_file_location("volym_soak", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    ops, frames, checks = mod.main(8.0, 7)
    assert ops > 1000 and frames > 1000 and checks > 20
    ops, frames, checks = mod.main(5.0, 11, 2)          # the same with two frames in flight (VOLYM_OPT_FRAMES_IN_FLIGHT = 2)
    assert ops > 500 and frames > 500 and checks > 10
