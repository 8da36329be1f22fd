"""Host scene model: thin Python faces over the C++ shim in libvolym_hip.so
(volym_amd/csrc/scene.cpp, include/volym_host.h).  Names follow the reference
(src/camera.rs, src/state.rs, src/transfer_function.rs); no math is done in Python.
"""
import ctypes as C
import json

import numpy as np

from . import _lib
from ._lib import CameraUniforms, ParameterUniforms  # noqa: F401  (re-export)


def _f32p(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _u8p(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint8))


class Camera:
    """src/camera.rs:5-73"""

    def __init__(self, c=None):
        self.c = c if c is not None else _lib.CCamera()

    @staticmethod
    def default_with_aspect_and_pos(aspect, position):
        cam = Camera()
        pos = (C.c_float * 3)(*position)
        _lib.lib().volym_camera_default_with_aspect_and_pos(C.byref(cam.c), float(aspect), pos)
        return cam

    def orbit(self, horizontal_delta, vertical_delta, zoom_delta):
        _lib.lib().volym_camera_orbit(C.byref(self.c), float(horizontal_delta), float(vertical_delta),
                                      float(zoom_delta))

    def view_matrix(self):
        m = np.zeros((4, 4), np.float32)
        _lib.lib().volym_camera_view_matrix(C.byref(self.c), _f32p(m))
        return m

    def projection_matrix(self):
        m = np.zeros((4, 4), np.float32)
        _lib.lib().volym_camera_projection_matrix(C.byref(self.c), _f32p(m))
        return m

    def uniforms(self):
        """CameraUniforms::try_from(&Camera) (src/gpu_resources/camera.rs:66-85)"""
        u = CameraUniforms()
        rc = _lib.lib().volym_camera_uniforms_from(C.byref(self.c), C.byref(u))
        if rc != _lib.OK:
            raise _lib.VolymError(rc, "inverse_view_proj inversion failed")
        return u

    @property
    def position(self):
        return tuple(self.c.position)


class StateParameters:
    """src/state.rs:28-55; `benchmark()` = src/main.rs:180-190"""

    def __init__(self, c=None):
        if c is None:
            c = _lib.CStateParameters()
            _lib.lib().volym_state_parameters_default(C.byref(c))
        self.c = c

    @staticmethod
    def benchmark():
        c = _lib.CStateParameters()
        _lib.lib().volym_state_parameters_benchmark(C.byref(c))
        return StateParameters(c)

    def replace(self, **kw):
        c = _lib.CStateParameters.from_buffer_copy(bytes(self.c))
        for k, v in kw.items():
            if k == "camera_position":
                c.camera_position = (C.c_float * 3)(*v)
            else:
                if not hasattr(c, k):
                    raise AttributeError(k)
                setattr(c, k, v)
        return StateParameters(c)


class State:
    """src/state.rs:11-76, :153-155 (parameter half; window input is out of scope)"""

    def __init__(self, c):
        self.c = c

    @staticmethod
    def with_parameters(aspect, parameters):
        c = _lib.CState()
        _lib.lib().volym_state_with_parameters(C.byref(c), float(aspect), C.byref(parameters.c))
        return State(c)

    def update(self):
        _lib.lib().volym_state_update(C.byref(self.c))

    def process_mouse(self, dx, dy):
        _lib.lib().volym_camera_controller_process_mouse(C.byref(self.c.camera_controller), float(dx), float(dy))

    def process_scroll(self, line_delta):
        _lib.lib().volym_camera_controller_process_scroll(C.byref(self.c.camera_controller), float(line_delta))

    @property
    def camera(self):
        return Camera(self.c.camera)

    def camera_uniforms(self):
        return self.camera.uniforms()

    def parameter_uniforms(self):
        """ParameterUniforms::try_from(&State) (src/gpu_resources/parameters.rs:68-83)"""
        u = ParameterUniforms()
        _lib.check(_lib.lib().volym_parameter_uniforms_from(C.byref(self.c), C.byref(u)))
        return u


class TransferFunction:
    """src/transfer_function.rs; baked as GPUTransferFunction::new_texture_1d_rgbt does."""

    def __init__(self, rgb_points=None, alpha_points=None):
        self.rgb_points = [] if rgb_points is None else list(rgb_points)      # (iso, r, g, b)
        self.alpha_points = [] if alpha_points is None else list(alpha_points)  # (iso, a)

    @staticmethod
    def default():
        """impl Default for TransferFunction (src/transfer_function.rs:19-56)"""
        return TransferFunction(
            [(0.0, 0, 1, 0), (0.2, 0, 1, 1), (0.4, 1, 1, 0), (0.6, 1, 0, 1), (1.0, 1, 0, 0)],
            [(0.0, 0.0), (1.0, 1.0)])

    def add_rgb_control_point(self, iso, r, g, b):
        self.rgb_points.append((iso, r, g, b))

    def add_alpha_control_point(self, iso, a):
        self.alpha_points.append((iso, a))

    def bake_rgba8(self):
        rgb = np.ascontiguousarray(self.rgb_points, np.float32).reshape(-1, 4)
        al = np.ascontiguousarray(self.alpha_points, np.float32).reshape(-1, 2)
        out = np.zeros(1024, np.uint8)
        _lib.check(_lib.lib().volym_transfer_function_bake(_f32p(rgb), rgb.shape[0], _f32p(al), al.shape[0],
                                                           _u8p(out)))
        return out


def default_lut():
    out = np.zeros(1024, np.uint8)
    _lib.lib().volym_transfer_function_default_lut(_u8p(out))
    return out


def prepare_volume(raw, dims, flip_y=True):
    """GpuVolume::init's byte path (src/gpu_resources/volume.rs:38-61): pad/truncate, FlipMode::Y."""
    nx, ny, nz = dims
    raw = np.ascontiguousarray(raw, np.uint8).ravel()
    out = np.empty(nx * ny * nz, np.uint8)
    _lib.check(_lib.lib().volym_prepare_volume(_u8p(raw), raw.size, nx, ny, nz, 1 if flip_y else 0, _u8p(out)))
    return out


def load_segments(path_or_list):
    """Vec<SegmentInfo> (src/demos/simple/importance.rs:13-20) from the JSON the reference ships."""
    if isinstance(path_or_list, (list, tuple)):
        segs = list(path_or_list)
    else:
        with open(path_or_list) as f:
            segs = json.load(f)
    for s in segs:
        for k in ("label_value", "importance"):
            if not (isinstance(s[k], int) and 0 <= s[k] <= 255):
                raise ValueError("segment field %s must be a u8" % k)
    return segs


def map_segments_to_importance(labels, segments):
    """src/demos/simple/importance.rs:148-158"""
    data = np.array(labels, np.uint8, copy=True).ravel()
    lv = np.array([s["label_value"] for s in segments], np.uint8)
    im = np.array([s["importance"] for s in segments], np.uint8)
    _lib.check(_lib.lib().volym_map_segments_to_importance(_u8p(data), data.size, _u8p(lv), _u8p(im), len(segments)))
    return data
