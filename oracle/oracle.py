"""ctypes binding of the CPU oracle (TEST INFRASTRUCTURE, NOT PRODUCT CODE).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this.
PARITY UNPINNED: see oracle/volym_oracle.h.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libvolym_oracle.so")

FILTER_NEAREST = 0
FILTER_LINEAR = 1


class CameraUniforms(C.Structure):
    _fields_ = [
        ("view_matrix", (C.c_float * 4) * 4),
        ("projection_matrix", (C.c_float * 4) * 4),
        ("inverse_view_proj", (C.c_float * 4) * 4),
        ("camera_position", C.c_float * 3),
        ("_padding", C.c_float),
    ]


class Parameters(C.Structure):
    _fields_ = [
        ("density_threshold", C.c_float),
        ("use_cone_importance_check", C.c_uint32),
        ("use_importance_coloring", C.c_uint32),
        ("use_opacity", C.c_uint32),
        ("use_importance_rendering", C.c_uint32),
        ("use_gaussian_smoothing", C.c_uint32),
        ("importance_check_ahead_steps", C.c_uint32),
        ("raymarching_step_size", C.c_float),
    ]


class Camera(C.Structure):
    _fields_ = [
        ("position", C.c_float * 3),
        ("target", C.c_float * 3),
        ("up", C.c_float * 3),
        ("aspect", C.c_float),
        ("fovy", C.c_float),
        ("znear", C.c_float),
        ("zfar", C.c_float),
        ("horizontal_angle", C.c_float),
        ("vertical_angle", C.c_float),
        ("distance", C.c_float),
        ("max_distance", C.c_float),
        ("min_distance", C.c_float),
    ]


class Counters(C.Structure):
    _fields_ = [
        ("n_vol", C.c_uint64),
        ("n_imp", C.c_uint64),
        ("n_steps", C.c_uint64),
        ("n_dense", C.c_uint64),
        ("n_hit", C.c_uint64),
    ]

    def as_dict(self):
        return {k: int(getattr(self, k)) for k, _ in self._fields_}


def build(force=False):
    if force or not os.path.exists(_LIB_PATH) or (
        os.path.getmtime(_LIB_PATH) < os.path.getmtime(os.path.join(_HERE, "volym_oracle.c"))
    ):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        u8p = C.POINTER(C.c_uint8)
        f32p = C.POINTER(C.c_float)
        for name in ("vo_wgsl_log2", "vo_wgsl_exp2", "vo_wgsl_exp"):
            getattr(L, name).restype = C.c_float
            getattr(L, name).argtypes = [C.c_float]
        L.vo_wgsl_pow.restype = C.c_float
        L.vo_wgsl_pow.argtypes = [C.c_float, C.c_float]
        L.vo_tf_default_lut.argtypes = [u8p]
        L.vo_tf_default_lut.restype = None
        L.vo_tf_bake.argtypes = [f32p, C.c_int, f32p, C.c_int, u8p]
        L.vo_tf_bake.restype = None
        L.vo_camera_default.argtypes = [C.POINTER(Camera), C.c_float, f32p]
        L.vo_camera_default.restype = None
        L.vo_camera_orbit.argtypes = [C.POINTER(Camera), C.c_float, C.c_float, C.c_float]
        L.vo_camera_orbit.restype = None
        L.vo_camera_uniforms_from.argtypes = [C.POINTER(Camera), C.POINTER(CameraUniforms)]
        L.vo_camera_uniforms_from.restype = C.c_int
        L.vo_prepare_volume.argtypes = [u8p, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_int, u8p]
        L.vo_prepare_volume.restype = None
        L.vo_map_segments.argtypes = [u8p, C.c_size_t, u8p, u8p, C.c_int]
        L.vo_map_segments.restype = None
        L.vo_render.argtypes = [u8p, u8p, C.c_int, C.c_int, C.c_int, C.c_int, u8p, C.c_int,
                                C.POINTER(CameraUniforms), C.POINTER(Parameters),
                                C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                f32p, u8p, C.POINTER(Counters)]
        L.vo_render.restype = C.c_int
        L.vo_render_rowlist.argtypes = [u8p, u8p, C.c_int, C.c_int, C.c_int, C.c_int, u8p, C.c_int,
                                        C.POINTER(CameraUniforms), C.POINTER(Parameters),
                                        C.c_int, C.c_int, C.POINTER(C.c_int), C.c_int, C.c_int,
                                        f32p, u8p, C.POINTER(Counters)]
        L.vo_render_rowlist.restype = C.c_int
        L.vo_render_timed.argtypes = [u8p, u8p, C.c_int, C.c_int, C.c_int, C.c_int, u8p, C.c_int,
                                      C.POINTER(CameraUniforms), C.POINTER(Parameters),
                                      C.c_int, C.c_int, C.POINTER(C.c_int), C.c_int, C.c_int,
                                      C.c_int, C.POINTER(C.c_double), u8p, C.POINTER(Counters)]
        L.vo_render_timed.restype = C.c_int
        L.vo_blit.argtypes = [u8p, C.c_int, C.c_int, u8p, C.c_int, C.c_int]
        L.vo_blit.restype = C.c_int
        L.vo_render_pixel.argtypes = [u8p, u8p, C.c_int, C.c_int, C.c_int, C.c_int, u8p, C.c_int,
                                      C.POINTER(CameraUniforms), C.POINTER(Parameters),
                                      C.c_int, C.c_int, C.c_int, C.c_int, f32p,
                                      C.POINTER(Counters)]
        L.vo_render_pixel.restype = None
        _lib = L
    return _lib


def _u8(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint8))


def _f32(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def wgsl_pow(x, y):
    return float(lib().vo_wgsl_pow(float(x), float(y)))


def tf_default_lut():
    out = np.zeros(1024, np.uint8)
    lib().vo_tf_default_lut(_u8(out))
    return out


def tf_bake(rgb_points, alpha_points):
    rgb = np.ascontiguousarray(rgb_points, np.float32).reshape(-1, 4)
    al = np.ascontiguousarray(alpha_points, np.float32).reshape(-1, 2)
    out = np.zeros(1024, np.uint8)
    lib().vo_tf_bake(_f32(rgb), rgb.shape[0], _f32(al), al.shape[0], _u8(out))
    return out


def camera_default(aspect, position=(0.5, 0.5, 0.5)):
    c = Camera()
    p = (C.c_float * 3)(*position)
    lib().vo_camera_default(C.byref(c), float(aspect), p)
    return c


def camera_orbit(cam, h=0.0, v=0.0, zoom=0.0):
    lib().vo_camera_orbit(C.byref(cam), float(h), float(v), float(zoom))
    return cam


def camera_uniforms(cam):
    u = CameraUniforms()
    rc = lib().vo_camera_uniforms_from(C.byref(cam), C.byref(u))
    if rc != 0:
        raise ValueError("inverse_view_proj inversion failed")
    return u


def benchmark_camera_uniforms(aspect, h_deg=0.0, v_deg=0.0, zoom=0.0):
    """Pose the reference's frame loop ends up with (src/event_loop.rs:100 ->
    src/camera.rs:47-61): orbit about (.5,.5,.5) at distance 1 => eye (.5,.5,1.5)."""
    cam = camera_default(aspect, (0.5, 0.5, 3.5))
    camera_orbit(cam, h_deg, v_deg, zoom)
    return camera_uniforms(cam)


def prepare_volume(raw, dims, flip_y=True):
    nx, ny, nz = dims
    raw = np.ascontiguousarray(raw, np.uint8).ravel()
    out = np.empty(nx * ny * nz, np.uint8)
    lib().vo_prepare_volume(_u8(raw), raw.size, nx, ny, nz, 1 if flip_y else 0, _u8(out))
    return out


def map_segments(labels, segments):
    data = np.array(labels, np.uint8, copy=True).ravel()
    lv = np.array([s["label_value"] for s in segments], np.uint8)
    im = np.array([s["importance"] for s in segments], np.uint8)
    lib().vo_map_segments(_u8(data), data.size, _u8(lv), _u8(im), len(segments))
    return data


def make_parameters(density_threshold=0.15, use_cone_importance_check=0, use_importance_coloring=0,
                    use_opacity=1, use_importance_rendering=0, use_gaussian_smoothing=0,
                    importance_check_ahead_steps=15, raymarching_step_size=0.01):
    return Parameters(density_threshold, use_cone_importance_check, use_importance_coloring,
                      use_opacity, use_importance_rendering, use_gaussian_smoothing,
                      importance_check_ahead_steps, raymarching_step_size)


def render(volume, importances, dims, lut, cam_uniforms, params, W, H, filter=FILTER_NEAREST,
           threads=None, rows=None, want_f32=True, want_u8=True, rowlist=None):
    """Returns (rgba_f32 [H,W,4] or None, rgba_u8 [H,W,4] or None, counters dict).
    rows = (y0, y1) renders that range, rowlist = [y, ...] those rows; other rows stay zero."""
    nx, ny, nz = dims
    volume = np.ascontiguousarray(volume, np.uint8).ravel()
    importances = np.ascontiguousarray(importances, np.uint8).ravel()
    assert volume.size == nx * ny * nz and importances.size == nx * ny * nz
    lut = np.ascontiguousarray(lut, np.uint8).ravel()
    f32 = np.zeros((H, W, 4), np.float32) if want_f32 else None
    u8 = np.zeros((H, W, 4), np.uint8) if want_u8 else None
    k = Counters()
    y0, y1 = rows if rows is not None else (0, H)
    if threads is None:
        threads = os.cpu_count() or 1
    if rowlist is not None:
        rl = np.ascontiguousarray(rowlist, np.int32)
        rc = lib().vo_render_rowlist(_u8(volume), _u8(importances), nx, ny, nz, int(filter), _u8(lut),
                                     lut.size // 4, C.byref(cam_uniforms), C.byref(params), W, H,
                                     rl.ctypes.data_as(C.POINTER(C.c_int)), int(rl.size),
                                     int(threads), _f32(f32) if want_f32 else None,
                                     _u8(u8) if want_u8 else None, C.byref(k))
    else:
        rc = lib().vo_render(_u8(volume), _u8(importances), nx, ny, nz, int(filter), _u8(lut),
                             lut.size // 4, C.byref(cam_uniforms), C.byref(params), W, H, y0, y1,
                             int(threads), _f32(f32) if want_f32 else None,
                             _u8(u8) if want_u8 else None, C.byref(k))
    if rc != 0:
        raise ValueError("vo_render rejected its arguments")
    return f32, u8, k.as_dict()


def render_timed(volume, importances, dims, lut, cam_uniforms, params, W, H, passes, filter=FILTER_NEAREST,
                 threads=None, rowlist=None):
    """CPU-baseline timing: one persistent pool of `threads` workers renders the frame (or `rowlist`) `passes`
    times.  Returns (seconds per pass [passes], counters of one pass)."""
    nx, ny, nz = dims
    volume = np.ascontiguousarray(volume, np.uint8).ravel()
    importances = np.ascontiguousarray(importances, np.uint8).ravel()
    lut = np.ascontiguousarray(lut, np.uint8).ravel()
    if threads is None:
        threads = os.cpu_count() or 1
    secs = np.zeros(int(passes), np.float64)
    k = Counters()
    rl = np.ascontiguousarray(rowlist, np.int32) if rowlist is not None else None
    rc = lib().vo_render_timed(_u8(volume), _u8(importances), nx, ny, nz, int(filter), _u8(lut), lut.size // 4,
                               C.byref(cam_uniforms), C.byref(params), W, H,
                               rl.ctypes.data_as(C.POINTER(C.c_int)) if rl is not None else None,
                               int(rl.size) if rl is not None else 0, int(threads), int(passes),
                               secs.ctypes.data_as(C.POINTER(C.c_double)), None, C.byref(k))
    if rc != 0:
        raise ValueError("vo_render_timed failed (%d)" % rc)
    return secs, k.as_dict()


def blit(frame_rgba8, out_w, out_h):
    """shaders/render.wgsl:39-43: frame [H, W, 4] uint8 -> target [out_h, out_w, 4] uint8."""
    f = np.ascontiguousarray(frame_rgba8, np.uint8)
    H, W, _ = f.shape
    out = np.empty((int(out_h), int(out_w), 4), np.uint8)
    if lib().vo_blit(_u8(f), W, H, _u8(out), int(out_w), int(out_h)) != 0:
        raise ValueError("vo_blit rejected its arguments")
    return out
