"""ctypes binding of libvolym_hip.so (include/volym_hip.h + include/volym_host.h).

There is no CPU fallback: if the shared library is missing or a call fails, this raises.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("VOLYM_HIP_LIB") or os.path.join(_HERE, "libvolym_hip.so")   # the override is for A/B builds during development

OK, E_INVALID, E_HIP, E_NO_DEVICE, E_NOMEM, E_STATE = 0, -1, -2, -3, -4, -5
FILTER_NEAREST, FILTER_LINEAR = 0, 1
OPT_KERNEL, OPT_WRITE_F32, OPT_MACRO_CELLS = 1, 2, 3
OPT_VOLUME_LAYOUT, OPT_CULLING, OPT_COST_FEEDBACK, OPT_DEPTH_PARALLEL, OPT_XCD_BANDS, OPT_REBALANCE_ROUNDS = 4, 5, 6, 7, 8, 9
OPT_SETUP_IEEE = 10
OPT_FRAMES_IN_FLIGHT = 11


class VolymError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("volym error %d: %s" % (code, msg))
        self.code = code


class CameraUniforms(C.Structure):
    """src/gpu_resources/camera.rs:56-64"""
    _fields_ = [
        ("view_matrix", (C.c_float * 4) * 4),
        ("projection_matrix", (C.c_float * 4) * 4),
        ("inverse_view_proj", (C.c_float * 4) * 4),
        ("camera_position", C.c_float * 3),
        ("_padding", C.c_float),
    ]


class ParameterUniforms(C.Structure):
    """src/gpu_resources/parameters.rs:55-66"""
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


class Stats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("n_vol", "n_imp", "n_steps", "n_dense", "n_hit", "n_rays")]

    def as_dict(self):
        return {k: int(getattr(self, k)) for k, _ in self._fields_}


class CCamera(C.Structure):
    """src/camera.rs:5-19"""
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


class CCameraController(C.Structure):
    """src/camera.rs:76-83"""
    _fields_ = [(n, C.c_float) for n in
                ("rotate_horizontal", "rotate_vertical", "scroll", "sensitivity", "zoom_sensitivity")]


class CStateParameters(C.Structure):
    """src/state.rs:28-39"""
    _fields_ = [
        ("camera_position", C.c_float * 3),
        ("density_trheshold", C.c_float),
        ("use_cone_importance_check", C.c_uint32),
        ("use_importance_coloring", C.c_uint32),
        ("use_opacity", C.c_uint32),
        ("use_importance_rendering", C.c_uint32),
        ("use_gaussian_smoothing", C.c_uint32),
        ("importance_check_ahead_steps", C.c_uint32),
        ("raymarching_step_size", C.c_float),
    ]


class CState(C.Structure):
    """src/state.rs:11-26 (parameter half)"""
    _fields_ = [
        ("camera", CCamera),
        ("camera_controller", CCameraController),
        ("density_threshold", C.c_float),
        ("use_importance_coloring", C.c_uint32),
        ("use_cone_importance_check", C.c_uint32),
        ("use_opacity", C.c_uint32),
        ("use_importance_rendering", C.c_uint32),
        ("use_gaussian_smoothing", C.c_uint32),
        ("importance_check_ahead_steps", C.c_uint32),
        ("raymarching_step_size", C.c_float),
    ]


_u8p = C.POINTER(C.c_uint8)
_f32p = C.POINTER(C.c_float)
_ctx = C.c_void_p

# name -> (restype, argtypes): every symbol the two headers declare
SIGNATURES = {
    # include/volym_hip.h
    "volym_create": (C.c_int, [C.POINTER(_ctx), C.c_uint32, C.c_uint32, C.c_int]),
    "volym_destroy": (None, [_ctx]),
    "volym_last_error": (C.c_char_p, [_ctx]),
    "volym_abi_version": (C.c_int, []),
    "volym_set_stream": (C.c_int, [_ctx, C.c_void_p]),
    "volym_set_option": (C.c_int, [_ctx, C.c_int, C.c_int]),
    "volym_set_shard": (C.c_int, [_ctx, C.c_uint32, C.c_uint32]),
    "volym_set_volume": (C.c_int, [_ctx, _u8p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int]),
    "volym_set_importances": (C.c_int, [_ctx, _u8p, C.c_uint32, C.c_uint32, C.c_uint32]),
    "volym_set_transfer_function": (C.c_int, [_ctx, _u8p, C.c_uint32]),
    "volym_update": (C.c_int, [_ctx, C.POINTER(CameraUniforms), C.POINTER(ParameterUniforms)]),
    "volym_compute_pass": (C.c_int, [_ctx]),
    "volym_sync": (C.c_int, [_ctx]),
    "volym_settle": (C.c_int, [_ctx]),
    "volym_throttle": (C.c_int, [_ctx, C.c_uint32]),
    "volym_blit": (C.c_int, [_ctx, C.c_void_p, C.c_uint32, C.c_uint32]),
    "volym_read_blit": (C.c_int, [_ctx, _u8p]),
    "volym_read_rgba8": (C.c_int, [_ctx, _u8p]),
    "volym_read_rgba32f": (C.c_int, [_ctx, _f32p]),
    "volym_local_tiles": (C.c_uint32, [_ctx]),
    "volym_shard_bytes": (C.c_size_t, [_ctx]),
    "volym_shard_device_ptr": (C.c_void_p, [_ctx]),
    "volym_frame_device_ptr": (C.c_void_p, [_ctx]),
    "volym_bind_output": (C.c_int, [_ctx, C.c_void_p, C.c_void_p]),
    "volym_assemble": (C.c_int, [_ctx, C.c_void_p]),
    "volym_packed_shard_bytes": (C.c_size_t, [_ctx, C.c_uint32]),
    "volym_pack_shard": (C.c_int, [_ctx, C.c_void_p, C.c_size_t]),
    "volym_packed_tiles": (C.c_int, [_ctx, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    "volym_assemble_packed": (C.c_int, [_ctx, C.c_void_p, C.c_size_t]),
    "volym_read_shard": (C.c_int, [_ctx, _u8p]),
    "volym_assemble_host": (C.c_int, [_ctx, _u8p]),
    "volym_stats_pass": (C.c_int, [_ctx, C.POINTER(Stats)]),
    "volym_time_passes": (C.c_int, [_ctx, C.c_uint32, _f32p]),
    "volym_time_batch": (C.c_int, [_ctx, C.c_uint32, _f32p]),
    "volym_selftest_ray_setup": (C.c_int, [_ctx, C.POINTER(C.c_ulonglong)]),
    # include/volym_host.h
    "volym_camera_default_with_aspect_and_pos": (None, [C.POINTER(CCamera), C.c_float, _f32p]),
    "volym_camera_orbit": (None, [C.POINTER(CCamera), C.c_float, C.c_float, C.c_float]),
    "volym_camera_view_matrix": (None, [C.POINTER(CCamera), _f32p]),
    "volym_camera_projection_matrix": (None, [C.POINTER(CCamera), _f32p]),
    "volym_camera_uniforms_from": (C.c_int, [C.POINTER(CCamera), C.POINTER(CameraUniforms)]),
    "volym_camera_controller_new": (None, [C.POINTER(CCameraController), C.c_float, C.c_float]),
    "volym_camera_controller_process_mouse": (None, [C.POINTER(CCameraController), C.c_double, C.c_double]),
    "volym_camera_controller_process_scroll": (None, [C.POINTER(CCameraController), C.c_float]),
    "volym_camera_controller_update_camera": (None, [C.POINTER(CCameraController), C.POINTER(CCamera)]),
    "volym_state_parameters_default": (None, [C.POINTER(CStateParameters)]),
    "volym_state_parameters_benchmark": (None, [C.POINTER(CStateParameters)]),
    "volym_state_with_parameters": (None, [C.POINTER(CState), C.c_float, C.POINTER(CStateParameters)]),
    "volym_state_update": (None, [C.POINTER(CState)]),
    "volym_parameter_uniforms_from": (C.c_int, [C.POINTER(CState), C.POINTER(ParameterUniforms)]),
    "volym_transfer_function_default_lut": (None, [_u8p]),
    "volym_transfer_function_bake": (C.c_int, [_f32p, C.c_uint32, _f32p, C.c_uint32, _u8p]),
    "volym_prepare_volume": (C.c_int, [_u8p, C.c_size_t, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, _u8p]),
    "volym_map_segments_to_importance": (C.c_int, [_u8p, C.c_size_t, _u8p, _u8p, C.c_uint32]),
    "volym_synth_bonsai": (C.c_int, [C.c_uint32, C.c_uint32, _u8p, _u8p]),
    "volym_synth_teapot": (C.c_int, [C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, _u8p, _u8p]),
}

_lib = None


def lib():
    """Load libvolym_hip.so.  Fails loudly when it has not been built (no fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                "%s is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "or `make -C volym_amd/csrc` (hipcc --offload-arch=gfx950)" % LIB_PATH)
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)       # AttributeError if the .so lacks a declared symbol
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(rc, ctx=None):
    if rc != OK:
        msg = lib().volym_last_error(ctx)
        raise VolymError(rc, msg.decode() if msg else "")
    return rc
