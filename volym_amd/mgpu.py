"""ctypes face of the native multi-GPU frame loop (include/volym_mgpu.h): screen-tile sharding over the GPUs of one node,
packed shards gathered onto the root over RCCL (or device copies), assembled there; N frames per call, no Python per frame."""
import ctypes as C

import numpy as np

from . import _lib, scene

RCCL, COPY = 0, 1


class Timing(C.Structure):
    _fields_ = [("wall_ms", C.c_double), ("enqueue_us_per_frame", C.c_double), ("frames", C.c_uint32), ("graph_replays", C.c_uint32),
                ("msg_bytes", C.c_uint32), ("overflowed", C.c_uint32)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


class Split(C.Structure):
    _fields_ = [(n, C.c_float) for n in ("march_ms", "pack_ms", "collective_ms", "assemble_ms")]

    def as_dict(self):
        return {k: float(getattr(self, k)) for k, _ in self._fields_}


_mg = C.c_void_p
_u8p = C.POINTER(C.c_uint8)
SIGNATURES = {
    "volym_mgpu_unique_id": (C.c_int, [_u8p]),
    "volym_mgpu_create": (C.c_int, [C.POINTER(_mg), C.c_uint32, C.c_uint32, C.c_int, C.POINTER(C.c_int), C.c_int]),
    "volym_mgpu_create_rank": (C.c_int, [C.POINTER(_mg), C.c_uint32, C.c_uint32, C.c_int, C.c_int, C.c_int, _u8p]),
    "volym_mgpu_destroy": (None, [_mg]),
    "volym_mgpu_last_error": (C.c_char_p, [_mg]),
    "volym_mgpu_world": (C.c_int, [_mg]),
    "volym_mgpu_local_count": (C.c_int, [_mg]),
    "volym_mgpu_context": (C.c_void_p, [_mg, C.c_int]),
    "volym_mgpu_local_rank": (C.c_int, [_mg, C.c_int]),
    "volym_mgpu_set_volume": (C.c_int, [_mg, _u8p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int]),
    "volym_mgpu_set_importances": (C.c_int, [_mg, _u8p, C.c_uint32, C.c_uint32, C.c_uint32]),
    "volym_mgpu_set_transfer_function": (C.c_int, [_mg, _u8p, C.c_uint32]),
    "volym_mgpu_set_option": (C.c_int, [_mg, C.c_int, C.c_int]),
    "volym_mgpu_update": (C.c_int, [_mg, C.POINTER(_lib.CameraUniforms), C.POINTER(_lib.ParameterUniforms)]),
    "volym_mgpu_prepare": (C.c_int, [_mg, C.c_uint32]),
    "volym_mgpu_run": (C.c_int, [_mg, C.c_uint32, C.c_int, C.POINTER(Timing)]),
    "volym_mgpu_profile": (C.c_int, [_mg, C.c_uint32, C.POINTER(Split)]),
    "volym_mgpu_read_rgba8": (C.c_int, [_mg, _u8p]),
}
_bound = False


def lib():
    global _bound
    L = _lib.lib()
    if not _bound:
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _bound = True
    return L


def unique_id():
    """128-byte RCCL id: rank 0 creates it, the other ranks receive it by the launcher's means (bench.py: torch.distributed)."""
    buf = np.zeros(128, np.uint8)
    rc = lib().volym_mgpu_unique_id(scene._u8p(buf))
    if rc != _lib.OK:
        raise _lib.VolymError(rc, (lib().volym_mgpu_last_error(None) or b"").decode())
    return bytes(buf)


class MultiGpu:
    def __init__(self, width, height, devices=None, transport=RCCL, rank=None, world=None, device_id=0, uid=None):
        """devices=[...]: one process, those devices (COPY transport: a device may repeat -- virtual ranks).
        rank/world/device_id/uid: one process per device."""
        self.width, self.height = int(width), int(height)
        self._h = C.c_void_p()
        if rank is None:
            ids = (C.c_int * len(devices))(*devices)
            rc = lib().volym_mgpu_create(C.byref(self._h), self.width, self.height, len(devices), ids, int(transport))
        else:
            ub = np.frombuffer(uid, np.uint8).copy() if uid is not None else np.zeros(128, np.uint8)
            rc = lib().volym_mgpu_create_rank(C.byref(self._h), self.width, self.height, int(device_id), int(rank), int(world), scene._u8p(ub))
        if rc != _lib.OK:
            raise _lib.VolymError(rc, (lib().volym_mgpu_last_error(None) or b"").decode())

    def _ck(self, rc):
        if rc != _lib.OK:
            raise _lib.VolymError(rc, (lib().volym_mgpu_last_error(self._h) or b"").decode())

    def close(self):
        if self._h:
            lib().volym_mgpu_destroy(self._h)
            self._h = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def world(self):
        return int(lib().volym_mgpu_world(self._h))

    @property
    def local_count(self):
        return int(lib().volym_mgpu_local_count(self._h))

    def local_rank(self, i):
        return int(lib().volym_mgpu_local_rank(self._h, int(i)))

    def context_handle(self, i):
        return C.c_void_p(lib().volym_mgpu_context(self._h, int(i)))

    def set_volume(self, voxels, dims, filter=_lib.FILTER_NEAREST):
        v = np.ascontiguousarray(voxels, np.uint8).ravel()
        self._ck(lib().volym_mgpu_set_volume(self._h, scene._u8p(v), dims[0], dims[1], dims[2], int(filter)))

    def set_importances(self, importances, dims):
        v = np.ascontiguousarray(importances, np.uint8).ravel()
        self._ck(lib().volym_mgpu_set_importances(self._h, scene._u8p(v), dims[0], dims[1], dims[2]))

    def set_transfer_function(self, rgba8):
        t = np.ascontiguousarray(rgba8, np.uint8).ravel()
        self._ck(lib().volym_mgpu_set_transfer_function(self._h, scene._u8p(t), t.size // 4))

    def set_option(self, key, value):
        self._ck(lib().volym_mgpu_set_option(self._h, int(key), int(value)))

    def update(self, camera_uniforms, parameter_uniforms):
        self._ck(lib().volym_mgpu_update(self._h, C.byref(camera_uniforms), C.byref(parameter_uniforms)))

    def prepare(self, slack_percent=0):
        self._ck(lib().volym_mgpu_prepare(self._h, int(slack_percent)))

    def run(self, frames, use_graph=True):
        t = Timing()
        self._ck(lib().volym_mgpu_run(self._h, int(frames), 1 if use_graph else 0, C.byref(t)))
        return t.as_dict()

    def profile(self, frames=8):
        s = Split()
        self._ck(lib().volym_mgpu_profile(self._h, int(frames), C.byref(s)))
        return s.as_dict()

    def stats_pass(self, i=0):
        s = _lib.Stats()
        rc = _lib.lib().volym_stats_pass(self.context_handle(i), C.byref(s))
        if rc != _lib.OK:
            raise _lib.VolymError(rc, (_lib.lib().volym_last_error(self.context_handle(i)) or b"").decode())
        return s.as_dict()

    def read_rgba8(self):
        out = np.empty((self.height, self.width, 4), np.uint8)
        self._ck(lib().volym_mgpu_read_rgba8(self._h, scene._u8p(out)))
        return out
