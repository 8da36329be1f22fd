"""Python mirror of the reference's compute-plugin boundary over the C ABI.

    trait ComputeDemo { init(ctx, state, output_texture); update_gpu_state(ctx, state);
                        compute_pass(ctx) }              -- src/demos/mod.rs:9-17
    struct Simple                                        -- src/demos/simple/mod.rs:26-121

`GpuContext` stands where gpu_context.rs + GpuWriteTexture2D stood: a device, a stream
and a W x H rgba8 output.  All compute happens in libvolym_hip.so (HIP, gfx950).
"""
import ctypes as C

import numpy as np

from . import _lib, scene, synth


class GpuContext:
    """src/gpu_context.rs:20-62 + src/gpu_resources/texture.rs:40-59, headless."""

    def __init__(self, width, height, device_id=-1):
        self.width, self.height = int(width), int(height)
        self._h = C.c_void_p()
        rc = _lib.lib().volym_create(C.byref(self._h), self.width, self.height, int(device_id))
        if rc != _lib.OK:
            raise _lib.VolymError(rc, (_lib.lib().volym_last_error(None) or b"").decode())

    @classmethod
    def borrow(cls, handle, width, height):
        """A view of a context somebody else owns (the native multi-GPU loop's): close() leaves it alone."""
        self = cls.__new__(cls)
        self.width, self.height = int(width), int(height)
        self._h = handle
        self._borrowed = True
        return self

    @property
    def handle(self):
        if not self._h:
            raise RuntimeError("GpuContext is closed")
        return self._h

    def _ck(self, rc):
        if rc != _lib.OK:
            raise _lib.VolymError(rc, (_lib.lib().volym_last_error(self._h) or b"").decode())

    def close(self):
        if self._h:
            if not getattr(self, "_borrowed", False):
                _lib.lib().volym_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    # ---- options / sharding -----------------------------------------------------------------
    def set_option(self, key, value):
        self._ck(_lib.lib().volym_set_option(self.handle, int(key), int(value)))

    def set_stream(self, hip_stream):
        self._ck(_lib.lib().volym_set_stream(self.handle, C.c_void_p(hip_stream)))

    def set_shard(self, rank, world):
        self._ck(_lib.lib().volym_set_shard(self.handle, int(rank), int(world)))

    def bind_output(self, shard_ptr, frame_ptr):
        self._ck(_lib.lib().volym_bind_output(self.handle, C.c_void_p(shard_ptr), C.c_void_p(frame_ptr)))

    # ---- resources --------------------------------------------------------------------------
    def set_volume(self, voxels, dims, filter=_lib.FILTER_NEAREST):
        v = np.ascontiguousarray(voxels, np.uint8).ravel()
        nx, ny, nz = dims
        if v.size != nx * ny * nz:
            raise ValueError("volume has %d bytes, dims say %d" % (v.size, nx * ny * nz))
        self._ck(_lib.lib().volym_set_volume(self.handle, scene._u8p(v), nx, ny, nz, int(filter)))

    def set_importances(self, importances, dims):
        v = np.ascontiguousarray(importances, np.uint8).ravel()
        nx, ny, nz = dims
        if v.size != nx * ny * nz:
            raise ValueError("importances have %d bytes, dims say %d" % (v.size, nx * ny * nz))
        self._ck(_lib.lib().volym_set_importances(self.handle, scene._u8p(v), nx, ny, nz))

    def set_transfer_function(self, rgba8):
        t = np.ascontiguousarray(rgba8, np.uint8).ravel()
        self._ck(_lib.lib().volym_set_transfer_function(self.handle, scene._u8p(t), t.size // 4))

    # ---- per frame --------------------------------------------------------------------------
    def update(self, camera_uniforms, parameter_uniforms):
        self._ck(_lib.lib().volym_update(self.handle, C.byref(camera_uniforms), C.byref(parameter_uniforms)))

    def compute_pass(self):
        self._ck(_lib.lib().volym_compute_pass(self.handle))

    def sync(self):
        self._ck(_lib.lib().volym_sync(self.handle))

    def throttle(self, max_in_flight=3):
        """Frame-loop back-pressure (the swap chain's role, src/event_loop.rs:114): at most `max_in_flight` frames ahead."""
        self._ck(_lib.lib().volym_throttle(self.handle, int(max_in_flight)))

    def settle(self):
        """Wait until a re-deal of the work lists in flight (cost feedback) has been adopted."""
        self._ck(_lib.lib().volym_settle(self.handle))

    def blit(self, out_w, out_h, target_ptr=None):
        """RenderPipeline::render_pass (src/render_pipeline.rs:88-130): frame -> out_w x out_h target."""
        self._blit_size = (int(out_h), int(out_w))
        self._ck(_lib.lib().volym_blit(self.handle, C.c_void_p(target_ptr), int(out_w), int(out_h)))

    def read_blit(self):
        h, w = self._blit_size
        out = np.empty((h, w, 4), np.uint8)
        self._ck(_lib.lib().volym_read_blit(self.handle, scene._u8p(out)))
        return out

    # ---- output -----------------------------------------------------------------------------
    def read_rgba8(self):
        out = np.empty((self.height, self.width, 4), np.uint8)
        self._ck(_lib.lib().volym_read_rgba8(self.handle, scene._u8p(out)))
        return out

    def read_rgba32f(self):
        out = np.empty((self.height, self.width, 4), np.float32)
        self._ck(_lib.lib().volym_read_rgba32f(self.handle, scene._f32p(out)))
        return out

    def local_tiles(self):
        return int(_lib.lib().volym_local_tiles(self.handle))

    def shard_bytes(self):
        return int(_lib.lib().volym_shard_bytes(self.handle))

    def shard_device_ptr(self):
        return _lib.lib().volym_shard_device_ptr(self.handle)

    def frame_device_ptr(self):
        return _lib.lib().volym_frame_device_ptr(self.handle)

    def read_shard(self):
        out = np.empty(self.shard_bytes(), np.uint8)
        self._ck(_lib.lib().volym_read_shard(self.handle, scene._u8p(out)))
        return out

    def assemble(self, gathered_device_ptr):
        self._ck(_lib.lib().volym_assemble(self.handle, C.c_void_p(gathered_device_ptr)))

    def packed_shard_bytes(self, tiles):
        return int(_lib.lib().volym_packed_shard_bytes(self.handle, int(tiles)))

    def pack_shard(self, packed_device_ptr, capacity_bytes):
        self._ck(_lib.lib().volym_pack_shard(self.handle, C.c_void_p(packed_device_ptr), int(capacity_bytes)))

    def packed_tiles(self):
        """(tiles stored by the last pack, overflow flag); synchronises."""
        used, over = C.c_uint32(0), C.c_uint32(0)
        self._ck(_lib.lib().volym_packed_tiles(self.handle, C.byref(used), C.byref(over)))
        return int(used.value), int(over.value)

    def assemble_packed(self, gathered_device_ptr, stride_bytes):
        self._ck(_lib.lib().volym_assemble_packed(self.handle, C.c_void_p(gathered_device_ptr), int(stride_bytes)))

    def assemble_host(self, gathered):
        g = np.ascontiguousarray(gathered, np.uint8).ravel()
        self._ck(_lib.lib().volym_assemble_host(self.handle, scene._u8p(g)))

    # ---- measurement ------------------------------------------------------------------------
    def stats_pass(self):
        s = _lib.Stats()
        self._ck(_lib.lib().volym_stats_pass(self.handle, C.byref(s)))
        return s.as_dict()

    def selftest_ray_setup(self):
        """(rays whose shared-reciprocal set-up differs from plain divisions, rays of waves that fell back, rays)."""
        out = (C.c_ulonglong * 3)()
        self._ck(_lib.lib().volym_selftest_ray_setup(self.handle, out))
        return int(out[0]), int(out[1]), int(out[2])

    def time_batch(self, n):
        """n back-to-back passes between one pair of HIP events: total milliseconds."""
        ms = C.c_float(0.0)
        self._ck(_lib.lib().volym_time_batch(self.handle, int(n), C.byref(ms)))
        return float(ms.value)

    def time_passes(self, n):
        ms = np.zeros(int(n), np.float32)
        self._ck(_lib.lib().volym_time_passes(self.handle, int(n), scene._f32p(ms)))
        return ms


class ComputeDemo:
    """src/demos/mod.rs:9-17"""

    @classmethod
    def init(cls, ctx, state, **kw):
        raise NotImplementedError

    def update_gpu_state(self, ctx, state):
        raise NotImplementedError

    def compute_pass(self, ctx):
        raise NotImplementedError


class Simple(ComputeDemo):
    """src/demos/simple/mod.rs:35-121.  The reference hard-codes its asset paths
    (:40-55); here the caller passes raw bytes (a real .raw read from disk, or the
    synthetic stand-ins of volym_amd.synth) and the segments table."""

    def __init__(self, dims):
        self.dims = dims

    @classmethod
    def init(cls, ctx, state, volume_raw=None, labels_raw=None, segments=None, dims=(256, 256, 256),
             filter=_lib.FILTER_NEAREST, transfer_function=None):
        if volume_raw is None:   # the reference's default asset, synthesised (.MISSING_LARGE_BLOBS)
            volume_raw, labels_raw = synth.synth_teapot()
            segments = synth.TEAPOT_SEGMENTS
        if labels_raw is None:
            labels_raw = np.zeros(0, np.uint8)
        segments = scene.load_segments(segments if segments is not None else [])
        # GpuVolume::init (src/gpu_resources/volume.rs:35-101)
        volume = scene.prepare_volume(volume_raw, dims, flip_y=True)
        ctx.set_volume(volume, dims, filter)
        # GpuImportances::init (src/demos/simple/importance.rs:45-137): map, then pad/flip
        importances = scene.prepare_volume(scene.map_segments_to_importance(labels_raw, segments), dims, flip_y=True)
        ctx.set_importances(importances, dims)
        # TransferFunction::default() + bake (src/demos/simple/mod.rs:64-66)
        tf = transfer_function if transfer_function is not None else scene.TransferFunction.default()
        ctx.set_transfer_function(tf.bake_rgba8())
        self = cls(dims)
        self.update_gpu_state(ctx, state)   # GpuCamera::new / GpuParameters::new upload initial state
        return self

    def update_gpu_state(self, ctx, state):
        """BaseDemo::update_gpu_state (src/demos/pipeline.rs:208-212)"""
        ctx.update(state.camera_uniforms(), state.parameter_uniforms())

    def compute_pass(self, ctx):
        """BaseDemo::compute_pass -> DemoPipeline::compute_pass (src/demos/pipeline.rs:62-102, :214-225)"""
        ctx.compute_pass()
