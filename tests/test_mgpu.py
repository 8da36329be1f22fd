"""The native multi-GPU frame loop (include/volym_mgpu.h) on one GPU: virtual ranks through the COPY transport (the
protocol -- march, pack, gather, assemble, rotating buffers -- is the RCCL one with device copies in place of send/recv),
the single-rank RCCL/graph path, and the host mirror of the packed format."""
import numpy as np
import pytest

from tests import common


def _scene(W, H):
    from volym_amd import scene
    raw, labels = common.bonsai(64)
    dims = (64, 64, 64)
    vol = scene.prepare_volume(raw, dims, True)
    imp = scene.prepare_volume(scene.map_segments_to_importance(labels, common.BONSAI_SEGMENTS), dims, True)
    state = scene.State.with_parameters(W / H, scene.StateParameters.benchmark().replace(raymarching_step_size=0.01))
    state.update()
    return dims, vol, imp, scene.default_lut(), state


def _solo(W, H, dims, vol, imp, lut, state):
    from volym_amd import demo
    with demo.GpuContext(W, H, 0) as c:
        c.set_volume(vol, dims, 0)
        c.set_importances(imp, dims)
        c.set_transfer_function(lut)
        c.update(state.camera_uniforms(), state.parameter_uniforms())
        c.compute_pass()
        c.sync()
        return c.read_rgba8()


@pytest.mark.gpu
@pytest.mark.parametrize("world", [1, 2, 3, 8])
def test_virtual_ranks_native_loop(volym_lib, world):
    """N virtual ranks on device 0: the assembled frame after 1, 7 and 23 frames of the native loop equals the frame one
    context renders alone; no packed shard overflows; a view change with slack still fits."""
    from volym_amd import _lib, mgpu
    W, H = 310, 170
    dims, vol, imp, lut, state = _scene(W, H)
    full = _solo(W, H, dims, vol, imp, lut, state)
    with mgpu.MultiGpu(W, H, devices=[0] * world, transport=mgpu.COPY) as mg:
        with pytest.raises(_lib.VolymError) as e:        # frame twins are a single-context option (include/volym_hip.h)
            mg.set_option(_lib.OPT_FRAMES_IN_FLIGHT, 2)
        assert e.value.code == _lib.E_STATE
        mg.set_volume(vol, dims, 0)
        mg.set_importances(imp, dims)
        mg.set_transfer_function(lut)
        mg.update(state.camera_uniforms(), state.parameter_uniforms())
        mg.prepare(0)
        for frames in (1, 7, 23):
            t = mg.run(frames, use_graph=False)
            assert t["overflowed"] == 0 and t["frames"] == frames
            assert np.array_equal(mg.read_rgba8(), full), (world, frames)
        sp = mg.profile(3)
        assert sp["march_ms"] > 0
        # a moving view with 50 % slack in the messages
        mg.prepare(50)
        for k in range(5):
            state.process_mouse(-20.0, 5.0)
            state.update()
            mg.update(state.camera_uniforms(), state.parameter_uniforms())
            t = mg.run(3, use_graph=False)
            assert t["overflowed"] == 0
            assert np.array_equal(mg.read_rgba8(), _solo(W, H, dims, vol, imp, lut, state)), (world, k)


@pytest.mark.gpu
def test_packed_overflow_is_an_error_and_prepare_recovers(volym_lib):
    """Slots sized for a view in which the object is small overflow when the camera moves in: the run fails (a frame with
    transparent tiles in it is not a frame), and preparing again for the new view gives correct frames with the flag clear."""
    from volym_amd import _lib, mgpu
    W, H = 310, 170
    dims, vol, imp, lut, state = _scene(W, H)
    with mgpu.MultiGpu(W, H, devices=[0, 0], transport=mgpu.COPY) as mg:
        mg.set_volume(vol, dims, 0)
        mg.set_importances(imp, dims)
        mg.set_transfer_function(lut)
        state.process_scroll(-40.0)                      # zoom out: few tiles hold anything
        state.update()
        mg.update(state.camera_uniforms(), state.parameter_uniforms())
        mg.prepare(0)
        assert mg.run(3, use_graph=False)["overflowed"] == 0
        state.process_scroll(40.0)                       # back in: more tiles than the slots sized above
        state.update()
        mg.update(state.camera_uniforms(), state.parameter_uniforms())
        with pytest.raises(_lib.VolymError) as e:
            mg.run(3, use_graph=False)
        assert "overflowed" in str(e.value)
        mg.prepare(0)                                    # the documented recovery
        t = mg.run(3, use_graph=False)
        assert t["overflowed"] == 0
        assert np.array_equal(mg.read_rgba8(), _solo(W, H, dims, vol, imp, lut, state))


@pytest.mark.gpu
def test_throttle_accepts_its_whole_range(volym_lib):
    """volym_throttle(8) must wait for the mark made 8 calls ago, not for the frame just enqueued (ring of 9 marks)."""
    from volym_amd import demo
    W, H = 160, 96
    dims, vol, imp, lut, state = _scene(W, H)
    with demo.GpuContext(W, H, 0) as c:
        c.set_volume(vol, dims, 0)
        c.set_importances(imp, dims)
        c.set_transfer_function(lut)
        c.update(state.camera_uniforms(), state.parameter_uniforms())
        for depth in (1, 3, 8):
            for _ in range(20):
                c.compute_pass()
                c.throttle(depth)
        with pytest.raises(Exception):
            c.throttle(9)
        c.sync()


@pytest.mark.gpu
def test_single_rank_rccl_and_graph(volym_lib):
    """One process per device, world = 1: the create_rank path (no peer, so no RCCL traffic) and the graph replay of a
    static view give the solo frame."""
    from volym_amd import mgpu
    W, H = 320, 176
    dims, vol, imp, lut, state = _scene(W, H)
    full = _solo(W, H, dims, vol, imp, lut, state)
    uid = mgpu.unique_id()                                  # dlopen(librccl) + ncclGetUniqueId
    assert len(uid) == 128 and any(uid)
    with mgpu.MultiGpu(W, H, rank=0, world=1, device_id=0, uid=uid) as mg:   # ncclCommInitRank of a one-rank communicator
        mg.set_volume(vol, dims, 0)
        mg.set_importances(imp, dims)
        mg.set_transfer_function(lut)
        mg.update(state.camera_uniforms(), state.parameter_uniforms())
        mg.prepare(0)
        t = mg.run(40, use_graph=True)
        assert t["frames"] == 40
        assert np.array_equal(mg.read_rgba8(), full)
        t2 = mg.run(16, use_graph=False)
        assert t2["graph_replays"] == 0
        assert np.array_equal(mg.read_rgba8(), full)


def test_packed_format_host_mirror():
    """sharding.pack_packed / assemble_packed: the packed protocol's bytes on the host (round trip, constant tiles are not
    stored, a buffer one tile short raises the overflow flag)."""
    from volym_amd import sharding
    rng = np.random.default_rng(5)
    W, H = 150, 90
    frame = np.zeros((H, W, 4), np.uint8)
    frame[..., 3] = 255
    frame[20:70, 40:110] = rng.integers(0, 256, (50, 70, 4), dtype=np.uint8)
    for world in (1, 2, 3, 5):
        cap = sharding.packed_shard_bytes(W, H, world, 1 << 30)
        packs, used = [], []
        for r in range(world):
            shard = sharding.pack_shard(frame, r, world)
            p, u, over = sharding.pack_packed(shard, r, world, W, H, cap)
            assert over == 0
            packs.append(p); used.append(u)
        n_tiles = sharding.tiling(W, H)[2]
        assert sum(used) < n_tiles                      # the constant tiles outside the patch do not travel
        assert np.array_equal(sharding.assemble_packed(np.concatenate(packs), cap, W, H, world), frame)
        tight = sharding.packed_shard_bytes(W, H, world, max(used))
        packs = [sharding.pack_packed(sharding.pack_shard(frame, r, world), r, world, W, H, tight)[0] for r in range(world)]
        assert np.array_equal(sharding.assemble_packed(np.concatenate(packs), tight, W, H, world), frame)
        if max(used) > 0:
            short = sharding.packed_shard_bytes(W, H, world, max(used) - 1)
            r = int(np.argmax(used))
            assert sharding.pack_packed(sharding.pack_shard(frame, r, world), r, world, W, H, short)[2] == 1


@pytest.mark.gpu
def test_packed_kernels_match_host_mirror(volym_lib):
    """volym_pack_shard's bytes decode with the host mirror, and the mirror's packed shards assemble on the device."""
    from volym_amd import demo, sharding
    W, H = 310, 170
    dims, vol, imp, lut, state = _scene(W, H)
    full = _solo(W, H, dims, vol, imp, lut, state)
    world = 3
    with demo.GpuContext(2048, 1024, 0) as scratch:
        mem = scratch.frame_device_ptr()
        ctxs = []
        try:
            for r in range(world):
                c = demo.GpuContext(W, H, 0)
                c.set_shard(r, world)
                c.set_volume(vol, dims, 0)
                c.set_importances(imp, dims)
                c.set_transfer_function(lut)
                c.update(state.camera_uniforms(), state.parameter_uniforms())
                ctxs.append(c)
            cap = ctxs[0].packed_shard_bytes(1 << 30)
            assert cap == sharding.packed_shard_bytes(W, H, world, 1 << 30)
            for r, c in enumerate(ctxs):
                c.compute_pass()
                c.pack_shard(mem + r * cap, cap)
                c.sync()
            # read the packed bytes back through a raw copy of the scratch frame
            raw = scratch.read_rgba8().reshape(-1)[: world * cap]
            assert np.array_equal(sharding.assemble_packed(raw, cap, W, H, world), full)
        finally:
            for c in ctxs:
                c.close()
