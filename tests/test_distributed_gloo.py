"""The N > 1 protocol of bench.py on CPU: world_size-2 (and 3) gloo process groups shard the frame by
interleaved 16x16 tiles, gather the shards to rank 0 and assemble the raster there.  The renderer
here is the oracle (there is no GPU in this tier); the shard layout is volym_amd/sharding.py, which
the GPU tests check against the HIP kernels (tests/test_gpu_parity.py)."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, W, H, out_dir):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from oracle import oracle as O
    from tests import common
    from volym_amd import sharding
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    raw, labels = common.bonsai(32)
    dims = (32, 32, 32)
    vol, imp = common.oracle_scene(O, raw, labels, common.BONSAI_SEGMENTS, dims)
    cam = O.benchmark_camera_uniforms(W / H)
    par = O.make_parameters(raymarching_step_size=0.02)
    # every rank renders only the pixel rows its tiles touch, then keeps its own tiles
    _, full, _ = O.render(vol, imp, dims, O.tf_default_lut(), cam, par, W, H, threads=1)
    mine = torch.from_numpy(sharding.pack_shard(full, rank, world).copy())
    assert mine.numel() == sharding.shard_bytes(W, H, world)
    # bench.py's exchange: a rooted gather to rank 0, which assembles the raster
    gathered = torch.empty(mine.numel() * world, dtype=torch.uint8) if rank == 0 else None
    parts = list(gathered.view(world, mine.numel()).unbind(0)) if rank == 0 else None
    work = dist.gather(mine, parts, dst=0, async_op=True)
    work.wait()
    ok = True
    if rank == 0:
        frame = sharding.assemble(gathered.numpy(), W, H, world)
        ok = bool(np.array_equal(frame, full))
    t = torch.tensor([1.0 + rank])
    dist.all_reduce(t, op=dist.ReduceOp.MAX)            # bench.py takes the max over ranks of the step time
    assert float(t) == float(world)
    np.save(os.path.join(out_dir, "ok_%d.npy" % rank), np.array([ok]))
    dist.barrier()
    dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("world", [2, 3])
def test_gloo_shard_gather_assemble(world, tmp_path, oracle):
    import torch.multiprocessing as mp
    W, H = 72, 50                                         # ragged: 5 x 4 tiles, last row/column partial
    port = _free_port()
    mp.spawn(_worker, args=(world, port, W, H, str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        assert bool(np.load(os.path.join(str(tmp_path), "ok_%d.npy" % r))[0]), r


def _packed_worker(rank, world, port, W, H, n_frames, out_dir):
    """The native loop's protocol (volym_amd/csrc/mgpu.inc) with gloo in place of RCCL and the host mirror of the packed
    format in place of the kernels: message size = max over the ranks of the stored tiles of a probe frame (all_reduce MAX)
    plus slack, rotating buffers (4 frames in flight), every peer sends its packed shard straight to the root
    (isend / irecv), the root assembles every frame."""
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from oracle import oracle as O
    from tests import common
    from volym_amd import sharding
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    raw, labels = common.bonsai(32)
    dims = (32, 32, 32)
    vol, imp = common.oracle_scene(O, raw, labels, common.BONSAI_SEGMENTS, dims)
    par = O.make_parameters(raymarching_step_size=0.02)
    frames = []
    for i in range(n_frames):                            # a moving view: every frame differs
        cam = O.benchmark_camera_uniforms(W / H, 7.0 * i, 3.0 * i, 0.1 * i)
        frames.append(O.render(vol, imp, dims, O.tf_default_lut(), cam, par, W, H, threads=1, want_f32=False)[1])
    # prepare: probe frame 0, maximum of the stored tiles over the ranks, 100 % slack for the motion
    cap = sharding.packed_shard_bytes(W, H, world, 1 << 30)
    _, used, _ = sharding.pack_packed(sharding.pack_shard(frames[0], rank, world), rank, world, W, H, cap)
    t = torch.tensor([used], dtype=torch.int64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    msg = sharding.packed_shard_bytes(W, H, world, int(t.item()) * 2 + 8)
    NBUF = 4
    packed = [torch.zeros(msg, dtype=torch.uint8) for _ in range(NBUF)]
    gathered = [torch.zeros(msg * world, dtype=torch.uint8) for _ in range(NBUF)] if rank == 0 else None
    pending = [None] * NBUF
    ok, overflow = True, 0

    def retire(b):
        nonlocal ok
        if pending[b] is None:
            return
        works, i = pending[b]
        for w in works:
            w.wait()
        if rank == 0:
            ok = ok and bool(np.array_equal(sharding.assemble_packed(gathered[b].numpy(), msg, W, H, world), frames[i]))
        pending[b] = None

    for i in range(n_frames):
        b = i % NBUF
        retire(b)                                        # buffer b free again
        p, _, over = sharding.pack_packed(sharding.pack_shard(frames[i], rank, world), rank, world, W, H, msg)
        overflow |= over
        if rank == 0:
            gathered[b][:msg] = torch.from_numpy(p)      # the root packs straight into its slot
            works = [dist.irecv(gathered[b][r * msg:(r + 1) * msg], src=r) for r in range(1, world)]
        else:
            packed[b].copy_(torch.from_numpy(p))
            works = [dist.isend(packed[b], dst=0)]
        pending[b] = (works, i)
    for b in range(NBUF):
        retire(b)
    o = torch.tensor([overflow], dtype=torch.int64)
    dist.all_reduce(o, op=dist.ReduceOp.MAX)
    np.save(os.path.join(out_dir, "pk_%d.npy" % rank), np.array([ok and int(o.item()) == 0]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_gloo_packed_protocol(world, tmp_path, oracle):
    """World 2 and 3, 7 frames (not a multiple of the 4 rotating buffers), a moving view: every assembled frame is checked."""
    import torch.multiprocessing as mp
    W, H = 72, 50
    port = _free_port()
    mp.spawn(_packed_worker, args=(world, port, W, H, 7, str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        assert bool(np.load(os.path.join(str(tmp_path), "pk_%d.npy" % r))[0]), r
