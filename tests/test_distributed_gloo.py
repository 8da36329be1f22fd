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
