#!/usr/bin/env python3
"""bench.py -- Mrays/s + achieved algorithmic GB/s of the ray-march hot path.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

A "step" is one frame: one pass of the ray-march over every pixel of the viewport
(BASELINE.json: bonsai 256^3 uint8 @ 1920x1080, benchmark parameters of src/main.rs:180-190
with step 0.01, the reference's effective camera eye (0.5,0.5,1.5)).  The real .raw is not in
the repository (.MISSING_LARGE_BLOBS), so the volume is volym_amd.synth.synth_bonsai(256).

N > 1 (one process per GPU, launched by torch.distributed.run): the framebuffer is sharded by interleaved
16x16 screen tiles over the ranks, the volume is replicated, and every frame's PACKED shards (only the tiles
that are not constant) are sent straight to rank 0 over RCCL (grouped ncclSend / ncclRecv: the root's inbound
traffic uses all its xGMI links) and assembled into the raster there.  The frame loop is native
(include/volym_mgpu.h): one C call runs the K frames -- rotating buffers, a compute and a communication stream
per device, a captured HIP graph for the static view -- and torch.distributed (gloo) is only the control plane
(the RCCL id, the barriers, the maximum over the ranks of the time).  Total work per step is fixed: "strong".
--virtual-ranks N rehearses the same loop on ONE GPU (N contexts, device copies instead of RCCL).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8 TB/s spec


def build_scene(args):
    from volym_amd import scene, synth
    n = args.volume
    if args.teapot:      # the reference's default dataset, 256x256x178 zero-padded to 256^3 (src/gpu_resources/volume.rs:40-55)
        raw, labels = synth.synth_teapot()
        segments = synth.TEAPOT_SEGMENTS
        n = 256
    elif getattr(args, "scene", "bonsai") in ("ball", "vessels"):      # scheduling rows only (scripts/scene_rows.py): no label map
        raw = synth.synth_ball(n) if args.scene == "ball" else synth.synth_vessels(n)
        labels, segments = np.zeros(raw.size, np.uint8), []
    else:
        raw, labels = synth.synth_bonsai(n, with_labels=True)
        segments = [{"label_value": 2, "importance": 255}, {"label_value": 3, "importance": 0},
                    {"label_value": 4, "importance": 0}]
    dims = (n, n, n)
    volume = scene.prepare_volume(raw, dims, True)
    importances = scene.prepare_volume(scene.map_segments_to_importance(labels, segments), dims, True)
    params = scene.StateParameters.benchmark().replace(
        raymarching_step_size=args.step, use_importance_rendering=1 if args.importance else 0,
        use_cone_importance_check=1 if args.cone else 0, use_gaussian_smoothing=1 if args.gaussian else 0)
    state = scene.State.with_parameters(args.width / args.height, params)
    state.update()   # the frame loop's orbit(0,0,0): eye -> (0.5,0.5,1.5)  (src/event_loop.rs:100)
    return dims, volume, importances, scene.default_lut(), state


def cpu_baseline(args, dims, volume, importances, lut, state, budget_s=12.0):
    """The oracle (CPU restatement, NOT wgpu/lavapipe -- the reference cannot run here) timed on the host cores over
    a bounded sample of the same frame: 16x16-pixel units (the reference's workgroup) handed to one pool of threads
    that persists over the passes (oracle/volym_oracle.c vo_render_timed); median pass."""
    from oracle import oracle as O
    cam = O.CameraUniforms.from_buffer_copy(bytes(state.camera_uniforms()))
    par = O.Parameters.from_buffer_copy(bytes(state.parameter_uniforms()))
    W, H = args.width, args.height
    cores = os.cpu_count() or 1
    filt = 1 if args.linear else 0
    # probe: every 16th row, once, to size the sample
    probe = list(range(0, H, 16))
    secs, _ = O.render_timed(volume, importances, dims, lut, cam, par, W, H, 1, filter=filt, threads=cores, rowlist=probe)
    per_row = float(secs[0]) / len(probe)
    if per_row * H <= budget_s:
        passes = int(min(max(budget_s / max(per_row * H, 1e-6), 3), 25))     # whole frames until the budget is used
        secs, k = O.render_timed(volume, importances, dims, lut, cam, par, W, H, passes, filter=filt, threads=cores)
        dt = float(np.median(secs))
        rays, sample = W * H, "full %dx%d frame, median of %d passes" % (W, H, passes)
    else:
        stride = int(np.ceil(per_row * H / budget_s))
        rows = list(range(0, H, stride))
        secs, k = O.render_timed(volume, importances, dims, lut, cam, par, W, H, 1, filter=filt, threads=cores, rowlist=rows)
        dt = float(secs[0])
        rays, sample = W * len(rows), "every %d-th row of the %dx%d frame (%d rows), one pass" % (stride, W, H, len(rows))
    return {"value": rays / dt / 1e6, "unit": "Mrays/s", "cores": cores, "kind": "port",
            "sample": sample + "; 16x16-pixel units over a persistent pool of %d threads; CPU restatement (oracle/volym_oracle.c, gcc -O2), not wgpu/lavapipe" % cores}, k


def frame_check(args, dims, volume, importances, lut, state, got_u8, n_rows=68):
    """Untimed check of the frame the timed loop left behind (N = 1): `n_rows` sampled rows against the oracle,
    rgba8 within 1 LSB.  Returns "ok" or a description of the mismatch."""
    from oracle import oracle as O
    cam = O.CameraUniforms.from_buffer_copy(bytes(state.camera_uniforms()))
    par = O.Parameters.from_buffer_copy(bytes(state.parameter_uniforms()))
    W, H = args.width, args.height
    rows = sorted(set(int(round(y)) for y in np.linspace(0, H - 1, n_rows)))
    _, ref, _ = O.render(volume, importances, dims, lut, cam, par, W, H, filter=1 if args.linear else 0, rowlist=rows, want_f32=False)
    d = np.abs(got_u8[rows].astype(np.int32) - ref[rows].astype(np.int32))
    if int(d.max()) <= 1:
        return "ok"
    return "MISMATCH: %d of %d sampled pixels differ from the oracle by more than 1 LSB (max %d)" % (int((d.max(axis=-1) > 1).sum()), len(rows) * W, int(d.max()))


def make_context(args, demo, _lib, W, H, device, dims, volume, importances, lut, state, frames_in_flight=1):
    ctx = demo.GpuContext(W, H, device)
    if frames_in_flight == 2:                       # before the scene: it goes to both frame contexts (include/volym_hip.h)
        ctx.set_option(_lib.OPT_FRAMES_IN_FLIGHT, 2)
    ctx.set_option(_lib.OPT_KERNEL, args.kernel)
    if args.layout >= 0:
        ctx.set_option(_lib.OPT_VOLUME_LAYOUT, args.layout)
    if args.xcd_bands >= 0:
        ctx.set_option(_lib.OPT_XCD_BANDS, args.xcd_bands)
    if args.dp is not None:
        ctx.set_option(_lib.OPT_DEPTH_PARALLEL, args.dp)
    ctx.set_volume(volume, dims, _lib.FILTER_LINEAR if args.linear else _lib.FILTER_NEAREST)
    ctx.set_importances(importances, dims)
    ctx.set_transfer_function(lut)
    ctx.update(state.camera_uniforms(), state.parameter_uniforms())
    return ctx


class TorchGatherLoop:
    """The N-process frame loop with torch.distributed carrying the packed shards -- what `--gpus N` falls back to when the native
    loop (RCCL inside libvolym_hip.so, `volym_mgpu_*`) cannot be created on some rank.  The same protocol (volym_amd/csrc/mgpu.inc,
    tests/test_distributed_gloo.py): a probe frame sizes the message (maximum of the stored tiles over the ranks), every frame each
    rank marches its tiles, packs the ones that are not constant and the root gathers and assembles.  Backend "nccl" is RCCL as
    torch loads it (device buffers); VOLYM_BENCH_FALLBACK=gloo stages through the host (the rehearsal on a one-GPU box, where RCCL
    refuses two ranks on one device).  Frames are serialised on one stream: a fallback, not a tuned path."""

    def __init__(self, ctx, torch, dist, rank, world, dev, backend):
        self.ctx, self.torch, self.dist, self.rank, self.world, self.dev, self.backend = ctx, torch, dist, rank, world, dev, backend
        self.group = dist.new_group(backend=backend) if backend == "nccl" else None     # gloo: the default group
        ctx.set_shard(rank, world)
        ctx.set_stream(torch.cuda.current_stream(dev).cuda_stream)
        self.msg = 0

    def update(self, cu, pu):
        self.ctx.update(cu, pu)

    def prepare(self):
        torch, dist, ctx = self.torch, self.dist, self.ctx
        cap = ctx.packed_shard_bytes(1 << 30)
        probe = torch.empty(cap, dtype=torch.uint8, device=self.dev)
        ctx.compute_pass()
        ctx.pack_shard(probe.data_ptr(), cap)
        used, _ = ctx.packed_tiles()
        t = torch.tensor([used], dtype=torch.int64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        self.msg = ctx.packed_shard_bytes(int(t.item()))
        self.packed = torch.zeros(self.msg, dtype=torch.uint8, device=self.dev)
        self.gathered = torch.zeros(self.msg * self.world, dtype=torch.uint8, device=self.dev) if self.rank == 0 else None

    def run(self, frames):
        torch, dist, ctx, msg = self.torch, self.dist, self.ctx, self.msg
        over = 0
        for _ in range(frames):
            ctx.compute_pass()
            ctx.pack_shard(self.packed.data_ptr(), msg)
            if self.backend == "nccl":
                parts = [self.gathered[r * msg:(r + 1) * msg] for r in range(self.world)] if self.rank == 0 else None
                dist.gather(self.packed, parts, dst=0, group=self.group)
            else:
                torch.cuda.current_stream(self.dev).synchronize()
                host = self.packed.cpu()
                parts = [torch.empty(msg, dtype=torch.uint8) for _ in range(self.world)] if self.rank == 0 else None
                dist.gather(host, parts, dst=0)
                if self.rank == 0:
                    self.gathered.copy_(torch.cat(parts))
            if self.rank == 0:
                ctx.assemble_packed(self.gathered.data_ptr(), msg)
        torch.cuda.current_stream(self.dev).synchronize()
        over |= ctx.packed_tiles()[1]
        return {"overflowed": over, "graph_replays": 0, "msg_bytes": msg, "enqueue_us_per_frame": None, "frames": frames, "wall_ms": None}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--volume", type=int, default=256)
    ap.add_argument("--step", type=float, default=0.01)
    ap.add_argument("--dp", type=int, default=None, help="VOLYM_OPT_DEPTH_PARALLEL (tuning runs; default: the library's choice)")
    ap.add_argument("--layout", type=int, default=-1, help="volume layout: -1 by size (bricks beyond 64 MiB), 0 linear, 1 4x4x4 bricks")
    ap.add_argument("--kernel", type=int, default=2,
                    help="0 direct (BASELINE configs[1]), 1 macro-cell, 2 persistent workgroups + LDS-staged tables/distance field + shading queue (configs[2], default), 3 ray pool (DESIGN.md 4.1)")
    ap.add_argument("--virtual-ranks", type=int, default=0, help="rehearsal on one GPU: N contexts on device 0 through the native multi-GPU loop, device copies instead of RCCL")
    ap.add_argument("--no-graph", action="store_true", help="N > 1: plain enqueues instead of replaying a captured HIP graph")
    ap.add_argument("--linear", action="store_true", help="trilinear volume filter (north_star mode)")
    ap.add_argument("--importance", action="store_true")
    ap.add_argument("--gaussian", action="store_true", help="use_gaussian_smoothing = 1 (the interactive default, src/state.rs:50)")
    ap.add_argument("--cone", action="store_true")
    ap.add_argument("--xcd-bands", type=int, default=-1)
    ap.add_argument("--frames-in-flight", type=int, choices=[1, 2], default=2,
                    help="N = 1: VOLYM_OPT_FRAMES_IN_FLIGHT -- 2 (default): compute passes alternate between two frame contexts on the device "
                         "(two streams, two frame buffers), the next frame's workgroups take the CUs the previous frame's tail leaves idle; "
                         "1: one frame after the other.  The roofline leg always times the kernel alone")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-moving-view", action="store_true", help="skip the first-frame and turntable timings (N = 1)")
    ap.add_argument("--turntable-frames", type=int, default=720)
    ap.add_argument("--turntable-degrees", type=float, default=0.25, help="rotation between consecutive views of the turntable")
    ap.add_argument("--no-frame-check", action="store_true", help="skip the untimed check of the frame against the oracle")
    ap.add_argument("--scene", choices=["bonsai", "ball", "vessels"], default="bonsai", help="synthetic volume of the scheduling rows (the metric is quoted on bonsai)")
    ap.add_argument("--workload", choices=["c1", "c2", "c3", "c4", "c5"],
                    help="BASELINE.json configs[0..4] presets (default = c3, the configuration the metric is quoted on): "
                         "c1 teapot 512x512, c2 bonsai 1080p direct kernel, c3 bonsai 1080p, c4 bonsai 4K, c5 synthetic 1024^3 + labels 4K importance")
    args = ap.parse_args()
    if args.workload == "c1":
        args.width = args.height = 512
        args.teapot = True
    elif args.workload == "c2":
        args.kernel = 0
    elif args.workload == "c4":
        args.width, args.height = 3840, 2160
    elif args.workload == "c5":
        args.width, args.height, args.volume, args.importance = 3840, 2160, 1024, True
    args.teapot = getattr(args, "teapot", False)

    import torch
    import torch.distributed as dist
    from volym_amd import _lib, demo, mgpu

    procs = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if procs != args.gpus:
        if procs == 1 and args.gpus > 1:
            raise SystemExit("--gpus %d needs `python -m torch.distributed.run --nproc-per-node %d bench.py ...`" % (args.gpus, args.gpus))
        args.gpus = procs
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback exists)")
    if os.environ.get("VOLYM_BENCH_ONE_DEVICE") == "1":      # rehearsal of the N-process path on a one-GPU box (with VOLYM_BENCH_FALLBACK=gloo)
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    world = args.virtual_ranks if (procs == 1 and args.virtual_ranks > 1) else procs    # ranks the frame is sharded over
    if procs > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo")          # control plane only; the data path is RCCL inside libvolym_hip.so

    dims, volume, importances, lut, state = build_scene(args)
    W, H = args.width, args.height
    exit_code = 0
    gather_check = frame_check_result = None
    first_frame_ms = moving_view_ms = static_views_ms = ray_pool_ms = None
    mg_info = None

    if world == 1:
        # ---------------------------------------------- one GPU -----------------------------------------------------
        ctx = make_context(args, demo, _lib, W, H, local_rank, dims, volume, importances, lut, state, args.frames_in_flight)
        frame = torch.empty(W * H * 4, dtype=torch.uint8, device=dev)
        ctx.bind_output(None, frame.data_ptr())
        # ---- beside the steady state (N = 1), BEFORE the timed region (these ~50 ms of frames also bring the device to its
        # sustained clocks: a 200-step run from a cold device reads 8 % slower than the steady state it is meant to quote): the first frame of a view nobody has measured (centre-first list, what a
        # context without cost feedback runs every frame) and a moving view: a turntable of the orbit camera, every frame a new
        # pose (src/camera.rs:47-61, src/event_loop.rs:100-119), the lists following it through the asynchronous feedback ------
        if args.kernel == 2 and not args.no_moving_view:
            from volym_amd import scene
            ctx.set_option(_lib.OPT_COST_FEEDBACK, 0)
            ctx.update(state.camera_uniforms(), state.parameter_uniforms())
            ctx.time_batch(5)
            first_frame_ms = ctx.time_batch(20) / 20
            ctx.set_option(_lib.OPT_COST_FEEDBACK, 1)
            n_tt = args.turntable_frames
            deg = args.turntable_degrees
            st2 = scene.State.with_parameters(W / H, scene.StateParameters.benchmark().replace(raymarching_step_size=args.step))
            views = []
            for i in range(n_tt + 60):
                st2.process_mouse(-deg / 0.2, 0.0)                    # sensitivity 0.2 degrees per pixel (src/state.rs:63)
                st2.update()
                views.append((st2.camera_uniforms(), st2.parameter_uniforms()))
            # the steady state of the same views (each rendered until its own list is in place): what "static" means along the path
            static_ms = []
            for cu, pu in views[60::max(1, n_tt // 8)][:8]:
                ctx.update(cu, pu)
                for _ in range(2):           # a standing view after a move: a measuring list first, then the final one
                    ctx.time_batch(3)
                    ctx.settle()
                static_ms.append(ctx.time_batch(10) / 10)
            static_views_ms = float(np.mean(static_ms))
            # the turntable itself: one update + compute pass per view, back to back, at most 3 frames ahead of the device (the
            # back-pressure a swap chain gives the reference's loop, src/event_loop.rs:114)
            for cu, pu in views[:60]:                                # lead-in: the feedback picks the motion up
                ctx.update(cu, pu)
                ctx.compute_pass()
                ctx.throttle(3)
            ctx.sync()
            t1 = time.perf_counter()
            for cu, pu in views[60:]:
                ctx.update(cu, pu)
                ctx.compute_pass()
                ctx.throttle(3)
            ctx.sync()
            moving_view_ms = (time.perf_counter() - t1) / n_tt * 1e3
            # the other march kernel on the same view (VOLYM_OPT_KERNEL = 3, the ray pool of DESIGN.md 4.1: no work list, no feedback): an A/B
            # number beside the headline, not part of it
            ctx.set_option(_lib.OPT_KERNEL, 3)
            ctx.update(state.camera_uniforms(), state.parameter_uniforms())
            ctx.time_batch(20)
            ray_pool_ms = ctx.time_batch(100) / 100
            ctx.set_option(_lib.OPT_KERNEL, 2)
            ctx.update(state.camera_uniforms(), state.parameter_uniforms())   # back to the bench view for what follows
            for _ in range(2):
                ctx.time_batch(3)
                ctx.settle()

        for _ in range(args.warmup):
            ctx.compute_pass()
        ctx.sync()
        ctx.settle()        # the work lists dealt from the warm-up frames' costs are in place before the timed region
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(args.steps):
            ctx.compute_pass()
        ctx.sync()
        torch.cuda.synchronize(dev)
        dt = time.perf_counter() - t0
        local = ctx
    else:
        # ---------------------------------------------- N GPUs (or N virtual ranks) ---------------------------------
        fallback = None
        if procs > 1:
            # the native loop needs RCCL inside the library on EVERY rank: the ranks agree (gloo) before anybody enters a collective,
            # and fall back together to torch.distributed carrying the shards (TorchGatherLoop) when one of them could not
            mg, why = None, ""
            forced = os.environ.get("VOLYM_BENCH_FALLBACK", "")
            try:
                if forced:
                    raise RuntimeError("VOLYM_BENCH_FALLBACK=%s" % forced)
                obj = [mgpu.unique_id() if rank == 0 else None]
            except Exception as e:                               # librccl not loadable, no id
                obj, why = [None], "%s: %s" % (type(e).__name__, e)
            dist.broadcast_object_list(obj, src=0)
            ok = torch.tensor([0 if (obj[0] is None or why) else 1], dtype=torch.int64)
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            if int(ok.item()) == 1:
                mg = mgpu.MultiGpu(W, H, rank=rank, world=world, device_id=local_rank, uid=obj[0])
            else:
                fb_ctx = make_context(args, demo, _lib, W, H, local_rank, dims, volume, importances, lut, state)
                fallback = TorchGatherLoop(fb_ctx, torch, dist, rank, world, dev, "gloo" if forced == "gloo" else "nccl")
                if rank == 0:
                    print("bench.py: native multi-GPU loop unavailable (%s); torch.distributed (%s) carries the shards" % (why or "another rank failed", fallback.backend), file=sys.stderr)
        else:
            mg = mgpu.MultiGpu(W, H, devices=[local_rank] * world, transport=mgpu.COPY)
        if fallback is not None:
            fallback.update(state.camera_uniforms(), state.parameter_uniforms())
            fallback.prepare()
            fallback.run(max(args.warmup, 1))
            dist.barrier()
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            tim = fallback.run(args.steps)
            torch.cuda.synchronize(dev)
            dist.barrier()
            torch.cuda.synchronize(dev)
            dt = time.perf_counter() - t0
            t = torch.tensor([dt, float(tim["overflowed"])], dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt, over = float(t[0]), int(t[1])
            mg_info = {"loop": "fallback: Python loop, one stream, frames serialised (TorchGatherLoop)",
                       "transport": "torch.distributed gather, backend %s" % fallback.backend,
                       "packed_bytes_per_rank_and_frame": tim["msg_bytes"], "whole_shard_bytes": mgpu_shard_bytes(W, H, world),
                       "host_enqueue_us_per_frame": None, "per_stage_ms_rank0": None}
            if rank == 0:
                assembled = fb_ctx.read_rgba8()
            fb_ctx.set_shard(0, 1)                               # the reference frame of the self-check is the whole frame
            fb_ctx.update(state.camera_uniforms(), state.parameter_uniforms())
            if rank == 0:
                fb_ctx.compute_pass()
                fb_ctx.sync()
                ref = fb_ctx.read_rgba8()
                gather_check = "ok" if np.array_equal(assembled, ref) else "MISMATCH: the assembled frame differs from the frame one context renders alone in %d bytes" % int((assembled != ref).sum())
                if over:
                    gather_check = "OVERFLOW: a packed shard ran out of room"
                if gather_check == "ok" and not args.no_frame_check:
                    frame_check_result = frame_check(args, dims, volume, importances, lut, state, assembled)
            fb_ctx.set_shard(rank, world)
            fb_ctx.update(state.camera_uniforms(), state.parameter_uniforms())
            local = fb_ctx
        else:
            mg.set_option(_lib.OPT_KERNEL, args.kernel)
            if args.layout >= 0:
                mg.set_option(_lib.OPT_VOLUME_LAYOUT, args.layout)
            mg.set_volume(volume, dims, _lib.FILTER_LINEAR if args.linear else _lib.FILTER_NEAREST)
            mg.set_importances(importances, dims)
            mg.set_transfer_function(lut)
            mg.update(state.camera_uniforms(), state.parameter_uniforms())
            mg.prepare(0)                       # sizes the packed messages: one untimed frame, maximum over the ranks
            # HIP-graph replay of the frame cycle: on for one process (virtual ranks, device copies: tested on the 1-GPU box); with one
            # process per GPU the cycle contains grouped ncclSend/ncclRecv, and capturing those could not be rehearsed on a 1-GPU box
            # (RCCL refuses two ranks on one device) -- plain enqueues unless VOLYM_MGPU_GRAPH=1 asks for the graph
            use_graph = (not args.no_graph) and (procs == 1 or os.environ.get("VOLYM_MGPU_GRAPH", "0") == "1")
            mg.run(max(args.warmup, 1), use_graph)
            torch.cuda.synchronize(dev)
            if procs > 1:
                dist.barrier()
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            try:
                tim = mg.run(args.steps, use_graph)
            except _lib.VolymError as e:                      # (an overflowed packed shard is an error of the run: report it, keep the ranks in step)
                if "overflowed" not in str(e):
                    raise
                tim = {"overflowed": 1, "graph_replays": 0, "msg_bytes": 0, "enqueue_us_per_frame": 0.0, "frames": args.steps, "wall_ms": 0.0}
            torch.cuda.synchronize(dev)
            if procs > 1:
                dist.barrier()
            torch.cuda.synchronize(dev)
            dt = time.perf_counter() - t0
            over = tim["overflowed"]
            if procs > 1:
                t = torch.tensor([dt, float(over)], dtype=torch.float64)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                dt, over = float(t[0]), int(t[1])
            split = mg.profile(8)
            mg_info = {"loop": "native (volym_mgpu_run): %s" % ("HIP graph, %d replays of a 4-frame cycle" % tim["graph_replays"] if tim["graph_replays"] else "plain enqueues"),
                       "transport": "RCCL grouped ncclSend/ncclRecv to rank 0" if procs > 1 else "device copies (virtual ranks on one GPU)",
                       "packed_bytes_per_rank_and_frame": tim["msg_bytes"], "whole_shard_bytes": mgpu_shard_bytes(W, H, world),
                       "host_enqueue_us_per_frame": tim["enqueue_us_per_frame"], "per_stage_ms_rank0": split}
            # untimed self-check: the gathered + assembled frame must equal the frame one context renders alone
            if rank == 0:
                assembled = mg.read_rgba8()
                solo = make_context(args, demo, _lib, W, H, local_rank, dims, volume, importances, lut, state)
                solo.compute_pass()
                solo.sync()
                ref = solo.read_rgba8()
                solo.close()
                gather_check = "ok" if np.array_equal(assembled, ref) else "MISMATCH: the assembled frame differs from the frame one context renders alone in %d bytes" % int((assembled != ref).sum())
                if over:
                    gather_check = "OVERFLOW: a packed shard ran out of room"
                if gather_check == "ok" and not args.no_frame_check:
                    frame_check_result = frame_check(args, dims, volume, importances, lut, state, assembled)
            local = demo.GpuContext.borrow(mg.context_handle(0), W, H)

    # ---- untimed self-check of the N = 1 path: the frame the timed loop left behind (a steady-state frame: cost-ordered
    # work lists, super-fill stores, no float buffer) against a fresh context's first frame (bit-equal) and against the
    # oracle on sampled rows (<= 1 LSB) -------------------------------------------------------------------------------
    if world == 1 and not args.no_frame_check:
        got = frame.cpu().numpy().reshape(H, W, 4)
        solo = make_context(args, demo, _lib, W, H, local_rank, dims, volume, importances, lut, state)
        solo.compute_pass()
        solo.sync()
        first = solo.read_rgba8()
        solo.close()
        if not np.array_equal(first, got):
            frame_check_result = "MISMATCH: the steady-state frame differs from a fresh context's first frame in %d bytes" % int((first != got).sum())
        else:
            frame_check_result = frame_check(args, dims, volume, importances, lut, state, got)

    # ---- roofline of the dominant kernel: HIP events on the kernel's stream, algorithmic bytes from the
    # instrumented launch (reference fetch counts) -------------------------------------------------------
    # (on a warm device: the untimed checks above left it idle for a second, and a kernel of 35 us read from an idle device is
    # up to 10 % slower than the same kernel in a sustained run)
    n_ev = min(max(args.steps, 500), 2000)
    local.time_batch(3)
    local.settle()
    local.time_batch(3)
    local.settle()
    local.time_batch(1500)
    kernel_ms = local.time_batch(n_ev) / n_ev      # HIP events on the kernel's stream, one pair around n_ev launches
    local.sync()
    st = local.stats_pass()
    counts = [st["n_vol"], st["n_imp"], st["n_rays"]]
    if procs > 1:
        t = torch.tensor(counts, dtype=torch.int64)
        dist.all_reduce(t)
        counts = [int(x) for x in t.tolist()]
    elif world > 1:                                  # virtual ranks: add the other contexts' counts
        for i in range(1, world):
            s_i = mg.stats_pass(i)
            counts = [counts[0] + s_i["n_vol"], counts[1] + s_i["n_imp"], counts[2] + s_i["n_rays"]]
    n_vol, n_imp, n_rays = counts
    b_vol = 8 if args.linear else 1
    local_bytes = st["n_vol"] * b_vol + st["n_imp"] + 4 * st["n_rays"]      # this rank's launch
    frame_bytes = n_vol * b_vol + n_imp + 4 * W * H                           # whole frame (B_alg)
    achieved = local_bytes / (kernel_ms * 1e-3) / 1e9

    # the instantiation launch_march picks for this workload (raymarch.hip): <TABLE, COUNT, TRACE, K, IMP, BRICK, IR, WAVES>
    continuous = bool(args.linear or args.gaussian)
    bricked = args.layout == 1 or (args.layout < 0 and dims[0] * dims[1] * dims[2] > (64 << 20))
    ir = bool(args.importance and not continuous)
    pq_kernel_name = "volym_raymarch_pq_kernel<%s,false,false,%d,%s,%s,%s,%d>" % (
        "false" if continuous else "true", 1 if continuous else 4, "true" if continuous else "false",
        "true" if bricked else "false", "true" if ir else "false", 12 if (ir and bricked) else 16)
    # HBM-side traffic of one launch: PMC counters cannot be read from inside this process; use the committed
    # rocprofv3 measurement of this exact workload when there is one (profiles/rNN_traffic.json)
    traffic, traffic_src = None, None
    if world == 1 and args.kernel == 2 and not (args.linear or args.importance or args.cone or args.gaussian) and (W, H, args.volume, args.step) == (1920, 1080, 256, 0.01):
        import glob
        for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_traffic.json")), reverse=True):
            try:
                traffic = int(json.load(open(f))["traffic_bytes_per_launch"])
                traffic_src = os.path.basename(f)
                break
            except Exception:
                pass
    if rank == 0:
        rays = W * H
        out = {
            "metric": "Mrays/s + achieved HBM GB/s, 256^3 uint8 @ 1920x1080",
            "value": rays * args.steps / dt / 1e6,
            "unit": "Mrays/s",
            "n_gpus": procs if procs > 1 else 1,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "u8 voxels, f32 compositing",
            "data": "synthetic (volym_amd.synth.%s, seed 20250310)" % ("synth_teapot()" if args.teapot else "synth_%s(%d)" % (args.scene, args.volume)),
            "config": {
                "workload": ("teapot 256x256x178->256^3" if args.teapot else "bonsai %d^3" % args.volume) + " uint8 @ %dx%d, %s filter, step %g, thr 0.15, opacity on%s, kernel=%s (BASELINE configs[%d])"
                            % (W, H, "linear" if args.linear else "nearest (reference parity)", args.step,
                               (", importance look-ahead %s" % ("cone" if args.cone else "straight") if args.importance else "") + (", gaussian smoothing" if args.gaussian else ""),
                               {0: "direct", 1: "macro-cell", 2: "persistent workgroups, TF tables + distance field in LDS, shading queue, wave-ballot exit",
                                3: "ray pool: ray state and phase lists in LDS, pixel-block lattice dealing"}[args.kernel], 1 if args.kernel == 0 else 2),
                "viewport": [W, H], "volume": list(dims), "tile_sharding": "interleaved 16x16 tiles, k %% %d" % world,
                "virtual_ranks": (world if (procs == 1 and world > 1) else None),
                "gather": mg_info,
                "frames_in_flight": (args.frames_in_flight if world == 1 else 1),   # N = 1: VOLYM_OPT_FRAMES_IN_FLIGHT of the timed loop
            },
            "frames_in_flight": (args.frames_in_flight if world == 1 else None),
            "gather_check": gather_check,
            "frame_check": frame_check_result,
            "first_frame_ms": first_frame_ms,
            "moving_view_ms": moving_view_ms,
            "static_views_ms": static_views_ms,
            "ray_pool_ms": ray_pool_ms,
            "moving_view": (None if moving_view_ms is None else "turntable of the orbit camera, %d views %.2f degrees apart, one update + compute pass each, back to back with at most 3 frames in flight (wall clock); static_views_ms = steady state of 8 of those views" % (args.turntable_frames, args.turntable_degrees)),
            "achieved_gbs": frame_bytes * args.steps / dt / 1e9,
            "b_alg_bytes_per_frame": frame_bytes,
            "b_alg_bytes_per_ray": frame_bytes / rays,
            "roofline": {
                "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                "kernel": pq_kernel_name if args.kernel == 2 else "volym_raymarch_pool_kernel<false>" if args.kernel == 3 else "volym_raymarch_kernel<%d,false,false>" % args.kernel,
                "kernel_avg_ms": kernel_ms, "launch_algorithmic_bytes": local_bytes,
                "throughput_frac": frame_bytes * args.steps / dt / 1e9 / HBM_PEAK_GBS,      # algorithmic bytes of the timed loop / its time / peak

                "note": "algorithmic bytes = reference fetch count (n_vol*%d + n_imp) + 4 B/pixel of %s; the 32 MiB working set is "
                        "Infinity-Cache resident, so HBM traffic << algorithmic bytes (DESIGN.md)" % (b_vol, "this rank's launch" if world > 1 else "the launch")
                        + ("; frac and kernel_avg_ms are the kernel's own (back-to-back launches on one stream); ms_per_step is below kernel_avg_ms because the "
                           "timed loop keeps two frames in flight (VOLYM_OPT_FRAMES_IN_FLIGHT = 2: alternate frames on two streams, the next frame's workgroups "
                           "fill the CUs the previous frame's tail leaves idle) -- achieved_gbs is the algorithmic-byte rate of that loop" if (world == 1 and args.frames_in_flight == 2) else ""),
            },
        }
        if not args.no_cpu_baseline and world == 1:
            cb, _ = cpu_baseline(args, dims, volume, importances, lut, state)
            out["cpu_baseline"] = cb
        else:
            out["cpu_baseline"] = None
        failed = [c for c in (gather_check, frame_check_result) if c is not None and c != "ok"]
        if failed:                       # a wrong frame is not a benchmark result
            out["value"] = None
            out["roofline"]["frac"] = None
            out["error"] = "; ".join(failed)
        print(json.dumps(out))
        exit_code = 1 if failed else 0
    if world == 1:
        ctx.close()
    elif mg is not None:
        mg.close()
    else:
        local.close()
    if procs > 1:
        code = torch.tensor([exit_code], dtype=torch.int64)
        dist.broadcast(code, src=0)
        exit_code = int(code.item())
        dist.destroy_process_group()
    sys.exit(exit_code)


def mgpu_shard_bytes(W, H, world):
    from volym_amd import sharding
    return sharding.shard_bytes(W, H, world)


if __name__ == "__main__":
    main()
