"""`python -m volym_amd [run simple | benchmark] [-d]` -- the headless counterpart of the reference's CLI
(`cargo run`, `cargo run -- benchmark`; src/cli.rs:4-56, src/main.rs:42-49).

benchmark: the reference's sweep (src/main.rs:178-345) -- 4 step sizes x {Base, Importance x {10,15,20},
ImportanceCone x {10,15,20}} = 28 rows, 3 trials each, 1024x768, benchmark parameters -- written to
benchmark_results.csv with the reference's columns (src/main.rs:71-85) followed by Mrays/s, algorithmic
bytes/GB/s and the roofline fraction.  Unlike the reference (presented frames over 2 s of wall clock, with
blit, GUI and possibly vsync inside), a trial here is `--secs` of back-to-back compute passes timed with
HIP events.  The reference's teapot .raw files are not distributed (.MISSING_LARGE_BLOBS); pass
--volume/--labels/--segments for real data, otherwise the synthetic stand-in of volym_amd.synth is used.

run simple: renders the interactive default view (src/state.rs:41-55) once and writes screenshot_<unix time>.png
like the reference's `P` key (src/state.rs:85-113).
"""
import argparse
import csv
import sys
import time

import numpy as np

from . import _lib, demo, image, scene, synth

STEP_SIZES = [0.0030, 0.0050, 0.0100, 0.0200]   # src/main.rs:192
IMPORTANCE_STEPS = [10, 15, 20]                  # src/main.rs:193
NUM_TRIALS = 3                                   # src/main.rs:179
CSV_COLUMNS = ["algorithm", "step_size", "importance_steps", "use_cone", "avg_total_frames", "avg_total_time_ms",
               "avg_frame_time_ms", "avg_fps", "std_dev_total_frames", "std_dev_total_time_ms", "std_dev_frame_time_ms",
               "std_dev_fps"]                    # src/main.rs:71-85
EXTRA_COLUMNS = ["mrays_per_s", "b_alg_bytes_per_frame", "algorithmic_gb_per_s", "hbm_roofline_fraction", "n_gpus"]


def sweep_rows():
    """(algorithm, step_size, importance_steps, use_cone) in the reference's order (src/main.rs:197-335)."""
    rows = [("Base", s, 0, False) for s in STEP_SIZES]
    rows += [("Importance", s, n, False) for s in STEP_SIZES for n in IMPORTANCE_STEPS]
    rows += [("ImportanceCone", s, n, True) for s in STEP_SIZES for n in IMPORTANCE_STEPS]
    return rows


def _load_assets(args):
    if args.volume:
        raw = np.fromfile(args.volume, np.uint8)
        labels = np.fromfile(args.labels, np.uint8) if args.labels else np.zeros(0, np.uint8)
        segments = scene.load_segments(args.segments) if args.segments else []
        return raw, labels, segments, "file:" + args.volume
    raw, labels = synth.synth_teapot()
    return raw, labels, synth.TEAPOT_SEGMENTS, "synthetic teapot 256x256x178 (volym_amd.synth)"


def _stats(values):
    m = float(np.mean(values))
    return m, float(np.sqrt(np.mean((np.asarray(values, np.float64) - m) ** 2)))   # population std, src/main.rs:124-158


def benchmark(args):
    W, H = args.width, args.height
    raw, labels, segments, what = _load_assets(args)
    print("volym benchmark: %s, %dx%d, %d rows x %d trials of %.2f s" % (what, W, H, len(sweep_rows()), NUM_TRIALS, args.secs))
    base = scene.StateParameters.benchmark()                        # src/main.rs:180-190
    out_rows = []
    flight = getattr(args, "frames_in_flight", 1)
    with demo.GpuContext(W, H, args.device) as ctx:
        if flight == 2:                                             # before the scene (include/volym_hip.h VOLYM_OPT_FRAMES_IN_FLIGHT)
            ctx.set_option(_lib.OPT_FRAMES_IN_FLIGHT, 2)
        state = scene.State.with_parameters(W / H, base)
        d = demo.Simple.init(ctx, state, volume_raw=raw, labels_raw=labels, segments=segments, dims=(256, 256, 256))

        def trial(n):
            """total milliseconds of n frames: per-launch HIP events on one stream, or -- two frames in flight -- the wall clock of
            n compute passes enqueued back to back (as the reference counts presented frames over wall time, src/main.rs:113-135)"""
            if flight != 2:
                return float(ctx.time_passes(n).sum())
            ctx.sync()
            t0 = time.perf_counter()
            for _ in range(n):
                ctx.compute_pass()
            ctx.sync()
            return (time.perf_counter() - t0) * 1e3

        for (algo, step, isteps, cone) in sweep_rows():
            p = base.replace(raymarching_step_size=step)
            if algo != "Base":
                p = p.replace(use_importance_rendering=1, importance_check_ahead_steps=isteps, use_cone_importance_check=1 if cone else 0)
            state = scene.State.with_parameters(W / H, p)
            state.update()                                          # src/event_loop.rs:100
            d.update_gpu_state(ctx, state)
            ms = ctx.time_passes(8)                                 # warm-up; also sizes the trial
            ctx.settle()                                            # the work list dealt from the warm-up frames is in place
            per = max(float(np.median(ms)), 1e-3)
            n = int(min(max(args.secs * 1e3 / per, 4), 20000))
            frames, times, ftimes, fps = [], [], [], []
            if flight == 2:
                trial(8)                                            # both frame contexts warm, their lists in place
                ctx.settle()
            for _ in range(NUM_TRIALS):
                total = trial(n)
                frames.append(n); times.append(total); ftimes.append(total / n); fps.append(n / (total * 1e-3))
            st = ctx.stats_pass()
            b_alg = st["n_vol"] + st["n_imp"] + 4 * W * H
            ft = float(np.mean(ftimes))
            row = [algo, step, isteps, str(cone).lower()]
            for v in (frames, times, ftimes, fps):
                row.append(_stats(v)[0])
            for v in (frames, times, ftimes, fps):
                row.append(_stats(v)[1])
            row += [W * H / (ft * 1e-3) / 1e6, b_alg, b_alg / (ft * 1e-3) / 1e9, b_alg / (ft * 1e-3) / 8.0e12, 1]
            out_rows.append(row)
            print("%-14s step %.4f steps %2d: %8.3f ms/frame %9.1f fps %9.0f Mrays/s  B_alg %6.1f MB  %5.1f%% of HBM roofline" % (
                algo, step, isteps, ft, _stats(fps)[0], row[12], b_alg / 1e6, 100 * row[15]), flush=True)
    with open(args.output, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(CSV_COLUMNS + EXTRA_COLUMNS)
        w.writerows(out_rows)
    print("wrote", args.output)
    return 0


def run_simple(args):
    W, H = args.width, args.height
    raw, labels, segments, what = _load_assets(args)
    state = scene.State.with_parameters(W / H, scene.StateParameters())   # interactive defaults, src/state.rs:41-55
    state.update()
    with demo.GpuContext(W, H, args.device) as ctx:
        d = demo.Simple.init(ctx, state, volume_raw=raw, labels_raw=labels, segments=segments, dims=(256, 256, 256))
        d.update_gpu_state(ctx, state)
        d.compute_pass(ctx)
        ctx.sync()
        frame = ctx.read_rgba8()
    path = args.screenshot or ("screenshot_%d.png" % int(time.time()))
    image.write_png(path, frame)
    print("run simple: %s, %dx%d -> %s" % (what, W, H, path))
    return 0


def flythrough(args):
    """Scripted fly-through (volym_amd/flythrough.py): mouse orbit, scroll zoom and every
    widget of the reference's panel over its range (src/gui.rs:198-277), one event + update + compute pass per frame
    (src/event_loop.rs:100-119).  --out DIR keeps every --keep-every-th frame as PNG and writes frames.json: the uniforms
    each kept frame was rendered with (what a test needs to render the same frames with the oracle)."""
    import json
    import os
    from . import flythrough as ft
    W, H = args.width, args.height
    raw, labels, segments, what = _load_assets(args)
    state = scene.State.with_parameters(W / H, scene.StateParameters())      # the interactive defaults, src/state.rs:41-55
    state.update()
    kept, times = [], []
    with demo.GpuContext(W, H, args.device) as ctx:
        d = demo.Simple.init(ctx, state, volume_raw=raw, labels_raw=labels, segments=segments, dims=(256, 256, 256))
        for i, ev in enumerate(ft.script(args.frames)):
            ft.apply(state, ev)
            state.update()
            d.update_gpu_state(ctx, state)
            t0 = time.perf_counter()
            d.compute_pass(ctx)
            ctx.throttle(3)
            times.append(time.perf_counter() - t0)
            if args.out and i % max(1, args.keep_every) == 0:
                ctx.sync()
                name = "fly_%04d.png" % i
                image.write_png(os.path.join(args.out, name), ctx.read_rgba8())
                kept.append({"frame": i, "event": list(ev), "png": name,
                             "camera_uniforms": bytes(state.camera_uniforms()).hex(),
                             "parameter_uniforms": bytes(state.parameter_uniforms()).hex()})
        ctx.sync()
    if args.out:
        with open(os.path.join(args.out, "frames.json"), "w") as f:
            json.dump({"width": W, "height": H, "frames": kept}, f)
    t = np.array(times)
    print("flythrough: %s, %dx%d, %d frames, %d kept: mean %.3f ms per frame (enqueue + back-pressure, 3 frames in flight)" % (
        what, W, H, args.frames, len(kept), t.mean() * 1e3))
    return 0


def turntable(args):
    """The camera orbits the volume the way a mouse drag does (State.process_mouse -> CameraController -> Camera::orbit,
    src/camera.rs:47-61, :96-117), one update + compute pass per frame, every frame a new view: the moving-camera
    figure (the work lists follow the view through the asynchronous cost feedback)."""
    W, H = args.width, args.height
    raw, labels, segments, what = _load_assets(args)
    p = scene.StateParameters.benchmark().replace(raymarching_step_size=args.step)
    state = scene.State.with_parameters(W / H, p)
    state.update()
    times = []
    with demo.GpuContext(W, H, args.device) as ctx:
        d = demo.Simple.init(ctx, state, volume_raw=raw, labels_raw=labels, segments=segments, dims=(256, 256, 256))
        dx = -360.0 / args.frames / 0.2                     # mouse pixels per frame: sensitivity 0.2 deg/px, sign flipped
        dy = -20.0 / args.frames / 0.2
        for i in range(args.frames):
            state.process_mouse(dx, dy if i < args.frames // 2 else -dy)
            state.update()
            d.update_gpu_state(ctx, state)
            times.append(float(ctx.time_passes(1)[0]))
            if args.out and i % max(1, args.frames // args.keep) == 0:
                image.write_png("%s/turntable_%03d.png" % (args.out, i), ctx.read_rgba8())
    t = np.array(times[1:] if len(times) > 1 else times)
    print("turntable: %s, %dx%d, %d frames (one full turn): mean %.3f ms/frame, median %.3f, max %.3f => %.0f Mrays/s" % (
        what, W, H, args.frames, t.mean(), np.median(t), t.max(), W * H / (t.mean() * 1e-3) / 1e6))
    return 0


def main(argv=None):
    ap = argparse.ArgumentParser(prog="volym", description="MI355X ray-march path of volym")
    ap.add_argument("-d", "--debug", action="store_true", help="verbose logging (src/cli.rs:9-11)")
    ap.add_argument("--device", type=int, default=0)
    ap.add_argument("--volume"); ap.add_argument("--labels"); ap.add_argument("--segments")
    sub = ap.add_subparsers(dest="command")
    run = sub.add_parser("run", help="run the demo (default)")
    run.add_argument("demo", nargs="?", default="simple", choices=["simple"])
    run.add_argument("--width", type=int, default=1280); run.add_argument("--height", type=int, default=720)
    run.add_argument("--screenshot")
    b = sub.add_parser("benchmark", help="run benchmarks on all demos")
    b.add_argument("--width", type=int, default=1024); b.add_argument("--height", type=int, default=768)   # src/main.rs:356-359
    b.add_argument("--secs", type=float, default=0.25, help="GPU seconds per trial (the reference uses 2 s of wall clock)")
    b.add_argument("--output", default="benchmark_results.csv")
    b.add_argument("--frames-in-flight", type=int, choices=[1, 2], default=1,
                   help="2: compute passes alternate between two frame contexts on the device (VOLYM_OPT_FRAMES_IN_FLIGHT), frames per wall clock")
    tt = sub.add_parser("turntable", help="scripted orbit of the camera (moving-view timing)")
    tt.add_argument("--width", type=int, default=1920); tt.add_argument("--height", type=int, default=1080)
    tt.add_argument("--frames", type=int, default=72); tt.add_argument("--step", type=float, default=0.01)
    tt.add_argument("--out"); tt.add_argument("--keep", type=int, default=6)
    fl = sub.add_parser("flythrough", help="scripted fly-through: orbit, zoom and every GUI widget over its range")
    fl.add_argument("--width", type=int, default=1280); fl.add_argument("--height", type=int, default=720)
    fl.add_argument("--frames", type=int, default=120); fl.add_argument("--out"); fl.add_argument("--keep-every", type=int, default=10)
    dv = sub.add_parser("devtools", help="3D-Slicer .seg.nrrd -> segments.json + label .raw (volym_devtools)")
    dv.add_argument("nrrd"); dv.add_argument("segments_json"); dv.add_argument("binary_data")
    args = ap.parse_args(argv)
    if args.command == "devtools":
        from . import devtools
        segs, n = devtools.convert(args.nrrd, args.segments_json, args.binary_data)
        print("devtools: %d segments -> %s, %d label bytes -> %s" % (len(segs), args.segments_json, n, args.binary_data))
        return 0
    if args.command == "benchmark":
        return benchmark(args)
    if args.command == "flythrough":
        return flythrough(args)
    if args.command == "turntable":
        return turntable(args)
    if args.command is None:
        args.width, args.height, args.screenshot = 1280, 720, None
    return run_simple(args)


if __name__ == "__main__":
    sys.exit(main())
