#!/usr/bin/env python3
"""bench.py — headline benchmark of the path-trace hot path on MI355X.

Workload (BASELINE.json configs[1]): the reference default scene (143-triangle model + 2 Lambert materials,
main.cpp:102-107), default camera (camera.h:44-46), 1920x1080, 1 spp, 8 bounces, Kajiya mode, brute-force
LDS-staged intersect loop.  One "step" = one frame = one pass of the hot path over the whole image
(frame k continues the temporal accumulation of frame k-1, as RVPT::update does).  The camera stands still,
so accumulation frames may go out several at a time (--batch B, rvpt_hip_dispatch_frames: one launch over
B frames x pixels, bit-identical to B dispatches).  Default for brute force: B = 64, the ABI's maximum — the K steps go out as few
launches as that allows (20 steps: one launch; 200: 4 x 50), which measured best at every N.  K steps are always K frames of the same work.

  python bench.py [--gpus N] [--steps K] [--warmup W]          (N>1: launched by torch.distributed.run)

N ranks tile-partition the SAME image (strong scaling); the only collective is one RCCL gather of per-tile
radiance to rank 0 when the finished frame is requested after the K-th step; it is timed separately (`frame_request`),
the timed region holds exactly the K steps.  Before the W warm-up steps the GPU is kept busy for --ramp-seconds (untimed) so
that short runs are not measured at the idle shader clock; the clocks read before/after are in the JSON line (`clocks`).
Rank 0 prints one JSON line.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")  # before HIP initialises; see rvpt_amd/__init__.py

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0     # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
FP32_PEAK_TFLOPS = 157.3  # vector FP32 peak


def issue_yardstick(prof, sclk):
    """The measured yardstick beside the nominal issue limit (profiles/valu_issue_yardstick.json, from tools/microbench/valu_issue_probe.hip): shader clocks per
    wave-instruction per SIMD of the bare triangle test, and of the profiled launch of the whole kernel (its SQ_INSTS_VALU / its own duration at its own clock)."""
    try:
        y = json.loads((ROOT / "profiles" / "valu_issue_yardstick.json").read_text())
        bare = y["bare_triangle_test"]
        out = {"source": "profiles/valu_issue_yardstick.json <- profiles/r06_valu_issue_probe.txt", "nominal_clocks_per_wave_inst": 2.0,
               "bare_triangle_test_clocks_per_wave_inst": bare["clocks_per_wave_inst"], "bare_triangle_test_frac_of_nominal": round(2.0 / bare["clocks_per_wave_inst"], 4),
               "fast_class_clocks_per_wave_inst": y["fast_class_clocks_per_wave_inst"], "slow_class_clocks_per_wave_inst": y["slow_class_clocks_per_wave_inst"]}
        if prof and sclk and prof.get("valu_wave_insts_per_launch") and prof.get("kernel_avg_ns"):
            kernel = 1024 * sclk * 1e6 * prof["kernel_avg_ns"] * 1e-9 / prof["valu_wave_insts_per_launch"]
            out["kernel_clocks_per_wave_inst_profiled"] = round(kernel, 3)
            out["kernel_over_bare_test_rate"] = round(bare["clocks_per_wave_inst"] / kernel, 4)
        return out
    except Exception as e:  # the yardstick is a commentary on the fraction, never a reason to fail a bench line
        return {"error": str(e)}
FLOP_PER_TEST = 42        # ray-dependent half of intersect_triangle_fast as executed
VALU_PER_TEST = 38.25     # wave-level VALU instructions per ray-triangle test, counted in the ISA of the intersect loop — a model


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--aa", type=int, default=1)
    ap.add_argument("--bounces", type=int, default=8)
    ap.add_argument("--traversal", choices=["brute", "bvh", "bvh_ordered"], default="brute")
    ap.add_argument("--mode", type=int, default=9, help="render mode of all four quadrants (9 = Kajiya, compute_pass.comp:68-99)")
    ap.add_argument("--camera-mode", type=int, default=0, help="0 pinhole, 1 orthographic, 2 spherical (compute_pass.comp:102-118)")
    ap.add_argument("--batch", type=int, default=0,
                    help="consecutive accumulation frames per dispatch (rvpt_hip_dispatch_frames); 1 = one launch per frame; "
                         "0 = auto: brute force 64 (the ABI's maximum: launches of the packet kernel do not overlap, fewer and larger is better); BVH: "
                         "eight 1920x1080 frames' worth of samples per rank, at most 64 (tools/archive/sweep_batch.sh, tools/archive/sweep_batch_bpc.sh, profiles/README.md)")
    ap.add_argument("--scene", choices=["default", "cornell", "heightfield"], default="default")
    ap.add_argument("--per-lane", action="store_true",
                    help="BVH traversal: rounds 1-3's kernel (binary nodes, every segment per lane: RVPT_HIP_BVH_PER_LANE) instead of the wide-tree kernels (rvpt_bvh4.hip)")
    ap.add_argument("--mixed-packets", action="store_true",
                    help="brute force: round 2's frame kernel (a lane takes its next pixel the moment its pixel is finished) instead of the packet kernel")
    ap.add_argument("--simple", action="store_true", help="one-pixel-per-lane kernel (no ray regeneration)")
    ap.add_argument("--emulate-world", type=int, default=0,
                    help="diagnostic: render only ONE rank's share (--emulate-rank, default 0) of an N-rank tile partition on one GPU")
    ap.add_argument("--emulate-rank", type=int, default=0, help="diagnostic, with --emulate-world N: which rank's share (0..N-1); the N-GPU time is the MAX over ranks")
    ap.add_argument("--ramp-seconds", type=float, default=1.0,
                    help="untimed GPU work before the W warm-up steps so that the shader clock has left its idle state (a 25-frame "
                         "run is ~10 ms of GPU work: measured 1.15 ms vs 0.98 ms per launch cold vs ramped); 0 disables")
    ap.add_argument("--no-launch-split", action="store_true",
                    help="diagnostic: round 2's launch rule (batches of --batch frames, the remainder last: 20 steps at batch 8 = 8 + 8 + 4)")
    ap.add_argument("--launches", default="", help="diagnostic: the timed region's launch sizes, e.g. 10,10 (must sum to --steps); warm-up and ramp keep the rule")
    ap.add_argument("--serial-launches", action="store_true",
                    help="diagnostic (tools/gpu_profile.sh's kernel-trace pass): wait for every launch of the timed region before the next goes out, so "
                         "that a profiler's per-launch durations are durations of ONE kernel and not of three queued behind each other")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-one-frame-leg", action="store_true", help="skip the one-dispatch-per-frame measurement after the timed region (profiling runs: every dispatch of the process then has the same shape)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="target CPU time of the oracle sample")
    return ap.parse_args()


def cpu_baseline(args, tris, mats, nodes, cam, target_s, traversal=None, size=None):
    """The CPU oracle (a port: the reference GLSL cannot run without Vulkan) on a bounded sample of the SAME workload, all
    host cores this process may use (OpenMP over rows, one thread per usable CPU: affinity mask capped by the cgroup quota).  The per-triangle preparation is done once outside the timed calls and
    the thread pool is warmed; the figure is the MEDIAN of >= 5 timed full frames, the spread is reported next to it.  On
    a host too slow for that within the budget the sample is 8 evenly spaced row bands of the same frame, timed 5 times.
    `traversal` / `size` override the workload's own (cpu_baseline_bvh: the reference's LIVE intersect path, intersection.glsl:489-517 -> :361-413,
    on the same frame; cpu_baseline_c1: BASELINE config 1, 256 x 256)."""
    from oracle import oracle
    W, H = size or (args.width, args.height)
    traversal = traversal or args.traversal
    trav = {"bvh": oracle.TRAVERSAL_BVH, "brute": oracle.TRAVERSAL_BRUTE, "bvh_ordered": oracle.TRAVERSAL_BVH_ORDERED}[traversal]
    if size:  # the camera block carries the aspect ratio of the image (camera.cpp:55-66)
        cam = np.array(cam, dtype=np.float32)
        cam[16] = np.float32(W / H)
    s = oracle.settings_bytes(max_bounces=args.bounces, aa=args.aa, current_frame=0)
    # threads = the CPUs this process may really run on (affinity mask and cgroup quota): GPU boxes hand a container anything
    # from one core to all 256 hardware threads, and 256 OpenMP threads on a one-core allowance was the 10x spread of round 1
    cores = oracle.usable_cores()
    oracle.set_threads(cores)
    sc = oracle.PreparedScene(nodes, tris, mats)
    out = np.zeros((H, W, 4), np.float32)

    def bands(rows):
        t0 = time.perf_counter()
        for b in range(8):
            y0 = min(max(H - rows, 0), (b * H) // 8 + max(0, (H // 8 - rows) // 2))
            sc.render(s, cam, W, H, trav, out=out, y0=y0, y1=min(H, y0 + rows))
        return time.perf_counter() - t0

    # a band must feed every thread: the oracle hands out rows four at a time (schedule(dynamic, 4)), so a band of fewer than 4 x cores rows leaves threads
    # idle — round 5's first runs calibrated on 8-row bands, overestimated a frame eightfold on 16 cores and then timed 35-row bands at half the machine (2.2
    # instead of ~4 Msamples/s)
    band_rows = min(max(8, 4 * cores), max(8, H // 8))
    bands(min(4, H // 8) or 1)                # warm-up: thread pool, page faults of `out`
    est_frame = bands(band_rows) * H / (8.0 * band_rows)  # calibrate on 8 bands that keep every thread busy
    times = []
    if est_frame * 5 <= 2.5 * target_s:       # whole frames: at least 5, until ~target_s of CPU work has been done
        sc.render(s, cam, W, H, trav, out=out)
        while (len(times) < 5 or sum(times) < target_s) and len(times) < 256:
            t0 = time.perf_counter()
            sc.render(s, cam, W, H, trav, out=out)
            times.append(time.perf_counter() - t0)
        px = W * H * args.aa
        what = f"{len(times)} full {W}x{H} frame(s)"
    else:                                     # slow host: a bounded band sample of the same frame, 5 repetitions
        rows = max(min(band_rows, H // 8), min(H // 8, int(H * target_s / est_frame / 8 / 5)))
        times = [bands(rows) for _ in range(5)]
        px = 8 * rows * W * args.aa
        what = f"5 x 8 evenly spaced {rows}-row bands of the {W}x{H} frame"
    times.sort()
    med = times[len(times) // 2]
    return {"value": round(px / med / 1e6, 4), "unit": "Msamples/s", "cores": cores, "kind": "port",
            "min": round(px / times[-1] / 1e6, 4), "max": round(px / times[0] / 1e6, 4),
            "sample": f"median of {what}, {sum(times):.1f} s of oracle/rvpt_oracle.c ({traversal}), preparation hoisted, "
                      f"OpenMP on {cores} threads (= usable CPUs)"}


def fan_out(args):
    """`python bench.py --gpus N` (N > 1) started WITHOUT a launcher: become the launcher.  Either N devices are visible and the same
    command line is re-run under `python -m torch.distributed.run --nproc-per-node N` (one rank per GPU, exactly how the driver starts
    the N-GPU bench), or the run fails with the reason — never a one-GPU run that prints a line for N.  (RVPT_BENCH_SHARED_GPU=1, the
    one-GPU test box: all N ranks share cuda:0, gloo + host-staged gather; the line is labelled as a test.)"""
    import subprocess
    import torch
    visible = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if visible < args.gpus and not (os.environ.get("RVPT_BENCH_SHARED_GPU") and visible >= 1):
        raise SystemExit(f"bench.py: {args.gpus} GPUs requested, {visible} visible")
    # --standalone: the launcher's own c10d rendezvous on a port IT binds (no pre-picked port another process could take in between: ADVICE r4),
    # on the loopback address (the container's hostname may not resolve)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--standalone", "--local-addr", "127.0.0.1",
           str(Path(__file__).resolve()), *sys.argv[1:]]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: what RCCL needs between processes on this driver
    env.setdefault("OMP_NUM_THREADS", "1")             # (torchrun would set it, with a warning)
    raise SystemExit(subprocess.run(cmd, env=env).returncode)


def main():
    args = parse()
    if args.gpus > 1 and "RANK" not in os.environ and "WORLD_SIZE" not in os.environ and args.emulate_world <= 1:
        fan_out(args)
    import torch
    import torch.distributed as dist
    from rvpt_amd import build as rv_build, native, scene
    from rvpt_amd.distributed import DistributedRVPT
    from rvpt_amd.renderer import launch_sizes
    if args.no_launch_split:
        launch_sizes = lambda frames, batch, in_flight=1: [batch] * (frames // batch) + ([frames % batch] if frames % batch else [])  # noqa: E731
    if args.launches:
        forced = [int(x) for x in args.launches.split(",")]
        assert sum(forced) == args.steps, "--launches must sum to --steps"
        _rule = launch_sizes
        launch_sizes = lambda frames, batch, in_flight=3: forced if frames == args.steps else _rule(frames, batch, in_flight)  # noqa: E731
    in_flight_hint = [3]  # launches the library rotates over for this kind of launch; refreshed from rvpt_hip_get_launch_info after every run()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:  # (without a launcher and --gpus N > 1, fan_out above has already taken over)
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the product path has no CPU fallback)")
    # RVPT_BENCH_SHARED_GPU=1 (tests on a one-GPU box): every rank renders its tile share on cuda:0, the process group is gloo and
    # the frame gather is staged through the host — the whole N-process flow of this file except RCCL itself.  Not a bench line.
    shared_gpu = bool(os.environ.get("RVPT_BENCH_SHARED_GPU"))
    if shared_gpu:
        local_rank = 0
        os.environ["RVPT_NO_LIBRARY_COMM"] = "1"
    elif local_rank >= torch.cuda.device_count():
        raise SystemExit(f"bench.py: {args.gpus} GPUs requested, {torch.cuda.device_count()} visible (rank {rank} has no cuda:{local_rank})")
    torch.cuda.set_device(local_rank)
    use_dist = world > 1 or ("RANK" in os.environ and "MASTER_ADDR" in os.environ)  # launched by torch.distributed.run
    if use_dist:
        # torch.distributed is the CONTROL PLANE only (gloo: the 128-byte RCCL id, agreement between the ranks, host-side scalars).
        # The process has ONE RCCL communicator — the library's (rvpt_hip_comm_init): it carries the gather of per-tile radiance
        # and the barriers around the timed region (rvpt_hip_comm_barrier).  No torch-nccl group is created.
        dist.init_process_group(backend="gloo")
    reduce_device = "cpu"

    if args.batch <= 0:  # auto
        if args.traversal == "brute":
            # the packet kernel fills every CU's LDS share, so launches in flight do not overlap and every launch is one ramp and one tail:
            # as many frames per launch as the ABI takes (tools/archive/sweep_batch.sh, profiles/r03_batch_sweep.txt: 20 steps as one launch 8 940-9 000
            # against 8 680-8 750 Msamples/s as 7 + 7 + 6; 200 steps as 4 x 50 9 546 against 9 450-9 480 as 25 x 8)
            args.batch = native.MAX_FRAMES_PER_DISPATCH
        else:  # BVH: a launch carries eight 1920x1080 frames' worth of samples per rank (ramp-up and drain paid once per launch)
            share = args.width * args.height * args.aa / max(args.emulate_world, world, 1)
            args.batch = max(1, -(-(1920 * 1080) // int(max(share, 1)))) * 8
            # a rank's share of a partitioned image: the K steps as ONE launch when the ABI's 64 frames allow — a lone launch takes the whole CU
            # (rvpt_abi.hip: choose_launch) and two half launches lose its tail twice (tools/archive/sweep_share_shapes.sh, profiles/r04_share_shapes.txt:
            # rank 2 of 8, C3 one 20-frame launch 0.519 ms per frame against 0.56 as 10 + 10, C4 geometry 0.079 against 0.093)
            if max(args.emulate_world, world) > 1 and args.steps <= native.MAX_FRAMES_PER_DISPATCH:
                args.batch = max(args.batch, args.steps)
    args.batch = min(args.batch, native.MAX_FRAMES_PER_DISPATCH)
    W, H = args.width, args.height
    tris, mats = {"default": scene.default_scene, "cornell": scene.cornell_scene, "heightfield": scene.heightfield_scene}[args.scene]()
    flags = native.TIMING | native.COUNT_SEGMENTS | (native.KERNEL_SIMPLE if args.simple else 0)
    flags |= native.BVH_PER_LANE if args.per_lane else 0
    flags |= native.BRUTE_MIXED_PACKETS if args.mixed_packets else 0
    if args.emulate_world > 1:  # one rank's share of an N-way partition, for scaling forecasts (not a bench line)
        from rvpt_amd import RVPT
        assert 0 <= args.emulate_rank < args.emulate_world
        r = RVPT(W, H, device=local_rank, traversal=args.traversal, tile_rank=args.emulate_rank, tile_world=args.emulate_world, flags=flags)
        r.add_triangles(tris)
        for m in mats:
            r.add_material(m)
        r.render_settings.aa = args.aa
        r.render_settings.max_bounces = args.bounces
        if args.scene == "cornell":      # the cameras of the real path below
            r.scene_camera.translation = np.array([0.0, 2.0, -1.9])
        elif args.scene == "heightfield":
            r.scene_camera.translation = np.array([0.0, 2.5, -5.0])
            r.scene_camera.rotation = np.array([0.0, 25.0, 0.0])
        r.initialize()
        def run_share(frames):
            for n in launch_sizes(frames, args.batch, in_flight_hint[0]):
                r.update()
                r.draw() if n == 1 else r.draw_frames(n)
            in_flight_hint[0] = r.context.launch_info()[3]
        for _ in range(3):  # sample buffers of every slot grown to the largest timed launch, untimed
            run_share(max(launch_sizes(args.steps, args.batch, in_flight_hint[0])))
        if args.ramp_seconds > 0:  # as the real path: clocks out of idle, all untimed
            t_ramp = time.perf_counter()
            while time.perf_counter() - t_ramp < args.ramp_seconds:
                run_share(max(args.batch, 32))
                r.wait()
        run_share(args.warmup)
        r.wait(); r.context.reset_timing()
        t0 = time.perf_counter()
        run_share(args.steps)
        r.wait()
        dt = time.perf_counter() - t0
        _, ksum, n = r.context.timing()
        seg, smp = r.context.stats()
        print(json.dumps({"emulated_world": args.emulate_world, "emulated_rank": args.emulate_rank, "ms_per_frame_wall": round(dt / args.steps * 1e3, 5),
                          "kernel_ms": round(ksum / n, 5), "segments_per_sample": round(seg / max(smp, 1), 4), "launch": r.context.launch_info(),
                          "launches": launch_sizes(args.steps, args.batch, in_flight_hint[0])}))
        r.shutdown()
        return
    r = DistributedRVPT(W, H, traversal=args.traversal, flags=flags, rank=rank, world=world, device=local_rank)
    r.add_triangles(tris)
    for m in mats:
        r.add_material(m)
    r.render_settings.aa = args.aa
    r.render_settings.max_bounces = args.bounces
    rs = r.render_settings
    rs.top_left_render_mode = rs.top_right_render_mode = rs.bottom_left_render_mode = rs.bottom_right_render_mode = args.mode
    r.scene_camera.mode = args.camera_mode
    if args.scene == "cornell":      # inside the box, looking at the model
        r.scene_camera.translation = np.array([0.0, 2.0, -1.9])
    elif args.scene == "heightfield":  # above the terrain, pitched down
        r.scene_camera.translation = np.array([0.0, 2.5, -5.0])
        r.scene_camera.rotation = np.array([0.0, 25.0, 0.0])
    r.initialize()
    ctx = r.local.context

    def barrier():
        # this rank's work first (the library renders on its own streams), then the collective barrier on the library's
        # communicator (a one-float all-reduce; gloo's barrier when the ranks share one GPU in tests): issued while frame
        # kernels still fill every CU, the barrier's own kernel would queue behind them and cost ~1.3 ms instead of its latency
        ctx.wait()
        torch.cuda.synchronize()
        if use_dist:
            r.barrier()
            torch.cuda.synchronize()

    def run(frames):  # RVPT::update + RVPT::draw per frame, or per batch of consecutive accumulation frames
        # as few launches as --batch allows, of near-equal size (renderer.launch_sizes): 20 steps at batch 8 are 7 + 7 + 6, at batch 64
        # (8 ranks) one launch — measured best there (profiles/r03_launch_shapes.txt)
        for n in launch_sizes(frames, args.batch, in_flight_hint[0]):
            r.update()        # frame counter + uniforms
            if n == 1:
                r.draw()      # asynchronous dispatch of one frame
            else:
                r.draw_frames(n)  # ... of n frames as one launch (rvpt_hip_dispatch_frames)
            if args.serial_launches:
                ctx.wait()
        try:
            in_flight_hint[0] = ctx.launch_info()[3]
        except native.NativeError:  # a rank that owns no tile has dispatched nothing
            pass

    # Everything RCCL initialises lazily happens NOW, long before the timed region: the communicator's channels and the code
    # objects of the barrier's kernel are set up on first use, and a first use right before t0 made the first timed
    # launch ~2 ms slower (measured: 5 300 instead of 6 300 Msamples/s over 20 frames).
    barrier()
    barrier()
    run(max(launch_sizes(args.steps, args.batch, in_flight_hint[0])))  # the largest launch of the timed region once, untimed: the library grows
    for _ in range(2):                                                 # its per-launch sample buffers on first use (a drain + allocation), on every
        run(max(launch_sizes(args.steps, args.batch, in_flight_hint[0])))  # slot of the rotation — that must not happen between the barriers
    r.gather_frame()  # the frame-request path once, untimed (buffers, RCCL channels)
    ramp_frames = 0
    if args.ramp_seconds > 0:  # untimed: bring the shader clock out of idle (reported in the JSON line)
        t_ramp = time.perf_counter()
        while time.perf_counter() - t_ramp < args.ramp_seconds:
            run(max(args.batch, 32))
            ramp_frames += max(args.batch, 32)
            ctx.wait()  # ~1 s of GPU work, not 1 s of enqueueing (a deep backlog runs the chip into its sustained-power clocks)
    # the W warm-up steps follow the ramp without a gap and run right up to the opening barrier: a GPU that sat idle for a few
    # hundred microseconds (a host-side wait, a sysfs read) starts the timed region below its sustained clock — over 20 frames
    # that was 6 200-6 370 against 6 440-6 640 Msamples/s with a busy run-up
    run(args.warmup)

    # the timed region: EXACTLY K steps of the hot path between two barriers
    barrier()
    ctx.reset_timing()
    t0 = time.perf_counter()
    run(args.steps)
    barrier()
    elapsed = time.perf_counter() - t0
    # a finished frame is requested once, after the K steps: the one collective of the multi-GPU path (per-tile radiance ->
    # rank 0, untiled there).  Timed on its own — it is not a step of the hot path and happens once per K, whatever K is.
    t1 = time.perf_counter()
    frame = r.gather_frame()
    barrier()
    gather_s = time.perf_counter() - t1

    # (the counters of the timed region are read NOW: the leg below dispatches more frames on the same context)
    _, kernel_ms_sum, n_timed = ctx.timing()
    segments, samples = ctx.stats()
    try:
        timed_launch_info = ctx.launch_info()
    except native.NativeError:  # a rank that owns no tile has dispatched nothing
        timed_launch_info = None

    # One dispatch per frame — the reference's own shape (RVPT::draw: one compute pass per frame, rvpt.cpp:346-354; a moving camera leaves nothing to batch) — timed
    # the same way over 20 .. 300 frames (about a quarter of a second), after the timed region (so that `value` is untouched): what a caller that cannot batch gets (VERDICT r5 #2d)
    one_n = 0 if args.no_one_frame_leg else max(20, min(300, int(0.25 / max(elapsed / args.steps, 1e-9))))  # ~a quarter of a second of frames, 20 .. 300
    # (the counters above were read with the GPU idle: a burst of ~300 launches first — the launch shape's buffers, the shader clock back out of idle, as the ramp and the
    # warm-up steps do for `value` — then five repetitions of one_n frames, each between two barriers; the MEDIAN is reported, as tools/one_frame_per_launch.py does)
    one_reps = []
    for _ in range(300 if one_n else 0):
        r.update(); r.draw()
    if one_n:
        barrier()
    for _ in range(5 if one_n else 0):
        t2 = time.perf_counter()
        for _ in range(one_n):
            r.update(); r.draw()
        barrier()
        one_reps.append(time.perf_counter() - t2)
    one_s = sorted(one_reps)[len(one_reps) // 2] if one_reps else 0.0

    if use_dist:
        t = torch.tensor([elapsed, gather_s, one_s], dtype=torch.float64, device=reduce_device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed, gather_s, one_s = float(t[0].item()), float(t[1].item()), float(t[2].item())
    collective_path, rccl_ranks, rccl_rank, rccl_version = ("rccl" if r.library_comm else ("host-staged" if use_dist else "none")), 0, -1, 0
    if r.library_comm:
        try:
            rccl_ranks, rccl_rank, rccl_version = ctx.comm_info()
        except native.NativeError as e:
            collective_path = f"rccl (comm_info failed: {e})"
    if use_dist:  # every rank must have gathered over RCCL and must have seen the same communicator
        ok = torch.tensor([1 if (r.library_comm and rccl_ranks == world) else 0], dtype=torch.int32, device=reduce_device)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if not bool(ok.item()) and collective_path == "rccl":
            collective_path = "rccl (not on every rank)"
    if use_dist:
        agg = torch.tensor([segments, samples], dtype=torch.float64, device=reduce_device)
        dist.all_reduce(agg, op=dist.ReduceOp.SUM)
        segments, samples = int(agg[0].item()), int(agg[1].item())

    if rank == 0:
        assert frame is not None and tuple(frame.shape) == (H, W, 4)
        assert bool(torch.isfinite(frame).all())
        K = args.steps
        msamples = W * H * args.aa * K / elapsed / 1e6
        kernel_ms = kernel_ms_sum / max(n_timed, 1)  # rank 0's frame kernel, hipEvents on the library's stream
        n_tris = int(tris.shape[0])
        seg_per_sample = segments / max(samples, 1)
        # algorithmic bytes of ONE launch on rank 0 (DESIGN.md "Roofline"): accumulator read+write of the owned
        # pixels, plus the prepared-triangle records every work-group stages into LDS
        own_px = ctx.tile_buffer()[1] // 16
        grid_blocks, lds_bytes, variant, in_flight = timed_launch_info or ctx.launch_info()  # (of the timed region's last launch, not of the one-frame leg's)
        B_nominal = args.batch if in_flight > 1 else 1           # frames per launch asked for (--batch)
        timed_launches = launch_sizes(K, args.batch, in_flight) if in_flight > 1 else [1] * K
        B = K / len(timed_launches)                               # frames per launch of the timed region, on average (20 steps at batch 8: 7 + 7 + 6)
        if variant in (0, 6):  # LDS-resident (6: the packet kernel; its queue of parked paths never leaves LDS): every work-group stages the scene once per launch
            staged = grid_blocks * (lds_bytes - ((4 * (15 if args.aa == 1 else 18) * 64 * 4) if variant == 6 else 0))  # (LDS minus the four path queues: what a work-group stages)
        elif variant == 1:  # LDS-streamed: every WAVE (64 rays) stages the scene once per segment round: S * N * 64 / 64 bytes per sample (SURVEY §8d, R = 64)
            staged = int(segments / K * B / world / 64) * n_tris * 64
        else:               # BVH megakernel: no staging; node/triangle fetches are data dependent (not modelled)
            staged = 0
        # dominant kernel = the frame (trace) kernel.  With frames in flight it writes the 12 B/pixel sample mean
        # (the 48 B/pixel blend traffic belongs to blend_accumulate); fused (in_flight == 1) it reads+writes accum.
        px_bytes = 12 if in_flight > 1 else 32  # (the per-pixel sample mean is a 12-byte record since round 5: rvpt_kernels.h: SampleRGB)
        algo_bytes = own_px * px_bytes * B + staged
        achieved = algo_bytes / (kernel_ms * 1e-3) / 1e9
        # trace + blend, per frame per rank: 12 B written + 12 B read per sample mean, accumulator read+write once per launch
        frame_hbm_bytes = own_px * ((24 + 32 / B) if in_flight > 1 else 32)
        # HBM bytes per launch from the PMC counters are REPLAYED from the committed profile of this exact configuration — and only
        # while the kernel sources still hash to what that profile was taken on (a changed kernel must be re-profiled:
        # tools/gpu_profile.sh + tools/summarize_prof.py); otherwise null, with the reason.
        traffic, traffic_source = None, "no committed PMC profile of this configuration"
        valu_insts_per_frame = None  # SQ_INSTS_VALU per frame of the same profile (the executed instruction count)
        prof = None                  # the replayed entry itself (pipe figures: lane utilisation, wave-time split, the clock the profiled launch ran at)
        pmc = ROOT / "profiles" / "pmc_traffic.json"
        if pmc.exists():
            try:
                rec = json.loads(pmc.read_text())
                key = f"{args.scene}_{W}x{H}_aa{args.aa}_{args.traversal}_n{world}"
                ent = rec.get(key)
                if ent:
                    sha = rv_build.kernel_sha()
                    if ent.get("kernel_sha") == sha:
                        # the profile's launches carry ent["frames_per_launch"] frames; this run's carry B on average: per-launch figures scale with the frames
                        fpl = float(ent.get("frames_per_launch") or B)
                        traffic = int(ent.get("hbm_bytes_per_launch") * B / fpl)
                        prof = ent
                        if ent.get("valu_wave_insts_per_launch"):
                            valu_insts_per_frame = ent["valu_wave_insts_per_launch"] / fpl
                        traffic_source = (f"replayed from {ent.get('source')} (rocprofv3 FETCH_SIZE x2 + WRITE_SIZE per {fpl:g}-frame launch"
                                          + (f", scaled to this run's {B:.2f} frames per launch" if abs(B - fpl) > 1e-9 else "") + f"), kernel sha {sha}")
                    else:
                        traffic_source = f"stale: {ent.get('source')} was taken on kernel sha {ent.get('kernel_sha')}, the sources now hash to {sha}"
            except Exception as e:
                traffic_source = f"profiles/pmc_traffic.json unreadable: {e}"
        tests_per_step = segments / K * n_tris if args.traversal == "brute" else None  # whole job
        # Which roof binds.  Brute force: FP32 VALU (arithmetic intensity ~10^3 FLOP/B against a machine balance of ~20) —
        # `achieved` = ray-triangle tests/s x 42 FLOP (the reference's operation count of the ray-dependent half of
        # intersect_triangle_fast, FMA = 2), whole job.  The contract's HBM figures (algorithmic bytes of ONE launch of the dominant
        # kernel / its hipEvent duration) are kept under roofline.hbm.  BVH: no closed-form operation count exists for a data-dependent
        # walk, so the HBM form stays on top with the note saying what really binds.
        hbm = {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
               "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic, "traffic_source": traffic_source,
               "kernel_ms": round(kernel_ms, 5), "launch_concurrency": min(in_flight, len(timed_launches)),
               "kernel_ms_over_concurrency": round(kernel_ms / max(min(in_flight, len(timed_launches)), 1), 5),
               "algorithmic_bytes_per_launch": int(algo_bytes),
               "sustained": round(frame_hbm_bytes / (elapsed / K) / 1e9, 2)}
        if tests_per_step:
            # The contract's rate: algorithmic work of ONE launch of the dominant kernel / that kernel's launch duration, measured live with HIP
            # events on the stream it runs on (rank 0's launches of the timed region; launches queued behind each other on the rotating streams
            # record the queueing as duration, hence the division by the launches in flight) — this is the figure the committed rocprofv3 kernel
            # trace (profiles/<round>_packets_trace_kernel_stats.csv: AverageNs of a 20-frame launch) can be checked against.  The whole-job wall
            # clock rate (value's own clock: launch latency, barriers and the drain of the last launch included) is kept beside it.
            # ... when the timed region IS one launch (the driver's 20 steps).  Several launches queue behind each other on the rotating streams and their
            # event intervals include the queueing (a 4 x 50-frame run would come out at an impossible 0.71 of peak): then the rate is the wall clock's.
            one_launch = len(timed_launches) == 1 and kernel_ms > 0
            step_s_kernel = (kernel_ms * 1e-3) / B if one_launch else elapsed / K
            tps_wall = tests_per_step / (elapsed / K)
            tps = tests_per_step / step_s_kernel
            tf = tps * FLOP_PER_TEST / 1e12
            issue_nominal = tps * VALU_PER_TEST / 64 / (1024 * 2.4e9 / 2 * world)
            # the shader clock the kernel runs at under this load: GRBM_GUI_ACTIVE / 8 XCDs / launch duration of the committed profile (sysfs
            # reads the idle level on the driver's boxes — 94 MHz — a moment after the GPU went idle: dropped in round 5)
            sclk = (prof or {}).get("profiled_clock_mhz")
            if variant == 6:
                # the packet kernel DECIDES every ray-triangle test but executes only the plane distance (15 of 38.25 VALU) where no ray of a
                # camera packet can accept: the executed instruction count is a counter reading (SQ_INSTS_VALU of the committed profile of
                # this configuration, replayed under the same kernel-sha rule as `traffic`), not the ISA model
                issue_nominal = (valu_insts_per_frame * world / step_s_kernel / (1024 * 2.4e9 / 2 * world)) if valu_insts_per_frame else None
                insts = {"value": (round(valu_insts_per_frame * 64 / tests_per_step, 2) if valu_insts_per_frame else None),
                         "source": "rocprof SQ_INSTS_VALU per frame of the committed profile / decided tests per frame (everything the kernel executes, "
                                   "shading and queue traffic included); a full test is 38.25, its plane-distance half 15 (ISA count)"}
            else:
                insts = {"value": VALU_PER_TEST, "per_accepted_hit": 3, "source": "isa-count (DESIGN.md 5.1; a model, not a counter: "
                         "rocprof SQ_INSTS_VALU of the committed profile is 6 % above it with shading and regeneration)"}
            # Round 5: the packet kernel DECIDES most ray-triangle pairs without executing anything (screen rectangles in camera rounds, the bounce cull's table in
            # bounce rounds), so the reference-equivalent rate — every decided pair counted as the reference's 42 FLOP — exceeds the vector peak and bounds
            # nothing.  The top-level fraction is therefore what the VALU pipe really did: executed wave-instructions per second (SQ_INSTS_VALU of the committed
            # profile of this configuration, replayed under the kernel-sha rule) against the issue limit; the reference-equivalent figures stay beside it by name.
            # Without a current profile (a kernel edited after its last profile) the line falls back to the algorithmic form and says so.
            issue_peak = 1024 * 2.4e9 / 2 * world
            if variant == 6 and issue_nominal:
                top = {"bound": "valu_issue", "achieved": round(issue_nominal * issue_peak / 1e9, 2), "peak": round(issue_peak / 1e9, 1), "unit": "Gwave-inst/s",
                       "frac": round(issue_nominal, 4)}
            elif variant == 6:
                # no current profile to replay the executed instruction count from: the contract's HBM form (algorithmic bytes of one launch / its duration)
                # goes on top — always measurable live — rather than a reference-equivalent rate that exceeds the peak it is divided by
                top = {k: hbm[k] for k in ("bound", "achieved", "peak", "unit", "frac")}
            else:
                top = {"bound": "valu_fp32", "achieved": round(tf, 2), "peak": round(FP32_PEAK_TFLOPS * world, 1), "unit": "TFLOP/s",
                       "frac": round(tf / (FP32_PEAK_TFLOPS * world), 4)}
            roofline = {**top, "reference_equivalent_tflops": round(tf, 2), "vector_peak_tflops": round(FP32_PEAK_TFLOPS * world, 1),
                        # two fractions, named: frac_algorithmic (= frac, kept for continuity) counts every DECIDED ray-triangle test
                        # as the reference's 42 FLOP — algorithmic work / time, not pipe utilisation: the packet kernel skips the barycentric half of
                        # most camera-round tests; frac_issue is what the VALU pipe really did: executed wave-instructions (SQ_INSTS_VALU of the
                        # committed profile) per second against one wave64 instruction per 2 clocks per SIMD at 2.4 GHz
                        "frac_algorithmic": round(tf / (FP32_PEAK_TFLOPS * world), 4),
                        "frac_issue": (round(issue_nominal, 4) if issue_nominal else None),
                        # ... and the same rate counting only what the kernel EXECUTES: the algorithmic fraction scaled by executed / full
                        # instructions per test (counter reading of the committed profile; null when that profile is stale) — FLOP the VALU really did
                        "frac_executed": (round(tf / (FP32_PEAK_TFLOPS * world) * min(1.0, insts["value"] / VALU_PER_TEST), 4) if insts.get("value") else None),
                        "rate_is": ("work of one launch / its HIP-event duration on rank 0 (x ranks) — for frac / frac_issue that work is the committed profile's replayed "
                                    "SQ_INSTS_VALU count, i.e. a counter of the profiled run over a duration of this run; frac_issue_profiled is the profile's own pair; "
                                    "wall-clock figures: achieved_wall, frac_wall" if one_launch
                                    else "whole-job wall clock (the timed region is several launches queued behind each other: their event intervals overlap)"),
                        "achieved_wall": round(tps_wall * FLOP_PER_TEST / 1e12, 2),
                        "frac_wall": round(tps_wall * FLOP_PER_TEST / 1e12 / (FP32_PEAK_TFLOPS * world), 4),
                        "traffic": traffic, "traffic_source": traffic_source,
                        "achieved_is": ("executed VALU wave-instructions per second (frac = frac_issue); frac_algorithmic / reference_equivalent_tflops count every DECIDED "
                                        "ray-triangle pair as the reference's 42 FLOP — the culls decide most of them without executing anything, so that rate exceeds the "
                                        "vector peak and is a statement about work avoided, not about the pipe" if (variant == 6 and issue_nominal) else
                                        ("reference-equivalent: every decided ray-triangle test counted as the reference's 42 FLOP (no current PMC profile to replay the executed "
                                         "instruction count from)" if variant == 6 else "executed: 42 FLOP per ray-triangle test")),
                        "ray_triangle_tests_per_s": round(tps, 1), "flop_per_test": FLOP_PER_TEST,
                        "valu_insts_per_test": insts,
                        # executed VALU instructions against the chip's issue limit: one wave64 VALU instruction per 2 clocks per SIMD, 1024 SIMDs
                        "issue_frac_of_nominal": (round(issue_nominal, 4) if issue_nominal else None),
                        # ONE provenance per fraction.  `frac` / frac_issue divide the profile's replayed instruction count by THIS run's HIP-event
                        # launch duration (a counter from one run over a duration from another: live, and not recomputable from profiles/ alone);
                        # frac_issue_profiled and issue_frac_at_profiled_clock come from the committed profile alone — its count / its own lone-launch
                        # duration (profiles/<round>_packets_pmc.json + _trace_kernel_stats.csv), at the nominal 2.4 GHz and at the clock that launch ran at
                        "frac_issue_profiled": (round(prof["valu_wave_insts_per_launch"] / (prof["kernel_avg_ns"] * 1e-9) / (1024 * 2.4e9 / 2), 4)
                                                if (variant == 6 and prof and prof.get("valu_wave_insts_per_launch") and prof.get("kernel_avg_ns")) else None),
                        "issue_frac_at_profiled_clock": (round(prof["valu_wave_insts_per_launch"] / (prof["kernel_avg_ns"] * 1e-9) / (1024 * sclk * 1e6 / 2), 4)
                                                         if (variant == 6 and sclk and prof and prof.get("valu_wave_insts_per_launch") and prof.get("kernel_avg_ns"))
                                                         else (round(issue_nominal * 2400.0 / sclk, 4) if (sclk and issue_nominal) else None)),
                        "profiled_clock_mhz": sclk,
                        # what the nominal limit is worth on this chip (round 6, measured: profiles/r06_valu_issue_probe.txt): the kernel's own triangle test, alone in
                        # a loop with its records in registers, issues one wave-instruction per 3.39 shader clocks per SIMD — 0.59 of the nominal 2 — because half its
                        # instructions belong to classes that take 4 clocks or more; clocks_per_wave_inst is the profiled launch's own figure for the whole kernel
                        "issue_yardstick": issue_yardstick(prof, sclk) if variant == 6 else None,
                        "lane_utilisation": (prof or {}).get("lane_utilisation"), "wave_time_split": (prof or {}).get("wave_time_split"),
                        "lds_busy": (prof or {}).get("lds_busy"), "salu_per_valu": (prof or {}).get("salu_per_valu"),
                        "note": "the brute-force intersect loop is FP32-VALU-bound; north_star's >= 70 % of the HBM roofline is unreachable for this "
                                "arithmetic intensity (hbm.frac below is the contract's figure: algorithmic bytes of one launch / its duration — 2 % in round 4, ~9 % now that the "
                                "culls have removed four fifths of the arithmetic: the 12-byte sample record per pixel is what is left of the bytes)",
                        "hbm": hbm}
        else:
            # BVH: a data-dependent walk has no closed-form operation count, and no memory unit binds it (hbm.frac is a fraction of a percent).  The
            # pipe it does load is the VALU's ISSUE port — executed wave-instructions (SQ_INSTS_VALU of the committed profile of this configuration,
            # replayed under the kernel-sha rule) against one wave64 instruction per 2 clocks per SIMD, 1024 SIMDs, 2.4 GHz — at a lane utilisation that
            # says how many of the 64 lanes of an issued instruction did something.  frac = frac_issue: live rate (replayed instructions per frame /
            # this run's wall clock per frame); frac_issue_profiled: the profile's own launch (instructions / its clean duration: recomputable from
            # profiles/<round>_<tag>_pmc.json + _trace_kernel_stats.csv).  The contract's HBM form stays nested under `hbm`.
            peak_issue = 1024 * 2.4e9 / 2 * world
            live = (valu_insts_per_frame * world / (elapsed / K)) if valu_insts_per_frame else None
            profiled = (prof["valu_wave_insts_per_launch"] / (prof["kernel_avg_ns"] * 1e-9)) if (prof and prof.get("valu_wave_insts_per_launch") and prof.get("kernel_avg_ns")) else None
            sclk = (prof or {}).get("profiled_clock_mhz")
            lane = (prof or {}).get("lane_utilisation")
            roofline = {"bound": "valu_issue", "achieved": (round(live / 1e9, 2) if live else None), "peak": round(peak_issue / 1e9, 1), "unit": "Gwave-inst/s",
                        "frac": (round(live / peak_issue, 4) if live else None),
                        "frac_issue": (round(live / peak_issue, 4) if live else None),
                        "frac_issue_profiled": (round(profiled / (1024 * 2.4e9 / 2), 4) if profiled else None),
                        "issue_frac_at_profiled_clock": (round(profiled / (1024 * sclk * 1e6 / 2), 4) if (profiled and sclk) else None),
                        "profiled_clock_mhz": sclk,
                        "lane_utilisation": lane,
                        "frac_useful_lanes": (round(live / peak_issue * lane, 4) if (live and lane) else None),
                        "wave_time_split": (prof or {}).get("wave_time_split"), "salu_per_valu": (prof or {}).get("salu_per_valu"),
                        "lds_busy": (prof or {}).get("lds_busy"), "lds_bank_conflict_share": (prof or {}).get("lds_bank_conflict_share"),
                        "l2_hit_rate": (prof or {}).get("l2_hit_rate"),
                        "traffic": traffic, "traffic_source": traffic_source,
                        "traffic_over_algorithmic": (round(traffic / algo_bytes, 3) if (traffic and algo_bytes) else None),
                        "write_over_algorithmic": (round(prof["write_bytes_per_launch"] * B / float(prof.get("frames_per_launch") or B) / (own_px * px_bytes * B), 3)
                                                   if (prof and prof.get("write_bytes_per_launch")) else None),
                        "rate_is": "executed VALU wave-instructions per frame (rocprof SQ_INSTS_VALU of the committed profile of this configuration) / this run's wall clock per frame",
                        "note": "BVH traversal (persistent kernel): bound by the length of a traversal step's dependent chain x the waves per SIMD available to hide it; "
                                "the pipe it loads is VALU issue (frac_issue) at lane_utilisation; node / triangle fetches are L2-resident and no memory unit is near a roof "
                                "(hbm.frac; traffic_over_algorithmic counts them against the 12 B/sample stores the byte model holds) — DESIGN.md 5.3, 6",
                        "hbm": hbm}
        out = {
            "metric": "Msamples/s (pixels x spp) at 1920x1080, 8-bounce",
            "value": round(msamples, 2),
            "unit": "Msamples/s",
            "n_gpus": world,
            "steps": K,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / K * 1e3, 5),
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": f"{args.scene} scene ({n_tris} triangles), {W}x{H}, {args.aa} spp, {args.bounces}-bounce "
                                   f"{'Kajiya' if args.mode == 9 else 'render mode %d' % args.mode}, {'default camera' if args.camera_mode == 0 else 'camera mode %d' % args.camera_mode}, {args.traversal} traversal"
                                   f"{' (LDS-staged)' if args.traversal == 'brute' else ''}, "
                                   f"{'packet kernel (camera / bounce rounds of 64 rays, LDS path queue; screen-rectangle cull in camera rounds, bounce cull in bounce rounds)' if variant == 6 else ('wide-tree kernel, scene in LDS (camera rays walk the 4-wide tree as wave-uniform packets, bounce rays per lane)' if variant == 11 else ('wide-tree kernel (the reference walk over the 4-wide regrouping of its tree)' if variant in (10, 12, 13) else ('regenerating' if not args.simple else 'one-pixel-per-lane') + ' wave64 kernel'))}",
                       "parallelism": f"tile{world}" + (" (TEST: all ranks share cuda:0, gloo, host-staged gather)" if shared_gpu else ""), "segments_per_sample": round(seg_per_sample, 4),
                       "grid_blocks": grid_blocks, "lds_bytes_per_block": lds_bytes, "frames_in_flight": in_flight,
                       "frames_per_dispatch": B_nominal, "frames_per_launch_timed": round(B, 3), "launches": timed_launches},
            "roofline": roofline,
        }
        # One meaning of `value` across rounds: `value` = the K timed steps only, as the bench contract words it (since round 2;
        # round 1's figure had the gather + untile of the finished frame inside the timed region, one frame per launch and no clock ramp).
        # The gather-inclusive figure stays available under a stable key, and the knobs that reproduce round 1's methodology are named.
        out["value_gather_inclusive"] = round(W * H * args.aa * K / (elapsed + gather_s) / 1e6, 2)
        out["methodology"] = {"timed_region": "K steps between two barriers; the one gather + untile of the finished frame is timed separately (frame_request)",
                              "since": "round 2", "round1_equivalent": "value_gather_inclusive with --batch 1 --ramp-seconds 0"}
        if world > 1 or use_dist:
            # "did RCCL see N ranks?" — asked of RCCL itself (rvpt_hip_comm_info: ncclCommCount / ncclCommUserRank / ncclGetVersion), not echoed from the launcher
            out["collective"] = {"path": collective_path, "rccl_ranks": rccl_ranks, "rccl_rank0": rccl_rank, "rccl_version": rccl_version, "gather_ms": round(gather_s * 1e3, 4),
                                 "note": "the one collective of the path: every rank's tile-linear accumulator to rank 0 (grouped ncclSend / ncclRecv inside the library) + un-tiling, "
                                         "once per frame request; `host-staged` = the library communicator was not available on every rank and the gather went through the "
                                         "host over torch.distributed (a test path: bench.py exits non-zero on it unless RVPT_BENCH_SHARED_GPU is set)"}
        out["frame_request"] = {"gather_ms": round(gather_s * 1e3, 4),
                                "value_with_one_gather_per_K_steps": out["value_gather_inclusive"],
                                "note": "one gather + untile of the finished frame after the K timed steps (device to device; no host copy)"}
        out["value_one_frame_per_launch"] = None if not one_n else {"value": round(W * H * args.aa * one_n / one_s / 1e6, 2), "unit": "Msamples/s", "frames": one_n, "ms_per_frame": round(one_s / one_n * 1e3, 5), "repetitions_ms_per_frame": [round(x / one_n * 1e3, 5) for x in one_reps],
                                             "note": "the same frames sent out one rvpt_hip_dispatch each, no wait in between (the reference's shape: one vkCmdDispatch per frame; "
                                                     "what a moving camera gets), wall clock between two barriers, median of five repetitions after a burst of 300 untimed launches, measured after the timed region; `value` batches K still-camera "
                                                     "frames per launch (rvpt_hip_dispatch_frames)"}
        out["clocks"] = {"profiled_clock_mhz": (prof or {}).get("profiled_clock_mhz"), "ramp_seconds": args.ramp_seconds, "ramp_frames_untimed": ramp_frames}
        if not args.no_cpu_baseline and world == 1:
            # Three legs inside the same budget (--cpu-seconds, default 12 s of oracle time): the workload's own traversal (the algorithm the GPU path
            # runs), the reference's LIVE algorithm when that is a different one (the BVH walk; the headline's brute force has no reference
            # counterpart: like-for-like against the reference is cpu_baseline_bvh), and BASELINE config 1 (256 x 256, both traversals).
            scene_args = (args, r.sorted_triangles, np.stack(r.local.materials), r.bvh_nodes, r.scene_camera.get_data())
            other = "bvh" if args.traversal == "brute" else None
            main_s = args.cpu_seconds * (0.45 if other else 0.9)
            out["cpu_baseline"] = cpu_baseline(*scene_args, main_s)
            if other:
                out["cpu_baseline_bvh"] = cpu_baseline(*scene_args, args.cpu_seconds * 0.45, traversal=other)
                out["cpu_baseline_bvh"]["note"] = ("the reference's live intersect path (intersection.glsl:489-517 -> :361-413) on the same frame: the like-for-like CPU "
                                                   "figure for the reference's ALGORITHM; cpu_baseline above times the brute-force loop the headline kernel runs")
            c1 = {t: cpu_baseline(*scene_args, args.cpu_seconds * 0.05, traversal=t, size=(256, 256)) for t in dict.fromkeys([args.traversal, "bvh"])}
            out["cpu_baseline_c1"] = {"config": "BASELINE config 1: 256x256, 1 spp", **{t: {k: v[k] for k in ("value", "unit", "cores", "kind", "min", "max", "sample")} for t, v in c1.items()}}
        line = json.dumps(out)
    else:
        line = None
    r.shutdown()
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    # RCCL writes a version banner through C stdio when a communicator is created; piped, that buffer would only be flushed at
    # exit — after the JSON line.  Flush it now so that the ONE JSON line is the last thing on stdout.
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass
    if line is not None:
        sys.stdout.flush()
        print(line, flush=True)
    if use_dist and world > 1 and collective_path != "rccl" and not shared_gpu:
        # a scaling record must not silently measure the wrong collective: the line above says which path ran; the exit code says it was not RCCL
        print(f"bench.py: {world} ranks but the frame gather did not run over the library's RCCL communicator ({collective_path})", file=sys.stderr)
        sys.exit(3)


if __name__ == "__main__":
    main()
