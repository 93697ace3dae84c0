#!/usr/bin/env python3
"""bench.py — Msamples/s of the path tracer's hot path on MI355X.

A step is one render of the reference's default scene at 1920x1080, 64 spp,
depth 5, seed 0 (BASELINE.json configs[1]) through pt_render of libpt_hip.so.
With N GPUs (one process per GPU, torch.distributed "nccl" = RCCL) rank r renders
rows r, r+N, ... of the same frame and the tiles are all-gathered over xGMI inside
the timed region (two frames in flight on alternating streams: the gather of frame
k overlaps the render of frame k+1, whose first waves also fill the compute units
that frame k's last waves leave idle), so total work is fixed: "strong" scaling.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

Started without a launcher (`python bench.py --gpus N`, no WORLD_SIZE in the environment) with N > 1, this process
starts the N ranks itself as fresh child processes before it touches the GPU, relays rank 0's line and exits non-zero
if any rank does.

Rank 0 prints one JSON line.  `roofline` prices the render kernel against the
FP32 vector peak with ALGORITHMIC flops (the event table in
oracle/pt_oracle.hpp, counted per config by tools/count_flops.py into
profiles/flops_per_sample.json); `cpu_baseline` is the CPU oracle timed on this
host on a bounded sample of the same frame.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP32_VALU_PEAK_TFLOPS = 157.3      # 256 CU x 4 SIMD x 32 lanes x 2 flop x 2.4 GHz
WIDTH, HEIGHT, SPP, DEPTH, SEED = 1920, 1080, 64, 5, 0
FLOPS_KEY = "default_1920x1080_64spp_depth5_seed0"
CPU_ROW_STEP = 1                   # cpu_baseline sample: every CPU_ROW_STEP-th row of the frame (1 = the whole frame)


def flops_per_sample():
    path = os.path.join(ROOT, "profiles", "flops_per_sample.json")
    with open(path) as f:
        return float(json.load(f)[FLOPS_KEY]["flops_per_sample"])


STORED_COUNTERS = os.path.join("profiles", "r03", "pmc_config2.json")


def stored_counters(kernel, kernel_ms):
    """What the committed rocprofv3 --pmc passes of this kernel on this frame say (STORED_COUNTERS; five separate
    passes, per-launch averages), priced with this run's kernel time: HBM bytes per launch (WRITE_SIZE + 2 x
    FETCH_SIZE, in KB; the gfx950 fetch counter reports half the bytes of wide reads), the FP32 rate the hardware
    executed (issue slots of FMA x 2 + MUL + ADD + TRANS, x 64 lanes), the vector issue rate beside the
    microbenchmark's rate for the same mix, lane utilisation and occupancy.  The counters are NOT collected by this
    run.  The stored file carries a hash of the kernels' sources and compiler flags (tools/pmc_summary.py:
    kernel_source_hash, written when the passes are collected): None when it is not today's hash, or when no pass of
    that kernel is stored — a stale block drops out of the line instead of describing another kernel."""
    path = os.path.join(ROOT, STORED_COUNTERS)
    if not os.path.exists(path):
        return None
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import pmc_summary
    with open(path) as f:
        pmc = json.load(f)
    if kernel not in pmc.get("profiled_kernel", "") or pmc.get("source_hash") != pmc_summary.kernel_source_hash():
        return None
    d = pmc_summary.derive(pmc, kernel_ms)
    return {"source": STORED_COUNTERS + " (stored counter passes of these kernel sources, hash %s; not collected by this run)" % pmc["source_hash"],
            "traffic": d["hbm_bytes"], "executed_tflops": d["executed_tflops"], "executed_frac": d["executed_frac"],
            "executed_tflops_active_lanes": d["executed_tflops_active_lanes"],
            "valu_issue_rate": d["valu_issue_rate"], "valu_issue_ceiling": d["valu_issue_ceiling"],
            "valu_issue_unit": "wave-instructions / ns / SIMD", "lane_util": d["lane_util"], "occupancy": d["occupancy"],
            "valu_wave_instructions": d["valu_wave_instructions"]}


def cpu_baseline(gpu_image):
    """The CPU oracle on every CPU_ROW_STEP-th row of the same frame, all host threads of this process's share.
    Also returns how far the GPU image is from the oracle on those rows."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np
    import oracle_binding as ob
    prims, n, mats, nm = ob.default_scene()
    sc = ob.Scene(prims, n, mats, nm)
    cfg = ob.make_config(WIDTH, HEIGHT, SPP, n_bounces=DEPTH, seed=SEED)
    rows = np.arange(0, HEIGHT, CPU_ROW_STEP, dtype=np.int32)
    threads = ob.default_threads()
    t0 = time.perf_counter()
    ref = sc.render(cfg, threads=threads, rows=rows)
    dt = time.perf_counter() - t0
    samples = len(rows) * WIDTH * SPP
    base = {"value": samples / dt / 1e6, "unit": "Msamples/s", "cores": threads, "kind": "port",
            "sample": "%s of the same 1920x1080 64-spp frame (%d rows, %.1f Msamples, %.1f s)"
                      % ("every row" if CPU_ROW_STEP == 1 else "rows j %% %d == 0" % CPU_ROW_STEP, len(rows), samples / 1e6, dt)}
    img = gpu_image.cpu().numpy()[rows].astype(np.float64)
    ref = ref[rows].astype(np.float64)
    parity = {"pinned_by": "the CPU restatement only: the reference's own fixtures pin hit counts, nothing else (parity unpinned beyond hit counts)",
              "rows": int(len(rows)),
              "rmse_rel": float(np.sqrt(((img - ref) ** 2).mean(axis=(0, 1))).max() / ref.mean()),
              "bad_pixel_frac": float((np.abs(img - ref).max(axis=2) > 0.05).mean()),
              "mean_rel": float((np.abs(img.mean(axis=(0, 1)) - ref.mean(axis=(0, 1))) / ref.mean(axis=(0, 1))).max())}
    return base, parity


def launch_ranks(n, command, extra_env=None, out=None):
    """Starts `command` n times as child processes, one per rank (RANK, LOCAL_RANK, WORLD_SIZE, MASTER_ADDR,
    MASTER_PORT set as torch.distributed.run sets them), waits for all, writes rank 0's standard output to `out`
    (default: this process's) and returns 0 if every rank exited 0, else the first non-zero exit status (a rank
    killed by signal s counts as 128 + s).  When a rank fails, the ranks still running are ended (they would wait
    in a collective forever).  The caller must not have initialised the GPU: the children are fresh processes, this
    one only waits."""
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ)
        env.update({"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(n), "LOCAL_WORLD_SIZE": str(n),
                    "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port)})
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        env.update(extra_env or {})
        procs.append(subprocess.Popen(command, env=env, stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    status = 0
    pending = set(range(n))
    rank0_out = b""

    def drain():
        nonlocal rank0_out
        rank0_out = procs[0].stdout.read()
    t = threading.Thread(target=drain, daemon=True)
    t.start()
    while pending:
        for r in sorted(pending):
            rc = procs[r].poll()
            if rc is None:
                continue
            pending.discard(r)
            if rc != 0 and status == 0:
                status = rc if rc > 0 else 128 - rc
                for q in pending:
                    procs[q].terminate()
        time.sleep(0.05)
    t.join(timeout=10)
    (out or sys.stdout.buffer).write(rank0_out)
    (out or sys.stdout.buffer).flush()
    return status


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--math", choices=["fast", "parity", "scalar", "lds"], default="fast")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--general-kinds", action="store_true",
                    help="upload the scene with the general record kinds only (what a scene with rotated shapes gets); not the headline")
    ap.add_argument("--no-overlap", action="store_true", help="N > 1: gather each frame before rendering the next")
    ap.add_argument("--gather", choices=["torch", "abi"], default="torch",
                    help="N > 1: torch's all_gather_into_tensor + strided copy (default), or the C ABI's own collective "
                         "(pt_comm_init_rank + pt_gather_framebuffer), its id handed out through the torch process group")
    ap.add_argument("--tile-log2", type=int, default=-1, help="pixels per wave as a power of two (development aid; default: this script's choice)")
    ap.add_argument("--emulate-world", type=int, default=0,
                    help="development aid on one GPU: render rank 0's share of an N-GPU run with the N-GPU pipeline, no collectives")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # no launcher: be one.  Nothing above has touched the GPU (torch is not even imported yet).
        raise SystemExit(launch_ranks(args.gpus, [sys.executable, os.path.abspath(__file__)] + sys.argv[1:]))

    import torch
    import torch.distributed as dist
    import pathtracer_amd as pt
    from pathtracer_amd import multigpu

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    emulate = args.emulate_world if world == 1 else 0
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # PT_BENCH_REHEARSE=1: every rank on GPU 0 and the collectives through gloo, so that the whole N-rank flow (launcher,
    # partition, gather, pipeline, bitwise check, reductions) can run on a box with one GPU.  Its rates mean nothing.
    rehearse = os.environ.get("PT_BENCH_REHEARSE") == "1"
    if rehearse:
        local_rank = 0
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d (launch N>1 with torch.distributed.run)" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py: no GPU visible; the hot path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearse:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)

    debug = os.environ.get("PT_BENCH_DEBUG") == "1"

    def note(msg):
        if debug:
            print("[bench rank %d] %s" % (rank, msg), file=sys.stderr, flush=True)
    note("process group ready")
    prims, n, mats, nm = pt.default_scene()
    scene = pt.Scene(prims, n, mats, nm, device=local_rank, general_kinds=pt.PT_GENERAL_ALL if args.general_kinds else 0)
    math_mode = {"fast": pt.PT_MATH_FAST, "parity": pt.PT_MATH_PARITY, "scalar": pt.PT_MATH_FAST_SCALAR, "lds": pt.PT_MATH_FAST_LDS}[args.math]
    # rank r renders rows r, r+N, ... into a compact tile; all-gather + strided copy rebuild the image
    part = emulate or world              # ranks the frame is split over
    my_rows = len(multigpu.rank_rows(HEIGHT, rank, part))
    # PT_BENCH_FORCE_OVERLAP=1 exercises the pipeline on one GPU (development aid)
    overlap = (part > 1 or os.environ.get("PT_BENCH_FORCE_OVERLAP") == "1") and not args.no_overlap
    # Launches in flight on alternating streams fill each other's tails, so larger tiles than a lone launch wants (less
    # drain per wave) pay: measured per rank share of this frame with 32 pixels per wave / the library's own choice
    # (16 or 8 pixels below 0.9 Mpixel): N=2 1.343 / 1.341 ms, N=4 0.690 / 0.729, N=8 0.361 / 0.378 (64 pixels: 1.468
    # at N=2).  The image does not depend on the tile size.
    tile_log2 = 5 if overlap else 0
    if args.tile_log2 >= 0:
        tile_log2 = args.tile_log2
    cfg = pt.make_config(WIDTH, HEIGHT, SPP, n_bounces=DEPTH, seed=SEED, row_begin=rank, row_end=HEIGHT, row_step=part,
                         fb_compact=1, math_mode=math_mode, tile_log2=tile_log2)
    comm = None
    if args.gather == "abi" and world > 1:
        if rehearse:
            raise SystemExit("bench.py: --gather abi needs one GPU per rank (RCCL refuses two ranks on one device)")
        comm = pt.Comm(multigpu.exchange_comm_id(rank, pt.comm_unique_id), rank, world, device=local_rank)
        note("pt_comm ready")
    frames = [multigpu.TiledFrame(HEIGHT, WIDTH, part, device, collective=not emulate, comm=comm) for _ in range(2 if overlap else 1)]
    main_stream = torch.cuda.current_stream()
    comm_stream = torch.cuda.Stream(device=device) if overlap else None
    # With frames in flight, consecutive frames render on alternating streams: the next frame's first waves fill
    # the compute units the previous frame's last waves leave idle (a rank's share of the frame is a small launch).
    render_streams = [torch.cuda.Stream(device=device), torch.cuda.Stream(device=device)] if overlap else [main_stream]
    gathered_ev = [None, None]

    def step(k, ev_pair=None):
        """One frame: render this rank's rows, then gather.  With two frames in flight the gather of frame k runs on
        a second stream while frame k+1 renders; frame k's buffers are reused by frame k+2 only after its gather."""
        f = frames[k % len(frames)]
        rs = render_streams[k % len(render_streams)]
        if overlap and gathered_ev[k % 2] is not None:
            rs.wait_event(gathered_ev[k % 2])
        with torch.cuda.stream(rs):
            if ev_pair:
                ev_pair[0].record(rs)
            if my_rows:
                scene.render(cfg, out=f.local)
            if ev_pair:
                ev_pair[1].record(rs)
        if not overlap:
            f.gather()
            return
        done = torch.cuda.Event()
        done.record(rs)
        with torch.cuda.stream(comm_stream):
            comm_stream.wait_event(done)
            f.gather()
            g = torch.cuda.Event()
            g.record(comm_stream)
            gathered_ev[k % 2] = g

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    if overlap:
        for rs in render_streams:
            rs.wait_stream(main_stream)      # scene upload and buffer fills were issued on the current stream
    for k in range(args.warmup):
        step(k)
    note("warm-up issued")
    fence()
    note("warm-up done")
    # Kernel time: HIP events on the stream pt_render launches on (torch's current stream), around the
    # render launch only, inside the timed region.
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(args.warmup + k, ev[k])
    fence()
    elapsed = time.perf_counter() - t0
    note("timed steps done")
    kernel_ms = sum(a.elapsed_time(b) for a, b in ev) / args.steps
    image = frames[(args.warmup + args.steps - 1) % len(frames)].image

    # One frame at a time (render, then gather, nothing in flight beside it): what a single frame costs end to end.
    single_ms = None
    gather_ms = None
    bitwise = None
    if part > 1:
        n_single = max(1, min(args.steps, 10))
        f0 = frames[0]
        cfg_single = pt.make_config(WIDTH, HEIGHT, SPP, n_bounces=DEPTH, seed=SEED, row_begin=rank, row_end=HEIGHT, row_step=part,
                                    fb_compact=1, math_mode=math_mode)      # tile size: the library's choice for a lone launch
        fence()
        ts = time.perf_counter()
        gev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n_single)]
        for i in range(n_single):
            if my_rows:
                scene.render(cfg_single, out=f0.local)
            gev[i][0].record()
            f0.gather()
            gev[i][1].record()
            torch.cuda.synchronize()
        fence()
        single_ms = (time.perf_counter() - ts) / n_single * 1e3
        gather_ms = sum(a.elapsed_time(b) for a, b in gev) / n_single     # collective + reordering, on the device
        note("single frames done")
    if world > 1:
        # Rank 0 renders the whole frame alone (untimed) and compares the gathered image with it, bit for bit.
        flag = torch.ones(1, dtype=torch.int32, device="cpu" if rehearse else device)
        if rank == 0:
            alone = scene.render(pt.make_config(WIDTH, HEIGHT, SPP, n_bounces=DEPTH, seed=SEED, math_mode=math_mode))
            flag[0] = 1 if torch.equal(alone, image) else 0
        dist.broadcast(flag, src=0)
        bitwise = bool(int(flag[0]))
        note("bitwise check done")

    t = torch.tensor([elapsed, kernel_ms, single_ms or 0.0, gather_ms or 0.0], dtype=torch.float64, device="cpu" if rehearse else device)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed, kernel_ms = float(t[0]), float(t[1])
    if single_ms is not None:
        single_ms, gather_ms = float(t[2]), float(t[3])

    if rank == 0:
        total_samples = WIDTH * HEIGHT * SPP
        fps = flops_per_sample()
        if emulate:
            total_samples = my_rows * WIDTH * SPP     # one rank's share: the line is not a whole-job figure
        tile_samples = my_rows * WIDTH * SPP
        # Frames in flight render concurrently (alternating streams), so one launch's event-to-event time overlaps
        # its neighbour's; the launch rate is then what prices the kernel: algorithmic flops per launch / time per step.
        launch_ms = elapsed / args.steps * 1e3 if overlap else kernel_ms
        achieved = fps * tile_samples / (launch_ms * 1e-3) / 1e12
        kernel = {"fast": "render_fast_batched_kernel", "scalar": "render_fast_kernel", "lds": "render_fast_tiled_kernel",
                  "parity": "render_parity_kernel"}[args.math]
        counters = stored_counters(kernel, kernel_ms) if world == 1 and not emulate and not args.general_kinds else None
        out = {
            "metric": "Msamples/sec at 1080p 64spp",
            "value": total_samples * args.steps / elapsed / 1e6,
            "unit": "Msamples/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "reference default scene (main.cu:73-195), 1920x1080, 64 spp, depth 5, seed 0"
                                   + (" [general record kinds only: not the headline]" if args.general_kinds else ""),
                       "math_mode": args.math, "partition": "rows r, r+%d, ... per rank, all-gather + strided copy%s%s" % (part, ", gather of frame k overlapped with render of frame k+1, frames on alternating streams" if overlap else "", (" [EMULATED rank 0 of %d on one GPU, no collectives]" % emulate if emulate else "") + (" [REHEARSAL: all ranks on one GPU, gloo]" if rehearse else "")),
                       "flops_per_sample": fps},
            "roofline": {"bound": "valu_fp32", "kernel": kernel,
                         "achieved": achieved, "peak": FP32_VALU_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / FP32_VALU_PEAK_TFLOPS,
                         "achieved_is": "algorithmic flops (event table of the reference's arithmetic) / kernel time; executed_* below is what the hardware issued",
                         "traffic": counters["traffic"] if counters else None,
                         "kernel_ms": kernel_ms, "kernel_msamples_per_s": tile_samples / launch_ms / 1e3,
                         "launches_overlap": bool(overlap)},
        }
        if counters:
            out["roofline"].update({k: v for k, v in counters.items() if k != "traffic"})
        if part > 1:
            # the rate above is a pipeline rate: frames_in_flight frames render and gather concurrently; the time of
            # one frame rendered and gathered on its own is given beside it
            out["pipelined"] = bool(overlap)
            out["frames_in_flight"] = len(frames)
            out["single_frame_ms"] = single_ms
            out["gather"] = "pt_gather_framebuffer (C ABI: ncclAllGather + reordering kernel)" if comm is not None else "torch all_gather_into_tensor + strided copy"
            out["gather_ms"] = gather_ms
            out["bitwise_equal_to_single_gpu"] = bitwise
        if world == 1 and not emulate and not args.no_cpu_baseline:
            out["cpu_baseline"], out["parity_vs_oracle"] = cpu_baseline(image)
        print(json.dumps(out), flush=True)
    if comm is not None:
        torch.cuda.synchronize()
        comm.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
