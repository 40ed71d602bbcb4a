#!/usr/bin/env python3
"""Benchmark of the waldboost detection hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B]

A *step* is one pass of the hot path -- octave pyramid, fused grad_hist channel pyramid and the
dense 128-stage depth-2 cascade scan (BASELINE.json configs[1]) -- over one batch of B synthetic
1920x1080 uint8 images already resident in HBM, replayed from one hipGraph.  Consecutive steps
cycle through a pool of different resident images.  With N>1 there is one rank per GPU: either
the caller starts them (torch.distributed.run; RANK / WORLD_SIZE in the environment) or, when
WORLD_SIZE is not set, bench.py starts torch.distributed.run itself -- before torch is imported
or the GPU touched -- and exits with its code (the reference's own parallel idiom is a process pool
over images, scripts/waldboost-detect.py:64-67).  Every rank runs the same per-GPU workload on its
own images ("weak" scaling: BASELINE configs[3] is --gpus 8 --batch 64) and every round of steps
ends with the RCCL all-gather of the fixed-size detection prefixes, issued on a side stream so
that it overlaps the next round's kernels.

Rank 0 prints ONE JSON line: candidate windows/s (whole job), plus
  roofline     -- the dominant kernel's algorithmic HBM bytes per launch / its average launch
                  duration (HIP events on the launch stream) against the 8 TB/s HBM peak
  cpu_baseline -- the NumPy oracle (CPU restatement of the reference) on a bounded sample of
                  the same workload, multiprocessing.Pool over images like the reference's
                  scripts/waldboost-detect.py:64-67
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

MODELS = {"grad_hist": os.path.join(ROOT, "tests", "golden", "models", "cfg2_d2_T128.pb"),
          # secondary workload: the reference's integer channels, uint8 x4 per pixel
          "grad_hist_4_u1": os.path.join(ROOT, "tests", "golden", "models", "cfg2_gh4u1_d2_T128.pb")}
MODEL = MODELS["grad_hist"]
# --config: "2" = BASELINE configs[1] (the headline: what the metric is quoted on; configs[2] / [3] are --batch 64 /
# --gpus 8 of it), "5" = BASELINE configs[4] (4K, shrink 4 -- this build's extension of the reference's shrink in (1, 2),
# channels.py:120 --, 12 levels per octave, 256 stages, ~1e-4 survival)
WORKLOADS = {
    "2": dict(H=1080, W=1920, name="BASELINE configs[1]", models=MODELS, batch=1, stages=128, cpu_images_per_core=6),
    "5": dict(H=2160, W=3840, name="BASELINE configs[4]", batch=4, stages=256, cpu_images_per_core=1,
              models={"grad_hist": os.path.join(ROOT, "tests", "golden", "models", "cfg5_d2_T256.pb")}),
}
H, W = 1080, 1920
HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)
CFG3_IMAGES, CFG3_BATCH, CFG3_SEED0 = 512, 64, 30000      # BASELINE configs[3]: 512 x 1080p, 64 per GPU and launch


# ----------------------------------------------------------------------------- CPU baseline
def _cpu_worker(job):
    seed, model_path, h, w = job
    import waldboost_amd as wb          # host-side .pb reader only; no GPU is touched here
    from util import oracle_detect
    from waldboost_amd.synth import synth_image
    M = wb.load(model_path)
    t0 = time.perf_counter()
    res = oracle_detect(M, synth_image(h, w, seed))
    return res["n_loc"], time.perf_counter() - t0


def cpu_baseline(model_path, images_per_core=6, max_cores=16, h=1080, w=1920):
    import multiprocessing as mp
    cores = max(1, min(max_cores, os.cpu_count() or 1))
    n = cores * images_per_core
    ctx = mp.get_context("fork")
    t0 = time.perf_counter()
    with ctx.Pool(cores) as pool:
        out = pool.map(_cpu_worker, [(i, model_path, h, w) for i in range(n)])
    wall = time.perf_counter() - t0
    windows = sum(o[0] for o in out)
    return {"value": windows / wall, "unit": "windows/s", "cores": cores, "kind": "port",
            "sample": f"{n} synthetic {w}x{h} images, Pool({cores}) over images, NumPy oracle, {wall:.1f} s wall",
            "single_core_windows_per_s": float(np.mean([o[0] / o[1] for o in out]))}


def _oracle_worker(job):
    seed, model_path, h, w, stages = job
    import waldboost_amd as wb
    from util import oracle_detect
    from waldboost_amd.synth import synth_image
    M = wb.load(model_path)
    if stages:
        M.classifier, M.theta = M.classifier[:stages], M.theta[:stages]
    ref = oracle_detect(M, synth_image(h, w, seed))
    return seed, {k: ref[k] for k in ("level", "r", "c", "scores", "alive", "n_loc", "n_weak")}


def oracle_refs(seeds, model_path, h, w, stages=0):
    """The oracle's detections for the images the parity gates will look at, computed side by side in a process pool
    before anything touches the GPU (a 4K image takes the NumPy oracle half a minute)."""
    import multiprocessing as mp
    seeds = sorted(set(seeds))
    if not seeds:
        return {}
    with mp.get_context("fork").Pool(min(len(seeds), max(1, os.cpu_count() or 1))) as pool:
        return dict(pool.map(_oracle_worker, [(sd, model_path, h, w, stages) for sd in seeds]))


def _synth_worker(job):
    from waldboost_amd.synth import synth_image
    return synth_image(job[1], job[2], job[0])


def synth_shard(lo, hi, h, w, procs):
    """Images lo..hi-1 of the configs[3] batch (seeds CFG3_SEED0 + index), uint8 [hi - lo, h, w], generated by a small
    process pool before anything touches the GPU."""
    import multiprocessing as mp
    if hi <= lo:
        return np.zeros((0, h, w), np.uint8)
    jobs = [(CFG3_SEED0 + i, h, w) for i in range(lo, hi)]
    if procs <= 1:
        return np.stack([_synth_worker(j) for j in jobs])
    with mp.get_context("fork").Pool(procs) as pool:
        return np.stack(pool.map(_synth_worker, jobs, chunksize=2))


# ----------------------------------------------------------------------------- helpers
PROFILE_DIR = next((d for d in (os.path.join(ROOT, "profiles", r) for r in ("r04", "r03", "r02"))
                    if os.path.exists(os.path.join(d, "sq_counters.json"))), os.path.join(ROOT, "profiles", "r02"))
N_SIMD, CLOCK_GHZ = 256 * 4, 2.4          # MI355X_MICROARCH.md: 256 CUs x 4 SIMDs, 2.4 GHz max clock


def issue_bound(roof, batch, channels):
    """The bound that binds these kernels: vector-instruction issue.  Measured on this GPU
    (tools/valu_class_probe.hip, profiles/r02/valu_issue_classes.txt), at 4 waves per SIMD a wave64 VALU instruction
    holds its SIMD for ~2.5 cycles if it is a plain fp32 add / multiply / fmac (or a 32-bit add / and / or / shift-right /
    move) on vector-register operands, and for ~4.3 cycles otherwise (compares, selects, min / max, conversions, fp64,
    anything with a scalar-register operand).  From the committed SQ counters (three rocprofv3 --pmc passes,
    instructions per image):
        floor_us = (fp32 add + mul + fma instructions x 2.5 + all other VALU instructions x 4.3) cycles
                   / (1024 SIMDs x 2.4 GHz)
    -- a LOWER estimate of the issue time (integer adds / logic ops of the fast class are counted as fast only when the
    counters single them out: they do not, so they are priced at 4.3; fp32 operations with a scalar operand are priced
    at 2.5).  frac = floor / measured.  wave_time = where a wave's lifetime goes, from the same counters: issuing
    instructions, ready but waiting for an issue slot, waiting on memory / LDS / a barrier."""
    path = os.path.join(PROFILE_DIR, "sq_counters.json")
    if roof is None or channels != "grad_hist" or not os.path.exists(path) or roof.get("config", "2") != "2":
        return None
    with open(path) as f:
        k = json.load(f).get(roof["kernel"])
    if not k:
        return None
    g = lambda name: k.get(name + "_per_image", 0.0) * batch
    valu, salu = g("SQ_INSTS_VALU"), g("SQ_INSTS_SALU")
    fast = g("SQ_INSTS_VALU_ADD_F32") + g("SQ_INSTS_VALU_MUL_F32") + g("SQ_INSTS_VALU_FMA_F32")
    floor_us = (fast * 2.5 + (valu - fast) * 4.3) / (N_SIMD * CLOCK_GHZ * 1e3)
    wc = g("SQ_WAVE_CYCLES")
    # straight from the counters, no model: the share of the CUs' busy time in which a SIMD's VALU pipe executes (both
    # counters in quad-cycles per CU; 4 SIMDs per CU) and what one wave64 VALU instruction occupies it for
    busy, act = g("SQ_BUSY_CU_CYCLES"), g("SQ_ACTIVE_INST_VALU")
    valu_busy = act / busy if busy else None
    cycles_per_valu = 4.0 * act / valu if valu else None
    wave_time = None
    if wc:
        wave_time = {"issuing": g("SQ_ACTIVE_INST_ANY") / wc, "waiting_for_issue_slot": g("SQ_WAIT_INST_ANY") / wc,
                     "waiting_on_data_or_barrier": g("SQ_WAIT_ANY") / wc}
    # the same figure from TIME instead of the busy counter (SQ_ACTIVE_INST_VALU is nearly SQ_INSTS_VALU in quad-cycles, so
    # the counter-derived 4.05 is partly an identity of units): the measured launch time x SIMDs x clock / instructions --
    # SIMD-cycles that ELAPSED per wave64 VALU instruction issued, waits included
    time_cycles_per_valu = (roof["avg_launch_ms"] * 1e-3 * N_SIMD * CLOCK_GHZ * 1e9 / valu) if valu else None
    return {"source": f"{os.path.relpath(path, ROOT)} (committed rocprofv3 --pmc SQ passes; instruction counts are NOT measured in this "
                      "run, only measured_us is)",
            "kernel": roof["kernel"], "bound": "valu_issue", "valu_insts_per_launch": valu, "of_which_fp32_add_mul_fma": fast,
            "salu_insts_per_launch": salu, "floor_us": floor_us, "measured_us": roof["avg_launch_ms"] * 1e3,
            "frac": floor_us / (roof["avg_launch_ms"] * 1e3), "valu_busy": valu_busy, "cycles_per_valu": cycles_per_valu,
            "time_cycles_per_valu": time_cycles_per_valu,
            "wave_time": wave_time, "counters": os.path.relpath(path, ROOT),
            "note": "2.5 issue cycles per wave64 fp32 add/mul/fma, 4.3 per other VALU instruction (measured classes); "
                    "1024 SIMDs at 2.4 GHz"}


def event_time_ms(fn, iters, torch):
    """Average duration of fn() over `iters` back-to-back launches, HIP events on the launch stream."""
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def self_launch(n, argv):
    """`python bench.py --gpus N` with no launcher around it: start N ranks with torch.distributed.run (one process
    per GPU, rendezvous on 127.0.0.1 at a free port), pass the flags through, wait, return its exit code.  Runs before
    torch is imported and before anything touches the GPU -- the ranks are fresh child processes."""
    import socket
    import subprocess
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # (dmabuf IPC: what RCCL needs on this driver)
    env.setdefault("OMP_NUM_THREADS", "1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    return subprocess.run(cmd, env=env).returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", choices=sorted(WORKLOADS), default="2",
                    help="2: BASELINE configs[1] (1080p, shrink 2, 8 per octave, 128 stages -- the headline metric's workload); "
                         "5: BASELINE configs[4] (4K, shrink 4, 12 per octave, 256 stages; 4 images per step by default)")
    ap.add_argument("--batch", type=int, default=0, help="images per step and GPU (default: 1 for --config 2 = configs[1]; "
                                                          "configs[2] is --batch 64; 4 for --config 5)")
    ap.add_argument("--no-config3", action="store_true",
                    help="skip the configs[3] measurement (512 x 1080p cut over the ranks by shard_range, 64 images per launch, "
                         "through distributed.detect_sharded with its end-of-batch exchange) that every line carries")
    ap.add_argument("--config3-images", type=int, default=CFG3_IMAGES, help="diagnostic: size of the configs[3] batch")
    ap.add_argument("--pool", type=int, default=4, help="distinct resident image batches cycled through")
    ap.add_argument("--streams", type=int, default=4,
                    help="HIP streams the steps are spread over (<= pool): consecutive steps work on different "
                         "images, so their kernels may overlap on the GPU like frames of a video pipeline")
    ap.add_argument("--repeats", type=int, default=10,
                    help="the timed region of --steps steps is run this many times; `value` is the median region, "
                         "`value_spread` carries min/max")
    ap.add_argument("--warm-ms", type=float, default=50.0,
                    help="untimed device work in front of the timed regions, in ms (as whole regions of --steps steps): the "
                         "GPU's clocks ramp for ~20 ms after the idle seconds of the parity gates")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-through-api", action="store_true", help="skip the host ndarray -> Boxes measurement (Model.detect)")
    ap.add_argument("--channels", choices=sorted(MODELS), default="grad_hist",
                    help="channel function of the workload; grad_hist is the BASELINE config, grad_hist_4_u1 the "
                         "reference's integer channels (same geometry, uint8 channels, its own calibrated cascade)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N>1 (nccl = RCCL; gloo only to "
                                                      "rehearse the multi-rank loop with several ranks on one GPU)")
    ap.add_argument("--force-collective", action="store_true",
                    help="diagnostic: run the multi-rank step (pack + all_gather of the detection prefixes on a side stream) "
                         "with a single rank, to measure what the collective path costs per step")
    ap.add_argument("--stages", type=int, default=0, help="diagnostic: keep only the first N stages of the cascade")
    ap.add_argument("--no-graph", action="store_true", help="eager launches instead of hipGraph replay")
    ap.add_argument("--region-graph", choices=["auto", "on", "off"], default="auto",
                    help="capture the K steps of a timed region into one hipGraph per stream (n_streams launches per region) "
                         "instead of one replay per step; auto: when K <= 64 -- a short region is dominated by the K launches "
                         "and the staggered start of the streams (K=20: +5 %%), a long one amortises both")
    ap.add_argument("--only", choices=["all", "channels", "cascade", "octaves"], default="all",
                    help="profile helper: launch only one kernel group in the timed loop")
    ap.add_argument("--no-jit", action="store_true", help="stay on the generic cascade kernel (stage records through the scalar cache)")
    ap.add_argument("--dry-launch", action="store_true",
                    help="start the ranks, rendezvous (gloo on the CPU), let rank 0 print what it would run, stop: checks "
                         "the launch path on a machine without GPUs")
    args = ap.parse_args()

    global H, W
    wl = WORKLOADS[args.config]
    H, W = wl["H"], wl["W"]
    if args.channels not in wl["models"]:
        raise SystemExit(f"--config {args.config} has no {args.channels} model")
    model_path = wl["models"][args.channels]
    if not args.batch:
        args.batch = wl["batch"]
    want_cpu = not args.no_cpu_baseline and args.only == "all" and not args.dry_launch
    if args.no_jit:
        os.environ["WB_CASC_JIT"] = "0"                       # (also the engine's own after-a-few-scans policy; children inherit it)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # the CPU baseline of a self-launched multi-rank run: timed HERE, on the otherwise idle host, before any rank
        # exists, and handed to rank 0 through the environment (a multi-rank line carries it like a single-rank one)
        if want_cpu and os.path.exists("/dev/kfd"):
            os.environ["WB_BENCH_CPU_BASELINE"] = json.dumps(
                cpu_baseline(model_path, images_per_core=wl["cpu_images_per_core"], h=H, w=W))
        raise SystemExit(self_launch(args.gpus, sys.argv[1:]))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: the launcher's world size and --gpus must agree")
    # stdout carries rank 0's ONE JSON line and nothing else: whatever a library writes to file descriptor 1 meanwhile
    # (gloo prints a connection banner there) goes to stderr
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    def emit(obj):
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(obj) + "\n").encode())
    if args.dry_launch:
        import torch
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("gloo", rank=rank, world_size=world)
        seen = torch.zeros(world, dtype=torch.int64)
        dist.all_gather_into_tensor(seen, torch.tensor([rank * 10 + local_rank], dtype=torch.int64))
        if rank == 0:
            emit({"dry_launch": True, "n_gpus": args.gpus, "world_size_reported": dist.get_world_size(),
                  "ranks": [int(x) // 10 for x in seen], "local_ranks": [int(x) % 10 for x in seen],
                  "backend": args.backend, "steps": args.steps, "warmup": args.warmup, "batch_per_gpu": args.batch})
        dist.destroy_process_group()
        return

    if not os.path.exists("/dev/kfd"):                       # (no ROCm device node: fail before the minutes of host-side preparation)
        raise SystemExit("bench.py needs a GPU (no /dev/kfd on this machine); --dry-launch checks the launch path without one")
    cpu = None
    if rank == 0 and want_cpu:
        if os.environ.get("WB_BENCH_CPU_BASELINE"):
            cpu = json.loads(os.environ["WB_BENCH_CPU_BASELINE"])      # (timed by the launching parent, see above)
        else:
            # before the GPU is initialised (fork-safe); with a launcher's ranks around it (torch.distributed.run started
            # by the caller) the other ranks are generating their configs[3] images meanwhile: said in `sample`
            cpu = cpu_baseline(model_path, images_per_core=wl["cpu_images_per_core"], h=H, w=W)
            if world > 1:
                cpu["sample"] += f"; timed on rank 0 while {world - 1} other rank(s) prepared their inputs on the same host"
    # BASELINE configs[3]'s batch: THIS rank's contiguous shard of the 512 images, generated now (process pool, before
    # anything touches the GPU)
    cfg3_host = None
    do_cfg3 = args.config == "2" and args.channels == "grad_hist" and args.only == "all" and not args.no_config3 and not args.stages
    if do_cfg3:
        from waldboost_amd.distributed import shard_range
        lo3, hi3 = shard_range(args.config3_images, rank, world)
        t_gen = time.perf_counter()
        cfg3_host = synth_shard(lo3, hi3, H, W, max(1, min(16, (os.cpu_count() or 1) // max(world, 1))))
        t_gen = time.perf_counter() - t_gen

    # the oracle's answers for the images the parity gates check (first and last image of the first and the last engine's
    # batch), computed now, in parallel
    gate_refs = {}
    if args.only == "all" and not args.stages:
        Pn, Bn = max(1, args.pool), args.batch
        gate_refs = oracle_refs([(rank * Pn + j) * Bn + b for j in {0, min(Pn, args.steps) - 1} for b in {0, Bn - 1}],
                                model_path, H, W)

    import torch
    import torch.distributed as dist
    n_dev = torch.cuda.device_count()
    if n_dev == 0:
        raise SystemExit("bench.py needs a GPU (torch.cuda.device_count() == 0); --dry-launch checks the launch path without one")
    if world > n_dev and args.backend == "nccl":
        raise SystemExit(f"--gpus {world} with backend nccl (RCCL) needs {world} GPUs, this node shows {n_dev}; "
                         "--backend gloo rehearses the multi-rank loop with several ranks on one GPU")
    local_dev = local_rank % n_dev
    torch.cuda.set_device(local_dev)
    coll = world > 1 or args.force_collective                # the pack + all_gather path
    if coll:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_dev), rank=rank, world_size=world)
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world)

    import waldboost_amd as wb
    from waldboost_amd import _native as nat
    from waldboost_amd.engine import PyramidEngine, capturing
    from waldboost_amd.synth import synth_image
    from waldboost_amd.distributed import RoundGatherer

    M = wb.load(model_path)
    spec = wb.channels.channel_spec(M.channel_opts["channels"])
    shrink, n_per_oct, smooth = (int(M.channel_opts[k]) for k in ("shrink", "n_per_oct", "smooth"))
    if args.stages:
        M.classifier, M.theta = M.classifier[:args.stages], M.theta[:args.stages]
    dm = M.device_cascade()
    # the model-specialised cascade kernel (hiprtc, the stage records as constants): built up front, outside every
    # timed region -- what a caller of Model.detect gets by itself after a few scans (engine.py: _JIT_AFTER)
    jit = False
    if os.environ.get("WB_CASC_JIT", "1") != "0" and not args.no_jit:
        t_jit = time.perf_counter()
        try:
            jit = bool(dm.specialize())
        except Exception as exc:                              # (a compiler failure leaves the generic kernel: say so, go on)
            print(f"bench: model specialisation failed, staying on the generic cascade kernel: {exc}", file=sys.stderr)
        t_jit = time.perf_counter() - t_jit
        if not jit:                                           # (refused: no build passed the self-test against the generic kernel)
            from waldboost_amd import _native as _nat
            print(f"bench: no specialised cascade kernel for this model, the line is timed on the generic one: {_nat.last_error()}", file=sys.stderr)
    B, P = args.batch, max(1, args.pool)
    engines = []
    for i in range(P):
        e = PyramidEngine(H, W, np.uint8, shrink, n_per_oct, smooth, batch=B, det_capacity=16384 * B, channels=spec)
        seeds = [(rank * P + i) * B + b for b in range(B)]
        e.load_images(np.stack([synth_image(H, W, s) for s in seeds]))
        engines.append(e)
    plan = engines[0].plan
    n_loc = plan.n_loc(dm.m, dm.n)
    torch.cuda.synchronize()

    # every engine holds valid octaves / channels / detections before any partial loop is timed
    for e in engines:
        e.run(dm)
    torch.cuda.synchronize()

    # ---- step functions (fused: the channel kernel writes the channels as threshold ranks of the cascade, one byte
    #      each, and the cascade scans those -- what PyramidEngine.run / Model.detect do when the model allows it)
    fused = engines[0].ranks_for(dm)
    if args.only == "all":
        if args.no_graph:
            steps = [(lambda e=e: e.run(dm)) for e in engines]
        else:
            graphs = [e.capture(dm) for e in engines]
            steps = [g.replay for g in graphs]
    elif args.only == "channels":
        steps = [(lambda e=e: e.launch_channels(dm if fused else None, floats=not fused)) for e in engines]
    elif args.only == "octaves":
        steps = [e.launch_octaves for e in engines]
    else:
        steps = [(lambda e=e: e.run_cascade(dm, ranks=fused)) for e in engines]

    # ---- parity gate before timing, on the path that is timed: engine 0's step (a hipGraph replay unless
    #      --no-graph) runs once more and what IT wrote -- detections and alive[level, stage] of the first and the
    #      last image of the batch -- is compared with the oracle, bit for bit
    parity = None

    def poison(e):
        e.detb.counts.fill_(0x7fffffff)                      # stale values the step must overwrite
        e._casc_state(dm)["alive"].fill_(-1)

    def gate(e, path):
        """What the last launch left in engine `e` -- detections and alive[level, stage] of the first and the last
        image of its batch -- against the oracle, bit for bit."""
        from util import oracle_detect
        if e.detb.max_count() > e.detb.cap:
            raise SystemExit("bench: detection buffer overflow in the parity gate")
        d = e.sorted_detections().cpu().numpy().view(nat.DET_DTYPE).reshape(-1)
        alive = e._casc_state(dm)["alive"][:, :, :len(M)].cpu().numpy().astype(np.int64)
        checked = []
        j0 = engines.index(e)
        for b in sorted({0, B - 1}):
            db = d[d["image"] == b]
            seed = (rank * P + j0) * B + b
            if seed not in gate_refs:
                gate_refs[seed] = oracle_detect(M, synth_image(H, W, seed))
            ref = gate_refs[seed]
            ok = (np.array_equal(db["level"], ref["level"]) and np.array_equal(db["r"], ref["r"]) and
                  np.array_equal(db["c"], ref["c"]) and np.array_equal(db["score"].view(np.uint32), ref["scores"].view(np.uint32)) and
                  np.array_equal(alive[b], ref["alive"]))
            if not ok:
                raise SystemExit(f"bench: rank {rank}: image {b} of the replayed step differs from the oracle -- refusing to time a wrong kernel")
            checked.append({"image": f"seed {(rank * P + j0) * B + b}", "detections": int(ref["scores"].size),
                            "eval_cost": ref["n_weak"] / ref["n_loc"]})
        return {"path": path, "images": checked, "bit_exact": True}

    if args.only == "all" and not args.stages:               # on EVERY rank: each checks its own engine 0 (its own images)
        poison(engines[0])
        steps[0]()
        torch.cuda.synchronize()
        parity = gate(engines[0], "eager launches" if args.no_graph else "hipGraph replay per step")

    gath = comm = None
    n_streams = max(1, min(args.streams, P))
    lanes = [torch.cuda.Stream() for _ in range(n_streams)] if n_streams > 1 else [torch.cuda.current_stream()]
    if coll:
        # Several ranks: every step ends with its engine's valid records packed (wb_det_pack_launch, inside the step's
        # graph) into the engine's slot of a send buffer, and every round of P steps ends with ONE all_gather of the P
        # slots on a side stream -- overlapped with the next round, no host synchronisation.  (One collective per step
        # -- launch, two events, a stream hop: ~50 us of host time -- made the step host-bound: 95 us against 67.)
        # The gathered prefix is sized from what the workload produces (2x the fullest image set, agreed over the
        # ranks), not from the buffer's capacity: a few hundred KB per rank instead of the whole detection buffer.
        most = torch.tensor([max(int(e.detb.counts.sum().item()) for e in engines)], dtype=torch.int64,
                            device="cuda" if args.backend == "nccl" else "cpu")
        dist.all_reduce(most, op=dist.ReduceOp.MAX)
        # (the resident images are fixed, so a step's record count is known exactly: an eighth of slack.  The collective
        # moves the slot's capacity, valid or not -- at 2x the count, 20 steps and 8 ranks every rank took in 14 MB per
        # timed region, ~0.1 ms of a 1.3 ms region on a ring over xGMI; the gathered-content check below catches a
        # truncated slot before anything is timed)
        n_most = int(most.item())
        rows = max(1024, n_most + n_most // 8 + 64)
        gath = RoundGatherer(rows, P, engines[0].dev)
        comm = torch.cuda.Stream()
        ev_done = [torch.cuda.Event() for _ in engines]      # engine j's step (incl. its pack) finished
        ev_comm = [torch.cuda.Event() for _ in range(gath.SETS)]   # the collective that read send[set] finished
        if args.only == "all" and not args.no_graph:
            # the step graph again, once per buffer set, with the pack inside
            def capture_with_pack(e, out):
                g = torch.cuda.CUDAGraph()
                torch.cuda.synchronize()
                with capturing(g):
                    e.run(dm)
                    e.pack(out)
                return g
            coll_steps = [[capture_with_pack(e, gath.send[w][j]).replay for j, e in enumerate(engines)] for w in range(gath.SETS)]
        else:
            coll_steps = [[(lambda j=j, w=w: (steps[j](), engines[j].pack(gath.send[w][j]))) for j in range(P)]
                          for w in range(gath.SETS)]

    def flush_round(w):
        """One all_gather of buffer set w on the side stream, behind the steps that packed into it."""
        with torch.cuda.stream(comm):
            for j in range(P):
                comm.wait_event(ev_done[j])
            gath.gather(w)
            ev_comm[w].record(comm)

    def run_steps(k0, k):
        main = torch.cuda.current_stream()
        for st in lanes:
            if st is not main:
                st.wait_stream(main)
        for i in range(k0, k0 + k):
            j = i % P                                        # engine j always runs on stream j % n_streams
            st = lanes[j % n_streams]
            with torch.cuda.stream(st):
                if coll:
                    w = (i // P) % gath.SETS
                    st.wait_event(ev_comm[w])                # send[w] free again: the collective two rounds back is done
                    coll_steps[w][j]()
                    ev_done[j].record(st)
                else:
                    steps[j]()
            if coll and (j == P - 1 or i == k0 + k - 1):
                flush_round((i // P) % gath.SETS)            # (also at the end of a region: all its work ends inside it)
        for st in lanes:
            if st is not main:
                main.wait_stream(st)
        if coll:
            main.wait_stream(comm)

    run_steps(0, args.warmup)
    # Graph mode, short regions: the K steps of a timed region -- K x (octaves, channels, cascade), spread over the streams
    # exactly as run_steps spreads them -- are captured ONCE, one hipGraph per stream holding that stream's steps in
    # order, and a region is one replay per stream: n_streams launches per region instead of K, every stream fed from
    # its first microsecond (tools/region_launch_probe.py: 20 steps take 1.27 ms replayed step by step, 1.22 ms as four
    # per-stream graphs).  With several ranks every captured step also packs its engine's valid records into the step's
    # slot of one send buffer, and the region ends with ONE all_gather of the K slots behind the streams' graphs -- the
    # end-of-batch exchange (a rank's steps are then the same graphs as a single GPU's: measured with one rank over RCCL,
    # --force-collective, 20 steps: 0.082 ms per step with a collective per round of 4 steps, see below for this form).
    region = None
    gath_region = None
    # (auto: K <= 64 -- a longer region amortises its K launches by itself; with several ranks up to 1024 steps, because there
    # the step-by-step form also pays a collective per round of P steps: 0.0745 against 0.0597 ms per step at K = 200)
    use_region = args.region_graph == "on" or (args.region_graph == "auto" and args.steps <= (1024 if coll else 64))
    if args.only == "all" and not args.no_graph and use_region:
        eager = [(lambda e=e: e.run(dm)) for e in engines]
        if coll:
            gath_region = RoundGatherer(rows, args.steps, engines[0].dev)
        torch.cuda.synchronize()
        region = []
        for l, st in enumerate(lanes if n_streams > 1 else [None]):
            mine = [i for i in range(args.steps) if (i % P) % n_streams == l]
            if not mine:
                continue
            g = torch.cuda.CUDAGraph()
            with (capturing(g, stream=st) if st is not None else capturing(g)):
                for i in mine:
                    eager[i % P]()
                    if coll:
                        engines[i % P].pack(gath_region.send[0][i])
            region.append((st, g))

        def run_region():
            # (no cross-stream waits before the replays: a region starts after a device-wide synchronisation)
            for st, g in region:
                if st is None:
                    g.replay()
                else:
                    with torch.cuda.stream(st):
                        g.replay()
            if coll:
                main = torch.cuda.current_stream()
                for st, _ in region:
                    if st is not None:
                        main.wait_stream(st)
                gath_region.gather(0)                        # every rank's K slots to every rank, inside the region

        # the gate again, on THESE graphs: what their replay leaves in the first and the last engine they drive
        checked = sorted({0, min(P, args.steps) - 1})
        if parity is not None:
            for j in checked:
                poison(engines[j])
        run_region()                                         # (first replay outside the timed regions)
        torch.cuda.synchronize()
        if parity is not None:
            parity = {"path": "one hipGraph replay per stream and timed region", "bit_exact": True,
                      "engines": [dict(gate(engines[j], ""), engine=j) for j in checked]}
            for g in parity["engines"]:
                g.pop("path")
        if coll:
            # ... and what the collective delivered: this rank's own block of the last step's slot, as every rank received
            # it, against the engine that packed it
            i_last = args.steps - 1
            got = gath_region.merged(0, i_last, [B] * world)
            got = got[(got["image"] >= rank * B) & (got["image"] < (rank + 1) * B)]
            own = engines[i_last % P].sorted_detections().cpu().numpy().view(nat.DET_DTYPE).reshape(-1)
            same = (got.size == own.size and np.array_equal(got["image"] - rank * B, own["image"]) and
                    all(np.array_equal(got[k], own[k]) for k in ("level", "r", "c")) and
                    np.array_equal(got["score"].view(np.uint32), own["score"].view(np.uint32)))
            if not same:
                raise SystemExit(f"bench: rank {rank}: the gathered detections of step {i_last} differ from the engine that packed them")
            if parity is not None:
                parity["gathered"] = {"step": i_last, "detections": int(own.size), "bit_exact": True}
    # the W untimed steps once more, right in front of the timed regions: the parity gates above are seconds of host work
    # with an idle GPU behind them, and the first timed region used to start on a GPU that had clocked down
    # (value_spread.min 11 % under the median in round 2)
    # ... and for --warm-ms of device time: after idle seconds the GPU takes about 20 ms of work to reach its steady clocks
    # (twenty-step regions of 1.3 ms, timed back to back from a cold start: 0.067, 0.065, 0.064, ... 0.0595 ms per step
    # from the fifteenth on -- value_spread.ms_per_step_by_region shows whatever trend is left).  A count, not a clock:
    # every rank replays the same number of regions (they hold collectives).
    n_warm = max(1, int(round(args.warm_ms / max(args.steps * B * (0.06 if args.config == "2" else 0.35), 1e-3))))
    for _ in range(n_warm):
        if region is not None:
            run_region()
        else:
            run_steps(0, max(args.warmup, args.steps))
    dts = []
    for rep in range(max(1, args.repeats)):
        torch.cuda.synchronize()
        if coll:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        if region is not None:
            run_region()
        else:
            run_steps(args.warmup + rep * args.steps, args.steps)
        torch.cuda.synchronize()
        if coll:
            dist.barrier()
        torch.cuda.synchronize()
        dts.append(time.perf_counter() - t0)
    per_rank = None
    if coll:
        cdev = "cuda" if args.backend == "nccl" else "cpu"
        mine = torch.tensor([float(np.median(dts)) / args.steps * 1e3, 1.0 if parity is not None else 0.0], dtype=torch.float64, device=cdev)
        allr = torch.zeros((world, 2), dtype=torch.float64, device=cdev)
        dist.all_gather_into_tensor(allr, mine.view(1, 2))
        per_rank = {"ms_per_step": [float(x) for x in allr[:, 0].tolist()], "parity_gate_passed": [bool(x) for x in allr[:, 1].tolist()],
                    "world_size_reported": dist.get_world_size(), "backend": dist.get_backend()}
        tt = torch.tensor(dts, dtype=torch.float64, device=cdev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)            # every region: the slowest rank's time
        dts = [float(x) for x in tt.tolist()]
    dt = float(np.median(dts))

    # ---- per-kernel durations (HIP events, launch stream) and the roofline of the dominant one
    roof = None
    kern = {}
    if rank == 0 and args.only == "all":
        e = engines[0]
        it = max(20, min(args.steps, 100))
        kern["octaves_ms"] = event_time_ms(e.launch_octaves, it, torch)
        kern["channels_ms"] = event_time_ms(lambda: e.launch_channels(dm if fused else None, floats=not fused), it, torch)
        kern["cascade_ms"] = event_time_ms(lambda: e.run_cascade(dm, ranks=fused), it, torch)   # counter reset + tile kernel
        # the tile kernel alone, as rocprofv3 sees it (the counters are not reset in this loop; records past the
        # capacity are dropped, the work is the same)
        kern["cascade_tile_ms"] = event_time_ms(lambda: e.launch_cascade(dm, ranks=fused), it, torch)
        e.run_cascade(dm, ranks=fused)
        ab = plan.algorithmic_bytes(1)
        name = "channels_kernel" if kern["channels_ms"] >= kern["cascade_tile_ms"] else "cascade_kernel"
        ms = kern["channels_ms"] if name == "channels_kernel" else kern["cascade_tile_ms"]
        abytes = ab[name] * B
        # HBM bytes per launch: NOT measured in this run -- read from the committed rocprofv3 PMC passes (batch-1 launches of
        # the configs[1] workload only); `traffic_source` says so in the line
        traffic = tsrc = None
        tpath = os.path.join(PROFILE_DIR, "traffic_pmc.json" if args.config == "2" else f"traffic_pmc_cfg{args.config}.json")
        if os.path.exists(tpath) and B == wl["batch"] and args.channels == "grad_hist":
            with open(tpath) as f:
                tj = json.load(f)
            key = "channels_kernel" if name == "channels_kernel" else "cascade_tile_kernel"
            traffic = tj.get(key, {}).get("traffic_bytes_per_launch_b1" if args.config == "2" else "traffic_bytes_per_launch")
            if traffic is not None:
                tsrc = f"{os.path.relpath(tpath, ROOT)} (committed rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, not measured in this run)"
        roof = {"config": args.config, "bound": "hbm", "kernel": "channels_kernel" if name == "channels_kernel" else "cascade_tile_kernel", "achieved": abytes / (ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": abytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": tsrc,
                # what the hardware sees: the rank path writes one byte per channel value instead of SURVEY 8(d)'s float32
                # and most source re-reads hit the caches, so the bytes MOVED are well below the algorithmic figure --
                # frac is the contract's number, frac_moved the HBM utilisation (these kernels are VALU-issue-bound:
                # see issue_bound.valu_busy)
                "moved_bytes": traffic, "frac_moved": (traffic / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if traffic else None,
                "algorithmic_bytes_per_launch": abytes, "avg_launch_ms": ms,
                "avg_launch_source": "HIP events around back-to-back launches on the launch stream, this run"}

    # ---- the same workload through the reference's Python surface: host ndarray in, Boxes out (Model.detect,
    #      reference model.py:149-179) -- PCIe copies, kernels, ordering, boxes, synchronisation; never `value`
    through_api = None
    if rank == 0 and args.only == "all" and not args.no_through_api and not args.stages and args.config == "2":
        imgs = [synth_image(H, W, 7000 + i) for i in range(8)]
        for im in imgs[:3]:
            M.detect(im)
        torch.cuda.synchronize()
        n_api = 40
        t0 = time.perf_counter()
        for i in range(n_api):
            M.detect(imgs[i % len(imgs)])
        t_api = (time.perf_counter() - t0) / n_api
        through_api = {"call": "Model.detect(host uint8 ndarray) -> Boxes on the host", "ms_per_image": t_api * 1e3,
                       "windows_per_s": n_loc / t_api, "images": n_api}
        # the same call on PAGE-LOCKED arrays (a caller that decodes into pinned buffers; an extension of the reference's
        # pageable ndarray input): the upload is an asynchronous DMA on the call's stream instead of a blocking staged copy
        pins = [torch.from_numpy(im).pin_memory() for im in imgs]
        pimgs = [t.numpy() for t in pins]
        for im in pimgs[:3]:
            M.detect(im)
        torch.cuda.synchronize()
        per_call = []
        for i in range(n_api):
            t0 = time.perf_counter()
            M.detect(pimgs[i % len(pimgs)])
            per_call.append(time.perf_counter() - t0)
        t_pin = float(np.mean(per_call))
        through_api["pinned"] = {"call": "Model.detect(page-locked host uint8 ndarray) -> Boxes on the host", "ms_per_image": t_pin * 1e3,
                                 "ms_median": float(np.median(per_call)) * 1e3, "ms_max": float(np.max(per_call)) * 1e3,
                                 "windows_per_s": n_loc / t_pin, "images": n_api,
                                 "note": "the upload of 2 MB is PCIe time either way (~0.047 ms); page-locked, the host is not blocked by it"}
        dimg = torch.from_numpy(imgs[0]).to(engines[0].dev)
        for _ in range(3):
            M.detect(dimg)
        t0 = time.perf_counter()
        for i in range(n_api):
            M.detect(dimg)
        t_dev = (time.perf_counter() - t0) / n_api
        through_api["device_tensor"] = {"call": "Model.detect(2-D uint8 torch tensor already on the GPU) -> Boxes on the host",
                                        "ms_per_image": t_dev * 1e3, "windows_per_s": n_loc / t_dev, "images": n_api}
        # ... and the loop the reference's detection script runs over its files (scripts/waldboost-detect.py:64-67),
        # pipelined: Model.detect_stream keeps three lanes in flight (upload | scan | read-back and ordering), one image
        # or a batch of 16 per lane
        def stream_rate(batch, n_st):
            gen = lambda k: (imgs[i % len(imgs)] for i in range(k))
            list(M.detect_stream(gen(3 * 3 * batch), batch=batch))
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            n_box = sum(len(bx) for bx in M.detect_stream(gen(n_st), batch=batch))
            t_st = (time.perf_counter() - t0) / n_st
            return {"ms_per_image": t_st * 1e3, "windows_per_s": n_loc / t_st, "images": n_st, "lanes": 3, "batch": batch,
                    "boxes_per_image": n_box / n_st}
        through_api["stream"] = dict(stream_rate(1, 200), call="Model.detect_stream(iterable of host uint8 ndarrays) -> Boxes per image, in order")
        through_api["stream_batch16"] = dict(stream_rate(16, 480), call="Model.detect_stream(..., batch=16)")
        imgs = pimgs
        through_api["stream_pinned"] = dict(stream_rate(1, 200), call="Model.detect_stream(iterable of page-locked host uint8 ndarrays)")

    # ---- BASELINE configs[3]: 512 x 1080p cut over the ranks by shard_range, every rank's shard resident on its GPU and
    #      scanned 64 images per launch through the PRODUCT's multi-GPU entry, distributed.detect_sharded -- its chunked
    #      two-stream scan, agree_capacity (all-reduce MAX), the ordering of the records on the device, the gather of the
    #      records to rank 0 and the all-reduce (SUM) of alive[level, stage] all inside the timed call.  What the
    #      reference's detection script does with a process pool over files (scripts/waldboost-detect.py:64-67).  At N = 1
    #      this is the curve's first point (eight launches of 64, one rank in the RCCL group); at N = 8 BASELINE configs[3].
    config3 = None
    if do_cfg3:
        from waldboost_amd import distributed as wbd
        own_group = not dist.is_initialized()
        if own_group:                                        # (a single rank with no launcher: a one-rank RCCL group)
            import socket
            with socket.socket() as so:
                so.bind(("127.0.0.1", 0))
                os.environ.setdefault("MASTER_PORT", str(so.getsockname()[1]))
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            if args.backend == "nccl":
                dist.init_process_group("nccl", device_id=torch.device("cuda", local_dev), rank=0, world_size=1)
            else:
                dist.init_process_group(args.backend, rank=0, world_size=1)
        cdev = "cuda" if dist.get_backend() == "nccl" else "cpu"
        n3 = args.config3_images
        shard = torch.from_numpy(cfg3_host).to(engines[0].dev)                  # this rank's images, resident
        del cfg3_host
        walls, scans, phases, dets = [], [], [], None
        for rep in range(4):                                  # the first call is untimed: buffers, graphs, capacity growth
            torch.cuda.synchronize()
            dist.barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            det3, _, total3 = wbd.detect_sharded(M, shard, batch=CFG3_BATCH, per_image_alive=False)
            torch.cuda.synchronize()
            dist.barrier()
            torch.cuda.synchronize()
            lt = wbd.LAST_TIMING
            dt3 = torch.tensor([time.perf_counter() - t0, lt.get("scan_s", 0.0), lt.get("agree_s", 0.0), lt.get("order_s", 0.0),
                                lt.get("gather_s", 0.0)], dtype=torch.float64, device=cdev)
            allt = torch.zeros((world, 5), dtype=torch.float64, device=cdev)
            dist.all_gather_into_tensor(allt, dt3.view(1, 5))
            if rep:
                walls.append(float(allt[:, 0].max()))
                scans.append([float(x) for x in allt[:, 1].tolist()])
                phases.append([[round(float(x) * 1e3, 3) for x in allt[r, 2:].tolist()] for r in range(world)])
        if rank == 0:
            wall3 = float(np.median(walls))
            # rank 0's merged result against the single-image entry point, Model.detect_raw, for the first, the middle and
            # the last image of the global batch (regenerated here from their seeds)
            checked = []
            for g in sorted({0, n3 // 2, n3 - 1}):
                ref = M.detect_raw(synth_image(H, W, CFG3_SEED0 + g))
                mine = det3[det3["image"] == g]
                same = (mine.size == ref["scores"].size and np.array_equal(mine["level"], ref["level"]) and
                        np.array_equal(mine["r"], ref["r"].astype(np.uint16)) and np.array_equal(mine["c"], ref["c"].astype(np.uint16)) and
                        np.array_equal(mine["score"].view(np.uint32), ref["scores"].view(np.uint32)))
                if not same:
                    raise SystemExit(f"bench: configs[3]: image {g} of the gathered detections differs from Model.detect_raw")
                checked.append({"image": g, "detections": int(mine.size)})
            config3 = {
                "workload": f"BASELINE configs[3]: {n3} x 1920x1080 uint8 (seeds {CFG3_SEED0}+i) cut over {world} rank(s) by shard_range, each "
                            f"shard resident in HBM and scanned {CFG3_BATCH} images per launch by waldboost_amd.distributed.detect_sharded; "
                            "timed: the whole call between two barriers -- chunked scan on two streams, agree_capacity, ordering on "
                            "the device, gather of the records to rank 0 (+ their read-back to host memory), all-reduce of alive[level, stage]",
                "images": n3, "n_gpus": world, "images_per_rank": [wbd.shard_range(n3, r, world)[1] - wbd.shard_range(n3, r, world)[0] for r in range(world)],
                "images_per_launch": CFG3_BATCH, "backend": dist.get_backend(),
                "wall_ms": wall3 * 1e3, "wall_ms_all": [round(w * 1e3, 3) for w in walls], "calls_timed": len(walls), "calls_untimed": 1,
                "windows_per_s": n3 * n_loc / wall3, "images_per_s": n3 / wall3, "scaling": "strong (the batch is fixed, the ranks share it)",
                "per_rank_scan_ms": [round(x * 1e3, 3) for x in scans[int(np.argsort(walls)[len(walls) // 2])]],
                "per_rank_exchange_ms": {"columns": ["agree_capacity", "order on the device", "gather + read-back"],
                                         "rows": phases[int(np.argsort(walls)[len(walls) // 2])]},
                "detections": int(det3.size), "n_weak": int(total3.sum()), "eval_cost": float(total3.sum()) / (n3 * n_loc),
                "check": {"against": "Model.detect_raw per image (level, r, c, score bits)", "images": checked, "bit_exact": True},
                "host_prep_s": round(t_gen, 2),
            }
        del shard
        if own_group:
            dist.destroy_process_group()

    if rank == 0:
        windows = world * args.steps * B * n_loc
        ab = plan.algorithmic_bytes(1)
        n_stage = len(M)
        out = {
            "metric": f"candidate windows/s, {'1080p' if args.config == '2' else '4K'} {args.channels} pyramid, {n_stage}-stage depth-2 cascade",
            "value": windows / dt, "unit": "windows/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u8 image, f64/f32 channel arithmetic, f32 scores" if args.channels == "grad_hist"
                     else "u8 image, integer channel arithmetic (u8 channels), f32 scores",
            "data": "synthetic",
            "config": {"workload": f"{wl['name']}: {W}x{H} uint8, shrink={shrink} n_per_oct={n_per_oct} smooth={smooth} {args.channels}, "
                                   f"window (12,12,4), {n_stage}-stage depth-2 cascade, {B} image(s)/step/GPU"
                                   + (" (shrink 4 is this build's extension of the reference's shrink in (1, 2): oracle-defined)" if shrink == 4 else ""),
                       "batch_per_gpu": B, "levels": plan.n_levels, "windows_per_image": n_loc,
                       "launch": "eager" if args.no_graph else ("one hipGraph replay per stream and timed region of K steps" if region is not None
                                                                else "hipGraph replay per step"), "only": args.only,
                       "channels_in_hbm": (("16-bit threshold ranks of the cascade (WB_DTYPE_RANK16)" if dm.rank_dtype == nat.WB_DTYPE_RANK16
                                            else "uint8 threshold ranks of the cascade (WB_DTYPE_RANK8)") if fused else spec.dtype.name),
                       "streams": n_streams, "pool": P,
                       "warm": f"{n_warm} untimed region(s) of {args.steps} steps right before the timed ones (--warm-ms {args.warm_ms:g})",
                       "warm_steps_untimed": args.warmup + n_warm * (args.steps if region is not None else max(args.warmup, args.steps)),
                       "cascade_kernel": (f"model-specialised (hiprtc at model load, {t_jit:.1f} s incl. cache lookup)" if jit else "generic"),
                       "collective": ("none" if not coll else
                                      "one all_gather of the K steps' packed detection prefixes at the end of every timed region" if gath_region is not None
                                      else "one all_gather of the P packed detection prefixes per round of P steps (side stream)")},
            "mpixels_per_s": world * args.steps * B * H * W / dt / 1e6,
            "images_per_s": world * args.steps * B / dt,
            "pipeline_roofline_frac": (windows / dt) * (ab["total"] / n_loc) / (HBM_PEAK_GBS * 1e9 * world),
            "value_spread": {"repeats": len(dts), "min": windows / max(dts), "max": windows / min(dts),
                             "ms_per_step_min": min(dts) / args.steps * 1e3, "ms_per_step_max": max(dts) / args.steps * 1e3,
                             "ms_per_step_by_region": [round(x / args.steps * 1e3, 5) for x in dts]},
            "kernels": kern, "parity": parity, "ranks": per_rank, "through_api": through_api,
            "roofline": roof, "issue_bound": issue_bound(roof, B, args.channels), "cpu_baseline": cpu,
            "config3": config3,
        }
        emit(out)
    if coll:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
