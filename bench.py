#!/usr/bin/env python3
"""bench.py -- walk-steps/s of the Walk-on-Stars hot path on BASELINE.json configs[1]:
ladybug, uniform integrator, 1024^2 grid, 256 spp, max depth 64, epsilon shell 1.

A "step" of this bench is one full solve of that frame (one pass of the hot path over the
whole batch of 1024^2 x 256 walks).  With N GPUs the frame's 8x8-pixel tiles are dealt
round-robin to the ranks (strong scaling of the named frame, as north_star asks) and the
zero-padded fields are summed with one RCCL all-reduce inside the timed region.
Scene upload and LBVH build happen before the timed region (the reference's solve() timer
excludes them too: integrator/uniform/integrator.cu:666-672).

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

BYTES_PER_STEP = 98.0      # 37 B item read + 37 B successor + 16 B PCG read + 8 B PCG write
HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def cpu_baseline(problem, frame, spp, depth, eps, target_s=15.0):
    """The oracle (a literal CPU port of the reference path) on a bounded band of the same frame."""
    from oracle.oracle import Oracle
    o = Oracle()
    cores = os.cpu_count() or 1
    sd = problem.as_dict()
    mid = frame // 2
    # calibration: 4 rows at 8 spp
    t = time.time()
    r = o.solve(sd, frame, frame, 8, depth, eps, pixel_begin=mid * frame, pixel_end=(mid + 4) * frame, threads=cores)
    rate = r["walk_steps"] / max(r["seconds"], 1e-6)
    steps_per_row = r["walk_steps"] / 4.0 * (spp / 8.0)
    rows = int(max(2, min(frame // 2, target_s * rate / max(steps_per_row, 1.0))))
    b = (mid - rows // 2) * frame
    e = b + rows * frame
    r = o.solve(sd, frame, frame, spp, depth, eps, pixel_begin=b, pixel_end=e, threads=cores)
    return {
        "value": r["walk_steps"] / r["seconds"], "unit": "walk-steps/s", "cores": cores, "kind": "port",
        "sample": "rows %d..%d of the %dx%d frame at %d spp (%d walk steps, %.1f s)" % (
            b // frame, e // frame, frame, frame, spp, r["walk_steps"], r["seconds"]),
    }, (b, e, r["field"])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--scene", default="ladybug")
    ap.add_argument("--frame", type=int, default=1024)
    ap.add_argument("--spp", type=int, default=256)
    ap.add_argument("--depth", type=int, default=0, help="0 = the scene's maxWalkingDepth")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", default=None, help="torch.distributed backend (default nccl = RCCL)")
    ap.add_argument("--no-1spp", action="store_true", help="skip the time-to-1spp probe (profiling runs)")
    ap.add_argument("--steps-per-round", type=int, default=0)
    ap.add_argument("--opt", action="append", default=[], help="key=value passed to wost_set_option")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from elaina_amd import Problem, UniformIntegrator, UniformIntegratorSettings
    from elaina_amd import distributed as D

    rank, world, local = D.init_process_group(args.backend)
    if world != args.gpus and rank == 0:
        print("warning: --gpus %d but WORLD_SIZE %d; using WORLD_SIZE" % (args.gpus, world), file=sys.stderr)
    local = local % max(torch.cuda.device_count(), 1)   # several ranks may share one GPU in tests
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    problem = Problem.load_scene(args.scene)
    depth = args.depth or problem.default_max_depth
    eps = problem.default_eps
    frame = args.frame
    it = UniformIntegrator(problem, UniformIntegratorSettings((frame, frame), args.spp, depth, eps), device=local)
    if args.steps_per_round:
        it.set_option("steps_per_round", args.steps_per_round)
    for kv in args.opt:
        k, v = kv.split("=")
        it.set_option(k, float(v))
    field = torch.zeros(frame * frame * 3, dtype=torch.float32, device=dev)
    stream = torch.cuda.current_stream(dev)

    def one_pass():
        field.zero_()
        st = it.solve_sharded(rank, world, field.data_ptr(), stream.cuda_stream)
        D.reduce_field(field, world)
        return st

    # time-to-1spp (cold first pass of a fresh handle, then steady state), outside the timed region
    it1 = UniformIntegrator(problem, UniformIntegratorSettings((frame, frame), 1, depth, eps), device=local)
    t1 = [0.0, 0.0] if args.no_1spp else []
    for _ in range(0 if args.no_1spp else 4):
        field.zero_()
        torch.cuda.synchronize()
        t = time.perf_counter()
        it1.solve_sharded(rank, world, field.data_ptr(), stream.cuda_stream)
        D.reduce_field(field, world)
        torch.cuda.synchronize()
        t1.append((time.perf_counter() - t) * 1e3)
    it1.close()

    for _ in range(args.warmup):
        one_pass()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    steps_local = 0
    kernel_ms = 0.0
    launches = 0
    for _ in range(args.steps):
        st = one_pass()
        steps_local += st["walk_steps"]
        kernel_ms += st["kernel_ms"]
        launches += st["kernel_launches"]
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0

    tot = torch.tensor([float(steps_local), elapsed, kernel_ms, float(launches)], dtype=torch.float64, device=dev)
    if world > 1:
        mx = tot.clone()
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        elapsed = float(mx[1].item())
    total_steps = float(tot[0].item())

    if rank == 0:
        value = total_steps / elapsed
        # dominant kernel: walk_round_kernel.  Algorithmic bytes per launch = 98 B x the walk steps
        # that launch advanced; both summed over this rank's launches of the timed region.
        ach = (steps_local * BYTES_PER_STEP) / (kernel_ms * 1e-3) / 1e9 if kernel_ms > 0 else 0.0
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "walk_round_traffic.json")
        if os.path.exists(tpath):
            # PMC passes cannot run inside this process: the figure comes from the committed
            # rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE runs of this same command (tools/gpu_round.sh)
            traffic = json.load(open(tpath)).get("hbm_bytes_per_launch")
        out = {
            "metric": "walk-steps/s", "value": value, "unit": "walk-steps/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": "%s uniform %dx%d grid %d spp depth %d eps %g" % (
                args.scene, frame, frame, args.spp, depth, eps), "parallelism": "pixel-tiles x%d" % world,
                "walk_steps_per_pass": total_steps / args.steps},
            "time_to_1spp_ms": {"cold": t1[0], "steady": sorted(t1[1:])[len(t1[1:]) // 2]},
            "roofline": {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": ach / HBM_PEAK_GBS, "traffic": traffic,
                         "traffic_unit": "HBM bytes per launch (profiles/walk_round_traffic.json)",
                         "kernel": "walk_round_kernel", "launches": launches,
                         "avg_launch_ms": kernel_ms / max(launches, 1),
                         "algorithmic_bytes_per_walk_step": BYTES_PER_STEP},
        }
        if not args.no_cpu_baseline:
            # the host baseline is a reported figure at N = 1 only; at N > 1 a short band of the
            # assembled field is still checked against the oracle
            base, (b, e, ref_field) = cpu_baseline(problem, frame, args.spp, depth, eps, target_s=15.0 if world == 1 else 2.0)
            if world == 1:
                out["cpu_baseline"] = base
            import numpy as np
            got = field.cpu().numpy().reshape(-1, 3)[b:e]
            den = float(np.linalg.norm(ref_field)) or 1.0
            out["rel_l2_vs_oracle"] = float(np.linalg.norm(got - ref_field)) / den
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    it.close()


if __name__ == "__main__":
    main()
