#!/usr/bin/env python3
"""bench.py -- walk-steps/s of the Walk-on-Stars hot path on the configurations of BASELINE.json.

  --config 2 (default)  ladybug, uniform integrator, 1024^2, 256 spp, depth 64   (the headline line)
  --config 3            fille,   uniform integrator, 1024^2, 256 spp, depth 128
  --config 4            ladybug, guided integrator with online training, 1024^2, 256 spp (256 trained)
  --config 5            ladybug, guided integrator, 2048^2, 1024 spp (256 trained), pixel tiles over the ranks

A "step" of this bench is one full solve of the frame (one pass of the hot path over the whole batch
of walks).  With N GPUs the frame's 8x8-pixel tiles are dealt round-robin to the ranks (strong scaling
of the named frame, as north_star asks) and the zero-padded fields are summed with one RCCL all-reduce
inside the timed region.  Scene upload and acceleration-structure build happen before the timed region
(the reference's solve() timer excludes them too: integrator/uniform/integrator.cu:666-672).

`python bench.py --gpus N` without a launcher starts the N ranks itself (a torch.distributed.run child,
spawned before this process touches the GPU) and relays rank 0's JSON line; under an external launcher
(RANK / WORLD_SIZE in the environment) it is one of the ranks.  With the default --config 2 on one GPU
the line also carries "configs": one pass each of config 3 and config 4 (both network precisions), the 3-D scenes
(uniform3d, guided3d), a 2-D scene with a Neumann boundary on the tree (neumann2d), the guiding gain and the variance check
(--no-extras skips them); at N > 1 it carries one pass of config 5.

Rank 0 prints ONE JSON line on stdout, the complete one; when extras follow the headline it also goes to stderr, marked
"partial", as soon as it is measured (a run that is cut off keeps it in the log).
"""
import argparse
import json
import os
import subprocess
import sys
import time

T_START = time.time()
ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

BYTES_PER_STEP = 98.0      # 37 B item read + 37 B successor + 16 B PCG read + 8 B PCG write
HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
MFMA_F32_PEAK_TF = 157.3   # MI355X_MICROARCH.md: dense fp32 matrix peak
MFMA_F16_PEAK_TF = 2500.0  # MI355X_MICROARCH.md: dense bf16 / f16 matrix peak (~2.5 PF)
FLOP_PER_POINT = 26624.0   # 2 * (32*64 + 64*64 + 64*64 + 64*48) per network evaluation
GUIDED_AABB = ((-100.0, -100.0), (600.0, 600.0))   # scene.aabb of data/*/n.json

CONFIGS = {
    2: dict(scene="ladybug", kind="uniform", frame=1024, spp=256),
    3: dict(scene="fille", kind="uniform", frame=1024, spp=256),
    4: dict(scene="ladybug", kind="guided", frame=1024, spp=256, train_spp=256),
    5: dict(scene="ladybug", kind="guided", frame=2048, spp=1024, train_spp=256),
}


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", type=int, default=2, choices=sorted(CONFIGS))
    ap.add_argument("--scene", default=None)
    ap.add_argument("--frame", type=int, default=0)
    ap.add_argument("--spp", type=int, default=0)
    ap.add_argument("--train-spp", type=int, default=-1)
    ap.add_argument("--depth", type=int, default=0, help="0 = the scene's maxWalkingDepth")
    ap.add_argument("--shared-network", action="store_true", help="guided, N > 1: one network for all ranks")
    ap.add_argument("--net-precision", type=int, default=32, choices=[32, 16],
                    help="guided: 32 = fp32 network (bit-exact mode, default), 16 = the reference's half-precision network "
                         "(inference and the training passes on f16 MFMAs; fp32 master weights)")
    ap.add_argument("--pipeline", type=int, default=0, choices=[0, 1],
                    help="guided configs: 1 = training passes on a second stream (wost_guided_set_option; opt-in, never the parity mode)")
    ap.add_argument("--train-group", type=int, default=1, help="guided configs: samples per training launch (opt-in, never the parity mode)")
    ap.add_argument("--net-train-precision", type=int, default=0, choices=[0, 32, 16],
                    help="guided: precision of the training passes alone (0 = follow --net-precision)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="only the selected config (no \"configs\" object)")
    ap.add_argument("--backend", default=None, help="torch.distributed backend (default nccl = RCCL)")
    ap.add_argument("--budget-s", type=float, default=170.0, help="wall-clock seconds from process start after which no further extra starts")
    ap.add_argument("--no-1spp", action="store_true", help="skip the time-to-1spp probe (profiling runs)")
    ap.add_argument("--steps-per-round", type=int, default=0)
    ap.add_argument("--opt", action="append", default=[], help="key=value passed to wost_set_option")
    return ap.parse_args(argv)


def self_launch(args):
    """--gpus N without a launcher: start the N ranks as a child job BEFORE anything here touches the
    GPU (a process that initialised HIP must never exec), relay its output, return its exit code."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    child = subprocess.Popen(cmd, env=env, cwd=ROOT)
    return child.wait()


def band_of(frame, rows):
    mid = frame // 2
    b = (mid - rows // 2) * frame
    return b, b + rows * frame


def cpu_baseline(problem, frame, spp, depth, eps, target_s=15.0):
    """The oracle (a literal CPU port of the reference path) on a bounded band of the same frame."""
    from oracle.oracle import Oracle
    o = Oracle()
    cores = os.cpu_count() or 1
    sd = problem.as_dict()
    mid = frame // 2
    # calibration: 4 rows at 8 spp
    r = o.solve(sd, frame, frame, min(8, spp), depth, eps, pixel_begin=mid * frame, pixel_end=(mid + 4) * frame, threads=cores)
    rate = r["walk_steps"] / max(r["seconds"], 1e-6)
    steps_per_row = r["walk_steps"] / 4.0 * (spp / float(min(8, spp)))
    rows = int(max(2, min(frame // 2, target_s * rate / max(steps_per_row, 1.0))))
    b, e = band_of(frame, rows)
    r = o.solve(sd, frame, frame, spp, depth, eps, pixel_begin=b, pixel_end=e, threads=cores)
    return {
        "value": r["walk_steps"] / r["seconds"], "unit": "walk-steps/s", "cores": cores, "kind": "port",
        "sample": "rows %d..%d of the %dx%d frame at %d spp (%d walk steps, %.1f s)" % (
            b // frame, e // frame, frame, frame, spp, r["walk_steps"], r["seconds"]),
    }, (b, e, r["field"])


def committed_counters(name, keys):
    """Figures derived from rocprofv3 --pmc passes (they cannot run inside this process): tools/gpu_round.sh writes
    profiles/<name>.json together with the identity of the kernel sources it measured (elaina_amd.build.source_id);
    "stale" says whether the running tree still has those sources."""
    path = os.path.join(ROOT, "profiles", name + ".json")
    if not os.path.exists(path):
        return None
    from elaina_amd.build import source_id
    v = json.load(open(path))
    out = {k: v.get(k) for k in keys}
    out["measured_on_sources"] = v.get("source_id")
    out["stale"] = v.get("source_id") != source_id()
    return out


def valu_block():
    """VALU figures of walk_round_kernel (profiles/walk_round_valu.json)"""
    return committed_counters("walk_round_valu", ("pipe_busy", "lane_efficiency", "lane_instr_per_step", "source"))


def torch_zeros_like(t):
    import torch
    return torch.zeros_like(t)


def rel_l2(a, b):
    import numpy as np
    den = float(np.linalg.norm(b)) or 1.0
    return float(np.linalg.norm(a - b)) / den


class Env:
    """process group + device of this rank"""

    def __init__(self, args):
        import torch
        import torch.distributed as dist
        from elaina_amd import distributed as D
        self.torch, self.dist, self.D = torch, dist, D
        self.rank, self.world, local = D.init_process_group(args.backend)
        if self.world != args.gpus:
            raise SystemExit("bench.py: --gpus %d but the job has %d ranks" % (args.gpus, self.world))
        if self.world > 1:
            assert dist.get_world_size() == args.gpus
            self.backend = dist.get_backend()
        else:
            self.backend = None
        self.local = local % max(torch.cuda.device_count(), 1)   # several ranks may share one GPU in tests
        torch.cuda.set_device(self.local)
        self.dev = torch.device("cuda", self.local)

    def barrier(self):
        if self.world > 1:
            self.dist.barrier()
        self.torch.cuda.synchronize()

    def reduce(self, vec, op):
        if self.world > 1:
            self.dist.all_reduce(vec, op=op)
        return vec


def run_uniform(env, scene, frame, spp, depth, steps, warmup, args, one_spp=True):
    """timed solves of the uniform integrator; returns the result dict (rank 0 fills the checks)"""
    torch = env.torch
    from elaina_amd import Problem, UniformIntegrator, UniformIntegratorSettings
    problem = Problem.load_scene(scene)
    depth = depth or problem.default_max_depth
    eps = problem.default_eps
    t_create = time.perf_counter()
    it = UniformIntegrator(problem, UniformIntegratorSettings((frame, frame), spp, depth, eps), device=env.local)
    create_ms = (time.perf_counter() - t_create) * 1e3
    if args.steps_per_round:
        it.set_option("steps_per_round", args.steps_per_round)
    for kv in args.opt:
        k, v = kv.split("=")
        it.set_option(k, float(v))
    field = torch.zeros(frame * frame * 3, dtype=torch.float32, device=env.dev)
    stream = torch.cuda.current_stream(env.dev)

    # the one exchange of a pass, timed with events on the stream -- no host synchronisation inside a timed pass, so N-GPU
    # `value` / ms_per_step are measured like the one-GPU ones -- and read back only for the timed passes
    exchange_events = []

    def one_pass(integ, timed=False):
        field.zero_()
        st = integ.solve_sharded(env.rank, env.world, field.data_ptr(), stream.cuda_stream)
        if env.world > 1:
            e0, e1 = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) if timed else (None, None)
            if timed:
                e0.record(stream)
            env.D.assemble_field(field, env.world, env.rank, frame, frame)
            if timed:
                e1.record(stream)
                exchange_events.append((e0, e1))
        return st

    # time-to-1spp (cold first pass of a fresh handle, then steady state), outside the timed region
    t1 = None
    if one_spp:
        it1 = UniformIntegrator(problem, UniformIntegratorSettings((frame, frame), 1, depth, eps), device=env.local)
        for kv in args.opt:
            k, v = kv.split("=")
            it1.set_option(k, float(v))
        t1 = []
        for _ in range(4):
            torch.cuda.synchronize()
            t = time.perf_counter()
            one_pass(it1)
            torch.cuda.synchronize()
            t1.append((time.perf_counter() - t) * 1e3)
        it1.close()

    for _ in range(warmup):
        one_pass(it)
    if env.world > 1 and warmup == 0:
        env.D.assemble_field(torch.zeros_like(field), env.world, env.rank, frame, frame)      # a warm communicator before the timed passes
    torch.cuda.synchronize()
    env.barrier()
    t0 = time.perf_counter()
    steps_local, kernel_ms, launches = 0, 0.0, 0
    sched = {"visits": 0, "trav_trips": 0, "step_trips": 0}
    for _ in range(steps):
        st = one_pass(it, timed=True)
        steps_local += st["walk_steps"]
        kernel_ms += st["kernel_ms"]
        launches += st["kernel_launches"]
        sched["visits"] += st["inner_visits"]; sched["trav_trips"] += st["trav_trips"]; sched["step_trips"] += st["step_trips"]
    torch.cuda.synchronize()
    env.barrier()
    elapsed = time.perf_counter() - t0

    tot = torch.tensor([float(steps_local)], dtype=torch.float64, device=env.dev)
    mx = torch.tensor([elapsed], dtype=torch.float64, device=env.dev)
    env.reduce(tot, env.dist.ReduceOp.SUM)
    env.reduce(mx, env.dist.ReduceOp.MAX)
    elapsed = float(mx[0].item())
    total_steps = float(tot[0].item())
    # The dominant kernel launch of the last timed pass, priced by SURVEY 8(d): 98 algorithmic bytes per walk step x the walk steps
    # THAT launch took / its duration (HIP events around it on the solve's stream, wost_last_launches).  Since round 6 a full-frame
    # solve is one persistent launch of walk_round_kernel (resident lanes take whole pixels, longest expected chain first) plus a
    # few rounds of what it leaves; other launches run beside those rounds on streams of their own, so the solve's kernel time is
    # the union of the spans on the solve's stream ("solve" below), not a sum over kernels.
    ll = it.last_launches()
    real = [l for l in ll if l["kind"] != 4 and l["ms"] > 0]
    dom = max(real, key=lambda l: l["ms"]) if real else None
    ach = dom["steps"] * BYTES_PER_STEP / (dom["ms"] * 1e-3) / 1e9 if dom else 0.0
    ach_solve = (steps_local * BYTES_PER_STEP) / (kernel_ms * 1e-3) / 1e9 if kernel_ms > 0 else 0.0
    out = {
        "workload": "%s uniform %dx%d grid %d spp depth %d eps %g" % (scene, frame, frame, spp, depth, eps),
        "value": total_steps / elapsed, "ms_per_step": elapsed / steps * 1e3, "walk_steps_per_pass": total_steps / steps,
        # the unit of work is the reference's (one item consumed from the evaluation-point queue at one depth); the depth-0 steps among
        # them make no closest-point query here: every sample of a pixel starts at the same point, whose query is answered once per pixel
        "walk_step_note": "%.1f %% of the counted walk steps are depth-0 steps (= walks started), whose closest-point query is cached per pixel; "
                          "same field and same count as the oracle, which repeats the query like the reference" % (
                              100.0 * st["walks_started"] / max(st["walk_steps"], 1)),
        "roofline": {"bound": "hbm", "hbm_formula": "98 B x walk steps of the launch / its duration (SURVEY 8d)",
                     "what_binds": "the HBM formula is the contract's yardstick, not what limits this kernel: the walker state stays in registers inside a launch "
                                   "(traffic << algorithmic bytes); the VALU pipes are what is busy -- see valu_frac, valu",
                     "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                     "kernel": "walk_round_kernel",
                     "launch": ({"kind": dom["kind_name"], "walkers": dom["walkers"], "grid": dom["grid"], "ms": dom["ms"], "walk_steps": dom["steps"],
                                 "share_of_the_pass": dom["steps"] / max(st["walk_steps"], 1)} if dom else None),
                     "launches": launches, "avg_launch_ms": (dom["ms"] if dom else 0.0),
                     "launches_what": "avg_launch_ms = the duration of the dominant launch (one per pass: in a rocprofv3 --kernel-trace --stats summary of the same command "
                                      "the walk_round_kernel instantiation <.., true, false, false> -- REFILL -- has one call per pass, compare its average); "
                                      "`launches` counts the launches on the solve's stream over the timed passes",
                     "solve": {"achieved": ach_solve, "frac": ach_solve / HBM_PEAK_GBS, "kernel_ms_per_pass": kernel_ms / max(steps, 1),
                               "what": "98 B x all walk steps of a pass / the union of the launch spans on the solve's stream (launches beside it included "
                                       "through the spans they overlap and the wait at the end)"},
                     "pass_launches": [{"kind": l["kind_name"], "walkers": l["walkers"], "beside": l["walkers_beside"], "ms": round(l["ms"], 3),
                                        "walk_steps": l.get("steps")} for l in ll],
                     "algorithmic_bytes_per_walk_step": BYTES_PER_STEP},
    }
    if env.world > 1:
        # per-rank figures of the timed passes, so that an N-GPU record shows imbalance at a glance
        # (the event pair brackets the collective on this rank's stream: it includes the wait for the slowest rank's shard)
        exchange_ms = sum(a.elapsed_time(b) for a, b in exchange_events)
        mine = torch.tensor([kernel_ms / steps, float(steps_local) / steps, exchange_ms / max(len(exchange_events), 1)],
                            dtype=torch.float64, device=env.dev)
        allr = [torch.zeros_like(mine) for _ in range(env.world)]
        env.dist.all_gather(allr, mine)
        a = torch.stack(allr).cpu().numpy()
        out["ranks"] = {"kernel_ms_per_pass": {"min": float(a[:, 0].min()), "max": float(a[:, 0].max()), "mean": float(a[:, 0].mean())},
                        "walk_steps_per_pass": {"min": float(a[:, 1].min()), "max": float(a[:, 1].max())},
                        "exchange_ms_per_pass": {"max": float(a[:, 2].max()), "mean": float(a[:, 2].mean()),
                                                 "what": "one %s of the %d-byte field (waits for the slowest rank)" % (
                                                     "all-gather of disjoint shards" if field.numel() * 4 >= env.D.GATHER_THRESHOLD_BYTES
                                                     else "all-reduce of zero-padded frames", field.numel() * 4)}}
    # wost_create: the trees of both meshes (built by kernels in both dimensions: csrc/wost_build2.hip since round 5, wost_build3.hip;
    # configs.mesh_build2 / mesh_build3 time them against the host builders kept as their checkers), uploads, the frame's buffers
    # -- outside the solve timer like the reference's build_bvh
    out["create_ms"] = create_ms
    if t1:
        out["time_to_1spp_ms"] = {"cold": t1[0], "steady": sorted(t1[1:])[len(t1[1:]) // 2]}
    if sched["trav_trips"] and steps_local:
        # wave-level scheduler of walk_round_kernel on this rank: node visits per walk step, and how full the
        # two bodies run (a traversal trip is up to trav_burst = 3 visits of every traversing lane)
        out["scheduler"] = {"visits_per_step": sched["visits"] / steps_local,
                            "trav_lane_fill": sched["visits"] / (sched["trav_trips"] * 3.0 * 64.0),
                            "step_lane_fill": steps_local / (sched["step_trips"] * 64.0),
                            "trav_trips_per_wave_step": sched["trav_trips"] * 64.0 / steps_local,
                            "step_trips_per_wave_step": sched["step_trips"] * 64.0 / steps_local}
    res = {"out": out, "field": field, "problem": problem, "depth": depth, "eps": eps, "it": it}
    return res


def run_guided(env, scene, frame, spp, train_spp, depth, steps, warmup, args, precision=None, order=None):
    """timed solves of the guided integrator (training included: it is part of the path)"""
    torch = env.torch
    from elaina_amd import Problem
    from elaina_amd.guided import GuidedIntegrator, GuidedIntegratorSettings
    problem = Problem.load_scene(scene)
    depth = depth or problem.default_max_depth
    eps = problem.default_eps
    st = GuidedIntegratorSettings(frameSize=(frame, frame), samplesPerPixel=spp, trainSppCount=min(train_spp, spp),
                                  maxWalkingDepth=depth, epsilonShell=eps)
    field = torch.zeros(frame * frame * 3, dtype=torch.float32, device=env.dev)
    elapsed, agg = 0.0, None
    for i in range(warmup + steps):
        # a fresh integrator per pass: the network starts untrained, as in the reference's run_expr
        gi = GuidedIntegrator(problem, st, GUIDED_AABB, device=env.local)
        if args.shared_network and env.world > 1:
            gi.share_network()
        if (precision or args.net_precision) == 16:
            gi.network.set_option("precision", 16)
        if (args.net_train_precision or precision or args.net_precision) == 16:
            gi.network.set_option("train_precision", 16)
        # opt-in training orders (never the parity mode): (pipeline, train_group) of wost_guided_set_option
        pipeline, group = order if order else (args.pipeline, args.train_group)
        if pipeline:
            gi.set_option("pipeline", pipeline)
        if group > 1:
            gi.set_option("train_group", group)
        field.zero_()
        env.barrier()
        t0 = time.perf_counter()
        s = gi.solve_sharded(env.rank, env.world, field.data_ptr())
        env.D.assemble_field(field, env.world, env.rank, frame, frame)
        torch.cuda.synchronize()
        env.barrier()
        dt = time.perf_counter() - t0
        gi.close()
        if i < warmup:
            continue
        elapsed += dt
        keys = ("walk_steps", "guided_steps", "train_samples", "optimizer_steps", "net_points", "kernel_launches")
        vals = [float(s.get(k, 0)) for k in keys] + [s["train_ms"], s.get("net_infer_ms", 0.0), s["solve_ms"]]
        agg = vals if agg is None else [a + b for a, b in zip(agg, vals)]
    tot = torch.tensor(agg[:6], dtype=torch.float64, device=env.dev)
    mx = torch.tensor([elapsed] + agg[6:], dtype=torch.float64, device=env.dev)
    env.reduce(tot, env.dist.ReduceOp.SUM)
    env.reduce(mx, env.dist.ReduceOp.MAX)
    elapsed = float(mx[0].item())
    walk_steps, guided_steps, train_samples, opt_steps, net_points, launches = [float(x) for x in tot.tolist()]
    train_s, infer_s, solve_s = float(mx[1]) / 1e3, float(mx[2]) / 1e3, float(mx[3]) / 1e3
    walk_s = max(elapsed - train_s, 1e-9)
    half = (precision or args.net_precision) == 16
    probe = None
    if infer_s <= 0 and net_points > 0:
        # the solve evaluates the network inside its walk kernel (one launch per sample): the matrix fraction of the
        # network kernel is measured on a short pass of the per-depth path (same network arithmetic, launches timed apart)
        prev = os.environ.get("WOST_GUIDED_FUSED")
        os.environ["WOST_GUIDED_FUSED"] = "0"
        try:
            pst = GuidedIntegratorSettings(frameSize=(frame, frame), samplesPerPixel=min(spp, 6), trainSppCount=min(train_spp, spp, 6),
                                           maxWalkingDepth=depth, epsilonShell=eps)
            gi = GuidedIntegrator(problem, pst, GUIDED_AABB, device=env.local)
            if half:
                gi.network.set_option("precision", 16)
            scratch = torch.zeros_like(field)
            ps = gi.solve_sharded(env.rank, env.world, scratch.data_ptr())
            torch.cuda.synchronize()
            gi.close()
        finally:
            if prev is None:
                os.environ.pop("WOST_GUIDED_FUSED", None)
            else:
                os.environ["WOST_GUIDED_FUSED"] = prev
        pv = torch.tensor([float(ps.get("net_points", 0)), ps.get("net_infer_ms", 0.0)], dtype=torch.float64, device=env.dev)
        env.reduce(pv, env.dist.ReduceOp.SUM)
        if float(pv[1]) > 0:
            probe = float(pv[0]) * FLOP_PER_POINT / (float(pv[1]) * 1e-3) / 1e12       # per GPU: all points over all kernel time
            probe_points, probe_s = float(pv[0]) / max(env.world, 1), float(pv[1]) / max(env.world, 1) * 1e-3
    infer_tf = (net_points / max(env.world, 1)) * FLOP_PER_POINT / max(infer_s, 1e-9) / 1e12 if infer_s > 0 else probe
    half_train = (args.net_train_precision or precision or args.net_precision) == 16
    peak_tf = MFMA_F16_PEAK_TF if half else MFMA_F32_PEAK_TF
    # the training step: forward + backward + weight gradients = 3 x FLOP_PER_POINT per training sample
    train_tf = 3.0 * FLOP_PER_POINT * (train_samples / max(env.world, 1)) / max(train_s, 1e-9) / 1e12 if train_samples > 0 else None
    train_peak = MFMA_F16_PEAK_TF if half_train else MFMA_F32_PEAK_TF
    # the kernel the solve spends most of its time in: VALU / matrix-pipe / LDS figures from the committed PMC pass
    sample_kernel = committed_counters("guided_sample_f16" if half else "guided_sample_f32",
                                       ("kernel", "share_of_gpu_time", "pipe_busy", "lane_efficiency", "mfma_busy", "lds_conflict_ratio",
                                        "lane_instr_per_step", "source"))
    out = {
        "workload": "%s guided %dx%d grid %d spp (train %d) depth %d eps %g" % (scene, frame, frame, spp, min(train_spp, spp), depth, eps),
        "value": walk_steps / elapsed, "ms_per_step": elapsed / steps * 1e3, "walk_steps_per_pass": walk_steps / steps,
        "walk_phase_steps_per_s": walk_steps / walk_s, "train_s_per_pass": train_s / steps,
        "guided_steps_per_pass": guided_steps / steps, "optimizer_steps_per_pass": opt_steps / steps,
        "train_samples_per_pass": train_samples / steps, "kernel_launches_per_pass": launches / steps / max(env.world, 1),
        "shared_network": bool(args.shared_network and env.world > 1),
        "training_order": ("the reference's: a training pass between any two samples (the parity mode)" if not (pipeline or group > 1) else
                           "reordered (opt-in, statistically gated): %d samples per training launch%s" %
                           (group, ", their passes on a second stream while the next group walks" if pipeline else "")),
        "network_precision": ("f16 inference (v_mfma_f32_16x16x32_f16), " if half else "fp32 inference (v_mfma_f32_16x16x4_f32), ") +
                             ("f16 training passes, fp32 master weights" if half_train else "fp32 training"),
        "roofline": {"bound": "valu", "kernel": "guided_sample_kernel", "counters": sample_kernel,
                     "what": "the dominant kernel of this configuration (one launch per sample: walk, network inference on the matrix "
                             "cores and mixture sampling in one wave); the network-only figure is mfma_probe_of_net_forward_kernel (a kernel the solve does not launch)"},
        "training_step": {"ms": (train_s / max(opt_steps / max(env.world, 1), 1.0)) * 1e3 if opt_steps else None,
                          "achieved": train_tf, "peak": train_peak, "unit": "TFLOP/s", "frac": (train_tf / train_peak) if train_tf else None,
                          "flop_per_sample": 3.0 * FLOP_PER_POINT, "samples_per_step": train_samples / max(opt_steps, 1.0)},
        # the key says what it is: measured on the solve's own inference launches (the per-depth path), or a PROBE of the network
        # kernel on a short per-depth pass -- the fused solve evaluates the network inside guided_sample_kernel and launches no such kernel
        ("roofline_mfma" if infer_s > 0 else "mfma_probe_of_net_forward_kernel"): {"bound": "mfma", "kernel": ("net_forward_h_kernel" if half else "net_forward_mfma_kernel") +
                          (" (inference launches, HIP events)" if infer_s > 0 else
                           " (inference launches of a 6-spp pass of the per-depth path, HIP events; the solve itself evaluates the network inside guided_sample_kernel)"),
                          "achieved": infer_tf, "peak": peak_tf, "unit": "TFLOP/s",
                          "frac": (infer_tf / peak_tf) if infer_tf else None,
                          "flop_per_point": FLOP_PER_POINT, "points_per_pass": net_points / steps,
                          "kernel_s_per_pass": infer_s / steps if infer_s > 0 else None,
                          "probe": None if (infer_s > 0 or probe is None) else {"points": probe_points, "kernel_s": probe_s}},
    }
    return {"out": out, "field": field, "problem": problem, "depth": depth, "eps": eps}


REORDERED = (0, 16)      # (pipeline, train_group) of the configs.*_pipelined entries


def run_guiding_gain(env):
    """Does the guiding guide?  A small bright Dirichlet disc and a large dark one in a reflecting box
    (elaina_amd/scenes.py): RMSE of 64 trained + 64 guided samples against 128 uniform ones, both measured on a
    8192-sample field of the uniform integrator (tests/test_guided_integrator.py::test_gpu_guiding_reduces_the_variance
    gates the same numbers)."""
    import numpy as np
    from elaina_amd import UniformIntegrator, UniformIntegratorSettings
    from elaina_amd.guided import GuidedIntegrator, GuidedIntegratorSettings
    from elaina_amd.scenes import BRIGHT_DISC_AABB, bright_disc_scene
    p = bright_disc_scene()
    w, depth, eps = 128, 128, 0.05

    def uniform(spp):
        it = UniformIntegrator(p, UniformIntegratorSettings((w, w), spp, depth, eps), device=env.local)
        it.solve()
        f = it.solution.copy()
        it.close()
        return f

    ref, u = uniform(8192), uniform(128)
    out = {"scene": "bright disc (r 3) + dark disc (r 14) in a reflecting box, 128x128, depth 128, eps 0.05",
           "reference": "uniform integrator, 8192 spp", "rmse_uniform_128spp": float(np.sqrt(np.mean((u - ref) ** 2)))}
    for prec, order in ((32, None), (16, None), (16, REORDERED)):
        st = GuidedIntegratorSettings(frameSize=(w, w), samplesPerPixel=128, trainSppCount=64, maxWalkingDepth=depth, epsilonShell=eps,
                                      batchSize=65536, minBatchSize=8192)
        g = GuidedIntegrator(p, st, BRIGHT_DISC_AABB, device=env.local)
        if prec == 16:
            g.network.set_option("precision", 16)
            g.network.set_option("train_precision", 16)
        if order:
            g.set_option("pipeline", order[0])
            g.set_option("train_group", order[1])
        g.solve()
        r = float(np.sqrt(np.mean((g.solution - ref) ** 2)))
        tag = "f%d%s" % (prec, "_pipelined" if order else "")
        out["rmse_guided_%s_64+64spp" % tag] = r
        out["ratio_%s" % tag] = r / out["rmse_uniform_128spp"]
        g.close()
    return out


def icosphere(subdiv, radius):
    """unit icosphere subdivided `subdiv` times (20 * 4^subdiv triangles, outward normals), scaled"""
    import numpy as np
    t = (1.0 + 5.0 ** 0.5) / 2.0
    v = [(-1, t, 0), (1, t, 0), (-1, -t, 0), (1, -t, 0), (0, -1, t), (0, 1, t), (0, -1, -t), (0, 1, -t), (t, 0, -1), (t, 0, 1),
         (-t, 0, -1), (-t, 0, 1)]
    f = [(0, 11, 5), (0, 5, 1), (0, 1, 7), (0, 7, 10), (0, 10, 11), (1, 5, 9), (5, 11, 4), (11, 10, 2), (10, 7, 6), (7, 1, 8),
         (3, 9, 4), (3, 4, 2), (3, 2, 6), (3, 6, 8), (3, 8, 9), (4, 9, 5), (2, 4, 11), (6, 2, 10), (8, 6, 7), (9, 8, 1)]
    verts = [np.asarray(p, np.float64) / np.linalg.norm(p) for p in v]
    for _ in range(subdiv):
        cache, nf = {}, []
        for a, b, c in f:
            m = []
            for x, y in ((a, b), (b, c), (c, a)):
                k = (min(x, y), max(x, y))
                if k not in cache:
                    w = verts[x] + verts[y]
                    verts.append(w / np.linalg.norm(w))
                    cache[k] = len(verts) - 1
                m.append(cache[k])
            nf += [(a, m[0], m[2]), (b, m[1], m[0]), (c, m[2], m[1]), (m[0], m[1], m[2])]
        f = nf
    return (np.asarray(verts) * radius).astype(np.float32), np.asarray(f, np.int32)


def run_uniform3d(env, args, only=None):
    """SURVEY 8 f.3, the 3-D uniform integrator, synthetic scenes (the reference ships no 3-D data): (a) a Dirichlet
    icosphere of 1280 triangles with the harmonic boundary values x y + z, the slice z = 0.1 (tools/probes/bench3d.py's scene); (b) a Dirichlet ball
    inside a zero-flux Neumann shell of 1280 triangles (silhouette and ray queries on the tree).  One timed solve each
    after a warm-up solve; a band of (a) against the oracle."""
    import numpy as np
    from elaina_amd import UniformIntegratorSettings
    from elaina_amd.integrator3d import Problem3, UniformIntegrator3
    out = {}
    V, T = icosphere(3, 1.0)
    col = np.repeat((V[:, 0] * V[:, 1] + V[:, 2]).astype(np.float32)[:, None], 6, axis=1)
    ball = {"d_verts": V, "d_tris": T, "d_colors": col, "n_verts": None, "n_tris": None, "n_colors": None,
            "probe": (0.6, (0.0, 0.0, 0.1), (0.0, 1.0, 0.0), (1.0, 0.0, 0.0)), "dirichlet_intensity": 1.0, "neumann_intensity": 1.0}
    Vi, Ti = icosphere(2, 0.45)
    shell = {"d_verts": Vi, "d_tris": Ti, "d_colors": np.repeat(Vi[:, :1], 6, axis=1).astype(np.float32), "n_verts": V, "n_tris": T,
             "n_colors": np.zeros((len(V), 6), np.float32), "probe": (0.7, (0.0, 0.0, 0.0), (0.0, 1.0, 0.0), (1.0, 0.0, 0.0)),
             "dirichlet_intensity": 1.0, "neumann_intensity": 1.0}
    # (the 512^2 frames of the earlier rounds hold fewer walkers than the chip has lanes at six waves per SIMD -- latency-bound;
    # the 1024^2 frames, a million walkers, are the full-chip figure)
    for name, sd, frame, spp in (("dirichlet_icosphere_1280", ball, 512, 64), ("neumann_shell_1280", shell, 512, 16),
                                 ("dirichlet_icosphere_1280_1024", ball, 1024, 16), ("neumann_shell_1280_1024", shell, 1024, 4)):
        if only is not None and name not in only:
            continue
        it = UniformIntegrator3(Problem3.from_dict(sd), UniformIntegratorSettings((frame, frame), spp, 64, 2e-3), device=env.local)
        it.solve()
        it.solve()
        st = it.last_stats
        e = {"workload": "%s %dx%d %d spp depth 64 eps 2e-3" % (name, frame, frame, spp), "walk_steps": float(st["walk_steps"]),
             "kernel_ms": float(st["kernel_ms"]), "value": st["walk_steps"] / (st["kernel_ms"] * 1e-3), "unit": "walk-steps/s"}
        if name == "dirichlet_icosphere_1280" and not args.no_cpu_baseline:
            from oracle.oracle import Oracle
            b, e_ = band_of(frame, 4)
            ref = Oracle().solve3(sd, frame, frame, spp, 64, 2e-3, pixel_begin=b, pixel_end=e_, threads=os.cpu_count() or 1)
            e["rel_l2_vs_oracle"] = rel_l2(it.solution.reshape(-1, 3)[b:e_], ref["field"])
            e["rel_l2_band"] = "rows %d..%d" % (b // frame, e_ // frame)
        it.close()
        # what binds walk3_kernel on this scene: VALU figures of the committed PMC pass (tools/gpu_round.sh, stage pmc3d)
        big = name.endswith("_1024")
        cc = committed_counters("walk3_valu_1024" if big else "walk3_valu", ("scenes", "source"))
        name_cc = name[:-5] if big else name
        if cc and cc.get("scenes") and name_cc in cc["scenes"]:
            e["roofline"] = dict(cc["scenes"][name_cc], bound="valu", kernel="walk3_kernel", stale=cc["stale"], measured_on_sources=cc["measured_on_sources"],
                                 source=cc["source"])
        out[name] = e
    return out


def run_mesh_build3(env):
    """Problem<3>::build_bvh on the device (csrc/wost_build3.hip): a bumpy sphere of 81 920 triangles in shuffled index order, the
    device build against the host builder kept as its checker -- differing bytes per array, fastest of five builds each"""
    import numpy as np
    from elaina_amd.integrator3d import mesh_build_check
    rng = np.random.default_rng(4)
    V, T = icosphere(6, 1.0)
    V = (V.astype(np.float64) * (1.0 + 0.15 * np.sin(3 * V[:, :1] + 1.0) * np.cos(4 * V[:, 1:2] + 2.0))).astype(np.float32)
    T = np.ascontiguousarray(rng.permutation(T), np.int32)
    diff, compared, host_ms, dev_ms = mesh_build_check(V, T, None, repeat=5, device=env.local)
    return {"workload": "bumpy icosphere, %d triangles, %d vertices" % (len(T), len(V)), "device_build_ms": dev_ms, "host_build_ms": host_ms,
            "bytes_compared": compared, "differing_bytes": sum(diff.values()),
            "what": "wall clock of one mesh build, uploads and the final wait included; wost3_create builds on the device"}


def run_mesh_build2(env):
    """Problem<2>::build_bvh on the device (csrc/wost_build2.hip): the Dirichlet trees of the two BASELINE scenes, the device build
    against the host builder kept as its checker -- differing bytes over all arrays, fastest of five builds each way"""
    from elaina_amd import Problem
    from elaina_amd.integrator import mesh_build_check
    out = {}
    for name in ("ladybug", "fille"):
        p = Problem.load_scene(name)
        diff, compared, host_ms, dev_ms = mesh_build_check(p.d_verts, p.d_segs, p.d_colors, repeat=5, device=env.local)
        out[name] = {"segments": int(len(p.d_segs)), "device_build_ms": dev_ms, "host_build_ms": host_ms, "bytes_compared": compared,
                     "differing_bytes": sum(diff.values())}
    out["what"] = "wall clock of one tree build, uploads and the final wait included; wost_create builds on the device"
    return out


def run_neumann2d(env, args):
    """A Neumann boundary too large for the flat loops (its silhouette and ray queries on the tree, answered by the wave:
    wost_coop.h): the non-convex closed curve r(t) = 100 (1 + .2 sin 7t + .05 sin 31t) in 3000 segments, zero flux, around a
    Dirichlet circle of 400 segments; 512^2, 64 spp, depth 64 (tools/probes/bench2d_coop.py's scene).  One timed solve after a
    warm-up solve; a band against the oracle."""
    import numpy as np
    from elaina_amd import Problem, UniformIntegrator, UniformIntegratorSettings
    nn, nd = 3000, 400
    t = np.linspace(0.0, 2.0 * np.pi, nn, endpoint=False)
    r = 100.0 * (1.0 + 0.2 * np.sin(7 * t) + 0.05 * np.sin(31 * t))
    nv = np.stack([r * np.cos(t), r * np.sin(t)], 1).astype(np.float32)
    ns = np.stack([np.arange(nn), (np.arange(nn) + 1) % nn], 1).astype(np.int32)
    td = np.linspace(0.0, 2.0 * np.pi, nd, endpoint=False)
    dv = np.stack([15.0 * np.cos(td) + 5.0, 15.0 * np.sin(td) - 3.0], 1).astype(np.float32)
    ds = np.stack([np.arange(nd), (np.arange(nd) + 1) % nd], 1).astype(np.int32)
    dc = np.zeros((nd, 6), np.float32)
    dc[:, 0:3] = (0.5 + 0.5 * np.cos(td))[:, None] * np.array([1.0, 0.5, 0.25])
    dc[:, 3:6] = 0.3
    p = Problem(d_verts=dv, d_segs=ds, d_colors=dc, n_verts=nv, n_segs=ns, n_colors=None, probe=(110.0, 0.0, 0.0, 0.0, 1.0))
    frame, spp, depth, eps = 512, 64, 64, 0.05
    it = UniformIntegrator(p, UniformIntegratorSettings((frame, frame), spp, depth, eps), device=env.local)
    it.solve()
    it.solve()
    st = it.last_stats
    e = {"workload": "3000-segment zero-flux Neumann curve around a Dirichlet circle, %dx%d %d spp depth %d eps %g" % (frame, frame, spp, depth, eps),
         "walk_steps": float(st["walk_steps"]), "kernel_ms": float(st["kernel_ms"]), "value": st["walk_steps"] / (st["kernel_ms"] * 1e-3),
         "unit": "walk-steps/s"}
    if not args.no_cpu_baseline:
        from oracle.oracle import Oracle
        b, e_ = band_of(frame, 1)
        ref = Oracle().solve(p.as_dict(), frame, frame, spp, depth, eps, pixel_begin=b, pixel_end=e_, threads=os.cpu_count() or 1)
        e["rel_l2_vs_oracle"] = rel_l2(it.solution.reshape(-1, 3)[b:e_], ref["field"])
        e["rel_l2_band"] = "rows %d..%d" % (b // frame, e_ // frame)
    it.close()
    return e


def run_guided3d(env, args):
    """GuidedIntegrator<3> on the two 3-D bench scenes at 256^2, 16 spp (8 of them trained) and at 1024^2, 4 spp (2 trained), depth 64:
    the whole solve (walks, network inference, training), as walk-steps per second of wall time.  (The roofline counters of the 256^2
    entries come from one profiled run of all four solves: g3_fused_kernel is the 256^2 frames', g3_separate / g3_sample / g3_tail the 1024^2 frames'.)"""
    import numpy as np
    from elaina_amd.guided import GuidedIntegratorSettings
    from elaina_amd.integrator3d import GuidedIntegrator3, Problem3, default_net_config3
    out = {}
    V, T = icosphere(3, 1.0)
    col = np.repeat((V[:, 0] * V[:, 1] + V[:, 2]).astype(np.float32)[:, None], 6, axis=1)
    ball = {"d_verts": V, "d_tris": T, "d_colors": col, "n_verts": None, "n_tris": None, "n_colors": None,
            "probe": (0.6, (0.0, 0.0, 0.1), (0.0, 1.0, 0.0), (1.0, 0.0, 0.0)), "dirichlet_intensity": 1.0, "neumann_intensity": 1.0}
    Vi, Ti = icosphere(2, 0.45)
    shell = {"d_verts": Vi, "d_tris": Ti, "d_colors": np.repeat(Vi[:, :1], 6, axis=1).astype(np.float32), "n_verts": V, "n_tris": T,
             "n_colors": np.zeros((len(V), 6), np.float32), "probe": (0.7, (0.0, 0.0, 0.0), (0.0, 1.0, 0.0), (1.0, 0.0, 0.0)),
             "dirichlet_intensity": 1.0, "neumann_intensity": 1.0}
    # (256^2 = 65 536 walkers is a quarter of the chip's lanes at one wave per SIMD: latency-bound; the 1024^2 frames are the
    # full-chip figure, as for the uniform integrator)
    for name, sd, frame, spp in (("dirichlet_icosphere_1280", ball, 256, 16), ("neumann_shell_1280", shell, 256, 16),
                                 ("dirichlet_icosphere_1280_1024", ball, 1024, 4), ("neumann_shell_1280_1024", shell, 1024, 4)):
        st = GuidedIntegratorSettings(frameSize=(frame, frame), samplesPerPixel=spp, trainSppCount=spp // 2, maxWalkingDepth=64, epsilonShell=2e-3)
        gi = GuidedIntegrator3(Problem3.from_dict(sd), st, ((-1.1, -1.1, -1.1), (1.1, 1.1, 1.1)), network_config=default_net_config3(), seed=7,
                               device=env.local)
        dt = None
        for _ in range(2):
            t0 = time.perf_counter()
            gi.solve()
            dt = time.perf_counter() - t0
        g = gi.last_stats
        out[name] = {"workload": "%s guided %dx%d %d spp (train %d) depth 64 eps 2e-3" % (name, frame, frame, spp, spp // 2),
                     "walk_steps": float(g["walk_steps"]), "guided_steps": float(g["guided_steps"]), "optimizer_steps": float(g["optimizer_steps"]),
                     "ms_per_step": dt * 1e3, "value": g["walk_steps"] / dt, "unit": "walk-steps/s", "kernel_launches": float(g["kernel_launches"]),
                     "form": "one launch per sample (g3_fused_kernel)" if frame == 256 else "launches per depth (the frame fills the chip)",
                     "field_finite": bool(np.isfinite(gi.solution).all())}
        gi.close()
        if frame != 256:
            continue
        # (parity of GuidedIntegrator<3> is the business of tests/test_guided_3d.py -- frozen and trained solves against the oracle bit
        # for bit; a band of this frame through the oracle's scalar network cost the bench nine minutes of host time in round 4)
        cc = committed_counters("guided3d_valu", ("scenes", "source"))
        if cc and cc.get("scenes") and name in cc["scenes"]:
            out[name]["roofline"] = dict(cc["scenes"][name], bound="valu", stale=cc["stale"], measured_on_sources=cc["measured_on_sources"],
                                         source=cc["source"])
    return out


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args))

    env = Env(args)
    cfg = dict(CONFIGS[args.config])
    if args.scene:
        cfg["scene"] = args.scene
    if args.frame:
        cfg["frame"] = args.frame
    if args.spp:
        cfg["spp"] = args.spp
    if args.train_spp >= 0:
        cfg["train_spp"] = args.train_spp
    import numpy as np

    line = {"metric": "walk-steps/s", "unit": "walk-steps/s", "n_gpus": env.world, "steps": args.steps, "warmup": args.warmup,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic"}
    if env.world > 1:
        line["world_size"] = env.dist.get_world_size()
        line["backend"] = env.backend
    uniform_field = None
    if cfg["kind"] == "uniform":
        r = run_uniform(env, cfg["scene"], cfg["frame"], cfg["spp"], args.depth, args.steps, args.warmup, args,
                        one_spp=not args.no_1spp)
        o = r["out"]
        line.update({"value": o["value"], "ms_per_step": o["ms_per_step"],
                     "config": {"workload": o["workload"], "parallelism": "pixel-tiles x%d" % env.world,
                                "walk_steps_per_pass": o["walk_steps_per_pass"], "config": args.config, "walk_step_note": o.get("walk_step_note")}})
        if "time_to_1spp_ms" in o:
            line["time_to_1spp_ms"] = o["time_to_1spp_ms"]
        if "create_ms" in o:
            # wost_create of the workload's scene (trees built on the device since round 5; outside the timed passes like the reference's loadConfig / build_bvh)
            line["create_ms"] = o["create_ms"]
        if "scheduler" in o:
            line["scheduler"] = o["scheduler"]
        if "ranks" in o:
            line["ranks"] = o["ranks"]
        roof = o["roofline"]
        if env.rank == 0:
            # PMC passes cannot run inside this process: the figure comes from the committed
            # rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE runs of this same command (tools/gpu_round.sh)
            tr = committed_counters("walk_round_traffic", ("hbm_bytes_per_launch",)) if args.config == 2 else None
            roof["traffic"] = tr["hbm_bytes_per_launch"] if tr else None
            roof["traffic_unit"] = "HBM bytes per launch (profiles/walk_round_traffic.json)"
            roof["traffic_stale"] = tr["stale"] if tr else None
            roof["valu"] = valu_block() if args.config == 2 else None
            v = roof["valu"]
            if v and v.get("lane_instr_per_step") and roof.get("launch"):
                # the VALU roofline of the same launch: 1024 SIMDs x 16 lanes x 2.4 GHz lane-instructions per second / the lane
                # instructions a walk step takes (counters of the committed PMC passes) = the walk steps per second the chip could
                # make if every lane of every VALU instruction were useful
                peak_steps = 1024 * 16 * 2.4e9 / v["lane_instr_per_step"]
                roof["valu_frac"] = (roof["launch"]["walk_steps"] / (roof["launch"]["ms"] * 1e-3)) / peak_steps
                roof["valu_peak_walk_steps_per_s"] = peak_steps
            # achieved / peak / frac are the SURVEY 8(d) HBM formula (rounds remain comparable); "what_binds" and "valu" say what the counters
            # say binds the kernel instead
            line["roofline"] = roof
            if not args.no_cpu_baseline:
                # the host baseline is a reported figure at N = 1 only; at N > 1 a short band of the
                # assembled field is still checked against the oracle
                base, (b, e, ref_field) = cpu_baseline(r["problem"], cfg["frame"], cfg["spp"], r["depth"], r["eps"],
                                                       target_s=15.0 if env.world == 1 else 2.0)
                if env.world == 1:
                    line["cpu_baseline"] = base
                got = r["field"].cpu().numpy().reshape(-1, 3)[b:e]
                line["rel_l2_vs_oracle"] = rel_l2(got, ref_field)
        uniform_field = r["field"] if (args.config == 2 and cfg["frame"] == 1024 and cfg["scene"] == "ladybug") else None
        r["it"].close()
    else:
        r = run_guided(env, cfg["scene"], cfg["frame"], cfg["spp"], cfg["train_spp"], args.depth, args.steps, args.warmup, args)
        o = r["out"]
        line.update({"value": o["value"], "ms_per_step": o["ms_per_step"],
                     "config": {"workload": o["workload"], "parallelism": "pixel-tiles x%d" % env.world,
                                "walk_steps_per_pass": o["walk_steps_per_pass"], "config": args.config},
                     "guided": {k: o[k] for k in o if k not in ("workload", "value", "ms_per_step", "walk_steps_per_pass", "roofline_mfma",
                                                                "mfma_probe_of_net_forward_kernel", "roofline")},
                     "roofline": o["roofline"]})
        for k in ("roofline_mfma", "mfma_probe_of_net_forward_kernel"):
            if k in o:
                line[k] = o[k]
        if env.rank == 0:
            f = r["field"].cpu().numpy()
            line["field_mean"] = float(f.mean())
            line["field_finite"] = bool(np.isfinite(f).all())

    # ---- one pass each of the other single-GPU configurations (driver-run evidence for configs 3, 4, 5) ----
    # The headline is printed as soon as it is measured (a run that is cut off keeps it) and again, complete, at the end; the
    # extras run most important first inside a wall-clock budget (--budget-s, counted from process start): one that would start
    # past it is named in "skipped" instead of run.
    extras = {}
    want_extras = not args.no_extras and args.config == 2 and not (args.scene or args.frame or args.spp)
    if env.rank == 0 and want_extras:
        # (on stderr: stdout carries exactly one JSON line, the complete one)
        print(json.dumps(dict(line, partial="headline only; the complete line follows on stdout")), file=sys.stderr, flush=True)
    wall, skipped = {}, []

    if want_extras:
        if env.world == 1:
            fields = {}

            def cfg3():
                r3 = run_uniform(env, "fille", 1024, 256, 0, 1, 0, args, one_spp=True)
                e3 = r3["out"]
                if not args.no_cpu_baseline:
                    from oracle.oracle import Oracle
                    b, e = band_of(1024, 4)
                    ref = Oracle().solve(r3["problem"].as_dict(), 1024, 1024, 256, r3["depth"], r3["eps"], pixel_begin=b, pixel_end=e,
                                         threads=os.cpu_count() or 1)
                    e3["rel_l2_vs_oracle"] = rel_l2(r3["field"].cpu().numpy().reshape(-1, 3)[b:e], ref["field"])
                    e3["rel_l2_band"] = "rows %d..%d" % (b // 1024, e // 1024)
                r3["it"].close()
                return e3

            def cfg4(tag, **kw):
                def run():
                    r4 = run_guided(env, "ladybug", 1024, 256, 256, 0, 1, 0, args, **kw)
                    e4 = r4["out"]
                    fields[tag] = r4["field"].cpu().numpy()
                    if uniform_field is not None:
                        # the guided estimator is unbiased: its field agrees with the uniform integrator's
                        # (bit-exact against the oracle above) up to the Monte-Carlo noise of 256 spp
                        e4["rel_l2_vs_uniform_field"] = rel_l2(fields[tag], uniform_field.cpu().numpy())
                    return e4
                return run

            def variance_check():
                # SURVEY 8c, guided gate: against a 4096-spp field of the uniform integrator (bit-exact against the
                # oracle at any spp) the guided estimator must not be noisier than the uniform one at equal spp
                from elaina_amd import UniformIntegrator, UniformIntegratorSettings
                itr = UniformIntegrator(r["problem"], UniformIntegratorSettings((1024, 1024), 4096, r["depth"], r["eps"]), device=env.local)
                ref = torch_zeros_like(uniform_field)
                itr.solve_sharded(0, 1, ref.data_ptr(), env.torch.cuda.current_stream().cuda_stream)
                env.torch.cuda.synchronize()
                itr.close()
                refn = ref.cpu().numpy()
                out = {"reference": "ladybug 1024x1024, uniform integrator, 4096 spp",
                       "rel_l2_uniform_256spp": rel_l2(uniform_field.cpu().numpy(), refn)}
                for tag, key in (("cfg4", "guided"), ("cfg4_f16", "guided_f16"), ("cfg4_f16_pipelined", "guided_f16_pipelined")):
                    if tag in fields:
                        out["rel_l2_%s_256spp" % key] = rel_l2(fields[tag], refn)
                return out

            def cfg2_rounds():
                """the headline configuration as it ran until round 5: rounds only (wost_set_option persist 0) -- one pass, the same field"""
                import copy
                a2 = copy.copy(args)
                a2.opt = list(args.opt) + ["persist=0"]
                r2 = run_uniform(env, "ladybug", 1024, 256, 0, 1, 0, a2, one_spp=False)
                e2 = {k: r2["out"][k] for k in ("workload", "value", "ms_per_step")}
                e2["launches"] = r2["out"]["roofline"]["launches"]
                e2["what"] = "config 2 in rounds only (persist = 0): what the persistent first launch of round 6 replaced"
                if uniform_field is not None:
                    e2["field_identical_to_the_headline_solve"] = bool((r2["field"] == uniform_field).all().item())
                r2["it"].close()
                return e2

            todo = [("cfg3", cfg3), ("cfg2_rounds_only", cfg2_rounds), ("cfg4", cfg4("cfg4")),
                    # the same configuration with the reference's half-precision network (tolerance-gated mode)
                    ("cfg4_f16", cfg4("cfg4_f16", precision=16)),
                    # the same in the opt-in reordered training order (sixteen samples per training launch; cfg4 / cfg4_f16 stay exact-order)
                    ("cfg4_f16_pipelined", cfg4("cfg4_f16_pipelined", precision=16, order=REORDERED)),
                    ("uniform3d", lambda: run_uniform3d(env, args)), ("guided3d", lambda: run_guided3d(env, args)),
                    ("neumann2d", lambda: run_neumann2d(env, args)), ("mesh_build2", lambda: run_mesh_build2(env)),
                    ("mesh_build3", lambda: run_mesh_build3(env)),
                    ("guiding_gain", lambda: run_guiding_gain(env))]
            if uniform_field is not None:
                todo.append(("variance_check", variance_check))
        else:
            todo = [("cfg5", lambda: run_guided(env, "ladybug", 2048, 1024, 256, 0, 1, 0, args)["out"]),
                    ("cfg5_f16", lambda: run_guided(env, "ladybug", 2048, 1024, 256, 0, 1, 0, args, precision=16)["out"])]
        for name, fn in todo:
            go = time.time() - T_START <= args.budget_s
            if env.world > 1:
                # every rank takes rank 0's decision, or a skipped collective would hang the others
                flag = env.torch.tensor([1 if go else 0], device="cuda")
                env.dist.broadcast(flag, 0)
                go = bool(int(flag.item()))
            if not go:
                skipped.append(name)
                continue
            t0 = time.time()
            extras[name] = fn()
            wall[name] = round(time.time() - t0, 1)
            print("bench extra %s: %.1f s (%.0f s since start)" % (name, wall[name], time.time() - T_START), file=sys.stderr, flush=True)
        extras["wall_s"] = wall
        if skipped:
            extras["skipped"] = {"names": skipped, "why": "would have started past --budget-s %g of wall clock" % args.budget_s}
    if env.rank == 0:
        if extras:
            line["configs"] = extras
        print(json.dumps(line), flush=True)
    if env.world > 1:
        env.dist.barrier()
        env.dist.destroy_process_group()


if __name__ == "__main__":
    main()
