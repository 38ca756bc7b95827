"""Guided integrator runner.  One GPU: BASELINE config 4 (ladybug, guided integrator with online
training, 1024^2, 256 spp).  Several GPUs (config 5): launch with
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 tools/gpu_guided_bench.py --frame 2048 --spp 1024 --train-spp 256
every rank owns the 8x8 pixel tiles t % N == rank and trains its own network;
the fields are summed with one RCCL all-reduce.  Prints one JSON line on rank 0.
Usage: python tools/gpu_guided_bench.py [--scene ladybug] [--frame 1024] [--spp 256] [--train-spp 256] [--depth 64]"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import torch  # noqa: E402

from elaina_amd import Problem  # noqa: E402
from elaina_amd import distributed as D  # noqa: E402
from elaina_amd.guided import GuidedIntegrator, GuidedIntegratorSettings  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--scene", default="ladybug")
ap.add_argument("--frame", type=int, default=1024)
ap.add_argument("--spp", type=int, default=256)
ap.add_argument("--train-spp", type=int, default=256)
ap.add_argument("--depth", type=int, default=64)
ap.add_argument("--guided-depth", type=int, default=10)
ap.add_argument("--backend", default=None)
ap.add_argument("--shared-network", action="store_true", help="one network for all ranks (gradient all-reduce)")
ap.add_argument("--batch", type=int, default=65536 * 8)
ap.add_argument("--min-batch", type=int, default=65536)
ap.add_argument("--one-shard-of", type=int, default=0,
                help="single process: run only shard 0 of N (what ONE rank of an N-GPU job does, no reduce)")
ap.add_argument("--net-precision", type=int, default=32, choices=[32, 16])
ap.add_argument("--net-train-precision", type=int, default=0, choices=[0, 32, 16])
ap.add_argument("--pipeline", type=int, default=0, help="1: the pipelined training order (wost_guided_set_option)")
ap.add_argument("--train-group", type=int, default=1)
ap.add_argument("--opt", action="append", default=[], help="key=value passed to wost_guided_set_option")
ap.add_argument("--net-opt", action="append", default=[], help="key=value passed to wost_net_set_option")
ap.add_argument("--repeat", type=int, default=1, help="solves (fresh integrator each); the last one is reported")
a = ap.parse_args()

rank, world, local = D.init_process_group(a.backend)
shard_world = a.one_shard_of if (a.one_shard_of > 1 and world == 1) else world
device = local % max(torch.cuda.device_count(), 1)
torch.cuda.set_device(device)
prob = Problem.load_scene(a.scene)
st = GuidedIntegratorSettings(frameSize=(a.frame, a.frame), samplesPerPixel=a.spp, trainSppCount=a.train_spp,
                              maxWalkingDepth=a.depth, epsilonShell=1.0, maxGuidedDepthInTrainingPhase=a.guided_depth,
                              maxGuidedDepthInGuidingPhase=a.guided_depth, batchSize=a.batch, minBatchSize=a.min_batch)
field = torch.zeros(a.frame * a.frame * 3, dtype=torch.float32, device="cuda")
if world > 1:
    import torch.distributed as dist
for rep in range(a.repeat):
    t0 = time.time()
    gi = GuidedIntegrator(prob, st, ((-100.0, -100.0), (600.0, 600.0)), device=device)
    t_create = time.time() - t0
    if a.shared_network and world > 1:
        gi.share_network()
    if a.net_precision == 16:
        gi.network.set_option("precision", 16)
    if (a.net_train_precision or a.net_precision) == 16:
        gi.network.set_option("train_precision", 16)
    if a.pipeline:
        gi.set_option("pipeline", a.pipeline)
    if a.train_group > 1:
        gi.set_option("train_group", a.train_group)
    for kv in a.opt:
        k, v = kv.split("=")
        gi.set_option(k, float(v))
    for kv in a.net_opt:
        k, v = kv.split("=")
        gi.network.set_option(k, float(v))
    field.zero_()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    s = gi.solve_sharded(rank, shard_world, field.data_ptr())
    D.reduce_field(field, world)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if rep + 1 < a.repeat:
        gi.close()
tot = torch.tensor([float(s["walk_steps"]), float(s["guided_steps"]), float(s["train_samples"]), float(s["optimizer_steps"])],
                   dtype=torch.float64, device="cuda")
mx = torch.tensor([elapsed, s["train_ms"] / 1e3], dtype=torch.float64, device="cuda")
if world > 1:
    dist.all_reduce(tot, op=dist.ReduceOp.SUM)
    dist.all_reduce(mx, op=dist.ReduceOp.MAX)
# are the ranks' networks the same? (they are with --shared-network, they are not without)
psum = torch.tensor([float(abs(gi.network.params()).sum()), float(gi.network.params()[::97].sum())], dtype=torch.float64, device="cuda")
pmin, pmax = psum.clone(), psum.clone()
if world > 1:
    dist.all_reduce(pmin, op=dist.ReduceOp.MIN)
    dist.all_reduce(pmax, op=dist.ReduceOp.MAX)
if rank == 0:
    print(json.dumps({
        "networks_identical": bool((pmin == pmax).all().item()), "shared_network": bool(a.shared_network and world > 1),
        "workload": "%s guided %dx%d %d spp (train %d) depth %d" % (a.scene, a.frame, a.frame, a.spp, a.train_spp, a.depth),
        "pipeline": a.pipeline, "train_group": a.train_group, "n_gpus": world, "shard": "%d of %d" % (rank, shard_world), "solve_s": float(mx[0]), "train_s": float(mx[1]), "create_s": t_create,
        "walk_steps": int(tot[0]), "walk_steps_per_s": float(tot[0]) / float(mx[0]), "guided_steps": int(tot[1]),
        "train_samples": int(tot[2]), "optimizer_steps_all_ranks": int(tot[3]), "kernel_launches": s["kernel_launches"],
        "mean": float(field.mean().item()), "field_crc": int(torch.frombuffer(bytearray(field.cpu().numpy().tobytes()), dtype=torch.int32).to(torch.int64).sum().item()), "opts": a.opt + a.net_opt, "params_crc": int(torch.frombuffer(bytearray(gi.network.params().tobytes()), dtype=torch.int32).to(torch.int64).sum().item()),
    }))
if world > 1:
    dist.barrier()
    dist.destroy_process_group()
gi.close()
