"""gloo tests of the N>1 glue on CPU, 2 and 8 ranks: tile ownership, the one exchange of the field (all-reduce of
zero-padded frames or all-gather of the disjoint shards), the shared network's gradient sum.
The per-rank compute stand-in here is the oracle (allowed: tests only)."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, w, h, spp, depth, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    import torch
    import torch.distributed as dist
    from elaina_amd import Problem
    from elaina_amd import distributed as D
    from oracle.oracle import Oracle
    r, wsz, _ = D.init_process_group("gloo")
    p = Problem.load_scene("ladybug")
    sd = p.as_dict()
    own = D.owned_mask(w, h, r, wsz)
    sd["mask"] = own.astype(np.uint8)          # oracle stand-in: compute only owned pixels
    res = Oracle().solve(sd, w, h, spp, depth, 1.0, threads=2)
    field = torch.from_numpy(res["field"].reshape(-1).copy())
    D.reduce_field(field, wsz)
    steps = torch.tensor([res["walk_steps"]], dtype=torch.int64)
    dist.all_reduce(steps)
    if r == 0:
        q.put((field.numpy().reshape(-1, 3), int(steps.item())))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharded_solve_reduces_to_the_full_field(oracle, ladybug):
    import torch.multiprocessing as mp
    w, h, spp, depth = 40, 24, 3, 24
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, w, h, spp, depth, q)) for r in range(2)]
    for p in procs:
        p.start()
    field, steps = q.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    ref = oracle.solve(ladybug.as_dict(), w, h, spp, depth, 1.0)
    assert steps == ref["walk_steps"]
    assert np.array_equal(field, ref["field"])


def test_ownership_partitions_the_frame():
    from elaina_amd import distributed as D
    for w, h in ((64, 64), (37, 29), (1024, 1024)):
        for n in (1, 2, 3, 8):
            total = np.zeros(w * h, dtype=np.int32)
            sizes = []
            for r in range(n):
                m = D.owned_mask(w, h, r, n)
                total += m
                sizes.append(int(m.sum()))
            assert np.all(total == 1)
            if w * h >= 64 * 64:
                assert max(sizes) - min(sizes) <= 64 * ((h + 7) // 8)


def _guided_worker(rank, world, port, w, h, spp, depth, q):
    """guided integrator, BASELINE config 5 in miniature: every rank owns its tiles AND its own
    network; with a frozen network the union is exactly the single-process field"""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    import torch
    import torch.distributed as dist
    from conftest import box_problem
    from elaina_amd import distributed as D
    from oracle.oracle import Oracle, default_net_config, guided_settings
    r, wsz, _ = D.init_process_group("gloo")
    prob = box_problem(d_sides=(0, 2), n_sides=(1, 3), value=lambda x, y: y, flux=lambda x, y, s: 0.0, n_per_side=8)
    sd = prob.as_dict()
    sd["mask"] = D.owned_mask(w, h, r, wsz).astype(np.uint8)
    o = Oracle()
    cfg = default_net_config()
    params = np.random.default_rng(5).uniform(-0.3, 0.3, o.net_n_params(cfg)).astype(np.float32)
    gs = guided_settings(w, h, spp, depth, 1e-3, (-0.1, -0.1), (1.1, 1.1), train_spp_count=0)
    res = o.solve_guided(sd, gs, cfg, params, threads=2)
    field = torch.from_numpy(res["field"].reshape(-1).copy())
    D.reduce_field(field, wsz)
    steps = torch.tensor([res["walk_steps"], res["guided_steps"]], dtype=torch.int64)
    dist.all_reduce(steps)
    if r == 0:
        q.put((field.numpy().reshape(-1, 3), steps.tolist()))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_guided_solve_with_frozen_network(oracle):
    import torch.multiprocessing as mp
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from conftest import box_problem
    from oracle.oracle import default_net_config, guided_settings
    w, h, spp, depth = 24, 16, 2, 24
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_guided_worker, args=(r, 2, port, w, h, spp, depth, q)) for r in range(2)]
    for p in procs:
        p.start()
    field, (steps, guided) = q.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    prob = box_problem(d_sides=(0, 2), n_sides=(1, 3), value=lambda x, y: y, flux=lambda x, y, s: 0.0, n_per_side=8)
    cfg = default_net_config()
    params = np.random.default_rng(5).uniform(-0.3, 0.3, oracle.net_n_params(cfg)).astype(np.float32)
    gs = guided_settings(w, h, spp, depth, 1e-3, (-0.1, -0.1), (1.1, 1.1), train_spp_count=0)
    ref = oracle.solve_guided(prob.as_dict(), gs, cfg, params, threads=4)
    assert steps == ref["walk_steps"] and guided == ref["guided_steps"]
    assert np.array_equal(field, ref["field"])


def _eight_worker(rank, world, port, w, h, spp, depth, q):
    """8 ranks: the field assembled both ways (all-reduce of zero-padded frames, all-gather of the disjoint shards), a
    frozen-network guided solve, and the shared network's callback body (integer gradient sum, MIN, rank count)"""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    import ctypes as C
    import torch
    import torch.distributed as dist
    from conftest import box_problem
    from elaina_amd import Problem
    from elaina_amd import distributed as D
    from oracle.oracle import Oracle, default_net_config, guided_settings
    torch.set_num_threads(1)
    r, wsz, _ = D.init_process_group("gloo")
    o = Oracle()
    sd = Problem.load_scene("ladybug").as_dict()
    sd["mask"] = D.owned_mask(w, h, r, wsz).astype(np.uint8)
    res = o.solve(sd, w, h, spp, depth, 1.0, threads=1)
    reduced = D.assemble_field(torch.from_numpy(res["field"].reshape(-1).copy()), wsz, r, w, h, mode="reduce")
    gathered = D.assemble_field(torch.from_numpy(res["field"].reshape(-1).copy()), wsz, r, w, h, mode="gather")
    steps = torch.tensor([res["walk_steps"]], dtype=torch.int64)
    dist.all_reduce(steps)
    # guided, frozen network
    prob = box_problem(d_sides=(0, 2), n_sides=(1, 3), value=lambda x, y: y, flux=lambda x, y, s: 0.0, n_per_side=8)
    gd = prob.as_dict()
    gd["mask"] = D.owned_mask(24, 16, r, wsz).astype(np.uint8)
    cfg = default_net_config()
    params = np.random.default_rng(5).uniform(-0.3, 0.3, o.net_n_params(cfg)).astype(np.float32)
    gs = guided_settings(24, 16, 2, 24, 1e-3, (-0.1, -0.1), (1.1, 1.1), train_spp_count=0)
    gres = o.solve_guided(gd, gs, cfg, params, threads=1)
    gfield = D.assemble_field(torch.from_numpy(gres["field"].reshape(-1).copy()), wsz, r, 24, 16)
    # the shared network's sync callback: integer sums do not depend on the reduction order
    grad = torch.arange(1000, dtype=torch.int64) * (r + 1) - (1 << 40) * (r % 3)
    sync = D.make_network_sync(grad)
    assert sync(D.SYNC_SUM_I64_DEVICE, None, 1000) == 0
    v = C.c_int64(100 + 7 * ((r * 5) % wsz))
    assert sync(D.SYNC_MIN_I64_HOST, C.cast(C.pointer(v), C.c_void_p), 1) == 0
    n = C.c_int64(0)
    assert sync(D.SYNC_RANKS_I64_HOST, C.cast(C.pointer(n), C.c_void_p), 1) == 0
    assert sync(99, None, 0) == D.SYNC_UNSUPPORTED == 2      # 'op unknown' is told apart from a failure (1)
    if r == 0:
        q.put((reduced.numpy().reshape(-1, 3), gathered.numpy().reshape(-1, 3), int(steps.item()), gfield.numpy().reshape(-1, 3),
               grad.numpy().copy(), v.value, n.value))
    dist.barrier()
    dist.destroy_process_group()


def test_eight_ranks_field_assembly_guided_and_gradient_sum(oracle, ladybug):
    import torch.multiprocessing as mp
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from conftest import box_problem
    from oracle.oracle import default_net_config, guided_settings
    w, h, spp, depth = 64, 40, 2, 24
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_eight_worker, args=(r, 8, port, w, h, spp, depth, q)) for r in range(8)]
    for p in procs:
        p.start()
    reduced, gathered, steps, gfield, grad, vmin, nranks = q.get(timeout=600)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    ref = oracle.solve(ladybug.as_dict(), w, h, spp, depth, 1.0)
    assert steps == ref["walk_steps"]
    assert np.array_equal(reduced, ref["field"]) and np.array_equal(gathered, ref["field"])
    prob = box_problem(d_sides=(0, 2), n_sides=(1, 3), value=lambda x, y: y, flux=lambda x, y, s: 0.0, n_per_side=8)
    cfg = default_net_config()
    params = np.random.default_rng(5).uniform(-0.3, 0.3, oracle.net_n_params(cfg)).astype(np.float32)
    gs = guided_settings(24, 16, 2, 24, 1e-3, (-0.1, -0.1), (1.1, 1.1), train_spp_count=0)
    gref = oracle.solve_guided(prob.as_dict(), gs, cfg, params, threads=4)
    assert np.array_equal(gfield, gref["field"])
    expect = sum(np.arange(1000, dtype=np.int64) * (r + 1) - (1 << 40) * (r % 3) for r in range(8))
    assert np.array_equal(grad, expect) and vmin == 100 and nranks == 8
