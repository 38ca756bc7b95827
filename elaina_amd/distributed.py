"""Multi-GPU glue for the sharded solve: one process per GPU, torch.distributed
(backend "nccl" = RCCL over xGMI on ROCm; "gloo" in CPU tests).

The walk itself needs no collective: pixels are independent, every rank
holds the whole scene and owns the 64-pixel tiles t with t % world == rank
(wost_solve_sharded).  The only exchange is one sum-reduce of the zero-padded fields.
"""
import os

import numpy as np


def env_rank_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(
        os.environ.get("LOCAL_RANK", "0"))


def tile_of_pixels(width, height):
    """tile index (8x8 pixel tiles, row-major over tiles) of every pixel id; mirrors
    init_kernel in csrc/wost_hip.hip"""
    tiles_x = (width + 7) // 8
    y, x = np.divmod(np.arange(width * height), width)
    return (y // 8) * tiles_x + (x // 8)


def owned_mask(width, height, shard_index, shard_count):
    """boolean mask over pixel ids of the pixels a shard owns"""
    return (tile_of_pixels(width, height) % shard_count) == shard_index


def init_process_group(backend=None):
    import torch
    import torch.distributed as dist
    rank, world, local = env_rank_world()
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local % max(torch.cuda.device_count(), 1))
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def reduce_field(field, world):
    """sum the zero-padded per-rank fields in place (all ranks get the full field)"""
    if world > 1:
        import torch.distributed as dist
        dist.all_reduce(field, op=dist.ReduceOp.SUM)
    return field


_GATHER_PLANS = {}


def _gather_plan(width, height, world, device):
    """per rank the pixel ids it owns (tile-interleaved, as wost_solve_sharded deals them), padded to one length"""
    import torch
    key = (width, height, world, str(device))
    plan = _GATHER_PLANS.get(key)
    if plan is None:
        tiles = torch.from_numpy(tile_of_pixels(width, height) % world)
        ids = [torch.nonzero(tiles == r, as_tuple=False).flatten() for r in range(world)]
        longest = max(len(i) for i in ids)
        plan = (longest, [i.to(device) for i in ids])
        _GATHER_PLANS[key] = plan
    return plan


def gather_field(field, world, rank, width, height):
    """Assemble the full field from the ranks' DISJOINT shards with one all-gather (1/world of the all-reduce's
    bytes per rank, no additions): every rank packs the pixels it owns, the packed shards are exchanged, and each
    lands at its pixels.  Same result as reduce_field -- a shard's own pixels never receive another rank's
    contribution -- so callers may pick by message size (assemble_field)."""
    if world <= 1:
        return field
    import torch
    import torch.distributed as dist
    f = field.view(-1, 3)
    longest, ids = _gather_plan(width, height, world, field.device)
    mine = torch.zeros(longest, 3, dtype=field.dtype, device=field.device)
    mine[: len(ids[rank])] = f.index_select(0, ids[rank])
    parts = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(parts, mine)
    for r in range(world):
        if r != rank:
            f.index_copy_(0, ids[r], parts[r][: len(ids[r])])
    return field


# above this size the assembled field travels as an all-gather of the disjoint shards; below it the single
# all-reduce of zero-padded frames (north_star's wording) has fewer steps and the bytes do not matter
GATHER_THRESHOLD_BYTES = 32 << 20


def assemble_field(field, world, rank, width, height, mode="auto"):
    """the one exchange of a sharded solve; mode: "auto" (by message size), "reduce" or "gather" """
    if world <= 1:
        return field
    if mode == "gather" or (mode == "auto" and field.numel() * field.element_size() >= GATHER_THRESHOLD_BYTES):
        return gather_field(field, world, rank, width, height)
    return reduce_field(field, world)


SYNC_SUM_I64_DEVICE, SYNC_MIN_I64_HOST, SYNC_RANKS_I64_HOST = 0, 1, 2      # wost_sync_fn ops (include/wost.h)
SYNC_UNSUPPORTED = 2       # what a callback returns for an op it does not know (0 = done, anything else = failed)


def make_network_sync(grad, synchronize=None):
    """The body of the wost_sync_fn callback of a shared guiding network (wost_guided_set_sync): `grad` is the
    int64 fixed-point gradient tensor the library accumulates into (wost_net_set_gradient_buffer); before every
    Adam step it is summed over the ranks -- integers, so every rank ends with the same bits whatever the order --
    once per training pass the ranks agree on the number of full batches (MIN), and once per solve the library
    asks for the number of ranks (it divides the summed gradient by it).  Returns f(op, data_ptr, count) -> 0 on
    success, SYNC_UNSUPPORTED (2) for an op it does not know (never a silent MIN), 1 for a failure -- the library
    ends the solve on a failure and only tolerates "unsupported" for the rank-count op."""
    import ctypes as C
    import torch
    import torch.distributed as dist

    def sync(op, data, count):
        try:
            if op == SYNC_SUM_I64_DEVICE:
                dist.all_reduce(grad, op=dist.ReduceOp.SUM)
                if synchronize:
                    synchronize()
            elif op == SYNC_RANKS_I64_HOST:
                C.cast(data, C.POINTER(C.c_int64))[0] = dist.get_world_size()
            elif op == SYNC_MIN_I64_HOST:
                v = C.cast(data, C.POINTER(C.c_int64))
                t = torch.tensor([v[0]], dtype=torch.int64, device=grad.device)
                dist.all_reduce(t, op=dist.ReduceOp.MIN)
                v[0] = int(t.item())
            else:
                print("network sync: unknown op %d" % op)
                return SYNC_UNSUPPORTED
            return 0
        except Exception as e:     # never let an exception cross the C boundary
            print("network sync failed: %r" % (e,))
            return 1

    return sync
