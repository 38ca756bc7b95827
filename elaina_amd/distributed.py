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
