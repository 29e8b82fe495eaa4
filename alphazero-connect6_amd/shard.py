"""Multi-GPU sharding of self-play: one process per GPU, games sharded by index, no exchange while playing;
the only collective is an all-gather (RCCL over xGMI; gloo on CPU in the tests) that pools the fixed-size
training tuples -- the counterpart of the reference's mpsc collection, agent/async_agent/mod.rs:129-134."""
import numpy as np
import torch
import torch.distributed as dist


def shard_games(total_games, world_size, rank):
    """games per rank (weak scaling uses the same count on every rank) and this rank's first C6RNG game uid"""
    per = total_games // world_size
    rem = total_games % world_size
    mine = per + (1 if rank < rem else 0)
    first = rank * per + min(rank, rem)
    return mine, first


def uid_plan(games_per_rank, world_size, rank):
    """C6RNG game uids: slot i of rank r starts at r*G + i and a recycled slot advances by world*G, so that the
    union over ranks is the same set of games whatever the number of GPUs"""
    return rank * games_per_rank, world_size * games_per_rank


def gather_tuples(buf, count, world_size, group=None):
    """buf: uint8 tensor [cap, stride] (device or CPU), count: int32 tensor [1].
    Returns (pooled [world, cap, stride], counts [world]) on every rank."""
    if world_size == 1:
        return buf.unsqueeze(0), count.reshape(1)
    pooled = torch.empty((world_size,) + tuple(buf.shape), dtype=buf.dtype, device=buf.device)
    counts = torch.empty(world_size, dtype=count.dtype, device=count.device)
    dist.all_gather_into_tensor(pooled.view(-1), buf.reshape(-1).contiguous(), group=group)
    dist.all_gather_into_tensor(counts, count.reshape(1), group=group)
    return pooled, counts


def compact(pooled, counts):
    """concatenate the valid records of every rank (host side)"""
    pooled = pooled.cpu().numpy()
    counts = counts.cpu().numpy()
    cap = pooled.shape[1]
    return np.concatenate([pooled[r, : min(int(counts[r]), cap)] for r in range(pooled.shape[0])], axis=0)
