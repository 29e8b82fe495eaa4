"""One rank of the N > 1 self-play path with a REAL engine (child process of tests/test_gpu_configs.py and of
tools/shard_rehearsal.py): games sharded by uid (shard.uid_plan), played to the end on this rank's engine, training tuples
pooled over all ranks by shard.TuplePool (all-gather).  The reference's counterpart is the per-task policy + mpsc collection
of AsyncAgent::run (agent/async_agent/mod.rs:100-137): results are a concatenation in any order.

    python tests/shard_worker.py RANK WORLD PORT N GAMES_PER_RANK S SEED OUT.npz [device] [backend] [mode] [evaluator]

mode "play" (default): every game to its end.  mode "props": config 4's shape -- many games per rank for a few rounds, root
invariants on samples, then ONE game per rank finished from outside (c6_engine_apply_moves, the scripted 11-stone game of
agent_impl/tests.rs:65-130) so that the tuple exchange carries a payload.  evaluator "random" (default: the on-device RandomEvaluator) or "net" (the resident
policy/value net at config 2 / 4's shape: C = 256, stem + 3 residual blocks, blob seed NET_BLOB_SEED); with the net the props mode also
stores the roots of a few sampled games, which the parent test compares with the oracle driven by its own FP32 net.  WORLD = 1 with
backend "nccl" builds a ONE-rank RCCL group and makes TuplePool issue its all-gathers through it (collective=True).
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
NET_C, NET_BLOCKS, NET_BLOB_SEED = 256, 4, 12


def main():
    rank, world, port, n, G, S, seed = [int(x) for x in sys.argv[1:8]]
    out = sys.argv[8]
    device = int(sys.argv[9]) if len(sys.argv) > 9 else 0
    backend = sys.argv[10] if len(sys.argv) > 10 else "gloo"
    mode = sys.argv[11] if len(sys.argv) > 11 else "play"
    evaluator = sys.argv[12] if len(sys.argv) > 12 else "random"
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch
    import torch.distributed as dist

    from c6amd_loader import load_package
    pkg = load_package()
    from alphazero_connect6_amd import shard

    torch.cuda.set_device(device)
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", device))
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    base, stride_uid = shard.uid_plan(G, world, rank)
    if evaluator == "net":
        import __graft_entry__ as ge
        eng = pkg.Engine(board_size=n, num_games=G, num_simulation=S, seed=seed, evaluator=pkg.EVAL_NET, net_channels=NET_C, net_blocks=NET_BLOCKS,
                         device=device, game_uid_base=base, game_uid_stride=stride_uid)
        eng.load_weights(ge.make_random_blob(n, NET_C, NET_BLOCKS, np.random.default_rng(NET_BLOB_SEED)))
    else:
        eng = pkg.Engine(board_size=n, num_games=G, num_simulation=S, seed=seed, evaluator=pkg.EVAL_RANDOM, device=device,
                         game_uid_base=base, game_uid_stride=stride_uid)
    cap = G * n * n if mode == "play" else 4096
    pool = shard.TuplePool(eng, cap, world, backend=backend, collective=True if backend == "nccl" else None)
    records = []
    if mode == "props":
        return props(eng, pool, rank, world, n, G, out, dist, base, evaluator)
    # a few rounds at a time, pooling in between: the collective runs while the next rounds are already enqueued
    for _ in range(100000):
        eng.step(S)
        pool.exchange()
        got = pool.collect()
        if got is not None and got.shape[0]:
            records.append(got)
        done = torch.tensor([int(eng.stats()["live_games"] == 0)], dtype=torch.int32)
        if backend == "nccl":
            done = done.cuda()
        dist.all_reduce(done, op=dist.ReduceOp.MIN)
        if int(done.item()) == 1:
            break
    got = pool.flush()
    if got is not None and got.shape[0]:
        records.append(got)
    st = eng.stats()
    games = eng.fetch_games()
    allrec = np.concatenate(records, axis=0) if records else np.zeros((0, pool.stride), np.uint8)
    np.savez(out, records=allrec, finished=st["games_finished"], overflow=st["overflow_games"], dropped=pool.dropped,
             uids=np.array(sorted(g["uid"] for g in games), np.int64))
    eng.close()
    dist.barrier()
    dist.destroy_process_group()


PROPS_ROUNDS = 6


def props_samples(G):
    """slots whose roots travel back to the parent: the first and last leaves of the net's passes (8192 games = 2 passes of 32 768 images)
    and one inside each"""
    return sorted({0, G // 2 - 1, G // 2, G - 1} if G >= 4 else set(range(G)))


def props(eng, pool, rank, world, n, G, out, dist, uid_base, evaluator):
    R = PROPS_ROUNDS
    eng.step(R)
    pool.exchange()
    st = eng.stats()
    ok = st["simulations"] == G * R and st["overflow_games"] == 0 and st["live_games"] == G
    for g in list(range(0, G, max(1, G // 16))) + [G - 1]:
        r = eng.root(g)
        ok = ok and r["root_visit"] == R and int(r["visits"].sum()) == R - 1 and r["stones"] == 0
    extra = {}
    if evaluator == "net":
        sm = props_samples(G)
        roots = [eng.root(g) for g in sm]
        extra = dict(sample_slots=np.array(sm), sample_uids=np.array([uid_base + g for g in sm]),
                     sample_visits=np.stack([r["visits"] for r in roots]), sample_q=np.stack([r["q"] for r in roots]),
                     sample_prior=np.stack([r["prior"] for r in roots]), sample_value=np.array([r["root_value"] for r in roots], np.float32))
    gathers_before = pool.payload_gathers
    # game `rank` of this rank's engine: Black (0,0) | White (5,0),(5,1) | Black (0,1),(0,2) | White (5,2),(5,3) | Black (0,3),(0,4) |
    # White (5,4),(5,5): White's six -- 11 stones, 11 tuples
    script = [(0, 0), (5, 0), (5, 1), (0, 1), (0, 2), (5, 2), (5, 3), (0, 3), (0, 4), (5, 4), (5, 5)]
    for (r_, c_) in script:
        pos = np.full(G, -2, np.int32)
        pos[rank] = r_ * n + c_
        eng.apply_moves(pos)
    got = pool.flush()
    games = eng.fetch_games()
    np.savez(out, ok=int(ok), gathers=pool.payload_gathers - gathers_before, pooled=0 if got is None else got.shape[0],
             total=pool.pooled_total(), dropped=pool.dropped, finished=len(games), winner=games[0]["winner"] if games else 0,
             records=got if got is not None else np.zeros((0, pool.stride), np.uint8), collective=int(pool.collective),
             exchanges=pool.exchanges, **extra)
    eng.close()
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
