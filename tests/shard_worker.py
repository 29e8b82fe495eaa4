"""One rank of the N > 1 self-play path with a REAL engine (child process of tests/test_gpu_configs.py and of
tools/shard_rehearsal.py): games sharded by uid (shard.uid_plan), played to the end on this rank's engine, training tuples
pooled over all ranks by shard.TuplePool (all-gather).  The reference's counterpart is the per-task policy + mpsc collection
of AsyncAgent::run (agent/async_agent/mod.rs:100-137): results are a concatenation in any order.

    python tests/shard_worker.py RANK WORLD PORT N GAMES_PER_RANK S SEED OUT.npz [device] [backend]
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    rank, world, port, n, G, S, seed = [int(x) for x in sys.argv[1:8]]
    out = sys.argv[8]
    device = int(sys.argv[9]) if len(sys.argv) > 9 else 0
    backend = sys.argv[10] if len(sys.argv) > 10 else "gloo"
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
    eng = pkg.Engine(board_size=n, num_games=G, num_simulation=S, seed=seed, evaluator=pkg.EVAL_RANDOM, device=device,
                     game_uid_base=base, game_uid_stride=stride_uid)
    cap = G * n * n
    pool = shard.TuplePool(eng, cap, world, backend=backend)
    records = []
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
    np.savez(out, records=allrec, finished=st["games_finished"], overflow=st["overflow_games"],
             uids=np.array(sorted(g["uid"] for g in games), np.int64))
    eng.close()
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
