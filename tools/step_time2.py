import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as ge
from c6amd_loader import load_package
pkg = load_package()
from alphazero_connect6_amd import shard
mode = sys.argv[1] if len(sys.argv) > 1 else "pool"
torch.cuda.set_device(0)
eng = pkg.Engine(board_size=19, num_games=4096, num_simulation=400, seed=2024, evaluator=pkg.EVAL_NET, net_channels=256, net_blocks=4, recycle=True, stagger=True, net_precision=1)
eng.load_weights(ge.make_random_blob(19, 256, 4, np.random.default_rng(0)))
pool = shard.TuplePool(eng, 8192, 1, backend="nccl", keep=False) if mode != "nopool" else None
eng.step(2); eng.sync()
if mode == "stats":
    print(eng.stats())
if mode != "noprof":
    eng.profile(True); eng.profile_read(reset=True)
torch.cuda.synchronize()
for i in range(4):
    t0 = time.perf_counter()
    eng.step(1)
    t1 = time.perf_counter()
    if pool and i == 3: pool.exchange()
    eng.sync()
    t2 = time.perf_counter()
    print(f"{mode} round {i}: enqueue {1000 * (t1 - t0):.1f} ms, total {1000 * (t2 - t0):.1f} ms", flush=True)
eng.close()
