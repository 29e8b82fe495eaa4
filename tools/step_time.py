"""wall time of single engine rounds (config 2 geometry) per precision: python tools/step_time.py [precision=1] [profile=0] [games=4096]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
from c6amd_loader import load_package
pkg = load_package()
prec = int(sys.argv[1]) if len(sys.argv) > 1 else 1
prof = int(sys.argv[2]) if len(sys.argv) > 2 else 0
G = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
t0 = time.perf_counter()
eng = pkg.Engine(board_size=19, num_games=G, num_simulation=400, seed=1, evaluator=pkg.EVAL_NET, net_channels=256, net_blocks=4, recycle=True, stagger=True, net_precision=prec)
eng.load_weights(ge.make_random_blob(19, 256, 4, np.random.default_rng(0)))
eng.sync()
print(f"create+load {time.perf_counter() - t0:.2f} s", flush=True)
if prof:
    eng.profile(True)
for i in range(5):
    t0 = time.perf_counter()
    eng.step(1)
    t1 = time.perf_counter()
    eng.sync()
    t2 = time.perf_counter()
    print(f"round {i}: enqueue {1000 * (t1 - t0):.1f} ms, total {1000 * (t2 - t0):.1f} ms", flush=True)
if prof:
    print(eng.profile_read())
eng.close()
