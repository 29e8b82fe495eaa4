"""BASELINE config 1: 1 self-play game, num_simulation = 400, RandomEvaluator, seed 0 -- winner, path length
and wall time on (a) the CPU oracle (1 thread, like the Rust sample), (b) the GPU engine's built-in random evaluator,
(c) the drop-in cpp_self_play with a host callback of the reference sample's shape (Connect6/sample/cpp/main.cpp:6-28,
N = 15 only: the struct layout is 15x15).  One game cannot fill a GPU: this is the plumbing case, not a throughput number.

    python3 tools/config1.py [--json out.json]
"""
import argparse, json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
from c6amd_loader import load_package
pkg = load_package()
import c6oracle as o          # the CPU oracle: the baseline being timed here

ap = argparse.ArgumentParser(); ap.add_argument("--json", default=""); a = ap.parse_args()
out = {"config": "BASELINE config 1: 1 game, num_simulation 400, RandomEvaluator, seed 0", "cases": []}
for n in (19, 15):
    t0 = time.time(); ref = o.self_play(n, num=1, threads=1, num_simulation=400, seed=0)[0]; t_cpu = time.time() - t0
    t0 = time.time()
    e = pkg.Engine(board_size=n, num_games=1, num_simulation=400, seed=0, evaluator=pkg.EVAL_RANDOM); e.run()
    g = e.fetch_games()[0]; st = e.stats(); e.close(); t_gpu = time.time() - t0
    same = list(g["moves"]) == [r * n + c for _, _, (r, c) in ref["path"]] and g["winner"] == ref["winner"]
    out["cases"].append({"board": n, "winner": int(ref["winner"]), "stones": len(ref["path"]), "oracle_cpu_1thread_s": round(t_cpu, 2),
                         "gpu_engine_s": round(t_gpu, 2), "bit_identical": bool(same), "simulations": int(st["simulations"])})
rng = np.random.default_rng(0)
def sample_callback(turn, boards):     # random values / policies like the reference's sample policy
    k = boards.shape[0]
    return rng.uniform(-1, 1, k).astype(np.float32), rng.uniform(0, 1, (k, 225)).astype(np.float32)
param = pkg.default_param(); param.update(num_simulation=400, num_game_thread=1)
t0 = time.time(); res = pkg.self_play_np(sample_callback, param); t_drop = time.time() - t0
winner, path = res[0] if isinstance(res, (list, tuple)) and not isinstance(res[0], (int, np.integer)) else res
out["cases"].append({"board": 15, "entry": "cpp_self_play (drop-in ABI, host callback)", "winner": int(winner), "stones": len(path), "wall_s": round(t_drop, 2)})
print(json.dumps(out))
if a.json: json.dump(out, open(a.json, "w"), indent=1)
