"""Host-callback (PCIe-inclusive) rate of the drop-in boundary: cpp_self_play with G concurrent 15x15 games and a trivial
numpy evaluator callback.  Every simulation round ships 8*G boards (f32) to the host and the callback's values/policies
back -- the rate the reference's own boundary allows, as opposed to the resident-net engine measured by bench.py.

    python3 tools/dropin_rate.py [--games 256] [--sims 50]"""
import argparse, json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from c6amd_loader import load_package
pkg = load_package()
ap = argparse.ArgumentParser(); ap.add_argument("--games", type=int, default=256); ap.add_argument("--sims", type=int, default=50)
a = ap.parse_args()
calls = [0, 0]
def cb(turn, boards):
    calls[0] += 1; calls[1] += boards.shape[0]
    return np.zeros(boards.shape[0], np.float32), np.full((boards.shape[0], 225), 1.0 / 225, np.float32)
param = pkg.default_param(); param.update(num_simulation=a.sims, num_game_thread=a.games)
t0 = time.time(); res = pkg.self_play_np(cb, param); dt = time.time() - t0
stones = sum(len(p) for _, p in res)
sims = stones * a.sims
print(json.dumps({"games": a.games, "sims_per_move": a.sims, "stones": stones, "wall_s": round(dt, 2), "sims_per_s": round(sims / dt), "boards_per_s": round(calls[1] / dt),
                  "callback_calls": calls[0], "bytes_over_pcie_per_s": round(calls[1] * 225 * 4 * 2 / dt)}))
