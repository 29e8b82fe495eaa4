"""BASELINE.md 3's CPU baseline for the net workloads as planned there: 2 x cores games, >= 3 FULL moves each (400 searches of 8 FP32 net
evaluations per move), one game per task on a pool of `cores` workers (AsyncAgent::run, agent/async_agent/mod.rs:100-137), the CPU oracle
with its FP32 net.  Minutes of CPU time: not part of the default bench.py run; commit the JSON line as profiles/rNN_cpu_baseline_full.json.
    python tools/cpu_baseline_full.py [workload=config2] [moves=3] [stones_per_game=measured value]"""
import json
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as ge
import bench
import c6oracle as o

name = sys.argv[1] if len(sys.argv) > 1 else "config2"
moves = int(sys.argv[2]) if len(sys.argv) > 2 else 3
wl = bench.WORKLOADS[name]
try:      # (only for the library's random-stream specification: which probe record counts)
    from c6amd_loader import load_package
    rng_spec = load_package().rng_spec()
except Exception:      # a CPU-only box without the GPU library: the oracle's own statement of the specification it implements
    rng_spec = o.rng_spec() if hasattr(o, "rng_spec") else "C6RNG v5"
probe = bench.committed_stones_probe(name if name != "config4" else "config2", bench.WORKLOADS[name]['sims'], rng_spec)
stones = float(sys.argv[3]) if len(sys.argv) > 3 else (probe["stones_per_game"] if probe else wl["stones_prior"])
cores = bench.usable_cores()
n, S = wl["n"], wl["sims"]
net = o.Net(n, wl["channels"], wl["blocks"], ge.make_random_blob(n, wl["channels"], wl["blocks"], np.random.default_rng(0)))
fn, ctx = net.eval_fn()
ngames = 2 * cores
todo = list(range(ngames))
lock = threading.Lock()
played = [0] * ngames


def worker():
    while True:
        with lock:
            if not todo:
                return
            g = todo.pop(0)
        az = o.AlphaZero(n, num_simulation=S, eval_kind=o.EVAL_CUSTOM, eval_fn=fn, ctx=ctx, seed=0, game_uid=g)
        game = o.new_game(n)
        for _ in range(moves):
            pos = az.next(game)
            if pos is None or (isinstance(pos, int) and pos < 0):
                break
            r, c = (pos // n, pos % n) if isinstance(pos, int) else pos
            o.game_set(game, r, c)
            played[g] += 1
        az.close()


t0 = time.time()
th = [threading.Thread(target=worker) for _ in range(cores)]
for t in th:
    t.start()
while any(t.is_alive() for t in th):      # a heartbeat: a job that prints nothing for minutes is taken to be hung
    time.sleep(30)
    print(f"[cpu_baseline_full] {time.time() - t0:.0f} s: {sum(played)} of {ngames * moves} moves played", file=sys.stderr, flush=True)
for t in th:
    t.join()
dt = time.time() - t0
net.close()
sims = sum(played) * S
print(json.dumps({"workload": name, "cores": cores, "games": ngames, "full_moves_per_game": moves, "stones_played": sum(played), "seconds": dt,
                  "sims_per_s": sims / dt, "stones_per_s": sum(played) / dt, "stones_per_game_used": stones,
                  "value": sims / dt / (S * stones), "unit": "games/s", "kind": "port",
                  "sample": f"{ngames} games x {moves} full moves of {S} searches (8 FP32 net evaluations each) on {cores} cores, one game per task"}))
