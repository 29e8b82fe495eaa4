"""tree + board kernels only (uniform evaluator) at a workload's geometry: per-phase ms per round in the opening and after `settle` rounds
    python tools/tree_phase.py [n=19] [games=4096] [sims=400] [settle=42000] [rounds=200]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from c6amd_loader import load_package
pkg = load_package()
n, G, S, settle, K = [int(a) for a in (sys.argv[1:6] + ["19", "4096", "400", "42000", "200"][len(sys.argv) - 1:])]
eng = pkg.Engine(board_size=n, num_games=G, num_simulation=S, seed=2024, evaluator=pkg.EVAL_UNIFORM, recycle=True, stagger=True)
def phase(tag):
    eng.sync(); eng.profile(True); eng.profile_read(reset=True)
    t0 = time.perf_counter(); eng.step(K); eng.sync(); dt = time.perf_counter() - t0
    p = eng.profile_read(reset=True); eng.profile(False); st = eng.stats()
    print(json.dumps({"tag": tag, "ms_per_round": 1000 * dt / K, "select_ms": p["select_ms"] / K, "expand_ms": p["expand_ms"] / K, "eval_ms": (p["eval_other_ms"] + p["conv_ms"]) / K,
                      "move_ms": p["move_ms"] / K, "max_records_used": st["max_records_used"], "max_depth": st["max_depth"], "select_per_sim": st["select_steps"] / st["simulations"],
                      "finished": st["games_finished"], "stones": st["stones"], "abandoned": st["abandoned_games"]}), flush=True)
phase("opening")
left = settle
while left > 0:
    c = min(left, 4096); eng.step(c); left -= c; eng.maintain()
phase("steady")
eng.close()
