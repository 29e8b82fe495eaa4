"""Where a wave of c6_search_kernel spends its cycles (diagnostic build only):
    tools/ab_build.sh stamps tree_kernels.hip -DC6_AB_STAMPS && C6_LIB_PATH=$PWD/tools/ab/libstamps.so python3 tools/tree_stamps.py [n=9] [G=32768] [S=100] [settle=6800] [rounds=1024]
Per phase: cycles per simulation and wave (s_memtime, waits included; lane 0 of every wave accumulates), share of the wave's life."""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from c6amd_loader import load_package

pkg = load_package()
a = [int(x) for x in sys.argv[1:]]
n, G, S, settle, rounds = (a + [9, 32768, 100, 6800, 1024][len(a):])[:5]
eng = pkg.Engine(board_size=n, num_games=G, num_simulation=S, seed=2024, evaluator=pkg.EVAL_UNIFORM, recycle=True, stagger=True)
left = settle
while left > 0:
    c = min(left, 4096)
    eng.step(c); left -= c; eng.maintain()
eng.sync()
buf = (C.c_ulonglong * 16)()
L = pkg.lib()
assert L.c6_debug_stamps(buf, 1) == 0
s0 = eng.stats()
t0 = time.perf_counter()
for _ in range(rounds // 64):
    eng.step(64)
eng.sync()
dt = time.perf_counter() - t0
s1 = eng.stats()
assert L.c6_debug_stamps(buf, 0) == 0
sims = s1["simulations"] - s0["simulations"]
names = ["head (round head, game registers)", "scan (children: rootc row / tags + table, priors)", "noise (Gamma sampler + Dirichlet)", "score (PUCT, argmax, ties)",
         "step (into the child) + loop", "eval (uniform evaluator) + leaf hand-off", "expand + backup", "move / round tail", "tail", "outside"]
vals = [C.c_longlong(buf[i]).value for i in range(10)]
tot = sum(vals[:9])
print(f"{n}x{n}, {G} games, S = {S}: {1000 * dt / (rounds // 64 * 64):.4f} ms per round (stamped build), {sims} simulations, select depth {(s1['select_steps'] - s0['select_steps']) / sims:.3f}")
for nm, v in zip(names[:9], vals[:9]):
    print(f"  {nm:55s} {v / sims:9.1f} cycles / simulation  {100.0 * v / tot:5.1f} %")
print(f"  {'sum':55s} {tot / sims:9.1f}")
eng.close()
