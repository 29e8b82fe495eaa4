"""Measured stones per game for bench.py: complete self-play games of the benchmark's net (random-init weights, Dirichlet noise) at a
short simulation budget, on the GPU engine.  Writes one JSON line (commit it as profiles/rNN_stones_probe.json).
    python tools/stones_probe.py [workload=config2] [games=4096] [sims=8] [also_sims=0] [seed=7] [precision=fp32]
The record carries the random-stream specification it was played under (c6_rng_spec) and the hash of the kernel sources: bench.py uses a probe
only at the workload's own budget and under the library's current specification (bench.committed_stones_probe)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from c6amd_loader import load_package

pkg = load_package()
name = sys.argv[1] if len(sys.argv) > 1 else "config2"
games = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
sims = int(sys.argv[3]) if len(sys.argv) > 3 else 8
also = int(sys.argv[4]) if len(sys.argv) > 4 else 0
seed = int(sys.argv[5]) if len(sys.argv) > 5 else 7
pname = sys.argv[6] if len(sys.argv) > 6 else "fp32"
prec = {"fp32": pkg.PREC_FP32, "bf16x3": pkg.PREC_BF16X3}[pname]
wl = bench.WORKLOADS[name]
r = bench.stones_probe(pkg, wl, games, sims, prec, seed=seed, progress=True, partial_path=os.environ.get("C6_PROBE_PARTIAL"))
r["seed"] = seed
r.update(workload=name, precision=pname, board=wl["n"], net_blocks=wl["blocks"], net_channels=wl["channels"], src_hash=bench.kernel_source_hash(), rng=pkg.rng_spec())
if also:      # how the game length moves with the simulation budget: further budgets on fewer games
    r["other_budgets"] = [bench.stones_probe(pkg, wl, 256, b, prec, seed=8 + b, progress=True) for b in (2, also)]
print(json.dumps(r))
