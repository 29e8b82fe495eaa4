"""Pools independent runs of tools/stones_probe.py (same workload, budget, precision and random-stream specification) into the record bench.py reads:
    python tools/pool_probes.py OUT.json RUN1.json RUN2.json ...
mean and standard deviation over all games (from the runs' own means / deviations), standard error, the runs kept verbatim."""
import json
import math
import sys

out, files = sys.argv[1], sys.argv[2:]
runs = [json.loads(open(f).read().strip().splitlines()[-1]) for f in files]
for k in ("workload", "num_simulation", "precision", "rng", "board", "net_blocks", "net_channels"):      # (not src_hash: runs under one random-stream specification pool, bench.committed_stones_probe)
    assert len({json.dumps(r.get(k)) for r in runs}) == 1, f"runs differ in {k}"
n = sum(r["games"] for r in runs)
mean = sum(r["stones_per_game"] * r["games"] for r in runs) / n
var = sum(r["games"] * (r["stones_std"] ** 2 + (r["stones_per_game"] - mean) ** 2) for r in runs) / n
rec = dict(games=n, num_simulation=runs[0]["num_simulation"], stones_per_game=mean, stones_std=math.sqrt(var), stones_min=min(r["stones_min"] for r in runs),
           stones_max=max(r["stones_max"] for r in runs), black_wins=sum(r["black_wins"] for r in runs), white_wins=sum(r["white_wins"] for r in runs),
           draws=sum(r["draws"] for r in runs), standard_error=math.sqrt(var / n), workload=runs[0]["workload"], precision=runs[0]["precision"],
           board=runs[0]["board"], net_blocks=runs[0]["net_blocks"], net_channels=runs[0]["net_channels"], src_hash=runs[-1]["src_hash"], src_hashes=sorted({r["src_hash"] for r in runs}), rng=runs[0].get("rng", "C6RNG v4"),
           note=f"{len(runs)} independent runs of tools/stones_probe.py (seeds {', '.join(str(r.get('seed')) for r in runs)}): complete games of the benchmark's "
                "random-init net at the benchmark's simulation budget, pooled by tools/pool_probes.py", runs=runs)
json.dump(rec, open(out, "w"))
print(f"{n} games: {mean:.2f} stones per game, standard error {rec['standard_error']:.2f}")
