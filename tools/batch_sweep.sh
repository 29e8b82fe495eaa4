#!/bin/bash
# ms per simulation round of config 2's net at a range of batch sizes (run on the GPU box): tools/batch_sweep.sh [GAMES ...]
#   C6_W6_SPLIT=0 / 1 forces the GEMM's block / split form (default: the launcher's choice, split below 8 blocks per CU); --workload config3 etc. via BENCH_ARGS
cd "$(dirname "$0")/.."
games=${@:-1 8 64 512}
for g in $games; do
  python3 bench.py $BENCH_ARGS --games $g --single-precision --no-cpu-baseline --no-probe --steps ${STEPS:-100} --warmup ${WARMUP:-20} 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$g games: ms_per_step', round(d['ms_per_step'], 3), 'phase', {k: round(v / d['steps'], 3) for k, v in d.get('phase_ms', {}).items()}, 'gemm avg', d['roofline'].get('avg_launch_ms'))
"
done
