#!/bin/bash
# run on the GPU box: the ref15 workload (4096 games, 15x15, S = 800, stem + 9 blocks) on the two tilings of the 15x15 board, alternating, same box
for rep in 1 2; do
  for hook in 0 1; do
    C6_W6_N15=$hook timeout -k 10 200 python bench.py --workload ref15 --steps 10 --warmup 3 --no-probe --no-cpu-baseline 2>/dev/null | python -c "
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']
print('C6_W6_N15=$hook (%s): %.1f ms per round, %.0f simulations/s, GEMM %.2f ms per layer = %.1f TFLOP/s executed (frac %.3f), boundary %.2f ms per layer, layer_frac %.3f' % (
    'conv_wino6 tiling, 488 positions' if $hook else 'F(4x4) tiling, 534 positions', d['ms_per_step'], d['sims_per_s'], r['avg_launch_ms'], r['achieved'], r['frac'],
    (d['phase_ms']['conv_pre_ms'] + d['phase_ms']['conv_post_ms']) / r['launches'], r['layer_frac']))"
  done
done
