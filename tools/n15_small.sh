for g in 1 16 64 512; do
  for hook in 1 0; do
    ms=$(C6_W6_N15=$hook timeout -k 10 120 python bench.py --workload ref15 --games $g --single-precision --no-cpu-baseline --no-probe --steps 100 --warmup 20 2>/dev/null | python -c "import json,sys; print(round(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'],3))")
    echo "15x15 stem+9 blocks, $g games, C6_W6_N15=$hook: $ms ms per round"
  done
done
