#!/bin/bash
# round-6 bench lines beside the default one (run on the GPU box through gpurun): tools/r06_benches.sh
o=gpurun_out/r06b; mkdir -p $o
T="timeout -k 10"
$T 300 python bench.py --workload ref15 --steps 10 --warmup 3 > $o/bench_ref15.json 2> $o/ref15.err
$T 300 python bench.py --workload config3 --steps 10 --warmup 2 --single-precision --no-cpu-baseline > $o/bench_config3.json 2> $o/config3.err
$T 300 python bench.py --workload config5 --steps 2048 --warmup 256 > $o/bench_config5.json 2> $o/config5.err
$T 300 bash tools/share_rehearsal.sh 2 4096 5 > $o/bench_share2.json 2> $o/share2.err
$T 300 bash tools/share_rehearsal.sh 4 512 3 > $o/bench_share4.json 2> $o/share4.err
$T 300 python tools/train_bench.py --json $o/train_bench.json > $o/train.out 2> $o/train.err
$T 300 python tools/config1.py --json $o/config1.json > $o/config1.out 2> $o/config1.err
bash tools/group_bench.sh r06 > /dev/null 2>&1
for f in $o/*.json; do echo "== $f"; head -c 300 $f; echo; done
