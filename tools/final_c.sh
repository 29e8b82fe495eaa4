python bench.py --workload config3 --steps 10 --warmup 2 --no-cpu-baseline --no-probe --single-precision > gpurun_out/r04_bench_config3.json 2> gpurun_out/r04_bench_config3.err; tail -c 300 gpurun_out/r04_bench_config3.json
bash tools/share_rehearsal.sh 2 4096 5 > gpurun_out/r04_bench_share2.json 2> gpurun_out/r04_bench_share2.err; tail -c 200 gpurun_out/r04_bench_share2.json
bash tools/ab_many.sh config2 "tree r03" --steps 20 --warmup 5 > gpurun_out/r04_ab_r03_config2.txt 2>&1; cat gpurun_out/r04_ab_r03_config2.txt
bash tools/ab_many.sh config5 "tree r03" --steps 2048 --warmup 256 > gpurun_out/r04_ab_r03_config5.txt 2>&1; cat gpurun_out/r04_ab_r03_config5.txt
