bash tools/profile_round.sh r04 2>&1 | tail -12
python bench.py > gpurun_out/r04_bench_n1.json 2> gpurun_out/r04_bench_n1.err; tail -c 300 gpurun_out/r04_bench_n1.json
