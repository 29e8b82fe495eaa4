python -m pytest tests -m gpu -q -s > gpurun_out/r04_gpu_all.log 2>&1; tail -3 gpurun_out/r04_gpu_all.log
grep "net-vs-oracle" gpurun_out/r04_gpu_all.log > gpurun_out/r04_hostile_errors.txt; wc -l gpurun_out/r04_hostile_errors.txt
bash tools/pmc_tree.sh r04 > gpurun_out/r04_pmc_tree.txt 2>&1; bash tools/pmc_tree_traffic.sh r04 > gpurun_out/r04_pmc_tree_traffic.txt 2>&1; grep search_kernel gpurun_out/r04_pmc_tree.txt gpurun_out/r04_pmc_tree_traffic.txt
rm -rf gpurun_out/pmc_tree_r04 gpurun_out/pmc_tree_traffic_r04
