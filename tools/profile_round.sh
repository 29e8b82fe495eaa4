#!/bin/bash
# tools/profile_round.sh TAG [bench.py arguments]  -- run ON THE GPU BOX (through gpurun): the rocprofv3 passes behind profiles/<TAG>_*.
#   1. --kernel-trace --stats              -> per-kernel durations
#   2. --pmc FETCH_SIZE                    -> HBM read bytes   (TCC: 3 slots)
#   3. --pmc WRITE_SIZE                    -> HBM write bytes  (TCC: 2 slots; cannot share a pass with FETCH_SIZE)
#   4. --pmc SQ_VALU_MFMA_BUSY_CYCLES ...  -> matrix-core busy cycles, effective clock, LDS conflicts
# Counter passes carry no trace options (gpurun refuses the combination).  Afterwards, in the build container:
#   python profiles/summarize.py gpurun_out/prof_<TAG> <TAG>
tag=${1:-r02}
shift      # further arguments go to bench.py (e.g. --workload ref15)
out=gpurun_out/prof_$tag
mkdir -p $out
export TMPDIR=/tmp
# C6_NO_VMM=1: under rocprofv3 the record pools are plain allocations (a --kernel-trace run with hipMemMap'ed pools did not come back on this
# pool's runtime); nothing the profiled kernels do depends on how the pools were allocated
export C6_NO_VMM=1
B="python3 bench.py --no-cpu-baseline --no-probe $*"
T="timeout -k 10 500"
$T rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt -- $B --steps 2 --warmup 1 > $out/bench_kt.json 2> $out/kt.err
$T rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -- $B --steps 1 --warmup 1 > $out/bench_pmc1.json 2> $out/pmc1.err
$T rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -- $B --steps 1 --warmup 1 > $out/bench_pmc2.json 2> $out/pmc2.err
$T rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --output-format csv -d $out/pmc_sq -- $B --steps 1 --warmup 1 > $out/bench_pmc3.json 2> $out/pmc3.err
for f in kt pmc1 pmc2 pmc3; do echo "== $f"; tail -2 $out/$f.err; done
# keep what travels back small: drop the per-dispatch traces, keep stats + counter tables
find $out -name "*_kernel_trace.csv" -delete
find $out -name "*_agent_info.csv" -delete
du -sh $out
