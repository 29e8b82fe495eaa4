#!/bin/bash
# further runs of the stones probe (64 complete games each at the benchmark's budget), one seed per call: tools/probe_more.sh SEED
s=$1
timeout -k 10 1150 python tools/stones_probe.py config2 64 400 0 $s > gpurun_out/r04_probe_$s.json 2> gpurun_out/r04_probe_$s.err; tail -c 330 gpurun_out/r04_probe_$s.json
