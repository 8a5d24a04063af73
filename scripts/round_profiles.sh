#!/bin/bash
# One GPU-box call per configuration: PMC passes -> summary keyed by the kernel-source hash, the bench line that reads it
# (with cpu_baseline), and rocprofv3 kernel-trace stats of the same command.  bash scripts/round_profiles.sh <round> <config>
R=${1:-r02}; C=${2:-c2}
bash scripts/pmc_all.sh $R $C > gpurun_out/${R}_pmc_all_$C.txt 2>&1 || { tail -20 gpurun_out/${R}_pmc_all_$C.txt; exit 1; }
tail -12 gpurun_out/${R}_pmc_all_$C.txt | cut -c1-400
bash scripts/prof_config.sh ${R}_prof $C 3 > gpurun_out/${R}_prof_$C.txt 2>&1 || { tail -20 gpurun_out/${R}_prof_$C.txt; exit 1; }
cp gpurun_out/${R}_prof/kernel_stats_$C.csv gpurun_out/${R}_kernel_stats_$C.csv
head -12 gpurun_out/${R}_kernel_stats_$C.csv | cut -c1-160
