#!/bin/bash
# GPU box helper: inference bench + kernel-trace of the training bench, printing the three MLP kernels' averages.
# usage: gpurun -- bash tools/benchpair.sh
python bench.py --mode infer --steps 50 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('infer', d['ms_per_step'], d['roofline']['avg_launch_ms'])"
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_train -o t -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-psnr --no-opt-in > gpurun_out/train_prof.log 2>&1; grep -o "ms_per_step\": [0-9.]*" gpurun_out/train_prof.log
head -4 gpurun_out/prof_train/t_kernel_stats.csv | python -c "
import csv,sys
for r in csv.DictReader(sys.stdin): print(r['Name'][:60], r['Calls'], float(r['AverageNs'])/1e3)"
