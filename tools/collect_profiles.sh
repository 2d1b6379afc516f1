#!/bin/bash
# GPU box: kernel-trace stats + PMC passes (one counter group per run, never combined with traces) for the training
# and the inference step.  Outputs under gpurun_out/; summarise with tools/pmc_summary.py and copy into profiles/.
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/prof_train gpurun_out/prof_infer gpurun_out/pmc_train gpurun_out/pmc_infer
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_train -o t -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-psnr --no-opt-in > gpurun_out/train_prof.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_infer -o t -- python3 bench.py --mode infer --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/infer_prof.log 2>&1
for mode in train infer; do
  for c in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_WAIT_ANY SQ_WAVE_CYCLES" "TCC_HIT_sum TCC_MISS_sum"; do
    d=gpurun_out/pmc_$mode/$(echo $c | tr ' ' '_')
    rocprofv3 --pmc $c --output-format csv -d $d -o p -- python3 bench.py --mode $mode --steps 3 --warmup 2 --no-cpu-baseline --no-psnr --no-opt-in > gpurun_out/pmc_$mode.log 2>&1
    echo "pmc $mode $c done"
  done
done
