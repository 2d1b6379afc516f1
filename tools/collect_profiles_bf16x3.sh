cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/prof_r02_fast; rm -rf $O; mkdir -p $O/summary
COMMON="--steps 20 --warmup 5 --no-cpu-baseline --no-psnr --no-opt-in --no-extra"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/a -o t -- python3 bench.py $COMMON --math bf16x3 --mode train > $O/a.log 2>&1
cp $(find $O/a -name "t_kernel_stats.csv" | head -1) $O/summary/r02_train_bf16x3_kernel_stats.csv
rocprofv3 --kernel-trace --stats --output-format csv -d $O/b -o t -- python3 bench.py $COMMON --math bf16x3 --mode infer --field siren > $O/b.log 2>&1
cp $(find $O/b -name "t_kernel_stats.csv" | head -1) $O/summary/r02_siren_infer_bf16x3_kernel_stats.csv
grep -h -o '"ms_per_step": [0-9.]*' $O/a.log $O/b.log
head -4 $O/summary/r02_siren_infer_bf16x3_kernel_stats.csv | cut -c1-160
