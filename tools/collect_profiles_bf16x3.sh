# GPU box: kernel traces of the opt-in split-bf16 math (both fields, training and inference) -> gpurun_out/prof_r03_fast/summary/
# usage: gpurun --timeout 900 -- bash tools/collect_profiles_bf16x3.sh    (then copy summary/* into profiles/)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/prof_r03_fast; rm -rf $O; mkdir -p $O/summary
COMMON="--steps 20 --warmup 5 --no-cpu-baseline --no-psnr --no-opt-in --no-extra --math bf16x3"
for cfg in "siren_train:--field siren --mode train" "siren_infer:--field siren --mode infer" "train:--field nerf --mode train" "infer:--field nerf --mode infer"; do
  name=${cfg%%:*}; args=${cfg#*:}
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/$name -o t -- python3 bench.py $COMMON $args > $O/$name.log 2>&1
  cp $(find $O/$name -name "t_kernel_stats.csv" | head -1) $O/summary/r03_${name}_bf16x3_kernel_stats.csv
  grep -h -o '"ms_per_step": [0-9.]*' $O/$name.log | head -1 | sed "s/^/$name /"
done
