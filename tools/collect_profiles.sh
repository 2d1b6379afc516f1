#!/bin/bash
# (round 3: the FiLM-SIREN field is the headline; `--field nerf` = the reference's live ReLU NeRF)
# GPU box: kernel-trace stats + PMC passes (one counter group per run, never combined with traces) for the NeRF training /
# inference steps, the FiLM-SIREN training / inference steps and the EG3D renderer.  Outputs under gpurun_out/prof_r03/;
# summarise with tools/pmc_summary.py and copy into profiles/ (tools/collect_profiles.sh does the copying at the end into
# gpurun_out/prof_r03/summary/, which is what gets committed as profiles/r03_*).
# usage: gpurun --timeout 1100 -- bash tools/collect_profiles.sh
#        PROFILE_SET="siren_train siren_infer" gpurun ... -- bash tools/collect_profiles.sh   (only these steps; no EG3D trace;
#        pmc_latest.json then = the committed profiles/pmc_latest.json updated with the new summaries)
set -e
SET=${PROFILE_SET:-"siren_train siren_infer train infer eg3d"}
want() { case " $SET " in *" $1 "*) return 0;; *) return 1;; esac; }
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/prof_r03
rm -rf $O && mkdir -p $O/summary
COMMON="--steps 20 --warmup 5 --no-cpu-baseline --no-psnr --no-opt-in --no-extra"
run_trace() {   # name, bench args
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_$1 -o t -- python3 bench.py $COMMON $2 > $O/trace_$1.log 2>&1
  cp $O/trace_$1/*/t_kernel_stats.csv $O/summary/r03_$1_kernel_stats.csv 2>/dev/null || cp $(find $O/trace_$1 -name "t_kernel_stats.csv" | head -1) $O/summary/r03_$1_kernel_stats.csv
  grep -o '"ms_per_step": [0-9.]*' $O/trace_$1.log | head -1 | sed "s/^/$1 /"
}
want siren_train && run_trace siren_train "--field siren --mode train"
want siren_infer && run_trace siren_infer "--field siren --mode infer"
want train && run_trace train "--field nerf --mode train"
want infer && run_trace infer "--field nerf --mode infer"
if want eg3d; then
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_eg3d -o t -- python3 tools/bench_eg3d.py > $O/trace_eg3d.log 2>&1
  cp $(find $O/trace_eg3d -name "t_kernel_stats.csv" | head -1) $O/summary/r03_eg3d_kernel_stats.csv
fi
echo "traces done"
PMC="--steps 3 --warmup 2 --no-cpu-baseline --no-psnr --no-opt-in --no-extra --no-kernel-events"
for cfg in "siren_train:--field siren --mode train" "siren_infer:--field siren --mode infer" "train:--field nerf --mode train" "infer:--field nerf --mode infer"; do
  name=${cfg%%:*}; args=${cfg#*:}
  want $name || continue
  for c in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_WAIT_ANY SQ_WAVE_CYCLES"; do
    d=$O/pmc_$name/$(echo $c | tr ' ' '_')
    rocprofv3 --pmc $c --output-format csv -d $d -o p -- python3 bench.py $PMC $args > $O/pmc_$name.log 2>&1
    echo "pmc $name $c done"
  done
  python3 tools/pmc_summary.py $O/pmc_$name $O/summary/r03_pmc_$name.json > $O/summary/r03_pmc_$name.txt
done
echo "all done"; ls $O/summary
# pmc_latest.json = union of the four per-step summaries (what bench.py reads for roofline.traffic)
python3 - <<PY
import json, glob
import os
u = json.load(open("profiles/pmc_latest.json")) if os.path.exists("profiles/pmc_latest.json") else {}
for f in sorted(glob.glob("$O/summary/r03_pmc_*.json")):
    u.update(json.load(open(f)))
json.dump(u, open("$O/summary/pmc_latest.json", "w"), indent=1, sort_keys=True)
print("pmc_latest.json:", len(u), "kernels")
PY
