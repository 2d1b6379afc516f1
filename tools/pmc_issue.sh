# GPU box: where do a wave's cycles go?  SQ issue / wait counters for the training kernels of one field, one group per pass.
# usage: gpurun -- bash tools/pmc_issue.sh [siren|nerf] > gpurun_out/pmc_issue_<field>.txt
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/pmc_issue; rm -rf $O; mkdir -p $O
FIELD=${1:-siren}   # siren (the headline field) | nerf
PMC="--steps 3 --warmup 2 --no-cpu-baseline --no-psnr --no-opt-in --no-extra --no-kernel-events --mode train --field $FIELD"
i=0
for c in "SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES" "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC" "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS" "SQ_INSTS_VALU SQ_INSTS_MFMA" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" "SQ_INST_CYCLES_VMEM SQ_INSTS_LDS" "SQ_WAIT_ANY SQ_INSTS_VMEM" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"; do
  i=$((i+1))
  rocprofv3 --pmc $c --output-format csv -d $O/p$i -o p -- python3 bench.py $PMC > $O/p$i.log 2>&1 || echo "pass $i ($c) failed"
done
python3 - <<PY
import csv,glob,collections
acc=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.defaultdict(lambda: collections.Counter())
for f in glob.glob("$O/p*/**/p_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"].split("(")[0][-44:]
        acc[k][r["Counter_Name"]]+=float(r["Counter_Value"]); n[k][r["Counter_Name"]]+=1
for k,v in acc.items():
    if v.get("SQ_WAVE_CYCLES",0)/max(1,n[k]["SQ_WAVE_CYCLES"])>1e7:
        print(k)
        for c in sorted(v): print("   %-28s %.4g per launch"%(c, v[c]/n[k][c]))
PY
