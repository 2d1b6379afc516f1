#!/bin/bash
# GPU box: rocprofv3 --pmc passes (ONE counter group per run, never combined with a trace) over tools/pmc_eg3d_run.py for
# the tri-plane kernels of configs[4]: where are the texel requests served (L2 / fabric = Infinity Cache + HBM) and how many
# bytes cross each boundary.  Summarised by tools/pmc_eg3d_summary.py into profiles/r03_pmc_eg3d.{json,txt}.
# usage: gpurun --timeout 900 -- bash tools/pmc_eg3d.sh
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/pmc_eg3d
rm -rf $O && mkdir -p $O
for c in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCC_REQ_sum TCC_READ_sum" "TCC_WRITE_sum TCC_ATOMIC_sum" \
         "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_EA0_WRREQ_sum TCC_EA0_ATOMIC_sum" "TCP_TCC_READ_REQ_sum TCP_TCC_ATOMIC_WITHOUT_RET_REQ_sum" \
         "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_WRITE_REQ_sum" "GRBM_GUI_ACTIVE SQ_WAVES" "SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAVE_CYCLES"; do
  d=$O/$(echo $c | tr ' ' '_')
  if rocprofv3 --pmc $c --output-format csv -d $d -o p -- python3 tools/pmc_eg3d_run.py > $O/last.log 2>&1; then echo "pmc $c done"; else echo "pmc $c FAILED: $(tail -2 $O/last.log | tr '\n' ' ')"; fi
done
python3 tools/pmc_eg3d_summary.py $O $O/r03_pmc_eg3d.json > $O/r03_pmc_eg3d.txt 2>&1
cat $O/r03_pmc_eg3d.txt
