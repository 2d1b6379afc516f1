#!/bin/bash
cd $GRAFT_REPO_ROOT
export NERFMI_LIB=$GRAFT_REPO_ROOT/nerf_siren_amd/lib/libnerfmi_timing.so
for c in ${CHUNK_SETS:-"30,4,4" "29,6,6" "30,5,3" "30,3,5" "30,4,4"}; do
  NERFMI_SIREN_DW_CHUNKS=$c python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-psnr --no-extra 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['roofline']['kernels']['siren_dw_kernel']
print('$c', 'step %.4f ms'%d['ms_per_step'], 'dw avg %.4f ms fine %.4f'%(k['avg_launch_ms'], k['fine_launch']['avg_launch_ms']))"
done
