#!/bin/bash
# On the GPU box: correctness check + ablation timings of the band kernel.  usage: tools/band_quick.sh TAG [modes...]
TAG=${1:-q}; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/band_$TAG
mkdir -p $O
timeout -k 10 300 python3 $R/tools/fused_check.py > $O/check.log 2>&1; echo "check rc=$?" >> $O/check.log
for m in ${@:-0 1 3 7}; do
  echo "mode $m" >> $O/modes.log
  ACOSS_BAND_MODE=$m timeout -k 10 120 python3 $R/tools/fused_time.py 4096 4 >> $O/modes.log 2>&1
done
grep -v amdgpu.ids $O/check.log; grep -v amdgpu.ids $O/modes.log
