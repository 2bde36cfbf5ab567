#!/bin/bash
# On the GPU box: band kernel ablations (ACOSS_BAND_MODE) and two PMC passes.  usage: tools/band_ablate.sh TAG
set -o pipefail
TAG=${1:-ab}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/band_$TAG
mkdir -p $O
for m in 0 1 2 4 3 5 6 7; do
  echo "mode $m" >> $O/modes.log
  ACOSS_BAND_MODE=$m python3 $R/tools/fused_time.py 4096 4 >> $O/modes.log 2>&1
done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_SALU --kernel-trace --output-format csv -d $O/sq -- python3 $R/tools/fused_time.py 4096 2 > $O/sq.log 2>&1 || exit 5
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_SMEM --kernel-trace --output-format csv -d $O/sq2 -- python3 $R/tools/fused_time.py 4096 2 > $O/sq2.log 2>&1 || exit 6
rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_VALU_MFMA_MOPS_F32 --kernel-trace --output-format csv -d $O/sq3 -- python3 $R/tools/fused_time.py 4096 2 > $O/sq3.log 2>&1 || echo "sq3 failed" >> $O/modes.log
cd $R && python3 tools/pmc_json.py $O/pmc.json $O/sq $O/sq2 $O/sq3 > $O/pmc_table.txt
rm -rf $O/sq $O/sq2 $O/sq3
cat $O/modes.log | grep -v amdgpu.ids
