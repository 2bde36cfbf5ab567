#!/bin/bash
# On the GPU box: PMC passes of the fused path at bench size.  usage: tools/band_pmc.sh TAG [ACOSS_BAND_MODE]
set -o pipefail
TAG=${1:-p}
export ACOSS_BAND_MODE=${2:-0}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/bandpmc_$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_SALU --kernel-trace --output-format csv -d $O/sq -- python3 $R/tools/fused_time.py 4096 2 > $O/sq.log 2>&1 || exit 5
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_SMEM --kernel-trace --output-format csv -d $O/sq2 -- python3 $R/tools/fused_time.py 4096 2 > $O/sq2.log 2>&1 || exit 6
rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_INSTS_WAVE32_LDS SQ_INSTS_VALU_MFMA_MOPS_F32 --kernel-trace --output-format csv -d $O/sq3 -- python3 $R/tools/fused_time.py 4096 2 > $O/sq3.log 2>&1 || echo "sq3 failed"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/wr -- python3 $R/tools/fused_time.py 4096 2 > $O/wr.log 2>&1 || exit 3
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/rd -- python3 $R/tools/fused_time.py 4096 2 > $O/rd.log 2>&1 || exit 4
cd $R && python3 tools/pmc_json.py $O/pmc.json $O/sq $O/sq2 $O/sq3 $O/wr $O/rd > $O/pmc_table.txt
rm -rf $O/sq $O/sq2 $O/sq3 $O/wr $O/rd
python3 - <<PY
import json
d=json.load(open("$O/pmc.json"))
for k,v in d.items():
    if isinstance(v,dict) and 'band_kernel' in k:
        print(k)
        for kk,vv in sorted(v.items()): print('   %-32s %.4g'%(kk,vv))
PY
