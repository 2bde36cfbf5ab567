#!/bin/bash
# Run on the GPU box from the repo root: one bounded attempt at rocprofv3 PC sampling of the product chain (DESIGN.md
# section 7.0: "where select_cols_k16_kernel's waves wait").  Results land in gpurun_out/pcs_TAG/.
# usage: tools/pc_sample.sh TAG [stochastic|host_trap]
set -o pipefail
TAG=$1; METHOD=${2:-stochastic}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/pcs_$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
if [ "$METHOD" = stochastic ]; then UNIT="--pc-sampling-unit cycles --pc-sampling-interval 1048576"; else UNIT="--pc-sampling-unit time --pc-sampling-interval 100"; fi
timeout -k 10 420 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-method $METHOD $UNIT --kernel-trace --output-format csv -d $O/out \
    -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras > $O/run.log 2>&1
echo "exit $?" >> $O/run.log
ls -laR $O/out 2>/dev/null | head -40 >> $O/run.log
tail -30 $O/run.log
