#!/bin/bash
# Run on the GPU box from the repo root (one gpurun call): the un-profiled bench line, the rocprofv3 --kernel-trace --stats
# summary of the SAME command line (shortened to 5 steps) and the PMC passes behind profiles/ (each counter set in its own
# run with --kernel-trace only; FETCH_SIZE and WRITE_SIZE separately).  Results land in gpurun_out/prof_TAG/; copy them
# into profiles/ with tools/adopt_profiles.py TAG (run in the build container after the merge).
# usage: tools/collect_profiles.sh TAG [bench args]
set -o pipefail
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/prof_$TAG
mkdir -p $O
python3 $R/bench.py "$@" > $O/bench.log 2>&1 || exit 1
grep -E '^\{' $O/bench.log > $O/bench.json
echo "bench.py $*" > $O/command.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras "$@" > $O/stats.log 2>&1 || exit 2
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/wr -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras "$@" > $O/wr.log 2>&1 || exit 3
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/rd -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras "$@" > $O/rd.log 2>&1 || exit 4
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_SALU --kernel-trace --output-format csv -d $O/sq -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras "$@" > $O/sq.log 2>&1 || exit 5
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_SMEM --kernel-trace --output-format csv -d $O/sq2 -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras "$@" > $O/sq2.log 2>&1 || exit 6
cd $R && python3 tools/pmc_json.py $O/pmc.json $O/wr $O/rd $O/sq $O/sq2 > $O/pmc_table.txt
cp "$(ls -S $O/stats/*/*_kernel_stats.csv | head -1)" $O/kernel_stats.csv
# one more kernel trace with the side blocks (other forms of the filter, configs 3-5, scattering CSM): their kernels' durations
cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/statsx -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-child-ranks "$@" > $O/statsx.log 2>&1 || exit 7
# (the side blocks start child ranks of their own -- the sharded config-3 job --: every process leaves a summary; keep the parent's, the largest)
cd $R && cp "$(ls -S $O/statsx/*/*_kernel_stats.csv | head -1)" $O/kernel_stats_extras.csv
rm -rf $O/statsx $O/stats $O/wr $O/rd $O/sq $O/sq2
echo done
