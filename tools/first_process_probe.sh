#!/bin/bash
# (dev tool) bench lines of three successive processes on a box, each under rocprofv3 --kernel-trace: per process the step time, the mask
# stage's time, and inside the trace the time the GPU spends IN kernels per step against the wall time per step (gaps between kernels)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for i in 1 2 3; do
  rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/fp_$i -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras > $R/gpurun_out/fp_$i.log 2>&1
  python3 - <<PY
import json, csv, glob
d=[json.loads(l) for l in open("$R/gpurun_out/fp_$i.log") if l.startswith("{")][0]
f=sorted(glob.glob("$R/gpurun_out/fp_$i/*/*_kernel_trace.csv"))[0]
rows=[(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))]
rows.sort()
# the timed steps: the last 10 occurrences of the strip kernel start a step each
starts=[k for k,r in enumerate(rows) if "crp_rows32_kernel" in r[2]][-10:]
a, b = starts[0], starts[-1]
seg = rows[a:b]
busy = sum(e - s for s, e, _ in seg) / 1e6
wall = (rows[b][0] - rows[a][0]) / 1e6
gaps = sorted(((seg[k + 1][0] - seg[k][1]) / 1e3, seg[k][2][:40], seg[k + 1][2][:40]) for k in range(len(seg) - 1))[-3:]
print("process $i: %.0f pair-scores/s, %.3f ms/step, mask_bits %.3f; 9 steps in the trace: wall %.2f ms, inside kernels %.2f ms; largest gaps (us): %s" % (d["value"], d["ms_per_step"], d["stage_ms"]["mask_bits"], wall, busy, gaps))
PY
done
