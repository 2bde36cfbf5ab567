"""Copy what tools/collect_profiles.sh TAG left under gpurun_out/prof_TAG/ into profiles/ under round-3 names, with the meta
file bench.py uses to quote the profile's averages beside its live numbers.  usage: python tools/adopt_profiles.py TAG"""
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
src = os.path.join(ROOT, "gpurun_out", "prof_" + tag)
dst = os.path.join(ROOT, "profiles")
line = json.loads(open(os.path.join(src, "bench.json")).read())
for a, b in (("bench.json", "r03_bench.json"), ("kernel_stats.csv", "r03_kernel_stats.csv"), ("kernel_stats_extras.csv", "r03_kernel_stats_with_side_blocks.csv"),
             ("pmc.json", "r03_pmc.json"), ("pmc_table.txt", "r03_pmc_table.txt")):
    shutil.copy(os.path.join(src, a), os.path.join(dst, b))
cmd = open(os.path.join(src, "command.txt")).read().strip()
meta = {"command": "python3 " + cmd + " --steps 5 --warmup 2 --no-cpu-baseline --no-extras (under rocprofv3 --kernel-trace --stats)",
        "pairs_per_step": line["config"]["pairs_per_step_per_gpu"], "path": line["config"]["path"], "frames": 1000, "songs": 1000,
        "bench_line_value": line["value"], "bench_line_ms_per_step": line["ms_per_step"],
        "commit": subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip(),
        "collected_by": "tools/collect_profiles.sh " + tag}
json.dump(meta, open(os.path.join(dst, "r03_profile_meta.json"), "w"), indent=1)
print(json.dumps(meta, indent=1))
