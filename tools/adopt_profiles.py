"""Copy what tools/collect_profiles.sh TAG left under gpurun_out/prof_TAG/ into profiles/ under this round's names (ROUND below), with the meta
file bench.py uses to quote the profile's averages beside its live numbers.  usage: python tools/adopt_profiles.py TAG [rNN]"""
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
ROUND = sys.argv[2] if len(sys.argv) > 2 else "r05"
src = os.path.join(ROOT, "gpurun_out", "prof_" + tag)
dst = os.path.join(ROOT, "profiles")
line = json.loads(open(os.path.join(src, "bench.json")).read())
for a, b in (("bench.json", ROUND + "_bench.json"), ("kernel_stats.csv", ROUND + "_kernel_stats.csv"), ("kernel_stats_extras.csv", ROUND + "_kernel_stats_with_side_blocks.csv"),
             ("pmc.json", ROUND + "_pmc.json"), ("pmc_table.txt", ROUND + "_pmc_table.txt")):
    shutil.copy(os.path.join(src, a), os.path.join(dst, b))
cmd = open(os.path.join(src, "command.txt")).read().strip()
meta = {"command": "python3 " + cmd + " --steps 5 --warmup 2 --no-cpu-baseline --no-extras (under rocprofv3 --kernel-trace --stats)",
        "pairs_per_step": line["config"]["pairs_per_step_per_gpu"], "path": line["config"]["path"], "frames": 1000, "songs": 1000,
        "bench_line_value": line["value"], "bench_line_ms_per_step": line["ms_per_step"],
        # the commit whose kernels were profiled: the last one that touched the library's sources when the profile was taken
        "commit": subprocess.run(["git", "-C", ROOT, "log", "-1", "--format=%h", "--", "acoss_amd/csrc", "include"], capture_output=True, text=True).stdout.strip(),
        "head_when_adopted": subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip(),
        "collected_by": "tools/collect_profiles.sh " + tag}
json.dump(meta, open(os.path.join(dst, ROUND + "_profile_meta.json"), "w"), indent=1)
print(json.dumps(meta, indent=1))
