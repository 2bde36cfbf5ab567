"""Per-kernel means of rocprofv3 --pmc passes as JSON (dev tool): python tools/pmc_json.py OUT.json DIR [DIR...]"""
import collections, csv, glob, json, sys
import numpy as np
tab = collections.defaultdict(lambda: collections.defaultdict(list))
for d in sys.argv[2:]:
    for f in glob.glob(d + "/*/*_counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "acoss::" not in k:
                continue
            k = k.split("(")[0].replace("void acoss::", "").replace("acoss::", "")
            tab[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
            tab[k]["avg_us_under_pmc"].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
            tab[k]["vgpr"].append(float(r["VGPR_Count"]))
            tab[k]["lds_bytes"].append(float(r["LDS_Block_Size"]))
out = {k: {n: float(np.mean(v)) for n, v in c.items()} for k, c in tab.items()}
for k, c in out.items():
    if "WRITE_SIZE" in c and "FETCH_SIZE" in c:
        # MI355X guide: KiB units; FETCH_SIZE under-reports wide coalesced reads by 2x on gfx950
        c["hbm_write_GB"] = c["WRITE_SIZE"] * 1024 / 1e9
        c["hbm_fetch_GB_x2_corrected"] = 2 * c["FETCH_SIZE"] * 1024 / 1e9
import os
out["_pairs_per_step"] = int(os.environ.get("ACOSS_PROFILE_PAIRS", "4096"))
out["_path"] = os.environ.get("ACOSS_PROFILE_PATH", "fast16")
json.dump(out, open(sys.argv[1], "w"), indent=1, sort_keys=True)
out.pop("_pairs_per_step")
out.pop("_path")
for k in sorted(out):
    c = out[k]
    print("%-50s %9.1f us  W %.3f GB  R(x2) %.3f GB  vgpr %d" % (k, c["avg_us_under_pmc"], c.get("hbm_write_GB", float("nan")), c.get("hbm_fetch_GB_x2_corrected", float("nan")), c["vgpr"]))
