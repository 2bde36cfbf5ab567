"""Per-kernel averages of rocprofv3 --pmc CSV output (dev tool): python tools/pmc_table.py DIR [DIR...]"""
import collections, csv, glob, sys
import numpy as np
tab = collections.defaultdict(lambda: collections.defaultdict(list))
for d in sys.argv[1:]:
    for f in glob.glob(d + "/*/*_counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "acoss::" not in k:
                continue
            k = k.split("(")[0].replace("void acoss::", "").replace("acoss::", "")
            tab[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
            tab[k]["dur_us"].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
            tab[k]["_grid"].append(float(r["Grid_Size"]))
            tab[k]["_vgpr"].append(float(r["VGPR_Count"])); tab[k]["_sgpr"].append(float(r["SGPR_Count"])); tab[k]["_lds"].append(float(r["LDS_Block_Size"]))
for k, c in tab.items():
    print("== %s  (grid %d thr, vgpr %d sgpr %d lds %d)  %.1f us" % (k, np.mean(c["_grid"]), np.mean(c["_vgpr"]), np.mean(c["_sgpr"]), np.mean(c["_lds"]), np.mean(c["dur_us"])))
    for name in sorted(c):
        if name.startswith("_") or name == "dur_us":
            continue
        print("     %-24s %.4g" % (name, np.mean(c[name])))
    g = lambda n: np.mean(c[n]) if n in c else float("nan")
    if "SQ_WAVE_CYCLES" in c:
        wc = g("SQ_WAVE_CYCLES")
        print("     -> per wave: %.0f quad-cycles; VALU insts/wave %.0f; SALU/wave %.0f; active VALU frac %.2f; wait_any frac %.2f; wait_inst frac %.2f" % (
            wc / g("SQ_WAVES"), g("SQ_INSTS_VALU") / g("SQ_WAVES"), g("SQ_INSTS_SALU") / g("SQ_WAVES"),
            g("SQ_ACTIVE_INST_VALU") / wc, g("SQ_WAIT_ANY") / wc, g("SQ_WAIT_INST_ANY") / wc))
        print("     -> busy cycles %.4g ; avg waves in flight = wave_cycles*4/busy = %.1f per SQ(?)" % (g("SQ_BUSY_CYCLES"), 4 * wc / g("SQ_BUSY_CYCLES")))
    if "GRBM_GUI_ACTIVE" in c:
        print("     -> clock ~ %.2f GHz" % (g("GRBM_GUI_ACTIVE") / 8 / (np.mean(c["dur_us"]) * 1e-6) / 1e9))
