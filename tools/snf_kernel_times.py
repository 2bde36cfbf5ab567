"""Durations of snf_gemm_nt_kernel's 32-pair launches from rocprofv3 --kernel-trace directories (dev tool): python tools/snf_kernel_times.py DIR..."""
import csv, glob, sys
for d in sys.argv[1:]:
    f = glob.glob(d + "/*/*_kernel_trace.csv")[0]
    ts = sorted((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in csv.DictReader(open(f)) if "snf_gemm" in r["Kernel_Name"])
    big = ts[len(ts) // 2:]
    print("%-40s %d launches; the 32-pair ones: median %.0f us (min %.0f, max %.0f); 6830 us = the f64 matrix peak" % (d, len(ts), big[len(big) // 2], big[0], big[-1]))
