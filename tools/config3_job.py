"""The config-3 job of bench.py alone (2000 DA-TACOS-shaped songs through the plugin), for profiling: python tools/config3_job.py"""
import json, os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
res = bench.config3_job(tempfile.mkdtemp(prefix="acoss_c3_"))
print(json.dumps({k: res[k] for k in ("all_pairwise_seconds", "pairs_per_s", "Ds_crc32")}))
