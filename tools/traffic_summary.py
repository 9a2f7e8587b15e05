"""Reduce the PMC passes of tools/pmc_traffic.sh to profiles/rNN_traffic.json (bytes per forward per kernel)."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

root, out = sys.argv[1], sys.argv[2]
meta = dict(a.split("=") for a in sys.argv[3:])
tot = {"FETCH_SIZE": defaultdict(float), "WRITE_SIZE": defaultdict(float)}
calls = defaultdict(int)
for c in tot:
    for f in glob.glob(os.path.join(root, c, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if row["Counter_Name"] != c:
                continue
            k = row["Kernel_Name"].split("(")[0].replace("void ", "")
            tot[c][k] += float(row["Counter_Value"]) * 1024.0        # counters are in KiB
            if c == "FETCH_SIZE":
                calls[k] += 1
GEMM_FAMILY = ("conv_gemm", "focus_conv", "bottleneck")   # the implicit-GEMM kernels bench.py times as one family
focus = [k for k in calls if "focus_conv" in k or "focus_s2d" in k]
n_fwd = calls[focus[0]] / 2 if focus else 1          # two Focus launches per forward
kern = {}
for k in sorted(calls, key=lambda k: -(2 * tot["FETCH_SIZE"][k] + tot["WRITE_SIZE"][k])):
    kern[k] = {"launches_per_forward": calls[k] / n_fwd,
               "hbm_read_bytes": 2.0 * tot["FETCH_SIZE"][k] / n_fwd,     # gfx950: FETCH_SIZE counts 64 B per 128-B request
               "hbm_write_bytes": tot["WRITE_SIZE"][k] / n_fwd}
# Kernels of the profiled PROCESS that are not part of a forward: ATen / runtime kernels of the set-up (weight upload and packing, input
# generation, buffer fills; the forward itself launches no ATen op - their counts do not grow with --steps) and the one-time weight re-pack.
SETUP = ("at::native", "__amd_rocclr", "bneck_pack_w2")
byt = lambda v: v["hbm_read_bytes"] + v["hbm_write_bytes"]   # noqa: E731
gemm = sum(byt(v) for k, v in kern.items() if any(t in k for t in GEMM_FAMILY))
allb = sum(byt(v) for v in kern.values())
setup = sum(byt(v) for k, v in kern.items() if any(t in k for t in SETUP))
res = {"config": meta.get("config", "cfg3"), "batch": int(meta.get("batch", 64)), "size": int(meta.get("size", 640)),
       "dtype": meta.get("dtype", "bf16"), "forwards_profiled": n_fwd, "gemm_bytes_per_forward": gemm,
       "forward_kernels_bytes_per_forward": allb - setup,
       "all_kernels_bytes_per_forward": allb,
       "note": "all_kernels_* = every kernel of the profiled process / forwards profiled, i.e. INCLUDING the one-time set-up traffic "
               "(the figure rounds 1-3 quoted); forward_kernels_* excludes the set-up kernels (names containing " + ", ".join(SETUP) + ")",
       "kernels": kern}
json.dump(res, open(out, "w"), indent=1)
print(json.dumps({k: res[k] for k in ("forwards_profiled", "gemm_bytes_per_forward", "forward_kernels_bytes_per_forward", "all_kernels_bytes_per_forward")}))
