"""Summarise rocprofv3 counter_collection CSVs per kernel name (average per dispatch)."""
import csv
import glob
import os
import sys
from collections import defaultdict

root = sys.argv[1]
filt = sys.argv[2] if len(sys.argv) > 2 else "conv_gemm"
agg = defaultdict(lambda: defaultdict(float))
cnt = defaultdict(lambda: defaultdict(int))
dur = defaultdict(list)
for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            k = row.get("Kernel_Name", "?")
            if filt not in k:
                continue
            k = k.split("(")[0].replace("void ", "")
            agg[k][row["Counter_Name"]] += float(row["Counter_Value"])
            cnt[k][row["Counter_Name"]] += 1
for f in glob.glob(os.path.join(root, "sq1", "**", "*kernel_trace.csv"), recursive=True):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            k = row["Kernel_Name"].split("(")[0].replace("void ", "")
            if filt in k:
                dur[k].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3)
for k in sorted(agg):
    a = {c: agg[k][c] / cnt[k][c] for c in agg[k]}
    print(k, f"  avg {sum(dur[k]) / max(1, len(dur[k])):.1f} us over {len(dur[k])} dispatches (profiled pass)")
    for c in sorted(a):
        print(f"   {c:36s} {a[c]:16.1f}")
    wc = a.get("SQ_WAVE_CYCLES")
    if wc:
        print("   -- wave-cycle split: active %.0f%%  wait_any %.0f%%  wait_inst %.0f%%   VALU/MFMA instr ratio n/a" % (
            100 * a.get("SQ_ACTIVE_INST_ANY", 0) / wc, 100 * a.get("SQ_WAIT_ANY", 0) / wc, 100 * a.get("SQ_WAIT_INST_ANY", 0) / wc))
    if "FETCH_SIZE" in a:
        print("   -- HBM: fetch %.1f MB (x2 gfx950 correction = %.1f MB), write %.1f MB" % (a["FETCH_SIZE"] / 1024, a["FETCH_SIZE"] / 512, a.get("WRITE_SIZE", 0) / 1024))
