import csv, sys, glob
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
n = float(sys.argv[2])
for r in csv.DictReader(open(f)):
    if any(k in r["Name"] for k in ("spp", "attention", "layernorm", "tokenize", "upsample", "copy_channels", "add_kernel", "detect", "focus", "bottleneck")):
        print(r["Name"][:50], r["Calls"], "per-forward ms %.3f" % (float(r["TotalDurationNs"]) / n / 1e6), "avg us %.1f" % (float(r["AverageNs"]) / 1e3))
