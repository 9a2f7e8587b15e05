"""Print a gemm_bench.py json as a per-shape table: us per variant, ratio of the last to the first."""
import json, sys
d = json.load(open(sys.argv[1]))
rows = d if isinstance(d, list) else d.get("results", d)
for r in rows:
    v = r["variants"]; ks = list(v)
    us = [v[k]["us"] for k in ks]
    print(f"{r['shape'][:42]:42s} x{r['count']:2d} " + " ".join(f"{k}:{u:7.1f}" for k, u in zip(ks, us)) + f"  {us[-1]/us[0]:5.3f}  diff {max(v[k]['maxdiff_vs_first'] for k in ks)}")
