"""CPU study (test infrastructure, not a pytest file): WHICH storage roundings carry the bf16 error of the forward?

VERDICT r3 item 1a.  ``oracle/lowp_oracle.LowpOracle`` restates the reference forward (models/yolo_test.py:214-272 over
models/common.py) in fp32 with a rounding hook at every point where the 16-bit HIP path stores a tensor; here the hooks are
switched per SITE (kind of tensor) and per LAYER GROUP (depth), on the weights BASELINE.md section 2 prescribes:

  only:<site>      only this site rounds, everything else fp32      -> the site's own contribution
  without:<site>   every site but this one rounds                   -> what keeping it in fp32 would buy
  layers:<group>   only the roundings inside this yaml-layer range  -> where in the depth the error is injected
  res32            bf16 everywhere, the Bottleneck shortcut chain of every C3 carried in fp32 (one rounding per conv input)
  subsets          candidate mixed-precision policies with their extra HBM bytes

    python tests/bf16_sites.py [case ...] [--out profiles/r04_bf16_sites.json]
"""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import msod_amd  # noqa: E402,F401
from msod_amd.models.configs import named_config  # noqa: E402
from msod_amd.models.yolo_test import Model  # noqa: E402
from msod_amd.utils.seeded import seeded_inputs, seeded_state_dict  # noqa: E402
from oracle import cft_oracle as O  # noqa: E402
from oracle.lowp_oracle import SITES, LowpOracle  # noqa: E402

CASES = {"l_x3_flir_256": ("cfg3", 1, 256, 256, 5), "cfg3_640": ("cfg3", 1, 640, 640, 0), "cfg3_256_s0": ("cfg3", 1, 256, 256, 0)}

# yaml layer ranges of yolov5l_fusion_transformerx3_FLIR_aligned.yaml (both streams)
GROUPS = {"P1-P2 (0-2,5-7)": lambda i: i in (0, 1, 2, 5, 6, 7), "P3 C3x9 (3-4,8-9)": lambda i: i in (3, 4, 8, 9),
          "CFT1+Add2 (10-12)": lambda i: 10 <= i <= 12, "P4 C3x9 (13-16)": lambda i: 13 <= i <= 16,
          "CFT2+Add2 (17-19)": lambda i: 17 <= i <= 19, "P5 SPP C3 (20-25)": lambda i: 20 <= i <= 25,
          "CFT3+Add2 (26-28)": lambda i: 26 <= i <= 28, "Add (29-31)": lambda i: 29 <= i <= 31, "head (32-46)": lambda i: i >= 32}


def metrics(raw, wraw):
    a = torch.cat([r.reshape(-1) for r in raw]); b = torch.cat([r.reshape(-1) for r in wraw])
    d = a.sigmoid() - b.sigmoid()
    return {"sig_max": round(d.abs().max().item(), 5), "sig_rms": round(d.pow(2).mean().sqrt().item(), 6),
            "raw_rms_rel": round(((a - b).pow(2).mean().sqrt() / b.std()).item(), 6)}


@torch.no_grad()
def main():
    args = sys.argv[1:]
    out = None
    if "--out" in args:
        out = args[args.index("--out") + 1]
        del args[args.index("--out"):args.index("--out") + 2]
    names = args or ["l_x3_flir_256"]
    bf = torch.bfloat16
    result = {}
    for n in names:
        cname, b, h, w, seed = CASES[n]
        cfg = named_config(cname)
        sd = seeded_state_dict(Model(cfg).state_dict(), seed=seed)
        rgb, ir = seeded_inputs(b, h, w, seed=seed)
        t0 = time.time()
        _, wraw = O.OracleModel(cfg)(sd, rgb, ir)
        rows = {}

        def run(tag, **kw):
            _, raw = LowpOracle(cfg, kw.pop("dtype", bf), **kw)(sd, rgb, ir)
            rows[tag] = metrics(raw, wraw)
            print(json.dumps({"case": n, "policy": tag, **rows[tag]}), flush=True)

        run("bf16 (all sites)")
        run("f16 (all sites)", dtype=torch.float16)
        run("res32", res32=True)
        for s in SITES:
            run("only:" + s, sites=[s])
        for s in SITES:
            run("without:" + s, sites=[x for x in SITES if x != s])
        for g, flt in GROUPS.items():
            run("layers:" + g, layer_filter=flt)
        acts = [s for s in SITES if not s.startswith("w_") and s != "image"]
        run("weights only", sites=["w_conv", "w_gpt", "w_detect"])
        run("activations only", sites=acts + ["image"])
        # candidate policies (fp32 kept where no MFMA operand is formed, i.e. where it costs bytes but no matrix rate)
        run("res32 + add/add2 fp32", res32=True, sites=[x for x in SITES if x not in ("add", "add2")])
        run("res32 + CFT internals fp32", res32=True, sites=[x for x in SITES if not x.startswith("gpt_")])
        run("res32, head (>=32) fp32 activations", res32=True, layer_filter=lambda i: i < 32)
        # mixed 16-bit policies that cost no bytes and no matrix rate: fp16 where the value range is normalised
        h = torch.float16
        gpt = [s for s in SITES if s.startswith("gpt_")] + ["w_gpt"]
        run("CFT block internals + weights in fp16, maps bf16", site_dtype={s: h for s in gpt})
        run("CFT in fp16, Add from unrounded Add2 sums", site_dtype={s: h for s in gpt}, sites=[x for x in SITES if x != "add"])
        # bounds no bf16-MFMA implementation can reach without doubling the matrix work (bf16 x bf16 products only)
        run("all weights fp16-exact, activations bf16 (needs f16 x bf16 products: not an MFMA form)", site_dtype={s: h for s in ("w_conv", "w_gpt", "w_detect")})
        run("all weights fp32-exact, activations bf16 (hi+lo split weights: 2x matrix work)", sites=acts + ["image"])
        result[n] = {"seconds": round(time.time() - t0, 1), "rows": rows}
    if out:
        with open(out, "w") as f:
            json.dump(result, f, indent=1)


if __name__ == "__main__":
    main()
