"""CPU precision study (not a pytest file, test infrastructure): where does the 16-bit error of the
forward come from, and which storage / rounding policy meets the 1e-2 sigmoid-space bound?

The HIP kernels accumulate in fp32 and round ONCE per layer output, so their error is dominated by
the storage roundings.  This script runs oracle/lowp_oracle.py (the fp32 oracle with rounding hooks at
exactly the points where the product stores a 16-bit tensor) for several policies:

  bf16        every activation / weight rounded to bf16 (round-1 product)
  f16         the same roundings in IEEE half (the reference's own GPU precision, test.py:66-68)
  bf16+res32  bf16, but the Bottleneck residual chain of a C3 kept in fp32

    python tests/precision_study.py [case ...]
"""
import json
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import msod_amd  # noqa: E402,F401
from msod_amd.models.configs import named_config  # noqa: E402
from msod_amd.models.yolo_test import Model  # noqa: E402
from msod_amd.utils.seeded import seeded_inputs, seeded_state_dict  # noqa: E402
from oracle import cft_oracle as O  # noqa: E402
from oracle.lowp_oracle import LowpOracle  # noqa: E402


def metrics(pred, raw, wpred, wraw):
    a = torch.cat([r.reshape(-1) for r in raw]); b = torch.cat([r.reshape(-1) for r in wraw])
    return {"rms_rel": ((a - b).pow(2).mean().sqrt() / b.std()).item(),
            "raw_max": (a - b).abs().max().item(),
            "sigmoid_max": (a.sigmoid() - b.sigmoid()).abs().max().item(),
            "conf_max": (pred[..., 4:] - wpred[..., 4:]).abs().max().item()}


CASES = {"cfg3_256": ("cfg3", 1, 256, 256, 3), "cfg2_256": ("cfg2", 2, 256, 256, 3),
         "s_x3": ("yolov5s_fusion_transformerx3_vedai", 2, 192, 320, 3), "cfg3_256_s5": ("cfg3", 1, 256, 256, 5),
         "cfg3_640": ("cfg3", 1, 640, 640, 0)}


@torch.no_grad()
def main():
    names = sys.argv[1:] or ["cfg3_256", "cfg2_256", "s_x3"]
    pols = {"bf16": (torch.bfloat16, False), "f16": (torch.float16, False), "bf16+res32": (torch.bfloat16, True)}
    for n in names:
        cname, b, h, w, seed = CASES[n]
        cfg = named_config(cname)
        model = Model(cfg)
        sd = seeded_state_dict(model.state_dict(), seed=seed)
        rgb, ir = seeded_inputs(b, h, w, seed=seed)
        wpred, wraw = O.OracleModel(cfg)(sd, rgb, ir)
        for pn, pol in pols.items():
            pred, raw = LowpOracle(cfg, pol[0], res32=pol[1])(sd, rgb, ir)
            print(json.dumps({"case": n, "policy": pn, **{k: round(v, 5) for k, v in metrics(pred, raw, wpred, wraw).items()}}), flush=True)


if __name__ == "__main__":
    main()
