"""CPU precision study (not a pytest file, test infrastructure): where does the 16-bit error of the
forward come from, and which storage / rounding policy meets the 1e-2 sigmoid-space bound?

The HIP kernels accumulate in fp32 and round ONCE per layer output, so their error is dominated by
the storage roundings.  This script evaluates the oracle in fp32 with rounding hooks at exactly the
points where the product stores a 16-bit tensor, for several policies:

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


class Policy:
    def __init__(self, dtype, res32=False, tok32=True):
        self.dtype, self.res32 = dtype, res32

    def q(self, x):
        return x.to(self.dtype).float()


def emulate(cfg, sd, rgb, ir, pol):
    q = pol.q

    def conv(p, x, k, s, act=True, res=None, rnd=True):
        w = sd[p + "conv.weight"]
        if p + "bn.weight" in sd:
            scale = sd[p + "bn.weight"] / torch.sqrt(sd[p + "bn.running_var"] + O.BN_EPS)
            w = w * scale.view(-1, 1, 1, 1)
            b = sd[p + "bn.bias"] - sd[p + "bn.running_mean"] * scale
        else:
            b = sd[p + "conv.bias"]
        y = F.conv2d(x, q(w), b, s, k // 2)
        y = F.silu(y) if act else y
        if res is not None:
            y = y + res
        return q(y) if rnd else y

    def c3(p, x, n, shortcut):
        a = conv(p + "cv1.", x, 1, 1)
        b = conv(p + "cv2.", x, 1, 1)
        a32 = a
        for j in range(n):
            t = conv(f"{p}m.{j}.cv1.", a, 1, 1)
            if pol.res32 and shortcut:
                a32 = conv(f"{p}m.{j}.cv2.", t, 3, 1, res=a32, rnd=False)
                a = q(a32)
            else:
                a = conv(f"{p}m.{j}.cv2.", t, 3, 1, res=a if shortcut else None)
        return conv(p + "cv3.", torch.cat((a, b), 1), 1, 1)

    def lin(x, w, b):
        return F.linear(x, q(w), b)

    def gpt(p, r, t_):
        b, c, H, W = r.shape
        h, A = 8, 8
        rr = F.adaptive_avg_pool2d(r, (A, A)).reshape(b, c, -1)
        tt = F.adaptive_avg_pool2d(t_, (A, A)).reshape(b, c, -1)
        x = torch.cat([rr, tt], 2).permute(0, 2, 1) + sd[p + "pos_emb"]
        l = 0
        while f"{p}trans_blocks.{l}.ln_input.weight" in sd:
            bp = f"{p}trans_blocks.{l}."
            y = q(F.layer_norm(x, (c,), sd[bp + "ln_input.weight"], sd[bp + "ln_input.bias"], O.LN_EPS))
            dk = c // h
            sp = bp + "sa."
            qq = q(lin(y, sd[sp + "que_proj.weight"], sd[sp + "que_proj.bias"])).view(b, 128, h, dk).permute(0, 2, 1, 3)
            kk = q(lin(y, sd[sp + "key_proj.weight"], sd[sp + "key_proj.bias"])).view(b, 128, h, dk).permute(0, 2, 3, 1)
            vv = q(lin(y, sd[sp + "val_proj.weight"], sd[sp + "val_proj.bias"])).view(b, 128, h, dk).permute(0, 2, 1, 3)
            s = torch.matmul(qq, kk) / dk ** 0.5
            pe = q(torch.exp(s - s.max(-1, keepdim=True)[0]))
            o = torch.matmul(pe, vv) / pe.sum(-1, keepdim=True)
            o = q(o.permute(0, 2, 1, 3).reshape(b, 128, c))
            x = x + lin(o, sd[sp + "out_proj.weight"], sd[sp + "out_proj.bias"])
            y = q(F.layer_norm(x, (c,), sd[bp + "ln_output.weight"], sd[bp + "ln_output.bias"], O.LN_EPS))
            hid = q(F.gelu(lin(y, sd[bp + "mlp.0.weight"], sd[bp + "mlp.0.bias"])))
            x = x + lin(hid, sd[bp + "mlp.2.weight"], sd[bp + "mlp.2.bias"])
            l += 1
        x = F.layer_norm(x, (c,), sd[p + "ln_f.weight"], sd[p + "ln_f.bias"], O.LN_EPS)
        x = x.view(b, 2, A, A, c).permute(0, 1, 4, 2, 3)
        return (F.interpolate(x[:, 0].contiguous(), size=(H, W), mode="bilinear"),
                F.interpolate(x[:, 1].contiguous(), size=(H, W), mode="bilinear"))

    layers, save = O.build_graph(cfg)
    y = []
    x = rgb
    for L in layers:
        i, f, t = L["i"], L["f"], L["type"]
        p = f"model.{i}."
        if f == -4:
            xin = ir
        elif f == -1:
            xin = x
        elif isinstance(f, int):
            xin = y[f]
        else:
            xin = [x if j == -1 else y[j] for j in f]
        if t == "Conv":
            x = conv(p, xin, L["k"], L["s"])
        elif t == "Focus":
            z = torch.cat([xin[..., ::2, ::2], xin[..., 1::2, ::2], xin[..., ::2, 1::2], xin[..., 1::2, 1::2]], 1)
            x = conv(p + "conv.", q(z), L["k"], L["s"])
        elif t == "C3":
            x = c3(p, xin, L["n"], L["shortcut"])
        elif t == "SPP":
            a = conv(p + "cv1.", xin, 1, 1)
            x = conv(p + "cv2.", torch.cat([a] + [F.max_pool2d(a, k, 1, k // 2) for k in L["k"]], 1), 1, 1)
        elif t == "Concat":
            x = torch.cat(xin, 1)
        elif t == "Add":
            x = q(xin[0] + xin[1])
        elif t == "Add2":
            x = q(xin[0] + xin[1][L["index"]])
        elif t == "GPT":
            x = gpt(p, xin[0], xin[1])
        elif t == "nn.Upsample":
            x = F.interpolate(xin, scale_factor=float(L["scale"]), mode=L["mode"])
        elif t == "Detect":
            sdq = dict(sd)
            for j in range(len(xin)):
                sdq[f"{p}m.{j}.weight"] = q(sd[f"{p}m.{j}.weight"])
            ag = sd[p + "anchor_grid"] if p + "anchor_grid" in sd else O.sorted_anchors(L["anchors"])[1]
            x = O.detect(sdq, p, list(xin), L["nc"], ag)
        y.append(x if i in save else None)
    return x


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
    pols = {"bf16": Policy(torch.bfloat16), "f16": Policy(torch.float16), "bf16+res32": Policy(torch.bfloat16, res32=True)}
    for n in names:
        cname, b, h, w, seed = CASES[n]
        cfg = named_config(cname)
        model = Model(cfg)
        sd = seeded_state_dict(model.state_dict(), seed=seed)
        rgb, ir = seeded_inputs(b, h, w, seed=seed)
        wpred, wraw = O.OracleModel(cfg)(sd, rgb, ir)
        for pn, pol in pols.items():
            pred, raw = emulate(cfg, sd, rgb, ir, pol)
            print(json.dumps({"case": n, "policy": pn, **{k: round(v, 5) for k, v in metrics(pred, raw, wpred, wraw).items()}}), flush=True)


if __name__ == "__main__":
    main()
