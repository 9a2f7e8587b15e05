"""Storage-precision model of the 16-bit HIP path, on the CPU.

TEST INFRASTRUCTURE ONLY (same rules as ``cft_oracle.py``: imported by ``tests/`` only, never by
the product package).

The 16-bit kernels multiply 16-bit operands exactly, accumulate in fp32 and round ONCE per stored
tensor.  Their deviation from the fp32 reference forward is therefore - to first order - the sum
of those storage roundings.  ``LowpOracle`` evaluates the reference algorithm (the functions of
``cft_oracle.py``, i.e. reference models/common.py:36-50, 99-109, 131-143, 154-179, 211-243,
430-639 and models/yolo_test.py:25-64) in fp32 and applies ``x.to(dtype).float()`` at exactly the
points where the product stores a 16-bit tensor:

* folded conv / linear weights (BN folded in fp32 first, utils/torch_utils.py:181-201);
* every Conv / C3 / SPP / Focus output (after bias + SiLU + residual add), Add / Add2 outputs;
* in the CFT block: LayerNorm outputs, q/k/v, the exponentials P, the attention output and the
  GELU hidden layer; the token residual stream, ln_f and the Detect logits stay fp32.

It answers two questions the fp32 oracle cannot:
  1. is the HIP 16-bit result what its storage precision predicts (tight bound, same roundings), and
  2. how far can ANY implementation with 16-bit storage of that type be from the fp32 reference
     (bf16: 1.2-1.5e-2 in sigmoid space on the lively seeded weights; fp16: 1.3-1.6e-3).
"""
import torch
import torch.nn.functional as F

from .cft_oracle import BN_EPS, LN_EPS, build_graph, detect, sorted_anchors

def _forward(cfg, sd, rgb, ir, q0, res32=False, sites=None, layer_filter=None, site_q=None):
    """``sites``: None = every storage rounding is applied (the model of the shipped kernels); otherwise the set of site
    names (SITES) whose rounding is applied - every other site keeps fp32 (the ablation tests/bf16_sites.py runs).
    ``layer_filter``: optional predicate on the yaml layer index; roundings of layers it rejects are skipped.
    ``site_q``: optional {site: rounding function} overriding ``q0`` for those sites (mixed-precision policies, e.g. the
    CFT block's internals in fp16 beside bf16 feature maps)."""
    cur = {"i": -1}

    def q(x, site):
        if sites is not None and site not in sites:
            return x
        if layer_filter is not None and not layer_filter(cur["i"]):
            return x
        if site_q is not None and site in site_q:
            return site_q[site](x)
        return q0(x)

    def conv(p, x, k, s, act=True, res=None, rnd=True, site="conv"):
        w = sd[p + "conv.weight"]
        if p + "bn.weight" in sd:
            scale = sd[p + "bn.weight"] / torch.sqrt(sd[p + "bn.running_var"] + BN_EPS)
            w = w * scale.view(-1, 1, 1, 1)
            b = sd[p + "bn.bias"] - sd[p + "bn.running_mean"] * scale
        else:
            b = sd[p + "conv.bias"]
        y = F.conv2d(x, q(w, "w_conv"), b, s, k // 2)
        y = F.silu(y) if act else y
        if res is not None:
            y = y + res
        return q(y, site) if rnd else y

    def c3(p, x, n, shortcut):
        a = conv(p + "cv1.", x, 1, 1, site="c3_cv12")
        b = conv(p + "cv2.", x, 1, 1, site="c3_cv12")
        a32 = a
        for j in range(n):
            t = conv(f"{p}m.{j}.cv1.", a, 1, 1, site="bneck_hidden")
            if res32 and shortcut:
                a32 = conv(f"{p}m.{j}.cv2.", t, 3, 1, res=a32, rnd=False)
                a = q(a32, "bneck_out")
            else:
                a = conv(f"{p}m.{j}.cv2.", t, 3, 1, res=a if shortcut else None, site="bneck_out")
        return conv(p + "cv3.", torch.cat((a, b), 1), 1, 1, site="c3_cv3")

    def lin(x, w, b):
        return F.linear(x, q(w, "w_gpt"), b)

    def gpt(p, r, t_):
        b, c, H, W = r.shape
        h, A = 8, 8
        rr = F.adaptive_avg_pool2d(r, (A, A)).reshape(b, c, -1)
        tt = F.adaptive_avg_pool2d(t_, (A, A)).reshape(b, c, -1)
        x = torch.cat([rr, tt], 2).permute(0, 2, 1) + sd[p + "pos_emb"]
        l = 0
        while f"{p}trans_blocks.{l}.ln_input.weight" in sd:
            bp = f"{p}trans_blocks.{l}."
            y = q(F.layer_norm(x, (c,), sd[bp + "ln_input.weight"], sd[bp + "ln_input.bias"], LN_EPS), "gpt_ln")
            dk = c // h
            sp = bp + "sa."
            qq = q(lin(y, sd[sp + "que_proj.weight"], sd[sp + "que_proj.bias"]), "gpt_qkv").view(b, 128, h, dk).permute(0, 2, 1, 3)
            kk = q(lin(y, sd[sp + "key_proj.weight"], sd[sp + "key_proj.bias"]), "gpt_qkv").view(b, 128, h, dk).permute(0, 2, 3, 1)
            vv = q(lin(y, sd[sp + "val_proj.weight"], sd[sp + "val_proj.bias"]), "gpt_qkv").view(b, 128, h, dk).permute(0, 2, 1, 3)
            s = torch.matmul(qq, kk) / dk ** 0.5
            pe = torch.exp(s - s.max(-1, keepdim=True)[0])
            o = torch.matmul(q(pe, "gpt_p"), vv) / pe.sum(-1, keepdim=True)   # the row sum is taken before the rounding of P
            o = q(o.permute(0, 2, 1, 3).reshape(b, 128, c), "gpt_att")
            x = x + lin(o, sd[sp + "out_proj.weight"], sd[sp + "out_proj.bias"])
            y = q(F.layer_norm(x, (c,), sd[bp + "ln_output.weight"], sd[bp + "ln_output.bias"], LN_EPS), "gpt_ln")
            hid = q(F.gelu(lin(y, sd[bp + "mlp.0.weight"], sd[bp + "mlp.0.bias"])), "gpt_hid")
            x = x + lin(hid, sd[bp + "mlp.2.weight"], sd[bp + "mlp.2.bias"])
            l += 1
        x = F.layer_norm(x, (c,), sd[p + "ln_f.weight"], sd[p + "ln_f.bias"], LN_EPS)
        x = x.view(b, 2, A, A, c).permute(0, 1, 4, 2, 3)
        return (F.interpolate(x[:, 0].contiguous(), size=(H, W), mode="bilinear"),
                F.interpolate(x[:, 1].contiguous(), size=(H, W), mode="bilinear"))

    layers, save = build_graph(cfg)
    y = []
    unrounded = {}          # Add2 outputs before their storage rounding (see "Add" below)
    x = rgb
    for L in layers:
        i, f, t = L["i"], L["f"], L["type"]
        cur["i"] = i
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
            x = conv(p + "conv.", q(z, "image"), L["k"], L["s"], site="focus")
        elif t == "C3":
            x = c3(p, xin, L["n"], L["shortcut"])
        elif t == "SPP":
            a = conv(p + "cv1.", xin, 1, 1, site="spp")
            x = conv(p + "cv2.", torch.cat([a] + [F.max_pool2d(a, k, 1, k // 2) for k in L["k"]], 1), 1, 1, site="spp")
        elif t == "Concat":
            x = torch.cat(xin, 1)
        elif t == "Add":
            # The product's CFT output stage (cft_gpt_upsample_add2) forms the Add of two Add2 outputs from their UNROUNDED fp32 sums and
            # rounds once; an Add of anything else sums the stored (rounded) tensors.
            srcs = [i - 1 if j == -1 else j for j in f] if not isinstance(f, int) else []
            if len(srcs) == 2 and all(j in unrounded for j in srcs):
                x = q(unrounded[srcs[0]] + unrounded[srcs[1]], "add")
            else:
                x = q(xin[0] + xin[1], "add")
        elif t == "Add2":
            unrounded[i] = xin[0] + xin[1][L["index"]]
            x = q(unrounded[i], "add2")
        elif t == "GPT":
            x = gpt(p, xin[0], xin[1])
        elif t == "nn.Upsample":
            x = F.interpolate(xin, scale_factor=float(L["scale"]), mode=L["mode"])
        elif t == "Detect":
            sdq = dict(sd)
            for j in range(len(xin)):
                sdq[f"{p}m.{j}.weight"] = q(sd[f"{p}m.{j}.weight"], "w_detect")
            ag = sd[p + "anchor_grid"] if p + "anchor_grid" in sd else sorted_anchors(L["anchors"])[1]
            x = detect(sdq, p, list(xin), L["nc"], ag)
        y.append(x if i in save else None)
    return x


SITES = ("image", "w_conv", "w_gpt", "w_detect", "focus", "conv", "c3_cv12", "bneck_hidden", "bneck_out", "c3_cv3", "spp",
         "add", "add2", "gpt_ln", "gpt_qkv", "gpt_p", "gpt_att", "gpt_hid")


class LowpOracle:
    """``LowpOracle(cfg, torch.bfloat16 | torch.float16)(state_dict, rgb, ir) -> (pred, [raw]*3)``.
    ``sites`` / ``layer_filter``: rounding-site ablation, see ``_forward``."""

    def __init__(self, cfg, dtype, res32=False, sites=None, layer_filter=None, site_dtype=None):
        self.cfg, self.dtype, self.res32 = cfg, dtype, res32
        self.site_dtype = dict(site_dtype or {})
        self.sites = None if sites is None else frozenset(sites)
        self.layer_filter = layer_filter

    @torch.no_grad()
    def __call__(self, sd, rgb, ir):
        sd = {k: v.float() if v.is_floating_point() else v for k, v in sd.items()}
        dt = self.dtype
        site_q = {k: (lambda x, d=d: x.to(d).float()) for k, d in self.site_dtype.items()} or None
        return _forward(self.cfg, sd, rgb.float(), ir.float(), lambda x: x.to(dt).float(), self.res32, self.sites, self.layer_filter, site_q)


class AutocastOracle:
    """The fp32 oracle evaluated under ``torch.autocast("cpu", dtype)`` - the reference's own low-precision forward
    (train.py:755 ``amp.autocast``; the reference ``Model`` under CPU bf16 autocast is what tests/golden/lowp_ref.pt
    records).  ``OracleModel`` issues the same ATen ops in the same order as the reference modules, so the autocast
    policy casts the same tensors: tests/test_oracle_golden.py pins this class to those recorded outputs.  Used where
    the reference tree is not available (GPU box) as the live comparator 'how far is the REFERENCE's own 16-bit
    forward from its fp32 forward on these weights and this shape'."""

    def __init__(self, cfg, dtype=torch.bfloat16):
        from .cft_oracle import OracleModel
        self.model, self.dtype = OracleModel(cfg), dtype

    @torch.no_grad()
    def __call__(self, sd, rgb, ir):
        with torch.autocast("cpu", dtype=self.dtype):
            pred, raw = self.model(sd, rgb, ir)
        return pred.float(), [r.float() for r in raw]
