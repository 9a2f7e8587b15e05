"""CPU oracle for the two-stream YOLOv5 + CFT inference forward.

TEST INFRASTRUCTURE ONLY.  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import this file; the product package never does.

What it is: a functional, state-dict-driven restatement (plain torch fp32 on CPU, no
``nn.Module``) of the reference hot path, written from the reference's behaviour:

* graph build / routing ........ /root/reference models/yolo_test.py:479-555 (parse_model),
                                 :235-272 (forward_once; ``from == -4`` feeds the IR image)
* Conv (+BN eval, SiLU) ........ models/common.py:36-50, BN eps 1e-3 utils/torch_utils.py:144-153
* Bottleneck / C3 / SPP / Focus  models/common.py:99-109, :131-143, :154-165, :168-179
* Concat / Add / Add2 .......... models/common.py:211-243
* GPT (CFT block) .............. models/common.py:430-639
* Detect decode ................ models/yolo_test.py:25-64, stride [8,16,32] :201

All arithmetic is ATen (torch 2.10 CPU here), which is also what the reference itself
dispatches to (reference requirements.txt:10 pins only torch>=1.7).

Pinning: the reference ships no tests or golden vectors, so the pins
are outputs of the reference's own ``models.yolo_test.Model`` executed in the build
container: ``tests/golden/make_golden.py`` (committed) writes them to ``tests/golden/*.pt``
and ``tests/test_oracle_golden.py`` checks this oracle against them.
"""
import math

import torch
import torch.nn.functional as F

BN_EPS = 1e-3  # utils/torch_utils.py:150
BN_MOMENTUM = 0.03  # utils/torch_utils.py:151
LN_EPS = 1e-5  # nn.LayerNorm default, models/common.py:529-530,572
STRIDES = (8.0, 16.0, 32.0)  # models/yolo_test.py:201

# Training-mode forward (model.train(); models/common.py:45-47 with bn.training, models/yolo_test.py:50,59): while an
# OracleModel call with train=True is running this dict collects the updated BatchNorm running statistics
# {state-dict key: tensor}; BatchNorm then normalises with BATCH statistics.  Dropout (models/common.py:507,511,537,611)
# draws from torch's RNG and is only restated for p = 0 (identity) - the tests set every Dropout.p to 0 for parity.
_TRAIN = None


def make_divisible(x, divisor):  # utils/general.py:210-212
    return math.ceil(x / divisor) * divisor


# ----------------------------------------------------------------------------- graph
_CONV_LIKE = ("Conv", "Focus", "SPP", "C3", "Bottleneck")


def build_graph(cfg, ch=3):
    """Turn a model dict (yaml.safe_load of a reference ``models/transformer/*.yaml`` or an
    equivalent dict) into a flat list of layer records, following parse_model
    (models/yolo_test.py:479-555): depth gain ``max(round(n*gd),1)``, width gain
    ``make_divisible(c2*gw, 8)``, Focus forced to 3 input channels, GPT width = ch[f[0]]."""
    anchors, nc = cfg["anchors"], cfg["nc"]
    gd, gw = cfg["depth_multiple"], cfg["width_multiple"]
    na = len(anchors[0]) // 2
    no = na * (nc + 5)
    layers, save, chans = [], [], [ch]
    c2 = ch
    for i, (f, n, m, args) in enumerate(cfg["backbone"] + cfg["head"]):
        args = list(args)
        for j, a in enumerate(args):
            if isinstance(a, str):
                args[j] = {"nc": nc, "anchors": anchors, "None": None, "False": False, "True": True}.get(a, a)
        n = max(round(n * gd), 1) if n > 1 else n
        rec = {"i": i, "f": f, "type": m, "n": 1}
        if m in _CONV_LIKE:
            c1 = 3 if m == "Focus" else chans[f]
            c2 = args[0]
            if c2 != no:
                c2 = make_divisible(c2 * gw, 8)
            rest = args[1:]
            if m == "C3":
                rec.update(c1=c1, c2=c2, n=n, shortcut=(rest[0] if rest else True))
            elif m == "SPP":
                rec.update(c1=c1, c2=c2, k=tuple(rest[0]) if rest else (5, 9, 13))
            else:  # Conv / Focus: k, s
                k = rest[0] if len(rest) > 0 else 1
                s = rest[1] if len(rest) > 1 else 1
                rec.update(c1=c1, c2=c2, k=k, s=s)
        elif m == "Concat":
            c2 = sum(chans[x] for x in f)
        elif m in ("Add", "Add2", "GPT"):
            c2 = chans[f[0]]
            if m == "Add2":
                rec["index"] = args[1]
            if m == "GPT":
                rec["d_model"] = c2
        elif m == "nn.Upsample":
            rec.update(scale=args[1], mode=args[2])
            c2 = chans[f]
        elif m == "Detect":
            rec.update(nc=args[0], anchors=args[1], ch=[chans[x] for x in f])
        else:
            raise ValueError(f"oracle: unsupported module {m!r}")
        save.extend(x % i for x in ([f] if isinstance(f, int) else f) if x != -1)
        layers.append(rec)
        if i == 0:
            chans = []
        chans.append(c2)
    return layers, sorted(set(save))


# ----------------------------------------------------------------------------- ops
def conv_bn_silu(sd, p, x, k, s, act=True):
    """Conv.forward in eval mode: SiLU(BN(conv2d(x))), conv bias-free, pad k//2
    (models/common.py:36-47).  If the state dict holds a fused conv (``conv.bias`` present and
    no ``bn.*``; utils/torch_utils.py:181-201) the fused form :49-50 is evaluated."""
    w = sd[p + "conv.weight"]
    if p + "bn.weight" in sd:
        y = F.conv2d(x, w, None, s, k // 2)
        if _TRAIN is not None:      # bn.training: batch statistics, running statistics updated (momentum 0.03)
            rm, rv = sd[p + "bn.running_mean"].clone(), sd[p + "bn.running_var"].clone()
            y = F.batch_norm(y, rm, rv, sd[p + "bn.weight"], sd[p + "bn.bias"], True, BN_MOMENTUM, BN_EPS)
            _TRAIN[p + "bn.running_mean"], _TRAIN[p + "bn.running_var"] = rm, rv
        else:
            y = F.batch_norm(y, sd[p + "bn.running_mean"], sd[p + "bn.running_var"],
                             sd[p + "bn.weight"], sd[p + "bn.bias"], False, 0.0, BN_EPS)
    else:
        y = F.conv2d(x, w, sd[p + "conv.bias"], s, k // 2)
    return F.silu(y) if act else y


def bottleneck(sd, p, x, shortcut):  # models/common.py:99-109 (e=1.0 inside C3, c1==c2)
    y = conv_bn_silu(sd, p + "cv2.", conv_bn_silu(sd, p + "cv1.", x, 1, 1), 3, 1)
    return x + y if shortcut else y


def c3(sd, p, x, n, shortcut):  # models/common.py:131-143
    a = conv_bn_silu(sd, p + "cv1.", x, 1, 1)
    for j in range(n):
        a = bottleneck(sd, f"{p}m.{j}.", a, shortcut)
    b = conv_bn_silu(sd, p + "cv2.", x, 1, 1)
    return conv_bn_silu(sd, p + "cv3.", torch.cat((a, b), 1), 1, 1)


def spp(sd, p, x, ks):  # models/common.py:154-165
    x = conv_bn_silu(sd, p + "cv1.", x, 1, 1)
    pools = [F.max_pool2d(x, k, 1, k // 2) for k in ks]
    return conv_bn_silu(sd, p + "cv2.", torch.cat([x] + pools, 1), 1, 1)


def focus(sd, p, x, k, s):  # models/common.py:168-179
    z = torch.cat([x[..., ::2, ::2], x[..., 1::2, ::2], x[..., ::2, 1::2], x[..., 1::2, 1::2]], 1)
    return conv_bn_silu(sd, p + "conv.", z, k, s)


def self_attention(sd, p, x, h):  # models/common.py:475-513 (dropouts are identity in eval)
    b, t, d = x.shape
    dk = d // h
    q = F.linear(x, sd[p + "que_proj.weight"], sd[p + "que_proj.bias"]).view(b, t, h, dk).permute(0, 2, 1, 3)
    k = F.linear(x, sd[p + "key_proj.weight"], sd[p + "key_proj.bias"]).view(b, t, h, dk).permute(0, 2, 3, 1)
    v = F.linear(x, sd[p + "val_proj.weight"], sd[p + "val_proj.bias"]).view(b, t, h, dk).permute(0, 2, 1, 3)
    att = torch.softmax(torch.matmul(q, k) / math.sqrt(dk), -1)
    out = torch.matmul(att, v).permute(0, 2, 1, 3).reshape(b, t, d)
    return F.linear(out, sd[p + "out_proj.weight"], sd[p + "out_proj.bias"])


def transformer_block(sd, p, x, h):  # models/common.py:516-546, pre-LN, exact-erf GELU
    d = x.shape[-1]
    y = F.layer_norm(x, (d,), sd[p + "ln_input.weight"], sd[p + "ln_input.bias"], LN_EPS)
    x = x + self_attention(sd, p + "sa.", y, h)
    y = F.layer_norm(x, (d,), sd[p + "ln_output.weight"], sd[p + "ln_output.bias"], LN_EPS)
    y = F.linear(y, sd[p + "mlp.0.weight"], sd[p + "mlp.0.bias"])
    y = F.linear(F.gelu(y), sd[p + "mlp.2.weight"], sd[p + "mlp.2.bias"])
    return x + y


def gpt_tokens(sd, p, rgb, ir, h=8, anchors=8):
    """GPT.forward up to and including ln_f (models/common.py:593-625): 8x8 adaptive average
    pool of each stream, tokens ordered RGB (64) then IR (64), + pos_emb, n_layer blocks."""
    b, c, _, _ = rgb.shape
    r = F.adaptive_avg_pool2d(rgb, (anchors, anchors)).reshape(b, c, -1)
    t = F.adaptive_avg_pool2d(ir, (anchors, anchors)).reshape(b, c, -1)
    x = torch.cat([r, t], 2).permute(0, 2, 1) + sd[p + "pos_emb"]
    n_layer = 0
    while f"{p}trans_blocks.{n_layer}.ln_input.weight" in sd:
        n_layer += 1
    for l in range(n_layer):
        x = transformer_block(sd, f"{p}trans_blocks.{l}.", x, h)
    return F.layer_norm(x, (c,), sd[p + "ln_f.weight"], sd[p + "ln_f.bias"], LN_EPS)


def gpt(sd, p, rgb, ir, h=8, anchors=8):  # models/common.py:593-639
    b, c, H, W = rgb.shape
    x = gpt_tokens(sd, p, rgb, ir, h, anchors)
    x = x.view(b, 2, anchors, anchors, c).permute(0, 1, 4, 2, 3)
    r = F.interpolate(x[:, 0].contiguous(), size=(H, W), mode="bilinear")  # align_corners=False
    t = F.interpolate(x[:, 1].contiguous(), size=(H, W), mode="bilinear")
    return r, t


def sorted_anchors(anchors):
    """Anchors in stride units, order-checked (models/yolo_test.py:203-204,
    utils/autoanchor.py:12-20): flip the per-level order if area order disagrees with stride order."""
    a = torch.tensor(anchors, dtype=torch.float32).view(len(anchors), -1, 2)
    ag = a.clone().view(len(anchors), 1, -1, 1, 1, 2)
    a = a / torch.tensor(STRIDES).view(-1, 1, 1)
    area = ag.prod(-1).view(-1)
    if torch.sign(area[-1] - area[0]) != torch.sign(torch.tensor(STRIDES[-1] - STRIDES[0])):
        a, ag = a.flip(0), ag.flip(0)
    return a, ag


def detect(sd, p, xs, nc, anchor_grid):
    """Detect.forward, inference branch (models/yolo_test.py:41-59).  Returns
    (pred [B, sum(na*ny*nx), no], [raw_i [B,na,ny,nx,no]])."""
    no = nc + 5
    z, raws = [], []
    for i, x in enumerate(xs):
        y = F.conv2d(x, sd[f"{p}m.{i}.weight"], sd[f"{p}m.{i}.bias"])
        b, _, ny, nx = y.shape
        na = y.shape[1] // no
        y = y.view(b, na, no, ny, nx).permute(0, 1, 3, 4, 2).contiguous()
        raws.append(y)
        yv, xv = torch.meshgrid(torch.arange(ny), torch.arange(nx), indexing="ij")
        grid = torch.stack((xv, yv), 2).view(1, 1, ny, nx, 2).float()
        s = y.sigmoid()
        xy = (s[..., 0:2] * 2.0 - 0.5 + grid) * STRIDES[i]
        wh = (s[..., 2:4] * 2.0) ** 2 * anchor_grid[i]
        z.append(torch.cat((xy, wh, s[..., 4:]), -1).view(b, -1, no))
    return torch.cat(z, 1), raws


# ----------------------------------------------------------------------------- model
class OracleModel:
    """``OracleModel(cfg)(state_dict, rgb, ir) -> (pred, [raw]*3)``; ``taps`` (optional dict)
    receives every saved layer output for bisecting."""

    def __init__(self, cfg, ch=3):
        self.cfg = cfg
        self.layers, self.save = build_graph(cfg, ch)

    @torch.no_grad()
    def __call__(self, sd, rgb, ir, taps=None, tap_all=False, train=False):
        """train=True: the training-mode forward (BatchNorm batch statistics, Detect returns only the raw list,
        models/yolo_test.py:59); returns (raw list, {updated running statistics})."""
        global _TRAIN
        if train:
            _TRAIN = {}
            try:
                _, raws = self.__call__(sd, rgb, ir, taps, tap_all)
                return raws, _TRAIN
            finally:
                _TRAIN = None
        sd = {k: v.float() if v.is_floating_point() else v for k, v in sd.items()}
        y = []
        x = rgb.float()
        ir = ir.float()
        for L in self.layers:
            i, f, t = L["i"], L["f"], L["type"]
            p = f"model.{i}."
            if f == -4:           # second stream enters here (models/yolo_test.py:262-263)
                xin = ir
            elif f == -1:
                xin = x
            elif isinstance(f, int):
                xin = y[f]
            else:
                xin = [x if j == -1 else y[j] for j in f]
            if t == "Conv":
                x = conv_bn_silu(sd, p, xin, L["k"], L["s"])
            elif t == "Focus":
                x = focus(sd, p, xin, L["k"], L["s"])
            elif t == "C3":
                x = c3(sd, p, xin, L["n"], L["shortcut"])
            elif t == "SPP":
                x = spp(sd, p, xin, L["k"])
            elif t == "Concat":
                x = torch.cat(xin, 1)
            elif t == "Add":
                x = xin[0] + xin[1]
            elif t == "Add2":
                x = xin[0] + xin[1][L["index"]]
            elif t == "GPT":
                x = gpt(sd, p, xin[0], xin[1])
            elif t == "nn.Upsample":
                x = F.interpolate(xin, scale_factor=float(L["scale"]), mode=L["mode"])
            elif t == "Detect":
                ag = sd[p + "anchor_grid"] if p + "anchor_grid" in sd else sorted_anchors(L["anchors"])[1]
                x = detect(sd, p, list(xin), L["nc"], ag)
            y.append(x if (i in self.save or tap_all) else None)
            if taps is not None and (i in self.save or tap_all) and t != "Detect":
                taps[i] = x
        return x


def algorithmic_flops(cfg, height, width, ch=3):
    """Contraction FLOPs per image pair: convs 2*Cout*Ho*Wo*k*k*Cin,
    GPT linears 2*T*in*out, attention 4*T*T*d per layer; T=128, n_layer=8, block_exp=4."""
    layers, _ = build_graph(cfg, ch)
    shapes = {}
    total = {"conv": 0.0, "gpt": 0.0}
    hw = None
    prev = (height, width)
    for L in layers:
        i, f, t = L["i"], L["f"], L["type"]
        if f == -4:
            hin = (height, width)
        elif f == -1:
            hin = prev
        elif isinstance(f, int):
            hin = shapes[f % i]
        else:
            hin = prev if f[0] == -1 else shapes[f[0] % i]
        hout = hin

        def conv(c1, c2, k, s, h_w):
            ho, wo = (h_w[0] + 2 * (k // 2) - k) // s + 1, (h_w[1] + 2 * (k // 2) - k) // s + 1
            total["conv"] += 2.0 * c2 * ho * wo * k * k * c1
            return ho, wo
        if t == "Conv":
            hout = conv(L["c1"], L["c2"], L["k"], L["s"], hin)
        elif t == "Focus":
            hout = conv(12, L["c2"], L["k"], L["s"], (hin[0] // 2, hin[1] // 2))
        elif t == "C3":
            c_ = L["c2"] // 2
            conv(L["c1"], c_, 1, 1, hin); conv(L["c1"], c_, 1, 1, hin); conv(2 * c_, L["c2"], 1, 1, hin)
            for _ in range(L["n"]):
                conv(c_, c_, 1, 1, hin); conv(c_, c_, 3, 1, hin)
        elif t == "SPP":
            c_ = L["c1"] // 2
            conv(L["c1"], c_, 1, 1, hin); conv(c_ * (len(L["k"]) + 1), L["c2"], 1, 1, hin)
        elif t == "GPT":
            d, T = L["d_model"], 128
            total["gpt"] += 8 * (2.0 * T * d * d * 4 + 2.0 * T * d * 4 * d * 2 + 4.0 * T * T * d)
        elif t == "nn.Upsample":
            hout = (hin[0] * L["scale"], hin[1] * L["scale"])
        elif t == "Detect":
            no = (L["nc"] + 5) * (len(L["anchors"][0]) // 2)
            for c, j in zip(L["ch"], f):
                conv(c, no, 1, 1, shapes[j])
        shapes[i] = hout
        prev = hout
    total["total"] = total["conv"] + total["gpt"]
    return total
