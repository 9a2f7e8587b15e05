"""Two-stream (RGB + thermal) YOLOv5/CFT model graph, MI355X-native.

Drop-in for the reference's ``models/yolo_test.py`` - which, despite its name, is the two-stream
*model definition* (SURVEY.md D1): ``Detect`` :25-64, ``Model`` :165-327, ``parse_model`` :479-555
(file:line = /root/reference/models/yolo_test.py).  Same public surface:

    Model(cfg, ch=3, nc=None, anchors=None)      cfg = yaml path or dict (reference yamls load unchanged)
    model(x, x2) -> (pred [B, sum(na*ny*nx), nc+5], [raw_i [B,na,ny_i,nx_i,nc+5]] * nl)
    model.fuse(), .stride, .names, .yaml, .save, state-dict keys ``model.<i>....``

What differs is underneath: every layer runs hand-written gfx950 kernels on NHWC activations
(``models/common.py`` here), the compute precision is a model property
(``set_compute_dtype(torch.bfloat16 | torch.float16 | torch.float32)``; ``model.half()`` selects fp16 like it
does in the reference's callers, test.py:66-68; images arrive fp32 / fp16 / uint8 NCHW at the boundary), and
the whole forward can be captured once into a HIP graph (``capture``) so that the ~600 kernel
launches of yolov5l+CFTx3 cost one graph launch.
"""
import logging
import math
from copy import deepcopy
from pathlib import Path

import torch
import torch.nn as nn

from .. import ops
from .common import (GPT, NMS, SPP, Add, Add2, Bottleneck, C3, Concat, Conv, Focus, PendingBilinear, PendingConv, Upsample, _Packed, autoShape, resolve,
                     ACT_NONE, invalidate_packed)

logger = logging.getLogger(__name__)

# yaml module name -> class.  The reference resolves names with eval() inside
# ``from models.common import *`` (models/yolo_test.py:488); the hot-path classes are these.
MODULES = {
    "Conv": Conv, "Bottleneck": Bottleneck, "C3": C3, "SPP": SPP, "Focus": Focus, "Concat": Concat,
    "Add": Add, "Add2": Add2, "GPT": GPT, "nn.Upsample": Upsample, "Upsample": Upsample,
}


def make_divisible(x, divisor):  # reference utils/general.py:210-212
    return math.ceil(x / divisor) * divisor


class Detect(_Packed):
    """Detection head (reference models/yolo_test.py:25-64): per level a 1x1 conv (+bias) on the
    MFMA GEMM (fp32 logits), then one decode kernel writes both the permuted raw logits and the
    decoded rows of the concatenated prediction tensor."""
    stride = None
    export = False

    def __init__(self, nc=80, anchors=(), ch=()):
        super().__init__()
        self.nc = nc
        self.no = nc + 5
        self.nl = len(anchors)
        self.na = len(anchors[0]) // 2
        self.grid = [torch.zeros(1)] * self.nl          # kept for attribute parity; the kernel derives the grid
        a = torch.tensor(anchors).float().view(self.nl, -1, 2)
        self.register_buffer("anchors", a)
        self.register_buffer("anchor_grid", a.clone().view(self.nl, 1, -1, 1, 1, 2))
        self.m = nn.ModuleList(nn.Conv2d(x, self.no * self.na, 1) for x in ch)

    def _pack(self, dtype, device):
        anchors_px = self.anchor_grid.detach().to(device=device, dtype=torch.float32).view(self.nl, -1).contiguous()
        return [ops.pack_conv(m.weight, m.bias, dtype, device=device) for m in self.m], anchors_px

    def forward(self, x):
        if self.export:
            raise RuntimeError("Detect: the ONNX-export branch is outside the hot path")
        x = [resolve(t) for t in x]
        packed, anchors_px = self._packed(x[0].dtype, x[0].device)
        B = x[0].shape[0]
        dev = x[0].device
        rows = [self.na * t.shape[2] * t.shape[3] for t in x]
        pred = torch.empty((B, sum(rows), self.no), dtype=torch.float32, device=dev)
        raws, row0 = [], 0
        for i in range(self.nl):
            logits = ops.conv2d(x[i], packed[i], ACT_NONE, out_dtype=torch.float32)   # [B, pad8(na*no), ny, nx]
            ny, nx = logits.shape[2], logits.shape[3]
            raw = torch.empty((B, self.na, ny, nx, self.no), dtype=torch.float32, device=dev)
            ops.detect_decode(logits, raw, pred, anchors_px[i], self.na, self.no, float(self.stride[i]), row0)
            raws.append(raw)
            row0 += rows[i]
        return raws if self.training else (pred, raws)      # training: only the raw list (reference :59)


def check_anchor_order(m):
    """Flip the anchor levels if their area order disagrees with the stride order
    (reference utils/autoanchor.py:12-20)."""
    a = m.anchor_grid.prod(-1).view(-1)
    da = a[-1] - a[0]
    ds = m.stride[-1] - m.stride[0]
    if da.sign() != ds.sign():
        m.anchors[:] = m.anchors.flip(0)
        m.anchor_grid[:] = m.anchor_grid.flip(0)


def parse_model(d, ch):
    """Model dict -> (nn.Sequential, save list); same rules as reference models/yolo_test.py:479-555:
    depth gain max(round(n*gd),1), width gain make_divisible(c2*gw, 8), Focus forced to 3 input
    channels, C3 gets n as its 3rd argument, GPT width = channels of its first input, and each
    top-level module is tagged with .i/.f/.type/.np."""
    anchors, nc, gd, gw = d["anchors"], d["nc"], d["depth_multiple"], d["width_multiple"]
    na = (len(anchors[0]) // 2) if isinstance(anchors, list) else anchors
    no = na * (nc + 5)
    symbols = {"nc": nc, "anchors": anchors, "None": None, "False": False, "True": True}
    layers, save, c2 = [], [], ch[-1]
    for i, (f, n, m, args) in enumerate(d["backbone"] + d["head"]):
        name = m
        if isinstance(m, str):
            if m not in MODULES and m != "Detect":
                raise NotImplementedError(f"module {m!r} (layer {i}) is outside the CFT hot path; supported: "
                                          f"{sorted(MODULES) + ['Detect']}")
            m = Detect if m == "Detect" else MODULES[m]
        args = [symbols.get(a, a) if isinstance(a, str) else a for a in args]
        n = max(round(n * gd), 1) if n > 1 else n
        if m in (Conv, Bottleneck, SPP, Focus, C3):
            c1 = 3 if m is Focus else ch[f]
            c2 = args[0]
            if c2 != no:
                c2 = make_divisible(c2 * gw, 8)
            args = [c1, c2, *args[1:]]
            if m is C3:
                args.insert(2, n)
                n = 1
        elif m is Concat:
            c2 = sum(ch[x] for x in f)
        elif m is Add:
            c2 = ch[f[0]]
            args = [c2]
        elif m is Add2:
            c2 = ch[f[0]]
            args = [c2, args[1]]
        elif m is GPT:
            c2 = ch[f[0]]
            args = [c2]
        elif m is Detect:
            args.append([ch[x] for x in f])
            if isinstance(args[1], int):
                args[1] = [list(range(args[1] * 2))] * len(f)
        else:
            c2 = ch[f]
        m_ = nn.Sequential(*[m(*args) for _ in range(n)]) if n > 1 else m(*args)
        t = name if isinstance(name, str) else m.__name__
        np_ = sum(x.numel() for x in m_.parameters())
        m_.i, m_.f, m_.type, m_.np = i, f, t, np_
        save.extend(x % i for x in ([f] if isinstance(f, int) else f) if x != -1)
        layers.append(m_)
        if i == 0:
            ch = []
        ch.append(c2)
    return nn.Sequential(*layers), sorted(save)


class Model(nn.Module):
    def __init__(self, cfg="yolov5s.yaml", ch=3, nc=None, anchors=None):
        super().__init__()
        if isinstance(cfg, dict):
            self.yaml = deepcopy(cfg)
        else:
            import yaml
            self.yaml_file = Path(cfg).name
            with open(cfg) as fh:
                self.yaml = yaml.safe_load(fh)
        ch = self.yaml["ch"] = self.yaml.get("ch", ch)
        if nc and nc != self.yaml["nc"]:
            logger.info(f"Overriding model.yaml nc={self.yaml['nc']} with nc={nc}")
            self.yaml["nc"] = nc
        if anchors:
            logger.info(f"Overriding model.yaml anchors with anchors={anchors}")
            self.yaml["anchors"] = round(anchors)
        self.model, self.save = parse_model(deepcopy(self.yaml), ch=[ch])
        self.names = [str(i) for i in range(self.yaml["nc"])]
        m = self.model[-1]
        if isinstance(m, Detect):
            m.stride = torch.Tensor([8.0, 16.0, 32.0])      # hard-coded by the reference (:201)
            m.anchors /= m.stride.view(-1, 1, 1)
            check_anchor_order(m)
            self.stride = m.stride
            self._initialize_biases()
        self._graphs = {}
        self.compute_dtype = torch.bfloat16
        self.set_compute_dtype(torch.bfloat16)   # the Focus modules (where the network's precision is set) mirror it
        self.overlap_streams = True      # run the RGB and the IR backbone on two HIP streams
        self.eval()

    def __setstate__(self, state):
        """Un-pickling a checkpoint written by the reference restores only the reference's attributes."""
        super().__setstate__(state)
        self.__dict__.setdefault("_graphs", {})
        if "compute_dtype" not in self.__dict__:     # a reference pickle: compute in the precision its weights carry
            p = next(self.parameters(), None)       # (checkpoints are saved .half(), train.py:852; attempt_load then .float()s)
            self.__dict__["compute_dtype"] = p.dtype if p is not None and p.dtype in ops.COMPUTE_DTYPES else torch.float32
        self.__dict__.setdefault("overlap_streams", True)
        self.set_compute_dtype(self.compute_dtype)   # Focus.compute_dtype always mirrors Model.compute_dtype

    # ---- weight-change tracking: a captured graph replays the packed buffers of capture time -------
    def invalidate_packed(self):
        """Forget packed weights and captured graphs; call after editing weights through ``.data``."""
        invalidate_packed(self)
        return self

    def load_state_dict(self, *args, **kwargs):
        out = super().load_state_dict(*args, **kwargs)
        invalidate_packed(self)
        return out

    def _apply(self, fn, *args, **kwargs):
        """``.to()/.cuda()/.half()/.float()`` move or cast the parameters: packed copies and graphs are stale."""
        out = super()._apply(fn, *args, **kwargs)
        invalidate_packed(self)
        return out

    def half(self):
        """Reference callers switch to fp16 inference with ``model.half()`` (test.py:66-68, detect_twostream.py:
        40-41): parameters become fp16 as in ``nn.Module.half`` and the kernels compute in fp16 (fp32 accumulate)."""
        super().half()
        return self.set_compute_dtype(torch.float16)

    def float(self):
        """``attempt_load(...).float()``: fp32 parameters; a model put into fp16 by ``half()`` returns to fp32 compute."""
        super().float()
        if self.compute_dtype == torch.float16:
            self.set_compute_dtype(torch.float32)
        return self

    def bfloat16(self):
        super().bfloat16()
        return self.set_compute_dtype(torch.bfloat16)

    # ---- reference-compatible API -----------------------------------------------------------
    def forward(self, x, x2, augment=False, profile=False):
        if augment:
            raise NotImplementedError("augmented inference is broken for two-stream models in the reference "
                                      "(models/yolo_test.py:215-230 calls forward_once with one input)")
        if x.shape != x2.shape or x.dim() != 4 or x.shape[1] != 3:
            raise ValueError(f"expected two [B,3,H,W] image batches of equal shape, got {tuple(x.shape)} and {tuple(x2.shape)}")
        smax = int(self.stride.max()) if hasattr(self, "stride") else 32
        if x.shape[2] % smax or x.shape[3] % smax:
            raise ValueError(f"image height and width must be multiples of the largest stride ({smax}); got "
                             f"{x.shape[2]}x{x.shape[3]} (the reference letterboxes to such sizes, utils/datasets.py:1698-1728)")
        key = (tuple(x.shape), self.compute_dtype, x.dtype)
        g = None if (self.training or profile) else self._graphs.get(key)     # profile: per-layer events need eager launches
        if g is not None:
            if g.weights_key == self.weights_key():
                out = g.replay(x, x2)
                return self.model[-1](out) if type(self.model[-1]) is NMS else out      # the graph ends at Detect (NMS syncs with the host)
            self._graphs.clear()            # weights changed since capture: the graph would replay the old ones
        return self.forward_once(x, x2, profile)

    def weights_key(self):
        """Cheap fingerprint (~0.15 ms) of every parameter/buffer's in-place version counter, compared at replay
        time.  Re-allocations (``.to()``, ``.half()``, ``load_state_dict``, ``fuse()``) go through
        ``invalidate_packed`` instead, which also drops the cached tensor list used here."""
        ws = self.__dict__.get("_wlist")
        if ws is None:
            ws = self.__dict__["_wlist"] = list(self.parameters()) + list(self.buffers())
        return hash(tuple(t._version for t in ws))

    def stream_lanes(self):
        """Lane (HIP stream) of every layer: the IR backbone - everything reachable from an ``f == -4``
        entry through single-input edges, plus ``Add2(index=1)`` whose base input is the IR feature - is
        lane 1; joins (GPT, Add, Concat, Detect) and the RGB backbone/head are lane 0."""
        lanes = []
        for i, m in enumerate(self.model):
            f = m.f
            if f == -4:
                lane = 1
            elif i == 0:
                lane = 0
            elif isinstance(f, int):
                lane = lanes[i + f] if f < 0 else lanes[f]
            elif isinstance(m, Add2):
                j = f[0]
                lane = lanes[i + j] if j < 0 else lanes[j]
            else:
                lane = 0
            lanes.append(lane)
        return lanes

    # ---- concat plan: producers write straight into their slice of the consumer's concat buffer --------------
    def concat_plan(self):
        """{producer layer index: (concat layer index, channel offset, channels, total channels)} for every Concat
        source that is a Conv / C3 / Add (they take ``out=``); such a producer's output tensor then IS a channel
        slice of the concat buffer and ``Concat`` skips its copy.  Upsample sources stay deferred copies."""
        plan = self.__dict__.get("_concat_plan")
        if plan is not None:
            return plan
        layers = list(self.model)
        memo = {}

        def cout(i):
            if i in memo:
                return memo[i]
            m, f = layers[i], layers[i].f
            src = (lambda j: i + j if j < 0 else j)
            if isinstance(m, Focus):
                c = m.conv.conv.out_channels
            elif type(m) is Conv:
                c = m.conv.out_channels
            elif isinstance(m, C3):
                c = m.cv3.conv.out_channels
            elif isinstance(m, SPP):
                c = m.cv2.conv.out_channels
            elif isinstance(m, Concat):
                c = sum(cout(src(j)) for j in f)
            elif isinstance(m, (Add, Add2)):
                c = cout(src(f[0]))
            elif isinstance(m, nn.Upsample) and isinstance(f, int):
                c = cout(src(f))
            else:
                c = None                      # GPT tuples, Detect, Sequentials: never a planned source
            memo[i] = c
            return c

        plan = {}
        try:
            for i, m in enumerate(layers):
                if not isinstance(m, Concat) or isinstance(m.f, int):
                    continue
                srcs = [i + j if j < 0 else j for j in m.f]
                chans = [cout(j) for j in srcs]
                if any(c is None for c in chans):
                    continue
                off = 0
                for j, c in zip(srcs, chans):
                    if (type(layers[j]) is Conv or isinstance(layers[j], (C3, Add))) and j not in plan and layers[j].f != -4:
                        plan[j] = (i, off, c, sum(chans))
                    off += c
        except (IndexError, KeyError, TypeError, AttributeError):   # foreign module graph (bad `from` index, unknown
            plan = {}                                               # module type): no plan, Concat copies as before
        self.__dict__["_concat_plan"] = plan
        return plan

    # ---- CFT output fusion: both Add2 layers behind a GPT block and the Add that sums them run as ONE kernel ----------------
    @property
    def chain_convs(self):
        """Run producer / pointwise-consumer layer pairs as one ``cft_conv2d_chain`` kernel where eligible (default): the Conv handed to
        the C3 behind it (``chain_plan``) and, inside a C3 without shortcuts, ``Bottleneck[j].cv2`` + ``Bottleneck[j+1].cv1``.  ``False``
        runs every layer as its own launch (A/B; bit-identical results)."""
        return self.__dict__.get("_chain_convs", True)

    @chain_convs.setter
    def chain_convs(self, on):
        self.__dict__["_chain_convs"] = bool(on)
        for m in self.modules():
            if isinstance(m, C3):
                m.chain = bool(on)
        self.__dict__.get("_graphs", {}).clear()        # a captured graph replays the launches of capture time

    # Executor switches that change WHICH launches a walk issues: a captured graph replays the launches of capture time, so every
    # setter drops the graphs (an A/B flipped after capture() used to compare a path against itself).
    def _switch(name, default, doc):      # noqa: N805 - class-body helper
        def get(self):
            return self.__dict__.get("_" + name, default)

        def set_(self, v):
            self.__dict__["_" + name] = v
            self.__dict__.get("_graphs", {}).clear()
        return property(get, set_, doc=doc)

    fuse_cft_outputs = _switch("fuse_cft_outputs", True, "Run the two Add2 layers behind a GPT block and the Add that sums them as one kernel (cft_fusion_plan).")
    plan_concats = _switch("plan_concats", True, "Let Conv / C3 / Add layers that feed a head Concat write straight into their slice of its buffer (concat_plan).")
    depth_first = _switch("depth_first", None,
                          "None, or (chunks, rows): run the image-only prefix of each backbone (Focus -> Conv -> C3 -> ..., ``prefix_segments``) "
                          "DEPTH-FIRST over ``chunks`` sub-batches - all of its (at most ``rows``) layers for pairs [0, B/chunks), then for the next "
                          "sub-batch, ... - so that a layer reads what its producer just wrote while those bytes are still in the 256-MiB Infinity "
                          "Cache instead of HBM (at 64 pairs the P1 / P2 tensors are 420 - 840 MB each).  Per-image arithmetic is unchanged: "
                          "bit-identical outputs.  Inference only (BatchNorm batch statistics span the batch).")
    del _switch

    def prefix_segments(self, max_rows=None):
        """[(i0, i1)]: maximal runs of layers that depend on ONE image batch only - a ``Focus`` fed by ``x`` (row 0) or ``x2`` (``f == -4``)
        followed by ``f == -1`` Conv / C3 rows whose outputs have exactly one reader, the next row - cut to ``max_rows`` layers and
        to end on a C3 (a trailing Conv would be a ``PendingConv`` of a C3 outside the segment).  x3 configs: rows 0-4 and 5-9."""
        layers = list(self.model)
        readers = {}
        for j, m in enumerate(layers):
            if m.f == -4 or j == 0:
                continue
            for f in ([m.f] if isinstance(m.f, int) else m.f):
                readers.setdefault(j - 1 if f == -1 else f % j, set()).add(j)
        planned = self.concat_plan()
        segs = []
        for i0, m in enumerate(layers):
            if not isinstance(m, Focus) or not (i0 == 0 or m.f == -4):
                continue
            i1 = i0
            while (i1 + 1 < len(layers) and layers[i1 + 1].f == -1 and type(layers[i1 + 1]) in (Conv, C3) and readers.get(i1) == {i1 + 1}
                   and (i1 + 1) not in planned and (max_rows is None or i1 + 1 - i0 < max_rows)):
                i1 += 1
            while i1 > i0 and type(layers[i1]) is not C3:
                i1 -= 1
            if i1 > i0:
                segs.append((i0, i1))
        return segs

    def _run_segment(self, seg, img, cbufs, chunks):
        """Layers ``seg = (i0, i1)`` depth-first over ``chunks`` sub-batches of the image batch ``img``; returns the last layer's output
        for the whole batch (each sub-batch's C3 writes its batch slice of it)."""
        i0, i1 = seg
        layers = self.model
        B = img.shape[0]
        full = None
        for c in range(chunks):
            b0, b1 = B * c // chunks, B * (c + 1) // chunks
            if b1 == b0:
                continue
            x = layers[i0](img[b0:b1])
            for i in range(i0 + 1, i1):
                x = self._run_layer(layers[i], x, None, cbufs)
            shape = x.shape                                   # (PendingConv knows its output shape without running)
            if full is None:
                c2 = layers[i1].cv3.conv.out_channels
                dt = x.x.dtype if isinstance(x, PendingConv) else x.dtype
                full = ops.new_nhwc(B, shape[2], shape[3], c2, dt, img.device)
            layers[i1](x, out=full[b0:b1])
        return full

    @property
    def splitk(self):
        """Run the CFT blocks' out_proj / fc2 GEMMs as split-K launches where ``ops.splitk_choice`` splits them (default; the LayerNorm that
        follows folds the fp32 partial sums into the residual stream in a fixed order).  ``False``: one launch per GEMM with the residual
        add in its epilogue (A/B; results differ by fp32 summation order only)."""
        return self.__dict__.get("_splitk", True)

    @splitk.setter
    def splitk(self, on):
        from .common import myTransformerBlock
        self.__dict__["_splitk"] = bool(on)
        for m in self.modules():
            if isinstance(m, myTransformerBlock):
                m.splitk = bool(on)
        self.__dict__.get("_graphs", {}).clear()

    def chain_plan(self):
        """Indices of the ``Conv`` layers whose output is read by exactly one layer, the ``C3`` right behind them (``f == -1``): yaml
        rows 1, 3, 6, 8, 13, 15 of the x3 configs (the convs in front of SPP or Concat do not qualify).  Such a conv is handed to its
        C3 un-run (``PendingConv``); the C3 issues both as one kernel when ``ops.conv2d_chain_ok`` accepts the pair (a conv of 128
        or 256 output channels: rows 1, 3, 6, 8 of yolov5l), else it runs the conv itself.  Readers are counted from the ``f`` fields, not from
        ``self.save``: the reference's ``x % i`` book-keeping (models/yolo_test.py:349) files the IR Focus's ``f = -4`` as a reader of
        row 1, which nothing reads."""
        plan = self.__dict__.get("_chain_plan")
        if plan is None:
            layers = list(self.model)
            readers = {}
            for j, m in enumerate(layers):
                if m.f == -4 or j == 0:
                    continue
                for f in ([m.f] if isinstance(m.f, int) else m.f):
                    readers.setdefault(j - 1 if f == -1 else f % j, set()).add(j)
            plan = self.__dict__["_chain_plan"] = frozenset(
                i for i, m in enumerate(layers[:-1])
                if type(m) is Conv and type(layers[i + 1]) is C3 and layers[i + 1].f == -1 and readers.get(i) == {i + 1})
        return plan

    def cft_fusion_plan(self):
        """{index of the first Add2 behind a GPT block: (GPT index, index of the second Add2, index of the Add that consumes both
        or None)} - yaml rows 10-12 + 29 (17-19 + 30, 26-28 + 31) of the x3 configs.  The pattern is matched structurally; a config
        without it (add-fusion, 4-GPT variants with other wiring) simply has no entries."""
        plan = self.__dict__.get("_cft_plan")
        if plan is not None:
            return plan
        layers = list(self.model)
        src = lambda i, j: i + j if j < 0 else j      # noqa: E731
        plan = {}
        for i, m in enumerate(layers):
            if not isinstance(m, GPT) or not isinstance(m.f, (list, tuple)) or len(m.f) != 2:
                continue
            adds = [j for j, a in enumerate(layers) if isinstance(a, Add2) and isinstance(a.f, (list, tuple)) and len(a.f) == 2
                    and src(j, a.f[1]) == i]
            if len(adds) != 2 or {layers[adds[0]].index, layers[adds[1]].index} != {0, 1}:
                continue
            j1, j2 = adds
            gin = [src(i, f) for f in m.f]
            if src(j1, layers[j1].f[0]) != gin[layers[j1].index] or src(j2, layers[j2].f[0]) != gin[layers[j2].index]:
                continue                               # an Add2 whose base is not the GPT's own input of that stream
            k = next((kk for kk, a in enumerate(layers) if type(a) is Add and isinstance(a.f, (list, tuple))
                      and sorted(src(kk, f) for f in a.f) == [j1, j2]), None)
            plan[j1] = (i, j2, k)
        self.__dict__["_cft_plan"] = plan
        return plan

    def _fused_cft_outputs(self, m, x, y, cbufs):
        """Layer ``m`` is the first Add2 of a planned group and its GPT input is still deferred: run the dual de-tokeniser
        (``cft_gpt_upsample_add2``) and return {layer index: output} for the group, else None."""
        ent = self.cft_fusion_plan().get(m.i)
        if ent is None or not isinstance(x, (list, tuple)) or not isinstance(x[1], (list, tuple)):
            return None
        i, j2, k = ent
        p0, p1 = x[1][0], x[1][1]
        if not (isinstance(p0, PendingBilinear) and isinstance(p1, PendingBilinear) and p0.tokens is p1.tokens):
            return None
        if not ops.gpt_dual_tokens_ok(p0.tokens):        # the dual kernel is specialised for 2 x 64 fp32 tokens (8 x 8 anchors)
            return None
        layers = self.model
        src = lambda a, j: a + j if j < 0 else j      # noqa: E731
        base_m = resolve(x[0])
        base_o = y[src(j2, layers[j2].f[0])]
        if base_o is None or not base_m.is_cuda:
            return None
        base_o = resolve(base_o)
        bases = [None, None]
        bases[m.index], bases[layers[j2].index] = base_m, base_o
        sum_out = None
        if k is not None and cbufs is not None:       # the Add's planned concat slice (what _run_layer would hand to Add.forward)
            tgt = self.concat_plan().get(k)
            if tgt is not None:
                cidx, off, c, total = tgt
                B, _, H, W = base_m.shape
                buf = cbufs.get(cidx)
                if buf is None:
                    buf = cbufs[cidx] = ops.new_nhwc(B, H, W, total, base_m.dtype, base_m.device)
                if tuple(buf.shape) == (B, total, H, W):
                    sum_out = buf[:, off:off + c]
        o0, o1, osum = ops.gpt_upsample_add_dual(p0.tokens, bases[0], bases[1], p0.H, p0.W, p0.dtype, sum_out=sum_out, want_sum=k is not None)
        outs = {m.i: (o0, o1)[m.index], j2: (o0, o1)[layers[j2].index]}
        if k is not None:
            outs[k] = osum
        return outs

    def _run_layer(self, m, x, x2, cbufs, chain=True):
        """``chain=False`` (profiling walks): every layer issues its own launches, so the per-layer table charges a Conv's time to the
        Conv and not to the C3 that would otherwise run it."""
        if m.f == -4 or (m.i == 0 and isinstance(m, Focus)):
            img = x2 if m.f == -4 else x
            return m(img)
        tgt = self.concat_plan().get(m.i) if cbufs is not None else None
        if tgt is not None:
            cidx, off, c, total = tgt
            t = x[0] if isinstance(x, (list, tuple)) else x
            shape = t.shape if isinstance(t, PendingConv) else None     # (stays un-run: only its geometry is needed here)
            t = t.x if isinstance(t, PendingConv) else resolve(t)
            if isinstance(t, torch.Tensor) and t.is_cuda and t.dim() == 4:
                B, _, H, W = shape or t.shape
                if type(m) is Conv:
                    k, s_ = m.conv.kernel_size[0], m.conv.stride[0]
                    H, W = (H + 2 * (k // 2) - k) // s_ + 1, (W + 2 * (k // 2) - k) // s_ + 1
                buf = cbufs.get(cidx)
                if buf is None:
                    buf = cbufs[cidx] = ops.new_nhwc(B, H, W, total, t.dtype, t.device)
                if tuple(buf.shape) == (B, total, H, W):
                    return m(x, out=buf[:, off:off + c])
        if cbufs is not None and isinstance(m, Concat) and m.i in cbufs:
            return m(x, out=cbufs[m.i])
        if (chain and cbufs is not None and not self.training and m.i in self.chain_plan() and self.chain_convs
                and isinstance(x, torch.Tensor) and x.dtype in (torch.bfloat16, torch.float16)):
            return PendingConv(m, x)                 # left to the C3 behind it (one kernel for the conv and the C3's cv1|cv2)
        return m(x)

    def forward_once(self, x, x2, profile=False, until_detect=False):
        """Graph walk of reference models/yolo_test.py:235-272: ``f == -1`` previous output, int /
        list = saved outputs, ``f == -4`` = this layer consumes the IR image ``x2``.
        ``until_detect``: stop in front of a trailing ``NMS`` module (HIP-graph capture: NMS reads its counts on the host).

        The two backbones are independent between fusion points, so with ``overlap_streams`` they are
        enqueued on two HIP streams (fork/join by stream waits; inside a HIP-graph capture this becomes
        two parallel branches of the graph).  Every tensor that crosses lanes is a saved layer output
        and stays referenced in ``y`` until the walk ends, so the caching allocator cannot recycle it
        under a kernel of the other stream."""
        layers = list(self.model)
        if until_detect and layers and type(layers[-1]) is NMS:
            layers = layers[:-1]
        lanes = self.stream_lanes() if (self.overlap_streams and x.is_cuda and not profile) else None
        fused = {}                                                    # outputs of a CFT output group still to be handed to their layers
        fuse_cft = x.is_cuda and self.fuse_cft_outputs and not profile      # profile: one launch group per layer (the reference's per-layer table)
        cbufs = {} if (x.is_cuda and self.plan_concats) else None   # planned concat buffers of this walk
        # depth-first prefix (Model.depth_first): {first row: (i0, i1)} and the rows a segment covers
        seg_at, seg_rows, df = {}, set(), self.depth_first
        if df and x.is_cuda and not self.training and not profile and cbufs is not None and x.shape[0] >= 2 * int(df[0]) > 0:
            for sg in self.prefix_segments(df[1] if len(df) > 1 else None):
                seg_at[sg[0]] = sg
                seg_rows.update(range(sg[0], sg[1] + 1))
        if lanes is None or 1 not in lanes:
            y, marks = [], []
            if profile:     # reference :252-260,270-271: per-layer time / GFLOPS / params / type.  One stream, HIP events around
                flog, prev_log = [], ops._launch_log         # every layer, GFLOPS from the algorithmic FLOPs of its GEMM launches
                ops.set_launch_log(flog)
            try:
                for m in layers:
                    if m.i in seg_rows:                       # depth-first prefix: the whole segment runs at its first row
                        if m.i in seg_at:
                            x = self._run_segment(seg_at[m.i], x2 if m.f == -4 else x, cbufs, int(df[0]))
                            seg_end = seg_at[m.i][1]
                        y.append(x if (m.i == seg_end and m.i in self.save) else None)
                        continue
                    if m.f != -1 and m.f != -4:
                        x = y[m.f] if isinstance(m.f, int) else [x if j == -1 else y[j] for j in m.f]
                    if profile:
                        e0, e1, n0 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True), len(flog)
                        e0.record()
                    if m.i in fused:
                        x = fused.pop(m.i)
                    else:
                        grp = self._fused_cft_outputs(m, x, y, cbufs) if fuse_cft else None
                        if grp is not None:
                            x = grp.pop(m.i)
                            fused.update(grp)
                        else:
                            x = self._run_layer(m, x, x2, cbufs, chain=not profile)
                    if profile:
                        e1.record()
                        marks.append((m, e0, e1, sum(rec[1] for rec in flog[n0:])))
                    y.append(x if m.i in self.save else None)
            finally:
                if profile:
                    ops.set_launch_log(prev_log)
            if profile:
                torch.cuda.synchronize(x[0].device if isinstance(x, (tuple, list)) else x.device)
                self.profile_ms = [(m.i, m.type, e0.elapsed_time(e1), fl / 1e9, m.np) for m, e0, e1, fl in marks]
                logger.info(f"{'time (ms)':>10s} {'GFLOPS':>10s} {'params':>10s}  {'module'}")
                for i, t, ms, gf, np_ in self.profile_ms:
                    logger.info(f"{ms:10.2f} {gf:10.2f} {np_:10.0f}  {t}")
                logger.info('%.1fms total' % sum(r[2] for r in self.profile_ms))
            return x
        main = torch.cuda.current_stream(x.device)
        side = self.__dict__.get("_side_stream")
        if side is None or side.device != x.device:
            side = self.__dict__["_side_stream"] = torch.cuda.Stream(device=x.device)
        streams = (main, side)
        side.wait_stream(main)                       # fork: the side lane starts after everything queued so far
        y, keep = [], []
        for i, m in enumerate(layers):
            lane = lanes[i]
            f = m.f
            if m.i in seg_rows:                               # depth-first prefix: the whole segment runs at its first row, on its lane
                if m.i in seg_at:
                    with torch.cuda.stream(streams[lane]):
                        x = self._run_segment(seg_at[m.i], x2 if f == -4 else x, cbufs, int(df[0]))
                    seg_end = seg_at[m.i][1]
                    keep.append(x)
                y.append(x if (m.i == seg_end and m.i in self.save) else None)
                continue
            srcs = [] if (f == -4 or i == 0) else ([i - 1] if f == -1 else ([f % i] if isinstance(f, int) else [(i - 1 if j == -1 else j % i) for j in f]))
            if any(lanes[j] != lane for j in srcs):
                streams[lane].wait_stream(streams[1 - lane])
            if f != -1 and f != -4:
                x = y[f] if isinstance(f, int) else [x if j == -1 else y[j] for j in f]
            with torch.cuda.stream(streams[lane]):
                if m.i in fused:
                    x = fused.pop(m.i)       # produced by the group's kernel on the other lane: the cross-lane wait above orders it
                else:
                    grp = self._fused_cft_outputs(m, x, y, cbufs) if fuse_cft else None
                    if grp is not None:
                        x = grp.pop(m.i)
                        fused.update(grp)
                    else:
                        x = self._run_layer(m, x, x2, cbufs)
            keep.append(x)                           # keep every output alive until both lanes have joined
            y.append(x if m.i in self.save else None)
        main.wait_stream(side)                       # join
        return x

    def _initialize_biases(self, cf=None):  # reference :274-282
        m = self.model[-1]
        for mi, s in zip(m.m, m.stride):
            b = mi.bias.view(m.na, -1)
            b.data[:, 4] += math.log(8 / (640 / s) ** 2)
            b.data[:, 5:] += math.log(0.6 / (m.nc - 0.99)) if cf is None else torch.log(cf / cf.sum())
            mi.bias = torch.nn.Parameter(b.view(-1), requires_grad=True)

    def _print_biases(self):  # reference :284-289
        m = self.model[-1] if isinstance(self.model[-1], Detect) else self.model[-2]
        for mi in m.m:
            b = mi.bias.detach().view(m.na, -1).T
            logger.info(("%6g Conv2d.bias:" + "%10.3g" * 6) % (mi.weight.shape[1], *b[:5].mean(1).tolist(), b[5:].mean()))

    def nms(self, mode=True):
        """Add or remove the NMS module behind ``Detect`` (reference :306-318): with it ``model(x, x2)`` returns the per-image
        detection lists.  The module runs ``cft_nms``; a captured HIP graph covers the layers in front of it."""
        present = type(self.model[-1]) is NMS
        if mode and not present:
            logger.info("Adding NMS... ")
            m = NMS()
            m.f = -1
            m.i = self.model[-1].i + 1
            m.type, m.np = "models.common.NMS", 0
            self.model.add_module(name="%s" % m.i, module=m)
            self.eval()
        elif not mode and present:
            logger.info("Removing NMS... ")
            self.model = self.model[:-1]
        self.__dict__.pop("_concat_plan", None)
        self.__dict__.pop("_cft_plan", None)
        self.__dict__.pop("_chain_plan", None)
        self._graphs.clear()
        return self

    def autoshape(self):
        """Wrap the model for raw image input (reference :320-324); see ``models.common.autoShape`` (two-stream form)."""
        logger.info("Adding autoShape... ")
        return autoShape(self)

    def fuse(self):
        """Fold every BatchNorm into its conv (reference :296-304, utils/torch_utils.py:181-201).
        The kernels always execute the folded form, so this only changes the stored parameters
        (``conv.weight``/``conv.bias``, no ``bn``) exactly like the reference does."""
        for m in self.model.modules():
            if type(m) is Conv and hasattr(m, "bn"):
                w, b = ops.fold_bn(m.conv.weight, m.bn.weight, m.bn.bias, m.bn.running_mean, m.bn.running_var, m.bn.eps)
                c = m.conv
                fused = nn.Conv2d(c.in_channels, c.out_channels, c.kernel_size, c.stride, c.padding, bias=True)
                fused = fused.requires_grad_(False).to(c.weight.device)
                fused.weight.copy_(w.detach())
                fused.bias.copy_(b.detach())
                m.conv = fused
                delattr(m, "bn")
        invalidate_packed(self)
        return self

    def info(self, verbose=False, img_size=640):
        n_p = sum(x.numel() for x in self.parameters())
        logger.info(f"Model Summary: {len(list(self.modules()))} layers, {n_p} parameters")

    # ---- MI355X-specific ---------------------------------------------------------------------
    def set_compute_dtype(self, dtype):
        """bf16 (default) or fp16 - both with fp32 accumulation, fp32 CFT residual stream and fp32 logits; fp16 is
        the reference's own GPU precision and the 16-bit type that meets the 1e-2 parity bound - or
        fp32 (exact-fp32 MFMA path used for the 1e-3 tolerance configuration)."""
        if dtype not in ops.COMPUTE_DTYPES:
            raise TypeError("compute dtype must be torch.bfloat16, torch.float16 or torch.float32")
        if dtype != self.compute_dtype:
            self._graphs.clear()
        self.compute_dtype = dtype
        for m in self.modules():
            if isinstance(m, Focus):
                m.compute_dtype = dtype
        return self

    def prepare(self):
        """Pack every layer's weights for the current device/dtype now (otherwise done lazily)."""
        dev = next(self.parameters()).device
        for m in self.modules():
            if isinstance(m, _Packed):
                m._packed(self.compute_dtype, dev)
        return self

    def capture(self, batch, height, width, input_dtype=torch.float32):
        """Record the forward for a fixed input shape into a HIP graph; later ``model(x, x2)`` calls
        with that shape replay it (inputs are copied into static buffers, outputs are static
        tensors that the next replay overwrites)."""
        from ..graph import CapturedForward
        key = ((batch, 3, height, width), self.compute_dtype, input_dtype)
        self._graphs[key] = CapturedForward(self, batch, height, width, input_dtype=input_dtype)
        return self._graphs[key]

    def release_graphs(self):
        self._graphs.clear()
