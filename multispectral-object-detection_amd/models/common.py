"""Module library of the two-stream YOLOv5 + CFT detector, MI355X-native.

Drop-in for the hot-path classes of the reference's ``models/common.py`` (same class names,
constructor signatures, sub-module/parameter names and therefore state-dict keys):

    Conv :36-50   Bottleneck :99-109   C3 :131-143   SPP :154-165   Focus :168-179
    Concat :211-219   Add :222-229   Add2 :232-243
    SelfAttention :430-513   myTransformerBlock :516-546   GPT :549-639

(``file:line`` = /root/reference/models/common.py).  The parameters are ordinary fp32
``nn.Parameter``s, so ``load_state_dict`` from a reference model is loss-free; the forward of
every class, however, launches hand-written gfx950 kernels through ``libcft_hip.so`` (see
``include/cft_hip.h``) on NHWC tensors instead of calling ATen:

* Conv        one implicit-GEMM kernel with BatchNorm folded in and bias+SiLU fused;
* Bottleneck  the residual add is the epilogue of its 3x3 conv;
* C3          cv1|cv2 run as ONE GEMM into the concat buffer, the bottleneck chain ends in the
              same buffer, so ``torch.cat`` disappears;
* SPP         cv1 writes slice 0 of the 4-way concat buffer, one kernel fills the 3 pooled slices;
* GPT         fp32 residual stream, LayerNorm -> fused QKV GEMM -> single-tile MFMA attention ->
              out-proj(+residual) -> LayerNorm -> fc1(+GELU) -> fc2(+residual); the bilinear
              upsampling is deferred and fused with the Add2 that consumes it.

``model.eval()`` (the default, the benchmarked path): BatchNorm folded into the conv weights, dropout the identity.
``model.train()``: the training-mode FORWARD of the reference: BatchNorm with batch
statistics and running-stat updates (``cft_batchnorm_train`` after an un-folded conv), dropout in GPT / SelfAttention /
MLP (``cft_dropout``, counter-based masks), ``Detect`` returning the raw list.  No autograd graph is built.
"""
import math

import torch
import torch.nn as nn

from .. import ops
from ..ops import ACT_GELU, ACT_NONE, ACT_SILU

BN_EPS = 1e-3  # reference utils/torch_utils.py:150 (initialize_weights)


def autopad(k, p=None):  # reference models/common.py:24-28
    if p is None:
        p = k // 2 if isinstance(k, int) else [x // 2 for x in k]
    return p


# ------------------------------------------------------------------------------ lazy tensors
class _Pending:
    """A value whose producing kernel is deferred so that the consumer can fuse it."""

    def materialize(self):
        raise NotImplementedError


class PendingBilinear(_Pending):
    """GPT output stream ``s``: bilinear(tokens 8x8 -> HxW).  ``Add2`` fuses it with its add."""

    def __init__(self, tokens, s, H, W, dtype):
        self.tokens, self.s, self.H, self.W, self.dtype = tokens, s, H, W, dtype

    def materialize(self, base=None):
        return ops.gpt_upsample_add(self.tokens, self.s, base, self.H, self.W, self.dtype)

    @property
    def shape(self):
        return torch.Size((self.tokens.shape[0], self.tokens.shape[2], self.H, self.W))


class PendingUpsample(_Pending):
    """nearest 2^up upsampling of ``x``; ``Concat`` writes it straight into its buffer."""

    def __init__(self, x, up):
        self.x, self.up = x, up

    def materialize(self):
        B, C, H, W = self.x.shape
        out = ops.new_nhwc(B, H << self.up, W << self.up, C, self.x.dtype, self.x.device)
        return ops.copy_channels(self.x, out, self.up)

    @property
    def shape(self):
        B, C, H, W = self.x.shape
        return torch.Size((B, C, H << self.up, W << self.up))


class PendingConv(_Pending):
    """Output of a ``Conv`` whose only consumer is the ``C3`` behind it (``Model.chain_plan``): that C3 runs the conv and its own
    packed cv1|cv2 as ONE kernel (``ops.conv2d_chain``) when the pair is eligible, and the conv's output tensor never exists."""

    def __init__(self, conv, x):
        self.conv, self.x = conv, x

    def materialize(self):
        return self.conv(self.x)

    @property
    def shape(self):
        B, _, H, W = self.x.shape
        k, s = self.conv.conv.kernel_size[0], self.conv.conv.stride[0]
        return torch.Size((B, self.conv.conv.out_channels, (H + 2 * (k // 2) - k) // s + 1, (W + 2 * (k // 2) - k) // s + 1))


def resolve(x):
    return x.materialize() if isinstance(x, _Pending) else x


def _cache_key(module, dtype, device):
    """Identity of the packed weights: dtype, device and, per parameter/buffer, (storage address, in-place
    version).  ``load_state_dict``, ``.to()``, ``.half()`` and optimiser-style in-place updates all change it;
    edits through ``tensor.data`` do not bump ``_version`` - call ``invalidate_packed`` after those."""
    vers = tuple((int(p.data_ptr()), int(p._version)) for p in module.parameters()) \
        + tuple((int(b.data_ptr()), int(b._version)) for b in module.buffers())
    return (dtype, str(device), vers)


def invalidate_packed(root):
    """Drop the kernel-side weight copies of every module under ``root`` (they are rebuilt on the next forward)
    and, if ``root`` owns captured HIP graphs, those too - a graph replays the weight buffers it was captured
    with."""
    for m in root.modules():
        m.__dict__.pop("_cft_cache", None)
        m.__dict__.pop("_cft_cache_train", None)
        if "_graphs" in m.__dict__:
            m.__dict__["_graphs"].clear()
        m.__dict__.pop("_wlist", None)


class _Packed(nn.Module):
    """Mixin: lazily packed kernel-side weights, rebuilt when dtype/device/parameters change."""

    def _packed(self, dtype, device, train=False):
        """Kernel-side weights for the inference form (BatchNorm folded) or, ``train=True``, the training form
        (``_pack_train``: un-folded conv weights)."""
        key = (train,) + _cache_key(self, dtype, device)
        slot = "_cft_cache_train" if train else "_cft_cache"
        cache = self.__dict__.get(slot)
        if cache is None or cache[0] != key:
            with torch.no_grad():
                cache = (key, self._pack_train(dtype, device) if train else self._pack(dtype, device))
            self.__dict__[slot] = cache
        return cache[1]

    def _pack_train(self, dtype, device):
        raise NotImplementedError(f"{type(self).__name__} has no training-mode form")

    def _pack(self, dtype, device):
        raise NotImplementedError


def _folded(conv_module):
    """(weight, bias) of a Conv with BatchNorm folded, or of an already fused conv."""
    conv = conv_module.conv
    if hasattr(conv_module, "bn"):
        bn = conv_module.bn
        return ops.fold_bn(conv.weight, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.eps)
    bias = conv.bias.float() if conv.bias is not None else torch.zeros(conv.out_channels, device=conv.weight.device)
    return conv.weight.float(), bias


def _f32(t, device):
    """Contiguous fp32 device copy of a parameter/buffer that a kernel reads through a raw float pointer."""
    return t.detach().to(device=device, dtype=torch.float32).contiguous()


def _act_code(act):
    if isinstance(act, nn.SiLU):
        return ACT_SILU
    if isinstance(act, nn.Identity):
        return ACT_NONE
    raise NotImplementedError(f"activation {type(act).__name__} has no fused epilogue (SiLU / Identity only)")


# ------------------------------------------------------------------------------ Conv family
class Conv(_Packed):
    """Standard convolution: SiLU(BN(conv(x))) (reference models/common.py:36-50)."""

    def __init__(self, c1, c2, k=1, s=1, p=None, g=1, act=True):
        super().__init__()
        if g != 1:
            raise NotImplementedError("grouped convolution is not used by any CFT config")
        if autopad(k, p) != k // 2:
            raise NotImplementedError("only 'same' padding (k//2) is implemented")
        self.conv = nn.Conv2d(c1, c2, k, s, autopad(k, p), groups=g, bias=False)
        self.bn = nn.BatchNorm2d(c2, eps=BN_EPS, momentum=0.03)
        self.act = nn.SiLU() if act is True else (act if isinstance(act, nn.Module) else nn.Identity())

    def _pack(self, dtype, device, cin_pad=None):
        w, b = _folded(self)
        return ops.pack_conv(w, b, dtype, s=self.conv.stride[0], cin_pad=cin_pad or self.cin_pad, device=device)

    cin_pad = None      # Focus sets 16 on its inner Conv (space-to-depth output is padded to 16 channels)

    def _pack_train(self, dtype, device):
        return ops.pack_conv(self.conv.weight.float(), None, dtype, s=self.conv.stride[0], cin_pad=self.cin_pad, device=device)

    def forward(self, x, residual=None, out=None):
        x = resolve(x)
        if self.training and hasattr(self, "bn") and self.bn.training:
            # act(bn(conv(x))) with batch statistics (reference :45-47): fp32 conv output, then cft_batchnorm_train
            pk = self._packed(x.dtype, x.device, train=True)
            y32 = ops.conv2d(x, pk, ACT_NONE, out_dtype=torch.float32)
            return ops.batchnorm_train(y32, pk.n_valid, self.bn, _act_code(self.act), residual=residual, out=out, out_dtype=x.dtype)
        return ops.conv2d(x, self._packed(x.dtype, x.device), _act_code(self.act), residual=residual, out=out)

    fuseforward = forward  # the kernel always runs the folded form (reference :49-50)


class Bottleneck(nn.Module):
    """x + cv2(cv1(x)) (reference models/common.py:99-109); the add is cv2's epilogue."""

    def __init__(self, c1, c2, shortcut=True, g=1, e=0.5):
        super().__init__()
        c_ = int(c2 * e)
        self.cv1 = Conv(c1, c_, 1, 1)
        self.cv2 = Conv(c_, c2, 3, 1, g=g)
        self.add = shortcut and c1 == c2

    def forward(self, x, out=None):
        x = resolve(x)
        if x.is_cuda and not self.training:
            a1, a2 = _act_code(self.cv1.act), _act_code(self.cv2.act)
            pk1, pk2 = self.cv1._packed(x.dtype, x.device), self.cv2._packed(x.dtype, x.device)
            aliased = out is not None and out.data_ptr() == x.data_ptr()   # in-place: the fused kernel reads a halo of x
            if ops.bottleneck_fusable(x, pk1, pk2, a1, a2) and not aliased:
                return ops.bottleneck(x, pk1, pk2, self.add, out=out)      # 64-channel stage: one kernel, no hidden tensor
        return self.cv2(self.cv1(x), residual=x if self.add else None, out=out)


class C3(_Packed):
    """CSP bottleneck with 3 convolutions (reference models/common.py:131-143):
    cv3(cat(m(cv1(x)), cv2(x))).  cv1 and cv2 read the same input, so they run as one GEMM whose
    output IS the concat buffer; the last Bottleneck writes its result back into the first half."""

    def __init__(self, c1, c2, n=1, shortcut=True, g=1, e=0.5):
        super().__init__()
        c_ = int(c2 * e)
        self.cv1 = Conv(c1, c_, 1, 1)
        self.cv2 = Conv(c1, c_, 1, 1)
        self.cv3 = Conv(2 * c_, c2, 1)
        self.m = nn.Sequential(*[Bottleneck(c_, c_, shortcut, g, e=1.0) for _ in range(n)])

    def _pack(self, dtype, device):
        w1, b1 = _folded(self.cv1)
        w2, b2 = _folded(self.cv2)
        return ops.pack_conv(torch.cat([w1, w2], 0), torch.cat([b1, b2], 0), dtype, device=device)

    def forward(self, x, out=None):
        c_ = self.cv1.conv.out_channels
        if _act_code(self.cv1.act) != _act_code(self.cv2.act):
            raise NotImplementedError("C3.cv1 and C3.cv2 must share an activation")
        cat = None
        if cat is None and isinstance(x, PendingConv) and not self.training:      # the Conv in front of this C3 was left to it: one kernel for both
            src, conv = resolve(x.x), x.conv
            pk1 = conv._packed(src.dtype, src.device)
            if _act_code(conv.act) == ACT_SILU and ops.conv2d_chain_ok(src, pk1, self._packed(src.dtype, src.device)):
                cat = ops.conv2d_chain(src, pk1, self._packed(src.dtype, src.device), _act_code(self.cv1.act))
        if cat is not None:
            pass
        elif self.training:      # batch statistics are per BatchNorm: cv1 and cv2 run separately into the concat buffer
            x = resolve(x)
            B, _, H, W = x.shape
            cat = ops.new_nhwc(B, H, W, 2 * c_, x.dtype, x.device)
            self.cv1(x, out=cat[:, :c_])
            self.cv2(x, out=cat[:, c_:])
        else:
            x = resolve(x)
            cat = ops.conv2d(x, self._packed(x.dtype, x.device), _act_code(self.cv1.act))     # [B, 2c_, H, W]
        head = cat[:, :c_]
        y = head
        n = len(self.m)
        if self.chain and self.chain_pairs and n >= 2 and not self.training and self._res_chainable(head):
            # WITH shortcuts (backbone C3s, 256 channels): Bottleneck j's output is the next shortcut AND the input of Bottleneck j + 1's 1x1 -
            # cv2[j] (+ shortcut) and cv1[j + 1] run as one launch (ops.conv2d_chain_res): n + 1 launches instead of 2 n, no re-read of y
            h = self.m[0].cv1(y)
            for j in range(n - 1):
                blk, nxt = self.m[j], self.m[j + 1]
                y, h = ops.conv2d_chain_res(h, blk.cv2._packed(h.dtype, h.device), y, nxt.cv1._packed(h.dtype, h.device), _act_code(nxt.cv1.act))
            self.m[n - 1].cv2(h, residual=y, out=head)
            return self.cv3(cat, out=out)
        # Without shortcuts a Bottleneck's output has ONE reader, the next Bottleneck's 1x1 (reference models/common.py:108-109, :142):
        # cv2[j] + cv1[j + 1] then run as one launch (ops.conv2d_chain) and the tensor between the two Bottlenecks never exists.
        can = [self.chain and self.chain_pairs and j + 1 < n and self._chainable(self.m[j], self.m[j + 1], head) for j in range(n)]
        hidden = None              # cv1 output of Bottleneck j, already produced by the chained launch of Bottleneck j - 1
        for j, blk in enumerate(self.m):
            dst = head if j == n - 1 else None
            if hidden is None and not can[j]:
                y = blk(y, out=dst)
                continue
            h = hidden if hidden is not None else blk.cv1(y)
            if can[j]:
                nxt = self.m[j + 1]
                hidden = ops.conv2d_chain(h, blk.cv2._packed(h.dtype, h.device), nxt.cv1._packed(h.dtype, h.device), _act_code(nxt.cv1.act))
            else:
                hidden, y = None, blk.cv2(h, out=dst)
        return self.cv3(cat, out=out)

    chain = True                   # Model.chain_convs switches it (A/B)
    chain_pairs = True             # the Bottleneck-pair chains (3x3 of Bottleneck j + 1x1 of Bottleneck j + 1) on their own (A/B: bench.py --no-pair-chain)

    def _res_chainable(self, y):
        """Every Bottleneck has a shortcut and (cv2[j], cv1[j + 1]) is a pair ``ops.conv2d_chain_res`` takes on tensors shaped like ``y``."""
        if not (isinstance(y, torch.Tensor) and y.is_cuda and y.dtype in (torch.bfloat16, torch.float16)):
            return False
        key = (tuple(y.shape), y.dtype, y.device)          # (the decision depends on the pixel grid and the type only: h and y1 are dense tensors of this shape)
        cache = self.__dict__.setdefault("_res_chain_cache", {})
        if key not in cache:
            if len(cache) > 16:
                cache.clear()
            cache[key] = self._res_chainable_uncached(y)
        return cache[key]

    def _res_chainable_uncached(self, y):
        for j, blk in enumerate(self.m):
            if not blk.add or _act_code(blk.cv1.act) != ACT_SILU or _act_code(blk.cv2.act) != ACT_SILU:
                return False
            pk1, pk2 = blk.cv1._packed(y.dtype, y.device), blk.cv2._packed(y.dtype, y.device)
            if ops.bottleneck_fusable(y, pk1, pk2, ACT_SILU, ACT_SILU):
                return False               # 64 / 128 channels: the patch-resident Bottleneck kernel
            if j + 1 < len(self.m) and not ops.conv2d_chain_res_ok_geometry(y.shape[0], y.shape[2], y.shape[3], y.dtype, pk2,
                                                                            self.m[j + 1].cv1._packed(y.dtype, y.device)):
                return False               # (dense h / y1 of this pixel grid - what the launch below passes)
        return True

    def _chainable(self, blk, nxt, y):
        """``blk.cv2`` (3x3) and ``nxt.cv1`` (1x1) can run as one kernel on tensors shaped like ``y``: inference, no shortcut on either
        (``blk``'s output is then read by ``nxt.cv1`` only), SiLU, and a geometry the chained kernel takes."""
        if self.training or blk.add or nxt.add or not (isinstance(y, torch.Tensor) and y.is_cuda) or _act_code(blk.cv2.act) != ACT_SILU:
            return False
        if ops.bottleneck_fusable(y, blk.cv1._packed(y.dtype, y.device), blk.cv2._packed(y.dtype, y.device),
                                  _act_code(blk.cv1.act), _act_code(blk.cv2.act)):
            return False               # 64 / 128 channels: the patch-resident Bottleneck kernel is the better fusion
        return ops.conv2d_chain_ok_geometry(y.shape[0], y.shape[2], y.shape[3], y.dtype, blk.cv2._packed(y.dtype, y.device),
                                            nxt.cv1._packed(y.dtype, y.device))


class SPP(nn.Module):
    """Spatial pyramid pooling (reference models/common.py:154-165)."""

    def __init__(self, c1, c2, k=(5, 9, 13)):
        super().__init__()
        c_ = c1 // 2
        self.cv1 = Conv(c1, c_, 1, 1)
        self.cv2 = Conv(c_ * (len(k) + 1), c2, 1, 1)
        self.m = nn.ModuleList([nn.MaxPool2d(kernel_size=x, stride=1, padding=x // 2) for x in k])
        self.k = tuple(k)

    def forward(self, x):
        x = resolve(x)
        ks = tuple(int(mp.kernel_size) for mp in self.m)   # (not self.k: un-pickled reference modules lack it)
        if len(ks) != 3 or list(ks) != sorted(ks):
            raise NotImplementedError("SPP kernel implements exactly three ascending pooling sizes")
        c_ = self.cv1.conv.out_channels
        B, _, H, W = x.shape
        cat = ops.new_nhwc(B, H, W, 4 * c_, x.dtype, x.device)
        self.cv1(x, out=cat[:, :c_])
        ops.spp_maxpool(cat, c_, ks)
        return self.cv2(cat)


class Focus(_Packed):
    """Focus wh information into c-space (reference models/common.py:168-179): 2x2 space-to-depth
    then Conv.  Takes the fp32 image batch [B,3,H,W] (NCHW) and emits NHWC activations in
    ``compute_dtype`` - this module is where the compute precision of the whole network is set."""

    compute_dtype = None   # None: the precision of the parameters (fp32 -> exact fp32 path, .half() -> fp16), like torch;
                           # Model.set_compute_dtype() sets it explicitly (its default is bf16)

    def __init__(self, c1, c2, k=1, s=1, p=None, g=1, act=True):
        super().__init__()
        self.conv = Conv(c1 * 4, c2, k, s, p, g, act)
        self.conv.cin_pad = 16

    def _pack(self, dtype, device):
        if self.conv.conv.in_channels != 12:
            raise NotImplementedError("Focus kernel is specialised for 3-channel images")
        return self.conv._pack(dtype, device, cin_pad=16)

    def forward(self, x):
        x = resolve(x)
        dtype = self.compute_dtype or self.conv.conv.weight.dtype
        if self.training:      # space-to-depth, then the Conv's training form (batch statistics)
            self.conv.cin_pad = 16
            return self.conv(ops.focus_s2d(x, dtype))
        return ops.focus_conv(x, self._packed(dtype, x.device), _act_code(self.conv.act), dtype)


class Upsample(nn.Upsample):
    """``nn.Upsample(None, 2, 'nearest')`` of the head (yaml rows 33/37).  Returns a deferred value
    that ``Concat`` writes directly into its buffer."""

    def forward(self, x):
        x = resolve(x)
        sf = self.scale_factor
        if self.mode != "nearest" or self.size is not None or float(sf) not in (2.0, 4.0, 8.0):
            raise NotImplementedError("only nearest-neighbour upsampling by 2/4/8 is implemented")
        return PendingUpsample(x, int(math.log2(float(sf))))


class Concat(nn.Module):
    """Channel concatenation (reference models/common.py:211-219)."""

    def __init__(self, dimension=1):
        super().__init__()
        self.d = dimension

    def forward(self, x, out=None):
        """``out``: a pre-allocated concat buffer (Model's concat plan); sources that their producer already wrote
        into their slice of it are not copied again."""
        if self.d != 1:
            raise NotImplementedError("Concat is implemented along channels only")
        shapes = [t.shape for t in x]
        B, _, H, W = shapes[0]
        first = x[0].x if isinstance(x[0], PendingUpsample) else resolve(x[0])
        total = sum(s[1] for s in shapes)
        if out is None:
            out = ops.new_nhwc(B, H, W, total, first.dtype, first.device)
        elif tuple(out.shape) != (B, total, H, W):
            raise ValueError(f"Concat: planned buffer {tuple(out.shape)} does not match the inputs {(B, total, H, W)}")
        off = 0
        for t, s in zip(x, shapes):
            dst = out[:, off:off + s[1]]
            if isinstance(t, PendingUpsample):
                ops.copy_channels(t.x, dst, t.up)
            else:
                t = resolve(t)
                if not (t.data_ptr() == dst.data_ptr() and t.stride() == dst.stride()):   # else: already in place
                    ops.copy_channels(t, dst, 0)
            off += s[1]
        return out


class Add(nn.Module):
    """Sum of two feature maps (reference models/common.py:222-229)."""

    def __init__(self, arg):
        super().__init__()
        self.arg = arg

    def forward(self, x, out=None):
        return ops.add(resolve(x[0]), resolve(x[1]), out=out)


class Add2(nn.Module):
    """x[0] + transformer_output[index] (reference models/common.py:232-243); when the transformer
    output is still deferred the bilinear upsampling and the add are one kernel."""

    def __init__(self, c1, index):
        super().__init__()
        self.index = index

    def forward(self, x):
        if self.index not in (0, 1):
            raise ValueError("Add2 index must be 0 or 1")
        base, t = resolve(x[0]), x[1][self.index]
        if isinstance(t, PendingBilinear):
            return t.materialize(base)
        return ops.add(base, resolve(t))


# ------------------------------------------------------------------------------ CFT block
class SelfAttention(_Packed):
    """Multi-head self-attention (reference models/common.py:430-513).  q/k/v projections run as
    one GEMM; heads are laid out at a padded width ``dkp`` so the attention kernel needs no tails."""

    def __init__(self, d_model, d_k, d_v, h, attn_pdrop=.1, resid_pdrop=.1):
        super().__init__()
        assert d_k % h == 0
        self.d_model = d_model
        self.d_k = d_model // h
        self.d_v = d_model // h
        self.h = h
        self.que_proj = nn.Linear(d_model, h * self.d_k)
        self.key_proj = nn.Linear(d_model, h * self.d_k)
        self.val_proj = nn.Linear(d_model, h * self.d_v)
        self.out_proj = nn.Linear(h * self.d_v, d_model)
        self.attn_drop = nn.Dropout(attn_pdrop)
        self.resid_drop = nn.Dropout(resid_pdrop)
        for m in (self.que_proj, self.key_proj, self.val_proj, self.out_proj):  # reference :459-473
            nn.init.normal_(m.weight, std=0.001)
            nn.init.constant_(m.bias, 0)

    @staticmethod
    def padded_head(dk, dtype):
        step = 16 if dtype == torch.float32 else 32
        return ((dk + step - 1) // step) * step

    def _pack(self, dtype, device):
        h, dk, d = self.h, self.d_k, self.d_model
        dkp = self.padded_head(dk, dtype)
        ws, bs = [], []
        for lin in (self.que_proj, self.key_proj, self.val_proj):
            w = torch.zeros((h, dkp, d), dtype=torch.float32, device=device)
            b = torch.zeros((h, dkp), dtype=torch.float32, device=device)
            w[:, :dk] = lin.weight.detach().float().to(device).view(h, dk, d)
            b[:, :dk] = lin.bias.detach().float().to(device).view(h, dk)
            ws.append(w.view(h * dkp, d))
            bs.append(b.view(h * dkp))
        qkv = ops.pack_conv(torch.cat(ws, 0), torch.cat(bs, 0), dtype, device=device, n_true=3 * h * dk)
        wo = torch.zeros((d, h, dkp), dtype=torch.float32, device=device)
        wo[:, :, :dk] = self.out_proj.weight.detach().float().to(device).view(d, h, dk)
        out = ops.pack_conv(wo.view(d, h * dkp), self.out_proj.bias, dtype, device=device)
        out.flops_per_row = 2.0 * d * h * dk
        return qkv, out, dkp

    def forward(self, x, attention_mask=None, attention_weights=None, residual=None, splitk=False):
        """x: [B*128, d] in the compute dtype.  Returns out_proj(attn) (+ residual, fp32, in place).  ``splitk`` (the transformer block's
        inference path): returns the out_proj's fp32 split-K partial sums [s, rows, d] when ``ops.splitk_choice`` splits it (``residual`` is
        then NOT updated - ``ops.layernorm_reduce`` does that), else None after the in-place update."""
        if attention_mask is not None or attention_weights is not None:
            raise NotImplementedError("attention_mask / attention_weights are unused by GPT (reference :497-500)")
        qkv_w, out_w, dkp = self._packed(x.dtype, x.device)
        B = x.shape[0] // 128
        qkv = ops.linear(x, qkv_w)
        att = ops.attention(qkv, B, self.h, self.d_k, dkp, pdrop=self.attn_drop.p if self.training else 0.0)
        if self.training and self.resid_drop.p > 0 and residual is not None:
            proj = ops.linear(att, out_w, out_dtype=torch.float32)          # resid_drop(out_proj(.)) then the residual add
            ops.dropout_(proj, self.resid_drop.p)
            return ops.add_rows_(residual, proj)
        if splitk and residual is not None and not self.training:
            s_ = ops.splitk_choice(att.shape[0], out_w, att.dtype)
            if s_ > 1:
                return ops.linear_splitk(att, out_w, s_)         # fp32 partial sums; the caller's next LayerNorm folds them into ``residual``
        out = ops.linear(att, out_w, residual=residual, out=residual,
                         out_dtype=torch.float32 if residual is not None else None)
        return None if splitk else out


class myTransformerBlock(_Packed):
    """Pre-LN transformer block (reference models/common.py:516-546).  ``x`` is the fp32 residual
    stream [B*128, d]; it is updated in place by the two GEMM epilogues."""

    def __init__(self, d_model, d_k, d_v, h, block_exp, attn_pdrop, resid_pdrop):
        super().__init__()
        self.ln_input = nn.LayerNorm(d_model)
        self.ln_output = nn.LayerNorm(d_model)
        self.sa = SelfAttention(d_model, d_k, d_v, h, attn_pdrop, resid_pdrop)
        self.mlp = nn.Sequential(
            nn.Linear(d_model, block_exp * d_model),
            nn.GELU(),
            nn.Linear(block_exp * d_model, d_model),
            nn.Dropout(resid_pdrop),
        )

    def _pack(self, dtype, device):
        fc1 = ops.pack_conv(self.mlp[0].weight, self.mlp[0].bias, dtype, device=device)
        fc2 = ops.pack_conv(self.mlp[2].weight, self.mlp[2].bias, dtype, device=device)
        ln = [_f32(t, device) for t in (self.ln_input.weight, self.ln_input.bias, self.ln_output.weight, self.ln_output.bias)]
        return fc1, fc2, ln

    splitk = True        # inference: out_proj / fc2 as split-K GEMMs where ops.splitk_choice splits them (GPT.splitk switches all blocks; A/B)

    def forward(self, x, compute_dtype=torch.bfloat16, pending=None):
        """``pending``: fp32 split-K partial sums of the PREVIOUS block's fc2 that are not yet folded into ``x`` (or None).  Returns this
        block's own pending partial sums (or None when its fc2 updated ``x`` in place); ``GPT.forward`` hands them to the next block /
        ``ln_f``.  Each LayerNorm that follows a split GEMM is ``ops.layernorm_reduce``: x += partial sums (fixed order), then LayerNorm."""
        fc1, fc2, ln = self._packed(compute_dtype, x.device)
        split = self.splitk and not self.training and x.is_cuda

        def norm(parts, g, b, eps):
            if parts is not None:
                return ops.layernorm_reduce(x, parts, g, b, compute_dtype, eps)
            return ops.layernorm(x, g, b, compute_dtype, eps)
        y = norm(pending, ln[0], ln[1], self.ln_input.eps)
        parts = None
        if split:
            parts = self.sa(y, residual=x, splitk=True)           # x += out_proj(attention(LN(x))), or its partial sums
        else:
            self.sa(y, residual=x)
        y = norm(parts, ln[2], ln[3], self.ln_output.eps)
        hid = ops.linear(y, fc1, act=ACT_GELU)
        pd = self.mlp[3].p if (self.training and len(self.mlp) > 3) else 0.0
        if pd > 0:
            z = ops.linear(hid, fc2, out_dtype=torch.float32)
            ops.dropout_(z, pd)
            ops.add_rows_(x, z)
            return None
        if split:
            s_ = ops.splitk_choice(hid.shape[0], fc2, hid.dtype)
            if s_ > 1:
                return ops.linear_splitk(hid, fc2, s_)                     # folded into x by the next LayerNorm (next block's ln_input / ln_f)
        ops.linear(hid, fc2, residual=x, out=x, out_dtype=torch.float32)  # x += fc2(gelu(fc1(LN(x))))
        return None


class GPT(_Packed):
    """Cross-modality fusion transformer (reference models/common.py:549-639)."""

    def __init__(self, d_model, h=8, block_exp=4, n_layer=8, vert_anchors=8, horz_anchors=8,
                 embd_pdrop=0.1, attn_pdrop=0.1, resid_pdrop=0.1):
        super().__init__()
        self.n_embd = d_model
        self.vert_anchors = vert_anchors
        self.horz_anchors = horz_anchors
        d_k = d_model
        d_v = d_model
        self.pos_emb = nn.Parameter(torch.zeros(1, 2 * vert_anchors * horz_anchors, self.n_embd))
        self.trans_blocks = nn.Sequential(*[myTransformerBlock(d_model, d_k, d_v, h, block_exp, attn_pdrop, resid_pdrop)
                                            for _ in range(n_layer)])
        self.ln_f = nn.LayerNorm(self.n_embd)
        self.drop = nn.Dropout(embd_pdrop)
        self.avgpool = nn.AdaptiveAvgPool2d((self.vert_anchors, self.horz_anchors))
        self.apply(self._init_weights)

    @staticmethod
    def _init_weights(module):  # reference :582-591
        if isinstance(module, nn.Linear):
            module.weight.data.normal_(mean=0.0, std=0.02)
            if module.bias is not None:
                module.bias.data.zero_()
        elif isinstance(module, nn.LayerNorm):
            module.bias.data.zero_()
            module.weight.data.fill_(1.0)

    def _pack(self, dtype, device):   # fp32 copies of what the kernels read through raw pointers (fp16 after model.half())
        return _f32(self.pos_emb, device), _f32(self.ln_f.weight, device), _f32(self.ln_f.bias, device)

    def forward(self, x):
        if self.vert_anchors != 8 or self.horz_anchors != 8:
            raise NotImplementedError("the CFT kernels are specialised for the 8x8 anchor grid (128 tokens)")
        rgb, ir = resolve(x[0]), resolve(x[1])
        assert rgb.shape[0] == ir.shape[0]
        if rgb.shape != ir.shape or rgb.dtype != ir.dtype:
            raise ValueError("GPT: the two streams must have the same shape and dtype")
        B, C, H, W = rgb.shape
        dtype = rgb.dtype
        pos_emb, lnf_w, lnf_b = self._packed(dtype, rgb.device)
        tok = ops.gpt_tokenize(rgb, ir, pos_emb)                 # fp32 [B,128,C], pos_emb added
        if self.training:
            ops.dropout_(tok, self.drop.p)                       # self.drop(pos_emb + token_embeddings), reference :611
        t2 = tok.view(B * 128, C)
        pending = None                                           # split-K partial sums of the last fc2, not yet folded into t2
        for blk in self.trans_blocks:
            pending = blk(t2, dtype, pending)
        if pending is not None:
            tok_f = ops.layernorm_reduce(t2, pending, lnf_w, lnf_b, torch.float32, self.ln_f.eps).view(B, 128, C)
        else:
            tok_f = ops.layernorm(t2, lnf_w, lnf_b, torch.float32, self.ln_f.eps).view(B, 128, C)
        return PendingBilinear(tok_f, 0, H, W, dtype), PendingBilinear(tok_f, 1, H, W, dtype)


# ------------------------------------------------------------------------------ names outside the hot path
# The reference's graph file does ``from models.common import *`` and names these classes in its parse_model
# membership tests (models/yolo_test.py:496-531), so they must EXIST for that file to run over this module
# (strict form); no CFT / two-stream yaml instantiates them (SURVEY.md 2, rows marked OUT).
def _outside(name, where):
    def __init__(self, *a, **k):
        raise NotImplementedError(f"{name} (reference {where}) is outside the two-stream CFT hot path; "
                                  "no fusion yaml uses it (SURVEY.md section 2)")
    return type(name, (nn.Module,), {"__init__": __init__, "__doc__": f"Placeholder for the reference's {name} ({where})."})


def DWConv(c1, c2, k=1, s=1, act=True):  # reference models/common.py:31-33 (depthwise: grouped conv, not in any CFT yaml)
    raise NotImplementedError("DWConv (grouped convolution) is outside the two-stream CFT hot path")


TransformerLayer = _outside("TransformerLayer", "models/common.py:53-67")
TransformerBlock = _outside("TransformerBlock", "models/common.py:70-96")
BottleneckCSP = _outside("BottleneckCSP", "models/common.py:112-128")
C3TR = _outside("C3TR", "models/common.py:146-151")
Contract = _outside("Contract", "models/common.py:183-194")
Expand = _outside("Expand", "models/common.py:197-208")


class NMS(nn.Module):
    """Non-maximum-suppression module (reference models/common.py:247-257): what ``Model.nms()`` appends behind ``Detect``.
    ``forward(x)`` takes Detect's ``(pred, raw)`` and returns the per-image detection list of ``non_max_suppression`` -
    here ONE batched HIP kernel (``cft_nms``) instead of a Python loop around ``torchvision.ops.nms``."""
    conf = 0.25      # confidence threshold
    iou = 0.45       # IoU threshold
    classes = None   # (optional list) filter by class

    def forward(self, x):
        from ..utils.general import non_max_suppression
        return non_max_suppression(x[0], conf_thres=self.conf, iou_thres=self.iou, classes=self.classes)


class autoShape(nn.Module):
    """Input-robust wrapper (reference models/common.py:260-327) in the TWO-STREAM form the reference never finished: its
    ``autoShape.forward`` hands ONE image batch to a ``Model`` whose ``forward(x, x2)`` needs two (the call raises for every
    fusion yaml).  Here ``forward(rgb, ir, size=640)`` takes, per stream, one image or a list of images - HWC uint8 numpy arrays or
    tensors in RGB order (``cv2.imread(...)[:, :, ::-1]``, ``np.asarray(PIL.Image)``), or CHW - or two ready BCHW float tensors
    (passed straight to the model like the reference does, :283-285).  Pre-processing follows :288-309 (common inference
    shape = the largest scaled image rounded up to the stride, ``letterbox(auto=False)``, BHWC -> BCHW, / 255) with the letterbox on
    the device (``cft_letterbox_u8``) and the / 255 fused into Focus; post-processing :317-320 (``non_max_suppression`` =
    ``cft_nms``, ``scale_coords`` back to every original image).  Returns the list of per-image ``[n, 6]`` (xyxy, conf, cls)
    tensors in ORIGINAL-image pixels (the reference wraps the same list in its plotting class ``Detections``, out of scope)."""
    conf = 0.25
    iou = 0.45
    classes = None

    def __init__(self, model):
        super().__init__()
        self.model = model.eval()
        for k in ("yaml", "nc", "hyp", "names", "stride"):          # copy_attr(m, self, include=(...)) of reference :312-313
            if hasattr(model, k):
                setattr(self, k, getattr(model, k))

    def autoshape(self):
        return self      # already wrapped (reference :271-273)

    @staticmethod
    def _hwc_u8(im, device):
        import numpy as np
        if isinstance(im, torch.Tensor):
            t = im
        else:
            t = torch.from_numpy(np.ascontiguousarray(np.asarray(im)))
        if t.dim() == 2:
            t = t[:, :, None].expand(-1, -1, 3)
        if t.shape[0] < 5:                                          # CHW -> HWC (reference :298-299)
            t = t.permute(1, 2, 0)
        t = t[:, :, :3]
        if t.dtype != torch.uint8:
            raise TypeError("autoShape: images must be uint8 (or pass two BCHW float tensors)")
        return t.contiguous().to(device)

    @torch.no_grad()
    def forward(self, rgb, ir, size=640, augment=False, profile=False):
        from ..models.yolo_test import make_divisible
        from ..utils.datasets import letterbox
        from ..utils.general import non_max_suppression, scale_coords
        p = next(self.model.parameters())
        if isinstance(rgb, torch.Tensor) and rgb.dim() == 4:        # ready batches: straight to the model
            return self.model(rgb.to(p.device), ir.to(p.device), augment, profile)
        rgbs = list(rgb) if isinstance(rgb, (list, tuple)) else [rgb]
        irs = list(ir) if isinstance(ir, (list, tuple)) else [ir]
        if len(rgbs) != len(irs):
            raise ValueError("autoShape: the two streams must hold the same number of images")
        rgbs = [self._hwc_u8(im, p.device) for im in rgbs]
        irs = [self._hwc_u8(im, p.device) for im in irs]
        shape0 = [tuple(int(v) for v in im.shape[:2]) for im in rgbs]
        if any(tuple(b.shape[:2]) != s for b, s in zip(irs, shape0)):
            raise ValueError("autoShape: an RGB / IR pair must have the same height and width (aligned pairs)")
        smax = int(self.stride.max()) if hasattr(self, "stride") else 32
        shape1 = [max(s[d] * (size / max(s)) for s in shape0) for d in (0, 1)]
        shape1 = [make_divisible(v, smax) for v in shape1]          # one inference shape for the batch (:304)
        n = len(rgbs)
        batch = torch.empty((n, 6, shape1[0], shape1[1]), dtype=torch.uint8, device=p.device)
        for i in range(n):                                          # BGR->RGB flip is not wanted here (inputs are RGB): HWC out, then permute
            a, _, _ = letterbox(rgbs[i], new_shape=shape1, auto=False)
            b, _, _ = letterbox(irs[i], new_shape=shape1, auto=False)
            batch[i, :3] = a.permute(2, 0, 1)
            batch[i, 3:] = b.permute(2, 0, 1)
        y = self.model(batch[:, :3], batch[:, 3:])[0]               # uint8 views: Focus normalises (/255) while it reads
        y = non_max_suppression(y, conf_thres=self.conf, iou_thres=self.iou, classes=self.classes)
        for i in range(n):
            scale_coords(shape1, y[i][:, :4], shape0[i])
        return y

Classify = _outside("Classify", "models/common.py:417-427")
