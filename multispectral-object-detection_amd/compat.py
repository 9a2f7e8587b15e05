"""Make this package answer to the reference's module names.

The reference's callers (`train.py:26-27`, `test.py`, `detect_twostream.py`) do
``from models.yolo_test import Model`` / ``from models.common import *`` and its checkpoints are
pickled whole ``nn.Module`` objects (``train.py:850-860``) whose classes are recorded as
``models.yolo_test.Model`` and ``models.common.<Class>``.  After
``install_reference_aliases()`` those names resolve to the MI355X-native classes, so

    import msod_amd.compat as c; c.install_reference_aliases()
    from models.yolo_test import Model            # -> msod_amd.models.yolo_test.Model
    ckpt = torch.load("best.pt", weights_only=False)   # un-pickles into the HIP-backed modules

works without touching the caller.  (Un-pickling restores parameters and buffers; the kernel-side
packed weights are rebuilt lazily on first forward.)
"""
import sys
import types


def install_reference_aliases(force=False):
    from .models import common, yolo_test
    for name, mod in (("models.common", common), ("models.yolo_test", yolo_test)):
        if name in sys.modules and sys.modules[name] is not mod and not force:
            raise RuntimeError(f"{name} is already imported from {getattr(sys.modules[name], '__file__', '?')}; "
                               "pass force=True to override it")
        sys.modules[name] = mod
    pkg = sys.modules.get("models")
    if pkg is None or force:
        pkg = types.ModuleType("models")
        pkg.__path__ = []
        sys.modules["models"] = pkg
    pkg.common = common
    pkg.yolo_test = yolo_test
    return pkg


def install_common_alias(force=False):
    """Strict form of the drop-in: ONLY ``models.common`` resolves to this package; the
    reference's own graph file ``models/yolo_test.py`` (its ``Model`` / ``parse_model`` / ``forward_once`` /
    ``Detect``) is imported unchanged from the reference tree, which must be on ``sys.path``.  The modules then
    compute in the precision of their parameters (fp32, or fp16 after ``model.half()``), because the reference's
    ``Detect`` and ``nn.Upsample`` are torch modules that see the activations."""
    from .models import common
    if "models.common" in sys.modules and sys.modules["models.common"] is not common and not force:
        raise RuntimeError("models.common is already imported from "
                           f"{getattr(sys.modules['models.common'], '__file__', '?')}; pass force=True to override it")
    import importlib
    sys.modules.pop("models", None)
    pkg = importlib.import_module("models")          # the reference's package (namespace or regular)
    sys.modules["models.common"] = common
    pkg.common = common
    return common


def _ensemble_class():
    import torch
    import torch.nn as nn

    class Ensemble(nn.ModuleList):
        """Ensemble of two-stream models (reference models/experimental.py:98-110): every member sees the same
        pair, the prediction tensors are concatenated along the box axis ("nms ensemble", :108) and the caller's
        NMS merges them.  (The reference's own ``Ensemble.forward(x, augment)`` passes ONE image to members that
        need two - it predates the two-stream ``Model.forward(x, x2)``, models/yolo_test.py:214; the pair form is
        what a two-stream caller needs.)"""

        def forward(self, x, x2, augment=False):
            y = [module(x, x2, augment)[0] for module in self]
            return torch.cat(y, 1), None        # members' pred tensors are fresh fp32 [B, rows, no] buffers

    return Ensemble


def attempt_load(weights, map_location=None):
    """The reference's ``attempt_load`` (models/experimental.py:113-134): ``weights`` is one checkpoint path or a
    list of them; each pickled checkpoint -> ``ema`` or ``model`` -> ``.float().fuse().eval()`` (an fp32 model,
    as in the reference: callers then opt into 16-bit with ``model.half()``, test.py:66-68, or with
    ``set_compute_dtype``).  One path returns the model, several return an ``Ensemble`` carrying the last
    member's ``names`` / ``stride`` (:129-133)."""
    import torch
    install_reference_aliases()
    members = []
    for w in weights if isinstance(weights, (list, tuple)) else [weights]:
        ckpt = torch.load(w, map_location=map_location, weights_only=False)
        model = ckpt["ema" if ckpt.get("ema") else "model"] if isinstance(ckpt, dict) else ckpt
        adopt_torch_modules(model)
        members.append(model.float().fuse().eval())
    if len(members) == 1:
        return members[-1]
    ens = _ensemble_class()()
    for m in members:
        ens.append(m)
    for k in ("names", "stride"):
        setattr(ens, k, getattr(members[-1], k))
    return ens


def adopt_torch_modules(model):
    """A reference pickle holds plain ``torch.nn.Upsample`` layers (yaml rows 33/37); swap them for
    the HIP-backed subclass, keeping the parse_model tags."""
    import torch.nn as nn
    from .models.common import Upsample
    for name, child in list(model.model.named_children()):
        if type(child) is nn.Upsample:
            new = Upsample(child.size, child.scale_factor, child.mode)
            for tag in ("i", "f", "type", "np"):
                if hasattr(child, tag):
                    setattr(new, tag, getattr(child, tag))
            model.model._modules[name] = new
    return model
