"""Model dictionaries for the two-stream YOLOv5 + CFT detectors.

The reference describes each network as a yaml layer list
(/root/reference models/transformer/*.yaml, rows ``[from, number, module, args]``) that
``parse_model`` (models/yolo_test.py:479-555) turns into modules.  Reference yaml files are
accepted unchanged by ``Model(cfg)``; this module *generates* the same dictionaries so the
framework is usable (and benchmarkable) where the reference tree is not present.  The
generator is checked row-by-row against all 13 reference yamls in
``tests/test_configs.py`` whenever /root/reference is available.

Three fusion layouts exist in the reference:

``add``            two independent CSPDarknet backbones, summed at P3/P4/P5
``transformer``    GPT (CFT) fusion after P2, P3, P4 and P5 (4 blocks)
``transformerx3``  GPT fusion after P3, P4 and P5 (3 blocks) - the BASELINE flagship
"""
from copy import deepcopy

ANCHORS = [[10, 13, 16, 30, 33, 23], [30, 61, 62, 45, 59, 119], [116, 90, 156, 198, 373, 326]]
SIZES = {"s": (0.33, 0.50), "m": (0.67, 0.75), "l": (1.0, 1.0), "x": (1.33, 1.25)}
DATASET_NC = {"flir": 3, "llvip": 1, "vedai": 9}

_UP = ["nn.Upsample", [None, 2, "nearest"]]


def _stem():
    return [[None, 1, "Focus", [64, 3]], [-1, 1, "Conv", [128, 3, 2]], [-1, 3, "C3", [128]]]


def _stage(src, width, depth):
    return [[src, 1, "Conv", [width, 3, 2]], [-1, depth, "C3", [width]]]


def _top(src):
    return [[src, 1, "Conv", [1024, 3, 2]], [-1, 1, "SPP", [1024, [5, 9, 13]]], [-1, 3, "C3", [1024, False]]]


def _head(p3, p4, p5_is_prev, base):
    """PANet head; ``base`` = index of the first head layer, p3/p4 = fused backbone features."""
    assert p5_is_prev
    h = [
        [-1, 1, "Conv", [512, 1, 1]],
        [-1, 1] + deepcopy(_UP),
        [[-1, p4], 1, "Concat", [1]],
        [-1, 3, "C3", [512, False]],
        [-1, 1, "Conv", [256, 1, 1]],
        [-1, 1] + deepcopy(_UP),
        [[-1, p3], 1, "Concat", [1]],
        [-1, 3, "C3", [256, False]],
        [-1, 1, "Conv", [256, 3, 2]],
        [[-1, base + 4], 1, "Concat", [1]],
        [-1, 3, "C3", [512, False]],
        [-1, 1, "Conv", [512, 3, 2]],
        [[-1, base], 1, "Concat", [1]],
        [-1, 3, "C3", [1024, False]],
        [[base + 7, base + 10, base + 13], 1, "Detect", ["nc", "anchors"]],
    ]
    return h


def _fuse(rows, a, b, width):
    """Append GPT(a,b) and the two residual Add2 rows; return (rgb_idx, ir_idx)."""
    g = len(rows)
    rows.append([[a, b], 1, "GPT", [width]])
    rows.append([[a, g], 1, "Add2", [width, 0]])
    rows.append([[b, g], 1, "Add2", [width, 1]])
    return g + 1, g + 2


def cft_config(size="l", fusion="transformerx3", nc=3, anchors=None, rgb_top_relative=True):
    """Build a model dict equal to ``yaml.safe_load`` of the matching reference yaml.

    size: 's' | 'm' | 'l' | 'x' (depth/width multiples); fusion: 'add' | 'transformer' |
    'transformerx3'; nc: classes.  ``rgb_top_relative`` reproduces the one quirk of the
    4-GPT yamls, whose RGB P5 stage is written ``from=-2`` instead of an absolute index
    (yolov5l_fusion_transformer_FLIR.yaml:46).
    """
    gd, gw = SIZES[size]
    rows = []
    if fusion == "add":
        for src in (-1, -4):
            st = _stem(); st[0][0] = src
            rows += st + _stage(-1, 256, 9) + _stage(-1, 512, 9) + _top(-1)
        adds = [[4, 14], [6, 16], [9, 19]]
        for pair in adds:
            rows.append([pair, 1, "Add", [1]])
        p3, p4 = 20, 21
    elif fusion == "transformerx3":
        for src in (-1, -4):
            st = _stem(); st[0][0] = src
            rows += st + _stage(-1, 256, 9)
        r, t = _fuse(rows, 4, 9, 256)
        p3_pair = [r, t]
        for src in (r, t):
            rows += _stage(src, 512, 9)
        r, t = _fuse(rows, len(rows) - 3, len(rows) - 1, 512)
        p4_pair = [r, t]
        for src in (r, t):
            rows += _top(src)
        r, t = _fuse(rows, len(rows) - 4, len(rows) - 1, 1024)
        for pair in (p3_pair, p4_pair, [r, t]):
            rows.append([pair, 1, "Add", [1]])
        p3, p4 = len(rows) - 3, len(rows) - 2
    elif fusion == "transformer":
        for src in (-1, -4):
            st = _stem(); st[0][0] = src
            rows += st
        r, t = _fuse(rows, 2, 5, 128)
        pairs = []
        for width in (256, 512):
            for src in (r, t):
                rows += _stage(src, width, 9)
            r, t = _fuse(rows, len(rows) - 3, len(rows) - 1, width)
            pairs.append([r, t])
        for n, src in enumerate((r, t)):
            rows += _top(-2 if (n == 0 and rgb_top_relative) else src)
        r, t = _fuse(rows, len(rows) - 4, len(rows) - 1, 1024)
        pairs.append([r, t])
        for pair in pairs:
            rows.append([pair, 1, "Add", [1]])
        p3, p4 = len(rows) - 3, len(rows) - 2
    else:
        raise ValueError(f"unknown fusion layout {fusion!r}")
    head = _head(p3, p4, True, len(rows))
    return {
        "nc": nc,
        "depth_multiple": gd,
        "width_multiple": gw,
        "anchors": deepcopy(anchors or ANCHORS),
        "backbone": rows,
        "head": head,
    }


def single_cft_config(size="s", nc=9):
    """BASELINE.json config 2 ("yolov5s + 1xCFT"): no such yaml exists in the reference; it is
    the add-fusion network with one GPT block on the P5 features (SURVEY.md section 8d recipe):
    after row 19 insert GPT([9,19]) and two Add2 rows; every later absolute index shifts by 3."""
    cfg = cft_config(size, "add", nc)
    rows = cfg["backbone"][:20]
    r, t = _fuse(rows, 9, 19, 1024)
    rows += [[[4, 14], 1, "Add", [1]], [[6, 16], 1, "Add", [1]], [[r, t], 1, "Add", [1]]]
    cfg["backbone"] = rows
    cfg["head"] = _head(23, 24, True, len(rows))
    return cfg


# name -> (size, fusion, nc) of every reference yaml under models/transformer/
REFERENCE_YAMLS = {
    "yolov5l_fusion_add_FLIR_aligned": ("l", "add", 3),
    "yolov5l_fusion_add_llvip": ("l", "add", 1),
    "yolov5l_fusion_transformer_FLIR": ("l", "transformer", 3),
    "yolov5l_fusion_transformer_FLIR_aligned": ("l", "transformer", 3),
    "yolov5l_fusion_transformer_llvip": ("l", "transformer", 1),
    "yolov5l_fusion_transformer_vedai": ("l", "transformer", 9),
    "yolov5l_fusion_transformerx3_FLIR_aligned": ("l", "transformerx3", 3),
    "yolov5l_fusion_transformerx3_llvip": ("l", "transformerx3", 1),
    "yolov5s_fusion_add_vedai": ("s", "add", 9),
    "yolov5s_fusion_transformer_vedai": ("s", "transformer", 9),
    "yolov5s_fusion_transformerx3_vedai": ("s", "transformerx3", 9),
    "yolov5x_fusion_transformer_FLIR": ("x", "transformer", 3),
    "yolov5x_fusion_transformer_FLIR_aligned": ("x", "transformer", 3),
}


def named_config(name):
    """BASELINE.json configs by short name."""
    if name in REFERENCE_YAMLS:
        size, fusion, nc = REFERENCE_YAMLS[name]
        return cft_config(size, fusion, nc)
    table = {
        "cfg1": lambda: cft_config("s", "add", 9),                 # yolov5s two-stream, no CFT
        "cfg2": lambda: single_cft_config("s", 9),                 # yolov5s + 1 CFT block
        "cfg3": lambda: cft_config("l", "transformerx3", 3),       # FLIR flagship
        "cfg4": lambda: cft_config("l", "transformerx3", 1),       # LLVIP
        "cfg5": lambda: cft_config("x", "transformerx3", 3),       # derived yolov5x x3
    }
    if name not in table:
        raise KeyError(name)
    return table[name]()
