"""MI355X-native two-stream YOLOv5 + CFT (Cross-Modality Fusion Transformer) inference forward.

    from msod_amd.models.yolo_test import Model
    from msod_amd.models.configs import named_config
    model = Model(named_config("cfg3")).cuda()
    pred, raw = model(rgb, ir)          # fp32 [B,3,H,W] image batches on the GPU

The compute path is the hand-written gfx950 kernels of ``libcft_hip.so`` (C ABI in
``include/cft_hip.h``); there is no CPU or PyTorch-op fallback.
"""
__version__ = "0.1.0"
