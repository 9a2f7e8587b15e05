"""Shared helpers of the parity tests: tensor layout conversion and error metrics."""
import torch


def to_dev_nhwc(x, dev, dtype):
    """CPU NCHW fp32 -> device tensor, logical NCHW / NHWC memory, in ``dtype``."""
    B, C, H, W = x.shape
    y = torch.empty((B, H, W, C), dtype=dtype, device=dev).permute(0, 3, 1, 2)
    y.copy_(x.to(dev))
    return y


def to_cpu_f32(y):
    return y.detach().float().cpu().contiguous()


def rel_err(a, b):
    """max |a-b| / (max|b| + tiny): error relative to the tensor's scale."""
    a, b = a.float(), b.float()
    return ((a - b).abs().max() / (b.abs().max() + 1e-12)).item()


def bf16_round(x):
    return x.to(torch.bfloat16).float()


def tol(dtype):
    """Relative-to-scale tolerance of one kernel: fp32 = accumulation-order noise; bf16 = one
    output rounding (2^-8) with inputs pre-rounded to bf16 on the oracle side."""
    return {torch.float32: 2e-5, torch.bfloat16: 6e-3, torch.float16: 8e-4}[dtype]
