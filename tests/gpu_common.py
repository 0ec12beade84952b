"""helpers for the GPU parity tests"""
import numpy as np
import torch


def dev():
    return torch.device("cuda:0")


def t(a, dtype=None):
    x = torch.from_numpy(np.ascontiguousarray(a)).to(dev())
    return x.to(dtype) if dtype is not None else x


def n(x):
    return x.detach().float().cpu().numpy() if x.dtype == torch.bfloat16 else x.detach().cpu().numpy()


def bf16_round(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(torch.bfloat16).float().numpy()


def rel_err(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.max(np.abs(a - b)) / (np.max(np.abs(b)) + 1e-30))
