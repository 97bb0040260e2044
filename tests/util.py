"""Shared helpers for the parity tests (layout conversion, error metrics)."""
import torch


def to_cn(x):
    """(B,C,T,V) -> CN matrix [C][B*T*V] (include/sar_hip.h)."""
    B, C, T, V = x.shape
    return x.permute(1, 0, 2, 3).reshape(C, B * T * V).contiguous()


def from_cn(y, B, T, V):
    C = y.shape[0]
    return y.reshape(C, B, T, V).permute(1, 0, 2, 3).contiguous()


def rel_err(a, b):
    """max |a-b| / max |b| (norm-wise relative error; b is the reference)."""
    a = a.detach().double().cpu()
    b = b.detach().double().cpu()
    denom = b.abs().max().item()
    return (a - b).abs().max().item() / (denom if denom > 0 else 1.0)


def rel_err_fro(a, b):
    """||a-b||_2 / ||b||_2 (Frobenius-relative error; b is the reference)."""
    a = a.detach().double().cpu()
    b = b.detach().double().cpu()
    denom = b.norm().item()
    return (a - b).norm().item() / (denom if denom > 0 else 1.0)
