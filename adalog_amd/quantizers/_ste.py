"""Straight-through rounding helpers (reference quantizers/_ste.py:5-14).

Used by the BRECQ stage only: the forward value is the rounded one, the gradient is the identity.
"""
import torch


def round_ste(x: torch.Tensor):
    return x + (x.round() - x).detach()


def floor_ste(x: torch.Tensor):
    return x + (x.floor() - x).detach()


def ceil_ste(x: torch.Tensor):
    return x + (x.ceil() - x).detach()
