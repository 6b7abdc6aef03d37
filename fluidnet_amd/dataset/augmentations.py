"""Flip augmentation with the reference's API (ml/dataset/augmentations.py:4-35): horizontal (width-axis) flip with
probability p and negation of the x-velocity channel; density_{t-1}, masks and vz are only mirrored."""
from __future__ import annotations

from typing import Tuple

import torch


def apply_horizontal_flip(x: torch.Tensor, y: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    xf, yf = torch.flip(x, dims=[2]), torch.flip(y, dims=[2])
    xf[1] = -xf[1]
    yf[1] = -yf[1]
    return xf, yf


def apply_augmentation(x: torch.Tensor, y: torch.Tensor, flip_prob: float = 0.5) -> Tuple[torch.Tensor, torch.Tensor]:
    if torch.rand(1).item() < flip_prob:
        return apply_horizontal_flip(x, y)
    return x, y


def apply_rollout_augmentation(x_0: torch.Tensor, y_seq: torch.Tensor, masks: torch.Tensor,
                               flip_prob: float = 0.5) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    if torch.rand(1).item() < flip_prob:
        x_0 = torch.flip(x_0, dims=[2])        # (6,H,W)
        y_seq = torch.flip(y_seq, dims=[3])    # (K,3,H,W)
        masks = torch.flip(masks, dims=[3])    # (K|K+1,2,H,W)
        x_0[1] = -x_0[1]
        y_seq[:, 1, :, :] = -y_seq[:, 1, :, :]
    return x_0, y_seq, masks
