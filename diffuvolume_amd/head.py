"""Time embedding of the DiffuVolume filter (reference: SceneFlow/models/head.py:22-82).

``DynamicHead(d_model)``: sinusoidal(d_model) -> Linear(d, 4d) -> GELU -> Linear(4d, 4d)
-> SiLU -> Linear(4d, d), added to the noisy state as a per-(batch, channel) shift.
56 k parameters, < 1 MFLOP per step: the MLP stays in PyTorch (it runs once per DDIM step
on a [B] tensor); the add is fused into the HIP noise-prepare kernel, so ``shift()`` is
what the hot path calls.  Parameter names match the reference state_dict
(``time_mlp.1/.3``, ``block_time_mlp.1``).
"""
from __future__ import annotations

import math

import torch
from torch import nn


class SinusoidalPositionEmbeddings(nn.Module):
    def __init__(self, dim: int):
        super().__init__()
        self.dim = dim

    def forward(self, time: torch.Tensor) -> torch.Tensor:
        half = self.dim // 2
        freq = torch.exp(torch.arange(half, device=time.device) * -(math.log(10000) / (half - 1)))
        ang = time[:, None] * freq[None, :]
        return torch.cat((ang.sin(), ang.cos()), dim=-1)


class DynamicHead(nn.Module):
    def __init__(self, d_model: int):
        super().__init__()
        self.d_model = d_model
        width = d_model * 4
        self.time_mlp = nn.Sequential(SinusoidalPositionEmbeddings(d_model), nn.Linear(d_model, width),
                                      nn.GELU(), nn.Linear(width, width))
        self.block_time_mlp = nn.Sequential(nn.SiLU(), nn.Linear(width, d_model))
        for p in self.parameters():
            if p.dim() > 1:
                nn.init.xavier_uniform_(p)

    def shift(self, t: torch.Tensor) -> torch.Tensor:
        """[B] integer timesteps -> [B, d_model] fp32 shift."""
        return self.block_time_mlp(self.time_mlp(t))

    def forward(self, noisy: torch.Tensor, t: torch.Tensor) -> torch.Tensor:
        return noisy + self.shift(t).unsqueeze(-1).unsqueeze(-1)
