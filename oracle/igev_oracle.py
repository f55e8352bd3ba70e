"""CPU oracle for the IGEV pieces of the DiffuVolume hot path.  TEST INFRASTRUCTURE ONLY (same
rules as oracle/acv_oracle.py).  Pinned by tests/golden/igev_geo_lookup.npz, produced by the imported
reference KITTI15/core/geometry_ddim.py (oracle/make_golden.py).  The full IGEVStereo_ddim wrapper cannot
be constructed in the build container (timm pretrained backbone), so beyond these pieces IGEV parity is
unpinned."""
from __future__ import annotations

import torch
import torch.nn.functional as F


def all_pairs_corr(fmap1: torch.Tensor, fmap2: torch.Tensor) -> torch.Tensor:
    """geometry_ddim.py:72-80 -> [B,H,W1,W2]."""
    return torch.einsum("aijk,aijh->ajkh", fmap1, fmap2)


def _lerp_rows(rows: torch.Tensor, x: torch.Tensor) -> torch.Tensor:
    """rows [N,C,L], x [N,T] pixel positions -> [N,C,T]: bilinear_sampler + grid_sample(align_corners=True,
    zero padding) along the last axis (utils/utils.py:59-77), with the reference's normalise/unnormalise
    round trip in float32."""
    L = rows.shape[-1]
    xg = 2 * x / (L - 1) - 1
    ix = ((xg + 1) / 2) * (L - 1)
    fl = torch.floor(ix)
    i0 = fl.long()
    w0, w1 = (fl + 1) - ix, ix - fl

    def take(idx):
        ok = (idx >= 0) & (idx < L)
        v = torch.gather(rows, 2, idx.clamp(0, L - 1).unsqueeze(1).expand(-1, rows.shape[1], -1))
        return v * ok.unsqueeze(1)

    return take(i0) * w0.unsqueeze(1) + take(i0 + 1) * w1.unsqueeze(1)


def geo_filter_lookup(geo_volume, init_fmap1, init_fmap2, disp, coords, noisy, radius: int = 4, num_levels: int = 2):
    """Combined_Geo_Encoding_Volume(...)(disp, coords, noisy), geometry_ddim.py:7-69."""
    b, c, d, h, w = geo_volume.shape
    n = b * h * w
    geo = geo_volume.permute(0, 3, 4, 1, 2).reshape(n, c, d)                 # :19
    corr = all_pairs_corr(init_fmap1, init_fmap2).reshape(n, 1, -1)          # :21
    noi = noisy.reshape(n, 1, -1)                                            # :37 raw reshape (quirk)
    dx = torch.linspace(-radius, radius, 2 * radius + 1).view(1, -1)
    dflat, cflat = disp.reshape(n, 1), coords.reshape(n, 1)
    outs = []
    for i in range(num_levels):
        outs.append(_lerp_rows(geo * noi, dflat / 2 ** i + dx).reshape(n, -1))                   # :56-58
        outs.append(_lerp_rows(corr, cflat / 2 ** i - dflat / 2 ** i + dx).reshape(n, -1))      # :60-64
        geo = F.avg_pool1d(geo, 2, 2)
        corr = F.avg_pool1d(corr, 2, 2)
        noi = F.avg_pool1d(noi, 2, 2)
    out = torch.cat(outs, dim=-1).view(b, h, w, -1)
    return out.permute(0, 3, 1, 2).contiguous().float()
