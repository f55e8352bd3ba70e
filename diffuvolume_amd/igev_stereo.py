"""The origin IGEV-Stereo network (KITTI15/core/igev_stereo.py:91-221) -- the network whose output is the `flow_pr`
that KITTI15/evaluate_stereo.py:88-98 hands to `IGEVStereo_ddim` -- on the same pieces as the DiffuVolume flavour:
same modules and parameter names (the reference's 509-key ``state_dict`` loads with strict=True: no time embedding, no
schedule buffers), the cost-volume front and the update block on the HIP kernels, and the geometry lookup of
core/geometry.py (no noise filter; `coords` are the x coordinates) through the same HIP lookup with a unit filter.

``IGEVStereo(args).forward(image1, image2, iters=12, flow_init=None, test_mode=False)``: eval only;
``test_mode=True`` -> the full-resolution disparity after the last iteration [B,1,H,W] (:216-217),
``test_mode=False`` -> ``(init_disp [B,1,H,W], [disp_up per iteration])`` (:219-220)."""
from __future__ import annotations

from typing import Optional

import torch
import torch.nn.functional as F
from torch import nn

from . import _lib
from .igev_stereo_ddim import IGEVStereo_ddim, context_upsample, hip_sequential

_SCHEDULE = ("betas", "alphas_cumprod", "alphas_cumprod_prev", "sqrt_alphas_cumprod", "sqrt_one_minus_alphas_cumprod",
             "log_one_minus_alphas_cumprod", "sqrt_recip_alphas_cumprod", "sqrt_recipm1_alphas_cumprod",
             "posterior_variance", "posterior_log_variance_clipped", "posterior_mean_coef1", "posterior_mean_coef2")


class IGEVStereo(IGEVStereo_ddim):
    def __init__(self, args, feature: Optional[nn.Module] = None, cnet: Optional[nn.Module] = None):
        super().__init__(args, feature=feature, cnet=cnet)
        del self.time_embedding                          # igev_stereo.py:91-135 has neither the time MLP ...
        for name in _SCHEDULE:                           # ... nor the diffusion schedule
            del self._buffers[name]

    def forward(self, image1, image2, iters=12, flow_init=None, test_mode=False):
        if self.training:
            raise NotImplementedError("the MI355X path is inference-only (model.eval())")
        with torch.no_grad():
            features_left, stem_2x, init_disp, net_list, inp_list, geo_fn = self._front(image1, image2)
            spx_pred = None
            if not test_mode:                            # :187-191 (2-D InstanceNorm heads, once per pair: PyTorch)
                spx_pred = F.softmax(hip_sequential(self.spx, self.spx_2(hip_sequential(self.spx_4, features_left[0]), stem_2x)), 1)
            b, _, h, w = init_disp.shape
            coords = torch.arange(w, dtype=torch.float32, device=init_disp.device).view(1, 1, 1, w).expand(b, 1, h, w).contiguous()
            unit = torch.ones((b, self.args.max_disp // 4, h, w), dtype=torch.float32, device=init_disp.device)
            n, slow = self.args.n_gru_layers, self.args.slow_fast_gru
            disp, disp_preds, disp_up = init_disp, [], None
            for itr in range(iters):                     # :203-214
                geo_feat = geo_fn(disp, coords, unit)
                if n == 3 and slow:
                    net_list = self.update_block(net_list, inp_list, iter16=True, iter08=False, iter04=False, update=False)
                if n >= 2 and slow:
                    net_list = self.update_block(net_list, inp_list, iter16=n == 3, iter08=True, iter04=False, update=False)
                last = itr == iters - 1
                net_list, mask_feat_4, delta_disp = self.update_block(net_list, inp_list, geo_feat, disp, iter16=n == 3,
                                                                      iter08=n >= 2, mask=last or not test_mode)
                disp = disp + delta_disp
                if test_mode and not last:
                    continue
                disp_up = self.upsample_disp(disp, mask_feat_4, stem_2x)
                disp_preds.append(disp_up)
            if test_mode:
                return disp_up
            return context_upsample(init_disp, spx_pred.float(), scale=4.0).unsqueeze(1), disp_preds
