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


# ---------------------------------------------------------------------------------------------------
# DDIM loop of IGEVStereo_ddim (KITTI15/core/igev_stereo_ddim.py:226-359), pinned by tests/golden/igev_loop.npz
# (vectors produced by the reference's own two methods bound to a light object, oracle/make_golden_igev_loop.py)
# ---------------------------------------------------------------------------------------------------
def head180_shift(t, sd, bins: int = 48):
    """core/head.py:22-34,:74-77: sinusoidal(180) MLP, then linear interpolation of the 180-vector to `bins`."""
    import math
    half = 90
    freq = torch.exp(torch.arange(half) * -(math.log(10000) / (half - 1)))
    emb = t[:, None] * freq[None, :]
    emb = torch.cat((emb.sin(), emb.cos()), dim=-1)
    y = F.linear(emb, sd["time_mlp.1.weight"], sd["time_mlp.1.bias"])
    y = F.linear(F.gelu(y), sd["time_mlp.3.weight"], sd["time_mlp.3.bias"])
    y = F.linear(F.silu(y), sd["block_time_mlp.1.weight"], sd["block_time_mlp.1.bias"])
    return F.interpolate(y.unsqueeze(1), bins, mode="linear").squeeze(1)


class IGEVLoopOracle:
    def __init__(self, head_sd, update_block, upsample_disp, geo, f1, f2, sampling_timesteps=2, cof=(0.6, 0.1, 0.3),
                 net_list=None, inp_list=None):
        from . import acv_oracle as A
        # hidden states / context terms of the update block; the reference mutates `net_list` in place, so the
        # hidden state carries over from one DDIM step to the next (igev_stereo_ddim.py:240-250, :318)
        self.net_list = [None] if net_list is None else list(net_list)
        self.inp_list = [None] if inp_list is None else inp_list
        self.A, self.head_sd, self.update_block, self.upsample_disp = A, head_sd, update_block, upsample_disp
        self.geo, self.f1, self.f2 = geo, f1, f2
        self.S, self.cof = sampling_timesteps, cof
        ac = A.cosine_alphas_cumprod(1000)
        self.ac, self.sqrt_ac, self.sqrt_1m = ac, torch.sqrt(ac), torch.sqrt(1 - ac)
        self.sr, self.srm1 = torch.sqrt(1 / ac), torch.sqrt(1 / ac - 1)

    def corr_fn(self, disp, coords, noisy):
        return geo_filter_lookup(self.geo, self.f1, self.f2, disp, coords, noisy)

    def model_predictions(self, coords0, coords1, iters, x_t, t):
        """:226-292 (n_gru_layers=3, slow_fast_gru=False, flow_init=None)."""
        A = self.A
        n01 = ((torch.clamp(x_t + head180_shift(t, self.head_sd)[:, :, None, None], -1, 1)) + 1) / 2
        nets = self.net_list
        for itr in range(iters):
            flow = coords1 - coords0
            corr = self.corr_fn(flow, coords1, n01.float())
            nets, up_mask, delta = self.update_block(nets, self.inp_list, corr, flow, iter16=True, iter08=True)
            coords1 = coords1 + delta
        self.net_list = nets
        pred = self.upsample_disp(coords1 - coords0, up_mask, None)[:, :1]
        b, _, hh, ww = pred.shape
        dn = F.interpolate(torch.clamp(pred, 0, 47), size=(hh // 4, ww // 4), mode="bilinear") / 4
        tc = torch.clamp(coords0 + dn, 0, 47)
        x_start = torch.clamp(A.encode_two_hot(tc, 48) * 2 - 1.0, -1, 1)
        bs = (b, 1, 1, 1)
        pn = (self.sr.gather(-1, t).reshape(bs) * n01 - x_start) / self.srm1.gather(-1, t).reshape(bs)
        return pn, x_start, pred, coords1

    def ddim_sample(self, coords0, coords1, iters, used, asd, draw, trace=None):
        """:294-359.  ``trace`` (a list) receives one dict per step: the state entering the step (img, mask_in,
        coords1_in, nets_in), its outputs (disp, x_start, coords1_out, nets_out, mask_out, img_next) and the draws."""
        b, d, h, w = asd.shape
        img = draw("x_T", tuple(asd.shape), asd.dtype)
        final = [used]
        mask = torch.zeros(b, h, w)
        times = list(reversed(torch.linspace(-1, 999, steps=self.S + 1).int().tolist()))
        for time, time_next in zip(times[:-1], times[1:]):
            t = torch.full((b,), time, dtype=torch.long)
            rec = None
            if trace is not None:
                rec = {"time": time, "time_next": time_next, "img": img, "mask_in": mask, "coords1_in": coords1,
                       "nets_in": list(self.net_list), "eps": None, "fill": None, "img_next": None}
                trace.append(rec)
            pn, x_start, disp, coords1 = self.model_predictions(coords0, coords1, iters, img, t)
            dif = torch.abs(disp - used)
            keep = F.interpolate((dif < 5).float(), size=(h, w), mode="bilinear").squeeze(1)
            mask = torch.clamp(mask + keep, 0, 1)
            final.append(torch.where(dif < 3, disp, used))
            if rec is not None:
                rec.update(disp=disp, x_start=x_start, coords1_out=coords1, nets_out=list(self.net_list), mask_out=mask,
                           out=final[-1])
            if time_next < 0:
                img = x_start
                continue
            a, an = self.ac[time], self.ac[time_next]
            sigma = ((1 - a / an) * (1 - an) / (1 - a)).sqrt()
            c = (1 - an - sigma ** 2).sqrt()
            eps = draw("eps", tuple(img.shape), img.dtype)
            img = x_start * an.sqrt() + c * pn + sigma * eps
            tt = torch.full((1,), time, dtype=torch.long)
            fill = (self.sqrt_ac.gather(-1, tt).reshape(1, 1, 1, 1) * asd
                    + self.sqrt_1m.gather(-1, tt).reshape(1, 1, 1, 1) * draw("q", tuple(asd.shape), asd.dtype))
            img = torch.where(mask.unsqueeze(1) == 0, fill, img)
            if rec is not None:
                rec.update(eps=eps, fill=fill, img_next=img)
        stack = torch.cat(final, dim=1)                                  # [B, S+1, H, W]
        return (stack * torch.tensor(self.cof).view(1, -1, 1, 1)).sum(dim=1)


# ---------------------------------------------------------------------------------------------------
# Cost-volume front of IGEVStereo_ddim.forward (KITTI15/core/igev_stereo_ddim.py:377-386): a functional
# restatement over a flat state_dict, eval-mode BatchNorm.  Test infrastructure only.
# ---------------------------------------------------------------------------------------------------
def basic_conv(x, sd, p, *, is_3d, deconv=False, bn=True, relu=True, stride=1, padding=0):
    """BasicConv.forward (core/submodule.py:29-35): conv(bias=False) -> [BN eval] -> [LeakyReLU(0.01)]."""
    w = sd[p + ".conv.weight"]
    if is_3d:
        x = F.conv_transpose3d(x, w, None, stride, padding) if deconv else F.conv3d(x, w, None, stride, padding)
    else:
        x = F.conv2d(x, w, None, stride, padding)
    if bn:
        x = F.batch_norm(x, sd[p + ".bn.running_mean"], sd[p + ".bn.running_var"], sd[p + ".bn.weight"],
                         sd[p + ".bn.bias"], False, 0.0, 1e-5)
    return F.leaky_relu(x, 0.01) if relu else x


def feature_att(cv, feat, sd, p):
    """FeatureAtt.forward (core/submodule.py:234-239)."""
    a = basic_conv(feat, sd, p + ".feat_att.0", is_3d=False)
    a = F.conv2d(a, sd[p + ".feat_att.1.weight"], sd[p + ".feat_att.1.bias"])
    return torch.sigmoid(a).unsqueeze(2) * cv


def igev_hourglass(x, features, sd, p="cost_agg"):
    """hourglass.forward (igev_stereo_ddim.py:67-91)."""
    def pair(t, name, stride):
        t = basic_conv(t, sd, f"{p}.{name}.0", is_3d=True, stride=stride, padding=1)
        return basic_conv(t, sd, f"{p}.{name}.1", is_3d=True, stride=1, padding=1)

    def agg(t, name):
        t = basic_conv(t, sd, f"{p}.{name}.0", is_3d=True)
        t = basic_conv(t, sd, f"{p}.{name}.1", is_3d=True, padding=1)
        return basic_conv(t, sd, f"{p}.{name}.2", is_3d=True, padding=1)

    def up(t, name, bn=True, relu=True):
        return basic_conv(t, sd, f"{p}.{name}", is_3d=True, deconv=True, bn=bn, relu=relu, stride=2, padding=1)

    conv1 = feature_att(pair(x, "conv1", 2), features[1], sd, p + ".feature_att_8")
    conv2 = feature_att(pair(conv1, "conv2", 2), features[2], sd, p + ".feature_att_16")
    conv3 = feature_att(pair(conv2, "conv3", 2), features[3], sd, p + ".feature_att_32")
    conv2 = agg(torch.cat((up(conv3, "conv3_up"), conv2), dim=1), "agg_0")
    conv2 = feature_att(conv2, features[2], sd, p + ".feature_att_up_16")
    conv1 = agg(torch.cat((up(conv2, "conv2_up"), conv1), dim=1), "agg_1")
    conv1 = feature_att(conv1, features[1], sd, p + ".feature_att_up_8")
    return up(conv1, "conv1_up", bn=False, relu=False)


def igev_cost_volume(match_left, match_right, features_left, sd, max_disp=192):
    """igev_stereo_ddim.py:377-383: (geo_encoding_volume [B,8,D/4,h,w], init_disp [B,1,h,w])."""
    from .acv_oracle import build_gwc_volume, disparity_regression
    d4 = max_disp // 4
    gwc = build_gwc_volume(match_left, match_right, d4, 8)
    gwc = basic_conv(gwc, sd, "corr_stem", is_3d=True, padding=1)
    gwc = feature_att(gwc, features_left[0], sd, "corr_feature_att")
    geo = igev_hourglass(gwc, features_left, sd, "cost_agg")
    prob = F.softmax(F.conv3d(geo, sd["classifier.weight"], None, 1, 1).squeeze(1), dim=1)
    return geo, disparity_regression(prob, d4, keepdim=True)


# ---------------------------------------------------------------------------------------------------
# IGEV's recurrent update block (KITTI15/core/update.py), functional restatement over a flat state_dict.
# ---------------------------------------------------------------------------------------------------
def _conv(x, sd, p, pad):
    return F.conv2d(x, sd[p + ".weight"], sd[p + ".bias"], 1, pad)


def conv_gru(h, cz, cr, cq, xs, sd, p):
    """ConvGRU.forward (update.py:33-40)."""
    x = torch.cat(xs, dim=1)
    hx = torch.cat([h, x], dim=1)
    z = torch.sigmoid(_conv(hx, sd, p + ".convz", 1) + cz)
    r = torch.sigmoid(_conv(hx, sd, p + ".convr", 1) + cr)
    q = torch.tanh(_conv(torch.cat([r * h, x], dim=1), sd, p + ".convq", 1) + cq)
    return (1 - z) * h + z * q


def motion_encoder(disp, corr, sd, p="encoder"):
    """BasicMotionEncoder.forward (update.py:85-94)."""
    cor = F.relu(_conv(corr, sd, p + ".convc1", 0))
    cor = F.relu(_conv(cor, sd, p + ".convc2", 1))
    d = F.relu(_conv(disp, sd, p + ".convd1", 3))
    d = F.relu(_conv(d, sd, p + ".convd2", 1))
    out = F.relu(_conv(torch.cat([cor, d], dim=1), sd, p + ".conv", 1))
    return torch.cat([out, disp], dim=1)


def update_block(sd, net, inp, corr, disp, n_gru_layers=3, iter04=True, iter08=True, iter16=True, update=True):
    """BasicMultiUpdateBlock.forward (update.py:123-142); returns a new `net` list (the reference mutates its)."""
    net = list(net)
    pool2x = lambda t: F.avg_pool2d(t, 3, stride=2, padding=1)
    interp = lambda t, dest: F.interpolate(t, dest.shape[2:], mode="bilinear", align_corners=True)
    if iter16:
        net[2] = conv_gru(net[2], *inp[2], [pool2x(net[1])], sd, "gru16")
    if iter08:
        xs = [pool2x(net[0]), interp(net[2], net[1])] if n_gru_layers > 2 else [pool2x(net[0])]
        net[1] = conv_gru(net[1], *inp[1], xs, sd, "gru08")
    if iter04:
        mf = motion_encoder(disp, corr, sd)
        xs = [mf, interp(net[1], net[0])] if n_gru_layers > 1 else [mf]
        net[0] = conv_gru(net[0], *inp[0], xs, sd, "gru04")
    if not update:
        return net
    delta = _conv(F.relu(_conv(net[0], sd, "disp_head.conv1", 1)), sd, "disp_head.conv2", 1)
    mask = F.relu(_conv(net[0], sd, "mask_feat_4.0", 1))
    return net, mask, delta


def context_upsample(disp_low, up_weights):
    """core/submodule.py:241-253: 3x3 neighbourhood of the low-resolution disparity (zero padded), repeated x4 by
    nearest neighbour, weighted by the 9 per-pixel weights.  disp_low [B,1,h,w], up_weights [B,9,4h,4w] -> [B,4h,4w]."""
    b, c, h, w = disp_low.shape
    nb = F.unfold(disp_low.reshape(b, c, h, w), 3, 1, 1).reshape(b, -1, h, w)
    nb = F.interpolate(nb, (h * 4, w * 4), mode="nearest").reshape(b, 9, h * 4, w * 4)
    return (nb * up_weights).sum(1)


def upsample_disp(disp, logits):
    """IGEVStereo_ddim.upsample_disp after its two spx convolutions (igev_stereo_ddim.py:213-215)."""
    return context_upsample(disp * 4.0, F.softmax(logits, 1)).unsqueeze(1)
