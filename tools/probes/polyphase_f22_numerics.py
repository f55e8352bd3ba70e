"""fp32 error of the polyphase F(2,2) form of the k3 / s2 / p1 convolution and of the k3 / s2 / p1 / op1 transposed
convolution against the direct fp32 sum (CPU, torch emulation in the kernels' order of operations: weight sums formed in
fp64 and rounded once by the host pack, data differences and output sums in fp32).  Verdict r4 #1(a): the form is usable if
its error is <= 2x the direct sum's.     python tools/probes/polyphase_f22_numerics.py

Per axis a stride-2 3-tap filter reads, for the output pair (2t, 2t+1), the inputs r0..r4 = in[4t-1 .. 4t+3]:
    out[2t]   = w0 r0 + w1 r1 + w2 r2          out[2t+1] = w0 r2 + w1 r3 + w2 r4
The odd phase (r0, r2, r4) is a 2-tap filter (w0, w2) -> F(2,2): m1 = (r0-r2) w0, m2 = r2 (w0+w2), m3 = (r4-r2) w2; the even
phase is one tap.  Products that end in the same sum share an accumulator:
    A0 = (r0-r2) w0 + r1 w1      A1 = r2 (w0+w2)      A2 = r3 w1 + (r4-r2) w2      out[2t] = A0 + A1,  out[2t+1] = A1 + A2
5 multiplies per 2 outputs instead of 6; in-plane 25 instead of 36 per 2x2 outputs, into 9 accumulators.
"""
import torch

torch.manual_seed(0)
AMAP = [0, 0, 1, 2, 2]            # accumulator of A-operand i
BMAP = [0, 1, 2, 1, 3]            # weight combination of A-operand i: w0, w1, w0+w2, w2


def data_tf(p):                   # p [..., 5] -> [..., 5]  along the last axis
    return torch.stack([p[..., 0] - p[..., 2], p[..., 1], p[..., 2], p[..., 3], p[..., 4] - p[..., 2]], -1)


def weight_tf(w):                 # w [..., 3] -> [..., 4]
    return torch.stack([w[..., 0], w[..., 1], w[..., 0] + w[..., 2], w[..., 2]], -1)


def conv_s2_polyphase(x, w, dt):
    """x [C,H,W] (H, W multiples of 4), w [K,C,3,3]; k3 s2 p1 in-plane -> [K,H/2,W/2]."""
    C, H, W = x.shape
    K = w.shape[0]
    U = weight_tf(weight_tf(w.double()).transpose(-1, -2)).transpose(-1, -2).to(dt)       # [K,C,4,4] (by, bx)
    xp = torch.nn.functional.pad(x, (1, 3, 1, 3)).to(dt)
    p = xp.unfold(1, 5, 4).unfold(2, 5, 4)                                                  # [C,th,tw,5,5]
    V = data_tf(data_tf(p).transpose(-1, -2)).transpose(-1, -2)                             # (i, j)
    th, tw = p.shape[1], p.shape[2]
    acc = torch.zeros(K, th, tw, 3, 3, dtype=dt)
    for i in range(5):
        for j in range(5):
            acc[..., AMAP[i], AMAP[j]] += torch.einsum('kc,ctw->ktw', U[:, :, BMAP[i], BMAP[j]], V[..., i, j])
    y = torch.empty(K, th, tw, 2, 2, dtype=dt)
    for a in range(2):
        for b in range(2):
            y[..., a, b] = (acc[..., a, b] + acc[..., a, b + 1]) + (acc[..., a + 1, b] + acc[..., a + 1, b + 1])
    return y.permute(0, 1, 3, 2, 4).reshape(K, 2 * th, 2 * tw)


def deconv_s2_polyphase(x, w, dt):
    """x [C,H,W], w [C,K,3,3] (ConvTranspose layout); k3 s2 p1 op1 in-plane -> [K,2H,2W].
    Per axis, inputs d0, d1, d2 = in[2t .. 2t+2]:  out[4t] = w1 d0, out[4t+2] = w1 d1 (even class, one tap),
    out[4t+1] = w2 d0 + w0 d1, out[4t+3] = w2 d1 + w0 d2 (odd class, two taps -> F(2,2) with g = (w2, w0)):
    operands a = (d0, d0-d1, d1, d1, d2-d1), weights (w1, w2, w2+w0, w1, w0), accumulators E0, M1, M2, E1, M3;
    out = E0, M1+M2, E1, M2+M3."""
    C, H, W = x.shape
    K = w.shape[1]

    def dtf(p):
        return torch.stack([p[..., 0], p[..., 0] - p[..., 1], p[..., 1], p[..., 1], p[..., 2] - p[..., 1]], -1)

    def wtf(g):
        return torch.stack([g[..., 1], g[..., 2], g[..., 2] + g[..., 0], g[..., 1], g[..., 0]], -1)

    U = wtf(wtf(w.double()).transpose(-1, -2)).transpose(-1, -2).to(dt)                   # [C,K,5,5]
    xp = torch.nn.functional.pad(x, (0, 2, 0, 2)).to(dt)
    p = xp.unfold(1, 3, 2).unfold(2, 3, 2)                                                  # [C,th,tw,3,3]
    V = dtf(dtf(p).transpose(-1, -2)).transpose(-1, -2)                                     # [C,th,tw,5,5]
    M = torch.einsum('ckij,ctwij->ktwij', U, V)

    def otf(m):                                                                             # [...,5] -> [...,4]
        return torch.stack([m[..., 0], m[..., 1] + m[..., 2], m[..., 3], m[..., 2] + m[..., 4]], -1)

    y = otf(otf(M).transpose(-1, -2)).transpose(-1, -2)                                     # [K,th,tw,4,4]
    th, tw = y.shape[1], y.shape[2]
    return y.permute(0, 1, 3, 2, 4).reshape(K, 4 * th, 4 * tw)


def report(name, y, ref):
    e = (y.double() - ref).abs()
    s = ref.abs().max()
    print(f'{name:34s} max {e.max() / s:.2e}  rms {e.pow(2).mean().sqrt() / ref.pow(2).mean().sqrt():.2e}')
    return float(e.max() / s)


if __name__ == '__main__':
    C, K, H, W = 32, 64, 64, 96
    x = torch.relu(torch.randn(C, H, W))
    w = torch.randn(K, C, 3, 3) * (2 / (9 * C)) ** .5
    ref = torch.nn.functional.conv2d(x.double()[None], w.double(), stride=2, padding=1)[0]
    assert (conv_s2_polyphase(x, w, torch.float64) - ref).abs().max() < 1e-12       # the algebra
    print('k3 s2 p1 convolution, one 32 -> 64 plane')
    d = report('direct f32', torch.nn.functional.conv2d(x[None], w, stride=2, padding=1)[0], ref)
    p = report('polyphase F(2,2) f32', conv_s2_polyphase(x, w, torch.float32), ref)
    print(f'  ratio {p / d:.2f} (bar: <= 2)')

    wt = torch.randn(C, K, 3, 3) * (2 / (9 * C)) ** .5
    ref = torch.nn.functional.conv_transpose2d(x.double()[None], wt.double(), stride=2, padding=1, output_padding=1)[0]
    assert (deconv_s2_polyphase(x, wt, torch.float64) - ref).abs().max() < 1e-12
    print('k3 s2 p1 op1 transposed convolution, one 32 -> 64 plane')
    d = report('direct f32', torch.nn.functional.conv_transpose2d(x[None], wt, stride=2, padding=1, output_padding=1)[0], ref)
    p = report('polyphase F(2,2) f32', deconv_s2_polyphase(x, wt, torch.float32), ref)
    print(f'  ratio {p / d:.2f} (bar: <= 2)')
