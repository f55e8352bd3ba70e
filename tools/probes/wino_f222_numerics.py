"""fp32 error of the 3-D Winograd form F(2x2x2, 3x3x3) of the 3x3x3 stride-1 convolution against the direct fp32 sum and
against the shipped form (F(2x2, 3x3) in-plane, depth taps direct), CPU torch emulation: weights transformed in fp64 and
rounded once (the host pack), data transform / products / output transform in fp32.  python tools/probes/wino_f222_numerics.py"""
import torch

torch.manual_seed(0)
BT = torch.tensor([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=torch.float64)
G = torch.tensor([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], dtype=torch.float64)
AT = torch.tensor([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=torch.float64)


def tf(x, m, axis):                 # apply matrix m along `axis`
    return torch.movedim(torch.tensordot(m.to(x.dtype), torch.movedim(x, axis, 0), dims=([1], [0])), 0, axis)


def conv_wino(x, w, dims, dt):
    """x [C,D,H,W] (even sizes), w [K,C,3,3,3]; Winograd along the axes in `dims` (subset of (1,2,3)), direct along the others."""
    C = x.shape[0]
    K = w.shape[0]
    xp = torch.nn.functional.pad(x, (1, 1, 1, 1, 1, 1)).to(dt)
    U = w.double()
    for a in dims:
        U = tf(U, G, a + 1)                        # weight axes: K,C,kd,ky,kx
    U = U.to(dt)
    # patches: winograd axes unfold (4, step 2); direct axes unfold (3, step 1)
    p = xp
    for a in (1, 2, 3):
        p = p.unfold(a, 4, 2) if a in dims else p.unfold(a, 3, 1)
    # p [C, nd, nh, nw, pd, ph, pw]
    V = p
    for a in dims:
        V = tf(V, BT, 3 + a)
    M = torch.einsum('kcdef,cxyzdef->kxyzdef', U, V) if False else None
    # positions: multiply elementwise over Winograd axes, sum over direct axes and C
    sub = 'kc' + 'def'
    out = None
    # build einsum: winograd axes keep their position index, direct axes are summed
    keep = ''.join('def'[a - 1] for a in dims)
    M = torch.einsum(f'kcdef,cxyzdef->kxyz{keep}', U, V)
    for i, a in enumerate(dims):
        M = tf(M, AT, 4 + i)
    # M [K, nd, nh, nw, (2 per winograd axis)] -> interleave
    nd, nh, nw = M.shape[1:4]
    sizes = [2 if a in dims else 1 for a in (1, 2, 3)]
    M = M.reshape(K, nd, nh, nw, *sizes)
    M = M.permute(0, 1, 4, 2, 5, 3, 6).reshape(K, nd * sizes[0], nh * sizes[1], nw * sizes[2])
    return M


C, K, D, H, W = 32, 32, 8, 16, 16
for trial in range(3):
    x = torch.randn(C, D, H, W)
    w = torch.randn(K, C, 3, 3, 3) * (2.0 / (27 * C)) ** 0.5
    ref = torch.nn.functional.conv3d(x.double()[None], w.double(), padding=1)[0]
    scale = ref.abs().max().item()
    direct = torch.nn.functional.conv3d(x[None], w, padding=1)[0].double()
    w22 = conv_wino(x, w, (2, 3), torch.float32).double()
    w222 = conv_wino(x, w, (1, 2, 3), torch.float32).double()
    chk = conv_wino(x.double(), w.double(), (1, 2, 3), torch.float64)
    assert (chk - ref).abs().max() < 1e-10 * scale
    e = lambda y: ((y - ref).abs().max().item() / scale, ((y - ref) ** 2).mean().sqrt().item() / scale)
    print(f"trial {trial}: max|err|/scale, rms/scale   direct fp32 {e(direct)[0]:.2e} {e(direct)[1]:.2e}   "
          f"F(2x2)+depth direct {e(w22)[0]:.2e} {e(w22)[1]:.2e}   F(2x2x2) {e(w222)[0]:.2e} {e(w222)[1]:.2e}")
