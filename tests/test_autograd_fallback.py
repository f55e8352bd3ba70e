"""the builders are differentiable in the reference (training uses them); when autograd is recording
and an input requires grad they run a differentiable PyTorch statement instead of the HIP kernel.  Forward values and
gradients are checked against the oracle (itself pinned to the reference's goldens) on CPU."""
import pytest
import torch

import diffuvolume_amd as dv
from diffuvolume_amd import submodule as S
from diffuvolume_amd.synth import _gen
from oracle import acv_oracle as O


def _pair(shape, seed):
    b, c, h, w = shape
    L = torch.randn(b, c, h, w, generator=_gen(seed, "L"), requires_grad=True)
    R = torch.randn(b, c, h, w, generator=_gen(seed, "R"), requires_grad=True)
    return L, R


def _grads(fn, L, R, weight):
    L.grad = R.grad = None
    out = fn(L, R)
    (out * weight).sum().backward()
    return out.detach(), L.grad.clone(), R.grad.clone()


@pytest.mark.parametrize("shape,d,g", [((2, 16, 3, 10), 4, 4), ((1, 24, 2, 7), 9, 8), ((1, 320, 2, 12), 6, 40)])
def test_gwc_volume_values_and_gradients(shape, d, g):
    L, R = _pair(shape, 11)
    w = torch.randn(shape[0], g, d, shape[2], shape[3], generator=_gen(11, "w"))
    out, gl, gr = _grads(lambda a, b: dv.build_gwc_volume(a, b, d, g), L, R, w)
    ref, rl, rr = _grads(lambda a, b: O.build_gwc_volume(a, b, d, g), L, R, w)
    assert out.is_contiguous() and out.shape == ref.shape
    torch.testing.assert_close(out, ref, atol=1e-6, rtol=1e-6)
    assert float(out[:, :, min(d - 1, 3), :, :min(d - 1, 3)].abs().max()) == 0.0      # exact zeros where x < d
    torch.testing.assert_close(gl, rl, atol=1e-6, rtol=1e-6)
    torch.testing.assert_close(gr, rr, atol=1e-6, rtol=1e-6)


@pytest.mark.parametrize("zero_left", [False, True])
def test_concat_volume_values_and_gradients(zero_left):
    L, R = _pair((2, 6, 3, 11), 12)
    d = 5
    w = torch.randn(2, 12, d, 3, 11, generator=_gen(12, "w"))
    out, gl, gr = _grads(lambda a, b: dv.build_concat_volume(a, b, d, zero_left=zero_left), L, R, w)
    ref, rl, rr = _grads(lambda a, b: O.build_concat_volume(a, b, d, zero_left=zero_left), L, R, w)
    assert torch.equal(out, ref) and out.is_contiguous()
    torch.testing.assert_close(gl, rl, atol=2e-6, rtol=1e-6)        # sum over D in a different order
    torch.testing.assert_close(gr, rr, atol=2e-6, rtol=1e-6)


def test_disparity_regression_gradient():
    x = torch.softmax(torch.randn(2, 12, 4, 5, generator=_gen(13, "x")), dim=1).requires_grad_(True)
    out = dv.disparity_regression(x, 12)
    out.sum().backward()
    ref = O.disparity_regression(x.detach(), 12)
    torch.testing.assert_close(out.detach(), ref, atol=1e-6, rtol=1e-6)
    torch.testing.assert_close(x.grad, torch.arange(12.0).view(1, 12, 1, 1).expand_as(x))
    assert S.disparity_regression(x, 12, keepdim=True).shape == (2, 1, 4, 5)


def test_without_grad_the_cpu_is_still_refused():
    """No requires_grad -> the HIP path, which has no CPU fallback."""
    L, R = (t.detach() for t in _pair((1, 8, 2, 6), 14))
    with pytest.raises(dv._lib.DiffuVolumeError):
        dv.build_gwc_volume(L, R, 3, 4)
    with torch.no_grad(), pytest.raises(dv._lib.DiffuVolumeError):
        dv.build_gwc_volume(L.requires_grad_(True), R, 3, 4)
