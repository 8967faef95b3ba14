"""Fused HIP L1 + (1-SSIM) loss (csrc/gs_loss.hip) against the plain-torch restatement in fp64."""
import numpy as np
import pytest
import torch

from easy_gaussian_splatting_amd.loss import LossComputer

pytestmark = pytest.mark.gpu


# (129 x 257: 5 x 9 = 45 tiles, not a multiple of the eight XCD runs the blocks are dealt over; 200 x 333: ragged right and bottom tiles)
@pytest.mark.parametrize("H,W,use_mask", [(75, 100, False), (64, 96, True), (33, 45, True), (129, 257, True), (200, 333, False),
                                          (1080, 1920, False), (1080, 1920, True)])
def test_fused_loss_matches_torch(H, W, use_mask):
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(H * W)
    lowres = torch.rand(H // 4 + 1, W // 4 + 1, 3, generator=g, dtype=torch.float64)
    up = lambda t: torch.nn.functional.interpolate(t.permute(2, 0, 1)[None], size=(H, W), mode="bilinear")[0].permute(1, 2, 0)
    gt = up(lowres).contiguous()
    render = (gt + 0.15 * torch.randn(H, W, 3, generator=g, dtype=torch.float64)).clamp(0, 1).contiguous()
    mask = (torch.rand(H, W, generator=g, dtype=torch.float64) > 0.8).double() if use_mask else None
    # reference: plain torch, fp64, CPU
    r64 = render.clone().requires_grad_(True)
    ref = LossComputer(0.2, fused=False).get_loss_dict(r64, gt, mask)
    (ref["total"] * 1.7).backward()
    # HIP
    r32 = render.float().to(dev).requires_grad_(True)
    out = LossComputer(0.2, fused=True).get_loss_dict(r32, gt.float().to(dev), None if mask is None else mask.float().to(dev))
    (out["total"] * 1.7).backward()
    for k in ("l1", "ssim", "total"):
        assert abs(out[k].item() - ref[k].item()) <= 2e-5 * max(1.0, abs(ref[k].item())), k
    gref = r64.grad.numpy()
    err = np.abs(r32.grad.cpu().numpy() - gref).max()
    assert err <= 1e-3 * np.abs(gref).max(), err


@pytest.mark.parametrize("shape", [(37, 53, 3), (1, 5), (1080, 1920, 3)])
def test_clamp01_matches_torch_clamp(shape):
    from easy_gaussian_splatting_amd.model import clamp01
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(7)
    x = (torch.rand(shape, generator=g) * 1.6 - 0.3)
    x.view(-1)[::7] = 0.0   # exact boundaries: aten's clamp passes the gradient at x == 0 and x == 1
    x.view(-1)[3::11] = 1.0
    v = torch.randn(shape, generator=g)
    a = x.clone().to(dev).requires_grad_(True)
    b = x.clone().to(dev).requires_grad_(True)
    ya, yb = clamp01(a), torch.clamp(b, min=0.0, max=1.0)
    assert torch.equal(ya, yb)
    ya.backward(v.to(dev)); yb.backward(v.to(dev))
    assert torch.equal(a.grad, b.grad)


@pytest.mark.parametrize("use_mask", [False, True])
def test_loss_with_folded_clamp_equals_clamp_then_loss(use_mask):
    """LossComputer(clamp_input=True) on the un-clamped image == torch.clamp(., 0, 1) followed by the loss,
    values and gradient (aten's clamp passes the gradient at exactly 0 and 1)."""
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(11)
    H, W = 70, 93
    gt = torch.rand(H, W, 3, generator=g)
    x = gt + 0.5 * torch.randn(H, W, 3, generator=g)          # a good part lies outside [0, 1]
    x.view(-1)[::13] = 0.0; x.view(-1)[5::17] = 1.0
    mask = (torch.rand(H, W, generator=g) > 0.7).float().to(dev) if use_mask else None
    a = x.clone().to(dev).requires_grad_(True)
    b = x.clone().to(dev).requires_grad_(True)
    la = LossComputer(0.2, clamp_input=True).get_loss_dict(a, gt.to(dev), mask)
    lb = LossComputer(0.2).get_loss_dict(torch.clamp(b, min=0.0, max=1.0), gt.to(dev), mask)
    for k in ("l1", "ssim", "total"):
        assert abs(la[k].item() - lb[k].item()) <= 1e-6 * max(1.0, abs(lb[k].item())), k
    (la["total"] * 1.3).backward(); (lb["total"] * 1.3).backward()
    assert float((a.grad - b.grad).abs().max()) <= 1e-6 * float(b.grad.abs().max())
    outside = (x < 0) | (x > 1)
    assert float(a.grad.cpu()[outside].abs().max()) == 0.0
