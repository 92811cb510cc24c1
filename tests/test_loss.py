"""Fused L1 + D-SSIM loss kernel vs the reference formulation in PyTorch (restated in dg-mesh_amd/scene.py from
R/utils/loss_utils.py; that restatement itself is checked against the reference's functions in the golden below)."""
import numpy as np
import pytest
import torch

from conftest import pkg


def test_torch_ssim_matches_reference_golden():
    """Value computed here with the reference's own l1_loss/ssim (utils/loss_utils.py) when the golden was made."""
    S = pkg("scene")
    g = torch.Generator().manual_seed(0)
    a = torch.rand(3, 37, 53, generator=g)
    b = (a + 0.1 * torch.randn(3, 37, 53, generator=g)).clamp(0, 1)
    assert abs(S.ssim(a, b).item() - 0.9484018683433533) < 2e-6
    assert abs(S.l1_loss(a, b).item() - 0.07576507329940796) < 1e-7


@pytest.mark.gpu
@pytest.mark.parametrize("H,W", [(37, 53), (800, 800), (16, 16), (129, 64)])
def test_fused_loss_matches_torch(H, W):
    S, Lm = pkg("scene"), pkg("loss")
    dev = "cuda"
    g = torch.Generator().manual_seed(1)
    gt = torch.rand(3, H, W, generator=g).to(dev)
    img0 = (gt.cpu() + 0.2 * torch.randn(3, H, W, generator=g)).clamp(0, 1.2).to(dev)
    a = img0.clone().requires_grad_(True)
    b = img0.clone().requires_grad_(True)
    lam = 0.2
    ref = (1 - lam) * S.l1_loss(a, gt) + lam * (1 - S.ssim(a, gt))
    got = Lm.image_loss(b, gt, lam)
    assert abs(ref.item() - got.item()) < 1e-5 * max(1.0, abs(ref.item()))
    (ref * 3.0).backward()
    (got * 3.0).backward()
    err = (a.grad - b.grad).abs().max().item() / a.grad.abs().max().item()
    assert err < 1e-4, err


def _reference_loss_functions():
    """The reference's own l1_loss / ssim (R/utils/loss_utils.py:18-19, 45-76), byte-compiled by oracle/build_ref.sh into
    oracle/_ref/pyref/loss_utils.pyc (binaries only; they travel to the GPU box like the reference kernels)."""
    import os
    import sys
    from conftest import ROOT
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    import bench
    mods = bench.reference_host_modules()
    if mods is None:
        pytest.skip("oracle/_ref/pyref/loss_utils.pyc not built (oracle/build_ref.sh needs /root/reference)")
    return mods[1].l1_loss, mods[1].ssim


SHAPES = [(37, 53), (16, 16), (129, 64), (401, 333), (800, 800)]


@pytest.mark.parametrize("H,W", SHAPES[:4])
def test_torch_restatement_matches_the_reference_functions(H, W):
    """scene.l1_loss / scene.ssim (what the CPU tests and the port baseline use) against the reference's functions themselves,
    values and gradients, on the host."""
    S = pkg("scene")
    r_l1, r_ssim = _reference_loss_functions()
    g = torch.Generator().manual_seed(H * 1000 + W)
    gt = torch.rand(3, H, W, generator=g)
    img0 = (gt + 0.2 * torch.randn(3, H, W, generator=g)).clamp(0, 1.2)
    a, b = img0.clone().requires_grad_(True), img0.clone().requires_grad_(True)
    ours = 0.8 * S.l1_loss(a, gt) + 0.2 * (1 - S.ssim(a, gt))
    ref = 0.8 * r_l1(b, gt) + 0.2 * (1 - r_ssim(b, gt))
    assert abs(ours.item() - ref.item()) <= 1e-6 * max(1.0, abs(ref.item()))
    ours.backward()
    ref.backward()
    assert (a.grad - b.grad).abs().max().item() <= 1e-5 * b.grad.abs().max().item()


@pytest.mark.gpu
@pytest.mark.parametrize("H,W", SHAPES)
def test_fused_loss_matches_the_reference_functions(H, W):
    """dgm_image_loss_forward / backward against the REFERENCE's own l1_loss and ssim (loss_utils.py:18-19, 45-76; the loss of
    R/train.py:307-311) evaluated by PyTorch on the same GPU: value to 1e-5 relative, gradient to 1e-4 of its maximum, at the
    metric's 800 x 800 and at sizes that are no multiple of the kernel's 32 x 16 tiles."""
    Lm = pkg("loss")
    r_l1, r_ssim = _reference_loss_functions()
    dev = "cuda"
    g = torch.Generator().manual_seed(H * 1000 + W)
    gt = torch.rand(3, H, W, generator=g).to(dev)
    img0 = (gt.cpu() + 0.2 * torch.randn(3, H, W, generator=g)).clamp(0, 1.2).to(dev)
    lam = 0.2
    a, b = img0.clone().requires_grad_(True), img0.clone().requires_grad_(True)
    ref = (1.0 - lam) * r_l1(a, gt) + lam * (1.0 - r_ssim(a, gt))
    got = Lm.image_loss(b, gt, lam)
    assert abs(ref.item() - got.item()) <= 1e-5 * abs(ref.item()), (ref.item(), got.item())
    (ref * 3.0).backward()
    (got * 3.0).backward()
    err = (a.grad - b.grad).abs().max().item() / a.grad.abs().max().item()
    assert err <= 1e-4, err
