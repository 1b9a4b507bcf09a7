"""BASELINE.json config 5 end to end (bench.py --config c5e2e, tools/c5_models.py): the gradient that reaches the
producers through the BATCHED rasterizer equals the gradient of the reference's per-sample loop
(TrainTestGSASR/basicsr/models/gsasr_model.py:190-233), and the producers keep the shape/ordering contract the
rasterizer path relies on (utils/edsrbaseline.py:101-107, utils/fea2gs.py:546-556,609-635)."""
import os
import sys

import pytest
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
import c5_models  # noqa: E402


def test_producer_contracts_cpu():
    torch.manual_seed(0)
    enc, dec = c5_models.EncoderEDSRShaped(), c5_models.Fea2GSShaped()
    assert sum(p.numel() for p in enc.parameters()) == 1220416          # EDSRNOUP's parameter count
    x = torch.rand(2, 3, 6, 5)
    feat = enc(x)
    assert feat.shape == (2, 64, 6, 5)
    p = dec(feat, torch.tensor([4.0, 2.5]))
    assert p.shape == (2, 16 * 6 * 5, 9)                                # N = 16 h w, nine columns
    # means: raster order of the (4h x 4w) grid, x fastest, cell centres + a small offset
    mu = p[0, :, 7:9].detach().reshape(24, 20, 2)
    assert (mu[:, 1:, 0] > mu[:, :-1, 0]).all() and (mu[1:, :, 1] > mu[:-1, :, 1]).all()
    cx = (torch.arange(20) + 0.5) / 20
    assert (mu[0, :, 0] - cx).abs().max() < 0.5 / 20
    # the scale vector is an input of the decoder
    q = dec(feat, torch.tensor([2.0, 2.5]))
    assert not torch.allclose(p[0], q[0]) and torch.allclose(p[1], q[1])


def test_fea2gs_architecture_decoder_contract_cpu():
    """tools/c5_models.Fea2GSDecoder: the reference decoder's architecture at the shipped EDSR-baseline configuration
    (options/train/paper/train_GSASR_EDSR-Baseline_paper_bicubic_x1_4.yml:64-78; utils/fea2gs.py:451-635) -- parameter count
    class, output layout and ordering, dependence on the scale vector, windows talking to each other only through the shifted
    layers, gradients reaching the seeds and the image features"""
    torch.manual_seed(0)
    dec = c5_models.Fea2GSDecoder()
    n = sum(p.numel() for p in dec.parameters())
    assert 18e6 < n < 21e6, n
    feat = torch.randn(2, 64, 24, 12, requires_grad=True)                # 2 x 1 windows of 12 x 12
    p = dec(feat, torch.tensor([4.0, 2.5]))
    assert p.shape == (2, 16 * 24 * 12, 9)
    mu = p[0, :, 7:9].detach().reshape(96, 48, 2)
    cx, cy = (torch.arange(48) + 0.5) / 48, (torch.arange(96) + 0.5) / 96
    assert (mu[..., 0] - cx[None, :]).abs().max() < 1.0 / 48 and (mu[..., 1] - cy[:, None]).abs().max() < 1.0 / 96
    q = dec(feat, torch.tensor([2.0, 2.5]))
    assert not torch.allclose(p[0], q[0]) and torch.allclose(p[1], q[1], atol=1e-6)
    p[0, :, :7].sum().backward()
    assert float(dec.seed.grad.abs().max()) > 0 and float(feat.grad[0].abs().max()) > 0 and float(feat.grad[1].abs().max()) == 0.0


@pytest.mark.gpu
def test_batched_step_gradient_equals_per_sample_loop():
    dev = torch.device("cuda:0")
    torch.manual_seed(3)
    enc, dec = c5_models.EncoderEDSRShaped(num_block=2).to(dev), c5_models.Fea2GSShaped().to(dev)
    B = 4
    lq = torch.rand(B, 3, 12, 12, device=dev)
    sizes = [(48, 48), (40, 44), (48, 36), (30, 48)]                    # ragged gt sizes, as the datasets produce
    scales = [4.0, 3.6, 4.0, 3.9]
    gt = torch.rand(B, 3, 48, 48, device=dev)
    params = list(enc.parameters()) + list(dec.parameters())
    grads = {}
    for batched in (True, False):
        opt = torch.optim.SGD(params, lr=0.0)
        loss, _ = c5_models.training_step(enc, dec, opt, lq, gt, sizes, scales, batched=batched, dmax=0.5)
        grads[batched] = ([p.grad.clone() for p in params], float(loss))
    assert abs(grads[True][1] - grads[False][1]) <= 1e-6 * max(1.0, abs(grads[False][1]))
    worst = 0.0
    for a, b in zip(grads[True][0], grads[False][0]):
        assert torch.isfinite(a).all()
        worst = max(worst, float((a - b).abs().max()) / max(1e-12, float(b.abs().max())))
    assert worst <= 5e-4, worst
    # and the step really trains: Adam moves the loss
    opt = torch.optim.Adam(params, lr=1e-3)
    l0 = float(c5_models.training_step(enc, dec, opt, lq, gt, sizes, scales)[0])
    for _ in range(5):
        l1 = float(c5_models.training_step(enc, dec, opt, lq, gt, sizes, scales)[0])
    assert l1 < l0
