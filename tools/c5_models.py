"""Producers for BASELINE.json config 5 (the end-to-end training step) -- BENCH/TEST HARNESS, not product.

The rasterizer's callers in the reference are an encoder and the Fea2GS decoder.  Both are out of this
repository's scope; what the end-to-end measurement needs is something of the same SHAPE in front of the rasterizer,
honouring the contracts the rasterizer path depends on:

  encoder   `EDSRNOUP` (reference utils/edsrbaseline.py:83-107): conv 3->64, 16 residual blocks (conv-ReLU-conv, no
            BN, res_scale 1), conv 64->64; forward returns `res` = conv_after_body(body(x)) -- NOT res + x (:101-107).
  decoder   `Fea2GS` (reference utils/fea2gs.py:565-635): features [B,64,h,w] + scale vector [B] -> gs_parameters
            [B, 16*h*w, 9], 16 Gaussians per LR pixel in RASTER order of the (4h x 4w) grid, columns
            [sigma_x, sigma_y, rho, alpha, r, g, b, mu_x, mu_y] (:632-633), means = grid-cell centres
            ((j+0.5)/(4w), (i+0.5)/(4h)) (:547-556) + a predicted offset divided by (4w, 4h) (:623-630).

`EncoderEDSRShaped` is that encoder (own code; random init, as bench.py has no checkpoints).  `Fea2GSShaped` is NOT the
reference's window cross-attention decoder (19 M parameters of deformable attention that have nothing to do with the
rasterizer): it is a small convolutional stand-in with the same interface, output layout and ordering -- scale
embedding, two pixel-shuffle x2 stages, and per-point MLP heads of the reference's head shape (Linear c -> c -> 4c ->
out).  It exists so that the rasterizer's gradient has a real autograd graph to flow into and an optimizer to step.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F


class _ResBlock(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.a = nn.Conv2d(c, c, 3, 1, 1)
        self.b = nn.Conv2d(c, c, 3, 1, 1)

    def forward(self, x):
        return x + self.b(F.relu(self.a(x)))


class EncoderEDSRShaped(nn.Module):
    def __init__(self, num_feat=64, num_block=16):
        super().__init__()
        self.head = nn.Conv2d(3, num_feat, 3, 1, 1)
        self.body = nn.Sequential(*[_ResBlock(num_feat) for _ in range(num_block)])
        self.tail = nn.Conv2d(num_feat, num_feat, 3, 1, 1)

    def forward(self, x):
        return self.tail(self.body(self.head(x)))        # `res`, as the reference returns it


def _head(c, out):
    return nn.Sequential(nn.Linear(c, c), nn.ReLU(), nn.Linear(c, 4 * c), nn.ReLU(), nn.Linear(4 * c, out))


class Fea2GSShaped(nn.Module):
    """[B,c,h,w], scale[B] -> [B, 16 h w, 9] (see the module docstring for the contract it keeps)"""
    UP = 4      # 4 x 4 = 16 Gaussians per LR pixel (reference: num_gs_seed_sqrt * shuffle_scale1 * shuffle_scale2 per window)

    def __init__(self, channel=64):
        super().__init__()
        self.proj = nn.Conv2d(channel, channel, 3, 1, 1)
        self.scale_mlp = nn.Sequential(nn.Linear(1, 4 * channel), nn.ReLU(), nn.Linear(4 * channel, channel))
        self.up = nn.Sequential(nn.Conv2d(channel, channel * 4, 3, 1, 1), nn.PixelShuffle(2),
                                nn.Conv2d(channel, channel * 4, 3, 1, 1), nn.PixelShuffle(2))
        self.sigma, self.rho, self.alpha = _head(channel, 2), _head(channel, 1), _head(channel, 1)
        self.rgb, self.mean = _head(channel, 3), _head(channel, 2)

    def forward(self, feat, scale):
        b, c, h, w = feat.shape
        q = self.proj(feat) + self.scale_mlp((1.0 / scale).reshape(b, 1)).reshape(b, c, 1, 1)
        q = self.up(q).permute(0, 2, 3, 1)                              # [b, 4h, 4w, c]: raster order of the fine grid
        H, W = self.UP * h, self.UP * w
        mean = self.mean(q).reshape(b, -1, 2) / torch.tensor([W, H], dtype=q.dtype, device=q.device)
        ys = (torch.arange(H, device=q.device, dtype=q.dtype) + 0.5) / H
        xs = (torch.arange(W, device=q.device, dtype=q.dtype) + 0.5) / W
        ref = torch.stack([xs[None, :].expand(H, W), ys[:, None].expand(H, W)], -1).reshape(1, -1, 2)
        return torch.cat([self.sigma(q).reshape(b, -1, 2), self.rho(q).reshape(b, -1, 1), self.alpha(q).reshape(b, -1, 1),
                          self.rgb(q).reshape(b, -1, 3), mean + ref], dim=-1)


def training_step(encoder, decoder, optimizer, lq, gt, gt_sizes, scales, batched=True, dmax=0.5):
    """one `optimize_parameters` (reference TrainTestGSASR/basicsr/models/gsasr_model.py:175-245): encoder -> decoder ->
    rasterizer -> per-sample crop to gt_size -> L1 (mean over the batch) -> backward -> optimizer step.  `batched`
    renders the batch as one canvas (generate_2D_gaussian_splatting_batch), else with the reference's per-sample loop."""
    from gsasr_amd import gaussian_splatting as gsp
    optimizer.zero_grad(set_to_none=True)
    B = lq.shape[0]
    feat = encoder(lq)
    scale_vector = torch.tensor([float(s) for s in scales], device=lq.device)
    params = decoder(feat, scale_vector)                                 # [B, 16 h w, 9]
    sms = [(float(s), float(s)) for s in scales]
    loss = 0
    if batched:
        out = gsp.generate_2D_gaussian_splatting_batch(gt_sizes, params, list(scales), sms, dmax=dmax)
        for i in range(B):
            hi, wi = gt_sizes[i]
            loss = loss + F.l1_loss(out[i:i + 1, :, :hi, :wi], gt[i:i + 1, :, :hi, :wi])
    else:
        for i in range(B):
            hi, wi = gt_sizes[i]
            o = gsp.generate_2D_gaussian_splatting_step(gt_sizes[i], params[i], scales[i], sms[i], dmax=dmax).unsqueeze(0)
            loss = loss + F.l1_loss(o, gt[i:i + 1, :, :hi, :wi])
    loss = loss / B
    loss.backward()
    optimizer.step()
    return loss.detach(), params
