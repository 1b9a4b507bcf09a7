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

`EncoderEDSRShaped` is that encoder (own code; random init, as bench.py has no checkpoints).  Two decoders:

  `Fea2GSDecoder`  (round 5) the reference decoder's ARCHITECTURE in this harness's own code, at the shipped EDSR-baseline
            configuration (options/train/paper/train_GSASR_EDSR-Baseline_paper_bicubic_x1_4.yml:64-78: channel 180, 6 heads,
            12 x 12 windows, 144 learned Gaussian seeds per window, 1 cross-attention block of 2 layers, 6 self-attention
            blocks of 6 layers, pixel-shuffle 2 x 2, five MLP heads): per layer a seed-to-scale-embedding attention, an FFN, a
            window attention with a learned relative-position bias (seeds -> the window's image features, odd layers on
            the half-window-shifted feature map; seeds -> seeds, odd layers on the half-window-shifted seed grid) and an
            FFN, pre-norm residual throughout (utils/fea2gs.py:116-449, 565-635).  Same parameter count class (~19 M), same
            operator mix, so the end-to-end step of config 5 has the reference's producer cost in front of the rasterizer;
            random init -- no claim about image quality.
  `Fea2GSShaped`   the small convolutional stand-in of round 4 with the same interface, output layout and ordering (scale
            embedding, two pixel-shuffle x2 stages, the reference's head shape): what tests/test_c5_e2e.py steps, because
            gradient equality batched-vs-loop does not need 19 M parameters.
"""
import math

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


class _FFN(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.a, self.b = nn.Linear(c, c), nn.Linear(c, c)

    def forward(self, x):
        return self.b(F.relu(self.a(x)))


class _BiasedWindowAttention(nn.Module):
    """multi-head attention of `nq` query tokens on an n x n grid over `nk` key tokens on an m x m grid of the same window,
    with a learned bias per head and relative offset (both grids scaled to a common lattice)"""

    def __init__(self, c, heads, n, m):
        super().__init__()
        self.heads = heads
        self.q, self.k, self.v, self.o = nn.Linear(c, c), nn.Linear(c, c), nn.Linear(c, c), nn.Linear(c, c)
        # offsets between query cell centres and key cell centres on the lattice of n * m points per side
        qy, qx = torch.meshgrid(torch.arange(n), torch.arange(n), indexing="ij")
        ky, kx = torch.meshgrid(torch.arange(m), torch.arange(m), indexing="ij")
        cq = torch.stack([(2 * qy.flatten() + 1) * m, (2 * qx.flatten() + 1) * m], 1)          # [n*n, 2]
        ck = torch.stack([(2 * ky.flatten() + 1) * n, (2 * kx.flatten() + 1) * n], 1)          # [m*m, 2]
        d = cq[:, None, :] - ck[None, :, :]                                                     # [n*n, m*m, 2]
        span = 2 * n * m
        uy, iy = torch.unique(d[..., 0] + span, return_inverse=True)
        ux, ix = torch.unique(d[..., 1] + span, return_inverse=True)
        self.register_buffer("index", iy * len(ux) + ix, persistent=False)
        table = torch.zeros(len(uy) * len(ux), heads)
        nn.init.trunc_normal_(table, std=0.02)
        near = (d.float() ** 2).sum(-1).argmin(1)                      # the key cell nearest to each query cell starts favoured
        table[self.index[torch.arange(n * n), near]] += 2.0
        self.bias = nn.Parameter(table)

    def forward(self, xq, xk):
        b, nq, c = xq.shape
        h = self.heads
        q = self.q(xq).view(b, nq, h, c // h).transpose(1, 2)
        k = self.k(xk).view(b, -1, h, c // h).transpose(1, 2)
        v = self.v(xk).view(b, -1, h, c // h).transpose(1, 2)
        bias = self.bias[self.index].permute(2, 0, 1).unsqueeze(0)     # [1, heads, nq, nk]
        out = F.scaled_dot_product_attention(q, k, v, attn_mask=bias.to(q.dtype))
        return self.o(out.transpose(1, 2).reshape(b, nq, c))


class _SeedLayer(nn.Module):
    """one decoder layer: seeds attend to the scale embedding, FFN, window attention (to image features, or among the seeds),
    FFN -- pre-norm residuals.  `cross`: keys are the window's image features; else the seeds themselves."""

    def __init__(self, c, heads, win, shift, cross):
        super().__init__()
        self.win, self.shift, self.cross = win, shift, cross
        self.n1, self.n2, self.n3, self.n4 = (nn.LayerNorm(c) for _ in range(4))
        self.to_scale = nn.MultiheadAttention(c, heads, batch_first=True)
        self.f1, self.f2 = _FFN(c), _FFN(c)
        self.attn = _BiasedWindowAttention(c, heads, win, win)

    def forward(self, x, pos, feat, scale_emb, grid):
        b, m, n = grid                                               # images, windows per column / row
        y = self.n1(x)
        x = x + self.to_scale(y + pos, scale_emb, scale_emb, need_weights=False)[0]
        x = x + self.f1(self.n2(x))
        y = self.n3(x)
        w = self.win
        if self.cross:
            f = torch.roll(feat, (-self.shift, -self.shift), (2, 3)) if self.shift else feat
            c = f.shape[1]
            keys = f.view(b, c, m, w, n, w).permute(0, 2, 4, 3, 5, 1).reshape(b * m * n, w * w, c)
            x = x + self.attn(y + pos, keys)
        else:
            if self.shift:     # the seeds of all windows as one grid, rolled by half a window, cut into windows again
                c = y.shape[-1]
                g = y.view(b, m, n, w, w, c).permute(0, 1, 3, 2, 4, 5).reshape(b, m * w, n * w, c)
                g = torch.roll(g, (-self.shift, -self.shift), (1, 2))
                y = g.view(b, m, w, n, w, c).permute(0, 1, 3, 2, 4, 5).reshape(b * m * n, w * w, c)
            a = self.attn(y, y)
            if self.shift:
                g = a.view(b, m, n, w, w, c).permute(0, 1, 3, 2, 4, 5).reshape(b, m * w, n * w, c)
                g = torch.roll(g, (self.shift, self.shift), (1, 2))
                a = g.view(b, m, w, n, w, c).permute(0, 1, 3, 2, 4, 5).reshape(b * m * n, w * w, c)
            x = x + a
        return x + self.f2(self.n4(x))


class _SeedBlock(nn.Module):
    def __init__(self, c, heads, win, layers, cross):
        super().__init__()
        self.norm = nn.LayerNorm(c)
        self.layers = nn.ModuleList([_SeedLayer(c, heads, win, 0 if i % 2 == 0 else win // 2, cross) for i in range(layers)])
        self.mlp = _FFN(c)

    def forward(self, x, pos, feat, scale_emb, grid):
        y = self.norm(x)
        for layer in self.layers:
            y = layer(y, pos, feat, scale_emb, grid)
        return x + self.mlp(y)


class Fea2GSDecoder(nn.Module):
    """[B,c,h,w] (h, w multiples of the window), scale[B] -> [B, 16 h w, 9]: the reference decoder's architecture (module
    docstring), one seed per LR pixel, 16 Gaussians per seed after the two pixel shuffles, raster order of the 4h x 4w grid"""
    UP = 4

    def __init__(self, inchannel=64, channel=180, heads=6, window=12, cross_layers=2, self_blocks=6, self_layers=6):
        super().__init__()
        self.c, self.win = channel, window
        self.seed = nn.Parameter(torch.randn(window * window, channel))
        self.pos = nn.Parameter(torch.randn(window * window, channel))
        self.proj = nn.Sequential(nn.Conv2d(inchannel, channel, 3, 1, 1), nn.ReLU(), nn.Conv2d(channel, channel, 3, 1, 1))
        self.scale_mlp = nn.Sequential(nn.Linear(1, 4 * channel), nn.ReLU(), nn.Linear(4 * channel, channel))
        self.cross = nn.ModuleList([_SeedBlock(channel, heads, window, cross_layers, True)])
        self.selfs = nn.ModuleList([_SeedBlock(channel, heads, window, self_layers, False) for _ in range(self_blocks)])
        self.up = nn.Sequential(nn.Conv2d(channel, channel * 4, 3, 1, 1), nn.PixelShuffle(2),
                                nn.Conv2d(channel, channel * 4, 3, 1, 1), nn.PixelShuffle(2))
        self.sigma, self.rho, self.alpha = _head(channel, 2), _head(channel, 1), _head(channel, 1)
        self.rgb, self.mean = _head(channel, 3), _head(channel, 2)

    def forward(self, feat, scale):
        b, _, h, w = feat.shape
        win, c = self.win, self.c
        assert h % win == 0 and w % win == 0, "features must be padded to whole windows (the reference pads its input the same way)"
        m, n = h // win, w // win
        grid = (b, m, n)
        x = self.seed.expand(b * m * n, -1, -1)
        pos = self.pos.expand(b * m * n, -1, -1)
        se = self.scale_mlp((1.0 / scale).reshape(b, 1))                                   # [b, c]
        se = se[:, None, None, :].expand(b, m * n, win * win, c).reshape(b * m * n, win * win, c)
        f = self.proj(feat)
        for blk in self.cross:
            x = blk(x, pos, f, se, grid)
        y = x
        for blk in self.selfs:
            y = blk(y, pos, f, se, grid)
        x = x + y
        q = x.view(b, m, n, win, win, c).permute(0, 5, 1, 3, 2, 4).reshape(b, c, h, w)      # one seed per LR pixel
        q = self.up(q).permute(0, 2, 3, 1)                                                   # [b, 4h, 4w, c]
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
