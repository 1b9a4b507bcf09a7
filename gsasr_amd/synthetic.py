"""Deterministic synthetic Gaussians of GSASR's shape, used by bench.py and tests.

For an `h_lr x w_lr` LR grid and scale `s`: N = gpp*h_lr*w_lr Gaussians (gpp per LR pixel; the synthetic
configs of BASELINE.json use 1, Fea2GS emits 16), H = s*h_lr, W = s*w_lr.  Raw decoder-style parameters
`gs_parameters[N,9] = 0.5*randn` with the means on the LR pixel-centre grid plus U(-0.5,0.5)/w_lr jitter
(mimics reference utils/fea2gs.py:623-630, raster order).
"""
from __future__ import annotations

import torch


def gs_parameters(h_lr: int, w_lr: int, seed: int = 0, gpp: int = 1, device="cpu") -> torch.Tensor:
    g = torch.Generator(device="cpu").manual_seed(seed)
    n = h_lr * w_lr * gpp
    p = 0.5 * torch.randn(n, 9, generator=g)
    ii = torch.arange(h_lr).repeat_interleave(w_lr * gpp).float()
    jj = torch.arange(w_lr).repeat_interleave(gpp).repeat(h_lr).float()
    jit = torch.rand(n, 2, generator=g) - 0.5
    p[:, 7] = (jj + 0.5 + jit[:, 0]) / w_lr
    p[:, 8] = (ii + 0.5 + jit[:, 1]) / h_lr
    return p.to(device)


def kernel_inputs(h_lr: int, w_lr: int, scale: float, seed: int = 0, gpp: int = 1, device="cpu"):
    """(sigmas[N,3], coords[N,2], colors[N,3], H, W) as they reach GSCUDA.apply, via the package's own
    host prologue (gsasr_amd.gaussian_splatting)."""
    from .gaussian_splatting import _activate, _to_kernel_frame
    H, W = int(round(h_lr * scale)), int(round(w_lr * scale))
    p = gs_parameters(h_lr, w_lr, seed, gpp, device)
    sx, sy, rho, xy, col = _activate(p)
    sigmas, coords, colors, _, _ = _to_kernel_frame(sx, sy, rho, xy, col, (H, W), 1.2 / scale)
    return sigmas, coords, colors, H, W


def grad_image(H: int, W: int, seed: int = 1, device="cpu") -> torch.Tensor:
    g = torch.Generator(device="cpu").manual_seed(seed)
    return torch.rand(H, W, 3, generator=g).to(device)
