"""CPU restatements of the HOST side of the rasterizer path -- TEST INFRASTRUCTURE ONLY.

 * `prologue`        : what `generate_2D_gaussian_splatting_step` + `rendering_cuda[_dmax]` hand to
                       `GSCUDA.apply` (reference utils/gaussian_splatting.py:158-213 and :86-98,
                       :119-131), restated with numpy-style torch ops on CPU.
 * `rendering_python`: the reference's pure-PyTorch approximation (utils/gaussian_splatting.py:11-84),
                       used ONLY as the CPU throughput baseline of BASELINE.json config 1 and as a
                       plumbing check.  It is an approximation of the kernels (sampled kernel +
                       bilinear resampling), not a parity oracle -- SURVEY.md 8(a) row a6.
 * `autograd_render` : a vectorised torch expression of SURVEY.md 2.2 whose autograd gives an
                       independent check of the analytic backward formulas in oracle/gs_ref.c.

Parity status: pinned by tests/golden/prologue_*.npz and rendering_python_*.npz (captured from the
imported reference, tests/golden/make_golden.py).
"""
from __future__ import annotations

import math
from typing import Optional, Sequence

import torch
import torch.nn.functional as F


def activations(gs_parameters: torch.Tensor):
    """utils/gaussian_splatting.py:174-180 -- raw decoder output [N,9] -> Gaussian properties."""
    p = gs_parameters
    sigma_x = 0.99999 * torch.sigmoid(p[:, 0:1]) + 1e-6
    sigma_y = 0.99999 * torch.sigmoid(p[:, 1:2]) + 1e-6
    rho = 0.999999 * torch.tanh(p[:, 2:3])
    alpha = torch.sigmoid(p[:, 3:4])
    colours = torch.sigmoid(p[:, 4:7])
    coords = p[:, 7:9] * 2 - 1
    return sigma_x, sigma_y, rho, coords, colours * alpha


def prologue(gs_parameters: torch.Tensor, sr_size: Sequence[int], scale_modify,
             default_step_size: float = 1.2, dmax: float = 25.0, dmax_mode: str = "fix"):
    """Returns (sigmas[N,3], coords[N,2], colors[N,3], dmax) exactly as they reach GSCUDA.apply."""
    H, W = int(sr_size[0]), int(sr_size[1])
    step = default_step_size / scale_modify[0]                              # :171
    sx, sy, rho, coords, col = activations(gs_parameters)
    sigmas = torch.cat([sy / step * 2 / (W - 1), sx / step * 2 / (H - 1), rho], dim=-1)  # :121 (x/y swap)
    coords = coords.clone()
    coords[:, 0] = (coords[:, 0] + 1 - 1 / W) * W / (W - 1) - 1.0           # :122
    coords[:, 1] = (coords[:, 1] + 1 - 1 / H) * H / (H - 1) - 1.0           # :123
    if dmax_mode == "dynamic":
        dmax = (dmax + 2) / min(H, W)                                       # :203-204
    return sigmas.contiguous(), coords.contiguous(), col.contiguous(), dmax


def rendering_python(gs_parameters: torch.Tensor, sr_size: Sequence[int], scale_modify,
                     default_step_size: float = 1.2, max_buffer: int = 2000) -> torch.Tensor:
    """Restatement of the `cuda_rendering=False` branch (utils/gaussian_splatting.py:11-84)."""
    H, W = int(sr_size[0]), int(sr_size[1])
    step = default_step_size / scale_modify[0]          # :171 (a 0-dim fp32 tensor for tensor scale_modify: fp32 arithmetic below, as in the reference)
    sx, sy, rho, coords, col = activations(gs_parameters)
    n = sx.shape[0]
    # inverse covariance of [[sx^2, r sx sy],[r sx sy, sy^2]] and sqrt(det)                 (:15-29)
    det = (sx * sy) ** 2 * (1 - rho ** 2)
    if (det < 0).any():
        raise ValueError("Covariance matrix must be positive semi-definite")
    cov = torch.stack([torch.cat([sx ** 2, rho * sx * sy], -1),
                       torch.cat([rho * sx * sy, sy ** 2], -1)], dim=-2)     # [n,2,2]
    inv = torch.inverse(cov)
    num_step = int(10 * 2 / step)                                            # :32
    ax = torch.tensor([i * step for i in range(num_step)])
    ax = ax - ax.mean()                                                      # :33-36
    xx = ax[:, None].expand(num_step, num_step)                              # varies along rows
    yy = ax[None, :].expand(num_step, num_step)
    xy = torch.stack([xx, yy], dim=-1)                                       # [S,S,2]
    out = torch.zeros(3, H, W)
    for s0 in range(0, n, max_buffer):                                       # :47-57
        s1 = min(s0 + max_buffer, n)
        b = s1 - s0
        z = torch.einsum("xyi,bij,xyj->bxy", xy, -0.5 * inv[s0:s1], xy)      # :61
        kern = torch.exp(z) / (2 * math.pi * torch.sqrt(torch.det(cov[s0:s1])).view(b, 1, 1))
        kmax = kern.amax(dim=(-1, -2), keepdim=True)
        kern = kern / (kmax + 1e-4)                                          # :64-65
        kern = kern[:, None].expand(b, 3, num_step, num_step)
        theta = torch.zeros(b, 2, 3)
        theta[:, 0, 0] = W / num_step                                        # :72-76
        theta[:, 1, 1] = H / num_step
        theta[:, 0, 2] = -coords[s0:s1, 0] * W / num_step
        theta[:, 1, 2] = -coords[s0:s1, 1] * H / num_step
        grid = F.affine_grid(theta, size=(b, 3, H, W), align_corners=False)
        moved = F.grid_sample(kern, grid, align_corners=False)               # :78-80
        out += (col[s0:s1, :, None, None] * moved).sum(0)                    # :81-82
    return out


def autograd_render(sigmas: torch.Tensor, coords: torch.Tensor, colors: torch.Tensor, h: int, w: int,
                    dmax: Optional[float] = None) -> torch.Tensor:
    """Differentiable dense evaluation of SURVEY.md 2.2 (all pixels x all Gaussians); [h,w,3].

    Box decisions follow the reference: pixel coordinates are the float32 rounding of the double
    expression, differences and comparisons are taken in float32, whatever dtype the maths runs in.
    """
    dt = sigmas.dtype
    px64 = 2.0 * torch.arange(w, dtype=torch.float64) / (w - 1) - 1.0
    py64 = 2.0 * torch.arange(h, dtype=torch.float64) / (h - 1) - 1.0
    pxf, pyf = px64.float(), py64.float()
    dxf = pxf[None, :, None] - coords[:, 0].detach().float()[None, None, :]
    dyf = pyf[:, None, None] - coords[:, 1].detach().float()[None, None, :]
    dx = pxf.to(dt)[None, :, None] - coords[:, 0][None, None, :]
    dy = pyf.to(dt)[:, None, None] - coords[:, 1][None, None, :]
    sx, sy, rho = sigmas[:, 0], sigmas[:, 1], sigmas[:, 2]
    d = dx * dx / (sx * sx) - 2 * rho * dx * dy / (sx * sy) + dy * dy / (sy * sy)
    v = torch.exp(-0.5 / (1 - rho * rho) * d)
    if dmax is not None:
        inside = (dxf.abs() <= dmax) & (dyf.abs() <= dmax)
        v = torch.where(inside, v, torch.zeros((), dtype=dt))
    return torch.einsum("hws,sc->hwc", v, colors)
