"""Deterministic stand-ins for GSASR's encoder and Fea2GS decoder, used to pin the tiled-inference driver
(tests/golden/make_golden.py runs the REFERENCE's split_and_joint_image with them; tests/ run this package's)."""
import torch


def model_g(tile):
    """'encoder': the LR tile itself, [1,3,s,s]"""
    return tile


def model_fea2gs(feat, scale_vector):
    """'decoder': one Gaussian per LR pixel, raw parameters [1, s*s, 9] = [sx, sy, rho, alpha, r, g, b, mu_x, mu_y]
    computed from the pixel's colour (so neighbouring tiles disagree in their overlap, which is what the pasting
    rule has to resolve)."""
    _, _, h, w = feat.shape
    r, g, b = (feat[0, k].reshape(-1) for k in range(3))
    ii = torch.arange(h, dtype=feat.dtype, device=feat.device).repeat_interleave(w)
    jj = torch.arange(w, dtype=feat.dtype, device=feat.device).repeat(h)
    p = torch.stack([0.6 * (r - 0.5), 0.6 * (g - 0.5), b - 0.5, 1.0 + r, 2 * r - 1, 2 * g - 1, 2 * b - 1,
                     (jj + 0.5 + 0.4 * (g - 0.5)) / w, (ii + 0.5 + 0.4 * (b - 0.5)) / h], dim=1)
    return p.unsqueeze(0) * (scale_vector.reshape(1, 1, 1).to(p.dtype) * 0 + 1)
