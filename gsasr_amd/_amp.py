"""Autocast-safe boundary of the rasterizer's autograd Functions.

GSASR's AMP configs run the whole forward under `torch.autocast` (basicsr/models/gsasr_amp_model.py:208).  The
kernels read raw fp32, and the reference's `GSCUDA` has no `custom_fwd`: it works there only because the decoder
happens to emit fp32.  These decorators make it a contract: inside an autocast region floating-point inputs are
cast to fp32 and the Function runs with autocast disabled; outside a region they do nothing.
"""
import torch

fp32_boundary_fwd = torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.float32)
fp32_boundary_bwd = torch.amp.custom_bwd(device_type="cuda")
