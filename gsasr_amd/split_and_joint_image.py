"""Tiled (memory-bounded) inference driver -- mirror of the reference's `utils/split_and_joint_image.py:98-232`
(`split_and_joint_image`, same name, arguments, defaults and assertions).

The reference cuts the (reflect-padded) LR image into overlapping `split_size` tiles, runs encoder, decoder and
rasterizer tile by tile, and pastes the SR tiles into one canvas, dropping `crop_size` rows/columns on the sides
that overlap an earlier tile.  Here the encoder/decoder callables are still run per tile (they are the caller's),
but the rasterizer runs ALL tiles, as many at a time as fit one batched canvas (64 slots, 32 767 rows), (every tile has the same size,
`gsasr_amd.gaussian_splatting.generate_2D_gaussian_splatting_batch`), and the pasting is one rule instead of the
reference's case tree -- including its one irregularity, kept on purpose: with a fractional scale factor the
reference does not crop the top of a last-column tile (or the left of a last-row tile) that is neither in the
first row/column nor the corner (`:178-185`).

Multi-GPU (not in the reference): with `distribute=True` inside an initialised `torch.distributed` job every rank
runs encoder, decoder and rasterizer for the tiles `rank, rank+world, ...` only and ONE `all_gather` brings the SR
tiles together before pasting -- tiles are independent, so this is the tile shard of SURVEY.md 8e at the level of
the whole pipeline (the caller's models are replicated, as under DDP).
"""
import math

import torch
import torch.nn.functional as F

from .gaussian_splatting import generate_2D_gaussian_splatting_batch, generate_2D_gaussian_splatting_step, max_canvas_batch


def _paste_rule(i, j, nh, nw, crop, fractional):
    """rows / columns of tile (i, j) that are dropped before pasting (reference :160-222)"""
    top, left = (crop if i > 0 else 0), (crop if j > 0 else 0)
    if fractional and i > 0 and j > 0:
        if j == nw - 1 and i != nh - 1:
            top = 0
        elif i == nh - 1 and j != nw - 1:
            left = 0
    return top, left


def split_and_joint_image(lq, scale_factor, split_size, overlap_size, model_g, model_fea2gs, scale_modify,
                          crop_size=2, default_step_size=1.2, mode='scale_modify', cuda_rendering=True,
                          if_dmax=False, dmax_mode='fix', dmax=25, distribute=False, group=None):
    h_lq, w_lq = lq.shape[-2:]
    assert overlap_size > 0 and overlap_size < split_size // 2, f"overlap size is wrong"
    stride = split_size - overlap_size
    nh, nw = math.ceil((h_lq - overlap_size) / stride), math.ceil((w_lq - overlap_size) / stride)
    pad_h, pad_w = nh * stride + overlap_size - h_lq, nw * stride + overlap_size - w_lq
    assert pad_h < h_lq, f'pad_h_lq-{pad_h} should be smaller than h_lq-{h_lq}, please decrease the split_size-{split_size}'
    assert pad_w < w_lq, f'pad_w_lq-{pad_w} should be smaller than w_lq-{w_lq}, please decrease the split_size-{split_size}'
    lq_pad = F.pad(input=lq, pad=(0, pad_w, 0, pad_h), mode='reflect')

    size_sr = math.ceil(split_size * scale_factor)
    n_tiles = nh * nw
    rank, world = 0, 1
    if distribute and torch.distributed.is_available() and torch.distributed.is_initialized():
        rank, world = torch.distributed.get_rank(group), torch.distributed.get_world_size(group)
    mine = list(range(rank, n_tiles, world))      # this rank's tiles (raster index)

    # encoder + decoder per tile (the caller's models)
    params = []
    for k in mine:
        i, j = divmod(k, nw)
        tile = lq_pad[:, :, i * stride: i * stride + split_size, j * stride: j * stride + split_size]
        feat = model_g(tile)
        scale_vector = scale_modify[0].unsqueeze(0).to(feat.device)
        params.append(model_fea2gs(feat, scale_vector)[0, :])

    # rasterizer: all tiles have the same size and scale -> batched canvases of up to 64 tiles
    tiles = []
    if cuda_rendering and params and params[0].is_cuda and len(params) > 1:
        # tiles per canvas: 64 slots, 32 767 canvas rows (17 tiles of the reference's default 480-px tile at x4)
        per_canvas = max_canvas_batch(size_sr)
        for a in range(0, len(params), per_canvas):
            chunk = params[a: a + per_canvas]
            if len(chunk) == 1:
                break
            out = generate_2D_gaussian_splatting_batch([(size_sr, size_sr)] * len(chunk), torch.stack(chunk),
                                                       [scale_factor] * len(chunk), [scale_modify] * len(chunk),
                                                       default_step_size=default_step_size, mode=mode, if_dmax=if_dmax,
                                                       dmax_mode=dmax_mode, dmax=dmax)
            tiles.extend(out[k] for k in range(len(chunk)))
    for k in range(len(tiles), len(params)):
        tiles.append(generate_2D_gaussian_splatting_step(sr_size=torch.tensor([size_sr, size_sr]), gs_parameters=params[k],
                                                         scale=scale_factor, sample_coords=None, scale_modify=scale_modify,
                                                         default_step_size=default_step_size, mode=mode,
                                                         cuda_rendering=cuda_rendering, if_dmax=if_dmax,
                                                         dmax_mode=dmax_mode, dmax=dmax))
    if world > 1:   # one all_gather of equally sized stacks (ranks with one tile less pad with zeros)
        per = (n_tiles + world - 1) // world
        stack = lq.new_zeros(per, lq.shape[1], size_sr, size_sr)
        if tiles:
            stack[: len(tiles)] = torch.stack(tiles)
        everyone = lq.new_empty(world * per, lq.shape[1], size_sr, size_sr)
        torch.distributed.all_gather_into_tensor(everyone, stack, group=group)
        tiles = [everyone[(k % world) * per + k // world] for k in range(n_tiles)]
    assert tiles[0].shape[1] == size_sr and tiles[0].shape[2] == size_sr, \
        f'tile_sr_h-{tiles[0].shape[1]}, tile_sr_w-{tiles[0].shape[2]}, split_size_sr-{size_sr} is not the same'

    # paste in raster order (later tiles overwrite earlier ones where they overlap, as in the reference)
    overlap_sr = math.ceil(overlap_size * scale_factor)
    stride_sr = size_sr - overlap_sr
    sr = torch.zeros(lq.shape[0], lq.shape[1], (nh - 1) * stride_sr + size_sr, (nw - 1) * stride_sr + size_sr,
                     device=lq.device)
    fractional = scale_factor != int(scale_factor)
    for i in range(nh):
        for j in range(nw):
            top, left = _paste_rule(i, j, nh, nw, crop_size, fractional)
            y0, x0 = i * stride_sr, j * stride_sr
            sr[:, :, y0 + top: y0 + size_sr, x0 + left: x0 + size_sr] = tiles[i * nw + j].unsqueeze(0)[:, :, top:, left:]
    return sr
