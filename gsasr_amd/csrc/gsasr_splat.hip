// gsasr_splat.hip -- MI355X (gfx950 / CDNA4) 2D Gaussian-splatting rasterizer behind the C ABI of
// include/gsasr_splat.h.  Written for wave64 / 256 CUs / 8 XCDs; not a translation of the
// reference's CUDA (utils/gs_cuda*/gs.cu), whose semantics it reproduces:
//
//   forward : img[p,:] += sum_s [|dx|<=dmax & |dy|<=dmax] exp(w1_s * q_s(p)) * colors[s,:]
//   backward: analytic gradient of sum(grad_img * img) w.r.t. sigmas[s,3], coords[s,2], colors[s,3]
//
// Pipeline (all on the caller's stream, no host sync; DESIGN.md has the measurements):
//   plan     k_classify  per Gaussian: pixel window of (dmax box  ∩  sigma*sqrt(2 tau) support box) under the
//                        CONSERVATIVE cutoff tau = ln(s / eps); class {normal -> 16x16-px cell of its centre | large |
//                        dead (far / near: adapt_kcut)}; rank in its cell (one returning atomic per wave and distinct
//                        cell, one round trip); max extent of the normal class per group of 32 blocks; the px/py
//                        pixel-coordinate tables (double expression rounded to float, as gs.cu:27-28 does per pixel).
//            k_scan      exclusive scan of the cell histogram + max-extent reduction (one workgroup;
//                        k_scan_local + k_scan_fix for grids above 8192 cells).
//            k_bin       the cutoff the WINDOWS are built with -- tau' = ln(K / budget) <= tau, K counted from the cell
//                        histogram (adapt_kcut: same eps * max|colour| bound per pixel, 20-30% fewer pairs) -- then
//                        counting-sort placement (cell start + rank) fused with packing: 32-byte records
//                        {x, y, IX, NR, IY, r, g, b} (the coefficients of the completed-square exponent
//                        -(IX dx)^2 - (IY dy + NR IX dx)^2 in log2 units, computed in double), backward-epilogue constants, 16-byte windows with the
//                        per-tile-band column spans of the ellipse {exponent >= -tau}, and the first 8 bytes of the
//                        windows once more as a dense array for the coarse tests.
//   forward  k_render_fwd2 PIXEL-stationary: one wave64 = one 8x16 pixel sub-tile (2 px per lane, packed
//                        fp32), RGB accumulators in registers; four sub-tiles side by side per workgroup.  Two-level
//                        walk: the workgroup tests the candidates of its 32x16 tile (cell rows within the class'
//                        max extent, 64 per wave and chunk) ONCE, cooperatively, into a shared LDS list; each wave
//                        then runs the full window + span test on the survivors only, compacts the hits' records
//                        into its LDS stage and evaluates them from broadcast LDS reads; no atomics on the
//                        image, one coalesced store / read-modify-write of the tile.
//                        Images with 4096..8191 sub-tiles: eight waves per workgroup, two per sub-tile.
//            k_render_fwd_split   same per-wave code for small images: one sub-tile per workgroup, its
//                        chunks dealt to 2..16 waves, partial sums combined through LDS.
//   backward k_render_bwd  GAUSSIAN-stationary: one wave64 = one Gaussian (its records fetched by the scalar
//                        unit in one batch), lanes laid 16/32/64 wide over its window, two rows per trip
//                        (packed fp32), grad_img through L1/L2, three residual moments + three colour
//                        sums per lane, one LDS + DPP wave reduction, and the gradient written by the same
//                        wave (no finalize pass).  Gaussians of the "large" class are split into row chunks
//                        spread over all waves, combined with fp32 atomics; the wave finishing the last
//                        chunk writes the gradient.
//   prologue k_prologue_fwd/bwd  the reference's host prologue (activations + kernel frame) and its
//                        chain rule as one kernel each.
//   shard    k_band_select/merge  multi-GPU row bands: the Gaussians whose window crosses a band edge go to the
//                        neighbouring rank and their partial gradients come back.
//   batch    a batched canvas (gsasr_dims.batch > 1, row f2) runs B samples of different sizes through the same
//                        kernels: per-sample geometry (Geo) instead of the image's.
//   sampled  k_pts_count/scan/place, k_sample_fwd, k_pts_grads, k_sample_bwd   values and gradients at a list of
//                        pixels only (the reference's `sample_coords`, row f4): point-stationary two-level walk forward,
//                        Gaussian-stationary backward with eight Gaussians per wave.
//
// No MFMA: this is gather/scatter-accumulate with one transcendental per pair, not a contraction.
// This file is the WHOLE library as one translation unit (tools/mb.hip includes it); gsasr_amd/build.py compiles the parts
// below separately and links them.
#include "splat_common.h"
#include "splat_api.hip"
#include "splat_plan.hip"
#include "splat_forward.hip"
#include "splat_backward.hip"
#include "splat_backward_home.hip"
#include "splat_step.hip"
#include "splat_sampled.hip"
#include "splat_shard.hip"
