/*
 * gsasr_splat.h -- C ABI of the MI355X-native 2D Gaussian-splatting rasterizer (libgsasr_splat.so).
 *
 * This is the drop-in boundary for GSASR's rasterizer path.  Every entry point takes plain device
 * pointers, sizes and a HIP stream; there are no torch types.  What each one replaces in the
 * reference (paths relative to the reference tree):
 *
 *   gsasr_gs_render            <- _gs_render            utils/gs_cuda/gs.h:1-10       (gs.cu:64-79)
 *   gsasr_gs_render_backward   <- _gs_render_backward   utils/gs_cuda/gs.h:12-24      (gs.cu:180-198)
 *   gsasr_gs_render_dmax       <- _gs_render            utils/gs_cuda_dmax/gs.h:1-11  (gs.cu:67-83)
 *   gsasr_gs_render_backward_dmax <- _gs_render_backward utils/gs_cuda_dmax/gs.h:13-26 (gs.cu:167-186)
 *
 * i.e. exactly what utils/gs_cuda{,_dmax}/gswrapper.cpp:9-71 binds through pybind11 (`gs_render`,
 * `gs_render_backward`).  The four functions keep the reference argument order and meaning
 * (fp32, contiguous, `rendered_img` accumulated into, dmax-backward `+=` into caller-zeroed
 * outputs, unbounded backward overwrites) and add: the stream to launch on (the reference uses the
 * null stream) and an int status instead of void.
 *
 * The plan API below them is what the PyTorch host code actually uses: it exposes the binning
 * workspace so that forward and backward of one autograd node share it, and the [row0,row1) row
 * band for the multi-GPU HR-tile shard.  INTEGRATION.md shows the reference-side
 * binding.
 *
 * Conventions: all pointers are device pointers valid on the current HIP device (the one exception is the small
 * host array gsasr_dims.sample_hw of a batched canvas); `stream` is a
 * hipStream_t passed as void* (NULL = default stream); calls only enqueue work (no host sync);
 * return 0 on success or a negative gsasr_status / positive hipError_t, with a thread-local message
 * available from gsasr_last_error().  c must be 3 (the reference forward hard-codes stride 3,
 * gs_cuda/gs.cu:58, gs_cuda_dmax/gs.cu:29-31).  h, w >= 2 (the grid is 2*i/(n-1)-1).
 */
#ifndef GSASR_SPLAT_H
#define GSASR_SPLAT_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* the library is built with -fvisibility=hidden: these entry points are everything it exports */
#if defined(__GNUC__) || defined(__clang__)
#define GSASR_API __attribute__((visibility("default")))
#else
#define GSASR_API
#endif

#define GSASR_SPLAT_ABI_VERSION 7 /* 7: GSASR_FLAG_BWD_HOME; 2: gsasr_dims gained batch / slot / sample_hw; 3: grad_rows, the backward flags; 4: the _sm step entry points, step_size = NULL in the step backwards; 5: gsasr_dims.list_cap (tile lists); 6: the kernel-choice registry */

enum gsasr_status {
    GSASR_OK = 0,
    GSASR_ERR_ARG = -1,        /* bad dims / null pointer / c != 3 */
    GSASR_ERR_WORKSPACE = -2,  /* workspace too small or misaligned */
    GSASR_ERR_PLAN = -3        /* workspace does not hold a plan for these dims */
};

/* flags */
#define GSASR_FLAG_OVERWRITE_IMAGE 2u /* forward STORES the splat (img need not be initialised) instead of
                                         accumulating into it; saves the caller's memset and a 12 B/px read */
#define GSASR_FLAG_CHW_IMAGE 8u       /* forward writes img as planar [3, row1-row0, w] (the layout the host API
                                         returns, utils/gaussian_splatting.py:129) instead of [row1-row0, w, 3] */
#define GSASR_FLAG_OVERWRITE_GRADS 4u /* backward STORES the gradients (outputs need not be zeroed) instead
                                         of adding into them */
#define GSASR_FLAG_STRIDE8 16u        /* sigmas/coords/colors AND g_sigmas/g_coords/g_colors are columns of packed
                                         [s,8] records {sx,sy,rho,x,y,r,g,b}: element k of Gaussian i is at
                                         ptr[8*i+k].  Pass base, base+3, base+5 of one array; this is the wire
                                         format of the multi-GPU exchange, so nothing is repacked around it */

#define GSASR_FLAG_CHW_GRAD 32u       /* backward reads grad_img as planar [3, row1-row0, w] (what autograd hands back for the
                                         planar image of GSASR_FLAG_CHW_IMAGE; batched canvas: [B, 3, grad_rows, w]) instead
                                         of [row1-row0, w, 3]: no permute pass in front of the backward.  Tile backward only */
#define GSASR_FLAG_FORWARD_ONLY 64u   /* the plan will not be used by a backward (inference): the workspace carries no backward
                                         records, constants, accumulators, slots or gradient scratch (about half the bytes per
                                         Gaussian) and the plan does not write them; a backward on it is GSASR_ERR_PLAN */
#define GSASR_FLAG_BWD_GAUSSIAN 128u  /* backward kernel choice (default: the library picks): Gaussian-stationary
                                         (one wave per Gaussian sweeping its window through L1/L2) ...              */
#define GSASR_FLAG_BWD_TILE 256u      /* ... or tile-stationary (one workgroup per 32x16-px tile, grad_img staged once
                                         in LDS, deterministic partial-gradient slots + gather).  Set it on the dims the
                                         PLAN is made with: the workspace then carries the slots (32 * 8 or 16 bytes per
                                         Gaussian) and the per-quadrant ellipse spans.  Whether a workspace carries slots
                                         is the PLAN's decision (the library remembers it per workspace): a backward that
                                         asks for this kernel on a plan without slots runs the Gaussian-stationary kernel
                                         instead (with GSASR_FLAG_CHW_GRAD: the atomic variant below) -- never reads slots
                                         that were not written */
#define GSASR_FLAG_BWD_ATOMIC 512u    /* tile-stationary with ONE fp32 atomic set per (tile, Gaussian) instead of the
                                         slots (the measured alternative of DESIGN.md 3c; order-dependent rounding)  */
#define GSASR_FLAG_BWD_HOME 32768u    /* home-tile backward (ABI 7): a workgroup owns the Gaussians BINNED in its tile of plan cells, stages
                                         the tile + a 16-px halo of grad_img once in LDS and finishes each of its Gaussians itself
                                         (items of 8 x 8 px per lane, added in LDS, one write per Gaussian): no slots, no gather, no
                                         atomics, deterministic.  Needs nothing from the plan beyond what the Gaussian-stationary
                                         kernel reads; a Gaussian whose window does not fit the region is swept by a whole wave
                                         (that kernel's code), so any input stays correct.  Interleaved [rows, w, 3] gradients.
                                         The default of whole images and batched canvases denser than one Gaussian per two pixels on
                                         at least 1024 tiles of 32 x 16 px (1024^2 at GSASR's 16 per LR pixel: -8..-9% against the
                                         Gaussian-stationary kernel) */
#define GSASR_FLAG_COUNTERS_CLEAN 1024u /* plan: the caller keeps this workspace between plans and promises that the
                                         per-cell counters of the parity given by GSASR_FLAG_PARITY are zero: the plan
                                         skips its memset launch.  Every plan zeroes the OTHER parity's counters on the
                                         side (in its first kernel), so a workspace that was used with parity p is clean
                                         for parity 1-p: alternate the bit plan by plan.  (First use of a workspace: leave
                                         CLEAN off.) */
#define GSASR_FLAG_PARITY 2048u        /* which of the two counter arrays this plan counts in */
#define GSASR_FLAG_CUTOFF_CAP 4096u    /* bounded op with an explicit dims.cutoff: treat it as an UPPER bound -- classes and dead set
                                         use it as given, the windows the data-derived tau' <= cutoff (gsasr_plan_cutoff).  For
                                         callers that must name the conservative tau themselves (the row-band exchange: the
                                         selection of halo Gaussians and the neighbour's plan have to agree on it) */
#define GSASR_FLAG_FWD_WIDE 8192u      /* forward kernel choice (default: the library picks by scale factor and image size): 16 x 16
                                         sub-tiles, four pixels per lane (scale factors from x5 up) ... */
#define GSASR_FLAG_FWD_NARROW 16384u   /* ... or 8 x 16 sub-tiles, two pixels per lane.  Read by gsasr_splat_forward only (the plan
                                         is the same).  _WIDE forces the 16 x 16 kernel on ANY single image (tests, A/B runs),
                                         a batched canvas ignores it; _NARROW leaves the choice among the 8 x 16 kernels to the
                                         image size.  Same sums in a different order: results agree to fp32 rounding */

typedef struct gsasr_dims {
    int s;        /* number of Gaussians                                              */
    int h, w;     /* FULL HR grid (pixel centres at 2*i/(n-1)-1, gs.cu:27-28)          */
    int c;        /* channels, must be 3                                              */
    float dmax;   /* < 0: unbounded (gs_cuda);  >= 0: |dx|,|dy| <= dmax box (gs_cuda_dmax) */
    int row0, row1; /* HR rows [row0,row1) owned by this call; 0,h for a single GPU   */
    float cutoff; /* support cutoff tau: a Gaussian is skipped for an 8x8 pixel tile when its
                     exponent is < -tau everywhere on it (|d| > sigma*sqrt(2 tau)).
                     0 -> process default (adaptive, see below); < 0 -> never skip
                     (every in-box term is summed, as the reference does).                */
    unsigned flags;
    /* Batched canvas (the reference splats a training batch sample by sample,
     * basicsr/models/gsasr_model.py:191-233).  All three zero = one image.  With batch = B > 1 the s Gaussians
     * are B samples of s/B each (sample-major) and the "image" is a canvas of B slots stacked vertically:
     * h = B * slot rows (slot a multiple of 16), w columns, row0 = 0, row1 = h.  Sample b has its OWN pixel grid
     * of sample_hw[2b] x sample_hw[2b+1] pixels (each <= slot x w; pixel centres 2i/(n-1)-1 of that size, so every
     * sample is computed exactly as a single-image call would) in the top-left corner of slot b; the rest of
     * the slot is padding (stored as 0 with GSASR_FLAG_OVERWRITE_IMAGE, like the reference's F.pad).  Image
     * layouts: [B*slot, w, 3], or with GSASR_FLAG_CHW_IMAGE [B, 3, slot, w].  Not combinable with row bands. */
    int batch;
    int slot;
    const int *sample_hw; /* HOST array [2*batch]: (h_b, w_b) per sample, read during the call */
    int grad_rows;        /* batched canvas + GSASR_FLAG_CHW_GRAD: rows per plane of grad_img [B, 3, grad_rows, w]
                             (>= every h_b; 0 = slot), so that the [B,3,Hmax,Wmax] gradient autograd returns is read in place */
    int list_cap;         /* tile lists (ABI 5): the plan appends every Gaussian of the normal class to the hit list of each
                             32 x 16-px (wide forward: 32 x 32) tile its ellipse reaches, and the forward renders from those
                             lists instead of searching the cells around each tile.  0 = the library decides: lists for
                             DENSE plans (at least one Gaussian per four pixels -- GSASR's 16 per LR pixel up to x8 -- where
                             they save ~5% of a step), sized four times what GSASR-shaped Gaussians fill, and the search
                             kernels elsewhere (at one Gaussian per LR pixel the plan's list atomics cost more than the
                             search); > 0 = lists with this many entries per tile (rounded up to 64) on any image; < 0 = no
                             lists.  A tile whose list overflows is rendered by the search: same image, only slower.  Part of
                             the workspace layout: pass the same value to every call on a plan */
} gsasr_dims;

#define GSASR_MAX_BATCH 64

/* Default tau is ADAPTIVE and keeps the sum of the skipped terms on any pixel below
 * GSASR_SPLAT_DEFAULT_EPS * max|colour| for ANY input: a bound RELATIVE to the colour scale.  After the host prologue
 * (sigmoid * alpha) colours are <= 1 and the bound is 1e-5 absolute -- an order below the 1e-4 parity tolerance and at the
 * level of fp32 summation noise; the raw op accepts any float as a colour, and there the bound scales with it exactly as the
 * fp32 rounding of the sum itself does (tests/test_hip_parity.py::test_raw_op_colours_far_above_one).  Gradients lose the
 * same tail: relative error ~ tau * exp(-tau) < 1e-7.
 *   conservative   tau = ln(s / eps), clamped to [16, 104]: every skipped term is < exp(-tau) times its colour and at most s
 *                  terms can be skipped on one pixel.  Classes, the dead set and the halo selection of the multi-GPU exchange
 *                  use it.  (s = 65 536 -> 22.6; s = 2^20 -> 25.4.)
 *   data-derived   the WINDOWS are built with tau' = ln(K / budget) <= tau, where K bounds the live Gaussians that can lose
 *                  a non-negligible term on one pixel and is counted on the device from the plan's own cell histogram:
 *                  (largest cell count) x (cells that can hold such a Gaussian) + (large class), the cells being the
 *                  smaller of (a) those a dmax box around a pixel touches (bounded op: utils/gs_cuda_dmax/gs.cu:41-50, only
 *                  a Gaussian whose box covers the pixel adds anything) and (b) those within the class' largest support of
 *                  the pixel (both ops); `budget` is eps minus exp(-tau) for every term of the geometric tail of everything
 *                  farther and for every dead Gaussian whose tails the op would still add to these rows.  Config
 *                  2: K = 336, tau' = 17.3; x8 (config 4): K = 392, tau' = 17.5.  gsasr_plan_cutoff reports both;
 *                  gsasr_amd/csrc/gsasr_splat.hip (adapt_kcut) has the derivation, tests/test_adaptive_cutoff.py the checks
 *                  incl. adversarial inputs (everything stacked on one spot: K ~ s, tau' = tau).
 * A fixed tau can be set per call (dims.cutoff) or per process (gsasr_set_default_cutoff / environment
 * GSASR_SPLAT_CUTOFF) and is used as given (GSASR_FLAG_CUTOFF_CAP: as an upper bound for the data-derived one):
 * tau = 104 (GSASR_SPLAT_EXACT_CUTOFF) skips only terms for which fp32 expf() in the
 * reference returns exactly +0 (exp(-104) < 2^-150), i.e. it sums the same set of non-zero terms as the
 * reference; tau < 0 never skips. */
#define GSASR_SPLAT_DEFAULT_EPS 1e-5f
#define GSASR_SPLAT_EXACT_CUTOFF 104.0f
/* The BACKWARD's own cutoff under the adaptive default (round 6).  The forward's tau' grows with K because the skipped terms of
 * up to K Gaussians add up on ONE pixel (GSASR's 16 Gaussians per LR pixel: K ~ 7 000, tau' = 20.4).  A Gaussian's gradient is a
 * sum over ITS OWN pixels only -- nothing accumulates across Gaussians -- so its window needs no more than the tau at which the
 * integrals themselves are complete: outside the ellipse {exponent >= -tau} lies exp(-tau) of a Gaussian's mass,
 * (2 / sqrt(pi)) sqrt(tau) exp(-tau) of its first and (1 + tau) exp(-tau) of its second moments (the d/dmu, d/dsigma, d/drho
 * integrands): 1.1e-7, 5e-7 and 1.9e-6 at tau = 16 -- the LOWEST value the data-derived tau' of the forward ever takes (sparse
 * plans), i.e. the per-Gaussian gradient accuracy every plan with few Gaussians per pixel has always had.  The
 * Gaussian-stationary and home-tile backward sweep the window of min(tau', GSASR_SPLAT_GRAD_TAU), whatever K is.  (14.3 --
 * "complete to 1e-5" -- was measured first: another 6% on the backward, and on stacked Gaussians 1.3e-5 of the tensor's
 * max-abs on d/dmu, whose odd integrand leaves a signed value far below the unsigned mass the truncation is relative to: no
 * margin under the per-Gaussian parity bar.)  Whole images and batched canvases only (a row band's share of a gradient may be
 * all tail); an explicit cutoff (dims.cutoff, GSASR_SPLAT_CUTOFF) is used as given by both directions. */
#define GSASR_SPLAT_GRAD_TAU 16.0f

GSASR_API int gsasr_abi_version(void);
GSASR_API const char *gsasr_last_error(void);

/* Bytes of scratch the plan needs for these dims (0 on bad dims). 256-byte aligned base required. */
GSASR_API size_t gsasr_splat_workspace_bytes(const gsasr_dims *dims);

/* Bin the Gaussians for [row0,row1) into `workspace` (classify -> scan -> scatter -> pack). */
GSASR_API int gsasr_splat_plan(const float *sigmas /*[s,3]*/, const float *coords /*[s,2]*/,
                     const float *colors /*[s,3]*/, const gsasr_dims *dims, void *workspace,
                     size_t workspace_bytes, void *stream);

/* img[row1-row0, w, 3] += splat (= splat with GSASR_FLAG_OVERWRITE_IMAGE).  `workspace` must hold the
 * plan of the same inputs and dims. */
GSASR_API int gsasr_splat_forward(const gsasr_dims *dims, const void *workspace, size_t workspace_bytes,
                        float *img, void *stream);

/* g_* += d(sum(grad_img*img))/d{sigmas,coords,colors} over rows [row0,row1).  Outputs must be
 * zero-initialised by the caller when a plain gradient is wanted (the reference wrapper does
 * torch.zeros_like, gs_cuda_dmax/gswrapper.py:40-42), unless GSASR_FLAG_OVERWRITE_GRADS is set. */
GSASR_API int gsasr_splat_backward(const float *sigmas, const float *coords, const float *colors,
                         const float *grad_img /*[row1-row0, w, 3]*/, float *g_sigmas,
                         float *g_coords, float *g_colors, const gsasr_dims *dims,
                         const void *workspace, size_t workspace_bytes, void *stream);

/* Fused host prologue of the path: raw decoder output gs_parameters[n,9] =
 * [sigma_x, sigma_y, rho, alpha, r, g, b, mu_x, mu_y] -> the kernel-frame tensors handed to the splat,
 * i.e. the activations of utils/gaussian_splatting.py:174-180 followed by the conversion of :121-123
 * (x/y swap of the sigmas, align-corners fix of the means, colour * alpha) in ONE kernel instead of ~15
 * elementwise launches.  `step_size` points at one float in DEVICE memory (default_step_size / scale, which
 * the reference holds as a 0-dim tensor), so no host sync is needed.  The backward applies the chain rule
 * and overwrites g_parameters[n,9]. */
GSASR_API int gsasr_prologue_forward(const float *gs_parameters, const float *step_size, int n, int h, int w,
                           float *sigmas, float *coords, float *colors, void *stream);
GSASR_API int gsasr_prologue_backward(const float *gs_parameters, const float *step_size, int n, int h, int w,
                            const float *g_sigmas, const float *g_coords, const float *g_colors,
                            float *g_parameters, void *stream);

/* The whole host-API step in one call each way (what generate_2D_gaussian_splatting_step enqueues):
 * prologue + plan + forward, and splat-backward + prologue-backward.  The workspace additionally holds the
 * kernel-frame tensors and their gradients (gsasr_step_workspace_bytes >= gsasr_splat_workspace_bytes).
 * Forward honours dims.flags (OVERWRITE_IMAGE, CHW_IMAGE); backward always stores g_parameters[n,9] and
 * reads grad_img as [row1-row0, w, 3], or with GSASR_FLAG_CHW_GRAD as planar [3, row1-row0, w] (batched canvas:
 * [B, 3, grad_rows, w], grad_rows >= every sample's height).  step_size = NULL in the backward: the step sizes the
 * forward's prologue used (kept in the workspace) -- the counterpart of the _sm forward below.
 *
 * gsasr_step_forward_sm: the reference's `scale_modify` calling convention (utils/gaussian_splatting.py:166-171:
 * `assert scale_modify[0] == scale_modify[1]`, step = default_step_size / scale_modify[0]) evaluated on the DEVICE by the
 * plan's first kernel.  scale_modify = device floats, sample b's pair at scale_modify[b * sm_stride + {0, 1}]
 * (sm_stride >= 2; one image: b = 0).  mismatch = device int[2] or NULL: set to {1 + b, bits of scale_modify[b][0]} when
 * a pair differs and never cleared here -- the caller reads it when convenient (no host synchronisation per call). */
GSASR_API size_t gsasr_step_workspace_bytes(const gsasr_dims *dims);
GSASR_API int gsasr_step_forward(const float *gs_parameters, const float *step_size, const gsasr_dims *dims, void *workspace,
                       size_t workspace_bytes, float *img, void *stream);
GSASR_API int gsasr_step_forward_sm(const float *gs_parameters, const float *scale_modify, int sm_stride, float default_step_size,
                          int *mismatch, const gsasr_dims *dims, void *workspace, size_t workspace_bytes, float *img,
                          void *stream);
GSASR_API int gsasr_step_backward(const float *gs_parameters, const float *step_size, const float *grad_img,
                        float *g_parameters, const gsasr_dims *dims, void *workspace, size_t workspace_bytes,
                        void *stream);

/* Sampled pixels.  With `sample_coords` the reference renders the whole [3,H,W] image and
 * then picks the S requested pixels out of it, one indexing op per point (utils/gaussian_splatting.py:214-216;
 * the points come from basicsr/data/continuous_bicubic_downsample_dataset.py:86-88).  These entry points evaluate
 * only those pixels: forward = the values out[3, n_points] at the points (one wave per point, its lanes spread over
 * the Gaussians binned within reach), backward = the gradient of sum(grad_out * out) (one wave per Gaussian, its
 * lanes spread over the points inside its window, found through a counting sort of the points).
 *
 *   points    device int32 [n_points, 2] = (row, column) on the image's own grid; negative values wrap once like
 *             Python indices; a point that is still out of range yields 0 and takes no part in the backward
 *             (the reference raises IndexError -- checking would cost a host synchronisation).  Repeated points
 *             are evaluated independently, as the reference's gather does.
 *   out / grad_out   [3, n_points], written (not accumulated).
 *   batched canvas (dims.batch = B): points [B, n_points, 2] on each sample's own grid, out [B, 3, n_points].
 *   workspace a plan of the same dims (gsasr_splat_plan, or the step entry point below); the whole image
 *             (row0 = 0, row1 = h).  Flags: OVERWRITE_GRADS / STRIDE8 as for gsasr_splat_backward.
 *   sample_ws scratch of gsasr_sample_workspace_bytes(dims, n_points) bytes, 256-byte aligned: the sorted points.
 *             The backward re-sorts `points`, or, given points = NULL, uses what the forward call left there.
 * The step variants fuse the host prologue exactly as gsasr_step_forward / gsasr_step_backward do. */
GSASR_API size_t gsasr_sample_workspace_bytes(const gsasr_dims *dims, int n_points);
GSASR_API int gsasr_splat_sample_forward(const gsasr_dims *dims, const void *workspace, size_t workspace_bytes, const int *points,
                               int n_points, float *out, void *sample_ws, size_t sample_ws_bytes, void *stream);
GSASR_API int gsasr_splat_sample_backward(const float *sigmas, const float *coords, const float *colors, const float *grad_out,
                                float *g_sigmas, float *g_coords, float *g_colors, const gsasr_dims *dims,
                                const void *workspace, size_t workspace_bytes, const int *points, int n_points,
                                void *sample_ws, size_t sample_ws_bytes, void *stream);
GSASR_API int gsasr_step_sample_forward(const float *gs_parameters, const float *step_size, const gsasr_dims *dims, void *workspace,
                              size_t workspace_bytes, const int *points, int n_points, float *out, void *sample_ws,
                              size_t sample_ws_bytes, void *stream);
GSASR_API int gsasr_step_sample_forward_sm(const float *gs_parameters, const float *scale_modify, int sm_stride,
                                 float default_step_size, int *mismatch, const gsasr_dims *dims, void *workspace,
                                 size_t workspace_bytes, const int *points, int n_points, float *out, void *sample_ws,
                                 size_t sample_ws_bytes, void *stream);
GSASR_API int gsasr_step_sample_backward(const float *gs_parameters, const float *step_size, const float *grad_out,
                               float *g_parameters, const gsasr_dims *dims, void *workspace, size_t workspace_bytes,
                               const int *points, int n_points, void *sample_ws, size_t sample_ws_bytes, void *stream);

/* Reference-shaped launchers: the argument lists of `_gs_render` / `_gs_render_backward` in utils/gs_cuda/gs.h:4-24 and
 * utils/gs_cuda_dmax/gs.h:4-26 (+ the stream, + a status instead of void).  The reference's launchers take no workspace,
 * so these keep their plan scratch per (device, stream) between calls (allocated stream-ordered on first use, reused in
 * stream order, re-planned on every call -- a backward never trusts the plan of an earlier forward: the arrays may have
 * changed); gsasr_release_launcher_scratch() frees what they hold. */
GSASR_API int gsasr_release_launcher_scratch(void);
GSASR_API int gsasr_gs_render(const float *sigmas, const float *coords, const float *colors,
                    float *rendered_img, int s, int h, int w, int c, void *stream);
GSASR_API int gsasr_gs_render_backward(const float *sigmas, const float *coords, const float *colors,
                             const float *grads, float *grads_sigmas, float *grads_coords,
                             float *grads_colors, int s, int h, int w, int c, void *stream);
GSASR_API int gsasr_gs_render_dmax(const float *sigmas, const float *coords, const float *colors,
                         float *rendered_img, int s, int h, int w, int c, float dmax, void *stream);
GSASR_API int gsasr_gs_render_backward_dmax(const float *sigmas, const float *coords, const float *colors,
                                  const float *grads, float *grads_sigmas, float *grads_coords,
                                  float *grads_colors, int s, int h, int w, int c, float dmax,
                                  void *stream);

/* Row-band shard (one process per GPU): neighbour exchange of the Gaussians whose footprint
 * crosses an edge of this rank's band.  The reference has no counterpart (its rasterizer is rank-local,
 * basicsr/models/base_model.py:96-99); these two kernels are the device side of gsasr_amd/shard.py.
 *
 * gsasr_band_select: `packed` is this rank's [s,8] Gaussians, dims = FULL grid h,w with [row0,row1) the
 * rank's band, dmax/cutoff as for the plan (the footprint is the plan's own row window: box ∩ support).
 * Gaussians reaching rows < row0 are appended to up[cap,8] (+ their index to up_index[cap]), those reaching
 * rows >= row1 to down/down_index; unused slots are filled with NaN records, which every kernel of this
 * library treats as dead Gaussians.  counts[4] (device) = {n_up, n_down, n_far, 0}: n_up/n_down may exceed
 * `cap` (overflow: the excess was dropped -- the caller must check and re-run with a larger cap), n_far =
 * Gaussians reaching beyond the adjacent band (more than rows_above above row0 / rows_below below row1),
 * which a nearest-neighbour exchange cannot serve.  rows_above/rows_below = 0: no neighbour on that side.
 *
 * gsasr_band_merge: g_packed[index[j], :] += g_up[j, :] for j < min(n_up, cap), same for down: adds the
 * partial gradients a neighbour computed for the records it was sent. */
GSASR_API int gsasr_band_select(const float *packed, const gsasr_dims *dims, int rows_above, int rows_below, int cap,
                      float *up, float *down, int *up_index, int *down_index, int *counts, void *stream);
GSASR_API int gsasr_band_merge(float *g_packed, int s, const float *g_up, const float *g_down, const int *up_index,
                     const int *down_index, const int *counts, int cap, void *stream);

/* Process-wide default used when dims.cutoff == 0: 0 = adaptive (initial state), otherwise a fixed tau
 * (initially the value of the environment variable GSASR_SPLAT_CUTOFF if set).  gsasr_resolve_cutoff
 * returns the tau a plan with dims.cutoff = `cutoff` over `s` Gaussians uses. */
GSASR_API void gsasr_set_default_cutoff(float tau);
GSASR_API float gsasr_get_default_cutoff(void);
GSASR_API float gsasr_resolve_cutoff(float cutoff, int s);

/* The cutoff the windows of the plan in `workspace` were actually built with (synchronises `stream`; for reports and
 * tests): under the adaptive default the data-derived tau' described above, *k_box = its K (0 when the cutoff is not
 * data-derived: explicit tau without GSASR_FLAG_CUTOFF_CAP, GSASR_SPLAT_ADAPT=0). */
GSASR_API int gsasr_plan_cutoff(const gsasr_dims *dims, const void *workspace, size_t workspace_bytes, void *stream,
                                float *tau, unsigned *k_box);

/* Which forward kernel gsasr_splat_forward runs for these dims (flags included; no device work; for reports and tests):
 * the width of a wave's sub-tile in pixels -- 16 = the wide forward (16 x 16 sub-tiles, four pixels per lane: scale factors
 * from x5 up on single images of 2 Mpx and more, or GSASR_FLAG_FWD_WIDE), 8 = the 8 x 16 kernels.  < 0: invalid dims. */
GSASR_API int gsasr_forward_subtile_width(const gsasr_dims *dims);

/* Kernel choices registered per problem shape (round 5; gsasr_amd/tune.py measures and registers them).
 * The library picks its kernels -- 8 x 16 or 16 x 16 forward sub-tiles, Gaussian- or tile-stationary backward, tile lists or the
 * search -- from the SHAPE of the problem (pixels per Gaussian, image size): the window sizes that really decide live on the device,
 * and no entry point synchronises to read them.  On Gaussians much smaller or larger than "about one LR pixel" another
 * combination can be 5..45% faster (profiles/history/r05_policy_regret.txt).  A caller who knows better -- it timed the combinations on
 * its own data -- registers the choice once; every later plan / forward / backward whose dims match `shape` in {s, h, w, row0,
 * row1, batch, slot, dmax, cutoff, GSASR_FLAG_FORWARD_ONLY} and carry NO explicit choice of their own (flags below, list_cap != 0)
 * behaves as if it had been given
 *     flags    : any of GSASR_FLAG_FWD_WIDE | GSASR_FLAG_FWD_NARROW, GSASR_FLAG_BWD_TILE | GSASR_FLAG_BWD_GAUSSIAN
 *     list_cap : as gsasr_dims.list_cap (0 = the library's rule, > 0 entries per tile, < 0 no lists).
 * Results are the same sums in another order.  Register BEFORE sizing workspaces for the shape (gsasr_splat_workspace_bytes
 * follows the registered choice; a workspace sized earlier may be too small and is refused, never overrun).  Process-wide,
 * thread-safe; at most 256 shapes (GSASR_ERR_ARG beyond, or on flags outside the four above). */
GSASR_API int gsasr_set_kernel_choice(const gsasr_dims *shape, unsigned flags, int list_cap);
/* 1 and the registered values if `shape` has a registered choice, else 0 */
GSASR_API int gsasr_get_kernel_choice(const gsasr_dims *shape, unsigned *flags, int *list_cap);
GSASR_API void gsasr_clear_kernel_choices(void);

#ifdef __cplusplus
}
#endif
#endif /* GSASR_SPLAT_H */
