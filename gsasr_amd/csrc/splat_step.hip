// splat_step.hip -- the reference's host prologue fused with the rasterizer: prologue kernels, gsasr_prologue_*, gsasr_step_*
// (one translation unit of libgsasr_splat.so; gsasr_splat.hip has the overview of the whole pipeline)
#include "splat_common.h"

using namespace gsasr_detail;

namespace {

// ---------------------------------------------------------------------------------------------------
// fused host prologue (reference utils/gaussian_splatting.py:174-180 and :121-123) and its backward
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_prologue_fwd(const float *__restrict__ p, const float *__restrict__ step_ptr,
                                                      int n, int h, int w, float *__restrict__ sigmas,
                                                      float *__restrict__ coords, float *__restrict__ colors,
                                                      int nper, const int4 *__restrict__ geo)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (geo && h == 0) {  // batched canvas: sample i/nper has its own size and step size (h, w given: one size for all)
        const int4 g = geo[i / nper];
        h = g.x;
        w = g.y;
    }
    float o[8];
    prologue_one(p + (size_t)i * 9, step_ptr[geo ? i / nper : 0], h, w, o);
    sigmas[i * 3 + 0] = o[0]; sigmas[i * 3 + 1] = o[1]; sigmas[i * 3 + 2] = o[2];
    coords[i * 2 + 0] = o[3]; coords[i * 2 + 1] = o[4];
    colors[i * 3 + 0] = o[5]; colors[i * 3 + 1] = o[6]; colors[i * 3 + 2] = o[7];
}

// chain rule of k_prologue_fwd for one Gaussian: q = its raw parameters, gs/gc/gk = d/d{sigmas, coords, colors}
__device__ __forceinline__ void prologue_chain(const float *__restrict__ q, float step, int h, int w, float gs0, float gs1,
                                               float gs2, float gc0, float gc1, float k0, float k1, float k2,
                                               float *__restrict__ o)
{
    const float W = (float)w, H = (float)h;
    const float s0 = sigmoidf_(q[0]), s1 = sigmoidf_(q[1]), th = tanhf(q[2]), al = sigmoidf_(q[3]);
    const float r = sigmoidf_(q[4]), g = sigmoidf_(q[5]), b = sigmoidf_(q[6]);
    o[0] = gs1 * (2.f / (H - 1.f) / step) * 0.99999f * s0 * (1.f - s0);
    o[1] = gs0 * (2.f / (W - 1.f) / step) * 0.99999f * s1 * (1.f - s1);
    o[2] = gs2 * 0.999999f * (1.f - th * th);
    o[3] = (k0 * r + k1 * g + k2 * b) * al * (1.f - al);
    o[4] = k0 * al * r * (1.f - r);
    o[5] = k1 * al * g * (1.f - g);
    o[6] = k2 * al * b * (1.f - b);
    o[7] = gc0 * 2.f * W / (W - 1.f);
    o[8] = gc1 * 2.f * H / (H - 1.f);
}

__global__ __launch_bounds__(256) void k_prologue_bwd(const float *__restrict__ p, const float *__restrict__ step_ptr,
                                                      int n, int h, int w, const float *__restrict__ gs,
                                                      const float *__restrict__ gc, const float *__restrict__ gk,
                                                      float *__restrict__ gp, int nper, const int4 *__restrict__ geo)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (geo && h == 0) {
        const int4 g = geo[i / nper];
        h = g.x;
        w = g.y;
    }
    const float step = step_ptr[geo ? i / nper : 0];
    prologue_chain(p + (size_t)i * 9, step, h, w, gs[i * 3 + 0], gs[i * 3 + 1], gs[i * 3 + 2], gc[i * 2 + 0], gc[i * 2 + 1],
                   gk[i * 3 + 0], gk[i * 3 + 1], gk[i * 3 + 2], gp + (size_t)i * 9);
}

// the same behind the tile-stationary backward: the gather of the partial-gradient slots (bwd_gather) and the chain rule
// in one kernel, one thread per Gaussian in cell order -- the kernel-frame gradients never go through memory
__global__ __launch_bounds__(256) void k_prologue_bwd_gather(Params P, PlanView V, int use_atomics, const float *__restrict__ p,
                                                             const float *__restrict__ step_ptr, float *__restrict__ gp)
{
    const unsigned j = blockIdx.x * 256u + threadIdx.x;
    if (j >= (unsigned)P.s) return;
    float o[8];
    const unsigned i = bwd_gather(P, V, j, use_atomics != 0, o);
    int h = P.h, w = P.w;
    float step = step_ptr[0];
    if (P.batch > 1) {
        const Geo g = sample_geo(P, V, (int)(i / (unsigned)P.nper));
        h = g.h;
        w = g.w;
        step = step_ptr[i / (unsigned)P.nper];
    }
    prologue_chain(p + (size_t)i * 9, step, h, w, o[2], o[3], o[4], o[0], o[1], o[5], o[6], o[7], gp + (size_t)i * 9);
}

// planar [3, rows, w] (batched canvas: [B, 3, grad_rows, w], sample b's rows at the top of its planes) -> interleaved
// [rows, w, 3] / [B * slot, w, 3]: what autograd hands back for the planar image -> what k_render_bwd sweeps.  One
// thread per pixel: three coalesced plane reads, one 12-byte store.  Rows of a slot beyond grad_rows are left alone:
// the backward never reads outside a sample's own grid.
__global__ __launch_bounds__(256) void k_chw_to_hwc(const float *__restrict__ src, float *__restrict__ dst, int w, int rows,
                                                    int batch, int slot, int grad_rows)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const int X = (int)(i % (size_t)w);
    const size_t Y = i / (size_t)w;
    if (Y >= (size_t)rows) return;
    size_t plane = (size_t)rows * w, at = Y * w + X;
    if (batch > 1) {
        const int b = (int)(Y / (size_t)slot), y = (int)(Y - (size_t)b * slot);
        if (y >= grad_rows) return;
        plane = (size_t)grad_rows * w;
        at = (size_t)b * 3 * plane + (size_t)y * w + X;
    }
    float *o = dst + (Y * w + X) * 3;
    o[0] = src[at];
    o[1] = src[at + plane];
    o[2] = src[at + 2 * plane];
}

}  // namespace

extern "C" {

int gsasr_prologue_forward(const float *gs_parameters, const float *step_size, int n, int h, int w, float *sigmas,
                           float *coords, float *colors, void *stream)
{
    if (n < 0 || h < 2 || w < 2) return fail(GSASR_ERR_ARG, "bad n/h/w");
    if (n == 0) return GSASR_OK;
    if (!gs_parameters || !step_size || !sigmas || !coords || !colors) return fail(GSASR_ERR_ARG, "null pointer");
    hipLaunchKernelGGL(k_prologue_fwd, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       gs_parameters, step_size, n, h, w, sigmas, coords, colors, 0, (const int4 *)nullptr);
    HIP_TRY(hipGetLastError());
    return GSASR_OK;
}

int gsasr_prologue_backward(const float *gs_parameters, const float *step_size, int n, int h, int w,
                            const float *g_sigmas, const float *g_coords, const float *g_colors, float *g_parameters,
                            void *stream)
{
    if (n < 0 || h < 2 || w < 2) return fail(GSASR_ERR_ARG, "bad n/h/w");
    if (n == 0) return GSASR_OK;
    if (!gs_parameters || !step_size || !g_sigmas || !g_coords || !g_colors || !g_parameters)
        return fail(GSASR_ERR_ARG, "null pointer");
    hipLaunchKernelGGL(k_prologue_bwd, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       gs_parameters, step_size, n, h, w, g_sigmas, g_coords, g_colors, g_parameters, 0, (const int4 *)nullptr);
    HIP_TRY(hipGetLastError());
    return GSASR_OK;
}

// ---- whole-step entry points ----------------------------------------------------------------------
}  // extern "C"

namespace gsasr_detail {
StepLayout make_step_layout(const gsasr_dims *d, const void *planned_ws)
{
    StepLayout S;
    const size_t n = (size_t)d->s;
    S.plan_bytes = (planned_ws ? plan_layout(d, planned_ws) : make_layout(d)).total;   // (the plan's own slot count: plan_layout)
    size_t o = S.plan_bytes;
    S.off_step = o; o += align_up(GSASR_MAX_BATCH * 4, 256);   // the step size of every sample, as the prologue used it
    S.off_sig = o;  o += align_up(n * 12, 256);
    S.off_xy = o;   o += align_up(n * 8, 256);
    S.off_col = o;  o += align_up(n * 12, 256);
    const size_t nb = (d->flags & GSASR_FLAG_FORWARD_ONLY) ? 0 : n;   // (no gradient scratch for a forward-only step)
    S.off_gsig = o; o += align_up(nb * 12, 256);
    S.off_gxy = o;  o += align_up(nb * 8, 256);
    S.off_gcol = o; o += align_up(nb * 12, 256);
    // a planar upstream gradient (GSASR_FLAG_CHW_GRAD) in front of the Gaussian-stationary backward is interleaved into
    // this scratch by k_chw_to_hwc (the tile-stationary backward stages the planes directly and needs none)
    S.off_ghwc = o;
    if ((d->flags & GSASR_FLAG_CHW_GRAD) && !(d->flags & (GSASR_FLAG_BWD_TILE | GSASR_FLAG_BWD_ATOMIC | GSASR_FLAG_FORWARD_ONLY)))
        o += align_up((size_t)(d->row1 - d->row0) * (size_t)d->w * 12, 256);
    S.total = o;
    return S;
}

// chain rule of the prologue on a batched canvas (per-sample grid sizes): shared by the step and the sampled-step backward
int prologue_backward_batched(const float *gs_parameters, const float *step_size, const gsasr_dims *dims, void *workspace,
                              const float *gs, const float *gc, const float *gk, float *g_parameters, void *stream)
{
    const PlanView V = make_view(make_layout(dims), workspace);
    int uh, uw;
    batch_uniform(dims, uh, uw);
    hipLaunchKernelGGL(k_prologue_bwd, dim3((unsigned)((dims->s + 255) / 256)), dim3(256), 0, (hipStream_t)stream, gs_parameters,
                       step_size, dims->s, uh, uw, gs, gc, gk, g_parameters, dims->s / dims->batch, (const int4 *)V.geo);
    HIP_TRY(hipGetLastError());
    return GSASR_OK;
}
}  // namespace gsasr_detail

extern "C" {

size_t gsasr_step_workspace_bytes(const gsasr_dims *dims)
{
    if (!dims_ok(dims)) {
        fail(GSASR_ERR_ARG, "bad dims");
        return 0;
    }
    return make_step_layout(dims).total;
}

}  // extern "C"

namespace gsasr_detail {
// prologue (per-sample sizes and step sizes on a batched canvas) + plan of a whole-step call
int step_prologue_plan(const float *gs_parameters, StepSrc SS, const gsasr_dims *dims, void *workspace,
                       size_t workspace_bytes, void *stream, StepLayout &S)
{
    if (!dims_ok(dims)) return fail(GSASR_ERR_ARG, "bad dims");
    if (dims->flags & GSASR_FLAG_STRIDE8) return fail(GSASR_ERR_ARG, "GSASR_FLAG_STRIDE8 does not apply to the step entry points");
    S = make_step_layout(dims);
    if (!workspace || ((uintptr_t)workspace & 255u) || workspace_bytes < S.total)
        return fail(GSASR_ERR_WORKSPACE, "workspace null, misaligned or smaller than gsasr_step_workspace_bytes()");
    char *b = (char *)workspace;
    float *sig = (float *)(b + S.off_sig), *xy = (float *)(b + S.off_xy), *col = (float *)(b + S.off_col);
    if (dims->s > 0 && (!gs_parameters || (!SS.step && !SS.sm))) return fail(GSASR_ERR_ARG, "null pointer");
    if (SS.sm && SS.stride < 2) return fail(GSASR_ERR_ARG, "scale_modify stride must be >= 2");
    SS.keep = (float *)(b + S.off_step);
    if (dims->batch > 1 && dims->s > 0) {  // the per-sample geometry must be in place before the classify kernel reads it
        const PlanView V = make_view(make_layout(dims), workspace, dims->flags);
        if (int rc = launch_batch_geo(dims, V, (hipStream_t)stream)) return rc;
    }
    // (the prologue runs inside the plan's first kernel: k_classify<true>)
    return plan_impl(sig, xy, col, dims, workspace, S.plan_bytes, stream, dims->s > 0 ? gs_parameters : nullptr, SS);
}
}  // namespace gsasr_detail

extern "C" {

int gsasr_step_forward(const float *gs_parameters, const float *step_size, const gsasr_dims *dims, void *workspace,
                       size_t workspace_bytes, float *img, void *stream)
{
    StepLayout S;
    StepSrc SS{};
    SS.step = step_size;
    if (int rc = step_prologue_plan(gs_parameters, SS, dims, workspace, workspace_bytes, stream, S)) return rc;
    return gsasr_splat_forward(dims, workspace, S.plan_bytes, img, stream);
}

int gsasr_step_forward_sm(const float *gs_parameters, const float *scale_modify, int sm_stride, float default_step_size,
                          int *mismatch, const gsasr_dims *dims, void *workspace, size_t workspace_bytes, float *img,
                          void *stream)
{
    StepLayout S;
    StepSrc SS{};
    SS.sm = scale_modify; SS.stride = sm_stride; SS.def_step = default_step_size; SS.mismatch = mismatch;
    if (!scale_modify && dims && dims->s > 0) return fail(GSASR_ERR_ARG, "null pointer");
    if (int rc = step_prologue_plan(gs_parameters, SS, dims, workspace, workspace_bytes, stream, S)) return rc;
    return gsasr_splat_forward(dims, workspace, S.plan_bytes, img, stream);
}

int gsasr_step_backward(const float *gs_parameters, const float *step_size, const float *grad_img,
                        float *g_parameters, const gsasr_dims *dims, void *workspace, size_t workspace_bytes,
                        void *stream)
{
    if (!dims_ok(dims)) return fail(GSASR_ERR_ARG, "bad dims");
    if (dims->flags & GSASR_FLAG_STRIDE8) return fail(GSASR_ERR_ARG, "GSASR_FLAG_STRIDE8 does not apply to the step entry points");
    const StepLayout S = make_step_layout(dims, workspace);
    if (!workspace || ((uintptr_t)workspace & 255u) || workspace_bytes < S.total)
        return fail(GSASR_ERR_WORKSPACE, "workspace null, misaligned or smaller than gsasr_step_workspace_bytes()");
    char *b = (char *)workspace;
    float *sig = (float *)(b + S.off_sig), *xy = (float *)(b + S.off_xy), *col = (float *)(b + S.off_col);
    float *gs = (float *)(b + S.off_gsig), *gc = (float *)(b + S.off_gxy), *gk = (float *)(b + S.off_gcol);
    if (!step_size) step_size = (const float *)(b + S.off_step);   // what the forward's prologue used (gsasr_step_forward_sm)
    gsasr_dims d = *dims;
    d.flags |= GSASR_FLAG_OVERWRITE_GRADS;
    if ((d.flags & GSASR_FLAG_CHW_GRAD) && S.total > S.off_ghwc && dims->s > 0 && d.row1 > d.row0) {
        // Gaussian-stationary kernel behind a planar gradient: interleave it into the scratch first
        if (!grad_img) return fail(GSASR_ERR_ARG, "null pointer");
        float *hwc = (float *)(b + S.off_ghwc);
        const int rows = d.row1 - d.row0;
        const size_t px = (size_t)rows * d.w;
        hipLaunchKernelGGL(k_chw_to_hwc, dim3((unsigned)((px + 255) / 256)), dim3(256), 0, (hipStream_t)stream, grad_img, hwc,
                           d.w, rows, batch_of(&d), d.batch > 1 ? d.slot : rows, d.grad_rows > 0 ? d.grad_rows : (d.batch > 1 ? d.slot : rows));
        HIP_TRY(hipGetLastError());
        grad_img = hwc;
        d.flags &= ~GSASR_FLAG_CHW_GRAD;
        if (!(d.flags & GSASR_FLAG_BWD_HOME)) d.flags |= GSASR_FLAG_BWD_GAUSSIAN;      // (the home-tile kernel sweeps interleaved gradients too)
    }
    int mode = 0;
    if (int rc = splat_backward(sig, xy, col, grad_img, gs, gc, gk, &d, workspace, S.plan_bytes, stream, false, &mode)) return rc;
    if (dims->s == 0) return GSASR_OK;
    if (!gs_parameters || !step_size || !g_parameters) return fail(GSASR_ERR_ARG, "null pointer");
    const dim3 grid((unsigned)((dims->s + 255) / 256)), block(256);
    if (mode == 1 || mode == 2) {   // tile-stationary: gather of the slots + chain rule in one kernel
        const Layout L = plan_layout(dims, workspace);
        const PlanView V = make_view(L, workspace);
        hipLaunchKernelGGL(k_prologue_bwd_gather, grid, block, 0, (hipStream_t)stream, make_params(&d, L), V, (int)(mode == 2 || d.row1 == d.row0),
                           gs_parameters, step_size, g_parameters);
        HIP_TRY(hipGetLastError());
        return GSASR_OK;
    }
    if (dims->batch > 1) {
        return prologue_backward_batched(gs_parameters, step_size, dims, workspace, gs, gc, gk, g_parameters, stream);
    }
    return gsasr_prologue_backward(gs_parameters, step_size, dims->s, dims->h, dims->w, gs, gc, gk, g_parameters, stream);
}

}  // extern "C"
