/*
 * tests/c_abi/c_abi_check.c -- the drop-in boundary exercised from plain C: no Python, no torch.
 *
 * Links libgsasr_splat.so (the product) through include/gsasr_splat.h exactly the way a C/C++ maintainer
 * of the reference would after swapping gs.h's launchers, and checks the
 * results against the CPU oracle's double-precision truth (oracle/libgs_ref.so), both for the
 * reference-shaped launchers, the plan API (row band), a batched canvas and the band-exchange kernels.  Built and
 * run by tests/test_c_abi.py on the GPU box.
 */
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include "gsasr_splat.h"

void gsref_forward_f64(const float *, const float *, const float *, double *, int, int, int, float, int, int);
void gsref_backward_f64(const float *, const float *, const float *, const float *, double *, double *, double *, int,
                        int, int, float, int, int);

#define CK(x)                                                                      \
    do {                                                                           \
        hipError_t e_ = (x);                                                       \
        if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } \
    } while (0)
#define OK(x)                                                                      \
    do {                                                                           \
        int rc_ = (x);                                                             \
        if (rc_ != 0) { fprintf(stderr, "%s -> %d: %s\n", #x, rc_, gsasr_last_error()); return 3; } \
    } while (0)

static float frand(unsigned *s) { *s = *s * 1664525u + 1013904223u; return (float)((*s >> 8) & 0xffffff) / 16777216.0f; }

static double maxrel(const float *got, const double *want, int n)
{
    double m = 0, e = 0;
    for (int i = 0; i < n; ++i) { if (fabs(want[i]) > m) m = fabs(want[i]); }
    for (int i = 0; i < n; ++i) { double d = fabs((double)got[i] - want[i]); if (d > e) e = d; }
    return e / (m > 1e-12 ? m : 1e-12);
}

int main(void)
{
    const int s = 3000, h = 150, w = 211;
    unsigned seed = 7;
    float *sig = malloc(sizeof(float) * 3 * s), *xy = malloc(sizeof(float) * 2 * s), *col = malloc(sizeof(float) * 3 * s);
    float *wgt = malloc(sizeof(float) * 3 * h * w);
    for (int i = 0; i < s; ++i) {
        sig[3 * i + 0] = 0.004f + 0.05f * frand(&seed);
        sig[3 * i + 1] = 0.004f + 0.05f * frand(&seed);
        sig[3 * i + 2] = 1.8f * frand(&seed) - 0.9f;
        xy[2 * i + 0] = 2.1f * frand(&seed) - 1.05f;
        xy[2 * i + 1] = 2.1f * frand(&seed) - 1.05f;
        for (int k = 0; k < 3; ++k) col[3 * i + k] = frand(&seed);
    }
    sig[0] = 0.7f; sig[1] = 0.9f;  /* one image-spanning Gaussian (large class) */
    for (int i = 0; i < 3 * h * w; ++i) wgt[i] = frand(&seed);

    float *d_sig, *d_xy, *d_col, *d_img, *d_wgt, *d_gs, *d_gc, *d_gk;
    CK(hipMalloc((void **)&d_sig, sizeof(float) * 3 * s)); CK(hipMalloc((void **)&d_xy, sizeof(float) * 2 * s));
    CK(hipMalloc((void **)&d_col, sizeof(float) * 3 * s)); CK(hipMalloc((void **)&d_img, sizeof(float) * 3 * h * w));
    CK(hipMalloc((void **)&d_wgt, sizeof(float) * 3 * h * w)); CK(hipMalloc((void **)&d_gs, sizeof(float) * 3 * s));
    CK(hipMalloc((void **)&d_gc, sizeof(float) * 2 * s)); CK(hipMalloc((void **)&d_gk, sizeof(float) * 3 * s));
    CK(hipMemcpy(d_sig, sig, sizeof(float) * 3 * s, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_xy, xy, sizeof(float) * 2 * s, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_col, col, sizeof(float) * 3 * s, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_wgt, wgt, sizeof(float) * 3 * h * w, hipMemcpyHostToDevice));
    hipStream_t st;
    CK(hipStreamCreate(&st));

    float *img = malloc(sizeof(float) * 3 * h * w), *gs = malloc(sizeof(float) * 3 * s), *gc = malloc(sizeof(float) * 2 * s),
          *gk = malloc(sizeof(float) * 3 * s);
    double *ref = malloc(sizeof(double) * 3 * h * w), *rs = malloc(sizeof(double) * 3 * s), *rc = malloc(sizeof(double) * 2 * s),
           *rk = malloc(sizeof(double) * 3 * s);
    int bad = 0;
    const float dmaxs[2] = {0.3f, -1.f};
    for (int v = 0; v < 2; ++v) {
        const float dmax = dmaxs[v];
        /* reference-shaped launchers: gs.h's _gs_render / _gs_render_backward with a stream and a status */
        CK(hipMemsetAsync(d_img, 0, sizeof(float) * 3 * h * w, st));
        CK(hipMemsetAsync(d_gs, 0, sizeof(float) * 3 * s, st)); CK(hipMemsetAsync(d_gc, 0, sizeof(float) * 2 * s, st));
        CK(hipMemsetAsync(d_gk, 0, sizeof(float) * 3 * s, st));
        if (dmax >= 0) {
            OK(gsasr_gs_render_dmax(d_sig, d_xy, d_col, d_img, s, h, w, 3, dmax, st));
            OK(gsasr_gs_render_backward_dmax(d_sig, d_xy, d_col, d_wgt, d_gs, d_gc, d_gk, s, h, w, 3, dmax, st));
        } else {
            OK(gsasr_gs_render(d_sig, d_xy, d_col, d_img, s, h, w, 3, st));
            OK(gsasr_gs_render_backward(d_sig, d_xy, d_col, d_wgt, d_gs, d_gc, d_gk, s, h, w, 3, st));
        }
        CK(hipStreamSynchronize(st));
        CK(hipMemcpy(img, d_img, sizeof(float) * 3 * h * w, hipMemcpyDeviceToHost));
        CK(hipMemcpy(gs, d_gs, sizeof(float) * 3 * s, hipMemcpyDeviceToHost));
        CK(hipMemcpy(gc, d_gc, sizeof(float) * 2 * s, hipMemcpyDeviceToHost));
        CK(hipMemcpy(gk, d_gk, sizeof(float) * 3 * s, hipMemcpyDeviceToHost));
        gsref_forward_f64(sig, xy, col, ref, s, h, w, dmax, 0, h);
        gsref_backward_f64(sig, xy, col, wgt, rs, rc, rk, s, h, w, dmax, 0, h);
        double eimg = 0;
        for (int i = 0; i < 3 * h * w; ++i) { double d = fabs((double)img[i] - ref[i]); if (d > eimg) eimg = d; }
        const double e1 = maxrel(gs, rs, 3 * s), e2 = maxrel(gc, rc, 2 * s), e3 = maxrel(gk, rk, 3 * s);
        printf("launchers dmax=%g: image max|err| %.3e, grad rel err %.2e %.2e %.2e\n", dmax, eimg, e1, e2, e3);
        if (!(eimg <= 2e-4) || !(e1 <= 2e-4) || !(e2 <= 2e-4) || !(e3 <= 2e-4)) bad = 1;

        /* plan API, bottom half of the image as a row band, stored (not accumulated) outputs */
        gsasr_dims d = {s, h, w, 3, dmax, h / 2, h, 0.f, GSASR_FLAG_OVERWRITE_IMAGE | GSASR_FLAG_OVERWRITE_GRADS};
        const size_t bytes = gsasr_splat_workspace_bytes(&d);
        void *ws;
        CK(hipMalloc(&ws, bytes));
        OK(gsasr_splat_plan(d_sig, d_xy, d_col, &d, ws, bytes, st));
        OK(gsasr_splat_forward(&d, ws, bytes, d_img, st));
        OK(gsasr_splat_backward(d_sig, d_xy, d_col, d_wgt + (size_t)(h / 2) * w * 3, d_gs, d_gc, d_gk, &d, ws, bytes, st));
        CK(hipStreamSynchronize(st));
        const int rows = h - h / 2;
        CK(hipMemcpy(img, d_img, sizeof(float) * 3 * rows * w, hipMemcpyDeviceToHost));
        CK(hipMemcpy(gk, d_gk, sizeof(float) * 3 * s, hipMemcpyDeviceToHost));
        gsref_forward_f64(sig, xy, col, ref, s, h, w, dmax, h / 2, h);
        gsref_backward_f64(sig, xy, col, wgt + (size_t)(h / 2) * w * 3, rs, rc, rk, s, h, w, dmax, h / 2, h);
        eimg = 0;
        for (int i = 0; i < 3 * rows * w; ++i) { double dd = fabs((double)img[i] - ref[i]); if (dd > eimg) eimg = dd; }
        const double e4 = maxrel(gk, rk, 3 * s);
        printf("plan API  dmax=%g band [%d,%d): image max|err| %.3e, grad(colors) rel err %.2e\n", dmax, h / 2, h, eimg, e4);
        if (!(eimg <= 2e-4) || !(e4 <= 2e-4)) bad = 1;
        CK(hipFree(ws));

        /* the same band from the plan's tile lists (gsasr_dims.list_cap, ABI 5): forward rendered from the lists, the
         * tile-stationary backward reading them too; a bigger workspace, the same numbers */
        gsasr_dims dl = d;
        dl.flags |= GSASR_FLAG_BWD_TILE;
        dl.list_cap = 256;
        const size_t lbytes = gsasr_splat_workspace_bytes(&dl);
        if (!(lbytes > bytes)) { printf("list_cap did not grow the workspace\n"); bad = 1; }
        CK(hipMalloc(&ws, lbytes));
        OK(gsasr_splat_plan(d_sig, d_xy, d_col, &dl, ws, lbytes, st));
        OK(gsasr_splat_forward(&dl, ws, lbytes, d_img, st));
        OK(gsasr_splat_backward(d_sig, d_xy, d_col, d_wgt + (size_t)(h / 2) * w * 3, d_gs, d_gc, d_gk, &dl, ws, lbytes, st));
        CK(hipStreamSynchronize(st));
        CK(hipMemcpy(img, d_img, sizeof(float) * 3 * rows * w, hipMemcpyDeviceToHost));
        CK(hipMemcpy(gk, d_gk, sizeof(float) * 3 * s, hipMemcpyDeviceToHost));
        eimg = 0;
        for (int i = 0; i < 3 * rows * w; ++i) { double dd = fabs((double)img[i] - ref[i]); if (dd > eimg) eimg = dd; }
        const double e5 = maxrel(gk, rk, 3 * s);
        printf("tile lists dmax=%g band [%d,%d): image max|err| %.3e, grad(colors) rel err %.2e\n", dmax, h / 2, h, eimg, e5);
        if (!(eimg <= 2e-4) || !(e5 <= 2e-4)) bad = 1;
        CK(hipFree(ws));

        /* the same choice REGISTERED for the shape instead of written into the dims (gsasr_set_kernel_choice, ABI 6): the plain dims
         * `d` now size, plan and render like `dl`; explicit flags still win; clearing the registry restores the library's rule */
        {
            unsigned rf = 0; int rcap = 0;
            if (gsasr_get_kernel_choice(&d, &rf, &rcap) != 0) { printf("a choice before any registration\n"); bad = 1; }
            OK(gsasr_set_kernel_choice(&d, GSASR_FLAG_BWD_TILE | GSASR_FLAG_FWD_NARROW, 256));
            if (gsasr_get_kernel_choice(&d, &rf, &rcap) != 1 || rf != (GSASR_FLAG_BWD_TILE | GSASR_FLAG_FWD_NARROW) || rcap != 256) { printf("registered choice not returned\n"); bad = 1; }
            if (gsasr_set_kernel_choice(&d, GSASR_FLAG_OVERWRITE_IMAGE, 0) == GSASR_OK) { printf("a non-choice flag was accepted\n"); bad = 1; }
            const size_t rbytes = gsasr_splat_workspace_bytes(&d);
            if (rbytes != lbytes) { printf("registered choice: workspace %zu, explicit flags %zu\n", rbytes, lbytes); bad = 1; }
            gsasr_dims dg = d;
            dg.flags |= GSASR_FLAG_BWD_GAUSSIAN;
            dg.list_cap = -1;
            if (gsasr_splat_workspace_bytes(&dg) != bytes) { printf("explicit flags did not win over the registration\n"); bad = 1; }
            CK(hipMalloc(&ws, rbytes));
            OK(gsasr_splat_plan(d_sig, d_xy, d_col, &d, ws, rbytes, st));
            OK(gsasr_splat_forward(&d, ws, rbytes, d_img, st));
            OK(gsasr_splat_backward(d_sig, d_xy, d_col, d_wgt + (size_t)(h / 2) * w * 3, d_gs, d_gc, d_gk, &d, ws, rbytes, st));
            CK(hipStreamSynchronize(st));
            CK(hipMemcpy(img, d_img, sizeof(float) * 3 * rows * w, hipMemcpyDeviceToHost));
            CK(hipMemcpy(gk, d_gk, sizeof(float) * 3 * s, hipMemcpyDeviceToHost));
            eimg = 0;
            for (int i = 0; i < 3 * rows * w; ++i) { double dd = fabs((double)img[i] - ref[i]); if (dd > eimg) eimg = dd; }
            const double e6 = maxrel(gk, rk, 3 * s);
            printf("registered choice dmax=%g band [%d,%d): image max|err| %.3e, grad(colors) rel err %.2e\n", dmax, h / 2, h, eimg, e6);
            if (!(eimg <= 2e-4) || !(e6 <= 2e-4)) bad = 1;
            CK(hipFree(ws));
            gsasr_clear_kernel_choices();
            if (gsasr_get_kernel_choice(&d, &rf, &rcap) != 0 || gsasr_splat_workspace_bytes(&d) != bytes) { printf("clearing the registry did not restore the rule\n"); bad = 1; }
        }
    }
    /* batched canvas (gsasr_dims.batch): two samples of different size from the same Gaussians, kernel-frame inputs;
     * every sample must equal the oracle's single-image result on ITS grid, padding must be zero */
    {
        const int B = 2, hw[4] = {40, 56, 64, 37}, n = 400, slot = 64, wmax = 56;
        const float dmax = 0.25f;
        gsasr_dims d = {B * n, B * slot, wmax, 3, dmax, 0, B * slot, 0.f, GSASR_FLAG_OVERWRITE_IMAGE | GSASR_FLAG_OVERWRITE_GRADS,
                        B, slot, hw};
        const size_t bytes = gsasr_splat_workspace_bytes(&d);
        if (!bytes) { fprintf(stderr, "batched dims rejected: %s\n", gsasr_last_error()); return 4; }
        void *ws;
        float *d_can, *d_gcan;
        CK(hipMalloc(&ws, bytes));
        CK(hipMalloc((void **)&d_can, sizeof(float) * 3 * B * slot * wmax));
        CK(hipMalloc((void **)&d_gcan, sizeof(float) * 3 * B * slot * wmax));
        CK(hipMemcpy(d_gcan, wgt, sizeof(float) * 3 * B * slot * wmax, hipMemcpyHostToDevice));   /* incl. junk in the padding */
        /* the first 2n Gaussians of the arrays above: sample 0 = [0,n), sample 1 = [n,2n) */
        OK(gsasr_splat_plan(d_sig, d_xy, d_col, &d, ws, bytes, st));
        OK(gsasr_splat_forward(&d, ws, bytes, d_can, st));
        OK(gsasr_splat_backward(d_sig, d_xy, d_col, d_gcan, d_gs, d_gc, d_gk, &d, ws, bytes, st));
        CK(hipStreamSynchronize(st));
        float *can = malloc(sizeof(float) * 3 * B * slot * wmax);
        CK(hipMemcpy(can, d_can, sizeof(float) * 3 * B * slot * wmax, hipMemcpyDeviceToHost));
        CK(hipMemcpy(gk, d_gk, sizeof(float) * 3 * B * n, hipMemcpyDeviceToHost));
        for (int b = 0; b < B; ++b) {
            const int hb = hw[2 * b], wb = hw[2 * b + 1];
            gsref_forward_f64(sig + 3 * b * n, xy + 2 * b * n, col + 3 * b * n, ref, n, hb, wb, dmax, 0, hb);
            /* the sample's gradient image: its h_b x w_b corner of slot b */
            float *gsub = malloc(sizeof(float) * 3 * hb * wb);
            for (int i = 0; i < hb; ++i)
                for (int j = 0; j < 3 * wb; ++j) gsub[(size_t)i * 3 * wb + j] = wgt[((size_t)(b * slot + i) * wmax) * 3 + j];
            gsref_backward_f64(sig + 3 * b * n, xy + 2 * b * n, col + 3 * b * n, gsub, rs, rc, rk, n, hb, wb, dmax, 0, hb);
            double eimg = 0, epad = 0;
            for (int i = 0; i < slot; ++i)
                for (int j = 0; j < wmax; ++j)
                    for (int k = 0; k < 3; ++k) {
                        const double got = can[((size_t)(b * slot + i) * wmax + j) * 3 + k];
                        if (i < hb && j < wb) { const double dd = fabs(got - ref[((size_t)i * wb + j) * 3 + k]); if (dd > eimg) eimg = dd; }
                        else if (fabs(got) > epad) epad = fabs(got);
                    }
            const double e5 = maxrel(gk + 3 * b * n, rk, 3 * n);
            printf("batched canvas sample %d (%dx%d): image max|err| %.3e, padding max %.1e, grad(colors) rel err %.2e\n", b, hb,
                   wb, eimg, epad, e5);
            if (!(eimg <= 2e-4) || epad != 0.0 || !(e5 <= 2e-4)) bad = 1;
            free(gsub);
        }
        free(can);
        CK(hipFree(ws)); CK(hipFree(d_can)); CK(hipFree(d_gcan));
    }
    /* band exchange: select on packed [s,8] records, NaN padding, counts; merge adds the returned rows */
    {
        const int n = 1000, cap = 512, hh = 200, ww = 64;
        float *pk = malloc(sizeof(float) * 8 * n), *d_pk, *d_up, *d_dn, *d_g;
        int *d_iu, *d_id, *d_cnt, cnt[4];
        for (int i = 0; i < n; ++i) {
            pk[8 * i + 0] = 0.02f; pk[8 * i + 1] = 0.02f; pk[8 * i + 2] = 0.f;
            pk[8 * i + 3] = 1.9f * frand(&seed) - 0.95f; pk[8 * i + 4] = 1.9f * frand(&seed) - 0.95f;
            pk[8 * i + 5] = pk[8 * i + 6] = pk[8 * i + 7] = 0.5f;
        }
        CK(hipMalloc((void **)&d_pk, sizeof(float) * 8 * n)); CK(hipMalloc((void **)&d_up, sizeof(float) * 8 * cap));
        CK(hipMalloc((void **)&d_dn, sizeof(float) * 8 * cap)); CK(hipMalloc((void **)&d_g, sizeof(float) * 8 * n));
        CK(hipMalloc((void **)&d_iu, sizeof(int) * cap)); CK(hipMalloc((void **)&d_id, sizeof(int) * cap));
        CK(hipMalloc((void **)&d_cnt, sizeof(int) * 4));
        CK(hipMemcpy(d_pk, pk, sizeof(float) * 8 * n, hipMemcpyHostToDevice));
        gsasr_dims d = {n, hh, ww, 3, 0.1f, 50, 150, 0.f, GSASR_FLAG_STRIDE8};   /* this rank's band: rows [50,150) */
        OK(gsasr_band_select(d_pk, &d, 50, 50, cap, d_up, d_dn, d_iu, d_id, d_cnt, st));
        CK(hipStreamSynchronize(st));
        CK(hipMemcpy(cnt, d_cnt, sizeof(cnt), hipMemcpyDeviceToHost));
        /* expected: the dmax box (0.1 * 199/2 = 9.95 px, far inside the 2.6-sigma support) reaches above row 50 / below 150 */
        int eu = 0, ed = 0;
        for (int i = 0; i < n; ++i) {
            const double cy = ((double)pk[8 * i + 4] + 1.0) * 0.5 * (hh - 1), ey = 0.1 * 0.5 * (hh - 1);
            const double lo = ceil(cy - ey - 0.02), hi = floor(cy + ey + 0.02);
            if (hi < 0 || lo > hh - 1) continue;
            if ((lo < 0 ? 0 : lo) < 50) ++eu;
            if ((hi > hh - 1 ? hh - 1 : hi) >= 150) ++ed;
        }
        printf("band select: up %d (expected %d), down %d (expected %d), far %d\n", cnt[0], eu, cnt[1], ed, cnt[2]);
        if (cnt[0] != eu || cnt[1] != ed || cnt[2] != 0 || eu == 0 || ed == 0 || eu > cap || ed > cap) bad = 1;
        /* merge: g[index[j]] += rows of ones */
        float *ones = malloc(sizeof(float) * 8 * cap), *g = malloc(sizeof(float) * 8 * n);
        int *iu = malloc(sizeof(int) * cap);
        for (int i = 0; i < 8 * cap; ++i) ones[i] = 1.f;
        CK(hipMemcpy(d_up, ones, sizeof(float) * 8 * cap, hipMemcpyHostToDevice));
        CK(hipMemcpy(d_dn, ones, sizeof(float) * 8 * cap, hipMemcpyHostToDevice));
        CK(hipMemsetAsync(d_g, 0, sizeof(float) * 8 * n, st));
        OK(gsasr_band_merge(d_g, n, d_up, d_dn, d_iu, d_id, d_cnt, cap, st));
        CK(hipStreamSynchronize(st));
        CK(hipMemcpy(g, d_g, sizeof(float) * 8 * n, hipMemcpyDeviceToHost));
        double tot = 0;
        for (int i = 0; i < 8 * n; ++i) tot += g[i];
        printf("band merge: %.0f added (expected %d)\n", tot, 8 * (eu + ed));
        if (tot != 8.0 * (eu + ed)) bad = 1;
        free(pk); free(ones); free(g); free(iu);
    }
    /* sampled pixels: values and gradients at a list of points only == gather of the full image / backward of a
       gradient image that is zero except at the points (repeats accumulate) */
    {
        const int np = 777;
        const float dmax = 0.3f;
        int *pts = malloc(sizeof(int) * 2 * np), *d_pts;
        float *go = malloc(sizeof(float) * 3 * np), *out = malloc(sizeof(float) * 3 * np), *d_go, *d_out;
        float *wsp = calloc((size_t)3 * h * w, sizeof(float));
        for (int i = 0; i < np; ++i) {
            pts[2 * i] = (int)(frand(&seed) * h) % h;
            pts[2 * i + 1] = (int)(frand(&seed) * w) % w;
        }
        pts[2] = pts[0]; pts[3] = pts[1];            /* a repeated point */
        pts[4] = -1; pts[5] = -w;                    /* wraps to (h-1, 0) */
        for (int i = 0; i < 3 * np; ++i) go[i] = frand(&seed);
        for (int i = 0; i < np; ++i) {
            const int r = pts[2 * i] < 0 ? pts[2 * i] + h : pts[2 * i], c = pts[2 * i + 1] < 0 ? pts[2 * i + 1] + w : pts[2 * i + 1];
            for (int k = 0; k < 3; ++k) wsp[((size_t)r * w + c) * 3 + k] += go[(size_t)k * np + i];
        }
        gsasr_dims d = {s, h, w, 3, dmax, 0, h, 0.f, GSASR_FLAG_OVERWRITE_GRADS};
        const size_t bytes = gsasr_splat_workspace_bytes(&d), sbytes = gsasr_sample_workspace_bytes(&d, np);
        void *ws, *sws;
        CK(hipMalloc(&ws, bytes)); CK(hipMalloc(&sws, sbytes));
        CK(hipMalloc((void **)&d_pts, sizeof(int) * 2 * np)); CK(hipMalloc((void **)&d_go, sizeof(float) * 3 * np));
        CK(hipMalloc((void **)&d_out, sizeof(float) * 3 * np));
        CK(hipMemcpy(d_pts, pts, sizeof(int) * 2 * np, hipMemcpyHostToDevice));
        CK(hipMemcpy(d_go, go, sizeof(float) * 3 * np, hipMemcpyHostToDevice));
        OK(gsasr_splat_plan(d_sig, d_xy, d_col, &d, ws, bytes, st));
        OK(gsasr_splat_sample_forward(&d, ws, bytes, d_pts, np, d_out, sws, sbytes, st));
        OK(gsasr_splat_sample_backward(d_sig, d_xy, d_col, d_go, d_gs, d_gc, d_gk, &d, ws, bytes, NULL, np, sws, sbytes, st));
        CK(hipStreamSynchronize(st));
        CK(hipMemcpy(out, d_out, sizeof(float) * 3 * np, hipMemcpyDeviceToHost));
        CK(hipMemcpy(gs, d_gs, sizeof(float) * 3 * s, hipMemcpyDeviceToHost));
        CK(hipMemcpy(gc, d_gc, sizeof(float) * 2 * s, hipMemcpyDeviceToHost));
        CK(hipMemcpy(gk, d_gk, sizeof(float) * 3 * s, hipMemcpyDeviceToHost));
        gsref_forward_f64(sig, xy, col, ref, s, h, w, dmax, 0, h);
        gsref_backward_f64(sig, xy, col, wsp, rs, rc, rk, s, h, w, dmax, 0, h);
        double eo = 0;
        for (int i = 0; i < np; ++i) {
            const int r = pts[2 * i] < 0 ? pts[2 * i] + h : pts[2 * i], c = pts[2 * i + 1] < 0 ? pts[2 * i + 1] + w : pts[2 * i + 1];
            for (int k = 0; k < 3; ++k) {
                const double dd = fabs((double)out[(size_t)k * np + i] - ref[((size_t)r * w + c) * 3 + k]);
                if (dd > eo) eo = dd;
            }
        }
        const double e1 = maxrel(gs, rs, 3 * s), e2 = maxrel(gc, rc, 2 * s), e3 = maxrel(gk, rk, 3 * s);
        printf("sampled pixels (%d points): value max|err| %.3e, grad rel err %.2e %.2e %.2e\n", np, eo, e1, e2, e3);
        if (!(eo <= 2e-4) || !(e1 <= 2e-4) || !(e2 <= 2e-4) || !(e3 <= 2e-4)) bad = 1;
        gsasr_dims band = d;
        band.row0 = 10;
        if (gsasr_splat_sample_forward(&band, ws, bytes, d_pts, np, d_out, sws, sbytes, st) == 0) { printf("row band accepted by the sampled path\n"); bad = 1; }
        CK(hipFree(ws)); CK(hipFree(sws)); CK(hipFree(d_pts)); CK(hipFree(d_go)); CK(hipFree(d_out));
        free(pts); free(go); free(out); free(wsp);
    }
    /* tile-stationary backward (GSASR_FLAG_BWD_TILE on the plan's dims: the workspace carries the slots), planar upstream
     * gradient (GSASR_FLAG_CHW_GRAD), and a workspace kept across two plans without a memset in the second
     * (GSASR_FLAG_COUNTERS_CLEAN | GSASR_FLAG_PARITY) */
    {
        const float dmax = 0.3f;
        float *chw = malloc(sizeof(float) * 3 * h * w), *d_chw;
        for (int i = 0; i < h * w; ++i)
            for (int k = 0; k < 3; ++k) chw[(size_t)k * h * w + i] = wgt[(size_t)i * 3 + k];
        CK(hipMalloc((void **)&d_chw, sizeof(float) * 3 * h * w));
        CK(hipMemcpy(d_chw, chw, sizeof(float) * 3 * h * w, hipMemcpyHostToDevice));
        gsasr_dims d = {s, h, w, 3, dmax, 0, h, 0.f, GSASR_FLAG_OVERWRITE_GRADS | GSASR_FLAG_BWD_TILE};
        gsasr_dims plain = d;
        plain.flags = GSASR_FLAG_OVERWRITE_GRADS;
        const size_t bytes = gsasr_splat_workspace_bytes(&d);
        if (!(bytes > gsasr_splat_workspace_bytes(&plain))) { printf("tile-backward plan carries no slots\n"); bad = 1; }
        void *ws;
        CK(hipMalloc(&ws, bytes));
        gsref_backward_f64(sig, xy, col, wgt, rs, rc, rk, s, h, w, dmax, 0, h);
        for (int pass = 0; pass < 2; ++pass) {
            gsasr_dims dp = d;
            if (pass == 1) dp.flags |= GSASR_FLAG_COUNTERS_CLEAN | GSASR_FLAG_PARITY | GSASR_FLAG_CHW_GRAD;
            OK(gsasr_splat_plan(d_sig, d_xy, d_col, &dp, ws, bytes, st));
            OK(gsasr_splat_backward(d_sig, d_xy, d_col, pass == 1 ? d_chw : d_wgt, d_gs, d_gc, d_gk, &dp, ws, bytes, st));
            CK(hipStreamSynchronize(st));
            CK(hipMemcpy(gs, d_gs, sizeof(float) * 3 * s, hipMemcpyDeviceToHost));
            CK(hipMemcpy(gc, d_gc, sizeof(float) * 2 * s, hipMemcpyDeviceToHost));
            CK(hipMemcpy(gk, d_gk, sizeof(float) * 3 * s, hipMemcpyDeviceToHost));
            const double e1 = maxrel(gs, rs, 3 * s), e2 = maxrel(gc, rc, 2 * s), e3 = maxrel(gk, rk, 3 * s);
            printf("tile backward (%s): grad rel err %.2e %.2e %.2e\n", pass ? "planar gradient, kept workspace" : "interleaved gradient", e1, e2, e3);
            if (!(e1 <= 2e-4) || !(e2 <= 2e-4) || !(e3 <= 2e-4)) bad = 1;
        }
        CK(hipFree(ws)); CK(hipFree(d_chw));
        free(chw);
    }
    /* the cutoff a plan actually used (data-derived under the adaptive default) and the launchers' kept scratch, from plain C */
    {
        gsasr_dims d = {s, h, w, 3, 0.3f, 0, h, 0.f, 0u};
        const size_t bytes = gsasr_splat_workspace_bytes(&d);
        void *ws = NULL;
        CK(hipMalloc(&ws, bytes));
        OK(gsasr_splat_plan(d_sig, d_xy, d_col, &d, ws, bytes, st));
        float tau = -1.f;
        unsigned k = 0u;
        OK(gsasr_plan_cutoff(&d, ws, bytes, st, &tau, &k));
        const float tau_n = gsasr_resolve_cutoff(0.f, s);
        printf("plan cutoff: tau' = %.3f (K = %u) against the conservative %.3f\n", tau, k, tau_n);
        if (!(tau >= 16.f && tau <= tau_n + 1e-4f && k > 0u)) bad = 1;
        d.cutoff = 32.f;
        OK(gsasr_splat_plan(d_sig, d_xy, d_col, &d, ws, bytes, st));
        OK(gsasr_plan_cutoff(&d, ws, bytes, st, &tau, &k));
        if (!(fabsf(tau - 32.f) < 1e-3f && k == 0u)) { printf("explicit cutoff not kept: %.3f %u\n", tau, k); bad = 1; }
        /* forward kernel choice: a small x? image keeps the 8 x 16 sub-tiles, the flag forces the wide kernel, and both give
         * the image of the plan above */
        {
            gsasr_dims dw = d;
            const int w0 = gsasr_forward_subtile_width(&dw);
            dw.flags |= GSASR_FLAG_FWD_WIDE;
            const int w1 = gsasr_forward_subtile_width(&dw);
            printf("forward sub-tile width: default %d, GSASR_FLAG_FWD_WIDE %d\n", w0, w1);
            if (w0 != 8 || w1 != 16) bad = 1;
            float *im2 = (float *)malloc(sizeof(float) * 3 * h * w);
            for (int pass = 0; pass < 2; ++pass) {
                gsasr_dims df = d;
                df.flags |= GSASR_FLAG_OVERWRITE_IMAGE | (pass ? GSASR_FLAG_FWD_WIDE : GSASR_FLAG_FWD_NARROW);
                OK(gsasr_splat_forward(&df, ws, bytes, d_img, st));
                CK(hipStreamSynchronize(st));
                CK(hipMemcpy(pass ? im2 : img, d_img, sizeof(float) * 3 * h * w, hipMemcpyDeviceToHost));
            }
            double e = 0;
            for (int i = 0; i < 3 * h * w; ++i) { double dd = fabs((double)img[i] - (double)im2[i]); if (dd > e) e = dd; }
            printf("wide vs narrow forward: max |diff| %.3g\n", e);
            if (!(e <= 1e-5)) bad = 1;
            free(im2);
        }
        CK(hipFree(ws));
        /* the launchers above left their scratch behind for this stream: a second round reuses it, a release frees it,
         * and the round after that allocates again -- same image every time */
        for (int round = 0; round < 3; ++round) {
            CK(hipMemsetAsync(d_img, 0, sizeof(float) * 3 * h * w, st));
            OK(gsasr_gs_render_dmax(d_sig, d_xy, d_col, d_img, s, h, w, 3, 0.3f, st));
            CK(hipStreamSynchronize(st));
            CK(hipMemcpy(img, d_img, sizeof(float) * 3 * h * w, hipMemcpyDeviceToHost));
            gsref_forward_f64(sig, xy, col, ref, s, h, w, 0.3f, 0, h);
            double e = 0;
            for (int i = 0; i < 3 * h * w; ++i) { double dd = fabs((double)img[i] - ref[i]); if (dd > e) e = dd; }
            if (!(e <= 1e-4)) { printf("launcher round %d: image max|err| %.3e\n", round, e); bad = 1; }
            if (round == 1) OK(gsasr_release_launcher_scratch());
        }
    }
    /* error behaviour: status + message instead of a crash */
    if (gsasr_gs_render_dmax(d_sig, d_xy, d_col, d_img, s, h, w, 4, 0.1f, st) == 0) { printf("c=4 accepted\n"); bad = 1; }
    printf("%s\n", bad ? "C-ABI CHECK FAILED" : "C-ABI CHECK OK");
    return bad;
}
