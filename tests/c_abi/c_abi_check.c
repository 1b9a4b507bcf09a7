/*
 * tests/c_abi/c_abi_check.c -- the drop-in boundary exercised from plain C: no Python, no torch.
 *
 * Links libgsasr_splat.so (the product) through include/gsasr_splat.h exactly the way a C/C++ maintainer
 * of the reference would after swapping gs.h's launchers, and checks the
 * results against the CPU oracle's double-precision truth (oracle/libgs_ref.so), both for the
 * reference-shaped launchers and for the plan API.  Built and run by tests/test_c_abi.py on the GPU box.
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
    }
    /* error behaviour: status + message instead of a crash */
    if (gsasr_gs_render_dmax(d_sig, d_xy, d_col, d_img, s, h, w, 4, 0.1f, st) == 0) { printf("c=4 accepted\n"); bad = 1; }
    printf("%s\n", bad ? "C-ABI CHECK FAILED" : "C-ABI CHECK OK");
    return bad;
}
