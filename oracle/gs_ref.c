/*
 * oracle/gs_ref.c -- CPU restatement of GSASR's 2D Gaussian-splatting rasterizer kernels.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under gsasr_amd/ may import, link or execute this file; it is
 * the checker for tests/, __graft_entry__.smoke() and the cpu_baseline leg of bench.py.
 *
 * Parity status: PINNED.  The fp32 paths below are checked (tests/test_oracle.py) against golden
 * vectors produced in the authoring container by importing the reference's own pure-PyTorch
 * restatement of the kernels (utils/gs_cuda/check.py:4-27 and utils/gs_cuda_dmax/check.py:4-31,
 * `torch_version`) and autograd through it; see tests/golden/make_golden.py.
 *
 * What is restated (file:line are into the reference tree, /root/reference):
 *   variant 0  "gs_cuda"       forward  utils/gs_cuda/gs.cu:9-61       backward utils/gs_cuda/gs.cu:82-178
 *   variant 1  "gs_cuda_dmax"  forward  utils/gs_cuda_dmax/gs.cu:7-64  backward utils/gs_cuda_dmax/gs.cu:85-165
 *
 * The fp32 functions keep the reference's operand types and association (which sub-expressions are
 * evaluated in double because a `1.0`/`2.0`/`0.5` literal promotes them, which are float), the float
 * box test on float differences, and `+=` vs `=` on the outputs.  Two things cannot be carried over:
 *   - nvcc contracts a*b+c into FMA at its own discretion; this file is compiled with
 *     -ffp-contract=off.  The difference is ulp-level except in `1 - rho*rho` when |rho| -> 1
 *     (cancellation); `use_fma` != 0 evaluates that one expression with fmaf() as nvcc would.
 *   - the dmax forward accumulates with atomicAdd (order undefined); here every pixel is summed in
 *     Gaussian-index order (one of the orders the reference can produce, and gs_cuda's order).
 *
 * The f64 functions ("truth") evaluate the same formulas in double from the same fp32 inputs, but
 * take every box decision exactly as the reference does: pixel coordinate = (float)(double
 * expression), difference and comparison in float.  They are what the HIP kernels are held to
 * (1e-4 absolute per pixel, north_star), because they carry no accumulation-order noise.
 *
 * Build: see oracle/Makefile (gcc -O2 -fopenmp -ffp-contract=off -shared -fPIC).
 */
#include <math.h>
#include <stddef.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* pixel-centre coordinate on the align-corners grid, double expression rounded to float
 * (gs_cuda/gs.cu:27-28, gs_cuda_dmax/gs.cu:39,46) */
static inline float grid_coord(int i, int n) { return (float)(2.0 * i / (n - 1) - 1.0); }

static inline float one_minus_rho2(float rho, int use_fma)
{
    if (use_fma) return fmaf(-rho, rho, 1.0f);
    return 1 - rho * rho;
}

int gsref_num_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

void gsref_set_num_threads(int n)
{
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

/* ------------------------------------------------------------------------------------------------
 * fp32 forward.  img[h,w,c] += sum_s v(s,p) * colors[s,:]     (accumulate-into, like the kernels)
 *   variant 0: no box test, per-Gaussian quantities recomputed per pixel as gs_cuda/gs.cu:33-52
 *   variant 1: box test |dx|<=dmax, |dy|<=dmax and the precomputed form of gs_cuda_dmax/gs.cu:33-56
 * rows [row0,row1) of the full h x w grid are produced; img points at row `row0` (slab layout).
 * ---------------------------------------------------------------------------------------------- */
void gsref_forward_f32(const float *sigmas, const float *coords, const float *colors, float *img,
                       int s, int h, int w, int c, float dmax, int variant, int use_fma,
                       int row0, int row1)
{
    /* per-Gaussian constants of the dmax kernel (gs_cuda_dmax/gs.cu:33-36) */
    float *k = NULL;
    if (variant == 1) {
        k = (float *)malloc(sizeof(float) * 4 * (size_t)(s > 0 ? s : 1));
        for (int si = 0; si < s; ++si) {
            float sx = sigmas[si * 3 + 0], sy = sigmas[si * 3 + 1], rho = sigmas[si * 3 + 2];
            k[si * 4 + 0] = (float)(-0.5 / one_minus_rho2(rho, use_fma)); /* double / float */
            k[si * 4 + 1] = (float)(1.0 / sx / sx);                        /* double chain   */
            k[si * 4 + 2] = (float)(1.0 / sy / sy);
            k[si * 4 + 3] = 2 * rho / sx / sy;                             /* float chain    */
        }
    }
#pragma omp parallel for schedule(dynamic, 1)
    for (int hi = row0; hi < row1; ++hi) {
        float py = grid_coord(hi, h);
        for (int wi = 0; wi < w; ++wi) {
            float px = grid_coord(wi, w);
            float *out = img + ((size_t)(hi - row0) * w + wi) * c;
            for (int si = 0; si < s; ++si) {
                float x = coords[si * 2 + 0], y = coords[si * 2 + 1];
                float dx = px - x, dy = py - y;
                float v;
                if (variant == 1) {
                    if (dy > dmax || dy < -dmax) continue; /* gs_cuda_dmax/gs.cu:41-43 */
                    if (dx > dmax || dx < -dmax) continue; /* gs_cuda_dmax/gs.cu:48-50 */
                    v = k[si * 4 + 1] * dx * dx;
                    v -= k[si * 4 + 3] * dx * dy;
                    v += k[si * 4 + 2] * dy * dy;
                    v *= k[si * 4 + 0];
                } else {
                    float sx = sigmas[si * 3 + 0], sy = sigmas[si * 3 + 1], rho = sigmas[si * 3 + 2];
                    float q = (float)(1.0 / one_minus_rho2(rho, use_fma)); /* gs_cuda/gs.cu:42 */
                    float ox = (float)(1.0 / sx), oy = (float)(1.0 / sy);  /* gs_cuda/gs.cu:43-44 */
                    v = ox * ox * dx * dx;
                    v -= 2 * rho * dx * dy * ox * oy;
                    v += dy * dy * oy * oy;
                    v = (float)(v * (-q / 2.0));                           /* gs_cuda/gs.cu:51 */
                }
                v = expf(v);
                /* colour stride is hard-coded 3 in both forwards (gs.cu:58 / dmax gs.cu:29-31) */
                for (int ci = 0; ci < c && ci < 3; ++ci) out[ci] += v * colors[si * 3 + ci];
            }
        }
    }
    free(k);
}

/* ------------------------------------------------------------------------------------------------
 * fp32 backward.  One Gaussian per outer iteration, pixels row-major, like both kernels.
 *   variant 0 (gs_cuda/gs.cu:112-176): register sums, outputs overwritten (`=`)
 *   variant 1 (gs_cuda_dmax/gs.cu:113-160): 4-compare box test, per-channel `+=` straight into the
 *             outputs (which the wrapper zero-initialises, gs_cuda_dmax/gswrapper.py:40-42)
 * grads points at row `row0` of the upstream gradient (slab layout); rows [row0,row1) are visited.
 * ---------------------------------------------------------------------------------------------- */
void gsref_backward_f32(const float *sigmas, const float *coords, const float *colors,
                        const float *grads, float *g_sigmas, float *g_coords, float *g_colors,
                        int s, int h, int w, int c, float dmax, int variant, int use_fma,
                        int row0, int row1)
{
#pragma omp parallel for schedule(dynamic, 16)
    for (int si = 0; si < s; ++si) {
        float sx = sigmas[si * 3 + 0], sy = sigmas[si * 3 + 1], rho = sigmas[si * 3 + 2];
        float x = coords[si * 2 + 0], y = coords[si * 2 + 1];
        float w1 = (float)(-0.5 / one_minus_rho2(rho, use_fma));
        float w2 = (float)(1.0 / (sx * sx));
        float w3 = (float)(1.0 / (sx * sy));
        float w4 = (float)(1.0 / (sy * sy));
        float od_sx = (float)(1.0 / sx), od_sy = (float)(1.0 / sy);

        float acc_col[3] = {0.f, 0.f, 0.f}, acc_x = 0.f, acc_y = 0.f;
        float acc_sx = 0.f, acc_sy = 0.f, acc_rho = 0.f;
        float *oc = g_colors + (size_t)si * c, *op = g_coords + (size_t)si * 2,
              *os = g_sigmas + (size_t)si * 3;

        for (int hi = row0; hi < row1; ++hi) {
            for (int wi = 0; wi < w; ++wi) {
                float px = grid_coord(wi, w), py = grid_coord(hi, h);
                float dx = px - x, dy = py - y;
                if (variant == 1 && (dx > dmax || dx < -dmax || dy > dmax || dy < -dmax)) continue;
                float d = w2 * dx * dx - 2 * rho * w3 * dx * dy + w4 * dy * dy;
                float v = expf(w1 * d);
                float v2w1 = v * 2 * w1;
                float t_x = v2w1 * (-w2 * dx + rho * w3 * dy);
                float t_y = v2w1 * (-w4 * dy + rho * w3 * dx);
                float t_sx = v2w1 * od_sx * (w3 * rho * dx * dy - w2 * dx * dx);
                float t_sy = v2w1 * od_sy * (w3 * rho * dx * dy - w4 * dy * dy);
                float t_rho = -v2w1 * (2 * w1 * rho * d + w3 * dx * dy);
                const float *gp = grads + ((size_t)(hi - row0) * w + wi) * c;
                if (variant == 1) {
                    /* channel loop with `+=` into the outputs (gs_cuda_dmax/gs.cu:148-160) */
                    for (int ci = 0; ci < c; ++ci) {
                        float gc = gp[ci];
                        float gpt = gc * colors[si * c + ci];
                        oc[ci] += v * gc;
                        op[0] += gpt * t_x;
                        op[1] += gpt * t_y;
                        os[0] += gpt * t_sx;
                        os[1] += gpt * t_sy;
                        os[2] += gpt * t_rho;
                    }
                } else {
                    /* three hard-coded channels, register sums (gs_cuda/gs.cu:131-166) */
                    float gr = gp[0], gg = gp[1], gb = gp[2];
                    acc_col[0] += v * gr;
                    acc_col[1] += v * gg;
                    acc_col[2] += v * gb;
                    float gpt = gr * colors[si * 3 + 0] + gg * colors[si * 3 + 1] + gb * colors[si * 3 + 2];
                    acc_x += gpt * t_x;
                    acc_y += gpt * t_y;
                    acc_sx += gpt * t_sx;
                    acc_sy += gpt * t_sy;
                    acc_rho += gpt * t_rho;
                }
            }
        }
        if (variant != 1) { /* gs_cuda/gs.cu:169-176 */
            os[0] = acc_sx; os[1] = acc_sy; os[2] = acc_rho;
            op[0] = acc_x;  op[1] = acc_y;
            oc[0] = acc_col[0]; oc[1] = acc_col[1]; oc[2] = acc_col[2];
        }
    }
}

/* ------------------------------------------------------------------------------------------------
 * f64 "truth": same formulas, double arithmetic, reference-exact box decisions.
 * dmax < 0 means unbounded.  Outputs are double and are OVERWRITTEN.
 * For the bounded variant the candidate column/row range of a Gaussian is narrowed first (a
 * conservative index window, +-2 px of slack) and the reference's float test is then applied inside
 * the window, so the cost is O(h*s + pairs) instead of O(h*w*s); the result is unchanged.
 * ---------------------------------------------------------------------------------------------- */
static inline void index_window(float centre, float dmax, int n, int bounded, int *lo, int *hi)
{
    if (!bounded || !(dmax < 4.0f)) { *lo = 0; *hi = n - 1; return; }
    double a = ((double)centre - (double)dmax + 1.0) * 0.5 * (n - 1);
    double b = ((double)centre + (double)dmax + 1.0) * 0.5 * (n - 1);
    double l = floor(a) - 2.0, u = ceil(b) + 2.0;
    if (l < 0) l = 0;
    if (u > n - 1) u = n - 1;
    if (!(l <= u)) { *lo = 0; *hi = -1; return; } /* also catches NaN */
    *lo = (int)l; *hi = (int)u;
}

void gsref_forward_f64(const float *sigmas, const float *coords, const float *colors, double *img,
                       int s, int h, int w, float dmax, int row0, int row1)
{
    const int bounded = dmax >= 0.f;
    float *pxs = (float *)malloc(sizeof(float) * (size_t)w);
    for (int wi = 0; wi < w; ++wi) pxs[wi] = grid_coord(wi, w);
#pragma omp parallel for schedule(dynamic, 1)
    for (int hi = row0; hi < row1; ++hi) {
        float py = grid_coord(hi, h);
        double *row = img + (size_t)(hi - row0) * w * 3;
        memset(row, 0, sizeof(double) * (size_t)w * 3);
        for (int si = 0; si < s; ++si) {
            float xf = coords[si * 2 + 0];
            float dyf = py - coords[si * 2 + 1];
            if (bounded && (dyf > dmax || dyf < -dmax)) continue;
            double sx = sigmas[si * 3 + 0], sy = sigmas[si * 3 + 1], rho = sigmas[si * 3 + 2];
            double w1 = -0.5 / (1 - rho * rho);
            double c0 = colors[si * 3 + 0], c1 = colors[si * 3 + 1], c2 = colors[si * 3 + 2];
            int lo, up;
            index_window(xf, dmax, w, bounded, &lo, &up);
            for (int wi = lo; wi <= up; ++wi) {
                float dxf = pxs[wi] - xf;
                if (bounded && (dxf > dmax || dxf < -dmax)) continue;
                double dx = dxf, dy = dyf;
                double d = dx * dx / (sx * sx) - 2 * rho * dx * dy / (sx * sy) + dy * dy / (sy * sy);
                double v = exp(w1 * d);
                row[wi * 3 + 0] += v * c0;
                row[wi * 3 + 1] += v * c1;
                row[wi * 3 + 2] += v * c2;
            }
        }
    }
    free(pxs);
}

void gsref_backward_f64(const float *sigmas, const float *coords, const float *colors,
                        const float *grads, double *g_sigmas, double *g_coords, double *g_colors,
                        int s, int h, int w, float dmax, int row0, int row1)
{
    const int bounded = dmax >= 0.f;
    float *pxs = (float *)malloc(sizeof(float) * (size_t)w);
    for (int wi = 0; wi < w; ++wi) pxs[wi] = grid_coord(wi, w);
#pragma omp parallel for schedule(dynamic, 16)
    for (int si = 0; si < s; ++si) {
        double sx = sigmas[si * 3 + 0], sy = sigmas[si * 3 + 1], rho = sigmas[si * 3 + 2];
        float xf = coords[si * 2 + 0], yf = coords[si * 2 + 1];
        double cr = colors[si * 3 + 0], cg = colors[si * 3 + 1], cb = colors[si * 3 + 2];
        double w1 = -0.5 / (1 - rho * rho), w2 = 1 / (sx * sx), w3 = 1 / (sx * sy), w4 = 1 / (sy * sy);
        double a[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        int rlo, rup, clo, cup;
        index_window(yf, dmax, h, bounded, &rlo, &rup);
        index_window(xf, dmax, w, bounded, &clo, &cup);
        if (rlo < row0) rlo = row0;
        if (rup > row1 - 1) rup = row1 - 1;
        for (int hi = rlo; hi <= rup; ++hi) {
            float dyf = grid_coord(hi, h) - yf;
            if (bounded && (dyf > dmax || dyf < -dmax)) continue;
            for (int wi = clo; wi <= cup; ++wi) {
                float dxf = pxs[wi] - xf;
                if (bounded && (dxf > dmax || dxf < -dmax)) continue;
                double dx = dxf, dy = dyf;
                double d = w2 * dx * dx - 2 * rho * w3 * dx * dy + w4 * dy * dy;
                double v = exp(w1 * d), v2w1 = 2 * w1 * v;
                const float *gp = grads + ((size_t)(hi - row0) * w + wi) * 3;
                double gpt = gp[0] * cr + gp[1] * cg + gp[2] * cb;
                a[0] += gpt * v2w1 * (1 / sx) * (w3 * rho * dx * dy - w2 * dx * dx);
                a[1] += gpt * v2w1 * (1 / sy) * (w3 * rho * dx * dy - w4 * dy * dy);
                a[2] += gpt * -v2w1 * (2 * w1 * rho * d + w3 * dx * dy);
                a[3] += gpt * v2w1 * (-w2 * dx + rho * w3 * dy);
                a[4] += gpt * v2w1 * (-w4 * dy + rho * w3 * dx);
                a[5] += v * gp[0];
                a[6] += v * gp[1];
                a[7] += v * gp[2];
            }
        }
        g_sigmas[si * 3 + 0] = a[0]; g_sigmas[si * 3 + 1] = a[1]; g_sigmas[si * 3 + 2] = a[2];
        g_coords[si * 2 + 0] = a[3]; g_coords[si * 2 + 1] = a[4];
        g_colors[si * 3 + 0] = a[5]; g_colors[si * 3 + 1] = a[6]; g_colors[si * 3 + 2] = a[7];
    }
    free(pxs);
}
