// splat_backward_home.hip -- backward, HOME-TILE kernel (round 6): a workgroup owns the Gaussians BINNED in its tile of plan
// cells, stages the upstream gradient of the tile plus a halo once in LDS and finishes every one of its Gaussians itself
// (one translation unit of libgsasr_splat.so; gsasr_splat.hip has the overview of the whole pipeline)
//
// Why a third backward.  The Gaussian-stationary kernel (k_render_bwd, one wave per Gaussian) pays ~350 instructions of
// per-wave skeleton for ~90 of sweep at GSASR's x4 windows, and at GSASR's real density (16 Gaussians per LR pixel: a
// million waves on a 1024^2 image) the wave launches alone are 120 of its 390 us.  The tile-stationary kernel
// (k_render_bwd_tile) evaluates (Gaussian, 8 x 8-px quadrant) ITEMS one per lane -- no masks, no cross-lane reduction over
// pixels -- but a Gaussian's items are spread over the 2-6 tiles its window meets, so every (tile, Gaussian) pair stores a
// slot and a gather pass adds the slots (48 us at 1024^2 for a million Gaussians, + 20 us of plan for the slots).
// Here the tile's region is the tile + a 16-px halo, which holds the WHOLE window of a x4-sized Gaussian (half-extent
// <= 16 px: all but ~1% at 16 per LR pixel), so all items of a Gaussian are evaluated by one workgroup, added in LDS and
// written once: no slots, no gather, no atomics, a few thousand workgroups instead of a million waves.  A Gaussian whose
// window does not fit the region (or whose items overflow a round's list) is swept by a whole wave with the
// Gaussian-stationary code (bwd_item) -- any input stays correct, at that kernel's cost per such Gaussian.
//
//   tile      TCX x TCY plan cells (16 px each); region = tile + HM_HALO px on every side, as (RW/8) x (RH/8) quadrants.
//             Batched canvas: tiles never straddle two samples (they are counted per slot).
//   stage     the region's gradient goes to LDS once: per quadrant 16 entries (column, half of 4 rows) of 12 floats
//             {r0..r3, g0..g3, b0..b3}; zero outside the image / the sample / the row band.
//   round     the tile's Gaussians (contiguous runs of the cell order) in rounds of 64 per wave: lane = Gaussian -- its
//             window, the 8-px column span of its ellipse {exponent >= -tau'} per quadrant row (k_bin's span arithmetic,
//             relative to the centre), the item count; one exclusive scan over the workgroup gives every Gaussian its place
//             in the round's item list.
//   items     chunks of <= 64 items, cut where a Gaussian's items end, CLAIMED by the waves from one queue (as in
//             k_render_bwd_tile); lane = (Gaussian, quadrant): 64 pixels, gradients from LDS (lanes of one quadrant read
//             the same address), the residual-form sums of bwd_sweep.  The records of the NEXT chunk are in flight while
//             this one is evaluated.
//   reduce    the lanes park their eight scaled sums in LDS; lane 8 r + k adds component k over the items of the chunk's
//             r-th Gaussian and stores it (or adds it atomically without GSASR_FLAG_OVERWRITE_GRADS).  The order of the
//             additions is fixed: results do not depend on scheduling.
//   tails     large class (row chunks over all waves, atomics: bwd_item) and, with OVERWRITE_GRADS, zeros for the dead class.
#include "splat_common.h"
#include "splat_bwd_sweep.h"

using namespace gsasr_detail;

namespace {

constexpr int HM_HALO = 16;       // px of halo around the tile
constexpr int HM_QS = 196;        // floats per staged quadrant: 16 entries x 12, + 4: neighbouring quadrants of a row start 4 banks apart
// ... and a row of QX quadrants is padded so that the next row starts 16 banks further (mod 64): the 16-byte bank slot of quadrant
// (qy, qx) is (qx + 4 qy) mod 16 -- the up to 4 x 4 quadrants of a x4-sized window hit sixteen different slots, so the lanes of a
// ds_read_b128 group (16 lanes = items of two or three neighbouring Gaussians) rarely meet on a bank (measured with the plain
// stride: 43% of the kernel's LDS cycles were conflict cycles)
constexpr int hm_row_stride(int qx) { return (qx * (HM_QS / 4) + (4 - (qx * (HM_QS / 4)) % 16 + 16) % 16) * 4; }
constexpr float HM_EPS = 0.05f;   // px; covers fp32 centre coordinates (k_bin forms them in double with WINDOW_EPS = 0.02)

// One item: the Gaussian {ra, rb, fa, isy} at the 64 pixels of one staged quadrant (gq), whose column / row coordinates are
// pxq[8] / pyq[8].  a[] = the eight gradient sums {x, y | sx, sy, rho | r, g, b} of bwd_sweep's expansion, SCALED by the
// Gaussian's constants (bwd_scale).  Same arithmetic as k_render_bwd_tile's bt_eval; three 16-byte LDS reads per column half.
template <bool TEST>
__device__ __forceinline__ void hm_eval(const float4 ra, const float4 rb, const float4 fa, const float isy, const float dm,
                                        const float *gq, const float *pxq, const float *pyq, float (&a)[8])
{
    constexpr float HALF_LOG2E = 0.72134752044448170368f;
    const float x = ra.x, y = ra.y, cr = rb.y, cg = rb.z, cb = rb.w;
    const float cinv = fa.x, kappa = fa.y, rho = fa.z, isx = fa.w;
    // exponent (log2) = -h u^2 - h c B^2 with u = dx/sx, B = dy/sy - rho u, c = 1/(1-rho^2) (bwd_trip); B is carried
    // pre-scaled by sB = sqrt(h c), so that the exponent is K0(u) - B'^2
    const float sB = __builtin_amdgcn_sqrtf(HALF_LOG2E * cinv), inv_sB = __builtin_amdgcn_rcpf(sB);
    const float isyB = isy * sB, rsB = rho * sB;
    const float4 ya = *reinterpret_cast<const float4 *>(pyq), yb = *reinterpret_cast<const float4 *>(pyq + 4);
    const float pys[8] = {ya.x, ya.y, ya.z, ya.w, yb.x, yb.y, yb.z, yb.w};
    v2f vp[4], rt[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const v2f dy = (v2f){pys[2 * p], pys[2 * p + 1]} - y;
        vp[p] = dy * isyB;
        if (TEST) rt[p] = (v2f){fabsf(dy.x) <= dm ? 0.f : -INFINITY, fabsf(dy.y) <= dm ? 0.f : -INFINITY};
    }
    float s_uM = 0.f, s_uuM = 0.f, s_N1 = 0.f, s_uN1 = 0.f, s_N2 = 0.f;
    v2f Cr = {0.f, 0.f}, Cg = {0.f, 0.f}, Cb = {0.f, 0.f};
    // A ROLLED loop over the columns whose body requests the gradients of column half s + 1 before it sums half s: unrolled, the
    // scheduler either issued each half's three reads a few instructions ahead of their use (a full LDS round trip at the start
    // of every half, sixteen per item) or -- with the reads written ahead -- clustered all 48 of them at the top (190 VGPRs).
    const float *e = gq;
    float4 R = *reinterpret_cast<const float4 *>(e), G = *reinterpret_cast<const float4 *>(e + 4), B = *reinterpret_cast<const float4 *>(e + 8);
    float pxn = pxq[0];
#pragma clang loop unroll(disable)
    for (int c = 0; c < 8; ++c) {
        const float dx = pxn - x;
        pxn = pxq[min(c + 1, 7)];
        const float u = dx * isx, ru = rsB * u;
        float K0 = -HALF_LOG2E * u * u;
        if (TEST) K0 = fabsf(dx) <= dm ? K0 : -INFINITY;   // exponent -inf: v = 0 exactly, every product with it is 0
        v2f M0 = {0.f, 0.f}, N1 = {0.f, 0.f}, N2 = {0.f, 0.f};
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const float4 Rc = R, Gc = G, Bc = B;
            e += (c == 7 && h == 1) ? 0 : 12;      // (the last half re-reads itself: no read past the quadrant)
#ifdef HM_EXP_NOLDS     // what-if build: synthetic gradients instead of the three LDS reads of a column half
            R = make_float4(x + (float)c, y, dx, u); G = make_float4(u, x, y, dx); B = make_float4(y, u, dx, x);
#else
            R = *reinterpret_cast<const float4 *>(e);
            G = *reinterpret_cast<const float4 *>(e + 4);
            B = *reinterpret_cast<const float4 *>(e + 8);
#endif
#pragma unroll
            for (int pp = 0; pp < 2; ++pp) {
                const int p = 2 * h + pp;
                const v2f gr = pp ? (v2f){Rc.z, Rc.w} : (v2f){Rc.x, Rc.y};
                const v2f gn = pp ? (v2f){Gc.z, Gc.w} : (v2f){Gc.x, Gc.y};
                const v2f gb = pp ? (v2f){Bc.z, Bc.w} : (v2f){Bc.x, Bc.y};
                const v2f Bv = vp[p] - ru;
                v2f pw = K0 - Bv * Bv;
                if (TEST) pw += rt[p];
                const v2f v = {__builtin_amdgcn_exp2f(pw.x), __builtin_amdgcn_exp2f(pw.y)};
                const v2f gp = gb * cb + (gn * cg + gr * cr);   // gs.cu:150
                const v2f qq = gp * v, qB = qq * Bv;
                M0 += qq;
                N1 += qB;
                N2 += qB * Bv;
                Cr += v * gr;
                Cg += v * gn;
                Cb += v * gb;
            }
        }
        // the column's sums, as polynomials in its u (expanded after the last column)
        const float m0 = M0.x + M0.y, n1 = N1.x + N1.y, n2 = N2.x + N2.y;
        const float um = u * m0;
        s_uM += um;
        s_uuM = fmaf(u, um, s_uuM);
        s_N1 += n1;
        s_uN1 = fmaf(u, n1, s_uN1);
        s_N2 += n2;
    }
    // undo the scale of B, then  sum qA = kappa sum(u M0) - rho sum N1  etc. (A = kappa u - rho B, v = B + rho u), times
    // the Gaussian's constants (bwd_scale)
    const float N1t = s_N1 * inv_sB, uN1t = s_uN1 * inv_sB, N2t = s_N2 * inv_sB * inv_sB;
    const float fx = cinv * isx, fy = cinv * isy;
    a[0] = (kappa * s_uM - rho * N1t) * fx;
    a[1] = N1t * fy;
    a[2] = (kappa * s_uuM - rho * uN1t) * fx;
    a[3] = (N2t + rho * uN1t) * fy;
    a[4] = (kappa * uN1t - rho * N2t) * (cinv * cinv);
    a[5] = Cr.x + Cr.y;
    a[6] = Cg.x + Cg.y;
    a[7] = Cb.x + Cb.y;
}

// component k of {x, y | sx, sy, rho | r, g, b} of the Gaussian with original index i (bwd_write's addressing)
__device__ __forceinline__ void hm_write(float v, int k, unsigned i, const Params &P, float *__restrict__ g_sigmas,
                                         float *__restrict__ g_coords, float *__restrict__ g_colors)
{
    float *pc = g_coords + (size_t)i * stride2(P), *ps = g_sigmas + (size_t)i * stride3(P) - 2, *pk = g_colors + (size_t)i * stride3(P) - 5;
    float *dst = (k < 2 ? pc : (k < 5 ? ps : pk)) + k;
    if (P.flags & GSASR_FLAG_OVERWRITE_GRADS) *dst = v;
    else atomicAdd(dst, v);
}

template <bool BOUNDED, int TCX, int TCY, int WAVES>
__global__ __launch_bounds__(64 * WAVES) __attribute__((amdgpu_waves_per_eu(WAVES == 8 ? 4 : (TCX * TCY >= 8 ? 1 : 2)))) void k_render_bwd_home(Params P, PlanView V, const float *__restrict__ grad,
                                                                float *__restrict__ g_sigmas, float *__restrict__ g_coords,
                                                                float *__restrict__ g_colors, int tiles_x, int tps, int cps)
{
    constexpr int THREADS = 64 * WAVES;
    constexpr int TW = CELL * TCX, TH = CELL * TCY, RW = TW + 2 * HM_HALO, RH = TH + 2 * HM_HALO;
    constexpr int QX = RW / 8, QY = RH / 8, NQ = QX * QY;
    constexpr int QBITS = NQ > 64 ? 7 : 6, GBITS = 15 - QBITS;
    constexpr unsigned GMASK = (1u << GBITS) - 1u, QMASK = (1u << QBITS) - 1u;
    constexpr int ROUND = 64 * WAVES;             // Gaussians per round
    constexpr int ITEMS = ROUND * 12;             // items a round's list holds (GSASR-shaped Gaussians at x4 have ~10 each)
    static_assert(ROUND <= (1 << GBITS), "item encoding: Gaussian id bits");
    static_assert(NQ <= (1 << QBITS) && QX <= 15 && QY <= 8, "item encoding: quadrant bits / span nibbles");
    constexpr int RS = hm_row_stride(QX);
    __shared__ __attribute__((aligned(16))) float s_g[QY * RS];
    __shared__ __attribute__((aligned(16))) float s_px[RW];
    __shared__ __attribute__((aligned(16))) float s_py[RH];
    __shared__ __attribute__((aligned(16))) float s_red[WAVES][512];   // parked sums of a chunk; bwd_item's reduction scratch
    __shared__ unsigned s_misc[WAVES][128];       // per chunk: run r -> start | len << 8, then its original index; bwd_item's row values
    __shared__ unsigned short s_items[ITEMS];     // item: Gaussian of the round | quadrant << GBITS | last of its Gaussian << 15
    __shared__ unsigned s_gj[ROUND];              // the round's Gaussians: index in cell order | needs the dmax test << 31
    __shared__ unsigned s_tot[WAVES], s_fit[WAVES], s_head;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned tt = xcd_swizzle(blockIdx.x, gridDim.x);
    const int tx = (int)(tt % (unsigned)tiles_x), ty = (int)(tt / (unsigned)tiles_x);
    // (batched canvas: tiles are counted per slot, so that a tile's cells -- and Gaussians -- belong to one sample)
    const int smp = ty / tps, cy0 = smp * cps + (ty - smp * tps) * TCY, cy1 = min(cy0 + TCY, (smp + 1) * cps) - 1;
    const int cx0 = tx * TCX, cx1 = min(cx0 + TCX, P.ncx) - 1;
    const int X0 = cx0 * CELL - HM_HALO, Y0 = cy0 * CELL - HM_HALO;
    const unsigned long long below = (1ull << lane) - 1ull;
    float *red = s_red[wv];
    unsigned *misc = s_misc[wv];

    // the tile's Gaussians: one contiguous run of the cell order per row of cells
    const unsigned *__restrict__ cs = V.cell_start;
    unsigned rbeg = 0u, rlen = 0u;
    if (lane <= cy1 - cy0) {
        rbeg = cs[(cy0 + lane) * P.ncx + cx0];
        rlen = cs[(cy0 + lane) * P.ncx + cx1 + 1] - rbeg;
    }
    unsigned rb_[TCY], rl_[TCY];
    unsigned n = 0u;
#pragma unroll
    for (int k = 0; k < TCY; ++k) {
        rb_[k] = (unsigned)__builtin_amdgcn_readlane((int)rbeg, k);
        rl_[k] = (unsigned)__builtin_amdgcn_readlane((int)rlen, k);
        n += rl_[k];
    }

    if (n != 0u) {
        const Geo g = sample_geo(P, V, smp);
        // ---- stage the region -------------------------------------------------------------------------------
        {
            const int ylo = max(P.row0, g.base), yhi = min(P.row1, g.base + g.h);
            for (int i = tid; i < RW * RH; i += THREADS) {
                const int row = i / RW, col = i - row * RW, X = X0 + col, Y = Y0 + row;
                float r = 0.f, gg = 0.f, b = 0.f;
                if (X >= 0 && X < g.w && Y >= ylo && Y < yhi) {
                    const float *q = grad + ((size_t)(Y - P.row0) * P.w + X) * 3;
                    r = q[0]; gg = q[1]; b = q[2];
                }
                float *e = s_g + (row >> 3) * RS + (col >> 3) * HM_QS + ((col & 7) * 2 + ((row & 7) >> 2)) * 12 + (row & 3);
                e[0] = r; e[4] = gg; e[8] = b;
            }
            if (tid < RW) s_px[tid] = V.px[g.pxo + min(max(X0 + tid, 0), P.w - 1)];
            for (int i = tid; i < RH; i += THREADS) s_py[i] = V.py[min(max(Y0 + i, 0), P.h - 1)];
            if (tid == 0) s_head = 0u;
        }
        // sqrt(2 tau) of the ellipse the items cover: the windows' tau' (0: no cutoff), or the backward's own smaller one
        // (GSASR_SPLAT_GRAD_TAU: a gradient sums over its own pixels only)
        float kw = __uint_as_float(V.hdr[3]);
        if (P.kb_max > 0.f && kw > P.kb_max) kw = P.kb_max;
        const float hx = 0.5f * (float)(g.w - 1), hy = 0.5f * (float)(g.h - 1);
        __syncthreads();

        for (unsigned base = 0u; base < n; base += (unsigned)ROUND) {
            // ---- phase A: lane = Gaussian: window, quadrant spans, item count -----------------------------------
            const unsigned t = base + (unsigned)(wv * 64 + lane);
            const bool valid = t < n;
            unsigned j = 0u;
            {
                unsigned tt_ = valid ? t : 0u;
#pragma unroll
                for (int k = 0; k < TCY; ++k) {
                    if (tt_ < rl_[k] || k == TCY - 1) { j = rb_[k] + tt_; break; }
                    tt_ -= rl_[k];
                }
            }
            unsigned n_i = 0u, lo4 = 0u, hi4 = 0u;
            bool sweep = false;       // swept by a whole wave instead (window outside the region, too many items)
            bool test = false;
            unsigned orig = 0u;
            if (valid) {
                const uint2 w = V.win[j];
                const float4 ra = V.rec[2 * (size_t)j], fa = V.fin[2 * (size_t)j], fb = V.fin[2 * (size_t)j + 1];
                orig = __float_as_uint(fb.w);
                const int c0 = (int)(w.x & 0x7fffu), c1 = (int)(w.x >> 16), r0 = (int)(w.y & 0x7fffu), r1 = (int)(w.y >> 16);
                test = BOUNDED && (w.x & 0x8000u) != 0u;
                if (c0 <= c1 && r0 <= r1) {
                    if (c0 < X0 || c1 >= X0 + RW || r0 < Y0 || r1 >= Y0 + RH) {
                        sweep = true;
                    } else {
                        // k_bin's span arithmetic (splat_plan.hip), per quadrant row of the REGION: the range of 8-px columns
                        // the ellipse {qa u^2 + qb u v + qc v^2 <= tau'} reaches on the rows the window has in that quadrant row
                        const float rho = fa.z, omr = fa.y;
                        const float spx = hx * __builtin_amdgcn_rcpf(fa.w), spy = hy * __builtin_amdgcn_rcpf(fb.x);     // sigmas in pixels
                        const float cxp = (ra.x + 1.f) * hx, cyp = (ra.y + 1.f) * hy + (float)g.base;
                        const float tau = 0.5f * kw * kw;
                        const float iq = 1.f / (omr * spx * spy);
                        const float qa = 0.5f * iq * (spy / spx), qb = -rho * iq, qc = 0.5f * iq * (spx / spy);
                        const float umax = fabsf(spx) * kw, vmax = fabsf(spy) * kw;
                        const float vstar = -qb * umax / (2.f * qc);
                        const float disc0 = 4.f * qa * tau, disc2 = iq * iq * omr, i2qa = 0.5f / qa;      // (4 qa qc - qb^2 = iq^2 (1 - rho^2): no cancellation)
                        const bool spans = kw > 0.f && !(umax != umax) && !(vmax != vmax);
#pragma unroll
                        for (int qy = 0; qy < QY; ++qy) {
                            const int ya = max(Y0 + 8 * qy, r0), yb = min(Y0 + 8 * qy + 7, r1);
                            int xl = c0, xh = c1;
                            if (ya > yb) { xl = 1; xh = 0; }
                            else if (spans) {
                                const float v0 = ((float)ya - cyp) - HM_EPS, v1 = ((float)yb - cyp) + HM_EPS;
                                if (v1 >= -vmax && v0 <= vmax) {
                                    const float a0 = fmaxf(v0, -vmax), a1 = fminf(v1, vmax);
                                    const float vr = fminf(fmaxf(vstar, a0), a1), vl = fminf(fmaxf(-vstar, a0), a1);
                                    const float dr_ = disc0 - disc2 * vr * vr, dl_ = disc0 - disc2 * vl * vl;
                                    const float uhi = (-qb * vr + sqrtf(fmaxf(dr_, 0.f))) * i2qa;
                                    const float ulo = (-qb * vl - sqrtf(fmaxf(dl_, 0.f))) * i2qa;
                                    xl = max(c0, (int)fmaxf(ceilf(cxp + (ulo - HM_EPS)), -1.f));
                                    xh = min(c1, (int)fminf(floorf(cxp + (uhi + HM_EPS)), 40000.f));
                                } else { xl = 1; xh = 0; }
                            }
                            if (xl <= xh) {
                                const unsigned lo = (unsigned)((xl - X0) >> 3), hi = (unsigned)((xh - X0) >> 3);
                                lo4 |= lo << (4 * qy);
                                hi4 |= hi << (4 * qy);
                                n_i += hi - lo + 1u;
                            } else {
                                lo4 |= 1u << (4 * qy);      // empty: lo = 1 > hi = 0
                            }
                        }
                        if (n_i > 64u) { sweep = true; n_i = 0u; }
                    }
                }
            }
            // place in the round's item list: exclusive scan over the lanes, then over the waves
            unsigned inc = n_i;
            for (int o = 1; o < 64; o <<= 1) {
                const unsigned v = (unsigned)__shfl_up((int)inc, o);
                if (lane >= o) inc += v;
            }
            if (lane == 63) s_tot[wv] = inc;
            __syncthreads();
            unsigned wbase = 0u;
#pragma unroll
            for (int k = 0; k < WAVES; ++k) wbase += k < wv ? s_tot[k] : 0u;
            const bool fit = n_i != 0u && wbase + inc <= (unsigned)ITEMS;
            if (n_i != 0u && !fit) sweep = true;         // the list is full: a prefix of the round's Gaussians fits, the rest is swept
            {
                const unsigned fe = wave_max_u32(fit ? wbase + inc : 0u);
                if (lane == 0) s_fit[wv] = fe;
            }
            const unsigned gid = (unsigned)(wv * 64 + lane);
            if (valid) s_gj[gid] = j | (test ? 0x80000000u : 0u);
            if (fit) {
                unsigned off = wbase + inc - n_i;
                const unsigned last = off + n_i - 1u;
#pragma unroll
                for (int qy = 0; qy < QY; ++qy) {
                    const unsigned lo = (lo4 >> (4 * qy)) & 15u, hi = (hi4 >> (4 * qy)) & 15u;
                    for (unsigned qx = lo; qx <= hi; ++qx, ++off)
                        s_items[off] = (unsigned short)(gid | ((unsigned)(qy * QX) + qx) << GBITS | (off == last ? 0x8000u : 0u));
                }
            }
            // a Gaussian of the tile that reaches no pixel: its gradient is zero (stored gradients must be written)
            if (valid && n_i == 0u && !sweep && (P.flags & GSASR_FLAG_OVERWRITE_GRADS)) {
#pragma unroll
                for (int k = 0; k < 8; ++k) hm_write(0.f, k, orig, P, g_sigmas, g_coords, g_colors);
            }
            __syncthreads();
            unsigned nitems = 0u;
#pragma unroll
            for (int k = 0; k < WAVES; ++k) nitems = max(nitems, s_fit[k]);

            // ---- phase B: chunks of <= 64 items claimed from the queue; the next chunk's records in flight ----------
            auto claim = [&](unsigned &it, int &tlast) -> bool {
                for (;;) {
                    const unsigned p0 = (unsigned)__builtin_amdgcn_readfirstlane((int)*(volatile unsigned *)&s_head);
                    if (p0 >= nitems) return false;
                    const unsigned idx = p0 + (unsigned)lane;
                    it = idx < nitems ? (unsigned)s_items[idx] : 0u;
                    const unsigned long long tails = __ballot(idx < nitems && (it & 0x8000u) != 0u);
                    tlast = 63 - __builtin_clzll(tails);      // (a Gaussian has at most 64 items and the list ends on a marked one: tails != 0)
                    unsigned got = 0u;
                    if (lane == 0) got = atomicCAS(&s_head, p0, p0 + (unsigned)tlast + 1u);
                    if ((unsigned)__builtin_amdgcn_readfirstlane((int)got) == p0) return true;
                }
            };
            auto fetch = [&](unsigned it, int tlast, unsigned &e, float4 &ra, float4 &rb, float4 &fa, float4 &fb) {
                e = 0u;
                if (lane <= tlast) {
                    e = s_gj[it & GMASK];
                    const size_t jj = (size_t)(e & 0x7fffffffu);
                    ra = V.rec[2 * jj]; rb = V.rec[2 * jj + 1];
                    fa = V.fin[2 * jj]; fb = V.fin[2 * jj + 1];
                }
            };
            unsigned itA = 0u, eA = 0u;
            int tlA = -1;
            float4 raA = make_float4(0.f, 0.f, 0.f, 0.f), rbA = raA, faA = raA, fbA = raA;
            bool have = claim(itA, tlA);
            if (have) fetch(itA, tlA, eA, raA, rbA, faA, fbA);
            while (have) {
                unsigned itB = 0u, eB = 0u;
                int tlB = -1;
                float4 raB = make_float4(0.f, 0.f, 0.f, 0.f), rbB = raB, faB = raB, fbB = raB;
                const bool haveB = claim(itB, tlB);
                if (haveB) fetch(itB, tlB, eB, raB, rbB, faB, fbB);
                // -- evaluate chunk A
                const bool ok = lane <= tlA;
                const unsigned q = (itA >> GBITS) & QMASK;
                const unsigned qy = q / (unsigned)QX, qx = q - qy * (unsigned)QX;
                float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                // (the dmax test costs an instruction per pixel pair: only chunks holding a Gaussian that needs it pay)
#ifdef HM_EXP_NOEVAL     // what-if build: no evaluation at all (staging, rounds, claims, records, reduction and writes remain)
                if (ok) { for (int k = 0; k < 8; ++k) a[k] = raA.x + fbA.x + (float)(k + q); }
                else
#endif
                if (BOUNDED && __ballot(ok && (eA >> 31)) != 0ull) {
                    if (ok) hm_eval<true>(raA, rbA, faA, fbA.x, (eA >> 31) ? P.dmax : INFINITY, s_g + qy * RS + qx * HM_QS, s_px + qx * 8u, s_py + qy * 8u, a);
                } else {
                    if (ok) hm_eval<false>(raA, rbA, faA, fbA.x, INFINITY, s_g + qy * RS + qx * HM_QS, s_px + qx * 8u, s_py + qy * 8u, a);
                }
                // -- add up the items of each Gaussian (adjacent lanes) through LDS and write its gradient
#ifdef HM_EXP_NOREDUCE   // what-if build: no parking, no reduction; every item's first sum is written somewhere
                if (ok) hm_write(a[0] + a[1] + a[2] + a[3] + a[4] + a[5] + a[6] + a[7], (int)(q & 7u), __float_as_uint(fbA.w), P, g_sigmas, g_coords, g_colors);
                have = haveB; itA = itB; tlA = tlB; eA = eB;
                raA = raB; rbA = rbB; faA = faB; fbA = fbB;
                continue;
#endif
                const unsigned key = ok ? (itA & GMASK) : 0xffffu;
                const unsigned prev = (unsigned)__shfl_up((int)key, 1);
                const bool head = ok && (lane == 0 || prev != key);
                const unsigned long long H = __ballot(head);
                const int nruns = __builtin_popcountll(H);
                __builtin_amdgcn_wave_barrier();
                {
                    float4 *rp = reinterpret_cast<float4 *>(red + lane * 8);
                    rp[0] = make_float4(a[0], a[1], a[2], a[3]);
                    rp[1] = make_float4(a[4], a[5], a[6], a[7]);
                }
                if (head) {
                    const int rank = __builtin_popcountll(H & below);
                    const unsigned long long above = H & ~((2ull << lane) - 1ull);
                    const int end = above ? __builtin_ctzll(above) : tlA + 1;
                    misc[rank] = (unsigned)lane | (unsigned)(end - lane) << 8;
                    misc[64 + rank] = __float_as_uint(fbA.w);
                }
                __builtin_amdgcn_wave_barrier();
                for (int r0_ = 0; r0_ < nruns; r0_ += 8) {
                    const int r = r0_ + (lane >> 3), k = lane & 7;
                    if (r < nruns) {
                        const unsigned w = misc[r];
                        const int st = (int)(w & 0xffu), len = (int)(w >> 8);
                        // (four reads in flight per step: a loop of single dependent LDS reads was ~12 round trips per chunk;
                        // reads past the run stay inside the wave's 64 parked rows and are dropped)
                        float sum = 0.f;
                        for (int i = 0; i < len; i += 4) {
                            const float p0 = red[(st + i) * 8 + k], p1 = red[min(st + i + 1, 63) * 8 + k];
                            const float p2 = red[min(st + i + 2, 63) * 8 + k], p3 = red[min(st + i + 3, 63) * 8 + k];
                            sum += p0;
                            sum += i + 1 < len ? p1 : 0.f;
                            sum += i + 2 < len ? p2 : 0.f;
                            sum += i + 3 < len ? p3 : 0.f;
                        }
                        hm_write(sum, k, misc[64 + r], P, g_sigmas, g_coords, g_colors);
                    }
                }
                __builtin_amdgcn_wave_barrier();
                have = haveB; itA = itB; tlA = tlB; eA = eB;
                raA = raB; rbA = rbB; faA = faB; fbA = fbB;
            }
            // ---- the Gaussians this wave could not put on the list: one wave-wide sweep each (k_render_bwd's code) ----
            {
                unsigned long long sm = __ballot(sweep);
                while (sm) {
                    const int gl = __builtin_ctzll(sm);
                    sm &= sm - 1ull;
                    const unsigned js = (unsigned)__builtin_amdgcn_readlane((int)j, gl);
                    BwdRec G;
                    bwd_fetch(V, js, G);
                    bwd_item<BOUNDED, false>(js, G, -1, false, lane, P, V, grad, reinterpret_cast<float *>(misc), red, g_sigmas, g_coords, g_colors);
                }
            }
            __syncthreads();
            if (tid == 0) s_head = 0u;
            // (the next round's first barrier orders this store before any claim)
        }
    }

    // ---- tails: the large class (row chunks over all waves, atomics) and the zeros of the dead class --------------
    const unsigned large_beg = cs[P.ncells], large_end = cs[P.ncells + 1];
    const unsigned nlarge = (large_end - large_beg) * (unsigned)NCH;
    if (nlarge) {
        const unsigned nwaves = gridDim.x * (unsigned)WAVES, gw = tt * (unsigned)WAVES + (unsigned)wv;
        BwdRec G;
        for (unsigned it = gw; it < nlarge; it += nwaves) {
            const unsigned jl = large_beg + it / (unsigned)NCH;
            bwd_fetch(V, jl, G);
            bwd_item<BOUNDED, false>(jl, G, (int)(it % (unsigned)NCH), true, lane, P, V, grad, reinterpret_cast<float *>(misc), red, g_sigmas, g_coords, g_colors);
        }
    }
    if (P.flags & GSASR_FLAG_OVERWRITE_GRADS) {
        for (unsigned jd = large_end + tt * (unsigned)THREADS + (unsigned)tid; jd < (unsigned)P.s; jd += gridDim.x * (unsigned)THREADS) {
            const unsigned i = __float_as_uint(V.fin[2 * (size_t)jd + 1].w);
#pragma unroll
            for (int k = 0; k < 8; ++k) hm_write(0.f, k, i, P, g_sigmas, g_coords, g_colors);
        }
    }
}

}  // namespace

namespace gsasr_detail {

// Launch the home-tile backward over the whole cell grid of the plan (rows of cells outside a row band hold no live Gaussian:
// their workgroups only take part in the tails).  variant: 0 = 32 x 16-px tiles, eight waves (dense plans); 1 = 32 x 32, four
// waves; 2 = 64 x 32, four waves (sparse plans).
int launch_bwd_home(const Params &P, const PlanView &V, const float *grad_img, float *g_sigmas, float *g_coords, float *g_colors,
                    int variant, hipStream_t st)
{
    const int cps = P.batch > 1 ? P.slot / CELL : P.ncy;
#define GSASR_HOME(TCX, TCY, W) do { \
        const int tiles_x = (P.ncx + (TCX) - 1) / (TCX), tps = (cps + (TCY) - 1) / (TCY); \
        const dim3 grid((unsigned)tiles_x * (unsigned)tps * (unsigned)P.batch), block(64 * (W)); \
        if (P.bounded) hipLaunchKernelGGL((k_render_bwd_home<true, TCX, TCY, W>), grid, block, 0, st, P, V, grad_img, g_sigmas, g_coords, g_colors, tiles_x, tps, cps); \
        else hipLaunchKernelGGL((k_render_bwd_home<false, TCX, TCY, W>), grid, block, 0, st, P, V, grad_img, g_sigmas, g_coords, g_colors, tiles_x, tps, cps); } while (0)
    if (variant == 0) GSASR_HOME(2, 1, 8);
    else if (variant == 1) GSASR_HOME(2, 2, 4);
    else if (variant == 3) GSASR_HOME(1, 1, 4);
    else GSASR_HOME(4, 2, 4);
#undef GSASR_HOME
    HIP_TRY(hipGetLastError());
    return GSASR_OK;
}

}  // namespace gsasr_detail
