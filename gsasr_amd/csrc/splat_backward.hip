// splat_backward.hip -- backward kernels (Gaussian-stationary, tile-stationary + gather) and gsasr_splat_backward
// (one translation unit of libgsasr_splat.so; gsasr_splat.hip has the overview of the whole pipeline)
#include "splat_common.h"
#include "splat_bwd_sweep.h"

using namespace gsasr_detail;

namespace {


// (occupancy targets: the unrolled sweep fits 6 waves per SIMD at the price of five spilled dwords, -2.7% at config 4;
// forcing the plain sweep to 8 costs more in spills than it gains)
template <bool BOUNDED, bool UNROLL>
__global__ __launch_bounds__(64 * BWD_WAVES) __attribute__((amdgpu_waves_per_eu(UNROLL ? BWD_UNROLL_OCC : BWD_OCC))) void k_render_bwd(Params P, PlanView V, const float *__restrict__ grad,
                                                    float *__restrict__ g_sigmas, float *__restrict__ g_coords,
                                                    float *__restrict__ g_colors)
{
    const int lane = threadIdx.x & 63;
    // XCD-aware order (block b runs on XCD b%8): each XCD sweeps a contiguous run of the cell-ordered
    // Gaussians, i.e. one band of the image, so the grad_img rows it re-reads stay in ITS 4 MiB L2
    const unsigned nb = gridDim.x, b = blockIdx.x;
    const unsigned q = nb >> 3, r = nb & 7u, xcd = b & 7u;
    const unsigned t = xcd * q + min(xcd, r) + (b >> 3);
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned gw = t * (unsigned)BWD_WAVES + (unsigned)wv;
    const unsigned nwaves = nb * (unsigned)BWD_WAVES;
    __shared__ float s_py[BWD_WAVES][128];  // per wave: v = dy/sy of a 64-row block, then (TEST) the raw dy
    __shared__ __attribute__((aligned(16))) float s_red[BWD_WAVES][512];
    float *spy = s_py[wv], *red = s_red[wv];
    // one Gaussian per wave, dispatched by the hardware (a persistent-workgroup variant with a static
    // partition was measured 13% slower at config 2 and 60% slower at config 3: load imbalance)
    // (two or four Gaussians per wave, one after the other, measured the same: wave launch is not the cost)
    BwdRec G;
    u2v lim;
    bwd_fetch_first(V, V.cell_start + P.ncells, min(gw, (unsigned)P.s - 1u), G, lim);  // speculative: class checked below
    const unsigned large_beg = lim.x, large_end = lim.y;
    if (gw < large_beg)
        bwd_item<BOUNDED, UNROLL>(gw, G, -1, false, lane, P, V, grad, spy, red, g_sigmas, g_coords, g_colors);
    else if (gw < large_end)
        bwd_item<BOUNDED, UNROLL>(gw, G, 0, true, lane, P, V, grad, spy, red, g_sigmas, g_coords, g_colors);
    else if (gw < (unsigned)P.s && (P.flags & GSASR_FLAG_OVERWRITE_GRADS))   // dead class: the gradient is zero
        bwd_write(0.f, lane, P, G.fin[7], g_sigmas, g_coords, g_colors);
    // remaining row chunks of the large class, spread over all waves
    const unsigned extra = (large_end - large_beg) * (unsigned)(NCH - 1);
    for (unsigned it = gw; it < extra; it += nwaves) {
        const unsigned j = large_beg + it / (unsigned)(NCH - 1);
        const int chunk = 1 + (int)(it % (unsigned)(NCH - 1));
        bwd_fetch(V, j, G);
        bwd_item<BOUNDED, UNROLL>(j, G, chunk, true, lane, P, V, grad, spy, red, g_sigmas, g_coords, g_colors);
    }
}


// ---------------------------------------------------------------------------------------------------
// backward, Gaussian-stationary with EIGHT Gaussians per wave (round 5).  k_render_bwd spends a wave per Gaussian: at GSASR's
// x4 a window is ~20 x 20 px, i.e. six trips of 21 VALU instructions under ~80 instructions of per-wave fixed work (record
// fetch, tables, expansion, wave reduction, write) with 20 of the 32 lanes of a row in use.  Here a Gaussian gets the 8 lanes
// of one DPP half-row: lane = one column of an 8-column strip, two rows per trip (packed fp32), strips and row pairs in
// per-lane loops (the wave runs until its tallest Gaussian is done; cell-ordered neighbours have like windows).  The fixed
// work is shared by eight Gaussians, 20 of 24 lanes of a strip are in use, a gradient load instruction serves eight windows
// (7.5 wave-loads per Gaussian instead of 13 on the CU's vector-memory pipe), and the reduction is three DPP steps inside
// the half-row.  The sums are bwd_sweep's (residual form: nothing cancels as |rho| -> 1).  The sample-point backward
// (k_sample_bwd) has had this shape since round 2; the full-image one got it when the per-Gaussian cost, not latency, turned
// out to bound GSASR's real density (1 M waves at 16 Gaussians per LR pixel).
// The "large" class (split into row chunks over all waves, atomics) keeps the wave-per-item code: bwd_item below.
// ---------------------------------------------------------------------------------------------------
constexpr int B8_ROWS = 32;    // rows whose dy values a Gaussian's lanes stage in LDS at a time (4 per lane)

template <bool BOUNDED>
__global__ __launch_bounds__(256) void k_render_bwd8(Params P, PlanView V, const float *__restrict__ grad,
                                                     float *__restrict__ g_sigmas, float *__restrict__ g_coords,
                                                     float *__restrict__ g_colors)
{
    constexpr float HALF_LOG2E = 0.72134752044448170368f;
    __shared__ __attribute__((aligned(16))) float s_dy[4][8][2][B8_ROWS];   // per wave and Gaussian: dy / sy, raw dy of a row block
    __shared__ float s_py[4][128];                                          // (large class: bwd_item's scratch)
    __shared__ __attribute__((aligned(16))) float s_red[4][512];
    const int lane = threadIdx.x & 63, sl = lane & 7, grp = lane >> 3;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned t = xcd_swizzle(blockIdx.x, gridDim.x);
    const unsigned j = (t * 256u + threadIdx.x) >> 3;          // this lane's Gaussian (cell order)
    const unsigned large_beg = V.cell_start[P.ncells], large_end = V.cell_start[P.ncells + 1];
    const bool valid = j < (unsigned)P.s;
    const size_t jj = valid ? j : (size_t)P.s - 1;
    const bool normal = valid && j < large_beg;
    uint2 bb = make_uint2(0x7fffu, 0x7fffu);
    float4 ra = make_float4(0.f, 0.f, 0.f, 0.f), rb4 = ra, fa = ra;
    const float4 fb = V.fin[2 * jj + 1];     // {1/sy, px-table offset, sample, original index}: written for every slot
    if (normal) {
        bb = *reinterpret_cast<const uint2 *>(V.bbox + 2 * jj);
        ra = V.rec[2 * jj];
        rb4 = V.rec[2 * jj + 1];
        fa = V.fin[2 * jj];
    }
    const int c0 = (int)(bb.x & 0x7fffu), c1 = (int)(bb.x >> 16);
    const int r0 = (int)(bb.y & 0x7fffu), r1 = (int)(bb.y >> 16);
    const bool live = normal && c0 <= c1;
    const float x = ra.x, y = ra.y, cr = rb4.y, cg = rb4.z, cb = rb4.w;
    const float cinv = fa.x, kappa = fa.y, rho = fa.z, isx = fa.w, isy = fb.x;
    const float nK1 = -HALF_LOG2E * cinv;
    const bool test = BOUNDED && (bb.x & 0x8000u) != 0u;
    const float dmax = test ? P.dmax : INFINITY;
    // (the exact dmax test costs two instructions per trip: a wave pays them only if one of its Gaussians needs it -- the others
    // then test against +inf)
    const bool anytest = BOUNDED && __ballot(test) != 0ull;
    const float *__restrict__ pxt = V.px + __float_as_uint(fb.y);
    const float *__restrict__ pyt = V.py;
    const unsigned pitchb = (unsigned)P.w * 12u;
    // one buffer resource over the whole slab: per-lane byte offsets (the host takes this kernel for slabs below 4 GiB only);
    // reads past the end return 0
    const unsigned long long slab = (unsigned long long)(unsigned)(P.row1 - P.row0) * pitchb;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(grad), 0, (int)(unsigned)(slab < 0xffffffffull ? slab : 0xffffffffull), 0x00020000);
    float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    float *dyn = s_dy[wv][grp][0], *dyr = s_dy[wv][grp][1];
    const int bw = c1 - c0 + 1;
    for (int rb = r0; live && rb <= r1; rb += B8_ROWS) {
        const int nrow = min(r1 - rb + 1, B8_ROWS);
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int k = 0; k < 4; ++k) {      // the block's row values: dy / sy for the exponent, raw dy for the exact dmax test
            const float d = pyt[min(rb + 4 * sl + k, r1)] - y;
            dyn[4 * sl + k] = d * isy;
            dyr[4 * sl + k] = d;
        }
        __builtin_amdgcn_wave_barrier();
        for (int strip = 0; strip < bw; strip += 8) {
            const int X = c0 + min(strip + sl, bw - 1);
            const float dx = pxt[X] - x;
            const bool inx = strip + sl < bw && fabsf(dx) <= dmax;
            const float u = dx * isx, rho_u = rho * u;
            const float K0 = inx ? -HALF_LOG2E * u * u : -INFINITY;     // a column outside the window (or the box): v = 0 exactly
            BwdRow R;
            R.m1 = R.m2 = R.k01 = (v2f){0.f, 0.f};
            R.ka2 = R.kb0 = R.kb1 = R.kb2 = 0.f;
            int voff = (int)((unsigned)(rb - P.row0) * pitchb + (unsigned)X * 12u);
            int rr = 0;
            for (; rr + 1 < nrow; rr += 2, voff += (int)(2u * pitchb)) {
                const u3v ga = __builtin_amdgcn_raw_buffer_load_b96(rsrc, voff, 0, 0);
                const u3v gb = __builtin_amdgcn_raw_buffer_load_b96(rsrc, voff, (int)pitchb, 0);
                Grad6 g;
                g.a0 = __uint_as_float(ga.x); g.a1 = __uint_as_float(ga.y); g.a2 = __uint_as_float(ga.z);
                g.b0 = __uint_as_float(gb.x); g.b1 = __uint_as_float(gb.y); g.b2 = __uint_as_float(gb.z);
                const v2f n0 = *reinterpret_cast<const v2f *>(dyn + rr);
                const v2f w0 = BOUNDED ? *reinterpret_cast<const v2f *>(dyr + rr) : n0;
                if (BOUNDED && anytest) bwd_trip<true, false>(R, g, n0, w0, true, true, K0, nK1, rho_u, cr, cg, cb, dmax);
                else bwd_trip<false, false>(R, g, n0, w0, true, true, K0, nK1, rho_u, cr, cg, cb, dmax);
            }
            if (rr < nrow) {     // odd last row: the second pixel of the pair is switched off (and reads the same, valid row)
                const u3v ga = __builtin_amdgcn_raw_buffer_load_b96(rsrc, voff, 0, 0);
                Grad6 g;
                g.a0 = g.b0 = __uint_as_float(ga.x); g.a1 = g.b1 = __uint_as_float(ga.y); g.a2 = g.b2 = __uint_as_float(ga.z);
                const v2f n0 = {dyn[rr], dyn[rr]};
                const v2f w0 = {dyr[rr], dyr[rr]};
                if (BOUNDED && anytest) bwd_trip<true, true>(R, g, n0, w0, true, false, K0, nK1, rho_u, cr, cg, cb, dmax);
                else bwd_trip<false, true>(R, g, n0, w0, true, false, K0, nK1, rho_u, cr, cg, cb, dmax);
            }
            // the column's three sums expanded to the five gradient sums (bwd_sweep)
            const float Kr = R.k01.x + R.kb0, Kg = R.k01.y + R.kb1, Kb = R.ka2 + R.kb2;
            const float M0 = fmaf(Kb, cb, fmaf(Kg, cg, Kr * cr)), N1 = R.m1.x + R.m1.y, N2 = R.m2.x + R.m2.y;
            const float ue = inx ? u : 0.f, uk = ue * kappa;
            const float sA = uk * M0 - rho * N1;
            a[0] += sA; a[1] += N1; a[2] += ue * sA; a[3] += N2 + rho * ue * N1; a[4] += uk * N1 - rho * N2;
            a[5] += Kr; a[6] += Kg; a[7] += Kb;
        }
    }
    if (live) bwd_scale(a, cinv, isx, isy);
    // sum over the Gaussian's 8 lanes (half a DPP row): its first lane gets the totals
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        float v = a[k];
        v += dpp_row_shl<4>(v);
        v += dpp_row_shl<2>(v);
        v += dpp_row_shl<1>(v);
        a[k] = v;
    }
    // normal class: its gradient; dead class (behind the large one in cell order): zero when gradients are stored
    const bool dead_slot = valid && j >= large_end;
    if (sl == 0 && (normal || dead_slot)) {
        const unsigned orig = __float_as_uint(fb.w);
        const bool store = (P.flags & GSASR_FLAG_OVERWRITE_GRADS) != 0u;
        float *pc = g_coords + (size_t)orig * stride2(P), *ps = g_sigmas + (size_t)orig * stride3(P),
              *pk = g_colors + (size_t)orig * stride3(P);
        if (store) {
            pc[0] = a[0]; pc[1] = a[1]; ps[0] = a[2]; ps[1] = a[3]; ps[2] = a[4]; pk[0] = a[5]; pk[1] = a[6]; pk[2] = a[7];
        } else if (live) {
            atomicAdd(pc, a[0]); atomicAdd(pc + 1, a[1]); atomicAdd(ps, a[2]); atomicAdd(ps + 1, a[3]); atomicAdd(ps + 2, a[4]);
            atomicAdd(pk, a[5]); atomicAdd(pk + 1, a[6]); atomicAdd(pk + 2, a[7]);
        }
    }
    // the large class: NCH row chunks per Gaussian, dealt to all waves of the launch, one wave per chunk (k_render_bwd's code)
    const unsigned nitems = (large_end - large_beg) * (unsigned)NCH;
    if (nitems) {
        const unsigned nwaves = gridDim.x * 4u, gw = t * 4u + (unsigned)wv;
        BwdRec G;
        for (unsigned it = gw; it < nitems; it += nwaves) {
            const unsigned jl = large_beg + it / (unsigned)NCH;
            bwd_fetch(V, jl, G);
            bwd_item<BOUNDED, false>(jl, G, (int)(it % (unsigned)NCH), true, lane, P, V, grad, s_py[wv], s_red[wv], g_sigmas, g_coords, g_colors);
        }
    }
}


// ---------------------------------------------------------------------------------------------------
// backward, TILE-stationary (BASELINE.json north_star's shape: a workgroup owns an HR tile, stages its grad_img ONCE
// in LDS and streams the Gaussians binned near it).  Measured against the Gaussian-stationary k_render_bwd above in
// DESIGN.md 3c; the host picks between the two (gsasr_splat_backward).
//
//   tile      32 x 16 px = 8 "quadrants" of 8 x 8 px, one workgroup of two waves per tile; from x8 up (bt_tall) 32 x 32 px =
//             16 quadrants and four waves.  XCD-banded tile order.
//   stage     the tile's gradient (HWC or planar CHW, zero outside the image / the sample / the row band) goes to LDS as
//             packed row pairs {r_a, r_b, g_a, g_b, b_a, b_b} per (column, row pair) of each quadrant, with the px / py
//             table entries of the tile.  Every pixel of grad_img is read once per tile that holds it -- exactly once.
//   level 1   as in the forward (fwd_block): the four waves test the windows of the Gaussians binned within reach of
//             the tile, 64 per wave and chunk, and append the survivors to a list in LDS.  A survivor also appends one
//             ITEM per quadrant its window touches (1..8 or 16), survivor-major, the last one marked.
//   level 2   LANE = ITEM = (Gaussian, quadrant): a lane loads its Gaussian's records once and evaluates it at the 64
//             pixels of its quadrant -- gradients read from LDS (lanes of different quadrants hit disjoint banks), two
//             rows per packed-fp32 operation, columns in the outer loop so that u = dx/sx is constant in the inner one
//             and the same residual-form sums as bwd_sweep apply.  No cross-lane reduction of pixels, no masks: a pixel
//             outside the Gaussian's window adds a term below exp(-tau), a pixel outside the image adds 0 * v.
//             The items of one Gaussian sit in adjacent lanes (chunks are cut at the last marked lane, so a Gaussian never
//             straddles two chunks): three (four) shuffle steps add them up, and the first lane of each run stores the eight raw
//             sums into the Gaussian's slot for THIS tile (PlanView::part) -- plain 32-byte stores, no atomics, no
//             dependence on scheduling.  (Measured on this chip: fp32 global atomics retire ~19 G cache-line requests/s
//             chip-wide and ds_add_f32 ~3 cycles per lane; tools/atomic_rate.hip.  One atomic set per (tile, Gaussian)
//             would be 14 us of atomic traffic at config 2.)
//   gather    k_bwd_gather (or the fused k_prologue_bwd_gather of the step entry points): one thread per Gaussian adds
//             the slots of its window's tiles in order, applies the Gaussian's constants and writes the gradient.
// A Gaussian whose window spans more tiles than it has slots (or the "large" class) adds into PlanView::sums with
// atomics instead; the gather adds those as well.
// ---------------------------------------------------------------------------------------------------
// One item: Gaussian j (cell order) at the 64 pixels of one quadrant.  gq = the quadrant's block of staged gradients,
// pxq / pyq = its 8 column / row coordinates.  a[] = raw sums {qA, qB, quA, qvB, qAB, Cr, Cg, Cb} (cf. bwd_sweep).
template <bool TEST>
__device__ __forceinline__ void bt_eval(const PlanView &V, unsigned j, float dm, const float *gq, const float *pxq,
                                        const float *pyq, float (&a)[8])
{
    constexpr float HALF_LOG2E = 0.72134752044448170368f;
    const float4 ra = V.rec[2 * (size_t)j], rb = V.rec[2 * (size_t)j + 1];
    const float4 fa = V.fin[2 * (size_t)j];
    const float isy = V.fin[2 * (size_t)j + 1].x;
    const float x = ra.x, y = ra.y, cr = rb.y, cg = rb.z, cb = rb.w;
    const float cinv = fa.x, kappa = fa.y, rho = fa.z, isx = fa.w;
    // exponent (log2) = -h u^2 - h c B^2 with u = dx/sx, B = dy/sy - rho u, c = 1/(1-rho^2) (bwd_trip); B is carried
    // pre-scaled by sB = sqrt(h c), so that the exponent is K0(u) - B'^2
    const float sB = __builtin_amdgcn_sqrtf(HALF_LOG2E * cinv), inv_sB = __builtin_amdgcn_rcpf(sB);
    const float isyB = isy * sB, rsB = rho * sB;
    v2f vp[4], rt[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const v2f dy = (v2f){pyq[2 * p], pyq[2 * p + 1]} - y;
        vp[p] = dy * isyB;
        if (TEST) rt[p] = (v2f){fabsf(dy.x) <= dm ? 0.f : -INFINITY, fabsf(dy.y) <= dm ? 0.f : -INFINITY};
    }
    float s_uM = 0.f, s_uuM = 0.f, s_N1 = 0.f, s_uN1 = 0.f, s_N2 = 0.f;
    v2f Cr = {0.f, 0.f}, Cg = {0.f, 0.f}, Cb = {0.f, 0.f};
    for (int c = 0; c < 8; ++c) {
        const float dx = pxq[c] - x;
        const float u = dx * isx, ru = rsB * u;
        float K0 = -HALF_LOG2E * u * u;
        if (TEST) K0 = fabsf(dx) <= dm ? K0 : -INFINITY;   // exponent -inf: v = 0 exactly, every product with it is 0
        v2f M0 = {0.f, 0.f}, N1 = {0.f, 0.f}, N2 = {0.f, 0.f};
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const float4 d0 = *reinterpret_cast<const float4 *>(gq + (c * 4 + p) * 8);
            const float2 d1 = *reinterpret_cast<const float2 *>(gq + (c * 4 + p) * 8 + 4);
            const v2f Bv = vp[p] - ru;
            v2f pw = K0 - Bv * Bv;
            if (TEST) pw += rt[p];
            const v2f v = {__builtin_amdgcn_exp2f(pw.x), __builtin_amdgcn_exp2f(pw.y)};
            const v2f gr = {d0.x, d0.y}, gn = {d0.z, d0.w}, gb = {d1.x, d1.y};
            const v2f gp = gb * cb + (gn * cg + gr * cr);   // gs.cu:150
            const v2f qq = gp * v, qB = qq * Bv;
            M0 += qq;
            N1 += qB;
            N2 += qB * Bv;
            Cr += v * gr;
            Cg += v * gn;
            Cb += v * gb;
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
    // undo the scale of B, then  sum qA = kappa sum(u M0) - rho sum N1  etc.: bwd_sweep's per-column expansion summed
    // over the columns (A = kappa u - rho B, v = B + rho u)
    const float N1t = s_N1 * inv_sB, uN1t = s_uN1 * inv_sB, N2t = s_N2 * inv_sB * inv_sB;
    a[0] = kappa * s_uM - rho * N1t;
    a[1] = N1t;
    a[2] = kappa * s_uuM - rho * uN1t;
    a[3] = N2t + rho * uN1t;
    a[4] = kappa * uN1t - rho * N2t;
    a[5] = Cr.x + Cr.y;
    a[6] = Cg.x + Cg.y;
    a[7] = Cb.x + Cb.y;
}

// (the 32-row tile runs twice the waves per workgroup with half the chunks each: the same rounds, the same waves per CU under
// its 27 KB of LDS)
// LISTS (round 5): the tile's survivors and their quadrants come from the plan's tile list (k_bin's tl_emit: {slot in cell order |
// test << 31, quadrant mask | slot in part[] << 16}) instead of level 1's walk over the cells around the tile; a tile whose list
// overflowed its capacity walks as before.
// WV = waves per tile: two per 16 rows (BT_WAVES); FOUR on 32 x 16-px tiles of dense plans read from lists (round 5) -- at 16
// Gaussians per LR pixel a tile's list holds ~2 000 entries = 75 item chunks, and 2 048 tiles x 2 waves leave the chip one
// generation of four waves per SIMD with nothing to balance: config-5 canvas 320 -> 266 us, 1024^2 at 16 per LR pixel 459 -> 424
// (profiles/history/r05_bt_variants.txt; the same four waves on x8's 32 x 32-px tiles lose: 1 210 -> 1 648 us at config 4).
template <bool BOUNDED, int BT_CHUNKS, int HLOG, bool LISTS, int WV = (BT_WAVES << (HLOG - 4))>
__global__ __launch_bounds__(64 * WV) __attribute__((amdgpu_waves_per_eu((BT_CHUNKS * WV) <= 4 ? 5 : 4, 5))) void k_render_bwd_tile(
    Params P, PlanView V, const float *__restrict__ grad, int tiles_x, int use_atomics)
{
    constexpr int BT_H = 1 << HLOG, NQY = BT_H / 8, NQ = 4 * NQY;   // tile height, quadrant rows, quadrants (8 or 16)
    constexpr int WAVES = WV, THREADS = 64 * WAVES;
    constexpr int BT_LIST = WAVES * BT_CHUNKS * 64;     // survivors per round at most (512 / 256)
    __shared__ __attribute__((aligned(16))) float s_g[NQ * BT_QSTRIDE];
    __shared__ float s_px[BT_W], s_py[BT_H];
    __shared__ unsigned s_list[BT_LIST];            // survivor: index in cell order | needs the dmax test << 31
    __shared__ unsigned char s_slot[BT_LIST];       // its slot in part[] for this tile, or BT_WIDE
    __shared__ unsigned short s_items[BT_LIST * NQ]; // item: survivor (9 bits) | quadrant << 9 | last of its survivor << 13
    __shared__ unsigned s_cnt[3];                   // survivors, items of the round; head of the item queue (level 2)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned tt = xcd_swizzle(blockIdx.x, gridDim.x);
    const int tx = (int)(tt % (unsigned)tiles_x), ty = (int)(tt / (unsigned)tiles_x);
    const int bx0 = tx * BT_W, by0 = P.row0 + ty * BT_H;
    const int bx1 = min(bx0 + BT_W - 1, P.w - 1), by1 = min(by0 + BT_H - 1, P.row1 - 1);
    const int smp = P.batch > 1 ? by0 / P.slot : 0;
    const Geo g = sample_geo(P, V, smp);

    // ---- stage the tile ---------------------------------------------------------------------------------
    {
        const int ylim = min(P.row1, g.base + g.h);
        const bool chw = (P.flags & GSASR_FLAG_CHW_GRAD) != 0u;
        size_t plane = (size_t)(P.row1 - P.row0) * P.w, org = 0;     // planar: [3, rows, w]; batched [B, 3, grad_rows, w]
        int yoff = P.row0;
        if (chw && P.batch > 1) {
            plane = (size_t)P.grad_rows * P.w;
            org = (size_t)smp * 3 * plane;
            yoff = g.base;
        }
#pragma unroll
        for (int i = tid; i < BT_W * BT_H; i += THREADS) {
            const int row = i >> 5, col = i & 31, X = bx0 + col, Y = by0 + row;
            float r = 0.f, gg = 0.f, b = 0.f;
            if (X < g.w && Y < ylim) {
                if (chw) {
                    const float *q = grad + org + (size_t)(Y - yoff) * P.w + X;
                    r = q[0]; gg = q[plane]; b = q[2 * plane];
                } else {
                    const float *q = grad + ((size_t)(Y - P.row0) * P.w + X) * 3;
                    r = q[0]; gg = q[1]; b = q[2];
                }
            }
            float *e = s_g + ((row >> 3) * 4 + (col >> 3)) * BT_QSTRIDE + (((col & 7) * 4 + ((row & 7) >> 1)) * 8) + (row & 1);
            e[0] = r; e[2] = gg; e[4] = b;
        }
        if (tid < BT_W) s_px[tid] = V.px[g.pxo + min(bx0 + tid, P.w - 1)];
        else if (tid < BT_W + BT_H) s_py[tid - BT_W] = V.py[min(by0 + tid - BT_W, P.h - 1)];
        if (tid < 3) s_cnt[tid] = 0u;
    }

    // ---- segment table of the tile (every wave builds the same one; cf. fwd_block) ------------------------
    const unsigned *__restrict__ cs = V.cell_start;
    const int rx = (int)V.hdr[8], ry = (int)V.hdr[9];
    int nseg = 0;
    unsigned sbeg = 0, send = 0;
    if (rx > 0) {
        const int cx0 = max(bx0 - rx, 0) >> CELL_SHIFT, cx1 = min((bx1 + rx) >> CELL_SHIFT, P.ncx - 1);
        const int cy0 = max(by0 - ry, 0) >> CELL_SHIFT, cy1 = min((by1 + ry) >> CELL_SHIFT, P.ncy - 1);
        nseg = cy1 - cy0 + 1;
        if (lane < nseg) {
            sbeg = cs[(cy0 + lane) * P.ncx + cx0];
            send = cs[(cy0 + lane) * P.ncx + cx1 + 1];
        }
    }
    if (lane == nseg) {
        sbeg = cs[P.ncells];
        send = cs[P.ncells + 1];
    }
    ++nseg;
    const unsigned len = send - sbeg;
    unsigned pin = len;
    for (int o = 1; o < 64; o <<= 1) {
        const unsigned v = (unsigned)__shfl_up((int)pin, o);
        if (lane >= o) pin += v;
    }
    const unsigned pex = pin - len;
    const unsigned total = (unsigned)__builtin_amdgcn_readlane((int)pin, nseg - 1);
    // the tile's list, if the plan wrote one that holds everything; else (and for the large class, which no list holds) the walk
    const unsigned lcnt = LISTS ? (unsigned)__builtin_amdgcn_readfirstlane((int)V.tl_cursor[(size_t)tt * TL_STRIDE]) : 0u;
    const bool uselist = LISTS && lcnt <= (unsigned)P.tl_cap;
    const uint2 *__restrict__ ent = V.tl_entries + (size_t)tt * (size_t)P.tl_cap;
    const unsigned nlarge_c = uselist ? (((unsigned)__builtin_amdgcn_readlane((int)len, nseg - 1) + 63u) >> 6) : 0u;   // chunks of the large segment
    const unsigned lchunks = (lcnt + 63u) >> 6;
    const unsigned nchunks = uselist ? lchunks + nlarge_c : (total + 63u) >> 6;
    const unsigned large_q0 = (unsigned)__builtin_amdgcn_readlane((int)pex, nseg - 1);     // where the large segment starts in the flat walk
    const unsigned long long below = (1ull << lane) - 1ull;
    int rseg = 0;
    __syncthreads();

    for (unsigned base = 0; base < nchunks; base += (unsigned)(WAVES * BT_CHUNKS)) {
        // ---- level 1: candidates -> survivors + items ----------------------------------------------------
        unsigned cj[BT_CHUNKS];
        uint2 cw[BT_CHUNKS];
#pragma unroll
        for (int k = 0; k < BT_CHUNKS; ++k) {
            const unsigned c = base + (unsigned)wv + (unsigned)(WAVES * k);
            cw[k] = make_uint2(0x7fffu, 0x7fffu);
            if (uselist && c < lchunks) {          // a chunk of list entries: cw = the entry itself
                const unsigned e = c * 64u + (unsigned)lane;
                cj[k] = 0xffffffffu;
                if (e < lcnt) {
                    cw[k] = ent[e];
                    cj[k] = cw[k].x & 0x7fffffffu;
                }
                continue;
            }
            if (uselist) {                          // behind the list: the large class alone, walked as ever
                const unsigned q = large_q0 + (c - lchunks) * 64u + (unsigned)lane;
                cj[k] = (c < nchunks && q < total) ? (unsigned)__builtin_amdgcn_readlane((int)sbeg, nseg - 1) + (q - large_q0) : 0xffffffffu;
            } else {
                cj[k] = c < nchunks ? fwd_candidate(c, lane, nseg, rseg, sbeg, pex, pin) : 0xffffffffu;
            }
            if (cj[k] != 0xffffffffu) cw[k] = V.win[cj[k]];
        }
#pragma unroll
        for (int k = 0; k < BT_CHUNKS; ++k) {
            const bool from_list = uselist && base + (unsigned)wv + (unsigned)(WAVES * k) < lchunks;     // wave-uniform
            const int c0 = (int)(cw[k].x & 0x7fffu), c1 = (int)(cw[k].x >> 16);
            const int r0 = (int)(cw[k].y & 0x7fffu), r1 = (int)(cw[k].y >> 16);
            const bool hit = from_list ? cj[k] != 0xffffffffu : (c0 <= bx1) & (c1 >= bx0) & (r0 <= by1) & (r1 >= by0);
            const unsigned long long m = __ballot(hit);
            if (m == 0ull) continue;
            unsigned at = 0;
            if (lane == 0) at = atomicAdd(&s_cnt[0], (unsigned)__builtin_popcountll(m));
            at = (unsigned)__builtin_amdgcn_readfirstlane((int)at);
            const unsigned pos = at + (unsigned)__builtin_popcountll(m & below);
            // quadrants of the tile the window touches, row by row trimmed to the columns the ellipse reaches (k_bin's
            // per-8-row spans): the corners of the window are empty for every Gaussian, most of it for a correlated one
            const int qx0 = max(c0 - bx0, 0) >> 3, qx1 = min(c1 - bx0, BT_W - 1) >> 3;
            const int qy0 = max(r0 - by0, 0) >> 3, qy1 = min(r1 - by0, BT_H - 1) >> 3;
            int xl[NQY], xh[NQY];
#pragma unroll
            for (int qy = 0; qy < NQY; ++qy) { xl[qy] = 1; xh[qy] = 0; }
            if (hit && !from_list) {
                // per-8-row spans (qspan) when the window has at most eight such bands; a taller window (x12 and up) still has
                // the forward's per-16-row spans in its window words: both quadrant rows of this tile then share one band
                uint4 qs = make_uint4(0u, 0xffffffffu, 0u, 0xffffffffu);
                const int q0 = (r0 - P.row0) >> 3, b8 = (by0 - P.row0) >> 3;
                const bool fine = ((r1 - P.row0) >> 3) - q0 < 8;
                if (fine) {
                    if (V.qspan) qs = V.qspan[cj[k]];
                } else if (cw[k].y & 0x8000u) {
                    const uint2 *sp = reinterpret_cast<const uint2 *>(V.bbox + 2 * (size_t)cj[k]);
                    const uint2 s0 = sp[1], s1 = sp[2];
                    qs = make_uint4(s0.x, s0.y, s1.x, s1.y);
                }
                const int cu = (c0 >> 3) - (bx0 >> 3);
#pragma unroll
                for (int qy = 0; qy < NQY; ++qy) {
                    const unsigned t = (unsigned)(fine ? b8 - q0 + qy : ((b8 + qy) >> 1) - (q0 >> 1)) & 7u, sh = (t & 3u) * 8u;
                    const int lo = (int)(((t < 4u ? qs.x : qs.z) >> sh) & 0xffu), hi = (int)(((t < 4u ? qs.y : qs.w) >> sh) & 0xffu);
                    if (qy >= qy0 && qy <= qy1) {
                        // (hi = 255 is "as far as the window goes": the default of a window k_bin computed no spans for --
                        // one wider than 255 columns of 8 px among them, whose far tiles would otherwise lose their quadrants)
                        xl[qy] = max(qx0, cu + lo);
                        xh[qy] = hi == 255 ? qx1 : min(qx1, cu + hi);
                    }
                }
            }
            // the quadrants as a mask (bit 4 qy + qx): from the spans just cut, or as the plan's list holds it
            unsigned qmask = 0u;
#pragma unroll
            for (int qy = 0; qy < NQY; ++qy)
                if (xh[qy] >= xl[qy]) qmask |= ((2u << xh[qy]) - (1u << xl[qy])) << (4 * qy);
            if (from_list) qmask = hit ? (cw[k].y & 0xffffu) : 0u;
            const unsigned n_i = (unsigned)__builtin_popcount(qmask);
            unsigned inc = n_i;
            for (int o = 1; o < 64; o <<= 1) {
                const unsigned v = (unsigned)__shfl_up((int)inc, o);
                if (lane >= o) inc += v;
            }
            unsigned ib = 0;
            if (lane == 63) ib = atomicAdd(&s_cnt[1], inc);
            ib = (unsigned)__builtin_amdgcn_readlane((int)ib, 63);
            if (hit) {
                unsigned slot;
                if (from_list) {       // (entry = {j | test << 31, mask | slot << 16})
                    s_list[pos] = cw[k].x;
                    slot = (cw[k].y >> 16) & 0xffu;
                } else {
                    s_list[pos] = cj[k] | ((cw[k].x & 0x8000u) << 16);
                    int ntx, wtx0, wty0;
                    const int nt = bt_tile_span(cw[k].x, cw[k].y, P.row0, HLOG, ntx, wtx0, wty0);
                    slot = nt <= P.part_k ? (unsigned)((ty - wty0) * ntx + (tx - wtx0)) : BT_WIDE;
                }
                s_slot[pos] = (unsigned char)slot;
                unsigned off = ib + inc - n_i;
                const unsigned last = off + n_i - 1u;
                for (unsigned mm = qmask; mm; mm &= mm - 1u, ++off)
                    s_items[off] = (unsigned short)((unsigned)pos | (unsigned)__builtin_ctz(mm) << 9 | (off == last ? 0x2000u : 0u));
                // (the walk only: for the Gaussians of its lists the plan zeroes such slots itself, tl_emit)
                if (!from_list && n_i == 0u && slot != BT_WIDE && !use_atomics) {   // the ellipse misses the tile: its slot is still read
                    float4 *o = reinterpret_cast<float4 *>(V.part + ((size_t)cj[k] * P.part_k + slot) * 8);
                    o[0] = make_float4(0.f, 0.f, 0.f, 0.f);
                    o[1] = make_float4(0.f, 0.f, 0.f, 0.f);
                }
            }
        }
        __syncthreads();
        const unsigned nsurv = (unsigned)__builtin_amdgcn_readfirstlane((int)s_cnt[0]);
        const unsigned nitems = (unsigned)__builtin_amdgcn_readfirstlane((int)s_cnt[1]);
        (void)nsurv;
        // ---- level 2: chunks of <= 64 items, cut where a Gaussian's items end, CLAIMED by the waves from one queue ----
        // (a static split of the item list between the waves leaves every wave a ragged last chunk: with ~150 items per wave
        // that is 3-3.5 chunk iterations for 2.4 chunks of work, the largest single loss of this kernel.  A chunk's end
        // depends on its items, so a wave reads the queue head, finds its cut and claims [head, cut] with a compare-and-swap.)
        for (;;) {
            unsigned p0, it = 0u;
            int tlast = 0;
            for (;;) {
                p0 = (unsigned)__builtin_amdgcn_readfirstlane((int)*(volatile unsigned *)&s_cnt[2]);
                if (p0 >= nitems) break;
                const unsigned idx = p0 + (unsigned)lane;
                it = idx < nitems ? s_items[idx] : 0u;
                const unsigned long long tails = __ballot(idx < nitems && (it & 0x2000u));
                tlast = 63 - __builtin_clzll(tails);                  // (the list ends on a marked item: tails != 0)
                unsigned got = 0u;
                if (lane == 0) got = atomicCAS(&s_cnt[2], p0, p0 + (unsigned)tlast + 1u);
                if ((unsigned)__builtin_amdgcn_readfirstlane((int)got) == p0) break;
            }
            if (p0 >= nitems) break;
            const bool valid = lane <= tlast;
            const unsigned lidx = it & 0x1ffu, q = (it >> 9) & 15u;
            float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            unsigned j = 0u;
            unsigned e = 0u;
            if (valid) {
                e = s_list[lidx];
                j = e & 0x7fffffffu;
            }
            // (the dmax test costs an instruction per pixel pair: only chunks holding a Gaussian that needs it pay)
            if (BOUNDED && __ballot(valid && (e >> 31)) != 0ull) {
                if (valid) bt_eval<true>(V, j, (e >> 31) ? P.dmax : INFINITY, s_g + q * BT_QSTRIDE, s_px + (q & 3u) * 8u, s_py + (q >> 2) * 8u, a);
            } else {
                if (valid) bt_eval<false>(V, j, INFINITY, s_g + q * BT_QSTRIDE, s_px + (q & 3u) * 8u, s_py + (q >> 2) * 8u, a);
            }
            // add up the items of each Gaussian (adjacent lanes, at most NQ): three or four shuffle steps; its first lane gets the total
            const unsigned key = valid ? lidx : 0xffffu;
#pragma unroll
            for (int o = 1; o < NQ; o <<= 1) {
                const bool same = (unsigned)__shfl_down((int)key, o) == key && lane + o < 64;
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const float other = __shfl_down(a[k], o);
                    a[k] += same ? other : 0.f;
                }
            }
            // (the shuffle must run with every lane enabled: a lane that has been switched off by a short-circuit
            // supplies 0 to its neighbour)
            const unsigned prev = (unsigned)__shfl_up((int)key, 1);
            const bool head = valid && (lane == 0 || prev != key);
            if (head) {
                const unsigned slot = s_slot[lidx];
                if (slot != BT_WIDE && !use_atomics) {
                    float4 *o = reinterpret_cast<float4 *>(V.part + ((size_t)j * P.part_k + slot) * 8);
                    o[0] = make_float4(a[0], a[1], a[2], a[3]);
                    o[1] = make_float4(a[4], a[5], a[6], a[7]);
                } else {
#pragma unroll
                    for (int k = 0; k < 8; ++k) atomicAdd(V.sums + 8 * (size_t)j + k, a[k]);
                }
            }
        }
        __syncthreads();
        if (tid < 3) s_cnt[tid] = 0u;
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void k_bwd_gather(Params P, PlanView V, int use_atomics, float *__restrict__ g_sigmas,
                                                    float *__restrict__ g_coords, float *__restrict__ g_colors)
{
    const unsigned j = blockIdx.x * 256u + threadIdx.x;
    if (j >= (unsigned)P.s) return;
    float o[8];
    const unsigned i = bwd_gather(P, V, j, use_atomics != 0, o);
    float *pc = g_coords + (size_t)i * stride2(P), *ps = g_sigmas + (size_t)i * stride3(P), *pk = g_colors + (size_t)i * stride3(P);
    if (P.flags & GSASR_FLAG_OVERWRITE_GRADS) {
        pc[0] = o[0]; pc[1] = o[1]; ps[0] = o[2]; ps[1] = o[3]; ps[2] = o[4]; pk[0] = o[5]; pk[1] = o[6]; pk[2] = o[7];
    } else {   // (one thread per Gaussian: a plain read-modify-write)
        pc[0] += o[0]; pc[1] += o[1]; ps[0] += o[2]; ps[1] += o[3]; ps[2] += o[4]; pk[0] += o[5]; pk[1] += o[6]; pk[2] += o[7];
    }
}

}  // namespace

namespace gsasr_detail {

// mode: 0 = Gaussian-stationary, 1 = tile-stationary with slots, 2 = tile-stationary with atomics, 3 = home-tile
int bwd_mode(const gsasr_dims *dims, const Layout &L)
{
    // Default: whatever the plan was made for (bwd_wants_tile).  Measured on MI355X the two kernels are
    // within ~10% of each other at every scale -- both are bound by VALU issue: Gaussian-stationary ahead for GSASR's
    // LR-pixel sized Gaussians at x4 (38.4 vs 38.8 + 5.0 us gather at config 2), tile-stationary ahead from x8 up
    // (config 4: 1.67 vs 1.79 ms); the tile-stationary one is deterministic and reads the planar gradient autograd returns.
    const unsigned f = dims->flags;
    int mode = L.part_k > 0 ? 1 : 0;      // a plan with slots was made for the tile-stationary kernel (bwd_wants_tile)
    if (f & GSASR_FLAG_BWD_GAUSSIAN) mode = 0;
    else if (f & GSASR_FLAG_BWD_ATOMIC) mode = 2;
    else if ((f & GSASR_FLAG_BWD_HOME) && !(f & GSASR_FLAG_CHW_GRAD)) mode = 3;
    else if (f & (GSASR_FLAG_BWD_TILE | GSASR_FLAG_CHW_GRAD)) mode = 1;
    else if (const unsigned rf = registered_choice(dims).flags & (GSASR_FLAG_BWD_GAUSSIAN | GSASR_FLAG_BWD_TILE | GSASR_FLAG_BWD_HOME))
        mode = (rf & GSASR_FLAG_BWD_TILE) ? 1 : (rf & GSASR_FLAG_BWD_HOME) ? 3 : 0;
    else if (bwd_env()) mode = bwd_env() - 1;
    else if (mode == 0 && bwd_wants_home(dims)) mode = 3;
    if (L.part_k == 0 && mode == 1) mode = (f & GSASR_FLAG_CHW_GRAD) ? 2 : 0;   // a forward-only plan has no slots
    return mode;
}

// Backward of the splat.  With `gather` the kernel-frame gradients are written (or added) to g_*; without it a
// tile-stationary run stops after the tile kernel and the caller fuses the gather into its next kernel
// (k_prologue_bwd_gather) -- *mode_out tells which kernel ran.
int splat_backward(const float *sigmas, const float *coords, const float *colors, const float *grad_img, float *g_sigmas,
                   float *g_coords, float *g_colors, const gsasr_dims *dims, const void *workspace, size_t workspace_bytes,
                   void *stream, bool gather, int *mode_out)
{
    Layout L;
    if (int rc = check_ws(dims, workspace, workspace_bytes, L)) return rc;
    const int mode = bwd_mode(dims, L);
    if (mode_out) *mode_out = mode;
    if (dims->flags & GSASR_FLAG_FORWARD_ONLY) return fail(GSASR_ERR_PLAN, "the plan was made with GSASR_FLAG_FORWARD_ONLY: it holds no backward records");
    if (dims->s == 0) return GSASR_OK;
    if (gather && (!g_sigmas || !g_coords || !g_colors)) return fail(GSASR_ERR_ARG, "null pointer");
    if (mode == 0 && (dims->flags & GSASR_FLAG_CHW_GRAD))
        return fail(GSASR_ERR_ARG, "GSASR_FLAG_CHW_GRAD needs the tile-stationary backward");
    hipStream_t st = (hipStream_t)stream;
    const Params P = make_params(dims, L);
    const PlanView V = make_view(L, const_cast<void *>(workspace));
    const int rows = dims->row1 - dims->row0;
    if (rows > 0 && !grad_img) return fail(GSASR_ERR_ARG, "null pointer");
    if (mode == 0 || mode == 3) {
        if (!sigmas || !coords || !colors || !g_sigmas || !g_coords || !g_colors) return fail(GSASR_ERR_ARG, "null pointer");
        if (rows == 0) {  // empty band: the gradient is zero
            if (dims->flags & GSASR_FLAG_OVERWRITE_GRADS) {
                const size_t e3 = (dims->flags & GSASR_FLAG_STRIDE8) ? 0 : sizeof(float) * 3 * (size_t)dims->s;
                if (!e3) {
                    HIP_TRY(hipMemsetAsync(g_sigmas, 0, sizeof(float) * 8 * (size_t)dims->s, st));   // one packed [s,8] array
                } else {
                    HIP_TRY(hipMemsetAsync(g_sigmas, 0, e3, st));
                    HIP_TRY(hipMemsetAsync(g_coords, 0, sizeof(float) * 2 * (size_t)dims->s, st));
                    HIP_TRY(hipMemsetAsync(g_colors, 0, e3, st));
                }
            }
            return GSASR_OK;
        }
        if (mode == 3) {
            // home-tile kernel (splat_backward_home.hip).  Tile shape by density: 32 x 16-px tiles with eight waves where a tile
            // holds hundreds of Gaussians (GSASR's 16 per LR pixel), larger tiles with four waves for sparse plans -- a round
            // should find a wave's worth of Gaussians per wave.  development: GSASR_SPLAT_HOME_VARIANT=0|1|2|3
            static const int var_env = dev_switch("GSASR_SPLAT_HOME_VARIANT") ? atoi(dev_switch("GSASR_SPLAT_HOME_VARIANT")) : -1;
            const double per_cell = (double)dims->s / (double)(L.ncells > 0 ? L.ncells : 1);
            int variant = var_env >= 0 ? var_env : per_cell >= 64.0 ? 0 : per_cell >= 24.0 ? 1 : 2;
            if (var_env < 0 && variant == 0) {
                // 32 x 16-px tiles run two workgroups of eight waves per CU: 512 at a time.  Where their number leaves the last set
                // mostly empty (768^2: 1152 = 2.25 sets; 896^2: 3.06) one-cell tiles with four waves even the tail out: 768^2
                // 240 -> 233 us, 896^2 325 -> 309, 640^2 191 -> 173; at whole sets (512^2, 1024^2) the larger tile is 2-3% ahead
                // (profiles/r06_home_default.txt)
                const int cps = P.batch > 1 ? P.slot / CELL : P.ncy;
                const long n0 = (long)((P.ncx + 1) / 2) * (long)cps * (long)P.batch;
                const long over = n0 % 512;
                if (n0 > 512 && over != 0 && over <= 384) variant = 3;      // (up to one set: nothing to even out)
            }
            return launch_bwd_home(P, V, grad_img, g_sigmas, g_coords, g_colors, variant, st);
        }
        // Eight Gaussians per wave (k_render_bwd8): built in round 5, parity-green, and SLOWER than one wave per Gaussian --
        // config 2 39.4 vs 30.7 us, 16 Gaussians per LR pixel 475 vs 401, the config-5 canvas 271 vs 235
        // (profiles/history/r05_bwd8.txt): the eight windows of a wave differ (a wave runs max strips x max row pairs: 52 trips for a
        // mean of 30) and its trips are dependent round trips.  Kept behind the development switch GSASR_SPLAT_BWD8=1 (slabs
        // whose byte offsets fit 32 bits), exercised by tests/test_rows_vs_oracle.py.
        static const int bwd8_env = dev_switch("GSASR_SPLAT_BWD8") ? atoi(dev_switch("GSASR_SPLAT_BWD8")) : -1;
        const bool fits32 = (double)rows * (double)dims->w * 12.0 < 4294967295.0;
        if (fits32 && bwd8_env == 1) {
            const dim3 grid8((unsigned)(((size_t)dims->s * 8 + 255) / 256)), block8(256);
            if (P.bounded) hipLaunchKernelGGL(k_render_bwd8<true>, grid8, block8, 0, st, P, V, grad_img, g_sigmas, g_coords, g_colors);
            else hipLaunchKernelGGL(k_render_bwd8<false>, grid8, block8, 0, st, P, V, grad_img, g_sigmas, g_coords, g_colors);
            HIP_TRY(hipGetLastError());
            return GSASR_OK;
        }
        const dim3 grid((unsigned)((dims->s + BWD_WAVES - 1) / BWD_WAVES)), block(64 * BWD_WAVES);
        // Two instantiations of the same sweep (identical results): with the two-trip unrolled loop (88 VGPRs, 5 waves
        // per SIMD) for windows of many trips, without it (70 VGPRs, 7 waves) for small windows.  The window sizes are on
        // the device; GSASR's Gaussians are LR-pixel sized, so pixels per Gaussian is a good proxy (x4: 16, x8: 64).
        static const int unroll_env = dev_switch("GSASR_SPLAT_BWD_UNROLL") ? atoi(dev_switch("GSASR_SPLAT_BWD_UNROLL")) : -1;   // development: 0 | 1
        const bool unroll = unroll_env >= 0 ? unroll_env != 0 : (double)rows * (double)dims->w >= BWD_UNROLL_MIN * (double)dims->s;
        // (Measured dead ends, git history: two Gaussians per wave one after the other, side by side in half waves, and --
        // round 3 -- sharing every gradient load over the union of their windows: 44-50 us against 37 us at config 2; rows
        // or a cell's window staged in LDS; a planar-gradient sweep.  A wave's life is its chain of dependent round trips:
        // what helped was running the sweep one trip ahead)
#define GSASR_BWD(B, U) hipLaunchKernelGGL((k_render_bwd<B, U>), grid, block, 0, st, P, V, grad_img, g_sigmas, g_coords, g_colors)
        if (P.bounded) { if (unroll) GSASR_BWD(true, true); else GSASR_BWD(true, false); }
        else { if (unroll) GSASR_BWD(false, true); else GSASR_BWD(false, false); }
#undef GSASR_BWD
        HIP_TRY(hipGetLastError());
        return GSASR_OK;
    }
    if (L.part_k == 0)
        // the atomic variant, or a tile-stationary backward asked of a plan that was not made for it: the plan may have
        // left the accumulators alone (k_bin zeroes them only where it knows they will be used)
        HIP_TRY(hipMemsetAsync(V.sums, 0, (size_t)dims->s * 32, st));
    if (rows > 0) {
        const int bth = 1 << P.bt_hlog;
        const int tiles_x = (dims->w + BT_W - 1) / BT_W, tiles_y = (rows + bth - 1) / bth;
        dim3 grid((unsigned)tiles_x * (unsigned)tiles_y), block((unsigned)BT_THREADS << (P.bt_hlog - 4));
        // small rounds + five waves per SIMD from 32 HR pixels per Gaussian up (where this kernel is the default)
        const bool sparse = (double)rows * (double)dims->w >= 32.0 * (double)dims->s;
        // (the plan's tile lists serve this kernel when their tiles are its tiles)
        const bool lists = L.tl_ok && L.tl_hlog == P.bt_hlog;
#define GSASR_BT(B, C, H) do { if (lists) hipLaunchKernelGGL((k_render_bwd_tile<B, C, H, true>), grid, block, 0, st, P, V, grad_img, tiles_x, mode == 2); \
                               else hipLaunchKernelGGL((k_render_bwd_tile<B, C, H, false>), grid, block, 0, st, P, V, grad_img, tiles_x, mode == 2); } while (0)
#define GSASR_BT2(B, C) do { if (P.bt_hlog == 5) GSASR_BT(B, ((C) / 2 > 0 ? (C) / 2 : 1), 5); else GSASR_BT(B, C, 4); } while (0)
        if (lists && !sparse && P.bt_hlog == 4 && tl_dense(dims)) {   // dense plans from lists: four waves per tile (rounds of 512 entries)
            block = dim3(256);
            if (P.bounded) hipLaunchKernelGGL((k_render_bwd_tile<true, 2, 4, true, 4>), grid, block, 0, st, P, V, grad_img, tiles_x, mode == 2);
            else hipLaunchKernelGGL((k_render_bwd_tile<false, 2, 4, true, 4>), grid, block, 0, st, P, V, grad_img, tiles_x, mode == 2);
        } else if (P.bounded) { if (sparse) GSASR_BT2(true, 4 / BT_WAVES); else GSASR_BT2(true, 8 / BT_WAVES); }
        else { if (sparse) GSASR_BT2(false, 4 / BT_WAVES); else GSASR_BT2(false, 8 / BT_WAVES); }
#undef GSASR_BT2
#undef GSASR_BT
        HIP_TRY(hipGetLastError());
    }
    if (gather) {
        // (an empty band left no slots behind: the gather then only sees the zero accumulators)
        Params Pg = P;
        if (rows == 0) Pg.part_k = 0;
        hipLaunchKernelGGL(k_bwd_gather, dim3((unsigned)((dims->s + 255) / 256)), dim3(256), 0, st, Pg, V,
                           (int)(mode == 2 || rows == 0), g_sigmas, g_coords, g_colors);
        HIP_TRY(hipGetLastError());
    }
    return GSASR_OK;
}

}  // namespace gsasr_detail

extern "C" {

int gsasr_splat_backward(const float *sigmas, const float *coords, const float *colors, const float *grad_img,
                         float *g_sigmas, float *g_coords, float *g_colors, const gsasr_dims *dims,
                         const void *workspace, size_t workspace_bytes, void *stream)
{
    return splat_backward(sigmas, coords, colors, grad_img, g_sigmas, g_coords, g_colors, dims, workspace, workspace_bytes,
                          stream, true, nullptr);
}

}  // extern "C"
