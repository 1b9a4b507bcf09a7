// splat_bwd_sweep.h -- the wave-per-Gaussian backward sweep (bwd_sweep, bwd_item, the scalar record fetch, the wave reduction and the
// gradient write): shared by the Gaussian-stationary kernels of splat_backward.hip and by the home-tile kernel of
// splat_backward_home.hip, which sweeps the few Gaussians that do not fit its staged region with the same code.
#ifndef GSASR_SPLAT_BWD_SWEEP_H
#define GSASR_SPLAT_BWD_SWEEP_H
#include "splat_common.h"

namespace {
using namespace gsasr_detail;

// ---------------------------------------------------------------------------------------------------
// backward: one wave64 per Gaussian (cell order, so neighbouring waves read neighbouring pixels)
// ---------------------------------------------------------------------------------------------------
// Sweep the pixel window [c0,c0+bw) x [r0,r1] of one Gaussian with a wave.  Lanes are laid LX = 16/32/64
// wide (the narrowest that covers bw, a template parameter so all the lane geometry is constant) and
// 64/LX rows deep; a lane keeps ONE column (u = dx/sx is a lane constant) and handles TWO rows per trip, so
// the per-pixel arithmetic is 2-wide packed fp32.  Because u is constant per lane only three sums over
// rows are accumulated per pixel column,
//     M0 = sum q,  N1 = sum q*B,  N2 = sum q*B^2,      q = v * <grad, colour>,  B = dy/sy - rho u,
// and expanded at the end of the column (see below).
// The dy/sy values of a 64-row block are staged in LDS (256 B per wave); full trips carry no masks or
// address clamps, the ragged last trip is peeled.
// acc[] = {qA, qB, quA, qvB, qAB, Cr, Cg, Cb} (per lane, summed over the wave by the caller).
struct BwdRow {
    v2f m1, m2, k01;         // moments N1, N2 (row pair); colour sums r, g of the first row
    float ka2, kb0, kb1, kb2;  // colour sums: b of the first row, r g b of the second
};

struct Grad6 {  // the three gradient channels of the two pixels (rows Y, Y+RPI) a lane owns in one trip
    float a0, a1, a2, b0, b1, b2;
};

typedef unsigned u3v __attribute__((ext_vector_type(3)));
typedef unsigned u2v __attribute__((ext_vector_type(2)));
typedef unsigned u4v __attribute__((ext_vector_type(4)));

// Two 12-byte pixels through a raw buffer resource: address = base(SGPR x4) + per-lane byte offset (one VGPR per
// row of the pair, constant over the sweep) + ONE running row offset (SGPR), so a trip spends no VALU instruction and
// a single scalar add on addressing, and reads past the end of the slab return 0 instead of faulting.  (An instruction
// added to a trip of ANY kind, scalar or vector, costs 0.35 us at config 2: the trips are the wave's dependent chain, and
// that chain at seven waves per SIMD is the run time -- DESIGN.md 3c (c), (d).)
__device__ __forceinline__ Grad6 bwd_load(__amdgpu_buffer_rsrc_t rsrc, int voff, int voff_b, int soff_a)
{
    const int soff_b = soff_a;
#ifdef BWD_EXP_NOLOAD   // what-if build (tools/whatif.sh): synthetic gradient values instead of the two loads of a trip
    Grad6 s;
    s.a0 = __int_as_float(voff | 0x3f000000); s.a1 = __int_as_float(soff_a | 0x3f000000); s.a2 = 0.25f;
    s.b0 = __int_as_float(voff_b | 0x3f000000); s.b1 = 0.5f; s.b2 = __int_as_float(soff_b | 0x3e000000);
    return s;
#endif
#ifdef BWD_EXP_PAIRPLANAR   // what-if build (tools/whatif.sh): the cost of a gradient stored as row-pair planes {r0,r1 | g0,g1 | b0,b1}
    {                       // per (column, row pair) -- one 16-byte + one 8-byte load, three natural register pairs (values are WRONG)
        const u4v a4 = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, soff_a, 0);
        const u2v b2 = __builtin_amdgcn_raw_buffer_load_b64(rsrc, voff_b, soff_b, 0);
        Grad6 p;
        p.a0 = __uint_as_float(a4.x); p.a1 = __uint_as_float(a4.y); p.a2 = __uint_as_float(a4.z);
        p.b0 = __uint_as_float(a4.w); p.b1 = __uint_as_float(b2.x); p.b2 = __uint_as_float(b2.y);
        return p;
    }
#endif
    const u3v a = __builtin_amdgcn_raw_buffer_load_b96(rsrc, voff, soff_a, 0);
    const u3v b = __builtin_amdgcn_raw_buffer_load_b96(rsrc, voff_b, soff_b, 0);
    Grad6 g;
    g.a0 = __uint_as_float(a.x); g.a1 = __uint_as_float(a.y); g.a2 = __uint_as_float(a.z);
    g.b0 = __uint_as_float(b.x); g.b1 = __uint_as_float(b.y); g.b2 = __uint_as_float(b.z);
    return g;
}

template <bool TEST, bool TAIL>
__device__ __forceinline__ void bwd_trip(BwdRow &R, const Grad6 g, v2f dyn, v2f dyraw, bool ok1, bool ok2, float K0,
                                         float nK1, float rho_u, float cr, float cg, float cb, float dmax)
{
#ifdef BWD_EXP_NOMATH   // what-if build (tools/whatif.sh): the loads are consumed by six adds, the trip's arithmetic is gone
    R.k01 += (v2f){g.a0 + g.b0, g.a1 + g.b1};
    R.ka2 += g.a2 + g.b2 + dyn.x;
    return;
#endif
    // With u = dx/sx, v = dy/sy and B = v - rho u (the residual of v about its conditional mean given u) the
    // quadratic form completes to  u^2 - 2 rho u v + v^2 = (1-rho^2) u^2 + B^2,  so the exponent is
    //   log2(e) w1 (...) = K0 - K1 B^2,   K0 = -log2(e)/2 u^2 (lane constant),  K1 = log2(e)/2 / (1-rho^2),
    // and the same B feeds the gradient moments: nothing here cancels as |rho| -> 1.
    const v2f Bv = dyn - rho_u;
    const v2f pw = (Bv * nK1) * Bv + K0;
    v2f v = {__builtin_amdgcn_exp2f(pw.x), __builtin_amdgcn_exp2f(pw.y)};
    if (TEST || TAIL) {
        v.x = ((!TAIL || ok1) && (!TEST || fabsf(dyraw.x) <= dmax)) ? v.x : 0.f;
        v.y = ((!TAIL || ok2) && (!TEST || fabsf(dyraw.y) <= dmax)) ? v.y : 0.f;
    }
#ifdef BWD_EXP_PAIRPLANAR
    {   // three packed FMAs for <grad, colour> and three for the colour sums instead of six + five
        const v2f pr = {g.a0, g.a1}, pg = {g.a2, g.b0}, pb = {g.b1, g.b2};
        const v2f gpp = pb * cb + (pg * cg + pr * cr);
        const v2f qq = gpp * v, qqB = qq * Bv;
        R.m1 += qqB;
        R.m2 += qqB * Bv;
        R.k01 += pr * v;
        v2f kg = {R.ka2, R.kb0}, kb = {R.kb1, R.kb2};      // (two more packed accumulators)
        kg += pg * v;
        kb += pb * v;
        R.ka2 = kg.x; R.kb0 = kg.y; R.kb1 = kb.x; R.kb2 = kb.y;
        return;
    }
#endif
    // six consecutive-in-memory floats per pixel pair are used as they land: no register shuffling
    const v2f gp = {fmaf(g.a2, cb, fmaf(g.a1, cg, g.a0 * cr)), fmaf(g.b2, cb, fmaf(g.b1, cg, g.b0 * cr))};  // gs.cu:150
    const v2f q = gp * v, qB = q * Bv;
    // (M0 = sum q is not accumulated: it is <colour, colour sums>, formed once per column strip)
    R.m1 += qB;
    R.m2 += qB * Bv;
    // Only {a0, a1} is an aligned register pair as the two 12-byte loads land; the other four take scalar FMAs
    // (pairing them up costs five v_mov per trip -- more than the two packed operations save).
    R.k01 += (v2f){g.a0, g.a1} * v.x;
    R.ka2 = fmaf(g.a2, v.x, R.ka2);
    R.kb0 = fmaf(g.b0, v.y, R.kb0);
    R.kb1 = fmaf(g.b1, v.y, R.kb1);
    R.kb2 = fmaf(g.b2, v.y, R.kb2);
}

// LX = 16 / 21 / 32 / 64 columns: 4 / 3 / 2 / 1 row slots of lanes, i.e. 8 / 6 / 4 / 2 rows per trip.  LX = 21 (round 5; 63 lanes, the
// last one idle) is for the windows of 17..21 columns -- over half of GSASR's at x4 since the data-derived cutoff narrowed them:
// 6 rows per trip instead of 4, so a 20-row window takes 4 trips instead of 5-6 and 21 of 21 columns work instead of 21 of 32.
template <bool TEST, int LX, bool UNROLL>
__device__ __forceinline__ void bwd_sweep(int c0, int bw, int r0, int r1, int lane, const Params &P,
                                          const float *__restrict__ pxt, const float *__restrict__ pyt,
                                          const float *__restrict__ grad, float x, float y, float cr, float cg,
                                          float cb, float cinv, float rho, float kappa, float isx, float isy,
                                          float *spy, float (&acc)[8])
{
    constexpr int RPI = 64 / LX, RPT = 2 * RPI;        // row slots of lanes; rows per trip
    constexpr float HALF_LOG2E = 0.72134752044448170368f;
    const int col = lane % LX, rsub = lane / LX;
    const unsigned pitchb = (unsigned)P.w * 12u;   // bytes per gradient row (< 2^19); all offsets below are unsigned 32 x 32 -> 64
    const float nK1 = -HALF_LOG2E * cinv;
    // issued together with the px load below: one round trip for both tables instead of two dependent ones
    const float py_first = pyt[min(r0 + lane, r1)];
    for (int strip = 0; strip < bw; strip += 64) {
        const int cc = strip + col;
        const int X = c0 + min(cc, bw - 1);
        const float dx = pxt[X] - x;
        // lanes outside the window (or, with TEST, outside the dmax box in x) are switched off through K0:
        // the exponent becomes -inf, v = 0 exactly, and every product with it is 0
        const bool inx = cc < bw && (LX * RPI == 64 || rsub < RPI) && (!TEST || fabsf(dx) <= P.dmax);
        const float u = dx * isx, rho_u = rho * u;
        const float K0 = inx ? -HALF_LOG2E * u * u : -INFINITY;
        BwdRow R;
        R.m1 = R.m2 = R.k01 = (v2f){0.f, 0.f};
        R.ka2 = R.kb0 = R.kb1 = R.kb2 = 0.f;
        const int voff = (int)((unsigned)X * 12u + (unsigned)rsub * pitchb);
        const int halfb = (int)((unsigned)RPI * pitchb);
        for (int rb = r0; rb <= r1; rb += 64) {
            const int rend = min(r1, rb + 63);
            __builtin_amdgcn_wave_barrier();
            {   // per-row values of the block in LDS: v = dy/sy, and (TEST only) the raw dy for the exact box test
                const float dyr = (rb == r0 ? py_first : pyt[min(rb + lane, r1)]) - y;
                spy[lane] = dyr * isy;
                if (TEST) spy[64 + lane] = dyr;
            }
            __builtin_amdgcn_wave_barrier();
            const float *sp = spy + rsub;
            // buffer resource over the slab from row `rb` on (offsets stay far below 2^31 within a 64-row block)
            const char *blk = reinterpret_cast<const char *>(grad) + (unsigned long long)(unsigned)(rb - P.row0) * pitchb;
            const unsigned long long left = (unsigned long long)(unsigned)(P.row1 - rb) * pitchb;
            const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
                const_cast<char *>(blk), 0, (int)(left < 0x7fffffffu ? left : 0x7fffffffu), 0x00020000);
            int soff = 0;
            // trip counts up front: the loops below count down (one scalar add + compare + branch per iteration)
            const int nrows = rend - rb + 1, ntrip = nrows / RPT;
            const int voff_b = voff + halfb;
            // Lanes outside the window sit the trips out (exec mask): the backward is co-limited by the CU's
            // vector-memory pipe (two 768-byte loads per trip and wave, four SIMDs behind one L1), and idle
            // lanes would fetch gradient pixels only to multiply them by zero.
            if (inx) {
            // UNROLL: two trips per iteration, four gradient loads in flight before the first is consumed.  Pays
            // for windows of many trips (x8 and up); costs 18 VGPRs = two waves per SIMD, which small windows
            // (x4, 6 trips) need more: the host picks the instantiation (gsasr_splat_backward).
            int t = ntrip;
            for (; UNROLL && t >= 2; t -= 2, soff += 4 * halfb, sp += 4 * RPI) {
                const Grad6 g0 = bwd_load(rsrc, voff, voff_b, soff);
                const Grad6 g1 = bwd_load(rsrc, voff, voff_b, soff + 2 * halfb);
                const v2f n0 = {sp[0], sp[RPI]}, n1 = {sp[2 * RPI], sp[3 * RPI]};
                const v2f w0 = TEST ? (v2f){sp[64], sp[64 + RPI]} : n0, w1 = TEST ? (v2f){sp[64 + 2 * RPI], sp[64 + 3 * RPI]} : n1;
                bwd_trip<TEST, false>(R, g0, n0, w0, true, true, K0, nK1, rho_u, cr, cg, cb, P.dmax);
                bwd_trip<TEST, false>(R, g1, n1, w1, true, true, K0, nK1, rho_u, cr, cg, cb, P.dmax);
            }
            if (!UNROLL && t > 0) {
                // The plain loop runs ONE TRIP AHEAD: the two loads of trip k+1 are in flight while trip k is summed (two
                // register sets used alternately: no copies).  Left to the compiler every trip was a dependent round trip
                // -- issue, wait, sum -- and a wave's life at x4 is six of them: -5% at config 2, -4% on the config-5 crops,
                // at 71 VGPRs (seven waves per SIMD kept).  Two trips ahead spills (72-VGPR budget): +8%; the same rotation
                // in the unrolled instantiation: no gain (profiles/history/r03_bwd_experiments.txt).
#define GSASR_TRIP(G) { const v2f n0 = {sp[0], sp[RPI]}; const v2f w0 = TEST ? (v2f){sp[64], sp[64 + RPI]} : n0; \
                        bwd_trip<TEST, false>(R, G, n0, w0, true, true, K0, nK1, rho_u, cr, cg, cb, P.dmax); sp += 2 * RPI; }
                Grad6 ga = bwd_load(rsrc, voff, voff_b, soff);
                soff += 2 * halfb;
                for (; t >= 3; t -= 2) {
                    const Grad6 gb = bwd_load(rsrc, voff, voff_b, soff);
                    soff += 2 * halfb;
                    GSASR_TRIP(ga)
                    ga = bwd_load(rsrc, voff, voff_b, soff);
                    soff += 2 * halfb;
                    GSASR_TRIP(gb)
                }
                if (t == 2) {
                    const Grad6 gb = bwd_load(rsrc, voff, voff_b, soff);
                    soff += 2 * halfb;
                    GSASR_TRIP(ga)
                    GSASR_TRIP(gb)
                } else {
                    GSASR_TRIP(ga)
                }
#undef GSASR_TRIP
                t = 0;
            }
            for (; t > 0; --t, soff += 2 * halfb, sp += 2 * RPI) {   // (odd trip of the unrolled instantiation)
                const v2f n0 = {sp[0], sp[RPI]};
                const v2f w0 = TEST ? (v2f){sp[64], sp[64 + RPI]} : n0;
                bwd_trip<TEST, false>(R, bwd_load(rsrc, voff, voff_b, soff), n0, w0, true, true, K0, nK1, rho_u, cr,
                                      cg, cb, P.dmax);
            }
            if (nrows % RPT) {  // ragged last trip: rows past the window are masked (reads past the slab give 0)
                const int Yb = rb + ntrip * RPT;
                const int Ya = Yb + rsub, Yc = Ya + RPI;
                const int ia = min(Ya, rend) - rb, ic = min(Yc, rend) - rb;
                const v2f n0 = {spy[ia], spy[ic]};
                const v2f w0 = TEST ? (v2f){spy[64 + ia], spy[64 + ic]} : n0;
                bwd_trip<TEST, true>(R, bwd_load(rsrc, voff, voff_b, soff), n0, w0, Ya <= rend, Yc <= rend, K0, nK1,
                                     rho_u, cr, cg, cb, P.dmax);
            }
            }
        }
        // Expand the column's three sums M0 = sum q, N1 = sum q B, N2 = sum q B^2 (u = dx/sx is a lane constant,
        // A = u - rho v = u kappa - rho B, v = B + rho u):  sum qA, sum qB, sum q u A, sum q v B, sum q A B.
        // Every difference is formed between quantities of its own size, so nothing cancels as |rho| -> 1
        // (the plain monomial moments sum q dx^2, q dx dy, q dy^2 lose 1/(1-rho) digits there).
        const float Kr = R.k01.x + R.kb0, Kg = R.k01.y + R.kb1, Kb = R.ka2 + R.kb2;
        const float M0 = fmaf(Kb, cb, fmaf(Kg, cg, Kr * cr)), N1 = R.m1.x + R.m1.y, N2 = R.m2.x + R.m2.y;
        // (switched-off lanes have M0 = N1 = N2 = 0, but their u is meaningless: use 0)
        const float ue = inx ? u : 0.f, uk = ue * kappa;
        const float sA = uk * M0 - rho * N1;
        const float e[8] = {sA, N1, ue * sA, N2 + rho * ue * N1, uk * N1 - rho * N2,
                            Kr, Kg, Kb};
        // the first (usually only) 64-column strip assigns, so acc[] is not live during its sweep
#pragma unroll
        for (int k = 0; k < 8; ++k) acc[k] = strip == 0 ? e[k] : acc[k] + e[k];
    }
}

// Sum eight per-lane values over the wave through LDS: lanes park their 8 partials ([8][64] floats per wave),
// lane l then adds the 8 consecutive partials {l&7} of value {l>>3} (two ds_read_b128) and three butterfly
// steps (DPP) finish inside each 8-lane group.  Afterwards lane 8k holds the total of value k.  ~14 VALU
// instructions instead of ~45 for a register-only exchange network; the LDS pipe is otherwise idle here.
__device__ __forceinline__ float wave_sum8(const float (&a)[8], int lane, float *red)
{
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int k = 0; k < 8; ++k) red[k * 64 + lane] = a[k];
    __builtin_amdgcn_wave_barrier();
    const float4 u = *reinterpret_cast<const float4 *>(red + lane * 8);
    const float4 v = *reinterpret_cast<const float4 *>(red + lane * 8 + 4);
    float d = ((u.x + u.y) + (u.z + u.w)) + ((v.x + v.y) + (v.z + v.w));
    // lane 8k += lanes 8k+4, then +2, then +1, as DPP row shifts folded into the adds (a __shfl_xor is a
    // ds_bpermute round trip plus five address instructions each)
    d += dpp_row_shl<4>(d);
    d += dpp_row_shl<2>(d);
    d += dpp_row_shl<1>(d);
    return d;   // valid in lanes 8k only
}

__device__ __forceinline__ void bwd_write(float v, int lane, const Params &P, unsigned i, float *__restrict__ g_sigmas,
                                          float *__restrict__ g_coords, float *__restrict__ g_colors)
{
    if (lane & 7) return;
    const int k = lane >> 3;
    // one store (or atomic) through a per-lane pointer: the three arrays' bases are wave-uniform, pre-biased so that each
    // is indexed by k, and selected per lane -- three exec-masked branches cost twice the instructions
    float *pc = g_coords + (size_t)i * stride2(P), *ps = g_sigmas + (size_t)i * stride3(P) - 2,
          *pk = g_colors + (size_t)i * stride3(P) - 5;
    float *dst = (k < 2 ? pc : (k < 5 ? ps : pk)) + k;
    if (P.flags & GSASR_FLAG_OVERWRITE_GRADS) *dst = v;
    else atomicAdd(dst, v);   // fire-and-forget: the wave must not end on a load-add-store round trip
}

typedef unsigned u8v __attribute__((ext_vector_type(8)));

// Everything the sweep needs about Gaussian j, fetched by the SCALAR unit in one batch (one round trip).
// Left to the compiler these are vector loads + v_readfirstlane (the kernel also stores to the workspace, so
// it will not use the non-coherent scalar cache) issued as three dependent round trips, which was most of a
// wave's life.  The plan was written by an earlier kernel, so the scalar cache is coherent for it.
struct BwdRec {
    u8v bb;    // both bbox words: {c0|test|c1, r0|r1, spans.. | spans.., padded rows r0|r1 of the sweep, its columns c0|test|c1}
    u8v rec;   // {x, y, IX, NR | IY, r, g, b}
    u8v fin;   // {c, kappa, rho, 1/sx | 1/sy, px-table offset, sample, index}
};

__device__ __forceinline__ void bwd_fetch(const PlanView &V, unsigned j, BwdRec &R)
{
    const uint4 *pb = V.bbox + 2 * (size_t)j;
    const float4 *pr = V.rec + 2 * (size_t)j, *pf = V.fin + 2 * (size_t)j;
    asm volatile("s_load_dwordx8 %0, %3, 0x0\n\t"
                 "s_load_dwordx8 %1, %4, 0x0\n\t"
                 "s_load_dwordx8 %2, %5, 0x0\n\t"
                 "s_waitcnt lgkmcnt(0)"
                 : "=&s"(R.bb), "=&s"(R.rec), "=&s"(R.fin)
                 : "s"(pb), "s"(pr), "s"(pf)
                 : "memory");
}

// the same plus the two class boundaries cell_start[ncells], cell_start[ncells+1]
__device__ __forceinline__ void bwd_fetch_first(const PlanView &V, const unsigned *bounds, unsigned j, BwdRec &R, u2v &lim)
{
    const uint4 *pb = V.bbox + 2 * (size_t)j;
    const float4 *pr = V.rec + 2 * (size_t)j, *pf = V.fin + 2 * (size_t)j;
    asm volatile("s_load_dwordx2 %3, %7, 0x0\n\t"
                 "s_load_dwordx8 %0, %4, 0x0\n\t"
                 "s_load_dwordx8 %1, %5, 0x0\n\t"
                 "s_load_dwordx8 %2, %6, 0x0\n\t"
                 "s_waitcnt lgkmcnt(0)"
                 : "=&s"(R.bb), "=&s"(R.rec), "=&s"(R.fin), "=&s"(lim)
                 : "s"(pb), "s"(pr), "s"(pf), "s"(bounds)
                 : "memory");
}

template <bool BOUNDED, bool UNROLL>
__device__ __forceinline__ void bwd_item(unsigned j, const BwdRec &G, int chunk, bool atomic, int lane, const Params &P,
                                         const PlanView &V, const float *__restrict__ grad, float *spy, float *red,
                                         float *__restrict__ g_sigmas, float *__restrict__ g_coords,
                                         float *__restrict__ g_colors)
{
    unsigned bbx = G.bb[0];
    int c0 = (int)(bbx & 0x7fffu), c1 = (int)(bbx >> 16);
    if (c0 > c1) return;  // dead class (handled by the caller)
    if (chunk < 0) {      // the backward's own window (k_bin: min(tau', GSASR_SPLAT_GRAD_TAU)); row chunks of the large class keep the forward's
        bbx = G.bb[7];
        c0 = (int)(bbx & 0x7fffu);
        c1 = (int)(bbx >> 16);
    }
    int r0, r1;
    bool empty = false;
    if (chunk >= 0) {  // (row chunks of a large Gaussian must not overlap: they split the window's own rows)
        r0 = (int)(G.bb[1] & 0x7fffu);
        r1 = (int)(G.bb[1] >> 16);
        const int rpc = (r1 - r0 + NCH) / NCH;
        r0 = r0 + chunk * rpc;
        r1 = min(r1, r0 + rpc - 1);
        empty = r0 > r1;   // still counted as a finished chunk below
    } else {           // the plan's padded row range (whole trips; k_bin)
        r0 = (int)(G.bb[6] & 0xffffu);
        r1 = (int)(G.bb[6] >> 16);
    }
    const float x = __uint_as_float(G.rec[0]), y = __uint_as_float(G.rec[1]);
    const float cr = __uint_as_float(G.rec[5]), cg = __uint_as_float(G.rec[6]), cb = __uint_as_float(G.rec[7]);
    const float4 fa = make_float4(__uint_as_float(G.fin[0]), __uint_as_float(G.fin[1]), __uint_as_float(G.fin[2]),
                                  __uint_as_float(G.fin[3]));   // {c, kappa, rho, 1/sx}
    const float4 fb = make_float4(__uint_as_float(G.fin[4]), 0.f, 0.f, 0.f);   // {1/sy, ..}
    float a[8];
    const int bw = c1 - c0 + 1;
    const bool test = BOUNDED && (bbx & 0x8000u);
    float d = 0.f;
    if (!empty) {
#define GSASR_SWEEP(T, L) \
    bwd_sweep<T, L, UNROLL>(c0, bw, r0, r1, lane, P, V.px + G.fin[5], V.py, grad, x, y, cr, cg, cb, fa.x, fa.z, fa.y, fa.w, fb.x, spy, a)
#ifdef BWD_EXP_NOSWEEP   // what-if build (tools/whatif.sh): no sweep at all -- what remains is the record fetch, the wave reduction and the write
        for (int k = 0; k < 8; ++k) a[k] = __int_as_float((lane + k + bw) | 0x3f000000) * x;
#else
        if (bw <= 16) { if (test) GSASR_SWEEP(true, 16); else GSASR_SWEEP(false, 16); }
        else if (bw <= BWD_LX21_MAX) { if (test) GSASR_SWEEP(true, 21); else GSASR_SWEEP(false, 21); }
        else if (bw <= 32) { if (test) GSASR_SWEEP(true, 32); else GSASR_SWEEP(false, 32); }
        else { if (test) GSASR_SWEEP(true, 64); else GSASR_SWEEP(false, 64); }
#endif
#undef GSASR_SWEEP
        bwd_scale(a, fa.x, fa.w, fb.x);
#ifdef BWD_EXP_NOREDUCE   // what-if build: no wave reduction (lane 8k writes its own partial of value k)
        d = a[0];
        for (int k = 1; k < 8; ++k) d = (lane >> 3) == k ? a[k] : d;
#else
        d = wave_sum8(a, lane, red);   // lane 8k now holds gradient component k
#endif
    }
    if (atomic) {
        // Large class: the row chunks add into sums[] and count themselves; the wave that finishes the
        // last chunk takes the totals (re-arming the accumulators for the next backward) and writes the
        // gradient, so no separate finalize pass exists.
        if (!empty && (lane & 7) == 0) atomicAdd(V.sums + 8 * (size_t)j + (lane >> 3), d);
        __threadfence();
        unsigned prev = 0;
        if (lane == 0) prev = atomicAdd(&V.done[j], 1u);
        prev = (unsigned)__builtin_amdgcn_readfirstlane((int)prev);
        if (prev != (unsigned)(NCH - 1)) return;
        __threadfence();
        if ((lane & 7) == 0) d = atomicExch(V.sums + 8 * (size_t)j + (lane >> 3), 0.f);
        if (lane == 0) V.done[j] = 0u;
    }
    bwd_write(d, lane, P, G.fin[7], g_sigmas, g_coords, g_colors);
}

}  // namespace

#endif  // GSASR_SPLAT_BWD_SWEEP_H
