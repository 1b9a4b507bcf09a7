// splat_forward.hip -- forward kernels (pixel-stationary; search and tile-list variants) and gsasr_splat_forward
// (one translation unit of libgsasr_splat.so; gsasr_splat.hip has the overview of the whole pipeline)
#include "splat_common.h"

using namespace gsasr_detail;

namespace {

// ---------------------------------------------------------------------------------------------------
// forward: one wave64 per 8-wide x 16-tall pixel sub-tile (lane = column X, rows Y and Y+8, so the
// per-pair arithmetic is 2-wide packed fp32); four sub-tiles side by side per workgroup (32x16 px).
// Candidates are window-tested 64 at a time (one per lane); the records of the hits are compacted into a
// 2 KB per-wave LDS stage and then evaluated by all lanes from broadcast LDS reads (12 VALU instructions
// + 2 v_exp_f32 per record for 128 pixels).  Measured alternatives: fetching hit records with scalar
// loads (s_load_dwordx8, one or four in flight) was 4% slower at config 2 and 23% slower on small images.
// ---------------------------------------------------------------------------------------------------

// Evaluate `n` records staged in LDS (32 B each, broadcast reads).  Unlike scalar-memory loads, LDS reads
// return in order, so the compiler can keep several records in flight behind counted lgkmcnt waits.
template <bool TEST>
__device__ __forceinline__ void fwd_eval_one(const float4 a, const float4 b, float px, v2f py, float dmax, v2f &ar,
                                             v2f &ag, v2f &ab)
{
    // a = {x, y, IX, NR}, b = {IY, r, g, b}:  exponent (log2) = -U^2 - Bq^2,  U = IX dx,  Bq = IY dy + NR U   (k_bin)
    const float dx = px - a.x;
    const v2f dy = py - a.y;
    const float u = a.z * dx;
    const float k0 = -u * u, ru = a.w * u;
    const v2f bq = b.x * dy + ru;
    const v2f pw = k0 - bq * bq;
    v2f v = {__builtin_amdgcn_exp2f(pw.x), __builtin_amdgcn_exp2f(pw.y)};
    if (TEST) {
        const bool inx = fabsf(dx) <= dmax;
        v.x = (inx && fabsf(dy.x) <= dmax) ? v.x : 0.f;
        v.y = (inx && fabsf(dy.y) <= dmax) ? v.y : 0.f;
    }
    ar += v * b.y;
    ag += v * b.z;
    // ab += v * b.w with b.w read as the HIGH half of the (g, b) register pair.  The compiler folds four of the
    // five broadcasts {y, C, r, g, b} into op_sel but copies the fifth with a v_mov whatever the record order (one
    // VALU slot in 13.5 per record; -3% at config 2, -8% at config 3).  Inline asm is outside the compiler's
    // hazard recogniser, and this instruction may be scheduled right behind the v_exp_f32 that produces `v`
    // (trans-use hazard on gfx950: one wait state) -- hence the s_nop, without which results are garbage.
    {
        const v2f gb = {b.z, b.w};
        asm("s_nop 0\n\tv_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0]" : "+v"(ab) : "v"(gb), "v"(v));
    }
}

// The same evaluation split into its column part and its row part: the pixels of a lane of the WIDE forward
// (k_render_fwd16: four per lane, one column) share dx, U, -U^2 and NR U of a record -- four of the ten instructions.
struct FwdCol {
    float k0, ru;   // -U^2, NR U
    bool inx;       // (TEST) |dx| <= dmax
};

template <bool TEST>
__device__ __forceinline__ FwdCol fwd_eval_col(const float4 a, float px, float dmax)
{
    const float dx = px - a.x;
    const float u = a.z * dx;
    FwdCol c;
    c.k0 = -u * u;
    c.ru = a.w * u;
    c.inx = !TEST || fabsf(dx) <= dmax;
    return c;
}

template <bool TEST>
__device__ __forceinline__ void fwd_eval_row(const FwdCol c, const float4 a, const float4 b, v2f py, float dmax, v2f &ar,
                                             v2f &ag, v2f &ab)
{
    const v2f dy = py - a.y;
    const v2f bq = b.x * dy + c.ru;
    const v2f pw = c.k0 - bq * bq;
    v2f v = {__builtin_amdgcn_exp2f(pw.x), __builtin_amdgcn_exp2f(pw.y)};
    if (TEST) {
        v.x = (c.inx && fabsf(dy.x) <= dmax) ? v.x : 0.f;
        v.y = (c.inx && fabsf(dy.y) <= dmax) ? v.y : 0.f;
    }
    ar += v * b.y;
    ag += v * b.z;
    {   // (see fwd_eval_one: b.w as the high half of the (g, b) pair; s_nop for the trans-use hazard)
        const v2f gb = {b.z, b.w};
        asm("s_nop 0\n\tv_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0]" : "+v"(ab) : "v"(gb), "v"(v));
    }
}

// records [beg, end) of the stage on the lane's pixels; acc = {rA, gA, bA, rB, gB, bB}.  HALVES: which of the lane's two row
// pairs the records of this list reach -- 1 = pair A (the sub-tile's rows 0..7), 2 = pair B (rows 8..15), 3 = both
// (k_render_fwd16 sorts a chunk's hits by that: a window that ends in the upper half costs one row part, not two).
template <bool TEST, int HALVES>
__device__ __forceinline__ void fwd_eval_lds16(const float4 *__restrict__ st, int beg, int end, float px, v2f pyA, v2f pyB,
                                               float dmax, v2f (&acc)[6])
{
    int i = beg;
    for (; i + 1 < end; i += 2) {   // two records per iteration so their dependent chains interleave
        const float4 a0 = st[2 * i], b0 = st[2 * i + 1], a1 = st[2 * i + 2], b1 = st[2 * i + 3];
        const FwdCol c0 = fwd_eval_col<TEST>(a0, px, dmax), c1 = fwd_eval_col<TEST>(a1, px, dmax);
        if (HALVES & 1) {
            fwd_eval_row<TEST>(c0, a0, b0, pyA, dmax, acc[0], acc[1], acc[2]);
            fwd_eval_row<TEST>(c1, a1, b1, pyA, dmax, acc[0], acc[1], acc[2]);
        }
        if (HALVES & 2) {
            fwd_eval_row<TEST>(c0, a0, b0, pyB, dmax, acc[3], acc[4], acc[5]);
            fwd_eval_row<TEST>(c1, a1, b1, pyB, dmax, acc[3], acc[4], acc[5]);
        }
    }
    if (i < end) {
        const float4 a = st[2 * i], b = st[2 * i + 1];
        const FwdCol c = fwd_eval_col<TEST>(a, px, dmax);
        if (HALVES & 1) fwd_eval_row<TEST>(c, a, b, pyA, dmax, acc[0], acc[1], acc[2]);
        if (HALVES & 2) fwd_eval_row<TEST>(c, a, b, pyB, dmax, acc[3], acc[4], acc[5]);
    }
}

template <bool TEST>
__device__ __forceinline__ void fwd_eval_lds(const float4 *__restrict__ st, int beg, int end, float px, v2f py,
                                             float dmax, v2f &ar, v2f &ag, v2f &ab)
{
    int i = beg;
    for (; i + 1 < end; i += 2) {   // two records per iteration so their dependent chains interleave
        const float4 a0 = st[2 * i], b0 = st[2 * i + 1], a1 = st[2 * i + 2], b1 = st[2 * i + 3];
        fwd_eval_one<TEST>(a0, b0, px, py, dmax, ar, ag, ab);
        fwd_eval_one<TEST>(a1, b1, px, py, dmax, ar, ag, ab);
    }
    if (i < end) fwd_eval_one<TEST>(st[2 * i], st[2 * i + 1], px, py, dmax, ar, ag, ab);
}

// RECORD-PAIR evaluation (round 5).  fwd_eval_one packs the two PIXELS of a lane: of its instructions per record the four
// that depend on the column alone (dx, U, -U^2, NR U) have nothing to pack with.  Packed over two RECORDS instead -- the stage
// holds pairs interleaved {x0,x1, y0,y1, IX0,IX1, NR0,NR1 | IY0,IY1, r0,r1, g0,g1, b0,b1} -- every instruction is packed:
// 4 (column) + 2 rows x 6 = 16 packed + 4 v_exp_f32 per record PAIR and 128 pixels = 96 cycles against 2 x 64.  The sums of
// the even and the odd records of a list are kept apart (two accumulators per channel and row) and added at the end; a list
// of odd length ends in a zero record (colour 0).
// Which plans: the dense ones (tl_dense: at least one Gaussian per four pixels -- GSASR's 16 per LR pixel, the reference's published
// workload) -- measured (profiles/history/r05_whatif.txt, r05_lists_ab.txt): 16 Gaussians per LR pixel -5% on the forward, the published
// workload -11%; at one Gaussian per LR pixel (config 2: 40 hits per sub-tile) the 30 extra VGPRs cost two waves per SIMD and
// the forward +6%, so those keep the pixel-packed evaluation.  -DFWD_PAIR=0 / 1 forces one of them everywhere (what-if builds).
constexpr int STAGE_F4 = 136;    // float4 per wave's stage: 64 records + a zero record behind each of the two lists (pairs)

__device__ __forceinline__ void stage_put_pair(float4 *stage, int slot, const float4 a, const float4 b)
{
    float *p = reinterpret_cast<float *>(stage) + (slot >> 1) * 16 + (slot & 1);
    p[0] = a.x; p[2] = a.y; p[4] = a.z; p[6] = a.w; p[8] = b.x; p[10] = b.y; p[12] = b.z; p[14] = b.w;
}

template <bool TEST>
__device__ __forceinline__ void fwd_eval_pair(const float4 q0, const float4 q1, const float4 q2, const float4 q3, float px, v2f py,
                                              float dmax, v2f (&acc)[6])
{
    // q0 = {x0,x1,y0,y1}, q1 = {IX0,IX1,NR0,NR1}, q2 = {IY0,IY1,r0,r1}, q3 = {g0,g1,b0,b1}; acc = {rA, gA, bA, rB, gB, bB} (row A / B)
    const v2f x = {q0.x, q0.y}, y = {q0.z, q0.w}, ix = {q1.x, q1.y}, nr = {q1.z, q1.w}, iy = {q2.x, q2.y};
    const v2f cr = {q2.z, q2.w}, cg = {q3.x, q3.y}, cb = {q3.z, q3.w};
    const v2f dx = px - x;
    const v2f u = ix * dx;
    const v2f k0 = -u * u, ru = nr * u;
    const v2f dyA = py.x - y, dyB = py.y - y;
    const v2f bqA = iy * dyA + ru, bqB = iy * dyB + ru;
    const v2f pwA = k0 - bqA * bqA, pwB = k0 - bqB * bqB;
    v2f vA = {__builtin_amdgcn_exp2f(pwA.x), __builtin_amdgcn_exp2f(pwA.y)};
    v2f vB = {__builtin_amdgcn_exp2f(pwB.x), __builtin_amdgcn_exp2f(pwB.y)};
    if (TEST) {
        const bool in0 = fabsf(dx.x) <= dmax, in1 = fabsf(dx.y) <= dmax;
        vA.x = (in0 && fabsf(dyA.x) <= dmax) ? vA.x : 0.f;
        vA.y = (in1 && fabsf(dyA.y) <= dmax) ? vA.y : 0.f;
        vB.x = (in0 && fabsf(dyB.x) <= dmax) ? vB.x : 0.f;
        vB.y = (in1 && fabsf(dyB.y) <= dmax) ? vB.y : 0.f;
    }
    acc[0] += vA * cr; acc[1] += vA * cg; acc[2] += vA * cb;
    acc[3] += vB * cr; acc[4] += vB * cg; acc[5] += vB * cb;
}

// pairs [beg, end) of the stage
template <bool TEST>
__device__ __forceinline__ void fwd_eval_lds_pairs(const float4 *__restrict__ st, int beg, int end, float px, v2f py, float dmax,
                                                   v2f (&acc)[6])
{
    int i = beg;
    for (; i + 1 < end; i += 2) {   // two pairs per iteration so their dependent chains interleave
        const float4 a0 = st[4 * i], a1 = st[4 * i + 1], a2 = st[4 * i + 2], a3 = st[4 * i + 3];
        const float4 b0 = st[4 * i + 4], b1 = st[4 * i + 5], b2 = st[4 * i + 6], b3 = st[4 * i + 7];
        fwd_eval_pair<TEST>(a0, a1, a2, a3, px, py, dmax, acc);
        fwd_eval_pair<TEST>(b0, b1, b2, b3, px, py, dmax, acc);
    }
    if (i < end) fwd_eval_pair<TEST>(st[4 * i], st[4 * i + 1], st[4 * i + 2], st[4 * i + 3], px, py, dmax, acc);
}

// One chunk of a wave's walk: compact the hits' records (ra, rb of the hit lanes; those that need the dmax test behind the
// others) into the wave's LDS stage and evaluate them on the lane's two pixels from broadcast LDS reads.
template <bool BOUNDED, bool PAIR>
__device__ __forceinline__ void fwd_stage_eval(float4 *stage, bool hit, bool needs, const float4 ra, const float4 rb, int lane, float px,
                                               v2f py, float dmax, v2f &ar, v2f &ag, v2f &ab)
{
    const unsigned long long below = (1ull << lane) - 1ull;
#ifdef FWD_EXP_NOEVAL   // what-if build (tools/whatif.sh): the records are found, fetched and staged, ONE per chunk is evaluated
    const unsigned long long mall = __ballot(hit && !needs), m0 = mall & (0ull - mall), m1 = 0ull;    // (lowest set bit)
    hit = hit && !needs && ((m0 >> lane) & 1ull) != 0ull;
#else
    const unsigned long long m0 = __ballot(hit && !needs), m1 = BOUNDED ? __ballot(hit && needs) : 0ull;
#endif
    const int n0 = __builtin_popcountll(m0), n1 = __builtin_popcountll(m1);
    if (n0 + n1 == 0) return;
    if constexpr (PAIR) {
    const int b1 = (n0 + 1) & ~1;     // first slot of the tested list (the lists are padded to whole pairs)
    __builtin_amdgcn_wave_barrier();
    if (hit) {
        const int r = needs ? __builtin_popcountll(m1 & below) : __builtin_popcountll(m0 & below);
        const int slot = needs ? b1 + r : r;
        stage_put_pair(stage, slot, ra, rb);
        const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
        if (((needs ? n1 : n0) & 1) && r == (needs ? n1 : n0) - 1) stage_put_pair(stage, slot + 1, z, z);   // the odd list's zero record
    }
    __builtin_amdgcn_wave_barrier();
    v2f acc[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) acc[k] = (v2f){0.f, 0.f};
    fwd_eval_lds_pairs<false>(stage, 0, (n0 + 1) >> 1, px, py, dmax, acc);
    if (BOUNDED) fwd_eval_lds_pairs<true>(stage, b1 >> 1, (b1 + n1 + 1) >> 1, px, py, dmax, acc);
    ar += (v2f){acc[0].x + acc[0].y, acc[3].x + acc[3].y};
    ag += (v2f){acc[1].x + acc[1].y, acc[4].x + acc[4].y};
    ab += (v2f){acc[2].x + acc[2].y, acc[5].x + acc[5].y};
    } else {
    __builtin_amdgcn_wave_barrier();
    if (hit) {
        const int slot = needs ? n0 + __builtin_popcountll(m1 & below) : __builtin_popcountll(m0 & below);
        stage[2 * slot] = ra;
        stage[2 * slot + 1] = rb;
    }
    __builtin_amdgcn_wave_barrier();
    fwd_eval_lds<false>(stage, 0, n0, px, py, dmax, ar, ag, ab);
    if (BOUNDED) fwd_eval_lds<true>(stage, n0, n0 + n1, px, py, dmax, ar, ag, ab);
    }
}

// One wave, one 8x16 sub-tile at (sx0, sy0): accumulate every Gaussian binned near it into ar/ag/ab
// (lane = column sx0 + lane%8, rows sy0 + lane/8 and +8).  With nparts > 1 the 64-candidate chunks are
// dealt round-robin to `nparts` waves and the caller adds their partial sums.
// LARGE_ONLY: the walk over the "large" class alone (what a list kernel still has to scan: tile lists hold the normal class).
template <bool BOUNDED, bool PAIR, bool LARGE_ONLY = false>
__device__ __forceinline__ void fwd_tile(const Params &P, const PlanView &V, int sx0, int sy0, int lane,
                                         unsigned part, unsigned nparts, float4 *stage, v2f &ar, v2f &ag, v2f &ab)
{
    const int sx1 = min(sx0 + SUBX - 1, P.w - 1), sy1 = min(sy0 + SUBY - 1, P.row1 - 1);
    const int X = sx0 + (lane & 7), Y0 = sy0 + (lane >> 3), Y1 = Y0 + 8;
    // (batched canvas: slots are whole tile rows, so a sub-tile belongs to one sample; its px table is the sy0/slot-th)
    const float px = V.px[(P.batch > 1 ? (sy0 / P.slot) * P.w : 0) + min(X, P.w - 1)];
    const v2f py = {V.py[min(Y0, P.h - 1)], V.py[min(Y1, P.h - 1)]};

    const float4 *__restrict__ rec = V.rec;
    const uint4 *__restrict__ bbox = V.bbox;
    const unsigned *__restrict__ cs = V.cell_start;
    const int wty = (sy0 - P.row0) >> SUBY_SHIFT, wtx = sx0 >> SUBX_SHIFT;

    // Segment table: lane r holds [beg,end) of cell row cy0+r restricted to the columns a normal-class
    // Gaussian can reach this sub-tile from (max half-extent from the plan header); one more lane holds
    // the large class.  One vector round trip instead of a dependent scalar load per row.
    const int rx = LARGE_ONLY ? 0 : (int)V.hdr[8], ry = LARGE_ONLY ? 0 : (int)V.hdr[9];
    int nseg = 0;
    unsigned sbeg = 0, send = 0;
    if (rx > 0) {
        const int cx0 = max(sx0 - rx, 0) >> CELL_SHIFT, cx1 = min((sx1 + rx) >> CELL_SHIFT, P.ncx - 1);
        const int cy0 = max(sy0 - ry, 0) >> CELL_SHIFT, cy1 = min((sy1 + ry) >> CELL_SHIFT, P.ncy - 1);
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

    // Flat walk over full 64-candidate chunks of the concatenated segments (this wave takes chunks
    // part, part+nparts, ...), software-pipelined: the window record of the NEXT chunk is in flight while
    // the hits of the current one are evaluated.
    const unsigned len = send - sbeg;
    unsigned pin = len;  // inclusive prefix sum of the segment lengths over the lanes
    for (int o = 1; o < 64; o <<= 1) {
        const unsigned v = (unsigned)__shfl_up((int)pin, o);
        if (lane >= o) pin += v;
    }
    const unsigned pex = pin - len;
    const unsigned total = (unsigned)__builtin_amdgcn_readlane((int)pin, nseg - 1);
    const unsigned nchunks = (total + 63u) >> 6;
    const uint4 dead = make_uint4(0x7fffu, 0x7fffu, 0u, 0u);  // c0 = r0 = 32767 > c1 = r1 = 0: overlaps no tile
    int rseg = 0;
    unsigned c = part;
    unsigned j = c < nchunks ? fwd_candidate(c, lane, nseg, rseg, sbeg, pex, pin) : 0xffffffffu;
    uint4 bb = dead;
    uint2 bs = make_uint2(0u, 0u);
    if (j != 0xffffffffu) {
        bb = bbox[2 * (size_t)j];
        bs = *reinterpret_cast<const uint2 *>(bbox + 2 * (size_t)j + 1);
    }
    while (c < nchunks) {
        const unsigned nc = c + nparts;
        const unsigned nj = nc < nchunks ? fwd_candidate(nc, lane, nseg, rseg, sbeg, pex, pin) : 0xffffffffu;
        uint4 nbb = dead;
        uint2 nbs = make_uint2(0u, 0u);
        if (nj != 0xffffffffu) {
            nbb = bbox[2 * (size_t)nj];
            nbs = *reinterpret_cast<const uint2 *>(bbox + 2 * (size_t)nj + 1);
        }

        const int c0 = (int)(bb.x & 0x7fffu), c1 = (int)(bb.x >> 16);
        const int r0 = (int)(bb.y & 0x7fffu), r1 = (int)(bb.y >> 16);
        bool hit = (c0 <= sx1) & (c1 >= sx0) & (r0 <= sy1) & (r1 >= sy0);
        if (bb.y & 0x8000u) {  // per-tile-row column spans (k_bin)
            const unsigned t = (unsigned)(wty - ((r0 - P.row0) >> SUBY_SHIFT)) & 7u, sh = (t & 3u) * 8u;
            const unsigned lo = t < 4u ? bb.z : bs.x, hi = t < 4u ? bb.w : bs.y;
            const int txr = wtx - (c0 >> SUBX_SHIFT);
            hit &= (txr >= (int)((lo >> sh) & 0xffu)) & (txr <= (int)((hi >> sh) & 0xffu));
        }
        const bool needs = BOUNDED && (bb.x & 0x8000u) != 0u;
        // Compact the hits' records into this wave's LDS stage (untested ones first), then every lane
        // evaluates all of them from broadcast LDS reads.
        float4 ra = make_float4(0.f, 0.f, 0.f, 0.f), rb = ra;
        if (hit) {
            const float4 *src = rec + 2 * (size_t)j;
            ra = src[0];
            rb = src[1];
        }
        fwd_stage_eval<BOUNDED, PAIR>(stage, hit, needs, ra, rb, lane, px, py, P.dmax, ar, ag, ab);
        c = nc; j = nj; bb = nbb; bs = nbs;
    }
}

// ---- two-level walk (large images) ---------------------------------------------------------------------
// Measured: finding a sub-tile's hits -- fetching the 24-byte windows of every Gaussian binned within reach and
// testing them, 64 per wave and round trip -- is HALF of the forward's time (the candidates come from cells within
// the class' MAXIMUM extent, 4-5x more than hit).  The four waves of a workgroup render four sub-tiles side by side,
// and their candidate sets are almost the same, so the workgroup walks the candidates of its 32x16 tile ONCE,
// cooperatively: each wave tests a quarter of the chunks against the whole tile (window only, 8 bytes per
// candidate) and appends the survivors to a shared list in LDS; after a barrier every wave runs the full test
// (window + ellipse span) over that list only.  Rounds of 1024 candidates bound the list.
constexpr int COARSE_CHUNKS = 4;                          // coarse chunks per wave and round
constexpr int COARSE_LIST = 4 * COARSE_CHUNKS * 64;       // candidates per round and part = capacity of the shared list

// PARTS = 2: eight waves per workgroup, two per sub-tile taking alternate chunks of the survivor list (images with
// fewer sub-tiles than the chip has wave slots); the caller adds the two partial sums.
template <bool BOUNDED, int PARTS, bool PAIR>
__device__ __forceinline__ void fwd_block(const Params &P, const PlanView &V, int bx0, int by0, int wv, int lane,
                                          float4 *stage, unsigned *s_list, unsigned *s_cnt, v2f &ar, v2f &ag, v2f &ab)
{
    const int bx1 = min(bx0 + 4 * SUBX - 1, P.w - 1), by1 = min(by0 + SUBY - 1, P.row1 - 1);
    const int sx0 = bx0 + (wv & 3) * SUBX;
    const unsigned part = (unsigned)(wv >> 2);
    const bool live = sx0 < P.w;                              // wave-uniform (image width not a multiple of 32)
    const int sx1 = min(sx0 + SUBX - 1, P.w - 1);
    const int X = sx0 + (lane & 7);
    const float px = V.px[(P.batch > 1 ? (by0 / P.slot) * P.w : 0) + min(X, P.w - 1)];
    const int Y0 = by0 + (lane >> 3);
    const v2f py = {V.py[min(Y0, P.h - 1)], V.py[min(Y0 + 8, P.h - 1)]};
    const float4 *__restrict__ rec = V.rec;
    const uint4 *__restrict__ bbox = V.bbox;
    const unsigned *__restrict__ cs = V.cell_start;
    const int wtx = sx0 >> SUBX_SHIFT;

    // segment table of the 32x16 tile (every wave builds the same one: a single vector round trip)
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
    const unsigned nchunks = (total + 63u) >> 6;
    const unsigned long long below = (1ull << lane) - 1ull;
    int rseg = 0;

    for (unsigned base = 0, round = 0; base < nchunks; base += 4u * PARTS * COARSE_CHUNKS, ++round) {
        unsigned *cnt = s_cnt + (round & 1u);
        // ---- phase A: this wave's share of the round's chunks against the whole tile -------------------
        unsigned cj[COARSE_CHUNKS];
        uint2 cw[COARSE_CHUNKS];
#pragma unroll
        for (int k = 0; k < COARSE_CHUNKS; ++k) {     // all loads of the round in flight together
            const unsigned c = base + (unsigned)wv + 4u * PARTS * (unsigned)k;
            cj[k] = c < nchunks ? fwd_candidate(c, lane, nseg, rseg, sbeg, pex, pin) : 0xffffffffu;
            cw[k] = make_uint2(0x7fffu, 0x7fffu);
            if (cj[k] != 0xffffffffu) cw[k] = V.win[cj[k]];
        }
#pragma unroll
        for (int k = 0; k < COARSE_CHUNKS; ++k) {
            const int c0 = (int)(cw[k].x & 0x7fffu), c1 = (int)(cw[k].x >> 16);
            const int r0 = (int)(cw[k].y & 0x7fffu), r1 = (int)(cw[k].y >> 16);
            const bool hit = (c0 <= bx1) & (c1 >= bx0) & (r0 <= by1) & (r1 >= by0);
            const unsigned long long m = __ballot(hit);
            if (m) {
                unsigned at = 0;
                if (lane == 0) at = atomicAdd(cnt, (unsigned)__builtin_popcountll(m));
                at = (unsigned)__builtin_amdgcn_readfirstlane((int)at);
                if (hit) s_list[at + (unsigned)__builtin_popcountll(m & below)] = cj[k];
            }
        }
        __syncthreads();
        if (threadIdx.x == 0) s_cnt[(round + 1u) & 1u] = 0u;   // nobody touches the other counter before the next barrier
        const unsigned n = (unsigned)__builtin_amdgcn_readfirstlane((int)*cnt);
        // ---- phase B: the full test of the tile's survivors against this wave's sub-tile(s) -------------
        const int sy0 = by0, sy1 = by1;
        const int wty = (sy0 - P.row0) >> SUBY_SHIFT;
        if (live) {
            const unsigned q0 = part * 64u;
            unsigned j = q0 + lane < n ? s_list[q0 + lane] : 0xffffffffu;
            const uint4 dead = make_uint4(0x7fffu, 0x7fffu, 0u, 0u);
            uint4 bb = dead;
            uint2 bs = make_uint2(0u, 0u);
            if (j != 0xffffffffu) {
                bb = bbox[2 * (size_t)j];
                bs = *reinterpret_cast<const uint2 *>(bbox + 2 * (size_t)j + 1);
            }
            for (unsigned q = q0; q < n; q += 64u * PARTS) {
                const unsigned nq = q + 64u * PARTS + (unsigned)lane;
                const unsigned nj = nq < n ? s_list[nq] : 0xffffffffu;
                uint4 nbb = dead;
                uint2 nbs = make_uint2(0u, 0u);
                if (nj != 0xffffffffu) {
                    nbb = bbox[2 * (size_t)nj];
                    nbs = *reinterpret_cast<const uint2 *>(bbox + 2 * (size_t)nj + 1);
                }
                const int c0 = (int)(bb.x & 0x7fffu), c1 = (int)(bb.x >> 16);
                const int r0 = (int)(bb.y & 0x7fffu), r1 = (int)(bb.y >> 16);
                bool hit = (c0 <= sx1) & (c1 >= sx0) & (r0 <= sy1) & (r1 >= sy0);
                if (bb.y & 0x8000u) {  // per-tile-row column spans (k_bin)
                    const unsigned t = (unsigned)(wty - ((r0 - P.row0) >> SUBY_SHIFT)) & 7u, sh = (t & 3u) * 8u;
                    const unsigned lo = t < 4u ? bb.z : bs.x, hi = t < 4u ? bb.w : bs.y;
                    const int txr = wtx - (c0 >> SUBX_SHIFT);
                    hit &= (txr >= (int)((lo >> sh) & 0xffu)) & (txr <= (int)((hi >> sh) & 0xffu));
                }
                const bool needs = BOUNDED && (bb.x & 0x8000u) != 0u;
                float4 ra = make_float4(0.f, 0.f, 0.f, 0.f), rb = ra;
                if (hit) {
                    const float4 *src = rec + 2 * (size_t)j;
                    ra = src[0];
                    rb = src[1];
                }
                fwd_stage_eval<BOUNDED, PAIR>(stage, hit, needs, ra, rb, lane, px, py, P.dmax, ar, ag, ab);
                j = nj; bb = nbb; bs = nbs;
            }
        }
        __syncthreads();   // the list is rewritten in the next round
    }
}

__device__ __forceinline__ void fwd_store(const Params &P, const PlanView &V, float *__restrict__ img, int sx0, int sy0,
                                          int lane, v2f ar, v2f ag, v2f ab)
{
    const int X = sx0 + (lane & 7), Y0 = sy0 + (lane >> 3), Y1 = Y0 + 8;
    if (X >= P.w) return;
#ifdef FWD_EXP_NOSTORE   // what-if build (tools/whatif.sh): everything but the image store (one lane in 2^20 keeps the sums alive)
    if ((ar.x + ag.x + ab.x + ar.y + ag.y + ab.y) != 123456.789f) return;
#endif
    const bool store = P.flags & GSASR_FLAG_OVERWRITE_IMAGE;
    bool ok0 = Y0 < P.row1, ok1 = Y1 < P.row1;
    // CHW: planar [3, rows, w]; batched canvas: [B, 3, slot, w] (HWC is simply the canvas [B*slot, w, 3])
    size_t plane = (size_t)(P.row1 - P.row0) * P.w, chw0 = (size_t)(Y0 - P.row0) * P.w + X, chw1 = chw0 + 8 * (size_t)P.w;
    if (P.batch > 1) {
        // pixels of the slot outside the sample's own h_b x w_b grid are padding: stored as zero, never added to
        const int smp = sy0 / P.slot;
        const Geo g = sample_geo(P, V, smp);
        const bool inx = X < g.w, in0 = inx && Y0 - g.base < g.h, in1 = inx && Y1 - g.base < g.h;
        if (!in0) { ar.x = ag.x = ab.x = 0.f; ok0 = ok0 && store; }
        if (!in1) { ar.y = ag.y = ab.y = 0.f; ok1 = ok1 && store; }
        plane = (size_t)P.slot * P.w;
        chw0 = ((size_t)smp * 3 * P.slot + (size_t)(Y0 - g.base)) * P.w + X;
        chw1 = chw0 + 8 * (size_t)P.w;
    }
    if (P.flags & GSASR_FLAG_CHW_IMAGE) {
        if (ok0) {
            float *o = img + chw0;
            if (store) { o[0] = ar.x; o[plane] = ag.x; o[2 * plane] = ab.x; }
            else { o[0] += ar.x; o[plane] += ag.x; o[2 * plane] += ab.x; }
        }
        if (ok1) {
            float *o = img + chw1;
            if (store) { o[0] = ar.y; o[plane] = ag.y; o[2 * plane] = ab.y; }
            else { o[0] += ar.y; o[plane] += ag.y; o[2 * plane] += ab.y; }
        }
        return;
    }
    if (ok0) {
        float *o = img + ((size_t)(Y0 - P.row0) * P.w + X) * 3;
        if (store) { o[0] = ar.x; o[1] = ag.x; o[2] = ab.x; }
        else { o[0] += ar.x; o[1] += ag.x; o[2] += ab.x; }
    }
    if (ok1) {
        float *o = img + ((size_t)(Y1 - P.row0) * P.w + X) * 3;
        if (store) { o[0] = ar.y; o[1] = ag.y; o[2] = ab.y; }
        else { o[0] += ar.y; o[1] += ag.y; o[2] += ab.y; }
    }
}

// Two-level walk (fwd_block).  PARTS = 1: large images, the workgroup shape of k_render_fwd.  PARTS = 2: images
// with fewer sub-tiles than wave slots -- eight waves, two per sub-tile, partial sums combined through LDS.
template <bool BOUNDED, int PARTS, bool PAIR>
__global__ __launch_bounds__(256 * PARTS) void k_render_fwd2(Params P, PlanView V, float *__restrict__ img, int tiles_x)
{
    const unsigned t = xcd_swizzle(blockIdx.x, gridDim.x);
    const int bx = (int)(t % (unsigned)tiles_x), by = (int)(t / (unsigned)tiles_x);
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    __shared__ float4 s_stage[4 * PARTS][STAGE_F4];
    __shared__ unsigned s_list[COARSE_LIST * PARTS];
    __shared__ unsigned s_cnt[2];
    __shared__ float s_part[PARTS > 1 ? 4 : 1][6][64];
    if (threadIdx.x < 2) s_cnt[threadIdx.x] = 0u;
    __syncthreads();
    v2f ar = {0.f, 0.f}, ag = {0.f, 0.f}, ab = {0.f, 0.f};
    const int bx0 = bx * 4 * SUBX, by0 = P.row0 + by * SUBY;
    fwd_block<BOUNDED, PARTS, PAIR>(P, V, bx0, by0, wv, lane, s_stage[wv], s_list, s_cnt, ar, ag, ab);
    const int sub = wv & 3;
    if (PARTS > 1) {   // (fwd_block ends on a barrier)
        if (wv >= 4) {
            float (*o)[64] = s_part[sub];
            o[0][lane] = ar.x; o[1][lane] = ar.y; o[2][lane] = ag.x; o[3][lane] = ag.y; o[4][lane] = ab.x; o[5][lane] = ab.y;
        }
        __syncthreads();
        if (wv >= 4) return;
        float (*o)[64] = s_part[sub];
        ar.x += o[0][lane]; ar.y += o[1][lane]; ag.x += o[2][lane]; ag.y += o[3][lane]; ab.x += o[4][lane]; ab.y += o[5][lane];
    }
    if (bx0 + sub * SUBX < P.w) fwd_store(P, V, img, bx0 + sub * SUBX, by0, lane, ar, ag, ab);
}

// WIDE forward (large windows: x8 and up, single images).  A wave owns a 16 x 16 sub-tile, lane = column sx0 + lane%16 and
// the four rows sy0 + lane/16 + {0, 4 | 8, 12}: two packed row pairs in ONE column, so a record's column arithmetic is done
// once for four pixels (ten instructions + four v_exp_f32 per record and 256 pixels, against two times six + two for the
// 8 x 16 sub-tile), and half as many waves search.  The workgroup's tile is 32 x 32 (2 x 2 sub-tiles), walked like
// fwd_block: cooperative window test of the candidates against the tile, then every wave tests the survivors against its
// own sub-tile.  Against this stands the coarser cull ((w + 16)(h + 16) instead of (w + 8)(h + 16) evaluated pixels per
// window): it pays from ~40-px windows up.
constexpr int WIDE = 16;   // sub-tile side
#ifndef FWD8_DEFAULT
#define FWD8_DEFAULT 0     // 1: the fine forward (k_render_fwd8) is the default for sparse single images
#endif

__device__ __forceinline__ void fwd_store_px(const Params &P, float *__restrict__ img, int X, int Y, float r, float g, float b)
{
    if (Y >= P.row1) return;
    const bool store = P.flags & GSASR_FLAG_OVERWRITE_IMAGE;
    if (P.flags & GSASR_FLAG_CHW_IMAGE) {
        const size_t plane = (size_t)(P.row1 - P.row0) * P.w;
        float *o = img + (size_t)(Y - P.row0) * P.w + X;
        if (store) { o[0] = r; o[plane] = g; o[2 * plane] = b; }
        else { o[0] += r; o[plane] += g; o[2 * plane] += b; }
        return;
    }
    float *o = img + ((size_t)(Y - P.row0) * P.w + X) * 3;
    if (store) { o[0] = r; o[1] = g; o[2] = b; }
    else { o[0] += r; o[1] += g; o[2] += b; }
}

// FINE forward (round 6): the two-level walk of fwd_block with EIGHT waves per 32 x 16-px tile, each an 8 x 8-px sub-tile, ONE
// pixel per lane, the evaluation packed over record PAIRS.  Why: the forward is bound by VALU issue and at x4 its instructions
// are evaluations -- a ~20-px window is evaluated on every 8 x 16 sub-tile it touches, 57 hits per sub-tile = 57 evaluations per
// pixel where the windows hold 31.  On 8 x 8 sub-tiles a pixel sees 41 (-28%), and with one pixel per lane the packing goes
// over two records, whose column arithmetic packs as well: 10 packed + 2 v_exp_f32 per record pair and 64 pixels = 14 issue
// slots per 128 pairs against 16.  Against this stand twice the waves running level 2 over the tile's survivors.  The pair
// stage holds 6 accumulator registers here (the 8 x 16 pair kernel: 12, and 86 VGPRs), so eight waves per SIMD stay.
// Single images; sparse plans (the dense ones render from lists).
// MEASURED (profiles/r06_fwd8.txt) and NOT the default: parity-green (172 GPU tests under GSASR_SPLAT_FWD8=1) and SLOWER --
// config 2 forward 29.4 against 25.1 us, 2048^2 x4 98.6 against 74.4, x2 level.  The evaluation became LDS-bound: a record pair
// is four broadcast ds_read_b128 (16 LDS cycles, one LDS per CU) for 14 issue slots = 56 cycles on each of four SIMDs -- 64 LDS
// cycles per 56; the 8 x 16 kernels read two per record for 64 cycles (50%).  Every record has to reach all 64 lanes of every
// wave that evaluates it, so a sub-tile of half the pixels doubles the LDS reads per evaluated pixel; the instruction model
// that promised -21% did not count them.  Kept behind the development switch.
template <bool TEST>
__device__ __forceinline__ void fwd_eval_pair1(const float4 q0, const float4 q1, const float4 q2, const float4 q3, float px, float py,
                                               float dmax, v2f (&acc)[3])
{
    // q0 = {x0,x1,y0,y1}, q1 = {IX0,IX1,NR0,NR1}, q2 = {IY0,IY1,r0,r1}, q3 = {g0,g1,b0,b1} (stage_put_pair)
    const v2f x = {q0.x, q0.y}, y = {q0.z, q0.w}, ix = {q1.x, q1.y}, nr = {q1.z, q1.w}, iy = {q2.x, q2.y};
    const v2f cr = {q2.z, q2.w}, cg = {q3.x, q3.y}, cb = {q3.z, q3.w};
    const v2f dx = px - x, dy = py - y;
    const v2f u = ix * dx;
    const v2f k0 = -u * u, ru = nr * u;
    const v2f bq = iy * dy + ru;
    const v2f pw = k0 - bq * bq;
    v2f v = {__builtin_amdgcn_exp2f(pw.x), __builtin_amdgcn_exp2f(pw.y)};
    if (TEST) {
        v.x = (fabsf(dx.x) <= dmax && fabsf(dy.x) <= dmax) ? v.x : 0.f;
        v.y = (fabsf(dx.y) <= dmax && fabsf(dy.y) <= dmax) ? v.y : 0.f;
    }
    acc[0] += v * cr;
    acc[1] += v * cg;
    acc[2] += v * cb;
}

template <bool TEST>
__device__ __forceinline__ void fwd_eval_lds_pairs1(const float4 *__restrict__ st, int beg, int end, float px, float py, float dmax,
                                                    v2f (&acc)[3])
{
    int i = beg;
    for (; i + 1 < end; i += 2) {   // two pairs per iteration so their dependent chains interleave
        const float4 a0 = st[4 * i], a1 = st[4 * i + 1], a2 = st[4 * i + 2], a3 = st[4 * i + 3];
        const float4 b0 = st[4 * i + 4], b1 = st[4 * i + 5], b2 = st[4 * i + 6], b3 = st[4 * i + 7];
        fwd_eval_pair1<TEST>(a0, a1, a2, a3, px, py, dmax, acc);
        fwd_eval_pair1<TEST>(b0, b1, b2, b3, px, py, dmax, acc);
    }
    if (i < end) fwd_eval_pair1<TEST>(st[4 * i], st[4 * i + 1], st[4 * i + 2], st[4 * i + 3], px, py, dmax, acc);
}

// one chunk of a wave's walk (cf. fwd_stage_eval, PAIR): the hits' records go to the wave's stage as interleaved pairs, the
// tested ones behind the others, each list padded to whole pairs with a zero record
template <bool BOUNDED>
__device__ __forceinline__ void fwd_stage_eval1(float4 *stage, bool hit, bool needs, const float4 ra, const float4 rb, int lane, float px,
                                                float py, float dmax, v2f (&acc)[3])
{
    const unsigned long long below = (1ull << lane) - 1ull;
    const unsigned long long m0 = __ballot(hit && !needs), m1 = BOUNDED ? __ballot(hit && needs) : 0ull;
    const int n0 = __builtin_popcountll(m0), n1 = __builtin_popcountll(m1);
    if (n0 + n1 == 0) return;
    const int b1 = (n0 + 1) & ~1;
    __builtin_amdgcn_wave_barrier();
    if (hit) {
        const int r = needs ? __builtin_popcountll(m1 & below) : __builtin_popcountll(m0 & below);
        const int slot = needs ? b1 + r : r;
        stage_put_pair(stage, slot, ra, rb);
        const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
        if (((needs ? n1 : n0) & 1) && r == (needs ? n1 : n0) - 1) stage_put_pair(stage, slot + 1, z, z);
    }
    __builtin_amdgcn_wave_barrier();
    fwd_eval_lds_pairs1<false>(stage, 0, (n0 + 1) >> 1, px, py, dmax, acc);
    if (BOUNDED) fwd_eval_lds_pairs1<true>(stage, b1 >> 1, (b1 + n1 + 1) >> 1, px, py, dmax, acc);
}

template <bool BOUNDED>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_render_fwd8(Params P, PlanView V, float *__restrict__ img, int tiles_x)
{
    constexpr int FW = 8;                         // waves per tile
    const unsigned t = xcd_swizzle(blockIdx.x, gridDim.x);
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    __shared__ float4 s_stage[FW][STAGE_F4];
    __shared__ unsigned s_list[2 * COARSE_LIST];
    __shared__ unsigned s_cnt[2];
    if (threadIdx.x < 2) s_cnt[threadIdx.x] = 0u;
    __syncthreads();
    const int bx0 = (int)(t % (unsigned)tiles_x) * 4 * SUBX, by0 = P.row0 + (int)(t / (unsigned)tiles_x) * SUBY;
    const int bx1 = min(bx0 + 4 * SUBX - 1, P.w - 1), by1 = min(by0 + SUBY - 1, P.row1 - 1);
    const int sx0 = bx0 + (wv & 3) * SUBX, sy0 = by0 + (wv >> 2) * 8;
    const bool live = sx0 < P.w && sy0 < P.row1;                  // wave-uniform
    const int sx1 = min(sx0 + SUBX - 1, P.w - 1), sy1 = min(sy0 + 7, P.row1 - 1);
    const int X = sx0 + (lane & 7), Y = sy0 + (lane >> 3);
    const float px = V.px[min(X, P.w - 1)], py = V.py[min(Y, P.h - 1)];
    float4 *stage = s_stage[wv];
    const float4 *__restrict__ rec = V.rec;
    const uint4 *__restrict__ bbox = V.bbox;
    const unsigned *__restrict__ cs = V.cell_start;
    const int wtx = sx0 >> SUBX_SHIFT, wty = (by0 - P.row0) >> SUBY_SHIFT;     // in the units of k_bin's spans (8 columns, 16 rows)

    // segment table of the 32x16 tile (every wave builds the same one; cf. fwd_block)
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
    const unsigned nchunks = (total + 63u) >> 6;
    const unsigned long long below = (1ull << lane) - 1ull;
    int rseg = 0;
    v2f acc[3] = {(v2f){0.f, 0.f}, (v2f){0.f, 0.f}, (v2f){0.f, 0.f}};

    for (unsigned base = 0, round = 0; base < nchunks; base += (unsigned)(FW * COARSE_CHUNKS), ++round) {
        unsigned *cnt = s_cnt + (round & 1u);
        // ---- phase A: this wave's share of the round's chunks against the whole tile (window only, 8 bytes per candidate) ----
        unsigned cj[COARSE_CHUNKS];
        uint2 cw[COARSE_CHUNKS];
#pragma unroll
        for (int k = 0; k < COARSE_CHUNKS; ++k) {
            const unsigned c = base + (unsigned)wv + (unsigned)(FW * k);
            cj[k] = c < nchunks ? fwd_candidate(c, lane, nseg, rseg, sbeg, pex, pin) : 0xffffffffu;
            cw[k] = make_uint2(0x7fffu, 0x7fffu);
            if (cj[k] != 0xffffffffu) cw[k] = V.win[cj[k]];
        }
#pragma unroll
        for (int k = 0; k < COARSE_CHUNKS; ++k) {
            const int c0 = (int)(cw[k].x & 0x7fffu), c1 = (int)(cw[k].x >> 16);
            const int r0 = (int)(cw[k].y & 0x7fffu), r1 = (int)(cw[k].y >> 16);
            const bool hit = (c0 <= bx1) & (c1 >= bx0) & (r0 <= by1) & (r1 >= by0);
            const unsigned long long m = __ballot(hit);
            if (m) {
                unsigned at = 0;
                if (lane == 0) at = atomicAdd(cnt, (unsigned)__builtin_popcountll(m));
                at = (unsigned)__builtin_amdgcn_readfirstlane((int)at);
                if (hit) s_list[at + (unsigned)__builtin_popcountll(m & below)] = cj[k];
            }
        }
        __syncthreads();
        if (threadIdx.x == 0) s_cnt[(round + 1u) & 1u] = 0u;   // nobody touches the other counter before the next barrier
        const unsigned n = (unsigned)__builtin_amdgcn_readfirstlane((int)*cnt);
        // ---- phase B: the full test of the tile's survivors against this wave's 8 x 8 sub-tile ----------------------
        if (live) {
            unsigned j = (unsigned)lane < n ? s_list[lane] : 0xffffffffu;
            const uint4 dead = make_uint4(0x7fffu, 0x7fffu, 0u, 0u);
            uint4 bb = dead;
            uint2 bs = make_uint2(0u, 0u);
            if (j != 0xffffffffu) {
                bb = bbox[2 * (size_t)j];
                bs = *reinterpret_cast<const uint2 *>(bbox + 2 * (size_t)j + 1);
            }
            for (unsigned q = 0; q < n; q += 64u) {
                const unsigned nq = q + 64u + (unsigned)lane;
                const unsigned nj = nq < n ? s_list[nq] : 0xffffffffu;
                uint4 nbb = dead;
                uint2 nbs = make_uint2(0u, 0u);
                if (nj != 0xffffffffu) {
                    nbb = bbox[2 * (size_t)nj];
                    nbs = *reinterpret_cast<const uint2 *>(bbox + 2 * (size_t)nj + 1);
                }
                const int c0 = (int)(bb.x & 0x7fffu), c1 = (int)(bb.x >> 16);
                const int r0 = (int)(bb.y & 0x7fffu), r1 = (int)(bb.y >> 16);
                bool hit = (c0 <= sx1) & (c1 >= sx0) & (r0 <= sy1) & (r1 >= sy0);
                if (bb.y & 0x8000u) {  // per-16-row-band column spans (k_bin)
                    const unsigned tb = (unsigned)(wty - ((r0 - P.row0) >> SUBY_SHIFT)) & 7u, sh = (tb & 3u) * 8u;
                    const unsigned lo = tb < 4u ? bb.z : bs.x, hi = tb < 4u ? bb.w : bs.y;
                    const int txr = wtx - (c0 >> SUBX_SHIFT);
                    hit &= (txr >= (int)((lo >> sh) & 0xffu)) & (txr <= (int)((hi >> sh) & 0xffu));
                }
                const bool needs = BOUNDED && (bb.x & 0x8000u) != 0u;
                float4 ra = make_float4(0.f, 0.f, 0.f, 0.f), rb = ra;
                if (hit) {
                    const float4 *src = rec + 2 * (size_t)j;
                    ra = src[0];
                    rb = src[1];
                }
                fwd_stage_eval1<BOUNDED>(stage, hit, needs, ra, rb, lane, px, py, P.dmax, acc);
                j = nj; bb = nbb; bs = nbs;
            }
        }
        __syncthreads();   // the list is rewritten in the next round
    }
    if (live && X < P.w) fwd_store_px(P, img, X, Y, acc[0].x + acc[0].y, acc[1].x + acc[1].y, acc[2].x + acc[2].y);
}

template <bool BOUNDED>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_render_fwd16(Params P, PlanView V, float *__restrict__ img, int tiles_x)
{
    const unsigned tt = xcd_swizzle(blockIdx.x, gridDim.x);
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    __shared__ float4 s_stage[4][128];
    __shared__ unsigned s_list[COARSE_LIST];
    __shared__ unsigned s_cnt[2];
    if (threadIdx.x < 2) s_cnt[threadIdx.x] = 0u;
    __syncthreads();
    const int bx0 = (int)(tt % (unsigned)tiles_x) * 2 * WIDE, by0 = P.row0 + (int)(tt / (unsigned)tiles_x) * 2 * WIDE;
    const int bx1 = min(bx0 + 2 * WIDE - 1, P.w - 1), by1 = min(by0 + 2 * WIDE - 1, P.row1 - 1);
    const int sx0 = bx0 + (wv & 1) * WIDE, sy0 = by0 + (wv >> 1) * WIDE;
    const bool live = sx0 < P.w && sy0 < P.row1;                  // wave-uniform
    const int sx1 = min(sx0 + WIDE - 1, P.w - 1), sy1 = min(sy0 + WIDE - 1, P.row1 - 1);
    const int X = sx0 + (lane & 15), Y = sy0 + (lane >> 4);
    const float px = V.px[min(X, P.w - 1)];
    const v2f pyA = {V.py[min(Y, P.h - 1)], V.py[min(Y + 4, P.h - 1)]};
    const v2f pyB = {V.py[min(Y + 8, P.h - 1)], V.py[min(Y + 12, P.h - 1)]};
    float4 *stage = s_stage[wv];
    const float4 *__restrict__ rec = V.rec;
    const uint4 *__restrict__ bbox = V.bbox;
    const unsigned *__restrict__ cs = V.cell_start;
    const int wtx = sx0 >> SUBX_SHIFT, wty = (sy0 - P.row0) >> SUBY_SHIFT;   // in the units of k_bin's spans (8 columns, 16 rows)

    // segment table of the 32x32 tile (every wave builds the same one; cf. fwd_block)
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
    const unsigned nchunks = (total + 63u) >> 6;
    const unsigned long long below = (1ull << lane) - 1ull;
    int rseg = 0;
    v2f acc[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) acc[k] = (v2f){0.f, 0.f};

    for (unsigned base = 0, round = 0; base < nchunks; base += 4u * COARSE_CHUNKS, ++round) {
        unsigned *cnt = s_cnt + (round & 1u);
        // ---- phase A: this wave's share of the round's chunks against the whole tile -------------------
        unsigned cj[COARSE_CHUNKS];
        uint2 cw[COARSE_CHUNKS];
#pragma unroll
        for (int k = 0; k < COARSE_CHUNKS; ++k) {
            const unsigned c = base + (unsigned)wv + 4u * (unsigned)k;
            cj[k] = c < nchunks ? fwd_candidate(c, lane, nseg, rseg, sbeg, pex, pin) : 0xffffffffu;
            cw[k] = make_uint2(0x7fffu, 0x7fffu);
            if (cj[k] != 0xffffffffu) cw[k] = V.win[cj[k]];
        }
#pragma unroll
        for (int k = 0; k < COARSE_CHUNKS; ++k) {
            const int c0 = (int)(cw[k].x & 0x7fffu), c1 = (int)(cw[k].x >> 16);
            const int r0 = (int)(cw[k].y & 0x7fffu), r1 = (int)(cw[k].y >> 16);
            const bool hit = (c0 <= bx1) & (c1 >= bx0) & (r0 <= by1) & (r1 >= by0);
            const unsigned long long m = __ballot(hit);
            if (m) {
                unsigned at = 0;
                if (lane == 0) at = atomicAdd(cnt, (unsigned)__builtin_popcountll(m));
                at = (unsigned)__builtin_amdgcn_readfirstlane((int)at);
                if (hit) s_list[at + (unsigned)__builtin_popcountll(m & below)] = cj[k];
            }
        }
        __syncthreads();
        if (threadIdx.x == 0) s_cnt[(round + 1u) & 1u] = 0u;
        const unsigned n = (unsigned)__builtin_amdgcn_readfirstlane((int)*cnt);
        // ---- phase B: the full test of the tile's survivors against this wave's sub-tile ------------------
        if (live) {
            unsigned j = (unsigned)lane < n ? s_list[lane] : 0xffffffffu;
            const uint4 dead = make_uint4(0x7fffu, 0x7fffu, 0u, 0u);
            uint4 bb = dead;
            uint2 bs = make_uint2(0u, 0u);
            if (j != 0xffffffffu) {
                bb = bbox[2 * (size_t)j];
                bs = *reinterpret_cast<const uint2 *>(bbox + 2 * (size_t)j + 1);
            }
            for (unsigned q = 0; q < n; q += 64u) {
                const unsigned nq = q + 64u + (unsigned)lane;
                const unsigned nj = nq < n ? s_list[nq] : 0xffffffffu;
                uint4 nbb = dead;
                uint2 nbs = make_uint2(0u, 0u);
                if (nj != 0xffffffffu) {
                    nbb = bbox[2 * (size_t)nj];
                    nbs = *reinterpret_cast<const uint2 *>(bbox + 2 * (size_t)nj + 1);
                }
                const int c0 = (int)(bb.x & 0x7fffu), c1 = (int)(bb.x >> 16);
                const int r0 = (int)(bb.y & 0x7fffu), r1 = (int)(bb.y >> 16);
                bool hit = (c0 <= sx1) & (c1 >= sx0) & (r0 <= sy1) & (r1 >= sy0);
                if (bb.y & 0x8000u) {  // per-tile-row column spans (k_bin), in 8-column units: the sub-tile covers two
                    const unsigned t = (unsigned)(wty - ((r0 - P.row0) >> SUBY_SHIFT)) & 7u, sh = (t & 3u) * 8u;
                    const unsigned lo = t < 4u ? bb.z : bs.x, hi = t < 4u ? bb.w : bs.y;
                    const int txr = wtx - (c0 >> SUBX_SHIFT);
                    hit &= (txr + 1 >= (int)((lo >> sh) & 0xffu)) & (txr <= (int)((hi >> sh) & 0xffu));
                }
                const bool needs = BOUNDED && (bb.x & 0x8000u) != 0u;
                // Sort the hits by the row pairs their window reaches: both, only the upper eight rows of the sub-tile (pair A),
                // only the lower eight (pair B); windows cut by the dmax box (exact in-kernel test) stay one list on both.
                const int cls = !hit ? 4 : needs ? 3 : r1 < sy0 + 8 ? 1 : r0 >= sy0 + 8 ? 2 : 0;
                const unsigned long long m0 = __ballot(cls == 0), m1 = __ballot(cls == 1), m2 = __ballot(cls == 2);
                const unsigned long long m3 = BOUNDED ? __ballot(cls == 3) : 0ull;
                const int n0 = __builtin_popcountll(m0), n1 = __builtin_popcountll(m1), n2 = __builtin_popcountll(m2);
                const int n3 = __builtin_popcountll(m3);
                if (n0 + n1 + n2 + n3) {
                    __builtin_amdgcn_wave_barrier();
                    if (hit) {
                        const unsigned long long mine = cls == 0 ? m0 : cls == 1 ? m1 : cls == 2 ? m2 : m3;
                        const int base = cls == 0 ? 0 : cls == 1 ? n0 : cls == 2 ? n0 + n1 : n0 + n1 + n2;
                        const int slot = base + __builtin_popcountll(mine & below);
                        const float4 *src = rec + 2 * (size_t)j;
                        stage[2 * slot] = src[0];
                        stage[2 * slot + 1] = src[1];
                    }
                    __builtin_amdgcn_wave_barrier();
                    fwd_eval_lds16<false, 3>(stage, 0, n0, px, pyA, pyB, P.dmax, acc);
                    fwd_eval_lds16<false, 1>(stage, n0, n0 + n1, px, pyA, pyB, P.dmax, acc);
                    fwd_eval_lds16<false, 2>(stage, n0 + n1, n0 + n1 + n2, px, pyA, pyB, P.dmax, acc);
                    if (BOUNDED) fwd_eval_lds16<true, 3>(stage, n0 + n1 + n2, n0 + n1 + n2 + n3, px, pyA, pyB, P.dmax, acc);
                }
                j = nj; bb = nbb; bs = nbs;
            }
        }
        __syncthreads();   // the list is rewritten in the next round
    }
    if (live && X < P.w) {
        fwd_store_px(P, img, X, Y, acc[0].x, acc[1].x, acc[2].x);
        fwd_store_px(P, img, X, Y + 4, acc[0].y, acc[1].y, acc[2].y);
        fwd_store_px(P, img, X, Y + 8, acc[3].x, acc[4].x, acc[5].x);
        fwd_store_px(P, img, X, Y + 12, acc[3].y, acc[4].y, acc[5].y);
    }
}

// ---- forward from the plan's tile lists (round 5) -------------------------------------------------------------
// The tile's hit list was written by k_bin (tl_emit): nothing is searched and nothing is tested but a mask bit.  A wave streams
// the tile's entries 64 at a time, keeps those whose quadrant mask meets its own sub-tile, gathers their 32-byte records into
// its LDS stage and evaluates them as the search kernels do (same fwd_eval_lds: same sums in another order).  The records of
// chunk k+1 are in flight while chunk k is evaluated.  No barriers, no shared lists: the four waves of a workgroup only share
// the tile.  A tile whose list overflowed its capacity is rendered by the one-level search (fwd_tile); the "large" class is
// scanned by every tile as before.
// PER_SUB (round 6): one workgroup per SUB-TILE (64 x PARTS threads) instead of per tile.  The waves of a tile never shared more
// than its list, and a wave's work is the unit the chip's 1024 SIMDs are dealt: 1152 tile workgroups of four (eight) waves put
// five on some CUs and four on the others (config 5's canvas: the forward took as long as the fives), 4608 (x PARTS) sub-tile
// workgroups even out (profiles/r06_fwd_persub.txt).  The four sub-tiles of a tile stay on one XCD (consecutive units of its band).
template <bool BOUNDED, int PARTS, bool PAIR, bool PER_SUB = false>
__global__ __launch_bounds__(PER_SUB ? 64 * PARTS : 256 * PARTS) void k_render_fwd_list(Params P, PlanView V, float *__restrict__ img, int tiles_x)
{
    const unsigned unit = xcd_swizzle(blockIdx.x, gridDim.x);
    const unsigned t = PER_SUB ? unit >> 2 : unit;
    const int bx = (int)(t % (unsigned)tiles_x), by = (int)(t / (unsigned)tiles_x);
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int sub = PER_SUB ? (int)(unit & 3u) : wv & 3;
    const unsigned part = (unsigned)(PER_SUB ? wv : wv >> 2);
    constexpr int NSUB = PER_SUB ? 1 : 4;      // sub-tiles of this workgroup
    __shared__ float4 s_stage[NSUB * PARTS][STAGE_F4];
    __shared__ float s_part[PARTS > 1 ? NSUB : 1][6][64];
    float4 *stage = s_stage[wv];
    v2f ar = {0.f, 0.f}, ag = {0.f, 0.f}, ab = {0.f, 0.f};
    const int bx0 = bx * 4 * SUBX, by0 = P.row0 + by * SUBY;
    const int sx0 = bx0 + sub * SUBX;
    if (sx0 < P.w) {   // wave-uniform (image width not a multiple of 32)
        // (the first 64 entries are requested together with the cursor that says how many of them are real: one dependent
        // round trip less in a wave whose whole life is three or four of them)
        const uint2 *__restrict__ ent = V.tl_entries + (size_t)t * (size_t)P.tl_cap;
        const unsigned q_first = part * 64u;
        uint2 e_first = q_first + (unsigned)lane < (unsigned)P.tl_cap ? ent[q_first + lane] : make_uint2(0u, 0u);
        const unsigned cnt = (unsigned)__builtin_amdgcn_readfirstlane((int)V.tl_cursor[(size_t)t * TL_STRIDE]);
        if (cnt > (unsigned)P.tl_cap) {
            fwd_tile<BOUNDED, PAIR, false>(P, V, sx0, by0, lane, part, (unsigned)PARTS, stage, ar, ag, ab);
        } else {
            const int X = sx0 + (lane & 7);
            const float px = V.px[(P.batch > 1 ? (by0 / P.slot) * P.w : 0) + min(X, P.w - 1)];
            const int Y0 = by0 + (lane >> 3);
            const v2f py = {V.py[min(Y0, P.h - 1)], V.py[min(Y0 + 8, P.h - 1)]};
            const float4 *__restrict__ rec = V.rec;
            const unsigned mybits = 0x11u << sub;
            const uint2 none = make_uint2(0u, 0u);
            // chunk k: entries -> hits, slots -> records (registers) -> stage -> evaluation; k+1's records fly under k's evaluation
            unsigned q = q_first;
            uint2 e = q + (unsigned)lane < cnt ? e_first : none;
            bool hit = (e.y & mybits) != 0u, needs = BOUNDED && (e.x >> 31) != 0u;
            float4 ra = make_float4(0.f, 0.f, 0.f, 0.f), rb = ra;
            if (hit) {
                const float4 *src = rec + 2 * (size_t)(e.x & 0x7fffffffu);
                ra = src[0];
                rb = src[1];
            }
            unsigned nq = q + 64u * PARTS;
            uint2 ne = nq + (unsigned)lane < cnt ? ent[nq + lane] : none;
            while (q < cnt) {
                // next chunk: hits and their record loads, then the entries of the one after -- requested before this
                // chunk is evaluated
                const bool chit = hit, cneeds = needs;
                const float4 ca = ra, cb = rb;
                q = nq;
                e = ne;
                hit = (e.y & mybits) != 0u;
                needs = BOUNDED && (e.x >> 31) != 0u;
                if (hit) {
                    const float4 *src = rec + 2 * (size_t)(e.x & 0x7fffffffu);
                    ra = src[0];
                    rb = src[1];
                }
                nq = q + 64u * PARTS;
                ne = nq + (unsigned)lane < cnt ? ent[nq + lane] : none;
                fwd_stage_eval<BOUNDED, PAIR>(stage, chit, cneeds, ca, cb, lane, px, py, P.dmax, ar, ag, ab);
            }
            // the large class (half-extent > 128 px) is in nobody's list
            const unsigned nlarge = V.cell_start[P.ncells + 1] - V.cell_start[P.ncells];
            if (__builtin_amdgcn_readfirstlane((int)nlarge) != 0)
                fwd_tile<BOUNDED, PAIR, true>(P, V, sx0, by0, lane, part, (unsigned)PARTS, stage, ar, ag, ab);
        }
    }
    if (PARTS > 1) {
        float (*o)[64] = s_part[PER_SUB ? 0 : sub];
        if (part != 0u) {
            o[0][lane] = ar.x; o[1][lane] = ar.y; o[2][lane] = ag.x; o[3][lane] = ag.y; o[4][lane] = ab.x; o[5][lane] = ab.y;
        }
        __syncthreads();
        if (part != 0u) return;
        ar.x += o[0][lane]; ar.y += o[1][lane]; ag.x += o[2][lane]; ag.y += o[3][lane]; ab.x += o[4][lane]; ab.y += o[5][lane];
    }
    if (sx0 < P.w) fwd_store(P, V, img, sx0, by0, lane, ar, ag, ab);
}

// The wide forward from tile lists: 32 x 32-px list tiles = the 2 x 2 sub-tiles of 16 x 16 px of k_render_fwd16's workgroup.
// A wave's hits are the entries whose quadrant mask meets the four quadrants of its sub-tile; the row-pair classes of
// k_render_fwd16 (window reaches both halves of the sub-tile / the upper eight rows only / the lower eight only) come from the
// same mask.  An overflowed tile falls back to a one-level search over the tile's cells with the wide evaluation.
template <bool BOUNDED>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_render_fwd16_list(Params P, PlanView V, float *__restrict__ img, int tiles_x)
{
    const unsigned tt = xcd_swizzle(blockIdx.x, gridDim.x);
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    __shared__ float4 s_stage[4][128];
    float4 *stage = s_stage[wv];
    const int bx0 = (int)(tt % (unsigned)tiles_x) * 2 * WIDE, by0 = P.row0 + (int)(tt / (unsigned)tiles_x) * 2 * WIDE;
    const int sx0 = bx0 + (wv & 1) * WIDE, sy0 = by0 + (wv >> 1) * WIDE;
    if (!(sx0 < P.w && sy0 < P.row1)) return;                     // wave-uniform; no barriers below
    const int sx1 = min(sx0 + WIDE - 1, P.w - 1), sy1 = min(sy0 + WIDE - 1, P.row1 - 1);
    const int X = sx0 + (lane & 15), Y = sy0 + (lane >> 4);
    const float px = V.px[min(X, P.w - 1)];
    const v2f pyA = {V.py[min(Y, P.h - 1)], V.py[min(Y + 4, P.h - 1)]};
    const v2f pyB = {V.py[min(Y + 8, P.h - 1)], V.py[min(Y + 12, P.h - 1)]};
    const float4 *__restrict__ rec = V.rec;
    const unsigned long long below = (1ull << lane) - 1ull;
    v2f acc[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) acc[k] = (v2f){0.f, 0.f};

    // stage the hits of one chunk sorted by row-pair class (cls 0 both, 1 upper rows only, 2 lower rows only, 3 dmax-tested,
    // 4 none) and evaluate them
    auto stage_eval = [&](int cls, const float4 ra, const float4 rb) {
        const unsigned long long m0 = __ballot(cls == 0), m1 = __ballot(cls == 1), m2 = __ballot(cls == 2);
        const unsigned long long m3 = BOUNDED ? __ballot(cls == 3) : 0ull;
        const int n0 = __builtin_popcountll(m0), n1 = __builtin_popcountll(m1), n2 = __builtin_popcountll(m2);
        const int n3 = __builtin_popcountll(m3);
        if (n0 + n1 + n2 + n3 == 0) return;
        __builtin_amdgcn_wave_barrier();
        if (cls < 4) {
            const unsigned long long mine = cls == 0 ? m0 : cls == 1 ? m1 : cls == 2 ? m2 : m3;
            const int base = cls == 0 ? 0 : cls == 1 ? n0 : cls == 2 ? n0 + n1 : n0 + n1 + n2;
            const int slot = base + __builtin_popcountll(mine & below);
            stage[2 * slot] = ra;
            stage[2 * slot + 1] = rb;
        }
        __builtin_amdgcn_wave_barrier();
        fwd_eval_lds16<false, 3>(stage, 0, n0, px, pyA, pyB, P.dmax, acc);
        fwd_eval_lds16<false, 1>(stage, n0, n0 + n1, px, pyA, pyB, P.dmax, acc);
        fwd_eval_lds16<false, 2>(stage, n0 + n1, n0 + n1 + n2, px, pyA, pyB, P.dmax, acc);
        if (BOUNDED) fwd_eval_lds16<true, 3>(stage, n0 + n1 + n2, n0 + n1 + n2 + n3, px, pyA, pyB, P.dmax, acc);
    };
    // one-level search of this sub-tile over candidate segments (the large class; every class when the tile's list overflowed)
    auto search = [&](bool large_only) {
        const uint4 *__restrict__ bbox = V.bbox;
        const unsigned *__restrict__ cs = V.cell_start;
        const int wtx = sx0 >> SUBX_SHIFT, wty = (sy0 - P.row0) >> SUBY_SHIFT;
        const int rx = large_only ? 0 : (int)V.hdr[8], ry = large_only ? 0 : (int)V.hdr[9];
        int nseg = 0;
        unsigned sbeg = 0, send = 0;
        if (rx > 0) {
            const int cx0 = max(sx0 - rx, 0) >> CELL_SHIFT, cx1 = min((sx1 + rx) >> CELL_SHIFT, P.ncx - 1);
            const int cy0 = max(sy0 - ry, 0) >> CELL_SHIFT, cy1 = min((sy1 + ry) >> CELL_SHIFT, P.ncy - 1);
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
        const unsigned nchunks = (total + 63u) >> 6;
        int rseg = 0;
        for (unsigned c = 0; c < nchunks; ++c) {
            const unsigned j = fwd_candidate(c, lane, nseg, rseg, sbeg, pex, pin);
            uint4 bb = make_uint4(0x7fffu, 0x7fffu, 0u, 0u);
            uint2 bs = make_uint2(0u, 0u);
            if (j != 0xffffffffu) {
                bb = bbox[2 * (size_t)j];
                bs = *reinterpret_cast<const uint2 *>(bbox + 2 * (size_t)j + 1);
            }
            const int c0 = (int)(bb.x & 0x7fffu), c1 = (int)(bb.x >> 16);
            const int r0 = (int)(bb.y & 0x7fffu), r1 = (int)(bb.y >> 16);
            bool hit = (c0 <= sx1) & (c1 >= sx0) & (r0 <= sy1) & (r1 >= sy0);
            if (bb.y & 0x8000u) {
                const unsigned t = (unsigned)(wty - ((r0 - P.row0) >> SUBY_SHIFT)) & 7u, sh = (t & 3u) * 8u;
                const unsigned lo = t < 4u ? bb.z : bs.x, hi = t < 4u ? bb.w : bs.y;
                const int txr = wtx - (c0 >> SUBX_SHIFT);
                hit &= (txr + 1 >= (int)((lo >> sh) & 0xffu)) & (txr <= (int)((hi >> sh) & 0xffu));
            }
            const bool needs = BOUNDED && (bb.x & 0x8000u) != 0u;
            const int cls = !hit ? 4 : needs ? 3 : r1 < sy0 + 8 ? 1 : r0 >= sy0 + 8 ? 2 : 0;
            float4 ra = make_float4(0.f, 0.f, 0.f, 0.f), rb = ra;
            if (hit) {
                ra = rec[2 * (size_t)j];
                rb = rec[2 * (size_t)j + 1];
            }
            stage_eval(cls, ra, rb);
        }
    };

    const unsigned cnt = (unsigned)__builtin_amdgcn_readfirstlane((int)V.tl_cursor[(size_t)tt * TL_STRIDE]);
    if (cnt > (unsigned)P.tl_cap) {
        search(false);
    } else {
        const uint2 *__restrict__ ent = V.tl_entries + (size_t)tt * (size_t)P.tl_cap;
        // the sub-tile's quadrants: columns 2 sx, 2 sx + 1 of quadrant rows 2 sy (upper eight pixel rows) and 2 sy + 1 (lower)
        const unsigned up = 0x3u << (2 * (wv & 1) + 8 * (wv >> 1)), lo = up << 4;
        const uint2 none = make_uint2(0u, 0u);
        auto classify = [&](const uint2 e) {
            const bool u = (e.y & up) != 0u, l = (e.y & lo) != 0u;
            return !(u || l) ? 4 : (BOUNDED && (e.x >> 31)) ? 3 : !l ? 1 : !u ? 2 : 0;
        };
        unsigned q = 0;
        uint2 e = (unsigned)lane < cnt ? ent[lane] : none;
        int cls = classify(e);
        float4 ra = make_float4(0.f, 0.f, 0.f, 0.f), rb = ra;
        if (cls < 4) {
            const float4 *src = rec + 2 * (size_t)(e.x & 0x7fffffffu);
            ra = src[0];
            rb = src[1];
        }
        uint2 ne = 64u + (unsigned)lane < cnt ? ent[64 + lane] : none;
        while (q < cnt) {
            // (the next chunk's records are requested before this chunk is evaluated)
            const int ncls = classify(ne);
            float4 na = make_float4(0.f, 0.f, 0.f, 0.f), nb = na;
            if (ncls < 4) {
                const float4 *src = rec + 2 * (size_t)(ne.x & 0x7fffffffu);
                na = src[0];
                nb = src[1];
            }
            q += 64u;
            const uint2 nne = q + 64u + (unsigned)lane < cnt ? ent[q + 64u + lane] : none;
            stage_eval(cls, ra, rb);
            cls = ncls; ra = na; rb = nb; ne = nne;
        }
        const unsigned nlarge = V.cell_start[P.ncells + 1] - V.cell_start[P.ncells];
        if (__builtin_amdgcn_readfirstlane((int)nlarge) != 0) search(true);
    }
    if (X < P.w) {
        fwd_store_px(P, img, X, Y, acc[0].x, acc[1].x, acc[2].x);
        fwd_store_px(P, img, X, Y + 4, acc[0].y, acc[1].y, acc[2].y);
        fwd_store_px(P, img, X, Y + 8, acc[3].x, acc[4].x, acc[5].x);
        fwd_store_px(P, img, X, Y + 12, acc[3].y, acc[4].y, acc[5].y);
    }
}

// Small images (fewer sub-tiles than the chip has wave slots, e.g. the 192x192 training crops of
// BASELINE config 5): a workgroup = ONE sub-tile, its candidate chunks dealt to all `blockDim/64` waves,
// partial sums combined through LDS.  Parallelism comes from the Gaussian list instead of from pixels.
template <bool BOUNDED, bool PAIR>
__global__ __launch_bounds__(1024) void k_render_fwd_split(Params P, PlanView V, float *__restrict__ img, int subs_x)
{
    __shared__ float s_part[16][6][64];
    __shared__ float4 s_stage[16][STAGE_F4];
    const unsigned t = xcd_swizzle(blockIdx.x, gridDim.x);
    const int sx0 = (int)(t % (unsigned)subs_x) * SUBX, sy0 = P.row0 + (int)(t / (unsigned)subs_x) * SUBY;
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nw = (int)(blockDim.x >> 6);
    v2f ar = {0.f, 0.f}, ag = {0.f, 0.f}, ab = {0.f, 0.f};
    fwd_tile<BOUNDED, PAIR>(P, V, sx0, sy0, lane, (unsigned)wv, (unsigned)nw, s_stage[wv], ar, ag, ab);
    if (wv > 0) {
        s_part[wv][0][lane] = ar.x; s_part[wv][1][lane] = ar.y;
        s_part[wv][2][lane] = ag.x; s_part[wv][3][lane] = ag.y;
        s_part[wv][4][lane] = ab.x; s_part[wv][5][lane] = ab.y;
    }
    __syncthreads();
    if (wv == 0) {
        for (int k = 1; k < nw; ++k) {
            ar.x += s_part[k][0][lane]; ar.y += s_part[k][1][lane];
            ag.x += s_part[k][2][lane]; ag.y += s_part[k][3][lane];
            ab.x += s_part[k][4][lane]; ab.y += s_part[k][5][lane];
        }
        fwd_store(P, V, img, sx0, sy0, lane, ar, ag, ab);
    }
}

}  // namespace

extern "C" {

int gsasr_splat_forward(const gsasr_dims *dims, const void *workspace, size_t workspace_bytes, float *img,
                        void *stream)
{
    Layout L;
    if (int rc = check_ws(dims, workspace, workspace_bytes, L)) return rc;
    const int rows = dims->row1 - dims->row0;
    if (rows == 0) return GSASR_OK;
    if (!img) return fail(GSASR_ERR_ARG, "null image pointer");
    const Params P = make_params(dims, L);
    const PlanView V = make_view(L, const_cast<void *>(workspace));
    const int subs_x = (dims->w + SUBX - 1) / SUBX, tiles_y = (rows + SUBY - 1) / SUBY;
    hipStream_t st = (hipStream_t)stream;
    const long nsub = (long)subs_x * tiles_y;
#ifdef FWD_PAIR
    const bool pair = FWD_PAIR != 0;
#else
    const bool pair = tl_dense(dims);     // record-pair packed evaluation for dense plans (fwd_eval_pair)
#endif
    if (fwd_wants_wide(dims)) {
        const int wx = (dims->w + 2 * WIDE - 1) / (2 * WIDE), wy = (rows + 2 * WIDE - 1) / (2 * WIDE);
        const dim3 grid((unsigned)wx * (unsigned)wy), block(256);
        if (L.tl_ok && L.tl_hlog == 5) {    // the plan's tile lists (32 x 32-px tiles)
            if (P.bounded) hipLaunchKernelGGL(k_render_fwd16_list<true>, grid, block, 0, st, P, V, img, wx);
            else hipLaunchKernelGGL(k_render_fwd16_list<false>, grid, block, 0, st, P, V, img, wx);
        } else if (P.bounded) hipLaunchKernelGGL(k_render_fwd16<true>, grid, block, 0, st, P, V, img, wx);
        else hipLaunchKernelGGL(k_render_fwd16<false>, grid, block, 0, st, P, V, img, wx);
    } else if (nsub < 4096 && !(L.tl_ok && L.tl_hlog == 4) && tl_dense(dims) && fwd_split_env()) {
        // fewer sub-tiles than half the chip's 8192 wave slots on a DENSE plan (a 192^2 training crop at 16 Gaussians per LR
        // pixel: thousands of candidates per sub-tile): split each sub-tile's Gaussian list over 2..16 waves so that about one
        // full set of waves is in flight.  Sparse small images (one Gaussian per LR pixel: a sub-tile has a few dozen hits, and
        // sixteen waves would each search all the candidates for them) take the two-level kernel below, two waves per sub-tile:
        // 512^2 x4 forward 21.6 -> 11.8 us, 640^2 31.1 -> 15.8, 256^2 level (profiles/history/r05_split_vs_twolevel.txt)
        int nw = 2;
        while (nw < 16 && nsub * nw < 8192) nw *= 2;
        const dim3 grid((unsigned)nsub), block((unsigned)nw * 64u);
#define GSASR_F1(K, B) do { if (pair) hipLaunchKernelGGL((K<B, true>), grid, block, 0, st, P, V, img, subs_x); \
                            else hipLaunchKernelGGL((K<B, false>), grid, block, 0, st, P, V, img, subs_x); } while (0)
        if (P.bounded) GSASR_F1(k_render_fwd_split, true);
        else GSASR_F1(k_render_fwd_split, false);
#undef GSASR_F1
    } else {
        // the plan's tile lists (32 x 16-px tiles: the same workgroup tile as the two-level walk, and its two shapes), else the
        // two-level walk; images with fewer than 4096 sub-tiles (half the chip's wave slots) get two waves per sub-tile: 512^2
        // -8..-11%, 640^2 level, 768^2..896^2 +3..4%, above that +5..11% (profiles/history/r05_fwd_parts.txt; until the lists of round 5
        // the line was at 8192: the search of a 4608-sub-tile canvas gained 13% from the second wave).  From lists the second wave
        // pays a little longer -- 768^2 at 16 per LR pixel -1%, the 4608-sub-tile canvas of config 5 -3%, 896^2 +1%: 6144 there
        const int tx4 = (subs_x + 3) / 4;
        static const int parts_env = dev_switch("GSASR_SPLAT_FWD_PARTS") ? atoi(dev_switch("GSASR_SPLAT_FWD_PARTS")) : 0;   // development: 1 | 2 waves per sub-tile
        const bool lists = L.tl_ok && L.tl_hlog == 4;
        const bool two = parts_env ? parts_env == 2 : nsub < (lists ? 6144 : 4096);
        // the fine forward (8 x 8-px sub-tiles, record pairs): sparse single images without lists.  development: GSASR_SPLAT_FWD8=0|1
        static const int fwd8_env = dev_switch("GSASR_SPLAT_FWD8") ? atoi(dev_switch("GSASR_SPLAT_FWD8")) : -1;
        if (!lists && dims->batch <= 1 && (fwd8_env >= 0 ? fwd8_env == 1 : FWD8_DEFAULT && !pair)) {
            const dim3 grid8((unsigned)tx4 * (unsigned)tiles_y), block8(512);
            if (P.bounded) hipLaunchKernelGGL(k_render_fwd8<true>, grid8, block8, 0, st, P, V, img, tx4);
            else hipLaunchKernelGGL(k_render_fwd8<false>, grid8, block8, 0, st, P, V, img, tx4);
            HIP_TRY(hipGetLastError());
            return GSASR_OK;
        }
        const dim3 grid((unsigned)tx4 * (unsigned)tiles_y), block(two ? 512 : 256);
#define GSASR_F3(K, B, T) do { if (pair) hipLaunchKernelGGL((K<B, T, true>), grid, block, 0, st, P, V, img, tx4); \
                               else hipLaunchKernelGGL((K<B, T, false>), grid, block, 0, st, P, V, img, tx4); } while (0)
#define GSASR_F2(K) do { if (P.bounded) { if (two) GSASR_F3(K, true, 2); else GSASR_F3(K, true, 1); } \
                         else { if (two) GSASR_F3(K, false, 2); else GSASR_F3(K, false, 1); } } while (0)
        // from lists: one workgroup per sub-tile with its list dealt to two waves (k_render_fwd_list, PER_SUB) where that evens
        // out the load: tile workgroups put ceil(tiles / 256 CUs) waves on the fullest SIMD, half-list waves
        // ceil(2 sub-tiles / 1024 SIMDs) halves.  736^2 at 16 per LR pixel 158.6 -> 146.8 us, 832^2 197 -> 182, 896^2 222 -> 204,
        // 640^2 128 -> 117; where the tiles deal out evenly (704^2, 1024^2) the tile form is 2-3% ahead and stays
        // (profiles/r06_fwd_persub.txt).  development: GSASR_SPLAT_FWD_PERSUB=0|1
        static const int persub_env = dev_switch("GSASR_SPLAT_FWD_PERSUB") ? atoi(dev_switch("GSASR_SPLAT_FWD_PERSUB")) : -1;
        const long on_simd_tile = 2 * (((long)grid.x + 255) / 256), on_simd_half = (8 * (long)grid.x + 1023) / 1024;   // in half-lists
        const bool persub = lists && (persub_env >= 0 ? persub_env == 1 : (!parts_env && on_simd_half < on_simd_tile));
        if (persub) {
            const bool two = parts_env != 1;
            const dim3 gs(grid.x * 4u), bs(two ? 128 : 64);
#define GSASR_F4(B, T) do { if (pair) hipLaunchKernelGGL((k_render_fwd_list<B, T, true, true>), gs, bs, 0, st, P, V, img, tx4); \
                            else hipLaunchKernelGGL((k_render_fwd_list<B, T, false, true>), gs, bs, 0, st, P, V, img, tx4); } while (0)
            if (P.bounded) { if (two) GSASR_F4(true, 2); else GSASR_F4(true, 1); }
            else { if (two) GSASR_F4(false, 2); else GSASR_F4(false, 1); }
#undef GSASR_F4
        } else if (lists) GSASR_F2(k_render_fwd_list);
        else GSASR_F2(k_render_fwd2);
#undef GSASR_F2
#undef GSASR_F3
    }
    HIP_TRY(hipGetLastError());
    return GSASR_OK;
}

int gsasr_forward_subtile_width(const gsasr_dims *dims)
{
    if (!dims || dims->s <= 0 || dims->w <= 0 || dims->row1 < dims->row0) return fail(GSASR_ERR_ARG, "bad dims");
    return fwd_wants_wide(dims) ? WIDE : SUBX;
}

}  // extern "C"
