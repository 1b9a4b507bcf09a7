// splat_shard.hip -- multi-GPU row bands: selection of the Gaussians that cross a band edge, merge of their returned gradients
// (one translation unit of libgsasr_splat.so; gsasr_splat.hip has the overview of the whole pipeline)
#include "splat_common.h"

using namespace gsasr_detail;

namespace {

// ---------------------------------------------------------------------------------------------------
// row-band shard: neighbour exchange (device side of gsasr_amd/shard.py)
// ---------------------------------------------------------------------------------------------------
// One thread per Gaussian: its row window on the FULL grid (the same gaussian_box() the plan uses, so the
// selection is exactly the set of Gaussians the neighbour's plan would keep) against this rank's band.
__global__ __launch_bounds__(256) void k_band_select(Params P, int band0, int band1, int rows_above, int rows_below,
                                                     int cap, const float *__restrict__ packed,
                                                     float *__restrict__ up, float *__restrict__ down,
                                                     int *__restrict__ up_index, int *__restrict__ down_index,
                                                     int *__restrict__ counts)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63;
    bool go_up = false, go_down = false, far = false;
    float4 ra = make_float4(0.f, 0.f, 0.f, 0.f), rb = ra;
    if (i < P.s) {
        ra = reinterpret_cast<const float4 *>(packed)[2 * (size_t)i];      // sx sy rho x
        rb = reinterpret_cast<const float4 *>(packed)[2 * (size_t)i + 1];  // y r g b
        const Box b = gaussian_box(ra.x, ra.y, ra.w, rb.x, P, Geo{P.h, P.w, 0, 0}, P.kcut);   // P.row0/row1 = whole grid here
        if (b.cls != 2) {
            go_up = rows_above > 0 && b.r0 < band0;
            go_down = rows_below > 0 && b.r1 >= band1;
            far = (go_up && b.r0 < band0 - rows_above) || (go_down && b.r1 >= band1 + rows_below);
        }
    }
    // wave-aggregated slot allocation: one returning atomic per wave and list
    const unsigned long long mu = __ballot(go_up), md = __ballot(go_down), mf = __ballot(far);
    const unsigned long long below = (1ull << lane) - 1ull;
    int bu = 0, bd = 0;
    if (lane == 0) {
        if (mu) bu = atomicAdd(&counts[0], __builtin_popcountll(mu));
        if (md) bd = atomicAdd(&counts[1], __builtin_popcountll(md));
        if (mf) atomicAdd(&counts[2], __builtin_popcountll(mf));
    }
    bu = __shfl(bu, 0);
    bd = __shfl(bd, 0);
    if (go_up) {
        const int slot = bu + __builtin_popcountll(mu & below);
        if (slot < cap) {
            reinterpret_cast<float4 *>(up)[2 * (size_t)slot] = ra;
            reinterpret_cast<float4 *>(up)[2 * (size_t)slot + 1] = rb;
            up_index[slot] = i;
        }
    }
    if (go_down) {
        const int slot = bd + __builtin_popcountll(md & below);
        if (slot < cap) {
            reinterpret_cast<float4 *>(down)[2 * (size_t)slot] = ra;
            reinterpret_cast<float4 *>(down)[2 * (size_t)slot + 1] = rb;
            down_index[slot] = i;
        }
    }
}

// 8 threads per returned record; a Gaussian can sit in both lists, hence atomics (two adds at most per word)
__global__ __launch_bounds__(256) void k_band_merge(int s, int cap, float *__restrict__ g_packed,
                                                    const float *__restrict__ g_up, const float *__restrict__ g_down,
                                                    const int *__restrict__ up_index, const int *__restrict__ down_index,
                                                    const int *__restrict__ counts)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int j = t >> 3, k = t & 7;
    if (j >= 2 * cap) return;
    const bool is_down = j >= cap;
    const int jj = is_down ? j - cap : j;
    if (jj >= min(counts[is_down ? 1 : 0], cap)) return;
    const int i = (is_down ? down_index : up_index)[jj];
    if ((unsigned)i >= (unsigned)s) return;
    atomicAdd(&g_packed[(size_t)i * 8 + k], (is_down ? g_down : g_up)[(size_t)jj * 8 + k]);
}

}  // namespace

extern "C" {

int gsasr_band_select(const float *packed, const gsasr_dims *dims, int rows_above, int rows_below, int cap,
                      float *up, float *down, int *up_index, int *down_index, int *counts, void *stream)
{
    if (!dims_ok(dims) || dims->batch > 1) return fail(GSASR_ERR_ARG, "bad dims (the band exchange does not take a batched canvas)");
    if (cap < 0 || rows_above < 0 || rows_below < 0 || !counts || (cap > 0 && (!up || !down || !up_index || !down_index)) ||
        (dims->s > 0 && !packed))
        return fail(GSASR_ERR_ARG, "gsasr_band_select: null pointer or negative size");
    hipStream_t st = (hipStream_t)stream;
    gsasr_dims whole = *dims;   // the footprint is taken on the full grid, then compared with the band
    whole.row0 = 0;
    whole.row1 = dims->h;
    const Layout L = make_layout(&whole);
    const Params P = make_params(&whole, L);
    HIP_TRY(hipMemsetAsync(counts, 0, 4 * sizeof(int), st));
    if (cap > 0) {  // 0xff.. = NaN records: dead Gaussians for every kernel of this library
        HIP_TRY(hipMemsetAsync(up, 0xff, (size_t)cap * 32, st));
        HIP_TRY(hipMemsetAsync(down, 0xff, (size_t)cap * 32, st));
    }
    if (dims->s > 0)
        hipLaunchKernelGGL(k_band_select, dim3((dims->s + 255) / 256), dim3(256), 0, st, P, dims->row0, dims->row1,
                           rows_above, rows_below, cap, packed, up, down, up_index, down_index, counts);
    HIP_TRY(hipGetLastError());
    return GSASR_OK;
}

int gsasr_band_merge(float *g_packed, int s, const float *g_up, const float *g_down, const int *up_index,
                     const int *down_index, const int *counts, int cap, void *stream)
{
    if (s < 0 || cap < 0 || !counts || (cap > 0 && ((s > 0 && !g_packed) || !g_up || !g_down || !up_index || !down_index)))   // (a rank may own no Gaussian)
        return fail(GSASR_ERR_ARG, "gsasr_band_merge: null pointer or negative size");
    if (cap == 0 || s == 0) return GSASR_OK;
    hipLaunchKernelGGL(k_band_merge, dim3((2 * cap * 8 + 255) / 256), dim3(256), 0, (hipStream_t)stream, s, cap,
                       g_packed, g_up, g_down, up_index, down_index, counts);
    HIP_TRY(hipGetLastError());
    return GSASR_OK;
}

}  // extern "C"
