// Vector-memory (TA/TD) throughput on gfx950 for the load shapes of the Gaussian-stationary backward: how many cycles
// does the CU's texture path spend per wave-load, as a function of load width, lane stride and number of active lanes?
// Every wave re-reads a small (L1-resident) footprint, so what is measured is the pipe, not the caches behind it.
//   hipcc --offload-arch=gfx950 -O3 tools/ta_rate.hip -o tools/bin/ta_rate     (development aid; numbers in DESIGN.md)
#include <hip/hip_runtime.h>
#include <cstdio>

typedef unsigned u3v __attribute__((ext_vector_type(3)));
typedef unsigned u4v __attribute__((ext_vector_type(4)));
typedef unsigned u2v __attribute__((ext_vector_type(2)));

// MODE 0: dwordx3, 12-byte lane stride (two rows of 32 pixels: lanes 32.. are `pitch` bytes further)
// MODE 1: dwordx4, 16-byte lane stride (1024 contiguous bytes)
// MODE 2: dwordx2 + dword at 12-byte stride (same bytes as MODE 0 in two instructions)
// MODE 3: dword, 4-byte stride
// MODE 4: dwordx4 at 12-byte lane stride (over-fetch: the 4th dword is the next pixel's first channel)
// `active`: lanes (of each half of 32) that take part; the others are exec-masked
template <int MODE>
__global__ __launch_bounds__(256) void k_load(const float *src, float *out, int rounds, int active, int pitch)
{
    const int lane = threadIdx.x & 63, col = lane & 31, half = lane >> 5;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(src), 0, 1 << 20, 0x00020000);
    unsigned acc = 0;
    if (col < active) {
        const int lstride = MODE == 1 ? 16 : (MODE == 3 ? 4 : 12);
        const int voff = col * lstride + half * pitch;
        int soff = 0;
        for (int r = 0; r < rounds; ++r) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {   // four independent loads in flight per round
                if (MODE == 0) { const u3v v = __builtin_amdgcn_raw_buffer_load_b96(rsrc, voff, soff, 0); acc += v.x ^ v.y ^ v.z; }
                if (MODE == 1 || MODE == 4) { const u4v v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, soff, 0); acc += v.x ^ v.y ^ v.z ^ v.w; }
                if (MODE == 2) { const u2v v = __builtin_amdgcn_raw_buffer_load_b64(rsrc, voff, soff, 0);
                                 const unsigned w = __builtin_amdgcn_raw_buffer_load_b32(rsrc, voff + 8, soff, 0); acc += v.x ^ v.y ^ w; }
                if (MODE == 3) { acc += __builtin_amdgcn_raw_buffer_load_b32(rsrc, voff, soff, 0); }
                soff = (soff + 2 * pitch) & 0x3fff;   // stay inside 16 KB: L1 hits
            }
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = __uint_as_float(acc);
}

template <class F>
float time_ms(F f)
{
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    f();
    hipDeviceSynchronize();
    hipEventRecord(a);
    f();
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    return ms;
}

int main()
{
    float *src, *out;
    hipMalloc(&src, 2 << 20);
    hipMalloc(&out, 8192 * 256 * 4);
    hipMemset(src, 0, 2 << 20);
    const int blocks = 256 * 7, rounds = 256;   // 7 workgroups of 4 waves per CU
    const double clk = 2.4e9;
    const char *names[] = {"dwordx3  stride 12", "dwordx4  stride 16", "dwordx2+dword s12", "dword    stride 4", "dwordx4  stride 12"};
    for (int mode = 0; mode < 5; ++mode)
        for (int active : {32, 23, 16, 8}) {
            const int pitch = 1536;
            float ms = 0;
            auto run = [&](auto kern) { ms = time_ms([&] { kern<<<blocks, 256>>>(src, out, rounds, active, pitch); }); };
            if (mode == 0) run(k_load<0>);
            if (mode == 1) run(k_load<1>);
            if (mode == 2) run(k_load<2>);
            if (mode == 3) run(k_load<3>);
            if (mode == 4) run(k_load<4>);
            const double loads_per_cu = 7.0 * 4 * rounds * 4;     // wave-level load groups (MODE 2: pairs) per CU
            const double cyc = ms * 1e-3 * clk / loads_per_cu;
            const double bytes = (mode == 1 || mode == 4 ? 16.0 : mode == 3 ? 4.0 : 12.0) * 2 * active;
            printf("%-20s active %2d/32: %7.1f us  %6.1f cycles per wave-load per CU  (%5.1f useful B/clk/CU)\n", names[mode], active,
                   ms * 1e3, cyc, bytes / cyc);
        }
    return 0;
}
