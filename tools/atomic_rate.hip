// measure fp32 global-atomic (agent scope, no return) and LDS-atomic throughput on gfx950 for the access shapes a
// tile-stationary backward would use (development aid; numbers quoted in DESIGN.md)
//   hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics tools/atomic_rate.hip -o tools/bin/atomic_rate
#include <hip/hip_runtime.h>
#include <cstdio>

// MODE 0: lane l adds to acc[(base+l)*8 + k], k = 0..7     (8 instructions, 32-byte stride across lanes)
// MODE 1: instruction i covers records base+8i .. base+8i+7, lane l -> record l/8, component l%8 (256 contiguous bytes)
// MODE 2: as 1 but only one instruction per round (64 contiguous floats = 8 records): the cost of ONE coalesced line set
// `spread`: consecutive waves start `spread` records apart (overlap between waves = contention on the same words)
template <int MODE>
__global__ __launch_bounds__(256) void k_glob(float *acc, int nrec, int rounds, int spread)
{
    const int lane = threadIdx.x & 63;
    const unsigned wave = blockIdx.x * 4 + (threadIdx.x >> 6);
    for (int r = 0; r < rounds; ++r) {
        const unsigned base = (unsigned)(((unsigned long long)wave * (unsigned)spread + (unsigned long long)r * 7919u) % (unsigned)(nrec - 64));
        if (MODE == 0) {
#pragma unroll
            for (int k = 0; k < 8; ++k) atomicAdd(acc + (size_t)(base + lane) * 8 + k, 1.0f);
        } else if (MODE == 1) {
#pragma unroll
            for (int i = 0; i < 8; ++i) atomicAdd(acc + (size_t)(base + 8 * i) * 8 + lane, 1.0f);
        } else {
            atomicAdd(acc + (size_t)base * 8 + lane, 1.0f);
        }
    }
}

// LDS: 8 ds_add_f32 per round, component-major [8][512] (conflict-free) vs record-major [512][8] (8-way conflicts)
template <int MODE>
__global__ __launch_bounds__(256) void k_lds(float *out, int rounds)
{
    __shared__ float s[8 * 512 + 64];
    for (int i = threadIdx.x; i < 8 * 512; i += 256) s[i] = 0.f;
    __syncthreads();
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int r = 0; r < rounds; ++r) {
        const int base = ((r * 37 + wv * 101) & 255) + lane;   // record index < 512
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            if (MODE == 0) atomicAdd(&s[k * 512 + base], 1.0f);
            else atomicAdd(&s[base * 8 + k], 1.0f);
        }
    }
    __syncthreads();
    float t = 0.f;
    for (int i = threadIdx.x; i < 8 * 512; i += 256) t += s[i];
    out[blockIdx.x * 256 + threadIdx.x] = t;
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
    const int nrec = 65536;
    float *acc, *out;
    hipMalloc(&acc, (size_t)nrec * 8 * 4 + 4096);
    hipMalloc(&out, 4096 * 256 * 4);
    hipMemset(acc, 0, (size_t)nrec * 8 * 4);
    const int blocks = 2048, rounds = 64;
    const double waves = blocks * 4.0;
    for (int spread : {64, 16, 1}) {
        float ms = time_ms([&] { k_glob<0><<<blocks, 256>>>(acc, nrec, rounds, spread); });
        printf("global strided   spread %2d: %8.1f us  %7.2f G lane-atomics/s  (%.1f ns per wave-instr, chip-wide)\n", spread, ms * 1e3,
               waves * rounds * 8 * 64 / (ms * 1e-3) / 1e9, ms * 1e6 / (waves * rounds * 8));
        ms = time_ms([&] { k_glob<1><<<blocks, 256>>>(acc, nrec, rounds, spread); });
        printf("global coalesced spread %2d: %8.1f us  %7.2f G lane-atomics/s  (%.1f ns per wave-instr, chip-wide)\n", spread, ms * 1e3,
               waves * rounds * 8 * 64 / (ms * 1e-3) / 1e9, ms * 1e6 / (waves * rounds * 8));
        ms = time_ms([&] { k_glob<2><<<blocks, 256>>>(acc, nrec, rounds, spread); });
        printf("global one-instr spread %2d: %8.1f us  %7.2f G lane-atomics/s  (%.1f ns per wave-instr, chip-wide)\n", spread, ms * 1e3,
               waves * rounds * 64 / (ms * 1e-3) / 1e9, ms * 1e6 / (waves * rounds));
    }
    for (int m = 0; m < 2; ++m) {
        float ms = m == 0 ? time_ms([&] { k_lds<0><<<1024, 256>>>(out, 1024); }) : time_ms([&] { k_lds<1><<<1024, 256>>>(out, 1024); });
        printf("LDS ds_add_f32 %s: %8.1f us -> %.2f cycles per wave-instr per CU @2.4GHz (4 waves/CU x 4 WG)\n",
               m == 0 ? "component-major" : "record-major   ", ms * 1e3, ms * 1e-3 * 2.4e9 / (4.0 * 1024 * 8 * 4));
    }
    return 0;
}
