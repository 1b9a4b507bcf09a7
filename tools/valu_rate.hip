// measure VALU issue cost per wave64 instruction on gfx950: v_fma_f32, v_pk_fma_f32, v_exp_f32, v_mul_f32
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));
#define N_IT 4096
template <int MODE>
__global__ __launch_bounds__(256) void k(float *out, float s)
{
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    v2f p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, p4 = p0 + 1.f, p5 = p1 + 1.f, p6 = p2 + 1.f, p7 = p3 + 1.f;
    const v2f sv = {s, s * 0.5f};
    for (int i = 0; i < N_IT; ++i) {
        if (MODE == 0) {  // 8 independent v_fma_f32
            a0 = fmaf(a0, s, 1.f); a1 = fmaf(a1, s, 1.f); a2 = fmaf(a2, s, 1.f); a3 = fmaf(a3, s, 1.f);
            a4 = fmaf(a4, s, 1.f); a5 = fmaf(a5, s, 1.f); a6 = fmaf(a6, s, 1.f); a7 = fmaf(a7, s, 1.f);
        } else if (MODE == 1) {  // 8 independent v_pk_fma_f32
            p0 = p0 * sv + 1.f; p1 = p1 * sv + 1.f; p2 = p2 * sv + 1.f; p3 = p3 * sv + 1.f;
            p4 = p4 * sv + 1.f; p5 = p5 * sv + 1.f; p6 = p6 * sv + 1.f; p7 = p7 * sv + 1.f;
        } else if (MODE == 2) {  // 8 independent v_exp_f32
            a0 = __builtin_amdgcn_exp2f(a0); a1 = __builtin_amdgcn_exp2f(a1); a2 = __builtin_amdgcn_exp2f(a2); a3 = __builtin_amdgcn_exp2f(a3);
            a4 = __builtin_amdgcn_exp2f(a4); a5 = __builtin_amdgcn_exp2f(a5); a6 = __builtin_amdgcn_exp2f(a6); a7 = __builtin_amdgcn_exp2f(a7);
        } else if (MODE == 3) {  // 8 independent v_mul_f32
            a0 *= s; a1 *= s; a2 *= s; a3 *= s; a4 *= s; a5 *= s; a6 *= s; a7 *= s;
        } else if (MODE == 4) {  // 4 fma + 4 exp interleaved
            a0 = fmaf(a0, s, 1.f); a1 = __builtin_amdgcn_exp2f(a1); a2 = fmaf(a2, s, 1.f); a3 = __builtin_amdgcn_exp2f(a3);
            a4 = fmaf(a4, s, 1.f); a5 = __builtin_amdgcn_exp2f(a5); a6 = fmaf(a6, s, 1.f); a7 = __builtin_amdgcn_exp2f(a7);
        } else if (MODE == 5) {  // v_cndmask + v_cmp
            a0 = a0 > s ? a1 : a0; a2 = a2 > s ? a3 : a2; a4 = a4 > s ? a5 : a4; a6 = a6 > s ? a7 : a6;
            a1 = a1 > s ? a2 : a1; a3 = a3 > s ? a4 : a3; a5 = a5 > s ? a6 : a5; a7 = a7 > s ? a0 : a7;
        }
        asm volatile("" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        asm volatile("" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7));
    }
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p1.y + p2.x + p3.y + p4.x + p5.y + p6.x + p7.y;
}
template <int MODE>
void run(const char *name, int waves_per_simd, int instr_per_iter)
{
    float *out;
    hipMalloc(&out, 256 * 4096 * 4);
    const int blocks = 256 * waves_per_simd;  // 256 CUs x (waves_per_simd x 4 SIMDs / 4 waves per block)
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    k<MODE><<<blocks, 256>>>(out, 0.999f);
    hipDeviceSynchronize();
    hipEventRecord(a);
    k<MODE><<<blocks, 256>>>(out, 0.999f);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    const double instr_per_simd = (double)waves_per_simd * N_IT * instr_per_iter;
    printf("%-28s waves/SIMD=%d  %.1f us  -> %.2f ns per wave-instr per SIMD (= %.2f cycles @2.4GHz)\n", name, waves_per_simd,
           ms * 1e3, ms * 1e6 / instr_per_simd, ms * 1e6 / instr_per_simd * 2.4);
    hipFree(out);
}
int main()
{
    for (int w : {1, 2, 4, 8}) {
        run<0>("v_fma_f32", w, 8);
        run<1>("v_pk_fma_f32", w, 8);
        run<2>("v_exp_f32", w, 8);
        run<3>("v_mul_f32", w, 8);
        run<4>("fma+exp mix", w, 8);
        run<5>("cmp+cndmask (16 instr)", w, 16);
    }
    return 0;
}
