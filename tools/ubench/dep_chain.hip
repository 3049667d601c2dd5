// Microbenchmark: cycles per dependent v_add_f32 / independent ops for one wave on a SIMD (gfx950).
// SUPERSEDED by issue_rate.hip: every instruction here is an inline-asm statement, and the compiler puts an s_nop after
// each of those, so the figures this prints (8.7 cycles per dependent add, ...) include a nop per instruction.  Kept
// because DESIGN.md refers to the artefact.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ void k(float *out, unsigned long long *cyc, float seed)
{
    float a = seed + threadIdx.x, b = seed * 0.5f;
    float x0 = seed, x1 = seed + 1, x2 = seed + 2, x3 = seed + 3;
    f32x2 w = {seed, seed + 1}, y = {1.0001f, 0.9999f};
    unsigned long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
    for (int it = 0; it < 64; ++it) {
        if (MODE == 0) {          // 32 dependent adds
#pragma unroll
            for (int i = 0; i < 32; ++i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a) : "v"(b));
        } else if (MODE == 1) {   // 32 independent adds (4 chains)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                asm volatile("v_add_f32 %0, %0, %1" : "+v"(x0) : "v"(b));
                asm volatile("v_add_f32 %0, %0, %1" : "+v"(x1) : "v"(b));
                asm volatile("v_add_f32 %0, %0, %1" : "+v"(x2) : "v"(b));
                asm volatile("v_add_f32 %0, %0, %1" : "+v"(x3) : "v"(b));
            }
        } else if (MODE == 2) {   // GRU B pattern: 2 pk_mul + 4 dependent adds, x8
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                f32x2 p0, p1;
                asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(p0) : "v"(w), "v"(y));
                asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(p1) : "v"(y), "v"(w));
                asm volatile("v_add_f32 %0, %0, %1" : "+v"(a) : "v"(p0.x));
                asm volatile("v_add_f32 %0, %0, %1" : "+v"(a) : "v"(p0.y));
                asm volatile("v_add_f32 %0, %0, %1" : "+v"(a) : "v"(p1.x));
                asm volatile("v_add_f32 %0, %0, %1" : "+v"(a) : "v"(p1.y));
            }
        } else if (MODE == 3) {   // 32 independent pk_mul
#pragma unroll
            for (int i = 0; i < 32; ++i) { f32x2 p; asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(p) : "v"(w), "v"(y)); asm volatile("" :: "v"(p)); }
        } else if (MODE == 4) {   // 2 interleaved dependent chains (16 each)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                asm volatile("v_add_f32 %0, %0, %1" : "+v"(a) : "v"(b));
                asm volatile("v_add_f32 %0, %0, %1" : "+v"(x0) : "v"(b));
            }
        } else if (MODE == 5) {   // mul then dependent add (no pk): 16 x (mul, add)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                float p;
                asm volatile("v_mul_f32 %0, %1, %2" : "=v"(p) : "v"(x1), "v"(b));
                asm volatile("v_add_f32 %0, %0, %1" : "+v"(a) : "v"(p));
            }
        }
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a + x0 + x1 + x2 + x3;
    if (threadIdx.x % 64 == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

template <int MODE> void run_half(const char *name, int ops_per_iter)
{
    float *out; unsigned long long *cyc;
    hipMalloc(&out, 4096 * 4); hipMalloc(&cyc, 64 * 8);
    for (int threads : {64, 32, 16}) {
        hipLaunchKernelGGL(k<MODE>, dim3(1), dim3(threads), 0, 0, out, cyc, 1.0f);
        hipLaunchKernelGGL(k<MODE>, dim3(1), dim3(threads), 0, 0, out, cyc, 1.0f);
        hipDeviceSynchronize();
        unsigned long long h[1];
        hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
        printf("%-44s active lanes=%d: %.2f cycles/instr\n", name, threads, (double)h[0] / (64.0 * ops_per_iter));
    }
    hipFree(out); hipFree(cyc);
}

template <int MODE> void run(const char *name, int waves, int ops_per_iter)
{
    float *out; unsigned long long *cyc;
    hipMalloc(&out, 4096 * 4); hipMalloc(&cyc, 64 * 8);
    hipLaunchKernelGGL(k<MODE>, dim3(1), dim3(64 * waves), 0, 0, out, cyc, 1.0f);
    hipLaunchKernelGGL(k<MODE>, dim3(1), dim3(64 * waves), 0, 0, out, cyc, 1.0f);
    hipDeviceSynchronize();
    unsigned long long h[64];
    hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    printf("%-44s waves/WG=%d: ", name, waves);
    for (int w = 0; w < waves; ++w) printf("%.2f ", (double)h[w] / (64.0 * ops_per_iter));
    printf(" cycles/instr\n");
    hipFree(out); hipFree(cyc);
}

int main()
{
    run_half<0>("32 dependent v_add_f32", 32);
    run_half<1>("32 independent v_add_f32", 32);
    run_half<2>("8 x (2 pk_mul + 4 dependent add)", 48);
    for (int waves : {1}) {
        run<0>("32 dependent v_add_f32", waves, 32);
        run<1>("32 independent v_add_f32 (4 chains)", waves, 32);
        run<4>("2 interleaved dependent chains", waves, 32);
        run<3>("32 independent v_pk_mul_f32", waves, 32);
        run<2>("8 x (2 pk_mul + 4 dependent add)", waves, 48);
        run<5>("16 x (v_mul + dependent v_add)", waves, 32);
    }
    return 0;
}
