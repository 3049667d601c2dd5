// Microbenchmark (gfx950): per-wave VALU issue rate and dependent latency, written in plain C++ so that the compiler
// schedules hazards itself (inline-asm instructions get an s_nop after each one, which is what dep_chain.hip measured).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x2 __attribute__((ext_vector_type(2)));
#define OPAQUE(x) asm volatile("" : "+v"(x))

template <int MODE>
__global__ void __launch_bounds__(512) k(float *out, unsigned long long *cyc, float seed)
{
    float a = seed + threadIdx.x, b = seed * 0.5f, c = seed * 0.25f;
    float x0 = seed, x1 = seed + 1, x2 = seed + 2, x3 = seed + 3, x4 = seed + 4, x5 = seed + 5, x6 = seed + 6, x7 = seed + 7;
    f32x2 w0 = {seed, seed + 1}, w1 = {seed + 2, seed + 3}, y = {1.0001f, 0.9999f};
    OPAQUE(b); OPAQUE(c);
    unsigned long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
    for (int it = 0; it < 64; ++it) {
        if (MODE == 0) {          // 32 dependent adds
#pragma unroll
            for (int i = 0; i < 16; ++i) { a += b; a += c; }
        } else if (MODE == 1) {   // 32 independent adds (8 chains)
#pragma unroll
            for (int i = 0; i < 4; ++i) { x0 += b; x1 += b; x2 += b; x3 += b; x4 += c; x5 += c; x6 += c; x7 += c; }
        } else if (MODE == 2) {   // 8 x (2 pk_mul + 4 dependent adds) = the present GRU B inner pattern
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                OPAQUE(y);
                const f32x2 p0 = w0 * y, p1 = w1 * y;
                a += p0.x; a += p0.y; a += p1.x; a += p1.y;
            }
        } else if (MODE == 3) {   // 32 independent pk_mul
#pragma unroll
            for (int i = 0; i < 32; ++i) { OPAQUE(y); f32x2 p = w0 * y; asm volatile("" :: "v"(p)); }
        } else if (MODE == 4) {   // 2 interleaved dependent chains
#pragma unroll
            for (int i = 0; i < 16; ++i) { a += b; x0 += c; }
        } else if (MODE == 5) {   // 16 x (mul, dependent add)
#pragma unroll
            for (int i = 0; i < 16; ++i) { OPAQUE(x1); const float p = x1 * b; a += p; }
        } else if (MODE == 6) {   // 4 interleaved dependent chains
#pragma unroll
            for (int i = 0; i < 8; ++i) { a += b; x0 += c; x1 += b; x2 += c; }
        }
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a + x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
    if (threadIdx.x % 64 == 0) cyc[threadIdx.x / 64] = t1 - t0;
}

template <int MODE> void run(const char *name, int ops_per_iter)
{
    float *out; unsigned long long *cyc;
    (void)hipMalloc(&out, 4096 * 4); (void)hipMalloc(&cyc, 64 * 8);
    for (int waves : {1, 4, 8}) {
        hipLaunchKernelGGL(k<MODE>, dim3(1), dim3(64 * waves), 0, 0, out, cyc, 1.0f);
        hipLaunchKernelGGL(k<MODE>, dim3(1), dim3(64 * waves), 0, 0, out, cyc, 1.0f);
        (void)hipDeviceSynchronize();
        unsigned long long h[8];
        (void)hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
        printf("%-44s waves/WG=%d: %.2f cycles/instr (wave 0)\n", name, waves, (double)h[0] / (64.0 * ops_per_iter));
    }
    (void)hipFree(out); (void)hipFree(cyc);
}

int main()
{
    run<0>("32 dependent v_add_f32", 32);
    run<1>("32 independent v_add_f32 (8 chains)", 32);
    run<4>("2 interleaved dependent chains", 32);
    run<6>("4 interleaved dependent chains", 32);
    run<3>("32 independent v_pk_mul_f32", 32);
    run<2>("8 x (2 pk_mul + 4 dependent add)", 48);
    run<5>("16 x (v_mul + dependent v_add)", 32);
    return 0;
}
