// Microbenchmark (gfx950): what ONE SIMD sustains when 1..4 waves share it, for the instruction patterns a
// multi-utterance sample kernel would issue: scalar dependent add chains, PACKED dependent add chains
// (the two halves = two utterances), packed multiply + packed add pairs.  Plain C++ (the compiler pads hazards).
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-slp-vectorize -o simd_share simd_share.hip && ./simd_share
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x2 __attribute__((ext_vector_type(2)));
#define OPAQUE(x) asm volatile("" : "+v"(x))

template <int MODE>
__global__ void __launch_bounds__(1024) k(float *out, unsigned long long *cyc, float seed)
{
    float a = seed + threadIdx.x, b = seed * 0.5f, c = seed * 0.25f, x0 = seed, x1 = seed + 1, x2 = seed + 2;
    f32x2 A = {seed, seed + 1}, B = {seed + 2, seed + 3}, C = {seed + 4, seed + 5}, D = {seed + 6, seed + 7};
    f32x2 w0 = {seed, seed + 1}, w1 = {seed + 2, seed + 3}, y = {1.0001f, 0.9999f}, y2 = {1.0002f, 0.9998f};
    OPAQUE(b); OPAQUE(c);
    __syncthreads();
    unsigned long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
    for (int it = 0; it < 64; ++it) {
        if (MODE == 0) {          // 32 dependent scalar adds
#pragma unroll
            for (int i = 0; i < 16; ++i) { a += b; a += c; }
        } else if (MODE == 1) {   // 32 dependent packed adds (one chain)
#pragma unroll
            for (int i = 0; i < 16; ++i) { A += w0; A += w1; }
        } else if (MODE == 2) {   // 2 interleaved packed chains, 32 instr
#pragma unroll
            for (int i = 0; i < 16; ++i) { A += w0; B += w1; }
        } else if (MODE == 3) {   // 4 interleaved packed chains, 32 instr
#pragma unroll
            for (int i = 0; i < 8; ++i) { A += w0; B += w1; C += w0; D += w1; }
        } else if (MODE == 4) {   // 16 x (pk_mul + dependent pk_add): one chain
#pragma unroll
            for (int i = 0; i < 16; ++i) { OPAQUE(y); const f32x2 p = w0 * y; A += p; }
        } else if (MODE == 5) {   // 8 x 2 chains of (pk_mul + dependent pk_add)
#pragma unroll
            for (int i = 0; i < 8; ++i) { OPAQUE(y); OPAQUE(y2); const f32x2 p = w0 * y, q = w1 * y2; A += p; B += q; }
        } else if (MODE == 6) {   // 2 interleaved scalar chains
#pragma unroll
            for (int i = 0; i < 16; ++i) { a += b; x0 += c; }
        } else if (MODE == 7) {   // 4 interleaved scalar chains
#pragma unroll
            for (int i = 0; i < 8; ++i) { a += b; x0 += c; x1 += b; x2 += c; }
        } else if (MODE == 8) {   // 8 x (pk_mul forming 2 products + 2 dependent scalar adds) x 2 chains  (present GRU B pattern x2)
#pragma unroll
            for (int i = 0; i < 8; ++i) { OPAQUE(y); OPAQUE(y2); const f32x2 p = w0 * y, q = w1 * y2; a += p.x; x0 += q.x; a += p.y; x0 += q.y; }
        }
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a + x0 + x1 + x2 + A.x + A.y + B.x + B.y + C.x + C.y + D.x + D.y;
    if (threadIdx.x % 64 == 0) cyc[threadIdx.x / 64] = t1 - t0;
}

template <int MODE> void run(const char *name, int ops_per_iter)
{
    float *out; unsigned long long *cyc;
    (void)hipMalloc(&out, 4096 * 4); (void)hipMalloc(&cyc, 64 * 8);
    printf("%-52s", name);
    for (int waves : {1, 4, 8, 12, 16}) {
        hipLaunchKernelGGL(k<MODE>, dim3(1), dim3(64 * waves), 0, 0, out, cyc, 1.0f);
        hipLaunchKernelGGL(k<MODE>, dim3(1), dim3(64 * waves), 0, 0, out, cyc, 1.0f);
        (void)hipDeviceSynchronize();
        unsigned long long h[16];
        (void)hipMemcpy(h, cyc, sizeof(unsigned long long) * waves, hipMemcpyDeviceToHost);
        double mx = 0;
        for (int w = 0; w < waves; ++w) mx = h[w] > mx ? (double)h[w] : mx;
        const double per_wave = mx / (64.0 * ops_per_iter);               // cycles per instruction of the slowest wave
        const double per_simd = per_wave / ((waves + 3) / 4);             // SIMD cycles per instruction (waves/SIMD in flight)
        printf("  %2dw: %5.2f /wave %5.2f /simd", waves, per_wave, per_simd);
    }
    printf("\n");
    (void)hipFree(out); (void)hipFree(cyc);
}

int main()
{
    run<0>("32 dependent v_add_f32", 32);
    run<6>("2 interleaved scalar chains", 32);
    run<7>("4 interleaved scalar chains", 32);
    run<1>("32 dependent v_pk_add_f32", 32);
    run<2>("2 interleaved packed chains", 32);
    run<3>("4 interleaved packed chains", 32);
    run<4>("16 x (pk_mul + dependent pk_add)", 32);
    run<5>("8 x 2 chains (pk_mul + dependent pk_add)", 32);
    run<8>("8 x 2 chains (pk_mul + 2 dependent scalar adds)", 48);
    return 0;
}
