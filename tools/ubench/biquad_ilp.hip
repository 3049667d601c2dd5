// How fast does the HGA filter step run with 1, 2 or 4 independent columns per lane?  (development microbenchmark, gfx950)
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o biquad_ilp biquad_ilp.hip && ./biquad_ilp
// One step = csrc/hga_kernels.hip HGA_BIQUAD (5 fp64 products, 4 sums, unfused) behind a row_shr:1 DPP move of the
// neighbour lane's previous output.  Also: a chain of dependent v_add_f64 (latency per dependent fp64 operation).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#define CHECK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { printf("%s: %s\n", #e, hipGetErrorString(_e)); return 1; } } while (0)
__device__ __forceinline__ double shift_in(double x, double y)
{
    const unsigned long long ux = __builtin_bit_cast(unsigned long long, x), uy = __builtin_bit_cast(unsigned long long, y);
    const int lo = __builtin_amdgcn_update_dpp((int)(unsigned)ux, (int)(unsigned)uy, 0x111, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp((int)(unsigned)(ux >> 32), (int)(unsigned)(uy >> 32), 0x111, 0xf, 0xf, false);
    return __builtin_bit_cast(double, ((unsigned long long)(unsigned)hi << 32) | (unsigned)lo);
}
template <int NC, int STORE>
__global__ void __launch_bounds__(256) k(const double *coef, double *out, long long *cyc, int steps)
{
    __shared__ double xs[64][16 * NC];       // (with the stores: 8 + 8 + 2 KB per block at one column per lane, so that 8 blocks per CU are resident)
    __shared__ double ys[64][16];
    __shared__ double dump[256];
    for (int i = threadIdx.x; i < 64 * 16 * NC; i += blockDim.x) xs[i / (16 * NC)][i % (16 * NC)] = 1e-3 * (i % 97);
    __syncthreads();
    const int r = threadIdx.x & 15, pib = threadIdx.x >> 4;
    const double b0 = coef[r], b1 = coef[16 + r], b2 = coef[32 + r], a1 = coef[48 + r], a2 = coef[64 + r];
    double y[NC], z0[NC], z1[NC];
    for (int c = 0; c < NC; ++c) { y[c] = 0; z0[c] = 0.1 * c; z1[c] = 0.2 * c; }
    long long t0 = __builtin_readcyclecounter();
    for (int k4 = 0; k4 < steps; k4 += 4) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                const double in = shift_in(xs[(k4 + u) & 63][pib + 16 * c], y[c]);
                y[c] = b0 * in + z0[c];
                if (STORE == 1) { if (r == 15) ys[(k4 + u) & 63][pib] = y[c]; }                       // exec-masked store, last section's lane
                if (STORE == 2) { *(r == 15 ? &ys[(k4 + u) & 63][pib] : &dump[threadIdx.x]) = y[c]; }  // every lane stores
                z0[c] = b1 * in - a1 * y[c] + z1[c];
                z1[c] = b2 * in - a2 * y[c];
            }
        }
    }
    long long t1 = __builtin_readcyclecounter();
    double s = 0;
    for (int c = 0; c < NC; ++c) s += y[c] + z0[c] + z1[c];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}
__global__ void addchain(double *out, long long *cyc, double seed)
{
    double a = seed, b = seed * 0.5;
    asm volatile("" : "+v"(b));
    long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < 256; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u) a += b;
    }
    long long t1 = __builtin_readcyclecounter();
    out[threadIdx.x] = a;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
template <int NC, int STORE = 0> int run(const double *dc, double *dout, long long *dcyc, int blocks_per_cu)
{
    const int steps = 2048, blocks = 256 * blocks_per_cu;
    hipLaunchKernelGGL((k<NC, STORE>), dim3(blocks), dim3(256), 0, 0, dc, dout, dcyc, steps);
    hipLaunchKernelGGL((k<NC, STORE>), dim3(blocks), dim3(256), 0, 0, dc, dout, dcyc, steps);
    CHECK(hipDeviceSynchronize());
    std::vector<long long> c(blocks * 4);
    CHECK(hipMemcpy(c.data(), dcyc, c.size() * 8, hipMemcpyDeviceToHost));
    double s = 0; for (auto v : c) s += (double)v;
    const double per_step = s / c.size() / steps;
    printf("store mode %d: %d column(s) per lane, %d block(s) of 4 waves per CU: %7.1f cycles per step and wave = %6.1f per column-step; per SIMD %6.1f cycles per column-step\n",
           STORE, NC, blocks_per_cu, per_step, per_step / NC, per_step / NC / blocks_per_cu);
    return 0;
}
int main()
{
    double *dc, *dout; long long *dcyc;
    CHECK(hipMalloc(&dc, 80 * 8)); CHECK(hipMalloc(&dout, 256 * 8 * 256 * 8)); CHECK(hipMalloc(&dcyc, 256 * 8 * 4 * 8));
    std::vector<double> hc(80); for (int i = 0; i < 80; ++i) hc[i] = 0.3 + 0.001 * i;
    CHECK(hipMemcpy(dc, hc.data(), 80 * 8, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(addchain, dim3(1), dim3(64), 0, 0, dout, dcyc, 1.0);
    CHECK(hipDeviceSynchronize());
    long long c0; CHECK(hipMemcpy(&c0, dcyc, 8, hipMemcpyDeviceToHost));
    printf("dependent v_add_f64: %.1f cycles each (one wave)\n", (double)c0 / 4096.0);
    for (int bpc : {1, 2, 3, 5}) { run<1>(dc, dout, dcyc, bpc); run<2>(dc, dout, dcyc, bpc); run<4>(dc, dout, dcyc, bpc); }
    for (int bpc : {1, 3, 5, 8}) { run<1, 0>(dc, dout, dcyc, bpc); run<1, 1>(dc, dout, dcyc, bpc); run<1, 2>(dc, dout, dcyc, bpc); }
    return 0;
}
