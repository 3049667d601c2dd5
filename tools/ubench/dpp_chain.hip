// Microbenchmark: sequential fp32 sum whose products come from the neighbouring lane through DPP (gfx950).
// Pattern under test (GRU B "pairs"): two lanes share one row; lane parity q holds the weights of inputs 2k+q;
// per 8 terms: 1 ds_read_b128 of the lane's 4 inputs, 2 v_pk_mul_f32, 8 dependent v_add_f32_dpp.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#ifdef USE_ASM
#define ADD_DPP_EVEN(acc, p) asm volatile("v_add_f32_dpp %0, %1, %0 quad_perm:[0,0,2,2] row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(p))
#define ADD_DPP_ODD(acc, p) asm volatile("v_add_f32_dpp %0, %1, %0 quad_perm:[1,1,3,3] row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(p))
#else
__device__ __forceinline__ float dpp_q(float p, bool odd)
{
    const int v = __builtin_bit_cast(int, p);
    const int r = odd ? __builtin_amdgcn_update_dpp(0, v, 0xF5, 0xf, 0xf, false) : __builtin_amdgcn_update_dpp(0, v, 0xA0, 0xf, 0xf, false);
    return __builtin_bit_cast(float, r);
}
#define ADD_DPP_EVEN(acc, p) acc = dpp_q(p, false) + acc
#define ADD_DPP_ODD(acc, p) acc = dpp_q(p, true) + acc
#endif

template <int MODE>
__global__ void __launch_bounds__(512) k(float *out, unsigned long long *cyc, const float *win, float seed)
{
    __shared__ __attribute__((aligned(16))) float xs[2][192];
    const int lane = threadIdx.x & 63, q = lane & 1;
    for (int i = threadIdx.x; i < 384; i += blockDim.x) xs[i & 1][i >> 1] = seed * 0.001f * i;
    f32x2 W[96];
#pragma unroll
    for (int j = 0; j < 96; ++j) { W[j].x = win[(4 * j + q) * 64 + lane]; W[j].y = win[(4 * j + 2 + q) * 64 + lane]; }
    __syncthreads();
    float acc = seed;
    unsigned long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
    for (int it = 0; it < 16; ++it) {
        if (MODE == 0) {          // 32 dependent dpp adds
            float p = seed;
#pragma unroll
            for (int i = 0; i < 16; ++i) { ADD_DPP_EVEN(acc, p); ADD_DPP_ODD(acc, p); }
        } else {                  // full 384-term chain
            const float *xq = xs[q];
            f32x4 xa = *reinterpret_cast<const f32x4 *>(xq), xb;
#pragma unroll
            for (int n = 0; n < 48; n += 2) {
                xb = *reinterpret_cast<const f32x4 *>(xq + 4 * (n + 1));
                __builtin_amdgcn_sched_barrier(0);
                {
                    const f32x2 p0 = W[2 * n] * xa.lo, p1 = W[2 * n + 1] * xa.hi;
                    __builtin_amdgcn_sched_barrier(0);
                    ADD_DPP_EVEN(acc, p0.x); ADD_DPP_ODD(acc, p0.x); ADD_DPP_EVEN(acc, p0.y); ADD_DPP_ODD(acc, p0.y);
                    ADD_DPP_EVEN(acc, p1.x); ADD_DPP_ODD(acc, p1.x); ADD_DPP_EVEN(acc, p1.y); ADD_DPP_ODD(acc, p1.y);
                }
                __builtin_amdgcn_sched_barrier(0);
                if (n + 2 < 48) xa = *reinterpret_cast<const f32x4 *>(xq + 4 * (n + 2));
                __builtin_amdgcn_sched_barrier(0);
                {
                    const f32x2 p0 = W[2 * n + 2] * xb.lo, p1 = W[2 * n + 3] * xb.hi;
                    __builtin_amdgcn_sched_barrier(0);
                    ADD_DPP_EVEN(acc, p0.x); ADD_DPP_ODD(acc, p0.x); ADD_DPP_EVEN(acc, p0.y); ADD_DPP_ODD(acc, p0.y);
                    ADD_DPP_EVEN(acc, p1.x); ADD_DPP_ODD(acc, p1.x); ADD_DPP_EVEN(acc, p1.y); ADD_DPP_ODD(acc, p1.y);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
    if (lane == 0) cyc[threadIdx.x / 64] = t1 - t0;
}

int main()
{
    float *out, *win; unsigned long long *cyc;
    hipMalloc(&out, 4096 * 4); hipMalloc(&cyc, 64 * 8); hipMalloc(&win, 384 * 64 * 4);
    float *h = new float[384 * 64];
    for (int i = 0; i < 384 * 64; ++i) h[i] = 1.0f + 0.001f * (i % 97);
    hipMemcpy(win, h, 384 * 64 * 4, hipMemcpyHostToDevice);
    unsigned long long c[8];
    for (int waves : {1, 2, 8}) {
        hipLaunchKernelGGL(k<0>, dim3(1), dim3(64 * waves), 0, 0, out, cyc, win, 1.0f);
        hipLaunchKernelGGL(k<0>, dim3(1), dim3(64 * waves), 0, 0, out, cyc, win, 1.0f);
        hipDeviceSynchronize();
        hipMemcpy(c, cyc, sizeof(c), hipMemcpyDeviceToHost);
        printf("32 dependent v_add_f32_dpp, %d waves: %.2f cycles/instr\n", waves, (double)c[0] / (16.0 * 32));
        hipLaunchKernelGGL(k<1>, dim3(1), dim3(64 * waves), 0, 0, out, cyc, win, 1.0f);
        hipLaunchKernelGGL(k<1>, dim3(1), dim3(64 * waves), 0, 0, out, cyc, win, 1.0f);
        hipDeviceSynchronize();
        hipMemcpy(c, cyc, sizeof(c), hipMemcpyDeviceToHost);
        printf("384-term pair chain (ds_read_b128 + 2 pk_mul + 8 dpp adds per 8 terms), %d waves: %.1f cycles per chain = %.2f per term\n",
               waves, (double)c[0] / 16.0, (double)c[0] / 16.0 / 384.0);
    }
    return 0;
}
