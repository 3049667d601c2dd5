// What does one block of the pair kernel's dependent chains cost?  (development microbenchmark, gfx950)
//
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-slp-vectorize -o pk_chain pk_chain.hip && ./pk_chain
//
// One block = four inputs of one dependent chain for two utterances: 4 v_pk_add_f32 (dependent) interleaved with the
// 4 v_pk_mul_f32 (broadcast weight, op_sel) of the next block, as in csrc/lpcnet_sample_pair.hip DSS_PK_STEP4.
// Variants: registers only; + two ds_read_b128 per block (state pairs, broadcast address) fetched two blocks ahead;
// the same with the reads of two blocks issued together; the latency kernel's scalar form (2 v_pk_mul + 4 v_add per
// four inputs of ONE utterance, one ds_read_b128) for reference.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CHECK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { printf("%s: %s\n", #e, hipGetErrorString(_e)); return 1; } } while (0)

struct PairX { f32x4 a, c; };
#define STEP4(ACC, P, Q, X, WLO, WHI)                                                            \
    asm("v_pk_add_f32 %[a], %[a], %[p0]\n\t"                                                     \
        "v_pk_mul_f32 %[q0], %[wl], %[x0] op_sel_hi:[0,1]\n\t"                                   \
        "v_pk_add_f32 %[a], %[a], %[p1]\n\t"                                                     \
        "v_pk_mul_f32 %[q1], %[wl], %[x1] op_sel:[1,0] op_sel_hi:[1,1]\n\t"                      \
        "v_pk_add_f32 %[a], %[a], %[p2]\n\t"                                                     \
        "v_pk_mul_f32 %[q2], %[wh], %[x2] op_sel_hi:[0,1]\n\t"                                   \
        "v_pk_add_f32 %[a], %[a], %[p3]\n\t"                                                     \
        "v_pk_mul_f32 %[q3], %[wh], %[x3] op_sel:[1,0] op_sel_hi:[1,1]"                          \
        : [a] "+v"(ACC), [q0] "=&v"((Q)[0]), [q1] "=&v"((Q)[1]), [q2] "=&v"((Q)[2]), [q3] "=&v"((Q)[3]) \
        : [p0] "v"((P)[0]), [p1] "v"((P)[1]), [p2] "v"((P)[2]), [p3] "v"((P)[3]),                \
          [x0] "v"((X).a.lo), [x1] "v"((X).a.hi), [x2] "v"((X).c.lo), [x3] "v"((X).c.hi), [wl] "v"(WLO), [wh] "v"(WHI))
// the same without op_sel (is the broadcast itself slow?)
#define STEP4N(ACC, P, Q, X, WLO, WHI)                                                           \
    asm("v_pk_add_f32 %[a], %[a], %[p0]\n\t"                                                     \
        "v_pk_mul_f32 %[q0], %[wl], %[x0]\n\t"                                                   \
        "v_pk_add_f32 %[a], %[a], %[p1]\n\t"                                                     \
        "v_pk_mul_f32 %[q1], %[wl], %[x1]\n\t"                                                   \
        "v_pk_add_f32 %[a], %[a], %[p2]\n\t"                                                     \
        "v_pk_mul_f32 %[q2], %[wh], %[x2]\n\t"                                                   \
        "v_pk_add_f32 %[a], %[a], %[p3]\n\t"                                                     \
        "v_pk_mul_f32 %[q3], %[wh], %[x3]"                                                       \
        : [a] "+v"(ACC), [q0] "=&v"((Q)[0]), [q1] "=&v"((Q)[1]), [q2] "=&v"((Q)[2]), [q3] "=&v"((Q)[3]) \
        : [p0] "v"((P)[0]), [p1] "v"((P)[1]), [p2] "v"((P)[2]), [p3] "v"((P)[3]),                \
          [x0] "v"((X).a.lo), [x1] "v"((X).a.hi), [x2] "v"((X).c.lo), [x3] "v"((X).c.hi), [wl] "v"(WLO), [wh] "v"(WHI))
// sums only / products only (what do the two halves cost alone?)
#define ADD4(ACC, P)                                                                             \
    asm("v_pk_add_f32 %[a], %[a], %[p0]\n\tv_pk_add_f32 %[a], %[a], %[p1]\n\tv_pk_add_f32 %[a], %[a], %[p2]\n\tv_pk_add_f32 %[a], %[a], %[p3]" \
        : [a] "+v"(ACC) : [p0] "v"((P)[0]), [p1] "v"((P)[1]), [p2] "v"((P)[2]), [p3] "v"((P)[3]))

#define NBLK 48
template <int V>
__global__ void __launch_bounds__(512) k(const float *w, float *out, long long *cyc, int reps)
{
    __shared__ __attribute__((aligned(16))) float st[2 * 400];
    for (int i = threadIdx.x; i < 800; i += blockDim.x) st[i] = 1e-3f * (i & 15);
    __syncthreads();
    const int lane = threadIdx.x & 63;
    f32x2 WB[2 * NBLK];
#pragma unroll
    for (int j = 0; j < 2 * NBLK; ++j) { WB[j].x = w[(2 * j) * 64 + lane]; WB[j].y = w[(2 * j + 1) * 64 + lane]; }
    f32x2 acc = {0.f, 0.f};
    float sacc = 0.f;
    const char *an = reinterpret_cast<const char *>(st);
    long long t0 = __builtin_readcyclecounter();
    for (int r = 0; r < reps; ++r) {
        if constexpr (V == 0 || V == 4 || V == 5) {                    // registers only
            PairX X; X.a = *reinterpret_cast<const f32x4 *>(an); X.c = *reinterpret_cast<const f32x4 *>(an + 16);
            f32x2 P[2][4] = {{X.a.lo, X.a.hi, X.c.lo, X.c.hi}, {X.a.lo, X.a.hi, X.c.lo, X.c.hi}};
#pragma unroll
            for (int g = 0; g < NBLK; ++g) {
                if (V == 0) STEP4(acc, P[g & 1], P[(g + 1) & 1], X, WB[2 * g], WB[2 * g + 1]);
                if (V == 4) STEP4N(acc, P[g & 1], P[(g + 1) & 1], X, WB[2 * g], WB[2 * g + 1]);
                if (V == 5) ADD4(acc, P[g & 1]);
                __builtin_amdgcn_sched_barrier(0);
            }
        } else if constexpr (V == 1 || V == 2) {                       // + LDS reads two blocks ahead
            PairX GX[3];
            f32x2 GP[2][4];
#define LOADX(G) { GX[(G) % 3].a = *reinterpret_cast<const f32x4 *>(an + 32 * (G)); GX[(G) % 3].c = *reinterpret_cast<const f32x4 *>(an + 32 * (G) + 16); }
            LOADX(0) LOADX(1)
            GP[0][0] = GX[0].a.lo; GP[0][1] = GX[0].a.hi; GP[0][2] = GX[0].c.lo; GP[0][3] = GX[0].c.hi;
#pragma unroll
            for (int g = 0; g < NBLK; ++g) {
                if (V == 1) { if (g + 2 < NBLK) LOADX(g + 2) }
                else if ((g & 1) == 0) { if (g + 2 < NBLK) LOADX(g + 2) if (g + 3 < NBLK) { /* second block's reads with the first's */ } }
                __builtin_amdgcn_sched_barrier(0);
                STEP4(acc, GP[g & 1], GP[(g + 1) & 1], GX[(g + 1) % 3], WB[2 * g], WB[2 * g + 1]);
                __builtin_amdgcn_sched_barrier(0);
                if (V == 2 && (g & 1) == 1 && g + 2 < NBLK) LOADX(g + 2)
            }
        } else {                                                       // V == 3: the latency kernel's scalar form, one utterance
            f32x4 A[2];
            A[0] = *reinterpret_cast<const f32x4 *>(an);
#pragma unroll
            for (int g = 0; g < NBLK; ++g) {
                if (g + 1 < NBLK) A[(g + 1) & 1] = *reinterpret_cast<const f32x4 *>(an + 16 * (g + 1));
                __builtin_amdgcn_sched_barrier(0);
                const f32x2 p0 = WB[2 * g] * A[g & 1].lo, p1 = WB[2 * g + 1] * A[g & 1].hi;
                sacc += p0.x; sacc += p0.y; sacc += p1.x; sacc += p1.y;
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    long long t1 = __builtin_readcyclecounter();
    if (lane == 0) cyc[blockIdx.x * 8 + (threadIdx.x >> 6)] = t1 - t0;
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc.x + acc.y + sacc;
}

// ---- "two rows of a block per lane" (a possible next form of the latency kernel's h chain) against the present
// form, both with their real LDS traffic: block records in LDS, state vector in LDS, 16 resp. 8 row groups per wave.
//   R = 1  rows form: lane = rows q, q+4 of one of 16 groups; per block 2 ds_read_b128 of weights ([W0 q=0..3][W1 q=0..3], groups of a
//          16-lane pass 64 bytes apart), 1 ds_read_b128 of the state, 4 v_pk_mul_f32 (state broadcast with op_sel) + 4 v_pk_add_f32
//   R = 0  present form: lane = one row of one of 8 groups; per block 1 ds_read_b128 of weights, 1 of the state, 2 v_pk_mul_f32 + 4 v_add_f32
#define RSTEP4(ACC, P, Q, X, W0, W1)                                                             \
    asm("v_pk_add_f32 %[a], %[a], %[p0]\n\t"                                                     \
        "v_pk_mul_f32 %[q0], %[w0l], %[xl] op_sel:[0,0] op_sel_hi:[1,0]\n\t"                     \
        "v_pk_add_f32 %[a], %[a], %[p1]\n\t"                                                     \
        "v_pk_mul_f32 %[q1], %[w0h], %[xl] op_sel:[0,1] op_sel_hi:[1,1]\n\t"                     \
        "v_pk_add_f32 %[a], %[a], %[p2]\n\t"                                                     \
        "v_pk_mul_f32 %[q2], %[w1l], %[xh] op_sel:[0,0] op_sel_hi:[1,0]\n\t"                     \
        "v_pk_add_f32 %[a], %[a], %[p3]\n\t"                                                     \
        "v_pk_mul_f32 %[q3], %[w1h], %[xh] op_sel:[0,1] op_sel_hi:[1,1]"                         \
        : [a] "+v"(ACC), [q0] "=&v"((Q)[0]), [q1] "=&v"((Q)[1]), [q2] "=&v"((Q)[2]), [q3] "=&v"((Q)[3]) \
        : [p0] "v"((P)[0]), [p1] "v"((P)[1]), [p2] "v"((P)[2]), [p3] "v"((P)[3]),                \
          [xl] "v"((X).lo), [xh] "v"((X).hi), [w0l] "v"((W0).lo), [w0h] "v"((W0).hi), [w1l] "v"((W1).lo), [w1h] "v"((W1).hi))
#define RB 24                                    // blocks per list
template <int R>
__global__ void __launch_bounds__(512) krows(float *out, long long *cyc, int reps)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float *st = lds;                                              // 96 columns x 4 floats
    char *img = reinterpret_cast<char *>(lds + 400);              // 16 groups x (RB records of 128 B + 64 B)
    const int gstride = RB * 128 + 64;
    for (int i = threadIdx.x; i < 400 + 16 * gstride / 4; i += blockDim.x) lds[i] = 1e-3f * (i & 15);
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int g = R ? lane >> 2 : lane >> 3;
    const char *hw = img + g * gstride + (R ? (lane & 3) * 16 : (lane & 7) * 16);
    unsigned colseed = 17u * g + 5u;
    f32x2 acc = {0.f, 0.f};
    float sacc = 0.f;
    const char *xb = reinterpret_cast<const char *>(st);
    long long t0 = __builtin_readcyclecounter();
    for (int r = 0; r < reps; ++r) {
        asm volatile("" : "+v"(colseed));
        if constexpr (R == 1) {
            f32x4 W0[3], W1[3], X[3];
            f32x2 P[2][4];
#define RLOAD(S) { W0[(S) % 3] = *reinterpret_cast<const f32x4 *>(hw + (S) * 128); W1[(S) % 3] = *reinterpret_cast<const f32x4 *>(hw + (S) * 128 + 64); \
                   X[(S) % 3] = *reinterpret_cast<const f32x4 *>(xb + ((colseed + 7u * (S)) % 96u) * 16); }
            RLOAD(0) RLOAD(1)
            P[0][0] = X[0].lo; P[0][1] = X[0].hi; P[0][2] = X[0].lo; P[0][3] = X[0].hi;
#pragma unroll
            for (int s = 0; s < RB; ++s) {
                if (s + 2 < RB) RLOAD(s + 2)
                __builtin_amdgcn_sched_barrier(0);
                RSTEP4(acc, P[s & 1], P[(s + 1) & 1], X[(s + 1) % 3], W0[(s + 1) % 3], W1[(s + 1) % 3]);
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
            f32x4 HA[4], HB[4];
#define HLOAD(Q, C) { Q[0] = *reinterpret_cast<const f32x4 *>(hw + (2 * (C)) * 128); Q[1] = *reinterpret_cast<const f32x4 *>(hw + (2 * (C) + 1) * 128); \
                      Q[2] = *reinterpret_cast<const f32x4 *>(xb + ((colseed + 14u * (C)) % 96u) * 16); Q[3] = *reinterpret_cast<const f32x4 *>(xb + ((colseed + 14u * (C) + 7u) % 96u) * 16); }
#define HMAC(Q) { const f32x2 p0 = Q[0].lo * Q[2].lo, p1 = Q[0].hi * Q[2].hi, p2 = Q[1].lo * Q[3].lo, p3 = Q[1].hi * Q[3].hi; \
                  sacc += p0.x; sacc += p0.y; sacc += p1.x; sacc += p1.y; sacc += p2.x; sacc += p2.y; sacc += p3.x; sacc += p3.y; }
            HLOAD(HA, 0)
#pragma unroll
            for (int c = 0; c < RB / 2; c += 2) {
                HLOAD(HB, c + 1)
                __builtin_amdgcn_sched_barrier(0);
                HMAC(HA)
                __builtin_amdgcn_sched_barrier(0);
                if (c + 2 < RB / 2) HLOAD(HA, c + 2)
                __builtin_amdgcn_sched_barrier(0);
                HMAC(HB)
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    long long t1 = __builtin_readcyclecounter();
    if (lane == 0) cyc[blockIdx.x * 8 + (threadIdx.x >> 6)] = t1 - t0;
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc.x + acc.y + sacc;
}

template <int R>
static int runrows(const char *name, int waves, float *dout, long long *dcyc)
{
    const int reps = 200;
    const size_t dyn = (400 + 16 * (RB * 128 + 64) / 4) * 4;
    CHECK(hipFuncSetAttribute((const void *)krows<R>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn));
    for (int it = 0; it < 2; ++it) {
        hipLaunchKernelGGL(krows<R>, dim3(256), dim3(64 * waves), dyn, 0, dout, dcyc, reps);
        CHECK(hipDeviceSynchronize());
    }
    std::vector<long long> c(256 * 8);
    CHECK(hipMemcpy(c.data(), dcyc, c.size() * 8, hipMemcpyDeviceToHost));
    double s = 0; int n = 0;
    for (int b = 0; b < 256; ++b) for (int wv = 0; wv < waves; ++wv) { s += (double)c[b * 8 + wv]; ++n; }
    const double per_block = s / n / reps / RB;
    printf("%-64s %d waves/WG: %7.1f cycles per block step of a wave = %5.2f per row-block of the CU's chain work\n", name, waves, per_block,
           per_block / (R ? 2.0 : 1.0));
    return 0;
}

template <int V>
static int run(const char *name, int waves, const float *dw, float *dout, long long *dcyc)
{
    const int reps = 200;
    hipLaunchKernelGGL(k<V>, dim3(256), dim3(64 * waves), 0, 0, dw, dout, dcyc, reps);
    CHECK(hipDeviceSynchronize());
    hipLaunchKernelGGL(k<V>, dim3(256), dim3(64 * waves), 0, 0, dw, dout, dcyc, reps);
    CHECK(hipDeviceSynchronize());
    std::vector<long long> c(256 * 8);
    CHECK(hipMemcpy(c.data(), dcyc, c.size() * 8, hipMemcpyDeviceToHost));
    double s = 0; int n = 0;
    for (int b = 0; b < 256; ++b) for (int wv = 0; wv < waves; ++wv) { s += (double)c[b * 8 + wv]; ++n; }
    printf("%-64s %d waves/WG: %7.1f cycles per block of 4 inputs\n", name, waves, s / n / reps / NBLK);
    return 0;
}

int main()
{
    float *dw, *dout; long long *dcyc;
    CHECK(hipMalloc(&dw, 4 * NBLK * 64 * 4 + 1024)); CHECK(hipMalloc(&dout, 256 * 512 * 4)); CHECK(hipMalloc(&dcyc, 256 * 8 * 8));
    std::vector<float> hw(4 * NBLK * 64, 1.0001f);
    CHECK(hipMemcpy(dw, hw.data(), hw.size() * 4, hipMemcpyHostToDevice));
    for (int waves : {1, 4, 8}) {
        run<0>("pair step, registers only (4 dependent pk_add + 4 pk_mul op_sel)", waves, dw, dout, dcyc);
        run<4>("pair step, registers only, no op_sel", waves, dw, dout, dcyc);
        run<5>("4 dependent pk_add only", waves, dw, dout, dcyc);
        run<1>("pair step + 2 ds_read_b128 per block, two blocks ahead", waves, dw, dout, dcyc);
        run<2>("pair step + 2 ds_read_b128 per block, issued after the step", waves, dw, dout, dcyc);
        run<3>("latency-kernel form: 2 pk_mul + 4 add + 1 ds_read_b128 (one utt)", waves, dw, dout, dcyc);
        runrows<0>("h chain as it is: 1 row per lane, weights and state from LDS", waves, dout, dcyc);
        runrows<1>("h chain, rows q and q+4 per lane (packed), weights and state from LDS", waves, dout, dcyc);
    }
    return 0;
}
