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
    }
    return 0;
}
