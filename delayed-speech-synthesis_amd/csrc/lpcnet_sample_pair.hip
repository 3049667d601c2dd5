// csrc/lpcnet_sample_pair.hip -- LPCNet sample-rate network, throughput form: TWO utterances per persistent
// workgroup, carried as the two halves of packed fp32 instructions (gfx950).
//
// Same algorithm, same roles and barriers, same weight placement as lpcnet_sample.hip (xiph/LPCNet src/lpcnet.c
// lpcnet_synthesize_tail_impl() + run_sample_network(), src/nnet.c compute_gru_a_input / compute_sparse_gru / compute_gruB /
// sample_mdense, generic float path of src/vec.h; reached through extensions/lpcnet/cLPCNet.pxd:13 and, in bulk, through
// local/training.py:165-207).  What changes is the unit of work of an instruction: where the latency kernel adds one
// product of utterance A to one running sum, this kernel adds the products of utterances A and B to their two running
// sums with one v_pk_add_f32, and forms them with one v_pk_mul_f32 whose weight operand is broadcast to both halves
// (op_sel / op_sel_hi on src0).  Each half of a packed instruction is an ordinary IEEE fp32 operation, rounded on its
// own, and every chain keeps the C source's order (one product at a time, ascending input; -ffp-contract=off), so both
// utterances come out bit-identical to the scalar C path -- and to the latency kernel, whose decoder state they share.
//
// Why it pays: the CU-resident weights (VGPRs, LDS) are read once for two utterances, and a lone wave issues a packed
// instruction as fast as a plain one (tools/ubench/simd_share.hip: 4.9 cycles per instruction with two chains).  The
// state vectors live in LDS as (A, B) pairs, so one ds_read_b128 feeds two inputs of both utterances.  The GRU A chains
// multiply as they go (two utterances' products do not fit beside the weights); GRU B's one long chain is split into
// seven stages that ping-pong between the two relay waves, each wave forming its next stage's products while the other
// runs its sums, with the weights streamed from L2 (see DSS_PR_MUL).  The dual-FC waves stream their node's 32 weights
// the same way, once per sample: between barriers D and C they have no register to spare.
//
// The compiler does not fold a broadcast into op_sel (it copies the weight into a register pair instead, which would
// double the weight registers), so the products are inline asm; the sum chains whose order of issue matters (one
// dependent chain per lane: h gate, GRU B) are fixed-order asm blocks of one 8x4 block each.
#include <mutex>

#include "lpcnet_sample_common.h"
#undef HC

#define DSS_PAIR_HBLK_BYTES 144896        // dynamic LDS left beside PairLds (160 KB per CU)
// Ping-pong relay: the 96 blocks of four inputs go through stages 0..6, even stages on wave 7 (PR7 blocks each), odd
// ones on wave 6 (PR6 blocks each).  A wave forms the products of its next stage while the other wave runs its sums.
#ifndef PR7
#define PR7 12
#endif
#define PR6 ((NA / 4 - 4 * PR7) / 3)
static_assert(4 * PR7 + 3 * PR6 == NA / 4, "stages cover the 96 blocks");
#define DSS_PR_G0(S) ((((S) + 1) / 2) * PR7 + ((S) / 2) * PR6)      // first block of stage S

struct PairLds {
    float state_a[2][2 * (NA + 4)];       // double-buffered GRU A state, [unit][utterance]; "column 96" = eight zeros
    float gb_wrec[NB * NB3];              // GRU B recurrent weights [16][48]
    float tansig[208];
    float ulaw2lin[256];
    float spec_tab_pred[2][256];          // speculation over all 256 excitation values, per utterance:
    unsigned short spec_tab_idx[2][256];  //   next sample's prediction and its two mu-law indices (su | pu << 8)
    float spec_prod[DSS_LPC_ORDER][2];    // inputs of the speculation, published by wave 7: [0] = lpc[0] of the next sample's
                                          //   frame, [j] = history[j-1] * lpc[j] (the products are the same for every candidate)
    float spec_pred[2];                   //   and this sample's prediction
    float pad1[2];
    float gb_acc[64][2];                  // GRU B partial sums handed from wave 6 to wave 7
    float ah[NA][2];                      // h-gate pre-activation: written by a unit's h lane, read by its z/r lane
    float state_b[NB][2];
    float thr[2][8];
    unsigned bits[2][8];                  // decision bit of every tree node (256 bits) per utterance
    int idx[2][4];                        // last_sig_ulaw, pred_ulaw, last_exc (first sample of a call)
    int gb_flag;                          // sequence number of the sample whose gb_acc is valid
    int pad[3];
    short pcm[2][DSS_FRAME_SIZE];
};
static_assert(sizeof(PairLds) % 16 == 0, "dynamic LDS must start 16-byte aligned");
static_assert(sizeof(PairLds) + DSS_PAIR_HBLK_BYTES <= 160 * 1024, "LDS budget");

// One 8x4 block of the z gate and one of the r gate: two independent chains, so the sums of one hide the latency of the
// other and the products need no pipelining across blocks.  T[0..3] z products, T[4..7] r products (temporaries).
#define DSS_PK_ZR_MUL(T, XZ, XR, WZLO, WZHI, WRLO, WRHI)                                         \
    asm("v_pk_mul_f32 %[z0], %[wzl], %[xz0] op_sel_hi:[0,1]\n\t"                                 \
        "v_pk_mul_f32 %[r0], %[wrl], %[xr0] op_sel_hi:[0,1]\n\t"                                 \
        "v_pk_mul_f32 %[z1], %[wzl], %[xz1] op_sel:[1,0] op_sel_hi:[1,1]\n\t"                    \
        "v_pk_mul_f32 %[r1], %[wrl], %[xr1] op_sel:[1,0] op_sel_hi:[1,1]\n\t"                    \
        "v_pk_mul_f32 %[z2], %[wzh], %[xz2] op_sel_hi:[0,1]\n\t"                                 \
        "v_pk_mul_f32 %[r2], %[wrh], %[xr2] op_sel_hi:[0,1]\n\t"                                 \
        "v_pk_mul_f32 %[z3], %[wzh], %[xz3] op_sel:[1,0] op_sel_hi:[1,1]\n\t"                    \
        "v_pk_mul_f32 %[r3], %[wrh], %[xr3] op_sel:[1,0] op_sel_hi:[1,1]"                        \
        : [z0] "=&v"((T)[0]), [z1] "=&v"((T)[1]), [z2] "=&v"((T)[2]), [z3] "=&v"((T)[3]),        \
          [r0] "=&v"((T)[4]), [r1] "=&v"((T)[5]), [r2] "=&v"((T)[6]), [r3] "=&v"((T)[7])         \
        : [xz0] "v"((XZ).a.lo), [xz1] "v"((XZ).a.hi), [xz2] "v"((XZ).c.lo), [xz3] "v"((XZ).c.hi), \
          [xr0] "v"((XR).a.lo), [xr1] "v"((XR).a.hi), [xr2] "v"((XR).c.lo), [xr3] "v"((XR).c.hi), \
          [wzl] "v"(WZLO), [wzh] "v"(WZHI), [wrl] "v"(WRLO), [wrh] "v"(WRHI))
#define DSS_PK_ZR_ADD(AZ, AR, T)                                                                 \
    asm("v_pk_add_f32 %[az], %[az], %[z0]\n\t"                                                   \
        "v_pk_add_f32 %[ar], %[ar], %[r0]\n\t"                                                   \
        "v_pk_add_f32 %[az], %[az], %[z1]\n\t"                                                   \
        "v_pk_add_f32 %[ar], %[ar], %[r1]\n\t"                                                   \
        "v_pk_add_f32 %[az], %[az], %[z2]\n\t"                                                   \
        "v_pk_add_f32 %[ar], %[ar], %[r2]\n\t"                                                   \
        "v_pk_add_f32 %[az], %[az], %[z3]\n\t"                                                   \
        "v_pk_add_f32 %[ar], %[ar], %[r3]"                                                       \
        : [az] "+v"(AZ), [ar] "+v"(AR)                                                           \
        : [z0] "v"((T)[0]), [z1] "v"((T)[1]), [z2] "v"((T)[2]), [z3] "v"((T)[3]),                \
          [r0] "v"((T)[4]), [r1] "v"((T)[5]), [r2] "v"((T)[6]), [r3] "v"((T)[7]))
#define DSS_PK_ZR4(AZ, AR, T, XZ, XR, WZLO, WZHI, WRLO, WRHI)                                    \
    asm("v_pk_mul_f32 %[z0], %[wzl], %[xz0] op_sel_hi:[0,1]\n\t"                                 \
        "v_pk_mul_f32 %[r0], %[wrl], %[xr0] op_sel_hi:[0,1]\n\t"                                 \
        "v_pk_mul_f32 %[z1], %[wzl], %[xz1] op_sel:[1,0] op_sel_hi:[1,1]\n\t"                    \
        "v_pk_mul_f32 %[r1], %[wrl], %[xr1] op_sel:[1,0] op_sel_hi:[1,1]\n\t"                    \
        "v_pk_add_f32 %[az], %[az], %[z0]\n\t"                                                   \
        "v_pk_add_f32 %[ar], %[ar], %[r0]\n\t"                                                   \
        "v_pk_mul_f32 %[z2], %[wzh], %[xz2] op_sel_hi:[0,1]\n\t"                                 \
        "v_pk_mul_f32 %[r2], %[wrh], %[xr2] op_sel_hi:[0,1]\n\t"                                 \
        "v_pk_add_f32 %[az], %[az], %[z1]\n\t"                                                   \
        "v_pk_add_f32 %[ar], %[ar], %[r1]\n\t"                                                   \
        "v_pk_mul_f32 %[z3], %[wzh], %[xz3] op_sel:[1,0] op_sel_hi:[1,1]\n\t"                    \
        "v_pk_mul_f32 %[r3], %[wrh], %[xr3] op_sel:[1,0] op_sel_hi:[1,1]\n\t"                    \
        "v_pk_add_f32 %[az], %[az], %[z2]\n\t"                                                   \
        "v_pk_add_f32 %[ar], %[ar], %[r2]\n\t"                                                   \
        "v_pk_add_f32 %[az], %[az], %[z3]\n\t"                                                   \
        "v_pk_add_f32 %[ar], %[ar], %[r3]"                                                       \
        : [az] "+v"(AZ), [ar] "+v"(AR),                                                          \
          [z0] "=&v"((T)[0]), [z1] "=&v"((T)[1]), [z2] "=&v"((T)[2]), [z3] "=&v"((T)[3]),        \
          [r0] "=&v"((T)[4]), [r1] "=&v"((T)[5]), [r2] "=&v"((T)[6]), [r3] "=&v"((T)[7])         \
        : [xz0] "v"((XZ).a.lo), [xz1] "v"((XZ).a.hi), [xz2] "v"((XZ).c.lo), [xz3] "v"((XZ).c.hi), \
          [xr0] "v"((XR).a.lo), [xr1] "v"((XR).a.hi), [xr2] "v"((XR).c.lo), [xr3] "v"((XR).c.hi), \
          [wzl] "v"(WZLO), [wzh] "v"(WZHI), [wrl] "v"(WRLO), [wrh] "v"(WRHI))
// One block of four inputs of the two dual-FC layers for both utterances: S0 += w0 * (bA, bB), S1 += w1 * (bA, bB)
#define DSS_PK_FC4(S0, S1, T, U, W0, W1, W2, W3)                                                 \
    asm("v_pk_mul_f32 %[t0], %[w0], %[u0] op_sel_hi:[0,1]\n\t"                                   \
        "v_pk_mul_f32 %[v0], %[w0], %[u0] op_sel:[1,0] op_sel_hi:[1,1]\n\t"                      \
        "v_pk_mul_f32 %[t1], %[w1], %[u1] op_sel_hi:[0,1]\n\t"                                   \
        "v_pk_mul_f32 %[v1], %[w1], %[u1] op_sel:[1,0] op_sel_hi:[1,1]\n\t"                      \
        "v_pk_add_f32 %[s0], %[s0], %[t0]\n\t"                                                   \
        "v_pk_add_f32 %[s1], %[s1], %[v0]\n\t"                                                   \
        "v_pk_mul_f32 %[t2], %[w2], %[u2] op_sel_hi:[0,1]\n\t"                                   \
        "v_pk_mul_f32 %[v2], %[w2], %[u2] op_sel:[1,0] op_sel_hi:[1,1]\n\t"                      \
        "v_pk_add_f32 %[s0], %[s0], %[t1]\n\t"                                                   \
        "v_pk_add_f32 %[s1], %[s1], %[v1]\n\t"                                                   \
        "v_pk_mul_f32 %[t3], %[w3], %[u3] op_sel_hi:[0,1]\n\t"                                   \
        "v_pk_mul_f32 %[v3], %[w3], %[u3] op_sel:[1,0] op_sel_hi:[1,1]\n\t"                      \
        "v_pk_add_f32 %[s0], %[s0], %[t2]\n\t"                                                   \
        "v_pk_add_f32 %[s1], %[s1], %[v2]\n\t"                                                   \
        "v_pk_add_f32 %[s0], %[s0], %[t3]\n\t"                                                   \
        "v_pk_add_f32 %[s1], %[s1], %[v3]"                                                       \
        : [s0] "+v"(S0), [s1] "+v"(S1),                                                          \
          [t0] "=&v"((T)[0]), [t1] "=&v"((T)[1]), [t2] "=&v"((T)[2]), [t3] "=&v"((T)[3]),        \
          [v0] "=&v"((T)[4]), [v1] "=&v"((T)[5]), [v2] "=&v"((T)[6]), [v3] "=&v"((T)[7])         \
        : [u0] "v"((U).a.lo), [u1] "v"((U).a.hi), [u2] "v"((U).c.lo), [u3] "v"((U).c.hi),        \
          [w0] "v"(W0), [w1] "v"(W1), [w2] "v"(W2), [w3] "v"(W3))

// vec.h tanh_approx on an (A, B) pair: per half exactly the operations of dss_tanh_approx, in its order
__device__ __forceinline__ f32x2 dss_tanh_pk(const float *tab, f32x2 x)
{
    const bool n0 = x.x < 0, n1 = x.y < 0;
    f32x2 a = {n0 ? -x.x : x.x, n1 ? -x.y : x.y};
    const f32x2 sg = {n0 ? -1.f : 1.f, n1 ? -1.f : 1.f};
    const f32x2 t = .5f + 25 * a;
    int i0 = (int)floorf(t.x), i1 = (int)floorf(t.y);
    i0 = i0 < 0 ? 0 : i0; i0 = i0 > 200 ? 200 : i0;
    i1 = i1 < 0 ? 0 : i1; i1 = i1 > 200 ? 200 : i1;
    const f32x2 y = {tab[i0], tab[i1]};
    a -= .04f * (f32x2){(float)i0, (float)i1};
    const f32x2 dy = 1 - y * y;
    const f32x2 r = y + a * dy * (1 - y * a);
    return sg * r;
}
__device__ __forceinline__ f32x2 dss_sigmoid_pk(const float *tab, f32x2 x) { return .5f + .5f * dss_tanh_pk(tab, .5f * x); }
// two independent pairs, all four table reads issued before any is consumed (same arithmetic per element)
__device__ __forceinline__ void dss_tanh_pk2(const float *tab, f32x2 x, f32x2 w, f32x2 &ox, f32x2 &ow)
{
    const bool n0 = x.x < 0, n1 = x.y < 0, n2 = w.x < 0, n3 = w.y < 0;
    f32x2 a = {n0 ? -x.x : x.x, n1 ? -x.y : x.y}, c = {n2 ? -w.x : w.x, n3 ? -w.y : w.y};
    const f32x2 sa = {n0 ? -1.f : 1.f, n1 ? -1.f : 1.f}, sc = {n2 ? -1.f : 1.f, n3 ? -1.f : 1.f};
    const f32x2 ta = .5f + 25 * a, tc = .5f + 25 * c;
    int i0 = (int)floorf(ta.x), i1 = (int)floorf(ta.y), i2 = (int)floorf(tc.x), i3 = (int)floorf(tc.y);
    i0 = i0 < 0 ? 0 : i0; i0 = i0 > 200 ? 200 : i0;
    i1 = i1 < 0 ? 0 : i1; i1 = i1 > 200 ? 200 : i1;
    i2 = i2 < 0 ? 0 : i2; i2 = i2 > 200 ? 200 : i2;
    i3 = i3 < 0 ? 0 : i3; i3 = i3 > 200 ? 200 : i3;
    const f32x2 ya = {tab[i0], tab[i1]}, yc = {tab[i2], tab[i3]};
    a -= .04f * (f32x2){(float)i0, (float)i1};
    c -= .04f * (f32x2){(float)i2, (float)i3};
    const f32x2 da = 1 - ya * ya, dc = 1 - yc * yc;
    const f32x2 ra = ya + a * da * (1 - ya * a), rc = yc + c * dc * (1 - yc * c);
    ox = sa * ra;
    ow = sc * rc;
}
__device__ __forceinline__ void dss_sigmoid_pk2(const float *tab, f32x2 x, f32x2 w, f32x2 &ox, f32x2 &ow)
{
    f32x2 tx, tw;
    dss_tanh_pk2(tab, .5f * x, .5f * w, tx, tw);
    ox = .5f + .5f * tx;
    ow = .5f + .5f * tw;
}

// Speculation for one candidate excitation value and both utterances: the output sample, the next LPC prediction (the
// reference's 16 sequential subtractions; the products history[j-1] * lpc[j] do not depend on the candidate and come
// from wave 7) and both mu-law indices.  Same expressions, same order as lpcnet_synthesize_tail_impl().
__device__ __forceinline__ void dss_pair_speculate(PairLds &L, int cand, float u2l_c)
{
    const f32x2 sp = *reinterpret_cast<const f32x2 *>(L.spec_pred);
    const f32x2 pcm_c = sp + u2l_c;
    f32x4 q = *reinterpret_cast<const f32x4 *>(&L.spec_prod[0][0]);
    f32x2 pc = 0.f - pcm_c * q.lo;
    pc -= q.hi;
#pragma unroll
    for (int k = 1; k < DSS_LPC_ORDER / 2; ++k) {
        q = *reinterpret_cast<const f32x4 *>(&L.spec_prod[2 * k][0]);
        pc -= q.lo;
        pc -= q.hi;
    }
    // two evaluations at a time: all four interleaved cost the dual-FC waves registers they do not have
    const int su0 = dss_lin2ulaw(pcm_c.x), su1 = dss_lin2ulaw(pcm_c.y);
    __builtin_amdgcn_sched_barrier(0);
    const int pu0 = dss_lin2ulaw(pc.x), pu1 = dss_lin2ulaw(pc.y);
    L.spec_tab_pred[0][cand] = pc.x;
    L.spec_tab_pred[1][cand] = pc.y;
    L.spec_tab_idx[0][cand] = (unsigned short)(su0 | (pu0 << 8));
    L.spec_tab_idx[1][cand] = (unsigned short)(su1 | (pu1 << 8));
}

// what a workgroup runs: rows ua and ub of the call (frames, PCM), continuing decoder slots sa and sb, frames [f0, f1)
// (ub == ua and !has_b: a single utterance, the B halves compute a copy that is never stored)
struct PairJob { int ua, ub, sa, sb; bool has_b; int fc0, f0, f1; };
// The two utterances of a workgroup run as one packed job when both exist and have the same number of silent frames
// ahead of them (frame_count < FEATURES_DELAY: fresh decoders); otherwise one after the other, each with its copy in
// the B halves.  Ragged calls (RAGGED instantiation: rows name their decoder slot and frame count): the packed job covers
// the frames both rows have, the longer row finishes alone in a second job, which picks its decoder state up from
// global memory exactly as the next call would.  All fields are wave-uniform.
struct PairPlan { int u0, u1, fa, fb, n_jobs; bool packed; int sa, sb, nfa, nfb; };
template <bool RAGGED>
__device__ __forceinline__ PairJob dss_pair_job(const PairPlan &p, int jn, int n_frames)
{
    if constexpr (!RAGGED) {
        if (p.n_jobs == 1) return PairJob{p.u0, p.packed ? p.u0 + 1 : p.u0, p.u0, p.packed ? p.u0 + 1 : p.u0, p.packed, p.fa, 0, n_frames};
        return jn == 0 ? PairJob{p.u0, p.u0, p.u0, p.u0, false, p.fa, 0, n_frames}
                       : PairJob{p.u0 + 1, p.u0 + 1, p.u0 + 1, p.u0 + 1, false, p.fb, 0, n_frames};
    } else {
        const PairJob only_a = {p.u0, p.u0, p.sa, p.sa, false, p.fa, 0, p.nfa};
        const PairJob only_b = {p.u1, p.u1, p.sb, p.sb, false, p.fb, 0, p.nfb};
        if (!p.packed) return jn == 0 ? only_a : only_b;             // (a lone last row: n_jobs == 1)
        const int nmin = min(p.nfa, p.nfb);
        if (jn == 0) return PairJob{p.u0, p.u1, p.sa, p.sb, true, p.fa, 0, nmin};
        PairJob rest = p.nfa > p.nfb ? only_a : only_b;
        rest.f0 = nmin;
        return rest;
    }
}

// decoder state of the job's utterances into LDS (all 512 threads), followed by a barrier at the caller
__device__ __forceinline__ void dss_pair_job_init(PairLds &L, const DssBatchDev &b, const PairJob &j, int tid)
{
    if (tid < NA) {
        f32x2 s = {b.gru_a_state[(size_t)j.sa * NA + tid], b.gru_a_state[(size_t)j.sb * NA + tid]};
        *reinterpret_cast<f32x2 *>(&L.state_a[0][2 * tid]) = s;
    }
    if (tid < 16) L.state_a[tid >> 3][2 * NA + (tid & 7)] = 0.f;
    if (tid < NB) {
        f32x2 s = {b.gru_b_state[(size_t)j.sa * NB + tid], b.gru_b_state[(size_t)j.sb * NB + tid]};
        *reinterpret_cast<f32x2 *>(&L.state_b[tid][0]) = s;
    }
    if (tid == 0) L.gb_flag = 0;
}

// h-gate chain of one lane for both utterances: rbh + dgh*st, then its row group's blocks in idx order.  Block
// records (LDS image) and state pairs are fetched two blocks ahead of the block being summed.
#define DSS_PH_LOAD(S, BUF)                                                                      \
    {                                                                                            \
        HW[BUF] = *reinterpret_cast<const f32x4 *>(hw + (S) * 128);                              \
        dss_pair_loadx(HX[BUF], xbase + DSS_H_COL(S) * 32);                                      \
    }
// block S: sums of its products (in HP[S & 1]), products of block S+1 (into HP[(S + 1) & 1])
#define DSS_PH_STEP(S) DSS_PK_STEP4(ah, HP[(S) & 1], HP[((S) + 1) & 1], HX[((S) + 1) % 3], HW[((S) + 1) % 3].lo, HW[((S) + 1) % 3].hi)
#define DSS_PH_CHAIN(XBUF)                                                                       \
    {                                                                                            \
        const char *xbase = reinterpret_cast<const char *>(XBUF);                                \
        f32x4 HW[3];                                                                             \
        PairX HX[3];                                                                             \
        f32x2 HP[2][4];                                                                          \
        f32x2 ah = rbh2 + dgh2 * *reinterpret_cast<const f32x2 *>(xbase + uh * 8);               \
        DSS_PH_LOAD(0, 0)                                                                        \
        DSS_PH_LOAD(1, 1)                                                                        \
        DSS_PK_MUL4(HP[0], HX[0], HW[0].lo, HW[0].hi);                                           \
        /* nh is even: blocks s and s+1 exist whenever s < nh, one test per pair.  (No break: the asm blocks are      \
           convergent calls, and a loop with a data-dependent exit around them is not unrolled.) */                 \
        _Pragma("unroll") for (int s = 0; s < HC; s += 2) {                                      \
            if (s < nh) {                                                                        \
                const bool more = s + 2 < HC && s + 2 < nh;                                      \
                if (more) DSS_PH_LOAD(s + 2 < HC ? s + 2 : 0, (s + 2) % 3)                       \
                __builtin_amdgcn_sched_barrier(0);                                               \
                DSS_PH_STEP(s);                                                                  \
                __builtin_amdgcn_sched_barrier(0);                                               \
                if (more) {                                                                      \
                    DSS_PH_LOAD(s + 3 < HC ? s + 3 : 0, (s + 3) % 3)                             \
                    __builtin_amdgcn_sched_barrier(0);                                           \
                    DSS_PH_STEP(s + 1);                                                          \
                } else {                                                                         \
                    DSS_PK_ADD4(ah, HP[(s + 1) & 1]);                                            \
                }                                                                                \
                __builtin_amdgcn_sched_barrier(0);                                               \
            }                                                                                    \
        }                                                                                        \
        *reinterpret_cast<f32x2 *>(&L.ah[uh][0]) = ah;                                           \
    }

// ---- GRU B: a ping-pong relay -------------------------------------------------------------------------------------------
// One dependent chain of 384 sums per row and utterance.  The 96 blocks of four inputs go through seven stages that
// alternate between waves 7 and 6: while one wave runs the sums of its stage (v_pk_add_f32, (A, B) halves), the other
// forms the products of its next stage, so the chain itself is sums and hand-overs only.  The weights are not resident:
// every stage's 4 x PR floats per lane come from L2 (m.gb_w_quad) into the registers the previous stage's weights just left.
// Products of a stage's NBK blocks, first block G0, into PS[g][0..3]; the state pairs are read two blocks ahead.  The
// weights of block g sit in RW[g] (four inputs of this lane's row, loaded from m.gb_w_quad); once a block is multiplied
// its weight registers are reloaded with block g of the wave's NEXT stage (first block GN), which is multiplied a sum
// stage and a hand-over later: the L2 latency never shows.
#ifndef DSS_PRD
#define DSS_PRD 2                 // blocks of state pairs read ahead of the products
#endif
#define DSS_PR_MUL(NBK, G0, GN)                                                                  \
    {                                                                                            \
        PairX RX[DSS_PRD + 1];                                                                   \
        _Pragma("unroll") for (int u = 0; u < DSS_PRD; ++u) dss_pair_loadx(RX[u], an + 32 * ((G0) + u)); \
        _Pragma("unroll") for (int g = 0; g < (NBK); ++g) {                                      \
            if (g + DSS_PRD < (NBK)) dss_pair_loadx(RX[(g + DSS_PRD) % (DSS_PRD + 1)], an + 32 * ((G0) + g + DSS_PRD)); \
            __builtin_amdgcn_sched_barrier(0);                                                   \
            DSS_PK_MUL4(PS[g], RX[g % (DSS_PRD + 1)], RW[g].lo, RW[g].hi);                       \
            __builtin_amdgcn_sched_barrier(0);                                                   \
            RW[g] = *reinterpret_cast<const f32x4 *>(wq + (wo + (unsigned)((GN) + g) * 1024u));  \
        }                                                                                        \
        __builtin_amdgcn_sched_barrier(0);                                                       \
    }
// the sums of a stage: one dependent chain, input order
#define DSS_PR_ADD(NBK)                                                                          \
    {                                                                                            \
        _Pragma("unroll") for (int g = 0; g < (NBK); ++g) DSS_PK_ADD4(acc, PS[g]);               \
        __builtin_amdgcn_sched_barrier(0);                                                       \
    }
// hand the running sums to the other relay wave / take them over (gb_flag = 8 * sample number + stages done)
#define DSS_PR_PUBLISH(V)                                                                        \
    {                                                                                            \
        *reinterpret_cast<f32x2 *>(&L.gb_acc[lane][0]) = acc;                                    \
        __hip_atomic_store(&L.gb_flag, (V), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);     \
    }
#define DSS_PR_AWAIT(V)                                                                          \
    {                                                                                            \
        while (__hip_atomic_load(&L.gb_flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) != (V)) \
            ;                                    /* tight poll: one LDS round trip per iteration is pause enough */ \
        acc = *reinterpret_cast<const f32x2 *>(&L.gb_acc[lane][0]);                              \
    }

// wave 7: fold the sampled excitation into the signal history and emit the PCM sample (lpcnet_synthesize_tail_impl);
// lanes 0..15 carry utterance A, lanes 16..31 utterance B (lanes 32..63 repeat them and store nothing)
#define DSS_PS_UPDATE()                                                                          \
    {                                                                                            \
        float pcm = upd_pred + L.ulaw2lin[upd_exc];                                              \
        if (TRACE && owner) {                                                                    \
            const size_t o = ((size_t)my_utt * n_frames + f) * DSS_FRAME_SIZE + upd_i;           \
            b.trace_exc[o] = (float)upd_exc;                                                     \
            b.trace_pcm[o] = pcm;                                                                \
        }                                                                                        \
        ls_lane = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, pcm),   \
                                     __builtin_bit_cast(int, ls_lane), 0x111, 0xf, 0xf, false));  \
        last_exc = upd_exc;                                                                      \
        pcm += 0.85f * deemph;                                                                   \
        deemph = pcm;                                                                            \
        if (pcm < -32767) pcm = -32767;                                                          \
        if (pcm > 32767) pcm = 32767;                                                            \
        if (hl == 0 && lane < 32) L.pcm[hb][upd_i] = (short)(int)floor(.5 + (double)pcm);        \
        upd_pending = false;                                                                     \
    }

// =====================================================================================================
// role A: GRU A (+ dual-FC on waves 0..3, + the speculation on waves 0, 1, 5) for the two utterances of the job
// =====================================================================================================
template <bool TRACE, bool STAMP, int Z, bool HAS_FC, bool RAGGED>
__device__ __forceinline__ void dss_pair_role_a(PairLds &L, float *hblk_lds, const DssModelDev &m, const DssBatchDev &b,
                                                int n_frames, const PairPlan plan, int tid, int wave, int lane)
{
    constexpr int HC = DSS_HC;
    const int unit = m.unit_of[tid];                             // z/r chains + gates of this unit
    const int uh = m.unit_h[tid];                                // h-gate chain of this (other) unit
    const int nh = __builtin_amdgcn_readfirstlane(m.wave_nh[wave]);
    const int nzr = __builtin_amdgcn_readfirstlane(m.wave_nzr[wave]);
    const char *hw = reinterpret_cast<const char *>(hblk_lds + m.grp_hoff[tid >> 3]) + (lane & 7) * 16;
    f32x4 WZ[2 * ZRC];                                           // [0,ZRC) z slots, [ZRC,2ZRC) r slots
    unsigned PZ[(2 * ZRL + 3) / 4], PH[HC / 4];
#pragma unroll
    for (int s = 0; s < 2 * ZRC; ++s) {
        const int slot = s < ZRC ? s : ZRL + (s - ZRC);          // layout numbering
        WZ[s].x = m.zr_w[((size_t)slot * 4 + 0) * NA + tid];
        WZ[s].y = m.zr_w[((size_t)slot * 4 + 1) * NA + tid];
        WZ[s].z = m.zr_w[((size_t)slot * 4 + 2) * NA + tid];
        WZ[s].w = m.zr_w[((size_t)slot * 4 + 3) * NA + tid];
    }
#pragma unroll
    for (int s = 0; s < (2 * ZRL + 3) / 4; ++s) PZ[s] = m.zr_col[(size_t)s * NA + tid];
#pragma unroll
    for (int s = 0; s < HC / 4; ++s) PH[s] = m.h_col[(size_t)s * NA + tid];
    const float rbz = m.gru_a_rbias[unit], rbr = m.gru_a_rbias[NA + unit], rbh = m.gru_a_rbias[2 * NA + uh];
    const float dgz = m.gru_a_diag[unit], dgr = m.gru_a_diag[NA + unit], dgh = m.gru_a_diag[2 * NA + uh];
    const f32x2 rbh2 = {rbh, rbh}, dgh2 = {dgh, dgh};
    // dual-FC constants of tree node `tid` (waves 0..3).  The node's 32 weights are not kept: they are loaded once per
    // sample, after the h chain, into registers that are free by then (m.fc_w_pair: (layer 0, layer 1) weight of input j
    // side by side) -- these waves have no register to spare between barriers D and C.
    float fb0 = 0, fb1 = 0, ff0 = 0, ff1 = 0;
    if constexpr (HAS_FC) {
        const int node = tid;
        fb0 = m.fc_bias[node]; fb1 = m.fc_bias[DSS_FC_OUT + node];
        ff0 = m.fc_factor[node]; ff1 = m.fc_factor[DSS_FC_OUT + node];
    }
    // speculation: the quarter of the 256 candidates this wave covers between B and C, or -1 (quarter 0: wave 6)
    const int myq = wave == 1 ? 3 : wave == 4 ? 2 : wave == 5 ? 1 : -1;
    const float u2l_c = L.ulaw2lin[(myq * 64 + lane) & 255];     // this lane's excitation candidate
    const int level = 31 - __clz(tid | 1);                       // FC node = (1 << level) | prefix
    const bool recur_first = m.h.gru_a_order == DSS_GRUA_RECUR_FIRST;     // wave-uniform (kernel argument)
    unsigned long long sa[8] = {0, 0, 0, 0, 0, 0, 0, 0}, ta = 0;   // diagnostic build only

    for (int jn = 0; jn < plan.n_jobs; ++jn) {
        const PairJob job = dss_pair_job<RAGGED>(plan, jn, n_frames);
        const int ua = job.ua, ub = job.ub, fc0 = job.fc0;
        dss_pair_job_init(L, b, job, tid);
        __syncthreads();                                             // job barrier 0: decoder state in LDS
        int cur = 0;
        f32x2 st = *reinterpret_cast<const f32x2 *>(&L.state_a[0][2 * unit]);
        bool first_sample = true;
        DSS_PH_CHAIN(L.state_a[0])                                   // first sample of this call
        __syncthreads();                                             // job barrier 1: L.ah of every unit visible to its z/r lane

        for (int f = job.f0; f < job.f1; ++f) {
            if (fc0 + f < DSS_FEATURES_DELAY) continue;              // silent frame: decoder state untouched
            const float *foa = b.frame_out + ((size_t)ua * n_frames + f) * DSS_COND_STRIDE;     // wave-uniform bases
            const float *fob = b.frame_out + ((size_t)ub * n_frames + f) * DSS_COND_STRIDE;
            const f32x2 cz = {foa[(unsigned)unit], fob[(unsigned)unit]};
            const f32x2 cr = {foa[(unsigned)(NA + unit)], fob[(unsigned)(NA + unit)]};
            const f32x2 ch = {foa[(unsigned)(2 * NA + unit)], fob[(unsigned)(2 * NA + unit)]};
            for (int i = 0; i < DSS_FRAME_SIZE; ++i) {
                // waves 4, 5 carry the longest z/r lists and share their SIMDs with waves 0, 1: they go first between D and B
                if (!HAS_FC) __builtin_amdgcn_s_setprio(1);
                // keep the packed column ids opaque so the per-slot unpacking is not hoisted out of the sample loop
#pragma unroll
                for (int k = 0; k < (2 * ZRL + 3) / 4; ++k) asm volatile("" : "+v"(PZ[k]));
#pragma unroll
                for (int k = 0; k < HC / 4; ++k) asm volatile("" : "+v"(PH[k]));
                f32x2 az = rbz + dgz * st;                           // compute_sparse_gru, before the input term
                f32x2 ar = rbr + dgr * st;
                const f32x2 ahv = *reinterpret_cast<const f32x2 *>(&L.ah[unit][0]);   // h-gate pre-activation (its h lane, B..C)
                // The three embedding indices of either utterance.  First sample of a call: wave 7 computes them (L.idx,
                // barrier A).  Every later sample: this wave walks both sampling trees itself and looks the speculated
                // indices up.
                int sia, pia, eia, sib, pib, eib;
                if (first_sample) {
                    __syncthreads();                                                    // barrier A (first sample only)
                    sia = L.idx[0][0]; pia = L.idx[0][1]; eia = L.idx[0][2];
                    sib = L.idx[1][0]; pib = L.idx[1][1]; eib = L.idx[1][2];
                    first_sample = false;
                } else {
                    int ea_, eb_;
                    DSS_TREE_WALK_AT(ea_, L.bits[0])
                    DSS_TREE_WALK_AT(eb_, L.bits[1])
                    const unsigned xa = __builtin_amdgcn_readfirstlane((unsigned)L.spec_tab_idx[0][ea_]);
                    const unsigned xb_ = __builtin_amdgcn_readfirstlane((unsigned)L.spec_tab_idx[1][eb_]);
                    sia = (int)(xa & 0xFF); pia = (int)(xa >> 8); eia = ea_;
                    sib = (int)(xb_ & 0xFF); pib = (int)(xb_ >> 8); eib = eb_;
                }
                sia = __builtin_amdgcn_readfirstlane(sia); pia = __builtin_amdgcn_readfirstlane(pia); eia = __builtin_amdgcn_readfirstlane(eia);
                sib = __builtin_amdgcn_readfirstlane(sib); pib = __builtin_amdgcn_readfirstlane(pib); eib = __builtin_amdgcn_readfirstlane(eib);
                if (STAMP) ta = __builtin_readcyclecounter();
                typedef float f32x3 __attribute__((ext_vector_type(3)));
                const f32x3 esa = *reinterpret_cast<const f32x3 *>(m.embed_lane[0] + ((unsigned)sia * NA + (unsigned)tid) * 3);
                const f32x3 epa = *reinterpret_cast<const f32x3 *>(m.embed_lane[1] + ((unsigned)pia * NA + (unsigned)tid) * 3);
                const f32x3 eea = *reinterpret_cast<const f32x3 *>(m.embed_lane[2] + ((unsigned)eia * NA + (unsigned)tid) * 3);
                const f32x3 esb = *reinterpret_cast<const f32x3 *>(m.embed_lane[0] + ((unsigned)sib * NA + (unsigned)tid) * 3);
                const f32x3 epb = *reinterpret_cast<const f32x3 *>(m.embed_lane[1] + ((unsigned)pib * NA + (unsigned)tid) * 3);
                const f32x3 eeb = *reinterpret_cast<const f32x3 *>(m.embed_lane[2] + ((unsigned)eib * NA + (unsigned)tid) * 3);
                // The first two z/r blocks' products do not depend on the embedding rows: formed under their latency.
                const char *xb = reinterpret_cast<const char *>(L.state_a[cur]);
                PairX XZ[2], XR[2];
                f32x2 T0[8], T1[8], TT[8];
                dss_pair_loadx(XZ[0], xb + DSS_ZR_COL(0) * 32);
                dss_pair_loadx(XR[0], xb + DSS_ZR_COL(ZRL) * 32);
                dss_pair_loadx(XZ[1], xb + DSS_ZR_COL(1) * 32);
                dss_pair_loadx(XR[1], xb + DSS_ZR_COL(ZRL + 1) * 32);
                DSS_PK_ZR_MUL(T0, XZ[0], XR[0], WZ[0].lo, WZ[0].hi, WZ[ZRC].lo, WZ[ZRC].hi);
                if (2 < nzr) {
                    dss_pair_loadx(XZ[0], xb + DSS_ZR_COL(2) * 32);
                    dss_pair_loadx(XR[0], xb + DSS_ZR_COL(ZRL + 2) * 32);
                }
                DSS_PK_ZR_MUL(T1, XZ[1], XR[1], WZ[1].lo, WZ[1].hi, WZ[ZRC + 1].lo, WZ[ZRC + 1].hi);
                if (2 < nzr) {
                    dss_pair_loadx(XZ[1], xb + DSS_ZR_COL(3) * 32);
                    dss_pair_loadx(XR[1], xb + DSS_ZR_COL(ZRL + 3) * 32);
                }
                __builtin_amdgcn_sched_barrier(0);
                if (STAMP) { unsigned long long t = __builtin_readcyclecounter(); sa[0] += t - ta; ta = t; }
                f32x2 gz, gr, gh;                                                     // compute_gru_a_input
                gz.x = ((cz.x + esa.x) + epa.x) + eea.x;  gz.y = ((cz.y + esb.x) + epb.x) + eeb.x;
                gr.x = ((cr.x + esa.y) + epa.y) + eea.y;  gr.y = ((cr.y + esb.y) + epb.y) + eeb.y;
                gh.x = ((ch.x + esa.z) + epa.z) + eea.z;  gh.y = ((ch.y + esb.z) + epb.z) + eeb.z;
                // nnet.c 2021 (default): (bias + diag*state) + input, then the blocks in idx order;
                // nnet.c 2019-20 (blob flag): the blocks first, the input last
                if (!recur_first) { az = az + gz; ar = ar + gr; }
                if (STAMP) { asm volatile("" :: "v"(az), "v"(ar)); unsigned long long t = __builtin_readcyclecounter(); sa[1] += t - ta; ta = t; }
                if (0 < nzr) {                                       // (a wave whose row groups have no z/r blocks at all)
                    DSS_PK_ZR_ADD(az, ar, T0);
                    DSS_PK_ZR_ADD(az, ar, T1);
                }
#pragma unroll
                for (int s = 2; s < ZRC; s += 2) {
                    if (s < nzr) {                                   // nzr is even: slots s and s+1 exist; their inputs are in flight
                        __builtin_amdgcn_sched_barrier(0);
                        DSS_PK_ZR4(az, ar, TT, XZ[0], XR[0], WZ[s].lo, WZ[s].hi, WZ[ZRC + s].lo, WZ[ZRC + s].hi);
                        __builtin_amdgcn_sched_barrier(0);
                        if (s + 2 < ZRC && s + 2 < nzr) {
                            dss_pair_loadx(XZ[0], xb + DSS_ZR_COL(s + 2 < ZRC ? s + 2 : 0) * 32);
                            dss_pair_loadx(XR[0], xb + DSS_ZR_COL(ZRL + (s + 2 < ZRC ? s + 2 : 0)) * 32);
                        }
                        __builtin_amdgcn_sched_barrier(0);
                        DSS_PK_ZR4(az, ar, TT, XZ[1], XR[1], WZ[s + 1].lo, WZ[s + 1].hi, WZ[ZRC + s + 1].lo, WZ[ZRC + s + 1].hi);
                        __builtin_amdgcn_sched_barrier(0);
                        if (s + 2 < ZRC && s + 2 < nzr) {
                            dss_pair_loadx(XZ[1], xb + DSS_ZR_COL(s + 3 < ZRC ? s + 3 : 0) * 32);
                            dss_pair_loadx(XR[1], xb + DSS_ZR_COL(ZRL + (s + 3 < ZRC ? s + 3 : 0)) * 32);
                        }
                    }
                }
                if (recur_first) { az = gz + az; ar = gr + ar; }
                if (STAMP) { asm volatile("" :: "v"(az), "v"(ar)); unsigned long long t = __builtin_readcyclecounter(); sa[2] += t - ta; ta = t; }
                f32x2 z, r;
                dss_sigmoid_pk2(L.tansig, az, ar, z, r);
                f32x2 h = ahv * r + gh;
                h = dss_tanh_pk(L.tansig, h);
                st = z * st + (1 - z) * h;
                *reinterpret_cast<f32x2 *>(&L.state_a[cur ^ 1][2 * unit]) = st;
                if (STAMP) { asm volatile("" :: "v"(st)); unsigned long long t = __builtin_readcyclecounter(); sa[3] += t - ta; ta = t; }
                if (!HAS_FC) __builtin_amdgcn_s_setprio(0);
                __syncthreads();                                                        // barrier B
                if (STAMP) { unsigned long long t = __builtin_readcyclecounter(); sa[4] += t - ta; ta = t; }
                DSS_PH_CHAIN(L.state_a[cur ^ 1])                     // next sample's h chain, under GRU B
                if (myq >= 0) dss_pair_speculate(L, myq * 64 + lane, u2l_c);
                f32x4 fwq[HAS_FC ? NB / 2 : 1];                      // fwq[k] = weights of inputs 2k, 2k+1, each as (layer 0, layer 1)
                if constexpr (HAS_FC) {
                    __builtin_amdgcn_sched_barrier(0);
                    unsigned fo = (unsigned)tid * 16;
                    asm volatile("" : "+v"(fo));                     // a new offset as far as the compiler knows: the loads stay here
                    const char *fp = reinterpret_cast<const char *>(m.fc_w_pair);    // scalar base + 32-bit lane offset
#pragma unroll
                    for (int k = 0; k < NB / 2; ++k) fwq[k] = *reinterpret_cast<const f32x4 *>(fp + (fo + (unsigned)k * DSS_FC_OUT * 16));
                }
                if (STAMP) { unsigned long long t = __builtin_readcyclecounter(); sa[5] += t - ta; ta = t; }
                __syncthreads();                                                        // barrier C
                if (STAMP) { unsigned long long t = __builtin_readcyclecounter(); sa[6] += t - ta; ta = t; }
                if constexpr (HAS_FC) {                                                 // sample_mdense, all nodes, both utterances
                    const float thr_a = L.thr[0][level], thr_b = L.thr[1][level];      // issued first, used last
                    f32x2 s0 = {fb0, fb0}, s1 = {fb1, fb1};                            // layer 0 / layer 1 sums of (A, B)
                    PairX bj[NB / 4];                                                  // all of GRU B's state first: one LDS round trip
#pragma unroll
                    for (int j4 = 0; j4 < NB / 4; ++j4) dss_pair_loadx(bj[j4], reinterpret_cast<const char *>(&L.state_b[4 * j4][0]));
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int j4 = 0; j4 < NB / 4; ++j4) {
                        f32x2 ft[8];
                        DSS_PK_FC4(s0, s1, ft, bj[j4], fwq[2 * j4].lo, fwq[2 * j4].hi, fwq[2 * j4 + 1].lo, fwq[2 * j4 + 1].hi);
                    }
                    f32x2 t1, t2;
                    dss_tanh_pk2(L.tansig, s0, s1, t1, t2);
                    f32x2 lg = ff0 * t1;
                    const f32x2 lg2 = ff1 * t2;
                    lg += lg2;
                    bool bit_a = thr_a < lg.x, bit_b = thr_b < lg.y;
                    if constexpr (TRACE) {               // teacher forcing (tests): record every logit, bend the walk
                        const size_t oa = ((size_t)ua * n_frames + f) * DSS_FRAME_SIZE + i;
                        const size_t ob = ((size_t)ub * n_frames + f) * DSS_FRAME_SIZE + i;
                        if (b.trace_logits) {
                            b.trace_logits[oa * 256 + tid] = tid ? lg.x : 0.f;
                            if (job.has_b) b.trace_logits[ob * 256 + tid] = tid ? lg.y : 0.f;
                        }
                        if (b.force_exc) {
                            const int va = b.force_exc[oa], vb = b.force_exc[ob];       // bits b7..b0, b7 decided at level 0
                            if ((tid ^ (1 << level)) == (va >> (8 - level))) bit_a = (va >> (7 - level)) & 1;
                            if ((tid ^ (1 << level)) == (vb >> (8 - level))) bit_b = (vb >> (7 - level)) & 1;
                        }
                    }
                    const unsigned long long mask_a = __ballot(bit_a), mask_b = __ballot(bit_b);
                    if (lane == 0) {
                        L.bits[0][2 * wave] = (unsigned)mask_a; L.bits[0][2 * wave + 1] = (unsigned)(mask_a >> 32);
                        L.bits[1][2 * wave] = (unsigned)mask_b; L.bits[1][2 * wave + 1] = (unsigned)(mask_b >> 32);
                    }
                }
                __syncthreads();                                                        // barrier D
                if (STAMP) { unsigned long long t = __builtin_readcyclecounter(); sa[7] += t - ta; ta = t; }
                cur ^= 1;
            }
        }
        __syncthreads();                                                                // final barrier of the job
        b.gru_a_state[(size_t)job.sa * NA + unit] = st.x;
        if (job.has_b) b.gru_a_state[(size_t)job.sb * NA + unit] = st.y;
        __syncthreads();                                                                // job barrier 2: LDS free for the next job
    }
    if (STAMP && lane == 0 && b.trace_exc)
        for (int k = 0; k < 8; ++k) b.trace_exc[((size_t)blockIdx.x * 6 + wave) * 8 + k] = (float)sa[k];
}

template <bool TRACE, bool STAMP, int Z, bool RAGGED>
__global__ void __launch_bounds__(512)
lpcnet_sample_pair_kernel(DssModelDev m, DssBatchDev b, int n_utts, int n_frames, short *__restrict__ pcm_out)
{
    __shared__ __attribute__((aligned(16))) PairLds L;
    extern __shared__ __attribute__((aligned(16))) float hblk_lds[];       // h-gate block records (size per model)
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;

    PairPlan plan;
    {
        int u0 = 2 * blockIdx.x, u1 = u0 + 1;
        const bool has1 = u1 < n_utts;
        if constexpr (RAGGED) if (b.row_of) {                        // dispatch order: the longest rows first, neighbours in length together
            u0 = __builtin_amdgcn_readfirstlane(b.row_of[2 * blockIdx.x]);
            u1 = has1 ? __builtin_amdgcn_readfirstlane(b.row_of[2 * blockIdx.x + 1]) : u0 + 1;
        }
        const int fa = b.fc0[u0], fb = has1 ? b.fc0[u1] : fa;
        const bool same = min(fa, DSS_FEATURES_DELAY) == min(fb, DSS_FEATURES_DELAY);
        plan.u0 = u0; plan.u1 = u1; plan.fa = __builtin_amdgcn_readfirstlane(fa); plan.fb = __builtin_amdgcn_readfirstlane(fb);
        plan.packed = has1 && same;
        plan.n_jobs = (has1 && !same) ? 2 : 1;
        plan.sa = u0; plan.sb = has1 ? u1 : u0; plan.nfa = plan.nfb = n_frames;
        if constexpr (RAGGED) {                                      // rows name their decoder slot and frame count
            const int ub_ = has1 ? u1 : u0;
            if (b.slot_of) { plan.sa = __builtin_amdgcn_readfirstlane(b.slot_of[u0]); plan.sb = __builtin_amdgcn_readfirstlane(b.slot_of[ub_]); }
            if (b.count_of) {
                plan.nfa = __builtin_amdgcn_readfirstlane(min(b.count_of[u0], n_frames));
                plan.nfb = __builtin_amdgcn_readfirstlane(min(b.count_of[ub_], n_frames));
            }
            if (plan.packed && plan.nfa != plan.nfb) plan.n_jobs = 2;    // the longer row finishes alone
        }
    }

    // ---------------- one-time staging into LDS -------------------------------------------------------
    for (int k = tid * 4; k < m.hblk_floats; k += 512 * 4)
        *reinterpret_cast<f32x4 *>(&hblk_lds[k]) = *reinterpret_cast<const f32x4 *>(&m.hblk[k]);
    for (int k = tid; k < NB * NB3; k += 512) L.gb_wrec[k] = m.gru_b_w_rec[k];
    if (tid < 201) L.tansig[tid] = m.tansig[tid];
    if (tid < 256) L.ulaw2lin[tid] = m.ulaw2lin[tid];
    __syncthreads();

    if (wave < 4) {
        dss_pair_role_a<TRACE, STAMP, (Z < 8 ? Z : 8), true, RAGGED>(L, hblk_lds, m, b, n_frames, plan, tid, wave, lane);
    } else if (wave < 6) {
        dss_pair_role_a<TRACE, STAMP, Z, false, RAGGED>(L, hblk_lds, m, b, n_frames, plan, tid, wave, lane);
    } else if (wave == 6) {
        // =====================================================================================================
        // role B1: GRU B, the odd stages of the relay (lane = row: 0..15 z, 16..31 r, 32..47 h; both utterances per lane)
        // =====================================================================================================
        f32x4 RW[PR6];                               // weights of this wave's next stage (stage 1 to begin with)
#pragma unroll
        for (int g = 0; g < PR6; ++g)
            RW[g] = *reinterpret_cast<const f32x4 *>(m.gb_w_quad + ((size_t)(DSS_PR_G0(1) + g) * 64 + lane) * 4);
        __builtin_amdgcn_s_setprio(3);               // everything this wave does is on the sample's critical path
        const float u2l_c = L.ulaw2lin[lane];
        unsigned long long relay6 = 0, atc6 = 0, t6 = 0;   // diagnostic build: barrier B to this wave's last hand-over / to its arrival at C
        for (int jn = 0; jn < plan.n_jobs; ++jn) {
            const PairJob job = dss_pair_job<RAGGED>(plan, jn, n_frames);
            dss_pair_job_init(L, b, job, tid);
            __syncthreads();                                             // job barrier 0
            int cur = 0, seq = 0;
            __syncthreads();                                             // job barrier 1
            for (int f = job.f0; f < job.f1; ++f) {
                if (job.fc0 + f < DSS_FEATURES_DELAY) continue;
                for (int i = 0; i < DSS_FRAME_SIZE; ++i) {
                    f32x2 acc;                                                              // (the chain starts and ends on wave 7)
                    ++seq;
                    if (seq == 1) __syncthreads();                                          // barrier A (first sample only)
                    __syncthreads();                                                        // barrier B
                    const char *an = reinterpret_cast<const char *>(L.state_a[cur ^ 1]);
                    if (STAMP) t6 = __builtin_readcyclecounter();
                    unsigned wo = (unsigned)lane * 16;
                    asm volatile("" : "+v"(wo));             // a new offset as far as the compiler knows: the weight loads stay in the loop
                    const char *wq = reinterpret_cast<const char *>(m.gb_w_quad);    // scalar base + 32-bit lane offset
                    f32x2 PS[PR6][4];
                    DSS_PR_MUL(PR6, DSS_PR_G0(1), DSS_PR_G0(3))
                    DSS_PR_AWAIT(seq * 8 + 1)
                    DSS_PR_ADD(PR6)
                    DSS_PR_PUBLISH(seq * 8 + 2)
                    DSS_PR_MUL(PR6, DSS_PR_G0(3), DSS_PR_G0(5))
                    DSS_PR_AWAIT(seq * 8 + 3)
                    DSS_PR_ADD(PR6)
                    DSS_PR_PUBLISH(seq * 8 + 4)
                    DSS_PR_MUL(PR6, DSS_PR_G0(5), DSS_PR_G0(1))
                    DSS_PR_AWAIT(seq * 8 + 5)
                    DSS_PR_ADD(PR6)
                    DSS_PR_PUBLISH(seq * 8 + 6)
                    if (STAMP) relay6 += __builtin_readcyclecounter() - t6;
                    dss_pair_speculate(L, lane, u2l_c);          // this wave is idle from here to barrier B: candidates 0..63
                    if (STAMP) atc6 += __builtin_readcyclecounter() - t6;
                    __syncthreads();                                                        // barrier C
                    __syncthreads();                                                        // barrier D
                    cur ^= 1;
                }
            }
            __syncthreads();                                                                // final barrier of the job
            __syncthreads();                                                                // job barrier 2
        }
        if (STAMP && lane == 0 && b.trace_pcm && gridDim.x == 1) { b.trace_pcm[65] = (float)relay6; b.trace_pcm[66] = (float)atc6; }
    } else {
        // =====================================================================================================
        // role B2 + S (wave 7): GRU B, the even stages of the relay and the gates (both utterances per lane); scalar recurrences with
        // utterance A in lanes 0..15 and utterance B in lanes 16..31
        // =====================================================================================================
        f32x4 RW[PR7];                               // weights of this wave's next stage (stage 0 to begin with)
#pragma unroll
        for (int g = 0; g < PR7; ++g)
            RW[g] = *reinterpret_cast<const f32x4 *>(m.gb_w_quad + ((size_t)(DSS_PR_G0(0) + g) * 64 + lane) * 4);
        const int row = lane < NB3 ? lane : 0;
        __builtin_amdgcn_s_setprio(3);               // everything this wave does is on the sample's critical path
        const float gbb0 = m.gru_b_bias[row];
        const float gbb1 = m.gru_b_bias[NB3 + row];
        const int hb = (lane >> 4) & 1, hl = lane & (DSS_LPC_ORDER - 1);
        unsigned long long stamp_acc[6] = {0, 0, 0, 0, 0, 0};
        unsigned long long t_prev = 0, relay7 = 0, atc7 = 0;   // barrier B to the end of the chain / to this wave's arrival at C (diagnostic build)
        for (int jn = 0; jn < plan.n_jobs; ++jn) {
            const PairJob job = dss_pair_job<RAGGED>(plan, jn, n_frames);
            const int my_utt = hb ? job.ub : job.ua;                 // row of the call: features, PCM, traces
            const int my_slot = hb ? job.sb : job.sa;                // decoder it continues (uniform calls: the row itself)
            const bool owner = hl == 0 && lane < 32 && (hb == 0 || job.has_b);   // the lane that stores its utterance's scalars
            dss_pair_job_init(L, b, job, tid);
            __syncthreads();                                             // job barrier 0
            // signal history and LPC of the current frame, element j of utterance hb in lane 16*hb + j
            float ls_lane = b.last_sig[(size_t)my_slot * DSS_LPC_ORDER + hl], lpc_lane = 0.f;
            float deemph = b.deemph[my_slot];
            int last_exc = b.last_exc[my_slot];
            DssKiss99 rng = {b.rng[my_slot * 4 + 0], b.rng[my_slot * 4 + 1], b.rng[my_slot * 4 + 2], b.rng[my_slot * 4 + 3]};
            int cur = 0, seq = 0;
            float pred = 0.f, upd_pred = 0.f;
            int upd_exc = 0, upd_i = 0;
            bool have_spec = false, next_exists = false, upd_pending = false;
            __syncthreads();                                             // job barrier 1
            for (int f = job.f0; f < job.f1; ++f) {
                short *pcm_a = pcm_out + ((size_t)job.ua * n_frames + f) * DSS_FRAME_SIZE;
                short *pcm_b = pcm_out + ((size_t)job.ub * n_frames + f) * DSS_FRAME_SIZE;
                if (job.fc0 + f < DSS_FEATURES_DELAY) {             // lpcnet.c: frame_count <= FEATURES_DELAY -> silence
                    for (int k = lane; k < DSS_FRAME_SIZE / 2; k += 64) {
                        reinterpret_cast<int *>(pcm_a)[k] = 0;
                        if (job.has_b) reinterpret_cast<int *>(pcm_b)[k] = 0;
                    }
                    if (TRACE)
                        for (int k = lane; k < DSS_FRAME_SIZE; k += 64) {
                            b.trace_exc[((size_t)job.ua * n_frames + f) * DSS_FRAME_SIZE + k] = -1.f;
                            b.trace_pcm[((size_t)job.ua * n_frames + f) * DSS_FRAME_SIZE + k] = 0.f;
                            if (job.has_b) {
                                b.trace_exc[((size_t)job.ub * n_frames + f) * DSS_FRAME_SIZE + k] = -1.f;
                                b.trace_pcm[((size_t)job.ub * n_frames + f) * DSS_FRAME_SIZE + k] = 0.f;
                            }
                        }
                    continue;
                }
                const float *fo = b.frame_out + ((size_t)my_utt * n_frames + f) * DSS_COND_STRIDE;
                lpc_lane = fo[3 * NA + NB3 + hl];
                const f32x2 gbc = {b.frame_out[((size_t)job.ua * n_frames + f) * DSS_COND_STRIDE + 3 * NA + row],
                                   b.frame_out[((size_t)job.ub * n_frames + f) * DSS_COND_STRIDE + 3 * NA + row]};
                for (int i = 0; i < DSS_FRAME_SIZE; ++i) {
                    if (STAMP) t_prev = __builtin_readcyclecounter();
                    if (!have_spec) {        // first sample of the call: prediction and indices computed directly
                        pred = 0;
#pragma unroll
                        for (int j = 0; j < DSS_LPC_ORDER; ++j) {
                            const int src = ((lane & 0x30) | j) * 4;
                            pred -= __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(src, __builtin_bit_cast(int, ls_lane))) *
                                    __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(src, __builtin_bit_cast(int, lpc_lane)));
                        }
                        const int su = dss_lin2ulaw(__builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute((lane & 0x30) * 4, __builtin_bit_cast(int, ls_lane))));
                        const int pu = dss_lin2ulaw(pred);
                        if (hl == 0 && lane < 32) { L.idx[hb][0] = su; L.idx[hb][1] = pu; L.idx[hb][2] = last_exc; }
                    }
                    ++seq;
                    if (STAMP) { unsigned long long t = __builtin_readcyclecounter(); stamp_acc[0] += t - t_prev; t_prev = t; }
                    if (seq == 1) __syncthreads();                                          // barrier A (first sample only)
                    if (STAMP) { unsigned long long t = __builtin_readcyclecounter(); stamp_acc[1] += t - t_prev; t_prev = t; }
                    if (upd_pending) { DSS_PS_UPDATE() }                                    // previous sample's bookkeeping
                    {   // off the critical path: this sample's 8 thresholds per utterance
                        const uint32_t r0 = dss_kiss99_rand(rng);
                        const uint32_t r1 = dss_kiss99_rand(rng);
                        if (hl < 8 && lane < 32) {
                            const uint32_t r = hl < 4 ? r0 : r1;
                            L.thr[hb][hl] = m.logit_table[(r >> (8 * (hl & 3))) & 0xFF];     // 1 KB table, L2/L1 resident
                        }
                    }
                    {   // inputs of the speculation the other waves run between barriers B and C
                        const bool last_of_frame = (i == DSS_FRAME_SIZE - 1);
                        next_exists = !(last_of_frame && f == job.f1 - 1);
                        float lp = lpc_lane;
                        if (last_of_frame && next_exists)
                            lp = b.frame_out[((size_t)my_utt * n_frames + f + 1) * DSS_COND_STRIDE + 3 * NA + NB3 + hl];
                        // history[j-1] arrives in lane j (row_shr:1 inside each row of 16 lanes)
                        const float ls_up = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, ls_lane), 0x111, 0xf, 0xf, false));
                        const float prod = ls_up * lp;
                        if (lane < 32) L.spec_prod[hl][hb] = hl ? prod : lp;
                        if (hl == 0 && lane < 32) L.spec_pred[hb] = pred;
                    }
                    f32x2 rec = {gbb1, gbb1};                                               // GRU B's recurrent half
#pragma unroll
                    for (int j = 0; j < NB; ++j) rec += L.gb_wrec[j * NB3 + row] * *reinterpret_cast<const f32x2 *>(&L.state_b[j][0]);
                    const f32x2 sb_old = *reinterpret_cast<const f32x2 *>(&L.state_b[lane & (NB - 1)][0]);   // the h lanes' own unit
                    __syncthreads();                                                        // barrier B
                    if (STAMP) { unsigned long long t = __builtin_readcyclecounter(); stamp_acc[2] += t - t_prev; t_prev = t; }
                    const char *an = reinterpret_cast<const char *>(L.state_a[cur ^ 1]);
                    unsigned wo = (unsigned)lane * 16;
                    asm volatile("" : "+v"(wo));             // a new offset as far as the compiler knows: the weight loads stay in the loop
                    const char *wq = reinterpret_cast<const char *>(m.gb_w_quad);    // scalar base + 32-bit lane offset
                    f32x2 PS[PR7][4];
                    f32x2 acc = gbb0 + gbc;                                                 // compute_gruB
                    DSS_PR_MUL(PR7, DSS_PR_G0(0), DSS_PR_G0(2))
                    DSS_PR_ADD(PR7)
                    DSS_PR_PUBLISH(seq * 8 + 1)
                    DSS_PR_MUL(PR7, DSS_PR_G0(2), DSS_PR_G0(4))
                    DSS_PR_AWAIT(seq * 8 + 2)
                    DSS_PR_ADD(PR7)
                    DSS_PR_PUBLISH(seq * 8 + 3)
                    DSS_PR_MUL(PR7, DSS_PR_G0(4), DSS_PR_G0(6))
                    DSS_PR_AWAIT(seq * 8 + 4)
                    DSS_PR_ADD(PR7)
                    DSS_PR_PUBLISH(seq * 8 + 5)
                    DSS_PR_MUL(PR7, DSS_PR_G0(6), DSS_PR_G0(0))
                    DSS_PR_AWAIT(seq * 8 + 6)
                    DSS_PR_ADD(PR7)
                    if (STAMP) { asm volatile("" : "+v"(acc)); relay7 += __builtin_readcyclecounter() - t_prev; }
                    {   // gates: lanes 0..15 z, 16..31 r, 32..47 h.  r and z travel up to their unit's h lane with gfx950's
                        // row/half swaps (VALU); the new state is formed in the h lanes.  Only the first result of a swap
                        // is used, with distinct operands (see lpcnet_sample.hip).
                        const f32x2 zr = dss_sigmoid_pk(L.tansig, acc + rec);
                        f32x2 r_for_h, z_for_h;
                        {
                            const float zrx = zr.x;           // (a bit cast applied to a vector ELEMENT reads element 0 with this clang)
                            const unsigned zb = __builtin_bit_cast(unsigned, zrx);
                            const unsigned r_row0 = __builtin_amdgcn_permlane16_swap(zb, 0u, false, false)[1];
                            r_for_h.x = __builtin_bit_cast(float, __builtin_amdgcn_permlane32_swap(0u, r_row0, false, false)[0]);
                            z_for_h.x = __builtin_bit_cast(float, __builtin_amdgcn_permlane32_swap(0u, zb, false, false)[0]);
                        }
                        {
                            const float zry = zr.y;
                            const unsigned zb = __builtin_bit_cast(unsigned, zry);
                            const unsigned r_row0 = __builtin_amdgcn_permlane16_swap(zb, 0u, false, false)[1];
                            r_for_h.y = __builtin_bit_cast(float, __builtin_amdgcn_permlane32_swap(0u, r_row0, false, false)[0]);
                            z_for_h.y = __builtin_bit_cast(float, __builtin_amdgcn_permlane32_swap(0u, zb, false, false)[0]);
                        }
                        f32x2 hh = acc + rec * r_for_h;
                        hh = dss_tanh_pk(L.tansig, hh);
                        if (lane >= 2 * NB && lane < NB3)
                            *reinterpret_cast<f32x2 *>(&L.state_b[lane - 2 * NB][0]) = z_for_h * sb_old + (1 - z_for_h) * hh;
                    }
                    if (STAMP) atc7 += __builtin_readcyclecounter() - t_prev;
                    __syncthreads();                                                        // barrier C
                    if (STAMP) { unsigned long long t = __builtin_readcyclecounter(); stamp_acc[3] += t - t_prev; t_prev = t; }
                    __syncthreads();                                                        // barrier D
                    if (STAMP) { unsigned long long t = __builtin_readcyclecounter(); stamp_acc[4] += t - t_prev; t_prev = t; }
                    cur ^= 1;
                    int va, vb;
                    DSS_TREE_WALK_AT(va, L.bits[0])
                    DSS_TREE_WALK_AT(vb, L.bits[1])
                    const int exc = hb ? vb : va;
                    // the next sample's prediction and mu-law indices were precomputed for every possible exc
                    const float pred_next = L.spec_tab_pred[hb][exc];   // (the GRU A waves look the mu-law indices up themselves)
                    have_spec = next_exists;
                    // Everything below only updates this wave's own state; except at the end of a frame (whose PCM is
                    // copied out right after the loop) it is deferred until after the next barrier A.
                    upd_exc = exc; upd_pred = pred; upd_i = i; upd_pending = true;
                    pred = pred_next;
                    if (i == DSS_FRAME_SIZE - 1) { DSS_PS_UPDATE() }
                    if (STAMP) { unsigned long long t = __builtin_readcyclecounter(); stamp_acc[5] += t - t_prev; t_prev = t; }
                }
                // wave 7 owns L.pcm: LDS operations of one wave are ordered, no barrier needed
                for (int k = lane; k < DSS_FRAME_SIZE / 2; k += 64) {
                    reinterpret_cast<int *>(pcm_a)[k] = reinterpret_cast<const int *>(L.pcm[0])[k];
                    if (job.has_b) reinterpret_cast<int *>(pcm_b)[k] = reinterpret_cast<const int *>(L.pcm[1])[k];
                }
            }
            __syncthreads();                                                                // final barrier of the job
            if (lane < NB) {
                b.gru_b_state[(size_t)job.sa * NB + lane] = L.state_b[lane][0];
                if (job.has_b) b.gru_b_state[(size_t)job.sb * NB + lane] = L.state_b[lane][1];
            }
            if (lane < 32 && (hb == 0 || job.has_b)) b.last_sig[(size_t)my_slot * DSS_LPC_ORDER + hl] = ls_lane;
            if (owner) {
                b.deemph[my_slot] = deemph;
                b.last_exc[my_slot] = last_exc;
                b.rng[my_slot * 4 + 0] = rng.z; b.rng[my_slot * 4 + 1] = rng.w; b.rng[my_slot * 4 + 2] = rng.jsr; b.rng[my_slot * 4 + 3] = rng.jcong;
            }
            __syncthreads();                                                                // job barrier 2
        }
        if (STAMP && lane == 0 && b.trace_pcm)          // diagnostic build only
        {
            for (int k = 0; k < 6; ++k) b.trace_pcm[(size_t)blockIdx.x * 6 + k] = (float)stamp_acc[k];
            if (gridDim.x == 1) { b.trace_pcm[64] = (float)relay7; b.trace_pcm[67] = (float)atc7; }
        }
    }
}

// 1 when the CU-resident layout of model m also fits beside two utterances' state (plain layouts only: models that
// need the z/r tail or long-list paths stay on the latency kernel)
int dss_pair_fits(const DssModelDev &m)
{
    return m.fast_ok && !m.ext && (size_t)m.hblk_floats * sizeof(float) <= DSS_PAIR_HBLK_BYTES;
}

// trace: 0, 1 (excitation / pcm trace and teacher forcing), 2 (phase stamps of the diagnostic build: uniform calls only).
// Ragged calls (b.slot_of / b.count_of) run the RAGGED instantiation: rows 2k and 2k+1 share a workgroup, so a caller
// that orders its rows by length (longest first, as the synthesis queue does) pairs rows of near-equal length.
int dss_launch_sample_network_pair(const DssModelDev &m, DssBatchDev &b, int n_utts, int n_frames, short *d_pcm, int trace,
                                   hipStream_t s)
{
    if (!dss_pair_fits(m)) { dss_set_error("pair kernel: model layout not supported"); return DSS_EINVAL; }
    const bool ragged = b.slot_of || b.count_of;
    if (ragged && trace == 2) { dss_set_error("phase stamps are taken on uniform calls only"); return DSS_EINVAL; }
    const size_t dyn = ((size_t)m.hblk_floats * sizeof(float) + 15) & ~(size_t)15;
    const bool z10 = m.zr_cap <= 10;
    static std::mutex attr_mu;
    static unsigned long long attr_set = 0;
    int dev = 0;
    DSS_HIP_CHECK(hipGetDevice(&dev));
    {
        std::lock_guard<std::mutex> attr_lk(attr_mu);
        if (!(attr_set >> (dev & 63) & 1)) {
#define DSS_SET_ATTR(K) DSS_HIP_CHECK(hipFuncSetAttribute((const void *)K, hipFuncAttributeMaxDynamicSharedMemorySize, DSS_PAIR_HBLK_BYTES))
            DSS_SET_ATTR((lpcnet_sample_pair_kernel<false, false, 10, false>)); DSS_SET_ATTR((lpcnet_sample_pair_kernel<false, false, 12, false>));
            DSS_SET_ATTR((lpcnet_sample_pair_kernel<true, false, 10, false>));  DSS_SET_ATTR((lpcnet_sample_pair_kernel<true, false, 12, false>));
            DSS_SET_ATTR((lpcnet_sample_pair_kernel<false, true, 10, false>));  DSS_SET_ATTR((lpcnet_sample_pair_kernel<false, true, 12, false>));
            DSS_SET_ATTR((lpcnet_sample_pair_kernel<false, false, 10, true>));  DSS_SET_ATTR((lpcnet_sample_pair_kernel<false, false, 12, true>));
            DSS_SET_ATTR((lpcnet_sample_pair_kernel<true, false, 10, true>));   DSS_SET_ATTR((lpcnet_sample_pair_kernel<true, false, 12, true>));
#undef DSS_SET_ATTR
            attr_set |= 1ull << (dev & 63);
        }
    }
    const dim3 grid((n_utts + 1) / 2), block(512);
#define DSS_LAUNCH(T, S2, R)                                                                                           \
    do {                                                                                                               \
        if (z10) hipLaunchKernelGGL((lpcnet_sample_pair_kernel<T, S2, 10, R>), grid, block, dyn, s, m, b, n_utts, n_frames, d_pcm); \
        else hipLaunchKernelGGL((lpcnet_sample_pair_kernel<T, S2, 12, R>), grid, block, dyn, s, m, b, n_utts, n_frames, d_pcm);     \
    } while (0)
    if (trace == 2) DSS_LAUNCH(false, true, false);
    else if (trace && ragged) DSS_LAUNCH(true, false, true);
    else if (trace) DSS_LAUNCH(true, false, false);
    else if (ragged) DSS_LAUNCH(false, false, true);
    else DSS_LAUNCH(false, false, false);
#undef DSS_LAUNCH
    DSS_HIP_CHECK(hipGetLastError());
    return DSS_OK;
}
