// csrc/lpcnet_sample_multi.hip -- LPCNet sample-rate network, THROUGHPUT form: U = 3 or 4 utterances per persistent
// workgroup, software-pipelined over the same eight role-specialised waves and the same CU-resident weights as
// lpcnet_sample.hip (which stays the latency form: one utterance per workgroup, used up to one utterance per CU).
//
// Why: with one utterance per CU every wave sits idle about half of a sample period (the serial chain GRU A -> GRU B ->
// dual-FC -> tree walk leaves one role working at a time), and a second workgroup cannot share the CU because one
// already fills its registers and LDS with weights.  Here the weights are shared and the roles work on DIFFERENT
// utterances at the same time.  Time is cut into slots; in slot s the utterance a = s mod U is "at the front":
//
//   waves 0..5 (role A)   dual-FC of a's finished sample (waves 0..3; waves 4, 5 take half of c's speculation instead) ->
//                         "bits published" counter in LDS (no s_barrier) -> tree walk, speculated embedding indices,
//                         embedding rows + conditioning (L2) ... while those are in flight: the h-gate chain of b = a-1
//                         (state written last slot) ... -> z/r block products of a -> "old state read" counter -> z/r sums,
//                         gates, new GRU A state of a
//   wave 6                GRU B inputs 0..191 of b (its GRU A state is one slot old); speculation candidates 0..127 of c
//   wave 7                GRU B inputs 192..383 + gates of c (wave 6 did its first half one slot ago); then the scalar
//                         bookkeeping of b: tree walk, PCM / de-emphasis / history, next kiss99 thresholds, the inputs
//                         of b's next speculation
//   ONE workgroup barrier per slot.
//
// STATUS (round 2, measured on MI355X, tools/multi_time.py, 1024 x 1-s utterances): bit-exact, but 196 ms per batch against
// 172 ms for four rounds of the latency kernel -- a slot takes ~7.8 k cycles where the design needs < 6.4 k to break even.
// The counters (tools/prof_multi.sh) show why: VALU issue and LDS array are each busy only a third of the time; every
// phase of a role-A wave (products, dual-FC, h chain, speculation, sums) is a dependent chain that waits on LDS / L2
// round trips, and with all eight waves active those round trips are about twice as long as in the latency kernel, where
// the same phases run while most other waves sit at a barrier.  dss_lpcnet_batch_set_multi() therefore keeps this form
// opt-in; the automatic choice is the latency kernel.  DESIGN.md section 5 has the numbers and what would change them.
//
// An utterance therefore advances one sample every U slots (3 slots of work; with U = 4 one slot of slack), and the
// workgroup finishes one utterance-sample per slot.  All arithmetic, and its order, is that of lpcnet_sample.hip (same
// macros, same helper functions): summation order of xiph's sparse_sgemv_accum8x4 / sgemv_accum, -ffp-contract=off,
// exact speculation.  Output is bit-identical to the latency kernel and to the CPU oracle (tests/test_gpu_lpcnet.py).
// Reference binding: extensions/lpcnet/cLPCNet.pxd:13; the caller this serves: local/training.py:165-207 (bulk synthesis).
#include <mutex>

#define DSS_H_STORE(V) ah_dst[uh] = (V)
#include "lpcnet_sample_common.h"

#define MU_MAX 4
#define GBH 192                           // GRU B inputs per relay wave (wave 6: 0..191, wave 7: 192..383), all in VGPRs

struct MuShared {                         // static LDS, shared by the utterances of the workgroup
    float gb_wrec[NB * NB3];              // GRU B recurrent weights [16][48]
    float tansig[208];
    float ulaw2lin[256];
    int cnt_bits;                         // dual-FC waves that have published the front utterance's decision bits (monotonic)
    int cnt_reads;                        // role-A waves that have finished reading the front utterance's old state (monotonic)
    int pad[2];
};

struct MuUtt {                            // dynamic LDS, one per utterance, behind the h-gate block image
    float state_a[NA];                    // GRU A state (single buffer: the rendezvous separates its readers from its writers)
    float zero4[4];                       // "column 96": the input of the h-gate slots a row group does not use
    float ah[NA];                         // h-gate pre-activation of the coming sample
    float spec_tab_pred[256];             // speculation tables over the 256 possible excitations of the current sample
    unsigned short spec_tab_idx[256];
    float spec_ls[DSS_LPC_ORDER];         // inputs of the speculation, published by wave 7
    float spec_lpc[DSS_LPC_ORDER];
    float gb_acc[64];                     // GRU B partial sums, wave 6 -> wave 7 (one slot apart)
    float state_b[NB];
    float thr[8];
    unsigned bits[8];
    int idx[4];                           // embedding indices of the first sample (computed directly)
    float spec_pred;                      // this sample's prediction (input of the speculation)
    float pred;                           // wave 7: prediction of the sample whose excitation is being sampled
    float deemph;
    int last_exc;
    unsigned rng[4];
    float ls[DSS_LPC_ORDER];              // wave 7: signal history
    float pad[DSS_LPC_ORDER];
    short pcm[DSS_FRAME_SIZE];
};
static_assert(sizeof(MuUtt) % 16 == 0, "utterance records must keep 16-byte alignment");
static_assert(sizeof(MuShared) % 16 == 0, "dynamic LDS must start 16-byte aligned");

// tree walk over the decision bits of one utterance (scalar code; see lpcnet_sample.hip DSS_TREE_WALK)
#define MU_TREE_WALK(VAL, BITS)                                                                  \
    {                                                                                            \
        const uint4 b0 = *reinterpret_cast<const uint4 *>(&(BITS)[0]);                           \
        const uint4 b1 = *reinterpret_cast<const uint4 *>(&(BITS)[4]);                           \
        const unsigned long long m0 = ((unsigned long long)__builtin_amdgcn_readfirstlane(b0.y) << 32) | (unsigned)__builtin_amdgcn_readfirstlane(b0.x); \
        const unsigned long long m1 = ((unsigned long long)__builtin_amdgcn_readfirstlane(b0.w) << 32) | (unsigned)__builtin_amdgcn_readfirstlane(b0.z); \
        const unsigned long long m2 = ((unsigned long long)__builtin_amdgcn_readfirstlane(b1.y) << 32) | (unsigned)__builtin_amdgcn_readfirstlane(b1.x); \
        const unsigned long long m3 = ((unsigned long long)__builtin_amdgcn_readfirstlane(b1.w) << 32) | (unsigned)__builtin_amdgcn_readfirstlane(b1.z); \
        int tnode_;                                                                              \
        unsigned long long mm_;                                                                  \
        asm volatile(                                                                            \
            "s_mov_b32 %0, 0\n\t"                                                                \
            "s_bitcmp1_b64 %3, 1\n\t"          "s_addc_u32 %0, %0, %0\n\t"                       \
            "s_or_b32 %1, %0, 2\n\t"           "s_bitcmp1_b64 %3, %1\n\t"   "s_addc_u32 %0, %0, %0\n\t" \
            "s_or_b32 %1, %0, 4\n\t"           "s_bitcmp1_b64 %3, %1\n\t"   "s_addc_u32 %0, %0, %0\n\t" \
            "s_or_b32 %1, %0, 8\n\t"           "s_bitcmp1_b64 %3, %1\n\t"   "s_addc_u32 %0, %0, %0\n\t" \
            "s_or_b32 %1, %0, 16\n\t"          "s_bitcmp1_b64 %3, %1\n\t"   "s_addc_u32 %0, %0, %0\n\t" \
            "s_or_b32 %1, %0, 32\n\t"          "s_bitcmp1_b64 %3, %1\n\t"   "s_addc_u32 %0, %0, %0\n\t" \
            "s_bitcmp1_b64 %4, %0\n\t"         "s_addc_u32 %0, %0, %0\n\t"                       \
            "s_cmp_lt_u32 %0, 64\n\t"          "s_cselect_b64 %2, %5, %6\n\t"                    \
            "s_bitcmp1_b64 %2, %0\n\t"         "s_addc_u32 %0, %0, %0"                            \
            : "=&s"(VAL), "=&s"(tnode_), "=&s"(mm_)                                              \
            : "s"(m0), "s"(m1), "s"(m2), "s"(m3)                                                 \
            : "scc");                                                                            \
    }

// one candidate excitation of utterance UC's current sample: next prediction and both mu-law indices
// (lpcnet_synthesize_tail_impl's expressions, in its order; see lpcnet_sample.hip)
#define MU_SPECULATE(UC, CAND, U2L)                                                              \
    {                                                                                            \
        const int cand_ = (CAND);                                                                \
        const float pcm_c = (UC).spec_pred + (U2L);                                              \
        float pc = 0;                                                                            \
        pc -= pcm_c * (UC).spec_lpc[0];                                                          \
        _Pragma("unroll") for (int j = 1; j < DSS_LPC_ORDER; ++j) pc -= (UC).spec_ls[j - 1] * (UC).spec_lpc[j]; \
        const int su_c = dss_lin2ulaw(pcm_c), pu_c = dss_lin2ulaw(pc);                           \
        (UC).spec_tab_pred[cand_] = pc;                                                          \
        (UC).spec_tab_idx[cand_] = (unsigned short)(su_c | (pu_c << 8));                         \
    }

// diagnostic build only: cycles per segment, accumulated per wave (never used for timing claims)
#define MU_STAMP(SEG) if (STAMP) { const unsigned long long t_ = __builtin_readcyclecounter(); sacc[SEG] += t_ - tprev; tprev = t_; }

// z/r block products with every state read in flight before the first multiplication: the product registers are the
// landing zone of the loads (DSS_ZR_PRODUCTS waits for each pair of reads before it multiplies, which is free under the
// GRU B shadow of the latency kernel and an LDS round trip per slot pair here)
#define MU_ZR_PRODUCTS(XBUF)                                                                     \
    {                                                                                            \
        const char *xb = reinterpret_cast<const char *>(XBUF);                                   \
        _Pragma("unroll") for (int s2 = 0; s2 < ZRC; s2 += 2) {                                  \
            if (s2 >= nzr) break;                                                                \
            _Pragma("unroll") for (int u = 0; u < 2; ++u) {                                      \
                PR[s2 + u] = *reinterpret_cast<const f32x4 *>(xb + DSS_ZR_COL(s2 + u) * 16);     \
                PR[ZRC + s2 + u] = *reinterpret_cast<const f32x4 *>(xb + DSS_ZR_COL(ZRL + s2 + u) * 16); \
            }                                                                                    \
        }                                                                                        \
        _Pragma("unroll") for (int s2 = 0; s2 < ZRC; s2 += 2) {                                  \
            if (s2 >= nzr) break;                                                                \
            _Pragma("unroll") for (int u = 0; u < 2; ++u) {                                      \
                PR[s2 + u].lo = WZ[s2 + u].lo * PR[s2 + u].lo;             PR[s2 + u].hi = WZ[s2 + u].hi * PR[s2 + u].hi;             \
                PR[ZRC + s2 + u].lo = WZ[ZRC + s2 + u].lo * PR[ZRC + s2 + u].lo; PR[ZRC + s2 + u].hi = WZ[ZRC + s2 + u].hi * PR[ZRC + s2 + u].hi; \
            }                                                                                    \
        }                                                                                        \
    }

// per-utterance schedule constants (identical in every wave: computed from the same global data)
struct MuPlan {
    int utt[MU_MAX];      // row of the call (= decoder slot), -1 when the workgroup has fewer utterances
    int f0[MU_MAX];       // first frame that is synthesised (the first FEATURES_DELAY frames of a fresh decoder are silent)
    int n[MU_MAX];        // samples to synthesise
};

// P.x[i] for a wave-uniform run-time i: a select chain over constant indices keeps the plan in SGPRs (a dynamically
// indexed private array would live in scratch memory, hundreds of cycles per access)
__device__ __forceinline__ int mu_pick(const int (&v)[MU_MAX], int i)
{
    return i == 0 ? v[0] : i == 1 ? v[1] : i == 2 ? v[2] : v[3];
}

__device__ __forceinline__ void mu_make_plan(MuPlan &P, const DssBatchDev &b, int n_frames, int n_utts, int U, int &n_max)
{
    n_max = 0;
#pragma unroll
    for (int j = 0; j < MU_MAX; ++j) {
        const int u = blockIdx.x * U + j;
        const bool ok = j < U && u < n_utts;
        P.utt[j] = ok ? u : -1;
        int f0 = 0, n = 0;
        if (ok) {
            const int fc0 = __builtin_amdgcn_readfirstlane(b.fc0[u]);
            f0 = fc0 < DSS_FEATURES_DELAY ? DSS_FEATURES_DELAY - fc0 : 0;
            if (f0 > n_frames) f0 = n_frames;
            n = (n_frames - f0) * DSS_FRAME_SIZE;
        }
        P.f0[j] = f0;
        P.n[j] = n;
        n_max = n > n_max ? n : n_max;
    }
}

// =====================================================================================================
// role A: GRU A (+ dual-FC on waves 0..3, + a share of the speculation on waves 0, 1, 5)
// =====================================================================================================
template <int Z, bool HAS_FC, bool STAMP>
__device__ __forceinline__ void mu_role_a(MuShared &S, MuUtt *UT, float *hblk_lds, const DssModelDev &m, const DssBatchDev &b,
                                          int n_frames, int n_utts, int U, int tid, int wave, int lane)
{
    MuPlan P;
    int n_max;
    mu_make_plan(P, b, n_frames, n_utts, U, n_max);
    const int total_slots = U * (n_max + 1) + 2;

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
    f32x2 fw[HAS_FC ? NB : 1];
    float fb0 = 0, fb1 = 0, ff0 = 0, ff1 = 0;
    if constexpr (HAS_FC) {
        const int node = tid;
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            fw[j].x = m.fc_w[(size_t)node * 2 * NB + j];
            fw[j].y = m.fc_w[(size_t)node * 2 * NB + NB + j];
        }
        fb0 = m.fc_bias[node]; fb1 = m.fc_bias[DSS_FC_OUT + node];
        ff0 = m.fc_factor[node]; ff1 = m.fc_factor[DSS_FC_OUT + node];
    }
    const float u2l_c = S.ulaw2lin[(tid - 128) & 255];           // this lane's excitation candidate (waves 4, 5: 128..255)
    const int level = 31 - __clz(tid | 1);                       // FC node = (1 << level) | prefix
    const bool recur_first = m.h.gru_a_order == DSS_GRUA_RECUR_FIRST;
    // h-gate chain of every utterance's FIRST sample (later ones run one slot after the state they need was written)
#pragma unroll
    for (int j = 0; j < MU_MAX; ++j)
        if (j < U && P.n[j] > 0) {
            float *ah_dst = UT[j].ah;
            DSS_H_CHAIN(UT[j].state_a)
        }
    __syncthreads();                                             // prologue barrier (wave 7 has set the utterances up)

    unsigned long long sacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tprev = STAMP ? __builtin_readcyclecounter() : 0;
    int a = 0, k = 0;                                            // front utterance and its A-slot index: a = s mod U, k = s div U
    for (int s = 0; s < total_slots; ++s) {
#pragma unroll
        for (int q = 0; q < (2 * ZRL + 3) / 4; ++q) asm volatile("" : "+v"(PZ[q]));
#pragma unroll
        for (int q = 0; q < HC / 4; ++q) asm volatile("" : "+v"(PH[q]));
        const int bq = a == 0 ? U - 1 : a - 1;                   // utterance one slot behind the front
        const int cq = bq == 0 ? U - 1 : bq - 1;                 // ... two slots behind
        const int kb = bq < a ? k : k - 1;                       // their A-slot indices
        const int kc = cq < a ? k : k - 1;
        MuUtt &UA = UT[a];
        const bool do_a1 = k < mu_pick(P.n, a);
        const bool do_fc = k >= 1 && k <= mu_pick(P.n, a);
        const bool do_h = s >= 1 && kb >= 0 && kb + 1 < mu_pick(P.n, bq);
        const bool do_s = s >= 2 && kc >= 0 && kc + 1 < mu_pick(P.n, cq);

        // ---- (1) dual-FC of the front utterance's finished sample (waves 0..3): its decision bits gate everybody's tree walk.
        // Waves 4 and 5 have no dual-FC: they take their quarter of c's speculation meanwhile. ----------------------------------
        if constexpr (HAS_FC) {
            if (do_fc) {                                                            // sample_mdense, all nodes
                const float thr_lv = UA.thr[level];
                f32x2 s12 = {fb0, fb1};
#pragma unroll
                for (int j4 = 0; j4 < NB / 4; ++j4) {
                    const f32x4 bj = *reinterpret_cast<const f32x4 *>(UA.state_b + 4 * j4);
                    const f32x2 q0 = fw[4 * j4 + 0] * (f32x2){bj.x, bj.x};
                    const f32x2 q1 = fw[4 * j4 + 1] * (f32x2){bj.y, bj.y};
                    const f32x2 q2 = fw[4 * j4 + 2] * (f32x2){bj.z, bj.z};
                    const f32x2 q3 = fw[4 * j4 + 3] * (f32x2){bj.w, bj.w};
                    s12 += q0;
                    s12 += q1;
                    s12 += q2;
                    s12 += q3;
                }
                float s1 = s12.x, s2 = s12.y;
                float t1, t2;
                dss_tanh_approx2(S.tansig, s1, s2, t1, t2);
                s1 = ff0 * t1;
                s2 = ff1 * t2;
                s1 += s2;
                const bool bit = thr_lv < s1;
                const unsigned long long mask = __ballot(bit);
                if (lane == 0) { UA.bits[2 * wave] = (unsigned)mask; UA.bits[2 * wave + 1] = (unsigned)(mask >> 32); }
            }
            // relaxed: a wave's LDS operations execute in issue order, so the bits are written before the count moves
            if (lane == 0) __hip_atomic_fetch_add(&S.cnt_bits, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        } else {
            if (do_s) {
                MuUtt &UC = UT[cq];
                MU_SPECULATE(UC, tid - 128, u2l_c)               // wave 4: candidates 128..191, wave 5: 192..255
            }
        }
        MU_STAMP(0)
        // ---- (2) the decision bits of all four dual-FC waves, tree walk, speculated indices; embedding rows and this frame's
        // conditioning are requested from L2 now and used in (5) -------------------------------------------------------------------
        while (__hip_atomic_load(&S.cnt_bits, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < 4 * (s + 1))
            ;
        MU_STAMP(1)
        int si = 0, pi = 0, ei = 0;
        if (do_fc && do_a1) {
            int exc_;
            MU_TREE_WALK(exc_, UA.bits)
            const unsigned sidx = __builtin_amdgcn_readfirstlane((unsigned)UA.spec_tab_idx[exc_]);
            si = (int)(sidx & 0xFF); pi = (int)(sidx >> 8); ei = exc_;
        } else if (do_a1) {                                      // first sample of the call: wave 7 computed them directly
            si = UA.idx[0]; pi = UA.idx[1]; ei = UA.idx[2];
        }
        si = __builtin_amdgcn_readfirstlane(si); pi = __builtin_amdgcn_readfirstlane(pi); ei = __builtin_amdgcn_readfirstlane(ei);
        typedef float f32x3 __attribute__((ext_vector_type(3)));
        f32x3 es = {0, 0, 0}, ep = {0, 0, 0}, ee = {0, 0, 0};
        float cz = 0, cr = 0, ch = 0;
        if (do_a1) {
            const float *fo = b.frame_out + ((size_t)mu_pick(P.utt, a) * n_frames + mu_pick(P.f0, a) + k / DSS_FRAME_SIZE) * DSS_COND_STRIDE;
            cz = fo[(unsigned)unit]; cr = fo[(unsigned)(NA + unit)]; ch = fo[(unsigned)(2 * NA + unit)];
            es = *reinterpret_cast<const f32x3 *>(m.embed_lane[0] + ((unsigned)si * NA + (unsigned)tid) * 3);
            ep = *reinterpret_cast<const f32x3 *>(m.embed_lane[1] + ((unsigned)pi * NA + (unsigned)tid) * 3);
            ee = *reinterpret_cast<const f32x3 *>(m.embed_lane[2] + ((unsigned)ei * NA + (unsigned)tid) * 3);
        }
        MU_STAMP(2)
        // ---- (3) side work under the L2 latency: h-gate chain of b (its state was written last slot).  The product registers
        // are not live yet, which is what keeps this role inside 256 VGPRs without spill reloads in the slot loop. ------------
        if (do_h) {
            float *ah_dst = UT[bq].ah;
            DSS_H_CHAIN(UT[bq].state_a)                          // h-gate chain of b's coming sample
        }
        MU_STAMP(3)
        // ---- (4) front utterance: old state, z/r block products; then "my reads of its state are done" ------------------------
        f32x4 PR[2 * ZRC];                                       // z/r block products of the front utterance's coming sample
        float st = 0, ahv = 0;
        if (do_a1) {
            st = UA.state_a[unit];
            ahv = UA.ah[unit];
            MU_ZR_PRODUCTS(UA.state_a)
        }
        if (lane == 0) __hip_atomic_fetch_add(&S.cnt_reads, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        MU_STAMP(4)
        // ---- (5) the front utterance's sample: z/r sums, gates, new state (nobody overwrites the state another wave still reads)
        while (__hip_atomic_load(&S.cnt_reads, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < 6 * (s + 1))
            ;
        if (do_a1) {
            float az = rbz + dgz * st;                           // compute_sparse_gru, before the input term
            float ar = rbr + dgr * st;
            const float gz = ((cz + es.x) + ep.x) + ee.x;        // compute_gru_a_input
            const float gr = ((cr + es.y) + ep.y) + ee.y;
            const float gh = ((ch + es.z) + ep.z) + ee.z;
            if (!recur_first) { az = az + gz; ar = ar + gr; }
#pragma unroll
            for (int s2 = 0; s2 < ZRC; s2 += 2) {
                if (s2 >= nzr) break;
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    az += PR[s2 + u].x; ar += PR[ZRC + s2 + u].x;
                    az += PR[s2 + u].y; ar += PR[ZRC + s2 + u].y;
                    az += PR[s2 + u].z; ar += PR[ZRC + s2 + u].z;
                    az += PR[s2 + u].w; ar += PR[ZRC + s2 + u].w;
                }
            }
            if (recur_first) { az = gz + az; ar = gr + ar; }
            float z, r;
            dss_sigmoid_approx2(S.tansig, az, ar, z, r);
            float h = ahv * r + gh;
            h = dss_tanh_approx(S.tansig, h);
            st = z * st + (1 - z) * h;
            UA.state_a[unit] = st;
        }
        MU_STAMP(5)
        __syncthreads();                                                            // slot barrier
        MU_STAMP(6)
        if (++a == U) { a = 0; ++k; }
    }
    if (STAMP && lane == 0 && blockIdx.x == 0 && b.trace_pcm)
        for (int q = 0; q < 8; ++q) b.trace_pcm[wave * 8 + q] = (float)sacc[q];
    // write the GRU A states back (every lane owns one unit; utterance records are quiescent after the last barrier)
#pragma unroll
    for (int j = 0; j < MU_MAX; ++j)
        if (j < U && P.utt[j] >= 0) b.gru_a_state[(size_t)P.utt[j] * NA + unit] = UT[j].state_a[unit];
}

template <int Z, bool STAMP>
__global__ void __launch_bounds__(512)
lpcnet_sample_multi_kernel(DssModelDev m, DssBatchDev b, int n_frames, short *__restrict__ pcm_out, int n_utts, int U)
{
    __shared__ __attribute__((aligned(16))) MuShared S;
    extern __shared__ __attribute__((aligned(16))) float dyn_lds[];         // h-gate block records, then U utterance records
    float *hblk_lds = dyn_lds;
    MuUtt *UT = reinterpret_cast<MuUtt *>(reinterpret_cast<char *>(dyn_lds) + (((size_t)m.hblk_floats * sizeof(float) + 15) & ~(size_t)15));
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;

    // ---------------- one-time staging into LDS -------------------------------------------------------
    for (int q = tid * 4; q < m.hblk_floats; q += 512 * 4)
        *reinterpret_cast<f32x4 *>(&hblk_lds[q]) = *reinterpret_cast<const f32x4 *>(&m.hblk[q]);
    for (int q = tid; q < NB * NB3; q += 512) S.gb_wrec[q] = m.gru_b_w_rec[q];
    if (tid < 201) S.tansig[tid] = m.tansig[tid];
    if (tid < 256) S.ulaw2lin[tid] = m.ulaw2lin[tid];
    if (tid == 0) { S.cnt_bits = 0; S.cnt_reads = 0; }
    for (int j = 0; j < U; ++j) {
        const int u = blockIdx.x * U + j;
        if (tid < 4) UT[j].zero4[tid] = 0.f;
        if (u < n_utts) {
            if (tid < NA) UT[j].state_a[tid] = b.gru_a_state[(size_t)u * NA + tid];
            if (tid < NB) UT[j].state_b[tid] = b.gru_b_state[(size_t)u * NB + tid];
        }
    }
    __syncthreads();

    if (wave < 4) {
        mu_role_a<(Z < 8 ? Z : 8), true, STAMP>(S, UT, hblk_lds, m, b, n_frames, n_utts, U, tid, wave, lane);
        return;
    }
    if (wave < 6) {
        mu_role_a<Z, false, STAMP>(S, UT, hblk_lds, m, b, n_frames, n_utts, U, tid, wave, lane);
        return;
    }
    MuPlan P;
    int n_max;
    mu_make_plan(P, b, n_frames, n_utts, U, n_max);
    const int total_slots = U * (n_max + 1) + 2;
    const int row = lane < NB3 ? lane : 0;
    f32x2 WB[GBH / 2];
    if (wave == 6) {
        // =====================================================================================================
        // role B1: GRU B over inputs 0..191 of the utterance one slot behind the front; lane = row
        // =====================================================================================================
#pragma unroll
        for (int j = 0; j < GBH / 2; ++j) {
            WB[j].x = m.gb_w_lane[(size_t)(2 * j) * 64 + lane];
            WB[j].y = m.gb_w_lane[(size_t)(2 * j + 1) * 64 + lane];
        }
        const float gbb0 = m.gru_b_bias[row];
        __builtin_amdgcn_s_setprio(1);          // the youngest waves of the workgroup lose every issue arbitration otherwise
        __syncthreads();                                             // prologue barrier
        unsigned long long sacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tprev = STAMP ? __builtin_readcyclecounter() : 0;
        int a = 0, k = 0;
        for (int s = 0; s < total_slots; ++s) {
            const int bq = a == 0 ? U - 1 : a - 1;
            const int cq = bq == 0 ? U - 1 : bq - 1;
            const int kb = bq < a ? k : k - 1;
            const int kc = cq < a ? k : k - 1;
            if (s >= 1 && kb >= 0 && kb < mu_pick(P.n, bq)) {
                MuUtt &UB = UT[bq];
                const float gbc = b.frame_out[((size_t)mu_pick(P.utt, bq) * n_frames + mu_pick(P.f0, bq) + kb / DSS_FRAME_SIZE) * DSS_COND_STRIDE + 3 * NA + row];
                float acc = gbb0 + gbc;                                                 // compute_gruB
                const float *an = UB.state_a;
                DSS_GB_CHAIN(an, GBH)
                UB.gb_acc[lane] = acc;
            }
            MU_STAMP(0)
            if (s >= 2 && kc >= 0 && kc + 1 < mu_pick(P.n, cq)) {
                MuUtt &UC = UT[cq];
                MU_SPECULATE(UC, lane, S.ulaw2lin[lane])                               // candidates 0..63
                MU_SPECULATE(UC, 64 + lane, S.ulaw2lin[64 + lane])                     // ... and 64..127
            }
            MU_STAMP(1)
            __syncthreads();                                                            // slot barrier
            MU_STAMP(2)
            if (++a == U) { a = 0; ++k; }
        }
        if (STAMP && lane == 0 && blockIdx.x == 0 && b.trace_pcm)
            for (int q = 0; q < 8; ++q) b.trace_pcm[wave * 8 + q] = (float)sacc[q];
        return;
    }
    // =====================================================================================================
    // role B2 + S (wave 7): GRU B inputs 192..383 and gates of the utterance two slots behind the front, then the
    // scalar recurrences of the utterance one slot behind (its decision bits were written last slot)
    // =====================================================================================================
#pragma unroll
    for (int j = 0; j < GBH / 2; ++j) {
        WB[j].x = m.gb_w_lane[(size_t)(GBH + 2 * j) * 64 + lane];
        WB[j].y = m.gb_w_lane[(size_t)(GBH + 2 * j + 1) * 64 + lane];
    }
    const float gbb1 = m.gru_b_bias[NB3 + row];
    // ---- prologue: silent frames, scalar state, and the first sample's prediction / indices / thresholds ----------
#pragma unroll
    for (int j = 0; j < MU_MAX; ++j) {
        const int u = P.utt[j];
        if (j >= U || u < 0) continue;
        MuUtt &UJ = UT[j];
        for (int f = 0; f < P.f0[j]; ++f) {                  // lpcnet.c: frame_count <= FEATURES_DELAY -> silence
            short *pcm_frame = pcm_out + ((size_t)u * n_frames + f) * DSS_FRAME_SIZE;
            for (int q = lane; q < DSS_FRAME_SIZE / 2; q += 64) reinterpret_cast<int *>(pcm_frame)[q] = 0;
        }
        const float ls_lane = b.last_sig[(size_t)u * DSS_LPC_ORDER + (lane & (DSS_LPC_ORDER - 1))];
        DssKiss99 rng = {b.rng[u * 4 + 0], b.rng[u * 4 + 1], b.rng[u * 4 + 2], b.rng[u * 4 + 3]};
        const int last_exc = b.last_exc[u];
        float pred = 0.f;
        if (P.n[j] > 0) {
            const float lpc_lane = b.frame_out[((size_t)u * n_frames + P.f0[j]) * DSS_COND_STRIDE + 3 * NA + NB3 + (lane & (DSS_LPC_ORDER - 1))];
#pragma unroll
            for (int q = 0; q < DSS_LPC_ORDER; ++q)
                pred -= __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, ls_lane), q)) *
                        __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, lpc_lane), q));
            const int su = dss_lin2ulaw(__builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, ls_lane), 0)));
            const int pu = dss_lin2ulaw(pred);
            if (lane == 0) { UJ.idx[0] = su; UJ.idx[1] = pu; UJ.idx[2] = last_exc; }
            const uint32_t r0 = dss_kiss99_rand(rng);                                   // thresholds of sample 0
            const uint32_t r1 = dss_kiss99_rand(rng);
            if (lane < 8) {
                const uint32_t r = lane < 4 ? r0 : r1;
                UJ.thr[lane] = m.logit_table[(r >> (8 * (lane & 3))) & 0xFF];
            }
            // inputs of sample 0's speculation: its prediction, the history as it stands, the LPC of sample 1's frame
            // (the same frame: a frame has 160 samples)
            if (lane < DSS_LPC_ORDER) { UJ.spec_lpc[lane] = lpc_lane; UJ.spec_ls[lane] = ls_lane; }
        }
        if (lane < DSS_LPC_ORDER) UJ.ls[lane] = ls_lane;
        if (lane == 0) {
            UJ.spec_pred = pred; UJ.pred = pred; UJ.deemph = b.deemph[u]; UJ.last_exc = last_exc;
            UJ.rng[0] = rng.z; UJ.rng[1] = rng.w; UJ.rng[2] = rng.jsr; UJ.rng[3] = rng.jcong;
        }
    }
    __builtin_amdgcn_s_setprio(1);
    __syncthreads();                                                 // prologue barrier
    unsigned long long sacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tprev = STAMP ? __builtin_readcyclecounter() : 0;
    int a = 0, k = 0;
    for (int s = 0; s < total_slots; ++s) {
        const int bq = a == 0 ? U - 1 : a - 1;
        const int cq = bq == 0 ? U - 1 : bq - 1;
        const int kb = bq < a ? k : k - 1;
        const int kc = cq < a ? k : k - 1;
        // ---- bookkeeping inputs that come from L2: issued first, used last -------------------------------------------
        const bool do_book = s >= 1 && kb >= 1 && kb <= mu_pick(P.n, bq);      // finalise sample kb-1 of utterance b
        const int jb = kb - 1;                                        // ... that sample
        float lp_next = 0.f;
        if (do_book && jb + 1 < mu_pick(P.n, bq)) {
            // LPC in force for the prediction of sample jb+2 (the speculation of sample jb+1 needs it); the last sample's
            // speculation is never used, so clamp to the last frame
            int fr = (jb + 2 < mu_pick(P.n, bq) ? jb + 2 : jb + 1) / DSS_FRAME_SIZE;
            lp_next = b.frame_out[((size_t)mu_pick(P.utt, bq) * n_frames + mu_pick(P.f0, bq) + fr) * DSS_COND_STRIDE + 3 * NA + NB3 + (lane & (DSS_LPC_ORDER - 1))];
        }
        // ---- GRU B, second half + gates, utterance c, sample kc ------------------------------------------------------------
        if (s >= 2 && kc >= 0 && kc < mu_pick(P.n, cq)) {
            MuUtt &UC = UT[cq];
            float rec = gbb1;
#pragma unroll
            for (int j = 0; j < NB; ++j) rec += S.gb_wrec[j * NB3 + row] * UC.state_b[j];
            const float sb_old = UC.state_b[lane & (NB - 1)];
            float acc = UC.gb_acc[lane];
            const float *an = UC.state_a + GBH;
            DSS_GB_CHAIN(an, GBH)
            {   // gates: lanes 0..15 z, 16..31 r, 32..47 h (see lpcnet_sample.hip)
                const float zr = dss_sigmoid_approx(S.tansig, acc + rec);
                const unsigned zb = __builtin_bit_cast(unsigned, zr);
                const unsigned r_row0 = __builtin_amdgcn_permlane16_swap(zb, 0u, false, false)[1];                  // lanes 0..15 <- 16..31
                const float r_for_h = __builtin_bit_cast(float, __builtin_amdgcn_permlane32_swap(0u, r_row0, false, false)[0]);  // 32..47 <- 0..15
                const float z_for_h = __builtin_bit_cast(float, __builtin_amdgcn_permlane32_swap(0u, zb, false, false)[0]);      // 32..47 <- 0..15
                float hh = acc + rec * r_for_h;
                hh = dss_tanh_approx(S.tansig, hh);
                if (lane >= 2 * NB && lane < NB3) UC.state_b[lane - 2 * NB] = z_for_h * sb_old + (1 - z_for_h) * hh;
            }
        }
        MU_STAMP(0)
        // ---- scalar recurrences of utterance b: the excitation of sample jb has been decided (bits of last slot) ------------
        if (do_book) {
            MuUtt &UB = UT[bq];
            int exc;
            MU_TREE_WALK(exc, UB.bits)
            const int i = jb % DSS_FRAME_SIZE, f = mu_pick(P.f0, bq) + jb / DSS_FRAME_SIZE;
            float pcm = UB.pred + S.ulaw2lin[exc];
            // signal history: element j lives in lane j; shift by one lane (row_shr:1), lane 0 takes the new sample
            float ls_lane = UB.ls[lane & (DSS_LPC_ORDER - 1)];
            ls_lane = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, pcm),
                                         __builtin_bit_cast(int, ls_lane), 0x111, 0xf, 0xf, false));
            if (lane < DSS_LPC_ORDER) UB.ls[lane] = ls_lane;
            float deemph = UB.deemph;
            pcm += 0.85f * deemph;
            deemph = pcm;
            if (pcm < -32767) pcm = -32767;
            if (pcm > 32767) pcm = 32767;
            if (lane == 0) { UB.pcm[i] = (short)(int)floor(.5 + (double)pcm); UB.deemph = deemph; UB.last_exc = exc; }
            if (jb + 1 < mu_pick(P.n, bq)) {
                const float pred_next = UB.spec_tab_pred[exc];               // speculated for every possible excitation
                DssKiss99 rng = {UB.rng[0], UB.rng[1], UB.rng[2], UB.rng[3]};
                const uint32_t r0 = dss_kiss99_rand(rng);                    // thresholds of sample jb+1
                const uint32_t r1 = dss_kiss99_rand(rng);
                if (lane < 8) {
                    const uint32_t r = lane < 4 ? r0 : r1;
                    UB.thr[lane] = m.logit_table[(r >> (8 * (lane & 3))) & 0xFF];
                }
                if (lane < DSS_LPC_ORDER) { UB.spec_lpc[lane] = lp_next; UB.spec_ls[lane] = ls_lane; }
                if (lane == 0) {
                    UB.pred = pred_next; UB.spec_pred = pred_next;
                    UB.rng[0] = rng.z; UB.rng[1] = rng.w; UB.rng[2] = rng.jsr; UB.rng[3] = rng.jcong;
                }
            }
            if (i == DSS_FRAME_SIZE - 1) {       // wave 7 owns the frame buffer: LDS operations of one wave are ordered
                short *pcm_frame = pcm_out + ((size_t)mu_pick(P.utt, bq) * n_frames + f) * DSS_FRAME_SIZE;
                for (int q = lane; q < DSS_FRAME_SIZE / 2; q += 64)
                    reinterpret_cast<int *>(pcm_frame)[q] = reinterpret_cast<const int *>(UB.pcm)[q];
            }
        }
        MU_STAMP(1)
        __syncthreads();                                                                // slot barrier
        MU_STAMP(2)
        if (++a == U) { a = 0; ++k; }
    }
    if (STAMP && lane == 0 && blockIdx.x == 0 && b.trace_pcm)
        for (int q = 0; q < 8; ++q) b.trace_pcm[wave * 8 + q] = (float)sacc[q];
    // ---- write the persistent state back ------------------------------------------------------------------------------
#pragma unroll
    for (int j = 0; j < MU_MAX; ++j) {
        const int u = P.utt[j];
        if (j >= U || u < 0) continue;
        MuUtt &UJ = UT[j];
        if (lane < NB) b.gru_b_state[(size_t)u * NB + lane] = UJ.state_b[lane];
        if (lane < DSS_LPC_ORDER) b.last_sig[(size_t)u * DSS_LPC_ORDER + lane] = UJ.ls[lane];
        if (lane == 0) {
            b.deemph[u] = UJ.deemph;
            b.last_exc[u] = UJ.last_exc;
            b.rng[u * 4 + 0] = UJ.rng[0]; b.rng[u * 4 + 1] = UJ.rng[1]; b.rng[u * 4 + 2] = UJ.rng[2]; b.rng[u * 4 + 3] = UJ.rng[3];
        }
    }
}

size_t dss_multi_lds_bytes(const DssModelDev &m, int U)
{
    return (((size_t)m.hblk_floats * sizeof(float) + 15) & ~(size_t)15) + (size_t)U * sizeof(MuUtt);
}

// Largest U in {4, 3} whose LDS image fits beside the model's h-gate blocks (0: none; the caller uses the latency kernel)
int dss_multi_max_u(const DssModelDev &m)
{
    if (!m.fast_ok || m.ext) return 0;                       // z/r tails and long h lists: latency kernel only
    const size_t cap = 160 * 1024 - sizeof(MuShared);
    for (int U = MU_MAX; U >= 3; --U)
        if (dss_multi_lds_bytes(m, U) <= cap) return U;
    return 0;
}

int dss_launch_sample_network_multi(const DssModelDev &m, DssBatchDev &b, int n_utts, int n_frames, short *d_pcm, int U,
                                    int stamp, hipStream_t s)
{
    if (U < 3 || U > MU_MAX || U > dss_multi_max_u(m)) { dss_set_error("multi-utterance kernel: U = %d not available for this model", U); return DSS_EINVAL; }
    const size_t dyn = dss_multi_lds_bytes(m, U);
    const bool z10 = m.zr_cap <= 10;
    static std::mutex attr_mu;
    static unsigned long long attr_set = 0;
    int dev = 0;
    DSS_HIP_CHECK(hipGetDevice(&dev));
    {
        std::lock_guard<std::mutex> lk(attr_mu);
        if (!(attr_set >> (dev & 63) & 1)) {
            const int cap = 160 * 1024 - (int)sizeof(MuShared);
            DSS_HIP_CHECK(hipFuncSetAttribute((const void *)lpcnet_sample_multi_kernel<10, false>, hipFuncAttributeMaxDynamicSharedMemorySize, cap));
            DSS_HIP_CHECK(hipFuncSetAttribute((const void *)lpcnet_sample_multi_kernel<12, false>, hipFuncAttributeMaxDynamicSharedMemorySize, cap));
            DSS_HIP_CHECK(hipFuncSetAttribute((const void *)lpcnet_sample_multi_kernel<10, true>, hipFuncAttributeMaxDynamicSharedMemorySize, cap));
            attr_set |= 1ull << (dev & 63);
        }
    }
    const int grid = (n_utts + U - 1) / U;
    if (stamp && z10) hipLaunchKernelGGL((lpcnet_sample_multi_kernel<10, true>), dim3(grid), dim3(512), dyn, s, m, b, n_frames, d_pcm, n_utts, U);
    else if (z10) hipLaunchKernelGGL((lpcnet_sample_multi_kernel<10, false>), dim3(grid), dim3(512), dyn, s, m, b, n_frames, d_pcm, n_utts, U);
    else hipLaunchKernelGGL((lpcnet_sample_multi_kernel<12, false>), dim3(grid), dim3(512), dyn, s, m, b, n_frames, d_pcm, n_utts, U);
    DSS_HIP_CHECK(hipGetLastError());
    return DSS_OK;
}
