// csrc/lpcnet_sample_pkh.hip -- LPCNet sample-rate network, latency form, round 4: one persistent workgroup per utterance
// with BOTH co-critical sides of the B..C phase cut (gfx950).
//
// Same algorithm and bit-identical results as lpcnet_sample.hip (xiph/LPCNet src/lpcnet.c lpcnet_synthesize_tail_impl() +
// run_sample_network(), src/nnet.c compute_gru_a_input / compute_sparse_gru / compute_gruB / sample_mdense, generic float
// path of src/vec.h; reached through extensions/lpcnet/cLPCNet.pxd:13, local/units.py:531-538, local/training.py:165-207).
// Two things are different, and they are different together because either alone gains nothing (profiles/
// r2_knockout_experiments.txt: the GRU B relay and the GRU A waves' work for the next sample are co-critical):
//
//  (a) The h-gate chain of GRU A runs on THREE waves instead of six: a lane carries rows q and q+4 of one 8x4 block as
//      the two halves of packed fp32 instructions (the block's four state values broadcast with op_sel, the weight
//      pairs (W[q][c], W[q+4][c]) side by side in the LDS record), 16 row groups per wave.  One instruction per
//      multiply-add instead of 1.5, the chain order of every row untouched.  Those waves (0..2) carry no dual-FC and
//      no speculation, and they do not take part in the hand-over C: they have until barrier D.
//  (b) GRU B is a ping-pong relay between waves 7 and 6 with no resident weights (the pair kernel's form): while one
//      wave runs the sums of its stage -- the one dependent chain -- the other forms the products of its next stage,
//      the weights streamed from L2 two stages ahead (m.gb_w_quad).  The chain approaches one dependent add per input.
//
// Between B and C a SIMD that carries a relay wave should carry nothing else that is heavy: a packed-fp32 instruction holds
// the SIMD for 4.4 cycles, and the first build of this kernel, with h-chain waves beside the relay waves, ran the relay at
// 4.6 k cycles instead of 4.1 k.  So the roles are laid out by SIMD (waves w and w + 4 share one):
//   SIMD a: wave 0  GRU A D..B; z/r block products + packed h chain (longest lists) B..D
//           wave 4  GRU A D..B; z/r block products + one speculation pass; dual-FC C..D
//   SIMD b: wave 1  GRU A D..B; z/r block products + packed h chain B..D
//           wave 5  GRU A D..B; z/r block products + packed h chain (shortest lists) B..D
//   SIMD c: wave 2  GRU A D..B (heaviest z/r row groups); z/r block products + one speculation pass; dual-FC C..D
//           wave 6  GRU B odd stages; one speculation pass; dual-FC C..D
//   SIMD d: wave 3  as wave 2
//           wave 7  GRU B even stages + gates; scalar recurrences (as in lpcnet_sample.hip)
// "Barrier C" is an LDS word that wave 7 sets behind GRU B's new state; waves 0, 1, 4, 5 never wait for it.  Per sample:
// barriers B and D.  Models that need the extended paths (z/r tails, long h lists) or more than 10 z/r register slots stay
// on lpcnet_sample.hip.
#include <cstddef>
#include <mutex>

#include "lpcnet_sample_common.h"
#undef HC

#define DSS_PKH_HBLK_BYTES 151552          // dynamic LDS left beside PkhLds (160 KB per CU)
// Ping-pong relay: the 96 blocks of four inputs go through stages 0..6, even stages on wave 7, odd ones on wave 6.  Both
// waves start multiplying at barrier B, and wave 7's first products are the one part of the relay nothing hides, so stage 0
// is half a stage: 8, 16, 16, 16, 16, 16, 8 blocks.
#define DSS_QR_G0(S) ((S) == 0 ? 0 : 8 + 16 * ((S) - 1))            // first block of stage S
#define DSS_QR_NB(S) (((S) == 0 || (S) == 6) ? 8 : 16)             // its blocks

struct PkhLds {
    float state_a[2][NA + 4];             // double-buffered GRU A state; "column 96" of either buffer is four zeros
    float gb_wrec[NB * NB3];              // GRU B recurrent weights [16][48]
    float tansig[208];
    float ulaw2lin[256];
    float spec_tab_pred[256];             // speculation over all 256 excitation values: next sample's prediction
    unsigned short spec_tab_idx[256];     //   and its two mu-law indices (su | pu << 8)
    float spec_ls[DSS_LPC_ORDER];         // inputs of the speculation, published by wave 7
    float spec_lpc[DSS_LPC_ORDER];
    float spec_pred;
    float pad1[3];
    float gb_acc[64][2];                  // GRU B running sums handed between the relay waves: (sum, 8 * sample number + stages done)
    float ah[NA];                         // h-gate pre-activation: written by the h lanes (waves 0..2), read by the unit's z/r lane
    float state_b[NB];
    int c_flag;                           // "barrier C": number of the sample whose state_b is complete (wave 7 -> dual-FC waves);
    int pad[3];                           //   directly behind state_b: a poll reads the word and the state in one go
    float thr[8];
    unsigned bits[8];                     // decision bit of every tree node (256 bits)
    int idx[4];                           // last_sig_ulaw, pred_ulaw, last_exc (first sample of a call)
    short pcm[DSS_FRAME_SIZE];
};
static_assert(sizeof(PkhLds) % 16 == 0, "dynamic LDS must start 16-byte aligned");
static_assert(offsetof(PkhLds, c_flag) == offsetof(PkhLds, state_b) + 64, "c_flag directly behind state_b");
static_assert(sizeof(PkhLds) + DSS_PKH_HBLK_BYTES <= 160 * 1024, "LDS budget");

// ---- packed-rows h chain ----------------------------------------------------------------------------------------------
// Lane = rows q and q+4 (q = lane & 3) of row group lane >> 2.  Record of one block (128 B): [half 2][q 4][4 floats] =
// (W[q][c], W[q+4][c], W[q][c+1], W[q+4][c+1]) for c = 2 * half: two ds_read_b128 per lane; the block's four state values
// are one more (the same address in the four lanes of a group).  The products of block s+1 are formed in the shadow of
// the four dependent sums of block s (DSS_PK_STEP4 with the broadcast operand = the state pair).
#define DSS_QH_LOAD(S, BUF)                                                                      \
    {                                                                                            \
        HX[BUF] = *reinterpret_cast<const f32x4 *>(xbase + DSS_H_COL(S) * 16);                   \
        HW[BUF].a = *reinterpret_cast<const f32x4 *>(hw + (S) * 128);                            \
        HW[BUF].c = *reinterpret_cast<const f32x4 *>(hw + (S) * 128 + 64);                       \
    }
#define DSS_QH_STEP(S) DSS_PK_STEP4(ah, HP[(S) & 1], HP[((S) + 1) & 1], HW[((S) + 1) % 3], HX[((S) + 1) % 3].lo, HX[((S) + 1) % 3].hi)
// Blocks S and S + 1 (nh is even: both exist whenever S < nh).  On entry the products of block S are formed and the
// operands of block S + 1 are in flight.  The operands of blocks S + 2 and S + 3 are fetched UNCONDITIONALLY -- past the end
// of the lists they are the following records times "column 96" (the image is padded by two records), never used.  With a
// conditional fetch the two paths into every step differ in the number of LDS reads outstanding, and the compiler's wait
// before the step becomes lgkmcnt(0): the whole LDS latency per step (measured: 186 cycles per step instead of ~100).
// No early exit either: fourteen returns out of this sequence sent the register allocator into hundreds of spills.
#define DSS_QH_PAIR(S)                                                                           \
    if constexpr ((S) < HC) {                                                                    \
        if ((S) < nh) {                                                                          \
            DSS_QH_LOAD((S) + 2 < HC ? (S) + 2 : 0, ((S) + 2) % 3)                               \
            __builtin_amdgcn_sched_barrier(0);                                                   \
            DSS_QH_STEP(S);                                                                      \
            __builtin_amdgcn_sched_barrier(0);                                                   \
            DSS_QH_LOAD((S) + 3 < HC ? (S) + 3 : 0, ((S) + 3) % 3)                               \
            __builtin_amdgcn_sched_barrier(0);                                                   \
            if ((S) + 2 < nh) {                                                                  \
                DSS_QH_STEP((S) + 1);                                                            \
            } else {                                                                             \
                DSS_PK_ADD4(ah, HP[((S) + 1) & 1]);                                              \
            }                                                                                    \
            __builtin_amdgcn_sched_barrier(0);                                                   \
        }                                                                                        \
    }
// h-gate chains of rows uh and uh + 4: ah = rbh + dgh * state on entry; nh (even, wave-uniform) slots; returns the two
// pre-activations.  Written out, not a loop: an inline-asm block is a convergent call, a loop with a data-dependent exit
// around it is not unrolled, and the column ids (PH) must be indexed with constants.
template <int HC>
__device__ __forceinline__ f32x2 dss_qh_chain(const char *xbase, const char *hw, const unsigned (&PH)[HC / 4], int nh, f32x2 ah)
{
    f32x4 HX[3];
    PairX HW[3];
    f32x2 HP[2][4];
    DSS_QH_LOAD(0, 0)
    DSS_QH_LOAD(1, 1)
    DSS_PK_MUL4(HP[0], HW[0], HX[0].lo, HX[0].hi);
    DSS_QH_PAIR(0)  DSS_QH_PAIR(2)  DSS_QH_PAIR(4)  DSS_QH_PAIR(6)  DSS_QH_PAIR(8)  DSS_QH_PAIR(10) DSS_QH_PAIR(12)
    DSS_QH_PAIR(14) DSS_QH_PAIR(16) DSS_QH_PAIR(18) DSS_QH_PAIR(20) DSS_QH_PAIR(22) DSS_QH_PAIR(24) DSS_QH_PAIR(26)
    static_assert(HC <= 28, "add DSS_QH_PAIR lines");
    return ah;
}
#define DSS_QH_CHAIN(XBUF)                                                                       \
    {                                                                                            \
        const f32x2 ah0 = rbh2 + dgh2 * (f32x2){(XBUF)[uh], (XBUF)[uh + 4]};                     \
        const f32x2 ah = dss_qh_chain<HC>(reinterpret_cast<const char *>(XBUF), hw, PH, nh, ah0); \
        L.ah[uh] = ah.x;                                                                         \
        L.ah[uh + 4] = ah.y;                                                                     \
    }

// ---- GRU B: a ping-pong relay (one utterance) --------------------------------------------------------------------------
// A stage = NBK blocks of four inputs, first block G0.  Its state values (one ds_read_b128 per block, every lane the same
// address) are fetched eight blocks at a time: the first eight BEFORE the wave waits for its turn and runs the previous
// stage's sums (DSS_QR_READ), the second eight while the first are multiplied.  The weights of block g sit in RW[g] (four
// inputs of this lane's row, m.gb_w_quad); once a block is multiplied its weight registers are reloaded with the block the
// wave multiplies there next (RELOAD(g): a block of the wave's next stage, or of the stage after), so the L2 latency never
// shows.  Per block: one ds_read_b128, two v_pk_mul_f32, one global_load_dwordx4 (scalar base + lane offset + immediate).
#define DSS_QR_READ(G0)                                                                          \
    {                                                                                            \
        _Pragma("unroll") for (int g = 0; g < 8; ++g) RX[g] = *reinterpret_cast<const f32x4 *>(an + 4 * ((G0) + g)); \
    }
#define DSS_QR_WLOAD(BLK) *reinterpret_cast<const f32x4 *>(wq + (size_t)wvo[(BLK) >> 3] + (ptrdiff_t)(((BLK) & 7) * 1024 - 4096))
#define DSS_QR_MUL(NBK, G0, RELOAD)                                                              \
    {                                                                                            \
        DSS_WAIT_LGKM(0);                                                                        \
        __builtin_amdgcn_sched_barrier(0);                                                       \
        _Pragma("unroll") for (int g = 0; g < 8; ++g) {                                          \
            PS[2 * g] = RW[g].lo * RX[g].lo;                                                     \
            PS[2 * g + 1] = RW[g].hi * RX[g].hi;                                                 \
            asm volatile("" : "+v"(PS[2 * g]), "+v"(PS[2 * g + 1]));   /* pinned: left alone, the products sink below the hand-over wait */ \
            __builtin_amdgcn_sched_barrier(0);                                                   \
            RW[g] = DSS_QR_WLOAD(RELOAD(g));                                                     \
            if ((NBK) > 8) RX[g] = *reinterpret_cast<const f32x4 *>(an + 4 * ((G0) + 8 + g));    \
            __builtin_amdgcn_sched_barrier(0);                                                   \
        }                                                                                        \
        if ((NBK) > 8) {                                                                         \
            DSS_WAIT_LGKM(0);                                                                    \
            __builtin_amdgcn_sched_barrier(0);                                                   \
            _Pragma("unroll") for (int g = 8; g < (NBK); ++g) {                                  \
                PS[2 * g] = RW[g].lo * RX[g - 8].lo;                                             \
                PS[2 * g + 1] = RW[g].hi * RX[g - 8].hi;                                         \
                asm volatile("" : "+v"(PS[2 * g]), "+v"(PS[2 * g + 1]));                         \
                __builtin_amdgcn_sched_barrier(0);                                               \
                RW[g] = DSS_QR_WLOAD(RELOAD(g));                                                 \
                __builtin_amdgcn_sched_barrier(0);                                               \
            }                                                                                    \
        }                                                                                        \
    }
// the sums of a stage: one dependent chain, input order
#define DSS_QR_ADD(NBK)                                                                          \
    {                                                                                            \
        _Pragma("unroll") for (int g = 0; g < (NBK); ++g) {                                      \
            acc += PS[2 * g].x;                                                                  \
            acc += PS[2 * g].y;                                                                  \
            acc += PS[2 * g + 1].x;                                                              \
            acc += PS[2 * g + 1].y;                                                              \
        }                                                                                        \
        __builtin_amdgcn_sched_barrier(0);                                                       \
    }
// Hand the running sums to the other relay wave / take them over.  One 8-byte LDS word per lane carries the sum and its
// tag (8 * sample number + stages done): the taker polls its own lane's word, and the read that sees every lane's tag has
// the sums in it -- one LDS round trip per hand-over instead of flag, wait, sums.  Hand-written ds_ instructions: a
// volatile vector access becomes a flat_load here, and __builtin_bit_cast of a vector ELEMENT reads element 0 with this
// clang (the first build of this polled the sum instead of the tag, for ever).
#define DSS_QR_PUBLISH(V)                                                                        \
    {                                                                                            \
        const int tagi_ = (int)(V);                                                              \
        const f32x2 pw_ = {acc, __builtin_bit_cast(float, tagi_)};                               \
        asm volatile("ds_write_b64 %0, %1" :: "v"(gb_addr), "v"(pw_) : "memory");                \
    }
#define DSS_QR_AWAIT(V)                                                                          \
    {                                                                                            \
        unsigned tag_;                                                                           \
        do {                                                                                     \
            asm volatile("ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(pv_) : "v"(gb_addr) : "memory"); \
            const float tf_ = pv_.y;                                                             \
            tag_ = __builtin_bit_cast(unsigned, tf_);                                            \
        } while (__any(tag_ != (unsigned)(V)));                                                  \
        acc = pv_.x;                                                                             \
    }

// one candidate excitation value of the speculation (inputs published by wave 7 before barrier B): the output sample,
// the next LPC prediction (16 sequential subtractions, reference order) and both mu-law indices.  Same expressions, same
// order as lpcnet_synthesize_tail_impl().
__device__ __forceinline__ void dss_pkh_speculate(PkhLds &L, int cand, float u2l_c)
{
    const float pcm_c = L.spec_pred + u2l_c;
    float pc = 0;
    pc -= pcm_c * L.spec_lpc[0];
#pragma unroll
    for (int j = 1; j < DSS_LPC_ORDER; ++j) pc -= L.spec_ls[j - 1] * L.spec_lpc[j];
    const int su_c = dss_lin2ulaw(pcm_c), pu_c = dss_lin2ulaw(pc);
    L.spec_tab_pred[cand] = pc;
    L.spec_tab_idx[cand] = (unsigned short)(su_c | (pu_c << 8));
}

// the dual-FC constants of one tree node (lane = node): the two dense layers of the node run as the two halves of packed
// fp32 instructions: weights as pairs (layer 0 input j, layer 1 input j), the two running sums as one register pair
struct PkhFc {
    f32x2 fw[NB];
    float fb0, fb1, ff0, ff1;
    int level;
};
__device__ __forceinline__ void dss_pkh_fc_load(PkhFc &fc, const DssModelDev &m, int node)
{
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        fc.fw[j].x = m.fc_w[(size_t)node * 2 * NB + j];
        fc.fw[j].y = m.fc_w[(size_t)node * 2 * NB + NB + j];
    }
    fc.fb0 = m.fc_bias[node]; fc.fb1 = m.fc_bias[DSS_FC_OUT + node];
    fc.ff0 = m.fc_factor[node]; fc.ff1 = m.fc_factor[DSS_FC_OUT + node];
    fc.level = 31 - __clz(node | 1);                             // FC node = (1 << level) | prefix
}
// sample_mdense for node `node` (all 255 nodes in parallel on four waves); k = which of the four dual-FC waves; bq = GRU B's
// new state
template <bool TRACE>
__device__ __forceinline__ void dss_pkh_fc(PkhLds &L, const PkhFc &fc, const DssBatchDev &b, const f32x4 (&bq)[NB / 4], int node,
                                           int k, int lane, size_t o)
{
    const float thr_lv = L.thr[fc.level];                        // issued first, used last
    f32x2 s12 = {fc.fb0, fc.fb1};
#pragma unroll
    for (int j4 = 0; j4 < NB / 4; ++j4) {
        const f32x4 bj = bq[j4];
        // per input: one packed product for both layers, one packed sum (each half rounds on its own, exactly as the two
        // scalar chains); the four products first: a packed result needs a wait state before it can be read
        const f32x2 q0 = fc.fw[4 * j4 + 0] * (f32x2){bj.x, bj.x};
        const f32x2 q1 = fc.fw[4 * j4 + 1] * (f32x2){bj.y, bj.y};
        const f32x2 q2 = fc.fw[4 * j4 + 2] * (f32x2){bj.z, bj.z};
        const f32x2 q3 = fc.fw[4 * j4 + 3] * (f32x2){bj.w, bj.w};
        s12 += q0;
        s12 += q1;
        s12 += q2;
        s12 += q3;
    }
    float s1 = s12.x, s2 = s12.y;
    float t1, t2;
    dss_tanh_approx2(L.tansig, s1, s2, t1, t2);
    s1 = fc.ff0 * t1;
    s2 = fc.ff1 * t2;
    s1 += s2;
    bool bit = thr_lv < s1;
    if constexpr (TRACE) {                   // teacher forcing (tests): record every logit, bend the walk
        if (b.trace_logits) b.trace_logits[o * 256 + node] = node ? s1 : 0.f;
        if (b.force_exc) {
            const int v = b.force_exc[o];                                   // bits b7..b0, b7 decided at level 0
            if ((node ^ (1 << fc.level)) == (v >> (8 - fc.level))) bit = (v >> (7 - fc.level)) & 1;
        }
    }
    const unsigned long long mask = __ballot(bit);
    if (lane == 0) { L.bits[2 * k] = (unsigned)mask; L.bits[2 * k + 1] = (unsigned)(mask >> 32); }
}
// wave 7: fold the sampled excitation into the signal history and emit the PCM sample (lpcnet_synthesize_tail_impl)
#define DSS_QS_UPDATE()                                                                          \
    {                                                                                            \
        float pcm = upd_pred + L.ulaw2lin[upd_exc];                                              \
        if (TRACE && lane == 0) {                                                                \
            const size_t o = ((size_t)utt * n_frames + f) * DSS_FRAME_SIZE + upd_i;              \
            b.trace_exc[o] = (float)upd_exc;                                                     \
            b.trace_pcm[o] = pcm;                                                                \
        }                                                                                        \
        /* signal history: element j lives in lane j; shift by one lane (row_shr:1), lane 0 keeps the new sample */ \
        ls_lane = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, pcm),   \
                                     __builtin_bit_cast(int, ls_lane), 0x111, 0xf, 0xf, false));  \
        last_exc = upd_exc;                                                                      \
        pcm += 0.85f * deemph;                                                                   \
        deemph = pcm;                                                                            \
        if (pcm < -32767) pcm = -32767;                                                          \
        if (pcm > 32767) pcm = 32767;                                                            \
        if (lane == 0) L.pcm[upd_i] = (short)(int)floor(.5 + (double)pcm);                       \
        upd_pending = false;                                                                     \
    }

// =====================================================================================================
// role A: GRU A for the lane's unit (D..B), its z/r block products for the next sample (after B), and then
//   ROLE_H    (waves 0, 1, 5): the packed h chain of 16 row groups, until barrier D;
//   ROLE_FC   (waves 2, 3, 4): one speculation pass, then the dual-FC of 64 tree nodes once wave 7 has published GRU B's state.
// =====================================================================================================
#define ROLE_H 0
#define ROLE_FC 2
template <bool TRACE, bool STAMP, int Z, int ROLE>
__device__ __forceinline__ void dss_pkh_role_a(PkhLds &L, float *hblk_lds, const DssModelDev &m, const DssBatchDev &b,
                                               int n_frames, int utt, int slot, int nf, int fc0, int tid, int wave, int lane)
{
    constexpr int HC = DSS_HC;
    constexpr bool HROLE = ROLE == ROLE_H;
    const int unit = m.pk_unit_of[tid];                          // z/r chains + gates of this unit
    const int nzr = __builtin_amdgcn_readfirstlane(m.pk_wave_nzr[wave]);
    // packed h chain: h wave 0, 1, 2 = wave 0, 1, 5; rows uh and uh + 4
    const int hidx = wave == 5 ? 2 : wave, htid = hidx * 64 + lane;
    const int uh = HROLE ? m.pk_unit[htid] : 0;
    const int nh = HROLE ? __builtin_amdgcn_readfirstlane(m.pk_nh[hidx]) : 0;
    const char *hw = reinterpret_cast<const char *>(hblk_lds + (HROLE ? m.pk_hoff[htid >> 2] : 0)) + (lane & 3) * 16;
    f32x4 WZ[2 * ZRC];                                           // [0,ZRC) z slots, [ZRC,2ZRC) r slots
    unsigned PZ[(2 * ZRL + 3) / 4], PH[HROLE ? HC / 4 : 1];
#pragma unroll
    for (int s = 0; s < 2 * ZRC; ++s) {
        const int sl = s < ZRC ? s : ZRL + (s - ZRC);            // layout numbering
        WZ[s].x = m.pk_zr_w[((size_t)sl * 4 + 0) * NA + tid];
        WZ[s].y = m.pk_zr_w[((size_t)sl * 4 + 1) * NA + tid];
        WZ[s].z = m.pk_zr_w[((size_t)sl * 4 + 2) * NA + tid];
        WZ[s].w = m.pk_zr_w[((size_t)sl * 4 + 3) * NA + tid];
    }
#pragma unroll
    for (int s = 0; s < (2 * ZRL + 3) / 4; ++s) PZ[s] = m.pk_zr_col[(size_t)s * NA + tid];
    if constexpr (HROLE) {
#pragma unroll
        for (int s = 0; s < HC / 4; ++s) PH[s] = m.pk_hcol[(size_t)s * 192 + htid];
    }
    const float rbz = m.gru_a_rbias[unit], rbr = m.gru_a_rbias[NA + unit];
    const float dgz = m.gru_a_diag[unit], dgr = m.gru_a_diag[NA + unit];
    f32x2 rbh2 = {0.f, 0.f}, dgh2 = {0.f, 0.f};
    if constexpr (HROLE) {
        rbh2 = (f32x2){m.gru_a_rbias[2 * NA + uh], m.gru_a_rbias[2 * NA + uh + 4]};
        dgh2 = (f32x2){m.gru_a_diag[2 * NA + uh], m.gru_a_diag[2 * NA + uh + 4]};
    }
    // dual-FC (waves 2, 3, 4 = dual-FC waves 0, 1, 3; wave 6 is 2) and this lane's excitation candidate of the speculation
    const int fck = wave == 4 ? 3 : wave - 2, node = fck * 64 + lane;
    const int cand = (wave - 1) * 64 + lane;                     // wave 2: 64.., wave 3: 128.., wave 4: 192..
    PkhFc fc;
    float u2l_c = 0.f;
    if constexpr (ROLE == ROLE_FC) dss_pkh_fc_load(fc, m, node);
    if constexpr (!HROLE) u2l_c = L.ulaw2lin[cand & 255];
    const bool recur_first = m.h.gru_a_order == DSS_GRUA_RECUR_FIRST;     // wave-uniform (kernel argument)
    int cur = 0, seq = 0;
    float st = L.state_a[0][unit];
    unsigned long long sa[8] = {0, 0, 0, 0, 0, 0, 0, 0}, ta = 0;   // diagnostic build only
    f32x4 PR[2 * ZRC];                                           // z/r block products of the coming sample
    bool first_sample = true;
    DSS_ZR_PRODUCTS(L.state_a[0])                                // first sample of this call
    if constexpr (HROLE) DSS_QH_CHAIN(L.state_a[0])
    __syncthreads();                                             // L.ah of every unit visible to its z/r lane

    for (int f = 0; f < nf; ++f) {
        if (fc0 + f < DSS_FEATURES_DELAY) continue;              // silent frame: decoder state untouched
        const float *fo = b.frame_out + ((size_t)utt * n_frames + f) * DSS_COND_STRIDE;     // wave-uniform base
        const float cz = fo[(unsigned)unit], cr = fo[(unsigned)(NA + unit)], ch = fo[(unsigned)(2 * NA + unit)];
        for (int i = 0; i < DSS_FRAME_SIZE; ++i) {
            // keep the packed column ids opaque so the per-slot unpacking is not hoisted out of the sample loop
#pragma unroll
            for (int k = 0; k < (2 * ZRL + 3) / 4; ++k) asm volatile("" : "+v"(PZ[k]));
            if constexpr (HROLE) {
#pragma unroll
                for (int k = 0; k < HC / 4; ++k) asm volatile("" : "+v"(PH[k]));
            }
            ++seq;
            float az = rbz + dgz * st;                           // compute_sparse_gru, before the input term
            float ar = rbr + dgr * st;
            const float ahv = L.ah[unit];                        // this unit's h-gate pre-activation (its h lane, B..D)
            // The three embedding indices.  First sample of a call: wave 7 computes them (L.idx, barrier A).  Every later
            // sample: this wave walks the sampling tree itself and looks the speculated indices up.
            int si, pi, ei;
            if (first_sample) {
                __syncthreads();                                                    // barrier A (first sample only)
                si = L.idx[0]; pi = L.idx[1]; ei = L.idx[2];
                first_sample = false;
            } else {
                int exc_;
                DSS_TREE_WALK_AT(exc_, L.bits)
                const unsigned sidx = __builtin_amdgcn_readfirstlane((unsigned)L.spec_tab_idx[exc_]);
                si = (int)(sidx & 0xFF); pi = (int)(sidx >> 8); ei = exc_;
            }
            si = __builtin_amdgcn_readfirstlane(si); pi = __builtin_amdgcn_readfirstlane(pi); ei = __builtin_amdgcn_readfirstlane(ei);
            if (STAMP) ta = __builtin_readcyclecounter();
            {
                typedef float f32x3 __attribute__((ext_vector_type(3)));
                const f32x3 es = *reinterpret_cast<const f32x3 *>(m.pk_embed_lane[0] + ((unsigned)si * NA + (unsigned)tid) * 3);
                const f32x3 ep = *reinterpret_cast<const f32x3 *>(m.pk_embed_lane[1] + ((unsigned)pi * NA + (unsigned)tid) * 3);
                const f32x3 ee = *reinterpret_cast<const f32x3 *>(m.pk_embed_lane[2] + ((unsigned)ei * NA + (unsigned)tid) * 3);
                const float es0 = es.x, es1 = es.y, es2 = es.z, ep0 = ep.x, ep1 = ep.y, ep2 = ep.z, ee0 = ee.x, ee1 = ee.y, ee2 = ee.z;
                if (STAMP) { unsigned long long t = __builtin_readcyclecounter(); sa[0] += t - ta; ta = t; }
                const float gz = ((cz + es0) + ep0) + ee0;                          // compute_gru_a_input
                const float gr = ((cr + es1) + ep1) + ee1;
                const float gh = ((ch + es2) + ep2) + ee2;
                // nnet.c 2021 (default): (bias + diag*state) + input, then the blocks in idx order;
                // nnet.c 2019-20 (blob flag): the blocks first, the input last
                if (!recur_first) { az = az + gz; ar = ar + gr; }
                if (STAMP) { asm volatile("" :: "v"(az), "v"(ar)); unsigned long long t = __builtin_readcyclecounter(); sa[1] += t - ta; ta = t; }
                // the block products were formed right after the previous sample's state update; what is left on the
                // critical path are the dependent sums, z and r chains interleaved
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
                if (STAMP) { asm volatile("" :: "v"(az), "v"(ar)); unsigned long long t = __builtin_readcyclecounter(); sa[2] += t - ta; ta = t; }
                float z, r;
                dss_sigmoid_approx2(L.tansig, az, ar, z, r);
                float h = ahv * r + gh;
                h = dss_tanh_approx(L.tansig, h);
                st = z * st + (1 - z) * h;
                L.state_a[cur ^ 1][unit] = st;
                if (STAMP) { asm volatile("" :: "v"(st)); unsigned long long t = __builtin_readcyclecounter(); sa[3] += t - ta; ta = t; }
            }
            __syncthreads();                                                        // barrier B
            if (STAMP) { unsigned long long t = __builtin_readcyclecounter(); sa[4] += t - ta; ta = t; }
#ifndef PKH_NO_PROD
            DSS_ZR_PRODUCTS(L.state_a[cur ^ 1])                  // next sample's z/r block products (sums come later)
#endif
            if constexpr (HROLE) {
#ifndef PKH_NO_H
                DSS_QH_CHAIN(L.state_a[cur ^ 1])                 // next sample's h chains, rows uh and uh + 4
#endif
                if (STAMP) { unsigned long long t = __builtin_readcyclecounter(); sa[5] += t - ta; ta = t; }
            } else {
#ifndef PKH_NO_SPEC
                dss_pkh_speculate(L, cand, u2l_c);
#endif
                if (STAMP) { unsigned long long t = __builtin_readcyclecounter(); sa[5] += t - ta; ta = t; }
                if constexpr (ROLE == ROLE_FC) {
                    f32x4 bq[NB / 4];
                    dss_await_c(L.state_b, seq, bq);                                    // "barrier C": GRU B's new state
                    if (STAMP) { unsigned long long t = __builtin_readcyclecounter(); sa[6] += t - ta; ta = t; }
                    dss_pkh_fc<TRACE>(L, fc, b, bq, node, fck, lane, ((size_t)utt * n_frames + f) * DSS_FRAME_SIZE + i);
                }
            }
            __syncthreads();                                                        // barrier D
            if (STAMP) { unsigned long long t = __builtin_readcyclecounter(); sa[7] += t - ta; ta = t; }
            cur ^= 1;
        }
    }
    __syncthreads();                                                                // final barrier
    if (STAMP && lane == 0 && b.trace_exc)
        for (int k = 0; k < 8; ++k) b.trace_exc[((size_t)utt * 6 + wave) * 8 + k] = (float)sa[k];
    b.gru_a_state[(size_t)slot * NA + unit] = st;
}

// which block a relay wave loads into RW[g] once it has multiplied stage S (see DSS_QR_MUL)
#define DSS_R7_0(g) (24 + (g))                            // wave 7, after stage 0 (blocks 0..7): stage 2's first eight
#define DSS_R7_2(g) (56 + (g))                            //   after stage 2 (24..39): stage 4
#define DSS_R7_4(g) ((g) < 8 ? 88 + (g) : 24 + (g))       //   after stage 4 (56..71): stage 6, and the NEXT sample's stage 2 blocks 32..39
#define DSS_R7_6(g) (g)                                   //   after stage 6 (88..95): the next sample's stage 0
#define DSS_R6_1(g) (40 + (g))                            // wave 6, after stage 1 (8..23): stage 3
#define DSS_R6_3(g) (72 + (g))                            //   after stage 3 (40..55): stage 5
#define DSS_R6_5(g) (8 + (g))                             //   after stage 5 (72..87): the next sample's stage 1
// lane offsets into m.gb_w_quad for the eight-block windows (8 KB each): block b is at wq + wvo[b >> 3] + ((b & 7) * 1024 - 4096),
// an immediate the load instruction carries
#define DSS_QR_SETUP()                                                                           \
    const unsigned gb_addr = dss_lds_addr(&L.gb_acc[lane][0]);                                   \
    const char *wq = reinterpret_cast<const char *>(m.gb_w_quad);                                \
    unsigned wvo[12];                                                                            \
    _Pragma("unroll") for (int k = 0; k < 12; ++k) {                                             \
        wvo[k] = (unsigned)lane * 16 + (unsigned)k * 8192u + 4096u;                              \
        asm volatile("" : "+v"(wvo[k]));                                                         \
    }

// RAGGED: rows name their decoder slot and frame count (b.slot_of / b.count_of); the trace build always honours the lists.
template <bool TRACE, bool STAMP, int Z, bool RAGGED>
__global__ void __launch_bounds__(512)
lpcnet_sample_pkh_kernel(DssModelDev m, DssBatchDev b, int n_frames, short *__restrict__ pcm_out)
{
    __shared__ __attribute__((aligned(16))) PkhLds L;
    extern __shared__ __attribute__((aligned(16))) float hblk_lds[];       // packed h-gate block records (size per model)
    constexpr bool RG = RAGGED || TRACE;
    const int utt = (RG && b.row_of) ? __builtin_amdgcn_readfirstlane(b.row_of[blockIdx.x]) : b.utt0 + (int)blockIdx.x;
    const int slot = (RG && b.slot_of) ? b.slot_of[utt] : utt;                     // decoder state it continues
    const int nf = (RG && b.count_of) ? min(b.count_of[utt], n_frames) : n_frames; // its own frame count
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;

    // ---------------- one-time staging into LDS -------------------------------------------------------
    for (int k = tid * 4; k < m.hblk_pk_floats; k += 512 * 4)
        *reinterpret_cast<f32x4 *>(&hblk_lds[k]) = *reinterpret_cast<const f32x4 *>(&m.hblk_pk[k]);
    for (int k = tid; k < NB * NB3; k += 512) L.gb_wrec[k] = m.gru_b_w_rec[k];
    if (tid < 201) L.tansig[tid] = m.tansig[tid];
    if (tid < 256) L.ulaw2lin[tid] = m.ulaw2lin[tid];
    if (tid < NA) L.state_a[0][tid] = b.gru_a_state[(size_t)slot * NA + tid];
    if (tid < 8) L.state_a[tid >> 2][NA + (tid & 3)] = 0.f;
    if (tid < NB) L.state_b[tid] = b.gru_b_state[(size_t)slot * NB + tid];
    if (tid < 128) L.gb_acc[tid >> 1][tid & 1] = 0.f;
    if (tid == 0) L.c_flag = 0;
    const int fc0 = b.fc0[utt];
    __syncthreads();

    if (wave == 0 || wave == 1 || wave == 5) {
        dss_pkh_role_a<TRACE, STAMP, 8, ROLE_H>(L, hblk_lds, m, b, n_frames, utt, slot, nf, fc0, tid, wave, lane);
    } else if (wave == 4) {
        dss_pkh_role_a<TRACE, STAMP, 8, ROLE_FC>(L, hblk_lds, m, b, n_frames, utt, slot, nf, fc0, tid, wave, lane);
    } else if (wave < 4) {
        dss_pkh_role_a<TRACE, STAMP, Z, ROLE_FC>(L, hblk_lds, m, b, n_frames, utt, slot, nf, fc0, tid, wave, lane);
    } else if (wave == 6) {
        // =====================================================================================================
        // role B1: GRU B, the odd stages of the relay (lane = row: 0..15 z, 16..31 r, 32..47 h); then speculation
        // candidates 0..63 and the dual-FC of tree nodes 128..191
        // =====================================================================================================
        DSS_QR_SETUP()
        f32x4 RW[16];                                // weights of this wave's next stage (stage 1 to begin with)
#pragma unroll
        for (int g = 0; g < 16; ++g) RW[g] = DSS_QR_WLOAD(DSS_R6_5(g));
        PkhFc fc;
        dss_pkh_fc_load(fc, m, 128 + lane);
        const float u2l_c = L.ulaw2lin[lane];
        unsigned long long relay6 = 0, atc6 = 0, t6 = 0;   // diagnostic build: barrier B to the last hand-over / to the end of the speculation
        int cur = 0, seq = 0;
        __syncthreads();                                             // matches role A's prologue barrier
        for (int f = 0; f < nf; ++f) {
            if (fc0 + f < DSS_FEATURES_DELAY) continue;
            for (int i = 0; i < DSS_FRAME_SIZE; ++i) {
                float acc;                                                              // (the chain starts and ends on wave 7)
                ++seq;
                if (seq == 1) __syncthreads();                                          // barrier A (first sample only)
                __syncthreads();                                                        // barrier B
                __builtin_amdgcn_s_setprio(3);               // the relay is the sample's critical path
                const float *an = L.state_a[cur ^ 1];
                if (STAMP) t6 = __builtin_readcyclecounter();
                f32x2 PS[32];
                f32x4 RX[8];
                f32x2 pv_;
                DSS_QR_READ(DSS_QR_G0(1))
                DSS_QR_MUL(16, DSS_QR_G0(1), DSS_R6_1)
                DSS_QR_AWAIT(seq * 8 + 1)
                DSS_QR_ADD(16)
                DSS_QR_PUBLISH(seq * 8 + 2)
                DSS_QR_READ(DSS_QR_G0(3))
                DSS_QR_MUL(16, DSS_QR_G0(3), DSS_R6_3)
                DSS_QR_AWAIT(seq * 8 + 3)
                DSS_QR_ADD(16)
                DSS_QR_PUBLISH(seq * 8 + 4)
                DSS_QR_READ(DSS_QR_G0(5))
                DSS_QR_MUL(16, DSS_QR_G0(5), DSS_R6_5)
                DSS_QR_AWAIT(seq * 8 + 5)
                DSS_QR_ADD(16)
                DSS_QR_PUBLISH(seq * 8 + 6)
                __builtin_amdgcn_s_setprio(0);
                if (STAMP) relay6 += __builtin_readcyclecounter() - t6;
                dss_pkh_speculate(L, lane, u2l_c);           // candidates 0..63, while wave 7 runs the last stage and the gates
                if (STAMP) atc6 += __builtin_readcyclecounter() - t6;
                f32x4 bq[NB / 4];
                dss_await_c(L.state_b, seq, bq);                                            // "barrier C"
                dss_pkh_fc<TRACE>(L, fc, b, bq, 128 + lane, 2, lane, ((size_t)utt * n_frames + f) * DSS_FRAME_SIZE + i);
                __syncthreads();                                                        // barrier D
                cur ^= 1;
            }
        }
        __syncthreads();                                                                // final barrier
        if (STAMP && lane == 0 && b.trace_pcm && gridDim.x == 1) { b.trace_pcm[65] = (float)relay6; b.trace_pcm[66] = (float)atc6; }
    } else {
        // =====================================================================================================
        // role B2 + S (wave 7): GRU B, the even stages of the relay and the gates; scalar recurrences replicated across lanes
        // =====================================================================================================
        DSS_QR_SETUP()
        f32x4 RW[16];                                // [0..7] stage 0 to begin with, [8..15] the second half of stage 2
#pragma unroll
        for (int g = 0; g < 8; ++g) RW[g] = DSS_QR_WLOAD(DSS_R7_6(g));
#pragma unroll
        for (int g = 8; g < 16; ++g) RW[g] = DSS_QR_WLOAD(DSS_R7_4(g));
        const int row = lane < NB3 ? lane : 0;
        __builtin_amdgcn_s_setprio(3);               // everything this wave does is on the sample's critical path
        const float gbb0 = m.gru_b_bias[row];
        const float gbb1 = m.gru_b_bias[NB3 + row];
        // signal history and LPC of the current frame, element j in lane j (j < 16)
        float ls_lane = b.last_sig[(size_t)slot * DSS_LPC_ORDER + (lane & (DSS_LPC_ORDER - 1))], lpc_lane = 0.f;
        float deemph = b.deemph[slot];
        int last_exc = b.last_exc[slot];
        DssKiss99 rng = {b.rng[slot * 4 + 0], b.rng[slot * 4 + 1], b.rng[slot * 4 + 2], b.rng[slot * 4 + 3]};
        unsigned long long stamp_acc[6] = {0, 0, 0, 0, 0, 0};
        unsigned long long t_prev = 0, relay7 = 0;
        int cur = 0, seq = 0;
        float pred = 0.f, upd_pred = 0.f;
        int upd_exc = 0, upd_i = 0;
        bool have_spec = false, next_exists = false, upd_pending = false;
        __syncthreads();                                             // matches role A's prologue barrier
        for (int f = 0; f < nf; ++f) {
            short *pcm_frame = pcm_out + ((size_t)utt * n_frames + f) * DSS_FRAME_SIZE;
            if (fc0 + f < DSS_FEATURES_DELAY) {             // lpcnet.c: frame_count <= FEATURES_DELAY -> silence
                for (int k = lane; k < DSS_FRAME_SIZE / 2; k += 64) reinterpret_cast<int *>(pcm_frame)[k] = 0;
                if (TRACE)
                    for (int k = lane; k < DSS_FRAME_SIZE; k += 64) {
                        b.trace_exc[((size_t)utt * n_frames + f) * DSS_FRAME_SIZE + k] = -1.f;
                        b.trace_pcm[((size_t)utt * n_frames + f) * DSS_FRAME_SIZE + k] = 0.f;
                    }
                continue;
            }
            const float *fo = b.frame_out + ((size_t)utt * n_frames + f) * DSS_COND_STRIDE;
            lpc_lane = fo[3 * NA + NB3 + (lane & (DSS_LPC_ORDER - 1))];
            const float gbc = fo[3 * NA + row];
            for (int i = 0; i < DSS_FRAME_SIZE; ++i) {
                if (STAMP) t_prev = __builtin_readcyclecounter();
                if (!have_spec) {        // first sample of the call: prediction and indices computed directly
                    pred = 0;
#pragma unroll
                    for (int j = 0; j < DSS_LPC_ORDER; ++j)
                        pred -= __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, ls_lane), j)) *
                                __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, lpc_lane), j));
                    const int su = dss_lin2ulaw(__builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, ls_lane), 0)));
                    const int pu = dss_lin2ulaw(pred);
                    if (lane == 0) { L.idx[0] = su; L.idx[1] = pu; L.idx[2] = last_exc; }
                }
                ++seq;
                if (STAMP) { unsigned long long t = __builtin_readcyclecounter(); stamp_acc[0] += t - t_prev; t_prev = t; }
                if (seq == 1) __syncthreads();                                          // barrier A (first sample only)
                if (STAMP) { unsigned long long t = __builtin_readcyclecounter(); stamp_acc[1] += t - t_prev; t_prev = t; }
                if (upd_pending) { DSS_QS_UPDATE() }                                    // previous sample's bookkeeping
                {   // off the critical path: this sample's 8 thresholds
                    const uint32_t r0 = dss_kiss99_rand(rng);
                    const uint32_t r1 = dss_kiss99_rand(rng);
                    if (lane < 8) {
                        const uint32_t r = lane < 4 ? r0 : r1;
                        L.thr[lane] = m.logit_table[(r >> (8 * (lane & 3))) & 0xFF];     // 1 KB table, L2/L1 resident
                    }
                }
                {   // inputs of the speculation the other waves run between barriers B and C
                    const bool last_of_frame = (i == DSS_FRAME_SIZE - 1);
                    next_exists = !(last_of_frame && f == nf - 1);
                    float lp = lpc_lane;                     // lane j < 16 publishes element j
                    const float ls = ls_lane;
                    if (last_of_frame && next_exists && lane < DSS_LPC_ORDER)
                        lp = b.frame_out[((size_t)utt * n_frames + f + 1) * DSS_COND_STRIDE + 3 * NA + NB3 + lane];
                    if (lane < DSS_LPC_ORDER) { L.spec_lpc[lane] = lp; L.spec_ls[lane] = ls; }
                    if (lane == 0) L.spec_pred = pred;
                }
                float rec = gbb1;                                                       // GRU B's recurrent half
#pragma unroll
                for (int j = 0; j < NB; ++j) rec += L.gb_wrec[j * NB3 + row] * L.state_b[j];
                const float sb_old = L.state_b[lane & (NB - 1)];     // the h lanes' own unit: read here, not after the chain
                __syncthreads();                                                        // barrier B
                if (STAMP) { unsigned long long t = __builtin_readcyclecounter(); stamp_acc[2] += t - t_prev; t_prev = t; }
                const float *an = L.state_a[cur ^ 1];
                f32x2 PS[32];
                f32x4 RX[8];
                f32x2 pv_;
                float acc = gbb0 + gbc;                                                 // compute_gruB
                DSS_QR_READ(DSS_QR_G0(0))
                DSS_QR_MUL(8, DSS_QR_G0(0), DSS_R7_0)
                DSS_QR_ADD(8)
                DSS_QR_PUBLISH(seq * 8 + 1)
                DSS_QR_READ(DSS_QR_G0(2))
                DSS_QR_MUL(16, DSS_QR_G0(2), DSS_R7_2)
                DSS_QR_AWAIT(seq * 8 + 2)
                DSS_QR_ADD(16)
                DSS_QR_PUBLISH(seq * 8 + 3)
                DSS_QR_READ(DSS_QR_G0(4))
                DSS_QR_MUL(16, DSS_QR_G0(4), DSS_R7_4)
                DSS_QR_AWAIT(seq * 8 + 4)
                DSS_QR_ADD(16)
                DSS_QR_PUBLISH(seq * 8 + 5)
                DSS_QR_READ(DSS_QR_G0(6))
                DSS_QR_MUL(8, DSS_QR_G0(6), DSS_R7_6)
                DSS_QR_AWAIT(seq * 8 + 6)
                DSS_QR_ADD(8)
                if (STAMP) { asm volatile("" : "+v"(acc)); relay7 += __builtin_readcyclecounter() - t_prev; }
                {   // gates: lanes 0..15 z, 16..31 r, 32..47 h.  r and z travel up to their unit's h lane with gfx950's
                    // row/half swaps (VALU); the new state is formed in the h lanes.  Only the first result of a swap
                    // is used, with distinct operands (see lpcnet_sample.hip).
                    const float zr = dss_sigmoid_approx(L.tansig, acc + rec);
                    const unsigned zb = __builtin_bit_cast(unsigned, zr);
                    const unsigned r_row0 = __builtin_amdgcn_permlane16_swap(zb, 0u, false, false)[1];                  // lanes 0..15 <- 16..31
                    const float r_for_h = __builtin_bit_cast(float, __builtin_amdgcn_permlane32_swap(0u, r_row0, false, false)[0]);  // 32..47 <- 0..15
                    const float z_for_h = __builtin_bit_cast(float, __builtin_amdgcn_permlane32_swap(0u, zb, false, false)[0]);      // 32..47 <- 0..15
                    float hh = acc + rec * r_for_h;
                    hh = dss_tanh_approx(L.tansig, hh);
                    dss_publish_c(L.state_b, lane, z_for_h * sb_old + (1 - z_for_h) * hh, seq);      // "barrier C"
                }
                if (STAMP) { unsigned long long t = __builtin_readcyclecounter(); stamp_acc[3] += t - t_prev; t_prev = t; }
                __syncthreads();                                                        // barrier D
                if (STAMP) { unsigned long long t = __builtin_readcyclecounter(); stamp_acc[4] += t - t_prev; t_prev = t; }
                cur ^= 1;
                int val;
                DSS_TREE_WALK_AT(val, L.bits)
                const int exc = val;
                // the next sample's prediction and mu-law indices were precomputed for every possible exc
                const float pred_next = L.spec_tab_pred[exc];     // (the GRU A waves look the mu-law indices up themselves)
                have_spec = next_exists;
                // Everything below only updates this wave's own state; except at the end of a frame (whose PCM is
                // copied out right after the loop) it is deferred until after the next barrier A.
                upd_exc = exc; upd_pred = pred; upd_i = i; upd_pending = true;
                pred = pred_next;
                if (i == DSS_FRAME_SIZE - 1) { DSS_QS_UPDATE() }
                if (STAMP) { unsigned long long t = __builtin_readcyclecounter(); stamp_acc[5] += t - t_prev; t_prev = t; }
            }
            // wave 7 owns L.pcm: LDS operations of one wave are ordered, no barrier needed
            for (int k = lane; k < DSS_FRAME_SIZE / 2; k += 64)
                reinterpret_cast<int *>(pcm_frame)[k] = reinterpret_cast<const int *>(L.pcm)[k];
        }
        __syncthreads();                                                                // final barrier
        if (STAMP && lane == 0 && b.trace_pcm) {        // diagnostic build only
            for (int k = 0; k < 6; ++k) b.trace_pcm[(size_t)utt * 6 + k] = (float)stamp_acc[k];
            if (gridDim.x == 1) b.trace_pcm[64] = (float)relay7;
        }
        if (lane < NB) b.gru_b_state[(size_t)slot * NB + lane] = L.state_b[lane];
        if (lane < DSS_LPC_ORDER) b.last_sig[(size_t)slot * DSS_LPC_ORDER + lane] = ls_lane;
        if (lane == 0) {
            b.deemph[slot] = deemph;
            b.last_exc[slot] = last_exc;
            b.rng[slot * 4 + 0] = rng.z; b.rng[slot * 4 + 1] = rng.w; b.rng[slot * 4 + 2] = rng.jsr; b.rng[slot * 4 + 3] = rng.jcong;
        }
    }
}

// 1 when model m can run on this kernel: the plain CU-resident layout with at most 10 z/r register slots per gate on
// waves 4 and 5 (they carry dual-FC weights here) and a packed h image that fits beside PkhLds
int dss_pkh_fits(const DssModelDev &m)
{
    return m.fast_ok && !m.ext && m.pkh_ok && m.zr_cap <= 10 && (size_t)m.hblk_pk_floats * sizeof(float) <= DSS_PKH_HBLK_BYTES;
}

// trace: 0, 1 (excitation / pcm trace and teacher forcing), 2 (phase stamps of the diagnostic build: uniform calls only).
// Rows [b.utt0, b.utt0 + n_rows) of the call, one workgroup each.
int dss_launch_sample_network_pkh(const DssModelDev &m, DssBatchDev &b, int n_rows, int n_frames, short *d_pcm, int trace,
                                  hipStream_t s)
{
    if (!dss_pkh_fits(m)) { dss_set_error("packed-h kernel: model layout not supported"); return DSS_EINVAL; }
    const bool ragged = b.slot_of || b.count_of;
    if (ragged && trace == 2) { dss_set_error("phase stamps are taken on uniform calls only"); return DSS_EINVAL; }
    const size_t dyn = ((size_t)m.hblk_pk_floats * sizeof(float) + 15) & ~(size_t)15;
    static std::mutex attr_mu;                  // states on different devices may launch from different threads
    static unsigned long long attr_set = 0;     // per device: the attribute belongs to the device's code object
    int dev = 0;
    DSS_HIP_CHECK(hipGetDevice(&dev));
    {
        std::lock_guard<std::mutex> attr_lk(attr_mu);
        if (!(attr_set >> (dev & 63) & 1)) {
#define DSS_SET_ATTR(K) DSS_HIP_CHECK(hipFuncSetAttribute((const void *)K, hipFuncAttributeMaxDynamicSharedMemorySize, DSS_PKH_HBLK_BYTES))
            DSS_SET_ATTR((lpcnet_sample_pkh_kernel<false, false, 10, false>));
            DSS_SET_ATTR((lpcnet_sample_pkh_kernel<false, false, 10, true>));
            DSS_SET_ATTR((lpcnet_sample_pkh_kernel<true, false, 10, false>));
            DSS_SET_ATTR((lpcnet_sample_pkh_kernel<false, true, 10, false>));
#undef DSS_SET_ATTR
            attr_set |= 1ull << (dev & 63);
        }
    }
    const dim3 grid(n_rows), block(512);
    if (trace == 2) hipLaunchKernelGGL((lpcnet_sample_pkh_kernel<false, true, 10, false>), grid, block, dyn, s, m, b, n_frames, d_pcm);
    else if (trace) hipLaunchKernelGGL((lpcnet_sample_pkh_kernel<true, false, 10, false>), grid, block, dyn, s, m, b, n_frames, d_pcm);
    else if (ragged) hipLaunchKernelGGL((lpcnet_sample_pkh_kernel<false, false, 10, true>), grid, block, dyn, s, m, b, n_frames, d_pcm);
    else hipLaunchKernelGGL((lpcnet_sample_pkh_kernel<false, false, 10, false>), grid, block, dyn, s, m, b, n_frames, d_pcm);
    DSS_HIP_CHECK(hipGetLastError());
    return DSS_OK;
}
