// csrc/lpcnet_sample.hip -- LPCNet sample-rate (16 kHz autoregressive) network, one persistent
// workgroup per utterance, all weights resident on the CU (gfx950).
//
// Restates xiph/LPCNet src/lpcnet.c lpcnet_synthesize_tail_impl() + run_sample_network() and
// src/nnet.c compute_gru_a_input / compute_sparse_gru / compute_gruB / sample_mdense (generic float path
// of src/vec.h), as reached through the reference's binding extensions/lpcnet/cLPCNet.pxd:13.
//
// Where the ~300 KB of weights live for the whole launch (per sample only three embedding rows come from L2,
// as one 12-byte load per lane and table from lane-ordered copies of the tables):
//   GRU A z- and r-gate 8x4 blocks (15k floats)  VGPRs of waves 0..5 (lane = unit; 8 slots per gate on waves 0..3,
//                                                 all DSS_ZRC on waves 4..5, which get the heaviest row groups)
//   GRU A h-gate 8x4 blocks (30k floats)          LDS, one 128-byte record per block, grouped per wave
//   GRU B input weights (18k floats)              inputs 0..95 and 288..351 in VGPRs of waves 6 and 7 (lane = row); the other 224
//                                                 stream from L2 every sample into the registers that will hold their products
//   dual-FC (8k floats)                           VGPRs of waves 0..3 (lane = tree node)
// Roles inside the 512-thread workgroup (8 waves, 2 per SIMD); three workgroup barriers B C D per sample (a fourth,
// A, only on the first sample of a call):
//   waves 0..5  GRU A (dss_role_a, compiled once with and once without the dual-FC).  D..B: tree walk over the
//               decision bits, speculated embedding indices, embedding rows, z and r chains, activations, new state.
//               B..C: the h-gate recurrent chain and the z/r block products of the NEXT sample (they need only the
//               new state) and the speculation over the 256 possible excitations -- hidden under GRU B.
//               C..D (waves 0..3): dual-FC logits of all 255 tree nodes -> decision bits.
//   wave 6      GRU B, segments 1 and 3 of the chain (see GB1..GB4 below): lane = output row, one sequential chain per row,
//               handed between the two waves through LDS as a tagged 8-byte word per row (no barrier).
//   wave 7      GRU B, segments 2 and 4 + gates, and the scalar recurrences: mu-law / de-emphasis / PCM bookkeeping, kiss99
//               thresholds, its own tree walk.
// Summation order inside every row is exactly the C source's (one product at a time, ascending input),
// and the library is built with -ffp-contract=off, so results are bit-identical to the scalar C path.
#include <cstddef>
#include <mutex>

// h-gate slots DSS_HCX.. of a long list: column ids from the table behind the LDS image instead of registers, one chunk of
// two blocks per trip (a dependent LDS round trip for the ids: slower per block than the register-id loop, same sums)
#define DSS_H_TAIL                                                                               \
    if constexpr (EXT) if (nh > HC) {                                                            \
        const unsigned char *hx = reinterpret_cast<const unsigned char *>(hblk_lds + m.ext_tab) + \
                                  (NA / 8) * 2 * 4 + (NA / 8) * 2 * DSS_ZR_TAIL + (tid >> 3) * DSS_HX; \
        unsigned c0 = hx[0], c1 = hx[1];             /* columns are read one trip ahead */      \
        for (int s = HC; s < nh; s += 2) {                                                       \
            const unsigned c0n = hx[s - HC + 2], c1n = hx[s - HC + 3];                           \
            f32x4 HT[4];                                                                         \
            HT[0] = *reinterpret_cast<const f32x4 *>(hw + s * 128);                              \
            HT[1] = *reinterpret_cast<const f32x4 *>(hw + (s + 1) * 128);                        \
            HT[2] = *reinterpret_cast<const f32x4 *>(xbase + c0 * 16);                           \
            HT[3] = *reinterpret_cast<const f32x4 *>(xbase + c1 * 16);                           \
            DSS_H_MAC(HT)                                                                        \
            c0 = c0n; c1 = c1n;                                                                  \
        }                                                                                        \
    }
#include "lpcnet_sample_common.h"
#undef HC
// diagnostic build: the low 32 bits of the shader clock (differences of stamps; 64-bit accumulators cost the stamped build registers)
#define DSS_NOW() ((unsigned)__builtin_readcyclecounter())                                  // a constant of the role here: DSS_HC, or DSS_HCX with the extended paths

// GRU B: one dependent chain of 384 sums per row.  Four segments that alternate between the two relay waves; every segment
// but the first is summed from products formed while the other wave was summing (the chain itself is then one v_add_f32
// per input instead of 1.8 instructions).  Only segment 1's weights (and 64 of segment 4's) live in registers: the others
// come from L2 (m.gb_w_quad, 73 KB, shared by every workgroup) INTO the registers that will hold their products, issued where
// the wave idles anyway, a sample ahead (DSS_GBG_LOADS), and are multiplied in place once barrier B has released the state:
//   segment 1  inputs [0, GB1)            wave 6, multiplied as it goes (nothing can be formed before barrier B): ~10.5 cycles/input
//   segment 2  [GB1, GB1+GB2)             wave 7; weights loaded behind barrier C of the sample before, products while wave 6 runs segment 1
//   segment 3  [GB1+GB2, GB1+GB2+GB3)     wave 6; weights loaded behind barrier C, products while wave 7 sums segment 2
//   segment 4  the last GB4 inputs        wave 7; products while wave 6 sums segment 3: the first GB4R with weights in VGPRs, the
//                                         last GB4H with weights loaded (into the registers segment 2 has left) when segment 2 is summed
// Sums run at 4.4 cycles per input, in-place products at ~5.5, a hand-over takes 130-150 cycles.
// Measured on the way (profiles/r4_latency_kernel_experiment.md): rounds 1-3 had two segments (208 + 176 inputs, 48 of them
// pre-multiplied): 4.06 k cycles from barrier B to C.  The first four-segment form kept 192 + 64 weights in registers and 32 in
// LDS (multiplied by a GRU A wave): its segment 1 had to be 144 inputs long to cover the L2 loads of segment 2, issued after
// barrier B.  "Barrier C" as an LDS word polled by the dual-FC waves was slower than the barrier (the polls take issue slots
// and LDS cycles from the relay waves on the same SIMDs).  One state read ahead is not enough for the in-place products: the
// LDS answers in 50+ cycles while the six GRU A waves run their h chains, so DSS_GBG_MUL keeps several pairs of reads in flight.
// (round 5: 80 / 104 / 104 / 96 -- a shorter on-the-fly segment now that the GRU A waves' address arithmetic is lighter; same-box
//  A/B against 96 / 96 / 96 / 96: 38.3 vs 38.5 ms.  80/96/112 the same, 64/96/128, 80/96/128 and 96/96/112 slower: 38.9 / 41.4 / 40.9)
#ifndef GB1
#define GB1 80
#endif
#ifndef GB2
#define GB2 104
#endif
#ifndef GB3
#define GB3 104
#endif
#define GB4 (NA - GB1 - GB2 - GB3)
#ifndef DSS_SPEC_WAVE
#define DSS_SPEC_WAVE 4           // which of waves 4, 5 takes candidates 64..127 of the speculation
#endif
#ifndef DSS_D3
#define DSS_D3 3                  // pairs of state reads in flight (+ 1) while segment 3 / segment 2 is multiplied in place
#endif                            //   (deeper was slower: 4 / 3 39.8 ms, 6 / 3 39.8, 3 / 2 39.6, 2 / 2 40.0 at 80/96/112/96 inputs)
#ifndef DSS_D2
#define DSS_D2 2
#endif
#ifndef DSS_RELAY_STAMP
#define DSS_RELAY_STAMP 0         // development builds (-DDSS_RELAY_STAMP=1, tools/relay_stamps.py): the relay's way points from the TIMED instantiation
#endif
#ifndef DSS_RELAY_MASK
#define DSS_RELAY_MASK 0
#endif
#ifndef DSS_KNOCKOUT
#define DSS_KNOCKOUT 0
#endif
#ifndef GB4H
#define GB4H 32
#endif
#define GB4R (GB4 - GB4H)
static_assert(GB4 <= GB2, "segment 4's products reuse segment 2's registers");
static_assert(GB1 % 16 == 0 && GB2 % 8 == 0 && GB3 % 8 == 0 && GB4R % 8 == 0 && GB4H % 8 == 0 && GB4H >= 0 && GB4R > 0, "segment sizes");
struct SampleLds {
    float state_a[2][NA + 4];             // GRU A state; "column 96" is four zeros: the input of the h-gate slots a row group
                                          //   does not use (see DSS_H_CHAIN).  The state is WRITTEN between barriers D and B and
                                          //   READ between B and C (h chains, z/r products, GRU B), so one buffer is enough -- and
                                          //   makes every block column's LDS address `column * 16 + constant`: the constant goes
                                          //   into the instruction's offset field and a column costs one VALU instruction (an SDWA
                                          //   shift of its byte) instead of 2.25 (round 5; DSS_NEW / DSS_OLD below).  Only the
                                          //   extended paths read the OLD state while the new one is written (z/r tail blocks,
                                          //   between D and B): that instantiation keeps both buffers.
    float gb_wrec[NB * NB3];              // GRU B recurrent weights [16][48]
    float tansig[208];
    float ulaw2lin[256];
    float spec_tab_pred[256];             // speculation over all 256 excitation values (see role A, B..C):
    unsigned short spec_tab_idx[256];     //   next sample's prediction and its two mu-law indices (su | pu << 8)
    float spec_prod[DSS_LPC_ORDER];       // inputs of the speculation, published by wave 7: [0] = lpc[0] of the next sample's frame,
                                          //   [j] = history[j-1] * lpc[j] (these products are the same for every candidate),
    float spec_pred;                      //   and this sample's prediction
    float pad1[3];
    float gb_acc[64][2];                  // GRU B running sums handed between the relay waves: (sum, 4 * sample number + segments done)
    float ah[NA];                         // h-gate pre-activation: written by a unit's h lane, read by its z/r lane
    float state_b[NB];
    float thr[8];
    unsigned bits[8];                     // decision bit of every tree node (256 bits)
    int idx[4];                           // last_sig_ulaw, pred_ulaw, last_exc
    short pcm[DSS_FRAME_SIZE];
};

// Products of a relay segment formed ahead of its sums.  Resident weights: N inputs (multiple of 16) starting at AN, weights
// WB[WOFF..] (pairs), into PQ[0 .. N/4); the state values are read one group of 16 inputs ahead.
#define DSS_GB_PREMUL(AN, N, WOFF, POFF)                                                               \
    {                                                                                            \
        f32x4 xq[2][2];                          /* eight inputs per trip: a larger buffer spilled wave 7's weights */ \
        xq[0][0] = *reinterpret_cast<const f32x4 *>((AN));                                       \
        xq[0][1] = *reinterpret_cast<const f32x4 *>((AN) + 4);                                   \
        _Pragma("unroll") for (int g = 0; g < (N) / 8; ++g) {                                    \
            if (g + 1 < (N) / 8) {                                                               \
                xq[(g + 1) & 1][0] = *reinterpret_cast<const f32x4 *>((AN) + 8 * (g + 1));       \
                xq[(g + 1) & 1][1] = *reinterpret_cast<const f32x4 *>((AN) + 8 * (g + 1) + 4);   \
            }                                                                                    \
            __builtin_amdgcn_sched_barrier(0);                                                   \
            if (g + 1 < (N) / 8) DSS_WAIT_LGKM(2); else DSS_WAIT_LGKM(0);                        \
            __builtin_amdgcn_sched_barrier(0);                                                   \
            _Pragma("unroll") for (int u = 0; u < 2; ++u) {                                      \
                PQ[(POFF) + 2 * g + u].lo = WB[(WOFF) + 2 * (2 * g + u)] * xq[g & 1][u].lo;      \
                PQ[(POFF) + 2 * g + u].hi = WB[(WOFF) + 2 * (2 * g + u) + 1] * xq[g & 1][u].hi;  \
            }                                                                                    \
            /* pinned: left alone, the compiler sinks the multiplications below the hand-over wait */ \
            asm volatile("" : "+v"(PQ[(POFF) + 2 * g]), "+v"(PQ[(POFF) + 2 * g + 1]));           \
            __builtin_amdgcn_sched_barrier(0);                                                   \
        }                                                                                        \
    }
// ... with the weights streamed from L2 (segments 2 and 3; m.gb_w_quad: [block of four inputs][lane][4]).  Two halves:
// DSS_GBG_LOADS issues all N/4 loads INTO the product registers (scalar base + lane offset + immediate per load) -- one sample
// AHEAD, where the wave idles anyway (wave 7: behind barrier C, while the dual-FC waves work; wave 6: once it has handed
// segment 3 over), so that neither their issue time (~20 cycles each) nor the L2 latency stands in the relay;
// DSS_GBG_MUL multiplies every quad in place by its four state values once barrier B has released them.
#define DSS_GBG_LOADS(N, BLK0) DSS_GBG_LOADS_AT(N, BLK0, 0, gvo)
#define DSS_GBG_LOADS_AT(N, BLK0, POFF, GVO)                                                     \
    {                                                                                            \
        _Pragma("unroll") for (int k = 0; k < ((N) / 4 + 7) / 8; ++k) asm volatile("" : "+v"(GVO[k]));   /* (not loop-invariant) */ \
        _Pragma("unroll") for (int g = 0; g < (N) / 4; ++g)                                      \
            PQ[(POFF) + g] = *reinterpret_cast<const f32x4 *>(gq + (size_t)GVO[g >> 3] + (ptrdiff_t)(((BLK0) + g) * 1024 - ((BLK0) + (g & ~7) + 4) * 1024)); \
    }
#define DSS_GBG_MUL(AN, N, D) DSS_GBG_MUL_AT(AN, N, D, 0)
#define DSS_GBG_MUL_AT(AN, N, D, POFF)                                                           \
    {                                                                                            \
        /* two quads per trip (one wait, one hazard slot per eight inputs), D - 1 pairs of state reads in flight: with a   \
           single read ahead every quad waited a full LDS round trip, and the LDS is busy with the h chains here */        \
        f32x4 xq[D][2];                                                                          \
        _Pragma("unroll") for (int u = 0; u < (D) - 1; ++u) {                                    \
            xq[u][0] = *reinterpret_cast<const f32x4 *>((AN) + 8 * u);                           \
            xq[u][1] = *reinterpret_cast<const f32x4 *>((AN) + 8 * u + 4);                       \
        }                                                                                        \
        _Pragma("unroll") for (int g = 0; g < (N) / 8; ++g) {                                    \
            if (g + (D) - 1 < (N) / 8) {                                                         \
                xq[(g + (D) - 1) % (D)][0] = *reinterpret_cast<const f32x4 *>((AN) + 8 * (g + (D) - 1));     \
                xq[(g + (D) - 1) % (D)][1] = *reinterpret_cast<const f32x4 *>((AN) + 8 * (g + (D) - 1) + 4); \
            }                                                                                    \
            __builtin_amdgcn_sched_barrier(0);                                                   \
            PQ[(POFF) + 2 * g].lo = PQ[(POFF) + 2 * g].lo * xq[g % (D)][0].lo;                   \
            PQ[(POFF) + 2 * g].hi = PQ[(POFF) + 2 * g].hi * xq[g % (D)][0].hi;                   \
            PQ[(POFF) + 2 * g + 1].lo = PQ[(POFF) + 2 * g + 1].lo * xq[g % (D)][1].lo;           \
            PQ[(POFF) + 2 * g + 1].hi = PQ[(POFF) + 2 * g + 1].hi * xq[g % (D)][1].hi;           \
            asm volatile("" : "+v"(PQ[(POFF) + 2 * g]), "+v"(PQ[(POFF) + 2 * g + 1]));           \
            __builtin_amdgcn_sched_barrier(0);                                                   \
        }                                                                                        \
    }
// the sums of a pre-multiplied segment: one dependent chain, input order
#define DSS_GB_SUMS(N)                                                                           \
    {                                                                                            \
        _Pragma("unroll") for (int q = 0; q < (N) / 4; ++q) {                                    \
            acc += PQ[q].x;                                                                      \
            acc += PQ[q].y;                                                                      \
            acc += PQ[q].z;                                                                      \
            acc += PQ[q].w;                                                                      \
        }                                                                                        \
        __builtin_amdgcn_sched_barrier(0);                                                       \
    }
// Hand the running sums to the other relay wave / take them over.  One 8-byte LDS word per lane carries the sum and its tag
// (4 * sample number + segments done): the taker polls its own lane's word, and the read that sees every lane's tag has the
// sums in it -- one LDS round trip per hand-over instead of flag, wait, sums.  Hand-written ds_ instructions: the word must be
// written and read as ONE 8-byte access, and __builtin_bit_cast of a vector ELEMENT reads element 0 with this clang.
#define DSS_GB_PUBLISH(V)                                                                        \
    {                                                                                            \
        const int tagi_ = (int)(V);                                                              \
        const f32x2 pw_ = {acc, __builtin_bit_cast(float, tagi_)};                               \
        asm volatile("ds_write_b64 %0, %1" :: "v"(gb_addr), "v"(pw_) : "memory");                \
    }
#define DSS_GB_AWAIT(V)                                                                          \
    {                                                                                            \
        f32x2 pv_;                                                                               \
        unsigned tag_;                                                                           \
        do {                                                                                     \
            asm volatile("ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(pv_) : "v"(gb_addr) : "memory"); \
            const float tf_ = pv_.y;                                                             \
            tag_ = __builtin_bit_cast(unsigned, tf_);                                            \
        } while (__any(tag_ != (unsigned)(V)));                                                  \
        acc = pv_.x;                                                                             \
    }

// a way point of a relay wave (diagnostic builds): V is pinned on both sides of the clock read, so that what produces V is
// issued before the stamp and what consumes it after (the clock read is scalar and would otherwise float among the sums)
#define DSS_RSTAMP(V, SLOT, T0)                                                                  \
    if (RS) {                                                                                    \
        asm volatile("" : "+v"(V));                                                              \
        SLOT += DSS_NOW() - (T0);                                                                \
        asm volatile("" : "+v"(V));                                                              \
    }

#define DSS_TREE_WALK(VAL) DSS_TREE_WALK_AT(VAL, L.bits)

// one candidate excitation value of the speculation, for wave 6 (inputs published by wave 7 right after its tree walk)
#define DSS_SPECULATE(CAND)                                                                      \
    {                                                                                            \
        const int cand_ = (CAND);                                                                \
        const float pcm_c = L.spec_pred + L.ulaw2lin[cand_];                                     \
        float pc = 0;                                                                            \
        pc -= pcm_c * L.spec_prod[0];                                                            \
        _Pragma("unroll") for (int j = 1; j < DSS_LPC_ORDER; ++j) pc -= L.spec_prod[j];          \
        const int su_c = dss_lin2ulaw(pcm_c), pu_c = dss_lin2ulaw(pc);                           \
        L.spec_tab_pred[cand_] = pc;                                                             \
        L.spec_tab_idx[cand_] = (unsigned short)(su_c | (pu_c << 8));                            \
    }

// wave 7: fold the sampled excitation into the signal history and emit the PCM sample (lpcnet_synthesize_tail_impl)
#define DSS_S_UPDATE()                                                                           \
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
        deemph = dss_uniform(pcm);                                                               \
        if (pcm < -32767) pcm = -32767;                                                          \
        if (pcm > 32767) pcm = 32767;                                                            \
        if (lane == 0) L.pcm[upd_i] = (short)(int)floor(.5 + (double)pcm);                       \
        upd_pending = false;                                                                     \
    }

// =====================================================================================================
// role A: GRU A (+ dual-FC on waves 0..3, + the speculation on waves 0, 1 and DSS_SPEC_WAVE).  Two instantiations share this text:
//   waves 0..3  HAS_FC, at most 8 z/r register slots per gate (the dual-FC weights take 32 registers);
//   waves 4..5  no dual-FC, all Z slots -- the host gives them the row groups with the most z/r blocks.
// Keeping the two apart is what keeps either under the 256-VGPR budget without spill reloads in the sample loop.
// =====================================================================================================
// which of L.state_a's buffers holds the state being written / read for sample parity `cur` (see SampleLds::state_a)
#define DSS_NEW(cur) (EXT ? ((cur) ^ 1) : 0)
#define DSS_OLD(cur) (EXT ? (cur) : 0)

template <bool TRACE, bool STAMP, int Z, bool HAS_FC, bool EXT>
__device__ __forceinline__ void dss_role_a(SampleLds &L, float *hblk_lds, const DssModelDev &m, const DssBatchDev &b,
                                           int n_frames, int utt, int slot, int nf, int fc0, int tid, int wave, int lane,
                                           short *pcm_out_dbg)
{
    constexpr int HC = EXT ? DSS_HCX : DSS_HC;                   // h slots with register-held column ids
    const int unit = m.unit_of[tid];                             // z/r chains + gates of this unit
    const int uh = m.unit_h[tid];                                // h-gate chain of this (other) unit
    const int nh = __builtin_amdgcn_readfirstlane(m.wave_nh[wave]);
    const int nzr = __builtin_amdgcn_readfirstlane(m.wave_nzr[wave]);
    const int nzt = EXT ? __builtin_amdgcn_readfirstlane(m.wave_nzt[wave]) : 0;   // z/r blocks beyond the register slots (LDS records)
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
    // dual-FC constants of tree node `tid` (waves 0..3)
    // the two dense layers of the node run as the two halves of packed fp32 instructions: weights as pairs
    // (layer 0 input j, layer 1 input j), the two running sums as one register pair
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
    const float u2l_c = L.ulaw2lin[(HAS_FC ? 128 + tid : tid - 64 * DSS_SPEC_WAVE + 64) & 255];   // this lane's excitation candidate (waves 0, 1, DSS_SPEC_WAVE)
    const int level = 31 - __clz(tid | 1);                       // FC node = (1 << level) | prefix
    const bool recur_first = m.h.gru_a_order == DSS_GRUA_RECUR_FIRST;     // wave-uniform (kernel argument)
    int cur = 0, seq = 0;
    float st = L.state_a[0][unit];
    unsigned sa[8] = {0, 0, 0, 0, 0, 0, 0, 0}, ta = 0;   // diagnostic build only
    f32x4 PR[2 * ZRC];                                           // z/r block products of the coming sample
    bool first_sample = true;
    DSS_H_CHAIN(L.state_a[0])                                    // first sample of this call
    DSS_ZR_PRODUCTS(L.state_a[0])
    __syncthreads();                                             // L.ah of every unit visible to its z/r lane

    for (int f = 0; f < nf; ++f) {
        if (fc0 + f < DSS_FEATURES_DELAY) continue;              // silent frame: decoder state untouched
        const float *fo = b.frame_out + ((size_t)utt * n_frames + f) * DSS_COND_STRIDE;     // wave-uniform base
        const float cz = fo[(unsigned)unit], cr = fo[(unsigned)(NA + unit)], ch = fo[(unsigned)(2 * NA + unit)];
        for (int i = 0; i < DSS_FRAME_SIZE; ++i) {
            // keep the packed column ids opaque so the per-slot unpacking is not hoisted out of the sample
            // loop into (spilled) registers
#pragma unroll
            for (int k = 0; k < (2 * ZRL + 3) / 4; ++k) asm volatile("" : "+v"(PZ[k]));
#pragma unroll
            for (int k = 0; k < HC / 4; ++k) asm volatile("" : "+v"(PH[k]));
            float az = rbz + dgz * st;                           // compute_sparse_gru, before the input term
            float ar = rbr + dgr * st;
            const float ahv = L.ah[unit];                        // this unit's h-gate pre-activation (its h lane, B..C)
            // The three embedding indices.  First sample of a call: wave 7 computes them and hands them over through
            // L.idx and barrier A.  Every later sample: this wave walks the sampling tree itself (the same scalar
            // code wave 7 runs for its bookkeeping) and looks the speculated indices up, so neither a barrier nor
            // wave 7 stands between the dual-FC and the embedding loads.
            int si, pi, ei;
            if (first_sample) {
                __syncthreads();                                                    // barrier A (first sample only)
                si = L.idx[0]; pi = L.idx[1]; ei = L.idx[2];
                first_sample = false;
            } else {
                int exc_;
                DSS_TREE_WALK(exc_)
                const unsigned sidx = __builtin_amdgcn_readfirstlane((unsigned)L.spec_tab_idx[exc_]);
                si = (int)(sidx & 0xFF); pi = (int)(sidx >> 8); ei = exc_;
            }
            // wave-uniform by construction: keep them scalar, so that the row offsets are SALU work and the loads take an
            // SGPR base (the first-sample path would otherwise drag them into VGPRs)
            si = __builtin_amdgcn_readfirstlane(si); pi = __builtin_amdgcn_readfirstlane(pi); ei = __builtin_amdgcn_readfirstlane(ei);
            if (STAMP) ta = DSS_NOW();
            {
                // the nine embedding values of this lane: three 12-byte loads from the lane-ordered copies of the tables
                // (m.embed_lane: [index][lane][gate]), 768 contiguous bytes per wave and table
                typedef float f32x3 __attribute__((ext_vector_type(3)));
                const f32x3 es = *reinterpret_cast<const f32x3 *>(m.embed_lane[0] + ((unsigned)si * NA + (unsigned)tid) * 3);
                const f32x3 ep = *reinterpret_cast<const f32x3 *>(m.embed_lane[1] + ((unsigned)pi * NA + (unsigned)tid) * 3);
                const f32x3 ee = *reinterpret_cast<const f32x3 *>(m.embed_lane[2] + ((unsigned)ei * NA + (unsigned)tid) * 3);
                const float es0 = es.x, es1 = es.y, es2 = es.z, ep0 = ep.x, ep1 = ep.y, ep2 = ep.z, ee0 = ee.x, ee1 = ee.y, ee2 = ee.z;
                if (STAMP) { const unsigned t = DSS_NOW(); sa[0] += t - ta; ta = t; }
                const float gz = ((cz + es0) + ep0) + ee0;                          // compute_gru_a_input
                const float gr = ((cr + es1) + ep1) + ee1;
                const float gh = ((ch + es2) + ep2) + ee2;
                // nnet.c 2021 (default): (bias + diag*state) + input, then the blocks in idx order;
                // nnet.c 2019-20 (blob flag): the blocks first, the input last
                if (!recur_first) { az = az + gz; ar = ar + gr; }
                if (STAMP) { asm volatile("" :: "v"(az), "v"(ar)); const unsigned t = DSS_NOW(); sa[1] += t - ta; ta = t; }
                // the block products were formed right after the previous sample's state update (under GRU B);
                // what is left on the critical path are the dependent sums, z and r chains interleaved
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
                if constexpr (EXT) if (nzt) {
                    // models with skewed sparsity: the row group's blocks beyond its register slots, still in idx order --
                    // weights from the tail records behind the h-gate image, columns from the table, state of this sample
                    const int *toff = reinterpret_cast<const int *>(hblk_lds + m.ext_tab) + (tid >> 3) * 2;
                    const unsigned char *tc = reinterpret_cast<const unsigned char *>(hblk_lds + m.ext_tab) + (NA / 8) * 2 * 4 +
                                              (tid >> 3) * 2 * DSS_ZR_TAIL;
                    const char *tz = reinterpret_cast<const char *>(hblk_lds + toff[0]) + (lane & 7) * 16;
                    const char *tr = reinterpret_cast<const char *>(hblk_lds + toff[1]) + (lane & 7) * 16;
                    const char *xb = reinterpret_cast<const char *>(L.state_a[DSS_OLD(cur)]);
                    unsigned cz = tc[0], cr = tc[DSS_ZR_TAIL];        // the columns of a slot are read one trip ahead
                    for (int s = 0; s < nzt; ++s) {
                        const unsigned czn = tc[s + 1], crn = tc[DSS_ZR_TAIL + s + 1];   // (one byte past a row: unused)
                        const f32x4 wz = *reinterpret_cast<const f32x4 *>(tz + s * 128);
                        const f32x4 wr = *reinterpret_cast<const f32x4 *>(tr + s * 128);
                        const f32x4 xz = *reinterpret_cast<const f32x4 *>(xb + cz * 16);
                        const f32x4 xr = *reinterpret_cast<const f32x4 *>(xb + cr * 16);
                        const f32x2 pz0 = wz.lo * xz.lo, pz1 = wz.hi * xz.hi, pr0 = wr.lo * xr.lo, pr1 = wr.hi * xr.hi;
                        az += pz0.x; ar += pr0.x; az += pz0.y; ar += pr0.y;
                        az += pz1.x; ar += pr1.x; az += pz1.y; ar += pr1.y;
                        cz = czn; cr = crn;
                    }
                }
                if (recur_first) { az = gz + az; ar = gr + ar; }
                if (STAMP) { asm volatile("" :: "v"(az), "v"(ar)); const unsigned t = DSS_NOW(); sa[2] += t - ta; ta = t; }
                float z, r;
                dss_sigmoid_approx2(L.tansig, az, ar, z, r);
                float h = ahv * r + gh;
                h = dss_tanh_approx(L.tansig, h);
                st = z * st + (1 - z) * h;
                L.state_a[DSS_NEW(cur)][unit] = st;
                if (STAMP) { asm volatile("" :: "v"(st)); const unsigned t = DSS_NOW(); sa[3] += t - ta; ta = t; }
            }
            __syncthreads();                                                        // barrier B
            if (STAMP) { const unsigned t = DSS_NOW(); sa[4] += t - ta; ta = t; }
            else if (DSS_RELAY_STAMP) ta = DSS_NOW();
#if DSS_KNOCKOUT      /* development builds only (results WRONG on purpose): what the relay's SIMD neighbours cost it, profiles/r5_latency_kernel_experiment.md */
            if (!((DSS_KNOCKOUT & 1) && (wave == 2 || wave == 3)) && !((DSS_KNOCKOUT & 4) && (wave == 0 || wave == 1))) { DSS_H_CHAIN(L.state_a[DSS_NEW(cur)]) }
            if (!((DSS_KNOCKOUT & 2) && (wave == 2 || wave == 3))) { DSS_ZR_PRODUCTS(L.state_a[DSS_NEW(cur)]) }
#else
            DSS_H_CHAIN(L.state_a[DSS_NEW(cur)])                      // next sample's h chain, under GRU B
            DSS_ZR_PRODUCTS(L.state_a[DSS_NEW(cur)])                  // ... and its z/r block products (sums come later)
#endif
            if (wave == DSS_SPEC_WAVE || wave < 2) {
                // Speculation over all 256 possible excitation values of THIS sample, one candidate per lane of wave DSS_SPEC_WAVE
                // (candidates 64..127), of waves 0, 1, which have the lightest B..C load of the dual-FC waves (128..255),
                // and of wave 6 once it has handed its half of the GRU B chain over (0..63): the next sample's LPC prediction and mu-law indices, so that once the tree walk has
                // picked the value nobody has to run the two ~40-step dependent chains.  Same expressions, same order
                // as lpcnet_synthesize_tail_impl().
                const int cand = HAS_FC ? 128 + tid : tid - 64 * DSS_SPEC_WAVE + 64;
                const float pcm_c = L.spec_pred + u2l_c;
                float pc = 0;
                pc -= pcm_c * L.spec_prod[0];
#pragma unroll
                for (int j = 1; j < DSS_LPC_ORDER; ++j) pc -= L.spec_prod[j];       // (history[j-1] * lpc[j], formed once by wave 7)
                const int su_c = dss_lin2ulaw(pcm_c), pu_c = dss_lin2ulaw(pc);
                L.spec_tab_pred[cand] = pc;
                L.spec_tab_idx[cand] = (unsigned short)(su_c | (pu_c << 8));
            }
            if (STAMP) { const unsigned t = DSS_NOW(); sa[5] += t - ta; ta = t; }
            else if (DSS_RELAY_STAMP) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); sa[5] += DSS_NOW() - ta; }   // B .. ready for C
            __syncthreads();                                                        // barrier C
            if (STAMP) { const unsigned t = DSS_NOW(); sa[6] += t - ta; ta = t; }
            if constexpr (HAS_FC) {                                                 // sample_mdense, all nodes
                const float thr_lv = L.thr[level];                                  // issued first, used last
                f32x2 s12 = {fb0, fb1};
#pragma unroll
                for (int j4 = 0; j4 < NB / 4; ++j4) {
                    const f32x4 bj = *reinterpret_cast<const f32x4 *>(L.state_b + 4 * j4);
                    // per input: one packed product for both layers, one packed sum (each half rounds on its own,
                    // exactly as the two scalar chains did)
                    // (the four products first: a packed result needs a wait state before it can be read)
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
                dss_tanh_approx2(L.tansig, s1, s2, t1, t2);
                s1 = ff0 * t1;
                s2 = ff1 * t2;
                s1 += s2;
                bool bit = thr_lv < s1;
                if constexpr (TRACE) {               // teacher forcing (tests): record every logit, bend the walk
                    const size_t o = ((size_t)utt * n_frames + f) * DSS_FRAME_SIZE + i;
                    if (b.trace_logits) b.trace_logits[o * 256 + tid] = tid ? s1 : 0.f;
                    if (b.force_exc) {
                        const int v = b.force_exc[o];                                   // bits b7..b0, b7 decided at level 0
                        if ((tid ^ (1 << level)) == (v >> (8 - level))) bit = (v >> (7 - level)) & 1;
                    }
                }
                const unsigned long long mask = __ballot(bit);
                if (lane == 0) { L.bits[2 * wave] = (unsigned)mask; L.bits[2 * wave + 1] = (unsigned)(mask >> 32); }
            }
            __syncthreads();                                                        // barrier D
            if (STAMP) { const unsigned t = DSS_NOW(); sa[7] += t - ta; ta = t; }
            cur ^= 1;
        }
    }
    __syncthreads();                                                                // final barrier
    if (STAMP && lane == 0 && b.trace_exc)
        for (int k = 0; k < 8; ++k) b.trace_exc[((size_t)utt * 6 + wave) * 8 + k] = (float)sa[k];
    if (DSS_RELAY_STAMP && !STAMP && lane == 0 && blockIdx.x == 0)
        reinterpret_cast<unsigned *>(pcm_out_dbg + (size_t)utt * n_frames * DSS_FRAME_SIZE)[16 + wave] = sa[5];
    b.gru_a_state[(size_t)slot * NA + unit] = st;
}

// RAGGED: rows name their decoder slot and frame count (b.slot_of / b.count_of).  A separate instantiation, so the
// uniform form keeps its register allocation (the two extra live scalars cost 1.3 % there); the trace build always
// honours the lists.
template <bool TRACE, bool STAMP, int Z, bool RAGGED, bool EXT>
__global__ void __launch_bounds__(512)
lpcnet_sample_kernel(DssModelDev m, DssBatchDev b, int n_frames, short *__restrict__ pcm_out)
{
    __shared__ __attribute__((aligned(16))) SampleLds L;
    extern __shared__ __attribute__((aligned(16))) float hblk_lds[];       // h-gate block records (size per model)
    static_assert(sizeof(SampleLds) % 16 == 0, "dynamic LDS must start 16-byte aligned");
    static_assert(sizeof(SampleLds) + DSS_HBLK_BYTES <= 160 * 1024, "LDS budget");
    constexpr bool RG = RAGGED || TRACE;
    constexpr bool RS = STAMP || DSS_RELAY_STAMP;          // way points of the relay waves
    // row of this call (scratch, features, PCM): ragged calls with counts start their longest rows first (b.row_of)
    const int utt = (RG && b.row_of) ? __builtin_amdgcn_readfirstlane(b.row_of[blockIdx.x]) : b.utt0 + (int)blockIdx.x;
    const int slot = (RG && b.slot_of) ? b.slot_of[utt] : utt;                     // decoder state it continues
    const int nf = (RG && b.count_of) ? min(b.count_of[utt], n_frames) : n_frames; // its own frame count
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;

    // ---------------- one-time staging into LDS -------------------------------------------------------
    for (int k = tid * 4; k < m.hblk_floats; k += 512 * 4)
        *reinterpret_cast<f32x4 *>(&hblk_lds[k]) = *reinterpret_cast<const f32x4 *>(&m.hblk[k]);
    for (int k = tid; k < NB * NB3; k += 512) L.gb_wrec[k] = m.gru_b_w_rec[k];
    if (tid < 201) L.tansig[tid] = m.tansig[tid];
    if (tid < 256) L.ulaw2lin[tid] = m.ulaw2lin[tid];
    if (tid < NA) L.state_a[0][tid] = b.gru_a_state[(size_t)slot * NA + tid];
    if (tid < 8) L.state_a[tid >> 2][NA + (tid & 3)] = 0.f;
    if (tid < NB) L.state_b[tid] = b.gru_b_state[(size_t)slot * NB + tid];
    if (tid < 128) L.gb_acc[tid >> 1][tid & 1] = 0.f;
    const int fc0 = b.fc0[utt];
    __syncthreads();

    if (wave < 4) {
        dss_role_a<TRACE, STAMP, (Z < 8 ? Z : 8), true, EXT>(L, hblk_lds, m, b, n_frames, utt, slot, nf, fc0, tid, wave, lane, pcm_out);
    } else if (wave < 6) {
        dss_role_a<TRACE, STAMP, Z, false, EXT>(L, hblk_lds, m, b, n_frames, utt, slot, nf, fc0, tid, wave, lane, pcm_out);
    } else if (wave == 6) {
        // =====================================================================================================
        // role B1: GRU B, segments 1 and 3; lane = row (0..15 z, 16..31 r, 32..47 h)
        // =====================================================================================================
        f32x2 WB[GB1 / 2];                           // segment 1's weights; segment 3's come from L2 every sample
#pragma unroll
        for (int j = 0; j < GB1 / 2; ++j) {
            WB[j].x = m.gb_w_lane[(size_t)(2 * j) * 64 + lane];
            WB[j].y = m.gb_w_lane[(size_t)(2 * j + 1) * 64 + lane];
        }
        const char *gq = reinterpret_cast<const char *>(m.gb_w_quad);
        unsigned gvo[(GB3 / 4 + 7) / 8];             // lane offsets of segment 3's eight-block windows in m.gb_w_quad
#pragma unroll
        for (int k = 0; k < (GB3 / 4 + 7) / 8; ++k) gvo[k] = (unsigned)lane * 16 + (unsigned)((GB1 + GB2) / 4 + 8 * k + 4) * 1024u;
        f32x4 PQ[GB3 / 4];                           // segment 3: weights (loaded a sample ahead), then products
        const unsigned gb_addr = dss_lds_addr(&L.gb_acc[lane][0]);
        const int row = lane < NB3 ? lane : 0;
        __builtin_amdgcn_s_setprio(3);               // everything this wave does is on the sample's critical path
        const float gbb0 = m.gru_b_bias[row];
        int cur = 0, seq = 0;
        unsigned r6[5] = {0, 0, 0, 0, 0};     // diagnostic build: cycles from barrier B to the relay's way points on this wave
        __syncthreads();                                             // matches role A's prologue barrier
        for (int f = 0; f < nf; ++f) {
            if (fc0 + f < DSS_FEATURES_DELAY) continue;     // (peeling the silent frames off here as on wave 7: 2 % slower)
            const float gbc = b.frame_out[((size_t)utt * n_frames + f) * DSS_COND_STRIDE + 3 * NA + row];
            for (int i = 0; i < DSS_FRAME_SIZE; ++i) {
                float acc = gbb0 + gbc;                                                 // compute_gruB
                ++seq;
                if (seq == 1) {
                    DSS_GBG_LOADS(GB3, (GB1 + GB2) / 4)                                 // (every later sample: loaded at the end of the one before)
                    __syncthreads();                                                    // barrier A (first sample only)
                }
                __syncthreads();                                                        // barrier B
                const float *an = L.state_a[DSS_NEW(cur)];
                unsigned t6 = 0;
                if (RS) t6 = DSS_NOW();
#if DSS_RELAY_MASK
                if (lane < NB3) {        // 48 rows: the other 16 lanes only cost LDS return cycles (every state read is 16 B per ACTIVE lane)
#endif
                DSS_GB_CHAIN(an, GB1)                                                   // segment 1, multiplied as it goes
                DSS_GB_PUBLISH(seq * 4 + 1)
                DSS_RSTAMP(acc, r6[0], t6)
                DSS_GBG_MUL(an + GB1 + GB2, GB3, DSS_D3)                                        // segment 3's products, while wave 7 sums segment 2
                DSS_RSTAMP(PQ[GB3 / 4 - 1], r6[1], t6)
                DSS_GB_AWAIT(seq * 4 + 2)
                DSS_RSTAMP(acc, r6[2], t6)
                DSS_GB_SUMS(GB3)
                DSS_GB_PUBLISH(seq * 4 + 3)
                DSS_RSTAMP(acc, r6[3], t6)
#if DSS_RELAY_MASK
                }
#endif
                DSS_SPECULATE(lane)                          // this wave is idle from here to barrier B: candidates 0..63
                if (RS) r6[4] += DSS_NOW() - t6;
                __syncthreads();                                                        // barrier C
                DSS_GBG_LOADS(GB3, (GB1 + GB2) / 4)          // the next sample's segment 3 weights (nothing else to do before barrier B)
                __syncthreads();                                                        // barrier D
                cur ^= 1;
            }
        }
        __syncthreads();                                                                // final barrier
        if (STAMP && lane == 0 && b.trace_pcm && gridDim.x == 1)
            for (int k = 0; k < 4; ++k) b.trace_pcm[72 + k] = (float)r6[k];
        if (DSS_RELAY_STAMP && !STAMP && lane == 0 && blockIdx.x == 0)      // development builds only: into the (silent) first frame's PCM
            for (int k = 0; k < 5; ++k) reinterpret_cast<unsigned *>(pcm_out + (size_t)utt * n_frames * DSS_FRAME_SIZE)[8 + k] = r6[k];
    } else {
        // =====================================================================================================
        // role B2 + S (wave 7): GRU B, segments 2 and 4, and gates; scalar recurrences replicated across lanes
        // =====================================================================================================
        f32x2 WB[GB4R / 2];                          // the weights of segment 4's first GB4R inputs
#pragma unroll
        for (int j = 0; j < GB4R / 2; ++j) {
            WB[j].x = m.gb_w_lane[(size_t)(NA - GB4 + 2 * j) * 64 + lane];
            WB[j].y = m.gb_w_lane[(size_t)(NA - GB4 + 2 * j + 1) * 64 + lane];
        }
        const unsigned gb_addr = dss_lds_addr(&L.gb_acc[lane][0]);
        // segment 2's weights come from L2 every sample: lane offsets of its eight-block windows in m.gb_w_quad
        const char *gq = reinterpret_cast<const char *>(m.gb_w_quad);
        unsigned gvo[(GB2 / 4 + 7) / 8];
#pragma unroll
        for (int k = 0; k < (GB2 / 4 + 7) / 8; ++k) gvo[k] = (unsigned)lane * 16 + (unsigned)(GB1 / 4 + 8 * k + 4) * 1024u;
        unsigned gvo4[(GB4H / 4 + 7) / 8 + 1];       // ... and of segment 4's last GB4H inputs
#pragma unroll
        for (int k = 0; k < (GB4H / 4 + 7) / 8; ++k) gvo4[k] = (unsigned)lane * 16 + (unsigned)((NA - GB4H) / 4 + 8 * k + 4) * 1024u;
        f32x4 PQ[GB2 / 4];                           // segment 2: weights (loaded a sample ahead), then products; then segment 4's products
        const int row = lane < NB3 ? lane : 0;
        __builtin_amdgcn_s_setprio(3);               // everything this wave does is on the sample's critical path
        const float gbb1 = m.gru_b_bias[NB3 + row];
        // signal history and LPC of the current frame, element j in lane j (j < 16): one register each instead of 32,
        // a one-instruction shift per sample, and lane j publishes element j for the speculation as it is
        float ls_lane = b.last_sig[(size_t)slot * DSS_LPC_ORDER + (lane & (DSS_LPC_ORDER - 1))], lpc_lane = 0.f;
        float deemph = b.deemph[slot];
        int last_exc = b.last_exc[slot];
        DssKiss99 rng = {b.rng[slot * 4 + 0], b.rng[slot * 4 + 1], b.rng[slot * 4 + 2], b.rng[slot * 4 + 3]};
        unsigned stamp_acc[6] = {0, 0, 0, 0, 0, 0};
        unsigned t_prev = 0, r7[8] = {0, 0, 0, 0, 0, 0, 0, 0};      // r7: cycles from barrier B to the relay's way points on this wave
        int cur = 0, seq = 0;
        float pred = 0.f, upd_pred = 0.f;
        int upd_exc = 0, upd_i = 0;
        bool have_spec = false, next_exists = false, upd_pending = false;
        __syncthreads();                                             // matches role A's prologue barrier
        // Silent frames -- frame_count below FEATURES_DELAY, only at the start of a decoder's life -- first, in a loop of their
        // own: with the test and a `continue` inside the frame loop the compiler kept the product registers alive around
        // that path and wrote 80 of them to scratch at EVERY frame (0.7 GB of stores per 256 x 1 s launch).
        int f_first = 0;
        for (; f_first < nf && fc0 + f_first < DSS_FEATURES_DELAY; ++f_first) {      // lpcnet.c: frame_count <= FEATURES_DELAY -> silence
            short *pcm_frame = pcm_out + ((size_t)utt * n_frames + f_first) * DSS_FRAME_SIZE;
            for (int k = lane; k < DSS_FRAME_SIZE / 2; k += 64) reinterpret_cast<int *>(pcm_frame)[k] = 0;
            if (TRACE)
                for (int k = lane; k < DSS_FRAME_SIZE; k += 64) {
                    b.trace_exc[((size_t)utt * n_frames + f_first) * DSS_FRAME_SIZE + k] = -1.f;
                    b.trace_pcm[((size_t)utt * n_frames + f_first) * DSS_FRAME_SIZE + k] = 0.f;
                }
        }
        for (int f = f_first; f < nf; ++f) {
            short *pcm_frame = pcm_out + ((size_t)utt * n_frames + f) * DSS_FRAME_SIZE;
            const float *fo = b.frame_out + ((size_t)utt * n_frames + f) * DSS_COND_STRIDE;
            lpc_lane = fo[3 * NA + NB3 + (lane & (DSS_LPC_ORDER - 1))];
            for (int i = 0; i < DSS_FRAME_SIZE; ++i) {
                if (STAMP) t_prev = DSS_NOW();
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
                if (STAMP) { const unsigned t = DSS_NOW(); stamp_acc[0] += t - t_prev; t_prev = t; }
                if (seq == 1) {
                    DSS_GBG_LOADS(GB2, GB1 / 4)                                         // (every later sample: loaded behind barrier C of the one before)
                    __syncthreads();                                                    // barrier A (first sample only)
                }
                if (STAMP) { const unsigned t = DSS_NOW(); stamp_acc[1] += t - t_prev; t_prev = t; }
                if (upd_pending) { DSS_S_UPDATE() }                                     // previous sample's bookkeeping
                {   // off the critical path: this sample's 8 thresholds and GRU B's recurrent half
                    const uint32_t r0 = dss_kiss99_rand(rng);
                    const uint32_t r1 = dss_kiss99_rand(rng);
                    rng.z = dss_uniform(rng.z); rng.w = dss_uniform(rng.w); rng.jsr = dss_uniform(rng.jsr); rng.jcong = dss_uniform(rng.jcong);
                    if (lane < 8) {
                        const uint32_t r = lane < 4 ? r0 : r1;
                        L.thr[lane] = m.logit_table[(r >> (8 * (lane & 3))) & 0xFF];     // 1 KB table, L2/L1 resident
                    }
                }
                {   // inputs of the speculation the GRU A waves run between barriers B and C
                    const bool last_of_frame = (i == DSS_FRAME_SIZE - 1);
                    next_exists = !(last_of_frame && f == nf - 1);
                    float lp = lpc_lane;                     // lane j < 16 publishes element j
                    if (last_of_frame && next_exists && lane < DSS_LPC_ORDER)
                        lp = b.frame_out[((size_t)utt * n_frames + f + 1) * DSS_COND_STRIDE + 3 * NA + NB3 + lane];
                    // history[j-1] arrives in lane j (row_shr:1): the 15 products history[j-1] * lpc[j] of the next prediction
                    // do not depend on the candidate, so they are formed here once instead of by every speculating lane
                    const float ls_up = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, ls_lane), 0x111, 0xf, 0xf, false));
                    const float prod = ls_up * lp;
                    if (lane < DSS_LPC_ORDER) L.spec_prod[lane] = lane ? prod : lp;
                    if (lane == 0) L.spec_pred = pred;
                }
                float rec = gbb1;
#pragma unroll
                for (int j = 0; j < NB; ++j) rec += L.gb_wrec[j * NB3 + row] * L.state_b[j];
                const float sb_old = L.state_b[lane & (NB - 1)];     // the h lanes' own unit: read here, not after the chain
                __syncthreads();                                                        // barrier B
                if (STAMP) { const unsigned t = DSS_NOW(); stamp_acc[2] += t - t_prev; t_prev = t; }
                else if (RS) t_prev = DSS_NOW();
                // While wave 6 runs segment 1, this wave forms the products of segment 2 (weights from L2); while wave 6 sums
                // segment 3, those of segment 4's first GB4R inputs into the same registers.  Its own part of the chain is sums only.
                const float *an = L.state_a[DSS_NEW(cur)];
#if DSS_RELAY_MASK
                if (lane < NB3) {
#endif
                float acc;
                DSS_GBG_MUL(an + GB1, GB2, DSS_D2)
                DSS_RSTAMP(PQ[GB2 / 4 - 1], r7[0], t_prev)
                DSS_GB_AWAIT(seq * 4 + 1)
                DSS_RSTAMP(acc, r7[1], t_prev)
                DSS_GB_SUMS(GB2)
                DSS_GB_PUBLISH(seq * 4 + 2)
                DSS_RSTAMP(acc, r7[2], t_prev)
                // Segment 4: the weights of its last GB4H inputs from L2 into the registers segment 2 has left (issued BEFORE the
                // products below: after them was 1.3 % slower); its first GB4R products (weights in VGPRs); the last GB4H in place.
                DSS_GBG_LOADS_AT(GB4H, (NA - GB4H) / 4, GB4R / 4, gvo4)
                DSS_GB_PREMUL(an + (NA - GB4), GB4R, 0, 0)
                DSS_RSTAMP(PQ[GB4R / 4 - 1], r7[7], t_prev)
                DSS_GBG_MUL_AT(an + (NA - GB4H), GB4H, 3, GB4R / 4)
                DSS_RSTAMP(PQ[GB4 / 4 - 1], r7[3], t_prev)
                DSS_GB_AWAIT(seq * 4 + 3)
                DSS_RSTAMP(acc, r7[4], t_prev)
                DSS_GB_SUMS(GB4)
                DSS_RSTAMP(acc, r7[5], t_prev)
                {   // gates: lanes 0..15 z, 16..31 r, 32..47 h.  r and z travel up to their unit's h lane with gfx950's
                    // row/half swaps (VALU) instead of ds_bpermute (an LDS round trip each, on the sample's critical
                    // path); the new state is formed in the h lanes.  Only the first result of a swap is used, with
                    // distinct operands: the second one came back wrong from this compiler.
                    const float zr = dss_sigmoid_approx(L.tansig, acc + rec);
                    const unsigned zb = __builtin_bit_cast(unsigned, zr);
                    const unsigned r_row0 = __builtin_amdgcn_permlane16_swap(zb, 0u, false, false)[1];                  // lanes 0..15 <- 16..31
                    const float r_for_h = __builtin_bit_cast(float, __builtin_amdgcn_permlane32_swap(0u, r_row0, false, false)[0]);  // 32..47 <- 0..15
                    const float z_for_h = __builtin_bit_cast(float, __builtin_amdgcn_permlane32_swap(0u, zb, false, false)[0]);      // 32..47 <- 0..15
                    float hh = acc + rec * r_for_h;
                    hh = dss_tanh_approx(L.tansig, hh);
                    if (lane >= 2 * NB && lane < NB3) L.state_b[lane - 2 * NB] = z_for_h * sb_old + (1 - z_for_h) * hh;
                }
#if DSS_RELAY_MASK
                }
#endif
                __syncthreads();                                                        // barrier C
                if (RS) r7[6] += DSS_NOW() - t_prev;
                if (STAMP) { const unsigned t = DSS_NOW(); stamp_acc[3] += t - t_prev; t_prev = t; }
                DSS_GBG_LOADS(GB2, GB1 / 4)                  // the next sample's segment 2 weights, while the dual-FC waves work
                __syncthreads();                                                        // barrier D
                if (STAMP) { const unsigned t = DSS_NOW(); stamp_acc[4] += t - t_prev; t_prev = t; }
                cur ^= 1;
                int val;
                DSS_TREE_WALK(val)
                const int exc = val;
                // the next sample's prediction and mu-law indices were precomputed for every possible exc
                const float pred_next = dss_uniform(L.spec_tab_pred[exc]);     // (the GRU A waves look the mu-law indices up themselves)
                have_spec = next_exists;
                // Everything below only updates this wave's own state; except at the end of a frame (whose PCM is
                // copied out right after the loop) it is deferred until after the next barrier A, off the path
                // that the GRU A waves are waiting on.
                upd_exc = exc; upd_pred = pred; upd_i = i; upd_pending = true;       // (all wave-uniform: scalar registers)
                pred = pred_next;
                if (i == DSS_FRAME_SIZE - 1) { DSS_S_UPDATE() }
                if (STAMP) { const unsigned t = DSS_NOW(); stamp_acc[5] += t - t_prev; t_prev = t; }
            }
            // wave 7 owns L.pcm: LDS operations of one wave are ordered, no barrier needed
            for (int k = lane; k < DSS_FRAME_SIZE / 2; k += 64)
                reinterpret_cast<int *>(pcm_frame)[k] = reinterpret_cast<const int *>(L.pcm)[k];
        }
        __syncthreads();                                                                // final barrier
        if (STAMP && lane == 0 && b.trace_pcm) {        // diagnostic build only
            for (int k = 0; k < 6; ++k) b.trace_pcm[(size_t)utt * 6 + k] = (float)stamp_acc[k];
            if (gridDim.x == 1) for (int k = 0; k < 6; ++k) b.trace_pcm[64 + k] = (float)r7[k];
        }
        if (DSS_RELAY_STAMP && !STAMP && lane == 0 && blockIdx.x == 0)
            for (int k = 0; k < 8; ++k) reinterpret_cast<unsigned *>(pcm_out + (size_t)utt * n_frames * DSS_FRAME_SIZE)[k] = r7[k];
        if (lane < NB) b.gru_b_state[(size_t)slot * NB + lane] = L.state_b[lane];
        if (lane < DSS_LPC_ORDER) b.last_sig[(size_t)slot * DSS_LPC_ORDER + lane] = ls_lane;
        if (lane == 0) {
            b.deemph[slot] = deemph;
            b.last_exc[slot] = last_exc;
            b.rng[slot * 4 + 0] = rng.z; b.rng[slot * 4 + 1] = rng.w; b.rng[slot * 4 + 2] = rng.jsr; b.rng[slot * 4 + 3] = rng.jcong;
        }
    }
}

// Number of CUs of the current device, asked once per device (the eager streaming tick launches this kernel every 40 ms).
static int dss_cu_count()
{
    static std::mutex mu;
    static int cached[64] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 256;
    std::lock_guard<std::mutex> lk(mu);
    int &c = cached[dev & 63];
    if (!c && hipDeviceGetAttribute(&c, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) c = 256;
    return c;
}

int dss_launch_sample_network(const DssModelDev &m, DssBatchDev &b, int n_utts, int n_frames, short *d_pcm, int trace,
                              int pair, hipStream_t s)
{
    if (!m.fast_ok || trace >= 16) return dss_launch_sample_network_generic(m, b, n_utts, n_frames, d_pcm, trace & 15, s);
    // Calls with more rows than the chip has CUs run two utterances per workgroup (lpcnet_sample_pair.hip: the same
    // roles with the utterances as the halves of packed fp32 instructions); with a CU per utterance the one-utterance
    // form below is faster.  pair: 0 = this rule, -1 = never, 2 = always (tests, A/B timing).
    b.utt0 = 0;
    int n_pair = 0;                              // rows [0, n_pair) on the pair kernel, the rest on the kernel below
    if (pair >= 0 && trace <= 2 && !(trace == 2 && (b.slot_of || b.count_of)) && dss_pair_fits(m) &&
        (pair == 2 || (!trace && n_utts > 128 && n_utts > dss_cu_count()))) {
        n_pair = n_utts;
        // Uniform calls: workgroups run in rounds of one per CU, a round of the pair kernel takes 1.67x a round of the
        // one-utterance kernel and carries twice the rows.  Full rounds go to the pair kernel; a remainder of at most one
        // row per CU is cheaper as one round of the one-utterance kernel (600 rows on 256 CUs: 71 + 43 ms instead of 2 x 71).
        if (pair == 0 && !b.slot_of && !b.count_of) {
            const int per_round = 2 * dss_cu_count();
            const int rem = n_utts % per_round;
            if (rem > 0 && 2 * rem <= per_round && n_utts > per_round) n_pair = n_utts - rem;
        }
        const int rc = dss_launch_sample_network_pair(m, b, n_pair, n_frames, d_pcm, trace, s);
        if (rc || n_pair == n_utts) return rc;
        b.utt0 = n_pair;
    }
    const int n_rows = n_utts - n_pair;          // rows of this launch
    const size_t dyn = ((size_t)m.hblk_floats * sizeof(float) + 15) & ~(size_t)15;
    // two register-slot capacities are compiled: 10 per gate (no spills) and 12 (a few spilled registers)
    const bool z10 = m.zr_cap <= 10;
    static std::mutex attr_mu;                  // states on different devices may launch from different threads
    static unsigned long long attr_set = 0;     // per device: the attribute belongs to the device's code object
    int dev = 0;
    DSS_HIP_CHECK(hipGetDevice(&dev));
    {
    std::lock_guard<std::mutex> attr_lk(attr_mu);
    if (!(attr_set >> (dev & 63) & 1)) {      // one workgroup uses (almost) the whole 160 KB of the CU
#define DSS_SET_ATTR(K) DSS_HIP_CHECK(hipFuncSetAttribute((const void *)K, hipFuncAttributeMaxDynamicSharedMemorySize, DSS_HBLK_BYTES))
        DSS_SET_ATTR((lpcnet_sample_kernel<false, false, 10, false, false>)); DSS_SET_ATTR((lpcnet_sample_kernel<false, false, 12, false, false>));
        DSS_SET_ATTR((lpcnet_sample_kernel<false, false, 10, true, false>));  DSS_SET_ATTR((lpcnet_sample_kernel<false, false, 12, true, false>));
        DSS_SET_ATTR((lpcnet_sample_kernel<true, false, 10, false, false>));  DSS_SET_ATTR((lpcnet_sample_kernel<true, false, 12, false, false>));
        DSS_SET_ATTR((lpcnet_sample_kernel<false, true, 10, false, false>));  DSS_SET_ATTR((lpcnet_sample_kernel<false, true, 12, false, false>));
        // models with z/r tails or long h lists (m.ext): always on the 10-slot layout, which has registers to spare for them
        DSS_SET_ATTR((lpcnet_sample_kernel<false, false, 10, false, true>));  DSS_SET_ATTR((lpcnet_sample_kernel<false, false, 10, true, true>));
        DSS_SET_ATTR((lpcnet_sample_kernel<true, false, 10, false, true>));
#undef DSS_SET_ATTR
        attr_set |= 1ull << (dev & 63);
    }
    }
    const bool ragged = b.slot_of || b.count_of;
    if (ragged && trace == 2) { dss_set_error("phase stamps are taken on uniform calls only"); return DSS_EINVAL; }
    if (m.ext && trace == 2) { dss_set_error("phase stamps are not built for models with z/r tails"); return DSS_EINVAL; }
#define DSS_LAUNCH(T, S2, R)                                                                                           \
    do {                                                                                                               \
        if (z10) hipLaunchKernelGGL((lpcnet_sample_kernel<T, S2, 10, R, false>), dim3(n_rows), dim3(512), dyn, s, m, b, n_frames, d_pcm); \
        else hipLaunchKernelGGL((lpcnet_sample_kernel<T, S2, 12, R, false>), dim3(n_rows), dim3(512), dyn, s, m, b, n_frames, d_pcm);     \
    } while (0)
#define DSS_LAUNCH_EXT(T, R) hipLaunchKernelGGL((lpcnet_sample_kernel<T, false, 10, R, true>), dim3(n_rows), dim3(512), dyn, s, m, b, n_frames, d_pcm)
    if (m.ext) {
        if (trace) DSS_LAUNCH_EXT(true, false);
        else if (ragged) DSS_LAUNCH_EXT(false, true);
        else DSS_LAUNCH_EXT(false, false);
    }
    else if (trace == 2) DSS_LAUNCH(false, true, false);   // diagnostic: phase stamps (never used for timing claims)
    else if (trace) DSS_LAUNCH(true, false, false);
    else if (ragged) DSS_LAUNCH(false, false, true);
    else DSS_LAUNCH(false, false, false);
#undef DSS_LAUNCH
#undef DSS_LAUNCH_EXT
    b.utt0 = 0;
    DSS_HIP_CHECK(hipGetLastError());
    return DSS_OK;
}
