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
//   GRU B input weights (18k floats)              VGPRs of waves 6 and 7 (lane = row; 208 + 112 inputs), last 64 in LDS
//   dual-FC (8k floats)                           VGPRs of waves 0..3 (lane = tree node)
// Roles inside the 512-thread workgroup (8 waves, 2 per SIMD); three workgroup barriers B C D per sample (a fourth,
// A, only on the first sample of a call):
//   waves 0..5  GRU A (dss_role_a, compiled once with and once without the dual-FC).  D..B: tree walk over the
//               decision bits, speculated embedding indices, embedding rows, z and r chains, activations, new state.
//               B..C: the h-gate recurrent chain and the z/r block products of the NEXT sample (they need only the
//               new state) and the speculation over the 256 possible excitations -- hidden under GRU B.
//               C..D (waves 0..3): dual-FC logits of all 255 tree nodes -> decision bits.
//   wave 6      GRU B, inputs 0..207: lane = output row, one sequential chain per row, then hands the partial
//               sums to wave 7 through LDS (flag, no barrier).
//   wave 7      GRU B inputs 208..383 (the products of the first 48 formed while it waits for wave 6) + gates, and
//               the scalar recurrences: mu-law / de-emphasis / PCM bookkeeping, kiss99 thresholds, its own tree walk.
// Summation order inside every row is exactly the C source's (one product at a time, ascending input),
// and the library is built with -ffp-contract=off, so results are bit-identical to the scalar C path.
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
#undef HC                                  // a constant of the role here: DSS_HC, or DSS_HCX with the extended paths

#define GBH6 208                          // GRU B inputs whose weights sit in wave 6's VGPRs (0..207)
#define GBH7 112                          // ... in wave 7's VGPRs (208..319); wave 7 also carries the scalar state
#define GBHL (NA - GBH6 - GBH7)           // ... and the last 64 inputs' weights in LDS, [row][GBL_STRIDE]
#define GBL_STRIDE 68
#ifndef GBP
#define GBP 48                            // ... of wave 7's inputs, the first GBP are multiplied while it waits for wave 6
#endif
struct SampleLds {
    float state_a[2][NA + 4];             // double-buffered GRU A state; "column 96" of either buffer is four zeros: the
                                          //   input of the h-gate slots a row group does not use (see DSS_H_CHAIN)
    float gb_wrec[NB * NB3];              // GRU B recurrent weights [16][48]
    float gb_wl[NB3 * GBL_STRIDE];        // GRU B input weights of the last GBHL inputs, row-major
    float tansig[208];
    float ulaw2lin[256];
    float spec_tab_pred[256];             // speculation over all 256 excitation values (see role A, B..C):
    unsigned short spec_tab_idx[256];     //   next sample's prediction and its two mu-law indices (su | pu << 8)
    float spec_ls[DSS_LPC_ORDER];         // inputs of the speculation, published by wave 7: signal history,
    float spec_lpc[DSS_LPC_ORDER];        //   the LPC of the next sample's frame,
    float spec_pred;                      //   and this sample's prediction
    float pad1[3];
    float gb_acc[64];                     // GRU B partial sums handed from wave 6 to wave 7
    float ah[NA];                         // h-gate pre-activation: written by a unit's h lane, read by its z/r lane
    float state_b[NB];
    float thr[8];
    unsigned bits[8];                     // decision bit of every tree node (256 bits)
    int idx[4];                           // last_sig_ulaw, pred_ulaw, last_exc
    int gb_flag;                          // sequence number of the sample whose gb_acc is valid
    int pad[3];
    short pcm[DSS_FRAME_SIZE];
};

// the same chain over the inputs whose weights live in LDS (wave 7's tail): 8 inputs per group
#define DSS_GBL_LOAD(T, G)                                                                       \
    {                                                                                            \
        T[0] = *reinterpret_cast<const f32x4 *>(al + 8 * (G));                                   \
        T[1] = *reinterpret_cast<const f32x4 *>(al + 8 * (G) + 4);                               \
        T[2] = *reinterpret_cast<const f32x4 *>(wl + 8 * (G));                                   \
        T[3] = *reinterpret_cast<const f32x4 *>(wl + 8 * (G) + 4);                               \
    }
#define DSS_GBL_GROUP(T)                                                                         \
    _Pragma("unroll") for (int u = 0; u < 2; ++u) {                                              \
        const f32x2 p0 = T[2 + u].lo * T[u].lo;                                                  \
        const f32x2 p1 = T[2 + u].hi * T[u].hi;                                                  \
        acc += p0.x;                                                                             \
        acc += p0.y;                                                                             \
        acc += p1.x;                                                                             \
        acc += p1.y;                                                                             \
    }

#define DSS_TREE_WALK(VAL) DSS_TREE_WALK_AT(VAL, L.bits)

// one candidate excitation value of the speculation, for wave 6 (inputs published by wave 7 right after its tree walk)
#define DSS_SPECULATE(CAND)                                                                      \
    {                                                                                            \
        const int cand_ = (CAND);                                                                \
        const float pcm_c = L.spec_pred + L.ulaw2lin[cand_];                                     \
        float pc = 0;                                                                            \
        pc -= pcm_c * L.spec_lpc[0];                                                             \
        _Pragma("unroll") for (int j = 1; j < DSS_LPC_ORDER; ++j) pc -= L.spec_ls[j - 1] * L.spec_lpc[j]; \
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
        deemph = pcm;                                                                            \
        if (pcm < -32767) pcm = -32767;                                                          \
        if (pcm > 32767) pcm = 32767;                                                            \
        if (lane == 0) L.pcm[upd_i] = (short)(int)floor(.5 + (double)pcm);                       \
        upd_pending = false;                                                                     \
    }

// =====================================================================================================
// role A: GRU A (+ dual-FC on waves 0..3, + the speculation on waves 4..5).  Two instantiations share this text:
//   waves 0..3  HAS_FC, at most 8 z/r register slots per gate (the dual-FC weights take 32 registers);
//   waves 4..5  no dual-FC, all Z slots -- the host gives them the row groups with the most z/r blocks.
// Keeping the two apart is what keeps either under the 256-VGPR budget without spill reloads in the sample loop.
// =====================================================================================================
template <bool TRACE, bool STAMP, int Z, bool HAS_FC, bool EXT>
__device__ __forceinline__ void dss_role_a(SampleLds &L, float *hblk_lds, const DssModelDev &m, const DssBatchDev &b,
                                           int n_frames, int utt, int slot, int nf, int fc0, int tid, int wave, int lane)
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
    const float u2l_c = L.ulaw2lin[(HAS_FC ? 128 + tid : tid - 256) & 255];   // this lane's excitation candidate (waves 0, 1, 4, 5)
    const int level = 31 - __clz(tid | 1);                       // FC node = (1 << level) | prefix
    const bool recur_first = m.h.gru_a_order == DSS_GRUA_RECUR_FIRST;     // wave-uniform (kernel argument)
    int cur = 0;
    float st = L.state_a[0][unit];
    unsigned long long sa[8] = {0, 0, 0, 0, 0, 0, 0, 0}, ta = 0;   // diagnostic build only
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
            if (STAMP) ta = __builtin_readcyclecounter();
            {
                // the nine embedding values of this lane: three 12-byte loads from the lane-ordered copies of the tables
                // (m.embed_lane: [index][lane][gate]), 768 contiguous bytes per wave and table
                typedef float f32x3 __attribute__((ext_vector_type(3)));
                const f32x3 es = *reinterpret_cast<const f32x3 *>(m.embed_lane[0] + ((unsigned)si * NA + (unsigned)tid) * 3);
                const f32x3 ep = *reinterpret_cast<const f32x3 *>(m.embed_lane[1] + ((unsigned)pi * NA + (unsigned)tid) * 3);
                const f32x3 ee = *reinterpret_cast<const f32x3 *>(m.embed_lane[2] + ((unsigned)ei * NA + (unsigned)tid) * 3);
                const float es0 = es.x, es1 = es.y, es2 = es.z, ep0 = ep.x, ep1 = ep.y, ep2 = ep.z, ee0 = ee.x, ee1 = ee.y, ee2 = ee.z;
                if (STAMP) { unsigned long long t = __builtin_readcyclecounter(); sa[0] += t - ta; ta = t; }
                const float gz = ((cz + es0) + ep0) + ee0;                          // compute_gru_a_input
                const float gr = ((cr + es1) + ep1) + ee1;
                const float gh = ((ch + es2) + ep2) + ee2;
                // nnet.c 2021 (default): (bias + diag*state) + input, then the blocks in idx order;
                // nnet.c 2019-20 (blob flag): the blocks first, the input last
                if (!recur_first) { az = az + gz; ar = ar + gr; }
                if (STAMP) { asm volatile("" :: "v"(az), "v"(ar)); unsigned long long t = __builtin_readcyclecounter(); sa[1] += t - ta; ta = t; }
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
                    const char *xb = reinterpret_cast<const char *>(L.state_a[cur]);
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
            DSS_H_CHAIN(L.state_a[cur ^ 1])                      // next sample's h chain, under GRU B
            DSS_ZR_PRODUCTS(L.state_a[cur ^ 1])                  // ... and its z/r block products (sums come later)
            if (wave == 5 || wave < 2) {
                // Speculation over all 256 possible excitation values of THIS sample, one candidate per lane of wave 5
                // (candidates 64..127), of waves 0, 1, which have the lightest B..C load of the dual-FC waves (128..255),
                // and of wave 6 once it has handed its half of the GRU B chain over (0..63): the next sample's LPC prediction and mu-law indices, so that once the tree walk has
                // picked the value nobody has to run the two ~40-step dependent chains.  Same expressions, same order
                // as lpcnet_synthesize_tail_impl().
                const int cand = HAS_FC ? 128 + tid : tid - 256;
                const float pcm_c = L.spec_pred + u2l_c;
                float pc = 0;
                pc -= pcm_c * L.spec_lpc[0];
#pragma unroll
                for (int j = 1; j < DSS_LPC_ORDER; ++j) pc -= L.spec_ls[j - 1] * L.spec_lpc[j];
                const int su_c = dss_lin2ulaw(pcm_c), pu_c = dss_lin2ulaw(pc);
                L.spec_tab_pred[cand] = pc;
                L.spec_tab_idx[cand] = (unsigned short)(su_c | (pu_c << 8));
            }
            if (STAMP) { unsigned long long t = __builtin_readcyclecounter(); sa[5] += t - ta; ta = t; }
            __syncthreads();                                                        // barrier C
            if (STAMP) { unsigned long long t = __builtin_readcyclecounter(); sa[6] += t - ta; ta = t; }
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
            if (STAMP) { unsigned long long t = __builtin_readcyclecounter(); sa[7] += t - ta; ta = t; }
            cur ^= 1;
        }
    }
    __syncthreads();                                                                // final barrier
    if (STAMP && lane == 0 && b.trace_exc)
        for (int k = 0; k < 8; ++k) b.trace_exc[((size_t)utt * 6 + wave) * 8 + k] = (float)sa[k];
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
    constexpr bool RG = RAGGED || TRACE;
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
    for (int k = tid; k < NB3 * GBHL; k += 512) {
        const int row = k / GBHL, j = k - row * GBHL;
        L.gb_wl[row * GBL_STRIDE + j] = m.gb_w_lane[(size_t)(GBH6 + GBH7 + j) * 64 + row];
    }
    if (tid < 201) L.tansig[tid] = m.tansig[tid];
    if (tid < 256) L.ulaw2lin[tid] = m.ulaw2lin[tid];
    if (tid < NA) L.state_a[0][tid] = b.gru_a_state[(size_t)slot * NA + tid];
    if (tid < 8) L.state_a[tid >> 2][NA + (tid & 3)] = 0.f;
    if (tid < NB) L.state_b[tid] = b.gru_b_state[(size_t)slot * NB + tid];
    if (tid == 0) L.gb_flag = 0;
    const int fc0 = b.fc0[utt];
    __syncthreads();

    if (wave < 4) {
        dss_role_a<TRACE, STAMP, (Z < 8 ? Z : 8), true, EXT>(L, hblk_lds, m, b, n_frames, utt, slot, nf, fc0, tid, wave, lane);
    } else if (wave < 6) {
        dss_role_a<TRACE, STAMP, Z, false, EXT>(L, hblk_lds, m, b, n_frames, utt, slot, nf, fc0, tid, wave, lane);
    } else if (wave == 6) {
        // =====================================================================================================
        // role B1: GRU B over inputs 0..207, lane = row (0..15 z, 16..31 r, 32..47 h)
        // =====================================================================================================
        f32x2 WB[GBH6 / 2];
#pragma unroll
        for (int j = 0; j < GBH6 / 2; ++j) {
            WB[j].x = m.gb_w_lane[(size_t)(2 * j) * 64 + lane];
            WB[j].y = m.gb_w_lane[(size_t)(2 * j + 1) * 64 + lane];
        }
        const int row = lane < NB3 ? lane : 0;
        __builtin_amdgcn_s_setprio(3);               // everything this wave does is on the sample's critical path
        const float gbb0 = m.gru_b_bias[row];
        int cur = 0, seq = 0;
        __syncthreads();                                             // matches role A's prologue barrier
        for (int f = 0; f < nf; ++f) {
            if (fc0 + f < DSS_FEATURES_DELAY) continue;
            const float gbc = b.frame_out[((size_t)utt * n_frames + f) * DSS_COND_STRIDE + 3 * NA + row];
            for (int i = 0; i < DSS_FRAME_SIZE; ++i) {
                float acc = gbb0 + gbc;                                                 // compute_gruB
                ++seq;
                if (seq == 1) __syncthreads();                                          // barrier A (first sample only)
                __syncthreads();                                                        // barrier B
                const float *an = L.state_a[cur ^ 1];
                DSS_GB_CHAIN(an, GBH6)
                L.gb_acc[lane] = acc;
                __hip_atomic_store(&L.gb_flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                DSS_SPECULATE(lane)                          // this wave is idle from here to barrier B: candidates 0..63
                __syncthreads();                                                        // barrier C
                __syncthreads();                                                        // barrier D
                cur ^= 1;
            }
        }
        __syncthreads();                                                                // final barrier
    } else {
        // =====================================================================================================
        // role B2 + S (wave 7): GRU B inputs 208..383 and gates; scalar recurrences replicated across lanes
        // =====================================================================================================
        f32x2 WB[GBH7 / 2];
#pragma unroll
        for (int j = 0; j < GBH7 / 2; ++j) {
            WB[j].x = m.gb_w_lane[(size_t)(GBH6 + 2 * j) * 64 + lane];
            WB[j].y = m.gb_w_lane[(size_t)(GBH6 + 2 * j + 1) * 64 + lane];
        }
        const int row = lane < NB3 ? lane : 0;
        __builtin_amdgcn_s_setprio(3);               // everything this wave does is on the sample's critical path
        const float gbb1 = m.gru_b_bias[NB3 + row];
        // signal history and LPC of the current frame, element j in lane j (j < 16): one register each instead of 32,
        // a one-instruction shift per sample, and lane j publishes element j for the speculation as it is
        float ls_lane = b.last_sig[(size_t)slot * DSS_LPC_ORDER + (lane & (DSS_LPC_ORDER - 1))], lpc_lane = 0.f;
        float deemph = b.deemph[slot];
        int last_exc = b.last_exc[slot];
        DssKiss99 rng = {b.rng[slot * 4 + 0], b.rng[slot * 4 + 1], b.rng[slot * 4 + 2], b.rng[slot * 4 + 3]};
        unsigned long long stamp_acc[6] = {0, 0, 0, 0, 0, 0};
        unsigned long long t_prev = 0;
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
                if (upd_pending) { DSS_S_UPDATE() }                                     // previous sample's bookkeeping
                {   // off the critical path: this sample's 8 thresholds and GRU B's recurrent half
                    const uint32_t r0 = dss_kiss99_rand(rng);
                    const uint32_t r1 = dss_kiss99_rand(rng);
                    if (lane < 8) {
                        const uint32_t r = lane < 4 ? r0 : r1;
                        L.thr[lane] = m.logit_table[(r >> (8 * (lane & 3))) & 0xFF];     // 1 KB table, L2/L1 resident
                    }
                }
                {   // inputs of the speculation the GRU A waves run between barriers B and C
                    const bool last_of_frame = (i == DSS_FRAME_SIZE - 1);
                    next_exists = !(last_of_frame && f == nf - 1);
                    float lp = lpc_lane;                     // lane j < 16 publishes element j
                    const float ls = ls_lane;
                    if (last_of_frame && next_exists && lane < DSS_LPC_ORDER)
                        lp = b.frame_out[((size_t)utt * n_frames + f + 1) * DSS_COND_STRIDE + 3 * NA + NB3 + lane];
                    if (lane < DSS_LPC_ORDER) { L.spec_lpc[lane] = lp; L.spec_ls[lane] = ls; }
                    if (lane == 0) L.spec_pred = pred;
                }
                float rec = gbb1;
#pragma unroll
                for (int j = 0; j < NB; ++j) rec += L.gb_wrec[j * NB3 + row] * L.state_b[j];
                const float sb_old = L.state_b[lane & (NB - 1)];     // the h lanes' own unit: read here, not after the chain
                __syncthreads();                                                        // barrier B
                if (STAMP) { unsigned long long t = __builtin_readcyclecounter(); stamp_acc[2] += t - t_prev; t_prev = t; }
                // While wave 6 runs the first half of the chain, this wave forms the products of its first GBP inputs (it
                // has about 1900 idle cycles and, until its own part of the chain starts, the registers of the prefetch
                // buffers): when wave 6 hands over, those inputs cost one sum each instead of 1.8 instructions.
                const float *an = L.state_a[cur ^ 1] + GBH6;
                f32x4 PQ[GBP / 4];
#pragma unroll
                for (int g = 0; g < GBP / 16; ++g) {
                    f32x4 xq[4];
                    DSS_GB_LOAD(xq, an, g)
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        PQ[4 * g + u].lo = WB[2 * (4 * g + u)] * xq[u].lo;
                        PQ[4 * g + u].hi = WB[2 * (4 * g + u) + 1] * xq[u].hi;
                    }
                    // pinned: left alone, the compiler sinks the multiplications below the hand-off wait
                    asm volatile("" : "+v"(PQ[4 * g].x), "+v"(PQ[4 * g].y), "+v"(PQ[4 * g].z), "+v"(PQ[4 * g].w),
                                      "+v"(PQ[4 * g + 1].x), "+v"(PQ[4 * g + 1].y), "+v"(PQ[4 * g + 1].z), "+v"(PQ[4 * g + 1].w),
                                      "+v"(PQ[4 * g + 2].x), "+v"(PQ[4 * g + 2].y), "+v"(PQ[4 * g + 2].z), "+v"(PQ[4 * g + 2].w),
                                      "+v"(PQ[4 * g + 3].x), "+v"(PQ[4 * g + 3].y), "+v"(PQ[4 * g + 3].z), "+v"(PQ[4 * g + 3].w));
                }
                f32x4 avA[4], avB[4];
                DSS_GB_LOAD(avA, an + GBP, 0)                // first state group of the part it multiplies on the fly
                __builtin_amdgcn_sched_barrier(0);
                while (__hip_atomic_load(&L.gb_flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) != seq)
                    ;                                    // tight poll: one LDS round trip per iteration is pause enough
                float acc = L.gb_acc[lane];
#pragma unroll
                for (int q = 0; q < GBP / 4; ++q) {
                    acc += PQ[q].x;
                    acc += PQ[q].y;
                    acc += PQ[q].z;
                    acc += PQ[q].w;
                }
                {
                    DSS_GB_CHAIN_RUN_OFF(an + GBP, GBH7 - GBP, GBP / 2)
                    const float *al = an + GBH7, *wl = L.gb_wl + row * GBL_STRIDE;
                    f32x4 tA[4], tB[4];                   // [0..1] state, [2..3] weights of two groups of 4 inputs
                    DSS_GBL_LOAD(tA, 0)
#pragma unroll
                    for (int g = 0; g < GBHL / 8; g += 2) {
                        if (g + 1 < GBHL / 8) DSS_GBL_LOAD(tB, g + 1)
                        __builtin_amdgcn_sched_barrier(0);
                        if (g + 1 < GBHL / 8) DSS_WAIT_LGKM(4); else DSS_WAIT_LGKM(0);
                        __builtin_amdgcn_sched_barrier(0);
                        DSS_GBL_GROUP(tA)
                        __builtin_amdgcn_sched_barrier(0);
                        if (g + 2 < GBHL / 8) DSS_GBL_LOAD(tA, g + 2)
                        __builtin_amdgcn_sched_barrier(0);
                        if (g + 1 < GBHL / 8) {
                            if (g + 2 < GBHL / 8) DSS_WAIT_LGKM(4); else DSS_WAIT_LGKM(0);
                            __builtin_amdgcn_sched_barrier(0);
                            DSS_GBL_GROUP(tB)
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
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
                __syncthreads();                                                        // barrier C
                if (STAMP) { unsigned long long t = __builtin_readcyclecounter(); stamp_acc[3] += t - t_prev; t_prev = t; }
                __syncthreads();                                                        // barrier D
                if (STAMP) { unsigned long long t = __builtin_readcyclecounter(); stamp_acc[4] += t - t_prev; t_prev = t; }
                cur ^= 1;
                int val;
                DSS_TREE_WALK(val)
                const int exc = val;
                // the next sample's prediction and mu-law indices were precomputed for every possible exc
                const float pred_next = L.spec_tab_pred[exc];     // (the GRU A waves look the mu-law indices up themselves)
                have_spec = next_exists;
                // Everything below only updates this wave's own state; except at the end of a frame (whose PCM is
                // copied out right after the loop) it is deferred until after the next barrier A, off the path
                // that the GRU A waves are waiting on.
                upd_exc = exc; upd_pred = pred; upd_i = i; upd_pending = true;
                pred = pred_next;
                if (i == DSS_FRAME_SIZE - 1) { DSS_S_UPDATE() }
                if (STAMP) { unsigned long long t = __builtin_readcyclecounter(); stamp_acc[5] += t - t_prev; t_prev = t; }
            }
            // wave 7 owns L.pcm: LDS operations of one wave are ordered, no barrier needed
            for (int k = lane; k < DSS_FRAME_SIZE / 2; k += 64)
                reinterpret_cast<int *>(pcm_frame)[k] = reinterpret_cast<const int *>(L.pcm)[k];
        }
        __syncthreads();                                                                // final barrier
        if (STAMP && lane == 0 && b.trace_pcm)          // diagnostic build only
            for (int k = 0; k < 6; ++k) b.trace_pcm[(size_t)utt * 6 + k] = (float)stamp_acc[k];
        if (lane < NB) b.gru_b_state[(size_t)slot * NB + lane] = L.state_b[lane];
        if (lane < DSS_LPC_ORDER) b.last_sig[(size_t)slot * DSS_LPC_ORDER + lane] = ls_lane;
        if (lane == 0) {
            b.deemph[slot] = deemph;
            b.last_exc[slot] = last_exc;
            b.rng[slot * 4 + 0] = rng.z; b.rng[slot * 4 + 1] = rng.w; b.rng[slot * 4 + 2] = rng.jsr; b.rng[slot * 4 + 3] = rng.jcong;
        }
    }
}

int g_dss_latency_kernel = 0;

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
    // One utterance per workgroup: the packed-h form (lpcnet_sample_pkh.hip) for every model it fits; the kernel below for
    // models with extended paths or 12 z/r register slots (g_dss_latency_kernel: tests and A/B timing)
    if (g_dss_latency_kernel != 1 && dss_pkh_fits(m)) {
        const int rc = dss_launch_sample_network_pkh(m, b, n_rows, n_frames, d_pcm, trace, s);
        b.utt0 = 0;
        return rc;
    }
    if (g_dss_latency_kernel == 2) { b.utt0 = 0; dss_set_error("the packed-h kernel does not fit this model"); return DSS_EINVAL; }
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
