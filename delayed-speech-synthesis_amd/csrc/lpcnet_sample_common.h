// csrc/lpcnet_sample_common.h -- register-level building blocks shared by the two CU-resident sample-rate kernels
// (lpcnet_sample.hip: one utterance per workgroup, latency-optimised; lpcnet_sample_pair.hip: two utterances per
// workgroup as the halves of packed fp32 instructions, throughput-optimised).  Every macro keeps the summation order of xiph's
// sparse_sgemv_accum8x4 / sgemv_accum (src/vec.h generic path): one product at a time, ascending input.
#pragma once
#include "dss_common.h"
#include "lpcnet_device.h"

#define NA DSS_GRU_A
#define NB DSS_GRU_B
#define NB3 (3 * DSS_GRU_B)
#define ZRC Z                             // template parameter: z/r register slots per gate (10 or 12)
#define ZRL DSS_ZRC                       // slot stride of the host layout (zr_w, zr_col)
#define HC DSS_HC
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// a value every lane holds alike, moved to a scalar register (the relay waves have no vector register to spare)
__device__ __forceinline__ float dss_uniform(float x) { return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, x))); }
__device__ __forceinline__ int dss_uniform(int x) { return __builtin_amdgcn_readfirstlane(x); }
__device__ __forceinline__ uint32_t dss_uniform(uint32_t x) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)x); }

// LDS byte address of an object in the workgroup's LDS (for the hand-written ds_ instructions below)
__device__ __forceinline__ unsigned dss_lds_addr(const void *p)
{
    return (unsigned)(unsigned long long)(__attribute__((address_space(3))) const void *)p;
}


// where DSS_H_CHAIN leaves the finished h-gate pre-activation of unit `uh` (the including kernel may redefine it)
#ifndef DSS_H_STORE
#define DSS_H_STORE(V) L.ah[uh] = (V)
#endif
// slots beyond DSS_HC of a long h list (models with skewed sparsity; only the latency kernel defines it)
#ifndef DSS_H_TAIL
#define DSS_H_TAIL
#endif

// one 8x4 block applied to one row: four products accumulated one at a time (sparse_sgemv_accum8x4 order)
// z/r chunk C = slots 2C, 2C+1 of the z list and of the r list (Q[0..1] z, Q[2..3] r)
#define DSS_ZR_COL(S) ((PZ[(S) >> 2] >> (8 * ((S) & 3))) & 0xFFu)      /* S in layout numbering: z s, r ZRL + s */
#define DSS_ZR_LOADX(Q, C)                                                                       \
    {                                                                                            \
        Q[0] = *reinterpret_cast<const f32x4 *>(xbase + DSS_ZR_COL(2 * (C)) * 16);               \
        Q[1] = *reinterpret_cast<const f32x4 *>(xbase + DSS_ZR_COL(2 * (C) + 1) * 16);           \
        Q[2] = *reinterpret_cast<const f32x4 *>(xbase + DSS_ZR_COL(ZRC + 2 * (C)) * 16);         \
        Q[3] = *reinterpret_cast<const f32x4 *>(xbase + DSS_ZR_COL(ZRC + 2 * (C) + 1) * 16);     \
    }
#define DSS_ZR_MUL(Q, C)                                                                         \
    {                                                                                            \
        Q[0].lo = WZ[2 * (C)].lo * Q[0].lo;             Q[0].hi = WZ[2 * (C)].hi * Q[0].hi;             \
        Q[1].lo = WZ[2 * (C) + 1].lo * Q[1].lo;         Q[1].hi = WZ[2 * (C) + 1].hi * Q[1].hi;         \
        Q[2].lo = WZ[ZRC + 2 * (C)].lo * Q[2].lo;       Q[2].hi = WZ[ZRC + 2 * (C)].hi * Q[2].hi;       \
        Q[3].lo = WZ[ZRC + 2 * (C) + 1].lo * Q[3].lo;   Q[3].hi = WZ[ZRC + 2 * (C) + 1].hi * Q[3].hi;   \
    }
#define DSS_ZR_ADD(Q)                                                                            \
    {                                                                                            \
        az += Q[0].x; ar += Q[2].x; az += Q[0].y; ar += Q[2].y;                                  \
        az += Q[0].z; ar += Q[2].z; az += Q[0].w; ar += Q[2].w;                                  \
        az += Q[1].x; ar += Q[3].x; az += Q[1].y; ar += Q[3].y;                                  \
        az += Q[1].z; ar += Q[3].z; az += Q[1].w; ar += Q[3].w;                                  \
    }
// z/r block products of one lane for the coming sample: state vectors from LDS, weights from registers
#define DSS_ZR_PRODUCTS(XBUF)                                                                    \
    {                                                                                            \
        const char *xb = reinterpret_cast<const char *>(XBUF);                                   \
        _Pragma("unroll") for (int s2 = 0; s2 < ZRC; s2 += 2) {                                  \
            if (s2 >= nzr) break;                                                                \
            _Pragma("unroll") for (int u = 0; u < 2; ++u) {                                      \
                const f32x4 xz = *reinterpret_cast<const f32x4 *>(xb + DSS_ZR_COL(s2 + u) * 16); \
                const f32x4 xr = *reinterpret_cast<const f32x4 *>(xb + DSS_ZR_COL(ZRL + s2 + u) * 16); \
                PR[s2 + u].lo = WZ[s2 + u].lo * xz.lo;             PR[s2 + u].hi = WZ[s2 + u].hi * xz.hi;             \
                PR[ZRC + s2 + u].lo = WZ[ZRC + s2 + u].lo * xr.lo; PR[ZRC + s2 + u].hi = WZ[ZRC + s2 + u].hi * xr.hi; \
            }                                                                                    \
        }                                                                                        \
    }
// h-gate chain of one lane: rbh + dgh*st, then its row group's blocks in idx order.  Chunk C = slots 2C, 2C+1;
// block records (LDS) and state vectors of the next chunk are fetched while the current chunk's sums run.
#define DSS_H_COL(S) ((PH[(S) >> 2] >> (8 * ((S) & 3))) & 0xFFu)
#define DSS_H_LOAD(Q, C)                                                                         \
    {                                                                                            \
        Q[0] = *reinterpret_cast<const f32x4 *>(hw + (2 * (C)) * 128);                           \
        Q[1] = *reinterpret_cast<const f32x4 *>(hw + (2 * (C) + 1) * 128);                       \
        Q[2] = *reinterpret_cast<const f32x4 *>(xbase + DSS_H_COL(2 * (C)) * 16);                \
        Q[3] = *reinterpret_cast<const f32x4 *>(xbase + DSS_H_COL(2 * (C) + 1) * 16);            \
    }
#define DSS_H_MAC(Q)                                                                             \
    {                                                                                            \
        const f32x2 p0 = Q[0].lo * Q[2].lo, p1 = Q[0].hi * Q[2].hi;                              \
        const f32x2 p2 = Q[1].lo * Q[3].lo, p3 = Q[1].hi * Q[3].hi;                              \
        ah += p0.x; ah += p0.y; ah += p1.x; ah += p1.y;                                          \
        ah += p2.x; ah += p2.y; ah += p3.x; ah += p3.y;                                          \
    }
// Chunks C (in HA) and C + 1 (into HB), C even; on entry the operands of chunk C are in flight.  The operands of the next
// two chunks are fetched UNCONDITIONALLY whenever chunk C exists -- past the end of a wave's lists they are the following
// records times "column 96" (the image is padded by four records), never used -- so that every path into a MAC has the
// same number of LDS reads outstanding behind the ones it needs and the compiler waits with lgkmcnt(4).  With the
// fetches under their own tests (rounds 1-3) half of the MACs waited with lgkmcnt(0): the full LDS latency, seven times
// per 28-slot list and sample.
#define DSS_H_PAIR(C)                                                                            \
    if constexpr (2 * (C) < HC) {                                                                \
        if (2 * (C) < nh) {                                                                      \
            DSS_H_LOAD(HB, (2 * ((C) + 1) < HC ? (C) + 1 : 0))                                   \
            __builtin_amdgcn_sched_barrier(0);                                                   \
            DSS_H_MAC(HA)                                                                        \
            __builtin_amdgcn_sched_barrier(0);                                                   \
            DSS_H_LOAD(HA, (2 * ((C) + 2) < HC ? (C) + 2 : 0))                                   \
            __builtin_amdgcn_sched_barrier(0);                                                   \
            if (2 * ((C) + 1) < nh) {                                                            \
                DSS_H_MAC(HB)                                                                    \
            }                                                                                    \
            __builtin_amdgcn_sched_barrier(0);                                                   \
        }                                                                                        \
    }
#define DSS_H_CHAIN(XBUF)                                                                        \
    {                                                                                            \
        const char *xbase = reinterpret_cast<const char *>(XBUF);                                \
        f32x4 HA[4], HB[4];                                                                      \
        float ah = rbh + dgh * (XBUF)[uh];                                                       \
        DSS_H_LOAD(HA, 0)                                                                        \
        DSS_H_PAIR(0) DSS_H_PAIR(2) DSS_H_PAIR(4) DSS_H_PAIR(6) DSS_H_PAIR(8) DSS_H_PAIR(10) DSS_H_PAIR(12) DSS_H_PAIR(14) \
        static_assert(HC <= 32, "add DSS_H_PAIR terms");                                         \
        DSS_H_TAIL                                                                               \
        DSS_H_STORE(ah);                                                                         \
    }
// N inputs (multiple of 16) of the GRU B chain of one row.  Weights come from this lane's registers, the new
// GRU A state from LDS (same address in every lane -> broadcast), fetched one group of 16 inputs ahead so the
// LDS latency hides behind the previous group's arithmetic.  Products two at a time on aligned register
// pairs (v_pk_mul_f32), sums strictly one at a time in input order.
#define DSS_GB_GROUP(AV, G)                                                                      \
    _Pragma("unroll") for (int u = 0; u < 4; ++u) {                                              \
        const f32x2 p0 = WB[2 * (4 * (G) + u)] * (AV)[u].lo;                                     \
        const f32x2 p1 = WB[2 * (4 * (G) + u) + 1] * (AV)[u].hi;                                 \
        acc += p0.x;                                                                             \
        acc += p0.y;                                                                             \
        acc += p1.x;                                                                             \
        acc += p1.y;                                                                             \
    }
#define DSS_GB_GROUP_OFF(AV, G, WOFF)                                                            \
    _Pragma("unroll") for (int u = 0; u < 4; ++u) {                                              \
        const f32x2 p0 = WB[(WOFF) + 2 * (4 * (G) + u)] * (AV)[u].lo;                            \
        const f32x2 p1 = WB[(WOFF) + 2 * (4 * (G) + u) + 1] * (AV)[u].hi;                        \
        acc += p0.x;                                                                             \
        acc += p0.y;                                                                             \
        acc += p1.x;                                                                             \
        acc += p1.y;                                                                             \
    }
#define DSS_GB_LOAD(AV, AN, G)                                                                   \
    _Pragma("unroll") for (int u = 0; u < 4; ++u)                                                \
        (AV)[u] = *reinterpret_cast<const f32x4 *>((AN) + 16 * (G) + 4 * u);
// One s_waitcnt per group of four reads instead of the compiler's one per read (gfx9 encoding, vmcnt/expcnt left at
// their maxima): a wave issues about one instruction per 5 cycles whatever its kind, waits included.
#define DSS_WAIT_LGKM(N) __builtin_amdgcn_s_waitcnt(0xC07F | ((N) << 8))
#define DSS_GB_CHAIN(AN, N)                                                                      \
    {                                                                                            \
        f32x4 avA[4], avB[4];                                                                    \
        DSS_GB_LOAD(avA, AN, 0)                                                                  \
        DSS_GB_CHAIN_RUN(AN, N)                                                                  \
    }
// the same with the first group's reads already issued by the caller (avA, avB declared there)
#define DSS_GB_CHAIN_RUN(AN, N)                                                                  \
    {                                                                                            \
        _Pragma("unroll") for (int g = 0; g < (N) / 16; g += 2) {                                \
            if (g + 1 < (N) / 16) DSS_GB_LOAD(avB, AN, g + 1)                                    \
            __builtin_amdgcn_sched_barrier(0);   /* keep the prefetch ahead of the arithmetic */ \
            if (g + 1 < (N) / 16) DSS_WAIT_LGKM(4); else DSS_WAIT_LGKM(0);                       \
            __builtin_amdgcn_sched_barrier(0);                                                   \
            DSS_GB_GROUP(avA, g)                                                                 \
            __builtin_amdgcn_sched_barrier(0);                                                   \
            if (g + 2 < (N) / 16) DSS_GB_LOAD(avA, AN, g + 2)                                    \
            __builtin_amdgcn_sched_barrier(0);                                                   \
            if (g + 1 < (N) / 16) {                                                              \
                if (g + 2 < (N) / 16) DSS_WAIT_LGKM(4); else DSS_WAIT_LGKM(0);                   \
                __builtin_amdgcn_sched_barrier(0);                                               \
                DSS_GB_GROUP(avB, g + 1)                                                         \
            }                                                                                    \
            __builtin_amdgcn_sched_barrier(0);                                                   \
        }                                                                                        \
    }

// ... and with the weights taken from WB[WOFF + ...] (a chain that starts in the middle of the lane's weights)
#define DSS_GB_CHAIN_RUN_OFF(AN, N, WOFF)                                                                  \
    {                                                                                            \
        _Pragma("unroll") for (int g = 0; g < (N) / 16; g += 2) {                                \
            if (g + 1 < (N) / 16) DSS_GB_LOAD(avB, AN, g + 1)                                    \
            __builtin_amdgcn_sched_barrier(0);   /* keep the prefetch ahead of the arithmetic */ \
            if (g + 1 < (N) / 16) DSS_WAIT_LGKM(4); else DSS_WAIT_LGKM(0);                       \
            __builtin_amdgcn_sched_barrier(0);                                                   \
            DSS_GB_GROUP_OFF(avA, g, WOFF)                                                                 \
            __builtin_amdgcn_sched_barrier(0);                                                   \
            if (g + 2 < (N) / 16) DSS_GB_LOAD(avA, AN, g + 2)                                    \
            __builtin_amdgcn_sched_barrier(0);                                                   \
            if (g + 1 < (N) / 16) {                                                              \
                if (g + 2 < (N) / 16) DSS_WAIT_LGKM(4); else DSS_WAIT_LGKM(0);                   \
                __builtin_amdgcn_sched_barrier(0);                                               \
                DSS_GB_GROUP_OFF(avB, g + 1, WOFF)                                                         \
            }                                                                                    \
            __builtin_amdgcn_sched_barrier(0);                                                   \
        }                                                                                        \
    }

// Tree walk over the 256 decision bits the dual-FC waves left in BITS (8 words): every operand is wave-uniform, so this is scalar
// bit arithmetic.  8 levels, 3 scalar instructions each: test the node's bit (s_bitcmp1_b64 -> SCC), val = 2*val + SCC
// (s_addc_u32), next node index.  Nodes 1..63 live in m0, 64..127 in m1, 128..191 in m2, 192..255 in m3.
#define DSS_TREE_WALK_AT(VAL, BITS)                                                                       \
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

// a pair buffer of one 8x4 block: the block's four inputs for both utterances (two ds_read_b128)
struct PairX { f32x4 a, c; };          // a = inputs 0, 1; c = inputs 2, 3; each as (A, B)
__device__ __forceinline__ void dss_pair_loadx(PairX &q, const char *p)
{
    q.a = *reinterpret_cast<const f32x4 *>(p);
    q.c = *reinterpret_cast<const f32x4 *>(p + 16);
}

// ---- packed fp32 with a broadcast weight --------------------------------------------------------------------------
// (w.x * x.x, w.x * x.y) and (w.y * x.x, w.y * x.y): src0 low half (high half) to both lanes of the packed multiply
__device__ __forceinline__ f32x2 dss_pk_mul_lo(f32x2 w, f32x2 x)
{
    f32x2 r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]" : "=v"(r) : "v"(w), "v"(x));
    return r;
}
__device__ __forceinline__ f32x2 dss_pk_mul_hi(f32x2 w, f32x2 x)
{
    f32x2 r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[1,1]" : "=v"(r) : "v"(w), "v"(x));
    return r;
}
// One 8x4 block (or four consecutive dense inputs) of ONE dependent chain, software-pipelined: the four sums of this
// block's products P (formed one step earlier) alternate with the four products Q of the NEXT block, so that every
// dependent v_pk_add_f32 (8.9 cycles of latency) has an independent multiplication behind it.  W = (w0, w1) (w2, w3);
// X0..X3 = the (A, B) pairs of the next block's inputs 0..3.  Products go to registers of their own (the inputs are
// halves of 128-bit LDS reads: modified in place, half of them were copied first).
#define DSS_PK_STEP4(ACC, P, Q, X, WLO, WHI)                                                     \
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
// the first block's products (nothing to add yet) and the last block's sums (nothing left to multiply)
#define DSS_PK_MUL4(Q, X, WLO, WHI)                                                              \
    asm("v_pk_mul_f32 %[q0], %[wl], %[x0] op_sel_hi:[0,1]\n\t"                                   \
        "v_pk_mul_f32 %[q1], %[wl], %[x1] op_sel:[1,0] op_sel_hi:[1,1]\n\t"                      \
        "v_pk_mul_f32 %[q2], %[wh], %[x2] op_sel_hi:[0,1]\n\t"                                   \
        "v_pk_mul_f32 %[q3], %[wh], %[x3] op_sel:[1,0] op_sel_hi:[1,1]"                          \
        : [q0] "=&v"((Q)[0]), [q1] "=&v"((Q)[1]), [q2] "=&v"((Q)[2]), [q3] "=&v"((Q)[3])         \
        : [x0] "v"((X).a.lo), [x1] "v"((X).a.hi), [x2] "v"((X).c.lo), [x3] "v"((X).c.hi), [wl] "v"(WLO), [wh] "v"(WHI))
#define DSS_PK_ADD4(ACC, P)                                                                      \
    asm("v_pk_add_f32 %[a], %[a], %[p0]\n\t"                                                     \
        "v_pk_add_f32 %[a], %[a], %[p1]\n\t"                                                     \
        "v_pk_add_f32 %[a], %[a], %[p2]\n\t"                                                     \
        "v_pk_add_f32 %[a], %[a], %[p3]"                                                         \
        : [a] "+v"(ACC)                                                                          \
        : [p0] "v"((P)[0]), [p1] "v"((P)[1]), [p2] "v"((P)[2]), [p3] "v"((P)[3]))
