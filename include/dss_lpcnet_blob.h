/*
 * include/dss_lpcnet_blob.h -- on-disk / in-memory layout of an LPCNet weight blob.
 *
 * The reference links xiph/LPCNet's generated `src/nnet_data.c` (listed at
 * extensions/lpcnet/setup.py:34-36; the file itself is fetched from the network by xiph's
 * autogen.sh and is absent from /root/reference).  This build keeps the same tensors in one
 * flat little-endian blob so that weights are data, not compiled code: a blob converted from a
 * real nnet_data.c drops in without rebuilding the library.
 *
 * Layout: dss_blob_header, then the sections below, tightly packed, in this order.
 * All sections are float32 except gru_a_idx (int32).  Dense matrices are INPUT-major
 * ([n_in][n_out], the layout xiph's sgemv_accum reads: weights[j*stride + i]).
 *
 *   embed_pitch      [pitch_max][embed_pitch_dim]
 *   conv1_w          [3*(nb_features+embed_pitch_dim)][conv1_out]     conv1_b [conv1_out]
 *   conv2_w          [3*conv1_out][conv2_out]                         conv2_b [conv2_out]
 *   dense1_w         [conv2_out][dense1_out]                          dense1_b[dense1_out]
 *   dense2_w         [dense1_out][dense2_out]                         dense2_b[dense2_out]
 *   gru_a_dense_w    [dense2_out][3*gru_a]                            gru_a_dense_b[3*gru_a]
 *   gru_b_dense_w    [dense2_out][3*gru_b]                            gru_b_dense_b[3*gru_b]
 *   embed_sig        [256][3*gru_a]
 *   embed_pred       [256][3*gru_a]
 *   embed_exc        [256][3*gru_a]
 *   gru_a_rbias      [3*gru_a]         recurrent bias (gru->bias[3N..6N) in xiph's layout)
 *   gru_a_diag       [3*gru_a]         diagonal of the recurrent matrix, per gate
 *   gru_a_idx        int32[sparse_idx_len]   for each group of 8 output rows (3*gru_a/8 groups):
 *                                            count, then `count` input positions (multiples of 4)
 *   gru_a_w          [sparse_nblocks][4][8]  one 8x4 block per idx entry, input-major inside
 *   gru_b_bias       [2][3*gru_b]      input bias, recurrent bias
 *   gru_b_w_in       [gru_a][3*gru_b]
 *   gru_b_w_rec      [gru_b][3*gru_b]
 *   dual_fc_bias     [2*dual_fc_out]
 *   dual_fc_w        [dual_fc_out][2][gru_b]     row i: channel-0 weights then channel-1 weights
 *   dual_fc_factor   [2*dual_fc_out]
 */
#ifndef DSS_LPCNET_BLOB_H
#define DSS_LPCNET_BLOB_H

#include <stdint.h>

#define DSS_BLOB_MAGIC "DSSLPCN1"

typedef struct dss_blob_header {
    char magic[8];
    int32_t version;          /* 1 */
    int32_t nb_features;      /* 20: 18 Bark cepstra + pitch period + pitch correlation */
    int32_t nb_bands;         /* 18 */
    int32_t embed_pitch_dim;  /* 64 */
    int32_t pitch_max;        /* 256 */
    int32_t conv1_out;        /* 128 */
    int32_t conv2_out;        /* 128 */
    int32_t dense1_out;       /* 128 */
    int32_t dense2_out;       /* 128 */
    int32_t gru_a;            /* 384 */
    int32_t gru_b;            /* 16 */
    int32_t dual_fc_out;      /* 256 */
    int32_t lpc_order;        /* 16 */
    int32_t sparse_nblocks;   /* number of 8x4 blocks in gru_a_w */
    int32_t sparse_idx_len;   /* number of int32 in gru_a_idx */
    int32_t gru_a_order;      /* association order of the z/r pre-activations of compute_sparse_gru:
                               *   DSS_GRUA_INPUT_FIRST (0)  ((bias + diag*state) + input) + blocks...   xiph nnet.c since the
                               *                             8x4-block / int8 rewrite (early 2021), the era of the reference's
                               *                             TU list (extensions/lpcnet/setup.py:34-36)
                               *   DSS_GRUA_RECUR_FIRST (1)  input + ((bias + diag*state) + blocks...)    xiph nnet.c 2019-2020
                               *                             (zrh = input; recur = bias + diag*state + sparse; zrh += recur)
                               * The h gate is the same in both (r*recur_h + input_h; a float add commutes).  Blobs written
                               * before this field existed have 0 here. */
    int32_t source_branches;  /* which sides of nnet_data.c's `#ifdef DOT_PROD` pairs the source had: bit 0 the `#else` side
                               * (float weights, generic vec.h -- what this blob holds and the kernels compute), bit 1 the
                               * `#ifdef` side (int8 qweight blocks, scales, subias: vec_avx.h / vec_neon.h builds; kept by
                               * the converter in <blob>.dotprod.npz, read by no kernel).  0 = unknown (synthetic models,
                               * blobs written before this field existed).  Informational: see DESIGN.md 2. */
    int32_t reserved[5];
} dss_blob_header;            /* 96 bytes */

#define DSS_BLOB_BRANCH_FLOAT 1
#define DSS_BLOB_BRANCH_DOT_PROD 2

#define DSS_GRUA_INPUT_FIRST 0
#define DSS_GRUA_RECUR_FIRST 1

#endif
