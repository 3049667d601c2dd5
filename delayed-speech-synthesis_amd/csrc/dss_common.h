// csrc/dss_common.h -- shared host/device declarations of libdss_hip.so (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <string>

#include "../../include/dss_hip.h"
#include "../../include/dss_lpcnet_blob.h"

// ---- fixed architecture constants of the path (checked against the blob at load) ----
#define DSS_NB_FEATURES 20
#define DSS_NB_BANDS 18
#define DSS_LPC_ORDER 16
#define DSS_FRAME_SIZE 160
#define DSS_GRU_A 384
#define DSS_GRU_B 16
#define DSS_FC_OUT 256
#define DSS_COND_STRIDE (3 * DSS_GRU_A + 3 * DSS_GRU_B + DSS_LPC_ORDER)   // 1216 floats per frame
#define DSS_FEATURES_DELAY 2
// Capacities of the CU-resident sample-rate kernel (lpcnet_sample.hip).  Row groups with more z/r blocks than their
// wave has register slots keep the surplus ("tail", in idx order behind the register slots) as LDS records next to the
// h-gate image, and h lists longer than DSS_HCX take the column ids of the further slots from LDS: slower per extra
// block, same results.  A model that exceeds the outer limits (DSS_ZR_TAIL, DSS_HX, or whose LDS image does not fit
// DSS_HBLK_BYTES) runs on the generic kernel instead.
#define DSS_ZRC 12            // register slots per lane for the z-gate and for the r-gate 8x4 blocks
#define DSS_HC 28             // h-gate slots per row group whose column ids sit in VGPRs (7)
#define DSS_ZR_TAIL 16        // max z (and r) blocks per row group beyond its wave's register slots
#define DSS_HCX 32            // ... in the instantiation with the extended paths (8 VGPRs)
#define DSS_HX 32             // max h-gate blocks per row group beyond DSS_HCX (column ids from LDS)
#define DSS_HBLK_BYTES 151552  // dynamic LDS left after the kernel's static 11.9 KB (160 KB per CU)

void dss_set_error(const char *fmt, ...);

#define DSS_HIP_CHECK(expr)                                                                        \
    do {                                                                                           \
        hipError_t _e = (expr);                                                                    \
        if (_e != hipSuccess) {                                                                    \
            dss_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            return DSS_ENODEV;                                                                     \
        }                                                                                          \
    } while (0)

// ---- device-resident model (built once per loaded blob, per device) ---------------------------------
struct DssSparseGate {
    int slots;            // padded number of 8x4 blocks per unit for this gate
    const int *pos4;      // [slots][384]  byte offset (pos * 4) of the block's first input in the state vector
    const float *w;       // [slots][4][384]
};

struct DssModelDev {
    dss_blob_header h;
    // frame-rate network (input-major dense matrices, as in the blob)
    const float *embed_pitch, *conv1_w, *conv1_b, *conv2_w, *conv2_b, *dense1_w, *dense1_b, *dense2_w, *dense2_b;
    const float *gru_a_dense_w, *gru_a_dense_b, *gru_b_dense_w, *gru_b_dense_b;
    // sample-rate network
    const float *embed_sig, *embed_pred, *embed_exc;        // [256][1152]
    const float *embed_lane[3];   // the same three tables as [256][384 lanes][3 gates] in the fast kernel's lane order: one
                                  // 12-byte load per lane and table, 768 contiguous bytes per wave
    const float *gru_a_rbias, *gru_a_diag;                  // [1152]
    DssSparseGate gate[3];
    const float *gru_b_bias;                                // [2][48]
    const float *gru_b_w_in;                                // [384][48]
    const float *gru_b_w_rec;                               // [16][48]
    const float *fc_bias, *fc_w, *fc_factor;                // [512], [256][2][16], [512]
    // derived tables (host libm, see dss_capi.cpp)
    const float *tansig;          // [201]
    const float *logit_table;     // [256]
    const float *ulaw2lin;        // [256]
    const float *dct_table;       // [18*18]
    const float *cos_table;       // [320]
    const float *cos_kl;          // [160][17] cos_table[(bin * lag) mod 320]: the inverse-DFT factor of every (bin, lag)
    const float *interp_a, *interp_b;   // [160] (1-frac), frac of interp_band_gain
    const int *interp_band;       // [160] band index i of each bin
    const double *lag_window;     // [17] 1 - 6e-5*i*i
    // register-resident layout of the sample-rate kernel (lpcnet_sample.hip)
    int fast_ok;                  // 1 when the model fits the capacities above
    int nzr_max;                  // max(z blocks, r blocks) over all row groups, rounded up to even (reporting)
    int zr_cap;                   // register slots per gate of waves 4, 5: 10 or 12 (selects the instantiation)
    int hmax;                     // max h blocks over all row groups (reporting)
    int ext;                      // 1 when the model uses z/r tails or h slots beyond DSS_HC (latency kernel only)
    int ext_tab;                  // float offset inside hblk of the tables the extended paths read (see dss_capi.cpp)
    int hblk_floats;              // size of hblk
    const int *unit_of;           // [384] lane of waves 0..5 -> GRU A unit whose z and r chains it runs
    const int *unit_h;            // [384] lane of waves 0..5 -> GRU A unit whose h-gate chain it runs
    const int *wave_nh;           // [8]   per wave: h-gate slots (even), [6],[7] unused
    const int *grp_hoff;          // [48]  per (wave, lane / 8): float offset of that row group's block records inside hblk
    const int *wave_nzr;          // [8]   per wave: z/r register slots actually used (even)
    const int *wave_nzt;          // [8]   per wave: z/r tail slots (LDS records), max over its groups and both gates
    const float *zr_w;            // [2*DSS_ZRC][4][384] z then r block weights per lane slot, zero padded
    const unsigned *zr_col;       // [2*DSS_ZRC/4][384]  four 8-bit block column ids (pos/4) per word
    const unsigned *h_col;        // [DSS_HCX/4][384]    same for the h-gate slots of the lane's row group
    const float *hblk;            // LDS image: one list of [8 rows][4] records per row group, lists back to back (see dss_capi.cpp)
    const float *gb_w_lane;       // [384][64] GRU B input weights, input-major, lane = row (rows 48..63 zero)
    const float *fc_w_pair;       // dual-FC weights as [k 8][node 256][4]: (layer 0, layer 1) weights of inputs 2k, 2k+1 (pair kernel)
    const float *gb_w_quad;       // the same weights as [384/4][64 lanes][4 inputs]: one 16-byte load per lane and block of four inputs
};

// ---- per-batch device state ---------------------------------------------------------------------------
struct DssBatchDev {
    int max_utts, max_frames;
    // decoder state, one row per utterance
    float *gru_a_state;   // [B][384]
    float *gru_b_state;   // [B][16]
    float *last_sig;      // [B][16]
    int *last_exc;        // [B]
    float *deemph;        // [B]
    uint32_t *rng;        // [B][4]
    int *frame_count;     // [B]
    float *conv1_mem;     // [B][2][84]
    float *conv2_mem;     // [B][2][128]
    float *old_lpc;       // [B][2][16]   row 0 = older (old_lpc[1]), row 1 = newer (old_lpc[0])
    // per-call scratch of the frame-rate network
    float *in_buf;        // [B][F+2][84]
    float *c1_buf;        // [B][F+2][128]
    float *c2_buf;        // [B][F][128]
    float *d1_buf;        // [B][F][128]
    float *cond_buf;      // [B][F][128]
    float *lpc_buf;       // [B][F+2][16]
    float *frame_out;     // [B][F][DSS_COND_STRIDE]  gru_a_condition | gru_b_condition | lpc
    int *fc0;             // [B] frame_count at the start of the call
    // ragged / slot-indexed calls (NULL = row i continues slot i, every row has n_frames frames)
    const int *slot_of;   // [n] decoder slot continued by row i of the call
    const int *count_of;  // [n] frames of row i (<= n_frames of the call; 0 leaves the slot untouched)
    const int *row_of;    // [n] ragged calls with counts: row handled by the k-th workgroup (slot), longest rows first
    int utt0;             // first row of this launch of the one-utterance sample kernel (a call split over two launches)
    // trace (optional)
    float *trace_exc, *trace_pcm;   // [B][F*160]
    // teacher forcing (tests; honoured by the TRACE instantiations only): sample k of row u takes the excitation
    // force_exc[u*F*160 + k] instead of the sampled one, and the pre-threshold logits of all 255 tree nodes go to
    // trace_logits[(u*F*160 + k)*256 + node]
    const unsigned char *force_exc;
    float *trace_logits;
};

// kernels (defined in the .hip files)
int dss_launch_frame_network(const DssModelDev &m, DssBatchDev &b, const float *d_features, int n_utts, int n_frames,
                             int feat_stride, hipStream_t s);
// pair: 0 = choose (two utterances per workgroup when the call has more utterances than the chip has CUs), -1 = never,
// 2 = always (uniform calls of models that fit, see dss_pair_fits)
int dss_launch_sample_network(const DssModelDev &m, DssBatchDev &b, int n_utts, int n_frames, short *d_pcm,
                              int trace, int pair, hipStream_t s);
int dss_launch_sample_network_pair(const DssModelDev &m, DssBatchDev &b, int n_utts, int n_frames, short *d_pcm, int trace,
                                   hipStream_t s);
int dss_pair_fits(const DssModelDev &m);
int dss_launch_sample_network_generic(const DssModelDev &m, DssBatchDev &b, int n_utts, int n_frames, short *d_pcm,
                                      int trace, hipStream_t s);
int dss_launch_exp10_selftest(const float *d_x, const float *d_comp, float *d_out, long n, hipStream_t s);
int dss_launch_lin2ulaw_selftest(unsigned start, unsigned stride, long n, unsigned char *d_out, hipStream_t s);
int dss_launch_lpcnet_reset(const DssModelDev &m, DssBatchDev &b, int utt, hipStream_t s);

// ---- speech-segment gate (speech_gate.hip) ---------------------------------------------------------------
#define DSS_GATE_STATE_INTS 8     // sm_write, sm_read, hist_write, speech_count, future_count, mask_lo, mask_hi, frames_seen
struct DssGateDev {
    int S, C;
    int sm_ctx, sm_size;          // smoothing context frames, window = 2*ctx+1 (<= 64)
    int hist_size, hist_ctx;      // segment ring length, context frames kept on both sides of a speech run
    int max_events;               // segments one stream can complete in one push
    double threshold;             // proportion of speech labels in the window that makes a frame speech
    float *sm_buf;                // [S][sm_size][C]
    float *hist;                  // [S][hist_size][C]
    float *seg_out;               // [S][max_events][hist_size][C] completed segments of the last push
    int *state;                   // [S][DSS_GATE_STATE_INTS]
    int *events;                  // [S][2 + max_events]: n_events, n_speech_labels, lengths
};
int dss_launch_gate(const DssGateDev &g, const double *d_frames, const int *d_labels, int W, hipStream_t s);
#define DSS_GATE_COLLECT_MAX 64
struct DssGateCollect { int stream[DSS_GATE_COLLECT_MAX], event[DSS_GATE_COLLECT_MAX], dst_row[DSS_GATE_COLLECT_MAX]; };
int dss_launch_gate_collect(const DssGateDev &g, const DssGateCollect &a, int n, float *d_dst, long dst_row_floats, hipStream_t s);
int dss_launch_gate_reset(const DssGateDev &g, int stream, hipStream_t s);

// ---- neural voice-activity detector (vad_lstm.hip) -----------------------------------------------------------
#define DSS_VAD_MAXH 160          // capacities of the kernels, checked when a handle is created
#define DSS_VAD_MAXC 128
#define DSS_DEC_MAXH 128
#define DSS_DEC_MAXC 256
#define DSS_DEC_MAXO 32
struct DssVadDev {
    int S, C, H;                  // streams, inputs per frame, hidden units (two LSTM layers, two classes)
    const float *wT0;             // [(Cp + Hp) / 4][4H][4]: weight_ih_l0 then weight_hh_l0, four consecutive inputs of a row side by
                                  //   side, input counts padded to multiples of 4 (gate order i, f, g, o)
    const float *b0;              // [4H]  bias_ih_l0 + bias_hh_l0
    const float *wT1;             // [2 Hp / 4][4H][4]: weight_ih_l1 then weight_hh_l1
    const float *b1;              // [4H]
    const float *wc, *bc;         // classifier [2][H], [2]
    float *h, *c;                 // [2 layers][S][H] each
};
int dss_launch_vad(const DssVadDev &v, const void *d_frames, int frames_f64, int W, int *d_labels, float *d_logits, hipStream_t s);

// ---- bidirectional recurrent decoder (bilstm_decoder.hip) ----------------------------------------------------
struct DssDecDev {
    int S_max, T_max, C, H, O;    // capacity (streams x frames per call), inputs per frame, hidden units per direction, outputs (20)
    const float *wT[2][2];        // [layer][direction]: [(Cin_p + Hp) / 4][4H][4]: weight_ih then weight_hh, four consecutive inputs of
                                  //   a row side by side, input counts padded to multiples of 4 (Cin = C, then 2H; gate order i, f, g, o)
    const float *b[2][2];         // [4H]  bias_ih + bias_hh
    const float *wr, *br;         // regressor [O][2H], [O]
    float *mid, *top;             // [S_max][T_max][2H] each: the outputs of layer 0 / layer 1 (forward | backward)
};
// d_counts (frames per stream, <= T), d_in_row (input row of stream s in a buffer of Tin frames per row): device arrays or NULL
int dss_launch_decoder(const DssDecDev &d, const void *d_frames, int frames_f64, int S, int T, float *d_feats,
                       const int *d_counts, const int *d_in_row, int Tin, hipStream_t s);

struct DssHgaDev {
    int S, C, fs, nsec;
    float wl, ws;
    int frame_length, overlap, cap_rows;
    double *zi;        // [S][2][8][2][C]
    double *rows;      // [S][cap_rows][C]
    double sos[2][8][6];
    const double *zs_mean, *zs_std;   // optional z-score of the frames, [C] each (device-resident entry points)
    int force_path;    // tests / A-B timing only: 0 choose, 1 hga_fused_kernel, 2 the three-launch form
};
// the front end of a call (local/common.py:16-58,308-345): raw amplifier rows (S, n, c_raw) instead of (S, n, C)
struct DssHgaFrontDev {
    const double *raw;
    int c_raw, n_grids;
    const int *src_col, *grid_of, *comp_cols, *comp_off;
};
// d_data is (S, n, C); fe must be nullptr (raw amplifier rows go through dss_launch_hga_frontend first)
int dss_launch_hga(const DssHgaDev &h, const double *d_data, const DssHgaFrontDev *fe, int n, int row0, int zero_rows, int rows,
                   int W, double *d_out, int apply_log, hipStream_t s);
int dss_launch_hga_frontend(const double *d_raw, double *d_pre, int S, int n, int c_raw, int C, const int *src_col,
                            const int *grid_of, int n_grids, const int *comp_cols, const int *comp_off, hipStream_t s);
int dss_launch_hga_wire(const float *d_payload, double *d_rows, int S, int C, int n, hipStream_t s);
int dss_launch_hga_reset(const DssHgaDev &h, const double *d_zi_hg, const double *d_zi_fh, hipStream_t s);
int dss_launch_log_power(const double *d_data, int T, int C, int sr, float wl, float ws, int W, double *d_out,
                         int apply_log, hipStream_t s);
