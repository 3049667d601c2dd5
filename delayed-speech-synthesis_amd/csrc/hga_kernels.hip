// csrc/hga_kernels.hip -- high-gamma feature extraction on gfx950 (float64, bit-exact operation order).
//
// Replaces, for many streams at once:
//   scipy.signal.sosfilt x2 with carried state      local/units.py:151-152
//   WarmStartFrameBuffer.insert                     extensions/hga/hga_optimized.pyx:96-131
//   compute_log_power_features                      extensions/hga/hga_optimized.pyx:27-47
//
// Mapping: one lane per (stream, channel).  The path is a chain of 16 dependent biquads per sample and a
// sequential 50-term sum per window, so time is the serial axis and channels x streams the parallel
// one; consecutive lanes are consecutive channels, which makes every global access a coalesced
// 8-byte-per-lane row segment of the (T, C) row-major layout the reference uses.
// Built with -ffp-contract=off: every product and sum rounds separately, as in scipy's C loop and in
// the reference's Cython kernel, which is what makes the mean power bit-identical.
#include "dss_common.h"

#define HGA_CHUNK 8

struct HgaSos { double k[2][8][6]; };

__device__ __forceinline__ int hga_win_start(int win, float ws, int sr)
{
    // pyx:43  int(round((win * window_shift) * sr)) -- float32 products, C round()
    return (int)round((double)((win * ws) * sr));
}
__device__ __forceinline__ int hga_win_stop(int start, float wl, int sr)
{
    // pyx:44  int(round(start_eeg + window_length * sr))
    return (int)round((double)(start + wl * sr));
}

// Filters n new rows of every (stream, channel) column, appends them to the column's row buffer at
// row0, emits W window features from rows [0, rows) and keeps the last `overlap` rows for the next call.
__global__ void __launch_bounds__(256)
hga_extract_kernel(const double *__restrict__ data, double *__restrict__ zi, double *__restrict__ rowbuf,
                   double *__restrict__ out, HgaSos sos, int S, int C, int n, int nsec, int row0, int rows, int W,
                   int cap_rows, int overlap, int sr, float wl, float ws, int apply_log, int zero_rows)
{
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= S * C) return;
    const int s = gid / C, c = gid - s * C;

    double z[2][8][2];
    double *zp = zi + (size_t)s * 2 * 8 * 2 * C + c;
#pragma unroll
    for (int f = 0; f < 2; ++f)
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            z[f][q][0] = zp[((f * 8 + q) * 2 + 0) * (size_t)C];
            z[f][q][1] = zp[((f * 8 + q) * 2 + 1) * (size_t)C];
        }

    const double *x = data + (size_t)s * n * C + c;
    double *col = rowbuf + (size_t)s * cap_rows * C + c;
    // CASE 2 of the frame buffer (first chunk shorter than a frame): left zero padding, pyx:116
    for (int r = 0; r < zero_rows; ++r) col[(size_t)r * C] = 0.0;

    for (int t0 = 0; t0 < n; t0 += HGA_CHUNK) {
        double v[HGA_CHUNK];
#pragma unroll
        for (int u = 0; u < HGA_CHUNK; ++u) v[u] = (t0 + u < n) ? x[(size_t)(t0 + u) * C] : 0.0;
#pragma unroll
        for (int u = 0; u < HGA_CHUNK; ++u) {
            if (t0 + u < n) {
                double cur = v[u];
#pragma unroll
                for (int f = 0; f < 2; ++f)
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        if (q < nsec) {
                            const double b0 = sos.k[f][q][0], b1 = sos.k[f][q][1], b2 = sos.k[f][q][2];
                            const double a1 = sos.k[f][q][4], a2 = sos.k[f][q][5];
                            const double y = b0 * cur + z[f][q][0];
                            z[f][q][0] = b1 * cur - a1 * y + z[f][q][1];
                            z[f][q][1] = b2 * cur - a2 * y;
                            cur = y;
                        }
                    }
                col[(size_t)(row0 + t0 + u) * C] = cur;
            }
        }
    }
#pragma unroll
    for (int f = 0; f < 2; ++f)
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            zp[((f * 8 + q) * 2 + 0) * (size_t)C] = z[f][q][0];
            zp[((f * 8 + q) * 2 + 1) * (size_t)C] = z[f][q][1];
        }

    // windowed mean power over this lane's own column (only this lane ever wrote it)
    double *o = out + (size_t)s * W * C + c;
    for (int win = 0; win < W; ++win) {
        const int start = hga_win_start(win, ws, sr);
        const int stop = hga_win_stop(start, wl, sr);
        double sum = 0.0;
        for (int r = start; r < stop; ++r) {
            const double q = col[(size_t)r * C];
            sum += q * q;
        }
        double p = sum / (double)(stop - start) + 0.01;
        o[(size_t)win * C] = apply_log ? log(p) : p;
    }
    // keep the last `overlap` rows (ascending copy: source row is always ahead of destination row)
    for (int k = 0; k < overlap; ++k) col[(size_t)k * C] = col[(size_t)(rows - overlap + k) * C];
}

int dss_launch_hga(const DssHgaDev &h, const double *d_data, int n, int row0, int zero_rows, int rows, int W,
                   double *d_out, int apply_log, hipStream_t st)
{
    HgaSos sos;
    memcpy(&sos, h.sos, sizeof(sos));
    const int total = h.S * h.C;
    const int block = 64;                      // small blocks: few lanes, long serial chains -> spread over CUs
    const int grid = (total + block - 1) / block;
    hipLaunchKernelGGL(hga_extract_kernel, dim3(grid), dim3(block), 0, st, d_data, h.zi, h.rows, d_out, sos, h.S, h.C, n,
                       h.nsec, row0, rows, W, h.cap_rows, h.overlap, h.fs, h.wl, h.ws, apply_log, zero_rows);
    DSS_HIP_CHECK(hipGetLastError());
    return DSS_OK;
}

// state <- sosfilt_zi tiled over channels (units.py:128-132); zero the overlap rows (pyx:79-82)
__global__ void hga_reset_kernel(double *zi, double *rowbuf, const double *zi_hg, const double *zi_fh, int S, int C,
                                 int nsec, int cap_rows, int overlap)
{
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= S * C) return;
    const int s = gid / C, c = gid - s * C;
    double *zp = zi + (size_t)s * 2 * 8 * 2 * C + c;
    for (int f = 0; f < 2; ++f)
        for (int q = 0; q < 8; ++q)
            for (int k = 0; k < 2; ++k) {
                const double *src = f ? zi_fh : zi_hg;
                zp[((f * 8 + q) * 2 + k) * (size_t)C] = (q < nsec) ? src[q * 2 + k] : 0.0;
            }
    double *col = rowbuf + (size_t)s * cap_rows * C + c;
    for (int r = 0; r < overlap; ++r) col[(size_t)r * C] = 0.0;
}

int dss_launch_hga_reset(const DssHgaDev &h, const double *d_zi_hg, const double *d_zi_fh, hipStream_t st)
{
    const int total = h.S * h.C;
    hipLaunchKernelGGL(hga_reset_kernel, dim3((total + 255) / 256), dim3(256), 0, st, h.zi, h.rows, d_zi_hg, d_zi_fh, h.S,
                       h.C, h.nsec, h.cap_rows, h.overlap);
    DSS_HIP_CHECK(hipGetLastError());
    return DSS_OK;
}

// Stateless compute_log_power_features: one lane per (window, channel); 50-term sequential sum.
__global__ void __launch_bounds__(256)
hga_log_power_kernel(const double *__restrict__ data, double *__restrict__ out, int T, int C, int W, int sr, float wl,
                     float ws, int apply_log)
{
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= W * C) return;
    const int win = gid / C, c = gid - win * C;
    const int start = hga_win_start(win, ws, sr);
    const int stop = hga_win_stop(start, wl, sr);
    double sum = 0.0;
    for (int r = start; r < stop; ++r) {
        const double q = data[(size_t)r * C + c];
        sum += q * q;
    }
    const double p = sum / (double)(stop - start) + 0.01;
    out[gid] = apply_log ? log(p) : p;
}

int dss_launch_log_power(const double *d_data, int T, int C, int sr, float wl, float ws, int W, double *d_out,
                         int apply_log, hipStream_t st)
{
    const int total = W * C;
    if (total <= 0) return DSS_OK;
    hipLaunchKernelGGL(hga_log_power_kernel, dim3((total + 255) / 256), dim3(256), 0, st, d_data, d_out, T, C, W, sr, wl,
                       ws, apply_log);
    DSS_HIP_CHECK(hipGetLastError());
    return DSS_OK;
}


// Fused front end: column reorder + per-grid common average reference + channel selection, one lane per
// (stream, sample).  The grid mean is a sequential sum in the reference's column order (numpy reduces the
// Fortran-ordered fancy-index view column by column), then one division: bit-identical to the numpy chain.
__global__ void __launch_bounds__(64)
hga_frontend_kernel(const double *__restrict__ raw, double *__restrict__ pre, int total, int c_raw, int C,
                    const int *__restrict__ src_col, const int *__restrict__ grid_of, int n_grids,
                    const int *__restrict__ comp_cols, const int *__restrict__ comp_off)
{
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= total) return;
    const double *row = raw + (size_t)gid * c_raw;
    double mean[4] = {0.0, 0.0, 0.0, 0.0};
    for (int g = 0; g < n_grids && g < 4; ++g) {
        double sum = 0.0;
        const int a = comp_off[g], b2 = comp_off[g + 1];
        for (int k = a; k < b2; ++k) sum += row[comp_cols[k]];
        mean[g] = sum / (double)(b2 - a);
    }
    double *o = pre + (size_t)gid * C;
    for (int c = 0; c < C; ++c) {
        const int g = grid_of[c];
        const double v = row[src_col[c]];
        o[c] = g >= 0 ? v - mean[g] : v;
    }
}

int dss_launch_hga_frontend(const double *d_raw, double *d_pre, int S, int n, int c_raw, int C, const int *src_col,
                            const int *grid_of, int n_grids, const int *comp_cols, const int *comp_off, hipStream_t st)
{
    const int total = S * n;
    hipLaunchKernelGGL(hga_frontend_kernel, dim3((total + 63) / 64), dim3(64), 0, st, d_raw, d_pre, total, c_raw, C, src_col,
                       grid_of, n_grids, comp_cols, comp_off);
    DSS_HIP_CHECK(hipGetLastError());
    return DSS_OK;
}
