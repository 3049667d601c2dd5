// csrc/hga_kernels.hip -- high-gamma feature extraction on gfx950 (float64, bit-exact operation order).
//
// Replaces, for many streams at once:
//   scipy.signal.sosfilt x2 with carried state      local/units.py:151-152
//   WarmStartFrameBuffer.insert                     extensions/hga/hga_optimized.pyx:96-131
//   compute_log_power_features                      extensions/hga/hga_optimized.pyx:27-47
//
// ONE launch per call (hga_fused_kernel): the IIR cascades with their sections spread over the 16 lanes of a DPP row (a
// column advances one sample per step instead of one per 16 dependent biquads); the filtered samples of a tile go into an
// LDS ring instead of HBM, every window that has become complete is summed from the ring (a 50-term sequential sum per
// (window, channel) lane, exactly array_sum_and_power's order), and the last `overlap` rows are left in the row buffer
// for the next call.  Only the input, the W x C frames and 40 rows per column touch HBM: at 1024 streams x 1.04 s that
// removes the 545 MB write + read of the filtered rows which the three-launch form (hga_filter_kernel +
// hga_window_kernel + hga_overlap_kernel, kept as the fallback for window shapes whose ring would not fit LDS) needed.
// hga_stream_kernel (further down) is the same computation with the phases on different waves and the front end inside
// the launch; it is exact and, as measured in round 3, slower, so it runs only on request.
// Built with -ffp-contract=off: every product and sum rounds separately, as in scipy's C loop and in
// the reference's Cython kernel, which is what makes the mean power bit-identical.
#include <stdlib.h>

#include "dss_common.h"

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

// ---- stage 1: the two IIR cascades, pipelined over their sections --------------------------------------------
// The 2*nsec (<= 16) second-order sections of one (stream, channel) column occupy the 16 lanes of one DPP row: at
// step k lane r filters sample k - r, taking its input from lane r-1's output of the previous step (row_shr:1), so
// the column advances one sample per step instead of one per 16 biquads.  Each section's arithmetic is exactly
// scipy's sosfilt inner loop (one product, one sum at a time).  A 256-thread block handles 16 consecutive columns
// (same stream, consecutive channels), whose input samples are staged through LDS in tiles of HGA_TT rows with
// 128-byte row segments; lane 2*nsec-1 writes the filtered sample to the column's row buffer at row0 + t.
#define HGA_TT 64
#define HGA_WTAB 256          // windows whose row ranges the fused kernel tabulates in LDS (2 KB)

// lanes 1..15 of every row receive their left neighbour's `y`; lane 0 (no neighbour) keeps `x`: the column's input
__device__ __forceinline__ double hga_shift_in(double x, double y)
{
    const unsigned long long ux = __builtin_bit_cast(unsigned long long, x), uy = __builtin_bit_cast(unsigned long long, y);
    const int lo = __builtin_amdgcn_update_dpp((int)(unsigned)ux, (int)(unsigned)uy, 0x111, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp((int)(unsigned)(ux >> 32), (int)(unsigned)(uy >> 32), 0x111, 0xf, 0xf, false);
    return __builtin_bit_cast(double, ((unsigned long long)(unsigned)hi << 32) | (unsigned)lo);
}

// one step of one section: scipy _sosfilt's inner statement order
#define HGA_BIQUAD(IN)                                                                           \
    {                                                                                            \
        y = b0 * (IN) + z0;                        /* x_c = b0*x_n + zi0        */               \
        z0 = b1 * (IN) - a1 * y + z1;              /* zi0 = b1*x_n - a1*x_c + zi1 */             \
        z1 = b2 * (IN) - a2 * y;                   /* zi1 = b2*x_n - a2*x_c     */               \
    }

__global__ void __launch_bounds__(256)
hga_filter_kernel(const double *__restrict__ data, double *__restrict__ zi, double *__restrict__ rowbuf,
                  HgaSos sos, int S, int C, int n, int nsec, int row0, int cap_rows, int zero_rows)
{
    __shared__ double xs[HGA_TT + 1][16];      // inputs of the tile's steps, one row segment of 16 columns per step (+1: prefetch)
    __shared__ double ys[HGA_TT][16];          // outputs produced at the tile's steps (sample k - (2*nsec-1) at step k)
    __shared__ double coef[16][5];
    __shared__ double dump[256];               // where the lanes that do not hold the last section "store" their y
    const int tid = threadIdx.x, r = tid & 15, pib = tid >> 4;
    const long total = (long)S * C;
    const long pair0 = (long)blockIdx.x * 16;
    long pair = pair0 + pib;
    const bool valid = pair < total;
    if (!valid) pair = total - 1;
    const int s = (int)(pair / C), c = (int)(pair - (long)s * C);
    const int nsec2 = 2 * nsec;
    const bool has_sec = r < nsec2;
    const bool is_last = r == nsec2 - 1;
    // branch-free output store of the steady-state loop: the last section's lane walks down its ys column, every other
    // lane rewrites its own dump word
    char *const ybase = is_last ? reinterpret_cast<char *>(&ys[0][pib]) : reinterpret_cast<char *>(&dump[tid]);
    const int ystride = is_last ? 16 * (int)sizeof(double) : 0;
    const int f = has_sec ? r / nsec : 0, q = has_sec ? r - f * nsec : 0;
    if (tid < 16) {           // section coefficients by lane (a kernel argument cannot be indexed per lane)
        const int ff = tid < nsec2 ? tid / nsec : 0, qq = tid < nsec2 ? tid - ff * nsec : 0;
        coef[tid][0] = sos.k[ff][qq][0]; coef[tid][1] = sos.k[ff][qq][1]; coef[tid][2] = sos.k[ff][qq][2];
        coef[tid][3] = sos.k[ff][qq][4]; coef[tid][4] = sos.k[ff][qq][5];
    }
    double *zp = zi + (size_t)s * 2 * 8 * 2 * C + c;
    double z0 = zp[((f * 8 + q) * 2 + 0) * (size_t)C], z1 = zp[((f * 8 + q) * 2 + 1) * (size_t)C];
    // CASE 2 of the frame buffer (first chunk shorter than a frame): left zero padding, pyx:116
    if (valid) {
        double *col = rowbuf + (size_t)s * cap_rows * C + c;
        for (int k = r; k < zero_rows; k += 16) col[(size_t)k * C] = 0.0;
    }
    __syncthreads();
    const double b0 = coef[r][0], b1 = coef[r][1], b2 = coef[r][2], a1 = coef[r][3], a2 = coef[r][4];
    const int steps = n + nsec2 - 1;
    double y = 0.0;
    for (int base = 0; base < steps; base += HGA_TT) {
        __syncthreads();                                   // previous tile fully consumed and written out
        for (int idx = tid; idx < HGA_TT * 16; idx += 256) {
            const int tt = idx >> 4, p = idx & 15;
            const long pp = pair0 + p;
            const int t = base + tt;
            double v = 0.0;
            if (t < n && pp < total) {
                const int sp = (int)(pp / C), cp = (int)(pp - (long)sp * C);
                v = data[((size_t)sp * n + t) * C + cp];
            }
            xs[tt][p] = v;
        }
        __syncthreads();
        const int kend = min(base + HGA_TT, steps);
        int k = base;
        // steps on which some lanes have no sample yet (pipeline filling) or no more (draining): predicated
        for (; k < kend && (k < nsec2 - 1 || k >= n); ++k) {
            const double in = hga_shift_in(xs[k - base][pib], y);
            const int t = k - r;
            if (has_sec && t >= 0 && t < n) {
                HGA_BIQUAD(in)
                if (is_last) ys[k - base][pib] = y;
            }
        }
        // steady state: every section has a sample
        const int ksteady = min(kend, n);
        {
            char *yp = ybase + (k - base) * ystride;
            const double *xp = &xs[k - base][pib];
            double xcur = *xp;
            for (; k < ksteady; ++k) {
                xp += 16;
                const double xnext = *xp;                  // next step's input: its LDS latency hides under this step
                const double in = hga_shift_in(xcur, y);
                HGA_BIQUAD(in)
                *reinterpret_cast<double *>(yp) = y;
                yp += ystride;
                xcur = xnext;
            }
        }
        for (; k < kend; ++k) {                            // draining steps at the end of the last tile(s)
            const double in = hga_shift_in(xs[k - base][pib], y);
            const int t = k - r;
            if (has_sec && t >= 0 && t < n) {
                HGA_BIQUAD(in)
                if (is_last) ys[k - base][pib] = y;
            }
        }
        __syncthreads();
        // the tile's finished samples go out as 128-byte row segments
        for (int idx = tid; idx < HGA_TT * 16; idx += 256) {
            const int tt = idx >> 4, p = idx & 15;
            const long pp = pair0 + p;
            const int t = base + tt - (nsec2 - 1);
            if (base + tt < kend && t >= 0 && t < n && pp < total) {
                const int sp = (int)(pp / C), cp = (int)(pp - (long)sp * C);
                rowbuf[((size_t)sp * cap_rows + row0 + t) * C + cp] = ys[tt][p];
            }
        }
    }
    if (valid && has_sec) {
        zp[((f * 8 + q) * 2 + 0) * (size_t)C] = z0;
        zp[((f * 8 + q) * 2 + 1) * (size_t)C] = z1;
    }
}

// ---- fused form: filter tile -> LDS ring -> completed windows -> overlap rows -------------------------------------------
// Same block shape and the same per-step code as hga_filter_kernel (16 columns x 16 section lanes).  `ring` holds the last
// R rows of the block's 16 columns (overlap / zero rows first, then the new samples at row0 + t), row i at position i mod R;
// R >= frame_length + window shift + TT + 2 so that no row a pending window still needs is overwritten by the next tile.
// Round 3: R carries no power-of-two padding and the lanes that do not hold the last section have no dummy store targets
// any more (18 KB of static LDS + 16 KB of ring instead of 37 KB).  A 32-step-tile instantiation (22 KB, 7 blocks per CU
// instead of 5) is kept for A/B timing: it is no faster at 1024 streams and slower on small calls (profiles/
// r3_hga_tile_ab_timing.txt), so 64-step tiles stay the default.
#define HGF_WTAB 128
template <int TT>
__global__ void __launch_bounds__(256, TT == 32 ? 7 : 4)
hga_fused_kernel(const double *__restrict__ data, double *__restrict__ zi, double *__restrict__ rowbuf, double *__restrict__ out,
                 HgaSos sos, int S, int C, int n, int nsec, int row0, int cap_rows, int zero_rows, int rows, int W, int overlap,
                 int sr, float wl, float ws, int apply_log, int R, const double *__restrict__ zs_mean,
                 const double *__restrict__ zs_std)
{
    __shared__ double xs[TT + 1][16];
    __shared__ double ys[TT][16];
    __shared__ double coef[16][5];
    __shared__ int wtab[2][HGF_WTAB];          // first and one-past-last row of the first HGF_WTAB windows (the float32 /
                                               //   round() arithmetic of pyx:43-44 once per block instead of once per tile)
    extern __shared__ __attribute__((aligned(16))) double ring[];         // [R][16]
    const int tid = threadIdx.x, r = tid & 15, pib = tid >> 4;
    const long total = (long)S * C;
    const long pair0 = (long)blockIdx.x * 16;
    long pair = pair0 + pib;
    const bool valid = pair < total;
    if (!valid) pair = total - 1;
    const int s = (int)(pair / C), c = (int)(pair - (long)s * C);
    const int nsec2 = 2 * nsec;
    const bool has_sec = r < nsec2;
    const bool is_last = r == nsec2 - 1;
    const int f = has_sec ? r / nsec : 0, q = has_sec ? r - f * nsec : 0;
    if (tid < 16) {
        const int ff = tid < nsec2 ? tid / nsec : 0, qq = tid < nsec2 ? tid - ff * nsec : 0;
        coef[tid][0] = sos.k[ff][qq][0]; coef[tid][1] = sos.k[ff][qq][1]; coef[tid][2] = sos.k[ff][qq][2];
        coef[tid][3] = sos.k[ff][qq][4]; coef[tid][4] = sos.k[ff][qq][5];
    }
    double *zp = zi + (size_t)s * 2 * 8 * 2 * C + c;
    double z0 = zp[((f * 8 + q) * 2 + 0) * (size_t)C], z1 = zp[((f * 8 + q) * 2 + 1) * (size_t)C];
    for (int w = tid; w < W && w < HGF_WTAB; w += 256) {
        const int st = hga_win_start(w, ws, sr);
        wtab[0][w] = st;
        wtab[1][w] = hga_win_stop(st, wl, sr);
    }
    // rows [0, row0) of the ring: zeros (CASE 2, pyx:116) or the overlap the previous call left (CASE 3, pyx:123-131)
    for (int idx = tid; idx < row0 * 16; idx += 256) {
        const int rr = idx >> 4, p = idx & 15;
        const long pp = pair0 + p;
        double v = 0.0;
        if (rr >= zero_rows && pp < total) {
            const int sp = (int)(pp / C), cp = (int)(pp - (long)sp * C);
            v = rowbuf[((size_t)sp * cap_rows + rr) * C + cp];
        }
        ring[(rr % R) * 16 + p] = v;
    }
    __syncthreads();
    const double b0 = coef[r][0], b1 = coef[r][1], b2 = coef[r][2], a1 = coef[r][3], a2 = coef[r][4];
    const int steps = n + nsec2 - 1;
    double y = 0.0;
    int w_next = 0;                                        // first window not yet written (block-uniform)
    // Input tiles: thread (tt0, p) owns column p of rows tt0, tt0 + 16, ... of every tile.  The next tile's samples are
    // fetched into registers before the current tile's steps run and go to LDS after them: the HBM latency hides under
    // the filter arithmetic instead of standing between two barriers.
    const int lp = tid & 15, ltt = tid >> 4;
    const long lpp = pair0 + lp;
    const bool lvalid = lpp < total;
    const double *lcol = data;
    if (lvalid) { const int sp = (int)(lpp / C), cp = (int)(lpp - (long)sp * C); lcol = data + (size_t)sp * n * C + cp; }
    double pre[TT / 16];
#define HGF_FETCH(BASE)                                                                          \
    _Pragma("unroll") for (int j = 0; j < TT / 16; ++j) {                                        \
        const int t = (BASE) + ltt + 16 * j;                                                     \
        pre[j] = (lvalid && t < n) ? lcol[(size_t)t * C] : 0.0;                                  \
    }
    HGF_FETCH(0)
    // ring position of row (row0 + t) for the first output of the current tile, kept modulo R (block-uniform)
    int tile_pos = (row0 + R * 16 - (nsec2 - 1)) % R;      // row0 + t for t = 0 - (nsec2 - 1): the row "before" the first output
    for (int base = 0; base < steps; base += TT) {
        __syncthreads();                                   // previous tile fully consumed
#pragma unroll
        for (int j = 0; j < TT / 16; ++j) xs[ltt + 16 * j][lp] = pre[j];
        if (base + TT < steps) { HGF_FETCH(base + TT) }
        __syncthreads();
        const int kend = min(base + TT, steps);
        int k = base;
        for (; k < kend && (k < nsec2 - 1 || k >= n); ++k) {
            const double in = hga_shift_in(xs[k - base][pib], y);
            const int t = k - r;
            if (has_sec && t >= 0 && t < n) {
                HGA_BIQUAD(in)
                if (is_last) ys[k - base][pib] = y;
            }
        }
        const int ksteady = min(kend, n);
        {
            const double *xp = &xs[k - base][pib];
            double xcur = *xp;
            // four steps per trip with constant LDS offsets (a DPP move is convergent: the compiler will not unroll a loop of
            // unknown length around it): per step 9 fp64 operations, 2 DPP moves, one LDS read, one LDS write
            double *const yl = &ys[k - base][pib];
            int kk = 0;
            for (; k + 4 <= ksteady; k += 4, kk += 4) {
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const double xnext = xp[16 * (kk + u + 1)];
                    const double in = hga_shift_in(xcur, y);
                    HGA_BIQUAD(in)
                    if (is_last) yl[16 * (kk + u)] = y;
                    xcur = xnext;
                }
            }
            for (; k < ksteady; ++k, ++kk) {
                const double xnext = xp[16 * (kk + 1)];
                const double in = hga_shift_in(xcur, y);
                HGA_BIQUAD(in)
                if (is_last) yl[16 * kk] = y;
                xcur = xnext;
            }
        }
        for (; k < kend; ++k) {
            const double in = hga_shift_in(xs[k - base][pib], y);
            const int t = k - r;
            if (has_sec && t >= 0 && t < n) {
                HGA_BIQUAD(in)
                if (is_last) ys[k - base][pib] = y;
            }
        }
        __syncthreads();
        // the tile's finished samples join the ring at their row (position row mod R, advanced from tile to tile)
        for (int idx = tid; idx < TT * 16; idx += 256) {
            const int tt = idx >> 4, p = idx & 15;
            const int t = base + tt - (nsec2 - 1);
            int pos = tile_pos + tt;
            if (pos >= R) pos -= R;
            if (base + tt < kend && t >= 0 && t < n) ring[pos * 16 + p] = ys[tt][p];
        }
        tile_pos += TT;
        if (tile_pos >= R) tile_pos -= R;
        __syncthreads();
        // every window that is complete now: lane = (window, column), a sequential sum over its rows (pyx:9-22, 42-46)
        int t_done = kend - (nsec2 - 1);
        t_done = t_done < 0 ? 0 : (t_done > n ? n : t_done);
        const int rows_done = row0 + t_done;
        {
            const int p = tid & 15, wi = tid >> 4;
            const long pp = pair0 + p;
            for (int w = w_next + wi; w < W; w += 16) {
                const int start = w < HGF_WTAB ? wtab[0][w] : hga_win_start(w, ws, sr);
                const int stop = w < HGF_WTAB ? wtab[1][w] : hga_win_stop(start, wl, sr);
                if (stop > rows_done) break;
                double sum = 0.0;
                // the window's rows sit at positions start mod R ... (wrapping once at most: stop - start <= R)
                int pos = start % R, left = stop - start;
                while (left > 0) {
                    const int run = min(left, R - pos);
                    const double *rp = ring + pos * 16 + p;
                    int i = 0;
                    for (; i + 8 <= run; i += 8) {             // eight ring reads in flight, then their terms in row order
                        double v[8];
#pragma unroll
                        for (int u = 0; u < 8; ++u) v[u] = rp[(i + u) * 16];
#pragma unroll
                        for (int u = 0; u < 8; ++u) sum += v[u] * v[u];
                    }
                    for (; i < run; ++i) {
                        const double v = rp[i * 16];
                        sum += v * v;
                    }
                    left -= run;
                    pos = 0;
                }
                if (pp < total) {
                    const int sp = (int)(pp / C), cp = (int)(pp - (long)sp * C);
                    const double pw = sum / (double)(stop - start) + 0.01;
                    double o = apply_log ? log(pw) : pw;
                    if (zs_mean) o = (o - zs_mean[cp]) / zs_std[cp];          // ZScoreNormalization (local/common.py:367-376)
                    out[((size_t)sp * W + w) * C + cp] = o;
                }
            }
            while (w_next < W && (w_next < HGF_WTAB ? wtab[1][w_next] : hga_win_stop(hga_win_start(w_next, ws, sr), wl, sr)) <= rows_done)
                ++w_next;
        }
    }
    __syncthreads();
    // the last `overlap` rows become rows 0..overlap-1 of the next call (WarmStartFrameBuffer.remainder_data)
    for (int idx = tid; idx < overlap * 16; idx += 256) {
        const int kk = idx >> 4, p = idx & 15;
        const long pp = pair0 + p;
        if (pp < total) {
            const int sp = (int)(pp / C), cp = (int)(pp - (long)sp * C);
            rowbuf[((size_t)sp * cap_rows + kk) * C + cp] = ring[((rows - overlap + kk) % R) * 16 + p];
        }
    }
    if (valid && has_sec) {
        zp[((f * 8 + q) * 2 + 0) * (size_t)C] = z0;
        zp[((f * 8 + q) * 2 + 1) * (size_t)C] = z1;
    }
#undef HGF_FETCH
}

// ---- streamed form (opt-in): four filter waves that only filter, two helper waves that feed and drain them
// hga_fused_kernel alternates phases inside every wave (load a tile, filter it, copy it to the ring, sum the windows it
// completed), so the fp64 pipes idle through three of the four.  Here a 384-thread block gives the 16 columns' filter steps
// to waves 0..3 and everything else to two helper waves:
//   wave 4 (stager)   the NEXT tile's inputs into the other half of a double-buffered LDS tile -- either 128-byte row
//                     segments of a (S, n, C) array, or, with the front end, the raw amplifier rows themselves: the
//                     tile's raw rows come in with 16-byte loads, one lane per (row, grid) runs the grid mean as the
//                     reference's sequential sum (local/common.py:308-345), and the block's 16 channels are selected,
//                     referenced and written to the tile.  The four blocks of a stream sit on one XCD (block index
//                     mapping below), so the raw rows reach HBM once and the other three reads are L2 hits.
//   wave 5 (windows)  the windows the PREVIOUS tile completed, summed from the ring in the reference's sequential order
//                     (pyx:9-22, 42-46), log, optional z-score ((x - mean) / std, local/common.py:367-376), stored.
// The last section's lane leaves its outputs in a second double-buffered tile; both helper waves copy it into their view of
// the ring.  One workgroup barrier per tile of HS_TT steps.
#define HS_TT 32
struct HgaFront {                      // raw == nullptr: `data` is (S, n, C); else `raw` is (S, n, c_raw)
    const double *raw;
    int c_raw, n_grids;
    const int *src_col, *grid_of, *comp_cols, *comp_off;
};

template <bool FRONT>                  // FRONT: raw amplifier rows in, front end in the helper waves (a separate instantiation:
__global__ void __launch_bounds__(384) //   its prefetch registers would cost the plain form two waves per SIMD)
hga_stream_kernel(const double *__restrict__ data, HgaFront fe, double *__restrict__ zi, double *__restrict__ rowbuf,
                  double *__restrict__ out, HgaSos sos, int S, int C, int n, int nsec, int row0, int cap_rows, int zero_rows,
                  int rows, int W, int overlap, int sr, float wl, float ws, int apply_log, const double *__restrict__ zs_mean,
                  const double *__restrict__ zs_std, int ring_mask)
{
    __shared__ double xs[2][HS_TT + 4][16];    // inputs of a tile's steps, double buffered (+4 rows: the loop reads a group of four steps ahead)
    __shared__ double ys[2][HS_TT][16];        // the last section's outputs of a tile's steps (sample k - (2*nsec-1) at step k), double buffered
    __shared__ double coef[16][5];
    __shared__ int wtab[2][HGA_WTAB];
    __shared__ int fcol[2][16];                // front end: source column and grid of the block's 16 channels
    __shared__ int hsync;                      // front end: filter waves that have put their pieces of a tile's raw rows into LDS (4 per tile)
    extern __shared__ __attribute__((aligned(16))) double dyn[];       // ring [ring_mask + 1][16] | raw tile | means | member columns
    double *ring = dyn;
    double *rawt = dyn + (size_t)(ring_mask + 1) * 16;                  // [HS_TT][c_raw] (front end only)
    double *means = rawt + (size_t)HS_TT * fe.c_raw;                    // [HS_TT][4]
    int *cols = reinterpret_cast<int *>(means + HS_TT * 4);
    const int tid = threadIdx.x;
    // block -> (stream, group of 16 channels): the Q = C / 16 groups of a stream get block indices that are equal mod 8,
    // i.e. one XCD under round-robin placement (speed only)
    const int Q = C >> 4;
    const int chunk = blockIdx.x / (8 * Q), within = blockIdx.x - chunk * 8 * Q;
    const int s = chunk * 8 + (within & 7), c0 = (within >> 3) * 16;
    if (s >= S) return;                                    // (whole block: S not a multiple of 8)
    const int nsec2 = 2 * nsec;
    const int steps = n + nsec2 - 1;
    const int ntiles = (steps + HS_TT - 1) / HS_TT;
    constexpr bool raw_mode = FRONT;
    if (tid == 0) hsync = 0;
    __syncthreads();

    if (tid < 256) {
        // ================================ filter waves ================================================================
        const int r = tid & 15, pib = tid >> 4;
        const bool has_sec = r < nsec2;
        const bool is_last = r == nsec2 - 1;
        const int f = has_sec ? r / nsec : 0, q = has_sec ? r - f * nsec : 0;
        if (tid < 16) {
            const int ff = tid < nsec2 ? tid / nsec : 0, qq = tid < nsec2 ? tid - ff * nsec : 0;
            coef[tid][0] = sos.k[ff][qq][0]; coef[tid][1] = sos.k[ff][qq][1]; coef[tid][2] = sos.k[ff][qq][2];
            coef[tid][3] = sos.k[ff][qq][4]; coef[tid][4] = sos.k[ff][qq][5];
        }
        double *zp = zi + (size_t)s * 2 * 8 * 2 * C + c0 + pib;
        double z0 = zp[((f * 8 + q) * 2 + 0) * (size_t)C], z1 = zp[((f * 8 + q) * 2 + 1) * (size_t)C];
        double b0 = 0, b1 = 0, b2 = 0, a1 = 0, a2 = 0, y = 0.0;
        // Front end: the filter waves also move the raw rows.  Their 256 lanes fetch a tile's raw rows (contiguous, 16-byte
        // pieces, nine per lane) TWO tiles ahead into registers the filter loop leaves free, and put them into LDS one tile
        // ahead, where the stager finds them (arrivals counted in hsync): a 32-step tile is shorter than an HBM round trip.
        constexpr int RP = 9;                                  // 9 x 256 lanes x 2 doubles = 32 rows of up to 144 columns
        double2 rv[FRONT ? RP : 1];
        const int c_raw = fe.c_raw;
        const double *praw = FRONT ? fe.raw + (size_t)s * n * c_raw : nullptr;
#define HS_RAW_FETCH(NT)                                                                         \
        if constexpr (FRONT) {                                                                   \
            const int base_ = (NT) * HS_TT;                                                      \
            const int n_el_ = max(0, min(HS_TT, n - base_)) * c_raw;                             \
            const double *src_ = praw + (size_t)base_ * c_raw;                                   \
            const bool al_ = (reinterpret_cast<size_t>(src_) & 15) == 0;                         \
            _Pragma("unroll") for (int u = 0; u < RP; ++u) {                                     \
                const int i2 = (u * 256 + tid) * 2;            /* first double of the piece */   \
                double2 v = {0.0, 0.0};                                                          \
                if (i2 + 1 < n_el_) {                                                            \
                    if (al_) v = *reinterpret_cast<const double2 *>(src_ + i2);                  \
                    else { v.x = src_[i2]; v.y = src_[i2 + 1]; }                                 \
                } else if (i2 < n_el_) v.x = src_[i2];                                           \
                rv[u] = v;                                                                       \
            }                                                                                    \
        }
#define HS_RAW_STORE(NT)                                                                         \
        if constexpr (FRONT) {                                                                   \
            const int n_el_ = max(0, min(HS_TT, n - (NT) * HS_TT)) * c_raw;                      \
            _Pragma("unroll") for (int u = 0; u < RP; ++u) {                                     \
                const int i2 = (u * 256 + tid) * 2;                                              \
                if (i2 + 1 < n_el_) *reinterpret_cast<double2 *>(rawt + i2) = rv[u];             \
                else if (i2 < n_el_) rawt[i2] = rv[u].x;                                         \
            }                                                                                    \
            if ((tid & 63) == 0) __hip_atomic_fetch_add(&hsync, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP); \
        }
        HS_RAW_FETCH(0)
        HS_RAW_STORE(0)
        if (1 < ntiles) HS_RAW_FETCH(1)
        for (int it = 0; it < ntiles; ++it) {
            __syncthreads();                                   // tile `it` staged (and, it == 0, coef and the ring's first rows)
            if (it == 0) { b0 = coef[r][0]; b1 = coef[r][1]; b2 = coef[r][2]; a1 = coef[r][3]; a2 = coef[r][4]; }
            if (it + 1 < ntiles) HS_RAW_STORE(it + 1)          // raw rows of the next tile (fetched during the previous one) for the stager
            if (it + 2 < ntiles) HS_RAW_FETCH(it + 2)
            const int base = it * HS_TT;
            const double(*xt)[16] = xs[it & 1];
            double(*yt)[16] = ys[it & 1];
            const int kend = min(base + HS_TT, steps);
            int k = base;
            // steps on which some lanes have no sample yet (pipeline filling) or no more (draining): predicated
            for (; k < kend && (k < nsec2 - 1 || k >= n); ++k) {
                const double in = hga_shift_in(xt[k - base][pib], y);
                const int t = k - r;
                if (has_sec && t >= 0 && t < n) {
                    HGA_BIQUAD(in)
                    if (is_last) yt[k - base][pib] = y;
                }
            }
            const int ksteady = min(kend, n);
            {
                // Steady state, four steps per trip.  The four inputs of the NEXT trip are read before this trip's outputs
                // are written: LDS operations of a wave complete in order, so a read issued behind a write would make
                // every step wait for its predecessor's store.  Constant LDS offsets; per step 9 fp64 operations, 2 DPP
                // moves, one read and (last section's lane) one write.
                const double *xp = &xt[k - base][pib];
                double *yl = &yt[k - base][pib];
                double xa[4], xb[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) xa[u] = xp[16 * u];
                int kk = 0;
                for (; k + 4 <= ksteady; k += 4, kk += 4) {
#pragma unroll
                    for (int u = 0; u < 4; ++u) xb[u] = xp[16 * (kk + 4 + u)];      // (rows beyond the tile: the +4 spare rows, unused)
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const double in = hga_shift_in(xa[u], y);
                        HGA_BIQUAD(in)
                        if (is_last) yl[16 * (kk + u)] = y;
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int u = 0; u < 4; ++u) xa[u] = xb[u];
                }
#pragma unroll
                for (int u = 0; u < 3; ++u) {                      // up to three steps left in the tile
                    if (k < ksteady) {
                        const double in = hga_shift_in(xa[u], y);
                        HGA_BIQUAD(in)
                        if (is_last) yl[16 * (kk + u)] = y;
                        ++k;
                    }
                }
            }
            for (; k < kend; ++k) {                            // draining steps at the end of the last tile(s)
                const double in = hga_shift_in(xt[k - base][pib], y);
                const int t = k - r;
                if (has_sec && t >= 0 && t < n) {
                    HGA_BIQUAD(in)
                    if (is_last) yt[k - base][pib] = y;
                }
            }
        }
        __syncthreads();                                       // the last tile's outputs are in ys
        if (has_sec) {
            zp[((f * 8 + q) * 2 + 0) * (size_t)C] = z0;
            zp[((f * 8 + q) * 2 + 1) * (size_t)C] = z1;
        }
#undef HS_RAW_FETCH
#undef HS_RAW_STORE
    } else {
        // ================================ helper waves ================================================================
        // wave 4: stager (next tile's inputs; front end: grid means and channel selection over the raw rows the filter
        //         waves have put into LDS); wave 5: windows (completed windows from the ring).
        // Global loads run TWO tiles ahead of the filter, in registers: a tile of 32 steps is shorter than an HBM round trip.
        const bool stager = tid < 320;
        const int hl = tid & 63;
        const int p = hl & 15, wi = hl >> 4;
        const int c_raw = fe.c_raw;
        constexpr int PL = HS_TT / 4;                          // plain: rows per lane and tile
        double pv[FRONT ? 1 : PL];
#define HS_PLAIN_FETCH(NT)                                                                       \
        if constexpr (!FRONT) {                                                                  \
            const int base_ = (NT) * HS_TT;                                                      \
            const int nrows_ = max(0, min(HS_TT, n - base_));                                    \
            const double *src_ = data + ((size_t)s * n + base_) * C + c0;                        \
            _Pragma("unroll") for (int j = 0; j < PL; ++j) {                                     \
                const int tt = wi + 4 * j;                                                       \
                pv[j] = tt < nrows_ ? src_[(size_t)tt * C + p] : 0.0;                            \
            }                                                                                    \
        }
        __builtin_amdgcn_s_setprio(2);                         // little work, but the filter waves wait for it at every barrier
        if (FRONT && stager) {
            if (hl < 16) { fcol[0][hl] = fe.src_col[c0 + hl]; fcol[1][hl] = fe.grid_of[c0 + hl]; }
            for (int k = hl; k < fe.comp_off[fe.n_grids]; k += 64) cols[k] = fe.comp_cols[k];
        }
        if (!stager) {
            for (int w = hl; w < W && w < HGA_WTAB; w += 64) {
                const int st = hga_win_start(w, ws, sr);
                wtab[0][w] = st;
                wtab[1][w] = hga_win_stop(st, wl, sr);
            }
        }
        // Both helper waves keep the ring: each copies every tile's outputs into it and reads back only what it wrote itself
        // (LDS operations of one wave are ordered), so the two never have to wait for each other.
        // Rows [0, row0) first: zeros (CASE 2, pyx:116) or the overlap the previous call left (CASE 3, pyx:123-131).
        for (int idx = hl; idx < row0 * 16; idx += 64) {
            const int rr = idx >> 4, pp = idx & 15;
            ring[(rr & ring_mask) * 16 + pp] = rr >= zero_rows ? rowbuf[((size_t)s * cap_rows + rr) * C + c0 + pp] : 0.0;
        }
        const double zm = zs_mean ? zs_mean[c0 + p] : 0.0, zd = zs_std ? zs_std[c0 + p] : 1.0;
        // windows: without the front end the two waves take alternate windows; with it the stager has the grid means to do
        constexpr int NSPLIT = FRONT ? 1 : 2;
        const bool sums = FRONT ? !stager : true;
        int w_mine = (FRONT || stager) ? 0 : 1;                // next window of this wave's share (wave-uniform)
        if (stager) HS_PLAIN_FETCH(0)                          // prologue: tile 0 now, tile 1 in flight
        for (int it = -1; it <= ntiles; ++it) {
            if (it >= 0) __syncthreads();                      // it < ntiles: tile `it` handed over; it == ntiles: the final barrier
            const int nt = it + 1;                             // the tile being staged while tile `it` is filtered
            if (nt < ntiles) {
                double(*xt)[16] = xs[nt & 1];
                if constexpr (FRONT) {
                    if (stager) {
                        // all four filter waves have put their pieces of tile nt's raw rows into LDS
                        while (__hip_atomic_load(&hsync, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < 4 * (nt + 1)) __builtin_amdgcn_s_sleep(2);
                        const int nrows = max(0, min(HS_TT, n - nt * HS_TT));
                        // grid means: one lane per (row, grid), members added in list order (numpy's column-at-a-time mean)
                        for (int ch = hl; ch < nrows * fe.n_grids; ch += 64) {
                            const int rr = ch / fe.n_grids, g = ch - rr * fe.n_grids;
                            const double *row = rawt + (size_t)rr * c_raw;
                            const int a = fe.comp_off[g], b2 = fe.comp_off[g + 1];
                            double sum = 0.0;
                            int k = a;
                            for (; k + 8 <= b2; k += 8) {
                                double v[8];
#pragma unroll
                                for (int u = 0; u < 8; ++u) v[u] = row[cols[k + u]];
#pragma unroll
                                for (int u = 0; u < 8; ++u) sum += v[u];
                            }
                            for (; k < b2; ++k) sum += row[cols[k]];
                            means[rr * 4 + g] = sum / (double)(b2 - a);
                        }
                        // the block's 16 channels: select, reference, store (rows without samples: zeros)
#pragma unroll
                        for (int j = 0; j < PL; ++j) {
                            const int tt = wi + 4 * j;
                            double o = 0.0;
                            if (tt < nrows) {
                                const int g = fcol[1][p];
                                const double v = rawt[(size_t)tt * c_raw + fcol[0][p]];
                                o = g >= 0 ? v - means[tt * 4 + g] : v;
                            }
                            xt[tt][p] = o;
                        }
                    }
                } else {
                    if (stager) {
#pragma unroll
                        for (int j = 0; j < PL; ++j) xt[wi + 4 * j][p] = pv[j];
                        if (nt + 1 < ntiles) HS_PLAIN_FETCH(nt + 1)
                    }
                }
            }
            if (it < 1) continue;
            // ---- tile it - 1 is finished: its outputs join the ring at their rows, then the windows it completed ----
            const int pbase = (it - 1) * HS_TT;
            const int kend = min(it * HS_TT, steps);
            {
                const double(*yt)[16] = ys[(it - 1) & 1];
                double v[PL];
#pragma unroll
                for (int j = 0; j < PL; ++j) v[j] = yt[wi + 4 * j][p];
#pragma unroll
                for (int j = 0; j < PL; ++j) {
                    const int kstep = pbase + wi + 4 * j, t = kstep - (nsec2 - 1);
                    if (kstep < kend && t >= 0 && t < n) ring[((row0 + t) & ring_mask) * 16 + p] = v[j];
                }
            }
            if (!sums) continue;
            int t_done = kend - (nsec2 - 1);
            t_done = t_done < 0 ? 0 : (t_done > n ? n : t_done);
            const int rows_done = row0 + t_done;
            for (int w = w_mine + wi * NSPLIT; w < W; w += 4 * NSPLIT) {
                const int start = w < HGA_WTAB ? wtab[0][w] : hga_win_start(w, ws, sr);
                const int stop = w < HGA_WTAB ? wtab[1][w] : hga_win_stop(start, wl, sr);
                if (stop > rows_done) break;
                double sum = 0.0;
                int rr = start;
                for (; rr + 8 <= stop; rr += 8) {              // eight ring reads in flight, then their terms in row order
                    double v[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) v[u] = ring[((rr + u) & ring_mask) * 16 + p];
#pragma unroll
                    for (int u = 0; u < 8; ++u) sum += v[u] * v[u];
                }
                for (; rr < stop; ++rr) {
                    const double v = ring[(rr & ring_mask) * 16 + p];
                    sum += v * v;
                }
                const double pw = sum / (double)(stop - start) + 0.01;
                double o = apply_log ? log(pw) : pw;
                if (zs_mean) o = (o - zm) / zd;
                out[((size_t)s * W + w) * C + c0 + p] = o;
            }
            while (w_mine < W && (w_mine < HGA_WTAB ? wtab[1][w_mine] : hga_win_stop(hga_win_start(w_mine, ws, sr), wl, sr)) <= rows_done)
                w_mine += NSPLIT;
        }
        if (stager) {
            // the last `overlap` rows become rows 0..overlap-1 of the next call (WarmStartFrameBuffer.remainder_data)
            for (int idx = hl; idx < overlap * 16; idx += 64) {
                const int kk = idx >> 4, pp = idx & 15;
                rowbuf[((size_t)s * cap_rows + kk) * C + c0 + pp] = ring[((rows - overlap + kk) & ring_mask) * 16 + pp];
            }
        }
#undef HS_PLAIN_FETCH
    }
}

// ---- stage 2: windowed mean power ---------------------------------------------------------------------------------
// A 256-thread block takes 8 consecutive windows x 32 consecutive channels of one stream: the rows those windows cover
// (50 + 7*10 for the reference's 50 ms / 10 ms at 1 kHz) are staged once through LDS in 256-byte row segments instead
// of being fetched by each of the five windows that overlap them; lane (window, channel) then runs its window's
// sequential sum over rows exactly as array_sum_and_power does (pyx:9-22).
#define HGA_WB 8
#define HGA_WC 32
#define HGA_WROWS 160            // LDS rows per block; a block whose windows span more falls back to global reads

__global__ void __launch_bounds__(256)
hga_window_kernel(const double *__restrict__ rowbuf, double *__restrict__ out, int S, int C, int W, int cap_rows, int sr,
                  float wl, float ws, int apply_log)
{
    __shared__ double tile[HGA_WROWS][HGA_WC];
    const int tid = threadIdx.x, cl = tid & (HGA_WC - 1), wi = tid / HGA_WC;
    const int cgroups = (C + HGA_WC - 1) / HGA_WC, wchunks = (W + HGA_WB - 1) / HGA_WB;
    int bid = blockIdx.x;
    const int wc = bid % wchunks; bid /= wchunks;
    const int cg = bid % cgroups;
    const int s = bid / cgroups;
    const int w0 = wc * HGA_WB, wlast = min(w0 + HGA_WB, W) - 1;
    const int row_lo = hga_win_start(w0, ws, sr);
    const int row_hi = hga_win_stop(hga_win_start(wlast, ws, sr), wl, sr);
    const int nrows = row_hi - row_lo;
    const bool staged = nrows <= HGA_WROWS;
    const double *base = rowbuf + (size_t)s * cap_rows * C;
    const int c0 = cg * HGA_WC;
    if (staged) {
        for (int idx = tid; idx < nrows * HGA_WC; idx += 256) {
            const int rr = idx / HGA_WC, cc = idx - rr * HGA_WC;
            tile[rr][cc] = (c0 + cc < C) ? base[(size_t)(row_lo + rr) * C + c0 + cc] : 0.0;
        }
    }
    __syncthreads();
    const int win = w0 + wi, c = c0 + cl;
    if (win >= W || c >= C) return;
    const int start = hga_win_start(win, ws, sr);
    const int stop = hga_win_stop(start, wl, sr);
    double sum = 0.0;
    if (staged) {
        for (int rr = start; rr < stop; ++rr) {
            const double v = tile[rr - row_lo][cl];
            sum += v * v;
        }
    } else {
        for (int rr = start; rr < stop; ++rr) {
            const double v = base[(size_t)rr * C + c];
            sum += v * v;
        }
    }
    const double p = sum / (double)(stop - start) + 0.01;
    out[((size_t)s * W + win) * C + c] = apply_log ? log(p) : p;
}

// ---- stage 3: keep the last `overlap` rows (ascending copy: a source row is always ahead of its destination) ------
__global__ void __launch_bounds__(256)
hga_overlap_kernel(double *__restrict__ rowbuf, int S, int C, int cap_rows, int rows, int overlap)
{
    const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= (long)S * C) return;
    const int s = (int)(gid / C), c = (int)(gid - (long)s * C);
    double *col = rowbuf + (size_t)s * cap_rows * C + c;
    for (int k = 0; k < overlap; ++k) col[(size_t)k * C] = col[(size_t)(rows - overlap + k) * C];
}

// dynamic LDS of hga_stream_kernel for this extractor (and front end), 0 when the streamed form cannot take the shape
static size_t hga_stream_lds(const DssHgaDev &h, const DssHgaFrontDev *fe, int *ring_rows_out)
{
    if ((h.C & 15) != 0 || h.force_path != 3) return 0;     // opt-in (dss_selftest_hga_force_path(h, 3)): measured slower than hga_fused_kernel
    const int shift = h.frame_length - h.overlap;
    int ring_rows = 64;
    while (ring_rows < h.frame_length + shift + HS_TT + 2 && ring_rows < (1 << 20)) ring_rows *= 2;
    size_t dyn = (size_t)ring_rows * 16 * sizeof(double);
    if (fe) dyn += ((size_t)HS_TT * fe->c_raw + HS_TT * 4) * sizeof(double) + (size_t)4 * fe->c_raw * sizeof(int);
    if (dyn > 96 * 1024 || h.overlap > ring_rows || (fe && (fe->n_grids > 4 || fe->c_raw > 144))) return 0;   // 144: 36 pieces x 64 lanes x 2 doubles per 32 rows
    if (ring_rows_out) *ring_rows_out = ring_rows;
    return dyn;
}

int dss_hga_stream_fits(const DssHgaDev &h, const DssHgaFrontDev *fe) { return hga_stream_lds(h, fe, nullptr) != 0; }

int dss_launch_hga(const DssHgaDev &h, const double *d_data, const DssHgaFrontDev *fe, int n, int row0, int zero_rows, int rows,
                   int W, double *d_out, int apply_log, hipStream_t st)
{
    HgaSos sos;
    memcpy(&sos, h.sos, sizeof(sos));
    const long pairs = (long)h.S * h.C;
    const int shift = h.frame_length - h.overlap;
    // Streamed form (one launch, filter waves + helper waves): channel groups of 16, ring = frame + shift + two tiles of rows
    {
        int ring_rows = 0;
        const size_t dyn = hga_stream_lds(h, fe, &ring_rows);
        if (dyn && rows < (1 << 30)) {
            static unsigned long long attr_set = 0;            // per device; benign if two threads both set it
            int dev = 0;
            DSS_HIP_CHECK(hipGetDevice(&dev));
            if (!(attr_set >> (dev & 63) & 1)) {
                DSS_HIP_CHECK(hipFuncSetAttribute((const void *)hga_stream_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
                DSS_HIP_CHECK(hipFuncSetAttribute((const void *)hga_stream_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
                attr_set |= 1ull << (dev & 63);
            }
            HgaFront f = {nullptr, 0, 0, nullptr, nullptr, nullptr, nullptr};
            if (fe) f = HgaFront{fe->raw, fe->c_raw, fe->n_grids, fe->src_col, fe->grid_of, fe->comp_cols, fe->comp_off};
            const int Q = h.C / 16;
            const unsigned grid = (unsigned)(((h.S + 7) / 8) * 8 * Q);
            if (fe)
                hipLaunchKernelGGL(hga_stream_kernel<true>, dim3(grid), dim3(384), dyn, st, d_data, f, h.zi, h.rows, d_out, sos, h.S, h.C,
                                   n, h.nsec, row0, h.cap_rows, zero_rows, rows, W, h.overlap, h.fs, h.wl, h.ws, apply_log, h.zs_mean,
                                   h.zs_std, ring_rows - 1);
            else
                hipLaunchKernelGGL(hga_stream_kernel<false>, dim3(grid), dim3(384), dyn, st, d_data, f, h.zi, h.rows, d_out, sos, h.S, h.C,
                                   n, h.nsec, row0, h.cap_rows, zero_rows, rows, W, h.overlap, h.fs, h.wl, h.ws, apply_log, h.zs_mean,
                                   h.zs_std, ring_rows - 1);
            DSS_HIP_CHECK(hipGetLastError());
            return DSS_OK;
        }
    }
    if (fe) { dss_set_error("HGA: the one-launch front end needs the streamed form, a multiple of 16 channels and tiles that fit LDS"); return DSS_EINVAL; }
    if (h.force_path != 2) {   // fused form when the ring (frame + shift + one tile of rows) fits beside the tiles
        const int TT = h.force_path == 1 ? 32 : 64;                  // 1: 32-step tiles, 7 blocks per CU instead of 5 (A/B timing: no faster)
        int ring_rows = h.frame_length + shift + TT + 2;
        ring_rows = (ring_rows + 7) & ~7;
        const size_t ring_bytes = (size_t)ring_rows * 16 * sizeof(double);
        if (ring_bytes <= 40 * 1024 && rows - h.overlap + TT <= (1 << 30) && h.overlap <= ring_rows && row0 <= ring_rows) {
            if (TT == 32)
                hipLaunchKernelGGL(hga_fused_kernel<32>, dim3((unsigned)((pairs + 15) / 16)), dim3(256), ring_bytes, st, d_data, h.zi,
                                   h.rows, d_out, sos, h.S, h.C, n, h.nsec, row0, h.cap_rows, zero_rows, rows, W, h.overlap, h.fs, h.wl,
                                   h.ws, apply_log, ring_rows, h.zs_mean, h.zs_std);
            else
                hipLaunchKernelGGL(hga_fused_kernel<64>, dim3((unsigned)((pairs + 15) / 16)), dim3(256), ring_bytes, st, d_data, h.zi,
                                   h.rows, d_out, sos, h.S, h.C, n, h.nsec, row0, h.cap_rows, zero_rows, rows, W, h.overlap, h.fs, h.wl,
                                   h.ws, apply_log, ring_rows, h.zs_mean, h.zs_std);
            DSS_HIP_CHECK(hipGetLastError());
            return DSS_OK;
        }
    }
    if (h.zs_mean) { dss_set_error("HGA: no z-score epilogue in the three-launch form (window shape too large for the LDS ring)"); return DSS_EINVAL; }
    hipLaunchKernelGGL(hga_filter_kernel, dim3((unsigned)((pairs + 15) / 16)), dim3(256), 0, st, d_data, h.zi, h.rows, sos,
                       h.S, h.C, n, h.nsec, row0, h.cap_rows, zero_rows);
    DSS_HIP_CHECK(hipGetLastError());
    if (W > 0) {
        const long blocks = (long)h.S * ((h.C + HGA_WC - 1) / HGA_WC) * ((W + HGA_WB - 1) / HGA_WB);
        hipLaunchKernelGGL(hga_window_kernel, dim3((unsigned)blocks), dim3(256), 0, st, h.rows, d_out, h.S, h.C, W, h.cap_rows,
                           h.fs, h.wl, h.ws, apply_log);
        DSS_HIP_CHECK(hipGetLastError());
    }
    hipLaunchKernelGGL(hga_overlap_kernel, dim3((unsigned)((pairs + 255) / 256)), dim3(256), 0, st, h.rows, h.S, h.C,
                       h.cap_rows, rows, h.overlap);
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


// Fused front end: column reorder + per-grid common average reference + channel selection.  A block stages FE_ROWS
// consecutive (stream, sample) rows of the raw packet through LDS with coalesced loads (lanes across the c_raw columns;
// the odd row length keeps column walks conflict-free), then one lane per (row, grid) runs the grid mean as the
// SEQUENTIAL sum the reference's numpy expression performs (np.mean over a Fortran-ordered fancy-index copy adds one
// column at a time) followed by one division, and finally lanes across the C output channels subtract and store
// coalesced.  Bit-identical to the numpy chain (local/common.py:16-58,308-345).
#define FE_ROWS 32
__global__ void __launch_bounds__(256)
hga_frontend_kernel(const double *__restrict__ raw, double *__restrict__ pre, int total, int c_raw, int C,
                    const int *__restrict__ src_col, const int *__restrict__ grid_of, int n_grids,
                    const int *__restrict__ comp_cols, const int *__restrict__ comp_off)
{
    extern __shared__ __attribute__((aligned(16))) double fe_tile[];     // [FE_ROWS][c_raw], [FE_ROWS][4] means, [<= 4 * c_raw] member columns
    double *means = fe_tile + (size_t)FE_ROWS * c_raw;
    int *cols = reinterpret_cast<int *>(means + FE_ROWS * 4);            // the grids' member columns, in summation order
    const int tid = threadIdx.x;
    const long row_base = (long)blockIdx.x * FE_ROWS;
    const int nrows = (int)min((long)FE_ROWS, (long)total - row_base);
    const double *src = raw + (size_t)row_base * c_raw;
    // rows are contiguous: fully coalesced; eight loads per thread in flight (a loop of unknown length is not unrolled,
    // and one load per trip would expose the HBM latency sixteen times per tile)
    const int n_el = nrows * c_raw;
    for (int base = 0; base < n_el; base += 8 * 256) {
        double v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { const int idx = base + u * 256 + tid; v[u] = idx < n_el ? src[idx] : 0.0; }
#pragma unroll
        for (int u = 0; u < 8; ++u) { const int idx = base + u * 256 + tid; if (idx < n_el) fe_tile[idx] = v[u]; }
    }
    for (int k = tid; k < comp_off[n_grids]; k += 256) cols[k] = comp_cols[k];
    __syncthreads();
    if (tid < nrows * n_grids) {
        const int rr = tid / n_grids, g = tid - rr * n_grids;
        const double *row = fe_tile + (size_t)rr * c_raw;
        const int a = comp_off[g], b2 = comp_off[g + 1];
        double sum = 0.0;
        int k = a;
        for (; k + 8 <= b2; k += 8) {              // eight members' reads in flight, then their terms in list order
            double v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = row[cols[k + u]];
#pragma unroll
            for (int u = 0; u < 8; ++u) sum += v[u];
        }
        for (; k < b2; ++k) sum += row[cols[k]];
        means[rr * 4 + g] = sum / (double)(b2 - a);
    }
    __syncthreads();
    const int n_out = nrows * C;                    // eight independent elements per thread and trip, as for the loads
    for (int base = 0; base < n_out; base += 8 * 256) {
        double o[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int idx = base + u * 256 + tid;
            o[u] = 0.0;
            if (idx < n_out) {
                const int rr = idx / C, c = idx - rr * C;
                const int g = grid_of[c];
                const double v = fe_tile[(size_t)rr * c_raw + src_col[c]];
                o[u] = g >= 0 ? v - means[rr * 4 + g] : v;
            }
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int idx = base + u * 256 + tid;
            if (idx < n_out) pre[(size_t)row_base * C + idx] = o[u];
        }
    }
}

int dss_launch_hga_frontend(const double *d_raw, double *d_pre, int S, int n, int c_raw, int C, const int *src_col,
                            const int *grid_of, int n_grids, const int *comp_cols, const int *comp_off, hipStream_t st)
{
    const int total = S * n;
    const size_t lds = ((size_t)FE_ROWS * c_raw + FE_ROWS * 4) * sizeof(double) + (size_t)4 * c_raw * sizeof(int);   // member lists: at most 4 grids of c_raw columns
    if (lds > 64 * 1024 || n_grids > 4) { dss_set_error("front end: %d raw columns / %d grids exceed the tile", c_raw, n_grids); return DSS_EINVAL; }
    hipLaunchKernelGGL(hga_frontend_kernel, dim3((total + FE_ROWS - 1) / FE_ROWS), dim3(256), lds, st, d_raw, d_pre, total, c_raw, C,
                       src_col, grid_of, n_grids, comp_cols, comp_off);
    DSS_HIP_CHECK(hipGetLastError());
    return DSS_OK;
}
