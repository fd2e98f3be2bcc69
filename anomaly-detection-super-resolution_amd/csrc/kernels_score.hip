// kernels_score.hip - reconstruction-error scorer on gfx950 (reference src/evaluate.py:204-265,
// src/metrics.py:15-108, src/trainer.py:45-47).  HBM/L2-bound integer + fp work, no MFMA.
//
// SSIM window sweep: the reference evaluates ssim_numpy once per (image pair, window size) with a
// Python loop over every pixel (O(H W ws^2)).  Here each image pair gets five float64 summed-area
// tables (x, y, x*x, y*y, x*y of the luminance planes) built once; any window size is then 4 table
// look-ups per quantity and pixel, and numpy's "reflect" padding becomes at most 3 x 3 rectangles of
// the unpadded table (a window that hangs over an edge covers the reflected rows/columns a second
// time).  The per-pixel SSIM arithmetic is done in fp32 in the reference's operation order.
#include "engine.h"
#include "../../include/srad.h"
#include <algorithm>
#include <math.h>
#include <stdlib.h>
#include <type_traits>
#include <vector>

namespace {

constexpr int kQ = 5;            // x, y, xx, yy, xy
constexpr int kEvalPix = 1024;   // pixels per workgroup in the evaluation kernel

__device__ __forceinline__ float luma_of(const uint8_t* px, int C) {
  // hr.astype(float32) / 255.0, then tensordot with [65.738, 129.057, 25.064] / 256 (metrics.py:37-39)
  if (C == 1) return (float)px[0] / 255.0f;
  const float c0 = 65.738f / 256.0f, c1 = 129.057f / 256.0f, c2 = 25.064f / 256.0f;
  const float r = (float)px[0] / 255.0f, g = (float)px[1] / 255.0f, b = (float)px[2] / 255.0f;
  return r * c0 + g * c1 + b * c2;
}

// Table layout: planar, sat[img][q][row + 1][col + 1] (row 0 and column 0 are zero), so that the 64 lanes of a wave - 64
// consecutive pixels of a row - read and write 512 contiguous bytes per quantity.
//
// Row pass: one wave per (image, table row).  The row is walked in 64-pixel chunks; a chunk's five running sums are a
// 6-step wave scan (float64 shuffles) plus the carry of the chunks before it.
__global__ __launch_bounds__(256) void sat_rows_kernel(const uint8_t* __restrict__ sr, const uint8_t* __restrict__ hr,
                                                       double* __restrict__ sat, int n_img, int H, int W, int C) {
  const int lane = threadIdx.x & 63;
  const int t = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (t >= n_img * (H + 1)) return;
  const int img = t / (H + 1), r = t - img * (H + 1);
  const size_t plane = (size_t)(H + 1) * (W + 1);
  double* const base = sat + (size_t)img * kQ * plane + (size_t)r * (W + 1);
  if (r == 0) {
    for (int c = lane; c <= W; c += 64)
#pragma unroll
      for (int q = 0; q < kQ; ++q) base[q * plane + c] = 0.0;
    return;
  }
  if (lane == 0) {
#pragma unroll
    for (int q = 0; q < kQ; ++q) base[q * plane] = 0.0;
  }
  const uint8_t* ps = sr + ((size_t)img * H + (r - 1)) * W * C;
  const uint8_t* ph = hr + ((size_t)img * H + (r - 1)) * W * C;
  double carry[kQ] = {0, 0, 0, 0, 0};
  for (int c0 = 0; c0 < W; c0 += 64) {
    const int c = c0 + lane;
    double v[kQ] = {0, 0, 0, 0, 0};
    if (c < W) {
      const float x = luma_of(ph + (size_t)c * C, C);     // "ref" = HR
      const float y = luma_of(ps + (size_t)c * C, C);     // "out" = SR
      v[0] = (double)x; v[1] = (double)y;
      v[2] = (double)(x * x); v[3] = (double)(y * y); v[4] = (double)(x * y);    // fp32 products as in metrics.py:60-62
    }
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
#pragma unroll
      for (int q = 0; q < kQ; ++q) {
        const double up = __shfl_up(v[q], o);
        if (lane >= o) v[q] += up;
      }
    }
#pragma unroll
    for (int q = 0; q < kQ; ++q) {
      v[q] += carry[q];
      if (c < W) base[q * plane + c + 1] = v[q];
      carry[q] = __shfl(v[q], 63);
    }
  }
}

// Column pass, in two launches so that a 1024-row table is not one thread's 1024 dependent steps: (1) every 32-row
// segment of a column is summed down in place (its last row = the segment's total, also copied to `segtot`); (2) each
// segment adds the totals of the segments above it.  Threads of a wave are consecutive columns: coalesced rows.
constexpr int kSeg = 32;
__global__ void sat_cols_local_kernel(double* __restrict__ sat, double* __restrict__ segtot, int n_planes, int H, int W, int nseg) {
  const size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  const size_t per = (size_t)(W + 1);
  if (t >= (size_t)n_planes * nseg * per) return;
  const int col = (int)(t % per);
  const int seg = (int)((t / per) % nseg);
  const int pl = (int)(t / (per * nseg));
  double* p = sat + (size_t)pl * (H + 1) * per + col;
  const int r0 = seg * kSeg + 1, r1 = min(H, r0 + kSeg - 1);
  double acc = 0.0;
  for (int r = r0; r <= r1; ++r) {
    acc += p[(size_t)r * per];
    p[(size_t)r * per] = acc;
  }
  segtot[((size_t)pl * nseg + seg) * per + col] = acc;
}
__global__ void sat_cols_carry_kernel(double* __restrict__ sat, const double* __restrict__ segtot, int n_planes, int H, int W, int nseg) {
  const size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  const size_t per = (size_t)(W + 1);
  if (t >= (size_t)n_planes * (nseg - 1) * per) return;
  const int col = (int)(t % per);
  const int seg = (int)((t / per) % (nseg - 1)) + 1;
  const int pl = (int)(t / (per * (nseg - 1)));
  double carry = 0.0;
  for (int s = 0; s < seg; ++s) carry += segtot[((size_t)pl * nseg + s) * per + col];
  double* p = sat + (size_t)pl * (H + 1) * per + col;
  const int r0 = seg * kSeg + 1, r1 = min(H, r0 + kSeg - 1);
  for (int r = r0; r <= r1; ++r) p[(size_t)r * per] += carry;
}

// Reflect-padded window along one axis as table differences.  Padded coordinates lo..hi (each side leaves [0, n-1] by less
// than n) cover the real indices [max(lo,0), min(hi,n-1)] once, [1, -lo] once more if lo < 0, and [2(n-1)-hi, n-2] once more if
// hi > n-1 (numpy "reflect": the edge sample is not repeated).  With P the prefix table (P[0] = 0) the covered sum is three
// (+, -) pairs of table entries; a pair that is not needed has both indices equal and cancels.
__device__ __forceinline__ void axis_pairs(int lo, int hi, int n, int (&idx)[6]) {
  idx[0] = min(hi, n - 1) + 1; idx[1] = max(lo, 0);                       // + P[idx0] - P[idx1]: the part inside the image
  idx[2] = lo < 0 ? -lo + 1 : 1; idx[3] = 1;                              // + P[-lo + 1] - P[1]: reflected over the low edge
  idx[4] = n - 1; idx[5] = hi > n - 1 ? 2 * (n - 1) - hi : n - 1;         // + P[n - 1] - P[2 (n - 1) - hi]: over the high edge
}

// Evaluation: one workgroup = kEvalPix consecutive pixels of one image for ONE window size; up to kWsGroup window sizes share
// a launch (the list travels as a kernel argument; an image's window sizes run back to back).  SSIM map values summed per
// block (float64, fixed order).  A pixel row i with pad p = ws / 2 reads table rows i - p and i + p + 1 (plus the reflected ones
// near the borders): 80 B per pixel and window size, each table row exactly twice per window size.
//
// ROWS = the image width is a multiple of 64 (every MVTec size): a wave's 64 pixels are consecutive pixels of ONE row, so the row
// pairs are wave-uniform (scalar row offsets) and the reflection pairs of both axes are skipped by wave-uniform branches when no
// lane needs them; all loads of a (row pair, column pair) are issued before the first add.  Otherwise every lane works out its
// own pixel (any width).  Same arithmetic either way.
// (Tried: XCD k owns a strip of pixel blocks and window size k works p_k rows further down, so that all window sizes of a
// launch share their top table row in one L2 - 5.0 -> 5.8 ms at 1024 px: the sweep is not bound by the fabric side.)
constexpr int kWsGroup = 16;
struct WsList { int ws[kWsGroup]; double dinv[kWsGroup]; };    // window sizes of a launch and 1 / ws^2 (a float64 division per pixel otherwise)

// Sum of the 64 SSIM map values of a wave's segment, in every lane.  The values are float32 in [-1, 1]; they are summed as
// integers of 2^-24 (exact, so the order is irrelevant and DPP lane swizzles can do four of the six steps: a float64 shuffle
// chain was 12 ds_bpermute round trips + 30 vector instructions per 64 pixels of a kernel that is VALU-bound).
__device__ __forceinline__ double wave_sum_map(float m) {
  int v = __float2int_rn(m * 16777216.0f);
  v += __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xF, 0xF, true);
  v += __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xF, 0xF, true);
  v += __builtin_amdgcn_update_dpp(0, v, 0x141, 0xF, 0xF, true);
  v += __builtin_amdgcn_update_dpp(0, v, 0x140, 0xF, 0xF, true);
  v += __shfl_xor(v, 16);
  v += __shfl_xor(v, 32);
  return (double)v * (1.0 / 16777216.0);
}
template <bool ROWS>
__global__ __launch_bounds__(256) void ssim_eval_kernel(const double* __restrict__ sat, double* __restrict__ partial,
                                                        int H, int W, const WsList wl, int nblk, int g, int n_img) {
  const int L = blockIdx.x, per_img = nblk * g;
  const int img = L / per_img, r0 = L - img * per_img;
  const int kw = r0 / nblk, pb = r0 - kw * nblk;
  const int ws = wl.ws[kw], pad = ws / 2;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int iper = W + 1;
  const size_t plane = (size_t)(H + 1) * iper;
  const double* const S = sat + (size_t)img * kQ * plane;
  const double dinv = wl.dinv[kw];
  const float C1 = (float)(0.01 * 0.01), C2 = (float)(0.03 * 0.03);
  const int npix = H * W;
  double local = 0.0;
#pragma unroll 1
  for (int it = 0; it < kEvalPix / 256; ++it) {
    const int seg0 = pb * kEvalPix + (wave * (kEvalPix / 256) + it) * 64;    // first pixel of this wave's 64 (wave-uniform)
    if (seg0 >= npix) break;
    int i, j;
    if constexpr (ROWS) { i = seg0 / W; j = seg0 - i * W + lane; }           // i scalar
    else { const int pix = min(seg0 + lane, npix - 1); i = pix / W; j = pix - i * W; }
    int ri[6], ci[6];
    axis_pairs(i - pad, i + ws - 1 - pad, H, ri);
    axis_pairs(j - pad, j + ws - 1 - pad, W, ci);
    // does any lane of the wave need the reflection pairs?  (ROWS: the row answers are scalar anyway)
    const bool row_lo = __builtin_amdgcn_ballot_w64(i - pad < 0) != 0, row_hi = __builtin_amdgcn_ballot_w64(i + ws - 1 - pad > H - 1) != 0;
    const bool col_lo = __builtin_amdgcn_ballot_w64(j - pad < 0) != 0, col_hi = __builtin_amdgcn_ballot_w64(j + ws - 1 - pad > W - 1) != 0;
    double sum[kQ] = {0, 0, 0, 0, 0};
    auto row_pair = [&](auto RP) __attribute__((always_inline)) {             // + row ri[2 rp], - row ri[2 rp + 1]
      constexpr int rp = decltype(RP)::value;
      const int oa = ri[2 * rp] * iper, ob = ri[2 * rp + 1] * iper;
      auto col_pair = [&](auto CP) __attribute__((always_inline)) {
        constexpr int cp = decltype(CP)::value;
        double va[kQ], vb[kQ], vc[kQ], vd[kQ];
#pragma unroll
        for (int q = 0; q < kQ; ++q) {
          const double* Sq = S + (size_t)q * plane;
          va[q] = Sq[oa + ci[2 * cp]]; vb[q] = Sq[oa + ci[2 * cp + 1]];
          vc[q] = Sq[ob + ci[2 * cp]]; vd[q] = Sq[ob + ci[2 * cp + 1]];
        }
#pragma unroll
        for (int q = 0; q < kQ; ++q) sum[q] += (va[q] - vb[q]) - (vc[q] - vd[q]);
      };
      col_pair(std::integral_constant<int, 0>{});
      if (col_lo) col_pair(std::integral_constant<int, 1>{});
      if (col_hi) col_pair(std::integral_constant<int, 2>{});
    };
    row_pair(std::integral_constant<int, 0>{});
    if (row_lo) row_pair(std::integral_constant<int, 1>{});
    if (row_hi) row_pair(std::integral_constant<int, 2>{});
    const float mu1 = (float)(sum[0] * dinv), mu2 = (float)(sum[1] * dinv);
    const float mu1_sq = mu1 * mu1, mu2_sq = mu2 * mu2, mu12 = mu1 * mu2;
    const float s1 = (float)(sum[2] * dinv) - mu1_sq;
    const float s2 = (float)(sum[3] * dinv) - mu2_sq;
    const float s12 = (float)(sum[4] * dinv) - mu12;
    const float m = ((2.0f * mu12 + C1) * (2.0f * s12 + C2)) / ((mu1_sq + mu2_sq + C1) * (s1 + s2 + C2));
    if (ROWS || seg0 + lane < npix) local += (double)m;
  }
  __shared__ double red[256];
  red[threadIdx.x] = local;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) partial[((size_t)img * kWsGroup + kw) * nblk + pb] = red[0];
}

// ------------------------------------------------------------------------------------------------------------------------
// The sweep for the image widths MVTec runs at (64 / 128 / 256 / 512 / 1024: a power of two that divides 1024): table rows
// through LDS.
//
// Along the rows a reflect-padded window lo..hi is  B(hi) - T(lo)  with
//     T(lo) = P[max(lo, 0)] - (lo < 0 ? P[1 - lo] - P[1] : 0)          B(hi) = P[min(hi, H-1) + 1] + (hi > H-1 ? P[H-1] - P[2(H-1) - hi] : 0)
// (P[r] = table row r, all W + 1 columns of all five quantities).  T depends on the window's first row only, and pixel row i with
// pad p has lo = i - p: for a fixed lo = t EVERY window size of the launch meets the same T(t), at pixel row i = t + p_k.  So a
// 1024-thread workgroup owns 1024 / W consecutive values of t (1024 pixels per window size), keeps its T rows in registers, and
// for each window size of the group streams the B rows once, coalesced, writes D = B - T (the row-summed window: W + 1 prefix
// values per quantity and row) to LDS and evaluates its 1024 pixels from there - column pairs of D instead of 4 - 36 table
// corners per quantity - while the next window size's B rows are in flight in registers (two LDS images, one barrier per
// window size).
//
// The corner kernel above is bound by vector instructions and 8-byte loads (~45 per pixel and window size with the reflection
// pairs of the big windows: 4.5 ms for 2 pairs x 102 window sizes at 1024 px), so this one is built to issue few of either: a
// thread's row elements are (quantity q, row r = tid / W, column tid % W) for q = 0 .. 4 - the same table column for every
// quantity; the quantity's plane and the row (wave-uniform: W >= 64) are the SCALAR offset of a buffer load, ~6 loads and no
// address arithmetic per pixel and window size - plus column W of every (q, r) on the first 5 * 1024 / W threads; 1 / ws^2
// comes from the host; the 64 map values of a wave are summed as integers through DPP.  2.06 ms (1024 px) / 0.38 ms (78 pairs at
// 128 px, with the first, per-element form of this kernel).  What is left is a balance of LDS bytes (40 B per column term), the
// B rows' 8-byte loads and ~120 vector instructions per pixel; the order in which workgroups take the rows (XCD strips, rows
// shared between window sizes kept in one L2) changed nothing: the tables are served by the Infinity Cache fast enough.
typedef __attribute__((ext_vector_type(2))) unsigned int sc_u32x2;
__global__ __launch_bounds__(1024) void ssim_rows_lds_kernel(const double* __restrict__ sat, double* __restrict__ partial,
                                                             int H, int lw, const WsList wl, int g, int t_first, int nt_blocks) {
  const int W = 1 << lw, iper = W + 1, rb = 1024 >> lw, rowq = rb * iper, n_el = kQ * rowq;
  extern __shared__ __attribute__((aligned(16))) double Dl[];              // two images of [kQ][rb][W + 1], then [2][16] wave sums
  double* const wsum = Dl + 2 * n_el;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int img = blockIdx.x / nt_blocks, t0 = t_first + (blockIdx.x - img * nt_blocks) * rb;
  const size_t plane = (size_t)(H + 1) * iper;
  const double* const S = sat + (size_t)img * kQ * plane;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(S), 0, (int)(kQ * plane * 8), 0x00020000);
  auto ld = [&](int voff, size_t soff_el) __attribute__((always_inline)) -> double {     // table entry at voff bytes + soff_el elements
    return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(rs, voff, (int)(soff_el * 8), 0));
  };
  const int r = __builtin_amdgcn_readfirstlane((wave * 64) >> lw);         // this wave's row of the workgroup's rb (scalar)
  const int c = tid & (W - 1);
  const int t = t0 + r;
  const int v_main = c * 8;
  // column W of (quantity q, row rr) on thread q * rb + rr: per-lane rows, so the row offset is part of the vector offset
  const bool tail = tid < kQ * rb;
  const int tq = tail ? tid / rb : 0, trr = tail ? tid - tq * rb : 0;
  const bool tail_wave = wave * 64 < kQ * rb;                              // scalar
  auto tail_off = [&](int row) __attribute__((always_inline)) { return (int)((tq * plane + (size_t)row * iper + W) * 8); };
  // T(t) = P[max(t, 0)] - (t < 0 ? P[1 - t] - P[1] : 0): unconditional loads, the pair cancels for t >= 0 (row 0 = zeros)
  double Tv[kQ], Tt = 0.0;
  {
    const int ta = min(max(t, 0), H), tx = t < 0 ? min(1 - t, H) : 1;
#pragma unroll
    for (int q = 0; q < kQ; ++q) Tv[q] = ld(v_main, q * plane + (size_t)ta * iper) - (ld(v_main, q * plane + (size_t)tx * iper) - ld(v_main, q * plane + iper));
    if (tail_wave) {
      const int tt = t0 + trr, tta = min(max(tt, 0), H), ttx = tt < 0 ? min(1 - tt, H) : 1;
      Tt = ld(tail_off(tta), 0) - (ld(tail_off(ttx), 0) - ld(tail_off(1), 0));
    }
  }
  const float C1 = (float)(0.01 * 0.01), C2 = (float)(0.03 * 0.03);
  double b0[kQ], bx[kQ], b0t = 0.0, bxt = 0.0;
  auto load_b = [&](int kw) __attribute__((always_inline)) {
    const int ws = wl.ws[kw], pad = ws / 2;
    const int i = t + pad, hi = i + ws - 1 - pad;                           // scalar
    const int rmain = min(max(hi, 0), H - 1) + 1;
    if (hi > H - 1 && i < H) {                                             // scalar branch around straight-line loads
      const int ry = 2 * (H - 1) - hi;
#pragma unroll
      for (int q = 0; q < kQ; ++q) {
        b0[q] = ld(v_main, q * plane + (size_t)rmain * iper);
        bx[q] = ld(v_main, q * plane + (size_t)(H - 1) * iper) - ld(v_main, q * plane + (size_t)ry * iper);
      }
    } else {
#pragma unroll
      for (int q = 0; q < kQ; ++q) { b0[q] = ld(v_main, q * plane + (size_t)rmain * iper); bx[q] = 0.0; }
    }
    if (tail_wave) {                                                       // per-lane rows: unconditional loads, the pair cancels where not needed
      const int ti = t0 + trr + pad, thi = ti + ws - 1 - pad;
      const bool over = thi > H - 1 && ti < H;
      b0t = ld(tail_off(min(max(thi, 0), H - 1) + 1), 0);
      bxt = ld(tail_off(H - 1), 0) - ld(tail_off(over ? 2 * (H - 1) - thi : H - 1), 0);
    }
  };
  load_b(0);
  for (int kw = 0; kw < g; ++kw) {
    const int ws = wl.ws[kw], pad = ws / 2;
    const int i = t + pad;                                                 // scalar: this wave's pixel row for this window size
    const bool row_ok = i >= 0 && i < H;
    const double dinv = wl.dinv[kw];
    double* const Dk = Dl + (kw & 1) * n_el;
    {
      double* const Dw = Dk + r * iper + c;
#pragma unroll
      for (int q = 0; q < kQ; ++q) Dw[q * rowq] = (b0[q] + bx[q]) - Tv[q];   // (rows that do not exist for this window size are never read)
      if (tail) Dk[tq * rowq + trr * iper + W] = (b0t + bxt) - Tt;
    }
    __syncthreads();
    // the previous window size's 16 wave sums -> one partial per (window size, pixel row): thread 0 of each row's first wave
    if (kw > 0 && (tid & (W - 1)) == 0) {
      const int ip = t + wl.ws[kw - 1] / 2;
      if (ip >= 0 && ip < H) {
        const double* w0 = wsum + ((kw - 1) & 1) * 16 + wave;
        double acc = 0.0;
        for (int k = 0; k < (W >> 6); ++k) acc += w0[k];
        partial[((size_t)img * kWsGroup + kw - 1) * H + ip] = acc;
      }
    }
    if (kw + 1 < g) load_b(kw + 1);
    float m_val = 0.f;
    if (row_ok) {
      const int lo = c - pad, hi = c + ws - 1 - pad;
      const double* const Dr = Dk + r * iper;
      const int c1 = min(hi, W - 1) + 1, c0 = max(lo, 0);
      double sum[kQ];
#pragma unroll
      for (int q = 0; q < kQ; ++q) sum[q] = Dr[q * rowq + c1] - Dr[q * rowq + c0];
      if (__builtin_amdgcn_ballot_w64(lo < 0) != 0) {
        const int ca = lo < 0 ? 1 - lo : 1;
#pragma unroll
        for (int q = 0; q < kQ; ++q) sum[q] += Dr[q * rowq + ca] - Dr[q * rowq + 1];
      }
      if (__builtin_amdgcn_ballot_w64(hi > W - 1) != 0) {
        const int cb = hi > W - 1 ? 2 * (W - 1) - hi : W - 1;
#pragma unroll
        for (int q = 0; q < kQ; ++q) sum[q] += Dr[q * rowq + W - 1] - Dr[q * rowq + cb];
      }
      const float mu1 = (float)(sum[0] * dinv), mu2 = (float)(sum[1] * dinv);
      const float mu1_sq = mu1 * mu1, mu2_sq = mu2 * mu2, mu12 = mu1 * mu2;
      const float s1 = (float)(sum[2] * dinv) - mu1_sq;
      const float s2 = (float)(sum[3] * dinv) - mu2_sq;
      const float s12 = (float)(sum[4] * dinv) - mu12;
      m_val = ((2.0f * mu12 + C1) * (2.0f * s12 + C2)) / ((mu1_sq + mu2_sq + C1) * (s1 + s2 + C2));
    }
    const double seg = wave_sum_map(m_val);
    if (lane == 0) wsum[(kw & 1) * 16 + wave] = seg;
  }
  __syncthreads();
  if ((tid & (W - 1)) == 0) {
    const int ip = t + wl.ws[g - 1] / 2;
    if (ip >= 0 && ip < H) {
      const double* w0 = wsum + ((g - 1) & 1) * 16 + wave;
      double acc = 0.0;
      for (int k = 0; k < (W >> 6); ++k) acc += w0[k];
      partial[((size_t)img * kWsGroup + g - 1) * H + ip] = acc;
    }
  }
}

// one workgroup per (image, window size of the group): thread i sums every 256th partial, then a fixed-order block sum
__global__ __launch_bounds__(256) void ssim_finish_kernel(const double* __restrict__ partial, double* __restrict__ out, int n_img, int n_grp,
                                                          int nblk, int out_stride, int out_col0, double inv_count) {
  const int img = blockIdx.x / n_grp, kw = blockIdx.x - img * n_grp;
  const double* p = partial + ((size_t)img * kWsGroup + kw) * nblk;
  double s = 0.0;
  for (int b = threadIdx.x; b < nblk; b += 256) s += p[b];
  __shared__ double red[256];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[(size_t)img * out_stride + out_col0 + kw] = red[0] * inv_count;
}

// mean((sr/255 - hr/255)^2) over all H*W*C values of an image: block partial sums (grid = blocks x images) ...
__global__ __launch_bounds__(256) void mse_partial_kernel(const uint8_t* __restrict__ sr, const uint8_t* __restrict__ hr,
                                                          double* __restrict__ partial, size_t n, int nb) {
  const int img = blockIdx.y;
  const uint8_t* a = sr + (size_t)img * n;
  const uint8_t* b = hr + (size_t)img * n;
  double local = 0.0;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)nb * 256) {
    const float d = (float)a[i] / 255.0f - (float)b[i] / 255.0f;
    local += (double)(d * d);
  }
  __shared__ double red[256];
  red[threadIdx.x] = local;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) partial[(size_t)img * nb + blockIdx.x] = red[0];
}
// ... and one thread per image: the mean as float32 (np.mean of a float32 array), PSNR with data_range 1
__global__ void mse_finish_kernel(const double* __restrict__ partial, double* __restrict__ mse, double* __restrict__ psnr, int n_img,
                                  int nb, double n) {
  const int img = blockIdx.x * blockDim.x + threadIdx.x;
  if (img >= n_img) return;
  double s = 0.0;
  for (int b = 0; b < nb; ++b) s += partial[(size_t)img * nb + b];
  const double m = (double)(float)(s / n);
  mse[img] = m;
  psnr[img] = m == 0.0 ? INFINITY : 10.0 * log10(1.0 / m);
}

__global__ void to_u8_hwc_kernel(const float* __restrict__ x, uint8_t* __restrict__ out, int B, int C, int HW, float mul) {
  const size_t total = (size_t)B * C * HW;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    const size_t pix = i / C;
    const int b = (int)(pix / HW);
    const int hw = (int)(pix - (size_t)b * HW);
    float v = x[((size_t)b * C + c) * HW + hw] * mul;
    v = fminf(fmaxf(v, 0.0f), 255.0f);            // clamp(0, 255); NaN -> 0 like torch's clamp+byte on ROCm is undefined
    out[i] = (uint8_t)v;                           // .byte(): truncation toward zero (evaluate.py:214-215)
  }
}

__global__ void quantize_kernel(const float* __restrict__ x, float* __restrict__ y, size_t n, float pr) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const float v = fminf(fmaxf(x[i] * pr, 0.0f), 255.0f);
    y[i] = rintf(v) / pr;                          // torch.round = half to even (trainer.py:45-47)
  }
}

__global__ __launch_bounds__(256) void l1_kernel(const float* __restrict__ a, const float* __restrict__ b, size_t n,
                                                 double* __restrict__ partial) {
  double local = 0.0;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    local += (double)fabsf(a[i] - b[i]);
  __shared__ double red[256];
  red[threadIdx.x] = local;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}
__global__ void l1_finish_kernel(const double* __restrict__ partial, int nb, double inv_n, double* __restrict__ out) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    double s = 0.0;
    for (int i = 0; i < nb; ++i) s += partial[i];
    *out = s * inv_n;
  }
}

// Validation metrics of Trainer.test (metrics.py:70-108): one workgroup per image.
__global__ __launch_bounds__(256) void val_metrics_kernel(const float* __restrict__ sr, const float* __restrict__ hr, int C,
                                                          int H, int W, float rgb_range, double* __restrict__ psnr,
                                                          double* __restrict__ ssim) {
  const int img = blockIdx.x;
  const size_t plane = (size_t)H * W;
  const float* s = sr + (size_t)img * C * plane;
  const float* h = hr + (size_t)img * C * plane;
  const int shave = 4;
  const bool do_shave = W > 2 * shave;                        // "if sr.size(-1) > 2 * shave"
  const int y0 = do_shave ? shave : 0, y1 = do_shave ? H - shave : H;
  const int x0 = do_shave ? shave : 0, x1 = do_shave ? W - shave : W;
  const int h2 = y1 - y0, w2 = x1 - x0;
  __shared__ double red[256];
  // ---- PSNR: diff = (sr - hr) / range over all channels of the shaved region ----
  double local = 0.0;
  for (int c = 0; c < C; ++c)
    for (int p = threadIdx.x; p < h2 * w2; p += 256) {
      const int yy = y0 + p / w2, xx = x0 + p % w2;
      const float d = (s[c * plane + (size_t)yy * W + xx] - h[c * plane + (size_t)yy * W + xx]) / rgb_range;
      local += (double)(d * d);
    }
  red[threadIdx.x] = local;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const double m = (double)(float)(red[0] / ((double)C * h2 * w2));
    psnr[img] = m == 0.0 ? INFINITY : 10.0 * log10(1.0 / m);
  }
  __syncthreads();
  // ---- SSIM: clamp(x/range, 0, 1), luminance if C > 1, 11x11 box with ZERO padding, C1/C2 * 255^2 ----
  auto lum = [&](const float* t, int yy, int xx) -> float {
    if (yy < y0 || yy >= y1 || xx < x0 || xx >= x1) return 0.0f;           // zero padding of the shaved image
    if (C == 1) return fminf(fmaxf(t[(size_t)yy * W + xx] / rgb_range, 0.0f), 1.0f);
    const float cv[3] = {65.738f / 256.0f, 129.057f / 256.0f, 25.064f / 256.0f};
    float acc = 0.0f;
    for (int c = 0; c < 3; ++c) acc += fminf(fmaxf(t[c * plane + (size_t)yy * W + xx] / rgb_range, 0.0f), 1.0f) * cv[c];
    return acc;
  };
  const float C1 = (float)(0.01 * 0.01 * 255.0 * 255.0), C2 = (float)(0.03 * 0.03 * 255.0 * 255.0);
  const int win = 11, pad = 5;
  local = 0.0;
  for (int p = threadIdx.x; p < h2 * w2; p += 256) {
    const int yy = y0 + p / w2, xx = x0 + p % w2;
    double a1 = 0, a2 = 0, a11 = 0, a22 = 0, a12 = 0;
    for (int dy = -pad; dy <= pad; ++dy)
      for (int dx = -pad; dx <= pad; ++dx) {
        const float u = lum(s, yy + dy, xx + dx), v = lum(h, yy + dy, xx + dx);
        a1 += u; a2 += v; a11 += (double)(u * u); a22 += (double)(v * v); a12 += (double)(u * v);
      }
    const double k = 1.0 / (win * win);
    const float mu1 = (float)(a1 * k), mu2 = (float)(a2 * k);
    const float s1 = (float)(a11 * k) - mu1 * mu1, s2 = (float)(a22 * k) - mu2 * mu2, s12 = (float)(a12 * k) - mu1 * mu2;
    const float m = ((2 * mu1 * mu2 + C1) * (2 * s12 + C2)) / ((mu1 * mu1 + mu2 * mu2 + C1) * (s1 + s2 + C2));
    local += (double)m;
  }
  red[threadIdx.x] = local;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) ssim[img] = red[0] / ((double)h2 * w2);
}

int chunk_images(int n_img, int H, int W) {
  const size_t per = (size_t)(H + 1) * (W + 1);
  const size_t cap = (size_t)1 << 25;                  // table points per chunk (x 40 bytes)
  size_t c = cap / per;
  if (c < 1) c = 1;
  return (int)std::min<size_t>(c, (size_t)n_img);
}
// widths the LDS-staged sweep takes (ssim_rows_lds_kernel), and the partial sums per (image, window size) either kernel writes
inline bool lds_sweep_ok(int H, int W) {
  return W >= 64 && W <= 1024 && (W & (W - 1)) == 0 && (long long)kQ * (H + 1) * (W + 1) * 8 < (1ll << 31) && getenv("SRAD_SCORE_NO_LDS") == nullptr;
}
inline int partial_slots(int H, int W) { return lds_sweep_ok(H, W) ? H : (H * W + kEvalPix - 1) / kEvalPix; }
inline int grid1d(size_t total) {
  size_t b = (total + 255) / 256;
  return (int)std::min<size_t>(std::max<size_t>(b, 1), 4096);
}

}  // namespace

extern "C" {

int srad_to_u8_hwc(const float* x, int B, int C, int H, int W, float rgb_range, uint8_t* out, void* stream) {
  SRAD_REQUIRE(x && out && B > 0 && C > 0 && H > 0 && W > 0, "to_u8_hwc: bad argument");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const size_t total = (size_t)B * C * H * W;
  SradProfScope prof(s, SRAD_K_SCORE, (double)total, 5.0 * total);
  hipLaunchKernelGGL(to_u8_hwc_kernel, dim3(grid1d(total)), dim3(256), 0, s, x, out, B, C, H * W, 255.0f / rgb_range);
  SRAD_CHECK_HIP(hipGetLastError());
  return SRAD_OK;
}

int srad_quantize(const float* x, float* y, int64_t n, float rgb_range, void* stream) {
  SRAD_REQUIRE(x && y && n > 0, "quantize: bad argument");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(quantize_kernel, dim3(grid1d((size_t)n)), dim3(256), 0, s, x, y, (size_t)n, 255.0f / rgb_range);
  SRAD_CHECK_HIP(hipGetLastError());
  return SRAD_OK;
}

int srad_score_workspace_bytes(int n_img, int H, int W, size_t* bytes) {
  SRAD_REQUIRE(bytes && n_img > 0 && H > 0 && W > 0, "score_workspace_bytes: bad argument");
  const int chunk = chunk_images(n_img, H, W);
  const int nblk = partial_slots(H, W);
  const int nseg = (H + kSeg - 1) / kSeg;
  *bytes = srad_align_up((size_t)chunk * (H + 1) * (W + 1) * kQ * sizeof(double), 256) +
           srad_align_up((size_t)chunk * kWsGroup * nblk * sizeof(double), 256) +
           srad_align_up((size_t)chunk * kQ * nseg * (W + 1) * sizeof(double), 256);
  return SRAD_OK;
}

int srad_score_pairs(const uint8_t* sr, const uint8_t* hr, int n_img, int H, int W, int C, const int32_t* ws_host,
                     int n_ws, double* ssim_out, double* mse_out, double* psnr_out, void* workspace,
                     size_t workspace_bytes, void* stream) {
  SRAD_REQUIRE(sr && hr && workspace && n_img > 0 && H > 1 && W > 1, "score_pairs: bad argument");
  SRAD_REQUIRE(C == 1 || C == 3, "score_pairs: channels must be 1 or 3 (got %d)", C);
  SRAD_REQUIRE((long long)(H + 1) * (W + 1) < (1ll << 31), "score_pairs: %dx%d images are too large for the 32-bit table offsets", H, W);
  SRAD_REQUIRE(n_ws == 0 || (ws_host && ssim_out), "score_pairs: window list / output missing");
  size_t need = 0;
  SRAD_TRY(srad_score_workspace_bytes(n_img, H, W, &need));
  SRAD_REQUIRE(workspace_bytes >= need, "score_pairs: workspace %zu bytes, %zu needed", workspace_bytes, need);
  for (int k = 0; k < n_ws; ++k)
    SRAD_REQUIRE(ws_host[k] >= 1 && ws_host[k] / 2 < H && ws_host[k] / 2 < W && ws_host[k] - 1 - ws_host[k] / 2 < H &&
                     ws_host[k] - 1 - ws_host[k] / 2 < W,
                 "score_pairs: window %d needs more than one reflection of a %dx%d image", ws_host[k], H, W);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const int chunk = chunk_images(n_img, H, W);
  const int nblk = partial_slots(H, W);
  double* sat = reinterpret_cast<double*>(workspace);
  double* partial = reinterpret_cast<double*>(reinterpret_cast<char*>(workspace) +
                                              srad_align_up((size_t)chunk * (H + 1) * (W + 1) * kQ * sizeof(double), 256));
  double* segtot = reinterpret_cast<double*>(reinterpret_cast<char*>(partial) + srad_align_up((size_t)chunk * kWsGroup * nblk * sizeof(double), 256));
  const int nseg = (H + kSeg - 1) / kSeg;
  const size_t img_bytes = (size_t)H * W * C;
  const bool want_mse = mse_out && psnr_out;
  for (int i0 = 0; i0 < n_img && (n_ws > 0 || want_mse); i0 += chunk) {
    const int n = std::min(chunk, n_img - i0);
    const uint8_t* srp = sr + (size_t)i0 * img_bytes;
    const uint8_t* hrp = hr + (size_t)i0 * img_bytes;
    if (want_mse) {                                   // block partials in the (not yet used) SSIM partial area: nb <= kWsGroup * nblk
      const int nb = std::min(64, nblk);
      SradProfScope prof(s, SRAD_K_SCORE, 3.0 * n * img_bytes, 2.0 * n * img_bytes);
      hipLaunchKernelGGL(mse_partial_kernel, dim3(nb, n), dim3(256), 0, s, srp, hrp, partial, img_bytes, nb);
      hipLaunchKernelGGL(mse_finish_kernel, dim3((n + 63) / 64), dim3(64), 0, s, partial, mse_out + i0, psnr_out + i0, n, nb, (double)img_bytes);
    }
    if (n_ws == 0) continue;
    {
      SradProfScope prof(s, SRAD_K_SCORE, 10.0 * n * H * W, 2.0 * n * img_bytes + 40.0 * n * (H + 1) * (W + 1));
      hipLaunchKernelGGL(sat_rows_kernel, dim3((n * (H + 1) + 3) / 4), dim3(256), 0, s, srp, hrp, sat, n, H, W, C);
    }
    {
      const size_t t = (size_t)n * kQ * nseg * (W + 1);
      SradProfScope prof(s, SRAD_K_SCORE, 1.0 * n * (H + 1) * (W + 1) * kQ, 80.0 * n * (H + 1) * (W + 1));
      hipLaunchKernelGGL(sat_cols_local_kernel, dim3((unsigned)((t + 255) / 256)), dim3(256), 0, s, sat, segtot, n * kQ, H, W, nseg);
      if (nseg > 1) {
        const size_t t2 = (size_t)n * kQ * (nseg - 1) * (W + 1);
        hipLaunchKernelGGL(sat_cols_carry_kernel, dim3((unsigned)((t2 + 255) / 256)), dim3(256), 0, s, sat, segtot, n * kQ, H, W, nseg);
      }
    }
    for (int k0 = 0; k0 < n_ws; k0 += kWsGroup) {      // up to kWsGroup window sizes per launch
      const int g = std::min(kWsGroup, n_ws - k0);
      WsList wl{};
      for (int k = 0; k < g; ++k) { wl.ws[k] = (int)ws_host[k0 + k]; wl.dinv[k] = 1.0 / ((double)wl.ws[k] * (double)wl.ws[k]); }
      {
        // algorithmic bytes per (pair, window): the two fp32 luminance planes read once
        SradProfScope prof(s, SRAD_K_SCORE, 40.0 * n * H * W * g, 8.0 * n * H * W * g);
        static const bool generic = getenv("SRAD_SCORE_GENERIC") != nullptr;     // A/B: the per-lane kernel for every width
        if (lds_sweep_ok(H, W)) {
          int pmin = wl.ws[0] / 2, pmax = pmin;
          for (int k = 1; k < g; ++k) { pmin = std::min(pmin, wl.ws[k] / 2); pmax = std::max(pmax, wl.ws[k] / 2); }
          const int rb = 1024 / W, t_first = -pmax, nt = H - 1 - pmin - t_first + 1, nt_blocks = (nt + rb - 1) / rb;
          const size_t lds = ((size_t)2 * kQ * rb * (W + 1) + 32) * sizeof(double);
          static SradOncePerDevice configured;
          if (configured.need()) {
            SRAD_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(ssim_rows_lds_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
            configured.done();
          }
          int lw = 0;
          while ((1 << lw) < W) ++lw;
          hipLaunchKernelGGL(ssim_rows_lds_kernel, dim3((unsigned)((size_t)nt_blocks * n)), dim3(1024), lds, s, sat, partial, H, lw, wl, g, t_first, nt_blocks);
        } else if (W % 64 == 0 && !generic)
          hipLaunchKernelGGL(ssim_eval_kernel<true>, dim3((unsigned)((size_t)nblk * g * n)), dim3(256), 0, s, sat, partial, H, W, wl, nblk, g, n);
        else
          hipLaunchKernelGGL(ssim_eval_kernel<false>, dim3((unsigned)((size_t)nblk * g * n)), dim3(256), 0, s, sat, partial, H, W, wl, nblk, g, n);
      }
      hipLaunchKernelGGL(ssim_finish_kernel, dim3(n * g), dim3(256), 0, s, partial, ssim_out + (size_t)i0 * n_ws, n, g,
                         nblk, n_ws, k0, 1.0 / ((double)H * W));
    }
  }
  SRAD_CHECK_HIP(hipGetLastError());
  return SRAD_OK;
}

int srad_val_metrics(const float* sr, const float* hr, int B, int C, int H, int W, float rgb_range, double* psnr_out,
                     double* ssim_out, void* workspace, size_t workspace_bytes, void* stream) {
  (void)workspace; (void)workspace_bytes;
  SRAD_REQUIRE(sr && hr && psnr_out && ssim_out && B > 0, "val_metrics: bad argument");
  SRAD_REQUIRE(C == 1 || C == 3, "val_metrics: channels must be 1 or 3 (got %d)", C);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(val_metrics_kernel, dim3(B), dim3(256), 0, s, sr, hr, C, H, W, rgb_range, psnr_out, ssim_out);
  SRAD_CHECK_HIP(hipGetLastError());
  return SRAD_OK;
}

// sklearn.metrics.roc_auc_score for binary labels == Mann-Whitney U with ties counted one half
// (reference call sites src/evaluate.py:245,263-265).  Host arithmetic: n is the number of test images.
int srad_roc_auc(const int32_t* labels, const double* scores, int n, double* auc) {
  SRAD_REQUIRE(labels && scores && auc && n > 0, "roc_auc: bad argument");
  std::vector<int> idx(n);
  for (int i = 0; i < n; ++i) idx[i] = i;
  std::stable_sort(idx.begin(), idx.end(), [&](int a, int b) { return scores[a] < scores[b]; });
  double npos = 0, nneg = 0, rank_sum = 0;
  for (int i = 0; i < n; ++i) (labels[i] ? npos : nneg) += 1;
  if (npos == 0 || nneg == 0)
    return srad_set_error(SRAD_ERR_ARG, "Only one class present in y_true. ROC AUC score is not defined in that case.");
  int i = 0;
  while (i < n) {
    int j = i;
    while (j + 1 < n && scores[idx[j + 1]] == scores[idx[i]]) ++j;
    const double r = 0.5 * (i + j) + 1.0;            // average rank of the tie group
    for (int k = i; k <= j; ++k)
      if (labels[idx[k]]) rank_sum += r;
    i = j + 1;
  }
  *auc = (rank_sum - npos * (npos + 1) / 2.0) / (npos * nneg);
  return SRAD_OK;
}

int srad_l1_workspace_bytes(size_t* bytes) {
  SRAD_REQUIRE(bytes, "l1_workspace_bytes: null");
  *bytes = 1024 * sizeof(double);
  return SRAD_OK;
}

// mean |a - b| -> *out (device double).  workspace: >= srad_l1_workspace_bytes().
int srad_l1_loss(const float* a, const float* b, int64_t n, double* out, void* workspace, void* stream) {
  SRAD_REQUIRE(a && b && out && workspace && n > 0, "l1_loss: bad argument");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const int nb = (int)std::min<size_t>(1024, ((size_t)n + 255) / 256);
  double* partial = reinterpret_cast<double*>(workspace);
  hipLaunchKernelGGL(l1_kernel, dim3(nb), dim3(256), 0, s, a, b, (size_t)n, partial);
  hipLaunchKernelGGL(l1_finish_kernel, dim3(1), dim3(64), 0, s, partial, nb, 1.0 / (double)n, out);
  SRAD_CHECK_HIP(hipGetLastError());
  return SRAD_OK;
}

}  // extern "C"
