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

struct Seg { int a, b; };   // inclusive index range of the unpadded image

__device__ __forceinline__ int reflect_segments(int lo, int hi, int n, Seg* s) {
  // padded coordinates lo..hi (may leave [0, n-1] on either side by less than n): numpy "reflect"
  int k = 0;
  s[k++] = Seg{max(lo, 0), min(hi, n - 1)};
  if (lo < 0) s[k++] = Seg{1, -lo};
  if (hi > n - 1) s[k++] = Seg{2 * (n - 1) - hi, n - 2};
  return k;
}

// grid (pixel blocks, window sizes of the group, images): SSIM map values, block partial sums (float64).  Up to kWsGroup
// window sizes share one launch (the list travels as a kernel argument).
constexpr int kWsGroup = 16;
struct WsList { int ws[kWsGroup]; };
__global__ __launch_bounds__(256) void ssim_eval_kernel(const double* __restrict__ sat, double* __restrict__ partial,
                                                        int H, int W, const WsList wl, int nblk) {
  const int kw = blockIdx.y, img = blockIdx.z;     // an image's window sizes run back to back: its tables stay in L2
  const int ws = wl.ws[kw];
  const size_t per = (size_t)(W + 1), plane = (size_t)(H + 1) * per;
  const int iper = W + 1;
  const double* S = sat + (size_t)img * kQ * plane;
  const int pad = ws / 2;
  const double dinv = 1.0 / (double)(ws * ws);
  const float C1 = (float)(0.01 * 0.01), C2 = (float)(0.03 * 0.03);
  double local = 0.0;
  for (int k = 0; k < kEvalPix / 256; ++k) {
    const int pix = blockIdx.x * kEvalPix + k * 256 + threadIdx.x;
    if (pix >= H * W) continue;
    const int i = pix / W, j = pix - i * W;
    const int rlo = i - pad, rhi = i + ws - 1 - pad, clo = j - pad, chi = j + ws - 1 - pad;
    double sum[kQ] = {0, 0, 0, 0, 0};
    auto rect = [&](int ra, int rb, int ca, int cb) {          // rows ra..rb, columns ca..cb of the unpadded image (32-bit offsets:
      const int o11 = (rb + 1) * iper + cb + 1, o01 = ra * iper + cb + 1, o10 = (rb + 1) * iper + ca, o00 = ra * iper + ca;   // a chunk's tables are < 2^31 doubles)
#pragma unroll
      for (int q = 0; q < kQ; ++q) {
        const double* Sq = S + (size_t)q * plane;
        sum[q] += (Sq[o11] - Sq[o01]) - (Sq[o10] - Sq[o00]);
      }
    };
    if (rlo >= 0 && rhi < H && clo >= 0 && chi < W) {          // the window lies inside the image: one rectangle
      rect(rlo, rhi, clo, chi);
    } else {
      Seg rs[3], cs[3];
      const int nr = reflect_segments(rlo, rhi, H, rs);
      const int nc = reflect_segments(clo, chi, W, cs);
      for (int a = 0; a < nr; ++a)
        for (int b = 0; b < nc; ++b) rect(rs[a].a, rs[a].b, cs[b].a, cs[b].b);
    }
    const float mu1 = (float)(sum[0] * dinv), mu2 = (float)(sum[1] * dinv);
    const float mu1_sq = mu1 * mu1, mu2_sq = mu2 * mu2, mu12 = mu1 * mu2;
    const float s1 = (float)(sum[2] * dinv) - mu1_sq;
    const float s2 = (float)(sum[3] * dinv) - mu2_sq;
    const float s12 = (float)(sum[4] * dinv) - mu12;
    const float m = ((2.0f * mu12 + C1) * (2.0f * s12 + C2)) / ((mu1_sq + mu2_sq + C1) * (s1 + s2 + C2));
    local += (double)m;
  }
  // deterministic block reduction
  __shared__ double red[256];
  red[threadIdx.x] = local;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) partial[((size_t)img * kWsGroup + kw) * nblk + blockIdx.x] = red[0];
}

// one wave per (image, window size of the group): lanes sum every 64th block partial, then a fixed-order wave sum
__global__ __launch_bounds__(256) void ssim_finish_kernel(const double* __restrict__ partial, double* __restrict__ out, int n_img, int n_grp,
                                                          int nblk, int out_stride, int out_col0, double inv_count) {
  const int lane = threadIdx.x & 63;
  const int t = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (t >= n_img * n_grp) return;
  const int img = t / n_grp, kw = t - img * n_grp;
  const double* p = partial + ((size_t)img * kWsGroup + kw) * nblk;
  double s = 0.0;
  for (int b = lane; b < nblk; b += 64) s += p[b];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  if (lane == 0) out[(size_t)img * out_stride + out_col0 + kw] = s * inv_count;
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
  const int nblk = (H * W + kEvalPix - 1) / kEvalPix;
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
  const int nblk = (H * W + kEvalPix - 1) / kEvalPix;
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
      for (int k = 0; k < g; ++k) wl.ws[k] = (int)ws_host[k0 + k];
      {
        // algorithmic bytes per (pair, window): the two fp32 luminance planes read once
        SradProfScope prof(s, SRAD_K_SCORE, 40.0 * n * H * W * g, 8.0 * n * H * W * g);
        hipLaunchKernelGGL(ssim_eval_kernel, dim3(nblk, g, n), dim3(256), 0, s, sat, partial, H, W, wl, nblk);
      }
      hipLaunchKernelGGL(ssim_finish_kernel, dim3((n * g + 3) / 4), dim3(256), 0, s, partial, ssim_out + (size_t)i0 * n_ws, n, g,
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
