// kernels_thin.hip - 3x3 stride-1 convolutions with very few INPUT channels: DRN's head (3 -> F, src/drn.py:247) and the data
// gradients of its tail convolutions (3 -> F 2^p, drn.py:256-258, 265-267) in training.  On the tiled MFMA GEMM these pay for K
// padded to 32 per tap (8x the MACs) at the largest pixel count of the model: the head took 149 us at 256 px x 8 images against
// ~15 us of memory time; here 31 us.
// Here: one output pixel per thread, fp32 FMAs against weights held in LDS (every lane reads the same address: a broadcast), the
// input straight from global memory (each pixel is read by nine neighbouring threads: L1 / L2 hits).  Weights are the bf16
// values of the packed layer (what the MFMA path multiplies by), activations stay fp32.
#include "srad_common.h"
#include <algorithm>

namespace {

template <int N4>      // output channels = 4 * N4 (1: tails, 5: head of the x2 / x4 presets, 10 / 20: the tails' data gradients)
__global__ __launch_bounds__(256) void conv_thin_kernel(const GemmParams p) {
  extern __shared__ __attribute__((aligned(16))) float wl[];      // [9][Cin][4 N4] weights, then [4 N4] bias
  constexpr int NP = 4 * N4;
  const int Cin = p.Cin, c4n = Cin >> 2, tid = threadIdx.x;
  {
    const __bf16* const Wp = reinterpret_cast<const __bf16*>(p.Wp);
    for (int i = tid; i < 9 * Cin * NP; i += 256) {
      const int n = i % NP, c = (i / NP) % Cin, tap = i / (NP * Cin);
      wl[i] = (float)Wp[((size_t)n * 9 + tap) * p.Cp + c];          // rows beyond the layer's N are zeros in the pack
    }
    if (tid < NP) wl[9 * Cin * NP + tid] = (p.bias && tid < p.N) ? p.bias[tid] : 0.f;
  }
  __syncthreads();
  const int H = p.Hi, W = p.Wi;
  const size_t pix = (size_t)blockIdx.x * 256 + tid;
  if (pix >= (size_t)p.M) return;
  const int x = (int)(pix % W), y = (int)((pix / W) % H), b = (int)(pix / ((size_t)W * H));
  f32x4 acc[N4];
#pragma unroll
  for (int j = 0; j < N4; ++j) acc[j] = *reinterpret_cast<const f32x4*>(wl + 9 * Cin * NP + 4 * j);
#pragma unroll
  for (int tap = 0; tap < 9; ++tap) {
    const int yy = y + tap / 3 - 1, xx = x + tap % 3 - 1;
    const bool in = yy >= 0 && yy < H && xx >= 0 && xx < W;
    const float* const src = p.X + ((size_t)(b * H + min(max(yy, 0), H - 1)) * W + min(max(xx, 0), W - 1)) * p.ldx;   // clamped: no load behind a branch
    const float* const wt = wl + tap * Cin * NP;
#pragma unroll 2
    for (int c4 = 0; c4 < c4n; ++c4) {
      f32x4 xv = *reinterpret_cast<const f32x4*>(src + 4 * c4);
      if (!in) xv = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float* const wr = wt + (4 * c4 + e) * NP;
#pragma unroll
        for (int j = 0; j < N4; ++j) acc[j] += *reinterpret_cast<const f32x4*>(wr + 4 * j) * xv[e];
      }
    }
  }
  // float4 stores when the output rows allow (N, row stride, offset multiples of 4); else element by element (RGB tails: N = 3)
  const bool vec = ((p.N | p.ldy | p.yoff | (p.R ? p.ldr : 0)) & 3) == 0 && (((uintptr_t)p.Y | (uintptr_t)p.R) & 15) == 0;
#pragma unroll
  for (int j = 0; j < N4; ++j) {
    if (4 * j >= p.N) break;
    f32x4 v = acc[j];
    if (p.act == SRAD_ACT_RELU) {
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
    } else if (p.act == SRAD_ACT_LRELU) {
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : v[e] * p.slope;
    }
    v = v * p.alpha;
    if (vec) {
      if (p.R) v += *reinterpret_cast<const f32x4*>(p.R + pix * p.ldr + 4 * j);
      *reinterpret_cast<f32x4*>(p.Y + pix * p.ldy + p.yoff + 4 * j) = v;
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (4 * j + e < p.N) p.Y[pix * p.ldy + p.yoff + 4 * j + e] = v[e] + (p.R ? p.R[pix * p.ldr + 4 * j + e] : 0.f);
    }
  }
}

// ---- the tails: 40 / 80 input channels -> <= 4 outputs (RGB) at the model's largest pixel counts.  On the tiled GEMM the im2col
// through LDS made these 120 / 45 / 16 us launches (256 / 128 / 64 px x 8 images) for 84 / 42 / 10 MB of input; here 62 / 33 / 12.
// (What is left is the texture path: every pixel's channels are loaded nine times, 27 / 45 load instructions per 16 pixels.
// Trimming the vector instructions per chunk - one clamped index per tap, selects on packed words - bought 5 %, a third wave
// per SIMD nothing.  conv_tail40_strip_kernel below loads each pixel three times instead: 62 -> 34 us at 40 channels.)  Here a wave owns 16 consecutive
// pixels of an image row and runs one 16x16x32 MFMA per (tap, 32-channel chunk) with BOTH operands straight from registers: the
// weight fragments of all 9 x ceil(Cin / 32) chunks stay in the wave's registers for its lifetime (rows >= N of the pack are
// zeros), a pixel fragment is the lane's own 8 channels of pixel (y + dy, x + dx) loaded from global memory (two float4, rounded
// to bf16 as the GEMM's staging rounds them, zeros outside the image) - no LDS, no barrier.  Results land transposed: the lanes
// 0 .. 15 hold outputs 0 .. 3 of their pixel.
template <int CIN>
__global__ __launch_bounds__(256) void conv_tail_kernel(const GemmParams p, int tiles) {
  constexpr int CPT = (CIN + 31) / 32, CP = CPT * 32;
  const int lane = threadIdx.x & 63, fr = lane & 15, fq = lane >> 4;
  const int gw = blockIdx.x * 4 + (threadIdx.x >> 6), nw = gridDim.x * 4;
  // k slots of a 32-channel chunk: lane group fq holds channels [4 fq, 4 fq + 4) and [16 + 4 fq, 16 + 4 fq + 4) - for BOTH operands,
  // so the product is unchanged - and each of the two float4 loads of a pixel fragment reads 64 contiguous bytes per pixel
  // across its four lane groups (the natural 8 fq .. 8 fq + 7 slots read 16-byte pieces at a 32-byte stride: 72.7 -> 65.2 us at 256 px x 8)
  bf16x8 wf[9][CPT];
  {
    const __bf16* const Wp = reinterpret_cast<const __bf16*>(p.Wp) + (size_t)fr * 9 * CP + 4 * fq;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
      for (int j = 0; j < CPT; ++j) {
        const bf16x4 lo = *reinterpret_cast<const bf16x4*>(Wp + tap * CP + 32 * j);
        const bf16x4 hi = *reinterpret_cast<const bf16x4*>(Wp + tap * CP + 32 * j + 16);
        wf[tap][j] = bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
      }
  }
  f32x4 bias4 = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int r = 0; r < 4; ++r) bias4[r] = (p.bias && fq == 0) ? p.bias[min(r, p.N - 1)] : 0.f;
  const int H = p.Hi, W = p.Wi, tpr = W >> 4;
  typedef f32x4 pix_t[CPT][2];
  for (int t = gw; t < tiles; t += nw) {
    const int row = t / tpr, x = ((t - row * tpr) << 4) + fr;
    const int b = row / H, y = row - b * H;
    // per tile, not per tap: the pixel index (a tap adds a constant; clamped into the tensor with one med3 - a tap outside the
    // image then reads some other pixel, which its mask drops) and the four border masks
    const int pix0 = row * W + x;
    const bool up = y > 0, dn = y < H - 1, lf = x > 0, rt = x < W - 1;
    auto load_tap = [&](int tap, pix_t& v) __attribute__((always_inline)) {
      const int pidx = min(max(pix0 + (tap / 3 - 1) * W + (tap % 3 - 1), 0), p.M - 1);
      const float* const src = p.X + (size_t)pidx * p.ldx;
#pragma unroll
      for (int j = 0; j < CPT; ++j) {
        v[j][0] = *reinterpret_cast<const f32x4*>(src + min(32 * j + 4 * fq, CIN - 4));
        if (32 * j + 16 < CIN) v[j][1] = *reinterpret_cast<const f32x4*>(src + min(32 * j + 16 + 4 * fq, CIN - 4));   // (else: no lane's slot is a channel)
      }
    };
    auto mul_tap = [&](int tap, const pix_t& v, f32x4 acc) __attribute__((always_inline)) {
      const bool in = (tap / 3 == 0 ? up : (tap / 3 == 2 ? dn : true)) && (tap % 3 == 0 ? lf : (tap % 3 == 2 ? rt : true));
#pragma unroll
      for (int j = 0; j < CPT; ++j) {
        const bool live0 = in && 32 * j + 4 * fq < CIN;      // (the weight's pad columns are zeros, but 0 * garbage need not be)
        const bool live1 = in && 32 * j + 16 + 4 * fq < CIN;
        // round first, then select on the packed words (4 selects per chunk instead of 8)
        bf16x4 lo;
#pragma unroll
        for (int e = 0; e < 4; ++e) lo[e] = (__bf16)v[j][0][e];
        u32x2 l = __builtin_bit_cast(u32x2, lo), h = u32x2{0u, 0u};
        l[0] = live0 ? l[0] : 0u; l[1] = live0 ? l[1] : 0u;
        if (32 * j + 16 < CIN) {
          bf16x4 hi;
#pragma unroll
          for (int e = 0; e < 4; ++e) hi[e] = (__bf16)v[j][1][e];
          h = __builtin_bit_cast(u32x2, hi);
          h[0] = live1 ? h[0] : 0u; h[1] = live1 ? h[1] : 0u;
        }
        const bf16x8 a = __builtin_bit_cast(bf16x8, u32x4{l[0], l[1], h[0], h[1]});
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[tap][j], a, acc, 0, 0, 0);
      }
      return acc;
    };
    f32x4 acc = bias4;
    pix_t va, vb;
    load_tap(0, va);
#pragma unroll
    for (int tap = 0; tap < 9; tap += 2) {                   // the next tap's pixels in flight under this tap's conversions + MFMAs
      if (tap + 1 < 9) load_tap(tap + 1, vb);
      acc = mul_tap(tap, va, acc);
      if (tap + 2 < 9) load_tap(tap + 2, va);
      if (tap + 1 < 9) acc = mul_tap(tap + 1, vb, acc);
    }
    if (fq == 0) {
      const size_t pix = (size_t)pix0;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float o = acc[r];
        if (p.act == SRAD_ACT_RELU) o = fmaxf(o, 0.f);
        else if (p.act == SRAD_ACT_LRELU) o = o > 0.f ? o : o * p.slope;
        o *= p.alpha;
        if (r < p.N) p.Y[pix * p.ldy + p.yoff + r] = o + (p.R ? p.R[pix * p.ldr + r] : 0.f);
      }
    }
  }
}

// The 40-channel tail as a column strip: a wave owns 16 columns x 16 rows and walks down, holding the three input rows an output
// row needs as bf16 fragments (three x-shifts each), so every pixel's channels are loaded three times instead of nine
// (conv_tail_kernel is bound by its load instructions, not by bytes or MFMAs).  Same k-slot permutation, same weights in
// registers; the next input row's nine loads are in flight under the current row's 18 MFMAs.  (At 80 channels three rows of
// fragments do not fit next to the weights.)
constexpr int CT_RS = 16;
__global__ __launch_bounds__(256) void conv_tail40_strip_kernel(const GemmParams p, int nstrips) {
  constexpr int CIN = 40, CP = 64;
  const int lane = threadIdx.x & 63, fr = lane & 15, fq = lane >> 4;
  const int gw = blockIdx.x * 4 + (threadIdx.x >> 6), nw = gridDim.x * 4;
  bf16x8 wf[9][2];
  {
    const __bf16* const Wp = reinterpret_cast<const __bf16*>(p.Wp) + (size_t)fr * 9 * CP + 4 * fq;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const bf16x4 lo = *reinterpret_cast<const bf16x4*>(Wp + tap * CP + 32 * j);
        const bf16x4 hi = *reinterpret_cast<const bf16x4*>(Wp + tap * CP + 32 * j + 16);
        wf[tap][j] = bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
      }
  }
  f32x4 bias4 = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int r = 0; r < 4; ++r) bias4[r] = (p.bias && fq == 0) ? p.bias[min(r, p.N - 1)] : 0.f;
  const int H = p.Hi, W = p.Wi, spr = W >> 4, spi = (H / CT_RS) * spr;      // strips per row of strips / per image
  const bool c1live = 32 + 4 * fq < CIN;                                     // chunk 1: channels 32 .. 39 in the lo slots of fq < 2
  struct Frag { u32x4 c0; u32x2 c1; };                                       // one pixel fragment: chunk 0 (8 slots), chunk 1 (lo 4 slots)
  struct Stage { f32x4 a, b, c; };                                           // the three float4 it is made from
  for (int st = gw; st < nstrips; st += nw) {
    const int b = st / spi, rem = st - b * spi, sy = rem / spr, x = ((rem - sy * spr) << 4) + fr;
    const int y0 = sy * CT_RS;
    const bool lf = x > 0, rt = x < W - 1;
    auto load_row = [&](int r, Stage (&sg)[3]) __attribute__((always_inline)) {
      const int base = (b * H + r) * W + x;
#pragma unroll
      for (int dx = 0; dx < 3; ++dx) {
        const float* const src = p.X + (size_t)min(max(base + dx - 1, 0), p.M - 1) * p.ldx;      // (rows -1 / H, columns -1 / W: masked below)
        sg[dx].a = *reinterpret_cast<const f32x4*>(src + 4 * fq);
        sg[dx].b = *reinterpret_cast<const f32x4*>(src + 16 + 4 * fq);
        sg[dx].c = *reinterpret_cast<const f32x4*>(src + min(32 + 4 * fq, CIN - 4));
      }
    };
    auto convert_row = [&](int r, const Stage (&sg)[3], Frag (&f)[3]) __attribute__((always_inline)) {
      const bool rowin = r >= 0 && r < H;
#pragma unroll
      for (int dx = 0; dx < 3; ++dx) {
        const bool in = rowin && (dx == 0 ? lf : (dx == 2 ? rt : true));
        bf16x4 a, bq, c;
#pragma unroll
        for (int e = 0; e < 4; ++e) { a[e] = (__bf16)sg[dx].a[e]; bq[e] = (__bf16)sg[dx].b[e]; c[e] = (__bf16)sg[dx].c[e]; }
        const u32x2 ua = __builtin_bit_cast(u32x2, a), ub = __builtin_bit_cast(u32x2, bq), uc = __builtin_bit_cast(u32x2, c);
        f[dx].c0 = u32x4{in ? ua[0] : 0u, in ? ua[1] : 0u, in ? ub[0] : 0u, in ? ub[1] : 0u};
        f[dx].c1 = u32x2{(in && c1live) ? uc[0] : 0u, (in && c1live) ? uc[1] : 0u};
      }
    };
    Frag F[3][3];
    Stage sg[3];
    load_row(y0 - 1, sg);
    convert_row(y0 - 1, sg, F[0]);
    load_row(y0, sg);
    convert_row(y0, sg, F[1]);
    load_row(y0 + 1, sg);
#pragma unroll
    for (int i = 0; i < CT_RS; ++i) {
      convert_row(y0 + i + 1, sg, F[(i + 2) % 3]);
      if (i + 1 < CT_RS) load_row(y0 + i + 2, sg);             // in flight under this row's MFMAs
      f32x4 acc = bias4;
#pragma unroll
      for (int dy = 0; dy < 3; ++dy)
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
          const Frag& f = F[(i + dy) % 3][dx];
          acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[dy * 3 + dx][0], __builtin_bit_cast(bf16x8, f.c0), acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[dy * 3 + dx][1], __builtin_bit_cast(bf16x8, u32x4{f.c1[0], f.c1[1], 0u, 0u}), acc, 0, 0, 0);
        }
      if (fq == 0) {
        const size_t pix = (size_t)(b * H + y0 + i) * W + x;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float o = acc[r];
          if (p.act == SRAD_ACT_RELU) o = fmaxf(o, 0.f);
          else if (p.act == SRAD_ACT_LRELU) o = o > 0.f ? o : o * p.slope;
          o *= p.alpha;
          if (r < p.N) p.Y[pix * p.ldy + p.yoff + r] = o + (p.R ? p.R[pix * p.ldr + r] : 0.f);
        }
      }
    }
  }
}

}  // namespace

bool srad_conv_tail_supported(int prec, const GemmParams& p) {
  static const bool off = getenv("SRAD_NO_CONV_TAIL") != nullptr;
  return !off && prec == SRAD_PREC_BF16 && p.ntaps == 9 && p.stride == 1 && p.Hi == p.Ho && p.Wi == p.Wo && !p.ln_g && p.ps == 0 &&
         p.hsplit_hd == 0 && !p.row_scale && !p.Ypre && !p.pool_part && !p.Xh && !p.Yh && !p.Rh && p.sp_q < 0 &&
         (p.act == SRAD_ACT_NONE || p.act == SRAD_ACT_RELU || p.act == SRAD_ACT_LRELU) && (!p.R || p.rmode == SRAD_RMODE_ADD) &&
         p.N >= 1 && p.N <= 4 && (p.Cin == 40 || p.Cin == 80) && p.Cp == srad_cp(p.Cin) && (p.Wi & 15) == 0 && (p.ldx & 3) == 0 &&
         ((uintptr_t)p.X & 15) == 0 && p.M >= 32768;
}

int srad_launch_conv_tail(const GemmParams& p, hipStream_t stream) {
  SRAD_REQUIRE(srad_conv_tail_supported(SRAD_PREC_BF16, p), "conv_tail: unsupported problem");
  const int tiles = p.M / 16;
  // every wave keeps the whole weight in registers: few, long-lived waves.  Measured (us, 256 / 512 / 768 / 1024 workgroups):
  // 40 -> 3 at 256 px x 8: 81 / 65 / 67 / 70 (62 after the instruction trim);  80 -> 3 at 128 px x 8: 33.9 / 34.7 / 38 / 39;  at 64 px x 8: 11.8 / 14.8 / 14.8 / 14.5
  const int cap = tiles >= 16384 ? 512 : 256;
  const int wgs = tiles / 4 < cap ? tiles / 4 : cap;
  SradProfScope prof(stream, SRAD_K_GEMM_BN16, 2.0 * p.M * p.N * 9.0 * p.Cin, 4.0 * p.M * ((double)p.Cin + p.N * (p.R ? 2 : 1)));
  static const bool no_strip = getenv("SRAD_TAIL_NO_STRIP") != nullptr;
  if (p.Cin == 40 && p.Hi % CT_RS == 0 && !no_strip) {
    const int nstrips = tiles / CT_RS;
    const int swgs = std::max(1, std::min(nstrips / 4, 512));
    hipLaunchKernelGGL(conv_tail40_strip_kernel, dim3(swgs), dim3(256), 0, stream, p, nstrips);
  } else if (p.Cin == 40) hipLaunchKernelGGL(conv_tail_kernel<40>, dim3(wgs), dim3(256), 0, stream, p, tiles);
  else hipLaunchKernelGGL(conv_tail_kernel<80>, dim3(wgs), dim3(256), 0, stream, p, tiles);
  SRAD_CHECK_HIP(hipGetLastError());
  return SRAD_OK;
}

bool srad_conv_thin_supported(int prec, const GemmParams& p) {
  static const bool off = getenv("SRAD_NO_CONV_THIN") != nullptr;
  const int n4 = (p.N + 3) / 4;
  return !off && prec == SRAD_PREC_BF16 && p.ntaps == 9 && p.stride == 1 && p.Hi == p.Ho && p.Wi == p.Wo && !p.ln_g && p.ps == 0 &&
         p.hsplit_hd == 0 && !p.row_scale && !p.Ypre && !p.pool_part && !p.Xh && !p.Yh && !p.Rh &&
         (p.act == SRAD_ACT_NONE || p.act == SRAD_ACT_RELU || p.act == SRAD_ACT_LRELU) && (!p.R || p.rmode == SRAD_RMODE_ADD) &&
         // few INPUT channels only: with 40 / 80 input channels and 3 outputs (the tails) this kernel is a chain of 90 - 180 dependent
         // load rounds per pixel and ran 1.6x SLOWER than the tiled GEMM (195 vs 120 us at 256 px) - those stay there
         (p.Cin & 3) == 0 && p.N >= 1 && p.Cin <= 8 && p.Cin * 4 * n4 <= 320 && (n4 == 1 || n4 == 5 || n4 == 10 || n4 == 20) &&
         (p.ldx & 3) == 0 && ((uintptr_t)p.X & 15) == 0 && p.M >= 32768;
}

int srad_launch_conv_thin(const GemmParams& p, hipStream_t stream) {
  SRAD_REQUIRE(srad_conv_thin_supported(SRAD_PREC_BF16, p), "conv_thin: unsupported problem");
  const int n4 = (p.N + 3) / 4;
  const size_t lds = (size_t)(9 * p.Cin * 4 * n4 + 4 * n4) * sizeof(float);
  const dim3 grid((unsigned)((p.M + 255) / 256));
  const int cls = p.N >= 64 ? SRAD_K_GEMM_BN64 : (p.N > 16 ? SRAD_K_GEMM_BN32 : SRAD_K_GEMM_BN16);
  SradProfScope prof(stream, cls, 2.0 * p.M * p.N * 9.0 * p.Cin, 4.0 * p.M * ((double)p.Cin + p.N * (p.R ? 2 : 1)));
  switch (n4) {
    case 1: hipLaunchKernelGGL(conv_thin_kernel<1>, grid, dim3(256), lds, stream, p); break;
    case 5: hipLaunchKernelGGL(conv_thin_kernel<5>, grid, dim3(256), lds, stream, p); break;
    case 10: hipLaunchKernelGGL(conv_thin_kernel<10>, grid, dim3(256), lds, stream, p); break;
    default: hipLaunchKernelGGL(conv_thin_kernel<20>, grid, dim3(256), lds, stream, p); break;
  }
  SRAD_CHECK_HIP(hipGetLastError());
  return SRAD_OK;
}
