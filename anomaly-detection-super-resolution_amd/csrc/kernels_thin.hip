// kernels_thin.hip - 3x3 stride-1 convolutions with very few INPUT channels: DRN's head (3 -> F, src/drn.py:247) and the data
// gradients of its tail convolutions (3 -> F 2^p, drn.py:256-258, 265-267) in training.  On the tiled MFMA GEMM these pay for K
// padded to 32 per tap (8x the MACs) at the largest pixel count of the model: the head took 149 us at 256 px x 8 images against
// ~15 us of memory time; here 31 us.
// Here: one output pixel per thread, fp32 FMAs against weights held in LDS (every lane reads the same address: a broadcast), the
// input straight from global memory (each pixel is read by nine neighbouring threads: L1 / L2 hits).  Weights are the bf16
// values of the packed layer (what the MFMA path multiplies by), activations stay fp32.
#include "srad_common.h"

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

}  // namespace

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
