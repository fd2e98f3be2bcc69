// kernels_gemm.hip - row-gather MFMA GEMM for gfx950: Linear, 1x1 and 3x3 convolutions of the
// DRCT / DRN forward passes (reference src/drct.py: qkv/proj/fc1/fc2 Linear 262-264,178-181,
// adjust 1x1 convs 334-374, 3x3 convs 782,837,844-847, Upsample 694-713; src/drn.py convs).
//
// One workgroup = 256 threads = 4 waves computes a BM x BN output tile.  K is walked in
// 32-wide chunks (tap, channel-chunk); each chunk of A is gathered from the NHWC activation
// (fp32 in HBM/L2), optionally LayerNorm-ed, converted to the compute type and staged in LDS
// next to the matching chunk of the pre-packed weight.  The next chunk's global loads are
// issued before the current chunk's MFMAs (register-staged prefetch).
//   BF16 mode: v_mfma_f32_16x16x32_bf16, fp32 accumulate.
//   F32  mode: v_mfma_f32_16x16x4_f32 (exact fp32 fma chain) - the parity mode.
#include "srad_common.h"

namespace {

template <int PREC> struct PrecT;
template <> struct PrecT<SRAD_PREC_BF16> { using type = __bf16; static constexpr int STRIDE = 40; };
template <> struct PrecT<SRAD_PREC_F32>  { using type = float;  static constexpr int STRIDE = 36; };

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }

template <int PREC, int BM, int BN, int WAVES_M, int WAVES_N>
__global__ __launch_bounds__(256) void gemm_kernel(const GemmParams p) {
  constexpr int BK = 32;
  constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N;
  constexpr int MT = WM / 16, NT = WN / 16;
  constexpr int RPT = BM / 32;                       // A rows staged per thread
  using T = typename PrecT<PREC>::type;
  constexpr int ST = PrecT<PREC>::STRIDE;
  static_assert(WAVES_M * WAVES_N == 4, "4 waves per workgroup");

  __shared__ __attribute__((aligned(16))) T As[BM * ST];
  __shared__ __attribute__((aligned(16))) T Ws[BN * ST];
  __shared__ float s_mean[BM];
  __shared__ float s_rstd[BM];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm0 = (wave / WAVES_N) * WM;
  const int wn0 = (wave % WAVES_N) * WN;
  const int m0 = blockIdx.y * BM;
  const int n0 = blockIdx.x * BN;

  const bool conv = (p.ntaps == 9) || (p.stride != 1);
  const bool vec = ((p.Cin & 3) == 0) && ((p.ldx & 3) == 0);
  const int Kp = p.ntaps * p.Cp;

  // ---- per-thread A row bookkeeping (rows tid/8 + 32*i, float4 column tid%8) ----
  const int col4 = tid & 7;
  int r_b[RPT], r_oy[RPT], r_ox[RPT];
  bool r_ok[RPT];
#pragma unroll
  for (int i = 0; i < RPT; ++i) {
    const int m = m0 + (tid >> 3) + 32 * i;
    r_ok[i] = m < p.M;
    if (conv) {
      const int hw = p.Ho * p.Wo;
      const int mm = r_ok[i] ? m : 0;
      r_b[i] = mm / hw;
      const int rem = mm - r_b[i] * hw;
      r_oy[i] = rem / p.Wo;
      r_ox[i] = rem - r_oy[i] * p.Wo;
    } else {
      r_b[i] = r_ok[i] ? m : 0; r_oy[i] = 0; r_ox[i] = 0;
    }
  }

  // ---- optional LayerNorm statistics for this tile's rows (two-pass, biased variance) ----
  const bool ln = p.ln_g != nullptr;
  if (ln) {
    constexpr int TPR = 256 / BM;                    // threads per row (4 or 2)
    const int row = tid / TPR, sub = tid % TPR;
    const int m = m0 + row;
    float s = 0.f;
    const float* xr = p.X + (size_t)(m < p.M ? m : 0) * p.ldx;
    for (int c = sub; c < p.Cin; c += TPR) s += xr[c];
#pragma unroll
    for (int o = 1; o < TPR; o <<= 1) s += __shfl_xor(s, o);
    const float mean = s / (float)p.Cin;
    float v = 0.f;
    for (int c = sub; c < p.Cin; c += TPR) { const float d = xr[c] - mean; v += d * d; }
#pragma unroll
    for (int o = 1; o < TPR; o <<= 1) v += __shfl_xor(v, o);
    if (sub == 0) { s_mean[row] = mean; s_rstd[row] = rsqrtf(v / (float)p.Cin + p.ln_eps); }
    __syncthreads();
  }

  const int nchunk_c = p.Cp / BK;
  const int nchunks = p.ntaps * nchunk_c;

  float4 a_reg[RPT];
  constexpr int W_SEGS = (PREC == SRAD_PREC_BF16) ? 4 : 8;      // 16-byte segments per W row chunk
  constexpr int W_PT = (BN * W_SEGS + 255) / 256;
  uint4 w_reg[W_PT];

  auto load_chunk = [&](int ch) {
    const int tap = ch / nchunk_c;
    const int c0 = (ch - tap * nchunk_c) * BK;
    const int ky = tap / 3, kx = tap - ky * 3;
    const int c = c0 + col4 * 4;
#pragma unroll
    for (int i = 0; i < RPT; ++i) {
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      bool ok = r_ok[i];
      size_t pix;
      if (conv) {
        const int iy = r_oy[i] * p.stride + (p.ntaps == 9 ? ky - 1 : 0);
        const int ix = r_ox[i] * p.stride + (p.ntaps == 9 ? kx - 1 : 0);
        ok = ok && iy >= 0 && iy < p.Hi && ix >= 0 && ix < p.Wi;
        pix = ((size_t)r_b[i] * p.Hi + (ok ? iy : 0)) * p.Wi + (ok ? ix : 0);
      } else {
        pix = (size_t)r_b[i];
      }
      if (ok) {
        const float* src = p.X + pix * p.ldx + c;
        if (vec) {
          if (c < p.Cin) v = *reinterpret_cast<const float4*>(src);
        } else {
          if (c + 0 < p.Cin) v.x = src[0];
          if (c + 1 < p.Cin) v.y = src[1];
          if (c + 2 < p.Cin) v.z = src[2];
          if (c + 3 < p.Cin) v.w = src[3];
        }
        if (ln) {
          const int row = (tid >> 3) + 32 * i;
          const float mu = s_mean[row], rs = s_rstd[row];
          if (c + 0 < p.Cin) v.x = (v.x - mu) * rs * p.ln_g[c + 0] + p.ln_b[c + 0];
          if (c + 1 < p.Cin) v.y = (v.y - mu) * rs * p.ln_g[c + 1] + p.ln_b[c + 1];
          if (c + 2 < p.Cin) v.z = (v.z - mu) * rs * p.ln_g[c + 2] + p.ln_b[c + 2];
          if (c + 3 < p.Cin) v.w = (v.w - mu) * rs * p.ln_g[c + 3] + p.ln_b[c + 3];
        }
      }
      a_reg[i] = v;
    }
    const char* wbase = reinterpret_cast<const char*>(p.Wp);
#pragma unroll
    for (int j = 0; j < W_PT; ++j) {
      const int idx = tid + 256 * j;
      if (idx < BN * W_SEGS) {
        const int row = idx / W_SEGS, seg = idx - row * W_SEGS;
        const size_t off = ((size_t)(n0 + row) * Kp + (size_t)ch * BK) * sizeof(T) + (size_t)seg * 16;
        w_reg[j] = *reinterpret_cast<const uint4*>(wbase + off);
      }
    }
  };

  auto store_chunk = [&]() {
#pragma unroll
    for (int i = 0; i < RPT; ++i) {
      const int row = (tid >> 3) + 32 * i;
      T* dst = As + row * ST + col4 * 4;
      if constexpr (PREC == SRAD_PREC_BF16) {
        bf16x4 h;
        h[0] = (__bf16)a_reg[i].x; h[1] = (__bf16)a_reg[i].y; h[2] = (__bf16)a_reg[i].z; h[3] = (__bf16)a_reg[i].w;
        *reinterpret_cast<bf16x4*>(dst) = h;
      } else {
        *reinterpret_cast<float4*>(dst) = a_reg[i];
      }
    }
#pragma unroll
    for (int j = 0; j < W_PT; ++j) {
      const int idx = tid + 256 * j;
      if (idx < BN * W_SEGS) {
        const int row = idx / W_SEGS, seg = idx - row * W_SEGS;
        *reinterpret_cast<uint4*>(reinterpret_cast<char*>(Ws + row * ST) + seg * 16) = w_reg[j];
      }
    }
  };

  f32x4 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int fr = lane & 15, fq = lane >> 4;

  load_chunk(0);
  for (int ch = 0; ch < nchunks; ++ch) {
    store_chunk();
    __syncthreads();
    if (ch + 1 < nchunks) load_chunk(ch + 1);
    if constexpr (PREC == SRAD_PREC_BF16) {
      bf16x8 a[MT], b[NT];
#pragma unroll
      for (int i = 0; i < MT; ++i) a[i] = *reinterpret_cast<const bf16x8*>(As + (wm0 + i * 16 + fr) * ST + 8 * fq);
#pragma unroll
      for (int j = 0; j < NT; ++j) b[j] = *reinterpret_cast<const bf16x8*>(Ws + (wn0 + j * 16 + fr) * ST + 8 * fq);
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
    } else {
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        float4 a[MT], b[NT];
#pragma unroll
        for (int i = 0; i < MT; ++i) a[i] = *reinterpret_cast<const float4*>(As + (wm0 + i * 16 + fr) * ST + ks * 16 + 4 * fq);
#pragma unroll
        for (int j = 0; j < NT; ++j) b[j] = *reinterpret_cast<const float4*>(Ws + (wn0 + j * 16 + fr) * ST + ks * 16 + 4 * fq);
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
          for (int j = 0; j < NT; ++j) {
            // lane group g holds k = 16*ks + 4*g + e: the e-th MFMA sums k over the four groups,
            // so e = 0..3 together cover all 16 k of the step (same permutation on A and B).
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].x, b[j].x, acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].y, b[j].y, acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].z, b[j].z, acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].w, b[j].w, acc[i][j], 0, 0, 0);
          }
      }
    }
    __syncthreads();
  }

  // ---- epilogue: C layout col = lane&15, row = (lane>>4)*4 + e ----
  const int hw_o = p.Ho * p.Wo;
#pragma unroll
  for (int i = 0; i < MT; ++i) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int m = m0 + wm0 + i * 16 + fq * 4 + e;
      if (m >= p.M) continue;
      size_t ybase;
      int pb = 0;
      if (p.ps == 2) {
        pb = m / hw_o;
        const int rem = m - pb * hw_o;
        const int oy = rem / p.Wo, ox = rem - oy * p.Wo;
        ybase = ((size_t)pb * (2 * p.Ho) + 2 * oy) * (2 * p.Wo) + 2 * ox;    // output pixel (dy=dx=0)
      } else {
        ybase = (size_t)m * p.ldy + p.yoff;
        if (p.pool) pb = m / hw_o;
      }
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const int n = n0 + wn0 + j * 16 + fr;
        if (n >= p.N) continue;
        float v = acc[i][j][e];
        if (p.bias) v += p.bias[n];
        if (p.act == SRAD_ACT_GELU) v = gelu_erf(v);
        else if (p.act == SRAD_ACT_LRELU) v = v > 0.f ? v : v * p.slope;
        else if (p.act == SRAD_ACT_RELU) v = fmaxf(v, 0.f);
        v *= p.alpha;
        if (p.R) v += p.R[(size_t)m * p.ldr + n];
        if (p.ps == 2) {
          const int c = n >> 2, dy = (n >> 1) & 1, dx = n & 1;
          p.Y[(ybase + (size_t)dy * (2 * p.Wo) + dx) * p.ldy + p.yoff + c] = v;
        } else {
          p.Y[ybase + n] = v;
        }
        if (p.pool) atomicAdd(p.pool + (size_t)pb * p.N + n, v);
      }
    }
  }
}

// ---- weight packer: PyTorch [N][Cin][taps] fp32 -> [Np][taps][Cp] compute type, zero padded ----
template <int PREC>
__global__ void pack_weight_kernel(const float* __restrict__ src, void* __restrict__ dst, int N, int Cin,
                                   int ntaps, int Np, int Cp) {
  using T = typename PrecT<PREC>::type;
  const size_t total = (size_t)Np * ntaps * Cp;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % Cp);
    const int tap = (int)((i / Cp) % ntaps);
    const int n = (int)(i / ((size_t)Cp * ntaps));
    float v = 0.f;
    if (n < N && c < Cin) v = src[((size_t)n * Cin + c) * ntaps + tap];
    reinterpret_cast<T*>(dst)[i] = (T)v;
  }
}

template <int PREC, int BM, int BN, int WMV, int WNV>
int launch_cfg(const GemmParams& p, hipStream_t s) {
  dim3 grid((p.N + BN - 1) / BN, (p.M + BM - 1) / BM);
  const int cls = BN == 64 ? SRAD_K_GEMM_64x64 : (BN == 32 ? SRAD_K_GEMM_128x32 : SRAD_K_GEMM_128x16);
  // algorithmic work: 2*M*N*K flops; bytes = A rows once + packed W once + Y once (+ residual)
  const double K = (double)p.ntaps * p.Cin;
  const double wbytes = (double)p.N * K * (PREC == SRAD_PREC_BF16 ? 2 : 4);
  const double abytes = 4.0 * (p.ntaps == 9 ? (double)p.M * p.stride * p.stride : (double)p.M) * p.Cin;
  SradProfScope prof(s, cls, 2.0 * p.M * p.N * K, abytes + wbytes + 4.0 * p.M * p.N * (p.R ? 2 : 1));
  hipLaunchKernelGGL((gemm_kernel<PREC, BM, BN, WMV, WNV>), grid, dim3(256), 0, s, p);
  return SRAD_OK;
}

template <int PREC>
int launch_prec(const GemmParams& p, hipStream_t s) {
  if (p.N <= 16) return launch_cfg<PREC, 128, 16, 4, 1>(p, s);
  if (p.N <= 32) return launch_cfg<PREC, 128, 32, 4, 1>(p, s);
  return launch_cfg<PREC, 64, 64, 2, 2>(p, s);
}

}  // namespace

int srad_launch_gemm(int prec, const GemmParams& p, hipStream_t stream) {
  SRAD_REQUIRE(p.M > 0 && p.N > 0 && p.Cin > 0, "gemm: empty problem M=%d N=%d Cin=%d", p.M, p.N, p.Cin);
  SRAD_REQUIRE(p.Cp == srad_cp(p.Cin), "gemm: Cp %d does not match Cin %d", p.Cp, p.Cin);
  SRAD_REQUIRE(p.ntaps == 1 || p.ntaps == 9, "gemm: ntaps must be 1 or 9 (got %d)", p.ntaps);
  SRAD_REQUIRE(p.ps == 0 || (p.ps == 2 && p.N % 4 == 0), "gemm: pixel-shuffle needs N %% 4 == 0");
  SRAD_REQUIRE(!(p.ln_g && (p.ntaps != 1 || p.stride != 1)), "gemm: LayerNorm fusion only for row-identity A");
  if (p.ntaps == 9 || p.stride != 1 || p.ps || p.pool)
    SRAD_REQUIRE(p.Ho > 0 && p.Wo > 0 && p.M % (p.Ho * p.Wo) == 0, "gemm: M=%d not a multiple of Ho*Wo=%d*%d", p.M, p.Ho, p.Wo);
  int rc = prec == SRAD_PREC_BF16 ? launch_prec<SRAD_PREC_BF16>(p, stream) : launch_prec<SRAD_PREC_F32>(p, stream);
  if (rc) return rc;
  SRAD_CHECK_HIP(hipGetLastError());
  return SRAD_OK;
}

int srad_launch_pack_weight(int prec, const float* src, void* dst, int n, int cin, int ntaps, hipStream_t stream) {
  const int Np = srad_np(n), Cp = srad_cp(cin);
  const size_t total = (size_t)Np * ntaps * Cp;
  const int blocks = (int)((total + 255) / 256 > 2048 ? 2048 : (total + 255) / 256);
  SradProfScope prof(stream, SRAD_K_PACK, 0.0, 4.0 * n * cin * ntaps + (prec == SRAD_PREC_BF16 ? 2.0 : 4.0) * total);
  if (prec == SRAD_PREC_BF16)
    hipLaunchKernelGGL((pack_weight_kernel<SRAD_PREC_BF16>), dim3(blocks), dim3(256), 0, stream, src, dst, n, cin, ntaps, Np, Cp);
  else
    hipLaunchKernelGGL((pack_weight_kernel<SRAD_PREC_F32>), dim3(blocks), dim3(256), 0, stream, src, dst, n, cin, ntaps, Np, Cp);
  SRAD_CHECK_HIP(hipGetLastError());
  return SRAD_OK;
}
