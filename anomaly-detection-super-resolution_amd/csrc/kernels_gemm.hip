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
//   BF16X3 mode: split-bf16 - A and W staged as hi + lo bf16 planes, three bf16 MFMAs per product (hi.hi + hi.lo + lo.hi),
//                fp32 accumulate: fp32-grade results at the bf16 pipe's rate (srad_common.h).
#include "srad_common.h"
#include <stdlib.h>

namespace {

template <int PREC> struct PrecT;
template <> struct PrecT<SRAD_PREC_BF16> { using type = __bf16; static constexpr int PAD = 8; };
template <> struct PrecT<SRAD_PREC_F32>  { using type = float;  static constexpr int PAD = 4; };
template <> struct PrecT<SRAD_PREC_BF16X3> { using type = __bf16; static constexpr int PAD = 8; };   // two planes (hi, lo) of this type

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
// d/dx of the exact-erf GELU: Phi(x) + x * phi(x)
__device__ __forceinline__ float dgelu_erf(float x) {
  return 0.5f * (1.0f + erff(x * 0.70710678118654752440f)) + x * 0.39894228040143267794f * expf(-0.5f * x * x);
}

// CPS = 32-wide K chunks per stage.  A whole stage (up to CPS*32 columns of K for the BM rows of A
// and the BN rows of W) is loaded with every load in flight at once, written to LDS, and consumed by
// the MFMAs while the next stage's loads are already in flight in registers.  At these sizes
// (K = 180..1728, operands L2/MALL resident) the kernel is latency bound, so the stage is made as
// deep as LDS allows instead of looping over 32-wide slices.
//
// Rules this kernel is written to (each one measured on MI355X; DESIGN.md "GEMM: what made it slow"):
//   * every global load is UNCONDITIONAL on a clamped, always-valid address.  A load inside a
//     per-element branch is waited for before the next one issues.
//   * every independent load (first A/W stage, bias, residual, LayerNorm gamma/beta) is issued at the
//     top of the kernel, so a workgroup pays ONE memory latency, not one per phase.
//   * nothing consumes a loaded register inside load_*(): LayerNorm / conv zero-fill are applied when
//     the registers are written to LDS, so the loads stay in flight across the MFMAs.
//   * LayerNorm statistics come from the A registers themselves (the 8 lanes that share a row hold
//     its whole K <= 320), not from a second pass over memory.
//   * mode switches (LayerNorm, conv gather, special epilogues) are template parameters.
//   * the kernel is VALU-issue bound (one wave per SIMD, ~4 cycles per instruction), so addresses are
//     uniform 64-bit bases (SALU) plus per-thread 32-bit byte offsets computed once.
//   * activations always have a channel count / row stride that is a multiple of 4 floats.
// SPECIAL = pixel-shuffle scatter / head-padded columns in the epilogue.
template <int PREC, int BM, int BN, int WAVES_M, int WAVES_N, int CPS, bool LN, bool CONV, bool SPECIAL>
__global__ __launch_bounds__(256) void gemm_kernel(const GemmParams p) {
  constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N;
  constexpr int MT = WM / 16, NT = WN / 16;
  constexpr int RPT = BM / 32;                       // A rows staged per thread
  constexpr int CPA = LN ? CPS + (PREC == SRAD_PREC_BF16 ? 2 : 6) : CPS;   // LN: the whole K (<= 320) stays resident
  using T = typename PrecT<PREC>::type;
  constexpr bool X3 = PREC == SRAD_PREC_BF16X3;
  constexpr int PADE = PrecT<PREC>::PAD;
  constexpr int STA = CPA * 32 + PADE;               // LDS row stride of A (elements)
  constexpr int STW = CPS * 32 + PADE;               // LDS row stride of W
  static_assert(WAVES_M * WAVES_N == 4, "4 waves per workgroup");
  static_assert(!(LN && CONV), "LayerNorm fusion is for row-identity A only");
  static_assert(!LN || CPA * 32 >= 320, "LayerNorm variants keep K <= 320 resident");

  extern __shared__ __attribute__((aligned(16))) char smem[];
  T* As = reinterpret_cast<T*>(smem);
  T* Ws = As + BM * STA;
  T* AsL = Ws + BN * STW;                                  // split-bf16: the lo planes of both tiles
  T* WsL = AsL + BM * STA;
  float* s_g = reinterpret_cast<float*>(X3 ? WsL + BN * STW : Ws + BN * STW);   // LN only: [CPA*32] gamma, [CPA*32] beta
  float* s_b = s_g + CPA * 32;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm0 = (wave / WAVES_N) * WM;
  const int wn0 = (wave % WAVES_N) * WN;
  // XCD-aware tile mapping: workgroups are dealt round-robin over the 8 XCDs (private L2 each), so
  // linear id L runs on XCD L % 8.  Give every XCD a contiguous range of (m-tile, n-tile) pairs with
  // n fastest: all N-tiles of one row-tile then share one L2 and the A rows are fetched from the
  // Infinity Cache once instead of once per N-tile.  Pure speed: any placement is still correct.
  int m_tile, n_tile;
  {
    const int nx = gridDim.x, total = gridDim.x * gridDim.y;
    const int L = blockIdx.x + nx * blockIdx.y;
    const int xcd = L & 7, slot = L >> 3;
    const int q = total >> 3, r = total & 7;
    const int t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
    m_tile = t / nx;
    n_tile = t - m_tile * nx;
  }
  const int m0 = m_tile * BM;
  const int n0 = n_tile * BN;
  const int Kp = p.ntaps * p.Cp;
  const char* const Xb = reinterpret_cast<const char*>(p.X);
  const int fr = lane & 15, fq = lane >> 4;

  // ---- per-thread A bookkeeping: rows tid/8 + 32*i, float4 column tid%8 of each 32-wide chunk.
  //      a_off[i] = byte offset of the row / centre pixel from X, computed once.  Columns are clamped
  //      to Cin-4 (never read past a row) and the K padding is zero-filled at LDS-store time. ----
  const int col4 = tid & 7;
  unsigned a_off[RPT];
  [[maybe_unused]] int r_iy[RPT], r_ix[RPT];         // conv: top-left input coordinate of the 3x3 window
#pragma unroll
  for (int i = 0; i < RPT; ++i) {
    const int mm = min(m0 + (tid >> 3) + 32 * i, p.M - 1);
    if constexpr (CONV) {
      const int hw = p.Ho * p.Wo;
      const int bb = mm / hw;
      const int rem = mm - bb * hw;
      const int oy = rem / p.Wo, ox = rem - oy * p.Wo;
      const int pad = p.ntaps == 9 ? 1 : 0;
      r_iy[i] = oy * p.stride - pad;
      r_ix[i] = ox * p.stride - pad;
      a_off[i] = (unsigned)(((bb * p.Hi + oy * p.stride) * p.Wi + ox * p.stride) * p.ldx) * 4u;
    } else {
      a_off[i] = (unsigned)(mm * p.ldx) * 4u;
    }
  }

  const int nchunk_c = p.Cp / 32;
  const int nchunks = p.ntaps * nchunk_c;
  const int nstages = (nchunks + CPS - 1) / CPS;

  f32x4 a_reg[RPT][CPA];
  unsigned a_ok = 0u;                                          // bit (i*CPA + j): block holds real data
  constexpr int W_SEGS = 32 * (int)sizeof(T) / 16;           // 16-byte segments per chunk per W row
  constexpr int SEGS = CPS * W_SEGS;                         // 16-byte segments per W stage row
  constexpr int WRP = 256 / SEGS;                            // W rows per pass of the 256 threads
  static_assert(SEGS == 32 || SEGS == 16, "W stage rows are 32 or 16 x 16 bytes");
  static_assert(RPT * CPA <= 32, "a_ok bitmask too small");
  constexpr int W_PT = BN / WRP;                              // rows tid/SEGS + WRP*j, segment tid%SEGS
  static_assert(W_PT * WRP == BN, "W rows must divide over the threads");
  u32x4 w_reg[W_PT];
  [[maybe_unused]] u32x4 w_lo[X3 ? W_PT : 1];
  const size_t w_lo_off = (size_t)((p.N + 127) / 128 * 128) * Kp * 2;   // split-bf16: hi plane -> lo plane of the pack (srad_packed_lo_offset)
  const int w_seg = tid % SEGS;
  const unsigned w_off0 = (unsigned)((tid / SEGS) * Kp) * (unsigned)sizeof(T);
  const char* const Wb = reinterpret_cast<const char*>(p.Wp) + (size_t)n0 * Kp * sizeof(T);
  [[maybe_unused]] int ld_tap = 0, ld_c0 = 0;                  // conv: running (tap, channel) of the next chunk

  auto load_a = [&](int st) {
    const int ch0 = st * CPS;
    a_ok = 0u;
#pragma unroll
    for (int j = 0; j < CPA; ++j) {
      if constexpr (CONV) {
        const int ky = ld_tap / 3, kx = ld_tap - ky * 3;                 // ld_tap <= 8: cheap scalar math
        const int dy = p.ntaps == 9 ? ky : 0, dx = p.ntaps == 9 ? kx : 0;
        const int pad = p.ntaps == 9 ? 1 : 0;
        const int delta = ((dy - pad) * p.Wi + (dx - pad)) * p.ldx * 4;
        const int c = ld_c0 + col4 * 4;
        const bool c_ok = c < p.Cin;
        const unsigned coff = (unsigned)min(c, p.Cin - 4) * 4u;
#pragma unroll
        for (int i = 0; i < RPT; ++i) {
          const int iy = r_iy[i] + dy, ix = r_ix[i] + dx;
          const bool in = iy >= 0 && iy < p.Hi && ix >= 0 && ix < p.Wi;
          const unsigned off = a_off[i] + (unsigned)(in ? delta : 0) + coff;     // centre pixel when outside
          a_reg[i][j] = *reinterpret_cast<const f32x4*>(Xb + off);
          a_ok |= ((in && c_ok) ? 1u : 0u) << (i * CPA + j);
        }
        if (ch0 + j + 1 < nchunks) {                                       // clamp: never step past the end
          ld_c0 += 32;
          if (ld_c0 == p.Cp) { ld_c0 = 0; ++ld_tap; }
        }
      } else {
        const int c = min(ch0 + j, nchunks - 1) * 32 + col4 * 4;
        const unsigned coff = (unsigned)min(c, p.Cin - 4) * 4u;
        const bool c_ok = c < p.Cin;
#pragma unroll
        for (int i = 0; i < RPT; ++i) {
          a_reg[i][j] = *reinterpret_cast<const f32x4*>(Xb + a_off[i] + coff);
          a_ok |= (c_ok ? 1u : 0u) << (i * CPA + j);
        }
      }
    }
  };
  auto load_w = [&](int st) {
    const int ch0 = st * CPS;
    const int nch = min(CPS, nchunks - ch0);
    const unsigned woff = w_off0 + (unsigned)((w_seg < nch * W_SEGS ? w_seg : 0) * 16);
    const char* wb = Wb + (size_t)ch0 * 32 * sizeof(T);
#pragma unroll
    for (int j = 0; j < W_PT; ++j)
      w_reg[j] = *reinterpret_cast<const u32x4*>(wb + (size_t)j * WRP * Kp * sizeof(T) + woff);
    if constexpr (X3) {
#pragma unroll
      for (int j = 0; j < W_PT; ++j)
        w_lo[j] = *reinterpret_cast<const u32x4*>(wb + w_lo_off + (size_t)j * WRP * Kp * sizeof(T) + woff);
    }
  };
  [[maybe_unused]] float ln_mu[RPT], ln_rs[RPT];
  auto store_a = [&]() {
#pragma unroll
    for (int j = 0; j < CPA; ++j) {
#pragma unroll
      for (int i = 0; i < RPT; ++i) {
        const int row = (tid >> 3) + 32 * i;
        f32x4 v = a_reg[i][j];
        if constexpr (LN) {
          const f32x4 g4 = *reinterpret_cast<const f32x4*>(s_g + j * 32 + col4 * 4);
          const f32x4 b4 = *reinterpret_cast<const f32x4*>(s_b + j * 32 + col4 * 4);
          v = (v - ln_mu[i]) * ln_rs[i] * g4 + b4;               // gamma = beta = 0 in the K padding
        }
        {
          // zero-fill: K padding (always) and taps outside the image (conv).  The clamped loads may
          // have fetched anything, including NaN bit patterns from a neighbouring buffer.
          const bool ok = (a_ok >> (i * CPA + j)) & 1u;
          v = ok ? v : f32x4{0.f, 0.f, 0.f, 0.f};
        }
        T* dst = As + row * STA + j * 32 + col4 * 4;
        if constexpr (X3) {
          bf16x4 h, l;
          srad_split4(v, h, l);
          *reinterpret_cast<bf16x4*>(dst) = h;
          *reinterpret_cast<bf16x4*>(AsL + row * STA + j * 32 + col4 * 4) = l;
        } else if constexpr (PREC == SRAD_PREC_BF16) {
          bf16x4 h;
          h[0] = (__bf16)v[0]; h[1] = (__bf16)v[1]; h[2] = (__bf16)v[2]; h[3] = (__bf16)v[3];
          *reinterpret_cast<bf16x4*>(dst) = h;
        } else {
          *reinterpret_cast<f32x4*>(dst) = v;
        }
      }
    }
  };
  auto store_w = [&]() {
    char* wdst = reinterpret_cast<char*>(Ws) + ((tid / SEGS) * STW) * (int)sizeof(T) + w_seg * 16;
#pragma unroll
    for (int j = 0; j < W_PT; ++j) *reinterpret_cast<u32x4*>(wdst + j * WRP * STW * (int)sizeof(T)) = w_reg[j];
    if constexpr (X3) {
      char* ldst = reinterpret_cast<char*>(WsL) + ((tid / SEGS) * STW) * (int)sizeof(T) + w_seg * 16;
#pragma unroll
      for (int j = 0; j < W_PT; ++j) *reinterpret_cast<u32x4*>(ldst + j * WRP * STW * (int)sizeof(T)) = w_lo[j];
    }
  };

  // ---- issue every independent load now: first A/W stage, LN gamma/beta, bias, residual ----
  load_a(0);
  load_w(0);
  [[maybe_unused]] float lg = 0.f, lb = 0.f;
  if constexpr (LN) {
    // threads 0 .. CPA*32-1 fetch one gamma/beta each (clamped; zero beyond Cin so K padding stays 0)
    const int c = min(tid, p.Cin - 1);
    lg = p.ln_g[c];
    lb = p.ln_b[c];
  }
  int ncol[NT];
  float bv[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) { ncol[j] = n0 + wn0 + j * 16 + fr; bv[j] = 0.f; }
  if (p.bias) {
#pragma unroll
    for (int j = 0; j < NT; ++j) bv[j] = p.bias[min(ncol[j], p.N - 1)];
  }
  float rv[MT][4][NT];
  if (p.R) {
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const unsigned rrow = (unsigned)(min(m0 + wm0 + i * 16 + fq * 4 + e, p.M - 1) * p.ldr);
#pragma unroll
        for (int j = 0; j < NT; ++j) rv[i][e][j] = p.R[rrow + (unsigned)min(ncol[j], p.N - 1)];
      }
  }

  if constexpr (LN) {
    static_assert(CPA * 32 <= 512, "gamma/beta staging uses one or two elements per thread");
    for (int c = tid; c < CPA * 32; c += 256) {
      float g = lg, bb = lb;
      if (c >= 256) { const int cc = min(c, p.Cin - 1); g = p.ln_g[cc]; bb = p.ln_b[cc]; }
      s_g[c] = c < p.Cin ? g : 0.f;
      s_b[c] = c < p.Cin ? bb : 0.f;
    }
    // row statistics from the registers: the 8 lanes tid%8 of a row hold all of its K
#pragma unroll
    for (int i = 0; i < RPT; ++i) {
      float s = 0.f, ss = 0.f;
#pragma unroll
      for (int j = 0; j < CPA; ++j) {
        const bool in = (j * 32 + col4 * 4) < p.Cin;                 // Cin % 4 == 0: whole float4 in or out
        const f32x4 vw = in ? a_reg[i][j] : f32x4{0.f, 0.f, 0.f, 0.f};
        s += (vw[0] + vw[1]) + (vw[2] + vw[3]);
        ss += (vw[0] * vw[0] + vw[1] * vw[1]) + (vw[2] * vw[2] + vw[3] * vw[3]);
      }
      { s = srad_row8_sum(s); ss = srad_row8_sum(ss); }
      const float mean = s / (float)p.Cin;
      ln_mu[i] = mean;
      ln_rs[i] = rsqrtf(fmaxf(ss / (float)p.Cin - mean * mean, 0.f) + p.ln_eps);
    }
    __syncthreads();                                 // gamma/beta visible
  }

  f32x4 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const T* const a_rd = As + (wm0 + fr) * STA + (PREC != SRAD_PREC_F32 ? 8 : 4) * fq;
  const T* const w_rd = Ws + (wn0 + fr) * STW + (PREC != SRAD_PREC_F32 ? 8 : 4) * fq;

  for (int st = 0; st < nstages; ++st) {
    if (!LN || st == 0) store_a();
    store_w();
    __syncthreads();
    if (st + 1 < nstages) {
      if constexpr (!LN) load_a(st + 1);
      load_w(st + 1);
    }
    const int nch = min(CPS, nchunks - st * CPS);
    const T* a_st = a_rd + (LN ? st * CPS * 32 : 0);
#pragma unroll
    for (int cc = 0; cc < CPS; ++cc) {
      if (cc < nch) {
        if constexpr (X3) {
          bf16x8 a[MT], al[MT], b[NT], bl[NT];
#pragma unroll
          for (int i = 0; i < MT; ++i) {
            a[i] = *reinterpret_cast<const bf16x8*>(a_st + i * 16 * STA + cc * 32);
            al[i] = *reinterpret_cast<const bf16x8*>(a_st + (AsL - As) + i * 16 * STA + cc * 32);
          }
#pragma unroll
          for (int j = 0; j < NT; ++j) {
            b[j] = *reinterpret_cast<const bf16x8*>(w_rd + j * 16 * STW + cc * 32);
            bl[j] = *reinterpret_cast<const bf16x8*>(w_rd + (WsL - Ws) + j * 16 * STW + cc * 32);
          }
#pragma unroll
          for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j) {
              acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[i], b[j], acc[i][j], 0, 0, 0);
              acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], bl[j], acc[i][j], 0, 0, 0);
              acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
            }
        } else if constexpr (PREC == SRAD_PREC_BF16) {
          bf16x8 a[MT], b[NT];
#pragma unroll
          for (int i = 0; i < MT; ++i) a[i] = *reinterpret_cast<const bf16x8*>(a_st + i * 16 * STA + cc * 32);
#pragma unroll
          for (int j = 0; j < NT; ++j) b[j] = *reinterpret_cast<const bf16x8*>(w_rd + j * 16 * STW + cc * 32);
#pragma unroll
          for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
        } else {
#pragma unroll
          for (int ks = 0; ks < 2; ++ks) {
            f32x4 a[MT], b[NT];
#pragma unroll
            for (int i = 0; i < MT; ++i) a[i] = *reinterpret_cast<const f32x4*>(a_st + i * 16 * STA + cc * 32 + ks * 16);
#pragma unroll
            for (int j = 0; j < NT; ++j) b[j] = *reinterpret_cast<const f32x4*>(w_rd + j * 16 * STW + cc * 32 + ks * 16);
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
              for (int j = 0; j < NT; ++j) {
                // lane group g holds k = 16*ks + 4*g + e: the e-th MFMA sums k over the four groups,
                // so e = 0..3 together cover all 16 k of the step (same permutation on A and B).
#pragma unroll
                for (int e = 0; e < 4; ++e)
                  acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i][e], b[j][e], acc[i][j], 0, 0, 0);
              }
          }
        }
      }
    }
    __syncthreads();
  }

  // ---- epilogue: C layout col = lane&15, row = (lane>>4)*4 + e ----
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] += bv[j];
  if (p.Ypre) {                                       // training: keep the pre-activation (GELU backward needs it)
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int m = m0 + wm0 + i * 16 + fq * 4 + e;
#pragma unroll
        for (int j = 0; j < NT; ++j)
          if (m < p.M && ncol[j] < p.N) p.Ypre[(unsigned)(m * p.ldy + p.yoff) + (unsigned)ncol[j]] = acc[i][j][e];
      }
  }
  if (p.act == SRAD_ACT_GELU) {
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[i][j][e] = gelu_erf(acc[i][j][e]);
  } else if (p.act == SRAD_ACT_LRELU) {
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[i][j][e] = acc[i][j][e] > 0.f ? acc[i][j][e] : acc[i][j][e] * p.slope;
  } else if (p.act == SRAD_ACT_RELU) {
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[i][j][e] = fmaxf(acc[i][j][e], 0.f);
  }
  unsigned ycol[NT];                                  // element offset of column j inside an output row
  float hq[NT];                                       // bf16 head-split output (Yh): the column's factor (q slices: the attention's scale) ...
  int hpad[NT];                                       // ... and, for a slice's last real column, what its padding starts with (0 / 1), else -1
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int n = ncol[j];
    ycol[j] = (unsigned)n;
    hq[j] = 1.f; hpad[j] = -1;
    if constexpr (SPECIAL) {
      if (p.ps == 2) {
        const int c = n >> 2, dy = (n >> 1) & 1, dx = n & 1;
        ycol[j] = (unsigned)((dy * (2 * p.Wo) + dx) * p.ldy + c);
      } else if (p.hsplit_hd > 0) {
        const int hh = n / p.hsplit_hd, cc = n - hh * p.hsplit_hd;
        ycol[j] = (unsigned)(hh * p.hsplit_hdp + cc);
        hq[j] = hh < p.hsplit_heads ? p.hsplit_qscale : 1.f;
        if (cc == p.hsplit_hd - 1) hpad[j] = hh >= 2 * p.hsplit_heads ? 1 : 0;
      }
    }
  }
  float csum[NT];                                     // pool_part: this lane's column sums over its rows of the tile
#pragma unroll
  for (int j = 0; j < NT; ++j) csum[j] = 0.f;
#pragma unroll
  for (int i = 0; i < MT; ++i) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int m = m0 + wm0 + i * 16 + fq * 4 + e;
      const bool m_ok = m < p.M;
      const int mc = m_ok ? m : 0;
      unsigned yrow = (unsigned)(mc * p.ldy + p.yoff);
      if constexpr (SPECIAL) {
        if (p.ps == 2) {
          const int hw_o = p.Ho * p.Wo;
          const int pb = mc / hw_o;
          const int rem = mc - pb * hw_o;
          const int oy = rem / p.Wo, ox = rem - oy * p.Wo;
          yrow = (unsigned)(((pb * (2 * p.Ho) + 2 * oy) * (2 * p.Wo) + 2 * ox) * p.ldy + p.yoff);
        }
      }
      float rs = p.alpha;
      if (p.row_scale) rs *= p.row_scale[mc / p.rps];
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        float v = acc[i][j][e] * rs;
        if (p.R) {
          const float r = rv[i][e][j];
          if (p.rmode == SRAD_RMODE_ADD) v += r;
          else if (p.rmode == SRAD_RMODE_DGELU) v *= dgelu_erf(r);
          else v *= r > 0.f ? 1.f : p.slope;
        }
        csum[j] += m_ok ? v : 0.f;
        if (m_ok && ncol[j] < p.N) {
          if constexpr (SPECIAL) {
            if (p.Yh) {                                    // head-split bf16 output for the window attention (see GemmParams::Yh)
              __bf16* const dst = p.Yh + yrow + ycol[j];
              dst[0] = (__bf16)(v * hq[j]);
              if (hpad[j] >= 0) {                          // the slice's last real column: its (at most 3) padding columns go out with it
                dst[1] = (__bf16)(float)hpad[j];
                for (int q = 2; q <= p.hsplit_hdp - p.hsplit_hd; ++q) dst[q] = (__bf16)0.f;
              }
              continue;
            }
          }
          p.Y[yrow + ycol[j]] = v;
        }
      }
    }
  }
  if (p.pool_part) {                                  // kernel-uniform: the tile's column sums, waves combined in fixed order
    float* const red = reinterpret_cast<float*>(smem);
#pragma unroll
    for (int j = 0; j < NT; ++j) { csum[j] += __shfl_xor(csum[j], 16); csum[j] += __shfl_xor(csum[j], 32); }
    __syncthreads();                                  // every wave is done with the staging tiles
    if (fq == 0) {
#pragma unroll
      for (int j = 0; j < NT; ++j) red[(wave / WAVES_N) * BN + wn0 + j * 16 + fr] = csum[j];
    }
    __syncthreads();
    if (tid < BN && n0 + tid < p.N) {
      float t = 0.f;
#pragma unroll
      for (int w = 0; w < WAVES_M; ++w) t += red[w * BN + tid];
      p.pool_part[(size_t)m_tile * p.N + n0 + tid] = t;
    }
  }
}

// ---- weight packer: PyTorch [N][Cin][taps] fp32 -> [Np][taps][Cp] compute type, zero padded.
// Optional channel-group padding: the destination channels are groups of grp_pad of which the first
// grp_real are real (dst channel c <- src channel (c / grp_pad) * grp_real + c % grp_pad), and rows
// n >= N are zero - how DRN's 10-channel level is widened to 12 for the float4 kernels. ----
// one packed element: bf16 / fp32, or (split-bf16) the hi term here and the lo term `total` elements further on
template <int PREC>
__device__ __forceinline__ void pack_store(void* dst, size_t i, size_t total, float v) {
  if constexpr (PREC == SRAD_PREC_F32) reinterpret_cast<float*>(dst)[i] = v;
  else {
    const __bf16 h = (__bf16)v;
    reinterpret_cast<__bf16*>(dst)[i] = h;
    if constexpr (PREC == SRAD_PREC_BF16X3) reinterpret_cast<__bf16*>(dst)[total + i] = (__bf16)(v - (float)h);
  }
}

template <int PREC>
__global__ void pack_weight_kernel(const float* __restrict__ src, void* __restrict__ dst, int N, int Cin,
                                   int ntaps, int Np, int Cp, int grp_real, int grp_pad) {
  const size_t total = (size_t)Np * ntaps * Cp;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % Cp);
    const int tap = (int)((i / Cp) % ntaps);
    const int n = (int)(i / ((size_t)Cp * ntaps));
    int sc = c;
    if (grp_pad > 0) {
      const int blk = c / grp_pad, o = c - blk * grp_pad;
      sc = o < grp_real ? blk * grp_real + o : -1;
    }
    float v = 0.f;
    if (n < N && sc >= 0 && sc < Cin) v = src[((size_t)n * Cin + sc) * ntaps + tap];
    pack_store<PREC>(dst, i, total, v);
  }
}

// Transposed packing for the data gradient: dst row = input channel c, column = output channel n, tap mirrored.
template <int PREC>
__global__ void pack_weight_t_kernel(const float* __restrict__ src, void* __restrict__ dst, int N, int Cin, int ntaps,
                                     int Rp, int Kp) {
  const size_t total = (size_t)Rp * ntaps * Kp;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int n = (int)(i % Kp);
    const int tap = (int)((i / Kp) % ntaps);
    const int c = (int)(i / ((size_t)Kp * ntaps));
    float v = 0.f;
    if (n < N && c < Cin) v = src[((size_t)n * Cin + c) * ntaps + (ntaps - 1 - tap)];
    pack_store<PREC>(dst, i, total, v);
  }
}

// Fragment-major bf16 pack for kernels that feed MFMA operands straight from global memory (kernels_fused.hip):
// 16-row x 32-k tiles, tile (n / 16, k / 32) is 1 KB contiguous, row-major inside - exactly the 64 lanes x 16 bytes
// of one v_mfma_f32_16x16x32_bf16 operand, so a wave's fragment load is one fully coalesced request.
// LO: write the lo terms bf16(w - float(bf16(w))) instead (the split-bf16 kernels' second weight stream)
__device__ __forceinline__ __bf16 frag_term(float v, bool lo) {
  const __bf16 h = (__bf16)v;
  return lo ? (__bf16)(v - (float)h) : h;
}
template <bool LO>
__global__ void pack_weight_frag_kernel(const float* __restrict__ src, __bf16* __restrict__ dst, int N, int Cin, int Np, int Kp,
                                        int sn, int sk) {
  const size_t total = (size_t)Np * Kp;
  const int ktiles = Kp / 32;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int kin = (int)(i & 31), rin = (int)((i >> 5) & 15);
    const size_t tile = i >> 9;
    const int kt = (int)(tile % ktiles), nt = (int)(tile / ktiles);
    const int n = nt * 16 + rin, k = kt * 32 + kin;
    dst[i] = frag_term((n < N && k < Cin) ? src[(size_t)n * sn + (size_t)k * sk] : 0.f, LO);
  }
}

template <bool LO>
__global__ void pack_qkv_frag_kernel(const float* __restrict__ src, __bf16* __restrict__ dst, int d, int heads, int hd, int HDP, int Kp) {
  const size_t total = (size_t)heads * 3 * HDP * Kp;
  const int ktiles = Kp / 32, rtiles = 3 * HDP / 16;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int kin = (int)(i & 31), rin = (int)((i >> 5) & 15);
    const size_t tile = i >> 9;
    const int kt = (int)(tile % ktiles);
    const int rg = (int)(tile / ktiles);
    const int h = rg / rtiles, vr = (rg - h * rtiles) * 16 + rin;
    const int which = vr / HDP, c = vr - which * HDP, k = kt * 32 + kin;
    dst[i] = frag_term((c < hd && k < d) ? src[(size_t)(which * d + h * hd + c) * d + k] : 0.f, LO);
  }
}

template <int PREC, int BM, int BN, int WMV, int WNV, bool LN, bool CONV, bool SPECIAL, int CPS_ = 0>
int launch_one(const GemmParams& p, hipStream_t s) {
  constexpr int CPS = CPS_ ? CPS_ : (PREC == SRAD_PREC_BF16 ? 8 : 4);
  constexpr int CPA = LN ? CPS + (PREC == SRAD_PREC_BF16 ? 2 : 6) : CPS;
  using T = typename PrecT<PREC>::type;
  constexpr size_t lds = ((size_t)BM * (CPA * 32 + PrecT<PREC>::PAD) + (size_t)BN * (CPS * 32 + PrecT<PREC>::PAD)) * sizeof(T) *
                             (PREC == SRAD_PREC_BF16X3 ? 2 : 1) +
                         (LN ? 2 * (size_t)CPA * 32 * sizeof(float) : 0);
  auto kern = gemm_kernel<PREC, BM, BN, WMV, WNV, CPS, LN, CONV, SPECIAL>;
  static SradOncePerDevice configured;
  if (configured.need()) {
    SRAD_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    configured.done();
  }
  // pool_part rows are summed per image by callers that size them with srad_gemm_tile_rows(): a tile choice that drifts from that
  // mirror, or an image that is not a whole number of tiles, must fail here instead of averaging the wrong rows
  if (p.pool_part)
    SRAD_REQUIRE(BM == srad_gemm_tile_rows(PREC, p) && p.Ho > 0 && (p.Ho * p.Wo) % BM == 0,
                 "gemm: pool_part with %d-row tiles, but srad_gemm_tile_rows says %d and an image has %d rows", BM, srad_gemm_tile_rows(PREC, p), p.Ho * p.Wo);
  dim3 grid((p.N + BN - 1) / BN, (p.M + BM - 1) / BM);
  const int cls = BN >= 64 ? SRAD_K_GEMM_BN64 : (BN == 32 ? SRAD_K_GEMM_BN32 : SRAD_K_GEMM_BN16);
  // algorithmic work: 2*M*N*K flops; bytes = A rows once + packed W once + Y once (+ residual)
  const double K = (double)p.ntaps * p.Cin;
  const double wbytes = (double)p.N * K * (PREC == SRAD_PREC_BF16 ? 2 : 4);   // (split-bf16: two bf16 planes)
  const double abytes = 4.0 * (p.ntaps == 9 ? (double)p.M * p.stride * p.stride : (double)p.M) * p.Cin;
  SradProfScope prof(s, cls, 2.0 * p.M * p.N * K, abytes + wbytes + 4.0 * p.M * p.N * (p.R ? 2 : 1));
  hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, p);
  return SRAD_OK;
}

template <int PREC, int BM, int BN, int WMV, int WNV>
int launch_cfg(const GemmParams& p, hipStream_t s) {
  const bool conv = p.ntaps == 9 || p.stride != 1;
  const bool special = p.ps == 2 || p.hsplit_hd > 0;
  if (conv) {
    if constexpr (PREC == SRAD_PREC_BF16 && ((BM == 64 && (BN == 64 || BN == 80)) || (BM == 128 && BN == 80))) {
      // 128-wide K stages here too: DRN's 80-channel 3x3 convs 185 -> 225 TFLOP/s
      return special ? launch_one<PREC, BM, BN, WMV, WNV, false, true, true, 4>(p, s)
                     : launch_one<PREC, BM, BN, WMV, WNV, false, true, false, 4>(p, s);
    }
    return special ? launch_one<PREC, BM, BN, WMV, WNV, false, true, true>(p, s)
                   : launch_one<PREC, BM, BN, WMV, WNV, false, true, false>(p, s);
  }
  if (p.ln_g) {
    if constexpr (BM <= 64) {
      return special ? launch_one<PREC, BM, BN, WMV, WNV, true, false, true>(p, s)
                     : launch_one<PREC, BM, BN, WMV, WNV, true, false, false>(p, s);
    } else {
      return srad_set_error(SRAD_ERR_ARG, "gemm: LayerNorm fusion needs N > 32");
    }
  }
  if constexpr (PREC == SRAD_PREC_BF16 && BM == 64 && BN == 64) {
    // 128-wide K stages (half the LDS, four workgroups per CU, deeper load / MFMA overlap): 4 % on the training step
    return special ? launch_one<PREC, BM, BN, WMV, WNV, false, false, true, 4>(p, s)
                   : launch_one<PREC, BM, BN, WMV, WNV, false, false, false, 4>(p, s);
  }
  return special ? launch_one<PREC, BM, BN, WMV, WNV, false, false, true>(p, s)
                 : launch_one<PREC, BM, BN, WMV, WNV, false, false, false>(p, s);
}

// Tile choice: the problems are small (M = B*H*W rows, a few thousand), so prefer the largest tile
// that still yields >= ~1.5 workgroups per CU.
template <int PREC>
int launch_prec(const GemmParams& p, hipStream_t s) {
  auto tiles = [&](int bm, int bn) { return (long)((p.M + bm - 1) / bm) * ((p.N + bn - 1) / bn); };
  if (!p.ln_g) {
    if (p.N <= 16 && tiles(128, 16) >= 384) return launch_cfg<PREC, 128, 16, 4, 1>(p, s);
    if (p.N <= 32 && tiles(128, 32) >= 384) return launch_cfg<PREC, 128, 32, 4, 1>(p, s);
  }
  if (p.N <= 32) return launch_cfg<PREC, 32, 32, 2, 2>(p, s);
  // N = 65..80 (DRN's 80-channel RCAB convolutions): one 80-wide tile instead of a full and a mostly empty 64-wide one
  if constexpr (PREC == SRAD_PREC_BF16) {
    // ... and 128 rows per workgroup once that still gives every CU one (C3: 32768 pixels -> 256 workgroups in ONE round; the
    // kernel is a chain of load -> LDS -> MFMA stages per workgroup, so twice the rows per stage cost little more per stage)
    if (!p.ln_g && p.N > 64 && p.N <= 80 && tiles(128, 80) >= 256 && getenv("SRAD_GEMM_NO_BM128") == nullptr) return launch_cfg<PREC, 128, 80, 4, 1>(p, s);
  }
  if (!p.ln_g && p.N > 64 && p.N <= 80 && tiles(64, 80) >= 384) return launch_cfg<PREC, 64, 80, 4, 1>(p, s);
  if (tiles(64, 64) >= 384) return launch_cfg<PREC, 64, 64, 2, 2>(p, s);
  if (tiles(32, 64) >= 384) return launch_cfg<PREC, 32, 64, 2, 2>(p, s);
  return launch_cfg<PREC, 32, 32, 2, 2>(p, s);
}

}  // namespace

int srad_gemm_tile_rows(int prec, const GemmParams& p) {      // keep in step with launch_prec
  if (srad_conv80_supported(prec, p)) return 128;              // 4 x 32-pixel tiles of conv80_kernel
  auto tiles = [&](int bm, int bn) { return (long)((p.M + bm - 1) / bm) * ((p.N + bn - 1) / bn); };
  if (!p.ln_g) {
    if (p.N <= 16 && tiles(128, 16) >= 384) return 128;
    if (p.N <= 32 && tiles(128, 32) >= 384) return 128;
  }
  if (p.N <= 32) return 32;
  if (prec == SRAD_PREC_BF16 && !p.ln_g && p.N > 64 && p.N <= 80 && tiles(128, 80) >= 256 && getenv("SRAD_GEMM_NO_BM128") == nullptr) return 128;
  if (!p.ln_g && p.N > 64 && p.N <= 80 && tiles(64, 80) >= 384) return 64;
  if (tiles(64, 64) >= 384) return 64;
  return 32;
}

int srad_launch_gemm(int prec, const GemmParams& p, hipStream_t stream) {
  SRAD_REQUIRE(p.M > 0 && p.N > 0 && p.Cin > 0, "gemm: empty problem M=%d N=%d Cin=%d", p.M, p.N, p.Cin);
  SRAD_REQUIRE(p.Cp == srad_cp(p.Cin), "gemm: Cp %d does not match Cin %d", p.Cp, p.Cin);
  SRAD_REQUIRE(p.ntaps == 1 || p.ntaps == 9, "gemm: ntaps must be 1 or 9 (got %d)", p.ntaps);
  SRAD_REQUIRE(p.ps == 0 || (p.ps == 2 && p.N % 4 == 0), "gemm: pixel-shuffle needs N %% 4 == 0");
  SRAD_REQUIRE(!(p.ln_g && (p.ntaps != 1 || p.stride != 1)), "gemm: LayerNorm fusion only for row-identity A");
  SRAD_REQUIRE(!(p.ln_g && p.Cin > 320), "gemm: fused LayerNorm supports up to 320 channels (got %d); run srad_launch_layernorm first", p.Cin);
  SRAD_REQUIRE((double)p.M * (p.ntaps == 9 || p.stride != 1 ? p.stride * p.stride : 1) * p.ldx * 4.0 < 4.0e9 &&
                   (double)p.M * (p.ps == 2 ? 4 : 1) * p.ldy < 4.0e9,
               "gemm: tensors above 4 GB are not addressable with the 32-bit offsets this kernel uses");
  SRAD_REQUIRE((p.Cin & 3) == 0 && (p.ldx & 3) == 0 && ((uintptr_t)p.X & 15) == 0,
               "gemm: activations need a channel count and row stride that are multiples of 4 floats (Cin=%d ldx=%d)", p.Cin, p.ldx);
  if (p.ntaps == 9 || p.stride != 1 || p.ps)
    SRAD_REQUIRE(p.Ho > 0 && p.Wo > 0 && p.M % (p.Ho * p.Wo) == 0, "gemm: M=%d not a multiple of Ho*Wo=%d*%d", p.M, p.Ho, p.Wo);
  if (srad_conv80_supported(prec, p)) return srad_launch_conv80(p, stream);
  if (prec == SRAD_PREC_BF16 && p.ps == 2 && p.N == 320 && p.Cin == 80 && p.bias && getenv("SRAD_NO_UPCONV_SPLIT") == nullptr) {
    // DRN's Upsampler conv (80 -> 320 channels + PixelShuffle(2)): four 80 -> 80 convolutions through the weight-resident kernel,
    // one per sub-pixel position (rows 4 c + q of the pack), each writing its quarter of the 2x image - 217 us on this tiled
    // GEMM at 128 px x 8 against 4 x 26
    GemmParams q = p;
    q.ps = 0; q.N = 80; q.sp_q = 0;
    if (srad_conv80_supported(prec, q)) {
      for (int k = 0; k < 4; ++k) { q.sp_q = k; SRAD_TRY(srad_launch_conv80(q, stream)); }
      return SRAD_OK;
    }
  }
  if (srad_conv_thin_supported(prec, p)) return srad_launch_conv_thin(p, stream);
  if (srad_conv_tail_supported(prec, p)) return srad_launch_conv_tail(p, stream);
  SRAD_REQUIRE(!p.Xh && !p.Rh && (!p.Yh || p.hsplit_hd > 0), "gemm: bf16 activations in / out are the 80-channel conv kernel's (srad_conv80_supported)");
  int rc = prec == SRAD_PREC_BF16 ? launch_prec<SRAD_PREC_BF16>(p, stream)
           : prec == SRAD_PREC_BF16X3 ? launch_prec<SRAD_PREC_BF16X3>(p, stream) : launch_prec<SRAD_PREC_F32>(p, stream);
  if (rc) return rc;
  SRAD_CHECK_HIP(hipGetLastError());
  return SRAD_OK;
}

int srad_launch_pack_weight_padded(int prec, const float* src, void* dst, int n, int cin, int ntaps, int n_pad,
                                   int grp_real, int grp_pad, hipStream_t stream) {
  const int cin_pad = grp_pad > 0 ? (cin / grp_real) * grp_pad : cin;
  SRAD_REQUIRE(n_pad >= n && (grp_pad == 0 || (grp_real > 0 && grp_pad >= grp_real && cin % grp_real == 0)),
               "pack_weight: bad padding n=%d->%d cin=%d groups %d->%d", n, n_pad, cin, grp_real, grp_pad);
  const int Np = srad_np(n_pad), Cp = srad_cp(cin_pad);
  const size_t total = (size_t)Np * ntaps * Cp;
  const int blocks = (int)((total + 255) / 256 > 2048 ? 2048 : (total + 255) / 256);
  SradProfScope prof(stream, SRAD_K_PACK, 0.0, 4.0 * n * cin * ntaps + (prec == SRAD_PREC_BF16 ? 2.0 : 4.0) * total);
  if (prec == SRAD_PREC_BF16)
    hipLaunchKernelGGL((pack_weight_kernel<SRAD_PREC_BF16>), dim3(blocks), dim3(256), 0, stream, src, dst, n, cin, ntaps, Np, Cp, grp_real, grp_pad);
  else if (prec == SRAD_PREC_BF16X3)
    hipLaunchKernelGGL((pack_weight_kernel<SRAD_PREC_BF16X3>), dim3(blocks), dim3(256), 0, stream, src, dst, n, cin, ntaps, Np, Cp, grp_real, grp_pad);
  else
    hipLaunchKernelGGL((pack_weight_kernel<SRAD_PREC_F32>), dim3(blocks), dim3(256), 0, stream, src, dst, n, cin, ntaps, Np, Cp, grp_real, grp_pad);
  SRAD_CHECK_HIP(hipGetLastError());
  return SRAD_OK;
}

int srad_launch_pack_weight(int prec, const float* src, void* dst, int n, int cin, int ntaps, hipStream_t stream) {
  return srad_launch_pack_weight_padded(prec, src, dst, n, cin, ntaps, n, 0, 0, stream);
}

int srad_launch_pack_weight_transposed(int prec, const float* src, void* dst, int n, int cin, int ntaps, int n_pad,
                                       int cin_pad, hipStream_t stream) {
  SRAD_REQUIRE(n_pad >= n && cin_pad >= cin, "pack_weight_transposed: bad padding");
  const int Rp = srad_np(cin_pad), Kp = srad_cp(n_pad);
  const size_t total = (size_t)Rp * ntaps * Kp;
  const int blocks = (int)((total + 255) / 256 > 2048 ? 2048 : (total + 255) / 256);
  SradProfScope prof(stream, SRAD_K_PACK, 0.0, 4.0 * n * cin * ntaps + (prec == SRAD_PREC_BF16 ? 2.0 : 4.0) * total);
  if (prec == SRAD_PREC_BF16)
    hipLaunchKernelGGL((pack_weight_t_kernel<SRAD_PREC_BF16>), dim3(blocks), dim3(256), 0, stream, src, dst, n, cin, ntaps, Rp, Kp);
  else if (prec == SRAD_PREC_BF16X3)
    hipLaunchKernelGGL((pack_weight_t_kernel<SRAD_PREC_BF16X3>), dim3(blocks), dim3(256), 0, stream, src, dst, n, cin, ntaps, Rp, Kp);
  else
    hipLaunchKernelGGL((pack_weight_t_kernel<SRAD_PREC_F32>), dim3(blocks), dim3(256), 0, stream, src, dst, n, cin, ntaps, Rp, Kp);
  SRAD_CHECK_HIP(hipGetLastError());
  return SRAD_OK;
}

static int pack_frag(const float* src, void* dst, int n, int cin, bool lo, hipStream_t stream) {
  const int Np = srad_np(n), Kp = srad_cp(cin);
  const size_t total = (size_t)Np * Kp;
  const int blocks = (int)((total + 255) / 256 > 2048 ? 2048 : (total + 255) / 256);
  SradProfScope prof(stream, SRAD_K_PACK, 0.0, 4.0 * n * cin + 2.0 * total);
  if (lo) hipLaunchKernelGGL(pack_weight_frag_kernel<true>, dim3(blocks), dim3(256), 0, stream, src, reinterpret_cast<__bf16*>(dst), n, cin, Np, Kp, cin, 1);
  else hipLaunchKernelGGL(pack_weight_frag_kernel<false>, dim3(blocks), dim3(256), 0, stream, src, reinterpret_cast<__bf16*>(dst), n, cin, Np, Kp, cin, 1);
  SRAD_CHECK_HIP(hipGetLastError());
  return SRAD_OK;
}
int srad_launch_pack_weight_frag(const float* src, void* dst, int n, int cin, hipStream_t stream) { return pack_frag(src, dst, n, cin, false, stream); }
int srad_launch_pack_weight_frag_lo(const float* src, void* dst, int n, int cin, hipStream_t stream) { return pack_frag(src, dst, n, cin, true, stream); }

// the fragment-major pack of W^T (rows = input channels, k = output channels) from the same [n][cin] source
int srad_launch_pack_weight_frag_t(const float* src, void* dst, int n, int cin, hipStream_t stream) {
  const int Np = srad_np(srad_round_up(cin, 4)), Kp = srad_cp(srad_round_up(n, 4));
  const size_t total = (size_t)Np * Kp;
  const int blocks = (int)((total + 255) / 256 > 2048 ? 2048 : (total + 255) / 256);
  SradProfScope prof(stream, SRAD_K_PACK, 0.0, 4.0 * n * cin + 2.0 * total);
  hipLaunchKernelGGL(pack_weight_frag_kernel<false>, dim3(blocks), dim3(256), 0, stream, src, reinterpret_cast<__bf16*>(dst), cin, n, Np, Kp, 1, cin);
  SRAD_CHECK_HIP(hipGetLastError());
  return SRAD_OK;
}

static int pack_qkv(const float* src, void* dst, int d, int heads, bool lo, hipStream_t stream) {
  const int hd = d / heads, HDP = srad_qkv_hdp(d, heads), Kp = srad_cp(d);
  const size_t total = (size_t)heads * 3 * HDP * Kp;
  const int blocks = (int)((total + 255) / 256 > 2048 ? 2048 : (total + 255) / 256);
  SradProfScope prof(stream, SRAD_K_PACK, 0.0, 12.0 * d * d + 2.0 * total);
  if (lo) hipLaunchKernelGGL(pack_qkv_frag_kernel<true>, dim3(blocks), dim3(256), 0, stream, src, reinterpret_cast<__bf16*>(dst), d, heads, hd, HDP, Kp);
  else hipLaunchKernelGGL(pack_qkv_frag_kernel<false>, dim3(blocks), dim3(256), 0, stream, src, reinterpret_cast<__bf16*>(dst), d, heads, hd, HDP, Kp);
  SRAD_CHECK_HIP(hipGetLastError());
  return SRAD_OK;
}
int srad_launch_pack_qkv_frag(const float* src, void* dst, int d, int heads, hipStream_t stream) { return pack_qkv(src, dst, d, heads, false, stream); }
int srad_launch_pack_qkv_frag_lo(const float* src, void* dst, int d, int heads, hipStream_t stream) { return pack_qkv(src, dst, d, heads, true, stream); }
