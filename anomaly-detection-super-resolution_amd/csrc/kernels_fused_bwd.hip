// kernels_fused_bwd.hip - backward of the MLP branch of a DRCT Swin block as ONE launch (bf16 MFMA, fp32 accumulate):
//
//   x2 = x1 + rs2 * fc2(GELU(fc1(LayerNorm2(x1))))                 (src/drct.py:510, 184-190)
//
//   dh  = (dx2 . W2) * rs2 * gelu'(hpre)     [M][m]   written: fc1's weight gradient contracts it with LN2(x1)
//   dxn = dh . W1                             [M][d]   never leaves the CU
//   dx1 = dx2 + LayerNorm2'(dxn; x1, gamma)   [M][d]   dgamma / dbeta as per-workgroup partial rows (split-K queue)
//
// Same scheme as the forward kernel (kernels_fused.hip): a workgroup owns FM token rows for the whole chain, the A
// operands live in LDS as bf16, the two TRANSPOSED weight matrices stream from their fragment-major packs straight
// into MFMA registers three stages ahead, results are transposed so a lane owns 4 consecutive columns of a token row.
// It replaces two data-gradient GEMM launches and the LayerNorm backward launch of the unfused chain and the
// [M][m] + [M][d] round trips between them.
#include "srad_common.h"
bool srad_mlp_bwd_bf16_out(int M);
#include <type_traits>

namespace {

template <int B, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (B < N) {
    f(std::integral_constant<int, B>{});
    static_for<B + 1, N>(f);
  }
}

constexpr int FB_LDA = 400;        // LDS row stride of the <=384-wide bf16 dx2 tile
constexpr int FB_LDH = 528;        // LDS row stride of the <=512-wide bf16 dh tile
constexpr int FB_SC = 128;         // output columns per weight stage

// d/dx of the exact-erf GELU, Phi(x) + x phi(x), with erf by Abramowitz-Stegun 7.1.26 (|error| <= 1.5e-7)
__device__ __forceinline__ float dgelu_fast(float x) {
  const float z = fabsf(x) * 0.70710678118654752440f;
  const float t = 1.0f / (1.0f + 0.3275911f * z);
  const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
  const float ex = __expf(-z * z);                   // exp(-x^2 / 2)
  const float erfa = 1.0f - poly * ex;
  return 0.5f * (1.0f + (x < 0.f ? -erfa : erfa)) + x * 0.39894228040143267794f * ex;
}

// LayerNorm backward over the d real columns of FM token rows held in the transposed-result layout (lane: token row
// 16 rt + fr, columns 128 g + 16 wave + 4 fq + 0..3): dxn = gradient of the LayerNorm output, xv = its input rows,
// r2 = what is added to the result (residual path, previous content).  out rows = out0 + row * ld_out; the workgroup's
// dgamma | dbeta column sums go to prow.  Statistics exactly as the forward / unfused kernels (two-pass variance).
template <int NRT, int GD>
__device__ __forceinline__ void ln_bwd_tail(f32x4 (&dxn)[GD][NRT], f32x4 (&xv)[GD][NRT], const f32x4 (&r2)[GD][NRT],
                                            const float* v_g, float* red, int d, int wave, int fr, int fq, float* out0,
                                            int ld_out, float* prow, __bf16* a_out = nullptr, int lda_out = 0,
                                            __bf16* h_out0 = nullptr, const float* h_scale = nullptr) {
  const f32x4 z4 = f32x4{0.f, 0.f, 0.f, 0.f};
  auto col4_of = [&](int g) { return FB_SC * g + 16 * wave + 4 * fq; };
  const float invC = 1.0f / (float)d;
  auto row_reduce2 = [&](float (&a)[NRT], float (&b)[NRT]) {     // sums over the whole row: 4 fq groups x 8 waves
#pragma unroll
    for (int rt = 0; rt < NRT; ++rt) {
      a[rt] += __shfl_xor(a[rt], 16); b[rt] += __shfl_xor(b[rt], 16);
      a[rt] += __shfl_xor(a[rt], 32); b[rt] += __shfl_xor(b[rt], 32);
      if (fq == 0) {
        red[((rt * 16 + fr) * 8 + wave) * 2 + 0] = a[rt];
        red[((rt * 16 + fr) * 8 + wave) * 2 + 1] = b[rt];
      }
    }
    __syncthreads();
#pragma unroll
    for (int rt = 0; rt < NRT; ++rt) {
      const float* r = red + (rt * 16 + fr) * 16;
      float su = 0.f, s2 = 0.f;
#pragma unroll
      for (int w = 0; w < 8; ++w) { su += r[2 * w]; s2 += r[2 * w + 1]; }
      a[rt] = su; b[rt] = s2;
    }
  };
  float sm[NRT], sq[NRT];
#pragma unroll
  for (int rt = 0; rt < NRT; ++rt) { sm[rt] = 0.f; sq[rt] = 0.f; }
#pragma unroll
  for (int g = 0; g < GD; ++g) {
    const bool in = col4_of(g) < d;
#pragma unroll
    for (int rt = 0; rt < NRT; ++rt) {
      xv[g][rt] = in ? xv[g][rt] : z4;
      sm[rt] += (xv[g][rt][0] + xv[g][rt][1]) + (xv[g][rt][2] + xv[g][rt][3]);
    }
  }
  row_reduce2(sm, sq);
  float mu[NRT], rstd[NRT];
#pragma unroll
  for (int rt = 0; rt < NRT; ++rt) { mu[rt] = sm[rt] * invC; sq[rt] = 0.f; sm[rt] = 0.f; }
#pragma unroll
  for (int g = 0; g < GD; ++g) {
    const bool in = col4_of(g) < d;
#pragma unroll
    for (int rt = 0; rt < NRT; ++rt) {
      xv[g][rt] = in ? xv[g][rt] - mu[rt] : z4;
      sq[rt] += (xv[g][rt][0] * xv[g][rt][0] + xv[g][rt][1] * xv[g][rt][1]) + (xv[g][rt][2] * xv[g][rt][2] + xv[g][rt][3] * xv[g][rt][3]);
    }
  }
  __syncthreads();                                               // everyone has read the first sums
  row_reduce2(sq, sm);
#pragma unroll
  for (int rt = 0; rt < NRT; ++rt) rstd[rt] = rsqrtf(sq[rt] * invC + 1e-5f);
  float s1[NRT], s2[NRT];
  f32x4 dg[GD], db[GD];
#pragma unroll
  for (int rt = 0; rt < NRT; ++rt) { s1[rt] = 0.f; s2[rt] = 0.f; }
#pragma unroll
  for (int g = 0; g < GD; ++g) {
    const int c4 = col4_of(g);
    const bool in = c4 < d;
    const f32x4 gam = *reinterpret_cast<const f32x4*>(v_g + min(c4, 380));
    dg[g] = z4; db[g] = z4;
#pragma unroll
    for (int rt = 0; rt < NRT; ++rt) {
      const f32x4 dy = in ? dxn[g][rt] : z4;
      xv[g][rt] = xv[g][rt] * rstd[rt];                          // xhat
      dg[g] += dy * xv[g][rt];
      db[g] += dy;
      dxn[g][rt] = dy * gam;                                     // gy
      const f32x4 t = dxn[g][rt] * xv[g][rt];
      s1[rt] += (dxn[g][rt][0] + dxn[g][rt][1]) + (dxn[g][rt][2] + dxn[g][rt][3]);
      s2[rt] += (t[0] + t[1]) + (t[2] + t[3]);
    }
  }
  __syncthreads();
  row_reduce2(s1, s2);
#pragma unroll
  for (int g = 0; g < GD; ++g) {
    const int c4 = col4_of(g);
#pragma unroll
    for (int rt = 0; rt < NRT; ++rt) {
      const f32x4 o4 = (dxn[g][rt] - s1[rt] * invC - xv[g][rt] * (s2[rt] * invC)) * rstd[rt] + r2[g][rt];
      if (c4 < d) *reinterpret_cast<f32x4*>(out0 + (size_t)(rt * 16 + fr) * ld_out + c4) = o4;
      if (h_out0 && c4 < d) {                                    // and times a per-row factor as bf16 [rows][d]: a weight gradient's operand
        const f32x4 v = o4 * h_scale[rt];
        bf16x4 hq;
        hq[0] = (__bf16)v[0]; hq[1] = (__bf16)v[1]; hq[2] = (__bf16)v[2]; hq[3] = (__bf16)v[3];
        *reinterpret_cast<bf16x4*>(h_out0 + (size_t)(rt * 16 + fr) * d + c4) = hq;
      }
      if (a_out) {                                               // the result as the next GEMM's bf16 A tile (zero beyond d)
        const f32x4 v = c4 < d ? o4 : z4;
        bf16x4 h;
        h[0] = (__bf16)v[0]; h[1] = (__bf16)v[1]; h[2] = (__bf16)v[2]; h[3] = (__bf16)v[3];
        *reinterpret_cast<bf16x4*>(a_out + (rt * 16 + fr) * lda_out + c4) = h;
      }
    }
  }
  // dgamma / dbeta of this workgroup's rows: sum over the 16 token rows of the lane group, one partial row per workgroup
#pragma unroll
  for (int g = 0; g < GD; ++g) {
    const int c4 = col4_of(g);
#pragma unroll
    for (int e = 0; e < 4; ++e) { dg[g][e] = srad_row16_sum(dg[g][e]); db[g][e] = srad_row16_sum(db[g][e]); }
    if (fr == 0 && c4 < d) {
      *reinterpret_cast<f32x4*>(prow + c4) = dg[g];
      *reinterpret_cast<f32x4*>(prow + SRAD_LNB_CP + c4) = db[g];
    }
  }
}

// GD / GM = 128-column groups of the block dim / hidden, KGD / KGM = 256-wide k groups of the block dim / hidden
// KCA > 0: the adjust 1x1 conv's data gradient runs first, in the same launch (KCA = 32-wide k chunks of its output
// channels: 1 for the 32-channel adjust1-4, 6 for adjust5's 180): dx2 = alpha * (dA (.) lrelu'(y)) . Wadj is computed from
// the [FM][<=192] gradient tile, written out (fc2's weight gradient reads it) and kept on chip as this kernel's input.
// KCD / KCM = exact 32-wide k chunks of the block dim / hidden (a stage loads and multiplies only its real chunks).
// HOUT: dh (and, with the adjust prologue, the dx2 copy times its DropPath factor) leave as bf16 for the weight gradients
template <int FM, int GD, int KCD, int GM, int KCM, int KCA, bool HOUT = false>
__global__ __launch_bounds__(512) void mlp_bwd_kernel(const MlpBwdParams p, float* __restrict__ part) {
  constexpr int KGD = (KCD + 7) / 8, KGM = (KCM + 7) / 8;       // 256-wide k groups
  constexpr int NRT = FM / 16;
  constexpr int LDAA = KCA * 32 + 16;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  __bf16* A1 = reinterpret_cast<__bf16*>(smem);                  // [FM][FB_LDA] dx2
  __bf16* Hs = A1 + FM * FB_LDA;                                 // [FM][FB_LDH] dh
  float* v_g = reinterpret_cast<float*>(Hs + FM * FB_LDH);       // [384] gamma
  float* red = v_g + 384;                                        // [FM][8 waves][2] row partial sums
  [[maybe_unused]] __bf16* Aa = reinterpret_cast<__bf16*>(red + FM * 16);   // [FM][LDAA] gradient of the adjust conv's output

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 15, fq = lane >> 4;
  int tile_i = blockIdx.x;                                      // XCD affinity with the neighbouring kernels (see qkv_attn_kernel)
  if ((gridDim.x & 7) == 0 && !p.no_xcd_map) tile_i = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
  const int m0 = tile_i * FM;
  const int d = p.d, m = p.m;
  constexpr int n_adj = KCA > 0 ? GD : 0, n_fc2 = GM * KGD, n_fc1 = GD * KGM, n_stages = n_adj + n_fc2 + n_fc1;

  // waves whose 16 output columns are all padding skip their weight loads and MFMAs (see mlp_block_kernel)
  const int wave_s = __builtin_amdgcn_readfirstlane(wave);
  // stage s: phase (-1 adjust, 0 fc2^T, 1 fc1^T), 128-column group g, 256-wide k group kg - all compile-time
  struct StageGeo { int ph, g, kg, nch, kc; };
  auto geo = [](int s) constexpr -> StageGeo {
    if (s > n_stages - 1) s = n_stages - 1;
    int ph = 0, kgs = 1, kc = 1;
    if (s < n_adj) { ph = -1; kgs = 1; kc = KCA; }
    else if ((s -= n_adj) < n_fc2) { ph = 0; kgs = KGD; kc = KCD; }
    else { s -= n_fc2; ph = 1; kgs = KGM; kc = KCM; }
    const int g = s / kgs, kg = s - g * kgs;
    const int nch = kc - kg * 8 < 8 ? kc - kg * 8 : 8;
    return StageGeo{ph, g, kg, nch, kc};
  };
  constexpr int NSETS = 3;
  u32x4 w_reg[NSETS][8];
  auto load_w = [&](auto S, u32x4 (&reg)[8]) {
    constexpr StageGeo sg = geo(decltype(S)::value);
    const char* w = (const char*)(sg.ph == -1 ? p.w_adjt : (sg.ph == 0 ? p.w_fc2t : p.w_fc1t));
    const int nreal = sg.ph == 0 ? m : d;
    // unconditional (a load inside a branch makes hipcc's wait-count pass drain every older load at the join, see
    // mlp_block_kernel): a wave without real columns in the stage reads one 16-byte word per load instead
    const bool live = (sg.g * 8 + wave_s) * 16 < nreal;
    const char* base = live ? w + ((size_t)(sg.g * 8 + wave) * sg.kc + sg.kg * 8) * 1024 + fr * 64 + fq * 16 : w;
    const int step = live ? 1024 : 0;
#pragma unroll
    for (int cc = 0; cc < sg.nch; ++cc) reg[cc] = *reinterpret_cast<const u32x4*>(base + cc * step);
  };
  auto mma_stage = [&](const __bf16* A, int lda, int k0, int nch, const u32x4 (&reg)[8], f32x4 (&c)[NRT]) {
    const __bf16* ar = A + fr * lda + k0 + 8 * fq;
    // the fragments of G chunks in flight, THEN their MFMAs (see mlp_block_kernel: the scheduler sinks each read to its use)
    constexpr int G = NRT == 1 ? 8 : (NRT == 2 ? 4 : 2);
#pragma unroll
    for (int cc0 = 0; cc0 < 8; cc0 += G) {
      bf16x8 af[G][NRT];
#pragma unroll
      for (int g = 0; g < G; ++g)
        if (cc0 + g < nch) {
#pragma unroll
          for (int rt = 0; rt < NRT; ++rt) af[g][rt] = *reinterpret_cast<const bf16x8*>(ar + rt * 16 * lda + (cc0 + g) * 32);
        }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int g = 0; g < G; ++g)
        if (cc0 + g < nch) {
          const bf16x8 b0 = __builtin_bit_cast(bf16x8, reg[cc0 + g]);
#pragma unroll
          for (int rt = 0; rt < NRT; ++rt) c[rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b0, af[g][rt], c[rt], 0, 0, 0);
        }
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  auto col4_of = [&](int g) { return FB_SC * g + 16 * wave + 4 * fq; };
  auto store_bf4 = [&](__bf16* base, int ld, int rt, int c4, f32x4 v) {
    bf16x4 h;
    h[0] = (__bf16)v[0]; h[1] = (__bf16)v[1]; h[2] = (__bf16)v[2]; h[3] = (__bf16)v[3];
    *reinterpret_cast<bf16x4*>(base + (rt * 16 + fr) * ld + c4) = h;
  };
  const f32x4 z4 = f32x4{0.f, 0.f, 0.f, 0.f};

  // ---- independent loads: dx2 tile (A operand layout), first weight stages, gamma, dx2 / x1 in the result layout ----
  constexpr int NAQ = KCA > 0 ? 8 * KCA : 32 * GD;            // float4 per row of the tile read from global memory
  constexpr int NAJ = (FM * NAQ + 511) / 512;
  f32x4 a_reg[NAJ];
  [[maybe_unused]] f32x4 y_reg[NAJ];
#pragma unroll
  for (int j = 0; j < NAJ; ++j) {
    const int idx = min(tid + 512 * j, FM * NAQ - 1), row = idx / NAQ, c = (idx - row * NAQ) * 4;
    if constexpr (KCA > 0) {
      a_reg[j] = *reinterpret_cast<const f32x4*>(p.dA + (size_t)(m0 + row) * p.ld_dA + min(c, p.KA - 4));
      // unconditional load (from dA itself when there is no activation: the values are not used then)
      typedef const f32x4 __attribute__((address_space(1)))* gf4_p;     // (C-style cast: a pointer select is generic to hipcc)
      y_reg[j] = *(gf4_p)((p.y_act ? p.y_act + (size_t)(m0 + row) * p.ld_y : p.dA + (size_t)(m0 + row) * p.ld_dA) + min(c, p.KA - 4));
    } else {
      a_reg[j] = *reinterpret_cast<const f32x4*>(p.dx2 + (size_t)(m0 + row) * d + min(c, d - 4));
    }
  }
  // (the weight stages are issued BEHIND the activations: vmcnt retires in issue order and the CU's load queue is
  //  first-in first-out, see mlp_block_kernel)
  const float gq = p.ln_g[min(tid, d - 1)];
  float rs2v[NRT];
#pragma unroll
  for (int rt = 0; rt < NRT; ++rt) {
    typedef const float __attribute__((address_space(1)))* gfloat_p;   // a select of two kernel-argument pointers is a generic pointer to hipcc: flat loads, vmcnt(0) everywhere
    const float r = ((gfloat_p)(p.rs2 ? p.rs2 : p.ln_g))[p.rs2 ? (m0 + rt * 16 + fr) / p.rps : 0];
    rs2v[rt] = p.rs2 ? r : 1.f;
  }
  f32x4 r2[GD][NRT], xv[GD][NRT];
#pragma unroll
  for (int g = 0; g < GD; ++g) {
    const int c4 = min(col4_of(g), d - 4);
#pragma unroll
    for (int rt = 0; rt < NRT; ++rt) {
      if constexpr (KCA == 0) r2[g][rt] = *reinterpret_cast<const f32x4*>(p.dx2 + (size_t)(m0 + rt * 16 + fr) * d + c4);
      xv[g][rt] = *reinterpret_cast<const f32x4*>(p.x1 + (size_t)(m0 + rt * 16 + fr) * d + c4);
    }
  }
  // every group's fc1 pre-activations now: they come from HBM (written a whole forward ago) and a group is only one
  // or two stages long, so a load issued at its start would be waited for in full at its end
  typedef typename std::conditional<HOUT, u32x2, f32x4>::type hp_t;     // the bf16-output instances read it as bf16 (half the
  hp_t hp[GM][NRT];                                                      // bytes, half of the registers held across the kernel)
#pragma unroll
  for (int g = 0; g < GM; ++g) {
    const int c4 = min(col4_of(g), m - 4);
#pragma unroll
    for (int rt = 0; rt < NRT; ++rt) {
      if constexpr (HOUT) hp[g][rt] = *reinterpret_cast<const u32x2*>(p.hpre_h + (size_t)(m0 + rt * 16 + fr) * m + c4);
      else hp[g][rt] = *reinterpret_cast<const f32x4*>(p.hpre + (size_t)(m0 + rt * 16 + fr) * m + c4);
    }
  }
  load_w(std::integral_constant<int, 0>{}, w_reg[0]);
  if (tid < 384) v_g[tid] = gq;
#pragma unroll
  for (int j = 0; j < NAJ; ++j) {
    const int idx = tid + 512 * j, row = idx / NAQ, c = (idx - row * NAQ) * 4;
    if (idx < FM * NAQ) {
      bf16x4 h;
      if constexpr (KCA > 0) {           // dA = dY (.) lrelu'(y): kept for the adjust conv's weight gradient, bf16 tile in LDS
        f32x4 v = a_reg[j];
        if (p.y_act) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = y_reg[j][e] > 0.f ? v[e] : v[e] * p.slope;
        }
        v = c < p.KA ? v : z4;
        h[0] = (__bf16)v[0]; h[1] = (__bf16)v[1]; h[2] = (__bf16)v[2]; h[3] = (__bf16)v[3];
        if (p.dA_out_h) { if (c < p.KA) *reinterpret_cast<bf16x4*>(p.dA_out_h + (size_t)(m0 + row) * p.KA + c) = h; }
        else if (p.dA_out && c < p.KA) *reinterpret_cast<f32x4*>(p.dA_out + (size_t)(m0 + row) * p.KA + c) = v;
        *reinterpret_cast<bf16x4*>(Aa + row * LDAA + c) = h;
      } else {
        const f32x4 v = c < d ? a_reg[j] : z4;
        h[0] = (__bf16)v[0]; h[1] = (__bf16)v[1]; h[2] = (__bf16)v[2]; h[3] = (__bf16)v[3];
        *reinterpret_cast<bf16x4*>(A1 + row * FB_LDA + c) = h;
      }
    }
  }
  static_for<1, NSETS>([&](auto Q) { load_w(Q, w_reg[decltype(Q)::value]); });   // the rest of the weight look-ahead

  f32x4 dxn[GD][NRT];
  f32x4 c[NRT];
  static_for<0, n_stages>([&](auto S) {
    constexpr int s = decltype(S)::value;
    constexpr int ph = s < n_adj ? -1 : (s < n_adj + n_fc2 ? 0 : 1);
    constexpr int ls = s - (ph == -1 ? 0 : (ph == 0 ? n_adj : n_adj + n_fc2));
    constexpr int kgs = ph == -1 ? 1 : (ph == 0 ? KGD : KGM);
    constexpr int g = ls / kgs, kg = ls - g * kgs;
    u32x4 (&reg)[8] = w_reg[s % NSETS];
    constexpr int kc = ph == -1 ? KCA : (ph == 0 ? KCD : KCM);
    constexpr int nch = kc - kg * 8 < 8 ? kc - kg * 8 : 8;
    if constexpr (ls == 0) __syncthreads();            // the phase's activation tile (Aa / A1 / Hs) is complete
    if constexpr (kg == 0) {
#pragma unroll
      for (int rt = 0; rt < NRT; ++rt) c[rt] = z4;
    }
    if ((g * 8 + wave_s) * 16 < (ph == 0 ? m : d))
      mma_stage(ph == -1 ? Aa : (ph == 0 ? A1 : Hs), ph == -1 ? LDAA : (ph == 0 ? FB_LDA : FB_LDH), kg * 256, nch, reg, c);
    load_w(std::integral_constant<int, s + NSETS>{}, reg);
    if constexpr (kg == kgs - 1) {
      if constexpr (ph == -1) {                        // dx2 = alpha * dA . Wadj: residual registers, global copy, bf16 A tile
        const int c4 = col4_of(g);
#pragma unroll
        for (int rt = 0; rt < NRT; ++rt) {
          const f32x4 v = c4 < d ? c[rt] * p.aalpha : z4;
          r2[g][rt] = v;
          if constexpr (HOUT) {                        // bf16, times the DropPath factor: all fc2's weight gradient needs
            if (c4 < d) {
              const f32x4 vs = v * rs2v[rt];
              bf16x4 hq;
              hq[0] = (__bf16)vs[0]; hq[1] = (__bf16)vs[1]; hq[2] = (__bf16)vs[2]; hq[3] = (__bf16)vs[3];
              *reinterpret_cast<bf16x4*>(p.dx2s_h + (size_t)(m0 + rt * 16 + fr) * d + c4) = hq;
            }
          } else if (c4 < d) {
            *reinterpret_cast<f32x4*>(p.dx2 + (size_t)(m0 + rt * 16 + fr) * d + c4) = v;
          }
          store_bf4(A1, FB_LDA, rt, c4, v);
        }
      } else if constexpr (ph == 0) {
        const int c4 = col4_of(g);
#pragma unroll
        for (int rt = 0; rt < NRT; ++rt) {
          f32x4 v, hv;
          if constexpr (HOUT) {
            const bf16x4 hb = __builtin_bit_cast(bf16x4, hp[g][rt]);
            hv = f32x4{(float)hb[0], (float)hb[1], (float)hb[2], (float)hb[3]};
          } else {
            hv = hp[g][rt];
          }
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = c[rt][e] * rs2v[rt] * dgelu_fast(hv[e]);
          v = c4 < m ? v : z4;                         // m % 4 == 0
          if constexpr (HOUT) {
            if (c4 < m) {
              bf16x4 hq;
              hq[0] = (__bf16)v[0]; hq[1] = (__bf16)v[1]; hq[2] = (__bf16)v[2]; hq[3] = (__bf16)v[3];
              *reinterpret_cast<bf16x4*>(p.dh_h + (size_t)(m0 + rt * 16 + fr) * m + c4) = hq;
            }
          } else if (c4 < m) {
            *reinterpret_cast<f32x4*>(p.dh + (size_t)(m0 + rt * 16 + fr) * m + c4) = v;
          }
          store_bf4(Hs, FB_LDH, rt, c4, v);
        }
      } else {
#pragma unroll
        for (int rt = 0; rt < NRT; ++rt) dxn[g][rt] = c[rt];
      }
    }
  });

  // ---- optional epilogue phase: the attention projection's data gradient dO = (dx1 . Wproj) * rs1 from the dx1 rows
  //      just produced (bf16 tile in A1); its first weight stages are in flight during the LayerNorm arithmetic ----
  constexpr int n_proj = GD * KGD;
  auto load_wp = [&](auto S, u32x4 (&reg)[8]) {
    constexpr int sc = decltype(S)::value < n_proj - 1 ? decltype(S)::value : n_proj - 1;
    constexpr int g = sc / KGD, kg = sc - g * KGD;
    constexpr int nch = KCD - kg * 8 < 8 ? KCD - kg * 8 : 8;
    const bool live = (g * 8 + wave_s) * 16 < d;
    const char* base = live ? (const char*)p.w_projt + ((size_t)(g * 8 + wave) * KCD + kg * 8) * 1024 + fr * 64 + fq * 16 : (const char*)p.w_projt;
    const int step = live ? 1024 : 0;
#pragma unroll
    for (int cc = 0; cc < nch; ++cc) reg[cc] = *reinterpret_cast<const u32x4*>(base + cc * step);
  };
  float rs1v[NRT];
  if (p.w_projt) {
    static_for<0, NSETS>([&](auto Q) { load_wp(Q, w_reg[decltype(Q)::value]); });
#pragma unroll
    for (int rt = 0; rt < NRT; ++rt) {
      typedef const float __attribute__((address_space(1)))* gfloat_p;
      const float r = ((gfloat_p)(p.rs1 ? p.rs1 : p.ln_g))[p.rs1 ? (m0 + rt * 16 + fr) / p.rps : 0];
      rs1v[rt] = p.rs1 ? r : 1.f;
    }
  }
  ln_bwd_tail<NRT, GD>(dxn, xv, r2, v_g, red, d, wave, fr, fq, p.dx1 + (size_t)m0 * d, d, part + (size_t)tile_i * (2 * SRAD_LNB_CP),
                       p.w_projt ? A1 : nullptr, FB_LDA, (HOUT && p.w_projt && p.dx1s_h) ? p.dx1s_h + (size_t)m0 * d : nullptr, rs1v);
  if (p.w_projt) {
    __syncthreads();                                             // the dx1 tile is complete
    static_for<0, n_proj>([&](auto S) {
      constexpr int s = decltype(S)::value;
      constexpr int g = s / KGD, kg = s - g * KGD;
      u32x4 (&reg)[8] = w_reg[s % NSETS];
      constexpr int nch = KCD - kg * 8 < 8 ? KCD - kg * 8 : 8;
      if constexpr (kg == 0) {
#pragma unroll
        for (int rt = 0; rt < NRT; ++rt) c[rt] = z4;
      }
      if ((g * 8 + wave_s) * 16 < d) mma_stage(A1, FB_LDA, kg * 256, nch, reg, c);
      load_wp(std::integral_constant<int, s + NSETS>{}, reg);
      if constexpr (kg == KGD - 1) {
        const int c4 = col4_of(g);
        if (HOUT && p.dO_h) {
          // bf16, one hp-wide slot per head (what the all-bf16 attention backward stages with 16-byte loads); the head dim is
          // even, so the column pairs (c4, c4 + 1) and (c4 + 2, c4 + 3) never straddle a head
          if (c4 < d) {
            const int hdo = d / p.dO_heads;
            const int ldo = p.dO_heads * p.dO_hp;
            if (hdo & 1) {                                       // odd head dim (53, 77): a pair can straddle two heads' slots
              int o[4];
#pragma unroll
              for (int e = 0; e < 4; ++e) { const int hh = (c4 + e) / hdo; o[e] = hh * p.dO_hp + (c4 + e - hh * hdo); }
#pragma unroll
              for (int rt = 0; rt < NRT; ++rt) {
                const f32x4 v = c[rt] * rs1v[rt];
                __bf16* row = p.dO_h + (size_t)(m0 + rt * 16 + fr) * ldo;
#pragma unroll
                for (int e = 0; e < 4; ++e) row[o[e]] = (__bf16)v[e];
              }
            } else {
              const int h0 = c4 / hdo, h1 = (c4 + 2) / hdo;
              const int o0 = h0 * p.dO_hp + (c4 - h0 * hdo), o1 = h1 * p.dO_hp + (c4 + 2 - h1 * hdo);
#pragma unroll
              for (int rt = 0; rt < NRT; ++rt) {
                const f32x4 v = c[rt] * rs1v[rt];
                __bf16* row = p.dO_h + (size_t)(m0 + rt * 16 + fr) * ldo;
                *reinterpret_cast<bf16x2*>(row + o0) = bf16x2{(__bf16)v[0], (__bf16)v[1]};
                *reinterpret_cast<bf16x2*>(row + o1) = bf16x2{(__bf16)v[2], (__bf16)v[3]};
              }
            }
          }
        } else if (c4 < d) {
#pragma unroll
          for (int rt = 0; rt < NRT; ++rt)
            *reinterpret_cast<f32x4*>(p.dO + (size_t)(m0 + rt * 16 + fr) * d + c4) = c[rt] * rs1v[rt];
        }
      }
    });
  }
}

// ------------------------------------------------------------------------------------------
// dX = dY . W (K up to 4 x 256) + LayerNorm backward + residual: the qkv Linear / LayerNorm1 end of a Swin block.
// GD = 128-column groups of d, KG = 256-wide k groups of K.
// ------------------------------------------------------------------------------------------
// YH: dY is stored as bf16 (compile-time: a run-time branch around the tile's loads drains the wait counts at its join)
template <int FM, int GD, int KC, bool YH = false>
__global__ __launch_bounds__(512) void lin_ln_bwd_kernel(const LinLnBwdParams p, float* __restrict__ part) {
  constexpr int KG = (KC + 7) / 8;                               // KC = exact 32-wide k chunks of K
  constexpr int NRT = FM / 16;
  constexpr int LDA = KG * 256 + 16;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  __bf16* A1 = reinterpret_cast<__bf16*>(smem);                  // [FM][LDA] dY
  float* v_g = reinterpret_cast<float*>(A1 + FM * LDA);          // [384] gamma
  float* red = v_g + 384;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 15, fq = lane >> 4;
  int tile_i = blockIdx.x;                                      // XCD affinity with the neighbouring kernels (see qkv_attn_kernel)
  if ((gridDim.x & 7) == 0 && !p.no_xcd_map) tile_i = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
  const int m0 = tile_i * FM;
  const int d = p.d, K = p.K;
  constexpr int n_stages = GD * KG;
  const f32x4 z4 = f32x4{0.f, 0.f, 0.f, 0.f};
  const int wave_s = __builtin_amdgcn_readfirstlane(wave);

  constexpr int NSETS = 3;
  u32x4 w_reg[NSETS][8];
  auto load_w = [&](auto S, u32x4 (&reg)[8]) {
    constexpr int sc = decltype(S)::value < n_stages - 1 ? decltype(S)::value : n_stages - 1;
    constexpr int g = sc / KG, kg = sc - g * KG;
    constexpr int nch = KC - kg * 8 < 8 ? KC - kg * 8 : 8;
    const bool live = (g * 8 + wave_s) * 16 < d;      // all-padding column tiles: one 16-byte word per load, no MFMAs (see mlp_block_kernel)
    const char* base = live ? (const char*)p.w_t + ((size_t)(g * 8 + wave) * KC + kg * 8) * 1024 + fr * 64 + fq * 16 : (const char*)p.w_t;
    const int step = live ? 1024 : 0;
#pragma unroll
    for (int cc = 0; cc < nch; ++cc) reg[cc] = *reinterpret_cast<const u32x4*>(base + cc * step);
  };
  auto col4_of = [&](int g) { return FB_SC * g + 16 * wave + 4 * fq; };
  // loads in the order their data is needed, all unconditional (see mlp_block_kernel): residual / input rows, gamma, the dY
  // tile, then the weight stages
  const float gq = p.ln_g[min(tid, d - 1)];
  f32x4 r2[GD][NRT], xv[GD][NRT];
#pragma unroll
  for (int g = 0; g < GD; ++g) {
    const int c4 = min(col4_of(g), d - 4);
#pragma unroll
    for (int rt = 0; rt < NRT; ++rt) {
      const size_t row = (size_t)(m0 + rt * 16 + fr);
      typedef const f32x4 __attribute__((address_space(1)))* gf4_p;
      const f32x4 rd = *(gf4_p)(p.dres ? p.dres + row * p.ld_dres + c4 : p.x + row * p.ldx + c4);
      const f32x4 ro = *(gf4_p)(p.accumulate ? p.out + row * p.ld_out + c4 : p.x + row * p.ldx + c4);
      r2[g][rt] = (p.dres ? rd : z4) + (p.accumulate ? ro : z4);
      xv[g][rt] = *reinterpret_cast<const f32x4*>(p.x + row * p.ldx + c4);
    }
  }
  // dY tile -> bf16 -> A1 (zero beyond K): KQ float4 per row, eight loads in flight per thread
  {
    const int KQ = KG * 64;
    const int total = FM * KQ;
    for (int i0 = tid; i0 < total; i0 += 512 * 8) {
      [[maybe_unused]] f32x4 v[8];
      [[maybe_unused]] u32x2 vh[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int idx = min(i0 + 512 * u, total - 1), row = idx / KQ, c = (idx - row * KQ) * 4;
        const size_t e = (size_t)(m0 + row) * p.ld_dy + min(c, K - 4);
        if constexpr (YH) vh[u] = *reinterpret_cast<const u32x2*>(reinterpret_cast<const __bf16*>(p.dY) + e);
        else v[u] = *reinterpret_cast<const f32x4*>(p.dY + e);
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int idx = i0 + 512 * u;
        if (idx < total) {
          const int row = idx / KQ, c = (idx - row * KQ) * 4;
          if constexpr (YH) {
            *reinterpret_cast<u32x2*>(A1 + row * LDA + c) = c < K ? vh[u] : u32x2{0u, 0u};
          } else {
            const f32x4 w = c < K ? v[u] : z4;
            bf16x4 h;
            h[0] = (__bf16)w[0]; h[1] = (__bf16)w[1]; h[2] = (__bf16)w[2]; h[3] = (__bf16)w[3];
            *reinterpret_cast<bf16x4*>(A1 + row * LDA + c) = h;
          }
        }
      }
    }
  }
  if (tid < 384) v_g[tid] = gq;
  static_for<0, NSETS>([&](auto Q) { load_w(Q, w_reg[decltype(Q)::value]); });
  __syncthreads();

  f32x4 dxn[GD][NRT];
  f32x4 c[NRT];
  static_for<0, n_stages>([&](auto S) {
    constexpr int s = decltype(S)::value;
    constexpr int g = s / KG, kg = s - g * KG;
    u32x4 (&reg)[8] = w_reg[s % NSETS];
    constexpr int nch = KC - kg * 8 < 8 ? KC - kg * 8 : 8;
    if constexpr (kg == 0) {
#pragma unroll
      for (int rt = 0; rt < NRT; ++rt) c[rt] = z4;
    }
    if ((g * 8 + wave_s) * 16 < d) {
      const __bf16* ar = A1 + fr * LDA + kg * 256 + 8 * fq;
      constexpr int G = NRT == 1 ? 8 : (NRT == 2 ? 4 : 2);
#pragma unroll
      for (int cc0 = 0; cc0 < 8; cc0 += G) {
        bf16x8 af[G][NRT];
#pragma unroll
        for (int gg = 0; gg < G; ++gg)
          if (cc0 + gg < nch) {
#pragma unroll
            for (int rt = 0; rt < NRT; ++rt) af[gg][rt] = *reinterpret_cast<const bf16x8*>(ar + rt * 16 * LDA + (cc0 + gg) * 32);
          }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int gg = 0; gg < G; ++gg)
          if (cc0 + gg < nch) {
            const bf16x8 b0 = __builtin_bit_cast(bf16x8, reg[cc0 + gg]);
#pragma unroll
            for (int rt = 0; rt < NRT; ++rt) c[rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b0, af[gg][rt], c[rt], 0, 0, 0);
          }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    load_w(std::integral_constant<int, s + NSETS>{}, reg);
    if constexpr (kg == KG - 1) {
#pragma unroll
      for (int rt = 0; rt < NRT; ++rt) dxn[g][rt] = c[rt];
    }
  });
  ln_bwd_tail<NRT, GD>(dxn, xv, r2, v_g, red, d, wave, fr, fq, p.out + (size_t)m0 * p.ld_out, p.ld_out,
                       part + (size_t)tile_i * (2 * SRAD_LNB_CP));
}

template <int FM, int GD, int KC>
int launch_lin_fm(const LinLnBwdParams& p, WgradQueue& q, hipStream_t stream) {
  constexpr int KG = (KC + 7) / 8;
  constexpr size_t lds = (size_t)FM * (KG * 256 + 16) * 2 + (384 + FM * 16) * sizeof(float);
  auto kern = p.dy_bf16 ? lin_ln_bwd_kernel<FM, GD, KC, true> : lin_ln_bwd_kernel<FM, GD, KC, false>;
  static SradOncePerDevice configured[2];
  if (configured[p.dy_bf16 ? 1 : 0].need()) {
    SRAD_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    configured[p.dy_bf16 ? 1 : 0].done();
  }
  float* part = nullptr;
  SRAD_TRY(srad_wgrad_queue_ln_partials(q, p.dgamma, p.dbeta, p.d, p.M / FM, stream, &part));
  SradProfScope prof(stream, SRAD_K_MLP_BWD, 2.0 * p.M * p.K * p.d, 4.0 * p.M * ((double)p.K + 4.0 * p.d) + 2.0 * p.K * p.d);
  hipLaunchKernelGGL(kern, dim3(p.M / FM), dim3(512), lds, stream, p, part);
  SRAD_CHECK_HIP(hipGetLastError());
  return SRAD_OK;
}
template <int GD, int KC>
int launch_lin(const LinLnBwdParams& p, WgradQueue& q, hipStream_t stream) {
  if (p.M >= 8192 && p.M % 32 == 0) return launch_lin_fm<32, GD, KC>(p, q, stream);
  return launch_lin_fm<16, GD, KC>(p, q, stream);
}
// (128-column groups of d, 32-wide k chunks of K = 3 d) of DRCT-L's Swin blocks: d = 180, 212, 244, 276, 308
#define SRAD_LIN_CFGS(X) X(2, 17) X(2, 20) X(2, 23) X(3, 26) X(3, 29)

struct BwdCfg { int gd, kcd, gm, kcm, kca; };
inline BwdCfg bwd_cfg(int d, int m, int KA) {
  return BwdCfg{(d + FB_SC - 1) / FB_SC, srad_cp(d) / 32, (m + FB_SC - 1) / FB_SC, srad_cp(m) / 32, (KA + 31) / 32};
}
template <int FM, int GD, int KCD, int GM, int KCM, int KCA, bool HOUT = false>
int launch_bwd_fm(const MlpBwdParams& p, WgradQueue& q, hipStream_t stream) {
  constexpr size_t lds = (size_t)(FM * FB_LDA + FM * FB_LDH) * 2 + (384 + FM * 16) * sizeof(float) +
                         (KCA > 0 ? (size_t)FM * (KCA * 32 + 16) * 2 : 0);
  auto kern = mlp_bwd_kernel<FM, GD, KCD, GM, KCM, KCA, HOUT>;
  static SradOncePerDevice configured;
  if (configured.need()) {
    SRAD_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    configured.done();
  }
  float* part = nullptr;
  SRAD_TRY(srad_wgrad_queue_ln_partials(q, p.dgamma, p.dbeta, p.d, p.M / FM, stream, &part));
  SradProfScope prof(stream, SRAD_K_MLP_BWD, 8.0 * p.M * p.d * p.m + (KCA > 0 ? 2.0 * p.M * p.d * p.KA : 0.0),
                     4.0 * p.M * (3.0 * p.d + 2.0 * p.m) + 4.0 * p.d * p.m);
  hipLaunchKernelGGL(kern, dim3(p.M / FM), dim3(512), lds, stream, p, part);
  SRAD_CHECK_HIP(hipGetLastError());
  return SRAD_OK;
}
template <int GD, int KCD, int GM, int KCM, int KCA>
int launch_bwd(const MlpBwdParams& p, WgradQueue& q, hipStream_t stream) {
  if (p.dh_h) {                                  // bf16 outputs for the weight gradients: the 32-row instances (srad_mlp_bwd_bf16_out)
    SRAD_REQUIRE(srad_mlp_bwd_bf16_out(p.M) && (KCA == 0 || p.dx2s_h) && p.hpre_h && ((uintptr_t)p.hpre_h & 7) == 0,
                 "mlp_bwd: bf16 outputs need M >= 8192, M %% 32 == 0, the fc1 pre-activation as bf16 (and the dx2 copy with the adjust prologue)");
    SRAD_REQUIRE(!p.w_projt || (p.dO_h ? (p.dO_heads > 0 && p.d % p.dO_heads == 0 && p.dO_hp % 8 == 0 &&
                                          p.dO_hp >= p.d / p.dO_heads)
                                       : p.dO != nullptr),
                 "mlp_bwd: dO goes out as fp32 [M][d], or as bf16 per head ([M][heads][hp], hp %% 8 == 0)");
    return launch_bwd_fm<32, GD, KCD, GM, KCM, KCA, true>(p, q, stream);
  }
  if (p.M >= 8192 && p.M % 32 == 0) return launch_bwd_fm<32, GD, KCD, GM, KCM, KCA>(p, q, stream);
  return launch_bwd_fm<16, GD, KCD, GM, KCM, KCA>(p, q, stream);
}
// stage geometries of DRCT-L's Swin blocks (embed 180 + k*32; mlp ratio 2, 2, 2, 1, 1); last number: 32-wide chunks of
// the adjust conv's output channels when its data gradient is part of the launch (adjust1-4: 32, adjust5: 180), 0 = not
#define SRAD_BWD_CFGS(X) \
  X(2, 6, 3, 12, 0) X(2, 7, 4, 14, 0) X(2, 8, 4, 16, 0) X(3, 9, 3, 9, 0) X(3, 10, 3, 10, 0) \
  X(2, 6, 3, 12, 1) X(2, 7, 4, 14, 1) X(2, 8, 4, 16, 1) X(3, 9, 3, 9, 1) X(3, 10, 3, 10, 6)
}  // namespace

bool srad_mlp_bwd_bf16_out(int M) { return M >= 8192 && M % 32 == 0; }

bool srad_mlp_bwd_supported(int prec, int M, int d, int m, int KA) {
  if (!(prec == SRAD_PREC_BF16 && M % 16 == 0 && d % 4 == 0 && m % 4 == 0 && d >= 32 && d <= SRAD_LNB_CP && m >= 32 && m <= 512 &&
        KA >= 0 && KA % 4 == 0))
    return false;
  const BwdCfg c = bwd_cfg(d, m, KA);
#define X(a, b, cc, dd, e) if (c.gd == a && c.kcd == b && c.gm == cc && c.kcm == dd && c.kca == e) return true;
  SRAD_BWD_CFGS(X)
#undef X
  return false;
}

int srad_launch_mlp_bwd(const MlpBwdParams& p, WgradQueue& q, hipStream_t stream) {
  SRAD_REQUIRE(srad_mlp_bwd_supported(SRAD_PREC_BF16, p.M, p.d, p.m, p.KA), "mlp_bwd: unsupported shape M=%d d=%d m=%d KA=%d", p.M, p.d, p.m, p.KA);
  SRAD_REQUIRE(p.dx2 && p.w_fc2t && p.hpre && p.dh && p.w_fc1t && p.x1 && p.ln_g && p.dx1, "mlp_bwd: null argument");
  SRAD_REQUIRE((((uintptr_t)p.dx2 | (uintptr_t)p.hpre | (uintptr_t)p.dh | (uintptr_t)p.x1 | (uintptr_t)p.dx1) & 15) == 0,
               "mlp_bwd: tensors must be 16-byte aligned");
  SRAD_REQUIRE(!p.rs2 || p.rps > 0, "mlp_bwd: rows per sample missing");
  if (p.KA > 0) {
    SRAD_REQUIRE(p.dA && p.w_adjt && (p.ld_dA & 3) == 0 && ((uintptr_t)p.dA & 15) == 0 &&
                     (!p.y_act || ((p.ld_y & 3) == 0 && ((uintptr_t)p.y_act & 15) == 0)) && ((uintptr_t)p.dA_out & 15) == 0,
                 "mlp_bwd: the adjust gradient rows must be float4-addressable");
  }
  const BwdCfg c = bwd_cfg(p.d, p.m, p.KA);
#define X(a, b, cc, dd, e) if (c.gd == a && c.kcd == b && c.gm == cc && c.kcm == dd && c.kca == e) return launch_bwd<a, b, cc, dd, e>(p, q, stream);
  SRAD_BWD_CFGS(X)
#undef X
  return srad_set_error(SRAD_ERR_ARG, "mlp_bwd: no kernel instance for this geometry");
}

bool srad_lin_ln_bwd_supported(int prec, int M, int K, int d) {
  if (!(prec == SRAD_PREC_BF16 && M % 16 == 0 && d % 4 == 0 && K % 4 == 0 && d >= 32 && d <= SRAD_LNB_CP && K >= 32 && K <= 1024)) return false;
  const int gd = (d + FB_SC - 1) / FB_SC, kc = srad_cp(K) / 32;
#define X(a, b) if (gd == a && kc == b) return true;
  SRAD_LIN_CFGS(X)
#undef X
  return false;
}

int srad_launch_lin_ln_bwd(const LinLnBwdParams& p, WgradQueue& q, hipStream_t stream) {
  SRAD_REQUIRE(srad_lin_ln_bwd_supported(SRAD_PREC_BF16, p.M, p.K, p.d), "lin_ln_bwd: unsupported shape M=%d K=%d d=%d", p.M, p.K, p.d);
  SRAD_REQUIRE(p.dY && p.w_t && p.x && p.ln_g && p.out, "lin_ln_bwd: null argument");
  SRAD_REQUIRE(((p.ld_dy | p.ldx | p.ld_out | (p.dres ? p.ld_dres : 0)) & 3) == 0 &&
                   (((uintptr_t)p.dY | (uintptr_t)p.x | (uintptr_t)p.out | (uintptr_t)p.dres) & 15) == 0,
               "lin_ln_bwd: rows must be 16-byte aligned (strides multiples of 4 floats)");
  const int gd = (p.d + FB_SC - 1) / FB_SC, kc = srad_cp(p.K) / 32;
#define X(a, b) if (gd == a && kc == b) return launch_lin<a, b>(p, q, stream);
  SRAD_LIN_CFGS(X)
#undef X
  return srad_set_error(SRAD_ERR_ARG, "lin_ln_bwd: no kernel instance for this geometry");
}
