// kernels_fused.hip - the second half of a DRCT Swin block as ONE launch (bf16 MFMA, fp32 accumulate):
//
//   x1  = shortcut + proj(attn) + b_proj                         (src/drct.py:300, 509)
//   x2  = x1 + fc2(GELU(fc1(LayerNorm2(x1)) + b1)) + b2          (src/drct.py:510, 184-190)
//   out = act(adjust(x2) + b_adj) * alpha (+ R)                  (src/drct.py:389-396: 1x1 conv,
//                                                                  LeakyReLU 0.2 / "*0.2 + x" for adjust5)
//
// Everything after the attention is row-wise, so a workgroup owns 16 (or 64) token rows for the whole chain:
// the rows' activations never leave the CU (A operands in LDS as bf16, x1/x2 in accumulator registers
// in fp32), and the four weight matrices are streamed through one LDS stage buffer, 128 output columns
// x up to 256 k per stage, with later stages' global loads in flight behind the current MFMAs.
// This replaces four launches (proj, fc1, fc2, adjust) and three HBM round trips per block.
//
// Measured on MI355X (tools/fused_bench.py) and designed around:
//   * bias / gamma / beta loads inside the per-group epilogues each cost a memory round trip: they are
//     staged into LDS once at the top.
//   * 16x32 wave tiles made the kernel LDS-read bound (3 KB of fragments per 2 MFMAs); each wave now owns a
//     32x32 tile of the stage (2 A + 2 B fragments per 4 MFMAs).
//   * a runtime column-group index sends the accumulator tiles to scratch, so the stage geometry is a
//     template parameter and the stage sequence is unrolled with static_for.
// Rules carried over from kernels_gemm.hip: unconditional loads on clamped addresses, padding zeroed by
// selects when values are written to LDS (never "x * 0": the clamped reads may be NaN bit patterns).
#include "srad_common.h"
#include <cstdlib>
#include <type_traits>

namespace {

// compile-time loop: f(std::integral_constant<int, I>{}) for I = B .. N-1
template <int B, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (B < N) {
    f(std::integral_constant<int, B>{});
    static_for<B + 1, N>(f);
  }
}

#ifndef SRAD_MLP_NSETS16
#define SRAD_MLP_NSETS16 3
#endif
// split-bf16: 1 = a register set holds the hi AND the lo weights of a stage (two sets; an activation fragment is read from LDS
// once per plane and meets both), 0 = hi and lo stream as half stages through four sets (the hi plane is read twice)
#ifndef SRAD_X3_FULLSTAGE
#define SRAD_X3_FULLSTAGE 1
#endif
constexpr int F_LDA = 400;         // LDS row stride of the <=384-wide bf16 activation tile (2 mod 4 sixteen-byte units: conflict-free for the lane groups of ds_read_b128, see kernels_conv80.hip)
constexpr int F_LDH = 528;         // LDS row stride of the <=512-wide hidden tile
constexpr int F_SC = 128;          // output columns per weight stage
constexpr int F_NV = 2560;         // floats of bias / gamma / beta staged in LDS (10 per thread)

// erf by Abramowitz-Stegun 7.1.26 (|error| <= 1.5e-7): enough below bf16 resolution, ~4x fewer
// instructions than erff
__device__ __forceinline__ float gelu_fast(float x) {
  const float z = fabsf(x) * 0.70710678118654752440f;
  const float t = 1.0f / (1.0f + 0.3275911f * z);
  const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
  const float erfa = 1.0f - poly * __expf(-z * z);
  return 0.5f * x * (1.0f + (x < 0.f ? -erfa : erfa));
}

// GD/GM/GN = 128-column groups of the block dim / hidden / adjust output, KGD/KGM = 256-wide k groups
// of the block dim / hidden.
// FM = token rows per workgroup (16, or 64 for large token counts: every workgroup streams all four weight
// matrices, so at 65536 tokens 16-row tiles move 4096 x 0.5 MB per launch through L2).
// KCD / KCM = 32-wide k chunks of the block dim / hidden (exact: a stage loads and multiplies only its real chunks; a
// run-time chunk count meant duplicate loads of chunk 0 or branches around the loads, both measured slower).
// SPLIT = the split-bf16 mode (SRAD_PREC_BF16X3, inference): both activation tiles exist as a hi and a lo bf16 plane, every
// weight stage streams twice (hi pack, then lo pack: "half stages" through the same register sets) and a product is
// W_hi.A_hi + W_hi.A_lo + W_lo.A_hi; the attention output comes in as fp32.  Nothing else changes - same stages, same epilogues.
template <int FM, int GD, int KGD, int GM, int KGM, int GN, int KCD, int KCM, bool STAMP = false, bool SPLIT = false>
__global__ __launch_bounds__(512) void mlp_block_kernel(const MlpBlockParams p) {
  static_assert(KGD == (KCD + 7) / 8 && KGM == (KCM + 7) / 8, "k groups are 8 chunks wide");
  static_assert(!SPLIT || (FM <= 32 && !STAMP), "split-bf16: 16 / 32-row tiles (two planes of each tile in LDS)");
  constexpr int NRT = FM / 16;       // 16-row MFMA tiles per workgroup
  extern __shared__ __attribute__((aligned(16))) char smem[];
  __bf16* A1 = reinterpret_cast<__bf16*>(smem);                  // [FM][F_LDA] attn tile -> LN2(x1) -> x2
  __bf16* Hs = A1 + FM * F_LDA;                                  // [FM][F_LDH] GELU(fc1)
  __bf16* A1L = Hs + FM * F_LDH;                                 // split-bf16: the lo planes of the two tiles
  __bf16* HsL = A1L + FM * F_LDA;
  float* vec = reinterpret_cast<float*>(SPLIT ? HsL + FM * F_LDH : Hs + FM * F_LDH);   // b_proj | b_fc1 | b_fc2 | b_adj | gamma | beta
  float* red = vec + F_NV;                                       // [FM][8 waves][2] LayerNorm partial sums

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 15, fq = lane >> 4;
  // XCD affinity (see qkv_attn_kernel): XCD k takes the row tiles of strip k of the token range
  int tile_i = blockIdx.x;
  if ((gridDim.x & 7) == 0 && !p.no_xcd_map) tile_i = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
  const int m0 = tile_i * FM;
  const int d = p.d, m = p.m, no = p.no;
  const int dbg = STAMP ? p.dbg : 0;                               // the switch-off experiments exist in the diagnostic build only
  // diagnostic build only (tools/stamp_bench.py): shader-clock stamps of every wave at the phase boundaries
  auto stamp = [&](int idx) {
    if constexpr (STAMP) {
      unsigned long long t;
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
      __builtin_amdgcn_sched_barrier(0);
      if (lane == 0) p.stamps[((size_t)blockIdx.x * 8 + wave) * 16 + idx] = t;
    }
  };
  stamp(0);
  constexpr int Kd = KCD * 32, Km = KCM * 32;                    // packed K of the weights
  constexpr int n_proj = GD * KGD, n_fc1 = GM * KGD, n_fc2 = GD * KGM, n_adj = GN * KGD;
  constexpr int n_stages = n_proj + n_fc1 + n_fc2 + n_adj;
  constexpr bool FULLST = SPLIT && SRAD_X3_FULLSTAGE;
  constexpr int NPART = (SPLIT && !FULLST) ? 2 : 1;              // weight streams per stage (hi | hi, lo)
  constexpr int RPS = FULLST ? 16 : 8;                           // registers (32-wide k chunks) per set
  constexpr int n_vst = n_stages * NPART;                        // (stage, part) pairs in stream order
  // offsets of the staged vectors
  float* const v_bp = vec; float* const v_b1 = vec + 384; float* const v_b2 = vec + 896; float* const v_ba = vec + 1280;
  float* const v_g = vec + 1664; float* const v_b = vec + 2048;

  // A wave whose 16 output columns all lie beyond the matrix (column groups are 128 wide: d = 276 fills 17.25 of its
  // 24 sixteen-column tiles, the 32-channel adjust conv 2 of 8) would stream zero rows of the padded pack and multiply
  // them: it skips both (wave-uniform scalar branch; its accumulators stay zero, which is what the epilogues expect).
  const int wave_s = __builtin_amdgcn_readfirstlane(wave);
  // stage s: phase (0 proj, 1 fc1, 2 fc2, 3 adjust), 128-column group g, 256-wide k group kg - all compile-time
  struct StageGeo { int ph, g, kg, nch, kc; };
  auto geo = [](int s) constexpr -> StageGeo {
    if (s > n_stages - 1) s = n_stages - 1;
    int ph = 0, kgs = 1, kc = 1;
    if (s < n_proj) { ph = 0; kgs = KGD; kc = KCD; }
    else if ((s -= n_proj) < n_fc1) { ph = 1; kgs = KGD; kc = KCD; }
    else if ((s -= n_fc1) < n_fc2) { ph = 2; kgs = KGM; kc = KCM; }
    else { s -= n_fc2; ph = 3; kgs = KGD; kc = KCD; }
    const int g = s / kgs, kg = s - g * kgs;
    const int nch = kc - kg * 8 < 8 ? kc - kg * 8 : 8;
    return StageGeo{ph, g, kg, nch, kc};
  };
  // 8 waves (two per SIMD), each owns 16 output columns of a 128-column stage, i.e. 16 rows of the packed weight that
  // NOBODY else reads: the weights go straight from global memory into MFMA fragment registers, three stages ahead (no
  // LDS stage, no barrier per stage), from the fragment-major pack (one contiguous kilobyte per wave load).
  constexpr int NSETS = FULLST ? 2 : (SPLIT ? 4 : (FM == 16 ? SRAD_MLP_NSETS16 : 3));   // 16-row tiles leave ~100 VGPRs free at 2 waves per SIMD: deeper weight stream
  u32x4 w_reg[NSETS][RPS];
  auto load_w = [&](auto S, u32x4 (&reg)[RPS]) __attribute__((always_inline)) {
    constexpr int vs = decltype(S)::value < n_vst - 1 ? decltype(S)::value : n_vst - 1;
    constexpr StageGeo sg = geo(vs / NPART);
    constexpr bool lo = (vs % NPART) == 1;
    const char* w = lo ? (const char*)(sg.ph == 0 ? p.w_proj_lo : (sg.ph == 1 ? p.w_fc1_lo : (sg.ph == 2 ? p.w_fc2_lo : p.w_adj_lo)))
                       : (const char*)(sg.ph == 0 ? p.w_proj : (sg.ph == 1 ? p.w_fc1 : (sg.ph == 2 ? p.w_fc2 : p.w_adj)));
    const int nreal = sg.ph == 0 ? d : (sg.ph == 1 ? m : (sg.ph == 2 ? d : no));
    // fragment-major pack: 1 KB tiles (16 rows x 32 k), tile (n / 16, k / 32); this wave's row tile is g * 8 + wave;
    // lane: row fr, k 8 fq .. of the tile - the 64 lanes cover its 1 KB
    // A wave whose 16 columns are all padding loads nothing useful - but a load inside a branch makes hipcc's wait-count
    // pass assume the branch was NOT taken at the join (vmcnt of every older load drops to ~0: the vectors and the activation
    // tile were waited for together with three whole weight stages).  So the loads are unconditional and such a wave reads one
    // 16-byte word per load instead (all lanes the same address: one request).
    const bool live = (sg.g * 8 + wave_s) * 16 < nreal;
    const char* base = live ? w + ((size_t)(sg.g * 8 + wave) * sg.kc + sg.kg * 8) * 1024 + fr * 64 + fq * 16 : w;
    const int step = live ? 1024 : 0;
#pragma unroll
    for (int cc = 0; cc < sg.nch; ++cc) reg[cc] = *reinterpret_cast<const u32x4*>(base + cc * step);
    if constexpr (FULLST) {                                         // the lo terms of the same fragments: same offsets in the lo pack
      const char* wl = (const char*)(sg.ph == 0 ? p.w_proj_lo : (sg.ph == 1 ? p.w_fc1_lo : (sg.ph == 2 ? p.w_fc2_lo : p.w_adj_lo)));
      const char* bl = wl + (base - w);
#pragma unroll
      for (int cc = 0; cc < sg.nch; ++cc) reg[8 + cc] = *reinterpret_cast<const u32x4*>(bl + cc * step);
    }
  };
  // c[rt] += (A[16 rows][k0 .. k0 + nch*32) . W[16 columns of this wave][..]^T)^T
  // (split-bf16: `reg` holds the hi weights and both planes of A are multiplied - AL = the lo plane -, or the lo weights
  // against the hi plane only: AL = null)
  auto mma_stage = [&](const __bf16* A, const __bf16* AL, int lda, int k0, int nch, const u32x4 (&reg)[RPS], f32x4 (&c)[NRT]) __attribute__((always_inline)) {
    const __bf16* ar = A + fr * lda + k0 + 8 * fq;
#pragma unroll
    for (int cc = 0; cc < 8; ++cc) {
      if (cc < nch) {
        const bf16x8 b0 = __builtin_bit_cast(bf16x8, reg[cc]);
        if constexpr (FULLST) continue;                             // (below, with the fragment reads of several chunks in flight)
        if constexpr (SPLIT) {
          if (AL) {
            const __bf16* al = AL + fr * lda + k0 + 8 * fq;
#pragma unroll
            for (int rt = 0; rt < NRT; ++rt) {
              const bf16x8 a = *reinterpret_cast<const bf16x8*>(al + rt * 16 * lda + cc * 32);
              c[rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b0, a, c[rt], 0, 0, 0);
            }
          }
        }
        if constexpr (SPLIT) {
          // operands swapped (W as the MFMA "A", activations as "B"): the result tile is transposed, so a lane
          // holds 4 CONSECUTIVE OUTPUT COLUMNS (4*fq + e) of one token row (fr) - exactly the k-contiguous
          // quad the next GEMM's A operand, the bias/residual vectors and the global store want.
#pragma unroll
          for (int rt = 0; rt < NRT; ++rt) {
            const bf16x8 a = *reinterpret_cast<const bf16x8*>(ar + rt * 16 * lda + cc * 32);
            c[rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b0, a, c[rt], 0, 0, 0);
          }
        }
      }
    }
    if constexpr (FULLST) {                                         // W_hi.A_lo, W_lo.A_hi, W_hi.A_hi: each plane read once
      constexpr int G = NRT == 1 ? 4 : (NRT == 2 ? 2 : 1);
      const __bf16* al = AL + fr * lda + k0 + 8 * fq;
#pragma unroll
      for (int cc0 = 0; cc0 < 8; cc0 += G) {
        bf16x8 af[G][NRT], afl[G][NRT];
#pragma unroll
        for (int g = 0; g < G; ++g)
          if (cc0 + g < nch) {
#pragma unroll
            for (int rt = 0; rt < NRT; ++rt) {
              af[g][rt] = *reinterpret_cast<const bf16x8*>(ar + rt * 16 * lda + (cc0 + g) * 32);
              afl[g][rt] = *reinterpret_cast<const bf16x8*>(al + rt * 16 * lda + (cc0 + g) * 32);
            }
          }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int g = 0; g < G; ++g)
          if (cc0 + g < nch) {
            const bf16x8 b0 = __builtin_bit_cast(bf16x8, reg[cc0 + g]);
            const bf16x8 bl = __builtin_bit_cast(bf16x8, reg[8 + cc0 + g]);
#pragma unroll
            for (int rt = 0; rt < NRT; ++rt) {
              c[rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b0, afl[g][rt], c[rt], 0, 0, 0);
              c[rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bl, af[g][rt], c[rt], 0, 0, 0);
              c[rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b0, af[g][rt], c[rt], 0, 0, 0);
            }
          }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if constexpr (!SPLIT) {
      // bf16: the activation fragments of G chunks are requested together, THEN multiplied (sched_barrier: left alone the
      // scheduler sinks every LDS read to just above its MFMA - s_waitcnt lgkmcnt(0 / 1) in front of each of a stage's eight
      // dependent MFMAs, i.e. the LDS latency exposed four to eight times per stage and ten to sixteen stages per launch)
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
    }
  };

  // ---- issue the independent loads, in the order their data is needed: the attn tile and the shortcut rows (first GEMM /
  //      its epilogue), the staged vectors, then the weight stages.  vmcnt retires in issue order, so whatever is issued
  //      before a load is waited for with it: with the weights first (as it was) the prologue took three dependent round
  //      trips (stamp build: 6000 of the kernel's 22000 cycles before the first MFMA). ----
  constexpr int NAQ = 32 * GD;                                    // float4 per row of the attn tile actually needed
  constexpr int NAJ = FM * NAQ / 512;                             // float4 per thread
  static_assert(FM * NAQ % 512 == 0, "attn tile must divide over the workgroup");
  u32x2 a_reg[SPLIT ? 1 : NAJ];                                   // four bf16 columns each
  f32x4 a_regf[SPLIT ? NAJ : 1];                                  // split-bf16: the fp32 attention output
#pragma unroll
  for (int j = 0; j < NAJ; ++j) {
    const int idx = tid + 512 * j, row = idx / NAQ, c = (idx - row * NAQ) * 4;
    if constexpr (SPLIT)
      a_regf[j] = *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(p.attn_f) + (size_t)(m0 + row) * p.ld_attn * 4 +
                                                  (unsigned)min(c, d - 4) * 4u);
    else
      a_reg[j] = *reinterpret_cast<const u32x2*>(reinterpret_cast<const char*>(p.attn_h) + (size_t)(m0 + row) * p.ld_attn * 2 +
                                                 (unsigned)min(c, d - 4) * 2u);
  }
  // accumulator layout of this lane: token row 16*rt + fr, columns 128*g + 16*wave + 4*fq + (0..3)
  f32x4 x1[GD][NRT];
#pragma unroll
  for (int g = 0; g < GD; ++g) {
    const int c4 = min(F_SC * g + 16 * wave + 4 * fq, d - 4);
#pragma unroll
    for (int rt = 0; rt < NRT; ++rt)
      x1[g][rt] = (dbg & 32) ? f32x4{0.f, 0.f, 0.f, 0.f}
                               : *reinterpret_cast<const f32x4*>(p.shortcut + (size_t)(m0 + rt * 16 + fr) * p.ld_short + c4);
  }
  // staged vectors: [0,384) b_proj, [384,896) b_fc1, [896,1280) b_fc2, [1280,1664) b_adj, [1664,2048) gamma, [2048,2432) beta.
  // Wave w < 6 stages vector w: a wave-uniform base pointer (scalar select, no per-lane pointer table in memory) and two
  // clamped float4 per lane; entries past a vector's length are never used.
  f32x4 vq[2];
  int v_off = 0, v_len = 0;
  {
    const float* src = p.b_proj; int n = d;
    if (wave_s == 1) { src = p.b_fc1; n = m; v_off = 384; }
    else if (wave_s == 2) { src = p.b_fc2; n = d; v_off = 896; }
    else if (wave_s == 3) { src = p.b_adj; n = no; v_off = 1280; }
    else if (wave_s == 4) { src = p.ln_g; n = d; v_off = 1664; }
    else if (wave_s == 5) { src = p.ln_b; n = d; v_off = 2048; }
    v_len = wave_s == 1 ? 512 : 384;
    vq[0] = *reinterpret_cast<const f32x4*>(src + min(4 * lane, n - 4));
    vq[1] = *reinterpret_cast<const f32x4*>(src + min(256 + 4 * lane, n - 4));
  }
  // DropPath factors of this lane's token rows (training; 1 otherwise)
  float rs1v[NRT], rs2v[NRT];
#pragma unroll
  for (int rt = 0; rt < NRT; ++rt) {
    const int smp = p.rps > 0 ? (m0 + rt * 16 + fr) / p.rps : 0;
    // unconditional loads from a global pointer either way (see load_w; a pointer to a __device__ constant would make
    // these flat loads, which count out of order: hipcc then waits vmcnt(0) for everything)
    typedef const float __attribute__((address_space(1)))* gfloat_p;   // a select of two kernel-argument pointers is a generic pointer to hipcc
    const float r1 = ((gfloat_p)(p.rs1 ? p.rs1 : p.b_proj))[p.rs1 ? smp : 0], r2 = ((gfloat_p)(p.rs2 ? p.rs2 : p.b_proj))[p.rs2 ? smp : 0];
    rs1v[rt] = p.rs1 ? r1 : 1.f;
    rs2v[rt] = p.rs2 ? r2 : 1.f;
  }
  load_w(std::integral_constant<int, 0>{}, w_reg[0]);             // only what the first stage needs rides with the activations:
  stamp(1);                                                        // the CU's load queue is first-in first-out and ~200 KB of
                                                                   // weight requests ahead of them cost 3000 cycles (stamp build)
  if (wave_s < 6) {
    *reinterpret_cast<f32x4*>(vec + v_off + 4 * lane) = vq[0];
    if (256 + 4 * lane < v_len) *reinterpret_cast<f32x4*>(vec + v_off + 256 + 4 * lane) = vq[1];
  }
  stamp(2);                                                        // the vectors have arrived
  {  // attn tile -> bf16 -> A1 (zero beyond d)
#pragma unroll
    for (int j = 0; j < NAJ; ++j) {
      const int idx = tid + 512 * j, row = idx / NAQ, c = (idx - row * NAQ) * 4;
      if constexpr (SPLIT) {
        bf16x4 hh, ll;
        srad_split4(c < d ? a_regf[j] : f32x4{0.f, 0.f, 0.f, 0.f}, hh, ll);
        *reinterpret_cast<bf16x4*>(A1 + row * F_LDA + c) = hh;
        *reinterpret_cast<bf16x4*>(A1L + row * F_LDA + c) = ll;
      } else {
        *reinterpret_cast<u32x2*>(A1 + row * F_LDA + c) = c < d ? a_reg[j] : u32x2{0u, 0u};
      }
    }
  }
  static_for<1, NSETS>([&](auto Q) { load_w(Q, w_reg[decltype(Q)::value]); });   // the rest of the look-ahead, behind the tile

  // ---- phase epilogues (run when the last k-group of a 128-column group is done) ----
  auto col4_of = [&](int g) { return F_SC * g + 16 * wave + 4 * fq; };
  // bf16 quad into the LDS tile and, when `gsave` is set and the quad holds real columns, into the global bf16 copy the
  // weight gradient reads ([M][gld])
  auto store_bf4 = [&](__bf16* base, int ld, int rt, int c4, f32x4 v, __bf16* gsave = nullptr, int gld = 0, bool real = false) __attribute__((always_inline)) {
    bf16x4 h;
    if constexpr (SPLIT) {                                           // hi plane here, lo plane at the same place of the tile's twin
      bf16x4 l;
      srad_split4(v, h, l);
      *reinterpret_cast<bf16x4*>((base == A1 ? A1L : HsL) + (rt * 16 + fr) * ld + c4) = l;
    } else {
      h[0] = (__bf16)v[0]; h[1] = (__bf16)v[1]; h[2] = (__bf16)v[2]; h[3] = (__bf16)v[3];
    }
    *reinterpret_cast<bf16x4*>(base + (rt * 16 + fr) * ld + c4) = h;
    if (gsave && real) *reinterpret_cast<bf16x4*>(gsave + (size_t)(m0 + rt * 16 + fr) * gld + c4) = h;
  };
  auto epi_proj = [&](auto G, const f32x4 (&c)[NRT]) __attribute__((always_inline)) {
    constexpr int g = decltype(G)::value;
    {
      const f32x4 bp = *reinterpret_cast<const f32x4*>(v_bp + min(col4_of(g), 380));
#pragma unroll
      for (int rt = 0; rt < NRT; ++rt) x1[g][rt] += (c[rt] + bp) * rs1v[rt];
    }
    if constexpr (g != GD - 1) return;
    if (p.save_x1) {
#pragma unroll
      for (int gg = 0; gg < GD; ++gg) {
        const int c4 = col4_of(gg);
#pragma unroll
        for (int rt = 0; rt < NRT; ++rt)
          if (c4 < d) *reinterpret_cast<f32x4*>(p.save_x1 + (size_t)(m0 + rt * 16 + fr) * d + c4) = x1[gg][rt];
      }
    }
    // LayerNorm2 over the d real columns of x1 -> bf16 -> A1 (all proj reads of A1 are behind a barrier)
    float sm[NRT], sq[NRT];
#pragma unroll
    for (int rt = 0; rt < NRT; ++rt) { sm[rt] = 0.f; sq[rt] = 0.f; }
#pragma unroll
    for (int gg = 0; gg < GD; ++gg) {
      const bool in = col4_of(gg) < d;                             // d % 4 == 0: a quad is all in or all out
#pragma unroll
      for (int rt = 0; rt < NRT; ++rt) {
        const f32x4 v = in ? x1[gg][rt] : f32x4{0.f, 0.f, 0.f, 0.f};
        sm[rt] += (v[0] + v[1]) + (v[2] + v[3]);
        sq[rt] += (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
      }
    }
#pragma unroll
    for (int rt = 0; rt < NRT; ++rt) {
      sm[rt] += __shfl_xor(sm[rt], 16); sq[rt] += __shfl_xor(sq[rt], 16);
      sm[rt] += __shfl_xor(sm[rt], 32); sq[rt] += __shfl_xor(sq[rt], 32);
      if (fq == 0) {
        red[((rt * 16 + fr) * 8 + wave) * 2 + 0] = sm[rt];
        red[((rt * 16 + fr) * 8 + wave) * 2 + 1] = sq[rt];
      }
    }
    __syncthreads();
    float mu[NRT], rstd[NRT];
#pragma unroll
    for (int rt = 0; rt < NRT; ++rt) {
      const float* r = red + (rt * 16 + fr) * 16;
      float su = 0.f, s2 = 0.f;
#pragma unroll
      for (int w = 0; w < 8; ++w) { su += r[2 * w]; s2 += r[2 * w + 1]; }
      mu[rt] = su / (float)d;
      rstd[rt] = rsqrtf(fmaxf(s2 / (float)d - mu[rt] * mu[rt], 0.f) + 1e-5f);
    }
#pragma unroll
    for (int gg = 0; gg < GD; ++gg) {
      const int c4 = col4_of(gg);
      const bool in = c4 < d;
      const f32x4 gam = *reinterpret_cast<const f32x4*>(v_g + min(c4, 380));
      const f32x4 bet = *reinterpret_cast<const f32x4*>(v_b + min(c4, 380));
#pragma unroll
      for (int rt = 0; rt < NRT; ++rt) {
        const f32x4 v = in ? (x1[gg][rt] - mu[rt]) * rstd[rt] * gam + bet : f32x4{0.f, 0.f, 0.f, 0.f};
        if (p.save_xn2 && in) *reinterpret_cast<f32x4*>(p.save_xn2 + (size_t)(m0 + rt * 16 + fr) * d + c4) = v;
        store_bf4(A1, F_LDA, rt, c4, v, p.save_xn2_h, d, in);
      }
    }
  };
  auto epi_fc1 = [&](int g, const f32x4 (&c)[NRT]) __attribute__((always_inline)) {
    const int c4 = col4_of(g);
    const f32x4 b1 = *reinterpret_cast<const f32x4*>(v_b1 + min(c4, 508));
#pragma unroll
    for (int rt = 0; rt < NRT; ++rt) {
      f32x4 v = c[rt] + b1;
      if (p.save_hpre && c4 < m) *reinterpret_cast<f32x4*>(p.save_hpre + (size_t)(m0 + rt * 16 + fr) * m + c4) = v;
      if (p.save_hpre_h && c4 < m) {
        bf16x4 hq;
        hq[0] = (__bf16)v[0]; hq[1] = (__bf16)v[1]; hq[2] = (__bf16)v[2]; hq[3] = (__bf16)v[3];
        *reinterpret_cast<bf16x4*>(p.save_hpre_h + (size_t)(m0 + rt * 16 + fr) * m + c4) = hq;
      }
      if (!(dbg & 4)) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = gelu_fast(v[e]);
      }
      if (p.save_hact && c4 < m) *reinterpret_cast<f32x4*>(p.save_hact + (size_t)(m0 + rt * 16 + fr) * m + c4) = v;
      store_bf4(Hs, F_LDH, rt, c4, c4 < m ? v : f32x4{0.f, 0.f, 0.f, 0.f}, p.save_hact_h, m, c4 < m);      // m % 4 == 0
    }
  };
  auto epi_fc2 = [&](auto G, const f32x4 (&c)[NRT]) __attribute__((always_inline)) {
    constexpr int g = decltype(G)::value;
    {
      const f32x4 b2 = *reinterpret_cast<const f32x4*>(v_b2 + min(col4_of(g), 380));
#pragma unroll
      for (int rt = 0; rt < NRT; ++rt) x1[g][rt] += (c[rt] + b2) * rs2v[rt];
    }
    if constexpr (g != GD - 1) return;
    if (p.save_x2) {
#pragma unroll
      for (int gg = 0; gg < GD; ++gg) {
        const int c4 = col4_of(gg);
#pragma unroll
        for (int rt = 0; rt < NRT; ++rt)
          if (c4 < d) *reinterpret_cast<f32x4*>(p.save_x2 + (size_t)(m0 + rt * 16 + fr) * d + c4) = x1[gg][rt];
      }
    }
    // x2 -> bf16 -> A1 (its last readers, the fc1 stages, finished long ago)
#pragma unroll
    for (int gg = 0; gg < GD; ++gg) {
      const int c4 = col4_of(gg);
#pragma unroll
      for (int rt = 0; rt < NRT; ++rt) store_bf4(A1, F_LDA, rt, c4, c4 < d ? x1[gg][rt] : f32x4{0.f, 0.f, 0.f, 0.f}, p.save_x2_h, d, c4 < d);
    }
  };
  auto epi_adj = [&](int g, const f32x4 (&c)[NRT]) __attribute__((always_inline)) {
    const int c4 = col4_of(g);
    const int cc = min(c4, no - 4);                                 // no % 4 == 0
    const f32x4 ba = *reinterpret_cast<const f32x4*>(v_ba + min(c4, 380));
    f32x4 rv[NRT];
#pragma unroll
    for (int rt = 0; rt < NRT; ++rt) rv[rt] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (p.R) {
#pragma unroll
      for (int rt = 0; rt < NRT; ++rt) rv[rt] = *reinterpret_cast<const f32x4*>(p.R + (size_t)(m0 + rt * 16 + fr) * p.ldr + cc);
    }
#pragma unroll
    for (int rt = 0; rt < NRT; ++rt) {
      f32x4 v = c[rt] + ba;
      if (p.act == SRAD_ACT_LRELU) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : v[e] * p.slope;
      }
      v = v * p.alpha + rv[rt];
      if (c4 < no) *reinterpret_cast<f32x4*>(p.Y + (size_t)(m0 + rt * 16 + fr) * p.ldy + p.yoff + c4) = v;
    }
  };

  // ---- all weight stages, unrolled at compile time over the register sets ----
  f32x4 c[NRT];
  static_for<0, n_vst>([&](auto S) {
    constexpr int vs = decltype(S)::value;
    constexpr int s = vs / NPART, part = vs % NPART;               // split-bf16: part 0 = hi weights, 1 = lo weights
    constexpr int ph = s < n_proj ? 0 : (s < n_proj + n_fc1 ? 1 : (s < n_proj + n_fc1 + n_fc2 ? 2 : 3));
    constexpr int ls = s - (ph == 0 ? 0 : (ph == 1 ? n_proj : (ph == 2 ? n_proj + n_fc1 : n_proj + n_fc1 + n_fc2)));
    constexpr int kgs = ph == 2 ? KGM : KGD;
    constexpr int g = ls / kgs, kg = ls - g * kgs;
    u32x4 (&reg)[RPS] = w_reg[vs % NSETS];
    constexpr int Kp = ph == 2 ? Km : Kd;
    constexpr int nch = (Kp >> 5) - kg * 8 < 8 ? (Kp >> 5) - kg * 8 : 8;
    if constexpr (ls == 0 && part == 0) {
      stamp(3 + 3 * ph);                               // this wave's previous phase (its epilogue included) is done
      __syncthreads();                                 // first stage of a phase: the activation tile (A1 / Hs) and,
                                                       // at s == 0, the staged vectors written before are visible
      stamp(4 + 3 * ph);
    }
    if constexpr (kg == 0 && part == 0) {
#pragma unroll
      for (int rt = 0; rt < NRT; ++rt) c[rt] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const bool live = (g * 8 + wave_s) * 16 < (ph == 0 ? d : (ph == 1 ? m : (ph == 2 ? d : no)));
    if (!(dbg & 2) && live)
      mma_stage(ph == 2 ? Hs : A1, (SPLIT && part == 0) ? (ph == 2 ? HsL : A1L) : nullptr, ph == 2 ? F_LDH : F_LDA, kg * 256, nch, reg, c);   // (part == 0 always when a set holds both packs)
    // refill this set, NSETS stages ahead.  Issuing a load can block (the queue is full while the weights stream), so where
    // an epilogue follows that other waves wait for (LayerNorm2, the tile hand-overs) the refill goes behind it
    constexpr bool epi_first = kg == kgs - 1 && part == NPART - 1;
    if constexpr (!epi_first) { if (!(dbg & 1)) load_w(std::integral_constant<int, vs + NSETS>{}, reg); }
    if constexpr (epi_first && g == (ph == 0 ? GD : (ph == 1 ? GM : (ph == 2 ? GD : GN))) - 1) stamp(5 + 3 * ph);   // last MFMAs of the phase issued
    if (!(dbg & 16)) if constexpr (epi_first) {
      if constexpr (ph == 0) epi_proj(std::integral_constant<int, g>{}, c);
      else if constexpr (ph == 1) epi_fc1(g, c);
      else if constexpr (ph == 2) epi_fc2(std::integral_constant<int, g>{}, c);
      else epi_adj(g, c);
    }
    if constexpr (epi_first) { if (!(dbg & 1)) load_w(std::integral_constant<int, vs + NSETS>{}, reg); }
  });
  stamp(15);
}

struct FusedCfg { int gd, kgd, gm, kgm, gn, kcd, kcm; };
inline FusedCfg fused_cfg(int d, int m, int no) {
  const int Kd = srad_cp(d), Km = srad_cp(m);
  return FusedCfg{(d + F_SC - 1) / F_SC, (Kd + 255) / 256, (m + F_SC - 1) / F_SC, (Km + 255) / 256, (no + F_SC - 1) / F_SC, Kd / 32, Km / 32};
}
template <int FM, int GD, int KGD, int GM, int KGM, int GN, int KCD, int KCM, bool STAMP = false, bool SPLIT = false>
int launch_mlp_fm(const MlpBlockParams& p, hipStream_t stream) {
  constexpr size_t lds = (size_t)(FM * F_LDA + FM * F_LDH) * 2 * (SPLIT ? 2 : 1) + (F_NV + FM * 16) * sizeof(float);
  auto kern = mlp_block_kernel<FM, GD, KGD, GM, KGM, GN, KCD, KCM, STAMP, SPLIT>;
  static SradOncePerDevice configured;
  if (configured.need()) {
    SRAD_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    configured.done();
  }
  const double flops = 2.0 * p.M * ((double)p.d * p.d + 2.0 * p.d * p.m + (double)p.no * p.d);
  const double bytes = 4.0 * p.M * (2.0 * p.d + p.no) + 2.0 * ((double)p.d * p.d + 2.0 * p.d * p.m + (double)p.no * p.d);
  SradProfScope prof(stream, SRAD_K_MLP_BLOCK, flops, bytes);
  hipLaunchKernelGGL(kern, dim3(p.M / FM), dim3(512), lds, stream, p);
  SRAD_CHECK_HIP(hipGetLastError());
  return SRAD_OK;
}
template <int GD, int KGD, int GM, int KGM, int GN, int KCD, int KCM>
int launch_mlp(const MlpBlockParams& p, hipStream_t stream) {
  if (p.split) {                                                  // split-bf16: 32-row tiles once every CU still gets a workgroup
    if ((p.fm == 32 || (p.fm == 0 && p.M >= 8192)) && p.M % 32 == 0) return launch_mlp_fm<32, GD, KGD, GM, KGM, GN, KCD, KCM, false, true>(p, stream);
    return launch_mlp_fm<16, GD, KGD, GM, KGM, GN, KCD, KCM, false, true>(p, stream);
  }
  // 16-row tiles until there are enough tokens to fill the chip several times with 64-row tiles
  if (p.fm == 64 && p.M % 64 == 0) return launch_mlp_fm<64, GD, KGD, GM, KGM, GN, KCD, KCM>(p, stream);
  if (p.fm == 32 && p.M % 32 == 0) return launch_mlp_fm<32, GD, KGD, GM, KGM, GN, KCD, KCM>(p, stream);
  if (p.stamps && p.M % 16 == 0) return launch_mlp_fm<16, GD, KGD, GM, KGM, GN, KCD, KCM, true>(p, stream);   // diagnostic build
  if (p.fm == 16) return launch_mlp_fm<16, GD, KGD, GM, KGM, GN, KCD, KCM>(p, stream);
  if (p.M >= 32768 && p.M % 64 == 0) return launch_mlp_fm<64, GD, KGD, GM, KGM, GN, KCD, KCM>(p, stream);
  if (p.M >= 8192 && p.M % 32 == 0) return launch_mlp_fm<32, GD, KGD, GM, KGM, GN, KCD, KCM>(p, stream);
  return launch_mlp_fm<16, GD, KGD, GM, KGM, GN, KCD, KCM>(p, stream);
}
// stage geometries of DRCT-L's five Swin blocks (embed 180 + k*32; mlp ratio 2,2,2,1,1; adjust to 32 / 180 channels):
// (column groups of d, k groups of d, column groups of m, k groups of m, column groups of the adjust output, k chunks of d, of m)
#define SRAD_FUSED_CFGS(X) X(2, 1, 3, 2, 1, 6, 12) X(2, 1, 4, 2, 1, 7, 14) X(2, 1, 4, 2, 1, 8, 16) X(3, 2, 3, 2, 1, 9, 9) X(3, 2, 3, 2, 2, 10, 10)
}  // namespace

bool srad_mlp_block_supported(int prec, int M, int d, int m, int no) {
  if (!((prec == SRAD_PREC_BF16 || prec == SRAD_PREC_BF16X3) && M % 16 == 0 && d % 4 == 0 && m % 4 == 0 && no % 4 == 0 && d >= 32 && d <= 384 && m >= 32 &&
        m <= 512 && no >= 4 && no <= 384))
    return false;
  const FusedCfg c = fused_cfg(d, m, no);
#define X(a, b, cc, dd, e, f, g) if (c.gd == a && c.kgd == b && c.gm == cc && c.kgm == dd && c.gn == e && c.kcd == f && c.kcm == g) return true;
  SRAD_FUSED_CFGS(X)
#undef X
  return false;
}

int srad_launch_mlp_block(const MlpBlockParams& p, hipStream_t stream) {
  SRAD_REQUIRE(srad_mlp_block_supported(SRAD_PREC_BF16, p.M, p.d, p.m, p.no), "mlp_block: unsupported shape M=%d d=%d m=%d no=%d", p.M, p.d, p.m, p.no);
  if (p.split) {
    SRAD_REQUIRE(p.attn_f && (p.ld_attn & 3) == 0 && ((uintptr_t)p.attn_f & 15) == 0, "mlp_block (split-bf16): the fp32 attn rows must be float4-addressable");
    SRAD_REQUIRE(p.w_proj_lo && p.w_fc1_lo && p.w_fc2_lo && p.w_adj_lo, "mlp_block (split-bf16): the lo weight packs are missing");
    SRAD_REQUIRE(p.fm == 0 || p.fm == 16 || p.fm == 32, "mlp_block (split-bf16): 16 or 32 rows per workgroup");
    SRAD_REQUIRE(!p.stamps && !p.rs1 && !p.rs2 && !p.save_x1 && !p.save_xn2 && !p.save_hpre && !p.save_hact && !p.save_x2 && !p.save_hpre_h &&
                     !p.save_xn2_h && !p.save_hact_h && !p.save_x2_h, "mlp_block (split-bf16): inference only");
  } else
  SRAD_REQUIRE((p.ld_attn & 3) == 0 && ((uintptr_t)p.attn_h & 7) == 0, "mlp_block: the bf16 attn rows must be 8-byte addressable");
  SRAD_REQUIRE((p.ld_short & 3) == 0 && ((uintptr_t)p.shortcut & 15) == 0 && (p.ldy & 3) == 0 && (p.yoff & 3) == 0 &&
                   ((uintptr_t)p.Y & 15) == 0 && (!p.R || ((p.ldr & 3) == 0 && ((uintptr_t)p.R & 15) == 0)),
               "mlp_block: shortcut / output / residual rows must be float4-addressable");
  SRAD_REQUIRE((((uintptr_t)p.b_proj | (uintptr_t)p.b_fc1 | (uintptr_t)p.b_fc2 | (uintptr_t)p.b_adj | (uintptr_t)p.ln_g | (uintptr_t)p.ln_b) & 15) == 0,
               "mlp_block: bias / LayerNorm vectors must be 16-byte aligned");
  const FusedCfg c = fused_cfg(p.d, p.m, p.no);
#define X(a, b, cc, dd, e, f, g) if (c.gd == a && c.kgd == b && c.gm == cc && c.kgm == dd && c.gn == e && c.kcd == f && c.kcm == g) return launch_mlp<a, b, cc, dd, e, f, g>(p, stream);
  SRAD_FUSED_CFGS(X)
#undef X
  return srad_set_error(SRAD_ERR_ARG, "mlp_block: no kernel instance for this geometry");
}
