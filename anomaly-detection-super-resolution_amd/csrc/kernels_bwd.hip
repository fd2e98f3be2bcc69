// kernels_bwd.hip - backward kernels of the training step on gfx950 (reference src/trainer.py:152-222:
// loss.backward() through src/drct.py / src/drn.py; the arithmetic is PyTorch autograd's for Linear,
// conv2d, LayerNorm, softmax attention, GELU / LeakyReLU and PixelShuffle).
//
//   wgrad_kernel            dW += dY^T A(X)      MFMA straight from global memory, no LDS staging
//   window_attn_bwd_kernel  dq, dk, dv, dtable   recomputes P per (window, head) from the saved q|k|v
//   ln_bwd_kernel           dx, dgamma, dbeta    one wave per row, statistics recomputed
//   small elementwise kernels (activation backward, pixel un-shuffle, column copies, L1 seed, Adam)
//
// Data gradients (dX = dY W) are not here: they are the forward GEMM of kernels_gemm.hip run on the
// transposed packed weight (srad_launch_pack_weight_transposed).
#include "srad_common.h"
#include <type_traits>
#include <algorithm>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

namespace {

static inline int grid_for(size_t total) {
  size_t b = (total + 255) / 256;
  return (int)(b > 4096 ? 4096 : (b < 1 ? 1 : b));
}

// ------------------------------------------------------------------------------------------
// Weight gradient.  The contraction runs over the rows m (tokens / pixels), which are the SLOW axis of
// both operands in memory ([m][n] and [m][c], channel contiguous).  An MFMA lane supplies A[i][k] and
// B[k][j] for ONE i / j and a few k, and nothing fixes which n a tile row i stands for - so lane
// (fq, fr) loads the float4 dY[m][n0 + 4 fr .. +3] and uses component e as row fr of n-subtile e
// (n = n0 + 4 fr + e), likewise X for the c-subtiles.  Sixteen lanes then read 256 contiguous bytes of
// one row: fully coalesced fragment loads, 2 loads feed 16 MFMAs, and no LDS transpose is needed.
//   fp32: v_mfma_f32_16x16x4_f32, k = 4 rows per step (row m0 + fq)
//   bf16: v_mfma_f32_16x16x32_bf16, k = 32 rows per step (rows m0 + 8 fq + t, t = 0..7)
// One workgroup = 4 waves on one 64 x 64 (n, c) tile of one tap, each wave on its own rows of the
// workgroup's row range (split-K); waves are summed through LDS.
// Split-K partials go to a workspace with plain stores and are summed (fixed order: bit-reproducible
// gradients) by wgrad_reduce_kernel, ONE launch for a batch of up to 8 layers.  Two things that were tried
// and measured on MI355X before this: float atomicAdd into dW (device-scope atomics on a few hundred
// addresses serialise: 60-170 us per layer) and a last-arriver reduction inside the kernel (the agent-scope
// release/acquire fences it needs write back / invalidate a whole XCD L2 because the eight L2s are not
// coherent with each other: 60-90 us per layer).  A kernel boundary is the cheap cross-XCD barrier.
// ------------------------------------------------------------------------------------------
#ifndef SRAD_WGRAD_PREFETCH
#define SRAD_WGRAD_PREFETCH 1   /* two register sets: measured best together with the two-stream backward */
#endif
constexpr int WG_TS = 64 * 64 + 64;          // floats per partial: the tile and its 64 bias sums
constexpr int LNB_RPW = 16;                  // LayerNorm backward: rows per workgroup
constexpr int LNB_CP = SRAD_LNB_CP;                  // columns of a dgamma | dbeta partial row (channel counts <= 320)

// FULL: Linear layers whose row splits are whole 128-row steps and whose DropPath factor is constant over a 32-row
// wave step - no row / column / padding masks at all (columns past N / Cin read clamped real data into tile rows the
// final store drops), the factor is one value per wave step: fewer registers (two waves per SIMD) and ~200 fewer
// VALU instructions per step.
// XH / YH (FULL only): X / dY are stored as bf16 - half the operand bytes and prefetch registers; the 8 rows x 4 columns a
// lane holds are transposed into the four 8-row MFMA operands with v_perm_b32 instead of being converted.  A bf16 dY is
// already multiplied by its DropPath factor (its producer did that), so no per-step factor either.
// LDS2: the waves' tiles are summed pairwise through TWO tile slots instead of four (waves 2, 3 store, waves 0, 1 add and
// store, then one sum of two): 36 KB instead of 70 KB, two barriers instead of one - for the all-bf16 kernel, whose
// registers allow three workgroups per CU.  PF2: two register sets of operand rows (the next step in flight during this one's
// MFMAs); the all-bf16 kernel at three waves per SIMD runs with one.
template <int PREC, bool CONV, bool FULL = false, bool XH = false, bool YH = false, bool LDS2 = false, bool PF2 = (SRAD_WGRAD_PREFETCH != 0)>
__device__ __forceinline__ void wgrad_body(const WgradParams& p, const int ksplit, const int tn, const int tc,
                                           float* __restrict__ part, const int L) {
  static_assert(!(XH || YH) || (!CONV && PREC == SRAD_PREC_BF16), "bf16 operand storage: Linear layers, bf16 MFMA path");
  extern __shared__ __attribute__((aligned(16))) float wsm[];     // [4 waves][64][68] + [4][64] bias + flag
  constexpr int TST = 68;
  float* const dbs = wsm + (LDS2 ? 2 : 4) * 64 * TST;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 15, fq = lane >> 4;
  // XCD-aware order: workgroups are dealt round-robin over the 8 XCDs (linear id L runs on XCD L % 8).  With the row
  // split fastest (ks = L % ksplit, ksplit a multiple of 8) every tile of one row range lands on the same XCD, so the
  // range's dY / X rows are fetched into ONE L2 and all the tiles' re-reads of them hit there.
  const int ks = L % ksplit, tile_id = L / ksplit;
  const int bx = tile_id % tn, by = (tile_id / tn) % tc, tap = tile_id / (tn * tc);
  const int n0 = bx * 64, c0 = by * 64;

  constexpr int KR = PREC == SRAD_PREC_BF16 ? 32 : 4;        // rows per wave step
  constexpr int RL = PREC == SRAD_PREC_BF16 ? 8 : 1;         // rows per lane per step
  const int rows_per = ((p.M + ksplit - 1) / ksplit + 4 * KR - 1) / (4 * KR) * (4 * KR);
  const int mb = ks * rows_per;
  const int me = min(p.M, mb + rows_per);

  const int ncol = n0 + 4 * fr, ccol = c0 + 4 * fr;
  const bool n_ok = ncol < p.N, c_ok = ccol < p.Cin;
  const unsigned noff = (unsigned)(min(ncol, p.N - 4) + p.ycol0);
  const unsigned coff = (unsigned)min(ccol, p.Cin - 4);
  [[maybe_unused]] const int pad = p.ntaps == 9 ? 1 : 0;
  [[maybe_unused]] const int ky = p.ntaps == 9 ? tap / 3 : 0, kx = p.ntaps == 9 ? tap - (tap / 3) * 3 : 0;
  [[maybe_unused]] const int hwo = p.Ho * p.Wo;

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  f32x4 bsum = f32x4{0.f, 0.f, 0.f, 0.f};
  const f32x4 zero4 = f32x4{0.f, 0.f, 0.f, 0.f};

  // every load is unconditional on a clamped address; masking happens on the registers afterwards.
  // The next step's loads are issued before this step's MFMAs (two register sets).
  constexpr int NSET = PF2 ? 2 : 1;
  f32x4 av[NSET][YH ? 1 : RL], bv[NSET][XH ? 1 : RL];
  u32x2 avh[NSET][YH ? RL : 1], bvh[NSET][XH ? RL : 1];   // bf16 storage: 4 values = 8 bytes per row
  unsigned okm[2];                    // bit t: a row valid, bit 8 + t: b row valid
  float rs[NSET][FULL ? 1 : RL];
  auto load_step = [&](int m0, auto set_c) {
    constexpr int set = decltype(set_c)::value;
    if constexpr (FULL) {
#pragma unroll
      for (int t = 0; t < RL; ++t) {
        const size_t mr = (size_t)(m0 + RL * fq + t);
        if constexpr (YH) avh[set][t] = *reinterpret_cast<const u32x2*>(reinterpret_cast<const __bf16*>(p.dY) + mr * p.ldy + noff);
        else av[set][t] = *reinterpret_cast<const f32x4*>(p.dY + mr * p.ldy + noff);
        if constexpr (XH) bvh[set][t] = *reinterpret_cast<const u32x2*>(reinterpret_cast<const __bf16*>(p.X) + mr * p.ldx + coff);
        else bv[set][t] = *reinterpret_cast<const f32x4*>(p.X + mr * p.ldx + coff);
      }
      if constexpr (!YH) rs[set][0] = p.row_scale ? p.row_scale[m0 / p.rps] : 1.f;
      return;
    }
    unsigned ok = 0u;
    [[maybe_unused]] int bb = 0, oy = 0, ox = 0;
    if constexpr (CONV) {                 // pixel of the lane's first row; the following rows step from it
      const int mf = min(m0 + RL * fq, p.M - 1);
      bb = mf / hwo;
      const int rem = mf - bb * hwo;
      oy = rem / p.Wo;
      ox = rem - oy * p.Wo;
    }
#pragma unroll
    for (int t = 0; t < RL; ++t) {
      const int m = m0 + RL * fq + t;
      const int mc = min(m, p.M - 1);
      if constexpr (YH) avh[set][t] = *reinterpret_cast<const u32x2*>(reinterpret_cast<const __bf16*>(p.dY) + (size_t)mc * p.ldy + noff);
      else av[set][t] = *reinterpret_cast<const f32x4*>(p.dY + (size_t)mc * p.ldy + noff);
      size_t xr = (size_t)mc;
      bool in = true;
      if constexpr (CONV) {
        if (t > 0 && m < p.M) {           // next pixel in raster order
          ++ox;
          if (ox == p.Wo) { ox = 0; ++oy; if (oy == p.Ho) { oy = 0; ++bb; } }
        }
        const int iy = oy * p.stride - pad + ky, ix = ox * p.stride - pad + kx;
        in = iy >= 0 && iy < p.Hi && ix >= 0 && ix < p.Wi;
        xr = (size_t)((bb * p.Hi + min(max(iy, 0), p.Hi - 1)) * p.Wi + min(max(ix, 0), p.Wi - 1));
      }
      if constexpr (XH) bvh[set][t] = *reinterpret_cast<const u32x2*>(reinterpret_cast<const __bf16*>(p.X) + xr * p.ldx + coff);
      else bv[set][t] = *reinterpret_cast<const f32x4*>(p.X + xr * p.ldx + coff);
      rs[set][t] = (!YH && p.row_scale) ? p.row_scale[mc / p.rps] : 1.f;
      ok |= ((m < me && n_ok) ? 1u : 0u) << t;
      ok |= ((m < me && c_ok && in) ? 1u : 0u) << (8 + t);
    }
    okm[set] = ok;
  };
  auto compute_step = [&](auto set_c) {
    constexpr int set = decltype(set_c)::value;
    f32x4 a[RL], b[RL];
    // 8 rows x (2 dwords = 4 bf16 columns) -> operand e = column e of the 8 rows: dword t/2 of operand 2 w + half takes the
    // low (half 0) or high (half 1) 16 bits of dword w of rows t and t + 1
    auto transpose_h = [&](const u32x2 (&src)[RL], bf16x8 (&dst)[4]) {
#pragma unroll
      for (int w = 0; w < 2; ++w) {
        u32x4 lo, hi;
#pragma unroll
        for (int t2 = 0; t2 < 4; ++t2) {
          lo[t2] = __builtin_amdgcn_perm(src[2 * t2 + 1][w], src[2 * t2][w], 0x05040100u);
          hi[t2] = __builtin_amdgcn_perm(src[2 * t2 + 1][w], src[2 * t2][w], 0x07060302u);
        }
        dst[2 * w] = __builtin_bit_cast(bf16x8, lo);
        dst[2 * w + 1] = __builtin_bit_cast(bf16x8, hi);
      }
    };
#pragma unroll
    for (int t = 0; t < RL; ++t) {
      if constexpr (FULL) {
        if constexpr (YH) {                                 // fp32 view of the row for the bias sums only
#pragma unroll
          for (int w = 0; w < 2; ++w) {
            a[t][2 * w] = __builtin_bit_cast(float, avh[set][t][w] << 16);
            a[t][2 * w + 1] = __builtin_bit_cast(float, avh[set][t][w] & 0xffff0000u);
          }
        } else {
          a[t] = av[set][t] * rs[set][0];
        }
        if constexpr (!XH) b[t] = bv[set][t];
      } else {
        if constexpr (YH) {
          if (!((okm[set] >> t) & 1u)) avh[set][t] = u32x2{0u, 0u};
#pragma unroll
          for (int w = 0; w < 2; ++w) {
            a[t][2 * w] = __builtin_bit_cast(float, avh[set][t][w] << 16);
            a[t][2 * w + 1] = __builtin_bit_cast(float, avh[set][t][w] & 0xffff0000u);
          }
        } else {
          a[t] = ((okm[set] >> t) & 1u) ? av[set][t] * rs[set][t] : zero4;
        }
        if constexpr (XH) { if (!((okm[set] >> (8 + t)) & 1u)) bvh[set][t] = u32x2{0u, 0u}; }
        else b[t] = ((okm[set] >> (8 + t)) & 1u) ? bv[set][t] : zero4;
      }
      bsum += a[t];
    }
    if constexpr (PREC == SRAD_PREC_BF16) {
      bf16x8 ah[4], bh[4];
      if constexpr (YH) {
        transpose_h(avh[set], ah);
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int t = 0; t < 8; ++t) ah[e][t] = (__bf16)a[t][e];
      }
      if constexpr (XH) {
        transpose_h(bvh[set], bh);
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int t = 0; t < 8; ++t) bh[e][t] = (__bf16)b[t][e];
      }
#pragma unroll
      for (int en = 0; en < 4; ++en)
#pragma unroll
        for (int ec = 0; ec < 4; ++ec)
          acc[en][ec] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[en], bh[ec], acc[en][ec], 0, 0, 0);
    } else {
#pragma unroll
      for (int en = 0; en < 4; ++en)
#pragma unroll
        for (int ec = 0; ec < 4; ++ec)
          acc[en][ec] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[0][en], b[0][ec], acc[en][ec], 0, 0, 0);
    }
  };

  {
    using S0 = std::integral_constant<int, 0>;
    using S1 = std::integral_constant<int, 1>;
    constexpr int ST = 4 * KR;
    int m0 = mb + wave * KR;
    if constexpr (PF2) {
      if (m0 < me) load_step(m0, S0{});
      while (m0 < me) {
        if (m0 + ST < me) load_step(m0 + ST, S1{});
        compute_step(S0{});
        m0 += ST;
        if (m0 >= me) break;
        if (m0 + ST < me) load_step(m0 + ST, S0{});
        compute_step(S1{});
        m0 += ST;
      }
    } else {
      (void)sizeof(S1);
      for (; m0 < me; m0 += ST) { load_step(m0, S0{}); compute_step(S0{}); }
    }
  }

  // ---- the four waves' partial tiles through LDS (plain 16-byte stores): lane (fq, fr) element e of
  //      acc[en][ec] is (n = 16 fq + 4 e + en, c = 4 fr + ec) ----
  {
    float* const mine = wsm + (LDS2 ? (wave & 1) : wave) * 64 * TST;
    auto put_tile = [&]() __attribute__((always_inline)) {
#pragma unroll
      for (int en = 0; en < 4; ++en)
#pragma unroll
        for (int e = 0; e < 4; ++e)
          *reinterpret_cast<f32x4*>(mine + (16 * fq + 4 * e + en) * TST + 4 * fr) =
              f32x4{acc[en][0][e], acc[en][1][e], acc[en][2][e], acc[en][3][e]};
    };
    if constexpr (LDS2) {
      if (wave >= 2) put_tile();
      __syncthreads();
      if (wave < 2) {
#pragma unroll
        for (int en = 0; en < 4; ++en)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const f32x4 o = *reinterpret_cast<const f32x4*>(mine + (16 * fq + 4 * e + en) * TST + 4 * fr);
#pragma unroll
            for (int ec = 0; ec < 4; ++ec) acc[en][ec][e] += o[ec];
          }
        put_tile();                       // its own slot again: no other wave touches it before the barrier below
      }
    } else {
      put_tile();
    }
    // bias partial: sum over the four row groups fq, then lanes fq == 0 hold n = 4 fr + e
#pragma unroll
    for (int e = 0; e < 4; ++e) { bsum[e] += __shfl_xor(bsum[e], 16); bsum[e] += __shfl_xor(bsum[e], 32); }
    if (fq == 0) *reinterpret_cast<f32x4*>(dbs + wave * 64 + 4 * fr) = bsum;
  }
  __syncthreads();
  const bool do_bias = p.db != nullptr && by == 0 && tap == 0;
  // thread t owns the float4s e4 = t + 256 j of the tile (n = e4 / 16, c = 4 (e4 % 16))
  f32x4 v[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int e4 = tid + 256 * j, nl = e4 >> 4, cl = (e4 & 15) * 4;
    const float* q = wsm + nl * TST + cl;
    if constexpr (LDS2)
      v[j] = *reinterpret_cast<const f32x4*>(q) + *reinterpret_cast<const f32x4*>(q + 64 * TST);
    else
      v[j] = (*reinterpret_cast<const f32x4*>(q) + *reinterpret_cast<const f32x4*>(q + 64 * TST)) +
             (*reinterpret_cast<const f32x4*>(q + 2 * 64 * TST) + *reinterpret_cast<const f32x4*>(q + 3 * 64 * TST));
  }
  float vb = 0.f;
  if (tid < 64) vb = (dbs[tid] + dbs[64 + tid]) + (dbs[128 + tid] + dbs[192 + tid]);

  if (ksplit > 1) {                      // partial tile for wgrad_reduce_kernel
    float* const mypart = part + ((size_t)tile_id * ksplit + ks) * WG_TS;
#pragma unroll
    for (int j = 0; j < 4; ++j) *reinterpret_cast<f32x4*>(mypart + 4 * (tid + 256 * j)) = v[j];
    if (tid < 64) mypart[4096 + tid] = vb;
    return;
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int e4 = tid + 256 * j, nl = e4 >> 4, cl = (e4 & 15) * 4;
    const int n = n0 + nl;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int c = srad_real_channel(c0 + cl + e, p.grp_real, p.grp_pad, p.cin_real);
      if (n < p.n_real && c < p.cin_real) {
        float* dst = p.dW + ((size_t)n * p.cin_real + c) * p.ntaps + tap;
        *dst += v[j][e] * p.alpha;
      }
    }
  }
  if (do_bias && tid < 64 && n0 + tid < p.n_real) p.db[n0 + tid] += vb * p.alpha;
}

// ------------------------------------------------------------------------------------------
// Weight gradient of an 80 -> 80 channel convolution (DRN-L's 160 RCAB convolutions, src/drn.py:143-158) with ONE
// 80 x 80 tile per tap: the 64 x 64 tiles pad 80 channels to 128 on both sides (2.56x the MFMAs and operand loads).
// Same scheme as wgrad_body - fragments straight from global memory, contraction over the token axis, four waves on
// different rows, partial tiles for the reduce kernel - but a lane owns FIVE consecutive channels (a 16-byte and a
// 4-byte load at channel 5 fr): component e of the quintuple is row fr of sub-tile e, so tile row i of sub-tile en
// stands for channel 5 i + en and the tile is 5 x 5 MFMA tiles.  bf16 MFMA, stride 1, 1 or 9 taps.
// ------------------------------------------------------------------------------------------
constexpr int W80_TS = 84;                          // LDS row stride of a wave's 80 x 80 partial tile
constexpr int W80_PART = 80 * 80 + 80;              // floats per partial: the tile (row-major) and 80 bias sums
constexpr size_t W80_LDS = (size_t)(4 * 80 * W80_TS + 4 * 80) * sizeof(float);
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));

template <bool CONV>
__global__ __launch_bounds__(256) void wgrad80_kernel(const WgradParams p, const int ksplit, float* __restrict__ part) {
  extern __shared__ __attribute__((aligned(16))) float wsm[];
  float* const dbs = wsm + 4 * 80 * W80_TS;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 15, fq = lane >> 4;
  const int L = blockIdx.x;
  const int ks = L % ksplit, tap = L / ksplit;      // row split fastest: one XCD per row range (see wgrad_body)
  const int rows_per = ((p.M + ksplit - 1) / ksplit + 127) / 128 * 128;
  const int mb = ks * rows_per;
  const int me = min(p.M, mb + rows_per);
  const unsigned noff = (unsigned)(5 * fr + p.ycol0), coff = (unsigned)(5 * fr);
  [[maybe_unused]] const int pad = p.ntaps == 9 ? 1 : 0;
  [[maybe_unused]] const int ky = p.ntaps == 9 ? tap / 3 : 0, kx = p.ntaps == 9 ? tap - (tap / 3) * 3 : 0;
  [[maybe_unused]] const int hwo = p.Ho * p.Wo;

  f32x4 acc[5][5];
#pragma unroll
  for (int i = 0; i < 5; ++i)
#pragma unroll
    for (int j = 0; j < 5; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  float bsum[5] = {0.f, 0.f, 0.f, 0.f, 0.f};

  f32x4 av4[2][8], bv4[2][8];
  float av1[2][8], bv1[2][8];
  unsigned okm[2];
  auto load_step = [&](int m0, auto set_c) {
    constexpr int set = decltype(set_c)::value;
    unsigned ok = 0u;
    [[maybe_unused]] int bb = 0, oy = 0, ox = 0;
    if constexpr (CONV) {
      const int mf = min(m0 + 8 * fq, p.M - 1);
      bb = mf / hwo;
      const int rem = mf - bb * hwo;
      oy = rem / p.Wo;
      ox = rem - oy * p.Wo;
    }
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const int m = m0 + 8 * fq + t;
      const int mc = min(m, p.M - 1);
      const float* ap = p.dY + (size_t)mc * p.ldy + noff;
      av4[set][t] = *reinterpret_cast<const f32x4u*>(ap);
      av1[set][t] = ap[4];
      size_t xr = (size_t)mc;
      bool in = true;
      if constexpr (CONV) {
        if (t > 0 && m < p.M) {
          ++ox;
          if (ox == p.Wo) { ox = 0; ++oy; if (oy == p.Ho) { oy = 0; ++bb; } }
        }
        const int iy = oy - pad + ky, ix = ox - pad + kx;
        in = iy >= 0 && iy < p.Hi && ix >= 0 && ix < p.Wi;
        xr = (size_t)((bb * p.Hi + min(max(iy, 0), p.Hi - 1)) * p.Wi + min(max(ix, 0), p.Wi - 1));
      }
      const float* bp = p.X + xr * p.ldx + coff;
      bv4[set][t] = *reinterpret_cast<const f32x4u*>(bp);
      bv1[set][t] = bp[4];
      ok |= (m < me ? 1u : 0u) << t;
      ok |= ((m < me && in) ? 1u : 0u) << (8 + t);
    }
    okm[set] = ok;
  };
  auto compute_step = [&](auto set_c) {
    constexpr int set = decltype(set_c)::value;
    bf16x8 ah[5], bh[5];
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const bool aok = (okm[set] >> t) & 1u, bok = (okm[set] >> (8 + t)) & 1u;
#pragma unroll
      for (int e = 0; e < 5; ++e) {
        const float a = aok ? (e < 4 ? av4[set][t][e < 4 ? e : 0] : av1[set][t]) : 0.f;
        const float b = bok ? (e < 4 ? bv4[set][t][e < 4 ? e : 0] : bv1[set][t]) : 0.f;
        bsum[e] += a;
        ah[e][t] = (__bf16)a;
        bh[e][t] = (__bf16)b;
      }
    }
#pragma unroll
    for (int en = 0; en < 5; ++en)
#pragma unroll
      for (int ec = 0; ec < 5; ++ec)
        acc[en][ec] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[en], bh[ec], acc[en][ec], 0, 0, 0);
  };
  {
    using S0 = std::integral_constant<int, 0>;
    using S1 = std::integral_constant<int, 1>;
    int m0 = mb + wave * 32;
    if (m0 < me) load_step(m0, S0{});
    while (m0 < me) {
      if (m0 + 128 < me) load_step(m0 + 128, S1{});
      compute_step(S0{});
      m0 += 128;
      if (m0 >= me) break;
      if (m0 + 128 < me) load_step(m0 + 128, S0{});
      compute_step(S1{});
      m0 += 128;
    }
  }
  // ---- the four waves' partial tiles through LDS: lane (fq, fr) element e of acc[en][ec] is
  //      (n = 5 (4 fq + e) + en, c = 5 fr + ec) ----
  {
    float* const mine = wsm + wave * 80 * W80_TS;
#pragma unroll
    for (int en = 0; en < 5; ++en)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float* row = mine + (5 * (4 * fq + e) + en) * W80_TS + 5 * fr;
#pragma unroll
        for (int ec = 0; ec < 5; ++ec) row[ec] = acc[en][ec][e];
      }
#pragma unroll
    for (int e = 0; e < 5; ++e) { bsum[e] += __shfl_xor(bsum[e], 16); bsum[e] += __shfl_xor(bsum[e], 32); }
    if (fq == 0) {
#pragma unroll
      for (int e = 0; e < 5; ++e) dbs[wave * 80 + 5 * fr + e] = bsum[e];
    }
  }
  __syncthreads();
  const bool do_bias = p.db != nullptr && tap == 0;
  float* const mypart = part + ((size_t)tap * ksplit + ks) * W80_PART;
  for (int idx = tid; idx < 1600; idx += 256) {           // float4 idx of the 80 x 80 tile: n = idx / 20, c = 4 (idx % 20)
    const int n = idx / 20, c4 = (idx - n * 20) * 4;
    const float* q = wsm + n * W80_TS + c4;
    const f32x4 v = (*reinterpret_cast<const f32x4*>(q) + *reinterpret_cast<const f32x4*>(q + 80 * W80_TS)) +
                    (*reinterpret_cast<const f32x4*>(q + 2 * 80 * W80_TS) + *reinterpret_cast<const f32x4*>(q + 3 * 80 * W80_TS));
    if (ksplit > 1) {
      *reinterpret_cast<f32x4*>(mypart + 4 * idx) = v;
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) p.dW[((size_t)n * 80 + c4 + e) * p.ntaps + tap] += v[e] * p.alpha;
    }
  }
  if (tid < 80) {
    const float vb = (dbs[tid] + dbs[80 + tid]) + (dbs[160 + tid] + dbs[240 + tid]);
    if (ksplit > 1) mypart[6400 + tid] = vb;
    else if (do_bias) p.db[tid] += vb * p.alpha;
  }
}

// ------------------------------------------------------------------------------------------
// Weight gradient of a 3x3, stride-1, C -> C channel convolution (C <= 80: DRN-L's 160 RCAB convolutions,
// src/drn.py:143-158) with ALL NINE taps in one workgroup.  The per-tap kernels above read dY and X nine times (9 x 21 MB per
// layer at 64 px, batch 8) in workgroups of four waves that live for four row steps: 38 us at 64 px, ~130 us at 128 px.
// Here a workgroup stages a 4 x 32 pixel tile of dY and its 6 x 34 halo tile of X ONCE, as bf16, in LDS ([pixel][channel],
// the memory order).  Both MFMA operands are k-contiguous along the PIXEL axis, i.e. transposed with respect to the tiles,
// so they come out of LDS with ds_read_tr16_b64, and a tap is a constant row offset into the halo tile.  MFMA wave t of eight
// owns tap t's C x C accumulator; the ninth tap's tiles are one tile COLUMN for each of the first C / 16 waves.
// A workgroup walks `cpw` consecutive tiles, so the partial tiles (9 C^2 floats per workgroup, the cost of split-K here)
// are amortised.  Bias sums: fp32, by the staging threads.
// Versions measured on the way (80 channels, 64 / 128 px, alone on the chip; DESIGN.md 4b has the table): nine waves staging
// and computing in turns 37 / 68 us; eight waves with the next tile prefetched into registers behind the MFMAs 31.4 / 52.7 us;
// the staging on four loader waves of its own (this kernel) 29.3 / 45.3 us.
// ------------------------------------------------------------------------------------------
constexpr int WC9_TW = 32, WC9_TR = 4, WC9_PT = WC9_TW * WC9_TR, WC9_HW = WC9_TW + 2, WC9_HT = (WC9_TR + 2) * WC9_HW;
template <int NT> struct Wc9 {
  // LDS row stride (bf16): an odd multiple of 32 bytes.  A transposing read's 32 lanes (one phase of the 64 banks) then take
  // EIGHT CONSECUTIVE pixel rows x 32 bytes = every bank once; which pixel stands for which k of the MFMA is free as long as
  // dY and X agree, so lane (fq, tq) reads pixels 4 fq + tq and 16 + 4 fq + tq of a 32-pixel step.  (With pixels 8 fq + tq
  // and + 4 - the operand layout read literally - rows 0-3 and 8-11 share a phase and collide for EVERY stride that is a
  // multiple of 32 bytes: SQ_LDS_BANK_CONFLICT was 50 % of the LDS cycles, and LDS reads are what bounds this kernel.)
  static constexpr int HS = NT == 2 ? 48 : NT == 4 ? 80 : 16 * NT;
  static constexpr size_t LDS = (size_t)2 * (WC9_PT + WC9_HT) * HS * sizeof(__bf16);       // two tile buffers (>= the 4 KB bias exchange)
};

// Twelve waves: eight MFMA waves (accumulators and fragments only) and FOUR LOADER WAVES (waves 8 - 11), three per SIMD = 168
// registers each.  The loader waves hold the next-but-one tile's rows in registers (28 float4 per thread) and convert the next
// tile into the other LDS buffer WHILE the MFMA waves work on this one.  Both roles pass exactly one barrier per tile (and one
// before and one after the loop); each role has a loop of its own so that the loaders' outstanding loads never meet a join.
constexpr int WC9S_THREADS = 768, WC9S_LOADERS = 256;

// YH / XH: dY / X are bf16 arrays (DRN's bf16 training chain: the loaders move half the bytes and convert nothing; the MFMA
// operands are these bf16 values either way, only the bias sums see the rounded gradient)
template <int NT, bool YH = false, bool XH = false>
__global__ __launch_bounds__(WC9S_THREADS) void wgrad_conv9_kernel(const WgradParams p, const int cpw, const int nchunks,
                                                                    const int ksplit, float* __restrict__ part) {
  constexpr int HS = Wc9<NT>::HS, PT = WC9_PT, HT = WC9_HT, HW = WC9_HW;
  constexpr int ROWS_MIN = WC9S_LOADERS / (4 * NT);            // loader thread rows (threads / channel float4s), at least
  constexpr int NQY_MIN = ROWS_MIN / 4, NQX_MIN = ROWS_MIN / 6 < WC9_HW ? ROWS_MIN / 6 : WC9_HW;
  constexpr int NY = (WC9_TW + NQY_MIN - 1) / NQY_MIN, NX = (WC9_HW + NQX_MIN - 1) / NQX_MIN;
  static_assert(NY + NX <= 32, "item masks are one 32-bit word");
  constexpr int BUF = (PT + HT) * HS;
  typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;
  extern __shared__ __attribute__((aligned(16))) float wsm[];
  const int C = p.N, c4n = C >> 2, W = p.Wo, H = p.Ho;
  __bf16* const lds0 = reinterpret_cast<__bf16*>(wsm);
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ks = blockIdx.x;
  const int c0 = ks * cpw, c_end = min(nchunks, c0 + cpw);
  const int PART = C * C + C;
  const int rows = WC9S_LOADERS / c4n, nqy = rows >> 2, nqx = min(WC9_HW, rows / 6);

  if (wave >= 8) {
    // =========================== loader waves ===========================
    const int tiles_x = W / WC9_TW;
    const int cpi = ((H + WC9_TR - 1) / WC9_TR) * tiles_x;
    // a loader thread keeps ONE channel float4 (sch) and ONE row of the tile (dY: yy of 4, X: hy of 6) and walks the row's
    // pixels with a constant stride; the roles are derived again from the thread id wherever they are used (see above)
    struct Roles { int sch, t2, yy, qy, hy, qx; };
    auto roles = [&]() __attribute__((always_inline)) -> Roles {
      int t = threadIdx.x - 512;
      asm volatile("" : "+v"(t));
      const int t2 = t / c4n;
      const int h = t2 / 6;
      return Roles{t - t2 * c4n, t2, t2 & 3, t2 >> 2, t2 - h * 6, h};
    };
    unsigned st = 0u;                                          // items that exist: bit i: x < 32, bit NY + i: hx < 34
    {
      const Roles ro = roles();
#pragma unroll
      for (int i = 0; i < NY; ++i) if (ro.qy < nqy && ro.qy + nqy * i < WC9_TW) st |= 1u << i;
#pragma unroll
      for (int i = 0; i < NX; ++i) if (ro.qx < nqx && ro.qx + nqx * i < WC9_HW) st |= 1u << (NY + i);
    }
    typedef typename std::conditional<YH, u32x2, f32x4>::type vy_t;
    typedef typename std::conditional<XH, u32x2, f32x4>::type vx_t;
    const vy_t* const dYb = reinterpret_cast<const vy_t*>(reinterpret_cast<const char*>(p.dY) + (size_t)p.ycol0 * (YH ? 2 : 4));
    const vx_t* const Xb = reinterpret_cast<const vx_t*>(p.X);
    vy_t vy[NY];
    vx_t vx[NX];
    f32x4 bsum = f32x4{0.f, 0.f, 0.f, 0.f};
    unsigned okm = 0u;
    auto issue = [&](const int chunk) __attribute__((always_inline)) {
      const int b = chunk / cpi, r = chunk - b * cpi;
      const int ty = r / tiles_x, tx = r - ty * tiles_x;
      const int y0 = ty * WC9_TR, x0 = tx * WC9_TW;
      const unsigned img = (unsigned)b * H;
      const Roles ro = roles();
      const int sch = ro.sch, qx = ro.qx;
      const int y = y0 + ro.yy, iy = y0 - 1 + ro.hy;
      okm = y < H ? (st & ((1u << NY) - 1u)) : 0u;
      const unsigned oy = ((img + min(y, H - 1)) * W + x0 + min(ro.qy, WC9_TW - 1)) * p.ldy + 4 * sch;
      unsigned sy = (unsigned)nqy * p.ldy;
      int nqx_o = nqx;
      asm volatile("" : "+v"(sy), "+v"(nqx_o));                 // per-item offsets are recomputed, not kept in registers between tiles
#pragma unroll
      for (int i = 0; i < NY; ++i) {
        const unsigned off = ((st >> i) & 1u) ? oy + i * sy : oy;
        vy[i] = dYb[off >> 2];                                   // off: elements, a multiple of 4
      }
      const unsigned rowx = (img + min(max(iy, 0), H - 1)) * W;
      const bool rok = iy >= 0 && iy < H;
#pragma unroll
      for (int i = 0; i < NX; ++i) {
        const int ix = x0 - 1 + qx + nqx_o * i;
        if (rok && (unsigned)ix < (unsigned)W) okm |= st & (1u << (NY + i));
        const unsigned off = (rowx + min(max(ix, 0), W - 1)) * p.ldx + 4 * sch;
        vx[i] = Xb[off >> 2];
      }
    };
    auto to_h4 = [](const f32x4 v) __attribute__((always_inline)) -> bf16x4 {
      bf16x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = (__bf16)v[e];
      return o;
    };
    auto store = [&](__bf16* const buf) __attribute__((always_inline)) {
      const Roles ro = roles();
      __bf16* const yrow = buf + (ro.yy * WC9_TW + ro.qy) * HS + 4 * ro.sch;
      __bf16* const xrow = buf + (PT + ro.hy * HW + ro.qx) * HS + 4 * ro.sch;
      int ysl = nqy * HS, xsl = nqx * HS;
      asm volatile("" : "+v"(ysl), "+v"(xsl));
#pragma unroll
      for (int i = 0; i < NY; ++i) {
        if constexpr (YH) {
          const u32x2 v = ((okm >> i) & 1u) ? vy[i] : u32x2{0u, 0u};
          const bf16x4 h = __builtin_bit_cast(bf16x4, v);
          bsum += f32x4{(float)h[0], (float)h[1], (float)h[2], (float)h[3]};
          if ((st >> i) & 1u) *reinterpret_cast<u32x2*>(yrow + i * ysl) = v;
        } else {
          const f32x4 v = ((okm >> i) & 1u) ? vy[i] : f32x4{0.f, 0.f, 0.f, 0.f};
          bsum += v;
          if ((st >> i) & 1u) *reinterpret_cast<bf16x4*>(yrow + i * ysl) = to_h4(v);
        }
      }
#pragma unroll
      for (int i = 0; i < NX; ++i) {
        if constexpr (XH) {
          const u32x2 v = ((okm >> (NY + i)) & 1u) ? vx[i] : u32x2{0u, 0u};
          if ((st >> (NY + i)) & 1u) *reinterpret_cast<u32x2*>(xrow + i * xsl) = v;
        } else {
          const f32x4 v = ((okm >> (NY + i)) & 1u) ? vx[i] : f32x4{0.f, 0.f, 0.f, 0.f};
          if ((st >> (NY + i)) & 1u) *reinterpret_cast<bf16x4*>(xrow + i * xsl) = to_h4(v);
        }
      }
    };
    issue(c0);
    store(lds0);
    issue(min(c0 + 1, c_end - 1));
    __syncthreads();                                           // tile c0 is in buffer 0
    for (int c = c0; c < c_end; ++c) {
      if (c + 1 < c_end) store(lds0 + (((c + 1 - c0) & 1) ? BUF : 0));
      issue(min(c + 2, c_end - 1));                            // unconditional: clamped re-reads at the end of the range
      __syncthreads();
    }
    // bias sums: one float4 per loader thread, summed over the thread rows in fixed order by the first MFMA wave
    f32x4* const bs = reinterpret_cast<f32x4*>(wsm);
    {
      const Roles ro = roles();
      if (ro.qy < nqy) bs[ro.t2 * c4n + ro.sch] = bsum;
    }
    __syncthreads();
    return;
  }

  // =========================== MFMA waves ===========================
  const int lane = tid & 63;
  const int fr = lane & 15, fq = lane >> 4, tq = fr >> 2, tp = fr & 3;
  const int ky = wave / 3, kx = wave - ky * 3;
  // the ninth tap: wave w < NT takes COLUMN w of its NT x NT MFMA tiles - the dY fragments it holds anyway and one more X
  // fragment (dealing the tiles w, w + 8, .. over all eight waves cost two fragment reads per MFMA: 16 of a wave's 36
  // transposing reads per step for 4 of its 29 MFMAs, and LDS reads are what bounds a tile)
  f32x4 acc[NT][NT], accx[NT];
#pragma unroll
  for (int i = 0; i < NT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < NT; ++i) accx[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  auto tr8 = [&](const __bf16* r0) __attribute__((always_inline)) -> bf16x8 {
    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(r0));
    const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(r0 + 16 * HS));
    bf16x8 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) { o[e] = lo[e]; o[4 + e] = hi[e]; }
    return o;
  };
  __syncthreads();                                             // tile c0 is in buffer 0
  for (int c = c0; c < c_end; ++c) {
    const __bf16* const cur = lds0 + (((c - c0) & 1) ? BUF : 0);
#pragma unroll 1
    for (int s4 = 0; s4 < WC9_TR; ++s4) {
      const __bf16* const arow = cur + (s4 * 32 + 4 * fq + tq) * HS + 4 * tp;
      const __bf16* const brow = cur + (PT + (s4 + ky) * HW + kx + 4 * fq + tq) * HS + 4 * tp;
      const __bf16* const b8row = cur + (PT + (s4 + 2) * HW + 2 + 4 * fq + tq) * HS + 4 * tp;
      bf16x8 ah[NT];
#pragma unroll
      for (int t = 0; t < NT; ++t) ah[t] = tr8(arow + 16 * t);
      bf16x8 bh = tr8(brow);
#pragma unroll
      for (int ec = 0; ec < NT; ++ec) {                         // the next X fragment is in flight during this one's MFMAs
        const bf16x8 bn = tr8(ec + 1 < NT ? brow + 16 * (ec + 1) : b8row + 16 * min(wave, NT - 1));
#pragma unroll
        for (int en = 0; en < NT; ++en) acc[en][ec] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh, ah[en], acc[en][ec], 0, 0, 0);
        bh = bn;
      }
      if (wave < NT) {                                          // bh: column `wave` of the ninth tap's X fragments (a dummy read otherwise)
#pragma unroll
        for (int en = 0; en < NT; ++en) accx[en] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh, ah[en], accx[en], 0, 0, 0);
      }
    }
    __syncthreads();
  }
  // ---- partial tiles.  The MFMAs compute the TRANSPOSED tiles (X fragment as the A operand): lane (fq, fr) holds the four
  //      consecutive input channels c = 16 ec + 4 fq + 0..3 of output channel n = 16 en + fr - one float4 of the row-major
  //      partial per tile (the other way round a lane holds four ROWS: 116 four-byte stores per lane instead of 30 float4; the
  //      launch takes the same time either way, the stores are not what its ~13 us of fixed cost are).
  //      The indices are derived again from the thread id (kept across the loop they spilled) ----
  int tid_e = threadIdx.x;
  asm volatile("" : "+v"(tid_e));
  const int fr_e = tid_e & 15, fq_e = (tid_e >> 4) & 3;
  {
    float* const mypart = part + ((size_t)wave * ksplit + ks) * PART;
#pragma unroll
    for (int en = 0; en < NT; ++en) {
      const int n = 16 * en + fr_e;
#pragma unroll
      for (int ec = 0; ec < NT; ++ec) {
        const int c = 16 * ec + 4 * fq_e;
        if (n < C && c < C) *reinterpret_cast<f32x4*>(mypart + n * C + c) = acc[en][ec];
      }
    }
    if (wave < NT) {
      float* const part8 = part + ((size_t)8 * ksplit + ks) * PART;
#pragma unroll
      for (int en = 0; en < NT; ++en) {
        const int n = 16 * en + fr_e, c = 16 * wave + 4 * fq_e;
        if (n < C && c < C) *reinterpret_cast<f32x4*>(part8 + n * C + c) = accx[en];
      }
    }
  }
  __syncthreads();                                             // the loaders' bias sums are in LDS
  if (tid_e < c4n) {
    const f32x4* const bs = reinterpret_cast<const f32x4*>(wsm);
    f32x4 t = bs[tid_e];
    for (int r = 1; r < 4 * nqy; ++r) t += bs[r * c4n + tid_e];
    *reinterpret_cast<f32x4*>(part + (size_t)ks * PART + C * C + 4 * tid_e) = t;
  }
}

template <int PREC, bool CONV>
__global__ __launch_bounds__(256) void wgrad_kernel(const WgradParams p, const int ksplit, const int tn, const int tc,
                                                    float* __restrict__ part) {
  wgrad_body<PREC, CONV>(p, ksplit, tn, tc, part, blockIdx.x);
}

// Several Linear layers' weight gradients in ONE launch (the five of a Swin block): fewer launch ramps on the side
// stream, and the layers' workgroups fill the chip together.  Each layer's block range starts at a multiple of 8 so
// the XCD mapping of wgrad_body holds.
template <int PREC, bool FULL>
__global__ __launch_bounds__(256, FULL ? 2 : 1) void wgrad_multi_kernel(const WgradMulti mp) {
  int i = 0;
#pragma unroll
  for (int k = 1; k < SRAD_WGRAD_MULTI; ++k)
    if (k < mp.count && (int)blockIdx.x >= mp.blk0[k]) i = k;
  const int L = blockIdx.x - mp.blk0[i];
  if (L >= mp.nblk[i]) return;                       // padding blocks between layers
  if constexpr (PREC == SRAD_PREC_BF16) {
    // operand storage is per layer (the adjust conv's gradient operands stay fp32): workgroup-uniform branch
    if (mp.p[i].x_bf16 && mp.p[i].dy_bf16) { wgrad_body<PREC, false, FULL, true, true>(mp.p[i], mp.ksplit[i], mp.tn[i], mp.tc[i], mp.part[i], L); return; }
    if (mp.p[i].x_bf16) { wgrad_body<PREC, false, FULL, true, false>(mp.p[i], mp.ksplit[i], mp.tn[i], mp.tc[i], mp.part[i], L); return; }
  }
  wgrad_body<PREC, false, FULL>(mp.p[i], mp.ksplit[i], mp.tn[i], mp.tc[i], mp.part[i], L);
}

// The same for a launch whose layers ALL have both operands stored as bf16 (the training step's blocks 1-4 of every RDG):
// only that body, so the kernel's register allocation is that body's, and with the two-slot cross-wave sum (36 KB of LDS)
// four workgroups fit a CU: with ONE register set of operand rows it needs 128 VGPRs (four waves per SIMD cover each other's loads).
__global__ __launch_bounds__(256, 4) void wgrad_multi_hh_kernel(const WgradMulti mp) {
  int i = 0;
#pragma unroll
  for (int k = 1; k < SRAD_WGRAD_MULTI; ++k)
    if (k < mp.count && (int)blockIdx.x >= mp.blk0[k]) i = k;
  const int L = blockIdx.x - mp.blk0[i];
  if (L >= mp.nblk[i]) return;
  wgrad_body<SRAD_PREC_BF16, false, true, true, true, true, false>(mp.p[i], mp.ksplit[i], mp.tn[i], mp.tc[i], mp.part[i], L);
}

// One workgroup per QUARTER of a 64 x 64 output tile of one of the batch's layers (16 rows n, one float4 per
// thread): dW += alpha * sum_k partial[k], k in fixed order.
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const WgradReduceBatch b) {
  const int gt = blockIdx.x >> 2, quarter = blockIdx.x & 3;
  int it = 0;
#pragma unroll
  for (int i = 1; i < SRAD_WGRAD_BATCH; ++i)
    if (i < b.count && gt >= b.it[i].tile0) it = i;
  const WgradReduceItem& d = b.it[it];
  const int tile_id = gt - d.tile0;
  if (d.ntaps == 0) {
    // column sums: dst[c] += sum_k part[k * cin_real + c], c < n_real (LayerNorm dgamma / dbeta, bias-table
    // gradient); 16 columns per workgroup, 16 row phases per column
    __shared__ float red[16][17];
    const int r = threadIdx.x >> 4, cl = threadIdx.x & 15;
    const int col = (tile_id * 4 + quarter) * 16 + cl;
    const int cc = min(col, d.n_real - 1);
    float sum = 0.f;
    int k = r;
    for (; k + 112 < d.ksplit; k += 128) {                      // 8 loads in flight (512 partial rows of a LayerNorm: 32 loads per thread)
      float t[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) t[u] = d.part[(size_t)(k + 16 * u) * d.cin_real + cc];
#pragma unroll
      for (int u = 0; u < 8; ++u) sum += t[u];
    }
    for (; k < d.ksplit; k += 16) sum += d.part[(size_t)k * d.cin_real + cc];
    red[r][cl] = sum;
    __syncthreads();
    if (threadIdx.x < 16 && col < d.n_real) {
      float t = 0.f;
#pragma unroll
      for (int i = 0; i < 16; ++i) t += red[i][cl];
      d.dW[col] += t * d.alpha;
    }
    return;
  }
  if (d.wc) {
    // one wc x wc tile per tap (wgrad80_kernel, wgrad_conv9_kernel): row-major + wc bias sums per partial.  d.tn reduce
    // "tiles" (x 4 workgroups) per tap, d.tc float4 of the tile per workgroup (32 x 50 for 80 channels); the split-K partials
    // are dealt over four thread groups and combined in fixed order.  Lanes d.tc and d.tc + 1 of the tap-0 workgroups take a
    // float4 of the bias sums each.  (Four workgroups per tap - 36 for a whole 3x3 layer - with one thread group and 80 threads
    // walking the bias sums took 36 us for 15 MB: 3.2 ms of a DRN-L training step's side stream.)
    __shared__ f32x4 ph[4][64];
    const int C = d.wc, cc4 = C * C / 4, c4n = C / 4, per = d.tc;
    const int PART = C * C + C;
    const int tap = tile_id / d.tn, sub = (tile_id - tap * d.tn) * 4 + quarter;
    const float* const tb = d.part + (size_t)tap * d.ksplit * PART;
    const int il = threadIdx.x & 63, phase = threadIdx.x >> 6;
    const int bidx = sub * 2 + (il - per);                       // bias float4 of lanes per, per + 1
    const bool is_w = il < per && sub * per + il < cc4;
    const bool is_b = il >= per && il < per + 2 && tap == 0 && bidx < c4n && d.db != nullptr;
    const int idx = is_w ? sub * per + il : (is_b ? cc4 + bidx : 0);
    f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
    int k = phase;
    for (; k + 28 < d.ksplit; k += 32) {                        // 8 loads in flight
      f32x4 t[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) t[u] = *reinterpret_cast<const f32x4*>(tb + (size_t)(k + 4 * u) * PART + 4 * idx);
#pragma unroll
      for (int u = 0; u < 8; ++u) v += t[u];
    }
    for (; k < d.ksplit; k += 4) v += *reinterpret_cast<const f32x4*>(tb + (size_t)k * PART + 4 * idx);
    ph[phase][il] = v;
    __syncthreads();
    if (threadIdx.x < 64 && (is_w || is_b)) {
      const f32x4 t = (ph[0][il] + ph[1][il]) + (ph[2][il] + ph[3][il]);
      if (is_w) {
        const int n = idx / c4n, c0 = (idx - n * c4n) * 4;
#pragma unroll
        for (int e = 0; e < 4; ++e) d.dW[((size_t)n * C + c0 + e) * d.ntaps + tap] += t[e] * d.alpha;
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) d.db[4 * bidx + e] += t[e] * d.alpha;
      }
    }
    return;
  }
  const int nx = tile_id % d.tn, cy = (tile_id / d.tn) % d.tc, tap = tile_id / (d.tn * d.tc);
  const int tid = threadIdx.x;
  const int e4 = quarter * 256 + tid;
  const float* const base = d.part + (size_t)tile_id * d.ksplit * WG_TS + 4 * e4;
  f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
  int k = 0;
  for (; k + 8 <= d.ksplit; k += 8) {                          // 8 loads in flight
    f32x4 t[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) t[u] = *reinterpret_cast<const f32x4*>(base + (size_t)(k + u) * WG_TS);
#pragma unroll
    for (int u = 0; u < 8; ++u) v += t[u];
  }
  for (; k < d.ksplit; ++k) v += *reinterpret_cast<const f32x4*>(base + (size_t)k * WG_TS);
  const int n0 = nx * 64, c0 = cy * 64;
  const int n = n0 + (e4 >> 4), cl = (e4 & 15) * 4;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int c = srad_real_channel(c0 + cl + e, d.grp_real, d.grp_pad, d.cin_real);
    if (n < d.n_real && c < d.cin_real) {
      float* dst = d.dW + ((size_t)n * d.cin_real + c) * d.ntaps + tap;
      *dst += v[e] * d.alpha;
    }
  }
  if (d.db != nullptr && cy == 0 && tap == 0 && tid < 16 && n0 + quarter * 16 + tid < d.n_real) {
    const float* bb = d.part + (size_t)tile_id * d.ksplit * WG_TS + 4096 + quarter * 16 + tid;
    float vb = 0.f;
    int kk = 0;
    for (; kk + 8 <= d.ksplit; kk += 8) {                      // 8 loads in flight (one after the other they were the launch's tail)
      float t[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) t[u] = bb[(size_t)(kk + u) * WG_TS];
#pragma unroll
      for (int u = 0; u < 8; ++u) vb += t[u];
    }
    for (; kk < d.ksplit; ++kk) vb += bb[(size_t)kk * WG_TS];
    d.db[n0 + quarter * 16 + tid] += vb * d.alpha;
  }
}

struct WgradPlan { int tn, tc, ksplit; long tiles; float* part; };

// tile / split geometry of one layer, its partial-tile workspace and its entry in the reduce batch
template <int PREC>
int plan_wgrad(const WgradParams& p, WgradQueue& q, hipStream_t s, WgradPlan& pl, const long wg_target = 512) {
  pl.tn = (p.N + 63) / 64; pl.tc = (p.Cin + 63) / 64;
  pl.tiles = (long)pl.tn * pl.tc * p.ntaps;
  constexpr int KR = PREC == SRAD_PREC_BF16 ? 32 : 4;
  // about two workgroups per CU for a launch of its own (wg_target 512; layers that share a launch ask for fewer,
  // longer workgroups: less ramp, fewer partial tiles), at least two row steps per wave; a power of two, so that
  // from 8 up it is a multiple of 8 (one XCD per row range, see the kernel) and the row ranges come out equal
  long target = (wg_target + pl.tiles - 1) / pl.tiles;
  const long kmax = (p.M + 8 * KR - 1) / (8 * KR);
  if (target > kmax) target = kmax;
  long ksplit = 1;
  while (ksplit * 2 <= kmax && ksplit * 10 <= target * 7) ksplit *= 2;      // nearest power of two (rounding up from 1.43x)
  if (const char* e = getenv("SRAD_WGRAD_KSPLIT")) ksplit = atoi(e) > 0 ? atoi(e) : ksplit;   // tools/: timing experiments
  {  // drop empty splits: rows_per is rounded up to whole steps
    const long rows_per = ((p.M + ksplit - 1) / ksplit + 4 * KR - 1) / (4 * KR) * (4 * KR);
    ksplit = (p.M + rows_per - 1) / rows_per;
  }
  pl.ksplit = (int)ksplit;
  pl.part = nullptr;
  if (ksplit > 1) {
    const size_t need = (size_t)pl.tiles * ksplit * WG_TS;
    SRAD_REQUIRE(q.ws && need <= q.ws_floats, "wgrad: split-K workspace too small (%zu floats needed, %zu given)", need, q.ws_floats);
    if (q.batch.count == SRAD_WGRAD_BATCH || q.used + need > q.ws_floats) {
      SRAD_REQUIRE(q.multi.count == 0, "wgrad: split-K workspace too small for the deferred layers");
      SRAD_TRY(srad_wgrad_flush(q, s));
    }
    pl.part = q.ws + q.used;
    q.used += need;
    WgradReduceItem& it = q.batch.it[q.batch.count++];
    it.dW = p.dW; it.db = p.db; it.part = pl.part; it.n_real = p.n_real; it.cin_real = p.cin_real; it.ntaps = p.ntaps;
    it.grp_real = p.grp_real; it.grp_pad = p.grp_pad;
    it.tn = pl.tn; it.tc = pl.tc; it.ksplit = (int)ksplit; it.tile0 = q.tiles; it.alpha = p.alpha; it.wc = 0;
    q.tiles += (int)pl.tiles;
  }
  return SRAD_OK;
}

constexpr size_t WG_LDS = (size_t)(4 * 64 * 68 + 4 * 64) * sizeof(float);
constexpr size_t WG_LDS2 = (size_t)(2 * 64 * 68 + 4 * 64) * sizeof(float);   // wgrad_body<.., LDS2 = true>

// reduce geometry of a layer whose partials are one C x C tile per tap (see wgrad_reduce_kernel): reduce tiles to add to the queue
static int square_reduce_tiles(WgradReduceItem& it, const int C) {
  const int cc4 = C * C / 4;
  it.wc = C;
  it.tn = (cc4 + 199) / 200;                      // x 4 workgroups per tap, <= 50 float4 each
  it.tc = (cc4 + 4 * it.tn - 1) / (4 * it.tn);
  return it.ntaps * it.tn;
}

// 80 -> 80 channels, stride 1, bf16: one 80 x 80 tile per tap
int launch_wgrad80(const WgradParams& p, WgradQueue& q, hipStream_t s) {
  // about two workgroups per CU; a power of two from 8 up (one XCD per row range), at least two row steps per wave
  const long kmax = (p.M + 255) / 256;
  long target = (512 + p.ntaps - 1) / p.ntaps;
  if (target > kmax) target = kmax;
  long ksplit = 1;
  while (ksplit * 2 <= kmax && ksplit * 10 <= target * 7) ksplit *= 2;
  {
    const long rows_per = ((p.M + ksplit - 1) / ksplit + 127) / 128 * 128;
    ksplit = (p.M + rows_per - 1) / rows_per;
  }
  float* part = nullptr;
  if (ksplit > 1) {
    const size_t need = (size_t)p.ntaps * ksplit * W80_PART;
    SRAD_REQUIRE(q.ws && need <= q.ws_floats, "wgrad: split-K workspace too small (%zu floats needed, %zu given)", need, q.ws_floats);
    if (q.batch.count == SRAD_WGRAD_BATCH || q.used + need > q.ws_floats) {
      SRAD_REQUIRE(q.multi.count == 0, "wgrad: split-K workspace too small for the deferred layers");
      SRAD_TRY(srad_wgrad_flush(q, s));
    }
    part = q.ws + q.used;
    q.used += need;
    WgradReduceItem& it = q.batch.it[q.batch.count++];
    it.dW = p.dW; it.db = p.db; it.part = part; it.n_real = 80; it.cin_real = 80; it.ntaps = p.ntaps; it.grp_real = it.grp_pad = 0;
    it.ksplit = (int)ksplit; it.tile0 = q.tiles; it.alpha = p.alpha;
    q.tiles += square_reduce_tiles(it, 80);
  }
  SradProfScope prof(s, SRAD_K_WGRAD, 2.0 * p.M * 80.0 * 80.0 * p.ntaps, 4.0 * p.M * 160.0 + 8.0 * 6400.0 * p.ntaps);
  auto launch = [&](auto kern) -> int {
    static SradOncePerDevice configured;
    if (configured.need()) {
      SRAD_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)W80_LDS));
      configured.done();
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)(p.ntaps * ksplit)), dim3(256), W80_LDS, s, p, (int)ksplit, part);
    return SRAD_OK;
  };
  const int rc = p.ntaps == 9 ? launch(wgrad80_kernel<true>) : launch(wgrad80_kernel<false>);
  if (rc) return rc;
  SRAD_CHECK_HIP(hipGetLastError());
  return SRAD_OK;
}

// 3x3 stride-1 C -> C convolution, all nine taps per workgroup (wgrad_conv9_kernel); false: not this layer's kernel
static bool conv9_supported(const WgradParams& p) {
  if (p.ntaps != 9 || p.stride != 1 || p.N != p.Cin || p.n_real != p.N || p.cin_real != p.Cin || p.N > 80 || (p.N & 3) || p.row_scale ||
      ((p.x_bf16 || p.dy_bf16) && !(p.N == 80 && p.dy_bf16)) || p.Hi != p.Ho || p.Wi != p.Wo || (p.Wo % WC9_TW) || (p.ldy & 3) || (p.ldx & 3) || (p.ycol0 & 3) ||
      (size_t)p.M < 8192 || (size_t)p.M * (size_t)std::max(p.ldy, p.ldx) >= ((size_t)1 << 31) || getenv("SRAD_NO_WGRAD_CONV9") != nullptr)
    return false;
  return ((reinterpret_cast<uintptr_t>(p.dY) | reinterpret_cast<uintptr_t>(p.X)) & 15) == 0;
}

int launch_wgrad_conv9(const WgradParams& p, WgradQueue& q, hipStream_t s) {
  const int C = p.N, nt = (C + 15) / 16;
  const int B = p.M / (p.Ho * p.Wo);
  const int nchunks = B * ((p.Ho + WC9_TR - 1) / WC9_TR) * (p.Wo / WC9_TW);
  // workgroups: a 128-pixel tile costs ~2 us, a workgroup ~6 us of ramp plus its partial tiles (9 C^2 floats written and
  // read again): ~4 sqrt(tiles) workgroups balance the two (64 for 256 tiles, 128 for 1024)
  double kmul = 4.0;
  if (const char* e = getenv("SRAD_CONV9_KMUL")) kmul = atof(e) > 0 ? atof(e) : kmul;          // tools/: timing experiments
  int ksplit = (int)(kmul * sqrt((double)nchunks) + 0.5);
  if (const char* e = getenv("SRAD_WGRAD_KSPLIT")) ksplit = atoi(e) > 0 ? atoi(e) : ksplit;   // tools/: timing experiments
  ksplit = std::max(1, std::min(std::min(ksplit, 256), nchunks));
  const int cpw = (nchunks + ksplit - 1) / ksplit;
  ksplit = (nchunks + cpw - 1) / cpw;
  const size_t PART = (size_t)C * C + C;
  const size_t need = 9 * (size_t)ksplit * PART;
  SRAD_REQUIRE(q.ws && need <= q.ws_floats, "wgrad: split-K workspace too small (%zu floats needed, %zu given)", need, q.ws_floats);
  if (q.batch.count == SRAD_WGRAD_BATCH || q.used + need > q.ws_floats) {
    SRAD_REQUIRE(q.multi.count == 0, "wgrad: split-K workspace too small for the deferred layers");
    SRAD_TRY(srad_wgrad_flush(q, s));
  }
  float* const part = q.ws + q.used;
  q.used += need;
  WgradReduceItem& it = q.batch.it[q.batch.count++];
  it.dW = p.dW; it.db = p.db; it.part = part; it.n_real = C; it.cin_real = C; it.ntaps = 9; it.grp_real = it.grp_pad = 0;
  it.ksplit = ksplit; it.tile0 = q.tiles; it.alpha = p.alpha;
  q.tiles += square_reduce_tiles(it, C);
  SradProfScope prof(s, SRAD_K_WGRAD, 2.0 * p.M * C * C * 9.0, (double)p.M * C * ((p.dy_bf16 ? 2 : 4) + (p.x_bf16 ? 2 : 4)) + 8.0 * 9.0 * PART * ksplit);
  auto launch = [&](auto kern, const size_t lds, SradOncePerDevice& configured) -> int {
    if (configured.need()) {
      SRAD_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      configured.done();
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)ksplit), dim3(WC9S_THREADS), lds, s, p, cpw, nchunks, ksplit, part);
    return SRAD_OK;
  };
  static SradOncePerDevice cfg[6], cfg_h[2];
  int rc = SRAD_OK;
  if (p.dy_bf16) {                                              // 80 channels only (conv9_supported): DRN's bf16 training chain
    rc = p.x_bf16 ? launch(wgrad_conv9_kernel<5, true, true>, Wc9<5>::LDS, cfg_h[1]) : launch(wgrad_conv9_kernel<5, true, false>, Wc9<5>::LDS, cfg_h[0]);
    if (rc) return rc;
    SRAD_CHECK_HIP(hipGetLastError());
    return SRAD_OK;
  }
  switch (nt) {
    case 1: rc = launch(wgrad_conv9_kernel<1>, Wc9<1>::LDS, cfg[1]); break;
    case 2: rc = launch(wgrad_conv9_kernel<2>, Wc9<2>::LDS, cfg[2]); break;
    case 3: rc = launch(wgrad_conv9_kernel<3>, Wc9<3>::LDS, cfg[3]); break;
    case 4: rc = launch(wgrad_conv9_kernel<4>, Wc9<4>::LDS, cfg[4]); break;
    default: rc = launch(wgrad_conv9_kernel<5>, Wc9<5>::LDS, cfg[5]); break;
  }
  if (rc) return rc;
  SRAD_CHECK_HIP(hipGetLastError());
  return SRAD_OK;
}

template <int PREC>
int launch_wgrad(const WgradParams& p, WgradQueue& q, hipStream_t s) {
  if (PREC == SRAD_PREC_BF16 && conv9_supported(p)) return launch_wgrad_conv9(p, q, s);
  if (PREC == SRAD_PREC_BF16 && p.N == 80 && p.Cin == 80 && p.n_real == 80 && p.cin_real == 80 && p.stride == 1 && !p.row_scale &&
      (p.ntaps == 1 || (p.Hi == p.Ho && p.Wi == p.Wo)) && getenv("SRAD_NO_WGRAD80") == nullptr)
    return launch_wgrad80(p, q, s);
  const bool conv = p.ntaps == 9 || p.stride != 1;
  WgradPlan pl;
  SRAD_TRY(plan_wgrad<PREC>(p, q, s, pl));
  dim3 grid((unsigned)(pl.tiles * pl.ksplit));
  const double K = (double)p.ntaps * p.cin_real;
  SradProfScope prof(s, SRAD_K_WGRAD, 2.0 * p.M * p.n_real * K, 4.0 * p.M * ((double)p.N + p.Cin) + 8.0 * p.n_real * K);
  auto launch = [&](auto kern) -> int {
    static SradOncePerDevice configured;
    if (configured.need()) {
      SRAD_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)WG_LDS));
      configured.done();
    }
    hipLaunchKernelGGL(kern, grid, dim3(256), WG_LDS, s, p, pl.ksplit, pl.tn, pl.tc, pl.part);
    return SRAD_OK;
  };
  const int rc = conv ? launch(wgrad_kernel<PREC, true>) : launch(wgrad_kernel<PREC, false>);
  if (rc) return rc;
  SRAD_CHECK_HIP(hipGetLastError());
  return SRAD_OK;
}

template <int PREC>
int defer_wgrad(const WgradParams& p, WgradQueue& q, hipStream_t s) {
  SRAD_REQUIRE(p.ntaps == 1 && p.stride == 1, "wgrad: only Linear layers can be deferred");
  if (q.multi.count == SRAD_WGRAD_MULTI) SRAD_TRY(srad_wgrad_launch_deferred(PREC, q, s));
  WgradPlan pl;
  SRAD_TRY(plan_wgrad<PREC>(p, q, s, pl, 144));      // the launch is shared by up to SRAD_WGRAD_MULTI layers
  WgradMulti& m = q.multi;
  const int i = m.count++;
  m.p[i] = p; m.ksplit[i] = pl.ksplit; m.tn[i] = pl.tn; m.tc[i] = pl.tc; m.part[i] = pl.part;
  m.blk0[i] = i == 0 ? 0 : (m.blk0[i - 1] + m.nblk[i - 1] + 7) / 8 * 8;
  m.nblk[i] = (int)(pl.tiles * pl.ksplit);
  q.multi_flops += 2.0 * p.M * p.n_real * (double)p.cin_real;
  q.multi_bytes += 4.0 * p.M * ((double)p.N + p.Cin) + 8.0 * p.n_real * (double)p.cin_real;
  return SRAD_OK;
}

// ------------------------------------------------------------------------------------------
// LayerNorm backward, one wave per row, RPW rows per 16-wave workgroup (two rows per wave, four waves per SIMD
// to cover the load latency).  Mean / rstd are recomputed exactly as the forward kernel does; dgamma / dbeta
// partials live in registers across the wave's rows, are summed over the waves in LDS and left in the split-K
// workspace for wgrad_reduce_kernel.
// ------------------------------------------------------------------------------------------

// HAS_RES / ACC are template flags: a load behind a run-time test is waited for before the next one issues.
// One wave per row, 32 rows per 4-wave workgroup.
// Lane l owns channels [4 l, 4 l + 4) and [256 + 4 l, 256 + 4 l + 4): 16-byte loads, two per tensor per row.
template <bool HAS_RES, bool ACC, int J4>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const LnBwdParams p, float* __restrict__ part) {
  __shared__ __attribute__((aligned(16))) float red[4][2][512];      // per wave: dgamma | dbeta partial rows
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const f32x4 z4 = f32x4{0.f, 0.f, 0.f, 0.f};
  int cofs[J4];
  bool cok[J4];
  f32x4 gam[J4], dg[J4], db[J4];
#pragma unroll
  for (int j = 0; j < J4; ++j) {
    const int c = 4 * lane + 256 * j;
    cok[j] = c < p.C;                          // C % 4 == 0: a float4 is all in or all out
    cofs[j] = min(c, p.C - 4);
    gam[j] = *reinterpret_cast<const f32x4*>(p.gamma + cofs[j]);
    dg[j] = z4; db[j] = z4;
  }
  const float invC = 1.0f / (float)p.C;
  const int r_end = min(p.rows, (int)(blockIdx.x + 1) * LNB_RPW);
  // two rows per wave and trip: their loads are in flight together and the four reduction chains interleave
  for (int row0 = blockIdx.x * LNB_RPW + wave; row0 < r_end; row0 += 8) {
    f32x4 xv[2][J4], dy[2][J4], rv[2][J4], ov[2][J4];
    int rows_[2];
    bool rok[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      rok[u] = row0 + 4 * u < r_end;
      rows_[u] = rok[u] ? row0 + 4 * u : row0;
#pragma unroll
      for (int j = 0; j < J4; ++j) {
        xv[u][j] = *reinterpret_cast<const f32x4*>(p.x + (size_t)rows_[u] * p.ldx + cofs[j]);
        dy[u][j] = *reinterpret_cast<const f32x4*>(p.dxn + (size_t)rows_[u] * p.ld_dxn + cofs[j]);
        rv[u][j] = z4; ov[u][j] = z4;
        if constexpr (HAS_RES) rv[u][j] = *reinterpret_cast<const f32x4*>(p.dres + (size_t)rows_[u] * p.ld_dres + cofs[j]);
        if constexpr (ACC) ov[u][j] = *reinterpret_cast<const f32x4*>(p.out + (size_t)rows_[u] * p.ld_out + cofs[j]);
      }
    }
    float s[2], v[2], s1[2], s2[2], rstd[2];
    f32x4 xh[2][J4], gy[2][J4];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      s[u] = 0.f;
#pragma unroll
      for (int j = 0; j < J4; ++j) {
        xv[u][j] = cok[j] ? xv[u][j] : z4;
        dy[u][j] = cok[j] && rok[u] ? dy[u][j] : z4;
        s[u] += (xv[u][j][0] + xv[u][j][1]) + (xv[u][j][2] + xv[u][j][3]);
      }
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) s[u] = srad_wave_sum(s[u]) * invC;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      v[u] = 0.f;
#pragma unroll
      for (int j = 0; j < J4; ++j) {
        xh[u][j] = cok[j] ? xv[u][j] - s[u] : z4;
        v[u] += (xh[u][j][0] * xh[u][j][0] + xh[u][j][1] * xh[u][j][1]) + (xh[u][j][2] * xh[u][j][2] + xh[u][j][3] * xh[u][j][3]);
      }
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) { v[u] = srad_wave_sum(v[u]); rstd[u] = rsqrtf(v[u] * invC + p.eps); }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      s1[u] = 0.f; s2[u] = 0.f;
#pragma unroll
      for (int j = 0; j < J4; ++j) {
        xh[u][j] = xh[u][j] * rstd[u];
        gy[u][j] = dy[u][j] * gam[j];
        const f32x4 t = gy[u][j] * xh[u][j];
        s1[u] += (gy[u][j][0] + gy[u][j][1]) + (gy[u][j][2] + gy[u][j][3]);
        s2[u] += (t[0] + t[1]) + (t[2] + t[3]);
        dg[j] += dy[u][j] * xh[u][j];
        db[j] += dy[u][j];
      }
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) { s1[u] = srad_wave_sum(s1[u]) * invC; s2[u] = srad_wave_sum(s2[u]) * invC; }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
#pragma unroll
      for (int j = 0; j < J4; ++j) {
        const f32x4 o4 = (gy[u][j] - s1[u] - xh[u][j] * s2[u]) * rstd[u] + rv[u][j] + ov[u][j];
        if (cok[j] && rok[u]) *reinterpret_cast<f32x4*>(p.out + (size_t)rows_[u] * p.ld_out + 4 * lane + 256 * j) = o4;
      }
    }
  }
#pragma unroll
  for (int j = J4; j < 2; ++j) {
    *reinterpret_cast<f32x4*>(&red[wave][0][4 * lane + 256 * j]) = z4;
    *reinterpret_cast<f32x4*>(&red[wave][1][4 * lane + 256 * j]) = z4;
  }
  // LDS float atomics cost ~13 us here (measured); plain 16-byte stores per wave, then a 4-way sum
#pragma unroll
  for (int j = 0; j < J4; ++j) {
    *reinterpret_cast<f32x4*>(&red[wave][0][4 * lane + 256 * j]) = dg[j];
    *reinterpret_cast<f32x4*>(&red[wave][1][4 * lane + 256 * j]) = db[j];
  }
  __syncthreads();
  // per-workgroup column sums -> workspace row [dgamma LNB_CP | dbeta LNB_CP]; wgrad_reduce_kernel adds them up
  float* row = part + (size_t)blockIdx.x * (2 * LNB_CP);
  for (int i = tid; i < 2 * LNB_CP; i += 256) {
    const int which = i >= LNB_CP ? 1 : 0, c = i - which * LNB_CP;
    row[i] = (red[0][which][c] + red[1][which][c]) + (red[2][which][c] + red[3][which][c]);
  }
}

// ------------------------------------------------------------------------------------------
// Window attention backward for 8x8 windows (N = 64 tokens), one workgroup per (window, head).
// Pass A walks the head dimension in chunks of 32 columns accumulating S = (q*scale) k^T and
// dP = dO v^T in MFMA accumulators; the softmax is redone in registers exactly as the forward kernel
// does (bias table + 0/-100 shift mask), dS = P (dP - rowsum(P dP)); P and dS go to LDS.
// Pass B walks the chunks again: dq = scale dS k, dk = dS^T (q*scale), dv = P^T dO.
// All MFMAs are v_mfma_f32_16x16x4_f32 (fp32 in, fp32 out) in both precision modes.
// ------------------------------------------------------------------------------------------
constexpr int AB_HC = 32, AB_HS = AB_HC + 4, AB_PS = 64 + 4;
constexpr size_t AB_LDS = (size_t)(4 * 64 * AB_HS + 2 * 64 * AB_PS + 2 * 256) * sizeof(float) + 2 * 64 * sizeof(int);

__global__ __launch_bounds__(256) void window_attn_bwd_kernel(const AttnBwdParams p, float* __restrict__ tpart) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* Qs = reinterpret_cast<float*>(smem);
  float* Ks = Qs + 64 * AB_HS;
  float* Vs = Ks + 64 * AB_HS;
  float* Gs = Vs + 64 * AB_HS;
  float* Pm = Gs + 64 * AB_HS;
  float* Dm = Pm + 64 * AB_PS;
  float* tbl = Dm + 64 * AB_PS;      // [225] (256 reserved)
  float* dtb = tbl + 256;
  int* tok = reinterpret_cast<int*>(dtb + 256);
  int* inf = tok + 64;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 15, fq = lane >> 4;
  const int ws = 8, d = p.d, heads = p.heads, hd = d / heads, hdp = p.hdp;
  const int ldq = 3 * heads * hdp;
  const int nWx = p.W / ws, nW = (p.H / ws) * nWx;
  // workgroups are dealt round-robin over the 8 XCDs: keep all heads of a window on ONE XCD (its L2 then holds the
  // window's dO / dqkv lines, which the heads share at 4-byte granularity, once)
  int win, h;
  {
    const int L = blockIdx.x, nwin = p.B * nW;
    if ((nwin & 7) == 0) { const int slot = L >> 3; win = (slot / heads) * 8 + (L & 7); h = slot - (slot / heads) * heads; }
    else { win = L / heads; h = L - win * heads; }
  }
  const int b = win / nW, widx = win - b * nW;
  const int wy = widx / nWx, wx = widx - wy * nWx;
  const float scale = rsqrtf((float)hd);
  const int tw = 2 * ws - 1;

  if (tid < 64) {
    const int py = tid / ws, px = tid - py * ws;
    const int r = wy * ws + py, c = wx * ws + px;
    int orr = r + p.shift; if (orr >= p.H) orr -= p.H;
    int occ = c + p.shift; if (occ >= p.W) occ -= p.W;
    tok[tid] = (b * p.H + orr) * p.W + occ;
    const int rh = r < p.H - ws ? 0 : (r < p.H - p.shift ? 1 : 2);
    const int rw = c < p.W - ws ? 0 : (c < p.W - p.shift ? 1 : 2);
    inf[tid] = ((rh * 3 + rw) << 16) | (py << 8) | px;
  }
  if (tid < tw * tw) tbl[tid] = p.table[(size_t)tid * heads + h];
  __syncthreads();

  // chunk staging: q/k/v as float4 (head-padded rows are 16-byte aligned), dO as scalars
  auto stage = [&](int ch, bool need_v) {
    const int col0 = ch * AB_HC;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int idx = tid + 256 * i;
      const int row = idx >> 3, c = col0 + (idx & 7) * 4;
      const float* base = p.qkv + (size_t)tok[row] * ldq + h * hdp + min(c, hdp - 4);
      f32x4 q4 = *reinterpret_cast<const f32x4*>(base);
      f32x4 k4 = *reinterpret_cast<const f32x4*>(base + heads * hdp);
      f32x4 v4 = need_v ? *reinterpret_cast<const f32x4*>(base + 2 * heads * hdp) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const bool ok = c + e < hd;
        q4[e] = ok ? q4[e] * scale : 0.f;
        k4[e] = ok ? k4[e] : 0.f;
        v4[e] = ok ? v4[e] : 0.f;
      }
      *reinterpret_cast<f32x4*>(Qs + row * AB_HS + (idx & 7) * 4) = q4;
      *reinterpret_cast<f32x4*>(Ks + row * AB_HS + (idx & 7) * 4) = k4;
      if (need_v) *reinterpret_cast<f32x4*>(Vs + row * AB_HS + (idx & 7) * 4) = v4;
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int idx = tid + 256 * i;
      const int row = idx >> 5, cl = idx & 31, c = col0 + cl;
      const float g = p.dout[(size_t)tok[row] * d + h * hd + min(c, hd - 1)];
      Gs[row * AB_HS + cl] = c < hd ? g : 0.f;
    }
  };

  const int nch = (hd + AB_HC - 1) / AB_HC;
  f32x4 s[4], dp[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) { s[j] = f32x4{0.f, 0.f, 0.f, 0.f}; dp[j] = f32x4{0.f, 0.f, 0.f, 0.f}; }
  for (int ch = 0; ch < nch; ++ch) {
    if (ch > 0) __syncthreads();
    stage(ch, true);
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < AB_HC; kk += 16) {
      const f32x4 a = *reinterpret_cast<const f32x4*>(Qs + (wave * 16 + fr) * AB_HS + kk + 4 * fq);
      const f32x4 g = *reinterpret_cast<const f32x4*>(Gs + (wave * 16 + fr) * AB_HS + kk + 4 * fq);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const f32x4 kb = *reinterpret_cast<const f32x4*>(Ks + (j * 16 + fr) * AB_HS + kk + 4 * fq);
        const f32x4 vb = *reinterpret_cast<const f32x4*>(Vs + (j * 16 + fr) * AB_HS + kk + 4 * fq);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          s[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[e], kb[e], s[j], 0, 0, 0);
          dp[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(g[e], vb[e], dp[j], 0, 0, 0);
        }
      }
    }
  }

  // ---- softmax (row = 16 wave + 4 fq + e, key = 16 j + fr), dS, bias-table gradient ----
  {
    int kinf[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) kinf[j] = inf[j * 16 + fr];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int row = wave * 16 + fq * 4 + e;
      const int qi = inf[row];
      const int qy = (qi >> 8) & 0xff, qx = qi & 0xff, qr = qi >> 16;
      int bi[4];
      float mx = -1e30f;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int kyy = (kinf[j] >> 8) & 0xff, kxx = kinf[j] & 0xff, kr = kinf[j] >> 16;
        bi[j] = (qy - kyy + ws - 1) * tw + (qx - kxx + ws - 1);
        float v = s[j][e] + tbl[bi[j]];
        if (p.shift > 0 && qr != kr) v += -100.0f;
        s[j][e] = v;
        mx = fmaxf(mx, v);
      }
      mx = srad_row16_max(mx);
      float sum = 0.f;
#pragma unroll
      for (int j = 0; j < 4; ++j) { s[j][e] = expf(s[j][e] - mx); sum += s[j][e]; }
      sum = srad_row16_sum(sum);
      const float inv = 1.0f / sum;
      float dl = 0.f;
#pragma unroll
      for (int j = 0; j < 4; ++j) { s[j][e] *= inv; dl += s[j][e] * dp[j][e]; }
      dl = srad_row16_sum(dl);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float ds = s[j][e] * (dp[j][e] - dl);
        Pm[row * AB_PS + j * 16 + fr] = s[j][e];
        Dm[row * AB_PS + j * 16 + fr] = ds;
      }
    }
  }
  __syncthreads();
  // bias-table gradient of this (window, head): entry t = (dy + 7) * 15 + (dx + 7) collects dS[q][k] over all query
  // positions q whose key k = q - (dy, dx) lies in the window (<= 64 terms), read from the dS tile - no atomics.
  // The row goes to the split-K workspace; wgrad_reduce_kernel sums the windows.
  if (tid < tw * tw) {
    const int dy = tid / tw - (ws - 1), dx = tid - (tid / tw) * tw - (ws - 1);
    float acc = 0.f;
    for (int qy = max(0, dy); qy < min(ws, ws + dy); ++qy)
      for (int qx = max(0, dx); qx < min(ws, ws + dx); ++qx)
        acc += Dm[(qy * ws + qx) * AB_PS + (qy - dy) * ws + (qx - dx)];
    tpart[(size_t)win * (tw * tw * heads) + (size_t)tid * heads + h] = acc;
  }

  // ---- pass B: dq, dk, dv per 32-column chunk ----
  for (int ch = 0; ch < nch; ++ch) {
    if (ch > 0 || nch > 1) { __syncthreads(); stage(ch, false); __syncthreads(); }
    f32x4 dq[2], dk[2], dv[2];
#pragma unroll
    for (int jt = 0; jt < 2; ++jt) { dq[jt] = f32x4{0.f, 0.f, 0.f, 0.f}; dk[jt] = dq[jt]; dv[jt] = dq[jt]; }
#pragma unroll
    for (int kk = 0; kk < 64; kk += 16) {
      const f32x4 a = *reinterpret_cast<const f32x4*>(Dm + (wave * 16 + fr) * AB_PS + kk + 4 * fq);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int kr = kk + 4 * fq + e;
        const float at = Dm[kr * AB_PS + wave * 16 + fr];     // dS^T
        const float pt = Pm[kr * AB_PS + wave * 16 + fr];     // P^T
#pragma unroll
        for (int jt = 0; jt < 2; ++jt) {
          const float kb = Ks[kr * AB_HS + jt * 16 + fr];
          const float qb = Qs[kr * AB_HS + jt * 16 + fr];
          const float gb = Gs[kr * AB_HS + jt * 16 + fr];
          dq[jt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[e], kb, dq[jt], 0, 0, 0);
          dk[jt] = __builtin_amdgcn_mfma_f32_16x16x4f32(at, qb, dk[jt], 0, 0, 0);
          dv[jt] = __builtin_amdgcn_mfma_f32_16x16x4f32(pt, gb, dv[jt], 0, 0, 0);
        }
      }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float* dst = p.dqkv + (size_t)tok[wave * 16 + fq * 4 + e] * (3 * d) + h * hd;
#pragma unroll
      for (int jt = 0; jt < 2; ++jt) {
        const int c = ch * AB_HC + jt * 16 + fr;
        if (c < hd) {
          dst[c] = dq[jt][e] * scale;
          dst[d + c] = dk[jt][e];
          dst[2 * d + c] = dv[jt][e];
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------
// The same backward for ANY window size up to 16 (N = ws^2 <= 256 tokens; the reference's CLI presets build windows of
// 2, 4, 8 and 16: window_size = img_size // 4, src/main.py:286 - the 8 x 8 case has its own kernels above).  One workgroup
// per (window, head); the window's tokens are padded to NB blocks of 64 (padding keys get probability 0, padding queries
// are never written).  For each block of 64 queries the whole score row block (64 x 64 NB) and dP stay in MFMA
// accumulators, so the softmax is exact in one pass as above; then key block after key block the P / dS tiles go to LDS
// and feed dq (accumulated in registers over the key blocks), dk and dv (accumulated over the query blocks by
// read-add-write of the workgroup's own rows: one owner, fixed order).  The bias-table gradient is collected in LDS over
// the tiles.  All MFMAs are v_mfma_f32_16x16x4_f32 in every precision mode: these presets are small images, the kernel is
// for coverage, not for the benchmarked configuration.
// ------------------------------------------------------------------------------------------
template <int NB>
__global__ __launch_bounds__(256) void window_attn_bwd_gen_kernel(const AttnBwdParams p, float* __restrict__ tpart) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* Qs = reinterpret_cast<float*>(smem);
  float* Ks = Qs + 64 * AB_HS;
  float* Vs = Ks + 64 * AB_HS;
  float* Gs = Vs + 64 * AB_HS;
  float* Pm = Gs + 64 * AB_HS;
  float* Dm = Pm + 64 * AB_PS;
  float* tbl = Dm + 64 * AB_PS;      // [(2 ws - 1)^2] <= 961 (1024 reserved)
  float* dtb = tbl + 1024;           // its gradient
  int* tok = reinterpret_cast<int*>(dtb + 1024);   // [64 NB]
  int* inf = tok + 64 * NB;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 15, fq = lane >> 4;
  const int ws = p.ws, N = ws * ws, d = p.d, heads = p.heads, hd = d / heads, hdp = p.hdp;
  const int ldq = 3 * heads * hdp;
  const int nWx = p.W / ws, nW = (p.H / ws) * nWx;
  const int win = blockIdx.x / heads, h = blockIdx.x - win * heads;
  const int b = win / nW, widx = win - b * nW;
  const int wy = widx / nWx, wx = widx - wy * nWx;
  const float scale = rsqrtf((float)hd);
  const int tw = 2 * ws - 1, ntbl = tw * tw;

  for (int t = tid; t < 64 * NB; t += 256) {
    const int tc = min(t, N - 1);                                 // padding rows point at a real token and are masked
    const int py = tc / ws, px = tc - py * ws;
    const int r = wy * ws + py, c = wx * ws + px;
    int orr = r + p.shift; if (orr >= p.H) orr -= p.H;
    int occ = c + p.shift; if (occ >= p.W) occ -= p.W;
    tok[t] = (b * p.H + orr) * p.W + occ;
    const int rh = r < p.H - ws ? 0 : (r < p.H - p.shift ? 1 : 2);
    const int rw = c < p.W - ws ? 0 : (c < p.W - p.shift ? 1 : 2);
    inf[t] = ((rh * 3 + rw) << 16) | (py << 8) | px;
  }
  for (int t = tid; t < ntbl; t += 256) { tbl[t] = p.table[(size_t)t * heads + h]; dtb[t] = 0.f; }
  __syncthreads();

  // stage a 32-column chunk of 64 rows (block `blk` of the window's tokens) of q*scale / k / v / dO into its LDS tile
  auto stage_rows = [&](float* dst, int blk, int ch, int which /* 0 q, 1 k, 2 v */) {
    const int col0 = ch * AB_HC;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int idx = tid + 256 * i;
      const int row = idx >> 3, c = col0 + (idx & 7) * 4;
      const float* base = p.qkv + (size_t)tok[blk * 64 + row] * ldq + (which * heads + h) * hdp + min(c, hdp - 4);
      f32x4 v4 = *reinterpret_cast<const f32x4*>(base);
#pragma unroll
      for (int e = 0; e < 4; ++e) v4[e] = c + e < hd ? (which == 0 ? v4[e] * scale : v4[e]) : 0.f;
      *reinterpret_cast<f32x4*>(dst + row * AB_HS + (idx & 7) * 4) = v4;
    }
  };
  auto stage_g = [&](int blk, int ch) {
    const int col0 = ch * AB_HC;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int idx = tid + 256 * i;
      const int row = idx >> 5, cl = idx & 31, c = col0 + cl;
      const float g = p.dout[(size_t)tok[blk * 64 + row] * d + h * hd + min(c, hd - 1)];
      Gs[row * AB_HS + cl] = c < hd ? g : 0.f;
    }
  };

  const int nch = (hd + AB_HC - 1) / AB_HC;                        // <= 4 (head dims up to 128)
  for (int qb = 0; qb < NB; ++qb) {
    if (qb * 64 >= N) break;
    // ---- S = (q scale) k^T and dP = dO v^T for this query block against every key block ----
    f32x4 s[NB][4], dp[NB][4];
#pragma unroll
    for (int kb = 0; kb < NB; ++kb)
#pragma unroll
      for (int j = 0; j < 4; ++j) { s[kb][j] = f32x4{0.f, 0.f, 0.f, 0.f}; dp[kb][j] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    for (int ch = 0; ch < nch; ++ch) {
      __syncthreads();
      stage_rows(Qs, qb, ch, 0);
      stage_g(qb, ch);
#pragma unroll
      for (int kb = 0; kb < NB; ++kb) {
        if (kb > 0) __syncthreads();
        stage_rows(Ks, kb, ch, 1);
        stage_rows(Vs, kb, ch, 2);
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < AB_HC; kk += 16) {
          const f32x4 a = *reinterpret_cast<const f32x4*>(Qs + (wave * 16 + fr) * AB_HS + kk + 4 * fq);
          const f32x4 g = *reinterpret_cast<const f32x4*>(Gs + (wave * 16 + fr) * AB_HS + kk + 4 * fq);
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const f32x4 kbv = *reinterpret_cast<const f32x4*>(Ks + (j * 16 + fr) * AB_HS + kk + 4 * fq);
            const f32x4 vb = *reinterpret_cast<const f32x4*>(Vs + (j * 16 + fr) * AB_HS + kk + 4 * fq);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              s[kb][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[e], kbv[e], s[kb][j], 0, 0, 0);
              dp[kb][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(g[e], vb[e], dp[kb][j], 0, 0, 0);
            }
          }
        }
      }
    }
    // ---- softmax over the window's N keys (row = 64 qb + 16 wave + 4 fq + e, key = 64 kb + 16 j + fr), dS in place of dp ----
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int qi = inf[qb * 64 + wave * 16 + fq * 4 + e];
      const int qy = (qi >> 8) & 0xff, qx = qi & 0xff, qr = qi >> 16;
      float mx = -1e30f;
#pragma unroll
      for (int kb = 0; kb < NB; ++kb)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int key = kb * 64 + j * 16 + fr;
          const int ki = inf[key];
          const int kyy = (ki >> 8) & 0xff, kxx = ki & 0xff, kr = ki >> 16;
          float v = s[kb][j][e] + tbl[(qy - kyy + ws - 1) * tw + (qx - kxx + ws - 1)];
          if (p.shift > 0 && qr != kr) v += -100.0f;
          if (key >= N) v = -1e30f;
          s[kb][j][e] = v;
          mx = fmaxf(mx, v);
        }
      mx = srad_row16_max(mx);
      float sum = 0.f;
#pragma unroll
      for (int kb = 0; kb < NB; ++kb)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float pv = kb * 64 + j * 16 + fr < N ? expf(s[kb][j][e] - mx) : 0.f;
          s[kb][j][e] = pv;
          sum += pv;
        }
      sum = srad_row16_sum(sum);
      const float inv = 1.0f / sum;
      float dl = 0.f;
#pragma unroll
      for (int kb = 0; kb < NB; ++kb)
#pragma unroll
        for (int j = 0; j < 4; ++j) { s[kb][j][e] *= inv; dl += s[kb][j][e] * dp[kb][j][e]; }
      dl = srad_row16_sum(dl);
      const bool qreal = qb * 64 + wave * 16 + fq * 4 + e < N;      // a padding query contributes nothing to dk / dv / the table
#pragma unroll
      for (int kb = 0; kb < NB; ++kb)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          dp[kb][j][e] = qreal ? s[kb][j][e] * (dp[kb][j][e] - dl) : 0.f;
          if (!qreal) s[kb][j][e] = 0.f;
        }
    }
    // ---- key block after key block: P / dS tiles -> LDS, table gradient, dq (registers), dk / dv (read-add-write) ----
    f32x4 dq[4][2];
#pragma unroll
    for (int ch = 0; ch < 4; ++ch) { dq[ch][0] = f32x4{0.f, 0.f, 0.f, 0.f}; dq[ch][1] = dq[ch][0]; }
#pragma unroll
    for (int kb = 0; kb < NB; ++kb) {
      if (kb * 64 >= N) continue;
      __syncthreads();
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          Pm[(wave * 16 + fq * 4 + e) * AB_PS + j * 16 + fr] = s[kb][j][e];
          Dm[(wave * 16 + fq * 4 + e) * AB_PS + j * 16 + fr] = dp[kb][j][e];
        }
      __syncthreads();
      // table entry t = (dy + ws - 1) (2 ws - 1) + (dx + ws - 1): the dS of every (query, key = query - (dy, dx)) pair of this tile
      for (int t = tid; t < ntbl; t += 256) {
        const int dy = t / tw - (ws - 1), dx = t - (t / tw) * tw - (ws - 1);
        float acc = 0.f;
        for (int qy = max(0, dy); qy < min(ws, ws + dy); ++qy)
          for (int qx = max(0, dx); qx < min(ws, ws + dx); ++qx) {
            const int qn = qy * ws + qx - qb * 64, kn = (qy - dy) * ws + (qx - dx) - kb * 64;
            if (qn >= 0 && qn < 64 && kn >= 0 && kn < 64) acc += Dm[qn * AB_PS + kn];
          }
        dtb[t] += acc;                                              // (one thread per entry: no race)
      }
#pragma unroll
      for (int ch = 0; ch < 4; ++ch) {
        if (ch >= nch) continue;
        __syncthreads();
        stage_rows(Ks, kb, ch, 1);
        stage_rows(Qs, qb, ch, 0);
        stage_g(qb, ch);
        __syncthreads();
        f32x4 dk[2], dv[2];
#pragma unroll
        for (int jt = 0; jt < 2; ++jt) { dk[jt] = f32x4{0.f, 0.f, 0.f, 0.f}; dv[jt] = dk[jt]; }
#pragma unroll
        for (int kk = 0; kk < 64; kk += 16) {
          const f32x4 a = *reinterpret_cast<const f32x4*>(Dm + (wave * 16 + fr) * AB_PS + kk + 4 * fq);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int kr = kk + 4 * fq + e;
            const float at = Dm[kr * AB_PS + wave * 16 + fr];     // dS^T
            const float pt = Pm[kr * AB_PS + wave * 16 + fr];     // P^T
#pragma unroll
            for (int jt = 0; jt < 2; ++jt) {
              const float kbv = Ks[kr * AB_HS + jt * 16 + fr];
              const float qbv = Qs[kr * AB_HS + jt * 16 + fr];
              const float gb = Gs[kr * AB_HS + jt * 16 + fr];
              dq[ch][jt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[e], kbv, dq[ch][jt], 0, 0, 0);
              dk[jt] = __builtin_amdgcn_mfma_f32_16x16x4f32(at, qbv, dk[jt], 0, 0, 0);
              dv[jt] = __builtin_amdgcn_mfma_f32_16x16x4f32(pt, gb, dv[jt], 0, 0, 0);
            }
          }
        }
        // dk / dv rows of key block kb (row = 64 kb + 16 wave + 4 fq + e): first query block writes, later ones add
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int key = kb * 64 + wave * 16 + fq * 4 + e;
          if (key >= N) continue;
          float* dst = p.dqkv + (size_t)tok[key] * (3 * d) + h * hd;
#pragma unroll
          for (int jt = 0; jt < 2; ++jt) {
            const int c = ch * AB_HC + jt * 16 + fr;
            if (c < hd) {
              if (qb == 0) { dst[d + c] = dk[jt][e]; dst[2 * d + c] = dv[jt][e]; }
              else { dst[d + c] += dk[jt][e]; dst[2 * d + c] += dv[jt][e]; }
            }
          }
        }
      }
    }
    // dq rows of this query block
#pragma unroll
    for (int ch = 0; ch < 4; ++ch) {
      if (ch >= nch) continue;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int qrow = qb * 64 + wave * 16 + fq * 4 + e;
        if (qrow >= N) continue;
        float* dst = p.dqkv + (size_t)tok[qrow] * (3 * d) + h * hd;
#pragma unroll
        for (int jt = 0; jt < 2; ++jt) {
          const int c = ch * AB_HC + jt * 16 + fr;
          if (c < hd) dst[c] = dq[ch][jt][e] * scale;
        }
      }
    }
    __threadfence_block();
  }
  __syncthreads();
  for (int t = tid; t < ntbl; t += 256) tpart[(size_t)win * (ntbl * heads) + (size_t)t * heads + h] = dtb[t];
}
constexpr size_t ABG_LDS(int nb) { return (size_t)(4 * 64 * AB_HS + 2 * 64 * AB_PS + 2 * 1024) * sizeof(float) + 2 * 64 * nb * sizeof(int); }

// ------------------------------------------------------------------------------------------
// The same backward with bf16 MFMA operands (v_mfma_f32_16x16x32_bf16, fp32 accumulation) for the bf16 precision
// mode: q, k, v, dO chunks are staged as bf16; P and dS leave the softmax as bf16 tiles - dS in both orientations
// (row-major for dq = dS k, transposed for dk = dS^T q), P transposed (dv = P^T dO) - and the second operand of
// those three products (k, q, dO: contraction over tokens, their slow axis in LDS) comes through the transposing
// LDS read ds_read_b64_tr_b16.  Softmax statistics, dS and the bias-table gradient stay fp32.
// ------------------------------------------------------------------------------------------
constexpr int AH_HS = 32 + 8, AH_PS = 64 + 8;
constexpr size_t AH_LDS = (size_t)(4 * 64 * AH_HS + 3 * 64 * AH_PS) * sizeof(__bf16) + (size_t)(256 + 64 * 68) * sizeof(float) + 2 * 64 * sizeof(int);

__global__ __launch_bounds__(256) void window_attn_bwd_bf16_kernel(const AttnBwdParams p, float* __restrict__ tpart) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  __bf16* Qs = reinterpret_cast<__bf16*>(smem);          // [64][AH_HS] q * scale chunk
  __bf16* Ks = Qs + 64 * AH_HS;
  __bf16* Vs = Ks + 64 * AH_HS;
  __bf16* Gs = Vs + 64 * AH_HS;                          // dO chunk
  __bf16* Dm = Gs + 64 * AH_HS;                          // [query][key] dS
  __bf16* DmT = Dm + 64 * AH_PS;                         // [key][query] dS
  __bf16* PmT = DmT + 64 * AH_PS;                        // [key][query] P
  float* tbl = reinterpret_cast<float*>(PmT + 64 * AH_PS);   // [225] (256 reserved)
  float* Df = tbl + 256;                                 // [64][68] fp32 dS for the bias-table gradient
  int* tok = reinterpret_cast<int*>(Df + 64 * 68);
  int* inf = tok + 64;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 15, fq = lane >> 4;
  const int ws = 8, d = p.d, heads = p.heads, hd = d / heads, hdp = p.hdp;
  const int ldq = 3 * heads * hdp;
  const int nWx = p.W / ws, nW = (p.H / ws) * nWx;
  // workgroups are dealt round-robin over the 8 XCDs: keep all heads of a window on ONE XCD (its L2 then holds the
  // window's dO / dqkv lines, which the heads share at 4-byte granularity, once)
  int win, h;
  {
    const int L = blockIdx.x, nwin = p.B * nW;
    if ((nwin & 7) == 0) { const int slot = L >> 3; win = (slot / heads) * 8 + (L & 7); h = slot - (slot / heads) * heads; }
    else { win = L / heads; h = L - win * heads; }
  }
  const int b = win / nW, widx = win - b * nW;
  const int wy = widx / nWx, wx = widx - wy * nWx;
  const float scale = rsqrtf((float)hd);
  const int tw = 2 * ws - 1;

  if (tid < 64) {
    const int py = tid / ws, px = tid - py * ws;
    const int r = wy * ws + py, c = wx * ws + px;
    int orr = r + p.shift; if (orr >= p.H) orr -= p.H;
    int occ = c + p.shift; if (occ >= p.W) occ -= p.W;
    tok[tid] = (b * p.H + orr) * p.W + occ;
    const int rh = r < p.H - ws ? 0 : (r < p.H - p.shift ? 1 : 2);
    const int rw = c < p.W - ws ? 0 : (c < p.W - p.shift ? 1 : 2);
    inf[tid] = ((rh * 3 + rw) << 16) | (py << 8) | px;
  }
  if (tid < tw * tw) tbl[tid] = p.table[(size_t)tid * heads + h];
  __syncthreads();

  auto stage = [&](int ch, bool need_v) {
    const int col0 = ch * 32;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int idx = tid + 256 * i;
      const int row = idx >> 3, cl = (idx & 7) * 4, c = col0 + cl;
      const float* base = p.qkv + (size_t)tok[row] * ldq + h * hdp + min(c, hdp - 4);
      const f32x4 q4 = *reinterpret_cast<const f32x4*>(base);
      const f32x4 k4 = *reinterpret_cast<const f32x4*>(base + heads * hdp);
      const f32x4 v4 = need_v ? *reinterpret_cast<const f32x4*>(base + 2 * heads * hdp) : f32x4{0.f, 0.f, 0.f, 0.f};
      bf16x4 qh, kh, vh;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const bool ok = c + e < hd;
        qh[e] = (__bf16)(ok ? q4[e] * scale : 0.f);
        kh[e] = (__bf16)(ok ? k4[e] : 0.f);
        vh[e] = (__bf16)(ok ? v4[e] : 0.f);
      }
      *reinterpret_cast<bf16x4*>(Qs + row * AH_HS + cl) = qh;
      *reinterpret_cast<bf16x4*>(Ks + row * AH_HS + cl) = kh;
      if (need_v) *reinterpret_cast<bf16x4*>(Vs + row * AH_HS + cl) = vh;
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int idx = tid + 256 * i;
      const int row = idx >> 5, cl = idx & 31, c = col0 + cl;
      const float g = p.dout[(size_t)tok[row] * d + h * hd + min(c, hd - 1)];
      Gs[row * AH_HS + cl] = (__bf16)(c < hd ? g : 0.f);
    }
  };

  const int nch = (hd + 31) / 32;
  f32x4 s[4], dp[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) { s[j] = f32x4{0.f, 0.f, 0.f, 0.f}; dp[j] = f32x4{0.f, 0.f, 0.f, 0.f}; }
  for (int ch = 0; ch < nch; ++ch) {
    if (ch > 0) __syncthreads();
    stage(ch, true);
    __syncthreads();
    const bf16x8 a = *reinterpret_cast<const bf16x8*>(Qs + (wave * 16 + fr) * AH_HS + 8 * fq);
    const bf16x8 g = *reinterpret_cast<const bf16x8*>(Gs + (wave * 16 + fr) * AH_HS + 8 * fq);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const bf16x8 kb = *reinterpret_cast<const bf16x8*>(Ks + (j * 16 + fr) * AH_HS + 8 * fq);
      const bf16x8 vb = *reinterpret_cast<const bf16x8*>(Vs + (j * 16 + fr) * AH_HS + 8 * fq);
      s[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, kb, s[j], 0, 0, 0);
      dp[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(g, vb, dp[j], 0, 0, 0);
    }
  }

  // ---- softmax (row = 16 wave + 4 fq + e, key = 16 j + fr), dS ----
  {
    int kinf[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) kinf[j] = inf[j * 16 + fr];
    f32x4 pr[4], dsr[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int row = wave * 16 + fq * 4 + e;
      const int qi = inf[row];
      const int qy = (qi >> 8) & 0xff, qx = qi & 0xff, qr = qi >> 16;
      float mx = -1e30f;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int kyy = (kinf[j] >> 8) & 0xff, kxx = kinf[j] & 0xff, kr = kinf[j] >> 16;
        float v = s[j][e] + tbl[(qy - kyy + ws - 1) * tw + (qx - kxx + ws - 1)];
        if (p.shift > 0 && qr != kr) v += -100.0f;
        s[j][e] = v;
        mx = fmaxf(mx, v);
      }
      mx = srad_row16_max(mx);
      float sum = 0.f;
#pragma unroll
      for (int j = 0; j < 4; ++j) { s[j][e] = __expf(s[j][e] - mx); sum += s[j][e]; }
      sum = srad_row16_sum(sum);
      const float inv = 1.0f / sum;
      float dl = 0.f;
#pragma unroll
      for (int j = 0; j < 4; ++j) { s[j][e] *= inv; dl += s[j][e] * dp[j][e]; }
      dl = srad_row16_sum(dl);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float ds = s[j][e] * (dp[j][e] - dl);
        pr[j][e] = s[j][e]; dsr[j][e] = ds;
        Dm[row * AH_PS + j * 16 + fr] = (__bf16)ds;
        Df[row * 68 + j * 16 + fr] = ds;
      }
    }
    // transposed tiles: this lane's four rows 4 fq .. 4 fq + 3 of query slab `wave` are consecutive along k
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      bf16x4 ph, dh;
#pragma unroll
      for (int e = 0; e < 4; ++e) { ph[e] = (__bf16)pr[j][e]; dh[e] = (__bf16)dsr[j][e]; }
      *reinterpret_cast<bf16x4*>(PmT + (j * 16 + fr) * AH_PS + wave * 16 + 4 * fq) = ph;
      *reinterpret_cast<bf16x4*>(DmT + (j * 16 + fr) * AH_PS + wave * 16 + 4 * fq) = dh;
    }
  }
  __syncthreads();
  if (tid < tw * tw) {                                   // bias-table gradient row of this (window, head), fp32
    const int dy = tid / tw - (ws - 1), dx = tid - (tid / tw) * tw - (ws - 1);
    float acc = 0.f;
    for (int qy = max(0, dy); qy < min(ws, ws + dy); ++qy)
      for (int qx = max(0, dx); qx < min(ws, ws + dx); ++qx)
        acc += Df[(qy * ws + qx) * 68 + (qy - dy) * ws + (qx - dx)];
    tpart[(size_t)win * (tw * tw * heads) + (size_t)tid * heads + h] = acc;
  }

  // ---- pass B: dq, dk, dv per 32-column chunk ----
  typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;
  const int tq = fr >> 2, tp = fr & 3;
  for (int ch = 0; ch < nch; ++ch) {
    if (ch > 0 || nch > 1) { __syncthreads(); stage(ch, false); __syncthreads(); }
    f32x4 dq[2], dk[2], dv[2];
#pragma unroll
    for (int jt = 0; jt < 2; ++jt) { dq[jt] = f32x4{0.f, 0.f, 0.f, 0.f}; dk[jt] = dq[jt]; dv[jt] = dq[jt]; }
#pragma unroll
    for (int kk = 0; kk < 64; kk += 32) {
      const bf16x8 a_ds = *reinterpret_cast<const bf16x8*>(Dm + (wave * 16 + fr) * AH_PS + kk + 8 * fq);
      const bf16x8 a_dst = *reinterpret_cast<const bf16x8*>(DmT + (wave * 16 + fr) * AH_PS + kk + 8 * fq);
      const bf16x8 a_pt = *reinterpret_cast<const bf16x8*>(PmT + (wave * 16 + fr) * AH_PS + kk + 8 * fq);
#pragma unroll
      for (int jt = 0; jt < 2; ++jt) {
        // B[k = token kk + 8 fq + t][j = column 16 jt + fr] of the row-major chunk tiles: two transposing reads each
        auto tr8 = [&](const __bf16* tile) -> bf16x8 {
          const __bf16* r0 = tile + (kk + 8 * fq + tq) * AH_HS + jt * 16 + 4 * tp;
          const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(r0));
          const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(r0 + 4 * AH_HS));
          bf16x8 o;
#pragma unroll
          for (int e = 0; e < 4; ++e) { o[e] = lo[e]; o[4 + e] = hi[e]; }
          return o;
        };
        dq[jt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_ds, tr8(Ks), dq[jt], 0, 0, 0);
        dk[jt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_dst, tr8(Qs), dk[jt], 0, 0, 0);
        dv[jt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_pt, tr8(Gs), dv[jt], 0, 0, 0);
      }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float* dst = p.dqkv + (size_t)tok[wave * 16 + fq * 4 + e] * (3 * d) + h * hd;
#pragma unroll
      for (int jt = 0; jt < 2; ++jt) {
        const int c = ch * 32 + jt * 16 + fr;
        if (c < hd) {
          if (p.dqkv_h) {
            __bf16* dh_ = p.dqkv_h + (size_t)tok[wave * 16 + fq * 4 + e] * (3 * d) + h * hd;
            dh_[c] = (__bf16)(dq[jt][e] * scale);
            dh_[d + c] = (__bf16)dk[jt][e];
            dh_[2 * d + c] = (__bf16)dv[jt][e];
          } else {
            dst[c] = dq[jt][e] * scale;
            dst[d + c] = dk[jt][e];
            dst[2 * d + c] = dv[jt][e];
          }
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------
// All-bf16 form of the same backward, the training step's case: q (already scaled and rounded, exactly the operand the
// forward's MFMA took) | k | v and dO arrive as bf16 in per-head slots of hp columns, so a thread stages its share of a
// 32-column chunk with four 16-byte loads issued before anything else (it derives its row's token itself), and dq | dk | dv
// leave as bf16.  NCH = 32-column chunks of the head dim (1 .. 4).  With ONE chunk (head dim <= 32, DRCT-L's 30) the second
// operands of the three pass-B products are fetched (transposing LDS reads) right after pass A and the staging tiles are
// dead from then on: the fp32 dS copy the bias-table gradient sums is laid over them.  That brings the workgroup to 48.5 KB
// of LDS - three per CU, which for the 768 (window, head) pairs of the 8-image training batch is ONE resident round on
// 256 CUs instead of one and a half.  With more chunks pass B stages them again and the dS copy has its own 17 KB.
// ------------------------------------------------------------------------------------------
template <int NCH> constexpr size_t ag_lds_bytes() {
  return (size_t)(4 * 64 * AH_HS + 3 * 64 * AH_PS) * sizeof(__bf16) + 256 * sizeof(float) + 2 * 64 * sizeof(int) +
         (NCH > 1 ? (size_t)64 * 68 * sizeof(float) : 0);
}
static_assert((size_t)64 * 68 * sizeof(float) <= (size_t)4 * 64 * AH_HS * sizeof(__bf16), "the fp32 dS copy must fit in the staging tiles");

template <int NCH>
__global__ __launch_bounds__(256) void window_attn_bwd_h_kernel(const AttnBwdParams p, float* __restrict__ tpart) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  __bf16* Qs = reinterpret_cast<__bf16*>(smem);          // [64][AH_HS] q * scale (32-column chunk)
  __bf16* Ks = Qs + 64 * AH_HS;
  __bf16* Vs = Ks + 64 * AH_HS;
  __bf16* Gs = Vs + 64 * AH_HS;                          // dO
  __bf16* Dm = Gs + 64 * AH_HS;                          // [query][key] dS
  __bf16* DmT = Dm + 64 * AH_PS;                         // [key][query] dS
  __bf16* PmT = DmT + 64 * AH_PS;                        // [key][query] P
  float* tbl = reinterpret_cast<float*>(PmT + 64 * AH_PS);   // [225] (256 reserved)
  int* tok = reinterpret_cast<int*>(tbl + 256);
  int* inf = tok + 64;
  // [64][68] fp32 dS: over Qs .. Gs once pass A is done with them (one chunk), else behind everything
  float* Df = NCH == 1 ? reinterpret_cast<float*>(smem) : reinterpret_cast<float*>(inf + 64);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 15, fq = lane >> 4;
  const int ws = 8, d = p.d, heads = p.heads, hd = d / heads, hp = p.hp_h;
  const int nWx = p.W / ws, nW = (p.H / ws) * nWx;
  int win, h;                                            // all heads of a window on one XCD (see the kernel above)
  {
    const int L = blockIdx.x, nwin = p.B * nW;
    if ((nwin & 7) == 0 && !p.no_xcd_map) { const int slot = L >> 3; win = (L & 7) * (nwin >> 3) + slot / heads; h = slot - (slot / heads) * heads; }   // XCD k: window strip k (affinity with the row-tile kernels)
    else if ((nwin & 7) == 0) { const int slot = L >> 3; win = (slot / heads) * 8 + (L & 7); h = slot - (slot / heads) * heads; }
    else { win = L / heads; h = L - win * heads; }
  }
  const int b = win / nW, widx = win - b * nW;
  const int wy = widx / nWx, wx = widx - wy * nWx;
  const float scale = rsqrtf((float)hd);
  const int tw = 2 * ws - 1;

  auto geometry = [&](int row, int& token, int& info) {
    const int py = row >> 3, px = row & 7;
    const int r = wy * ws + py, c = wx * ws + px;
    int orr = r + p.shift; if (orr >= p.H) orr -= p.H;
    int occ = c + p.shift; if (occ >= p.W) occ -= p.W;
    token = (b * p.H + orr) * p.W + occ;
    const int rh = r < p.H - ws ? 0 : (r < p.H - p.shift ? 1 : 2);
    const int rw = c < p.W - ws ? 0 : (c < p.W - p.shift ? 1 : 2);
    info = ((rh * 3 + rw) << 16) | (py << 8) | px;
  };

  // ---- staging: thread = (row tid / 4, 8 columns of the chunk), everything in flight before the first LDS store ----
  const int srow = tid >> 2, scl = (tid & 3) * 8;
  int stok, sinf;
  geometry(srow, stok, sinf);
  const __bf16* const qrow = p.qkv_h + (size_t)stok * (3 * heads * hp) + h * hp;
  const __bf16* const grow = p.dout_h + (size_t)stok * (heads * hp) + h * hp;
  u32x4 sq, sk, sv, sg;
  auto load_chunk = [&](int ch, bool need_v) __attribute__((always_inline)) {
    const int off = min(ch * 32 + scl, hp - 8);
    sq = *reinterpret_cast<const u32x4*>(qrow + off);
    sk = *reinterpret_cast<const u32x4*>(qrow + heads * hp + off);
    if (need_v) sv = *reinterpret_cast<const u32x4*>(qrow + 2 * heads * hp + off);
    sg = *reinterpret_cast<const u32x4*>(grow + off);
  };
  auto store_chunk = [&](int ch, bool need_v) __attribute__((always_inline)) {
    auto put = [&](__bf16* tile, const u32x4& raw) {        // columns at or beyond the head dim are zero in LDS
      bf16x8 v = __builtin_bit_cast(bf16x8, raw);
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = ch * 32 + scl + e < hd ? v[e] : (__bf16)0.f;
      *reinterpret_cast<bf16x8*>(tile + srow * AH_HS + scl) = v;
    };
    put(Qs, sq); put(Ks, sk); if (need_v) put(Vs, sv); put(Gs, sg);
  };
  load_chunk(0, true);
  {
    const float tv = p.table[(size_t)min(tid, tw * tw - 1) * heads + h];
    if ((tid & 3) == 0) { tok[srow] = stok; inf[srow] = sinf; }
    tbl[tid] = tv;
  }
  store_chunk(0, true);
  __syncthreads();

  // ---- pass A: S = q k^T, dP = dO v^T (row = 16 wave + 4 fq + e, key = 16 j + fr) ----
  f32x4 s[4], dp[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) { s[j] = f32x4{0.f, 0.f, 0.f, 0.f}; dp[j] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
  for (int ch = 0; ch < NCH; ++ch) {
    if (ch + 1 < NCH) load_chunk(ch + 1, true);             // the next chunk's rows are in flight over this chunk's MFMAs
    const bf16x8 a = *reinterpret_cast<const bf16x8*>(Qs + (wave * 16 + fr) * AH_HS + 8 * fq);
    const bf16x8 g = *reinterpret_cast<const bf16x8*>(Gs + (wave * 16 + fr) * AH_HS + 8 * fq);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const bf16x8 kb = *reinterpret_cast<const bf16x8*>(Ks + (j * 16 + fr) * AH_HS + 8 * fq);
      const bf16x8 vb = *reinterpret_cast<const bf16x8*>(Vs + (j * 16 + fr) * AH_HS + 8 * fq);
      s[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, kb, s[j], 0, 0, 0);
      dp[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(g, vb, dp[j], 0, 0, 0);
    }
    if (ch + 1 < NCH) { __syncthreads(); store_chunk(ch + 1, true); __syncthreads(); }
  }
  // second operands of pass B: B[k = token kk + 8 fq + t][j = column 16 jt + fr] of the row-major tiles
  typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;
  const int tq = fr >> 2, tp = fr & 3;
  auto tr8 = [&](const __bf16* tile, int kk, int jt) __attribute__((always_inline)) -> bf16x8 {
    const __bf16* r0 = tile + (kk + 8 * fq + tq) * AH_HS + jt * 16 + 4 * tp;
    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(r0));
    const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(r0 + 4 * AH_HS));
    bf16x8 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) { o[e] = lo[e]; o[4 + e] = hi[e]; }
    return o;
  };
  bf16x8 bK[2][2], bQ[2][2], bG[2][2];
  if constexpr (NCH == 1) {
#pragma unroll
    for (int k2 = 0; k2 < 2; ++k2)
#pragma unroll
      for (int jt = 0; jt < 2; ++jt) { bK[k2][jt] = tr8(Ks, 32 * k2, jt); bQ[k2][jt] = tr8(Qs, 32 * k2, jt); bG[k2][jt] = tr8(Gs, 32 * k2, jt); }
  } else {
    load_chunk(0, false);                                   // pass B walks the chunks again (k, q, dO): chunk 0 in flight over the softmax
  }

  // ---- softmax and dS in registers (their LDS reads are tbl / inf only) ----
  f32x4 pr[4], dsr[4];
  {
    int kinf[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) kinf[j] = inf[j * 16 + fr];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int row = wave * 16 + fq * 4 + e;
      const int qi = inf[row];
      const int qy = (qi >> 8) & 0xff, qx = qi & 0xff, qr = qi >> 16;
      float mx = -1e30f;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int kyy = (kinf[j] >> 8) & 0xff, kxx = kinf[j] & 0xff, kr = kinf[j] >> 16;
        float v = s[j][e] + tbl[(qy - kyy + ws - 1) * tw + (qx - kxx + ws - 1)];
        if (p.shift > 0 && qr != kr) v += -100.0f;
        s[j][e] = v;
        mx = fmaxf(mx, v);
      }
      mx = srad_row16_max(mx);
      float sum = 0.f;
#pragma unroll
      for (int j = 0; j < 4; ++j) { s[j][e] = __expf(s[j][e] - mx); sum += s[j][e]; }
      sum = srad_row16_sum(sum);
      const float inv = 1.0f / sum;
      float dl = 0.f;
#pragma unroll
      for (int j = 0; j < 4; ++j) { s[j][e] *= inv; dl += s[j][e] * dp[j][e]; }
      dl = srad_row16_sum(dl);
#pragma unroll
      for (int j = 0; j < 4; ++j) { pr[j][e] = s[j][e]; dsr[j][e] = s[j][e] * (dp[j][e] - dl); }
    }
  }
  __syncthreads();                                       // every wave has read the staging tiles: Df (one chunk) / chunk 0 may overwrite them
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int row = wave * 16 + fq * 4 + e;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      Dm[row * AH_PS + j * 16 + fr] = (__bf16)dsr[j][e];
      Df[row * 68 + j * 16 + fr] = dsr[j][e];
    }
  }
  // transposed tiles: this lane's four rows 4 fq .. 4 fq + 3 of query slab `wave` are consecutive along k
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    bf16x4 ph, dh;
#pragma unroll
    for (int e = 0; e < 4; ++e) { ph[e] = (__bf16)pr[j][e]; dh[e] = (__bf16)dsr[j][e]; }
    *reinterpret_cast<bf16x4*>(PmT + (j * 16 + fr) * AH_PS + wave * 16 + 4 * fq) = ph;
    *reinterpret_cast<bf16x4*>(DmT + (j * 16 + fr) * AH_PS + wave * 16 + 4 * fq) = dh;
  }
  if constexpr (NCH > 1) store_chunk(0, false);
  __syncthreads();

  // ---- pass B: dq = dS k, dk = dS^T q, dv = P^T dO, 32 columns at a time ----
#pragma unroll
  for (int ch = 0; ch < NCH; ++ch) {
    if constexpr (NCH > 1) {
      if (ch + 1 < NCH) load_chunk(ch + 1, false);
#pragma unroll
      for (int k2 = 0; k2 < 2; ++k2)
#pragma unroll
        for (int jt = 0; jt < 2; ++jt) { bK[k2][jt] = tr8(Ks, 32 * k2, jt); bQ[k2][jt] = tr8(Qs, 32 * k2, jt); bG[k2][jt] = tr8(Gs, 32 * k2, jt); }
    }
    f32x4 dq[2], dk[2], dv[2];
#pragma unroll
    for (int jt = 0; jt < 2; ++jt) { dq[jt] = f32x4{0.f, 0.f, 0.f, 0.f}; dk[jt] = dq[jt]; dv[jt] = dq[jt]; }
#pragma unroll
    for (int k2 = 0; k2 < 2; ++k2) {
      const bf16x8 a_ds = *reinterpret_cast<const bf16x8*>(Dm + (wave * 16 + fr) * AH_PS + 32 * k2 + 8 * fq);
      const bf16x8 a_dst = *reinterpret_cast<const bf16x8*>(DmT + (wave * 16 + fr) * AH_PS + 32 * k2 + 8 * fq);
      const bf16x8 a_pt = *reinterpret_cast<const bf16x8*>(PmT + (wave * 16 + fr) * AH_PS + 32 * k2 + 8 * fq);
#pragma unroll
      for (int jt = 0; jt < 2; ++jt) {
        dq[jt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_ds, bK[k2][jt], dq[jt], 0, 0, 0);
        dk[jt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_dst, bQ[k2][jt], dk[jt], 0, 0, 0);
        dv[jt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_pt, bG[k2][jt], dv[jt], 0, 0, 0);
      }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      __bf16* dst = p.dqkv_h + (size_t)tok[wave * 16 + fq * 4 + e] * (3 * d) + h * hd;
#pragma unroll
      for (int jt = 0; jt < 2; ++jt) {
        const int c = ch * 32 + jt * 16 + fr;
        if (c < hd) {
          dst[c] = (__bf16)(dq[jt][e] * scale);
          dst[d + c] = (__bf16)dk[jt][e];
          dst[2 * d + c] = (__bf16)dv[jt][e];
        }
      }
    }
    if constexpr (NCH > 1) {
      if (ch + 1 < NCH) { __syncthreads(); store_chunk(ch + 1, false); __syncthreads(); }
    }
  }
  if (tid < tw * tw) {                                   // bias-table gradient row of this (window, head), fp32
    const int dy = tid / tw - (ws - 1), dx = tid - (tid / tw) * tw - (ws - 1);
    float acc = 0.f;
    for (int qy = max(0, dy); qy < min(ws, ws + dy); ++qy)
      for (int qx = max(0, dx); qx < min(ws, ws + dx); ++qx)
        acc += Df[(qy * ws + qx) * 68 + (qy - dy) * ws + (qx - dx)];
    tpart[(size_t)win * (tw * tw * heads) + (size_t)tid * heads + h] = acc;
  }
}

// ------------------------------------------------------------------------------------------ elementwise
__global__ void dact_kernel(const float* __restrict__ dy, int ld_dy, const float* __restrict__ y, int ld_y,
                            float* __restrict__ out, int ld_out, int rows, int C, float slope) {
  const size_t total = (size_t)rows * C;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const size_t m = i / C;
    const int c = (int)(i - m * C);
    out[m * ld_out + c] = dy[m * ld_dy + c] * (y[m * ld_y + c] > 0.f ? 1.f : slope);
  }
}

__global__ void unshuffle_kernel(const float* __restrict__ src, float* __restrict__ dst, int B, int H, int W, int F) {
  const size_t total = (size_t)B * H * W * 4 * F;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    // iterate in SOURCE order (coalesced reads): i = ((b*2H + y2)*2W + x2)*F + c
    const int c = (int)(i % F);
    size_t pix = i / F;
    const int x2 = (int)(pix % (2 * W)); pix /= 2 * W;
    const int y2 = (int)(pix % (2 * H));
    const int b = (int)(pix / (2 * H));
    const int n = c * 4 + (y2 & 1) * 2 + (x2 & 1);
    dst[(((size_t)b * H + (y2 >> 1)) * W + (x2 >> 1)) * 4 * F + n] = src[i];
  }
}

__global__ void copy_cols_kernel(const float* __restrict__ src, int ld_src, float* __restrict__ dst, int ld_dst, int rows, int C) {
  const size_t total = (size_t)rows * ld_dst;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const size_t m = i / ld_dst;
    const int c = (int)(i - m * ld_dst);
    dst[i] = c < C ? src[m * ld_src + c] : 0.f;
  }
}

__global__ void l1_grad_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out, size_t n, float scale) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const float d = a[i] - b[i];
    out[i] = d > 0.f ? scale : (d < 0.f ? -scale : 0.f);
  }
}

// torch.optim.Adam single-tensor arithmetic (torch/optim/adam.py _single_tensor_adam, amsgrad off, maximize off)
__device__ __forceinline__ void adam_body(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                          float* __restrict__ v, size_t n, float lr, float b1, float b2, float eps, float wd,
                                          float bc1, float bc2_sqrt, float gs) {
  const float step_size = lr / bc1;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    float grad = g[i] * gs;
    const float w = p[i];
    if (wd != 0.f) grad = grad + wd * w;
    const float mi = m[i] + (grad - m[i]) * (1.f - b1);          // exp_avg.lerp_(grad, 1 - beta1)
    const float vi = v[i] * b2 + (1.f - b2) * grad * grad;       // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, 1 - beta2)
    m[i] = mi; v[i] = vi;
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    p[i] = w - step_size * (mi / denom);
  }
}
__global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                            size_t n, float lr, float b1, float b2, float eps, float wd, float bc1, float bc2_sqrt, float gs) {
  adam_body(p, g, m, v, n, lr, b1, b2, eps, wd, bc1, bc2_sqrt, gs);
}
// the same step with the per-step scalars [lr, 1 - beta1^t, sqrt(1 - beta2^t), grad_scale] read from device memory, so
// that a captured hipGraph of a whole training step can be replayed with a new step count / learning rate
__global__ void adam_dev_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                                size_t n, float b1, float b2, float eps, float wd, const float* __restrict__ hyper) {
  adam_body(p, g, m, v, n, hyper[0], b1, b2, eps, wd, hyper[1], hyper[2], hyper[3]);
}

}  // namespace

int srad_wgrad_flush(WgradQueue& q, hipStream_t stream) {
  if (q.batch.count > 0) {
    SradProfScope prof(stream, SRAD_K_WGRAD_REDUCE, 0.0, 4.0 * q.used);
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(4 * q.tiles), dim3(256), 0, stream, q.batch);
    SRAD_CHECK_HIP(hipGetLastError());
  }
  q.batch.count = 0; q.tiles = 0; q.used = 0;
  return SRAD_OK;
}

static int check_wgrad(const WgradParams& p);

int srad_launch_wgrad_deferred(int prec, const WgradParams& p, WgradQueue& q, hipStream_t stream) {
  SRAD_TRY(check_wgrad(p));
  return prec == SRAD_PREC_BF16 ? defer_wgrad<SRAD_PREC_BF16>(p, q, stream) : defer_wgrad<SRAD_PREC_F32>(p, q, stream);
}

int srad_wgrad_launch_deferred(int prec, WgradQueue& q, hipStream_t stream) {
  WgradMulti& m = q.multi;
  if (m.count == 0) return SRAD_OK;
  const int total = m.blk0[m.count - 1] + m.nblk[m.count - 1];
  {
    SradProfScope prof(stream, SRAD_K_WGRAD, q.multi_flops, q.multi_bytes);
    auto launch = [&](auto kern) -> int {
      static SradOncePerDevice configured;
      if (configured.need()) {
        SRAD_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)WG_LDS));
        configured.done();
      }
      hipLaunchKernelGGL(kern, dim3(total), dim3(256), WG_LDS, stream, m);
      return SRAD_OK;
    };
    bool full = prec == SRAD_PREC_BF16;            // every layer: whole 128-row steps, DropPath factor constant per wave step
    for (int i = 0; i < m.count && full; ++i) {
      const long rows_per = ((m.p[i].M + m.ksplit[i] - 1) / m.ksplit[i] + 127) / 128 * 128;
      full = rows_per * m.ksplit[i] == m.p[i].M && (!m.p[i].row_scale || m.p[i].rps % 32 == 0);
    }
    for (int i = 0; i < m.count; ++i)
      SRAD_REQUIRE(!(m.p[i].x_bf16 || m.p[i].dy_bf16) || (prec == SRAD_PREC_BF16 && (!m.p[i].dy_bf16 || m.p[i].x_bf16)),
                   "wgrad: bf16 operand storage needs the bf16 MFMA path, dY only together with X");
    bool all_hh = full;                                // every layer with both operands as bf16: the lean kernel
    for (int i = 0; i < m.count && all_hh; ++i) all_hh = m.p[i].x_bf16 && m.p[i].dy_bf16;
    static const bool no_hh = getenv("SRAD_WGRAD_NO_HH") != nullptr;
    if (all_hh && !no_hh) {
      static SradOncePerDevice configured_hh;
      if (configured_hh.need()) {
        SRAD_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_multi_hh_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)WG_LDS2));
        configured_hh.done();
      }
      hipLaunchKernelGGL(wgrad_multi_hh_kernel, dim3(total), dim3(256), WG_LDS2, stream, m);
    } else {
      const int rc = prec != SRAD_PREC_BF16 ? launch(wgrad_multi_kernel<SRAD_PREC_F32, false>)
                     : full                 ? launch(wgrad_multi_kernel<SRAD_PREC_BF16, true>)
                                            : launch(wgrad_multi_kernel<SRAD_PREC_BF16, false>);
      if (rc) return rc;
    }
    SRAD_CHECK_HIP(hipGetLastError());
  }
  m.count = 0; q.multi_flops = 0; q.multi_bytes = 0;
  return SRAD_OK;
}

static int check_wgrad(const WgradParams& p) {
  SRAD_REQUIRE(p.M > 0 && p.N > 0 && p.Cin > 0 && p.dW, "wgrad: empty problem M=%d N=%d Cin=%d", p.M, p.N, p.Cin);
  SRAD_REQUIRE((p.N & 3) == 0 && (p.Cin & 3) == 0 && (p.ldy & 3) == 0 && (p.ldx & 3) == 0 && (p.ycol0 & 3) == 0 &&
                   ((uintptr_t)p.dY & 15) == 0 && ((uintptr_t)p.X & 15) == 0,
               "wgrad: operands need channel counts / strides that are multiples of 4 floats (N=%d Cin=%d ldy=%d ldx=%d)", p.N, p.Cin, p.ldy, p.ldx);
  SRAD_REQUIRE(p.n_real > 0 && p.n_real <= p.N && p.cin_real > 0 && p.cin_real <= p.Cin, "wgrad: bad real extents");
  SRAD_REQUIRE(p.grp_pad == 0 || (p.grp_real > 0 && p.grp_real <= p.grp_pad && p.cin_real % p.grp_real == 0 &&
                                  (p.cin_real / p.grp_real) * p.grp_pad <= p.Cin), "wgrad: bad channel groups %d -> %d", p.grp_real, p.grp_pad);
  SRAD_REQUIRE(p.ntaps == 1 || p.ntaps == 9, "wgrad: ntaps must be 1 or 9");
  if (p.ntaps == 9 || p.stride != 1)
    SRAD_REQUIRE(p.Ho > 0 && p.Wo > 0 && p.Hi > 0 && p.Wi > 0 && p.M % (p.Ho * p.Wo) == 0, "wgrad: bad conv geometry");
  SRAD_REQUIRE(!p.row_scale || p.rps > 0, "wgrad: row_scale needs rows-per-sample");
  return SRAD_OK;
}

bool srad_wgrad_conv9_supported(const WgradParams& p) { return conv9_supported(p); }

int srad_launch_wgrad(int prec, const WgradParams& p, WgradQueue& q, hipStream_t stream) {
  SRAD_TRY(check_wgrad(p));
  SRAD_REQUIRE((!p.x_bf16 && !p.dy_bf16) || (prec == SRAD_PREC_BF16 && conv9_supported(p)),
               "wgrad: bf16 operand storage is for deferred Linear layers and the nine-tap 80-channel convolution kernel only");
  return prec == SRAD_PREC_BF16 ? launch_wgrad<SRAD_PREC_BF16>(p, q, stream) : launch_wgrad<SRAD_PREC_F32>(p, q, stream);
}

static int queue_colsum(WgradQueue& q, float* dst, const float* part, int ncols, int row_stride, int rows, float alpha,
                        hipStream_t stream) {
  if (q.batch.count == SRAD_WGRAD_BATCH) SRAD_TRY(srad_wgrad_flush(q, stream));
  WgradReduceItem& it = q.batch.it[q.batch.count++];
  it.dW = dst; it.db = nullptr; it.part = part; it.n_real = ncols; it.cin_real = row_stride; it.ntaps = 0; it.grp_real = it.grp_pad = 0;   // ntaps 0: column sums
  it.tn = it.tc = 1; it.ksplit = rows; it.tile0 = q.tiles; it.alpha = alpha; it.wc = 0;
  q.tiles += (ncols + 63) / 64;
  return SRAD_OK;
}

int srad_wgrad_queue_ln_partials(WgradQueue& q, float* dgamma, float* dbeta, int C, int nrows, hipStream_t stream, float** part) {
  const size_t need = (size_t)nrows * 2 * LNB_CP;
  SRAD_REQUIRE(q.ws && need <= q.ws_floats, "ln_bwd: workspace too small");
  if (q.batch.count + 2 > SRAD_WGRAD_BATCH || q.used + need > q.ws_floats) SRAD_TRY(srad_wgrad_flush(q, stream));
  *part = q.ws + q.used;
  q.used += need;
  if (dgamma) SRAD_TRY(queue_colsum(q, dgamma, *part, C, 2 * LNB_CP, nrows, 1.f, stream));
  if (dbeta) SRAD_TRY(queue_colsum(q, dbeta, *part + LNB_CP, C, 2 * LNB_CP, nrows, 1.f, stream));
  return SRAD_OK;
}

int srad_launch_ln_bwd(const LnBwdParams& p, WgradQueue& q, hipStream_t stream) {
  SRAD_REQUIRE(p.rows > 0 && p.C >= 4 && p.C <= LNB_CP && (p.C & 3) == 0, "ln_bwd: channel count %d unsupported (4..%d, multiple of 4)", p.C, LNB_CP);
  SRAD_REQUIRE(((p.ldx | p.ld_dxn | p.ld_out | (p.dres ? p.ld_dres : 0)) & 3) == 0 &&
                   (((uintptr_t)p.x | (uintptr_t)p.dxn | (uintptr_t)p.out | (uintptr_t)p.dres | (uintptr_t)p.gamma) & 15) == 0,
               "ln_bwd: rows must be 16-byte aligned (strides multiples of 4 floats)");
  SRAD_REQUIRE(p.dxn && p.x && p.gamma && p.out, "ln_bwd: null argument");
  const int nwg = (p.rows + LNB_RPW - 1) / LNB_RPW;
  float* part = nullptr;
  SRAD_TRY(srad_wgrad_queue_ln_partials(q, p.dgamma, p.dbeta, p.C, nwg, stream, &part));
  SradProfScope prof(stream, SRAD_K_LN_BWD, 16.0 * p.rows * p.C, 4.0 * p.rows * p.C * (3 + (p.dres ? 1 : 0) + (p.accumulate ? 1 : 0)));
  auto go = [&](auto j4) {
    constexpr int J = decltype(j4)::value;
    if (p.dres && p.accumulate) hipLaunchKernelGGL((ln_bwd_kernel<true, true, J>), dim3(nwg), dim3(256), 0, stream, p, part);
    else if (p.dres) hipLaunchKernelGGL((ln_bwd_kernel<true, false, J>), dim3(nwg), dim3(256), 0, stream, p, part);
    else if (p.accumulate) hipLaunchKernelGGL((ln_bwd_kernel<false, true, J>), dim3(nwg), dim3(256), 0, stream, p, part);
    else hipLaunchKernelGGL((ln_bwd_kernel<false, false, J>), dim3(nwg), dim3(256), 0, stream, p, part);
  };
  if (p.C <= 256) go(std::integral_constant<int, 1>{});
  else go(std::integral_constant<int, 2>{});
  SRAD_CHECK_HIP(hipGetLastError());
  return SRAD_OK;
}

// windows other than 8 x 8 (N = ws^2 <= 256): window_attn_bwd_gen_kernel, fp32 operands and results in every precision mode
static int launch_attn_bwd_gen(const AttnBwdParams& p, WgradQueue& q, hipStream_t stream) {
  SRAD_REQUIRE(p.ws >= 1 && p.ws <= 16, "window_attn_bwd: window sizes 1 .. 16 train (got %d)", p.ws);
  SRAD_REQUIRE(p.qkv && p.dout && p.dqkv && !p.qkv_h && !p.dqkv_h, "window_attn_bwd: window sizes other than 8 take fp32 q | k | v, dO and dqkv");
  SRAD_REQUIRE(p.d / p.heads <= 128, "window_attn_bwd: head dims up to 128 (got %d)", p.d / p.heads);
  const int nW = (p.H / p.ws) * (p.W / p.ws), tw = 2 * p.ws - 1;
  const int ncols = tw * tw * p.heads, nwin = p.B * nW;
  const size_t need = (size_t)nwin * ncols;
  SRAD_REQUIRE(q.ws && need <= q.ws_floats, "window_attn_bwd: workspace too small");
  if (q.used + need > q.ws_floats) SRAD_TRY(srad_wgrad_flush(q, stream));
  float* tpart = q.ws + q.used;
  q.used += need;
  SRAD_TRY(queue_colsum(q, p.dtable, tpart, ncols, ncols, nwin, 1.f, stream));
  const double T = (double)p.B * p.H * p.W;
  SradProfScope prof(stream, SRAD_K_ATTN_BWD, 10.0 * T * p.ws * p.ws * p.d, 4.0 * T * 8 * p.d);
  const int nb = (p.ws * p.ws + 63) / 64;
  auto go = [&](auto kern, size_t lds) -> int {
    static SradOncePerDevice configured;                    // (one per kernel instance)
    if (configured.need()) {
      SRAD_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      configured.done();
    }
    hipLaunchKernelGGL(kern, dim3(nwin * p.heads), dim3(256), lds, stream, p, tpart);
    return SRAD_OK;
  };
  if (nb == 1) SRAD_TRY(go(window_attn_bwd_gen_kernel<1>, ABG_LDS(1)));
  else if (nb == 2) SRAD_TRY(go(window_attn_bwd_gen_kernel<2>, ABG_LDS(2)));
  else if (nb == 3) SRAD_TRY(go(window_attn_bwd_gen_kernel<3>, ABG_LDS(3)));
  else SRAD_TRY(go(window_attn_bwd_gen_kernel<4>, ABG_LDS(4)));
  SRAD_CHECK_HIP(hipGetLastError());
  return SRAD_OK;
}

int srad_launch_window_attn_bwd(int prec, const AttnBwdParams& p, WgradQueue& q, hipStream_t stream) {
  SRAD_REQUIRE(p.H % p.ws == 0 && p.W % p.ws == 0, "window_attn_bwd: %dx%d not a multiple of the window", p.H, p.W);
  SRAD_REQUIRE(p.d % p.heads == 0 && p.hdp % 4 == 0 && p.hdp >= p.d / p.heads, "window_attn_bwd: bad head geometry");
  SRAD_REQUIRE(p.shift >= 0 && p.shift < (p.ws > 0 ? p.ws : 1), "window_attn_bwd: bad shift %d", p.shift);
  if (p.ws != 8) return launch_attn_bwd_gen(p, q, stream);
  SRAD_REQUIRE(p.d % p.heads == 0 && p.hdp % 4 == 0 && p.hdp >= p.d / p.heads, "window_attn_bwd: bad head geometry");
  SRAD_REQUIRE(p.shift >= 0 && p.shift < p.ws, "window_attn_bwd: bad shift %d", p.shift);
  static SradOncePerDevice configured;
  if (configured.need()) {
    SRAD_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(window_attn_bwd_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)AB_LDS));
    configured.done();
  }
  const int nW = (p.H / p.ws) * (p.W / p.ws);
  const double T = (double)p.B * p.H * p.W;
  const int ncols = 225 * p.heads, nwin = p.B * nW;
  const size_t need = (size_t)nwin * ncols;
  SRAD_REQUIRE(q.ws && need <= q.ws_floats, "window_attn_bwd: workspace too small");
  if (q.used + need > q.ws_floats) SRAD_TRY(srad_wgrad_flush(q, stream));
  float* tpart = q.ws + q.used;
  q.used += need;
  SRAD_TRY(queue_colsum(q, p.dtable, tpart, ncols, ncols, nwin, 1.f, stream));
  SradProfScope prof(stream, SRAD_K_ATTN_BWD, 10.0 * T * 64 * p.d, 4.0 * T * 8 * p.d);
  if (prec == SRAD_PREC_BF16 && p.qkv_h) {
    const int hd = p.d / p.heads;
    SRAD_REQUIRE(p.dout_h && p.dqkv_h && hd <= 128 && p.hp_h % 8 == 0 && p.hp_h >= hd && p.hp_h <= 128 &&
                     (((uintptr_t)p.qkv_h | (uintptr_t)p.dout_h) & 15) == 0,
                 "window_attn_bwd: the all-bf16 form takes head dims <= 128 in 16-byte aligned slots of hp columns, and writes bf16");
    auto go = [&](auto kern, size_t lds) -> int {
      static SradOncePerDevice configured;                  // (one per instantiation of this lambda = per kernel instance)
      if (configured.need()) {
        SRAD_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        configured.done();
      }
      hipLaunchKernelGGL(kern, dim3(p.B * nW * p.heads), dim3(256), lds, stream, p, tpart);
      return SRAD_OK;
    };
    const int nch = (hd + 31) / 32;
    if (nch == 1) SRAD_TRY(go(window_attn_bwd_h_kernel<1>, ag_lds_bytes<1>()));
    else if (nch == 2) SRAD_TRY(go(window_attn_bwd_h_kernel<2>, ag_lds_bytes<2>()));
    else if (nch == 3) SRAD_TRY(go(window_attn_bwd_h_kernel<3>, ag_lds_bytes<3>()));
    else SRAD_TRY(go(window_attn_bwd_h_kernel<4>, ag_lds_bytes<4>()));
  } else if (prec == SRAD_PREC_BF16) {
    static SradOncePerDevice configured16;
    if (configured16.need()) {
      SRAD_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(window_attn_bwd_bf16_kernel),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)AH_LDS));
      configured16.done();
    }
    hipLaunchKernelGGL(window_attn_bwd_bf16_kernel, dim3(p.B * nW * p.heads), dim3(256), AH_LDS, stream, p, tpart);
  } else {
    hipLaunchKernelGGL(window_attn_bwd_kernel, dim3(p.B * nW * p.heads), dim3(256), AB_LDS, stream, p, tpart);
  }
  SRAD_CHECK_HIP(hipGetLastError());
  return SRAD_OK;
}

int srad_launch_dact(const float* dy, int ld_dy, const float* y, int ld_y, float* out, int ld_out, int rows, int C,
                     float slope, hipStream_t stream) {
  const size_t total = (size_t)rows * C;
  SradProfScope prof(stream, SRAD_K_MISC, 1.0 * total, 12.0 * total);
  hipLaunchKernelGGL(dact_kernel, dim3(grid_for(total)), dim3(256), 0, stream, dy, ld_dy, y, ld_y, out, ld_out, rows, C, slope);
  SRAD_CHECK_HIP(hipGetLastError());
  return SRAD_OK;
}

int srad_launch_unshuffle(const float* src, float* dst, int B, int H, int W, int F, hipStream_t stream) {
  const size_t total = (size_t)B * H * W * 4 * F;
  SradProfScope prof(stream, SRAD_K_LAYOUT, 0.0, 8.0 * total);
  hipLaunchKernelGGL(unshuffle_kernel, dim3(grid_for(total)), dim3(256), 0, stream, src, dst, B, H, W, F);
  SRAD_CHECK_HIP(hipGetLastError());
  return SRAD_OK;
}

int srad_launch_copy_cols(const float* src, int ld_src, float* dst, int ld_dst, int rows, int C, hipStream_t stream) {
  const size_t total = (size_t)rows * ld_dst;
  SradProfScope prof(stream, SRAD_K_LAYOUT, 0.0, 4.0 * total + 4.0 * rows * C);
  hipLaunchKernelGGL(copy_cols_kernel, dim3(grid_for(total)), dim3(256), 0, stream, src, ld_src, dst, ld_dst, rows, C);
  SRAD_CHECK_HIP(hipGetLastError());
  return SRAD_OK;
}

int srad_launch_l1_grad(const float* a, const float* b, float* out, size_t n, float scale, hipStream_t stream) {
  SradProfScope prof(stream, SRAD_K_OPTIM, 2.0 * n, 12.0 * n);
  hipLaunchKernelGGL(l1_grad_kernel, dim3(grid_for(n)), dim3(256), 0, stream, a, b, out, n, scale);
  SRAD_CHECK_HIP(hipGetLastError());
  return SRAD_OK;
}

int srad_launch_adam(float* p, const float* g, float* m, float* v, size_t n, float lr, float beta1, float beta2,
                     float eps, float weight_decay, int step, float grad_scale, hipStream_t stream) {
  SRAD_REQUIRE(step >= 1, "adam: step counts from 1 (got %d)", step);
  const double bc1 = 1.0 - pow((double)beta1, step), bc2 = 1.0 - pow((double)beta2, step);
  SradProfScope prof(stream, SRAD_K_OPTIM, 12.0 * n, 28.0 * n);
  hipLaunchKernelGGL(adam_kernel, dim3(grid_for(n)), dim3(256), 0, stream, p, g, m, v, n, lr, beta1, beta2, eps,
                     weight_decay, (float)bc1, (float)sqrt(bc2), grad_scale);
  SRAD_CHECK_HIP(hipGetLastError());
  return SRAD_OK;
}

int srad_launch_adam_dev(float* p, const float* g, float* m, float* v, size_t n, float beta1, float beta2, float eps,
                         float weight_decay, const float* hyper, hipStream_t stream) {
  SradProfScope prof(stream, SRAD_K_OPTIM, 12.0 * n, 28.0 * n);
  hipLaunchKernelGGL(adam_dev_kernel, dim3(grid_for(n)), dim3(256), 0, stream, p, g, m, v, n, beta1, beta2, eps, weight_decay, hyper);
  SRAD_CHECK_HIP(hipGetLastError());
  return SRAD_OK;
}


namespace {
__global__ void set4_kernel(float* __restrict__ dst, float a, float b, float c, float d) {
  if (threadIdx.x == 0) { dst[0] = a; dst[1] = b; dst[2] = c; dst[3] = d; }
}
}  // namespace
int srad_launch_set4(float* dst, float a, float b, float c, float d, hipStream_t stream) {
  hipLaunchKernelGGL(set4_kernel, dim3(1), dim3(64), 0, stream, dst, a, b, c, d);
  SRAD_CHECK_HIP(hipGetLastError());
  return SRAD_OK;
}
