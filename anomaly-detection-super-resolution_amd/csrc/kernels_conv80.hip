// kernels_conv80.hip - the 80 -> 80 channel 3x3 convolution of DRN-L's RCAB chains (reference src/drn.py:143-158: 160 of
// them per forward, most of the model's FLOPs) as a weight-resident, persistent kernel for gfx950.
//
// The tiled GEMM (kernels_gemm.hip) gives every 128-pixel tile its own workgroup, which streams the whole 115 KB weight and
// nine shifted copies of its input rows through six load -> LDS -> barrier -> MFMA stages: 41 us per launch, 230 TFLOP/s.
// Here a workgroup keeps the WHOLE weight in LDS (80 rows x 736 k as bf16, 119 KB) for its lifetime and walks pixel tiles of
// 4 rows x 32 columns: a tile's input halo (6 x 34 pixels x 80 channels) is fetched once - each input pixel once, not once
// per tap - rounded to bf16 into a 36 KB LDS tile, and all nine taps read it through shifted addresses.  The next tile's
// halo is in flight in registers while the current one is multiplied.  K runs over (tap, channel) = 720 values in 32-wide
// chunks; a chunk may straddle two taps, which only changes the pixel offset per lane group (a lane's eight k values never
// straddle: 8 divides 80).  MFMA operands are swapped (A = weight fragment, B = activation fragment), so a lane ends with four
// consecutive output channels of one pixel: bias / residual / output are float4.
// Same arithmetic as the GEMM path (bf16 operands, fp32 accumulation); the summation order over k differs.
#include "srad_common.h"
#include <algorithm>
#include <type_traits>
#include <vector>

namespace {

template <int B, int N, class F>
__device__ __forceinline__ void c80_static_for(F&& f) {
  if constexpr (B < N) {
    f(std::integral_constant<int, B>{});
    c80_static_for<B + 1, N>(f);
  }
}

constexpr int C80_K = 736;                // 9 taps x 80 channels = 720, padded to 23 chunks of 32 (the weight rows are zero there)
// LDS strides.  ds_read_b128 is served in four groups of 16 lanes that are NOT lane-contiguous ({0-3, 12-15, 20-27}, {4-11, 16-19,
// 28-31}, ...: MI355X_MICROARCH.md, LDS): with lane = 16 fq + fr reading row fr at byte 16 fq, a group holds eight rows at one
// fq and the OTHER eight rows at the next fq, so its sixteen 16-byte pieces fall on distinct bank quads iff the row stride
// is 2 (mod 4) sixteen-byte units.  The strides chosen for lane-contiguous groups (1488 B = 93 units, 176 B = 11 units) gave
// 2-way conflicts on every fragment read: SQ_LDS_BANK_CONFLICT 46 % of the LDS cycles of a kernel that LDS reads bound.
constexpr int C80_WLD = C80_K + 16;       // weight row stride (elements): 1504 B = 94 units
constexpr int C80_PLD = 80;               // halo pixel stride (elements): 160 B = 10 units - the pixels are contiguous, stores conflict-free too
constexpr int C80_TH = 4, C80_TW = 32;    // output tile: 4 rows x 32 columns = 128 pixels; wave w owns row w
constexpr int C80_HH = C80_TH + 2, C80_HW = C80_TW + 2;
constexpr int C80_HCH = C80_HH * C80_HW * 20;      // float4 chunks of a halo tile (20 per pixel)
constexpr int C80_NT = 512;                         // threads: wave = (tile row, half of the output channels)
constexpr int C80_NL = (C80_HCH + C80_NT - 1) / C80_NT;      // per thread
constexpr size_t C80_LDS = (size_t)(80 * C80_WLD + C80_HH * C80_HW * C80_PLD) * sizeof(__bf16) + 4 * 80 * sizeof(float);

// STAMP: diagnostic build (SRAD_C80_STAMP=1, tools/conv_bench.py): s_memtime of every wave at the phase boundaries
// RM: the residual operand - 0 none, 1 fp32 (p.R), 2 bf16 (p.Rh).  Its tile is requested BEFORE the tile's MFMAs and the bias sits
// in registers for the kernel's lifetime: as loads inside the epilogue they were a dependent global round trip per tile, and
// the epilogue took 4 300 of a tile's 12 300 cycles (stamps: SRAD_C80_STAMP=1).
template <bool XH, int RM = 0, bool STAMP = false>
__global__ __launch_bounds__(C80_NT) void conv80_kernel(const GemmParams p, const int ntiles, const int tiles_x, const int tiles_per_img,
                                                        unsigned long long* __restrict__ stamps = nullptr) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  __bf16* const Ws = reinterpret_cast<__bf16*>(smem);                       // [80][C80_WLD]
  __bf16* const Hs = Ws + 80 * C80_WLD;                                     // [6 * 34][C80_PLD]
  float* const red = reinterpret_cast<float*>(Hs + C80_HH * C80_HW * C80_PLD);   // [4][80] pool partials
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 15, fq = lane >> 4;
  const int H = p.Hi, W = p.Wi;
  [[maybe_unused]] int stamp_i = 0;
  auto stamp = [&]() __attribute__((always_inline)) {
    if constexpr (STAMP) {
      unsigned long long t;
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
      __builtin_amdgcn_sched_barrier(0);
      if (lane == 0 && stamp_i < 32) stamps[((size_t)blockIdx.x * 8 + wave) * 32 + stamp_i] = t;
      ++stamp_i;
    }
  };
  stamp();                                                                  // 0: start

  // ---- halo of tile t -> registers (clamped addresses, zero outside the image) ----
  typedef typename std::conditional<XH, u32x2, f32x4>::type hreg_t;       // four channels of a halo pixel
  hreg_t hreg[C80_NL];
  unsigned hok = 0u;
  auto load_halo = [&](int t) __attribute__((always_inline)) {
    const int b = t / tiles_per_img, tl = t - b * tiles_per_img;
    const int ty = tl / tiles_x, tx = tl - ty * tiles_x;
    const int y0 = ty * C80_TH - 1, x0 = tx * C80_TW - 1;
    hok = 0u;
#pragma unroll
    for (int i = 0; i < C80_NL; ++i) {
      const int idx = min(tid + C80_NT * i, C80_HCH - 1);
      const int hp = idx / 20, c4 = idx - hp * 20;
      const int hy = hp / C80_HW, hx = hp - hy * C80_HW;
      const int yy = y0 + hy, xx = x0 + hx;
      const bool in = yy >= 0 && yy < H && xx >= 0 && xx < W;
      const int yc = min(max(yy, 0), H - 1), xc = min(max(xx, 0), W - 1);
      const size_t off = ((size_t)(b * H + yc) * W + xc) * p.ldx + 4 * c4;
      if constexpr (XH) hreg[i] = *reinterpret_cast<const u32x2*>(p.Xh + off);
      else hreg[i] = *reinterpret_cast<const f32x4*>(p.X + off);
      hok |= in ? (1u << i) : 0u;
    }
  };
  auto store_halo = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < C80_NL; ++i) {
      const int idx = tid + C80_NT * i;
      if (idx < C80_HCH) {
        const int hp = idx / 20, c4 = idx - hp * 20;
        if constexpr (XH) {
          *reinterpret_cast<u32x2*>(Hs + hp * C80_PLD + 4 * c4) = (hok >> i) & 1u ? hreg[i] : u32x2{0u, 0u};
        } else {
          const f32x4 v = (hok >> i) & 1u ? hreg[i] : f32x4{0.f, 0.f, 0.f, 0.f};
          bf16x4 h;
          h[0] = (__bf16)v[0]; h[1] = (__bf16)v[1]; h[2] = (__bf16)v[2]; h[3] = (__bf16)v[3];
          *reinterpret_cast<bf16x4*>(Hs + hp * C80_PLD + 4 * c4) = h;
        }
      }
    }
  };

  int t = blockIdx.x;
  load_halo(t);
  // ---- the weight, once: global [Np][9][96] bf16 -> LDS [80][tap * 80 + ch | 16 zeros] ----
  {
    // 80 x 92 sixteen-byte pieces (90 real + 2 zero per row): all of a batch's loads in flight before its first LDS store
    // (as a run-time loop each of the 29 rounds waited for its own load: 25 us per launch)
    const u32x4* const wg = reinterpret_cast<const u32x4*>(p.Wp);
    constexpr int NP = 80 * 92, NB = 15;
#pragma unroll
    for (int i0 = 0; i0 < (NP + C80_NT - 1) / C80_NT; i0 += NB) {
      u32x4 wr[NB];
#pragma unroll
      for (int i = 0; i < NB; ++i) {
        const int idx = min(tid + C80_NT * (i0 + i), NP - 1);
        const int n = idx / 92, pc = min(idx - n * 92, 89);
        const int tap = pc / 10, c8 = pc - tap * 10;
        const int nrow = p.sp_q >= 0 ? 4 * n + p.sp_q : n;        // sub-pixel mode: every fourth row of the 320-row pack
        wr[i] = wg[(nrow * (9 * 96) + tap * 96 + c8 * 8) / 8];
      }
#pragma unroll
      for (int i = 0; i < NB; ++i) {
        const int idx = tid + C80_NT * (i0 + i);
        const int n = idx / 92, pc = idx - n * 92;
        if (idx < NP) *reinterpret_cast<u32x4*>(Ws + n * C80_WLD + pc * 8) = pc < 90 ? wr[i] : u32x4{0u, 0u, 0u, 0u};
      }
    }
  }
  stamp();                                                                  // 1: weight in LDS (stores issued)
  const int nt0w = (wave >> 2) ? 3 : 0;
  f32x4 bias_r[3] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
  if (p.bias) {
#pragma unroll
    for (int nt = 0; nt < 3; ++nt) {
      const int c = min((nt0w + nt) * 16 + 4 * fq, 76);
      if (p.sp_q >= 0) bias_r[nt] = f32x4{p.bias[4 * c + p.sp_q], p.bias[4 * (c + 1) + p.sp_q], p.bias[4 * (c + 2) + p.sp_q], p.bias[4 * (c + 3) + p.sp_q]};
      else bias_r[nt] = *reinterpret_cast<const f32x4*>(p.bias + c);
    }
  }
  store_halo();
  stamp();                                                                  // 2: first halo arrived and stored
  __syncthreads();
  stamp();                                                                  // 3

  const f32x4 z4 = f32x4{0.f, 0.f, 0.f, 0.f};
  const int row = wave & 3, nh = wave >> 2;
  // A lane's first k of chunk kc is 32 kc + 8 fq: (tap, channel) advances by 32 channels per chunk and wraps into the next tap
  // at 80.  The 23 halo-tile offsets this gives do not depend on the tile: they live in registers for the kernel's lifetime, and
  // the MFMA loop issues no address arithmetic at all (the incremental walk was ~12 VALU instructions per chunk and wave on the
  // port the MFMAs issue through).
  constexpr int C80_NCH = C80_K / 32;
  int aoff[C80_NCH];
  {
    int ch = 8 * fq, dx = 0, rowoff = (row * C80_HW + fr) * C80_PLD;
#pragma unroll
    for (int kc = 0; kc < C80_NCH; ++kc) {
      aoff[kc] = rowoff + dx * C80_PLD + ch;
      if (kc == C80_NCH - 1 && fq >= 2) aoff[kc] = (row * C80_HW + fr) * C80_PLD;   // k >= 720: the weights are zero there, any valid address does
      ch += 32;
      if (ch >= 80) {
        ch -= 80;
        if (++dx == 3) { dx = 0; rowoff += C80_HW * C80_PLD; }
      }
    }
  }                   // this wave: tile row, and output channel tiles {0, 1, 2} or {3, 4}
  // One tile: NTW channel tiles from nt0 on.
  auto tile = [&](auto NTW_c, const int nt0, const int t) __attribute__((always_inline)) {
    constexpr int NTW = decltype(NTW_c)::value;
    f32x4 acc[2][NTW];
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
      for (int nt = 0; nt < NTW; ++nt) acc[rt][nt] = z4;
    const int b = t / tiles_per_img, tl = t - b * tiles_per_img;
    const int ty = tl / tiles_x, tx = tl - ty * tiles_x;
    const int yy = ty * C80_TH + row;
    const size_t pix0 = (size_t)(b * H + yy) * W + tx * C80_TW + fr;
    // where the outputs of pixels pix0 / pix0 + 16 go: the same pixels, or (sub-pixel mode) their (dy, dx) position in the 2x image
    const bool sp = p.sp_q >= 0;
    const size_t opix0 = sp ? ((size_t)(b * 2 * H + 2 * yy + (p.sp_q >> 1)) * (2 * W) + 2 * (tx * C80_TW + fr) + (p.sp_q & 1)) : pix0;
    const size_t opix16 = sp ? 32 : 16;                        // output-pixel distance of the tile's second 16-pixel half
    typedef typename std::conditional<RM == 2, u32x2, f32x4>::type rreg_t;
    [[maybe_unused]] rreg_t rres[2][NTW];
    if constexpr (RM != 0) {
#pragma unroll
      for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt) {
          const size_t o = (pix0 + 16 * rt) * p.ldr + (nt0 + nt) * 16 + 4 * fq;
          if constexpr (RM == 2) rres[rt][nt] = *reinterpret_cast<const u32x2*>(p.Rh + o);
          else rres[rt][nt] = *reinterpret_cast<const f32x4*>(p.R + o);
        }
    }
    const __bf16* const wrow = Ws + (nt0 * 16 + fr) * C80_WLD + 8 * fq;
    // fragments of chunk kc + DEPTH are requested before chunk kc is multiplied (three register sets): with two waves per SIMD
    // and the compiler's one-chunk look-ahead the waves spent 59 % of their time parked on LDS latency (SQ_WAIT_ANY)
    constexpr int NCH = C80_NCH, DEPTH = 2;
    bf16x8 fa[DEPTH + 1][2], fw[DEPTH + 1][NTW];
    auto fetch = [&](auto KC) __attribute__((always_inline)) {
      constexpr int kc = decltype(KC)::value;
      if constexpr (kc < NCH) {
        constexpr int st = kc % (DEPTH + 1);
        fa[st][0] = *reinterpret_cast<const bf16x8*>(Hs + aoff[kc]);
        fa[st][1] = *reinterpret_cast<const bf16x8*>(Hs + aoff[kc] + 16 * C80_PLD);
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt) fw[st][nt] = *reinterpret_cast<const bf16x8*>(wrow + nt * 16 * C80_WLD + 32 * kc);
      }
    };
    c80_static_for<0, DEPTH>([&](auto K) { fetch(K); });
    c80_static_for<0, NCH>([&](auto K) {
      constexpr int kc = decltype(K)::value, st = kc % (DEPTH + 1);
      fetch(std::integral_constant<int, kc + DEPTH>{});
#pragma unroll
      for (int nt = 0; nt < NTW; ++nt) {
        acc[0][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[st][nt], fa[st][0], acc[0][nt], 0, 0, 0);
        acc[1][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[st][nt], fa[st][1], acc[1][nt], 0, 0, 0);
      }
      // Pin the software pipeline: this step's LDS reads (chunk kc + DEPTH) go out one behind each of its MFMAs (chunk kc) and
      // nothing crosses the step.  Left alone the scheduler sinks every read to just above its first use (s_waitcnt lgkmcnt(1)
      // in front of most MFMAs): the LDS latency of every chunk was exposed, MFMA and LDS time ADDED (7 300 cycles per tile for
      // 3 700 cycles of MFMA and 3 300 of LDS reads).
      constexpr int nrd = kc + DEPTH < NCH ? 2 + NTW : 0, nmf = 2 * NTW;
#pragma unroll
      for (int i = 0; i < nrd; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);      // one MFMA
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);      // one LDS read
      }
      if constexpr (nmf > nrd) __builtin_amdgcn_sched_group_barrier(0x008, nmf - nrd, 0);
      __builtin_amdgcn_sched_barrier(0);
    });
    stamp();                                                                // 4 + 5 i: MFMAs of tile i done
    // ---- epilogue: + bias -> activation -> * alpha -> residual mode -> store (GemmParams semantics) ----
    f32x4 csum[NTW];
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt) csum[nt] = z4;
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt) {
      const int c = (nt0 + nt) * 16 + 4 * fq;
      f32x4 vv[2];
#pragma unroll
      for (int rt = 0; rt < 2; ++rt) {
        f32x4 v = acc[rt][nt] + bias_r[nt];
        if (p.act == SRAD_ACT_RELU) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
        } else if (p.act == SRAD_ACT_LRELU) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : v[e] * p.slope;
        }
        v = v * p.alpha;
        if constexpr (RM != 0) {
          f32x4 r;
          if constexpr (RM == 2) {
            const bf16x4 rh = __builtin_bit_cast(bf16x4, rres[rt][nt]);
            r = f32x4{(float)rh[0], (float)rh[1], (float)rh[2], (float)rh[3]};
          } else {
            r = rres[rt][nt];
          }
          if (p.rmode == SRAD_RMODE_ADD) v += r;
          else {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] *= r[e] > 0.f ? 1.f : p.slope;
          }
        }
        csum[nt] += v;
        vv[rt] = v;
      }
      if (p.Yh) {
        // bf16 output: a lane holds 4 channels (8 bytes) of pixel fr and of pixel 16 + fr.  8-byte stores are issue-bound (six
        // of them were 2 800 of a tile's 8 000 cycles): the lane pair (fq, fq ^ 1) trades halves with v_permlane16_swap - rows of
        // 16 lanes ARE the fq groups - so that the even lane ends with 8 consecutive channels of pixel fr, the odd lane with
        // the same 8 channels of pixel 16 + fr: one 16-byte store each.
        bf16x4 h0, h1;
#pragma unroll
        for (int e = 0; e < 4; ++e) { h0[e] = (__bf16)vv[0][e]; h1[e] = (__bf16)vv[1][e]; }
        const u32x2 a = __builtin_bit_cast(u32x2, h0), bq = __builtin_bit_cast(u32x2, h1);
        const auto s0 = __builtin_amdgcn_permlane16_swap(a[0], bq[0], false, false);
        const auto s1 = __builtin_amdgcn_permlane16_swap(a[1], bq[1], false, false);
        const size_t pix = opix0 + opix16 * (fq & 1);
        *reinterpret_cast<u32x4*>(p.Yh + pix * p.ldy + p.yoff + (nt0 + nt) * 16 + 4 * (fq & ~1)) = u32x4{s0[0], s1[0], s0[1], s1[1]};
      } else {
        *reinterpret_cast<f32x4*>(p.Y + opix0 * p.ldy + p.yoff + c) = vv[0];
        *reinterpret_cast<f32x4*>(p.Y + (opix0 + opix16) * p.ldy + p.yoff + c) = vv[1];
      }
    }
    if (p.pool_part) {                                        // the tile's column sums: 16 pixels per lane group, then the four rows
#pragma unroll
      for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
        for (int e = 0; e < 4; ++e) csum[nt][e] = srad_row16_sum(csum[nt][e]);
      if (fr == 0) {
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt) *reinterpret_cast<f32x4*>(red + row * 80 + (nt0 + nt) * 16 + 4 * fq) = csum[nt];
      }
    }
  };
  while (true) {
    const int tn = t + gridDim.x;
    const bool more = tn < ntiles;
    load_halo(more ? tn : t);                                 // (the last tile loads itself again: no load behind a branch)
    if (nh == 0) tile(std::integral_constant<int, 3>{}, 0, t);
    else tile(std::integral_constant<int, 2>{}, 3, t);
    stamp();                                                  // 5 + 5 i: epilogue issued
    __syncthreads();                                          // everyone is done with the halo tile (and red is written)
    stamp();                                                  // 6 + 5 i
    if (p.pool_part && tid < 80) p.pool_part[(size_t)t * 80 + tid] = (red[tid] + red[80 + tid]) + (red[160 + tid] + red[240 + tid]);
    if (!more) break;
    store_halo();
    stamp();                                                  // 7 + 5 i: next halo stored
    t = tn;
    __syncthreads();
    stamp();                                                  // 8 + 5 i
  }
}

}  // namespace

bool srad_conv80_supported(int prec, const GemmParams& p) {
  static const bool off = getenv("SRAD_NO_CONV80") != nullptr;
  return !off && prec == SRAD_PREC_BF16 && p.ntaps == 9 && p.stride == 1 && p.Cin == 80 && p.N == 80 && !p.ln_g && p.ps == 0 &&
         p.hsplit_hd == 0 && !p.row_scale && !p.Ypre && (p.act == SRAD_ACT_NONE || p.act == SRAD_ACT_RELU || p.act == SRAD_ACT_LRELU) &&   // the epilogue has no GELU: such a call stays on the tiled GEMM
         p.Hi == p.Ho && p.Wi == p.Wo && p.Hi % C80_TH == 0 && p.Wi % C80_TW == 0 &&
         ((!p.R && !p.Rh) || p.rmode == SRAD_RMODE_ADD || p.rmode == SRAD_RMODE_DLRELU) && !(p.R && p.Rh) && (p.ldx & 3) == 0 && (p.ldy & 3) == 0 &&
         (p.yoff & 3) == 0 && ((!p.R && !p.Rh) || (p.ldr & 3) == 0) && ((uintptr_t)p.Rh & 7) == 0 && (((uintptr_t)p.X | (uintptr_t)p.Y | (uintptr_t)p.R | (uintptr_t)p.bias | (uintptr_t)p.Wp) & 15) == 0 &&
         ((uintptr_t)p.Xh & 7) == 0 && ((uintptr_t)p.Yh & 15) == 0 && (!p.Yh || ((p.ldy & 7) == 0 && (p.yoff & 7) == 0)) &&
         (p.sp_q < 0 || (p.sp_q <= 3 && !p.R && !p.Rh && !p.pool_part && p.bias)) &&
         p.M >= 128 * 64;                                        // small launches stay on the tiled GEMM (one tile per workgroup anyway)
}

int srad_launch_conv80(const GemmParams& p, hipStream_t stream) {
  SRAD_REQUIRE(srad_conv80_supported(SRAD_PREC_BF16, p), "conv80: unsupported problem");
  const int tiles_x = p.Wi / C80_TW, tiles_per_img = (p.Hi / C80_TH) * tiles_x;
  const int B = p.M / (p.Hi * p.Wi), ntiles = B * tiles_per_img;
  const int nwg = ntiles < 256 ? ntiles : 256;
  const double K = 9.0 * 80;
  SradProfScope prof(stream, SRAD_K_CONV80, 2.0 * p.M * 80 * K,
                     (double)p.M * 80 * ((p.Xh ? 2 : 4) + (p.Yh ? 2 : 4) + (p.R ? 4 : p.Rh ? 2 : 0)) + 2.0 * 80 * K);
  static const bool stamp_build = getenv("SRAD_C80_STAMP") != nullptr;
  auto launch = [&](auto kern, SradOncePerDevice& configured) -> int {
    if (configured.need()) {
      SRAD_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)C80_LDS));
      configured.done();
    }
    hipLaunchKernelGGL(kern, dim3(nwg), dim3(C80_NT), C80_LDS, stream, p, ntiles, tiles_x, tiles_per_img, (unsigned long long*)nullptr);
    return SRAD_OK;
  };
  static SradOncePerDevice cfg[6];
  const int rm = p.R ? 1 : p.Rh ? 2 : 0;
  if (stamp_build && p.Xh && rm == 0) {                         // diagnostic: phase stamps of every wave, medians to stderr (synchronous)
    static unsigned long long* dbuf = nullptr;
    const size_t n = (size_t)256 * 8 * 32;
    if (!dbuf) {
      SRAD_CHECK_HIP(hipMalloc(&dbuf, n * sizeof(unsigned long long)));
      SRAD_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(conv80_kernel<true, 0, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)C80_LDS));
    }
    SRAD_CHECK_HIP(hipMemsetAsync(dbuf, 0, n * sizeof(unsigned long long), stream));
    hipLaunchKernelGGL((conv80_kernel<true, 0, true>), dim3(nwg), dim3(C80_NT), C80_LDS, stream, p, ntiles, tiles_x, tiles_per_img, dbuf);
    SRAD_CHECK_HIP(hipStreamSynchronize(stream));
    static int printed_small = 0, printed_big = 0;
    int& printed = ntiles <= 256 ? printed_small : printed_big;
    if (++printed < 5 || printed > 6) return SRAD_OK;              // the fifth and sixth launch of each size class
    std::vector<unsigned long long> hb(n);
    SRAD_CHECK_HIP(hipMemcpy(hb.data(), dbuf, n * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    const int nst = 4 + 5 * ((ntiles + nwg - 1) / nwg) - 2;
    fprintf(stderr, "conv80 stamps (s_memtime ticks; median over workgroups, wave 0 | wave 4), %d tiles on %d workgroups:\n", ntiles, nwg);
    for (int i = 1; i < nst && i < 32; ++i) {
      std::vector<double> d0, d4;
      for (int g = 0; g < nwg; ++g) {
        const unsigned long long* a = hb.data() + ((size_t)g * 8 + 0) * 32;
        const unsigned long long* b = hb.data() + ((size_t)g * 8 + 4) * 32;
        if (a[i] && a[i - 1]) d0.push_back((double)(a[i] - a[i - 1]));
        if (b[i] && b[i - 1]) d4.push_back((double)(b[i] - b[i - 1]));
      }
      std::sort(d0.begin(), d0.end()); std::sort(d4.begin(), d4.end());
      fprintf(stderr, "  phase %2d: %7.0f | %7.0f   (n = %zu)\n", i, d0.empty() ? 0.0 : d0[d0.size() / 2], d4.empty() ? 0.0 : d4[d4.size() / 2], d0.size());
    }
    return SRAD_OK;
  }
  int rc;
  if (p.Xh) rc = rm == 0 ? launch(conv80_kernel<true, 0>, cfg[0]) : rm == 1 ? launch(conv80_kernel<true, 1>, cfg[1]) : launch(conv80_kernel<true, 2>, cfg[2]);
  else rc = rm == 0 ? launch(conv80_kernel<false, 0>, cfg[3]) : rm == 1 ? launch(conv80_kernel<false, 1>, cfg[4]) : launch(conv80_kernel<false, 2>, cfg[5]);
  if (rc) return rc;
  SRAD_CHECK_HIP(hipGetLastError());
  return SRAD_OK;
}
