// kernels_fused_attn.hip - the FIRST half of a DRCT Swin block as one bf16 launch per (window, head):
//
//   xn      = LayerNorm1(x[window tokens])                                   (src/drct.py:477)
//   q|k|v   = xn . Wqkv[head]^T + b                                          (src/drct.py:278)
//   out     = softmax(q*scale k^T + rel-pos bias + shift mask) v             (src/drct.py:281-299)
//
// with the cyclic shift / window partition / reverse (drct.py:482-504) as token index arithmetic.
// The window's 64 tokens (window size 8) are gathered once, normalised in registers, and stay in LDS as
// the A operand; the head's three weight slices stream from a per-head fragment pack straight into MFMA
// registers (no LDS weight stage: 33.8 KB less LDS, two workgroups per CU for the 6-head blocks); q, k, v
// never exist in HBM.  This replaces the LN1+QKV GEMM launch, its [T][3d] round trip and the attention
// launch.  Same rules as kernels_gemm.hip / kernels_fused.hip (unconditional clamped loads, selects for
// padding, template-unrolled stages, transposed MFMA results so a lane owns 4 consecutive columns).
#include "srad_common.h"
#include <type_traits>

namespace {

template <int B, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (B < N) {
    f(std::integral_constant<int, B>{});
    static_for<B + 1, N>(f);
  }
}

constexpr int QA_LDP = 80;     // probability tile row stride (2 mod 4 sixteen-byte units: conflict-free for the lane groups of ds_read_b128, see kernels_conv80.hip)
constexpr int QA_LDB = 68;     // bias table row stride (floats): the 16 keys a float4 read group touches fall on 16 different bank quads

// HDT = ceil(head_dim / 16) (2, 3, 4, 5 or 8), KC = ceil(d / 32) 32-wide k chunks (<= 10)
// SPLIT = the split-bf16 mode (SRAD_PREC_BF16X3, inference): the normalised window, q, k, v and the probabilities each exist as a
// hi and a lo bf16 plane, the weights stream twice (hi pack, lo pack) and every product is three MFMAs (hi.hi + hi.lo + lo.hi).
// Two planes of everything do not fit next to each other (head dim 122: 208 KB), so q | k | v and P OVERLAY the normalised
// window: a wave keeps its q | k | v accumulators of all (<= 3) column stages in registers until every wave is done reading
// the window (one more barrier), then writes them.  The output is fp32.
template <int HDT, int KC, bool STAMP = false, bool SPLIT = false>
__global__ __launch_bounds__(512) void qkv_attn_kernel(const QkvAttnParams p) {
  constexpr int KG = (KC + 7) / 8;         // 256-wide k groups per weight stage
  constexpr int QA_LDX = KC * 32 + 16;     // LDS row stride of the normalised window tile
  constexpr int HDP = 16 * HDT;            // padded head dim
  constexpr int HDP32 = (HDP + 31) & ~31;  // k extent of the q.k^T MFMA steps
  constexpr int HS = HDP32 + 16;           // q/k/v LDS row stride
  constexpr int NV = 3 * HDP;              // virtual output columns [q | k | v]
  constexpr int NS = (NV + 127) / 128;     // 128-column stages: each of the 8 waves owns 16 of them
  constexpr int n_stages = NS * KG;
  constexpr int NPART = SPLIT ? 2 : 1;                          // weight streams per stage (hi | hi, lo)
  constexpr int n_vst = n_stages * NPART;
  constexpr int XN_E = 64 * QA_LDX, QKV_E = 3 * 64 * HS, PS_E = 64 * QA_LDP;     // bf16 elements per plane
  constexpr int R0_E = SPLIT ? (2 * XN_E > 2 * (QKV_E + PS_E) ? 2 * XN_E : 2 * (QKV_E + PS_E)) : XN_E + QKV_E + PS_E;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  __bf16* XN = reinterpret_cast<__bf16*>(smem);                 // [64][QA_LDX]
  __bf16* XNL = XN + XN_E;                                      // split-bf16: its lo plane
  __bf16* QKV = SPLIT ? XN : XN + XN_E;                         // [3][64][HS] (split-bf16: over the window tile, see above)
  __bf16* QKVL = QKV + QKV_E;
  __bf16* Ps = SPLIT ? QKVL + QKV_E : QKV + QKV_E;              // [64][QA_LDP] probabilities
  __bf16* PsL = Ps + PS_E;
  float* v_g = reinterpret_cast<float*>(XN + R0_E);             // [320] gamma
  float* v_b = v_g + 320;                                       // [320] beta
  float* v_bias = v_b + 320;                                    // [3][HDP] q|k|v bias of this head (0 in padding)
  float* tbl = v_bias + 3 * HDP;                                // [225] relative position bias of this head
  float* lsum = tbl + 232;                                      // [2][64] softmax denominators of the two key halves
  float* mxs = lsum + 128;                                      // [2][64] row maxima of the two key halves
  float* BM = mxs + 128;                                        // [64 keys][QA_LDB] bias + shift mask, query-contiguous
  int* tok = reinterpret_cast<int*>(BM + 64 * QA_LDB);          // [64] token index
  int* inf = tok + 64;                                          // [64] (region << 16) | (py << 8) | px

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 15, fq = lane >> 4;
  const int rt = wave & 3, chh = wave >> 2;                     // P.V: row tile, column-tile parity
  const int wave_s = __builtin_amdgcn_readfirstlane(wave);
  const int d = p.d, heads = p.heads, hd = d / heads;
  // diagnostic build only (tools/stamp_bench.py): shader-clock stamps of every wave at the phase boundaries
  auto stamp = [&](int idx) {
    if constexpr (STAMP) {
      unsigned long long t;
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
      __builtin_amdgcn_sched_barrier(0);
      if (lane == 0) p.stamps[((size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 8 + wave) * 16 + idx] = t;
    }
  };
  stamp(0);
  // grid = (windows, heads): linear ids of one window's heads differ by a multiple of 8 when the window
  // count is, so they share an XCD and the window's x rows are fetched into one L2 only
  // XCD affinity with the kernels before and after (workgroup L runs on XCD L % 8, each XCD has its own L2): XCD k takes
  // windows [k n/8, (k+1) n/8) - a contiguous strip of the images, the same strip whose token rows mlp_block gives XCD k -
  // so the rows this workgroup gathers were mostly written into ITS L2 by the previous launch
  const int h = blockIdx.y;
  int win = blockIdx.x;
  if ((gridDim.x & 7) == 0 && !p.no_xcd_map) win = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
  const int ws = 8, nWx = p.W / ws, nW = (p.H / ws) * nWx;
  const int b = win / nW, widx = win - b * nW;
  const int wy = widx / nWx, wx = widx - wy * nWx;
  const float scale = rsqrtf((float)hd);

  // token of window position t (cyclic shift + window partition as index arithmetic, drct.py:482-504)
  auto token_of = [&](int t, int& info) {
    const int py = t >> 3, px = t & 7;
    const int r = wy * ws + py, c = wx * ws + px;                 // coordinates in the shifted image
    int orr = r + p.shift; if (orr >= p.H) orr -= p.H;
    int occ = c + p.shift; if (occ >= p.W) occ -= p.W;
    const int rh = r < p.H - ws ? 0 : (r < p.H - p.shift ? 1 : 2);
    const int rw = c < p.W - ws ? 0 : (c < p.W - p.shift ? 1 : 2);
    info = ((rh * 3 + rw) << 16) | (py << 8) | px;
    return (b * p.H + orr) * p.W + occ;
  };
  const int xrow = tid >> 3, col4 = tid & 7;
  int my_info;
  const int my_tok = token_of(xrow, my_info);                     // every thread knows its own row: no barrier before the loads
  if (tid < 64) {
    int info;
    tok[tid] = token_of(tid, info);
    inf[tid] = info;
  }

  // ---- weights: the head's 3 * HDP virtual rows [q_h | k_h | v_h] as fragments (srad_launch_pack_qkv_frag).  A wave owns 16
  //      virtual columns of a 128-column stage for all 64 tokens and NOBODY else reads their weights, so they go straight
  //      from global memory into MFMA operand registers, three stages ahead: no LDS weight stage, no barrier per stage (as in
  //      mlp_block_kernel).  The loads are unconditional - a load inside a branch makes hipcc's wait-count pass drain every
  //      older load at the join; a wave without columns in a stage reads one 16-byte word per load instead. ----
  constexpr int NSETS = SPLIT ? 4 : 3;
  u32x4 w_reg[NSETS][8];
  const char* const Wh = reinterpret_cast<const char*>(p.w_qkv) + (size_t)h * (NV / 16) * KC * 1024;
  const char* const Wl = reinterpret_cast<const char*>(SPLIT ? p.w_qkv_lo : p.w_qkv) + (size_t)h * (NV / 16) * KC * 1024;
  auto load_w = [&](auto S, u32x4 (&reg)[8]) {
    constexpr int vs = decltype(S)::value < n_vst - 1 ? decltype(S)::value : n_vst - 1;
    constexpr int sc = vs / NPART;
    constexpr int st = sc / KG, kg = sc - st * KG;
    constexpr int nch = KC - kg * 8 < 8 ? KC - kg * 8 : 8;
    const bool live = (st * 8 + wave_s) * 16 < NV;
    const char* const Wp = (vs % NPART) == 1 ? Wl : Wh;
    const char* base = live ? Wp + ((size_t)(st * 8 + wave) * KC + kg * 8) * 1024 + fr * 64 + fq * 16 : Wp;
    const int step = live ? 1024 : 0;
#pragma unroll
    for (int cc = 0; cc < nch; ++cc) reg[cc] = *reinterpret_cast<const u32x4*>(base + cc * step);
  };

  // ---- issue the loads in the order their data is needed (vmcnt retires in issue order): the window's rows of x
  //      (gathered), the staged vectors, the first weight stage; the rest of the weight look-ahead follows the LayerNorm ----
  f32x4 a_reg[KC];
  {
    const char* src = reinterpret_cast<const char*>(p.x) + (size_t)my_tok * p.ldx * 4;
#pragma unroll
    for (int j = 0; j < KC; ++j) a_reg[j] = *reinterpret_cast<const f32x4*>(src + (unsigned)min(j * 32 + col4 * 4, d - 4) * 4u);
  }
  // gamma | beta (320 each) | bias (3 * HDP) | table (225) are contiguous in LDS.  One role per wave, wave-uniform base /
  // stride / count (scalar selects), five clamped 4-byte loads per lane: no per-lane branch, no pointer table in memory.
  typedef const float __attribute__((address_space(1)))* gfloat_p;
  float vq[5];
  int v_dst, v_cnt, v_n;
  {
    gfloat_p src; int stride = 1;
    if (wave_s == 0) { src = (gfloat_p)p.ln_g; v_n = d; v_cnt = 320; v_dst = 0; }
    else if (wave_s == 1) { src = (gfloat_p)p.ln_b; v_n = d; v_cnt = 320; v_dst = 320; }
    else if (wave_s < 5) { src = (gfloat_p)p.b_qkv + (wave_s - 2) * d + h * hd; v_n = hd; v_cnt = HDP; v_dst = 640 + (wave_s - 2) * HDP; }
    else { src = (gfloat_p)p.table + (size_t)(wave_s - 5) * 75 * heads + h; stride = heads; v_n = 75; v_cnt = 75; v_dst = 640 + 3 * HDP + (wave_s - 5) * 75; }
#pragma unroll
    for (int q = 0; q < 5; ++q) vq[q] = src[(size_t)min(lane + 64 * q, v_n - 1) * stride];
  }
  load_w(std::integral_constant<int, 0>{}, w_reg[0]);
  stamp(1);                                                       // prologue loads issued
#pragma unroll
  for (int q = 0; q < 5; ++q) {
    const int e = lane + 64 * q;
    if (e < v_cnt) v_g[v_dst + e] = e < v_n ? vq[q] : 0.f;        // bias entries of the padded head columns are 0
  }
  auto zero_qk_pad = [&]() {
    if constexpr (HDP32 != HDP) {
      // head dims padded to 48 / 80: the last 32-wide k step of q.k^T also covers 16 columns no epilogue writes
      const int row = tid >> 2, part = tid & 3;                   // 128 (q and k) rows x 4 quads of 4 columns
      *reinterpret_cast<bf16x4*>(QKV + row * HS + HDP + 4 * part) = bf16x4{(__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f};
      if constexpr (SPLIT) *reinterpret_cast<bf16x4*>(QKVL + row * HS + HDP + 4 * part) = bf16x4{(__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f};
    }
  };
  if constexpr (!SPLIT) zero_qk_pad();                            // (split-bf16: q | k | v overlay the window tile - after the GEMM)

  stamp(2);                                                       // vectors staged
  // ---- LayerNorm1 from the registers (the 8 lanes of a row hold all its columns) -> bf16 -> XN ----
  {
    float s = 0.f, ss = 0.f;
#pragma unroll
    for (int j = 0; j < KC; ++j) {
      const bool in = j * 32 + col4 * 4 < d;
      const f32x4 v = in ? a_reg[j] : f32x4{0.f, 0.f, 0.f, 0.f};
      s += (v[0] + v[1]) + (v[2] + v[3]);
      ss += (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
    }
    { s = srad_row8_sum(s); ss = srad_row8_sum(ss); }
    const float mu = s / (float)d;
    const float rstd = rsqrtf(fmaxf(ss / (float)d - mu * mu, 0.f) + 1e-5f);
    stamp(3);                                                     // x rows arrived, statistics done
    __syncthreads();                                              // gamma / beta / table / window geometry staged
    stamp(4);
    {
      // relative position bias + 0 / -100 shift mask (drct.py:284-292, 449-470) of this head for every (key, query) pair,
      // once per workgroup: the softmax then adds one float4 per 4 scores instead of unpacking coordinates per score
      const int k = tid >> 3, q0 = (tid & 7) * 8;
      const int ki = inf[k], ky = (ki >> 8) & 0xff, kx = ki & 0xff, kr = ki >> 16;
      float bm[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int qi = inf[q0 + i], qy = (qi >> 8) & 0xff, qx = qi & 0xff, qr = qi >> 16;
        float v = tbl[(qy - ky + 7) * 15 + (qx - kx + 7)];
        if (p.shift > 0 && qr != kr) v += -100.0f;
        bm[i] = v;
      }
      *reinterpret_cast<f32x4*>(BM + k * QA_LDB + q0) = f32x4{bm[0], bm[1], bm[2], bm[3]};
      *reinterpret_cast<f32x4*>(BM + k * QA_LDB + q0 + 4) = f32x4{bm[4], bm[5], bm[6], bm[7]};
    }
#pragma unroll
    for (int j = 0; j < KC; ++j) {
      const int c = j * 32 + col4 * 4;
      const f32x4 g4 = *reinterpret_cast<const f32x4*>(v_g + c);
      const f32x4 b4 = *reinterpret_cast<const f32x4*>(v_b + c);
      const f32x4 v = c < d ? (a_reg[j] - mu) * rstd * g4 + b4 : f32x4{0.f, 0.f, 0.f, 0.f};
      if (p.save_xn && h == 0 && c < d) *reinterpret_cast<f32x4*>(p.save_xn + (size_t)my_tok * d + c) = v;
      bf16x4 hh;
      if constexpr (SPLIT) {
        bf16x4 ll;
        srad_split4(v, hh, ll);
        *reinterpret_cast<bf16x4*>(XNL + xrow * QA_LDX + c) = ll;
      } else {
        hh[0] = (__bf16)v[0]; hh[1] = (__bf16)v[1]; hh[2] = (__bf16)v[2]; hh[3] = (__bf16)v[3];
      }
      if (p.save_xn_h && h == 0 && c < d) *reinterpret_cast<bf16x4*>(p.save_xn_h + (size_t)my_tok * d + c) = hh;
      *reinterpret_cast<bf16x4*>(XN + xrow * QA_LDX + c) = hh;
    }
  }
  stamp(5);                                                       // normalised rows written
  static_for<1, NSETS>([&](auto Q) { load_w(Q, w_reg[decltype(Q)::value]); });   // the rest of the weight look-ahead
  stamp(6);
  __syncthreads();                                                // the normalised window is in LDS
  stamp(7);

  // ---- q|k|v = xn . W^T : 128 virtual columns per stage, transposed result (lane: token fr of row tile t, 4 columns) ----
  f32x4 acc[SPLIT ? NS : 1][4];
  // bias, the attention's scale on q, zero padding -> the bf16 tile(s) the attention reads (and the training saves)
  auto epi_qkv = [&](auto ST, const f32x4 (&a4)[4]) __attribute__((always_inline)) {
    constexpr int st = decltype(ST)::value;
    const bool live = (st * 8 + wave_s) * 16 < NV;
    if (live) {
      const int vc = (st * 8 + wave) * 16 + 4 * fq;               // virtual column of element 0
      const int which = vc / HDP, c = vc - which * HDP;           // HDP % 16 == 0: the wave's 16 columns stay in one slice
      const f32x4 bias = *reinterpret_cast<const f32x4*>(v_bias + which * HDP + c);
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        f32x4 v = a4[t] + bias;
        if (p.save_qkv && c < p.hdp)
          *reinterpret_cast<f32x4*>(p.save_qkv + (size_t)tok[t * 16 + fr] * (3 * heads * p.hdp) + (which * heads + h) * p.hdp + c) = v;
        if (which == 0) v = v * scale;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = c + e < hd ? v[e] : 0.f;
        bf16x4 hh;
        if constexpr (SPLIT) {
          bf16x4 ll;
          srad_split4(v, hh, ll);
          *reinterpret_cast<bf16x4*>(QKVL + (which * 64 + t * 16 + fr) * HS + c) = ll;
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) hh[e] = (__bf16)v[e];
        }
        *reinterpret_cast<bf16x4*>(QKV + (which * 64 + t * 16 + fr) * HS + c) = hh;
        if (p.save_qkv_h && c < p.hp_h)
          *reinterpret_cast<bf16x4*>(p.save_qkv_h + (size_t)tok[t * 16 + fr] * (3 * heads * p.hp_h) + (which * heads + h) * p.hp_h + c) = hh;
      }
    }
  };
  static_for<0, n_vst>([&](auto S) {
    constexpr int vs = decltype(S)::value;
    constexpr int s = vs / NPART, part = vs % NPART;              // split-bf16: part 0 = hi weights, 1 = lo weights
    constexpr int st = s / KG, kg = s - st * KG;
    u32x4 (&reg)[8] = w_reg[vs % NSETS];
    f32x4 (&ac)[4] = acc[SPLIT ? st : 0];
    constexpr int nch = KC - kg * 8 < 8 ? KC - kg * 8 : 8;
    const bool live = (st * 8 + wave_s) * 16 < NV;
    if constexpr (kg == 0 && part == 0) {
#pragma unroll
      for (int t = 0; t < 4; ++t) ac[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    if (live) {
      const __bf16* ar = XN + fr * QA_LDX + kg * 256 + 8 * fq;
      if constexpr (SPLIT) {
#pragma unroll
        for (int cc = 0; cc < nch; ++cc) {                          // one chunk's fragments (both planes with the hi weights) in flight, then its MFMAs
          const bf16x8 b0 = __builtin_bit_cast(bf16x8, reg[cc]);
          bf16x8 af[4];
          [[maybe_unused]] bf16x8 afl[4];
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            if constexpr (part == 0) afl[t] = *reinterpret_cast<const bf16x8*>(ar + XN_E + t * 16 * QA_LDX + cc * 32);
            af[t] = *reinterpret_cast<const bf16x8*>(ar + t * 16 * QA_LDX + cc * 32);
          }
          __builtin_amdgcn_sched_barrier(0);
          if constexpr (part == 0) {
#pragma unroll
            for (int t = 0; t < 4; ++t) ac[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b0, afl[t], ac[t], 0, 0, 0);
          }
#pragma unroll
          for (int t = 0; t < 4; ++t) ac[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b0, af[t], ac[t], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
        }
      } else {
        // the window fragments of two chunks are requested together, THEN multiplied (sched_barrier: the scheduler otherwise
        // sinks each LDS read to just above its MFMA and the LDS latency is paid per MFMA; see mlp_block_kernel)
#pragma unroll
        for (int cc0 = 0; cc0 < nch; cc0 += 2) {
          bf16x8 af[2][4];
#pragma unroll
          for (int g = 0; g < 2; ++g)
            if (cc0 + g < nch) {
#pragma unroll
              for (int t = 0; t < 4; ++t) af[g][t] = *reinterpret_cast<const bf16x8*>(ar + t * 16 * QA_LDX + (cc0 + g) * 32);
            }
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int g = 0; g < 2; ++g)
            if (cc0 + g < nch) {
              const bf16x8 b0 = __builtin_bit_cast(bf16x8, reg[cc0 + g]);
#pragma unroll
              for (int t = 0; t < 4; ++t) ac[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b0, af[g][t], ac[t], 0, 0, 0);
            }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
    load_w(std::integral_constant<int, vs + NSETS>{}, reg);
    if constexpr (!SPLIT && kg == KG - 1) epi_qkv(std::integral_constant<int, st>{}, ac);
  });
  if constexpr (SPLIT) {
    __syncthreads();                                              // every wave is done reading the window tile: q | k | v go over it
    zero_qk_pad();
    static_for<0, NS>([&](auto ST) { epi_qkv(ST, acc[decltype(ST)::value]); });
  }
  stamp(8);                                                       // this wave's q | k | v columns done
  __syncthreads();                                                // q, k, v complete
  stamp(9);

  const __bf16* Qs = QKV;
  const __bf16* Ks = QKV + 64 * HS;
  const __bf16* Vs = QKV + 2 * 64 * HS;

  // ---- S = q k^T, bias, mask, softmax on all 8 waves: wave = (query row tile rt, key half kh), 16 queries x 32 keys each;
  //      the two halves of a row exchange their maxima through LDS, so the probabilities are exp(s - row max) as before ----
  {
    const int kh = chh;
    f32x4 sc[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) sc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kk = 0; kk < HDP32; kk += 32) {
      const bf16x8 a = *reinterpret_cast<const bf16x8*>(Qs + (rt * 16 + fr) * HS + kk + 8 * fq);
      [[maybe_unused]] bf16x8 al;
      if constexpr (SPLIT) al = *reinterpret_cast<const bf16x8*>(Qs + QKV_E + (rt * 16 + fr) * HS + kk + 8 * fq);
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const bf16x8 bb = *reinterpret_cast<const bf16x8*>(Ks + ((2 * kh + j) * 16 + fr) * HS + kk + 8 * fq);
        if constexpr (SPLIT) {
          const bf16x8 bl = *reinterpret_cast<const bf16x8*>(Ks + QKV_E + ((2 * kh + j) * 16 + fr) * HS + kk + 8 * fq);
          sc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bb, sc[j], 0, 0, 0);
          sc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, bl, sc[j], 0, 0, 0);
        }
        sc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, bb, sc[j], 0, 0, 0);
      }
    }
    // lane: queries rt*16 + 4 fq + e (e = 0..3), keys (2 kh + j) * 16 + fr
    float mx[4];
#pragma unroll
    for (int j = 0; j < 2; ++j) sc[j] += *reinterpret_cast<const f32x4*>(BM + ((2 * kh + j) * 16 + fr) * QA_LDB + rt * 16 + 4 * fq);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      mx[e] = srad_row16_max(fmaxf(sc[0][e], sc[1][e]));
      if (fr == 0) mxs[kh * 64 + rt * 16 + fq * 4 + e] = mx[e];
    }
    __syncthreads();
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int row = rt * 16 + fq * 4 + e;
      const float m = fmaxf(mx[e], mxs[(kh ^ 1) * 64 + row]);
      const float p0 = __expf(sc[0][e] - m), p1 = __expf(sc[1][e] - m);
      const float rs = srad_row16_sum(p0 + p1);
      if (fr == 0) lsum[kh * 64 + row] = rs;
      const __bf16 p0h = (__bf16)p0, p1h = (__bf16)p1;
      Ps[row * QA_LDP + (2 * kh) * 16 + fr] = p0h;
      Ps[row * QA_LDP + (2 * kh + 1) * 16 + fr] = p1h;
      if constexpr (SPLIT) {
        PsL[row * QA_LDP + (2 * kh) * 16 + fr] = (__bf16)(p0 - (float)p0h);
        PsL[row * QA_LDP + (2 * kh + 1) * 16 + fr] = (__bf16)(p1 - (float)p1h);
      }
    }
  }
  stamp(10);                                                      // scores + softmax
  __syncthreads();
  stamp(11);

  // ---- O = P V (transposed result: lane owns token fr of row tile rt, 4 consecutive output columns) ----
  {
    const int tq = fr >> 2, tp = fr & 3;
#pragma unroll
    for (int j = 0; j < HDT; ++j) {
      if ((j & 1) != chh) continue;                               // output column tiles split over the two wave groups
      f32x4 o = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kk = 0; kk < 64; kk += 32) {
        const bf16x8 pa = *reinterpret_cast<const bf16x8*>(Ps + (rt * 16 + fr) * QA_LDP + kk + 8 * fq);
        typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;
        const __bf16* r0 = Vs + (kk + 8 * fq + tq) * HS + j * 16 + 4 * tp;
        const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(r0));
        const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(r0 + 4 * HS));
        bf16x8 vb;
#pragma unroll
        for (int e = 0; e < 4; ++e) { vb[e] = lo[e]; vb[4 + e] = hi[e]; }
        if constexpr (SPLIT) {                                      // v_lo . p_hi and v_hi . p_lo first (the small terms)
          const bf16x4 llo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(r0 + QKV_E));
          const bf16x4 lhi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(r0 + QKV_E + 4 * HS));
          bf16x8 vl;
#pragma unroll
          for (int e = 0; e < 4; ++e) { vl[e] = llo[e]; vl[4 + e] = lhi[e]; }
          const bf16x8 pl = *reinterpret_cast<const bf16x8*>(PsL + (rt * 16 + fr) * QA_LDP + kk + 8 * fq);
          o = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vl, pa, o, 0, 0, 0);
          o = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vb, pl, o, 0, 0, 0);
        }
        o = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vb, pa, o, 0, 0, 0);
      }
      const int row = rt * 16 + fr;
      const float inv = 1.0f / (lsum[row] + lsum[64 + row]);
      if (p.out_h) {                                              // bf16 hand-off to mlp_block
        __bf16* dst = p.out_h + (size_t)tok[row] * p.ld_out + h * hd;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int c = j * 16 + 4 * fq + e;
          if (c < hd) dst[c] = (__bf16)(o[e] * inv);
        }
      } else {
        float* dst = p.out + (size_t)tok[row] * p.ld_out + h * hd;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int c = j * 16 + 4 * fq + e;
          if (c < hd) dst[c] = o[e] * inv;
        }
      }
    }
  }
  stamp(15);
}

template <int HDT, int KC, bool STAMP = false, bool SPLIT = false>
int launch_qa(const QkvAttnParams& p, hipStream_t stream) {
  constexpr int HDP = 16 * HDT, HS = ((HDP + 31) & ~31) + 16;
  constexpr int XN_E = 64 * (KC * 32 + 16), QP_E = 3 * 64 * HS + 64 * QA_LDP;
  constexpr size_t lds = (size_t)(SPLIT ? 2 * (XN_E > QP_E ? XN_E : QP_E) : XN_E + QP_E) * 2 +
                         (size_t)(640 + 3 * HDP + 232 + 256 + 64 * QA_LDB) * sizeof(float) + 128 * sizeof(int);
  static_assert(lds <= 160 * 1024, "qkv_attn: LDS budget");
  auto kern = qkv_attn_kernel<HDT, KC, STAMP, SPLIT>;
  static SradOncePerDevice configured;
  if (configured.need()) {
    SRAD_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    configured.done();
  }
  const double T = (double)p.B * p.H * p.W;
  const double flops = 2.0 * T * 3.0 * p.d * p.d + 4.0 * T * 64.0 * p.d;
  const double bytes = 4.0 * T * p.d * 2 + 2.0 * 3.0 * p.d * p.d;
  SradProfScope prof(stream, SRAD_K_QKV_ATTN, flops, bytes);
  const int nW = (p.H / 8) * (p.W / 8);
  hipLaunchKernelGGL(kern, dim3(p.B * nW, p.heads), dim3(512), lds, stream, p);
  SRAD_CHECK_HIP(hipGetLastError());
  return SRAD_OK;
}

// ------------------------------------------------------------------------------------------
// LayerNorm1 + the qkv Linear for the 64 x 64-window attention (BASELINE config C5: 65536 tokens per tile), whose q | k | v
// it leaves as the bf16 operands that kernel stages as they are: [T][3][heads][hdp], q times the attention's scale (incl.
// log2 e), padding columns 0, column head_dim of the v slices 1.  The first half of qkv_attn_kernel without the
// attention: 64 token rows per workgroup, normalised once into LDS, then EVERY head's [q | k | v] weight fragments
// stream through the waves' registers (a wave owns 16 virtual columns of a head for all 64 rows) - a row tile meets the
// whole 3 d x d weight, where the tiled GEMM launched one 64 x 64 tile per workgroup, each re-reading and re-normalising
// its rows for 1.5 MFLOP (130 TFLOP/s at this size).
// ------------------------------------------------------------------------------------------
// SPLIT (split-bf16): the normalised rows as a hi and a lo bf16 plane, the weights streamed twice (hi pack against both planes, lo
// pack against the hi plane), fp32 output (LnQkvParams::qkv_f).
template <int HDT, int KC, int HEADS, bool SPLIT = false>
__global__ __launch_bounds__(512) void ln_qkv_kernel(const LnQkvParams p) {
  constexpr int KG = (KC + 7) / 8;
  constexpr int LDX = KC * 32 + 16;
  constexpr int HDP = 16 * HDT;
  constexpr int NV = 3 * HDP;
  constexpr int NS = (NV + 127) / 128;
  constexpr int n_stages = NS * KG;                              // per head
  constexpr int n_all = HEADS * n_stages;
  constexpr int NPART = SPLIT ? 2 : 1;                           // weight streams per stage (hi | hi, lo)
  constexpr int n_vall = n_all * NPART;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  __bf16* XN = reinterpret_cast<__bf16*>(smem);                 // [64][LDX]
  [[maybe_unused]] __bf16* XNL = XN + 64 * LDX;                 // split-bf16: the lo plane
  float* v_g = reinterpret_cast<float*>(XN + (SPLIT ? 2 : 1) * 64 * LDX);   // [320] gamma
  float* v_b = v_g + 320;                                       // [320] beta
  float* v_bias = v_b + 320;                                    // [HEADS][3][HDP] bias (0 in padding)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 15, fq = lane >> 4;
  const int wave_s = __builtin_amdgcn_readfirstlane(wave);
  const int d = p.d, hd = d / HEADS, hdp = p.hdp;
  int tile_i = blockIdx.x;                                      // XCD affinity with mlp_block's row tiles (see qkv_attn_kernel)
  if ((gridDim.x & 7) == 0) tile_i = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
  const int m0 = tile_i * 64;
  const int xrow = tid >> 3, col4 = tid & 7;

  constexpr int NSETS = 3;
  u32x4 w_reg[NSETS][8];
  auto load_w = [&](auto S, u32x4 (&reg)[8]) __attribute__((always_inline)) {
    constexpr int va = decltype(S)::value < n_vall - 1 ? decltype(S)::value : n_vall - 1;
    constexpr int sa = va / NPART;
    constexpr int hh = sa / n_stages, sc = sa - hh * n_stages;
    constexpr int st = sc / KG, kg = sc - st * KG;
    constexpr int nch = KC - kg * 8 < 8 ? KC - kg * 8 : 8;
    const char* const Wh = reinterpret_cast<const char*>((va % NPART) == 1 ? p.w_qkv_lo : p.w_qkv) + (size_t)hh * (NV / 16) * KC * 1024;
    const bool live = (st * 8 + wave_s) * 16 < NV;
    const char* base = live ? Wh + ((size_t)(st * 8 + wave) * KC + kg * 8) * 1024 + fr * 64 + fq * 16 : Wh;
    const int step = live ? 1024 : 0;
#pragma unroll
    for (int cc = 0; cc < nch; ++cc) reg[cc] = *reinterpret_cast<const u32x4*>(base + cc * step);
  };

  // loads in the order their data is needed: the rows, the vectors, the first weight stage
  f32x4 a_reg[KC];
  {
    const char* src = reinterpret_cast<const char*>(p.x) + (size_t)(m0 + xrow) * p.ldx * 4;
#pragma unroll
    for (int j = 0; j < KC; ++j) a_reg[j] = *reinterpret_cast<const f32x4*>(src + (unsigned)min(j * 32 + col4 * 4, d - 4) * 4u);
  }
  typedef const float __attribute__((address_space(1)))* gfloat_p;
  constexpr int NBIAS = HEADS * 3 * HDP;                          // <= 768
  float gq = 0.f, bq = 0.f, biasq[2];
  {
    const int e = min(tid, d - 1);
    if (tid < 320) { gq = ((gfloat_p)p.ln_g)[e]; bq = ((gfloat_p)p.ln_b)[e]; }
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int idx = min(tid + 512 * q, NBIAS - 1);
      const int slice = idx / HDP, c = idx - slice * HDP, hh = slice / 3, which = slice - 3 * hh;
      biasq[q] = ((gfloat_p)p.b_qkv)[which * d + hh * hd + min(c, hd - 1)];
      if (c >= hd) biasq[q] = 0.f;
    }
  }
  load_w(std::integral_constant<int, 0>{}, w_reg[0]);
  if (tid < 320) { v_g[tid] = tid < d ? gq : 0.f; v_b[tid] = tid < d ? bq : 0.f; }
#pragma unroll
  for (int q = 0; q < 2; ++q)
    if (tid + 512 * q < NBIAS) v_bias[tid + 512 * q] = biasq[q];

  // LayerNorm1 from the registers (the 8 lanes of a row hold all its columns) -> bf16 -> XN
  {
    float s = 0.f, ss = 0.f;
#pragma unroll
    for (int j = 0; j < KC; ++j) {
      const bool in = j * 32 + col4 * 4 < d;
      const f32x4 v = in ? a_reg[j] : f32x4{0.f, 0.f, 0.f, 0.f};
      s += (v[0] + v[1]) + (v[2] + v[3]);
      ss += (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
    }
    { s = srad_row8_sum(s); ss = srad_row8_sum(ss); }
    const float mu = s / (float)d;
    const float rstd = rsqrtf(fmaxf(ss / (float)d - mu * mu, 0.f) + 1e-5f);
    __syncthreads();                                              // gamma / beta / bias staged
#pragma unroll
    for (int j = 0; j < KC; ++j) {
      const int c = j * 32 + col4 * 4;
      const f32x4 g4 = *reinterpret_cast<const f32x4*>(v_g + min(c, 316));
      const f32x4 b4 = *reinterpret_cast<const f32x4*>(v_b + min(c, 316));
      const f32x4 v = c < d ? (a_reg[j] - mu) * rstd * g4 + b4 : f32x4{0.f, 0.f, 0.f, 0.f};
      bf16x4 hh;
      if constexpr (SPLIT) {
        bf16x4 ll;
        srad_split4(v, hh, ll);
        *reinterpret_cast<bf16x4*>(XNL + xrow * LDX + c) = ll;
      } else {
        hh[0] = (__bf16)v[0]; hh[1] = (__bf16)v[1]; hh[2] = (__bf16)v[2]; hh[3] = (__bf16)v[3];
      }
      *reinterpret_cast<bf16x4*>(XN + xrow * LDX + c) = hh;
    }
  }
  static_for<1, NSETS>([&](auto Q) { load_w(Q, w_reg[decltype(Q)::value]); });
  __syncthreads();                                                // the normalised rows are in LDS

  // q|k|v = xn . W^T, head after head: 128 virtual columns per stage, transposed result (lane: token fr of row tile t, 4 columns)
  f32x4 acc[4];
  const int ldq = 3 * HEADS * hdp;
  static_for<0, n_vall>([&](auto S) {
    constexpr int va = decltype(S)::value;
    constexpr int sa = va / NPART, part = va % NPART;               // split-bf16: part 0 = hi weights, 1 = lo weights
    constexpr int hh = sa / n_stages, s = sa - hh * n_stages;
    constexpr int st = s / KG, kg = s - st * KG;
    u32x4 (&reg)[8] = w_reg[va % NSETS];
    constexpr int nch = KC - kg * 8 < 8 ? KC - kg * 8 : 8;
    const bool live = (st * 8 + wave_s) * 16 < NV;
    if constexpr (kg == 0 && part == 0) {
#pragma unroll
      for (int t = 0; t < 4; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    if (live) {
      const __bf16* ar = XN + fr * LDX + kg * 256 + 8 * fq;
#pragma unroll
      for (int cc0 = 0; cc0 < nch; cc0 += 2) {                     // two chunks' fragments in flight, then their MFMAs (see qkv_attn_kernel)
        bf16x8 af[2][4];
        [[maybe_unused]] bf16x8 afl[2][4];
#pragma unroll
        for (int g = 0; g < 2; ++g)
          if (cc0 + g < nch) {
#pragma unroll
            for (int t = 0; t < 4; ++t) {
              af[g][t] = *reinterpret_cast<const bf16x8*>(ar + t * 16 * LDX + (cc0 + g) * 32);
              if constexpr (SPLIT && part == 0) afl[g][t] = *reinterpret_cast<const bf16x8*>(ar + 64 * LDX + t * 16 * LDX + (cc0 + g) * 32);
            }
          }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int g = 0; g < 2; ++g)
          if (cc0 + g < nch) {
            const bf16x8 b0 = __builtin_bit_cast(bf16x8, reg[cc0 + g]);
            if constexpr (SPLIT && part == 0) {
#pragma unroll
              for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b0, afl[g][t], acc[t], 0, 0, 0);
            }
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b0, af[g][t], acc[t], 0, 0, 0);
          }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    load_w(std::integral_constant<int, va + NSETS>{}, reg);
    if constexpr (kg == KG - 1 && part == NPART - 1) {
      if (live) {
        const int vc = (st * 8 + wave) * 16 + 4 * fq;             // virtual column of element 0
        const int which = vc / HDP, c = vc - which * HDP;         // HDP % 16 == 0: the wave's 16 columns stay in one slice
        const f32x4 bias = *reinterpret_cast<const f32x4*>(v_bias + (hh * 3 + which) * HDP + c);
        if (c < hdp) {
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            f32x4 v = acc[t] + bias;
            if constexpr (SPLIT) {                                  // plain fp32 q | k | v: the attention kernel scales and splits while staging
              *reinterpret_cast<f32x4*>(p.qkv_f + (size_t)(m0 + t * 16 + fr) * ldq + (which * HEADS + hh) * hdp + c) = v;
              continue;
            }
            if (which == 0) v = v * p.qscale;
            bf16x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = (__bf16)(c + e < hd ? v[e] : ((which == 2 && c + e == hd) ? 1.f : 0.f));
            *reinterpret_cast<bf16x4*>(p.qkv_h + (size_t)(m0 + t * 16 + fr) * ldq + (which * HEADS + hh) * hdp + c) = o;
          }
        }
      }
    }
  });
}

template <int HDT, int KC, int HEADS, bool SPLIT = false>
int launch_ln_qkv(const LnQkvParams& p, hipStream_t stream) {
  constexpr size_t lds = (size_t)64 * (KC * 32 + 16) * 2 * (SPLIT ? 2 : 1) + (size_t)(640 + HEADS * 3 * 16 * HDT) * sizeof(float);
  static_assert(lds <= 160 * 1024, "ln_qkv: LDS budget");
  auto kern = ln_qkv_kernel<HDT, KC, HEADS, SPLIT>;
  static SradOncePerDevice configured;
  if (configured.need()) {
    SRAD_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    configured.done();
  }
  SradProfScope prof(stream, SRAD_K_LN_QKV, 2.0 * p.M * 3.0 * p.d * p.d, 4.0 * p.M * p.d + 2.0 * p.M * 3.0 * p.d + 2.0 * 3.0 * p.d * p.d);
  hipLaunchKernelGGL(kern, dim3(p.M / 64), dim3(512), lds, stream, p);
  SRAD_CHECK_HIP(hipGetLastError());
  return SRAD_OK;
}

// (head tiles, k chunks) of DRCT-L's five Swin blocks: d = 180/212/244/276/308, heads 6/4/2/6/4
#define SRAD_QA_CFGS(X) X(2, 6) X(4, 7) X(8, 8) X(3, 9) X(5, 10)
#define SRAD_LQ_CFGS(X) X(2, 6, 6) X(4, 7, 4) X(8, 8, 2) X(3, 9, 6) X(5, 10, 4)
}  // namespace

bool srad_ln_qkv_supported(int prec, int M, int d, int heads) {
  if ((prec != SRAD_PREC_BF16 && prec != SRAD_PREC_BF16X3) || M <= 0 || M % 64 || d % 4 || d > 320 || d < 32 || heads < 1 || d % heads) return false;
  const int hdt = (d / heads + 15) / 16, kc = (d + 31) / 32;
#define X(a, b, c) if (hdt == a && kc == b && heads == c) return true;
  SRAD_LQ_CFGS(X)
#undef X
  return false;
}

int srad_launch_ln_qkv(const LnQkvParams& p, hipStream_t stream) {
  SRAD_REQUIRE(srad_ln_qkv_supported(SRAD_PREC_BF16, p.M, p.d, p.heads), "ln_qkv: unsupported shape M=%d d=%d heads=%d", p.M, p.d, p.heads);
  SRAD_REQUIRE(p.x && p.ln_g && p.ln_b && p.w_qkv && p.b_qkv && (p.qkv_h || (p.qkv_f && p.w_qkv_lo)) && !(p.qkv_h && p.qkv_f), "ln_qkv: null argument");
  SRAD_REQUIRE((p.ldx & 3) == 0 && ((uintptr_t)p.x & 15) == 0 && ((uintptr_t)p.qkv_h & 7) == 0 && ((uintptr_t)p.qkv_f & 15) == 0 && p.hdp % 4 == 0 && p.hdp >= p.d / p.heads &&
                   p.hdp <= 16 * ((p.d / p.heads + 15) / 16),
               "ln_qkv: x rows must be float4-addressable, the head slots 4-column multiples within the head's 16-column tiles");
  const int hdt = (p.d / p.heads + 15) / 16, kc = (p.d + 31) / 32;
  if (p.qkv_f) {                                                  // split-bf16
#define X(a, b, c) if (hdt == a && kc == b && p.heads == c) return launch_ln_qkv<a, b, c, true>(p, stream);
    SRAD_LQ_CFGS(X)
#undef X
  }
#define X(a, b, c) if (hdt == a && kc == b && p.heads == c) return launch_ln_qkv<a, b, c>(p, stream);
  SRAD_LQ_CFGS(X)
#undef X
  return srad_set_error(SRAD_ERR_ARG, "ln_qkv: no kernel instance for d=%d heads=%d", p.d, p.heads);
}

bool srad_qkv_attn_supported(int prec, int ws, int H, int W, int d, int heads) {
  if ((prec != SRAD_PREC_BF16 && prec != SRAD_PREC_BF16X3) || ws != 8 || H % 8 || W % 8 || d % 4 || d > 320 || d < 32 || heads < 1 || d % heads) return false;
  const int hdt = (d / heads + 15) / 16, kc = (d + 31) / 32;
#define X(a, b) if (hdt == a && kc == b) return true;
  SRAD_QA_CFGS(X)
#undef X
  return false;
}

int srad_launch_qkv_attn(const QkvAttnParams& p, hipStream_t stream) {
  SRAD_REQUIRE(srad_qkv_attn_supported(SRAD_PREC_BF16, 8, p.H, p.W, p.d, p.heads), "qkv_attn: unsupported shape d=%d heads=%d %dx%d", p.d, p.heads, p.H, p.W);
  SRAD_REQUIRE((p.ldx & 3) == 0 && ((uintptr_t)p.x & 15) == 0, "qkv_attn: x rows must be float4-addressable");
  SRAD_REQUIRE(p.shift >= 0 && p.shift < 8, "qkv_attn: shift %d must be in [0, 8)", p.shift);
  const int hdt = (p.d / p.heads + 15) / 16, kc = (p.d + 31) / 32;
  if (p.split) {                                                  // split-bf16 (inference)
    SRAD_REQUIRE(p.w_qkv_lo && p.out && !p.out_h, "qkv_attn (split-bf16): needs the lo weight pack and an fp32 output");
    SRAD_REQUIRE(!p.stamps && !p.save_xn && !p.save_xn_h && !p.save_qkv && !p.save_qkv_h, "qkv_attn (split-bf16): inference only");
#define X(a, b) if (hdt == a && kc == b) return launch_qa<a, b, false, true>(p, stream);
    SRAD_QA_CFGS(X)
#undef X
  }
  if (p.stamps) {                                                 // diagnostic build
#define X(a, b) if (hdt == a && kc == b) return launch_qa<a, b, true>(p, stream);
    SRAD_QA_CFGS(X)
#undef X
  }
#define X(a, b) if (hdt == a && kc == b) return launch_qa<a, b>(p, stream);
  SRAD_QA_CFGS(X)
#undef X
  return srad_set_error(SRAD_ERR_ARG, "qkv_attn: no kernel instance for head tiles %d, k chunks %d", hdt, kc);
}
