// kernels_attn.hip - fused shifted-window attention for DRCT on gfx950
// (reference src/drct.py: WindowAttention.forward 271-302, window_partition/reverse 193-220,
//  cyclic shift 482-504, calculate_mask 449-470, relative_position_index 250-260).
//
// One workgroup (4 waves) = one (window, head, 64-query tile).  It gathers the window's tokens
// straight from the raster-ordered q/k/v tensor (cyclic shift and window partition are index
// arithmetic, never copies), streams the window's keys in chunks of 64 with an online softmax
// (so N = ws^2 of 4 ... 4096 tokens all take the same path), adds the relative-position bias
// and the 0/-100 shift mask computed on the fly, and scatters P.V back to raster order.
//   S^T = k (q*scale)^T : MFMA with swapped operands (A = K chunk, B = Q slab, both head-dim contiguous in LDS):
//                         a lane then holds 16 scores of ONE query
//   softmax             : per-lane scalars + two cross-row shuffles per reduction
//   O^T += V^T P^T      : P^T is already the B operand in registers (never through LDS); V stays row-major in LDS
//                         and is read as the A operand with ds_read_b64_tr_b16 (hardware transpose read)
//
// Input layout ("head-padded"): row t = [q | k | v], each [heads][hdp] floats, hdp = head_dim
// rounded up to a multiple of 4 so every (token, head) slice is float4-addressable.  The QKV GEMM
// epilogue writes this layout directly (GemmParams::hsplit_*); pad columns are never read as data and need no
// initialisation.
#include "srad_common.h"
#include <cstdlib>

namespace {

template <int PREC> struct AT;
// PAD (bf16): Q / K / V rows of 16 x odd elements - 2 (mod 4) sixteen-byte units for the lane groups ds_read_b128 is served in
// (kernels_conv80.hip has the derivation) AND an odd multiple of 32 bytes for the transposing reads of V (two groups of 32
// lanes, eight consecutive rows x 32 bytes per group = every bank once).  The head tiles are 32 / 64 / 96 / 128 wide: + 16 each.
template <> struct AT<SRAD_PREC_BF16> { using type = __bf16; static constexpr int PAD = 16; };
template <> struct AT<SRAD_PREC_F32>  { using type = float;  static constexpr int PAD = 4; };
// split-bf16 (SRAD_PREC_BF16X3): Q, K, V tiles as a hi and a lo bf16 plane each (the lo planes behind the three hi tiles), the
// probabilities split in registers, every product three bf16 MFMAs (hi.hi + hi.lo + lo.hi).  Input fp32 (head-padded), output fp32.
template <> struct AT<SRAD_PREC_BF16X3> { using type = __bf16; static constexpr int PAD = 16; };

template <int PREC>
__device__ __forceinline__ void store4(typename AT<PREC>::type* dst, f32x4 v) {
  if constexpr (PREC == SRAD_PREC_BF16) {
    bf16x4 h;
    h[0] = (__bf16)v[0]; h[1] = (__bf16)v[1]; h[2] = (__bf16)v[2]; h[3] = (__bf16)v[3];
    *reinterpret_cast<bf16x4*>(dst) = h;
  } else {
    *reinterpret_cast<f32x4*>(dst) = v;
  }
}

// TBL_LDS (bias table staged in LDS, every window size up to 64) is a template flag: as a run-time flag the
// sixteen look-ups per lane and chunk became branchy flat loads, each waited for before the next.
// NW = waves per workgroup = 16-query slabs per workgroup: 4 for windows of up to 64 tokens, 8 above (the K / V
// chunk and the bias table in LDS are then shared by 128 queries and two waves per SIMD hide each other's latency).
// FULL: the window's token count is a multiple of 64, no key of a chunk needs masking out.
// ROW64 (bf16, 64 x 64 windows = BASELINE config C5, 4096-token windows): a 64-key chunk is exactly one ROW of the window,
// so the key's y and its shift-mask row region are chunk constants and its column region depends only on the 16-key
// tile: the relative-position bias of a lane's 16 keys is 16 loads at constant offsets from one per-chunk address and
// the 0 / -100 mask is one per-tile addend - no per-score index arithmetic.  Scores live in the log2 domain (q pre-scaled
// by log2 e, table staged x log2 e), so the probabilities are bare v_exp_f32.  The softmax was the bound of this kernel
// (~14 VALU instructions per score against 0.5 - 2 MFMA issue slots); this path needs ~6.
// QH (ROW64 only): q | k | v arrive as bf16, already scaled / zero-padded / with V's ones column (the QKV GEMM's epilogue did
// that ONCE per token; staged from fp32 every one of a window's 32 workgroups converted and masked all 4096 keys again -
// a quarter of this VALU-bound kernel's vector instructions): a chunk is staged with 8-byte loads and plain LDS stores.
template <int PREC, int NT_O, bool TBL_LDS, int NW, bool FULL, bool ROW64 = false, bool QH = false>
__global__ __launch_bounds__(NW * 64) void window_attn_kernel(const AttnParams p) {
  static_assert(!QH || (ROW64 && PREC == SRAD_PREC_BF16), "bf16 input: the one-row-per-chunk bf16 path only");
  static_assert(!ROW64 || PREC != SRAD_PREC_F32, "the one-row-per-chunk path stages bf16 tiles");
  using T = typename AT<PREC>::type;
  constexpr int PAD = AT<PREC>::PAD;
  constexpr int HDP = NT_O * 16;
  constexpr int HS = HDP + PAD;      // Q/K/V row stride (elements)
  constexpr int V4R = HDP / 4;       // float4 per staged row
  constexpr int NT = NW * 64;        // threads
  constexpr int BQ = NW * 16;        // query rows per workgroup
  constexpr int NLV = 64 * V4R / NT; // float4 per thread per 64-row K / V chunk
  constexpr int NLQ = BQ * V4R / NT; // float4 per thread for the query tile
  static_assert(NLV >= 1 && NLV * NT == 64 * V4R, "K / V chunk must divide over the threads");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  T* Qs = reinterpret_cast<T*>(smem);
  T* Ks = Qs + BQ * HS;
  T* Vs = Ks + 64 * HS;
  constexpr bool X3 = PREC == SRAD_PREC_BF16X3;
  constexpr int LOFF = (BQ + 128) * HS;                  // lo plane of a tile = its hi plane + LOFF elements (split-bf16)
  int* tokq = reinterpret_cast<int*>(Vs + 64 * HS + (X3 ? LOFF : 0));
  int* infq = tokq + BQ;             // packed (region << 24) | (py * (2 ws - 1) + px)
  int* tokk = infq + BQ;             // [3][64]: key chunks kc, kc + 1 (loads in flight), kc + 2 (being computed)
  int* infk = tokk + 3 * 64;         // [3][64]
  float* tbl = reinterpret_cast<float*>(infk + 3 * 64);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 15, fq = lane >> 4;
  const int ws = p.ws, N = ws * ws, d = p.d, heads = p.heads, hd = d / heads, hdp = p.hdp;
  const int ldq = 3 * heads * hdp;
  const int nWx = p.W / ws, nW = (p.H / ws) * nWx;
  int h = blockIdx.y, win = blockIdx.z, qchunk = blockIdx.x;
  if constexpr (ROW64) {
    // all query chunks of a (window, head) on ONE XCD (workgroup L runs on XCD L % 8): its 0.5 MB of K / V is then fetched into
    // one L2 instead of all eight (a window's 32 workgroups were dealt over every XCD: 8x the L2 fills)
    const int units = gridDim.y * gridDim.z, nq = gridDim.x;
    if ((units & 7) == 0 && (nq & 7) == 0) {
      const int L = blockIdx.x + nq * (blockIdx.y + gridDim.y * blockIdx.z);
      const int s = L >> 3, u = (s / nq) * 8 + (L & 7);
      qchunk = s - (s / nq) * nq;
      win = u / (int)gridDim.y;
      h = u - win * (int)gridDim.y;
    }
  }
  const int b = win / nW, widx = win - b * nW;
  const int wy = widx / nWx, wx = widx - wy * nWx;
  const int q0 = qchunk * BQ;
  constexpr float LOG2E = 1.4426950408889634f;
  const float scale = rsqrtf((float)hd) * (ROW64 ? LOG2E : 1.0f);
  const int tw = 2 * ws - 1;

  auto fexp = [](float x) -> float {     // bf16 mode: v_exp_f32 (2 ulp); fp32 / split-bf16 (parity) modes: libm expf
    if constexpr (PREC == SRAD_PREC_BF16) return __expf(x); else return expf(x);
  };
  auto token_info = [&](int pos, int& tok, int& inf) {
    const int py = pos / ws, px = pos - py * ws;
    const int r = wy * ws + py, c = wx * ws + px;              // coordinates in the shifted image
    int orr = r + p.shift; if (orr >= p.H) orr -= p.H;
    int occ = c + p.shift; if (occ >= p.W) occ -= p.W;
    tok = (b * p.H + orr) * p.W + occ;
    const int rh = r < p.H - ws ? 0 : (r < p.H - p.shift ? 1 : 2);
    const int rw = c < p.W - ws ? 0 : (c < p.W - p.shift ? 1 : 2);
    inf = ((rh * 3 + rw) << 24) | (py * tw + px);        // shift-mask region | linear position in the bias table
  };

  // tokens of the query tile (threads 0 .. BQ-1) and of the first two key chunks (the next 128 threads)
  if (tid < BQ + 128) {
    const int pos = tid < BQ ? q0 + tid : tid - BQ;
    int tok = 0, inf = 0;
    if (pos < N) token_info(pos, tok, inf);
    if (tid < BQ) { tokq[tid] = tok; infq[tid] = inf; }
    else { tokk[tid - BQ] = tok; infk[tid - BQ] = inf; }
  }
  // ROW64: the 128 queries of a workgroup are two rows of the window (qy0, qy0 + 1) and chunk kc is key row kc, so a chunk
  // indexes exactly two rows of the 127 x 127 bias table, R0(kc) = qy0 - kc + 63 and R0(kc) + 1 = R0(kc - 1): ONE new row
  // per chunk.  The rows slide through a four-slot ring in LDS (slot = R & 3, 128 floats each) instead of the whole table
  // (64.5 KB) being resident: 2 - 3 workgroups fit a CU, which is what hides this kernel's per-chunk latency chain.
  const int r64_qy0 = q0 / 64;
  auto r64_row_load = [&](int R) -> float {                       // threads 0 .. 126: column tid of table row R (clamped)
    const int Rc = min(max(R, 0), 126), cc = min(tid, 126);
    return p.table[(size_t)(Rc * 127 + cc) * heads + h] * LOG2E;
  };
  float r64_next = 0.f;                                           // row of the NEXT chunk, in flight
  if constexpr (ROW64) {
    if (tid < 128) {
      tbl[((r64_qy0 + 64) & 3) * 128 + tid] = r64_row_load(r64_qy0 + 64);      // R0(0) + 1
      tbl[((r64_qy0 + 63) & 3) * 128 + tid] = r64_row_load(r64_qy0 + 63);      // R0(0)
      r64_next = r64_row_load(r64_qy0 + 62);                                   // R0(1)
    }
  } else if constexpr (TBL_LDS) {
    for (int i = tid; i < tw * tw; i += NT) tbl[i] = p.table[(size_t)i * heads + h];
  }
  __syncthreads();

  // All global loads are unconditional on clamped addresses and masked by selects afterwards:
  // a load inside a per-element branch is waited for before the next one issues.
  typedef typename std::conditional<QH, u32x2, f32x4>::type ld_t;     // four columns of a staged row
  ld_t qv[NLQ], kv[NLV], vv[NLV];
  auto load_tile = [&](const int* toks, int which, auto& dst) {
    constexpr int NL = sizeof(dst) / sizeof(ld_t);
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      const int idx = tid + NT * i;
      const int row = idx / V4R, c = (idx - row * V4R) * 4;
      const size_t off = (size_t)toks[row] * ldq + (which * heads + h) * hdp + min(c, hdp - 4);
      if constexpr (QH) dst[i] = *reinterpret_cast<const u32x2*>(p.qkv_h + off);
      else dst[i] = *reinterpret_cast<const f32x4*>(p.qkv + off);
    }
  };
  auto store_tile = [&](T* base, int first, float mul, const auto& src, bool ones_col = false) {
    constexpr int NL = sizeof(src) / sizeof(ld_t);
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      const int idx = tid + NT * i;
      const int row = idx / V4R, c = (idx - row * V4R) * 4;
      if constexpr (QH) {                                  // ready made: only the tile columns past the stored slot are cleared
        *reinterpret_cast<u32x2*>(base + row * HS + c) = c < hdp ? src[i] : u32x2{0u, 0u};
        continue;
      } else {
      // selects, not a multiply by 0: the pad columns of the head-padded rows (and rows past N) may hold anything,
      // NaN bit patterns included - nobody has to clear them
      const bool rok = first + row < N;
      f32x4 v;
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = (rok && c + e < hd) ? src[i][e] * mul : ((ones_col && c + e == hd) ? 1.0f : 0.f);
      if constexpr (X3) {
        bf16x4 hh, ll;
        srad_split4(v, hh, ll);
        *reinterpret_cast<bf16x4*>(base + row * HS + c) = hh;
        *reinterpret_cast<bf16x4*>(base + LOFF + row * HS + c) = ll;
      } else {
        store4<PREC>(base + row * HS + c, v);
      }
      }
    }
  };

  load_tile(tokq, 0, qv);
  load_tile(tokk, 1, kv);
  load_tile(tokk, 2, vv);
  store_tile(Qs, q0, scale, qv);

  f32x4 o[NT_O];
#pragma unroll
  for (int j = 0; j < NT_O; ++j) o[j] = f32x4{0.f, 0.f, 0.f, 0.f};
  // S^T formulation: the MFMA operands are swapped (A = K chunk, B = Q slab), so lane (fq, fr) holds, for ONE query
  // (row wave * 16 + fr), the scores of the 16 keys 16 j + 4 fq + e.  The softmax statistics of a query are then
  // per-lane scalars (two cross-row shuffles per reduction instead of a 16-lane butterfly per score row), and P^T is
  // already laid out as the B operand of O^T += V^T P^T: P never goes through LDS.
  float mrow = -1e30f, lrow = 0.f;
  const int qinf = infq[wave * 16 + fr];
  // bias index (qy - ky + ws - 1) * tw + (qx - kx + ws - 1) = lin_q - lin_k + (ws - 1) * (tw + 1)
  const int aq = (qinf & 0xffffff) + (ws - 1) * (tw + 1), qr = qinf >> 24;
  // ROW64: the query's window coordinates and shift-mask regions (rh = row region, rw = column region; qr = 3 rh + rw)
  const int r64_lin = qinf & 0xffffff, r64_qy = r64_lin / tw, r64_qx = r64_lin - r64_qy * tw;
  const int r64_rhq = qr / 3, r64_rwq = qr - 3 * r64_rhq;
  auto r64_rw = [&](int c) { return c < p.W - 64 ? 0 : (c < p.W - p.shift ? 1 : 2); };
  const bool r64_cdiff[2] = {r64_rwq != r64_rw(wx * 64), r64_rwq != r64_rw(wx * 64 + 32)};    // key columns 0..31 / 32..63

  // Key chunks are software-pipelined: while chunk kc is being multiplied, the K / V rows of chunk kc + 1 are in
  // flight in registers and the token list of chunk kc + 2 is being written (three-slot ring in LDS).
  const int nchunk = (N + 63) / 64;
  for (int kc = 0; kc < nchunk; ++kc) {
    const int k0 = kc * 64;
    const int* const infk_c = infk + (kc % 3) * 64;
    if (kc > 0) __syncthreads();           // previous chunk fully consumed (K / V / token slot (kc + 2) % 3 free)
    store_tile(Ks, k0, 1.f, kv);
    store_tile(Vs, k0, 1.f, vv, ROW64);          // ROW64: column hd of V is 1, so P.V also accumulates the softmax denominator
    if constexpr (ROW64) {
      // the table row chunk kc + 1 adds (its slot was last read four chunks ago), and the load of the one after it
      if (tid < 128 && kc + 1 < nchunk) tbl[((r64_qy0 + 62 - kc) & 3) * 128 + tid] = r64_next;
      if (tid < 128) r64_next = r64_row_load(r64_qy0 + 61 - kc);
    }
    __syncthreads();
    if (kc + 1 < nchunk) {
      const int* tk = tokk + ((kc + 1) % 3) * 64;
      load_tile(tk, 1, kv);
      load_tile(tk, 2, vv);
    }
    if (kc + 2 < nchunk && tid < 64) {
      int tok = 0, inf = 0;
      if (k0 + 128 + tid < N) token_info(k0 + 128 + tid, tok, inf);
      tokk[((kc + 2) % 3) * 64 + tid] = tok; infk[((kc + 2) % 3) * 64 + tid] = inf;
    }

    // ---- S^T = K (Q * scale)^T: s[j][e] = score(key 16 j + 4 fq + e, query wave * 16 + fr) ----
    f32x4 s[4];
    if constexpr (ROW64) {
      // the accumulators START as bias + mask: key position k0 + i = (row kc, column i), bias index
      // (qy - kc + 63) * 127 + (qx - kx + 63) with kx = 16 j + 4 fq + e -> sixteen loads at constant offsets
      const float* const bp = tbl + ((r64_qy - kc + 63) & 3) * 128 + r64_qx - 4 * fq;
      const bool rowdiff = r64_rhq != (wy * 64 + kc < p.H - 64 ? 0 : (wy * 64 + kc < p.H - p.shift ? 1 : 2));
      const float madd[2] = {(p.shift > 0 && (rowdiff || r64_cdiff[0])) ? -100.0f * LOG2E : 0.f,
                             (p.shift > 0 && (rowdiff || r64_cdiff[1])) ? -100.0f * LOG2E : 0.f};
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) s[j][e] = bp[63 - 16 * j - e] + madd[j >> 1];
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) s[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    if constexpr (X3) {
#pragma unroll
      for (int kk = 0; kk < HDP; kk += 32) {                     // small terms first: k_hi.q_lo, k_lo.q_hi, then k_hi.q_hi
        const bf16x8 qf = *reinterpret_cast<const bf16x8*>(Qs + (wave * 16 + fr) * HS + kk + 8 * fq);
        const bf16x8 ql = *reinterpret_cast<const bf16x8*>(Qs + LOFF + (wave * 16 + fr) * HS + kk + 8 * fq);
        bf16x8 kf[4], kl[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          kf[j] = *reinterpret_cast<const bf16x8*>(Ks + (j * 16 + fr) * HS + kk + 8 * fq);
          kl[j] = *reinterpret_cast<const bf16x8*>(Ks + LOFF + (j * 16 + fr) * HS + kk + 8 * fq);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          s[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[j], ql, s[j], 0, 0, 0);
          s[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kl[j], qf, s[j], 0, 0, 0);
          s[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[j], qf, s[j], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    } else if constexpr (PREC == SRAD_PREC_BF16) {
#pragma unroll
      for (int kk = 0; kk < HDP; kk += 32) {                     // a step's five fragments in flight, then its four MFMAs (the scheduler
        const bf16x8 qf = *reinterpret_cast<const bf16x8*>(Qs + (wave * 16 + fr) * HS + kk + 8 * fq);   // sinks each read to its use otherwise)
        bf16x8 kf[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) kf[j] = *reinterpret_cast<const bf16x8*>(Ks + (j * 16 + fr) * HS + kk + 8 * fq);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < 4; ++j) s[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[j], qf, s[j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    } else {
#pragma unroll
      for (int kk = 0; kk < HDP; kk += 16) {
        const f32x4 qf = *reinterpret_cast<const f32x4*>(Qs + (wave * 16 + fr) * HS + kk + 4 * fq);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const f32x4 kf = *reinterpret_cast<const f32x4*>(Ks + (j * 16 + fr) * HS + kk + 4 * fq);
#pragma unroll
          for (int e = 0; e < 4; ++e) s[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[e], qf[e], s[j], 0, 0, 0);
        }
      }
    }

    // ---- bias + mask + online softmax of this lane's query over its 16 keys, then across the four lane groups ----
    float mx = -1e30f;
    if constexpr (ROW64) {
#pragma unroll
      for (int j = 0; j < 4; ++j) mx = fmaxf(mx, fmaxf(fmaxf(s[j][0], s[j][1]), fmaxf(s[j][2], s[j][3])));
    } else {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int4 ki = *reinterpret_cast<const int4*>(infk_c + j * 16 + 4 * fq);
      const int kinf[4] = {ki.x, ki.y, ki.z, ki.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int bi = aq - (kinf[e] & 0xffffff);
        float bias;
        if constexpr (TBL_LDS) bias = tbl[bi]; else bias = p.table[(size_t)bi * heads + h];
        float v = s[j][e] + bias;
        if (p.shift > 0 && qr != (kinf[e] >> 24)) v += -100.0f;
        if constexpr (!FULL) { if (k0 + j * 16 + 4 * fq + e >= N) v = -1e30f; }
        s[j][e] = v;
        mx = fmaxf(mx, v);
      }
    }
    }
    mx = fmaxf(mx, __shfl_xor(mx, 16));
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    const float mnew = fmaxf(mrow, mx);
    const float alpha = ROW64 ? __builtin_amdgcn_exp2f(mrow - mnew) : fexp(mrow - mnew);
    if constexpr (ROW64) {
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) s[j][e] = __builtin_amdgcn_exp2f(s[j][e] - mnew);   // the denominator comes out of P.V (ones column)
    } else {
      float rs = 0.f;
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float pv = (FULL || k0 + j * 16 + 4 * fq + e < N) ? fexp(s[j][e] - mnew) : 0.f;
          s[j][e] = pv;
          rs += pv;
        }
      rs += __shfl_xor(rs, 16);
      rs += __shfl_xor(rs, 32);
      lrow = lrow * alpha + rs;
    }
    mrow = mnew;
#pragma unroll
    for (int j = 0; j < NT_O; ++j) o[j] *= alpha;

    // ---- O^T += V^T P^T: o[ct][e] = out(channel 16 ct + 4 fq + e, query wave * 16 + fr) ----
    if constexpr (PREC != SRAD_PREC_F32) {
      // k slot (fq, t) of a 32-key step stands for key 32 ks + 4 fq + t (t < 4) or 32 ks + 16 + 4 fq + t - 4: exactly the
      // keys whose probabilities this lane holds in s[2 ks] and s[2 ks + 1].  V^T comes from the row-major V tile through
      // two transposing reads: lane 4q+p of a 16-lane group addresses row q, columns 4p..4p+3 of a 4x16 block and
      // receives column (lane & 15) of its 4 rows.
      const int tq = fr >> 2, tp = fr & 3;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        bf16x8 pb;
        [[maybe_unused]] bf16x8 pl;                              // split-bf16: the probabilities' lo terms
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          pb[e] = (__bf16)s[2 * ks][e]; pb[4 + e] = (__bf16)s[2 * ks + 1][e];
          if constexpr (X3) {
            pl[e] = (__bf16)(s[2 * ks][e] - (float)pb[e]);
            pl[4 + e] = (__bf16)(s[2 * ks + 1][e] - (float)pb[4 + e]);
          }
        }
#pragma unroll
        for (int j = 0; j < NT_O; ++j) {
          typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;
          const T* r0 = Vs + (32 * ks + 4 * fq + tq) * HS + j * 16 + 4 * tp;
          const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(r0));
          const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(r0 + 16 * HS));
          bf16x8 vf;
#pragma unroll
          for (int e = 0; e < 4; ++e) { vf[e] = lo[e]; vf[4 + e] = hi[e]; }
          if constexpr (X3) {
            const bf16x4 llo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(r0 + LOFF));
            const bf16x4 lhi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(r0 + LOFF + 16 * HS));
            bf16x8 vl;
#pragma unroll
            for (int e = 0; e < 4; ++e) { vl[e] = llo[e]; vl[4 + e] = lhi[e]; }
            o[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vl, pb, o[j], 0, 0, 0);
            o[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pl, o[j], 0, 0, 0);
          }
          o[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pb, o[j], 0, 0, 0);
        }
      }
    } else {
      // fp32: the MFMA step (jt, e) sums the four keys 16 jt + 4 fq + e, fq = 0..3
#pragma unroll
      for (int jt = 0; jt < 4; ++jt) {
#pragma unroll
        for (int j = 0; j < NT_O; ++j) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float vt = Vs[(jt * 16 + 4 * fq + e) * HS + j * 16 + fr];
            o[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(vt, s[jt][e], o[j], 0, 0, 0);
          }
        }
      }
    }
  }

  // ---- normalise and scatter back to raster order ([T][d], plain head-major columns) ----
  {
    const int row = wave * 16 + fr;
    if constexpr (ROW64) {
      // channel hd of O^T holds the denominator: lane group fq = (hd % 16) / 4, register hd % 4 of column tile hd / 16
      float l = 0.f;
#pragma unroll
      for (int j = 0; j < NT_O; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (j == hd / 16 && e == (hd & 3)) l = o[j][e];
      lrow = __shfl(l, (((hd & 15) >> 2) << 4) | fr);
    }
    if (q0 + row < N) {
      const float inv = 1.0f / lrow;
      if (p.out_h) {                                            // bf16 hand-off to mlp_block
        __bf16* dst = p.out_h + (size_t)tokq[row] * d + h * hd;
#pragma unroll
        for (int j = 0; j < NT_O; ++j)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int c = j * 16 + 4 * fq + e;
            if (c < hd) dst[c] = (__bf16)(o[j][e] * inv);
          }
      } else {
        float* dst = p.out + (size_t)tokq[row] * d + h * hd;
#pragma unroll
        for (int j = 0; j < NT_O; ++j)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int c = j * 16 + 4 * fq + e;
            if (c < hd) dst[c] = o[j][e] * inv;
          }
      }
    }
  }
}

template <int PREC, int NT_O, int NW>
int launch_attn(const AttnParams& p, hipStream_t stream) {
  using T = typename AT<PREC>::type;
  constexpr int PAD = AT<PREC>::PAD;
  constexpr int HDP = NT_O * 16, HS = HDP + PAD, BQ = NW * 16;
  const int N = p.ws * p.ws;
  const int tw = 2 * p.ws - 1;
  constexpr int PLANES = PREC == SRAD_PREC_BF16X3 ? 2 : 1;
  size_t base = (size_t)((BQ + 128) * HS) * sizeof(T) * PLANES + (2 * BQ + 6 * 64) * sizeof(int);
  base = srad_align_up(base, 16);
  const int tbl_in_lds = (base + (size_t)tw * tw * 4) <= 150 * 1024 ? 1 : 0;
  size_t lds = base + (tbl_in_lds ? (size_t)tw * tw * 4 : 0);
  const int full = N % 64 == 0 ? 1 : 0;
  auto kern = tbl_in_lds ? (full ? window_attn_kernel<PREC, NT_O, true, NW, true> : window_attn_kernel<PREC, NT_O, true, NW, false>)
                         : (full ? window_attn_kernel<PREC, NT_O, false, NW, true> : window_attn_kernel<PREC, NT_O, false, NW, false>);
  // 64 x 64 windows with a shift of 0 or 32 (what DRCT builds: window_size // 2): the one-row-per-chunk path
  const bool row64 = PREC != SRAD_PREC_F32 && NW == 8 && p.ws == 64 && base + (size_t)4 * 128 * 4 <= 158 * 1024 &&
                     (p.shift == 0 || p.shift == 32) && getenv("SRAD_NO_ROW64") == nullptr;
  SRAD_REQUIRE(!p.qkv_h || (row64 && (p.d / p.heads) % 4 != 0 && ((uintptr_t)p.qkv_h & 7) == 0),
               "window_attn: bf16 q | k | v are for the 64 x 64-window bf16 path (srad_window_attn_bf16_in)");
  if constexpr (PREC == SRAD_PREC_BF16 && NW == 8) {
    if (row64) {
      kern = p.qkv_h ? window_attn_kernel<PREC, NT_O, true, NW, true, true, true> : window_attn_kernel<PREC, NT_O, true, NW, true, true>;
      lds = base + (size_t)4 * 128 * 4;            // a four-row ring of the bias table
    }
  }
  if constexpr (PREC == SRAD_PREC_BF16X3 && NW == 8) {
    if (row64) {
      kern = window_attn_kernel<PREC, NT_O, true, NW, true, true>;
      lds = base + (size_t)4 * 128 * 4;
    }
  }
  static size_t configured_dev[16][6] = {};          // per device (hipFuncSetAttribute applies to the current one)
  size_t (&configured)[6] = configured_dev[srad_device_slot()];
  const int slot = row64 ? (p.qkv_h ? 5 : 4) : tbl_in_lds * 2 + full;
  if (lds > configured[slot]) {
    SRAD_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    configured[slot] = lds;
  }
  const int nW = (p.H / p.ws) * (p.W / p.ws);
  dim3 grid((N + BQ - 1) / BQ, p.heads, p.B * nW);
  const double Ttok = (double)p.B * p.H * p.W;
  SradProfScope prof(stream, SRAD_K_ATTN, 4.0 * Ttok * N * p.d, 4.0 * Ttok * 4 * p.d);   // q,k,v read + out written
  hipLaunchKernelGGL(kern, grid, dim3(NW * 64), lds, stream, p);
  SRAD_CHECK_HIP(hipGetLastError());
  return SRAD_OK;
}

template <int PREC, int NT_O>
int launch_attn_nw(const AttnParams& p, hipStream_t stream) {
  using T = typename AT<PREC>::type;
  constexpr int HS = NT_O * 16 + AT<PREC>::PAD;
  constexpr int PLANES = PREC == SRAD_PREC_BF16X3 ? 2 : 1;
  // 128-query workgroups when the window has that many tokens and their tiles + the bias table fit in LDS
  const int tw = 2 * p.ws - 1;
  const size_t base8 = (size_t)((128 + 128) * HS) * sizeof(T) * PLANES + (2 * 128 + 6 * 64) * sizeof(int) + 16;
  const size_t lds8 = base8 + (size_t)tw * tw * 4;
  // (64 x 64 windows keep a four-row ring of the table, not the table: launch_attn's one-row-per-chunk path)
  const bool ring = PREC != SRAD_PREC_F32 && p.ws == 64 && (p.shift == 0 || p.shift == 32) && base8 + 2048 <= 158 * 1024;
  if (p.ws * p.ws >= 128 && (lds8 <= 150 * 1024 || ring)) return launch_attn<PREC, NT_O, 8>(p, stream);
  return launch_attn<PREC, NT_O, 4>(p, stream);
}

template <int PREC>
int launch_attn_prec(const AttnParams& p, hipStream_t stream) {
  const int hd = p.d / p.heads;
  if (hd <= 32) return launch_attn_nw<PREC, 2>(p, stream);
  if (hd <= 64) return launch_attn_nw<PREC, 4>(p, stream);
  if (hd <= 96) return launch_attn_nw<PREC, 6>(p, stream);
  if (hd <= 128) return launch_attn_nw<PREC, 8>(p, stream);
  return srad_set_error(SRAD_ERR_ARG, "window_attn: head_dim %d > 128 unsupported", hd);
}

}  // namespace

bool srad_window_attn_bf16_in(int prec, int ws, int shift, int d, int heads, float* qscale) {
  // the conditions of launch_attn's one-row-per-chunk path (its table ring always fits), plus a spare column for V's ones
  const int hd = heads > 0 ? d / heads : 0;
  const bool ok = prec == SRAD_PREC_BF16 && ws == 64 && (shift == 0 || shift == 32) && hd > 0 && hd <= 128 && hd % 4 != 0 &&
                  getenv("SRAD_NO_ROW64") == nullptr && getenv("SRAD_ATTN_F32IN") == nullptr;
  if (ok && qscale) *qscale = (1.0f / sqrtf((float)hd)) * 1.4426950408889634f;
  return ok;
}

int srad_launch_window_attn(int prec, const AttnParams& p, hipStream_t stream) {
  SRAD_REQUIRE(p.ws >= 1 && p.ws <= 128, "window_attn: window size %d out of range", p.ws);
  SRAD_REQUIRE(p.H % p.ws == 0 && p.W % p.ws == 0, "window_attn: %dx%d not a multiple of window %d", p.H, p.W, p.ws);
  SRAD_REQUIRE(p.d % p.heads == 0, "window_attn: dim %d not divisible by heads %d", p.d, p.heads);
  SRAD_REQUIRE(p.shift >= 0 && p.shift < p.ws, "window_attn: shift %d must be in [0, ws)", p.shift);
  SRAD_REQUIRE(p.hdp >= p.d / p.heads && p.hdp % 4 == 0 && (p.qkv_h || ((uintptr_t)p.qkv & 15) == 0),
               "window_attn: head-padded layout needs hdp %% 4 == 0 and hdp >= head_dim (hdp=%d)", p.hdp);
  if (prec == SRAD_PREC_BF16X3) {
    SRAD_REQUIRE(!p.qkv_h && !p.out_h && p.out, "window_attn (split-bf16): fp32 q | k | v in, fp32 out");
    return launch_attn_prec<SRAD_PREC_BF16X3>(p, stream);
  }
  return prec == SRAD_PREC_BF16 ? launch_attn_prec<SRAD_PREC_BF16>(p, stream) : launch_attn_prec<SRAD_PREC_F32>(p, stream);
}
