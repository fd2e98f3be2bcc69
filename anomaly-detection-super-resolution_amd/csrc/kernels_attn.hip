// kernels_attn.hip - fused shifted-window attention for DRCT on gfx950
// (reference src/drct.py: WindowAttention.forward 271-302, window_partition/reverse 193-220,
//  cyclic shift 482-504, calculate_mask 449-470, relative_position_index 250-260).
//
// One workgroup (4 waves) = one (window, head, 64-query tile).  It gathers the window's tokens
// straight from the raster-ordered qkv tensor (cyclic shift and window partition are index
// arithmetic, never copies), streams the window's keys in chunks of 64 with an online softmax
// (so N = ws^2 of 4 ... 4096 tokens all take the same path), adds the relative-position bias
// and the 0/-100 shift mask computed on the fly, and scatters P.V back to raster order.
//   S = (q*scale) k^T : MFMA, A = Q tile in LDS, B = K chunk in LDS (both head-dim contiguous)
//   softmax           : C-layout registers, 16-lane xor-shuffle row reductions
//   O += P V          : P through LDS (per-wave 16-row slab), V chunk staged transposed
#include "srad_common.h"

namespace {

template <int PREC> struct AT;
template <> struct AT<SRAD_PREC_BF16> { using type = __bf16; static constexpr int PAD = 8; };
template <> struct AT<SRAD_PREC_F32>  { using type = float;  static constexpr int PAD = 4; };

struct WinGeom {
  int b, wy, wx;
};

template <int PREC, int NT_O>
__global__ __launch_bounds__(256) void window_attn_kernel(const AttnParams p, const int tbl_in_lds) {
  using T = typename AT<PREC>::type;
  constexpr int PAD = AT<PREC>::PAD;
  constexpr int HDP = NT_O * 16;
  constexpr int HS = HDP + PAD;      // Q/K row stride (elements)
  constexpr int KS = 64 + PAD;       // Vt / P row stride
  extern __shared__ __attribute__((aligned(16))) char smem[];
  T* Qs = reinterpret_cast<T*>(smem);
  T* Ks = Qs + 64 * HS;
  T* Vt = Ks + 64 * HS;
  T* Ps = Vt + HDP * KS;
  int* tokq = reinterpret_cast<int*>(Ps + 64 * KS);
  int* tokk = tokq + 64;
  int* infq = tokk + 64;             // packed (region << 16) | (py << 8) | px
  int* infk = infq + 64;
  float* tbl = reinterpret_cast<float*>(infk + 64);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 15, fq = lane >> 4;
  const int ws = p.ws, N = ws * ws, d = p.d, heads = p.heads, hd = d / heads;
  const int h = blockIdx.y;
  const int nWx = p.W / ws, nW = (p.H / ws) * nWx;
  const int win = blockIdx.z;
  const int b = win / nW, widx = win - b * nW;
  const int wy = widx / nWx, wx = widx - wy * nWx;
  const int q0 = blockIdx.x * 64;
  const float scale = rsqrtf((float)hd);
  const int tw = 2 * ws - 1;

  auto token_info = [&](int pos, int& tok, int& inf) {
    const int py = pos / ws, px = pos - py * ws;
    const int r = wy * ws + py, c = wx * ws + px;              // coordinates in the shifted image
    int orr = r + p.shift; if (orr >= p.H) orr -= p.H;
    int occ = c + p.shift; if (occ >= p.W) occ -= p.W;
    tok = (b * p.H + orr) * p.W + occ;
    const int rh = r < p.H - ws ? 0 : (r < p.H - p.shift ? 1 : 2);
    const int rw = c < p.W - ws ? 0 : (c < p.W - p.shift ? 1 : 2);
    inf = ((rh * 3 + rw) << 16) | (py << 8) | px;
  };

  if (tid < 64) {
    int tok = 0, inf = 0;
    if (q0 + tid < N) token_info(q0 + tid, tok, inf);
    tokq[tid] = tok; infq[tid] = inf;
  }
  if (tbl_in_lds)
    for (int i = tid; i < tw * tw; i += 256) tbl[i] = p.table[(size_t)i * heads + h];
  __syncthreads();

  // ---- stage Q (scaled) ----
  for (int idx = tid; idx < 64 * HDP; idx += 256) {
    const int row = idx / HDP, c = idx - row * HDP;
    float v = 0.f;
    if (c < hd && q0 + row < N) v = p.qkv[(size_t)tokq[row] * (3 * d) + h * hd + c] * scale;
    Qs[row * HS + c] = (T)v;
  }

  f32x4 o[NT_O];
#pragma unroll
  for (int j = 0; j < NT_O; ++j) o[j] = f32x4{0.f, 0.f, 0.f, 0.f};
  float mrow[4], lrow[4];
  int qinf[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) { mrow[e] = -1e30f; lrow[e] = 0.f; qinf[e] = infq[wave * 16 + fq * 4 + e]; }

  const int nchunk = (N + 63) / 64;
  for (int kc = 0; kc < nchunk; ++kc) {
    const int k0 = kc * 64;
    __syncthreads();                       // previous chunk fully consumed (and Q staged)
    if (tid < 64) {
      int tok = 0, inf = 0;
      if (k0 + tid < N) token_info(k0 + tid, tok, inf);
      tokk[tid] = tok; infk[tid] = inf;
    }
    __syncthreads();
    for (int idx = tid; idx < 64 * HDP; idx += 256) {
      const int row = idx / HDP, c = idx - row * HDP;
      float kv = 0.f, vv = 0.f;
      if (c < hd && k0 + row < N) {
        const float* src = p.qkv + (size_t)tokk[row] * (3 * d) + h * hd + c;
        kv = src[d];
        vv = src[2 * d];
      }
      Ks[row * HS + c] = (T)kv;
      Vt[c * KS + row] = (T)vv;
    }
    __syncthreads();

    // ---- S = Q K^T for this wave's 16 query rows x 64 keys ----
    f32x4 s[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) s[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    if constexpr (PREC == SRAD_PREC_BF16) {
#pragma unroll
      for (int kk = 0; kk < HDP; kk += 32) {
        const bf16x8 a = *reinterpret_cast<const bf16x8*>(Qs + (wave * 16 + fr) * HS + kk + 8 * fq);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const bf16x8 bb = *reinterpret_cast<const bf16x8*>(Ks + (j * 16 + fr) * HS + kk + 8 * fq);
          s[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, bb, s[j], 0, 0, 0);
        }
      }
    } else {
#pragma unroll
      for (int kk = 0; kk < HDP; kk += 16) {
        const float4 a = *reinterpret_cast<const float4*>(Qs + (wave * 16 + fr) * HS + kk + 4 * fq);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float4 bb = *reinterpret_cast<const float4*>(Ks + (j * 16 + fr) * HS + kk + 4 * fq);
          s[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, bb.x, s[j], 0, 0, 0);
          s[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, bb.y, s[j], 0, 0, 0);
          s[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, bb.z, s[j], 0, 0, 0);
          s[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, bb.w, s[j], 0, 0, 0);
        }
      }
    }

    // ---- bias + mask + online softmax (row = fq*4+e, key = j*16+fr) ----
    int kinf[4];
    bool kval[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) { kinf[j] = infk[j * 16 + fr]; kval[j] = (k0 + j * 16 + fr) < N; }
    float pmax[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int qy = (qinf[e] >> 8) & 0xff, qx = qinf[e] & 0xff, qr = qinf[e] >> 16;
      float mx = -1e30f;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int ky = (kinf[j] >> 8) & 0xff, kx = kinf[j] & 0xff, kr = kinf[j] >> 16;
        const int bi = (qy - ky + ws - 1) * tw + (qx - kx + ws - 1);
        float v = s[j][e] + (tbl_in_lds ? tbl[bi] : p.table[(size_t)bi * heads + h]);
        if (p.shift > 0 && qr != kr) v += -100.0f;
        if (!kval[j]) v = -1e30f;
        s[j][e] = v;
        mx = fmaxf(mx, v);
      }
      pmax[e] = mx;
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
#pragma unroll
      for (int off = 1; off < 16; off <<= 1) pmax[e] = fmaxf(pmax[e], __shfl_xor(pmax[e], off));
      const float mnew = fmaxf(mrow[e], pmax[e]);
      const float alpha = expf(mrow[e] - mnew);
      float rs = 0.f;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float pv = kval[j] ? expf(s[j][e] - mnew) : 0.f;
        s[j][e] = pv;
        rs += pv;
      }
#pragma unroll
      for (int off = 1; off < 16; off <<= 1) rs += __shfl_xor(rs, off);
      lrow[e] = lrow[e] * alpha + rs;
      mrow[e] = mnew;
#pragma unroll
      for (int j = 0; j < NT_O; ++j) o[j][e] *= alpha;
#pragma unroll
      for (int j = 0; j < 4; ++j) Ps[(wave * 16 + fq * 4 + e) * KS + j * 16 + fr] = (T)s[j][e];
    }
    __syncthreads();

    // ---- O += P V ----
    if constexpr (PREC == SRAD_PREC_BF16) {
#pragma unroll
      for (int kk = 0; kk < 64; kk += 32) {
        const bf16x8 a = *reinterpret_cast<const bf16x8*>(Ps + (wave * 16 + fr) * KS + kk + 8 * fq);
#pragma unroll
        for (int j = 0; j < NT_O; ++j) {
          const bf16x8 bb = *reinterpret_cast<const bf16x8*>(Vt + (j * 16 + fr) * KS + kk + 8 * fq);
          o[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, bb, o[j], 0, 0, 0);
        }
      }
    } else {
#pragma unroll
      for (int kk = 0; kk < 64; kk += 16) {
        const float4 a = *reinterpret_cast<const float4*>(Ps + (wave * 16 + fr) * KS + kk + 4 * fq);
#pragma unroll
        for (int j = 0; j < NT_O; ++j) {
          const float4 bb = *reinterpret_cast<const float4*>(Vt + (j * 16 + fr) * KS + kk + 4 * fq);
          o[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, bb.x, o[j], 0, 0, 0);
          o[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, bb.y, o[j], 0, 0, 0);
          o[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, bb.z, o[j], 0, 0, 0);
          o[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, bb.w, o[j], 0, 0, 0);
        }
      }
    }
  }

  // ---- normalise and scatter back to raster order ----
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int row = wave * 16 + fq * 4 + e;
    if (q0 + row >= N) continue;
    const float inv = 1.0f / lrow[e];
    float* dst = p.out + (size_t)tokq[row] * d + h * hd;
#pragma unroll
    for (int j = 0; j < NT_O; ++j) {
      const int c = j * 16 + fr;
      if (c < hd) dst[c] = o[j][e] * inv;
    }
  }
}

template <int PREC, int NT_O>
int launch_attn(const AttnParams& p, hipStream_t stream) {
  using T = typename AT<PREC>::type;
  constexpr int PAD = AT<PREC>::PAD;
  constexpr int HDP = NT_O * 16, HS = HDP + PAD, KS = 64 + PAD;
  const int N = p.ws * p.ws;
  const int tw = 2 * p.ws - 1;
  size_t base = (size_t)(2 * 64 * HS + HDP * KS + 64 * KS) * sizeof(T) + 4 * 64 * sizeof(int);
  base = srad_align_up(base, 16);
  const int tbl_in_lds = (base + (size_t)tw * tw * 4) <= 150 * 1024 ? 1 : 0;
  const size_t lds = base + (tbl_in_lds ? (size_t)tw * tw * 4 : 0);
  auto kern = window_attn_kernel<PREC, NT_O>;
  static size_t configured = 0;
  if (lds > configured) {
    SRAD_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    configured = lds;
  }
  const int nW = (p.H / p.ws) * (p.W / p.ws);
  dim3 grid((N + 63) / 64, p.heads, p.B * nW);
  const double Ttok = (double)p.B * p.H * p.W;
  SradProfScope prof(stream, SRAD_K_ATTN, 4.0 * Ttok * N * p.d, 4.0 * Ttok * 4 * p.d);   // q,k,v read + out written
  hipLaunchKernelGGL(kern, grid, dim3(256), lds, stream, p, tbl_in_lds);
  SRAD_CHECK_HIP(hipGetLastError());
  return SRAD_OK;
}

template <int PREC>
int launch_attn_prec(const AttnParams& p, hipStream_t stream) {
  const int hd = p.d / p.heads;
  if (hd <= 32) return launch_attn<PREC, 2>(p, stream);
  if (hd <= 64) return launch_attn<PREC, 4>(p, stream);
  if (hd <= 96) return launch_attn<PREC, 6>(p, stream);
  if (hd <= 128) return launch_attn<PREC, 8>(p, stream);
  return srad_set_error(SRAD_ERR_ARG, "window_attn: head_dim %d > 128 unsupported", hd);
}

}  // namespace

int srad_launch_window_attn(int prec, const AttnParams& p, hipStream_t stream) {
  SRAD_REQUIRE(p.ws >= 1 && p.ws <= 128, "window_attn: window size %d out of range", p.ws);
  SRAD_REQUIRE(p.H % p.ws == 0 && p.W % p.ws == 0, "window_attn: %dx%d not a multiple of window %d", p.H, p.W, p.ws);
  SRAD_REQUIRE(p.d % p.heads == 0, "window_attn: dim %d not divisible by heads %d", p.d, p.heads);
  SRAD_REQUIRE(p.shift >= 0 && p.shift < p.ws, "window_attn: shift %d must be in [0, ws)", p.shift);
  return prec == SRAD_PREC_BF16 ? launch_attn_prec<SRAD_PREC_BF16>(p, stream) : launch_attn_prec<SRAD_PREC_F32>(p, stream);
}
