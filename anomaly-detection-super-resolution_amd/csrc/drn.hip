// drn.hip - DRN-L forward engine, the dual regression model, and their C ABI (include/srad.h).
// Follows reference src/drn.py: DRN.forward 241-270, DownBlock 83-119, RCAB 143-158, CALayer 123-139,
// Upsampler 55-81, MeanShift 44-52; src/model.py:8-44 (dual DownBlock).
//
// HBM layout: NHWC fp32 everywhere.  torch.cat((x, copies[..]), 1) (drn.py:263) is a buffer with
// 2*c channels per pixel: the up-branch 1x1 conv writes channels [0, c), the down path wrote its
// skip copy into channels [c, 2c) on the way down, and the next stage reads all 2c - no copy.
// The MeanShift layers are 1x1 convs with TRAINABLE weights in the reference (the requires_grad
// assignment there is a no-op), so they are applied as full CxC matrices, fused into the bicubic
// upsampler (sub_mean) and into the NHWC->NCHW output kernel (add_mean).
#include "train_common.h"
#include "../../include/srad.h"
#include <math.h>
#include <new>

namespace {

// ---- bicubic x scale (align_corners=False, A=-0.75, border clamp) + sub_mean, NCHW -> NHWC[4] ----
__device__ __forceinline__ float cubic1(float x, float A) { return ((A + 2.f) * x - (A + 3.f)) * x * x + 1.f; }
__device__ __forceinline__ float cubic2(float x, float A) { return ((A * x - 5.f * A) * x + 8.f * A) * x - 4.f * A; }

__global__ void bicubic_submean_kernel(const float* __restrict__ x, float* __restrict__ y, int B, int C, int H, int W,
                                       int scale, const float* __restrict__ mw, const float* __restrict__ mb,
                                       float* __restrict__ raw /* training: the upsampled image before sub_mean */) {
  const int Ho = H * scale, Wo = W * scale;
  const size_t total = (size_t)B * Ho * Wo;
  const float A = -0.75f, rs = 1.0f / (float)scale;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int ox = (int)(i % Wo);
    const int oy = (int)((i / Wo) % Ho);
    const int b = (int)(i / ((size_t)Wo * Ho));
    const float ry = rs * ((float)oy + 0.5f) - 0.5f, rx = rs * ((float)ox + 0.5f) - 0.5f;
    const float fy = floorf(ry), fx = floorf(rx);
    const int iy = (int)fy, ix = (int)fx;
    const float ty = ry - fy, tx = rx - fx;
    const float cy[4] = {cubic2(ty + 1.f, A), cubic1(ty, A), cubic1(1.f - ty, A), cubic2(2.f - ty, A)};
    const float cx[4] = {cubic2(tx + 1.f, A), cubic1(tx, A), cubic1(1.f - tx, A), cubic2(2.f - tx, A)};
    float v[3] = {0.f, 0.f, 0.f};
    for (int c = 0; c < C; ++c) {
      const float* pl = x + ((size_t)b * C + c) * H * W;
      float acc = 0.f;
      for (int k = 0; k < 4; ++k) {
        const int yy = min(max(iy - 1 + k, 0), H - 1);
        const float* row = pl + (size_t)yy * W;
        const float r = row[min(max(ix - 1, 0), W - 1)] * cx[0] + row[min(max(ix, 0), W - 1)] * cx[1] +
                        row[min(max(ix + 1, 0), W - 1)] * cx[2] + row[min(max(ix + 2, 0), W - 1)] * cx[3];
        acc += r * cy[k];
      }
      v[c] = acc;
    }
    float o[4] = {0.f, 0.f, 0.f, 0.f};
    for (int co = 0; co < C; ++co) {                 // sub_mean: 1x1 conv (drn.py:246)
      float acc = mb[co];
      for (int ci = 0; ci < C; ++ci) acc += mw[co * C + ci] * v[ci];
      o[co] = acc;
    }
    *reinterpret_cast<float4*>(y + i * 4) = make_float4(o[0], o[1], o[2], o[3]);
    if (raw) *reinterpret_cast<float4*>(raw + i * 4) = make_float4(v[0], v[1], v[2], 0.f);
  }
}

// add_mean (1x1 conv CxC + bias) fused with NHWC -> NCHW           (drn.py:257,266)
__global__ void affine_to_nchw_kernel(const float* __restrict__ x, int ldx, float* __restrict__ y, int B, int C, int HW,
                                      const float* __restrict__ mw, const float* __restrict__ mb) {
  const size_t total = (size_t)B * HW;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int b = (int)(i / HW);
    const int hw = (int)(i - (size_t)b * HW);
    float v[3] = {0.f, 0.f, 0.f};
    for (int c = 0; c < C; ++c) v[c] = x[i * ldx + c];
    for (int co = 0; co < C; ++co) {
      float acc = mb[co];
      for (int ci = 0; ci < C; ++ci) acc += mw[co * C + ci] * v[ci];
      y[((size_t)b * C + co) * HW + hw] = acc;
    }
  }
}

// partial[b][chunk][c] = sum over the chunk's pixels of r[pix][c] (* g[pix][c] when g is given): the global average
// pool of the channel attention (drn.py:127,136) and, with g, its gate gradient.  Partial rows instead of atomics:
// float atomicAdd of every conv output element onto B * C addresses made the pooled convolutions 5x slower.
template <bool RH = false>   // RH: r is a bf16 array (the bf16 training chain's saved conv result)
__global__ __launch_bounds__(256) void pool_dot_kernel(const float* __restrict__ g, const float* __restrict__ r, float* __restrict__ part,
                                                       int hw, int C, int nchunk) {
  auto load_r = [&](size_t o) __attribute__((always_inline)) -> f32x4 {
    if constexpr (RH) {
      const bf16x4 h = *reinterpret_cast<const bf16x4*>(reinterpret_cast<const __bf16*>(r) + o);
      return f32x4{(float)h[0], (float)h[1], (float)h[2], (float)h[3]};
    } else {
      return *reinterpret_cast<const f32x4*>(r + o);
    }
  };
  __shared__ float red[256 * 4];
  const int b = blockIdx.y, chunk = blockIdx.x;
  const int c4n = C / 4, nph = 256 / c4n;
  const int c4 = threadIdx.x % c4n, ph = threadIdx.x / c4n;
  const int per = (hw + nchunk - 1) / nchunk;
  const int p0 = chunk * per, p1 = min(hw, p0 + per);
  f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
  if (ph < nph) {
    // four pixels (eight 16-byte loads) in flight per thread: one at a time the gate-gradient pass ran at 2.4 TB/s on one
    // four-wave workgroup per CU
    int px = p0 + ph;
    if (g) {
      for (; px + 3 * nph < p1; px += 4 * nph) {
        f32x4 rv[4], gv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const size_t o = ((size_t)b * hw + px + u * nph) * C + c4 * 4;
          rv[u] = load_r(o);
          gv[u] = *reinterpret_cast<const f32x4*>(g + o);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) acc += gv[u] * rv[u];
      }
    } else {
      for (; px + 3 * nph < p1; px += 4 * nph) {
        f32x4 rv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) rv[u] = load_r(((size_t)b * hw + px + u * nph) * C + c4 * 4);
#pragma unroll
        for (int u = 0; u < 4; ++u) acc += rv[u];
      }
    }
    for (; px < p1; px += nph) {
      const size_t o = ((size_t)b * hw + px) * C + c4 * 4;
      const f32x4 rv = load_r(o);
      acc += g ? *reinterpret_cast<const f32x4*>(g + o) * rv : rv;
    }
  }
  *reinterpret_cast<f32x4*>(red + threadIdx.x * 4) = acc;
  __syncthreads();
  if (threadIdx.x < c4n) {
    f32x4 t = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int q = 0; q < nph; ++q) t += *reinterpret_cast<const f32x4*>(red + (q * c4n + threadIdx.x) * 4);
    *reinterpret_cast<f32x4*>(part + ((size_t)b * nchunk + chunk) * C + threadIdx.x * 4) = t;
  }
}

constexpr int DRN_POOL_CHUNKS = 32;
constexpr int DRN_POOL_MAXCHUNKS = 256;   // partial rows per image when the conv's epilogue writes them (one per row tile)

// channel-attention gate: sigmoid(W2 relu(W1 mean + b1) + b2), one workgroup per image (drn.py:128-139); sums the
// pool_dot partial rows first and (training) keeps the pooled sums
__global__ __launch_bounds__(128) void ca_gate_kernel(const float* __restrict__ part, int nchunk, float inv_hw, int C, int Cr,
                                                      const float* __restrict__ w1, const float* __restrict__ b1,
                                                      const float* __restrict__ w2, const float* __restrict__ b2,
                                                      float* __restrict__ gate, float* __restrict__ pool_out) {
  __shared__ float mean[512];
  __shared__ float hid[64];
  const int b = blockIdx.x;
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    float sum = 0.f;
    for (int k = 0; k < nchunk; ++k) sum += part[((size_t)b * nchunk + k) * C + c];
    if (pool_out) pool_out[(size_t)b * C + c] = sum;
    mean[c] = sum * inv_hw;
  }
  __syncthreads();
  for (int j = threadIdx.x; j < Cr; j += blockDim.x) {
    float acc = b1[j];
    for (int c = 0; c < C; ++c) acc += w1[j * C + c] * mean[c];
    hid[j] = fmaxf(acc, 0.f);
  }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    float acc = b2[c];
    for (int j = 0; j < Cr; ++j) acc += w2[c * Cr + j] * hid[j];
    gate[(size_t)b * C + c] = 1.0f / (1.0f + expf(-acc));
  }
}

// out = r * gate[b] + x   (RCAB tail, drn.py:139,156-157), float4 over [T][C]
__global__ void scale_add_kernel(const float* __restrict__ r, const float* __restrict__ gate, const float* __restrict__ x,
                                 int ldx, float* __restrict__ y, size_t T, int C, int hw) {
  const int c4n = C / 4;
  const size_t total = T * c4n;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const size_t pix = i / c4n;
    const int c = (int)(i - pix * c4n) * 4;
    const int b = (int)(pix / hw);
    const f32x4 rv = *reinterpret_cast<const f32x4*>(r + pix * C + c);
    const f32x4 g = *reinterpret_cast<const f32x4*>(gate + (size_t)b * C + c);
    const f32x4 xv = *reinterpret_cast<const f32x4*>(x + pix * ldx + c);
    *reinterpret_cast<f32x4*>(y + pix * C + c) = rv * g + xv;
  }
}

// The two kernels above as one launch: every workgroup recomputes its image's gate (C * C/16 MACs from the pooled partial
// rows - cheaper than a launch boundary, and the lone one-workgroup-per-image gate kernel took 12 us) and then streams
// its slice of the image: out = r * gate[b] + x.  grid = (slices per image, B); the slice-0 workgroups also write the gate
// and the pooled sums when the caller keeps them (training).
// NI > 0: a thread's <= NI items of r and x are requested BEFORE the gate chain and wait in registers (every workgroup of the
// launch is resident at once, so the kernel lasts as long as one workgroup's chain: gate round trips + stream round trip)
template <bool RH, bool XH = false, bool YH = false, int NI = 0>   // RH: r is a bf16 array (the eval path's conv80 wrote it so; half the bytes of this
                                                       // bandwidth-bound pass's largest read); XH / YH: so are the chain tensor read / written
__global__ __launch_bounds__(256) void ca_scale_add_kernel(const float* __restrict__ part, int nchunk, float inv_hw, int C, int Cr,
                                                           const float* __restrict__ w1, const float* __restrict__ b1,
                                                           const float* __restrict__ w2, const float* __restrict__ b2,
                                                           float* __restrict__ gate_out, float* __restrict__ pool_out,
                                                           const float* __restrict__ r, const float* __restrict__ x, int ldx,
                                                           float* __restrict__ y, int hw) {
  __shared__ __attribute__((aligned(16))) float gt[512];
  __shared__ float mean[512], hid[64];
  __shared__ f32x4 red[256];
  const int b = blockIdx.y, tid = threadIdx.x;
  const int c4n = C / 4;
  const int per = (hw + gridDim.x - 1) / gridDim.x;
  const int p0 = blockIdx.x * per, p1 = min(hw, p0 + per);
  const size_t base = (size_t)b * hw;
  const int items = (p1 - p0) * c4n;
  typedef typename std::conditional<RH, bf16x4, f32x4>::type rreg_t;
  typedef typename std::conditional<XH, bf16x4, f32x4>::type xreg_t;
  constexpr int NR = NI > 0 ? NI : 1;
  rreg_t r_pre[NR];
  xreg_t x_pre[NR];
  if constexpr (NI > 0) {
#pragma unroll
    for (int u = 0; u < NI; ++u) {
      const int i = max(min(tid + 256 * u, items - 1), 0);     // clamped: no load behind a branch
      const int pl = i / c4n, c = (i - pl * c4n) * 4;
      const size_t pix = base + min(p0 + pl, hw - 1);         // (a slice past the image's end has no items)
      r_pre[u] = *reinterpret_cast<const rreg_t*>(reinterpret_cast<const char*>(r) + (pix * C + c) * (RH ? 2 : 4));
      x_pre[u] = *reinterpret_cast<const xreg_t*>(reinterpret_cast<const char*>(x) + (pix * ldx + c) * (XH ? 2 : 4));
    }
  }
  // The gate is a chain of dependent steps (pool sums -> hidden units -> gate) in front of a bandwidth-side pass, in every
  // workgroup: keep its round trips few.  The small weights go to registers before anything else (C <= 96, C / 16 <= 8: every
  // RCAB of the reference's presets); the partial rows are summed by (row class, channel float4) threads with all of a thread's
  // rows in flight (one thread per channel walking 32 .. 256 rows eight at a time was 4 .. 32 dependent round trips).
  const bool small = C <= 96 && Cr <= 8;
  const int hj = tid >> 5, hl = tid & 31;
  float w1v[3] = {0.f, 0.f, 0.f}, w2v[8], b1v = 0.f, b2v = 0.f;
#pragma unroll
  for (int j = 0; j < 8; ++j) w2v[j] = 0.f;
  if (small) {
#pragma unroll
    for (int u = 0; u < 3; ++u) w1v[u] = w1[min(hj, Cr - 1) * C + min(hl + 32 * u, C - 1)];
    b1v = b1[min(hj, Cr - 1)];
#pragma unroll
    for (int j = 0; j < 8; ++j) w2v[j] = w2[min(tid, C - 1) * Cr + min(j, Cr - 1)];
    b2v = b2[min(tid, C - 1)];
  }
  {
    const int nparts = 256 / c4n, pt = tid / c4n, c4 = tid - pt * c4n;
    f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
    if (pt < nparts) {
      const float* pp = part + (size_t)b * nchunk * C + 4 * c4;
      // twelve rows per trip, clamped and masked: a remainder loop of single rows was one dependent round trip per row (three of
      // them at 32 rows per image, where the eight-row trip never ran); same order of additions
      for (int k = pt; k < nchunk; k += 12 * nparts) {
        f32x4 t[12];
#pragma unroll
        for (int u = 0; u < 12; ++u) t[u] = *reinterpret_cast<const f32x4*>(pp + (size_t)min(k + u * nparts, nchunk - 1) * C);
#pragma unroll
        for (int u = 0; u < 12; ++u) if (k + u * nparts < nchunk) acc += t[u];
      }
      red[tid] = acc;
    }
    __syncthreads();
    for (int c = tid; c < C; c += 256) {
      float sum = 0.f;
      for (int q = 0; q < nparts; ++q) sum += red[q * c4n + (c >> 2)][c & 3];      // fixed order: bit-reproducible
      if (pool_out && blockIdx.x == 0) pool_out[(size_t)b * C + c] = sum;
      mean[c] = sum * inv_hw;
    }
  }
  __syncthreads();
  if (small) {
    float acc = 0.f;
#pragma unroll
    for (int u = 0; u < 3; ++u) acc += hl + 32 * u < C ? w1v[u] * mean[hl + 32 * u] : 0.f;
#pragma unroll
    for (int o = 16; o >= 1; o >>= 1) acc += __shfl_xor(acc, o);
    if (hj < Cr && hl == 0) hid[hj] = fmaxf(acc + b1v, 0.f);
  } else {
    for (int j0 = 0; j0 < Cr; j0 += 8) {                   // 8 hidden units per pass, 32 lanes each
      const int j = j0 + hj;
      float acc = 0.f;
      if (j < Cr) for (int c = hl; c < C; c += 32) acc += w1[j * C + c] * mean[c];
#pragma unroll
      for (int o = 16; o >= 1; o >>= 1) acc += __shfl_xor(acc, o);
      if (j < Cr && hl == 0) hid[j] = fmaxf(acc + b1[j], 0.f);
    }
  }
  __syncthreads();
  for (int c = tid; c < C; c += 256) {
    float acc;
    if (small) {
      acc = b2v;
#pragma unroll
      for (int j = 0; j < 8; ++j) acc += j < Cr ? w2v[j] * hid[j] : 0.f;
    } else {
      acc = b2[c];
      for (int j = 0; j < Cr; ++j) acc += w2[c * Cr + j] * hid[j];
    }
    const float g = 1.0f / (1.0f + expf(-acc));
    gt[c] = g;
    if (gate_out && blockIdx.x == 0) gate_out[(size_t)b * C + c] = g;
  }
  __syncthreads();
  auto widen = [](auto v) {
    if constexpr (std::is_same<decltype(v), bf16x4>::value) return f32x4{(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
    else return v;
  };
  auto finish = [&](int pl, int c, rreg_t rr, xreg_t xx) {
    const size_t pix = base + p0 + pl;
    const f32x4 o = widen(rr) * *reinterpret_cast<const f32x4*>(gt + c) + widen(xx);
    if constexpr (YH) {
      bf16x4 oh;
      oh[0] = (__bf16)o[0]; oh[1] = (__bf16)o[1]; oh[2] = (__bf16)o[2]; oh[3] = (__bf16)o[3];
      *reinterpret_cast<bf16x4*>(reinterpret_cast<__bf16*>(y) + pix * C + c) = oh;
    } else {
      *reinterpret_cast<f32x4*>(y + pix * C + c) = o;
    }
  };
  if constexpr (NI > 0) {
#pragma unroll
    for (int u = 0; u < NI; ++u) {
      const int i = tid + 256 * u;
      const int pl = i / c4n, c = (i - pl * c4n) * 4;
      if (i < items) finish(pl, c, r_pre[u], x_pre[u]);
    }
  } else {
    for (int i = tid; i < items; i += 256) {
      const int pl = i / c4n, c = (i - pl * c4n) * 4;
      const size_t pix = base + p0 + pl;
      finish(pl, c, *reinterpret_cast<const rreg_t*>(reinterpret_cast<const char*>(r) + (pix * C + c) * (RH ? 2 : 4)),
             *reinterpret_cast<const xreg_t*>(reinterpret_cast<const char*>(x) + (pix * ldx + c) * (XH ? 2 : 4)));
    }
  }
}

// can the conv's epilogue write the pool's partial rows (GemmParams::pool_part)?  Rows per image then, else 0
inline int pool_rows_per_image(int prec, const GemmParams& p, int hw) {
  static const bool off = getenv("SRAD_DRN_NO_POOL_FUSE") != nullptr;
  const int bm = srad_gemm_tile_rows(prec, p);
  if (off || bm < 64 || hw % bm != 0 || hw / bm > DRN_POOL_MAXCHUNKS) return 0;
  return hw / bm;
}

// bf16 along a level's RCAB chain in TRAINING (forward saves and backward operands): both convolutions on the 80-channel
// kernel, their weight gradients on the nine-tap kernel, the pool sums from the conv's epilogue.  Decided from the shape
// alone so that the forward and the backward agree; every launch re-checks its kernel's own predicate.
inline bool drn_level_bf16(int prec, int B, int Hl, int Wl, int ch) {
  static const bool off = getenv("SRAD_DRN_CHAIN_F32") || getenv("SRAD_NO_CONV80") || getenv("SRAD_NO_WGRAD_CONV9") ||
                          getenv("SRAD_DRN_NO_POOL_FUSE") || getenv("SRAD_DRN_T_F32");
  if (off || prec != SRAD_PREC_BF16 || ch != 80 || Hl % 4 || Wl % 32) return false;
  const size_t M = (size_t)B * Hl * Wl;
  const int hw = Hl * Wl;
  return M >= 128 * 64 && M * 80 < ((size_t)1 << 31) && hw % 128 == 0 && hw / 128 <= DRN_POOL_MAXCHUNKS;
}

inline int ca_slices(int hw) { return hw >= 128 * 128 ? 128 : (hw >= 1024 ? 64 : (hw >= 64 ? 8 : 1)); }

// ca_scale_add_kernel's instance for the operand storage (r / x / y as bf16) and the slice size: 5 or 10 items per thread wait in
// registers over the gate chain, larger slices take the loop (SRAD_DRN_CA_NO_PRELOAD: always the loop)
template <class... A>
inline void launch_ca_scale_add(bool r_h, bool x_h, bool y_h, int slices, int B, int hw, int C, hipStream_t s, A... args) {
  static const bool no_pre = getenv("SRAD_DRN_CA_NO_PRELOAD") != nullptr;
  const int items = ((hw + slices - 1) / slices) * (C / 4);
  const int ni = no_pre || items > 2560 ? 0 : (items <= 1280 ? 5 : 10);
  const dim3 grid(slices, B);
  auto go = [&](auto kern) { hipLaunchKernelGGL(kern, grid, dim3(256), 0, s, args...); };
  auto by_ni = [&](auto RH, auto XH, auto YH) {
    constexpr bool rh = decltype(RH)::value, xh = decltype(XH)::value, yh = decltype(YH)::value;
    if (ni == 5) go(ca_scale_add_kernel<rh, xh, yh, 5>);
    else if (ni == 10) go(ca_scale_add_kernel<rh, xh, yh, 10>);
    else go(ca_scale_add_kernel<rh, xh, yh, 0>);
  };
  using T = std::true_type; using F = std::false_type;
  if (!r_h) by_ni(F{}, F{}, F{});
  else if (x_h && y_h) by_ni(T{}, T{}, T{});
  else if (x_h) by_ni(T{}, T{}, F{});
  else if (y_h) by_ni(T{}, F{}, T{});
  else by_ni(T{}, F{}, F{});
}

inline int grid1d(size_t total) {
  size_t b = (total + 255) / 256;
  return (int)(b > 4096 ? 4096 : (b < 1 ? 1 : b));
}

struct RcabW {
  ConvW c0, c1;
  int w1, b1, w2, b2;   // CA 1x1 convs, raw fp32
  int ch;
};

}  // namespace

struct srad_drn {
  srad_drn_config cfg;
  ParamTable pt;
  int phase;
  int sub_w, sub_b, add_w, add_b;
  ConvW head;
  std::vector<ConvW> down_s2, down_s1;          // per phase
  std::vector<std::vector<RcabW>> rcab;         // [phase][n_blocks]
  std::vector<ConvW> up_conv, up_1x1;           // per phase
  std::vector<ConvW> tail;                      // phase + 1
  GraphCache gc;
  TrainState ts;                                // training (second half of this file)
  hipStream_t side = nullptr;                   // weight gradients of the RCAB chain run here, beside the data-gradient chain
  std::vector<hipEvent_t> events;
};

namespace {

GemmParams conv_params(const srad_drn* h, const ConvW& c, const float* X, int ldx, int B, int Hi, int Wi, int stride,
                       float* Y, int ldy, int yoff) {
  GemmParams p{};
  const int pad = c.ntaps == 9 ? 1 : 0, k = c.ntaps == 9 ? 3 : 1;
  p.Hi = Hi; p.Wi = Wi;
  p.Ho = (Hi + 2 * pad - k) / stride + 1;
  p.Wo = (Wi + 2 * pad - k) / stride + 1;
  p.stride = stride;
  p.X = X; p.ldx = ldx; p.M = B * p.Ho * p.Wo; p.Cin = c.cin; p.Cp = srad_cp(c.cin); p.ntaps = c.ntaps;
  p.ln_g = p.ln_b = nullptr; p.ln_eps = 1e-5f;
  p.Wp = h->pt.ptr(c.w); p.N = c.n; p.bias = h->pt.fptr(c.b);
  p.act = SRAD_ACT_NONE; p.slope = 0.f; p.alpha = 1.f;
  p.R = nullptr; p.ldr = 0;
  p.Y = Y; p.ldy = ldy; p.yoff = yoff; p.ps = 0;
  p.hsplit_hd = 0; p.hsplit_hdp = 0;
  return p;
}

struct DrnWs {
  float* up0;                       // [T0][4]
  std::vector<float*> cat;          // level L: [T_L][2 F 2^L]
  float* deep;                      // [T_P][F 2^P]
  float* dtmp;                      // down-block intermediate
  float *ra, *rb, *rt, *rr;         // RCAB ping/pong, relu(conv), conv result
  float* ups;                       // conv + pixel-shuffle output
  float* timg;                      // tail conv output [T][C]
  float* pool;                      // [B][DRN_POOL_CHUNKS][chmax] partial sums of the global average pool
  float* gate;                      // [B][chmax]
  size_t bytes;
};

DrnWs plan_ws(const srad_drn* h, int B, int H, int W, void* base, size_t cap) {
  const srad_drn_config& c = h->cfg;
  const int P = h->phase, F = c.n_feats, s = c.scale;
  const size_t T0 = (size_t)B * H * s * W * s;
  Bump bp(base, cap);
  DrnWs w;
  w.up0 = bp.take(T0 * SRAD_IMG_CPAD);
  const int F0 = srad_round_up(F, 4);                 // level-0 channel count as stored (x8 preset: 10 -> 12)
  for (int L = 0; L < P; ++L) w.cat.push_back(bp.take((T0 >> (2 * L)) * 2 * (L == 0 ? F0 : F << L)));
  const size_t TP = T0 >> (2 * P);
  const int top = F << P;
  w.deep = bp.take(TP * top);
  w.dtmp = bp.take((T0 >> 2) * F0);                    // largest: level 0 stride-2 output [T0/4][F]
  // RCAB stacks: idx 0 at level P (top channels), idx >= 1 at level P-idx with 2 F 2^(P-idx) channels
  size_t rmax = TP * top;
  for (int idx = 1; idx < P; ++idx) {
    const size_t e = (T0 >> (2 * (P - idx))) * 2 * F * (1 << (P - idx));
    if (e > rmax) rmax = e;
  }
  w.ra = bp.take(rmax); w.rb = bp.take(rmax); w.rt = bp.take(rmax); w.rr = bp.take(rmax);
  // Upsampler output (before the 1x1): 4x the pixels of its level, same channels
  size_t umax = 0;
  for (int idx = 0; idx < P; ++idx) {
    const int lvl = P - idx;
    const int cin = idx == 0 ? top : 2 * F * (1 << lvl);
    const size_t e = (T0 >> (2 * (lvl - 1))) * cin;
    if (e > umax) umax = e;
  }
  w.ups = bp.take(umax);
  w.timg = bp.take(T0 * SRAD_IMG_CPAD);
  w.pool = bp.take((size_t)B * DRN_POOL_MAXCHUNKS * top);
  w.gate = bp.take((size_t)B * top);
  w.bytes = bp.used;
  return w;
}

int tail_out(srad_drn* h, const ConvW& t, const float* X, int ldx, int B, int Hh, int Ww, const DrnWs& w, float* y,
             hipStream_t s) {
  const int C = h->cfg.n_colors;
  GemmParams p = conv_params(h, t, X, ldx, B, Hh, Ww, 1, w.timg, C, 0);
  SRAD_TRY(srad_launch_gemm(h->cfg.precision, p, s));
  const size_t tot = (size_t)B * Hh * Ww;
  SradProfScope prof(s, SRAD_K_LAYOUT, 2.0 * tot * C * C, 8.0 * tot * C);
  hipLaunchKernelGGL(affine_to_nchw_kernel, dim3(grid1d(tot)), dim3(256), 0, s, w.timg, C, y, B, C, Hh * Ww,
                     h->pt.fptr(h->add_w), h->pt.fptr(h->add_b));
  SRAD_CHECK_HIP(hipGetLastError());
  return SRAD_OK;
}

int forward_body(srad_drn* h, const float* x, int B, int H, int W, float* const* ys, const DrnWs& w, hipStream_t s) {
  const srad_drn_config& c = h->cfg;
  const int prec = c.precision, P = h->phase, F = c.n_feats, sc = c.scale, C = c.n_colors;
  const int H0 = H * sc, W0 = W * sc;
  const int top = F << P;
  const int F0 = srad_round_up(F, 4);   // stored width of the level-0 feature groups; pad columns are exact zeros
  // bicubic upsample to the target size + sub_mean            (drn.py:243-246)
  {
    const size_t tot = (size_t)B * H0 * W0;
    SradProfScope prof(s, SRAD_K_MISC, 40.0 * tot * C, 4.0 * tot * (4 + C));
    hipLaunchKernelGGL(bicubic_submean_kernel, dim3(grid1d(tot)), dim3(256), 0, s, x, w.up0, B, C, H, W, sc,
                       h->pt.fptr(h->sub_w), h->pt.fptr(h->sub_b), (float*)nullptr);
    SRAD_CHECK_HIP(hipGetLastError());
  }
  // head -> copies[0], stored in cat[0][:, F:2F]               (drn.py:247, 252)
  {
    GemmParams p = conv_params(h, h->head, w.up0, SRAD_IMG_CPAD, B, H0, W0, 1, w.cat[0], 2 * F0, F0);
    p.Cin = SRAD_IMG_CPAD;
    SRAD_TRY(srad_launch_gemm(prec, p, s));
  }
  // down phases                                               (drn.py:250-253, 83-119)
  for (int L = 0; L < P; ++L) {
    const int f = L == 0 ? F0 : F << L, Hl = H0 >> L, Wl = W0 >> L;
    GemmParams p = conv_params(h, h->down_s2[L], w.cat[L] + f, 2 * f, B, Hl, Wl, 2, w.dtmp, f, 0);
    p.act = SRAD_ACT_LRELU; p.slope = c.negval;
    SRAD_TRY(srad_launch_gemm(prec, p, s));
    float* dst = L + 1 < P ? w.cat[L + 1] : w.deep;
    const int f1 = F << (L + 1);
    const int ldd = L + 1 < P ? 2 * f1 : f1, off = L + 1 < P ? f1 : 0;
    GemmParams q = conv_params(h, h->down_s1[L], w.dtmp, f, B, Hl / 2, Wl / 2, 1, dst, ldd, off);
    SRAD_TRY(srad_launch_gemm(prec, q, s));
  }
  // coarsest output                                           (drn.py:256-258)
  SRAD_TRY(tail_out(h, h->tail[0], w.deep, top, B, H0 >> P, W0 >> P, w, ys[0], s));

  const float* xin = w.deep;
  int ldin = top;
  for (int idx = 0; idx < P; ++idx) {
    const int lvl = P - idx;
    const int Hl = H0 >> lvl, Wl = W0 >> lvl;
    const int ch = h->rcab[idx][0].ch;
    const size_t T = (size_t)B * Hl * Wl;
    float* cur = w.ra;
    float* nxt = w.rb;
    // bf16 along the chain (bf16 mode, both convolutions on conv80, pool sums from the conv's epilogue): relu(conv), the conv
    // result AND the chain tensor between the blocks are bf16 arrays - every pass of the chain is bandwidth-side work, and the
    // MFMA operands are bf16 anyway.  The level's input and its last block's output (the generic GEMMs' operands) stay fp32.
    static const bool chain_f32 = getenv("SRAD_DRN_CHAIN_F32") != nullptr;
    bool x_h = false;                                          // xin is a bf16 array
    for (int b = 0; b < c.n_blocks; ++b) {
      const RcabW& r = h->rcab[idx][b];
      bool t_bf16 = false;      // the ReLU output between the two convolutions as bf16 (only the second conv reads it) when both take conv80
      {  // conv + ReLU                                         (drn.py:147-150)
        GemmParams p = conv_params(h, r.c0, x_h ? nullptr : xin, ldin, B, Hl, Wl, 1, w.rt, ch, 0);
        if (x_h) p.Xh = reinterpret_cast<const __bf16*>(xin);
        p.act = SRAD_ACT_RELU;
        GemmParams q = conv_params(h, r.c1, w.rt, ch, B, Hl, Wl, 1, w.rr, ch, 0);
        t_bf16 = srad_conv80_supported(prec, p) && srad_conv80_supported(prec, q) && getenv("SRAD_DRN_T_F32") == nullptr;
        SRAD_REQUIRE(t_bf16 || !x_h, "drn_forward: the bf16 chain tensor needs the 80-channel conv kernel");
        if (t_bf16) p.Yh = reinterpret_cast<__bf16*>(w.rt);
        SRAD_TRY(srad_launch_gemm(prec, p, s));
      }
      int nchunk = DRN_POOL_CHUNKS;
      bool nchunk_fused = false;
      {  // conv; its epilogue also leaves the global average pool's partial sums, one row per row tile  (drn.py:147-150, 127)
        GemmParams p = conv_params(h, r.c1, w.rt, ch, B, Hl, Wl, 1, w.rr, ch, 0);
        if (t_bf16) { p.Xh = reinterpret_cast<const __bf16*>(w.rt); p.Yh = reinterpret_cast<__bf16*>(w.rr); }   // r as bf16 too: the pool sums come from the fp32 values in the epilogue
        nchunk = pool_rows_per_image(prec, p, Hl * Wl);
        nchunk_fused = nchunk > 0;
        if (nchunk > 0) p.pool_part = w.pool;
        else p.Yh = nullptr;                                      // the separate pooling pass reads r as fp32
        SRAD_TRY(srad_launch_gemm(prec, p, s));
      }
      if (nchunk == 0) {  // global average pool (partial rows) as a pass of its own                          (drn.py:127-138)
        nchunk = DRN_POOL_CHUNKS;
        SradProfScope prof(s, SRAD_K_MISC, 1.0 * T * ch + 4.0 * B * ch * (ch / 16), 4.0 * T * ch);
        hipLaunchKernelGGL(pool_dot_kernel<false>, dim3(DRN_POOL_CHUNKS, B), dim3(256), 0, s, (const float*)nullptr, w.rr, w.pool, Hl * Wl, ch,
                           DRN_POOL_CHUNKS);
      }
      const bool r_h = t_bf16 && nchunk_fused;
      SRAD_REQUIRE(r_h || !x_h, "drn_forward: the bf16 chain tensor needs the conv epilogue's pool sums");
      const bool y_h = r_h && !chain_f32 && b + 1 < c.n_blocks;   // the next block's conv80 reads it (same shape: supported there too)
      {  // the gate, and res = body(x) * gate + x               (drn.py:128-139, 156-157)
        SradProfScope prof(s, SRAD_K_MISC, 2.0 * T * ch, (double)((r_h ? 2 : 4) + (x_h ? 2 : 4) + (y_h ? 2 : 4)) * T * ch);
        launch_ca_scale_add(r_h, x_h, y_h, ca_slices(Hl * Wl), B, Hl * Wl, ch, s, (const float*)w.pool, nchunk, 1.0f / (float)(Hl * Wl), ch, ch / 16,
                            (const float*)h->pt.fptr(r.w1), (const float*)h->pt.fptr(r.b1), (const float*)h->pt.fptr(r.w2), (const float*)h->pt.fptr(r.b2),
                            (float*)nullptr, (float*)nullptr, (const float*)w.rr, (const float*)xin, ldin, (float*)cur, Hl * Wl);
      }
      SRAD_CHECK_HIP(hipGetLastError());
      xin = cur; ldin = ch; x_h = y_h;
      float* t = cur; cur = nxt; nxt = t;
    }
    // Upsampler: conv ch -> 4 ch + PixelShuffle(2), then the 1x1 reducing conv into cat[lvl-1][:, :cout]
    const int cout = lvl == 1 ? F0 : F << (lvl - 1);
    {
      GemmParams p = conv_params(h, h->up_conv[idx], xin, ldin, B, Hl, Wl, 1, w.ups, ch, 0);
      p.ps = 2;
      SRAD_TRY(srad_launch_gemm(prec, p, s));
    }
    {
      GemmParams p = conv_params(h, h->up_1x1[idx], w.ups, ch, B, 2 * Hl, 2 * Wl, 1, w.cat[lvl - 1], 2 * cout, 0);
      SRAD_TRY(srad_launch_gemm(prec, p, s));
    }
    // torch.cat((x, copies[..]), 1) is cat[lvl-1] as it stands  (drn.py:263)
    xin = w.cat[lvl - 1];
    ldin = 2 * cout;
    SRAD_TRY(tail_out(h, h->tail[idx + 1], xin, ldin, B, 2 * Hl, 2 * Wl, w, ys[idx + 1], s));
  }
  return SRAD_OK;
}

int drn_check_shape(const srad_drn* h, int B, int H, int W) {
  SRAD_REQUIRE(B > 0 && H > 0 && W > 0, "drn: empty input %dx%dx%d", B, H, W);
  SRAD_REQUIRE((double)B * H * W * h->cfg.scale * h->cfg.scale * (h->cfg.n_feats << h->phase) * 4.0 < 3.9e9,
               "drn: problem too large for the 32-bit offsets of the conv kernel");
  return SRAD_OK;
}

}  // namespace

extern "C" {

int srad_drn_create(const srad_drn_config* cfg, srad_drn_t** out) {
  SRAD_REQUIRE(cfg && out, "drn_create: null argument");
  SRAD_REQUIRE(cfg->n_colors == 1 || cfg->n_colors == 3, "drn_create: n_colors must be 1 or 3 (got %d)", cfg->n_colors);
  SRAD_REQUIRE(cfg->scale == 2 || cfg->scale == 4 || cfg->scale == 8, "drn_create: scale must be 2, 4 or 8 (got %d)", cfg->scale);
  SRAD_REQUIRE(cfg->n_feats > 0 && cfg->n_feats % 2 == 0,
               "drn_create: n_feats must be even (got %d): levels >= 1 then hold multiples of 4 channels and only "
               "level 0 is zero-padded to one", cfg->n_feats);
  SRAD_REQUIRE(cfg->n_blocks > 0, "drn_create: n_blocks must be positive");
  SRAD_REQUIRE(cfg->precision == SRAD_PREC_F32 || cfg->precision == SRAD_PREC_BF16 || cfg->precision == SRAD_PREC_BF16X3,
               "drn_create: bad precision %d", cfg->precision);
  srad_drn* h = new (std::nothrow) srad_drn();
  if (!h) return srad_set_error(SRAD_ERR_NOMEM, "drn_create: out of host memory");
  h->cfg = *cfg;
  h->pt.prec = cfg->precision;
  int P = 0;
  for (int s = cfg->scale; s > 1; s >>= 1) ++P;
  h->phase = P;
  const int C = cfg->n_colors, F = cfg->n_feats, top = F << P;
  const int F0 = srad_round_up(F, 4), g0 = F0 != F ? F : 0, g0p = F0 != F ? F0 : 0;   // level-0 group padding
  SRAD_REQUIRE(top <= 512 && top / 16 <= 64, "drn_create: n_feats*2^phase = %d too wide for the gate kernel", top);
  h->sub_w = h->pt.add_raw("sub_mean.weight", C * C);
  h->sub_b = h->pt.add_raw("sub_mean.bias", C);
  h->add_w = h->pt.add_raw("add_mean.weight", C * C);
  h->add_b = h->pt.add_raw("add_mean.bias", C);
  h->head = h->pt.add_layer_padded("head", F, C, 9, true, F0, 0, 0);
  for (int p = 0; p < P; ++p) {
    const int f = F << p;
    const std::string d = "down." + std::to_string(p) + ".dual_module.";
    if (p == 0) {
      h->down_s2.push_back(h->pt.add_layer_padded(d + "0.0", f, f, 9, false, F0, g0, g0p));
      h->down_s1.push_back(h->pt.add_layer_padded(d + "1", 2 * f, f, 9, false, 2 * f, g0, g0p));
    } else {
      h->down_s2.push_back(h->pt.add_layer(d + "0.0", f, f, 9, false));
      h->down_s1.push_back(h->pt.add_layer(d + "1", 2 * f, f, 9, false));
    }
  }
  h->rcab.resize(P);
  for (int idx = 0; idx < P; ++idx) {
    const int ch = idx == 0 ? top : F << (P - idx + 1);
    for (int b = 0; b < cfg->n_blocks; ++b) {
      const std::string q = "up_blocks." + std::to_string(idx) + "." + std::to_string(b) + ".body.";
      RcabW r;
      r.ch = ch;
      r.c0 = h->pt.add_layer(q + "0", ch, ch, 9, true);
      r.c1 = h->pt.add_layer(q + "2", ch, ch, 9, true);
      r.w1 = h->pt.add_raw(q + "3.conv_du.0.weight", (int64_t)(ch / 16) * ch);
      r.b1 = h->pt.add_raw(q + "3.conv_du.0.bias", ch / 16);
      r.w2 = h->pt.add_raw(q + "3.conv_du.2.weight", (int64_t)ch * (ch / 16));
      r.b2 = h->pt.add_raw(q + "3.conv_du.2.bias", ch);
      h->rcab[idx].push_back(r);
    }
    const int cin = idx == 0 ? top : 2 * F * (1 << (P - idx));
    const int cout = F << (P - idx - 1);
    const std::string u = "up_blocks." + std::to_string(idx) + ".";
    h->up_conv.push_back(h->pt.add_layer(u + std::to_string(cfg->n_blocks) + ".0", 4 * cin, cin, 9, true));
    h->up_1x1.push_back(h->pt.add_layer_padded(u + std::to_string(cfg->n_blocks + 1), cout, cin, 1, true,
                                               idx == P - 1 ? F0 : cout, 0, 0));
  }
  h->tail.push_back(h->pt.add_layer("tail.0", C, top, 9, true));
  for (int j = 1, p = P; p >= 1; ++j, --p)   // tail.P reads the level-0 concat, two groups of F stored as F0 each
    h->tail.push_back(h->pt.add_layer_padded("tail." + std::to_string(j), C, F << p, 9, true, C, p == 1 ? g0 : 0, p == 1 ? g0p : 0));
  *out = h;
  return SRAD_OK;
}

void srad_drn_destroy(srad_drn_t* h) {
  if (!h) return;
  h->gc.reset();
  for (hipEvent_t e : h->events) (void)hipEventDestroy(e);
  if (h->side) (void)hipStreamDestroy(h->side);
  delete h;
}

int srad_drn_arena_bytes(const srad_drn_t* h, size_t* bytes) {
  SRAD_REQUIRE(h && bytes, "drn_arena_bytes: null argument");
  *bytes = h->pt.bytes;
  return SRAD_OK;
}

int srad_drn_bind_arena(srad_drn_t* h, void* arena, size_t bytes) {
  SRAD_REQUIRE(h && arena, "drn_bind_arena: null argument");
  SRAD_REQUIRE(bytes >= h->pt.bytes, "drn_bind_arena: %zu bytes given, %zu needed", bytes, h->pt.bytes);
  SRAD_REQUIRE(((uintptr_t)arena & 255) == 0, "drn_bind_arena: arena must be 256-byte aligned");
  h->pt.arena = reinterpret_cast<char*>(arena);
  h->pt.arena_bytes = bytes;
  h->gc.reset();
  return SRAD_OK;
}

int srad_drn_num_params(const srad_drn_t* h) { return h ? (int)h->pt.entries.size() : 0; }

int srad_drn_param_info(const srad_drn_t* h, int idx, const char** name, int64_t* numel) {
  SRAD_REQUIRE(h && idx >= 0 && idx < (int)h->pt.entries.size(), "drn_param_info: index %d out of range", idx);
  if (name) *name = h->pt.entries[idx].name.c_str();
  if (numel) *numel = h->pt.entries[idx].numel;
  return SRAD_OK;
}

int srad_drn_set_param(srad_drn_t* h, const char* name, const float* dev_src, int64_t numel, void* stream) {
  SRAD_REQUIRE(h && name && dev_src, "drn_set_param: null argument");
  return h->pt.set(name, dev_src, numel, reinterpret_cast<hipStream_t>(stream));
}

int srad_drn_workspace_bytes(const srad_drn_t* h, int B, int H, int W, size_t* bytes) {
  SRAD_REQUIRE(h && bytes, "drn_workspace_bytes: null argument");
  SRAD_TRY(drn_check_shape(h, B, H, W));
  *bytes = plan_ws(h, B, H, W, nullptr, 0).bytes;
  return SRAD_OK;
}

int srad_drn_forward(srad_drn_t* h, const float* x, int B, int H, int W, float* const* ys, int n_out, void* workspace,
                     size_t workspace_bytes, void* stream) {
  SRAD_REQUIRE(h && x && ys && workspace, "drn_forward: null argument");
  SRAD_REQUIRE(n_out == h->phase + 1, "drn_forward: %d outputs given, the model returns %d", n_out, h->phase + 1);
  for (int i = 0; i < n_out; ++i) SRAD_REQUIRE(ys[i] != nullptr, "drn_forward: output %d is null", i);
  if (!h->pt.arena) return srad_set_error(SRAD_ERR_STATE, "drn_forward: no weight arena bound");
  SRAD_TRY(drn_check_shape(h, B, H, W));
  SRAD_REQUIRE(((uintptr_t)workspace & 255) == 0, "drn_forward: workspace must be 256-byte aligned");
  const DrnWs w = plan_ws(h, B, H, W, workspace, workspace_bytes);
  SRAD_REQUIRE(w.bytes <= workspace_bytes, "drn_forward: workspace %zu bytes, %zu needed", workspace_bytes, w.bytes);
  std::vector<float*> outs(ys, ys + n_out);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  return srad_run_with_graph(h->gc, h->cfg.use_graph != 0, x, outs.back(), workspace, B, H, W, s,
                             [&](hipStream_t st) { return forward_body(h, x, B, H, W, outs.data(), w, st); });
}

int srad_drn_flops(const srad_drn_t* h, int B, int H, int W, double* flops) {
  SRAD_REQUIRE(h && flops, "drn_flops: null argument");
  const int P = h->phase, sc = h->cfg.scale;
  const double T0 = (double)B * H * sc * W * sc;
  double f = 0;
  auto conv = [&](const ConvW& l, double pix) {   // real (unpadded) sizes
    const ParamEntry& e = h->pt.entries[l.w];
    f += 2.0 * pix * e.n * e.cin * e.ntaps;
  };
  conv(h->head, T0);
  for (int L = 0; L < P; ++L) { conv(h->down_s2[L], T0 / pow(4.0, L + 1)); conv(h->down_s1[L], T0 / pow(4.0, L + 1)); }
  conv(h->tail[0], T0 / pow(4.0, P));
  for (int idx = 0; idx < P; ++idx) {
    const double T = T0 / pow(4.0, P - idx);
    for (const RcabW& r : h->rcab[idx]) { conv(r.c0, T); conv(r.c1, T); }
    conv(h->up_conv[idx], T);
    conv(h->up_1x1[idx], 4 * T);
    conv(h->tail[idx + 1], 4 * T);
  }
  *flops = f;
  return SRAD_OK;
}

// ------------------------------------------------------------------ dual regression model
int srad_dual_workspace_bytes(int B, int C, int H, int W, int n_feats, size_t* bytes) {
  SRAD_REQUIRE(bytes && B > 0 && C > 0 && H > 0 && W > 0 && n_feats > 0, "dual_workspace_bytes: bad argument");
  const size_t T = (size_t)B * H * W, T2 = (size_t)B * ((H + 1) / 2) * ((W + 1) / 2);
  const int Fp = srad_round_up(n_feats, 4);
  *bytes = srad_align_up(T * SRAD_IMG_CPAD * 4, 256) + srad_align_up(T2 * Fp * 4, 256) + srad_align_up(T2 * SRAD_IMG_CPAD * 4, 256) +
           srad_align_up(srad_packed_bytes(SRAD_PREC_F32, Fp, C, 9), 256) + srad_align_up(srad_packed_bytes(SRAD_PREC_F32, C, Fp, 9), 256);
  return SRAD_OK;
}

int srad_dual_forward(const float* w0, const float* w1, int C, int n_feats, float negval, const float* x, int B, int H,
                      int W, float* y, void* workspace, size_t workspace_bytes, int precision, void* stream) {
  SRAD_REQUIRE(w0 && w1 && x && y && workspace, "dual_forward: null argument");
  SRAD_REQUIRE(C == 1 || C == 3, "dual_forward: channels must be 1 or 3 (got %d)", C);
  const int Fp = srad_round_up(n_feats, 4);   // hidden width as stored; pad columns are zeros (x8 preset: 10 -> 12)
  size_t need = 0;
  SRAD_TRY(srad_dual_workspace_bytes(B, C, H, W, n_feats, &need));
  SRAD_REQUIRE(workspace_bytes >= need, "dual_forward: workspace %zu bytes, %zu needed", workspace_bytes, need);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  Bump bp(workspace, workspace_bytes);
  const int H2 = (H + 1) / 2, W2 = (W + 1) / 2;
  const size_t T = (size_t)B * H * W, T2 = (size_t)B * H2 * W2;
  float* xin = bp.take(T * SRAD_IMG_CPAD);
  float* mid = bp.take(T2 * Fp);
  float* outn = bp.take(T2 * SRAD_IMG_CPAD);
  void* p0 = bp.take(srad_packed_bytes(SRAD_PREC_F32, Fp, C, 9) / 4);
  void* p1 = bp.take(srad_packed_bytes(SRAD_PREC_F32, C, Fp, 9) / 4);
  const float zero3[3] = {0.f, 0.f, 0.f};
  SRAD_TRY(srad_launch_nchw_to_nhwc(x, xin, B, C, SRAD_IMG_CPAD, H, W, zero3, 1.0f, s));
  SRAD_TRY(srad_launch_pack_weight_padded(precision, w0, p0, n_feats, C, 9, Fp, 0, 0, s));
  SRAD_TRY(srad_launch_pack_weight_padded(precision, w1, p1, C, n_feats, 9, C, Fp != n_feats ? n_feats : 0, Fp != n_feats ? Fp : 0, s));
  GemmParams a{};
  a.X = xin; a.ldx = SRAD_IMG_CPAD; a.Cin = SRAD_IMG_CPAD; a.Cp = srad_cp(SRAD_IMG_CPAD); a.ntaps = 9;
  a.Hi = H; a.Wi = W; a.Ho = H2; a.Wo = W2; a.stride = 2; a.M = (int)T2;
  a.ln_eps = 1e-5f; a.Wp = p0; a.N = Fp; a.act = SRAD_ACT_LRELU; a.slope = negval; a.alpha = 1.f;
  a.Y = mid; a.ldy = Fp;
  SRAD_TRY(srad_launch_gemm(precision, a, s));
  GemmParams b{};
  b.X = mid; b.ldx = Fp; b.Cin = Fp; b.Cp = srad_cp(Fp); b.ntaps = 9;
  b.Hi = H2; b.Wi = W2; b.Ho = H2; b.Wo = W2; b.stride = 1; b.M = (int)T2;
  b.ln_eps = 1e-5f; b.Wp = p1; b.N = C; b.alpha = 1.f;
  b.Y = outn; b.ldy = SRAD_IMG_CPAD;
  SRAD_TRY(srad_launch_gemm(precision, b, s));
  return srad_launch_nhwc_to_nchw(outn, SRAD_IMG_CPAD, y, B, C, H2, W2, zero3, 1.0f, s);
}

}  // extern "C"

// =====================================================================================================
// Training: DRN forward with saved activations + backward (reference src/trainer.py:161-205 on src/drn.py),
// and the backward of the dual regression model (src/model.py:8-44).  Same plumbing as the DRCT training engine:
// flat fp32 parameter / gradient buffers, data gradients = the forward GEMM on transposed packs, weight gradients
// through the split-K queue.  The stride-2 convolutions' data gradient is the stride-1 transposed convolution of
// the gradient with zeros inserted between its pixels.
// =====================================================================================================
namespace {

// Z[b][2 oy][2 ox][c] = dY[b][oy][ox][c], all other pixels 0                  (stride-2 conv backward)
__global__ void zero_upsample_kernel(const float* __restrict__ dy, float* __restrict__ z, int B, int Ho, int Wo, int C) {
  const int c4n = C / 4;
  const size_t total = (size_t)B * 2 * Ho * 2 * Wo * c4n;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int c4 = (int)(i % c4n);
    size_t pix = i / c4n;
    const int x = (int)(pix % (2 * Wo)); pix /= 2 * Wo;
    const int y = (int)(pix % (2 * Ho));
    const int b = (int)(pix / (2 * Ho));
    f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
    if (!(x & 1) && !(y & 1)) v = *reinterpret_cast<const f32x4*>(dy + (((size_t)b * Ho + (y >> 1)) * Wo + (x >> 1)) * C + c4 * 4);
    *reinterpret_cast<f32x4*>(z + i * 4) = v;
  }
}

// CALayer backward (src/drn.py:123-139) for all images of the batch in ONE workgroup (the work is C * C/16 MACs per
// image): dgate = sum of the pool_dot partials; through the sigmoid, the two 1x1 convs and the ReLU back to the mean;
// dpool[b][c] = dmean[c] / HW.  Weight / bias gradients are summed over the images in registers and added to the flat
// gradient buffer without atomics.  The images are processed eight at a time with the work spread over (image, channel)
// pairs and the two small weight matrices staged in LDS: the first version walked the images one after the other with
// the C/16 hidden units on C/16 threads reading the weights from global memory - 94 us per launch, 80 launches per step.
constexpr int CAB_G = 8;
__global__ __launch_bounds__(256) void ca_bwd_kernel(const float* __restrict__ part, int nchunk, const float* __restrict__ pool,
                                                     const float* __restrict__ gate, float inv_hw, int B, int C, int Cr,
                                                     const float* __restrict__ w1, const float* __restrict__ b1,
                                                     const float* __restrict__ w2, float* __restrict__ dw1,
                                                     float* __restrict__ db1, float* __restrict__ dw2, float* __restrict__ db2,
                                                     float* __restrict__ dpool) {
  __shared__ float s[CAB_G][512], mean[CAB_G][512];
  __shared__ float hid[CAB_G][64], dhid[CAB_G][64];
  __shared__ float lw1[4096], lw2[4096], lb1[64];    // C * Cr <= 4096 entries each
  const int tid = threadIdx.x;
  const int nw = C * Cr;
  for (int i = tid; i < nw; i += 256) { lw1[i] = w1[i]; lw2[i] = w2[i]; }
  if (tid < Cr) lb1[tid] = b1[tid];
  float aw1[16], aw2[16], ab2[2] = {0.f, 0.f}, ab1 = 0.f;
#pragma unroll
  for (int q = 0; q < 16; ++q) { aw1[q] = 0.f; aw2[q] = 0.f; }
  for (int b0 = 0; b0 < B; b0 += CAB_G) {
    const int g = min(CAB_G, B - b0);
    __syncthreads();                                       // weights staged / the previous pass is fully consumed
    // dgate through the sigmoid, mean.  One (image, channel float4) per thread with 16 partial rows in flight: per channel with
    // eight in flight a thread walked 2.5 channels x 4 batches = 10 dependent round trips, most of the launch's 18 us
    // (80 launches on the critical path of a DRN-L step)
    const int c4n = C >> 2;
    for (int idx = tid; idx < g * c4n; idx += 256) {
      const int bb = idx / c4n, c4 = idx - bb * c4n, b = b0 + bb;
      const float* pp = part + (size_t)b * nchunk * C + 4 * c4;
      f32x4 dg = f32x4{0.f, 0.f, 0.f, 0.f};
      int k = 0;
      for (; k + 16 <= nchunk; k += 16) {
        f32x4 t[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) t[u] = *reinterpret_cast<const f32x4*>(pp + (size_t)(k + u) * C);
#pragma unroll
        for (int u = 0; u < 16; ++u) dg += t[u];
      }
      for (; k < nchunk; ++k) dg += *reinterpret_cast<const f32x4*>(pp + (size_t)k * C);
      const f32x4 gt = *reinterpret_cast<const f32x4*>(gate + (size_t)b * C + 4 * c4);
      const f32x4 pl = *reinterpret_cast<const f32x4*>(pool + (size_t)b * C + 4 * c4);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        s[bb][4 * c4 + e] = dg[e] * gt[e] * (1.f - gt[e]);
        mean[bb][4 * c4 + e] = pl[e] * inv_hw;
      }
    }
    __syncthreads();
    for (int idx = tid; idx < g * Cr; idx += 256) {        // hidden units of every image
      const int bb = idx / Cr, j = idx - bb * Cr;
      float acc = lb1[j], dh = 0.f;
      for (int c = 0; c < C; ++c) {
        acc += lw1[j * C + c] * mean[bb][c];
        dh += s[bb][c] * lw2[c * Cr + j];
      }
      hid[bb][j] = fmaxf(acc, 0.f);
      dhid[bb][j] = acc > 0.f ? dh : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 16; ++q) {                         // weight gradients, summed over the pass's images
      const int i = tid + 256 * q;
      if (i < nw) {
        const int c = i / Cr, j = i - c * Cr;              // w2 [C][Cr]
        const int j1 = i / C, c1 = i - j1 * C;             // w1 [Cr][C]
        float t2 = 0.f, t1 = 0.f;
        for (int bb = 0; bb < g; ++bb) { t2 += s[bb][c] * hid[bb][j]; t1 += dhid[bb][j1] * mean[bb][c1]; }
        aw2[q] += t2; aw1[q] += t1;
      }
    }
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int c = tid + 256 * q;
      if (c < C) for (int bb = 0; bb < g; ++bb) ab2[q] += s[bb][c];
    }
    if (tid < Cr) for (int bb = 0; bb < g; ++bb) ab1 += dhid[bb][tid];
    for (int idx = tid; idx < g * C; idx += 256) {         // back to the pooled mean
      const int bb = idx / C, c = idx - bb * C;
      float dm = 0.f;
      for (int j = 0; j < Cr; ++j) dm += dhid[bb][j] * lw1[j * C + c];
      dpool[(size_t)(b0 + bb) * C + c] = dm * inv_hw;
    }
  }
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    const int i = tid + 256 * q;
    if (i < nw) { dw1[i] += aw1[q]; dw2[i] += aw2[q]; }
  }
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int c = tid + 256 * q;
    if (c < C) db2[c] += ab2[q];
  }
  if (tid < Cr) db1[tid] += ab1;
}

// dr = g * gate[b] + dpool[b]                                                   (RCAB: r * gate + x, drn.py:139,156)
// with dpool recomputed per workgroup from the pool_dot partial rows (as the forward's ca_scale_add recomputes the gate): the
// channel attention's data gradient - sigmoid, the two 1x1 convs, the ReLU, back to the mean - is C * C/16 * 3 MACs per image,
// while the one-workgroup ca_bwd_kernel that used to hand it over sat on the critical path of every block (16 us x 80 per
// step); that kernel now only produces the weight gradients, on the side stream.  grid = (slices per image, B).
template <bool YH = false, int NI = 0>   // YH: dr is written as a bf16 array (only MFMA operands read it: the conv's data and weight gradients)
                                         // NI > 0: a thread's <= NI float4 of g wait in registers over the chain (see ca_scale_add_kernel)
__global__ __launch_bounds__(256) void ca_apply_bwd_kernel(const float* __restrict__ g, const float* __restrict__ gate,
                                                           const float* __restrict__ part, int nchunk, const float* __restrict__ pool,
                                                           float inv_hw, int C, int Cr, const float* __restrict__ w1,
                                                           const float* __restrict__ b1, const float* __restrict__ w2,
                                                           float* __restrict__ dr, int hw) {
  __shared__ __attribute__((aligned(16))) float gt[512], dpl[512];
  __shared__ float sg[512], mean[512], dhid[64];
  __shared__ f32x4 red[256];
  const int b = blockIdx.y, tid = threadIdx.x;
  const int c4n = C / 4;
  const int hj = tid >> 5, hl = tid & 31;
  const int per = (hw + gridDim.x - 1) / gridDim.x;
  const int p0 = blockIdx.x * per, p1 = min(hw, p0 + per);
  const size_t base = (size_t)b * hw;
  const int items = (p1 - p0) * c4n;
  f32x4 g_pre[NI > 0 ? NI : 1];
  if constexpr (NI > 0) {
#pragma unroll
    for (int u = 0; u < NI; ++u) {
      const int i = max(min(tid + 256 * u, items - 1), 0);
      const int pl = i / c4n, c = (i - pl * c4n) * 4;
      g_pre[u] = *reinterpret_cast<const f32x4*>(g + (base + min(p0 + pl, hw - 1)) * C + c);
    }
  }
  // the small weights first, into registers (C <= 96, C / 16 <= 8: every RCAB of the reference's presets), so that the chain
  // below is one memory round trip and not one per step
  const bool small = C <= 96 && Cr <= 8;
  float w1h[3] = {0.f, 0.f, 0.f}, w2h[3] = {0.f, 0.f, 0.f}, w1c[8], b1h = 0.f;
#pragma unroll
  for (int j = 0; j < 8; ++j) w1c[j] = 0.f;
  if (small) {
    const int j = min(hj, Cr - 1);
#pragma unroll
    for (int u = 0; u < 3; ++u) {
      const int c = min(hl + 32 * u, C - 1);
      w1h[u] = w1[j * C + c];
      w2h[u] = w2[c * Cr + j];
    }
    b1h = b1[j];
#pragma unroll
    for (int q = 0; q < 8; ++q) w1c[q] = w1[min(q, Cr - 1) * C + min(tid, C - 1)];
  }
  {  // dgate = sum of the partial rows, all of a thread's rows in flight (see ca_scale_add_kernel)
    const int nparts = 256 / c4n, pt = tid / c4n, c4 = tid - pt * c4n;
    f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
    if (pt < nparts) {
      const float* pp = part + (size_t)b * nchunk * C + 4 * c4;
      for (int k = pt; k < nchunk; k += 4 * nparts) {          // four rows per trip, clamped and masked (see ca_scale_add_kernel)
        f32x4 t[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) t[u] = *reinterpret_cast<const f32x4*>(pp + (size_t)min(k + u * nparts, nchunk - 1) * C);
#pragma unroll
        for (int u = 0; u < 4; ++u) if (k + u * nparts < nchunk) acc += t[u];
      }
      red[tid] = acc;
    }
    for (int c = tid; c < C; c += 256) {
      gt[c] = gate[(size_t)b * C + c];
      mean[c] = pool[(size_t)b * C + c] * inv_hw;
    }
    __syncthreads();
    for (int c = tid; c < C; c += 256) {
      float dg = 0.f;
      for (int q = 0; q < nparts; ++q) dg += red[q * c4n + (c >> 2)][c & 3];
      sg[c] = dg * gt[c] * (1.f - gt[c]);                  // through the sigmoid
    }
  }
  __syncthreads();
  if (small) {                                             // hidden units: pre-activation (its sign is the ReLU's mask) and gradient
    float acc = 0.f, dh = 0.f;
#pragma unroll
    for (int u = 0; u < 3; ++u) {
      const int c = hl + 32 * u;
      if (c < C) { acc += w1h[u] * mean[c]; dh += sg[c] * w2h[u]; }
    }
#pragma unroll
    for (int o = 16; o >= 1; o >>= 1) { acc += __shfl_xor(acc, o); dh += __shfl_xor(dh, o); }
    if (hj < Cr && hl == 0) dhid[hj] = acc + b1h > 0.f ? dh : 0.f;
  } else {
    for (int j0 = 0; j0 < Cr; j0 += 8) {
      const int j = j0 + hj;
      float acc = 0.f, dh = 0.f;
      if (j < Cr)
        for (int c = hl; c < C; c += 32) {
          acc += w1[j * C + c] * mean[c];
          dh += sg[c] * w2[c * Cr + j];
        }
#pragma unroll
      for (int o = 16; o >= 1; o >>= 1) { acc += __shfl_xor(acc, o); dh += __shfl_xor(dh, o); }
      if (j < Cr && hl == 0) dhid[j] = acc + b1[j] > 0.f ? dh : 0.f;
    }
  }
  __syncthreads();
  for (int c = tid; c < C; c += 256) {                     // back to the pooled mean
    float dm = 0.f;
    if (small) {
#pragma unroll
      for (int j = 0; j < 8; ++j) dm += j < Cr ? dhid[j] * w1c[j] : 0.f;
    } else {
      for (int j = 0; j < Cr; ++j) dm += dhid[j] * w1[j * C + c];
    }
    dpl[c] = dm * inv_hw;
  }
  __syncthreads();
  auto finish = [&](int pl, int c, f32x4 gv) {
    const size_t pix = base + p0 + pl;
    const f32x4 v = gv * *reinterpret_cast<const f32x4*>(gt + c) + *reinterpret_cast<const f32x4*>(dpl + c);
    if constexpr (YH) {
      bf16x4 h;
      h[0] = (__bf16)v[0]; h[1] = (__bf16)v[1]; h[2] = (__bf16)v[2]; h[3] = (__bf16)v[3];
      *reinterpret_cast<bf16x4*>(reinterpret_cast<__bf16*>(dr) + pix * C + c) = h;
    } else {
      *reinterpret_cast<f32x4*>(dr + pix * C + c) = v;
    }
  };
  if constexpr (NI > 0) {
#pragma unroll
    for (int u = 0; u < NI; ++u) {
      const int i = tid + 256 * u;
      const int pl = i / c4n, c = (i - pl * c4n) * 4;
      if (i < items) finish(pl, c, g_pre[u]);
    }
  } else {
    for (int i = tid; i < items; i += 256) {
      const int pl = i / c4n, c = (i - pl * c4n) * 4;
      finish(pl, c, *reinterpret_cast<const f32x4*>(g + (base + p0 + pl) * C + c));
    }
  }
}

// dst[m][0..C) += src[m][0..C)
__global__ void add_cols_kernel(const float* __restrict__ src, int ld_src, float* __restrict__ dst, int ld_dst, size_t rows, int C) {
  const int c4n = C / 4;
  const size_t total = rows * c4n;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const size_t m = i / c4n;
    const int c = (int)(i - m * c4n) * 4;
    f32x4* d = reinterpret_cast<f32x4*>(dst + m * ld_dst + c);
    *d = *d + *reinterpret_cast<const f32x4*>(src + m * ld_src + c);
  }
}

// MeanShift backward (1x1 conv CxC + bias, trainable in the reference): dt = W^T dy (optional), dW += dy (x) t, db += dy.
// dy comes either as NCHW (add_mean, the network outputs) or as NHWC[4] rows (sub_mean); t is NHWC[ldt].
__global__ __launch_bounds__(256) void affine_bwd_kernel(const float* __restrict__ dy_nchw, const float* __restrict__ dy_rows,
                                                         const float* __restrict__ t, int ldt, float* __restrict__ dt, int B,
                                                         int C, int HW, const float* __restrict__ w, float* __restrict__ dw,
                                                         float* __restrict__ db) {
  __shared__ float red[12 * 256];
  float a[12];
#pragma unroll
  for (int k = 0; k < 12; ++k) a[k] = 0.f;
  const size_t total = (size_t)B * HW;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int b = (int)(i / HW);
    const int hw = (int)(i - (size_t)b * HW);
    float d[3] = {0.f, 0.f, 0.f}, tv[3] = {0.f, 0.f, 0.f};
    for (int c = 0; c < C; ++c) {
      d[c] = dy_nchw ? dy_nchw[((size_t)b * C + c) * HW + hw] : dy_rows[i * 4 + c];
      tv[c] = t[i * ldt + c];
    }
    for (int c = 0; c < C; ++c) {
      for (int c2 = 0; c2 < C; ++c2) a[c * 3 + c2] += d[c] * tv[c2];
      a[9 + c] += d[c];
    }
    if (dt) {
      float o[4] = {0.f, 0.f, 0.f, 0.f};
      for (int c2 = 0; c2 < C; ++c2)
        for (int c = 0; c < C; ++c) o[c2] += w[c * C + c2] * d[c];
      *reinterpret_cast<float4*>(dt + i * 4) = make_float4(o[0], o[1], o[2], o[3]);
    }
  }
#pragma unroll
  for (int k = 0; k < 12; ++k) red[k * 256 + threadIdx.x] = a[k];
  __syncthreads();
  if (threadIdx.x < 12) {
    float sum = 0.f;
    for (int q = 0; q < 256; ++q) sum += red[threadIdx.x * 256 + q];
    const int k = threadIdx.x;
    if (k < 9) { if (k / 3 < C && k % 3 < C) atomicAdd(dw + (k / 3) * C + (k % 3), sum); }
    else if (k - 9 < C) atomicAdd(db + (k - 9), sum);
  }
}

struct RcabSave { float *t, *r, *xo, *gate, *pool; };

struct DrnTrainWs {
  float *uraw, *up0, *deep;
  std::vector<float*> cat, dtmp, ups, timg;
  std::vector<std::vector<RcabSave>> rc;
  // backward
  float *gdeep, *ga, *gb, *dr2[2], *dt2[2], *dups, *dus, *ddtmp, *zup, *dtimg, *dup0, *ppart, *ppart2[2], *dpool2[2];   // dr / dt / pool_dot rows double-buffered (two streams)
  std::vector<float*> gcat;
  size_t bytes;
};

DrnTrainWs plan_train_ws(const srad_drn* h, int B, int H, int W, void* base, size_t cap) {
  const srad_drn_config& c = h->cfg;
  const int P = h->phase, F = c.n_feats, s = c.scale, top = F << P;
  const int F0 = srad_round_up(F, 4);                    // level 0 as stored (x8 preset: 10 -> 12), pad columns exact zeros
  auto fw = [&](int L) { return L == 0 ? F0 : F << L; };
  const size_t T0 = (size_t)B * H * s * W * s;
  Bump bp(base, cap);
  DrnTrainWs w;
  w.uraw = bp.take(T0 * SRAD_IMG_CPAD);
  w.up0 = bp.take(T0 * SRAD_IMG_CPAD);
  for (int L = 0; L < P; ++L) w.cat.push_back(bp.take((T0 >> (2 * L)) * 2 * fw(L)));
  const size_t TP = T0 >> (2 * P);
  w.deep = bp.take(TP * top);
  for (int L = 0; L < P; ++L) w.dtmp.push_back(bp.take((T0 >> (2 * (L + 1))) * fw(L)));
  size_t rmax = 0, umax = 0;
  w.rc.resize(P);
  for (int idx = 0; idx < P; ++idx) {
    const int lvl = P - idx;
    const int ch = h->rcab[idx][0].ch;
    const size_t T = T0 >> (2 * lvl);
    for (int b = 0; b < c.n_blocks; ++b) {
      RcabSave r;
      r.t = bp.take(T * ch); r.r = bp.take(T * ch); r.xo = bp.take(T * ch);
      r.gate = bp.take((size_t)B * ch); r.pool = bp.take((size_t)B * ch);
      w.rc[idx].push_back(r);
    }
    w.ups.push_back(bp.take(4 * T * ch));
    if (T * ch > rmax) rmax = T * ch;
    if (4 * T * ch > umax) umax = 4 * T * ch;
  }
  for (int j = 0; j <= P; ++j) w.timg.push_back(bp.take((T0 >> (2 * (P - j))) * SRAD_IMG_CPAD));
  // backward
  for (int L = 0; L < P; ++L) w.gcat.push_back(bp.take((T0 >> (2 * L)) * 2 * fw(L)));
  w.gdeep = bp.take(TP * top);
  w.ga = bp.take(rmax); w.gb = bp.take(rmax);
  for (int i = 0; i < 2; ++i) { w.dr2[i] = bp.take(rmax); w.dt2[i] = bp.take(rmax); }
  w.dups = bp.take(umax); w.dus = bp.take(umax);
  w.ddtmp = bp.take((T0 >> 2) * F0);
  w.zup = bp.take(T0 * F0);
  w.dtimg = bp.take(T0 * SRAD_IMG_CPAD);
  w.dup0 = bp.take(T0 * SRAD_IMG_CPAD);
  w.ppart = bp.take((size_t)B * DRN_POOL_MAXCHUNKS * top);
  w.ppart2[0] = w.ppart;                                   // the backward's pool_dot rows (DRN_POOL_CHUNKS per image): the forward's are spent
  w.ppart2[1] = bp.take((size_t)B * DRN_POOL_CHUNKS * top);
  for (int i = 0; i < 2; ++i) w.dpool2[i] = bp.take((size_t)B * top);
  w.bytes = bp.used;
  return w;
}

int drn_train_check(const srad_drn* h, int B, int H, int W) {
  SRAD_REQUIRE(h->ts.ready, "drn training: call srad_drn_train_bind() and srad_drn_sync_params() first");
  const int top = h->cfg.n_feats << h->phase;
  SRAD_REQUIRE(top * (top / 16) <= 4096 && top <= 512, "drn training: channel attention too wide for the backward kernel");
  return drn_check_shape(h, B, H, W);
}

GemmParams drn_dgrad(const srad_drn* h, const ConvW& c, const float* dY, int ldy, int B, int Hh, int Ww, float* dX, int ldx, int xoff) {
  GemmParams p{};
  const int npad = srad_round_up(c.n, 4), cpad = srad_round_up(c.cin, 4);
  p.Hi = p.Ho = Hh; p.Wi = p.Wo = Ww; p.stride = 1;
  p.X = dY; p.ldx = ldy; p.M = B * Hh * Ww; p.Cin = npad; p.Cp = srad_cp(npad); p.ntaps = c.ntaps; p.ln_eps = 1e-5f;
  p.Wp = h->ts.tarena + h->ts.t_off[c.w]; p.N = cpad; p.alpha = 1.f;
  p.Y = dX; p.ldy = ldx; p.yoff = xoff;
  return p;
}
void accumulate_into(GemmParams& p) { p.R = p.Y + p.yoff; p.ldr = p.ldy; p.rmode = SRAD_RMODE_ADD; }

WgradParams drn_wgrad(const srad_drn* h, const ConvW& c, float* G, const float* dY, int ldy, int ycol0, const float* X, int ldx,
                      int B, int Hi, int Wi, int stride) {
  WgradParams p{};
  const int pad = c.ntaps == 9 ? 1 : 0, k = c.ntaps == 9 ? 3 : 1;
  p.Hi = Hi; p.Wi = Wi; p.Ho = (Hi + 2 * pad - k) / stride + 1; p.Wo = (Wi + 2 * pad - k) / stride + 1; p.stride = stride;
  p.dY = dY; p.ldy = ldy; p.ycol0 = ycol0; p.X = X; p.ldx = ldx; p.M = B * p.Ho * p.Wo;
  p.N = srad_round_up(c.n, 4); p.Cin = srad_round_up(c.cin, 4); p.ntaps = c.ntaps;
  const ParamEntry& e = h->pt.entries[c.w];             // the flat gradient holds the real tensor; c carries the stored extents
  p.n_real = e.n; p.cin_real = e.cin; p.grp_real = e.grp_real; p.grp_pad = e.grp_pad; p.alpha = 1.f;
  p.dW = G + h->ts.flat_off[c.w];
  p.db = c.b >= 0 ? G + h->ts.flat_off[c.b] : nullptr;
  return p;
}

}  // namespace

extern "C" {

int srad_drn_train_param_floats(srad_drn_t* h, int64_t* total) {
  SRAD_REQUIRE(h && total, "train_param_floats: null argument");
  return train_param_floats(h->pt, h->ts, total);
}
int srad_drn_train_param_offset(srad_drn_t* h, int idx, int64_t* off_floats) {
  int64_t tot = 0;
  SRAD_REQUIRE(h, "train_param_offset: null argument");
  SRAD_TRY(train_param_floats(h->pt, h->ts, &tot));
  SRAD_REQUIRE(off_floats && idx >= 0 && idx < (int)h->pt.entries.size(), "train_param_offset: index %d out of range", idx);
  *off_floats = h->ts.flat_off[idx];
  return SRAD_OK;
}
int srad_drn_train_arena_bytes(srad_drn_t* h, size_t* bytes) {
  SRAD_REQUIRE(h && bytes, "train_arena_bytes: null argument");
  return train_arena_bytes(h->pt, h->ts, bytes);
}
int srad_drn_train_bind(srad_drn_t* h, void* train_arena, size_t bytes) {
  SRAD_REQUIRE(h, "train_bind: null argument");
  return train_bind(h->pt, h->ts, train_arena, bytes);
}
int srad_drn_sync_params(srad_drn_t* h, const float* flat_params, void* stream) {
  SRAD_REQUIRE(h, "sync_params: null argument");
  SRAD_TRY(train_sync_params(h->pt, h->ts, flat_params, reinterpret_cast<hipStream_t>(stream)));
  h->gc.reset();
  return SRAD_OK;
}
int srad_drn_train_workspace_bytes(const srad_drn_t* h, int B, int H, int W, size_t* bytes) {
  SRAD_REQUIRE(h && bytes && B > 0 && H > 0 && W > 0, "train_workspace_bytes: bad argument");
  *bytes = plan_train_ws(h, B, H, W, nullptr, 0).bytes;
  return SRAD_OK;
}

// Gradient buckets of the flat buffer in the order srad_drn_backward completes them: 0 = the tail convolutions, 1 .. P = the
// up phases finest first (RCAB chain + upsampler of one level each), P + 1 = everything before them in the table (MeanShift
// layers, head, down blocks).  A data-parallel trainer starts a bucket's all-reduce from srad_drn_backward's hook.
int srad_drn_num_buckets(const srad_drn_t* h) { return h ? h->phase + 2 : 0; }

int srad_drn_bucket_range(srad_drn_t* h, int bucket, int64_t* off_floats, int64_t* n_floats) {
  SRAD_REQUIRE(h && off_floats && n_floats, "drn_bucket_range: null argument");
  int64_t tot = 0;
  SRAD_TRY(train_param_floats(h->pt, h->ts, &tot));
  const int P = h->phase;
  SRAD_REQUIRE(bucket >= 0 && bucket < P + 2, "drn_bucket_range: bucket %d out of range", bucket);
  auto start_of_phase = [&](int idx) { return idx < P ? h->ts.flat_off[h->rcab[idx][0].c0.w] : h->ts.flat_off[h->tail[0].w]; };
  int64_t a, b;
  if (bucket == 0) { a = start_of_phase(P); b = tot; }
  else if (bucket <= P) { const int idx = P - bucket; a = start_of_phase(idx); b = start_of_phase(idx + 1); }
  else { a = 0; b = start_of_phase(0); }
  *off_floats = a; *n_floats = b - a;
  return SRAD_OK;
}

// Training-mode DRN.forward (src/drn.py:241-270): as srad_drn_forward, every tensor the backward needs stays in `workspace`.
int srad_drn_forward_train(srad_drn_t* h, const float* x, int B, int H, int W, float* const* ys, int n_out, void* workspace,
                           size_t workspace_bytes, void* stream) {
  SRAD_REQUIRE(h && x && ys && workspace, "drn_forward_train: null argument");
  SRAD_REQUIRE(n_out == h->phase + 1, "drn_forward_train: %d outputs given, the model returns %d", n_out, h->phase + 1);
  SRAD_TRY(drn_train_check(h, B, H, W));
  SRAD_REQUIRE(((uintptr_t)workspace & 255) == 0, "drn_forward_train: workspace must be 256-byte aligned");
  const DrnTrainWs w = plan_train_ws(h, B, H, W, workspace, workspace_bytes);
  SRAD_REQUIRE(w.bytes <= workspace_bytes, "drn_forward_train: workspace %zu bytes, %zu needed", workspace_bytes, w.bytes);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const srad_drn_config& c = h->cfg;
  const int prec = c.precision, P = h->phase, F = c.n_feats, sc = c.scale, C = c.n_colors;
  const int H0 = H * sc, W0 = W * sc, top = F << P;
  const int F0 = srad_round_up(F, 4);
  auto fw = [&](int L) { return L == 0 ? F0 : F << L; };  // stored feature width of level L
  {
    const size_t tot = (size_t)B * H0 * W0;
    hipLaunchKernelGGL(bicubic_submean_kernel, dim3(grid1d(tot)), dim3(256), 0, s, x, w.up0, B, C, H, W, sc,
                       h->pt.fptr(h->sub_w), h->pt.fptr(h->sub_b), w.uraw);
    SRAD_CHECK_HIP(hipGetLastError());
  }
  {
    GemmParams p = conv_params(h, h->head, w.up0, SRAD_IMG_CPAD, B, H0, W0, 1, w.cat[0], 2 * F0, F0);
    p.Cin = SRAD_IMG_CPAD;
    SRAD_TRY(srad_launch_gemm(prec, p, s));
  }
  for (int L = 0; L < P; ++L) {
    const int f = fw(L), Hl = H0 >> L, Wl = W0 >> L;
    GemmParams p = conv_params(h, h->down_s2[L], w.cat[L] + f, 2 * f, B, Hl, Wl, 2, w.dtmp[L], f, 0);
    p.act = SRAD_ACT_LRELU; p.slope = c.negval;
    SRAD_TRY(srad_launch_gemm(prec, p, s));
    float* dst = L + 1 < P ? w.cat[L + 1] : w.deep;
    const int f1 = F << (L + 1);
    GemmParams q = conv_params(h, h->down_s1[L], w.dtmp[L], f, B, Hl / 2, Wl / 2, 1, dst, L + 1 < P ? 2 * f1 : f1, L + 1 < P ? f1 : 0);
    SRAD_TRY(srad_launch_gemm(prec, q, s));
  }
  auto tail_out = [&](int j, const float* X, int ldx, int Hh, int Ww) -> int {
    GemmParams p = conv_params(h, h->tail[j], X, ldx, B, Hh, Ww, 1, w.timg[j], SRAD_IMG_CPAD, 0);
    SRAD_TRY(srad_launch_gemm(prec, p, s));
    const size_t tot = (size_t)B * Hh * Ww;
    hipLaunchKernelGGL(affine_to_nchw_kernel, dim3(grid1d(tot)), dim3(256), 0, s, w.timg[j], SRAD_IMG_CPAD, ys[j], B, C, Hh * Ww,
                       h->pt.fptr(h->add_w), h->pt.fptr(h->add_b));
    SRAD_CHECK_HIP(hipGetLastError());
    return SRAD_OK;
  };
  SRAD_TRY(tail_out(0, w.deep, top, H0 >> P, W0 >> P));
  const float* xin = w.deep;
  int ldin = top;
  for (int idx = 0; idx < P; ++idx) {
    const int lvl = P - idx, Hl = H0 >> lvl, Wl = W0 >> lvl;
    const int ch = h->rcab[idx][0].ch;
    // bf16 chain (drn_level_bf16): relu(conv), the conv result and the block outputs are saved as bf16 arrays in the same slots
    // (what the MFMAs of the forward AND of the backward read are these bf16 values either way); the level's input and its last
    // block's output - operands of the generic GEMMs - stay fp32
    const bool lh = drn_level_bf16(prec, B, Hl, Wl, ch);
    bool x_h = false;
    for (int b = 0; b < c.n_blocks; ++b) {
      const RcabW& r = h->rcab[idx][b];
      const RcabSave& sv = w.rc[idx][b];
      {
        GemmParams p = conv_params(h, r.c0, x_h ? nullptr : xin, ldin, B, Hl, Wl, 1, sv.t, ch, 0);
        p.act = SRAD_ACT_RELU;
        if (lh) {
          if (x_h) p.Xh = reinterpret_cast<const __bf16*>(xin);
          p.Yh = reinterpret_cast<__bf16*>(sv.t);
          SRAD_REQUIRE(srad_conv80_supported(prec, p), "drn_forward_train: bf16 chain without the 80-channel conv kernel");
        }
        SRAD_TRY(srad_launch_gemm(prec, p, s));
      }
      int nchunk = DRN_POOL_CHUNKS;
      {
        GemmParams p = conv_params(h, r.c1, sv.t, ch, B, Hl, Wl, 1, sv.r, ch, 0);
        if (lh) { p.Xh = reinterpret_cast<const __bf16*>(sv.t); p.Yh = reinterpret_cast<__bf16*>(sv.r); }
        nchunk = pool_rows_per_image(prec, p, Hl * Wl);      // the pool's partial rows from the conv's epilogue when the tiles allow
        if (nchunk > 0) p.pool_part = w.ppart;
        SRAD_REQUIRE(!lh || (nchunk > 0 && srad_conv80_supported(prec, p)), "drn_forward_train: bf16 chain without the conv epilogue's pool sums");
        SRAD_TRY(srad_launch_gemm(prec, p, s));
      }
      if (nchunk == 0) {
        nchunk = DRN_POOL_CHUNKS;
        hipLaunchKernelGGL(pool_dot_kernel<false>, dim3(DRN_POOL_CHUNKS, B), dim3(256), 0, s, (const float*)nullptr, sv.r, w.ppart, Hl * Wl, ch,
                           DRN_POOL_CHUNKS);
      }
      const bool y_h = lh && b + 1 < c.n_blocks;
      {
        launch_ca_scale_add(lh, x_h, y_h, ca_slices(Hl * Wl), B, Hl * Wl, ch, s, (const float*)w.ppart, nchunk, 1.0f / (float)(Hl * Wl), ch, ch / 16,
                            (const float*)h->pt.fptr(r.w1), (const float*)h->pt.fptr(r.b1), (const float*)h->pt.fptr(r.w2), (const float*)h->pt.fptr(r.b2),
                            (float*)sv.gate, (float*)sv.pool, (const float*)sv.r, (const float*)xin, ldin, (float*)sv.xo, Hl * Wl);
      }
      SRAD_CHECK_HIP(hipGetLastError());
      xin = sv.xo; ldin = ch; x_h = y_h;
    }
    const int cout = fw(lvl - 1);
    {
      GemmParams p = conv_params(h, h->up_conv[idx], xin, ldin, B, Hl, Wl, 1, w.ups[idx], ch, 0);
      p.ps = 2;
      SRAD_TRY(srad_launch_gemm(prec, p, s));
    }
    {
      GemmParams p = conv_params(h, h->up_1x1[idx], w.ups[idx], ch, B, 2 * Hl, 2 * Wl, 1, w.cat[lvl - 1], 2 * cout, 0);
      SRAD_TRY(srad_launch_gemm(prec, p, s));
    }
    xin = w.cat[lvl - 1];
    ldin = 2 * cout;
    SRAD_TRY(tail_out(idx + 1, xin, ldin, 2 * Hl, 2 * Wl));
  }
  return SRAD_OK;
}

// Backward of srad_drn_forward_train: dys[j] = dLoss/d(output j) (NCHW, null = this output does not enter the loss);
// parameter gradients are ACCUMULATED into flat_grad (offsets as the flat parameter buffer).  `on_bucket(user, bucket)`
// (optional) is called on the host right after the last kernel that writes `bucket` (srad_drn_bucket_range) is enqueued.
int srad_drn_backward(srad_drn_t* h, const float* const* dys, int n_out, int B, int H, int W, float* flat_grad,
                      void* workspace, size_t workspace_bytes, void* stream, srad_bucket_fn on_bucket, void* user) {
  SRAD_REQUIRE(h && dys && flat_grad && workspace, "drn_backward: null argument");
  SRAD_REQUIRE(n_out == h->phase + 1, "drn_backward: %d gradients given, the model returns %d outputs", n_out, h->phase + 1);
  SRAD_TRY(drn_train_check(h, B, H, W));
  const DrnTrainWs w = plan_train_ws(h, B, H, W, workspace, workspace_bytes);
  SRAD_REQUIRE(w.bytes <= workspace_bytes, "drn_backward: workspace %zu bytes, %zu needed", workspace_bytes, w.bytes);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const srad_drn_config& c = h->cfg;
  const int prec = c.precision, P = h->phase, F = c.n_feats, sc = c.scale, C = c.n_colors;
  const int H0 = H * sc, W0 = W * sc, top = F << P;
  const int F0 = srad_round_up(F, 4);
  auto fw = [&](int L) { return L == 0 ? F0 : F << L; };  // stored feature width of level L
  const size_t T0 = (size_t)B * H0 * W0;
  float* G = flat_grad;
  WgradQueue wq = train_wgrad_queue(h->ts);
  // Two streams for the RCAB chains (160 of the 170 convolutions): the data-gradient chain stays on the caller's stream, the
  // two weight gradients of a block and their reduce run on a side stream while the next block's chain proceeds (dr / dt
  // and the split-K workspace are double-buffered; a block waits for the side work of the block two before it).
  // SRAD_BWD_ONE_STREAM=1 keeps everything on the caller's stream.
  static const bool one_stream = getenv("SRAD_BWD_ONE_STREAM") != nullptr;
  if (!one_stream && !h->side) SRAD_CHECK_HIP(hipStreamCreateWithFlags(&h->side, hipStreamNonBlocking));
  hipStream_t side = one_stream ? s : h->side;
  size_t ev_next = 0;
  auto next_event = [&](hipEvent_t* out) -> int {
    if (ev_next == h->events.size()) {
      hipEvent_t ev = nullptr;
      SRAD_CHECK_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
      h->events.push_back(ev);
    }
    *out = h->events[ev_next++];
    return SRAD_OK;
  };
  auto a_waits_b = [&](hipStream_t a, hipStream_t b) -> int {
    if (a == b) return SRAD_OK;
    hipEvent_t ev = nullptr;
    SRAD_TRY(next_event(&ev));
    SRAD_CHECK_HIP(hipEventRecord(ev, b));
    SRAD_CHECK_HIP(hipStreamWaitEvent(a, ev, 0));
    return SRAD_OK;
  };
  float* const wq_base = wq.ws;
  const size_t wq_half = wq.ws_floats / 2;
  hipEvent_t side_done[2] = {nullptr, nullptr};
  int blk_count = 0;

  for (int L = 0; L < P; ++L) SRAD_CHECK_HIP(hipMemsetAsync(w.gcat[L], 0, (T0 >> (2 * L)) * 2 * fw(L) * sizeof(float), s));
  SRAD_CHECK_HIP(hipMemsetAsync(w.gdeep, 0, (T0 >> (2 * P)) * top * sizeof(float), s));

  // ---- tails + add_mean (drn.py:256-258, 265-267): every output's gradient lands in the buffer its tail conv read ----
  for (int j = 0; j <= P; ++j) {
    if (!dys[j]) continue;
    const int lvl_in = j == 0 ? P : P - j;                 // level of the pixels the tail conv runs on
    const int Hh = H0 >> lvl_in, Ww = W0 >> lvl_in;
    const float* X = j == 0 ? w.deep : w.cat[lvl_in];
    float* GX = j == 0 ? w.gdeep : w.gcat[lvl_in];
    const int ld = j == 0 ? top : 2 * fw(lvl_in);
    const size_t tot = (size_t)B * Hh * Ww;
    hipLaunchKernelGGL(affine_bwd_kernel, dim3(grid1d(tot) > 256 ? 256 : grid1d(tot)), dim3(256), 0, s, dys[j], (const float*)nullptr,
                       w.timg[j], SRAD_IMG_CPAD, w.dtimg, B, C, Hh * Ww, h->pt.fptr(h->add_w), G + h->ts.flat_off[h->add_w],
                       G + h->ts.flat_off[h->add_b]);
    SRAD_CHECK_HIP(hipGetLastError());
    WgradParams g = drn_wgrad(h, h->tail[j], G, w.dtimg, SRAD_IMG_CPAD, 0, X, ld, B, Hh, Ww, 1);
    SRAD_TRY(srad_launch_wgrad(prec, g, wq, s));
    GemmParams p = drn_dgrad(h, h->tail[j], w.dtimg, SRAD_IMG_CPAD, B, Hh, Ww, GX, ld, 0);
    accumulate_into(p);
    SRAD_TRY(srad_launch_gemm(prec, p, s));
    SRAD_TRY(srad_wgrad_flush(wq, s));                       // dtimg is reused by the next output
  }
  if (on_bucket) on_bucket(user, 0);                         // (add_mean's own gradients travel with the last bucket)

  // ---- up phases, finest first (drn.py:260-268) ----
  for (int idx = P - 1; idx >= 0; --idx) {
    const int lvl = P - idx, Hl = H0 >> lvl, Wl = W0 >> lvl;
    const int ch = h->rcab[idx][0].ch;
    const size_t T = (size_t)B * Hl * Wl;
    const int cout = fw(lvl - 1);
    const float* drive = w.gcat[lvl - 1];                    // columns [0, cout): gradient of the 1x1 conv's output
    {
      WgradParams g = drn_wgrad(h, h->up_1x1[idx], G, drive, 2 * cout, 0, w.ups[idx], ch, B, 2 * Hl, 2 * Wl, 1);
      SRAD_TRY(srad_launch_wgrad(prec, g, wq, s));
      GemmParams p = drn_dgrad(h, h->up_1x1[idx], drive, 2 * cout, B, 2 * Hl, 2 * Wl, w.dups, ch, 0);
      SRAD_TRY(srad_launch_gemm(prec, p, s));
    }
    SRAD_TRY(srad_launch_unshuffle(w.dups, w.dus, B, Hl, Wl, ch, s));
    const float* chain_out = w.rc[idx][c.n_blocks - 1].xo;
    {
      // ch -> 4 ch, nine taps: four ch -> ch column blocks of dY when the nine-tap kernel takes them (the tiled kernel runs this
      // layer at ~90 TFLOP/s: 0.64 ms at 128 px against 4 x 45 us)
      WgradParams g = drn_wgrad(h, h->up_conv[idx], G, w.dus, 4 * ch, 0, chain_out, ch, B, Hl, Wl, 1);
      WgradParams gq = g;
      gq.N = gq.n_real = ch;
      if (prec == SRAD_PREC_BF16 && g.N == 4 * ch && g.n_real == g.N && g.cin_real == ch && srad_wgrad_conv9_supported(gq)) {
        for (int q = 0; q < 4; ++q) {
          gq.ycol0 = q * ch;
          gq.dW = g.dW + (size_t)q * ch * ch * 9;
          gq.db = g.db ? g.db + q * ch : nullptr;
          SRAD_TRY(srad_launch_wgrad(prec, gq, wq, s));
        }
      } else {
        SRAD_TRY(srad_launch_wgrad(prec, g, wq, s));
      }
      GemmParams p = drn_dgrad(h, h->up_conv[idx], w.dus, 4 * ch, B, Hl, Wl, w.ga, ch, 0);
      SRAD_TRY(srad_launch_gemm(prec, p, s));
    }
    float* ga = w.ga;
    float* gb = w.gb;
    const float* x0 = idx == 0 ? w.deep : w.cat[lvl];       // input of the first RCAB
    const int ld0 = idx == 0 ? top : ch;
    const bool lh = drn_level_bf16(prec, B, Hl, Wl, ch);    // as the forward: bf16 saves along this level's chain
    SRAD_TRY(srad_wgrad_flush(wq, s));                       // the up convs' partials: reduced on the caller's stream
    for (int b = c.n_blocks - 1; b >= 0; --b) {              // RCAB (drn.py:143-158)
      const RcabW& r = h->rcab[idx][b];
      const RcabSave& sv = w.rc[idx][b];
      const float* xin = b == 0 ? x0 : w.rc[idx][b - 1].xo;
      const int ldin = b == 0 ? ld0 : ch;
      const int set = blk_count & 1;
      if (side != s && side_done[set]) SRAD_CHECK_HIP(hipStreamWaitEvent(s, side_done[set], 0));   // block n - 2 fully consumed
      float* dr = w.dr2[set];
      float* dt = w.dt2[set];
      const bool x_h = lh && b > 0;                            // this block's input is the previous block's bf16 output
      float* const pp = w.ppart2[set];
      if (lh) hipLaunchKernelGGL(pool_dot_kernel<true>, dim3(DRN_POOL_CHUNKS, B), dim3(256), 0, s, ga, sv.r, pp, Hl * Wl, ch, DRN_POOL_CHUNKS);
      else hipLaunchKernelGGL(pool_dot_kernel<false>, dim3(DRN_POOL_CHUNKS, B), dim3(256), 0, s, ga, sv.r, pp, Hl * Wl, ch, DRN_POOL_CHUNKS);
      {
        static const bool no_pre = getenv("SRAD_DRN_CA_NO_PRELOAD") != nullptr;
        const int slices = ca_slices(Hl * Wl);
        const int items = ((Hl * Wl + slices - 1) / slices) * (ch / 4);
        const int ni = no_pre || items > 2560 ? 0 : (items <= 1280 ? 5 : 10);
        const dim3 grid(slices, B);
        auto launch = [&](auto kern) {
          hipLaunchKernelGGL(kern, grid, dim3(256), 0, s, ga, sv.gate, pp, DRN_POOL_CHUNKS, sv.pool, 1.0f / (float)(Hl * Wl), ch, ch / 16,
                             h->pt.fptr(r.w1), h->pt.fptr(r.b1), h->pt.fptr(r.w2), dr, Hl * Wl);
        };
        if (lh) { if (ni == 5) launch(ca_apply_bwd_kernel<true, 5>); else if (ni == 10) launch(ca_apply_bwd_kernel<true, 10>); else launch(ca_apply_bwd_kernel<true, 0>); }
        else { if (ni == 5) launch(ca_apply_bwd_kernel<false, 5>); else if (ni == 10) launch(ca_apply_bwd_kernel<false, 10>); else launch(ca_apply_bwd_kernel<false, 0>); }
      }
      SRAD_CHECK_HIP(hipGetLastError());
      {
        GemmParams p = drn_dgrad(h, r.c1, dr, ch, B, Hl, Wl, dt, ch, 0);
        p.ldr = ch; p.rmode = SRAD_RMODE_DLRELU; p.slope = 0.f;                       // through the ReLU
        if (lh) {     // dr, the saved relu(conv) (only its sign is used) and dt as bf16 arrays: MFMA operands of this conv / the two weight gradients / the next conv
          p.X = nullptr; p.Xh = reinterpret_cast<const __bf16*>(dr);
          p.Rh = reinterpret_cast<const __bf16*>(sv.t);
          p.Yh = reinterpret_cast<__bf16*>(dt);
          SRAD_REQUIRE(srad_conv80_supported(prec, p), "drn_backward: bf16 chain without the 80-channel conv kernel");
        } else {
          p.R = sv.t;
        }
        SRAD_TRY(srad_launch_gemm(prec, p, s));
      }
      // both weight gradients of the block + their reduce on the side stream (dr and dt exist now)
      SRAD_TRY(a_waits_b(side, s));
      wq.ws = wq_base + (size_t)set * wq_half; wq.ws_floats = wq_half;
      // the channel attention's weight / bias gradients (one workgroup, all images): nothing on the data path waits for them
      hipLaunchKernelGGL(ca_bwd_kernel, dim3(1), dim3(256), 0, side, pp, DRN_POOL_CHUNKS, sv.pool, sv.gate, 1.0f / (float)(Hl * Wl), B,
                         ch, ch / 16, h->pt.fptr(r.w1), h->pt.fptr(r.b1), h->pt.fptr(r.w2), G + h->ts.flat_off[r.w1],
                         G + h->ts.flat_off[r.b1], G + h->ts.flat_off[r.w2], G + h->ts.flat_off[r.b2], w.dpool2[set]);
      SRAD_CHECK_HIP(hipGetLastError());
      {
        WgradParams g = drn_wgrad(h, r.c1, G, dr, ch, 0, sv.t, ch, B, Hl, Wl, 1);
        WgradParams g0 = drn_wgrad(h, r.c0, G, dt, ch, 0, xin, ldin, B, Hl, Wl, 1);
        if (lh) {
          g.dy_bf16 = 1; g.x_bf16 = 1; g0.dy_bf16 = 1; g0.x_bf16 = x_h ? 1 : 0;
          SRAD_REQUIRE(srad_wgrad_conv9_supported(g) && srad_wgrad_conv9_supported(g0), "drn_backward: bf16 chain without the nine-tap weight-gradient kernel");
        }
        SRAD_TRY(srad_launch_wgrad(prec, g, wq, side));
        SRAD_TRY(srad_launch_wgrad(prec, g0, wq, side));
        SRAD_TRY(srad_wgrad_flush(wq, side));
      }
      if (side != s) {
        SRAD_TRY(next_event(&side_done[set]));
        SRAD_CHECK_HIP(hipEventRecord(side_done[set], side));
      }
      {
        GemmParams p = drn_dgrad(h, r.c0, dt, ch, B, Hl, Wl, gb, ch, 0);
        p.R = ga; p.ldr = ch;                                                        // + the skip path
        if (lh) { p.X = nullptr; p.Xh = reinterpret_cast<const __bf16*>(dt); }
        SRAD_TRY(srad_launch_gemm(prec, p, s));
      }
      float* t = ga; ga = gb; gb = t;
      ++blk_count;
    }
    SRAD_TRY(a_waits_b(s, side));                           // the chain's weight gradients are final on the caller's stream
    wq.ws = wq_base; wq.ws_floats = 2 * wq_half;
    // gradient of the chain's input: deep (idx 0) or the concat buffer of this level
    float* GX = idx == 0 ? w.gdeep : w.gcat[lvl];
    hipLaunchKernelGGL(add_cols_kernel, dim3(grid1d(T * ch / 4)), dim3(256), 0, s, ga, ch, GX, ch, T, ch);
    SRAD_CHECK_HIP(hipGetLastError());
    if (on_bucket) on_bucket(user, P - idx);
  }

  // ---- down path (drn.py:250-253, DownBlock 83-119) ----
  for (int L = P - 1; L >= 0; --L) {
    const int f = fw(L), f1 = F << (L + 1), Hl = H0 >> L, Wl = W0 >> L;
    const float* gout = L + 1 < P ? w.gcat[L + 1] + f1 : w.gdeep;
    const int ldg = L + 1 < P ? 2 * f1 : f1;
    {
      WgradParams g = drn_wgrad(h, h->down_s1[L], G, gout, ldg, 0, w.dtmp[L], f, B, Hl / 2, Wl / 2, 1);
      SRAD_TRY(srad_launch_wgrad(prec, g, wq, s));
      GemmParams p = drn_dgrad(h, h->down_s1[L], gout, ldg, B, Hl / 2, Wl / 2, w.ddtmp, f, 0);
      p.R = w.dtmp[L]; p.ldr = f; p.rmode = SRAD_RMODE_DLRELU; p.slope = c.negval;     // through the LeakyReLU
      SRAD_TRY(srad_launch_gemm(prec, p, s));
    }
    {
      WgradParams g = drn_wgrad(h, h->down_s2[L], G, w.ddtmp, f, 0, w.cat[L] + f, 2 * f, B, Hl, Wl, 2);
      SRAD_TRY(srad_launch_wgrad(prec, g, wq, s));
      const size_t tot = (size_t)B * Hl * Wl * f / 4;
      hipLaunchKernelGGL(zero_upsample_kernel, dim3(grid1d(tot)), dim3(256), 0, s, w.ddtmp, w.zup, B, Hl / 2, Wl / 2, f);
      SRAD_CHECK_HIP(hipGetLastError());
      GemmParams p = drn_dgrad(h, h->down_s2[L], w.zup, f, B, Hl, Wl, w.gcat[L], 2 * f, f);
      accumulate_into(p);
      SRAD_TRY(srad_launch_gemm(prec, p, s));
    }
    SRAD_TRY(srad_wgrad_flush(wq, s));
  }
  // ---- head + sub_mean (drn.py:243-247) ----
  {
    WgradParams g = drn_wgrad(h, h->head, G, w.gcat[0] + F0, 2 * F0, 0, w.up0, SRAD_IMG_CPAD, B, H0, W0, 1);
    SRAD_TRY(srad_launch_wgrad(prec, g, wq, s));
    GemmParams p = drn_dgrad(h, h->head, w.gcat[0] + F0, 2 * F0, B, H0, W0, w.dup0, SRAD_IMG_CPAD, 0);
    SRAD_TRY(srad_launch_gemm(prec, p, s));
    const size_t tot = (size_t)B * H0 * W0;
    hipLaunchKernelGGL(affine_bwd_kernel, dim3(grid1d(tot) > 256 ? 256 : grid1d(tot)), dim3(256), 0, s, (const float*)nullptr, w.dup0,
                       w.uraw, SRAD_IMG_CPAD, (float*)nullptr, B, C, H0 * W0, h->pt.fptr(h->sub_w), G + h->ts.flat_off[h->sub_w],
                       G + h->ts.flat_off[h->sub_b]);
    SRAD_CHECK_HIP(hipGetLastError());
  }
  SRAD_TRY(srad_wgrad_flush(wq, s));
  if (on_bucket) on_bucket(user, P + 1);
  return SRAD_OK;
}

// Backward of the dual regression model (srad_dual_forward): dy [B,C,H/2,W/2] -> dw0 / dw1 accumulated (PyTorch layouts),
// dx [B,C,H,W] optional.  The hidden activation is recomputed.  The hidden width is stored rounded up to a multiple of 4
// (x8 preset: 10 -> 12) with exact-zero pad columns, as in srad_dual_forward.
int srad_dual_backward_workspace_bytes(int B, int C, int H, int W, int n_feats, size_t* bytes) {
  SRAD_REQUIRE(bytes && B > 0 && C > 0 && H > 0 && W > 0 && n_feats > 0, "dual_backward_workspace_bytes: bad argument");
  const size_t T = (size_t)B * H * W, T2 = (size_t)B * (H / 2) * (W / 2);
  const int fm = srad_round_up(n_feats, 4);
  const size_t pk = srad_align_up(srad_packed_bytes(SRAD_PREC_F32, fm, fm, 9), 256);   // covers [F][4], [4][F] and their transposes
  *bytes = 2 * srad_align_up(T * SRAD_IMG_CPAD * 4, 256) + 2 * srad_align_up(T2 * fm * 4, 256) + srad_align_up(T2 * SRAD_IMG_CPAD * 4, 256) +
           srad_align_up(T * fm * 4, 256) + 4 * pk + srad_align_up(SRAD_WGRAD_WS_BYTES, 256);
  return SRAD_OK;
}

int srad_dual_backward(const float* w0, const float* w1, int C, int n_feats, float negval, const float* x, int B, int H,
                       int W, const float* dy, float* dx, float* dw0, float* dw1, void* workspace, size_t workspace_bytes,
                       int precision, void* stream) {
  SRAD_REQUIRE(w0 && w1 && x && dy && dw0 && dw1 && workspace, "dual_backward: null argument");
  SRAD_REQUIRE((C == 1 || C == 3) && n_feats > 0 && H % 2 == 0 && W % 2 == 0, "dual_backward: needs 1 or 3 channels, even H and W");
  size_t need = 0;
  SRAD_TRY(srad_dual_backward_workspace_bytes(B, C, H, W, n_feats, &need));
  SRAD_REQUIRE(workspace_bytes >= need && ((uintptr_t)workspace & 255) == 0, "dual_backward: workspace %zu bytes, %zu needed (256-byte aligned)", workspace_bytes, need);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  Bump bp(workspace, workspace_bytes);
  const int H2 = H / 2, W2 = W / 2, Fh = srad_round_up(n_feats, 4);   // hidden width as stored
  const size_t T = (size_t)B * H * W, T2 = (size_t)B * H2 * W2;
  float* xin = bp.take(T * SRAD_IMG_CPAD);
  float* dxin = bp.take(T * SRAD_IMG_CPAD);
  float* mid = bp.take(T2 * Fh);
  float* dmid = bp.take(T2 * Fh);
  float* dyn = bp.take(T2 * SRAD_IMG_CPAD);
  float* zup = bp.take(T * Fh);
  const size_t pk = srad_packed_bytes(SRAD_PREC_F32, Fh, Fh, 9) / 4;
  void* p0 = bp.take(pk);     // w0 forward   [F][C]
  void* p1t = bp.take(pk);    // w1 transposed (rows F, K = C)
  void* p0t = bp.take(pk);    // w0 transposed (rows C, K = F)
  WgradQueue wq;
  wq.ws = bp.take(SRAD_WGRAD_WS_BYTES / 4);
  wq.ws_floats = SRAD_WGRAD_WS_BYTES / sizeof(float);
  const float zero3[3] = {0.f, 0.f, 0.f};
  SRAD_TRY(srad_launch_nchw_to_nhwc(x, xin, B, C, SRAD_IMG_CPAD, H, W, zero3, 1.0f, s));
  SRAD_TRY(srad_launch_nchw_to_nhwc(dy, dyn, B, C, SRAD_IMG_CPAD, H2, W2, zero3, 1.0f, s));
  SRAD_TRY(srad_launch_pack_weight_padded(precision, w0, p0, n_feats, C, 9, Fh, 0, 0, s));
  SRAD_TRY(srad_launch_pack_weight_transposed(precision, w1, p1t, C, n_feats, 9, SRAD_IMG_CPAD, Fh, s));
  SRAD_TRY(srad_launch_pack_weight_transposed(precision, w0, p0t, n_feats, C, 9, Fh, SRAD_IMG_CPAD, s));
  {  // recompute mid = lrelu(conv_s2(x))
    GemmParams a{};
    a.X = xin; a.ldx = SRAD_IMG_CPAD; a.Cin = SRAD_IMG_CPAD; a.Cp = srad_cp(SRAD_IMG_CPAD); a.ntaps = 9;
    a.Hi = H; a.Wi = W; a.Ho = H2; a.Wo = W2; a.stride = 2; a.M = (int)T2; a.ln_eps = 1e-5f;
    a.Wp = p0; a.N = Fh; a.act = SRAD_ACT_LRELU; a.slope = negval; a.alpha = 1.f; a.Y = mid; a.ldy = Fh;
    SRAD_TRY(srad_launch_gemm(precision, a, s));
  }
  {  // second conv: dw1, dmid = (dy . w1) * lrelu'(mid)
    WgradParams g{};
    g.dY = dyn; g.ldy = SRAD_IMG_CPAD; g.X = mid; g.ldx = Fh; g.M = (int)T2; g.N = SRAD_IMG_CPAD; g.Cin = Fh; g.ntaps = 9;
    g.n_real = C; g.cin_real = n_feats; g.Hi = g.Ho = H2; g.Wi = g.Wo = W2; g.stride = 1; g.alpha = 1.f; g.dW = dw1;
    SRAD_TRY(srad_launch_wgrad(precision, g, wq, s));
    GemmParams p{};
    p.Hi = p.Ho = H2; p.Wi = p.Wo = W2; p.stride = 1; p.X = dyn; p.ldx = SRAD_IMG_CPAD; p.M = (int)T2; p.Cin = SRAD_IMG_CPAD;
    p.Cp = srad_cp(SRAD_IMG_CPAD); p.ntaps = 9; p.ln_eps = 1e-5f; p.Wp = p1t; p.N = Fh; p.alpha = 1.f;
    p.R = mid; p.ldr = Fh; p.rmode = SRAD_RMODE_DLRELU; p.slope = negval; p.Y = dmid; p.ldy = Fh;
    SRAD_TRY(srad_launch_gemm(precision, p, s));
  }
  {  // first conv (stride 2): dw0, dx
    WgradParams g{};
    g.dY = dmid; g.ldy = Fh; g.X = xin; g.ldx = SRAD_IMG_CPAD; g.M = (int)T2; g.N = Fh; g.Cin = SRAD_IMG_CPAD; g.ntaps = 9;
    g.n_real = n_feats; g.cin_real = C; g.Hi = H; g.Wi = W; g.Ho = H2; g.Wo = W2; g.stride = 2; g.alpha = 1.f; g.dW = dw0;
    SRAD_TRY(srad_launch_wgrad(precision, g, wq, s));
    if (dx) {
      const size_t tot = T * Fh / 4;
      hipLaunchKernelGGL(zero_upsample_kernel, dim3(grid1d(tot)), dim3(256), 0, s, dmid, zup, B, H2, W2, Fh);
      SRAD_CHECK_HIP(hipGetLastError());
      GemmParams p{};
      p.Hi = p.Ho = H; p.Wi = p.Wo = W; p.stride = 1; p.X = zup; p.ldx = Fh; p.M = (int)T; p.Cin = Fh; p.Cp = srad_cp(Fh);
      p.ntaps = 9; p.ln_eps = 1e-5f; p.Wp = p0t; p.N = SRAD_IMG_CPAD; p.alpha = 1.f; p.Y = dxin; p.ldy = SRAD_IMG_CPAD;
      SRAD_TRY(srad_launch_gemm(precision, p, s));
      SRAD_TRY(srad_launch_nhwc_to_nchw(dxin, SRAD_IMG_CPAD, dx, B, C, H, W, zero3, 1.0f, s));
    }
  }
  return srad_wgrad_flush(wq, s);
}

}  // extern "C"
