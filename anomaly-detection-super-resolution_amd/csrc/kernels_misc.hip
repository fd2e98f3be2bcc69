// kernels_misc.hip - HBM-bound helper kernels: LayerNorm rows, layout changes with the mean
// shift folded in, bicubic upsampling, channel-attention gate.  All are one pass over their
// data with coalesced accesses; they are a few percent of the forward's time.
#include "srad_common.h"
#include <stdlib.h>
#include <stdarg.h>
#include <stdio.h>
#include <vector>

// ---------------------------------------------------------------- error state (thread local)
static thread_local char g_err[512] = "";
int srad_set_error(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}
extern "C" const char* srad_last_error(void) { return g_err; }

// ---------------------------------------------------------------- event profiler
namespace {
struct ProfRec { hipEvent_t a, b; int cls; double flops, bytes; };
struct Prof {
  int enabled = 0;
  std::vector<ProfRec> recs;
  std::vector<hipEvent_t> pool;
  hipEvent_t get() {
    if (!pool.empty()) { hipEvent_t e = pool.back(); pool.pop_back(); return e; }
    hipEvent_t e = nullptr;
    (void)hipEventCreate(&e);
    return e;
  }
} g_prof;
const char* const kClassNames[SRAD_K_COUNT] = {"gemm_bn64", "gemm_bn32", "gemm_bn16", "window_attn", "layernorm",
                                               "layout", "pack_weight", "score", "misc", "mlp_block", "qkv_attn",
                                               "wgrad", "window_attn_bwd", "layernorm_bwd", "optim", "wgrad_reduce", "mlp_bwd", "ln_qkv", "conv80"};
}  // namespace

bool srad_no_xcd_map() {
  static const bool off = getenv("SRAD_NO_XCD_MAP") != nullptr;
  return off;
}

SradProfScope::SradProfScope(hipStream_t stream, int cls, double flops, double bytes) : s(stream), active(0) {
  if (!g_prof.enabled) return;
  hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
  if (stream && hipStreamIsCapturing(stream, &st) == hipSuccess && st != hipStreamCaptureStatusNone) return;
  ProfRec r{g_prof.get(), g_prof.get(), cls, flops, bytes};
  (void)hipEventRecord(r.a, stream);
  g_prof.recs.push_back(r);
  active = 1;
}
SradProfScope::~SradProfScope() {
  if (active) (void)hipEventRecord(g_prof.recs.back().b, s);
}

extern "C" int srad_prof_enable(int on) {
  g_prof.enabled = on;
  return SRAD_OK;
}
extern "C" int srad_prof_num_classes(void) { return SRAD_K_COUNT; }
extern "C" const char* srad_prof_class_name(int cls) { return cls >= 0 && cls < SRAD_K_COUNT ? kClassNames[cls] : ""; }
// Sums the recorded launches per class since the last collect (synchronises on the events).
// launches/ms/flops/bytes: arrays of srad_prof_num_classes() entries.
extern "C" int srad_prof_collect(int64_t* launches, double* ms, double* flops, double* bytes) {
  for (int i = 0; i < SRAD_K_COUNT; ++i) { launches[i] = 0; ms[i] = 0; flops[i] = 0; bytes[i] = 0; }
  for (ProfRec& r : g_prof.recs) {
    SRAD_CHECK_HIP(hipEventSynchronize(r.b));
    float t = 0.f;
    SRAD_CHECK_HIP(hipEventElapsedTime(&t, r.a, r.b));
    launches[r.cls] += 1; ms[r.cls] += t; flops[r.cls] += r.flops; bytes[r.cls] += r.bytes;
    g_prof.pool.push_back(r.a);
    g_prof.pool.push_back(r.b);
  }
  g_prof.recs.clear();
  return SRAD_OK;
}

namespace {

// One wave per row: LayerNorm over C contiguous channels (reference nn.LayerNorm, eps 1e-5;
// src/drct.py:798,833 patch_embed.norm / norm).  Rows are 16-byte aligned and C % 4 == 0, so lane l holds
// channels [4 l, 4 l + 4) and [256 + 4 l, ..): the row is read once with 16-byte loads and stays in registers.
__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ x, int ldx, float* __restrict__ y,
                                                        int ldy, int rows, int C, const float* __restrict__ g,
                                                        const float* __restrict__ b, float eps) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const f32x4 z4 = f32x4{0.f, 0.f, 0.f, 0.f};
  f32x4 xv[2], gv[2], bv[2];
  bool ok[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int c = 4 * lane + 256 * j, cc = min(c, C - 4);
    ok[j] = c < C;
    xv[j] = *reinterpret_cast<const f32x4*>(x + (size_t)row * ldx + cc);
    gv[j] = *reinterpret_cast<const f32x4*>(g + cc);
    bv[j] = *reinterpret_cast<const f32x4*>(b + cc);
  }
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < 2; ++j) { xv[j] = ok[j] ? xv[j] : z4; s += (xv[j][0] + xv[j][1]) + (xv[j][2] + xv[j][3]); }
  s = srad_wave_sum(s);
  const float mean = s / (float)C;
  float v = 0.f;
  f32x4 d[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    d[j] = ok[j] ? xv[j] - mean : z4;
    v += (d[j][0] * d[j][0] + d[j][1] * d[j][1]) + (d[j][2] * d[j][2] + d[j][3] * d[j][3]);
  }
  v = srad_wave_sum(v);
  const float rstd = rsqrtf(v / (float)C + eps);
#pragma unroll
  for (int j = 0; j < 2; ++j)
    if (ok[j]) *reinterpret_cast<f32x4*>(y + (size_t)row * ldy + 4 * lane + 256 * j) = d[j] * rstd * gv[j] + bv[j];
}

__global__ void nchw_to_nhwc_kernel(const float* __restrict__ x, float* __restrict__ y, int B, int C, int Cpad, int HW,
                                    float m0, float m1, float m2, float scale) {
  const size_t total = (size_t)B * Cpad * HW;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % Cpad);
    const size_t pix = i / Cpad;
    const int bb = (int)(pix / HW);
    const int hw = (int)(pix - (size_t)bb * HW);
    const float mean = c == 0 ? m0 : (c == 1 ? m1 : m2);
    const float v = x[((size_t)bb * C + min(c, C - 1)) * HW + hw];
    y[i] = c < C ? (v - mean) * scale : 0.f;       // channels [C, Cpad) are zero padding
  }
}

__global__ void nhwc_to_nchw_kernel(const float* __restrict__ x, int ldx, float* __restrict__ y, int B, int C, int HW,
                                    float m0, float m1, float m2, float scale) {
  const size_t total = (size_t)B * C * HW;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int hw = (int)(i % HW);
    const size_t bc = i / HW;
    const int c = (int)(bc % C);
    const int bb = (int)(bc / C);
    const float mean = c == 0 ? m0 : (c == 1 ? m1 : m2);
    y[i] = x[((size_t)bb * HW + hw) * ldx + c] * scale + mean;
  }
}

static inline int grid_for(size_t total) {
  size_t b = (total + 255) / 256;
  return (int)(b > 2048 ? 2048 : (b < 1 ? 1 : b));
}

}  // namespace

int srad_launch_layernorm(const float* x, int ldx, float* y, int ldy, int rows, int C, const float* g, const float* b,
                          float eps, hipStream_t stream) {
  SRAD_REQUIRE(rows > 0 && C >= 4 && C <= 512 && (C & 3) == 0, "layernorm: channel count %d unsupported (4..512, multiple of 4)", C);
  SRAD_REQUIRE(((ldx | ldy) & 3) == 0 && (((uintptr_t)x | (uintptr_t)y | (uintptr_t)g | (uintptr_t)b) & 15) == 0,
               "layernorm: rows must be 16-byte aligned (strides multiples of 4 floats)");
  SradProfScope prof(stream, SRAD_K_LAYERNORM, 8.0 * rows * C, 8.0 * rows * C);
  hipLaunchKernelGGL(layernorm_kernel, dim3((rows + 3) / 4), dim3(256), 0, stream, x, ldx, y, ldy, rows, C, g, b, eps);
  SRAD_CHECK_HIP(hipGetLastError());
  return SRAD_OK;
}

int srad_launch_nchw_to_nhwc(const float* x, float* y, int B, int C, int Cpad, int H, int W, const float* mean3,
                             float scale, hipStream_t stream) {
  SRAD_REQUIRE(C >= 1 && C <= 3 && Cpad >= C, "layout: channel count %d unsupported (1..3)", C);
  const size_t total = (size_t)B * Cpad * H * W;
  SradProfScope prof(stream, SRAD_K_LAYOUT, 2.0 * total, 8.0 * total);
  hipLaunchKernelGGL(nchw_to_nhwc_kernel, dim3(grid_for(total)), dim3(256), 0, stream, x, y, B, C, Cpad, H * W,
                     mean3[0], mean3[1], mean3[2], scale);
  SRAD_CHECK_HIP(hipGetLastError());
  return SRAD_OK;
}

int srad_launch_nhwc_to_nchw(const float* x, int ldx, float* y, int B, int C, int H, int W, const float* mean3,
                             float scale, hipStream_t stream) {
  SRAD_REQUIRE(C >= 1 && C <= 3, "layout: channel count %d unsupported (1..3)", C);
  const size_t total = (size_t)B * C * H * W;
  SradProfScope prof(stream, SRAD_K_LAYOUT, 2.0 * total, 8.0 * total);
  hipLaunchKernelGGL(nhwc_to_nchw_kernel, dim3(grid_for(total)), dim3(256), 0, stream, x, ldx, y, B, C, H * W, mean3[0],
                     mean3[1], mean3[2], scale);
  SRAD_CHECK_HIP(hipGetLastError());
  return SRAD_OK;
}
