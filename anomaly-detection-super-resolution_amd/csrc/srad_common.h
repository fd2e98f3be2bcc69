// srad_common.h - shared declarations for the gfx950 (CDNA4 / MI355X) kernels of libsrad.
// Written for wave64 + MFMA only; there is no other target.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

#define SRAD_OK 0
#define SRAD_ERR_ARG 1
#define SRAD_ERR_STATE 2
#define SRAD_ERR_HIP 3
#define SRAD_ERR_NOMEM 4

// SRAD_PREC_BF16X3: split-bf16.  Every MFMA operand x is held as two bf16 terms, hi = bf16(x) and lo = bf16(x - hi) (~16 mantissa
// bits), and a product is three v_mfma_f32_16x16x32_bf16 (hi.hi + hi.lo + lo.hi), fp32 accumulation: fp32-grade results (measured
// ~1e-5 per op against the fp32 oracle) from the bf16 matrix pipe.  Kernels without a split instance run their exact-fp32 form
// in this mode (every `prec == SRAD_PREC_BF16 ? bf16 : fp32` dispatch), so the mode is never less accurate than asked.
enum { SRAD_PREC_F32 = 0, SRAD_PREC_BF16 = 1, SRAD_PREC_BF16X3 = 2 };
enum { SRAD_ACT_NONE = 0, SRAD_ACT_GELU = 1, SRAD_ACT_LRELU = 2, SRAD_ACT_RELU = 3 };

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

int srad_set_error(int code, const char* fmt, ...);

#if defined(__HIPCC__)
// Sum over the 64 lanes of a wave, result in every lane.  The first four butterfly steps are DPP lane swizzles
// (VALU speed: quad_perm [1,0,3,2], [2,3,0,1], row_half_mirror, row_mirror); only the last two cross a 16-lane
// row and go through ds_bpermute.  A plain __shfl_xor chain is six dependent LDS round trips.
template <int CTRL>
__device__ __forceinline__ float srad_dpp(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
// Butterfly over the 16 lanes of a DPP row (the lanes that share lane >> 4), result in all 16: VALU only.
__device__ __forceinline__ float srad_row16_sum(float v) {
  v += srad_dpp<0xB1>(v);
  v += srad_dpp<0x4E>(v);
  v += srad_dpp<0x141>(v);
  v += srad_dpp<0x140>(v);
  return v;
}
// ... over aligned groups of 8 lanes
__device__ __forceinline__ float srad_row8_sum(float v) {
  v += srad_dpp<0xB1>(v);
  v += srad_dpp<0x4E>(v);
  v += srad_dpp<0x141>(v);
  return v;
}
__device__ __forceinline__ float srad_row16_max(float v) {
  v = fmaxf(v, srad_dpp<0xB1>(v));
  v = fmaxf(v, srad_dpp<0x4E>(v));
  v = fmaxf(v, srad_dpp<0x141>(v));
  v = fmaxf(v, srad_dpp<0x140>(v));
  return v;
}
// split-bf16 terms of four values: hi = bf16(v) (round to nearest even), lo = bf16(v - hi); hi + lo carries ~16 mantissa bits
__device__ __forceinline__ void srad_split4(const f32x4 v, bf16x4& hi, bf16x4& lo) {
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    hi[e] = (__bf16)v[e];
    lo[e] = (__bf16)(v[e] - (float)hi[e]);
  }
}
__device__ __forceinline__ float srad_wave_sum(float v) {
  v += srad_dpp<0xB1>(v);
  v += srad_dpp<0x4E>(v);
  v += srad_dpp<0x141>(v);
  v += srad_dpp<0x140>(v);
  v += __shfl_xor(v, 16);
  v += __shfl_xor(v, 32);
  return v;
}
#endif
// "dynamic LDS limit of this kernel instance is set" - per DEVICE: hipFuncSetAttribute applies to the current device only, so a
// process that drives two devices (not how this build runs - one rank per process - but legal) configures each once.
struct SradOncePerDevice {
  unsigned long long mask = 0, cur = 0;
  bool need() { int d = 0; (void)hipGetDevice(&d); cur = 1ull << (d & 63); return (mask & cur) == 0; }
  void done() { mask |= cur; }
};
static inline int srad_device_slot() { int d = 0; (void)hipGetDevice(&d); return d & 15; }

#define SRAD_CHECK_HIP(expr)                                                              \
  do {                                                                                    \
    hipError_t _e = (expr);                                                               \
    if (_e != hipSuccess)                                                                 \
      return srad_set_error(SRAD_ERR_HIP, "%s:%d %s -> %s", __FILE__, __LINE__, #expr,    \
                            hipGetErrorString(_e));                                       \
  } while (0)
#define SRAD_REQUIRE(cond, ...)                                                           \
  do {                                                                                    \
    if (!(cond)) return srad_set_error(SRAD_ERR_ARG, __VA_ARGS__);                        \
  } while (0)

#define SRAD_TRY(expr)            \
  do {                            \
    int _rc = (expr);             \
    if (_rc) return _rc;          \
  } while (0)

static inline size_t srad_align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }
static inline int srad_round_up(int v, int a) { return (v + a - 1) / a * a; }

// ------------------------------------------------------------------------------------------
// Row-gather GEMM:  Y[m][n] = epi( sum_{tap,c} A(m,tap,c) * W[n][tap*Cp + c] + bias[n] )
//   * Linear / 1x1 conv : ntaps = 1, A(m,0,c) = X[m*ldx + c]
//   * 3x3 conv (pad 1)  : ntaps = 9, A(m,tap,c) = X[pix(m,tap)*ldx + c] (0 outside the image),
//                         stride 1 or 2; activations are NHWC so c is contiguous
//   * optional LayerNorm over the Cin channels of each row fused into the A staging
//   * epilogue: +bias -> activation -> *alpha -> +R[m*ldr+n] -> store (plain / pixel-shuffle 2x)
// Weights are pre-packed by pack_weight_kernel: [Np][ntaps*Cp], zero padded, bf16 or fp32.
// ------------------------------------------------------------------------------------------
struct GemmParams {
  const float* X;
  int ldx;
  int M;
  int Cin;
  int Cp;
  int ntaps;
  int Hi, Wi, Ho, Wo, stride;
  const float* ln_g;
  const float* ln_b;
  float ln_eps;
  const void* Wp;
  int N;
  const float* bias;
  int act;
  float slope;
  float alpha;
  const float* R;
  int ldr;
  float* Y;
  int ldy;
  int yoff;
  int ps;          // 0: plain store, 2: PixelShuffle(2) scatter (N = 4 * ldy-channels)
  int hsplit_hd;   // > 0: column n is stored at (n / hsplit_hd) * hsplit_hdp + n % hsplit_hd, i.e. every
  int hsplit_hdp;  //      head's slice is padded to hsplit_hdp floats (the attention kernel's input layout)
  // hsplit output as bf16 for the one-row-per-chunk window attention (srad_window_attn_bf16_in): written to Yh instead of Y
  // (same element offsets), the q slices (heads [0, hsplit_heads)) times hsplit_qscale before rounding, and each slice's
  // padding columns written too: 0, except 1 in column hsplit_hd of the v slices (P.V then also sums the probabilities)
  __bf16* Yh = nullptr; int hsplit_heads = 0; float hsplit_qscale = 1.f;
  // optional second output: pool_part[m_tile][n] = sum of the tile's rows of Y (column sums per workgroup, plain stores,
  // fixed order) - the global average pool of DRN's channel attention without a pass of its own; an image must be a whole
  // number of row tiles (srad_gemm_tile_rows)
  float* pool_part = nullptr;
  // conv80 only (srad_conv80_supported): the input as bf16 (same ldx, in elements) / the output as bf16 through Yh without
  // hsplit - the ReLU output between an RCAB's two convolutions, which only the second one reads
  const __bf16* Xh = nullptr;
  const __bf16* Rh = nullptr;   // conv80 only: R as a bf16 array (ldr in elements) - the saved ReLU output as the backward's mask
  // conv80 only, sub-pixel mode (sp_q in 0..3): this launch is one of the four 80 -> 80 convolutions an 80 -> 320 convolution +
  // PixelShuffle(2) (DRN's Upsampler, src/drn.py:55-81) splits into: output channel c of the launch is row 4 c + sp_q of the
  // packed weight / bias, and lands at pixel (2 y + sp_q / 2, 2 x + sp_q % 2) of the [B][2 Ho][2 Wo][ldy] output
  int sp_q = -1;
  // ---- training extensions (all off when zero) ----
  int rmode;                 // how R enters: 0 v = act(acc)*alpha*rs + R (residual); 1 v = acc*alpha*rs * gelu'(R);
                             //               2 v = acc*alpha*rs * (R > 0 ? 1 : slope)   (backward through an activation)
  const float* row_scale;    // optional per-sample factor rs = row_scale[m / rps] (DropPath keep mask / keep_prob)
  int rps;                   // rows per sample
  float* Ypre;               // optional second output: acc + bias before the activation (same ldy / yoff as Y)
};
enum { SRAD_RMODE_ADD = 0, SRAD_RMODE_DGELU = 1, SRAD_RMODE_DLRELU = 2 };

int srad_launch_gemm(int prec, const GemmParams& p, hipStream_t stream);
int srad_gemm_tile_rows(int prec, const GemmParams& p);     // rows per workgroup tile srad_launch_gemm will pick for p
// DRN-L's 80 -> 80 channel 3x3 convolutions: weight-resident persistent kernel (kernels_conv80.hip); srad_launch_gemm routes to it
bool srad_conv80_supported(int prec, const GemmParams& p);
int srad_launch_conv80(const GemmParams& p, hipStream_t stream);
// 3x3 convolutions with <= 8 input or <= 4 output channels at large pixel counts (DRN's head / tails and their data gradients):
// direct fp32 FMAs, one pixel per thread (kernels_thin.hip); routed from srad_launch_gemm in bf16 mode
bool srad_conv_thin_supported(int prec, const GemmParams& p);
int srad_launch_conv_thin(const GemmParams& p, hipStream_t stream);
// DRN's tail convolutions (40 / 80 -> <= 4 channels): MFMA with both operands from registers, no LDS (kernels_thin.hip)
bool srad_conv_tail_supported(int prec, const GemmParams& p);
int srad_launch_conv_tail(const GemmParams& p, hipStream_t stream);

// Packed weight geometry shared by the packer and the GEMM
static inline int srad_cp(int cin) { return srad_round_up(cin, 32); }
static inline int srad_np(int n) { return srad_round_up(n, 128); }   // rows padded so 64- and 128-row stages never leave the tensor
// (split-bf16: a bf16 hi plane followed by a bf16 lo plane of the same geometry = the fp32 pack's size)
static inline size_t srad_packed_bytes(int prec, int n, int cin, int ntaps) {
  return (size_t)srad_np(n) * ntaps * srad_cp(cin) * (prec == SRAD_PREC_BF16 ? 2 : 4);
}
static inline size_t srad_packed_lo_offset(int n, int cin, int ntaps) {    // bytes from the hi plane to the lo plane
  return (size_t)srad_np(n) * ntaps * srad_cp(cin) * 2;
}
// src: PyTorch layout [N][Cin][kh][kw] (kh*kw = ntaps) or [N][Cin] for Linear
int srad_launch_pack_weight(int prec, const float* src, void* dst, int n, int cin, int ntaps,
                            hipStream_t stream);
// same with zero rows up to n_pad and input channels regrouped from groups of grp_real to grp_pad (0: none);
// the packed geometry is that of (n_pad, (cin / grp_real) * grp_pad)
int srad_launch_pack_weight_padded(int prec, const float* src, void* dst, int n, int cin, int ntaps, int n_pad,
                                   int grp_real, int grp_pad, hipStream_t stream);
// Linear weights once more as bf16 MFMA fragments (16 x 32 tiles of 1 KB, tile-major): same byte size as the bf16 pack
int srad_launch_pack_weight_frag(const float* src, void* dst, int n, int cin, hipStream_t stream);
// the lo terms of the same pack, bf16(w - float(bf16(w))): the second operand set of the split-bf16 kernels
int srad_launch_pack_weight_frag_lo(const float* src, void* dst, int n, int cin, hipStream_t stream);
int srad_launch_pack_weight_frag_t(const float* src, void* dst, int n, int cin, hipStream_t stream);   // W^T: rows = cin, k = n
// qkv.weight [3d][d] as per-head fragments: head h owns 3 * HDP virtual rows [q_h | k_h | v_h], each slice padded with zero
// rows from head_dim to HDP = ceil16(head_dim); layout [head][virtual row / 16][k / 32][16 x 32] (1 KB tiles)
static inline int srad_qkv_hdp(int d, int heads) { return srad_round_up(d / heads, 16); }
static inline size_t srad_qkv_frag_bytes(int d, int heads) { return (size_t)heads * 3 * srad_qkv_hdp(d, heads) * srad_cp(d) * 2; }
int srad_launch_pack_qkv_frag(const float* src, void* dst, int d, int heads, hipStream_t stream);
int srad_launch_pack_qkv_frag_lo(const float* src, void* dst, int d, int heads, hipStream_t stream);
// Data-gradient operand: the same tensor packed as the weight of the transposed convolution,
// dst[c][8 - tap][n] (taps mirrored for 3x3, identity for 1x1), geometry (rows cin_pad, columns n_pad).
int srad_launch_pack_weight_transposed(int prec, const float* src, void* dst, int n, int cin, int ntaps, int n_pad,
                                       int cin_pad, hipStream_t stream);

// ------------------------------------------------------------------------------------------
// Window attention (DRCT): qkv [T][3d] -> out [T][d], tokens in raster order per image.
// ------------------------------------------------------------------------------------------
struct AttnParams {
  const float* qkv;   // [T][3][heads][hdp] head-padded q | k | v
  float* out;         // [T][d]
  const float* table; // [(2ws-1)^2][heads]
  int B, H, W, ws, shift, d, heads;
  int hdp;            // padded head_dim (multiple of 4, >= d / heads)
  __bf16* out_h = nullptr;   // write the output as bf16 [T][d] instead (the hand-off to mlp_block, which rounds it to bf16 anyway)
  const __bf16* qkv_h = nullptr;   // q | k | v as bf16 in the same layout, prepared by the QKV GEMM (GemmParams::Yh): q scaled, padding set
};
int srad_launch_window_attn(int prec, const AttnParams& p, hipStream_t stream);
// does the window attention of this geometry take its input as bf16 (the 64 x 64-window path)?  *qscale = what q must carry
bool srad_window_attn_bf16_in(int prec, int ws, int shift, int d, int heads, float* qscale);

// ------------------------------------------------------------------------------------------
// Fused second half of a Swin block (kernels_fused.hip): proj + shortcut -> LayerNorm2 -> fc1 -> GELU
// -> fc2 + residual -> adjust 1x1 conv (+ LeakyReLU | * alpha + R), 32 token rows per workgroup.
// Weights are the fragment-major bf16 packs (srad_launch_pack_weight_frag).
// ------------------------------------------------------------------------------------------
struct MlpBlockParams {
  const __bf16* attn_h; int ld_attn;     // [M][d] attention output (pre-proj) as bf16 - what the MFMA takes; ld in elements, % 4 == 0
  const float* shortcut; int ld_short;   // [M][>=d] block input (residual)
  int M, d, m, no;                       // rows, block dim, MLP hidden, adjust output channels
  const void *w_proj, *w_fc1, *w_fc2, *w_adj;
  const float *b_proj, *b_fc1, *b_fc2, *b_adj, *ln_g, *ln_b;
  int act; float slope, alpha;           // adjust epilogue
  const float* R; int ldr;
  float* Y; int ldy, yoff;
  int dbg;                               // timing experiments only (tools/): 1 skip W loads, 2 skip MFMA, 4 skip GELU
  int fm;                                // token rows per workgroup: 0 = chosen from M, else 16 / 32 / 64
  int no_xcd_map;                        // 1: plain workgroup -> row tile order (A/B of the XCD-affine mapping)
  unsigned long long* stamps;            // diagnostic build (16-row tiles): [workgroup][8 waves][16] s_memtime stamps, else null
  // ---- split-bf16 (SRAD_PREC_BF16X3; inference only, 16 / 32 rows per workgroup): the attention output as fp32 and the lo
  //      fragment packs of the four weights (srad_launch_pack_weight_frag_lo); attn_h is unused ----
  int split;
  const float* attn_f;                   // [M][ld_attn] fp32
  const void *w_proj_lo, *w_fc1_lo, *w_fc2_lo, *w_adj_lo;
  // ---- training (all optional): DropPath factors of the two residual branches and the tensors the backward needs ----
  const float *rs1, *rs2; int rps;       // per-sample factors (row m belongs to sample m / rps), null = 1
  float *save_x1, *save_xn2, *save_hpre, *save_hact, *save_x2;   // [M][d] x + attn branch, [M][d] LayerNorm2, [M][m] fc1 pre-activation, [M][m] GELU, [M][d] block output
  __bf16 *save_hpre_h;                                           // the fc1 pre-activation as bf16 instead (the bf16-output mlp_bwd's GELU' takes it so)
  __bf16 *save_xn2_h, *save_hact_h, *save_x2_h;                  // bf16 forms of the three only the weight gradients read (used instead of the fp32 ones when set)
};
bool srad_mlp_block_supported(int prec, int M, int d, int m, int no);
int srad_launch_mlp_block(const MlpBlockParams& p, hipStream_t stream);

// ------------------------------------------------------------------------------------------
// Fused first half of a Swin block (kernels_fused_attn.hip): LayerNorm1 -> q|k|v of one head ->
// shifted-window attention, one workgroup per (window, head); window size 8 only.
// ------------------------------------------------------------------------------------------
struct QkvAttnParams {
  const float* x; int ldx;                 // block input rows [T][ldx] (columns [0, d) are read)
  const float *ln_g, *ln_b;
  const void* w_qkv; const float* b_qkv;   // per-head fragment pack (srad_launch_pack_qkv_frag), bias [3d]
  const float* table;                      // [225][heads]
  float* out; int ld_out;                  // attention output [T][d] ...
  __bf16* out_h;                           // ... or, when set, as bf16 (same ld, in elements): the hand-off to mlp_block
  int no_xcd_map;                          // 1: plain workgroup -> window order (A/B of the XCD-affine mapping)
  int B, H, W, shift, d, heads;
  // ---- training (optional): what the backward needs ----
  float* save_xn;                          // [T][d] LayerNorm1(x) (written by the head-0 workgroups)
  __bf16* save_xn_h;                       // the same as bf16 (only the weight gradient reads it: half the bytes), or null
  float* save_qkv; int hdp;                // [T][3][heads][hdp] head-padded q | k | v (q unscaled), as the QKV GEMM writes it
  __bf16* save_qkv_h; int hp_h;            // or [T][3][heads][hp_h] bf16 exactly as the attention used them (q scaled, padding 0), hp_h % 8 == 0
  unsigned long long* stamps;              // diagnostic build: [workgroup][8 waves][16] s_memtime stamps, else null
  // split-bf16 (SRAD_PREC_BF16X3; inference only): lo terms of the per-head pack (srad_launch_pack_qkv_frag_lo); the output is fp32 (`out`)
  int split; const void* w_qkv_lo;
};
bool srad_no_xcd_map();     // SRAD_NO_XCD_MAP=1 (read once): plain workgroup order in the Swin-block kernels, for A/B runs
bool srad_qkv_attn_supported(int prec, int ws, int H, int W, int d, int heads);
// LayerNorm1 + qkv Linear -> bf16 head-split q | k | v for the 64 x 64-window attention (kernels_fused_attn.hip ln_qkv_kernel)
struct LnQkvParams {
  const float* x; int ldx;                 // block input rows [M][ldx] (columns [0, d) are read)
  int M, d, heads;                         // M % 64 == 0
  const float *ln_g, *ln_b;
  const void* w_qkv; const float* b_qkv;   // per-head fragment pack (srad_launch_pack_qkv_frag), bias [3d]
  __bf16* qkv_h; int hdp;                  // out: [M][3][heads][hdp] as the attention's MFMA operands (see AttnParams::qkv_h)
  float qscale;                            // factor of the q slices (srad_window_attn_bf16_in)
  // split-bf16 (SRAD_PREC_BF16X3): the lo terms of the fragment pack, and the output as fp32 [M][3][heads][hdp] (plain q | k | v +
  // bias, no factor, pad columns not written) for the split window attention kernel, which splits them again while staging
  const void* w_qkv_lo = nullptr; float* qkv_f = nullptr;
};
bool srad_ln_qkv_supported(int prec, int M, int d, int heads);
int srad_launch_ln_qkv(const LnQkvParams& p, hipStream_t stream);
int srad_launch_qkv_attn(const QkvAttnParams& p, hipStream_t stream);

// ------------------------------------------------------------------------------------------
// Backward kernels (kernels_bwd.hip) - the training path of reference src/trainer.py:152-222.
// ------------------------------------------------------------------------------------------
// Weight gradient of a Linear / conv layer, accumulated (+=) into the PyTorch-layout fp32 tensor:
//   dW[n][c][tap] += alpha * sum_m rs(m) * dY[m][ycol0 + n] * A(m, tap, c),   db[n] += alpha * sum_m rs(m) * dY[m][ycol0 + n]
// with A the forward's row gather (identity for Linear, the 3x3 / strided window for convs).
// stored input channel c of a layer whose input is `grp_pad`-wide groups with `grp_real` real channels each -> the real
// channel index, or `none` for a pad column (grp_pad == 0: the identity)
__host__ __device__ inline int srad_real_channel(int c, int grp_real, int grp_pad, int none) {
  if (grp_pad <= 0) return c;
  const int g = c / grp_pad, r = c - g * grp_pad;
  return r < grp_real ? g * grp_real + r : none;
}

struct WgradParams {
  const float* dY; int ldy, ycol0;
  const float* X; int ldx;
  int M, N, Cin, ntaps;        // N, Cin: padded to multiples of 4 as stored in dY / X
  int n_real, cin_real;        // extents of dW (columns/rows beyond are padding and never written)
  int grp_real, grp_pad;       // grp_pad > 0: X's channels are groups of grp_pad holding grp_real real ones each (DRN x8 level 0)
  int Hi, Wi, Ho, Wo, stride;  // conv geometry (M = B*Ho*Wo); ignored for ntaps == 1 && stride == 1
  const float* row_scale; int rps;
  float alpha;
  float* dW;                   // [n_real][cin_real][ntaps]
  float* db;                   // [n_real] or null
  int x_bf16, dy_bf16;         // operand storage: 1 = the pointer is a __bf16 array (ld in elements); Linear layers on the
                               // mask-free (FULL) bf16 path only.  A bf16 dY is already multiplied by its DropPath factor.
};
// Split-K bookkeeping: srad_launch_wgrad() writes partial tiles into `ws` and queues the layer; srad_wgrad_flush()
// sums the queued layers into their dW / db with one launch (automatic when the batch or `ws` is full).
// A queue lives on the host for the duration of one backward pass; flush before anyone reads the gradients.
#define SRAD_WGRAD_BATCH 12
struct WgradReduceItem {
  float* dW; float* db; const float* part;
  int n_real, cin_real, ntaps, tn, tc, ksplit, tile0;
  int grp_real, grp_pad;         // as WgradParams
  float alpha;
  int wc;                        // C > 0: one C x C partial tile per tap (wgrad80_kernel, wgrad_conv9_kernel), row-major + C bias sums;
                                 // then tn = reduce tiles per tap, tc = float4 per reduce workgroup
};
struct WgradReduceBatch { WgradReduceItem it[SRAD_WGRAD_BATCH]; int count; };
#define SRAD_WGRAD_MULTI 5
struct WgradMulti {                            // Linear layers whose weight-gradient kernels go out as one launch
  WgradParams p[SRAD_WGRAD_MULTI];
  float* part[SRAD_WGRAD_MULTI];
  int ksplit[SRAD_WGRAD_MULTI], tn[SRAD_WGRAD_MULTI], tc[SRAD_WGRAD_MULTI], blk0[SRAD_WGRAD_MULTI], nblk[SRAD_WGRAD_MULTI];
  int count;
};
struct WgradQueue {
  float* ws = nullptr; size_t ws_floats = 0;   // caller-owned device workspace, 16-byte aligned
  size_t used = 0; int tiles = 0;
  WgradReduceBatch batch{};
  WgradMulti multi{};
  double multi_flops = 0, multi_bytes = 0;
};
// queue a Linear layer's weight gradient without launching; srad_wgrad_launch_deferred sends all queued ones as one
// launch (call it before srad_wgrad_flush, which only sums partials that have been written)
int srad_launch_wgrad_deferred(int prec, const WgradParams& p, WgradQueue& q, hipStream_t stream);
int srad_wgrad_launch_deferred(int prec, WgradQueue& q, hipStream_t stream);
int srad_launch_wgrad(int prec, const WgradParams& p, WgradQueue& q, hipStream_t stream);
bool srad_wgrad_conv9_supported(const WgradParams& p);   // the nine-tap kernel takes this layer (bf16 mode; the only one with bf16 conv operands)
int srad_wgrad_flush(WgradQueue& q, hipStream_t stream);

#define SRAD_WGRAD_WS_BYTES ((size_t)256 << 20)   /* what the engines give the queue */

// LayerNorm backward over rows: out (+)= dLN(dxn; x, gamma) + dres ; dgamma/dbeta += column sums (atomicAdd)
struct LnBwdParams {
  const float* dxn; int ld_dxn;
  const float* x; int ldx;
  const float* gamma;
  const float* dres; int ld_dres;    // optional gradient arriving over the residual path
  float* out; int ld_out; int accumulate;
  float* dgamma; float* dbeta;
  int rows, C; float eps;
};


// dgamma / dbeta go through the split-K queue too (per-workgroup column sums, added up at the next flush)
int srad_launch_ln_bwd(const LnBwdParams& p, WgradQueue& q, hipStream_t stream);
// `nrows` partial rows [dgamma SRAD_LNB_CP | dbeta SRAD_LNB_CP] in the queue's workspace + their two column-sum items
constexpr int SRAD_LNB_CP = 320;
int srad_wgrad_queue_ln_partials(WgradQueue& q, float* dgamma, float* dbeta, int C, int nrows, hipStream_t stream, float** part);

// ------------------------------------------------------------------------------------------
// Fused backward of the MLP branch of a Swin block (kernels_fused_bwd.hip, bf16 mode):
//   dh  = (dx2 . W2) * rs2 * gelu'(hpre)            -> written (operand of fc1's weight gradient)
//   dx1 = dx2 + LayerNorm2'(dh . W1)                 dgamma / dbeta partial rows -> the split-K queue
// Weights are fragment-major packs of the TRANSPOSED Linear weights (training arena).
// ------------------------------------------------------------------------------------------
struct MlpBwdParams {
  int M, d, m;
  float* dx2;                           // [M][d] gradient of the block output before the adjust conv (KA > 0: written)
  const float* rs2; int rps;            // DropPath factor of the MLP branch per sample (null = 1)
  const void* w_fc2t;                   // fragments of fc2.weight^T: rows = hidden, k = d
  const float* hpre;                    // [M][m] fc1 pre-activation
  const __bf16* hpre_h = nullptr;       // ... as bf16: what the bf16-output instances (dh_h set) read instead
  float* dh;                            // [M][m]
  __bf16 *dh_h, *dx2s_h;                // bf16 forms for the weight gradients (used instead of dh / the dx2 copy when set):
                                        // dh, and dx2 ALREADY multiplied by the MLP branch's DropPath factor (KA > 0 only)
  const void* w_fc1t;                   // fragments of fc1.weight^T: rows = d, k = hidden
  const float* x1;                      // [M][d] LayerNorm2 input
  const float* ln_g;
  float* dx1;                           // [M][d]
  float *dgamma, *dbeta;
  // ---- optional prologue (KA > 0): the adjust 1x1 conv's data gradient, dx2 = aalpha * (dA (.) lrelu'(y_act)) . Wadj ----
  int KA;                               // output channels of the adjust conv (32 or 180), 0 = dx2 is an input
  const float* dA; int ld_dA;           // [M][>=KA] gradient of the adjust conv's output (before its activation's derivative)
  const float* y_act; int ld_y;         // the conv's forward output for LeakyReLU' (null: no activation)
  float slope, aalpha;
  float* dA_out;                        // [M][KA] dA (.) lrelu'(y): operand of the conv's weight gradient (null: not needed)
  __bf16* dA_out_h = nullptr;           // ... written as bf16 instead when set
  const void* w_adjt;                   // fragments of adjust.weight^T: rows = d, k = KA
  // ---- optional epilogue (w_projt != null): the attention projection's data gradient dO = (dx1 . Wproj) * rs1 ----
  const void* w_projt;                  // fragments of proj.weight^T: rows = d, k = d
  const float* rs1;                     // DropPath factor of the attention branch per sample (null = 1; rows per sample = rps)
  float* dO;                            // [M][d]
  __bf16* dx1s_h = nullptr;             // bf16-output instances: also dx1 times rs1 as bf16 [M][d] (proj's weight-gradient operand)
  int no_xcd_map = 0;                   // 1: plain workgroup -> row tile order
  __bf16* dO_h = nullptr; int dO_heads = 0, dO_hp = 0;   // bf16-output instances: dO as [M][heads][hp] bf16 instead (column c -> head c / (d / heads))
};
bool srad_mlp_bwd_supported(int prec, int M, int d, int m, int KA);
// dX = dY . W followed by the backward of the LayerNorm that produced the Linear's input, one launch (bf16 mode):
//   out (+)= dres + LayerNorm'(dY . W; x, gamma)          dgamma / dbeta partial rows -> the split-K queue
// (the qkv Linear + LayerNorm1 end of a Swin block's backward).  w_t = fragments of W^T: rows = d, k = K.
struct LinLnBwdParams {
  int M, K, d;
  const float* dY; int ld_dy;           // [M][K]
  int dy_bf16;                          // 1: dY is a __bf16 array (ld in elements)
  const void* w_t;
  const float* x; int ldx;              // LayerNorm input rows
  const float* ln_g;
  const float* dres; int ld_dres;       // gradient arriving over the residual path
  float* out; int ld_out; int accumulate;
  float *dgamma, *dbeta;
  int no_xcd_map = 0;   // 1: plain workgroup -> row tile order
};
bool srad_lin_ln_bwd_supported(int prec, int M, int K, int d);
int srad_launch_lin_ln_bwd(const LinLnBwdParams& p, WgradQueue& q, hipStream_t stream);
int srad_launch_mlp_bwd(const MlpBwdParams& p, WgradQueue& q, hipStream_t stream);
bool srad_mlp_bwd_bf16_out(int M);     // are the bf16-output instances of mlp_bwd built for this row count

// Shifted-window attention backward (window size 8): recomputes P from the saved head-padded q|k|v.
struct AttnBwdParams {
  const float* qkv;    // [T][3][heads][hdp] as written by the forward
  const float* dout;   // [T][d] gradient of the attention output (pre-proj)
  float* dqkv;         // [T][3*d] compact q | k | v gradients (plain Linear output order)
  __bf16* dqkv_h;      // bf16 kernel: write them as bf16 instead (what both readers round them to anyway), or null
  const float* table;  // [(2ws-1)^2][heads]
  float* dtable;       // accumulated (atomicAdd)
  int B, H, W, ws, shift, d, heads, hdp;
  // all-bf16 form (window_attn_bwd_h_kernel; head dim <= 32): q (scaled) | k | v as the fused forward saved them,
  // [T][3][heads][hp_h], and dO as [T][heads][hp_h] (padding columns may hold anything); needs dqkv_h
  const __bf16* qkv_h = nullptr; const __bf16* dout_h = nullptr; int hp_h = 0;
  int no_xcd_map = 0;        // 1: windows dealt round-robin over the XCDs (A/B of the strip mapping)
};
int srad_launch_window_attn_bwd(int prec, const AttnBwdParams& p, WgradQueue& q, hipStream_t stream);   // dtable via the queue

// out[m][c] = dy[m*ld_dy + c] * (y[m*ld_y + c] > 0 ? 1 : slope), c < C  (backward through (Leaky)ReLU)
int srad_launch_dact(const float* dy, int ld_dy, const float* y, int ld_y, float* out, int ld_out, int rows, int C,
                     float slope, hipStream_t stream);
// inverse of PixelShuffle(2) on NHWC: src [B][2H][2W][F] -> dst [B*H*W][4F], channel n = c*4 + dy*2 + dx
int srad_launch_unshuffle(const float* src, float* dst, int B, int H, int W, int F, hipStream_t stream);
// dst[m][0..ld_dst) = src[m][0..C) followed by zeros
int srad_launch_copy_cols(const float* src, int ld_src, float* dst, int ld_dst, int rows, int C, hipStream_t stream);
// out = scale * sign(a - b)
int srad_launch_l1_grad(const float* a, const float* b, float* out, size_t n, float scale, hipStream_t stream);
// torch.optim.Adam (L2-style weight decay) on flat fp32 buffers; grad_scale multiplies the gradient first
int srad_launch_adam(float* p, const float* g, float* m, float* v, size_t n, float lr, float beta1, float beta2,
                     float eps, float weight_decay, int step, float grad_scale, hipStream_t stream);
int srad_launch_set4(float* dst, float a, float b, float c, float d, hipStream_t stream);
int srad_launch_adam_dev(float* p, const float* g, float* m, float* v, size_t n, float beta1, float beta2, float eps,
                         float weight_decay, const float* hyper, hipStream_t stream);

// ------------------------------------------------------------------------------------------
// misc kernels
// ------------------------------------------------------------------------------------------
int srad_launch_layernorm(const float* x, int ldx, float* y, int ldy, int rows, int C,
                          const float* g, const float* b, float eps, hipStream_t stream);
// NCHW -> NHWC [pix][Cpad] with (x - mean[c]) * scale (channels >= C zero) ; NHWC -> NCHW with x * scale + mean[c]
#define SRAD_IMG_CPAD 4   /* image tensors are kept with 4 channels so every row is float4-addressable */
int srad_launch_nchw_to_nhwc(const float* x, float* y, int B, int C, int Cpad, int H, int W,
                             const float* mean3, float scale, hipStream_t stream);
int srad_launch_nhwc_to_nchw(const float* x, int ldx, float* y, int B, int C, int H, int W,
                             const float* mean3, float scale, hipStream_t stream);

// ------------------------------------------------------------------------------------------
// Optional per-kernel-class timing with HIP events on the launch stream (bench.py's roofline
// numbers).  Off by default; never active while a stream is being captured.
// ------------------------------------------------------------------------------------------
enum {
  SRAD_K_GEMM_BN64 = 0, SRAD_K_GEMM_BN32, SRAD_K_GEMM_BN16, SRAD_K_ATTN, SRAD_K_LAYERNORM,
  SRAD_K_LAYOUT, SRAD_K_PACK, SRAD_K_SCORE, SRAD_K_MISC, SRAD_K_MLP_BLOCK, SRAD_K_QKV_ATTN,
  SRAD_K_WGRAD, SRAD_K_ATTN_BWD, SRAD_K_LN_BWD, SRAD_K_OPTIM, SRAD_K_WGRAD_REDUCE, SRAD_K_MLP_BWD, SRAD_K_LN_QKV, SRAD_K_CONV80, SRAD_K_COUNT
};
struct SradProfScope {
  hipStream_t s; int active;
  SradProfScope(hipStream_t stream, int cls, double flops, double bytes);
  ~SradProfScope();
};
