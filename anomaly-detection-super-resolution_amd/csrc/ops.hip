// ops.hip - operator-level C ABI: the single kernels behind the engines, exposed so that each
// row of SURVEY.md §8(a) (A4-A7, A9, B3-B6) can be parity-tested on its own against the oracle.
#include "engine.h"
#include "../../include/srad.h"

extern "C" {

// Linear / 1x1 conv / 3x3 conv with the fused prologue+epilogue of kernels_gemm.hip.
//   x      : [M_in rows][ldx] fp32 NHWC activations
//   w      : PyTorch-layout fp32 weight [N][Cin][taps] (device), packed into `scratch` first
//   conv   : ntaps = 9 -> 3x3 pad 1 with the given stride over an (B, Hi, Wi) image; ntaps = 1 and
//            stride 1 -> rows are used as they are
//   ln_g/b : optional LayerNorm over the Cin channels of each row (eps 1e-5), ntaps = 1 only
//   epilogue: + bias -> act (0 none, 1 GELU-erf, 2 LeakyReLU(slope), 3 ReLU) -> * alpha -> + r
//   ps     : 0 plain rows [M][ldy] at column yoff; 2 = PixelShuffle(2) scatter
int srad_op_gemm(int precision, const float* x, int ldx, int B, int Hi, int Wi, int Cin, const float* w, int N,
                 int ntaps, int stride, const float* bias, const float* ln_g, const float* ln_b, int act, float slope,
                 float alpha, const float* r, int ldr, float* y, int ldy, int yoff, int ps, void* scratch,
                 size_t scratch_bytes, void* stream) {
  SRAD_REQUIRE(x && w && y && scratch, "op_gemm: null argument");
  SRAD_REQUIRE(ntaps == 1 || ntaps == 9, "op_gemm: ntaps must be 1 or 9");
  SRAD_REQUIRE(stride == 1 || stride == 2, "op_gemm: stride must be 1 or 2");
  const size_t need = srad_packed_bytes(precision, N, Cin, ntaps);
  SRAD_REQUIRE(scratch_bytes >= need, "op_gemm: scratch %zu bytes, %zu needed", scratch_bytes, need);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  SRAD_TRY(srad_launch_pack_weight(precision, w, scratch, N, Cin, ntaps, s));
  GemmParams p{};
  const int pad = ntaps == 9 ? 1 : 0, k = ntaps == 9 ? 3 : 1;
  p.Hi = Hi; p.Wi = Wi;
  p.Ho = (Hi + 2 * pad - k) / stride + 1;
  p.Wo = (Wi + 2 * pad - k) / stride + 1;
  p.stride = stride;
  p.X = x; p.ldx = ldx; p.M = B * p.Ho * p.Wo; p.Cin = Cin; p.Cp = srad_cp(Cin); p.ntaps = ntaps;
  p.ln_g = ln_g; p.ln_b = ln_b; p.ln_eps = 1e-5f;
  p.Wp = scratch; p.N = N; p.bias = bias;
  p.act = act; p.slope = slope; p.alpha = alpha;
  p.R = r; p.ldr = ldr; p.Y = y; p.ldy = ldy; p.yoff = yoff; p.ps = ps;
  p.hsplit_hd = 0; p.hsplit_hdp = 0;
  return srad_launch_gemm(precision, p, s);
}

// The 80 -> 80 channel 3x3 convolution with bf16 operands as DRN's bf16 chains issue it (conv80_kernel<XH, RM>): x_h [B*H*W][80]
// bf16; output bf16 (y_h) or fp32 (y); residual operand bf16 (r_h) or fp32 (r) or none, rmode as GemmParams (0 add, 2 the
// LeakyReLU / ReLU mask of a backward); optional per-tile column sums.  Fails if the shape does not take that kernel.
int srad_op_conv80_h(const void* x_h, const float* w, const float* bias, int act, float slope, const void* r_h, const float* r,
                     int rmode, int B, int H, int W, void* y_h, float* y, float* pool_part, void* scratch, size_t scratch_bytes,
                     void* stream) {
  SRAD_REQUIRE(x_h && w && (y_h || y) && scratch, "op_conv80_h: null argument");
  const size_t need = srad_packed_bytes(SRAD_PREC_BF16, 80, 80, 9);
  SRAD_REQUIRE(scratch_bytes >= need, "op_conv80_h: scratch %zu bytes, %zu needed", scratch_bytes, need);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  SRAD_TRY(srad_launch_pack_weight(SRAD_PREC_BF16, w, scratch, 80, 80, 9, s));
  GemmParams p{};
  p.Hi = p.Ho = H; p.Wi = p.Wo = W; p.stride = 1;
  p.Xh = reinterpret_cast<const __bf16*>(x_h); p.ldx = 80; p.M = B * H * W; p.Cin = 80; p.Cp = srad_cp(80); p.ntaps = 9; p.ln_eps = 1e-5f;
  p.Wp = scratch; p.N = 80; p.bias = bias; p.act = act; p.slope = slope; p.alpha = 1.f;
  p.R = r; p.Rh = reinterpret_cast<const __bf16*>(r_h); p.ldr = 80; p.rmode = rmode;
  p.Y = y; p.Yh = reinterpret_cast<__bf16*>(y_h); p.ldy = 80; p.pool_part = pool_part;
  SRAD_REQUIRE(srad_conv80_supported(SRAD_PREC_BF16, p), "op_conv80_h: this shape does not take the 80-channel conv kernel");
  return srad_launch_gemm(SRAD_PREC_BF16, p, s);
}

// Diagnostic: launch the same GEMM `iters` times back to back on `stream` (weights packed once) and
// return the average device time per launch in microseconds (HIP events; synchronises the stream).
int srad_bench_gemm(int precision, const float* x, int ldx, int B, int Hi, int Wi, int Cin, const float* w, int N,
                    int ntaps, int stride, const float* bias, const float* ln_g, const float* ln_b, int act,
                    const float* r, int ldr, float* y, int ldy, int hsplit_hd, int hsplit_hdp, void* scratch,
                    size_t scratch_bytes, int iters, float* us_out, void* stream) {
  SRAD_REQUIRE(x && w && y && scratch && us_out && iters > 0, "bench_gemm: bad argument");
  const size_t need = srad_packed_bytes(precision, N, Cin, ntaps);
  SRAD_REQUIRE(scratch_bytes >= need, "bench_gemm: scratch %zu bytes, %zu needed", scratch_bytes, need);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  SRAD_TRY(srad_launch_pack_weight(precision, w, scratch, N, Cin, ntaps, s));
  GemmParams p{};
  const int pad = ntaps == 9 ? 1 : 0, k = ntaps == 9 ? 3 : 1;
  p.Hi = Hi; p.Wi = Wi;
  p.Ho = (Hi + 2 * pad - k) / stride + 1;
  p.Wo = (Wi + 2 * pad - k) / stride + 1;
  p.stride = stride;
  p.X = x; p.ldx = ldx; p.M = B * p.Ho * p.Wo; p.Cin = Cin; p.Cp = srad_cp(Cin); p.ntaps = ntaps;
  p.ln_g = ln_g; p.ln_b = ln_b; p.ln_eps = 1e-5f;
  p.Wp = scratch; p.N = N; p.bias = bias;
  p.act = act; p.slope = 0.2f; p.alpha = 1.f;
  p.R = r; p.ldr = ldr; p.Y = y; p.ldy = ldy; p.yoff = 0; p.ps = 0;
  p.hsplit_hd = hsplit_hd; p.hsplit_hdp = hsplit_hdp;
  for (int i = 0; i < 5; ++i) SRAD_TRY(srad_launch_gemm(precision, p, s));
  hipEvent_t a, b;
  SRAD_CHECK_HIP(hipEventCreate(&a));
  SRAD_CHECK_HIP(hipEventCreate(&b));
  SRAD_CHECK_HIP(hipEventRecord(a, s));
  for (int i = 0; i < iters; ++i) SRAD_TRY(srad_launch_gemm(precision, p, s));
  SRAD_CHECK_HIP(hipEventRecord(b, s));
  SRAD_CHECK_HIP(hipEventSynchronize(b));
  float ms = 0.f;
  SRAD_CHECK_HIP(hipEventElapsedTime(&ms, a, b));
  (void)hipEventDestroy(a);
  (void)hipEventDestroy(b);
  *us_out = ms * 1e3f / iters;
  return SRAD_OK;
}

int srad_bench_window_attn(int precision, const float* qkv, float* out, const float* table, int B, int H, int W, int ws,
                           int shift, int d, int heads, int hdp, int iters, float* us_out, void* stream) {
  SRAD_REQUIRE(qkv && out && table && us_out && iters > 0, "bench_window_attn: bad argument");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  AttnParams a{qkv, out, table, B, H, W, ws, shift, d, heads, hdp};
  for (int i = 0; i < 5; ++i) SRAD_TRY(srad_launch_window_attn(precision, a, s));
  hipEvent_t e0, e1;
  SRAD_CHECK_HIP(hipEventCreate(&e0));
  SRAD_CHECK_HIP(hipEventCreate(&e1));
  SRAD_CHECK_HIP(hipEventRecord(e0, s));
  for (int i = 0; i < iters; ++i) SRAD_TRY(srad_launch_window_attn(precision, a, s));
  SRAD_CHECK_HIP(hipEventRecord(e1, s));
  SRAD_CHECK_HIP(hipEventSynchronize(e1));
  float ms = 0.f;
  SRAD_CHECK_HIP(hipEventElapsedTime(&ms, e0, e1));
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  *us_out = ms * 1e3f / iters;
  return SRAD_OK;
}

// Diagnostic: the fused MLP block on synthetic operands, `iters` back-to-back launches, microseconds per launch.
// scratch must hold the four packed bf16 weights; x/y are [M][320] fp32 buffers.
int srad_bench_mlp_block(int M, int d, int m, int no, const void* attn /* bf16 [M][320] */, const float* shortcut, float* y,
                         const float* w_fp32 /* >= 512*512 floats */, void* scratch, size_t scratch_bytes, int dbg,
                         int iters, float* us_out, void* stream) {
  SRAD_REQUIRE(attn && shortcut && y && w_fp32 && scratch && us_out && iters > 0, "bench_mlp_block: bad argument");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const size_t b1 = srad_align_up(srad_packed_bytes(SRAD_PREC_BF16, d, d, 1), 256), b2 = srad_align_up(srad_packed_bytes(SRAD_PREC_BF16, m, d, 1), 256),
               b3 = srad_align_up(srad_packed_bytes(SRAD_PREC_BF16, d, m, 1), 256), b4 = srad_align_up(srad_packed_bytes(SRAD_PREC_BF16, no, d, 1), 256);
  SRAD_REQUIRE(scratch_bytes >= b1 + b2 + b3 + b4, "bench_mlp_block: scratch too small");
  char* sc = reinterpret_cast<char*>(scratch);
  SRAD_TRY(srad_launch_pack_weight_frag(w_fp32, sc, d, d, s));
  SRAD_TRY(srad_launch_pack_weight_frag(w_fp32, sc + b1, m, d, s));
  SRAD_TRY(srad_launch_pack_weight_frag(w_fp32, sc + b1 + b2, d, m, s));
  SRAD_TRY(srad_launch_pack_weight_frag(w_fp32, sc + b1 + b2 + b3, no, d, s));
  MlpBlockParams q{};
  q.attn_h = reinterpret_cast<const __bf16*>(attn); q.ld_attn = 320; q.shortcut = shortcut; q.ld_short = 320; q.M = M; q.d = d; q.m = m; q.no = no;
  q.w_proj = sc; q.w_fc1 = sc + b1; q.w_fc2 = sc + b1 + b2; q.w_adj = sc + b1 + b2 + b3;
  q.b_proj = q.b_fc1 = q.b_fc2 = q.b_adj = q.ln_g = q.ln_b = w_fp32;
  q.act = SRAD_ACT_LRELU; q.slope = 0.2f; q.alpha = 1.f; q.R = nullptr; q.ldr = 0; q.Y = y; q.ldy = 320; q.yoff = 0; q.dbg = dbg & 0xff;
  q.fm = (dbg >> 8) & 0xff;                         // bits 8..15: rows per workgroup (0 = auto)
  if (dbg & 0x20000) {                              // bit 17: the split-bf16 kernel; `attn` is then an fp32 [M][320] array and the
    char* lo = sc + b1 + b2 + b3 + b4;              // lo packs follow the hi packs in scratch
    SRAD_REQUIRE(scratch_bytes >= 2 * (b1 + b2 + b3 + b4), "bench_mlp_block: scratch too small for the lo packs");
    SRAD_TRY(srad_launch_pack_weight_frag_lo(w_fp32, lo, d, d, s));
    SRAD_TRY(srad_launch_pack_weight_frag_lo(w_fp32, lo + b1, m, d, s));
    SRAD_TRY(srad_launch_pack_weight_frag_lo(w_fp32, lo + b1 + b2, d, m, s));
    SRAD_TRY(srad_launch_pack_weight_frag_lo(w_fp32, lo + b1 + b2 + b3, no, d, s));
    q.split = 1; q.attn_f = reinterpret_cast<const float*>(attn);
    q.w_proj_lo = lo; q.w_fc1_lo = lo + b1; q.w_fc2_lo = lo + b1 + b2; q.w_adj_lo = lo + b1 + b2 + b3;
    q.dbg = 0;
  } else
  if (dbg & 0x100ff) {                              // bit 16 or any switch-off bit: the diagnostic build; its stamps go behind the packed weights
    SRAD_REQUIRE(scratch_bytes >= b1 + b2 + b3 + b4 + (size_t)(M / 16) * 8 * 16 * 8, "bench_mlp_block: scratch too small for the stamps");
    q.stamps = reinterpret_cast<unsigned long long*>(sc + b1 + b2 + b3 + b4);
  }
  for (int i = 0; i < 3; ++i) SRAD_TRY(srad_launch_mlp_block(q, s));
  hipEvent_t a, b;
  SRAD_CHECK_HIP(hipEventCreate(&a));
  SRAD_CHECK_HIP(hipEventCreate(&b));
  SRAD_CHECK_HIP(hipEventRecord(a, s));
  for (int i = 0; i < iters; ++i) SRAD_TRY(srad_launch_mlp_block(q, s));
  SRAD_CHECK_HIP(hipEventRecord(b, s));
  SRAD_CHECK_HIP(hipEventSynchronize(b));
  float ms = 0.f;
  SRAD_CHECK_HIP(hipEventElapsedTime(&ms, a, b));
  (void)hipEventDestroy(a);
  (void)hipEventDestroy(b);
  *us_out = ms * 1e3f / iters;
  return SRAD_OK;
}

// ---- the two fused forward kernels of a Swin block (bf16 / split-bf16, window 8), weights in PyTorch layout, packed into
//      scratch: five hi packs, then (split-bf16) the five lo packs ----
static size_t swin_scratch_parts(int d, int heads, int m, int no, size_t* off) {
  size_t o = 0;
  for (int lo = 0; lo < 2; ++lo) {
    off[5 * lo + 0] = o; o += srad_align_up(srad_qkv_frag_bytes(d, heads), 256);
    off[5 * lo + 1] = o; o += srad_align_up(srad_packed_bytes(SRAD_PREC_BF16, d, d, 1), 256);
    off[5 * lo + 2] = o; o += srad_align_up(srad_packed_bytes(SRAD_PREC_BF16, m, d, 1), 256);
    off[5 * lo + 3] = o; o += srad_align_up(srad_packed_bytes(SRAD_PREC_BF16, d, m, 1), 256);
    off[5 * lo + 4] = o; o += srad_align_up(srad_packed_bytes(SRAD_PREC_BF16, no, d, 1), 256);
  }
  return o;
}

size_t srad_op_swin_scratch_bytes(int d, int heads, int m, int no) {
  size_t off[10];
  return swin_scratch_parts(d, heads > 0 ? heads : 1, m > 0 ? m : 4, no > 0 ? no : 4, off);
}

// First half of a Swin block: LayerNorm1 -> qkv Linear -> shifted-window attention (src/drct.py:477-504, 271-299).
//   x [B*H*W][ldx] (columns [0, d)), w_qkv [3d][d], b_qkv [3d], table [225][heads] -> out [B*H*W][d]
int srad_op_qkv_attn(int precision, const float* x, int ldx, int B, int H, int W, int shift, int d, int heads, const float* ln_g,
                     const float* ln_b, const float* w_qkv, const float* b_qkv, const float* table, void* out, int out_bf16,
                     void* scratch, size_t scratch_bytes, void* stream) {
  SRAD_REQUIRE(x && ln_g && ln_b && w_qkv && b_qkv && table && out && scratch, "op_qkv_attn: null argument");
  SRAD_REQUIRE(precision == SRAD_PREC_BF16 || precision == SRAD_PREC_BF16X3, "op_qkv_attn: precision must be bf16 or split-bf16 (fp32 runs the unfused kernels)");
  SRAD_REQUIRE(srad_qkv_attn_supported(precision, 8, H, W, d, heads), "op_qkv_attn: unsupported shape d=%d heads=%d %dx%d", d, heads, H, W);
  const size_t wb = srad_align_up(srad_qkv_frag_bytes(d, heads), 256);
  const bool x3 = precision == SRAD_PREC_BF16X3;
  SRAD_REQUIRE(scratch_bytes >= wb * (x3 ? 2 : 1) && ((uintptr_t)scratch & 255) == 0, "op_qkv_attn: scratch too small or not 256-byte aligned");
  SRAD_REQUIRE(!(x3 && out_bf16), "op_qkv_attn: the split-bf16 kernel writes fp32");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  SRAD_TRY(srad_launch_pack_qkv_frag(w_qkv, scratch, d, heads, s));
  QkvAttnParams a{};
  if (x3) {
    SRAD_TRY(srad_launch_pack_qkv_frag_lo(w_qkv, reinterpret_cast<char*>(scratch) + wb, d, heads, s));
    a.split = 1; a.w_qkv_lo = reinterpret_cast<char*>(scratch) + wb;
  }
  a.x = x; a.ldx = ldx; a.ln_g = ln_g; a.ln_b = ln_b; a.w_qkv = scratch; a.b_qkv = b_qkv; a.table = table;
  if (out_bf16) a.out_h = reinterpret_cast<__bf16*>(out); else a.out = reinterpret_cast<float*>(out);
  a.ld_out = d; a.B = B; a.H = H; a.W = W; a.shift = shift; a.d = d; a.heads = heads;
  return srad_launch_qkv_attn(a, s);
}

// Diagnostic twin of srad_bench_mlp_block for the first half: microseconds per launch, back-to-back launches.
int srad_bench_qkv_attn(const float* x, int ldx, int B, int H, int W, int shift, int d, int heads, const float* w_fp32,
                        void* out, void* scratch, size_t scratch_bytes, int iters, float* us_out, void* stream) {
  SRAD_REQUIRE(x && w_fp32 && out && scratch && us_out && iters > 0, "bench_qkv_attn: bad argument");
  SRAD_REQUIRE(srad_qkv_attn_supported(SRAD_PREC_BF16, 8, H, W, d, heads), "bench_qkv_attn: unsupported shape");
  SRAD_REQUIRE(scratch_bytes >= srad_align_up(srad_qkv_frag_bytes(d, heads), 256), "bench_qkv_attn: scratch too small");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  SRAD_TRY(srad_launch_pack_qkv_frag(w_fp32, scratch, d, heads, s));
  QkvAttnParams a{};
  a.x = x; a.ldx = ldx; a.ln_g = w_fp32; a.ln_b = w_fp32; a.w_qkv = scratch; a.b_qkv = w_fp32; a.table = w_fp32;
  a.out_h = reinterpret_cast<__bf16*>(out); a.ld_out = d;        // the hand-off the engines use (bf16, in the caller's buffer)
  a.B = B; a.H = H; a.W = W; a.shift = shift & 0xff; a.d = d; a.heads = heads;
  if (shift & 0x20000) {                            // bit 17 of `shift`: the split-bf16 kernel (fp32 output [T][d], lo pack behind the hi pack)
    const size_t wb = srad_align_up(srad_qkv_frag_bytes(d, heads), 256);
    SRAD_REQUIRE(scratch_bytes >= 2 * wb, "bench_qkv_attn: scratch too small for the lo pack");
    SRAD_TRY(srad_launch_pack_qkv_frag_lo(w_fp32, reinterpret_cast<char*>(scratch) + wb, d, heads, s));
    a.split = 1; a.w_qkv_lo = reinterpret_cast<char*>(scratch) + wb;
    a.out = reinterpret_cast<float*>(out); a.out_h = nullptr;
  } else
  if (shift & 0x10000) {                            // bit 16 of `shift`: the stamp build; stamps go behind the weight pack in scratch
    const size_t wb = srad_align_up(srad_qkv_frag_bytes(d, heads), 256);
    SRAD_REQUIRE(scratch_bytes >= wb + (size_t)B * (H / 8) * (W / 8) * heads * 8 * 16 * 8, "bench_qkv_attn: scratch too small for the stamps");
    a.stamps = reinterpret_cast<unsigned long long*>(reinterpret_cast<char*>(scratch) + wb);
  }
  for (int i = 0; i < 3; ++i) SRAD_TRY(srad_launch_qkv_attn(a, s));
  hipEvent_t e0, e1;
  SRAD_CHECK_HIP(hipEventCreate(&e0));
  SRAD_CHECK_HIP(hipEventCreate(&e1));
  SRAD_CHECK_HIP(hipEventRecord(e0, s));
  for (int i = 0; i < iters; ++i) SRAD_TRY(srad_launch_qkv_attn(a, s));
  SRAD_CHECK_HIP(hipEventRecord(e1, s));
  SRAD_CHECK_HIP(hipEventSynchronize(e1));
  float ms = 0.f;
  SRAD_CHECK_HIP(hipEventElapsedTime(&ms, e0, e1));
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  *us_out = ms * 1e3f / iters;
  return SRAD_OK;
}

// Second half of a Swin block + the RDG's adjust conv (src/drct.py:300, 509-510, 184-190, 389-396):
//   x1 = shortcut + proj(attn); x2 = x1 + fc2(GELU(fc1(LN2(x1)))); Y[:, yoff:yoff+no] = act(adjust(x2)) * alpha (+ R)
//   fm: token rows per workgroup (16 / 32 / 64; 0 = the engine's choice for M)
int srad_op_mlp_block(int precision, int M, int d, int m, int no, int fm, const void* attn /* bf16 [M][d]; split-bf16: fp32 */, const float* shortcut, int ld_short,
                      const float* w_proj, const float* b_proj, const float* ln_g, const float* ln_b, const float* w_fc1,
                      const float* b_fc1, const float* w_fc2, const float* b_fc2, const float* w_adj, const float* b_adj, int act,
                      float slope, float alpha, const float* r, int ldr, float* y, int ldy, int yoff, void* scratch,
                      size_t scratch_bytes, void* stream) {
  SRAD_REQUIRE(attn && shortcut && w_proj && b_proj && ln_g && ln_b && w_fc1 && b_fc1 && w_fc2 && b_fc2 && w_adj && b_adj && y && scratch,
               "op_mlp_block: null argument");
  SRAD_REQUIRE(precision == SRAD_PREC_BF16 || precision == SRAD_PREC_BF16X3, "op_mlp_block: precision must be bf16 or split-bf16 (fp32 runs the unfused kernels)");
  SRAD_REQUIRE(srad_mlp_block_supported(precision, M, d, m, no), "op_mlp_block: unsupported shape M=%d d=%d m=%d no=%d", M, d, m, no);
  SRAD_REQUIRE(fm == 0 || ((fm == 16 || fm == 32 || fm == 64) && M % fm == 0), "op_mlp_block: fm must be 0, 16, 32 or 64 and divide M");
  size_t off[10];
  const size_t need = swin_scratch_parts(d, 1, m, no, off);
  SRAD_REQUIRE(scratch_bytes >= need && ((uintptr_t)scratch & 255) == 0, "op_mlp_block: scratch %zu bytes, %zu needed (256-byte aligned)", scratch_bytes, need);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  char* sc = reinterpret_cast<char*>(scratch);
  SRAD_TRY(srad_launch_pack_weight_frag(w_proj, sc + off[1], d, d, s));
  SRAD_TRY(srad_launch_pack_weight_frag(w_fc1, sc + off[2], m, d, s));
  SRAD_TRY(srad_launch_pack_weight_frag(w_fc2, sc + off[3], d, m, s));
  SRAD_TRY(srad_launch_pack_weight_frag(w_adj, sc + off[4], no, d, s));
  MlpBlockParams q{};
  q.attn_h = reinterpret_cast<const __bf16*>(attn); q.ld_attn = d; q.shortcut = shortcut; q.ld_short = ld_short; q.M = M; q.d = d; q.m = m; q.no = no;
  q.w_proj = sc + off[1]; q.w_fc1 = sc + off[2]; q.w_fc2 = sc + off[3]; q.w_adj = sc + off[4];
  if (precision == SRAD_PREC_BF16X3) {
    SRAD_TRY(srad_launch_pack_weight_frag_lo(w_proj, sc + off[6], d, d, s));
    SRAD_TRY(srad_launch_pack_weight_frag_lo(w_fc1, sc + off[7], m, d, s));
    SRAD_TRY(srad_launch_pack_weight_frag_lo(w_fc2, sc + off[8], d, m, s));
    SRAD_TRY(srad_launch_pack_weight_frag_lo(w_adj, sc + off[9], no, d, s));
    q.split = 1; q.attn_f = reinterpret_cast<const float*>(attn);
    q.w_proj_lo = sc + off[6]; q.w_fc1_lo = sc + off[7]; q.w_fc2_lo = sc + off[8]; q.w_adj_lo = sc + off[9];
  }
  q.b_proj = b_proj; q.b_fc1 = b_fc1; q.b_fc2 = b_fc2; q.b_adj = b_adj; q.ln_g = ln_g; q.ln_b = ln_b;
  q.act = act; q.slope = slope; q.alpha = alpha; q.R = r; q.ldr = ldr; q.Y = y; q.ldy = ldy; q.yoff = yoff; q.fm = fm;
  return srad_launch_mlp_block(q, s);
}

int srad_op_wgrad(int precision, const float* dy, int ldy, const float* x, int ldx, int B, int Hi, int Wi, int N,
                  int Cin, int ntaps, int stride, const float* row_scale, float alpha, float* dw, float* db,
                  void* workspace, void* stream) {
  SRAD_REQUIRE(dy && x && dw && workspace, "op_wgrad: null argument");
  SRAD_REQUIRE(((uintptr_t)workspace & 255) == 0, "op_wgrad: workspace must be 256-byte aligned");
  SRAD_REQUIRE(stride == 1 || stride == 2, "op_wgrad: stride must be 1 or 2");
  WgradParams p{};
  const int pad = ntaps == 9 ? 1 : 0, k = ntaps == 9 ? 3 : 1;
  p.Hi = Hi; p.Wi = Wi; p.Ho = (Hi + 2 * pad - k) / stride + 1; p.Wo = (Wi + 2 * pad - k) / stride + 1; p.stride = stride;
  p.dY = dy; p.ldy = ldy; p.X = x; p.ldx = ldx; p.M = B * p.Ho * p.Wo;
  p.N = N; p.Cin = Cin; p.n_real = N; p.cin_real = Cin; p.ntaps = ntaps;
  p.row_scale = row_scale; p.rps = p.Ho * p.Wo; p.alpha = alpha; p.dW = dw; p.db = db;
  WgradQueue q;
  q.ws = reinterpret_cast<float*>(workspace);
  q.ws_floats = SRAD_WGRAD_WS_BYTES / sizeof(float);
  SRAD_TRY(srad_launch_wgrad(precision, p, q, reinterpret_cast<hipStream_t>(stream)));
  return srad_wgrad_flush(q, reinterpret_cast<hipStream_t>(stream));
}

// Weight / bias gradient of the 80 -> 80 channel 3x3 convolution from bf16 operands (wgrad_conv9_kernel<5, YH, XH>, DRN's bf16
// chains): dy_h [B*H*W][80] bf16, x either bf16 (x_bf16 = 1) or fp32.  Fails if the shape does not take that kernel.
int srad_op_wgrad_conv9_h(const void* dy_h, const void* x, int x_bf16, int B, int H, int W, float* dw, float* db, void* workspace,
                          void* stream) {
  SRAD_REQUIRE(dy_h && x && dw && workspace, "op_wgrad_conv9_h: null argument");
  SRAD_REQUIRE(((uintptr_t)workspace & 255) == 0, "op_wgrad_conv9_h: workspace must be 256-byte aligned");
  WgradParams p{};
  p.Hi = p.Ho = H; p.Wi = p.Wo = W; p.stride = 1;
  p.dY = reinterpret_cast<const float*>(dy_h); p.ldy = 80; p.dy_bf16 = 1;
  p.X = reinterpret_cast<const float*>(x); p.ldx = 80; p.x_bf16 = x_bf16 ? 1 : 0;
  p.M = B * H * W; p.N = p.Cin = p.n_real = p.cin_real = 80; p.ntaps = 9; p.alpha = 1.f; p.dW = dw; p.db = db;
  SRAD_REQUIRE(srad_wgrad_conv9_supported(p), "op_wgrad_conv9_h: this shape does not take the nine-tap kernel");
  WgradQueue q;
  q.ws = reinterpret_cast<float*>(workspace);
  q.ws_floats = SRAD_WGRAD_WS_BYTES / sizeof(float);
  SRAD_TRY(srad_launch_wgrad(SRAD_PREC_BF16, p, q, reinterpret_cast<hipStream_t>(stream)));
  return srad_wgrad_flush(q, reinterpret_cast<hipStream_t>(stream));
}

// A Linear layer's weight gradient the way the training step issues it: queued, sent with the (one-layer) deferred launch,
// reduced.  x_bf16 / dy_bf16: the operand pointer is a bf16 array (ld in elements).
int srad_op_wgrad_deferred(int precision, const void* dy, int ldy, int dy_bf16, const void* x, int ldx, int x_bf16, int M, int N,
                           int Cin, const float* row_scale, int rps, float alpha, float* dw, float* db, void* workspace,
                           void* stream) {
  SRAD_REQUIRE(dy && x && dw && workspace, "op_wgrad_deferred: null argument");
  WgradParams p{};
  p.dY = reinterpret_cast<const float*>(dy); p.ldy = ldy; p.X = reinterpret_cast<const float*>(x); p.ldx = ldx; p.M = M;
  p.N = N; p.Cin = Cin; p.n_real = N; p.cin_real = Cin; p.ntaps = 1; p.stride = 1;
  p.row_scale = row_scale; p.rps = rps; p.alpha = alpha; p.dW = dw; p.db = db; p.x_bf16 = x_bf16; p.dy_bf16 = dy_bf16;
  WgradQueue q;
  q.ws = reinterpret_cast<float*>(workspace);
  q.ws_floats = SRAD_WGRAD_WS_BYTES / sizeof(float);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  SRAD_TRY(srad_launch_wgrad_deferred(precision, p, q, s));
  SRAD_TRY(srad_wgrad_launch_deferred(precision, q, s));
  return srad_wgrad_flush(q, s);
}

size_t srad_op_wgrad_workspace_bytes(void) { return SRAD_WGRAD_WS_BYTES; }

// tools/wgrad_bench.py: the five weight gradients of one Swin block (qkv, proj, fc1, fc2, adjust) exactly as the training
// step issues them - deferred into one launch, then the reduce - `iters` times.  storage bit 0: X operands are bf16,
// bit 1: dY operands are bf16.  All layers read the same two operand buffers (the timing does not care).
int srad_bench_wgrad_block(int M, int d, int hidden, int KA, int storage, const void* xbuf, const void* ybuf, float* dw,
                           void* workspace, int iters, void* stream) {
  SRAD_REQUIRE(xbuf && ybuf && dw && workspace && M > 0 && iters > 0, "bench_wgrad_block: bad argument");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const int N[5] = {3 * d, d, hidden, d, KA}, C[5] = {d, d, d, hidden, d};
  WgradQueue q;
  q.ws = reinterpret_cast<float*>(workspace);
  q.ws_floats = SRAD_WGRAD_WS_BYTES / sizeof(float);
  for (int it = 0; it < iters; ++it) {
    float* w = dw;
    for (int l = 0; l < 5; ++l) {
      WgradParams p{};
      p.dY = reinterpret_cast<const float*>(ybuf); p.ldy = N[l]; p.X = reinterpret_cast<const float*>(xbuf); p.ldx = C[l];
      p.M = M; p.N = N[l]; p.Cin = C[l]; p.n_real = N[l]; p.cin_real = C[l]; p.ntaps = 1; p.stride = 1; p.alpha = 1.f;
      p.dW = w; p.x_bf16 = storage & 1; p.dy_bf16 = (storage >> 1) & 1;
      w += (size_t)N[l] * C[l];
      SRAD_TRY(srad_launch_wgrad_deferred(SRAD_PREC_BF16, p, q, s));
    }
    SRAD_TRY(srad_wgrad_launch_deferred(SRAD_PREC_BF16, q, s));
    SRAD_TRY(srad_wgrad_flush(q, s));
  }
  return SRAD_OK;
}

int srad_op_dgrad(int precision, const float* dy, int ldy, int B, int H, int W, int N, const float* w, int Cin,
                  int ntaps, const float* r, int ldr, int rmode, float slope, float alpha, const float* row_scale,
                  float* dx, int ldx, void* scratch, size_t scratch_bytes, void* stream) {
  SRAD_REQUIRE(dy && w && dx && scratch, "op_dgrad: null argument");
  SRAD_REQUIRE((N & 3) == 0 && (Cin & 3) == 0, "op_dgrad: N and Cin must be multiples of 4");
  const size_t need = srad_packed_bytes(precision, Cin, N, ntaps);
  SRAD_REQUIRE(scratch_bytes >= need, "op_dgrad: scratch %zu bytes, %zu needed", scratch_bytes, need);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  SRAD_TRY(srad_launch_pack_weight_transposed(precision, w, scratch, N, Cin, ntaps, N, Cin, s));
  GemmParams p{};
  p.Hi = p.Ho = H; p.Wi = p.Wo = W; p.stride = 1;
  p.X = dy; p.ldx = ldy; p.M = B * H * W; p.Cin = N; p.Cp = srad_cp(N); p.ntaps = ntaps; p.ln_eps = 1e-5f;
  p.Wp = scratch; p.N = Cin; p.slope = slope; p.alpha = alpha;
  p.R = r; p.ldr = ldr; p.rmode = rmode; p.row_scale = row_scale; p.rps = H * W;
  p.Y = dx; p.ldy = ldx;
  return srad_launch_gemm(precision, p, s);
}

int srad_op_layernorm_bwd(const float* dxn, const float* x, int ldx, const float* gamma, const float* dres, float* out,
                          int accumulate, float* dgamma, float* dbeta, int rows, int C, void* workspace, void* stream) {
  SRAD_REQUIRE(dxn && x && gamma && out && workspace, "op_layernorm_bwd: null argument");
  LnBwdParams l{};
  l.dxn = dxn; l.ld_dxn = C; l.x = x; l.ldx = ldx; l.gamma = gamma; l.dres = dres; l.ld_dres = C;
  l.out = out; l.ld_out = C; l.accumulate = accumulate; l.dgamma = dgamma; l.dbeta = dbeta; l.rows = rows; l.C = C; l.eps = 1e-5f;
  WgradQueue q;
  q.ws = reinterpret_cast<float*>(workspace);
  q.ws_floats = SRAD_WGRAD_WS_BYTES / sizeof(float);
  SRAD_TRY(srad_launch_ln_bwd(l, q, reinterpret_cast<hipStream_t>(stream)));
  return srad_wgrad_flush(q, reinterpret_cast<hipStream_t>(stream));
}

int srad_op_window_attn_bwd(int precision, const float* qkv, const float* dout, float* dqkv, const float* table, float* dtable,
                            int B, int H, int W, int ws, int shift, int d, int heads, int hdp, void* workspace, void* stream) {
  SRAD_REQUIRE(qkv && dout && dqkv && table && dtable && workspace, "op_window_attn_bwd: null argument");
  AttnBwdParams a{qkv, dout, dqkv, nullptr, table, dtable, B, H, W, ws, shift, d, heads, hdp};
  WgradQueue q;
  q.ws = reinterpret_cast<float*>(workspace);
  q.ws_floats = SRAD_WGRAD_WS_BYTES / sizeof(float);
  SRAD_TRY(srad_launch_window_attn_bwd(precision, a, q, reinterpret_cast<hipStream_t>(stream)));
  return srad_wgrad_flush(q, reinterpret_cast<hipStream_t>(stream));
}

// The all-bf16 form (head dim <= 128): qkv_h [T][3][heads][hp] with q already scaled, dout_h [T][heads][hp] (padding columns
// are ignored), dqkv_h [T][3 d] - all bf16; table / dtable fp32.
int srad_op_window_attn_bwd_h(const void* qkv_h, const void* dout_h, void* dqkv_h, const float* table, float* dtable,
                              int B, int H, int W, int ws, int shift, int d, int heads, int hp, void* workspace, void* stream) {
  SRAD_REQUIRE(qkv_h && dout_h && dqkv_h && table && dtable && workspace, "op_window_attn_bwd_h: null argument");
  AttnBwdParams a{nullptr, nullptr, nullptr, reinterpret_cast<__bf16*>(dqkv_h), table, dtable, B, H, W, ws, shift, d, heads,
                  srad_round_up(d / (heads > 0 ? heads : 1), 4)};
  a.qkv_h = reinterpret_cast<const __bf16*>(qkv_h); a.dout_h = reinterpret_cast<const __bf16*>(dout_h); a.hp_h = hp;
  WgradQueue q;
  q.ws = reinterpret_cast<float*>(workspace);
  q.ws_floats = SRAD_WGRAD_WS_BYTES / sizeof(float);
  SRAD_TRY(srad_launch_window_attn_bwd(SRAD_PREC_BF16, a, q, reinterpret_cast<hipStream_t>(stream)));
  return srad_wgrad_flush(q, reinterpret_cast<hipStream_t>(stream));
}

static size_t tfrag_bytes(int n, int cin) {   // fragment pack of W^T for W [n][cin]
  return srad_align_up(srad_packed_bytes(SRAD_PREC_BF16, srad_round_up(cin, 4), srad_round_up(n, 4), 1), 256);
}

size_t srad_op_mlp_bwd_scratch_bytes(int d, int m, int KA) {
  return tfrag_bytes(m, d) + tfrag_bytes(d, m) + tfrag_bytes(d, d) + (KA > 0 ? tfrag_bytes(KA, d) : 0);
}

int srad_op_mlp_bwd(int M, int d, int m, float* dx2, const float* hpre, const float* x1, const float* gamma,
                    const float* w_fc1, const float* w_fc2, const float* rs2, int rps, float* dh, float* dx1, float* dgamma,
                    float* dbeta, int KA, const float* dA, int ld_dA, const float* y_act, int ld_y, float slope,
                    float aalpha, const float* w_adj, float* dA_out, const float* w_proj, const float* rs1, float* dO,
                    void* scratch, size_t scratch_bytes, void* workspace, void* stream) {
  SRAD_REQUIRE(dx2 && hpre && x1 && gamma && w_fc1 && w_fc2 && dh && dx1 && scratch && workspace, "op_mlp_bwd: null argument");
  SRAD_REQUIRE(scratch_bytes >= srad_op_mlp_bwd_scratch_bytes(d, m, KA) && ((uintptr_t)scratch & 255) == 0,
               "op_mlp_bwd: scratch too small or not 256-byte aligned");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  char* sp = reinterpret_cast<char*>(scratch);
  MlpBwdParams p{};
  p.M = M; p.d = d; p.m = m; p.dx2 = dx2; p.rs2 = rs2; p.rps = rps; p.hpre = hpre; p.dh = dh; p.x1 = x1; p.ln_g = gamma;
  p.dx1 = dx1; p.dgamma = dgamma; p.dbeta = dbeta;
  p.w_fc1t = sp; SRAD_TRY(srad_launch_pack_weight_frag_t(w_fc1, sp, m, d, s)); sp += tfrag_bytes(m, d);
  p.w_fc2t = sp; SRAD_TRY(srad_launch_pack_weight_frag_t(w_fc2, sp, d, m, s)); sp += tfrag_bytes(d, m);
  if (w_proj) {
    SRAD_REQUIRE(dO, "op_mlp_bwd: dO missing");
    p.w_projt = sp; SRAD_TRY(srad_launch_pack_weight_frag_t(w_proj, sp, d, d, s));
    p.rs1 = rs1; p.dO = dO;
  }
  sp += tfrag_bytes(d, d);
  if (KA > 0) {
    SRAD_REQUIRE(dA && w_adj, "op_mlp_bwd: adjust operands missing");
    p.KA = KA; p.dA = dA; p.ld_dA = ld_dA; p.y_act = y_act; p.ld_y = ld_y; p.slope = slope; p.aalpha = aalpha; p.dA_out = dA_out;
    p.w_adjt = sp; SRAD_TRY(srad_launch_pack_weight_frag_t(w_adj, sp, KA, d, s));
  }
  WgradQueue q;
  q.ws = reinterpret_cast<float*>(workspace);
  q.ws_floats = SRAD_WGRAD_WS_BYTES / sizeof(float);
  SRAD_TRY(srad_launch_mlp_bwd(p, q, s));
  return srad_wgrad_flush(q, s);
}

size_t srad_op_lin_ln_bwd_scratch_bytes(int K, int d) { return tfrag_bytes(K, d); }

int srad_op_lin_ln_bwd(int M, int K, int d, const float* dY, const float* w, const float* x, int ldx, const float* gamma,
                       const float* dres, float* out, int ld_out, int accumulate, float* dgamma, float* dbeta, void* scratch,
                       size_t scratch_bytes, void* workspace, void* stream) {
  SRAD_REQUIRE(dY && w && x && gamma && out && scratch && workspace, "op_lin_ln_bwd: null argument");
  SRAD_REQUIRE(scratch_bytes >= tfrag_bytes(K, d) && ((uintptr_t)scratch & 255) == 0, "op_lin_ln_bwd: scratch too small or unaligned");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  SRAD_TRY(srad_launch_pack_weight_frag_t(w, scratch, K, d, s));
  LinLnBwdParams p{};
  p.M = M; p.K = K; p.d = d; p.dY = dY; p.ld_dy = K; p.w_t = scratch; p.x = x; p.ldx = ldx; p.ln_g = gamma;
  p.dres = dres; p.ld_dres = d; p.out = out; p.ld_out = ld_out; p.accumulate = accumulate; p.dgamma = dgamma; p.dbeta = dbeta;
  WgradQueue q;
  q.ws = reinterpret_cast<float*>(workspace);
  q.ws_floats = SRAD_WGRAD_WS_BYTES / sizeof(float);
  SRAD_TRY(srad_launch_lin_ln_bwd(p, q, s));
  return srad_wgrad_flush(q, s);
}

size_t srad_op_gemm_scratch_bytes(int precision, int N, int Cin, int ntaps) {
  return srad_packed_bytes(precision, N, Cin, ntaps);
}

// WindowAttention core (reference src/drct.py:281-299 plus the roll/partition/reverse around it):
// qkv [B*H*W][3][heads][hdp] (head slices padded to hdp floats, hdp % 4 == 0, pad columns never read as data)
// -> out [B*H*W][d], raster token order.
int srad_op_window_attn(int precision, const float* qkv, float* out, const float* table, int B, int H, int W, int ws,
                        int shift, int d, int heads, int hdp, void* stream) {
  SRAD_REQUIRE(qkv && out && table, "op_window_attn: null argument");
  AttnParams a{qkv, out, table, B, H, W, ws, shift, d, heads, hdp};
  return srad_launch_window_attn(precision, a, reinterpret_cast<hipStream_t>(stream));
}

// ---- the two kernels of the 64 x 64-window attention of BASELINE config C5 (bf16), each on its own ----
// the factor q carries into the attention's bf16 operands (head_dim^-0.5 * log2 e), or 0 when this geometry does not take
// the bf16-operand path (srad_window_attn_bf16_in)
float srad_op_window_attn_qscale(int ws, int shift, int d, int heads) {
  float qs = 0.f;
  return srad_window_attn_bf16_in(SRAD_PREC_BF16, ws, shift, d, heads, &qs) ? qs : 0.f;
}
size_t srad_op_ln_qkv_scratch_bytes(int d, int heads) { return srad_align_up(srad_qkv_frag_bytes(d, heads > 0 ? heads : 1), 256); }
// LayerNorm1 + attn.qkv (src/drct.py:477, 278) -> qkv_h [M][3][heads][hdp] bf16: q times qscale, padding columns 0, column
// head_dim of every v slice 1 (ln_qkv_kernel).  M % 64 == 0.
int srad_op_ln_qkv(const float* x, int ldx, int M, int d, int heads, const float* ln_g, const float* ln_b, const float* w_qkv,
                   const float* b_qkv, void* qkv_h, int hdp, float qscale, void* scratch, size_t scratch_bytes, void* stream) {
  SRAD_REQUIRE(x && ln_g && ln_b && w_qkv && b_qkv && qkv_h && scratch, "op_ln_qkv: null argument");
  SRAD_REQUIRE(srad_ln_qkv_supported(SRAD_PREC_BF16, M, d, heads), "op_ln_qkv: unsupported shape M=%d d=%d heads=%d", M, d, heads);
  SRAD_REQUIRE(scratch_bytes >= srad_op_ln_qkv_scratch_bytes(d, heads) && ((uintptr_t)scratch & 255) == 0, "op_ln_qkv: scratch too small or not 256-byte aligned");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  SRAD_TRY(srad_launch_pack_qkv_frag(w_qkv, scratch, d, heads, s));
  LnQkvParams q{};
  q.x = x; q.ldx = ldx; q.M = M; q.d = d; q.heads = heads; q.ln_g = ln_g; q.ln_b = ln_b; q.w_qkv = scratch; q.b_qkv = b_qkv;
  q.qkv_h = reinterpret_cast<__bf16*>(qkv_h); q.hdp = hdp; q.qscale = qscale;
  return srad_launch_ln_qkv(q, s);
}
// the attention on those operands (window_attn_kernel, ROW64 / QH path; src/drct.py:281-299 + roll / partition / reverse):
// qkv_h as above -> out [B*H*W][d] fp32
int srad_op_window_attn_bf16_in(const void* qkv_h, float* out, const float* table, int B, int H, int W, int ws, int shift, int d,
                                int heads, int hdp, void* stream) {
  SRAD_REQUIRE(qkv_h && out && table, "op_window_attn_bf16_in: null argument");
  float qs = 0.f;
  SRAD_REQUIRE(srad_window_attn_bf16_in(SRAD_PREC_BF16, ws, shift, d, heads, &qs), "op_window_attn_bf16_in: this geometry does not take bf16 operands");
  AttnParams a{nullptr, out, table, B, H, W, ws, shift, d, heads, hdp};
  a.qkv_h = reinterpret_cast<const __bf16*>(qkv_h);
  return srad_launch_window_attn(SRAD_PREC_BF16, a, reinterpret_cast<hipStream_t>(stream));
}

int srad_op_layernorm(const float* x, int ldx, float* y, int ldy, int rows, int C, const float* g, const float* b,
                      void* stream) {
  SRAD_REQUIRE(x && y && g && b, "op_layernorm: null argument");
  return srad_launch_layernorm(x, ldx, y, ldy, rows, C, g, b, 1e-5f, reinterpret_cast<hipStream_t>(stream));
}

}  // extern "C"
