// drct_train.hip - DRCT training step on gfx950: forward that keeps what the backward needs, the
// backward itself, and the parameter plumbing around them.  Stands in for loss.backward() /
// optimizer.step() of reference src/trainer.py:186-205 on the model of src/drct.py (DropPath 107-133,
// SwinTransformerBlock.forward 472-512, RDG.forward 388-396, DRCT.forward 886-898).
//
// Memory plan (288 GB of HBM: nothing is recomputed except LayerNorm statistics and the softmax):
//   * parameters live in ONE flat fp32 buffer owned by the caller (PyTorch parameters are views of it),
//     gradients in a second flat buffer with the same offsets; srad_drct_sync_params() refreshes the
//     packed forward weights AND their transposed twins (data-gradient operand) in one launch.
//   * every RDG keeps its dense [T][embed + 4 gc] buffer; every Swin block keeps LN1(x), q|k|v, the
//     attention output, x + attn, LN2(.), fc1 pre-activation, GELU output and the block output.
//     ~2.8 k floats per token per block -> 5.5 GB for DRCT-L at 8 x 32 x 32 tokens.
//   * weight gradients accumulate (atomicAdd) into the flat gradient buffer: zero it per step.
#include <math.h>
#include <stdlib.h>
#include "drct_engine.h"

namespace {

// ------------------------------------------------------------------------------------------ workspace
struct BlockSave { float *xn1, *qkv, *attn, *x1, *xn2, *hpre, *hact, *x2; };

struct TrainWs {
  float *xin, *feat0;
  std::vector<float*> dense;          // n_rdg + 1 buffers [T][D]
  std::vector<BlockSave> blk;
  float *body, *c1, *c2, *outn;
  std::vector<float*> upb;
  // backward temporaries
  float *dimg, *dus, *dc2, *dc1, *dbody, *g0, *g1, *dxn, *dO, *dfeat, *dxin;
  float *dx2[2], *dx1[2], *dA[2], *dh[2], *dqkv[2];   // read by the side stream's weight gradients: two sets, alternating per block
  std::vector<float*> dup;            // gradient of upb[j]
  size_t bytes;
};

TrainWs plan_train_ws(const srad_drct* h, int B, int H, int W, void* base, size_t cap) {
  const srad_drct_config& c = h->cfg;
  const size_t T = (size_t)B * H * W;
  const int E = c.embed_dim, D = E + 4 * c.gc, F = c.num_feat;
  Bump bp(base, cap);
  TrainWs w;
  w.xin = bp.take(T * SRAD_IMG_CPAD);
  w.feat0 = bp.take(T * E);
  for (int i = 0; i <= c.n_rdg; ++i) w.dense.push_back(bp.take(T * D));
  for (const SwinW& sw : h->blocks) {
    BlockSave s;
    const int d = sw.d, q = 3 * sw.heads * hdp_of(d, sw.heads);
    s.xn1 = bp.take(T * d); s.qkv = bp.take(T * q); s.attn = bp.take(T * d); s.x1 = bp.take(T * d);
    s.xn2 = bp.take(T * d); s.hpre = bp.take(T * sw.hidden); s.hact = bp.take(T * sw.hidden); s.x2 = bp.take(T * d);
    w.blk.push_back(s);
  }
  w.body = bp.take(T * E); w.c1 = bp.take(T * E); w.c2 = bp.take(T * F);
  size_t t = T;
  for (size_t j = 0; j < h->up.size(); ++j) { t *= 4; w.upb.push_back(bp.take(t * F)); }
  const size_t Tout = t;
  w.outn = bp.take(Tout * SRAD_IMG_CPAD);
  w.dimg = bp.take(Tout * SRAD_IMG_CPAD);
  t = T;
  for (size_t j = 0; j < h->up.size(); ++j) { t *= 4; w.dup.push_back(bp.take(t * F)); }
  w.dus = bp.take((h->up.empty() ? T : Tout / 4) * 4 * F);
  w.dc2 = bp.take(T * F); w.dc1 = bp.take(T * E); w.dbody = bp.take(T * E);
  w.g0 = bp.take(T * D); w.g1 = bp.take(T * D);
  w.dxn = bp.take(T * h->dmax); w.dO = bp.take(T * h->dmax);
  for (int k = 0; k < 2; ++k) {
    w.dx2[k] = bp.take(T * h->dmax); w.dx1[k] = bp.take(T * h->dmax); w.dA[k] = bp.take(T * E);
    w.dh[k] = bp.take(T * h->hmax); w.dqkv[k] = bp.take(T * 3 * h->dmax);
  }
  w.dfeat = bp.take(T * E); w.dxin = bp.take(T * SRAD_IMG_CPAD);
  w.bytes = bp.used;
  return w;
}

GemmParams fwd_gemm(const srad_drct* h, const ConvW& c, const float* X, int ldx, int M, float* Y, int ldy) {
  GemmParams p{};
  p.X = X; p.ldx = ldx; p.M = M; p.Cin = c.cin; p.Cp = srad_cp(c.cin); p.ntaps = c.ntaps;
  p.stride = 1; p.ln_eps = 1e-5f;
  p.Wp = h->pt.ptr(c.w); p.N = c.n; p.bias = h->pt.fptr(c.b);
  p.alpha = 1.f; p.Y = Y; p.ldy = ldy;
  return p;
}
// dX = dY W : the forward GEMM on the transposed pack (rows = input channels, K = output channels)
GemmParams dgrad_gemm(const srad_drct* h, const ConvW& c, const float* dY, int ldy, int M, float* dX, int ldx) {
  GemmParams p{};
  const int npad = srad_round_up(c.n, 4), cpad = srad_round_up(c.cin, 4);
  p.X = dY; p.ldx = ldy; p.M = M; p.Cin = npad; p.Cp = srad_cp(npad); p.ntaps = c.ntaps;
  p.stride = 1; p.ln_eps = 1e-5f;
  p.Wp = h->ts.tarena + h->ts.t_off[c.w]; p.N = cpad; p.bias = nullptr;
  p.alpha = 1.f; p.Y = dX; p.ldy = ldx;
  return p;
}
WgradParams wgrad_of(const srad_drct* h, const ConvW& c, float* flat_grad, const float* dY, int ldy, int ycol0,
                     const float* X, int ldx, int M) {
  WgradParams p{};
  p.dY = dY; p.ldy = ldy; p.ycol0 = ycol0; p.X = X; p.ldx = ldx; p.M = M;
  p.N = srad_round_up(c.n, 4); p.Cin = srad_round_up(c.cin, 4); p.ntaps = c.ntaps;
  p.n_real = c.n; p.cin_real = c.cin; p.stride = 1; p.alpha = 1.f;
  p.dW = flat_grad + h->ts.flat_off[c.w];
  p.db = c.b >= 0 ? flat_grad + h->ts.flat_off[c.b] : nullptr;
  return p;
}
void geom(GemmParams& p, int H, int W) { p.Hi = p.Ho = H; p.Wi = p.Wo = W; }
void geom(WgradParams& p, int H, int W) { p.Hi = p.Ho = H; p.Wi = p.Wo = W; }

// Which kernels a Swin block's backward takes, and with them the storage form of what the forward saves for it.  The
// training forward and the backward both ask here (same inputs -> same answer); the forward records what it saved and the
// backward refuses a mismatch (SRAD_NO_FUSE changing between the two).
struct BlockPlan {
  bool xh;         // fused forward (qkv_attn + mlp_block): LN1(x), attention output, LN2(.), GELU(.), block output saved as bf16
  bool fuse_mlp;   // mlp_bwd: fc2 / fc1 data gradients + LayerNorm2 backward in one launch
  bool fuse_proj;  // ... with the projection's data gradient behind them
  bool fuse_adj;   // ... and the adjust conv's in front
  bool fuse_qkv;   // lin_ln_bwd: qkv data gradient + LayerNorm1 backward in one launch
  bool attn_h;     // all-bf16 attention backward: q | k | v saved as bf16, dO as bf16 per head, dqkv as bf16
  bool yh_dh;      // bf16-output mlp_bwd (32-row instances): dh (and dO) as bf16, the fc1 pre-activation SAVED as bf16
  bool yh_dx2;     // ... and the dx2 copy (times its DropPath factor)
  bool yh_qkv;     // dqkv as bf16 for both of its readers
};
BlockPlan plan_block(const srad_drct* h, const SwinW& sw, int H, int W, int T, int KA) {
  const srad_drct_config& c = h->cfg;
  const int prec = h->pt.prec, d = sw.d, hd = d / sw.heads;
  const TrainState& ts = h->ts;
  // (windows other than 8 x 8 - the 32 / 64 / 256 px presets of src/main.py:218-219,286 - take the unfused forward, so they take
  // the unfused backward with it: fp32 saves, the general attention backward kernel)
  const bool fused_bwd = prec == SRAD_PREC_BF16 && c.window_size == 8 && getenv("SRAD_NO_FUSE") == nullptr;
  auto tf = [&](int w) { return !ts.tf_off.empty() && ts.tf_off[w] >= 0; };
  BlockPlan b{};
  b.xh = h->fuse_mlp && srad_qkv_attn_supported(prec, c.window_size, H, W, d, sw.heads) && srad_mlp_block_supported(prec, T, d, sw.hidden, KA);
  b.fuse_mlp = fused_bwd && tf(sw.fc1.w) && tf(sw.fc2.w) && srad_mlp_bwd_supported(prec, T, d, sw.hidden, 0);
  b.fuse_proj = b.fuse_mlp && tf(sw.proj.w);
  b.fuse_adj = b.fuse_mlp && tf(sw.adjust.w) && srad_mlp_bwd_supported(prec, T, d, sw.hidden, KA);
  b.fuse_qkv = fused_bwd && tf(sw.qkv.w) && srad_lin_ln_bwd_supported(prec, T, 3 * d, d);
  b.yh_qkv = b.xh && b.fuse_qkv;
  b.attn_h = b.xh && b.fuse_proj && b.yh_qkv && srad_mlp_bwd_bf16_out(T) && hd <= 128 && c.window_size == 8 &&
             getenv("SRAD_ATTN_BWD_F32IO") == nullptr;
  b.yh_dh = b.xh && b.fuse_mlp && srad_mlp_bwd_bf16_out(T);    // (dO: bf16 per head for the all-bf16 attention backward, else fp32)
  b.yh_dx2 = b.yh_dh && b.fuse_adj;
  return b;
}
int attn_hp(const SwinW& sw) { return srad_round_up(sw.d / sw.heads, 8); }

int train_check(const srad_drct* h, int B, int H, int W) {
  SRAD_REQUIRE(h->ts.ready, "drct training: call srad_drct_train_bind() and srad_drct_sync_params() first");
  SRAD_REQUIRE(B > 0 && H > 0 && W > 0, "drct training: empty input");
  const int ws = h->cfg.window_size;
  SRAD_REQUIRE(ws >= 1 && ws <= 16, "drct training: window sizes 1 .. 16 (got %d; the reference's presets build 2, 4, 8 and 16)", ws);
  SRAD_REQUIRE(H % ws == 0 && W % ws == 0, "drct training: input %dx%d is not a multiple of the window size %d", H, W, ws);
  SRAD_REQUIRE((double)B * H * W * h->cfg.upscale * h->cfg.upscale * h->cfg.num_feat * 4.0 < 3.9e9 &&
               (double)B * H * W * 3 * h->dmax * 4.0 < 3.9e9, "drct training: batch too large for 32-bit offsets");
  return SRAD_OK;
}

}  // namespace

extern "C" {

int srad_drct_train_param_floats(srad_drct_t* h, int64_t* total) {
  SRAD_REQUIRE(h && total, "train_param_floats: null argument");
  return train_param_floats(h->pt, h->ts, total);
}

int srad_drct_train_param_offset(srad_drct_t* h, int idx, int64_t* off_floats) {
  int64_t tot = 0;
  SRAD_REQUIRE(h, "train_param_offset: null argument");
  SRAD_TRY(train_param_floats(h->pt, h->ts, &tot));
  SRAD_REQUIRE(off_floats && idx >= 0 && idx < (int)h->pt.entries.size(), "train_param_offset: index %d out of range", idx);
  *off_floats = h->ts.flat_off[idx];
  return SRAD_OK;
}

int srad_drct_train_arena_bytes(srad_drct_t* h, size_t* bytes) {
  SRAD_REQUIRE(h && bytes, "train_arena_bytes: null argument");
  return train_arena_bytes(h->pt, h->ts, bytes);
}

// Binds the caller-owned training arena (transposed weight packs + the descriptor table of the sync kernel +
// the split-K workspace).  Synchronous; call once after srad_drct_bind_arena.
int srad_drct_train_bind(srad_drct_t* h, void* train_arena, size_t bytes) {
  SRAD_REQUIRE(h, "train_bind: null argument");
  return train_bind(h->pt, h->ts, train_arena, bytes);
}

// Refreshes every packed weight (forward and transposed) and raw parameter from the flat fp32 master buffer:
// one launch, to be called after each optimizer step (and once after loading a checkpoint).
int srad_drct_sync_params(srad_drct_t* h, const float* flat_params, void* stream) {
  SRAD_REQUIRE(h, "sync_params: null argument");
  SRAD_TRY(train_sync_params(h->pt, h->ts, flat_params, reinterpret_cast<hipStream_t>(stream)));
  h->gc.reset();
  return SRAD_OK;
}

int srad_drct_train_workspace_bytes(const srad_drct_t* h, int B, int H, int W, size_t* bytes) {
  SRAD_REQUIRE(h && bytes && B > 0 && H > 0 && W > 0, "train_workspace_bytes: bad argument");
  *bytes = plan_train_ws(h, B, H, W, nullptr, 0).bytes;
  return SRAD_OK;
}

// Gradient buckets in the order the backward completes them: 0 = everything after the RDGs (norm ...
// conv_last), 1 .. n_rdg = layers.(n_rdg-1) ... layers.0, n_rdg + 1 = conv_first + patch_embed.norm.
int srad_drct_num_buckets(const srad_drct_t* h) { return h ? h->cfg.n_rdg + 2 : 0; }

int srad_drct_bucket_range(srad_drct_t* h, int bucket, int64_t* off_floats, int64_t* n_floats) {
  SRAD_REQUIRE(h && off_floats && n_floats, "bucket_range: null argument");
  int64_t tot = 0;
  SRAD_TRY(train_param_floats(h->pt, h->ts, &tot));
  const int R = h->cfg.n_rdg;
  SRAD_REQUIRE(bucket >= 0 && bucket < R + 2, "bucket_range: bucket %d out of range", bucket);
  auto start_of_rdg = [&](int i) { return i < R ? h->ts.flat_off[h->blocks[i * 5].n1g] : h->ts.flat_off[h->norm_g]; };
  int64_t a, b;
  if (bucket == 0) { a = h->ts.flat_off[h->norm_g]; b = tot; }
  else if (bucket <= R) { const int i = R - bucket; a = start_of_rdg(i); b = start_of_rdg(i + 1); }
  else { a = 0; b = start_of_rdg(0); }
  *off_floats = a; *n_floats = b - a;
  return SRAD_OK;
}

// Training-mode DRCT.forward: as srad_drct_forward, with DropPath (src/drct.py:107-133) applied through
// `keep_scale` ([2 * n_blocks][B] floats: 0 or 1/keep_prob per sample, row 2j for the attention branch of
// block j and 2j+1 for its MLP branch; null = no DropPath) and every tensor the backward needs left in
// `workspace`, which the caller must hand unchanged to srad_drct_backward.
int srad_drct_forward_train(srad_drct_t* h, const float* x, int B, int H, int W, float* y, const float* keep_scale,
                            void* workspace, size_t workspace_bytes, void* stream) {
  SRAD_REQUIRE(h && x && y && workspace, "drct_forward_train: null argument");
  SRAD_TRY(train_check(h, B, H, W));
  SRAD_REQUIRE(((uintptr_t)workspace & 255) == 0, "drct_forward_train: workspace must be 256-byte aligned");
  const TrainWs w = plan_train_ws(h, B, H, W, workspace, workspace_bytes);
  SRAD_REQUIRE(w.bytes <= workspace_bytes, "drct_forward_train: workspace %zu bytes, %zu needed", workspace_bytes, w.bytes);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const srad_drct_config& c = h->cfg;
  const int prec = c.precision;
  const int T = B * H * W, HW = H * W;
  const int E = c.embed_dim, D = E + 4 * c.gc, F = c.num_feat;
  const float mean3[3] = {c.in_chans == 3 ? 0.4488f : 0.f, c.in_chans == 3 ? 0.4371f : 0.f, c.in_chans == 3 ? 0.4040f : 0.f};

  SRAD_TRY(srad_launch_nchw_to_nhwc(x, w.xin, B, c.in_chans, SRAD_IMG_CPAD, H, W, mean3, c.img_range, s));
  {
    GemmParams p = fwd_gemm(h, h->conv_first, w.xin, SRAD_IMG_CPAD, T, w.feat0, E);
    p.Cin = SRAD_IMG_CPAD; geom(p, H, W);
    SRAD_TRY(srad_launch_gemm(prec, p, s));
  }
  SRAD_TRY(srad_launch_layernorm(w.feat0, E, w.dense[0], D, T, E, h->pt.fptr(h->pe_g), h->pt.fptr(h->pe_b), 1e-5f, s));
  for (int i = 0; i < c.n_rdg; ++i) {
    float* cur = w.dense[i];
    float* nxt = w.dense[i + 1];
    for (int k = 0; k < 5; ++k) {
      const int bi = i * 5 + k;
      const SwinW& sw = h->blocks[bi];
      const BlockSave& sv = w.blk[bi];
      const int d = sw.d, hdp = hdp_of(d, sw.heads);
      const float* ks1 = keep_scale ? keep_scale + (size_t)(2 * bi) * B : nullptr;
      const float* ks2 = keep_scale ? keep_scale + (size_t)(2 * bi + 1) * B : nullptr;
      const int no = k < 4 ? c.gc : E;
      if (h->saved_h.size() != h->blocks.size()) h->saved_h.assign(h->blocks.size(), 0);
      h->saved_h[bi] = 0;
      if (h->fuse_mlp && srad_qkv_attn_supported(prec, c.window_size, H, W, d, sw.heads) &&
          srad_mlp_block_supported(prec, T, d, sw.hidden, no)) {
        // bf16: the two fused launches of the inference path, which also leave what the backward needs
        // (LN1(x), q|k|v, x + attn, LN2(.), fc1 pre-activation, GELU(.), block output) and apply DropPath
        QkvAttnParams a{};
        a.x = cur; a.ldx = D; a.ln_g = h->pt.fptr(sw.n1g); a.ln_b = h->pt.fptr(sw.n1b);
        a.w_qkv = h->pt.frag_ptr(sw.qkv.w); a.b_qkv = h->pt.fptr(sw.qkv.b); a.table = h->pt.fptr(sw.table);
        a.out_h = reinterpret_cast<__bf16*>(sv.attn); a.ld_out = d;      // bf16 hand-off to mlp_block (and to the proj weight gradient)
        a.B = B; a.H = H; a.W = W; a.shift = sw.shift; a.d = d; a.heads = sw.heads;
        // the tensors only the weight gradients read (LN1(x), LN2(.), GELU(.), the block output) are left as bf16, which is
        // what the MFMA would round them to anyway: half the bytes written here and read (3 - 9 times each) by wgrad
        a.save_xn_h = reinterpret_cast<__bf16*>(sv.xn1); a.hdp = hdp;
        const BlockPlan bp = plan_block(h, sw, H, W, T, no);
        h->saved_h[bi] = (char)((bp.attn_h ? 1 : 0) | (bp.yh_dh ? 2 : 0));
        if (bp.attn_h) { a.save_qkv_h = reinterpret_cast<__bf16*>(sv.qkv); a.hp_h = attn_hp(sw); }   // as the MFMA took them
        else a.save_qkv = sv.qkv;
        a.no_xcd_map = srad_no_xcd_map();
        SRAD_TRY(srad_launch_qkv_attn(a, s));
        MlpBlockParams q{};
        q.attn_h = reinterpret_cast<const __bf16*>(sv.attn); q.ld_attn = d; q.shortcut = cur; q.ld_short = D;
        q.M = T; q.d = d; q.m = sw.hidden; q.no = no;
        q.w_proj = h->pt.frag_ptr(sw.proj.w); q.w_fc1 = h->pt.frag_ptr(sw.fc1.w); q.w_fc2 = h->pt.frag_ptr(sw.fc2.w);
        q.w_adj = h->pt.frag_ptr(sw.adjust.w);
        q.b_proj = h->pt.fptr(sw.proj.b); q.b_fc1 = h->pt.fptr(sw.fc1.b); q.b_fc2 = h->pt.fptr(sw.fc2.b); q.b_adj = h->pt.fptr(sw.adjust.b);
        q.ln_g = h->pt.fptr(sw.n2g); q.ln_b = h->pt.fptr(sw.n2b);
        q.rs1 = ks1; q.rs2 = ks2; q.rps = HW;
        q.save_x1 = sv.x1;
        if (bp.yh_dh) q.save_hpre_h = reinterpret_cast<__bf16*>(sv.hpre);     // GELU' takes it as bf16 in the bf16-output mlp_bwd
        else q.save_hpre = sv.hpre;
        q.save_xn2_h = reinterpret_cast<__bf16*>(sv.xn2); q.save_hact_h = reinterpret_cast<__bf16*>(sv.hact); q.save_x2_h = reinterpret_cast<__bf16*>(sv.x2);
        if (k < 4) { q.act = SRAD_ACT_LRELU; q.slope = 0.2f; q.alpha = 1.f; q.Y = cur; q.ldy = D; q.yoff = d; }
        else { q.act = SRAD_ACT_NONE; q.alpha = 0.2f; q.R = cur; q.ldr = D; q.Y = nxt; q.ldy = D; q.yoff = 0; }
        q.no_xcd_map = srad_no_xcd_map();
        SRAD_TRY(srad_launch_mlp_block(q, s));
        continue;
      }
      SRAD_TRY(srad_launch_layernorm(cur, D, sv.xn1, d, T, d, h->pt.fptr(sw.n1g), h->pt.fptr(sw.n1b), 1e-5f, s));
      {
        GemmParams p = fwd_gemm(h, sw.qkv, sv.xn1, d, T, sv.qkv, 3 * sw.heads * hdp);
        p.hsplit_hd = d / sw.heads; p.hsplit_hdp = hdp;
        SRAD_TRY(srad_launch_gemm(prec, p, s));
      }
      {
        AttnParams a{sv.qkv, sv.attn, h->pt.fptr(sw.table), B, H, W, c.window_size, sw.shift, d, sw.heads, hdp};
        SRAD_TRY(srad_launch_window_attn(prec, a, s));
      }
      {  // x1 = shortcut + drop_path(proj(attn))          (drct.py:300, 509)
        GemmParams p = fwd_gemm(h, sw.proj, sv.attn, d, T, sv.x1, d);
        p.R = cur; p.ldr = D; p.row_scale = ks1; p.rps = HW;
        SRAD_TRY(srad_launch_gemm(prec, p, s));
      }
      SRAD_TRY(srad_launch_layernorm(sv.x1, d, sv.xn2, d, T, d, h->pt.fptr(sw.n2g), h->pt.fptr(sw.n2b), 1e-5f, s));
      {  // fc1 + GELU, pre-activation kept            (drct.py:185-186)
        GemmParams p = fwd_gemm(h, sw.fc1, sv.xn2, d, T, sv.hact, sw.hidden);
        p.act = SRAD_ACT_GELU; p.Ypre = sv.hpre;
        SRAD_TRY(srad_launch_gemm(prec, p, s));
      }
      {  // x2 = x1 + drop_path(fc2(.))                (drct.py:188, 510)
        GemmParams p = fwd_gemm(h, sw.fc2, sv.hact, sw.hidden, T, sv.x2, d);
        p.R = sv.x1; p.ldr = d; p.row_scale = ks2; p.rps = HW;
        SRAD_TRY(srad_launch_gemm(prec, p, s));
      }
      if (k < 4) {                                      // (drct.py:389-392)
        GemmParams p = fwd_gemm(h, sw.adjust, sv.x2, d, T, cur, D);
        p.yoff = d; p.act = SRAD_ACT_LRELU; p.slope = 0.2f;
        SRAD_TRY(srad_launch_gemm(prec, p, s));
      } else {                                          // (drct.py:393, 396)
        GemmParams p = fwd_gemm(h, sw.adjust, sv.x2, d, T, nxt, D);
        p.alpha = 0.2f; p.R = cur; p.ldr = D;
        SRAD_TRY(srad_launch_gemm(prec, p, s));
      }
    }
  }
  SRAD_TRY(srad_launch_layernorm(w.dense[c.n_rdg], D, w.body, E, T, E, h->pt.fptr(h->norm_g), h->pt.fptr(h->norm_b), 1e-5f, s));
  {
    GemmParams p = fwd_gemm(h, h->conv_after_body, w.body, E, T, w.c1, E);
    geom(p, H, W); p.R = w.feat0; p.ldr = E;
    SRAD_TRY(srad_launch_gemm(prec, p, s));
  }
  {
    GemmParams p = fwd_gemm(h, h->conv_before_up, w.c1, E, T, w.c2, F);
    geom(p, H, W); p.act = SRAD_ACT_LRELU; p.slope = 0.01f;
    SRAD_TRY(srad_launch_gemm(prec, p, s));
  }
  const float* src = w.c2;
  int hh = H, ww = W;
  for (size_t j = 0; j < h->up.size(); ++j) {
    GemmParams p = fwd_gemm(h, h->up[j], src, F, B * hh * ww, w.upb[j], F);
    geom(p, hh, ww); p.ps = 2;
    SRAD_TRY(srad_launch_gemm(prec, p, s));
    src = w.upb[j]; hh *= 2; ww *= 2;
  }
  {
    GemmParams p = fwd_gemm(h, h->conv_last, src, F, B * hh * ww, w.outn, SRAD_IMG_CPAD);
    geom(p, hh, ww);
    SRAD_TRY(srad_launch_gemm(prec, p, s));
  }
  return srad_launch_nhwc_to_nchw(w.outn, SRAD_IMG_CPAD, y, B, c.in_chans, hh, ww, mean3, 1.0f / c.img_range, s);
}

// Backward of srad_drct_forward_train.  dy [B,C,H*s,W*s] is dLoss/dy; parameter gradients are ACCUMULATED
// into flat_grad (same offsets as the flat parameter buffer); dx (optional) receives dLoss/dx.
// `on_bucket(user, bucket)` (optional) is called on the host right after the last kernel that writes bucket
// `bucket` (srad_drct_bucket_range) has been enqueued - the hook a data-parallel trainer uses to start that
// bucket's all-reduce on another stream while the rest of the backward is still running.
int srad_drct_backward(srad_drct_t* h, const float* dy, int B, int H, int W, const float* keep_scale, float* dx,
                       float* flat_grad, void* workspace, size_t workspace_bytes, void* stream,
                       srad_bucket_fn on_bucket, void* user) {
  SRAD_REQUIRE(h && dy && flat_grad && workspace, "drct_backward: null argument");
  SRAD_TRY(train_check(h, B, H, W));
  const TrainWs w = plan_train_ws(h, B, H, W, workspace, workspace_bytes);
  SRAD_REQUIRE(w.bytes <= workspace_bytes, "drct_backward: workspace %zu bytes, %zu needed", workspace_bytes, w.bytes);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const srad_drct_config& c = h->cfg;
  const int prec = c.precision;
  const int T = B * H * W, HW = H * W;
  const int E = c.embed_dim, D = E + 4 * c.gc, F = c.num_feat;
  const float zero3[3] = {0.f, 0.f, 0.f};
  const int stages = (int)h->up.size();
  int hh = H << stages, ww = W << stages;
  float* G = flat_grad;
  WgradQueue wq = train_wgrad_queue(h->ts);   // split-K partials of the weight gradients, reduced once per Swin block
  // Two streams: the data-gradient chain (each kernel needs the previous one's output) stays on the caller's stream;
  // the weight gradients, which nothing in the backward waits for, run on a side stream and fill the idle CUs.
  // Set SRAD_BWD_ONE_STREAM=1 to keep everything on the caller's stream.
  static const bool one_stream = getenv("SRAD_BWD_ONE_STREAM") != nullptr;
  // (bf16 mode: which fused backward kernels a block takes is plan_block's decision; SRAD_NO_FUSE=1: separate launches)
  // (default priority: a low-priority side stream gained nothing here, and a process that had created one ran later
  //  hipGraph replays of other models at half speed - measured with bench.py's C3 leg)
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
  auto side_waits_main = [&]() -> int {          // everything enqueued on the caller's stream so far is visible to the side stream
    if (side == s) return SRAD_OK;
    hipEvent_t ev;
    SRAD_TRY(next_event(&ev));
    SRAD_CHECK_HIP(hipEventRecord(ev, s));
    SRAD_CHECK_HIP(hipStreamWaitEvent(side, ev, 0));
    return SRAD_OK;
  };
  auto main_waits_side = [&]() -> int {
    if (side == s) return SRAD_OK;
    hipEvent_t ev;
    SRAD_TRY(next_event(&ev));
    SRAD_CHECK_HIP(hipEventRecord(ev, side));
    SRAD_CHECK_HIP(hipStreamWaitEvent(s, ev, 0));
    return SRAD_OK;
  };
  const size_t wq_half = wq.ws_floats / 2;
  float* const wq_base = wq.ws;
  hipEvent_t side_done[2] = {nullptr, nullptr};    // side stream finished block n - 2 / n - 1 (their temporaries are free)
  int blk_count = 0;

  // dLoss/d(outn) = dy / img_range, NCHW -> NHWC (pad channels zero)           (drct.py:897)
  SRAD_TRY(srad_launch_nchw_to_nhwc(dy, w.dimg, B, c.in_chans, SRAD_IMG_CPAD, hh, ww, zero3, 1.0f / c.img_range, s));
  {  // conv_last                                                                (drct.py:895)
    const float* src = stages ? w.upb[stages - 1] : w.c2;
    float* dsrc = stages ? w.dup[stages - 1] : w.dc2;
    WgradParams g = wgrad_of(h, h->conv_last, G, w.dimg, SRAD_IMG_CPAD, 0, src, F, B * hh * ww);
    geom(g, hh, ww);
    SRAD_TRY(srad_launch_wgrad(prec, g, wq, s));
    GemmParams p = dgrad_gemm(h, h->conv_last, w.dimg, SRAD_IMG_CPAD, B * hh * ww, dsrc, F);
    geom(p, hh, ww);
    SRAD_TRY(srad_launch_gemm(prec, p, s));
  }
  for (int j = stages - 1; j >= 0; --j) {   // Upsample stage j: conv F -> 4F + PixelShuffle(2)   (drct.py:694-713)
    hh /= 2; ww /= 2;
    const float* src = j ? w.upb[j - 1] : w.c2;
    float* dsrc = j ? w.dup[j - 1] : w.dc2;
    SRAD_TRY(srad_launch_unshuffle(w.dup[j], w.dus, B, hh, ww, F, s));
    WgradParams g = wgrad_of(h, h->up[j], G, w.dus, 4 * F, 0, src, F, B * hh * ww);
    geom(g, hh, ww);
    SRAD_TRY(srad_launch_wgrad(prec, g, wq, s));
    GemmParams p = dgrad_gemm(h, h->up[j], w.dus, 4 * F, B * hh * ww, dsrc, F);
    geom(p, hh, ww);
    SRAD_TRY(srad_launch_gemm(prec, p, s));
  }
  {  // conv_before_upsample + LeakyReLU(0.01)                                   (drct.py:844-845)
    SRAD_TRY(srad_launch_dact(w.dc2, F, w.c2, F, w.dc2, F, T, F, 0.01f, s));
    WgradParams g = wgrad_of(h, h->conv_before_up, G, w.dc2, F, 0, w.c1, E, T);
    geom(g, H, W);
    SRAD_TRY(srad_launch_wgrad(prec, g, wq, s));
    GemmParams p = dgrad_gemm(h, h->conv_before_up, w.dc2, F, T, w.dc1, E);
    geom(p, H, W);
    SRAD_TRY(srad_launch_gemm(prec, p, s));
  }
  {  // conv_after_body(...) + x                                                 (drct.py:893)
    WgradParams g = wgrad_of(h, h->conv_after_body, G, w.dc1, E, 0, w.body, E, T);
    geom(g, H, W);
    SRAD_TRY(srad_launch_wgrad(prec, g, wq, s));
    GemmParams p = dgrad_gemm(h, h->conv_after_body, w.dc1, E, T, w.dbody, E);
    geom(p, H, W);
    SRAD_TRY(srad_launch_gemm(prec, p, s));
  }
  float* gn = w.g0;     // gradient of the RDG stack's output lives in gn[:, :E]
  float* gc = w.g1;
  {  // norm                                                                      (drct.py:881)
    LnBwdParams l{};
    l.dxn = w.dbody; l.ld_dxn = E; l.x = w.dense[c.n_rdg]; l.ldx = D; l.gamma = h->pt.fptr(h->norm_g);
    l.out = gn; l.ld_out = D; l.dgamma = G + h->ts.flat_off[h->norm_g]; l.dbeta = G + h->ts.flat_off[h->norm_b];
    l.rows = T; l.C = E; l.eps = 1e-5f;
    SRAD_TRY(srad_launch_ln_bwd(l, wq, s));
  }
  SRAD_TRY(srad_wgrad_flush(wq, s));
  if (on_bucket) on_bucket(user, 0);

  for (int i = c.n_rdg - 1; i >= 0; --i) {
    const float* cur = w.dense[i];
    // gc = [ gn[:, :E] | 0 ]: the residual path of out = 0.2 x5 + x               (drct.py:396)
    SRAD_TRY(srad_launch_copy_cols(gn, D, gc, D, T, E, s));
    for (int k = 4; k >= 0; --k) {
      const int bi = i * 5 + k;
      const SwinW& sw = h->blocks[bi];
      const BlockSave& sv = w.blk[bi];
      const int d = sw.d, hdp = hdp_of(d, sw.heads);
      const float* ks1 = keep_scale ? keep_scale + (size_t)(2 * bi) * B : nullptr;
      const float* ks2 = keep_scale ? keep_scale + (size_t)(2 * bi + 1) * B : nullptr;
      const int KA = k < 4 ? c.gc : E;
      // which kernels this block takes and what form the forward left its saved tensors in (see srad_drct_forward_train)
      const BlockPlan bp = plan_block(h, sw, H, W, T, KA);
      const bool xh = bp.xh, fuse_mlp = bp.fuse_mlp, fuse_proj = bp.fuse_proj, fuse_adj = bp.fuse_adj, attn_h = bp.attn_h;
      const bool yh_dh = bp.yh_dh, yh_dx2 = bp.yh_dx2, yh_qkv = bp.yh_qkv;
      {
        const int saved = bi < (int)h->saved_h.size() ? h->saved_h[bi] : 0;
        if (xh && saved != ((attn_h ? 1 : 0) | (yh_dh ? 2 : 0)))
          return srad_set_error(SRAD_ERR_STATE, "drct_backward: block %d was saved by the forward for other backward kernels than this call "
                                "takes (q|k|v %s, fc1 pre-activation %s): SRAD_NO_FUSE / SRAD_ATTN_BWD_F32IO must not change between the two",
                                bi, (saved & 1) ? "bf16" : "fp32", (saved & 2) ? "bf16" : "fp32");
      }
      const int set = blk_count & 1;                       // temporaries + partial workspace of this block
      if (side != s && side_done[set]) SRAD_CHECK_HIP(hipStreamWaitEvent(s, side_done[set], 0));   // block n - 2 fully consumed
      wq.ws = wq_base + (size_t)set * wq_half; wq.ws_floats = wq_half;
      float *dx2 = w.dx2[set], *dx1 = w.dx1[set], *dh = w.dh[set], *dqkv = w.dqkv[set];
      // with the adjust prologue fused the dx2 buffer only holds the bf16 copy (its first half): dx1 * rs1 as bf16 goes behind it
      const bool yh_dx1 = yh_dx2 && fuse_proj;
      __bf16* const dA5_h = reinterpret_cast<__bf16*>(dh) + (size_t)T * sw.hidden;     // second half of the dh buffer (dh itself is bf16 then)
      __bf16* const dx1s_h = reinterpret_cast<__bf16*>(dx2) + (size_t)T * d;
      // ---- adjust_k: 1x1 conv (+ LeakyReLU 0.2 | * 0.2)                        (drct.py:389-393)
      // ---- MLP branch: x2 = x1 + rs2 * fc2(gelu(fc1(LN2(x1))))                 (drct.py:510, 184-190)
      const float* dA; int ldA; float aalpha = 1.f;
      if (k < 4) {
        if (!fuse_adj) SRAD_TRY(srad_launch_dact(gc + d, D, cur + d, D, w.dA[set], c.gc, T, c.gc, 0.2f, s));
        dA = w.dA[set]; ldA = c.gc;          // fused: written by mlp_bwd
      } else {
        dA = gn; ldA = D; aalpha = 0.2f;
      }
      {
        WgradParams g = wgrad_of(h, sw.adjust, G, dA, ldA, 0, sv.x2, d, T);
        g.alpha = aalpha; g.x_bf16 = xh; g.dy_bf16 = yh_dx2 && k < 4;      // (mlp_bwd writes w.dA as bf16 then)
        if (yh_dx2 && k == 4) {   // adjust5: mlp_bwd leaves a bf16 copy of its incoming gradient (the RDG's input gradient, fp32, ld D)
          g.dY = reinterpret_cast<const float*>(dA5_h); g.ldy = KA; g.dy_bf16 = 1;      // behind the bf16 dh: every layer of the launch is bf16
        }
        SRAD_TRY(srad_launch_wgrad_deferred(prec, g, wq, side));
        if (!fuse_adj) {
          GemmParams p = dgrad_gemm(h, sw.adjust, dA, ldA, T, dx2, d);
          p.alpha = aalpha;
          SRAD_TRY(srad_launch_gemm(prec, p, s));
        }
      }
      // the gradients only the weight gradients (and bf16 MFMA stagings) read go out as bf16 too: dh always when the MLP
      // backward is fused, dx2 (pre-multiplied by its DropPath factor) when the adjust prologue computes it in that launch
      {
        WgradParams g = wgrad_of(h, sw.fc2, G, dx2, d, 0, sv.hact, sw.hidden, T);
        g.row_scale = ks2; g.rps = HW; g.x_bf16 = xh; g.dy_bf16 = yh_dx2;
        if (yh_dx2) g.row_scale = nullptr;
        SRAD_TRY(srad_launch_wgrad_deferred(prec, g, wq, side));
        WgradParams g1 = wgrad_of(h, sw.fc1, G, dh, sw.hidden, 0, sv.xn2, d, T);
        g1.x_bf16 = xh; g1.dy_bf16 = yh_dh;
        SRAD_TRY(srad_launch_wgrad_deferred(prec, g1, wq, side));
      }
      if (fuse_mlp) {   // both data gradients and the LayerNorm2 backward in one launch (kernels_fused_bwd.hip)
        MlpBwdParams mb{};
        mb.M = T; mb.d = d; mb.m = sw.hidden; mb.dx2 = dx2; mb.rs2 = ks2; mb.rps = HW;
        mb.w_fc2t = h->ts.tarena + h->ts.tf_off[sw.fc2.w]; mb.hpre = sv.hpre; mb.dh = dh;
        if (yh_dh) mb.hpre_h = reinterpret_cast<const __bf16*>(sv.hpre);
        if (yh_dh) mb.dh_h = reinterpret_cast<__bf16*>(dh);
        if (yh_dx2) mb.dx2s_h = reinterpret_cast<__bf16*>(dx2);
        mb.w_fc1t = h->ts.tarena + h->ts.tf_off[sw.fc1.w]; mb.x1 = sv.x1; mb.ln_g = h->pt.fptr(sw.n2g); mb.dx1 = dx1;
        mb.dgamma = G + h->ts.flat_off[sw.n2g]; mb.dbeta = G + h->ts.flat_off[sw.n2b];
        if (fuse_adj) {   // ... and the adjust conv's data gradient (with its LeakyReLU') in front of them
          mb.KA = KA; mb.w_adjt = h->ts.tarena + h->ts.tf_off[sw.adjust.w]; mb.aalpha = aalpha; mb.slope = 0.2f;
          if (k < 4) {
            mb.dA = gc + d; mb.ld_dA = D; mb.y_act = cur + d; mb.ld_y = D; mb.dA_out = w.dA[set];
            if (yh_dx2) mb.dA_out_h = reinterpret_cast<__bf16*>(w.dA[set]);       // the adjust weight gradient takes it as bf16
          }
          else { mb.dA = gn; mb.ld_dA = D; if (yh_dx2) mb.dA_out_h = dA5_h; }
        }
        if (fuse_proj) { mb.w_projt = h->ts.tarena + h->ts.tf_off[sw.proj.w]; mb.rs1 = ks1; mb.rps = HW; mb.dO = w.dO; }
        if (fuse_proj && attn_h) { mb.dO_h = reinterpret_cast<__bf16*>(w.dO); mb.dO_heads = sw.heads; mb.dO_hp = attn_hp(sw); }
        if (yh_dx1) mb.dx1s_h = dx1s_h;
        mb.no_xcd_map = srad_no_xcd_map();
        SRAD_TRY(srad_launch_mlp_bwd(mb, wq, s));
      } else {
        {
          GemmParams p = dgrad_gemm(h, sw.fc2, dx2, d, T, dh, sw.hidden);
          p.row_scale = ks2; p.rps = HW; p.R = sv.hpre; p.ldr = sw.hidden; p.rmode = SRAD_RMODE_DGELU;
          SRAD_TRY(srad_launch_gemm(prec, p, s));
        }
        {
          GemmParams p = dgrad_gemm(h, sw.fc1, dh, sw.hidden, T, w.dxn, d);
          SRAD_TRY(srad_launch_gemm(prec, p, s));
        }
        {  // dx1 = dx2 + dLN2(dxn)
          LnBwdParams l{};
          l.dxn = w.dxn; l.ld_dxn = d; l.x = sv.x1; l.ldx = d; l.gamma = h->pt.fptr(sw.n2g);
          l.dres = dx2; l.ld_dres = d; l.out = dx1; l.ld_out = d;
          l.dgamma = G + h->ts.flat_off[sw.n2g]; l.dbeta = G + h->ts.flat_off[sw.n2b];
          l.rows = T; l.C = d; l.eps = 1e-5f;
          SRAD_TRY(srad_launch_ln_bwd(l, wq, s));
        }
      }
      // dqkv as bf16 when both of its readers take it that way (the fused qkv + LayerNorm1 backward and the weight gradient)
      // ---- attention branch: x1 = x + rs1 * proj(attn(LN1(x)))                  (drct.py:477-509)
      {
        WgradParams g = wgrad_of(h, sw.proj, G, dx1, d, 0, sv.attn, d, T);
        g.row_scale = ks1; g.rps = HW; g.x_bf16 = xh;          // the fused forward left the attention output as bf16
        if (yh_dx1) { g.dY = reinterpret_cast<const float*>(dx1s_h); g.dy_bf16 = 1; g.row_scale = nullptr; }   // dx1 * rs1 as bf16, from mlp_bwd
        SRAD_TRY(srad_launch_wgrad_deferred(prec, g, wq, side));
        if (!fuse_proj) {
          GemmParams p = dgrad_gemm(h, sw.proj, dx1, d, T, w.dO, d);
          p.row_scale = ks1; p.rps = HW;
          SRAD_TRY(srad_launch_gemm(prec, p, s));
        }
      }
      {
        AttnBwdParams a{sv.qkv, w.dO, dqkv, nullptr, h->pt.fptr(sw.table), G + h->ts.flat_off[sw.table], B, H, W, c.window_size,
                        sw.shift, d, sw.heads, hdp};
        if (yh_qkv) a.dqkv_h = reinterpret_cast<__bf16*>(dqkv);
        if (attn_h) { a.qkv_h = reinterpret_cast<const __bf16*>(sv.qkv); a.dout_h = reinterpret_cast<const __bf16*>(w.dO); a.hp_h = attn_hp(sw); }
        a.no_xcd_map = srad_no_xcd_map();
        SRAD_TRY(srad_launch_window_attn_bwd(prec, a, wq, s));
      }
      {
        WgradParams g = wgrad_of(h, sw.qkv, G, dqkv, 3 * d, 0, sv.xn1, d, T);
        g.x_bf16 = xh; g.dy_bf16 = yh_qkv;
        SRAD_TRY(srad_launch_wgrad_deferred(prec, g, wq, side));
      }
      if (bp.fuse_qkv) {
        // gc[:, :d] += dx1 + dLN1(dqkv . Wqkv): data gradient + LayerNorm1 backward in one launch (kernels_fused_bwd.hip)
        LinLnBwdParams lb{};
        lb.M = T; lb.K = 3 * d; lb.d = d; lb.dY = dqkv; lb.ld_dy = 3 * d; lb.dy_bf16 = yh_qkv; lb.w_t = h->ts.tarena + h->ts.tf_off[sw.qkv.w];
        lb.x = cur; lb.ldx = D; lb.ln_g = h->pt.fptr(sw.n1g); lb.dres = dx1; lb.ld_dres = d;
        lb.out = gc; lb.ld_out = D; lb.accumulate = 1;
        lb.dgamma = G + h->ts.flat_off[sw.n1g]; lb.dbeta = G + h->ts.flat_off[sw.n1b];
        lb.no_xcd_map = srad_no_xcd_map();
        SRAD_TRY(srad_launch_lin_ln_bwd(lb, wq, s));
      } else {
        {
          GemmParams p = dgrad_gemm(h, sw.qkv, dqkv, 3 * d, T, w.dxn, d);
          SRAD_TRY(srad_launch_gemm(prec, p, s));
        }
        {  // gc[:, :d] += dx1 + dLN1(dxn)
          LnBwdParams l{};
          l.dxn = w.dxn; l.ld_dxn = d; l.x = cur; l.ldx = D; l.gamma = h->pt.fptr(sw.n1g);
          l.dres = dx1; l.ld_dres = d; l.out = gc; l.ld_out = D; l.accumulate = 1;
          l.dgamma = G + h->ts.flat_off[sw.n1g]; l.dbeta = G + h->ts.flat_off[sw.n1b];
          l.rows = T; l.C = d; l.eps = 1e-5f;
          SRAD_TRY(srad_launch_ln_bwd(l, wq, s));
        }
      }
      // this block's five weight gradients as ONE launch on the side stream (all their operands exist now), then
      // the reduce of their partials and of the LayerNorm / bias-table column sums the caller's stream has written
      SRAD_TRY(side_waits_main());
      SRAD_TRY(srad_wgrad_launch_deferred(prec, wq, side));
      SRAD_TRY(srad_wgrad_flush(wq, side));
      if (side != s) {
        SRAD_TRY(next_event(&side_done[set]));
        SRAD_CHECK_HIP(hipEventRecord(side_done[set], side));
      }
      ++blk_count;
    }
    SRAD_TRY(main_waits_side());                         // the RDG's gradients are final on the caller's stream
    float* t = gn; gn = gc; gc = t;
    if (on_bucket) on_bucket(user, c.n_rdg - i);
  }
  wq.ws = wq_base; wq.ws_floats = 2 * wq_half;           // the remaining layers run on the caller's stream only
  {  // patch_embed.norm, joined by the long skip of conv_after_body(...) + x      (drct.py:873, 893)
    LnBwdParams l{};
    l.dxn = gn; l.ld_dxn = D; l.x = w.feat0; l.ldx = E; l.gamma = h->pt.fptr(h->pe_g);
    l.dres = w.dc1; l.ld_dres = E; l.out = w.dfeat; l.ld_out = E;
    l.dgamma = G + h->ts.flat_off[h->pe_g]; l.dbeta = G + h->ts.flat_off[h->pe_b];
    l.rows = T; l.C = E; l.eps = 1e-5f;
    SRAD_TRY(srad_launch_ln_bwd(l, wq, s));
  }
  {  // conv_first                                                                 (drct.py:892)
    WgradParams g = wgrad_of(h, h->conv_first, G, w.dfeat, E, 0, w.xin, SRAD_IMG_CPAD, T);
    geom(g, H, W);
    SRAD_TRY(srad_launch_wgrad(prec, g, wq, s));
    if (dx) {
      GemmParams p = dgrad_gemm(h, h->conv_first, w.dfeat, E, T, w.dxin, SRAD_IMG_CPAD);
      geom(p, H, W);
      SRAD_TRY(srad_launch_gemm(prec, p, s));
      SRAD_TRY(srad_launch_nhwc_to_nchw(w.dxin, SRAD_IMG_CPAD, dx, B, c.in_chans, H, W, zero3, c.img_range, s));
    }
  }
  SRAD_TRY(srad_wgrad_flush(wq, s));
  if (on_bucket) on_bucket(user, c.n_rdg + 1);
  return SRAD_OK;
}

/* sign(a - b) * scale: the gradient seed of nn.L1Loss(reduction='mean') when scale = 1 / numel (src/loss.py:84) */
int srad_l1_grad(const float* a, const float* b, float* out, int64_t n, float scale, void* stream) {
  SRAD_REQUIRE(a && b && out && n > 0, "l1_grad: bad argument");
  return srad_launch_l1_grad(a, b, out, (size_t)n, scale, reinterpret_cast<hipStream_t>(stream));
}

/* torch.optim.Adam step on flat buffers (src/trainer.py:49-59: lr 1e-4, betas (0.9, 0.999), eps 1e-8, L2 weight decay) */
int srad_adam_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t n, float lr, float beta1,
                   float beta2, float eps, float weight_decay, int step, float grad_scale, void* stream) {
  SRAD_REQUIRE(params && grads && exp_avg && exp_avg_sq && n > 0, "adam_step: bad argument");
  return srad_launch_adam(params, grads, exp_avg, exp_avg_sq, (size_t)n, lr, beta1, beta2, eps, weight_decay, step,
                          grad_scale, reinterpret_cast<hipStream_t>(stream));
}


/* the same step with [lr, 1 - beta1^t, sqrt(1 - beta2^t), grad_scale] in device memory (hipGraph replay of a training step) */
int srad_adam_step_dev(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t n, float beta1,
                       float beta2, float eps, float weight_decay, const float* dev_hyper, void* stream) {
  SRAD_REQUIRE(params && grads && exp_avg && exp_avg_sq && dev_hyper && n > 0, "adam_step_dev: bad argument");
  return srad_launch_adam_dev(params, grads, exp_avg, exp_avg_sq, (size_t)n, beta1, beta2, eps, weight_decay, dev_hyper,
                              reinterpret_cast<hipStream_t>(stream));
}

/* dst[0..3] = (a, b, c, d) by value through a kernel: how the host hands Adam's per-step scalars to a replayed graph */
int srad_set4(float* dst, float a, float b, float c, float d, void* stream) {
  SRAD_REQUIRE(dst, "set4: null argument");
  return srad_launch_set4(dst, a, b, c, d, reinterpret_cast<hipStream_t>(stream));
}

}  // extern "C"
