// drct.hip - DRCT-L forward engine + its C ABI (include/srad.h).
// Follows reference src/drct.py: DRCT.forward 886-898, forward_features 870-884, RDG.forward
// 388-396, SwinTransformerBlock.forward 472-512, Upsample 694-713.
//
// HBM layout: everything is NHWC == token-major [T = B*H*W][C] fp32, so PatchEmbed /
// PatchUnEmbed (drct.py:650-654,687-690) are no-ops.  Each RDG works in one dense buffer
// [T][embed+4*gc]: swin_k reads channels [0, d_k) and adjust_k writes channels [d_k, d_k+gc),
// which is exactly torch.cat((x, x1, ..), -1) without any copy.  Two dense buffers ping-pong
// between RDGs (adjust5 writes 0.2*x5 + x into the next one).
#include "../../include/srad.h"
#include <math.h>
#include <stdlib.h>
#include <new>

#include "drct_engine.h"

namespace {

GemmParams base_gemm(const srad_drct* h, const ConvW& c, const float* X, int ldx, int M, float* Y, int ldy) {
  GemmParams p{};
  p.X = X; p.ldx = ldx; p.M = M; p.Cin = c.cin; p.Cp = srad_cp(c.cin); p.ntaps = c.ntaps;
  p.Hi = p.Wi = p.Ho = p.Wo = 0; p.stride = 1;
  p.ln_g = p.ln_b = nullptr; p.ln_eps = 1e-5f;
  p.Wp = h->pt.ptr(c.w); p.N = c.n; p.bias = h->pt.fptr(c.b);
  p.act = SRAD_ACT_NONE; p.slope = 0.f; p.alpha = 1.f;
  p.R = nullptr; p.ldr = 0;
  p.Y = Y; p.ldy = ldy; p.yoff = 0; p.ps = 0;
  return p;
}

void conv_geom(GemmParams& p, int H, int W) { p.Hi = p.Ho = H; p.Wi = p.Wo = W; p.stride = 1; }

struct DrctWs {
  float *xin, *feat0, *dense0, *dense1, *qkv, *attn, *x1, *hid, *x2, *body, *c1, *c2, *outn;
  std::vector<float*> upb;
  size_t bytes;
};

DrctWs plan_ws(const srad_drct* h, int B, int H, int W, void* base, size_t cap) {
  const srad_drct_config& c = h->cfg;
  const size_t T = (size_t)B * H * W;
  const int E = c.embed_dim, D = E + 4 * c.gc;
  Bump bp(base, cap);
  DrctWs w;
  w.xin = bp.take(T * SRAD_IMG_CPAD);
  w.feat0 = bp.take(T * E);
  w.dense0 = bp.take(T * D);
  w.dense1 = bp.take(T * D);
  w.qkv = bp.take(T * h->qkvmax);
  w.attn = bp.take(T * h->dmax);
  w.x1 = bp.take(T * h->dmax);
  w.hid = bp.take(T * h->hmax);
  w.x2 = bp.take(T * h->dmax);
  w.body = bp.take(T * E);
  w.c1 = bp.take(T * E);
  w.c2 = bp.take(T * c.num_feat);
  size_t t = T;
  for (size_t j = 0; j < h->up.size(); ++j) {
    t *= 4;
    w.upb.push_back(bp.take(t * c.num_feat));
  }
  w.outn = bp.take(t * c.in_chans);
  w.bytes = bp.used;
  return w;
}

int forward_body(srad_drct* h, const float* x, int B, int H, int W, float* y, const DrctWs& w, hipStream_t s) {
  const srad_drct_config& c = h->cfg;
  const int prec = c.precision;
  const int T = B * H * W;
  const int E = c.embed_dim, D = E + 4 * c.gc;
  const float mean3[3] = {c.in_chans == 3 ? 0.4488f : 0.f, c.in_chans == 3 ? 0.4371f : 0.f, c.in_chans == 3 ? 0.4040f : 0.f};

  // (x - mean) * img_range, NCHW -> NHWC            (drct.py:887-888)
  SRAD_TRY(srad_launch_nchw_to_nhwc(x, w.xin, B, c.in_chans, SRAD_IMG_CPAD, H, W, mean3, c.img_range, s));
  // conv_first                                       (drct.py:892)
  {
    GemmParams p = base_gemm(h, h->conv_first, w.xin, SRAD_IMG_CPAD, T, w.feat0, E);
    p.Cin = SRAD_IMG_CPAD;                 // padded image channels; the packed weight is zero there
    conv_geom(p, H, W);
    SRAD_TRY(srad_launch_gemm(prec, p, s));
  }
  // patch_embed.norm -> residual stream in dense0[:, :E]   (drct.py:873, 650-654)
  SRAD_TRY(srad_launch_layernorm(w.feat0, E, w.dense0, D, T, E, h->pt.fptr(h->pe_g), h->pt.fptr(h->pe_b), 1e-5f, s));

  float* cur = w.dense0;
  float* nxt = w.dense1;
  for (int i = 0; i < c.n_rdg; ++i) {
    for (int k = 0; k < 5; ++k) {
      const SwinW& sw = h->blocks[i * 5 + k];
      const int d = sw.d;
      const int no = k < 4 ? c.gc : E;
      // the attention output is handed to the fused second half as bf16 (what its MFMA rounds it to anyway: half the bytes);
      // split-bf16: as fp32, both fused kernels with their lo weight packs
      const bool x3 = prec == SRAD_PREC_BF16X3;
      const bool mlp_fused = h->fuse_mlp && srad_mlp_block_supported(prec, T, d, sw.hidden, no);
      __bf16* const attn_h = reinterpret_cast<__bf16*>(w.attn);
      if (h->fuse_mlp && srad_qkv_attn_supported(prec, c.window_size, H, W, d, sw.heads)) {
        // norm1 + qkv + shifted-window attention of one (window, head) per workgroup, one launch
        // (drct.py:477-504, 278-299)
        QkvAttnParams a{};
        a.x = cur; a.ldx = D; a.ln_g = h->pt.fptr(sw.n1g); a.ln_b = h->pt.fptr(sw.n1b);
        a.w_qkv = h->pt.frag_ptr(sw.qkv.w); a.b_qkv = h->pt.fptr(sw.qkv.b); a.table = h->pt.fptr(sw.table);
        a.out = w.attn; a.out_h = mlp_fused && !x3 ? attn_h : nullptr; a.ld_out = d;
        a.split = x3; a.w_qkv_lo = h->pt.frag_lo_ptr(sw.qkv.w);
        a.B = B; a.H = H; a.W = W; a.shift = sw.shift; a.d = d; a.heads = sw.heads;
        a.no_xcd_map = srad_no_xcd_map();
        SRAD_TRY(srad_launch_qkv_attn(a, s));
      } else {
      // norm1 + qkv                                   (drct.py:477, 278)
        // (64 x 64 windows, bf16: the GEMM leaves q | k | v as the bf16 operands the attention's MFMAs take)
        float qscale = 1.f;
        const bool qkv_bf16 = srad_window_attn_bf16_in(prec, c.window_size, sw.shift, d, sw.heads, &qscale);
        if (qkv_bf16 && h->pt.frag_ptr(sw.qkv.w) && srad_ln_qkv_supported(prec, T, d, sw.heads) && getenv("SRAD_NO_LN_QKV") == nullptr) {
          // one launch, 64 rows per workgroup against the whole weight (kernels_fused_attn.hip ln_qkv_kernel)
          LnQkvParams q{};
          q.x = cur; q.ldx = D; q.M = T; q.d = d; q.heads = sw.heads; q.ln_g = h->pt.fptr(sw.n1g); q.ln_b = h->pt.fptr(sw.n1b);
          q.w_qkv = h->pt.frag_ptr(sw.qkv.w); q.b_qkv = h->pt.fptr(sw.qkv.b);
          q.qkv_h = reinterpret_cast<__bf16*>(w.qkv); q.hdp = hdp_of(d, sw.heads); q.qscale = qscale;
          SRAD_TRY(srad_launch_ln_qkv(q, s));
        } else if (x3 && c.window_size == 64 && h->pt.frag_ptr(sw.qkv.w) && h->pt.frag_lo_ptr(sw.qkv.w) && srad_ln_qkv_supported(prec, T, d, sw.heads) &&
                   getenv("SRAD_NO_LN_QKV") == nullptr) {
          // split-bf16: the same launch with hi + lo planes and packs, fp32 q | k | v out (the split attention kernel scales and splits them)
          LnQkvParams q{};
          q.x = cur; q.ldx = D; q.M = T; q.d = d; q.heads = sw.heads; q.ln_g = h->pt.fptr(sw.n1g); q.ln_b = h->pt.fptr(sw.n1b);
          q.w_qkv = h->pt.frag_ptr(sw.qkv.w); q.w_qkv_lo = h->pt.frag_lo_ptr(sw.qkv.w); q.b_qkv = h->pt.fptr(sw.qkv.b);
          q.qkv_f = w.qkv; q.hdp = hdp_of(d, sw.heads); q.qscale = 1.f;
          SRAD_TRY(srad_launch_ln_qkv(q, s));
        } else {
          const int hdp = hdp_of(d, sw.heads);
          GemmParams p = base_gemm(h, sw.qkv, cur, D, T, w.qkv, 3 * sw.heads * hdp);
          p.hsplit_hd = d / sw.heads; p.hsplit_hdp = hdp;      // head-padded q|k|v rows for the attention kernel
          p.ln_g = h->pt.fptr(sw.n1g); p.ln_b = h->pt.fptr(sw.n1b);
          if (qkv_bf16) { p.Yh = reinterpret_cast<__bf16*>(w.qkv); p.hsplit_heads = sw.heads; p.hsplit_qscale = qscale; }
          SRAD_TRY(srad_launch_gemm(prec, p, s));
        }
        // shifted-window attention                      (drct.py:481-504, 281-299)
        {
          AttnParams a{w.qkv, w.attn, h->pt.fptr(sw.table), B, H, W, c.window_size, sw.shift, d, sw.heads,
                       hdp_of(d, sw.heads)};
          if (mlp_fused && !x3) a.out_h = attn_h;
          if (qkv_bf16) a.qkv_h = reinterpret_cast<const __bf16*>(w.qkv);
          SRAD_TRY(srad_launch_window_attn(prec, a, s));
        }
      }
      if (mlp_fused) {
        // proj + shortcut -> norm2 -> fc1 -> GELU -> fc2 + residual -> adjust_k, one launch
        // (drct.py:300, 509-510, 184-190, 389-396)
        MlpBlockParams q{};
        q.attn_h = attn_h; q.ld_attn = d; q.shortcut = cur; q.ld_short = D;
        q.split = x3; q.attn_f = w.attn;
        q.w_proj_lo = h->pt.frag_lo_ptr(sw.proj.w); q.w_fc1_lo = h->pt.frag_lo_ptr(sw.fc1.w); q.w_fc2_lo = h->pt.frag_lo_ptr(sw.fc2.w);
        q.w_adj_lo = h->pt.frag_lo_ptr(sw.adjust.w);
        q.M = T; q.d = d; q.m = sw.hidden; q.no = no;
        q.w_proj = h->pt.frag_ptr(sw.proj.w); q.w_fc1 = h->pt.frag_ptr(sw.fc1.w); q.w_fc2 = h->pt.frag_ptr(sw.fc2.w);
        q.w_adj = h->pt.frag_ptr(sw.adjust.w);
        q.b_proj = h->pt.fptr(sw.proj.b); q.b_fc1 = h->pt.fptr(sw.fc1.b); q.b_fc2 = h->pt.fptr(sw.fc2.b); q.b_adj = h->pt.fptr(sw.adjust.b);
        q.ln_g = h->pt.fptr(sw.n2g); q.ln_b = h->pt.fptr(sw.n2b); q.dbg = 0;
        if (k < 4) { q.act = SRAD_ACT_LRELU; q.slope = 0.2f; q.alpha = 1.f; q.R = nullptr; q.ldr = 0; q.Y = cur; q.ldy = D; q.yoff = d; }
        else { q.act = SRAD_ACT_NONE; q.slope = 0.f; q.alpha = 0.2f; q.R = cur; q.ldr = D; q.Y = nxt; q.ldy = D; q.yoff = 0; }
        q.no_xcd_map = srad_no_xcd_map();
        SRAD_TRY(srad_launch_mlp_block(q, s));
        continue;
      }
      // proj + shortcut                               (drct.py:300, 509)
      {
        GemmParams p = base_gemm(h, sw.proj, w.attn, d, T, w.x1, d);
        p.R = cur; p.ldr = D;
        SRAD_TRY(srad_launch_gemm(prec, p, s));
      }
      // norm2 + fc1 + GELU                            (drct.py:510, 185-186)
      {
        GemmParams p = base_gemm(h, sw.fc1, w.x1, d, T, w.hid, sw.hidden);
        p.ln_g = h->pt.fptr(sw.n2g); p.ln_b = h->pt.fptr(sw.n2b);
        p.act = SRAD_ACT_GELU;
        SRAD_TRY(srad_launch_gemm(prec, p, s));
      }
      // fc2 + residual                                (drct.py:188, 510)
      {
        GemmParams p = base_gemm(h, sw.fc2, w.hid, sw.hidden, T, w.x2, d);
        p.R = w.x1; p.ldr = d;
        SRAD_TRY(srad_launch_gemm(prec, p, s));
      }
      // adjust_k 1x1 conv (+ LeakyReLU 0.2) into the dense buffer; adjust5: *0.2 + x into the next
      if (k < 4) {                                     // (drct.py:389-392)
        GemmParams p = base_gemm(h, sw.adjust, w.x2, d, T, cur, D);
        p.yoff = d; p.act = SRAD_ACT_LRELU; p.slope = 0.2f;
        SRAD_TRY(srad_launch_gemm(prec, p, s));
      } else {                                         // (drct.py:393, 396)
        GemmParams p = base_gemm(h, sw.adjust, w.x2, d, T, nxt, D);
        p.alpha = 0.2f; p.R = cur; p.ldr = D;
        SRAD_TRY(srad_launch_gemm(prec, p, s));
      }
    }
    float* t = cur; cur = nxt; nxt = t;
  }
  // norm                                              (drct.py:881)
  SRAD_TRY(srad_launch_layernorm(cur, D, w.body, E, T, E, h->pt.fptr(h->norm_g), h->pt.fptr(h->norm_b), 1e-5f, s));
  // conv_after_body(...) + x                          (drct.py:893)
  {
    GemmParams p = base_gemm(h, h->conv_after_body, w.body, E, T, w.c1, E);
    conv_geom(p, H, W);
    p.R = w.feat0; p.ldr = E;
    SRAD_TRY(srad_launch_gemm(prec, p, s));
  }
  // conv_before_upsample + LeakyReLU(0.01)            (drct.py:844-845, 894)
  {
    GemmParams p = base_gemm(h, h->conv_before_up, w.c1, E, T, w.c2, c.num_feat);
    conv_geom(p, H, W);
    p.act = SRAD_ACT_LRELU; p.slope = 0.01f;
    SRAD_TRY(srad_launch_gemm(prec, p, s));
  }
  // Upsample: conv 64->256 + PixelShuffle(2) per stage (drct.py:694-713)
  const float* src = w.c2;
  int hh = H, ww = W;
  for (size_t j = 0; j < h->up.size(); ++j) {
    GemmParams p = base_gemm(h, h->up[j], src, c.num_feat, B * hh * ww, w.upb[j], c.num_feat);
    conv_geom(p, hh, ww);
    p.ps = 2;
    SRAD_TRY(srad_launch_gemm(prec, p, s));
    src = w.upb[j];
    hh *= 2; ww *= 2;
  }
  // conv_last                                         (drct.py:895)
  {
    GemmParams p = base_gemm(h, h->conv_last, src, c.num_feat, B * hh * ww, w.outn, c.in_chans);
    conv_geom(p, hh, ww);
    SRAD_TRY(srad_launch_gemm(prec, p, s));
  }
  // x / img_range + mean, NHWC -> NCHW                (drct.py:897)
  SRAD_TRY(srad_launch_nhwc_to_nchw(w.outn, c.in_chans, y, B, c.in_chans, hh, ww, mean3, 1.0f / c.img_range, s));
  return SRAD_OK;
}

}  // namespace

extern "C" {

int srad_version(void) { return 100; }

int srad_drct_create(const srad_drct_config* cfg, srad_drct_t** out) {
  SRAD_REQUIRE(cfg && out, "drct_create: null argument");
  SRAD_REQUIRE(cfg->in_chans == 1 || cfg->in_chans == 3, "drct_create: in_chans must be 1 or 3 (got %d)", cfg->in_chans);
  SRAD_REQUIRE(cfg->upscale >= 1 && (cfg->upscale & (cfg->upscale - 1)) == 0, "drct_create: upscale %d is not 2^n", cfg->upscale);
  SRAD_REQUIRE(cfg->window_size >= 1 && cfg->window_size <= 128, "drct_create: window_size %d out of range", cfg->window_size);
  SRAD_REQUIRE(cfg->embed_dim > 0 && cfg->embed_dim % 4 == 0 && cfg->gc % 4 == 0, "drct_create: embed_dim/gc must be multiples of 4");
  SRAD_REQUIRE(cfg->embed_dim % cfg->num_heads == 0, "drct_create: embed_dim %d not divisible by num_heads %d", cfg->embed_dim, cfg->num_heads);
  SRAD_REQUIRE(cfg->precision == SRAD_PREC_F32 || cfg->precision == SRAD_PREC_BF16 || cfg->precision == SRAD_PREC_BF16X3,
               "drct_create: bad precision %d", cfg->precision);
  srad_drct* h = new (std::nothrow) srad_drct();
  if (!h) return srad_set_error(SRAD_ERR_NOMEM, "drct_create: out of host memory");
  h->cfg = *cfg;
  h->pt.prec = cfg->precision;
  h->fuse_mlp = getenv("SRAD_NO_FUSE") == nullptr;
  const int E = cfg->embed_dim, C = cfg->in_chans, ws = cfg->window_size, F = cfg->num_feat;
  h->conv_first = h->pt.add_layer("conv_first", E, C, 9, true);
  h->pe_g = h->pt.add_raw("patch_embed.norm.weight", E);
  h->pe_b = h->pt.add_raw("patch_embed.norm.bias", E);
  h->dmax = 0; h->hmax = 0; h->qkvmax = 0;
  for (int i = 0; i < cfg->n_rdg; ++i) {
    for (int k = 0; k < 5; ++k) {
      SwinW sw;
      sw.d = E + k * cfg->gc;
      sw.heads = k == 0 ? cfg->num_heads : cfg->num_heads - (sw.d % cfg->num_heads);   // drct.py:337-367
      if (sw.heads <= 0 || sw.d % sw.heads != 0) {
        delete h;
        return srad_set_error(SRAD_ERR_ARG, "drct_create: block %d dim %d has no valid head count", k + 1, sw.d);
      }
      sw.hidden = (int)(sw.d * (k < 3 ? cfg->mlp_ratio : 1.0f));
      sw.shift = (k == 1 || k == 3) ? ws / 2 : 0;
      const std::string p = "layers." + std::to_string(i) + ".swin" + std::to_string(k + 1) + ".";
      sw.n1g = h->pt.add_raw(p + "norm1.weight", sw.d);
      sw.n1b = h->pt.add_raw(p + "norm1.bias", sw.d);
      sw.table = h->pt.add_raw(p + "attn.relative_position_bias_table", (int64_t)(2 * ws - 1) * (2 * ws - 1) * sw.heads);
      sw.qkv = h->pt.add_layer(p + "attn.qkv", 3 * sw.d, sw.d, 1, true);
      if (cfg->precision == SRAD_PREC_BF16) h->pt.entries[sw.qkv.w].tfrag = true;   // operand of the fused qkv + LayerNorm1 backward
      const bool frag = cfg->precision == SRAD_PREC_BF16 || cfg->precision == SRAD_PREC_BF16X3;   // operands of the fused block kernels
      if (frag && sw.d % sw.heads == 0) h->pt.add_qkv_frag(sw.qkv, sw.d, sw.heads);   // fused attention kernel
      sw.proj = frag ? h->pt.add_layer_frag(p + "attn.proj", sw.d, sw.d, true) : h->pt.add_layer(p + "attn.proj", sw.d, sw.d, 1, true);
      sw.n2g = h->pt.add_raw(p + "norm2.weight", sw.d);
      sw.n2b = h->pt.add_raw(p + "norm2.bias", sw.d);
      const std::string an = "layers." + std::to_string(i) + ".adjust" + std::to_string(k + 1);
      const int ao = k < 4 ? cfg->gc : E;
      sw.fc1 = frag ? h->pt.add_layer_frag(p + "mlp.fc1", sw.hidden, sw.d, true) : h->pt.add_layer(p + "mlp.fc1", sw.hidden, sw.d, 1, true);
      sw.fc2 = frag ? h->pt.add_layer_frag(p + "mlp.fc2", sw.d, sw.hidden, true) : h->pt.add_layer(p + "mlp.fc2", sw.d, sw.hidden, 1, true);
      sw.adjust = frag ? h->pt.add_layer_frag(an, ao, sw.d, true) : h->pt.add_layer(an, ao, sw.d, 1, true);
      if (sw.d > h->dmax) h->dmax = sw.d;
      if (3 * sw.heads * hdp_of(sw.d, sw.heads) > h->qkvmax) h->qkvmax = 3 * sw.heads * hdp_of(sw.d, sw.heads);
      if (sw.hidden > h->hmax) h->hmax = sw.hidden;
      h->blocks.push_back(sw);
    }
  }
  h->norm_g = h->pt.add_raw("norm.weight", E);
  h->norm_b = h->pt.add_raw("norm.bias", E);
  h->conv_after_body = h->pt.add_layer("conv_after_body", E, E, 9, true);
  h->conv_before_up = h->pt.add_layer("conv_before_upsample.0", F, E, 9, true);
  int stages = 0;
  for (int s = cfg->upscale; s > 1; s >>= 1) ++stages;
  for (int j = 0; j < stages; ++j) h->up.push_back(h->pt.add_layer("upsample." + std::to_string(2 * j), 4 * F, F, 9, true));
  h->conv_last = h->pt.add_layer("conv_last", C, F, 9, true);
  *out = h;
  return SRAD_OK;
}

void srad_drct_destroy(srad_drct_t* h) {
  if (!h) return;
  h->gc.reset();
  for (hipEvent_t ev : h->events) (void)hipEventDestroy(ev);
  if (h->side) (void)hipStreamDestroy(h->side);
  delete h;
}

int srad_drct_arena_bytes(const srad_drct_t* h, size_t* bytes) {
  SRAD_REQUIRE(h && bytes, "drct_arena_bytes: null argument");
  *bytes = h->pt.bytes;
  return SRAD_OK;
}

int srad_drct_bind_arena(srad_drct_t* h, void* arena, size_t bytes) {
  SRAD_REQUIRE(h && arena, "drct_bind_arena: null argument");
  SRAD_REQUIRE(bytes >= h->pt.bytes, "drct_bind_arena: %zu bytes given, %zu needed", bytes, h->pt.bytes);
  SRAD_REQUIRE(((uintptr_t)arena & 255) == 0, "drct_bind_arena: arena must be 256-byte aligned");
  h->pt.arena = reinterpret_cast<char*>(arena);
  h->pt.arena_bytes = bytes;
  h->gc.reset();
  return SRAD_OK;
}

int srad_drct_num_params(const srad_drct_t* h) { return h ? (int)h->pt.entries.size() : 0; }

int srad_drct_param_info(const srad_drct_t* h, int idx, const char** name, int64_t* numel) {
  SRAD_REQUIRE(h && idx >= 0 && idx < (int)h->pt.entries.size(), "drct_param_info: index %d out of range", idx);
  if (name) *name = h->pt.entries[idx].name.c_str();
  if (numel) *numel = h->pt.entries[idx].numel;
  return SRAD_OK;
}

int srad_drct_set_param(srad_drct_t* h, const char* name, const float* dev_src, int64_t numel, void* stream) {
  SRAD_REQUIRE(h && name && dev_src, "drct_set_param: null argument");
  return h->pt.set(name, dev_src, numel, reinterpret_cast<hipStream_t>(stream));
}

static int drct_check_shape(const srad_drct_t* h, int B, int H, int W) {
  SRAD_REQUIRE(B > 0 && H > 0 && W > 0, "drct: empty input %dx%dx%d", B, H, W);
  SRAD_REQUIRE(H % h->cfg.window_size == 0 && W % h->cfg.window_size == 0,
               "drct: input %dx%d is not a multiple of the window size %d", H, W, h->cfg.window_size);
  SRAD_REQUIRE((int64_t)B * H * W * h->cfg.upscale * h->cfg.upscale < (1LL << 31), "drct: problem too large for 32-bit pixel indices");
  return SRAD_OK;
}

int srad_drct_workspace_bytes(const srad_drct_t* h, int B, int H, int W, size_t* bytes) {
  SRAD_REQUIRE(h && bytes, "drct_workspace_bytes: null argument");
  SRAD_TRY(drct_check_shape(h, B, H, W));
  *bytes = plan_ws(h, B, H, W, nullptr, 0).bytes;
  return SRAD_OK;
}

int srad_drct_forward(srad_drct_t* h, const float* x, int B, int H, int W, float* y, void* workspace,
                      size_t workspace_bytes, void* stream) {
  SRAD_REQUIRE(h && x && y && workspace, "drct_forward: null argument");
  if (!h->pt.arena) return srad_set_error(SRAD_ERR_STATE, "drct_forward: no weight arena bound");
  SRAD_TRY(drct_check_shape(h, B, H, W));
  SRAD_REQUIRE(((uintptr_t)workspace & 255) == 0, "drct_forward: workspace must be 256-byte aligned");
  const DrctWs w = plan_ws(h, B, H, W, workspace, workspace_bytes);
  SRAD_REQUIRE(w.bytes <= workspace_bytes, "drct_forward: workspace %zu bytes, %zu needed", workspace_bytes, w.bytes);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  return srad_run_with_graph(h->gc, h->cfg.use_graph != 0, x, y, workspace, B, H, W, s,
                             [&](hipStream_t st) { return forward_body(h, x, B, H, W, y, w, st); });
}

int srad_drct_flops(const srad_drct_t* h, int B, int H, int W, double* flops) {
  SRAD_REQUIRE(h && flops, "drct_flops: null argument");
  const srad_drct_config& c = h->cfg;
  const double T = (double)B * H * W;
  const double N = (double)c.window_size * c.window_size;
  double f = 0;
  auto conv = [&](const ConvW& l, double pix) { f += 2.0 * pix * l.n * l.cin * l.ntaps; };
  conv(h->conv_first, T);
  for (const SwinW& sw : h->blocks) {
    conv(sw.qkv, T); conv(sw.proj, T); conv(sw.fc1, T); conv(sw.fc2, T); conv(sw.adjust, T);
    f += 2.0 * 2.0 * T * N * sw.d;      // q k^T and p v
  }
  conv(h->conv_after_body, T); conv(h->conv_before_up, T);
  double pix = T;
  for (const ConvW& u : h->up) { conv(u, pix); pix *= 4; }
  conv(h->conv_last, pix);
  *flops = f;
  return SRAD_OK;
}

}  // extern "C"
