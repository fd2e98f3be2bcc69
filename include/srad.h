/* srad.h - C ABI of libsrad, the MI355X (gfx950) engine for the SR forward / scorer hot path of
 * Benedict3007/anomaly-detection-super-resolution.
 *
 * The reference has no FFI layer; its seam for this path is Python (`make_model(opt)` /
 * `Model.forward`, reference src/model.py:46-52,95-100; `ssim_numpy`/`psnr_numpy`,
 * src/metrics.py:15-67; `roc_auc_score` call sites src/evaluate.py:245,263-265).  Each entry
 * point below names the reference interface it stands in for.  INTEGRATION.md shows the
 * ctypes binding a maintainer of the reference would add.
 *
 * Conventions
 *   - every function returns 0 on success, non-zero on failure; `srad_last_error()` gives the
 *     thread-local message.  Nothing aborts.
 *   - all pointers named `dev_*`, `x`, `y`, `workspace`, `arena` are DEVICE pointers owned by the
 *     caller (e.g. PyTorch's caching allocator).  The library allocates no device memory.
 *   - `stream` is a `hipStream_t` passed as `void*`; all work is enqueued on it asynchronously.
 *     Functions that return host scalars say so and synchronise that stream.
 *   - a handle is not thread-safe: one per process / rank.
 *   - images are fp32 NCHW in [0, rgb_range], exactly what the reference modules take/return.
 */
#ifndef SRAD_H
#define SRAD_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SRAD_PRECISION_F32 0  /* v_mfma_f32_16x16x4_f32: exact fp32, the parity mode            */
#define SRAD_PRECISION_BF16 1 /* v_mfma_f32_16x16x32_bf16, fp32 accumulate, fp32 residual stream */
#define SRAD_PRECISION_BF16X3 2 /* split-bf16: every MFMA operand as hi + lo bf16 terms, three bf16 MFMAs per product
                                 * (hi.hi + hi.lo + lo.hi), fp32 accumulate - fp32-grade outputs (the 1e-3 / AUC +-0.002 bar)
                                 * from the bf16 matrix pipe.  Inference only (forward / scoring); training is fp32 or bf16. */

const char* srad_last_error(void);
int srad_version(void);

/* ------------------------------------------------------------------ DRCT-L (src/drct.py:716-898) */
typedef struct srad_drct srad_drct_t;
typedef struct {
  int32_t in_chans;    /* opt.n_colors                          */
  int32_t img_size;    /* opt.img_size (LR side built for)      */
  int32_t window_size; /* opt.window_size = img_size // 4       */
  int32_t upscale;     /* opt.upscale (2^n)                     */
  int32_t embed_dim;   /* 180                                   */
  int32_t n_rdg;       /* len(opt.depths) = 12                  */
  int32_t num_heads;   /* 6                                     */
  int32_t gc;          /* 32                                    */
  int32_t num_feat;    /* 64                                    */
  float mlp_ratio;     /* 2                                     */
  float img_range;     /* 1.0                                   */
  int32_t precision;   /* SRAD_PRECISION_*                      */
  int32_t use_graph;   /* replay the forward from a hipGraph when shapes/pointers repeat */
} srad_drct_config;

/* DRCT.__init__ (src/drct.py:718-849) */
int srad_drct_create(const srad_drct_config* cfg, srad_drct_t** out);
void srad_drct_destroy(srad_drct_t* h);
/* packed-weight arena the caller must provide before loading parameters */
int srad_drct_arena_bytes(const srad_drct_t* h, size_t* bytes);
int srad_drct_bind_arena(srad_drct_t* h, void* arena, size_t bytes);
/* state_dict surface: fp32 tensors in the reference's key names (src/model.py:114-116,155-170) */
int srad_drct_num_params(const srad_drct_t* h);
int srad_drct_param_info(const srad_drct_t* h, int idx, const char** name, int64_t* numel);
int srad_drct_set_param(srad_drct_t* h, const char* name, const float* dev_src, int64_t numel, void* stream);
int srad_drct_workspace_bytes(const srad_drct_t* h, int B, int H, int W, size_t* bytes);
/* DRCT.forward (src/drct.py:886-898): x [B,C,H,W] -> y [B,C,H*s,W*s]; H, W multiples of the window */
int srad_drct_forward(srad_drct_t* h, const float* x, int B, int H, int W, float* y, void* workspace,
                      size_t workspace_bytes, void* stream);
/* FLOPs (2 per MAC) of one forward at this shape, for roofline accounting */
int srad_drct_flops(const srad_drct_t* h, int B, int H, int W, double* flops);

/* ------------------------------------------------------------------ DRCT training step (src/trainer.py:152-222)
 * loss.backward() and optimizer.step() for the DRCT model.  Parameters and gradients live in two flat fp32
 * device buffers owned by the caller (PyTorch parameters / .grad are views into them): entry idx of
 * srad_drct_param_info starts at srad_drct_train_param_offset(idx) floats, total srad_drct_train_param_floats. */
int srad_drct_train_param_floats(srad_drct_t* h, int64_t* total);
int srad_drct_train_param_offset(srad_drct_t* h, int idx, int64_t* off_floats);
/* second caller-owned arena: transposed weight packs for the data gradients (bind after srad_drct_bind_arena) */
int srad_drct_train_arena_bytes(srad_drct_t* h, size_t* bytes);
int srad_drct_train_bind(srad_drct_t* h, void* train_arena, size_t bytes);
/* refresh all packed weights from the flat fp32 parameters: call after loading and after every optimizer step */
int srad_drct_sync_params(srad_drct_t* h, const float* dev_flat_params, void* stream);
int srad_drct_train_workspace_bytes(const srad_drct_t* h, int B, int H, int W, size_t* bytes);
/* model.train() forward (src/drct.py:886-898 with DropPath 107-133 active): keep_scale is a device array
 * [2 * n_rdg * 5][B] of per-sample factors floor(keep + U) / keep (rows 2j / 2j+1: attention / MLP branch of
 * block j), or NULL for no DropPath.  Leaves the saved activations in `workspace` for srad_drct_backward. */
int srad_drct_forward_train(srad_drct_t* h, const float* x, int B, int H, int W, float* y, const float* keep_scale,
                            void* workspace, size_t workspace_bytes, void* stream);
/* loss.backward() (src/trainer.py:198): dy = dLoss/dy [B,C,H*s,W*s]; parameter gradients are ACCUMULATED into
 * dev_flat_grad; dx (optional) receives dLoss/dx.  on_bucket (optional) is called on the host as soon as the
 * last kernel writing gradient bucket b has been enqueued (buckets: srad_drct_bucket_range, in completion
 * order) so a data-parallel caller can start that bucket's all-reduce while the backward continues. */
typedef void (*srad_bucket_fn)(void* user, int bucket);
int srad_drct_num_buckets(const srad_drct_t* h);
int srad_drct_bucket_range(srad_drct_t* h, int bucket, int64_t* off_floats, int64_t* n_floats);
int srad_drct_backward(srad_drct_t* h, const float* dy, int B, int H, int W, const float* keep_scale, float* dx,
                       float* dev_flat_grad, void* workspace, size_t workspace_bytes, void* stream,
                       srad_bucket_fn on_bucket, void* user);
/* gradient seed of nn.L1Loss(reduction='mean') (src/loss.py:84): out = scale * sign(a - b), scale = 1/numel */
int srad_l1_grad(const float* a, const float* b, float* out, int64_t n, float scale, void* stream);
/* torch.optim.Adam.step on flat buffers (src/trainer.py:49-59; L2 weight decay, amsgrad off); step counts from 1;
 * grad_scale multiplies the gradient first (1/world_size after a summing all-reduce) */
int srad_adam_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t n, float lr,
                   float beta1, float beta2, float eps, float weight_decay, int step, float grad_scale, void* stream);
/* the same step with the per-step scalars in device memory: dev_hyper = [lr, 1 - beta1^step, sqrt(1 - beta2^step),
 * grad_scale] (4 floats), so a hipGraph capturing a whole training step can be replayed for every step */
int srad_adam_step_dev(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t n, float beta1,
                       float beta2, float eps, float weight_decay, const float* dev_hyper, void* stream);
/* dst[0..3] = (a, b, c, d), passed by value through a kernel launch (no host copy, no synchronisation) */
int srad_set4(float* dst, float a, float b, float c, float d, void* stream);

/* ------------------------------------------------------------------ DRN-L (src/drn.py:160-270) */
typedef struct srad_drn srad_drn_t;
typedef struct {
  int32_t n_colors;  /* opt.n_colors                */
  int32_t scale;     /* max(opt.scale): 2, 4 or 8   */
  int32_t n_blocks;  /* opt.n_blocks                */
  int32_t n_feats;   /* opt.n_feats                 */
  float negval;      /* opt.negval = 0.2            */
  float rgb_range;   /* opt.rgb_range = 255         */
  int32_t precision;
  int32_t use_graph;
} srad_drn_config;

int srad_drn_create(const srad_drn_config* cfg, srad_drn_t** out);
void srad_drn_destroy(srad_drn_t* h);
int srad_drn_arena_bytes(const srad_drn_t* h, size_t* bytes);
int srad_drn_bind_arena(srad_drn_t* h, void* arena, size_t bytes);
int srad_drn_num_params(const srad_drn_t* h);
int srad_drn_param_info(const srad_drn_t* h, int idx, const char** name, int64_t* numel);
int srad_drn_set_param(srad_drn_t* h, const char* name, const float* dev_src, int64_t numel, void* stream);
int srad_drn_workspace_bytes(const srad_drn_t* h, int B, int H, int W, size_t* bytes);
/* DRN.forward (src/drn.py:241-270): returns phase+1 images coarse -> fine; ys[i] is the device
 * pointer of output i with shape [B, C, H*2^i*..]: ys[0] = LR size, ys[phase] = H*scale. */
int srad_drn_forward(srad_drn_t* h, const float* x, int B, int H, int W, float* const* ys, int n_out,
                     void* workspace, size_t workspace_bytes, void* stream);
int srad_drn_flops(const srad_drn_t* h, int B, int H, int W, double* flops);

/* Dual regression model DownBlock(opt, 2) (src/model.py:8-44,78-82): two bias-free 3x3 convs,
 * the first stride 2 + LeakyReLU(negval).  w0 [n_feats,C,3,3], w1 [C,n_feats,3,3] fp32 device.
 * x [B,C,H,W] -> y [B,C,H/2,W/2].  workspace >= srad_dual_workspace_bytes. */
int srad_dual_workspace_bytes(int B, int C, int H, int W, int n_feats, size_t* bytes);
int srad_dual_forward(const float* w0, const float* w1, int C, int n_feats, float negval, const float* x, int B,
                      int H, int W, float* y, void* workspace, size_t workspace_bytes, int precision, void* stream);

/* ------------------------------------------------------------------ DRN training step (src/trainer.py:161-205)
 * Same scheme as the DRCT training entry points: flat fp32 parameter / gradient buffers with the offsets of
 * srad_drn_train_param_offset, a second caller-owned arena, srad_drn_sync_params after every optimizer step.
 * n_feats must be a multiple of 4 (the x2 / x4 presets). */
int srad_drn_train_param_floats(srad_drn_t* h, int64_t* total);
int srad_drn_train_param_offset(srad_drn_t* h, int idx, int64_t* off_floats);
int srad_drn_train_arena_bytes(srad_drn_t* h, size_t* bytes);
int srad_drn_train_bind(srad_drn_t* h, void* train_arena, size_t bytes);
int srad_drn_sync_params(srad_drn_t* h, const float* dev_flat_params, void* stream);
int srad_drn_train_workspace_bytes(const srad_drn_t* h, int B, int H, int W, size_t* bytes);
/* model.train() forward of DRN (src/drn.py:241-270); saved activations stay in `workspace` for srad_drn_backward */
int srad_drn_forward_train(srad_drn_t* h, const float* x, int B, int H, int W, float* const* ys, int n_out,
                           void* workspace, size_t workspace_bytes, void* stream);
/* dys[j] = dLoss/d(output j) (NCHW device pointers, NULL when output j is not in the loss); parameter gradients are
 * ACCUMULATED into dev_flat_grad (incl. the trainable MeanShift layers, hazard H4).  on_bucket (optional): as
 * srad_drct_backward - called on the host when the last kernel writing gradient bucket b has been enqueued; buckets
 * (srad_drn_bucket_range, completion order): 0 = tail convs, 1..phase = up phases finest first, phase + 1 = the rest. */
int srad_drn_num_buckets(const srad_drn_t* h);
int srad_drn_bucket_range(srad_drn_t* h, int bucket, int64_t* off_floats, int64_t* n_floats);
int srad_drn_backward(srad_drn_t* h, const float* const* dys, int n_out, int B, int H, int W, float* dev_flat_grad,
                      void* workspace, size_t workspace_bytes, void* stream, srad_bucket_fn on_bucket, void* user);
/* Backward of the dual regression model (srad_dual_forward; src/trainer.py:168-185 back-propagates through it into the
 * SR outputs): dw0 [n_feats,C,3,3] / dw1 [C,n_feats,3,3] accumulated, dx [B,C,H,W] optional; H and W even */
int srad_dual_backward_workspace_bytes(int B, int C, int H, int W, int n_feats, size_t* bytes);
int srad_dual_backward(const float* w0, const float* w1, int C, int n_feats, float negval, const float* x, int B, int H,
                       int W, const float* dy, float* dx, float* dw0, float* dw1, void* workspace, size_t workspace_bytes,
                       int precision, void* stream);

/* ------------------------------------------------------------------ scorer (src/evaluate.py:204-265) */
/* `mul(255/range).clamp(0,255).byte()` TRUNCATING u8 conversion + NCHW->HWC (evaluate.py:214-215). */
int srad_to_u8_hwc(const float* x, int B, int C, int H, int W, float rgb_range, uint8_t* out, void* stream);
/* quantize (src/trainer.py:45-47): mul, clamp, round-half-even, div; in-place allowed */
int srad_quantize(const float* x, float* y, int64_t n, float rgb_range, void* stream);
/* Per-pair scores for `n_img` u8 HWC image pairs (sr, hr, each [n_img,H,W,C]):
 *   ssim_out[i*n_ws + j] = ssim_numpy(hr_i/255, sr_i/255, ws[j])   (src/metrics.py:26-67)
 *   mse_out[i] = mean((sr_i/255 - hr_i/255)^2), psnr_out[i] = psnr_numpy (inf when mse == 0)
 * Outputs are DEVICE double arrays; workspace >= srad_score_workspace_bytes. */
int srad_score_workspace_bytes(int n_img, int H, int W, size_t* bytes);
int srad_score_pairs(const uint8_t* sr, const uint8_t* hr, int n_img, int H, int W, int C, const int32_t* ws_host,
                     int n_ws, double* ssim_out, double* mse_out, double* psnr_out, void* workspace,
                     size_t workspace_bytes, void* stream);
/* Validation metrics of Trainer.test (src/metrics.py:70-108) on fp32 NCHW tensors, one value per
 * image, device double outputs; reproduces the zero padding, 4-px shave and the 255^2 constants. */
int srad_val_metrics(const float* sr, const float* hr, int B, int C, int H, int W, float rgb_range,
                     double* psnr_out, double* ssim_out, void* workspace, size_t workspace_bytes, void* stream);
/* Binary ROC-AUC == sklearn.metrics.roc_auc_score (ties count one half).  labels/scores are HOST
 * arrays (n is the number of test images, a few hundred); returns non-zero if one class is absent. */
int srad_roc_auc(const int32_t* labels, const double* scores, int n, double* auc);

/* mean |a-b| (nn.L1Loss, src/loss.py:84) -> *out (device double); workspace >= srad_l1_workspace_bytes */
int srad_l1_workspace_bytes(size_t* bytes);
int srad_l1_loss(const float* a, const float* b, int64_t n, double* out, void* workspace, void* stream);

/* The loss types of the reference's loss factory `Loss(opt, ckp)` (src/loss.py:72-121; the CLI only ever builds '1*L1'):
 *   SRAD_LOSS_L1   nn.L1Loss(reduction='mean')                     (src/loss.py:84)
 *   SRAD_LOSS_MSE  nn.MSELoss()                                    (src/loss.py:82)
 *   SRAD_LOSS_PSNR PSNRLoss: -10 log10(255^2 / (mse + 1e-8))       (src/loss.py:63-70)
 *   SRAD_LOSS_SSIM SSIMLoss / calc_ssim: sum(1 - ssim_map) / batch_size with the 10-px shave, zero-padded 11x11 mean
 *                  filter and the 255^2-scaled constants           (src/loss.py:9-61)
 * sr [B,C,sH,sW] and hr [B,C,H,W] fp32 NCHW (sH, sW may exceed H, W only for SSIM, which crops sr like the reference).
 * forward: out2[0] = loss, out2[1] = state for the backward (device doubles).  backward: dsr (+)= weight * gscale_dev[0]
 * * dLoss/dsr (gscale_dev may be NULL = 1); SSIM's backward reads the workspace its forward filled. */
#define SRAD_LOSS_L1 0
#define SRAD_LOSS_MSE 1
#define SRAD_LOSS_PSNR 2
#define SRAD_LOSS_SSIM 3
int srad_loss_workspace_bytes(int kind, int B, int C, int H, int W, size_t* bytes);
int srad_loss_forward(int kind, const float* sr, const float* hr, int B, int C, int sH, int sW, int H, int W, float rgb_range,
                      int batch_size, double* out2, void* workspace, size_t workspace_bytes, void* stream);
int srad_loss_backward(int kind, const float* sr, const float* hr, int B, int C, int sH, int sW, int H, int W, float rgb_range,
                       int batch_size, const double* fwd_out2, const float* gscale_dev, float weight, float* dsr, int accumulate,
                       void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------ single operators
 * NHWC / token-major fp32 activations [rows][ld]; channel counts and row strides multiples of 4 floats. */
/* Linear / 1x1 conv / 3x3 conv (pad 1, stride 1|2) with fused prologue + epilogue:
 *   y = act(LN?(x) . w^T + bias) * alpha + r ;  w in PyTorch layout [N][Cin][taps] (packed into scratch)
 *   act: 0 none, 1 GELU(erf), 2 LeakyReLU(slope), 3 ReLU; ps = 2 -> PixelShuffle(2) scatter of the output
 *   nn.Linear (src/drct.py:178-181,262-264), nn.Conv2d (src/drct.py:782,837,844-847; src/drn.py:29-33,95-112) */
size_t srad_op_gemm_scratch_bytes(int precision, int N, int Cin, int ntaps);
int srad_op_gemm(int precision, const float* x, int ldx, int B, int Hi, int Wi, int Cin, const float* w, int N,
                 int ntaps, int stride, const float* bias, const float* ln_g, const float* ln_b, int act, float slope,
                 float alpha, const float* r, int ldr, float* y, int ldy, int yoff, int ps, void* scratch,
                 size_t scratch_bytes, void* stream);
/* WindowAttention core + cyclic shift + window partition/reverse (src/drct.py:271-302,472-506):
 * qkv [B*H*W][3][heads][hdp] (each head slice padded to hdp floats, hdp % 4 == 0) -> out [B*H*W][d] */
int srad_op_window_attn(int precision, const float* qkv, float* out, const float* table, int B, int H, int W, int ws,
                        int shift, int d, int heads, int hdp, void* stream);
/* nn.LayerNorm over the last dimension, eps 1e-5 (src/drct.py:798,833) */
int srad_op_layernorm(const float* x, int ldx, float* y, int ldy, int rows, int C, const float* g, const float* b,
                      void* stream);

/* The two fused launches a Swin block becomes in bf16 and split-bf16 mode with 8 x 8 windows (BASELINE configs C2 / C4; what
 * bench.py times).  precision = SRAD_PRECISION_BF16 | SRAD_PRECISION_BF16X3; in split-bf16 mode the hand-off between the two
 * (out of the first, attn of the second) is fp32 and out_bf16 must be 0.  Weights in PyTorch layout (device fp32), packed into `scratch` (>= srad_op_swin_scratch_bytes, 256-byte aligned).
 *   srad_op_qkv_attn : norm1 -> attn.qkv -> cyclic shift + window partition -> softmax(q k^T * scale + relative position
 *                      bias + 0/-100 shift mask) v -> window reverse + shift back
 *                      (SwinTransformerBlock.forward src/drct.py:477-504 up to attn.proj; WindowAttention.forward 271-299)
 *                      x [B*H*W][ldx] (columns [0,d)) , w_qkv [3d][d], b_qkv [3d], table [225][heads] -> out [B*H*W][d],
 *                      fp32, or bf16 with out_bf16 = 1 (the form the engines hand to the second half)
 *   srad_op_mlp_block: x1 = shortcut + attn.proj(attn); x2 = x1 + mlp(norm2(x1)) (src/drct.py:300, 509-510, 184-190), then
 *                      the RDG's 1x1 adjust conv: y[:, yoff:yoff+no] = act(adjust(x2) + b) * alpha (+ r) (src/drct.py:389-396)
 *                      fm = token rows per workgroup (16 | 32 | 64, 0 = the engine's choice for M); attn is a bf16 [M][d]
 *                      array: the first half's output in the form the MFMA takes it */
size_t srad_op_swin_scratch_bytes(int d, int heads, int m, int no);
int srad_op_qkv_attn(int precision, const float* x, int ldx, int B, int H, int W, int shift, int d, int heads, const float* ln_g,
                     const float* ln_b, const float* w_qkv, const float* b_qkv, const float* table, void* out, int out_bf16,
                     void* scratch, size_t scratch_bytes, void* stream);
int srad_op_mlp_block(int precision, int M, int d, int m, int no, int fm, const void* attn, const float* shortcut, int ld_short,
                      const float* w_proj, const float* b_proj, const float* ln_g, const float* ln_b, const float* w_fc1,
                      const float* b_fc1, const float* w_fc2, const float* b_fc2, const float* w_adj, const float* b_adj, int act,
                      float slope, float alpha, const float* r, int ldr, float* y, int ldy, int yoff, void* scratch,
                      size_t scratch_bytes, void* stream);

/* The two kernels of BASELINE config C5's attention (bf16, 64 x 64 windows = 4096-token windows; src/main.py:286), each on its own:
 *   srad_op_ln_qkv              norm1 + attn.qkv (src/drct.py:477, 278) -> qkv_h [M][3][heads][hdp] bf16, the attention's MFMA
 *                               operands: q times qscale (= head_dim^-0.5 log2 e, srad_op_window_attn_qscale), padding columns 0,
 *                               column head_dim of every v slice 1 (P.V then also yields the softmax denominator); M % 64 == 0
 *   srad_op_window_attn_bf16_in WindowAttention.forward on those operands (src/drct.py:281-299 and the roll / partition / reverse
 *                               around it, 482-504): online softmax over 64-key chunks = rows of the window, relative position
 *                               bias rows sliding through an LDS ring, 0 / -100 shift mask; out [B*H*W][d] fp32 */
float srad_op_window_attn_qscale(int ws, int shift, int d, int heads);
size_t srad_op_ln_qkv_scratch_bytes(int d, int heads);
int srad_op_ln_qkv(const float* x, int ldx, int M, int d, int heads, const float* ln_g, const float* ln_b, const float* w_qkv,
                   const float* b_qkv, void* qkv_h, int hdp, float qscale, void* scratch, size_t scratch_bytes, void* stream);
int srad_op_window_attn_bf16_in(const void* qkv_h, float* out, const float* table, int B, int H, int W, int ws, int shift, int d,
                                int heads, int hdp, void* stream);

/* Backward operators (autograd of the rows above).
 * Weight/bias gradient of Linear / conv: dw[N][Cin][taps] += alpha * dy^T A(x), db[N] += alpha * colsum(dy);
 * dy [B*Ho*Wo][ldy], x [B*Hi*Wi][ldx]; N, Cin multiples of 4; row_scale optional per-sample factor [B].
 * Split-K partials are summed in a fixed order by a second kernel: results are bit-reproducible. */
int srad_op_wgrad(int precision, const float* dy, int ldy, const float* x, int ldx, int B, int Hi, int Wi, int N,
                  int Cin, int ntaps, int stride, const float* row_scale, float alpha, float* dw, float* db,
                  void* workspace, void* stream);
/* The 80 -> 80 channel 3x3 convolution (src/drn.py:143-158, RCAB) with bf16 operands, as DRN's bf16 chains issue it in the
 * forward and as the data gradient of the backward: x_h [B*H*W][80] bf16; w [80][80][3][3] fp32; output bf16 (y_h) or fp32 (y);
 * residual operand bf16 (r_h) or fp32 (r) or neither; rmode 0 = add, 2 = multiply by (r > 0 ? 1 : slope) (backward through
 * ReLU / LeakyReLU); pool_part (optional): [B*H*W/128][80] column sums per 128-pixel tile.  H % 4 == 0, W % 32 == 0,
 * B*H*W >= 8192; scratch >= srad_op_gemm_scratch_bytes(SRAD_PRECISION_BF16, 80, 80, 9). */
int srad_op_conv80_h(const void* x_h, const float* w, const float* bias, int act, float slope, const void* r_h, const float* r,
                     int rmode, int B, int H, int W, void* y_h, float* y, float* pool_part, void* scratch, size_t scratch_bytes,
                     void* stream);
/* Its weight / bias gradient from a bf16 dY and a bf16 (x_bf16 = 1) or fp32 X: dw [80][80][3][3], db [80] accumulated;
 * workspace as for srad_op_wgrad.  W % 32 == 0, B*H*W >= 8192. */
int srad_op_wgrad_conv9_h(const void* dy_h, const void* x, int x_bf16, int B, int H, int W, float* dw, float* db, void* workspace,
                          void* stream);
/* The same for a Linear layer, issued as the training step issues it (queued, one deferred launch, reduce).  x_bf16 /
 * dy_bf16 = 1: that operand is a bf16 array (ld in elements; a bf16 dY is taken as already multiplied by its per-sample
 * factor, so row_scale must be null with it, and it needs a bf16 X).  precision SRAD_PREC_BF16 for bf16 operands. */
int srad_op_wgrad_deferred(int precision, const void* dy, int ldy, int dy_bf16, const void* x, int ldx, int x_bf16, int M, int N,
                           int Cin, const float* row_scale, int rps, float alpha, float* dw, float* db, void* workspace,
                           void* stream);
/* split-K workspace of srad_op_wgrad (256-byte aligned scratch, contents irrelevant, reusable by later calls on the
 * same stream) */
size_t srad_op_wgrad_workspace_bytes(void);
/* Data gradient dx = (dy . w) * alpha * row_scale (* gelu'(r) if rmode 1, * lrelu'(r) if rmode 2), stride 1:
 * the forward GEMM on the transposed pack of w [N][Cin][taps]; scratch >= srad_op_gemm_scratch_bytes(p, Cin, N, taps) */
int srad_op_dgrad(int precision, const float* dy, int ldy, int B, int H, int W, int N, const float* w, int Cin,
                  int ntaps, const float* r, int ldr, int rmode, float slope, float alpha, const float* row_scale,
                  float* dx, int ldx, void* scratch, size_t scratch_bytes, void* stream);
/* LayerNorm backward: out (+)= dLN(dxn; x, gamma) + dres; dgamma / dbeta accumulated; workspace as for srad_op_wgrad */
int srad_op_layernorm_bwd(const float* dxn, const float* x, int ldx, const float* gamma, const float* dres, float* out,
                          int accumulate, float* dgamma, float* dbeta, int rows, int C, void* workspace, void* stream);
/* Window attention backward (window size 8): qkv head-padded as for srad_op_window_attn, dout [T][d],
 * dqkv [T][3d] compact, dtable accumulated; workspace as for srad_op_wgrad */
int srad_op_window_attn_bwd(int precision, const float* qkv, const float* dout, float* dqkv, const float* table,
                            float* dtable, int B, int H, int W, int ws, int shift, int d, int heads, int hdp,
                            void* workspace, void* stream);
/* The all-bf16 form of the same backward, for head dims <= 128 (what the DRCT training step runs): qkv_h [T][3][heads][hp]
 * bf16 with q ALREADY multiplied by head_dim^-0.5 (the forward's MFMA operand), dout_h [T][heads][hp] bf16 (columns at
 * or beyond the head dim are ignored), dqkv_h [T][3 d] bf16 out; hp % 8 == 0; table / dtable fp32. */
int srad_op_window_attn_bwd_h(const void* qkv_h, const void* dout_h, void* dqkv_h, const float* table, float* dtable,
                              int B, int H, int W, int ws, int shift, int d, int heads, int hp, void* workspace, void* stream);
/* Fused backward of the MLP branch of a Swin block (bf16 MFMA; src/drct.py:510, 184-190, 389-396, 300):
 *   [KA > 0] dx2 = aalpha * (dA (.) lrelu'(y_act)) . w_adj          (written; dA (.) lrelu' -> dA_out if given)
 *   dh  = (dx2 . w_fc2) * rs2 * gelu'(hpre);  dx1 = dx2 + LayerNorm'(dh . w_fc1; x1, gamma);  dgamma / dbeta accumulated
 *   [w_proj] dO = (dx1 . w_proj) * rs1
 * Weights in PyTorch layout: w_fc1 [m][d], w_fc2 [d][m], w_adj [KA][d], w_proj [d][d]; rs1 / rs2 per sample (rps rows
 * each) or NULL.  scratch >= srad_op_mlp_bwd_scratch_bytes (256-byte aligned), workspace as for srad_op_wgrad. */
size_t srad_op_mlp_bwd_scratch_bytes(int d, int m, int KA);
int srad_op_mlp_bwd(int M, int d, int m, float* dx2, const float* hpre, const float* x1, const float* gamma,
                    const float* w_fc1, const float* w_fc2, const float* rs2, int rps, float* dh, float* dx1, float* dgamma,
                    float* dbeta, int KA, const float* dA, int ld_dA, const float* y_act, int ld_y, float slope,
                    float aalpha, const float* w_adj, float* dA_out, const float* w_proj, const float* rs1, float* dO,
                    void* scratch, size_t scratch_bytes, void* workspace, void* stream);
/* Data gradient of a Linear + backward of the LayerNorm that fed it (bf16 MFMA): out (+)= dres + LayerNorm'(dY . w; x,
 * gamma), w [K][d] in PyTorch layout, dY [M][K]; dgamma / dbeta accumulated */
size_t srad_op_lin_ln_bwd_scratch_bytes(int K, int d);
int srad_op_lin_ln_bwd(int M, int K, int d, const float* dY, const float* w, const float* x, int ldx, const float* gamma,
                       const float* dres, float* out, int ld_out, int accumulate, float* dgamma, float* dbeta, void* scratch,
                       size_t scratch_bytes, void* workspace, void* stream);

/* ------------------------------------------------------------------ diagnostics
 * Per-kernel-class device timing with HIP events on the launch stream (bench.py's roofline numbers). */
int srad_prof_enable(int on);
int srad_prof_num_classes(void);
const char* srad_prof_class_name(int cls);
int srad_prof_collect(int64_t* launches, double* ms, double* flops, double* bytes);
/* Back-to-back launches of one kernel, average microseconds per launch (tools/gemm_bench.py). */
int srad_bench_gemm(int precision, const float* x, int ldx, int B, int Hi, int Wi, int Cin, const float* w, int N,
                    int ntaps, int stride, const float* bias, const float* ln_g, const float* ln_b, int act,
                    const float* r, int ldr, float* y, int ldy, int hsplit_hd, int hsplit_hdp, void* scratch,
                    size_t scratch_bytes, int iters, float* us_out, void* stream);
int srad_bench_mlp_block(int M, int d, int m, int no, const void* attn /* bf16 [M][320] */, const float* shortcut, float* y,
                         const float* w_fp32, void* scratch, size_t scratch_bytes, int dbg, int iters, float* us_out,
                         void* stream);
int srad_bench_window_attn(int precision, const float* qkv, float* out, const float* table, int B, int H, int W, int ws,
                           int shift, int d, int heads, int hdp, int iters, float* us_out, void* stream);
int srad_bench_qkv_attn(const float* x, int ldx, int B, int H, int W, int shift, int d, int heads, const float* w_fp32,
                        void* out /* written as bf16 [B*H*W][d] */, void* scratch, size_t scratch_bytes, int iters, float* us_out,
                        void* stream);
/* tools/wgrad_bench.py: one Swin block's five weight gradients as the training step issues them (one deferred launch +
 * the reduce), `iters` times; storage bit 0 / 1: the X / dY operands are bf16. */
int srad_bench_wgrad_block(int M, int d, int hidden, int KA, int storage, const void* xbuf, const void* ybuf, float* dw,
                           void* workspace, int iters, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SRAD_H */
