// drct_engine.h - the DRCT handle shared by the inference engine (drct.hip) and the training engine
// (drct_train.hip).
#pragma once
#include "engine.h"
#include "../../include/srad.h"

struct SwinW {
  int d, heads, hidden, shift;
  int n1g, n1b, n2g, n2b, table;
  ConvW qkv, proj, fc1, fc2, adjust;
};

static inline int hdp_of(int d, int heads) { return srad_round_up(d / heads, 4); }

struct SyncDesc;

struct srad_drct {
  srad_drct_config cfg;
  ParamTable pt;
  ConvW conv_first, conv_after_body, conv_before_up, conv_last;
  std::vector<ConvW> up;
  int pe_g, pe_b, norm_g, norm_b;
  std::vector<SwinW> blocks;      // n_rdg * 5
  int dmax, hmax, qkvmax;         // widest block dim / hidden / head-padded qkv row
  bool fuse_mlp = true;           // bf16: second half of each Swin block as one launch (kernels_fused.hip)
  GraphCache gc;
  // ---- training (drct_train.hip) ----
  std::vector<int64_t> flat_off;  // per table entry: offset (floats) in the flat fp32 parameter / gradient buffers
  int64_t flat_total = 0;
  std::vector<size_t> t_off;      // per table entry: byte offset of the transposed pack in the training arena
  size_t t_wgrad_off = 0;         // byte offset of the weight-gradient split-K workspace
  size_t t_desc_off = 0;          // byte offset of the device descriptor table inside the training arena
  size_t t_bytes = 0;
  int n_sync_blocks = 0;
  char* tarena = nullptr;
  bool train_ready = false;
};
