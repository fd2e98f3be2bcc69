// drct_engine.h - the DRCT handle shared by the inference engine (drct.hip) and the training engine
// (drct_train.hip).
#pragma once
#include "train_common.h"
#include "../../include/srad.h"

struct SwinW {
  int d, heads, hidden, shift;
  int n1g, n1b, n2g, n2b, table;
  ConvW qkv, proj, fc1, fc2, adjust;
};

static inline int hdp_of(int d, int heads) { return srad_round_up(d / heads, 4); }

struct srad_drct {
  srad_drct_config cfg;
  ParamTable pt;
  ConvW conv_first, conv_after_body, conv_before_up, conv_last;
  std::vector<ConvW> up;
  int pe_g, pe_b, norm_g, norm_b;
  std::vector<SwinW> blocks;      // n_rdg * 5
  int dmax, hmax, qkvmax;         // widest block dim / hidden / head-padded qkv row
  std::vector<char> saved_h;      // per Swin block, what the last training forward saved as bf16: 1 q | k | v, 2 the fc1 pre-activation (plan_block)
  bool fuse_mlp = true;           // bf16: second half of each Swin block as one launch (kernels_fused.hip)
  GraphCache gc;
  TrainState ts;                  // training (drct_train.hip)
  hipStream_t side = nullptr;     // weight gradients run here, next to the data-gradient chain on the caller's stream
  std::vector<hipEvent_t> events; // reused across backward calls
};
