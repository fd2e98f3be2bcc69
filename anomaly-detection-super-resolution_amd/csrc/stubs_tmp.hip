// temporary: entry points not implemented yet
#include "engine.h"
#include "../../include/srad.h"
#define NI(name) return srad_set_error(SRAD_ERR_STATE, name ": not implemented yet")
extern "C" {
int srad_drn_create(const srad_drn_config*, srad_drn_t**) { NI("drn_create"); }
void srad_drn_destroy(srad_drn_t*) {}
int srad_drn_arena_bytes(const srad_drn_t*, size_t*) { NI("drn"); }
int srad_drn_bind_arena(srad_drn_t*, void*, size_t) { NI("drn"); }
int srad_drn_num_params(const srad_drn_t*) { return 0; }
int srad_drn_param_info(const srad_drn_t*, int, const char**, int64_t*) { NI("drn"); }
int srad_drn_set_param(srad_drn_t*, const char*, const float*, int64_t, void*) { NI("drn"); }
int srad_drn_workspace_bytes(const srad_drn_t*, int, int, int, size_t*) { NI("drn"); }
int srad_drn_forward(srad_drn_t*, const float*, int, int, int, float* const*, int, void*, size_t, void*) { NI("drn"); }
int srad_drn_flops(const srad_drn_t*, int, int, int, double*) { NI("drn"); }
int srad_dual_workspace_bytes(int, int, int, int, int, size_t*) { NI("dual"); }
int srad_dual_forward(const float*, const float*, int, int, float, const float*, int, int, int, float*, void*, size_t, int, void*) { NI("dual"); }
int srad_to_u8_hwc(const float*, int, int, int, int, float, uint8_t*, void*) { NI("u8"); }
int srad_quantize(const float*, float*, int64_t, float, void*) { NI("quantize"); }
int srad_score_workspace_bytes(int, int, int, size_t*) { NI("score"); }
int srad_score_pairs(const uint8_t*, const uint8_t*, int, int, int, int, const int32_t*, int, double*, double*, double*, void*, size_t, void*) { NI("score"); }
int srad_val_metrics(const float*, const float*, int, int, int, int, float, double*, double*, void*, size_t, void*) { NI("val"); }
int srad_roc_auc(const int32_t*, const double*, int, double*) { NI("auc"); }
int srad_l1_loss(const float*, const float*, int64_t, double*, void*) { NI("l1"); }
}
