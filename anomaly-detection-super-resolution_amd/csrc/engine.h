// engine.h - host-side bookkeeping shared by the DRCT and DRN engines: a parameter table that
// maps the reference's state-dict names onto offsets in a caller-owned device arena, the packed
// layer descriptors, a bump allocator for the per-call workspace, and hipGraph replay.
#pragma once
#include "srad_common.h"
#include <string>
#include <vector>

struct ParamEntry {
  std::string name;
  int64_t numel;     // elements of the fp32 source tensor
  size_t off;        // byte offset in the arena
  int packed;        // 1: GEMM weight packed by srad_launch_pack_weight, 0: raw fp32 copy
  int n, cin, ntaps; // packed geometry (real sizes of the source tensor)
  int n_pad = 0, grp_real = 0, grp_pad = 0;   // optional padding (srad_launch_pack_weight_padded)
  long long frag_off = -1;                    // >= 0: second copy as bf16 MFMA fragments (srad_launch_pack_weight_frag)
  bool tfrag = false;                         // training: keep W^T as bf16 MFMA fragments too (fused backward kernels)
  int qkv_heads = 0;                          // > 0: frag_off holds the per-head [q | k | v] fragment pack (srad_launch_pack_qkv_frag)
  long long frag_lo_off = -1;                 // >= 0 (split-bf16): the lo terms of the fragment pack at frag_off, same geometry
};

struct ConvW {       // one Linear / conv layer
  int w = -1, b = -1;      // indices into the parameter table (-1: no bias)
  int n = 0, cin = 0, ntaps = 1;
};

struct ParamTable {
  std::vector<ParamEntry> entries;
  size_t bytes = 0;
  int prec = SRAD_PREC_F32;
  char* arena = nullptr;
  size_t arena_bytes = 0;

  int add_raw(const std::string& name, int64_t numel) {
    ParamEntry e{name, numel, bytes, 0, 0, 0, 0};
    bytes += srad_align_up((size_t)numel * 4, 256);
    entries.push_back(e);
    return (int)entries.size() - 1;
  }
  int add_packed(const std::string& name, int n, int cin, int ntaps) {
    ParamEntry e{name, (int64_t)n * cin * ntaps, bytes, 1, n, cin, ntaps};
    bytes += srad_align_up(srad_packed_bytes(prec, n, cin, ntaps), 256);
    entries.push_back(e);
    return (int)entries.size() - 1;
  }
  ConvW add_layer(const std::string& prefix, int n, int cin, int ntaps, bool bias) {
    ConvW c;
    c.n = n; c.cin = cin; c.ntaps = ntaps;
    c.w = add_packed(prefix + ".weight", n, cin, ntaps);
    if (bias) c.b = add_raw(prefix + ".bias", n);
    return c;
  }
  // Layer whose output rows are zero-padded to n_pad and whose input channels are regrouped from groups of
  // grp_real to grp_pad channels (0: keep).  The returned ConvW carries the PADDED sizes the GEMM runs with;
  // the bias slot is n_pad long: set() zeroes it and writes the first n entries.
  ConvW add_layer_padded(const std::string& prefix, int n, int cin, int ntaps, bool bias, int n_pad, int grp_real, int grp_pad) {
    const int cin_pad = grp_pad > 0 ? (cin / grp_real) * grp_pad : cin;
    ConvW c;
    c.n = n_pad; c.cin = cin_pad; c.ntaps = ntaps;
    ParamEntry e{prefix + ".weight", (int64_t)n * cin * ntaps, bytes, 1, n, cin, ntaps};
    e.n_pad = n_pad; e.grp_real = grp_real; e.grp_pad = grp_pad;
    bytes += srad_align_up(srad_packed_bytes(prec, n_pad, cin_pad, ntaps), 256);
    entries.push_back(e);
    c.w = (int)entries.size() - 1;
    if (bias) {
      ParamEntry b{prefix + ".bias", n, bytes, 0, 0, 0, 0};
      b.n_pad = n_pad;
      bytes += srad_align_up((size_t)n_pad * 4, 256);
      entries.push_back(b);
      c.b = (int)entries.size() - 1;
    }
    return c;
  }
  int find(const char* name) const {
    for (size_t i = 0; i < entries.size(); ++i)
      if (entries[i].name == name) return (int)i;
    return -1;
  }
  // Linear layer that is also kept as a fragment-major bf16 pack (the fused block kernels' operand)
  ConvW add_layer_frag(const std::string& prefix, int n, int cin, bool bias) {
    ConvW c = add_layer(prefix, n, cin, 1, bias);
    entries[c.w].frag_off = (long long)bytes;
    bytes += srad_align_up(srad_packed_bytes(SRAD_PREC_BF16, n, cin, 1), 256);
    if (prec == SRAD_PREC_BF16X3) {
      entries[c.w].frag_lo_off = (long long)bytes;
      bytes += srad_align_up(srad_packed_bytes(SRAD_PREC_BF16, n, cin, 1), 256);
    }
    return c;
  }
  // qkv Linear of a Swin block that is also kept as per-head fragments (the fused attention kernel's operand)
  void add_qkv_frag(const ConvW& c, int d, int heads) {
    entries[c.w].frag_off = (long long)bytes;
    entries[c.w].qkv_heads = heads;
    bytes += srad_align_up(srad_qkv_frag_bytes(d, heads), 256);
    if (prec == SRAD_PREC_BF16X3) {
      entries[c.w].frag_lo_off = (long long)bytes;
      bytes += srad_align_up(srad_qkv_frag_bytes(d, heads), 256);
    }
  }
  const void* ptr(int idx) const { return idx < 0 ? nullptr : arena + entries[idx].off; }
  const void* frag_ptr(int idx) const { return idx < 0 || entries[idx].frag_off < 0 ? nullptr : arena + entries[idx].frag_off; }
  const void* frag_lo_ptr(int idx) const { return idx < 0 || entries[idx].frag_lo_off < 0 ? nullptr : arena + entries[idx].frag_lo_off; }
  const float* fptr(int idx) const { return reinterpret_cast<const float*>(ptr(idx)); }

  int set(const char* name, const float* src, int64_t numel, hipStream_t s) {
    if (!arena) return srad_set_error(SRAD_ERR_STATE, "set_param(%s): no arena bound", name);
    const int i = find(name);
    if (i < 0) return srad_set_error(SRAD_ERR_ARG, "set_param: unknown parameter '%s'", name);
    const ParamEntry& e = entries[i];
    if (numel != e.numel)
      return srad_set_error(SRAD_ERR_ARG, "set_param(%s): got %lld elements, expected %lld", name, (long long)numel,
                            (long long)e.numel);
    if (e.packed) {
      if (e.frag_off >= 0 && e.qkv_heads > 0) SRAD_TRY(srad_launch_pack_qkv_frag(src, arena + e.frag_off, e.cin, e.qkv_heads, s));
      else if (e.frag_off >= 0) SRAD_TRY(srad_launch_pack_weight_frag(src, arena + e.frag_off, e.n, e.cin, s));
      if (e.frag_lo_off >= 0 && e.qkv_heads > 0) SRAD_TRY(srad_launch_pack_qkv_frag_lo(src, arena + e.frag_lo_off, e.cin, e.qkv_heads, s));
      else if (e.frag_lo_off >= 0) SRAD_TRY(srad_launch_pack_weight_frag_lo(src, arena + e.frag_lo_off, e.n, e.cin, s));
      return srad_launch_pack_weight_padded(prec, src, arena + e.off, e.n, e.cin, e.ntaps, e.n_pad > 0 ? e.n_pad : e.n, e.grp_real, e.grp_pad, s);
    }
    if (e.n_pad > numel) SRAD_CHECK_HIP(hipMemsetAsync(arena + e.off, 0, (size_t)e.n_pad * 4, s));
    SRAD_CHECK_HIP(hipMemcpyAsync(arena + e.off, src, (size_t)numel * 4, hipMemcpyDeviceToDevice, s));
    return SRAD_OK;
  }
};

struct Bump {
  char* base;
  size_t cap, used = 0;
  Bump(void* b, size_t c) : base(reinterpret_cast<char*>(b)), cap(c) {}
  float* take(size_t floats) {
    const size_t off = used;
    used += srad_align_up(floats * 4, 256);
    return reinterpret_cast<float*>(base + off);       // base may be null when only sizing
  }
};

// hipGraph replay of a fixed launch sequence keyed on the call's pointers and shape.
struct GraphCache {
  hipGraphExec_t exec = nullptr;
  hipGraph_t graph = nullptr;
  const void* key[3] = {nullptr, nullptr, nullptr};
  int shape[3] = {0, 0, 0};
  int seen = 0;   // eager runs with the current key (the first call runs eagerly, the second captures)
  bool matches(const void* a, const void* b, const void* c, int B, int H, int W) const {
    return key[0] == a && key[1] == b && key[2] == c && shape[0] == B && shape[1] == H && shape[2] == W;
  }
  void reset() {
    if (exec) (void)hipGraphExecDestroy(exec);
    if (graph) (void)hipGraphDestroy(graph);
    exec = nullptr; graph = nullptr; seen = 0;
  }
  void rekey(const void* a, const void* b, const void* c, int B, int H, int W) {
    reset();
    key[0] = a; key[1] = b; key[2] = c; shape[0] = B; shape[1] = H; shape[2] = W;
  }
};

// Run `body(stream)` eagerly the first time a (pointers, shape) key is seen, capture it into a
// hipGraph the second time, and replay the graph afterwards.
template <class F>
int srad_run_with_graph(GraphCache& gc, bool enable, const void* a, const void* b, const void* c, int B, int H, int W,
                        hipStream_t stream, F&& body) {
  if (!enable || stream == nullptr) return body(stream);   // the legacy default stream cannot be captured
  if (!gc.matches(a, b, c, B, H, W)) gc.rekey(a, b, c, B, H, W);
  if (gc.exec) {
    SRAD_CHECK_HIP(hipGraphLaunch(gc.exec, stream));
    return SRAD_OK;
  }
  if (gc.seen == 0) {
    gc.seen = 1;
    return body(stream);
  }
  SRAD_CHECK_HIP(hipStreamBeginCapture(stream, hipStreamCaptureModeThreadLocal));
  const int rc = body(stream);
  hipGraph_t g = nullptr;
  const hipError_t e = hipStreamEndCapture(stream, &g);
  if (rc) { if (g) (void)hipGraphDestroy(g); return rc; }
  if (e != hipSuccess) return srad_set_error(SRAD_ERR_HIP, "hipStreamEndCapture: %s", hipGetErrorString(e));
  gc.graph = g;
  SRAD_CHECK_HIP(hipGraphInstantiate(&gc.exec, g, nullptr, nullptr, 0));
  SRAD_CHECK_HIP(hipGraphLaunch(gc.exec, stream));
  return SRAD_OK;
}
