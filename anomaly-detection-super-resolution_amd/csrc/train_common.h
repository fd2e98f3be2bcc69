// train_common.h - training plumbing shared by the DRCT and DRN engines: the flat fp32 parameter / gradient layout,
// the training arena (transposed weight packs, descriptor table, split-K workspace) and the one-launch re-pack of
// every parameter after an optimizer step.
#pragma once
#include "engine.h"
#include <type_traits>

struct TrainState {
  std::vector<int64_t> flat_off;  // per table entry: offset (floats) in the flat fp32 parameter / gradient buffers
  int64_t flat_total = 0;
  std::vector<size_t> t_off;      // per table entry: byte offset of the transposed pack in the training arena
  std::vector<long long> tf_off;  // per table entry: byte offset of the fragment-major pack of the TRANSPOSED weight (-1: none)
  size_t t_desc_off = 0;          // byte offset of the device descriptor table inside the training arena
  size_t t_wgrad_off = 0;         // byte offset of the split-K workspace
  size_t t_bytes = 0;
  int n_sync_blocks = 0;
  char* tarena = nullptr;
  bool ready = false;
};

struct SyncDesc {
  long long src_off;    // floats into the flat parameter buffer
  long long dst_off;    // bytes into the forward arena
  long long tdst_off;   // bytes into the training arena (-1: no transposed pack)
  long long fdst_off;   // bytes into the forward arena of the fragment-major bf16 pack (-1: none)
  long long tfdst_off;  // bytes into the training arena of the fragment-major bf16 pack of the transposed weight (-1: none)
  int packed, n, cin, ntaps, Np, Cp, tRp, tKp;
  int qkv_heads;        // > 0: the fragment pack at fdst_off is the per-head [q | k | v] layout
  int n_st, cin_st;     // stored extents: output rows zero-padded to n_st, input channels regrouped to cin_st (DRN x8 level 0)
  int grp_real, grp_pad;
  long long numel;
  unsigned blk0, nblk;
  int tiled, tile_nb;   // bf16 Linear layers: 64 x 64 source tiles through LDS (tile_nb = tiles along n), see sync_linear_tile
};

namespace {

constexpr int SYNC_EPB = 2048;   // elements per block of the sync kernel

// One 64 (n) x 64 (c) tile of a bf16 Linear weight W[n][cin] to every pack that holds it: the rows are read once, coalesced,
// rounded to bf16 into LDS, and each destination - the row-major pack, the 16 x 32 fragment tiles (or the per-head q | k | v
// fragments), the transposed row-major pack and the fragments of W^T - is written in 32-byte pieces (the transposed ones from
// LDS columns).  The element-wise loops this replaces read W with a stride of `cin` floats per lane for the transposed packs
// and did index arithmetic per 2-byte store: 0.32 ms of every training step.
__device__ __forceinline__ void sync_linear_tile(const SyncDesc& d, const float* __restrict__ src, char* __restrict__ arena,
                                                 char* __restrict__ tarena, const int tb) {
  constexpr int LD = 72;                                         // LDS row stride (elements): 144 B, 16-byte aligned rows
  __shared__ __attribute__((aligned(16))) __bf16 tile[64 * LD];
  const int t = threadIdx.x, r = t >> 2, seg = t & 3;
  const int bn = tb % d.tile_nb, bc = tb / d.tile_nb;
  const int n0 = bn * 64, c0 = bc * 64;
  {
    const int n = n0 + r, c = c0 + seg * 16;
    const float* row = src + (size_t)min(n, d.n - 1) * d.cin;
    f32x4 v[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = *reinterpret_cast<const f32x4*>(row + min(c + 4 * i, d.cin - 4));     // cin % 4 == 0
    bf16x8 h[2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int e = 0; e < 4; ++e) h[i >> 1][(i & 1) * 4 + e] = (__bf16)((n < d.n && c + 4 * i < d.cin) ? v[i][e] : 0.f);
    *reinterpret_cast<bf16x8*>(tile + r * LD + seg * 16) = h[0];
    *reinterpret_cast<bf16x8*>(tile + r * LD + seg * 16 + 8) = h[1];
  }
  __syncthreads();
  {  // ---- row pieces: 16 consecutive c of row n ----
    const int n = n0 + r, c = c0 + seg * 16;
    const bf16x8 a = *reinterpret_cast<const bf16x8*>(tile + r * LD + seg * 16), b = *reinterpret_cast<const bf16x8*>(tile + r * LD + seg * 16 + 8);
    if (n < d.Np && c < d.Cp) {
      __bf16* dst = reinterpret_cast<__bf16*>(arena + d.dst_off) + (size_t)n * d.Cp + c;
      *reinterpret_cast<bf16x8*>(dst) = a; *reinterpret_cast<bf16x8*>(dst + 8) = b;
      const int ktiles = d.Cp / 32;
      if (d.fdst_off >= 0 && d.qkv_heads == 0) {
        __bf16* f = reinterpret_cast<__bf16*>(arena + d.fdst_off) + ((size_t)(n >> 4) * ktiles + (c >> 5)) * 512 + (n & 15) * 32 + (c & 31);
        *reinterpret_cast<bf16x8*>(f) = a; *reinterpret_cast<bf16x8*>(f + 8) = b;
      } else if (d.fdst_off >= 0 && n < d.n) {                     // per-head [q | k | v] fragments; their padding rows stay zero
        const int dd = d.cin, hd = dd / d.qkv_heads, HDP = (hd + 15) / 16 * 16, rtiles = 3 * HDP / 16;
        const int which = n / dd, nn = n - which * dd, hh = nn / hd, cc = nn - hh * hd, vr = which * HDP + cc;
        __bf16* f = reinterpret_cast<__bf16*>(arena + d.fdst_off) + ((size_t)(hh * rtiles + (vr >> 4)) * ktiles + (c >> 5)) * 512 + (vr & 15) * 32 + (c & 31);
        *reinterpret_cast<bf16x8*>(f) = a; *reinterpret_cast<bf16x8*>(f + 8) = b;
      }
    }
  }
  if (d.tdst_off >= 0) {  // ---- column pieces: 16 consecutive n of column c ----
    const int c = c0 + r, n = n0 + seg * 16;
    if (c < d.tRp && n < d.tKp) {
      bf16x8 a, b;
#pragma unroll
      for (int i = 0; i < 8; ++i) { a[i] = tile[(seg * 16 + i) * LD + r]; b[i] = tile[(seg * 16 + 8 + i) * LD + r]; }
      __bf16* dst = reinterpret_cast<__bf16*>(tarena + d.tdst_off) + (size_t)c * d.tKp + n;
      *reinterpret_cast<bf16x8*>(dst) = a; *reinterpret_cast<bf16x8*>(dst + 8) = b;
      if (d.tfdst_off >= 0) {
        const int ktiles = d.tKp / 32;
        __bf16* f = reinterpret_cast<__bf16*>(tarena + d.tfdst_off) + ((size_t)(c >> 4) * ktiles + (n >> 5)) * 512 + (c & 15) * 32 + (n & 31);
        *reinterpret_cast<bf16x8*>(f) = a; *reinterpret_cast<bf16x8*>(f + 8) = b;
      }
    }
  }
}

template <int PREC>
__global__ __launch_bounds__(256) void sync_params_kernel(const SyncDesc* __restrict__ descs, int ndesc,
                                                          const float* __restrict__ flat, char* __restrict__ arena,
                                                          char* __restrict__ tarena) {
  using T = typename std::conditional<PREC == SRAD_PREC_BF16, __bf16, float>::type;
  // binary search: last descriptor with blk0 <= blockIdx.x
  int lo = 0, hi = ndesc - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (descs[mid].blk0 <= blockIdx.x) lo = mid; else hi = mid - 1;
  }
  const SyncDesc d = descs[lo];
  const long long base = (long long)(blockIdx.x - d.blk0) * SYNC_EPB;
  const float* src = flat + d.src_off;
  if constexpr (PREC == SRAD_PREC_BF16) {
    if (d.tiled) { sync_linear_tile(d, src, arena, tarena, (int)(blockIdx.x - d.blk0)); return; }
  }
  if (!d.packed) {
    float* dst = reinterpret_cast<float*>(arena + d.dst_off);
    for (long long i = base + threadIdx.x; i < base + SYNC_EPB && i < d.numel; i += 256) dst[i] = src[i];
    return;
  }
  const long long ftotal = (long long)d.Np * d.ntaps * d.Cp;
  T* dst = reinterpret_cast<T*>(arena + d.dst_off);
  for (long long i = base + threadIdx.x; i < base + SYNC_EPB && i < ftotal; i += 256) {
    const int c = (int)(i % d.Cp);
    const int tap = (int)((i / d.Cp) % d.ntaps);
    const int n = (int)(i / ((long long)d.Cp * d.ntaps));
    const int cr = srad_real_channel(c, d.grp_real, d.grp_pad, d.cin);
    float v = 0.f;
    if (n < d.n && cr < d.cin) v = src[((long long)n * d.cin + cr) * d.ntaps + tap];
    dst[i] = (T)v;
  }
  if (d.fdst_off >= 0 && d.qkv_heads > 0) {   // per-head [q | k | v] fragments (srad_launch_pack_qkv_frag)
    __bf16* fdst = reinterpret_cast<__bf16*>(arena + d.fdst_off);
    const int dd = d.cin, hd = dd / d.qkv_heads, HDP = (hd + 15) / 16 * 16;
    const long long ptotal = (long long)d.qkv_heads * 3 * HDP * d.Cp;
    const int ktiles = d.Cp / 32, rtiles = 3 * HDP / 16;
    for (long long i = base + threadIdx.x; i < base + SYNC_EPB && i < ptotal; i += 256) {
      const int kin = (int)(i & 31), rin = (int)((i >> 5) & 15);
      const long long tile = i >> 9;
      const int kt = (int)(tile % ktiles), rg = (int)(tile / ktiles);
      const int h = rg / rtiles, vr = (rg - h * rtiles) * 16 + rin;
      const int which = vr / HDP, c = vr - which * HDP, k = kt * 32 + kin;
      fdst[i] = (__bf16)((c < hd && k < dd) ? src[(long long)(which * dd + h * hd + c) * dd + k] : 0.f);
    }
  } else if (d.fdst_off >= 0) {               // 16 x 32 tiles, tile-major (srad_launch_pack_weight_frag)
    __bf16* fdst = reinterpret_cast<__bf16*>(arena + d.fdst_off);
    const long long ptotal = (long long)d.Np * d.Cp;
    const int ktiles = d.Cp / 32;
    for (long long i = base + threadIdx.x; i < base + SYNC_EPB && i < ptotal; i += 256) {
      const int kin = (int)(i & 31), rin = (int)((i >> 5) & 15);
      const long long tile = i >> 9;
      const int kt = (int)(tile % ktiles), nt = (int)(tile / ktiles);
      const int n = nt * 16 + rin, k = kt * 32 + kin;
      fdst[i] = (__bf16)((n < d.n && k < d.cin) ? src[(long long)n * d.cin + k] : 0.f);
    }
  }
  if (d.tfdst_off >= 0) {                     // the same tiles of W^T: rows = input channels, k = output channels
    __bf16* fdst = reinterpret_cast<__bf16*>(tarena + d.tfdst_off);
    const long long ptotal = (long long)d.tRp * d.tKp;
    const int ktiles = d.tKp / 32;
    for (long long i = base + threadIdx.x; i < base + SYNC_EPB && i < ptotal; i += 256) {
      const int kin = (int)(i & 31), rin = (int)((i >> 5) & 15);
      const long long tile = i >> 9;
      const int kt = (int)(tile % ktiles), nt = (int)(tile / ktiles);
      const int c = nt * 16 + rin, n = kt * 32 + kin;
      fdst[i] = (__bf16)((n < d.n && c < d.cin) ? src[(long long)n * d.cin + c] : 0.f);
    }
  }
  if (d.tdst_off >= 0) {
    const long long ttotal = (long long)d.tRp * d.ntaps * d.tKp;
    T* tdst = reinterpret_cast<T*>(tarena + d.tdst_off);
    for (long long i = base + threadIdx.x; i < base + SYNC_EPB && i < ttotal; i += 256) {
      const int n = (int)(i % d.tKp);
      const int tap = (int)((i / d.tKp) % d.ntaps);
      const int c = (int)(i / ((long long)d.tKp * d.ntaps));
      const int cr = srad_real_channel(c, d.grp_real, d.grp_pad, d.cin);
      float v = 0.f;
      if (n < d.n && cr < d.cin) v = src[((long long)n * d.cin + cr) * d.ntaps + (d.ntaps - 1 - tap)];
      tdst[i] = (T)v;
    }
  }
}

// extents a packed layer is stored with (engine.h add_layer_padded): output rows padded to n_pad, input channels regrouped
inline int stored_n(const ParamEntry& e) { return e.n_pad > 0 ? e.n_pad : e.n; }
inline int stored_cin(const ParamEntry& e) { return e.grp_pad > 0 ? (e.cin / e.grp_real) * e.grp_pad : e.cin; }

inline int train_param_floats(const ParamTable& pt, TrainState& ts, int64_t* total) {
  if (ts.flat_off.empty()) {
    int64_t off = 0;
    for (const ParamEntry& e : pt.entries) {
      ts.flat_off.push_back(off);
      off += (e.numel + 63) / 64 * 64;
    }
    ts.flat_total = off;
  }
  *total = ts.flat_total;
  return SRAD_OK;
}

inline int train_arena_bytes(const ParamTable& pt, TrainState& ts, size_t* bytes) {
  SRAD_REQUIRE(pt.prec != SRAD_PREC_BF16X3, "training runs in fp32 or bf16: the split-bf16 precision is an inference mode");
  int64_t tot = 0;
  SRAD_TRY(train_param_floats(pt, ts, &tot));
  if (ts.t_off.empty()) {
    size_t off = 0;
    for (const ParamEntry& e : pt.entries) {
      ts.t_off.push_back(off);
      if (e.packed) off += srad_align_up(srad_packed_bytes(pt.prec, srad_round_up(stored_cin(e), 4), srad_round_up(stored_n(e), 4), e.ntaps), 256);
    }
    for (const ParamEntry& e : pt.entries) {      // Linear layers of the fused kernels: W^T as MFMA fragments too
      const bool frag = e.packed && (e.frag_off >= 0 || e.tfrag) && e.ntaps == 1;
      ts.tf_off.push_back(frag ? (long long)off : -1);
      if (frag) off += srad_align_up(srad_packed_bytes(SRAD_PREC_BF16, srad_round_up(e.cin, 4), srad_round_up(e.n, 4), 1), 256);
    }
    ts.t_desc_off = off;
    off += srad_align_up(pt.entries.size() * sizeof(SyncDesc), 256);
    ts.t_wgrad_off = off;                       // split-K workspace of the weight-gradient kernel
    off += srad_align_up(SRAD_WGRAD_WS_BYTES, 256);
    ts.t_bytes = off;
  }
  *bytes = ts.t_bytes;
  return SRAD_OK;
}

// Binds the caller-owned training arena.  Synchronous (one small host-to-device copy).
inline int train_bind(const ParamTable& pt, TrainState& ts, void* train_arena, size_t bytes) {
  SRAD_REQUIRE(train_arena, "train_bind: null argument");
  size_t need = 0;
  SRAD_TRY(train_arena_bytes(pt, ts, &need));
  SRAD_REQUIRE(bytes >= need, "train_bind: %zu bytes given, %zu needed", bytes, need);
  SRAD_REQUIRE(((uintptr_t)train_arena & 255) == 0, "train_bind: arena must be 256-byte aligned");
  if (!pt.arena) return srad_set_error(SRAD_ERR_STATE, "train_bind: bind the forward arena first");
  std::vector<SyncDesc> descs;
  unsigned blk = 0;
  for (size_t i = 0; i < pt.entries.size(); ++i) {
    const ParamEntry& e = pt.entries[i];
    SyncDesc d{};
    d.src_off = ts.flat_off[i]; d.dst_off = (long long)e.off; d.tdst_off = -1; d.fdst_off = e.frag_off; d.tfdst_off = -1;
    d.packed = e.packed; d.numel = e.numel;
    long long work = e.numel;
    if (e.packed) {
      // zero-padded layers (DRN x8, n_feats = 10): the packs hold the stored extents, the flat buffers the real tensor
      d.n = e.n; d.cin = e.cin; d.ntaps = e.ntaps; d.n_st = stored_n(e); d.cin_st = stored_cin(e);
      d.grp_real = e.grp_real; d.grp_pad = e.grp_pad;
      SRAD_REQUIRE((d.n_st == d.n && d.cin_st == d.cin) || (e.frag_off < 0 && !e.tfrag), "train_bind: a padded layer cannot carry fragment packs");
      d.Np = srad_np(d.n_st); d.Cp = srad_cp(d.cin_st);
      d.tRp = srad_np(srad_round_up(d.cin_st, 4)); d.tKp = srad_cp(srad_round_up(d.n_st, 4));
      d.tdst_off = (long long)ts.t_off[i];
      d.tfdst_off = ts.tf_off[i];
      d.qkv_heads = e.qkv_heads;
      const long long f = (long long)d.Np * d.ntaps * d.Cp, t = (long long)d.tRp * d.ntaps * d.tKp;
      work = f > t ? f : t;
      if (e.qkv_heads > 0) {                   // the per-head pack can be longer than the plain one (head slices padded to 16)
        const long long q = (long long)(srad_qkv_frag_bytes(e.cin, e.qkv_heads) / 2);
        if (q > work) work = q;
      }
    }
    d.blk0 = blk;
    d.nblk = (unsigned)((work + SYNC_EPB - 1) / SYNC_EPB);
    if (d.nblk == 0) d.nblk = 1;
    // bf16 Linear layers go tile by tile (sync_linear_tile); the per-head q | k | v pack's padding rows are not rewritten
    // there - they are zero since the arena was packed at creation and nothing else touches them
    if (e.packed && pt.prec == SRAD_PREC_BF16 && e.ntaps == 1 && e.grp_pad == 0 && e.cin % 4 == 0 && e.cin >= 4 && getenv("SRAD_SYNC_ELEMENTWISE") == nullptr) {
      const int ext_n = d.Np > d.tKp ? d.Np : d.tKp, ext_c = d.Cp > d.tRp ? d.Cp : d.tRp;
      d.tiled = 1; d.tile_nb = (ext_n + 63) / 64;
      d.nblk = (unsigned)(d.tile_nb * ((ext_c + 63) / 64));
    }
    blk += d.nblk;
    descs.push_back(d);
  }
  ts.n_sync_blocks = (int)blk;
  ts.tarena = reinterpret_cast<char*>(train_arena);
  SRAD_CHECK_HIP(hipMemcpy(ts.tarena + ts.t_desc_off, descs.data(), descs.size() * sizeof(SyncDesc), hipMemcpyHostToDevice));
  ts.ready = false;
  return SRAD_OK;
}

// Refreshes every packed weight (forward and transposed) and raw parameter from the flat fp32 master buffer: one launch.
inline int train_sync_params(const ParamTable& pt, TrainState& ts, const float* flat_params, hipStream_t s) {
  SRAD_REQUIRE(flat_params, "sync_params: null argument");
  if (!ts.tarena) return srad_set_error(SRAD_ERR_STATE, "sync_params: no training arena bound");
  const SyncDesc* descs = reinterpret_cast<const SyncDesc*>(ts.tarena + ts.t_desc_off);
  SradProfScope prof(s, SRAD_K_PACK, 0.0, 4.0 * ts.flat_total + 2.0 * (pt.bytes + ts.t_bytes));
  if (pt.prec == SRAD_PREC_BF16)
    hipLaunchKernelGGL((sync_params_kernel<SRAD_PREC_BF16>), dim3(ts.n_sync_blocks), dim3(256), 0, s, descs,
                       (int)pt.entries.size(), flat_params, pt.arena, ts.tarena);
  else
    hipLaunchKernelGGL((sync_params_kernel<SRAD_PREC_F32>), dim3(ts.n_sync_blocks), dim3(256), 0, s, descs,
                       (int)pt.entries.size(), flat_params, pt.arena, ts.tarena);
  SRAD_CHECK_HIP(hipGetLastError());
  ts.ready = true;
  return SRAD_OK;
}

inline WgradQueue train_wgrad_queue(const TrainState& ts) {
  WgradQueue q;
  q.ws = reinterpret_cast<float*>(ts.tarena + ts.t_wgrad_off);
  q.ws_floats = SRAD_WGRAD_WS_BYTES / sizeof(float);
  return q;
}

}  // namespace
