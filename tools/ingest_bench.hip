// ingest_bench.hip - how fast can a CU pull a weight set that EVERY workgroup of the launch reads (the access pattern of
// mlp_block / qkv_attn: 1 KB per wave instruction, 16 B per lane, fragment-major packs served by the XCD's L2)?
// Build + run on the GPU box:  hipcc -O3 --offload-arch=gfx950 tools/ingest_bench.hip -o /tmp/ingest && /tmp/ingest
// Prints microseconds per launch and GB/s per workgroup for grids of 64 .. 512 workgroups and several weight-set sizes.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// each of the 8 waves owns every 8th kilobyte of the set, DEPTH loads in flight
template <int DEPTH>
__global__ __launch_bounds__(512) void stream_kernel(const char* __restrict__ w, int kb_total, unsigned* __restrict__ sink) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  u32x4 acc = {0, 0, 0, 0};
  u32x4 r[DEPTH];
  const char* base = w + lane * 16;
  int kb = wave;
#pragma unroll
  for (int i = 0; i < DEPTH; ++i) { r[i] = *reinterpret_cast<const u32x4*>(base + (size_t)min(kb, kb_total - 1) * 1024); kb += 8; }
  for (; kb - 8 * DEPTH < kb_total; ) {
#pragma unroll
    for (int i = 0; i < DEPTH; ++i) {
      acc ^= r[i];
      r[i] = *reinterpret_cast<const u32x4*>(base + (size_t)min(kb, kb_total - 1) * 1024);
      kb += 8;
    }
  }
  if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345u) sink[blockIdx.x] = acc[0];
}

int main() {
  const size_t max_bytes = 4u << 20;
  char* w;
  unsigned* sink;
  hipMalloc(&w, max_bytes);
  hipMalloc(&sink, 4096);
  hipMemset(w, 1, max_bytes);
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  const int sizes_kb[] = {160, 336, 590, 1250, 2500};
  const int grids[] = {32, 64, 128, 256, 512};
  for (int depth : {8, 24}) {
    for (int kb : sizes_kb) {
      for (int g : grids) {
        auto launch = [&]() {
          if (depth == 8) hipLaunchKernelGGL(stream_kernel<8>, dim3(g), dim3(512), 0, 0, w, kb, sink);
          else hipLaunchKernelGGL(stream_kernel<24>, dim3(g), dim3(512), 0, 0, w, kb, sink);
        };
        for (int i = 0; i < 5; ++i) launch();
        hipEventRecord(a, 0);
        const int iters = 50;
        for (int i = 0; i < iters; ++i) launch();
        hipEventRecord(b, 0);
        hipEventSynchronize(b);
        float ms;
        hipEventElapsedTime(&ms, a, b);
        const double us = ms * 1e3 / iters;
        printf("depth %2d KB in flight/wave  set %5d KB  grid %3d: %7.2f us/launch  %6.1f GB/s per WG  %6.2f TB/s chip\n", depth, kb, g, us,
               kb * 1024.0 / (us * 1e-6) / 1e9, (double)g * kb * 1024.0 / (us * 1e-6) / 1e12);
      }
    }
  }
  return 0;
}
