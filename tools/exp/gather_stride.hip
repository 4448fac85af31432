// Latency of a wave's dependent gathers as a function of the distance between its lanes' addresses: 64 lanes, each walking
// its own chain of dependent 8-byte loads inside a region of 420 elements (a merged list of config 2) that starts
// lane * stride bytes behind the wave's base; the caches are flushed (a 1 GB buffer is rewritten) before every timed launch.
// build: hipcc --offload-arch=gfx950 -O3 -o build/gather_stride tools/exp/gather_stride.hip ; run on the GPU box
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
__global__ void init(uint2* buf, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    buf[i] = make_uint2((unsigned)(i * 2654435761ull >> 11), (unsigned)i);
}
__global__ void flush(uint4* buf, size_t n, unsigned v) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) buf[i] = make_uint4(v, v, v, v);
}
__global__ void k(const uint2* __restrict__ buf, long stride_el, long wave_el, int span_el, int n, unsigned* out) {
  const int lane = threadIdx.x & 63;
  const uint2* p = buf + (long)blockIdx.x * wave_el + lane * stride_el;
  unsigned idx = (lane * 7u + blockIdx.x) % span_el;
  unsigned acc = 0;
  for (int i = 0; i < n; ++i) {
    const uint2 v = p[idx];
    acc ^= v.y;
    idx = (v.x ^ acc) % span_el;                               // the next index depends on what was loaded
  }
  if (acc == 0x12345u) out[0] = acc;
}
int main() {
  const int n = 256, span_el = 420;
  const size_t total = (size_t)40 << 30;
  uint2* d; if (hipMalloc(&d, total) != hipSuccess) { printf("alloc failed\n"); return 1; }
  init<<<4096, 256>>>(d, total / 8);
  uint4* fl; hipMalloc(&fl, (size_t)1 << 30);
  unsigned* out; hipMalloc(&out, 64);
  hipDeviceSynchronize();
  const long strides[] = {3360, 4096, 8192, 16384, 32768, 65536, 86016, 131072};   // bytes between the lanes' lists
  for (int waves : {256, 3840}) {
    for (long sb : strides) {
      const long stride_el = sb / 8, wave_el = stride_el * 64;
      if ((size_t)waves * wave_el * 8 > total) { printf("waves %5d stride %7ld: skipped (buffer)\n", waves, sb); continue; }
      float best = 1e9;
      for (int rep = 0; rep < 3; ++rep) {
        flush<<<2048, 256>>>(fl, ((size_t)1 << 30) / 16, (unsigned)rep);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(k, dim3(waves), dim3(64), 0, 0, d, stride_el, wave_el, span_el, n, out);
        hipEventRecord(e1, 0);
        hipDeviceSynchronize();
        float ms = 0; hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best;
      }
      printf("waves %5d, lists %7ld B apart: kernel %.1f us = %.0f ns per dependent gather\n", waves, sb, best * 1e3, best * 1e6 / n);
    }
  }
  return 0;
}
