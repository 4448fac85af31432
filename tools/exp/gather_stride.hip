// Latency of a wave's dependent gathers as a function of the distance between its lanes' addresses: 64 lanes, each walking
// its own chain of pointer-free dependent loads (index = f(previous value)) inside a region of `span` bytes that starts
// lane * stride bytes into a buffer.  Prints cycles per dependent load for one wave per CU and for 15 waves per CU.
// build: hipcc --offload-arch=gfx950 -O3 -o build/gather_stride tools/exp/gather_stride.hip ; run on the GPU box
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
__global__ void k(const uint2* __restrict__ buf, long stride_el, int span_el, int n, unsigned long long* out, int waves) {
  const int lane = threadIdx.x & 63;
  const long wave = (long)blockIdx.x;
  const uint2* p = buf + (wave * 64 + lane) * stride_el;
  unsigned idx = (lane * 7u) % span_el;
  unsigned acc = 0;
  unsigned long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < n; ++i) {
    const uint2 v = p[idx];
    acc += v.y;
    idx = (v.x + acc * 0u + idx * 13u + 5u) % span_el;      // depends on the loaded value
  }
  unsigned long long t1 = __builtin_readcyclecounter();
  if (lane == 0) out[wave] = t1 - t0 + (acc & 0u);
}
int main() {
  const int n = 64;
  const size_t total = (size_t)3 << 30;                     // 3 GB
  uint2* d; hipMalloc(&d, total);
  std::vector<uint2> h(total / 8 / 64);
  for (size_t i = 0; i < h.size(); ++i) h[i] = make_uint2((unsigned)(i * 2654435761u >> 7), 1u);
  for (int r = 0; r < 64; ++r) hipMemcpy((char*)d + r * (total / 64), h.data(), total / 64, hipMemcpyHostToDevice);
  unsigned long long* out; hipMalloc(&out, 8 * 8192);
  std::vector<unsigned long long> ho(8192);
  const long strides[] = {512, 4096, 8192, 16384, 65536, 86016, 131072, 524288};   // bytes between lanes
  for (int waves : {256, 3840}) {
    for (long sb : strides) {
      const long stride_el = sb / 8;
      const int span_el = (int)(sb / 8 < 420 ? sb / 8 : 420);     // a list of ~420 segments
      if ((size_t)waves * 64 * sb > total) { printf("waves %5d stride %7ld: skipped (buffer)\n", waves, sb); continue; }
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      hipLaunchKernelGGL(k, dim3(waves), dim3(64), 0, 0, d, stride_el, span_el, n, out, waves);
      hipEventRecord(e0, 0);
      hipLaunchKernelGGL(k, dim3(waves), dim3(64), 0, 0, d, stride_el, span_el, n, out, waves);
      hipEventRecord(e1, 0);
      hipDeviceSynchronize();
      float ms = 0; hipEventElapsedTime(&ms, e0, e1);
      hipMemcpy(ho.data(), out, 8 * waves, hipMemcpyDeviceToHost);
      double s = 0, mx = 0; for (int i = 0; i < waves; ++i) { s += ho[i]; if (ho[i] > mx) mx = ho[i]; }
      printf("waves %5d stride %7ld B: %.0f ticks per dependent gather (mean), %.0f (slowest wave); kernel %.1f us = %.0f ns per gather\n", waves, sb, s / waves / n, mx / n, ms * 1e3, ms * 1e6 / n);
    }
  }
  return 0;
}
