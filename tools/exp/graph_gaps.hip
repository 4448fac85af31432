// What a dependent kernel boundary costs on one stream, launched one by one against replayed from a captured hipGraph:
// chains of N short kernels (each ~T us of spinning on s_memtime-free arithmetic), wall time of the chain by events.
// build: hipcc --offload-arch=gfx950 -O3 -o build/graph_gaps tools/exp/graph_gaps.hip ; run on the GPU box
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void spin(float* p, int iters) {
  float x = p[threadIdx.x & 63];
  for (int i = 0; i < iters; ++i) x = x * 1.0000001f + 0.5f;
  if (x == 12345.f) p[0] = x;
}
int main() {
  float* d; hipMalloc(&d, 4096); hipMemset(d, 0, 4096);
  hipStream_t st; hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int N = 12;
  for (int blocks : {256, 4096}) {
    for (int iters : {2000, 40000}) {
      auto chain = [&]() { for (int k = 0; k < N; ++k) hipLaunchKernelGGL(spin, dim3(blocks), dim3(64), 0, st, d, iters); };
      // one kernel alone
      float one = 1e9, direct = 1e9, graph = 1e9;
      for (int rep = 0; rep < 5; ++rep) {
        hipEventRecord(e0, st); hipLaunchKernelGGL(spin, dim3(blocks), dim3(64), 0, st, d, iters); hipEventRecord(e1, st);
        hipStreamSynchronize(st); float ms; hipEventElapsedTime(&ms, e0, e1); one = ms < one ? ms : one;
      }
      for (int rep = 0; rep < 5; ++rep) {
        hipEventRecord(e0, st); chain(); hipEventRecord(e1, st);
        hipStreamSynchronize(st); float ms; hipEventElapsedTime(&ms, e0, e1); direct = ms < direct ? ms : direct;
      }
      hipGraph_t g; hipGraphExec_t ge;
      hipStreamBeginCapture(st, hipStreamCaptureModeGlobal); chain(); hipStreamEndCapture(st, &g);
      hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
      for (int rep = 0; rep < 5; ++rep) {
        hipEventRecord(e0, st); hipGraphLaunch(ge, st); hipEventRecord(e1, st);
        hipStreamSynchronize(st); float ms; hipEventElapsedTime(&ms, e0, e1); graph = ms < graph ? ms : graph;
      }
      printf("%5d blocks x %6d iterations: one kernel %.1f us; %d in a row %.1f us (%.1f us per boundary); as a graph %.1f us (%.1f us per boundary)\n",
             blocks, iters, one * 1e3, N, direct * 1e3, (direct - N * one) * 1e3 / (N - 1), graph * 1e3, (graph - N * one) * 1e3 / (N - 1));
      hipGraphExecDestroy(ge); hipGraphDestroy(g);
    }
  }
  return 0;
}
