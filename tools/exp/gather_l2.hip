// Rate of independent random 16-byte gathers served by the L2s: what bounds k_place_grid's record look-ups (refdata: 368 M
// look-ups per 10 000 samples, one 16-byte record each out of 4.5 MB of records).  Every wave issues eight independent gathers
// per trip (a chunk of k_place), waits for them, goes on.  Variants: one table for the whole chip (every XCD's L2 holds all
// of it) against one table per XCD (blockIdx % 8: the stride with which workgroups go round the XCDs) of an eighth of the size;
// 8 or 16 waves per CU.
// build: hipcc --offload-arch=gfx950 -O3 -o build/gather_l2 tools/exp/gather_l2.hip ; run on the GPU box
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
__global__ void init(uint4* buf, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    buf[i] = make_uint4((unsigned)i, (unsigned)(i * 2654435761ull >> 7), 1u, 2u);
}
__global__ void k(const uint4* __restrict__ buf, unsigned table_el, int per_xcd, int trips, unsigned* out) {
  const unsigned tid = blockIdx.x * blockDim.x + threadIdx.x;
  const uint4* __restrict__ t = buf + (per_xcd ? (size_t)(blockIdx.x & 7) * table_el : 0);
  unsigned x = tid * 2654435761u + 12345u, acc = 0;
  for (int i = 0; i < trips; ++i) {
    unsigned idx[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) { x = x * 1664525u + 1013904223u; idx[c] = (unsigned)(((unsigned long long)(x >> 4) * table_el) >> 28); }
    uint4 v[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) v[c] = t[idx[c]];
#pragma unroll
    for (int c = 0; c < 8; ++c) acc += v[c].x ^ v[c].y;
  }
  if (acc == 0x12345u) out[0] = acc;
}
int main() {
  const size_t total_bytes = (size_t)64 << 20;
  uint4* d; if (hipMalloc(&d, total_bytes) != hipSuccess) { printf("alloc failed\n"); return 1; }
  init<<<1024, 256>>>(d, total_bytes / 16);
  unsigned* out; hipMalloc(&out, 64);
  hipDeviceSynchronize();
  const int trips = 400;
  for (int wg_per_cu : {1, 2}) {
    for (size_t table_bytes : {(size_t)512 << 10, (size_t)1 << 20, (size_t)2 << 20, (size_t)4608 << 10, (size_t)9 << 20, (size_t)32 << 20}) {
      for (int per_xcd : {0, 1}) {
        const unsigned el = (unsigned)((per_xcd ? table_bytes / 8 : table_bytes) / 16);
        const int blocks = 256 * wg_per_cu;
        float best = 1e9;
        for (int rep = 0; rep < 3; ++rep) {
          hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
          hipEventRecord(e0, 0);
          hipLaunchKernelGGL(k, dim3(blocks), dim3(512), 0, 0, d, el, per_xcd, trips, out);
          hipEventRecord(e1, 0);
          hipDeviceSynchronize();
          float ms = 0; hipEventElapsedTime(&ms, e0, e1);
          best = ms < best ? ms : best;
        }
        const double req = (double)blocks * 512 * trips * 8;
        printf("%2d waves per CU, records %5zu KB in all, %s: %.3f ms, %.0f G gathers/s\n", 8 * wg_per_cu, table_bytes >> 10,
               per_xcd ? "an eighth per XCD" : "one table         ", best, req / best / 1e6);
      }
    }
  }
  return 0;
}
