// Micro-benchmark (MI355X): k_place's written-out step (GAT_STEP_SIMPLE_ASM, gat_kernels.h) alone -- no row loads, no flush --
// with 1..4 waves per SIMD: does the step's instruction stream overlap between the waves of a SIMD?  Variants: the step as it
// is; with the chunk's eight rank look-ups (random LDS reads) in front; without the ring store.
// Build: hipcc --offload-arch=gfx950 -O2 -o build/place_step tools/ubench/place_step.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

#define STEP(Y, LR1, JJ1, STORE)                                                                               \
  {                                                                                                            \
    uint32_t t0_, t1_, t2_, t3_;                                                                               \
    uint64_t sa_, sb_, sc_;                                                                                    \
    asm volatile(                                                                                              \
        "v_and_b32 %7, %16, %14\n\t"                                                                           \
        "v_and_b32 %8, %26, %14\n\t"                                                                           \
        "v_cmp_ge_u32_e64 %12, %17, %7\n\t"                                                                    \
        "v_add_u32 %9, %19, %0\n\t"                                                                            \
        "v_cmp_ge_u32 vcc, %18, %8\n\t"                                                                        \
        "v_cmp_le_u32_e64 %13, %7, %9\n\t"                                                                     \
        "s_and_b64 %11, %4, vcc\n\t"                                                                           \
        "v_cmp_le_i32 vcc, %1, %15\n\t"                                                                        \
        "s_and_b64 %12, %5, %12\n\t"                                                                           \
        "s_and_b64 %13, %6, %13\n\t"                                                                           \
        "s_and_b64 vcc, %11, vcc\n\t"                                                                          \
        "v_cndmask_b32_e64 %0, %0, %15, %11\n\t"                                                               \
        "v_cndmask_b32_e64 %3, %3, %25, vcc\n\t"                                                               \
        "s_xor_b64 %4, %4, %11\n\t"                                                                            \
        "s_andn2_b64 %11, %11, vcc\n\t"                                                                        \
        "s_xor_b64 %5, %5, %12\n\t"                                                                            \
        "s_xor_b64 %6, %6, %13\n\t"                                                                            \
        "s_or_b64 %4, %4, %13\n\t"                                                                             \
        "s_or_b64 %5, %5, %11\n\t"                                                                             \
        "s_or_b64 %6, %6, %12\n\t"                                                                             \
        "s_and_saveexec_b64 %11, %13\n\t"                                                                      \
        "v_add_u32 %9, %20, %7\n\t"                                                                            \
        "v_and_or_b32 %8, %2, %23, %24\n\t"                                                                    \
        "v_sub_u32 %7, %9, %0\n\t"                                                                             \
        "v_add_u32 %2, 0x200, %2\n\t"                                                                          \
        "v_min_i32 %10, %22, %9\n\t"                                                                           \
        "v_max_i32 %7, 0, %7\n\t"                                                                              \
        "v_sub_u32 %1, %1, %10\n\t"                                                                            \
        "v_max_i32 %10, %21, %7\n\t"                                                                           \
        STORE                                                                                                  \
        "v_mov_b32 %3, %25\n\t"                                                                                \
        "v_add_u32 %1, %1, %10\n\t"                                                                            \
        "s_mov_b64 exec, %11"                                                                                  \
        : "+v"(len), "+v"(rem), "+v"(nS9), "+v"(used_lo), "+s"(mL), "+s"(mP), "+s"(mO),                        \
          "=&v"(t0_), "=&v"(t1_), "=&v"(t2_), "=&v"(t3_), "=&s"(sa_), "=&s"(sb_), "=&s"(sc_)                   \
        : "v"(Y), "v"(LR1), "s"(maskP), "s"(rangeP), "s"(rangeL), "s"(c_r3), "s"(c_ss), "s"(wsx), "s"(wsy),   \
          "s"(0x1e00u), "v"(lane_ring), "n"(JJ1), "s"(maskL)                                                   \
        : "vcc", "scc", "memory");                                                                             \
  }
#define ST_ON "ds_write2_b32 %8, %7, %9 offset1:1\n\t"
#define ST_OFF ""

// VARIANT 0: the step; 1: + look-ups; 2: the step without its ring store; 3: look-ups + no store
template <int VARIANT>
__global__ void __launch_bounds__(64) k(unsigned* out, unsigned long long* cyc, int iters, unsigned seed) {
  __shared__ __attribute__((aligned(8192))) uint2 ring[16][64];
  __shared__ unsigned tab[VARIANT & 1 ? 1024 : 1];
  const int lane = threadIdx.x;
  if (VARIANT & 1) for (int i = lane; i < 1024; i += 64) tab[i] = 100 + (i * 7) % 900;
  __syncthreads();
  const uint32_t lane_ring = (uint32_t)(size_t)(__attribute__((address_space(3))) void*)&ring[0][lane];
  const uint32_t maskP = 0x0fffffffu, rangeP = 0x0c000001u, maskL = 1023u, rangeL = 785u, c_r3 = 0x0c000000u, wsx = 0u, wsy = c_r3 + 2u;
  const int32_t c_ss = 1;
  uint32_t len = 0, nS9 = 0, used_lo = 0;
  int32_t rem = 0x7fffffff;
  uint64_t mL = ~0ull, mP = 0, mO = 0;
  uint32_t y = seed + blockIdx.x * 977u + lane * 2654435761u;
  unsigned long long t0, t1;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
  for (int i = 0; i < iters; ++i) {
    uint32_t ya[8], lr[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) { y = y * 1664525u + 1013904223u; ya[c] = y ^ (y >> 15); }
#pragma unroll
    for (int c = 0; c < 8; ++c) lr[c] = (VARIANT & 1) ? tab[ya[c] & maskL] : 100u + (ya[c] & 511u);
    if (VARIANT & 2) {
      STEP(ya[0], lr[0], 1, ST_OFF) STEP(ya[1], lr[1], 2, ST_OFF) STEP(ya[2], lr[2], 3, ST_OFF) STEP(ya[3], lr[3], 4, ST_OFF)
      STEP(ya[4], lr[4], 5, ST_OFF) STEP(ya[5], lr[5], 6, ST_OFF) STEP(ya[6], lr[6], 7, ST_OFF) STEP(ya[7], lr[7], 8, ST_OFF)
    } else {
      STEP(ya[0], lr[0], 1, ST_ON) STEP(ya[1], lr[1], 2, ST_ON) STEP(ya[2], lr[2], 3, ST_ON) STEP(ya[3], lr[3], 4, ST_ON)
      STEP(ya[4], lr[4], 5, ST_ON) STEP(ya[5], lr[5], 6, ST_ON) STEP(ya[6], lr[6], 7, ST_ON) STEP(ya[7], lr[7], 8, ST_ON)
    }
  }
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
  out[blockIdx.x * 64 + lane] = len + rem + nS9 + used_lo + (unsigned)(mL + mP + mO) + ring[3][lane].x;
  if (lane == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int VARIANT>
static void run(const char* name, int waves_per_simd, unsigned* out, unsigned long long* cyc) {
  const int iters = 1000, blocks = 1024 * waves_per_simd;           // one-wave workgroups: the dispatcher spreads them over the SIMDs
  hipEvent_t e0, e1;
  CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
  k<VARIANT><<<blocks, 64>>>(out, cyc, 10, 1u);
  CHK(hipDeviceSynchronize());
  CHK(hipEventRecord(e0));
  k<VARIANT><<<blocks, 64>>>(out, cyc, iters, 2u);
  CHK(hipEventRecord(e1));
  CHK(hipDeviceSynchronize());
  float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
  std::vector<unsigned long long> h(blocks);
  CHK(hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost));
  double sum = 0; for (auto v : h) sum += (double)v;
  const double rows = 8.0 * iters;
  printf("%-30s waves/SIMD %d: %7.3f ms, cycles per row per wave %6.1f, ns per row per wave %6.1f, per SIMD %6.1f\n", name, waves_per_simd, ms,
         sum / h.size() / rows, ms * 1e6 / rows, ms * 1e6 / rows / waves_per_simd);
}

int main() {
  setvbuf(stdout, nullptr, _IONBF, 0);
  unsigned* out; unsigned long long* cyc;
  CHK(hipMalloc(&out, 1024 * 4 * 64 * 4)); CHK(hipMalloc(&cyc, 1024 * 4 * 8));
  for (int w : {1, 2, 3, 4}) {
    run<0>("step", w, out, cyc);
    run<2>("step, no ring store", w, out, cyc);
    if (w <= 3) { run<1>("step + look-ups", w, out, cyc); run<3>("look-ups, no ring store", w, out, cyc); }
  }
  return 0;
}
