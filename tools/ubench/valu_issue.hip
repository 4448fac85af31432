// Micro-benchmark (MI355X): what one SIMD issues per cycle for the instruction kinds of k_place's state machine, with 1..4 waves
// per SIMD.  Build: hipcc --offload-arch=gfx950 -O2 -o build/valu_issue tools/ubench/valu_issue.hip ; run on the GPU box.
// Each wave runs `iters` trips of a body of 64 instructions of one kind (independent: eight accumulators) and stamps
// s_memtime around the loop; printed: cycles per instruction per wave and per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define REP8(X) X X X X X X X X
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int KIND>
__global__ void __launch_bounds__(256) k(unsigned* out, unsigned long long* cyc, int iters) {
  unsigned a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  unsigned long long s0 = blockIdx.x, s1 = 1, s2 = 2, s3 = 3;
  unsigned long long t0, t1;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
  for (int i = 0; i < iters; ++i) {
    if constexpr (KIND == 0) {        // v_add_u32, eight independent chains
      REP8(asm volatile("v_add_u32 %0, %0, %8\n\tv_add_u32 %1, %1, %8\n\tv_add_u32 %2, %2, %8\n\tv_add_u32 %3, %3, %8\n\t"
                        "v_add_u32 %4, %4, %8\n\tv_add_u32 %5, %5, %8\n\tv_add_u32 %6, %6, %8\n\tv_add_u32 %7, %7, %8"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(a0));)
    } else if constexpr (KIND == 1) { // one dependent chain of v_add_u32
      REP8(asm volatile("v_add_u32 %0, %0, %1\n\tv_add_u32 %0, %0, %1\n\tv_add_u32 %0, %0, %1\n\tv_add_u32 %0, %0, %1\n\t"
                        "v_add_u32 %0, %0, %1\n\tv_add_u32 %0, %0, %1\n\tv_add_u32 %0, %0, %1\n\tv_add_u32 %0, %0, %1"
                        : "+v"(a0) : "v"(a1));)
    } else if constexpr (KIND == 2) { // v_cmp -> sgpr pair, v_cndmask from it (the domain crossing), independent pairs
      REP8(asm volatile("v_cmp_le_u32 %4, %0, %1\n\tv_cndmask_b32 %2, %2, %3, %4\n\tv_cmp_le_u32 %5, %1, %0\n\tv_cndmask_b32 %3, %3, %2, %5\n\t"
                        "v_cmp_le_u32 %4, %2, %1\n\tv_cndmask_b32 %0, %0, %3, %4\n\tv_cmp_le_u32 %5, %3, %0\n\tv_cndmask_b32 %1, %1, %2, %5"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+s"(s0), "+s"(s1) :: "scc");)
    } else if constexpr (KIND == 3) { // SALU only: s_and_b64 / s_or_b64 / s_add_u32, independent
      REP8(asm volatile("s_and_b64 %0, %0, %1\n\ts_or_b64 %2, %2, %3\n\ts_andn2_b64 %1, %1, %0\n\ts_xor_b64 %3, %3, %2\n\t"
                        "s_and_b64 %0, %0, %1\n\ts_or_b64 %2, %2, %3\n\ts_andn2_b64 %1, %1, %0\n\ts_xor_b64 %3, %3, %2"
                        : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3) :: "scc");)
    } else if constexpr (KIND == 4) { // VALU and SALU alternating (independent of each other)
      REP8(asm volatile("v_add_u32 %0, %0, %1\n\ts_and_b64 %4, %4, %5\n\tv_add_u32 %1, %1, %0\n\ts_or_b64 %5, %5, %4\n\t"
                        "v_add_u32 %2, %2, %3\n\ts_and_b64 %4, %4, %5\n\tv_add_u32 %3, %3, %2\n\ts_or_b64 %5, %5, %4"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+s"(s0), "+s"(s1) :: "scc");)
    } else if constexpr (KIND == 5) { // the crossing as a dependent chain: v_cmp -> s_and -> v_cndmask -> v_cmp ...
      REP8(asm volatile("v_cmp_le_u32 %2, %0, %1\n\ts_and_b64 %2, %2, %3\n\tv_cndmask_b32 %0, %0, %1, %2\n\tv_add_u32 %0, %0, %1\n\t"
                        "v_cmp_le_u32 %2, %0, %1\n\ts_and_b64 %2, %2, %3\n\tv_cndmask_b32 %0, %0, %1, %2\n\tv_add_u32 %0, %0, %1"
                        : "+v"(a0), "+v"(a1), "+s"(s0), "+s"(s1) :: "scc");)
    } else if constexpr (KIND == 6) { // 64-bit shift and clz-like ops (v_lshrrev_b64 is a quarter-rate candidate), v_ffbh
      REP8(asm volatile("v_ffbh_u32 %0, %1\n\tv_lshrrev_b32 %1, %0, %2\n\tv_and_b32 %2, %1, %3\n\tv_sub_u32 %3, %2, %0\n\t"
                        "v_ffbh_u32 %0, %1\n\tv_lshrrev_b32 %1, %0, %2\n\tv_and_b32 %2, %1, %3\n\tv_sub_u32 %3, %2, %0"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));)
    } else if constexpr (KIND == 7) { // v_mov with DPP
      REP8(asm volatile("v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %1, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                        "v_mov_b32_dpp %2, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %3, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                        "v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %1, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                        "v_mov_b32_dpp %2, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %3, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));)
    } else if constexpr (KIND == 8) { // v_add_f32 (the guide's 2-cycle case), independent
      REP8(asm volatile("v_add_f32 %0, %0, %8\n\tv_add_f32 %1, %1, %8\n\tv_add_f32 %2, %2, %8\n\tv_add_f32 %3, %3, %8\n\t"
                        "v_add_f32 %4, %4, %8\n\tv_add_f32 %5, %5, %8\n\tv_add_f32 %6, %6, %8\n\tv_add_f32 %7, %7, %8"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(a0));)
    } else if constexpr (KIND == 9) { // v_cmp into vcc + s_and_saveexec-free ballot use: v_cmp then s_cmp on the mask (k_place's loop control)
      REP8(asm volatile("v_cmp_le_u32 %2, %0, %1\n\ts_cmp_eq_u64 %2, 0\n\ts_cselect_b64 %3, %2, %3\n\tv_add_u32 %0, %0, %1\n\t"
                        "v_cmp_le_u32 %2, %0, %1\n\ts_cmp_eq_u64 %2, 0\n\ts_cselect_b64 %3, %2, %3\n\tv_add_u32 %0, %0, %1"
                        : "+v"(a0), "+v"(a1), "+s"(s0), "+s"(s1) :: "scc");)
    }
  }
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (unsigned)(s0 + s1 + s2 + s3);
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

template <int KIND>
static void run(const char* name, int waves_per_simd, unsigned* out, unsigned long long* cyc) {
  const int iters = 2000, blocks = 256 * waves_per_simd;           // one 256-thread workgroup = one wave per SIMD of a CU
  hipEvent_t e0, e1;
  CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
  k<KIND><<<blocks, 256>>>(out, cyc, 10);
  CHK(hipDeviceSynchronize());
  CHK(hipEventRecord(e0));
  k<KIND><<<blocks, 256>>>(out, cyc, iters);
  CHK(hipEventRecord(e1));
  CHK(hipDeviceSynchronize());
  float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
  std::vector<unsigned long long> h(blocks * 4);
  CHK(hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost));
  double sum = 0; for (auto v : h) sum += (double)v;
  const double n_inst = 64.0 * iters;
  // s_memtime counts at a constant 100 MHz on this chip: report wall nanoseconds per instruction as well
  printf("%-34s waves/SIMD %d: %7.3f ms, memtime ticks per wave %9.0f, ns per instruction per wave %.3f, per SIMD %.3f\n", name, waves_per_simd, ms,
         sum / h.size(), ms * 1e6 / n_inst, ms * 1e6 / n_inst / waves_per_simd);
}

int main() {
  setvbuf(stdout, nullptr, _IONBF, 0);
  printf("start\n");
  unsigned* out; unsigned long long* cyc;
  CHK(hipMalloc(&out, 256 * 8 * 256 * 4)); CHK(hipMalloc(&cyc, 256 * 8 * 4 * 8));
  for (int w : {1, 2, 3, 4, 8}) {
    run<0>("v_add_u32 independent", w, out, cyc);
    run<1>("v_add_u32 dependent chain", w, out, cyc);
    run<8>("v_add_f32 independent", w, out, cyc);
    run<2>("v_cmp->sgpr + v_cndmask", w, out, cyc);
    run<3>("s_and/or_b64 independent", w, out, cyc);
    run<4>("VALU/SALU alternating", w, out, cyc);
    run<5>("v_cmp->s_and->v_cndmask chain", w, out, cyc);
    run<6>("v_ffbh/lshr/and/sub chain", w, out, cyc);
    run<7>("v_mov_dpp", w, out, cyc);
    run<9>("v_cmp->s_cmp->s_cselect", w, out, cyc);
  }
  return 0;
}
