// gat_stats.h -- the numbers AnnotatorResult takes from a null distribution (gat/Engine.pyx:1635-1718,
// makeEnrichmentStatistics / getTwoSidedPValue), computed where the count matrix is: per (counter, track) row of S
// sampled counts the mean, the standard deviation, the two order statistics of the 95 % interval and the numbers of
// samples below / equal to the observed value.  At the config-4 scale (1 000 tracks x 100 000 samples) the host needs
// 2 ms per row for them -- longer than the device needs for the sampling.
//
// Bit-exactness is the point.  numpy.mean / numpy.std (which the reference calls) sum a float64 array in chunks of 8 192
// elements (the ufunc buffer), each chunk by pairwise summation -- halves split at a multiple of 8 down to blocks of at
// most 128 elements, a block with 8 running sums combined ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)) and a serial remainder --
// and add the chunk sums left to right (checked against numpy 2.2.6 for lengths 1..300 and up to 250 000:
// tests/test_host_logic.py::test_numpy_summation_model).  The kernel performs exactly those additions: a full chunk is 64
// blocks of 128 (one per lane, then a butterfly in the recursion's order); the last, partial chunk follows a table the
// host derives from its length.  Every operation is an explicit round-to-nearest intrinsic (no contraction to FMA).
#pragma once
#include "gat_device.h"

namespace gat {

constexpr int kStatsThreads = 256;
constexpr int kStatsWaves = kStatsThreads / kWave;
constexpr int kNpChunk = 8192, kNpBlock = 128;

struct StatsArgs {
  const int64_t* counts;      // [row][S] 8-byte slots: int64, or IEEE double bits for rows flagged in is_double
  int64_t row_stride;
  int32_t n_rows, S;
  const uint8_t* is_double;   // per row
  const double* vals;         // per row: the value the p-value is asked for (observed, or observed / reference fold)
  double* out;                // per row 8 doubles: mean, sum of squared deviations from it, lower95, upper95, n_less, n_eq, 0, 0
  int32_t lo_i, hi_i;         // sorted positions of the interval's ends
  // the last, partial chunk: blocks [leaf_off, +leaf_len) relative to its start and the order in which their sums combine
  const int32_t* leaf_off;
  const int32_t* leaf_len;
  int32_t n_leaves;
  const int32_t* prog;        // >= 0: push that block's sum; -1: replace the two topmost by their sum (left + right)
  int32_t n_prog;
};

template <typename F>
__device__ __forceinline__ double np_block_sum(F get, int off, int len) {
#pragma clang fp contract(off)
  if (len < 8) {
    double r = 0.0;
    for (int i = 0; i < len; ++i) r = __dadd_rn(r, get(off + i));
    return r;
  }
  double r[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) r[j] = get(off + j);
  int i = 8;
  for (; i < len - (len % 8); i += 8) {
#pragma unroll
    for (int j = 0; j < 8; ++j) r[j] = __dadd_rn(r[j], get(off + i + j));
  }
  double res = __dadd_rn(__dadd_rn(__dadd_rn(r[0], r[1]), __dadd_rn(r[2], r[3])), __dadd_rn(__dadd_rn(r[4], r[5]), __dadd_rn(r[6], r[7])));
  for (; i < len; ++i) res = __dadd_rn(res, get(off + i));
  return res;
}

// numpy's sum of get(0..S): result valid in thread 0.  chunk_sum: LDS, S / 8192 + 1 doubles; leaf_sum: LDS, n_leaves doubles.
template <typename F>
__device__ __forceinline__ double np_sum(const StatsArgs& A, F get, double* chunk_sum, double* leaf_sum) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n_full = A.S / kNpChunk, rest = A.S - n_full * kNpChunk;
  for (int c = wave; c < n_full; c += kStatsWaves) {
    double s = np_block_sum(get, c * kNpChunk + lane * kNpBlock, kNpBlock);
    // the recursion over 64 equal blocks is a perfect tree: (0+1), (2+3), ... then pairs of those, left operand first
#pragma unroll
    for (int m = 1; m < kWave; m <<= 1) {
      const double t = __shfl_down(s, m);
      s = __dadd_rn(s, t);
    }
    if (lane == 0) chunk_sum[c] = s;
  }
  if (rest > 0) {
    const int w = n_full % kStatsWaves;
    if (wave == w)
      for (int k = lane; k < A.n_leaves; k += kWave) leaf_sum[k] = np_block_sum(get, n_full * kNpChunk + A.leaf_off[k], A.leaf_len[k]);
  }
  __syncthreads();
  double total = 0.0;
  if (tid == 0) {
    bool first = true;
    for (int c = 0; c < n_full; ++c) { total = first ? chunk_sum[c] : __dadd_rn(total, chunk_sum[c]); first = false; }
    if (rest > 0) {
      double stack[24];
      int sp = 0;
      for (int q = 0; q < A.n_prog; ++q) {
        const int op = A.prog[q];
        if (op >= 0) stack[sp++] = leaf_sum[op];
        else { const double b = stack[--sp], a = stack[--sp]; stack[sp++] = __dadd_rn(a, b); }
      }
      total = first ? stack[0] : __dadd_rn(total, stack[0]);
    }
  }
  __syncthreads();
  return total;
}

__device__ __forceinline__ unsigned long long stats_key(double d) {
  const unsigned long long b = (unsigned long long)__double_as_longlong(d);
  return (b >> 63) ? ~b : (b | 0x8000000000000000ull);      // order of the keys == order of the doubles
}
__device__ __forceinline__ double stats_unkey(unsigned long long k) {
  const unsigned long long b = (k >> 63) ? (k & 0x7fffffffffffffffull) : ~k;
  return __longlong_as_double((long long)b);
}

// one workgroup per row
__global__ __launch_bounds__(kStatsThreads) void k_null_stats(StatsArgs A) {
  extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
  __shared__ uint32_t hist[2][256];
  __shared__ double bc[2];
  __shared__ unsigned long long sel[2];
  __shared__ uint32_t cnt[2];
  const int row = blockIdx.x;
  if (row >= A.n_rows) return;
  const int tid = threadIdx.x, lane = tid & 63;
  const int64_t* __restrict__ src = A.counts + (int64_t)row * A.row_stride;
  const bool dbl = A.is_double[row] != 0;
  double* chunk_sum = reinterpret_cast<double*>(lds);
  double* leaf_sum = chunk_sum + (A.S / kNpChunk + 1);
  auto value = [&](int i) -> double { const int64_t v = src[i]; return dbl ? __longlong_as_double(v) : (double)v; };
  // mean = numpy.mean(samples)
  const double total = np_sum(A, value, chunk_sum, leaf_sum);
  if (tid == 0) bc[0] = __ddiv_rn(total, (double)A.S);
  __syncthreads();
  const double mean = bc[0];
  // stddev = numpy.std(samples) = sqrt(mean(abs(x - mean) ** 2))
  // (x - mean) ** 2, two roundings: the library is built with -ffp-contract=off (HIP's __dmul_rn / __dadd_rn are plain
  //  operators and clang contracts a * b + c into one fused operation by default -- a different number)
  auto sq = [&](int i) -> double {
#pragma clang fp contract(off)
    const double d = __dsub_rn(value(i), mean);
    return __dmul_rn(d, d);
  };
  const double total2 = np_sum(A, sq, chunk_sum, leaf_sum);
  // samples below / equal to the observed value (getTwoSidedPValue's searchsorted and tie walk, gat/Engine.pyx:1543-1576)
  const double val = A.vals[row];
  uint32_t n_less = 0, n_eq = 0;
  for (int i = tid; i < A.S; i += kStatsThreads) { const double v = value(i); n_less += v < val ? 1u : 0u; n_eq += v == val ? 1u : 0u; }
  n_less = wave_total_u32(n_less);
  n_eq = wave_total_u32(n_eq);
  if (tid == 0) { cnt[0] = 0; cnt[1] = 0; }
  __syncthreads();
  if (lane == 0) { atomicAdd(&cnt[0], n_less); atomicAdd(&cnt[1], n_eq); }
  // the values at sorted positions lo_i and hi_i: radix select, 8 bits a pass, both positions in every pass
  unsigned long long prefix[2] = {0ull, 0ull}, mask = 0ull;
  uint32_t k[2] = {(uint32_t)A.lo_i, (uint32_t)A.hi_i};
  for (int shift = 56; shift >= 0; shift -= 8) {
    for (int i = tid; i < 512; i += kStatsThreads) (&hist[0][0])[i] = 0u;
    __syncthreads();
    for (int i = tid; i < A.S; i += kStatsThreads) {
      const unsigned long long key = stats_key(value(i));
      const uint32_t b = (uint32_t)(key >> shift) & 255u;
      if ((key & mask) == prefix[0]) atomicAdd(&hist[0][b], 1u);
      if ((key & mask) == prefix[1]) atomicAdd(&hist[1][b], 1u);
    }
    __syncthreads();
    if (tid < 2) {
      uint32_t run = 0, kk = k[tid];
      int bin = 255;
      for (int b = 0; b < 256; ++b) {
        const uint32_t h = hist[tid][b];
        if (kk < run + h) { bin = b; break; }
        run += h;
      }
      sel[tid] = ((unsigned long long)bin << 32) | (unsigned long long)(kk - run);
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 2; ++q) { prefix[q] |= (sel[q] >> 32) << shift; k[q] = (uint32_t)sel[q]; }
    mask |= 0xffull << shift;
    __syncthreads();
  }
  if (tid == 0) {
    double* o = A.out + (int64_t)row * 8;
    o[0] = mean;
    o[1] = total2;                                       // (the caller divides and takes the root: IEEE sqrt on the host)
    o[2] = stats_unkey(prefix[0]);
    o[3] = stats_unkey(prefix[1]);
    o[4] = (double)cnt[0];
    o[5] = (double)cnt[1];
    o[6] = 0.0; o[7] = 0.0;
  }
}

}  // namespace gat
