// gat_device.h -- wave64 device primitives for the GAT hot path on gfx950 (CDNA4).
//
// Everything here is written for one 64-lane wavefront that owns one Monte-Carlo work unit
// (sample, isochore): the MT19937 state and the segment buffer live in that wave's LDS slice,
// the serial placement chain runs on wave-uniform (scalar) values, and sort / merge /
// intersect / trim are lane-parallel over the LDS buffer.  No MFMA: this path is integer
// compare/index work.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "gat_types.h"

namespace gat {

constexpr int kMtM = 397;
constexpr int kMtLdsWords = 640;   // 624 state words, padded so the segment buffer stays 16-B aligned

__device__ __forceinline__ uint32_t rfl(uint32_t v) {
  return (uint32_t)__builtin_amdgcn_readfirstlane((int)v);
}
__device__ __forceinline__ int rfl(int v) { return __builtin_amdgcn_readfirstlane(v); }

// LDS hand-off between lanes of ONE wave: LDS operations of a wave execute in order, so only the
// compiler has to be kept from reordering; the workgroup is a single wave (launch_bounds 64).
// MEM = false: the data handed between the lanes is in LDS -- wait for the wave's LDS operations, nothing else; loads from
// global memory the wave has in flight stay in flight (a __syncthreads() here -- s_waitcnt vmcnt(0) lgkmcnt(0) + s_barrier
// -- drained them at every step of a sort or merge).  MEM = true: the list itself lives in global memory (the HUGE
// variants of k_sampler / k_contig, k_resume_big): the full form, as before.
template <bool MEM = false>
__device__ __forceinline__ void wave_sync() {
  if constexpr (MEM) __syncthreads();
  else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
}

// the same between the lanes of one wave inside a workgroup of several waves that do not run in step (no barrier may be
// used): LDS operations of a wave execute in order; wait for them and keep the compiler from moving accesses across
__device__ __forceinline__ void wave_fence() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

__device__ __forceinline__ uint64_t lanemask_lt(int lane) { return (1ull << lane) - 1ull; }

// ------------------------------------------------------------------------------------------
// numpy legacy RandomState == MT19937 (init_genrand seeding) + masked rejection, restated for a
// wave: the 624-word state sits in LDS, the twist runs 64 lanes wide in place, and tempered
// outputs are handed out from a lane-distributed register (v_readlane) 64 at a time.
// Reference call sites: gat/Engine.pyx:299,326,420,433,620 (numpy.random.randint).
struct WaveRng {
  uint32_t* mt;     // LDS, kMtN words
  uint32_t rbuf;    // lane i holds tempered output (pos & ~63) + i
  int pos;          // next state word to hand out, 0..624 (wave-uniform)
  uint32_t ndraws;  // raw outputs consumed (wave-uniform)
  // pre-generated source (k_rng rows): output j of this stream is pre[j*64]; exhausted is set (and
  // zeros are returned, which every rejection loop accepts) when the rows run out, so that the
  // caller can redo the unit from its seed with the in-LDS generator
  const uint32_t* pre;
  uint32_t pre_j, pre_rows;
  uint32_t pre_base;   // rbuf holds outputs [pre_base, pre_base+64) of the stream (one vector load per 64 draws)
  bool use_pre, exhausted;
  // where the rows run out the stream goes on from the in-LDS generator, seeded and moved up to the outputs consumed so far
  // (rng_switch: the seeding chain + one twist per 624 outputs -- microseconds, against replaying every placement of the unit
  // from its seed): can_switch says the caller has the LDS words (mt) and the stream's seed for it
  bool can_switch;
  uint32_t seed;
};

// init_genrand(seed): mt[0]=seed; mt[i] = 1812433253*(mt[i-1]^(mt[i-1]>>30)) + i.
// The recurrence is serial; it runs on scalar registers and is scattered to the lanes with
// v_writelane (lane index as an inline constant), one LDS store per 64 words.
template <int J>
struct SeedStep {
  static __device__ __forceinline__ void run(uint32_t& buf, uint32_t& s, uint32_t base) {
    asm("v_writelane_b32 %0, %1, %2" : "+v"(buf) : "s"(s), "n"(J));
    s = 1812433253u * (s ^ (s >> 30)) + (base + (uint32_t)J + 1u);
    SeedStep<J + 1>::run(buf, s, base);
  }
};
template <>
struct SeedStep<kWave> {
  static __device__ __forceinline__ void run(uint32_t&, uint32_t&, uint32_t) {}
};

__device__ __forceinline__ void rng_seed(WaveRng& r, uint32_t seed, int lane) {
  uint32_t s = seed;
  for (int base = 0; base < kMtN; base += kWave) {
    uint32_t buf = 0;
    SeedStep<0>::run(buf, s, (uint32_t)base);
    if (base + lane < kMtN) r.mt[base + lane] = buf;
  }
  r.pos = kMtN;
  r.rbuf = 0;
  r.ndraws = 0;
  r.use_pre = false;
  r.exhausted = false;
  r.can_switch = false;
  wave_sync();
}

__device__ __forceinline__ void rng_twist(WaveRng& r, int lane);

// The stream's pre-generated rows are used up: seed the in-LDS generator and move it to output number r.ndraws -- what
// numpy's generator would hand out next.  (The LDS words may have served as scratch while the rows lasted.)
__device__ __forceinline__ void rng_switch(WaveRng& r, int lane) {
  const uint32_t n = r.ndraws, seed = r.seed;
  const uint32_t* pre = r.pre;
  rng_seed(r, seed, lane);
  const uint32_t blocks = n / (uint32_t)kMtN, rem = n - blocks * (uint32_t)kMtN;
  for (uint32_t b = 0; b <= blocks; ++b) rng_twist(r, lane);         // the state behind outputs [blocks * 624, + 624)
  r.pos = (int)rem;
  r.ndraws = n;
  r.pre = pre;
  r.seed = seed;
  r.can_switch = false;
  if (rem & (uint32_t)(kWave - 1)) {                                 // (rng_next tempers a block of 64 when it enters one)
    const int i = (int)(rem & ~(uint32_t)(kWave - 1)) + lane;
    uint32_t y = r.mt[i < kMtN ? i : kMtN - 1];
    y ^= (y >> 11);
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= (y >> 18);
    r.rbuf = y;
  }
}

// genrand twist, in place, 64 lanes per step.  Word i needs old[i], old[i+1] and
// (i < 227 ? old[i+397] : new[i-227]); processing chunks of 64 in increasing i keeps exactly
// those versions in LDS (a chunk never reads a word it writes except through old[i+1], which
// is loaded before the store of the same instruction group).
__device__ __forceinline__ void rng_twist(WaveRng& r, int lane) {
  for (int base = 0; base < kMtN; base += kWave) {
    const int i = base + lane;
    uint32_t v = 0;
    if (i < kMtN) {
      const uint32_t a = r.mt[i];
      const uint32_t b = r.mt[i + 1 == kMtN ? 0 : i + 1];
      const uint32_t c = r.mt[i + kMtM >= kMtN ? i + kMtM - kMtN : i + kMtM];
      const uint32_t y = (a & 0x80000000u) | (b & 0x7fffffffu);
      v = c ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
    }
    wave_sync();
    if (i < kMtN) r.mt[i] = v;
    wave_sync();
  }
}

__device__ __forceinline__ uint32_t rng_next(WaveRng& r, int lane) {
  if (r.use_pre && r.pre_j >= r.pre_rows) {
    if (!r.can_switch) { r.exhausted = true; return 0u; }
    rng_switch(r, lane);
  }
  if (r.use_pre) {
    if (r.pre_j - r.pre_base >= (uint32_t)kWave) {
      r.pre_base = r.pre_j;
      const uint32_t row = r.pre_base + (uint32_t)lane;
      r.rbuf = row < r.pre_rows ? r.pre[(size_t)row * kWave] : 0u;
    }
    const uint32_t x = (uint32_t)__builtin_amdgcn_readlane((int)r.rbuf, (int)(r.pre_j - r.pre_base));
    r.pre_j++;
    r.ndraws++;
    return x;
  }
  if (r.pos == kMtN) {
    rng_twist(r, lane);
    r.pos = 0;
  }
  if ((r.pos & (kWave - 1)) == 0) {
    const int i = r.pos + lane;
    uint32_t y = r.mt[i < kMtN ? i : kMtN - 1];
    y ^= (y >> 11);
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= (y >> 18);
    r.rbuf = y;
  }
  const uint32_t x = (uint32_t)__builtin_amdgcn_readlane((int)r.rbuf, r.pos & (kWave - 1));
  r.pos++;
  r.ndraws++;
  return x;
}

// numpy.random.randint(lo, lo+range+1) - lo for range < 2^32-1:
// random_bounded_uint64 masked path: range 0 consumes nothing; else reject (next & mask) > range.
__device__ __forceinline__ uint32_t rng_range(WaveRng& r, uint32_t range, int lane) {
  if (range == 0) return 0;
  const uint32_t mask = 0xffffffffu >> __builtin_clz(range);
  uint32_t v;
  do {
    v = rng_next(r, lane) & mask;
  } while (v > range);
  return v;
}

// utils/gat_utils.c:36-60 searchsorted with cmpPosition (gat/Engine.pyx:119): leftmost i with
// (int)(a[i]-t) >= 0.  All operands wave-uniform: runs on the scalar unit / scalar cache.
__device__ __forceinline__ int bisect_u32(const uint32_t* __restrict__ a, int n, uint32_t t) {
  int lo = 0, hi = n;
  while (lo < hi) {
    const int mid = lo + ((hi - lo) >> 1);
    const uint32_t v = a[mid];
    if ((int32_t)(v - t) < 0) lo = mid + 1; else hi = mid;
  }
  return lo;
}

// ------------------------------------------------------------------------------------------
// wave reductions / scans
__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t v) {
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d);
  return v;
}
// all-lanes reduction on the DPP path (result broadcast from lane 63); id = identity of op
template <typename Op>
__device__ __forceinline__ int wave_reduce_dpp(int v, int id, Op op) {
  int x = v;
  x = op(x, __builtin_amdgcn_update_dpp(id, x, 0xB1, 0xf, 0xf, false));    // quad_perm [1,0,3,2]
  x = op(x, __builtin_amdgcn_update_dpp(id, x, 0x4E, 0xf, 0xf, false));    // quad_perm [2,3,0,1]
  x = op(x, __builtin_amdgcn_update_dpp(id, x, 0x141, 0xf, 0xf, false));   // row_half_mirror
  x = op(x, __builtin_amdgcn_update_dpp(id, x, 0x140, 0xf, 0xf, false));   // row_mirror
  x = op(x, __builtin_amdgcn_update_dpp(id, x, 0x142, 0xa, 0xf, false));   // row_bcast15 -> rows 1, 3
  x = op(x, __builtin_amdgcn_update_dpp(id, x, 0x143, 0xc, 0xf, false));   // row_bcast31 -> rows 2, 3
  return __builtin_amdgcn_readlane(x, 63);
}
__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v) {
  return (uint32_t)wave_reduce_dpp((int)v, -1, [](int a, int b) { return (int)((uint32_t)a < (uint32_t)b ? (uint32_t)a : (uint32_t)b); });
}
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v) {
  return (uint32_t)wave_reduce_dpp((int)v, 0, [](int a, int b) { return (int)((uint32_t)a > (uint32_t)b ? (uint32_t)a : (uint32_t)b); });
}
// wave total on the VALU data-parallel-primitive path (no LDS crossbar): quad swaps, row mirrors,
// then the two row broadcasts; the total lands in lane 63
__device__ __forceinline__ uint32_t wave_total_u32(uint32_t v) {
  int x = (int)v;
  x += __builtin_amdgcn_update_dpp(0, x, 0xB1, 0xf, 0xf, true);    // quad_perm [1,0,3,2]
  x += __builtin_amdgcn_update_dpp(0, x, 0x4E, 0xf, 0xf, true);    // quad_perm [2,3,0,1]
  x += __builtin_amdgcn_update_dpp(0, x, 0x141, 0xf, 0xf, true);   // row_half_mirror
  x += __builtin_amdgcn_update_dpp(0, x, 0x140, 0xf, 0xf, true);   // row_mirror
  x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xa, 0xf, false);  // row_bcast15 -> rows 1, 3
  x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xc, 0xf, false);  // row_bcast31 -> rows 2, 3
  return (uint32_t)__builtin_amdgcn_readlane(x, 63);
}
// inclusive wave scans on the VALU cross-lane (DPP) path: shifts by 1, 2, 3 inside a row of 16, then by 4
// and 8 for the upper banks, then the row broadcasts (lane 15 -> next row, lane 31 -> rows 2, 3).  A lane
// without a source (row start, masked bank / row) receives the identity.  ~14 VALU instructions, no LDS
// crossbar round trips (the __shfl_up form costs six dependent ds_bpermute latencies).
template <typename Op>
__device__ __forceinline__ int wave_incl_scan_dpp(int v, int id, Op op) {
  int x = op(v, __builtin_amdgcn_update_dpp(id, v, 0x111, 0xf, 0xf, false));   // row_shr:1
  x = op(x, __builtin_amdgcn_update_dpp(id, v, 0x112, 0xf, 0xf, false));       // row_shr:2
  x = op(x, __builtin_amdgcn_update_dpp(id, v, 0x113, 0xf, 0xf, false));       // row_shr:3
  x = op(x, __builtin_amdgcn_update_dpp(id, x, 0x114, 0xf, 0xe, false));       // row_shr:4, banks 1-3
  x = op(x, __builtin_amdgcn_update_dpp(id, x, 0x118, 0xf, 0xc, false));       // row_shr:8, banks 2-3
  x = op(x, __builtin_amdgcn_update_dpp(id, x, 0x142, 0xa, 0xf, false));       // row_bcast:15 -> rows 1, 3
  x = op(x, __builtin_amdgcn_update_dpp(id, x, 0x143, 0xc, 0xf, false));       // row_bcast:31 -> rows 2, 3
  return x;
}
__device__ __forceinline__ int32_t wave_incl_max_i32(int32_t m, int lane) {
  (void)lane;
  return wave_incl_scan_dpp(m, INT32_MIN, [](int a, int b) { return a > b ? a : b; });
}
__device__ __forceinline__ uint32_t wave_incl_sum_u32(uint32_t v, int lane) {
  (void)lane;
  return (uint32_t)wave_incl_scan_dpp((int)v, 0, [](int a, int b) { return (int)((uint32_t)a + (uint32_t)b); });
}

// ------------------------------------------------------------------------------------------
// SegmentList.sort (gat/SegmentList.pyx:478-486: qsort by start only) as an in-LDS bitonic
// network in its all-ascending ("flip") form, so indices >= n act as +inf without being stored.
// Tie order among equal starts is unspecified in the reference too; merge() is independent of it.
__device__ __forceinline__ void cmpex(uint2* seg, int i, int j) {
  const uint2 a = seg[i], b = seg[j];
  if (b.x < a.x) { seg[i] = b; seg[j] = a; }
}
template <bool MEM = false>
__device__ __forceinline__ void wave_sort_by_start(uint2* seg, int n, int lane) {
  if (n < 2) return;
  int lP = 1;
  while ((1 << lP) < n) ++lP;
  const int half_total = 1 << (lP - 1);
  wave_sync<MEM>();
  for (int lk = 1; lk <= lP; ++lk) {
    // flip step: i and its mirror image inside blocks of k = 2^lk
    const int hmask = (1 << (lk - 1)) - 1;
    for (int t = lane; t < half_total; t += kWave) {
      const int base = (t >> (lk - 1)) << lk, off = t & hmask;
      const int i = base + off, j = base + ((1 << lk) - 1 - off);
      if (j < n) cmpex(seg, i, j);
    }
    wave_sync<MEM>();
    for (int ld = lk - 2; ld >= 0; --ld) {
      const int dmask = (1 << ld) - 1;
      for (int t = lane; t < half_total; t += kWave) {
        const int i = ((t >> ld) << (ld + 1)) + (t & dmask), j = i + (1 << ld);
        if (j < n) cmpex(seg, i, j);
      }
      wave_sync<MEM>();
    }
  }
}

// The same sort with the elements held in registers (E per lane, element r*64+lane in register r).
// Partners at a distance below 64 are fetched with a cross-lane shuffle, larger
// distances are register-to-register; no LDS round trip or barrier per stage.
template <int E>
__device__ __forceinline__ void sort_stage_lanes(uint32_t (&ks)[E], uint32_t (&ke)[E], int ml, int hb, int lane) {
  const int src = (lane ^ ml) << 2;
#pragma unroll
  for (int r = 0; r < E; ++r) {
    const uint32_t ps = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)ks[r]);
    const uint32_t pe = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)ke[r]);
    const bool lower = (((r << 6) | lane) & (1 << hb)) == 0;
    const bool take = lower ? (ps < ks[r]) : (ps > ks[r]);      // equal starts: neither side moves
    ks[r] = take ? ps : ks[r];
    ke[r] = take ? pe : ke[r];
  }
}
// One list of at most 64 segments, one per lane: the same network (21 stages) with the partner of a stage fetched by a DPP
// modifier wherever it lies within the lane's row of 16 -- 18 stages: quad permutes (xor 1, 2, 3), row_half_mirror (xor 7),
// row_mirror (xor 15), row_ror:8 (xor 8), a shift pair under bank masks (xor 4) -- and by ds_bpermute only across rows (xor 16,
// 31, 63): 6 LDS permutes per list instead of 42.  CTRL: the DPP control word; CTRL2 != 0: second half of the xor-4 pair.
template <int CTRL, int BANK = 0xf, int CTRL2 = 0, int BANK2 = 0>
__device__ __forceinline__ uint32_t lane_partner_dpp(uint32_t v) {
  uint32_t r = (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, CTRL, 0xf, BANK, false);
  if constexpr (CTRL2 != 0) r = (uint32_t)__builtin_amdgcn_update_dpp((int)r, (int)v, CTRL2, 0xf, BANK2, false);
  return r;
}
__device__ __forceinline__ void sort64_exchange(uint32_t& ks, uint32_t& ke, uint32_t ps, uint32_t pe, bool lower) {
  const bool take = lower ? (ps < ks) : (ps > ks);              // equal starts: neither side moves
  ks = take ? ps : ks;
  ke = take ? pe : ke;
}
template <int HB, int CTRL, int BANK = 0xf, int CTRL2 = 0, int BANK2 = 0>
__device__ __forceinline__ void sort64_stage_dpp(uint32_t& ks, uint32_t& ke, int lane) {
  const uint32_t ps = lane_partner_dpp<CTRL, BANK, CTRL2, BANK2>(ks), pe = lane_partner_dpp<CTRL, BANK, CTRL2, BANK2>(ke);
  sort64_exchange(ks, ke, ps, pe, (lane & (1 << HB)) == 0);
}
template <int HB, int ML>
__device__ __forceinline__ void sort64_stage_perm(uint32_t& ks, uint32_t& ke, int lane) {
  const int src = (lane ^ ML) << 2;
  const uint32_t ps = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)ks), pe = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)ke);
  sort64_exchange(ks, ke, ps, pe, (lane & (1 << HB)) == 0);
}
__device__ __forceinline__ void sort64_by_start(uint32_t& ks, uint32_t& ke, int lane) {
  constexpr int X1 = 0xB1, X2 = 0x4E, X3 = 0x1B, X7 = 0x141, X15 = 0x140, X8 = 0x128, SHL4 = 0x104, SHR4 = 0x114;
  auto x1 = [&]() { sort64_stage_dpp<0, X1>(ks, ke, lane); };
  auto x2 = [&]() { sort64_stage_dpp<1, X2>(ks, ke, lane); };
  auto x4 = [&]() { sort64_stage_dpp<2, SHL4, 0x5, SHR4, 0xA>(ks, ke, lane); };
  auto x8 = [&]() { sort64_stage_dpp<3, X8>(ks, ke, lane); };
  x1();                                                                        // blocks of 2: flip
  sort64_stage_dpp<1, X3>(ks, ke, lane); x1();                                 // 4: flip, disperse 1
  sort64_stage_dpp<2, X7>(ks, ke, lane); x2(); x1();                           // 8
  sort64_stage_dpp<3, X15>(ks, ke, lane); x4(); x2(); x1();                    // 16
  sort64_stage_perm<4, 31>(ks, ke, lane); x8(); x4(); x2(); x1();              // 32
  sort64_stage_perm<5, 63>(ks, ke, lane); sort64_stage_perm<4, 16>(ks, ke, lane); x8(); x4(); x2(); x1();   // 64
}

// partner differs in register index (RMASK) and, for the flip steps, mirrors the lane (xor 63)
template <int E, int RMASK, bool MIRROR>
__device__ __forceinline__ void sort_stage_regs(uint32_t (&ks)[E], uint32_t (&ke)[E], int lane) {
  uint32_t ns[E], ne[E];
  const int src = (lane ^ 63) << 2;
#pragma unroll
  for (int r = 0; r < E; ++r) {
    uint32_t ps = ks[r ^ RMASK], pe = ke[r ^ RMASK];
    if (MIRROR) {
      ps = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)ps);
      pe = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)pe);
    }
    // the highest bit of the mask is the top bit of RMASK: lower element iff that bit of r is clear
    constexpr int top = RMASK >= 16 ? 16 : RMASK >= 8 ? 8 : RMASK >= 4 ? 4 : RMASK >= 2 ? 2 : 1;
    const bool lower = (r & top) == 0;
    const bool take = lower ? (ps < ks[r]) : (ps > ks[r]);
    ns[r] = take ? ps : ks[r];
    ne[r] = take ? pe : ke[r];
  }
#pragma unroll
  for (int r = 0; r < E; ++r) { ks[r] = ns[r]; ke[r] = ne[r]; }
}
template <int E>
__device__ __forceinline__ void sort_disperse_below64(uint32_t (&ks)[E], uint32_t (&ke)[E], int ld_from, int lane) {
  for (int ld = ld_from; ld >= 0; --ld) sort_stage_lanes<E>(ks, ke, 1 << ld, ld, lane);
}
template <int E, bool MEM = false>
__device__ __forceinline__ void wave_sort_regs(uint2* seg, int n, int lane) {
  uint32_t ks[E], ke[E];
  wave_sync<MEM>();
#pragma unroll
  for (int r = 0; r < E; ++r) {
    const int i = r * kWave + lane;
    uint2 x = make_uint2(0xffffffffu, 0xffffffffu);
    if (i < n) x = seg[i];
    ks[r] = x.x; ke[r] = x.y;
  }
  for (int lk = 1; lk <= 6; ++lk) {                             // blocks of 2..64: lane-only stages
    sort_stage_lanes<E>(ks, ke, (1 << lk) - 1, lk - 1, lane);   // flip
    sort_disperse_below64<E>(ks, ke, lk - 2, lane);
  }
  if (E >= 2) {                                                 // k = 128
    sort_stage_regs<E, 1, true>(ks, ke, lane);
    sort_disperse_below64<E>(ks, ke, 5, lane);
  }
  if (E >= 4) {                                                 // k = 256
    sort_stage_regs<E, (E >= 4 ? 3 : 0), true>(ks, ke, lane);
    sort_stage_regs<E, 1, false>(ks, ke, lane);
    sort_disperse_below64<E>(ks, ke, 5, lane);
  }
  if (E >= 8) {                                                 // k = 512
    sort_stage_regs<E, (E >= 8 ? 7 : 0), true>(ks, ke, lane);
    sort_stage_regs<E, (E >= 4 ? 2 : 0), false>(ks, ke, lane);
    sort_stage_regs<E, 1, false>(ks, ke, lane);
    sort_disperse_below64<E>(ks, ke, 5, lane);
  }
  if (E >= 16) {                                                // k = 1024
    sort_stage_regs<E, (E >= 16 ? 15 : 0), true>(ks, ke, lane);
    sort_stage_regs<E, (E >= 8 ? 4 : 0), false>(ks, ke, lane);
    sort_stage_regs<E, (E >= 4 ? 2 : 0), false>(ks, ke, lane);
    sort_stage_regs<E, 1, false>(ks, ke, lane);
    sort_disperse_below64<E>(ks, ke, 5, lane);
  }
#pragma unroll
  for (int r = 0; r < E; ++r) {
    const int i = r * kWave + lane;
    if (i < n) seg[i] = make_uint2(ks[r], ke[r]);
  }
  wave_sync<MEM>();
}
// The sort behind the bucket sort: up to a wave's worth in registers, longer lists through the in-LDS network.
// It only runs for short lists, clustered keys and units redone from their seed, and it is kept SMALL on purpose:
// the register networks for 2..16 elements per lane tripled the kernel beyond the instruction cache, which cost
// the common path more than they saved here.
template <bool MEM = false>
__device__ __forceinline__ void wave_sort_auto(uint2* seg, int n, int lane) {
  if (n < 2) return;
  if (n <= 64) wave_sort_regs<1, MEM>(seg, n, lane);
  else wave_sort_by_start<MEM>(seg, n, lane);
}

// Sort by start for keys that are spread over a range (placed segments are): 512 buckets by
// position (about one element per bucket), LDS atomics give every element a slot in its bucket, a
// prefix sum the bucket offsets, and a rank among the few members of its own bucket the final
// index.  ~10x fewer instructions than the sorting network.  Returns false (nothing written) when
// the keys are too clustered (a bucket with more than 16 members, or all starts equal); the caller
// then uses the network.  scratch: 513 words of LDS.
template <int E, bool MEM = false>
__device__ __forceinline__ bool wave_sort_bucket(uint2* seg, int n, uint32_t* scratch, int lane) {
  constexpr int NB = 512, PER = NB / kWave;
  uint32_t ks[E], ke[E];
  wave_sync<MEM>();
  uint32_t lo = 0xffffffffu, hi = 0u;
#pragma unroll
  for (int r = 0; r < E; ++r) {
    const int i = r * kWave + lane;
    ks[r] = 0; ke[r] = 0;
    if (i < n) { const uint2 x = seg[i]; ks[r] = x.x; ke[r] = x.y; lo = x.x < lo ? x.x : lo; hi = x.x > hi ? x.x : hi; }
  }
  lo = wave_min_u32(lo); hi = wave_max_u32(hi);
  const uint32_t span = hi - lo;
  if (span == 0) return false;
  const bool direct = span < (uint32_t)NB;                       // fewer positions than buckets
  const uint32_t scale = direct ? 0u : (uint32_t)(((uint64_t)NB << 32) / ((uint64_t)span + 1u));
  for (int i = lane; i <= NB; i += kWave) scratch[i] = 0;
  wave_sync<MEM>();
  uint32_t bk[E], slot[E];
#pragma unroll
  for (int r = 0; r < E; ++r) {
    const int i = r * kWave + lane;
    bk[r] = 0; slot[r] = 0;
    if (i < n) {
      const uint32_t d = ks[r] - lo;
      bk[r] = direct ? d : __umulhi(d, scale);
      slot[r] = atomicAdd(&scratch[bk[r]], 1u);
    }
  }
  wave_sync<MEM>();
  // exclusive prefix over the bucket counts: lane owns PER consecutive buckets
  uint32_t c[PER], sum = 0, maxc = 0;
#pragma unroll
  for (int q = 0; q < PER; ++q) { c[q] = scratch[lane * PER + q]; sum += c[q]; maxc = c[q] > maxc ? c[q] : maxc; }
  uint32_t run = wave_incl_sum_u32(sum, lane) - sum;
  maxc = wave_max_u32(maxc);
  if (maxc > 16u) return false;
  wave_sync<MEM>();
#pragma unroll
  for (int q = 0; q < PER; ++q) { scratch[lane * PER + q] = run; run += c[q]; }
  if (lane == kWave - 1) scratch[NB] = run;
  wave_sync<MEM>();
  uint32_t base[E], cnt[E];
#pragma unroll
  for (int r = 0; r < E; ++r) {
    const int i = r * kWave + lane;
    base[r] = 0; cnt[r] = 0;
    if (i < n) { base[r] = scratch[bk[r]]; cnt[r] = scratch[bk[r] + 1] - base[r]; seg[base[r] + slot[r]] = make_uint2(ks[r], ke[r]); }
  }
  wave_sync<MEM>();
  uint32_t rank[E];
#pragma unroll
  for (int r = 0; r < E; ++r) rank[r] = 0;
  for (uint32_t m = 0; m < maxc; ++m) {
#pragma unroll
    for (int r = 0; r < E; ++r) {
      if (m < cnt[r]) {
        const uint32_t s2 = seg[base[r] + m].x;
        rank[r] += (s2 < ks[r] || (s2 == ks[r] && m < slot[r])) ? 1u : 0u;
      }
    }
  }
  wave_sync<MEM>();
#pragma unroll
  for (int r = 0; r < E; ++r) {
    const int i = r * kWave + lane;
    if (i < n) seg[base[r] + rank[r]] = make_uint2(ks[r], ke[r]);
  }
  wave_sync<MEM>();
  return true;
}
// Counting sort by position bucket for long lists whose unsorted source is still in global memory
// (the slab k_place wrote): histogram pass, exclusive prefix, scatter pass into the LDS list
// (atomic cursor per bucket), then every lane insertion-sorts whole buckets (about one element
// each).  nb buckets (power of two >= n), scratch = nb + 1 words of LDS.  Returns false (dst not
// written) when the keys are too clustered for that to be cheap.
__device__ __forceinline__ bool wave_sort_bucket_global(uint2* dst, const uint2* __restrict__ src, int n,
                                                        uint32_t* scratch, int nb, int lane) {
  // the three passes over the source read it in batches of kB rounds: kB loads in flight per lane instead of one
  // (one wave per unit, three waves per CU at these list lengths: nothing else hides the latency)
  constexpr int kB = 8;
  uint32_t lo = 0xffffffffu, hi = 0u;
  for (int base = 0; base < n; base += kB * kWave) {
    uint32_t x[kB];
#pragma unroll
    for (int q = 0; q < kB; ++q) { const int i = base + q * kWave + lane; x[q] = i < n ? src[i].x : 0u; }
#pragma unroll
    for (int q = 0; q < kB; ++q) {
      if (base + q * kWave + lane < n) { lo = x[q] < lo ? x[q] : lo; hi = x[q] > hi ? x[q] : hi; }
    }
  }
  lo = wave_min_u32(lo); hi = wave_max_u32(hi);
  const uint32_t span = hi - lo;
  if (span == 0) return false;
  const bool direct = span < (uint32_t)nb;
  const uint32_t scale = direct ? 0u : (uint32_t)(((uint64_t)nb << 32) / ((uint64_t)span + 1u));
  wave_sync();
  for (int i = lane; i <= nb; i += kWave) scratch[i] = 0;
  wave_sync();
  for (int base = 0; base < n; base += kB * kWave) {
    uint32_t x[kB];
#pragma unroll
    for (int q = 0; q < kB; ++q) { const int i = base + q * kWave + lane; x[q] = i < n ? src[i].x : 0u; }
#pragma unroll
    for (int q = 0; q < kB; ++q) {
      if (base + q * kWave + lane < n) { const uint32_t d = x[q] - lo; atomicAdd(&scratch[direct ? d : __umulhi(d, scale)], 1u); }
    }
  }
  wave_sync();
  // exclusive prefix: lane owns nb/64 consecutive buckets
  const int per = nb / kWave;
  uint32_t sum = 0, maxc = 0;
  for (int q = 0; q < per; ++q) { const uint32_t c = scratch[lane * per + q]; sum += c; maxc = c > maxc ? c : maxc; }
  uint32_t run = wave_incl_sum_u32(sum, lane) - sum;
  if (wave_max_u32(maxc) > 48u) return false;
  for (int q = 0; q < per; ++q) { const uint32_t c = scratch[lane * per + q]; scratch[lane * per + q] = run; run += c; }
  wave_sync();
  for (int base = 0; base < n; base += kB * kWave) {
    uint2 v[kB];
#pragma unroll
    for (int q = 0; q < kB; ++q) { const int i = base + q * kWave + lane; v[q] = i < n ? src[i] : make_uint2(0u, 0u); }
#pragma unroll
    for (int q = 0; q < kB; ++q) {
      if (base + q * kWave + lane < n) {
        const uint32_t d = v[q].x - lo;
        const uint32_t pos = atomicAdd(&scratch[direct ? d : __umulhi(d, scale)], 1u);
        dst[pos] = v[q];
      }
    }
  }
  wave_sync();
  // scratch[b] is now the END of bucket b; lane sorts the buckets lane, lane+64, ...
  for (int b = lane; b < nb; b += kWave) {
    const int e = (int)scratch[b], s0 = b ? (int)scratch[b - 1] : 0;
    for (int i = s0 + 1; i < e; ++i) {
      const uint2 v = dst[i];
      int j = i - 1;
      while (j >= s0 && dst[j].x > v.x) { dst[j + 1] = dst[j]; --j; }
      dst[j + 1] = v;
    }
  }
  wave_sync();
  return true;
}

// The same counting sort with the list in registers (E rounds of 64, all loaded in ONE round trip by the caller) instead
// of three passes over the slab, each a round trip or two of its own: for a kernel lean enough to hold 2 E registers more.
template <int E>
__device__ __forceinline__ bool wave_sort_bucket_regs(uint2* dst, const uint2 (&v)[E], int n, uint32_t* scratch, int nb, int lane) {
  uint32_t lo = 0xffffffffu, hi = 0u;
#pragma unroll
  for (int q = 0; q < E; ++q)
    if (q * kWave + lane < n) { lo = v[q].x < lo ? v[q].x : lo; hi = v[q].x > hi ? v[q].x : hi; }
  lo = wave_min_u32(lo); hi = wave_max_u32(hi);
  const uint32_t span = hi - lo;
  if (span == 0) return false;
  const bool direct = span < (uint32_t)nb;
  const uint32_t scale = direct ? 0u : (uint32_t)(((uint64_t)nb << 32) / ((uint64_t)span + 1u));
  wave_sync();
  for (int i = lane; i <= nb; i += kWave) scratch[i] = 0;
  wave_sync();
#pragma unroll
  for (int q = 0; q < E; ++q)
    if (q * kWave + lane < n) { const uint32_t d = v[q].x - lo; atomicAdd(&scratch[direct ? d : __umulhi(d, scale)], 1u); }
  wave_sync();
  const int per = nb / kWave;
  uint32_t sum = 0, maxc = 0;
  for (int q = 0; q < per; ++q) { const uint32_t c = scratch[lane * per + q]; sum += c; maxc = c > maxc ? c : maxc; }
  uint32_t run = wave_incl_sum_u32(sum, lane) - sum;
  if (wave_max_u32(maxc) > 48u) return false;
  for (int q = 0; q < per; ++q) { const uint32_t c = scratch[lane * per + q]; scratch[lane * per + q] = run; run += c; }
  wave_sync();
#pragma unroll
  for (int q = 0; q < E; ++q)
    if (q * kWave + lane < n) {
      const uint32_t d = v[q].x - lo;
      dst[atomicAdd(&scratch[direct ? d : __umulhi(d, scale)], 1u)] = v[q];
    }
  wave_sync();
  for (int b = lane; b < nb; b += kWave) {                          // scratch[b] is now the END of bucket b
    const int e = (int)scratch[b], s0 = b ? (int)scratch[b - 1] : 0;
    for (int i = s0 + 1; i < e; ++i) {
      const uint2 x = dst[i];
      int j = i - 1;
      while (j >= s0 && dst[j].x > x.x) { dst[j + 1] = dst[j]; --j; }
      dst[j + 1] = x;
    }
  }
  wave_sync();
  return true;
}

// bucket sort when possible (scratch available, list short enough), else the sorting network
template <int MAXE = 8, bool MEM = false>
__device__ __forceinline__ void wave_sort_fast(uint2* seg, int n, uint32_t* scratch, int lane) {
  if (n < 2) return;
  bool done = false;
  if (scratch != nullptr && n > 64) {
    if (n <= 256) done = wave_sort_bucket<4, MEM>(seg, n, scratch, lane);
    else if (n <= 512) done = wave_sort_bucket<8, MEM>(seg, n, scratch, lane);
    // (k_sampler, MAXE = 8: the first consolidation sorts longer lists straight from the slab, wave_sort_bucket_global;
    //  a later full sort of such a list -- after a trim with many new segments -- is rare and takes the network)
    else if (MAXE >= 16 && n <= 1024) done = wave_sort_bucket<16, MEM>(seg, n, scratch, lane);
  }
  if (!done) wave_sort_auto<MEM>(seg, n, lane);
}

// SegmentList.merge(0) (gat/SegmentList.pyx:756-816) on a list already sorted by start (empty
// segments anywhere are skipped, as the reference skips them): in place, 64 elements per step.
// head[i] = first non-empty, or int32(start) - 0 > running max end; the previous group's end is
// the running max seen just before the next head.  Returns the new length.
template <bool MEM = false>
__device__ __forceinline__ int wave_merge0(uint2* seg, int n, int lane) {
  int count = 0;
  int32_t carry = INT32_MIN;
  bool any = false;
  wave_sync<MEM>();
  for (int base = 0; base < n; base += kWave) {
    const int i = base + lane;
    uint32_t s = 0, e = 0;
    bool valid = false;
    if (i < n) {
      const uint2 v = seg[i];
      s = v.x; e = v.y;
      valid = (s != e);
    }
    int32_t m = wave_incl_max_i32(valid ? (int32_t)e : INT32_MIN, lane);
    const int32_t incl = m > carry ? m : carry;
    const int32_t excl = __builtin_amdgcn_update_dpp(carry, incl, 0x138, 0xf, 0xf, false);   // wave_shr:1, lane 0 keeps carry
    const uint64_t vb = __ballot(valid);
    const bool prev_valid = any || (vb & lanemask_lt(lane)) != 0;
    const bool head = valid && (!prev_valid || (int32_t)s > excl);
    const uint64_t hb = __ballot(head);
    const int pos = count + __popcll(hb & lanemask_lt(lane));
    wave_sync<MEM>();
    if (head) {
      seg[pos].x = s;
      if (pos > 0) seg[pos - 1].y = (uint32_t)excl;
    }
    count += __popcll(hb);
    carry = __builtin_amdgcn_readlane(incl, kWave - 1);
    any = any || (vb != 0);
    wave_sync<MEM>();
  }
  if (count > 0 && lane == 0) seg[count - 1].y = (uint32_t)carry;
  wave_sync<MEM>();
  return count;
}

// Insert nS <= 64 unsorted segments seg[nU..nU+nS) into the sorted list seg[0..nU) (the same
// result as re-sorting everything; equal starts may end up in a different order than qsort would
// leave them, which merge() does not see).  Every element's final index is its own index plus
// the number of elements of the other list that sort before it; the old list is shifted in
// place from its last 64-row backwards (shifts are monotone and <= nS, so nothing unread is
// overwritten) and rows in front of the first insertion point are not touched.
template <bool MEM = false>
__device__ __forceinline__ void wave_insert_sorted(uint2* seg, int nU, int nS, int lane) {
  wave_sync<MEM>();
  uint2 nv = make_uint2(0xffffffffu, 0xffffffffu);
  if (lane < nS) nv = seg[nU + lane];
  int nrank = 0;
  for (int j = 0; j < nS; ++j) {
    const uint32_t sj = (uint32_t)__builtin_amdgcn_readlane((int)nv.x, j);
    nrank += (sj < nv.x || (sj == nv.x && j < lane)) ? 1 : 0;
  }
  int cu = 0;
  if (lane < nS) {
    int lo = 0, hi = nU;
    while (lo < hi) {
      const int mid = lo + ((hi - lo) >> 1);
      if (seg[mid].x <= nv.x) lo = mid + 1; else hi = mid;
    }
    cu = lo;
  }
  const int newpos = cu + nrank;
  // the old element at index i moves up by the number of new elements that go in at or before it, i.e. with
  // insertion index cu <= i.  Lane r gets the r-th smallest insertion index (the new elements in sorted order), so
  // that number is a ballot for a whole block of 64 old elements plus the few new ones that fall inside the block
  int cs_sorted = 0x7fffffff;
  {
    // scatter cu to the lane given by its rank: lane r reads from the lane whose nrank == r
    int src_lane = 0;
    for (int j = 0; j < nS; ++j) {
      const int rj = __builtin_amdgcn_readlane(nrank, j);
      if (rj == lane) src_lane = j;
    }
    const int got = __builtin_amdgcn_ds_bpermute(src_lane << 2, cu);
    if (lane < nS) cs_sorted = got;
  }
  for (int base = ((nU - 1) >> 6) << 6; base >= 0; base -= kWave) {
    const int i = base + lane;
    const bool ok = i < nU;
    uint2 v = make_uint2(0u, 0u);
    if (ok) v = seg[i];
    const int below = __popcll(__ballot(cs_sorted < base + 1));            // new elements going in at or before `base`
    const int upto = __popcll(__ballot(cs_sorted <= base + kWave - 1));    // ... at or before the block's last element
    int sh = below;
    for (int j = below; j < upto; ++j) {
      const int cj = __builtin_amdgcn_readlane(cs_sorted, j);
      sh += cj <= i ? 1 : 0;
    }
    sh = ok ? sh : 0;
    wave_sync<MEM>();
    if (sh > 0) seg[i + sh] = v;
    wave_sync<MEM>();
    if (upto == 0) break;                                                  // nothing goes in at or before this block
  }
  if (lane < nS) seg[newpos] = nv;
  wave_sync<MEM>();
}

// ------------------------------------------------------------------------------------------
// WsTree: static 16-ary search tree over a sorted u32 array in global memory.  Level 0 is the array
// itself padded to whole 64-byte nodes; entry j of level l+1 is the largest key of node j of level
// l.  A search reads ONE node per level (four 16-byte loads) and counts its keys below the target:
// 4 dependent round trips for a 50 000-segment workspace where a binary search makes 16.  The pad
// value is never below any target, and a target above every key walks down the last nodes.
// (kWsTreeMin, kWsTreeLevels: gat_types.h)
struct WsTreeGeom {                 // wave-uniform, derived from the number of keys
  int nlev;
  int off[kWsTreeLevels];           // first word of the level
  int nodes[kWsTreeLevels];         // nodes of the level
};
__device__ __forceinline__ WsTreeGeom ws_tree_geom(int n) {
  WsTreeGeom g;
  int off = 0, l = 0;
#pragma unroll
  for (int i = 0; i < kWsTreeLevels; ++i) { g.off[i] = 0; g.nodes[i] = 1; }
  bool done = false;
#pragma unroll
  for (int i = 0; i < kWsTreeLevels; ++i) {
    if (!done) {
      const int nodes = (n + 15) >> 4;
      g.off[i] = off; g.nodes[i] = nodes;
      off += nodes << 4;
      l = i + 1;
      done = n <= 16;
      n = nodes;
    }
  }
  g.nlev = l;
  return g;
}
// SIGNED: keys compare as the reference's cmpPosition does ((int)(key - target) < 0, utils/gat_utils.c:36 +
// gat/Engine.pyx:119; pad 0x7fffffff); otherwise as unsigned start < target (pad 0xffffffff).
template <bool SIGNED>
__device__ __forceinline__ int ws_tree_below16(const uint4 a, const uint4 b, const uint4 c, const uint4 d, uint32_t t) {
  auto lt = [&](uint32_t k) -> int { return SIGNED ? ((int32_t)(k - t) < 0 ? 1 : 0) : (k < t ? 1 : 0); };
  return lt(a.x) + lt(a.y) + lt(a.z) + lt(a.w) + lt(b.x) + lt(b.y) + lt(b.z) + lt(b.w) +
         lt(c.x) + lt(c.y) + lt(c.z) + lt(c.w) + lt(d.x) + lt(d.y) + lt(d.z) + lt(d.w);
}
// number of keys below each of U targets (the U searches of a lane advance level by level, loads overlapped)
template <bool SIGNED, int U>
__device__ __forceinline__ void ws_tree_count(const uint32_t* __restrict__ tree, const WsTreeGeom& g,
                                              const uint32_t (&t)[U], int (&pos)[U]) {
#pragma unroll
  for (int u = 0; u < U; ++u) pos[u] = 0;
#pragma unroll
  for (int l = kWsTreeLevels - 1; l >= 0; --l) {
    if (l >= g.nlev) continue;
    const uint4* __restrict__ lev = reinterpret_cast<const uint4*>(tree + g.off[l]);
    const int last = g.nodes[l] - 1;
    uint4 q[U][4];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      pos[u] = pos[u] < last ? pos[u] : last;
#pragma unroll
      for (int w = 0; w < 4; ++w) q[u][w] = lev[pos[u] * 4 + w];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) pos[u] = pos[u] * 16 + ws_tree_below16<SIGNED>(q[u][0], q[u][1], q[u][2], q[u][3], t[u]);
  }
}

// bases of a normalized list W (starts/ends + cdf[i] = cumlen_i - 1) below U positions per lane: k = #segments with start < p
// from the tree, then the bases of the first k-1 segments plus the part of segment k-1 below p
template <int U>
__device__ __forceinline__ void cov_below_tree(const uint2* __restrict__ w, const uint32_t* __restrict__ cdf,
                                               const uint32_t* __restrict__ tree, const WsTreeGeom& g,
                                               const uint32_t (&p)[U], uint32_t (&out)[U]) {
  int k[U];
  ws_tree_count<false, U>(tree, g, p, k);
  uint2 prev[U];
  uint32_t before[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    prev[u] = w[k[u] > 0 ? k[u] - 1 : 0];
    before[u] = k[u] >= 2 ? cdf[k[u] - 2] + 1u : 0u;
  }
#pragma unroll
  for (int u = 0; u < U; ++u)
    out[u] = k[u] == 0 ? 0u : before[u] + (p[u] < prev[u].y ? p[u] : prev[u].y) - prev[u].x;
}
// overlap of R segments per lane with a long normalized list (seg_overlap_with, via the tree)
template <int R>
__device__ __forceinline__ void seg_overlap_tree(const uint2* __restrict__ w, const uint32_t* __restrict__ cdf,
                                                 const uint32_t* __restrict__ tree, const WsTreeGeom& g,
                                                 const uint2 (&v)[R], uint32_t (&ov)[R]) {
  uint32_t p[2 * R], c[2 * R];
#pragma unroll
  for (int r = 0; r < R; ++r) { p[2 * r] = v[r].x; p[2 * r + 1] = v[r].y; }
  cov_below_tree<2 * R>(w, cdf, tree, g, p, c);
#pragma unroll
  for (int r = 0; r < R; ++r) ov[r] = c[2 * r + 1] - c[2 * r];
}
__device__ __forceinline__ uint32_t seg_overlap_tree1(const uint2* __restrict__ w, const uint32_t* __restrict__ cdf,
                                                      const uint32_t* __restrict__ tree, const WsTreeGeom& g,
                                                      uint32_t s, uint32_t e) {
  const uint2 v[1] = {make_uint2(s, e)};
  uint32_t ov[1];
  seg_overlap_tree<1>(w, cdf, tree, g, v, ov);
  return ov[0];
}

// Overlap of [s, e) with a long normalized list W through its position grid (gat_prep.hip; UnitDev::pgrid_off): entry c = the
// first segment whose end lies beyond c << shift, so no segment in front of entry s >> shift reaches s, and the segments that
// can overlap [s, e) are walked from there -- one or two where the pieces of W are longer than [s, e).  One 4-byte and one or two
// 8-byte accesses, dependent, where the trees read 2 x (four 64-byte nodes + a segment + a running length).
// pg: the grid's entries (behind its header), shift / cells: header words 0 / 1.
__device__ __forceinline__ uint32_t ws_overlap_pgrid(const uint2* __restrict__ w, int n, const uint32_t* __restrict__ pg,
                                                     uint32_t shift, uint32_t cells, uint32_t s, uint32_t e) {
  uint32_t c = s >> shift;
  c = c < cells ? c : cells;
  int j = (int)pg[c];
  uint32_t ov = 0;
  while (j < n) {
    const uint2 x = w[j];
    if (x.x >= e) break;
    const uint32_t lo = s > x.x ? s : x.x, hi = e < x.y ? e : x.y;
    ov += hi > lo ? hi - lo : 0u;
    if (x.y >= e) break;
    ++j;
  }
  return ov;
}

// A unit's workspace held in registers (lane i = workspace segment i), for units with <= 64
// workspace segments: SegmentListSampler's CDF lookup becomes one v_cmp + ballot and the chosen
// segment is fetched with v_readlane -- no memory access in the placement loop.
struct WsRegs {
  uint32_t start, end, cdf;   // lane i: segment i (lanes >= n: start=end=0xffffffff)
  int n;
};
__device__ __forceinline__ WsRegs ws_load(const uint2* __restrict__ w, const uint32_t* __restrict__ cdf, int n, int lane) {
  WsRegs r;
  r.n = n;
  r.start = 0xffffffffu; r.end = 0xffffffffu; r.cdf = 0xffffffffu;
  if (lane < n) { const uint2 v = w[lane]; r.start = v.x; r.end = v.y; r.cdf = cdf[lane]; }
  return r;
}
// bases of [s,e) inside a register-resident workspace: sum over its segments (wave-uniform loop,
// each segment broadcast with v_readlane); exact for normalized lists.
__device__ __forceinline__ uint32_t ws_overlap_regs(const WsRegs& W, uint32_t s, uint32_t e) {
  uint32_t ov = 0;
  for (int j = 0; j < W.n; ++j) {
    const uint32_t ws = (uint32_t)__builtin_amdgcn_readlane((int)W.start, j);
    const uint32_t we = (uint32_t)__builtin_amdgcn_readlane((int)W.end, j);
    const uint32_t lo = s > ws ? s : ws, hi = e < we ? e : we;
    ov += hi > lo ? hi - lo : 0u;
  }
  return ov;
}

// The same by binary search over the lanes (cross-lane reads with a per-lane source: ds_bpermute) instead of a loop over
// all workspace segments: bases of the workspace below a position = the whole segments in front of it (their cumulated
// lengths) + the part of the last one that starts below it.  7 x n instructions become ~60 whatever n is: a unit of an
// isochore-partitioned workspace has dozens of workspace segments (config 3: 31 on chr1), and k_consolidate, which calls
// this once per unit there, is bound by its instruction count.  EVERY lane of the wave must be active at the call (an
// inactive lane's registers read as 0 through ds_bpermute).
__device__ __forceinline__ uint32_t ws_below_search(const WsRegs& W, uint32_t x) {
  int k = 0;                                          // number of workspace segments with start < x
#pragma unroll
  for (int step = 32; step >= 1; step >>= 1) {
    const uint32_t sv = (uint32_t)__builtin_amdgcn_ds_bpermute((k + step - 1) << 2, (int)W.start);
    k = sv < x ? k + step : k;                        // (lanes behind the last segment hold 0xffffffff)
  }
  const int j = k > 0 ? k - 1 : 0;
  const uint32_t ps = (uint32_t)__builtin_amdgcn_ds_bpermute(j << 2, (int)W.start);
  const uint32_t pe = (uint32_t)__builtin_amdgcn_ds_bpermute(j << 2, (int)W.end);
  const uint32_t before = (uint32_t)__builtin_amdgcn_ds_bpermute((k >= 2 ? k - 2 : 0) << 2, (int)W.cdf);
  const uint32_t whole = k >= 2 ? before + 1u : 0u;   // cdf[i] = cumulated length - 1
  return k == 0 ? 0u : whole + (x < pe ? x : pe) - ps;
}
__device__ __forceinline__ uint32_t ws_overlap_search(const WsRegs& W, uint32_t s, uint32_t e) {
  return ws_below_search(W, e) - ws_below_search(W, s);
}

}  // namespace gat
