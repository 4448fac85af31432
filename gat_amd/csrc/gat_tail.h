// gat_tail.h -- SamplerAnnotator.sample behind k_place as lean kernels: the split path of the sampler (k_consolidate,
// k_tail, k_finalize) and its counterpart for long lists behind k_merge_big (k_tail_big, k_resume_big).
//
// k_sampler runs the rest of gat/Engine.pyx:572-646 with one WAVE per (sample, unit): a third of its time is the first
// consolidation -- lane-parallel work over the whole list -- and the rest is the loop's tail: a handful of draws, two or
// three placements, one trim, each paid for with 64-wide passes over the list (where does the new segment go, which
// segment holds base p, shift everything behind it) of which one lane's worth is needed.  Its speed follows the waves a
// CU holds, not the instructions (1.85 / 2.4 / 3.7 / 5.4 ms at 13 / 10 / 6 / 4 waves per CU on config 2), and LDS -- the
// list lives there through all of it -- keeps that number low.  Here the parts are matched to the hardware separately:
//
//   k_consolidate  one wave per (sample, unit): the first consolidation (:582-606) -- sort, merge(0), workspace coverage --
//                  and nothing else; leaves the merged list in the slab with its running lengths (cum).
//   k_tail         one LANE per (sample, unit), like k_place: the tail of the loop as a per-lane state machine whose list
//                  accesses are binary searches in the merged list (L2): new segments that touch nothing become "extras",
//                  the overshoot trim is applied to the list where it is (it is the last thing that happens to it).
//                  Whatever does not fit that shape -- a new segment touching a neighbour, a second trim, more than four
//                  new segments, a long workspace -- is left, untouched, to k_sampler, which resumes from the merged list.
//   k_finalize     one wave per (sample, unit): merged list (trimmed in place by k_tail) + extras, placeholders dropped
//                  (:639-646), the unit's list written where its consumers expect it.  No LDS.  Not run when the
//                  consumer takes (merged list, extras) as they are: k_contig, k_count_seg / k_count_merged<.., PATCH>.
//   k_tail_big     one LANE per (sample, unit) of a LONG list: the placement rounds behind k_merge_big's consolidation;
//                  new segments are logged or united in place with the one segment they touch.
//   k_resume_big   one wave per such unit: the log into the list, the trim(s), the final filter, the list in the slab
//                  (no list in LDS: as many waves per CU as registers allow).
//
// Results are those of k_sampler (and of the reference) bit for bit: every branch below cites the line it restates.
#pragma once
#include <cstddef>
#include "gat_kernels.h"

namespace gat {

constexpr int kTailMaxExtra = 4;      // new segments k_tail keeps aside per unit
// (kTailMaxWs = 64 workspace segments k_tail scans linearly: gat_types.h)
constexpr int kTailMaxWalk = 6;       // segments an overshoot trim may touch
#ifndef GAT_TAIL_LONG_WALK
#define GAT_TAIL_LONG_WALK 96
#endif
constexpr int kTailMaxWalkLong = GAT_TAIL_LONG_WALK;   // ... in a fragmented workspace (k_tail<true>): the placement loop books a segment's overlap with
                                      // the CHOSEN workspace segment only (gat/Engine.pyx:331-343), what it covers of the neighbouring pieces
                                      // shows at the consolidation -- the overshoot, and with it the trim's walk, is dozens of segments there
constexpr int kTailRows = 8;          // random rows fetched at a time

// what k_tail hands on, per (sample, unit) by launch position.  A finished unit (state 1) is: its merged list in the
// slab -- the overshoot trim already applied to it in place, emptied segments left as [0, 0) -- plus these extras.
struct TailPatch {
  int32_t state;          // 0: not handled (k_sampler resumes from the merged list, which is untouched), 1: finished
  int32_t n_extra;
  uint32_t placed, ndraws, nuns, pad;
  uint2 extra[kTailMaxExtra];          // sorted by start, trim applied
  int32_t pos[kTailMaxExtra];          // merged-list elements in front of each
};

static_assert(offsetof(TailPatch, pad) == kPatchPad * 4 && offsetof(TailPatch, nuns) == kPatchNuns * 4, "TailPatch words");
static_assert(sizeof(TailPatch) == kPatchWords * 4 && offsetof(TailPatch, extra) == kPatchExtra * 4 &&
              offsetof(TailPatch, pos) == kPatchPos * 4 && offsetof(TailPatch, placed) == kPatchPlaced * 4 &&
              offsetof(TailPatch, n_extra) == kPatchNExtra * 4 && offsetof(TailPatch, ndraws) == kPatchNdraws * 4,
              "k_contig / k_count_seg read TailPatch records as words");

struct TailArgs {
  SamplerArgs S;
  uint32_t* cum;          // [batch][slab_stride / 8]: inclusive running length of the merged list through every eighth element (and the last)
  TailPatch* patch;       // [launch position][rec_stride] (GAT_REC)
  uint32_t* todo_count;   // units left to k_sampler: k_tail queues them
  uint32_t* todo;
  int32_t no_log_map;     // k_tail_big: every step scans the lane's log (round 3's form; GAT_TB_NO_LOG_MAP)
  int32_t no_bridge;      // k_tail_big: a segment that joins its two neighbours ends the lane's round (round 3's form; GAT_TB_NO_BRIDGE)
  int32_t loose_ok;       // k_resume_big: the lists' only readers are the segment-side count kernels (or k_contig, which
                          // merge(0)s them again): what a trim emptied may stay in the list as [0, 0) -- no compaction pass
};

// ------------------------------------------------------------------------------------------------------------------
// k_consolidate: sort + merge(0) + coverage of what k_place placed (gat/Engine.pyx:582-606, first time round).
template <bool TREE>
__global__ __launch_bounds__(64) void k_consolidate(TailArgs T) {
  extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
  const SamplerArgs& A = T.S;
  const int lane = threadIdx.x;
  const int sidx = blockIdx.x;
  const int a = A.a_base + (int)(blockIdx.y + blockIdx.z * gridDim.y);
  if (a >= A.n_active || (A.a_end > 0 && a >= A.a_end)) return;
  const UnitDev* __restrict__ Up = A.units_o + a;
  const int64_t sa = GAT_REC(A, sidx, a);
  const int4 pre = A.st[sa];
  if (lane == 0) T.patch[sa].state = 0;
  // (launched behind k_merge_big when the problem has long lists: launch positions below n_long carry its verdict)
  if (a < A.n_long) { if ((A.st2[sa].w & 1) == 1) return; }
  else if (lane == 0) A.st2[sa] = make_int4(0, 0, 0, 0);
  const int n = pre.x;
  if (pre.z < 0 || n <= 0 || n > A.lds_cap) return;               // not handed over by k_place / beyond this kernel's LDS: k_sampler's
#ifdef GAT_DIAG_CONS
  // (tools/diag_consolidate.sh: cycles of a unit per phase -- 0 record + workspace, 1 list in registers, 2 sort, 3 merge(0),
  //  4 coverage + write-back; the stamps fence the schedule: shares, not a timing)
  // (a build of its own, without GAT_DIAG: k_place's stamps push its hand-pipelined loops over their registers)
  unsigned long long dg[8] = {0, 0, 0, 0, 0, 0, 0, 0}, dg_t;
#define GAT_CSTAMP(T) { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(T) :: "memory"); }
  GAT_CSTAMP(dg_t);
#define GAT_CPHASE(K) { unsigned long long t__; GAT_CSTAMP(t__); dg[K] += t__ - dg_t; dg_t = t__; }
#else
#define GAT_CPHASE(K)
#endif
  uint32_t* scratch = lds;                                          // bucket-sort scratch
  uint2* seg = reinterpret_cast<uint2*>(lds + kSortScratchWords);
  uint2* out = A.slab + (int64_t)sidx * A.slab_stride + Up->slab_off;
  uint32_t* cum = T.cum + (((int64_t)sidx * A.slab_stride + Up->slab_off) >> 3);      // (one entry per block of eight slab entries)
  const int nws = Up->n_ws;
  const uint2* __restrict__ ws = A.ws + Up->ws_off;
  const uint32_t* __restrict__ ws_cdf = A.ws_cdf + Up->ws_off;
  constexpr int kWsRegMax = 64, kWsLoopMax = 32;
  const WsRegs W = ws_load(ws, ws_cdf, nws < kWsRegMax ? nws : kWsRegMax, lane);
  GAT_CPHASE(0)
  if (n <= kWave) {
    // A list within one round (the units of an isochore problem: ~50 segments) never touches LDS: one load, the sorting
    // network on registers, merge(0) as one scan -- heads by the running maximum of the ends, a head's merged end = the
    // running maximum in front of the next head --, coverage and running lengths on the head lanes, one store.
    const uint2 x = lane < n ? out[lane] : make_uint2(0xffffffffu, 0xffffffffu);
    uint32_t ks[1] = {x.x}, ke[1] = {x.y};
    GAT_CPHASE(1)
#ifdef GAT_SORT64_BPERMUTE
    for (int lk = 1; lk <= 6; ++lk) {
      sort_stage_lanes<1>(ks, ke, (1 << lk) - 1, lk - 1, lane);   // flip
      sort_disperse_below64<1>(ks, ke, lk - 2, lane);
    }
#else
    sort64_by_start(ks[0], ke[0], lane);                          // (partners within a row of 16 lanes by DPP modifiers)
#endif
    GAT_CPHASE(2)
    const bool valid = ks[0] != ke[0];                            // (empty segments are dropped; the padding is empty)
    const int32_t m = wave_incl_max_i32(valid ? (int32_t)ke[0] : INT32_MIN, lane);
    const int32_t excl = __builtin_amdgcn_update_dpp(INT32_MIN, m, 0x138, 0xf, 0xf, false);   // wave_shr:1
    const uint64_t vb = __ballot(valid);
    const bool head = valid && ((vb & lanemask_lt(lane)) == 0 || (int32_t)ks[0] > excl);
    const uint64_t hb = __ballot(head);
    const int pos = __popcll(hb & lanemask_lt(lane));
    const int nU1 = __popcll(hb);
    const uint64_t later = lane < kWave - 1 ? hb >> (lane + 1) : 0ull;
    const int last = later ? lane + __builtin_ctzll(later) : kWave - 1;                        // the lane in front of the next head
    const uint32_t mend = (uint32_t)__builtin_amdgcn_ds_bpermute(last << 2, m);
    const uint32_t s1 = head ? ks[0] : 0u, e1 = head ? mend : 0u;
    GAT_CPHASE(3)
    uint32_t cov1 = 0;
    if (nws <= 8) cov1 = ws_overlap_regs(W, s1, e1);
    else if (nws <= kWsLoopMax) cov1 = ws_overlap_search(W, s1, e1);
    else if constexpr (TREE) {
      // (round 6: the position grid instead of two tree searches per segment)
      const uint32_t* __restrict__ pg1 = A.ws_tree + Up->pgrid_off;
      if (head) cov1 = ws_overlap_pgrid(ws, nws, pg1 + kGridHeader, pg1[0], pg1[1], s1, e1);
    }
    const uint32_t incl1 = wave_incl_sum_u32(e1 - s1, lane);
    if (head) {
      out[pos] = make_uint2(s1, e1);
      if ((pos & 7) == 7 || pos == nU1 - 1) cum[pos >> 3] = incl1;                     // (running lengths per block of eight: GAT_CUM8)
    }
    // (a merged segment that is not wholly inside the unit's workspace may reach into another unit's segment: bit 1 of st2.w,
    //  beside bit 16 of the record's `nuns` (k_tail's, for the segments it added) what k_contig<., true> asks before it looks closer)
    const bool strad1 = __ballot(head && cov1 != e1 - s1) != 0ull;
    cov1 = wave_total_u32(cov1);
    const uint32_t run1 = (uint32_t)__builtin_amdgcn_readlane((int)incl1, kWave - 1);
    if (lane == 0) A.st2[sa] = make_int4(nU1, (int)cov1, (int)run1, strad1 ? 3 : 1);
    GAT_CPHASE(4)
#ifdef GAT_DIAG_CONS
    if (lane == 0 && A.diag != nullptr) for (int k = 0; k < 8; ++k) A.diag[((int64_t)sidx * A.n_units + Up->pad) * 8 + k] = dg[k];
#endif
    return;
  }
  if (n > 1024) {
    if (!wave_sort_bucket_global(seg, out, n, scratch, 512, lane)) {
      for (int i = lane; i < n; i += kWave) seg[i] = out[i];
      wave_sort_auto(seg, n, lane);
    }
  } else if (n > 512) {
    // 513..1 024 segments: the counting sort with the list in registers (one round trip instead of the six of the three
    // passes over the slab; this kernel has the registers: the old k_sampler did not)
    uint2 v[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) { const int i = r * kWave + lane; v[r] = i < n ? out[i] : make_uint2(0u, 0u); }
    if (!wave_sort_bucket_regs<16>(seg, v, n, scratch, 512, lane)) {
#pragma unroll
      for (int r = 0; r < 16; ++r) { const int i = r * kWave + lane; if (i < n) seg[i] = v[r]; }
      wave_sort_auto(seg, n, lane);
    }
  } else {
    // (all loads of the list in flight together: nothing else hides their latency)
    uint2 v[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) { const int i = r * kWave + lane; v[r] = i < n ? out[i] : make_uint2(0u, 0u); }
#pragma unroll
    for (int r = 0; r < 8; ++r) { const int i = r * kWave + lane; if (i < n) seg[i] = v[r]; }
    GAT_CPHASE(1)
    // (round 5, measured and dropped: the bucket sort taking the list from these registers instead of from LDS, and in
    //  straight-line form -- pads instead of `if (i < n)` around every LDS operation, selects instead of branches: the dynamic
    //  instruction count did not fall (selects for branches, one for one), config 2 0.69 -> 0.68 ms, and the arrays that
    //  form keeps alive took the kernel from 54 to 80 registers and k_contig from 99 to 132: config 3 1.52 -> 1.87 and
    //  0.99 -> 1.13 ms)
    wave_sort_fast<8>(seg, n, scratch, lane);
  }
  GAT_CPHASE(2)
  const int nU = wave_merge0(seg, n, lane);
  GAT_CPHASE(3)
  // coverage (intersect(workspace).sum()), total length, running lengths; the merged list goes back to the slab
  uint32_t cov = 0, run = 0;
  bool strad = false;
  const uint32_t* __restrict__ pg = A.ws_tree + (Up->pgrid_off >= 0 ? Up->pgrid_off : 0);
  const uint32_t pshift = TREE ? pg[0] : 0u, pcells = TREE ? pg[1] : 0u;
  for (int base = 0; base < nU; base += kWave) {
    const int i = base + lane;
    uint2 v = make_uint2(0u, 0u);
    if (i < nU) v = seg[i];
    // (all lanes are here: the search's cross-lane reads see every lane's workspace segment; a short workspace is
    //  cheaper by the loop)
    uint32_t cv = 0;
    if (nws <= 8) cv = ws_overlap_regs(W, v.x, v.y);
    else if (nws <= kWsLoopMax) cv = ws_overlap_search(W, v.x, v.y);
    else if constexpr (TREE) { if (i < nU) cv = ws_overlap_pgrid(ws, nws, pg + kGridHeader, pshift, pcells, v.x, v.y); }
    cov += cv;
    strad |= i < nU && cv != v.y - v.x;
    const uint32_t incl = run + wave_incl_sum_u32(v.y - v.x, lane);
    if (i < nU) {
      out[i] = v;
      if ((i & 7) == 7 || i == nU - 1) cum[i >> 3] = incl;
    }
    run = (uint32_t)__builtin_amdgcn_readlane((int)incl, kWave - 1);
  }
  cov = wave_total_u32(cov);
  const bool strad_any = __ballot(strad) != 0ull;
  if (lane == 0) A.st2[sa] = make_int4(nU, (int)cov, (int)run, strad_any ? 3 : 1);
  GAT_CPHASE(4)
#ifdef GAT_DIAG_CONS
  if (lane == 0 && A.diag != nullptr) for (int k = 0; k < 8; ++k) A.diag[((int64_t)sidx * A.n_units + Up->pad) * 8 + k] = dg[k];
#endif
#undef GAT_CPHASE
#undef GAT_CSTAMP
}

// ------------------------------------------------------------------------------------------------------------------
// k_tail: the loop of gat/Engine.pyx:572-635 from behind the first consolidation to its end, one stream per lane.
// Workgroup = one tile of 64 samples x one unit (the tiles of k_place); everything about the unit is wave-uniform.
// The first index i of [0, n) with val(i) > key, val non-decreasing (an upper bound), for a lane by itself.  A binary
// search is log2(n) DEPENDENT probes, each a round trip to the L2 or the HBM; the values searched here -- starts of
// segments placed uniformly over a workspace, running lengths of such a list -- grow nearly linearly with the index, so the
// place of a key between two known values is nearly proportional: two probes at the ends (in flight together), up to four
// interpolated ones, and the bisection of what is left (usually a handful of elements): 9 dependent probes become ~5 for a
// list of 400, 6 become ~3 for one of 50.  The guess only decides where to look, never what is returned.
template <typename Val>
__device__ __forceinline__ int interp_upper_bound(int n, uint32_t key, Val val) {
  if (n <= 0) return 0;
  if (n <= 96) {                                    // (short lists: the bisection's six probes; measured no gain below ~100)
    int lo = 0, hi = n;
    while (lo < hi) {
      const int mid = lo + ((hi - lo) >> 1);
      if (val(mid) > key) hi = mid; else lo = mid + 1;
    }
    return lo;
  }
  int L = 0, H = n - 1;                             // val(L) <= key < val(H) once past the two tests below
  uint32_t vL = val(0), vH = val(n - 1);
  if (vL > key) return 0;
  if (!(vH > key)) return n;
#pragma unroll 1
  for (int it = 0; it < 4 && H - L > 4; ++it) {
    const float t = (float)(key - vL) / (float)(vH - vL);
    int mid = L + 1 + (int)(t * (float)(H - L - 1));
    mid = mid < L + 1 ? L + 1 : (mid > H - 1 ? H - 1 : mid);
    const uint32_t v = val(mid);
    if (v > key) { H = mid; vH = v; } else { L = mid; vL = v; }
  }
  int lo = L + 1, hi = H;                           // the answer is in [L + 1, H]
  while (lo < hi) {
    const int mid = lo + ((hi - lo) >> 1);
    if (val(mid) > key) hi = mid; else lo = mid + 1;
  }
  return lo;
}

struct TailRng {
  const uint32_t* rp;     // the lane's column of the tile's rows
  uint32_t used, rows;    // raw outputs consumed / generated
  uint32_t buf[kTailRows];
  uint32_t have;          // buf holds rows [base, base + have)
  uint32_t base;
  bool out_of_rows;
};
__device__ __forceinline__ void tail_fetch(TailRng& r) {
  r.base = r.used;
#pragma unroll
  for (int k = 0; k < kTailRows; ++k) r.buf[k] = r.used + (uint32_t)k < r.rows ? r.rp[(size_t)(r.used + (uint32_t)k) * kWave] : 0u;
  r.have = kTailRows;
}
__device__ __forceinline__ uint32_t tail_next(TailRng& r) {
  if (r.used >= r.rows) { r.out_of_rows = true; return 0u; }
  if (r.used - r.base >= r.have) tail_fetch(r);
  const uint32_t k = r.used - r.base;
  uint32_t x = r.buf[0];
#pragma unroll
  for (int q = 1; q < kTailRows; ++q) x = k == (uint32_t)q ? r.buf[q] : x;
  r.used++;
  return x;
}
// numpy's masked rejection (gat_device.h rng_range), per lane
__device__ __forceinline__ uint32_t tail_range(TailRng& r, uint32_t range) {
  if (range == 0) return 0;
  const uint32_t mask = 0xffffffffu >> __builtin_clz(range);
  uint32_t v;
  do { v = tail_next(r) & mask; } while (v > range && !r.out_of_rows);
  return v;
}

// LONGWS (round 6): units of more than kTailMaxWs workspace segments are taken too -- the position draw's segment through the
// tree over the cumulated lengths (a handful of draws per unit), the overlaps through the position grid (ws_overlap_pgrid); the
// instantiation without it is what problems of short workspaces run (such units are k_sampler's there)
template <bool LONGWS>
__global__ __launch_bounds__(64) void k_tail(TailArgs T) {
  __shared__ uint32_t l_ws[3 * kTailMaxWs];         // starts, ends, cdf of the unit's workspace
  const SamplerArgs& A = T.S;
  const int lane = threadIdx.x;
  const int sb = blockIdx.x, a = (int)(blockIdx.y + blockIdx.z * gridDim.y);
  if (a >= A.n_active) return;
  const UnitDev* __restrict__ Up = A.units_o + a;
  const int nws = Up->n_ws;
  const bool longws = LONGWS && nws > kTailMaxWs;    // (wave-uniform)
  if (nws > kTailMaxWs && !longws) {                 // long workspace: left to k_sampler (search trees)
    const int sx = sb * kWave + lane;
    if (sx < A.batch) T.todo[atomicAdd(T.todo_count, 1u)] = (uint32_t)sx * (uint32_t)A.n_active + (uint32_t)a;
    return;
  }
  const uint2* __restrict__ wsg = A.ws + Up->ws_off;
  const uint32_t* __restrict__ pgh = A.ws_tree + (longws ? Up->pgrid_off : 0);
  const uint32_t pshift = longws ? pgh[0] : 0u, pcells = longws ? pgh[1] : 0u;
  const uint32_t* __restrict__ tree_cdf = A.ws_tree + (longws ? Up->tree_cdf_off : 0);
  const WsTreeGeom G = ws_tree_geom(nws);
  const uint32_t hist_total = Up->hist_total, bucket = Up->bucket, ws_total = Up->ws_total;
  const int32_t ltotal = Up->ltotal;
  const int cap = Up->slab_cap;
  const uint32_t* __restrict__ rank_len = A.rank_len + Up->rank_off;
  if (!longws)
    for (int i = lane; i < nws; i += kWave) {
      const uint2 w = A.ws[Up->ws_off + i];
      l_ws[i] = w.x; l_ws[kTailMaxWs + i] = w.y; l_ws[2 * kTailMaxWs + i] = A.ws_cdf[Up->ws_off + i];
    }
  __syncthreads();
  const int sidx = sb * kWave + lane;
  if (sidx >= A.batch) return;
  const int64_t sa = GAT_REC(A, sidx, a);
  const int4 pre = A.st[sa];
  const int4 c2 = A.st2[sa];
  const uint32_t qe = (uint32_t)sidx * (uint32_t)A.n_active + (uint32_t)a;
  // (every unit that is not finished here goes to k_sampler's queue)
  if (pre.z < 0 || (c2.w & 1) != 1 || c2.x <= 0) { T.todo[atomicAdd(T.todo_count, 1u)] = qe; return; }   // not consolidated
  uint2* U = A.slab + (int64_t)sidx * A.slab_stride + Up->slab_off;     // (read; the trim is written into it at the very end)
  const uint32_t* __restrict__ cum = T.cum + (((int64_t)sidx * A.slab_stride + Up->slab_off) >> 3);
  const int nU = c2.x;
  uint32_t cov = (uint32_t)c2.y, total = (uint32_t)c2.z;

  TailRng rng;
  rng.rows = (uint32_t)A.rng_rows[a];
  rng.rp = A.rng_out + A.rng_off[a] + (int64_t)sb * rng.rows * kWave + lane;
  rng.used = (uint32_t)pre.w;
  rng.out_of_rows = false;
  rng.have = 0; rng.base = rng.used;

  // bases of [s, e) inside the workspace (SegmentList.intersect(workspace).sum() of one segment)
  auto ws_overlap = [&](uint32_t s, uint32_t e) -> uint32_t {
    if constexpr (LONGWS) { if (longws) return ws_overlap_pgrid(wsg, nws, pgh + kGridHeader, pshift, pcells, s, e); }
    uint32_t ov = 0;
    for (int j = 0; j < nws; ++j) {
      const uint32_t ws0 = l_ws[j], we0 = l_ws[kTailMaxWs + j];
      const uint32_t lo = s > ws0 ? s : ws0, hi = e < we0 ? e : we0;
      ov += hi > lo ? hi - lo : 0u;
    }
    return ov;
  };

  uint2 ex[kTailMaxExtra];
  int epos[kTailMaxExtra];
#pragma unroll
  for (int j = 0; j < kTailMaxExtra; ++j) { ex[j] = make_uint2(0xffffffffu, 0xffffffffu); epos[j] = 0x7fffffff; }
  int nE = 0;                  // extras in place (sorted, with their position in the merged list)
  uint2 pend[kTailMaxExtra];   // placed since the last consolidation, in placement order
  int nP = 0;
  uint32_t placed = (uint32_t)pre.x;
  int32_t length = pre.z;
  int32_t remaining = ltotal - (int32_t)cov, true_remaining = ltotal;
  int nuns = 0;
  if (true_remaining == remaining) nuns++; else true_remaining = remaining;           // :601-605
  bool done = !(true_remaining != 0 && nuns < 20);
  bool bail = false;
  // The overshoot trims.  One in a workspace that is a contig: the trim takes exactly the overshoot off the coverage.  In a
  // fragmented workspace (LONGWS) trim_ends takes `size` bases off SEGMENTS wherever they lie (gat/SegmentList.pyx:545-597), the
  // ones in gaps do not lower the coverage, and the loop trims again -- smaller amounts each time, only trims behind a trim (the
  // coverage never falls below the target).  Up to kMaxTrims of them are kept as records and applied at the end: a trim's touched
  // elements are a contiguous range [t_lo, t_hi] of the list (full ones emptied, one partly), a later position draw maps through
  // the earlier ranges (t_before: the running length in front of a range, t_s: the bases it lost, t_rem: what is left of its
  // partly trimmed element), and a later trim that would start in or walk into an earlier range -- or any that wraps round the
  // list's end -- leaves the unit to k_sampler.
  constexpr int kMaxTrims = LONGWS ? 3 : 1;
  uint32_t t_flags[kMaxTrims], t_part[kMaxTrims], t_before[kMaxTrims], t_s[kMaxTrims], t_rem[kMaxTrims];
  int t_v0[kMaxTrims], t_full[kMaxTrims], t_lo[kMaxTrims], t_hi[kMaxTrims];
#pragma unroll
  for (int t = 0; t < kMaxTrims; ++t) { t_flags[t] = 0u; t_part[t] = 0u; t_before[t] = 0u; t_s[t] = 0u; t_rem[t] = 0u; t_v0[t] = 0; t_full[t] = 0; t_lo[t] = 0; t_hi[t] = -1; }
  int nT = 0;
  bool wrapped = false;
  uint32_t drop_len = 0;
  bool strad_extra = false;    // a new segment that is not wholly inside the unit's workspace (TailPatch::pad, bit 16)

  // element v of the list with the extras in place
  auto vget = [&](int v) -> uint2 {
    int c = 0;
    uint2 r = make_uint2(0u, 0u);
    bool is_extra = false;
#pragma unroll
    for (int j = 0; j < kTailMaxExtra; ++j) {
      if (j < nE) {
        const int vj = epos[j] + j;
        if (vj < v) c++;
        if (vj == v) { r = ex[j]; is_extra = true; }
      }
    }
    return is_extra ? r : U[v - c];
  };

  for (int step = 0; step < 48 && !done && !bail; ++step) {
    // ---- overshoot: trim (:608-626) -- only as the last thing that happens to the unit
    if (true_remaining < 0) {
      if (nT >= kMaxTrims || wrapped) { bail = true; break; }        // one trim more than is kept: k_sampler's
      const uint32_t p = tail_range(rng, total - 1u);
      // p counts bases of the list AS IT IS NOW; the running lengths at hand are the list's before any trim.  Earlier trims in
      // list order: in front of a range nothing has changed; inside it only the partly trimmed element is left (t_rem bases);
      // behind it everything has moved down by the range's t_s
      uint32_t key = p;
      bool direct = false;
      if constexpr (kMaxTrims > 1) {
        int o0 = 0, o1 = 1;
        if (nT == 2 && t_lo[1] < t_lo[0]) { o0 = 1; o1 = 0; }
        uint32_t offset = 0;
        bool stop = false;                                             // (p lies in front of the range at hand: nothing behind it matters)
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          const int t = q == 0 ? o0 : o1;
          if (q < nT && !direct && !stop) {
            uint32_t tb = 0, tr = 0, ts = 0;
#pragma unroll
            for (int u = 0; u < kMaxTrims; ++u) if (u == t) { tb = t_before[u]; tr = t_rem[u]; ts = t_s[u]; }
            const uint32_t rb = tb - offset;
            if (p < rb) stop = true;
            else if (p < rb + tr) direct = true;
            else offset += ts;
          }
        }
        key = p + offset;
      }
      if (direct) { bail = true; break; }                              // it starts in what an earlier trim left of a segment
      // leftmost element whose inclusive running length exceeds key (searchsorted over cdf = incl - 1)
      // ((int32_t)(c - 1 - key) >= 0 is c > key: running lengths stay below 2^31)
      // The running lengths are kept per BLOCK of eight merged segments (cum[b] = through element 8 b + 7, the last entry through
      // the last element): an eighth of what k_consolidate used to write, and the search below touches the two lines that hold
      // a list's block sums instead of a line per probe; the block found, its eight segments -- one 64-byte piece of the list
      // -- are walked.  val(i) = merged running length through i + the extras in front of or at i, non-decreasing in i, so the
      // first block whose last element exceeds key holds the answer.
      const int nB = (nU + 7) >> 3;
      const int bl = interp_upper_bound(nB, key, [&](int b) -> uint32_t {
        const int last = b * 8 + 7 < nU - 1 ? b * 8 + 7 : nU - 1;
        uint32_t c = cum[b];
#pragma unroll
        for (int j = 0; j < kTailMaxExtra; ++j) if (j < nE && epos[j] <= last) c += ex[j].y - ex[j].x;
        return c;
      });
      int lo = nU;
      uint32_t before = nB > 0 && bl > 0 ? cum[(bl < nB ? bl : nB) - 1] : 0u;      // merged running length in front of element lo
      if (bl < nB) {
        uint2 blk[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) blk[q] = bl * 8 + q < nU ? U[bl * 8 + q] : make_uint2(0u, 0u);
        uint32_t runm = before;
        bool hit = false;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const int i = bl * 8 + q;
          if (!hit && i < nU) {
            uint32_t c = runm + (blk[q].y - blk[q].x);
#pragma unroll
            for (int j = 0; j < kTailMaxExtra; ++j) if (j < nE && epos[j] <= i) c += ex[j].y - ex[j].x;
            if (c > key) { hit = true; lo = i; before = runm; }
            else runm += blk[q].y - blk[q].x;
          }
        }
        if (!hit) before = runm;
      }
      // extras standing right in front of merged-list element lo come first
      int nbefore = 0;
#pragma unroll
      for (int j = 0; j < kTailMaxExtra; ++j) if (j < nE && epos[j] < lo) { before += ex[j].y - ex[j].x; nbefore++; }
      int v = lo + nbefore;                                          // (every extra in front of it counted)
      bool found = false;
      uint32_t before_v = before;                                    // the running length in front of element v (before any trim)
#pragma unroll
      for (int j = 0; j < kTailMaxExtra; ++j) {
        if (j < nE && epos[j] == lo && !found) {
          before_v = before;
          before += ex[j].y - ex[j].x;
          if ((int32_t)(before - 1u - key) >= 0) found = true; else { v++; before_v = before; }
        }
      }
      const int nV = nU + nE;
      if (v >= nV) { bail = true; break; }                           // (cannot happen: p < total)
      if constexpr (kMaxTrims > 1) {
#pragma unroll
        for (int t = 0; t < kMaxTrims; ++t) if (t < nT && v >= t_lo[t] && v <= t_hi[t]) bail = true;   // (cannot happen either)
        if (bail) break;
      }
      const uint2 chosen = vget(v);
      (void)tail_range(rng, chosen.y - 1u - chosen.x);               // position inside the segment: only its index matters
      const uint32_t forward = tail_range(rng, 1u);                  // numpy.random.randint(0, 2)
      int32_t s = -true_remaining;
      if (rng.out_of_rows) break;
      if (!((uint64_t)total > (uint64_t)(uint32_t)s)) { bail = true; break; }   // gat/SegmentList.pyx:560 asserts: k_sampler reports it
      // trim_ends(pos, s, forward) (gat/SegmentList.pyx:545-597)
      uint32_t removed = 0;
      int idx = v, full = 0;
      uint32_t part = 0, part_len = 0, walked_behind = 0;            // (walked_behind: the lengths of the elements walked behind the first)
      bool wraps = false;
      for (int w = 0; s > 0; ++w) {
        if (w >= (longws ? kTailMaxWalkLong : kTailMaxWalk)) { bail = true; break; }
        if constexpr (kMaxTrims > 1) {
#pragma unroll
          for (int t = 0; t < kMaxTrims; ++t) if (t < nT && idx >= t_lo[t] && idx <= t_hi[t]) bail = true;   // into an earlier trim's range
          if (bail) break;
        }
        const uint2 x = vget(idx);
        const int32_t l = (int32_t)x.y - (int32_t)x.x;
        if (w > 0) walked_behind += (uint32_t)l;
        uint32_t ra, rb;
        if (l < s) { s -= l; ra = x.x; rb = x.y; full++; }
        else {
          part = (uint32_t)s; part_len = (uint32_t)l;
          if (forward) { ra = x.x; rb = x.x + (uint32_t)s; } else { ra = (uint32_t)((int32_t)x.y - s); rb = x.y; }
          s = 0;
        }
        if (rb > ra) removed += ws_overlap(ra, rb);
        if (forward) { idx++; if (idx == nV) { idx = 0; wraps = s > 0; } } else { idx--; if (idx < 0) { idx = nV - 1; wraps = s > 0; } }
      }
      if (bail) break;
      if (wraps) { if (nT > 0) { bail = true; break; } wrapped = true; }   // (a walk round the list's end: kept only as the one trim)
      const int touched = full + (part > 0 ? 1 : 0);
#pragma unroll
      for (int t = 0; t < kMaxTrims; ++t) {
        if (t == nT) {
          t_flags[t] = 1u | (forward ? 2u : 0u);
          t_v0[t] = v; t_full[t] = full; t_part[t] = part;
          t_lo[t] = forward ? v : v - touched + 1;
          t_hi[t] = forward ? v + touched - 1 : v;
          t_s[t] = (uint32_t)(-true_remaining);
          t_rem[t] = part > 0 ? part_len - part : 0u;
          t_before[t] = forward ? before_v : before_v - walked_behind;
        }
      }
      if (part > 0) {
        // the final filter(workspace) (:644) can only drop what this trim leaves of the partly trimmed segment: every
        // other segment is a union of placed segments, each of which overlaps its workspace segment (:331-343)
        const int last = forward ? (idx == 0 ? nV - 1 : idx - 1) : (idx == nV - 1 ? 0 : idx + 1);
        const uint2 x = vget(last);
        const uint32_t ks = forward ? x.x + part : x.x, ke = forward ? x.y : x.y - part;
        if (!(ke > ks && ws_overlap(ks, ke) > 0)) {
#pragma unroll
          for (int t = 0; t < kMaxTrims; ++t) if (t == nT) t_flags[t] |= 4u;
          drop_len += ke > ks ? ke - ks : 0u;
        }
      }
      nT++;
      cov -= removed;
      total -= (uint32_t)(-true_remaining);
      true_remaining = 1;
      // back to the top of the loop: hs.sample(), then remaining (still negative) <= length: a consolidation with
      // nothing new (coverage known)
    } else {
      // ---- sls.sample(length) (:279-343)
      const uint32_t p = tail_range(rng, ws_total - 1u);
      int k = 0;
      uint32_t cs, ce;
      int32_t pe = INT32_MIN;
      bool lw = false;
      if constexpr (LONGWS) lw = longws;
      if (lw) {
        const uint32_t tp[1] = {p};
        int kk[1];
        ws_tree_count<true, 1>(tree_cdf, G, tp, kk);                                  // searchsorted + cmpPosition
        k = kk[0] < nws ? kk[0] : nws - 1;
        const uint4 w4 = A.ws_rec[Up->ws_off + k];
        cs = w4.x; ce = w4.y; pe = (int32_t)w4.z;
      } else {
        for (int j = 0; j < nws; ++j) k += ((int32_t)(l_ws[2 * kTailMaxWs + j] - p) < 0) ? 1 : 0;   // searchsorted + cmpPosition
        k = k < nws ? k : nws - 1;
        cs = l_ws[k]; ce = l_ws[kTailMaxWs + k];
        if (k > 0) pe = (int32_t)l_ws[kTailMaxWs + k - 1];
      }
      int32_t sampling_start = (int32_t)cs - length + 1;
      sampling_start = pe > sampling_start ? pe : sampling_start;
      const uint32_t range = ce - 1u - (uint32_t)sampling_start;
      const int32_t q = sampling_start + (int32_t)tail_range(rng, range);
      if (rng.out_of_rows) break;
      const uint32_t start = (uint32_t)(q > 0 ? q : 0), end = (uint32_t)(q + length);
      const int32_t omin = (int32_t)ce < (int32_t)end ? (int32_t)ce : (int32_t)end;
      const int32_t omax = (int32_t)cs > (int32_t)start ? (int32_t)cs : (int32_t)start;
      const int32_t overlap = omin - omax > 0 ? omin - omax : 0;
      // (true_remaining > 0 here: the loop runs only then)
      if (nE + nP >= kTailMaxExtra || nU + nE + nP >= cap) { bail = true; break; }
#pragma unroll
      for (int j = 0; j < kTailMaxExtra; ++j) if (j == nP) pend[j] = make_uint2(start, end);
      nP++;
      placed++;
      remaining -= overlap;
    }
    // ---- hs.sample() (:413-435)
    {
      uint32_t r = 1;
      if (hist_total > 1) r = 1u + tail_range(rng, hist_total - 2u);
      uint32_t len_u = rank_len[r] * bucket;
      if (bucket > 1) len_u += tail_range(rng, bucket - 1u);
      length = (int32_t)len_u;
      if (rng.out_of_rows) break;
    }
    // ---- consolidate (:582-606)
    if (remaining <= length) {
      // the new segments join the merged list if none of them is empty or touches anything (merge(0) joins at
      // start <= previous end); otherwise the general consolidation is needed: k_sampler's
      for (int q = 0; q < nP && !bail; ++q) {
        uint2 x = make_uint2(0u, 0u);
#pragma unroll
        for (int j = 0; j < kTailMaxExtra; ++j) if (j == q) x = pend[j];
        if (x.x == x.y) { bail = true; break; }
        // merged-list elements with start <= x.start
        const int lo = interp_upper_bound(nU, x.x, [&](int i) -> uint32_t { return U[i].x; });
        if (lo > 0 && (int32_t)x.x <= (int32_t)U[lo - 1].y) { bail = true; break; }
        if (lo < nU && (int32_t)U[lo].x <= (int32_t)x.y) { bail = true; break; }
#pragma unroll
        for (int j = 0; j < kTailMaxExtra; ++j) {
          if (j < nE) {
            const bool x_first = x.x < ex[j].x;
            const uint2 f = x_first ? x : ex[j], g = x_first ? ex[j] : x;
            if ((int32_t)g.x <= (int32_t)f.y) bail = true;
          }
        }
        if (bail) break;
        // into the sorted extras
        int at = 0;
#pragma unroll
        for (int j = 0; j < kTailMaxExtra; ++j) if (j < nE && ex[j].x < x.x) at++;
#pragma unroll
        for (int j = kTailMaxExtra - 1; j > 0; --j) if (j > at && j <= nE) { ex[j] = ex[j - 1]; epos[j] = epos[j - 1]; }
#pragma unroll
        for (int j = 0; j < kTailMaxExtra; ++j) if (j == at) { ex[j] = x; epos[j] = lo; }
        nE++;
        {
          const uint32_t ovx = ws_overlap(x.x, x.y);
          cov += ovx;
          strad_extra |= ovx != x.y - x.x;
        }
        total += x.y - x.x;
      }
      if (bail) break;
      nP = 0;
      remaining = ltotal - (int32_t)cov;
      if (true_remaining == remaining) nuns++; else true_remaining = remaining;
      if (!(true_remaining != 0 && nuns < 20)) done = true;
    }
  }
  // (result.sum() > 0 is asserted at the end, gat/Engine.pyx:645: a unit that would fail is k_sampler's to report)
  if (total - drop_len == 0u) bail = true;
  if (!done || bail || rng.out_of_rows) {                // patch.state stays 0: k_sampler resumes from the merged list
    T.todo[atomicAdd(T.todo_count, 1u)] = qe;
    return;
  }
#pragma unroll
  for (int t = 0; t < kMaxTrims; ++t) {
    if (t >= nT) continue;
    const uint32_t trim = t_flags[t], trim_part = t_part[t];
    const int trim_v0 = t_v0[t], trim_full = t_full[t];
   {
    // the unit is finished here, the trim was the last thing that happened to its list (a placement behind a trim cannot
    // be: the trim leaves remaining <= 0): apply it where the segments are -- trim_ends walked from v0
    // (gat/SegmentList.pyx:567-596): `full` segments emptied, `part` bases off the next, which is dropped altogether
    // when what is left of it lies outside the workspace (the final filter, :644)
    const int nV = nU + nE;
    int v = trim_v0;
    for (int d = 0; d <= trim_full; ++d) {
      if (d == trim_full && trim_part == 0) break;
      int c = 0, which = -1;
#pragma unroll
      for (int j = 0; j < kTailMaxExtra; ++j) {
        if (j < nE) { const int vj = epos[j] + j; if (vj < v) c++; if (vj == v) which = j; }
      }
      uint2 y = make_uint2(0u, 0u);
      if (d == trim_full && !(trim & 4u)) {
        y = which >= 0 ? make_uint2(0u, 0u) : U[v - c];
#pragma unroll
        for (int j = 0; j < kTailMaxExtra; ++j) if (which == j) y = ex[j];
        if (trim & 2u) y.x += trim_part; else y.y -= trim_part;
      }
      if (which >= 0) {
#pragma unroll
        for (int j = 0; j < kTailMaxExtra; ++j) if (which == j) ex[j] = y;
      } else U[v - c] = y;
      if (trim & 2u) { v++; if (v == nV) v = 0; } else { v--; if (v < 0) v = nV - 1; }
    }
  }
  }
  TailPatch* P = T.patch + sa;
  P->n_extra = nE;
  // (bit 16 of nuns: a new segment reaches out of the unit's workspace -- beside bit 1 of st2.w, k_consolidate's for the merged list:
  //  what k_contig<., true> asks before it looks at the unit's segments.  A store of its own for it cost the kernel 7 %)
  P->placed = placed; P->ndraws = rng.used; P->nuns = (uint32_t)nuns | (strad_extra ? 0x10000u : 0u);
#pragma unroll
  for (int j = 0; j < kTailMaxExtra; ++j) { P->extra[j] = ex[j]; P->pos[j] = epos[j]; }
  P->state = 1;
  // (the unit's statistics: its consumers may take the record as it is, without k_finalize)
  *reinterpret_cast<uint4*>(A.ws_stat + GAT_REC(A, sidx, Up->pad) * 4) = make_uint4(placed, rng.used, (uint32_t)nuns, 0u);
}

// ------------------------------------------------------------------------------------------------------------------
// k_tail_big: the placement rounds behind the first consolidation of a LONG list (k_merge_big's), one stream per lane.
// Such a list needs ~n x coverage / 2 more segments (60 for 4 000 segments covering 3 % of their workspace), placed in
// three to five rounds, and k_sampler paid every round with a handful of passes over the whole list by one wave (config-4
// shape: 33 of 87 ms at 1.4 waves per CU).  Here a round costs a binary search per new segment:
//   * a new segment that touches nothing is logged (from the end of the unit's slab region backwards),
//   * one that touches exactly one segment (of the merged list, or a logged one) is united with it in place
//     (merge(0) joins at start <= previous end, gat/SegmentList.pyx:756-816; the union touches nothing else, so the list
//     stays what merge(0) of everything would give), coverage and total length move by the difference,
//   * anything else -- an empty segment, one touching two neighbours -- ends the lane's work: the segments of the round
//     not yet applied are handed on as "sampled since the last consolidation", which is what they are.
// The lane stops at the consolidation after which the loop would trim or end (:601-626) -- with that consolidation's
// bookkeeping still to do -- and k_sampler resumes there: merged list + log (inserted in one pass per 64) + loop state.
constexpr int kLogMapWords = 16;     // k_tail_big: a lane's map of where its logged segments stand, 512 cells over the unit's span
__global__ __launch_bounds__(64) void k_tail_big(TailArgs T) {
  __shared__ uint32_t l_ws[3 * kTailMaxWs];
  __shared__ uint32_t l_map[kLogMapWords * kWave];
  const SamplerArgs& A = T.S;
  const int lane = threadIdx.x;
  const int sb = blockIdx.x, a = (int)(blockIdx.y + blockIdx.z * gridDim.y);
  if (a >= A.n_long) return;                           // (only the launch positions k_merge_big was given)
  const UnitDev* __restrict__ Up = A.units_o + a;
  const int nws = Up->n_ws;
  const int sidx = sb * kWave + lane;
  const int64_t sa = GAT_REC(A, sidx < A.batch ? sidx : 0, a);
  TailPatch* P = T.patch + sa;
  if (sidx < A.batch) P->state = 0;
  if (nws > kTailMaxWs) return;                        // long workspace: k_sampler's (search trees)
  const uint32_t hist_total = Up->hist_total, bucket = Up->bucket, ws_total = Up->ws_total;
  const int32_t ltotal = Up->ltotal;
  const int cap = Up->slab_cap;
  const uint32_t* __restrict__ rank_len = A.rank_len + Up->rank_off;
  for (int i = lane; i < nws; i += kWave) {
    const uint2 w = A.ws[Up->ws_off + i];
    l_ws[i] = w.x; l_ws[kTailMaxWs + i] = w.y; l_ws[2 * kTailMaxWs + i] = A.ws_cdf[Up->ws_off + i];
  }
  __syncthreads();
  if (sidx >= A.batch) return;
  const int4 pre = A.st[sa];
  const int4 c2 = A.st2[sa];
  if (pre.z < 0 || (c2.w & 1) != 1 || c2.x <= 0) return;     // not consolidated by k_merge_big: k_sampler's as before
  uint2* U = A.slab + (int64_t)sidx * A.slab_stride + Up->slab_off;
  int nU = c2.x;
  uint32_t cov = (uint32_t)c2.y, total = (uint32_t)c2.z;

  TailRng rng;
  rng.rows = (uint32_t)A.rng_rows[a];
  rng.rp = A.rng_out + A.rng_off[a] + (int64_t)sb * rng.rows * kWave + lane;
  rng.used = (uint32_t)pre.w;
  rng.out_of_rows = false;
  rng.have = 0; rng.base = rng.used;

  auto ws_overlap = [&](uint32_t s, uint32_t e) -> uint32_t {
    uint32_t ov = 0;
    for (int j = 0; j < nws; ++j) {
      const uint32_t ws0 = l_ws[j], we0 = l_ws[kTailMaxWs + j];
      const uint32_t lo = s > ws0 ? s : ws0, hi = e < we0 ? e : we0;
      ov += hi > lo ? hi - lo : 0u;
    }
    return ov;
  };
  // merge(0) joins b to a (a.start <= b.start) when b.start <= a.end
  auto touches = [](uint2 p, uint2 q) -> bool {
    const bool p_first = p.x <= q.x;
    const uint2 f = p_first ? p : q, g = p_first ? q : p;
    return (int32_t)g.x <= (int32_t)f.y;
  };

  // Where the lane's logged segments stand (round 6): a bit per cell of 1/512 of the unit's span, set for the cells a logged
  // segment covers.  A new segment whose cells are all clear touches none of them and skips the scan of the log -- the kernel is
  // bound by the rate of its gathers (a lane's 8-byte loads are requests of their own), and the scan was 30 of a step's ~50; with
  // ~30 segments logged one step in twelve still scans.  Cells are monotone in the position, so two closed intervals that share
  // a point share a cell; a segment over more than four cells (or a log entry that long) scans as before.
  uint32_t map_lo, map_shift;
  {
    const uint32_t max_len = rank_len[hist_total] * bucket + bucket;
    const uint32_t w0 = l_ws[0], w1 = l_ws[kTailMaxWs + nws - 1];
    map_lo = w0 > max_len ? w0 - max_len : 0u;
    const uint32_t span = w1 + max_len - map_lo;
    const int bits = 32 - __builtin_clz(span | 1u);
    map_shift = bits > 9 ? (uint32_t)(bits - 9) : 0u;
  }
#pragma unroll
  for (int w = 0; w < kLogMapWords; ++w) l_map[w * kWave + lane] = 0u;
  bool map_all = T.no_log_map != 0;                                 // a logged segment too long for the map: every step scans
  auto map_cell = [&](uint32_t p) -> uint32_t {
    const uint32_t c = (p > map_lo ? p - map_lo : 0u) >> map_shift;
    return c < (uint32_t)(kLogMapWords * 32 - 1) ? c : (uint32_t)(kLogMapWords * 32 - 1);
  };
  auto map_mark = [&](uint32_t a, uint32_t b) {
    const uint32_t c0 = map_cell(a), c1 = map_cell(b);
    if (c1 - c0 >= 4u) { map_all = true; return; }
    for (uint32_t c = c0; c <= c1; ++c) l_map[(c >> 5) * kWave + lane] |= 1u << (c & 31u);
  };
  auto map_maybe = [&](uint32_t a, uint32_t b) -> bool {
    const uint32_t c0 = map_cell(a), c1 = map_cell(b);
    if (map_all || c1 - c0 >= 4u) return true;
    bool any = false;
    for (uint32_t c = c0; c <= c1; ++c) any = any || ((l_map[(c >> 5) * kWave + lane] >> (c & 31u)) & 1u) != 0u;
    return any;
  };
  int nE = 0, nP = 0;          // logged segments U[cap - 1 - j]; placements of this round NOT applied: U[nU + j]
  bool broken = false;         // a placement of this round could not be applied: the rest of the round is only recorded
  uint32_t placed = (uint32_t)pre.x;
  int32_t length = pre.z;
  int32_t remaining = ltotal - (int32_t)cov, true_remaining = ltotal;
  int nuns = 0;
  if (true_remaining == remaining) nuns++; else true_remaining = remaining;           // :601-605, first consolidation
  if (!(true_remaining > 0 && nuns < 20)) return;      // a trim or the end right away: k_sampler resumes as before (state 0)

  bool handed = false;
  for (int step = 0; step < 4096; ++step) {
    // ---- sls.sample(length) (:279-343), as in k_tail
    if (nU + nP + nE + 2 > cap) { atomicOr(A.flags, kStatusOverflow); return; }
    const uint32_t p = tail_range(rng, ws_total - 1u);
    int k = 0;
    for (int j = 0; j < nws; ++j) k += ((int32_t)(l_ws[2 * kTailMaxWs + j] - p) < 0) ? 1 : 0;
    k = k < nws ? k : nws - 1;
    const uint32_t cs = l_ws[k], ce = l_ws[kTailMaxWs + k];
    int32_t sampling_start = (int32_t)cs - length + 1;
    if (k > 0) { const int32_t pe = (int32_t)l_ws[kTailMaxWs + k - 1]; sampling_start = pe > sampling_start ? pe : sampling_start; }
    const uint32_t range = ce - 1u - (uint32_t)sampling_start;
    const int32_t q = sampling_start + (int32_t)tail_range(rng, range);
    if (rng.out_of_rows) break;
    const uint2 x = make_uint2((uint32_t)(q > 0 ? q : 0), (uint32_t)(q + length));
    const int32_t omin = (int32_t)ce < (int32_t)x.y ? (int32_t)ce : (int32_t)x.y;
    const int32_t omax = (int32_t)cs > (int32_t)x.x ? (int32_t)cs : (int32_t)x.x;
    const int32_t overlap = omin - omax > 0 ? omin - omax : 0;
    placed++;
    remaining -= overlap;
    // The segment goes into the structure at once (every lane does the same work in every step; at the consolidation
    // only the bookkeeping is left): where it stands in the merged list, what it touches there and in the log
    bool applied = false;
#ifdef GAT_DBG_QUEUE
    int dbg_why = 10;
#endif
    if (!broken && x.x != x.y) {
      // merged-list elements with start <= x.start
#ifdef GAT_EXP_TB_NOSEARCH
      const int lo = (int)(((uint64_t)x.x * (uint64_t)nU) >> 32) % (nU > 0 ? nU : 1);      // (timing experiment: wrong results)
#else
      const int lo = interp_upper_bound(nU, x.x, [&](int i) -> uint32_t { return U[i].x; });
#endif
      const uint2 pv = lo > 0 ? U[lo - 1] : make_uint2(0u, 0u), nv = lo < nU ? U[lo] : make_uint2(0u, 0u);
      const uint2 nn = lo + 1 < nU ? U[lo + 1] : make_uint2(0xffffffffu, 0xffffffffu);
      const bool tl = lo > 0 && (int32_t)x.x <= (int32_t)pv.y;
      const bool tr = lo < nU && (int32_t)nv.x <= (int32_t)x.y;
      const bool tr2 = tr && lo + 1 < nU && (int32_t)nn.x <= (int32_t)x.y;
      int nt = 0, tj = 0;                                            // logged segments it touches
#ifndef GAT_TB_LOGCHUNK
#define GAT_TB_LOGCHUNK 4
#endif
      constexpr int kLC = GAT_TB_LOGCHUNK;                           // logged segments looked at per round trip
#ifdef GAT_EXP_TB_NOLOGSCAN
      const int nE_scan = 0;                                         // (timing experiment: wrong results)
#else
      const int nE_scan = map_maybe(x.x, x.y) ? nE : 0;
#endif
      for (int j0 = 0; j0 < nE_scan; j0 += kLC) {
        uint2 e[kLC];
#pragma unroll
        for (int r = 0; r < kLC; ++r) e[r] = j0 + r < nE ? U[cap - 1 - (j0 + r)] : make_uint2(0xffffffffu, 0xffffffffu);
#pragma unroll
        for (int r = 0; r < kLC; ++r) if (j0 + r < nE && touches(e[r], x)) { nt++; tj = j0 + r; }
      }
      // (a neighbour that is the placeholder of an earlier bridge says nothing about what stands there: left to the round's end)
      const bool gone = (lo > 0 && pv.x == pv.y) || (lo < nU && nv.x == nv.y);
#ifdef GAT_DBG_QUEUE
      dbg_why = gone ? 11 : (tl && tr) ? 12 : tr2 ? 13 : nt >= 2 ? 14 : 15;
#endif
      if (!gone && tl && tr && !tr2 && nt == 0 && !(T.no_bridge & 1)) {
        // a bridge (round 6): it touches both neighbours and nothing else -- the three are one segment where the left one stands,
        // and the right one stays as an EMPTY segment at its own start: the list keeps its order and its length, merge(0), the
        // trim's running lengths, the final filter and the count kernels all pass over an empty segment (these were 0.9 % of
        // the config-4 shape's units, each a wave of k_sampler's with the list in LDS: 1.4 ms of the step)
        const uint32_t ye = (int32_t)pv.y > (int32_t)x.y ? pv.y : x.y;
        const uint2 u = make_uint2(pv.x, (int32_t)nv.y > (int32_t)ye ? nv.y : ye);
        U[lo - 1] = u;
        U[lo] = make_uint2(nv.x, nv.x);
        cov += ws_overlap(u.x, u.y) - ws_overlap(pv.x, pv.y) - ws_overlap(nv.x, nv.y);
        total += (u.y - u.x) - (pv.y - pv.x) - (nv.y - nv.x);
        applied = true;
      } else if (!gone && !tl && tr2 && nt == 0 && !(T.no_bridge & 2) && nn.x != nn.y) {
        // ... and the same on the right: it starts in front of the next segment, covers it and reaches the one behind (as
        // frequent as the bridge where short segments are common) -- unless it reaches a third
        const uint2 n3 = lo + 2 < nU ? U[lo + 2] : make_uint2(0xffffffffu, 0xffffffffu);
        const uint32_t ye = (int32_t)nv.y > (int32_t)x.y ? nv.y : x.y;
        const uint2 u = make_uint2(x.x, (int32_t)nn.y > (int32_t)ye ? nn.y : ye);
        if (!(lo + 2 < nU && (int32_t)n3.x <= (int32_t)u.y)) {
          U[lo] = u;
          U[lo + 1] = make_uint2(nn.x, nn.x);
          cov += ws_overlap(u.x, u.y) - ws_overlap(nv.x, nv.y) - ws_overlap(nn.x, nn.y);
          total += (u.y - u.x) - (nv.y - nv.x) - (nn.y - nn.x);
          applied = true;
        }
      } else if (!gone && !(tl && tr) && !tr2 && nt <= 1 && !(nt == 1 && (tl || tr))) {
        if (tl || tr || nt == 1) {
          const int at = tl ? lo - 1 : (tr ? lo : cap - 1 - tj);
          const uint2 o = U[at];
          const uint2 u = make_uint2(o.x < x.x ? o.x : x.x, (int32_t)o.y > (int32_t)x.y ? o.y : x.y);
          U[at] = u;
          if (!tl && !tr) map_mark(u.x, u.y);                         // (a logged segment grew)
          cov += ws_overlap(u.x, u.y) - ws_overlap(o.x, o.y);
          total += (u.y - u.x) - (o.y - o.x);
        } else {
          U[cap - 1 - nE] = x;
          nE++;
          map_mark(x.x, x.y);
          cov += ws_overlap(x.x, x.y);
          total += x.y - x.x;
        }
        applied = true;
      }
    }
#ifdef GAT_DBG_QUEUE
    if (!applied && !broken) atomicAdd(reinterpret_cast<unsigned long long*>(A.stat) + dbg_why, 1ull);
#endif
    if (!applied) { broken = true; U[nU + nP] = x; nP++; }
    // ---- hs.sample() (:413-435)
    {
      uint32_t r = 1;
      if (hist_total > 1) r = 1u + tail_range(rng, hist_total - 2u);
      uint32_t len_u = rank_len[r] * bucket;
      if (bucket > 1) len_u += tail_range(rng, bucket - 1u);
      length = (int32_t)len_u;
      if (rng.out_of_rows) break;
    }
    // ---- consolidate (:582-606)
    if (remaining <= length) {
      const int32_t r_new = ltotal - (int32_t)cov;
      const bool same = true_remaining == r_new;
      const int32_t tr_new = same ? true_remaining : r_new;
      const int nuns_new = nuns + (same ? 1 : 0);
      if (broken || !(tr_new > 0 && nuns_new < 20)) {
        // hand on at this consolidation: its bookkeeping is k_sampler's (it redoes it from true_remaining / nuns as they
        // were), the segments not applied are its "sampled" ones
        handed = true;
        break;
      }
      true_remaining = tr_new; nuns = nuns_new; remaining = r_new;
    }
  }
  const int hand_nS = nP, hand_at = nU;
  if (!handed) { P->state = 3; return; }               // rows ran out (or the step limit): the unit is redone from its seed
  A.st2[sa] = make_int4(nU, (int)cov, (int)total, 1);
  int32_t* R = reinterpret_cast<int32_t*>(P);
  R[kPatchNExtra] = nE;
  R[kPatchPlaced] = (int32_t)placed;
  R[kPatchNdraws] = (int32_t)rng.used;
  R[kPatchNuns] = nuns;
  R[kTbTrueRemaining] = true_remaining;
  R[kTbNSampled] = hand_nS;
  R[kTbSampledAt] = hand_at;
  R[kTbPending] = length;
  R[kTbRemaining] = remaining;
  R[kPatchState] = 2;
}

// ------------------------------------------------------------------------------------------------------------------
// k_resume_big: what is left of a long list's unit behind k_tail_big -- the log inserted, the consolidation's bookkeeping,
// the overshoot trim(s), the final filter -- one wave per unit with the list where it is, in the slab.  k_sampler does the
// same with the list in LDS, which for lists of thousands means one or two waves per CU and every 64-element round of
// every pass an exposed latency (9.5 of 44 ms on the config-4 shape).  Here every pass runs eight rounds of loads at a
// time and the occupancy is what registers allow.  Whatever does not fit (segments k_tail_big could not apply, a log
// beyond 128 entries, workspaces beyond the register loop, rows running out) is left to k_sampler as before (record state
// 2), or handed to it for a redo from the seed (state 3).  A finished unit is marked state 1.
//
// VIRT (round 6; the lists' only reader takes its segments one by one, in any order: k_count_merged without contig lists): the
// log is NOT inserted.  Its entries touch nothing, so the list as a set is already what merge(0) of everything gives; only the
// trim cares for the order (a position in the running lengths of the sorted list, then a walk over neighbours).  The sorted log
// goes right behind the merged list, out[nU .. nU + nE), and the trim works on VIRTUAL indices: v -> the log entry r where
// vpos[r] = (merged elements with start <= its start) + r equals v, the merged element v - #{r: vpos[r] < v} otherwise
// (vpos sorted, in LDS).  The insertion was a read and a write of the whole list -- 20 of this kernel's 25 GB on the config-4 shape.
template <bool VIRT>
__global__ __launch_bounds__(64) void k_resume_big(TailArgs T) {
  __shared__ int32_t l_cs[2 * kWave];
  const SamplerArgs& A = T.S;
  const int lane = threadIdx.x;
  const int sidx = blockIdx.x, a = (int)(blockIdx.y + blockIdx.z * gridDim.y);
  if (a >= A.n_long) return;
  const UnitDev* __restrict__ Up = A.units_o + a;
  const int64_t sa = GAT_REC(A, sidx, a);
  int32_t* R = reinterpret_cast<int32_t*>(T.patch + sa);
  if (R[kPatchState] != 2) return;
  const int nE = R[kPatchNExtra];
  const int nws = Up->n_ws;
  constexpr int kWsLoopMax = 32;
  if (R[kTbNSampled] > 0 || nE > 2 * kWave || nws > kWsLoopMax) return;
  const int4 c2 = A.st2[sa];
  int nU = c2.x;
  uint32_t cov = (uint32_t)c2.y, total = (uint32_t)c2.z;
  const int32_t ltotal = Up->ltotal;
  int32_t true_remaining = R[kTbTrueRemaining];
  int nuns = R[kPatchNuns];
  {
    // the bookkeeping of the consolidation k_tail_big stopped at (:601-605)
    const int32_t remaining = ltotal - (int32_t)cov;
    if (true_remaining == remaining) nuns++; else true_remaining = remaining;
  }
  if (true_remaining > 0 && nuns < 20) return;          // (k_tail_big goes on in that case: cannot be; k_sampler's anyway)
  const int u = Up->pad;
  const int cap = Up->slab_cap;
  const uint32_t hist_total = Up->hist_total, bucket = Up->bucket;
  const uint32_t* __restrict__ rank_len = A.rank_len + Up->rank_off;
  const WsRegs W = ws_load(A.ws + Up->ws_off, A.ws_cdf + Up->ws_off, nws, lane);
  uint2* out = A.slab + (int64_t)sidx * A.slab_stride + Up->slab_off;

  // ---- the log (nE <= 128 segments that touch nothing, unsorted) into the merged list, in place from the back
  int vp0 = 0x7fffffff, vp1 = 0x7fffffff, nL = 0;         // VIRT: virtual positions of the sorted log entries lane, 64 + lane; their number
  if (nE > 0) {
    uint2 e[2];
    int cu[2], rk[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int j = h * kWave + lane;
      e[h] = j < nE ? out[cap - 1 - j] : make_uint2(0xffffffffu, 0xffffffffu);
    }
#pragma unroll
    for (int h = 0; h < 2; ++h) {                                   // list elements with start <= the new segment's
      int lo = 0, hi = (h * kWave + lane < nE) ? nU : 0;
      while (lo < hi) { const int mid = lo + ((hi - lo) >> 1); if (out[mid].x <= e[h].x) lo = mid + 1; else hi = mid; }
      cu[h] = lo;
      rk[h] = 0;
    }
    for (int j = 0; j < kWave; ++j) {                               // rank among the new ones: by start, then by log index
#pragma unroll
      for (int g = 0; g < 2; ++g) {
        const uint32_t sj = (uint32_t)__builtin_amdgcn_readlane((int)e[g].x, j);
        const int jj = g * kWave + j;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int me = h * kWave + lane;
          rk[h] += (jj < nE && (sj < e[h].x || (sj == e[h].x && jj < me))) ? 1 : 0;
        }
      }
    }
    wave_sync<true>();
#pragma unroll
    for (int h = 0; h < 2; ++h) if (h * kWave + lane < nE) l_cs[rk[h]] = VIRT ? cu[h] + rk[h] : cu[h];
    wave_sync<true>();
    if constexpr (VIRT) {
      // (every entry is in a register by now -- the searches above compared it --: the sorted run may overlap the log's slots)
#pragma unroll
      for (int h = 0; h < 2; ++h) if (h * kWave + lane < nE) out[nU + rk[h]] = e[h];
      vp0 = lane < nE ? l_cs[lane] : 0x7fffffff;
      vp1 = kWave + lane < nE ? l_cs[kWave + lane] : 0x7fffffff;
      nL = nE;
      wave_sync<true>();
    } else {
    const int cs0 = lane < nE ? l_cs[lane] : 0x7fffffff, cs1 = kWave + lane < nE ? l_cs[kWave + lane] : 0x7fffffff;
    constexpr int kB = 8;
    for (int top = ((nU - 1) >> 6) << 6; top >= 0; top -= kB * kWave) {
      uint2 v[kB];
#pragma unroll
      for (int q = 0; q < kB; ++q) { const int i = top - q * kWave + lane; v[q] = (i >= 0 && i < nU && top - q * kWave >= 0) ? out[i] : make_uint2(0u, 0u); }
      int sh[kB];
      bool nothing_below = false;
#pragma unroll
      for (int q = 0; q < kB; ++q) {
        const int base = top - q * kWave;
        const int i = base + lane;
        const int below = __popcll(__ballot(cs0 < base + 1)) + __popcll(__ballot(cs1 < base + 1));
        const int upto = __popcll(__ballot(cs0 <= base + kWave - 1)) + __popcll(__ballot(cs1 <= base + kWave - 1));
        int s2 = below;
        for (int j = below; j < upto; ++j) s2 += l_cs[j] <= i ? 1 : 0;
        sh[q] = (base >= 0 && i < nU) ? s2 : 0;
        if (base >= 0 && upto == 0) nothing_below = true;           // nothing goes in at or before this block
      }
#pragma unroll
      for (int q = 0; q < kB; ++q) { const int i = top - q * kWave + lane; if (sh[q] > 0) out[i + sh[q]] = v[q]; }
      if (nothing_below) break;
    }
#pragma unroll
    for (int h = 0; h < 2; ++h) if (h * kWave + lane < nE) out[cu[h] + rk[h]] = e[h];
    nU += nE;
    wave_sync<true>();                                                    // (the list is read again below: stores done)
    }
  }
  const int nV = nU + nL;                                  // elements of the (virtual) sorted list
  // VIRT: where the 64 virtual indices [base, base + 64) stand in the slab (all lanes call it together)
  auto phys_block = [&](int base) -> int {
    const int v = base + lane;
    if (!VIRT || nL == 0) return v;
    const int below = __popcll(__ballot(vp0 < base)) + __popcll(__ballot(vp1 < base));
    const int upto = __popcll(__ballot(vp0 < base + kWave)) + __popcll(__ballot(vp1 < base + kWave));
    int c = below, r = -1;
    for (int j = below; j < upto; ++j) { const int vp = l_cs[j]; c += vp < v ? 1 : 0; if (vp == v) r = j; }
    return r >= 0 ? nU + r : v - c;
  };
  // ... and one virtual index (a single lane's walk)
  auto phys_one = [&](int v) -> int {
    if (!VIRT || nL == 0) return v;
    int lo = 0, hi = nL;
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (l_cs[mid] < v) lo = mid + 1; else hi = mid; }
    return (lo < nL && l_cs[lo] == v) ? nU + lo : v - lo;
  };

  WaveRng rng;
  rng.mt = nullptr;
  rng.can_switch = false; rng.seed = 0u;               // (no LDS words for a generator here: out of rows = redo, state 3)
  rng.use_pre = true; rng.exhausted = false; rng.ndraws = (uint32_t)R[kPatchNdraws]; rng.pos = 0; rng.rbuf = 0;
  rng.pre_rows = (uint32_t)A.rng_rows[a];
  rng.pre_j = rng.ndraws;
  rng.pre_base = rng.pre_j - (uint32_t)kWave;
  rng.pre = A.rng_out + A.rng_off[a] + (int64_t)(sidx >> 6) * rng.pre_rows * kWave + (sidx & 63);
  bool redo = false, trim_assert = false;
  constexpr int kMaxPartial = 4;
  int part_idx[kMaxPartial] = {-1, -1, -1, -1};                     // the segments trims left partly standing
  int n_part = 0;
  while (true_remaining != 0 && nuns < 20) {
    if (true_remaining > 0) { redo = true; break; }                 // (a trim leaves remaining <= 0: cannot be)
    // ---- overshoot: trim (:608-626), as in k_sampler, the list in the slab
    const uint32_t p = rng_range(rng, total - 1u, lane);
    int k = 0;
    {
      uint32_t run = 0;
      bool found = false;
      constexpr int kB = 8;
      for (int base0 = 0; base0 < nV && !found; base0 += kB * kWave) {
        uint2 v[kB];
#pragma unroll
        for (int q = 0; q < kB; ++q) {
          const int i = base0 + q * kWave + lane;
          const int ph = phys_block(base0 + q * kWave);
          v[q] = i < nV ? out[ph] : make_uint2(0u, 0u);
        }
#pragma unroll
        for (int q = 0; q < kB; ++q) {
          if (found) continue;
          const int i = base0 + q * kWave + lane;
          const uint32_t incl = run + wave_incl_sum_u32(v[q].y - v[q].x, lane);
          const bool ge = i < nV && (int32_t)(incl - 1u - p) >= 0;  // cdf[i] = incl - 1; leftmost i with (int)(cdf[i] - p) >= 0
          const uint64_t b = __ballot(ge);
          if (b != 0) { k = base0 + q * kWave + (int)__builtin_ctzll(b); found = true; }
          run = (uint32_t)__builtin_amdgcn_readlane((int)incl, kWave - 1);
        }
      }
    }
    const uint2 chosen = out[phys_one(k)];
    (void)rng_range(rng, chosen.y - 1u - chosen.x, lane);           // position inside the segment: only its index matters
    const uint32_t forward = rng_range(rng, 1u, lane);              // numpy.random.randint(0, 2)
    int32_t s = -true_remaining;
    if (rng.exhausted) { redo = true; break; }
    if (!((uint64_t)total > (uint64_t)(uint32_t)s)) { trim_assert = true; break; }
    uint32_t removed = 0;                                           // workspace bases taken away by the trim (lane 0)
    int partial = -1;                                               // ... and the segment it left partly standing
    if (lane == 0) {
      int idx = k;
      while (s > 0) {
        const int at = phys_one(idx);
        const uint2 v = out[at];
        const int32_t l = (int32_t)v.y - (int32_t)v.x;
        uint32_t ra, rb;
        if (l < s) { out[at] = make_uint2(0u, 0u); s -= l; ra = v.x; rb = v.y; }
        else {
          if (forward) { out[at] = make_uint2(v.x + (uint32_t)s, v.y); ra = v.x; rb = v.x + (uint32_t)s; }
          else { out[at] = make_uint2(v.x, (uint32_t)((int32_t)v.y - s)); ra = (uint32_t)((int32_t)v.y - s); rb = v.y; }
          s = 0;
          partial = at;
        }
        if (rb > ra) removed += ws_overlap_regs(W, ra, rb);
        if (forward) { idx++; if (idx == nV) idx = 0; }
        else { idx--; if (idx < 0) idx = nV - 1; }
      }
    }
    partial = __builtin_amdgcn_readfirstlane(partial);
    if (partial >= 0) {
#pragma unroll
      for (int j = 0; j < kMaxPartial; ++j) if (j == n_part) part_idx[j] = partial;
      n_part++;
    }
    cov -= (uint32_t)__builtin_amdgcn_readfirstlane((int)removed);
    total -= (uint32_t)(-true_remaining);
    wave_sync<true>();
    true_remaining = 1;
    // ---- hs.sample() (:413-435) and the consolidation with nothing new (:582-606)
    {
      uint32_t r = 1;
      if (hist_total > 1) r = 1u + rng_range(rng, hist_total - 2u, lane);
      if (bucket > 1) (void)rng_range(rng, bucket - 1u, lane);
      (void)rank_len; (void)r;                                      // (any length >= 1 >= remaining (<= 0): only the draws count)
      if (rng.exhausted) { redo = true; break; }
    }
    const int32_t remaining = ltotal - (int32_t)cov;
    if (true_remaining == remaining) nuns++; else true_remaining = remaining;
  }
  if (redo) { if (lane == 0) R[kPatchState] = 3; return; }
  const int64_t so = (int64_t)sidx * A.n_units + u;
  if (trim_assert) {
    if (lane == 0) { A.unit_n[so] = 0; atomicOr(A.flags, kStatusTrimAssert); R[kPatchState] = 1; }
    return;
  }
  // ---- result = unintersected.merge(0).filter(workspace) (:639-646): placeholders and segments outside the workspace
  // dropped, compacted in place (forward: a round's output never passes its input)
  int nout = 0;
  uint32_t tsum = 0;
  if (T.loose_ok && n_part <= kMaxPartial) {
    // The readers of this list skip [0, 0): no compaction pass (a read and a write of the whole list, a third of this
    // kernel's traffic -- it is bound by the HBM: 45 GB per 12 500 samples of the config-4 shape at 5.6 TB/s).  What the
    // filter would drop besides the emptied segments can only be what a trim left of a partly trimmed one (every other
    // segment is a union of placed segments, each overlapping its workspace segment, :331-343): those are looked at here
    uint32_t gone = 0;
    if (lane == 0) {
#pragma unroll
      for (int j = 0; j < kMaxPartial; ++j) {
        if (j < n_part) {
          const uint2 v = out[part_idx[j]];
          if (v.x != v.y && ws_overlap_regs(W, v.x, v.y) == 0) { out[part_idx[j]] = make_uint2(0u, 0u); gone += v.y - v.x; }
        }
      }
    }
    nout = nV;
    tsum = total - (uint32_t)__builtin_amdgcn_readfirstlane((int)gone);
  } else {
    // (VIRT: the slab's order -- the merged list, then the sorted log -- is as good as any for this list's reader)
    constexpr int kB = 8;
    for (int base0 = 0; base0 < nV; base0 += kB * kWave) {
      uint2 v[kB];
      bool keep[kB];
#pragma unroll
      for (int q = 0; q < kB; ++q) { const int i = base0 + q * kWave + lane; v[q] = i < nV ? out[i] : make_uint2(0u, 0u); }
#pragma unroll
      for (int q = 0; q < kB; ++q) keep[q] = base0 + q * kWave + lane < nV && ws_overlap_regs(W, v[q].x, v[q].y) > 0;
#pragma unroll
      for (int q = 0; q < kB; ++q) {
        const uint64_t b = __ballot(keep[q]);
        if (keep[q]) { out[nout + __popcll(b & lanemask_lt(lane))] = v[q]; tsum += v[q].y - v[q].x; }
        nout += __popcll(b);
      }
    }
    tsum = wave_total_u32(tsum);
  }
  if (lane == 0) {
    A.unit_n[so] = tsum > 0 ? nout : 0;
    if (!(tsum > 0)) atomicOr(A.flags, kStatusAssert);
    *reinterpret_cast<uint4*>(A.ws_stat + GAT_REC(A, sidx, u) * 4) = make_uint4((uint32_t)R[kPatchPlaced], rng.ndraws, (uint32_t)nuns, 0u);
    R[kPatchState] = 1;
  }
}

// ------------------------------------------------------------------------------------------------------------------
// k_queue_rest: long lists -- every (sample, launch position) k_resume_big did not finish (and the units k_merge_big was not
// given) into k_sampler's queue, entries sidx * n_active + launch position as k_tail's
__global__ __launch_bounds__(256) void k_queue_rest(TailArgs T, int n_act) {
  const SamplerArgs& A = T.S;
  const int lane = threadIdx.x & 63;
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x, tot = (int64_t)A.batch * n_act;
  bool push = false;
  uint32_t e = 0;
  if (i < tot) {
    const int sidx = (int)(i % A.batch), a = (int)(i / A.batch);
    push = a >= A.n_long || T.patch[GAT_REC(A, sidx, a)].state != 1;
    e = (uint32_t)sidx * (uint32_t)n_act + (uint32_t)a;
#ifdef GAT_DBG_QUEUE
    // (why a unit is queued: words 10..15 of the statistics -- 10 k_merge_big declined, 11 k_tail_big left at the first
    //  consolidation, 12 a round broken, 13 a log beyond 128, 14 another hand-over k_resume_big left, 15 rows ran out)
    if (push && a < A.n_long) {
      const int32_t* R = reinterpret_cast<const int32_t*>(T.patch + GAT_REC(A, sidx, a));
      const int st = R[kPatchState];
      int why = 14;
      if (st == 0) why = (A.st2[GAT_REC(A, sidx, a)].w & 1) != 1 ? 10 : 11;
      else if (st == 3) why = 15;
      else if (R[kTbNSampled] > 0) why = 12;
      else if (R[kPatchNExtra] > 2 * kWave) why = 13;
      (void)why;   // (the histogram is k_tail_big's, of what broke the round)
    }
#endif
  }
  const uint64_t b = __ballot(push);
  if (b == 0) return;
  uint32_t base = 0;
  if (lane == 0) base = atomicAdd(T.todo_count, (uint32_t)__popcll(b));
  base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
  if (push) T.todo[base + (uint32_t)__popcll(b & lanemask_lt(lane))] = e;
}

// ------------------------------------------------------------------------------------------------------------------
// k_finalize: result = unintersected.merge(0).filter(workspace) (gat/Engine.pyx:639-646) from the merged list and
// k_tail's record, written OUT OF PLACE into the second slab (the extras shift what is behind them: in place every
// element would have to be held until everything in front of it is written).  Units k_tail left alone are queued for
// k_sampler, which writes its final lists to the same slab.
__global__ __launch_bounds__(64) void k_finalize(TailArgs T) {
  const SamplerArgs& A = T.S;
  const int lane = threadIdx.x;
  const int sidx = blockIdx.x;
  const int a = (int)(blockIdx.y + blockIdx.z * gridDim.y);
  if (a >= A.n_active) return;
  const int64_t sa = GAT_REC(A, sidx, a);
  const TailPatch* __restrict__ P = T.patch + sa;
  if (P->state != 1) return;                             // (k_tail queued it for k_sampler)
  const UnitDev* __restrict__ Up = A.units_o + a;
  const int u = Up->pad;
  const uint2* __restrict__ src = A.slab + (int64_t)sidx * A.slab_stride + Up->slab_off;
  uint2* __restrict__ dst = A.slab_final + (int64_t)sidx * A.slab_stride + Up->slab_off;
  const int nU = A.st2[sa].x, nE = P->n_extra, nV = nU + nE;
  uint2 ex[kTailMaxExtra];
  int vj[kTailMaxExtra];
#pragma unroll
  for (int j = 0; j < kTailMaxExtra; ++j) { ex[j] = P->extra[j]; vj[j] = j < nE ? P->pos[j] + j : 0x7fffffff; }
  int nout = 0;
  uint32_t total = 0;
  constexpr int kB = 4;                                    // rounds whose loads are in flight together
  for (int base = 0; base < nV; base += kB * kWave) {
    uint2 x[kB];
#pragma unroll
    for (int q = 0; q < kB; ++q) {
      const int v = base + q * kWave + lane;
      int c = 0, which = -1;
#pragma unroll
      for (int j = 0; j < kTailMaxExtra; ++j) { if (vj[j] < v) c++; if (vj[j] == v) which = j; }
      x[q] = make_uint2(0u, 0u);
      if (v < nV) {
        if (which >= 0) {
#pragma unroll
          for (int j = 0; j < kTailMaxExtra; ++j) if (which == j) x[q] = ex[j];
        } else x[q] = src[v - c];
      }
    }
#pragma unroll
    for (int q = 0; q < kB; ++q) {
      if (base + q * kWave >= nV) break;
      const int v = base + q * kWave + lane;
      const uint2 y = x[q];
      // merge(0) drops the placeholders and what the trim emptied (nothing touches); filter(workspace) could only drop
      // the partly trimmed segment, which k_tail has looked at (every other one is a union of placed segments, each
      // overlapping its workspace)
      const bool keep = v < nV && y.x != y.y;
      const uint64_t b = __ballot(keep);
      if (keep) { dst[nout + __popcll(b & lanemask_lt(lane))] = y; total += y.y - y.x; }
      nout += __popcll(b);
    }
  }
  total = wave_total_u32(total);
  if (lane == 0) {
    const int64_t so = (int64_t)sidx * A.n_units + u;
    A.unit_n[so] = nout;
    if (!(total > 0)) atomicOr(A.flags, kStatusAssert);
    *reinterpret_cast<uint4*>(A.ws_stat + GAT_REC(A, sidx, u) * 4) = make_uint4(P->placed, P->ndraws, P->nuns & 0xffffu, 0u);
  }
}

}  // namespace gat
