// gat_kernels.h -- HIP kernels of the GAT hot path (gfx950).
//
//   k_sampler : one wave per (sample, isochore unit): SamplerAnnotator.sample
//               (gat/Engine.pyx:515-646) with HistogramSampler (:387-435) and
//               SegmentListSampler (:245-348), consolidation = sort + merge(0) + intersect/sum
//               (gat/SegmentList.pyx:478, :756, :1469, :1607), overshoot trim (:545-597),
//               final merge(0) + filter (:1401).
//   k_contig  : one wave per (sample, contig): IntervalDictionary.fromIsochores
//               (gat/Engine.pyx:2857-2876): concat the contig's units, sort, merge(0).
//   k_count_seg / k_count_anno : Counter*.__call__ (gat/Engine.pyx:1417-1472) summed over
//               contigs as gat/__init__.py:578-587 does.
#pragma once
#include "gat_device.h"

namespace gat {

// (UnitDev and the kStatus* bits: gat_types.h)

struct SamplerArgs {
  const UnitDev* units;
  const int32_t* order;       // active unit ids, largest first
  const UnitDev* units_o;     // units[order[a]] with the unit id in `pad`: one load instead of two dependent ones
  int32_t n_units;
  int32_t batch;              // samples in this batch
  int32_t rec_stride;         // samples the scratch was sized for: stride of the [unit][sample] hand-over records (GAT_REC)
  const uint2* ws;
  const uint32_t* ws_cdf;
  const uint32_t* rank_len;
  const uint32_t* ws_tree;    // 16-ary search trees of the long workspaces (UnitDev::tree_*_off) and their grids (pgrid_off, cgrid_off)
  const uint4* ws_rec;        // per workspace segment {start, end, previous end, cdf} (k_place_grid)
  int4* st2;                  // [launch position][rec_stride]: first consolidation done by k_merge_big
                              // {merged segments, workspace bases covered, sum of lengths, 1}, .w == 0 otherwise
  int32_t n_long;             // launch positions [0, n_long) were given to k_merge_big
  int32_t n_active;           // active units; their launch position is blockIdx.y + blockIdx.z * gridDim.y (grid y, z <= 65535)
  uint32_t seed;
  int64_t sample_begin;       // global id of sample 0 of this batch
  uint2* slab;                // [batch][slab_stride]
  int64_t slab_stride;
  int32_t* unit_n;            // [batch][n_units]
  int32_t* flags;             // OR of kStatus*
  unsigned long long* stat;   // [0]=placed [1]=draws [2]=unsuccessful rounds [3]=output segments [4]=full-mode units
  uint32_t* ws_stat;          // [unit][rec_stride][4]: the same per work unit (summed by k_reduce_stats; one
                              // atomic per work unit on a single line costs more than the sampling itself)
  // lane-parallel front end (k_rng + k_place); all null/0 when the sampler runs stand-alone
  const int64_t* rng_off;     // per active index: word offset of the unit's first tile in rng_out
  const int32_t* rng_rows;    // per active index: rows (raw outputs per stream) generated
  uint32_t* rng_out;          // tile (active a, sample block sb): rng_out[rng_off[a] + (sb*rows + j)*64 + lane]
  uint32_t* rng_ckpt;         // k_seed -> k_rng: word 39 w of every stream's seeded state, [tile = a * n_blocks + sb][w < 16][lane]
  int4* st;                   // [launch position][rec_stride] hand-off from k_place, one 16-byte record per work unit:
                              //   x = segments placed before the first consolidation, y = `remaining` at that point,
                              //   z = the pending length (>0), -1: run the unit in full, -2: SamplerSegments complete,
                              //   w = raw outputs consumed so far
  int32_t sampler_kind;       // 0 SamplerAnnotator, 1 SamplerSegments (st_length -2: unit complete after k_place)
  int32_t place_plain_step;   // != 0: k_place runs GAT_STEP_SIMPLE_B where GAT_STEP_SIMPLE_C would do (GAT_PLACE_NO_CM: tests, A/B)
  int32_t big_buckets;        // > 0: LDS holds that many + 1 scratch words behind the segment buffer (units > 1024 segments)
  int32_t lds_cap;            // segment capacity of the LDS buffer
  int32_t a_base, a_end;      // k_sampler / k_merge_big / k_consolidate: the launch covers launch positions [a_base, a_end) -- one
                              // launch per size class, so that the dynamic LDS of a launch fits ITS longest list (a_end 0: all)
  uint32_t* cum;              // split path: inclusive running lengths of the merged lists, parallel to the slab (k_merge_big fills it too)
  uint2* slab_final;          // split path: where the units' FINAL lists go (a second slab: k_finalize writes out of place)
  const int32_t* skip;        // split path: skip[GAT_REC(a, sidx) * skip_stride] != 0: the unit was finished by k_tail / k_finalize
  int32_t skip_stride;
  const uint32_t* todo_count; // split path: k_sampler works off the queue of units k_tail left alone (k_finalize fills it)
  const uint32_t* todo;       //   entries sidx * n_active + launch position
  const int32_t* tb;          // long lists: k_tail_big's hand-over records (TailPatch words, skip_stride apart), or nullptr
  // the reference's own stream (k_serial): ONE MT19937 state for the whole run, 624 words + the position
  uint32_t* serial_state;
  const int32_t* unit_pos;    // unit id -> launch position, -1: inactive (k_serial walks the units in the reference's order)
  unsigned long long* diag;   // diagnostic build (-DGAT_DIAG) only: [work unit][8] shader cycles per phase of k_sampler
  unsigned long long* diag_place;   // ... and [launch position][8]: k_place's cycles per phase of its loop, summed over the unit's tiles
  unsigned long long* diag_tiles;   // ... and [launch position][sample block][2]: when a tile of k_place began and ended (s_memrealtime, 100 MHz)
};

// The hand-over records between the sampler's kernels -- st, st2, TailPatch by launch position, ws_stat by unit id -- are laid
// out [unit][sample]: the 64 lanes of a tile of the lane-per-stream kernels (k_place, k_tail, k_tail_big: one unit, 64
// consecutive samples) touch consecutive records -- eight lines for a wave's 16-byte records where [sample][unit] made every
// lane's access a line of its own (k_tail on config 3: 2.4 GB fetched per 10 000 samples, ten lines per lane of which four
// were records).  The stride is the number of samples the scratch was sized for, the same in every batch of a call.
#define GAT_REC(A_, sidx_, a_) ((int64_t)(a_) * (int64_t)(A_).rec_stride + (int64_t)(sidx_))

// layout of a TailPatch record in 32-bit words (gat_tail.h static_asserts it)
constexpr int kPatchState = 0, kPatchNExtra = 1, kPatchPlaced = 2, kPatchNdraws = 3, kPatchNuns = 4, kPatchPad = 5, kPatchExtra = 6,
              kPatchPos = 14, kPatchWords = 18;
// k_tail_big's hand-over (state 2) keeps its loop state in the words a finished unit's extras would take
constexpr int kTbTrueRemaining = kPatchPad, kTbNSampled = kPatchPos, kTbSampledAt = kPatchPos + 1, kTbPending = kPatchPos + 2,
              kTbRemaining = kPatchPos + 3;

// In-kernel stamps of the diagnostic build (tools/diag_sampler.sh): shares of a work unit's life per phase.  Stamps
// fence the schedule, so that build's run time is not quoted; the product build compiles none of this.
#ifdef GAT_DIAG
#define GAT_STAMP(T) { asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(T) :: "memory"); }
#define GAT_PHASE(K) { unsigned long long t__; GAT_STAMP(t__); dg[K] += t__ - dg_t; dg_t = t__; }
// k_place's loop (tools/diag_place.sh): 0 row wait (the hand-pipelined loads' s_waitcnt + takes, or the compiler's), 1 the
// chunk's look-ups (rank lengths from LDS / workspace segments), 2 the eight steps of the state machine incl. ring stores,
// 3 the flush of the ring, 4 loop control (ballot, branch, the next chunk's loads issued)
#define GAT_PSTAMP(K) { unsigned long long t__; GAT_STAMP(t__); pdg[K] += t__ - pdg_t; pdg_t = t__; }
#else
#define GAT_PSTAMP(K)
#define GAT_PHASE(K) {}
#endif

constexpr uint32_t kMtUpper = 0x80000000u, kMtLower = 0x7fffffffu, kMtMag = 0x9908b0dfu;

__device__ __forceinline__ uint32_t mt_temper(uint32_t y) {
  y ^= (y >> 11);
  y ^= (y << 7) & 0x9d2c5680u;
  y ^= (y << 15) & 0xefc60000u;
  y ^= (y >> 18);
  return y;
}

// ------------------------------------------------------------------------------------------
// k_rng: one MT19937 stream per LANE (64 streams of one unit = one tile per workgroup).  The
// state of lane l is mt[i*64 + l] (bank-conflict free, 156 KB: one workgroup per CU).  Seeding and
// the twist are the serial reference recurrences run by all 64 lanes at once.  The workgroup has
// 2 NT waves (NT = 4, two per SIMD): in step s the twist waves 0..NT-1 twist chunks NT s + w in place while the
// temper waves temper the chunks of step s-1 and store them to HBM as rows of 64 (one coalesced 256-B store
// per output index, the order k_place consumes), one barrier per step.  The twist of word i reads words i,
// i+1 and i+397 (old) or i-227 (new, written >= 4 chunks = one step earlier); the only word another wave
// changes in the same step is the old word that follows a wave's chunk, which it reads one step ahead.
constexpr int kRngChunk = 24;      // 624 = 26 * 24

constexpr int kRngTwistWaves = 8;  // NT twist waves + NT temper waves per workgroup (four waves per SIMD)
constexpr int kRngThreads = 2 * kRngTwistWaves * kWave;
constexpr int kRngWaves = kRngThreads / kWave;
constexpr int kSeedSpan = kMtN / kRngWaves;          // 39: words of a stream's seeded state between two checkpoints
static_assert(kSeedSpan * kRngWaves == kMtN, "16 waves x 39 words = 624");

__device__ __forceinline__ uint32_t mt_seed_step(uint32_t x, uint32_t i) {       // init_genrand: word i from word i - 1
  return 1812433253u * (x ^ (x >> 30)) + i;
}

// k_seed: init_genrand of every stream of the batch, one stream per LANE, keeping only every 39th word.  The recurrence
// is one dependent chain of 624 multiply-adds per stream (28 cycles a step on gfx950: v_mul_lo_u32 is a quarter-rate
// instruction); inside k_rng -- whose 156 KB of LDS admit one workgroup per CU -- it kept ONE wave per CU busy for
// 17 500 cycles per tile while fifteen waited (config 3: 0.86 of k_rng's 1.57 ms).  Here it runs with nothing but a few
// registers per wave, so every SIMD has waves to issue from, and leaves 16 checkpoints per stream (4 KB per tile);
// k_rng's sixteen waves then regenerate 39 words each, side by side: 39 steps instead of 624 in front of the first twist.
constexpr int kSeedThreads = 256;
__global__ __launch_bounds__(kSeedThreads) void k_seed(SamplerArgs A, int n_blocks) {
  const int lane = threadIdx.x & 63;
  const int64_t tile = (int64_t)blockIdx.x * (kSeedThreads / kWave) + (threadIdx.x >> 6);
  if (tile >= (int64_t)A.n_active * n_blocks) return;
  const int a = (int)(tile / n_blocks), sb = (int)(tile - (int64_t)a * n_blocks);
  const int u = A.units_o[a].pad;
  const uint64_t sample_id = (uint64_t)(A.sample_begin + (int64_t)sb * kWave + lane);
  uint32_t x = (uint32_t)((uint64_t)A.seed + sample_id * (uint64_t)A.n_units + (uint64_t)u);
  uint32_t* __restrict__ dst = A.rng_ckpt + tile * (kRngWaves * kWave) + lane;
  for (int w = 0; w < kRngWaves; ++w) {
    dst[w * kWave] = x;                                             // word 39 w
#pragma unroll
    for (int j = 1; j <= kSeedSpan; ++j) x = mt_seed_step(x, (uint32_t)(w * kSeedSpan + j));
  }
}

__global__ __launch_bounds__(kRngThreads) void k_rng(SamplerArgs A) {
  extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
  constexpr int NT = kRngTwistWaves;
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));   // wave-uniform: keeps all index math scalar
  uint32_t* mt = lds + lane;                        // lane column, stride 64
  const int sb = blockIdx.x, a = (int)(blockIdx.y + blockIdx.z * gridDim.y);
  if (a >= A.n_active) return;                      // (whole workgroup: no barrier has been reached)
  const int rows = A.rng_rows[a];
  {
    // words 39 wv .. 39 wv + 38 of the seeded state from k_seed's checkpoint (init_genrand's recurrence, 39 steps)
    uint32_t x = A.rng_ckpt[((int64_t)a * gridDim.x + sb) * (kRngWaves * kWave) + wv * kWave + lane];
#pragma unroll
    for (int j = 0; j < kSeedSpan; ++j) {
      const int i = wv * kSeedSpan + j;
      mt[i * kWave] = x;
      x = mt_seed_step(x, (uint32_t)(i + 1));
    }
  }
  __syncthreads();
  uint32_t* __restrict__ out = A.rng_out + A.rng_off[a] + (int64_t)sb * rows * kWave + lane;
  const int nchunks = (rows + kRngChunk - 1) / kRngChunk;
  constexpr int kPerBlock = kMtN / kRngChunk;       // 13 chunks per 624-word block
  constexpr int kGroup = 8;                          // words per batch of LDS reads (rows and kRngChunk are multiples of 8)
  static_assert(kRngChunk % kGroup == 0 && kMtN % kRngChunk == 0, "chunking");
  static_assert((kMtN - kMtM) / kRngChunk >= NT, "a step must not read new words written in the same step");
  // old first word of the chunk after this wave's chunk of the NEXT step; in that step the neighbouring twist wave
  // rewrites it while this wave still needs the old value (the last twist wave's successor belongs to a later step)
  uint32_t next_old = 0;
  if (wv < NT - 1) next_old = mt[((wv + 1) % kPerBlock) * kRngChunk * kWave];
  __syncthreads();                                   // ... and nobody twists before everybody has it
  for (int s = 0; NT * (s - 1) < nchunks; ++s) {
    if (wv < NT) {
      const int t = NT * s + wv;
      if (t < nchunks) {
        const int e0 = t * kRngChunk;
        const int cnt = rows - e0 < kRngChunk ? rows - e0 : kRngChunk;
        const int i0 = (t % kPerBlock) * kRngChunk;
        uint32_t cur = mt[i0 * kWave];
        for (int k = 0; k < cnt; k += kGroup) {
          // all reads of the group first (none of them is a word this group writes: i+397 / i-227 are far away and
          // the old word i+1 is read before it is rewritten), so one LDS latency is exposed per group, not per word
          uint32_t far[kGroup], nx[kGroup];
#pragma unroll
          for (int q = 0; q < kGroup; ++q) {
            const int i = i0 + k + q;
            const int in = i + 1 == kMtN ? 0 : i + 1;                        // word 623 pairs with the NEW word 0
            const int jf = i + kMtM >= kMtN ? i + kMtM - kMtN : i + kMtM;    // i < 227: old word i+397, else new word i-227
            far[q] = mt[jf * kWave];
            nx[q] = mt[in * kWave];
          }
          if (wv < NT - 1 && k + kGroup == kRngChunk) nx[kGroup - 1] = next_old;   // the next wave is rewriting that word right now
#pragma unroll
          for (int q = 0; q < kGroup; ++q) {
            const uint32_t y = (cur & kMtUpper) | (nx[q] & kMtLower);
            mt[(i0 + k + q) * kWave] = far[q] ^ (y >> 1) ^ ((y & 1u) ? kMtMag : 0u);
            cur = nx[q];                                                      // old word i+1 is the next word's word i
          }
        }
      }
      if (wv < NT - 1) {                             // first word of chunk NT(s+1)+wv+1: untouched until the next step
        const int tn = NT * (s + 1) + wv + 1;
        next_old = mt[((tn % kPerBlock) * kRngChunk) * kWave];
      }
    } else {
      const int t = NT * (s - 1) + (wv - NT);        // temper waves: the chunks twisted in the previous step
      if (t >= 0 && t < nchunks) {
        const int e0 = t * kRngChunk;
        const int cnt = rows - e0 < kRngChunk ? rows - e0 : kRngChunk;
        const uint32_t* __restrict__ src = mt + ((t % kPerBlock) * kRngChunk) * kWave;
        uint32_t* __restrict__ dst = out + (int64_t)e0 * kWave;
        for (int k = 0; k < cnt; k += kGroup) {
          uint32_t w[kGroup];
#pragma unroll
          for (int q = 0; q < kGroup; ++q) w[q] = src[(k + q) * kWave];
#pragma unroll
          for (int q = 0; q < kGroup; ++q) {
#ifndef GAT_RNG_NO_NT
            // written once, read once and gigabytes later: past the L2 (config 2: k_rng 0.44 -> 0.49 ms but k_place, which no
            // longer shares the memory system with these lines' write-back, 0.84 -> 0.75; config 3 0.94 -> 0.88 and 0.96 -> 0.89)
            __builtin_nontemporal_store(mt_temper(w[q]), &dst[(k + q) * kWave]);
#else
            dst[(k + q) * kWave] = mt_temper(w[q]);
#endif
          }
        }
      }
    }
    __syncthreads();
  }
}

// ------------------------------------------------------------------------------------------
// k_place: the placement loop of SamplerAnnotator.sample (gat/Engine.pyx:572-635) up to the first
// consolidation, ONE STREAM PER LANE.  Every lane consumes exactly one raw MT19937 output per
// iteration (row j of its tile), so the loads are coalesced and data independent; what a lane
// does with the output depends on its state: L (length rank draw, :419-422), B (bucket offset,
// :432-433), P (workspace position, :299), O (offset inside the chosen segment, :326).  Rejected
// outputs (masked value > range) leave the state unchanged -- exactly numpy's masked rejection.
// A lane halts when `remaining <= length` (:582): the pending length, the number of outputs
// consumed and `remaining` are handed to k_sampler, which consolidates and finishes the unit.
// (kPlaceWsLds = 256 workspace segments and kPlaceRankLds = 1024 length-rank entries kept in LDS: gat_types.h)
constexpr int kPlaceChunk = 8;        // rows fetched per step (624 = 8 * 78)

// MODE (chosen by the host from the units' shapes, so that the common problems run a lean kernel): 1 every unit is of
// the single-workspace-segment shape with its rank table within the LDS table, only that loop is compiled in; 3 the
// same shape with units of thousands of segments (k_place_wide): a workgroup is kPlaceWide tiles of ONE unit that
// share the unit's whole rank table in (dynamic) LDS -- read from global memory, the eight gathers of a chunk are 512
// separate lines for the CU's vector cache to look up, which bounded the kernel on the config-4 shape; no lean kernel
// per tile can afford 32 KB of table; 0 no workspace beyond the LDS table; 2 everything.
// SMALL: every unit has at most 64 workspace segments and fewer than 256 working segments: quarter-size LDS tables,
// so that more tiles are resident per CU (the kernel has few waves and hides latency by their number).
// SMALL 2: at most 64 workspace segments, rank table at full size (a quarter of the workspace table's LDS back).
// PIPE: the rows of the single-workspace-segment loop are prefetched into pinned registers (v96..v127) by hand-written
// loads and waits, see GAT_PLACE_LOOP_PIPE below (k_place_pipe, k_place_wide).  Nothing reserves those registers: the
// loop's own values fit below v96 and tools/check_pinned_regs.py, run by the Makefile, fails the build when they do not.
template <int KIND, int MODE, int SMALL, bool PIPE>
__device__ __forceinline__ void place_body(const SamplerArgs& A) {
  constexpr bool ALL_SIMPLE = MODE == 1 || MODE == 3;
  constexpr bool TREES = MODE == 2;
  constexpr bool GRID = MODE == 5;                                         // k_place_grid: see GAT_PRE_WS3
  constexpr int kWsTab = SMALL ? 64 : kPlaceWsLds, kRankTab = MODE == 3 ? 1 : (SMALL == 1 ? 256 : kPlaceRankLds);
  constexpr int WIDE = MODE == 3 ? kPlaceWide : (MODE == 5 ? kPlaceGridTiles : 1);   // tiles (waves) of a workgroup
  __shared__ uint4 l_ws[kWsTab];          // {cdf, start, end, previous segment's end (INT32_MIN for the first)}
  __shared__ uint32_t l_rank_tab[kRankTab];
  __shared__ __attribute__((aligned(8192))) uint2 l_out_all[WIDE][16][kWave];   // (8 KB-aligned: GAT_STEP_SIMPLE_ASM) ring of 16 placed segments per lane, flushed 8 at a time as one 64-byte burst
  extern __shared__ __attribute__((aligned(16))) uint32_t l_rank_wide[];   // MODE 3: the unit's rank table, entry v = length of rank 1 + v
                                                                           // MODE 4 / 5: the image of the unit's cdf grid (UnitDev::cgrid_off)
  uint32_t* const l_rank = MODE == 3 ? l_rank_wide : l_rank_tab;
  const int lane = threadIdx.x & (kWave - 1), wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  uint2 (*const l_out)[kWave] = l_out_all[wv];
  const int n_tiles = (A.batch + kWave - 1) / kWave;
  const int a = (int)(blockIdx.y + blockIdx.z * gridDim.y);
  const int sb_own = (int)blockIdx.x * WIDE + wv;                       // (a wide workgroup's last waves may be beyond the batch: they idle
  const int sb = sb_own < n_tiles ? sb_own : n_tiles - 1;  //  through the loop on the last tile's rows and write nothing)
  if (a >= A.n_active) return;
  const UnitDev* __restrict__ Up = A.units_o + a;
  const int nws = Up->n_ws;
  const uint32_t hist_total = Up->hist_total, bucket = Up->bucket, ws_total = Up->ws_total;
  const int cap = Up->slab_cap;
  const uint2* __restrict__ ws = A.ws + Up->ws_off;
  const uint32_t* __restrict__ ws_cdf = A.ws_cdf + Up->ws_off;
  const uint32_t* __restrict__ rank_len = A.rank_len + Up->rank_off;
  const int rows = A.rng_rows[a];
  const int sidx = sb * kWave + lane;
  const bool live = sb_own < n_tiles && sidx < A.batch;
  const int64_t so = GAT_REC(A, sidx, a);                // hand-off record, indexed by launch position

  // wave-uniform draw parameters (numpy masked rejection: accept (y & mask) <= range)
  constexpr bool kind1 = KIND == 1;           // SamplerSegments: fixed number of placements, no trigger
  const int target = Up->n_target;
  const bool drawL = hist_total > 2;                 // randint(1,total): range total-2; range 0 consumes nothing
  const bool drawP = ws_total > 1;
  if (!drawL || !drawP) {                            // degenerate unit (<= 2 segments / 1-base workspace):
    if (live) A.st[so] = make_int4(0, Up->ltotal, -1, 0);
    return;                                          // k_sampler runs it from its seed
  }
  const uint32_t rangeL = hist_total - 2u, maskL = 0xffffffffu >> __builtin_clz(rangeL);
  const bool drawB = bucket > 1;
  const uint32_t rangeB = bucket - 1u, maskB = drawB ? 0xffffffffu >> __builtin_clz(rangeB) : 0u;
  const uint32_t rangeP = ws_total - 1u, maskP = 0xffffffffu >> __builtin_clz(rangeP);

  // all tiles of a batch are resident at once and the kernel ends with the tiles of the largest unit:
  // give those waves issue priority (units are ordered by size, a = 0 is the largest)
  {
    const int q = (4 * a) / A.n_active;
    if (q == 0) __builtin_amdgcn_s_setprio(3);
    else if (q == 1) __builtin_amdgcn_s_setprio(2);
    else if (q == 2) __builtin_amdgcn_s_setprio(1);
  }
  const bool ws_lds = nws <= kWsTab;
  const bool rank_lds = MODE == 3 || hist_total < (uint32_t)kRankTab;
  // longer workspaces are looked up through their 16-ary tree in global memory (WsTree, gat_device.h)
  const uint32_t* __restrict__ tree_cdf = A.ws_tree + (Up->tree_cdf_off >= 0 ? Up->tree_cdf_off : 0);
  const WsTreeGeom G = ws_tree_geom(nws);
  if (!ALL_SIMPLE && ws_lds)
    for (int i = lane; i < nws; i += kWave) {
      const uint2 v = ws[i];
      l_ws[i] = make_uint4(ws_cdf[i], v.x, v.y, i > 0 ? ws[i - 1].y : 0x80000000u);
    }
  const uint2 ws0 = ws[0];
  // the common shape -- one workspace segment (longer than one base), bucket size 1, rank table in
  // LDS -- needs no workspace search, no bucket draw and never an immediate placement
  const bool simple_shape = nws == 1 && !drawB && ws0.y - ws0.x > 1u;
  const bool simple_lds = ALL_SIMPLE || (simple_shape && rank_lds);      // GAT_STEP_SIMPLE_B's loop with the rank table in LDS runs
  if constexpr (MODE == 3) {
    for (int i = (int)threadIdx.x; i <= (int)rangeL; i += WIDE * kWave) l_rank[i] = rank_len[i + 1];
  } else if (simple_lds) {
    // entry v = length of rank 1 + v for v <= rangeL (the others are never accepted)
    for (int i = lane; i <= (int)maskL && i < kRankTab; i += kWave) l_rank[i] = (uint32_t)i <= rangeL ? rank_len[i + 1] : 0u;
  } else if (rank_lds)
    for (int i = lane; i <= (int)hist_total; i += kWave) l_rank[i] = rank_len[i];
  // k_place_grid: a workspace beyond the LDS table is looked up through the grid over its cumulated lengths (gat_prep.hip;
  // UnitDev::cgrid_off): g[c] = #{i : cdf[i] < c << cshift} and 16-bit keys cdf[i] & cmask, copied into LDS by the workgroup's
  // waves -- kPlaceGridTiles tiles of ONE unit --, and the segment's {start, end, previous end} as one 16-byte record
  int cshift = 0, cspan = 0;
  uint32_t cmask = 0u;
  const uint16_t* l_g16 = nullptr;
  const uint16_t* l_k16 = nullptr;
  const uint4* __restrict__ wrec = A.ws_rec + Up->ws_off;
  if constexpr (GRID) {
    if (!ws_lds) {
      const uint32_t* __restrict__ cg = A.ws_tree + Up->cgrid_off;
      cshift = (int)cg[0]; cspan = (int)cg[2];
      const int ccells = (int)cg[1], nwords = (int)cg[3];
      cmask = (1u << cshift) - 1u;
      const uint4* __restrict__ src4 = reinterpret_cast<const uint4*>(cg + kGridHeader);
      uint4* dst4 = reinterpret_cast<uint4*>(l_rank_wide);
      for (int i = (int)threadIdx.x; i < (nwords + 3) / 4; i += WIDE * kWave) dst4[i] = src4[i];
      l_g16 = reinterpret_cast<const uint16_t*>(l_rank_wide);
      l_k16 = reinterpret_cast<const uint16_t*>(l_rank_wide + (ccells + 2) / 2);
    }
  }
  __syncthreads();
  // The workspace segment of a position draw from a grid over the cumulated lengths instead of a halving search over the
  // whole table (five dependent LDS reads per raw output for an isochore unit's 31 blocks, looked up for EVERY output of a
  // chunk since the look-up depends on the value alone): up to 128 cells, cell c = the first segment whose cumulated length
  // reaches c << gshift.  The segment of p lies between grid[cell] and grid[cell + 1]: the halving search runs over the
  // widest such span of the unit only -- not at all when no cell spans more than two segments (equal blocks, assembly
  // pieces), where the look-up's last step decides between the two.  The grid (one byte per cell) lives in the unused end
  // of the workspace table.
  bool use_grid = false;
  int gshift = 0, gspan = 0;
  uint8_t* l_grid = reinterpret_cast<uint8_t*>(l_ws + (kWsTab - 9));
  if (!ALL_SIMPLE && ws_lds && !simple_shape && nws > 2 && nws <= kWsTab - 9) {          // (wave-uniform)
    const uint32_t top = ws_total - 1u;
    gshift = top >= 128u ? (32 - __builtin_clz(top)) - 7 : 0;
    const int cells = (int)(top >> gshift) + 1;
    for (int c = lane; c <= cells; c += kWave) {
      const uint32_t t = (uint32_t)c << gshift;
      int lo_ = 0, hi_ = nws;                          // leftmost i with (int)(cdf[i] - t) >= 0
      while (lo_ < hi_) {
        const int mid = lo_ + ((hi_ - lo_) >> 1);
        if ((int32_t)(l_ws[mid].x - t) < 0) lo_ = mid + 1; else hi_ = mid;
      }
      l_grid[c] = (uint8_t)(lo_ < nws ? lo_ : nws - 1);
    }
    __syncthreads();
    uint32_t span = 0;
    for (int c = lane; c < cells; c += kWave) span = max(span, (uint32_t)l_grid[c + 1] - (uint32_t)l_grid[c]);
    gspan = (int)wave_max_u32(span);
    use_grid = true;
  }
  const uint32_t* __restrict__ rp = A.rng_out + A.rng_off[a] + (int64_t)sb * rows * kWave + lane;
  uint2* __restrict__ out = A.slab + (int64_t)(live ? sidx : 0) * A.slab_stride + Up->slab_off;

  int32_t rem = Up->ltotal;
  int nS = 0;
  uint32_t len = 0, cs = 0, ce = 0;
  int32_t sstart = 0;
  int32_t pend = -1;           // pending length at the trigger; -1 = not triggered; -2 = SamplerSegments complete
  uint32_t used = 0;           // raw outputs consumed by this lane when it halted
  int flag = 0;


  // Straight-line form: a lane's state only selects which of the small results it keeps, so the wave runs one
  // instruction stream instead of divergent ones.  The state is held as booleans (in L / in P / in O; none: halted), not
  // as a number: the compiler keeps them as lane masks in scalar registers and the transitions become scalar logic beside
  // the vector work (51 -> 3x vector instructions per output).  LR1 = the length of rank 1 + (y & maskL), read for the
  // whole chunk up front (one exposed latency per chunk instead of one per output).
  bool sL = live, sP = false, sO = false;
  uint32_t omaskS = 0u, orangeS = 0u;
  const int32_t c_ss = (int32_t)ws0.x + 1;               // sampling_start = ws.start - length + 1 (:318)
  const uint32_t c_r3 = ws0.y - ws0.x - 2u;              // its range: ws.end - 1 - sampling_start = c_r3 + length
#define GAT_STEP_SIMPLE_B(Y, LR1, JJ)                                                                          \
  {                                                                                                            \
    const uint32_t y_ = (Y);                                                                                   \
    /* would this output be accepted as a rank draw / a position draw / an offset draw (numpy's masked rejection; the   \
       offset's range follows from the length drawn last: sampling_start = c_ss - len, range c_r3 + len) */    \
    const bool accL = (y_ & maskL) <= rangeL;                                                                  \
    const bool accP = (y_ & maskP) <= rangeP;                                                                  \
    /* (the offset draw's mask and range are kept from the position draw on, as the general step does: the chain          \
        length -> range -> leading zeros -> mask -> value -> test is off the row's critical path) */          \
    const uint32_t vO = y_ & omaskS;                                                                           \
    const bool accO = vO <= orangeS;                                                                           \
    const bool isL = sL && accL, isP = sP && accP, isO = sO && accO;                                           \
    const uint32_t range3 = c_r3 + len;                                                                        \
    omaskS = isP ? 0xffffffffu >> __builtin_clz(range3 | 1u) : omaskS;                                         \
    orangeS = isP ? range3 : orangeS;                                                                          \
    const bool trig = isL && !kind1 && rem <= (int32_t)(LR1);          /* :582 -> consolidate */               \
    const int32_t q = c_ss - (int32_t)len + (int32_t)vO;                                                       \
    const uint32_t start = (uint32_t)(q > 0 ? q : 0);                                                          \
    const uint32_t end = (uint32_t)(q + (int32_t)len);                                                         \
    const int32_t omin = (int32_t)ws0.y < (int32_t)end ? (int32_t)ws0.y : (int32_t)end;                        \
    const int32_t omax = (int32_t)ws0.x > (int32_t)start ? (int32_t)ws0.x : (int32_t)start;                    \
    const int32_t overlap = omin - omax > 0 ? omin - omax : 0;                                                 \
    const bool full = isO && nS >= cap;                                                                        \
    const bool put = isO && !full;                                                                             \
    /* (k_place_wide stores always -- a slot that is not kept is written again: 11.2 -> 10.75 ms on the config-4 shape; \
        the lean kernels lose by it, 0.95 -> 0.96 ms on config 2) */                                          \
    if (MODE == 3 || put) l_out[nS & 15][lane] = make_uint2(start, end);                                       \
    nS += put ? 1 : 0;                                                                                         \
    rem -= put ? overlap : 0;                                                                                  \
    flag |= full ? kStatusOverflow : 0;                                                                        \
    const bool fin = kind1 && put && nS == target;                                                             \
    len = isL ? (LR1) : len;                                                                                   \
    pend = trig ? (int32_t)len : (fin ? -2 : pend);                                                            \
    used = (trig || put) ? (JJ) + 1u : used;       /* (a lane that runs out of rows is resumed behind its last placement) */ \
    sL = (sL && !isL) || (put && !fin);                                                                        \
    sP = (sP && !isP) || (isL && !trig);                                                                       \
    sO = (sO && !isO) || isP;                                                                                  \
  }

  // The same step where the offset draw's MASK does not depend on the length drawn: its range is c_r3 + length, the position
  // draw's c_r3 + 1, and unless a power of two lies between c_r3 + 1 and c_r3 + the unit's longest length every length gives
  // the position draw's mask (a contig of 50-250 Mb against segments of hundreds of bases: cm_ok below; a unit where it does
  // not hold takes the step above).  Then y & maskP serves both tests, the chain length -> range -> leading zeros -> mask and
  // its two selects per row are gone, and what only an accepted offset draw needs -- position, clipping, overlap, the ring
  // store, the counters -- stands inside `if (isO)`: one skipped region instead of selects per row.  The slab's capacity is not
  // tested per row either: a lane that runs past it keeps counting (the ring is LDS; flush() stops at the capacity) and is
  // flagged after the loop -- the unit is run again either way.  A triggered lane's pending length is the length it holds
  // when the loop ends (nothing changes it once the lane is in no state).
  const uint32_t lmaxS = rank_len[1u + rangeL];           // (the table ascends: the longest length that can be drawn)
  const bool cm_ok = A.place_plain_step == 0 && (uint64_t)c_r3 + (uint64_t)lmaxS <= (uint64_t)maskP;
  // The step itself is written out (one asm statement per row): the compiler's form of the same logic costs a lone wave ~8
  // cycles per instruction (every v_cmp -> s_and -> v_cndmask hop waits out its predecessor; tools/ubench/valu_issue.hip:
  // that chain alone runs at 8.1 cycles per instruction for one wave, independent instructions at 4.6), here the four
  // compares are issued back to back, the lane-mask logic follows as one scalar block, and what only an accepted offset draw
  // needs runs with exec = those lanes (no branch: with 64 lanes some lane places in nearly every row).  18 vector, 13
  // scalar, one LDS instruction per row.  State: mL / mP / mO lane masks (in L / in P / in O), len, rem, nS9 = segments
  // placed << 9 (the ring slot's byte offset is nS9 & 0x1e00: the ring is 8 KB-aligned), used_lo = 1 + the row within the
  // current trip of 32 rows at which the lane last placed or triggered (an inline constant; folded into `used` per trip).
  // overlap = min(ws.end, end) - max(ws.start, start) is at least 1 for an accepted offset (q in [ws.start + 1 - len,
  // ws.end - 1], coordinates below 2^31: gat_problem_create), so the reference's max(0, .) (gat/Engine.pyx:340-342) is not taken.
  uint64_t mL = 0, mP = 0, mO = 0;
  uint32_t nS9 = 0, used_lo = 0;
  const uint32_t lane_ring = (uint32_t)(size_t)(__attribute__((address_space(3))) void*)&l_out[0][lane];
#define GAT_STEP_SIMPLE_ASM(YREG, LR1, JJ1)                                                                       \
  {                                                                                                            \
    uint32_t t0_, t1_, t2_, t3_;                                                                               \
    uint64_t sa_, sb_, sc_;                                                                                    \
    asm volatile(                                                                                              \
        "v_and_b32 %7, %16, v" #YREG "\n\t"                /* vO = y & maskP */                                     \
        "v_and_b32 %8, %26, v" #YREG "\n\t"                /* y & maskL */                                          \
        "v_cmp_ge_u32_e64 %12, %17, %7\n\t"           /* accP: vO <= rangeP */                                 \
        "v_add_u32 %9, %19, %0\n\t"                   /* c_r3 + len */                                         \
        "v_cmp_ge_u32 vcc, %18, %8\n\t"               /* accL */                                               \
        "v_cmp_le_u32_e64 %13, %7, %9\n\t"            /* accO: vO <= c_r3 + len */                             \
        "s_and_b64 %11, %4, vcc\n\t"                  /* isL */                                                \
        "v_cmp_le_i32 vcc, %1, %15\n\t"               /* remaining <= length of this rank */                   \
        "s_and_b64 %12, %5, %12\n\t"                  /* isP */                                                \
        "s_and_b64 %13, %6, %13\n\t"                  /* isO */                                                \
        "s_and_b64 vcc, %11, vcc\n\t"                 /* trigger = isL && remaining <= length (:582) */        \
        "v_cndmask_b32_e64 %0, %0, %15, %11\n\t"      /* len = isL ? length : len */                           \
        "v_cndmask_b32_e64 %3, %3, %25, vcc\n\t"      /* used_lo = trigger ? row + 1 : used_lo */              \
        "s_xor_b64 %4, %4, %11\n\t"                                                                            \
        "s_andn2_b64 %11, %11, vcc\n\t"               /* isL && !trigger */                                    \
        "s_xor_b64 %5, %5, %12\n\t"                                                                            \
        "s_xor_b64 %6, %6, %13\n\t"                                                                            \
        "s_or_b64 %4, %4, %13\n\t"                    /* L: left by an accepted rank draw, entered by a placement */ \
        "s_or_b64 %5, %5, %11\n\t"                    /* P: ... position draw / rank draw without trigger */   \
        "s_or_b64 %6, %6, %12\n\t"                    /* O: ... offset draw / position draw */                 \
        "s_and_saveexec_b64 %11, %13\n\t"             /* the lanes that place (:318-340) */                    \
        "v_add_u32 %9, %20, %7\n\t"                   /* end = sampling_start + offset + len = c_ss + vO */    \
        "v_and_or_b32 %8, %2, %23, %24\n\t"           /* ring slot */                                          \
        "v_sub_u32 %7, %9, %0\n\t"                    /* q = end - len */                                      \
        "v_add_u32 %2, 0x200, %2\n\t"                                                                          \
        "v_min_i32 %10, %22, %9\n\t"                  /* min(ws.end, end) */                                   \
        "v_max_i32 %7, 0, %7\n\t"                     /* start = max(q, 0) */                                  \
        "v_sub_u32 %1, %1, %10\n\t"                   /* remaining -= overlap = min(..) - max(..) */           \
        "v_max_i32 %10, %21, %7\n\t"                  /* max(ws.start, start) */                               \
        "ds_write2_b32 %8, %7, %9 offset1:1\n\t"                                                               \
        "v_mov_b32 %3, %25\n\t"                                                                                \
        "v_add_u32 %1, %1, %10\n\t"                                                                            \
        "s_mov_b64 exec, %11"                                                                                  \
        : "+v"(len), "+v"(rem), "+v"(nS9), "+v"(used_lo), "+s"(mL), "+s"(mP), "+s"(mO),                        \
          "=&v"(t0_), "=&v"(t1_), "=&v"(t2_), "=&v"(t3_), "=&s"(sa_), "=&s"(sb_), "=&s"(sc_)                   \
        : "n"(0), "v"(LR1), "s"(maskP), "s"(rangeP), "s"(rangeL), "s"(c_r3), "s"(c_ss), "s"(ws0.x), "s"(ws0.y), \
          "s"(0x1e00u), "v"(lane_ring), "n"(JJ1), "s"(maskL)                                                   \
        : "vcc", "scc", "memory");                                                                             \
  }

  // rows are consumed in chunks of kPlaceChunk; the next chunk is in flight while this one is worked on
  // (few waves per SIMD: nothing else hides the load latency)
  // The ring is flushed behind every SECOND chunk: a placement takes at least two accepted outputs, so two chunks add at
  // most kPlaceChunk = 8 segments to the at most 7 a flush leaves -- 15 of the ring's 16 slots, the one left is where the
  // next segment goes (k_place_wide writes it unconditionally).  In-kernel stamps (tools/diag_place.sh,
  // profiles/r04_k_place_phases.txt) had the flush at a quarter of a wave's cycles per row: it runs whenever ANY lane has
  // eight segments waiting -- with 64 lanes out of step that is every chunk -- and its eight LDS reads stand in front of the
  // stores with nothing to overlap them (the row loads' asm statements fence the schedule)
  int nF = 0;                  // segments already written to the slab (multiple of 8)
  auto flush = [&]() __attribute__((always_inline)) {
    if (nS - nF >= 8 && nF + 8 <= cap) {                  // (beyond the capacity: GAT_STEP_SIMPLE_ASM's lanes run on, flagged below)
      uint4* __restrict__ dst = reinterpret_cast<uint4*>(out + nF);
      const int w0 = nF & 15;
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        const uint2 e0 = l_out[w0 + 2 * w][lane], e1 = l_out[w0 + 2 * w + 1][lane];
#ifdef GAT_EXP_NO_STORES
        if (e0.x == 0xfffffff1u && e1.y == 0xfffffff2u)            // (never: the reads stay, the stores go)
#endif
        dst[w] = make_uint4(e0.x, e0.y, e1.x, e1.y);
      }
      nF += 8;
    }
  };
  // The general shape in the same straight-line form: several workspace segments (table in LDS or search tree),
  // optional bucket draw.  The workspace lookup of an output depends only on its value, not on the lane's state, so the
  // lookups of a whole chunk (PCS/PCE/PPE: chosen segment's start, end, previous end) are done up front, interleaved; a
  // lane keeps the one it needs when it is in state P at that output.  State as booleans (in L / B / P / O; none: halted):
  // whether an output would be accepted is worked out for every kind of draw, the lane's state picks; the offset draw's
  // mask and range (they follow from the chosen workspace segment and the length) are kept from the position draw on.
  bool sB = false;
  bool alive_c = true;         // the lane is in some state at the head of the chunk (GAT_PRE_WS3 asks for no record otherwise)
  uint32_t omask = 0u, orange = 0u;
#define GAT_STEP_TABLE_B(Y, LR, PCS, PCE, PPE, JJ)                                                             \
  {                                                                                                            \
    const uint32_t y_ = (Y);                                                                                   \
    const uint32_t vB = y_ & maskB, vO = y_ & omask;                                                           \
    const bool isL = sL && (y_ & maskL) <= rangeL, isB = sB && vB <= rangeB;                                   \
    const bool isP = sP && (y_ & maskP) <= rangeP, isO = sO && vO <= orange;                                   \
    len = isL ? (LR) * bucket : (isB ? len + vB : len);                /* :419-433 */                          \
    const bool have_len = drawB ? isB : isL;                                                                   \
    const bool trig = have_len && !kind1 && rem <= (int32_t)len;       /* :582 -> consolidate */               \
    cs = isP ? (PCS) : cs;                                                                                     \
    ce = isP ? (PCE) : ce;                                                                                     \
    int32_t sstartP = (int32_t)(PCS) - (int32_t)len + 1;               /* :318-325 */                          \
    sstartP = (int32_t)(PPE) > sstartP ? (int32_t)(PPE) : sstartP;                                             \
    const uint32_t range3 = (PCE) - 1u - (uint32_t)sstartP;                                                    \
    const bool placeP = isP && range3 == 0;                            /* range 0: randint consumes nothing */ \
    const bool place = isO || placeP;                                                                          \
    const int32_t q = isO ? sstart + (int32_t)vO : sstartP;                                                    \
    const uint32_t start = (uint32_t)(q > 0 ? q : 0);                  /* :331-343, :630-635 */                \
    const uint32_t end = (uint32_t)(q + (int32_t)len);                                                         \
    const int32_t omin = (int32_t)ce < (int32_t)end ? (int32_t)ce : (int32_t)end;                              \
    const int32_t omax = (int32_t)cs > (int32_t)start ? (int32_t)cs : (int32_t)start;                          \
    const int32_t overlap = omin - omax > 0 ? omin - omax : 0;                                                 \
    const bool full = place && nS >= cap;                                                                      \
    const bool put = place && !full;                                                                           \
    if (put) l_out[nS & 15][lane] = make_uint2(start, end);                                                    \
    nS += put ? 1 : 0;                                                                                         \
    rem -= put ? overlap : 0;                                                                                  \
    flag |= full ? kStatusOverflow : 0;                                                                        \
    const bool fin = kind1 && put && nS == target;                                                             \
    pend = trig ? (int32_t)len : (fin ? -2 : pend);                                                            \
    used = (trig || put) ? (JJ) + 1u : used;                                                                   \
    sstart = isP ? sstartP : sstart;                                                                           \
    omask = isP ? 0xffffffffu >> __builtin_clz(range3 | 1u) : omask;                                           \
    orange = isP ? range3 : orange;                                                                            \
    sL = (sL && !isL) || (put && !fin);                                                                        \
    sB = (sB && !isB) || (isL && drawB);                                                                       \
    sP = (sP && !isP) || (have_len && !trig);                                                                  \
    sO = (sO && !isO) || (isP && !placeP);                                                                     \
  }

  uint32_t ya[kPlaceChunk], yb[kPlaceChunk], lr[kPlaceChunk] = {0, 0, 0, 0, 0, 0, 0, 0};
  uint32_t pcs[kPlaceChunk] = {0, 0, 0, 0, 0, 0, 0, 0}, pce[kPlaceChunk] = {0, 0, 0, 0, 0, 0, 0, 0},
           ppe[kPlaceChunk] = {0, 0, 0, 0, 0, 0, 0, 0};
  // The general step written out like GAT_STEP_SIMPLE_ASM, for units without a bucket draw (bucket size 1: every problem
  // whose longest segment is below nbuckets): the acceptance tests and the lane-mask logic, then what an accepted POSITION
  // draw needs with exec = those lanes (the chosen workspace segment, sampling_start = max(previous end, start - len + 1)
  // (:318-325), the offset draw's range and mask; a range of 0 places at once: randint consumes nothing), then what a
  // placement needs with exec = the lanes that place.  31 vector + 19 scalar + one LDS instruction per row.  The overlap with
  // the chosen segment is at least 1 (q in [sampling_start, segment end - 1], sampling_start >= segment start - len + 1).
#define GAT_STEP_TABLE_ASM(YREG, LR1, PCS, PCE, PPE, JJ1)                                                         \
  {                                                                                                            \
    uint32_t t0_, t1_, t2_, t3_;                                                                               \
    uint64_t sa_, sb_, sc_;                                                                                    \
    asm volatile(                                                                                              \
        "v_and_b32 %12, %24, v" #YREG "\n\t"               /* y & maskP */                                          \
        "v_and_b32 %13, %27, v" #YREG "\n\t"               /* y & maskL */                                          \
        "v_and_b32 %14, %8, v" #YREG "\n\t"                /* vO = y & the offset draw's mask */                    \
        "v_cmp_ge_u32_e64 %17, %25, %12\n\t"          /* accP */                                               \
        "v_cmp_ge_u32 vcc, %26, %13\n\t"              /* accL */                                               \
        "v_cmp_le_u32_e64 %18, %14, %9\n\t"           /* accO: vO <= its range */                              \
        "s_and_b64 %16, %4, vcc\n\t"                  /* isL */                                                \
        "v_cmp_le_i32 vcc, %1, %20\n\t"               /* remaining <= length of this rank */                   \
        "s_and_b64 %17, %5, %17\n\t"                  /* isP */                                                \
        "s_and_b64 %18, %6, %18\n\t"                  /* isO */                                                \
        "s_and_b64 vcc, %16, vcc\n\t"                 /* trigger (:582) */                                     \
        "v_cndmask_b32_e64 %0, %0, %20, %16\n\t"      /* len */                                                \
        "v_cndmask_b32_e64 %3, %3, %30, vcc\n\t"      /* used_lo */                                            \
        "s_xor_b64 %4, %4, %16\n\t"                                                                            \
        "s_andn2_b64 %16, %16, vcc\n\t"               /* isL && !trigger */                                    \
        "s_xor_b64 %5, %5, %17\n\t"                                                                            \
        "s_xor_b64 %6, %6, %18\n\t"                                                                            \
        "s_or_b64 %5, %5, %16\n\t"                    /* P entered by a rank draw without trigger */           \
        "s_and_saveexec_b64 %16, %17\n\t"             /* the lanes whose position draw is accepted */          \
        "v_sub_u32 %12, %21, %0\n\t"                  /* segment start - len */                                \
        "v_mov_b32 %10, %21\n\t"                      /* cs */                                                 \
        "v_add_u32 %12, 1, %12\n\t"                                                                            \
        "v_mov_b32 %11, %22\n\t"                      /* ce */                                                 \
        "v_max_i32 %7, %23, %12\n\t"                  /* sampling_start */                                     \
        "v_mov_b32 %14, 0\n\t"                        /* (an immediate placement is at sampling_start) */      \
        "v_sub_u32 %12, %22, %7\n\t"                                                                           \
        "v_add_u32 %9, -1, %12\n\t"                   /* range of the offset draw = ce - 1 - sampling_start */ \
        "v_or_b32 %12, 1, %9\n\t"                                                                              \
        "v_ffbh_u32 %12, %12\n\t"                                                                              \
        "v_cmp_eq_u32 vcc, 0, %9\n\t"                 /* range 0: placed at once */                            \
        "v_lshrrev_b32_e64 %8, %12, -1\n\t"           /* its mask */                                           \
        "s_mov_b64 exec, %16\n\t"                                                                              \
        "s_andn2_b64 %17, %17, vcc\n\t"               /* isP && !immediate */                                  \
        "s_or_b64 %18, %18, vcc\n\t"                  /* the lanes that place */                               \
        "s_or_b64 %6, %6, %17\n\t"                    /* O entered by a position draw */                       \
        "s_or_b64 %4, %4, %18\n\t"                    /* L entered by a placement */                           \
        "s_and_saveexec_b64 %16, %18\n\t"                                                                      \
        "v_add_u32 %12, %7, %14\n\t"                  /* q = sampling_start + offset */                        \
        "v_and_or_b32 %13, %2, %28, %29\n\t"          /* ring slot */                                          \
        "v_add_u32 %15, %12, %0\n\t"                  /* end = q + len */                                      \
        "v_max_i32 %12, 0, %12\n\t"                   /* start = max(q, 0) */                                  \
        "v_add_u32 %2, 0x200, %2\n\t"                                                                          \
        "v_min_i32 %14, %11, %15\n\t"                 /* min(ce, end) */                                       \
        "v_sub_u32 %1, %1, %14\n\t"                   /* remaining -= overlap */                               \
        "v_max_i32 %14, %10, %12\n\t"                 /* max(cs, start) */                                     \
        "ds_write2_b32 %13, %12, %15 offset1:1\n\t"                                                            \
        "v_mov_b32 %3, %30\n\t"                                                                                \
        "v_add_u32 %1, %1, %14\n\t"                                                                            \
        "s_mov_b64 exec, %16"                                                                                  \
        : "+v"(len), "+v"(rem), "+v"(nS9), "+v"(used_lo), "+s"(mL), "+s"(mP), "+s"(mO),                        \
          "+v"(sstart), "+v"(omask), "+v"(orange), "+v"(cs), "+v"(ce),                                         \
          "=&v"(t0_), "=&v"(t1_), "=&v"(t2_), "=&v"(t3_), "=&s"(sa_), "=&s"(sb_), "=&s"(sc_)                   \
        : "n"(0), "v"(LR1), "v"(PCS), "v"(PCE), "v"(PPE), "s"(maskP), "s"(rangeP), "s"(rangeL), "s"(maskL),    \
          "s"(0x1e00u), "v"(lane_ring), "n"(JJ1)                                                               \
        : "vcc", "scc", "memory");                                                                             \
  }
  // per-chunk look-ups that depend only on the output value: length of rank 1 + (y & maskL), from the LDS copy of the
  // table or (units with >= kPlaceRankLds segments) from global memory ...
#define GAT_PRE_RANK_L(Y)                                                                                      \
  _Pragma("unroll") for (int c = 0; c < kPlaceChunk; ++c) {                                                    \
    const uint32_t v = (Y)[c] & maskL; lr[c] = l_rank[v <= rangeL ? 1u + v : 0u]; }
#define GAT_PRE_RANK_G(Y)                                                                                      \
  _Pragma("unroll") for (int c = 0; c < kPlaceChunk; ++c) {                                                    \
    const uint32_t v = (Y)[c] & maskL; lr[c] = rank_len[v <= rangeL ? 1u + v : 0u]; }
  // ... and the workspace segment holding position (y & maskP): leftmost k with (int)(cdf[k] - p) >= 0
  // (utils/gat_utils.c:36 + cmpPosition) by a halving search whose trip count depends on nws only
#define GAT_PRE_WS(Y)                                                                                          \
  {                                                                                                            \
    uint32_t pv[kPlaceChunk]; int lo[kPlaceChunk];                                                             \
    _Pragma("unroll") for (int c = 0; c < kPlaceChunk; ++c) {                                                  \
      const uint32_t v = (Y)[c] & maskP; pv[c] = v <= rangeP ? v : rangeP; lo[c] = 0; }                        \
    if (use_grid) {                                                                                            \
      _Pragma("unroll") for (int c = 0; c < kPlaceChunk; ++c) lo[c] = (int)l_grid[pv[c] >> gshift];             \
    }                                                                                                          \
    for (int n = use_grid ? gspan : nws; n > 1;) {      /* (the last step below covers a span of two) */        \
      const int half = n >> 1;                                                                                 \
      _Pragma("unroll") for (int c = 0; c < kPlaceChunk; ++c) {                                                \
        const int pr = lo[c] + half - 1;                /* (a span may reach beyond the table: its last entry holds) */ \
        lo[c] = (int32_t)(l_ws[pr < nws ? pr : nws - 1].x - pv[c]) < 0 ? lo[c] + half : lo[c];                 \
      }                                                                                                        \
      n -= half;                                                                                               \
    }                                                                                                          \
    _Pragma("unroll") for (int c = 0; c < kPlaceChunk; ++c) {                                                  \
      uint4 w4 = l_ws[lo[c]];                                                                                  \
      if ((int32_t)(w4.x - pv[c]) < 0) w4 = l_ws[lo[c] + 1];                                                   \
      pcs[c] = w4.y; pce[c] = w4.z; ppe[c] = w4.w; }                                                           \
  }
  // the same for a workspace beyond the LDS table: its search tree, eight look-ups level by level
#define GAT_PRE_WS2(Y)                                                                                         \
  {                                                                                                            \
    uint32_t pv[kPlaceChunk]; int lo[kPlaceChunk];                                                             \
    _Pragma("unroll") for (int c = 0; c < kPlaceChunk; ++c) {                                                  \
      const uint32_t v = (Y)[c] & maskP; pv[c] = v <= rangeP ? v : rangeP; }                                   \
    {                                                   /* two batches of four: a node is 16 registers */     \
      const uint32_t pa[4] = {pv[0], pv[1], pv[2], pv[3]}, pb[4] = {pv[4], pv[5], pv[6], pv[7]};               \
      int la[4], lb[4];                                                                                        \
      ws_tree_count<true, 4>(tree_cdf, G, pa, la);                                                             \
      ws_tree_count<true, 4>(tree_cdf, G, pb, lb);                                                             \
      _Pragma("unroll") for (int c = 0; c < 4; ++c) { lo[c] = la[c]; lo[4 + c] = lb[c]; }                      \
    }                                                                                                          \
    _Pragma("unroll") for (int c = 0; c < kPlaceChunk; ++c) {                                                  \
      const uint2 w2 = ws[lo[c]];                                                                              \
      pcs[c] = w2.x; pce[c] = w2.y; ppe[c] = lo[c] > 0 ? ws[lo[c] - 1].y : 0x80000000u; }                      \
  }
  // the same for a workspace beyond the LDS table where its grid is in LDS (k_place_grid): the cell's two entries, a halving
  // search over the widest cell's span among the cell's 16-bit keys (entries behind the cell's end count as above every p),
  // and ONE 16-byte record of the chosen segment from global memory -- where the tree took four dependent 64-byte nodes
#ifdef GAT_EXP_NOGATHER                 /* timing-only builds (results are wrong): every look-up reads record 0 */
#define GAT_EXP_REC(I) ((I) & 0)
#else
#define GAT_EXP_REC(I) (I)
#endif
#ifdef GAT_EXP_NOSEARCH                 /* ... / the LDS search left out: the cell's first segment it is */
#define GAT_EXP_SPAN(N) 0
#else
#define GAT_EXP_SPAN(N) (N)
#endif
#define GAT_PRE_WS3(Y) GAT_PRE_WS3X(Y, pcs, pce, ppe)
#define GAT_PRE_WS3X(Y, PCS, PCE, PPE)                                                                         \
  {                                                                                                            \
    uint32_t tk[kPlaceChunk]; int lo[kPlaceChunk], hi[kPlaceChunk];                                            \
    bool need[kPlaceChunk];                                                                                    \
    _Pragma("unroll") for (int c = 0; c < kPlaceChunk; ++c) {                                                  \
      const uint32_t v = (Y)[c] & maskP; const uint32_t pv = v <= rangeP ? v : rangeP;                         \
      /* a lane that has halted, or a value numpy's rejection would not take as a position: no record is wanted -- all   \
         such lanes read record 0, one request per instruction instead of one per lane (the look-ups run at the rate    \
         the L2s serve 16-byte gathers: a fifth of the values are rejected, a sixth of a tile's lane-rows are idle) */ \
      need[c] = alive_c && v <= rangeP;                                                                        \
      const uint32_t cell = pv >> cshift; tk[c] = pv & cmask;                                                  \
      lo[c] = (int)l_g16[cell]; hi[c] = (int)l_g16[cell + 1u]; }                                               \
    /* (the eight reads of a step issued together, every one of them: written as `in && key < t` the compiler sank each read   \
        into a branch of its own behind the bound test, with its s_waitcnt -- 32 LDS round trips one after the other per    \
        chunk, most of k_place_grid's time; the empty asm pins the value outside any branch) */                               \
    for (int n = GAT_EXP_SPAN(cspan); n > 1;) {                                                                \
      const int half = n >> 1;                                                                                 \
      uint32_t key[kPlaceChunk];                                                                               \
      _Pragma("unroll") for (int c = 0; c < kPlaceChunk; ++c) {                                                \
        const int pr = lo[c] + half - 1;                                                                       \
        key[c] = (uint32_t)l_k16[pr < nws ? pr : nws - 1];                                                     \
        asm volatile("" : "+v"(key[c]));                                                                       \
      }                                                                                                        \
      _Pragma("unroll") for (int c = 0; c < kPlaceChunk; ++c) {                                                \
        const int pr = lo[c] + half - 1;                                                                       \
        const uint32_t kx = pr < hi[c] ? key[c] : 0xffffffffu;      /* (behind the cell's end: above every p) */  \
        lo[c] = kx < tk[c] ? lo[c] + half : lo[c];                                                             \
      }                                                                                                        \
      n -= half;                                                                                               \
    }                                                                                                          \
    {                                                                                                          \
      uint32_t key[kPlaceChunk];                                                                               \
      _Pragma("unroll") for (int c = 0; c < kPlaceChunk; ++c) {                                                \
        key[c] = (uint32_t)l_k16[lo[c] < nws ? lo[c] : nws - 1];                                               \
        asm volatile("" : "+v"(key[c]));                                                                       \
      }                                                                                                        \
      _Pragma("unroll") for (int c = 0; c < kPlaceChunk; ++c) {                                                \
        const uint32_t kx = lo[c] < hi[c] ? key[c] : 0xffffffffu;                                              \
        lo[c] += kx < tk[c] ? 1 : 0; }                                                                         \
    }                                                                                                          \
    _Pragma("unroll") for (int c = 0; c < kPlaceChunk; ++c) {                                                  \
      const uint4 w4 = wrec[GAT_EXP_REC(need[c] ? lo[c] : 0)];                                                 \
      PCS[c] = w4.x; PCE[c] = w4.y; PPE[c] = w4.z; }                                                           \
  }
#define GAT_PRE_SIMPLE_L(Y)                                                                                    \
  _Pragma("unroll") for (int c = 0; c < kPlaceChunk; ++c) lr[c] = l_rank[(Y)[c] & maskL];
#define GAT_PRE_SIMPLE_W(Y)                    /* (k_place_wide's table ends at rangeL) */                    \
  _Pragma("unroll") for (int c = 0; c < kPlaceChunk; ++c) {                                                    \
    const uint32_t v = (Y)[c] & maskL; lr[c] = l_rank[v <= rangeL ? v : 0u]; }
#define GAT_PRE_SIMPLE_G(Y) GAT_PRE_RANK_G(Y)
#define GAT_PRE_TABLE_LL(Y) GAT_PRE_RANK_L(Y) GAT_PRE_WS(Y)
#define GAT_PRE_TABLE_GL(Y) GAT_PRE_RANK_G(Y) GAT_PRE_WS(Y)
#define GAT_PRE_TABLE_LG(Y) GAT_PRE_RANK_L(Y) GAT_PRE_WS2(Y)
#define GAT_PRE_TABLE_GG(Y) GAT_PRE_RANK_G(Y) GAT_PRE_WS2(Y)
#define GAT_PRE_TABLE_L3(Y) GAT_PRE_RANK_L(Y) GAT_PRE_WS3(Y)
#define GAT_PRE_TABLE_G3(Y) GAT_PRE_RANK_G(Y) GAT_PRE_WS3(Y)
#define GAT_ONE_SIMPLE_B(Y, C, JJ) GAT_STEP_SIMPLE_B((Y)[C], lr[C], JJ)
#define GAT_ALIVE_B (sL || sP || sO)
#define GAT_ONE_TABLE(Y, C, JJ) GAT_STEP_TABLE_B((Y)[C], lr[C], pcs[C], pce[C], ppe[C], JJ)
#define GAT_ALIVE_TB (sL || sB || sP || sO)
  // (macros, not a lambda taking the step closure: that form kept the closures in scratch memory)
#ifdef GAT_DIAG
  unsigned long long pdg[5] = {0, 0, 0, 0, 0}, pdg_t, pdg_rows = 0, tile_t0;
  asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tile_t0) :: "memory");
  GAT_STAMP(pdg_t);
#define GAT_PROWS pdg_rows += kPlaceChunk;
#else
#define GAT_PROWS
#endif
  // Two loops.  GAT_PLACE_LOOP: two chunk buffers, loads left to the compiler -- which, for a register loaded in one
  // trip and used in the next, waits for EVERYTHING in flight (loads and stores share one in-order counter, vmcnt, and the
  // flush above stores under a lane-dependent condition, so it cannot count): every chunk then exposes a round trip to
  // memory.  The kernel has few waves (every tile of the batch is resident at once, 3-4 per SIMD) and cannot end before
  // the serial chain of the longest unit's tile has: tools/place_scaling.py, config 2: 0.70 ms for 5 000 samples and
  // 0.91 ms for 10 000 with this loop, 0.58 / 0.89 ms with the one below (the floor, 1 250 samples: 0.54 ms).
  // GAT_PLACE_LOOP_PIPE, for the single-workspace-segment loop with its rank table in LDS: four chunk buffers in registers
  // the compiler does not use there (v96..v127: see PIPE above), loads and waits written out by hand.  A chunk's
  // loads are issued three chunks (24 rows) before it is worked on and waited for with vmcnt(24): the 24 loads issued
  // after it may stay in flight (returns are in order; stores issued in between only make the wait stricter).  Every
  // load is always issued (rows beyond the tile's end re-read its last chunk and are not used), so the count holds.
  // (The same with the buffers as asm operands did not work: the compiler copies them around the loop's back edge while
  // loads are on their way into them.)
#ifndef GAT_ROW_NT
#define GAT_ROW_NT " nt"              /* the rows are read once: streamed past the L2's lines (the ring's half-written ones stay) */
#endif
// (timing-only builds, tools/exp_place_bound.sh: GAT_EXP_SAME_ROWS -- every tile of the launch reads the same 64 rows, 16 KB that
//  stay in the caches: the kernel without its row traffic; GAT_EXP_NO_STORES -- the flush's stores left out.  Results are wrong.)
#ifdef GAT_EXP_SAME_ROWS
#define GAT_EXP_ROWPTR(ROW0) (A.rng_out + lane + (int64_t)((int)(ROW0) & 56) * kWave)
#else
#define GAT_EXP_ROWPTR(ROW0) (rp + (int64_t)r0_ * kWave)
#endif
#define GAT_PIN_LOAD(R0, R1, R2, R3, R4, R5, R6, R7, ROW0)                                                     \
  {                                                                                                            \
    const int r0_ = (int)(ROW0) < rows - kPlaceChunk ? (int)(ROW0) : rows - kPlaceChunk;                       \
    (void)r0_;                                                                                                 \
    const uint32_t* p_ = GAT_EXP_ROWPTR(ROW0);                                                                 \
    asm volatile("global_load_dword v" #R0 ", %0, off" GAT_ROW_NT "\n\tglobal_load_dword v" #R1 ", %0, off offset:256" GAT_ROW_NT "\n\t" \
                 "global_load_dword v" #R2 ", %0, off offset:512" GAT_ROW_NT "\n\tglobal_load_dword v" #R3 ", %0, off offset:768" GAT_ROW_NT "\n\t" \
                 "global_load_dword v" #R4 ", %0, off offset:1024" GAT_ROW_NT "\n\tglobal_load_dword v" #R5 ", %0, off offset:1280" GAT_ROW_NT "\n\t" \
                 "global_load_dword v" #R6 ", %0, off offset:1536" GAT_ROW_NT "\n\tglobal_load_dword v" #R7 ", %0, off offset:1792" GAT_ROW_NT \
                 :: "v"(p_) : "memory", "v" #R0, "v" #R1, "v" #R2, "v" #R3, "v" #R4, "v" #R5, "v" #R6, "v" #R7); \
  }
#define GAT_PIN_TAKE(R0, R1, R2, R3, R4, R5, R6, R7)                                                           \
  asm volatile("s_waitcnt vmcnt(24)\n\tv_mov_b32 %0, v" #R0 "\n\tv_mov_b32 %1, v" #R1 "\n\tv_mov_b32 %2, v" #R2 "\n\t" \
               "v_mov_b32 %3, v" #R3 "\n\tv_mov_b32 %4, v" #R4 "\n\tv_mov_b32 %5, v" #R5 "\n\tv_mov_b32 %6, v" #R6 "\n\t" \
               "v_mov_b32 %7, v" #R7                                                                            \
               : "=v"(ya[0]), "=v"(ya[1]), "=v"(ya[2]), "=v"(ya[3]), "=v"(ya[4]), "=v"(ya[5]), "=v"(ya[6]), "=v"(ya[7]) \
               :: "memory");
#define GAT_PLACE_CHUNK(PRE, ONE, K)                                                                         \
  GAT_PSTAMP(0)                                                                                              \
  PRE(ya)                                                                                                    \
  GAT_PSTAMP(1)                                                                                              \
  _Pragma("unroll") for (int c = 0; c < kPlaceChunk; ++c) ONE(ya, c, (uint32_t)(j + (K) * kPlaceChunk + c))  \
  GAT_PSTAMP(2)                                                                                              \
  if ((K) & 1) flush();                /* every second chunk: see flush */                                   \
  GAT_PSTAMP(3)                                                                                              \
  GAT_PROWS
#define GAT_PLACE_LOOP_PIPE(PRE, ONE, ALIVE)                                                                 \
  {                                                                                                          \
    static_assert(kPlaceChunk == 8, "the loads above are written out for chunks of 8");                      \
    asm volatile("; GAT_PINNED_BEGIN" ::: "memory");     /* (markers for tools/check_pinned_regs.py) */      \
    GAT_PIN_LOAD(96, 97, 98, 99, 100, 101, 102, 103, 0)                                                      \
    GAT_PIN_LOAD(104, 105, 106, 107, 108, 109, 110, 111, kPlaceChunk)                                        \
    GAT_PIN_LOAD(112, 113, 114, 115, 116, 117, 118, 119, 2 * kPlaceChunk)                                    \
    for (int j = 0; j < rows; j += 4 * kPlaceChunk) {                                                        \
      if (__ballot(ALIVE) == 0) break;                                                                       \
      GAT_PSTAMP(4)                                                                                          \
      GAT_PIN_LOAD(120, 121, 122, 123, 124, 125, 126, 127, j + 3 * kPlaceChunk)                              \
      GAT_PIN_TAKE(96, 97, 98, 99, 100, 101, 102, 103)                                                       \
      GAT_PLACE_CHUNK(PRE, ONE, 0)                                                                           \
      if (j + 1 * kPlaceChunk >= rows || __ballot(ALIVE) == 0) break;                                        \
      GAT_PSTAMP(4)                                                                                          \
      GAT_PIN_LOAD(96, 97, 98, 99, 100, 101, 102, 103, j + 4 * kPlaceChunk)                                  \
      GAT_PIN_TAKE(104, 105, 106, 107, 108, 109, 110, 111)                                                   \
      GAT_PLACE_CHUNK(PRE, ONE, 1)                                                                           \
      if (j + 2 * kPlaceChunk >= rows || __ballot(ALIVE) == 0) break;                                        \
      GAT_PSTAMP(4)                                                                                          \
      GAT_PIN_LOAD(104, 105, 106, 107, 108, 109, 110, 111, j + 5 * kPlaceChunk)                              \
      GAT_PIN_TAKE(112, 113, 114, 115, 116, 117, 118, 119)                                                   \
      GAT_PLACE_CHUNK(PRE, ONE, 2)                                                                           \
      if (j + 3 * kPlaceChunk >= rows || __ballot(ALIVE) == 0) break;                                        \
      GAT_PSTAMP(4)                                                                                          \
      GAT_PIN_LOAD(112, 113, 114, 115, 116, 117, 118, 119, j + 6 * kPlaceChunk)                              \
      GAT_PIN_TAKE(120, 121, 122, 123, 124, 125, 126, 127)                                                   \
      GAT_PLACE_CHUNK(PRE, ONE, 3)                                                                           \
    }                                                                                                        \
    asm volatile("s_waitcnt vmcnt(0)\n\t; GAT_PINNED_END" ::: "memory");   /* nothing on its way into a register at the end */\
  }
  // the same loop around GAT_STEP_SIMPLE_ASM: the state is lane masks (the loop ends when no lane is in a state), a trip's
  // events are folded into `used` at the next trip's head and behind the loop
#define GAT_ONE_SIMPLE_ASM(C, JJ1, YREG) GAT_STEP_SIMPLE_ASM(YREG, lr[C], JJ1)
#define GAT_ONE_TABLE_ASM(C, JJ1, YREG) GAT_STEP_TABLE_ASM(YREG, lr[C], pcs[C], pce[C], ppe[C], JJ1)
#define GAT_PLACE_CHUNK_ASM(PRE, ONE, K, R0, R1, R2, R3, R4, R5, R6, R7)                                     \
  GAT_PSTAMP(0)                                                                                              \
  if constexpr (GRID) alive_c = (((mL | mP | mO) >> lane) & 1ull) != 0ull;                                   \
  PRE(ya)                                                                                                    \
  GAT_PSTAMP(1)                                                                                              \
  ONE(0, (K) * 8 + 1, R0) ONE(1, (K) * 8 + 2, R1) ONE(2, (K) * 8 + 3, R2) ONE(3, (K) * 8 + 4, R3)            \
  ONE(4, (K) * 8 + 5, R4) ONE(5, (K) * 8 + 6, R5) ONE(6, (K) * 8 + 7, R6) ONE(7, (K) * 8 + 8, R7)            \
  GAT_PSTAMP(2)                                                                                              \
  if ((K) & 1) { nS = (int)(nS9 >> 9); flush(); }                                                            \
  GAT_PSTAMP(3)                                                                                              \
  GAT_PROWS
#define GAT_FOLD_USED(BASE) { used = used_lo != 0u ? (uint32_t)(BASE) + used_lo : used; used_lo = 0u; }
#define GAT_PLACE_LOOP_PIPE_ASM(PRE, ONE)                                                                       \
  {                                                                                                          \
    static_assert(kPlaceChunk == 8, "the loads above are written out for chunks of 8");                      \
    mL = __ballot(sL); mP = 0; mO = 0;                                                                       \
    int jbase = 0;                                                                                           \
    asm volatile("; GAT_PINNED_BEGIN" ::: "memory");                                                         \
    GAT_PIN_LOAD(96, 97, 98, 99, 100, 101, 102, 103, 0)                                                      \
    GAT_PIN_LOAD(104, 105, 106, 107, 108, 109, 110, 111, kPlaceChunk)                                        \
    GAT_PIN_LOAD(112, 113, 114, 115, 116, 117, 118, 119, 2 * kPlaceChunk)                                    \
    for (int j = 0; j < rows; j += 4 * kPlaceChunk) {                                                        \
      if ((mL | mP | mO) == 0) break;                                                                        \
      GAT_PSTAMP(4)                                                                                          \
      GAT_FOLD_USED(jbase)                                                                                   \
      jbase = j;                                                                                             \
      GAT_PIN_LOAD(120, 121, 122, 123, 124, 125, 126, 127, j + 3 * kPlaceChunk)                              \
      GAT_PIN_TAKE(96, 97, 98, 99, 100, 101, 102, 103)                                                       \
      GAT_PLACE_CHUNK_ASM(PRE, ONE, 0, 96, 97, 98, 99, 100, 101, 102, 103)                                                                           \
      if (j + 1 * kPlaceChunk >= rows || (mL | mP | mO) == 0) break;                                         \
      GAT_PSTAMP(4)                                                                                          \
      GAT_PIN_LOAD(96, 97, 98, 99, 100, 101, 102, 103, j + 4 * kPlaceChunk)                                  \
      GAT_PIN_TAKE(104, 105, 106, 107, 108, 109, 110, 111)                                                   \
      GAT_PLACE_CHUNK_ASM(PRE, ONE, 1, 104, 105, 106, 107, 108, 109, 110, 111)                                                                           \
      if (j + 2 * kPlaceChunk >= rows || (mL | mP | mO) == 0) break;                                         \
      GAT_PSTAMP(4)                                                                                          \
      GAT_PIN_LOAD(104, 105, 106, 107, 108, 109, 110, 111, j + 5 * kPlaceChunk)                              \
      GAT_PIN_TAKE(112, 113, 114, 115, 116, 117, 118, 119)                                                   \
      GAT_PLACE_CHUNK_ASM(PRE, ONE, 2, 112, 113, 114, 115, 116, 117, 118, 119)                                                                           \
      if (j + 3 * kPlaceChunk >= rows || (mL | mP | mO) == 0) break;                                         \
      GAT_PSTAMP(4)                                                                                          \
      GAT_PIN_LOAD(112, 113, 114, 115, 116, 117, 118, 119, j + 6 * kPlaceChunk)                              \
      GAT_PIN_TAKE(120, 121, 122, 123, 124, 125, 126, 127)                                                   \
      GAT_PLACE_CHUNK_ASM(PRE, ONE, 3, 120, 121, 122, 123, 124, 125, 126, 127)                                                                           \
    }                                                                                                        \
    asm volatile("s_waitcnt vmcnt(0)\n\t; GAT_PINNED_END" ::: "memory");                                     \
    GAT_FOLD_USED(jbase)                                                                                     \
    nS = (int)(nS9 >> 9);                                                                                    \
    sL = (mL >> lane) & 1; sP = (mP >> lane) & 1; sO = (mO >> lane) & 1;                                     \
  }
  // k_place_grid with eight tiles (MODE 5): the same loop with the look-ups of a chunk ONE CHUNK AHEAD of its steps.  A
  // look-up ends in a 16-byte record from global memory; with the record asked for and used in the same trip every chunk of
  // eight rows exposed a round trip to the L2, and at the two waves per SIMD the LDS image leaves the kernel nothing hides it
  // (refdata: 2.2 ms, 1.1 with the records left out).  Here the rows of chunk k + 1 are taken, searched and their records asked
  // for BEFORE the steps of chunk k run -- two sets of records (A / B, by the chunk's parity) and of taken rows, a register
  // budget only this variant has (two waves per SIMD: 256 registers each), which is why its rows are pinned at v192..v223.
  // Order of a sub-trip: take k + 1 (vmcnt(24): younger than its loads are the two chunks behind it and the eight records of
  // chunk k), search + gathers k + 1, rank lengths k, steps k (the compiler waits for k's records: everything but the eight
  // youngest loads), flush, loads of chunk k + 4 into the buffer chunk k has just left.
#define GAT_PIN_TAKE2(Y, R0, R1, R2, R3, R4, R5, R6, R7)                                                       \
  asm volatile("s_waitcnt vmcnt(24)\n\tv_mov_b32 %0, v" #R0 "\n\tv_mov_b32 %1, v" #R1 "\n\tv_mov_b32 %2, v" #R2 "\n\t" \
               "v_mov_b32 %3, v" #R3 "\n\tv_mov_b32 %4, v" #R4 "\n\tv_mov_b32 %5, v" #R5 "\n\tv_mov_b32 %6, v" #R6 "\n\t" \
               "v_mov_b32 %7, v" #R7                                                                            \
               : "=v"(Y[0]), "=v"(Y[1]), "=v"(Y[2]), "=v"(Y[3]), "=v"(Y[4]), "=v"(Y[5]), "=v"(Y[6]), "=v"(Y[7]) \
               :: "memory");
#define GAT_ONE_TABLE_ASM2(C, JJ1, YREG, PCS, PCE, PPE) GAT_STEP_TABLE_ASM(YREG, lr[C], PCS[C], PCE[C], PPE[C], JJ1)
#define GAT_DEEP_SUB(K, YK, PCSK, PCEK, PPEK, YN, PCSN, PCEN, PPEN, N0, N1, N2, N3, N4, N5, N6, N7,              \
                     R0, R1, R2, R3, R4, R5, R6, R7, NEXTROW)                                                    \
  GAT_PIN_TAKE2(YN, N0, N1, N2, N3, N4, N5, N6, N7)                                                             \
  alive_c = (((mL | mP | mO) >> lane) & 1ull) != 0ull;                                                         \
  GAT_PRE_WS3X(YN, PCSN, PCEN, PPEN)                                                                            \
  GAT_PRE_RANK_L(YK)                                                                                           \
  GAT_ONE_TABLE_ASM2(0, (K) * 8 + 1, R0, PCSK, PCEK, PPEK) GAT_ONE_TABLE_ASM2(1, (K) * 8 + 2, R1, PCSK, PCEK, PPEK) \
  GAT_ONE_TABLE_ASM2(2, (K) * 8 + 3, R2, PCSK, PCEK, PPEK) GAT_ONE_TABLE_ASM2(3, (K) * 8 + 4, R3, PCSK, PCEK, PPEK) \
  GAT_ONE_TABLE_ASM2(4, (K) * 8 + 5, R4, PCSK, PCEK, PPEK) GAT_ONE_TABLE_ASM2(5, (K) * 8 + 6, R5, PCSK, PCEK, PPEK) \
  GAT_ONE_TABLE_ASM2(6, (K) * 8 + 7, R6, PCSK, PCEK, PPEK) GAT_ONE_TABLE_ASM2(7, (K) * 8 + 8, R7, PCSK, PCEK, PPEK) \
  if ((K) & 1) { nS = (int)(nS9 >> 9); flush(); }                                                              \
  GAT_PIN_LOAD(R0, R1, R2, R3, R4, R5, R6, R7, NEXTROW)
#define GAT_PLACE_LOOP_DEEP_ASM                                                                              \
  {                                                                                                          \
    static_assert(kPlaceChunk == 8, "the loads above are written out for chunks of 8");                      \
    mL = __ballot(sL); mP = 0; mO = 0;                                                                       \
    int jbase = 0;                                                                                           \
    uint32_t pcsB[kPlaceChunk], pceB[kPlaceChunk], ppeB[kPlaceChunk];                                        \
    asm volatile("; GAT_PINNED_BEGIN BASE=192" ::: "memory");                                                \
    GAT_PIN_LOAD(192, 193, 194, 195, 196, 197, 198, 199, 0)                                                  \
    GAT_PIN_LOAD(200, 201, 202, 203, 204, 205, 206, 207, kPlaceChunk)                                        \
    GAT_PIN_LOAD(208, 209, 210, 211, 212, 213, 214, 215, 2 * kPlaceChunk)                                    \
    GAT_PIN_LOAD(216, 217, 218, 219, 220, 221, 222, 223, 3 * kPlaceChunk)                                    \
    GAT_PIN_TAKE2(ya, 192, 193, 194, 195, 196, 197, 198, 199)                                                \
    alive_c = sL;                                                                                            \
    GAT_PRE_WS3X(ya, pcs, pce, ppe)                                                                          \
    for (int j = 0; j < rows; j += 4 * kPlaceChunk) {                                                        \
      if ((mL | mP | mO) == 0) break;                                                                        \
      GAT_FOLD_USED(jbase)                                                                                   \
      jbase = j;                                                                                             \
      GAT_DEEP_SUB(0, ya, pcs, pce, ppe, yb, pcsB, pceB, ppeB, 200, 201, 202, 203, 204, 205, 206, 207,        \
                   192, 193, 194, 195, 196, 197, 198, 199, j + 4 * kPlaceChunk)                               \
      if (j + 1 * kPlaceChunk >= rows || (mL | mP | mO) == 0) break;                                         \
      GAT_DEEP_SUB(1, yb, pcsB, pceB, ppeB, ya, pcs, pce, ppe, 208, 209, 210, 211, 212, 213, 214, 215,        \
                   200, 201, 202, 203, 204, 205, 206, 207, j + 5 * kPlaceChunk)                               \
      if (j + 2 * kPlaceChunk >= rows || (mL | mP | mO) == 0) break;                                         \
      GAT_DEEP_SUB(2, ya, pcs, pce, ppe, yb, pcsB, pceB, ppeB, 216, 217, 218, 219, 220, 221, 222, 223,        \
                   208, 209, 210, 211, 212, 213, 214, 215, j + 6 * kPlaceChunk)                               \
      if (j + 3 * kPlaceChunk >= rows || (mL | mP | mO) == 0) break;                                         \
      GAT_DEEP_SUB(3, yb, pcsB, pceB, ppeB, ya, pcs, pce, ppe, 192, 193, 194, 195, 196, 197, 198, 199,        \
                   216, 217, 218, 219, 220, 221, 222, 223, j + 7 * kPlaceChunk)                               \
    }                                                                                                        \
    asm volatile("s_waitcnt vmcnt(0)\n\t; GAT_PINNED_END" ::: "memory");                                     \
    GAT_FOLD_USED(jbase)                                                                                     \
    nS = (int)(nS9 >> 9);                                                                                    \
    sL = (mL >> lane) & 1; sP = (mP >> lane) & 1; sO = (mO >> lane) & 1;                                     \
  }
#define GAT_PLACE_LOOP(PRE, ONE, ALIVE)                                                                        \
  {                                                                                                            \
    const uint32_t* __restrict__ rq = rp;                                                                      \
    _Pragma("unroll") for (int c = 0; c < kPlaceChunk; ++c) ya[c] = rq[c * kWave];                             \
    for (int j = 0; j < rows; j += 2 * kPlaceChunk) {                                                          \
      if (__ballot(ALIVE) == 0) break;                                                                         \
      GAT_PSTAMP(4)                                                                                            \
      const bool more_b = j + kPlaceChunk < rows;                                                              \
      if (more_b) {                                                                                            \
        _Pragma("unroll") for (int c = 0; c < kPlaceChunk; ++c) yb[c] = rq[(kPlaceChunk + c) * kWave];         \
      }                                                                                                        \
      GAT_PSTAMP(0)                                                                                            \
      if constexpr (GRID) alive_c = (ALIVE);                                                                   \
      PRE(ya)                                                                                                  \
      GAT_PSTAMP(1)                                                                                            \
      _Pragma("unroll") for (int c = 0; c < kPlaceChunk; ++c) ONE(ya, c, (uint32_t)(j + c))                    \
      GAT_PSTAMP(2)                                                                                            \
      GAT_PSTAMP(3)                                                                                            \
      GAT_PROWS                                                                                                \
      if (!more_b || __ballot(ALIVE) == 0) break;                                                              \
      if (j + 2 * kPlaceChunk < rows) {                                                                        \
        _Pragma("unroll") for (int c = 0; c < kPlaceChunk; ++c) ya[c] = rq[(2 * kPlaceChunk + c) * kWave];     \
      }                                                                                                        \
      rq += 2 * kPlaceChunk * kWave;                                                                           \
      GAT_PSTAMP(0)                                                                                            \
      if constexpr (GRID) alive_c = (ALIVE);                                                                   \
      PRE(yb)                                                                                                  \
      GAT_PSTAMP(1)                                                                                            \
      _Pragma("unroll") for (int c = 0; c < kPlaceChunk; ++c) ONE(yb, c, (uint32_t)(j + kPlaceChunk + c))      \
      GAT_PSTAMP(2)                                                                                            \
      flush();                                                                                                 \
      GAT_PSTAMP(3)                                                                                            \
      GAT_PROWS                                                                                                \
    }                                                                                                          \
  }
  bool plain_step = true;                    // false: GAT_STEP_SIMPLE_ASM ran -- the pending length and the capacity are settled below
  if constexpr (MODE == 3) {
    static_assert(PIPE, "k_place_wide runs the hand-pipelined loop");
    if (!kind1 && cm_ok) { plain_step = false; GAT_PLACE_LOOP_PIPE_ASM(GAT_PRE_SIMPLE_W, GAT_ONE_SIMPLE_ASM) }
    else GAT_PLACE_LOOP_PIPE(GAT_PRE_SIMPLE_W, GAT_ONE_SIMPLE_B, GAT_ALIVE_B)
  } else if (simple_lds) {
    if constexpr (PIPE) {
      if (!kind1 && cm_ok) { plain_step = false; GAT_PLACE_LOOP_PIPE_ASM(GAT_PRE_SIMPLE_L, GAT_ONE_SIMPLE_ASM) }
      else GAT_PLACE_LOOP_PIPE(GAT_PRE_SIMPLE_L, GAT_ONE_SIMPLE_B, GAT_ALIVE_B)
    } else GAT_PLACE_LOOP(GAT_PRE_SIMPLE_L, GAT_ONE_SIMPLE_B, GAT_ALIVE_B)
  } else if constexpr (!ALL_SIMPLE) {
    if (simple_shape) GAT_PLACE_LOOP(GAT_PRE_SIMPLE_G, GAT_ONE_SIMPLE_B, GAT_ALIVE_B)
    else {
      if (ws_lds) {
        // (not through the pinned registers: a chunk of this loop has more values in flight than the 96 registers below them hold)
        if (rank_lds) {
          if (!kind1 && !drawB && A.place_plain_step == 0) { plain_step = false; GAT_PLACE_LOOP_PIPE_ASM(GAT_PRE_TABLE_LL, GAT_ONE_TABLE_ASM) }
          else GAT_PLACE_LOOP(GAT_PRE_TABLE_LL, GAT_ONE_TABLE, GAT_ALIVE_TB)
        } else GAT_PLACE_LOOP(GAT_PRE_TABLE_GL, GAT_ONE_TABLE, GAT_ALIVE_TB)
      } else if constexpr (TREES) {
        if (rank_lds) GAT_PLACE_LOOP(GAT_PRE_TABLE_LG, GAT_ONE_TABLE, GAT_ALIVE_TB) else GAT_PLACE_LOOP(GAT_PRE_TABLE_GG, GAT_ONE_TABLE, GAT_ALIVE_TB)
      } else if constexpr (GRID) {
        if (rank_lds) {
          if (!kind1 && !drawB && A.place_plain_step == 0) {
            plain_step = false;
            GAT_PLACE_LOOP_DEEP_ASM
          } else GAT_PLACE_LOOP(GAT_PRE_TABLE_L3, GAT_ONE_TABLE, GAT_ALIVE_TB)
        } else GAT_PLACE_LOOP(GAT_PRE_TABLE_G3, GAT_ONE_TABLE, GAT_ALIVE_TB)
      }
    }
  }
#undef GAT_PROWS
#ifdef GAT_DIAG
  if (lane == 0 && A.diag_place != nullptr && sb_own < n_tiles) {
    for (int k = 0; k < 5; ++k) atomicAdd(&A.diag_place[(int64_t)a * 8 + k], pdg[k]);
    atomicAdd(&A.diag_place[(int64_t)a * 8 + 5], pdg_rows);
    atomicAdd(&A.diag_place[(int64_t)a * 8 + 6], 1ull);
    if (A.diag_tiles != nullptr) {
      unsigned long long tile_t1;
      asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tile_t1) :: "memory");
      A.diag_tiles[((int64_t)a * n_tiles + sb_own) * 2] = tile_t0;
      A.diag_tiles[((int64_t)a * n_tiles + sb_own) * 2 + 1] = tile_t1;
    }
  }
#endif
#undef GAT_PLACE_LOOP
#undef GAT_PLACE_LOOP_PIPE
#undef GAT_PLACE_CHUNK
#undef GAT_PIN_TAKE
#undef GAT_PIN_LOAD
#undef GAT_ONE_TABLE
#undef GAT_ONE_SIMPLE_B
#undef GAT_ALIVE_B
#undef GAT_STEP_SIMPLE_B
#undef GAT_STEP_SIMPLE_ASM
#undef GAT_STEP_TABLE_ASM
#undef GAT_ONE_SIMPLE_ASM
#undef GAT_ONE_TABLE_ASM
#undef GAT_PLACE_LOOP_PIPE_ASM
#undef GAT_PLACE_CHUNK_ASM
#undef GAT_FOLD_USED
#undef GAT_PRE_TABLE_GG
#undef GAT_PRE_TABLE_L3
#undef GAT_PRE_TABLE_G3
#undef GAT_PRE_WS3
#undef GAT_PRE_WS3X
#undef GAT_PLACE_LOOP_DEEP_ASM
#undef GAT_DEEP_SUB
#undef GAT_ONE_TABLE_ASM2
#undef GAT_PIN_TAKE2
#undef GAT_PRE_TABLE_LG
#undef GAT_PRE_TABLE_GL
#undef GAT_PRE_TABLE_LL
#undef GAT_PRE_SIMPLE_G
#undef GAT_PRE_SIMPLE_W
#undef GAT_PRE_SIMPLE_L
#undef GAT_PRE_WS2
#undef GAT_PRE_WS
#undef GAT_PRE_RANK_G
#undef GAT_PRE_RANK_L
#undef GAT_STEP_TABLE_B
#undef GAT_ALIVE_TB
  if (live) {
    const bool halted = !(sL || sB || sP || sO);
    if (!plain_step) {
      if (!kind1 && halted) pend = (int32_t)len;                  // (only the trigger halts a lane of that loop)
      if (nS > cap) { flag |= kStatusOverflow; nS = nF; }
    }
    for (int i = nF; i < nS; ++i) out[i] = l_out[i & 15][lane];   // what the last flush left
    // halted at the trigger (pend > 0) or complete (-2); rows ran out with the lane still placing: -3, k_sampler goes on
    // behind the last placement (nS segments, `rem`, `used` outputs); overflow (or SamplerSegments out of rows): -1, the unit
    // is run in full
    A.st[so] = make_int4(nS, rem, flag != 0 ? -1 : (halted ? (pend != -1 ? pend : -1) : (kind1 ? -1 : -3)), (int)used);
    if (flag) atomicOr(A.flags, flag);
  }
}

template <int KIND, int MODE, int SMALL = 0>
// (SMALL 1 -- 10 KB of LDS, sixteen waves per CU -- is asked for four waves per SIMD: without the hint the register that holds
//  spilled scalars lands behind the pinned rows, v128, and a fourth wave no longer fits)
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(MODE == 0 && SMALL == 1 ? 4 : 1))) void k_place(SamplerArgs A) {
  place_body<KIND, MODE, SMALL, false>(A);
}

// the single-workspace-segment units' rows pipelined through v96..v127 by hand (MODE 1, or MODE 0 where such units hold
// most of the working segments: gat_problem::pipe_pays).  (amdgpu_num_vgpr does NOT keep the compiler below v96 on
// gfx950 -- the unified register file doubles the number -- it is kept because the allocation it was tuned with is.)
template <int KIND, int MODE, int SMALL = 0>
__global__ __launch_bounds__(64) __attribute__((amdgpu_num_vgpr(96), amdgpu_waves_per_eu(MODE == 0 && SMALL == 1 ? 4 : 1))) void k_place_pipe(SamplerArgs A) {
  place_body<KIND, MODE, SMALL, true>(A);
}

// MODE 3: kPlaceWide tiles of one unit per workgroup around the unit's rank table in LDS (dynamic: 4 bytes per working segment)
template <int KIND>
__global__ __launch_bounds__(kPlaceWide * 64) void k_place_wide(SamplerArgs A) {
  place_body<KIND, 3, 0, true>(A);
}

// MODE 5 (round 6): problems with workspaces beyond the LDS table (fragmented workspaces: the reference's own test data has
// 6 600 - 21 000 workspace segments per contig).  kPlaceGridTiles = 8 tiles of one unit per workgroup around
// the image of the unit's cdf grid in dynamic LDS (2 bytes per workspace segment + 2 per cell): the position draw's
// searchsorted is a handful of LDS reads and one 16-byte record from global memory, and the table step runs written out on
// the hand-pipelined rows -- k_place<., 2> walked a 16-ary tree in global memory for every random number of a chunk, on the
// compiler's step and loads (refdata, 8 549 segments, 10 000 samples: 7.0 ms)
__global__ __launch_bounds__(kPlaceGridTiles * 64) void k_place_grid(SamplerArgs A) {
  place_body<0, 5, 0, true>(A);
}

// ------------------------------------------------------------------------------------------
// k_place_scan: the same placement loop with ONE STREAM WALKED BY THE 64 LANES OF A WAVE, for small calls of problems of
// simple units (one workspace segment, bucket size 1, rank table within LDS, the offset draw's mask independent of the
// length: cm_ok above).  k_place gives every stream a lane, so a call cannot end before one lane has walked the longest
// unit's ~3 000 rows one by one (0.37 ms on config 2 whatever the sample count); a call of a few hundred samples leaves most
// of the chip idle meanwhile.  Here lane j of a wave takes row j of a chunk of 64 rows of one stream:
//   * whether a row is accepted as a rank / position / offset draw is a function of the row alone (but for the band
//     c_r3 + 1 < y & maskP <= c_r3 + longest length, one row in ~10^5: a chunk with such a row is scanned twice),
//     so a row is a function on the three states L -> P -> O -> L (advance if accepted, else stay) and the state in front of
//     every row is a PREFIX SCAN of function composition -- a function is three bytes holding the images of L, P, O, and a
//     composition is ONE v_perm_b32 with the earlier function as the byte selector (the DPP scan of gat_device.h does seven);
//   * the length in force at an offset draw is the one of the last accepted rank draw in front of it (a ballot, a
//     count-leading-zeros and one ds_bpermute), `remaining` in front of a row the chunk's entry value less the prefix sum of
//     the overlaps placed before it, the trigger (gat/Engine.pyx:582) the first accepted rank draw whose length reaches it;
//   * the placed segments go to the slab compacted by a ballot's population count.
// Rows reach the wave transposed through LDS: a workgroup is a QUARTER TILE -- 16 streams, four waves with four streams
// each -- that loads blocks of 64 rows x 16 lanes (64 contiguous bytes of every row; the other quarters of the tile read
// their shares of the same lines out of the same XCD's L2: blockIdx -> tile is dealt so that a tile's four workgroups have
// one residue mod 8, the stride with which workgroups go round the XCDs) and stores them column-major, a stream's 64 rows
// one conflict-free LDS read.  The next block's loads are in flight while a block is worked on; one barrier per block.
// Hand-over (st, raw segments in the slab, flags) exactly k_place's.
#ifndef GAT_SCAN_WAVES
#define GAT_SCAN_WAVES 8      /* config 2, k_place_scan at 625 / 1 250 / 2 500 samples: 4 waves 0.142 / 0.205 / 0.360 ms, 8 waves 0.123 / 0.209 / 0.393, 16: 0.24 / 0.45 / 0.86 */
#endif
constexpr int kScanStreams = 16, kScanWaves = GAT_SCAN_WAVES, kScanPer = kScanStreams / kScanWaves, kScanPitch = kWave + 1;
constexpr int kScanRankTab = kPlaceRankLds;
constexpr int kFnIdentity = 0x03020100;      // byte s = the image of state s (0 L, 1 P, 2 O; byte 3 is never looked at)

// h = a after b: byte s of h = byte b(s) of a -- v_perm_b32 with b as the selector (selector values 0..3 pick bytes of the
// second source)
__device__ __forceinline__ int scan_fn_compose(int a, int b) {
  return (int)__builtin_amdgcn_perm(0u, (uint32_t)a, (uint32_t)b);
}

struct ScanStream {
  uint32_t state8;       // 8 x the state in front of the next row (the bit offset of its byte in a function): 0 L, 8 P, 16 O
  uint32_t len;          // the length in force (drawn by the last accepted rank draw)
  int32_t rem;           // `remaining`
  int32_t nS;            // segments placed
  uint32_t used;         // raw outputs consumed up to the last placement / the trigger
  int32_t pend;          // > 0: triggered with this length pending; -1 not (yet)
  int32_t flag;
  bool done;             // halted (trigger, overflow) or not a stream of the batch
};

struct ScanUnit {
  uint32_t maskL, rangeL, maskP, rangeP, c_r3;
  int32_t c_ss, ws_start, ws_end, cap;
};

// one chunk of up to 64 rows of one stream, lane j = row j (rows at and beyond nrows do nothing).
// The offset draw's range is c_r3 + length: a row is accepted for certain up to c_r3 + 1 (every length is at least 1) and
// in the band above it -- one row in ~10^5 -- only for lengths that reach it.  The scan starts from "accepted for certain";
// once the states and the lengths in force are known, every row IN STATE O is tested exactly, and if one of them was guessed
// wrong the scan is repeated with the exact answers.  (The first row whose state differs from the truth is preceded by a
// wrongly guessed row in state O whose length in force is already the true one, so a pass without a wrong guess is the
// truth; every pass extends the correct prefix.)  pessimist (GAT_PLACE_SCAN_SEQ, tests): start from "none accepted".
__device__ __forceinline__ void place_scan_chunk(ScanStream& T, const ScanUnit& U, uint32_t y, uint32_t lr, int nrows, uint32_t row0,
                                                 uint2* __restrict__ out, int lane, bool pessimist) {
  const bool in = lane < nrows;
  const uint32_t vO = y & U.maskP;
  const bool aL = in && (y & U.maskL) <= U.rangeL, aP = in && vO <= U.rangeP;
  bool aO = in && !pessimist && vO <= U.c_r3 + 1u;
  const uint64_t lt = lanemask_lt(lane), le = lt | (1ull << lane);
  const int fLP = 0x03000000 | (aL ? 1 : 0) | ((aP ? 2 : 1) << 8);
  int Fi;
  uint32_t len;
  bool isL, isO;
  uint64_t bL;
  for (;;) {
    const int f = fLP | ((aO ? 0 : 2) << 16);
    Fi = wave_incl_scan_dpp(f, kFnIdentity, [](int a, int b) { return scan_fn_compose(a, b); });
    const int Fe = __builtin_amdgcn_update_dpp(kFnIdentity, Fi, 0x138, 0xf, 0xf, false);        // wave_shr:1: the rows in front
    const uint32_t st = __builtin_amdgcn_ubfe((uint32_t)Fe, T.state8, 8);                         // the state in front of row j
    isL = aL && st == 0u;
    bL = __ballot(isL);
    // the length in force: the last accepted rank draw in front of (or at) this row, else the chunk's entry value
    const uint64_t mL = bL & le;
    const int src = mL ? 63 - __builtin_clzll(mL) : lane;
    const uint32_t got = (uint32_t)__builtin_amdgcn_ds_bpermute(src << 2, (int)lr);
    len = mL ? got : T.len;
    const bool exact = in && vO <= U.c_r3 + len;
    isO = aO && st == 2u;
    if (!__any(st == 2u && exact != aO)) break;
    aO = st == 2u ? exact : aO;
  }
  const int32_t q = U.c_ss - (int32_t)len + (int32_t)vO;
  const uint32_t start = (uint32_t)(q > 0 ? q : 0), end = (uint32_t)(q + (int32_t)len);
  const int32_t omin = U.ws_end < (int32_t)end ? U.ws_end : (int32_t)end, omax = U.ws_start > (int32_t)start ? U.ws_start : (int32_t)start;
  const uint32_t ov = isO ? (uint32_t)(omin - omax > 0 ? omin - omax : 0) : 0u;
  const uint32_t incl = wave_incl_sum_u32(ov, lane);
  const int32_t rem_before = T.rem - (int32_t)(incl - ov);
  const bool trig = isL && rem_before <= (int32_t)lr;                       // gat/Engine.pyx:582
  const uint64_t bT = __ballot(trig);
  const int jt = bT ? __builtin_ctzll(bT) : kWave - 1;                      // rows behind the first trigger do not happen
  const uint64_t keep = bT ? ((1ull << jt) - 1ull) : ~0ull;
  const uint64_t bO = __ballot(isO) & keep;
  const int idx = T.nS + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(bO >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bO, 0u));
  const int nput = __popcll(bO);
  const bool over = T.nS + nput > U.cap;                                    // (k_place: the unit is run in full with a larger slab)
  if (!over && ((bO >> lane) & 1ull)) out[idx] = make_uint2(start, end);
  // the stream behind the chunk: at the trigger (row jt) / at the chunk's last row
  const int32_t rem_t = __builtin_amdgcn_readlane(rem_before, jt), lr_t = __builtin_amdgcn_readlane((int)lr, jt);
  const int32_t tot = __builtin_amdgcn_readlane((int)incl, kWave - 1);
  const uint32_t len_end = bL ? (uint32_t)__builtin_amdgcn_readlane((int)lr, 63 - __builtin_clzll(bL)) : T.len;
  const uint32_t st_end = 8u * __builtin_amdgcn_ubfe((uint32_t)__builtin_amdgcn_readlane(Fi, kWave - 1), T.state8, 8);
  const uint32_t used_put = bO ? row0 + (uint32_t)(63 - __builtin_clzll(bO)) + 1u : T.used;
  T.flag |= over ? kStatusOverflow : 0;
  T.done = over || bT != 0;
  if (!over) {
    T.nS += nput;
    T.used = bT ? row0 + (uint32_t)jt + 1u : used_put;
    T.pend = bT ? lr_t : T.pend;
    T.rem = bT ? rem_t : T.rem - tot;
    T.len = len_end;
    T.state8 = st_end;
  }
}

__global__ __launch_bounds__(kScanWaves * 64) void k_place_scan(SamplerArgs A, int n_sb, int pessimist) {
  __shared__ uint32_t l_rows[2][kScanStreams * kScanPitch];
  __shared__ uint32_t l_rank[kScanRankTab];
  __shared__ int32_t l_live[2][kScanWaves];
  const int lane = threadIdx.x & (kWave - 1), wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  // tile = (launch position a, sample block sb), largest units first; the four quarters of a tile on one XCD
  const int64_t n_tiles = (int64_t)n_sb * A.n_active;
  const int64_t b = blockIdx.x, slot = b >> 3;
  const int64_t tile = (slot >> 2) * 8 + (b & 7);
  const int quarter = (int)(slot & 3);
  if (tile >= n_tiles) return;
  const int a = (int)(tile / n_sb), sb = (int)(tile % n_sb);
  const UnitDev* __restrict__ Up = A.units_o + a;
  const uint32_t hist_total = Up->hist_total, ws_total = Up->ws_total;
  const int rows = A.rng_rows[a];
  const uint2 ws0 = A.ws[Up->ws_off];
  const uint32_t* __restrict__ rank_len = A.rank_len + Up->rank_off;
  const int col0 = quarter * kScanStreams;
  ScanUnit U;
  ScanStream T[kScanPer];
  uint2* outp[kScanPer];
  int64_t so[kScanPer];
  bool live[kScanPer];
  const bool degenerate = !(hist_total > 2 && ws_total > 1);              // k_sampler runs those from their seed (as k_place)
#pragma unroll
  for (int q = 0; q < kScanPer; ++q) {
    const int sidx = sb * kWave + col0 + wv * kScanPer + q;
    live[q] = sidx < A.batch;
    so[q] = GAT_REC(A, live[q] ? sidx : 0, a);
    outp[q] = A.slab + (int64_t)(live[q] ? sidx : 0) * A.slab_stride + Up->slab_off;
    T[q].state8 = 0u; T[q].len = 0u; T[q].rem = Up->ltotal; T[q].nS = 0; T[q].used = 0u; T[q].pend = -1; T[q].flag = 0;
    T[q].done = !live[q];
  }
  if (degenerate) {
#pragma unroll
    for (int q = 0; q < kScanPer; ++q)
      if (live[q] && lane == 0) A.st[so[q]] = make_int4(0, Up->ltotal, -1, 0);
    return;
  }
  U.rangeL = hist_total - 2u; U.maskL = 0xffffffffu >> __builtin_clz(U.rangeL);
  U.rangeP = ws_total - 1u; U.maskP = 0xffffffffu >> __builtin_clz(U.rangeP);
  U.c_ss = (int32_t)ws0.x + 1; U.c_r3 = ws0.y - ws0.x - 2u;
  U.ws_start = (int32_t)ws0.x; U.ws_end = (int32_t)ws0.y; U.cap = Up->slab_cap;
  for (int i = (int)threadIdx.x; i <= (int)U.maskL && i < kScanRankTab; i += kScanWaves * kWave)
    l_rank[i] = (uint32_t)i <= U.rangeL ? rank_len[i + 1] : 0u;
  // rows of the tile: rp[j * 64 + lane of the stream]; a load instruction takes 4 rows x this quarter's 16 lanes
  const uint32_t* __restrict__ rp = A.rng_out + A.rng_off[a] + (int64_t)sb * rows * kWave + col0;
  const int lrow = lane >> 4, lcol = lane & 15;
  const int nblk = (rows + kWave - 1) / kWave;
  // (a block is 16 groups of 4 rows x 16 lanes; a wave loads kScanPer of them)
  uint32_t ld[kScanPer];
  auto load_block = [&](int blk) {
#pragma unroll
    for (int i = 0; i < kScanPer; ++i) {
      int r = blk * kWave + (wv * kScanPer + i) * 4 + lrow;
      r = r < rows ? r : rows - 1;
      ld[i] = rp[(int64_t)r * kWave + lcol];
    }
  };
  auto store_block = [&](int buf) {
#pragma unroll
    for (int i = 0; i < kScanPer; ++i) l_rows[buf][lcol * kScanPitch + (wv * kScanPer + i) * 4 + lrow] = ld[i];
  };
  load_block(0);
  store_block(0);
  __syncthreads();
  for (int blk = 0; blk < nblk; ++blk) {
    const int buf = blk & 1;
    if (blk + 1 < nblk) load_block(blk + 1);
    const int nrows = rows - blk * kWave < kWave ? rows - blk * kWave : kWave;
    int nlive = 0;
#pragma unroll
    for (int q = 0; q < kScanPer; ++q) {
      if (!T[q].done) {
        const uint32_t y = l_rows[buf][(wv * kScanPer + q) * kScanPitch + lane];
        const uint32_t lr = l_rank[y & U.maskL];
        place_scan_chunk(T[q], U, y, lr, nrows, (uint32_t)(blk * kWave), outp[q], lane, pessimist != 0);
      }
      nlive += T[q].done ? 0 : 1;
    }
    if (lane == 0) l_live[buf][wv] = nlive;
    if (blk + 1 < nblk) store_block(buf ^ 1);
    __syncthreads();
    int all_live = 0;
#pragma unroll
    for (int w = 0; w < kScanWaves; ++w) all_live += l_live[buf][w];
    if (all_live == 0) break;
  }
#pragma unroll
  for (int q = 0; q < kScanPer; ++q) {
    if (!live[q]) continue;
    // halted at the trigger: the pending length; rows ran out with the stream still placing: -3, k_sampler goes on behind the
    // last placement; overflow: -1, the unit is run in full (k_place's record)
    if (lane == 0) A.st[so[q]] = make_int4(T[q].nS, T[q].rem, T[q].flag != 0 ? -1 : (T[q].pend > 0 ? T[q].pend : -3), (int)T[q].used);
    if (T[q].flag != 0 && lane == 0) atomicOr(A.flags, T[q].flag);
  }
}

// ------------------------------------------------------------------------------------------
// k_merge_big: the FIRST consolidation (sort + merge(0) + workspace coverage, gat/Engine.pyx:582-606) of the long
// lists, one workgroup of 256 threads per (sample, unit).  A single wave needs dozens of rounds per pass over a list of
// thousands of segments and, with the list in LDS, only two or three such waves fit a CU; here four waves share the list
// and the passes: counting sort from the slab k_place wrote into LDS (position buckets, LDS atomics, block prefix sum,
// one thread per bucket for the final order), merge(0) as a block-wide running maximum of the ends, the merged list
// written back to the slab, coverage and total length reduced over the block.  k_sampler resumes behind it with a
// clean merged list (st2) and goes straight to the tail.  Lists the counting sort declines (clustered keys) are left
// to k_sampler's own sort.
constexpr int kSortScratchWords = 528;   // bucket-sort scratch of a wave (513 words, padded to 16 bytes)
#ifndef GAT_MERGE_THREADS
#define GAT_MERGE_THREADS 512
#endif
constexpr int kMergeThreads = GAT_MERGE_THREADS;
constexpr int kMergeWaves = kMergeThreads / kWave;
#ifndef GAT_MERGE_RANK
#define GAT_MERGE_RANK 4
#endif
constexpr int kMergeRank = GAT_MERGE_RANK;            // ... and look-ups of the order inside the buckets it keeps in flight
constexpr int kMergeRegs = 24;           // list elements a thread of k_merge_big<., REGS> keeps in registers at most (lists of up to 12 288)

__device__ __forceinline__ uint32_t block_reduce_u32(uint32_t v, uint32_t* red, int tid, bool want_max, bool want_min) {
  const int lane = tid & 63, wave = tid >> 6;
  uint32_t w = want_max ? wave_max_u32(v) : (want_min ? wave_min_u32(v) : wave_total_u32(v));
  __syncthreads();
  if (lane == 0) red[wave] = w;
  __syncthreads();
  uint32_t r = red[0];
  for (int k = 1; k < kMergeWaves; ++k) {
    const uint32_t x = red[k];
    r = want_max ? (x > r ? x : r) : (want_min ? (x < r ? x : r) : r + x);
  }
  return r;
}

template <bool TREE, int REGS>
__global__ __launch_bounds__(kMergeThreads) void k_merge_big(SamplerArgs A) {
  extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
  __shared__ uint32_t red[kMergeWaves];
  __shared__ int32_t redi[kMergeWaves];
  __shared__ uint32_t red2[kMergeWaves];
  __shared__ int32_t redi2[kMergeWaves];
  __shared__ uint32_t wsl[2 * kWsTreeMin];                       // short workspaces: starts, ends
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int sidx = blockIdx.x, a = A.a_base + (int)(blockIdx.y + blockIdx.z * gridDim.y);
  if (a >= A.n_long || (A.a_end > 0 && a >= A.a_end)) return;
  const UnitDev* __restrict__ Up = A.units_o + a;
  const int64_t sa = GAT_REC(A, sidx, a);
  const int4 pre = A.st[sa];
  const int n = pre.x;
  if (tid == 0) A.st2[sa] = make_int4(0, 0, 0, 0);
  if (pre.z < 0 || n <= 1024 || n > A.lds_cap) return;          // not handed over by k_place / short list: k_sampler does it
#ifdef GAT_DIAG_CONS
  // (tools/diag_consolidate.sh: a workgroup's cycles per phase -- 0 record + workspace, 1 histogram pass over the slab, 2 prefix,
  //  3 scatter pass over the slab into LDS, 4 buckets sorted thread by thread, 5 merge(0), 6 coverage, 7 running lengths; a
  //  barrier and a drained wave in front of every stamp)
  unsigned long long dg[8] = {0, 0, 0, 0, 0, 0, 0, 0}, dg_t;
#define GAT_MSTAMP(T) { __syncthreads(); asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(T) :: "memory"); }
  GAT_MSTAMP(dg_t);
#define GAT_MPHASE(K) { unsigned long long t__; GAT_MSTAMP(t__); dg[K] += t__ - dg_t; dg_t = t__; }
#else
#define GAT_MPHASE(K)
#endif
  uint2* seg = reinterpret_cast<uint2*>(lds);                     // lds_cap entries
  // REGS: the sort works on the starts alone (4 bytes an element: the lower half of the list's LDS), the histogram lives in the
  // upper half -- as many buckets as elements for nothing -- and the elements come in from the registers when both are done with
  uint32_t* hist = REGS > 0 ? lds + (size_t)A.lds_cap : lds + 2 * (size_t)A.lds_cap;     // big_buckets + 1
  uint2* out = A.slab + (int64_t)sidx * A.slab_stride + Up->slab_off;
  const int nws = Up->n_ws;
  const uint2* __restrict__ ws = A.ws + Up->ws_off;
  const uint32_t* __restrict__ ws_cdf = A.ws_cdf + Up->ws_off;
  if (nws <= kWsTreeMin)
    for (int k = tid; k < nws; k += kMergeThreads) { const uint2 w = ws[k]; wsl[k] = w.x; wsl[kWsTreeMin + k] = w.y; }

  // ---- counting sort by position bucket: src = slab, dst = LDS
  constexpr int kB = 8;                                           // rounds of loads in flight (one block per CU: nothing else hides them)
  // the range of the starts from the unit's workspace instead of a pass over the list: a placed segment starts at
  // max(0, q) with q >= workspace start - length + 1 and q < workspace end (gat/Engine.pyx:318-331); the buckets only
  // have to be monotone in the start
  uint32_t lo, hi;
  {
    const uint32_t max_len = A.rank_len[Up->rank_off + Up->hist_total] * Up->bucket + Up->bucket;
    const uint32_t w0 = ws[0].x, w1 = ws[nws - 1].y;
    lo = w0 > max_len ? w0 - max_len : 0u;
    hi = w1;
  }
  const uint32_t span = hi - lo;
  if (span == 0) return;
  int nb = 1024;
  while (nb < n && nb < A.big_buckets) nb <<= 1;
  const bool direct = span < (uint32_t)nb;
  const uint32_t scale = direct ? 0u : (uint32_t)(((uint64_t)nb << 32) / ((uint64_t)span + 1u));
  // (REGS: a word of padding behind every thread's buckets -- the prefix pass below reads them thread by thread, and a stride of
  //  16 words put a whole wave on two LDS banks)
  const int per_b = nb / kMergeThreads;
  const int per_shift = __builtin_ctz((unsigned)per_b);
  auto hidx = [&](uint32_t b) -> uint32_t { return REGS > 0 ? b + (b >> per_shift) : b; };
  if constexpr (REGS == 0) {
    for (int i = tid; i <= nb; i += kMergeThreads) hist[i] = 0;
    __syncthreads();
    GAT_MPHASE(0)
  }
  int count = 0;
  uint32_t cov = 0, tot = 0;
  auto bucket_of = [&](uint32_t x) -> uint32_t { const uint32_t d = x - lo; return direct ? d : __umulhi(d, scale); };
  if constexpr (REGS > 0) {
    if (n > REGS * kMergeThreads) return;                          // (the host picks REGS for the class's longest list)
    // (round 6) The list is read ONCE, into registers (a thread keeps elements tid, tid + 512, ...), and stays there through
    // histogram, scatter and the order inside the buckets: every ELEMENT counts the starts of its bucket in front of it (four
    // look-ups side by side; a thread per BUCKET running an insertion sort was 36 % of the kernel: sixteen dependent rounds, each
    // as long as its wave's fullest bucket).  The histogram's atomic hands every element its place of arrival in its bucket, so
    // the scatter needs no second one.  merge(0) then runs wave by wave on contiguous eighths of the list, four elements a lane
    // -- its running maximum handed from wave to wave through two barriers in all (a barrier four times per 512 elements was
    // another 35 %) --, compacts in LDS, and the copy to the slab adds up coverage and total length on its way.
    // (every store into the register arrays is unconditional: a conditional one makes the compiler carry the whole array as one
    //  value through the branch -- hundreds of spilled registers)
    uint2 v[REGS];
    int at[REGS];
#pragma unroll
    for (int q = 0; q < REGS; ++q) {
      const int i = q * kMergeThreads + tid;
      v[q] = out[i < n ? i : n - 1];
    }
    for (int i = tid; i <= nb + kMergeThreads; i += kMergeThreads) hist[i] = 0;       // (while the list is on its way)
    __syncthreads();
    GAT_MPHASE(0)
#pragma unroll
    for (int q = 0; q < REGS; ++q) {
      int c_ = 0;
      if (q * kMergeThreads + tid < n) c_ = (int)atomicAdd(&hist[hidx(bucket_of(v[q].x))], 1u);
      at[q] = c_;                                                  // place of arrival in its bucket
    }
    __syncthreads();
    GAT_MPHASE(1)
    {
      uint32_t sum = 0, maxc = 0;
      uint32_t* hb_ = hist + tid * (per_b + 1);                    // this thread's buckets (per_b of them, then the padding)
      for (int q = 0; q < per_b; ++q) { const uint32_t c = hb_[q]; sum += c; maxc = c > maxc ? c : maxc; }
      const uint32_t incl = wave_incl_sum_u32(sum, lane);
      maxc = block_reduce_u32(maxc, red, tid, true, false);
      if (maxc > 48u) return;                                     // clustered: left to k_sampler's own sort
      __syncthreads();
      if (lane == 63) red[wave] = incl;
      __syncthreads();
      uint32_t run = incl - sum;
      for (int k = 0; k < wave; ++k) run += red[k];
      for (int q = 0; q < per_b; ++q) { const uint32_t c = hb_[q]; hb_[q] = run; run += c; }
      if (tid == kMergeThreads - 1) hist[hidx((uint32_t)nb)] = run;       // (= n: the end of the last bucket)
    }
    __syncthreads();
    GAT_MPHASE(2)
    // hist[b] is the START of bucket b.  The starts go to where they arrived; an element remembers its bucket's first slot (14
    // bits), length (6) and its own place of arrival (6)
#pragma unroll
    for (int q = 0; q < REGS; ++q) {
      int w_ = 0;
      if (q * kMergeThreads + tid < n) {
        const uint32_t b = bucket_of(v[q].x);
        const int s0 = (int)hist[hidx(b)], e = (int)hist[hidx(b + 1u)];
        lds[s0 + at[q]] = v[q].x;
        w_ = s0 | ((e - s0) << 14) | (at[q] << 20);
      }
      at[q] = w_;
    }
    __syncthreads();
    GAT_MPHASE(3)
    // place of an element among its bucket's: by start, then by arrival
#pragma unroll
    for (int q0 = 0; q0 < REGS; q0 += kMergeRank) {
      int rk[kMergeRank];
      int mx = 0;
#pragma unroll
      for (int r = 0; r < kMergeRank; ++r) { const int len = (at[q0 + r] >> 14) & 63; rk[r] = 0; mx = len > mx ? len : mx; }
      if (mx > 1) {
#pragma unroll 1
        for (int k = 0; k < mx; ++k) {
#pragma unroll
          for (int r = 0; r < kMergeRank; ++r) {
            const int s0 = at[q0 + r] & 0x3fff, len = (at[q0 + r] >> 14) & 63, c = at[q0 + r] >> 20;
            const uint32_t x = lds[s0 + (k < len ? k : 0)];
            rk[r] += (k < len && (x < v[q0 + r].x || (x == v[q0 + r].x && k < c))) ? 1 : 0;
          }
        }
      }
#pragma unroll
      for (int r = 0; r < kMergeRank; ++r) at[q0 + r] = (at[q0 + r] & 0x3fff) + rk[r];
    }
    __syncthreads();                                               // every look-up done before the elements come in (over starts and histogram)
#pragma unroll
    for (int q = 0; q < REGS; ++q)
      if (q * kMergeThreads + tid < n) seg[at[q]] = v[q];
    __syncthreads();
    GAT_MPHASE(4)

    // ---- merge(0) (gat/SegmentList.pyx:756-816): head = first non-empty, or int32(start) > running max end
    constexpr int kE = 4;                                          // consecutive elements a lane takes per round
    const int chunk = ((n + kMergeWaves * kWave * kE - 1) / (kMergeWaves * kWave * kE)) * kWave * kE;   // a wave's elements: [w0, w1)
    const int w0 = wave * chunk < n ? wave * chunk : n, w1 = w0 + chunk < n ? w0 + chunk : n;
    auto load4 = [&](int base, uint2* x) {
#pragma unroll
      for (int t = 0; t < kE; ++t) { const int i = base + lane * kE + t; x[t] = i < w1 ? seg[i] : make_uint2(0u, 0u); }
    };
    {
      int32_t wm = INT32_MIN;
      for (int base = w0; base < w1; base += kWave * kE) {
        uint2 x[kE];
        load4(base, x);
#pragma unroll
        for (int t = 0; t < kE; ++t) if (x[t].x != x[t].y) wm = (int32_t)x[t].y > wm ? (int32_t)x[t].y : wm;
      }
      const uint64_t wany = __ballot(wm != INT32_MIN);             // (a valid end is a coordinate: never INT32_MIN)
      wm = (int32_t)wave_max_u32((uint32_t)wm ^ 0x80000000u) ^ (int32_t)0x80000000;
      if (lane == 0) { redi[wave] = wm; red[wave] = wany != 0 ? 1u : 0u; }
    }
    __syncthreads();
    int32_t carry = INT32_MIN, total = INT32_MIN;
    bool any = false;
    for (int k = 0; k < kMergeWaves; ++k) {
      const int32_t x = redi[k];
      if (k < wave) { carry = x > carry ? x : carry; any = any || red[k] != 0; }
      total = x > total ? x : total;
    }
    int cnt = 0;
    int32_t first_excl = 0;
    uint32_t* sw = reinterpret_cast<uint32_t*>(seg);
    uint2 nx[kE];
    load4(w0, nx);
    for (int base = w0; base < w1; base += kWave * kE) {
      uint2 x[kE];
#pragma unroll
      for (int t = 0; t < kE; ++t) x[t] = nx[t];
      load4(base + kWave * kE, nx);        // (the next round, ahead of this one's stores: they land at or before this round's slots)
      // the running maximum of the ends inside the lane, then over the lanes in front of it
      int32_t lm[kE];
      bool lv[kE];                                                 // a valid (non-empty) element at or before t in this lane
#pragma unroll
      for (int t = 0; t < kE; ++t) {
        const bool valid = x[t].x != x[t].y;                       // (beyond the chunk: [0, 0))
        const int32_t e = valid ? (int32_t)x[t].y : INT32_MIN;
        lm[t] = t == 0 ? e : (e > lm[t - 1] ? e : lm[t - 1]);
        lv[t] = t == 0 ? valid : (valid || lv[t - 1]);
      }
      const int32_t sc = wave_incl_max_i32(lm[kE - 1], lane);
      int32_t ex = __builtin_amdgcn_update_dpp(INT32_MIN, sc, 0x138, 0xf, 0xf, false);      // wave_shr:1, lane 0: nothing in front
      ex = ex > carry ? ex : carry;
      const uint64_t vb = __ballot(lv[kE - 1]);
      const bool pv = any || (vb & lanemask_lt(lane)) != 0;        // a valid element in front of this lane
      bool head[kE];
      int32_t excl[kE];
      int h = 0;
      int32_t fe = 0;
#pragma unroll
      for (int t = 0; t < kE; ++t) {
        excl[t] = t == 0 ? ex : (lm[t - 1] > ex ? lm[t - 1] : ex);
        const bool prev_valid = pv || (t > 0 && lv[t - 1]);
        head[t] = x[t].x != x[t].y && (!prev_valid || (int32_t)x[t].x > excl[t]);
        if (head[t] && h == 0) fe = excl[t];
        h += head[t] ? 1 : 0;
      }
      const uint32_t ps = wave_incl_sum_u32((uint32_t)h, lane);
      const uint64_t hb = __ballot(h > 0);
      if (cnt == 0 && hb != 0) first_excl = __builtin_amdgcn_readlane(fe, (int)__builtin_ctzll(hb));
      // (compacted where the wave's elements stand: a head's slot is at or before its element, and the round is in registers)
      int p_ = w0 + cnt + (int)ps - h;
#pragma unroll
      for (int t = 0; t < kE; ++t) {
        if (head[t]) {
          sw[2 * p_] = x[t].x;
          if (p_ > w0) sw[2 * (p_ - 1) + 1] = (uint32_t)excl[t];
          p_++;
        }
      }
      cnt += __builtin_amdgcn_readlane((int)ps, kWave - 1);
      const int32_t top = __builtin_amdgcn_readlane(sc, kWave - 1);
      carry = top > carry ? top : carry;
      any = any || vb != 0;
      wave_fence();
    }
    if (lane == 0) { red2[wave] = (uint32_t)cnt; redi2[wave] = first_excl; }
    __syncthreads();
    int offset = 0;
    int32_t last_end = total;                                      // end of the wave's last merged segment: up to the next head
    bool next_seen = false;
    for (int k = 0; k < kMergeWaves; ++k) {
      const int c = (int)red2[k];
      if (k < wave) offset += c;
      if (k > wave && c > 0 && !next_seen) { last_end = redi2[k]; next_seen = true; }
      count += c;
    }
    if (cnt > 0 && lane == 0) sw[2 * (w0 + cnt - 1) + 1] = (uint32_t)last_end;
    wave_fence();
    GAT_MPHASE(5)
    // ---- the merged list to the slab; its coverage inside the workspace and its total length on the way
    const uint32_t* __restrict__ pg = A.ws_tree + (TREE && Up->pgrid_off >= 0 ? Up->pgrid_off : 0);
    const uint32_t pshift = TREE && nws > kWsTreeMin ? pg[0] : 0u, pcells = TREE && nws > kWsTreeMin ? pg[1] : 0u;
    for (int j = lane; j < cnt; j += kWave) {
      const uint2 x = seg[w0 + j];
      out[offset + j] = x;
      tot += x.y - x.x;
      if (nws <= kWsTreeMin) {
        for (int k = 0; k < nws; ++k) {
          const uint32_t l2 = x.x > wsl[k] ? x.x : wsl[k], h2 = x.y < wsl[kWsTreeMin + k] ? x.y : wsl[kWsTreeMin + k];
          cov += h2 > l2 ? h2 - l2 : 0u;
        }
      } else if constexpr (TREE) {
        cov += ws_overlap_pgrid(ws, nws, pg + kGridHeader, pshift, pcells, x.x, x.y);
      }
    }
    __syncthreads();                                                // the merged list in the slab, visible to the block
  } else {
    for (int base = 0; base < n; base += kB * kMergeThreads) {
      uint32_t x[kB];
  #pragma unroll
      for (int q = 0; q < kB; ++q) { const int i = base + q * kMergeThreads + tid; x[q] = i < n ? out[i].x : 0u; }
  #pragma unroll
      for (int q = 0; q < kB; ++q)
        if (base + q * kMergeThreads + tid < n) { const uint32_t d = x[q] - lo; atomicAdd(&hist[direct ? d : __umulhi(d, scale)], 1u); }
    }
    __syncthreads();
    GAT_MPHASE(1)
    {
      // exclusive prefix over the buckets: thread owns nb/256 consecutive ones
      const int per = nb / kMergeThreads;
      uint32_t sum = 0, maxc = 0;
      for (int q = 0; q < per; ++q) { const uint32_t c = hist[tid * per + q]; sum += c; maxc = c > maxc ? c : maxc; }
      const uint32_t incl = wave_incl_sum_u32(sum, lane);
      maxc = block_reduce_u32(maxc, red, tid, true, false);         // (two barriers: red is free again below)
      if (maxc > 48u) return;                                       // clustered: not worth sorting buckets thread by thread
      __syncthreads();                                              // everybody has read red
      if (lane == 63) red[wave] = incl;
      __syncthreads();
      uint32_t run = incl - sum;
      for (int k = 0; k < wave; ++k) run += red[k];
      for (int q = 0; q < per; ++q) { const uint32_t c = hist[tid * per + q]; hist[tid * per + q] = run; run += c; }
    }
    __syncthreads();
    GAT_MPHASE(2)
    for (int base = 0; base < n; base += kB * kMergeThreads) {
      uint2 v[kB];
  #pragma unroll
      for (int q = 0; q < kB; ++q) { const int i = base + q * kMergeThreads + tid; v[q] = i < n ? out[i] : make_uint2(0u, 0u); }
  #pragma unroll
      for (int q = 0; q < kB; ++q)
        if (base + q * kMergeThreads + tid < n) {
          const uint32_t d = v[q].x - lo;
          seg[atomicAdd(&hist[direct ? d : __umulhi(d, scale)], 1u)] = v[q];
        }
    }
    __syncthreads();
    GAT_MPHASE(3)
    for (int b = tid; b < nb; b += kMergeThreads) {                // hist[b] is now the END of bucket b
      const int e = (int)hist[b], s0 = b ? (int)hist[b - 1] : 0;
      for (int i = s0 + 1; i < e; ++i) {
        const uint2 v = seg[i];
        int j = i - 1;
        while (j >= s0 && seg[j].x > v.x) { seg[j + 1] = seg[j]; --j; }
        seg[j + 1] = v;
      }
    }
    __syncthreads();
    GAT_MPHASE(4)

    // ---- merge(0) (gat/SegmentList.pyx:756-816): head = first non-empty, or int32(start) > running max end
    int32_t carry = INT32_MIN;
    bool any = false;
    for (int base = 0; base < n; base += kMergeThreads) {
      const int i = base + tid;
      uint32_t s0 = 0, e0 = 0;
      bool valid = false;
      if (i < n) { const uint2 v = seg[i]; s0 = v.x; e0 = v.y; valid = s0 != e0; }
      const int32_t m = wave_incl_max_i32(valid ? (int32_t)e0 : INT32_MIN, lane);
      const uint64_t vb = __ballot(valid);
      if (lane == 63) { redi[wave] = m; }
      if (lane == 0) red[wave] = vb != 0 ? 1u : 0u;
      __syncthreads();
      int32_t before = carry;                                       // running max of everything before this wave
      bool any_before = any;
      for (int k = 0; k < wave; ++k) { const int32_t x = redi[k]; before = x > before ? x : before; any_before = any_before || red[k] != 0; }
      int32_t total = carry;
      bool any_now = any;
      for (int k = 0; k < kMergeWaves; ++k) { const int32_t x = redi[k]; total = x > total ? x : total; any_now = any_now || red[k] != 0; }
      const int32_t incl = m > before ? m : before;
      const int32_t excl = __builtin_amdgcn_update_dpp(before, incl, 0x138, 0xf, 0xf, false);   // wave_shr:1, lane 0 keeps `before`
      const bool prev_valid = any_before || (vb & lanemask_lt(lane)) != 0;
      const bool head = valid && (!prev_valid || (int32_t)s0 > excl);
      const uint64_t hb = __ballot(head);
      __syncthreads();                                              // red / redi are reused for the head counts
      if (lane == 0) red[wave] = (uint32_t)__popcll(hb);
      __syncthreads();
      int pos = count + __popcll(hb & lanemask_lt(lane));
      int heads = 0;
      for (int k = 0; k < kMergeWaves; ++k) { if (k < wave) pos += (int)red[k]; heads += (int)red[k]; }
      if (head) {
        out[pos].x = s0;
        if (pos > 0) out[pos - 1].y = (uint32_t)excl;
      }
      count += heads;
      carry = total;
      any = any_now;
      __syncthreads();
    }
    if (count > 0 && tid == 0) out[count - 1].y = (uint32_t)carry;
    __syncthreads();                                                // the merged list in the slab, visible to the block
    GAT_MPHASE(5)

    // ---- coverage of the merged list inside the workspace, and its total length
    if (nws <= kWsTreeMin) {
      for (int base = 0; base < count; base += kB * kMergeThreads) {
        uint2 v[kB];
  #pragma unroll
        for (int q = 0; q < kB; ++q) { const int i = base + q * kMergeThreads + tid; v[q] = i < count ? out[i] : make_uint2(0u, 0u); }
  #pragma unroll
        for (int q = 0; q < kB; ++q) {
          tot += v[q].y - v[q].x;
          for (int k = 0; k < nws; ++k) {
            const uint32_t l2 = v[q].x > wsl[k] ? v[q].x : wsl[k], h2 = v[q].y < wsl[kWsTreeMin + k] ? v[q].y : wsl[kWsTreeMin + k];
            cov += h2 > l2 ? h2 - l2 : 0u;
          }
        }
      }
    } else if constexpr (TREE) {
      const uint32_t* __restrict__ pg = A.ws_tree + Up->pgrid_off;          // (round 6: the position grid, not two tree searches)
      const uint32_t pshift = pg[0], pcells = pg[1];
      for (int i = tid; i < count; i += kMergeThreads) {
        const uint2 v = out[i];
        tot += v.y - v.x;
        cov += ws_overlap_pgrid(ws, nws, pg + kGridHeader, pshift, pcells, v.x, v.y);
      }
    }
  }
  {
    // (both sums through one pair of barriers)
    const uint32_t wc = wave_total_u32(cov), wt = wave_total_u32(tot);
    __syncthreads();
    if (lane == 0) { red[wave] = wc; red2[wave] = wt; }
    __syncthreads();
    cov = 0; tot = 0;
    for (int k = 0; k < kMergeWaves; ++k) { cov += red[k]; tot += red2[k]; }
  }
  GAT_MPHASE(6)
  if (A.cum != nullptr) {
    // split path: the running lengths k_tail's position draw searches (block-wide inclusive scan, 256 elements a round)
    uint32_t* __restrict__ cum = A.cum + (((int64_t)sidx * A.slab_stride + Up->slab_off) >> 3);
    uint32_t run = 0;
    for (int base = 0; base < count; base += kMergeThreads) {
      const int i = base + tid;
      uint32_t len = 0;
      if (i < count) { const uint2 v = out[i]; len = v.y - v.x; }
      const uint32_t incl = wave_incl_sum_u32(len, lane);
      __syncthreads();
      if (lane == 63) red[wave] = incl;
      __syncthreads();
      uint32_t before = run, all = 0;
      for (int k = 0; k < kMergeWaves; ++k) { if (k < wave) before += red[k]; all += red[k]; }
      if (i < count && ((i & 7) == 7 || i == count - 1)) cum[i >> 3] = before + incl;     // (per block of eight: k_tail)
      run += all;
    }
  }
  if (tid == 0) A.st2[sa] = make_int4(count, (int)cov, (int)tot, 1);
  GAT_MPHASE(7)
#ifdef GAT_DIAG_CONS
  if (tid == 0 && A.diag != nullptr) for (int k = 0; k < 8; ++k) A.diag[((int64_t)sidx * A.n_units + Up->pad) * 8 + k] = dg[k];
#undef GAT_MSTAMP
#endif
#undef GAT_MPHASE
}

// ------------------------------------------------------------------------------------------
// k_sampler: one wave per (sample, unit).  Stand-alone it runs the whole of
// SamplerAnnotator.sample; behind k_place it resumes at the first consolidation with the
// placed segments, `remaining`, the pending length and the position in the stream handed over.
// TREE: some unit's workspace is beyond the register loop (> kWsTreeMin segments) and is searched through its trees.
// HUGE: some unit's list does not fit LDS (about 15 000 segments): the list then lives where k_place left it, in the
// unit's slab region in global memory, and the same code works on it there -- slow (every pass is a round trip to
// memory, the sort is the in-place network) but without a size limit.
// WPE: waves per SIMD the register budget is set for: 4 (LDS allows no more for lists of hundreds of segments), or 5 for
// problems whose lists are so short that registers, not LDS, decide how many waves a CU holds.
template <int KIND, bool BIG, bool TREE, bool HUGE>
__device__ __forceinline__ void sampler_unit(const SamplerArgs& A, const int sidx, const int a, uint32_t* lds, const int lane,
                                             WaveRng* serial = nullptr) {
  uint32_t* mt = lds;
  const UnitDev* __restrict__ Up = A.units_o + a;
  const int u = Up->pad;
  const int nws = Up->n_ws;
  const uint32_t hist_total = Up->hist_total;
  const uint32_t bucket = Up->bucket;
  const uint32_t ws_total = Up->ws_total;
  const int32_t ltotal = Up->ltotal;
  const int cap = Up->slab_cap;
  const uint2* __restrict__ ws = A.ws + Up->ws_off;
  const uint32_t* __restrict__ ws_cdf = A.ws_cdf + Up->ws_off;
  const uint32_t* __restrict__ rank_len = A.rank_len + Up->rank_off;
  constexpr int kWsRegMax = 64, kWsLoopMax = 32;
  static_assert(kWsLoopMax == kWsTreeMin, "every workspace beyond the register loop has its search trees");
  const bool ws_in_regs = nws <= kWsRegMax;
  const uint32_t* __restrict__ tree_start = A.ws_tree + (Up->tree_start_off >= 0 ? Up->tree_start_off : 0);
  const uint32_t* __restrict__ tree_cdf = A.ws_tree + (Up->tree_cdf_off >= 0 ? Up->tree_cdf_off : 0);
  const WsTreeGeom G = ws_tree_geom(nws);
  auto ws_bisect = [&](uint32_t p) -> int {        // leftmost k with (int)(cdf[k] - p) >= 0 (utils/gat_utils.c:36)
    if constexpr (TREE) {
      if (nws > kWsTreeMin) {
        const uint32_t t[1] = {p};
        int k[1];
        ws_tree_count<true, 1>(tree_cdf, G, t, k);
        return k[0];
      }
    }
    return bisect_u32(ws_cdf, nws, p);
  };
  // (round 6: overlaps with a long workspace through its position grid -- a cell and the one or two pieces a segment can
  //  reach -- where the tree over the starts was searched twice per segment)
  const uint32_t* __restrict__ pgrid = A.ws_tree + (Up->pgrid_off >= 0 ? Up->pgrid_off : 0);
  const uint32_t pshift = TREE && Up->pgrid_off >= 0 ? pgrid[0] : 0u, pcells = TREE && Up->pgrid_off >= 0 ? pgrid[1] : 0u;
  (void)tree_start;
  auto ws_overlap1 = [&](uint32_t s, uint32_t e) -> uint32_t {   // one segment, a workspace beyond the register loop
    if constexpr (TREE) return ws_overlap_pgrid(ws, nws, pgrid + kGridHeader, pshift, pcells, s, e);
    else return 0u;                                              // (no such unit in this instantiation)
  };
  const WsRegs W = ws_load(ws, ws_cdf, nws < kWsRegMax ? nws : kWsRegMax, lane);
  uint2* out = A.slab + (int64_t)sidx * A.slab_stride + Up->slab_off;
  uint2* fin = (!HUGE && A.slab_final != nullptr) ? A.slab_final + (int64_t)sidx * A.slab_stride + Up->slab_off : out;
  uint2* seg = HUGE ? out : reinterpret_cast<uint2*>(lds + kMtLdsWords);
  const int64_t so = (int64_t)sidx * A.n_units + u;     // unit_n: [sample][unit]
  const int64_t sw = GAT_REC(A, sidx, u);                // ws_stat: [unit][sample]

  // per-unit stream: numpy.random.seed((seed + sample*n_units + unit) mod 2^32)
  const uint64_t sample_id = (uint64_t)(A.sample_begin + sidx);
  const uint32_t seed = (uint32_t)((uint64_t)A.seed + sample_id * (uint64_t)A.n_units + (uint64_t)u);
  if (A.skip != nullptr && A.skip[GAT_REC(A, sidx, a) * A.skip_stride] != 0) return;
  const bool have_pre = A.st != nullptr;
  const int4 pre = have_pre ? A.st[GAT_REC(A, sidx, a)] : make_int4(0, 0, -1, 0);
  const int32_t pre_len = pre.z;

  int nout = 0, status = 0, nuns = 0;
  uint32_t placed = 0, ndraws = 0, full_units = 0;
  bool switched = false;       // the unit was resumed and its rows ran out: the stream went on from the in-LDS generator
#ifdef GAT_DIAG
  unsigned long long dg[8] = {0, 0, 0, 0, 0, 0, 0, 0}, dg_t;
  GAT_STAMP(dg_t);
#endif
  if constexpr (KIND == 1) {
    // SamplerSegments.sample (gat/Engine.pyx:695-737): len(segments) placements, no consolidation.  Normally
    // k_place has done all of it (st_length == -2); otherwise the unit is run here from its seed.
    if (pre_len == -2) {
      nout = pre.x;
      placed = (uint32_t)nout;
      ndraws = (uint32_t)pre.w;
    } else {
      WaveRng rng;
      rng.mt = mt;
      if (serial != nullptr) { rng = *serial; rng.ndraws = 0; }     // (k_serial: the run's one stream goes on)
      else rng_seed(rng, seed, lane);
      rng.pre = nullptr; rng.pre_j = 0; rng.pre_rows = 0; rng.pre_base = 0;
      full_units = 1;
      const int target = Up->n_target;
      for (int x = 0; x < target; ++x) {
        uint32_t r = 1;
        if (hist_total > 1) r = 1u + rng_range(rng, hist_total - 2u, lane);
        uint32_t len_u = rank_len[r] * bucket;
        if (bucket > 1) len_u += rng_range(rng, bucket - 1u, lane);
        const int32_t length = (int32_t)len_u;
        const uint32_t p = rng_range(rng, ws_total - 1u, lane);
        const int k = ws_bisect(p);
        const uint2 chosen = ws[k];
        int32_t sampling_start = (int32_t)chosen.x - length + 1;
        if (k > 0) { const int32_t pe = (int32_t)ws[k - 1].y; sampling_start = pe > sampling_start ? pe : sampling_start; }
        const uint32_t range = chosen.y - 1u - (uint32_t)sampling_start;
        const int32_t q = sampling_start + (int32_t)rng_range(rng, range, lane);
        if (nout >= cap) { status |= kStatusOverflow; break; }
        if (lane == 0) out[nout] = make_uint2((uint32_t)(q > 0 ? q : 0), (uint32_t)(q + length));
        nout++;
      }
      placed = (uint32_t)nout;
      ndraws = rng.ndraws;
      if (serial != nullptr) *serial = rng;
    }
    if (lane == 0) {
      A.unit_n[so] = status ? 0 : nout;
      if (status) atomicOr(A.flags, status);
      *reinterpret_cast<uint4*>(A.ws_stat + sw * 4) = make_uint4(placed, ndraws, 0u, full_units);
    }
    return;
  }
  // (pre_len: > 0 the pending length at the first consolidation; -3: k_place ran out of rows while placing -- the loop goes
  //  on behind its last placement; -1: the unit is run in full from its seed)
  for (int attempt = ((pre_len >= 0 || pre_len == -3) ? 0 : 1); attempt < 2; ++attempt) {
    const bool resume = attempt == 0;
    WaveRng rng;
    rng.mt = mt;
    int nU = 0, nS = 0;   // seg[0..nU): unintersected (merged, sorted); seg[nU..nU+nS): sampled since
    int32_t remaining = ltotal, true_remaining = ltotal;
    int32_t pending = -1;
    bool dirty = false;          // unintersected holds trim placeholders not yet merged away
    bool cov_valid = false;      // cov_known = workspace coverage of unintersected as it stands
    uint32_t cov_known = 0, total_known = 0;
    nuns = 0; status = 0; placed = 0;
    if (resume) {
      nS = pre.x;
      int4 pre2 = make_int4(0, 0, 0, 0);
      if (!HUGE && A.st2 != nullptr && a < A.n_long) pre2 = A.st2[GAT_REC(A, sidx, a)];
      const int32_t* __restrict__ R = nullptr;
      pre2.w &= 1;                                             // (bit 1: k_consolidate's "a segment reaches out of the unit's workspace")
      if (BIG && !HUGE && A.tb != nullptr && pre2.w == 1) {
        R = A.tb + GAT_REC(A, sidx, a) * A.skip_stride;
        if (R[kPatchState] == 1) return;                       // k_resume_big has finished the unit
        if (R[kPatchState] == 3) continue;                     // k_tail_big ran out of rows: the unit is redone from its seed
        if (R[kPatchState] != 2) R = nullptr;
      }
      if (pre2.w == 1) {
        // k_merge_big has done the first consolidation: the slab holds the merged list; its coverage and total
        // length come with it, and the pending length below makes the loop take them up at once
        nU = pre2.x;
        nS = 0;
        cov_known = (uint32_t)pre2.y;
        total_known = (uint32_t)pre2.z;
        cov_valid = true;
        if (HUGE) { for (int i = lane; i < nU; i += kWave) seg[i] = out[i]; }
        else {
          // (eight rounds of loads in flight: a list of thousands, one wave, nothing else hides the latency)
          for (int base = 0; base < nU; base += 8 * kWave) {
            uint2 v[8];
#pragma unroll
            for (int r = 0; r < 8; ++r) { const int i = base + r * kWave + lane; v[r] = i < nU ? out[i] : make_uint2(0u, 0u); }
#pragma unroll
            for (int r = 0; r < 8; ++r) { const int i = base + r * kWave + lane; if (i < nU) seg[i] = v[r]; }
          }
        }
        if (R != nullptr) {
          // k_tail_big has run the placement rounds behind it: the merged list (unions applied in place), nE new segments
          // that touch nothing (logged from the region's end backwards), and nS it placed but did not consolidate
          const int nE = R[kPatchNExtra], nSp = R[kTbNSampled], at = R[kTbSampledAt];
          for (int i = lane; i < nE; i += kWave) seg[nU + i] = out[cap - 1 - i];
          for (int i = lane; i < nSp; i += kWave) seg[nU + nE + i] = out[at + i];
          wave_sync<HUGE>();
          for (int done = 0; done < nE; done += kWave) {
            const int chunk = nE - done < kWave ? nE - done : kWave;
            wave_insert_sorted<HUGE>(seg, nU + done, chunk, lane);         // (the next chunk stands right behind the grown list)
          }
          nU += nE;
          nS = nSp;
        }
      } else
      // (a list the counting sort will take straight from the slab need not be copied first -- unless the unit goes on PLACING,
      //  pre_len -3: what it places joins the list in LDS)
      if (!HUGE && (pre_len == -3 || (!(BIG && nS > 1024 && A.big_buckets > 0) && !(nS > 512 && nS <= 1024))))
        for (int i = lane; i < nS; i += kWave) seg[i] = out[i];
      remaining = pre.y;
      pending = pre_len >= 0 ? pre_len : -1;
      rng.use_pre = true; rng.exhausted = false; rng.ndraws = (uint32_t)pre.w; rng.pos = 0; rng.rbuf = 0;
      rng.seed = seed; rng.can_switch = true;        // (where the rows run out the stream goes on from the in-LDS generator)
      rng.pre_rows = (uint32_t)A.rng_rows[a];
      rng.pre_j = rng.ndraws;
      rng.pre_base = rng.pre_j - (uint32_t)kWave;     // forces the first prefetch
      rng.pre = A.rng_out + A.rng_off[a] + (int64_t)(sidx >> 6) * rng.pre_rows * kWave + (sidx & 63);
      placed = (uint32_t)pre.x;
      if (R != nullptr) {
        // ... and the loop goes on where k_tail_big left it: at a consolidation whose bookkeeping (:601-605) is still to do
        remaining = R[kTbRemaining];
        pending = R[kTbPending];
        true_remaining = R[kTbTrueRemaining];
        nuns = R[kPatchNuns];
        placed = (uint32_t)R[kPatchPlaced];
        rng.ndraws = (uint32_t)R[kPatchNdraws];
        rng.pre_j = rng.ndraws;
        rng.pre_base = rng.pre_j - (uint32_t)kWave;
      }
      wave_sync<HUGE>();
      GAT_PHASE(0)                                   // prologue: unit record, workspace, hand-off record, list into LDS
    } else {
      if (serial != nullptr) { rng = *serial; rng.mt = mt; rng.ndraws = 0; }      // (k_serial: the run's one stream goes on)
      else rng_seed(rng, seed, lane);
      rng.pre = nullptr; rng.pre_j = 0; rng.pre_rows = 0; rng.pre_base = 0;
      full_units = 1;
    }

    while (true_remaining > 0 && nuns < 20) {                       // gat/Engine.pyx:572
      // ---- hs.sample() (:413-435)
      int32_t length;
      if (pending >= 0) { length = pending; pending = -1; }
      else {
        uint32_t r = 1;
        if (hist_total > 1) r = 1u + rng_range(rng, hist_total - 2u, lane);
        uint32_t len_u = rank_len[r] * bucket;     // == cdf bisect (utils/gat_utils.c:36), tabulated per rank
        if (bucket > 1) len_u += rng_range(rng, bucket - 1u, lane);
        length = (int32_t)len_u;
      }
      GAT_PHASE(6)                                   // draws + placement arithmetic

      // ---- consolidate (:582-606)
      if (remaining <= length) {
        uint32_t cov;
        bool handled = false;
        if (nS == 0 && cov_valid) {
          // nothing was placed since the last consolidation, only trimmed: merge(0) would just drop the
          // placeholders (left for the final pass) and the coverage is the old one minus what the trim removed
          cov = cov_known;
          handled = true;
        } else if (nS == 1 && cov_valid && !dirty && nU > 0) {
          // one new segment and a clean merged list: if it neither overlaps nor touches its neighbours
          // (merge(0) joins at start <= previous max end) the merged list is the old one plus this segment
          const uint2 xv = seg[nU];
          const uint32_t xs = rfl(xv.x), xe = rfl(xv.y);
          int pcount = 0;                                   // #U with start <= xs
          for (int base = 0; base < nU; base += kWave) {
            const int i = base + lane;
            pcount += __popcll(__ballot(i < nU && seg[i].x <= xs));
          }
          const uint2 pv = seg[pcount > 0 ? pcount - 1 : 0], nv = seg[pcount < nU ? pcount : nU - 1];
          const bool touch_prev = pcount > 0 && (int32_t)xs <= (int32_t)rfl(pv.y);
          const bool touch_next = pcount < nU && (int32_t)rfl(nv.x) <= (int32_t)xe;
          // (an EMPTY neighbour -- what a trim emptied, or the placeholder one of k_tail_big's bridges left at its start -- says
          //  nothing about what stands there: the full merge below)
          const bool hole = (pcount > 0 && rfl(pv.x) == rfl(pv.y)) || (pcount < nU && rfl(nv.x) == rfl(nv.y));
          if (xs != xe && !touch_prev && !touch_next && !hole) {
            wave_insert_sorted<HUGE>(seg, nU, 1, lane);
            nU += 1;
            nS = 0;
            cov = cov_known + (nws <= kWsLoopMax ? ws_overlap_regs(W, xs, xe) : ws_overlap1(xs, xe));
            total_known += xe - xs;
            handled = true;
          }
        }
        GAT_PHASE(4)                                 // consolidation fast paths (nothing new / one new segment)
        if (!handled) {
          if (BIG && dirty && nU > 0 && nS <= kWave && nU + nS > 1024) {      // (up to 1024 the bucket sort is cheaper)
            // the trim left the merged list sorted but for its placeholders, which merge(0) skips wherever they
            // are: one compaction pass (instead of sorting a long list again with the network) and it is clean and sorted,
            // ready for the few new segments to be inserted
            uint2 moved = make_uint2(0u, 0u);
            if (lane < nS) moved = seg[nU + lane];
            const int kept = wave_merge0<HUGE>(seg, nU, lane);
            if (lane < nS) seg[kept + lane] = moved;
            wave_sync<HUGE>();
            nU = kept;
            dirty = false;
          }
          const int n = nU + nS;
          if (BIG && !HUGE && resume && pre_len >= 0 && nU == 0 && !dirty && n > 1024 && A.big_buckets > 0) {
            // long list straight from the slab k_place wrote: counting sort into LDS (it was not copied at resume)
            int nb = 1024;
            while (nb < n && nb < A.big_buckets) nb <<= 1;
            uint32_t* big = reinterpret_cast<uint32_t*>(seg + A.lds_cap);
            if (!wave_sort_bucket_global(seg, out, n, big, nb, lane)) {
              for (int i = lane; i < n; i += kWave) seg[i] = out[i];
              wave_sort_by_start<HUGE>(seg, n, lane);
            }
          } else if (!HUGE && resume && pre_len >= 0 && rng.use_pre && nU == 0 && !dirty && n > 512 && n <= 1024 && n == pre.x) {
            // 513..1024 segments, still where k_place wrote them: the same counting sort with 512 buckets (the bucket sort
            // that holds the whole list in registers needs 16 elements per lane here -- over a hundred registers, which cost
            // every wave of the kernel spills: 128 VGPRs + 76 bytes of scratch against 106 without)
            if (!wave_sort_bucket_global(seg, out, n, mt, 512, lane)) {
              for (int i = lane; i < n; i += kWave) seg[i] = out[i];
              wave_sort_auto<HUGE>(seg, n, lane);
            }
          } else if (BIG && nU > 0 && !dirty && n > 1024 && nS <= 1024) {
            // a long clean list and some new segments: insert them 64 at a time (a pass over the list each) rather than
            // sort everything with the network (thousands of passes)
            for (int done = 0; done < nS; done += kWave)
              wave_insert_sorted<HUGE>(seg, nU + done, nS - done < kWave ? nS - done : kWave, lane);
          } else
          if (nU == 0 || nS > kWave || dirty) wave_sort_fast<8, HUGE>(seg, n, (resume && rng.use_pre) ? mt : nullptr, lane);   // SegmentList.sort of everything
                                        // (the MT19937 words are idle scratch while the stream comes from k_rng)
          else if (nS > 0) wave_insert_sorted<HUGE>(seg, nU, nS, lane);         // same order, few new segments
          GAT_PHASE(1)                               // sort / insert
          nU = wave_merge0<HUGE>(seg, n, lane);
          GAT_PHASE(2)                               // merge(0)
          nS = 0;
          dirty = false;
          cov = 0;
          uint32_t tot = 0;
          if (nws <= kWsLoopMax) {
            for (int i = lane; i < nU; i += kWave) { const uint2 v = seg[i]; cov += ws_overlap_regs(W, v.x, v.y); tot += v.y - v.x; }
          } else if constexpr (TREE) {
            constexpr int R = 2;                              // segments per lane whose workspace searches run interleaved
            for (int base = 0; base < nU; base += R * kWave) {
              uint2 v[R];
              uint32_t ov[R];
#pragma unroll
              for (int r = 0; r < R; ++r) {
                const int i = base + r * kWave + lane;
                v[r] = i < nU ? seg[i] : make_uint2(0u, 0u);
                tot += v[r].y - v[r].x;
              }
#pragma unroll
              for (int r = 0; r < R; ++r) ov[r] = ws_overlap_pgrid(ws, nws, pgrid + kGridHeader, pshift, pcells, v[r].x, v[r].y);
#pragma unroll
              for (int r = 0; r < R; ++r) cov += ov[r];
            }
          }
          cov = wave_total_u32(cov);
          total_known = wave_total_u32(tot);             // sum() of the merged list, for the trim's position draw
          GAT_PHASE(3)                               // workspace coverage of the merged list
        }
        cov_known = cov;
        cov_valid = true;
        remaining = ltotal - (int32_t)cov;
        if (true_remaining == remaining) nuns++; else true_remaining = remaining;
        // the reference still draws a position here (:628) before its loop test fails; the draws and
        // the segment are discarded and the unit's stream ends, so nothing observable depends on them
        if (!(true_remaining != 0 && nuns < 20)) {
          if (serial != nullptr) {
            // (the run's one stream goes on behind this unit: the draws of that discarded position are consumed -- on both
            //  ways out, nothing left to place or twenty rounds without progress; a negative true_remaining cannot stand
            //  here: it only equals `remaining` right behind a trim, which sets it to 1)
            const uint32_t p_ = rng_range(rng, ws_total - 1u, lane);
            int k_;
            uint2 ch_;
            int32_t pe_ = 0;
            if (ws_in_regs) {
              const uint64_t b_ = __ballot(lane < nws && (int32_t)(W.cdf - p_) >= 0);
              k_ = (int)__builtin_ctzll(b_);
              ch_.x = (uint32_t)__builtin_amdgcn_readlane((int)W.start, k_);
              ch_.y = (uint32_t)__builtin_amdgcn_readlane((int)W.end, k_);
              if (k_ > 0) pe_ = __builtin_amdgcn_readlane((int)W.end, k_ - 1);
            } else {
              k_ = ws_bisect(p_);
              ch_ = ws[k_];
              if (k_ > 0) pe_ = (int32_t)ws[k_ - 1].y;
            }
            int32_t ss_ = (int32_t)ch_.x - length + 1;
            if (k_ > 0) ss_ = pe_ > ss_ ? pe_ : ss_;
            (void)rng_range(rng, ch_.y - 1u - (uint32_t)ss_, lane);
          }
          break;
        }
      }

      // ---- overshoot: trim (:608-626)
      if (true_remaining < 0) {
        // SegmentListSampler(unintersected).sample(1): position draw over the cumulated lengths
        const uint32_t total = total_known;           // kept by the consolidation / earlier trims
        const uint32_t p = rng_range(rng, total - 1u, lane);
        int k = 0;
        uint32_t run = 0;
        for (int base = 0; base < nU; base += kWave) {
          const int i = base + lane;
          uint32_t len = 0;
          if (i < nU) { const uint2 v = seg[i]; len = v.y - v.x; }
          const uint32_t incl = run + wave_incl_sum_u32(len, lane);
          // cdf[i] = incl-1; leftmost i with (int)(cdf[i]-p) >= 0
          const bool ge = (i < nU) && ((int32_t)(incl - 1u - p) >= 0);
          const uint64_t b = __ballot(ge);
          if (b != 0) { k = base + (int)__builtin_ctzll(b); break; }
          run = (uint32_t)__builtin_amdgcn_readlane((int)incl, kWave - 1);
        }
        // unintersected is merged(0): previous.end < chosen.start, so sampling_start == chosen.start
        const uint2 chosen = seg[k];
        const uint32_t cs = rfl(chosen.x), ce = rfl(chosen.y);
        (void)rng_range(rng, ce - 1u - cs, lane);                       // position inside the segment: only its index k matters
        const uint32_t forward = rng_range(rng, 1u, lane);            // numpy.random.randint(0, 2)
        int32_t s = -true_remaining;
        if (rng.exhausted) break;
        if (!((uint64_t)total > (uint64_t)(uint32_t)s)) { status |= kStatusTrimAssert; break; }
        // trim_ends(pos, s, forward) (gat/SegmentList.pyx:545-597); _getInsertionPoint(pos,pos+1) == k
        wave_sync<HUGE>();
        uint32_t removed = 0;                      // workspace bases taken away by the trim (lane 0)
        if (lane == 0) {
          int idx = k;
          while (s > 0) {
            const uint2 v = seg[idx];
            const int32_t l = (int32_t)v.y - (int32_t)v.x;
            uint32_t ra, rb;                       // removed range
            if (l < s) { seg[idx] = make_uint2(0u, 0u); s -= l; ra = v.x; rb = v.y; }
            else {
              if (forward) { seg[idx] = make_uint2(v.x + (uint32_t)s, v.y); ra = v.x; rb = v.x + (uint32_t)s; }
              else { seg[idx] = make_uint2(v.x, (uint32_t)((int32_t)v.y - s)); ra = (uint32_t)((int32_t)v.y - s); rb = v.y; }
              s = 0;
            }
            if (rb > ra) removed += nws <= kWsLoopMax ? ws_overlap_regs(W, ra, rb) : ws_overlap1(ra, rb);
            if (forward) { idx++; if (idx == nU) idx = 0; }
            else { idx--; if (idx < 0) idx = nU - 1; }
          }
        }
        cov_known -= rfl(removed);
        total_known -= (uint32_t)(-true_remaining);      // trim_ends removes exactly that many bases
        wave_sync<HUGE>();
        dirty = true;
        true_remaining = 1;
        GAT_PHASE(5)                                 // overshoot trim
        continue;
      }

      // ---- sls.sample(length) (:279-343)
      const uint32_t p = rng_range(rng, ws_total - 1u, lane);
      int k;
      uint2 chosen;
      int32_t prev_end = 0;
      if (ws_in_regs) {
        // leftmost i with (int)(cdf[i]-p) >= 0 (utils/gat_utils.c:36 + cmpPosition), one compare per lane
        const uint64_t b = __ballot(lane < nws && (int32_t)(W.cdf - p) >= 0);
        k = (int)__builtin_ctzll(b);
        chosen.x = (uint32_t)__builtin_amdgcn_readlane((int)W.start, k);
        chosen.y = (uint32_t)__builtin_amdgcn_readlane((int)W.end, k);
        if (k > 0) prev_end = __builtin_amdgcn_readlane((int)W.end, k - 1);
      } else {
        k = ws_bisect(p);
        chosen = ws[k];
        if (k > 0) prev_end = (int32_t)ws[k - 1].y;
      }
      int32_t sampling_start = (int32_t)chosen.x - length + 1;
      if (k > 0) sampling_start = prev_end > sampling_start ? prev_end : sampling_start;
      const uint32_t range = chosen.y - 1u - (uint32_t)sampling_start;
      const int32_t q = sampling_start + (int32_t)rng_range(rng, range, lane);
      if (rng.exhausted) break;
      const uint32_t start = (uint32_t)(q > 0 ? q : 0);
      const uint32_t end = (uint32_t)(q + length);
      const int32_t omin = (int32_t)chosen.y < (int32_t)end ? (int32_t)chosen.y : (int32_t)end;
      const int32_t omax = (int32_t)chosen.x > (int32_t)start ? (int32_t)chosen.x : (int32_t)start;
      const int32_t overlap = omin - omax > 0 ? omin - omax : 0;
      if (true_remaining > 0) {
        if (nU + nS >= cap) { status |= kStatusOverflow; break; }
        if (lane == 0) seg[nU + nS] = make_uint2(start, end);
        nS++;
        placed++;
        remaining -= overlap;
      }
      GAT_PHASE(6)
    }
    ndraws = rng.ndraws;
    switched = resume && !rng.use_pre;
    if (serial != nullptr) *serial = rng;
    GAT_PHASE(6)
    if (rng.use_pre && rng.exhausted) continue;       // rows ran out: redo this unit from its seed

    // ---- result = unintersected.merge(0).filter(workspace) (:639-646); pending sampled are dropped
    nout = 0;
    if (status == 0) {
      // (after a trim the list holds [0, 0) placeholders; nothing touches that did not before -- a trim only shortens --
      //  and the filter below drops what overlaps nothing, placeholders included: no merge(0) pass of its own)
      uint32_t total = 0;
      if (nws <= kWsLoopMax) {
        // (four rounds at a time: their LDS reads and workspace tests do not depend on one another, only the output
        //  positions do -- a long list is worked on by one wave with little else on its SIMD to hide a round's latency)
        for (int base = 0; base < nU; base += 4 * kWave) {
          uint2 v[4];
          bool keep[4];
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int i = base + q * kWave + lane;
            v[q] = i < nU ? seg[i] : make_uint2(0u, 0u);
          }
#pragma unroll
          for (int q = 0; q < 4; ++q) keep[q] = base + q * kWave + lane < nU && ws_overlap_regs(W, v[q].x, v[q].y) > 0;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const uint64_t b = __ballot(keep[q]);
            if (keep[q]) { fin[nout + __popcll(b & lanemask_lt(lane))] = v[q]; total += v[q].y - v[q].x; }
            nout += __popcll(b);
          }
        }
      } else if constexpr (TREE) {
        constexpr int R = 2;
        for (int base = 0; base < nU; base += R * kWave) {
          uint2 v[R];
          uint32_t ov[R];
#pragma unroll
          for (int r = 0; r < R; ++r) {
            const int i = base + r * kWave + lane;
            v[r] = i < nU ? seg[i] : make_uint2(0u, 0u);
          }
#pragma unroll
          for (int r = 0; r < R; ++r) ov[r] = ws_overlap_pgrid(ws, nws, pgrid + kGridHeader, pshift, pcells, v[r].x, v[r].y);
#pragma unroll
          for (int r = 0; r < R; ++r) {
            const bool keep = ov[r] > 0;                     // (0,0) fillers overlap nothing
            const uint64_t b = __ballot(keep);
            if (keep) { fin[nout + __popcll(b & lanemask_lt(lane))] = v[r]; total += v[r].y - v[r].x; }
            nout += __popcll(b);
          }
        }
      }
      total = wave_total_u32(total);
      if (!(total > 0)) status |= kStatusAssert;
    }
    GAT_PHASE(7)                                     // final merge (after a trim), workspace filter, list to the slab
    break;
  }
#if defined(GAT_DIAG) && !defined(GAT_DIAG_CONS)
  if (lane == 0 && A.diag != nullptr)
    for (int k = 0; k < 8; ++k) A.diag[so * 8 + k] = dg[k];
#endif
  if (lane == 0) {
    A.unit_n[so] = nout;
    if (status) atomicOr(A.flags, status);
    *reinterpret_cast<uint4*>(A.ws_stat + sw * 4) = make_uint4(placed, ndraws, (uint32_t)nuns, full_units | (switched ? 0x10000u : 0u));
  }
}

template <int KIND, bool BIG, bool TREE, bool HUGE, int WPE = 4>
__global__ __launch_bounds__(64, WPE) void k_sampler(SamplerArgs A) {
  extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
  const int lane = threadIdx.x;
  if (A.todo_count != nullptr) {
    // behind the split path: the few units k_tail left alone, from a queue (a launch over all units, nearly all of which
    // leave at once, took longer than everything else the split path does)
    const uint32_t count = *A.todo_count;
    for (uint32_t w = blockIdx.x; w < count; w += gridDim.x) {
      const uint32_t e = A.todo[w];
      const int qa = (int)(e % (uint32_t)A.n_active);
      if (A.a_end > 0 && (qa < A.a_base || qa >= A.a_end)) continue;       // (long lists: a launch per size class, each takes its own)
      sampler_unit<KIND, BIG, TREE, HUGE>(A, (int)(e / (uint32_t)A.n_active), qa, lds, lane);
      wave_sync<HUGE>();
    }
    return;
  }
  const int a = A.a_base + (int)(blockIdx.y + blockIdx.z * gridDim.y);
  if (a >= A.n_active || (A.a_end > 0 && a >= A.a_end)) return;
  sampler_unit<KIND, BIG, TREE, HUGE>(A, (int)blockIdx.x, a, lds, lane);
}

// sums the per-work-unit statistics: one atomic per block instead of one per work unit
// k_serial: the reference's OWN random stream -- numpy.random.seed(seed) once (scripts/gat-run.py:267-271), then every
// (sample, unit) in the order of gat/__init__.py:531-541 drawing from that one MT19937 -- so that `gat-run.py
// --random-seed=N` of an unpatched reference can be reproduced table for table.  One stream is one chain: ONE wave runs
// the whole batch, unit after unit, with the sampler code of k_sampler in its stand-alone form (slow by construction --
// 0.4 ms per unit of 400 segments, 106 samples/s on config 2 -- and still five times the reference's engine).  The state (624 words +
// position) comes from and goes back to global memory, so batches, segment tracks and calls continue each other.
template <int KIND, bool BIG, bool TREE, bool HUGE>
__global__ __launch_bounds__(64) void k_serial(SamplerArgs A) {
  extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
  const int lane = threadIdx.x;
  uint32_t* mt = lds;
  for (int i = lane; i < kMtN; i += kWave) mt[i] = A.serial_state[i];
  WaveRng rng;
  rng.mt = mt;
  rng.pos = (int)A.serial_state[kMtN];
  rng.ndraws = 0;
  rng.use_pre = false; rng.exhausted = false;
  rng.pre = nullptr; rng.pre_j = 0; rng.pre_rows = 0; rng.pre_base = 0;
  wave_sync();
  rng.rbuf = 0;
  if (rng.pos < kMtN && (rng.pos & (kWave - 1)) != 0) {           // inside a block of 64 outputs: its tempered words
    const int i = (rng.pos & ~(kWave - 1)) + lane;
    rng.rbuf = mt_temper(mt[i < kMtN ? i : kMtN - 1]);
  }
  for (int s = 0; s < A.batch; ++s) {
    for (int u = 0; u < A.n_units; ++u) {
      const int a = A.unit_pos[u];
      if (a < 0) continue;                                          // (gat/__init__.py:536-538: no segments or no workspace)
      sampler_unit<KIND, BIG, TREE, HUGE>(A, s, a, lds, lane, &rng);
      wave_sync<HUGE>();
    }
  }
  wave_sync();
  for (int i = lane; i < kMtN; i += kWave) A.serial_state[i] = mt[i];
  if (lane == 0) A.serial_state[kMtN] = (uint32_t)rng.pos;
}

// (records [unit][rec_stride]: the nb samples of this batch are the first nb of every unit's row)
__global__ __launch_bounds__(256) void k_reduce_stats(const uint32_t* __restrict__ ws_stat, int64_t nb, int64_t n_units, int64_t rec_stride,
                                                      unsigned long long* __restrict__ stat,
                                                      const int32_t* __restrict__ skip, int skip_stride) {
  unsigned long long a0 = 0, a1 = 0, a2 = 0, a4 = 0, a3 = 0, a5 = 0;
  const int64_t n = nb * n_units;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t u = i / nb, r = u * rec_stride + (i - u * nb);
    const uint4 v = *reinterpret_cast<const uint4*>(ws_stat + r * 4);
    a0 += v.x; a1 += v.y; a2 += v.z; a4 += v.w & 0xffffu; a5 += v.w >> 16;   // (.w: run in full | resumed with a stream moved up << 16)
    // work units finished on the split path (k_tail's records are indexed by launch position: n_units rows as well)
    if (skip != nullptr) a3 += skip[r * skip_stride] != 0 ? 1 : 0;
  }
  __shared__ unsigned long long red[4][6];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) {
    a0 += __shfl_xor(a0, d); a1 += __shfl_xor(a1, d); a2 += __shfl_xor(a2, d); a4 += __shfl_xor(a4, d); a3 += __shfl_xor(a3, d);
    a5 += __shfl_xor(a5, d);
  }
  if (lane == 0) { red[wave][0] = a0; red[wave][1] = a1; red[wave][2] = a2; red[wave][3] = a3; red[wave][4] = a4; red[wave][5] = a5; }
  __syncthreads();
  if (threadIdx.x < 6) {
    const unsigned long long t = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
    atomicAdd(&stat[threadIdx.x], t);
  }
}

// ------------------------------------------------------------------------------------------
// fromIsochores (gat/Engine.pyx:2857-2876): new[contig].extend(unit lists) then merge(0).
constexpr int kCandSlots = 256;    // regions of the candidate buffer (CountArgs::cand), each with a counter of its own
struct ContigArgs {
  const int32_t* contig_unit_off;  // n_contigs+1: range into contig_units
  const int32_t* contig_units;     // unit ids grouped by contig, reference order
  const int4* cu_rec;              // the same order: {unit id, its place in a sample's slab, its launch position (-1: inactive), 0} --
                                   // what a lane needs of unit k in ONE load instead of three dependent ones
  const UnitDev* units;
  const int32_t* contig_slab_off;  // n_contigs: output region of the contig inside a sample slab
  int32_t n_units, n_contigs;
  const uint2* slab_in;            // sampler output
  uint2* slab_out;                 // contig-level lists
  int64_t slab_stride;
  const int32_t* unit_n;           // [batch][n_units]
  int32_t* contig_n;               // [batch][n_contigs]
  unsigned long long* stat;
  // split path without k_finalize: units k_tail finished are taken as (merged list in slab_merged, k_tail's record)
  const uint2* slab_merged;        // nullptr: every unit's final list is in slab_in
  const int32_t* unit_pos;         // unit id -> launch position (patch / st2 are indexed by it), -1: inactive
  const int4* st2;                 // .x = merged segments
  const int32_t* patch;            // TailPatch records as words (layout: gat_tail.h), patch_stride words each
  int32_t patch_stride;
  int32_t rec_stride;              // (GAT_REC: st2 / patch / ws_stat are [unit][rec_stride])
  uint32_t* ws_stat;               // per-unit statistics (k_finalize's job otherwise)
  // one launch per size class of contigs: launch position p of this launch is contig order[base + p]
  const int32_t* order;
  int32_t base, count;
  int32_t lds_cap;                 // segments the launch's LDS holds (not HUGE)
  int32_t* flags;
  // k_contig<., true> (NOSORT): the lists only concatenated, the candidates for k_units_overlap noted (CountArgs::cand)
  const uint32_t* bmap;            // per contig, bmap_off[c] words in: one bit per 2^bshift bases, a workspace boundary lies in the cell
  const int64_t* bmap_off;
  int32_t bshift;
  uint4* cand;
  uint32_t* cand_count;
  uint32_t cand_cap;
};

// HUGE: a contig's list does not fit LDS; it is gathered, sorted and merged in its output region in global memory.
// NOSORT (round 6): counts alone through the merged index, whose look-ups do not care for the order -- the units' lists are only
// concatenated (no bucket sort, no merge(0): 40 % of the kernel), and the segments of the units whose record says that one of
// theirs reaches out of the unit's workspace (st2.w bit 1, TailPatch::nuns bit 16; a unit k_sampler finished: always) are tested against the
// contig's boundary map: a segment whose cells hold a boundary is a candidate for k_units_overlap.
template <bool HUGE, bool NOSORT = false>
__global__ __launch_bounds__(64) void k_contig(ContigArgs A) {
  extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
  uint32_t* scratch = lds;                               // kSortScratchWords for the bucket sort
  const int lane = threadIdx.x;
  const int sidx = blockIdx.x;
  const int cp = (int)(blockIdx.y + blockIdx.z * gridDim.y);     // (contigs beyond one grid dimension)
  if (cp >= A.count) return;
  const int c = A.order[A.base + cp];
  uint2* out = A.slab_out + (int64_t)sidx * A.slab_stride + A.contig_slab_off[c];
  constexpr bool INPLACE = HUGE || NOSORT;                // the list is put together where it goes (NOSORT: nothing is sorted in LDS)
  uint2* seg = INPLACE ? out : reinterpret_cast<uint2*>(lds + kSortScratchWords);
  int n = 0;
  const int u0 = A.contig_unit_off[c], u1 = A.contig_unit_off[c + 1];
  // Units are taken 64 at a time: lane k looks up unit k's list (count, place in the slab) -- one round trip for all of
  // them instead of two dependent ones per unit -- and the first 64 segments of up to eight lists are in flight together.
  for (int ub = u0; ub < u1; ub += kWave) {
    const int nu = u1 - ub < kWave ? u1 - ub : kWave;
    int my_cnt = 0, my_off = 0, my_copy = 0;
    bool my_patched = false, my_flag = false;
    const int32_t* my_patch = nullptr;
    int my_u = 0;
    uint4 pw[4] = {make_uint4(0u, 0u, 0u, 0u), make_uint4(0u, 0u, 0u, 0u), make_uint4(0u, 0u, 0u, 0u), make_uint4(0u, 0u, 0u, 0u)};
    if (lane < nu) {
      // the unit's record, then everything that depends on it at once: k_tail's hand-over record (state, extras), the merged
      // list's length, the final list's length -- one round trip where state -> length -> extras were three
      const int4 rec = A.cu_rec[ub + lane];
      my_u = rec.x;
      my_off = rec.y;
      const bool have_patch = A.slab_merged != nullptr && rec.z >= 0;
      const int64_t sa = GAT_REC(A, sidx, have_patch ? rec.z : 0);
      int4 c2 = make_int4(0, 0, 0, 0);
      if (have_patch) {
        my_patch = A.patch + sa * A.patch_stride;
        const uint2* __restrict__ p2 = reinterpret_cast<const uint2*>(my_patch);       // (72-byte records: 8-byte aligned)
        // words 0..13: state, n_extra, placed, ndraws, nuns, pad, four extras
        const uint2 w0 = p2[0], w1 = p2[1], w2 = p2[2], w3 = p2[3], w4 = p2[4], w5 = p2[5], w6 = p2[6];
        pw[0] = make_uint4(w0.x, w0.y, w1.x, w1.y);
        pw[1] = make_uint4(w2.x, w2.y, w3.x, w3.y);
        pw[2] = make_uint4(w4.x, w4.y, w5.x, w5.y);
        pw[3] = make_uint4(w6.x, w6.y, 0u, 0u);
        c2 = A.st2[sa];
      }
      const int final_n = A.unit_n[(int64_t)sidx * A.n_units + my_u];
      my_patched = have_patch && (int32_t)pw[0].x == 1;
      if (my_patched) { my_copy = c2.x; my_cnt = my_copy + (int32_t)pw[0].y; }
      else my_cnt = my_copy = final_n;
      // (a segment of the unit reaches out of its workspace: st2.w bit 1 for the merged list, the record's nuns bit 16 for k_tail's extras)
      my_flag = my_cnt > 0 && (!my_patched || (c2.w & 2) != 0 || (pw[1].x >> 16) != 0u);
    }
    const int my_dst = n + (int)(wave_incl_sum_u32((uint32_t)my_cnt, lane) - (uint32_t)my_cnt);
    if (!INPLACE && n + (int)wave_total_u32((uint32_t)my_cnt) > A.lds_cap) {      // (wave-uniform; nothing of this batch is kept)
      if (lane == 0) atomicOr(A.flags, kStatusContigLds);
      return;
    }
    const uint2* __restrict__ base_final = A.slab_in + (int64_t)sidx * A.slab_stride;
    const uint2* __restrict__ base_merged = A.slab_merged != nullptr ? A.slab_merged + (int64_t)sidx * A.slab_stride : base_final;
    const uint64_t patched_mask = __ballot(my_patched);
    constexpr int kU = 8;
    for (int k0 = 0; k0 < nu; k0 += kU) {
      uint2 v[kU];
      int cnt[kU], dst[kU], off[kU];
      const uint2* src[kU];
#pragma unroll
      for (int q = 0; q < kU; ++q) {
        const int k = k0 + q < nu ? k0 + q : nu - 1;
        cnt[q] = k0 + q < nu ? __builtin_amdgcn_readlane(my_copy, k) : 0;
        dst[q] = __builtin_amdgcn_readlane(my_dst, k);
        off[q] = __builtin_amdgcn_readlane(my_off, k);
        src[q] = ((patched_mask >> k) & 1ull) ? base_merged : base_final;
        v[q] = make_uint2(0u, 0u);
        if (lane < cnt[q]) v[q] = src[q][off[q] + lane];
      }
#pragma unroll
      for (int q = 0; q < kU; ++q) {
        if (lane < cnt[q]) seg[dst[q] + lane] = v[q];
        for (int i = kWave + lane; i < cnt[q]; i += kWave) seg[dst[q] + i] = src[q][off[q] + i];   // lists beyond 64 segments
      }
    }
    if (patched_mask != 0) {
      // one lane per finished unit: its extras behind the merged list (k_tail has applied the trim to both; the order
      // inside the contig's list does not matter: it is sorted below, and merge(0) drops the emptied segments)
      wave_sync<INPLACE>();
      if (my_patched) {
        // (record words: 0 state, 1 n_extra, 2 placed, 3 ndraws, 4 nuns, 5 pad, 6.. extras)
        const int nU = my_copy, nE = (int32_t)pw[0].y;
        const uint2 ex[4] = {make_uint2(pw[1].z, pw[1].w), make_uint2(pw[2].x, pw[2].y), make_uint2(pw[2].z, pw[2].w), make_uint2(pw[3].x, pw[3].y)};
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (j < nE) seg[my_dst + nU + j] = ex[j];
        *reinterpret_cast<uint4*>(A.ws_stat + GAT_REC(A, sidx, my_u) * 4) = make_uint4(pw[0].z, pw[0].w, pw[1].x & 0xffffu, 0u);
      }
      wave_sync<INPLACE>();
    }
    if constexpr (NOSORT) {
      uint64_t fm = __ballot(my_flag);
      if (fm != 0ull) {                                              // (one unit in forty on config 3)
        wave_sync<INPLACE>();
        const uint32_t* __restrict__ BM = A.bmap + A.bmap_off[c];
        const uint32_t slot_c = ((uint32_t)sidx + (uint32_t)cp) % (uint32_t)kCandSlots;
        while (fm != 0ull) {
          const int k = __builtin_ctzll(fm);
          fm &= fm - 1ull;
          const int cnt_k = __builtin_amdgcn_readlane(my_cnt, k), dst_k = __builtin_amdgcn_readlane(my_dst, k);
          for (int j0 = 0; j0 < cnt_k; j0 += kWave) {
            const int j = j0 + lane;
            const uint2 xv = j < cnt_k ? seg[dst_k + j] : make_uint2(0u, 0u);
            bool cand = false;
            if (xv.x != xv.y) {
              // the bits from the first base's cell to the last base's (a boundary strictly inside the segment lies in one of them)
              const uint32_t c0 = xv.x >> A.bshift, c1 = (xv.y - 1u) >> A.bshift, nb = c1 - c0 + 1u;
              const uint32_t* __restrict__ wp = BM + (c0 >> 5);
              const uint64_t win = ((uint64_t)wp[0] | ((uint64_t)wp[1] << 32)) >> (c0 & 31u);
              cand = nb > 32u || (win & ((1ull << nb) - 1ull)) != 0ull;
            }
            const uint64_t m = __ballot(cand);
            if (m == 0ull) continue;
            uint32_t base = 0;
            if (lane == 0) base = atomicAdd(&A.cand_count[slot_c], (uint32_t)__popcll(m));
            base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
            if (cand) {
              const uint32_t at = base + (uint32_t)__popcll(m & lanemask_lt(lane));
              if (at < A.cand_cap) A.cand[(size_t)slot_c * A.cand_cap + at] = make_uint4((uint32_t)sidx, (uint32_t)(ub + k), xv.x, xv.y);
              else atomicOr(&A.cand_count[kCandSlots], 2u);      // (the region is full: the host makes the buffer larger and repeats the batch)
            }
          }
        }
      }
    }
    n += (int)wave_total_u32((uint32_t)my_cnt);
  }
  if constexpr (!NOSORT) {
    wave_sort_fast<16, HUGE>(seg, n, scratch, lane);
    n = wave_merge0<HUGE>(seg, n, lane);
  } else {
    (void)scratch;
    wave_sync<INPLACE>();
  }
  if (!INPLACE)
    for (int i = lane; i < n; i += kWave) out[i] = seg[i];
  if (lane == 0) {
    A.contig_n[(int64_t)sidx * A.n_contigs + c] = n;
  }
}

// ------------------------------------------------------------------------------------------
// Counters.  Annotation track t, contig c is the normalized list starts/ends[off .. off+m) with
// cumx[i] = total length of its intervals before i.  For a sample segment x:
//   overlapWithSegments contribution (gat/SegmentList.pyx:1026-1076) = F(x.end) - F(x.start),
//     F(p) = annotation bases below p  (both lists normalized => equals the sum over pairs);
//   intersectionWithSegments (:1078-1146): x is tested against the FIRST annotation interval y
//     with y.end > x.start (the merge-join's current `other`); hit iff y.start < x.end, and in
//     midpoint mode additionally y.start <= x.start + (x.end-x.start)/2 < y.end.
struct CountArgs {
  // sample lists
  const uint2* seg;           // list (s, c) = seg[s*seg_stride + c_off[c] ..)
  int64_t seg_stride;
  const int32_t* c_off;       // n_contigs
  const int32_t* n_arr;       // n of list (s, c) = n_arr[s*n_stride + n_index[c]]
  int32_t n_stride;
  const int32_t* n_index;
  // annotations: list (t, c) = start/end/cumx[a_off[t*C+c] ..), grid[g_off[t*C+c] .. +cells[c]+1)
  const uint32_t* a_start;
  const uint32_t* a_end;
  const uint32_t* a_cumx;     // bases of the list's intervals before interval i
  const uint32_t* a_grid;     // grid[g] = #starts < (g << shift[c]); grid[cells[c]] = m
  const int64_t* a_off;       // n_tracks*n_contigs+1
  const int64_t* g_off;       // n_tracks*n_contigs+1
  const int32_t* c_shift;     // n_contigs
  const int32_t* c_cells;     // n_contigs
  const int64_t* cws_nseg;    // n_contigs
  int32_t n_contigs, n_tracks;
  int32_t n_samples;          // lists in this launch
  // outputs: slot(k, t, s) = out[(k*n_tracks + t)*out_stride + out_begin + s]
  int64_t* out;
  int64_t out_stride;
  int64_t out_begin;
  int32_t counter_slot[GAT_NUM_COUNTERS_DEV];   // output index k of each counter id, or -1
  int32_t tracks_per_block;
  int32_t samples_per_block;
  int32_t lds_entries;        // staging capacity (intervals); 0 => read annotations from global
  int32_t lds_grid;           // staging capacity (grid words)
  uint32_t* part;             // [n_contigs][3][n_tracks][n_samples] per-contig partial counts
  // merged multi-track index (k_count_merged): per contig all tracks' intervals sorted by start
  const uint2* mz;            // entries {start, length:16 | track:16}; long intervals cut into pieces; a sentinel start ends every contig
  const int64_t* mz_off;      // n_contigs+1
  const uint32_t* mfirst;     // per contig and position cell: first entry that reaches into the cell or starts in / after it
  const uint4* mcell;         // BLK 1: per cell a 32-byte record {first, -, entry first, entry first + 1, -, -} (same offsets, x 2)
  const int64_t* mf_off;      // n_contigs+1
  const int32_t* m_shift;     // n_contigs: log2 of the cell width
  const int32_t* m_cells;     // n_contigs
  const int32_t* m_slot_off;  // kMergedSlots+1: the contigs of an XCD slot are m_slot_contigs[m_slot_off[x] .. m_slot_off[x+1])
  const int32_t* m_slot_contigs;
  unsigned long long* mstat;  // k_count_merged's own traffic, for the byte model of its roofline: 256 pairs {4-byte words of
                              // index read (grid cells, cell records, entries), sample segments looked up}, a wave adds to
                              // pair blockIdx % 256 once per sample (nullptr: not kept)
  // split path without k_finalize (k_count_seg<.., PATCH>; contig == unit): a unit k_tail finished is read as (merged
  // list in seg_merged, k_tail's record), any other from seg as usual
  const uint2* seg_merged;
  const int32_t* unit_pos;    // unit id -> launch position (patch / st2 are indexed by it), -1: inactive
  const int4* st2;            // .x = merged segments
  const int32_t* patch;       // TailPatch records as words (kPatch*), patch_stride words each
  int32_t patch_stride, n_units;
  int32_t rec_stride;         // (GAT_REC: st2 / patch are [launch position][rec_stride])
  // Isochore problems counted from contig lists that k_contig<., true> only CONCATENATED (nucleotide counters through the merged
  // index; round 6): no sort, no merge(0) -- the index look-ups do not care for the order, and what fromIsochores' merge(0)
  // (gat/Engine.pyx:2857-2876) would have united -- a segment of one unit reaching over the end of its workspace piece into a
  // neighbouring unit's segment -- is found by k_units_overlap among the CANDIDATES k_contig notes (segments of units flagged in
  // their record's `pad` whose cells of the boundary map hold a workspace boundary) and taken off the partial sums again.  For
  // that kernel: the contig's units = units[contig_unit_off[c] ..), cu_rec = {unit id, its place in a sample's slab, its launch
  // position (-1: inactive), 0}; a unit k_tail finished = (merged list in seg_units_merged, its record), any other its final list
  // in seg_units with unit_n segments.
  const int32_t* contig_unit_off;
  const int4* cu_rec;
  const int32_t* unit_n;      // [sample][n_units]
  const uint2* seg_units;
  const uint2* seg_units_merged;
  uint4* cand;                // {sample, entry of cu_rec, start, end}: kCandSlots regions of cand_cap entries, a workgroup appends to
                              // region blockIdx % kCandSlots (one counter for the chip's appends would be most of the time)
  uint32_t* cand_count;       // [0, kCandSlots) candidates appended per region; [kCandSlots]: bit 0 the overlaps were not pairwise, bit 1 a
                              // region was too small; [kCandSlots + 1] overlaps taken off, [kCandSlots + 2] candidates in all (k_units_overlap adds them up)
  uint32_t cand_cap;
};

struct AnnoView {
  const uint32_t* start;
  const uint32_t* end;
  const uint32_t* cumx;
  const uint32_t* grid;
  int m, shift, cells;
};

// quantities of one sample segment x = [xs, xe) against one annotation list Y (Y.m >= 1).
//   k1 = #starts < xs comes from the position grid (about one start per cell) plus a short scan;
//   the scan leaves sk = start[k1], the first start >= xs.  A sample segment is far shorter than
//   the gaps between annotation intervals, so almost always sk >= xe: x then meets only interval
//   k1-1 and overlap = min(xe, end[k1-1]) - xs if that interval reaches past xs.  Otherwise the
//   general form F(xe) - F(xs), F(p) = cumx[k-1] + min(p, end[k-1]) - start[k-1], is used (behind a
//   wave-uniform __any test).  Measured alternatives that were slower: predicated fixed-count scans
//   (+7 %), eight lookups advanced in lock step (+37 %).
// SENT: start[m] holds the sentinel 0xffffffff (LDS copies have room for it), which ends both scans by itself: no
// second grid read, no index checks.
template <bool WANT_HITS, bool SENT>
__device__ __forceinline__ void seg_vs_anno(const AnnoView& Y, uint32_t xs, uint32_t xe,
                                            uint32_t& ov, uint32_t& hit, uint32_t& midhit) {
  uint32_t g = xs >> Y.shift;
  g = g < (uint32_t)(Y.cells - 1) ? g : (uint32_t)(Y.cells - 1);
  int k = (int)Y.grid[g];
  uint32_t sk;
  if (SENT) {
    sk = Y.start[k];
    while (sk < xs) { ++k; sk = Y.start[k]; }              // stops at the cell's end at the latest: that start is > xs
  } else {
    const int hi = (int)Y.grid[g + 1];
    sk = k < Y.m ? Y.start[k] : 0xffffffffu;
    while (k < hi && sk < xs) { ++k; sk = k < Y.m ? Y.start[k] : 0xffffffffu; }
  }
  const int k1 = k;
  const uint32_t pe_raw = Y.end[k1 > 0 ? k1 - 1 : 0];
  const bool reach = k1 > 0 && pe_raw > xs;                // the last interval starting before xs reaches past it
  ov = reach ? (xe < pe_raw ? xe : pe_raw) - xs : 0u;
  hit = 0; midhit = 0;
  if (WANT_HITS) {
    const uint32_t mid = xs + (xe - xs) / 2u;
    hit = reach ? 1u : 0u;                                 // first interval with end > xs is k1-1 and starts before xs
    midhit = (reach && mid < pe_raw) ? 1u : 0u;
  }
  const bool slow = sk < xe;                               // an interval starts inside x
  if (__any(slow)) {
    if (slow) {
      const uint32_t pe = k1 > 0 ? pe_raw : 0u;
      int k2 = k1 + 1;                                     // #starts < xe
      if (SENT) { while (Y.start[k2] < xe) ++k2; }
      else { while (k2 < Y.m && Y.start[k2] < xe) ++k2; }
      uint32_t f1 = 0;
      if (k1 > 0) f1 = Y.cumx[k1 - 1] + (xs < pe ? xs : pe) - Y.start[k1 - 1];
      const uint32_t ps2 = Y.start[k2 - 1], pe2 = Y.end[k2 - 1];
      ov = Y.cumx[k2 - 1] + (xe < pe2 ? xe : pe2) - ps2 - f1;
      if (WANT_HITS) {
        // first interval with end > xs: k1-1 if it reaches past xs, else k1 (which starts inside x)
        hit = 1;
        const uint32_t mid = xs + (xe - xs) / 2u;
        if (pe > xs) midhit = mid < pe ? 1u : 0u;
        else midhit = (sk <= mid && mid < Y.end[k1]) ? 1u : 0u;
      }
    }
  }
}

constexpr int kCountXR = 8;      // sample segments held per lane per pass (512 per wave)
#ifndef GAT_COUNT_NR
#define GAT_COUNT_NR 2
#endif
constexpr int kCountNR = GAT_COUNT_NR;   // ... of which segs_vs_pairs looks up this many side by side (4 pairs of registers each)
// The same quantities against a list STAGED in LDS as {start, end} pairs (k_count_seg<true, ..>), NR sample segments per lane
// side by side.  seg_vs_anno is a chain of four to five dependent LDS reads per segment (grid cell -> start -> [next start]
// -> end), walked segment by segment: the counters of round 4 (profiles/r04_config2_pmc.txt) have the kernel's waves parked at
// s_waitcnt for 78 % of their cycles with the vector unit 27 % and the LDS 15 % busy.  Here a segment's look-up is TWO
// dependent steps, and all NR segments of a lane take each step together: (1) the grid cells, (2) four consecutive pairs
// P[k-1 .. k+2] around the cell's first start (two ds_read2_b64).  With at most two starts between the cell's beginning and xs
// -- the grid holds about one start per two to four cells -- they hold everything: k1 = k + (s0 < xs) + (s1 < xs), the first
// start >= xs and the end of the interval in front of it.  What they do not cover (a third start in front of xs; an interval
// that starts inside x: the general form F(xe) - F(xs) over the running lengths) is redone lane by lane behind a wave-uniform
// test, as before.  Entry -1 of a list is {0, 0} (no interval in front of the first), three {~0, ~0} sentinels end it.
struct PairView {
  const uint2* P;             // P[-1] .. P[m + 2]
  const uint32_t* cumx;       // cumx[i] = bases of the list's intervals before interval i
  const uint32_t* grid;
  int shift, cells;
};
template <bool WANT_HITS, int NR>
__device__ __forceinline__ void segs_vs_pairs(const PairView& Y, const uint2* x, uint32_t& ov_sum, uint32_t& hit_sum,
                                              uint32_t& mid_sum) {
  int k[NR];
#pragma unroll
  for (int r = 0; r < NR; ++r) {
    uint32_t g = x[r].x >> Y.shift;
    g = g < (uint32_t)(Y.cells - 1) ? g : (uint32_t)(Y.cells - 1);
    k[r] = (int)Y.grid[g];
  }
  uint2 pm[NR], p0[NR], p1[NR], p2[NR];
#pragma unroll
  for (int r = 0; r < NR; ++r) {
    const uint2* q = Y.P + k[r];
    pm[r] = q[-1]; p0[r] = q[0]; p1[r] = q[1]; p2[r] = q[2];
  }
  bool any_slow = false;
  uint32_t slow_mask = 0;
#pragma unroll
  for (int r = 0; r < NR; ++r) {
    const uint32_t xs = x[r].x, xe = x[r].y;
    const bool c0 = p0[r].x < xs, c1 = p1[r].x < xs;                 // (sorted starts: c1 implies c0)
    const uint32_t sk = c1 ? p2[r].x : (c0 ? p1[r].x : p0[r].x);     // the first start >= xs (if p2's is)
    const uint32_t pe_raw = c1 ? p1[r].y : (c0 ? p0[r].y : pm[r].y); // end of the last interval that starts before xs
    const bool reach = pe_raw > xs;
    ov_sum += reach ? (xe < pe_raw ? xe : pe_raw) - xs : 0u;
    if (WANT_HITS) {
      const uint32_t mid = xs + (xe - xs) / 2u;
      hit_sum += reach ? 1u : 0u;
      mid_sum += (reach && mid < pe_raw) ? 1u : 0u;
    }
    // a third start in front of xs, or an interval starting inside x (padding -- xs = ~0 -- is neither: its xe is ~0 too)
    const bool slow = xs != 0xffffffffu && (p2[r].x < xs || sk < xe);
    slow_mask |= slow ? 1u << r : 0u;
    any_slow |= slow;
  }
  if (__any(any_slow)) {
#pragma unroll
    for (int r = 0; r < NR; ++r) {
      if (!(slow_mask >> r & 1u)) continue;
      const uint32_t xs = x[r].x, xe = x[r].y;
      // what the fast form added for this segment comes off again, then the general form (seg_vs_anno's)
      {
        const bool c0 = p0[r].x < xs, c1 = p1[r].x < xs;
        const uint32_t pe_raw = c1 ? p1[r].y : (c0 ? p0[r].y : pm[r].y);
        const bool reach = pe_raw > xs;
        ov_sum -= reach ? (xe < pe_raw ? xe : pe_raw) - xs : 0u;
        if (WANT_HITS) {
          const uint32_t mid = xs + (xe - xs) / 2u;
          hit_sum -= reach ? 1u : 0u;
          mid_sum -= (reach && mid < pe_raw) ? 1u : 0u;
        }
      }
      int k1 = k[r];
      uint32_t sk = Y.P[k1].x;
      while (sk < xs) { ++k1; sk = Y.P[k1].x; }
      const uint32_t pe_raw = Y.P[k1 - 1].y;                         // (P[-1] = {0, 0})
      const bool reach = pe_raw > xs;
      uint32_t ov = reach ? (xe < pe_raw ? xe : pe_raw) - xs : 0u, hit = reach ? 1u : 0u, midhit = 0;
      const uint32_t mid = xs + (xe - xs) / 2u;
      if (WANT_HITS) midhit = (reach && mid < pe_raw) ? 1u : 0u;
      if (sk < xe) {
        int k2 = k1 + 1;
        while (Y.P[k2].x < xe) ++k2;
        uint32_t f1 = 0;
        if (k1 > 0) f1 = Y.cumx[k1 - 1] + (xs < pe_raw ? xs : pe_raw) - Y.P[k1 - 1].x;
        const uint2 l2 = Y.P[k2 - 1];
        ov = Y.cumx[k2 - 1] + (xe < l2.y ? xe : l2.y) - l2.x - f1;
        if (WANT_HITS) {
          hit = 1;
          if (pe_raw > xs) midhit = mid < pe_raw ? 1u : 0u;
          else midhit = (sk <= mid && mid < Y.P[k1].y) ? 1u : 0u;
        }
      }
      ov_sum += ov;
      if (WANT_HITS) { hit_sum += hit; mid_sum += midhit; }
    }
  }
}


// One block per (sample chunk, track tile, contig): the tile's annotation slices of that contig
// (starts / ends / cumulated lengths + position grid) are staged into LDS once and every wave
// streams its samples' segment lists against them.  Per (sample, track, contig) the wave leaves
// three uint32 partials (overlap bases, segments hit, midpoint hits) in `part`; k_count_finish
// adds them up over the contigs in reference order.
// PATCH: the unit lists are taken as k_tail left them -- merged list (trimmed in place, emptied segments read as padding)
// + the record's extras -- so k_finalize and its round trip of the lists through HBM are not needed when only counts are
// asked for.  The sums do not depend on the order or on the padding; the elements are those k_finalize would write.
// a sampled list is read once by a count kernel: non-temporal, so that it does not push the annotation index out of the L2
__device__ __forceinline__ uint2 ld_list_nt(const uint2* p) {
  typedef uint32_t v2u_ __attribute__((ext_vector_type(2)));
  const v2u_ t = __builtin_nontemporal_load(reinterpret_cast<const v2u_*>(p));
  return make_uint2(t.x, t.y);
}
template <bool STAGED, bool WANT_HITS, bool PATCH = false>
__global__ __launch_bounds__(256) void k_count_seg(CountArgs A) {
  extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
  const int TT = A.tracks_per_block, SC = A.samples_per_block;
  int32_t* tile_off = reinterpret_cast<int32_t*>(lds);                // TT+1
  uint32_t* stage = reinterpret_cast<uint32_t*>(tile_off + ((TT + 1 + 3) & ~3));
  const int E = A.lds_entries;
  uint32_t* st_grid = stage + 3 * E;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;      // (the wave index as a scalar -- readfirstlane -- changes nothing: 0.218 ms either way)
  // (track tile, contig) = the linear index over grid y and z, tiles fastest: neither count is bound by a grid dimension
  const int n_tiles = (A.n_tracks + TT - 1) / TT;
  // (measured and dropped, round 5: the (tile, contig) pair as the fastest index of the launch, so that workgroups resident
  //  together read neighbouring regions of the same samples' slab: 0.238 against 0.214 ms on config 2)
  const int64_t lin = (int64_t)blockIdx.y + (int64_t)blockIdx.z * gridDim.y;
  const int chunk = (int)blockIdx.x;
  const int c = (int)(lin / n_tiles);
  if (c >= A.n_contigs) return;                                     // (whole workgroup, before any barrier)
  const int s0 = chunk * SC, t0 = (int)(lin % n_tiles) * TT;
  const int nt = min(TT, A.n_tracks - t0), ns = min(SC, A.n_samples - s0);
  const int shift = A.c_shift[c], cells = A.c_cells[c];
  // STAGED: per track of the tile {start, end} pairs -- entry -1 = {0, 0}, three sentinels {~0, ~0} behind the last -- the
  // running lengths and the position grid (segs_vs_pairs)
  uint2* st_pairs = reinterpret_cast<uint2*>(stage);
  uint32_t* st_cumx = stage + 2 * E;
  if (STAGED) {
    if (tid == 0) {
      int o = 0;
      for (int t = 0; t < nt; ++t) {
        tile_off[t] = o;                                    // (+4: the entry in front and the sentinels)
        o += (int)(A.a_off[(int64_t)(t0 + t) * A.n_contigs + c + 1] - A.a_off[(int64_t)(t0 + t) * A.n_contigs + c]) + 4;
      }
      tile_off[nt] = o;
    }
    __syncthreads();
    for (int t = 0; t < nt; ++t) {
      const int64_t g = A.a_off[(int64_t)(t0 + t) * A.n_contigs + c];
      const int o = tile_off[t], m = tile_off[t + 1] - o - 4;
      for (int i = tid; i < m; i += 256) {
        st_pairs[o + 1 + i] = make_uint2(A.a_start[g + i], A.a_end[g + i]);
        st_cumx[o + 1 + i] = A.a_cumx[g + i];
      }
      if (tid == 0) st_pairs[o] = make_uint2(0u, 0u);
      if (tid < 3) st_pairs[o + 1 + m + tid] = make_uint2(0xffffffffu, 0xffffffffu);
      const int64_t gg = A.g_off[(int64_t)(t0 + t) * A.n_contigs + c];
      for (int i = tid; i <= cells; i += 256) st_grid[t * (cells + 1) + i] = A.a_grid[gg + i];
    }
  }
  __syncthreads();
  const int64_t qstride = (int64_t)A.n_tracks * A.n_samples;
  const int upos = PATCH ? A.unit_pos[A.n_index[c]] : -1;
  const int nidx = A.n_index[c];
  const int64_t coff = A.c_off[c];
  // what a wave needs to know of a list before it can ask for its segments -- fetched one list ahead (the loop is a
  // chain of two dependent round trips per list otherwise): length; PATCH: state, extras, merged length of the unit
  struct Meta { int n, state, nE, nM; };
  auto fetch_meta = [&](int sl_) -> Meta {
    Meta m = {0, 0, 0, 0};
    if (sl_ < ns) {
      const int64_t s_ = s0 + sl_;
      m.n = A.n_arr[s_ * A.n_stride + nidx];
      if (PATCH && upos >= 0) {
        const int64_t sa = GAT_REC(A, s_, upos);
        const int32_t* __restrict__ R = A.patch + sa * A.patch_stride;
        m.state = R[kPatchState]; m.nE = R[kPatchNExtra]; m.nM = A.st2[sa].x;
      }
    }
    return m;
  };
  // a list as the wave reads it.  PATCH, a unit k_tail finished: merged list (nU segments, trimmed in place) and the
  // record's extras behind it
  struct View { int n, nU; const uint2* X; const uint2* Rex; };
  const uint2 kPad = make_uint2(0xffffffffu, 0xffffffffu);              // padding: beyond every interval, adds 0
  auto view_of = [&](const Meta& m, int sl_) -> View {
    const int64_t s_ = s0 + sl_;
    View v;
    v.n = m.n; v.nU = m.n;
    v.X = A.seg + s_ * A.seg_stride + coff;
    v.Rex = v.X;
    if (PATCH && m.state == 1) {
      v.nU = m.nM; v.n = m.nM + m.nE;
      v.X = A.seg_merged + s_ * A.seg_stride + coff;
      v.Rex = reinterpret_cast<const uint2*>(A.patch + GAT_REC(A, s_, upos) * A.patch_stride + kPatchExtra) - m.nM;
    }
    return v;
  };
  // Every load is issued whatever the lane's index: an index beyond the list re-reads its last element (a line the wave
  // fetches anyway) and is turned into padding afterwards.  The form `if (i >= n) return pad; y = *p; return y.x == y.y ? ..`
  // put each load, its s_waitcnt vmcnt(0) and the test into a branch of its own: the eight loads of a list went out one
  // after the other, seven exposed round trips per list (round 5: found in the assembly; the waves were parked 78 % of
  // their cycles, profiles/r04_config2_pmc.txt)
  auto load_raw = [&](const View& v, int i) -> uint2 {
    const int ic = i < v.n ? i : (v.n > 0 ? v.n - 1 : 0);
    if (!PATCH) return v.X[ic];                               // (non-temporal here: 0.288 -> 0.283 ms on config 2, not worth a second form)
    return *(ic < v.nU ? v.X + ic : v.Rex + ic);
  };
  auto load_fix = [&](const View& v, int i, uint2 y) -> uint2 {
    if (i >= v.n) return kPad;
    return (PATCH && y.x == y.y) ? kPad : y;                   // (emptied by the trim: merge(0) drops it)
  };
  // (measured and dropped, rounds 2 and 5: the next list's segments on their way while this one is counted -- 16 to 20 more
  //  registers per lane cost more occupancy than the overlap gains: 0.29 -> 0.33 ms on config 2 in round 2; behind the batched
  //  loads of round 5 0.218 -> 0.262)
  Meta next = fetch_meta(wave);
  for (int sl = wave; sl < ns; sl += 4) {
    const int s = s0 + sl;
    const View V = view_of(next, sl);
    next = fetch_meta(sl + 4);
    const int n = V.n;
    const bool one_pass = n <= kWave * kCountXR;
    uint2 x[kCountXR];
    if (one_pass) {
#pragma unroll
      for (int r = 0; r < kCountXR; ++r) x[r] = load_raw(V, r * kWave + lane);
#pragma unroll
      for (int r = 0; r < kCountXR; ++r) x[r] = load_fix(V, r * kWave + lane, x[r]);
    }
    for (int t = 0; t < nt; ++t) {
      uint32_t ov = 0, hit = 0, mid = 0;
      if constexpr (STAGED) {
        const int o = tile_off[t];
        PairView Y;
        Y.shift = shift; Y.cells = cells;
        Y.P = st_pairs + o + 1; Y.cumx = st_cumx + o + 1; Y.grid = st_grid + t * (cells + 1);
        if (tile_off[t + 1] - o - 4 > 0) {
          for (int base = 0; base < n; base += kWave * kCountXR) {
            if (!one_pass) {
#pragma unroll
              for (int r = 0; r < kCountXR; ++r) x[r] = load_raw(V, base + r * kWave + lane);
#pragma unroll
              for (int r = 0; r < kCountXR; ++r) x[r] = load_fix(V, base + r * kWave + lane, x[r]);
            }
            // (wave-uniform: a list's last pass often fills half of the rounds -- config 2's lists are ~400 segments)
#pragma unroll
            for (int r0 = 0; r0 < kCountXR; r0 += kCountNR) {
              if (base + r0 * kWave >= n) break;
              segs_vs_pairs<WANT_HITS, kCountNR>(Y, x + r0, ov, hit, mid);
            }
          }
          ov = wave_total_u32(ov);                          // uint32 accumulate within the contig (:1034)
          if (WANT_HITS) { hit = wave_total_u32(hit); mid = wave_total_u32(mid); }
        }
      } else {
        AnnoView Y;
        Y.shift = shift; Y.cells = cells;
        const int64_t g = A.a_off[(int64_t)(t0 + t) * A.n_contigs + c];
        Y.start = A.a_start + g; Y.end = A.a_end + g; Y.cumx = A.a_cumx + g;
        Y.grid = A.a_grid + A.g_off[(int64_t)(t0 + t) * A.n_contigs + c];
        Y.m = (int)(A.a_off[(int64_t)(t0 + t) * A.n_contigs + c + 1] - g);
        if (Y.m > 0) {
          for (int base = 0; base < n; base += kWave * kCountXR) {
            if (!one_pass) {
#pragma unroll
              for (int r = 0; r < kCountXR; ++r) x[r] = load_raw(V, base + r * kWave + lane);
#pragma unroll
              for (int r = 0; r < kCountXR; ++r) x[r] = load_fix(V, base + r * kWave + lane, x[r]);
            }
#pragma unroll
            for (int r = 0; r < kCountXR; ++r) {
              if (base + r * kWave >= n) break;              // (wave-uniform: whole rounds beyond the list are skipped)
              uint32_t o1, h1, m1;
              seg_vs_anno<WANT_HITS, false>(Y, x[r].x, x[r].y, o1, h1, m1);
              ov += o1; hit += h1; mid += m1;
            }
          }
          ov = wave_total_u32(ov);                          // uint32 accumulate within the contig (:1034)
          if (WANT_HITS) { hit = wave_total_u32(hit); mid = wave_total_u32(mid); }
        }
      }
      if (lane == 0) {
        // part[((c*3 + q)*n_tracks + t)*n_samples + s]
        const int64_t b = (((int64_t)c * 3) * A.n_tracks + t0 + t) * A.n_samples + s;
        A.part[b] = ov;
        if (WANT_HITS) { A.part[b + qstride] = hit; A.part[b + 2 * qstride] = mid; }
      }
    }
  }
}

// nucleotide-overlap / nucleotide-density against MANY tracks: one look-up per sample segment instead of one per
// (segment, track).  A sample segment overlaps an interval of very few tracks (config 3: 0.8 of 100 on average), so the
// host merges all tracks' intervals of a contig into one list sorted by start -- 8-byte entries {start, length:16 |
// track:16}; intervals longer than a bound are cut into pieces, which changes no sum -- with a position grid first[cell] =
// the first entry that reaches into the cell or starts in or behind it.  A lane scans from first[cell of x.start] while
// start < x.end and adds min(ends) - max(starts) of every entry it meets to the track's uint32 accumulator in LDS
// (gat/SegmentList.pyx:1026-1076 sums the same pairs per track; uint32 addition is order independent).
//   The look-ups are 8-byte gathers all over a contig's index, bound by the rate at which L2 hands out lines.  The work
// is dealt so that an XCD's L2 serves them: a workgroup owns (contig, group of samples), the contigs are spread over
// eight slots by the host (balanced by their entries), slot = blockIdx % 8 -- the stride with which workgroups are
// observed to go round the XCDs (speed only; any placement gives the same sums) -- and a slot walks its contigs one after
// the other.  Every wave takes whole samples (four look-ups per lane in flight): its accumulators are wave-private
// LDS, there is no workgroup barrier.  Per (contig, sample) the T partial sums go to part[c][s][t];
// k_count_merged_finish adds them over the contigs in reference order.
//   Measured and dropped: the same join streamed through LDS (index cut into chunks of 4 096 entries staged per workgroup,
// every wave advancing a cursor through its samples' sorted lists, look-ups in LDS).  It is bit-exact and no faster on
// config 3 (1.80 vs 1.82 ms per 10 000 samples) and slower on the config-4 shape (54.9 vs 34.1 ms per 4 096 samples: with
// 1 000 tracks only eight samples' accumulators fit beside a chunk, so a contig's 3.3 MB are staged 512 times).
// (kMergedSlots: gat_types.h)
#ifndef GAT_MERGED_THREADS
#define GAT_MERGED_THREADS 256
#endif
#ifndef GAT_MERGED_KR_BLOCKS
#define GAT_MERGED_KR_BLOCKS 4     // ... in the block form (config-4 shape: 1 -> 36.5 ms, 2 -> 36.9, 4 -> 34.4 per 12 500 samples)
#endif
#ifndef GAT_MERGED_KR
#define GAT_MERGED_KR 8            // look-ups a lane has in flight in the pair / cell-record forms (config 3: 2 -> 1.32 ms, 4 -> 1.27, 8 -> 1.16)
#endif
constexpr int kMergedThreads = GAT_MERGED_THREADS;
// PATCH (as in k_count_seg): a unit k_tail finished is read as merged list + the record's extras, no k_finalize.
// BLK: how a scan fetches the index: 8 = 64-byte blocks of eight entries; 2 = pairs (one 16-byte load); 1 = the first two
// entries out of the grid cell's own 32-byte record, pairs behind them.  The host picks by how many entries a scan is
// expected to pass (AnnoDev::merged_block): config 3 / config 5 pass 3 (blocks: 1.27 -> 1.49 ms and 1.76 -> 2.21 ms; cell
// records: see DESIGN.md), the config-4 shape passes 15 and takes blocks (48.7 -> 37.0 ms per 12 500 samples).
#ifndef GAT_MERGED_NO_NT
#define GAT_LD_LIST(P) ld_list_nt(P)     /* config-4 shape: k_count_merged 34.4 -> 32.9 ms; with its partial sums stored the same way 29.6 */
#else
#define GAT_LD_LIST(P) (*(P))
#endif
template <bool PATCH, int BLK>
__global__ __launch_bounds__(kMergedThreads) void k_count_merged(CountArgs A) {
  extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
  const int T = A.n_tracks;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  uint32_t* acc = lds + (size_t)wave * T;
  const int slot = blockIdx.x % kMergedSlots, q = blockIdx.x / kMergedSlots;
  const int SG = A.samples_per_block, n_groups = (A.n_samples + SG - 1) / SG;
  const int ci = q / n_groups, grp = q - ci * n_groups;
  if (ci >= A.m_slot_off[slot + 1] - A.m_slot_off[slot]) return;
  const int c = A.m_slot_contigs[A.m_slot_off[slot] + ci];
  const uint2* __restrict__ Z = A.mz + A.mz_off[c];
  const uint32_t* __restrict__ F = A.mfirst + A.mf_off[c];
  const int shift = A.m_shift[c];
  const uint32_t last = (uint32_t)(A.m_cells[c] - 1);
  const int nidx = A.n_index[c], coff = A.c_off[c];
  for (int t = lane; t < T; t += kWave) acc[t] = 0u;
  const int s_end = min(A.n_samples, (grp + 1) * SG);
  const int upos = PATCH ? A.unit_pos[nidx] : -1;
  for (int s = grp * SG + wave; s < s_end; s += kMergedThreads / kWave) {
    int n = A.n_arr[(int64_t)s * A.n_stride + nidx];
    const uint2* __restrict__ X = A.seg + (int64_t)s * A.seg_stride + coff;
    int nU = n;
    const uint2* __restrict__ Rex = X;
    if (PATCH && upos >= 0) {
      const int64_t sa = GAT_REC(A, s, upos);
      const int32_t* __restrict__ R = A.patch + sa * A.patch_stride;
      if (R[kPatchState] == 1) {
        nU = A.st2[sa].x;
        n = nU + R[kPatchNExtra];
        X = A.seg_merged + (int64_t)s * A.seg_stride + coff;
        Rex = reinterpret_cast<const uint2*>(R + kPatchExtra) - nU;
      }
    }
    uint32_t n_ent = 0;                                              // 4-byte words of index this lane's scans looked at: per segment the
                                                                     // grid cell + two per entry (the one that ended the scan too)
    if constexpr (BLK == 8) {
    constexpr int kR = GAT_MERGED_KR_BLOCKS;                         // segments per lane whose look-ups are in flight together
    for (int base = 0; base < n; base += kR * kWave) {
      uint2 x[kR];
      uint32_t k[kR];
#pragma unroll
      for (int r = 0; r < kR; ++r) {
        const int i = base + r * kWave + lane;
        // (an empty segment -- padding, or what a trim emptied -- meets nothing: start < 0 never holds)
        x[r] = i < n ? GAT_LD_LIST(PATCH ? (i < nU ? X + i : Rex + i) : X + i) : make_uint2(0u, 0u);
      }
#pragma unroll
      for (int r = 0; r < kR; ++r) { const uint32_t g = x[r].x >> shift; k[r] = F[g < last ? g : last]; }
      // entries are fetched a 64-byte block (eight of them) at a time: the four 16-byte loads of a block fall into one
      // 128-byte line and are issued back to back, so the line is asked of the L2 once -- the scan of a segment passes 3
      // (config 3) to 15 (1 000 tracks) entries, fetched in pairs every pair was a request of its own (too many lanes
      // are scanning for a line to survive in the 32 KB L1 between two loads), and the rate of those requests is what
      // bounds the kernel.  A block starts at a multiple of eight: the entries in front of first[cell] in it ended before
      // the cell began and add nothing (hi > lo fails); a contig's entries start at a multiple of eight and end with
      // sentinels (start 0xffffffff), and eight more stand behind the last contig.
      const uint4* __restrict__ Z4 = reinterpret_cast<const uint4*>(Z);
      uint4 zz[kR][4];
#pragma unroll
      for (int r = 0; r < kR; ++r) {
        const uint32_t b4 = (k[r] >> 3) << 2;
#pragma unroll
        for (int w = 0; w < 4; ++w) zz[r][w] = Z4[b4 + w];
      }
#pragma unroll
      for (int r = 0; r < kR; ++r) {
        uint32_t blk = k[r] >> 3;
        n_ent += 1u;                                                 // (the grid cell)
        uint4 q0 = zz[r][0], q1 = zz[r][1], q2 = zz[r][2], q3 = zz[r][3];
        while (true) {
          bool more = true;
#define GAT_MERGED_ENTRY(EX, EY)                                                                     \
          if (more) {                                                                               \
            n_ent += 2u;                                                                            \
            if (!((EX) < x[r].y)) more = false;       /* the contig's sentinel start 0xffffffff ends the scan */ \
            else {                                                                                  \
              const uint32_t ze = (EX) + ((EY) & 0xffffu);                                          \
              const uint32_t lo = (EX) > x[r].x ? (EX) : x[r].x, hi = ze < x[r].y ? ze : x[r].y;      \
              if (hi > lo) atomicAdd(&acc[(EY) >> 16], hi - lo);                                    \
            }                                                                                       \
          }
          GAT_MERGED_ENTRY(q0.x, q0.y) GAT_MERGED_ENTRY(q0.z, q0.w) GAT_MERGED_ENTRY(q1.x, q1.y) GAT_MERGED_ENTRY(q1.z, q1.w)
          GAT_MERGED_ENTRY(q2.x, q2.y) GAT_MERGED_ENTRY(q2.z, q2.w) GAT_MERGED_ENTRY(q3.x, q3.y) GAT_MERGED_ENTRY(q3.z, q3.w)
#undef GAT_MERGED_ENTRY
          if (!more) break;
          ++blk;
          q0 = Z4[blk * 4]; q1 = Z4[blk * 4 + 1]; q2 = Z4[blk * 4 + 2]; q3 = Z4[blk * 4 + 3];
        }
      }
    }
    } else if constexpr (BLK == 1) {
    // short scans (three entries on configs 3 and 5): the grid cell holds, beside the index of its first entry, that entry
    // and the next -- one 32-byte record, one request to the L2 where the cell and the first pair of entries were two; only
    // a scan that passes more goes on in the index, in pairs
    constexpr int kR = GAT_MERGED_KR;
    const uint4* __restrict__ FC = A.mcell + 2 * A.mf_off[c];
    const uint4* __restrict__ Z2 = reinterpret_cast<const uint4*>(Z);
    for (int base = 0; base < n; base += kR * kWave) {
      uint2 x[kR];
      uint4 c0[kR], c1[kR];
#pragma unroll
      for (int r = 0; r < kR; ++r) {
        const int i = base + r * kWave + lane;
        x[r] = i < n ? GAT_LD_LIST(PATCH ? (i < nU ? X + i : Rex + i) : X + i) : make_uint2(0u, 0u);
      }
#pragma unroll
      for (int r = 0; r < kR; ++r) {
        const uint32_t g = x[r].x >> shift, gi = g < last ? g : last;
        c0[r] = FC[2 * gi];
        c1[r] = FC[2 * gi + 1];
      }
#pragma unroll
      for (int r = 0; r < kR; ++r) {
        n_ent += 3u;                                                 // (words: the cell and the first entry)
        uint32_t kk = c0[r].x + 2u;
        bool more = true;
        {
          const uint2 e = make_uint2(c0[r].z, c0[r].w);
          if (!(e.x < x[r].y)) more = false;
          else {
            const uint32_t ze = e.x + (e.y & 0xffffu);
            const uint32_t lo = e.x > x[r].x ? e.x : x[r].x, hi = ze < x[r].y ? ze : x[r].y;
            if (hi > lo) atomicAdd(&acc[e.y >> 16], hi - lo);
          }
        }
        if (more) {
          n_ent += 2u;
          const uint2 e = make_uint2(c1[r].x, c1[r].y);
          if (!(e.x < x[r].y)) more = false;
          else {
            const uint32_t ze = e.x + (e.y & 0xffffu);
            const uint32_t lo = e.x > x[r].x ? e.x : x[r].x, hi = ze < x[r].y ? ze : x[r].y;
            if (hi > lo) atomicAdd(&acc[e.y >> 16], hi - lo);
          }
        }
        if (more) {
          const uint32_t k0 = kk;
          uint4 q = Z2[kk >> 1];
          while (true) {
            const uint2 e = (kk & 1u) ? make_uint2(q.z, q.w) : make_uint2(q.x, q.y);
            if (!(e.x < x[r].y)) break;                              // the contig's sentinel start 0xffffffff ends the scan
            const uint32_t ze = e.x + (e.y & 0xffffu);
            const uint32_t lo = e.x > x[r].x ? e.x : x[r].x, hi = ze < x[r].y ? ze : x[r].y;
            if (hi > lo) atomicAdd(&acc[e.y >> 16], hi - lo);
            ++kk;
            if (!(kk & 1u)) q = Z2[kk >> 1];
          }
          n_ent += 2u * (kk - k0 + 1u);
        }
      }
    }
    } else {
    constexpr int kR = GAT_MERGED_KR;
    for (int base = 0; base < n; base += kR * kWave) {
      uint2 x[kR];
      uint32_t k[kR];
#pragma unroll
      for (int r = 0; r < kR; ++r) {
        const int i = base + r * kWave + lane;
        // (an empty segment -- padding, or what a trim emptied -- meets nothing: start < 0 never holds)
        x[r] = i < n ? GAT_LD_LIST(PATCH ? (i < nU ? X + i : Rex + i) : X + i) : make_uint2(0u, 0u);
      }
#pragma unroll
      for (int r = 0; r < kR; ++r) { const uint32_t g = x[r].x >> shift; k[r] = F[g < last ? g : last]; }
      // entries are fetched two at a time (16 bytes, one request to the L2 instead of two: the scan of a segment against
      // 1 000 tracks passes eight entries, and the rate of those requests is what bounds the kernel); a contig's entries
      // start at an even index and end with two sentinels
      const uint4* __restrict__ Z2 = reinterpret_cast<const uint4*>(Z);
      uint4 zz[kR];
#pragma unroll
      for (int r = 0; r < kR; ++r) zz[r] = Z2[k[r] >> 1];
#pragma unroll
      for (int r = 0; r < kR; ++r) {
        uint32_t kk = k[r];
        uint4 q = zz[r];
        while (true) {
          const uint2 e = (kk & 1u) ? make_uint2(q.z, q.w) : make_uint2(q.x, q.y);
          if (!(e.x < x[r].y)) break;                                // the contig's sentinel start 0xffffffff ends the scan
          const uint32_t ze = e.x + (e.y & 0xffffu);
          const uint32_t lo = e.x > x[r].x ? e.x : x[r].x, hi = ze < x[r].y ? ze : x[r].y;
          if (hi > lo) atomicAdd(&acc[e.y >> 16], hi - lo);
          ++kk;
          if (!(kk & 1u)) q = Z2[kk >> 1];
        }
        n_ent += 2u * (kk - k[r] + 1u) + 1u;                         // (words: the entries and the grid cell)
      }
    }
    }
    if (A.mstat != nullptr) {
      const uint32_t tot = wave_total_u32(n_ent);
      if (lane == 0) {
        atomicAdd(&A.mstat[2 * (blockIdx.x & 255u)], (unsigned long long)tot);
        atomicAdd(&A.mstat[2 * (blockIdx.x & 255u) + 1], (unsigned long long)(n > 0 ? n : 0));
      }
    }
    wave_fence();
    uint32_t* __restrict__ dst = A.part + ((int64_t)c * A.n_samples + s) * T;
#ifndef GAT_MERGED_NO_NT
    // (1.2 GB of partial sums per launch on the config-4 shape, read once by k_count_merged_finish: past the L2, where the index lives)
    for (int t = lane; t < T; t += kWave) { __builtin_nontemporal_store(acc[t], &dst[t]); acc[t] = 0u; }
#else
    for (int t = lane; t < T; t += kWave) { dst[t] = acc[t]; acc[t] = 0u; }
#endif
    wave_fence();
  }
}

// k_units_overlap (round 6): what IntervalDictionary.fromIsochores' merge(0) (gat/Engine.pyx:2857-2876) unites when the units'
// lists are counted as they are (k_contig<., true> only concatenates them).  The lists of ONE unit are disjoint, so a base is counted twice only
// where a segment of one unit overlaps a segment of another -- the first reaches over the end of its workspace piece (the
// sampler's last step is filter, not intersect: gat/Engine.pyx:639-646).  One lane per candidate k_contig noted (a segment
// with a workspace boundary in its cells): is it a straddler at all (its overlap with its own unit's workspace is
// short of its length); which segments of the contig's other units does it overlap (a bisection per unit; a list k_tail trimmed
// in place has empty elements -- a probe that meets one scans the list instead); every overlap [max starts, min ends) is taken
// off the tracks' partial sums by the same walk through the merged index that added it twice.  Two straddlers that overlap each
// other both see the pair: the one of the lower unit takes it.  Overlaps of one segment that overlap EACH OTHER (a base under
// three units' segments: workspace pieces shorter than the segments) are not pairwise any more: word 1 of cand_count is set
// and the host repeats the batch through k_contig (and keeps to it for the problem).
struct UnitsOverlapArgs {
  CountArgs C;
  const UnitDev* units;
  const uint2* ws;
  const uint32_t* ws_tree;
};
__device__ __forceinline__ uint32_t unit_ws_overlap(const UnitsOverlapArgs& A, const UnitDev& U, uint32_t s, uint32_t e) {
  const uint2* __restrict__ w = A.ws + U.ws_off;
  if (U.pgrid_off >= 0) {
    const uint32_t* __restrict__ pg = A.ws_tree + U.pgrid_off;
    return ws_overlap_pgrid(w, U.n_ws, pg + kGridHeader, pg[0], pg[1], s, e);
  }
  uint32_t ov = 0;
  for (int j = 0; j < U.n_ws; ++j) {
    const uint2 x = w[j];
    if (x.x >= e) break;
    const uint32_t lo = s > x.x ? s : x.x, hi = e < x.y ? e : x.y;
    ov += hi > lo ? hi - lo : 0u;
  }
  return ov;
}
// the segments of unit entry k (of cu_rec) in sample s that overlap [xs, xe): f(segment) for each
template <typename F>
__device__ __forceinline__ void unit_overlapping(const CountArgs& A, const int s, const int k, const uint32_t xs, const uint32_t xe, F f) {
  const int4 r2 = A.cu_rec[k];
  const uint2* L = A.seg_units + (int64_t)s * A.seg_stride + r2.y;
  int nL = 0;
  bool patched = false;
  if (r2.z >= 0) {
    const int64_t sa = GAT_REC(A, s, r2.z);
    const int32_t* __restrict__ R = A.patch + sa * A.patch_stride;
    const uint2 r01 = *reinterpret_cast<const uint2*>(R);
    if ((int32_t)r01.x == 1) {
      patched = true;
      L = A.seg_units_merged + (int64_t)s * A.seg_stride + r2.y;
      nL = A.st2[sa].x;
      const uint2* __restrict__ ex = reinterpret_cast<const uint2*>(R + kPatchExtra);
      for (int j = 0; j < (int32_t)r01.y; ++j) { const uint2 y = ex[j]; if (y.x != y.y && y.x < xe && y.y > xs) f(y); }
    }
  }
  if (!patched) nL = A.unit_n[(int64_t)s * A.n_units + r2.x];
  // the first element that starts at or behind xs, by bisection (the list ascends but for what a trim emptied)
  int lo = 0, hi = nL;
  bool holes = false;
  while (lo < hi) {
    const int mid = lo + ((hi - lo) >> 1);
    const uint2 y = L[mid];
    if (y.x == y.y) { holes = true; break; }
    if (y.x < xs) lo = mid + 1; else hi = mid;
  }
  if (holes) {
    for (int j = 0; j < nL; ++j) { const uint2 y = L[j]; if (y.x != y.y && y.x < xe && y.y > xs) f(y); }
  } else {
    int j = lo - 1;
    while (j >= 0) { const uint2 y = L[j]; if (y.x != y.y) { if (y.y > xs) f(y); break; } --j; }   // the one in front may reach over xs
    for (j = lo; j < nL; ++j) {
      const uint2 y = L[j];
      if (y.x == y.y) continue;
      if (y.x >= xe) break;
      f(y);
    }
  }
}
// One lane per (candidate, OTHER unit of its contig): a lane that walked the contig's units one after the other was a chain of
// a hundred dependent reads, and the kernel as long as its chains (0.34 ms for config 3's 107 000 candidates and 230 overlaps).
__global__ __launch_bounds__(256) void k_units_overlap(UnitsOverlapArgs B, int max_units) {
  const CountArgs& A = B.C;
  // (blockIdx.y: the region; a region's workgroups beyond its candidates leave at once)
  const uint32_t item = blockIdx.x * 256u + threadIdx.x, region = blockIdx.y;
  const uint32_t total = A.cand_count[region] < A.cand_cap ? A.cand_count[region] : A.cand_cap;
  if (blockIdx.x == 0 && threadIdx.x == 0 && total != 0u) atomicAdd(&A.cand_count[kCandSlots + 2], total);
  const uint32_t idx = item / (uint32_t)max_units;
  if (idx >= total) return;
  const uint4 cd = A.cand[(size_t)region * A.cand_cap + idx];
  const int s = (int)cd.x, cu = (int)cd.y;
  const uint32_t xs = cd.z, xe = cd.w;
  const int4 rec = A.cu_rec[cu];
  const UnitDev U = B.units[rec.x];
  const int c = U.contig;
  const int k = A.contig_unit_off[c] + (int)(item - idx * (uint32_t)max_units);
  if (k >= A.contig_unit_off[c + 1] || k == cu) return;
  if (unit_ws_overlap(B, U, xs, xe) == xe - xs) return;            // inside its own workspace: overlaps nothing of another unit
  const UnitDev U2 = B.units[A.cu_rec[k].x];
  constexpr int kMaxPairs = 4;
  uint32_t is_[kMaxPairs], ie_[kMaxPairs];
  bool skip[kMaxPairs];
  int np = 0;
  bool bad = false;
  unit_overlapping(A, s, k, xs, xe, [&](const uint2 y) {
    if (np >= kMaxPairs) { bad = true; return; }
    is_[np] = y.x > xs ? y.x : xs; ie_[np] = y.y < xe ? y.y : xe;
    // both straddlers see this pair: the lower unit's takes it
    skip[np] = k < cu && unit_ws_overlap(B, U2, y.x, y.y) != y.y - y.x;
    ++np;
  });
  // an overlap that a THIRD unit's segment reaches into is not pairwise: the batch goes through the sorted lists
  for (int a = 0; a < np && !bad; ++a)
    for (int k3 = A.contig_unit_off[c]; k3 < A.contig_unit_off[c + 1] && !bad; ++k3) {
      if (k3 == cu || k3 == k) continue;
      unit_overlapping(A, s, k3, is_[a], ie_[a], [&](const uint2) { bad = true; });
    }
  if (bad) { atomicOr(&A.cand_count[kCandSlots], 1u); return; }     // (not pairwise: the host repeats the batch through the sorted lists)
  const uint2* __restrict__ Z = A.mz + A.mz_off[c];
  const uint32_t* __restrict__ F = A.mfirst + A.mf_off[c];
  const int shift = A.m_shift[c];
  const uint32_t last = (uint32_t)(A.m_cells[c] - 1);
  uint32_t* __restrict__ dst = A.part + ((int64_t)c * A.n_samples + s) * A.n_tracks;
  for (int a = 0; a < np; ++a) {
    if (skip[a]) continue;
    atomicAdd(&A.cand_count[kCandSlots + 1], 1u);
    const uint32_t g = is_[a] >> shift;
    uint32_t kk = F[g < last ? g : last];
    while (true) {
      const uint2 e = Z[kk];
      if (!(e.x < ie_[a])) break;                                      // (the contig's sentinel ends the walk)
      const uint32_t ze = e.x + (e.y & 0xffffu);
      const uint32_t lo = e.x > is_[a] ? e.x : is_[a], hi = ze < ie_[a] ? ze : ie_[a];
      if (hi > lo) atomicSub(&dst[e.y >> 16], hi - lo);
      ++kk;
    }
  }
}

// sum([...]) over the contigs in list(sample.keys()) order (gat/__init__.py:578-587) of k_count_merged's partials:
// Python ints, or left-to-right IEEE doubles of float(overlap)/len(workspace) (gat/Engine.pyx:1437-1441).  A workgroup
// takes 16 samples x 16 tracks: the partials are read with the tracks along the lanes (as they were written), the matrix
// is written with the samples along the lanes (its layout), the tile is turned in LDS.
__global__ __launch_bounds__(256) void k_count_merged_finish(CountArgs A) {
  __shared__ int64_t t_o[16][17];
  __shared__ double t_d[16][17];
  const int T = A.n_tracks;
  const int tiles_t = (T + 15) / 16;
  const int t0 = (int)(blockIdx.x % tiles_t) * 16, s0 = (int)(blockIdx.x / tiles_t) * 16;
  {
    const int ts = threadIdx.x >> 4, tt = threadIdx.x & 15;
    const int s = s0 + ts, t = t0 + tt;
    int64_t ov = 0;
    double dens = 0.0;
    if (s < A.n_samples && t < T) {
      for (int c = 0; c < A.n_contigs; ++c) {
        const uint32_t o = A.part[((int64_t)c * A.n_samples + s) * T + t];
        ov += (int64_t)o;
        const double nseg = (double)(uint32_t)A.cws_nseg[c];
        if (nseg != 0.0) dens += (double)o / nseg;
      }
    }
    t_o[tt][ts] = ov;
    t_d[tt][ts] = dens;
  }
  __syncthreads();
  const int tt = threadIdx.x >> 4, ts = threadIdx.x & 15;
  const int s = s0 + ts, t = t0 + tt;
  if (s >= A.n_samples || t >= T) return;
  const int64_t col = A.out_begin + s;
  const int k0 = A.counter_slot[0], k1 = A.counter_slot[1];
  if (k0 >= 0) A.out[((int64_t)k0 * T + t) * A.out_stride + col] = t_o[tt][ts];
  if (k1 >= 0) A.out[((int64_t)k1 * T + t) * A.out_stride + col] = __double_as_longlong(t_d[tt][ts]);
}

// Roles swapped for long sample lists (overlap is symmetric): one block per (sample, contig) builds
// the same lookup structure (starts / ends / cumulated lengths / position grid) over the SAMPLE list
// in LDS and every wave streams whole annotation tracks against it, lanes over the track's
// intervals.  With n' sample segments and m << n' annotation intervals per contig this does m
// lookups per (sample, track, contig) instead of n'.  nucleotide-overlap / nucleotide-density only.
constexpr int kSwapThreads = 1024;   // 16 waves share one indexed sample list: the lookups are latency bound

__global__ __launch_bounds__(kSwapThreads) void k_count_swap(CountArgs A) {
  extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
  const int capx = A.lds_entries, lcells = A.lds_grid;       // list capacity, log2 of the grid size
  uint32_t* xs = lds;
  uint32_t* xe = xs + capx;
  uint32_t* xcum = xe + capx;
  uint32_t* grid = xcum + capx;                               // (1 << lcells) + 1
  __shared__ uint32_t wsum[kSwapThreads / kWave];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int s = blockIdx.x, c = (int)(blockIdx.y + blockIdx.z * gridDim.y);
  if (c >= A.n_contigs) return;
  const int n = A.n_arr[(int64_t)s * A.n_stride + A.n_index[c]];
  const uint2* __restrict__ X = A.seg + (int64_t)s * A.seg_stride + A.c_off[c];
  const int64_t pbase = (((int64_t)c * 3) * A.n_tracks) * A.n_samples + s;
  if (n == 0 || n >= capx) {                                  // n >= capx cannot happen (capx = slab capacity, never filled)
    for (int t = tid; t < A.n_tracks; t += kSwapThreads) A.part[pbase + (int64_t)t * A.n_samples] = 0;
    return;
  }
  // 1. list into LDS; exclusive prefix sum of the lengths (thread owns a contiguous chunk)
  const int per = (n + kSwapThreads - 1) / kSwapThreads;
  const int i0 = tid * per, i1 = min(n, i0 + per);
  uint32_t local = 0;
  for (int i = i0; i < i1; ++i) { const uint2 v = X[i]; xs[i] = v.x; xe[i] = v.y; local += v.y - v.x; }
  uint32_t incl = wave_incl_sum_u32(local, lane);
  if (lane == 63) wsum[wave] = incl;
  __syncthreads();
  uint32_t base = incl - local;
  for (int w = 0; w < wave; ++w) base += wsum[w];
  for (int i = i0; i < i1; ++i) { xcum[i] = base; base += xe[i] - xs[i]; }
  __syncthreads();
  // 2. position grid over the starts: grid[g] = #starts < (g << shift)
  if (tid == 0) xs[n] = 0xffffffffu;                          // sentinel (n < capx: the slab never fills to capacity)
  const uint32_t maxstart = xs[n - 1];
  const int bits = maxstart ? 32 - __builtin_clz(maxstart) : 1;
  const int shift = bits > lcells ? bits - lcells : 0;
  const int cells = (int)(maxstart >> shift) + 1;
  for (int g = tid; g <= cells; g += kSwapThreads) {
    uint32_t k = (uint32_t)n;
    if (g < cells) {
      const uint32_t bound = (uint32_t)g << shift;
      int lo = 0, hi = n;
      while (lo < hi) { const int mid = lo + ((hi - lo) >> 1); if (xs[mid] < bound) lo = mid + 1; else hi = mid; }
      k = (uint32_t)lo;
    }
    grid[g] = k;
  }
  __syncthreads();
  AnnoView V;
  V.start = xs; V.end = xe; V.cumx = xcum; V.grid = grid; V.m = n; V.shift = shift; V.cells = cells;
  // 3. annotation tracks against the list
  for (int t = wave; t < A.n_tracks; t += kSwapThreads / kWave) {
    const int64_t g = A.a_off[(int64_t)t * A.n_contigs + c];
    const int m = (int)(A.a_off[(int64_t)t * A.n_contigs + c + 1] - g);
    uint32_t ov = 0;
    constexpr int kB = 4;                                      // rounds whose interval loads are in flight together
    for (int base = 0; base < m; base += kB * kWave) {
      uint32_t as[kB], ae[kB];
#pragma unroll
      for (int q = 0; q < kB; ++q) {
        const int i = base + q * kWave + lane;
        as[q] = i < m ? A.a_start[g + i] : 0u;
        ae[q] = i < m ? A.a_end[g + i] : 0u;                   // an empty interval overlaps nothing
      }
#pragma unroll
      for (int q = 0; q < kB; ++q) {
        if (base + q * kWave < m) {
          uint32_t o1, h1, m1;
          seg_vs_anno<false, true>(V, as[q], ae[q], o1, h1, m1);
          ov += (base + q * kWave + lane < m) ? o1 : 0u;
        }
      }
    }
    ov = wave_total_u32(ov);
    if (lane == 0) A.part[pbase + (int64_t)t * A.n_samples] = ov;
  }
}

// sum([...]) over the contigs in list(sample.keys()) order (gat/__init__.py:578-587): Python ints
// for the integer counters, left-to-right IEEE doubles of float(overlap)/len(workspace) for
// nucleotide-density (gat/Engine.pyx:1437-1441).  One thread per (track, sample).
__global__ __launch_bounds__(256) void k_count_finish(CountArgs A) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (int64_t)A.n_tracks * A.n_samples) return;
  const int t = (int)(i / A.n_samples), s = (int)(i - (int64_t)t * A.n_samples);
  int64_t ov = 0, hit = 0, mid = 0;
  double dens = 0.0;
  for (int c = 0; c < A.n_contigs; ++c) {
    const int64_t b = (((int64_t)c * 3) * A.n_tracks + t) * A.n_samples + s;
    const int64_t q = (int64_t)A.n_tracks * A.n_samples;
    const uint32_t o = A.part[b];
    ov += (int64_t)o;
    hit += (int64_t)A.part[b + q];
    mid += (int64_t)A.part[b + 2 * q];
    const double nseg = (double)(uint32_t)A.cws_nseg[c];
    if (nseg != 0.0) dens += (double)o / nseg;
  }
  const int64_t col = A.out_begin + s;
  const int k0 = A.counter_slot[0], k1 = A.counter_slot[1], k2 = A.counter_slot[2], k3 = A.counter_slot[3];
  if (k0 >= 0) A.out[((int64_t)k0 * A.n_tracks + t) * A.out_stride + col] = ov;
  if (k1 >= 0) A.out[((int64_t)k1 * A.n_tracks + t) * A.out_stride + col] = __double_as_longlong(dens);
  if (k2 >= 0) A.out[((int64_t)k2 * A.n_tracks + t) * A.out_stride + col] = hit;
  if (k3 >= 0) A.out[((int64_t)k3 * A.n_tracks + t) * A.out_stride + col] = mid;
}

// annotation-overlap / annotation-midoverlap with the SAMPLE list indexed in LDS (the k_count_swap scheme): one
// 256-thread block per (sample, contig) loads the list (starts with a sentinel, ends), builds a position grid over the
// starts and streams every track's intervals of that contig against it.  For an interval y: k = #starts <= y.start from
// the grid cell plus a short scan; the first segment with end > y.start is k-1 if that one reaches past y.start, else k
// (normalized list).  Partials per (contig, sample, track) go to `part`, k_count_anno_finish adds them over the contigs.
constexpr int kAnnoThreads = 256;
__global__ __launch_bounds__(kAnnoThreads) void k_count_anno_idx(CountArgs A) {
  extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
  const int capx = A.lds_entries, lcells = A.lds_grid;
  uint32_t* xs = lds;
  uint32_t* xe = xs + capx;
  uint32_t* grid = xe + capx;                                 // (1 << lcells) + 1
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int s = blockIdx.x, c = (int)(blockIdx.y + blockIdx.z * gridDim.y);
  if (c >= A.n_contigs) return;
  const int n = A.n_arr[(int64_t)s * A.n_stride + A.n_index[c]];
  const uint2* __restrict__ X = A.seg + (int64_t)s * A.seg_stride + A.c_off[c];
  const int64_t pbase = (((int64_t)c * 2) * A.n_tracks) * A.n_samples + s;
  const int64_t qstride = (int64_t)A.n_tracks * A.n_samples;
  if (n == 0 || n >= capx) {                                  // (n >= capx cannot happen: capx = capacity + 1)
    for (int t = tid; t < A.n_tracks; t += kAnnoThreads) {
      A.part[pbase + (int64_t)t * A.n_samples] = 0;
      A.part[pbase + qstride + (int64_t)t * A.n_samples] = 0;
    }
    return;
  }
  for (int i = tid; i < n; i += kAnnoThreads) { const uint2 v = X[i]; xs[i] = v.x; xe[i] = v.y; }
  if (tid == 0) xs[n] = 0xffffffffu;
  __syncthreads();
  const uint32_t maxstart = xs[n - 1];
  const int bits = maxstart ? 32 - __builtin_clz(maxstart) : 1;
  const int shift = bits > lcells ? bits - lcells : 0;
  const int cells = (int)(maxstart >> shift) + 1;
  for (int g = tid; g <= cells; g += kAnnoThreads) {
    uint32_t k = (uint32_t)n;
    if (g < cells) {
      const uint32_t bound = (uint32_t)g << shift;
      int lo = 0, hi = n;
      while (lo < hi) { const int mid = lo + ((hi - lo) >> 1); if (xs[mid] < bound) lo = mid + 1; else hi = mid; }
      k = (uint32_t)lo;
    }
    grid[g] = k;
  }
  __syncthreads();
  for (int t = wave; t < A.n_tracks; t += kAnnoThreads / kWave) {
    const int64_t g0 = A.a_off[(int64_t)t * A.n_contigs + c];
    const int m = (int)(A.a_off[(int64_t)t * A.n_contigs + c + 1] - g0);
    uint32_t hit = 0, mid = 0;
    constexpr int kB = 4;
    for (int base = 0; base < m; base += kB * kWave) {
      uint32_t ys[kB], ye[kB];
#pragma unroll
      for (int q = 0; q < kB; ++q) {
        const int i = base + q * kWave + lane;
        ys[q] = i < m ? A.a_start[g0 + i] : 0u;
        ye[q] = i < m ? A.a_end[g0 + i] : 0u;
      }
#pragma unroll
      for (int q = 0; q < kB; ++q) {
        if (base + q * kWave >= m) break;
        uint32_t gc = ys[q] >> shift;
        gc = gc < (uint32_t)(cells - 1) ? gc : (uint32_t)(cells - 1);
        int k = (int)grid[gc];
        while (xs[k] <= ys[q]) ++k;                           // k = #starts <= y.start (the sentinel ends the scan)
        const int j = (k > 0 && xe[k - 1] > ys[q]) ? k - 1 : k;   // first segment with end > y.start
        if (base + q * kWave + lane < m && j < n) {
          const uint32_t x0 = xs[j], x1 = xe[j];
          if (x0 < ye[q]) {                                   // gat/SegmentList.pyx:1127-1144, roles swapped
            hit++;
            const uint32_t mp = ys[q] + (ye[q] - ys[q]) / 2u;
            if (x0 <= mp && mp < x1) mid++;
          }
        }
      }
    }
    hit = wave_total_u32(hit);
    mid = wave_total_u32(mid);
    if (lane == 0) {
      A.part[pbase + (int64_t)t * A.n_samples] = hit;
      A.part[pbase + qstride + (int64_t)t * A.n_samples] = mid;
    }
  }
}

__global__ __launch_bounds__(256) void k_count_anno_finish(CountArgs A) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (int64_t)A.n_tracks * A.n_samples) return;
  const int t = (int)(i / A.n_samples), s = (int)(i - (int64_t)t * A.n_samples);
  const int64_t q = (int64_t)A.n_tracks * A.n_samples;
  int64_t hit = 0, mid = 0;
  for (int c = 0; c < A.n_contigs; ++c) {
    const int64_t b = (((int64_t)c * 2) * A.n_tracks + t) * A.n_samples + s;
    hit += (int64_t)A.part[b];
    mid += (int64_t)A.part[b + q];
  }
  const int64_t col = A.out_begin + s;
  const int k4 = A.counter_slot[4], k5 = A.counter_slot[5];
  if (k4 >= 0) A.out[((int64_t)k4 * A.n_tracks + t) * A.out_stride + col] = hit;
  if (k5 >= 0) A.out[((int64_t)k5 * A.n_tracks + t) * A.out_stride + col] = mid;
}

// annotation-overlap / annotation-midoverlap: roles swapped (gat/Engine.pyx:1458-1472):
// each annotation interval y is tested against the first sample segment x with x.end > y.start.
// One wave per (sample, track); lanes stride the annotation intervals, bisecting the sample list.
__global__ __launch_bounds__(256) void k_count_anno(CountArgs A) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t w = (int64_t)blockIdx.x * 4 + wave;
  if (w >= (int64_t)A.n_samples * A.n_tracks) return;
  const int s = (int)(w / A.n_tracks), t = (int)(w - (int64_t)s * A.n_tracks);
  int64_t hit_total = 0, mid_total = 0;
  for (int c = 0; c < A.n_contigs; ++c) {
    const int n = A.n_arr[(int64_t)s * A.n_stride + A.n_index[c]];
    const uint2* __restrict__ X = A.seg + (int64_t)s * A.seg_stride + A.c_off[c];
    const int64_t g = A.a_off[(int64_t)t * A.n_contigs + c];
    const int m = (int)(A.a_off[(int64_t)t * A.n_contigs + c + 1] - g);
    uint32_t hit = 0, mid = 0;
    if (n > 0) {
      for (int i = lane; i < m; i += 64) {
        const uint32_t ys = A.a_start[g + i], ye = A.a_end[g + i];
        int lo = 0, hi = n;                   // j = #x with x.end <= ys
        while (lo < hi) { const int md = lo + ((hi - lo) >> 1); if (X[md].y <= ys) lo = md + 1; else hi = md; }
        if (lo < n) {
          const uint2 x = X[lo];
          if (x.x < ye) {
            hit++;
            const uint32_t mp = ys + (ye - ys) / 2u;
            if (x.x <= mp && mp < x.y) mid++;
          }
        }
      }
    }
    hit_total += (int64_t)wave_sum_u32(hit);
    mid_total += (int64_t)wave_sum_u32(mid);
  }
  if (lane == 0) {
    const int64_t col = A.out_begin + s;
    const int k4 = A.counter_slot[4], k5 = A.counter_slot[5];
    if (k4 >= 0) A.out[((int64_t)k4 * A.n_tracks + t) * A.out_stride + col] = hit_total;
    if (k5 >= 0) A.out[((int64_t)k5 * A.n_tracks + t) * A.out_stride + col] = mid_total;
  }
}

}  // namespace gat
