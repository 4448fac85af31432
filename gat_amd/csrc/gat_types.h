// gat_types.h -- the records and constants the host side of libgat_mi355.so shares with the kernels: what a host
// translation unit that launches nothing (gat_prep.hip: problem creation, annotation tables) needs to know, without the
// kernels themselves.  Included by gat_device.h / gat_kernels.h / gat_tail.h, which own everything else.
#pragma once
#include <stdint.h>

namespace gat {

constexpr int kWave = 64;
constexpr int kMtN = 624;

constexpr int kWsTreeMin = 32;      // workspaces with more segments get trees (the shorter ones are searched in registers)
constexpr int kWsTreeLevels = 6;   // 16^6 keys; the level loops are unrolled so that the geometry stays in scalar registers

constexpr int kPlaceWsLds = 256;      // k_place: workspace segments kept in LDS (16 B each)
constexpr int kPlaceGridTiles = 8;    // k_place_grid (MODE 5): tiles (waves) of a workgroup around one unit's cdf grid in LDS: two waves per SIMD, each with
                                      // the registers for its look-ups one chunk ahead (rows pinned at v192..v223)
constexpr int kGridHeader = 4;        // words in front of a grid in ws_tree: {shift, cells, widest cell's span, words of the LDS image}
constexpr int kPlaceRankLds = 1024;   // k_place: length-rank table entries kept in LDS
#ifndef GAT_PLACE_WIDE_TILES
#define GAT_PLACE_WIDE_TILES 4         // (tuning builds: -DGAT_PLACE_WIDE_TILES=2 / 8; tools/exp_wide_tiles*.sh: config-4 shape k_place_wide
                                       //  10.7 ms with 8, 9.2 with 4, 11.4 with 2)
#endif
constexpr int kPlaceWide = GAT_PLACE_WIDE_TILES;   // k_place_wide: tiles (waves) of a workgroup, sharing one unit's rank table in LDS
constexpr int kPlaceWideMaxRank = 20480;  // ... whose size is bounded by the LDS beside the eight 8 KB rings (80 KB of 160)
constexpr int kTailMaxWs = 64;        // workspace segments k_tail scans linearly (wave-uniform loop)
constexpr int kMergedSlots = 8;       // k_count_merged: contigs are dealt to this many slots (blockIdx % 8: the XCD stride)

// per isochore unit, everything SamplerAnnotator.sample derives from (segments, workspace) before
// its loop (gat/Engine.pyx:543-565), hoisted to problem creation.
struct UnitDev {
  int32_t n_ws;         // workspace segments of the unit
  int32_t ws_off;       // offset into ws / ws_cdf
  int32_t tree_start_off;  // offset into ws_tree of the search tree over the workspace starts (-1: short workspace)
  int32_t tree_cdf_off;    // ... over the cumulated lengths
  uint32_t hist_total;  // HistogramSampler.total_size == number of working segments
  uint32_t bucket;      // bucket size (after the bucket_size==0 rule, gat/SegmentList.pyx:1164)
  uint32_t ws_total;    // SegmentListSampler.total_size == workspace bases
  int32_t ltotal;       // bases to reproduce (gat/Engine.pyx:550-552)
  int32_t slab_off;     // offset of the unit's output region inside a sample's slab
  int32_t slab_cap;     // capacity of that region == LDS buffer capacity used for the unit
  int32_t contig;
  int32_t rank_off;     // offset into rank_len: rank_len[r] = bucket index searchsorted(cdf, r) returns
  int32_t n_target;     // SamplerSegments: len(segments) placements (gat/Engine.pyx:726)
  int32_t pad;          // units_o: the unit id
  // fragmented workspaces (round 6; both -1 for a short workspace), offsets into ws_tree, kGridHeader words in front of each:
  int32_t pgrid_off;    // position grid: entry c = the first workspace segment whose END lies beyond c << shift (u32 entries;
                        //   cells + 1 of them, the last = n_ws): the overlap of [s, e) with the workspace is a walk from entry s >> shift
  int32_t cgrid_off;    // grid over the cumulated lengths (the position draw's searchsorted, gat/Engine.pyx:299-305): u16 entries
                        //   g[c] = #{i : cdf[i] < c << shift} (cells + 1, padded to a word), then u16 keys cdf[i] & ((1 << shift) - 1):
                        //   #{cdf < p} = g[c] + #{i in [g[c], g[c + 1]) : key[i] < (p & mask)} for c = p >> shift -- the image k_place_grid
                        //   copies into LDS (2 bytes per workspace segment + 2 per cell)
};
static_assert(sizeof(UnitDev) == 64, "UnitDev: sixteen words");

enum : int32_t {
  kStatusOverflow = 1,   // LDS/slab capacity exceeded: host retries with a larger slab
  kStatusAssert = 2,     // reference assert would fire (gat/Engine.pyx:645 sum()>0)
  kStatusTrimAssert = 4, // gat/SegmentList.pyx:560 sum() > size
  kStatusUnitsOverlap = 16, // k_units_overlap: overlaps between the units' lists that are not pairwise (or more candidates than the
                         // buffer holds): host repeats the batch through k_contig and keeps to it for the problem
  kStatusContigLds = 8,  // k_contig: a contig's lists exceed the LDS the launch was given (sized for what is expected,
                         // not for every unit at its capacity): host repeats the batch with the full size
};

}  // namespace gat
