// gat_mi355.hip -- host side of libgat_mi355.so (C ABI in include/gat_mi355.h).
//
// Host work is limited to what the reference does once per (segments, workspace) pair before
// sampling starts (gat/Engine.pyx:543-565, hoisted out of the per-sample loop), buffer management
// and kernel launches.  Sampling, fromIsochores and counting run only as HIP kernels
// (gat_kernels.h); there is no CPU path for them in this library.
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <rccl/rccl.h>          // types only: the library is loaded with dlopen at the first collective

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <climits>
#include <string>
#include <vector>
#include <thread>
#include <chrono>
#include <atomic>

#include "../../include/gat_mi355.h"
#define GAT_NUM_COUNTERS_DEV 6
#include "gat_kernels.h"
#include "gat_tail.h"
#include "gat_stats.h"

using gat::UnitDev;

static thread_local std::string g_last_error;

struct gat_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  hipEvent_t ev[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
  hipEvent_t ev_main[2] = {nullptr, nullptr};   // around the dominant count kernel alone (k_count_seg / k_count_swap)
  bool main_recorded = false;
  int count_kernel = 0;                         // GAT_COUNT_KERNEL_* of the last launch_count
  hipEvent_t ev_k[4] = {nullptr, nullptr, nullptr, nullptr};   // behind k_rng, k_place, k_merge_big, k_sampler
  bool k_recorded = false;
  hipEvent_t ev_t[2] = {nullptr, nullptr};      // split path: behind k_tail, k_finalize
  bool t_recorded = false;
  hipEvent_t ev_cnt[2] = {nullptr, nullptr};    // around the count phase
  // status word and statistics of a sampler batch, copied behind its kernels and read after the batch's ONE synchronisation
  int32_t* h_flags = nullptr;                   // pinned
  unsigned long long* h_stat = nullptr;         // pinned, 8 words
  std::string err;
  int max_lds = 65536;
};

static int set_err(gat_ctx* ctx, int code, const char* fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_last_error = buf;
  if (ctx) ctx->err = buf;
  return code;
}

#define HIPCHK(ctx, call)                                                                          \
  do {                                                                                             \
    hipError_t e__ = (call);                                                                       \
    if (e__ != hipSuccess)                                                                         \
      return set_err(ctx, GAT_ERR_DEVICE, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e__), \
                     __FILE__, __LINE__);                                                          \
  } while (0)

template <typename T>
struct DevBuf {
  T* p = nullptr;
  size_t n = 0;
  ~DevBuf() { release(); }
  void release() { if (p) { (void)hipFree(p); p = nullptr; n = 0; } }
  hipError_t alloc(size_t count) {
    release();
    if (count == 0) count = 1;
    hipError_t e = hipMalloc((void**)&p, count * sizeof(T));
    if (e == hipSuccess) n = count;
    return e;
  }
  hipError_t upload(const std::vector<T>& h, hipStream_t s) {
    hipError_t e = alloc(h.size());
    if (e != hipSuccess) return e;
    if (!h.empty()) e = hipMemcpyAsync(p, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    return e;
  }
};

// annotations on the device: SoA starts / ends / exclusive cumulated lengths + CSR offsets
struct AnnoDev {
  DevBuf<uint32_t> start, end, cumx, grid;
  DevBuf<int64_t> off, goff;
  DevBuf<int32_t> shift, cells;
  std::vector<int64_t> h_off;
  // merged multi-track index (k_count_merged), built for problems with several tracks
  DevBuf<uint2> mz;
  DevBuf<uint32_t> mfirst;
  DevBuf<int64_t> mz_off, mf_off;
  DevBuf<int32_t> m_shift, m_cells, m_slot_off, m_slot_contigs;
  int max_slot_contigs = 0;
  bool has_merged = false;
  int64_t merged_entries = 0;
  int64_t max_m = 0;
  int64_t max_cells = 0;
  int64_t total = 0;
};

static int check_list(gat_ctx* ctx, const gat_segment* s, int64_t n, const char* what, int64_t idx) {
  for (int64_t i = 0; i < n; ++i) {
    if (s[i].start >= s[i].end)
      return set_err(ctx, GAT_ERR_ASSERT, "%s list %lld is not normalized: empty/invalid segment %u-%u", what,
                     (long long)idx, s[i].start, s[i].end);
    if (s[i].end >= 0x80000000u)
      return set_err(ctx, GAT_ERR_ARG, "%s list %lld: coordinate %u >= 2^31 not supported", what, (long long)idx, s[i].end);
    if (i > 0 && s[i - 1].end > s[i].start)
      return set_err(ctx, GAT_ERR_ASSERT, "%s list %lld is not normalized: %u-%u overlaps/precedes %u-%u", what,
                     (long long)idx, s[i - 1].start, s[i - 1].end, s[i].start, s[i].end);
  }
  return GAT_OK;
}

// host threads for the per-list / per-contig preparation of gat_problem_create (GAT_HOST_THREADS, default min(16, cores))
template <typename F>
static void parallel_for(int64_t n, F body) {
  const char* env_t = getenv("GAT_HOST_THREADS");
  unsigned nthreads = env_t ? (unsigned)std::max(1, atoi(env_t)) : std::min(16u, std::max(1u, std::thread::hardware_concurrency()));
  nthreads = (unsigned)std::min<int64_t>(nthreads, std::max<int64_t>(1, n));
  std::atomic<int64_t> next(0);
  auto worker = [&]() { for (int64_t i = next.fetch_add(1); i < n; i = next.fetch_add(1)) body(i); };
  std::vector<std::thread> pool;
  for (unsigned t = 1; t < nthreads; ++t) pool.emplace_back(worker);
  worker();
  for (auto& th : pool) th.join();
}

// The merged index of k_count_merged: per group (contig) the intervals of ALL tracks in one list sorted by start, 8-byte
// entries {start, length:16 | track:16}.  Intervals longer than `bound` (a power of two, see below, at most 32 768) are
// cut into pieces -- an overlap sum does not change -- so that no entry keeps a scan alive over more
// than `bound` bases; first[g] is the first entry with end > g << shift or start >= g << shift, i.e. where a scan for a
// segment starting in cell g begins.
static int build_merged(gat_ctx* ctx, AnnoDev& A, const gat_segment* annos, const int64_t* anno_off, int64_t n_tracks,
                        int32_t n_groups) {
  if (n_tracks > 65535) return GAT_OK;                              // (track ids are 16 bits: such problems keep the per-track kernel)
  std::vector<int64_t> hz_off((size_t)n_groups + 1, 0), hf_off((size_t)n_groups + 1, 0);
  std::vector<int32_t> h_shift((size_t)n_groups, 0), h_cells((size_t)n_groups, 1);
  struct Ent { uint32_t s, e, t; };
  // a contig's index is built by itself (collect, sort, grid): the contigs are dealt to host threads -- sorting 10^7 entries
  // on one core made gat_problem_create 0.9 s on the config-4 shape
  std::vector<std::vector<uint2>> cz((size_t)n_groups);
  std::vector<std::vector<uint32_t>> cf((size_t)n_groups);
  std::vector<int> c_err((size_t)n_groups, 0);
  const char* env_bf = getenv("GAT_MERGED_BOUND");
  const uint64_t bfac = env_bf ? (uint64_t)std::max(1, atoi(env_bf)) : 2;
  auto build_one = [&](int c) {
    std::vector<Ent> e;
    std::vector<uint2>& hz = cz[(size_t)c];
    std::vector<uint32_t>& hf = cf[(size_t)c];
    uint64_t total_len = 0, cnt = 0;
    for (int64_t t = 0; t < n_tracks; ++t) {
      const int64_t l = t * n_groups + c;
      for (int64_t i = anno_off[l]; i < anno_off[l + 1]; ++i) { total_len += annos[i].end - annos[i].start; ++cnt; }
    }
    // the piece bound: a scan starts at the first entry that reaches into the segment's cell and passes everything up to
    // the segment's end, so it walks over about (bound + segment length) / spacing entries, most of which ended before the
    // segment began.  Twice the mean interval length or twice the mean spacing of the entries, whichever is larger (cutting
    // finer than the spacing only adds entries): config-4 shape, 1 000 tracks, one entry per 300 bases: 30.8 -> 25.9 ms per
    // 4 096 samples against the earlier 8 x mean length; config 3 (one per 3 000) keeps its bound.
    uint64_t span = 0;
    for (int64_t t = 0; t < n_tracks; ++t) {
      const int64_t l = t * n_groups + c;
      if (anno_off[l + 1] > anno_off[l]) span = std::max<uint64_t>(span, annos[anno_off[l + 1] - 1].end);
    }
    const uint64_t want = cnt > 0 ? std::max(bfac * (total_len / cnt), 2 * (span / cnt)) : 0;
    uint32_t bound = 256;
    while ((uint64_t)bound < want && bound < 32768u) bound <<= 1;
    e.reserve((size_t)cnt + (size_t)cnt / 4);
    for (int64_t t = 0; t < n_tracks; ++t) {
      const int64_t l = t * n_groups + c;
      for (int64_t i = anno_off[l]; i < anno_off[l + 1]; ++i) {
        uint32_t s0 = annos[i].start;
        const uint32_t e0 = annos[i].end;
        while (e0 - s0 > bound) { e.push_back(Ent{s0, s0 + bound, (uint32_t)t}); s0 += bound; }
        e.push_back(Ent{s0, e0, (uint32_t)t});
      }
    }
    std::sort(e.begin(), e.end(), [](const Ent& a, const Ent& b) { return a.s != b.s ? a.s < b.s : a.t < b.t; });
    const size_t ne = e.size();
    if (ne >= 0xfffffff0ull) { c_err[(size_t)c] = 1; return; }
    const uint32_t max_start = ne ? e[ne - 1].s : 0u;
    int64_t target = 64;
    while (target < (int64_t)(ne / 2)) target <<= 1;                // about two entries per cell
    int sh = 0;
    while (((int64_t)max_start >> sh) + 1 > target) ++sh;
    const int64_t cells = ((int64_t)max_start >> sh) + 1;
    h_shift[(size_t)c] = sh;
    h_cells[(size_t)c] = (int32_t)cells;
    hf.resize((size_t)cells);
    {
      size_t k = 0;                                                 // first entry starting at or behind the cell's start
      for (int64_t g = 0; g < cells; ++g) {
        const uint64_t cs = (uint64_t)g << sh;
        while (k < ne && (uint64_t)e[k].s < cs) ++k;
        hf[(size_t)g] = (uint32_t)k;
      }
      for (size_t i = 0; i < ne; ++i) {                             // ... or an earlier one that reaches past it
        for (int64_t g = ((int64_t)e[i].s >> sh) + 1; g < cells && ((uint64_t)g << sh) < (uint64_t)e[i].e; ++g)
          if ((uint32_t)i < hf[(size_t)g]) hf[(size_t)g] = (uint32_t)i;
      }
    }
    hz.reserve(ne + 3);
    for (size_t i = 0; i < ne; ++i) hz.push_back(make_uint2(e[i].s, ((e[i].t & 0xffffu) << 16) | ((e[i].e - e[i].s) & 0xffffu)));
    hz.push_back(make_uint2(0xffffffffu, 0u));                      // ends every scan
    hz.push_back(make_uint2(0xffffffffu, 0u));                      // (entries are read in pairs)
    if (hz.size() & 1) hz.push_back(make_uint2(0xffffffffu, 0u));   // ... and the next contig starts at an even index
  };
  parallel_for(n_groups, [&](int64_t c) { build_one((int)c); });
  std::vector<uint2> hz;
  std::vector<uint32_t> hf;
  for (int c = 0; c < n_groups; ++c) {
    if (c_err[(size_t)c]) return set_err(ctx, GAT_ERR_CAPACITY, "group %d: more than 2^32 annotation intervals", c);
    hz.insert(hz.end(), cz[(size_t)c].begin(), cz[(size_t)c].end());
    hf.insert(hf.end(), cf[(size_t)c].begin(), cf[(size_t)c].end());
    hz_off[(size_t)c + 1] = (int64_t)hz.size();
    hf_off[(size_t)c + 1] = (int64_t)hf.size();
    std::vector<uint2>().swap(cz[(size_t)c]);
    std::vector<uint32_t>().swap(cf[(size_t)c]);
  }
  HIPCHK(ctx, A.mz.upload(hz, ctx->stream));
  HIPCHK(ctx, A.mfirst.upload(hf, ctx->stream));
  HIPCHK(ctx, A.mz_off.upload(hz_off, ctx->stream));
  HIPCHK(ctx, A.mf_off.upload(hf_off, ctx->stream));
  HIPCHK(ctx, A.m_shift.upload(h_shift, ctx->stream));
  HIPCHK(ctx, A.m_cells.upload(h_cells, ctx->stream));
  {
    // the groups dealt to the eight XCD slots of k_count_merged: largest first, each to the slot with the least so far
    std::vector<std::pair<int64_t, int>> w;
    for (int c = 0; c < n_groups; ++c) w.push_back(std::make_pair(hz_off[(size_t)c + 1] - hz_off[(size_t)c], c));
    std::sort(w.begin(), w.end(), [](const std::pair<int64_t, int>& a, const std::pair<int64_t, int>& b) {
      return a.first != b.first ? a.first > b.first : a.second < b.second; });
    std::vector<std::vector<int32_t>> slots((size_t)gat::kMergedSlots);
    std::vector<int64_t> load((size_t)gat::kMergedSlots, 0);
    for (auto& x : w) {
      size_t best = 0;
      for (size_t k = 1; k < load.size(); ++k) if (load[k] < load[best]) best = k;
      slots[best].push_back(x.second);
      load[best] += x.first;
    }
    std::vector<int32_t> so((size_t)gat::kMergedSlots + 1, 0), sc;
    A.max_slot_contigs = 0;
    for (size_t k = 0; k < slots.size(); ++k) {
      for (int32_t c : slots[k]) sc.push_back(c);
      so[k + 1] = (int32_t)sc.size();
      A.max_slot_contigs = std::max<int>(A.max_slot_contigs, (int)slots[k].size());
    }
    if (sc.empty()) sc.push_back(0);
    HIPCHK(ctx, A.m_slot_off.upload(so, ctx->stream));
    HIPCHK(ctx, A.m_slot_contigs.upload(sc, ctx->stream));
  }
  A.has_merged = true;
  A.merged_entries = (int64_t)hz.size();
  return GAT_OK;
}

static int build_annos(gat_ctx* ctx, AnnoDev& A, const gat_segment* annos, const int64_t* anno_off, int64_t n_lists,
                       int32_t n_groups) {
  const int64_t total = anno_off[n_lists];
  std::vector<uint32_t> hs((size_t)total), he((size_t)total), hc((size_t)total);
  A.h_off.assign(anno_off, anno_off + n_lists + 1);
  A.max_m = 0;
  A.total = total;
  for (int64_t l = 0; l < n_lists; ++l) {
    const int64_t o = anno_off[l], m = anno_off[l + 1] - o;
    int rc = check_list(ctx, annos + o, m, "annotation", l);
    if (rc) return rc;
    A.max_m = std::max(A.max_m, m);
  }
  {
    constexpr int64_t kBlock = 256;                                 // lists per task
    parallel_for((n_lists + kBlock - 1) / kBlock, [&](int64_t b) {
      for (int64_t l = b * kBlock; l < std::min(n_lists, (b + 1) * kBlock); ++l) {
        const int64_t o = anno_off[l], m = anno_off[l + 1] - o;
        uint32_t cum = 0;
        for (int64_t i = 0; i < m; ++i) {
          hs[(size_t)(o + i)] = annos[o + i].start;
          he[(size_t)(o + i)] = annos[o + i].end;
          hc[(size_t)(o + i)] = cum;
          cum += annos[o + i].end - annos[o + i].start;
        }
      }
    });
  }
  // per group (contig) a uniform grid over the start coordinates, about one start per cell:
  // grid[g] = #starts < (g << shift); the count kernels look a position up instead of bisecting
  const int64_t n_tracks = n_groups > 0 ? n_lists / n_groups : 0;
  std::vector<int32_t> h_shift((size_t)std::max(1, n_groups), 0), h_cells((size_t)std::max(1, n_groups), 1);
  std::vector<int64_t> h_goff((size_t)n_lists + 1, 0);
  A.max_cells = 1;
  for (int c = 0; c < n_groups; ++c) {
    uint32_t max_start = 0;
    int64_t mc = 0;
    for (int64_t t = 0; t < n_tracks; ++t) {
      const int64_t l = t * n_groups + c, o = anno_off[l], m = anno_off[l + 1] - o;
      mc = std::max(mc, m);
      if (m > 0) max_start = std::max(max_start, annos[o + m - 1].start);
    }
    int64_t target = 16;
    while (target < mc) target <<= 1;
    { const char* env_g = getenv("GAT_GRID_FACTOR"); target *= env_g ? atoi(env_g) : 2; }   // about one start per two cells (measured best of 1, 2, 4, 8)
    int sh = 0;
    while (((int64_t)max_start >> sh) + 1 > target) ++sh;
    h_shift[(size_t)c] = sh;
    h_cells[(size_t)c] = (int32_t)(((int64_t)max_start >> sh) + 1);
    A.max_cells = std::max<int64_t>(A.max_cells, h_cells[(size_t)c]);
  }
  for (int64_t l = 0; l < n_lists; ++l) h_goff[(size_t)l + 1] = h_goff[(size_t)l] + h_cells[(size_t)(l % n_groups)] + 1;
  std::vector<uint32_t> hg((size_t)h_goff[(size_t)n_lists]);
  {
    constexpr int64_t kBlock = 256;
    parallel_for((n_lists + kBlock - 1) / kBlock, [&](int64_t b) {
      for (int64_t l = b * kBlock; l < std::min(n_lists, (b + 1) * kBlock); ++l) {
        const int c = (int)(l % n_groups);
        const int64_t o = anno_off[l], m = anno_off[l + 1] - o;
        const int sh = h_shift[(size_t)c], cells = h_cells[(size_t)c];
        uint32_t* g = hg.data() + h_goff[(size_t)l];
        int64_t k = 0;
        for (int cell = 0; cell <= cells; ++cell) {
          const uint64_t bound = (uint64_t)cell << sh;
          while (k < m && (uint64_t)annos[o + k].start < bound) ++k;
          g[cell] = (uint32_t)(cell == cells ? m : k);
        }
      }
    });
  }
  HIPCHK(ctx, A.grid.upload(hg, ctx->stream));
  HIPCHK(ctx, A.goff.upload(h_goff, ctx->stream));
  HIPCHK(ctx, A.shift.upload(h_shift, ctx->stream));
  HIPCHK(ctx, A.cells.upload(h_cells, ctx->stream));
  HIPCHK(ctx, A.start.upload(hs, ctx->stream));
  HIPCHK(ctx, A.end.upload(he, ctx->stream));
  HIPCHK(ctx, A.cumx.upload(hc, ctx->stream));
  HIPCHK(ctx, A.off.upload(A.h_off, ctx->stream));
  // the merged index: from four tracks up -- and for fewer when their lists do not fit the LDS tile of k_count_seg (it
  // would read them from global memory with four look-ups per sample segment; the index needs two: config-5 shape,
  // one track of a million intervals, count 2.86 -> 1.74 ms per 16 384 samples)
  const char* env_mm = getenv("GAT_MERGED_MIN_TRACKS");
  const char* env_e = getenv("GAT_COUNT_LDS_ENTRIES");
  const bool unstaged = A.max_m + 1 > (env_e ? atoi(env_e) : 1024) && !env_mm;
  if (n_groups > 0 && (n_tracks >= (env_mm ? atoi(env_mm) : 4) || unstaged)) {
    const auto t0 = std::chrono::steady_clock::now();
    int rc = build_merged(ctx, A, annos, anno_off, n_tracks, n_groups);
    if (rc) return rc;
    if (getenv("GAT_TIME_CREATE"))
      fprintf(stderr, "[gat] build_merged %.1f ms\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
  }
  return GAT_OK;
}

struct gat_problem {
  gat_ctx* ctx = nullptr;
  int32_t n_units = 0, n_contigs = 0, n_tracks = 0, merge_contigs = 0, sampler = 0;
  std::vector<UnitDev> h_units;
  std::vector<int32_t> h_order;          // active units, largest first
  std::vector<int32_t> h_base_cap;       // per unit capacity before scaling
  std::vector<int32_t> h_contig_unit_off, h_contig_units, h_contig_slab_off, h_count_c_off, h_count_n_index;
  std::vector<int64_t> h_cws_nseg;
  int cap_scale = 1;
  int64_t slab_stride = 0;
  int32_t max_unit_cap = 0, max_contig_cap = 0;
  // k_contig: contigs by expected list length (largest first), size classes of that order, LDS sized for the expectation
  std::vector<int32_t> h_contig_order, h_contig_need, h_contig_class_start;
  DevBuf<int32_t> d_contig_order;
  bool contig_tight = true;              // false after a batch whose lists did not fit: LDS for every unit at capacity
  int64_t n_seg_total = 0;               // input segments (for the algorithmic byte count)
  DevBuf<UnitDev> d_units;
  DevBuf<UnitDev> d_units_o;            // the active units' records in launch order (h_order), unit id in `pad`
  DevBuf<int32_t> d_order, d_contig_unit_off, d_contig_units, d_contig_slab_off, d_count_c_off, d_count_n_index;
  DevBuf<uint2> d_ws;
  DevBuf<uint32_t> d_ws_cdf, d_rank_len;
  DevBuf<uint32_t> d_ws_tree;            // 16-ary search trees over the starts and the cumulated lengths of long workspaces
  DevBuf<int64_t> d_cws_nseg;
  AnnoDev annos;
  // per-batch scratch
  int64_t batch = 0;
  DevBuf<uint2> d_slab, d_cslab;
  DevBuf<int32_t> d_unit_n, d_contig_n, d_flags;
  DevBuf<unsigned long long> d_stat;
  // lane-parallel front end (k_rng + k_place)
  std::vector<int32_t> h_rng_rows;       // per active index: raw outputs generated per stream
  std::vector<int64_t> h_rng_off;
  int64_t rng_rows_total = 0;            // sum of h_rng_rows
  DevBuf<int32_t> d_rng_rows;
  DevBuf<int4> d_st;
  DevBuf<int4> d_st2;                    // k_merge_big -> k_sampler hand-off (first consolidation of the long lists)
#ifdef GAT_DIAG
  DevBuf<unsigned long long> d_diag;     // diagnostic build: per work unit, cycles per phase of k_sampler
#endif
  DevBuf<int64_t> d_rng_off;
  DevBuf<uint32_t> d_rng_out, d_ws_stat, d_part;
  DevBuf<uint2> d_fslab;                 // split path: the units' final lists (k_finalize writes out of place)
  DevBuf<uint32_t> d_cum;                // split path: running lengths of the merged lists (parallel to the slab)
  DevBuf<gat::TailPatch> d_patch;        // ... and k_tail's record per work unit
  DevBuf<uint32_t> d_todo, d_todo_count; // ... and the units it leaves to k_sampler
  DevBuf<uint32_t> d_serial;             // gat_sample_and_count_serial: the MT19937 state (and its copy at the batch's start)
  DevBuf<int32_t> d_unit_pos;            // unit id -> launch position (k_contig reads k_tail's records by it)
  bool patched_contigs = false;          // the last batch skipped k_finalize: k_contig took (merged list, record)
  bool patched_counts = false;           // ... k_count_seg takes (merged list, record)
  std::vector<int32_t> h_class_start;    // launch positions where a size class begins (+ the end): one launch per class
  bool split_path = false;               // k_consolidate + k_tail + k_finalize in front of k_sampler
  bool split_ran = false;                // ... and the last sampler batch took it: the units' lists are in d_fslab
  const uint2* final_slab() const { return split_ran ? d_fslab.p : d_slab.p; }
  int sampler_mode = 1;                  // 1: k_rng + k_place + k_sampler(resume); 0: k_sampler alone
  uint32_t max_hist = 0;                 // longest length-rank table of an active unit
  bool long_lists = false;               // units beyond the wave's bucket sorts (k_merge_big, k_tail_big)
  bool all_simple = false;               // every active unit: one workspace segment (> 1 base), bucket 1, rank table in LDS
  int32_t max_nws = 0;                   // longest workspace among the active units (selects the kernel variants)
  bool small_tables = false;             // every active unit: <= 64 workspace segments, < 256 working segments
  int swap_capx = 0;                     // > 0: count with k_count_swap, sample lists of up to this many segments in LDS
};

// ------------------------------------------------------------------------------------------
extern "C" const char* gat_version(void) { return "gat_mi355 0.1 (gfx950)"; }

extern "C" const char* gat_last_error(const gat_ctx* ctx) { return ctx ? ctx->err.c_str() : g_last_error.c_str(); }

extern "C" int gat_ctx_create(gat_ctx** out, int device_id, void* stream) {
  if (!out) return set_err(nullptr, GAT_ERR_ARG, "gat_ctx_create: out is NULL");
  *out = nullptr;
  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev <= 0)
    return set_err(nullptr, GAT_ERR_DEVICE, "no HIP device available (%s): this library has no CPU path",
                   e == hipSuccess ? "device count 0" : hipGetErrorString(e));
  if (device_id < 0 || device_id >= ndev) return set_err(nullptr, GAT_ERR_ARG, "device %d out of range (0..%d)", device_id, ndev - 1);
  gat_ctx* ctx = new gat_ctx();
  ctx->device = device_id;
  HIPCHK(ctx, hipSetDevice(device_id));
  hipDeviceProp_t prop;
  HIPCHK(ctx, hipGetDeviceProperties(&prop, device_id));
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
    int rc = set_err(nullptr, GAT_ERR_DEVICE, "device %d is %s; this library carries gfx950 code only", device_id, prop.gcnArchName);
    delete ctx;
    return rc;
  }
  ctx->max_lds = (int)prop.sharedMemPerBlock;
  if (ctx->max_lds < 160 * 1024) ctx->max_lds = 160 * 1024;   // gfx950: 160 KiB per workgroup via the opt-in attribute
  if (stream) {
    ctx->stream = (hipStream_t)stream;
  } else {
    HIPCHK(ctx, hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));
    ctx->own_stream = true;
  }
  for (auto& ev : ctx->ev) HIPCHK(ctx, hipEventCreate(&ev));
  for (auto& ev : ctx->ev_main) HIPCHK(ctx, hipEventCreate(&ev));
  for (auto& ev : ctx->ev_k) HIPCHK(ctx, hipEventCreate(&ev));
  for (auto& ev : ctx->ev_t) HIPCHK(ctx, hipEventCreate(&ev));
  for (auto& ev : ctx->ev_cnt) HIPCHK(ctx, hipEventCreate(&ev));
  HIPCHK(ctx, hipHostMalloc((void**)&ctx->h_flags, 64, hipHostMallocDefault));
  HIPCHK(ctx, hipHostMalloc((void**)&ctx->h_stat, 64, hipHostMallocDefault));
  *out = ctx;
  return GAT_OK;
}

extern "C" void gat_ctx_destroy(gat_ctx* ctx) {
  if (!ctx) return;
  (void)hipSetDevice(ctx->device);
  for (auto& ev : ctx->ev) if (ev) (void)hipEventDestroy(ev);
  for (auto& ev : ctx->ev_main) if (ev) (void)hipEventDestroy(ev);
  for (auto& ev : ctx->ev_k) if (ev) (void)hipEventDestroy(ev);
  for (auto& ev : ctx->ev_t) if (ev) (void)hipEventDestroy(ev);
  for (auto& ev : ctx->ev_cnt) if (ev) (void)hipEventDestroy(ev);
  if (ctx->h_flags) (void)hipHostFree(ctx->h_flags);
  if (ctx->h_stat) (void)hipHostFree(ctx->h_stat);
  if (ctx->own_stream && ctx->stream) (void)hipStreamDestroy(ctx->stream);
  delete ctx;
}

extern "C" int gat_ctx_synchronize(gat_ctx* ctx) {
  if (!ctx) return set_err(nullptr, GAT_ERR_ARG, "ctx is NULL");
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  return GAT_OK;
}

extern "C" int gat_dev_alloc(gat_ctx* ctx, void** out, size_t bytes) {
  if (!ctx || !out) return set_err(ctx, GAT_ERR_ARG, "gat_dev_alloc: NULL argument");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  HIPCHK(ctx, hipMalloc(out, bytes ? bytes : 1));
  return GAT_OK;
}
extern "C" int gat_dev_free(gat_ctx* ctx, void* p) {
  if (!ctx) return set_err(ctx, GAT_ERR_ARG, "gat_dev_free: NULL ctx");
  HIPCHK(ctx, hipFree(p));
  return GAT_OK;
}
extern "C" int gat_memcpy_d2h(gat_ctx* ctx, void* dst, const void* src, size_t bytes) {
  if (!ctx) return set_err(ctx, GAT_ERR_ARG, "NULL ctx");
  HIPCHK(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  return GAT_OK;
}
extern "C" int gat_memcpy_h2d(gat_ctx* ctx, void* dst, const void* src, size_t bytes) {
  if (!ctx) return set_err(ctx, GAT_ERR_ARG, "NULL ctx");
  HIPCHK(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  return GAT_OK;
}

// ------------------------------------------------------------------------------------------
// host-side hoisted setup of one unit (gat/Engine.pyx:543-565)
static uint32_t host_overlap(const gat_segment* w, int64_t nw, uint32_t s, uint32_t e) {
  // bases of [s,e) inside the normalized list w
  uint32_t ov = 0;
  const gat_segment* it = std::lower_bound(w, w + nw, s, [](const gat_segment& a, uint32_t v) { return a.end <= v; });
  for (; it != w + nw && it->start < e; ++it) ov += std::min(e, it->end) - std::max(s, it->start);
  return ov;
}

static int32_t cap_for(int64_t n) {
  int64_t c = n + n / 4 + 96;
  if (getenv("GAT_TEST_SMALL_CAPS")) c = n / 2 + 8;      // tests: force the overflow / retry path
  c = (c + 63) / 64 * 64;
  return (int32_t)c;
}

static int layout_slab(gat_problem* P) {
  // regions in contig-major order so that a contig's units are adjacent (k_contig output region)
  int64_t off = 0;
  P->max_unit_cap = 0;
  P->max_contig_cap = 0;
  for (int c = 0; c < P->n_contigs; ++c) {
    P->h_contig_slab_off[c] = (int32_t)off;
    int64_t ccap = 0;
    for (int ui = P->h_contig_unit_off[c]; ui < P->h_contig_unit_off[c + 1]; ++ui) {
      const int u = P->h_contig_units[ui];
      UnitDev& U = P->h_units[u];
      const int64_t cap = (int64_t)P->h_base_cap[u] * P->cap_scale;
      U.slab_off = (int32_t)off;
      U.slab_cap = (int32_t)cap;
      off += cap;
      ccap += cap;
      P->max_unit_cap = std::max<int32_t>(P->max_unit_cap, (int32_t)cap);
    }
    P->max_contig_cap = std::max<int32_t>(P->max_contig_cap, (int32_t)ccap);
    if (!P->merge_contigs) {
      const int u = P->h_contig_units[P->h_contig_unit_off[c]];
      P->h_count_c_off[c] = P->h_units[u].slab_off;
      P->h_count_n_index[c] = u;
    } else {
      P->h_count_c_off[c] = P->h_contig_slab_off[c];
      P->h_count_n_index[c] = c;
    }
  }
  if (off >= (int64_t)1 << 31) return GAT_ERR_CAPACITY;
  P->slab_stride = off > 0 ? off : 1;
  {
    // what a contig's list is expected to need in k_contig: its units' segments + a quarter (a unit's capacity adds 96
    // slots of slack per unit: eight isochore units of fifty segments have 1 536 slots for ~400 segments)
    P->h_contig_need.assign((size_t)P->n_contigs, 64);
    for (int c = 0; c < P->n_contigs; ++c) {
      int64_t need = 0, ccap = 0;
      for (int ui = P->h_contig_unit_off[c]; ui < P->h_contig_unit_off[c + 1]; ++ui) {
        const int u = P->h_contig_units[ui];
        const int64_t n = (int64_t)P->h_units[u].hist_total;
        need += n + n / 4;
        ccap += P->h_units[u].slab_cap;
      }
      need = (need + 64 + 63) / 64 * 64;
      if (getenv("GAT_TEST_SMALL_CAPS")) need = std::max<int64_t>(64, need / 4 / 64 * 64);      // tests: force the repeat
      P->h_contig_need[c] = (int32_t)std::min<int64_t>(P->contig_tight ? need : ccap, std::max<int64_t>(64, ccap));
    }
    P->h_contig_order.resize((size_t)P->n_contigs);
    for (int c = 0; c < P->n_contigs; ++c) P->h_contig_order[c] = c;
    std::stable_sort(P->h_contig_order.begin(), P->h_contig_order.end(),
                     [&](int32_t x, int32_t y) { return P->h_contig_need[x] > P->h_contig_need[y]; });
    P->h_contig_class_start.clear();
    int32_t first = 0;
    for (int i = 0; i < P->n_contigs; ++i) {
      const int32_t need = P->h_contig_need[P->h_contig_order[i]];
      if (i == 0 || ((int64_t)need * 10 <= (int64_t)first * 7 && P->h_contig_class_start.size() < 6)) {
        P->h_contig_class_start.push_back(i);
        first = need;
      }
    }
    P->h_contig_class_start.push_back(P->n_contigs);
  }
  return GAT_OK;
}

static int upload_layout(gat_ctx* ctx, gat_problem* P) {
  HIPCHK(ctx, P->d_units.upload(P->h_units, ctx->stream));
  {
    // a wave finds its unit with one load (units_o[blockIdx.y]) instead of order[] -> units[]
    std::vector<UnitDev> o;
    o.reserve(P->h_order.size());
    for (int32_t u : P->h_order) { UnitDev x = P->h_units[(size_t)u]; x.pad = u; o.push_back(x); }
    if (o.empty()) o.push_back(UnitDev{});
    HIPCHK(ctx, P->d_units_o.upload(o, ctx->stream));
  }
  {
    // size classes of the launch order (largest unit first, capacities do not grow): a new class where the capacity has
    // dropped to 70 % of the class's first, at most six.  The wave-per-unit kernels keep a unit's list in LDS, their
    // speed follows the waves a CU holds, and the dynamic LDS of a launch is one number: sized for the longest list of
    // the PROBLEM, the config-4 shape ran ONE wave per CU (81 KB for chr1's 8 000 segments, 13 KB for chr21's).
    P->h_class_start.clear();
    const int N = (int)P->h_order.size();
    int32_t first_cap = 0;
    const char* env_c = getenv("GAT_SIZE_CLASSES");
    const int max_classes = env_c ? std::max(1, atoi(env_c)) : 6;
    for (int i = 0; i < N; ++i) {
      const int32_t cap = P->h_units[(size_t)P->h_order[(size_t)i]].slab_cap;
      if (i == 0 || ((int64_t)cap * 10 <= (int64_t)first_cap * 7 && (int)P->h_class_start.size() < max_classes)) {
        P->h_class_start.push_back(i);
        first_cap = cap;
      }
    }
    P->h_class_start.push_back(N);
  }
  HIPCHK(ctx, P->d_contig_slab_off.upload(P->h_contig_slab_off, ctx->stream));
  {
    std::vector<int32_t> o = P->h_contig_order;
    if (o.empty()) o.push_back(0);
    HIPCHK(ctx, P->d_contig_order.upload(o, ctx->stream));
  }
  HIPCHK(ctx, P->d_count_c_off.upload(P->h_count_c_off, ctx->stream));
  HIPCHK(ctx, P->d_count_n_index.upload(P->h_count_n_index, ctx->stream));
  P->batch = 0;   // scratch must be re-sized
  return GAT_OK;
}

extern "C" int gat_problem_create(gat_ctx* ctx, const gat_problem_desc* d, gat_problem** out) {
  if (!ctx || !d || !out) return set_err(ctx, GAT_ERR_ARG, "gat_problem_create: NULL argument");
  *out = nullptr;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  if (d->n_units < 0 || d->n_contigs < 0 || d->n_tracks < 0 || d->nbuckets <= 0)
    return set_err(ctx, GAT_ERR_ARG, "gat_problem_create: negative size / nbuckets <= 0");
  std::unique_ptr<gat_problem> P(new gat_problem());
  P->ctx = ctx;
  P->n_units = d->n_units;
  P->n_contigs = d->n_contigs;
  P->n_tracks = d->n_tracks;
  P->merge_contigs = d->merge_contigs ? 1 : 0;
  if (d->sampler != GAT_SAMPLER_ANNOTATOR && d->sampler != GAT_SAMPLER_SEGMENTS)
    return set_err(ctx, GAT_ERR_ARG, "unknown sampler %d", d->sampler);
  P->sampler = d->sampler;
  P->h_units.resize((size_t)d->n_units);
  P->h_base_cap.assign((size_t)d->n_units, 0);
  P->h_cws_nseg.assign(d->cws_nseg, d->cws_nseg + d->n_contigs);

  std::vector<uint2> h_ws;
  std::vector<uint32_t> h_ws_cdf, h_rank_len, h_ws_tree;
  std::vector<std::pair<int64_t, int32_t>> work;   // (working segments, unit)
  std::vector<std::vector<int32_t>> per_contig((size_t)d->n_contigs);
  std::vector<double> len_cv2((size_t)std::max(1, d->n_units), 0.0);

  for (int u = 0; u < d->n_units; ++u) {
    UnitDev& U = P->h_units[u];
    memset(&U, 0, sizeof(U));
    const gat_segment* us = d->segs + d->seg_off[u];
    const int64_t nus = d->seg_off[u + 1] - d->seg_off[u];
    const gat_segment* uw = d->ws + d->ws_off[u];
    const int64_t nuw = d->ws_off[u + 1] - d->ws_off[u];
    const int c = d->unit_contig[u];
    U.contig = c;
    P->n_seg_total += nus;
    const bool skipped = (nus == 0 || nuw == 0);     // gat/__init__.py:536-538
    if (c >= d->n_contigs || (c < 0 && !skipped))
      return set_err(ctx, GAT_ERR_ARG, "unit %d: contig index %d invalid (skipped units carry -1)", u, c);
    if (skipped) continue;
    if (c < 0) return set_err(ctx, GAT_ERR_ARG, "unit %d is not skipped by computeSample but has contig -1", u);
    per_contig[(size_t)c].push_back(u);
    int rc;
    if ((rc = check_list(ctx, us, nus, "segment", u))) return rc;     // gat/Engine.pyx:535
    if ((rc = check_list(ctx, uw, nuw, "workspace", u))) return rc;   // gat/Engine.pyx:536
    // working = segments.filter(workspace); ltotal = working.intersect(workspace).sum()
    uint32_t ltotal = 0, maxlen = 0;
    int64_t nwork = 0;
    std::vector<uint32_t> lens;
    lens.reserve((size_t)nus);
    for (int64_t i = 0; i < nus; ++i) {
      const uint32_t ov = host_overlap(uw, nuw, us[i].start, us[i].end);
      if (ov == 0) continue;
      ltotal += ov;
      const uint32_t l = us[i].end - us[i].start;
      lens.push_back(l);
      maxlen = std::max(maxlen, l);
      nwork++;
    }
    if (nwork == 0) continue;          // sample() returns an empty list, no RNG use (gat/Engine.pyx:545-546)
    // getLengthDistribution (gat/SegmentList.pyx:1148-1184)
    int64_t bucket = d->bucket_size;
    if (bucket == 0) bucket = (int64_t)std::ceil((double)(int32_t)maxlen / (double)d->nbuckets);
    std::map<uint32_t, uint32_t> hist;
    for (uint32_t l : lens) {
      const int64_t i = ((int64_t)l + bucket - 1) / bucket;
      if (i >= d->nbuckets)
        return set_err(ctx, GAT_ERR_VALUE, "unit %d: segment of length %u too large: increase nbuckets (%d) or bucket_size (%lld)",
                       u, l, d->nbuckets, (long long)bucket);
      hist[(uint32_t)i] += 1;
    }
    uint32_t cum = 0;
    U.rank_off = (int32_t)h_rank_len.size();
    h_rank_len.push_back(0u);                       // rank 0 is never drawn (r >= 1, gat/Engine.pyx:419-422)
    for (auto& kv : hist) {
      cum += kv.second;
      for (uint32_t q = 0; q < kv.second; ++q) h_rank_len.push_back(kv.first);   // ranks (cum-count, cum]
    }
    U.hist_total = cum;
    U.bucket = (uint32_t)bucket;
    {
      double m1 = 0, m2 = 0;                               // squared coefficient of variation of the lengths drawn
      for (uint32_t l : lens) { m1 += (double)l; m2 += (double)l * (double)l; }
      m1 /= (double)lens.size(); m2 /= (double)lens.size();
      len_cv2[(size_t)u] = m1 > 0 ? std::max(0.0, m2 / (m1 * m1) - 1.0) : 0.0;
    }
    // SegmentListSampler(workspace) (gat/Engine.pyx:261-277)
    U.n_ws = (int32_t)nuw;
    U.ws_off = (int32_t)h_ws.size();
    uint32_t tot = 0;
    for (int64_t i = 0; i < nuw; ++i) {
      tot += uw[i].end - uw[i].start;
      h_ws.push_back(make_uint2(uw[i].start, uw[i].end));
      h_ws_cdf.push_back(tot - 1u);
    }
    U.ws_total = tot;
    // long workspaces: 16-ary search trees (gat_device.h, WsTree) over the starts and over the cumulated lengths
    U.tree_start_off = -1;
    U.tree_cdf_off = -1;
    if (nuw > ((int64_t)1 << (4 * gat::kWsTreeLevels)))
      return set_err(ctx, GAT_ERR_CAPACITY, "unit %d: %lld workspace segments (> %lld)", u, (long long)nuw,
                     (long long)((int64_t)1 << (4 * gat::kWsTreeLevels)));
    if (nuw > gat::kWsTreeMin) {
      auto build = [&](auto key, uint32_t pad) {
        const int32_t off = (int32_t)h_ws_tree.size();
        std::vector<uint32_t> level((size_t)nuw);
        for (int64_t i = 0; i < nuw; ++i) level[(size_t)i] = key(i);
        for (;;) {
          const size_t n = level.size(), nodes = (n + 15) / 16;
          h_ws_tree.insert(h_ws_tree.end(), level.begin(), level.end());
          h_ws_tree.insert(h_ws_tree.end(), nodes * 16 - n, pad);
          if (n <= 16) break;
          std::vector<uint32_t> up(nodes);
          for (size_t j = 0; j < nodes; ++j) up[j] = level[std::min(16 * j + 15, n - 1)];   // largest key of node j
          level.swap(up);
        }
        return off;
      };
      U.tree_start_off = build([&](int64_t i) { return uw[i].start; }, 0xffffffffu);
      const size_t base = h_ws_cdf.size() - (size_t)nuw;
      U.tree_cdf_off = build([&](int64_t i) { return h_ws_cdf[base + (size_t)i]; }, 0x7fffffffu);
    }
    U.ltotal = (int32_t)ltotal;
    U.n_target = (int32_t)nus;                       // SamplerSegments places len(segments) segments
    P->h_base_cap[u] = cap_for(d->sampler == GAT_SAMPLER_SEGMENTS ? std::max<int64_t>(nwork, nus) : nwork);
    work.push_back(std::make_pair(nwork, (int32_t)u));
  }
  // contig -> units (reference order)
  P->h_contig_unit_off.assign((size_t)d->n_contigs + 1, 0);
  for (int c = 0; c < d->n_contigs; ++c) {
    if (per_contig[(size_t)c].empty())
      return set_err(ctx, GAT_ERR_ARG, "contig %d has no unit: contigs must be those of the non-skipped units", c);
    if (!P->merge_contigs && per_contig[(size_t)c].size() != 1)
      return set_err(ctx, GAT_ERR_ARG, "contig %d has %zu units but keys carry no isochore (merge_contigs=0)", c, per_contig[(size_t)c].size());
    for (int32_t u : per_contig[(size_t)c]) P->h_contig_units.push_back(u);
    P->h_contig_unit_off[(size_t)c + 1] = (int32_t)P->h_contig_units.size();
  }
  std::sort(work.begin(), work.end(), [](const std::pair<int64_t, int32_t>& a, const std::pair<int64_t, int32_t>& b) {
    return a.first != b.first ? a.first > b.first : a.second < b.second;
  });
  for (auto& w : work) P->h_order.push_back(w.second);
  P->all_simple = !P->h_order.empty();
  uint32_t max_hist = 0;
  for (int32_t u : P->h_order) {
    const UnitDev& U = P->h_units[(size_t)u];
    const bool degenerate = !(U.hist_total > 2 && U.ws_total > 1);          // k_place leaves those to k_sampler
    const bool simple = U.n_ws == 1 && U.bucket <= 1 && U.hist_total < (uint32_t)gat::kPlaceRankLds && U.ws_total > 1;
    if (!degenerate && !simple) P->all_simple = false;
    P->max_nws = std::max(P->max_nws, U.n_ws);
    max_hist = std::max(max_hist, U.hist_total);
  }
  P->small_tables = !P->h_order.empty() && P->max_nws <= 64 && max_hist < 256;
  P->long_lists = max_hist + max_hist / 8 > 1024;
  P->max_hist = max_hist;
  {
    // the split path pays when k_tail can take most units: SamplerAnnotator, lists the wave bucket sorts hold, workspaces
    // of up to kTailMaxWs segments
    size_t small_ws = 0;
    for (int32_t u : P->h_order) if (P->h_units[(size_t)u].n_ws <= gat::kTailMaxWs) ++small_ws;
    // (long lists: their tail places dozens of segments, not the handful k_tail keeps aside -- 0.2 % finished there on the
    //  config-4 shape -- so those problems stay with k_merge_big + k_sampler)
    P->split_path = P->sampler == GAT_SAMPLER_ANNOTATOR && !P->h_order.empty() && 2 * small_ws >= P->h_order.size() &&
                    max_hist + max_hist / 8 <= 1024 && !getenv("GAT_NO_SPLIT");
  }
  // expected raw MT19937 outputs per placement under masked rejection (mask+1)/(range+1) per draw;
  // rows = that x working segments + slack, in whole 624-word blocks.  Streams that still run out
  // are redone by k_sampler from their seed.
  {
    const char* env = getenv("GAT_SAMPLER_MODE");
    if (env && !strcmp(env, "wave")) P->sampler_mode = 0;
    auto expect = [](uint64_t range) {
      if (range == 0) return 0.0;
      uint64_t m = range; m |= m >> 1; m |= m >> 2; m |= m >> 4; m |= m >> 8; m |= m >> 16; m |= m >> 32;
      return (double)(m + 1) / (double)(range + 1);
    };
    for (int32_t u : P->h_order) {
      const UnitDev& U = P->h_units[u];
      // offset draw: range = chosen workspace segment + sampled length - 2, weighted by how often a segment
      // is chosen (its share of the workspace) and taken at the mean working-segment length
      const gat_segment* uw = d->ws + d->ws_off[u];
      const int64_t nuw = d->ws_off[u + 1] - d->ws_off[u];
      const double mean_len = U.hist_total ? (double)(uint32_t)U.ltotal / (double)U.hist_total : 1.0;
      double e = 0.0, v = 0.0;
      for (int64_t k = 0; k < nuw; ++k) {
        const double wl = (double)(uw[k].end - uw[k].start);
        const double p = expect((uint64_t)(wl + mean_len)) ;
        e += wl / (double)U.ws_total * p;
      }
      auto addvar = [&](double ex) { if (ex > 0) v += (ex - 1.0) * ex; };   // geometric: var = (1-p)/p^2 = ex(ex-1)
      addvar(e);
      if (U.hist_total > 2) { const double x = expect((uint64_t)U.hist_total - 2); e += x; addvar(x); }
      if (U.bucket > 1) { const double x = expect((uint64_t)U.bucket - 1); e += x; addvar(x); }
      if (U.ws_total > 1) { const double x = expect((uint64_t)U.ws_total - 1); e += x; addvar(x); }
      const char* env_sl = getenv("GAT_RNG_SLACK");
      const double slack = env_sl ? atof(env_sl) : 1.0;
      // Spread of the raw-output count of a stream: the NUMBER of placements until the unit's bases are reproduced varies
      // by cv(length) x sqrt(n) (a renewal count) and every placement costs e outputs -- that term dominates (measured on
      // config 2: 97 / 116 / 58 outputs for units of 778 / 444 / 166 segments = e x cv x sqrt(n)) -- plus the rejection
      // noise v per placement.  5 to 7.5 sigma and the tail's few dozen outputs: a stream that runs out is redone from its seed
      // by ONE wave, placement by placement, and such a straggler (0.5 ms) is now longer than the rest of the sampler.
      const double nplace = P->sampler == GAT_SAMPLER_SEGMENTS ? (double)U.n_target : (double)U.hist_total;
      const double var_n = P->sampler == GAT_SAMPLER_SEGMENTS ? 0.0 : len_cv2[(size_t)u] * e * e;
      // (the multiple follows what running out costs: ONE wave redoing the unit placement by placement -- 0.5 ms for 800
      //  segments, a few dozen us for 50, where five sigma are plenty and the rows saved are a sixth of k_rng's work)
      const double sigmas = std::min(7.5, std::max(5.0, 4.5 + nplace / 130.0));
      const double need = e * nplace * slack + sigmas * std::sqrt(nplace * (v + 0.5 + var_n)) + 96.0;
      int64_t rows = ((int64_t)std::ceil(need / 16.0)) * 16;        // whole k_place chunks (8) and k_rng read groups (16)
      rows = std::min<int64_t>(rows, (int64_t)gat::kMtN * 2048);
      P->h_rng_rows.push_back((int32_t)rows);
      P->rng_rows_total += rows;
    }
  }
  P->h_contig_slab_off.assign((size_t)d->n_contigs, 0);
  P->h_count_c_off.assign((size_t)d->n_contigs, 0);
  P->h_count_n_index.assign((size_t)d->n_contigs, 0);
  if (layout_slab(P.get())) return set_err(ctx, GAT_ERR_CAPACITY, "per-sample slab exceeds 2^31 segments");

  HIPCHK(ctx, P->d_order.upload(P->h_order, ctx->stream));
  {
    std::vector<int32_t> pos((size_t)std::max(1, d->n_units), -1);
    for (size_t a = 0; a < P->h_order.size(); ++a) pos[(size_t)P->h_order[a]] = (int32_t)a;
    HIPCHK(ctx, P->d_unit_pos.upload(pos, ctx->stream));
  }
  HIPCHK(ctx, P->d_rng_rows.upload(P->h_rng_rows, ctx->stream));
  HIPCHK(ctx, P->d_contig_unit_off.upload(P->h_contig_unit_off, ctx->stream));
  HIPCHK(ctx, P->d_contig_units.upload(P->h_contig_units, ctx->stream));
  HIPCHK(ctx, P->d_ws.upload(h_ws, ctx->stream));
  HIPCHK(ctx, P->d_ws_cdf.upload(h_ws_cdf, ctx->stream));
  if (h_ws_tree.empty()) h_ws_tree.assign(16, 0u);
  HIPCHK(ctx, P->d_ws_tree.upload(h_ws_tree, ctx->stream));
  HIPCHK(ctx, P->d_rank_len.upload(h_rank_len, ctx->stream));
  HIPCHK(ctx, P->d_cws_nseg.upload(P->h_cws_nseg, ctx->stream));
  int rc = upload_layout(ctx, P.get());
  if (rc) return rc;
  const auto t_annos = std::chrono::steady_clock::now();
  rc = build_annos(ctx, P->annos, d->annos, d->anno_off, (int64_t)d->n_tracks * d->n_contigs, d->n_contigs);
  if (getenv("GAT_TIME_CREATE"))
    fprintf(stderr, "[gat] build_annos (incl. build_merged) %.1f ms\n",
            std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_annos).count());
  if (rc) return rc;
  {
    // sample lists much longer than the annotation lists they meet: swap the roles in the count kernel
    const double avg_n = P->n_contigs ? (double)P->n_seg_total / P->n_contigs : 0.0;
    const double avg_m = (P->n_contigs && P->n_tracks) ? (double)P->annos.total / ((double)P->n_contigs * P->n_tracks) : 0.0;
    const int capx = P->merge_contigs ? P->max_contig_cap : P->max_unit_cap;
    const size_t lds_need = (size_t)3 * capx * 4 + (8192 + 1) * 4;
    (void)lds_need; (void)capx;
    if (avg_m > 0 && avg_n > 3.0 * avg_m) P->swap_capx = 1;       // capacity is taken from the slab layout at launch
  }
  HIPCHK(ctx, P->d_flags.alloc(1));
  HIPCHK(ctx, P->d_stat.alloc(8));
  *out = P.release();
  return GAT_OK;
}

extern "C" void gat_problem_destroy(gat_problem* p) {
  if (!p) return;
  if (p->ctx) (void)hipSetDevice(p->ctx->device);
  delete p;
}

extern "C" int gat_problem_info(const gat_problem* p, int64_t* n_units, int64_t* n_contigs, int64_t* n_tracks,
                                int64_t* slab, int64_t* bytes) {
  if (!p) return set_err(nullptr, GAT_ERR_ARG, "NULL problem");
  if (n_units) *n_units = p->n_units;
  if (n_contigs) *n_contigs = p->n_contigs;
  if (n_tracks) *n_tracks = p->n_tracks;
  if (slab) *slab = p->slab_stride;
  // B_sample = 8*sum n' + 8*sum_a sum_c m + 8*A, with n' ~ n input segments
  if (bytes) *bytes = 8 * p->n_seg_total + 8 * p->annos.total + 8 * (int64_t)p->n_tracks;
  return GAT_OK;
}

// ------------------------------------------------------------------------------------------
static int ensure_scratch(gat_ctx* ctx, gat_problem* P, int64_t want) {
  if (P->batch >= want) return GAT_OK;
  const char* env = getenv("GAT_SLAB_BYTES");
  // a batch as large as a tenth of the 288 GB takes: the lane-per-stream kernels have a fixed floor per launch (the serial
  // chain of the longest unit's tile), so fewer, larger batches are faster (config 3, 10 000 samples: 9.7 ms in two
  // batches under 12 GB, 9.3 ms in one); capped by what the device has free
  double budget = env ? atof(env) : 30.0 * 1024 * 1024 * 1024;
  if (!env) {
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) budget = std::min(budget, 0.6 * (double)free_b);
  }
  const int64_t per_sample = P->slab_stride * 8 * (P->merge_contigs ? 2 : 1) + 4 * ((int64_t)P->n_units + P->n_contigs) +
                             (P->sampler_mode ? P->rng_rows_total * 4 + 16 * (int64_t)P->n_units : 0) +
                             (P->split_path ? P->slab_stride * 12 + (int64_t)(sizeof(gat::TailPatch) + 4) * P->n_units : 0);
  int64_t b = (int64_t)(budget / (double)per_sample);
  b = std::max<int64_t>(1, std::min<int64_t>(b, want));
  if (P->batch >= b) return GAT_OK;
  HIPCHK(ctx, P->d_slab.alloc((size_t)(b * P->slab_stride)));
  if (P->merge_contigs) HIPCHK(ctx, P->d_cslab.alloc((size_t)(b * P->slab_stride)));
  HIPCHK(ctx, P->d_unit_n.alloc((size_t)(b * std::max(1, P->n_units))));
  HIPCHK(ctx, P->d_contig_n.alloc((size_t)(b * std::max(1, P->n_contigs))));
  HIPCHK(ctx, P->d_ws_stat.alloc((size_t)(b * std::max(1, P->n_units)) * 4));
  HIPCHK(ctx, hipMemsetAsync(P->d_unit_n.p, 0, (size_t)(b * std::max(1, P->n_units)) * 4, ctx->stream));
  HIPCHK(ctx, hipMemsetAsync(P->d_contig_n.p, 0, (size_t)(b * std::max(1, P->n_contigs)) * 4, ctx->stream));
  HIPCHK(ctx, hipMemsetAsync(P->d_ws_stat.p, 0, (size_t)(b * std::max(1, P->n_units)) * 16, ctx->stream));
  if (P->sampler_mode) {
    const int64_t nsb = (b + 63) / 64;
    P->h_rng_off.assign(P->h_order.size() + 1, 0);
    for (size_t a = 0; a < P->h_order.size(); ++a) P->h_rng_off[a + 1] = P->h_rng_off[a] + nsb * (int64_t)P->h_rng_rows[a] * 64;
    HIPCHK(ctx, P->d_rng_off.upload(P->h_rng_off, ctx->stream));
    HIPCHK(ctx, P->d_rng_out.alloc((size_t)P->h_rng_off.back()));
    const size_t ns = (size_t)(b * std::max(1, P->n_units));
    HIPCHK(ctx, P->d_st.alloc(ns));
    HIPCHK(ctx, P->d_st2.alloc(ns));
    if (!P->split_path && P->long_lists) {                                      // k_tail_big's hand-over records
      HIPCHK(ctx, P->d_patch.alloc(ns));
      HIPCHK(ctx, hipMemsetAsync(P->d_patch.p, 0, ns * sizeof(gat::TailPatch), ctx->stream));
    }
    if (P->split_path) {
      HIPCHK(ctx, P->d_cum.alloc((size_t)(b * P->slab_stride)));
      HIPCHK(ctx, P->d_fslab.alloc((size_t)(b * P->slab_stride)));
      HIPCHK(ctx, P->d_patch.alloc(ns));
      HIPCHK(ctx, P->d_todo.alloc(ns));
      HIPCHK(ctx, P->d_todo_count.alloc(1));
    }
  }
  P->batch = b;
  return GAT_OK;
}

struct Counters {
  int32_t slot[GAT_NUM_COUNTERS];
  bool any_seg = false, any_anno = false;
};

static int parse_counters(gat_ctx* ctx, const int32_t* ids, int n, Counters& C) {
  for (int i = 0; i < GAT_NUM_COUNTERS; ++i) C.slot[i] = -1;
  for (int k = 0; k < n; ++k) {
    if (ids[k] < 0 || ids[k] >= GAT_NUM_COUNTERS) return set_err(ctx, GAT_ERR_ARG, "unknown counter id %d", ids[k]);
    if (C.slot[ids[k]] >= 0) return set_err(ctx, GAT_ERR_ARG, "counter id %d given twice", ids[k]);
    C.slot[ids[k]] = k;
    if (ids[k] <= GAT_COUNTER_SEGMENT_MIDOVERLAP) C.any_seg = true; else C.any_anno = true;
  }
  return GAT_OK;
}

// which kernel serves the segment-side counters (the choice launch_count makes)
static int count_route(const gat_ctx* ctx, const AnnoDev& annos, const Counters& C, int n_contigs, int n_tracks, int swap_capx) {
  if (!C.any_seg || n_contigs <= 0) return GAT_COUNT_KERNEL_NONE;
  const bool only_overlap = C.slot[GAT_COUNTER_SEGMENT_OVERLAP] < 0 && C.slot[GAT_COUNTER_SEGMENT_MIDOVERLAP] < 0;
  const size_t lds_merged = (size_t)n_tracks * 4 * (gat::kMergedThreads / gat::kWave);
  if (only_overlap && annos.has_merged && (int64_t)lds_merged + 1024 <= ctx->max_lds && !getenv("GAT_COUNT_NO_MERGED"))
    return GAT_COUNT_KERNEL_MERGED;
  if (swap_capx > 0 && only_overlap && !getenv("GAT_COUNT_NO_SWAP")) return GAT_COUNT_KERNEL_SWAP;
  return GAT_COUNT_KERNEL_SEG;
}

// launch the count kernels over n_lists sample lists
static int launch_count(gat_ctx* ctx, const AnnoDev& annos, const Counters& C, gat::CountArgs A, DevBuf<uint32_t>& part,
                        int swap_capx = 0, int list_cap = 0) {
  for (int i = 0; i < GAT_NUM_COUNTERS; ++i) A.counter_slot[i] = C.slot[i];
  A.a_start = annos.start.p; A.a_end = annos.end.p; A.a_cumx = annos.cumx.p; A.a_off = annos.off.p;
  A.a_grid = annos.grid.p; A.g_off = annos.goff.p; A.c_shift = annos.shift.p; A.c_cells = annos.cells.p;
  if (A.n_samples <= 0 || A.n_tracks <= 0) return GAT_OK;
  if (C.any_seg) {
    const char* env_e = getenv("GAT_COUNT_LDS_ENTRIES");
    const int E_max = env_e ? atoi(env_e) : 1024;
    const char* env_sc = getenv("GAT_COUNT_SAMPLES_PER_BLOCK");
    int SC = env_sc ? atoi(env_sc) : 32;
    if (!env_sc) {   // enough blocks to fill 256 CUs several times over, large enough to amortise the staging
      const int64_t tiles0 = (A.n_tracks + 0) , work = (int64_t)A.n_samples * tiles0 * std::max(1, A.n_contigs) / 8192;
      SC = 8;
      while (SC < 128 && SC * 2 <= work) SC *= 2;
    }
    SC = std::max(1, std::min(SC, 256));
    const char* env_tt = getenv("GAT_COUNT_TRACKS_PER_BLOCK");
    const int TT_max = env_tt ? atoi(env_tt) : 16;
    int TT;
    const char* env_st = getenv("GAT_COUNT_STAGED");
    bool staged = annos.max_m > 0 && annos.max_m + 1 <= E_max && !(env_st && atoi(env_st) == 0);   // (+1: sentinel)
    if (staged) TT = (int)std::min<int64_t>(std::min<int64_t>(A.n_tracks, TT_max), E_max / (annos.max_m + 1));
    else TT = (int)std::min<int64_t>(A.n_tracks, TT_max);
    TT = std::max(1, TT);
    A.tracks_per_block = TT;
    A.samples_per_block = SC;
    A.lds_entries = staged ? (int)std::min<int64_t>((int64_t)E_max, (annos.max_m + 1) * TT) : 0;
    A.lds_grid = staged ? (int)((annos.max_cells + 1) * TT) : 0;
    const size_t lds = (size_t)((TT + 1 + 3) & ~3) * 4 + (size_t)3 * A.lds_entries * 4 + (size_t)A.lds_grid * 4;
    const size_t need = (size_t)A.n_contigs * 3 * (size_t)A.n_tracks * (size_t)A.n_samples;
    if (part.n < need) HIPCHK(ctx, part.alloc(need));
    A.part = part.p;
    // (track tile, contig) pairs over grid y and z, contigs alone likewise for the kernels gridded by contig
    const int64_t n_pairs = (int64_t)((A.n_tracks + TT - 1) / TT) * std::max(1, A.n_contigs);
    const unsigned gpy = (unsigned)std::min<int64_t>(n_pairs, 32768), gpz = (unsigned)((n_pairs + gpy - 1) / gpy);
    if (gpz > 65535) return set_err(ctx, GAT_ERR_CAPACITY, "more than 2^31 (track tile, contig) pairs");
    dim3 grid((unsigned)((A.n_samples + SC - 1) / SC), gpy, gpz);
    const unsigned gcy = (unsigned)std::min(std::max(1, A.n_contigs), 32768), gcz = ((unsigned)std::max(1, A.n_contigs) + gcy - 1) / gcy;
    const size_t lds_merged = (size_t)A.n_tracks * 4 * (gat::kMergedThreads / gat::kWave);
    const int route = count_route(ctx, annos, C, A.n_contigs, A.n_tracks, swap_capx);
    if (route == GAT_COUNT_KERNEL_MERGED) {
      // several tracks: one look-up per sample segment in the merged index of all tracks
      A.mz = annos.mz.p; A.mz_off = annos.mz_off.p; A.mfirst = annos.mfirst.p; A.mf_off = annos.mf_off.p;
      A.m_shift = annos.m_shift.p; A.m_cells = annos.m_cells.p;
      A.m_slot_off = annos.m_slot_off.p; A.m_slot_contigs = annos.m_slot_contigs.p;
      const char* env_sg = getenv("GAT_MERGED_SAMPLES_PER_BLOCK");
      // samples per workgroup: one per wave.  (Larger groups were meant to keep the contig's index hot; measured on config 3,
      // main kernel per 10 000 samples: 4 -> 1.60 ms, 8 -> 1.69, 16 -> 1.73, 32 -> 1.83, 64 -> 2.1: the finer deal wins.)
      int SG = env_sg ? atoi(env_sg) : 4;
      SG = std::max(1, std::min(SG, std::max(1, A.n_samples)));
      A.samples_per_block = SG;
      const int64_t n_sgroups = (A.n_samples + SG - 1) / SG;
      const int64_t nblocks = (int64_t)gat::kMergedSlots * annos.max_slot_contigs * n_sgroups;
      if (nblocks >= ((int64_t)1 << 31)) return set_err(ctx, GAT_ERR_CAPACITY, "more than 2^31 (contig, sample group) pairs");
      const size_t need = (size_t)A.n_contigs * (size_t)A.n_tracks * (size_t)A.n_samples;
      if (part.n < need) HIPCHK(ctx, part.alloc(need));
      A.part = part.p;
      HIPCHK(ctx, hipFuncSetAttribute((const void*)gat::k_count_merged<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_merged));
      HIPCHK(ctx, hipFuncSetAttribute((const void*)gat::k_count_merged<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_merged));
      HIPCHK(ctx, hipEventRecord(ctx->ev_main[0], ctx->stream));
      if (A.seg_merged != nullptr) hipLaunchKernelGGL(gat::k_count_merged<true>, dim3((unsigned)nblocks), dim3(gat::kMergedThreads), lds_merged, ctx->stream, A);
      else hipLaunchKernelGGL(gat::k_count_merged<false>, dim3((unsigned)nblocks), dim3(gat::kMergedThreads), lds_merged, ctx->stream, A);
      HIPCHK(ctx, hipGetLastError());
      HIPCHK(ctx, hipEventRecord(ctx->ev_main[1], ctx->stream));
      ctx->main_recorded = true;
      ctx->count_kernel = GAT_COUNT_KERNEL_MERGED;
      const int64_t tiles = (int64_t)((A.n_tracks + 15) / 16) * ((A.n_samples + 15) / 16);
      hipLaunchKernelGGL(gat::k_count_merged_finish, dim3((unsigned)tiles), dim3(256), 0, ctx->stream, A);
      HIPCHK(ctx, hipGetLastError());
      goto seg_done;
    }
    if (route == GAT_COUNT_KERNEL_SWAP) {
      // long sample lists against short annotation lists: index the sample list, stream the tracks
      gat::CountArgs B = A;
      int lcells = 4;
      while ((1 << lcells) < swap_capx && lcells < 13) ++lcells;
      B.lds_entries = swap_capx;
      B.lds_grid = lcells;
      const size_t lds_swap = (size_t)3 * swap_capx * 4 + ((size_t)(1 << lcells) + 1) * 4;
      HIPCHK(ctx, hipFuncSetAttribute((const void*)gat::k_count_swap, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_swap));
      HIPCHK(ctx, hipEventRecord(ctx->ev_main[0], ctx->stream));
      hipLaunchKernelGGL(gat::k_count_swap, dim3((unsigned)A.n_samples, gcy, gcz), dim3(gat::kSwapThreads), lds_swap, ctx->stream, B);
      HIPCHK(ctx, hipGetLastError());
      HIPCHK(ctx, hipEventRecord(ctx->ev_main[1], ctx->stream));
      ctx->main_recorded = true;
      ctx->count_kernel = GAT_COUNT_KERNEL_SWAP;
    } else
    if (A.n_contigs > 0) {
    const bool hits = C.slot[GAT_COUNTER_SEGMENT_OVERLAP] >= 0 || C.slot[GAT_COUNTER_SEGMENT_MIDOVERLAP] >= 0;
    HIPCHK(ctx, hipEventRecord(ctx->ev_main[0], ctx->stream));
    const int kv = (staged ? 4 : 0) + (hits ? 2 : 0) + (A.seg_merged != nullptr ? 1 : 0);
    const void* fn = kv == 7 ? (const void*)gat::k_count_seg<true, true, true> : kv == 6 ? (const void*)gat::k_count_seg<true, true, false>
                   : kv == 5 ? (const void*)gat::k_count_seg<true, false, true> : kv == 4 ? (const void*)gat::k_count_seg<true, false, false>
                   : kv == 3 ? (const void*)gat::k_count_seg<false, true, true> : kv == 2 ? (const void*)gat::k_count_seg<false, true, false>
                   : kv == 1 ? (const void*)gat::k_count_seg<false, false, true> : (const void*)gat::k_count_seg<false, false, false>;
    if (staged) HIPCHK(ctx, hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    switch (kv) {
      case 7: hipLaunchKernelGGL((gat::k_count_seg<true, true, true>), grid, dim3(256), lds, ctx->stream, A); break;
      case 6: hipLaunchKernelGGL((gat::k_count_seg<true, true, false>), grid, dim3(256), lds, ctx->stream, A); break;
      case 5: hipLaunchKernelGGL((gat::k_count_seg<true, false, true>), grid, dim3(256), lds, ctx->stream, A); break;
      case 4: hipLaunchKernelGGL((gat::k_count_seg<true, false, false>), grid, dim3(256), lds, ctx->stream, A); break;
      case 3: hipLaunchKernelGGL((gat::k_count_seg<false, true, true>), grid, dim3(256), lds, ctx->stream, A); break;
      case 2: hipLaunchKernelGGL((gat::k_count_seg<false, true, false>), grid, dim3(256), lds, ctx->stream, A); break;
      case 1: hipLaunchKernelGGL((gat::k_count_seg<false, false, true>), grid, dim3(256), lds, ctx->stream, A); break;
      default: hipLaunchKernelGGL((gat::k_count_seg<false, false, false>), grid, dim3(256), lds, ctx->stream, A); break;
    }
    HIPCHK(ctx, hipGetLastError());
    HIPCHK(ctx, hipEventRecord(ctx->ev_main[1], ctx->stream));
    ctx->main_recorded = true;
    ctx->count_kernel = GAT_COUNT_KERNEL_SEG;
    }
    {
      const int64_t nfin = (int64_t)A.n_tracks * A.n_samples;
      hipLaunchKernelGGL(gat::k_count_finish, dim3((unsigned)((nfin + 255) / 256)), dim3(256), 0, ctx->stream, A);
      HIPCHK(ctx, hipGetLastError());
    }
  seg_done:;
  }
  if (C.any_anno) {
    // the sample lists indexed in LDS when they fit (list_cap = longest list possible), else every interval bisects
    // the list in global memory
    int lcells = 4;
    while ((1 << lcells) < list_cap + 1 && lcells < 13) ++lcells;
    const size_t lds_a = (size_t)2 * (list_cap + 1) * 4 + ((size_t)(1 << lcells) + 1) * 4;
    if (list_cap > 0 && A.n_contigs > 0 && (int64_t)lds_a + 1024 <= ctx->max_lds && !getenv("GAT_COUNT_NO_SWAP")) {
      gat::CountArgs B = A;
      B.lds_entries = list_cap + 1;
      B.lds_grid = lcells;
      const size_t need = (size_t)A.n_contigs * 2 * (size_t)A.n_tracks * (size_t)A.n_samples;
      if (part.n < need) HIPCHK(ctx, part.alloc(need));
      B.part = part.p;
      const unsigned gcy = (unsigned)std::min(A.n_contigs, 32768), gcz = ((unsigned)A.n_contigs + gcy - 1) / gcy;
      HIPCHK(ctx, hipFuncSetAttribute((const void*)gat::k_count_anno_idx, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_a));
      hipLaunchKernelGGL(gat::k_count_anno_idx, dim3((unsigned)A.n_samples, gcy, gcz), dim3(gat::kAnnoThreads), lds_a, ctx->stream, B);
      HIPCHK(ctx, hipGetLastError());
      const int64_t nfin = (int64_t)A.n_tracks * A.n_samples;
      hipLaunchKernelGGL(gat::k_count_anno_finish, dim3((unsigned)((nfin + 255) / 256)), dim3(256), 0, ctx->stream, B);
      HIPCHK(ctx, hipGetLastError());
    } else {
      const int64_t waves = (int64_t)A.n_samples * A.n_tracks;
      hipLaunchKernelGGL(gat::k_count_anno, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, ctx->stream, A);
      HIPCHK(ctx, hipGetLastError());
    }
  }
  return GAT_OK;
}

// sampler (+ fromIsochores) for one batch; retries with a larger slab on overflow
// Returns GAT_OK, an error, or kRelayout: a unit's region overflowed, the slab was laid out again with doubled
// capacities (scratch released: the batch that fits the budget may now be smaller) and nothing of this batch is valid;
// the caller sizes the batch again and repeats it (results do not depend on the batching: streams are per unit).
constexpr int kRelayout = 1;
static int finish_sampler_batch(gat_ctx* ctx, gat_problem* P, int64_t nb, gat_stats* st, bool timed);
// defer: only enqueue (the caller adds the count kernels behind, synchronises once and calls finish_sampler_batch)
// records_ok: the consumer is k_count_seg alone, which reads (merged list, k_tail's record): no k_finalize
// serial_state: the run's ONE MT19937 state on the device (k_serial: the reference's own stream) instead of the per-unit streams
static int run_sampler_batch(gat_ctx* ctx, gat_problem* P, uint32_t seed, int64_t begin, int64_t nb,
                             gat_stats* st, bool timed, bool need_unit_lists = false, bool defer = false, bool records_ok = false,
                             uint32_t* serial_state = nullptr) {
  {
    int rc = ensure_scratch(ctx, P, nb);
    if (rc) return rc;
    if (P->batch < nb) return set_err(ctx, GAT_ERR_MEMORY, "internal: batch %lld > scratch %lld", (long long)nb, (long long)P->batch);
    // (unit_n, contig_n and ws_stat are zeroed once when allocated: the kernels rewrite every entry of the active units
    //  in every batch and never touch the others)
    HIPCHK(ctx, hipMemsetAsync(P->d_flags.p, 0, 4, ctx->stream));
    HIPCHK(ctx, hipMemsetAsync(P->d_stat.p, 0, 8 * 8, ctx->stream));
    if (timed) HIPCHK(ctx, hipEventRecord(ctx->ev[0], ctx->stream));
    const int32_t* skip_ptr = nullptr;
    int skip_stride = 0;
    if (!P->h_order.empty()) {
      const int smode = serial_state != nullptr ? 0 : P->sampler_mode;
      gat::SamplerArgs A;
      memset(&A, 0, sizeof(A));
      A.units = P->d_units.p; A.units_o = P->d_units_o.p; A.order = P->d_order.p; A.n_units = P->n_units; A.batch = (int32_t)nb;
      A.ws = P->d_ws.p; A.ws_cdf = P->d_ws_cdf.p; A.rank_len = P->d_rank_len.p; A.ws_tree = P->d_ws_tree.p;
      A.seed = seed; A.sample_begin = begin; A.sampler_kind = P->sampler;
      A.slab = P->d_slab.p; A.slab_stride = P->slab_stride;
      A.unit_n = P->d_unit_n.p; A.flags = P->d_flags.p; A.stat = P->d_stat.p; A.ws_stat = P->d_ws_stat.p;
#ifdef GAT_DIAG
      {
        const size_t nd = (size_t)nb * std::max(1, P->n_units) * 8;
        if (P->d_diag.n < nd) HIPCHK(ctx, P->d_diag.alloc(nd));
        HIPCHK(ctx, hipMemsetAsync(P->d_diag.p, 0, nd * 8, ctx->stream));
        A.diag = P->d_diag.p;
      }
#endif
      // the units' launch positions are spread over grid y and z (each <= 65535)
      const unsigned n_act = (unsigned)P->h_order.size();
      const unsigned gy = std::min(n_act, 32768u), gz = (n_act + gy - 1) / std::max(gy, 1u);
      A.n_active = (int32_t)n_act;
      if (smode) {
        // lane-parallel front end: the scratch was sized for P->batch samples, tiles are laid out for that
        const int64_t nsb_alloc = (P->batch + 63) / 64;
        const unsigned nsb = (unsigned)((nb + 63) / 64);
        (void)nsb_alloc;
        A.rng_off = P->d_rng_off.p; A.rng_rows = P->d_rng_rows.p; A.rng_out = P->d_rng_out.p;
        A.st = P->d_st.p;
        const size_t lds_rng = (size_t)gat::kMtN * 64 * 4;
        HIPCHK(ctx, hipFuncSetAttribute((const void*)gat::k_rng, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_rng));
        hipLaunchKernelGGL(gat::k_rng, dim3(nsb, gy, gz), dim3(gat::kRngThreads), lds_rng, ctx->stream, A);
        HIPCHK(ctx, hipGetLastError());
        if (timed) HIPCHK(ctx, hipEventRecord(ctx->ev_k[0], ctx->stream));
        const dim3 gp(nsb, gy, gz);
        const int mode = P->all_simple ? 1 : (P->max_nws > gat::kPlaceWsLds ? 2 : 0);
        if (P->sampler == GAT_SAMPLER_SEGMENTS) {
          if (mode == 1) hipLaunchKernelGGL((gat::k_place<1, 1>), gp, dim3(64), 0, ctx->stream, A);
          else if (mode == 0) hipLaunchKernelGGL((gat::k_place<1, 0>), gp, dim3(64), 0, ctx->stream, A);
          else hipLaunchKernelGGL((gat::k_place<1, 2>), gp, dim3(64), 0, ctx->stream, A);
        } else {
          // (k_place_pipe: every look-up of every unit in LDS -- the rows prefetched by hand, see GAT_PLACE_LOOP_PIPE)
          const bool pipe = !getenv("GAT_PLACE_NO_PIPE");
          const bool rank_fits = P->max_hist < (uint32_t)gat::kPlaceRankLds;
          if (mode == 1 && pipe) hipLaunchKernelGGL((gat::k_place_pipe<0, 1>), gp, dim3(64), 0, ctx->stream, A);
          else if (mode == 1) hipLaunchKernelGGL((gat::k_place<0, 1>), gp, dim3(64), 0, ctx->stream, A);
          else if (mode == 0 && P->small_tables && pipe) hipLaunchKernelGGL((gat::k_place_pipe<0, 0, 1>), gp, dim3(64), 0, ctx->stream, A);
          else if (mode == 0 && P->small_tables) hipLaunchKernelGGL((gat::k_place<0, 0, 1>), gp, dim3(64), 0, ctx->stream, A);
          else if (mode == 0 && P->max_nws <= 64 && rank_fits && pipe) hipLaunchKernelGGL((gat::k_place_pipe<0, 0, 2>), gp, dim3(64), 0, ctx->stream, A);
          else if (mode == 0 && P->max_nws <= 64) hipLaunchKernelGGL((gat::k_place<0, 0, 2>), gp, dim3(64), 0, ctx->stream, A);
          else if (mode == 0 && rank_fits && pipe) hipLaunchKernelGGL((gat::k_place_pipe<0, 0>), gp, dim3(64), 0, ctx->stream, A);
          else if (mode == 0) hipLaunchKernelGGL((gat::k_place<0, 0>), gp, dim3(64), 0, ctx->stream, A);
          else hipLaunchKernelGGL((gat::k_place<0, 2>), gp, dim3(64), 0, ctx->stream, A);
        }
        HIPCHK(ctx, hipGetLastError());
        if (timed) HIPCHK(ctx, hipEventRecord(ctx->ev_k[1], ctx->stream));
      }
      ctx->k_recorded = timed && smode;
      size_t lds = (size_t)(gat::kMtLdsWords + 2 * (size_t)P->max_unit_cap) * 4;
      // SamplerSegments never holds a list; a SamplerAnnotator list beyond LDS is worked on in the slab (HUGE variant)
      const bool huge = P->sampler != GAT_SAMPLER_SEGMENTS && ((int64_t)lds > ctx->max_lds || getenv("GAT_TEST_HUGE") != nullptr);
      if (huge || P->sampler == GAT_SAMPLER_SEGMENTS) lds = (size_t)gat::kMtLdsWords * 4;
      A.lds_cap = P->max_unit_cap;
      A.big_buckets = 0;
      uint32_t max_work = 0;
      for (int32_t u : P->h_order) max_work = std::max(max_work, P->h_units[u].hist_total);
      A.st2 = nullptr;
      // the split path (k_consolidate / k_merge_big + k_tail + k_finalize in front of k_sampler); lists beyond LDS: old path
      const bool split = P->split_path && smode && !huge;
      unsigned n_long_big = 0;                  // launch positions k_merge_big was given
      bool long_lists = false;                  // k_sampler<BIG>: the code for lists beyond the bucket sorts
      if (!huge && max_work + max_work / 8 > 1024) {
        long_lists = true;
        int nbk = 1024;
        while (nbk < P->max_unit_cap && nbk < 8192) nbk <<= 1;
        // the first consolidation of the long lists by whole workgroups (k_merge_big): its LDS holds one list and the
        // counting sort's histogram; launched over the long units, which come first in the launch order
        const size_t lds_m = (size_t)2 * P->max_unit_cap * 4 + (size_t)(nbk + 1) * 4;
        unsigned n_long = 0;
        for (int32_t u : P->h_order) { const uint32_t w = P->h_units[u].hist_total; if (w + w / 8 > 1024) ++n_long; else break; }
        if (smode && P->sampler != GAT_SAMPLER_SEGMENTS && (int64_t)lds_m + 1024 <= ctx->max_lds && n_long > 0 &&
            !getenv("GAT_NO_MERGE_BIG")) {
          gat::SamplerArgs M = A;
          M.st2 = P->d_st2.p;
          M.big_buckets = nbk;
          M.cum = split ? P->d_cum.p : nullptr;
          n_long_big = n_long;
          const bool tree_m = P->max_nws > gat::kWsTreeMin;
          const void* km = tree_m ? (const void*)gat::k_merge_big<true> : (const void*)gat::k_merge_big<false>;
          HIPCHK(ctx, hipFuncSetAttribute(km, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_m));
          M.n_long = (int32_t)n_long;
          for (size_t c = 0; c + 1 < P->h_class_start.size() && (unsigned)P->h_class_start[c] < n_long; ++c) {
            // one launch per size class: LDS for the class's longest list and its histogram
            const int a0 = P->h_class_start[c], a1 = std::min<int>(P->h_class_start[c + 1], (int)n_long);
            const int ccap = P->h_units[(size_t)P->h_order[(size_t)a0]].slab_cap;
            int cnbk = 1024;
            while (cnbk < ccap && cnbk < 8192) cnbk <<= 1;
            const size_t lds_c = (size_t)2 * ccap * 4 + (size_t)(cnbk + 1) * 4;
            M.a_base = a0; M.a_end = a1; M.lds_cap = ccap; M.big_buckets = cnbk;
            const unsigned cnt = (unsigned)(a1 - a0), gmy = std::min(cnt, 32768u);
            const dim3 gm((unsigned)nb, gmy, (cnt + gmy - 1) / gmy);
            if (tree_m) hipLaunchKernelGGL(gat::k_merge_big<true>, gm, dim3(gat::kMergeThreads), lds_c, ctx->stream, M);
            else hipLaunchKernelGGL(gat::k_merge_big<false>, gm, dim3(gat::kMergeThreads), lds_c, ctx->stream, M);
            HIPCHK(ctx, hipGetLastError());
          }
          A.st2 = P->d_st2.p;                    // (k_sampler reads it for those units only: see n_long below)
          A.n_long = (int32_t)n_long;
          if (!split && P->d_patch.p != nullptr && P->max_nws <= gat::kTailMaxWs && !getenv("GAT_NO_TAIL_BIG")) {
            // the placement rounds behind that consolidation, one stream per lane; k_sampler resumes at the trim
            gat::TailArgs TB;
            TB.S = A;
            TB.cum = nullptr; TB.patch = P->d_patch.p; TB.todo = nullptr; TB.todo_count = nullptr;
            const unsigned gby = std::min(n_long, 32768u);
            hipLaunchKernelGGL(gat::k_tail_big, dim3((unsigned)((nb + 63) / 64), gby, (n_long + gby - 1) / gby), dim3(64), 0,
                               ctx->stream, TB);
            HIPCHK(ctx, hipGetLastError());
            if (!getenv("GAT_NO_RESUME_BIG")) {
              // ... and the rest of the unit -- log inserted, trim, final filter -- with the list where it is
              hipLaunchKernelGGL(gat::k_resume_big, dim3((unsigned)nb, gby, (n_long + gby - 1) / gby), dim3(64), 0, ctx->stream, TB);
              HIPCHK(ctx, hipGetLastError());
            }
            A.tb = reinterpret_cast<const int32_t*>(P->d_patch.p);
            A.skip_stride = (int32_t)(sizeof(gat::TailPatch) / 4);
          }
        } else {
          // no workgroup pass: the wave's own counting sort, scratch behind the segment buffer if it fits
          while (nbk >= 1024 && (int64_t)(lds + (size_t)(nbk + 1) * 4) > ctx->max_lds) nbk >>= 1;
          if (nbk >= 1024) { A.big_buckets = nbk; lds += (size_t)(nbk + 1) * 4; }
        }
      }
      const bool tree = P->max_nws > gat::kWsTreeMin;
      ctx->t_recorded = false;
      P->split_ran = split;
      P->patched_contigs = false;
      P->patched_counts = false;
      if (split) {
        // the split path: first consolidation (wave per unit), the loop's tail (lane per unit), the final list (wave per
        // unit); k_sampler below then only resumes -- from the merged list -- the units k_tail left alone
        gat::TailArgs T;
        T.S = A;
        T.S.st2 = P->d_st2.p;
        T.S.n_long = (int32_t)n_long_big;                           // (whose verdict k_consolidate respects)
        T.S.lds_cap = std::min(P->max_unit_cap, 1280);              // k_consolidate: the lists the wave bucket sorts take
        T.S.slab_final = P->d_fslab.p;
        T.cum = P->d_cum.p;
        T.patch = P->d_patch.p;
        T.todo = P->d_todo.p;
        T.todo_count = P->d_todo_count.p;
        HIPCHK(ctx, hipMemsetAsync(P->d_todo_count.p, 0, 4, ctx->stream));
        const size_t lds_max = (size_t)(gat::kSortScratchWords + 2 * (size_t)T.S.lds_cap) * 4;
        const dim3 gu((unsigned)nb, gy, gz), gt((unsigned)((nb + 63) / 64), gy, gz);
        if (tree) HIPCHK(ctx, hipFuncSetAttribute((const void*)gat::k_consolidate<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_max));
        else HIPCHK(ctx, hipFuncSetAttribute((const void*)gat::k_consolidate<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_max));
        for (size_t c = 0; c + 1 < P->h_class_start.size(); ++c) {
          // one launch per size class, its LDS sized for the class's longest list
          const int a0 = P->h_class_start[c], a1 = P->h_class_start[c + 1];
          const int ccap = std::min(P->h_units[(size_t)P->h_order[(size_t)a0]].slab_cap, T.S.lds_cap);
          gat::TailArgs C = T;
          C.S.a_base = a0; C.S.a_end = a1; C.S.lds_cap = ccap;
          const size_t lds_c = (size_t)(gat::kSortScratchWords + 2 * (size_t)ccap) * 4;
          const unsigned cnt = (unsigned)(a1 - a0), cy = std::min(cnt, 32768u);
          const dim3 gc((unsigned)nb, cy, (cnt + cy - 1) / cy);
          if (tree) hipLaunchKernelGGL(gat::k_consolidate<true>, gc, dim3(64), lds_c, ctx->stream, C);
          else hipLaunchKernelGGL(gat::k_consolidate<false>, gc, dim3(64), lds_c, ctx->stream, C);
          HIPCHK(ctx, hipGetLastError());
        }
        if (timed) HIPCHK(ctx, hipEventRecord(ctx->ev_k[2], ctx->stream));
        hipLaunchKernelGGL(gat::k_tail, gt, dim3(64), 0, ctx->stream, T);
        HIPCHK(ctx, hipGetLastError());
        if (timed) HIPCHK(ctx, hipEventRecord(ctx->ev_t[0], ctx->stream));
        // isochore problems: k_contig re-sorts the units of a contig anyway and takes (merged list, k_tail's record) as
        // it is -- no final unit lists unless somebody asked for them (gat_sample_units)
        P->patched_contigs = P->merge_contigs && P->n_contigs > 0 && !need_unit_lists && !getenv("GAT_CONTIG_FINAL_LISTS");
        P->patched_counts = !P->merge_contigs && records_ok && !need_unit_lists;
        if (!P->patched_contigs && !P->patched_counts) hipLaunchKernelGGL(gat::k_finalize, gu, dim3(64), 0, ctx->stream, T);
        HIPCHK(ctx, hipGetLastError());
        if (timed) { HIPCHK(ctx, hipEventRecord(ctx->ev_t[1], ctx->stream)); ctx->t_recorded = true; }
        A.st2 = P->d_st2.p;
        A.n_long = (int32_t)n_act;
        A.slab_final = P->d_fslab.p;
        A.skip = &P->d_patch.p->state;
        A.skip_stride = (int32_t)(sizeof(gat::TailPatch) / 4);
        A.todo = P->d_todo.p;
        A.todo_count = P->d_todo_count.p;
      } else if (timed) HIPCHK(ctx, hipEventRecord(ctx->ev_k[2], ctx->stream));
      // variant: sampler kind x (long lists: counting-sort scratch) x (workspaces beyond the register loop: search trees)
      int variant = P->sampler == GAT_SAMPLER_SEGMENTS ? (tree ? 5 : 4)
                  : huge ? (tree ? 7 : 6) : (long_lists ? 2 : 0) + (tree ? 1 : 0);
      // short lists only (20 waves of this kernel fit a CU's LDS): the instantiation with registers for 5 waves per SIMD
      if (variant == 0 && (int64_t)lds * 20 <= ctx->max_lds && !getenv("GAT_NO_WPE5")) variant = 8;
      const void* ks = variant == 0 ? (const void*)gat::k_sampler<0, false, false, false>
                     : variant == 1 ? (const void*)gat::k_sampler<0, false, true, false>
                     : variant == 2 ? (const void*)gat::k_sampler<0, true, false, false>
                     : variant == 3 ? (const void*)gat::k_sampler<0, true, true, false>
                     : variant == 4 ? (const void*)gat::k_sampler<1, false, false, false>
                     : variant == 5 ? (const void*)gat::k_sampler<1, false, true, false>
                     : variant == 6 ? (const void*)gat::k_sampler<0, false, false, true>
                     : variant == 7 ? (const void*)gat::k_sampler<0, false, true, true>
                                    : (const void*)gat::k_sampler<0, false, false, false, 5>;
      HIPCHK(ctx, hipFuncSetAttribute(ks, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      auto launch_sampler = [&](const dim3& gs, size_t lds_, const gat::SamplerArgs& K) {
        switch (variant) {
          case 0: hipLaunchKernelGGL((gat::k_sampler<0, false, false, false>), gs, dim3(64), lds_, ctx->stream, K); break;
          case 1: hipLaunchKernelGGL((gat::k_sampler<0, false, true, false>), gs, dim3(64), lds_, ctx->stream, K); break;
          case 2: hipLaunchKernelGGL((gat::k_sampler<0, true, false, false>), gs, dim3(64), lds_, ctx->stream, K); break;
          case 3: hipLaunchKernelGGL((gat::k_sampler<0, true, true, false>), gs, dim3(64), lds_, ctx->stream, K); break;
          case 4: hipLaunchKernelGGL((gat::k_sampler<1, false, false, false>), gs, dim3(64), lds_, ctx->stream, K); break;
          case 5: hipLaunchKernelGGL((gat::k_sampler<1, false, true, false>), gs, dim3(64), lds_, ctx->stream, K); break;
          case 6: hipLaunchKernelGGL((gat::k_sampler<0, false, false, true>), gs, dim3(64), lds_, ctx->stream, K); break;
          case 7: hipLaunchKernelGGL((gat::k_sampler<0, false, true, true>), gs, dim3(64), lds_, ctx->stream, K); break;
          default: hipLaunchKernelGGL((gat::k_sampler<0, false, false, false, 5>), gs, dim3(64), lds_, ctx->stream, K); break;
        }
      };
      const bool list_in_lds = !huge && P->sampler != GAT_SAMPLER_SEGMENTS;
      if (serial_state != nullptr) {
        // the reference's own stream: one wave, every (sample, unit) of the batch in order
        gat::SamplerArgs K = A;
        K.serial_state = serial_state;
        K.unit_pos = P->d_unit_pos.p;
        const int sv = variant == 8 ? 0 : variant;
        const void* kf = sv == 0 ? (const void*)gat::k_serial<0, false, false, false> : sv == 1 ? (const void*)gat::k_serial<0, false, true, false>
                       : sv == 2 ? (const void*)gat::k_serial<0, true, false, false> : sv == 3 ? (const void*)gat::k_serial<0, true, true, false>
                       : sv == 4 ? (const void*)gat::k_serial<1, false, false, false> : sv == 5 ? (const void*)gat::k_serial<1, false, true, false>
                       : sv == 6 ? (const void*)gat::k_serial<0, false, false, true> : (const void*)gat::k_serial<0, false, true, true>;
        HIPCHK(ctx, hipFuncSetAttribute(kf, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        switch (sv) {
          case 0: hipLaunchKernelGGL((gat::k_serial<0, false, false, false>), dim3(1), dim3(64), lds, ctx->stream, K); break;
          case 1: hipLaunchKernelGGL((gat::k_serial<0, false, true, false>), dim3(1), dim3(64), lds, ctx->stream, K); break;
          case 2: hipLaunchKernelGGL((gat::k_serial<0, true, false, false>), dim3(1), dim3(64), lds, ctx->stream, K); break;
          case 3: hipLaunchKernelGGL((gat::k_serial<0, true, true, false>), dim3(1), dim3(64), lds, ctx->stream, K); break;
          case 4: hipLaunchKernelGGL((gat::k_serial<1, false, false, false>), dim3(1), dim3(64), lds, ctx->stream, K); break;
          case 5: hipLaunchKernelGGL((gat::k_serial<1, false, true, false>), dim3(1), dim3(64), lds, ctx->stream, K); break;
          case 6: hipLaunchKernelGGL((gat::k_serial<0, false, false, true>), dim3(1), dim3(64), lds, ctx->stream, K); break;
          default: hipLaunchKernelGGL((gat::k_serial<0, false, true, true>), dim3(1), dim3(64), lds, ctx->stream, K); break;
        }
      } else if (split) {
        launch_sampler(dim3((unsigned)std::min<int64_t>((int64_t)nb * n_act, 8192)), lds, A);      // off the queue
      } else if (list_in_lds && A.big_buckets == 0 && P->h_class_start.size() > 2) {
        // one launch per size class: LDS for the class's longest list
        for (size_t c = 0; c + 1 < P->h_class_start.size(); ++c) {
          const int a0 = P->h_class_start[c], a1 = P->h_class_start[c + 1];
          const int ccap = P->h_units[(size_t)P->h_order[(size_t)a0]].slab_cap;
          gat::SamplerArgs K = A;
          K.a_base = a0; K.a_end = a1;
          const unsigned cnt = (unsigned)(a1 - a0), cy = std::min(cnt, 32768u);
          launch_sampler(dim3((unsigned)nb, cy, (cnt + cy - 1) / cy), (size_t)(gat::kMtLdsWords + 2 * (size_t)ccap) * 4, K);
        }
      } else {
        launch_sampler(dim3((unsigned)nb, gy, gz), lds, A);
      }
      HIPCHK(ctx, hipGetLastError());
      if (timed) HIPCHK(ctx, hipEventRecord(ctx->ev_k[3], ctx->stream));
      skip_ptr = A.skip != nullptr ? A.skip : (A.tb != nullptr ? A.tb : nullptr);     // (n_tail_units: finished by k_tail / carried on by k_tail_big)
      skip_stride = A.skip_stride;
    }
    if (timed) HIPCHK(ctx, hipEventRecord(ctx->ev[1], ctx->stream));
    if (P->merge_contigs && P->n_contigs > 0) {
      gat::ContigArgs B;
      B.contig_unit_off = P->d_contig_unit_off.p; B.contig_units = P->d_contig_units.p; B.units = P->d_units.p;
      B.contig_slab_off = P->d_contig_slab_off.p; B.n_units = P->n_units; B.n_contigs = P->n_contigs;
      B.slab_in = P->final_slab(); B.slab_out = P->d_cslab.p; B.slab_stride = P->slab_stride;
      B.unit_n = P->d_unit_n.p; B.contig_n = P->d_contig_n.p; B.stat = P->d_stat.p;
      B.slab_merged = nullptr; B.unit_pos = P->d_unit_pos.p; B.st2 = nullptr; B.patch = nullptr; B.patch_stride = 0;
      B.ws_stat = P->d_ws_stat.p;
      if (P->split_ran && P->patched_contigs) {
        B.slab_merged = P->d_slab.p;
        B.st2 = P->d_st2.p;
        B.patch = reinterpret_cast<const int32_t*>(P->d_patch.p);
        B.patch_stride = (int32_t)(sizeof(gat::TailPatch) / 4);
      }
      const int need_max = P->h_contig_order.empty() ? 64 : P->h_contig_need[(size_t)P->h_contig_order[0]];
      size_t lds = (size_t)std::max(64, need_max) * 8 + 520 * 4;
      const bool huge_c = (int64_t)lds > ctx->max_lds || getenv("GAT_TEST_HUGE") != nullptr;   // list stays in the output slab
      if (huge_c) lds = 520 * 4;
      const void* kc = huge_c ? (const void*)gat::k_contig<true> : (const void*)gat::k_contig<false>;
      HIPCHK(ctx, hipFuncSetAttribute(kc, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      B.order = P->d_contig_order.p;
      B.flags = P->d_flags.p;
      // one launch per size class: LDS for the class's longest expected list (more waves per CU for the short contigs)
      for (size_t k = 0; k + 1 < P->h_contig_class_start.size(); ++k) {
        const int c0 = P->h_contig_class_start[k], c1 = P->h_contig_class_start[k + 1];
        if (huge_c && k > 0) break;
        B.base = huge_c ? 0 : c0;
        B.count = huge_c ? P->n_contigs : c1 - c0;
        B.lds_cap = huge_c ? 0 : std::max(64, P->h_contig_need[(size_t)P->h_contig_order[(size_t)c0]]);
        const size_t lds_k = huge_c ? lds : (size_t)B.lds_cap * 8 + 520 * 4;
        const unsigned gcy = (unsigned)std::min(B.count, 32768), gcz = ((unsigned)B.count + gcy - 1) / gcy;
        if (huge_c) hipLaunchKernelGGL(gat::k_contig<true>, dim3((unsigned)nb, gcy, gcz), dim3(64), lds_k, ctx->stream, B);
        else hipLaunchKernelGGL(gat::k_contig<false>, dim3((unsigned)nb, gcy, gcz), dim3(64), lds_k, ctx->stream, B);
        HIPCHK(ctx, hipGetLastError());
      }
    }
    if (timed) HIPCHK(ctx, hipEventRecord(ctx->ev[2], ctx->stream));
    if (!P->h_order.empty()) {
      // (behind k_contig: on isochore problems it is k_contig that writes the statistics of the units k_tail finished)
      hipLaunchKernelGGL(gat::k_reduce_stats, dim3(256), dim3(256), 0, ctx->stream, (const uint32_t*)P->d_ws_stat.p,
                         (int64_t)nb * P->n_units, P->d_stat.p, skip_ptr, skip_stride);
      HIPCHK(ctx, hipGetLastError());
    }
    HIPCHK(ctx, hipMemcpyAsync(ctx->h_flags, P->d_flags.p, 4, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(ctx->h_stat, P->d_stat.p, 64, hipMemcpyDeviceToHost, ctx->stream));
    if (defer) return GAT_OK;
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return finish_sampler_batch(ctx, P, nb, st, timed);
  }
}

// the checks and statistics of a sampler batch whose kernels have completed (the stream has been synchronised)
static int finish_sampler_batch(gat_ctx* ctx, gat_problem* P, int64_t nb, gat_stats* st, bool timed) {
  {
    int rc;
    const int32_t flags = *ctx->h_flags;
    const unsigned long long* stat = ctx->h_stat;
    if (flags & (gat::kStatusAssert | gat::kStatusTrimAssert))
      return set_err(ctx, GAT_ERR_ASSERT, "sampler assertion failed on device (flags=%d): %s", flags,
                     (flags & gat::kStatusAssert) ? "sampled list has no overlap with the workspace (gat/Engine.pyx:645)"
                                                  : "trimming more than the total length (gat/SegmentList.pyx:560)");
    if ((flags & gat::kStatusContigLds) && !(flags & gat::kStatusOverflow)) {
      // a contig's lists were longer than expected: LDS for every unit at its capacity from now on, batch repeated
      P->contig_tight = false;
      if (layout_slab(P)) return set_err(ctx, GAT_ERR_CAPACITY, "per-sample slab exceeds 2^31 segments");
      if ((rc = upload_layout(ctx, P))) return rc;
      if (st) st->n_retried += nb * (int64_t)P->h_order.size();
      return kRelayout;
    }
    if (flags & gat::kStatusOverflow) {
      if (P->cap_scale >= 64) return set_err(ctx, GAT_ERR_CAPACITY, "sampler slab overflow even at 64x capacity");
      P->cap_scale *= 2;
      if (layout_slab(P)) return set_err(ctx, GAT_ERR_CAPACITY, "per-sample slab exceeds 2^31 segments after growth");
      if ((rc = upload_layout(ctx, P))) return rc;
      if (st) st->n_retried += nb * (int64_t)P->h_order.size();
      return kRelayout;
    }
#ifdef GAT_DIAG
    if (const char* fn = getenv("GAT_DIAG_OUT")) {
      // shares of a k_sampler work unit's life per phase, summed over the batch (tools/diag_sampler.sh)
      const size_t nd = (size_t)nb * std::max(1, P->n_units) * 8;
      std::vector<unsigned long long> h(nd);
      HIPCHK(ctx, hipMemcpy(h.data(), P->d_diag.p, nd * 8, hipMemcpyDeviceToHost));
      double sum[8] = {0, 0, 0, 0, 0, 0, 0, 0};
      for (size_t i = 0; i < nd; ++i) sum[i & 7] += (double)h[i];
      if (FILE* f = fopen(fn, "a")) {
        fprintf(f, "{\"work_units\": %lld, \"cycles\": {\"prologue\": %.0f, \"sort\": %.0f, \"merge\": %.0f, \"coverage\": %.0f, "
                   "\"fast_paths\": %.0f, \"trim\": %.0f, \"draws_placement\": %.0f, \"final_filter_write\": %.0f}}\n",
                (long long)nb * (long long)P->h_order.size(), sum[0], sum[1], sum[2], sum[3], sum[4], sum[5], sum[6], sum[7]);
        fclose(f);
      }
    }
#endif
    if (st) {
      st->n_placed += (int64_t)stat[0];
      st->n_draws += (int64_t)stat[1];
      st->n_unsuccessful += (int64_t)stat[2];
      st->n_tail_units += (int64_t)stat[3];
      if (P->split_ran && (P->patched_contigs || P->patched_counts)) st->lists_from_records += 1;
      st->n_full_units += (int64_t)stat[4];
      if (timed) {
        float ms = 0;
        HIPCHK(ctx, hipEventElapsedTime(&ms, ctx->ev[0], ctx->ev[1]));
        st->ms_sampler += ms;
        HIPCHK(ctx, hipEventElapsedTime(&ms, ctx->ev[1], ctx->ev[2]));
        st->ms_contig += ms;
        if (ctx->k_recorded && !P->h_order.empty()) {
          HIPCHK(ctx, hipEventElapsedTime(&ms, ctx->ev[0], ctx->ev_k[0]));
          st->ms_rng += ms;
          HIPCHK(ctx, hipEventElapsedTime(&ms, ctx->ev_k[0], ctx->ev_k[1]));
          st->ms_place += ms;
          HIPCHK(ctx, hipEventElapsedTime(&ms, ctx->ev_k[1], ctx->ev_k[2]));
          st->ms_merge += ms;
          HIPCHK(ctx, hipEventElapsedTime(&ms, ctx->ev_k[2], ctx->ev_k[3]));
          st->ms_tail += ms;
          if (ctx->t_recorded) {
            HIPCHK(ctx, hipEventElapsedTime(&ms, ctx->ev_k[2], ctx->ev_t[0]));
            st->ms_ktail += ms;
            HIPCHK(ctx, hipEventElapsedTime(&ms, ctx->ev_t[0], ctx->ev_t[1]));
            st->ms_finalize += ms;
          }
        }
      }
    }
    return GAT_OK;
  }
}

static void fill_count_args(gat_problem* P, gat::CountArgs& A, int64_t nb) {
  A.seg = P->merge_contigs ? P->d_cslab.p : P->final_slab();
  A.seg_stride = P->slab_stride;
  A.c_off = P->d_count_c_off.p;
  A.n_arr = P->merge_contigs ? P->d_contig_n.p : P->d_unit_n.p;
  A.n_stride = P->merge_contigs ? P->n_contigs : P->n_units;
  A.n_index = P->d_count_n_index.p;
  A.cws_nseg = P->d_cws_nseg.p;
  A.n_contigs = P->n_contigs;
  A.n_tracks = P->n_tracks;
  A.n_samples = (int32_t)nb;
  if (P->split_ran && P->patched_counts) {     // no k_finalize ran: k_count_seg<.., PATCH> reads merged lists + records
    A.seg_merged = P->d_slab.p;
    A.unit_pos = P->d_unit_pos.p;
    A.st2 = P->d_st2.p;
    A.patch = reinterpret_cast<const int32_t*>(P->d_patch.p);
    A.patch_stride = (int32_t)(sizeof(gat::TailPatch) / 4);
    A.n_units = P->n_units;
  }
}

static int sample_and_count_impl(gat_ctx* ctx, gat_problem* P, const int32_t* counter_ids, int n_counters,
                                 uint32_t seed, int64_t sample_begin, int64_t sample_end, void* counts_dev,
                                 gat_stats* stats, uint32_t* state_host);

extern "C" int gat_sample_and_count(gat_ctx* ctx, gat_problem* P, const int32_t* counter_ids, int n_counters,
                                    uint32_t seed, int64_t sample_begin, int64_t sample_end, void* counts_dev,
                                    gat_stats* stats) {
  return sample_and_count_impl(ctx, P, counter_ids, n_counters, seed, sample_begin, sample_end, counts_dev, stats, nullptr);
}

extern "C" int gat_sample_and_count_serial(gat_ctx* ctx, gat_problem* P, const int32_t* counter_ids, int n_counters,
                                           uint32_t* mt_state, int64_t n_samples, void* counts_dev, gat_stats* stats) {
  if (!mt_state) return set_err(ctx, GAT_ERR_ARG, "gat_sample_and_count_serial: NULL state");
  if (mt_state[GAT_MT_STATE_WORDS - 1] > 624u) return set_err(ctx, GAT_ERR_ARG, "gat_sample_and_count_serial: position %u > 624", mt_state[GAT_MT_STATE_WORDS - 1]);
  return sample_and_count_impl(ctx, P, counter_ids, n_counters, 0u, 0, n_samples, counts_dev, stats, mt_state);
}

extern "C" void gat_mt19937_seed(uint32_t seed, uint32_t* mt_state) {
  // numpy.random.seed(int): init_genrand (numpy/random/src/mt19937/mt19937.c: mt19937_seed), position = 624
  uint32_t x = seed;
  for (int i = 0; i < 624; ++i) { mt_state[i] = x; x = 1812433253u * (x ^ (x >> 30)) + (uint32_t)(i + 1); }
  mt_state[624] = 624u;
}

static int sample_and_count_impl(gat_ctx* ctx, gat_problem* P, const int32_t* counter_ids, int n_counters,
                                 uint32_t seed, int64_t sample_begin, int64_t sample_end, void* counts_dev,
                                 gat_stats* stats, uint32_t* state_host) {
  if (!ctx || !P || !counts_dev || (n_counters > 0 && !counter_ids)) return set_err(ctx, GAT_ERR_ARG, "gat_sample_and_count: NULL argument");
  if (sample_end < sample_begin) return set_err(ctx, GAT_ERR_ARG, "sample_end < sample_begin");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  Counters C;
  int rc = parse_counters(ctx, counter_ids, n_counters, C);
  if (rc) return rc;
  if (P->sampler == GAT_SAMPLER_SEGMENTS && !P->merge_contigs && n_counters > 0)
    return set_err(ctx, GAT_ERR_ASSERT, "SamplerSegments output is not normalized unless fromIsochores merges it "
                   "(keys without isochores): the counters assert (gat/SegmentList.pyx:1031)");
  gat_stats local;
  memset(&local, 0, sizeof(local));
  const int64_t S = sample_end - sample_begin;
  HIPCHK(ctx, hipEventRecord(ctx->ev[3], ctx->stream));
  uint32_t* d_state = nullptr;
  if (state_host != nullptr) {
    // the run's one stream: its state lives on the device over the batches (a copy restores it when a batch is repeated)
    if (P->d_serial.n < 2 * (size_t)GAT_MT_STATE_WORDS) HIPCHK(ctx, P->d_serial.alloc(2 * (size_t)GAT_MT_STATE_WORDS));
    d_state = P->d_serial.p;
    HIPCHK(ctx, hipMemcpyAsync(d_state, state_host, GAT_MT_STATE_WORDS * 4, hipMemcpyHostToDevice, ctx->stream));
  }
  int64_t done = 0;
  while (done < S) {
    if ((rc = ensure_scratch(ctx, P, S - done))) return rc;
    const int64_t nb = std::min<int64_t>(P->batch, S - done);
    if (d_state != nullptr)
      HIPCHK(ctx, hipMemcpyAsync(d_state + GAT_MT_STATE_WORDS, d_state, GAT_MT_STATE_WORDS * 4, hipMemcpyDeviceToDevice, ctx->stream));
    int swap_capx = 0;
    if (P->swap_capx) {
      const int capx = P->merge_contigs ? P->max_contig_cap : P->max_unit_cap;
      if ((int64_t)3 * capx * 4 + (8192 + 1) * 4 <= (int64_t)ctx->max_lds - 1024) swap_capx = capx;
    }
    // counts alone, all of them k_count_seg's: it reads the units as k_tail left them (no final lists are written)
    const int route = count_route(ctx, P->annos, C, P->n_contigs, P->n_tracks, swap_capx);
    const bool records_ok = !C.any_anno && !getenv("GAT_COUNT_FINAL_LISTS") &&
                            (route == GAT_COUNT_KERNEL_SEG || route == GAT_COUNT_KERNEL_MERGED);
    if ((rc = run_sampler_batch(ctx, P, seed, sample_begin + done, nb, &local, true, false, true, records_ok, d_state))) return rc;   // (enqueued only)
    gat::CountArgs A;
    memset(&A, 0, sizeof(A));
    fill_count_args(P, A, nb);
    A.out = (int64_t*)counts_dev;
    A.out_stride = S;
    A.out_begin = done;
    HIPCHK(ctx, hipEventRecord(ctx->ev_cnt[0], ctx->stream));
    ctx->main_recorded = false;
    ctx->count_kernel = GAT_COUNT_KERNEL_NONE;
    if ((rc = launch_count(ctx, P->annos, C, A, P->d_part, swap_capx,
                           P->merge_contigs ? P->max_contig_cap : P->max_unit_cap))) return rc;
    HIPCHK(ctx, hipEventRecord(ctx->ev_cnt[1], ctx->stream));
    // ONE synchronisation per batch: the sampler's status word is read behind the count kernels (which ran on whatever
    // an overflowed unit left -- harmless, the batch is redone with doubled regions)
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    if ((rc = finish_sampler_batch(ctx, P, nb, &local, true)) == kRelayout) {
      if (d_state != nullptr)        // (the repeated batch draws from where this one began)
        HIPCHK(ctx, hipMemcpyAsync(d_state, d_state + GAT_MT_STATE_WORDS, GAT_MT_STATE_WORDS * 4, hipMemcpyDeviceToDevice, ctx->stream));
      continue;
    }
    if (rc) return rc;
    float ms = 0;
    HIPCHK(ctx, hipEventElapsedTime(&ms, ctx->ev_cnt[0], ctx->ev_cnt[1]));
    local.ms_count += ms;
    if (ctx->main_recorded) {
      HIPCHK(ctx, hipEventElapsedTime(&ms, ctx->ev_main[0], ctx->ev_main[1]));
      local.ms_count_main += ms;
    }
    local.count_kernel = ctx->count_kernel;
    done += nb;
  }
  HIPCHK(ctx, hipEventRecord(ctx->ev[4], ctx->stream));
  if (d_state != nullptr) HIPCHK(ctx, hipMemcpyAsync(state_host, d_state, GAT_MT_STATE_WORDS * 4, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  float ms = 0;
  HIPCHK(ctx, hipEventElapsedTime(&ms, ctx->ev[3], ctx->ev[4]));
  local.ms_total = ms;
  if (stats) *stats = local;
  return GAT_OK;
}

// gat_sample / gat_sample_units: the lists of every (sample, contig) after fromIsochores, or of every (sample, unit)
// as the sampler returned them
static int sample_lists(gat_ctx* ctx, gat_problem* P, uint32_t seed, int64_t sample_begin, int64_t sample_end,
                        gat_segment* out_host, int64_t cap, int64_t* off_host, gat_stats* stats, bool unit_level);

extern "C" int gat_sample(gat_ctx* ctx, gat_problem* P, uint32_t seed, int64_t sample_begin, int64_t sample_end,
                          gat_segment* out_host, int64_t cap, int64_t* off_host, gat_stats* stats) {
  return sample_lists(ctx, P, seed, sample_begin, sample_end, out_host, cap, off_host, stats, false);
}

extern "C" int gat_sample_units(gat_ctx* ctx, gat_problem* P, uint32_t seed, int64_t sample_begin, int64_t sample_end,
                                gat_segment* out_host, int64_t cap, int64_t* off_host, gat_stats* stats) {
  return sample_lists(ctx, P, seed, sample_begin, sample_end, out_host, cap, off_host, stats, true);
}

static int sample_lists(gat_ctx* ctx, gat_problem* P, uint32_t seed, int64_t sample_begin, int64_t sample_end,
                        gat_segment* out_host, int64_t cap, int64_t* off_host, gat_stats* stats, bool unit_level) {
  if (!ctx || !P || !off_host) return set_err(ctx, GAT_ERR_ARG, "gat_sample: NULL argument");
  if (sample_end < sample_begin) return set_err(ctx, GAT_ERR_ARG, "sample_end < sample_begin");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  gat_stats local;
  memset(&local, 0, sizeof(local));
  const int64_t S = sample_end - sample_begin;
  const int C = unit_level ? P->n_units : P->n_contigs;
  int rc;
  int64_t done = 0, total = 0;
  bool overflow = false;
  off_host[0] = 0;
  std::vector<uint2> h_slab;
  std::vector<int32_t> h_n;
  while (done < S) {
    if ((rc = ensure_scratch(ctx, P, S - done))) return rc;
    const int64_t nb = std::min<int64_t>(P->batch, S - done);
    if ((rc = run_sampler_batch(ctx, P, seed, sample_begin + done, nb, &local, true, unit_level)) == kRelayout) continue;
    if (rc) return rc;
    const bool from_contigs = P->merge_contigs && !unit_level;
    const uint2* src = from_contigs ? P->d_cslab.p : P->final_slab();
    const int32_t* nsrc = from_contigs ? P->d_contig_n.p : P->d_unit_n.p;
    const int nstride = from_contigs ? P->n_contigs : P->n_units;
    h_slab.resize((size_t)(nb * P->slab_stride));
    h_n.resize((size_t)(nb * std::max(1, nstride)));
    HIPCHK(ctx, hipMemcpyAsync(h_slab.data(), src, h_slab.size() * sizeof(uint2), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(h_n.data(), nsrc, h_n.size() * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    for (int64_t i = 0; i < nb; ++i) {
      for (int c = 0; c < C; ++c) {
        // (units that computeSample skips keep n == 0: their entries are never written and were zeroed at allocation)
        const int32_t n = unit_level ? h_n[(size_t)(i * nstride + c)] : h_n[(size_t)(i * nstride + P->h_count_n_index[c])];
        const int64_t soff = unit_level ? (int64_t)P->h_units[(size_t)c].slab_off : (int64_t)P->h_count_c_off[c];
        if (!overflow && out_host && total + n <= cap)
          memcpy(out_host + total, h_slab.data() + i * P->slab_stride + soff, (size_t)n * sizeof(uint2));
        else if (n > 0) overflow = true;
        total += n;
        off_host[(done + i) * C + c + 1] = total;
      }
    }
    done += nb;
  }
  local.n_sampled_segments = total;
  if (stats) *stats = local;
  if (overflow) return set_err(ctx, GAT_ERR_CAPACITY, "gat_sample: output needs %lld segments, cap is %lld", (long long)total, (long long)cap);
  return GAT_OK;
}

extern "C" int gat_count_lists(gat_ctx* ctx, const int32_t* counter_ids, int n_counters,
                               const gat_segment* lists, const int64_t* list_off, int64_t n_lists,
                               const gat_segment* annos, const int64_t* anno_off, int32_t n_tracks,
                               const int64_t* ws_nseg, int32_t n_groups, void* counts_host) {
  if (!ctx || !list_off || !anno_off || !counts_host || (n_groups > 0 && !ws_nseg))
    return set_err(ctx, GAT_ERR_ARG, "gat_count_lists: NULL argument");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  Counters C;
  int rc = parse_counters(ctx, counter_ids, n_counters, C);
  if (rc) return rc;
  AnnoDev A;
  if ((rc = build_annos(ctx, A, annos, anno_off, (int64_t)n_tracks * n_groups, n_groups))) return rc;
  for (int64_t l = 0; l < n_lists * n_groups; ++l)
    if ((rc = check_list(ctx, lists + list_off[l], list_off[l + 1] - list_off[l], "segment", l))) return rc;
  const int64_t total = list_off[n_lists * n_groups];
  std::vector<uint2> h_seg((size_t)total);
  for (int64_t i = 0; i < total; ++i) h_seg[(size_t)i] = make_uint2(lists[i].start, lists[i].end);
  DevBuf<uint2> d_seg;
  DevBuf<int32_t> d_c_off, d_n, d_index;
  DevBuf<int64_t> d_nseg, d_out;
  DevBuf<uint32_t> d_part;
  HIPCHK(ctx, d_seg.upload(h_seg, ctx->stream));
  std::vector<int64_t> h_nseg(ws_nseg, ws_nseg + n_groups);
  HIPCHK(ctx, d_nseg.upload(h_nseg, ctx->stream));
  const size_t nslots = (size_t)n_counters * n_tracks * n_lists;
  HIPCHK(ctx, d_out.alloc(nslots));
  HIPCHK(ctx, hipMemsetAsync(d_out.p, 0, std::max<size_t>(1, nslots) * 8, ctx->stream));
  std::vector<int32_t> h_index((size_t)n_groups);
  for (int g = 0; g < n_groups; ++g) h_index[(size_t)g] = g;
  HIPCHK(ctx, d_index.upload(h_index, ctx->stream));
  for (int64_t l = 0; l < n_lists; ++l) {
    // one launch per list: group offsets differ from list to list
    std::vector<int32_t> h_c_off((size_t)n_groups), h_n((size_t)n_groups);
    const int64_t base = list_off[l * n_groups];
    for (int g = 0; g < n_groups; ++g) {
      h_c_off[(size_t)g] = (int32_t)(list_off[l * n_groups + g] - base);
      h_n[(size_t)g] = (int32_t)(list_off[l * n_groups + g + 1] - list_off[l * n_groups + g]);
    }
    HIPCHK(ctx, d_c_off.upload(h_c_off, ctx->stream));
    HIPCHK(ctx, d_n.upload(h_n, ctx->stream));
    gat::CountArgs K;
    memset(&K, 0, sizeof(K));
    K.seg = d_seg.p + base; K.seg_stride = 0; K.c_off = d_c_off.p;
    K.n_arr = d_n.p; K.n_stride = 0; K.n_index = d_index.p;
    K.cws_nseg = d_nseg.p; K.n_contigs = n_groups; K.n_tracks = n_tracks; K.n_samples = 1;
    K.out = d_out.p; K.out_stride = n_lists; K.out_begin = l;
    int32_t longest = 0;
    for (int g = 0; g < n_groups; ++g) longest = std::max(longest, h_n[(size_t)g]);
    if ((rc = launch_count(ctx, A, C, K, d_part, 0, longest))) return rc;
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  }
  HIPCHK(ctx, hipMemcpyAsync(counts_host, d_out.p, nslots * 8, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  return GAT_OK;
}

// ------------------------------------------------------------------------------------------
// The one collective of the path: all-gather of the per-rank count blocks over xGMI (RCCL), replacing the reference's
// pool result collation (gat/__init__.py:681-700, :770-774).  RCCL is loaded lazily so that single-GPU hosts need not
// have it (and so that a process that already holds torch's copy does not get a second one by linking).
struct RcclApi {
  void* handle = nullptr;
  decltype(&ncclGetUniqueId) get_unique_id = nullptr;
  decltype(&ncclCommInitRank) comm_init_rank = nullptr;
  decltype(&ncclCommDestroy) comm_destroy = nullptr;
  decltype(&ncclAllGather) all_gather = nullptr;
  decltype(&ncclGetErrorString) error_string = nullptr;
};
static RcclApi* rccl_api() {
  static RcclApi api;
  static bool tried = false;
  if (!tried) {
    tried = true;
    for (const char* name : {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"}) {
      api.handle = dlopen(name, RTLD_NOW | RTLD_LOCAL);
      if (api.handle) break;
    }
    if (api.handle) {
      api.get_unique_id = (decltype(api.get_unique_id))dlsym(api.handle, "ncclGetUniqueId");
      api.comm_init_rank = (decltype(api.comm_init_rank))dlsym(api.handle, "ncclCommInitRank");
      api.comm_destroy = (decltype(api.comm_destroy))dlsym(api.handle, "ncclCommDestroy");
      api.all_gather = (decltype(api.all_gather))dlsym(api.handle, "ncclAllGather");
      api.error_string = (decltype(api.error_string))dlsym(api.handle, "ncclGetErrorString");
    }
  }
  const bool ok = api.handle && api.get_unique_id && api.comm_init_rank && api.comm_destroy && api.all_gather;
  return ok ? &api : nullptr;
}

struct gat_comm {
  ncclComm_t comm = nullptr;
  int n_ranks = 1, rank = 0;
};

static_assert(sizeof(ncclUniqueId) == GAT_COMM_ID_BYTES, "GAT_COMM_ID_BYTES");

extern "C" int gat_comm_unique_id(void* id_out) {
  if (!id_out) return set_err(nullptr, GAT_ERR_ARG, "gat_comm_unique_id: NULL argument");
  RcclApi* R = rccl_api();
  if (!R) return set_err(nullptr, GAT_ERR_DEVICE, "RCCL (librccl.so) is not available: %s", dlerror() ? dlerror() : "symbols missing");
  ncclUniqueId id;
  const ncclResult_t rc = R->get_unique_id(&id);
  if (rc != ncclSuccess) return set_err(nullptr, GAT_ERR_DEVICE, "ncclGetUniqueId: %s", R->error_string ? R->error_string(rc) : "error");
  memcpy(id_out, &id, sizeof(id));
  return GAT_OK;
}

extern "C" int gat_comm_create(gat_ctx* ctx, gat_comm** out, int n_ranks, int rank, const void* id) {
  if (!ctx || !out || !id || n_ranks < 1 || rank < 0 || rank >= n_ranks) return set_err(ctx, GAT_ERR_ARG, "gat_comm_create: bad argument");
  *out = nullptr;
  RcclApi* R = rccl_api();
  if (!R) return set_err(ctx, GAT_ERR_DEVICE, "RCCL (librccl.so) is not available");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  ncclUniqueId uid;
  memcpy(&uid, id, sizeof(uid));
  std::unique_ptr<gat_comm> C(new gat_comm());
  C->n_ranks = n_ranks;
  C->rank = rank;
  const ncclResult_t rc = R->comm_init_rank(&C->comm, n_ranks, uid, rank);
  if (rc != ncclSuccess) return set_err(ctx, GAT_ERR_DEVICE, "ncclCommInitRank: %s", R->error_string ? R->error_string(rc) : "error");
  *out = C.release();
  return GAT_OK;
}

extern "C" void gat_comm_destroy(gat_comm* comm) {
  if (!comm) return;
  RcclApi* R = rccl_api();
  if (R && comm->comm) (void)R->comm_destroy(comm->comm);
  delete comm;
}

extern "C" int gat_allgather_counts(gat_ctx* ctx, gat_comm* comm, const void* send_dev, void* recv_dev, int64_t n_slots) {
  if (!ctx || !comm || !send_dev || !recv_dev || n_slots < 0) return set_err(ctx, GAT_ERR_ARG, "gat_allgather_counts: bad argument");
  RcclApi* R = rccl_api();
  if (!R) return set_err(ctx, GAT_ERR_DEVICE, "RCCL (librccl.so) is not available");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  const ncclResult_t rc = R->all_gather(send_dev, recv_dev, (size_t)n_slots, ncclInt64, comm->comm, ctx->stream);
  if (rc != ncclSuccess) return set_err(ctx, GAT_ERR_DEVICE, "ncclAllGather: %s", R->error_string ? R->error_string(rc) : "error");
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  return GAT_OK;
}

// ------------------------------------------------------------------------------------------
// numpy's pairwise summation of a block of `n` elements starting at `off` (numpy/_core/src/umath/loops_utils.h.src,
// pairwise_sum): the blocks of at most 128 elements and the order in which their sums are added
static void np_pairwise_plan(int off, int n, std::vector<int32_t>& leaf_off, std::vector<int32_t>& leaf_len, std::vector<int32_t>& prog) {
  if (n <= gat::kNpBlock) {
    prog.push_back((int32_t)leaf_off.size());
    leaf_off.push_back(off);
    leaf_len.push_back(n);
    return;
  }
  int n2 = n / 2;
  n2 -= n2 % 8;
  np_pairwise_plan(off, n2, leaf_off, leaf_len, prog);
  np_pairwise_plan(off + n2, n - n2, leaf_off, leaf_len, prog);
  prog.push_back(-1);
}

extern "C" int gat_null_stats(gat_ctx* ctx, const void* counts_dev, int64_t n_rows, int64_t n_samples,
                              const uint8_t* is_double_host, const double* vals_host, int64_t lo_index, int64_t hi_index,
                              double* out_host) {
  if (!ctx || !counts_dev || !is_double_host || !vals_host || !out_host) return set_err(ctx, GAT_ERR_ARG, "gat_null_stats: NULL argument");
  if (n_rows <= 0) return GAT_OK;
  if (n_samples < 1 || n_samples >= ((int64_t)1 << 31) || lo_index < 0 || hi_index < 0 || lo_index >= n_samples || hi_index >= n_samples)
    return set_err(ctx, GAT_ERR_ARG, "gat_null_stats: bad sample count / positions");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  std::vector<int32_t> leaf_off, leaf_len, prog;
  const int rest = (int)(n_samples % gat::kNpChunk);
  if (rest > 0) np_pairwise_plan(0, rest, leaf_off, leaf_len, prog);
  DevBuf<int32_t> d_off, d_len, d_prog;
  DevBuf<uint8_t> d_dbl;
  DevBuf<double> d_vals, d_out;
  if (leaf_off.empty()) { leaf_off.push_back(0); leaf_len.push_back(0); prog.push_back(0); }
  HIPCHK(ctx, d_off.upload(leaf_off, ctx->stream));
  HIPCHK(ctx, d_len.upload(leaf_len, ctx->stream));
  HIPCHK(ctx, d_prog.upload(prog, ctx->stream));
  HIPCHK(ctx, d_dbl.upload(std::vector<uint8_t>(is_double_host, is_double_host + n_rows), ctx->stream));
  HIPCHK(ctx, d_vals.upload(std::vector<double>(vals_host, vals_host + n_rows), ctx->stream));
  HIPCHK(ctx, d_out.alloc((size_t)n_rows * 8));
  gat::StatsArgs A;
  A.counts = (const int64_t*)counts_dev; A.row_stride = n_samples; A.n_rows = (int32_t)n_rows; A.S = (int32_t)n_samples;
  A.is_double = d_dbl.p; A.vals = d_vals.p; A.out = d_out.p; A.lo_i = (int32_t)lo_index; A.hi_i = (int32_t)hi_index;
  A.leaf_off = d_off.p; A.leaf_len = d_len.p; A.n_leaves = rest > 0 ? (int32_t)leaf_off.size() : 0;
  A.prog = d_prog.p; A.n_prog = rest > 0 ? (int32_t)prog.size() : 0;
  const size_t lds = (size_t)(n_samples / gat::kNpChunk + 1 + leaf_off.size()) * 8;
  if ((int64_t)lds > ctx->max_lds - 4096) return set_err(ctx, GAT_ERR_CAPACITY, "gat_null_stats: %lld samples per row", (long long)n_samples);
  HIPCHK(ctx, hipFuncSetAttribute((const void*)gat::k_null_stats, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  for (int64_t r0 = 0; r0 < n_rows; r0 += 1 << 20) {               // (grid x: rows)
    gat::StatsArgs B = A;
    const int64_t nr = std::min<int64_t>(n_rows - r0, 1 << 20);
    B.counts = A.counts + r0 * n_samples; B.is_double = A.is_double + r0; B.vals = A.vals + r0; B.out = A.out + r0 * 8; B.n_rows = (int32_t)nr;
    hipLaunchKernelGGL(gat::k_null_stats, dim3((unsigned)nr), dim3(gat::kStatsThreads), lds, ctx->stream, B);
    HIPCHK(ctx, hipGetLastError());
  }
  HIPCHK(ctx, hipMemcpyAsync(out_host, d_out.p, (size_t)n_rows * 64, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  return GAT_OK;
}
