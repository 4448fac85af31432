// gat_host.h -- what the host translation units of libgat_mi355.so share: the context, device buffers, the
// annotation tables and the problem record.  gat_prep.hip (problem creation: everything the reference does once per
// (segments, workspace) pair before sampling, gat/Engine.pyx:543-565, and the look-up tables of the count kernels;
// launches nothing) and gat_mi355.hip (scratch, kernel launches, the batch seam) include it.
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "../../include/gat_mi355.h"
#include "gat_types.h"

namespace gat { struct TailPatch; }

using gat::UnitDev;

extern thread_local std::string g_last_error;

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

inline int set_err(gat_ctx* ctx, int code, const char* fmt, ...) {
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

int check_list(gat_ctx* ctx, const gat_segment* s, int64_t n, const char* what, int64_t idx);

// host threads for the per-list / per-contig preparation of gat_problem_create (GAT_HOST_THREADS, default min(16, cores))
template <typename F>
inline void parallel_for(int64_t n, F body) {
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

// gat_prep.hip
int build_annos(gat_ctx* ctx, AnnoDev& A, const gat_segment* annos, const int64_t* anno_off, int64_t n_lists, int32_t n_groups);
int layout_slab(gat_problem* P);
int upload_layout(gat_ctx* ctx, gat_problem* P);
