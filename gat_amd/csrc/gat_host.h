// gat_host.h -- what the host translation units of libgat_mi355.so share: the context, device buffers, the
// annotation tables and the problem record.  gat_prep.hip (problem creation: everything the reference does once per
// (segments, workspace) pair before sampling, gat/Engine.pyx:543-565, and the look-up tables of the count kernels;
// launches nothing) and gat_mi355.hip (scratch, kernel launches, the batch seam) include it.
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <memory>
#include <atomic>
#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/gat_mi355.h"
#include "gat_types.h"

namespace gat { struct TailPatch; }

using gat::UnitDev;

extern thread_local std::string g_last_error;

// What a call of the batch seam keeps while its batches are in flight (gat_sample_and_count_enqueue ... gat_wait): pinned
// words for the batches' status / statistics and the events that time the call.  Pinned allocations change the device's
// page tables (see staged_h2d), so the blocks belong to the context and are lent to a problem for the length of a call.
constexpr int kMaxInflight = 8;               // batches of one call enqueued before the first status word is read
struct CallBlock {
  unsigned long long* h_stat = nullptr;       // pinned, kMaxInflight x 16 words: per batch the statistics, the status word in word 8
  unsigned long long* h_mstat = nullptr;      // pinned, 512 words: k_count_merged's traffic counters of the call
  hipEvent_t ev_begin = nullptr, ev_end = nullptr;
  hipEvent_t ev_main[kMaxInflight][2] = {};   // around the dominant count kernel of every batch in flight
};
struct CallState {
  bool active = false;
  CallBlock* blk = nullptr;
  int32_t ids[GAT_NUM_COUNTERS] = {0, 0, 0, 0, 0, 0};
  int n_counters = 0;
  uint32_t seed = 0;
  int64_t begin = 0, S = 0;                   // samples [begin, begin + S)
  void* counts_dev = nullptr;
  uint32_t* state_host = nullptr;             // gat_sample_and_count_serial: the caller's MT19937 state
  int64_t done = 0;                           // samples whose batches have completed and passed their checks
  int64_t enq = 0;                            // samples enqueued (>= done)
  int n_flight = 0;                           // batches enqueued and not yet checked
  int64_t nb[kMaxInflight] = {};
  int count_kernel[kMaxInflight] = {};
  bool main_rec[kMaxInflight] = {};
  bool timed = false, mstat_on = false;
  int times_state = 0;             // gat_stats::kernel_times of the call
  bool end_recorded = false;                  // ev_end is on the stream behind the call's last batch (gat_wait waits for IT)
  bool count_pending = false;                 // the newest batch's sampler kernels are enqueued, its count kernels wait for the
                                              // annotation tables (an asynchronous build)
  gat_stats local;
};

struct gat_ctx {
  int refs = 1;                    // the handle + one per live problem: gat_ctx_destroy frees when the last one is gone
  bool closed = false;             // gat_ctx_destroy was called (problems still alive)
  std::vector<CallBlock*> call_blocks;          // idle blocks (see CallBlock)
  gat_ctx* aux_ctx = nullptr;                   // stream + staging buffer of gat_count_lists: observed counts do not queue behind samples in flight
  gat_ctx* build_ctx = nullptr;                 // stream + staging buffer of asynchronous annotation builds (made at the first one)
  struct gat_annotations* building = nullptr;   // ... and the object whose build has them (one at a time)
  int device = 0;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  size_t stage_used = 0;           // bytes at the start of h_stage that copies in flight read from (stage_push_h2d)
  bool kernel_times = false;       // gat_ctx_set_kernel_times: events behind the sampler's kernels, their times in gat_stats
  hipEvent_t ev[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
  hipEvent_t ev_main[2] = {nullptr, nullptr};   // around the dominant count kernel alone (gat_count_lists; the batch seam has its own: CallBlock)
  hipEvent_t ev_k[4] = {nullptr, nullptr, nullptr, nullptr};   // behind k_rng, k_place, k_merge_big, k_sampler
  bool k_recorded = false;
  hipEvent_t ev_t[2] = {nullptr, nullptr};      // split path: behind k_tail, k_finalize
  bool t_recorded = false;
  hipEvent_t ev_cnt[2] = {nullptr, nullptr};    // around the count phase
  const void* timed_owner = nullptr;            // the problem whose call in flight owns the per-kernel events above (they exist once
                                                // per context: a second call enqueued meanwhile -- run() keeps two problems' calls
                                                // in flight -- runs untimed instead of re-recording them under the first: ADVICE r4)
  // status word and statistics of a sampler batch, copied behind its kernels and read after the batch's ONE synchronisation
  std::vector<std::pair<void*, size_t>> user_allocs;   // gat_dev_alloc'ed blocks and their sizes (back to the pool at gat_dev_free)
  void* h_stage = nullptr;                      // pinned staging buffer of gat_memcpy_d2h (grows; pageable targets are filled from it)
  size_t h_stage_bytes = 0;
  int32_t* h_flags = nullptr;                   // pinned
  unsigned long long* h_stat = nullptr;         // pinned, 16 words: statistics and status word of a gat_sample batch
  std::string err;
  int max_lds = 65536;
  // gat_ctx_set_option: the context's own values of the tuning / testing knobs (GAT_*), in front of the process's -- the
  // environment as it was when the library was first asked (gat_opt); aux_ctx / build_ctx read their owner's
  std::map<std::string, std::string> options;
  mutable std::mutex options_mutex;
  const gat_ctx* options_owner = nullptr;
};

// The knobs are no longer read from the environment where they act: gat_opt answers from the context's
// options (gat_ctx_set_option), else from a snapshot of the process's GAT_* variables taken ONCE -- no getenv on a call's path,
// and two host threads with a context each no longer share what a test sets.  nullptr: not set (or set to the empty string).
const char* gat_opt(const gat_ctx* ctx, const char* key);

// k_count_seg stages a track tile's lists in LDS: a list of max_m intervals takes max_m + 4 entries (one in front of its first
// interval, three sentinels behind the last: segs_vs_pairs).  ONE condition for the launch (launch_count) and for the build
// (build_annos: lists beyond it get the merged index) -- they had drifted apart by three entries
inline int64_t count_lds_entries(const gat_ctx* ctx) { const char* e = gat_opt(ctx, "GAT_COUNT_LDS_ENTRIES"); return e ? atoll(e) : 1024; }
inline bool count_lists_staged(const gat_ctx* ctx, int64_t max_m) { return max_m > 0 && max_m + 4 <= count_lds_entries(ctx); }

void ctx_release(gat_ctx* ctx);     // gat_mi355.hip: drops one reference, frees the context with the last

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

// Every copy between host and device goes through the context's pinned buffer, in pieces of at most kStagePiece bytes:
// a copy from or to pageable memory makes the runtime pin and unpin the caller's pages around it, and every such change
// of the device's page tables stalled the next operation on the device by 25-30 ms once gigabytes were mapped
// (gat_amd.run() on config 3: the memsets behind gat_problem_create, the read-back behind gat_null_stats).  Synchronous:
// both return when the bytes have arrived.
constexpr size_t kStagePiece = (size_t)64 << 20;
// Small uploads (the dozens of offset and table arrays of a problem) do not each wait for their copy: they are packed into
// the staging buffer one behind the other (stage_push_h2d) and the stream is synchronised once, when the buffer is wanted
// for something else (ctx_stage) or the call that pushed them ends (stage_flush): 0.96 -> ~0.4 ms of config 2's
// gat_problem_create.
constexpr size_t kStagePushMax = (size_t)1 << 20, kStagePushArea = (size_t)8 << 20;
inline hipError_t stage_flush(gat_ctx* ctx) {
  if (ctx->stage_used == 0) return hipSuccess;
  ctx->stage_used = 0;
  return hipStreamSynchronize(ctx->stream);
}
inline hipError_t ctx_stage(gat_ctx* ctx, size_t bytes) {
  { const hipError_t e = stage_flush(ctx); if (e != hipSuccess) return e; }      // (copies in flight out of the buffer)
  if (ctx->h_stage_bytes >= bytes) return hipSuccess;
  if (ctx->h_stage) (void)hipHostFree(ctx->h_stage);
  ctx->h_stage = nullptr;
  ctx->h_stage_bytes = 0;
  size_t want = (size_t)1 << 20;
  while (want < bytes) want <<= 1;
  hipError_t e = hipHostMalloc(&ctx->h_stage, want, hipHostMallocDefault);
  if (e == hipSuccess) ctx->h_stage_bytes = want;
  return e;
}
void parallel_copy(void* dst, const void* src, size_t bytes);      // memcpy, on the host threads from a few megabytes up (gat_prep.hip)
inline hipError_t staged_h2d(gat_ctx* ctx, void* dst_dev, const void* src_host, size_t bytes);
inline hipError_t stage_push_h2d(gat_ctx* ctx, void* dst_dev, const void* src_host, size_t bytes) {
  if (bytes == 0) return hipSuccess;
  if (bytes > kStagePushMax) return staged_h2d(ctx, dst_dev, src_host, bytes);
  size_t off = (ctx->stage_used + 255) & ~(size_t)255;
  if (ctx->stage_used == 0 || off + bytes > ctx->h_stage_bytes) {
    hipError_t e = ctx_stage(ctx, kStagePushArea);                 // (flushes what is in flight; never shrinks the buffer)
    if (e != hipSuccess) return e;
    off = 0;
  }
  memcpy((char*)ctx->h_stage + off, src_host, bytes);
  const hipError_t e = hipMemcpyAsync(dst_dev, (char*)ctx->h_stage + off, bytes, hipMemcpyHostToDevice, ctx->stream);
  ctx->stage_used = off + bytes;
  return e;
}
inline hipError_t staged_h2d(gat_ctx* ctx, void* dst_dev, const void* src_host, size_t bytes) {
  for (size_t o = 0; o < bytes; o += kStagePiece) {
    const size_t n = std::min(kStagePiece, bytes - o);
    hipError_t e = ctx_stage(ctx, n);
    if (e != hipSuccess) return e;
    parallel_copy(ctx->h_stage, (const char*)src_host + o, n);
    if ((e = hipMemcpyAsync((char*)dst_dev + o, ctx->h_stage, n, hipMemcpyHostToDevice, ctx->stream)) != hipSuccess) return e;
    if ((e = hipStreamSynchronize(ctx->stream)) != hipSuccess) return e;
  }
  return hipSuccess;
}
inline hipError_t staged_d2h(gat_ctx* ctx, void* dst_host, const void* src_dev, size_t bytes) {
  for (size_t o = 0; o < bytes; o += kStagePiece) {
    const size_t n = std::min(kStagePiece, bytes - o);
    hipError_t e = ctx_stage(ctx, n);
    if (e != hipSuccess) return e;
    if ((e = hipMemcpyAsync(ctx->h_stage, (const char*)src_dev + o, n, hipMemcpyDeviceToHost, ctx->stream)) != hipSuccess) return e;
    if ((e = hipStreamSynchronize(ctx->stream)) != hipSuccess) return e;
    memcpy((char*)dst_host + o, ctx->h_stage, n);
  }
  return hipSuccess;
}

// Device memory of the size of a problem's scratch (gigabytes) costs milliseconds to map and to unmap, and hipFree waits
// for the device: blocks of a megabyte and more go back to a process-wide pool per device instead and are handed out again
// to the next request they fit (a host that creates one problem per segment track, or per run, asks for the same sizes
// again and again).  gat_prep.hip.
hipError_t dev_pool_alloc(void** out, size_t bytes);
void dev_pool_free(void* p, size_t bytes);
size_t dev_pool_held();            // bytes the pool of the current device holds back (free memory as far as a problem cares)

template <typename T>
struct DevBuf {
  T* p = nullptr;
  size_t n = 0;
  size_t bytes = 0;                 // (kept so that a buffer of a type this translation unit only knows by name can be released)
  ~DevBuf() { release(); }
  void release() { if (p) { dev_pool_free(p, bytes); p = nullptr; n = 0; bytes = 0; } }
  hipError_t alloc(size_t count) {
    release();
    if (count == 0) count = 1;
    hipError_t e = dev_pool_alloc((void**)&p, count * sizeof(T));
    if (e == hipSuccess) { n = count; bytes = count * sizeof(T); }
    return e;
  }
  hipError_t upload(const std::vector<T>& h, gat_ctx* ctx) {
    hipError_t e = alloc(h.size());
    if (e != hipSuccess) return e;
    if (!h.empty()) e = stage_push_h2d(ctx, p, h.data(), h.size() * sizeof(T));   // (small: no wait of its own)
    return e;
  }
  // count elements written by build(T*) straight into the pinned staging buffer (no host vector that is zero-filled first
  // and copied a second time); beyond one staging piece: a plain array and the copy in pieces
  template <typename Build>
  hipError_t upload_built(size_t count, gat_ctx* ctx, Build build) {
    hipError_t e = alloc(count);
    if (e != hipSuccess || count == 0) return e;
    const size_t nbytes = count * sizeof(T);
    if (nbytes <= kStagePiece) {
      if ((e = ctx_stage(ctx, nbytes)) != hipSuccess) return e;
      build(reinterpret_cast<T*>(ctx->h_stage));
      if ((e = hipMemcpyAsync(p, ctx->h_stage, nbytes, hipMemcpyHostToDevice, ctx->stream)) != hipSuccess) return e;
      return hipStreamSynchronize(ctx->stream);
    }
    std::unique_ptr<T[]> h(new T[count]);
    build(h.get());
    return staged_h2d(ctx, p, h.get(), nbytes);
  }
};

// annotations on the device: SoA starts / ends / exclusive cumulated lengths + CSR offsets
constexpr int kMergedWavesHost = 4;   // waves of a k_count_merged workgroup (gat_kernels.h: kMergedThreads / 64; asserted in gat_mi355.hip)
struct AnnoDev {
  bool per_track = true;       // the per-track tables (SoA, grids) exist; false: merged index only (GAT_ANNOTATIONS_NUCLEOTIDE_ONLY)
  DevBuf<uint32_t> start, end, cumx, grid;
  DevBuf<int64_t> off, goff;
  DevBuf<int32_t> shift, cells;
  std::vector<int64_t> h_off;
  // merged multi-track index (k_count_merged), built for problems with several tracks
  DevBuf<uint2> mz;
  DevBuf<uint32_t> mfirst;
  DevBuf<uint4> mcell;             // merged_block 1: per grid cell {first, 0, entry first, entry first + 1, 0, 0} (32 bytes)
  DevBuf<int64_t> mz_off, mf_off;
  DevBuf<int32_t> m_shift, m_cells, m_slot_off, m_slot_contigs;
  int max_slot_contigs = 0;
  bool has_merged = false;
  int merged_block = 2;            // how k_count_merged's scans fetch the index: 8 blocks of eight, 2 pairs, 1 cell records + pairs
  int64_t merged_entries = 0;
  int64_t max_m = 0;
  int64_t max_cells = 0;
  int64_t total = 0;
};

// The annotation side of a problem as an object of its own (gat_annotations_create): the contig-level lists of every track
// and the count kernels' look-up structures, built once per run() and shared by the problems of all segment tracks whose
// contigs are the same (the reference hands the same `annotations` to every track's sampling, gat/__init__.py:971-1010).
struct gat_annotations {
  gat_ctx* ctx = nullptr;
  int refs = 1;                    // the handle + one per problem that counts against it
  bool closed = false;
  int32_t n_tracks = 0, n_groups = 0, merge_groups = 0;
  AnnoDev dev;
  // GAT_ANNOTATIONS_ASYNC: the tables are built by a thread of the library (its own stream and staging buffer: ctx->build_ctx)
  // while the caller goes on -- a problem made against the object samples at once and counts when the tables are there
  std::thread worker;
  std::atomic<int> ready{1};       // 0 while the worker runs
  int build_rc = 0;
  std::string build_err;
  // known before the build (shape_known): whether the merged index will exist and how many intervals the tables will hold --
  // what decides the count kernel's route, and with it the sampler's last steps.  With four tracks or more the index exists
  // whatever the lists hold; lists that pass through ungrouped (no fromIsochores merge) have their sizes in the desc
  bool shape_known = false, will_merge = false;
  int64_t total_known = -1;        // intervals of the tables, or -1 (merged groups: known when built)
};
void annotations_release(gat_annotations* a);     // gat_prep.hip
inline bool annotations_ready(const gat_annotations* a) { return a->ready.load(std::memory_order_acquire) != 0; }
int annotations_wait(gat_ctx* ctx, gat_annotations* a);   // joins the build; returns its error, if any, as the caller's

// wall-clock stamps of problem creation (GAT_TIME_CREATE=1; tools/time_create.py)
struct PrepTimer {
  bool on;
  std::chrono::steady_clock::time_point t;
  PrepTimer() : on(gat_opt(nullptr, "GAT_TIME_CREATE") != nullptr), t(std::chrono::steady_clock::now()) {}
  void lap(const char* what) {
    if (!on) return;
    const auto n = std::chrono::steady_clock::now();
    fprintf(stderr, "[gat] %-34s %7.2f ms\n", what, std::chrono::duration<double, std::milli>(n - t).count());
    t = n;
  }
};

int check_list(gat_ctx* ctx, const gat_segment* s, int64_t n, const char* what, int64_t idx);

// host threads for the per-list / per-contig preparation of gat_problem_create, the observed counts' tables and the input
// statistics (GAT_HOST_THREADS, default min(16, cores)): ONE pool per process, created at the first use and kept -- a
// problem's creation is a dozen of these loops of 0.1-2 ms each, and fifteen threads created and joined per loop were
// 0.3-0.5 ms of every one (gat_prep.hip: host_pool_run; GAT_HOST_POOL=0: threads per loop as before)
void host_pool_run(int64_t n, void (*fn)(void*, int64_t), void* arg);
void host_pool_select(int pool);              // the calling thread's pool from now on: 0 callers (default), 1 the library's builder
template <typename F>
inline void parallel_for(int64_t n, F body) {
  if (n <= 0) return;
  if (n == 1) { body(0); return; }
  host_pool_run(n, [](void* a, int64_t i) { (*static_cast<F*>(a))(i); }, &body);
}

struct gat_problem {
  ~gat_problem() { if (anno) annotations_release(anno); }
  gat_ctx* ctx = nullptr;
  CallState call;                        // the call in flight, if any (one per problem)
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
  DevBuf<uint32_t> d_ws_tree;            // 16-ary search trees over the starts and the cumulated lengths of long workspaces,
                                         // and their grids (UnitDev::pgrid_off / cgrid_off)
  DevBuf<uint4> d_ws_rec;                // per workspace segment {start, end, previous segment's end (INT32_MIN: none), cdf}: what a
                                         // position draw needs of its segment in ONE 16-byte access (k_place_grid)
  DevBuf<int64_t> d_cws_nseg;
  gat_annotations* anno = nullptr;       // the annotation tables: its own (made from the lists of its desc) or a shared object
  // per-batch scratch
  int64_t batch = 0;
  DevBuf<uint2> d_slab, d_cslab;
  DevBuf<int32_t> d_unit_n, d_contig_n;
  // the status word of a batch lives behind its statistics (word 8 of d_stat): one memset, one read-back for both
  int32_t* flags_dev() const { return reinterpret_cast<int32_t*>(d_stat.p + 8); }
  uint32_t* todo_count_dev() const { return reinterpret_cast<uint32_t*>(d_stat.p + 9); }   // (k_tail's queue length: word 9)
  DevBuf<unsigned long long> d_stat;
  DevBuf<unsigned long long> d_mstat;    // k_count_merged: 256 pairs {index entries read, segments looked up} (CountArgs::mstat)
  // lane-parallel front end (k_rng + k_place)
  std::vector<int32_t> h_rng_rows;       // per active index: raw outputs generated per stream
  std::vector<int64_t> h_rng_off;
  int64_t rng_rows_total = 0;            // sum of h_rng_rows
  DevBuf<int32_t> d_rng_rows;
  DevBuf<int4> d_st;
  DevBuf<int4> d_st2;                    // k_merge_big -> k_sampler hand-off (first consolidation of the long lists)
#if defined(GAT_DIAG) || defined(GAT_DIAG_CONS)
  DevBuf<unsigned long long> d_diag;     // diagnostic build: per work unit, cycles per phase of k_sampler (GAT_DIAG_CONS: of k_consolidate)
  DevBuf<unsigned long long> d_diag_place;   // ... per launch position, cycles per phase of k_place's loop
  DevBuf<unsigned long long> d_diag_tiles;   // ... per tile of k_place, begin and end (s_memrealtime)
#endif
  DevBuf<int64_t> d_rng_off;
  DevBuf<uint32_t> d_rng_out, d_ws_stat, d_part;
  DevBuf<uint32_t> d_rng_ckpt;           // k_seed -> k_rng: 16 checkpoints of every stream's seeded state (4 KB per tile)
  DevBuf<uint2> d_fslab;                 // split path: the units' final lists (k_finalize writes out of place)
  DevBuf<uint32_t> d_cum;                // split path: running lengths of the merged lists (parallel to the slab)
  DevBuf<gat::TailPatch> d_patch;        // ... and k_tail's record per work unit
  DevBuf<uint32_t> d_todo;               // ... and the units it leaves to k_sampler (their number: todo_count_dev())
  DevBuf<uint32_t> d_serial;             // gat_sample_and_count_serial: the MT19937 state (and its copy at the batch's start)
  DevBuf<int32_t> d_unit_pos;            // unit id -> launch position (k_contig reads k_tail's records by it)
  std::vector<int32_t> h_unit_pos;
  DevBuf<int4> d_cu_rec;                 // k_contig: per entry of contig_units {unit, slab offset, launch position, 0} (follows the layout)
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
  bool all_cm_ok = false;                // ... and the offset draw's mask does not depend on the length drawn (k_place_scan's units)
  int32_t max_nws = 0;                   // longest workspace among the active units (selects the kernel variants)
  bool grid_place = false;               // some unit's workspace is beyond k_place's LDS table and every such unit has its cdf grid
                                         // (k_place_grid, MODE 4); false: k_place<., 2> searches the trees in global memory
  int32_t grid_lds_words = 0;            // ... the largest grid image: the launch's dynamic LDS
  bool tail_long_ws = false;             // k_tail takes units of more than kTailMaxWs workspace segments (their grids / trees exist)
  // isochore problems counted from the units' lists (k_count_merged<2, .> + k_units_overlap: no k_contig for the nucleotide counters)
  bool units_direct_ok = false;          // merge_contigs, the split path; false for good once a batch had to be repeated through k_contig
  bool units_direct = false;             // ... and the batch in flight took it
  DevBuf<uint32_t> d_bmap;               // per contig one bit per 2^bshift bases: a workspace boundary lies in the cell or the next
  DevBuf<int64_t> d_bmap_off;
  int32_t bshift = 12;
  DevBuf<uint4> d_cand;                  // the candidates k_count_merged notes for k_units_overlap (kCandSlots regions)
  DevBuf<uint32_t> d_cand_count;         // ... their numbers per region, and k_units_overlap's words behind them
  int cand_scale = 1;                    // ... the buffer's size against the estimate (x 4 when a region overflowed, up to 64)
  bool small_tables = false;             // every active unit: <= 64 workspace segments, < 256 working segments
  bool all_one_ws = false;               // every active unit: one workspace segment, bucket 1, and most of the working segments in units
                                         // whose rank table is beyond k_place's LDS table but within k_place_wide's (k_place MODE 3)
  bool pipe_pays = false;                // the single-workspace-segment units hold at least half of the working segments: only their
                                         // loop of k_place_pipe runs through the hand-pipelined rows, and the kernel costs registers
  int swap_capx = 0;                     // > 0: count with k_count_swap, sample lists of up to this many segments in LDS
  bool swap_decided = false;             // (decided at the first call: it takes the size of the annotation tables)
};

// gat_prep.hip
int build_annos(gat_ctx* ctx, AnnoDev& A, const gat_segment* annos, const int64_t* lbeg, const int64_t* lend, int64_t n_lists,
                int32_t n_groups, bool want_merged, bool checked, double mean_seg_len = 0.0, bool nucleotide_only = false);
int layout_slab(gat_problem* P);
int upload_layout(gat_ctx* ctx, gat_problem* P);
