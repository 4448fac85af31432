// gat_mi355.hip -- host side of libgat_mi355.so (C ABI in include/gat_mi355.h).
//
// Host work is limited to what the reference does once per (segments, workspace) pair before
// sampling starts (gat/Engine.pyx:543-565, hoisted out of the per-sample loop), buffer management
// and kernel launches.  Sampling, fromIsochores and counting run only as HIP kernels
// (gat_kernels.h); there is no CPU path for them in this library.
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <link.h>            // dl_iterate_phdr: the RCCL a host process has mapped already
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


#include "gat_host.h"

thread_local std::string g_last_error;

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
  HIPCHK(ctx, hipHostMalloc((void**)&ctx->h_stat, 128, hipHostMallocDefault));
  memset(ctx->h_stat, 0, 128);
  *out = ctx;
  return GAT_OK;
}

// pinned words + events of a call in flight: taken from the context's idle blocks, made when there is none
static int call_block_get(gat_ctx* ctx, CallBlock** out) {
  if (!ctx->call_blocks.empty()) { *out = ctx->call_blocks.back(); ctx->call_blocks.pop_back(); return GAT_OK; }
  std::unique_ptr<CallBlock> b(new CallBlock());
  HIPCHK(ctx, hipHostMalloc((void**)&b->h_stat, (size_t)kMaxInflight * 16 * 8 + 512 * 8, hipHostMallocDefault));
  b->h_mstat = b->h_stat + (size_t)kMaxInflight * 16;
  HIPCHK(ctx, hipEventCreate(&b->ev_begin));
  HIPCHK(ctx, hipEventCreate(&b->ev_end));
  for (auto& pair : b->ev_main) for (auto& ev : pair) HIPCHK(ctx, hipEventCreate(&ev));
  *out = b.release();
  return GAT_OK;
}
static void call_block_free(CallBlock* b) {
  if (!b) return;
  if (b->h_stat) (void)hipHostFree(b->h_stat);
  if (b->ev_begin) (void)hipEventDestroy(b->ev_begin);
  if (b->ev_end) (void)hipEventDestroy(b->ev_end);
  for (auto& pair : b->ev_main) for (auto& ev : pair) if (ev) (void)hipEventDestroy(ev);
  delete b;
}

// the context lives as long as its handle or a problem made on it does (a host may close them in either order)
void ctx_release(gat_ctx* ctx);
extern "C" void gat_ctx_destroy(gat_ctx* ctx) {
  if (!ctx || ctx->closed) return;
  ctx->closed = true;
  ctx_release(ctx);
}
void ctx_release(gat_ctx* ctx) {
  if (--ctx->refs > 0) return;
  (void)hipSetDevice(ctx->device);
  (void)hipStreamSynchronize(ctx->stream);
  ctx->stage_used = 0;
  for (CallBlock* b : ctx->call_blocks) call_block_free(b);
  ctx->call_blocks.clear();
  if (ctx->build_ctx) { gat_ctx_destroy(ctx->build_ctx); ctx->build_ctx = nullptr; }
  if (ctx->aux_ctx) { gat_ctx_destroy(ctx->aux_ctx); ctx->aux_ctx = nullptr; }
  for (auto& ev : ctx->ev) if (ev) (void)hipEventDestroy(ev);
  for (auto& ev : ctx->ev_main) if (ev) (void)hipEventDestroy(ev);
  for (auto& ev : ctx->ev_k) if (ev) (void)hipEventDestroy(ev);
  for (auto& ev : ctx->ev_t) if (ev) (void)hipEventDestroy(ev);
  for (auto& ev : ctx->ev_cnt) if (ev) (void)hipEventDestroy(ev);
  if (ctx->h_stage) (void)hipHostFree(ctx->h_stage);
  if (ctx->h_flags) (void)hipHostFree(ctx->h_flags);
  if (ctx->h_stat) (void)hipHostFree(ctx->h_stat);
  if (ctx->own_stream && ctx->stream) (void)hipStreamDestroy(ctx->stream);
  delete ctx;
}

extern "C" int gat_ctx_set_kernel_times(gat_ctx* ctx, int on) {
  if (!ctx) return set_err(nullptr, GAT_ERR_ARG, "ctx is NULL");
  ctx->kernel_times = on != 0;
  return GAT_OK;
}

extern "C" void* gat_ctx_stream(const gat_ctx* ctx) { return ctx ? (void*)ctx->stream : nullptr; }

extern "C" int gat_ctx_synchronize(gat_ctx* ctx) {
  if (!ctx) return set_err(nullptr, GAT_ERR_ARG, "ctx is NULL");
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
  return GAT_OK;
}

extern "C" int gat_dev_alloc(gat_ctx* ctx, void** out, size_t bytes) {
  if (!ctx || !out) return set_err(ctx, GAT_ERR_ARG, "gat_dev_alloc: NULL argument");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  if (bytes == 0) bytes = 1;
  HIPCHK(ctx, dev_pool_alloc(out, bytes));
  ctx->user_allocs.push_back(std::make_pair(*out, bytes));
  return GAT_OK;
}
extern "C" int gat_dev_free(gat_ctx* ctx, void* p) {
  if (!ctx) return set_err(ctx, GAT_ERR_ARG, "gat_dev_free: NULL ctx");
  if (!p) return GAT_OK;
  for (size_t i = 0; i < ctx->user_allocs.size(); ++i)
    if (ctx->user_allocs[i].first == p) {
      // (the caller may free right behind work it enqueued on the context's stream: a pooled block must be idle)
      HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
      dev_pool_free(p, ctx->user_allocs[i].second);
      ctx->user_allocs.erase(ctx->user_allocs.begin() + (long)i);
      return GAT_OK;
    }
  HIPCHK(ctx, hipFree(p));
  return GAT_OK;
}
extern "C" int gat_memcpy_d2h(gat_ctx* ctx, void* dst, const void* src, size_t bytes) {
  if (!ctx) return set_err(ctx, GAT_ERR_ARG, "NULL ctx");
  HIPCHK(ctx, staged_d2h(ctx, dst, src, bytes));
  return GAT_OK;
}
extern "C" int gat_memcpy_h2d(gat_ctx* ctx, void* dst, const void* src, size_t bytes) {
  if (!ctx) return set_err(ctx, GAT_ERR_ARG, "NULL ctx");
  HIPCHK(ctx, staged_h2d(ctx, dst, src, bytes));
  return GAT_OK;
}

// ------------------------------------------------------------------------------------------
static int ensure_scratch(gat_ctx* ctx, gat_problem* P, int64_t want) {
  if (P->batch >= want) return GAT_OK;
  PrepTimer tm;
  const char* env = gat_opt(ctx, "GAT_SLAB_BYTES");
  // a batch as large as a quarter of the 288 GB takes: the lane-per-stream kernels have a fixed floor per launch (the serial
  // chain of the longest unit's tile) and the wave-per-unit kernels of long-list problems fill the chip only with
  // thousands of samples in flight, so fewer, larger batches are faster (config 3, 10 000 samples: 9.7 ms in two batches
  // under 12 GB, 9.3 ms in one; config-4 shape: a rank's shard of 12 500 samples in one batch of 40 GB instead of two);
  // capped by what the device has free
  double budget = env ? atof(env) : 72.0 * 1024 * 1024 * 1024;
  if (!env) {
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) == hipSuccess)
      budget = std::min(std::min(budget, 0.25 * (double)total_b), 0.6 * (double)(free_b + dev_pool_held()));
  }
  const int64_t per_sample = P->slab_stride * 8 * (P->merge_contigs ? 2 : 1) + 4 * ((int64_t)P->n_units + P->n_contigs) +
                             (P->sampler_mode ? P->rng_rows_total * 4 + 16 * (int64_t)P->n_units : 0) +
                             (P->split_path ? P->slab_stride * 8 + P->slab_stride / 2 + (int64_t)(sizeof(gat::TailPatch) + 4) * P->n_units : 0);
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
    HIPCHK(ctx, P->d_rng_off.alloc(P->h_rng_off.size()));
    HIPCHK(ctx, staged_h2d(ctx, P->d_rng_off.p, P->h_rng_off.data(), P->h_rng_off.size() * 8));
    HIPCHK(ctx, P->d_rng_out.alloc((size_t)P->h_rng_off.back()));
    HIPCHK(ctx, P->d_rng_ckpt.alloc((size_t)nsb * std::max<size_t>(1, P->h_order.size()) * gat::kRngWaves * gat::kWave));
    const size_t ns = (size_t)(b * std::max(1, P->n_units));
    HIPCHK(ctx, P->d_st.alloc(ns));
    HIPCHK(ctx, P->d_st2.alloc(ns));
    if (!P->split_path && P->long_lists) {                                      // k_tail_big's hand-over records
      HIPCHK(ctx, P->d_patch.alloc(ns));
      HIPCHK(ctx, hipMemsetAsync(P->d_patch.p, 0, ns * sizeof(gat::TailPatch), ctx->stream));
      HIPCHK(ctx, P->d_todo.alloc(ns));                                         // ... and the queue of what k_resume_big leaves
    }
    if (P->split_path) {
      HIPCHK(ctx, P->d_cum.alloc((size_t)(b * P->slab_stride / 8 + 8)));      // (slab regions are multiples of 64 entries: cap_for)
      HIPCHK(ctx, P->d_fslab.alloc((size_t)(b * P->slab_stride)));
      HIPCHK(ctx, P->d_patch.alloc(ns));
      HIPCHK(ctx, P->d_todo.alloc(ns));
    }
  }
  P->batch = b;
  tm.lap("scratch for the batch");
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

static_assert(gat::kMergedThreads / gat::kWave == kMergedWavesHost, "gat_host.h: kMergedWavesHost");
// which kernel serves the segment-side counters (the choice launch_count makes)
// merged_only: the tables were BUILT for that route alone (AnnoDev::per_track false -- decided from the knobs as they were then):
// the route stands whatever the knobs say now (a knob flipped between the build and the call was GAT_ERR_ARG)
static int count_route(const gat_ctx* ctx, bool has_merged, const Counters& C, int n_contigs, int n_tracks, int swap_capx,
                       bool merged_only = false) {
  if (!C.any_seg || n_contigs <= 0) return GAT_COUNT_KERNEL_NONE;
  const bool only_overlap = C.slot[GAT_COUNTER_SEGMENT_OVERLAP] < 0 && C.slot[GAT_COUNTER_SEGMENT_MIDOVERLAP] < 0;
  const size_t lds_merged = (size_t)n_tracks * 4 * (gat::kMergedThreads / gat::kWave);
  if (only_overlap && has_merged && (int64_t)lds_merged + 1024 <= ctx->max_lds && (merged_only || !gat_opt(ctx, "GAT_COUNT_NO_MERGED")))
    return GAT_COUNT_KERNEL_MERGED;
  if (swap_capx > 0 && only_overlap && !gat_opt(ctx, "GAT_COUNT_NO_SWAP")) return GAT_COUNT_KERNEL_SWAP;
  return GAT_COUNT_KERNEL_SEG;
}

// launch the count kernels over n_lists sample lists
// ev_main: the pair of events recorded around the dominant kernel; main_recorded / count_kernel: what was launched
struct CountLaunch {
  hipEvent_t* ev_main = nullptr;
  bool main_recorded = false;
  int count_kernel = GAT_COUNT_KERNEL_NONE;
  // k_units_overlap (CountArgs::cu_rec set): the units and their workspaces
  const gat::UnitDev* units = nullptr;
  const uint2* ws = nullptr;
  const uint32_t* ws_tree = nullptr;
  int max_units = 1;          // ... the most units a contig has
};
static int launch_count(gat_ctx* ctx, const AnnoDev& annos, const Counters& C, gat::CountArgs A, DevBuf<uint32_t>& part,
                        int swap_capx, int list_cap, CountLaunch& L) {
  L.main_recorded = false;
  L.count_kernel = GAT_COUNT_KERNEL_NONE;
  for (int i = 0; i < GAT_NUM_COUNTERS; ++i) A.counter_slot[i] = C.slot[i];
  A.a_start = annos.start.p; A.a_end = annos.end.p; A.a_cumx = annos.cumx.p; A.a_off = annos.off.p;
  A.a_grid = annos.grid.p; A.g_off = annos.goff.p; A.c_shift = annos.shift.p; A.c_cells = annos.cells.p;
  if (A.n_samples <= 0 || A.n_tracks <= 0) return GAT_OK;
  if (!annos.per_track && (C.any_anno || (C.any_seg && count_route(ctx, annos.has_merged, C, A.n_contigs, A.n_tracks, swap_capx, true) != GAT_COUNT_KERNEL_MERGED)))
    return set_err(ctx, GAT_ERR_ARG, "the annotation tables were made for the nucleotide counters only (GAT_ANNOTATIONS_NUCLEOTIDE_ONLY): "
                                     "no per-track tables for the counters asked for");
  if (C.any_seg) {
    const int E_max = (int)count_lds_entries(ctx);
    const char* env_sc = gat_opt(ctx, "GAT_COUNT_SAMPLES_PER_BLOCK");
    int SC = env_sc ? atoi(env_sc) : 32;
    if (!env_sc) {   // enough blocks to fill 256 CUs several times over, large enough to amortise the staging
      const int64_t tiles0 = (A.n_tracks + 0) , work = (int64_t)A.n_samples * tiles0 * std::max(1, A.n_contigs) / 8192;
      SC = 8;
      while (SC < 128 && SC * 2 <= work) SC *= 2;
    }
    SC = std::max(1, std::min(SC, 256));
    const char* env_tt = gat_opt(ctx, "GAT_COUNT_TRACKS_PER_BLOCK");
    const int TT_max = env_tt ? atoi(env_tt) : 16;
    int TT;
    const char* env_st = gat_opt(ctx, "GAT_COUNT_STAGED");
    // (+4: a staged list has an entry in front of its first interval and three sentinels behind the last: segs_vs_pairs)
    bool staged = count_lists_staged(ctx, annos.max_m) && !(env_st && atoi(env_st) == 0);
    if (staged) TT = (int)std::min<int64_t>(std::min<int64_t>(A.n_tracks, TT_max), E_max / (annos.max_m + 4));
    else TT = (int)std::min<int64_t>(A.n_tracks, TT_max);
    TT = std::max(1, TT);
    A.tracks_per_block = TT;
    A.samples_per_block = SC;
    A.lds_entries = staged ? (int)std::min<int64_t>((int64_t)E_max, (annos.max_m + 4) * TT) : 0;
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
    const int route = count_route(ctx, annos.has_merged, C, A.n_contigs, A.n_tracks, swap_capx, !annos.per_track);
    if (route == GAT_COUNT_KERNEL_MERGED) {
      // several tracks: one look-up per sample segment in the merged index of all tracks
      A.mz = annos.mz.p; A.mz_off = annos.mz_off.p; A.mfirst = annos.mfirst.p; A.mf_off = annos.mf_off.p;
      A.m_shift = annos.m_shift.p; A.m_cells = annos.m_cells.p;
      A.m_slot_off = annos.m_slot_off.p; A.m_slot_contigs = annos.m_slot_contigs.p;
      const char* env_sg = gat_opt(ctx, "GAT_MERGED_SAMPLES_PER_BLOCK");
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
      const bool patch = A.seg_merged != nullptr;
      const int blk = annos.merged_block;
      A.mcell = annos.mcell.p;
      const void* km = patch ? (blk == 8 ? (const void*)gat::k_count_merged<true, 8> : blk == 1 ? (const void*)gat::k_count_merged<true, 1>
                                                                                                 : (const void*)gat::k_count_merged<true, 2>)
                             : (blk == 8 ? (const void*)gat::k_count_merged<false, 8> : blk == 1 ? (const void*)gat::k_count_merged<false, 1>
                                                                                                  : (const void*)gat::k_count_merged<false, 2>);
      HIPCHK(ctx, hipFuncSetAttribute(km, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_merged));
      HIPCHK(ctx, hipEventRecord(L.ev_main[0], ctx->stream));
      const dim3 gm((unsigned)nblocks), bm(gat::kMergedThreads);
      if (patch && blk == 8) hipLaunchKernelGGL((gat::k_count_merged<true, 8>), gm, bm, lds_merged, ctx->stream, A);
      else if (patch && blk == 1) hipLaunchKernelGGL((gat::k_count_merged<true, 1>), gm, bm, lds_merged, ctx->stream, A);
      else if (patch) hipLaunchKernelGGL((gat::k_count_merged<true, 2>), gm, bm, lds_merged, ctx->stream, A);
      else if (blk == 8) hipLaunchKernelGGL((gat::k_count_merged<false, 8>), gm, bm, lds_merged, ctx->stream, A);
      else if (blk == 1) hipLaunchKernelGGL((gat::k_count_merged<false, 1>), gm, bm, lds_merged, ctx->stream, A);
      else hipLaunchKernelGGL((gat::k_count_merged<false, 2>), gm, bm, lds_merged, ctx->stream, A);
      HIPCHK(ctx, hipGetLastError());
      if (A.cu_rec != nullptr) {
        // the contig lists were only concatenated (k_contig<., true>): what fromIsochores would have united -- the overlaps
        // between different units' segments -- off the partial sums again
        gat::UnitsOverlapArgs B;
        B.C = A; B.units = L.units; B.ws = L.ws; B.ws_tree = L.ws_tree;
        const int64_t items = (int64_t)A.cand_cap * std::max(1, L.max_units);
        hipLaunchKernelGGL(gat::k_units_overlap, dim3((unsigned)((items + 255) / 256), gat::kCandSlots), dim3(256), 0, ctx->stream, B,
                           std::max(1, L.max_units));
        HIPCHK(ctx, hipGetLastError());
      }
      HIPCHK(ctx, hipEventRecord(L.ev_main[1], ctx->stream));
      L.main_recorded = true;
      L.count_kernel = GAT_COUNT_KERNEL_MERGED;
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
      HIPCHK(ctx, hipEventRecord(L.ev_main[0], ctx->stream));
      hipLaunchKernelGGL(gat::k_count_swap, dim3((unsigned)A.n_samples, gcy, gcz), dim3(gat::kSwapThreads), lds_swap, ctx->stream, B);
      HIPCHK(ctx, hipGetLastError());
      HIPCHK(ctx, hipEventRecord(L.ev_main[1], ctx->stream));
      L.main_recorded = true;
      L.count_kernel = GAT_COUNT_KERNEL_SWAP;
    } else
    if (A.n_contigs > 0) {
    const bool hits = C.slot[GAT_COUNTER_SEGMENT_OVERLAP] >= 0 || C.slot[GAT_COUNTER_SEGMENT_MIDOVERLAP] >= 0;
    HIPCHK(ctx, hipEventRecord(L.ev_main[0], ctx->stream));
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
    HIPCHK(ctx, hipEventRecord(L.ev_main[1], ctx->stream));
    L.main_recorded = true;
    L.count_kernel = GAT_COUNT_KERNEL_SEG;
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
    if (list_cap > 0 && A.n_contigs > 0 && (int64_t)lds_a + 1024 <= ctx->max_lds && !gat_opt(ctx, "GAT_COUNT_NO_SWAP")) {
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
static int finish_sampler_batch(gat_ctx* ctx, gat_problem* P, int64_t nb, gat_stats* st, bool timed, const unsigned long long* h_stat);
// defer: only enqueue (the caller adds the count kernels behind, synchronises once and calls finish_sampler_batch)
// records_ok: the consumer is k_count_seg alone, which reads (merged list, k_tail's record): no k_finalize
// serial_state: the run's ONE MT19937 state on the device (k_serial: the reference's own stream) instead of the per-unit streams
// units_direct: an isochore problem whose counters (the nucleotide counters, through the merged index) take the units' lists as
// the sampler leaves them: no k_contig (k_count_merged<2, .> + k_units_overlap; P->units_direct says whether the batch took it)
static int run_sampler_batch(gat_ctx* ctx, gat_problem* P, uint32_t seed, int64_t begin, int64_t nb,
                             gat_stats* st, bool timed, bool need_unit_lists = false, bool defer = false, bool records_ok = false,
                             uint32_t* serial_state = nullptr, bool loose_ok = false, unsigned long long* h_stat = nullptr,
                             bool units_direct = false) {
  if (h_stat == nullptr) h_stat = ctx->h_stat;
  P->units_direct = false;
  {
    int rc = ensure_scratch(ctx, P, nb);
    if (rc) return rc;
    if (P->batch < nb) return set_err(ctx, GAT_ERR_MEMORY, "internal: batch %lld > scratch %lld", (long long)nb, (long long)P->batch);
    // (unit_n, contig_n and ws_stat are zeroed once when allocated: the kernels rewrite every entry of the active units
    //  in every batch and never touch the others)
#ifdef GAT_DBG_QUEUE
    HIPCHK(ctx, hipMemsetAsync(P->d_stat.p, 0, 16 * 8, ctx->stream));
#else
    HIPCHK(ctx, hipMemsetAsync(P->d_stat.p, 0, 10 * 8, ctx->stream));
#endif         // (statistics, status word, k_tail's queue length)
    if (units_direct && P->d_cand_count.n) HIPCHK(ctx, hipMemsetAsync(P->d_cand_count.p, 0, P->d_cand_count.n * 4, ctx->stream));
    if (timed) HIPCHK(ctx, hipEventRecord(ctx->ev[0], ctx->stream));
    const int32_t* skip_ptr = nullptr;
    int skip_stride = 0;
    if (!P->h_order.empty()) {
      const int smode = serial_state != nullptr ? 0 : P->sampler_mode;
      gat::SamplerArgs A;
      memset(&A, 0, sizeof(A));
      A.units = P->d_units.p; A.units_o = P->d_units_o.p; A.order = P->d_order.p; A.n_units = P->n_units; A.batch = (int32_t)nb;
      A.rec_stride = (int32_t)P->batch;                  // (the scratch's size in samples: the records' row length in every batch)
      A.ws = P->d_ws.p; A.ws_cdf = P->d_ws_cdf.p; A.rank_len = P->d_rank_len.p; A.ws_tree = P->d_ws_tree.p;
      A.ws_rec = P->d_ws_rec.p;
      A.seed = seed; A.sample_begin = begin; A.sampler_kind = P->sampler;
      A.place_plain_step = gat_opt(ctx, "GAT_PLACE_NO_CM") ? 1 : 0;
      A.slab = P->d_slab.p; A.slab_stride = P->slab_stride;
      A.unit_n = P->d_unit_n.p; A.flags = P->flags_dev(); A.stat = P->d_stat.p; A.ws_stat = P->d_ws_stat.p;
#if defined(GAT_DIAG) || defined(GAT_DIAG_CONS)
      {
        const size_t nd = (size_t)nb * std::max(1, P->n_units) * 8;
        if (P->d_diag.n < nd) HIPCHK(ctx, P->d_diag.alloc(nd));
        HIPCHK(ctx, hipMemsetAsync(P->d_diag.p, 0, nd * 8, ctx->stream));
        A.diag = P->d_diag.p;
      }
#endif
#ifdef GAT_DIAG
      {
        const size_t np = std::max<size_t>(1, P->h_order.size()) * 8;
        if (P->d_diag_place.n < np) HIPCHK(ctx, P->d_diag_place.alloc(np));
        HIPCHK(ctx, hipMemsetAsync(P->d_diag_place.p, 0, np * 8, ctx->stream));
        A.diag_place = P->d_diag_place.p;
        const size_t nt = std::max<size_t>(1, P->h_order.size()) * (size_t)((nb + 63) / 64) * 2;
        if (P->d_diag_tiles.n < nt) HIPCHK(ctx, P->d_diag_tiles.alloc(nt));
        HIPCHK(ctx, hipMemsetAsync(P->d_diag_tiles.p, 0, nt * 8, ctx->stream));
        A.diag_tiles = P->d_diag_tiles.p;
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
        A.rng_ckpt = P->d_rng_ckpt.p;
        A.st = P->d_st.p;
        {
          // the streams' seeding chains at full occupancy, 16 checkpoints each; k_rng's waves regenerate the rest
          const int64_t n_tiles = (int64_t)nsb * n_act, per_block = gat::kSeedThreads / gat::kWave;
          hipLaunchKernelGGL(gat::k_seed, dim3((unsigned)((n_tiles + per_block - 1) / per_block)), dim3(gat::kSeedThreads), 0, ctx->stream,
                             A, (int)nsb);
          HIPCHK(ctx, hipGetLastError());
        }
        const size_t lds_rng = (size_t)gat::kMtN * 64 * 4;
        HIPCHK(ctx, hipFuncSetAttribute((const void*)gat::k_rng, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_rng));
        hipLaunchKernelGGL(gat::k_rng, dim3(nsb, gy, gz), dim3(gat::kRngThreads), lds_rng, ctx->stream, A);
        HIPCHK(ctx, hipGetLastError());
        if (timed) HIPCHK(ctx, hipEventRecord(ctx->ev_k[0], ctx->stream));
        const dim3 gp(nsb, gy, gz);
        // (small calls of a problem of simple units -- about a wave per SIMD or less, e.g. the metric's 10 000 samples cut over
        //  eight GPUs -- take the wide kernel as well: its loop stores unconditionally and four tiles share a rank table; config 2,
        //  k_place at 1 250 / 2 500 / 5 000 samples: 0.46 / 0.47 / 0.56 ms against the lean kernel's 0.52 / 0.53 / 0.57, at 10 000
        //  1.05 against 0.92)
        const bool small_call = (int64_t)nsb * (int64_t)n_act <= 1536 && !gat_opt(ctx, "GAT_PLACE_NO_WIDE");
        const int mode = P->all_simple ? ((gat_opt(ctx, "GAT_PLACE_WIDE") || small_call) ? 3 : 1)
                                       : (P->all_one_ws && !gat_opt(ctx, "GAT_PLACE_NO_WIDE") ? 3 : (P->max_nws > gat::kPlaceWsLds ? 2 : 0));
        // (k_place_wide: kPlaceWide tiles per workgroup, the largest unit's rank table beside their rings)
        const dim3 gw((nsb + gat::kPlaceWide - 1) / gat::kPlaceWide, gy, gz);
        const size_t lds_wide = (size_t)P->max_hist * 4;
        // calls of a few hundred tiles of a problem of simple units: one stream walked by the 64 lanes of a wave (k_place_scan:
        // a stream's states as a prefix scan over its rows) instead of one lane walking it row by row -- the chain of the longest
        // unit's tile, 0.37 ms on config 2 whatever the sample count, is what such a call waited for
        const char* env_scan = gat_opt(ctx, "GAT_PLACE_SCAN_TILES");
        const int64_t scan_tiles = env_scan ? atoll(env_scan) : 768;      // (config 2: faster up to ~2 300 samples x 24 units per call)
        const bool scan = P->sampler == GAT_SAMPLER_ANNOTATOR && P->all_simple && P->all_cm_ok && A.place_plain_step == 0 &&
                          (int64_t)nsb * (int64_t)n_act <= scan_tiles;
        if (scan) {
          const int64_t n_tiles = (int64_t)nsb * n_act;
          const unsigned nblocks = (unsigned)(((n_tiles + 7) / 8) * 8 * 4);
          hipLaunchKernelGGL(gat::k_place_scan, dim3(nblocks), dim3(gat::kScanWaves * 64), 0, ctx->stream, A, (int)nsb,
                             gat_opt(ctx, "GAT_PLACE_SCAN_SEQ") ? 1 : 0);
        } else
        if (P->sampler == GAT_SAMPLER_SEGMENTS) {
          if (mode == 3) {
            HIPCHK(ctx, hipFuncSetAttribute((const void*)gat::k_place_wide<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_wide));
            hipLaunchKernelGGL((gat::k_place_wide<1>), gw, dim3(gat::kPlaceWide * 64), lds_wide, ctx->stream, A);
          } else if (mode == 1) hipLaunchKernelGGL((gat::k_place<1, 1>), gp, dim3(64), 0, ctx->stream, A);
          else if (mode == 0) hipLaunchKernelGGL((gat::k_place<1, 0>), gp, dim3(64), 0, ctx->stream, A);
          else hipLaunchKernelGGL((gat::k_place<1, 2>), gp, dim3(64), 0, ctx->stream, A);
        } else {
          // (k_place_pipe: the rows of the single-workspace-segment units prefetched by hand, see GAT_PLACE_LOOP_PIPE)
          const bool pipe = P->pipe_pays && !gat_opt(ctx, "GAT_PLACE_NO_PIPE");
          const bool rank_fits = P->max_hist < (uint32_t)gat::kPlaceRankLds;
          // fragmented workspaces: the cdf grids of the long workspaces in LDS, eight tiles of a unit per workgroup (k_place_grid);
          // static LDS: the workspace and rank tables (4 KB each) and an 8 KB ring per tile
          const size_t lds_grid = (size_t)P->grid_lds_words * 4 + 16;
          const bool grid_k = mode == 2 && P->grid_place && (int64_t)lds_grid + 8192 + 1024 + (int64_t)gat::kPlaceGridTiles * 8192 <= ctx->max_lds;
          if (grid_k) {
            HIPCHK(ctx, hipFuncSetAttribute((const void*)gat::k_place_grid, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_grid));
            hipLaunchKernelGGL(gat::k_place_grid, dim3((nsb + gat::kPlaceGridTiles - 1) / gat::kPlaceGridTiles, gy, gz),
                               dim3(gat::kPlaceGridTiles * 64), lds_grid, ctx->stream, A);
          } else
          if (mode == 3) {
            HIPCHK(ctx, hipFuncSetAttribute((const void*)gat::k_place_wide<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_wide));
            hipLaunchKernelGGL((gat::k_place_wide<0>), gw, dim3(gat::kPlaceWide * 64), lds_wide, ctx->stream, A);
          } else if (mode == 1 && pipe) hipLaunchKernelGGL((gat::k_place_pipe<0, 1>), gp, dim3(64), 0, ctx->stream, A);
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
      const bool huge = P->sampler != GAT_SAMPLER_SEGMENTS && ((int64_t)lds > ctx->max_lds || gat_opt(ctx, "GAT_TEST_HUGE") != nullptr);
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
            !gat_opt(ctx, "GAT_NO_MERGE_BIG")) {
          gat::SamplerArgs M = A;
          M.st2 = P->d_st2.p;
          M.big_buckets = nbk;
          M.cum = split ? P->d_cum.p : nullptr;
          n_long_big = n_long;
          const bool tree_m = P->max_nws > gat::kWsTreeMin;
          // (the form by the class's longest list: 8, 16 or 24 elements of it in every thread's registers; 0 = round 3's form,
          //  the list read twice and sorted bucket by bucket)
          typedef void (*merge_fn)(gat::SamplerArgs);
          static const merge_fn kMergeFns[2][4] = {
              {gat::k_merge_big<false, 0>, gat::k_merge_big<false, 8>, gat::k_merge_big<false, 16>, gat::k_merge_big<false, gat::kMergeRegs>},
              {gat::k_merge_big<true, 0>, gat::k_merge_big<true, 8>, gat::k_merge_big<true, 16>, gat::k_merge_big<true, gat::kMergeRegs>}};
          const bool merge_old = gat_opt(ctx, "GAT_MERGE_OLD") != nullptr;
          for (int f = 0; f < 4; ++f)
            HIPCHK(ctx, hipFuncSetAttribute((const void*)kMergeFns[tree_m ? 1 : 0][f], hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_m));
          M.n_long = (int32_t)n_long;
          for (size_t c = 0; c + 1 < P->h_class_start.size() && (unsigned)P->h_class_start[c] < n_long; ++c) {
            // one launch per size class: LDS for the class's longest list and its histogram
            const int a0 = P->h_class_start[c], a1 = std::min<int>(P->h_class_start[c + 1], (int)n_long);
            // LDS for what the class's longest unit is EXPECTED to have placed at its first consolidation, not for its slab
            // region (a quarter + 96 above the unit's segments): the count of placements is the unit's segments +- a
            // renewal count's spread, cv(length) x sqrt(n) -- 90 for 8 000 segments -- so a twelfth + 64 on top is seven of
            // those; a list beyond it takes k_sampler's own (slow) way.  And as many histogram buckets (a power of two, at
            // least 1 024: a few elements per bucket sort as fast as one) as leave the number of workgroups a CU can hold
            // at its maximum: this kernel is a chain of dependent passes, and chr1's 8 000 segments with 8 192 buckets were
            // 113 KB -- ONE workgroup per CU for the classes that take most of its time (config-4 shape: 22.9 of 27.4 ms)
            const int n0 = (int)P->h_units[(size_t)P->h_order[(size_t)a0]].hist_total;
            const int ccap = std::min<int>(P->h_units[(size_t)P->h_order[(size_t)a0]].slab_cap, (n0 + n0 / 12 + 64 + 63) / 64 * 64);
            auto wgs_at = [&](int nbk_) { return (int)(((size_t)ctx->max_lds) / ((size_t)2 * ccap * 4 + (size_t)(nbk_ + 1) * 4 + 1024)); };
            int cnbk = 8192;
            while (cnbk > 1024 && (cnbk / 2 >= ccap || wgs_at(cnbk) < wgs_at(1024))) cnbk >>= 1;
            const size_t lds_c = (size_t)2 * ccap * 4 + (size_t)(cnbk + 1) * 4;
            const int per = (ccap + gat::kMergeThreads - 1) / gat::kMergeThreads;
            const int form = merge_old || per > gat::kMergeRegs ? 0 : (per <= 8 ? 1 : (per <= 16 ? 2 : 3));
            size_t lds_k = lds_c;
            if (form > 0) {
              // (the register forms: the histogram shares the list's LDS -- up to about a bucket per element)
              int maxb = 8192;
              if (const char* e = gat_opt(ctx, "GAT_MERGE_BUCKETS")) maxb = std::max(1024, atoi(e));
              cnbk = 1024;
              while (2 * cnbk + gat::kMergeThreads + 1 <= ccap && 2 * cnbk <= maxb) cnbk <<= 1;     // (+ a word of padding per thread)
              lds_k = ((size_t)ccap + (size_t)std::max(ccap, cnbk + gat::kMergeThreads + 1)) * 4;   // (short lists: the 1 024 buckets and their padding need their own words)
            }
            M.a_base = a0; M.a_end = a1; M.lds_cap = ccap; M.big_buckets = cnbk;
            const unsigned cnt = (unsigned)(a1 - a0), gmy = std::min(cnt, 32768u);
            const dim3 gm((unsigned)nb, gmy, (cnt + gmy - 1) / gmy);
            hipLaunchKernelGGL(kMergeFns[tree_m ? 1 : 0][form], gm, dim3(gat::kMergeThreads), lds_k, ctx->stream, M);
            HIPCHK(ctx, hipGetLastError());
          }
          A.st2 = P->d_st2.p;                    // (k_sampler reads it for those units only: see n_long below)
          A.n_long = (int32_t)n_long;
          if (!split && P->d_patch.p != nullptr && P->max_nws <= gat::kTailMaxWs && !gat_opt(ctx, "GAT_NO_TAIL_BIG")) {
            // the placement rounds behind that consolidation, one stream per lane; k_sampler resumes at the trim
            gat::TailArgs TB;
            TB.S = A;
            TB.cum = nullptr; TB.patch = P->d_patch.p; TB.todo = nullptr; TB.todo_count = nullptr;
            // (counts alone, by k_count_merged -- or through k_contig, which merge(0)s the lists again: what a trim emptied may
            //  stay in the list as [0, 0))
            TB.loose_ok = (loose_ok && !need_unit_lists && !gat_opt(ctx, "GAT_RESUME_COMPACT")) ? 1 : 0;
            // (bit 0: no bridge between two neighbours, bit 1: none over two segments on the right; "1" or anything else: both off)
            TB.no_bridge = 0;
            if (const char* e = gat_opt(ctx, "GAT_TB_NO_BRIDGE")) { const int v = atoi(e); TB.no_bridge = (v == 2 || v == 5) ? (v == 2 ? 2 : 1) : 3; }
            TB.no_log_map = gat_opt(ctx, "GAT_TB_NO_LOG_MAP") ? 1 : 0;
            const unsigned gby = std::min(n_long, 32768u);
            hipLaunchKernelGGL(gat::k_tail_big, dim3((unsigned)((nb + 63) / 64), gby, (n_long + gby - 1) / gby), dim3(64), 0,
                               ctx->stream, TB);
            HIPCHK(ctx, hipGetLastError());
            if (!gat_opt(ctx, "GAT_NO_RESUME_BIG")) {
              // ... and the rest of the unit -- log inserted, trim, final filter -- with the list where it is
              // (a reader that takes the segments one by one in any order -- k_count_merged on the units' lists, no contig lists
              //  in between --: the log stays behind the merged list, the trim works on virtual indices)
              const dim3 gr((unsigned)nb, gby, (n_long + gby - 1) / gby);
              if (TB.loose_ok && !P->merge_contigs && !gat_opt(ctx, "GAT_RESUME_INSERT"))
                hipLaunchKernelGGL(gat::k_resume_big<true>, gr, dim3(64), 0, ctx->stream, TB);
              else
                hipLaunchKernelGGL(gat::k_resume_big<false>, gr, dim3(64), 0, ctx->stream, TB);
              HIPCHK(ctx, hipGetLastError());
            }
            A.tb = reinterpret_cast<const int32_t*>(P->d_patch.p);
            A.skip_stride = (int32_t)(sizeof(gat::TailPatch) / 4);
            if (!gat_opt(ctx, "GAT_NO_RESUME_BIG") && !gat_opt(ctx, "GAT_NO_LONG_QUEUE")) {
              // k_sampler behind them works off a queue: launched over every (sample, unit) -- one or two waves a CU with such
              // lists in LDS -- it took 3.1 ms per 12 500 samples of the config-4 shape to find every unit finished
              TB.todo = P->d_todo.p; TB.todo_count = P->todo_count_dev();
              const int64_t tot = (int64_t)nb * n_act;
              hipLaunchKernelGGL(gat::k_queue_rest, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, ctx->stream, TB, (int)n_act);
              HIPCHK(ctx, hipGetLastError());
              A.todo = P->d_todo.p;
              A.todo_count = P->todo_count_dev();
            }
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
        T.loose_ok = 0; T.no_bridge = 0; T.no_log_map = 0;
        T.S.st2 = P->d_st2.p;
        T.S.n_long = (int32_t)n_long_big;                           // (whose verdict k_consolidate respects)
        T.S.lds_cap = std::min(P->max_unit_cap, 1280);              // k_consolidate: the lists the wave bucket sorts take
        T.S.slab_final = P->d_fslab.p;
        T.cum = P->d_cum.p;
        T.patch = P->d_patch.p;
        T.todo = P->d_todo.p;
        T.todo_count = P->todo_count_dev();                         // (zeroed with the statistics at the batch's start)
        const size_t lds_max = (size_t)(gat::kSortScratchWords + 2 * (size_t)T.S.lds_cap) * 4;
        const dim3 gu((unsigned)nb, gy, gz), gt((unsigned)((nb + 63) / 64), gy, gz);
        if (tree) HIPCHK(ctx, hipFuncSetAttribute((const void*)gat::k_consolidate<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_max));
        else HIPCHK(ctx, hipFuncSetAttribute((const void*)gat::k_consolidate<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_max));
        // (a call of a few hundred tiles does not fill the chip whatever the LDS of a workgroup: one launch for all classes,
        //  each launch less is a tail less -- config 2 at 1 250 samples: k_consolidate 0.121 -> 0.101 ms)
        const bool one_class = (int64_t)((nb + 63) / 64) * (int64_t)n_act <= 1024 && !gat_opt(ctx, "GAT_SIZE_CLASSES");
        for (size_t c = 0; c + 1 < P->h_class_start.size(); ++c) {
          // one launch per size class, its LDS sized for the class's longest list
          if (one_class && c > 0) break;
          const int a0 = P->h_class_start[c], a1 = one_class ? P->h_class_start.back() : P->h_class_start[c + 1];
          // (as for k_merge_big: LDS for what the class's longest unit is expected to have placed -- its segments +- a renewal
          //  count's spread -- not for its slab region; the rare list beyond it is k_sampler's.  GAT_CONSOLIDATE_SLAB_LDS: the old size)
          const int n0 = (int)P->h_units[(size_t)P->h_order[(size_t)a0]].hist_total;
          const int tight = gat_opt(ctx, "GAT_CONSOLIDATE_SLAB_LDS") ? INT32_MAX : (n0 + n0 / 12 + 64 + 31) / 32 * 32;
          const int ccap = std::min(std::min(P->h_units[(size_t)P->h_order[(size_t)a0]].slab_cap, T.S.lds_cap), tight);
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
        if (P->tail_long_ws && P->max_nws > gat::kTailMaxWs) hipLaunchKernelGGL(gat::k_tail<true>, gt, dim3(64), 0, ctx->stream, T);
        else hipLaunchKernelGGL(gat::k_tail<false>, gt, dim3(64), 0, ctx->stream, T);
        HIPCHK(ctx, hipGetLastError());
        if (timed) HIPCHK(ctx, hipEventRecord(ctx->ev_t[0], ctx->stream));
        // isochore problems: k_contig re-sorts the units of a contig anyway and takes (merged list, k_tail's record) as
        // it is -- no final unit lists unless somebody asked for them (gat_sample_units)
        P->patched_contigs = P->merge_contigs && P->n_contigs > 0 && !need_unit_lists && !gat_opt(ctx, "GAT_CONTIG_FINAL_LISTS");
        P->units_direct = units_direct && P->patched_contigs && P->units_direct_ok && serial_state == nullptr;
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
        A.todo_count = P->todo_count_dev();
      } else if (timed) HIPCHK(ctx, hipEventRecord(ctx->ev_k[2], ctx->stream));
      // variant: sampler kind x (long lists: counting-sort scratch) x (workspaces beyond the register loop: search trees)
      int variant = P->sampler == GAT_SAMPLER_SEGMENTS ? (tree ? 5 : 4)
                  : huge ? (tree ? 7 : 6) : (long_lists ? 2 : 0) + (tree ? 1 : 0);
      // short lists only (20 waves of this kernel fit a CU's LDS): the instantiation with registers for 5 waves per SIMD
      if (variant == 0 && (int64_t)lds * 20 <= ctx->max_lds && !gat_opt(ctx, "GAT_NO_WPE5")) variant = 8;
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
      } else if (list_in_lds && A.big_buckets == 0 && P->h_class_start.size() > 2 &&
                 !(A.todo_count != nullptr && P->max_nws <= gat::kWsTreeMin && !gat_opt(ctx, "GAT_SIZE_CLASSES"))) {
        // (not behind k_resume_big where it takes every unit -- workspaces of up to 32 segments --: the queue then holds the
        //  per cent of units it declined, and one launch runs them side by side where five ran them class after class)
        // one launch per size class: LDS for the class's longest list
        for (size_t c = 0; c + 1 < P->h_class_start.size(); ++c) {
          const int a0 = P->h_class_start[c], a1 = P->h_class_start[c + 1];
          const int ccap = P->h_units[(size_t)P->h_order[(size_t)a0]].slab_cap;
          gat::SamplerArgs K = A;
          K.a_base = a0; K.a_end = a1;
          const unsigned cnt = (unsigned)(a1 - a0), cy = std::min(cnt, 32768u);
          // (off the queue -- long lists behind k_resume_big --: a one-dimensional launch, every class takes its own entries)
          const dim3 gc = A.todo_count != nullptr ? dim3((unsigned)std::min<int64_t>((int64_t)nb * cnt, 8192)) : dim3((unsigned)nb, cy, (cnt + cy - 1) / cy);   // (8 192: what is left is a few long units, one to a workgroup)
          launch_sampler(gc, (size_t)(gat::kMtLdsWords + 2 * (size_t)ccap) * 4, K);
        }
      } else if (A.todo_count != nullptr) {
        launch_sampler(dim3((unsigned)std::min<int64_t>((int64_t)nb * n_act, 8192)), lds, A);      // off the queue
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
      B.cu_rec = P->d_cu_rec.p;
      B.contig_slab_off = P->d_contig_slab_off.p; B.n_units = P->n_units; B.n_contigs = P->n_contigs;
      B.slab_in = P->final_slab(); B.slab_out = P->d_cslab.p; B.slab_stride = P->slab_stride;
      B.unit_n = P->d_unit_n.p; B.contig_n = P->d_contig_n.p; B.stat = P->d_stat.p;
      B.slab_merged = nullptr; B.unit_pos = P->d_unit_pos.p; B.st2 = nullptr; B.patch = nullptr; B.patch_stride = 0;
      B.ws_stat = P->d_ws_stat.p;
      B.rec_stride = (int32_t)P->batch;
      if (P->split_ran && P->patched_contigs) {
        B.slab_merged = P->d_slab.p;
        B.st2 = P->d_st2.p;
        B.patch = reinterpret_cast<const int32_t*>(P->d_patch.p);
        B.patch_stride = (int32_t)(sizeof(gat::TailPatch) / 4);
      }
      const int need_max = P->h_contig_order.empty() ? 64 : P->h_contig_need[(size_t)P->h_contig_order[0]];
      size_t lds = (size_t)std::max(64, need_max) * 8 + gat::kSortScratchWords * 4;
      const bool huge_c = (int64_t)lds > ctx->max_lds || gat_opt(ctx, "GAT_TEST_HUGE") != nullptr;   // list stays in the output slab
      if (huge_c) lds = gat::kSortScratchWords * 4;
      // (units_direct: the lists only concatenated, the candidates for k_units_overlap noted: k_contig<., true>)
      const bool nosort = P->units_direct;
      B.bmap = P->d_bmap.p; B.bmap_off = P->d_bmap_off.p; B.bshift = P->bshift;
      B.cand = P->d_cand.p; B.cand_cap = (uint32_t)(P->d_cand.n / gat::kCandSlots); B.cand_count = P->d_cand_count.p;
      const void* kc = nosort ? (huge_c ? (const void*)gat::k_contig<true, true> : (const void*)gat::k_contig<false, true>)
                              : (huge_c ? (const void*)gat::k_contig<true> : (const void*)gat::k_contig<false>);
      HIPCHK(ctx, hipFuncSetAttribute(kc, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      B.order = P->d_contig_order.p;
      B.flags = P->flags_dev();
      // one launch per size class: LDS for the class's longest expected list (more waves per CU for the short contigs)
      for (size_t k = 0; k + 1 < P->h_contig_class_start.size(); ++k) {
        const int c0 = P->h_contig_class_start[k], c1 = P->h_contig_class_start[k + 1];
        // (one launch where the list is put together in place: beyond LDS, or only concatenated)
        const bool one = huge_c || nosort;
        if (one && k > 0) break;
        B.base = one ? 0 : c0;
        B.count = one ? P->n_contigs : c1 - c0;
        B.lds_cap = one ? 0 : std::max(64, P->h_contig_need[(size_t)P->h_contig_order[(size_t)c0]]);
        const size_t lds_k = nosort ? 64 : (huge_c ? lds : (size_t)B.lds_cap * 8 + gat::kSortScratchWords * 4);
        const unsigned gcy = (unsigned)std::min(B.count, 32768), gcz = ((unsigned)B.count + gcy - 1) / gcy;
        if (nosort && huge_c) hipLaunchKernelGGL((gat::k_contig<true, true>), dim3((unsigned)nb, gcy, gcz), dim3(64), lds_k, ctx->stream, B);
        else if (nosort) hipLaunchKernelGGL((gat::k_contig<false, true>), dim3((unsigned)nb, gcy, gcz), dim3(64), lds_k, ctx->stream, B);
        else if (huge_c) hipLaunchKernelGGL(gat::k_contig<true>, dim3((unsigned)nb, gcy, gcz), dim3(64), lds_k, ctx->stream, B);
        else hipLaunchKernelGGL(gat::k_contig<false>, dim3((unsigned)nb, gcy, gcz), dim3(64), lds_k, ctx->stream, B);
        HIPCHK(ctx, hipGetLastError());
      }
    }
    if (timed) HIPCHK(ctx, hipEventRecord(ctx->ev[2], ctx->stream));
    if (!P->h_order.empty()) {
      // (behind k_contig: on isochore problems it is k_contig that writes the statistics of the units k_tail finished)
      hipLaunchKernelGGL(gat::k_reduce_stats, dim3(256), dim3(256), 0, ctx->stream, (const uint32_t*)P->d_ws_stat.p,
                         (int64_t)nb, (int64_t)P->n_units, (int64_t)P->batch, P->d_stat.p, skip_ptr, skip_stride);
      HIPCHK(ctx, hipGetLastError());
    }
#ifdef GAT_DBG_QUEUE
    { unsigned long long w[16]; hipStreamSynchronize(ctx->stream); hipMemcpy(w, P->d_stat.p, 16 * 8, hipMemcpyDeviceToHost);
      fprintf(stderr, "round broken by: empty segment %llu, placeholder neighbour %llu, both neighbours and more %llu, two on the right %llu, two logged %llu, logged + neighbour %llu\n", w[10], w[11], w[12], w[13], w[14], w[15]); }
#endif
    HIPCHK(ctx, hipMemcpyAsync(h_stat, P->d_stat.p, 10 * 8, hipMemcpyDeviceToHost, ctx->stream));   // (word 9: the queue's length)
    if (defer) return GAT_OK;
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return finish_sampler_batch(ctx, P, nb, st, timed, h_stat);
  }
}

// the checks and statistics of a sampler batch whose kernels have completed (the stream has been synchronised)
static int finish_sampler_batch(gat_ctx* ctx, gat_problem* P, int64_t nb, gat_stats* st, bool timed, const unsigned long long* h_stat) {
  {
    int rc;
    const int32_t flags = *reinterpret_cast<const int32_t*>(h_stat + 8);
    const unsigned long long* stat = h_stat;
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
      // (doubled until the lists fit: a unit whose segments are longer than its whole workspace beside one of a few bases places
      //  thousands of the small one between the large one's trims -- two of 200 000 edge-case seeds needed more than 64 x 128 slots,
      //  which the reference's lists simply grow to (round 6); the per-sample slab's 2^31 segments and the device's memory end it)
      if (P->cap_scale >= (1 << 20)) return set_err(ctx, GAT_ERR_CAPACITY, "sampler slab overflow even at 2^20 x capacity");
      P->cap_scale *= 2;
      if (layout_slab(P)) return set_err(ctx, GAT_ERR_CAPACITY, "per-sample slab exceeds 2^31 segments after growth");
      if ((rc = upload_layout(ctx, P))) return rc;
      if (st) st->n_retried += nb * (int64_t)P->h_order.size();
      return kRelayout;
    }
    {
      // k_units_overlap: overlaps between the units' lists that were not pairwise, or more candidates than the buffer holds --
      // the batch is repeated through k_contig, and the problem keeps to that
      const uint32_t* uw = reinterpret_cast<const uint32_t*>(h_stat + 10);
      if (P->units_direct && P->units_direct_ok && uw[0] != 0u) {
        if ((uw[0] & 1u) != 0u || P->cand_scale >= 64) P->units_direct_ok = false;   // not pairwise (or a buffer that will not do)
        else P->cand_scale *= 4;                                                       // a candidate region was too small
        if (st) st->n_retried += nb * (int64_t)P->h_order.size();
        return kRelayout;
      }
      if (st && P->units_direct) { st->n_straddle_candidates += (int64_t)uw[2]; st->n_unit_overlaps += (int64_t)uw[1]; }
    }
#if defined(GAT_DIAG) || defined(GAT_DIAG_CONS)
    if (const char* fn = gat_opt(ctx, "GAT_DIAG_OUT")) {
      // shares of a k_sampler work unit's life per phase, summed over the batch (tools/diag_sampler.sh)
      const size_t nd = (size_t)nb * std::max(1, P->n_units) * 8;
      std::vector<unsigned long long> h(nd);
      HIPCHK(ctx, staged_d2h(ctx, h.data(), P->d_diag.p, nd * 8));
      double sum[8] = {0, 0, 0, 0, 0, 0, 0, 0};
      for (size_t i = 0; i < nd; ++i) sum[i & 7] += (double)h[i];
      if (FILE* f = fopen(fn, "a")) {
        fprintf(f, "{\"work_units\": %lld, \"cycles\": {\"prologue\": %.0f, \"sort\": %.0f, \"merge\": %.0f, \"coverage\": %.0f, "
                   "\"fast_paths\": %.0f, \"trim\": %.0f, \"draws_placement\": %.0f, \"final_filter_write\": %.0f}}\n",
                (long long)nb * (long long)P->h_order.size(), sum[0], sum[1], sum[2], sum[3], sum[4], sum[5], sum[6], sum[7]);
        if (P->d_diag_place.n >= P->h_order.size() * 8 && !P->h_order.empty()) {
          // k_place's loop, per phase: over all tiles, and for the largest unit alone (launch position 0: the tile the kernel ends with)
          std::vector<unsigned long long> hp(P->h_order.size() * 8);
          if (staged_d2h(ctx, hp.data(), P->d_diag_place.p, hp.size() * 8) == hipSuccess) {
            double all[7] = {0, 0, 0, 0, 0, 0, 0};
            for (size_t i = 0; i < hp.size(); ++i) all[i & 7] += (i & 7) < 7 ? (double)hp[i] : 0.0;
            fprintf(f, "{\"k_place\": {\"all_tiles\": {\"row_wait\": %.0f, \"lookups\": %.0f, \"steps\": %.0f, \"flush\": %.0f, \"loop_control\": %.0f, "
                       "\"rows\": %.0f, \"tiles\": %.0f}, \"largest_unit\": {\"row_wait\": %llu, \"lookups\": %llu, \"steps\": %llu, \"flush\": %llu, "
                       "\"loop_control\": %llu, \"rows\": %llu, \"tiles\": %llu}}}\n",
                    all[0], all[1], all[2], all[3], all[4], all[5], all[6], hp[0], hp[1], hp[2], hp[3], hp[4], hp[5], hp[6]);
          }
        }
        {
          // k_place's hand-over records by kind: pending length (triggered), -1 run in full, -2 complete, -3 resume
          const size_t n_act = P->h_order.size();
          std::vector<int4> hs(n_act * (size_t)P->batch);
          if (n_act && P->d_st.n >= hs.size() && staged_d2h(ctx, hs.data(), P->d_st.p, hs.size() * sizeof(int4)) == hipSuccess) {
            long long kinds[5] = {0, 0, 0, 0, 0};
            int shown = 0;
            fprintf(f, "{\"k_place_handover_examples\": [");
            for (size_t a = 0; a < n_act; ++a)
              for (int64_t i = 0; i < nb; ++i) {
                const int4 r = hs[a * (size_t)P->batch + (size_t)i];
                kinds[r.z > 0 ? 0 : (r.z == -1 ? 1 : (r.z == -2 ? 2 : (r.z == -3 ? 3 : 4)))]++;
                if (r.z <= 0 && shown < 12) fprintf(f, "%s[%zu, %lld, %d, %d, %d, %d, %d]", shown++ ? ", " : "", a, (long long)i, r.x, r.y, r.z, r.w, P->h_rng_rows.empty() ? -1 : P->h_rng_rows[a]);
              }
            {
              long long sx = 0, sy = 0, sz = 0, sw = 0;
              for (size_t a = 0; a < n_act; ++a) for (int64_t i = 0; i < nb; ++i) { const int4 r = hs[a * (size_t)P->batch + (size_t)i]; sx += r.x; sy += r.y; sz += r.z; sw += r.w; }
              fprintf(f, "], \"k_place_handover_sums\": [%lld, %lld, %lld, %lld", sx, sy, sz, sw);
            }
            fprintf(f, "], \"k_place_handover\": {\"triggered\": %lld, \"in_full\": %lld, \"complete\": %lld, \"resume\": %lld, \"other\": %lld}}\n",
                    kinds[0], kinds[1], kinds[2], kinds[3], kinds[4]);
          }
        }
        {
          // k_place's tiles in time: per unit (launch position) the first begin, the median begin, the median and the last end of
          // its tiles, in microseconds from the kernel's first begin
          const size_t n_act = P->h_order.size(), ntl = (size_t)((nb + 63) / 64);
          std::vector<unsigned long long> ht(n_act * ntl * 2);
          if (n_act && ntl && P->d_diag_tiles.n >= ht.size() && staged_d2h(ctx, ht.data(), P->d_diag_tiles.p, ht.size() * 8) == hipSuccess) {
            unsigned long long t_min = ~0ull;
            for (size_t i = 0; i < ht.size(); i += 2) if (ht[i] && ht[i] < t_min) t_min = ht[i];
            fprintf(f, "{\"k_place_tiles\": [");
            for (size_t a = 0; a < n_act; ++a) {
              std::vector<double> b0, e0;
              for (size_t t = 0; t < ntl; ++t) {
                const unsigned long long x = ht[(a * ntl + t) * 2], y = ht[(a * ntl + t) * 2 + 1];
                if (x && y) { b0.push_back((double)(x - t_min) / 100.0); e0.push_back((double)(y - t_min) / 100.0); }
              }
              if (b0.empty()) continue;
              std::sort(b0.begin(), b0.end()); std::sort(e0.begin(), e0.end());
              fprintf(f, "%s{\"a\": %zu, \"segments\": %u, \"tiles\": %zu, \"begin_first\": %.1f, \"begin_median\": %.1f, \"begin_last\": %.1f, "
                         "\"end_first\": %.1f, \"end_median\": %.1f, \"end_last\": %.1f}", a ? ", " : "", a,
                      P->h_units[(size_t)P->h_order[a]].hist_total, b0.size(), b0.front(), b0[b0.size() / 2], b0.back(), e0.front(), e0[e0.size() / 2], e0.back());
            }
            fprintf(f, "]}\n");
          }
        }
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
      st->n_resumed_units += (int64_t)stat[5];
      st->n_queued_units += (int64_t)(stat[9] & 0xffffffffull);
      if (timed) {
        // (a timing query that fails leaves its figure at 0: it must not turn a call that computed its counts into an error)
        auto lap = [](hipEvent_t a, hipEvent_t b) { float ms = 0; if (hipEventElapsedTime(&ms, a, b) != hipSuccess) { (void)hipGetLastError(); ms = 0; } return ms; };
        st->ms_sampler += lap(ctx->ev[0], ctx->ev[1]);
        st->ms_contig += lap(ctx->ev[1], ctx->ev[2]);
        if (ctx->k_recorded && !P->h_order.empty()) {
          st->ms_rng += lap(ctx->ev[0], ctx->ev_k[0]);
          st->ms_place += lap(ctx->ev_k[0], ctx->ev_k[1]);
          st->ms_merge += lap(ctx->ev_k[1], ctx->ev_k[2]);
          st->ms_tail += lap(ctx->ev_k[2], ctx->ev_k[3]);
          if (ctx->t_recorded) {
            st->ms_ktail += lap(ctx->ev_k[2], ctx->ev_t[0]);
            st->ms_finalize += lap(ctx->ev_t[0], ctx->ev_t[1]);
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
  if (P->split_ran && P->units_direct) {       // k_contig only concatenated: k_units_overlap mends (it reads the units' lists themselves)
    A.contig_unit_off = P->d_contig_unit_off.p;
    A.cu_rec = P->d_cu_rec.p;
    A.unit_n = P->d_unit_n.p;
    A.seg_units = P->final_slab();
    A.seg_units_merged = P->d_slab.p;
    A.cand = P->d_cand.p; A.cand_cap = (uint32_t)(P->d_cand.n / gat::kCandSlots);
    A.cand_count = P->d_cand_count.p;
    A.st2 = P->d_st2.p;
    A.patch = reinterpret_cast<const int32_t*>(P->d_patch.p);
    A.patch_stride = (int32_t)(sizeof(gat::TailPatch) / 4);
    A.n_units = P->n_units;
    A.rec_stride = (int32_t)P->batch;
  }
  if (P->split_ran && P->patched_counts) {     // no k_finalize ran: k_count_seg<.., PATCH> reads merged lists + records
    A.seg_merged = P->d_slab.p;
    A.unit_pos = P->d_unit_pos.p;
    A.st2 = P->d_st2.p;
    A.patch = reinterpret_cast<const int32_t*>(P->d_patch.p);
    A.patch_stride = (int32_t)(sizeof(gat::TailPatch) / 4);
    A.n_units = P->n_units;
    A.rec_stride = (int32_t)P->batch;
  }
}

// ---- the batch seam: enqueue / wait ---------------------------------------------------------------------------
// A call is cut into batches that fit the scratch budget.  call_begin enqueues them one behind the other on the context's
// stream -- sampler kernels, count kernels, the copy of the batch's status word and statistics into a pinned slot of its own
// -- up to kMaxInflight of them, and returns; the host does what else it has to do.  call_wait synchronises once, reads the
// slots in order, and where a batch has to be repeated (a unit's region overflowed, a contig's lists beyond the launch's
// LDS) lays the slab out again and enqueues that batch and everything behind it once more: results do not depend on the
// batching (streams are per unit) and every batch writes its own columns of the count matrix.
// sample lists much longer than the annotation lists they meet: swap the roles in the count kernel (needs the tables' size)
static void decide_swap(gat_problem* P) {
  if (P->swap_decided) return;
  P->swap_decided = true;
  const int64_t total = annotations_ready(P->anno) ? P->anno->dev.total : P->anno->total_known;   // (callers: ready, or total_known >= 0)
  const double avg_n = P->n_contigs ? (double)P->n_seg_total / P->n_contigs : 0.0;
  const double avg_m = (P->n_contigs && P->n_tracks) ? (double)total / ((double)P->n_contigs * P->n_tracks) : 0.0;
  if (avg_m > 0 && avg_n > 3.0 * avg_m) P->swap_capx = 1;       // capacity is taken from the slab layout at launch
}

static int call_swap_capx(const gat_ctx* ctx, const gat_problem* P) {
  if (!P->swap_capx) return 0;
  const int capx = P->merge_contigs ? P->max_contig_cap : P->max_unit_cap;
  return ((int64_t)3 * capx * 4 + (8192 + 1) * 4 <= (int64_t)ctx->max_lds - 1024) ? capx : 0;
}

// block: wait for annotation tables that are still being built (gat_wait); else return with the newest batch's count
// kernels still to come (gat_sample_and_count_enqueue: the host goes on, the device samples)
static int call_enqueue_more(gat_ctx* ctx, gat_problem* P, bool block) {
  CallState& K = P->call;
  Counters C;
  int rc = parse_counters(ctx, K.ids, K.n_counters, C);
  if (rc) return rc;
  const bool serial = K.state_host != nullptr;
  uint32_t* d_state = serial ? P->d_serial.p : nullptr;
  // (the events behind every kernel and the serial stream's saved state exist once: such calls keep one batch in flight)
  const int max_flight = (K.timed || serial) ? 1 : kMaxInflight;
  PrepTimer tm;
  for (;;) {
    if (K.count_pending) {
      // ---- the count kernels of the newest batch
      if (!annotations_ready(P->anno) && !block) break;
      if ((rc = annotations_wait(ctx, P->anno))) return rc;
      decide_swap(P);
      const int slot = K.n_flight - 1;
      const int64_t nb = K.nb[slot];
      const int swap_capx = call_swap_capx(ctx, P);
      K.mstat_on = P->anno->dev.has_merged;
      if (K.mstat_on && K.enq == nb) {
        // (k_count_merged's own traffic counters: only a problem with a merged index can take that kernel; zeroed in front of
        //  the call's first count kernel)
        if (P->d_mstat.n < 512) HIPCHK(ctx, P->d_mstat.alloc(512));
        HIPCHK(ctx, hipMemsetAsync(P->d_mstat.p, 0, 512 * 8, ctx->stream));
      }
      gat::CountArgs A;
      memset(&A, 0, sizeof(A));
      fill_count_args(P, A, nb);
      A.out = (int64_t*)K.counts_dev;
      A.out_stride = K.S;
      A.out_begin = K.enq - nb;
      A.mstat = K.mstat_on ? P->d_mstat.p : nullptr;
      if (K.timed) HIPCHK(ctx, hipEventRecord(ctx->ev_cnt[0], ctx->stream));
      CountLaunch L;
      L.ev_main = K.blk->ev_main[slot];
      L.units = P->d_units.p; L.ws = P->d_ws.p; L.ws_tree = P->d_ws_tree.p;
      for (int c = 0; c < P->n_contigs; ++c) L.max_units = std::max(L.max_units, P->h_contig_unit_off[(size_t)c + 1] - P->h_contig_unit_off[(size_t)c]);
      if ((rc = launch_count(ctx, P->anno->dev, C, A, P->d_part, swap_capx,
                             P->merge_contigs ? P->max_contig_cap : P->max_unit_cap, L))) return rc;
      if (A.cu_rec != nullptr)       // k_units_overlap's words (candidates, "not pairwise", overlaps taken off) into the batch's slot
        HIPCHK(ctx, hipMemcpyAsync(K.blk->h_stat + (size_t)slot * 16 + 10, P->d_cand_count.p + gat::kCandSlots, 16, hipMemcpyDeviceToHost, ctx->stream));
      else { K.blk->h_stat[(size_t)slot * 16 + 10] = 0; K.blk->h_stat[(size_t)slot * 16 + 11] = 0; }
      if (K.timed) HIPCHK(ctx, hipEventRecord(ctx->ev_cnt[1], ctx->stream));
      K.main_rec[slot] = L.main_recorded;
      K.count_kernel[slot] = L.count_kernel;
      K.count_pending = false;
      if (K.enq == K.S) {
        // (the call's end rides on the last batch's synchronisation -- one round trip to the device less per call; a batch
        //  that has to be repeated enqueues it again)
        if (K.mstat_on) HIPCHK(ctx, hipMemcpyAsync(K.blk->h_mstat, P->d_mstat.p, 512 * 8, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(ctx, hipEventRecord(K.blk->ev_end, ctx->stream));
        K.end_recorded = true;
      }
    }
    if (!(K.enq < K.S && K.n_flight < max_flight)) break;
    // ---- the sampler kernels of the next batch
    // which count kernel will follow decides the sampler's last steps (final lists or not); with the tables still being built
    // it is known only where the shape alone says so -- many tracks, nucleotide counters: the merged index -- else wait
    int route;
    if (annotations_ready(P->anno)) {
      if ((rc = annotations_wait(ctx, P->anno))) return rc;
      decide_swap(P);
      route = count_route(ctx, P->anno->dev.has_merged, C, P->n_contigs, P->n_tracks, call_swap_capx(ctx, P), !P->anno->dev.per_track);
    } else {
      const bool only_overlap = C.slot[GAT_COUNTER_SEGMENT_OVERLAP] < 0 && C.slot[GAT_COUNTER_SEGMENT_MIDOVERLAP] < 0;
      const size_t lds_merged = (size_t)P->n_tracks * 4 * (gat::kMergedThreads / gat::kWave);
      if (C.any_seg && P->n_contigs > 0 && only_overlap && P->anno->will_merge && (int64_t)lds_merged + 1024 <= ctx->max_lds &&
          !gat_opt(ctx, "GAT_COUNT_NO_MERGED")) {
        route = GAT_COUNT_KERNEL_MERGED;                       // (whatever the swap decision would be)
      } else if (P->anno->shape_known && P->anno->total_known >= 0) {
        decide_swap(P);                                        // (from the sizes announced before the build)
        route = count_route(ctx, P->anno->will_merge, C, P->n_contigs, P->n_tracks, call_swap_capx(ctx, P));
      } else {
        if ((rc = annotations_wait(ctx, P->anno))) return rc;
        decide_swap(P);
        route = count_route(ctx, P->anno->dev.has_merged, C, P->n_contigs, P->n_tracks, call_swap_capx(ctx, P), !P->anno->dev.per_track);
      }
    }
    // the scratch is sized while nothing of this problem is in flight; batches behind the first fit by construction
    if (K.n_flight == 0 && (rc = ensure_scratch(ctx, P, K.S - K.enq))) return rc;
    const int64_t nb = std::min<int64_t>(P->batch, K.S - K.enq);
    const int slot = K.n_flight;
    if (d_state != nullptr)
      HIPCHK(ctx, hipMemcpyAsync(d_state + GAT_MT_STATE_WORDS, d_state, GAT_MT_STATE_WORDS * 4, hipMemcpyDeviceToDevice, ctx->stream));
    // counts alone, all of them k_count_seg's: it reads the units as k_tail left them (no final lists are written)
    const bool records_ok = !C.any_anno && !gat_opt(ctx, "GAT_COUNT_FINAL_LISTS") &&
                            (route == GAT_COUNT_KERNEL_SEG || route == GAT_COUNT_KERNEL_MERGED);
    // (k_count_merged skips empty segments: long lists may keep what a trim emptied, no compaction pass in k_resume_big)
    const bool loose_ok = records_ok && route == GAT_COUNT_KERNEL_MERGED;
    // isochore problems: the merged index takes the units' lists as they are, no k_contig (round 6)
    const bool units_direct = loose_ok && P->merge_contigs && P->units_direct_ok && !gat_opt(ctx, "GAT_COUNT_VIA_CONTIGS");
    if (units_direct) {
      // candidates: segments with a workspace boundary in their cells, of units that have a segment reaching out of their workspace
      // -- a fraction of a per cent of a batch's segments on isochore blocks much longer than the segments --, dealt to kCandSlots
      // regions by (sample, contig).  Sized for THIS batch (a call's first batch is its largest; a later, larger call makes it
      // anew while nothing of the problem is in flight); a region that overflows anyway has the batch repeated with four
      // times the buffer (cand_scale), and beyond 64 times through the sorted contig lists
      double est = 0.006 * (double)nb * (double)std::max<int64_t>(1, P->n_seg_total) / gat::kCandSlots + 512.0;
      if (gat_opt(ctx, "GAT_TEST_SMALL_CAPS")) est = 1.0;        // (tests: regions that overflow, the batch repeated with larger ones)
      const size_t want = (size_t)std::min(est * P->cand_scale, 1024.0 * 1024) * gat::kCandSlots;
      if (P->d_cand.n < want && K.n_flight == 0) {
        if (P->d_cand.n > 0) HIPCHK(ctx, hipStreamSynchronize(ctx->stream));   // (an earlier call's kernels may still read the old buffer)
        HIPCHK(ctx, P->d_cand.alloc(want));
      }
      if (P->d_cand_count.n == 0) {
        HIPCHK(ctx, P->d_cand_count.alloc(gat::kCandSlots + 4));
        HIPCHK(ctx, hipMemsetAsync(P->d_cand_count.p, 0, (gat::kCandSlots + 4) * 4, ctx->stream));
      }
    }
    if ((rc = run_sampler_batch(ctx, P, K.seed, K.begin + K.enq, nb, &K.local, K.timed, false, true, records_ok, d_state, loose_ok,
                                K.blk->h_stat + (size_t)slot * 16, units_direct))) return rc;   // (enqueued only)
    K.nb[slot] = nb;
    K.main_rec[slot] = false;
    K.count_kernel[slot] = GAT_COUNT_KERNEL_NONE;
    K.n_flight = slot + 1;
    K.enq += nb;
    K.count_pending = true;
  }
  tm.lap("batches enqueued");
  return GAT_OK;
}

static void call_end(gat_ctx* ctx, gat_problem* P) {
  CallState& K = P->call;
  if (K.blk) ctx->call_blocks.push_back(K.blk);
  K.blk = nullptr;
  K.active = false;
  if (ctx->timed_owner == (const void*)P) ctx->timed_owner = nullptr;
}

static int call_begin(gat_ctx* ctx, gat_problem* P, const int32_t* counter_ids, int n_counters, uint32_t seed,
                      int64_t sample_begin, int64_t sample_end, void* counts_dev, uint32_t* state_host) {
  if (!ctx || !P || !counts_dev || (n_counters > 0 && !counter_ids)) return set_err(ctx, GAT_ERR_ARG, "gat_sample_and_count: NULL argument");
  if (sample_end < sample_begin) return set_err(ctx, GAT_ERR_ARG, "sample_end < sample_begin");
  if (n_counters < 0 || n_counters > GAT_NUM_COUNTERS) return set_err(ctx, GAT_ERR_ARG, "%d counters (0..%d)", n_counters, GAT_NUM_COUNTERS);
  if (P->ctx != ctx) return set_err(ctx, GAT_ERR_ARG, "the problem was created on another context");
  CallState& K = P->call;
  if (K.active) return set_err(ctx, GAT_ERR_ARG, "a call is in flight on this problem: gat_wait first");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  Counters C;
  int rc = parse_counters(ctx, counter_ids, n_counters, C);
  if (rc) return rc;
  if (P->sampler == GAT_SAMPLER_SEGMENTS && !P->merge_contigs && n_counters > 0)
    return set_err(ctx, GAT_ERR_ASSERT, "SamplerSegments output is not normalized unless fromIsochores merges it "
                   "(keys without isochores): the counters assert (gat/SegmentList.pyx:1031)");
  if ((rc = call_block_get(ctx, &K.blk))) return rc;
  K.active = true;
  for (int i = 0; i < n_counters; ++i) K.ids[i] = counter_ids[i];
  K.n_counters = n_counters;
  K.seed = seed; K.begin = sample_begin; K.S = sample_end - sample_begin;
  K.counts_dev = counts_dev; K.state_host = state_host;
  K.done = K.enq = 0; K.n_flight = 0;
  memset(&K.local, 0, sizeof(K.local));
  // (an event behind every kernel of the sampler costs 50-60 us of a call: 2 % at 10 000 samples of config 2, 6 % at 1 250)
  const bool times_wanted = ctx->kernel_times || gat_opt(ctx, "GAT_KERNEL_TIMES") != nullptr;
  K.timed = times_wanted && ctx->timed_owner == nullptr;
  if (K.timed) ctx->timed_owner = (const void*)P;
  // (the per-kernel events exist once per context: a call enqueued while another problem's timed call is in flight runs
  //  untimed -- its ms_* split reads 0 -- and says so: gat_stats::kernel_times)
  K.times_state = !times_wanted ? 0 : (K.timed ? 1 : -1);
  K.mstat_on = false;                              // (set with the first count kernels: it takes the tables)
  K.count_pending = false;
  K.end_recorded = false;
  auto fail = [&](int code) { call_end(ctx, P); return code; };
  if (hipEventRecord(K.blk->ev_begin, ctx->stream) != hipSuccess) return fail(set_err(ctx, GAT_ERR_DEVICE, "hipEventRecord failed"));
  if (state_host != nullptr) {
    // the run's one stream: its state lives on the device over the batches (a copy restores it when a batch is repeated)
    if (P->d_serial.n < 2 * (size_t)GAT_MT_STATE_WORDS && P->d_serial.alloc(2 * (size_t)GAT_MT_STATE_WORDS) != hipSuccess)
      return fail(set_err(ctx, GAT_ERR_MEMORY, "device memory for the stream's state"));
    if (staged_h2d(ctx, P->d_serial.p, state_host, GAT_MT_STATE_WORDS * 4) != hipSuccess) return fail(set_err(ctx, GAT_ERR_DEVICE, "upload of the stream's state failed"));
  }
  if (K.S == 0) {                                                  // (no batch will run: nothing to ride on)
    if (hipEventRecord(K.blk->ev_end, ctx->stream) != hipSuccess) return fail(set_err(ctx, GAT_ERR_DEVICE, "hipEventRecord failed"));
    K.end_recorded = true;
    return GAT_OK;
  }
  if ((rc = call_enqueue_more(ctx, P, false))) return fail(rc);
  return GAT_OK;
}

static int call_wait(gat_ctx* ctx, gat_problem* P, gat_stats* stats) {
  if (!ctx || !P) return set_err(ctx, GAT_ERR_ARG, "gat_wait: NULL argument");
  CallState& K = P->call;
  if (!K.active) return set_err(ctx, GAT_ERR_ARG, "gat_wait: no call in flight on this problem");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  auto fail = [&](int code) { (void)hipStreamSynchronize(ctx->stream); call_end(ctx, P); return code; };
  PrepTimer tm;
  int rc;
  if (K.count_pending && (rc = call_enqueue_more(ctx, P, true))) return fail(rc);   // (the tables were still being built)
  for (;;) {
    // ONE synchronisation per flight of batches: a sampler's status word is read behind the count kernels, which ran on
    // whatever an overflowed unit left.  What it left is in bounds: the exits that set a status bit (region full, a contig's
    // lists beyond the launch's LDS) leave unit_n / contig_n / the hand-over records as an EARLIER batch of this layout wrote
    // them -- lengths within the regions of this layout -- or as ensure_scratch zeroed them: a new layout (layout_slab +
    // upload_layout) sets P->batch = 0, so the scratch is sized and zeroed again before the next kernel runs, and no length
    // written under another layout survives into this one.  The counts of such a batch, and of the batches enqueued behind
    // it, are thrown away (they are redone)
    // (the call's own end where it is on the stream -- its last batch enqueued --, not the stream's: another problem's call
    //  enqueued behind this one -- run() keeps two segment tracks in flight, bench.py two steps -- goes on running while the
    //  host reads this one's status words and enqueues the next)
    const hipError_t se = K.end_recorded ? hipEventSynchronize(K.blk->ev_end) : hipStreamSynchronize(ctx->stream);
    if (se != hipSuccess) return fail(set_err(ctx, GAT_ERR_DEVICE, "synchronising with the call's batches failed: %s", hipGetErrorString(hipGetLastError())));
    K.end_recorded = false;
    tm.lap("batches synchronised");
    for (int slot = 0; slot < K.n_flight; ++slot) {
      rc = finish_sampler_batch(ctx, P, K.nb[slot], &K.local, K.timed, K.blk->h_stat + (size_t)slot * 16);
      if (rc == kRelayout) {
        if (K.state_host != nullptr)        // (the repeated batch draws from where this one began)
          if (hipMemcpyAsync(P->d_serial.p, P->d_serial.p + GAT_MT_STATE_WORDS, GAT_MT_STATE_WORDS * 4, hipMemcpyDeviceToDevice, ctx->stream) != hipSuccess)
            return fail(set_err(ctx, GAT_ERR_DEVICE, "restoring the stream's state failed"));
        break;
      }
      if (rc) return fail(rc);
      float ms = 0;
      if (K.timed) {
        if (hipEventElapsedTime(&ms, ctx->ev_cnt[0], ctx->ev_cnt[1]) == hipSuccess) K.local.ms_count += ms;
      }
      if (K.main_rec[slot] && hipEventElapsedTime(&ms, K.blk->ev_main[slot][0], K.blk->ev_main[slot][1]) == hipSuccess)
        K.local.ms_count_main += ms;
      K.local.count_kernel = K.count_kernel[slot];
      K.local.merged_form = K.count_kernel[slot] == GAT_COUNT_KERNEL_MERGED ? P->anno->dev.merged_block : 0;
      K.local.n_batches += 1;
      K.done += K.nb[slot];
    }
    K.n_flight = 0;
    K.enq = K.done;                                              // (what a repeated batch had behind it is enqueued again)
    if (K.done == K.S) break;
    if ((rc = call_enqueue_more(ctx, P, true))) return fail(rc);
  }
  if (K.state_host != nullptr && staged_d2h(ctx, K.state_host, P->d_serial.p, GAT_MT_STATE_WORDS * 4) != hipSuccess)
    return fail(set_err(ctx, GAT_ERR_DEVICE, "read-back of the stream's state failed"));
  if (K.mstat_on && K.S > 0)
    for (int i = 0; i < 256; ++i) { K.local.n_index_entries += (int64_t)K.blk->h_mstat[2 * i]; K.local.n_index_lookups += (int64_t)K.blk->h_mstat[2 * i + 1]; }
  float ms = 0;
  if (hipEventElapsedTime(&ms, K.blk->ev_begin, K.blk->ev_end) == hipSuccess) K.local.ms_total = ms; else (void)hipGetLastError();
  K.local.kernel_times = K.times_state;
  if (stats) *stats = K.local;
  call_end(ctx, P);
  return GAT_OK;
}

extern "C" int gat_sample_and_count_enqueue(gat_ctx* ctx, gat_problem* P, const int32_t* counter_ids, int n_counters,
                                            uint32_t seed, int64_t sample_begin, int64_t sample_end, void* counts_dev) {
  return call_begin(ctx, P, counter_ids, n_counters, seed, sample_begin, sample_end, counts_dev, nullptr);
}

extern "C" int gat_wait(gat_ctx* ctx, gat_problem* P, gat_stats* stats) { return call_wait(ctx, P, stats); }

extern "C" int gat_sample_and_count(gat_ctx* ctx, gat_problem* P, const int32_t* counter_ids, int n_counters,
                                    uint32_t seed, int64_t sample_begin, int64_t sample_end, void* counts_dev,
                                    gat_stats* stats) {
  const int rc = call_begin(ctx, P, counter_ids, n_counters, seed, sample_begin, sample_end, counts_dev, nullptr);
  return rc ? rc : call_wait(ctx, P, stats);
}

extern "C" int gat_sample_and_count_serial(gat_ctx* ctx, gat_problem* P, const int32_t* counter_ids, int n_counters,
                                           uint32_t* mt_state, int64_t n_samples, void* counts_dev, gat_stats* stats) {
  if (!mt_state) return set_err(ctx, GAT_ERR_ARG, "gat_sample_and_count_serial: NULL state");
  if (mt_state[GAT_MT_STATE_WORDS - 1] > 624u) return set_err(ctx, GAT_ERR_ARG, "gat_sample_and_count_serial: position %u > 624", mt_state[GAT_MT_STATE_WORDS - 1]);
  const int rc = call_begin(ctx, P, counter_ids, n_counters, 0u, 0, n_samples, counts_dev, mt_state);
  return rc ? rc : call_wait(ctx, P, stats);
}

extern "C" void gat_mt19937_seed(uint32_t seed, uint32_t* mt_state) {
  // numpy.random.seed(int): init_genrand (numpy/random/src/mt19937/mt19937.c: mt19937_seed), position = 624
  uint32_t x = seed;
  for (int i = 0; i < 624; ++i) { mt_state[i] = x; x = 1812433253u * (x ^ (x >> 30)) + (uint32_t)(i + 1); }
  mt_state[624] = 624u;
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
  if (P->call.active) return set_err(ctx, GAT_ERR_ARG, "a call is in flight on this problem (its scratch is in use): gat_wait first");
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
    // (timed unless another problem's call in flight owns the context's per-kernel events)
    if ((rc = run_sampler_batch(ctx, P, seed, sample_begin + done, nb, &local, ctx->timed_owner == nullptr, unit_level)) == kRelayout) continue;
    if (rc) return rc;
    const bool from_contigs = P->merge_contigs && !unit_level;
    const uint2* src = from_contigs ? P->d_cslab.p : P->final_slab();
    const int32_t* nsrc = from_contigs ? P->d_contig_n.p : P->d_unit_n.p;
    const int nstride = from_contigs ? P->n_contigs : P->n_units;
    h_slab.resize((size_t)(nb * P->slab_stride));
    h_n.resize((size_t)(nb * std::max(1, nstride)));
    HIPCHK(ctx, staged_d2h(ctx, h_slab.data(), src, h_slab.size() * sizeof(uint2)));
    HIPCHK(ctx, staged_d2h(ctx, h_n.data(), nsrc, h_n.size() * 4));
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

static int count_lists_on(gat_ctx* ctx, const int32_t* counter_ids, int n_counters,
                          const gat_segment* lists, const int64_t* list_off, int64_t n_lists,
                          const gat_segment* annos, const int64_t* anno_begin, const int64_t* anno_end, int32_t n_tracks,
                          const int64_t* ws_nseg, int32_t n_groups, void* counts_host);

// Observed counts are independent of whatever samples are in flight on the caller's context: they run on a stream (and
// through a staging buffer) of their own -- every staged upload waits for its stream, and on the caller's stream that would
// be the end of the sampling (run() computes them while the device samples, gat/__init__.py:933-940)
static int count_lists_impl(gat_ctx* ctx, const int32_t* counter_ids, int n_counters,
                            const gat_segment* lists, const int64_t* list_off, int64_t n_lists,
                            const gat_segment* annos, const int64_t* anno_begin, const int64_t* anno_end, int32_t n_tracks,
                            const int64_t* ws_nseg, int32_t n_groups, void* counts_host) {
  if (!ctx || !list_off || !anno_begin || !anno_end || !counts_host || (n_groups > 0 && !ws_nseg))
    return set_err(ctx, GAT_ERR_ARG, "gat_count_lists: NULL argument");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  if (ctx->aux_ctx == nullptr) {
    int rc = gat_ctx_create(&ctx->aux_ctx, ctx->device, nullptr);
    if (rc == GAT_OK) ctx->aux_ctx->options_owner = ctx;              // (the knobs are its owner's)
    if (rc) return rc;
  }
  gat_ctx* x = ctx->aux_ctx;
  const int rc = count_lists_on(x, counter_ids, n_counters, lists, list_off, n_lists, annos, anno_begin, anno_end, n_tracks, ws_nseg,
                                n_groups, counts_host);
  x->stage_used = 0;                                // (count_lists_on has drained the stream before its buffers went back to the pool)
  if (rc) return set_err(ctx, rc, "%s", x->err.c_str());
  return GAT_OK;
}

// The device buffers of a gat_count_lists call.  They go back to the process-wide pool when they are destroyed, and the pool
// hands an idle block to any context or stream: on EVERY path out of the call -- an error in the middle of its launches
// too -- the stream is drained before that (count_lists_on owns them and synchronises behind count_lists_body: ADVICE r4)
struct CountListBufs {
  AnnoDev A;
  DevBuf<uint2> d_seg;
  DevBuf<int32_t> d_c_off, d_n, d_index;
  DevBuf<int64_t> d_nseg, d_out;
  DevBuf<uint32_t> d_part;
};
static int count_lists_body(gat_ctx* ctx, const int32_t* counter_ids, int n_counters,
                            const gat_segment* lists, const int64_t* list_off, int64_t n_lists,
                            const gat_segment* annos, const int64_t* anno_begin, const int64_t* anno_end, int32_t n_tracks,
                            const int64_t* ws_nseg, int32_t n_groups, void* counts_host, CountListBufs& B);
static int count_lists_on(gat_ctx* ctx, const int32_t* counter_ids, int n_counters,
                          const gat_segment* lists, const int64_t* list_off, int64_t n_lists,
                          const gat_segment* annos, const int64_t* anno_begin, const int64_t* anno_end, int32_t n_tracks,
                          const int64_t* ws_nseg, int32_t n_groups, void* counts_host) {
  CountListBufs B;
  const int rc = count_lists_body(ctx, counter_ids, n_counters, lists, list_off, n_lists, annos, anno_begin, anno_end, n_tracks,
                                  ws_nseg, n_groups, counts_host, B);
  (void)hipStreamSynchronize(ctx->stream);
  return rc;
}
static int count_lists_body(gat_ctx* ctx, const int32_t* counter_ids, int n_counters,
                            const gat_segment* lists, const int64_t* list_off, int64_t n_lists,
                            const gat_segment* annos, const int64_t* anno_begin, const int64_t* anno_end, int32_t n_tracks,
                            const int64_t* ws_nseg, int32_t n_groups, void* counts_host, CountListBufs& B) {
  Counters C;
  int rc = parse_counters(ctx, counter_ids, n_counters, C);
  if (rc) return rc;
  AnnoDev& A = B.A;
  // (the merged index pays when many lists are counted against it; for a handful -- the observed counts of a run's
  //  segment tracks -- building it costs more than the per-track kernel's extra look-ups)
  const bool want_merged = n_lists >= 16 || gat_opt(ctx, "GAT_COUNT_LISTS_MERGED") != nullptr;
  if ((rc = build_annos(ctx, A, annos, anno_begin, anno_end, (int64_t)n_tracks * n_groups, n_groups, want_merged, false))) return rc;
  for (int64_t l = 0; l < n_lists * n_groups; ++l)
    if ((rc = check_list(ctx, lists + list_off[l], list_off[l + 1] - list_off[l], "segment", l))) return rc;
  const int64_t total = list_off[n_lists * n_groups];
  std::vector<uint2> h_seg((size_t)total);
  for (int64_t i = 0; i < total; ++i) h_seg[(size_t)i] = make_uint2(lists[i].start, lists[i].end);
  DevBuf<uint2>& d_seg = B.d_seg;
  DevBuf<int32_t>&d_c_off = B.d_c_off, &d_n = B.d_n, &d_index = B.d_index;
  DevBuf<int64_t>&d_nseg = B.d_nseg, &d_out = B.d_out;
  DevBuf<uint32_t>& d_part = B.d_part;
  HIPCHK(ctx, d_seg.upload(h_seg, ctx));
  std::vector<int64_t> h_nseg(ws_nseg, ws_nseg + n_groups);
  HIPCHK(ctx, d_nseg.upload(h_nseg, ctx));
  const size_t nslots = (size_t)n_counters * n_tracks * n_lists;
  HIPCHK(ctx, d_out.alloc(nslots));
  HIPCHK(ctx, hipMemsetAsync(d_out.p, 0, std::max<size_t>(1, nslots) * 8, ctx->stream));
  std::vector<int32_t> h_index((size_t)n_groups);
  for (int g = 0; g < n_groups; ++g) h_index[(size_t)g] = g;
  HIPCHK(ctx, d_index.upload(h_index, ctx));
  // the group offsets and lengths of every list, uploaded once; one launch per list (they differ from list to list)
  std::vector<int32_t> h_c_off((size_t)std::max<int64_t>(1, n_lists * n_groups)), h_n((size_t)std::max<int64_t>(1, n_lists * n_groups));
  for (int64_t l = 0; l < n_lists; ++l) {
    const int64_t base = list_off[l * n_groups];
    for (int g = 0; g < n_groups; ++g) {
      h_c_off[(size_t)(l * n_groups + g)] = (int32_t)(list_off[l * n_groups + g] - base);
      h_n[(size_t)(l * n_groups + g)] = (int32_t)(list_off[l * n_groups + g + 1] - list_off[l * n_groups + g]);
    }
  }
  HIPCHK(ctx, d_c_off.upload(h_c_off, ctx));
  HIPCHK(ctx, d_n.upload(h_n, ctx));
  for (int64_t l = 0; l < n_lists; ++l) {
    const int64_t base = list_off[l * n_groups];
    gat::CountArgs K;
    memset(&K, 0, sizeof(K));
    K.seg = d_seg.p + base; K.seg_stride = 0; K.c_off = d_c_off.p + l * n_groups;
    K.n_arr = d_n.p + l * n_groups; K.n_stride = 0; K.n_index = d_index.p;
    K.cws_nseg = d_nseg.p; K.n_contigs = n_groups; K.n_tracks = n_tracks; K.n_samples = 1;
    K.out = d_out.p; K.out_stride = n_lists; K.out_begin = l;
    int32_t longest = 0;
    for (int g = 0; g < n_groups; ++g) longest = std::max(longest, h_n[(size_t)(l * n_groups + g)]);
    CountLaunch L;
    L.ev_main = ctx->ev_main;
    if ((rc = launch_count(ctx, A, C, K, d_part, 0, longest, L))) return rc;
  }
  HIPCHK(ctx, staged_d2h(ctx, counts_host, d_out.p, nslots * 8));
  return GAT_OK;
}

extern "C" int gat_count_lists(gat_ctx* ctx, const int32_t* counter_ids, int n_counters,
                               const gat_segment* lists, const int64_t* list_off, int64_t n_lists,
                               const gat_segment* annos, const int64_t* anno_off, int32_t n_tracks,
                               const int64_t* ws_nseg, int32_t n_groups, void* counts_host) {
  if (!anno_off) return set_err(ctx, GAT_ERR_ARG, "gat_count_lists: NULL argument");
  return count_lists_impl(ctx, counter_ids, n_counters, lists, list_off, n_lists, annos, anno_off, anno_off + 1, n_tracks, ws_nseg,
                          n_groups, counts_host);
}

extern "C" int gat_count_list_ranges(gat_ctx* ctx, const int32_t* counter_ids, int n_counters,
                                     const gat_segment* lists, const int64_t* list_off, int64_t n_lists,
                                     const gat_segment* annos, const int64_t* anno_begin, const int64_t* anno_end, int32_t n_tracks,
                                     const int64_t* ws_nseg, int32_t n_groups, void* counts_host) {
  return count_lists_impl(ctx, counter_ids, n_counters, lists, list_off, n_lists, annos, anno_begin, anno_end, n_tracks, ws_nseg,
                          n_groups, counts_host);
}

// ------------------------------------------------------------------------------------------
// The one collective of the path: all-gather of the per-rank count blocks over xGMI (RCCL), replacing the reference's
// pool result collation (gat/__init__.py:681-700, :770-774).  RCCL is loaded lazily so that single-GPU hosts need not
// have it (and so that a process that already holds torch's copy does not get a second one by linking).
struct RcclApi {
  void* handle = nullptr;
  bool preloaded = false;      // the library was in the process already (torch's copy, or one the host linked)
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
    // A process must not hold two RCCLs (two sets of communicators' bootstrap state, two IPC caches): where one is mapped
    // already -- torch brings its own copy, torch/lib/librccl.so, and resolves it by that name -- that one is taken:
    // RTLD_NOLOAD returns a handle only for a library that is loaded, by soname or by the path it was loaded from.  Failing
    // that, any object of the process that exports the entry points (RTLD_DEFAULT: a host that linked RCCL itself); only then
    // is a library loaded afresh.  GAT_RCCL_LIB names one explicitly.
    const char* env_lib = gat_opt(nullptr, "GAT_RCCL_LIB");
    if (env_lib && *env_lib) api.handle = dlopen(env_lib, RTLD_NOW | RTLD_LOCAL);
    if (!api.handle)
      for (const char* name : {"librccl.so", "librccl.so.1"}) {
        api.handle = dlopen(name, RTLD_NOW | RTLD_LOCAL | RTLD_NOLOAD);
        if (api.handle) break;
      }
    if (!api.handle) {
      // (mapped under a path dlopen does not search -- torch/lib --: find it among the process's objects)
      struct Find { std::string path; } f;
      dl_iterate_phdr([](struct dl_phdr_info* info, size_t, void* data) -> int {
        const char* nm = info->dlpi_name;
        if (nm && strstr(nm, "librccl.so")) { static_cast<Find*>(data)->path = nm; return 1; }
        return 0;
      }, &f);
      if (!f.path.empty()) api.handle = dlopen(f.path.c_str(), RTLD_NOW | RTLD_LOCAL | RTLD_NOLOAD);
    }
    api.preloaded = api.handle != nullptr;
    if (!api.handle && dlsym(RTLD_DEFAULT, "ncclAllGather") != nullptr) { api.handle = RTLD_DEFAULT; api.preloaded = true; }
    if (!api.handle)
      for (const char* name : {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"}) {
        api.handle = dlopen(name, RTLD_NOW | RTLD_LOCAL);
        if (api.handle) break;
      }
    if (api.handle || api.preloaded) {
      api.get_unique_id = (decltype(api.get_unique_id))dlsym(api.handle, "ncclGetUniqueId");
      api.comm_init_rank = (decltype(api.comm_init_rank))dlsym(api.handle, "ncclCommInitRank");
      api.comm_destroy = (decltype(api.comm_destroy))dlsym(api.handle, "ncclCommDestroy");
      api.all_gather = (decltype(api.all_gather))dlsym(api.handle, "ncclAllGather");
      api.error_string = (decltype(api.error_string))dlsym(api.handle, "ncclGetErrorString");
    }
  }
  const bool ok = (api.handle || api.preloaded) && api.get_unique_id && api.comm_init_rank && api.comm_destroy && api.all_gather;
  return ok ? &api : nullptr;
}

// which RCCL gat_comm_* resolved: 1 one the process had mapped already (RTLD_NOLOAD / among its objects), 0 loaded afresh, -1 none
extern "C" int gat_comm_library_preloaded(void) {
  RcclApi* R = rccl_api();
  return R ? (R->preloaded ? 1 : 0) : -1;
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
  HIPCHK(ctx, d_off.upload(leaf_off, ctx));
  HIPCHK(ctx, d_len.upload(leaf_len, ctx));
  HIPCHK(ctx, d_prog.upload(prog, ctx));
  HIPCHK(ctx, d_dbl.upload(std::vector<uint8_t>(is_double_host, is_double_host + n_rows), ctx));
  HIPCHK(ctx, d_vals.upload(std::vector<double>(vals_host, vals_host + n_rows), ctx));
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
  HIPCHK(ctx, staged_d2h(ctx, out_host, d_out.p, (size_t)n_rows * 64));
  return GAT_OK;
}
