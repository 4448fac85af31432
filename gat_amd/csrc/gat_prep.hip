// gat_prep.hip -- problem creation: what the reference does once per (segments, workspace) pair before sampling starts
// (gat/Engine.pyx:543-565: filter, ltotal, length histogram, both sampler CDFs -- hoisted out of the per-sample loop),
// the layout of the per-sample slab, and the look-up tables of the count kernels (SoA + position grids, the merged
// index).  Host code only: nothing here launches a kernel or computes a sample.
#include <chrono>
#include <climits>
#include <cmath>
#include <map>
#include <memory>

#include <condition_variable>
#include <mutex>
#include <unistd.h>

#include "gat_host.h"

// ------------------------------------------------------------------------------------------
// the device-memory pool behind DevBuf (gat_host.h)
namespace {
struct PoolBlock { void* p; size_t bytes; unsigned long long stamp; };
// blocks: idle ones; out: what the blocks handed out really hold (a request is served by a block up to a quarter larger: it
// comes back under its own size, not the request's, so the books -- held, the GAT_POOL_BYTES cap, dev_pool_held() -- stay true)
struct DevPool { std::vector<PoolBlock> blocks; size_t held = 0; std::map<void*, size_t> out; unsigned long long clock = 0; };
std::mutex g_pool_mutex;
std::map<int, DevPool> g_pools;
constexpr size_t kPoolMinBytes = (size_t)1 << 20;
size_t pool_limit() {
  static const size_t limit = [] {
    const char* env = gat_opt(nullptr, "GAT_POOL_BYTES");
    return env ? (size_t)atof(env) : (size_t)96 << 30;
  }();
  return limit;
}
}  // namespace

// small requests are rounded up to a power of two so that the next one of that class finds the block again: in the
// steady state of a host that creates problem after problem NOTHING reaches hipMalloc / hipFree -- every one of those calls
// changes the device's page tables, and the first operation on the device after such a change was seen to wait 25-30 ms
// when gigabytes are mapped (gat_amd.run() on config 3: the memsets behind gat_problem_create, the read-back behind
// gat_null_stats)
// ---- the knobs (GAT_*): a context's own values, else the process's environment as it was when first asked ------------------
extern char** environ;
static const std::map<std::string, std::string>& env_snapshot() {
  static const std::map<std::string, std::string> snap = [] {
    std::map<std::string, std::string> m;
    for (char** e = environ; e && *e; ++e) {
      if (strncmp(*e, "GAT_", 4) != 0) continue;
      const char* eq = strchr(*e, '=');
      if (eq) m[std::string(*e, (size_t)(eq - *e))] = std::string(eq + 1);
    }
    return m;
  }();
  return snap;
}
const char* gat_opt(const gat_ctx* ctx, const char* key) {
  if (ctx != nullptr) {
    const gat_ctx* own = ctx->options_owner ? ctx->options_owner : ctx;
    std::lock_guard<std::mutex> lock(own->options_mutex);
    auto it = own->options.find(key);
    if (it != own->options.end()) return it->second.empty() ? nullptr : it->second.c_str();   // (values live as long as the entry:
  }                                                                                             //  set_option between calls only)
  const auto& snap = env_snapshot();
  auto it = snap.find(key);
  return it != snap.end() && !it->second.empty() ? it->second.c_str() : nullptr;
}
extern "C" int gat_ctx_set_option(gat_ctx* ctx, const char* key, const char* value) {
  if (!ctx || !key || strncmp(key, "GAT_", 4) != 0) return set_err(ctx, GAT_ERR_ARG, "gat_ctx_set_option: a context and a key GAT_* are needed");
  std::lock_guard<std::mutex> lock(ctx->options_mutex);
  if (value == nullptr) ctx->options.erase(key);      // back to the process's value
  else ctx->options[key] = value;                     // ("" : not set, whatever the environment says)
  return GAT_OK;
}
extern "C" const char* gat_ctx_get_option(const gat_ctx* ctx, const char* key) {
  return key ? gat_opt(ctx, key) : nullptr;
}

static size_t pool_class(size_t bytes) {
  if (bytes >= kPoolMinBytes) return bytes;
  size_t c = 512;
  while (c < bytes) c <<= 1;
  return c;
}

hipError_t dev_pool_alloc(void** out, size_t bytes) {
  int dev = 0;
  (void)hipGetDevice(&dev);
  const size_t want = pool_class(bytes);
  {
    std::lock_guard<std::mutex> lock(g_pool_mutex);
    DevPool& P = g_pools[dev];
    size_t best = P.blocks.size();
    for (size_t i = 0; i < P.blocks.size(); ++i) {                  // the smallest block that fits without wasting a quarter
      const size_t have = P.blocks[i].bytes;
      const bool fits = want < kPoolMinBytes ? have == want : (have >= want && have <= want + want / 4 + kPoolMinBytes);
      if (fits && (best == P.blocks.size() || have < P.blocks[best].bytes)) best = i;
    }
    if (best != P.blocks.size()) {
      *out = P.blocks[best].p;
      P.held -= P.blocks[best].bytes;
      P.out[*out] = P.blocks[best].bytes;
      P.blocks[best] = P.blocks.back();
      P.blocks.pop_back();
      return hipSuccess;
    }
  }
  hipError_t e = hipMalloc(out, want);
  if (e != hipSuccess) {
    // out of memory with blocks held back: give them to the driver and ask again
    std::lock_guard<std::mutex> lock(g_pool_mutex);
    DevPool& P = g_pools[dev];
    if (!P.blocks.empty()) {
      (void)hipGetLastError();
      for (auto& b : P.blocks) (void)hipFree(b.p);
      P.blocks.clear();
      P.held = 0;
      e = hipMalloc(out, want);
    }
  }
  if (e == hipSuccess) {
    std::lock_guard<std::mutex> lock(g_pool_mutex);
    g_pools[dev].out[*out] = want;
  }
  return e;
}

// A block goes back idle: every caller frees behind a synchronisation of the stream it used the block on (a problem's
// destruction, the end of a blocking call), so an idle block may be handed to ANY context of the device -- the contexts of
// a device (the caller's, the observed counts', the annotation build's) run on streams of their own.
void dev_pool_free(void* p, size_t bytes) {
  if (!p) return;
  size_t cls = pool_class(bytes);
  hipPointerAttribute_t attr;
  int dev = 0;
  if (hipPointerGetAttributes(&attr, p) == hipSuccess) dev = attr.device; else { (void)hipGetLastError(); (void)hipGetDevice(&dev); }
  std::vector<void*> evict;
  bool kept = false;
  {
    std::lock_guard<std::mutex> lock(g_pool_mutex);
    DevPool& P = g_pools[dev];
    auto it = P.out.find(p);
    if (it != P.out.end()) { cls = it->second; P.out.erase(it); }     // (the block's own size)
    if (cls <= pool_limit() && P.blocks.size() < 4096) {
      // the block just freed is the one most likely to be asked for again (a host that creates problem after problem):
      // it stays, and what has lain idle longest goes back to the driver until the pool is within its limit -- refusing
      // the NEW block instead made a run over 16 segment tracks allocate and free its 15 GB of scratch per track (78 ms
      // each) once blocks of earlier, larger problems had filled the pool
      P.blocks.push_back(PoolBlock{p, cls, ++P.clock});
      P.held += cls;
      while (P.held > pool_limit() && P.blocks.size() > 1) {
        size_t old = 0;
        for (size_t i = 1; i < P.blocks.size(); ++i) if (P.blocks[i].stamp < P.blocks[old].stamp) old = i;
        if (P.blocks[old].p == p) break;
        evict.push_back(P.blocks[old].p);
        P.held -= P.blocks[old].bytes;
        P.blocks[old] = P.blocks.back();
        P.blocks.pop_back();
      }
      kept = true;
    }
  }
  for (void* q : evict) (void)hipFree(q);            // (outside the lock: hipFree waits for the device)
  if (!kept) (void)hipFree(p);
}

size_t dev_pool_held() {
  int dev = 0;
  (void)hipGetDevice(&dev);
  std::lock_guard<std::mutex> lock(g_pool_mutex);
  return g_pools[dev].held;
}

// ------------------------------------------------------------------------------------------
// the host threads behind parallel_for (gat_host.h): one pool per process, kept between calls.  A job is (fn, arg, n):
// the caller and the invited workers take indices from one atomic counter; every worker acknowledges every job (invited or
// not), the caller returns when all have.  Workers spin for a few dozen microseconds behind a job -- the loops of a problem's
// creation follow each other at that distance -- and then sleep on a condition variable.
namespace {
struct HostPool {
  std::mutex m;
  std::condition_variable cv;
  std::mutex run_mutex;                       // one job at a time; a second caller falls back to threads of its own
  std::vector<std::thread> threads;
  void (*fn)(void*, int64_t) = nullptr;
  void* arg = nullptr;
  int64_t n = 0;
  int invited = 0;                            // workers 0 .. invited-1 take part in the current job
  std::atomic<int64_t> next{0};
  std::atomic<int> pending{0};                // workers that have not acknowledged the current job
  std::atomic<uint64_t> gen{0};
  pid_t pid = 0;
};
HostPool* g_host_pool[2] = {nullptr, nullptr};   // 0: the callers' threads, 1: the asynchronous annotation build's
std::mutex g_host_pool_mutex;
thread_local bool t_in_pool_job = false;
thread_local int t_pool = 0;

inline void cpu_relax() { __builtin_ia32_pause(); }

// `seen` = the generation at the thread's creation (taken by the creator, which holds run_mutex: no job is out): a worker
// born after earlier jobs must only react to jobs published after it exists -- starting from 0 it made a spurious pass over
// the job fields while host_pool_run was writing them and acknowledged a job that had not been published
void host_pool_worker(HostPool* P, int index, uint64_t seen) {
  t_in_pool_job = true;                       // (a parallel_for inside a job's body runs on the calling worker)
  for (;;) {
    uint64_t g;
    int spins = 0;
    while ((g = P->gen.load(std::memory_order_acquire)) == seen) {
      if (++spins < 4000) { cpu_relax(); continue; }
      std::unique_lock<std::mutex> lk(P->m);
      P->cv.wait(lk, [&] { return P->gen.load(std::memory_order_acquire) != seen; });
    }
    seen = g;
    if (index < P->invited) {
      const int64_t n = P->n;
      for (int64_t i = P->next.fetch_add(1, std::memory_order_relaxed); i < n; i = P->next.fetch_add(1, std::memory_order_relaxed))
        P->fn(P->arg, i);
    }
    P->pending.fetch_sub(1, std::memory_order_release);
  }
}

void run_with_own_threads(int64_t n, void (*fn)(void*, int64_t), void* arg, unsigned nthreads) {
  std::atomic<int64_t> next(0);
  auto worker = [&]() { for (int64_t i = next.fetch_add(1); i < n; i = next.fetch_add(1)) fn(arg, i); };
  std::vector<std::thread> pool;
  for (unsigned t = 1; t < nthreads; ++t) pool.emplace_back(worker);
  worker();
  for (auto& th : pool) th.join();
}
}  // namespace

void host_pool_select(int pool) { t_pool = pool == 1 ? 1 : 0; }

void host_pool_run(int64_t n, void (*fn)(void*, int64_t), void* arg) {
  const char* env_t = gat_opt(nullptr, "GAT_HOST_THREADS");
  unsigned nthreads = env_t ? (unsigned)std::max(1, atoi(env_t)) : std::min(16u, std::max(1u, std::thread::hardware_concurrency()));
  nthreads = (unsigned)std::min<int64_t>(nthreads, std::max<int64_t>(1, n));
  if (nthreads <= 1 || t_in_pool_job) { for (int64_t i = 0; i < n; ++i) fn(arg, i); return; }
  const char* env_p = gat_opt(nullptr, "GAT_HOST_POOL");
  if (env_p && atoi(env_p) == 0) { run_with_own_threads(n, fn, arg, nthreads); return; }
  HostPool* P;
  {
    std::lock_guard<std::mutex> lock(g_host_pool_mutex);
    // (a forked child has the parent's pool object but none of its threads: it starts one of its own)
    HostPool*& slot = g_host_pool[t_pool];
    if (slot == nullptr || slot->pid != getpid()) { slot = new HostPool(); slot->pid = getpid(); }
    P = slot;
  }
  if (!P->run_mutex.try_lock()) { run_with_own_threads(n, fn, arg, nthreads); return; }
  while (P->threads.size() + 1 < nthreads) {                       // (grows to the largest team asked for; no job is out)
    const int index = (int)P->threads.size();
    P->threads.emplace_back(host_pool_worker, P, index, P->gen.load(std::memory_order_acquire));
    P->threads.back().detach();
  }
  P->fn = fn; P->arg = arg; P->n = n;
  P->invited = (int)nthreads - 1;
  P->next.store(0, std::memory_order_relaxed);
  P->pending.store((int)P->threads.size(), std::memory_order_relaxed);
  {
    std::lock_guard<std::mutex> lk(P->m);
    P->gen.fetch_add(1, std::memory_order_release);
  }
  P->cv.notify_all();
  t_in_pool_job = true;
  for (int64_t i = P->next.fetch_add(1, std::memory_order_relaxed); i < n; i = P->next.fetch_add(1, std::memory_order_relaxed)) fn(arg, i);
  t_in_pool_job = false;
  for (int spins = 0; P->pending.load(std::memory_order_acquire) != 0; ++spins) {
    if (spins < 20000) cpu_relax(); else std::this_thread::yield();
  }
  P->run_mutex.unlock();
}

int check_list(gat_ctx* ctx, const gat_segment* s, int64_t n, const char* what, int64_t idx) {
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

// staged copies: one thread's memcpy into pinned memory is 5-8 GB/s, a third of what the DMA behind it moves
void parallel_copy(void* dst, const void* src, size_t bytes) {
  constexpr size_t kChunk = (size_t)2 << 20;
  if (bytes < 4 * kChunk) { memcpy(dst, src, bytes); return; }
  parallel_for((int64_t)((bytes + kChunk - 1) / kChunk), [&](int64_t i) {
    const size_t o = (size_t)i * kChunk;
    memcpy((char*)dst + o, (const char*)src + o, std::min(kChunk, bytes - o));
  });
}

// stable LSD radix sort of 64-bit keys by their upper 32 bits (three passes of 11 / 11 / 10 bits); tmp is scratch
static void radix_sort_hi32(std::vector<uint64_t>& v, std::vector<uint64_t>& tmp) {
  const size_t n = v.size();
  if (n < 2) return;
  if (n < 512) { std::stable_sort(v.begin(), v.end(), [](uint64_t a, uint64_t b) { return (a >> 32) < (b >> 32); }); return; }
  tmp.resize(n);
  uint64_t* src = v.data();
  uint64_t* dst = tmp.data();
  uint32_t top = 0;
  for (size_t i = 0; i < n; ++i) top |= (uint32_t)(src[i] >> 32);
  static const int kShift[3] = {32, 43, 54}, kBits[3] = {11, 11, 10};
  for (int pass = 0; pass < 3; ++pass) {
    if (pass > 0 && (top >> (kShift[pass] - 32)) == 0) break;       // no key has a bit up there
    const int sh = kShift[pass];
    const uint32_t mask = (1u << kBits[pass]) - 1u;
    uint32_t cnt[2049];
    memset(cnt, 0, sizeof(cnt));
    for (size_t i = 0; i < n; ++i) cnt[((src[i] >> sh) & mask) + 1]++;
    for (uint32_t k = 0; k < mask + 1u; ++k) cnt[k + 1] += cnt[k];
    for (size_t i = 0; i < n; ++i) dst[cnt[(src[i] >> sh) & mask]++] = src[i];
    std::swap(src, dst);
  }
  if (src != v.data()) memcpy(v.data(), src, n * sizeof(uint64_t));
}

// The merged index of k_count_merged: per group (contig) the intervals of ALL tracks in one list sorted by start, 8-byte
// entries {start, length:16 | track:16}.  Intervals longer than `bound` (a power of two, see below, at most 32 768) are
// cut into pieces -- an overlap sum does not change -- so that no entry keeps a scan alive over more
// than `bound` bases; first[g] is the first entry with end > g << shift or start >= g << shift, i.e. where a scan for a
// segment starting in cell g begins.
static int build_merged(gat_ctx* ctx, AnnoDev& A, const gat_segment* annos, const int64_t* lbeg, const int64_t* lend,
                        int64_t n_tracks, int32_t n_groups, double mean_seg_len) {
  if (n_tracks > 65535) return GAT_OK;                              // (track ids are 16 bits: such problems keep the per-track kernel)
  PrepTimer tm;
  std::vector<int64_t> hz_off((size_t)n_groups + 1, 0), hf_off((size_t)n_groups + 1, 0);
  std::vector<int32_t> h_shift((size_t)n_groups, 0), h_cells((size_t)n_groups, 1);
  // a contig's index is built by itself (collect, sort, grid): the contigs are dealt to host threads -- sorting 10^7 entries
  // on one core made gat_problem_create 0.9 s on the config-4 shape.  An entry is one 64-bit key, start in the upper half,
  // (track << 16 | length) in the lower: collected track by track and sorted by a STABLE radix sort on the start, equal
  // starts stay in track order
  std::vector<std::vector<uint64_t>> ck((size_t)n_groups);
  std::vector<std::vector<uint32_t>> cf((size_t)n_groups);
  std::vector<int> c_err((size_t)n_groups, 0);
  std::vector<double> c_scan((size_t)n_groups, 0.0);                 // entries a scan is expected to pass, x the contig's entries
  const char* env_bf = gat_opt(ctx, "GAT_MERGED_BOUND");
  const uint64_t bfac = env_bf ? (uint64_t)std::max(1, atoi(env_bf)) : 2;
  auto build_one = [&](int c) {
    std::vector<uint64_t>& e = ck[(size_t)c];
    std::vector<uint64_t> tmp;
    std::vector<uint32_t>& hf = cf[(size_t)c];
    uint64_t total_len = 0, cnt = 0, span = 0;
    for (int64_t t = 0; t < n_tracks; ++t) {
      const int64_t l = t * n_groups + c;
      for (int64_t i = lbeg[l]; i < lend[l]; ++i) total_len += annos[i].end - annos[i].start;
      cnt += (uint64_t)(lend[l] - lbeg[l]);
      if (lend[l] > lbeg[l]) span = std::max<uint64_t>(span, annos[lend[l] - 1].end);
    }
    // the piece bound: a scan starts at the first entry that reaches into the segment's cell and passes everything up to
    // the segment's end, so it walks over about (bound + segment length) / spacing entries, most of which ended before the
    // segment began.  Twice the mean interval length or twice the mean spacing of the entries, whichever is larger (cutting
    // finer than the spacing only adds entries): config-4 shape, 1 000 tracks, one entry per 300 bases: 30.8 -> 25.9 ms per
    // 4 096 samples against the earlier 8 x mean length; config 3 (one per 3 000) keeps its bound.
    const uint64_t want = cnt > 0 ? std::max(bfac * (total_len / cnt), 2 * (span / cnt)) : 0;
    uint32_t bound = 256;
    while ((uint64_t)bound < want && bound < 32768u) bound <<= 1;
    e.reserve((size_t)cnt + (size_t)cnt / 4 + 4);
    for (int64_t t = 0; t < n_tracks; ++t) {
      const int64_t l = t * n_groups + c;
      const uint64_t tt = (uint64_t)t << 16;
      for (int64_t i = lbeg[l]; i < lend[l]; ++i) {
        uint32_t s0 = annos[i].start;
        const uint32_t e0 = annos[i].end;
        while (e0 - s0 > bound) { e.push_back(((uint64_t)s0 << 32) | tt | (uint64_t)bound); s0 += bound; }
        e.push_back(((uint64_t)s0 << 32) | tt | (uint64_t)(e0 - s0));
      }
    }
    radix_sort_hi32(e, tmp);
    const size_t ne = e.size();
    // a scan starts up to `bound` bases in front of its segment and ends with it
    if (span > 0) c_scan[(size_t)c] = (double)ne * ((double)bound * 0.5 + mean_seg_len) * (double)ne / (double)span;
    if (ne >= 0xfffffff0ull) { c_err[(size_t)c] = 1; return; }
    auto st_of = [&](size_t i) { return (uint32_t)(e[i] >> 32); };
    auto en_of = [&](size_t i) { return (uint32_t)(e[i] >> 32) + (uint32_t)(e[i] & 0xffffu); };
    const uint32_t max_start = ne ? st_of(ne - 1) : 0u;
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
        while (k < ne && (uint64_t)st_of(k) < cs) ++k;
        hf[(size_t)g] = (uint32_t)k;
      }
      for (size_t i = 0; i < ne; ++i) {                             // ... or an earlier one that reaches past it
        const uint64_t en = en_of(i);
        for (int64_t g = ((int64_t)st_of(i) >> sh) + 1; g < cells && ((uint64_t)g << sh) < en; ++g)
          if ((uint32_t)i < hf[(size_t)g]) hf[(size_t)g] = (uint32_t)i;
      }
    }
    // the device reads an entry as uint2 {start, track << 16 | length}: the two halves of the key the other way round
    for (size_t i = 0; i < ne; ++i) e[i] = (e[i] >> 32) | (e[i] << 32);
    e.push_back(0xffffffffull);                                     // {0xffffffff, 0} ends every scan
    while (e.size() & 7) e.push_back(0xffffffffull);                // (entries are read in blocks of eight: the next contig starts at one)
  };
  parallel_for(n_groups, [&](int64_t c) { build_one((int)c); });
  tm.lap("  merged index: per contig");
  for (int c = 0; c < n_groups; ++c) {
    if (c_err[(size_t)c]) return set_err(ctx, GAT_ERR_CAPACITY, "group %d: more than 2^32 annotation intervals", c);
    hz_off[(size_t)c + 1] = hz_off[(size_t)c] + (int64_t)ck[(size_t)c].size();
    hf_off[(size_t)c + 1] = hf_off[(size_t)c] + (int64_t)cf[(size_t)c].size();
  }
  static_assert(sizeof(uint2) == sizeof(uint64_t), "index entries");
  const size_t n_hz = (size_t)hz_off[(size_t)n_groups] + 8, n_hf = (size_t)hf_off[(size_t)n_groups];   // (+ a block of sentinels behind the last contig)
  HIPCHK(ctx, A.mz.upload_built(n_hz, ctx, [&](uint2* hz) {
    parallel_for(n_groups, [&](int64_t c) {
      if (!ck[(size_t)c].empty()) memcpy(hz + hz_off[(size_t)c], ck[(size_t)c].data(), ck[(size_t)c].size() * 8);
    });
    for (size_t i = n_hz - 8; i < n_hz; ++i) hz[i] = make_uint2(0xffffffffu, 0u);
  }));
  HIPCHK(ctx, A.mfirst.upload_built(n_hf, ctx, [&](uint32_t* hf) {
    parallel_for(n_groups, [&](int64_t c) {
      if (!cf[(size_t)c].empty()) memcpy(hf + hf_off[(size_t)c], cf[(size_t)c].data(), cf[(size_t)c].size() * 4);
    });
  }));
  HIPCHK(ctx, A.mz_off.upload(hz_off, ctx));
  HIPCHK(ctx, A.mf_off.upload(hf_off, ctx));
  HIPCHK(ctx, A.m_shift.upload(h_shift, ctx));
  HIPCHK(ctx, A.m_cells.upload(h_cells, ctx));
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
    HIPCHK(ctx, A.m_slot_off.upload(so, ctx));
    HIPCHK(ctx, A.m_slot_contigs.upload(sc, ctx));
  }
  A.has_merged = true;
  A.merged_entries = (int64_t)n_hz;
  {
    // entries per step of a scan (k_count_merged<.., BLK>): blocks of eight where a scan passes six or more on average
    double num = 0, den = 0;
    for (int c = 0; c < n_groups; ++c) { num += c_scan[(size_t)c]; den += (double)(hz_off[(size_t)c + 1] - hz_off[(size_t)c]); }
    const char* env_b = gat_opt(ctx, "GAT_MERGED_BLOCK");
    // short scans: the first two entries ride in the grid cell's own record (32 bytes per cell: where the cells are few
    // enough -- the records of the config-4 shape would be 160 MB, but its scans are long and take the blocks anyway)
    const int by_length = den > 0 && num / den >= 6.0 ? 8 : (n_hf <= ((size_t)64 << 20) / 32 ? 1 : 2);
    A.merged_block = env_b ? (atoi(env_b) == 8 ? 8 : atoi(env_b) == 1 ? 1 : 2) : by_length;
  }
  if (A.merged_block == 1) {
    HIPCHK(ctx, A.mcell.upload_built(n_hf * 2, ctx, [&](uint4* hc) {
      parallel_for(n_groups, [&](int64_t c) {
        const uint2* z = reinterpret_cast<const uint2*>(ck[(size_t)c].data());
        const size_t nz = ck[(size_t)c].size();                               // (entries + the sentinels behind them)
        const uint32_t* f = cf[(size_t)c].data();
        uint4* out = hc + 2 * hf_off[(size_t)c];
        for (size_t g = 0; g < cf[(size_t)c].size(); ++g) {
          const uint32_t first = f[g];
          const uint2 e0 = first < nz ? z[first] : make_uint2(0xffffffffu, 0u);
          const uint2 e1 = (size_t)first + 1 < nz ? z[first + 1] : make_uint2(0xffffffffu, 0u);
          out[2 * g] = make_uint4(first, 0u, e0.x, e0.y);
          out[2 * g + 1] = make_uint4(e1.x, e1.y, 0u, 0u);
        }
      });
    }));
  }
  tm.lap("  merged index: gather + upload");
  return GAT_OK;
}

// The annotation tables of the count kernels.  n_lists = n_tracks * n_groups lists, list l = annos[lbeg[l] .. lend[l])
// (a CSR array passes off and off + 1); checked: whether the lists have to be verified normalized (the ones the library
// merged itself are).  want_merged: build the merged index when the problem's shape asks for it (see below).
int build_annos(gat_ctx* ctx, AnnoDev& A, const gat_segment* annos, const int64_t* lbeg, const int64_t* lend, int64_t n_lists,
                int32_t n_groups, bool want_merged, bool checked, double mean_seg_len, bool nucleotide_only) {
  PrepTimer tm;
  A.h_off.assign((size_t)n_lists + 1, 0);
  A.max_m = 0;
  for (int64_t l = 0; l < n_lists; ++l) {
    const int64_t m = lend[l] - lbeg[l];
    if (m < 0) return set_err(ctx, GAT_ERR_ARG, "annotation list %lld: negative length", (long long)l);
    A.h_off[(size_t)l + 1] = A.h_off[(size_t)l] + m;
    A.max_m = std::max(A.max_m, m);
  }
  const int64_t total = A.h_off[(size_t)n_lists];
  A.total = total;
  if (!checked) {
    constexpr int64_t kBlock = 256;                                 // lists per task
    const int64_t nblocks = (n_lists + kBlock - 1) / kBlock;
    std::vector<int64_t> bad((size_t)std::max<int64_t>(1, nblocks), -1);
    parallel_for(nblocks, [&](int64_t b) {
      for (int64_t l = b * kBlock; l < std::min(n_lists, (b + 1) * kBlock); ++l) {
        const gat_segment* s = annos + lbeg[l];
        const int64_t m = lend[l] - lbeg[l];
        bool ok = true;
        for (int64_t i = 0; i < m && ok; ++i)
          ok = s[i].start < s[i].end && s[i].end < 0x80000000u && (i == 0 || s[i - 1].end <= s[i].start);
        if (!ok) { bad[(size_t)b] = l; return; }
      }
    });
    for (int64_t b = 0; b < nblocks; ++b)
      if (bad[(size_t)b] >= 0) {                                     // the first offender, with the reference's message
        const int64_t l = bad[(size_t)b];
        return check_list(ctx, annos + lbeg[l], lend[l] - lbeg[l], "annotation", l);
      }
  }
  // the merged index: from four tracks up -- and for fewer when their lists do not fit the LDS tile of k_count_seg (it
  // would read them from global memory with four look-ups per sample segment; the index needs two: config-5 shape,
  // one track of a million intervals, count 2.86 -> 1.74 ms per 16 384 samples)
  const int64_t n_tracks = n_groups > 0 ? n_lists / n_groups : 0;
  const char* env_mm = gat_opt(ctx, "GAT_MERGED_MIN_TRACKS");
  const char* env_e = gat_opt(ctx, "GAT_COUNT_LDS_ENTRIES");
  (void)env_e;
  const bool unstaged = !count_lists_staged(ctx, A.max_m) && A.max_m > 0 && !env_mm;
  const bool do_merged = want_merged && n_groups > 0 && (n_tracks >= (env_mm ? atoi(env_mm) : 4) || unstaged);
  // GAT_ANNOTATIONS_NUCLEOTIDE_ONLY: the per-track tables are k_count_seg's / k_count_anno's; a caller that will only ask for the
  // nucleotide counters never reaches them once the merged index exists (config 3: 2.0 of the build's 5.3 ms)
  // (... and the count launch would take it: count_route's conditions, gat_mi355.hip)
  A.per_track = !(nucleotide_only && do_merged && n_tracks * 4 * kMergedWavesHost + 1024 <= (int64_t)ctx->max_lds &&
                  !gat_opt(ctx, "GAT_COUNT_NO_MERGED"));
  if (A.per_track) {
  {
    // starts / ends / running lengths: one pass over the lists, written straight into the pinned staging buffer (three
    // arrays side by side) when they fit one piece of it, and sent from there
    const size_t tn = (size_t)std::max<int64_t>(total, 1);
    const bool in_stage = 3 * tn * 4 <= kStagePiece;
    std::unique_ptr<uint32_t[]> heap;
    uint32_t* hs;
    if (in_stage) { HIPCHK(ctx, ctx_stage(ctx, 3 * tn * 4)); hs = reinterpret_cast<uint32_t*>(ctx->h_stage); }
    else { heap.reset(new uint32_t[3 * tn]); hs = heap.get(); }
    uint32_t* he = hs + tn;
    uint32_t* hc = he + tn;
    constexpr int64_t kBlock = 256;                                 // lists per task
    parallel_for((n_lists + kBlock - 1) / kBlock, [&](int64_t b) {
      for (int64_t l = b * kBlock; l < std::min(n_lists, (b + 1) * kBlock); ++l) {
        const gat_segment* s = annos + lbeg[l];
        const int64_t o = A.h_off[(size_t)l], m = lend[l] - lbeg[l];
        uint32_t cum = 0;
        for (int64_t i = 0; i < m; ++i) {
          hs[(size_t)(o + i)] = s[i].start;
          he[(size_t)(o + i)] = s[i].end;
          hc[(size_t)(o + i)] = cum;
          cum += s[i].end - s[i].start;
        }
      }
    });
    HIPCHK(ctx, A.start.alloc((size_t)total));
    HIPCHK(ctx, A.end.alloc((size_t)total));
    HIPCHK(ctx, A.cumx.alloc((size_t)total));
    if (total > 0) {
      if (in_stage) {
        HIPCHK(ctx, hipMemcpyAsync(A.start.p, hs, (size_t)total * 4, hipMemcpyHostToDevice, ctx->stream));
        HIPCHK(ctx, hipMemcpyAsync(A.end.p, he, (size_t)total * 4, hipMemcpyHostToDevice, ctx->stream));
        HIPCHK(ctx, hipMemcpyAsync(A.cumx.p, hc, (size_t)total * 4, hipMemcpyHostToDevice, ctx->stream));
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
      } else {
        HIPCHK(ctx, staged_h2d(ctx, A.start.p, hs, (size_t)total * 4));
        HIPCHK(ctx, staged_h2d(ctx, A.end.p, he, (size_t)total * 4));
        HIPCHK(ctx, staged_h2d(ctx, A.cumx.p, hc, (size_t)total * 4));
      }
    }
  }
  // per group (contig) a uniform grid over the start coordinates, about one start per cell:
  // grid[g] = #starts < (g << shift); the count kernels look a position up instead of bisecting
  std::vector<int32_t> h_shift((size_t)std::max(1, n_groups), 0), h_cells((size_t)std::max(1, n_groups), 1);
  std::vector<int64_t> h_goff((size_t)n_lists + 1, 0);
  A.max_cells = 1;
  const char* env_g = gat_opt(ctx, "GAT_GRID_FACTOR");
  const int gfac = env_g ? atoi(env_g) : 2;                        // about one start per two cells (measured best of 1, 2, 4, 8)
  for (int c = 0; c < n_groups; ++c) {
    uint32_t max_start = 0;
    int64_t mc = 0;
    for (int64_t t = 0; t < n_tracks; ++t) {
      const int64_t l = t * n_groups + c, m = lend[l] - lbeg[l];
      mc = std::max(mc, m);
      if (m > 0) max_start = std::max(max_start, annos[lend[l] - 1].start);
    }
    int64_t target = 16;
    while (target < mc) target <<= 1;
    target *= gfac;
    int sh = 0;
    while (((int64_t)max_start >> sh) + 1 > target) ++sh;
    h_shift[(size_t)c] = sh;
    h_cells[(size_t)c] = (int32_t)(((int64_t)max_start >> sh) + 1);
    A.max_cells = std::max<int64_t>(A.max_cells, h_cells[(size_t)c]);
  }
  for (int64_t l = 0; l < n_lists; ++l) h_goff[(size_t)l + 1] = h_goff[(size_t)l] + h_cells[(size_t)(l % n_groups)] + 1;
  HIPCHK(ctx, A.grid.upload_built((size_t)h_goff[(size_t)n_lists], ctx, [&](uint32_t* hg) {
    constexpr int64_t kBlock = 256;
    parallel_for((n_lists + kBlock - 1) / kBlock, [&](int64_t b) {
      for (int64_t l = b * kBlock; l < std::min(n_lists, (b + 1) * kBlock); ++l) {
        const int c = (int)(l % n_groups);
        const gat_segment* s = annos + lbeg[l];
        const int64_t m = lend[l] - lbeg[l];
        const int sh = h_shift[(size_t)c], cells = h_cells[(size_t)c];
        uint32_t* g = hg + h_goff[(size_t)l];
        int64_t k = 0;
        for (int cell = 0; cell <= cells; ++cell) {
          const uint64_t bound = (uint64_t)cell << sh;
          while (k < m && (uint64_t)s[k].start < bound) ++k;
          g[cell] = (uint32_t)(cell == cells ? m : k);
        }
      }
    });
  }));
  tm.lap("  annotation tables: SoA + grids, sent");
  HIPCHK(ctx, A.goff.upload(h_goff, ctx));
  HIPCHK(ctx, A.shift.upload(h_shift, ctx));
  HIPCHK(ctx, A.cells.upload(h_cells, ctx));
  }
  HIPCHK(ctx, A.off.upload(A.h_off, ctx));
  tm.lap("  annotation tables: offsets sent");
  if (do_merged) {
    int rc = build_merged(ctx, A, annos, lbeg, lend, n_tracks, n_groups, mean_seg_len);
    if (rc) return rc;
  }
  return GAT_OK;
}

// The contig-level annotation lists from lists that carry a group id (gat_problem_desc::anno_group): what
// computeSample's contig_annotations are (gat/__init__.py:716-718: annotations ... fromIsochores(), i.e.
// IntervalDictionary.fromIsochores, gat/Engine.pyx:2857-2876): the lists of a (track, contig) concatenated and -- when
// the keys carry isochores -- sorted and merge(0)d (gat/SegmentList.pyx:756-816: empty segments dropped, a segment starting
// at or before the running end joins it); without isochores a key IS its contig and the list passes through (a later
// list of the same group replaces an earlier one, as the dictionary assignment does).  Groups are built by host threads
// into buf; list g = buf[gbeg[g] .. gend[g]).
static int group_annotations(gat_ctx* ctx, const gat_problem_desc* d, std::vector<gat_segment>& buf, std::vector<int64_t>& gbeg,
                             std::vector<int64_t>& gend) {
  const int64_t n_groups = (int64_t)d->n_tracks * d->n_contigs, nl = d->n_anno_lists;
  std::vector<int64_t> cnt((size_t)n_groups + 1, 0), size((size_t)n_groups + 1, 0);
  for (int64_t l = 0; l < nl; ++l) {
    const int32_t g = d->anno_group[l];
    if (g < -1 || g >= n_groups) return set_err(ctx, GAT_ERR_ARG, "annotation list %lld: group %d out of range", (long long)l, g);
    const int64_t b = d->anno_off[l], e = d->anno_end ? d->anno_end[l] : d->anno_off[l + 1];
    if (e < b) return set_err(ctx, GAT_ERR_ARG, "annotation list %lld: negative length", (long long)l);
    if (g < 0) continue;
    cnt[(size_t)g + 1]++;
    size[(size_t)g + 1] += d->merge_contigs ? e - b : std::max<int64_t>(size[(size_t)g + 1], e - b) - size[(size_t)g + 1];
  }
  for (int64_t g = 0; g < n_groups; ++g) { cnt[(size_t)g + 1] += cnt[(size_t)g]; size[(size_t)g + 1] += size[(size_t)g]; }
  std::vector<int64_t> member((size_t)cnt[(size_t)n_groups]), cur(cnt.begin(), cnt.end() - 1);
  for (int64_t l = 0; l < nl; ++l) if (d->anno_group[l] >= 0) member[(size_t)cur[(size_t)d->anno_group[l]]++] = l;
  buf.resize((size_t)size[(size_t)n_groups]);
  gbeg.assign(size.begin(), size.end() - 1);
  gend = gbeg;
  constexpr int64_t kBlock = 16;                                    // groups per task
  const int64_t nblocks = (n_groups + kBlock - 1) / kBlock;
  std::vector<int64_t> bad((size_t)std::max<int64_t>(1, nblocks), -1);
  parallel_for(nblocks, [&](int64_t blk) {
    std::vector<uint64_t> keys, tmp;
    std::vector<size_t> runs, nruns;
    for (int64_t g = blk * kBlock; g < std::min(n_groups, (blk + 1) * kBlock); ++g) {
      gat_segment* out = buf.data() + gbeg[(size_t)g];
      const int64_t m0 = cnt[(size_t)g], m1 = cnt[(size_t)g + 1];
      if (m1 == m0) continue;
      auto lb = [&](int64_t l) { return d->anno_off[l]; };
      auto le = [&](int64_t l) { return d->anno_end ? d->anno_end[l] : d->anno_off[l + 1]; };
      if (!d->merge_contigs) {
        const int64_t l = member[(size_t)(m1 - 1)];                  // new[isochore] = segmentlist: the last one stays
        const int64_t n = le(l) - lb(l);
        if (n > 0) memcpy(out, d->annos + lb(l), (size_t)n * sizeof(gat_segment));
        gend[(size_t)g] = gbeg[(size_t)g] + n;
        continue;
      }
      // concatenate, sort by start (one member, or members that follow one another, are sorted already), merge(0).  The
      // members are sorted lists themselves (the isochore pieces of a contig's track): runs that are merged pairwise -- three
      // passes over a contig's 400 intervals for eight isochore classes -- instead of sorted from scratch (1.7 of the 3.2 ms
      // this step took on config 3); a member that is not sorted sends the group to std::sort
      keys.clear();
      runs.clear();
      bool sorted = true, runs_sorted = true;
      uint32_t last = 0;
      for (int64_t q = m0; q < m1; ++q) {
        const int64_t l = member[(size_t)q];
        runs.push_back(keys.size());
        uint32_t last_in_run = 0;
        for (int64_t i = lb(l); i < le(l); ++i) {
          const gat_segment x = d->annos[i];
          if (x.start > x.end || x.end >= 0x80000000u) { bad[(size_t)blk] = l; return; }
          if (x.start == x.end) continue;                            // merge() drops empty segments
          sorted = sorted && x.start >= last;
          runs_sorted = runs_sorted && x.start >= last_in_run;
          last = x.start;
          last_in_run = x.start;
          keys.push_back(((uint64_t)x.start << 32) | x.end);
        }
      }
      runs.push_back(keys.size());
      if (!sorted) {
        if (!runs_sorted) std::sort(keys.begin(), keys.end());
        else {
          tmp.resize(keys.size());
          while (runs.size() > 2) {                                  // pairwise merges of neighbouring runs
            nruns.clear();
            for (size_t r = 0; r + 1 < runs.size(); r += 2) {
              const size_t a = runs[r], b = runs[r + 1], c = r + 2 < runs.size() ? runs[r + 2] : runs[r + 1];
              std::merge(keys.begin() + (long)a, keys.begin() + (long)b, keys.begin() + (long)b, keys.begin() + (long)c, tmp.begin() + (long)a);
              nruns.push_back(a);
            }
            nruns.push_back(keys.size());
            keys.swap(tmp);
            runs.swap(nruns);
          }
        }
      }
      int64_t n = 0;
      for (size_t i = 0; i < keys.size(); ++i) {
        const uint32_t s0 = (uint32_t)(keys[i] >> 32), e0 = (uint32_t)keys[i];
        if (n > 0 && s0 <= out[n - 1].end) { if (e0 > out[n - 1].end) out[n - 1].end = e0; }
        else { out[n].start = s0; out[n].end = e0; ++n; }
      }
      gend[(size_t)g] = gbeg[(size_t)g] + n;
    }
  });
  for (int64_t b = 0; b < nblocks; ++b)
    if (bad[(size_t)b] >= 0)
      return set_err(ctx, GAT_ERR_ARG, "annotation list %lld: segment with start > end or a coordinate >= 2^31", (long long)bad[(size_t)b]);
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

static int32_t cap_for(const gat_ctx* ctx, int64_t n) {
  int64_t c = n + n / 4 + 96;
  if (gat_opt(ctx, "GAT_TEST_SMALL_CAPS")) c = n / 2 + 8;      // tests: force the overflow / retry path
  c = (c + 63) / 64 * 64;
  return (int32_t)c;
}

int layout_slab(gat_problem* P) {
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
      if (gat_opt(P->ctx, "GAT_TEST_SMALL_CAPS")) need = std::max<int64_t>(64, need / 4 / 64 * 64);      // tests: force the repeat
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

int upload_layout(gat_ctx* ctx, gat_problem* P) {
  HIPCHK(ctx, P->d_units.upload(P->h_units, ctx));
  {
    // a wave finds its unit with one load (units_o[blockIdx.y]) instead of order[] -> units[]
    std::vector<UnitDev> o;
    o.reserve(P->h_order.size());
    for (int32_t u : P->h_order) { UnitDev x = P->h_units[(size_t)u]; x.pad = u; o.push_back(x); }
    if (o.empty()) o.push_back(UnitDev{});
    HIPCHK(ctx, P->d_units_o.upload(o, ctx));
  }
  {
    // size classes of the launch order (largest unit first, capacities do not grow): a new class where the capacity has
    // dropped to 70 % of the class's first, at most six.  The wave-per-unit kernels keep a unit's list in LDS, their
    // speed follows the waves a CU holds, and the dynamic LDS of a launch is one number: sized for the longest list of
    // the PROBLEM, the config-4 shape ran ONE wave per CU (81 KB for chr1's 8 000 segments, 13 KB for chr21's).
    P->h_class_start.clear();
    const int N = (int)P->h_order.size();
    int32_t first_cap = 0;
    const char* env_c = gat_opt(ctx, "GAT_SIZE_CLASSES");
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
  HIPCHK(ctx, P->d_contig_slab_off.upload(P->h_contig_slab_off, ctx));
  {
    // k_contig's per-unit record: the unit, where its list stands in a sample's slab under THIS layout, its launch position
    std::vector<int4> rec;
    rec.reserve(P->h_contig_units.size() + 1);
    for (int32_t u : P->h_contig_units)
      rec.push_back(make_int4(u, P->h_units[(size_t)u].slab_off, (size_t)u < P->h_unit_pos.size() ? P->h_unit_pos[(size_t)u] : -1, 0));
    if (rec.empty()) rec.push_back(make_int4(0, 0, -1, 0));
    HIPCHK(ctx, P->d_cu_rec.upload(rec, ctx));
  }
  {
    std::vector<int32_t> o = P->h_contig_order;
    if (o.empty()) o.push_back(0);
    HIPCHK(ctx, P->d_contig_order.upload(o, ctx));
  }
  HIPCHK(ctx, P->d_count_c_off.upload(P->h_count_c_off, ctx));
  HIPCHK(ctx, P->d_count_n_index.upload(P->h_count_n_index, ctx));
  P->batch = 0;   // scratch must be re-sized
  return GAT_OK;
}

// ------------------------------------------------------------------------------------------
// the annotation object
// the build itself: the contig-level lists, then the tables (on `ctx`'s stream and through its staging buffer)
static int annotations_build(gat_ctx* ctx, gat_annotations* A, const gat_annotations_desc* d) {
  PrepTimer tm;
  struct FlushOnExit { gat_ctx* c; ~FlushOnExit() { (void)stage_flush(c); } } flush_on_exit{ctx};
  int rc;
  if (d->anno_group != nullptr) {
    // lists with a group id each: the library forms the contig-level lists itself (fromIsochores)
    gat_problem_desc pd;
    memset(&pd, 0, sizeof(pd));
    pd.n_tracks = d->n_tracks; pd.n_contigs = d->n_contigs; pd.merge_contigs = d->merge_contigs;
    pd.annos = d->annos; pd.anno_off = d->anno_off; pd.n_anno_lists = d->n_anno_lists; pd.anno_end = d->anno_end; pd.anno_group = d->anno_group;
    std::vector<gat_segment> buf;
    std::vector<int64_t> gbeg, gend;
    if ((rc = group_annotations(ctx, &pd, buf, gbeg, gend))) return rc;
    tm.lap("annotations grouped by contig");
    rc = build_annos(ctx, A->dev, buf.data(), gbeg.data(), gend.data(), (int64_t)d->n_tracks * d->n_contigs, d->n_contigs, true,
                     d->merge_contigs != 0, d->mean_segment_length, (d->flags & GAT_ANNOTATIONS_NUCLEOTIDE_ONLY) != 0);
  } else {
    if ((int64_t)d->n_tracks * d->n_contigs > 0 && (!d->anno_off || (!d->annos && d->anno_off[(int64_t)d->n_tracks * d->n_contigs] > 0)))
      return set_err(ctx, GAT_ERR_ARG, "gat_annotations_create: NULL lists");
    static const int64_t kZero[2] = {0, 0};
    const int64_t* off = d->anno_off ? d->anno_off : kZero;
    rc = build_annos(ctx, A->dev, d->annos, off, off + 1, (int64_t)d->n_tracks * d->n_contigs, d->n_contigs, true, false,
                     d->mean_segment_length, (d->flags & GAT_ANNOTATIONS_NUCLEOTIDE_ONLY) != 0);
  }
  if (rc) return rc;
  HIPCHK(ctx, stage_flush(ctx));
  HIPCHK(ctx, hipStreamSynchronize(ctx->stream));    // (the tables are resident when the build is reported done)
  if (A->shape_known && (A->dev.has_merged != A->will_merge || (A->total_known >= 0 && A->dev.total != A->total_known)))
    return set_err(ctx, GAT_ERR_DEVICE, "internal: the annotation tables are not of the shape announced before their build "
                   "(merged index %d / %d, %lld / %lld intervals)", (int)A->dev.has_merged, (int)A->will_merge,
                   (long long)A->dev.total, (long long)A->total_known);
  return GAT_OK;
}

static void annotations_build_worker(gat_annotations* A, gat_annotations_desc d) {
  host_pool_select(1);                               // (the callers' pool stays theirs: they prepare observed counts meanwhile)
  gat_ctx* b = A->ctx->build_ctx;
  int rc = GAT_OK;
  if (hipSetDevice(b->device) != hipSuccess) rc = set_err(b, GAT_ERR_DEVICE, "hipSetDevice failed in the annotation build");
  if (rc == GAT_OK) {
    try { rc = annotations_build(b, A, &d); }
    catch (const std::exception& e) { rc = set_err(b, GAT_ERR_MEMORY, "annotation build: %s", e.what()); }   // (bad_alloc: reported, not fatal)
  }
  A->build_rc = rc;
  if (rc) A->build_err = b->err;
  A->ready.store(1, std::memory_order_release);
}

int annotations_wait(gat_ctx* ctx, gat_annotations* a) {
  if (a->worker.joinable()) {
    a->worker.join();
    if (a->ctx && a->ctx->building == a) a->ctx->building = nullptr;
  }
  if (a->build_rc) return set_err(ctx, a->build_rc, "%s", a->build_err.c_str());
  return GAT_OK;
}

extern "C" int gat_annotations_wait(gat_ctx* ctx, gat_annotations* a) {
  if (!a) return set_err(ctx, GAT_ERR_ARG, "gat_annotations_wait: NULL argument");
  return annotations_wait(ctx, a);
}

extern "C" int gat_annotations_create(gat_ctx* ctx, const gat_annotations_desc* d, gat_annotations** out) {
  if (!ctx || !d || !out) return set_err(ctx, GAT_ERR_ARG, "gat_annotations_create: NULL argument");
  *out = nullptr;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  if (d->n_tracks < 0 || d->n_contigs < 0) return set_err(ctx, GAT_ERR_ARG, "gat_annotations_create: negative size");
  std::unique_ptr<gat_annotations> A(new gat_annotations());
  A->ctx = ctx;
  A->n_tracks = d->n_tracks;
  A->n_groups = d->n_contigs;
  A->merge_groups = d->merge_contigs ? 1 : 0;
  // whether the merged index will exist is known from the shape alone when there are enough tracks (build_annos): only then
  // may a problem sample before the tables are there (the sampler's last steps depend on the count kernel that follows)
  const char* env_mm = gat_opt(ctx, "GAT_MERGED_MIN_TRACKS");
  const int64_t n_groups_all = (int64_t)d->n_tracks * d->n_contigs;
  A->will_merge = d->n_contigs > 0 && d->n_tracks >= (env_mm ? atoi(env_mm) : 4) && d->n_tracks <= 65535;
  A->shape_known = A->will_merge;
  if (!d->merge_contigs && d->anno_off != nullptr && n_groups_all > 0) {
    // the lists pass through as they are (a key is its contig; of several lists of a group the last one stays): their
    // lengths are in the desc, and with them everything build_annos decides the tables' form by
    std::vector<int64_t> len((size_t)n_groups_all, 0);
    bool ok = true;
    if (d->anno_group != nullptr) {
      for (int64_t l = 0; l < d->n_anno_lists && ok; ++l) {
        const int32_t g = d->anno_group[l];
        if (g < -1 || g >= n_groups_all) { ok = false; break; }
        if (g >= 0) len[(size_t)g] = (d->anno_end ? d->anno_end[l] : d->anno_off[l + 1]) - d->anno_off[l];
      }
    } else
      for (int64_t g = 0; g < n_groups_all; ++g) len[(size_t)g] = d->anno_off[g + 1] - d->anno_off[g];
    if (ok) {
      int64_t total = 0, max_m = 0;
      for (int64_t g = 0; g < n_groups_all; ++g) { if (len[(size_t)g] < 0) ok = false; total += len[(size_t)g]; max_m = std::max(max_m, len[(size_t)g]); }
      if (ok) {
        const char* env_e = gat_opt(ctx, "GAT_COUNT_LDS_ENTRIES");
        (void)env_e;
        const bool unstaged = !count_lists_staged(ctx, max_m) && max_m > 0 && !env_mm;
        A->will_merge = d->n_contigs > 0 && d->n_tracks <= 65535 && (d->n_tracks >= (env_mm ? atoi(env_mm) : 4) || unstaged);
        A->total_known = total;
        A->shape_known = true;
      }
    }
  }
  const char* env_a = gat_opt(ctx, "GAT_ANNOTATIONS_SYNC");
  const bool async = (d->flags & GAT_ANNOTATIONS_ASYNC) != 0 && A->shape_known && !(env_a && atoi(env_a) != 0);
  if (!async) {
    int rc = annotations_build(ctx, A.get(), d);
    if (rc) return rc;
  } else {
    if (ctx->building != nullptr) (void)annotations_wait(ctx, ctx->building);     // (one build at a time has the build context)
    if (ctx->build_ctx == nullptr) {
      int rc = gat_ctx_create(&ctx->build_ctx, ctx->device, nullptr);
      if (rc == GAT_OK) ctx->build_ctx->options_owner = ctx;          // (the knobs are its owner's)
      if (rc) return rc;
    }
    ctx->build_ctx->err.clear();
    A->ready.store(0, std::memory_order_release);
    ctx->building = A.get();
    A->worker = std::thread(annotations_build_worker, A.get(), *d);
  }
  ctx->refs += 1;
  *out = A.release();
  return GAT_OK;
}

void annotations_release(gat_annotations* a) {
  if (--a->refs > 0) return;
  gat_ctx* ctx = a->ctx;
  (void)annotations_wait(nullptr, a);                // (a build still running has its arrays in use)
  if (ctx) {
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);         // its blocks go back to the pool: nothing may still be reading them
    ctx->stage_used = 0;
  }
  delete a;
  if (ctx) ctx_release(ctx);
}

extern "C" void gat_annotations_destroy(gat_annotations* a) {
  if (!a || a->closed) return;
  a->closed = true;
  annotations_release(a);
}

extern "C" int gat_problem_create(gat_ctx* ctx, const gat_problem_desc* d, gat_problem** out) {
  if (!ctx || !d || !out) return set_err(ctx, GAT_ERR_ARG, "gat_problem_create: NULL argument");
  *out = nullptr;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  if (d->n_units < 0 || d->n_contigs < 0 || d->n_tracks < 0 || d->nbuckets <= 0)
    return set_err(ctx, GAT_ERR_ARG, "gat_problem_create: negative size / nbuckets <= 0");
  PrepTimer tm;
  std::unique_ptr<gat_problem> P(new gat_problem());
  // (declared behind P, so it runs first on every return: the small tables' copies in flight -- stage_push_h2d -- have
  //  landed before a half-built problem's buffers go back to the pool)
  struct FlushOnExit { gat_ctx* c; ~FlushOnExit() { (void)stage_flush(c); } } flush_on_exit{ctx};
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

  // Per unit, on the host threads (192 isochore units x ~50 segments, or 24 contigs x hundreds: the overlaps with the
  // workspace, the rank table -- a sort of the unit's lengths --, the search trees): everything a unit needs by itself into a
  // record of its own; the offsets into the shared tables are dealt out in unit order behind it.
  struct UnitPrep {
    int rc = 0;
    std::string err;
    bool active = false;
    std::vector<uint32_t> rank;          // rank 0 (never drawn) + the bucket indices in ascending order
    std::vector<uint2> ws;
    std::vector<uint32_t> cdf, tree_start, tree_cdf;
    std::vector<uint32_t> pgrid, cgrid;  // the grids of a fragmented workspace, header included (UnitDev::pgrid_off / cgrid_off)
    int64_t nwork = 0;
    double cv2 = 0.0;
  };
  std::vector<UnitPrep> prep((size_t)std::max(0, d->n_units));
  for (int u = 0; u < d->n_units; ++u) {
    UnitDev& U = P->h_units[u];
    memset(&U, 0, sizeof(U));
    const int64_t nus = d->seg_off[u + 1] - d->seg_off[u], nuw = d->ws_off[u + 1] - d->ws_off[u];
    const int c = d->unit_contig[u];
    U.contig = c;
    P->n_seg_total += nus;
    const bool skipped = (nus == 0 || nuw == 0);     // gat/__init__.py:536-538
    if (c >= d->n_contigs || (c < 0 && !skipped))
      return set_err(ctx, GAT_ERR_ARG, "unit %d: contig index %d invalid (skipped units carry -1)", u, c);
    if (skipped) continue;
    per_contig[(size_t)c].push_back(u);
  }
  auto fail_unit = [&](UnitPrep& R, int code, const char* fmt, auto... args) {
    char buf[512];
    snprintf(buf, sizeof(buf), fmt, args...);
    R.rc = code; R.err = buf;
  };
  parallel_for((d->n_units + 7) / 8, [&](int64_t blk) {
  for (int u = (int)blk * 8; u < std::min<int>(d->n_units, (int)blk * 8 + 8); ++u) {
    UnitDev& U = P->h_units[u];
    UnitPrep& R = prep[(size_t)u];
    const gat_segment* us = d->segs + d->seg_off[u];
    const int64_t nus = d->seg_off[u + 1] - d->seg_off[u];
    const gat_segment* uw = d->ws + d->ws_off[u];
    const int64_t nuw = d->ws_off[u + 1] - d->ws_off[u];
    if (nus == 0 || nuw == 0) continue;
    for (int pass = 0; pass < 2 && R.rc == 0; ++pass) {              // gat/Engine.pyx:535-536: both lists normalized
      const gat_segment* l = pass ? uw : us;
      const int64_t nl = pass ? nuw : nus;
      for (int64_t i = 0; i < nl; ++i)
        if (l[i].start >= l[i].end || l[i].end >= 0x80000000u || (i > 0 && l[i - 1].end > l[i].start)) { R.rc = GAT_ERR_ASSERT; R.err = pass ? "workspace" : "segment"; break; }
    }
    if (R.rc) continue;                                              // (the message comes from check_list below, in unit order)
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
    // the histogram over the buckets, cumulated, read as "rank r -> bucket": the bucket indices in ascending order (a sort
    // of the unit's lengths; a std::map insertion per segment was most of this stage: 0.8 of config 2's 1.2 ms)
    R.rank.reserve(lens.size() + 1);
    R.rank.push_back(0u);                           // rank 0 is never drawn (r >= 1, gat/Engine.pyx:419-422)
    for (uint32_t l : lens) {
      const int64_t i = ((int64_t)l + bucket - 1) / bucket;
      if (i >= d->nbuckets) {
        fail_unit(R, GAT_ERR_VALUE, "unit %d: segment of length %u too large: increase nbuckets (%d) or bucket_size (%lld)",
                  u, l, d->nbuckets, (long long)bucket);
        break;
      }
      R.rank.push_back((uint32_t)i);
    }
    if (R.rc) continue;
    std::sort(R.rank.begin() + 1, R.rank.end());                    // ranks (cum-count, cum] of a bucket hold it
    U.hist_total = (uint32_t)lens.size();
    U.bucket = (uint32_t)bucket;
    {
      double m1 = 0, m2 = 0;                               // squared coefficient of variation of the lengths drawn
      for (uint32_t l : lens) { m1 += (double)l; m2 += (double)l * (double)l; }
      m1 /= (double)lens.size(); m2 /= (double)lens.size();
      R.cv2 = m1 > 0 ? std::max(0.0, m2 / (m1 * m1) - 1.0) : 0.0;
    }
    // SegmentListSampler(workspace) (gat/Engine.pyx:261-277)
    U.n_ws = (int32_t)nuw;
    uint32_t tot = 0;
    R.ws.reserve((size_t)nuw); R.cdf.reserve((size_t)nuw);
    for (int64_t i = 0; i < nuw; ++i) {
      tot += uw[i].end - uw[i].start;
      R.ws.push_back(make_uint2(uw[i].start, uw[i].end));
      R.cdf.push_back(tot - 1u);
    }
    U.ws_total = tot;
    // long workspaces: 16-ary search trees (gat_device.h, WsTree) over the starts and over the cumulated lengths
    U.tree_start_off = -1;
    U.tree_cdf_off = -1;
    if (nuw > ((int64_t)1 << (4 * gat::kWsTreeLevels))) {
      fail_unit(R, GAT_ERR_CAPACITY, "unit %d: %lld workspace segments (> %lld)", u, (long long)nuw, (long long)((int64_t)1 << (4 * gat::kWsTreeLevels)));
      continue;
    }
    if (nuw > gat::kWsTreeMin || (d->merge_contigs && nuw > 2)) {   // (isochore problems: k_units_overlap asks every candidate's unit)
      // Round 6, fragmented workspaces (the reference's own test data: 6 600 - 21 000 workspace segments per contig).  A tree
      // search is four dependent 64-byte node reads; the two questions asked of a workspace have cheaper answers:
      // (a) "how many bases of [s, e) lie inside?" (SegmentList.intersect(workspace).sum(), gat/Engine.pyx:596-598): a grid over
      //     the POSITIONS, entry c = the first segment whose end lies beyond c << shift -- the segments that can overlap [s, e)
      //     are walked from entry s >> shift (one or two for segments shorter than the workspace's pieces).  About two cells
      //     per segment, at most 2^16.
      {
        const uint32_t top = uw[nuw - 1].end;                      // (coordinates are below 2^31)
        int shift = 0;
        int64_t want = 2 * nuw;
        if (want > 65536) want = 65536;
        while (((int64_t)top >> shift) + 1 > want) ++shift;
        const int64_t cells = ((int64_t)top >> shift) + 1;
        R.pgrid.assign((size_t)gat::kGridHeader + (size_t)cells + 1, 0u);
        R.pgrid[0] = (uint32_t)shift; R.pgrid[1] = (uint32_t)cells;
        int64_t j = 0;
        uint32_t span = 0, prev = 0;
        for (int64_t c = 0; c <= cells; ++c) {
          const uint64_t x = (uint64_t)c << shift;
          while (j < nuw && (uint64_t)uw[j].end <= x) ++j;
          R.pgrid[(size_t)gat::kGridHeader + (size_t)c] = (uint32_t)j;
          if (c > 0) span = std::max(span, (uint32_t)j - prev);
          prev = (uint32_t)j;
        }
        R.pgrid[(size_t)gat::kGridHeader + (size_t)cells] = (uint32_t)nuw;       // (a position beyond the last cell: nothing to walk)
        R.pgrid[2] = span;
      }
    }
    if (nuw > gat::kWsTreeMin) {
      auto build = [&](std::vector<uint32_t>& tree, auto key, uint32_t pad) {
        std::vector<uint32_t> level((size_t)nuw);
        for (int64_t i = 0; i < nuw; ++i) level[(size_t)i] = key(i);
        for (;;) {
          const size_t n = level.size(), nodes = (n + 15) / 16;
          tree.insert(tree.end(), level.begin(), level.end());
          tree.insert(tree.end(), nodes * 16 - n, pad);
          if (n <= 16) break;
          std::vector<uint32_t> up(nodes);
          for (size_t j = 0; j < nodes; ++j) up[j] = level[std::min(16 * j + 15, n - 1)];   // largest key of node j
          level.swap(up);
        }
      };
      build(R.tree_start, [&](int64_t i) { return uw[i].start; }, 0xffffffffu);
      build(R.tree_cdf, [&](int64_t i) { return R.cdf[(size_t)i]; }, 0x7fffffffu);
      // (b) "which segment holds base p of the workspace?" (SegmentListSampler.sample, gat/Engine.pyx:299-305: searchsorted over
      //     cdf[i] = cumulated length - 1 with cmpPosition): a grid over the CUMULATED lengths, g[c] = #{i : cdf[i] < c << shift},
      //     and 16-bit keys cdf[i] & mask -- within a cell the high bits agree, so #{cdf < p} = g[c] + #{i in [g[c], g[c + 1]) :
      //     key[i] < (p & mask)}.  2 bytes per segment + 2 per cell: k_place_grid keeps the image in LDS, where the trees (64 bytes
      //     per node and level, in global memory) were four dependent L2 round trips for EVERY random number of a chunk.  shift
      //     <= 16 (the keys), at most 65 535 segments (the entries), the widest cell at most 8 segments where the cells allow it.
      if (nuw > gat::kPlaceWsLds && nuw <= 65535 && tot > 1u && tot <= 0x80000000u) {
        const uint32_t topc = tot - 1u;                             // the largest p
        // (one cell per two segments -- GAT_GRID_CELL_SEGS -- to begin with; refdata, k_place_grid with eight tiles: 2.2 ms at
        //  two, 2.6 at eight: the halving search over a cell's span is LDS round trips on the lane's chain)
        const char* env_cs = gat_opt(ctx, "GAT_GRID_CELL_SEGS");
        const int64_t cell_segs = env_cs ? std::max<int64_t>(1, atoll(env_cs)) : 2;
        int shift = 16;
        while (shift > 0 && ((int64_t)topc >> shift) + 1 < nuw / cell_segs) --shift;
        for (;;) {
          const int64_t cells = ((int64_t)topc >> shift) + 1;
          std::vector<uint32_t> g((size_t)cells + 1);
          int64_t j = 0;
          uint32_t span = 0;
          for (int64_t c = 0; c <= cells; ++c) {
            const uint64_t x = (uint64_t)c << shift;
            while (j < nuw && (uint64_t)R.cdf[(size_t)j] < x) ++j;
            g[(size_t)c] = (uint32_t)j;
            if (c > 0) span = std::max(span, g[(size_t)c] - g[(size_t)c - 1]);
          }
          // finer while some cell holds more than 8 segments and the image stays below 96 KB (24 K words)
          const int64_t cells_next = shift > 0 ? ((int64_t)topc >> (shift - 1)) + 1 : cells;
          const int64_t words_next = (cells_next + 2) / 2 + (nuw + 1) / 2;
          if (span > 8 && shift > 0 && words_next <= 24576) { --shift; continue; }
          const size_t gw = ((size_t)cells + 2) / 2, kw = ((size_t)nuw + 1) / 2;
          R.cgrid.assign((size_t)gat::kGridHeader + gw + kw, 0u);
          R.cgrid[0] = (uint32_t)shift; R.cgrid[1] = (uint32_t)cells; R.cgrid[2] = span; R.cgrid[3] = (uint32_t)(gw + kw);
          uint16_t* g16 = reinterpret_cast<uint16_t*>(R.cgrid.data() + gat::kGridHeader);
          for (int64_t c = 0; c <= cells; ++c) g16[c] = (uint16_t)g[(size_t)c];
          uint16_t* k16 = reinterpret_cast<uint16_t*>(R.cgrid.data() + gat::kGridHeader + gw);
          const uint32_t mask = shift >= 32 ? 0xffffffffu : ((1u << shift) - 1u);
          for (int64_t i = 0; i < nuw; ++i) k16[i] = (uint16_t)(R.cdf[(size_t)i] & mask);
          break;
        }
      }
    }
    U.ltotal = (int32_t)ltotal;
    U.n_target = (int32_t)nus;                       // SamplerSegments places len(segments) segments
    R.nwork = nwork;
    R.active = true;
  }
  });
  for (int u = 0; u < d->n_units; ++u) {
    UnitPrep& R = prep[(size_t)u];
    UnitDev& U = P->h_units[u];
    if (R.rc == GAT_ERR_ASSERT) {                    // (the first offender in unit order, with the reference's message)
      int rc;
      if ((rc = check_list(ctx, d->segs + d->seg_off[u], d->seg_off[u + 1] - d->seg_off[u], "segment", u))) return rc;     // gat/Engine.pyx:535
      if ((rc = check_list(ctx, d->ws + d->ws_off[u], d->ws_off[u + 1] - d->ws_off[u], "workspace", u))) return rc;       // gat/Engine.pyx:536
    }
    if (R.rc) return set_err(ctx, R.rc, "%s", R.err.c_str());
    if (!R.active) continue;
    U.rank_off = (int32_t)h_rank_len.size();
    h_rank_len.insert(h_rank_len.end(), R.rank.begin(), R.rank.end());
    U.ws_off = (int32_t)h_ws.size();
    h_ws.insert(h_ws.end(), R.ws.begin(), R.ws.end());
    h_ws_cdf.insert(h_ws_cdf.end(), R.cdf.begin(), R.cdf.end());
    if (!R.tree_start.empty()) {
      U.tree_start_off = (int32_t)h_ws_tree.size();
      h_ws_tree.insert(h_ws_tree.end(), R.tree_start.begin(), R.tree_start.end());
      U.tree_cdf_off = (int32_t)h_ws_tree.size();
      h_ws_tree.insert(h_ws_tree.end(), R.tree_cdf.begin(), R.tree_cdf.end());
    }
    U.pgrid_off = -1;
    U.cgrid_off = -1;
    auto append16 = [&](const std::vector<uint32_t>& v) {          // (16-byte aligned: the images are copied in 16-byte pieces)
      while (h_ws_tree.size() & 3u) h_ws_tree.push_back(0u);
      const int32_t off = (int32_t)h_ws_tree.size();
      h_ws_tree.insert(h_ws_tree.end(), v.begin(), v.end());
      return off;
    };
    if (!R.pgrid.empty()) U.pgrid_off = append16(R.pgrid);
    if (!R.cgrid.empty()) U.cgrid_off = append16(R.cgrid);
    len_cv2[(size_t)u] = R.cv2;
    P->h_base_cap[u] = cap_for(ctx, d->sampler == GAT_SAMPLER_SEGMENTS ? std::max<int64_t>(R.nwork, d->seg_off[u + 1] - d->seg_off[u]) : R.nwork);
    work.push_back(std::make_pair(R.nwork, (int32_t)u));
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
  uint64_t work_simple = 0, work_all = 0, work_big_rank = 0;
  P->all_one_ws = !P->h_order.empty();
  P->all_cm_ok = true;
  for (int32_t u : P->h_order) {
    const UnitDev& U = P->h_units[(size_t)u];
    const bool degenerate = !(U.hist_total > 2 && U.ws_total > 1);          // k_place leaves those to k_sampler
    const bool simple = U.n_ws == 1 && U.bucket <= 1 && U.hist_total < (uint32_t)gat::kPlaceRankLds && U.ws_total > 1;
    if (!degenerate && !simple) P->all_simple = false;
    if (!degenerate && simple) {
      // k_place's cm_ok: the offset draw's mask (of the range workspace length - 2 + length) is the position draw's (of the range
      // workspace length - 1) for every length the unit can draw -- no power of two between them (k_place_scan runs only such units)
      const uint2 w0v = h_ws[(size_t)U.ws_off];
      const gat_segment w0 = {w0v.x, w0v.y};
      const uint32_t range_p = U.ws_total - 1u, mask_p = 0xffffffffu >> __builtin_clz(range_p);
      const uint32_t lmax = h_rank_len[(size_t)U.rank_off + U.hist_total - 1u];     // (rank_len[1 + rangeL], rangeL = hist_total - 2)
      if ((uint64_t)(w0.end - w0.start - 2u) + (uint64_t)lmax > (uint64_t)mask_p) P->all_cm_ok = false;
    }
    if (!degenerate) { work_all += U.hist_total; if (simple) work_simple += U.hist_total; }
    if (!degenerate && !(U.n_ws == 1 && U.bucket <= 1)) P->all_one_ws = false;
    if (!degenerate && U.hist_total >= (uint32_t)gat::kPlaceRankLds) work_big_rank += U.hist_total;
    P->max_nws = std::max(P->max_nws, U.n_ws);
    max_hist = std::max(max_hist, U.hist_total);
  }
  P->small_tables = !P->h_order.empty() && P->max_nws <= 64 && max_hist < 256;
  {
    // k_place_grid (MODE 4): some workspace is beyond k_place's LDS table and every such unit has the grid over its cumulated
    // lengths (UnitDev::cgrid_off), the largest image within the LDS the rings leave
    bool any = false, all = true;
    int32_t words = 0;
    for (int32_t u : P->h_order) {
      const UnitDev& U = P->h_units[(size_t)u];
      if (U.n_ws <= gat::kPlaceWsLds) continue;
      any = true;
      if (U.cgrid_off < 0) { all = false; break; }
      words = std::max(words, (int32_t)h_ws_tree[(size_t)U.cgrid_off + 3]);
    }
    P->grid_place = any && all && !gat_opt(ctx, "GAT_PLACE_NO_GRID");
    P->grid_lds_words = words;
  }
  P->pipe_pays = 2 * work_simple >= work_all;
  P->all_one_ws = P->all_one_ws && !P->all_simple && 2 * work_big_rank >= work_all && max_hist <= (uint32_t)gat::kPlaceWideMaxRank;
  P->long_lists = max_hist + max_hist / 8 > 1024;
  P->max_hist = max_hist;
  {
    // the split path pays when k_tail can take most units: SamplerAnnotator, lists the wave bucket sorts hold, workspaces
    // of up to kTailMaxWs segments
    size_t small_ws = 0;
    // (round 6: k_tail takes the longer workspaces too -- their position draw through the tree over the cumulated lengths, their
    //  overlaps through the position grid; GAT_TAIL_NO_LONG_WS: as before, such units are k_sampler's)
    P->tail_long_ws = !gat_opt(ctx, "GAT_TAIL_NO_LONG_WS");
    for (int32_t u : P->h_order) if (P->h_units[(size_t)u].n_ws <= gat::kTailMaxWs || P->tail_long_ws) ++small_ws;
    // (long lists: their tail places dozens of segments, not the handful k_tail keeps aside -- 0.2 % finished there on the
    //  config-4 shape -- so those problems stay with k_merge_big + k_sampler)
    P->split_path = P->sampler == GAT_SAMPLER_ANNOTATOR && !P->h_order.empty() && 2 * small_ws >= P->h_order.size() &&
                    max_hist + max_hist / 8 <= 1024 && !gat_opt(ctx, "GAT_NO_SPLIT");
  }
  // expected raw MT19937 outputs per placement under masked rejection (mask+1)/(range+1) per draw;
  // rows = that x working segments + slack, in whole 624-word blocks.  Streams that still run out
  // are redone by k_sampler from their seed.
  {
    const char* env = gat_opt(ctx, "GAT_SAMPLER_MODE");
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
      const char* env_sl = gat_opt(ctx, "GAT_RNG_SLACK");
      const double slack = env_sl ? atof(env_sl) : 1.0;
      // Spread of the raw-output count of a stream: the NUMBER of placements until the unit's bases are reproduced varies
      // by cv(length) x sqrt(n) (a renewal count) and every placement costs e outputs -- that term dominates (measured on
      // config 2: 97 / 116 / 58 outputs for units of 778 / 444 / 166 segments = e x cv x sqrt(n)) -- plus the rejection
      // noise v per placement.  5 to 7.5 sigma and the tail's few dozen outputs: a stream that runs out is redone from its seed
      // by ONE wave, placement by placement, and such a straggler (0.5 ms) is now longer than the rest of the sampler.
      const double nplace = P->sampler == GAT_SAMPLER_SEGMENTS ? (double)U.n_target : (double)U.hist_total;
      const double var_n = P->sampler == GAT_SAMPLER_SEGMENTS ? 0.0 : len_cv2[(size_t)u] * e * e;
      // The multiple follows what running out costs.  A unit of the split path (lists the wave sorts hold) that runs out of
      // rows is RESUMED by k_sampler where its lane stopped -- behind k_place's last placement, or at the consolidation k_tail
      // would have continued from -- with the stream moved up to its position by the in-LDS generator (rng_switch: the
      // seeding chain + a twist per 624 outputs, ~10 us): 3.5 sigma and 32 rows for the tail (one stream in two thousand runs
      // out; round 3's 5-7.5 sigma + 96, sized for a redo of every placement from the seed at 0.5 ms, generated 1.4x the rows
      // that were consumed: k_rng 0.48 -> 0.43 ms on config 2, 1.07 -> 0.92 on config 3).  A long list that runs out behind
      // k_tail_big's in-place unions is still redone from its seed -- milliseconds for thousands of placements: 7.5 sigma + 96.
      const char* env_s0 = gat_opt(ctx, "GAT_RNG_SIGMA_MIN");
      const char* env_s1 = gat_opt(ctx, "GAT_RNG_SIGMA_MAX");
      const char* env_tr = gat_opt(ctx, "GAT_RNG_TAIL_ROWS");
      const double s_min = env_s0 ? atof(env_s0) : 3.5, s_max = env_s1 ? atof(env_s1) : 7.5;
      const bool long_list = U.hist_total + U.hist_total / 8 > 1024 || P->sampler == GAT_SAMPLER_SEGMENTS;
      // (a resumed unit goes through k_sampler's wave-per-unit consolidation and tail: tens of microseconds for hundreds of
      //  segments, a few for fifty -- up to five sigma for the larger units of the split path: config 2, k_rng + k_sampler
      //  0.55 -> 0.51 ms, where 3.5 sigma throughout gave back in k_sampler what it saved in k_rng)
      const double sigmas = long_list ? s_max : std::min(std::max(s_min, 5.0), std::max(s_min, s_min - 0.5 + nplace / 130.0));
      const double tail_rows = env_tr ? atof(env_tr) : (long_list ? 96.0 : (nplace < 128 ? 32.0 : 48.0));
      const double need = e * nplace * slack + sigmas * std::sqrt(nplace * (v + 0.5 + var_n)) + tail_rows;
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

  HIPCHK(ctx, P->d_order.upload(P->h_order, ctx));
  {
    std::vector<int32_t> pos((size_t)std::max(1, d->n_units), -1);
    for (size_t a = 0; a < P->h_order.size(); ++a) pos[(size_t)P->h_order[a]] = (int32_t)a;
    HIPCHK(ctx, P->d_unit_pos.upload(pos, ctx));
    P->h_unit_pos = pos;
  }
  HIPCHK(ctx, P->d_rng_rows.upload(P->h_rng_rows, ctx));
  HIPCHK(ctx, P->d_contig_unit_off.upload(P->h_contig_unit_off, ctx));
  HIPCHK(ctx, P->d_contig_units.upload(P->h_contig_units, ctx));
  HIPCHK(ctx, P->d_ws.upload(h_ws, ctx));
  HIPCHK(ctx, P->d_ws_cdf.upload(h_ws_cdf, ctx));
  {
    // what a position draw needs of its workspace segment (gat/Engine.pyx:318-325) as one record
    std::vector<uint4> rec(std::max<size_t>(1, h_ws.size()));
    for (int32_t u : P->h_order) {
      const UnitDev& U = P->h_units[(size_t)u];
      for (int32_t i = 0; i < U.n_ws; ++i) {
        const size_t k = (size_t)U.ws_off + (size_t)i;
        rec[k] = make_uint4(h_ws[k].x, h_ws[k].y, i > 0 ? h_ws[k - 1].y : 0x80000000u, h_ws_cdf[k]);
      }
    }
    HIPCHK(ctx, P->d_ws_rec.upload(rec, ctx));
  }
  P->units_direct_ok = false;
  if (P->merge_contigs && P->n_contigs > 0 && P->split_path && !gat_opt(ctx, "GAT_COUNT_VIA_CONTIGS")) {
    // counting an isochore problem from the units' lists: where a segment could reach over the end of its workspace piece.
    // One bit per cell of 2^bshift bases (about two mean segment lengths): a boundary of some unit's workspace piece lies in the
    // cell -- a segment whose cells hold no boundary lies inside one piece (k_count_merged<2, .> tests the bits from its first
    // base's cell to its last's: a 64-bit window of the map)
    double bases = 0, segs = 0;
    for (int32_t u : P->h_order) { bases += (double)(uint32_t)P->h_units[(size_t)u].ltotal; segs += (double)P->h_units[(size_t)u].hist_total; }
    const double mean_len = segs > 0 ? bases / segs : 1.0;
    int bshift = 8;
    while (bshift < 24 && (double)(1u << bshift) < 2.0 * mean_len) ++bshift;
    P->bshift = bshift;
    std::vector<int64_t> boff((size_t)P->n_contigs + 1, 0);
    std::vector<uint32_t> extent((size_t)P->n_contigs, 0u);
    for (int32_t u : P->h_order) {
      const UnitDev& U = P->h_units[(size_t)u];
      extent[(size_t)U.contig] = std::max(extent[(size_t)U.contig], h_ws[(size_t)U.ws_off + (size_t)U.n_ws - 1].y);
    }
    for (int c = 0; c < P->n_contigs; ++c) boff[(size_t)c + 1] = boff[(size_t)c] + (((int64_t)(extent[(size_t)c] >> bshift) + 2) + 31) / 32 + 2;
    std::vector<uint32_t> bm((size_t)boff.back() + 2, 0u);
    for (int32_t u : P->h_order) {
      const UnitDev& U = P->h_units[(size_t)u];
      uint32_t* b = bm.data() + boff[(size_t)U.contig];
      auto mark = [&](uint32_t pos) { const int64_t j = (int64_t)(pos >> bshift); b[j >> 5] |= 1u << (j & 31); };
      for (int32_t i = 0; i < U.n_ws; ++i) { mark(h_ws[(size_t)U.ws_off + (size_t)i].x); mark(h_ws[(size_t)U.ws_off + (size_t)i].y); }
    }
    HIPCHK(ctx, P->d_bmap.upload(bm, ctx));
    HIPCHK(ctx, P->d_bmap_off.upload(boff, ctx));
    P->units_direct_ok = true;
  }
  if (h_ws_tree.empty()) h_ws_tree.assign(16, 0u);
  HIPCHK(ctx, P->d_ws_tree.upload(h_ws_tree, ctx));
  HIPCHK(ctx, P->d_rank_len.upload(h_rank_len, ctx));
  HIPCHK(ctx, P->d_cws_nseg.upload(P->h_cws_nseg, ctx));
  int rc = upload_layout(ctx, P.get());
  if (rc) return rc;
  tm.lap("units, layout, their uploads");
  double mean_seg_len = 0.0;                         // of the segments that are placed (for the merged index's scan estimate)
  {
    double bases = 0, segs = 0;
    for (int32_t u : P->h_order) { bases += (double)(uint32_t)P->h_units[(size_t)u].ltotal; segs += (double)P->h_units[(size_t)u].hist_total; }
    mean_seg_len = segs > 0 ? bases / segs : 0.0;
  }
  if (d->annotations != nullptr) {
    // the tables exist: made once for the run's annotations, shared by every segment track with these contigs
    gat_annotations* A = const_cast<gat_annotations*>(d->annotations);
    if (A->ctx != ctx) return set_err(ctx, GAT_ERR_ARG, "gat_problem_create: the annotations were made on another context");
    if (A->n_tracks != d->n_tracks || A->n_groups != d->n_contigs || (A->merge_groups != 0) != (d->merge_contigs != 0))
      return set_err(ctx, GAT_ERR_ARG, "gat_problem_create: annotations of %d tracks x %d contigs (merge %d) for a problem of %d x %d (merge %d)",
                     A->n_tracks, A->n_groups, A->merge_groups, d->n_tracks, d->n_contigs, d->merge_contigs);
    A->refs += 1;
    P->anno = A;
  } else {
    gat_annotations_desc ad;
    memset(&ad, 0, sizeof(ad));
    ad.n_tracks = d->n_tracks; ad.n_contigs = d->n_contigs; ad.merge_contigs = d->merge_contigs;
    ad.annos = d->annos; ad.anno_off = d->anno_off; ad.n_anno_lists = d->n_anno_lists; ad.anno_end = d->anno_end;
    ad.anno_group = d->anno_group; ad.mean_segment_length = mean_seg_len;
    gat_annotations* A = nullptr;
    if ((rc = gat_annotations_create(ctx, &ad, &A))) return rc;
    A->closed = true;                                // (no handle of its own: it goes with the problem)
    P->anno = A;
  }
  tm.lap("annotation tables (total)");
  HIPCHK(ctx, P->d_stat.alloc(16));                 // (8 statistics words, the status word in word 8)
  HIPCHK(ctx, stage_flush(ctx));                    // the small tables' copies (one wait for all of them)
  ctx->refs += 1;                                   // (the context outlives its handle while a problem made on it is alive)
  *out = P.release();
  return GAT_OK;
}

extern "C" void gat_problem_destroy(gat_problem* p) {
  if (!p) return;
  gat_ctx* ctx = p->ctx;
  if (ctx) {
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);         // its blocks go back to the pool: nothing may still be running on them
    ctx->stage_used = 0;
    if (p->call.blk) ctx->call_blocks.push_back(p->call.blk);     // (destroyed with a call in flight: the call is dropped)
    p->call.blk = nullptr;
  }
  delete p;                                          // (lets go of its annotation tables)
  if (ctx) ctx_release(ctx);
}

extern "C" int gat_problem_info(const gat_problem* p, int64_t* n_units, int64_t* n_contigs, int64_t* n_tracks,
                                int64_t* slab, int64_t* bytes) {
  if (!p) return set_err(nullptr, GAT_ERR_ARG, "NULL problem");
  if (n_units) *n_units = p->n_units;
  if (n_contigs) *n_contigs = p->n_contigs;
  if (n_tracks) *n_tracks = p->n_tracks;
  if (slab) *slab = p->slab_stride;
  // B_sample = 8*sum n' + 8*sum_a sum_c m + 8*A, with n' ~ n input segments
  if (bytes) {
    int rc = annotations_wait(p->ctx, p->anno);
    if (rc) return rc;
    *bytes = 8 * p->n_seg_total + 8 * p->anno->dev.total + 8 * (int64_t)p->n_tracks;
  }
  return GAT_OK;
}

extern "C" int64_t gat_problem_rng_rows(const gat_problem* p) { return p ? p->rng_rows_total : 0; }

// ------------------------------------------------------------------------------------------
// sizes of dictionary intersections for the overlap_* columns of the result rows (AnnotatorResultExtended.__init__,
// gat/Engine.pyx:1911-1928): one merge-join per (track, group) pair of normalized lists, as SegmentList.intersect walks
// them (gat/SegmentList.pyx:1469-1549: one output segment per overlapping pair), summed per track.  Input statistics on
// host threads; nothing of a sample passes here.
extern "C" int gat_intersection_sizes(const gat_segment* a, const int64_t* a_off, int32_t n_groups,
                                      const gat_segment* b, const int64_t* b_begin, const int64_t* b_end, int32_t n_tracks,
                                      int64_t* pairs_out, int64_t* bases_out) {
  if (!a_off || !b_begin || !b_end || !pairs_out || !bases_out || n_groups < 0 || n_tracks < 0)
    return set_err(nullptr, GAT_ERR_ARG, "gat_intersection_sizes: bad argument");
  parallel_for(n_tracks, [&](int64_t t) {
    int64_t pairs = 0, bases = 0;
    for (int32_t g = 0; g < n_groups; ++g) {
      const gat_segment* x = a + a_off[g];
      const int64_t nx = a_off[g + 1] - a_off[g];
      const gat_segment* y = b + b_begin[t * n_groups + g];
      const int64_t ny = b_end[t * n_groups + g] - b_begin[t * n_groups + g];
      uint32_t sum = 0;                                              // SegmentList.sum(): a Position (uint32) accumulator per list
      int64_t i = 0, j = 0;
      while (i < nx && j < ny) {
        const uint32_t lo = std::max(x[i].start, y[j].start), hi = std::min(x[i].end, y[j].end);
        if (hi > lo) { ++pairs; sum += hi - lo; }
        if (x[i].end < y[j].end) ++i; else if (y[j].end < x[i].end) ++j; else { ++i; ++j; }
      }
      bases += (int64_t)sum;
    }
    pairs_out[t] = pairs;
    bases_out[t] = bases;
  });
  return GAT_OK;
}

// IntervalDictionary.toIsochores (gat/Engine.pyx:2837-2855) for every list of a collection at once: list l (normalized: sorted,
// disjoint, no empty segment) of contig list_contig[l] against the isochore classes' segments -- cls_start / cls_end in
// coordinates (contig << 32) + position, sorted, disjoint (the classes partition the contigs), cls_label = the class --,
// truncate: one piece per (segment, class segment) that overlap (SegmentList.intersect, gat/SegmentList.pyx:1469-1549),
// else the whole segment once per class it touches (SegmentList.filter, :1401-1467).  Output list (l, k) =
// out[out_off[l * n_classes + k] .. out_off[l * n_classes + k + 1]), in position order.  Two calls: out == NULL counts
// (fills out_off, returns the total in *n_out), the second fills.  Returns 1 (nothing written) if a list is not normalized:
// the caller then splits list by list, which raises what the reference raises.  Host threads over the lists; input
// pipeline of a run (gat/IO.py:188-293), nothing of a sample passes here.
extern "C" int gat_isochore_split(const uint64_t* list_ptr, const int64_t* list_len, const int64_t* list_contig, int64_t n_lists,
                                  const int64_t* cls_start, const int64_t* cls_end, const int64_t* cls_label, int64_t n_cls,
                                  int32_t n_classes, int32_t truncate, gat_segment* out, int64_t* out_off, int64_t* n_out) {
  if (n_lists < 0 || n_cls < 0 || n_classes < 1 || !out_off || !n_out || (n_lists > 0 && (!list_ptr || !list_len || !list_contig)) ||
      (n_cls > 0 && (!cls_start || !cls_end || !cls_label)))
    return set_err(nullptr, GAT_ERR_ARG, "gat_isochore_split: bad argument");
  const int64_t K = n_classes;
  std::atomic<int> bad(0);
  const bool fill = out != nullptr;
  constexpr int64_t kBlock = 16;
  parallel_for((n_lists + kBlock - 1) / kBlock, [&](int64_t blk) {
    std::vector<int64_t> cur((size_t)K);
    std::vector<int64_t> last((size_t)K);
    for (int64_t l = blk * kBlock; l < std::min(n_lists, (blk + 1) * kBlock); ++l) {
      const gat_segment* a = reinterpret_cast<const gat_segment*>((uintptr_t)list_ptr[l]);
      const int64_t n = list_len[l], hi = list_contig[l] << 32;
      int64_t j = std::lower_bound(cls_start, cls_start + n_cls, hi) - cls_start;
      // (a class segment that begins in an earlier contig cannot reach into this one: ends stay below the next contig's base)
      const int64_t jend = std::lower_bound(cls_start, cls_start + n_cls, hi + ((int64_t)1 << 32)) - cls_start;
      if (fill) for (int64_t k = 0; k < K; ++k) cur[(size_t)k] = out_off[l * K + k];
      else for (int64_t k = 0; k < K; ++k) cur[(size_t)k] = 0;
      for (int64_t k = 0; k < K; ++k) last[(size_t)k] = -1;
      for (int64_t i = 0; i < n; ++i) {
        if (!fill && (a[i].start >= a[i].end || (i > 0 && a[i - 1].end > a[i].start))) { bad.store(1); break; }
        const int64_t as = hi + a[i].start, ae = hi + a[i].end;
        while (j < jend && cls_end[j] <= as) ++j;
        for (int64_t jj = j; jj < jend && cls_start[jj] < ae; ++jj) {
          const int64_t k = cls_label[jj];
          if (truncate) {
            if (fill) {
              gat_segment& o = out[cur[(size_t)k]];
              o.start = (uint32_t)(std::max(as, cls_start[jj]) & 0xffffffff);
              o.end = (uint32_t)(std::min(ae, cls_end[jj]) & 0xffffffff);
            }
            ++cur[(size_t)k];
          } else if (last[(size_t)k] != i) {                 // (a segment touching two pieces of one class is kept once)
            last[(size_t)k] = i;
            if (fill) out[cur[(size_t)k]] = a[i];
            ++cur[(size_t)k];
          }
        }
      }
      if (!fill) for (int64_t k = 0; k < K; ++k) out_off[l * K + k + 1] = cur[(size_t)k];   // (counts; the prefix sum follows)
    }
  });
  if (bad.load()) return 1;
  if (!fill) {
    out_off[0] = 0;
    for (int64_t g = 0; g < n_lists * K; ++g) out_off[g + 1] += out_off[g];
  }
  *n_out = out_off[n_lists * K];
  return GAT_OK;
}

// SegmentList.sum() of many lists at once (gat/SegmentList.pyx:1607: a Position accumulator per list)
extern "C" int gat_list_sums(const gat_segment* a, const int64_t* begin, const int64_t* end, int64_t n_lists, int64_t* sums_out) {
  if ((n_lists > 0 && (!begin || !end || !sums_out)) || n_lists < 0) return set_err(nullptr, GAT_ERR_ARG, "gat_list_sums: bad argument");
  constexpr int64_t kBlock = 512;
  parallel_for((n_lists + kBlock - 1) / kBlock, [&](int64_t b) {
    for (int64_t l = b * kBlock; l < std::min(n_lists, (b + 1) * kBlock); ++l) {
      uint32_t sum = 0;
      for (int64_t i = begin[l]; i < end[l]; ++i) sum += a[i].end - a[i].start;
      sums_out[l] = (int64_t)sum;
    }
  });
  return GAT_OK;
}
