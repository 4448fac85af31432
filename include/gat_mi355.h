/*
 * gat_mi355.h -- C ABI of libgat_mi355.so: the MI355X (gfx950) implementation of GAT's
 * Monte-Carlo sampling + overlap-counting hot path.
 *
 * The reference (AndreasHeger/gat 1.3.6) has no FFI; its seams for this path are Python-level
 * calls into Cython cdef classes.  Each entry point below names the reference
 * interface it replaces (file:line into the reference tree).  All buffers are caller-owned,
 * plain pointers and sizes; no Python, torch or C++ types cross the boundary.  Functions
 * return 0 on success or a negative GAT_ERR_* code; gat_last_error() gives the text.
 * One gat_ctx per host thread; a ctx is bound to one HIP device and one HIP stream.
 *
 * There is NO CPU fallback behind this ABI: every entry point that computes runs HIP kernels
 * and fails with GAT_ERR_DEVICE when no gfx950 device is usable.
 */
#ifndef GAT_MI355_H
#define GAT_MI355_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* == gat/SegmentList.pxd:31-38  struct Segment { Position start; Position end; }, half-open */
typedef struct { uint32_t start, end; } gat_segment;

typedef struct gat_ctx gat_ctx;
typedef struct gat_problem gat_problem;
typedef struct gat_annotations gat_annotations;

#define GAT_OK 0
#define GAT_ERR_VALUE (-1)      /* reference raises ValueError (gat/SegmentList.pyx:1170-1182)      */
#define GAT_ERR_ASSERT (-2)     /* reference raises AssertionError (gat/Engine.pyx:535-536, :645)   */
#define GAT_ERR_CAPACITY (-3)   /* a caller buffer or a device slab was too small                  */
#define GAT_ERR_MEMORY (-4)
#define GAT_ERR_DEVICE (-5)     /* HIP error / no usable gfx950 device                             */
#define GAT_ERR_ARG (-6)

/* counter ids == the classes of gat/Engine.pyx:1417-1472 (Counter*.name) */
#define GAT_COUNTER_NUCLEOTIDE_OVERLAP 0     /* CounterNucleotideOverlap        :1417 */
#define GAT_COUNTER_NUCLEOTIDE_DENSITY 1     /* CounterNucleotideDensity        :1428 */
#define GAT_COUNTER_SEGMENT_OVERLAP 2        /* CounterSegmentOverlap           :1443 */
#define GAT_COUNTER_SEGMENT_MIDOVERLAP 3     /* CounterSegmentMidpointOverlap   :1450 */
#define GAT_COUNTER_ANNOTATION_OVERLAP 4     /* CounterAnnotationOverlap        :1458 */
#define GAT_COUNTER_ANNOTATION_MIDOVERLAP 5  /* CounterAnnotationMidpointOverlap :1465 */
#define GAT_NUM_COUNTERS 6

/* samplers */
#define GAT_SAMPLER_ANNOTATOR 0   /* SamplerAnnotator: place until the workspace overlap matches (default)   */
#define GAT_SAMPLER_SEGMENTS 1    /* SamplerSegments: len(segments) placements, no consolidation; the lists are */
                                  /* normalized only by fromIsochores, so counters need isochore keys           */

/*
 * Flat description of what gat.computeSample (gat/__init__.py:494-591) walks for one segment
 * track: the isochore units in list(segs.keys()) order (:531), their contig after
 * IntervalDictionary.fromIsochores (gat/Engine.pyx:2857-2876), and the contig-level
 * annotations / workspace the counters see (:580-587).  All pointers are HOST pointers and are
 * only read during gat_problem_create.  Every list must be normalized (sorted, disjoint,
 * non-empty segments: gat/SegmentList.pyx:697) and all coordinates must be < 2^31 (the
 * reference's int32 lmin/lmax, gat/SegmentList.pyx:68-77, are order-preserving only there).
 */
typedef struct {
  int32_t n_units;              /* isochore keys, reference order                               */
  const gat_segment* segs;      /* per-unit segment lists, concatenated                         */
  const int64_t* seg_off;       /* n_units+1                                                    */
  const gat_segment* ws;        /* per-unit workspace lists, concatenated                       */
  const int64_t* ws_off;        /* n_units+1                                                    */
  const int32_t* unit_contig;   /* n_units: contig index, or -1 for a unit computeSample skips  */
  int32_t n_contigs;            /* contigs in list(sample.keys()) order after fromIsochores     */
  int32_t merge_contigs;        /* 1 if keys are "contig.isochore": fromIsochores merges(0)     */
  int32_t n_tracks;             /* annotation tracks                                            */
  const gat_segment* annos;     /* [track][contig] lists, concatenated                          */
  const int64_t* anno_off;      /* n_tracks*n_contigs+1                                         */
  const int64_t* cws_nseg;      /* n_contigs: len(contig_workspace[contig]) (Engine.pyx:1437)   */
  uint32_t bucket_size;         /* SamplerAnnotator(bucket_size, nbuckets): gat/Engine.pyx:498  */
  int32_t nbuckets;
  int32_t sampler;              /* GAT_SAMPLER_ANNOTATOR (gat/Engine.pyx:445) or GAT_SAMPLER_SEGMENTS (:653) */
  /* Optional (all 0 / NULL: annos / anno_off are the [track][contig] lists above).  With anno_group set, the caller hands
   * over the annotation lists as it holds them -- one per (track, isochore key), the `annotations` argument of
   * UnconditionalSampler.sample (gat/__init__.py:704) -- and the library forms computeSample's contig_annotations itself
   * (gat/__init__.py:716-718 -> IntervalDictionary.fromIsochores, gat/Engine.pyx:2857-2876: the lists of a contig
   * concatenated and, when merge_contigs, sorted and merge(0)d), on host threads: */
  int64_t n_anno_lists;         /* number of lists; list l = annos[anno_off[l] .. anno_end[l])                       */
  const int64_t* anno_end;      /* n_anno_lists, or NULL: anno_off has n_anno_lists + 1 entries (CSR)                 */
  const int32_t* anno_group;    /* n_anno_lists: track * n_contigs + contig of the list's key, or -1 (its contig has   */
                                /* no unit that computeSample samples: the counters never see it)                      */
  /* Optional: the annotation side as an object made before (gat_annotations_create) -- the reference passes the SAME
   * annotations to the sampling of every segment track (gat/__init__.py:971-1010); a host that loops over tracks builds
   * their tables once.  n_tracks / n_contigs / merge_contigs must be the object's; annos / anno_* above are ignored. */
  const gat_annotations* annotations;
} gat_problem_desc;

/* The annotation side of gat_problem_desc by itself: the tracks' lists per contig -- [track][contig] CSR, or with
 * anno_group one list per (track, isochore key) that the library groups as IntervalDictionary.fromIsochores does
 * (gat/Engine.pyx:2857-2876) -- for the contigs, IN THE ORDER, of the problems that will count against it. */
typedef struct {
  int32_t n_tracks;
  int32_t n_contigs;
  int32_t merge_contigs;
  const gat_segment* annos;
  const int64_t* anno_off;      /* n_tracks*n_contigs+1, or per list (see anno_end)                                   */
  int64_t n_anno_lists;         /* with anno_group: number of lists                                                    */
  const int64_t* anno_end;
  const int32_t* anno_group;
  double mean_segment_length;   /* of the segments that will be counted against them, 0 if unknown: picks the form of  */
                                /* the merged index (speed only, never a result)                                       */
  int32_t flags;                /* 0 or an OR of GAT_ANNOTATIONS_ASYNC, GAT_ANNOTATIONS_NUCLEOTIDE_ONLY                 */
} gat_annotations_desc;

/* gat_annotations_create returns at once and a thread of the library builds the tables (its own stream and staging
 * buffer): the arrays of the desc must stay valid until gat_annotations_wait -- or a gat_wait of a call counted against the
 * object -- has returned.  A problem made against such an object can be sampled straight away: gat_sample_and_count_enqueue
 * puts the sampler's kernels on the stream and the count kernels follow when the tables are there (at the latest in
 * gat_wait).  Errors of the build are reported by the call that first needs the tables.  Honoured where the shape alone
 * tells which count kernel will run (four tracks or more); otherwise the build is synchronous. */
#define GAT_ANNOTATIONS_ASYNC 1
/* The object will only be counted against with the nucleotide counters (GAT_COUNTER_NUCLEOTIDE_OVERLAP / _DENSITY: what
 * gat.run() does unless the caller asked for segment or annotation counters).  Where the merged index of all tracks is built
 * (four tracks or more, or lists beyond the per-track kernel's LDS tile) the per-track tables -- starts / ends / running
 * lengths and the position grids, a third of the build -- are left out; gat_sample_and_count with any other counter on a
 * problem made against such an object fails with GAT_ERR_ARG.  Ignored where no merged index is built. */
#define GAT_ANNOTATIONS_NUCLEOTIDE_ONLY 2

/* per-call statistics of gat_sample_and_count / gat_sample (device time from HIP events on the
 * ctx stream; counts summed over all (sample, unit) work units of the call).  ms_total and ms_count_main are always
 * measured; the other ms_* fields -- an event behind every kernel of the sampler, 2 % of a 10 000-sample call and 6 % of
 * a 1 250-sample one -- only after gat_ctx_set_kernel_times(ctx, 1), else 0 */
typedef struct {
  float ms_sampler;             /* placement + consolidation kernel                             */
  float ms_contig;              /* fromIsochores kernel (0 when keys carry no isochore)         */
  float ms_count;               /* overlap-count kernel(s)                                      */
  float ms_total;               /* whole call on the stream                                     */
  int64_t n_placed;             /* segments placed (sls.sample calls kept, gat/Engine.pyx:628)  */
  int64_t n_draws;              /* raw MT19937 outputs consumed                                 */
  int64_t n_sampled_segments;   /* contig-level segments returned (gat_sample only)               */
  int64_t n_unsuccessful;       /* sum of nunsuccessful_rounds (gat/Engine.pyx:570-572)         */
  int64_t n_retried;            /* work units redone with a larger slab                         */
  int64_t n_full_units;         /* work units run without the lane-parallel front end           */
  float ms_count_main;          /* the dominant count kernel alone (see count_kernel)             */
  float ms_rng;                 /* split of ms_sampler: k_rng (MT19937 rows)                      */
  float ms_place;               /*   k_place (placement up to the first consolidation)            */
  float ms_merge;               /*   first consolidation: k_consolidate, or k_merge_big for long lists */
  float ms_tail;                /*   everything behind it: k_tail + k_finalize + k_sampler        */
  int32_t count_kernel;         /* GAT_COUNT_KERNEL_* the call used for the overlap counters      */
  float ms_ktail;               /*   of ms_tail: k_tail (the loop's tail, one stream per lane)    */
  float ms_finalize;            /*   of ms_tail: k_finalize (merged list + extras -> final list)   */
  int64_t n_tail_units;         /* work units finished by k_tail, or (long lists) carried through their placement rounds
                                   by k_tail_big before k_sampler resumed them                        */
  int64_t lists_from_records;   /* != 0: no final unit lists were written; their consumer (k_contig or k_count_seg) took
                                   the merged lists and k_tail's records                              */
  int64_t n_index_entries;      /* k_count_merged: 4-byte words of index its scans read (grid cells, entries), and ...  */
  int64_t n_index_lookups;      /*   ... the sample segments it looked up: what the kernel's byte model is made of       */
  int64_t n_batches;            /* batches the call was cut into (the scratch budget decides how many samples one holds)   */
  int64_t merged_form;          /* k_count_merged: how its scans fetched the index: 8 blocks of eight entries, 2 pairs,    */
                                /* 1 the grid cell's record (its first two entries) + pairs; 0: the kernel did not run     */
  int64_t n_resumed_units;      /* work units whose pre-generated random rows ran out and whose stream went on from the     */
                                /* generator moved up to its position (instead of a run in full: n_full_units)              */
  int64_t n_straddle_candidates; /* isochore problems whose contig lists were only concatenated (nucleotide counters; k_contig<., true>):  */
                                /* segments with a workspace boundary in their cells, of units that have a segment reaching out of   */
                                /* their workspace -- what k_units_overlap looked at; 0: the sorted, merged contig lists were made   */
  int64_t n_unit_overlaps;      /* ... and the overlaps between different units' segments it took off the sums again -- what      */
                                /* IntervalDictionary.fromIsochores' merge(0) unites (gat/Engine.pyx:2857-2876)                   */
  int64_t kernel_times;         /* the ms_* split of the sampler's kernels: 1 recorded, 0 not asked for (gat_ctx_set_kernel_times),   */
                                /* -1 asked for but NOT recorded -- the events exist once per context and another problem's timed  */
                                /* call was in flight (gat_amd.run() keeps two segment tracks' calls in flight): the fields read 0 */
  int64_t n_queued_units;       /* work units k_tail (lists of up to 1 024 segments) or k_resume_big (longer ones) left to     */
                                /* k_sampler's queue: a wave each, the slow way (0 when neither ran: k_sampler took every unit)  */
} gat_stats;

#define GAT_COUNT_KERNEL_NONE 0
#define GAT_COUNT_KERNEL_SEG 1      /* k_count_seg: annotation tiles in LDS, sample segments looked up   */
#define GAT_COUNT_KERNEL_SWAP 2     /* k_count_swap: sample list indexed in LDS, annotation tracks streamed */
#define GAT_COUNT_KERNEL_MERGED 3   /* k_count_merged: one look-up per sample segment in a merged index of all tracks */

/* ---- context ---------------------------------------------------------------------------- */
/* device_id: HIP device ordinal.  stream: a hipStream_t to run on (e.g. a torch stream's handle; the default stream is
 * named by hipStreamLegacy, (hipStream_t)1 -- its own handle is NULL) or NULL to create a private non-blocking one. */
int gat_ctx_create(gat_ctx** out, int device_id, void* stream);
void gat_ctx_destroy(gat_ctx* ctx);
const char* gat_last_error(const gat_ctx* ctx);      /* ctx may be NULL: last error of the thread */
const char* gat_version(void);
int gat_ctx_synchronize(gat_ctx* ctx);
/* The tuning / testing knobs (GAT_*) of a context.  None is needed for normal use.  A knob's value is the
 * context's own (set here: value "" = not set, whatever the environment says; NULL = back to the process's), else the process's
 * environment variable of that name AS IT WAS when the library first looked (one snapshot: nothing reads the environment on a
 * call's path, and two host threads with a context each do not see each other's settings).  Values are read where they act:
 * a knob that shapes a problem or the annotation tables at their creation, one that picks a kernel at the call. */
int gat_ctx_set_option(gat_ctx* ctx, const char* key, const char* value);
const char* gat_ctx_get_option(const gat_ctx* ctx, const char* key);   /* NULL: not set */
/* the hipStream_t the context's work is enqueued on (the one given to gat_ctx_create, or its private stream): a host that
 * runs other work on the device -- a collective over the count matrix, a copy -- orders it against the library's with
 * events on this stream instead of synchronising the device */
void* gat_ctx_stream(const gat_ctx* ctx);
/* per-kernel device times in gat_stats (see there) on / off; off by default.  GAT_KERNEL_TIMES=1 in the environment
 * switches them on for every context. */
int gat_ctx_set_kernel_times(gat_ctx* ctx, int on);

/* device memory helpers so a ctypes host needs no other HIP binding */
int gat_dev_alloc(gat_ctx* ctx, void** out, size_t bytes);
int gat_dev_free(gat_ctx* ctx, void* p);
int gat_memcpy_d2h(gat_ctx* ctx, void* dst_host, const void* src_dev, size_t bytes);
int gat_memcpy_h2d(gat_ctx* ctx, void* dst_dev, const void* src_host, size_t bytes);

/* ---- problem ---------------------------------------------------------------------------- */
/* Uploads the inputs once and hoists what SamplerAnnotator.sample recomputes on every call
 * (gat/Engine.pyx:543-565: filter, ltotal, getLengthDistribution, both sampler CDFs).
 * Returns GAT_ERR_VALUE where the reference's first sample() would raise ValueError.
 * Limits (GAT_ERR_CAPACITY beyond them): 2^24 workspace segments per unit, 2^31 segments per sample; no limit on
 * the number of units or contigs.  There is no limit on the segments of a unit: lists that do not fit
 * on-chip memory are worked on in device memory. */
int gat_problem_create(gat_ctx* ctx, const gat_problem_desc* desc, gat_problem** out);
void gat_problem_destroy(gat_problem* p);

/* Uploads the annotation tracks and builds what the counters look them up in (SoA lists + position grids, the merged
 * index of all tracks): the part of gat_problem_create that does not depend on the segment track.  Problems created with
 * desc->annotations = this object share it; it may be destroyed before them (it lives until the last one is gone). */
int gat_annotations_create(gat_ctx* ctx, const gat_annotations_desc* desc, gat_annotations** out);
int gat_annotations_wait(gat_ctx* ctx, gat_annotations* a);      /* an asynchronous build has finished; its error, if any */
void gat_annotations_destroy(gat_annotations* a);

/* ---- the batch seam ---------------------------------------------------------------------
 * Replaces UnconditionalSampler.sample / computeSamples / computeSample
 * (gat/__init__.py:654-778, :494-591) for samples [sample_begin, sample_end):
 * per sample, every unit is sampled (SamplerAnnotator.sample, gat/Engine.pyx:515-646),
 * units are re-combined per contig (fromIsochores) and every counter x annotation track is
 * evaluated and summed over contigs.
 *
 * Random streams ("per-unit" contract): work unit (sample s, unit u) draws from
 * numpy's legacy RandomState seeded with (seed + s*n_units + u) mod 2^32, i.e. exactly what
 * numpy.random.seed(that) followed by the reference's sampler.sample() consumes.  Results do
 * not depend on how samples are split over calls, streams or GPUs.
 *
 * counts_dev: DEVICE pointer to n_counters*n_tracks*(sample_end-sample_begin) 8-byte slots laid
 * out [counter][track][sample]; int64 for the integer counters, IEEE double for
 * nucleotide-density.  Work is enqueued on the ctx stream; the call returns after the
 * kernels have completed and per-unit status words have been checked.  A context may be destroyed before the problems
 * made on it: it lives until the last of them is gone. */
int gat_sample_and_count(gat_ctx* ctx, gat_problem* p,
                         const int32_t* counter_ids, int n_counters,
                         uint32_t seed, int64_t sample_begin, int64_t sample_end,
                         void* counts_dev, gat_stats* stats /* nullable */);

/* The same call in two halves, for a host that has work of its own to do while the device samples (gat.run computes the
 * observed counts, gat/__init__.py:933-940, and the sizes of its result rows, :1000-1068, around the sampling; the
 * reference's process pool -- map_async, gat/__init__.py:681-700 -- is asynchronous in the same way).
 * gat_sample_and_count_enqueue validates the arguments, enqueues the call's batches on the ctx stream (up to 8 ahead) and
 * returns; counts_dev must stay valid and untouched until gat_wait.  gat_wait(ctx, p, stats) blocks until the batches have
 * completed, checks their status words -- a batch that has to be repeated (a unit's region of the slab overflowed) is
 * repeated in here, with everything that was enqueued behind it -- and reports what gat_sample_and_count would have
 * (GAT_ERR_ASSERT where the reference's sampler asserts, :645).  One call in flight per problem; several problems of one
 * context may each have one (they run one behind the other on the context's stream; gat_wait waits for the end of ITS call
 * -- an event behind its last batch --, not for the stream: what another problem has enqueued behind it keeps running
 * while the host reads the results and enqueues the next call).  gat_problem_destroy drops a call in flight.  gat_sample_and_count(...) == enqueue + wait. */
int gat_sample_and_count_enqueue(gat_ctx* ctx, gat_problem* p,
                                 const int32_t* counter_ids, int n_counters,
                                 uint32_t seed, int64_t sample_begin, int64_t sample_end, void* counts_dev);
int gat_wait(gat_ctx* ctx, gat_problem* p, gat_stats* stats /* nullable */);

/* Sampler only: replaces sampler.sample() + sample.fromIsochores() (gat/__init__.py:541, :563)
 * for a range of samples and returns the contig-level lists to the HOST:
 * off_host has (sample_end-sample_begin)*n_contigs+1 entries, segments of (sample i, contig c)
 * are out_host[off[i*n_contigs+c] .. off[i*n_contigs+c+1]).  GAT_ERR_CAPACITY if cap is too
 * small (off_host[last] then holds the required size). */
int gat_sample(gat_ctx* ctx, gat_problem* p, uint32_t seed,
               int64_t sample_begin, int64_t sample_end,
               gat_segment* out_host, int64_t cap, int64_t* off_host, gat_stats* stats);

/* The same at unit level: what sampler.sample(segs[isochore], workspace[isochore]) returned for every (sample, unit)
 * before fromIsochores (gat/__init__.py:541; the lists --output-samples-pattern writes, :549-559).  off_host has
 * (sample_end-sample_begin)*n_units+1 entries; units computeSample skips are empty. */
int gat_sample_units(gat_ctx* ctx, gat_problem* p, uint32_t seed,
                     int64_t sample_begin, int64_t sample_end,
                     gat_segment* out_host, int64_t cap, int64_t* off_host, gat_stats* stats);

/* Counters only, on caller-provided lists: replaces Engine.computeCounts
 * (gat/Engine.pyx:2164-2204; observed counts) and counter(segments, annotations, workspace)
 * (gat/Engine.pyx:1417-1472).  lists: n_lists*n_groups segment lists (HOST, CSR via list_off),
 * annotations: n_tracks*n_groups lists (HOST, CSR via anno_off), ws_nseg[n_groups] =
 * len(workspace[group]).  counts_host: [n_counters][n_tracks][n_lists] 8-byte slots. */
int gat_count_lists(gat_ctx* ctx, const int32_t* counter_ids, int n_counters,
                    const gat_segment* lists, const int64_t* list_off, int64_t n_lists,
                    const gat_segment* annos, const int64_t* anno_off, int32_t n_tracks,
                    const int64_t* ws_nseg, int32_t n_groups, void* counts_host);

/* The same with the annotation lists given as ranges of one array: list (track t, group g) =
 * annos[anno_begin[t * n_groups + g] .. anno_end[t * n_groups + g]) -- what a host that keeps every dictionary's lists
 * in one array passes without copying them into group order. */
int gat_count_list_ranges(gat_ctx* ctx, const int32_t* counter_ids, int n_counters,
                          const gat_segment* lists, const int64_t* list_off, int64_t n_lists,
                          const gat_segment* annos, const int64_t* anno_begin, const int64_t* anno_end, int32_t n_tracks,
                          const int64_t* ws_nseg, int32_t n_groups, void* counts_host);

/* Sizes of the intersection of two interval dictionaries, for the overlap_* columns of a result row: replaces
 * `overlap = track_segments.clone(); overlap.intersect(annotation_segments); overlap.counts(), overlap.sum()`
 * (AnnotatorResultExtended.__init__, gat/Engine.pyx:1911-1928; SegmentList.intersect, gat/SegmentList.pyx:1469-1549).
 * Host arithmetic on the INPUTS of a run (one merge-join per pair of lists, on host threads); no sample passes here.
 * a: n_groups normalized lists (CSR a_off); b: n_tracks * n_groups normalized lists as ranges of one array, list
 * (t, g) = b[b_begin[t * n_groups + g] .. b_end[..]).  Per track t: pairs_out[t] = number of overlapping (a, b) segment
 * pairs over all groups (= segments of the intersection), bases_out[t] = their total overlap. */
int gat_intersection_sizes(const gat_segment* a, const int64_t* a_off, int32_t n_groups,
                           const gat_segment* b, const int64_t* b_begin, const int64_t* b_end, int32_t n_tracks,
                           int64_t* pairs_out, int64_t* bases_out);

/* Sum of the segment lengths of every list a[begin[l] .. end[l]): SegmentList.sum() (gat/SegmentList.pyx:1607-1614, a
 * Position -- uint32 -- accumulator per list), for the *_size columns of the result rows (AnnotatorResultExtended,
 * gat/Engine.pyx:1911-1928) of a run over 10^4 lists.  Host arithmetic on the inputs, like gat_intersection_sizes. */
int gat_list_sums(const gat_segment* a, const int64_t* begin, const int64_t* end, int64_t n_lists, int64_t* sums_out);

/* IntervalDictionary.toIsochores (gat/Engine.pyx:2837-2855; what IO.applyIsochores, gat/IO.py:188-293, does to the segments,
 * annotations and workspace of a run) for every list of a collection in one call, on host threads.  list l: list_len[l]
 * segments at address list_ptr[l] (HOST; normalized: sorted, disjoint, none empty) of contig list_contig[l]; the isochore
 * classes' segments in coordinates (contig << 32) + position, sorted by start and disjoint (the classes partition the
 * contigs), cls_label[j] = class of segment j in [0, n_classes).  truncate != 0: a segment is cut at the class boundaries
 * (SegmentList.intersect, gat/SegmentList.pyx:1469-1549), else every class a segment touches receives it whole
 * (SegmentList.filter, :1401-1467).  Result list (l, k) = out[out_off[l * n_classes + k] .. out_off[l * n_classes + k + 1]).
 * Two calls: with out == NULL the lists are counted (out_off[0 .. n_lists * n_classes] filled, total in *n_out), with out
 * and the same out_off they are written.  Returns GAT_OK, GAT_ERR_ARG, or 1 when a list is not normalized (nothing is
 * written: the host then splits list by list and raises what the reference raises).  Input pipeline; no sample passes here. */
int gat_isochore_split(const uint64_t* list_ptr, const int64_t* list_len, const int64_t* list_contig, int64_t n_lists,
                       const int64_t* cls_start, const int64_t* cls_end, const int64_t* cls_label, int64_t n_cls,
                       int32_t n_classes, int32_t truncate, gat_segment* out, int64_t* out_off, int64_t* n_out);

/* ---- the reference's own random stream ---------------------------------------------------
 * scripts/gat-run.py:267-271 seeds numpy's global generator ONCE and every (sample, unit) of the run -- in the order of
 * gat/__init__.py:531-541, segment track after segment track -- draws from that one MT19937.  gat_sample_and_count's
 * per-unit streams are what makes the samples independent work; this entry reproduces an unpatched reference instead,
 * table for table, at the speed of one stream: one wave runs the samples [0, n_samples) one after the other.
 * mt_state: GAT_MT_STATE_WORDS words, the 624 state words + the position (numpy's `pos`, 624 after seeding); read at
 * entry, written back at return, so that calls (batches, segment tracks) continue each other.  gat_mt19937_seed fills
 * it as numpy.random.seed(seed) does for an integer seed.  Counts as in gat_sample_and_count with sample_begin 0. */
#define GAT_MT_STATE_WORDS 625
void gat_mt19937_seed(uint32_t seed, uint32_t* mt_state);
int gat_sample_and_count_serial(gat_ctx* ctx, gat_problem* prob, const int32_t* counter_ids, int n_counters,
                                uint32_t* mt_state, int64_t n_samples, void* counts_dev, gat_stats* stats);

/* ---- null-distribution statistics on the device -----------------------------------------
 * The numbers AnnotatorResult takes from a row of sampled counts (makeEnrichmentStatistics / getTwoSidedPValue,
 * gat/Engine.pyx:1635-1718, :1543-1576), computed from the count matrix where gat_sample_and_count left it:
 * counts_dev = n_rows rows of n_samples 8-byte slots (int64, or IEEE double for rows with is_double != 0 --
 * nucleotide-density).  Per row r, out_host[8r..8r+8) = { numpy.mean, the sum of squared deviations from it as numpy.std
 * forms it (std = sqrt(that / n_samples); bit for bit: numpy's chunked pairwise summation is restated), the values at sorted positions lo_index and hi_index, the number of samples < vals[r], the
 * number == vals[r], 0, 0 }.  The caller turns them into expected / CI / stddev / p-value as the reference does. */
int gat_null_stats(gat_ctx* ctx, const void* counts_dev, int64_t n_rows, int64_t n_samples,
                   const uint8_t* is_double_host, const double* vals_host, int64_t lo_index, int64_t hi_index,
                   double* out_host);

/* ---- multi-GPU: the one collective of the path ---------------------------------------------
 * Replaces the result collation of the reference's process pool (gat/__init__.py:681-700, :770-774):
 * every rank has computed the columns of its own contiguous sample range (gat_sample_and_count with
 * [sample_begin, sample_end) = its shard) and ONE all-gather over xGMI makes every rank hold all of them.
 * RCCL is loaded at the first call (dlopen of librccl.so: the library has no link-time dependency on it);
 * a host without it gets GAT_ERR_DEVICE.  Ranks are processes, one per GPU; rank 0 calls
 * gat_comm_unique_id and passes the 128 bytes to the others by whatever means the host has (the Python host
 * uses torch.distributed's store or MPI; a file works).  torch.distributed's "nccl" backend is the same
 * RCCL: gat_amd/distributed.py uses it when a process group exists and needs none of this. */
typedef struct gat_comm gat_comm;
#define GAT_COMM_ID_BYTES 128
int gat_comm_unique_id(void* id_out /* GAT_COMM_ID_BYTES */);
int gat_comm_create(gat_ctx* ctx, gat_comm** out, int n_ranks, int rank, const void* id /* GAT_COMM_ID_BYTES */);
void gat_comm_destroy(gat_comm* comm);
/* send_dev: this rank's n_slots 8-byte slots (its [counter][track][shard] block); recv_dev: n_ranks * n_slots
 * slots, rank r's block at r * n_slots.  Enqueued on the ctx stream; returns after completion. */
int gat_allgather_counts(gat_ctx* ctx, gat_comm* comm, const void* send_dev, void* recv_dev, int64_t n_slots);
/* Which RCCL the calls above resolved: 1 = one the process had mapped already (a host with torch holds torch/lib/librccl.so: it
 * is found with RTLD_NOLOAD / among the process's objects and used -- a process never holds two RCCLs), 0 = loaded afresh
 * (librccl.so, librccl.so.1, /opt/rocm/lib/librccl.so, or $GAT_RCCL_LIB), -1 = none found. */
int gat_comm_library_preloaded(void);

/* queries */
int gat_problem_info(const gat_problem* p, int64_t* n_units, int64_t* n_contigs, int64_t* n_tracks,
                     int64_t* slab_segments_per_sample, int64_t* algorithmic_bytes_per_sample);

/* raw MT19937 outputs generated ahead per sample (over all units: the rows of k_rng), to set against gat_stats.n_draws, the
 * outputs the samples consumed */
int64_t gat_problem_rng_rows(const gat_problem* p);

#ifdef __cplusplus
}
#endif
#endif
