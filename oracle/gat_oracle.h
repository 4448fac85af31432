/*
 * gat_oracle.h -- CPU restatement of the GAT sampling + overlap-counting hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product (gat_amd/, include/, bench.py's
 * timed GPU leg) may link or call this file; only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg use it, and only as the checker / reported CPU baseline.
 *
 * Every function restates one function of the reference (AndreasHeger/gat 1.3.6) and
 * cites it as file:line into /root/reference.  Parity of this restatement is PINNED:
 * tests/golden/ holds vectors produced by the reference itself (scratch build, see
 * tests/golden/make_goldens.py) and tests/test_oracle_*.py check this file against them.
 */
#ifndef GAT_ORACLE_H
#define GAT_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* gat/SegmentList.pxd:31-38: Position = unsigned int, PositionDifference = int,
 * struct Segment { Position start; Position end; } (half-open). */
typedef struct { uint32_t start, end; } gato_segment;

/* error codes (the reference raises Python exceptions in these places) */
#define GATO_OK 0
#define GATO_ERR_VALUE (-1)     /* ValueError  (gat/SegmentList.pyx:1170-1182) */
#define GATO_ERR_ASSERT (-2)    /* AssertionError (gat/Engine.pyx:535-536, :645; SegmentList.pyx:560) */
#define GATO_ERR_CAPACITY (-3)  /* caller buffer too small (no reference equivalent) */
#define GATO_ERR_MEMORY (-4)

/* ---- utils/gat_utils.c:36-60 searchsorted (left bisect) ------------------------------ */
long gato_searchsorted_u32(const uint32_t* base, size_t n, uint32_t target);   /* cmpPosition */
long gato_searchsorted_seg(const gato_segment* base, size_t n, uint32_t start); /* cmpSegments */

/* ---- gat/SegmentList.pyx interval algebra (in place unless noted) ------------------- */
size_t   gato_normalize(gato_segment* s, size_t n);                     /* :697-754 */
size_t   gato_merge(gato_segment* s, size_t n, int32_t distance);       /* :756-816 */
int      gato_check(const gato_segment* s, size_t n);                   /* :818-851, 1 if normalized */
size_t   gato_filter(gato_segment* a, size_t na,
                     const gato_segment* b, size_t nb);                 /* :1401-1467 */
long     gato_intersect(const gato_segment* a, size_t na, const gato_segment* b, size_t nb,
                        gato_segment* out, size_t cap);                 /* :1469-1549 */
uint32_t gato_sum(const gato_segment* s, size_t n);                     /* :1607-1616 */
uint32_t gato_overlap_with_segments(const gato_segment* a, size_t na,
                                    const gato_segment* b, size_t nb);  /* :1026-1076 */
uint32_t gato_intersection_with_segments(const gato_segment* a, size_t na,
                                         const gato_segment* b, size_t nb,
                                         int midpoint);                 /* :1078-1146 */
int      gato_get_insertion_point(const gato_segment* s, size_t n,
                                  uint32_t start, uint32_t end);        /* :853-887 */
int      gato_trim_ends(gato_segment* s, size_t n, uint32_t pos, uint32_t size,
                        int forward);                                   /* :545-597 */
/* :1148-1184; hist has nbuckets entries; returns GATO_OK / GATO_ERR_VALUE */
int      gato_length_distribution(const gato_segment* s, size_t n, uint32_t bucket_size,
                                  int nbuckets, int64_t* hist, uint32_t* bucket_size_out);

/* ---- numpy legacy RandomState (MT19937 + masked rejection) ---------------------------
 * Third-party: numpy (requirements.txt pins numpy>=1.6.1; semantics of numpy>=1.17,
 * probed on 2.2.6): numpy/random/src/mt19937/mt19937.c (init_genrand, genrand) and
 * numpy/random/src/distributions/distributions.c (random_bounded_uint64 masked path).
 * Reference call sites: gat/Engine.pyx:299,326,420,433,620; scripts/gat-run.py:267-271. */
typedef struct { uint32_t mt[624]; int mti; uint64_t ndraws; } gato_rng;
void     gato_rng_seed(gato_rng* r, uint32_t seed);
uint32_t gato_rng_u32(gato_rng* r);
int64_t  gato_randint(gato_rng* r, int64_t lo, int64_t hi);  /* numpy.random.randint(lo, hi) */

/* ---- gat/Engine.pyx samplers --------------------------------------------------------- */
/* SamplerAnnotator.sample (gat/Engine.pyx:515-646) incl. HistogramSampler (:387-435) and
 * SegmentListSampler (:245-348).  segs and ws must be normalized.  out receives the
 * sampled, merged(0), workspace-filtered list. */
int gato_sampler_annotator(gato_rng* rng,
                           const gato_segment* segs, size_t nsegs,
                           const gato_segment* ws, size_t nws,
                           uint32_t bucket_size, int nbuckets,
                           gato_segment* out, size_t out_cap, size_t* nout,
                           int* nunsuccessful_rounds);

/* SamplerSegments.sample (gat/Engine.pyx:695-737): len(segments) placements, no consolidation; the
 * result is in placement order (unsorted, may overlap). */
int gato_sampler_segments(gato_rng* rng,
                          const gato_segment* segs, size_t nsegs,
                          const gato_segment* ws, size_t nws,
                          uint32_t bucket_size, int nbuckets,
                          gato_segment* out, size_t out_cap, size_t* nout);

/* ---- batch seam: gat/__init__.py:494-591 computeSample over a sample range ------------
 * Flat (CSR) problem description, shared with the product's C ABI (include/gat_mi355.h). */
typedef struct {
  int32_t n_units;               /* isochore keys in list(segs.keys()) order (gat/__init__.py:531) */
  const gato_segment* segs;      /* concatenated per-unit segment lists */
  const int64_t* seg_off;        /* n_units+1 */
  const gato_segment* ws;        /* concatenated per-unit workspace lists */
  const int64_t* ws_off;         /* n_units+1 */
  const int32_t* unit_contig;    /* n_units: contig index of the unit (fromIsochores, Engine.pyx:2857) */
  int32_t n_contigs;             /* contigs in list(sample.keys()) order after fromIsochores */
  int32_t merge_contigs;         /* 1 if any key has a '.', i.e. fromIsochores applies merge(0) */
  int32_t n_tracks;              /* annotation tracks */
  const gato_segment* annos;     /* [track][contig] concatenated, contig-level */
  const int64_t* anno_off;       /* n_tracks*n_contigs+1 */
  const int64_t* cws_nseg;       /* n_contigs: len(contig_workspace[contig]) (Engine.pyx:1437) */
  uint32_t bucket_size;          /* SamplerAnnotator(bucket_size, nbuckets) */
  int32_t nbuckets;
  int32_t sampler;               /* 0: SamplerAnnotator, 1: SamplerSegments (gat/Engine.pyx:653) */
} gato_problem;

/* counter ids (gat/Engine.pyx:1417-1472) */
#define GATO_COUNTER_NUCLEOTIDE_OVERLAP 0
#define GATO_COUNTER_NUCLEOTIDE_DENSITY 1
#define GATO_COUNTER_SEGMENT_OVERLAP 2
#define GATO_COUNTER_SEGMENT_MIDOVERLAP 3
#define GATO_COUNTER_ANNOTATION_OVERLAP 4
#define GATO_COUNTER_ANNOTATION_MIDOVERLAP 5

/* counter(segments, annotations, workspace) for one contig, as a double (exact for the
 * integer counters).  */
double gato_counter(int counter_id, const gato_segment* segs, size_t nsegs,
                    const gato_segment* annos, size_t nannos, int64_t ws_nseg);

/* Stream modes (the reference has only mode 0; mode 1 is this build's parallel contract):
 *  mode 0 "global-serial": one MT19937 stream seeded once with `seed`
 *         (scripts/gat-run.py:267-271), consumed in (sample, unit) order;
 *         sample_begin must be 0 for that to mean anything.
 *  mode 1 "per-unit": before each sampler.sample() the stream is re-seeded with
 *         (seed + sample_id * n_units + unit) mod 2^32.
 * counts_out: [n_counters][n_tracks][n_samples] 8-byte slots; int64 for the integer
 * counters, double for nucleotide-density.
 * samples_out (nullable): sampled contig-level segments, appended per (sample, contig);
 * samples_off has n_samples*n_contigs+1 entries. */
int gato_run_samples(const gato_problem* p, const int32_t* counter_ids, int n_counters,
                     uint32_t seed, int stream_mode,
                     int64_t sample_begin, int64_t sample_end,
                     void* counts_out,
                     gato_segment* samples_out, int64_t samples_cap, int64_t* samples_off);

/* ---- statistics: gat/Engine.pyx:1543-1576 getTwoSidedPValue ---------------------------
 * sorted: samples sorted ascending; expected: numpy.mean(samples) computed by the caller. */
double gato_two_sided_pvalue(const double* sorted, long n, double expected, double val);

#ifdef __cplusplus
}
#endif
#endif
