/*
 * gat_oracle.c -- CPU restatement of the GAT sampling + overlap-counting hot path.
 *
 * TEST INFRASTRUCTURE ONLY (see gat_oracle.h).  Plain C, no dependencies beyond libc.
 * Each function follows the reference function cited above it; integer widths and
 * casts (uint32 Position, int32 PositionDifference, C long) are kept as in the Cython.
 */
#include "gat_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

typedef uint32_t Position;            /* gat/SegmentList.pxd:31 */
typedef int32_t PositionDifference;   /* gat/SegmentList.pxd:33 */
typedef gato_segment Segment;

/* gat/SegmentList.pyx:68-77 lmin/lmax: take and return PositionDifference (int32) */
static inline PositionDifference lmin(PositionDifference a, PositionDifference b) { return a < b ? a : b; }
static inline PositionDifference lmax(PositionDifference a, PositionDifference b) { return a > b ? a : b; }

/* gat/SegmentList.pyx:99-101 segment_overlap_raw */
static inline PositionDifference segment_overlap_raw(Segment a, Segment b) {
  return (PositionDifference)lmin((PositionDifference)a.end, (PositionDifference)b.end) -
         (PositionDifference)lmax((PositionDifference)a.start, (PositionDifference)b.start);
}
/* gat/SegmentList.pyx:93-97 range_overlap */
static inline PositionDifference range_overlap(Position astart, Position aend, Position bstart, Position bend) {
  return lmax(0, (PositionDifference)lmin((PositionDifference)aend, (PositionDifference)bend) -
                     (PositionDifference)lmax((PositionDifference)astart, (PositionDifference)bstart));
}
/* gat/SegmentList.pyx:104-105 segment_length */
static inline PositionDifference segment_length(Segment a) {
  return (PositionDifference)a.end - (PositionDifference)a.start;
}

/* ---------------------------------------------------------------------------------------
 * utils/gat_utils.c:36-60 searchsorted: leftmost i with compar(base[i], target) >= 0.
 * cmpPosition (gat/Engine.pyx:119-120) returns (unsigned)a - (unsigned)b converted to int,
 * i.e. the sign of the wrapped 32-bit difference.
 */
long gato_searchsorted_u32(const uint32_t* base, size_t n, uint32_t target) {
  size_t imin = 0, imax = n;
  while (imin < imax) {
    size_t imid = imin + ((imax - imin) >> 1);
    if ((int)(base[imid] - target) < 0) imin = imid + 1; else imax = imid;
  }
  return (long)imin;
}
/* cmpSegments (gat/SegmentList.pyx:119-121): int32(start1) - int32(start2) */
long gato_searchsorted_seg(const gato_segment* base, size_t n, uint32_t start) {
  size_t imin = 0, imax = n;
  while (imin < imax) {
    size_t imid = imin + ((imax - imin) >> 1);
    if ((int)((PositionDifference)base[imid].start - (PositionDifference)start) < 0) imin = imid + 1; else imax = imid;
  }
  return (long)imin;
}

static int cmp_segments(const void* s1, const void* s2) {   /* gat/SegmentList.pyx:119-121 */
  return (PositionDifference)((const Segment*)s1)->start - (PositionDifference)((const Segment*)s2)->start;
}

/* gat/SegmentList.pyx:478-486 sort: libc qsort by start only (tie order unspecified;
 * normalize/merge results do not depend on it). */
static void seg_sort(Segment* s, size_t n) {
  if (n == 0) return;
  qsort(s, n, sizeof(Segment), cmp_segments);
}

/* gat/SegmentList.pyx:697-754 normalize: adjacent segments are NOT merged */
size_t gato_normalize(gato_segment* seg, size_t nsegments) {
  long idx, insertion_idx;
  Position max_end;
  long n = (long)nsegments;
  if (n == 0) return 0;
  seg_sort(seg, nsegments);
  insertion_idx = 0;
  idx = 0;
  while (idx < n && seg[idx].start == seg[idx].end) idx++;
  if (idx == n) return 0;
  seg[insertion_idx].start = seg[idx].start;
  max_end = seg[idx].end;
  while (idx < n) {
    if (seg[idx].start == seg[idx].end) { idx++; continue; }
    if (seg[idx].start >= max_end) {
      seg[insertion_idx].end = max_end;
      insertion_idx++;
      seg[insertion_idx].start = seg[idx].start;
    }
    max_end = (Position)lmax((PositionDifference)seg[idx].end, (PositionDifference)max_end);
    idx++;
  }
  seg[insertion_idx].end = max_end;
  insertion_idx++;
  return (size_t)insertion_idx;
}

/* gat/SegmentList.pyx:756-816 merge(distance): distance 0 merges adjacent segments */
size_t gato_merge(gato_segment* seg, size_t nsegments, int32_t distance) {
  PositionDifference max_end;
  long idx, insertion_idx;
  long n = (long)nsegments;
  if (n == 0) return 0;
  seg_sort(seg, nsegments);
  insertion_idx = 0;
  idx = 0;
  while (idx < n && seg[idx].start == seg[idx].end) idx++;
  if (idx == n) return 0;
  seg[insertion_idx].start = seg[idx].start;
  max_end = (PositionDifference)seg[idx].end;
  while (idx < n) {
    if (seg[idx].start == seg[idx].end) { idx++; continue; }
    if ((PositionDifference)seg[idx].start - distance > max_end) {
      seg[insertion_idx].end = (Position)max_end;
      insertion_idx++;
      seg[insertion_idx].start = seg[idx].start;
    }
    max_end = lmax((PositionDifference)seg[idx].end, max_end);
    idx++;
  }
  seg[insertion_idx].end = (Position)max_end;
  insertion_idx++;
  return (size_t)insertion_idx;
}

/* gat/SegmentList.pyx:818-851 check */
int gato_check(const gato_segment* s, size_t n) {
  size_t idx;
  if (n == 0) return 1;
  if (s[0].start >= s[0].end) return 0;
  for (idx = 1; idx < n; idx++) {
    if (s[idx].start >= s[idx].end) return 0;
    if (s[idx - 1].start > s[idx].start) return 0;
    if (s[idx - 1].end > s[idx].start) return 0;
  }
  return 1;
}

/* gat/SegmentList.pyx:1401-1467 filter: keep whole segments of a that touch b */
size_t gato_filter(gato_segment* a, size_t na, const gato_segment* b, size_t nb) {
  long working_idx = 0, this_idx = 0, other_idx = 0, last_this_idx = -1, last_other_idx = -1;
  Segment this_segment = {0, 0}, other_segment = {0, 0};
  Position last_start;
  Segment* out;
  if ((const gato_segment*)a == b) return 0;      /* :1408-1410 self-self: clear */
  if (na == 0) return 0;
  out = (Segment*)malloc(na * sizeof(Segment));
  last_start = a[0].start - 1;                     /* :1429, uint32 wrap */
  while (this_idx < (long)na && other_idx < (long)nb) {
    if (last_this_idx != this_idx) { this_segment = a[this_idx]; last_this_idx = this_idx; }
    if (last_other_idx != other_idx) { other_segment = b[other_idx]; last_other_idx = other_idx; }
    if (this_segment.end <= other_segment.start) this_idx++;
    else if (other_segment.end <= this_segment.start) other_idx++;
    else {
      if (last_start != this_segment.start) {
        out[working_idx] = this_segment;
        working_idx++;
        last_start = this_segment.start;
      }
      if (this_segment.end < other_segment.end) this_idx++;
      else if (other_segment.end < this_segment.end) other_idx++;
      else { this_idx++; other_idx++; }
    }
  }
  memcpy(a, out, (size_t)working_idx * sizeof(Segment));
  free(out);
  return (size_t)working_idx;
}

/* gat/SegmentList.pyx:1469-1549 intersect: one output piece per overlapping pair */
long gato_intersect(const gato_segment* a, size_t na, const gato_segment* b, size_t nb,
                    gato_segment* out, size_t cap) {
  long working_idx = 0, this_idx = 0, other_idx = 0, last_this_idx = -1, last_other_idx = -1;
  Segment this_segment = {0, 0}, other_segment = {0, 0};
  if (na == 0) return 0;
  while (this_idx < (long)na && other_idx < (long)nb) {
    if (last_this_idx != this_idx) { this_segment = a[this_idx]; last_this_idx = this_idx; }
    if (last_other_idx != other_idx) { other_segment = b[other_idx]; last_other_idx = other_idx; }
    if (this_segment.end <= other_segment.start) this_idx++;
    else if (other_segment.end <= this_segment.start) other_idx++;
    else {
      if ((size_t)working_idx >= cap) return GATO_ERR_CAPACITY;
      out[working_idx].start = (Position)lmax((PositionDifference)this_segment.start, (PositionDifference)other_segment.start);
      out[working_idx].end = (Position)lmin((PositionDifference)this_segment.end, (PositionDifference)other_segment.end);
      working_idx++;
      if (this_segment.end < other_segment.end) this_idx++;
      else if (other_segment.end < this_segment.end) other_idx++;
      else { this_idx++; other_idx++; }
    }
  }
  return working_idx;
}

/* gat/SegmentList.pyx:1607-1616 sum: uint32 accumulate */
uint32_t gato_sum(const gato_segment* s, size_t n) {
  Position total = 0;
  size_t idx;
  for (idx = 0; idx < n; idx++) total += s[idx].end - s[idx].start;
  return total;
}

/* gat/SegmentList.pyx:1026-1076 overlapWithSegments */
uint32_t gato_overlap_with_segments(const gato_segment* a, size_t na, const gato_segment* b, size_t nb) {
  long this_idx = 0, other_idx = 0, last_this_idx = -1, last_other_idx = -1;
  Segment this_segment = {0, 0}, other_segment = {0, 0};
  Position overlap = 0;
  if (a == b) return gato_sum(a, na);             /* :1036-1037 same-buffer shortcut */
  while (this_idx < (long)na && other_idx < (long)nb) {
    if (last_this_idx != this_idx) { this_segment = a[this_idx]; last_this_idx = this_idx; }
    if (last_other_idx != other_idx) { other_segment = b[other_idx]; last_other_idx = other_idx; }
    if (this_segment.end <= other_segment.start) this_idx++;
    else if (other_segment.end <= this_segment.start) other_idx++;
    else {
      overlap += (Position)segment_overlap_raw(this_segment, other_segment);
      if (this_segment.end < other_segment.end) this_idx++;
      else if (other_segment.end < this_segment.end) other_idx++;
      else { this_idx++; other_idx++; }
    }
  }
  return overlap;
}

/* gat/SegmentList.pyx:1078-1146 intersectionWithSegments: number of segments of a hit by b
 * (mode "midpoint": whose midpoint lies in the b segment currently compared).  Only this_idx
 * advances after a hit (:1144). */
uint32_t gato_intersection_with_segments(const gato_segment* a, size_t na, const gato_segment* b, size_t nb,
                                         int midpoint_overlap) {
  long this_idx = 0, other_idx = 0, last_this_idx = -1, last_other_idx = -1;
  Segment this_segment = {0, 0}, other_segment = {0, 0};
  Position noverlap = 0;
  if (a == b) return gato_sum(a, na);             /* :1106 (returns bases: reference quirk) */
  while (this_idx < (long)na && other_idx < (long)nb) {
    if (last_this_idx != this_idx) { this_segment = a[this_idx]; last_this_idx = this_idx; }
    if (last_other_idx != other_idx) { other_segment = b[other_idx]; last_other_idx = other_idx; }
    if (this_segment.end <= other_segment.start) this_idx++;
    else if (other_segment.end <= this_segment.start) other_idx++;
    else {
      if (midpoint_overlap) {
        Position mid = this_segment.start + (this_segment.end - this_segment.start) / 2;
        if (other_segment.start <= mid && mid < other_segment.end) noverlap++;
      } else {
        noverlap++;
      }
      this_idx++;
    }
  }
  return noverlap;
}

/* gat/SegmentList.pyx:853-887 _getInsertionPoint */
int gato_get_insertion_point(const gato_segment* s, size_t n, uint32_t start, uint32_t end) {
  int idx;
  if (n == 0) return -1;
  if (start >= s[n - 1].end) return (int)n;
  if (end <= s[0].start) return -1;
  idx = (int)gato_searchsorted_seg(s, n, start);
  if (idx == (int)n) return idx - 1;
  else if (s[idx].start != start) return idx - 1;
  else return idx;
}

/* gat/SegmentList.pyx:545-597 trim_ends */
int gato_trim_ends(gato_segment* seg, size_t n, uint32_t pos, uint32_t size, int forward) {
  int idx;
  Position l;
  PositionDifference s = (PositionDifference)size;
  Segment sg;
  if (n == 0) return GATO_OK;
  /* assert self.sum() > s (:560): Position vs PositionDifference compares as Python ints */
  if (!((int64_t)gato_sum(seg, n) > (int64_t)s)) return GATO_ERR_ASSERT;
  idx = gato_get_insertion_point(seg, n, pos, pos + 1);
  if (idx == (int)n) idx = 0;
  if (idx < 0) idx = (int)n - 1;
  if (forward) {
    while (s > 0) {
      sg = seg[idx];
      l = (Position)segment_length(sg);
      if (segment_length(sg) < s) { seg[idx].start = 0; seg[idx].end = 0; s -= (PositionDifference)l; }
      else { seg[idx].start = sg.start + (Position)s; seg[idx].end = sg.end; s = 0; }
      idx++;
      if (idx == (int)n) idx = 0;
    }
  } else {
    while (s > 0) {
      sg = seg[idx];
      l = (Position)segment_length(sg);
      if (segment_length(sg) < s) { seg[idx].start = 0; seg[idx].end = 0; s -= (PositionDifference)l; }
      else { seg[idx].start = sg.start; seg[idx].end = (Position)((PositionDifference)sg.end - s); s = 0; }
      idx--;
      if (idx < 0) idx = (int)n - 1;
    }
  }
  return GATO_OK;
}

/* gat/SegmentList.pyx:1148-1184 getLengthDistribution (+ largest() :1618-1635).
 * bucket_size==0: int(math.ceil(len(largest) / float(nbuckets))).
 * bucket index i = <int>((l + bucket_size - 1) / bucket_size): Python-object true division
 * followed by C truncation; identical to integer floor division for operands < 2^32. */
int gato_length_distribution(const gato_segment* s, size_t n, uint32_t bucket_size_in, int nbuckets,
                             int64_t* hist, uint32_t* bucket_size_out) {
  size_t idx;
  int64_t bucket_size = bucket_size_in;
  memset(hist, 0, sizeof(int64_t) * (size_t)nbuckets);
  if (bucket_size == 0) {
    Position max_value = 0;
    if (n == 0) return GATO_ERR_VALUE;             /* largest() raises ValueError on empty */
    for (idx = 0; idx < n; idx++) {
      Position l = (Position)segment_length(s[idx]);
      if (l > max_value) max_value = l;
    }
    /* segment_length(largest) is a C int; / float(nbuckets) */
    bucket_size = (int64_t)ceil((double)(PositionDifference)max_value / (double)nbuckets);
  }
  for (idx = 0; idx < n; idx++) {
    Position l = (Position)segment_length(s[idx]);
    int i;
    if (bucket_size == 0) return GATO_ERR_VALUE;   /* ZeroDivisionError in the reference */
    i = (int)(((int64_t)l + bucket_size - 1) / bucket_size);
    if (i >= nbuckets) return GATO_ERR_VALUE;
    hist[i] += 1;
  }
  *bucket_size_out = (uint32_t)bucket_size;
  return GATO_OK;
}

/* ---------------------------------------------------------------------------------------
 * numpy legacy RandomState: MT19937.  numpy/random/src/mt19937/mt19937.c
 *   mt19937_seed  (init_genrand, Knuth multiplier 1812433253)
 *   mt19937_gen + mt19937_next (genrand_int32 with tempering)
 */
#define MT_N 624
#define MT_M 397
void gato_rng_seed(gato_rng* r, uint32_t seed) {
  int pos;
  seed &= 0xffffffffu;
  for (pos = 0; pos < MT_N; pos++) {
    r->mt[pos] = seed;
    seed = (1812433253u * (seed ^ (seed >> 30)) + (uint32_t)pos + 1) & 0xffffffffu;
  }
  r->mti = MT_N;
  r->ndraws = 0;
}
static void mt_gen(gato_rng* r) {
  uint32_t y;
  int i;
  for (i = 0; i < MT_N - MT_M; i++) {
    y = (r->mt[i] & 0x80000000u) | (r->mt[i + 1] & 0x7fffffffu);
    r->mt[i] = r->mt[i + MT_M] ^ (y >> 1) ^ (-(int32_t)(y & 1) & 0x9908b0dfu);
  }
  for (; i < MT_N - 1; i++) {
    y = (r->mt[i] & 0x80000000u) | (r->mt[i + 1] & 0x7fffffffu);
    r->mt[i] = r->mt[i + (MT_M - MT_N)] ^ (y >> 1) ^ (-(int32_t)(y & 1) & 0x9908b0dfu);
  }
  y = (r->mt[MT_N - 1] & 0x80000000u) | (r->mt[0] & 0x7fffffffu);
  r->mt[MT_N - 1] = r->mt[MT_M - 1] ^ (y >> 1) ^ (-(int32_t)(y & 1) & 0x9908b0dfu);
  r->mti = 0;
}
uint32_t gato_rng_u32(gato_rng* r) {
  uint32_t y;
  if (r->mti == MT_N) mt_gen(r);
  y = r->mt[r->mti++];
  y ^= (y >> 11);
  y ^= (y << 7) & 0x9d2c5680u;
  y ^= (y << 15) & 0xefc60000u;
  y ^= (y >> 18);
  r->ndraws++;
  return y;
}
/* numpy.random.randint(lo, hi) on the legacy RandomState, default dtype (int64):
 * _rand_int64(lo, hi-1) -> random_bounded_uint64(off=lo, rng=hi-1-lo, mask, use_masked=True)
 * (numpy/random/src/distributions/distributions.c): rng==0 -> off (no draw);
 * rng<=0xFFFFFFFF -> 32-bit generator: rng==0xFFFFFFFF -> one raw draw, else masked rejection;
 * otherwise 64-bit masked rejection with next_uint64 = (hi word << 32) | lo word. */
int64_t gato_randint(gato_rng* r, int64_t lo, int64_t hi) {
  uint64_t rng = (uint64_t)(hi - 1 - lo);
  uint64_t mask = rng;
  mask |= mask >> 1; mask |= mask >> 2; mask |= mask >> 4;
  mask |= mask >> 8; mask |= mask >> 16; mask |= mask >> 32;
  if (rng == 0) return lo;
  if (rng <= 0xFFFFFFFFull) {
    uint32_t val;
    if (rng == 0xFFFFFFFFull) return lo + (int64_t)gato_rng_u32(r);
    do { val = gato_rng_u32(r) & (uint32_t)mask; } while (val > (uint32_t)rng);
    return lo + (int64_t)val;
  } else {
    uint64_t val;
    do {
      uint64_t upper = (uint64_t)gato_rng_u32(r) << 32;
      uint64_t lower = gato_rng_u32(r);
      val = (upper | lower) & mask;
    } while (val > rng);
    return lo + (int64_t)val;
  }
}

/* ---------------------------------------------------------------------------------------
 * gat/Engine.pyx:387-435 HistogramSampler
 */
typedef struct { Position* cdf; Position bucket_size; Position nbuckets; Position total_size; } hist_sampler;
static int hist_sampler_init(hist_sampler* hs, const int64_t* histogram, int nbuckets, Position bucket_size) {
  Position i;
  hs->nbuckets = (Position)nbuckets;
  hs->cdf = (Position*)malloc(sizeof(Position) * (size_t)nbuckets);
  if (!hs->cdf) return GATO_ERR_MEMORY;
  hs->total_size = 0;
  for (i = 0; i < hs->nbuckets; i++) { hs->total_size += (Position)histogram[i]; hs->cdf[i] = hs->total_size; }
  hs->bucket_size = bucket_size;
  return GATO_OK;
}
static Position hist_sampler_sample(hist_sampler* hs, gato_rng* rng) {   /* :413-435 */
  Position base, index, r;
  if (hs->total_size > 1) r = (Position)gato_randint(rng, 1, (int64_t)hs->total_size);
  else r = 1;
  index = (Position)gato_searchsorted_u32(hs->cdf, hs->nbuckets, r);
  base = index * hs->bucket_size;
  if (hs->bucket_size > 1) return base + (Position)gato_randint(rng, 0, (int64_t)hs->bucket_size);
  return base;
}

/* gat/Engine.pyx:245-348 SegmentListSampler */
typedef struct { const Segment* segments; Position* cdf; Position total_size; int nsegments; } sl_sampler;
static int sl_sampler_init(sl_sampler* s, const Segment* segments, size_t n) {    /* :261-277 */
  size_t i;
  if (n == 0) return GATO_ERR_ASSERT;              /* "sampling from empty segment list" */
  s->segments = segments;
  s->nsegments = (int)n;
  s->cdf = (Position*)malloc(sizeof(Position) * n);
  if (!s->cdf) return GATO_ERR_MEMORY;
  s->total_size = 0;
  for (i = 0; i < n; i++) {
    s->total_size += (Position)segment_length(segments[i]);
    s->cdf[i] = s->total_size - 1;
  }
  return GATO_OK;
}
static int sl_sampler_sample(sl_sampler* s, gato_rng* rng, Position sample_length,
                             Position* start_out, Position* end_out, PositionDifference* overlap_out) { /* :279-343 */
  Position start, end;
  PositionDifference overlap;
  size_t segment_index;
  Segment chosen_segment;
  Position random_pos_in_workspace;
  long random_pos_in_segment, sampling_start;
  random_pos_in_workspace = (Position)gato_randint(rng, 0, (int64_t)s->total_size);
  segment_index = (size_t)gato_searchsorted_u32(s->cdf, (size_t)s->nsegments, random_pos_in_workspace);
  if (!(segment_index < (size_t)s->nsegments)) return GATO_ERR_ASSERT;
  chosen_segment = s->segments[segment_index];
  sampling_start = (long)chosen_segment.start - (long)sample_length + 1;
  if (segment_index > 0)
    sampling_start = lmax((PositionDifference)s->segments[segment_index - 1].end, (PositionDifference)sampling_start);
  random_pos_in_segment = (long)gato_randint(rng, (int64_t)sampling_start, (int64_t)chosen_segment.end);
  start = (Position)lmax(0, (PositionDifference)random_pos_in_segment);
  end = (Position)(random_pos_in_segment + (long)sample_length);
  overlap = range_overlap(chosen_segment.start, chosen_segment.end, start, end);
  if (!(overlap > 0)) return GATO_ERR_ASSERT;
  *start_out = start; *end_out = end; *overlap_out = overlap;
  return GATO_OK;
}

/* growable segment vector (SegmentList._add/extend, gat/SegmentList.pyx:488-543) */
typedef struct { Segment* v; size_t n, cap; } segvec;
static int segvec_reserve(segvec* s, size_t cap) {
  if (cap <= s->cap) return GATO_OK;
  if (cap < 2 * s->cap) cap = 2 * s->cap;
  if (cap < 64) cap = 64;
  {
    Segment* nv = (Segment*)realloc(s->v, cap * sizeof(Segment));
    if (!nv) return GATO_ERR_MEMORY;
    s->v = nv; s->cap = cap;
  }
  return GATO_OK;
}
static int segvec_add(segvec* s, Segment x) {
  int rc = segvec_reserve(s, s->n + 1);
  if (rc) return rc;
  s->v[s->n++] = x;
  return GATO_OK;
}

/* gat/Engine.pyx:515-646 SamplerAnnotator.sample */
int gato_sampler_annotator(gato_rng* rng, const gato_segment* segs, size_t nsegs,
                           const gato_segment* ws, size_t nws,
                           uint32_t bucket_size_cfg, int nbuckets,
                           gato_segment* out, size_t out_cap, size_t* nout, int* nunsuccessful_out) {
  PositionDifference remaining, true_remaining, overlap, ltotal, length;
  int nunsuccessful_rounds, max_unsuccessful_rounds, rc = GATO_OK;
  Position start, end, bucket_size = 0;
  segvec sampled = {0, 0, 0}, unintersected = {0, 0, 0}, intersected = {0, 0, 0};
  Segment* working = NULL;
  size_t nworking;
  int64_t* histogram = NULL;
  hist_sampler hs = {0, 0, 0, 0};
  sl_sampler sls = {0, 0, 0, 0};
  long ni;

  *nout = 0;
  if (nunsuccessful_out) *nunsuccessful_out = 0;
  if (!gato_check(segs, nsegs) || !gato_check(ws, nws)) return GATO_ERR_ASSERT;   /* :535-536 */

  /* :543-546 working = segments.clone().filter(workspace) */
  working = (Segment*)malloc((nsegs ? nsegs : 1) * sizeof(Segment));
  memcpy(working, segs, nsegs * sizeof(Segment));
  nworking = gato_filter(working, nsegs, ws, nws);
  if (nworking == 0) { free(working); return GATO_OK; }

  /* :550-552 ltotal = working.clone().intersect(workspace).sum() */
  if ((rc = segvec_reserve(&intersected, nworking + nws + 16))) goto done;
  ni = gato_intersect(working, nworking, ws, nws, intersected.v, intersected.cap);
  if (ni < 0) { rc = (int)ni; goto done; }
  ltotal = (PositionDifference)gato_sum(intersected.v, (size_t)ni);

  /* :559-562 */
  histogram = (int64_t*)malloc(sizeof(int64_t) * (size_t)nbuckets);
  if ((rc = gato_length_distribution(working, nworking, bucket_size_cfg, nbuckets, histogram, &bucket_size))) goto done;
  if ((rc = hist_sampler_init(&hs, histogram, nbuckets, bucket_size))) goto done;
  /* :565 */
  if ((rc = sl_sampler_init(&sls, ws, nws))) goto done;

  remaining = ltotal;
  true_remaining = remaining;
  nunsuccessful_rounds = 0;
  max_unsuccessful_rounds = 20;

  while (true_remaining > 0 && nunsuccessful_rounds < max_unsuccessful_rounds) {   /* :572 */
    length = (PositionDifference)hist_sampler_sample(&hs, rng);                    /* :576 */
    if (!(length > 0)) { rc = GATO_ERR_ASSERT; goto done; }

    if (remaining <= length) {                                                      /* :582 */
      if ((rc = segvec_reserve(&unintersected, unintersected.n + sampled.n + 1))) goto done;
      memcpy(unintersected.v + unintersected.n, sampled.v, sampled.n * sizeof(Segment));
      unintersected.n += sampled.n;
      unintersected.n = gato_merge(unintersected.v, unintersected.n, 0);
      sampled.n = 0;
      if ((rc = segvec_reserve(&intersected, unintersected.n + nws + 16))) goto done;
      ni = gato_intersect(unintersected.v, unintersected.n, ws, nws, intersected.v, intersected.cap);
      if (ni < 0) { rc = (int)ni; goto done; }
      remaining = ltotal - (PositionDifference)gato_sum(intersected.v, (size_t)ni);
      if (true_remaining == remaining) nunsuccessful_rounds++;
      else true_remaining = remaining;
    }

    if (true_remaining < 0) {                                                       /* :608 */
      sl_sampler temp = {0, 0, 0, 0};
      int forward;
      if ((rc = sl_sampler_init(&temp, unintersected.v, unintersected.n))) goto done;
      rc = sl_sampler_sample(&temp, rng, 1, &start, &end, &overlap);
      free(temp.cdf);
      if (rc) goto done;
      forward = (int)gato_randint(rng, 0, 2);
      if ((rc = gato_trim_ends(unintersected.v, unintersected.n, start, (Position)(-true_remaining), forward))) goto done;
      true_remaining = 1;
      continue;
    }

    if ((rc = sl_sampler_sample(&sls, rng, (Position)length, &start, &end, &overlap))) goto done;   /* :628 */
    if (true_remaining > 0) {
      Segment sg; sg.start = start; sg.end = end;
      if ((rc = segvec_add(&sampled, sg))) goto done;
      remaining -= overlap;
    }
  }
  if (nunsuccessful_out) *nunsuccessful_out = nunsuccessful_rounds;

  /* :639-646 */
  unintersected.n = gato_merge(unintersected.v, unintersected.n, 0);
  unintersected.n = gato_filter(unintersected.v, unintersected.n, ws, nws);
  if (!(gato_sum(unintersected.v, unintersected.n) > 0)) { rc = GATO_ERR_ASSERT; goto done; }
  if (unintersected.n > out_cap) { rc = GATO_ERR_CAPACITY; goto done; }
  memcpy(out, unintersected.v, unintersected.n * sizeof(Segment));
  *nout = unintersected.n;

done:
  free(working); free(histogram); free(hs.cdf); free(sls.cdf);
  free(sampled.v); free(unintersected.v); free(intersected.v);
  return rc;
}

/* gat/Engine.pyx:695-737 SamplerSegments.sample */
int gato_sampler_segments(gato_rng* rng, const gato_segment* segs, size_t nsegs,
                          const gato_segment* ws, size_t nws, uint32_t bucket_size_cfg, int nbuckets,
                          gato_segment* out, size_t out_cap, size_t* nout) {
  Segment* working = NULL;
  size_t nworking, x;
  int64_t* histogram = NULL;
  hist_sampler hs = {0, 0, 0, 0};
  sl_sampler sls = {0, 0, 0, 0};
  Position bucket_size = 0, start, end;
  PositionDifference overlap;
  int rc = GATO_OK;
  *nout = 0;
  if (!gato_check(ws, nws)) return GATO_ERR_ASSERT;            /* :706 */
  working = (Segment*)malloc((nsegs ? nsegs : 1) * sizeof(Segment));
  memcpy(working, segs, nsegs * sizeof(Segment));
  nworking = gato_filter(working, nsegs, ws, nws);              /* :711-712 */
  if (nworking == 0) { free(working); return GATO_OK; }
  histogram = (int64_t*)malloc(sizeof(int64_t) * (size_t)nbuckets);
  if ((rc = gato_length_distribution(working, nworking, bucket_size_cfg, nbuckets, histogram, &bucket_size))) goto done;
  if ((rc = hist_sampler_init(&hs, histogram, nbuckets, bucket_size))) goto done;
  if ((rc = sl_sampler_init(&sls, ws, nws))) goto done;
  if (nsegs > out_cap) { rc = GATO_ERR_CAPACITY; goto done; }
  for (x = 0; x < nsegs; x++) {                                 /* :726 for x in xrange(len(segments)) */
    Position length = hist_sampler_sample(&hs, rng);
    if (!(length > 0)) { rc = GATO_ERR_ASSERT; goto done; }
    if ((rc = sl_sampler_sample(&sls, rng, length, &start, &end, &overlap))) goto done;
    out[x].start = start; out[x].end = end;
  }
  *nout = nsegs;
done:
  free(working); free(histogram); free(hs.cdf); free(sls.cdf);
  return rc;
}

/* gat/Engine.pyx:1417-1472 Counter*.__call__ for one contig */
double gato_counter(int counter_id, const gato_segment* segs, size_t nsegs,
                    const gato_segment* annos, size_t nannos, int64_t ws_nseg) {
  switch (counter_id) {
    case GATO_COUNTER_NUCLEOTIDE_OVERLAP:
      return (double)gato_overlap_with_segments(annos, nannos, segs, nsegs);
    case GATO_COUNTER_NUCLEOTIDE_DENSITY: {
      Position l = (Position)ws_nseg;                     /* cdef Position l = len(workspace) */
      if (l == 0) return 0;
      return (double)gato_overlap_with_segments(annos, nannos, segs, nsegs) / (double)l;
    }
    case GATO_COUNTER_SEGMENT_OVERLAP:
      return (double)gato_intersection_with_segments(segs, nsegs, annos, nannos, 0);
    case GATO_COUNTER_SEGMENT_MIDOVERLAP:
      return (double)gato_intersection_with_segments(segs, nsegs, annos, nannos, 1);
    case GATO_COUNTER_ANNOTATION_OVERLAP:
      return (double)gato_intersection_with_segments(annos, nannos, segs, nsegs, 0);
    case GATO_COUNTER_ANNOTATION_MIDOVERLAP:
      return (double)gato_intersection_with_segments(annos, nannos, segs, nsegs, 1);
  }
  return NAN;
}

/* gat/__init__.py:494-591 computeSample for samples [sample_begin, sample_end);
 * fromIsochores: gat/Engine.pyx:2857-2876. */
static int run_samples_rng(const gato_problem* p, const int32_t* counter_ids, int n_counters,
                           gato_rng* rng, uint32_t seed, int stream_mode,
                           int64_t sample_begin, int64_t sample_end, void* counts_out,
                           gato_segment* samples_out, int64_t samples_cap, int64_t* samples_off) {
  int64_t n_samples = sample_end - sample_begin;
  int64_t s;
  int rc = GATO_OK;
  int u, c, k, a;
  size_t max_unit = 0, total = 0, cap;
  Segment* unit_out = NULL;
  segvec* contig = NULL;
  int64_t samples_n = 0;

  for (u = 0; u < p->n_units; u++) {
    size_t n = (size_t)(p->seg_off[u + 1] - p->seg_off[u]);
    if (n > max_unit) max_unit = n;
    total += n;
  }
  cap = 4 * max_unit + 1024;
  unit_out = (Segment*)malloc(cap * sizeof(Segment));
  contig = (segvec*)calloc((size_t)(p->n_contigs > 0 ? p->n_contigs : 1), sizeof(segvec));
  if (samples_off) samples_off[0] = 0;

  for (s = sample_begin; s < sample_end; s++) {
    for (c = 0; c < p->n_contigs; c++) contig[c].n = 0;
    for (u = 0; u < p->n_units; u++) {
      const Segment* us = p->segs + p->seg_off[u];
      size_t nus = (size_t)(p->seg_off[u + 1] - p->seg_off[u]);
      const Segment* uw = p->ws + p->ws_off[u];
      size_t nuw = (size_t)(p->ws_off[u + 1] - p->ws_off[u]);
      size_t nout = 0;
      if (nuw == 0 || nus == 0) continue;            /* gat/__init__.py:536-538, no RNG use */
      if (stream_mode == 1)
        gato_rng_seed(rng, (uint32_t)(((uint64_t)seed + (uint64_t)s * (uint64_t)p->n_units + (uint64_t)u) & 0xffffffffull));
      for (;;) {
        gato_rng save = *rng;
        if (p->sampler == 1) rc = gato_sampler_segments(rng, us, nus, uw, nuw, p->bucket_size, p->nbuckets, unit_out, cap, &nout);
        else rc = gato_sampler_annotator(rng, us, nus, uw, nuw, p->bucket_size, p->nbuckets, unit_out, cap, &nout, NULL);
        if (rc == GATO_ERR_CAPACITY) {               /* grow and redo this unit from the saved stream */
          *rng = save;
          cap *= 2;
          free(unit_out);
          unit_out = (Segment*)malloc(cap * sizeof(Segment));
          continue;
        }
        break;
      }
      if (rc) goto done;
      c = p->unit_contig[u];
      if ((rc = segvec_reserve(&contig[c], contig[c].n + nout + 1))) goto done;
      memcpy(contig[c].v + contig[c].n, unit_out, nout * sizeof(Segment));   /* new[contig].extend */
      contig[c].n += nout;
    }
    if (p->merge_contigs)
      for (c = 0; c < p->n_contigs; c++) contig[c].n = gato_merge(contig[c].v, contig[c].n, 0);

    if (samples_off) {
      for (c = 0; c < p->n_contigs; c++) {
        if (samples_out) {
          if (samples_n + (int64_t)contig[c].n > samples_cap) { rc = GATO_ERR_CAPACITY; goto done; }
          memcpy(samples_out + samples_n, contig[c].v, contig[c].n * sizeof(Segment));
        }
        samples_n += (int64_t)contig[c].n;
        samples_off[(s - sample_begin) * p->n_contigs + c + 1] = samples_n;
      }
    }

    /* counters assert isNormalized (gat/SegmentList.pyx:1031): raw SamplerSegments output is not, unless
     * fromIsochores merged it */
    if (n_counters > 0 && p->sampler == 1 && !p->merge_contigs) { rc = GATO_ERR_ASSERT; goto done; }
    /* gat/__init__.py:578-587: sum([...]) over contigs, Python ints exact / floats left-to-right */
    for (k = 0; k < n_counters; k++) {
      int cid = counter_ids[k];
      for (a = 0; a < p->n_tracks; a++) {
        int64_t slot = ((int64_t)k * p->n_tracks + a) * n_samples + (s - sample_begin);
        if (cid == GATO_COUNTER_NUCLEOTIDE_DENSITY) {
          double acc = 0.0;
          for (c = 0; c < p->n_contigs; c++) {
            int64_t o = p->anno_off[(int64_t)a * p->n_contigs + c];
            size_t na = (size_t)(p->anno_off[(int64_t)a * p->n_contigs + c + 1] - o);
            acc += gato_counter(cid, contig[c].v, contig[c].n, p->annos + o, na, p->cws_nseg[c]);
          }
          ((double*)counts_out)[slot] = acc;
        } else {
          int64_t acc = 0;
          for (c = 0; c < p->n_contigs; c++) {
            int64_t o = p->anno_off[(int64_t)a * p->n_contigs + c];
            size_t na = (size_t)(p->anno_off[(int64_t)a * p->n_contigs + c + 1] - o);
            acc += (int64_t)gato_counter(cid, contig[c].v, contig[c].n, p->annos + o, na, p->cws_nseg[c]);
          }
          ((int64_t*)counts_out)[slot] = acc;
        }
      }
    }
  }
done:
  free(unit_out);
  if (contig) { for (c = 0; c < p->n_contigs; c++) free(contig[c].v); free(contig); }
  (void)total;
  return rc;
}

int gato_run_samples(const gato_problem* p, const int32_t* counter_ids, int n_counters,
                     uint32_t seed, int stream_mode, int64_t sample_begin, int64_t sample_end,
                     void* counts_out, gato_segment* samples_out, int64_t samples_cap, int64_t* samples_off) {
  gato_rng rng;
  gato_rng_seed(&rng, seed);
  return run_samples_rng(p, counter_ids, n_counters, &rng, seed, stream_mode, sample_begin, sample_end,
                         counts_out, samples_out, samples_cap, samples_off);
}

/* gat/Engine.pyx:1543-1576 getTwoSidedPValue on the sorted sample values
 * (searchargsorted over sorted2sample == left bisect over the sorted values, cmpDouble :122-127) */
double gato_two_sided_pvalue(const double* sorted, long l, double expected, double val) {
  long imin = 0, imax = l, idx;
  double min_pval, pval;
  while (imin < imax) {
    long imid = imin + ((imax - imin) >> 1);
    double da = sorted[imid];
    if (((da > val) - (da < val)) < 0) imin = imid + 1; else imax = imid;
  }
  idx = imin;
  min_pval = 1.0 / (double)l;
  if (idx == l) idx = 1;
  else if (val > expected) {
    while (idx > 0 && sorted[idx] == val) idx--;
    idx = l - (idx + 1);
  } else {
    while (idx < l && sorted[idx] == val) idx++;
  }
  pval = (double)idx / (double)l;
  return min_pval > pval ? min_pval : pval;
}
