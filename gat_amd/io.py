"""Input/output around the hot path, mirroring gat/IO.py and the BED reader of gat/Engine.pyx:
BED tracks -> IntervalCollection (readFromBed, gat/Engine.pyx:2480-2556), buildSegments /
applyIsochores (gat/IO.py:88-293) and the result table (outputResults, gat/IO.py:457-539)."""
import collections
import glob
import gzip
import os
import re

import numpy as np

from . import engine, stats
from . import intervals as iv

_TRACK_RE = re.compile(r'([^\s=]+) *= *("[^"]*"|[^ ]*)')
# readFromBed: files from this size on go through the table parser (GAT_BED_TABLE_MIN_BYTES; GAT_BED_LINE_READER=1: never)
_BED_TABLE_MIN_BYTES = int(os.environ.get("GAT_BED_TABLE_MIN_BYTES", 8 << 20))


def openFile(filename, mode="r"):
    if filename.endswith(".gz"):
        return gzip.open(filename, mode + "t")
    return open(filename, mode)


def openOutputFile(section, options, mode="w"):
    """the side files of --output-stats / --output-bed: `section` substituted into --output-filename-pattern
    (gat/Experiment.py:554-582); "-" is the main output."""
    fn = re.sub("%s", section, getattr(options, "output_filename_pattern", None) or "%s")
    if fn == "-":
        return options.stdout
    if not getattr(options, "output_force", False) and os.path.exists(fn):
        raise OSError("file %s already exists, use --force to overwrite existing files." % fn)
    return openFile(fn, mode)


def _wanted(section, selected):
    return section in selected or "all" in selected or any(re.search(x, section) for x in selected)


def dumpStats(coll, section, options):
    """gat/IO.py:20-25."""
    if _wanted(section, getattr(options, "output_stats", None) or []):
        with openOutputFile(section, options) as f:
            coll.outputStats(f)


def dumpBed(coll, section, options):
    """gat/IO.py:28-32."""
    if _wanted(section, getattr(options, "output_bed", None) or []):
        with openOutputFile(section + ".bed", options) as f:
            coll.save(f)


def _bed_columns(filename):
    """the columns of a plain BED file in one pass of a C parser: (contigs, starts, ends, names or None), or None when the
    file needs the line-by-line reader below -- `track` lines, carriage returns, a line with fewer than three or more
    fields than the first, anything that is not an integer where a coordinate belongs: whatever could make the two readers
    differ.  (The line-by-line reader takes 0.28 s per 100 000 lines; the reference's own test workspace has 280 000.)"""
    import sys
    with (gzip.open(filename, "rb") if filename.endswith(".gz") else open(filename, "rb")) as f:
        data = f.read()
    # (importing the parser costs 0.4 s -- what the line reader needs for 150 000 lines: it pays from ~8 MB of text on)
    if len(data) < _BED_TABLE_MIN_BYTES and "pandas" not in sys.modules:
        return None
    try:
        import io as _io
        import pandas as pd
    except ImportError:
        return None
    if data.startswith(b"track") or b"\ntrack" in data or b"\r" in data or b"\x00" in data or b'"' in data:
        return None
    if data.startswith(b"#") or b"\n#" in data:
        data = re.sub(rb"(?m)^#[^\n]*\n?", b"", data)
    # the shape of the text, vectorised: every non-blank line the same number of fields (at least three), and no '.', 'e'
    # or 'E' inside a coordinate ("1.0" is an integer to the table parser, a ValueError to int())
    arr = np.frombuffer(data, dtype=np.uint8)
    if arr.size == 0:
        return np.empty(0, dtype=object), np.empty(0, dtype=np.int64), np.empty(0, dtype=np.int64), None
    starts = np.concatenate(([0], np.flatnonzero(arr == 10) + 1))
    starts = starts[starts < arr.size]
    ends = np.concatenate((starts[1:] - 1, [arr.size - (1 if arr[-1] == 10 else 0)]))     # (exclusive, without the newline)
    tabs = np.flatnonzero(arr == 9)
    before = np.searchsorted(tabs, starts)                                                  # tabs in front of each line
    at_end = np.searchsorted(tabs, ends)
    nonblank = ends > starts
    if not nonblank.any():
        return np.empty(0, dtype=object), np.empty(0, dtype=np.int64), np.empty(0, dtype=np.int64), None
    fields = (at_end - before)[nonblank] + 1
    ncol = int(fields[0])
    if ncol < 3 or (fields != ncol).any():
        return None
    suspects = np.flatnonzero((arr == 46) | (arr == 101) | (arr == 69))
    if suspects.size:
        line = np.searchsorted(starts, suspects, side="right") - 1
        col = np.searchsorted(tabs, suspects) - before[line]
        if ((col == 1) | (col == 2)).any():
            return None
    try:
        t = pd.read_csv(_io.BytesIO(data), sep="\t", header=None, names=list(range(ncol)), usecols=list(range(min(ncol, 4))),
                        dtype={0: str, 1: np.int64, 2: np.int64, 3: str}, quoting=3, keep_default_na=False, na_filter=False,
                        engine="c", skip_blank_lines=True)
    except Exception:
        return None
    chrom, start, end = t[0].to_numpy(dtype=object), t[1].to_numpy(), t[2].to_numpy()
    if start.dtype != np.int64 or end.dtype != np.int64 or (start > end).any():
        return None
    if pd.isna(chrom).any():
        return None
    names = None
    if ncol >= 4:
        names = t[3].to_numpy(dtype=object)
        if pd.isna(names).any():                               # (a short line: its name is missing, not empty)
            return None
    return chrom, start, end, names


def _bed_lines(filename, default_name, ignore_tracks):
    """one file line by line, gat/Engine.pyx:2480-2556: yields (track name, contig, start, end) in file order"""
    track = None
    with openFile(filename, "r") as infile:
        for line in infile:
            if line.startswith("#") or line.startswith("\n"):
                continue
            if line.startswith("track"):
                track = dict((k, v[1:-1] if v[:1] == '"' else v) for k, v in _TRACK_RE.findall(line[:-1]))
                continue
            fields = line.rstrip("\n").split("\t")
            if len(fields) < 3:
                raise IOError("malformatted entry in %s: %r" % (filename, line))
            if ignore_tracks:
                name = "merged"
            elif track is not None:
                if "name" not in track:
                    raise KeyError("track without field 'name' in file '%s'" % filename)
                name = track["name"]
            elif len(fields) >= 4 and fields[3]:
                name = fields[3]
            else:
                name = default_name
            yield name, fields[0], int(fields[1]), int(fields[2])


def readFromBed(filenames, allow_multiple=False, ignore_tracks=False):
    """gat/Engine.pyx:2480-2556: track -> IntervalDictionary.  Track name = `track name=...` line,
    else column 4, else the file's base name; ignore_tracks puts everything into 'merged'.  Tracks in the order of their
    first line, a track's contigs in the order of theirs, a list's intervals in file order."""
    if isinstance(filenames, str):
        filenames = [filenames]
    acc = collections.OrderedDict()        # track -> contig -> ([chunks of starts], [chunks of ends])
    tracks = {}

    def seen(name, filename):
        if name in tracks:
            if tracks[name] != filename:
                if not allow_multiple:
                    raise ValueError("track '%s' in multiple filenames: %s and %s" % (name, tracks[name], filename))
                tracks[name] = filename
        else:
            tracks[name] = filename

    for filename in filenames:
        default_name = os.path.basename(filename)
        cols = None if os.environ.get("GAT_BED_LINE_READER") else _bed_columns(filename)
        if cols is not None:
            chrom, start, end, names = cols
            if len(chrom) == 0:
                continue
            import pandas as pd
            if ignore_tracks or names is None:
                ncode, nuniq = np.zeros(len(chrom), dtype=np.int64), ["merged" if ignore_tracks else default_name]
            else:
                ncode, nuniq = pd.factorize(names)                       # (codes in the order of first appearance)
                nuniq = [x if x else default_name for x in nuniq]
                if len(set(nuniq)) != len(nuniq):                        # an empty name beside the file's own: one track
                    remap = {}
                    m = np.array([remap.setdefault(x, len(remap)) for x in nuniq], dtype=np.int64)
                    # (order of first appearance of the merged names)
                    first_row = np.full(len(remap), len(chrom), dtype=np.int64)
                    np.minimum.at(first_row, m[ncode], np.arange(len(chrom)))
                    rank = np.empty(len(remap), dtype=np.int64)
                    rank[np.argsort(first_row, kind="stable")] = np.arange(len(remap))
                    ncode = rank[m[ncode]]
                    inv = sorted(remap, key=lambda x: rank[remap[x]])
                    nuniq = inv
            ccode, cuniq = pd.factorize(chrom)
            pair = ncode * len(cuniq) + ccode
            order = np.argsort(pair, kind="stable")
            ps = pair[order]
            cut = np.concatenate(([0], np.flatnonzero(ps[1:] != ps[:-1]) + 1, [len(ps)]))
            groups = [(int(ps[cut[g]]) // len(cuniq), int(order[cut[g]]), g) for g in range(len(cut) - 1)]
            groups.sort()                                                  # by track, then by the pair's first line
            s_sorted, e_sorted = start[order], end[order]
            for name in nuniq:
                seen(name, filename)
            for n_, _, g in groups:
                name, contig = nuniq[n_], cuniq[int(ps[cut[g]]) % len(cuniq)]
                se = acc.setdefault(name, collections.OrderedDict()).setdefault(contig, ([], []))
                se[0].append(s_sorted[cut[g]:cut[g + 1]])
                se[1].append(e_sorted[cut[g]:cut[g + 1]])
            continue
        local = collections.OrderedDict()
        last_name = last_contig = se = None
        for name, contig, start, end in _bed_lines(filename, default_name, ignore_tracks):
            if name != last_name or contig != last_contig:       # (sorted files: runs of lines of one list)
                seen(name, filename)
                se = local.setdefault(name, collections.OrderedDict()).setdefault(contig, ([], []))
                last_name, last_contig = name, contig
            assert start <= end, "attempting to add invalid segment %i-%i" % (start, end)
            se[0].append(start)
            se[1].append(end)
        for name, per in local.items():
            for contig, (s_, e_) in per.items():
                se = acc.setdefault(name, collections.OrderedDict()).setdefault(contig, ([], []))
                se[0].append(np.array(s_, dtype=np.int64))
                se[1].append(np.array(e_, dtype=np.int64))
    out = collections.defaultdict(engine.IntervalDictionary)
    for name, per in acc.items():
        d = engine.IntervalDictionary()
        for contig, (s_, e_) in per.items():
            d.add(contig, engine.SegmentList(array=iv.make(np.concatenate(s_), np.concatenate(e_))))
        out[name] = d
    return out


def readSegmentList(label, filenames, enable_split_tracks=False, ignore_tracks=False):
    """gat/IO.py:36-64."""
    results = engine.IntervalCollection(name=label)
    results.intervals = readFromBed(filenames, allow_multiple=enable_split_tracks, ignore_tracks=ignore_tracks)
    return results


def expandGlobs(infiles):
    out = []
    for x in infiles:
        out.extend(glob.glob(x))
    return out


# The input pipeline of gat-run.py, the reference's gat/IO.py:88-293, written as two small tables: which files make up
# which collection and what each collection goes through before the sampler sees it.  The names and the order of the
# --output-stats / --output-bed side files follow the reference (tests/golden/cli/aux/stats/ pins them byte for byte).
_REQUIRED_INPUTS = (("segment_files", "segment"), ("annotation_files", "annotation"), ("workspace_files", "workspace"))


def buildSegments(options):
    """the four collections of a run -- segment tracks, annotation tracks, the collapsed workspace and (optionally)
    isochores restricted to it -- loaded and normalized as gat/IO.py:88-185 does."""
    for attr, what in _REQUIRED_INPUTS:
        setattr(options, attr, expandGlobs(getattr(options, attr)))
        if not getattr(options, attr):
            raise ValueError("please specify at least one %s file" % what)
    options.sample_files = expandGlobs(getattr(options, "sample_files", None) or [])      # gat/IO.py:100

    segments = readSegmentList("segments", options.segment_files, ignore_tracks=options.ignore_segment_tracks)
    segments.normalize()
    if segments.sum() == 0:
        raise ValueError("segments file is empty - run aborted")
    if len(segments) > 1000:
        raise ValueError("too many (%i) segment files - use track definitions or --ignore-segment-tracks" % len(segments))

    relabel = options.annotations_label is not None
    annotations = readSegmentList("annotations", options.annotation_files, enable_split_tracks=options.enable_split_tracks,
                                  ignore_tracks=relabel)
    if relabel:
        annotations.setName(options.annotations_label)
    points = getattr(options, "annotations_to_points", None)
    if points:
        annotations.toPositions(points)                    # before normalizing: coinciding positions count once
    if getattr(options, "overlapping_annotations", False):
        raise NotImplementedError("--overlapping-annotations is outside the accelerated path (counters need normalized lists)")
    annotations.normalize()

    workspaces = readSegmentList("workspaces", options.workspace_files, enable_split_tracks=options.enable_split_tracks)
    workspaces.normalize()
    dumpStats(workspaces, "stats_workspaces_input", options)
    workspaces.collapse()
    dumpStats(workspaces, "stats_workspaces_collapsed", options)
    workspaces.restrict("collapsed")

    isochores = None
    if options.isochore_files:
        isochores = engine.IntervalCollection(name="isochores")
        isochores.intervals = readFromBed(expandGlobs(options.isochore_files))
        dumpStats(isochores, "stats_isochores_raw", options)
        for step in (isochores.sort, isochores.check, isochores.normalize):
            step()
        isochores.intersect(workspaces["collapsed"])
    return segments, annotations, workspaces, isochores


def _all_segments_of(collection):
    """every track of a collection pooled into one IntervalDictionary; a 'merged' track that is already there is used,
    one made for the purpose is removed again."""
    if "merged" in collection:
        return collection["merged"], False
    collection.merge()
    return collection["merged"], True


def applyIsochores(segments, annotations, workspaces, options, isochores=None, truncate_segments_to_workspace=False,
                   truncate_workspace_to_annotations=False, restrict_workspace=False):
    """cut the three collections to the workspace -- per isochore when there are isochores -- and return the workspace
    the sampler runs in (gat/IO.py:188-293)."""
    truncate = options.truncate_segments_to_workspace
    if isochores:
        # (collection, cut to the isochore or keep whole segments that touch it, message if nothing is left)
        plan = ((workspaces, True, "workspaces"), (annotations, True, "annotations"), (segments, truncate, "segments"))
        for coll, cut, _ in plan:
            coll.toIsochores(isochores, truncate=cut)
        for coll, _, what in plan:
            if coll.sum() == 0:
                raise ValueError("isochores and %s do not overlap" % what)
        for dump, prefix in ((dumpStats, "stats_"), (dumpBed, "")):
            for coll, _, what in plan:
                dump(coll, "%s%s_isochores" % (prefix, what), options)
    else:
        within = workspaces["collapsed"]
        if truncate:
            segments.intersect(within)
        else:
            segments.filter(within)
        annotations.intersect(within)
        for coll, what in ((annotations, "annotations"), (segments, "segments")):
            dumpStats(coll, "stats_%s_truncated" % what, options)

    workspace = workspaces["collapsed"]
    if restrict_workspace:
        # only workspace segments that hold a segment of some track stay
        pooled, temporary = _all_segments_of(segments)
        workspace.filter(pooled)
        if temporary:
            del segments["merged"]
        dumpStats(workspaces, "stats_workspaces_restricted", options)
    if truncate_workspace_to_annotations:
        annotations.merge()
        annotations["merged"].normalize()
        workspace.intersect(annotations["merged"])
        del annotations["merged"]
        dumpStats(workspaces, "stats_workspaces_truncated", options)

    wanted = getattr(options, "output_stats", None) or []
    if "overlap" in wanted or "all" in wanted:
        for track in segments.tracks:
            with openOutputFile("overlap_%s" % track, options) as f:
                workspaces.outputOverlapStats(f, segments[track])
    return workspace


class DummyAnnotatorResult(object):
    """a row read back from a results table (gat/__init__.py:439-485): ten columns are kept and printed."""
    format_observed = "%i"
    format_expected = "%6.4f"
    format_fold = "%6.4f"
    format_pvalue = "%6.4e"

    @classmethod
    def _fromLine(cls, line):
        x = cls()
        data = line[:-1].split("\t")
        x.track, x.annotation = data[:2]
        x.counter = "na"
        (x.observed, x.expected, x.lower95, x.upper95, x.stddev, x.fold, x.l2fold, x.pvalue,
         x.qvalue) = [float(v) for v in data[2:11]]
        if len(data) > 11:
            [float(v) for v in data[11:24]]                       # parsed (and checked) but not kept, as in the reference
        return x

    def __str__(self):
        return "\t".join((self.track, self.annotation, self.format_observed % self.observed,
                          self.format_expected % self.expected, self.format_expected % self.lower95,
                          self.format_expected % self.upper95, self.format_expected % self.stddev,
                          self.format_fold % self.fold, self.format_pvalue % self.pvalue,
                          self.format_pvalue % self.qvalue))


def readAnnotatorResults(filename):
    """load rows of a tab-separated results table (gat/IO.py:67-81)."""
    results = []
    with openFile(filename, "r") as infile:
        for line in infile:
            if line.startswith("#") or line.startswith("track"):
                continue
            results.append(DummyAnnotatorResult._fromLine(line))
    return results


def readDescriptions(options):
    """--descriptions: a tab-separated table, first row the header, first column the annotation, the other columns
    are appended to the annotation's rows of the result table (gat/IO.py:296-328).
    Returns (description_header, descriptions, description_width)."""
    filename = getattr(options, "input_filename_descriptions", None)
    if not filename:
        return [], {}, 0
    with openFile(filename) as inf:
        rows = [line[:-1].split("\t") for line in inf if not line.startswith("#")]
    if not rows:
        return [], {}, 0
    width = len(rows[0]) - 1
    for row in rows:
        assert len(row) - 1 == width, "inconsistent number of descriptions in %s" % filename
    header = rows[0][1:]
    assert len(header) == width, "number of descriptions (%i) inconsistent with header (%s) in %s" % (width, len(header), filename)
    return header, dict((row[0], row[1:]) for row in rows[1:]), width


def outputResults(results, options, header, description_header=(), description_width=0, descriptions=None,
                  format_observed="%i"):
    """gat/IO.py:457-539: q-values, one table per counter, rows ordered by --order."""
    pvalues = [x.pvalue for x in results]
    qvalues = stats.getQValues(pvalues, method=options.qvalue_method,
                               vlambda=getattr(options, "qvalue_lambda", None),
                               pi0_method=getattr(options, "qvalue_pi0_method", "smoother"))
    for x, qvalue in zip(results, qvalues):
        x.qvalue = qvalue
        x.format_observed = format_observed
    counters = []
    for x in results:
        if x.counter not in counters:
            counters.append(x.counter)
    for counter in counters:
        if len(counters) == 1:
            outfile, output = options.stdout, list(results)
        else:
            outfile = openFile(re.sub("%s", counter, options.output_tables_pattern), "w")
            output = [x for x in results if x.counter == counter]
        outfile.write("\t".join(list(header) + list(description_header)) + "\n")
        keys = {"track": lambda x: (x.track, x.annotation), "observed": lambda x: x.observed,
                "annotation": lambda x: (x.annotation, x.track), "fold": lambda x: x.fold,
                "pvalue": lambda x: x.pvalue, "qvalue": lambda x: x.qvalue}
        if options.output_order not in keys:
            raise ValueError("unknown sort order %s" % options.output_order)
        output.sort(key=keys[options.output_order])
        for result in output:
            outfile.write(str(result))
            if descriptions:
                outfile.write("\t" + "\t".join(descriptions.get(result.annotation, [""] * description_width)))
            outfile.write("\n")
        if outfile is not options.stdout:
            outfile.close()
