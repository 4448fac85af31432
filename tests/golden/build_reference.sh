#!/bin/bash
# Scratch build of the reference (AndreasHeger/gat 1.3.6, /root/reference) used ONLY to generate
# the golden vectors in this directory (tests/golden/make_goldens.py).  Runs in this container
# only; nothing of the reference is copied into the repository and nothing here travels to the
# GPU box.  The reference is Python/Cython; it needs seven textual substitutions to compile with
# this image's Cython 3 / numpy 2 (removed numpy.int/numpy.float aliases, Cython-3 noexcept on
# the qsort/searchsorted comparators, Py2 division/except syntax -> language_level=2).  The
# substitutions touch no algorithmic line.  After building, the reference's own unit tests are
# run (59 pass) before any vector is taken from it.
set -euo pipefail
REF=${REF:-/root/reference}
OUT=${OUT:-/tmp/gatbuild}
rm -rf "$OUT" && cp -r "$REF" "$OUT" && chmod -R u+w "$OUT" && cd "$OUT"
sed -i 's/numpy\.int_t/numpy.npy_long/; s/numpy\.float_t/numpy.npy_double/; s/= numpy\.int$/= int/; s/= numpy\.float$/= float/;
        s/dtype *= *numpy\.int *)/dtype=int)/g; s/dtype *= *numpy\.float *)/dtype=float)/g; s/dtype=numpy\.float)/dtype=float)/g' \
    gat/SegmentList.pyx gat/Engine.pyx gat/PositionList.pyx gat/__init__.py
sed -i 's/numpy\.float\b/float/g; s/numpy\.int\b/int/g' gat/Stats.py gat/IOTools.py gat/IO.py gat/__init__.py gat/Experiment.py
# ConditionalSampler.sample (gat/__init__.py:832) joins the track name with an int and dies with a TypeError before
# sampling anything; the one-token fix below lets the conditional workspaces run so that they can be pinned too
sed -i "s/'_'.join((track, annoid))/'_'.join((track, str(annoid)))/" gat/__init__.py
# --sample-file (gat/__init__.py:957) builds its regex with re.sub("%s", "(\S+)", pattern): Python >= 3.7 rejects the
# unknown escape \S in a replacement string (re.error: bad escape), so the option cannot run at all under this image's
# Python; a raw, doubled backslash gives the regex the line was written for (Python 2) and lets the option be pinned
sed -i 's/re.sub("%s", "(\\S+)", output_samples_pattern)/re.sub("%s", r"(\\\\S+)", output_samples_pattern)/' gat/__init__.py
cat > setup_probe.py <<'PY'
import numpy
from setuptools import setup, Extension
from Cython.Build import cythonize
exts = [Extension("gat.%s" % n, ["gat/%s.pyx" % n, "utils/gat_utils.c"], libraries=['z', 'rt'],
                  include_dirs=["./utils", numpy.get_include()], language="c")
        for n in ("CoordinateList", "SegmentList", "PositionList", "Engine")]
setup(name="gat", ext_modules=cythonize(exts, include_path=["gat"],
      compiler_directives=dict(language_level=2, legacy_implicit_noexcept=True)))
PY
python setup_probe.py build_ext --inplace > build.log 2>&1 || { tail -30 build.log; exit 1; }
cd test && PYTHONPATH="$OUT" python -m pytest -q test_SegmentList.py test_PositionList.py test_gat.py test_gat_stats.py
echo "reference scratch build ready in $OUT"
