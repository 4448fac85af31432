#!/bin/bash
# rocprofv3 summaries behind DESIGN.md / bench.py's roofline, one BASELINE shape at a time: kernel-trace stats, then the
# counters in separate --pmc passes (FETCH_SIZE and WRITE_SIZE do not fit one pass on gfx950; no --pmc together with
# trace domains other than --kernel-trace).  usage (GPU box): bash tools/collect_profiles.sh <round tag> [config:samples ...]
# (gpurun merges what this writes into the local gpurun_out/ next to what earlier calls left there: remove the local
#  gpurun_out/prof_<tag> first, or tools/summarize_profiles.py sums the counters of old and new kernels alike)
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
TAG=$1; shift
SHAPES=${@:-"config2:10000 config3:10000 config5:125000 config4:12500 refdata:10000"}
for SH in $SHAPES; do
  CFG=${SH%%:*}; S=${SH##*:}
  OUT=$R/gpurun_out/prof_${TAG}/${CFG}; rm -rf $OUT; mkdir -p $OUT
  ARGS="--no-cpu-baseline --no-api --no-strong --sustain-seconds 0 --extra= --config $CFG --samples $S --steps 3 --warmup 1"
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py $ARGS --details $OUT/bench.json > $OUT/bench_trace.log 2>&1
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 bench.py $ARGS --details /dev/null > $OUT/bench_fetch.log 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 bench.py $ARGS --details /dev/null > $OUT/bench_write.log 2>&1
  rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $OUT/sq1 -- python3 bench.py $ARGS --details /dev/null > $OUT/bench_sq1.log 2>&1
  rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_INSTS_BRANCH --output-format csv -d $OUT/sq2 -- python3 bench.py $ARGS --details /dev/null > $OUT/bench_sq2.log 2>&1
  rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM --output-format csv -d $OUT/sq3 -- python3 bench.py $ARGS --details /dev/null > $OUT/bench_sq3.log 2>&1
  rocprofv3 --kernel-trace --pmc TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/tcc -- python3 bench.py $ARGS --details /dev/null > $OUT/bench_tcc.log 2>&1
  cp $OUT/trace/*/*kernel_stats.csv $OUT/kernel_stats.csv 2>/dev/null
done
python3 tools/summarize_profiles.py $R/gpurun_out/prof_${TAG} $TAG
