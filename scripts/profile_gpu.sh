#!/bin/bash
# Run on the GPU box (via gpurun): kernel trace + PMC passes of the bench command.
# Usage: scripts/profile_gpu.sh <tag> "<passes>" [bench args...]   -> gpurun_out/prof_<tag>/
# passes: any of trace sq1 sq2 fetch write tcc.  PMC passes are separate runs with --kernel-trace
# only (never combined with sys/hip traces).
set -u
TAG=${1:-r1}; PASSES=${2:-"trace sq1 sq2 fetch write tcc"}; shift; shift
# default: the driver's own command (bench.py with no flags = --steps 20 --warmup 5), less the CPU baseline, so that the
# summary's average launch duration is the bench line's roofline.avg_kernel_ms (4 alignments x 5 iterations timed)
ARGS=${@:-"--steps 20 --warmup 5 --no-cpu-baseline --no-host-align --no-families"}
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
run() { # name, rocprof flags...
  local name=$1; shift
  rm -rf /tmp/rp_$name
  timeout 900 rocprofv3 "$@" -d /tmp/rp_$name -o $name --output-format csv -- python $REPO/bench.py $ARGS > $OUT/$name.log 2>&1
  echo "== $name rc=$?"
  find /tmp/rp_$name -name "*.csv" | while read f; do cp "$f" $OUT/$(basename "$f"); done
}
for p in $PASSES; do
case $p in
trace) run trace --kernel-trace --stats ;;
sq1) run pmc_sq1 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM ;;
sq2) run pmc_sq2 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SMEM SQ_WAIT_INST_LDS ;;
fetch) run pmc_fetch --kernel-trace --pmc FETCH_SIZE ;;
write) run pmc_write --kernel-trace --pmc WRITE_SIZE ;;
tcc) run pmc_tcc --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum ;;
esac
done
python $REPO/scripts/summarize_prof.py $OUT > $OUT/summary.txt 2>&1; cat $OUT/summary.txt
# HBM bytes per launch from the FETCH/WRITE passes, stamped with the commit (bench.py reads profiles/pmc_traffic.json;
# copy $OUT/pmc_traffic.json there after a run on the GPU box)
case "$PASSES" in *fetch*write*|*write*fetch*)
  python $REPO/scripts/make_pmc_traffic.py $OUT "${GRAFT_COMMIT:-$(git -C $REPO rev-parse --short HEAD 2>/dev/null || echo unknown)}" "bench.py $ARGS" ;;
esac
