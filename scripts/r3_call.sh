#!/bin/bash
# one GPU-box call of round 3: tests + fuzz + A/B of the stand-off path.  usage: scripts/r3_call.sh <tag> [what...]
tag=${1:-c1}; shift
what=${*:-tests fuzz ab stats bench}
out=gpurun_out/r3_$tag; mkdir -p $out
export HSA_ENABLE_IPC_MODE_LEGACY=0
for w in $what; do
case $w in
tests) timeout 900 python -m pytest tests -m gpu -x -q > $out/tests.log 2>&1; echo "tests exit $?" >> $out/tests.log; tail -5 $out/tests.log;;
tests_fast) timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_loop.py -m gpu -x -q > $out/tests_fast.log 2>&1; echo "exit $?" >> $out/tests_fast.log; tail -5 $out/tests_fast.log;;
fuzz) timeout 600 python scratch/fuzz_knn.py 3 30 > $out/fuzz.log 2>&1; tail -3 $out/fuzz.log;;
ab) for so in 0 1; do echo "== PCLHIP_STANDOFF=$so" >> $out/ab.log; PCLHIP_STANDOFF=$so timeout 300 python scratch/iter_probe.py >> $out/ab.log 2>&1; done; cat $out/ab.log;;
stats) for so in 0 1; do echo "== PCLHIP_STANDOFF=$so" >> $out/stats.log; PCLHIP_STANDOFF=$so timeout 300 python scratch/stats_probe.py 10000000 >> $out/stats.log 2>&1; done; cat $out/stats.log;;
bench) for so in 0 1; do echo "== PCLHIP_STANDOFF=$so" >> $out/bench.log; PCLHIP_STANDOFF=$so timeout 600 python scratch/ab.py 20 >> $out/bench.log 2>&1; done; cat $out/bench.log;;
prof) bash scripts/profile_iter.sh r3_$tag > $out/prof.log 2>&1; cp gpurun_out/prof_r3_$tag/per_iter.txt $out/per_iter.txt; cat $out/per_iter.txt;;
stats1) timeout 300 python scratch/stats_probe.py 10000000 > $out/stats1.log 2>&1; cat $out/stats1.log;;
sprof) PCLHIP_LIB=pcl_amd/variants/libpclhip_prof.so timeout 300 python scratch/stats_probe.py 10000000 > $out/sprof.log 2>&1; grep "it0" $out/sprof.log;;
abv) # A/B of library variants x PCLHIP_SO_FACTOR: ABV="default:1 default:0.25 w3:1"
  for v in $ABV; do lib=${v%%:*}; f=${v##*:}; L=pcl_amd/libpclhip.so; [ $lib != default ] && L=pcl_amd/variants/libpclhip_$lib.so
    echo "== $lib factor $f" >> $out/abv.log; PCLHIP_LIB=$L PCLHIP_SO_FACTOR=$f timeout 300 python scratch/stats_probe.py 10000000 2>&1 | grep "it0" >> $out/abv.log; done; cat $out/abv.log;;
esac
done
