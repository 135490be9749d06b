#!/bin/bash
# Copies what scripts/r4_final_evidence.sh left under gpurun_out/<tag>/ into profiles/ under the round's names.
# Usage: scripts/collect_evidence.sh <tag> <commit>
set -e
R=gpurun_out/$1; C=$2; P=profiles
cp $R/prof_summary.txt $P/r04_final_trace_sq_fetch_write_summary.txt
cp $R/pmc_traffic.json $P/pmc_traffic.json
cp $R/bench3.json $P/r04_bench_config3_1gpu.json
cp $R/bench2.json $P/r04_bench_config2_1gpu.json
cp $R/bench4.json $P/r04_bench_config4_1gpu.json
cp $R/bench3_rejectors.json $P/r04_bench_config3_rejectors_median_trimmed.json
cp $R/bench3_reciprocal.json $P/r04_bench_config3_reciprocal.json
cp $R/bench5_100M_1gpu.json $P/r04_bench_config5_100M_1gpu.json
cp $R/bench5_100M_rank3_fullpass.json $P/r04_bench_config5_100M_rank3_of_8_fullpass.json
for r in 0 1 2 3 4 5 6 7; do cp $R/bench5_100M_rank$r.json $P/r04_bench_config5_100M_rank${r}_of_8.json; done
cp $R/per_iter.txt $P/r04_per_iteration_sq_counters.txt
( echo "# pytest tests -m gpu on the box, commit $C"; grep -E "passed|failed" $R/tests.log | tail -1
  echo "# normals + fuzz tests on a build with ONE record per lane (-DPCLHIP_REC_CAP=1)"; tail -1 $R/tests_rec1.log ) > $P/r04_gpu_tests.txt
grep -v "^/opt" $R/stats.log > $P/r04_work_counters_per_iteration.txt
cp $R/standoff_stage_ticks.log $P/r04_standoff_stage_ticks.txt
cp $R/normals_stage_ticks.log $P/r04_normals_stage_ticks.txt
cp $R/seeded_stage_ticks.log $P/r04_seeded_search_stage_ticks.txt
cp $R/active_lanes_per_round.log $P/r04_active_lanes_per_round.txt
( grep -v "^/opt" $R/knn_probe.log; grep -v "^/opt" $R/radius_probe.log; grep -v "^/opt" $R/misc.log ) > $P/r04_probes.txt
cat $R/fuzz_knn.log $R/fuzz_filters.log > $P/r04_fuzz_vs_oracle.txt
cat $P/r04_gpu_tests.txt
