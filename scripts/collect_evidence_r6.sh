#!/bin/bash
# Copies what scripts/r6_final_evidence.sh left under gpurun_out/<tag>/ into profiles/ under round 6's names.
# Usage: scripts/collect_evidence_r6.sh <tag> <commit>
R=gpurun_out/$1; C=$2; P=profiles
cp $R/prof_summary.txt $P/r06_final_trace_sq_fetch_write_summary.txt
cp $R/pmc_traffic.json $P/pmc_traffic.json
cp $R/bench3.json $P/r06_bench_config3_1gpu.json
cp $R/bench2.json $P/r06_bench_config2.json
cp $R/bench4.json $P/r06_bench_config4.json
cp $R/bench3_rejectors.json $P/r06_bench_config3_rejectors_median_trimmed.json
cp $R/bench3_reciprocal.json $P/r06_bench_config3_reciprocal.json
for c in cube layers clusters; do cp $R/bench3_cloud_$c.json $P/r06_bench_config3_cloud_$c.json; done
for r in 0 3 6; do cp $R/bench5_100M_rank$r.json $P/r06_bench_config5_100M_rank${r}_of_8.json; done
cp $R/bench5_100M_1gpu.json $P/r06_bench_config5_100M_1gpu.json
cp $R/per_iter.txt $P/r06_per_iteration_sq_counters.txt
( echo "# pytest tests -m gpu on the box, commit $C"; grep -E "passed|failed|^real" $R/tests.log | tail -2 ) > $P/r06_gpu_tests.txt
grep -v "^/opt" $R/stats.log > $P/r06_work_counters_per_iteration.txt
cp $R/standoff_stage_ticks.log $P/r06_standoff_stage_ticks.txt
cp $R/seeded_stage_ticks.log $P/r06_seeded_search_stage_ticks.txt
cp $R/active_lanes_per_round.log $P/r06_active_lanes_per_round.txt
( grep -v "^/opt" $R/knn_probe.log; grep -v "^/opt" $R/misc.log ) > $P/r06_probes.txt
grep -vE "^build|amdgpu.ids" $R/index_build_dispatches.log > $P/r06_index_build_dispatches.txt
cp $R/kd_block_kernel_ticks.log $P/r06_kd_block_kernel_ticks.txt 2>/dev/null
cat $P/r06_gpu_tests.txt
