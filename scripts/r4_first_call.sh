#!/bin/bash
# The first GPU-box call of round 4: everything round 3's last commit could not get because
# the round's GPU budget ended with its bench A/B.  Usage (through gpurun --timeout 2400; ~25 GPU-minutes, sections can be commented out):
#   GRAFT_COMMIT=$(git rev-parse --short HEAD) bash scripts/r4_first_call.sh
#   -> gpurun_out/r4first/ (+ gpurun_out/prof_r4first/); copy what is to be judged into profiles/ as r04_*.
set -u
export GRAFT_COMMIT=${GRAFT_COMMIT:-unknown}
OUT=gpurun_out/r4first
mkdir -p $OUT
# 1. the short evidence run: -m gpu tests, trace + FETCH/WRITE passes of the driver's command (the cold launch's fetch
#    volume under the front-ordered runs: 2.3 GB with one long run per wave), all bench lines, k-NN timings
EVIDENCE_SHORT=1 bash scripts/final_evidence.sh r4first > $OUT/evidence.log 2>&1; tail -3 $OUT/evidence.log
# 2. the gate on the index size of the stand-off search (PCLHIP_SO_MAX_MB, default 640 = between 10M and 15M points) dates
#    from the old schedule: cold launch at 12M / 15M / 20M points with the gate as it is and lifted
for n in 12000000 15000000 20000000; do
  for mb in 640 100000; do
    echo "== n $n PCLHIP_SO_MAX_MB $mb" >> $OUT/gate.log
    PCLHIP_SO_MAX_MB=$mb timeout 300 python scratch/big_cold_probe.py $n >> $OUT/gate.log 2>&1
  done
done
cat $OUT/gate.log
# 3. the same correspondence check the CPU emulation ran at 3M points (profiles/r03_wavesim_icp_check_3M.txt), on the
#    device at 3M and 10M: the emulation's answer and the hardware's must be the same lines
for n in 3000000 10000000; do timeout 600 python scratch/wavesim_icp_check.py $n > $OUT/icp_check_$n.log 2>&1; tail -4 $OUT/icp_check_$n.log; done
# 4. the stand-off schedule and its run starts (compile time).  Build the variants BEFORE the call, in the CPU container
#    (scripts/r4_build_variants.sh builds all of these and the ones of section 6):
#      scripts/build_variant.sh run3 "-DPCLHIP_COLD_RUN=3";  scripts/build_variant.sh run6 "-DPCLHIP_COLD_RUN=6"
#      scripts/build_variant.sh greedy "-DPCLHIP_SO_GREEDY_SEED=1"
#      scripts/build_variant.sh greedy_r2 "-DPCLHIP_SO_GREEDY_SEED=1 -DPCLHIP_COLD_RUN=2"
#      scripts/build_variant.sh greedy_r1 "-DPCLHIP_SO_GREEDY_SEED=1 -DPCLHIP_COLD_RUN=1"
#    (greedy: a run's first group takes its seed from one walk down the hierarchy towards the group's centre instead of one
#    lane's exact neighbour; on the emulation -- profiles/r03_wavesim_cold_seed_counters.txt -- that seed is BETTER than the
#    predecessor's match: runs of 1 do 12 % fewer evaluation rounds and lane-cull steps than the default's runs of 4, with
#    no one-lane searches at all, and every group is then independent of its neighbours: the schedule can be fully dynamic)
for v in run3 run6 greedy greedy_r2 greedy_r1 greedy_r1_grec; do
  L=pcl_amd/variants/libpclhip_$v.so
  [ -f $L ] && { echo "== $v" >> $OUT/run_ab.log; PCLHIP_LIB=$L timeout 300 python bench.py --no-cpu-baseline --no-host-align >> $OUT/run_ab.log 2>&1; }
done
[ -f $OUT/run_ab.log ] && cat $OUT/run_ab.log
# 5. the served-group lists of the sharded mode (opt-in until this has passed): their test on the device,
PCLHIP_HW_VALIDATE=1 timeout 900 python -m pytest tests/test_gpu_dist.py -m gpu -q -k served_group > $OUT/served_groups_test.log 2>&1
tail -3 $OUT/served_groups_test.log
#    then on one GPU: rank 3 of 8 of a 30M-point job with the
#    lists and with the full pass -- per-iteration time of ONE rank (bench.py --config 5 --virtual-world)
for og in 1 0; do  # (1 = the lists, 0 = the full pass)
  echo "== PCLHIP_OWNED_GROUPS=$og" >> $OUT/virtual_rank.log
  PCLHIP_OWNED_GROUPS=$og timeout 600 python bench.py --config 5 --points 30000000 --virtual-world 8 --virtual-rank 3 \
    --steps 20 --warmup 5 >> $OUT/virtual_rank.log 2>&1
done
cat $OUT/virtual_rank.log
# 6. group leaf lists across seeded iterations (-DPCLHIP_GROUP_LISTS=1, traverse.hpp: GroupRec; exact on the emulation, 100 %
#    of the converged groups searched from their record, node scans 2.8 -> 0, rounds 4.2 -> 3.2 per group): build the
#    variant BEFORE the call (scripts/build_variant.sh grec "-DPCLHIP_GROUP_LISTS=1"; also grec103 with
#    "-DPCLHIP_GROUP_LISTS=1 -DPCLHIP_GREC_GROW=1.03f": on the emulation a larger growth than 1.1 brings no earlier iteration in
#    -- the launches before are loose searches and leave no record -- and a smaller one lists fewer leaves), then A/B against
#    the default and check it against the oracle
for v in grec grec103; do
  L=pcl_amd/variants/libpclhip_$v.so
  [ -f $L ] || continue
  echo "== $v" >> $OUT/grec_ab.log
  PCLHIP_LIB=$L timeout 300 python bench.py --no-cpu-baseline --no-host-align >> $OUT/grec_ab.log 2>&1
  PCLHIP_LIB=$L timeout 300 python tests/wavesim/group_lists_probe.py 2000000 2>&1 | tail -3 >> $OUT/grec_ab.log
  PCLHIP_LIB=$L timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_loop.py tests/test_gpu_fuzz.py -m gpu -q -x 2>&1 | tail -2 >> $OUT/grec_ab.log
done
[ -f $OUT/grec_ab.log ] && cat $OUT/grec_ab.log
