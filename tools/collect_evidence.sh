#!/bin/bash
# Container side, after `gpurun -- bash tools/round_verify.sh <tag>`: copies the judged summaries from gpurun_out/ (scratch) into profiles/ (tracked).
TAG=${1:-r06}; R=$(cd "$(dirname "$0")/.." && pwd); G=$R/gpurun_out; P=$R/profiles
for f in kernel_stats kernel_stats_streams1 kernel_stats_streams1_etoile kernel_stats_streams1_phase_machine kernel_stats_streams1_refill \
         pmc_FETCH_SIZE pmc_WRITE_SIZE pmc_SQ_WAVES pmc_SQ_lane_utilisation pmc_TA pmc_TCC pmc_TCP calib_FETCH_SIZE calib_WRITE_SIZE; do
  [ -f $G/${TAG}_$f.csv ] && cp $G/${TAG}_$f.csv $P/${TAG}_$f.csv
done
[ -f $G/${TAG}_calib.json ] && cp $G/${TAG}_calib.json $P/
for f in bench_line_1440 bench_etoile bench_bidir_room bench_bidir_room_cap96; do [ -s $G/${TAG}final/$f.json ] && cp $G/${TAG}final/$f.json $P/${TAG}_$f.json; done
[ -f $G/${TAG}final/tests.log ] && tail -8 $G/${TAG}final/tests.log > $P/${TAG}_gpu_tests.log
[ -f $G/${TAG}final/smoke.log ] && cp $G/${TAG}final/smoke.log $P/${TAG}_smoke.log
python $R/tools/make_traffic_json.py $G $TAG $P/${TAG}_pmc_traffic.json 1440 cornell_box 0 0 2>&1 | tail -2
ls -la $P | grep ${TAG}_ | wc -l
