cd $GRAFT_REPO_ROOT
export GPU_MAX_HW_QUEUES=8
mkdir -p gpurun_out
tools/profile_round.sh r03 > gpurun_out/r03_profile_round.log 2>&1
python tools/make_traffic_json.py gpurun_out r03 gpurun_out/r03_pmc_traffic.json 1440 cornell_box 25 4 | tail -1
# the other two workloads: bench lines + traffic
for sc in etoile bidir_room; do
  res=720; [ $sc = bidir_room ] && res=1920
  tools/profile_traffic.sh r03_$sc $sc $res
  cp gpurun_out/r03_calib.json gpurun_out/r03_${sc}_calib.json; cp gpurun_out/r03_calib_FETCH_SIZE.csv gpurun_out/r03_${sc}_calib_FETCH_SIZE.csv; cp gpurun_out/r03_calib_WRITE_SIZE.csv gpurun_out/r03_${sc}_calib_WRITE_SIZE.csv
  python tools/make_traffic_json.py gpurun_out r03_$sc gpurun_out/r03_pmc_traffic_$sc.json $res $sc 25 4 | tail -1
done
# bench lines (default run = what the driver runs, with live traffic + cpu baseline), then the two other workloads
timeout 600 python bench.py > gpurun_out/r03_bench_line_1440.json 2> gpurun_out/r03_bench_line_1440.err; tail -c 600 gpurun_out/r03_bench_line_1440.json
for sc in etoile bidir_room; do
  res=720; [ $sc = bidir_room ] && res=1920
  timeout 600 python bench.py --scene $sc --res $res --steps 8 --warmup 2 > gpurun_out/r03_bench_$sc.json 2> gpurun_out/r03_bench_$sc.err
done
ls gpurun_out | grep r03_ | head -50
