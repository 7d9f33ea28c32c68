#!/bin/bash
# tools/profile_round.sh <tag> -- the rocprofv3 passes profiles/ is written from, on the GPU box from the repo root:
#   kernel trace + stats of bench.py for full4096 and full8192 (the same command bench.py's numbers come from), and the two
#   HBM-traffic counter passes (FETCH_SIZE, WRITE_SIZE: separate --pmc runs, no trace options) per workload.
# Outputs under gpurun_out/<tag>_*; tools/summarize_rocprof.py and tools/pmc_traffic.py turn them into profiles/ files.
TAG=${1:-r03}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for W in full4096 full8192; do
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${TAG}_stats_$W -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-also --workload $W > gpurun_out/${TAG}_stats_$W.json 2> gpurun_out/${TAG}_stats_$W.err || exit 1
  for C in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $C --output-format csv -d gpurun_out/${TAG}_pmc_${C}_$W -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-also --workload $W > gpurun_out/${TAG}_pmc_${C}_$W.log 2>&1 || exit 1
  done
  echo "$W done"
done
find gpurun_out -name "*kernel_stats.csv" -path "*${TAG}_stats*"; find gpurun_out -name "*counter_collection.csv" -path "*${TAG}_pmc*"
