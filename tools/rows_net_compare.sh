#!/bin/bash
# tools/rows_net_compare.sh <workload> -- moves/s of the search-kernel shapes (hz_search_run's rows_per_workgroup) with random-init
# and with sharp-policy nets (deep search paths); writes gpurun_out/rows_net_<workload>.txt
W=${1:-full8192}
OUT=gpurun_out/rows_net_$W.txt
: > $OUT
for net in random sharp; do
  for r in 16 -16 32 -32; do
    if [ "$W" = "full4096" ] && [ "${r#-}" = "32" ]; then continue; fi
    if [ "$W" != "full4096" ] && [ "${r#-}" = "16" ]; then continue; fi
    python bench.py --steps 40 --no-cpu-baseline --no-also --no-roofline --workload $W --net $net --rows-per-workgroup $r 2>/dev/null > /tmp/rn.json
    python -c "import json; d=json.load(open('/tmp/rn.json')); print('$W net=$net rows_per_workgroup=$r: %.0f moves/s' % d['value'])" >> $OUT
  done
done
cat $OUT
