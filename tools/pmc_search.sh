#!/bin/bash
# tools/pmc_search.sh -- hardware-counter passes over the persistent search kernel of a short bench.py run (one rocprofv3
# --pmc pass per counter group: L1 <-> L2 latency and stalls, texture addresser, sequencer, L2).  Run on the GPU box from the
# repo root; writes gpurun_out/pmc_search/<group>/ and a one-line-per-counter summary gpurun_out/pmc_search/summary.txt.
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/pmc_search
rm -rf $OUT && mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $ROOT
i=0
for group in \
  "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum" \
  "TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TA_TCP_STATE_READ_sum TCP_RFIFO_STALL_CYCLES_sum" \
  "TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum" \
  "TA_DATA_STALLED_BY_TC_CYCLES_sum TA_ADDR_STALLED_BY_TD_CYCLES_sum" \
  "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM" \
  "SQ_INST_CYCLES_VMEM_RD SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
  "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_MFMA" \
  "TCC_READ_REQ_LATENCY_sum TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum" \
  "TCC_BUSY_avr TCC_CYCLE_sum TCC_IB_STALL_sum TCC_LATENCY_FIFO_FULL_sum" ; do
  i=$((i+1))
  # (every group below fits one pass on gfx950 -- each was collected in rounds 1 and 2 -- so a pass that fails is an error: stop;
  # the timeout only bounds a pass that no longer finishes.  The interpreter itself follows `--`: no env / shell hop.)
  timeout -k 5 200 rocprofv3 --pmc $group --kernel-include-regex "k_search" --output-format csv -d $OUT/g$i -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-also "$@" > $OUT/g$i.log 2>&1 || { echo "group $i FAILED: $group (see $OUT/g$i.log)"; tail -5 $OUT/g$i.log; exit 1; }
  echo "group $i done: $group"
done
python - <<'PY'
import csv, glob, collections, os
out = os.path.join("gpurun_out", "pmc_search")
rows = collections.defaultdict(list)
for f in glob.glob(os.path.join(out, "g*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_search" in r["Kernel_Name"]:
            rows[r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(os.path.join(out, "summary.txt"), "w") as fh:
    for k in sorted(rows):
        v = rows[k]
        line = "%-45s launches %3d  mean %.6g" % (k, len(v), sum(v) / len(v))
        print(line)
        fh.write(line + "\n")
PY
