#!/bin/bash
# tools/ab_lib_kernel.sh <other library.so> [bench args...] -- as tools/ab_lib.sh, with bench.py's own timing of the search kernel
LIB="$1"; shift
P='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d["roofline"]; print("%-8s %.4f ms/step %.0f moves/s | k_search avg %.1f us, best %.1f us" % (sys.argv[1], d["ms_per_step"], d["value"], r["avg_launch_us"], r["min_launch_us"]))'
for rep in 1 2; do
  python bench.py --no-cpu-baseline --no-also --steps 200 "$@" 2>/dev/null | python -c "$P" product || exit 1
  HANABIZERO_HIP_LIB=$PWD/$LIB python bench.py --no-cpu-baseline --no-also --steps 200 "$@" 2>/dev/null | python -c "$P" other || exit 1
done
