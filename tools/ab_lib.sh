#!/bin/bash
# tools/ab_lib.sh <other library.so> [bench args...] -- A/B on ONE box: bench.py with the product library and with another build
# of it (e.g. the previous commit's, copied aside before rebuilding), alternating twice.
LIB="$1"; shift
for rep in 1 2; do
  python bench.py --no-cpu-baseline --no-roofline --no-also --steps 300 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('product   %.4f ms/step %.0f moves/s' % (d['ms_per_step'], d['value']))" || exit 1
  HANABIZERO_HIP_LIB=$PWD/$LIB python bench.py --no-cpu-baseline --no-roofline --no-also --steps 300 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('other     %.4f ms/step %.0f moves/s' % (d['ms_per_step'], d['value']))" || exit 1
done
