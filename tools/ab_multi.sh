#!/bin/bash
# tools/ab_multi.sh "<flags A>" "<flags B>" ... -- bench.py (4096 envs unless BENCH_ARGS says otherwise) with the product library
# and with scratch builds of the same sources + each flag set, all on ONE box, two rounds.
SRC=hanabizero_amd/csrc
mkdir -p scratch_ab
i=0
for F in "$@"; do
  i=$((i+1))
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -fno-fast-math \
    -fhip-fp32-correctly-rounded-divide-sqrt -w -I$SRC -Iinclude $F -o scratch_ab/libvariant$i.so \
    $SRC/hz_tree.hip $SRC/hz_env.hip $SRC/hz_selfplay.hip $SRC/hz_netglue.hip $SRC/hz_mlp.hip $SRC/hz_search.hip $SRC/hz_movetail.hip || exit 1
done
run() { python bench.py --no-cpu-baseline --no-roofline --no-also --steps 300 $BENCH_ARGS 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-40s %.4f ms/step %.0f moves/s' % ('$1', d['ms_per_step'], d['value']))"; }
for rep in 1 2; do
  run product || exit 1
  i=0
  for F in "$@"; do
    i=$((i+1))
    HANABIZERO_HIP_LIB=$PWD/scratch_ab/libvariant$i.so run "$F" || exit 1
  done
done
