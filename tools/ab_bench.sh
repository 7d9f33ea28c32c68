#!/bin/bash
# tools/ab_bench.sh "<extra hipcc flags>" [bench args...] -- A/B on ONE box: bench.py with the product library, then with a
# scratch build of the same sources + the extra flags (HANABIZERO_HIP_LIB), alternating twice.
FLAGS="$1"; shift
SRC=hanabizero_amd/csrc
mkdir -p scratch_ab
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -fno-fast-math \
  -fhip-fp32-correctly-rounded-divide-sqrt -w -I$SRC -Iinclude $FLAGS -o scratch_ab/libvariant.so \
  $SRC/hz_tree.hip $SRC/hz_env.hip $SRC/hz_selfplay.hip $SRC/hz_netglue.hip $SRC/hz_mlp.hip $SRC/hz_search.hip || exit 1
for rep in 1 2; do
  python bench.py --no-cpu-baseline --no-roofline --steps 300 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('product  %.4f ms/step %.0f moves/s' % (d['ms_per_step'], d['value']))"
  HANABIZERO_HIP_LIB=$PWD/scratch_ab/libvariant.so python bench.py --no-cpu-baseline --no-roofline --steps 300 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('variant  %.4f ms/step %.0f moves/s' % (d['ms_per_step'], d['value']))"
done
