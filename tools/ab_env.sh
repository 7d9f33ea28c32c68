#!/bin/bash
# tools/ab_env.sh VAR=VALUE [bench args...] -- A/B on ONE box: bench.py as it is, then with the environment variable set, alternating twice.
KV="$1"; shift
for rep in 1 2; do
  python bench.py --no-cpu-baseline --no-roofline --no-also --steps 300 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('default   %.4f ms/step %.0f moves/s' % (d['ms_per_step'], d['value']))" || exit 1
  env "$KV" python bench.py --no-cpu-baseline --no-roofline --no-also --steps 300 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$KV  %.4f ms/step %.0f moves/s' % (d['ms_per_step'], d['value']))" || exit 1
done
