# Developer helper: build_dog timing under a list of environment variants (one process each; the switches are read once).
# (the switches exist in the developer build only: SSRLCV_DEV_BUILD=1 selects it, ssrlcv_amd/_lib.py)
# usage: bash tools/ab_pyramid.sh "VAR=1 VAR2=x" "VAR=2" ...     ("" = defaults)
for v in "$@"; do
  echo "== ${v:-defaults}"
  env SSRLCV_DEV_BUILD=1 $v python3 tools/bench_pyramid.py --size 4096 --iters 12 2>&1 | grep build_dog
done
