# kernel-trace durations of the sampling kernels for each lanes-per-key-point setting (serial stage benchmark, developer library)
export TMPDIR=/tmp SSRLCV_DEV_BUILD=1 SSRLCV_SIFT_SERIAL=1
OUT=gpurun_out/r06_thetas
mkdir -p $OUT
for lanes in ${LANES_LIST:-1 2 4}; do
  export SSRLCV_THETAS_LANES=$lanes
  rm -rf $OUT/kt$lanes
  rocprofv3 --kernel-trace --stats -d $OUT/kt$lanes -o s --output-format csv -- python3 tools/bench_sift_stages.py --size 4096 --scene --stages 7 > $OUT/kt$lanes.log 2>&1
  f=$(find $OUT/kt$lanes -name "*kernel_stats.csv" | head -1)
  echo "lanes $lanes $EXTRA_TAG: $(python3 - "$f" <<'PY'
import csv, sys
out = []
for r in csv.DictReader(open(sys.argv[1])):
    n = r['Name']
    for k in ('k_thetas', 'k_descriptors', 'k_polar', 'k_expand_orient'):
        if k in n:
            out.append('%s calls %s avg %.1f us min %.1f' % (k, r['Calls'], float(r['AverageNs']) / 1e3, float(r['MinNs']) / 1e3))
print(' | '.join(sorted(out)))
PY
)"
  rm -rf $OUT/kt$lanes
done
