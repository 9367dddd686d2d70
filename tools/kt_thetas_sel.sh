# timing only (instrumented library): kernel-trace duration of k_thetas over selected (octave, blur segment) ranges
export TMPDIR=/tmp SSRLCV_HIP_LIB=$PWD/ssrlcv_amd/libssrlcv_hip_instrumented.so SSRLCV_SIFT_SERIAL=1
OUT=gpurun_out/r06_thetas
mkdir -p $OUT
for lanes in ${LANES_LIST:-1 4}; do
for sel in ${SEL_LIST:-FFFFF 40000 2000 100 8 4 2}; do
  export SSRLCV_THETAS_LANES=$lanes SSRLCV_TIMING_THETAS_SEL=$sel
  rm -rf $OUT/kts
  rocprofv3 --kernel-trace --stats -d $OUT/kts -o s --output-format csv -- python3 tools/bench_sift_stages.py --size 4096 --scene --stages 6 > $OUT/kts.log 2>&1
  f=$(find $OUT/kts -name "*kernel_stats.csv" | head -1)
  echo "lanes $lanes sel $sel: $(python3 -c "
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'k_thetas' in r['Name']: print('calls %s avg %.1f us min %.1f us' % (r['Calls'], float(r['AverageNs']) / 1e3, float(r['MinNs']) / 1e3))
" $f)"
done
done
rm -rf $OUT/kts
