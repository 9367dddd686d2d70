# timing only: the orientation kernel over selected (octave, blur segment) ranges (instrumented library)
mkdir -p gpurun_out/r06_thetas
export SSRLCV_HIP_LIB=$PWD/ssrlcv_amd/libssrlcv_hip_instrumented.so SSRLCV_SIFT_SERIAL=1
for lanes in ${LANES_LIST:-4}; do
for sel in FFFFF 0 8 4 2 1F FFFE0; do
  echo "lanes $lanes sel $sel: $(SSRLCV_THETAS_LANES=$lanes SSRLCV_TIMING_THETAS_SEL=$sel python tools/bench_sift_stages.py --size 4096 --scene --stages 5,6 2>&1 | grep 'stop=6')"
done
done
