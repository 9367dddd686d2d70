# orientation kernel with 1, 2, 4 lanes per key point: serial stage benchmark, stages 5 and 6 (developer library)
mkdir -p gpurun_out/r06_thetas
export SSRLCV_DEV_BUILD=1 SSRLCV_SIFT_SERIAL=1
for rep in 1 2; do
for lanes in 1 2 4; do
  echo "lanes $lanes: $(SSRLCV_THETAS_LANES=$lanes python tools/bench_sift_stages.py --size 4096 --scene --stages 5,6,7 2>&1 | grep 'stop=[67]' | tr '\n' ' ')"
done
done
