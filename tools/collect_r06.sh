# Developer helper: every profile of round 6 in one call on the GPU box (from the repository root).
# usage: bash tools/collect_r06.sh <commit> [tag]
set -e
COMMIT=$1
export TMPDIR=/tmp
TAG=${2:-r06_v1}
bash tools/collect_profiles.sh $TAG $COMMIT
bash tools/collect_floor.sh r06_floor 4096
python3 tools/pyramid_floor.py gpurun_out/r06_floor 4096 $COMMIT > gpurun_out/r06_floor/pyramid_floor.json
# config[4]'s size: kernel statistics + traffic of the stages on one 8192^2 image (serial: clean per-kernel numbers)
OUT=gpurun_out/r06_8192
mkdir -p $OUT
export SSRLCV_DEV_BUILD=1
export SSRLCV_SIFT_SERIAL=1
rocprofv3 --kernel-trace --stats -d $OUT/serial -o s --output-format csv -- python3 tools/bench_sift_stages.py --size 8192 --scene --stages 7 > $OUT/serial.log 2>&1
cp $(find $OUT/serial -name "*kernel_stats.csv" | head -1) $OUT/serial_kernel_stats.csv
rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc_FETCH_SIZE --output-format csv -- python3 tools/bench_sift_stages.py --size 8192 --scene --stages 7 > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc_WRITE_SIZE --output-format csv -- python3 tools/bench_sift_stages.py --size 8192 --scene --stages 7 > $OUT/pmc_write.log 2>&1
python3 tools/pmc_traffic.py $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE 4 $COMMIT > $OUT/pyramid_traffic.json
# the key-point stage's issue / LDS counters (is k_descriptors waiting for its LDS atomics?)
OUT=gpurun_out/r06_desc
mkdir -p $OUT
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS -d $OUT/pmc --output-format csv -- python3 tools/bench_sift_stages.py --size 4096 --scene --stages 7 > $OUT/pmc.log 2>&1
for k in k_descriptors k_thetas k_polar; do python3 tools/pmcsum.py "$k" $OUT/pmc; done > $OUT/describe_issue_lds.txt 2>&1 || true
unset SSRLCV_SIFT_SERIAL
unset SSRLCV_DEV_BUILD
rm -rf gpurun_out/r06_8192/serial gpurun_out/r06_desc/pmc
ls -la gpurun_out/$TAG gpurun_out/r06_floor gpurun_out/r06_8192 gpurun_out/r06_desc
