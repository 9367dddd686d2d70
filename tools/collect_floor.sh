# Developer helper: the passes behind profiles/rNN_pyramid_floor.json (run on the GPU box from the repository root).
# usage: bash tools/collect_floor.sh <tag> [size=4096]
# Serial (no side streams) per-dispatch kernel trace of the stage benchmark + separate FETCH_SIZE / WRITE_SIZE / MFMA-busy
# passes; tools/pyramid_floor.py reduces them per pyramid launch.  python3 directly after `--`, environment set here.
set -e
TAG=$1
SIZE=${2:-4096}
export TMPDIR=/tmp
OUT=gpurun_out/$TAG
mkdir -p $OUT
export SSRLCV_DEV_BUILD=1
export SSRLCV_SIFT_SERIAL=1
rocprofv3 --kernel-trace -d $OUT/trace -o t --output-format csv -- python3 tools/bench_sift_stages.py --size $SIZE --scene --stages 7 > $OUT/trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc_FETCH_SIZE --output-format csv -- python3 tools/bench_sift_stages.py --size $SIZE --scene --stages 7 > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc_WRITE_SIZE --output-format csv -- python3 tools/bench_sift_stages.py --size $SIZE --scene --stages 7 > $OUT/pmc_write.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU GRBM_GUI_ACTIVE -d $OUT/pmc_MFMA --output-format csv -- python3 tools/bench_sift_stages.py --size $SIZE --scene --stages 7 > $OUT/pmc_mfma.log 2>&1
find $OUT -name "*.csv" | xargs ls -la
