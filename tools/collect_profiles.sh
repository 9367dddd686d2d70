# Developer helper: the rocprofv3 passes behind profiles/rNN_* (run on the GPU box from the repository root).
# usage: bash tools/collect_profiles.sh <tag> <commit>      e.g. r03_v1 fad0457
# Writes under gpurun_out/<tag>/: kernel statistics + JSON line of the bench command (side streams on), the SERIAL
# per-kernel statistics of the stage benchmark, and the reductions of the FETCH_SIZE / WRITE_SIZE / SQ_INSTS passes
# (pyramid traffic, describe PMC), of an MFMA-busy pass over the Gaussian kernels and of the matcher's counters.
# Every rocprofv3 line has python3 directly after `--` and the environment set in THIS shell (no env / bash -c hop
# behind the profiler: that is a forbidden exec on this pool).
set -e
TAG=$1
COMMIT=$2
export TMPDIR=/tmp
OUT=gpurun_out/$TAG
mkdir -p $OUT
rocprofv3 --kernel-trace --stats -d $OUT/stats -o s --output-format csv -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-class-api --no-nview > $OUT/bench_line.json 2> $OUT/bench.err
cp $(find $OUT/stats -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv
# (the serial per-kernel passes need the developer build: the release library has no switches)
export SSRLCV_DEV_BUILD=1
export SSRLCV_SIFT_SERIAL=1
rocprofv3 --kernel-trace --stats -d $OUT/serial -o s --output-format csv -- python3 tools/bench_sift_stages.py --size 4096 --scene --stages 7 > $OUT/serial.log 2>&1
cp $(find $OUT/serial -name "*kernel_stats.csv" | head -1) $OUT/serial_kernel_stats.csv
rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc_FETCH_SIZE --output-format csv -- python3 tools/bench_sift_stages.py --size 4096 --scene --stages 7 > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc_WRITE_SIZE --output-format csv -- python3 tools/bench_sift_stages.py --size 4096 --scene --stages 7 > $OUT/pmc_write.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS -d $OUT/pmc_SQ --output-format csv -- python3 tools/bench_sift_stages.py --size 4096 --scene --stages 7 > $OUT/pmc_sq.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE -d $OUT/pmc_MFMA --output-format csv -- python3 tools/bench_sift_stages.py --size 4096 --scene --stages 7 > $OUT/pmc_mfma.log 2>&1
unset SSRLCV_SIFT_SERIAL
unset SSRLCV_DEV_BUILD
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VALU -d $OUT/pmc_MATCH --output-format csv -- python3 tools/bench_matcher.py 262144 2 > $OUT/pmc_match.log 2>&1
FEATURES=$(grep "stop=7" $OUT/pmc_sq.log | sed 's/.*n=//')
python3 tools/pmc_traffic.py $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE 4 $COMMIT > $OUT/pyramid_traffic.json
python3 tools/pmc_describe.py $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE $OUT/pmc_SQ 4 $FEATURES $COMMIT > $OUT/describe_pmc.json
for k in "k_gauss_mfma2<32, 256" "k_gauss_rm<24" "k_gauss_rm<16" "k_gauss_rm<12" "k_gauss_mfma2<24, 128" "k_gauss_mfma2<16, 128" "k_gauss_mfma2<12, 128"; do python3 tools/pmcsum.py "$k" $OUT/pmc_MFMA; done > $OUT/mfma_busy.txt 2>&1 || true
python3 tools/pmcsum.py "k_match_i8" $OUT/pmc_MATCH > $OUT/matcher_pmc.txt 2>&1 || true
tools/_build/mfma_i8_peak >> $OUT/matcher_pmc.txt 2>&1 || true
rm -rf $OUT/stats $OUT/serial $OUT/pmc_SQ $OUT/pmc_MFMA $OUT/pmc_MATCH  # (the FETCH / WRITE csv stay for re-reductions: a few MB)
ls -la $OUT
