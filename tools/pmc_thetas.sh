# issue / LDS / memory counters of the orientation kernel with 1 and 4 lanes per key point (serial stage benchmark)
export TMPDIR=/tmp SSRLCV_DEV_BUILD=1 SSRLCV_SIFT_SERIAL=1
OUT=gpurun_out/r06_thetas
mkdir -p $OUT
for lanes in 1 4; do
  export SSRLCV_THETAS_LANES=$lanes
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS -d $OUT/pmcA$lanes --output-format csv -- python3 tools/bench_sift_stages.py --size 4096 --scene --stages 6 > $OUT/pmcA$lanes.log 2>&1
  python3 tools/pmcsum.py k_thetas $OUT/pmcA$lanes
  rocprofv3 --pmc SQ_WAVES SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_WAIT_ANY SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM -d $OUT/pmcB$lanes --output-format csv -- python3 tools/bench_sift_stages.py --size 4096 --scene --stages 6 > $OUT/pmcB$lanes.log 2>&1
  python3 tools/pmcsum.py k_thetas $OUT/pmcB$lanes || tail -3 $OUT/pmcB$lanes.log
  rocprofv3 --pmc TA_BUSY_avr TA_TA_BUSY_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum -d $OUT/pmcC$lanes --output-format csv -- python3 tools/bench_sift_stages.py --size 4096 --scene --stages 6 > $OUT/pmcC$lanes.log 2>&1
  python3 tools/pmcsum.py k_thetas $OUT/pmcC$lanes || tail -3 $OUT/pmcC$lanes.log
done
rm -rf $OUT/pmc[ABC][14]
