# timeline of the last fused image of bench.py's SIFT loop (release library): usage: bash tools/timeline_now.sh
export TMPDIR=/tmp
OUT=gpurun_out/r06_tl
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace -d $OUT/bench -o bench --output-format rocpd -- python3 bench.py --steps 4 --warmup 2 --no-nview --no-matcher --no-class-api --no-cpu-baseline --no-pushbroom > $OUT/bench.log 2>&1
python3 tools/timeline.py $(find $OUT/bench -name "*.db" | head -1) 260 15 > $OUT/timeline.txt
rm -rf $OUT/bench
