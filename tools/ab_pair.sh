mkdir -p gpurun_out/r06_ab
for rep in 1 2 3; do
  for v in pair nopair; do
    if [ $v = nopair ]; then export SSRLCV_NO_GAUSS_PAIR=1; else unset SSRLCV_NO_GAUSS_PAIR; fi
    SSRLCV_DEV_BUILD=1 python bench.py --no-cpu-baseline --no-class-api --no-nview --no-matcher --no-pushbroom --steps 20 --warmup 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$v', round(d['ms_per_step'],3), round(d['roofline']['ms_per_image'],4), round(d['roofline']['stage_alone']['ms_per_image'],4), round(d['describe']['ms_per_image'],4))
"
  done
done
