# usage: bash tools/ab_env.sh "<label>=<VAR=1 VAR2=x>" ...   (developer build; bench.py's SIFT loop only, three alternations)
for rep in 1 2 3; do
  for spec in "$@"; do
    label=${spec%%=*}; vars=${spec#*=}
    [ "$vars" = "$spec" ] && vars=""
    env SSRLCV_DEV_BUILD=1 $vars python bench.py --no-cpu-baseline --no-class-api --no-nview --no-matcher --no-pushbroom --steps 20 --warmup 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$label', 'step', round(d['ms_per_step'],3), 'stage', round(d['roofline']['ms_per_image'],4), 'alone', round(d['roofline']['stage_alone']['ms_per_image'],4), 'describe', round(d['describe']['ms_per_image'],4))
"
  done
done
