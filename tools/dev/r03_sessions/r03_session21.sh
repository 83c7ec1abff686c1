# Round-3 GPU session 21: why does the default bench line read 0.3 ms/step above the A/B runs?  Same build, flags varied.
cd $GRAFT_REPO_ROOT; O=gpurun_out/r03u; mkdir -p $O
for t in "default:" "nocpu:--no-cpu-baseline" "noinfer:--no-infer --no-nms" "bare50:--no-cpu-baseline --no-nms --no-infer" "bare30:--no-cpu-baseline --no-nms --no-infer --steps 30 --warmup 8" "default2:"; do
  tag=${t%%:*}; fl=${t#*:}
  timeout 400 python bench.py $fl > $O/$tag.json 2> $O/$tag.err
  python -c "
import json; j=json.loads(open('$O/$tag.json').read().strip().splitlines()[-1]); print('%-10s %.3f ms/step  %.1f img/s' % ('$tag', j['ms_per_step'], j['value']))"
done
