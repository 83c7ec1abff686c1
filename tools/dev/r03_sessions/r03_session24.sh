# Round-3 GPU session 24: what does bench.py --force-dist add to the step? (HIP API call counts of both paths)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03w; mkdir -p $O
export MASTER_ADDR=127.0.0.1 MASTER_PORT=29555 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1
for tag in dist local; do
  fl=""; [ $tag = dist ] && fl="--force-dist"
  timeout 400 rocprofv3 --hip-runtime-trace --output-format csv -d $O/h$tag -o h -- python $R/bench.py $fl --steps 10 --warmup 3 --no-cpu-baseline --no-nms --no-infer > $O/h$tag.log 2>&1
  f=$(find $O/h$tag -name '*hip_api_trace.csv' | head -1)
  python - $f $tag <<'PY'
import csv, sys, collections
c = collections.Counter(r['Function'] for r in csv.DictReader(open(sys.argv[1])))
print(sys.argv[2], {k: v for k, v in c.most_common(14)})
PY
  rm -rf $O/h$tag
done
