# Round-3 GPU session 6: the whole -m gpu suite on the current product build + default bench line
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03f
mkdir -p $O; cd $R
( time timeout 1500 python -m pytest tests -x -q -m gpu ) > $O/tests.txt 2>&1
tail -8 $O/tests.txt
timeout 400 python bench.py > $O/bench.json 2> $O/bench.err
python -c "
import json
j=json.loads(open('$O/bench.json').read().strip().splitlines()[-1])
print(j['ms_per_step'], j['value'], j['roofline']['kernel'], j['roofline']['frac'], j['nms'], j['infer'])
print(j['cpu_baseline'])
"
echo done
