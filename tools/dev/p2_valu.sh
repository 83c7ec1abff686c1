# per-dispatch instruction counts of conv_p2_kernel on single layers, whole kernel and with phases ablated (build p2ablate: YS_DBG bits)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/p2valu; mkdir -p $O
for dbg in 0 4 6 7; do
  YS_DBG=$dbg timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD --output-format csv -d $O/dbg$dbg -o p -- python $R/tools/dev/p2_layers.py $R/build/libyolosharp_hip_p2abl.so > $O/run$dbg.log 2>&1
  f=$(find $O/dbg$dbg -name '*counter_collection.csv' | head -1)
  python - $f $dbg <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
d = collections.OrderedDict()
for r in rows:
    if 'conv_p2_kernel' not in r['Kernel_Name']: continue
    k = r['Dispatch_Id']
    d.setdefault(k, {'grid': r.get('Grid_Size'), 'lds': r.get('LDS_Block_Size')})[r['Counter_Name']] = float(r['Counter_Value'])
print('YS_DBG', sys.argv[2])
for k, v in d.items():
    w = v.get('SQ_WAVES', 0) or 1
    print('  disp %s grid %s lds %s waves %d | per wave: VALU %.0f SALU %.0f LDS %.0f VMEM_RD %.0f' % (k, v['grid'], v['lds'], w, v.get('SQ_INSTS_VALU', 0) / w, v.get('SQ_INSTS_SALU', 0) / w, v.get('SQ_INSTS_LDS', 0) / w, v.get('SQ_INSTS_VMEM_RD', 0) / w))
PY
done
