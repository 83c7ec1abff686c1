cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests/test_production_routing.py -m gpu -q --no-header -p no:cacheprovider -k "b8 or b64 or other_configs" 2>&1 | tail -5
grep -h "inf_" gpurun_out/prod_routing*_summary.txt | cut -c1-160
for f in gpurun_out/prod_routing*_summary.txt; do echo $f; grep "inf_" $f | cut -c1-160; done
