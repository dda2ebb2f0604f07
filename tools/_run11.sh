cd $GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -3
bash tools/record_profiles_r04.sh r04_final > /dev/null 2>&1
python - <<'PY'
import json
for f in ("r04_final_bench.json","r04_final_bench_cfg5.json","r04_final_bench_cfg3b.json","r04_final_bench_cfg2.json"):
    try:
        r=json.loads(open('gpurun_out/'+f).read().strip().split('\n')[-1])
        print(f, r['value'], r['ms_per_step'], r['roofline']['kernel_ms'], r['config']['kernel'])
        for k,v in r.get('extra',{}).items(): print('   ',k, v['float32_only']['ms_per_step'], v['with_float64_resolve']['ms_per_step'], v['n_resolved'])
    except Exception as e: print(f, 'ERR', e)
PY
cat gpurun_out/r04_final_pmc.json | grep -v "_n\""
head -4 gpurun_out/r04_final_kernel_stats_bench_cfg4.csv
bash tools/fuzz_campaign5.sh r04_fuzz_final2.txt
