cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q -x > gpurun_out/r04_pytest6.txt 2>&1; tail -5 gpurun_out/r04_pytest6.txt
bash tools/record_profiles_r04.sh r04_v8 > /dev/null 2>&1
python - <<'PY'
import json
for f in ("r04_v8_bench.json","r04_v8_bench_cfg5.json","r04_v8_bench_cfg3b.json","r04_v8_bench_cfg2.json"):
    try:
        r=json.loads(open('gpurun_out/'+f).read().strip().split('\n')[-1])
        print(f, r['value'], r['ms_per_step'], r['roofline']['kernel_ms'], r['config']['kernel'])
        for k,v in r.get('extra',{}).items(): print('   ',k, v['float32_only']['ms_per_step'], v['with_float64_resolve']['ms_per_step'], v['n_resolved'])
    except Exception as e: print(f, 'ERR', e)
PY
cat gpurun_out/r04_v8_resolve_cost.txt
(python tools/batch_sweep_fine.py HornsRev1_ HornsRev2_; SWEEP_STEP=8192 python tools/batch_sweep_fine.py Ormonde_ WMR_) 2>&1 | grep -v amdgpu.ids > gpurun_out/r04_v8_batch_sweep_fine.txt; grep "^#" gpurun_out/r04_v8_batch_sweep_fine.txt
