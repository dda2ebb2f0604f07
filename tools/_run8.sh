cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_fuzz_1043.txt; : > $O
for fs in 1 0; do echo "## seed 1043 wide speeds WF_LL_FAR_SKIP=$fs" >> $O; WF_LL_FAR_SKIP=$fs WF_FUZZ_SKIP=1 FUZZ_WS=2.5,26 python tests/tools/fuzz_parity.py 1000 1043 2>&1 | grep -E "^BAD|^fuzz" | cut -c1-2500 >> $O; done
grep -E "^##|^fuzz" $O; grep -c "^BAD" $O
WF_CALIBRATE=0 python tools/time_variants.py build/alt/lib_v8.so build/alt/lib_v9.so 2>&1 | grep ms/step > gpurun_out/r04_ab4.txt; cat gpurun_out/r04_ab4.txt
python -m pytest tests -m gpu -q -x > gpurun_out/r04_pytest8.txt 2>&1; tail -4 gpurun_out/r04_pytest8.txt
python bench.py --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/r04_bench8.json 2> gpurun_out/r04_bench8.err; python - <<'PY'
import json
r=json.loads(open('gpurun_out/r04_bench8.json').read().strip().split('\n')[-1])
print('headline', r['ms_per_step'], r['roofline']['kernel_ms'], r['config']['kernel'])
print('fused', r['fused_env_step']['ms_per_step'], {k:v['ms_per_step'] for k,v in r['env_level'].items() if k!='what'})
for k,v in r.get('extra',{}).items(): print(k, v['float32_only']['ms_per_step'], v['with_float64_resolve']['ms_per_step'], v['n_resolved'])
PY
