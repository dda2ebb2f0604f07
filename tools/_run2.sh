cd $GRAFT_REPO_ROOT
python tools/time_variants.py build/alt/lib_r3.so build/alt/lib_v3a.so build/alt/lib_v3b.so build/alt/lib_v6.so 2>&1 | grep ms/step > gpurun_out/r04_ab1.txt; cat gpurun_out/r04_ab1.txt
python -m pytest tests/test_hip_parity.py -m gpu -q -x -k "far_skip or one_block or full_size or reference_kat or parity_random" > gpurun_out/r04_pytest2.txt 2>&1; tail -8 gpurun_out/r04_pytest2.txt
bash tools/quick_pmc.sh r04_v6
python bench.py --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/r04_bench2.json 2> gpurun_out/r04_bench2.err; python - <<'PY'
import json
r=json.loads(open('gpurun_out/r04_bench2.json').read().strip().split('\n')[-1])
print('headline', r['ms_per_step'], r['roofline']['kernel_ms'], r['config']['kernel'])
for k,v in r.get('extra',{}).items(): print(k, v['float32_only']['ms_per_step'], v['with_float64_resolve']['ms_per_step'], v['n_resolved'])
PY
