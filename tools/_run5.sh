cd $GRAFT_REPO_ROOT
WF_CALIBRATE=0 python tools/time_variants.py build/alt/lib_r3.so build/alt/lib_v7.so build/alt/lib_v8.so 2>&1 | grep ms/step > gpurun_out/r04_ab3.txt; cat gpurun_out/r04_ab3.txt
python -m pytest tests/test_hip_parity.py tests/test_bench_gpu.py tests/test_resolve_gpu.py -m gpu -q -x > gpurun_out/r04_pytest5.txt 2>&1; tail -6 gpurun_out/r04_pytest5.txt
(python tests/tools/band_study.py HornsRev1_ 6 65536 reset; python tests/tools/band_study.py HornsRev1_ 4 65536 wide; python tests/tools/band_study.py fuzz 60 8192 wide; python tests/tools/band_study.py fuzz 40 8192 reset) 2>&1 | grep -v amdgpu.ids > gpurun_out/r04_band_study2.txt; cat gpurun_out/r04_band_study2.txt | cut -c1-400
python bench.py --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/r04_bench5.json 2> gpurun_out/r04_bench5.err; python - <<'PY'
import json
r=json.loads(open('gpurun_out/r04_bench5.json').read().strip().split('\n')[-1])
print('headline', r['ms_per_step'], r['roofline']['kernel_ms'], r['config']['kernel'])
for k,v in r.get('extra',{}).items(): print(k, v['float32_only']['ms_per_step'], v['with_float64_resolve']['ms_per_step'], v['n_resolved'])
PY
