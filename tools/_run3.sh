cd $GRAFT_REPO_ROOT
python tools/time_variants.py build/alt/lib_r3.so build/alt/lib_v6.so build/alt/lib_v7.so 2>&1 | grep ms/step > gpurun_out/r04_ab2.txt; cat gpurun_out/r04_ab2.txt
python -m pytest tests/test_hip_parity.py -m gpu -q -x -k "far_skip or one_block or full_size or reference_kat or parity_random" > gpurun_out/r04_pytest3.txt 2>&1; tail -5 gpurun_out/r04_pytest3.txt
bash tools/quick_pmc.sh r04_v7 | grep -E "INSTS_VALU\"|WAIT_ANY\"|WAVE_CYCLES\"|RDREQ_128B_sum\"|ACTIVE_INST_VALU\""
for fs in 1 0; do echo "per-env wind WF_LL_FAR_SKIP=$fs"; WF_LL_FAR_SKIP=$fs python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-env-leg --per-env-wind 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(r['ms_per_step'], r['roofline']['kernel_ms'], r['config']['kernel'])"; done
python bench.py --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/r04_bench3.json 2> gpurun_out/r04_bench3.err; python - <<'PY'
import json
r=json.loads(open('gpurun_out/r04_bench3.json').read().strip().split('\n')[-1])
print('headline', r['ms_per_step'], r['roofline']['kernel_ms'], r['config']['kernel'])
for k,v in r.get('extra',{}).items(): print(k, v['float32_only']['ms_per_step'], v['with_float64_resolve']['ms_per_step'], v['n_resolved'])
PY
