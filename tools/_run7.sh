cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q -x > gpurun_out/r04_pytest7.txt 2>&1; tail -5 gpurun_out/r04_pytest7.txt
(python3 tools/resolve_cost.py HornsRev1_ 65536; python3 tools/resolve_cost.py HornsRev2_ 131072) 2>&1 | grep -v amdgpu.ids > gpurun_out/r04_v9_resolve_cost.txt; cat gpurun_out/r04_v9_resolve_cost.txt
(python tools/batch_sweep_fine.py HornsRev1_ HornsRev2_; SWEEP_STEP=8192 python tools/batch_sweep_fine.py Ormonde_ WMR_) 2>&1 | grep -v amdgpu.ids > gpurun_out/r04_v9_batch_sweep_fine.txt; grep "^#" gpurun_out/r04_v9_batch_sweep_fine.txt
bash tools/fuzz_campaign5.sh r04_fuzz_final.txt
