cd $GRAFT_REPO_ROOT
python tools/ll_stamps.py build/alt/lib_stamp.so 2>&1 | grep -v amdgpu.ids > gpurun_out/r04_v7_phase_stamps.txt; cat gpurun_out/r04_v7_phase_stamps.txt
python -m pytest tests -m gpu -q -x > gpurun_out/r04_pytest4.txt 2>&1; tail -6 gpurun_out/r04_pytest4.txt
(python tests/tools/band_study.py HornsRev1_ 6 65536 reset; python tests/tools/band_study.py HornsRev2_ 4 65536 shared; python tests/tools/band_study.py HornsRev2_ 3 65536 reset) 2>&1 | grep -v amdgpu.ids > gpurun_out/r04_band_study.txt; cat gpurun_out/r04_band_study.txt
O=gpurun_out/r04_fuzz_skip2.txt; : > $O
for seed in 1041 1042; do echo "## WF_FUZZ_SKIP=1 seed $seed" >> $O; WF_FUZZ_SKIP=1 python tests/tools/fuzz_parity.py 1500 $seed 2>&1 | grep -E "^BAD|^fuzz" | cut -c1-1500 >> $O; done
echo "## WF_FUZZ_SKIP=1 WF_FUZZ_RESOLVE=1 seed 1044" >> $O; WF_FUZZ_SKIP=1 WF_FUZZ_RESOLVE=1 python tests/tools/fuzz_parity.py 1000 1044 2>&1 | grep -E "^BAD|^fuzz" | cut -c1-1500 >> $O
grep -E "^##|^fuzz" $O
