cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q --maxfail=15 -x -k "not eight_rank" > gpurun_out/r04_pytest1.txt 2>&1; tail -15 gpurun_out/r04_pytest1.txt
python bench.py --steps 30 --warmup 5 > gpurun_out/r04_bench1.json 2> gpurun_out/r04_bench1.err; tail -c 1500 gpurun_out/r04_bench1.json; tail -3 gpurun_out/r04_bench1.err
bash tools/fuzz_skip_ab.sh r04_fuzz_skip_ab.txt 1000
