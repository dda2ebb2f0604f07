cd $GRAFT_REPO_ROOT
bash tools/run_pmc.sh r03a > /dev/null 2>&1
python3 tools/parse_pmc.py r03a "wf_step_ll_kernel<2, 2" > gpurun_out/r03a_pmc_cfg4.json
cat gpurun_out/r03a_pmc_cfg4.json
ls gpurun_out/pmc_r03a/
tail -3 gpurun_out/pmc_r03a/tcc_rd.err
