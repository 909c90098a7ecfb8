cd $GRAFT_REPO_ROOT
O=gpurun_out/r4k; mkdir -p $O
timeout 600 python -m pytest "tests/test_dp_gpu.py::test_two_rank_step_on_one_gpu" -x -q -s > $O/dp.log 2>&1; echo "dp rc=$?" > $O/rc.txt
cat $O/rc.txt; tail -60 $O/dp.log | cut -c1-250
