cd $GRAFT_REPO_ROOT
timeout 1700 python -m pytest tests/ -x -q -m gpu 2>&1 | tail -5 > gpurun_out/t26.log
timeout 600 python bench.py > gpurun_out/bench_default.log 2> gpurun_out/bench_default.err
tail -1 gpurun_out/bench_default.log > gpurun_out/bench_default.json
cat gpurun_out/t26.log; tail -c 600 gpurun_out/bench_default.json
