# Round 5, GPU sessions 11, 13 and 14: the final tree - GPU suite, shape-fuzz scenarios with new seeds (2,000 in session 11; 1,000 in 13 after the caller-facts step; 300 in 14 after the slices-aside default went back), smoke, bench line
cd /root/repo; export TMPDIR=/tmp OCT_PHMM_ENV_SWITCHES=1
O=gpurun_out/r05_s14; mkdir -p $O
python -c "from octopus_amd import engine; print(engine.kernel_source_sha())" > $O/kernel_source_sha
timeout -k 5 900 python -m pytest tests -x -q -m gpu --durations=4 > $O/pytest_gpu.log 2>&1; echo "pytest_gpu rc=$?" >> $O/rc.log
timeout -k 5 900 python tools/gpu_fuzz.py shapes 300 50000 12 > $O/gpu_fuzz_shapes.log 2>&1; echo "fuzz shapes rc=$?" >> $O/rc.log
timeout -k 5 120 python __graft_entry__.py smoke > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/rc.log
timeout -k 5 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" >> $O/rc.log
cat $O/rc.log; tail -3 $O/pytest_gpu.log; tail -1 $O/gpu_fuzz_shapes.log | cut -c1-500; cut -c1-300 $O/bench.json | tail -1
