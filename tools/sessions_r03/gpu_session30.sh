cd /root/repo; export TMPDIR=/tmp
O=gpurun_out/r03_s30; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q > $O/gpu_tests.log 2>&1; echo "pytest rc=$?"
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; echo "smoke rc=$?"
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
tail -3 $O/gpu_tests.log; tail -1 $O/smoke.log; cut -c1-300 $O/bench.json
