# Round 5, GPU session 16 (the last): smoke + the GPU suite on the final tree (the model-file test is new; the product's host code changed, the kernel sources did not)
cd /root/repo; export TMPDIR=/tmp OCT_PHMM_ENV_SWITCHES=1
O=gpurun_out/r05_s16; mkdir -p $O
python -c "from octopus_amd import engine; print(engine.kernel_source_sha())" > $O/kernel_source_sha
timeout -k 5 60 python __graft_entry__.py smoke > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/rc.log
timeout -k 5 280 python -m pytest tests -x -q -m gpu --durations=4 > $O/pytest_gpu.log 2>&1; echo "pytest_gpu rc=$?" >> $O/rc.log
cat $O/rc.log $O/kernel_source_sha; tail -3 $O/smoke.log; tail -7 $O/pytest_gpu.log
