# The minimum a changed kernel source needs before it can ship (~3 GPU-minutes): the GPU parity suite, the four separate counter passes of the 100k x 128 bench
# (bench.py's roofline block cites them and checks their source stamp) and one bench line. Every command under `timeout -k`: a python stuck in a HIP wait
# ignores SIGTERM, and a call that runs into gpurun's own limit costs its whole budget (round 3, gpu_session37).
#   gpurun --timeout 420 -- 'bash tools/profile_restamp.sh <tag>'      then, here:
#   python tools/summarize_pmc.py gpurun_out/<tag>/pmc_* --out profiles/<round>_pmc_summary --sha $(cat gpurun_out/<tag>/kernel_source_sha)
cd /root/repo; export TMPDIR=/tmp OCT_PHMM_ENV_SWITCHES=1
O=gpurun_out/${1:-restamp}; mkdir -p $O
python -c "from octopus_amd import engine; print(engine.kernel_source_sha())" > $O/kernel_source_sha
timeout -k 5 120 python -m pytest tests/test_gpu_parity.py -x -q > $O/pytest_parity.log 2>&1; echo "parity rc=$?" >> $O/rc.log
export OCT_PHMM_SLICES=1
P="--no-small-batch --no-cpu-baseline --no-extras --steps 2 --warmup 1"
for C in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE" FETCH_SIZE WRITE_SIZE; do
  D=pmc_$(echo $C | cut -d' ' -f1)
  (cd /tmp && timeout -k 5 60 rocprofv3 --pmc $C --output-format csv -d /root/repo/$O/$D -o p -- python /root/repo/bench.py $P > /root/repo/$O/$D.json 2> /root/repo/$O/$D.err); echo "$D rc=$?" >> $O/rc.log
done
unset OCT_PHMM_SLICES OCT_PHMM_ENV_SWITCHES
timeout -k 5 120 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" >> $O/rc.log
cat $O/rc.log; tail -1 $O/pytest_parity.log; cut -c1-200 $O/bench.json
