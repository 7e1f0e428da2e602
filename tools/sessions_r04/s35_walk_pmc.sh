cd /root/repo; export TMPDIR=/tmp OCT_PHMM_ENV_SWITCHES=1
O=gpurun_out/r04_s35; mkdir -p $O
(cd /tmp && timeout -k 5 200 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES --output-format csv -d /root/repo/$O/pmc -o p -- python /root/repo/tools/long_read_legs.py ccs256x12 > /root/repo/$O/pmc.json 2> /root/repo/$O/pmc.err)
python tools/summarize_pmc.py $O/pmc --min-grid 1000 --out $O/walk_pmc > /dev/null 2>&1
grep "k_walk_rows\|k_dp_rows\|kernel" $O/walk_pmc.md | cut -c1-260
