cd /root/repo; export TMPDIR=/tmp
O=gpurun_out/r03_s39; mkdir -p $O
timeout -k 5 200 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
cut -c1-200 $O/bench.json
