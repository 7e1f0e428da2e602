# Round 6, session 31: two / three steps of the headline batch (and of the streams) in flight on as many handles, against the serial run-wait loop of bench.py.
cd /root/repo; export TMPDIR=/tmp
for W in 100kx128 stream stream-hq; do timeout -k 5 600 python tools/two_in_flight.py $W 12 2>&1 | tail -1; done
