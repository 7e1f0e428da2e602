"""tools/shape_replay.py: batch shapes of a real run (a CSV, or the reference's own debug line "Calculating likelihoods for N haplotypes", caller.cpp:1169) -> the region
files tools/region_calls_bench and bench.py's region-call legs read."""
import json
import struct
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]


def run(args):
    r = subprocess.run([sys.executable, str(ROOT / "tools" / "shape_replay.py")] + args, capture_output=True, text=True, check=True)
    return json.loads(r.stdout.strip().splitlines()[-1])


def header(path):
    b = Path(path).read_bytes()
    assert b[:4] == b"OCTR"
    return struct.unpack("<I", b[4:8])[0], struct.unpack("<5I2Q", b[8:44])


def test_shapes_from_a_csv_become_a_regions_file(tmp_path):
    csv = tmp_path / "shapes.csv"
    csv.write_text("R,H,T,Lh,lhs,rhs\n60,4,150,320,40,40\n25,2,150,300,,\n90,5,100,280,20,60\n")
    out = tmp_path / "regions.bin"
    s = run(["--csv", str(csv), "--out", str(out), "--classify-sample", "2"])
    n, (R, H, has_flank, lhs, rhs, read_bases, hap_bases) = header(out)
    assert n == 3 and s["regions"] == 3 and s["pairs"] == 60 * 4 + 25 * 2 + 90 * 5
    assert (R, H, has_flank, lhs, rhs, read_bases, hap_bases) == (60, 4, 1, 40, 40, 60 * 150, 4 * 320)
    c = s["classes_of_the_first_regions"]
    assert c["n_pairs"] == 60 * 4 + 25 * 2 and c["n_dp_score_only"] + c["n_dp_traceback"] + c["n_fast_path"] > 0


def test_shapes_from_the_reference_debug_line(tmp_path):
    log = tmp_path / "octopus_debug.log"
    log.write_text("[debug] Calculating likelihoods for 7 haplotypes\n  active candidates in chr20:1000200-1000420\n[debug] noise\n"
                   "[debug] Calculating likelihoods for 2 haplotypes\n")
    out = tmp_path / "regions.bin"
    s = run(["--octopus-debug-log", str(log), "--out", str(out), "--classify-sample", "0", "--reads-per-haplotype-median", "40"])
    n, (R, H, has_flank, lhs, rhs, read_bases, hap_bases) = header(out)
    assert n == 2 and H == 7 and hap_bases == 7 * (220 + 2 * 31) and read_bases == R * 150 and s["quantiles_5_50_95_max"]["H"][-1] == 7
