#!/usr/bin/env python3
"""Long randomised parity runs on the GPU.

  python tools/gpu_fuzz.py [seeds=40] [per_seed=50] [first_seed=1000]          tests/check_fuzz.py: small single-call scenarios (all bands, lane widths, align)
  python tools/gpu_fuzz.py shapes [n=2000] [first_seed=0] [workers=12]         tests/check_shapes.py: MULTI-REGION batches (1-64 regions, H 1-400, R 20-5,000, ragged reads,
                                                                               linked chunks, NULL / given vectors, random switches) through populate, the resident API and the
                                                                               region server from 8 threads; a sample of regions against the reference's populate
"""
import sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))

if __name__ == "__main__":
    t0 = time.time()
    if len(sys.argv) > 1 and sys.argv[1] == "shapes":
        import check_shapes
        n = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
        first = int(sys.argv[3]) if len(sys.argv) > 3 else 0
        workers = int(sys.argv[4]) if len(sys.argv) > 4 else 12
        with check_shapes.worker_pool(workers) as pool:            # before the first HIP call of this process
            st = check_shapes.check_shapes("gpu", range(first, first + n), pool=pool)
        print(f"{st['scenarios']} multi-region scenarios, seeds {first}..{first + n - 1}: {st['regions']} regions, {st['pairs']} pairs; {st['sampled_regions']} regions compared with the "
              f"reference's populate (max |delta| <= 1e-9), {st['server_calls']} region-server calls from 8 threads and {st['resident']} resident runs bit-equal to the flat run, "
              f"{st['null_vectors']} scenarios with library-made penalty vectors; switches: {st['switches']}; {time.time() - t0:.0f} s")
    else:
        import check_fuzz
        seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 40
        per = int(sys.argv[2]) if len(sys.argv) > 2 else 50
        first = int(sys.argv[3]) if len(sys.argv) > 3 else 1000
        n = 0
        for s in range(first, first + seeds):
            n += check_fuzz.check_fuzz("gpu", seed=s, n=per, tol=1e-9)
        print(f"{n} random scenarios, seeds {first}..{first + seeds - 1} (populate incl. the device mapper's positions; align on the alignable half) equal the oracle on the GPU, {time.time() - t0:.0f} s")
