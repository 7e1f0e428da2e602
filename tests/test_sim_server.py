"""CPU suite: the region server (dynamic batching of concurrent single-region calls) on the wave simulator."""
import check_server


def test_sim_server_answers_every_caller_as_its_own_populate_would():
    calls, batches = check_server.check_server("sim")
    assert calls == 70


def test_sim_server_over_several_devices_gives_the_same_answers():
    """oct_phmm_server_create_multi: one queue, workers and handles per listed device (the simulator's one device listed twice)."""
    calls, batches = check_server.check_server("sim", n_threads=4, per_thread=5, seed=23, devices=[0, 0])
    assert calls == 40


def test_sim_server_over_two_distinct_devices(monkeypatch):
    """Two DIFFERENT device ordinals (OCTSIM_DEVICES=2: the simulator's devices are ordinals only): the server's per-device workers, handles and pools are told apart
    by id - what a node with several MI355X runs and a one-GPU box cannot. Same answers; both devices take calls."""
    monkeypatch.setenv("OCTSIM_DEVICES", "2")
    calls, batches = check_server.check_server("sim", n_threads=6, per_thread=5, seed=29, devices=[0, 1])
    assert calls == 60


def test_sim_server_rejects_a_device_that_does_not_exist():
    import pytest
    from backends import build_sim
    from octopus_amd import abi, engine
    with pytest.raises(engine.EngineError) as e:
        engine.Server(abi.Config.default(max_indel_error=8), lib_path=build_sim(), devices=[0, 7])     # the simulator has one device
    assert e.value.code == abi.ENODEVICE


def test_sim_server_answers_a_malformed_call_with_einval_instead_of_crashing_a_worker():
    check_server.check_server_rejects_malformed_calls("sim")


def test_sim_region_calls_bench_file_mode_answers_equal_the_oracle(tmp_path):
    """tools/region_calls_bench --file (bench.py's region-call legs on the configs[3] stream): the record format of synth.write_regions_file, one call per region
    through the region server from several threads and from one handle, the answers written back - built against the simulator's C ABI and compared with the
    oracle region by region (regions with and without a flank state, different haplotype lengths)."""
    import json
    import subprocess
    import numpy as np
    import oracle
    from backends import ROOT, build_sim
    from octopus_amd import abi, synth
    sim = build_sim()
    exe = tmp_path / "region_calls_bench_sim"
    subprocess.run(["g++", "-O1", "-std=c++17", str(ROOT / "tools" / "region_calls_bench.cpp"), "-o", str(exe), f"-I{ROOT / 'include'}", f"-L{sim.parent}", "-lphmm_sim",
                    f"-Wl,-rpath,{sim.parent}", "-lpthread"], check=True)
    regions = synth.region_stream_shard(42, 7, B=16, positions="none", cap=(14, 3)) + synth.region_stream_shard(42, 3, B=16, positions="none", cap=(9, 2), hq=True)
    regions[2]["flank"] = None
    synth.write_regions_file(tmp_path / "regions.bin", regions)
    r = subprocess.run([str(exe), "--file", str(tmp_path / "regions.bin"), "--out", str(tmp_path / "out.bin"), "1", "3"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr
    rows = [json.loads(line) for line in r.stdout.splitlines() if line.startswith("{")]
    assert all(x.get("failures", 0) == 0 for x in rows) and any(x["mode"] == "handle per thread" for x in rows)
    assert next(x for x in rows if x["mode"] == "server vs plain calls")["regions_that_differ"] == 0
    got = np.fromfile(tmp_path / "out.bin", dtype=np.float64)
    cfg = abi.Config.default(max_indel_error=16)
    want = np.concatenate([oracle.populate(cfg, synth.batch_from_regions([g]), n_threads=2)[0] for g in regions])
    assert got.shape == want.shape and np.array_equal(got, want)


def test_sim_server_contract_violation_reaches_only_its_caller():
    import check_server
    assert check_server.check_server_contract_violation_reaches_only_its_caller("sim")
