"""CPU suite: the region server (dynamic batching of concurrent single-region calls) on the wave simulator."""
import check_server


def test_sim_server_answers_every_caller_as_its_own_populate_would():
    calls, batches = check_server.check_server("sim")
    assert calls == 70


def test_sim_server_over_several_devices_gives_the_same_answers():
    """oct_phmm_server_create_multi: one queue, workers and handles per listed device (the simulator's one device listed twice)."""
    calls, batches = check_server.check_server("sim", n_threads=4, per_thread=5, seed=23, devices=[0, 0])
    assert calls == 40


def test_sim_server_rejects_a_device_that_does_not_exist():
    import pytest
    from backends import build_sim
    from octopus_amd import abi, engine
    with pytest.raises(engine.EngineError) as e:
        engine.Server(abi.Config.default(max_indel_error=8), lib_path=build_sim(), devices=[0, 7])     # the simulator has one device
    assert e.value.code == abi.ENODEVICE


def test_sim_server_answers_a_malformed_call_with_einval_instead_of_crashing_a_worker():
    check_server.check_server_rejects_malformed_calls("sim")
