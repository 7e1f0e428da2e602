"""CPU suite: the region server (dynamic batching of concurrent single-region calls) on the wave simulator."""
import check_server


def test_sim_server_answers_every_caller_as_its_own_populate_would():
    calls, batches = check_server.check_server("sim")
    assert calls == 70
