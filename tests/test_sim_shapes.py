"""CPU suite: the multi-region shape fuzz (tests/check_shapes.py) at toy sizes on the wave simulator - the generator, the three entry points and the worker pool that
feeds the GPU run (tools/gpu_fuzz.py shapes; tests/test_gpu_parity.py runs a 100-scenario slice of the full-size generator on the device)."""
import check_shapes


def test_sim_multi_region_shapes_through_populate_resident_and_server():
    st = check_shapes.check_shapes("sim", range(100, 110))
    assert st["scenarios"] == 10 and st["sampled_regions"] >= 10 and st["server_calls"] > 0


def test_scenarios_come_out_of_spawned_workers_in_order():
    with check_shapes.worker_pool(2) as pool:
        st = check_shapes.check_shapes("sim", range(200, 204), pool=pool)
    assert st["scenarios"] == 4


def test_full_size_generator_covers_the_shapes_it_promises():
    """No device: the full-size generator's regions span what the docstring says (haplotype counts 1..48 densely and beyond, many-region batches, linked chunks, ragged reads)."""
    haps, n_regions, linked, ragged, eleven = set(), set(), 0, 0, 0
    for seed in range(60):
        scn = check_shapes.make_scenario(seed, "gpu")
        n_regions.add(len(scn["regions"]))
        for g in scn["regions"]:
            haps.add(len(g["haps"])); linked += g.get("row_off") is not None; ragged += g.get("read_len") is not None
    assert len([h for h in haps if h <= 48]) >= 40 and max(haps) >= 150 and max(n_regions) >= 24 and linked > 0 and ragged > 0


def test_generator_makes_a_scenario_for_every_seed():
    """The toy-scale generator drew `read_len` from an empty range for 3 % of the seeds (a read shorter than band + 4) until round 5's last simulator campaign met one."""
    for seed in range(1500):
        check_shapes.make_scenario(seed, "sim")
    for seed in range(6000, 6120):
        check_shapes.make_scenario(seed, "gpu")
