"""CPU suite: the product's penalty-vector generator (device-free host entry of the C ABI, and the in-call generation on the simulator)."""
import pytest

import check_error_model as ce
import oracle
from backends import build_sim


@pytest.mark.skipif(not oracle.have_ref(), reason="reference build absent")
def test_product_penalty_vectors_equal_the_reference_error_models_on_the_corpus():
    assert ce.check_host_entry(build_sim()) == 2400


def test_product_penalty_vectors_equal_the_oracle_restatement():
    assert ce.check_host_entry(build_sim(), n_strings=300, reference=False) == 300


@pytest.mark.skipif(not oracle.have_ref(), reason="reference build absent")
def test_sim_populate_generates_the_vectors_on_host_threads_and_on_the_device():
    assert ce.check_populate_generates_the_vectors("sim") > 0


def test_sim_align_and_server_generate_the_vectors():
    assert ce.check_align_and_server_generate_the_vectors("sim") >= 5


def test_sim_both_device_kernels_equal_the_host_entry_on_corpus_strings():
    assert ce.check_device_kernels_on_the_corpus("sim", 150) == 150


@pytest.mark.skipif(not oracle.have_ref(), reason="reference build absent")
def test_model_files_are_read_and_applied_like_the_reference_custom_model():
    """CustomRepeatBasedIndelErrorModel (`--sequence-error-model <file>`): 33 fixed texts + 105 random ones accepted / refused like make_penalty_map, vectors equal the reference class's."""
    n_ok, n_bad = ce.check_custom_model_file(build_sim())
    assert n_ok >= 80 and n_bad >= 50


@pytest.mark.skipif(not oracle.have_ref(), reason="reference build absent")
def test_edited_model_files_are_accepted_and_refused_like_the_reference_reader():
    ok, bad = ce.check_custom_model_mutations(build_sim())
    assert ok > 300 and bad > 800


def test_sim_calls_generate_the_vectors_from_a_model_file():
    assert ce.check_custom_model_in_calls("sim") > 0
