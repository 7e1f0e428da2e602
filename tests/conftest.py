import json
import sys
from pathlib import Path

import pytest

import os

os.environ.setdefault("OCT_PHMM_ENV_SWITCHES", "1")      # the suite steers code paths through OCT_PHMM_* variables: the library reads them only on this opt-in

ROOT = Path(__file__).resolve().parents[1]
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_records():
    return json.loads((ROOT / "tests" / "golden" / "pair_hmm_tests.json").read_text())["records"]
