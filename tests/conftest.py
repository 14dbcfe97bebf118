import json
import os
import sys

import pytest

# the library's fault-injection hook (RRX_debug_fail_alloc) is inert unless the process says it is a test process; the
# switch is read once, at init_ratelib, so it must be in the environment before the library is loaded
os.environ.setdefault("RSMP_TEST_HOOKS", "1")

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (HERE, ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def facts():
    with open(os.path.join(HERE, "golden", "survey_probe_facts.json")) as f:
        return json.load(f)
