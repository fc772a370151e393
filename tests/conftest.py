import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(REPO, "tests", "golden")


KNOB_DEFAULTS = {"count_lanes": "", "count_sort": "1", "force_exact": "0",
                 "proposal": "", "cx_cfg": "", "rerank": "", "score_batch": "0", "tail_aside": "1"}


@pytest.fixture(autouse=True)
def _reset_context_knobs():
    """GPU tests switch library paths with Context.set_option on the shared default context: put every knob back."""
    yield
    mod = sys.modules.get("phamers_amd._lib")
    if mod is None:
        return
    for ctx in list(getattr(mod, "_contexts", {}).values()):
        if getattr(ctx, "handle", None):
            for k, v in KNOB_DEFAULTS.items():
                ctx.set_option(k, v)
