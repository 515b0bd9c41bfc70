import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "gaussian-splatting_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # the oracle issues many small tensor ops; on a 256-core host torch's default intra-op pool is >10x slower
    try:
        import torch
        torch.set_num_threads(min(16, os.cpu_count() or 1))
    except Exception:
        pass


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def _is_heavy_candidate_test(nodeid: str) -> bool:
    """The long CPU runs of kernel forms that are OFF in the product (macros GSR_FWD_TL_DECAY / GSR_FWD_COMPACT / GSR_BWD_DPP_FUSE, and the LSD sort's
    rs_scatter under GSR_MATCH_BITOP3): ~100 s of the suite spent on code the library does not contain.  They run with GSR_TEST_CANDIDATES=1 (do that
    before a candidate's GPU bring-up, tools/gpu_r5_candidates.sh); the cheap pin of the digit-match candidate on the product's binning chain
    (test_simt_chain_cpu.py::test_candidate_digit_matching_leaves_the_bins_alone) and the ISA pins (test_isa_audit_cpu.py) always run."""
    return ("test_simt_forward_cpu.py::test_candidate_form" in nodeid) or ("test_simt_rows_cpu.py::" in nodeid and "[bitop3" in nodeid)


def pytest_collection_modifyitems(config, items):
    if os.environ.get("GSR_TEST_CANDIDATES", "0") != "1":
        skip_cand = pytest.mark.skip(reason="candidate kernel form that is off in the product: set GSR_TEST_CANDIDATES=1 to run it")
        for it in items:
            if _is_heavy_candidate_test(it.nodeid):
                it.add_marker(skip_cand)
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason="no HIP device in this container")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)
