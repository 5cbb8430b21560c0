import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLD = os.path.join(ROOT, "tests", "golden")
# the library honours its result-changing test hooks (csrc/switches.cpp: SW_HOOK) only under this gate; the suite uses several
# (tests/test_gpu_switches.py removes the gate and checks that every one of them is then ignored)
os.environ.setdefault("ITSX_TEST_HOOKS", "1")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


@pytest.fixture(scope="session")
def gold():
    return GOLD


@pytest.fixture(scope="session")
def fixture_reads():
    import gzip
    names, seqs = [], []
    with gzip.open(os.path.join(GOLD, "fixture_reads.fa.gz"), "rt") as f:
        for line in f:
            if line[0] == ">":
                names.append(line[1:].strip())
            else:
                seqs.append(line.strip())
    return names, seqs


@pytest.fixture(scope="session")
def mini_hmm_text():
    with open(os.path.join(GOLD, "mini.hmm")) as f:
        return f.read()


@pytest.fixture(scope="session")
def t_hmm_text():
    import gzip
    with gzip.open(os.path.join(GOLD, "T.hmm.gz"), "rt") as f:
        return f.read()


@pytest.fixture(scope="session")
def all_its2_hmm_text():
    """--taxa All --region ITS2 as create_runtime_hmm writes it (814 profiles; BASELINE configs[3])"""
    import gzip
    with gzip.open(os.path.join(GOLD, "all_its2.hmm.gz"), "rt") as f:
        return f.read()


@pytest.fixture(scope="session")
def engine():
    from itsxpress_amd import Engine
    e = Engine(0)
    yield e
    e.close()
