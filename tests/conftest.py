import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def zymo():
    from savont_amd.fastx import read_fastx
    seq, qual, off, ids = read_fastx(os.path.join(GOLDEN, "ont_zymo_1000.trimmed.fq.gz"))
    return dict(seq=seq, qual=qual, off=off, ids=ids)


@pytest.fixture(scope="session")
def zymo2():
    from savont_amd.fastx import read_fastx
    seq, qual, off, ids = read_fastx(os.path.join(GOLDEN, "ont_zymo_1000_2.trimmed.fq.gz"))
    return dict(seq=seq, qual=qual, off=off, ids=ids)


@pytest.fixture(scope="session")
def zymo_asvs():
    from savont_amd.fastx import read_fastx
    seq, _, off, ids = read_fastx(os.path.join(GOLDEN, "zymo_ref_asvs.fa.gz"))
    return dict(seq=seq, off=off, ids=ids)


@pytest.fixture(scope="session")
def dev():
    from savont_amd import hip
    d = hip.Device(0)   # raises loudly when the extension or the GPU is missing
    yield d
    d.close()


def rc_flags_of(ids):
    return np.array([1 if (i.split() and i.split()[-1] == "rc") else 0 for i in ids], np.uint8)
