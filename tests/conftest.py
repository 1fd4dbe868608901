import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import gaz_oracle
    gaz_oracle.build()
    return gaz_oracle


GOLDEN = os.path.join(ROOT, "tests", "golden")


def oracle_many(fn, arg_list, threads=8):
    """Play many oracle games concurrently (the C library holds no per-game global state and ctypes releases the GIL): the oracle
    is the slow side of the large parity tests."""
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=threads) as ex:
        return list(ex.map(lambda a: fn(*a[0], **a[1]), arg_list))
