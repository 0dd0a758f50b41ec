import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def dev():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


REDDIT = dict(N=232965, E=114615892, FEAT=602, C=41)


@pytest.fixture(scope="session")
def reddit(dev):
    """The Reddit-SHAPED synthetic graph of bench.py (BASELINE.json configs[2]/[3]) with its CSC
    and padded feature table resident on the device; built once per test session (~6 GB)."""
    import torch
    from graphsage_amd.datasets import reddit_like_device
    from graphsage_amd.samplers.uniform_loader import device_graph
    data = reddit_like_device(REDDIT["N"], REDDIT["E"], REDDIT["FEAT"], REDDIT["C"], seed=0, device=dev)
    graph = device_graph(data)
    yield data, graph
    del data, graph
    torch.cuda.empty_cache()
