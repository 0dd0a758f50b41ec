"""GPU parity: the static-shape engine (what bench.py times) computes the same training
steps as the module path (GraphSAGE.forward_batch + F.cross_entropy + autograd +
torch.optim.Adam), which test_gpu_model.py ties to the CPU oracle."""
import copy

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("aggregator", ["mean", "sum"])
def test_engine_matches_module_path(dev, aggregator):
    from graphsage_amd.datasets import planted_partition
    from graphsage_amd.engine import SageTrainEngine
    from graphsage_amd.models import GraphSAGE
    from graphsage_amd.samplers import NeighborSampler, UniformLoader
    from graphsage_amd.samplers.uniform_loader import device_graph
    data = planted_partition(3000, 60000, 40, 6, seed=2)
    B = 64
    loader = UniformLoader(data, num_neighbors=[5, 3], input_nodes=torch.arange(4 * B), batch_size=B, seed=7)
    torch.manual_seed(0)
    model_a = GraphSAGE(40, 32, aggregator, 2, 6).to(dev)
    model_b = copy.deepcopy(model_a)
    graph = device_graph(data)
    engine = SageTrainEngine(model_b, NeighborSampler(data, [5, 3], seed=7), graph.features(data),
                             graph.labels(data), B, lr=1e-2)
    opt = torch.optim.Adam(model_a.parameters(), lr=1e-2)
    for b, batch in enumerate(loader):
        opt.zero_grad()
        loss_a = F.cross_entropy(model_a.forward_batch(batch), batch.y[:B])
        loss_a.backward()
        grads_a = [p.grad.clone() for p in model_a.parameters()]
        opt.step()
        seeds = torch.arange(b * B, (b + 1) * B, device=dev)
        loss_b = engine.step(seeds, counter=b)
        torch.testing.assert_close(loss_b[0], loss_a.detach(), rtol=1e-5, atol=1e-6)
        for ga, pb in zip(grads_a, model_b.parameters()):
            torch.testing.assert_close(engine.gviews[pb], ga, rtol=1e-3, atol=1e-6)
        for pa, pb in zip(model_a.parameters(), model_b.parameters()):
            torch.testing.assert_close(pb.data, pa.data, rtol=1e-4, atol=1e-5)
    assert int(engine.step_dev) == 4
