"""GraphSAGE model with the reference's API (reference graphsage/models/graphsage.py:19-68):
a stack of SAGE layers sharing one edge_index, ReLU between layers, every layer's output
L2-normalised (the logits too), `inference(loader)` keeping the first batch_size rows.

Difference that is output-identical on the rows every caller keeps ([:batch_size],
reference trainers/node_level_graphsage_trainers.py:40, models/graphsage.py:63): when the
edge_index comes from UniformLoader (it carries the hop structure) and
`prune_to_seeds` is True, layer i is evaluated only on the rows that can influence the seed
rows — hop_nodes[L-1-i] targets — and forward returns [batch_size, C] instead of [N_b, C].
"""
import torch

from .. import functional as SF
from .. import settings
from ..layers import SAGE


def set_batch_size_attr(batch):
    try:
        getattr(batch, 'batch_size')
    except AttributeError:
        batch.batch_size = batch.num_graphs
    return batch


class GraphSAGE(torch.nn.Module):
    def __init__(self, in_channels: int, hidden_channels: int, aggregator: str = 'mean',
                 num_layers: int = 2, out_channels=None):
        super().__init__()
        self.aggregator = aggregator
        self.name = 'GraphSAGE-' + aggregator.upper()
        self.prune_to_seeds = True
        if out_channels is None:
            out_channels = hidden_channels
        self.layers = torch.nn.ModuleList(
            [SAGE(in_channels, hidden_channels, aggregator)] +
            [SAGE(hidden_channels, hidden_channels, aggregator) for _ in range(num_layers - 2)] +
            [SAGE(hidden_channels, out_channels, aggregator)]
        )

    def _targets(self, plan, i):
        """Rows of layer i's output that can reach the seed rows through the later layers."""
        L = len(self.layers)
        h = min(L - 1 - i, len(plan.hop_nodes) - 1)
        return plan.hop_nodes[h]

    def forward(self, x, edge_index):
        plan = getattr(edge_index, '_sage_plan', None)
        prunable = (self.prune_to_seeds and plan is not None and plan.hop_nodes is not None
                    and plan.num_src == x.size(0))
        if not prunable:
            for i, layer in enumerate(self.layers):
                x = layer(x, edge_index)
                if i < len(self.layers) - 1:
                    # out of place: the normalise Function saved its output for backward (the
                    # reference's relu_() at models/graphsage.py:48 has the same values)
                    x = x.relu()
            return x
        L = len(self.layers)
        for i, layer in enumerate(self.layers):
            t = self._targets(plan, i)
            n_src = plan.hop_nodes[min(L - i, len(plan.hop_nodes) - 1)]
            x = layer.apply_plan(x, x, plan, t, relu=(i < L - 1), num_src_rows=min(n_src, x.size(0)))
        return x

    def forward_batch(self, batch):
        """Same result as forward(batch.x, batch.edge_index)[:batch.batch_size] without
        materialising batch.x: layer 0 gathers from the global feature table through the
        batch's node vector."""
        plan = batch.edge_index._sage_plan
        if plan.table is None:
            return self.forward(batch.x, batch.edge_index)[:batch.batch_size]
        L = len(self.layers)
        x = plan.table
        node_map = plan.node
        for i, layer in enumerate(self.layers):
            t = self._targets(plan, i)
            n_src = plan.hop_nodes[min(L - i, len(plan.hop_nodes) - 1)]
            x = layer.apply_plan(x, x, plan, t, relu=(i < L - 1), node_map=node_map,
                                 num_src_rows=n_src)
            node_map = None
        return x

    @torch.no_grad()
    def inference(self, subgraph_loader):
        self.eval()
        device = settings.DEVICE
        x_all = []
        for batch in subgraph_loader:
            set_batch_size_attr(batch)
            x = batch.x.to(device)
            edge_index = batch.edge_index.to(device)
            if not hasattr(edge_index, '_sage_plan') and hasattr(batch.edge_index, '_sage_plan'):
                edge_index._sage_plan = batch.edge_index._sage_plan
            x = self.forward(x, edge_index)
            x_all.append(x[:batch.batch_size])
        return torch.cat(x_all, dim=0).to(device)
