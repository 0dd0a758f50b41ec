"""fig2b-style sweep (reference experiments/fig2b/run.py:23-47 + runners/runners.py:20-58): test
micro-F1 and test-pass runtime of supervised GraphSAGE against the neighbourhood sample size
(k1 = k2 = s for s in SAMPLE_SIZES).

Same names and return keys as the reference: `SampleSizeRunner(s).run()` -> {'test_time' (mean
over epochs and runs of one pass over the test split), 'test_f1' (best-validation epoch, mean over
runs)}, `get(s)`.  The trainer is the reference's own class for this experiment
(SupervisedGraphSageTrainerForNodeLevelTask: supervised + unsupervised term) on the device path;
the engine runs it when `use_engine` is left at its default.  Reference quirk kept: the hidden
width is BATCH_SIZE (runners.py:31 passes hidden_channels=fig2b_settings.BATCH_SIZE).  Results go
to stdout as JSON lines (no PNG: plotting is out of scope).

    python -m graphsage_amd.experiments.fig2b --dataset cora --sample-sizes 5 10 20
"""
import argparse
import json

import torch

from .. import settings
from ..models import GraphSAGE
from ..samplers import UniformLoader
from ..trainers import SupervisedGraphSageTrainerForNodeLevelTask
from .fig2a import load_dataset

NUM_LAYERS = 2
SAMPLE_SIZES = [5, 10, 20, 30, 70]      # fig2b/settings.py:7
BATCH_SIZE = 512                        # fig2b/settings.py:8
LEARNING_RATE = 1e-3                    # fig2b/settings.py:9


class SampleSizeRunner:
    def __init__(self, sample_size, dataset=None, aggregator='mean', num_epochs=None, num_runs=None,
                 hidden_channels=BATCH_SIZE, learning_rate=LEARNING_RATE, use_engine=True, verbose=False):
        self.sample_size = sample_size
        self.dataset = dataset if dataset is not None else load_dataset('cora')
        self.aggregator = aggregator
        self.num_epochs = num_epochs if num_epochs is not None else settings.NUM_EPOCHS
        self.num_runs = num_runs if num_runs is not None else settings.NUM_RUNS
        self.hidden_channels = hidden_channels
        self.learning_rate = learning_rate
        self.use_engine = use_engine
        self.verbose = verbose

    def run(self):
        data = self.dataset[0]
        device = settings.DEVICE
        test_times, test_f1s = [], []
        for _ in range(self.num_runs):
            model = GraphSAGE(in_channels=self.dataset.num_features, hidden_channels=self.hidden_channels,
                              out_channels=self.dataset.num_classes, num_layers=NUM_LAYERS,
                              aggregator=self.aggregator).to(device)
            best_result, all_results = SupervisedGraphSageTrainerForNodeLevelTask(
                data, UniformLoader,
                dataset_name='fig2b',
                model=model,
                k1=self.sample_size,
                k2=self.sample_size,
                num_epochs=self.num_epochs,
                loss_fn=torch.nn.CrossEntropyLoss(),
                optimizer=torch.optim.Adam(model.parameters(), lr=self.learning_rate),
                device=device,
                use_engine='auto' if self.use_engine else False,
            ).run(return_best_epoch_only=False, verbose=self.verbose)
            test_times.append(sum(r['test_time'] for r in all_results) / len(all_results))
            test_f1s.append(best_result['test_f1'])
        return {'test_time': sum(test_times) / len(test_times), 'test_f1': sum(test_f1s) / len(test_f1s)}


def get(sample_size, **kw):
    return SampleSizeRunner(sample_size, **kw)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--dataset', default='cora')
    ap.add_argument('--sample-sizes', type=int, nargs='+', default=SAMPLE_SIZES)
    ap.add_argument('--aggregator', default='mean')
    ap.add_argument('--epochs', type=int, default=None)
    ap.add_argument('--runs', type=int, default=None)
    a, _ = ap.parse_known_args()
    ds = load_dataset(a.dataset)
    for s in a.sample_sizes:
        out = get(s, dataset=ds, aggregator=a.aggregator, num_epochs=a.epochs, num_runs=a.runs).run()
        print(json.dumps({'sample_size': s, **out}), flush=True)


if __name__ == '__main__':
    main()
