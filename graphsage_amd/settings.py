"""Run-time settings; same names as the reference's graphsage/settings.py:29-62
(DEVICE, BATCH_SIZE, NUM_WORKERS, NUM_EPOCHS, ...), but parsed with parse_known_args so
importing the package under pytest / torchrun does not choke on foreign argv
(the reference calls parse_args() at import, settings.py:27)."""
import argparse
import os

import torch

parser = argparse.ArgumentParser(add_help=False)
parser.add_argument("--device", type=str, default=None)
parser.add_argument("--batch_size", type=int, default=512)
parser.add_argument("--num_workers", type=int, default=4)
parser.add_argument("--num_epochs", type=int, default=10)
parser.add_argument("--num_runs", type=int, default=1)
parser.add_argument("--persistent_workers", action="store_true")
parser.add_argument("--no_eval_train", action="store_true")
parser.add_argument("--lstm_num_inputs", type=int, default=None)
parser.add_argument("--use_triple_loss", action="store_true")
parser.add_argument("--seed", type=int, default=0)
# flags of the reference's settings.py:6-26 that its example / experiment scripts read from `args`
parser.add_argument("--aggregator", type=str, default="mean")
parser.add_argument("--dataset", type=str, default=None)
parser.add_argument("--ignore_datasets", nargs="*", type=str, default=None)
parser.add_argument("--ignore_aggregators", nargs="*", type=str, default=None)
parser.add_argument("--results_dir", type=str, default=None)
parser.add_argument("--no_show", action="store_true")
parser.add_argument("--no_extensions", action="store_true")
parser.add_argument("--std", action="store_true")
parser.add_argument("--colab", action="store_true")
args, _unknown = parser.parse_known_args()

DEVICE = torch.device(args.device if args.device else ("cuda" if torch.cuda.is_available() else "cpu"))
BATCH_SIZE = args.batch_size
NUM_WORKERS = args.num_workers          # accepted for API parity; sampling runs on the device
NUM_EPOCHS = args.num_epochs
NUM_RUNS = args.num_runs
PERSISTENT_WORKERS = args.persistent_workers
NO_EVAL_TRAIN = args.no_eval_train
SEED = args.seed
# The reference defines LSTM_NUM_INPUTS only when the flag is given (settings.py:55-56) and
# then crashes in SAGE.__init__; here it defaults to the first fan-out (25).
LSTM_NUM_INPUTS = args.lstm_num_inputs if args.lstm_num_inputs is not None else 25

if args.ignore_aggregators is None:
    args.ignore_aggregators = ["bilstm"]     # the reference also ignores lstm (settings.py:48-50: "when those are fixed")

ROOT_DIR = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# scripts join these with a dataset name; nothing is created here (the reference's package
# __init__ makes both directories at import) and the synthetic datasets ignore their `root`
DATA_DIR = os.path.join(ROOT_DIR, "data")
RESULTS_DIR = args.results_dir or os.path.join(ROOT_DIR, "results")
