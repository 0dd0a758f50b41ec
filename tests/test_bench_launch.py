"""bench.py's launch paths for N > 1 (BASELINE.json north_star: seed nodes shard over the GPUs of
one node; the reference itself has a single DEVICE, graphsage/settings.py:32).

* `python bench.py --gpus N` with no WORLD_SIZE in the environment starts its N ranks itself,
  as children of a process that has not touched the GPU;
* launched by torch.distributed.run (the driver's form) it runs as one rank.
The GPU test rehearses the first form with two ranks on the one GPU of the test box, gloo
carrying the all-reduce (SAGE_BENCH_BACKEND=gloo: RCCL refuses two ranks per device)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_self_launch_command(monkeypatch):
    sys.path.insert(0, ROOT)
    import bench
    seen = {}

    class Done:
        returncode = 7

    def fake_run(cmd, env=None, **kw):
        seen["cmd"], seen["env"] = cmd, env
        return Done()
    monkeypatch.setattr(subprocess, "run", fake_run)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "3"])
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    assert bench.self_launch(4) == 7                       # the children's exit code is relayed
    cmd = seen["cmd"]
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert cmd[cmd.index("--nproc-per-node") + 1] == "4"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert 1024 < int(cmd[cmd.index("--master-port") + 1]) < 65536
    k = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[k + 1:] == ["--gpus", "4", "--steps", "3"]  # the ranks see the same flags
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_schedule_rule_of_the_engine_and_the_bench():
    """overlap_gather='auto' (engine, trainers, fig2a, bench.py): the side-stream gather for the
    supervised step on rows wide enough to pay for the cross-stream hand-shake, the serial schedule
    for narrow rows and for the three-forward steps (profiles/r3_schedule_ab.txt)."""
    from graphsage_amd.engine import SageTrainEngine as E
    assert E.gather_pays_overlap(602, 512, [25, 10]) is True          # Reddit
    assert E.gather_pays_overlap(100, 512, [25, 10]) is True          # products
    assert E.gather_pays_overlap(50, 512, [25, 10]) is False          # PPI
    assert E.gather_pays_overlap(602, 512, [25, 10], unsup=True) is False
    assert E.gather_pays_overlap(602, 512, [25, 10], unsup='only') is False


def test_main_self_launches_only_without_world_size(monkeypatch):
    sys.path.insert(0, ROOT)
    import bench
    calls = []
    monkeypatch.setattr(bench, "self_launch", lambda n: calls.append(n) or 0)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "2"])
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 0 and calls == [2]


@pytest.mark.gpu
def test_bench_gpus2_self_launch_gloo_rehearsal(dev):
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    env["SAGE_BENCH_BACKEND"] = "gloo"
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--workload", "reddit-small",
           "--steps", "20", "--warmup", "5", "--settle-steps", "20", "--cpu-steps", "0", "--no-variant"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=420)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]               # rank 0 alone prints
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["config"]["parallelism"] == "dp2" and out["config"]["global_batch"] == 1024
    assert out["schedule"].startswith("data parallel, serial") and out["overlap_gather"] is False
    assert out["value"] > 0 and out["steps"] == 20
    assert out["roofline"]["bound"] == "hbm" and 0 < out["roofline"]["frac"] < 1.2
