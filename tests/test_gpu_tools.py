"""The re-hosted reference entry points (tools/det/*.py, flag names from
/root/reference/README.md:54-75) run end to end on the MI355X path with synthetic data."""
import os
import subprocess
import sys

import pytest
import torch

from tests.conftest import ROOT

pytestmark = pytest.mark.gpu


def _run(args):
    r = subprocess.run([sys.executable] + args, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    return r.stdout


def test_train_codet_trains_saves_and_resumes(tmp_path):
    tool = os.path.join(ROOT, "tools", "det", "train_codet.py")
    logs = str(tmp_path / "logs")
    out = _run([tool, "--com", "disco", "--batch", "1", "--nepoch", "2", "--steps_per_epoch", "2",
                "--num_agent", "3", "--logpath", logs])
    assert "epoch 1:" in out and "epoch 2:" in out
    ck = torch.load(os.path.join(logs, "epoch_2.pth"), map_location="cpu", weights_only=False)
    assert {"epoch", "model_state_dict", "optimizer_state_dict", "scheduler_state_dict", "loss"} <= set(ck)
    assert ck["optimizer_state_dict"]["step"] == 4
    # resume + knowledge distillation from a (random) teacher
    out = _run([tool, "--com", "disco", "--batch", "1", "--nepoch", "1", "--steps_per_epoch", "2",
                "--num_agent", "3", "--kd_flag", "1", "--resume", os.path.join(logs, "epoch_2.pth")])
    assert "resumed" in out and "epoch 3:" in out and "kd_loss" in out


def test_test_codet_runs_a_trained_checkpoint(tmp_path):
    logs = str(tmp_path / "logs")
    _run([os.path.join(ROOT, "tools", "det", "train_codet.py"), "--com", "disco", "--batch", "1", "--nepoch", "1",
          "--steps_per_epoch", "1", "--num_agent", "2", "--logpath", logs])
    out = _run([os.path.join(ROOT, "tools", "det", "test_codet.py"), "--com", "disco", "--num_agent", "2",
                "--frames", "2", "--resume", os.path.join(logs, "epoch_1.pth")])
    assert "loaded" in out and "frame 1:" in out
