"""bench.py's multi-GPU record must be self-proving: N ranks on fewer than N devices are
refused unless --rehearsal is given, and the identities that go into the line count distinct devices by UUID."""
import argparse
import importlib.util
import os

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_ranks_sharing_a_gpu_are_refused_without_rehearsal(monkeypatch):
    bench = _bench()
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 1)
    with pytest.raises(SystemExit) as e:
        bench.pick_device(argparse.Namespace(rehearsal=False), 1, 2)
    assert "rehearsal" in str(e.value)
    assert bench.pick_device(argparse.Namespace(rehearsal=True), 1, 2) == 0       # wraps, labelled by the caller
    assert bench.pick_device(argparse.Namespace(rehearsal=False), 0, 1) == 0
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 8)
    assert [bench.pick_device(argparse.Namespace(rehearsal=False), r, 8) for r in range(8)] == list(range(8))


def test_devices_seen_counts_distinct_uuids():
    bench = _bench()
    a = {"rank": 0, "host": "h", "uuid": "GPU-aa", "device": 0, "name": "x"}
    b = dict(a, rank=1)                      # a second rank on the SAME device
    c = dict(a, rank=1, uuid="GPU-bb", device=1)
    assert bench.devices_seen([a, b]) == 1
    assert bench.devices_seen([a, c]) == 2
    assert bench.devices_seen([a, dict(c, host="other", uuid="GPU-aa")]) == 2   # same UUID text, another host


def test_self_launch_refuses_before_spawning(monkeypatch):
    bench = _bench()
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 1)
    monkeypatch.delenv("RANK", raising=False)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    called = []
    monkeypatch.setattr(bench.subprocess, "call", lambda cmd: called.append(cmd) or 0)
    with pytest.raises(SystemExit) as e:
        bench.self_launch_if_needed(argparse.Namespace(gpus=2, rehearsal=False))
    assert "rehearsal" in str(e.value) and not called
