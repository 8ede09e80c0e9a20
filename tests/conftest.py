import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


REHEARSAL_LOG = os.path.join(ROOT, "gpurun_out", "ddp_rehearsal_pytest.log")


def pytest_sessionstart(session):
    """`-m gpu` sessions: run the two-rank rehearsal of the data-parallel train step (tools/ddp_rehearsal.py) as CHILD
    processes NOW, before this process has touched the GPU (a process that has initialised HIP must not start others on
    this pool).  tests/test_ddp_rehearsal_gpu.py reads the outcome."""
    expr = session.config.getoption("markexpr", "") or ""
    if "gpu" not in expr or "not gpu" in expr:
        return
    import socket
    import subprocess
    import torch
    if torch.cuda.device_count() < 1:                     # counting devices does not initialise the GPU
        return
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    os.makedirs(os.path.dirname(REHEARSAL_LOG), exist_ok=True)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "tools", "ddp_rehearsal.py"), "--size", "64",
           "--steps", "3"]
    try:
        res = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=900)
        text = f"rc={res.returncode}\n{res.stdout}\n{res.stderr[-4000:]}"
    except subprocess.TimeoutExpired as e:
        text = f"rc=timeout\n{e.stdout}\n{e.stderr}"
    with open(REHEARSAL_LOG, "w") as fh:
        fh.write(text)
