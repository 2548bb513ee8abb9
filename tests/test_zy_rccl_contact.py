"""The collectives the multi-GPU job makes, on RCCL itself: a one-rank world on backend
"nccl" in a child process (tools/rccl_contact.py; RCCL refuses two ranks on one device, so
one GPU cannot hold more) - init with device_id, barrier, all_gather_into_tensor of result
records and of the statistics table, all_reduce(MAX).  The multi-rank control flow is
covered on gloo (tests/test_dist_gloo.py)."""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_single_rank_rccl_collectives(device):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    r = subprocess.run([sys.executable, "-u", os.path.join(ROOT, "tools", "rccl_contact.py")], env=env,
                       capture_output=True, text=True, timeout=150)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "rccl contact ok" in r.stdout
