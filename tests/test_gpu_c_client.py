"""The C ABI from plain C: examples/c_abi_client (C99, gcc, libfpn_hip.so + the HIP runtime,
no Python and no torch in its process) samples and queries one synthetic frustum and checks
the results against the operators' definitions written as loops."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _binary():
    path = os.path.join(ROOT, "examples", "c_abi_client")
    if not os.path.exists(path):
        import sys
        sys.path.insert(0, ROOT)
        import __graft_entry__ as entry
        entry.build_c_client()
    return path


def test_c_client_builds_against_the_header():
    """include/fpn.h is a C header and the library links into a C program."""
    if os.path.exists(os.path.join(ROOT, "examples", "c_abi_client")):
        os.remove(os.path.join(ROOT, "examples", "c_abi_client"))
    assert os.access(_binary(), os.X_OK)


@pytest.mark.gpu
def test_c_client_runs(device):
    r = subprocess.run([_binary()], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "agree with the definitions" in r.stdout
