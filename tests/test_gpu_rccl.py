"""The collective calls of lvae.distributed through the real RCCL process group (backend "nccl") on the one GPU
of the box: a one-rank job in a fresh process (tests/rccl_single_rank.py).  The multi-rank logic itself is covered
on the CPU by the gloo tests; what this adds is ProcessGroupNCCL's stream and work-handle semantics."""
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_collectives_over_rccl_single_rank():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "rccl_single_rank.py")], env=env,
                       capture_output=True, text=True, timeout=110, cwd=ROOT)
    assert p.returncode == 0, (p.stdout[-1500:], p.stderr[-3000:])
    assert p.stdout.strip().splitlines()[-1] == "OK"
