"""Person-sharded rendering (BASELINE.json configs[3]) with two ranks; the ranks share the box's GPU(s) and
talk over gloo, which exercises the same code path as RCCL except for the all_to_all call itself."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu


def test_person_sharded_render_matches_single_process():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(29600 + os.getpid() % 300), os.path.join(root, "tests", "dist_person_sharded.py")]
    r = subprocess.run(cmd, cwd=root, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    print(r.stdout[-3000:])
    assert r.returncode == 0
