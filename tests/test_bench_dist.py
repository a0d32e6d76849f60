"""bench.py's multi-rank branches on the one GPU of the test box.

`--force-dist` makes a one-rank job open the RCCL process group and go through
every barrier and all-reduce a multi-rank job goes through (the shared-table
mode adds the job's one data exchange, the sum of the ranks' byte counts), under
the launcher the driver uses for N > 1.  The first multi-GPU run is then not the
first execution of those lines.
"""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra, port):
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
           "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "bench.py"), "--gpus", "1", "--force-dist", "--total-mib", "64",
           "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-pipelined", "--no-side"] + extra
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=240, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    return json.loads(lines[0])


@pytest.mark.parametrize("extra", [["--shared-table"], []], ids=["shared_table", "per_frame_codes"])
def test_bench_runs_its_multi_rank_branches_on_one_gpu(extra):
    line = _run(extra, 29650 + (os.getpid() % 200) + (7 if extra else 0))
    assert line["forced_dist"] is True and line["n_gpus"] == 1 and line["steps"] == 2
    assert line["value"] > 0 and line["ms_per_step"] > 0
    assert line["config"]["shared_table"] is bool(extra)
    assert line["roofline"]["frac"] > 0
    assert line["kernels"]["decode"]["ms"] > 0
