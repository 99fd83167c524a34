"""`python -m llm_amd.pipeline --selftest`: the first-contact check of the multi-GPU path (communicator inside the library,
one ring hop with a checked payload) as a one-rank run on the test box's GPU — RCCL accepts a send to oneself inside a group,
so the same code that runs under torchrun on 8 GPUs is exercised end to end here."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_pipeline_selftest_single_rank():
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29541",
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "llm_amd.pipeline", "--selftest"], cwd=ROOT, env=env, capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-2000:])
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["selftest"] == "passed"
    assert d["ranks"][0]["comm_ranks_seen_by_rccl"] == 1 and d["ranks"][0]["payload_intact"] is True
