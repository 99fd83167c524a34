"""Two REAL stages in two processes (the one-process-per-GPU launcher of bench.py --gpus N): a
6-layer model split 3 + 3, two sequences in flight, prompt chunks of 8 then greedy decode — the tokens must be the ones the
unsplit model produces in one process.  On a 1-GPU box both ranks drive GPU 0 and the residual crosses over gloo host
copies (RCCL refuses two ranks on one device; the worker says which hop ran); on a multi-GPU box the same test runs the
hop through RCCL inside the library.  tests/test_pipeline_cpu.py covers the message protocol with a stub stage,
tests/test_split_gpu.py the in-process split."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("wtype", [2, 8])
def test_two_stage_processes_produce_the_single_process_tokens(G, tmp_path, wtype):
    from llm_amd import llama, synth
    n_prompt, n_decode, world = 19, 6, 2
    out = tmp_path / "tokens.json"
    env = dict(os.environ)
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "tools", "pipeline_worker.py"), str(out), str(wtype),
           str(n_prompt), str(n_decode)]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    got = json.loads(out.read_text())
    print(got["hop"], got["layer_ranges"], "rccl ranks:", got["comm_ranks_seen_by_rccl"])
    assert [tuple(x[:2]) for x in got["layer_ranges"]] == [(0, 3), (3, 6)]
    # the same sequences through the whole model in this process
    hp0 = dict(n_vocab=256, n_embd=128, n_head=4, n_head_kv=4, n_layer=6, n_rot=32, n_ff=352, n_mult=32)
    hp, w = synth.make_llama(hp0, wtype, seed=23)
    model = llama.Llama(hp, w, context_size=128)
    rng = np.random.default_rng(77)
    for s in range(world):
        p = rng.integers(0, hp["n_vocab"], n_prompt).astype(np.int32)
        sess = model.start_session(n_batch=8)
        want = []
        for i in range(0, n_prompt, 8):
            logits = sess.evaluate(p[i:i + 8], want_all_logits=True)
            want.append(int(np.argmax(logits[-1])))
        for _ in range(n_decode):
            logits = sess.evaluate(np.array([want[-1]], np.int32), want_all_logits=True)
            want.append(int(np.argmax(logits[-1])))
        sess.free()
        assert got["tokens"][s] == want, (s, got["tokens"][s], want)
    model.free()
