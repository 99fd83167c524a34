"""Sanitizer job for the host C++ of the product (the reference's host side is Rust and gets its memory
safety from the compiler; the C++ restatement gets it from this job).  llm_amd/csrc/ggml_core.cpp (the ggml C API: arenas,
tensor builders, views, graph build / plan, quantizers) and llm_amd/csrc/host/llm_host.cpp (the InferenceSession / models/llama
mirror, the GGML / GGMF / GGJT reader, snapshots, the greedy sampler) are compiled with g++ -fsanitize=address,undefined and
linked against tests/sanitize/stub_backend.cpp (a host stand-in for the device backend: nothing is computed), then
tests/sanitize/driver.cpp walks the reference's call sequences — llm::load, two sessions on two threads over one model,
feed_prompt in n_batch chunks, infer_next_token, rewind, K/V out and in, snapshot / from_snapshot, top-k — the ggml wrapper's
context / scratch / view / graph calls, the five quantizers, and feeds the container reader truncated and corrupted copies
of a model file.  Any ASan / UBSan / LeakSanitizer report fails the test.  Runs on the CPU (no GPU, no HIP)."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("g++") is None, reason="g++ not installed")
def test_host_cpp_under_asan_and_ubsan(tmp_path):
    from llm_amd import ggml, synth
    exe = tmp_path / "sanitize_driver"
    srcs = ["llm_amd/csrc/ggml_core.cpp", "llm_amd/csrc/host/llm_host.cpp", "tests/sanitize/stub_backend.cpp",
            "tests/sanitize/driver.cpp"]
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
           "-fno-omit-frame-pointer", "-Iinclude", "-Illm_amd/csrc", "-pthread"] + srcs + ["-o", str(exe)]
    r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    # a two-layer LLaMA (Q4_0; n_ff a multiple of 64: the loader's dims[0] % 64 rule for Q4_0 / Q4_1) in the GGJT v3 container
    hp0 = dict(n_vocab=256, n_embd=128, n_head=4, n_head_kv=4, n_layer=2, n_rot=32, n_ff=384, n_mult=32)
    hp, w = synth.make_llama(hp0, ggml.TYPE_Q4_0)
    path = tmp_path / "tiny.ggjt"
    synth.write_ggjt(str(path), hp, w)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0:halt_on_error=1", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    r = subprocess.run([str(exe), str(path)], capture_output=True, text=True, timeout=600, env=env)
    tail = (r.stdout + r.stderr)[-4000:]
    if r.returncode != 0 and "LeakSanitizer has encountered a fatal error" in tail:  # ptrace-restricted sandbox: leaks unchecked
        env["ASAN_OPTIONS"] = "detect_leaks=0:halt_on_error=1"
        r = subprocess.run([str(exe), str(path)], capture_output=True, text=True, timeout=600, env=env)
        tail = (r.stdout + r.stderr)[-4000:]
    assert r.returncode == 0, tail
    assert "sanitize driver OK" in r.stdout
    assert "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr and "LeakSanitizer" not in r.stderr, tail


@pytest.mark.skipif(shutil.which("g++") is None, reason="g++ not installed")
def test_host_cpp_under_tsan(tmp_path):
    """The same driver under ThreadSanitizer: its two-sessions-on-two-threads walk (the reference's "several sessions for one
    model on several threads", crates/llm-base/src/inference_session.rs:43-48) must not race in the host C++ — the shared model,
    the process-wide host timers, the ggml arenas of the two sessions."""
    from llm_amd import ggml, synth
    exe = tmp_path / "tsan_driver"
    srcs = ["llm_amd/csrc/ggml_core.cpp", "llm_amd/csrc/host/llm_host.cpp", "tests/sanitize/stub_backend.cpp",
            "tests/sanitize/driver.cpp"]
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-fsanitize=thread", "-fno-omit-frame-pointer", "-Iinclude", "-Illm_amd/csrc",
           "-pthread"] + srcs + ["-o", str(exe)]
    r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    hp0 = dict(n_vocab=256, n_embd=128, n_head=4, n_head_kv=4, n_layer=2, n_rot=32, n_ff=384, n_mult=32)
    hp, w = synth.make_llama(hp0, ggml.TYPE_Q4_0)
    path = tmp_path / "tiny.ggjt"
    synth.write_ggjt(str(path), hp, w)
    env = dict(os.environ, TSAN_OPTIONS="halt_on_error=0")
    r = subprocess.run([str(exe), str(path)], capture_output=True, text=True, timeout=600, env=env)
    tail = (r.stdout + r.stderr)[-4000:]
    if "FATAL: ThreadSanitizer" in tail:  # a sandbox whose memory layout the runtime does not support
        pytest.skip("ThreadSanitizer cannot run here: " + tail.strip().splitlines()[-1])
    assert "WARNING: ThreadSanitizer" not in r.stderr, tail
    assert r.returncode == 0 and "sanitize driver OK" in r.stdout, tail
