"""llm_amd — MI355X-native drop-in for the quantized-matmul hot path of rustformers/llm.

Layout (only what the path needs):
  csrc/            HIP kernels for gfx950 + the C ABI (libggml_hip.so) + the C++ host mirror of the
                   reference's crates/ggml wrapper, llm-base InferenceSession and models/llama
  ggml.py          ctypes twin of the reference's `ggml-sys` bindings over that ABI
  llama.py         ctypes handle on the host mirror (Model / InferenceSession call sequence)
  synth.py         synthetic GGML-format weights (no real checkpoints are obtainable offline)
  gpt2.py          GPT-2 (BASELINE configs[0]) graph builder over the ctypes binding
  pipeline.py      layer split over several GPUs: stages, schedule, the residual hop through ggml_hip_comm_* (RCCL)
"""
__all__ = ["ggml", "llama", "synth"]
