/* llm_host.h — C entry points of the host-side mirror (llm_host.cpp) of the reference's
 * crates/llm-base InferenceSession + crates/models/llama, so that tests and bench.py (Python, ctypes)
 * can drive the SAME call sequence a Rust user of rustformers/llm drives:
 *   llm::load → Model::start_session → InferenceSession::{feed_prompt, infer_next_token} → Model::evaluate.
 * Not part of the ggml drop-in ABI (include/ggml_hip.h); exported from the same shared object. */
#ifndef LLM_HOST_H
#define LLM_HOST_H
#include <stddef.h>
#include <stdint.h>

#include "ggml_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* crates/models/llama/src/lib.rs:399-416 Hyperparameters (+ n_ff, which the reference derives from
 * the w1 tensor shape in the file) */
typedef struct {
    int32_t n_vocab, n_embd, n_mult, n_head, n_head_kv, n_layer, n_rot, file_type;
} llm_llama_hparams;

/* crates/llm-base/src/model/mod.rs:196-229 ModelParameters (the fields that touch this path) */
typedef struct {
    int32_t context_size;   /* default 2048 */
    int32_t use_gpu;        /* must be 1: this library has no CPU compute path */
    int32_t gpu_layers;     /* -1 = all */
    int32_t has_rope_overrides;
    float rope_frequency_scale;
    int32_t rope_frequency_base;
    /* layer-split extension: this process owns layers [layer_begin, layer_end) */
    int32_t layer_begin, layer_end; /* 0,-1 = all */
    /* ModelParameters::n_gqa (crates/llm-base/src/model/mod.rs:213-214): grouped-query factor, 0 = None.  As in the reference
     * (crates/models/llama/src/lib.rs:106-117) it is honoured for models of 80 layers and more only ("temporary fix for 70B"):
     * n_head_kv = n_head / n_gqa; the container carries no n_head_kv. */
    int32_t n_gqa;
} llm_model_params;

/* crates/llm-base/src/inference_session.rs:799-841 InferenceSessionConfig */
typedef struct {
    int32_t memory_k_type; /* ggml_type: F16 (default) or F32 */
    int32_t memory_v_type;
    int32_t n_batch;   /* default 8 */
    int32_t n_threads; /* default 8 (ignored by the device executor) */
} llm_session_config;

/* one tensor handed to the loader: the TensorLoader::load(name) contract of
 * crates/llm-base/src/loader.rs:651-678 — name, type, dims ([in, out]) and a host pointer to the GGML
 * bytes (an mmap'd file region or a caller-owned buffer that outlives the model). */
typedef struct {
    const char *name;
    int32_t type; /* ggml_type */
    int32_t n_dims;
    int64_t ne[2];
    void *data;
} llm_tensor_desc;

typedef struct llm_model llm_model;
typedef struct llm_session llm_session;

GGML_API llm_model *llm_llama_new(const llm_llama_hparams *hp, const llm_model_params *params,
                                  const llm_tensor_desc *tensors, int n_tensors);
GGML_API void llm_model_free(llm_model *m);
/* Layer split of ONE model over several device slots of this process: llm_llama_new / llm_llama_load
 * split the layers into contiguous ranges, one per slot, when ggml_hip_set_tensor_split gave more than one slot a
 * positive share (the reference's hook, crates/ggml/src/accelerator/mod.rs:68-77) or GGML_HIP_LAYER_SPLIT=G asks for
 * equal shares; sessions of such a model walk the stages, the residual crosses with ggml_hip_copy_between_devices.
 * Returns the number of stages and, for the first `cap`, their layer ranges [begin, end) and device slots. */
GGML_API int llm_model_stages(const llm_model *m, int *layer_begin, int *layer_end, int *device, int cap);

/* GGML / GGMF v1 / GGJT v1-3 container reader with an mmap'd tensor section:
 * crates/ggml/src/format/loader.rs:160-281 (container walk, 32-byte alignment of GGJT tensor data, the
 * dims[0] % 64 rule for Q4_0/Q4_1) + crates/llm-base/src/loader.rs:419-567, 702-755 (mmap loader) +
 * crates/models/llama/src/lib.rs:425-447 (LLaMA hyperparameters) + loader.rs:32-50 (ftype = format +
 * 1000 * quantization_version).  Returns NULL after printing the LoadError-style reason to stderr. */
typedef struct llm_ggml_file llm_ggml_file;
GGML_API llm_ggml_file *llm_ggml_file_open(const char *path);
GGML_API void llm_ggml_file_close(llm_ggml_file *f);
/* container: 0 ggml, 1 ggmf, 2 ggjt, 3 ggla; hp may be NULL */
GGML_API void llm_ggml_file_info(const llm_ggml_file *f, int *container, int *version, llm_llama_hparams *hp,
                                 int *n_tensors, int *n_vocab_entries);
/* i-th tensor in file order; desc->name / desc->data point into the mapping (valid until close) */
GGML_API int llm_ggml_file_tensor(const llm_ggml_file *f, int i, llm_tensor_desc *desc);
/* i-th vocabulary entry: returns the token's byte length, copies at most cap bytes, *score nullable */
GGML_API int llm_ggml_file_vocab(const llm_ggml_file *f, int i, char *buf, int cap, float *score);
/* llm::load for LLaMA: open + Llama::new over the mapping (kept alive by the model); hyperparameters from the file */
GGML_API llm_model *llm_llama_load(const char *path, const llm_model_params *params);
GGML_API llm_session *llm_start_session(llm_model *m, const llm_session_config *cfg);
/* a session of an unsplit model on another device slot of the model's GPU: own stream / shadows / K/V / plans, shared weights */
GGML_API llm_session *llm_start_session_on(llm_model *m, const llm_session_config *cfg, int slot);
GGML_API void llm_session_seek(llm_session *s, int n_past);
GGML_API void llm_session_set_speculate(llm_session *s, int on);
GGML_API void llm_session_free(llm_session *s);
/* Model::evaluate (models/llama/src/lib.rs:144-368): feeds n tokens at the session's n_past.
 * all_logits (nullable): n*n_vocab floats (OutputRequest.all_logits); embeddings (nullable): n_embd floats. */
GGML_API void llm_evaluate(llm_model *m, llm_session *s, const int32_t *tokens, int n, float *all_logits,
                           float *embeddings);
/* InferenceSession::feed_prompt with Prompt::Tokens (inference_session.rs:299-349): chunks by n_batch */
GGML_API void llm_feed_prompt(llm_model *m, llm_session *s, const int32_t *tokens, int n);
/* InferenceSession::infer_next_token with a greedy (argmax) sampler; returns the sampled token id */
GGML_API int32_t llm_infer_next_token_greedy(llm_model *m, llm_session *s);
/* The same step with the shape of the reference's DEFAULT sampler (samplers.rs:97-188: repetition penalty over the last 64 tokens,
 * top-k, temperature) instead of argmax; device_topk = 1 leaves the logits in HBM and reads k + 64 pairs (llm_session_topk),
 * 0 reads all n_vocab logits back as the reference does.  Both draw the same tokens from the same xorshift64* state. */
GGML_API int32_t llm_infer_next_token_topk(llm_model *m, llm_session *s, int k, float temperature, uint64_t *rng, int device_topk);
/* InferenceSession::rewind (inference_session.rs:352-378) */
/* n greedy tokens with the argmax on the device (ggml_hip_decode_greedy_chain): same ids and final logits as n calls
 * of llm_infer_next_token_greedy, no per-token logits read-back; falls back to that loop when chaining is impossible */
GGML_API int llm_infer_tokens_greedy_device(llm_model *m, llm_session *s, int n, int32_t *out);
GGML_API size_t llm_session_read_node_host(const llm_session *s, int from_end, void *dst, size_t max_bytes);
GGML_API int llm_session_rewind(llm_session *s, int num);
/* host nanoseconds per phase of the decode loop, accumulated: [0] adopt/build graph, [1] token write + plan,
 * [2] compute begin (match + enqueue), [3] speculative build of the next graph, [4] compute end (wait + copy),
 * [5] greedy argmax, [6] evaluate as a whole. */
GGML_API void llm_host_timing(double *out8, int reset);
GGML_API const float *llm_session_last_logits(const llm_session *s);
GGML_API int llm_session_n_past(const llm_session *s);
GGML_API int llm_model_n_vocab(const llm_model *m);
/* graph statistics of the last evaluate (for tests): nodes, leafs */
GGML_API void llm_session_last_graph_stats(const llm_session *s, int *n_nodes, int *n_leafs);

/* layer split: device addresses of the residual hand-off buffers of a stage session (NULL if absent) */
GGML_API void llm_session_stage_buffers(llm_session *s, void **in_dev, void **out_dev, size_t *nbytes);
/* raw K/V memory of a session (which: 0 = memory_k, 1 = memory_v; set: 0 = read into buf, 1 = write from buf);
 * buf == NULL returns the size.  The InferenceSnapshot payload (inference_session.rs:599-646). */
/* test hook: the greedy sampler's index (first maximum under `>`, NaNs never win); which = 1: the scalar loop */
GGML_API int llm_argmax_first(const float *logits, int n, int which);
/* layer ranges of an in-process split over G device slots for the fractions `split` (NULL = equal): bounds_out[0..G] */
GGML_API void llm_split_layers(int n_layer, int G, const float *split, int *bounds_out);
GGML_API size_t llm_session_kv(llm_session *s, int which, int set, void *buf, size_t nbytes);
/* InferenceSession::get_snapshot / from_snapshot (inference_session.rs:590-646): npast, config, tokens, last_logits and
 * the K/V memory (read from / written to the device).  snapshot: buf == NULL or cap too small returns the size needed.
 * from_snapshot: NULL on a malformed buffer or SnapshotError::MemorySizeMismatch. */
GGML_API size_t llm_session_snapshot(llm_session *s, void *buf, size_t cap);
GGML_API llm_session *llm_session_from_snapshot(llm_model *m, const void *buf, size_t n);
/* test hook: reads the device contents of a node of the last evaluated graph (by index, or k-th node named `name`) */
/* top-k prefilter of the last token's logits on the device (ggml_hip_topk): k (value, id) pairs, best first, then the
 * raw logits of extra_ids; 0 on success, -1 if there is no evaluated graph or the arguments are out of range */
GGML_API int llm_session_topk(const llm_session *s, int k, const int32_t *extra_ids, int n_extra, float *out_vals,
                              int32_t *out_ids);
GGML_API size_t llm_session_read_node(const llm_session *s, int index, const char *name, int occurrence, void *dst,
                                      size_t max_bytes);
/* synthetic GGML blocks for full-size benchmarks (deterministic in seed and block index) */
GGML_API void llm_synth_blocks(int type, void *dst, int64_t nblocks, uint64_t seed, float d_scale);
/* BASELINE.md section 4 weights at full size: ne1 rows of ne0 gaussians N(0, std^2) (counter-based generator, deterministic
 * in (seed, row)), each row quantized by ggml_quantize_q* into raw GGML blocks of `type` at dst; multi-threaded. */
GGML_API void llm_synth_gaussian(int type, void *dst, int64_t ne0, int64_t ne1, uint64_t seed, float std);

#ifdef __cplusplus
}
#endif
#endif
