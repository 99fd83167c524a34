// llm_host.cpp — C++ mirror of the host code on the hot path of rustformers/llm:
//   crates/llm-base/src/inference_session.rs  InferenceSession {new, compute, feed_prompt,
//                                              infer_next_token, rewind}            (:114-424)
//   crates/llm-base/src/model/{mod,common}.rs  ModelParameters, OutputRequest, read_last_token …
//   crates/models/llama/src/lib.rs             Llama {new, start_session, evaluate}  (:43-368)
// The reference is Rust and no Rust toolchain exists in this image; this file keeps its structure,
// names and call order so that the graph handed to ggml_graph_compute is node-for-node the graph the
// Rust code builds (tests/test_graph_shape.py counts the nodes).  All compute goes through the C ABI
// of include/ggml_hip.h.
#if defined(__x86_64__)
#include <immintrin.h>
#endif
#include "llm_host.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <cmath>
#include <cstring>
#include <functional>
#include <string>
#include <thread>
#include <vector>

#include "ggml_wrap.hpp"

using ggml::Backend;
using ggml::Buffer;
using ggml::ComputationGraph;
using ggml::Context;
using ggml::GraphExecutionPlan;
using ggml::Tensor;

namespace llm {

using TokenId = int32_t;

// crates/llm-base/src/model/mod.rs:196-251
struct ModelParameters {
    size_t context_size = 2048;
    bool use_gpu = false;
    long gpu_layers = -1;  // None
    bool has_rope_overrides = false;
    ggml::RoPEOverrides rope_overrides;
    long n_gqa = -1;
    // layer-split extension: this process owns layers [layer_begin, layer_end)
    size_t layer_begin = 0, layer_end = (size_t)-1;
    int main_device = -1;  // >= 0: a stage of an in-process layer split, bound to this device slot

    bool should_offload(size_t layer) const {
        if (!use_gpu) return false;
        return gpu_layers < 0 ? true : (long)layer < gpu_layers;
    }
    Backend backend(size_t layer) const { return should_offload(layer) ? Backend::Gpu : Backend::Cpu; }
};

// crates/llm-base/src/inference_session.rs:799-841
struct InferenceSessionConfig {
    ggml::Type memory_k_type = GGML_TYPE_F16;
    ggml::Type memory_v_type = GGML_TYPE_F16;
    size_t n_batch = 8;
    size_t n_threads = 8;
};

// crates/llm-base/src/model/mod.rs:257-266
struct OutputRequest {
    std::vector<float> *all_logits = nullptr;
    std::vector<float> *embeddings = nullptr;
    // extension (no reference counterpart): the caller samples from a device-side top-k (llm_session_topk) — the logits node stays
    // in HBM like every other node and last_logits is NOT refreshed: 40 pairs cross the bus instead of n_vocab floats
    bool logits_on_device = false;
    // extension: a chunk of feed_prompt that is not its last one — nobody can observe its last-token logits (the next chunk
    // overwrites last_logits before feed_prompt returns), so they are not fetched and the evaluation need not be waited for
    bool intermediate_chunk = false;
};

struct GraphOutputs {  // inference_session.rs:31-37
    Tensor result;
    Tensor embedding_result;
};

constexpr size_t SCRATCH_SIZE = 512ull * 1024 * 1024;  // inference_session.rs:19

struct BuildContext {  // inference_session.rs:96-109
    Context *ctx0;
    const Tensor *embd;
    const Tensor *memory_k;
    const Tensor *memory_v;
    const std::shared_ptr<Buffer> *scratch;
    const Buffer *get_scratch(size_t idx) const { return scratch[idx].get(); }
};

class InferenceSession {
   public:
    // inference_session.rs:114-217
    InferenceSession(const InferenceSessionConfig &config, const ModelParameters &params, size_t n_layer,
                     size_t n_embd, size_t n_vocab)
        : config(config), n_embd_(n_embd) {
        const size_t context_size = params.context_size;
        const size_t context_byte_size = [&] {
            double size = 0;
            size += (double)context_size * (double)n_layer * (double)n_embd * (double)ggml_type_sizef(config.memory_k_type);
            size += (double)context_size * (double)n_layer * (double)n_embd * (double)ggml_type_sizef(config.memory_v_type);
            return (size_t)size + (5 + 10 * n_layer) * 256;  // object overhead
        }();
        if (params.use_gpu) {
            // inference_session.rs: ggml::accelerator::initialize(0).  A stage of an in-process layer split lives on its
            // own device slot, which llm_start_session made current: it must stay current (and the caller's split stay set)
            if (params.main_device < 0) ggml::accelerator::initialize(0);
            ggml::accelerator::set_scratch_size(config.n_batch * 1024 * 1024);
        }
        session_ctx_ = std::make_shared<Context>(Context::new_with_allocate(context_byte_size));
        memory_size_ = context_byte_size;
        // Initialize key + value memory tensors (kv_memory, inference_session.rs:996-1021)
        const size_t n_mem = n_layer * context_size;
        const size_t n_elements = n_embd * n_mem;
        memory_k = session_ctx_->new_tensor_1d(config.memory_k_type, n_elements).set_name("memory_k");
        memory_v = session_ctx_->new_tensor_1d(config.memory_v_type, n_elements).set_name("memory_v");
        if (params.use_gpu) {
            memory_k.offload_no_scratch();
            memory_v.offload_no_scratch();
        }
        scratch_[0] = std::make_shared<Buffer>(SCRATCH_SIZE);
        scratch_[1] = std::make_shared<Buffer>(SCRATCH_SIZE);
        const size_t buf_size_mb = n_layer >= 80 ? 1536 : n_layer >= 60 ? 1280 : 1024;
        const size_t buf_size = buf_size_mb * 1024 * 1024 + ggml_graph_overhead();
        ctx0_[0] = Context::new_with_buffer(std::make_shared<Buffer>(buf_size));
        ctx0_[1] = Context::new_with_buffer(std::make_shared<Buffer>(buf_size));
        if (const char *v = getenv("LLM_HOST_SPECULATE")) speculate = atoi(v) != 0;
        last_logits.assign(n_vocab, 0.0f);
    }
    ~InferenceSession() {
        // inference_session.rs:659-665: free accelerator scratch; K/V are freed by the session ctx destructor
        ggml::accelerator::free_scratch();
    }

    using Builder = std::function<std::pair<ComputationGraph, GraphOutputs>(BuildContext &)>;
    // host time per phase of compute(), ns, accumulated (llm_host_timing): [0] adopt/build, [1] token write + plan,
    // [2] begin (match + enqueue), [3] speculative build of the next graph, [4] end (wait + result copy)
    // (process-wide and updated by every session thread: relaxed atomics — a statistic, but not a data race)
    static inline std::atomic<int64_t> host_ns[8] = {};
    static inline void host_ns_add(int k, double ns) { host_ns[k].fetch_add((int64_t)ns, std::memory_order_relaxed); }
    static inline double now_ns() {
        return (double)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count();
    }

    // inference_session.rs:220-295.  `next_builder` (optional) is a host-side latency optimisation that does not
    // change what is computed: while the device runs a single-token graph, the graph of the NEXT single-token call
    // (same session, n_past + 1) is built into the other of two ctx0 arenas; the following compute() adopts it if
    // the session is where the speculation assumed (otherwise it is discarded and the graph is built as usual).
    // The reference rebuilds the graph inside every evaluate; this hides that cost behind the device.
    GraphOutputs compute(const std::vector<TokenId> &input_tokens, const Builder &builder,
                         const std::function<Builder(size_t /*session_len*/)> &next_builder = nullptr,
                         const void *model_key = nullptr, size_t context_size = 0) {
        const bool single = input_tokens.size() == 1;
        double t0 = now_ns(), t1;
        auto lap = [&](int k) { t1 = now_ns(); host_ns_add(k, t1 - t0); t0 = t1; };
        Built built;
        if (single && pre_.valid && pre_.n_past == n_past && pre_.model_key == model_key) {
            cur_ = pre_.slot;  // adopt the speculatively built graph
            built = std::move(pre_.b);
        } else {
            cur_ ^= 1;
            built = build_into(cur_, input_tokens.size(), builder);
        }
        pre_.valid = false;
        lap(0);
        Context &ctx0 = ctx0_[cur_];
        built.embd.write_data(input_tokens.data(), input_tokens.size() * sizeof(TokenId));  // Write input tokens
        {
            GraphExecutionPlan plan(built.gf, config.n_threads);
            lap(1);
            // (pipeline_chunk: a chunk of feed_prompt behind which another one follows at once — begin without end: the next
            //  evaluation's begin, or any other entry point of the backend, completes it; the stream keeps the order)
            const bool pipe = !single && pipeline_chunk && speculate && !defer_end;
            const bool running = (single && speculate) || pipe ? plan.execute_begin(ctx0) : (plan.execute(ctx0), false);
            lap(2);
            if (running && next_builder && n_past + 2 <= context_size) {
                pre_.b = build_into(cur_ ^ 1, 1, next_builder(n_past + 1));
                pre_.slot = cur_ ^ 1;
                pre_.n_past = n_past + 1;
                pre_.model_key = model_key;
                pre_.valid = true;
                GraphExecutionPlan::prepare(pre_.b.gf);
            }
            lap(3);
            // a middle stage of an in-process layer split has nothing for the host to read: its wait is left to the slot's next
            // begin (finish_pending), so the host goes straight on to enqueue the next stage while this one runs
            if (single && speculate && !defer_end) GraphExecutionPlan::execute_end();
            lap(4);
        }
        last_n_nodes = built.gf.raw()->n_nodes;
        last_n_leafs = built.gf.raw()->n_leafs;
        last_graph = built.gf.raw();  // lives in its ctx0 arena until that arena is recreated
        if (mem_per_token == 0) mem_per_token = ctx0.used_mem() / n_embd_;
        n_past += input_tokens.size();
        return GraphOutputs{built.out.result.share(), built.out.embedding_result.share()};
    }

    // n tokens were decoded on the device behind this session's back (ggml_hip_decode_greedy_chain): the K/V memory
    // holds them; bring the bookkeeping of inference_session.rs (n_past, tokens) in line.
    void advance_device_chain(const TokenId *toks, size_t n) {
        tokens.insert(tokens.end(), toks, toks + n);
        n_past += n;
        pre_.valid = false;  // the speculatively built graph assumed n_past + 1
        last_graph = nullptr;
    }

    void drop_prebuilt() {  // a speculatively built next-token graph no longer describes the session's next call
        pre_.valid = false;
        last_graph = nullptr;
    }

    // hand-off buffers of a layer-split stage: n_embd * rows floats each, rows = n_batch to begin with.  Model::evaluate may
    // be called with more tokens than n_batch (an unsplit model takes any count): the buffers then grow to that count
    // (ensure_stage_rows, called by the graph builder before it makes its views).
    void make_stage_buffers(size_t n_embd, bool in, bool out, size_t rows = 0) {
        if (rows == 0) rows = config.n_batch;
        stage_in = Tensor();
        stage_out = Tensor();
        stage_ctx_.reset();  // drops the old buffers' device copies (Context drop frees offloaded tensors)
        stage_ctx_ = std::make_shared<Context>(Context::new_with_allocate(2 * (n_embd * rows * 4 + 1024)));
        if (in) {
            stage_in = stage_ctx_->new_tensor_1d(GGML_TYPE_F32, n_embd * rows).set_name("stage_in");
            stage_in.offload_no_scratch();
        }
        if (out) {
            stage_out = stage_ctx_->new_tensor_1d(GGML_TYPE_F32, n_embd * rows).set_name("stage_out");
            stage_out.offload_no_scratch();
        }
        stage_rows_ = rows;
        stage_has_in_ = in;
        stage_has_out_ = out;
    }
    void ensure_stage_rows(size_t n_embd, size_t rows) {
        if (rows <= stage_rows_) return;
        pre_.valid = false;  // a speculatively built graph holds views of the old buffers
        make_stage_buffers(n_embd, stage_has_in_, stage_has_out_, rows);
    }
    size_t stage_rows() const { return stage_rows_; }

    InferenceSessionConfig config;
    Tensor memory_k, memory_v;
    Tensor stage_in, stage_out;  // layer-split hand-off buffers (null for a whole-model session)
    size_t n_past = 0;
    size_t mem_per_token = 0;
    std::vector<TokenId> tokens;
    std::vector<float> last_logits;
    int last_n_nodes = 0, last_n_leafs = 0;
    ggml_cgraph *last_graph = nullptr;

   private:
    std::shared_ptr<Context> session_ctx_;
    std::shared_ptr<Context> stage_ctx_;
    size_t stage_rows_ = 0;
    bool stage_has_in_ = false, stage_has_out_ = false;
    size_t memory_size_ = 0;
    struct Built {
        ComputationGraph gf{nullptr};
        GraphOutputs out;
        Tensor embd;
    };
    Built build_into(int slot, size_t n_tokens, const Builder &builder) {
        Context &ctx0 = ctx0_[slot];
        ctx0.recreate();
        Built b;
        b.embd = ctx0.new_tensor_1d(GGML_TYPE_I32, n_tokens).set_name("embd");
        BuildContext bc{&ctx0, &b.embd, &memory_k, &memory_v, scratch_};
        auto built = builder(bc);
        b.gf = built.first;
        b.out = built.second;
        b.gf.build_forward_expand(b.out.result);  // Compute the graph
        return b;
    }
    struct Prebuilt {
        bool valid = false;
        size_t n_past = 0;
        const void *model_key = nullptr;
        int slot = 0;
        Built b;
    } pre_;
    Context ctx0_[2];
    int cur_ = 0;
    size_t n_embd_;
    std::shared_ptr<Buffer> scratch_[2];

   public:
    bool speculate = true;  // build the next single-token graph while the device runs (LLM_HOST_SPECULATE=0 disables)
    bool defer_end = false;  // set for the stages of an in-process split that produce no logits (Llama::start_session)
    bool pipeline_chunk = false;  // set by feed_prompt around every chunk but its last (see compute)
};

// crates/llm-base/src/model/common.rs:6-59
namespace common {
inline void read_last_token(InferenceSession &session, const Tensor &input_layer, size_t n_vocab, size_t n) {
    if (session.last_logits.size() != n_vocab) ggml::panic("last_logits size mismatch");
    input_layer.read_data(n_vocab * (n - 1) * sizeof(float), session.last_logits.data(), n_vocab * sizeof(float));
}
inline void extract_logits(OutputRequest &req, const Tensor &input_layer, size_t n_vocab, size_t n) {
    if (!req.all_logits) return;
    req.all_logits->assign(n_vocab * n, 0.0f);
    if (input_layer.nelements() != n_vocab * n) ggml::panic("extract_logits: element count mismatch");
    input_layer.read_data(0, req.all_logits->data(), n_vocab * n * sizeof(float));
}
inline void extract_embeddings(OutputRequest &req, const Tensor &embeddings_tensor, size_t n_embd, size_t n) {
    if (!req.embeddings) return;
    req.embeddings->assign(n_embd, 0.0f);
    std::vector<float> all(n_embd * n);
    if (embeddings_tensor.nelements() != n_embd * n) ggml::panic("extract_embeddings: element count mismatch");
    // The reference reads the host pointer (common.rs:52-56); the node is device-resident when offloaded, so
    // the mirror asks the backend for the device copy instead of reading stale host memory.
    ggml_hip_tensor_get(embeddings_tensor.ptr(), all.data(), 0, all.size() * sizeof(float));
    std::copy(all.begin() + n_embd * (n - 1), all.end(), req.embeddings->begin());
}
}  // namespace common

struct Hyperparameters {  // models/llama/src/lib.rs:399-416
    size_t n_vocab = 0, n_embd = 0, n_mult = 0, n_head = 0, n_head_kv = 0, n_layer = 0, n_rot = 0;
    int32_t file_type = 0;
};

struct Layer {  // models/llama/src/lib.rs:474-488
    Tensor attention_norm, wq, wk, wv, wo, ffn_norm, w1, w2, w3;
};

// TensorLoader (crates/llm-base/src/loader.rs:651-678): creates the named tensor in the model context and
// points its data at the caller's bytes (the mmap flavour: loader.rs:733-737).
class TensorLoader {
   public:
    TensorLoader(const llm_tensor_desc *t, int n) : descs_(t, t + n) {
        ctx_ = std::make_shared<Context>(
            Context::new_with_mmap((size_t)n * (sizeof(ggml_tensor) + sizeof(ggml_object) + 64) + 1024));
    }
    Tensor load(const std::string &name) {
        for (auto &d : descs_) {
            if (name != d.name) continue;
            Tensor t = d.n_dims == 1 ? ctx_->new_tensor_1d((ggml_type)d.type, (size_t)d.ne[0])
                                     : ctx_->new_tensor_2d((ggml_type)d.type, (size_t)d.ne[0], (size_t)d.ne[1]);
            t.set_data(d.data);
            t.set_name(name.c_str());
            return t;
        }
        fprintf(stderr, "llm: unknown tensor '%s'\n", name.c_str());  // LoadError::UnknownTensor
        abort();
    }
    std::shared_ptr<Context> finish() { return ctx_; }

   private:
    std::vector<llm_tensor_desc> descs_;
    std::shared_ptr<Context> ctx_;
};

class Llama {
   public:
    // models/llama/src/lib.rs:43-130
    Llama(Hyperparameters hp, ModelParameters params_, TensorLoader tl) : hyperparameters(hp), params(params_) {
        // layer split: this process owns [layer_begin, layer_end); the first stage also owns the
        // embedding table, the last one the final norm and the lm_head.  Default = the whole model = the reference.
        if (params.layer_end > hp.n_layer) params.layer_end = hp.n_layer;
        if (params.layer_begin >= params.layer_end) ggml::panic("empty layer range");
        if (is_first()) wte = tl.load("tok_embeddings.weight");
        const Backend backend = params.backend(0);
        if (is_last()) {
            norm = tl.load("norm.weight").transfer_to(backend);
            output = tl.load("output.weight").transfer_to(backend);
        }
        for (size_t i = params.layer_begin; i < params.layer_end; i++) {
            const Backend b = params.backend(i);
            auto name = [&](const char *s) { return "layers." + std::to_string(i) + "." + s; };
            Layer l;
            l.attention_norm = tl.load(name("attention_norm.weight")).transfer_to(b);
            l.wq = tl.load(name("attention.wq.weight")).transfer_to(b);
            l.wk = tl.load(name("attention.wk.weight")).transfer_to(b);
            l.wv = tl.load(name("attention.wv.weight")).transfer_to(b);
            l.wo = tl.load(name("attention.wo.weight")).transfer_to(b);
            l.ffn_norm = tl.load(name("ffn_norm.weight")).transfer_to(b);
            l.w1 = tl.load(name("feed_forward.w1.weight")).transfer_to(b);
            l.w2 = tl.load(name("feed_forward.w2.weight")).transfer_to(b);
            l.w3 = tl.load(name("feed_forward.w3.weight")).transfer_to(b);
            layers.push_back(l);
        }
        context = tl.finish();
    }

    bool is_first() const { return params.layer_begin == 0; }
    bool is_last() const { return params.layer_end == hyperparameters.n_layer; }
    size_t n_local_layers() const { return params.layer_end - params.layer_begin; }

    // on_slot >= 0: the session lives on that device slot (which the caller made current) instead of the model's — a sibling
    // slot of the same GPU: its own stream, shadows and plan workspace, the model's weights (llm_start_session_on)
    InferenceSession *start_session(const InferenceSessionConfig &config, int on_slot = -1) const {  // :133-141
        ModelParameters sp = params;
        if (on_slot >= 0) sp.main_device = on_slot;  // (keeps InferenceSession::new from re-initialising device 0, as for split stages)
        InferenceSession *s = new InferenceSession(config, sp, n_local_layers(), hyperparameters.n_embd,
                                                   hyperparameters.n_vocab);
        // stage hand-off buffers: persistent device tensors (like memory_k/v) that RCCL sends from / receives into
        if (!is_first() || !is_last()) s->make_stage_buffers(hyperparameters.n_embd, !is_first(), !is_last());
        s->defer_end = params.main_device >= 0 && !is_last();
        return s;
    }

    // The graph builder of Llama::evaluate (models/llama/src/lib.rs:166-362) for `input_len` tokens at position
    // `session_len`, as a value so that the session can also build the graph of the next call ahead of time.
    // logits_on_device: the logits node stays in HBM (like every other node) instead of being mirrored to the host after the
    // compute — taken for a multi-token evaluation that did not ask for all logits (feed_prompt: OutputRequest::default()):
    // read_last_token then fetches the one row it wants (65 MB of read-back per 512-token batch otherwise)
    InferenceSession::Builder make_builder(InferenceSession *session_ptr, size_t input_len, size_t session_len,
                                           bool logits_on_device = false) {
        return [this, session_ptr, input_len, session_len, logits_on_device](BuildContext &builder) {
            InferenceSession &session = *session_ptr;
            const size_t ctx_size = params.context_size;
            const size_t n_embd = hyperparameters.n_embd, n_head = hyperparameters.n_head,
                         n_head_kv = hyperparameters.n_head_kv, n_layer = hyperparameters.n_layer,
                         n_rot = hyperparameters.n_rot;
            const size_t n_embd_gqa = n_embd / (n_head / n_head_kv);

            Context &ctx0 = *builder.ctx0;
            const Tensor &embd = *builder.embd;
            (void)n_layer;
            // a stage of a layer split exchanges the residual through hand-off buffers of n_embd * rows floats (rows = n_batch
            // at first); ggml_view_1d has no bounds check (upstream neither), so the buffers must hold this call's rows
            if ((!is_first() || !is_last()) && input_len > session.stage_rows())
                ggml::panic("evaluate: the stage hand-off buffers hold fewer rows than this call (ensure_stage_rows not called)");
            Tensor input_layer = is_first()
                                     ? ctx0.op_get_rows(wte, embd)  // :170
                                     : ctx0.op_reshape_2d(ctx0.op_view_1d(session.stage_in, n_embd * input_len, 0), n_embd,
                                                          input_len);  // residual received from the previous stage
            ComputationGraph gf = ctx0.create_compute_graph();  // :172
            for (size_t il = 0; il < n_local_layers(); il++) {
                ctx0.set_offloading(params.should_offload(il));  // :175
                Tensor input_self_attention = input_layer.share();
                Tensor current;
                ctx0.use_scratch(builder.get_scratch(0));  // :180
                current = ctx0.op_rms_norm(input_layer);  // :183
                current = ctx0.op_mul(current, layers[il].attention_norm);  // :186
                const ggml::RoPEOverrides *overrides = params.has_rope_overrides ? &params.rope_overrides : nullptr;
                Tensor q_current =  // :191-204
                    ctx0.op_rope_inplace(ctx0.op_reshape_3d(ctx0.op_mul_mat(layers[il].wq, current), n_embd / n_head,
                                                            n_head, input_len),
                                         session_len, n_rot, 0, overrides)
                        .set_name("Qcur");
                Tensor k_current =  // :205-218
                    ctx0.op_rope_inplace(ctx0.op_reshape_3d(ctx0.op_mul_mat(layers[il].wk, current), n_embd / n_head,
                                                            n_head_kv, input_len),
                                         session_len, n_rot, 0, overrides)
                        .set_name("Kcur");
                Tensor v_current = ctx0.op_transpose(  // :222-226
                    ctx0.op_reshape_2d(ctx0.op_mul_mat(layers[il].wv, current), n_embd_gqa, input_len));
                const size_t kes = builder.memory_k->element_size(), ves = builder.memory_v->element_size();
                Tensor k = ctx0.op_view_1d(*builder.memory_k, input_len * n_embd_gqa,  // :228-232
                                           (kes * n_embd_gqa) * (il * ctx_size + session_len));
                Tensor v = ctx0.op_view_2d(*builder.memory_v, input_len, n_embd_gqa, ctx_size * ves,  // :234-240
                                           (il * ctx_size) * ves * n_embd_gqa + session_len * ves);
                gf.build_forward_expand(ctx0.op_cpy(k_current, k));  // :243
                gf.build_forward_expand(ctx0.op_cpy(v_current, v));  // :244
                Tensor q = ctx0.op_permute(q_current, 0, 2, 1, 3).set_name("Q");  // :246
                Tensor kk = ctx0.op_permute(  // :248-262
                                    ctx0.op_reshape_3d(ctx0.op_view_1d(*builder.memory_k,
                                                                       (session_len + input_len) * n_embd_gqa,
                                                                       il * ctx_size * kes * n_embd_gqa),
                                                       n_embd / n_head, n_head_kv, session_len + input_len),
                                    0, 2, 1, 3)
                                .set_name("K");
                Tensor k_q = ctx0.op_mul_mat(kk, q).set_name("KQ");  // :265
                Tensor kq_scale =  // :268-270
                    ctx0.new_f32(1.0f / std::sqrt((float)n_embd / (float)n_head)).set_name("1/sqrt(n_embd/n_head)");
                Tensor k_q_scaled = ctx0.op_scale_inplace(k_q, kq_scale).set_name("KQ_scaled");  // :271
                Tensor k_q_masked = ctx0.op_diag_mask_inf_inplace(k_q_scaled, session_len).set_name("KQ_masked");  // :274
                Tensor k_q_soft_max = ctx0.op_soft_max_inplace(k_q_masked).set_name("KQ_soft_max");  // :279
                Tensor vv = ctx0.op_view_3d(*builder.memory_v, session_len + input_len, n_embd / n_head, n_head_kv,  // :284-294
                                            ctx_size * ves, ctx_size * ves * n_embd / n_head,
                                            il * ctx_size * ves * n_embd_gqa)
                                .set_name("V");
                Tensor k_q_v = ctx0.op_mul_mat(vv, k_q_soft_max).set_name("KQV");  // :296
                Tensor k_q_v_merged = ctx0.op_permute(k_q_v, 0, 2, 1, 3).set_name("KQV_merged");  // :299
                current = ctx0.op_cpy(k_q_v_merged, ctx0.new_tensor_2d(GGML_TYPE_F32, n_embd, input_len))  // :302-307
                              .set_name("KQV_merged_contiguous");
                current = ctx0.op_mul_mat(layers[il].wo, current);  // :310
                ctx0.use_scratch(builder.get_scratch(1));  // :312
                Tensor input_feed_forward = ctx0.op_add(current, input_self_attention);  // :314
                current = ctx0.op_rms_norm(input_feed_forward);  // :318
                current = ctx0.op_mul(current, layers[il].ffn_norm);  // :321
                Tensor tmp = ctx0.op_mul_mat(layers[il].w3, current);  // :323
                current = ctx0.op_mul_mat(layers[il].w1, current);  // :325
                current = ctx0.op_silu(current);  // :328
                current = ctx0.op_mul(current, tmp);  // :330
                current = ctx0.op_mul_mat(layers[il].w2, current);  // :332
                current = ctx0.op_add(current, input_feed_forward);  // :334
                input_layer = current;  // :337
            }
            if (!is_last()) {  // hand the residual to the next stage through the persistent buffer
                ctx0.use_scratch(nullptr);
                Tensor out = ctx0.op_cpy(input_layer, ctx0.op_view_1d(session.stage_out, n_embd * input_len, 0));
                return std::make_pair(gf, GraphOutputs{out, out});
            }
            ctx0.use_scratch(builder.get_scratch(0));  // :340
            input_layer = ctx0.op_rms_norm(input_layer);  // :343
            input_layer = ctx0.op_mul(input_layer, norm);  // :346
            Tensor embedding_result = input_layer.share();
            ctx0.set_offloading(logits_on_device && params.use_gpu);  // :350 (false in the reference: the logits live on the host)
            input_layer = ctx0.op_mul_mat(output, input_layer);  // :352 lm_head
            ctx0.use_scratch(nullptr);  // :354
            return std::make_pair(gf, GraphOutputs{input_layer, embedding_result});
        };
    }

    // models/llama/src/lib.rs:144-368 — line numbers of the Rust builder are cited per step
    void evaluate(InferenceSession &session, const std::vector<TokenId> &input_tokens, OutputRequest &output_request) {
        const size_t input_len = input_tokens.size();
        const size_t session_len = session.n_past;
        const size_t ctx_size = params.context_size;
        const size_t n_vocab = hyperparameters.n_vocab, n_embd = hyperparameters.n_embd;

        InferenceSession *sp = &session;
        GraphOutputs outputs = session.compute(
            input_tokens, make_builder(sp, input_len, session_len, (input_len > 1 && !output_request.all_logits) || output_request.logits_on_device),
            [this, sp](size_t next_len) { return make_builder(sp, 1, next_len); }, this, ctx_size);
        if (!is_last()) return;
        // finish evaluation (:364-367)
        if (output_request.intermediate_chunk) return;  // (feed_prompt: only the last chunk's logits can be observed)
        if (!output_request.logits_on_device) common::read_last_token(session, outputs.result, n_vocab, input_len);
        common::extract_logits(output_request, outputs.result, n_vocab, input_len);
        common::extract_embeddings(output_request, outputs.embedding_result, n_embd, input_len);
    }

    Hyperparameters hyperparameters;
    ModelParameters params;
    Tensor wte, norm, output;
    std::vector<Layer> layers;
    std::shared_ptr<Context> context;  // must be kept alive for the model
};

}  // namespace llm

// ---------------------------------------------------------------------------------------------------
// C entry points
// ---------------------------------------------------------------------------------------------------
struct llm_ggml_file;
// A model is one Llama — or, for the ggml-style layer split of ONE session over several GPUs of this process (SURVEY.md
// section 8e; ModelParameters / ggml_cuda_set_tensor_split, crates/ggml/src/accelerator/mod.rs:68-77), one Llama per device
// slot, each owning a contiguous layer range [layer_begin, layer_end) and its K/V on its own device.  `llama` / `s` are
// the LAST stage's (final norm, lm_head, logits, token history), so everything that reads hyperparameters or logits is
// unchanged; evaluation walks the stages in order and hands the residual on with ggml_hip_copy_between_devices.
struct llm_model {
    llm::Llama *llama;
    llm_ggml_file *file = nullptr;  // mmap'd container the weights point into (llm_llama_load)
    std::vector<llm::Llama *> stages;  // empty: unsplit
    std::vector<int> devices;          // device slot of each stage
    int device = ggml_hip_get_main_device();  // unsplit: the slot that was current when the model was made (its weights live there)
};
struct llm_session {
    llm::InferenceSession *s;
    std::vector<llm::InferenceSession *> stage_sessions;  // parallel to llm_model::stages
    bool shares_streams = false;  // its stages of one GPU enqueue on one stream (ggml_hip_share_stream), undone when the session goes
    std::vector<int> devices;
    int device = 0;  // unsplit: the model's slot (K/V, shadows and plans of the session live there)
};

namespace {
std::atomic<int> g_split_sessions{0};  // live sessions of layer-split models (see llm_start_session)
// Model::evaluate for both kinds of model
// Every entry point leaves the caller's main device as it found it: an unsplit model runs on the slot it was loaded on
// (llm_model::device), a split one walks its stages' slots.
struct HomeDevice {
    int pinned = ggml_hip_thread_pinned_device();  // -1: the thread follows the process default, and must again afterwards
    int home = ggml_hip_get_main_device();
    void go(int d) const {
        if (ggml_hip_get_main_device() != d) ggml_hip_bind_thread_device(d);  // this thread only: the process default stays
    }
    ~HomeDevice() {
        if (pinned < 0)
            ggml_hip_unbind_thread_device();  // (a later ggml_hip_set_main_device from another thread reaches this one again)
        else
            go(home);
    }
};
void model_evaluate(llm_model *m, llm_session *s, const std::vector<llm::TokenId> &toks, llm::OutputRequest &req) {
    HomeDevice hd;
    if (m->stages.empty()) {
        hd.go(s->device);  // the session's slot: the model's, or a sibling slot of the same GPU (llm_start_session_on)
        if (!m->llama->is_first() || !m->llama->is_last())  // one stage of a per-process split (llm_amd/pipeline.py)
            s->s->ensure_stage_rows(m->llama->hyperparameters.n_embd, toks.size());
        m->llama->evaluate(*s->s, toks, req);
        return;
    }
    const size_t G = m->stages.size();
    const size_t hop_bytes = (size_t)m->llama->hyperparameters.n_embd * toks.size() * sizeof(float);
    for (size_t i = 0; i < G; i++) {  // a call with more tokens than n_batch: the hand-off buffers grow to it first
        hd.go(m->devices[i]);
        s->stage_sessions[i]->ensure_stage_rows(m->llama->hyperparameters.n_embd, toks.size());
    }
    for (size_t i = 0; i < G; i++) {
        ggml_hip_bind_thread_device(m->devices[i]);
        llm::InferenceSession &ss = *s->stage_sessions[i];
        if (i > 0) {  // the residual [n_embd, N] f32 of the previous stage -> this stage's hand-off buffer, device to device
            void *dst = ggml_hip_tensor_device_ptr(ss.stage_in.ptr());
            ggml_hip_bind_thread_device(m->devices[i - 1]);
            const void *src = ggml_hip_tensor_device_ptr(s->stage_sessions[i - 1]->stage_out.ptr());
            ggml_hip_copy_between_devices(m->devices[i], dst, m->devices[i - 1], src, hop_bytes);
            ggml_hip_bind_thread_device(m->devices[i]);
        }
        llm::OutputRequest none;
        m->stages[i]->evaluate(ss, toks, i + 1 == G ? req : none);
    }
}
// contiguous layer ranges for G stages in proportion to `split` (ggml's convention: device i takes split[i] / sum; all
// zero or NULL = equal shares); every stage gets at least one layer
std::vector<size_t> split_layers(size_t n_layer, int G, const float *split) {
    std::vector<double> w(G, 1.0);
    double sum = 0;
    bool any = false;
    for (int i = 0; i < G; i++) any = any || (split && split[i] > 0.0f);
    for (int i = 0; i < G; i++) {
        if (any) w[i] = std::max(0.0f, split[i]);
        sum += w[i];
    }
    std::vector<size_t> bounds(G + 1, 0);
    double acc = 0;
    for (int i = 0; i < G; i++) {
        acc += w[i];
        size_t b = (size_t)std::llround(acc / sum * (double)n_layer);
        b = std::max(b, bounds[i] + 1);                       // at least one layer per stage
        b = std::min(b, n_layer - (size_t)(G - 1 - i));       // ... and one left for each later stage
        bounds[i + 1] = b;
    }
    bounds[G] = n_layer;
    return bounds;
}
}  // namespace

extern "C" {

llm_model *llm_llama_new(const llm_llama_hparams *hp, const llm_model_params *mp, const llm_tensor_desc *tensors,
                         int n_tensors) {
    llm::Hyperparameters h;
    h.n_vocab = hp->n_vocab;
    h.n_embd = hp->n_embd;
    h.n_mult = hp->n_mult;
    h.n_head = hp->n_head;
    h.n_head_kv = hp->n_head_kv > 0 ? hp->n_head_kv : hp->n_head;
    h.n_layer = hp->n_layer;
    h.n_rot = hp->n_rot;
    h.file_type = hp->file_type;
    if (mp->n_gqa > 0 && h.n_layer >= 80) {  // models/llama/src/lib.rs:106-117: "temporary fix for 70B models"
        if (h.n_head % (size_t)mp->n_gqa != 0) ggml::panic("assuming 70B Llama2 model based on GQA == 8");  // the reference's assert_eq!
        h.n_head_kv = h.n_head / (size_t)mp->n_gqa;
    }
    llm::ModelParameters p;
    p.context_size = mp->context_size > 0 ? mp->context_size : 2048;
    p.use_gpu = mp->use_gpu != 0;
    p.gpu_layers = mp->gpu_layers;
    p.has_rope_overrides = mp->has_rope_overrides != 0;
    p.rope_overrides.frequency_scale = mp->rope_frequency_scale;
    p.rope_overrides.frequency_base = (size_t)mp->rope_frequency_base;
    p.layer_begin = mp->layer_begin > 0 ? (size_t)mp->layer_begin : 0;
    p.layer_end = mp->layer_end > 0 ? (size_t)mp->layer_end : (size_t)-1;
    if (!p.use_gpu) {
        fprintf(stderr, "llm_llama_new: use_gpu=0 requested, but libggml_hip has no CPU compute path\n");
        abort();
    }
    // Layer split inside this process: more than one device slot with a positive share in ggml_hip_set_layer_split (the
    // explicit-length sibling of the reference's split hook, crates/ggml/sys/src/cuda.rs:11, which itself carries ONE float:
    // accelerator/mod.rs:74-75), or GGML_HIP_LAYER_SPLIT=G for equal shares.
    // A caller that passes its own layer range (one process per GPU, llm_amd/pipeline.py) is left alone.
    std::vector<int> slots;     // the device slots that take part, in order
    std::vector<float> shares;  // their fractions (all zero = equal)
    {
        float split[16] = {0};
        const int nslot = std::min(ggml_hip_get_layer_split(split, 16), ggml_hip_device_count());
        for (int i = 0; i < nslot; i++)
            if (split[i] > 0.0f) {
                slots.push_back(i);
                shares.push_back(split[i]);
            }
        if (slots.size() < 2) {
            slots.clear();
            shares.clear();
            if (const char *v = getenv("GGML_HIP_LAYER_SPLIT"))
                for (int i = 0; i < std::min(atoi(v), ggml_hip_device_count()); i++) {
                    slots.push_back(i);
                    shares.push_back(0.0f);
                }
        }
    }
    const bool whole = p.layer_begin == 0 && p.layer_end >= (size_t)h.n_layer;  // the caller did not ask for a stage
    while (slots.size() > (size_t)h.n_layer) {
        slots.pop_back();
        shares.pop_back();
    }
    llm_model *m = new llm_model();
    if (slots.size() > 1 && whole) {
        const int G = (int)slots.size();
        const std::vector<size_t> bounds = split_layers(h.n_layer, G, shares.data());
        const int home = ggml_hip_get_main_device();
        for (int i = 0; i < G; i++) {
            llm::ModelParameters ps = p;
            ps.layer_begin = bounds[i];
            ps.layer_end = bounds[i + 1];
            ps.main_device = slots[i];
            ggml_hip_bind_thread_device(slots[i]);
            m->stages.push_back(new llm::Llama(h, ps, llm::TensorLoader(tensors, n_tensors)));
            m->devices.push_back(slots[i]);
        }
        ggml_hip_bind_thread_device(home);
        m->llama = m->stages.back();
        return m;
    }
    llm::TensorLoader tl(tensors, n_tensors);
    llm::ModelParameters pu = p;
    // a whole model made while a slot other than 0 is this thread's: its sessions stay on that slot (the reference's
    // InferenceSession::new always initialises device 0 — it knows one device)
    if (pu.main_device < 0 && m->device != 0) pu.main_device = m->device;
    m->llama = new llm::Llama(h, pu, std::move(tl));
    return m;
}
// The layer ranges an in-process split over G device slots gets for the fractions `split` (ggml_cuda_set_tensor_split's
// convention: slot i takes split[i] / sum; NULL or all zero = equal shares): bounds_out[0..G], stage i = layers
// [bounds_out[i], bounds_out[i+1]).  Pure host arithmetic (no device needed): what llm_llama_new applies.
void llm_split_layers(int n_layer, int G, const float *split, int *bounds_out) {
    const int Gu = std::max(1, std::min(G, n_layer));  // more slots than layers: the surplus slots get empty ranges at the end
    const std::vector<size_t> b = split_layers((size_t)std::max(1, n_layer), Gu, split);
    for (int i = 0; i <= G; i++) bounds_out[i] = i <= Gu ? (int)b[(size_t)i] : n_layer;
}
// the layer range and device slot of every stage of a model (1 entry for an unsplit one); returns the stage count
int llm_model_stages(const llm_model *m, int *layer_begin, int *layer_end, int *device, int cap) {
    if (m->stages.empty()) {
        if (cap > 0) {
            if (layer_begin) layer_begin[0] = (int)m->llama->params.layer_begin;
            if (layer_end) layer_end[0] = (int)std::min(m->llama->params.layer_end, m->llama->hyperparameters.n_layer);
            if (device) device[0] = m->device;
        }
        return 1;
    }
    for (size_t i = 0; i < m->stages.size() && (int)i < cap; i++) {
        if (layer_begin) layer_begin[i] = (int)m->stages[i]->params.layer_begin;
        if (layer_end) layer_end[i] = (int)m->stages[i]->params.layer_end;
        if (device) device[i] = m->devices[i];
    }
    return (int)m->stages.size();
}
void llm_model_free(llm_model *m) {
    if (!m) return;
    if (!m->stages.empty()) {
        const int home = ggml_hip_get_main_device();
        for (size_t i = 0; i < m->stages.size(); i++) {
            ggml_hip_bind_thread_device(m->devices[i]);
            delete m->stages[i];
        }
        ggml_hip_bind_thread_device(home);
        m->llama = nullptr;
    }
    if (m->llama) {
        HomeDevice hd;
        hd.go(m->device);
        delete m->llama;
    }
    if (m->file) llm_ggml_file_close(m->file);
    delete m;
}

// ---------------------------------------------------------------------------------------------------------------
// GGML / GGMF / GGJT container reader (crates/ggml/src/format/loader.rs:160-281), tensor data left in the mapping
// ---------------------------------------------------------------------------------------------------------------
struct llm_ggml_file {
    int fd = -1;
    const uint8_t *base = nullptr;
    size_t size = 0;
    int container = 0, version = 0;
    llm_llama_hparams hp{};
    struct Tok {
        const char *p;
        uint32_t len;
        float score;
    };
    std::vector<Tok> vocab;
    struct Ten {
        std::string name;
        int32_t type, n_dims;
        int64_t ne[2];
        size_t offset;
    };
    std::vector<Ten> tensors;
};

namespace {
struct Cursor {
    const uint8_t *p, *end;
    bool ok = true;
    uint32_t u32() {
        uint32_t v = 0;
        if ((size_t)(end - p) < 4) {
            ok = false;
            return v;
        }
        memcpy(&v, p, 4);
        p += 4;
        return v;
    }
    int32_t i32() { return (int32_t)u32(); }
    float f32() {
        const uint32_t u = u32();
        float v;
        memcpy(&v, &u, 4);
        return v;
    }
    const uint8_t *bytes(size_t n) {
        if ((size_t)(end - p) < n) {
            ok = false;
            return nullptr;
        }
        const uint8_t *r = p;
        p += n;
        return r;
    }
};
size_t ggml_file_type_size(int32_t t) {  // bytes per block; blck via ggml_blck_size
    return ggml_type_size((ggml_type)t);
}
}  // namespace

llm_ggml_file *llm_ggml_file_open(const char *path) {
    auto fail = [&](llm_ggml_file *f, const char *why) -> llm_ggml_file * {
        fprintf(stderr, "llm_ggml_file_open(%s): %s\n", path, why);
        if (f) llm_ggml_file_close(f);
        return nullptr;
    };
    llm_ggml_file *f = new llm_ggml_file();
    f->fd = open(path, O_RDONLY);
    if (f->fd < 0) return fail(f, "cannot open file");
    struct stat st;
    if (fstat(f->fd, &st) != 0 || st.st_size < 8) return fail(f, "cannot stat file / file too small");
    f->size = (size_t)st.st_size;
    void *m = mmap(nullptr, f->size, PROT_READ, MAP_SHARED, f->fd, 0);
    if (m == MAP_FAILED) return fail(f, "mmap failed");
    f->base = (const uint8_t *)m;
    Cursor c{f->base, f->base + f->size};
    // ContainerType::read (crates/ggml/src/lib.rs:58-84)
    const uint32_t magic = c.u32();
    if (magic == 0x67676d6cu) {  // 'ggml': unversioned
        f->container = 0;
    } else if (magic == 0x67676d66u || magic == 0x67676a74u || magic == 0x67676c61u) {
        f->container = magic == 0x67676d66u ? 1 : magic == 0x67676a74u ? 2 : 3;
        f->version = (int)c.u32();
    } else {
        return fail(f, "LoadError::InvalidMagic");
    }
    const bool ok_version = f->container == 0 || (f->container == 1 && f->version == 1) ||
                            (f->container == 2 && f->version >= 1 && f->version <= 3) ||
                            (f->container == 3 && f->version == 1);
    if (!ok_version) return fail(f, "LoadError::InvalidFormatVersion");
    // LLaMA hyperparameters (models/llama/src/lib.rs:425-447)
    f->hp.n_vocab = c.i32();
    f->hp.n_embd = c.i32();
    f->hp.n_mult = c.i32();
    f->hp.n_head = c.i32();
    f->hp.n_layer = c.i32();
    f->hp.n_rot = c.i32();
    f->hp.file_type = c.i32();
    f->hp.n_head_kv = f->hp.n_head;
    if (!c.ok || f->hp.n_vocab < 0 || f->hp.n_embd <= 0 || f->hp.n_head <= 0 || f->hp.n_layer <= 0)
        return fail(f, "LoadError: bad hyperparameters");
    for (int i = 0; i < f->hp.n_vocab; i++) {  // loader.rs:187-203
        const uint32_t len = c.u32();
        const uint8_t *tok = c.bytes(len);
        float score = 0.0f;
        if (f->container == 1 || f->container == 2) score = c.f32();
        if (!c.ok) return fail(f, "LoadError: truncated vocabulary");
        f->vocab.push_back({(const char *)tok, len, score});
    }
    const bool align = f->container >= 2;  // loader.rs:206-212
    while (c.p < c.end) {                  // load_weights, loader.rs:219-281
        llm_ggml_file::Ten t;
        t.n_dims = c.i32();
        const int32_t name_len = c.i32();
        const uint32_t ftype = c.u32();
        if (!c.ok || t.n_dims < 1 || t.n_dims > 2 || name_len < 0) return fail(f, "LoadError::InvariantBroken (n_dims <= 2)");
        t.ne[0] = t.ne[1] = 1;
        int64_t n_elements = 1;
        for (int i = 0; i < t.n_dims; i++) {
            t.ne[i] = c.i32();
            if (t.ne[i] <= 0) return fail(f, "LoadError: bad tensor dimension");
            n_elements *= t.ne[i];
        }
        const uint8_t *nm = c.bytes((size_t)name_len);
        if (!c.ok) return fail(f, "LoadError: truncated tensor header");
        t.name.assign((const char *)nm, (size_t)name_len);
        if (ftype >= GGML_TYPE_COUNT || ggml_blck_size((ggml_type)ftype) == 0 || ggml_file_type_size((int32_t)ftype) == 0)
            return fail(f, "LoadError::UnsupportedElementType");
        t.type = (int32_t)ftype;
        if ((t.type == GGML_TYPE_Q4_0 || t.type == GGML_TYPE_Q4_1) && t.ne[0] % 64 != 0)
            return fail(f, "LoadError::InvariantBroken (dims[0] % 64 == 0)");
        size_t off = (size_t)(c.p - f->base);
        if (align) off = (off + 31) & ~(size_t)31;
        const size_t n_bytes = ggml_file_type_size(t.type) * (size_t)n_elements / (size_t)ggml_blck_size((ggml_type)t.type);
        if (off + n_bytes > f->size) return fail(f, "LoadError: tensor data past the end of the file");
        t.offset = off;
        f->tensors.push_back(t);
        c.p = f->base + off + n_bytes;
    }
    return f;
}
void llm_ggml_file_close(llm_ggml_file *f) {
    if (!f) return;
    if (f->base) munmap((void *)f->base, f->size);
    if (f->fd >= 0) close(f->fd);
    delete f;
}
void llm_ggml_file_info(const llm_ggml_file *f, int *container, int *version, llm_llama_hparams *hp, int *n_tensors,
                        int *n_vocab_entries) {
    if (container) *container = f->container;
    if (version) *version = f->version;
    if (hp) *hp = f->hp;
    if (n_tensors) *n_tensors = (int)f->tensors.size();
    if (n_vocab_entries) *n_vocab_entries = (int)f->vocab.size();
}
int llm_ggml_file_tensor(const llm_ggml_file *f, int i, llm_tensor_desc *d) {
    if (i < 0 || i >= (int)f->tensors.size()) return -1;
    const auto &t = f->tensors[(size_t)i];
    d->name = t.name.c_str();
    d->type = t.type;
    d->n_dims = t.n_dims;
    d->ne[0] = t.ne[0];
    d->ne[1] = t.ne[1];
    d->data = (void *)(f->base + t.offset);
    return 0;
}
int llm_ggml_file_vocab(const llm_ggml_file *f, int i, char *buf, int cap, float *score) {
    if (i < 0 || i >= (int)f->vocab.size()) return -1;
    const auto &t = f->vocab[(size_t)i];
    if (buf && cap > 0) memcpy(buf, t.p, std::min<size_t>((size_t)cap, t.len));
    if (score) *score = t.score;
    return (int)t.len;
}
llm_model *llm_llama_load(const char *path, const llm_model_params *params) {
    llm_ggml_file *f = llm_ggml_file_open(path);
    if (!f) return nullptr;
    std::vector<llm_tensor_desc> descs(f->tensors.size());
    for (size_t i = 0; i < descs.size(); i++) llm_ggml_file_tensor(f, (int)i, &descs[i]);
    llm_model *m = llm_llama_new(&f->hp, params, descs.data(), (int)descs.size());
    m->file = f;
    return m;
}
llm_session *llm_start_session(llm_model *m, const llm_session_config *cfg) {
    llm::InferenceSessionConfig c;
    if (cfg) {
        c.memory_k_type = (ggml_type)cfg->memory_k_type;
        c.memory_v_type = (ggml_type)cfg->memory_v_type;
        c.n_batch = cfg->n_batch > 0 ? (size_t)cfg->n_batch : 8;
        c.n_threads = cfg->n_threads > 0 ? (size_t)cfg->n_threads : 8;
    }
    llm_session *s = new llm_session();
    if (!m->stages.empty()) {
        const int home = ggml_hip_get_main_device();
        for (size_t i = 0; i < m->stages.size(); i++) {
            ggml_hip_bind_thread_device(m->devices[i]);
            s->stage_sessions.push_back(m->stages[i]->start_session(c));
        }
        ggml_hip_bind_thread_device(home);
        // the stages of ONE split session run one after the other: on a GPU that hosts several of them (virtual slots) they are one
        // sharer of its compute units, not G (include/ggml_hip.h "serial_stage_slots"); with a second split session alive they are not
        const bool only_split_session = g_split_sessions.fetch_add(1) == 0;
        ggml_hip_set_option("serial_stage_slots", only_split_session ? (int)m->stages.size() : 0);
        // ... and stages that share a physical GPU share its queue too: the residual then crosses a stage boundary inside one stream
        // instead of through an event of another queue (ggml_hip_share_stream; it declines slots of different GPUs)
        if (only_split_session)
            for (size_t i = 1; i < m->devices.size(); i++)
                for (size_t j = 0; j < i; j++)
                    if (ggml_hip_share_stream(m->devices[i], m->devices[j])) break;
        s->shares_streams = only_split_session;
        s->s = s->stage_sessions.back();
        s->devices = m->devices;
        return s;
    }
    HomeDevice hd;
    hd.go(m->device);
    s->device = m->device;
    struct AfterNew {  // GGML_HIP_SESSION_SLOTS: the backend may have assigned this thread a sibling slot while the K/V memory was created
        llm_session *s;
        ~AfterNew() {
            if (s->s && ggml_hip_thread_session_slot() >= 0) s->device = ggml_hip_thread_session_slot();
        }
    } after_new{s};
    s->s = m->llama->start_session(c);
    return s;
}
// Several sessions of ONE model on one GPU (crates/llm-base/src/inference_session.rs:43-48: a session is Send;
// model/mod.rs:275-276: "spawn several sessions for one model"): every session gets its own device SLOT on the model's GPU — its
// own stream, arena shadows, K/V memory, plan cache and workspace — and reads the model's weights where the model's slot put them
// (slots of one physical device may read each other's weights, backend_state.inc extra_of).  Sessions on different slots run
// concurrently (per-slot locks; the host arenas are an event log, no cross-slot locking per token), so two decode streams fill
// each other's kernel-boundary and latency gaps.  `slot` must drive the same physical device as the model's slot
// (GGML_HIP_VIRTUAL_DEVICES=n provides n slots on a 1-GPU box; slot = -1: the model's own).  Unsplit models only.
llm_session *llm_start_session_on(llm_model *m, const llm_session_config *cfg, int slot) {
    if (slot < 0 || !m->stages.empty() || slot == m->device) return llm_start_session(m, cfg);
    // a sibling slot reads the model's ONE copy of the weights: it must drive the GPU that holds them (checked here, before any
    // K/V memory is allocated on the wrong one; the backend would only notice at the first evaluation)
    if (ggml_hip_slot_physical_device(slot) < 0 || ggml_hip_slot_physical_device(slot) != ggml_hip_slot_physical_device(m->device)) {
        fprintf(stderr, "llm_start_session_on: slot %d does not drive the GPU of the model's slot %d\n", slot, m->device);
        return nullptr;
    }
    llm::InferenceSessionConfig c;
    if (cfg) {
        c.memory_k_type = (ggml_type)cfg->memory_k_type;
        c.memory_v_type = (ggml_type)cfg->memory_v_type;
        c.n_batch = cfg->n_batch > 0 ? (size_t)cfg->n_batch : 8;
        c.n_threads = cfg->n_threads > 0 ? (size_t)cfg->n_threads : 8;
    }
    llm_session *s = new llm_session();
    HomeDevice hd;
    hd.go(slot);
    s->device = slot;
    s->s = m->llama->start_session(c, slot);
    return s;
}
// test / restore hook: the session continues at position n_past (what InferenceSession::from_snapshot does with the snapshot's
// npast, inference_session.rs:640) — the caller has put the K/V of the positions before it in place (llm_session_kv)
void llm_session_seek(llm_session *s, int n_past) {
    s->s->n_past = (size_t)n_past;
    for (auto *st : s->stage_sessions) st->n_past = (size_t)n_past;
    s->s->drop_prebuilt();
}
// 0 = the reference's own call sequence inside InferenceSession::compute (build the graph, then ggml_graph_compute, synchronously:
// inference_session.rs:220-295); 1 (default; env LLM_HOST_SPECULATE) = the patched sequence that builds the NEXT token's graph
// between ggml_hip_graph_compute_begin / _end while the device runs
void llm_session_set_speculate(llm_session *s, int on) {
    s->s->speculate = on != 0;
    for (auto *st : s->stage_sessions) st->speculate = on != 0;
    s->s->drop_prebuilt();
}
void llm_session_free(llm_session *s) {
    if (!s) return;
    if (!s->stage_sessions.empty()) {
        const int home = ggml_hip_get_main_device();
        for (size_t i = 0; i < s->stage_sessions.size(); i++) {
            ggml_hip_bind_thread_device(s->devices[i]);
            delete s->stage_sessions[i];
        }
        ggml_hip_bind_thread_device(home);
        s->s = nullptr;
        if (s->shares_streams)
            for (size_t i = 1; i < s->devices.size(); i++) ggml_hip_share_stream(s->devices[i], -1);
        g_split_sessions.fetch_sub(1);
        ggml_hip_set_option("serial_stage_slots", 0);
    }
    if (s->s) {
        HomeDevice hd;
        hd.go(s->device);
        delete s->s;
    }
    delete s;
}
void llm_evaluate(llm_model *m, llm_session *s, const int32_t *tokens, int n, float *all_logits, float *embeddings) {
    std::vector<llm::TokenId> toks(tokens, tokens + n);
    std::vector<float> logits, emb;
    llm::OutputRequest req;
    if (all_logits) req.all_logits = &logits;
    if (embeddings) req.embeddings = &emb;
    model_evaluate(m, s, toks, req);
    if (all_logits) memcpy(all_logits, logits.data(), logits.size() * sizeof(float));
    if (embeddings) memcpy(embeddings, emb.data(), emb.size() * sizeof(float));
}
void llm_feed_prompt(llm_model *m, llm_session *s, const int32_t *tokens, int n) {
    // inference_session.rs:311-316: ContextFull check, then chunks(n_batch)
    if (s->s->n_past + (size_t)n >= m->llama->params.context_size) {
        fprintf(stderr, "llm_feed_prompt: InferenceError::ContextFull\n");
        abort();
    }
    const size_t nb = s->s->config.n_batch;
    for (size_t i = 0; i < (size_t)n; i += nb) {
        const size_t len = std::min(nb, (size_t)n - i);
        std::vector<llm::TokenId> batch(tokens + i, tokens + i + len);
        // every chunk but the last is only enqueued (unsplit models): the host builds and matches the next chunk's graph while the
        // device runs this one, instead of waiting for it and for a row of logits nobody can see
        llm::OutputRequest req;
        static const bool pipeline_on = !(getenv("LLM_HOST_PIPELINE_CHUNKS") && atoi(getenv("LLM_HOST_PIPELINE_CHUNKS")) == 0);
        const bool more = pipeline_on && i + len < (size_t)n && m->stages.empty() && len > 1;
        req.intermediate_chunk = more;
        s->s->pipeline_chunk = more;
        model_evaluate(m, s, batch, req);
        s->s->pipeline_chunk = false;
        for (auto tk : batch) s->s->tokens.push_back(tk);
    }
}
// Greedy sampling on the host: the index the loop `best = 0; if (l[i] > l[best]) best = i` returns (first maximum; NaNs never win;
// a NaN in front keeps index 0).  The AVX2 version finds the maximum with the same `>` rule lane by lane, then the first
// position that holds it: 7 -> ~1.5 us for 32000 logits, on the critical path of every decoded token.
static size_t argmax_first_scalar(const float *l, size_t n) {
    size_t best = 0;
    for (size_t i = 1; i < n; i++)
        if (l[i] > l[best]) best = i;
    return best;
}
#if defined(__x86_64__)
__attribute__((target("avx2"))) static size_t argmax_first_avx2(const float *l, size_t n) {
    if (n < 16) return argmax_first_scalar(l, n);
    __m256 mx = _mm256_set1_ps(l[0]);
    size_t i = 0;
    for (; i + 8 <= n; i += 8) {
        const __m256 v = _mm256_loadu_ps(l + i);
        mx = _mm256_blendv_ps(mx, v, _mm256_cmp_ps(v, mx, _CMP_GT_OQ));
    }
    float lanes[8];
    _mm256_storeu_ps(lanes, mx);
    float m = l[0];
    for (int k = 0; k < 8; k++) m = lanes[k] > m ? lanes[k] : m;
    for (; i < n; i++) m = l[i] > m ? l[i] : m;
    if (!(m == m)) return 0;  // l[0] is NaN: nothing compares greater
    const __m256 mv = _mm256_set1_ps(m);
    for (i = 0; i + 8 <= n; i += 8) {
        const int mask = _mm256_movemask_ps(_mm256_cmp_ps(_mm256_loadu_ps(l + i), mv, _CMP_EQ_OQ));
        if (mask) return i + (size_t)__builtin_ctz((unsigned)mask);
    }
    for (; i < n; i++)
        if (l[i] == m) return i;
    return 0;
}
#endif
static size_t argmax_first(const float *l, size_t n) {
#if defined(__x86_64__)
    static const bool have_avx2 = __builtin_cpu_supports("avx2");
    if (have_avx2) return argmax_first_avx2(l, n);
#endif
    return argmax_first_scalar(l, n);
}
// test hook: which = 0 the dispatching version, 1 the scalar loop
int llm_argmax_first(const float *l, int n, int which) { return (int)(which ? argmax_first_scalar(l, (size_t)n) : argmax_first(l, (size_t)n)); }

int32_t llm_infer_next_token_greedy(llm_model *m, llm_session *s) {
    // inference_session.rs:381-424 with the sampler chain replaced by argmax over last_logits
    if (s->s->n_past + 1 >= m->llama->params.context_size) {
        fprintf(stderr, "llm_infer_next_token: InferenceError::ContextFull\n");
        abort();
    }
    const double t0 = llm::InferenceSession::now_ns();
    const std::vector<float> &l = s->s->last_logits;
    const llm::TokenId next = (llm::TokenId)argmax_first(l.data(), l.size());
    s->s->tokens.push_back(next);
    llm::OutputRequest req;
    const double t1 = llm::InferenceSession::now_ns();
    model_evaluate(m, s, std::vector<llm::TokenId>{next}, req);
    llm::InferenceSession::host_ns_add(5, t1 - t0);                                // argmax
    llm::InferenceSession::host_ns_add(6, llm::InferenceSession::now_ns() - t1);  // evaluate, all of it
    return next;
}
// The reference's token step with the SHAPE of its default sampler (crates/llm-base/src/samplers.rs:97-188: repetition penalty
// 1.30 over the last 64 tokens, top-k 40, [tail-free, typical, top-p: act on the <= 40 survivors], temperature 0.80) instead of
// greedy argmax: what a caller that samples gets from the unchanged sequence sample -> evaluate (inference_session.rs:381-424).
//   device_topk = 0: as the reference does it — all n_vocab logits are read back by the evaluation (model/common.rs:6-19), the
//                    candidates are found on the host (std::partial_sort);
//   device_topk = 1: the evaluation leaves the logits in HBM (OutputRequest::logits_on_device) and the k best + the raw logits of
//                    the penalty window come from llm_session_topk: k + 64 pairs instead of 128 KB per token.
// Same candidates, same arithmetic, same generator: both settings draw the same tokens (tests/test_device_tools_gpu.py).
// `rng`: the caller's xorshift64* state.  Not llm-samplers' chain itself — a measurement of what the read-back costs a sampler.
int32_t llm_infer_next_token_topk(llm_model *m, llm_session *s, int k, float temperature, uint64_t *rng, int device_topk) {
    if (s->s->n_past + 1 >= m->llama->params.context_size) {
        fprintf(stderr, "llm_infer_next_token_topk: InferenceError::ContextFull\n");
        abort();
    }
    const size_t V = s->s->last_logits.size();
    if (k < 1 || (size_t)k > V || !rng) return -1;
    const std::vector<llm::TokenId> &hist = s->s->tokens;
    std::vector<int32_t> window;  // the last 64 tokens, each once
    for (size_t i = hist.size() > 64 ? hist.size() - 64 : 0; i < hist.size(); i++)
        if (std::find(window.begin(), window.end(), (int32_t)hist[i]) == window.end()) window.push_back((int32_t)hist[i]);
    std::vector<float> vals(k + window.size());
    std::vector<int32_t> ids(k + window.size());
    if (device_topk) {
        if (llm_session_topk(s, k, window.empty() ? nullptr : window.data(), (int)window.size(), vals.data(), ids.data()) != 0) return -1;
    } else {
        const std::vector<float> &l = s->s->last_logits;
        std::vector<int32_t> order(V);
        for (size_t i = 0; i < V; i++) order[i] = (int32_t)i;
        std::partial_sort(order.begin(), order.begin() + k, order.end(),
                          [&](int32_t a, int32_t b) { return l[a] > l[b] || (l[a] == l[b] && a < b); });
        for (int i = 0; i < k; i++) { ids[i] = order[i]; vals[i] = l[order[i]]; }
        for (size_t i = 0; i < window.size(); i++) { ids[k + i] = window[i]; vals[k + i] = l[window[i]]; }
    }
    // candidates = top-k and the window's tokens (each once), penalised, the k best of them kept
    struct Cand { float v; int32_t id; };
    std::vector<Cand> c;
    for (size_t i = 0; i < ids.size(); i++) {
        bool dup = false;
        for (auto &x : c) dup = dup || x.id == ids[i];
        if (dup) continue;
        float v = vals[i];
        if (std::find(window.begin(), window.end(), ids[i]) != window.end()) v = v > 0.0f ? v / 1.30f : v * 1.30f;
        c.push_back({v, ids[i]});
    }
    std::sort(c.begin(), c.end(), [](const Cand &a, const Cand &b) { return a.v > b.v || (a.v == b.v && a.id < b.id); });
    if (c.size() > (size_t)k) c.resize(k);
    double sum = 0.0;
    std::vector<double> pr(c.size());
    for (size_t i = 0; i < c.size(); i++) sum += (pr[i] = std::exp((double)(c[i].v - c[0].v) / (double)temperature));
    uint64_t x = *rng;  // xorshift64*
    x ^= x >> 12; x ^= x << 25; x ^= x >> 27;
    *rng = x;
    const double u = (double)((x * 0x2545F4914F6CDD1Dull) >> 11) / 9007199254740992.0 * sum;
    double acc = 0.0;
    llm::TokenId next = (llm::TokenId)c.back().id;
    for (size_t i = 0; i < c.size(); i++) {
        acc += pr[i];
        if (u < acc) { next = (llm::TokenId)c[i].id; break; }
    }
    s->s->tokens.push_back(next);
    llm::OutputRequest req;
    req.logits_on_device = device_topk != 0;
    model_evaluate(m, s, std::vector<llm::TokenId>{next}, req);
    return next;
}
// n greedy tokens with the sampler on the device (ggml_hip_decode_greedy_chain): the same ids and the
// same final logits as n calls of llm_infer_next_token_greedy, without a logits read-back and host sync per token.
// Falls back to that loop when the backend cannot chain (last evaluation not a single-token fused-plan run, layer
// split stage, ...).  Returns the number of tokens written to out (n).
int llm_infer_tokens_greedy_device(llm_model *m, llm_session *s, int n, int32_t *out) {
    if (n < 1) return 0;
    if (s->s->n_past + (size_t)n >= m->llama->params.context_size) {
        fprintf(stderr, "llm_infer_tokens_greedy_device: InferenceError::ContextFull\n");
        abort();
    }
    int done = 0;
    // the chain continues a single-token fused-plan run; after a prompt chunk (or a fresh session) one normal step arms it
    for (int attempt = 0; attempt < 2 && done < n; attempt++) {
        if (m->stages.empty() && s->s->last_graph &&
            ggml_hip_decode_greedy_chain(s->s->last_graph, n - done, out + done, s->s->last_logits.data()) == 0) {
            s->s->advance_device_chain((const llm::TokenId *)(out + done), (size_t)(n - done));
            return n;
        }
        out[done++] = llm_infer_next_token_greedy(m, s);
    }
    for (; done < n; done++) out[done] = llm_infer_next_token_greedy(m, s);
    return n;
}
// Accumulated host nanoseconds per phase (see InferenceSession::host_ns; [5] greedy argmax, [6] evaluate as a whole);
// reset != 0 clears the accumulators after the read.
void llm_host_timing(double *out8, int reset) {
    for (int i = 0; i < 8; i++) out8[i] = (double)llm::InferenceSession::host_ns[i].load(std::memory_order_relaxed);
    if (reset) for (int i = 0; i < 8; i++) llm::InferenceSession::host_ns[i].store(0, std::memory_order_relaxed);
}
int llm_session_rewind(llm_session *s, int num) {
    if ((size_t)num >= s->s->n_past) return -1;  // RewindError::NotEnoughTokens
    if ((size_t)num <= s->s->tokens.size()) s->s->tokens.resize(s->s->tokens.size() - (size_t)num);
    s->s->n_past -= (size_t)num;
    for (size_t i = 0; i + 1 < s->stage_sessions.size(); i++) s->stage_sessions[i]->n_past -= (size_t)num;
    return 0;
}
const float *llm_session_last_logits(const llm_session *s) { return s->s->last_logits.data(); }
int llm_session_n_past(const llm_session *s) { return (int)s->s->n_past; }
int llm_model_n_vocab(const llm_model *m) { return (int)m->llama->hyperparameters.n_vocab; }
void llm_session_last_graph_stats(const llm_session *s, int *n_nodes, int *n_leafs) {
    if (n_nodes) *n_nodes = s->s->last_n_nodes;
    if (n_leafs) *n_leafs = s->s->last_n_leafs;
}

// Device addresses of the layer-split hand-off buffers of a session (NULL when the stage has none): the residual
// [n_embd * n_tokens] f32 is received into *in_dev before llm_evaluate and is in *out_dev after it.
void llm_session_stage_buffers(llm_session *s, void **in_dev, void **out_dev, size_t *nbytes) {
    HomeDevice hd;
    if (s->stage_sessions.empty()) hd.go(s->device);
    if (in_dev) *in_dev = s->s->stage_in.is_null() ? nullptr : ggml_hip_tensor_device_ptr(s->s->stage_in.ptr());
    if (out_dev) *out_dev = s->s->stage_out.is_null() ? nullptr : ggml_hip_tensor_device_ptr(s->s->stage_out.ptr());
    if (nbytes) *nbytes = !s->s->stage_in.is_null() ? s->s->stage_in.nbytes() : !s->s->stage_out.is_null() ? s->s->stage_out.nbytes() : 0;
}

// Session K/V memory as raw bytes (which: 0 = memory_k, 1 = memory_v) — the payload of the reference's
// InferenceSnapshot (inference_session.rs:599-646), which reads `memory_k.data()` on the host; here the
// authoritative copy lives on the device, so the snapshot goes through the backend.  set=0 reads, set=1 writes.
size_t llm_session_kv(llm_session *s, int which, int set, void *buf, size_t nbytes) {
    if (!s->stage_sessions.empty()) {  // the stages' caches in layer order = the whole model's layout (layer-major)
        size_t total = 0, off = 0;
        for (auto *ss : s->stage_sessions) total += (which == 0 ? ss->memory_k : ss->memory_v).nbytes();
        if (!buf) return total;
        const int home = ggml_hip_get_main_device();
        for (size_t i = 0; i < s->stage_sessions.size() && off < nbytes; i++) {
            ggml::Tensor &t = which == 0 ? s->stage_sessions[i]->memory_k : s->stage_sessions[i]->memory_v;
            const size_t n = std::min(t.nbytes(), nbytes - off);
            ggml_hip_bind_thread_device(s->devices[i]);
            if (set)
                ggml_hip_tensor_set(t.ptr(), (char *)buf + off, 0, n);
            else
                ggml_hip_tensor_get(t.ptr(), (char *)buf + off, 0, n);
            off += n;
        }
        ggml_hip_bind_thread_device(home);
        return off;
    }
    ggml::Tensor &t = which == 0 ? s->s->memory_k : s->s->memory_v;
    const size_t n = t.nbytes();
    if (!buf) return n;
    if (nbytes > n) nbytes = n;
    HomeDevice hd;
    hd.go(s->device);
    if (set)
        ggml_hip_tensor_set(t.ptr(), buf, 0, nbytes);
    else
        ggml_hip_tensor_get(t.ptr(), buf, 0, nbytes);
    return nbytes;
}

// InferenceSession::get_snapshot / from_snapshot (inference_session.rs:590-646): npast, config, tokens, last_logits,
// memory_k, memory_v.  The reference serialises the struct with serde/bincode; without a Rust toolchain the byte
// format here is our own (little-endian header + the four payloads), the CONTENT is the reference's.  The K/V memory
// lives on the device, so both directions go through the backend (ggml_hip_tensor_get / _set).
namespace {
struct SnapHeader {
    char magic[8];  // "LLMSNAP1"
    uint64_t npast, n_tokens, n_logits, k_bytes, v_bytes;
    int32_t memory_k_type, memory_v_type, n_batch, n_threads;
};
}  // namespace
size_t llm_session_snapshot(llm_session *s, void *buf, size_t cap) {
    llm::InferenceSession &ss = *s->s;
    SnapHeader h;
    memcpy(h.magic, "LLMSNAP1", 8);
    h.npast = ss.n_past;
    h.n_tokens = ss.tokens.size();
    h.n_logits = ss.last_logits.size();
    // a session split over device slots holds one K/V cache per stage: concatenated in layer order they ARE the unsplit
    // layout (llm_session_kv), so a snapshot does not depend on how the model was split when it was taken
    h.k_bytes = llm_session_kv(s, 0, 0, nullptr, 0);
    h.v_bytes = llm_session_kv(s, 1, 0, nullptr, 0);
    h.memory_k_type = (int32_t)ss.config.memory_k_type;
    h.memory_v_type = (int32_t)ss.config.memory_v_type;
    h.n_batch = (int32_t)ss.config.n_batch;
    h.n_threads = (int32_t)ss.config.n_threads;
    const size_t need = sizeof(h) + h.n_tokens * 4 + h.n_logits * 4 + h.k_bytes + h.v_bytes;
    if (!buf || cap < need) return need;
    char *p = (char *)buf;
    memcpy(p, &h, sizeof(h));
    p += sizeof(h);
    memcpy(p, ss.tokens.data(), h.n_tokens * 4);
    p += h.n_tokens * 4;
    memcpy(p, ss.last_logits.data(), h.n_logits * 4);
    p += h.n_logits * 4;
    llm_session_kv(s, 0, 0, p, h.k_bytes);
    p += h.k_bytes;
    llm_session_kv(s, 1, 0, p, h.v_bytes);
    return need;
}
// NULL = SnapshotError (bad header, or MemorySizeMismatch: the model's session has other K/V sizes than the snapshot)
llm_session *llm_session_from_snapshot(llm_model *m, const void *buf, size_t n) {
    SnapHeader h;
    if (!buf || n < sizeof(h)) return nullptr;
    memcpy(&h, buf, sizeof(h));
    if (memcmp(h.magic, "LLMSNAP1", 8) != 0) return nullptr;
    if (n != sizeof(h) + h.n_tokens * 4 + h.n_logits * 4 + h.k_bytes + h.v_bytes) return nullptr;
    llm_session_config cfg{h.memory_k_type, h.memory_v_type, h.n_batch, h.n_threads};
    llm_session *s = llm_start_session(m, &cfg);
    llm::InferenceSession &ss = *s->s;
    if (llm_session_kv(s, 0, 0, nullptr, 0) != h.k_bytes || llm_session_kv(s, 1, 0, nullptr, 0) != h.v_bytes ||
        ss.last_logits.size() != h.n_logits) {
        llm_session_free(s);
        return nullptr;  // SnapshotError::MemorySizeMismatch
    }
    const char *p = (const char *)buf + sizeof(h);
    ss.tokens.assign((const llm::TokenId *)p, (const llm::TokenId *)p + h.n_tokens);
    p += h.n_tokens * 4;
    memcpy(ss.last_logits.data(), p, h.n_logits * 4);
    p += h.n_logits * 4;
    llm_session_kv(s, 0, 1, (void *)p, h.k_bytes);
    p += h.k_bytes;
    llm_session_kv(s, 1, 1, (void *)p, h.v_bytes);
    ss.n_past = h.npast;
    for (auto *st : s->stage_sessions) st->n_past = h.npast;  // every stage of a split session is at the same position
    return s;
}

// test hook: device contents of a node of the last evaluated graph, by index (>= 0) or by the k-th node
// carrying `name` (index < 0).  Returns the number of bytes the node holds, 0 if not found.
size_t llm_session_read_node(const llm_session *s, int index, const char *name, int occurrence, void *dst,
                             size_t max_bytes) {
    ggml_cgraph *g = s->s->last_graph;
    if (!g) return 0;
    ggml_tensor *t = nullptr;
    if (index >= 0) {
        if (index < g->n_nodes) t = g->nodes[index];
    } else {
        int seen = 0;
        for (int i = 0; i < g->n_nodes && !t; i++)
            if (strcmp(g->nodes[i]->name, name) == 0 && seen++ == occurrence) t = g->nodes[i];
    }
    if (!t || !ggml_is_contiguous(t)) return 0;
    const size_t n = ggml_nbytes(t);
    HomeDevice hd;
    hd.go(s->stage_sessions.empty() ? s->device : s->devices.back());
    if (dst && n <= max_bytes) ggml_hip_tensor_get(t, dst, 0, n);
    return n;
}

// test hook: the HOST bytes of node `n_nodes - 1 - from_end` of the last evaluated graph (what an unchanged caller that reads
// tensor->data sees; from_end = 1 is the embedding_result node of the LLaMA graph, models/llama/src/lib.rs:343-347)
size_t llm_session_read_node_host(const llm_session *s, int from_end, void *dst, size_t max_bytes) {
    ggml_cgraph *g = s->s->last_graph;
    if (!g || from_end < 0 || from_end >= g->n_nodes) return 0;
    ggml_tensor *t = g->nodes[g->n_nodes - 1 - from_end];
    const size_t n = ggml_nbytes(t);
    if (dst && n <= max_bytes && t->data) memcpy(dst, t->data, n);
    return n;
}

// The k best logits of the last evaluated token without reading n_vocab floats back (ggml_hip_topk on the logits node of
// the last graph = its last node, last row): what the sampler chain's top-k stage needs (samplers.rs:289-306).
// `extra_ids`: tokens whose raw logits the caller wants too (repetition-penalty window, bias list).  0 on success.
int llm_session_topk(const llm_session *s, int k, const int32_t *extra_ids, int n_extra, float *out_vals, int32_t *out_ids) {
    ggml_cgraph *g = s->s->last_graph;
    if (!g || g->n_nodes < 1) return -1;
    const ggml_tensor *t = g->nodes[g->n_nodes - 1];
    if (t->type != GGML_TYPE_F32 || (size_t)t->ne[0] != s->s->last_logits.size()) return -1;
    HomeDevice hd;
    hd.go(s->stage_sessions.empty() ? s->device : s->devices.back());
    return ggml_hip_topk(t, t->ne[1] - 1, k, extra_ids, n_extra, out_vals, out_ids);
}

// Synthetic GGML blocks for full-size benchmarks (no checkpoints are obtainable offline): uniform random
// quants, f16 scale d = d_scale*(0.5+u), and for the *_1 types a min that centres the block.  Fills
// `nblocks` blocks of `type` at dst; deterministic in (seed, block index); multi-threaded.
void llm_synth_blocks(int type, void *dst, int64_t nblocks, uint64_t seed, float d_scale) {
    const size_t bs = ggml_type_size((ggml_type)type);
    const bool has_m = type == GGML_TYPE_Q4_1 || type == GGML_TYPE_Q5_1;
    const float lv = type == GGML_TYPE_Q4_1 ? 7.5f : 15.5f;
    unsigned nt = std::thread::hardware_concurrency();
    nt = nt == 0 ? 1 : (nt > 32 ? 32 : nt);
    auto work = [&](int64_t b0, int64_t b1) {
        uint8_t *p = (uint8_t *)dst + (size_t)b0 * bs;
        for (int64_t b = b0; b < b1; b++, p += bs) {
            uint64_t x = seed * 0x9E3779B97F4A7C15ull + (uint64_t)b * 0xD1B54A32D192ED03ull;
            auto next = [&x]() {
                x += 0x9E3779B97F4A7C15ull;
                uint64_t z = x;
                z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
                z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
                return z ^ (z >> 31);
            };
            for (size_t o = 0; o < bs; o += 8) {
                const uint64_t r = next();
                memcpy(p + o, &r, bs - o < 8 ? bs - o : 8);
            }
            const float u = (float)(next() >> 40) * (1.0f / 16777216.0f);
            const float d = d_scale * (0.5f + u);
            const ggml_fp16_t dh = ggml_fp32_to_fp16(d);
            if (type == GGML_TYPE_Q6_K) {  // block_q6_K: ql[128] qh[64] scales[16] d — int8 scales x 6-bit quants, |sc*q| ~ 1e3
                const ggml_fp16_t d6 = ggml_fp32_to_fp16(d * (1.0f / 1024.0f));
                memcpy(p + 208, &d6, 2);
                continue;
            }
            if (type == GGML_TYPE_Q5_K) {  // block_q5_K: d dmin scales[12] qh[32] qs[128] — x = d*sc*q - dmin*m, q five-bit, sc, m six-bit
                const ggml_fp16_t d5 = ggml_fp32_to_fp16(d * (1.0f / 64.0f)), m5 = ggml_fp32_to_fp16(d * (15.5f / 64.0f));
                memcpy(p, &d5, 2);
                memcpy(p + 2, &m5, 2);
                continue;
            }
            if (type == GGML_TYPE_Q3_K) {  // block_q3_K: hmask[32] qs[64] scales[12] d — x = d*(sc-32)*(q-4), sc six-bit, q three-bit
                const ggml_fp16_t d3 = ggml_fp32_to_fp16(d * (1.0f / 16.0f));
                memcpy(p + 108, &d3, 2);
                continue;
            }
            if (type == GGML_TYPE_Q2_K) {  // block_q2_K: scales[16] qs[64] d dmin — x = d*sc*q - dmin*m, sc, m four-bit, q two-bit
                const ggml_fp16_t d2 = ggml_fp32_to_fp16(d * (1.0f / 4.0f)), m2 = ggml_fp32_to_fp16(d * (1.5f / 4.0f));
                memcpy(p + 80, &d2, 2);
                memcpy(p + 82, &m2, 2);
                continue;
            }
            if (type == GGML_TYPE_Q4_K) {  // block_q4_K: d dmin scales[12] qs[128] — x = d*sc*q - dmin*m, sc, m six-bit
                const ggml_fp16_t d4 = ggml_fp32_to_fp16(d * (1.0f / 32.0f)), m4 = ggml_fp32_to_fp16(d * (7.5f / 32.0f));
                memcpy(p, &d4, 2);
                memcpy(p + 2, &m4, 2);
                continue;
            }
            memcpy(p, &dh, 2);
            if (has_m) {
                const ggml_fp16_t mh = ggml_fp32_to_fp16(-lv * ggml_fp16_to_fp32(dh));
                memcpy(p + 2, &mh, 2);
            }
        }
    };
    std::vector<std::thread> th;
    const int64_t per = (nblocks + nt - 1) / nt;
    for (unsigned t = 0; t < nt; t++) {
        const int64_t b0 = (int64_t)t * per, b1 = std::min<int64_t>(nblocks, b0 + per);
        if (b0 < b1) th.emplace_back(work, b0, b1);
    }
    for (auto &t : th) t.join();
}

// BASELINE.md section 4's synthetic weights at full size: rows of N(0, std^2) f32 quantized by ggml_quantize_q* (the
// product's host quantizer = what the reference's quantize tool calls), written as raw GGML blocks.  The gaussians come
// from a counter-based generator (splitmix64 + Box-Muller, deterministic in (seed, row)) instead of numpy's stream:
// 6.6e9 numpy draws take minutes of single-thread time before a bench could start.  Rows are cut over the host's
// threads.  `type` = a block type of 32 weights; ne0 = row length (multiple of 32); ne1 = rows.
void llm_synth_gaussian(int type, void *dst, int64_t ne0, int64_t ne1, uint64_t seed, float std) {
    const size_t row_bytes = (size_t)(ne0 / 32) * ggml_type_size((ggml_type)type);
    unsigned nt = std::thread::hardware_concurrency();
    nt = nt == 0 ? 1 : (nt > 64 ? 64 : nt);
    auto work = [&](int64_t r0, int64_t r1) {
        std::vector<float> row((size_t)ne0);
        int64_t hist[16] = {0};
        for (int64_t r = r0; r < r1; r++) {
            uint64_t x = seed * 0x9E3779B97F4A7C15ull + (uint64_t)r * 0xD1B54A32D192ED03ull;
            auto next = [&x]() {
                x += 0x9E3779B97F4A7C15ull;
                uint64_t z = x;
                z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
                z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
                return z ^ (z >> 31);
            };
            for (int64_t i = 0; i < ne0; i += 2) {
                const uint64_t u = next();
                const float u1 = ((float)(u >> 40) + 1.0f) * (1.0f / 16777216.0f);  // (0, 1]
                const float u2 = (float)((u >> 16) & 0xFFFFFF) * (1.0f / 16777216.0f);
                const float m = std * sqrtf(-2.0f * logf(u1));
                row[i] = m * cosf(6.2831853071795865f * u2);
                if (i + 1 < ne0) row[i + 1] = m * sinf(6.2831853071795865f * u2);
            }
            ggml_quantize_chunk((ggml_type)type, row.data(), (uint8_t *)dst + (size_t)r * row_bytes, 0, (int)ne0, hist);
        }
    };
    std::vector<std::thread> th;
    const int64_t per = (ne1 + nt - 1) / nt;
    for (unsigned t = 0; t < nt; t++) {
        const int64_t r0 = (int64_t)t * per, r1 = std::min<int64_t>(ne1, r0 + per);
        if (r0 < r1) th.emplace_back(work, r0, r1);
    }
    for (auto &t : th) t.join();
}

}  // extern "C"
