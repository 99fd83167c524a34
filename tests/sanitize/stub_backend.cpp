// stub_backend.cpp — a HOST stand-in for hip_backend.hip, used ONLY by the sanitizer job (tests/test_sanitize.py): the two
// host translation units of the product (llm_amd/csrc/ggml_core.cpp = the ggml C API: arenas, tensor builders, views, graph
// build / plan, quantizers; llm_amd/csrc/host/llm_host.cpp = the mirror of InferenceSession + models/llama, the GGML/GGMF/GGJT
// reader, snapshots, the greedy sampler) are compiled with -fsanitize=address,undefined and linked against THIS file instead
// of the device backend, so that every pointer computation, arena offset, view stride and container parse they do runs under
// ASan + UBSan on a machine without a GPU (the reference relies on Rust's checks there; C++ gets none).
// Nothing is computed: a "device tensor" is the tensor's own host memory, ggml_graph_compute fills the CPU-backend outputs
// (the logits) with a deterministic pattern.  Never linked into libggml_hip.so.
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <map>
#include <mutex>
#include <vector>

#include "ggml_hip.h"
#include "internal.h"

namespace {
std::atomic<int> g_main_device{0};
std::atomic<float> g_split{0.0f};
// the driver runs two sessions on two threads: the real backend serialises these behind its slot / arena locks, the stand-in
// needs its own (an unguarded std::map here made the sanitizer job fail about once in fifty runs)
std::mutex g_mu;
std::map<void *, size_t> g_arenas;
std::atomic<uint64_t> g_graphs{0};
thread_local ggml_cgraph *g_pending = nullptr;  // begin / end are called by the same thread

void fake_compute(ggml_cgraph *gr) {
    const uint64_t graphs = ++g_graphs;
    for (int i = 0; i < gr->n_nodes; i++) {
        ggml_tensor *t = gr->nodes[i];
        // touch every operand header the executor would read: a dangling src pointer trips ASan here
        for (int s = 0; s < GGML_MAX_SRC; s++)
            if (t->src[s]) (void)*(volatile int64_t *)&t->src[s]->ne[0];
        if (t->backend != GGML_BACKEND_CPU || !t->data || t->type != GGML_TYPE_F32) continue;
        if (t->op == GGML_OP_VIEW || t->op == GGML_OP_RESHAPE || t->op == GGML_OP_PERMUTE || t->op == GGML_OP_TRANSPOSE) continue;
        float *d = (float *)t->data;
        const int64_t n = ggml_nelements(t);
        for (int64_t k = 0; k < n; k++) {  // deterministic, position-dependent "logits": the argmax moves from graph to graph
            uint64_t x = (uint64_t)k * 0x9E3779B97F4A7C15ull + graphs * 0xD1B54A32D192ED03ull;
            x ^= x >> 29;
            d[k] = (float)(int32_t)(x & 0xFFFF) / 65536.0f;
        }
    }
}
}  // namespace

extern "C" {
void ggml_hip_internal_register_arena(void *host_base, size_t size, int) {
    std::lock_guard<std::mutex> lk(g_mu);
    g_arenas[host_base] = size;
}
void ggml_hip_internal_unregister_arena(void *host_base) {
    std::lock_guard<std::mutex> lk(g_mu);
    g_arenas.erase(host_base);
}
void ggml_hip_internal_graph_compute(struct ggml_cgraph *cgraph) { fake_compute(cgraph); }

void ggml_init_hipblas(void) {}
int ggml_hip_device_count(void) { return 1; }
int ggml_hip_slot_physical_device(int slot) { return slot >= 0 && slot < 4 ? 0 : -1; }
int ggml_hip_thread_session_slot(void) { return -1; }
void ggml_hip_set_option(const char *, int) {}
int ggml_hip_share_stream(int, int) { return 0; }
void ggml_hip_set_main_device(int d) { g_main_device = d; }
int ggml_hip_get_main_device(void) { return g_main_device; }
void ggml_hip_bind_thread_device(int) {}
int ggml_hip_thread_pinned_device(void) { return -1; }
void ggml_hip_unbind_thread_device(void) {}
void ggml_hip_set_tensor_split(const float *s) { g_split = s ? s[0] : 0.0f; }
int ggml_hip_get_layer_split(float *out, int cap) {
    if (out && cap > 0) out[0] = g_split;
    return 0;  // no layer split
}
void ggml_hip_set_scratch_size(size_t) {}
void ggml_hip_free_scratch(void) {}
void ggml_hip_transform_tensor(void *data, struct ggml_tensor *t) {
    // the real hook reads ggml_nbytes(t) bytes at `data`: do the same read so that a short mapping or a bad size trips ASan
    const size_t n = ggml_nbytes(t);
    volatile unsigned char acc = 0;
    for (size_t i = 0; i < n; i += 4096) acc ^= ((const unsigned char *)data)[i];
    if (n) acc ^= ((const unsigned char *)data)[n - 1];
    (void)acc;
    t->backend = GGML_BACKEND_GPU;
}
void ggml_hip_free_data(struct ggml_tensor *t) { t->backend = GGML_BACKEND_CPU; }
void ggml_hip_assign_buffers(struct ggml_tensor *t) { t->backend = GGML_BACKEND_GPU; }
void ggml_hip_assign_buffers_no_scratch(struct ggml_tensor *t) { t->backend = GGML_BACKEND_GPU; }
void *ggml_hip_tensor_device_ptr(const struct ggml_tensor *t) { return t->data; }
void ggml_hip_tensor_get(const struct ggml_tensor *t, void *dst, size_t off, size_t n) {
    if (off + n > ggml_nbytes(t)) { fprintf(stderr, "stub: tensor_get out of range\n"); abort(); }
    memcpy(dst, (const char *)t->data + off, n);
}
void ggml_hip_tensor_set(struct ggml_tensor *t, const void *src, size_t off, size_t n) {
    if (off + n > ggml_nbytes(t)) { fprintf(stderr, "stub: tensor_set out of range\n"); abort(); }
    memcpy((char *)t->data + off, src, n);
}
void ggml_hip_copy_between_devices(int, void *dst, int, const void *src, size_t n) { memcpy(dst, src, n); }
int ggml_hip_graph_compute_begin(struct ggml_cgraph *cgraph) {
    if (g_pending) fake_compute(g_pending);  // (a chunk of feed_prompt that was only enqueued: completed by the next begin)
    g_pending = cgraph;
    return 1;
}
int ggml_hip_graph_prepare(struct ggml_cgraph *) { return 0; }
void ggml_hip_graph_compute_end(void) {
    if (g_pending) fake_compute(g_pending);
    g_pending = nullptr;
}
int ggml_hip_decode_greedy_chain(struct ggml_cgraph *, int, int32_t *, float *) { return -1; }  // caller decodes token by token
int ggml_hip_topk(const struct ggml_tensor *t, int64_t row, int k, const int32_t *extra_ids, int n_extra, float *out_vals,
                  int32_t *out_ids) {
    if (!t || t->type != GGML_TYPE_F32 || row < 0 || row >= t->ne[1] || k < 1 || k > t->ne[0]) return -1;
    const float *r = (const float *)((const char *)t->data + row * t->nb[1]);
    std::vector<char> taken((size_t)t->ne[0], 0);
    for (int i = 0; i < k; i++) {
        int64_t best = -1;
        for (int64_t j = 0; j < t->ne[0]; j++)
            if (!taken[j] && (best < 0 || r[j] > r[best])) best = j;
        taken[best] = 1;
        out_vals[i] = r[best];
        out_ids[i] = (int32_t)best;
    }
    for (int i = 0; i < n_extra; i++) out_vals[k + i] = r[extra_ids[i]];
    return 0;
}
}
