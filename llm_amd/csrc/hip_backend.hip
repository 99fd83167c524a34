// hip_backend.hip — the MI355X (gfx950) backend of libggml_hip.so.
//
// Fills the plugin slot that crates/ggml/sys/src/cuda.rs:6-77 defines (19 accelerator hooks) and
// executes whole compute graphs for ggml_graph_compute (crates/ggml/src/lib.rs:374-376).  Design,
// MI355X-first rather than a translation of ggml-cuda:
//   * Device residency by address mirroring.  Every host arena (ggml context buffer, scratch buffer)
//     gets a lazily created device shadow of the same size; a tensor's device address is
//     shadow + (tensor->data - arena_base).  288 GB of HBM makes the 2-3 GiB of shadows a non-issue and
//     removes per-node buffer assignment, scratch pools and view bookkeeping from the hot path.
//   * Weights / KV cache get private persistent buffers (ggml_hip_transform_tensor /
//     ggml_hip_assign_buffers_no_scratch); quantized 2-D weights are re-laid-out once into the SoA
//     layout of kernels/common.h so the mat-vec streams them with aligned 16-byte loads.
//   * One in-order HIP stream; the graph's node order is the dependency order (as in ggml).
//   * No CPU compute: an unsupported op aborts with a message.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "ggml_hip.h"
#include "internal.h"
#include "kernels/common.h"
#include "kernels/mmvq.h"
#include "kernels/mmq.h"
#include "kernels/mmq_plain.h"
#include "kernels/mmq_dmap8.h"
#include "kernels/mmq_w16.h"
#include "kernels/mmq_w16_256.h"
#include "kernels/mmq_i8.h"
#include "kernels/kquant.h"
#include "kernels/kquant2.h"
#include "kernels/quantize.h"
#include "kernels/topk.h"
#include "kernels/gemm_f16.h"
#include "kernels/ops.h"
#include "kernels/decode.h"
#include "kernels/decode_big.h"
#include "kernels/decode_fused.h"
#include "kernels/decode_big8.h"
#include "kernels/mmq_cols.h"
#include "kernels/decode_attn_split.h"
#include "kernels/kquant_plan.h"
#include "kernels/kquant_big.h"
#include "kernels/prompt.h"
#include "kernels/prompt_attn.h"

// Last words of the library: stderr, and (GGML_HIP_FATAL_LOG=path) a file — a test runner that captures file descriptor 2
// swallows the message of an abort together with the process.
static void fatal_note(const char *fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    fprintf(stderr, "libggml_hip: %s\n", buf);
    if (const char *path = getenv("GGML_HIP_FATAL_LOG"))
        if (FILE *f = fopen(path, "a")) {
            fprintf(f, "libggml_hip: %s\n", buf);
            fclose(f);
        }
}

#define HIP_CHECK(expr)                                                                              \
    do {                                                                                             \
        hipError_t err__ = (expr);                                                                   \
        if (err__ != hipSuccess) {                                                                   \
            fatal_note("HIP error %d (%s) at %s:%d: %s", (int)err__, hipGetErrorString(err__), __FILE__,  \
                       __LINE__, #expr);                                                             \
            abort();                                                                                 \
        }                                                                                            \
    } while (0)

#define BK_ASSERT(x)                                                                                 \
    do {                                                                                             \
        if (!(x)) {                                                                                  \
            fatal_note("assertion failed at %s:%d: %s", __FILE__, __LINE__, #x);                     \
            abort();                                                                                 \
        }                                                                                            \
    } while (0)

extern "C" void ggml_hip_internal_set_option_here(const char *key, int value);
namespace {
#include "backend_state.inc"
#include "backend_memory.inc"
#include "backend_ops.inc"
#include "backend_executor.inc"
}  // namespace

// ===================================================================================================
// exported: internal seam
// ===================================================================================================
// Host arenas are known to every slot; registration only appends to the process-wide event log (backend_state.inc "host
// arenas"), each slot applies it under its own lock at its next entry point.
extern "C" void ggml_hip_internal_register_arena(void *host_base, size_t size, int is_scratch) {
    arena_event(true, host_base, size, is_scratch);
}
extern "C" void ggml_hip_internal_unregister_arena(void *host_base) { arena_event(false, host_base, 0, 0); }
namespace {
void evict_all_auto_tensors() {
    for (auto it = g.auto_tensors.begin(); it != g.auto_tensors.end();) {
        DevTensor *e = it->second;
        it = g.auto_tensors.erase(it);
        destroy_record(e);
    }
}
void register_arena_here(void *host_base, size_t size, int is_scratch) {
    const uintptr_t b = (uintptr_t)host_base;
    if (size == 0) return;
    auto it = g.arenas.find(b);
    if (it != g.arenas.end() && it->second.size == size) {
        if (!it->second.live && it->second.dev) g.dead_shadow_bytes -= it->second.size;
        if (!it->second.live) evict_overlapping(g.auto_tensors, b, size);
        it->second.live = true;  // same buffer re-initialised (ctx0.recreate()): keep the device shadow
        return;
    }
    // drop arenas that overlap the new range (the host memory was recycled)
    for (auto jt = g.arenas.begin(); jt != g.arenas.end();) {
        Arena &a = jt->second;
        if (a.base < b + size && b < a.base + a.size) {
            if (a.dev) {
                if (g.stream) HIP_CHECK(hipStreamSynchronize(g.stream));
                HIP_CHECK(hipFree(a.dev));
                if (!a.live) g.dead_shadow_bytes -= a.size;
            }
            jt = g.arenas.erase(jt);
        } else {
            ++jt;
        }
    }
    evict_overlapping(g.auto_tensors, b, size);
    Arena a;
    a.base = b;
    a.size = size;
    a.is_scratch = is_scratch != 0;
    g.arenas[b] = a;
}

void unregister_arena_here(void *host_base) {
    const uintptr_t b = (uintptr_t)host_base;
    auto it = g.arenas.find(b);
    if (it == g.arenas.end()) return;
    Arena &a = it->second;
    // auto-uploaded persistent tensors whose host bytes lived in this arena die with it
    evict_overlapping(g.auto_tensors, a.base, a.size);
    // ... and so do those whose tensor HEADER lived here (e.g. mmap'd weights named by a model context): once the
    // context is freed nothing can refer to them any more, and the host bytes may be recycled with new content
    for (auto jt = g.auto_tensors.begin(); jt != g.auto_tensors.end();) {
        DevTensor *e = jt->second;
        if (e->owner_hdr >= a.base && e->owner_hdr < a.base + a.size) {
            jt = g.auto_tensors.erase(jt);
            destroy_record(e);
        } else {
            ++jt;
        }
    }
    a.live = false;
    if (a.dev) {
        g.dead_shadow_bytes += a.size;
        if (g.dead_shadow_bytes > ((size_t)8 << 30)) {  // bound the memory parked in dead shadows
            if (g.stream) HIP_CHECK(hipStreamSynchronize(g.stream));
            for (auto jt = g.arenas.begin(); jt != g.arenas.end();) {
                if (!jt->second.live) {
                    if (jt->second.dev) HIP_CHECK(hipFree(jt->second.dev));
                    jt = g.arenas.erase(jt);
                } else {
                    ++jt;
                }
            }
            g.dead_shadow_bytes = 0;
        }
    } else {
        g.arenas.erase(it);
    }
}

}  // namespace

extern "C" void ggml_hip_internal_graph_compute(struct ggml_cgraph *cgraph) {
    SlotLock lk;
    const uint64_t t0 = now_ns();
    execute_graph(cgraph);
    g.ns_compute += now_ns() - t0;
}
// Split form of ggml_graph_compute for callers that have host work to overlap with the device (the session
// mirror builds the next token's graph meanwhile): begin() enqueues the graph and returns 1 if it is still running
// (fused decode plan), 0 if it was executed synchronously (any other graph); end() waits and finishes the
// read-back of the host-visible results.  Nothing else may read results before end().
extern "C" int ggml_hip_graph_compute_begin(struct ggml_cgraph *cgraph) {
    SlotLock lk;
    const uint64_t t0 = now_ns();
    ensure_init();
    if (g.pending_wait && g.pending_light && stg.pending.empty() && !g.results_ev_armed) {
        // a prompt chunk / batch that was only enqueued (feed_prompt's chunks but the last): nothing of it is host-visible, and
        // what this call enqueues runs behind it on the same stream — no wait
        g.pending_wait = false;
        g.ferr_plan = nullptr;
    } else {
        finish_pending();
    }
    ws_reset();
    g_qact.valid = false;
    g_xf16.valid = false;
    g_xi8.valid = false;
    g_xk.valid = false;
    int async = 0;
    if (try_decode_plan(cgraph, true)) {
        async = 1;
    } else {
        execute_graph(cgraph);
    }
    g.ns_compute += now_ns() - t0;
    return async;
}
extern "C" int ggml_hip_graph_prepare(struct ggml_cgraph *cgraph) {
    SlotLock lk;
    ensure_init();
    return prepare_decode_plan(cgraph) ? 1 : 0;  // host work only: nothing is enqueued, nothing is waited for
}
extern "C" void ggml_hip_graph_compute_end(void) {
    SlotLock lk;
    const uint64_t t0 = now_ns();
    finish_pending();
    g.ns_compute += now_ns() - t0;
}

// ===================================================================================================
// exported: the 19 accelerator hooks (crates/ggml/sys/src/cuda.rs:6-77)
// ===================================================================================================
extern "C" {

void ggml_init_hipblas(void) {
    SlotLock lk;
    ensure_init();
}
void ggml_hip_set_main_device(int main_device) {
    // crates/ggml/sys/src/cuda.rs:62 (accelerator/mod.rs:72): the slot every following call acts on — the process default
    // (threads that never chose a slot follow it) and the calling thread's slot
    if (main_device < 0 || main_device >= std::max(1, slot_count()))
        die("ggml_hip_set_main_device(%d): %d device slot(s) available", main_device, slot_count());
    g_default_slot.store(main_device, std::memory_order_relaxed);
    // a thread whose sessions were assigned a sibling slot of this GPU (GGML_HIP_SESSION_SLOTS) keeps it: the reference repeats
    // set_main_device(0) in every InferenceSession::new
    if (tl_auto_slot >= 0 && slot_physical_device(tl_auto_slot) == slot_physical_device(main_device)) {
        ggml_hip_bind_thread_device(tl_auto_slot);
        return;
    }
    ggml_hip_bind_thread_device(main_device);
}
// see backend_state.inc (GGML_HIP_SESSION_SLOTS); called in front of the first K/V memory a thread creates
static void auto_session_slot() {
    const int n = std::min(session_slots_env(), slot_count());
    if (n <= 1 || tl_auto_slot >= 0) return;
    if (!tl_pinned) g_cur = &g_backends[g_default_slot.load(std::memory_order_relaxed)];
    const int cur = (int)(g_cur - g_backends), phys = slot_physical_device(cur);
    int best = cur;
    {
        std::lock_guard<std::recursive_mutex> lk(g_mu);
        for (int i = 0; i < n; i++)
            if (slot_physical_device(i) == phys && g_session_threads[i] < g_session_threads[best]) best = i;
        g_session_threads[best]++;
        tl_auto_slot = best;
        (void)&tl_auto_release;  // (instantiates the thread's release object)
    }
    if (best != cur) ggml_hip_bind_thread_device(best);
    else tl_pinned = true, g_cur = &g_backends[cur];
}
int ggml_hip_thread_session_slot(void) { return tl_auto_slot; }
void ggml_hip_bind_thread_device(int device) {
    // the calling thread's slot only: sessions of one process driven from several threads bind their own model's slot
    if (device < 0 || device >= std::max(1, slot_count()))
        die("ggml_hip_bind_thread_device(%d): %d device slot(s) available", device, slot_count());
    tl_pinned = true;
    g_cur = &g_backends[device];
    SlotLock lk(g_cur);
    g.slot = device;
    if (g.inited) bind_device();
}
int ggml_hip_thread_pinned_device(void) { return tl_pinned ? (int)(g_cur - g_backends) : -1; }
void ggml_hip_unbind_thread_device(void) {
    tl_pinned = false;
    g_cur = &g_backends[g_default_slot.load(std::memory_order_relaxed)];
}
void ggml_hip_set_tensor_split(const float *tensor_split) {
    // crates/ggml/sys/src/cuda.rs:11.  The reference's only caller passes the address of ONE stack float
    // (crates/ggml/src/accelerator/mod.rs:74-75: `let split = 1.0f32; ggml_cuda_set_tensor_split(&split as *const f32)`,
    // LLAMA_MAX_DEVICES = 1), so exactly one float is read here, whatever the number of visible devices: upstream's hook
    // reads one per device, and on an 8-GPU box that is seven floats of the caller's stack.  A split over several slots is
    // asked for through ggml_hip_set_layer_split (explicit length) or GGML_HIP_LAYER_SPLIT.
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    for (int i = 0; i < GGML_HIP_MAX_BACKENDS; i++) g_tensor_split[i] = 0.0f;
    g_tensor_split[0] = tensor_split ? tensor_split[0] : 0.0f;
}
int ggml_hip_get_tensor_split(float *out, int cap) {
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    const int n = std::min(cap, GGML_HIP_MAX_BACKENDS);
    for (int i = 0; i < n; i++) out[i] = g_tensor_split[i];
    return std::min(n, 1);
}
void ggml_hip_set_layer_split(const float *fractions, int n) {
    // ggml's convention for a split (slot i takes fractions[i] / sum), applied to LAYERS: rows of one tensor are never
    // split (llm_split_layers, host/llm_host.cpp).  n <= 0 or NULL clears it; all-zero fractions mean equal shares.
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    g_layer_split_n = fractions && n > 0 ? std::min(n, GGML_HIP_MAX_BACKENDS) : 0;
    for (int i = 0; i < GGML_HIP_MAX_BACKENDS; i++) g_layer_split[i] = i < g_layer_split_n ? fractions[i] : 0.0f;
}
int ggml_hip_get_layer_split(float *out, int cap) {
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    const int n = std::min(cap, g_layer_split_n);
    for (int i = 0; i < n; i++) out[i] = g_layer_split[i];
    return n;
}
void ggml_hip_set_mul_mat_q(bool) {}  // quantized kernels are always used
void ggml_hip_set_scratch_size(size_t) {}  // activations live in arena shadows, there is no scratch pool
void ggml_hip_free_scratch(void) {
    SlotLock lk;
    if (!g.inited) return;
    HIP_CHECK(hipStreamSynchronize(g.stream));
    // Cached decode plans (and the plan a greedy chain may continue) hold device addresses inside these shadows
    // (logits / embedding mirrors): they go first.
    drop_all_plans();
    g.chain_plan = nullptr;
    g.chain_graph = nullptr;
    // Release the device shadows of dead arenas (freed contexts) and of every scratch buffer.  Scratch
    // registrations stay (the caller-owned Buffers may still be in use by another session); their
    // shadows hold only per-evaluation temporaries and are re-created lazily on next use.
    for (auto it = g.arenas.begin(); it != g.arenas.end();) {
        Arena &a = it->second;
        if (!a.live) {
            if (a.dev) HIP_CHECK(hipFree(a.dev));
            it = g.arenas.erase(it);
            continue;
        }
        if (a.is_scratch && a.dev) {
            HIP_CHECK(hipFree(a.dev));
            a.dev = nullptr;
        }
        ++it;
    }
    g.dead_shadow_bytes = 0;
}
void *ggml_hip_host_malloc(size_t size) {
    SlotLock lk;
    ensure_init();
    void *p = nullptr;
    if (hipHostMalloc(&p, size, hipHostMallocDefault) != hipSuccess) return nullptr;
    return p;
}
void ggml_hip_host_free(void *ptr) {
    if (ptr) HIP_CHECK(hipHostFree(ptr));
}
void ggml_hip_transform_tensor(void *data, struct ggml_tensor *tensor) {
    SlotLock lk;
    if (extra_of(tensor)) return;
    tensor->backend = GGML_BACKEND_GPU;
    tensor->extra = upload_tensor(data, tensor, false);
}

// ---- device quantizer (kernels/quantize.h) ----
static bool quantizable(int type) {
    return type == GGML_TYPE_Q4_0 || type == GGML_TYPE_Q4_1 || type == GGML_TYPE_Q5_0 || type == GGML_TYPE_Q5_1 || type == GGML_TYPE_Q8_0;
}
static void launch_quantize_blocks(const void *src_dev, bool f16_src, int type, int64_t nblocks, uint8_t *out_dev,
                                   unsigned long long *hist_dev) {
    Timed tm(GGML_HIP_KCLASS_OTHER, (double)nblocks * (f16_src ? 64 : 128) + (double)nblocks * ggml_type_size((ggml_type)type));
    if (f16_src)
        hipLaunchKernelGGL(k_quantize_blocks<true>, grid1(nblocks), dim3(256), 0, g.stream, src_dev, type, nblocks, out_dev, hist_dev);
    else
        hipLaunchKernelGGL(k_quantize_blocks<false>, grid1(nblocks), dim3(256), 0, g.stream, src_dev, type, nblocks, out_dev, hist_dev);
    HIP_CHECK(hipGetLastError());
}
size_t ggml_hip_quantize(enum ggml_type type, const float *src, void *dst, int64_t n, int64_t k, int64_t *hist) {
    SlotLock lk;
    ensure_init();
    if (!quantizable(type)) die("ggml_hip_quantize: %s has no device encoder", ggml_type_name(type));
    if (k % 32 != 0 || n % k != 0) die("ggml_hip_quantize: n = %lld must be rows of k = %lld, k %% 32 == 0", (long long)n, (long long)k);
    const size_t bs = ggml_type_size(type);
    const int64_t nblocks = n / 32;
    const int64_t piece = (int64_t)1 << 23;  // blocks per pass: 1 GiB of f32 in, <= 272 MiB out
    char *din = nullptr, *dout = nullptr;
    unsigned long long *dh = nullptr;
    const int64_t cap = std::min(nblocks, piece);
    dev_malloc((void **)&din, (size_t)cap * 128, "the quantizer input");
    dev_malloc((void **)&dout, (size_t)cap * bs, "the quantizer output");
    HIP_CHECK(hipMalloc((void **)&dh, 128));
    HIP_CHECK(hipMemsetAsync(dh, 0, 128, g.stream));
    for (int64_t b0 = 0; b0 < nblocks; b0 += piece) {
        const int64_t nb = std::min(piece, nblocks - b0);
        h2d_bulk(din, src + b0 * 32, (size_t)nb * 128);
        launch_quantize_blocks(din, false, (int)type, nb, (uint8_t *)dout, dh);
        d2h_queue((char *)dst + (size_t)b0 * bs, dout, (size_t)nb * bs);  // pinned staging, delivered by d2h_finish
        d2h_finish();
    }
    unsigned long long hh[16];
    HIP_CHECK(hipMemcpy(hh, dh, 128, hipMemcpyDeviceToHost));
    if (hist)
        for (int i = 0; i < 16; i++) hist[i] += (int64_t)hh[i];
    HIP_CHECK(hipFree(din));
    HIP_CHECK(hipFree(dout));
    HIP_CHECK(hipFree(dh));
    return (size_t)nblocks * bs;
}
int ggml_hip_quantize_resident(const struct ggml_tensor *src, struct ggml_tensor *dst, int64_t *hist) {
    SlotLock lk;
    ensure_init();
    if (!quantizable(dst->type) || (src->type != GGML_TYPE_F32 && src->type != GGML_TYPE_F16)) return -1;
    if (!ggml_is_contiguous(src) || src->ne[2] != 1 || src->ne[3] != 1 || src->ne[0] % 32 != 0 || src->ne[0] != dst->ne[0] ||
        src->ne[1] != dst->ne[1] || dst->ne[2] != 1 || dst->ne[3] != 1 || extra_of(dst) || dst->data == nullptr)
        return -1;
    if (!extra_of(src) && !find_tensor((uintptr_t)src->data)) return -1;  // the source must already be on the device
    finish_pending();
    const int64_t M = src->ne[1], nb = src->ne[0] / 32, nblocks = M * nb;
    const size_t bs = ggml_type_size(dst->type);
    char *raw = nullptr;
    unsigned long long *dh = nullptr;
    dev_malloc((void **)&raw, (size_t)nblocks * bs, "the resident quantizer output");
    HIP_CHECK(hipMalloc((void **)&dh, 128));
    HIP_CHECK(hipMemsetAsync(dh, 0, 128, g.stream));
    launch_quantize_blocks(dev_ptr(src), src->type == GGML_TYPE_F16, (int)dst->type, nblocks, (uint8_t *)raw, dh);
    DevTensor *e = new DevTensor();
    e->host = (uintptr_t)dst->data;
    e->nbytes = ggml_nbytes(dst);
    e->type = dst->type;
    for (int i = 0; i < 4; i++) e->ne[i] = dst->ne[i];
    const int qt = qt_of(dst->type);
    size_t off[5];
    const size_t total = qw_layout(qt, nblocks, off);
    dev_malloc((void **)&e->dev, total, "a weight tensor");
    e->dev_bytes = total;
    relayout_launch(raw, qt, M, nb, e->dev);
    unsigned long long hh[16];
    HIP_CHECK(hipMemcpyAsync(hh, dh, 128, hipMemcpyDeviceToHost, g.stream));
    HIP_CHECK(hipStreamSynchronize(g.stream));
    if (hist)
        for (int i = 0; i < 16; i++) hist[i] += (int64_t)hh[i];
    HIP_CHECK(hipFree(raw));
    HIP_CHECK(hipFree(dh));
    e->soa = true;
    e->qw = qw_at(e->dev, qt, M, nb);
    evict_overlapping(g.auto_tensors, e->host, std::max<size_t>(e->nbytes, 1));
    evict_overlapping(g.tensors, e->host, std::max<size_t>(e->nbytes, 1));
    g.tensors[e->host] = e;
    dst->backend = GGML_BACKEND_GPU;
    dst->extra = e;
    return 0;
}

// ---- device top-k prefilter (kernels/topk.h) ----
int ggml_hip_topk(const struct ggml_tensor *t, int64_t row, int k, const int32_t *extra_ids, int n_extra, float *out_vals,
                  int32_t *out_ids) {
    SlotLock lk;
    ensure_init();
    if (!t || t->type != GGML_TYPE_F32 || t->nb[0] != 4 || row < 0 || row >= t->ne[1] * t->ne[2] * t->ne[3] || t->ne[2] != 1 ||
        t->ne[3] != 1 || k < 1 || k > TOPK_MAX || k > t->ne[0] || n_extra < 0 || t->ne[0] > 0x7FFFFFFF || !out_vals || !out_ids ||
        (n_extra > 0 && !extra_ids) || t->data == nullptr)
        return -1;
    {   // the tensor must have a device image (a record or an arena shadow); dev_ptr would abort otherwise
        DevTensor *e = extra_of(t);
        if (!e) e = find_tensor((uintptr_t)t->data);
        if (e ? e->soa : find_arena((uintptr_t)t->data) == nullptr) return -1;
    }
    finish_pending();
    const float *x = (const float *)(dev_ptr(t) + row * (int64_t)t->nb[1]);
    char *buf = ws_alloc((size_t)(k + n_extra) * 8 + (size_t)n_extra * 4 + 64);
    float *dv = (float *)buf;
    int *di = (int *)(buf + (size_t)(k + n_extra) * 4);
    int *de = di + (k + n_extra);
    if (n_extra) h2d_small((char *)de, extra_ids, (size_t)n_extra * 4);  // through pinned staging, never from caller pages
    {
        Timed tm(GGML_HIP_KCLASS_OTHER, (double)t->ne[0] * 4 * 9);
        hipLaunchKernelGGL(k_topk, dim3(1), dim3(1024), 0, g.stream, x, (int)t->ne[0], k, (const int *)de, n_extra, dv, di);
        HIP_CHECK(hipGetLastError());
    }
    d2h_queue(out_vals, (const char *)dv, (size_t)(k + n_extra) * 4);
    d2h_queue(out_ids, (const char *)di, (size_t)(k + n_extra) * 4);
    d2h_finish();
    return 0;
}

void ggml_hip_free_data(struct ggml_tensor *tensor) {
    if (!tensor || !tensor->extra) return;
    DevTensor *e = (DevTensor *)tensor->extra;
    if (e->magic != 0x48495054) return;  // scratch-assigned node: nothing to free (as in the reference)
    // the OWNER slot's lock (the record, its plans and its accounting live there), whichever slot this thread is on
    const int owner = (e->slot >= 0 && e->slot < GGML_HIP_MAX_BACKENDS) ? e->slot : g_cur_slot();
    SlotLock lk(&g_backends[owner]);
    free_dev_tensor(e);
    tensor->extra = nullptr;
}
void ggml_hip_assign_buffers(struct ggml_tensor *tensor) {
    // Compute nodes need no per-node device buffer: their device address is the arena mirror of
    // tensor->data. Marking the backend keeps the results device-only (no D2H after compute).
    tensor->backend = GGML_BACKEND_GPU;
}
void ggml_hip_assign_buffers_force_inplace(struct ggml_tensor *tensor) { tensor->backend = GGML_BACKEND_GPU; }
void ggml_hip_assign_buffers_no_scratch(struct ggml_tensor *tensor) {
    if (tensor->op == GGML_OP_NONE && !tensor->extra) auto_session_slot();  // K/V memory of a new session (inference_session.rs:996-1021)
    SlotLock lk;
    tensor->backend = GGML_BACKEND_GPU;
    if (tensor->op != GGML_OP_NONE || extra_of(tensor)) return;
    // persistent, zero-initialised device tensor (the K/V memory: inference_session.rs:996-1021)
    tensor->extra = upload_tensor(tensor->data, tensor, true);
}
bool ggml_hip_can_mul_mat(const struct ggml_tensor *src0, const struct ggml_tensor *src1, struct ggml_tensor *dst) {
    // quantized operands: only the block formats with kernels behind them, in the shapes those kernels take (anything
    // else is answered "no" here instead of aborting inside the launch)
    bool q = src0->type == GGML_TYPE_F16 || src0->type == GGML_TYPE_F32;
    if (qt_of(src0->type) >= 0) q = src0->ne[0] % 32 == 0;
    if (kt_of(src0->type) >= 0) q = src0->ne[0] % 256 == 0;
    return q && src1->type == GGML_TYPE_F32 && dst->type == GGML_TYPE_F32;
}
size_t ggml_hip_mul_mat_get_wsize(const struct ggml_tensor *, const struct ggml_tensor *, struct ggml_tensor *) {
    return 0;
}
void ggml_hip_mul_mat(const struct ggml_tensor *src0, const struct ggml_tensor *src1, struct ggml_tensor *dst, void *,
                      size_t) {
    SlotLock lk;
    ensure_init();
    BK_ASSERT(dst->src[0] == src0 && dst->src[1] == src1);
    op_mul_mat(dst);
}
void ggml_hip_mul(const struct ggml_tensor *src0, const struct ggml_tensor *src1, struct ggml_tensor *dst) {
    SlotLock lk;
    ensure_init();
    BK_ASSERT(dst->src[0] == src0 && dst->src[1] == src1);
    op_bin(dst, BIN_MUL);
}
bool ggml_hip_compute_forward(struct ggml_compute_params *params, struct ggml_tensor *tensor) {
    // Per-node hook of the reference's CPU executor. This library executes whole graphs itself
    // (ggml_graph_compute), so the hook only has to answer for callers that drive nodes one by one.
    SlotLock lk;
    if (params && (params->ith != 0 || params->type != GGML_TASK_COMPUTE)) return true;
    ensure_init();
    ggml_cgraph *gr = (ggml_cgraph *)calloc(1, sizeof(ggml_cgraph));
    // single-node graph: sources must already be device-visible
    gr->nodes[0] = tensor;
    gr->n_nodes = 1;
    execute_graph(gr);
    free(gr);
    return true;
}

// cublas-named aliases (zero-change drop-in for the reference's `cublas` cfg arms)
void ggml_init_cublas(void) { ggml_init_hipblas(); }
void ggml_cuda_set_tensor_split(const float *s) { ggml_hip_set_tensor_split(s); }
void ggml_cuda_mul(const struct ggml_tensor *a, const struct ggml_tensor *b, struct ggml_tensor *d) { ggml_hip_mul(a, b, d); }
bool ggml_cuda_can_mul_mat(const struct ggml_tensor *a, const struct ggml_tensor *b, struct ggml_tensor *d) {
    return ggml_hip_can_mul_mat(a, b, d);
}
size_t ggml_cuda_mul_mat_get_wsize(const struct ggml_tensor *a, const struct ggml_tensor *b, struct ggml_tensor *d) {
    return ggml_hip_mul_mat_get_wsize(a, b, d);
}
void ggml_cuda_mul_mat(const struct ggml_tensor *a, const struct ggml_tensor *b, struct ggml_tensor *d, void *w, size_t s) {
    ggml_hip_mul_mat(a, b, d, w, s);
}
void *ggml_cuda_host_malloc(size_t size) { return ggml_hip_host_malloc(size); }
void ggml_cuda_host_free(void *ptr) { ggml_hip_host_free(ptr); }
void ggml_cuda_transform_tensor(void *data, struct ggml_tensor *t) { ggml_hip_transform_tensor(data, t); }
void ggml_cuda_free_data(struct ggml_tensor *t) { ggml_hip_free_data(t); }
void ggml_cuda_assign_buffers(struct ggml_tensor *t) { ggml_hip_assign_buffers(t); }
void ggml_cuda_assign_buffers_no_scratch(struct ggml_tensor *t) { ggml_hip_assign_buffers_no_scratch(t); }
void ggml_cuda_assign_buffers_force_inplace(struct ggml_tensor *t) { ggml_hip_assign_buffers_force_inplace(t); }
void ggml_cuda_set_main_device(int d) { ggml_hip_set_main_device(d); }
void ggml_cuda_set_mul_mat_q(bool q) { ggml_hip_set_mul_mat_q(q); }
void ggml_cuda_set_scratch_size(size_t s) { ggml_hip_set_scratch_size(s); }
void ggml_cuda_free_scratch(void) { ggml_hip_free_scratch(); }
bool ggml_cuda_compute_forward(struct ggml_compute_params *p, struct ggml_tensor *t) { return ggml_hip_compute_forward(p, t); }

// ===================================================================================================
// exported: extensions
// ===================================================================================================
int ggml_hip_device_count(void) { return slot_count(); }
int ggml_hip_slot_physical_device(int slot) {  // the GPU a slot drives (ensure_init's rule), without initialising it; -1: no such slot
    int n = 0;
    if (slot < 0 || slot >= slot_count() || hipGetDeviceCount(&n) != hipSuccess || n <= 0) return -1;
    int base = 0;
    if (const char *lr = getenv("GGML_HIP_DEVICE")) base = atoi(lr);
    return (base + slot) % n;
}
int ggml_hip_get_main_device(void) {  // the calling thread's slot
    if (!tl_pinned) g_cur = &g_backends[g_default_slot.load(std::memory_order_relaxed)];
    return (int)(g_cur - g_backends);
}
// The residual of a layer split crossing from one slot to another: dst on slot dst_device's stream waits for what
// src_device's stream has enqueued so far, then copies (peer copy over xGMI between two GPUs, a device copy when both slots
// sit on one GPU).  Asynchronous; ordered with both slots' later work on their own streams.
void ggml_hip_copy_between_devices(int dst_device, void *dst, int src_device, const void *src, size_t nbytes) {
    if (dst_device < 0 || src_device < 0 || dst_device >= GGML_HIP_MAX_BACKENDS || src_device >= GGML_HIP_MAX_BACKENDS)
        die("ggml_hip_copy_between_devices: bad slot");
    Backend &S = g_backends[src_device], &D = g_backends[dst_device];
    // both slots, lower index first (the only place two slot locks are held together)
    SlotLock lk_a(&g_backends[std::min(src_device, dst_device)]);
    SlotLock lk_b(&g_backends[std::max(src_device, dst_device)]);  // recursive: the same slot twice is fine
    if (!S.inited || !D.inited) die("ggml_hip_copy_between_devices: slot not initialised");
    Backend *keep = g_cur;
    g_cur = &S;
    bind_device();
    // one event per source slot, re-recorded per hop (a later record does not disturb a wait already enqueued on it)
    if (S.stream != D.stream) {  // (two slots that share a stream — ggml_hip_share_stream — are ordered by it)
        if (!S.xfer_ev) HIP_CHECK(hipEventCreateWithFlags(&S.xfer_ev, hipEventDisableTiming));
        HIP_CHECK(hipEventRecord(S.xfer_ev, S.stream));
    }
    g_cur = &D;
    bind_device();
    if (S.stream != D.stream) HIP_CHECK(hipStreamWaitEvent(D.stream, S.xfer_ev, 0));
    if (S.device == D.device)
        HIP_CHECK(hipMemcpyAsync(dst, src, nbytes, hipMemcpyDeviceToDevice, D.stream));
    else {
        // direct xGMI path: the destination device maps the source's memory (once per ordered pair); without peer access the
        // runtime stages the copy through host memory, which is still correct
        static bool peer_tried[GGML_HIP_MAX_BACKENDS][GGML_HIP_MAX_BACKENDS];
        if (!peer_tried[dst_device][src_device]) {
            peer_tried[dst_device][src_device] = true;
            int can = 0;
            if (hipDeviceCanAccessPeer(&can, D.device, S.device) == hipSuccess && can) {
                const hipError_t e = hipDeviceEnablePeerAccess(S.device, 0);  // current device = D.device (bind_device above)
                if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled)
                    fprintf(stderr, "ggml-hip: peer access %d -> %d not enabled (%s): copies go through the host\n", D.device, S.device,
                            hipGetErrorString(e));
            }
            (void)hipGetLastError();
        }
        HIP_CHECK(hipMemcpyPeerAsync(dst, D.device, src, S.device, nbytes, D.stream));
    }
    g_cur = keep;
    if (g.inited) bind_device();
}
// Slot `slot` enqueues on slot `with_slot`'s stream from now on (with_slot < 0: on its own again).  For the stages of ONE layer-split
// session that sit on one physical GPU (virtual slots; two stages of a split that has more stages than GPUs): they run strictly one
// after the other, so a second hardware queue buys nothing — and a wait on an event of ANOTHER queue resolves tens of microseconds
// late on this runtime (the hop of bench.py --mode split: 50-70 us per stage boundary, r06_sweeps.txt 6 / 20), where the same wait
// inside one queue is free.  Returns 1 if the slot now shares (same physical device, both initialised), 0 if nothing changed.
int ggml_hip_share_stream(int slot, int with_slot) {
    if (slot < 0 || slot >= GGML_HIP_MAX_BACKENDS || with_slot >= GGML_HIP_MAX_BACKENDS) die("ggml_hip_share_stream: bad slot");
    Backend &B = g_backends[slot];
    if (with_slot < 0 || with_slot == slot) {
        SlotLock lk(&B);
        if (!B.own_stream) return 0;
        Backend *keep = g_cur;
        g_cur = &B;
        bind_device();
        HIP_CHECK(hipStreamSynchronize(B.stream));
        B.stream = B.own_stream;
        B.own_stream = nullptr;
        g_cur = keep;
        if (g.inited) bind_device();
        return 0;
    }
    Backend &A = g_backends[with_slot];
    SlotLock lk_a(&g_backends[std::min(slot, with_slot)]);
    SlotLock lk_b(&g_backends[std::max(slot, with_slot)]);
    if (!A.inited || !B.inited || A.device != B.device || B.own_stream || A.own_stream) return 0;
    Backend *keep = g_cur;
    g_cur = &B;
    bind_device();
    HIP_CHECK(hipStreamSynchronize(B.stream));
    B.own_stream = B.stream;
    B.stream = A.stream;
    g_cur = keep;
    if (g.inited) bind_device();
    return 1;
}
void ggml_hip_synchronize(void) {
    SlotLock lk;
    if (g.inited) HIP_CHECK(hipStreamSynchronize(g.stream));
}
void ggml_hip_tensor_get(const struct ggml_tensor *tensor, void *host_dst, size_t offset, size_t nbytes) {
    SlotLock lk;
    ensure_init();
    d2h_queue(host_dst, dev_ptr(tensor) + offset, nbytes);
    d2h_finish();
}
void ggml_hip_tensor_set(struct ggml_tensor *tensor, const void *host_src, size_t offset, size_t nbytes) {
    SlotLock lk;
    ensure_init();
    h2d_bulk(dev_ptr(tensor) + offset, host_src, nbytes);
    HIP_CHECK(hipStreamSynchronize(g.stream));
}
// Raw copies on the backend stream, synchronous (layer-split driver: moving the residual between a stage's
// hand-off buffer and the communication library's buffers). kind: 0 = host→device, 1 = device→host, 2 = device→device.
void ggml_hip_memcpy(void *dst, const void *src, size_t nbytes, int kind) {
    SlotLock lk;
    ensure_init();
    if (kind == 0) {
        h2d_bulk((char *)dst, src, nbytes);
        HIP_CHECK(hipStreamSynchronize(g.stream));
    } else if (kind == 1) {
        d2h_queue(dst, (const char *)src, nbytes);
        d2h_finish();
    } else {
        HIP_CHECK(hipMemcpyAsync(dst, src, nbytes, hipMemcpyDeviceToDevice, g.stream));
        HIP_CHECK(hipStreamSynchronize(g.stream));
    }
}
void *ggml_hip_tensor_device_ptr(const struct ggml_tensor *tensor) {
    SlotLock lk;
    ensure_init();
    return dev_ptr(tensor);
}
void ggml_hip_timing_begin(void) {
    SlotLock lk;
    ensure_init();
    HIP_CHECK(hipStreamSynchronize(g.stream));
    for (int k = 0; k < GGML_HIP_KCLASS_COUNT; k++) {
        for (auto &r : g.timing.recs[k]) g.timing.pool.push_back(r);
        g.timing.recs[k].clear();
        g.timing.bytes[k] = 0;
        g.timing.launches[k] = 0;
    }
    g.timing.on = true;
}
void ggml_hip_timing_end(void) {
    SlotLock lk;
    g.timing.on = false;
    if (g.inited) HIP_CHECK(hipStreamSynchronize(g.stream));
}
void ggml_hip_timing_query(int kclass, double *ms, int64_t *launches, double *algo_bytes) {
    SlotLock lk;
    double total = 0;
    if (kclass >= 0 && kclass < GGML_HIP_KCLASS_COUNT) {
        for (auto &r : g.timing.recs[kclass]) {
            float t = 0;
            HIP_CHECK(hipEventElapsedTime(&t, r.a, r.b));
            total += t;
        }
        if (ms) *ms = total;
        if (launches) *launches = g.timing.launches[kclass];
        if (algo_bytes) *algo_bytes = g.timing.bytes[kclass];
    }
}
void ggml_hip_set_option(const char *key, int value) {
    // Options are process-wide.  Slots that exist get the value now; a slot initialised later replays the log (ensure_init):
    // setting an option never creates a stream, a context or a buffer on a device this process has not used yet.
    if (strcmp(key, "w16_release") != 0) {  // an action, not a state
        std::lock_guard<std::recursive_mutex> lk(g_mu);
        g_opt_log[key] = value;
    }
    bool known = false;
    for_each_slot([&] {
        if (g.inited) {
            ggml_hip_internal_set_option_here(key, value);
            known = true;
        }
    });
    if (!known) {  // no slot yet: validate the key (an unknown one must fail here, not at the first graph)
        Backend probe_state;
        Backend *keep = g_cur;
        g_cur = &probe_state;
        const bool device_key = !strcmp(key, "timeline") || !strcmp(key, "act_quant") || !strcmp(key, "w16_release") || !strcmp(key, "mmq_w16");
        if (!device_key) ggml_hip_internal_set_option_here(key, value);
        g_cur = keep;
    }
}
void ggml_hip_internal_set_option_here(const char *key, int value) {  // acts on the current slot
    const std::string k(key);
    g.opt_gen++;
    if (k == "fuse")
        g.opt_fuse = value;
    else if (k == "act_quant") {  // 0 = ggml's AVX2 activation quantizer (what the reference's build runs), 1 = its scalar branch
        g.opt_act_quant = value ? 1 : 0;
        apply_act_quant();
    }
    else if (k == "plan")
        g.opt_plan = value;
    else if (k == "graph")
        g.opt_graph = value;
    else if (k == "timeline") {
        drop_all_plans();
        if (g.timeline && value && (value == 1 ? 4 : value) != g.timeline_wgs) {
            HIP_CHECK(hipStreamSynchronize(g.stream));
            (void)hipFree(g.timeline);
            g.timeline = nullptr;
        }
        if (value && !g.timeline) {
            g.timeline_wgs = value == 1 ? 4 : value;
            g.timeline_bytes = (size_t)1024 * g.timeline_wgs * 8 * 8;
            HIP_CHECK(hipMalloc((void **)&g.timeline, g.timeline_bytes));
        }
        if (!value && g.timeline) {
            HIP_CHECK(hipStreamSynchronize(g.stream));
            (void)hipFree(g.timeline);
            g.timeline = nullptr;
        }
        if (g.timeline) HIP_CHECK(hipMemsetAsync(g.timeline, 0, g.timeline_bytes, g.stream));
    }
    else if (k == "fuse_heads") {
        if (g.opt_fuse_heads != value) drop_all_plans();
        g.opt_fuse_heads = value;
    }
    else if (k == "attn_one") {
        if (g.opt_attn_one != value) drop_all_plans();
        g.opt_attn_one = value;
    }
    else if (k == "attn_split") {
        if (g.opt_attn_split != value) drop_all_plans();
        g.opt_attn_split = value;
    }
    else if (k == "mmq_fuse")
        g.opt_mmq_fuse = value;
    else if (k == "chain_k")
        g.opt_chain_k = std::min(64, std::max(0, value));
    else if (k == "mmq_t256") {
        if (g.opt_mmq_t256 != value) drop_all_plans();
        g.opt_mmq_t256 = value;
    }
    else if (k == "mmq_w16") {
        if (g.opt_mmq_w16 != value) drop_all_plans();
        g.opt_mmq_w16 = value;
        if (!value) release_w16_copies();  // the copies are a cache of this option
    }
    else if (k == "w16_headroom_gb")
        g.opt_w16_headroom_gb = value;
    else if (k == "w16_release")  // drop the resident f16 weight copies now (they come back with the next prompt batch)
        release_w16_copies();
    else if (k == "mmq_cols") {
        if (g.opt_mmq_cols != value) drop_all_plans();
        g.opt_mmq_cols = value;
    }
    else if (k == "attn_fused")
        g.opt_attn_fused = value;
    else if (k == "plan_prompt") {
        if (g.opt_plan_prompt != value) drop_all_plans();
        g.opt_plan_prompt = value;
    }
    else if (k == "kbig") {
        if (g.opt_kbig != value) drop_all_plans();
        g.opt_kbig = value;
    }
    else if (k == "plan_k") {
        if (g.opt_plan_k != value) drop_all_plans();
        g.opt_plan_k = value;
    }
    else if (k == "plan_multi") {
        if (g.opt_plan_multi != value) drop_all_plans();
        g.opt_plan_multi = value;
    }
    else if (k == "fuse_attn") {
        if (g.opt_fuse_attn != value) drop_all_plans();
        g.opt_fuse_attn = value;
        g.fused_rearm_at = 0;  // an explicit choice outlives a pending re-arm
    }
    else if (k == "speculate_next") {  // 1 = run the greedy next token speculatively behind every single-token plan run (llama_plan.inc)
        spec_cancel();
        g.opt_speculate_next = value;
    }
    else if (k == "fuse_wo") {
        if (g.opt_fuse_wo != value) drop_all_plans();
        g.opt_fuse_wo = value;
    }
    else if (k == "prepare")
        g.opt_prepare = value;
    else if (k == "affine") {
        if (g.opt_affine != value) drop_all_plans();
        g.opt_affine = value;
    }
    else if (k == "warm_mb") {
        if (g.opt_warm_mb != value) drop_all_plans();
        g.opt_warm_mb = value;
    }
    else if (k == "big") {
        if (g.opt_big != value) drop_all_plans();
        g.opt_big = value;
    }
    else if (k == "serial_stage_slots") {  // how many of this device's session slots are stages of ONE split session (see device_sharers)
        if (g_dev_serial_stages[g.device & 63].exchange(value) != value) g_dev_gen[g.device & 63].fetch_add(1);  // captured graphs froze a choice of kernels
    }
    else if (k == "fused_rearm_tokens") {  // clean tokens on the two-launch forms after which the fused forms are taken back (0 = never)
        g.opt_fused_rearm_tokens = value;
        g.fused_rearm_stretch = 0;
    }
    else if (k == "fused_fallback")  // 1 = a token whose in-launch hand-off gave up is re-run on the two-launch forms; 0 = abort
        g.opt_fused_fallback = value;
    else if (k == "test_fused_timeout") {  // test hook: the attention workgroups of layer 0 of k_qkv_attn never get their rows
        if (g.opt_test_fused_timeout != value) drop_all_plans();
        g.opt_test_fused_timeout = value;
        if (!value) g.stat_fused_timeouts = 0;  // the hook's own give-ups do not count against the rest of the process

    }
    else if (k == "probe") {
        if (g.opt_probe != value) drop_all_plans();
        g.opt_probe = value;
    }
    else if (!strcmp(key, "mmq_min"))
        g.opt_mmq_min = value;
    else if (!strcmp(key, "k_prompt_min"))  // K-quant models: batch size from which the prompt plan (f16 copies) replaces the K plan's chunks
        g.opt_k_prompt_min = value;
    else if (!strcmp(key, "mmq_i8"))
        g.opt_mmq_i8 = value;
    else
        die("ggml_hip_set_option: unknown key '%s'", key);
}
// Roofline leg of bench.py: replays the kernels of ONE class of the most recent decode plan (e.g. the 129
// mat-vec launches of a LLaMA-7B token) `replays` times from a hipGraph that contains nothing else, bracketed
// by two HIP events on the backend stream.  Per-launch event pairs would add several µs of marker overhead
// to kernels that run for 2-15 µs; this measures the launches back to back instead (inter-kernel boundaries
// included, which rocprof's per-kernel durations exclude).  KV writes of the replay go to the last cache slot.
int ggml_hip_bench_plan_class(int kclass, int replays, double *ms_total, int64_t *launches_per_replay,
                              double *algo_bytes_per_replay) {
    SlotLock lk;
    unsigned kind_mask = ~0u;
    if (kclass >= GGML_HIP_KKIND_BASE && kclass < GGML_HIP_KKIND_BASE + 5) {  // one kind of mat-vec launch alone
        kind_mask = 1u << (kclass - GGML_HIP_KKIND_BASE);
        kclass = GGML_HIP_KCLASS_MMVQ;
    } else if (kclass >= GGML_HIP_KKIND_BASE + 8 && kclass < GGML_HIP_KKIND_BASE + 13) {  // every mat-vec launch BUT one kind
        kind_mask = ~(1u << (kclass - GGML_HIP_KKIND_BASE - 8));
        kclass = GGML_HIP_KCLASS_MMVQ;
    }
    if (g_plans.empty() || kclass < 0 || kclass >= GGML_HIP_KCLASS_COUNT || replays < 1) return -1;
    spec_cancel();  // the replays below run on the plan's buffers
    DecodePlan *p = g_plans.back();
    HIP_CHECK(hipStreamSynchronize(g.stream));
    DecParams saved, park;
    HIP_CHECK(hipMemcpy(&saved, p->prm, sizeof(saved), hipMemcpyDeviceToHost));
    park = saved;
    if (kclass == GGML_HIP_KCLASS_MMVQ) {  // K/V stores go to a scratch slot: the replay runs on the last token's stale activations
        if (g.opt_big)
            park.store_at = (int)p->m.C - 1;  // ... and the fused attention (k_qkv_attn) still meets the real context length
        else
            park.n_past = (int)p->m.C - 1;
    }
    HIP_CHECK(hipMemcpy(p->prm, &park, sizeof(park), hipMemcpyHostToDevice));
    PlanStats st;
    hipGraph_t gr = nullptr;
    hipGraphExec_t ex = nullptr;
    const bool was_on = g.timing.on;
    g.timing.on = false;
    hipEvent_t a, b;
    HIP_CHECK(hipEventCreate(&a));
    HIP_CHECK(hipEventCreate(&b));
    if (g.opt_graph) {
        HIP_CHECK(hipStreamBeginCapture(g.stream, hipStreamCaptureModeThreadLocal));
        plan_launch_decode(p, 1u << kclass, &st, kind_mask);
        HIP_CHECK(hipStreamEndCapture(g.stream, &gr));
        HIP_CHECK(hipGraphInstantiate(&ex, gr, nullptr, nullptr, 0));
        HIP_CHECK(hipGraphLaunch(ex, g.stream));  // warm
        HIP_CHECK(hipEventRecord(a, g.stream));
        for (int i = 0; i < replays; i++) HIP_CHECK(hipGraphLaunch(ex, g.stream));
        HIP_CHECK(hipEventRecord(b, g.stream));
    } else {  // GGML_HIP_GRAPH=0 (e.g. under rocprofv3, whose kernel tracing crashes on graph launches here)
        plan_launch_decode(p, 1u << kclass, &st, kind_mask);
        HIP_CHECK(hipEventRecord(a, g.stream));
        for (int i = 0; i < replays; i++) plan_launch_decode(p, 1u << kclass, nullptr, kind_mask);
        HIP_CHECK(hipEventRecord(b, g.stream));
    }
    HIP_CHECK(hipStreamSynchronize(g.stream));
    float ms = 0;
    HIP_CHECK(hipEventElapsedTime(&ms, a, b));
    HIP_CHECK(hipEventDestroy(a));
    HIP_CHECK(hipEventDestroy(b));
    if (ex) HIP_CHECK(hipGraphExecDestroy(ex));
    if (gr) HIP_CHECK(hipGraphDestroy(gr));
    HIP_CHECK(hipMemcpy(p->prm, &saved, sizeof(saved), hipMemcpyHostToDevice));
    g.timing.on = was_on;
    if (ms_total) *ms_total = ms;
    if (launches_per_replay) *launches_per_replay = st.launches[kclass];
    if (algo_bytes_per_replay) *algo_bytes_per_replay = st.bytes[kclass];
    return 0;
}

#include "backend_comm.inc"
#include "backend_tools.inc"
