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
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
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
#include "kernels/mmq_dma.h"
#include "kernels/mmq_dmap.h"
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
#include "kernels/decode_big8.h"
#include "kernels/mmq_cols.h"
#include "kernels/decode_attn_split.h"
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

namespace {

[[noreturn]] void die(const char *fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    fatal_note("%s", buf);
    abort();
}

int g_cur_slot();
struct Arena {
    uintptr_t base = 0;
    size_t size = 0;
    char *dev = nullptr;  // lazily allocated shadow
    bool live = true;
    bool is_scratch = false;  // registered through ggml_set_scratch (caller-owned Buffer, never unregistered)
};

// Record behind ggml_tensor.extra (and behind auto-uploaded persistent leaves).
struct DevTensor {
    uint32_t magic = 0x48495054;  // 'HIPT'
    int slot = (int)(g_cur_slot());  // the device slot that owns the copy
    uintptr_t host = 0;           // host data range this record mirrors
    size_t nbytes = 0;
    char *dev = nullptr;          // base of the device allocation
    size_t dev_bytes = 0;
    bool soa = false;             // quantized SoA layout (see QWeight)
    bool auto_uploaded = false;
    int refs = 1;                 // explicit records: tensors sharing this device copy
    bool zero_filled = false;     // created by assign_buffers_no_scratch (mutable state: never shared)
    uintptr_t owner_hdr = 0;      // auto-uploaded only: address of the ggml_tensor header that named this data
    QWeight qw{};
    char *w16 = nullptr;          // resident f16 copy of a quantized weight for the prompt GEMM (ensure_w16); qw.w16 names it
    size_t w16_size = 0;
    bool ksoa = false;            // K-quant planar layout (see KWeight, kernels/kquant.h)
    KWeight kw{};
    ggml_type type = GGML_TYPE_F32;
    int64_t ne[4] = {0, 0, 0, 0};
};

struct Timing {
    bool on = false;
    struct Rec {
        hipEvent_t a, b;
    };
    std::vector<Rec> recs[GGML_HIP_KCLASS_COUNT];
    std::vector<Rec> pool;
    double bytes[GGML_HIP_KCLASS_COUNT] = {0, 0, 0, 0};
    int64_t launches[GGML_HIP_KCLASS_COUNT] = {0, 0, 0, 0};
};

// One instance per device a process drives ("slot").  The reference's hooks address devices by index
// (ggml_cuda_set_main_device, crates/ggml/sys/src/cuda.rs:62; the split fractions of ggml_cuda_set_tensor_split, :11): here
// set_main_device makes a slot CURRENT and every entry point acts on the current slot — its stream, arena shadows, weight
// records, plan cache, options.  A session that spans several GPUs (the ggml-style layer split, host/llm_host.cpp) keeps the
// layers of a stage on one slot and switches slots between stages; the residual crosses with ggml_hip_copy_between_devices.
// GGML_HIP_VIRTUAL_DEVICES=n maps n slots onto the visible devices round-robin (several slots on ONE GPU: how the split is
// tested on a 1-GPU box).  One thread drives the library at a time (g_mu).
#define GGML_HIP_MAX_BACKENDS 16
struct Backend {
    int slot = 0;
    bool inited = false;
    int device = 0;
    hipStream_t stream = nullptr;
    int opt_attn_split = 1;  // long contexts: attention split over positions too (kernels/decode_attn_split.h)
    int opt_prefetch = 0;  // MB of w1|w3 (plus all of wo) that the idle CUs of the decode attention launch pull into the
                           // L2 of the XCD that will read them (llama_plan.inc, kernels/decode.h prefetch_slice); 0 = off
    int opt_prefetch_wo = 1;     // 0: leave wo out of the warm-up
    int opt_prefetch_delay = 0;  // ~0.2 us units before the warm-up's first request
    int opt_prefetch_wgs = 0;    // spare workgroups that take part (multiple of 8; 0 = all idle CUs): the warm-up's request rate
    std::map<uintptr_t, Arena> arenas;          // by base
    std::map<uintptr_t, DevTensor *> tensors;   // explicit records (transform_tensor / assign_buffers_no_scratch)
    std::map<uintptr_t, DevTensor *> auto_tensors;  // persistent leaves uploaded on first use (never offloaded by
                                                    // the caller); evicted when their host range is recycled
    // per-graph workspace (activation re-quantization, temp SoA): bump allocator over chunks; chunks
    // added mid-graph stay alive until the next graph starts, where they are merged into one.
    struct WsChunk {
        char *p;
        size_t size;
    };
    std::vector<WsChunk> ws_chunks;
    size_t ws_off = 0;  // offset in the last chunk
    Timing timing;
    // options
    int opt_fuse = 1;
    int opt_mmvq_rows = 0;  // 0 = auto
    int opt_plan_multi = 1; // fused plan for prompt chunks of 2..8 tokens (kernels/decode_big8.h)
    int opt_mmq_cols = 1;    // prompt chunks of 2..8 tokens: mat-muls on the integer matrix cores (kernels/mmq_cols.h) instead of k_mmvq_big8
    int opt_attn_fused = 1;  // prompt plan: K.Q, softmax and V.P as one launch with the scores in LDS (kernels/prompt_attn.h)
    int opt_plan_prompt = 1; // fused plan for prompt batches of >= mmq_min tokens (kernels/prompt.h)
    int opt_mmq_persist = 1; // prompt GEMM as a persistent kernel (kernels/mmq_dmap.h)
    int opt_mmq_w16 = 1;     // prompt GEMM on resident f16 copies of the quantized weights when HBM has room (kernels/mmq_w16.h)
    size_t w16_bytes = 0;    // HBM held by those copies
    uint64_t w16_gen = 1;    // bumped when copies are released: cached prompt plans re-resolve their pointers
    int opt_w16_headroom_gb = 16;  // HBM that must stay free after a copy is made (KV caches, workspaces, other models)
    int opt_mmq_t256 = 1;    // prompt GEMM on 256 x 256 tiles (kernels/mmq_w16_256.h) where the launch fills the chip with them; 2 = wherever legal (tests)
    int opt_mmq_t256_var = 0;  // measurement variants of k_mmq_w16_256 (see the kernel)
    int opt_mmq_waves = 8;   // waves per workgroup of the persistent prompt GEMM: 4 (mmq_dmap.h) or 8 (mmq_dmap8.h)
    int opt_mmq_fuse = 3;    // prompt plan: wq|wk|wv (bit 0) and w1|w3 (bit 1) as one GEMM launch each
    int opt_big = 1;        // decode mat-vec as one wave of 1024-thread workgroups (kernels/decode_big.h)
    int opt_probe = 0;      // measurement only: k_mmvq_big returns early (BigArgs::probe), tests/tools/launch_probe.py
    int num_cus = 256;
    long long *timeline = nullptr;  // device buffer of in-kernel timestamps (option "timeline")
    size_t timeline_bytes = 0;
    int timeline_wgs = 4;  // sampled workgroups per launch
    int opt_mmq_splitk = 1;
    int opt_mmq_splits = 0;  // measurement: K splits of the node-by-node executor's prompt GEMM launches (0 = the rule)
    int opt_mmq_dma = 1;    // prompt GEMM with LDS-DMA staging (kernels/mmq_dma.h) when K/32 is even; 2 = int8 activations
                            // dequantized in the kernel (13 KB instead of 20 KB per stage, 2x the VALU work: 413 vs 467 TFLOP/s)
    int opt_mmq_i8 = 0;     // 1 = prompt GEMM on the integer matrix cores (kernels/mmq_i8.h): ggml's exact block dots (error
                            // 2e-5 * scale instead of 1.1e-3), but the per-block scaling of every product is VALU-bound:
                            // 309 vs 464 TFLOP/s-equivalent on 7B Q4_0, so the f16 kernels stay the default
    int opt_mmq_xcdn = 0;   // pin XCDs to token tiles (measured slower than the tile-id walk: 393 vs 446 TFLOP/s)
    int opt_mmq_min = 32;   // token count from which mul_mat runs on the MFMA GEMM (0 = never)
    int opt_plan = 1;       // recognise the LLaMA decode graph and run the fused plan
    int opt_graph = 1;      // replay the plan from a captured hipGraph
    int opt_xsrc = 0;       // fuse norm / re-quantization into the mat-vec staging (see llama_plan.inc)
    uint64_t stat_plan_tokens = 0, stat_generic_graphs = 0, stat_split_tokens = 0, stat_prompt_plan_tokens = 0;
    // prompt-GEMM launches by kernel (ggml_hip_get_stat("mmq_launches_<name>")): bench.py labels its MFMA roofline with the
    // kernels that actually ran
    enum { MMQ_K_PLAIN, MMQ_K_DMA, MMQ_K_DMA_P, MMQ_K_DMA_P8, MMQ_K_W16_P8, MMQ_K_W16_256, MMQ_K_I8, MMQ_K_COUNT };
    uint64_t stat_mmq[MMQ_K_COUNT] = {0, 0, 0, 0, 0, 0, 0};
    void *chain_plan = nullptr;            // the plan a greedy chain may continue (set by its last single-token run)
    ggml_cgraph *chain_graph = nullptr;    // ... and the cgraph that run executed
    bool pending_wait = false;  // a decode plan was launched by graph_compute_begin and not yet waited for
    uint64_t ns_match = 0, ns_launch = 0, ns_wait = 0, ns_compute = 0;  // host-side time split of plan tokens
    size_t dead_shadow_bytes = 0;
};
std::recursive_mutex g_mu;
float g_tensor_split[GGML_HIP_MAX_BACKENDS] = {1.0f};
Backend g_backends[GGML_HIP_MAX_BACKENDS];
Backend *g_cur = &g_backends[0];
#define g (*g_cur)
int g_cur_slot() { return (int)(g_cur - g_backends); }
// a flag per slot for things done once per DEVICE (hipFuncSetAttribute acts on the current device's copy of a kernel)
struct DevOnce {
    bool done[GGML_HIP_MAX_BACKENDS] = {};
    bool first() {
        bool &d = done[g.slot];
        if (d) return false;
        d = true;
        return true;
    }
};

int kt_of(ggml_type t) {
    switch (t) {
        case GGML_TYPE_Q2_K: return KT_Q2_K;
        case GGML_TYPE_Q3_K: return KT_Q3_K;
        case GGML_TYPE_Q4_K: return KT_Q4_K;
        case GGML_TYPE_Q5_K: return KT_Q5_K;
        case GGML_TYPE_Q6_K: return KT_Q6_K;
        default: return -1;
    }
}
int qt_of(ggml_type t) {
    switch (t) {
        case GGML_TYPE_Q4_0: return QT_Q4_0;
        case GGML_TYPE_Q4_1: return QT_Q4_1;
        case GGML_TYPE_Q5_0: return QT_Q5_0;
        case GGML_TYPE_Q5_1: return QT_Q5_1;
        case GGML_TYPE_Q8_0: return QT_Q8_0;
        default: return -1;
    }
}

thread_local int tl_device = -1;  // the device this thread last made current
void bind_device() {
    if (tl_device != g.device) {
        HIP_CHECK(hipSetDevice(g.device));
        tl_device = g.device;
    }
}
// slots this process may address: GGML_HIP_VIRTUAL_DEVICES if set, else the visible devices
int slot_count() {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    if (const char *v = getenv("GGML_HIP_VIRTUAL_DEVICES")) n = std::max(n, atoi(v));
    return std::min(n, GGML_HIP_MAX_BACKENDS);
}
// runs f with every slot current in turn (host-side bookkeeping that all slots share: arenas, options)
template <typename F>
void for_each_slot(F f) {
    Backend *keep = g_cur;
    for (int i = 0; i < GGML_HIP_MAX_BACKENDS; i++) {
        g_cur = &g_backends[i];
        g.slot = i;
        if (g.inited) bind_device();
        f();
    }
    g_cur = keep;
    if (g.inited) bind_device();
}
void ensure_init() {
    if (g.inited) {
        bind_device();
        return;
    }
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        die("no HIP device available (hipGetDeviceCount -> %d, n=%d). This library has no CPU compute path.", (int)e, n);
    // slot s drives physical device (base + s) mod n.  base: GGML_HIP_DEVICE (one process per GPU launchers set it to the
    // local rank when they did not restrict visibility); several slots on one device when GGML_HIP_VIRTUAL_DEVICES asks
    // for more slots than there are devices.
    g.slot = (int)(g_cur - g_backends);
    int base = 0;
    if (const char *lr = getenv("GGML_HIP_DEVICE")) base = atoi(lr);
    g.device = (base + g.slot) % n;
    HIP_CHECK(hipSetDevice(g.device));
    tl_device = g.device;
    HIP_CHECK(hipStreamCreateWithFlags(&g.stream, hipStreamNonBlocking));
    if (const char *v = getenv("GGML_HIP_PREFETCH")) g.opt_prefetch = atoi(v);
    if (const char *v = getenv("GGML_HIP_PREFETCH_WO")) g.opt_prefetch_wo = atoi(v);
    if (const char *v = getenv("GGML_HIP_PREFETCH_DELAY")) g.opt_prefetch_delay = atoi(v);
    if (const char *v = getenv("GGML_HIP_PREFETCH_WGS")) g.opt_prefetch_wgs = atoi(v);
    if (const char *v = getenv("GGML_HIP_ATTN_SPLIT")) g.opt_attn_split = atoi(v);
    if (const char *v = getenv("GGML_HIP_FUSE")) g.opt_fuse = atoi(v);
    if (const char *v = getenv("GGML_HIP_PLAN")) g.opt_plan = atoi(v);
    if (const char *v = getenv("GGML_HIP_GRAPH")) g.opt_graph = atoi(v);
    if (const char *v = getenv("GGML_HIP_XSRC")) g.opt_xsrc = atoi(v);
    if (const char *v = getenv("GGML_HIP_MMVQ_R")) g.opt_mmvq_rows = atoi(v);
    if (const char *v = getenv("GGML_HIP_MMQ_MIN")) g.opt_mmq_min = atoi(v);
    if (const char *v = getenv("GGML_HIP_BIG")) g.opt_big = atoi(v);
    if (const char *v = getenv("GGML_HIP_PLAN_MULTI")) g.opt_plan_multi = atoi(v);
    if (const char *v = getenv("GGML_HIP_PLAN_PROMPT")) g.opt_plan_prompt = atoi(v);
    if (const char *v = getenv("GGML_HIP_ATTN_FUSED")) g.opt_attn_fused = atoi(v);
    if (const char *v = getenv("GGML_HIP_MMQ_COLS")) g.opt_mmq_cols = atoi(v);
    if (const char *v = getenv("GGML_HIP_MMQ_FUSE")) g.opt_mmq_fuse = atoi(v);
    if (const char *v = getenv("GGML_HIP_MMQ_PERSIST")) g.opt_mmq_persist = atoi(v);
    if (const char *v = getenv("GGML_HIP_MMQ_WAVES")) g.opt_mmq_waves = atoi(v);
    if (const char *v = getenv("GGML_HIP_MMQ_W16")) g.opt_mmq_w16 = atoi(v);
    if (const char *v = getenv("GGML_HIP_MMQ_T256")) g.opt_mmq_t256 = atoi(v);
    if (const char *v = getenv("GGML_HIP_W16_HEADROOM_GB")) g.opt_w16_headroom_gb = atoi(v);
    if (const char *v = getenv("GGML_HIP_MMQ_XCDN")) g.opt_mmq_xcdn = atoi(v);
    if (const char *v = getenv("GGML_HIP_MMQ_DMA")) g.opt_mmq_dma = atoi(v);
    if (const char *v = getenv("GGML_HIP_MMQ_I8")) g.opt_mmq_i8 = atoi(v);
    if (const char *v = getenv("GGML_HIP_MMQ_SPLITK")) g.opt_mmq_splitk = atoi(v);
    {
        hipDeviceProp_t prop;
        HIP_CHECK(hipGetDeviceProperties(&prop, g.device));
        g.num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
        if (const char *v = getenv("GGML_HIP_BIG_WGS")) g.num_cus = std::max(1, atoi(v));
    }
    g.inited = true;
}

// ---------------------------------------------------------------------------------------------------
// address translation
// ---------------------------------------------------------------------------------------------------
Arena *find_arena(uintptr_t p) {
    auto it = g.arenas.upper_bound(p);
    if (it == g.arenas.begin()) return nullptr;
    --it;
    Arena &a = it->second;
    if (p >= a.base && p < a.base + a.size && a.live) return &a;
    return nullptr;
}
DevTensor *find_in(std::map<uintptr_t, DevTensor *> &m, uintptr_t p) {
    auto it = m.upper_bound(p);
    if (it == m.begin()) return nullptr;
    --it;
    DevTensor *t = it->second;
    if (p >= t->host && p < t->host + std::max<size_t>(t->nbytes, 1)) return t;
    return nullptr;
}
// explicit records never overlap each other (enforced at registration), so the nearest base below p decides;
// auto records are consulted second and are evicted whenever anything else claims their host range.
DevTensor *find_tensor(uintptr_t p) {
    if (DevTensor *t = find_in(g.tensors, p)) return t;
    return find_in(g.auto_tensors, p);
}
void destroy_record(DevTensor *e);
size_t release_w16_copies();
// Every device allocation of the library goes through here.  The resident f16 weight copies of the prompt GEMM
// (ensure_w16) are a CACHE: when HBM runs out (a second model, a long-context KV cache, score workspaces) they are
// released, the plans that name them dropped, and the allocation retried before giving up.
void dev_malloc(void **p, size_t bytes, const char *what) {
    if (hipMalloc(p, bytes) == hipSuccess) return;
    (void)hipGetLastError();
    const size_t freed = release_w16_copies();
    if (freed && hipMalloc(p, bytes) == hipSuccess) {
        fprintf(stderr, "libggml_hip: released %.2f GB of resident f16 weight copies to allocate %.2f GB for %s\n", freed / 1e9,
                bytes / 1e9, what);
        return;
    }
    (void)hipGetLastError();
    size_t free_b = 0, total_b = 0;
    (void)hipMemGetInfo(&free_b, &total_b);
    die("out of device memory: %s needs %zu bytes, %zu of %zu free", what, bytes, free_b, total_b);
}
// drops every record of `m` whose host range intersects [b, b+size)
void evict_overlapping(std::map<uintptr_t, DevTensor *> &m, uintptr_t b, size_t size) {
    for (auto it = m.begin(); it != m.end();) {
        DevTensor *e = it->second;
        if (e->host < b + size && b < e->host + std::max<size_t>(e->nbytes, 1)) {
            it = m.erase(it);
            destroy_record(e);
        } else {
            ++it;
        }
    }
}
DevTensor *extra_of(const ggml_tensor *t) {
    DevTensor *e = (DevTensor *)t->extra;
    if (e && e->magic != 0x48495054) die("tensor '%s': extra does not belong to this backend", t->name);
    if (e && e->slot != g_cur_slot())
        die("tensor '%s' lives on device slot %d but slot %d is current (ggml_hip_set_main_device)", t->name, e->slot, g_cur_slot());
    return e;
}
char *arena_dev(Arena *a) {
    if (!a->dev) {
        ensure_init();
        dev_malloc((void **)&a->dev, a->size, "an arena shadow");
    }
    return a->dev;
}

// device address of the raw bytes of `t` (strided views included). Aborts for SoA weights.
char *dev_ptr(const ggml_tensor *t) {
    if (DevTensor *e = extra_of(t)) {
        if (e->soa) die("tensor '%s' is a re-laid-out quantized weight; only mul_mat/get_rows may read it", t->name);
        return e->dev + ((uintptr_t)t->data - e->host);
    }
    const uintptr_t p = (uintptr_t)t->data;
    if (p == 0) die("tensor '%s' has no data", t->name);
    if (DevTensor *e = find_tensor(p)) {
        if (e->soa) die("tensor '%s' aliases a re-laid-out quantized weight", t->name);
        return e->dev + (p - e->host);
    }
    if (Arena *a = find_arena(p)) return arena_dev(a) + (p - a->base);
    die("tensor '%s' (op %s): data pointer %p is in no registered arena and has no device copy", t->name,
        ggml_op_name(t->op), t->data);
}

TView view_of(const ggml_tensor *t) {
    TView v;
    v.p = dev_ptr(t);
    for (int i = 0; i < 4; i++) {
        v.ne[i] = t->ne[i];
        v.nb[i] = (int64_t)t->nb[i];
    }
    return v;
}

// ---------------------------------------------------------------------------------------------------
// workspace
// ---------------------------------------------------------------------------------------------------
struct QActBuf {
    const void *src_data = nullptr;  // host address identity of the source tensor data
    size_t src_bytes = 0;
    bool f16_d = false;
    int64_t nb = 0, ncols = 0;
    QAct act{};
    bool valid = false;
} g_qact_[GGML_HIP_MAX_BACKENDS];
#define g_qact (g_qact_[g.slot])
struct XF16Buf {  // the prefill GEMM's activation operand (kernels/mmq.h), cached like g_qact
    const void *src_data = nullptr;
    size_t src_bytes = 0;
    bool f16_d = false;
    int64_t nb = 0, ncols = 0;
    const _Float16 *x = nullptr;
    bool valid = false;
} g_xf16_[GGML_HIP_MAX_BACKENDS];
#define g_xf16 (g_xf16_[g.slot])
struct XQ8Buf {  // int8 + f16-scale activations of the X8 prompt GEMM (kernels/mmq_dma.h)
    const void *src_data = nullptr;
    size_t src_bytes = 0;
    bool f16_d = false;
    int64_t nb = 0, ncols = 0;
    const int8_t *q8 = nullptr;
    const _Float16 *dx = nullptr;
    bool valid = false;
} g_xq8_[GGML_HIP_MAX_BACKENDS];
#define g_xq8 (g_xq8_[g.slot])

char *ws_alloc(size_t bytes) {
    bytes = (bytes + 255) & ~(size_t)255;
    if (g.ws_chunks.empty() || g.ws_off + bytes > g.ws_chunks.back().size) {
        size_t total = 0;
        for (auto &c : g.ws_chunks) total += c.size;
        const size_t want = std::max<size_t>(std::max<size_t>(bytes, total), (size_t)256 << 20);
        Backend::WsChunk c{nullptr, want};
        dev_malloc((void **)&c.p, want, "the graph workspace");
        g.ws_chunks.push_back(c);
        g.ws_off = 0;
    }
    char *p = g.ws_chunks.back().p + g.ws_off;
    g.ws_off += bytes;
    return p;
}
// called at the start of every graph: rewind, and merge chunks that were added during the last graph
void ws_reset() {
    g.ws_off = 0;
    if (g.ws_chunks.size() <= 1) return;
    HIP_CHECK(hipStreamSynchronize(g.stream));
    size_t total = 0;
    for (auto &c : g.ws_chunks) {
        total += c.size;
        HIP_CHECK(hipFree(c.p));
    }
    g.ws_chunks.clear();
    Backend::WsChunk c{nullptr, total};
    dev_malloc((void **)&c.p, total, "the graph workspace");
    g.ws_chunks.push_back(c);
}

// ---------------------------------------------------------------------------------------------------
// host <-> device transfers through library-owned pinned staging.  Caller memory (ggml arenas, mmap'd
// weight files, numpy buffers) is pageable and may be unmapped/recycled at any time; copying through
// our own hipHostMalloc'd buffers keeps the runtime from pinning (and caching pins of) memory it does
// not own, and makes the small per-evaluation uploads truly asynchronous.
// ---------------------------------------------------------------------------------------------------
struct Staging {
    static constexpr size_t BIG = (size_t)32 << 20;   // bulk chunk (weights upload, large read-backs)
    static constexpr size_t SMALL = (size_t)8 << 20;  // per-graph bump region (token ids, constants, logits)
    char *big[2] = {nullptr, nullptr};
    hipEvent_t ev[2] = {nullptr, nullptr};
    bool busy[2] = {false, false};
    int next = 0;
    char *small = nullptr;
    size_t small_off = 0;
    struct Pending {
        void *host_dst;
        const char *stage;
        size_t n;
    };
    std::vector<Pending> pending;  // D2H copies whose staging → user memcpy happens after the stream sync
} stg_[GGML_HIP_MAX_BACKENDS];
#define stg (stg_[g.slot])

void staging_init() {
    if (stg.small) return;
    for (int i = 0; i < 2; i++) {
        HIP_CHECK(hipHostMalloc((void **)&stg.big[i], Staging::BIG, hipHostMallocDefault));
        HIP_CHECK(hipEventCreateWithFlags(&stg.ev[i], hipEventDisableTiming));
    }
    HIP_CHECK(hipHostMalloc((void **)&stg.small, Staging::SMALL, hipHostMallocDefault));
}

// bulk upload; returns when the data has left `src` (the device copy is ordered on g.stream)
void h2d_bulk(char *dst, const void *src, size_t n) {
    staging_init();
    size_t off = 0;
    while (off < n) {
        const int b = stg.next;
        stg.next ^= 1;
        if (stg.busy[b]) {
            HIP_CHECK(hipEventSynchronize(stg.ev[b]));
            stg.busy[b] = false;
        }
        const size_t len = std::min(Staging::BIG, n - off);
        memcpy(stg.big[b], (const char *)src + off, len);
        HIP_CHECK(hipMemcpyAsync(dst + off, stg.big[b], len, hipMemcpyHostToDevice, g.stream));
        HIP_CHECK(hipEventRecord(stg.ev[b], g.stream));
        stg.busy[b] = true;
        off += len;
    }
}
// small asynchronous upload for the current graph (staging lives until the graph's final sync)
void h2d_small(char *dst, const void *src, size_t n) {
    staging_init();
    const size_t need = (n + 63) & ~(size_t)63;
    if (stg.small_off + need > Staging::SMALL) {
        h2d_bulk(dst, src, n);
        return;
    }
    char *s = stg.small + stg.small_off;
    stg.small_off += need;
    memcpy(s, src, n);
    HIP_CHECK(hipMemcpyAsync(dst, s, n, hipMemcpyHostToDevice, g.stream));
}
// read-back: queued on the stream now, delivered to user memory by d2h_finish() (which synchronises)
void d2h_queue(void *host_dst, const char *src, size_t n) {
    staging_init();
    const size_t need = (n + 63) & ~(size_t)63;
    if (stg.small_off + need > Staging::SMALL) {
        // large (the [V, N] logits of a prompt batch: 65 MB at N = 512): 8 MiB pieces, the DMA of piece i+1 into one
        // pinned buffer overlaps the copy of piece i out of the other; that copy is split over 4 threads (one
        // thread moves ~8 GB/s into pageable memory, which made the read-back ~20 % of a 512-token batch)
        const size_t PIECE = (size_t)8 << 20;
        for (int i = 0; i < 2; i++)
            if (stg.busy[i]) {
                HIP_CHECK(hipEventSynchronize(stg.ev[i]));
                stg.busy[i] = false;
            }
        auto par_copy = [](char *dst, const char *src_, size_t len) {
            const int nt = len >= ((size_t)2 << 20) ? 4 : 1;
            if (nt == 1) {
                memcpy(dst, src_, len);
                return;
            }
            std::thread th[3];
            const size_t part = (len / nt + 63) & ~(size_t)63;
            for (int t = 1; t < nt; t++) {
                const size_t o = (size_t)t * part;
                if (o < len) th[t - 1] = std::thread([=] { memcpy(dst + o, src_ + o, std::min(part, len - o)); });
            }
            memcpy(dst, src_, std::min(part, len));
            for (int t = 1; t < nt; t++)
                if (th[t - 1].joinable()) th[t - 1].join();
        };
        size_t off = 0;
        int cur = 0;
        size_t len = std::min(PIECE, n);
        HIP_CHECK(hipMemcpyAsync(stg.big[cur], src, len, hipMemcpyDeviceToHost, g.stream));
        HIP_CHECK(hipEventRecord(stg.ev[cur], g.stream));
        while (off < n) {
            const size_t noff = off + len, nlen = noff < n ? std::min(PIECE, n - noff) : 0;
            if (nlen) {
                HIP_CHECK(hipMemcpyAsync(stg.big[cur ^ 1], src + noff, nlen, hipMemcpyDeviceToHost, g.stream));
                HIP_CHECK(hipEventRecord(stg.ev[cur ^ 1], g.stream));
            }
            HIP_CHECK(hipEventSynchronize(stg.ev[cur]));
            par_copy((char *)host_dst + off, stg.big[cur], len);
            off = noff;
            len = nlen;
            cur ^= 1;
        }
        return;
    }
    char *s = stg.small + stg.small_off;
    stg.small_off += need;
    HIP_CHECK(hipMemcpyAsync(s, src, n, hipMemcpyDeviceToHost, g.stream));
    stg.pending.push_back({host_dst, s, n});
}
void d2h_finish() {
    HIP_CHECK(hipStreamSynchronize(g.stream));
    for (auto &p : stg.pending) memcpy(p.host_dst, p.stage, p.n);
    stg.pending.clear();
    stg.small_off = 0;
    stg.busy[0] = stg.busy[1] = false;
}

// ---------------------------------------------------------------------------------------------------
// timing (HIP events on the backend stream, per kernel class)
// ---------------------------------------------------------------------------------------------------
struct Timed {
    int k;
    Timing::Rec rec{};
    bool on;
    Timed(int kclass, double algo_bytes) : k(kclass), on(g.timing.on) {
        if (!on) return;
        if (!g.timing.pool.empty()) {
            rec = g.timing.pool.back();
            g.timing.pool.pop_back();
        } else {
            HIP_CHECK(hipEventCreate(&rec.a));
            HIP_CHECK(hipEventCreate(&rec.b));
        }
        HIP_CHECK(hipEventRecord(rec.a, g.stream));
        g.timing.bytes[k] += algo_bytes;
        g.timing.launches[k]++;
    }
    ~Timed() {
        if (!on) return;
        HIP_CHECK(hipEventRecord(rec.b, g.stream));
        g.timing.recs[k].push_back(rec);
    }
};

inline dim3 grid1(int64_t n, int bs = 256) { return dim3((unsigned)((n + bs - 1) / bs)); }

// ---------------------------------------------------------------------------------------------------
// persistent device tensors
// ---------------------------------------------------------------------------------------------------
size_t qw_layout(int qt, int64_t nblocks, size_t off[5]) {
    // returns total bytes; off = {qs, qs2, qh, d, m}
    auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
    size_t o = 0;
    off[0] = o;
    o = al(o + (size_t)nblocks * 16);
    off[1] = o;
    if (qt == QT_Q8_0) o = al(o + (size_t)nblocks * 16);
    off[2] = o;
    if (qt == QT_Q5_0 || qt == QT_Q5_1) o = al(o + (size_t)nblocks * 4);
    off[3] = o;
    o = al(o + (size_t)nblocks * 2);
    off[4] = o;
    if (qt == QT_Q4_1 || qt == QT_Q5_1) o = al(o + (size_t)nblocks * 2);
    return o;
}

QWeight qw_at(char *base, int qt, int64_t M, int64_t nb) {
    size_t off[5];
    qw_layout(qt, M * nb, off);
    QWeight w;
    w.qs = (const uint8_t *)(base + off[0]);
    w.qs2 = (const uint8_t *)(base + off[1]);
    w.qh = (const uint32_t *)(base + off[2]);
    w.d = (const __half *)(base + off[3]);
    w.m = (const __half *)(base + off[4]);
    w.M = M;
    w.nb = nb;
    w.qt = qt;
    w.w16 = nullptr;
    return w;
}

void relayout_launch(const char *raw_dev, int qt, int64_t M, int64_t nb, char *soa_base) {
    QWeight w = qw_at(soa_base, qt, M, nb);
    const int64_t nblocks = M * nb;
    hipLaunchKernelGGL(k_relayout_q, grid1(nblocks), dim3(256), 0, g.stream, (const uint8_t *)raw_dev, qt, nblocks,
                       (uint8_t *)w.qs, (uint8_t *)w.qs2, (uint32_t *)w.qh, (__half *)w.d, (__half *)w.m);
    HIP_CHECK(hipGetLastError());
}

bool wants_soa(const ggml_tensor *t) {
    return qt_of(t->type) >= 0 && t->ne[2] == 1 && t->ne[3] == 1 && t->ne[0] % 32 == 0 && ggml_is_contiguous(t);
}

// K-quants (kernels/kquant.h, kquant2.h): planes {qs, aux (Q6_K high bits / Q3_K hmask / Q5_K qh), sc, d}; bytes per super-block
struct KPlanes { int qs, aux, sc, d; };
KPlanes k_planes(int kt) {
    switch (kt) {
        case KT_Q4_K: return {128, 0, 16, 4};
        case KT_Q6_K: return {128, 64, 16, 2};
        case KT_Q2_K: return {64, 0, 16, 4};
        case KT_Q3_K: return {64, 32, 16, 4};
        default: return {128, 32, 16, 4};  // KT_Q5_K
    }
}
double k_block_bytes(int kt) { return kt == KT_Q4_K ? 144.0 : kt == KT_Q6_K ? 210.0 : kt == KT_Q2_K ? 84.0 : kt == KT_Q3_K ? 110.0 : 176.0; }
size_t kw_layout(int kt, int64_t nsbt, size_t off[4]) {
    auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
    const KPlanes pl = k_planes(kt);
    size_t o = 0;
    off[0] = o;
    o = al(o + (size_t)nsbt * pl.qs);
    off[1] = o;
    if (pl.aux) o = al(o + (size_t)nsbt * pl.aux);
    off[2] = o;
    o = al(o + (size_t)nsbt * pl.sc);
    off[3] = o;
    o = al(o + (size_t)nsbt * pl.d);
    return o;
}
// rows row0 .. of a planar K weight (every plane is row-major)
KWeight kw_rows(KWeight w, int64_t row0, int64_t rows) {
    const KPlanes pl = k_planes(w.kt);
    const int64_t s = row0 * w.nsb;
    w.qs += s * pl.qs;
    w.aux += s * pl.aux / 4;
    w.sc += s * pl.sc;
    w.d += s * pl.d / 2;
    w.M = rows;
    return w;
}
KWeight kw_at(char *base, int kt, int64_t M, int64_t nsb) {
    size_t off[4];
    kw_layout(kt, M * nsb, off);
    KWeight w;
    w.qs = (const uint8_t *)(base + off[0]);
    w.aux = (const uint32_t *)(base + off[1]);
    w.sc = (const uint8_t *)(base + off[2]);
    w.d = (const __half *)(base + off[3]);
    w.M = M;
    w.nsb = nsb;
    w.kt = kt;
    return w;
}
void relayout_k_launch(const char *raw_dev, int kt, int64_t M, int64_t nsb, char *base) {
    const KWeight w = kw_at(base, kt, M, nsb);
    if (kt == KT_Q4_K || kt == KT_Q6_K)
        hipLaunchKernelGGL(k_relayout_k, grid1(M * nsb * 8), dim3(256), 0, g.stream, (const uint8_t *)raw_dev, kt, M * nsb,
                           (uint8_t *)w.qs, (uint32_t *)w.aux, (uint8_t *)w.sc, (__half *)w.d);
    else
        hipLaunchKernelGGL(k_relayout_k2, grid1(M * nsb), dim3(256), 0, g.stream, (const uint8_t *)raw_dev, kt, M * nsb,
                           (uint8_t *)w.qs, (uint32_t *)w.aux, (uint8_t *)w.sc, (__half *)w.d);
    HIP_CHECK(hipGetLastError());
}
bool wants_ksoa(const ggml_tensor *t) {
    return kt_of(t->type) >= 0 && t->ne[2] == 1 && t->ne[3] == 1 && t->ne[0] % 256 == 0 && ggml_is_contiguous(t);
}

// Uploads `nbytes` from host `data` as the device copy of `t`. Returns the record (registered in g.tensors).
DevTensor *upload_tensor(const void *data, const ggml_tensor *t, bool zero_fill, bool is_auto = false) {
    ensure_init();
    const size_t nbytes = ggml_nbytes(t);
    if (!is_auto && !zero_fill) {
        // The same host bytes offloaded again (two models over one mmap / one weight buffer): share the
        // device copy instead of evicting it from under the first owner.
        auto it = g.tensors.find((uintptr_t)data);
        if (it != g.tensors.end()) {
            DevTensor *o = it->second;
            if (o->nbytes == nbytes && o->type == t->type && o->ne[0] == t->ne[0] && o->ne[1] == t->ne[1] &&
                o->ne[2] == t->ne[2] && o->ne[3] == t->ne[3] && !o->zero_filled) {
                o->refs++;
                return o;
            }
        }
    }
    DevTensor *e = new DevTensor();
    e->zero_filled = zero_fill;
    e->host = (uintptr_t)data;
    e->nbytes = nbytes;
    e->type = t->type;
    for (int i = 0; i < 4; i++) e->ne[i] = t->ne[i];
    if (!zero_fill && wants_soa(t)) {
        const int qt = qt_of(t->type);
        const int64_t M = t->ne[1], nb = t->ne[0] / 32;
        size_t off[5];
        const size_t total = qw_layout(qt, M * nb, off);
        dev_malloc((void **)&e->dev, total, "a weight tensor");
        e->dev_bytes = total;
        char *tmp = nullptr;
        dev_malloc((void **)&tmp, nbytes, "an upload staging buffer");
        h2d_bulk(tmp, data, nbytes);
        relayout_launch(tmp, qt, M, nb, e->dev);
        HIP_CHECK(hipStreamSynchronize(g.stream));
        HIP_CHECK(hipFree(tmp));
        e->soa = true;
        e->qw = qw_at(e->dev, qt, M, nb);
    } else if (!zero_fill && wants_ksoa(t)) {
        const int kt = kt_of(t->type);
        const int64_t M = t->ne[1], nsb = t->ne[0] / 256;
        size_t off[4];
        const size_t total = kw_layout(kt, M * nsb, off);
        dev_malloc((void **)&e->dev, total, "a weight tensor");
        e->dev_bytes = total;
        char *tmp = nullptr;
        dev_malloc((void **)&tmp, nbytes, "an upload staging buffer");
        h2d_bulk(tmp, data, nbytes);
        relayout_k_launch(tmp, kt, M, nsb, e->dev);
        HIP_CHECK(hipStreamSynchronize(g.stream));
        HIP_CHECK(hipFree(tmp));
        e->ksoa = true;
        e->kw = kw_at(e->dev, kt, M, nsb);
    } else {
        dev_malloc((void **)&e->dev, std::max<size_t>(nbytes, 16), "a persistent tensor");
        e->dev_bytes = nbytes;
        if (zero_fill)
            HIP_CHECK(hipMemsetAsync(e->dev, 0, std::max<size_t>(nbytes, 16), g.stream));
        else
            h2d_bulk(e->dev, data, nbytes);
        HIP_CHECK(hipStreamSynchronize(g.stream));
    }
    e->auto_uploaded = is_auto;
    // whoever held this host range before is gone (the memory was recycled)
    evict_overlapping(g.auto_tensors, e->host, std::max<size_t>(nbytes, 1));
    if (is_auto) {
        g.auto_tensors[e->host] = e;
    } else {
        evict_overlapping(g.tensors, e->host, std::max<size_t>(nbytes, 1));
        g.tensors[e->host] = e;
    }
    return e;
}

void drop_all_plans();
void destroy_record(DevTensor *e) {
    drop_all_plans();  // cached decode plans hold device addresses of weight / KV records
    if (g.stream) HIP_CHECK(hipStreamSynchronize(g.stream));
    if (e->dev) HIP_CHECK(hipFree(e->dev));
    if (e->w16) {
        HIP_CHECK(hipFree(e->w16));
        g.w16_bytes -= e->w16_size;
    }
    e->magic = 0;
    delete e;
}
void free_dev_tensor(DevTensor *e) {
    if (--e->refs > 0) return;
    auto &m = e->auto_uploaded ? g.auto_tensors : g.tensors;
    auto it = m.find(e->host);
    if (it != m.end() && it->second == e) m.erase(it);
    destroy_record(e);
}

// the SoA view of a quantized mul_mat / get_rows operand; re-lays-out on the fly for raw arena tensors
QWeight qweight_of(const ggml_tensor *t) {
    DevTensor *e = extra_of(t);
    if (!e) e = find_tensor((uintptr_t)t->data);
    if (e && e->soa) {
        if ((uintptr_t)t->data != e->host || t->ne[0] != e->ne[0] || t->ne[1] != e->ne[1])
            die("tensor '%s': views of re-laid-out quantized weights are not supported", t->name);
        return e->qw;
    }
    if (!wants_soa(t)) die("tensor '%s': quantized operand must be a contiguous 2-D matrix with ne0 %% 32 == 0", t->name);
    // raw GGML blocks in an arena (e.g. a weight created in the compute context): convert into workspace
    const int qt = qt_of(t->type);
    const int64_t M = t->ne[1], nb = t->ne[0] / 32;
    size_t off[5];
    const size_t total = qw_layout(qt, M * nb, off);
    char *raw = dev_ptr(t);
    char *soa = ws_alloc(total);
    relayout_launch(raw, qt, M, nb, soa);
    return qw_at(soa, qt, M, nb);
}

KWeight kweight_of(const ggml_tensor *t) {
    DevTensor *e = extra_of(t);
    if (!e) e = find_tensor((uintptr_t)t->data);
    if (e && e->ksoa) {
        if ((uintptr_t)t->data != e->host || t->ne[0] != e->ne[0] || t->ne[1] != e->ne[1])
            die("tensor '%s': views of re-laid-out quantized weights are not supported", t->name);
        return e->kw;
    }
    if (!wants_ksoa(t)) die("tensor '%s': K-quant operand must be a contiguous 2-D matrix with ne0 %% 256 == 0", t->name);
    const int kt = kt_of(t->type);
    const int64_t M = t->ne[1], nsb = t->ne[0] / 256;
    size_t off[4];
    const size_t total = kw_layout(kt, M * nsb, off);
    char *soa = ws_alloc(total);
    relayout_k_launch(dev_ptr(t), kt, M, nsb, soa);
    return kw_at(soa, kt, M, nsb);
}

// ---------------------------------------------------------------------------------------------------
// op launchers
// ---------------------------------------------------------------------------------------------------
bool is_contig_f32(const ggml_tensor *t) { return t->type == GGML_TYPE_F32 && ggml_is_contiguous(t); }

QAct quantize_activation(const ggml_tensor *src1, bool f16_d) {
    BK_ASSERT(src1->type == GGML_TYPE_F32 && src1->nb[0] == 4 && src1->ne[2] == 1 && src1->ne[3] == 1);
    const int64_t K = src1->ne[0], N = src1->ne[1], nb = K / 32;
    if (g_qact.valid && g_qact.src_data == src1->data && g_qact.f16_d == f16_d && g_qact.nb == nb && g_qact.ncols == N)
        return g_qact.act;
    const size_t nblk = (size_t)nb * N;
    char *blk = ws_alloc(nblk * 40);
    char *lo = blk, *hi = blk + nblk * 16, *d = blk + nblk * 32, *s = blk + nblk * 36;
    const char *x = dev_ptr(src1);
    Timed tm(GGML_HIP_KCLASS_OTHER, (double)(K * N * 4 + nblk * 40));
    const int64_t threads = (int64_t)nblk * 32;
    if (f16_d)
        hipLaunchKernelGGL(k_quantize_act<true>, grid1(threads), dim3(256), 0, g.stream, x, (int64_t)src1->nb[1], nb, N,
                           (int8_t *)lo, (int8_t *)hi, (float *)d, (int *)s);
    else
        hipLaunchKernelGGL(k_quantize_act<false>, grid1(threads), dim3(256), 0, g.stream, x, (int64_t)src1->nb[1], nb,
                           N, (int8_t *)lo, (int8_t *)hi, (float *)d, (int *)s);
    HIP_CHECK(hipGetLastError());
    g_qact.valid = true;
    g_qact.src_data = src1->data;
    g_qact.src_bytes = ggml_nbytes(src1);
    g_qact.f16_d = f16_d;
    g_qact.nb = nb;
    g_qact.ncols = N;
    g_qact.act = QAct{(const i32x4 *)lo, (const i32x4 *)hi, (const float *)d, (const int *)s};
    return g_qact.act;
}

size_t blk_bytes(int qt) { return qt == QT_Q4_0 ? 18 : qt == QT_Q4_1 ? 20 : qt == QT_Q5_0 ? 22 : qt == QT_Q5_1 ? 24 : 34; }

template <int QT, int NCOLS>
void launch_mmvq_r(const MmvqArgs &a, int R, int nwg, size_t lds) {
    switch (R) {
        case 1: hipLaunchKernelGGL((k_mmvq<QT, NCOLS, 1>), dim3(nwg), dim3(256), lds, g.stream, a); break;
        case 2: hipLaunchKernelGGL((k_mmvq<QT, NCOLS, 2>), dim3(nwg), dim3(256), lds, g.stream, a); break;
        default: hipLaunchKernelGGL((k_mmvq<QT, NCOLS, 4>), dim3(nwg), dim3(256), lds, g.stream, a); break;
    }
}
template <int QT>
void launch_mmvq_c(const MmvqArgs &a, int ncols, int R, int nwg, size_t lds) {
    switch (ncols) {
        case 1: launch_mmvq_r<QT, 1>(a, R, nwg, lds); break;
        case 2: launch_mmvq_r<QT, 2>(a, R, nwg, lds); break;
        case 4: launch_mmvq_r<QT, 4>(a, R, nwg, lds); break;
        case 8: launch_mmvq_r<QT, 8>(a, R, nwg, lds); break;
        default: die("mmvq: bad ncols %d", ncols);
    }
}
void launch_mmvq(int qt, const MmvqArgs &a, int ncols, int R, int nwg, size_t lds) {
    switch (qt) {
        case QT_Q4_0: launch_mmvq_c<QT_Q4_0>(a, ncols, R, nwg, lds); break;
        case QT_Q4_1: launch_mmvq_c<QT_Q4_1>(a, ncols, R, nwg, lds); break;
        case QT_Q5_0: launch_mmvq_c<QT_Q5_0>(a, ncols, R, nwg, lds); break;
        case QT_Q5_1: launch_mmvq_c<QT_Q5_1>(a, ncols, R, nwg, lds); break;
        case QT_Q8_0: launch_mmvq_c<QT_Q8_0>(a, ncols, R, nwg, lds); break;
        default: die("mmvq: bad weight type");
    }
    HIP_CHECK(hipGetLastError());
}

const _Float16 *quantize_activation_f16(const ggml_tensor *src1, bool f16_d) {
    BK_ASSERT(src1->type == GGML_TYPE_F32 && src1->nb[0] == 4 && src1->ne[2] == 1 && src1->ne[3] == 1);
    const int64_t K = src1->ne[0], N = src1->ne[1], nb = K / 32;
    if (g_xf16.valid && g_xf16.src_data == src1->data && g_xf16.f16_d == f16_d && g_xf16.nb == nb && g_xf16.ncols == N)
        return g_xf16.x;
    _Float16 *out = (_Float16 *)ws_alloc((size_t)N * K * 2);
    Timed tm(GGML_HIP_KCLASS_OTHER, (double)(K * N * 6));
    const int64_t threads = nb * N * 32;
    if (f16_d)
        hipLaunchKernelGGL(k_quant_act_f16<true>, grid1(threads), dim3(256), 0, g.stream, dev_ptr(src1),
                           (int64_t)src1->nb[1], nb, N, out);
    else
        hipLaunchKernelGGL(k_quant_act_f16<false>, grid1(threads), dim3(256), 0, g.stream, dev_ptr(src1),
                           (int64_t)src1->nb[1], nb, N, out);
    HIP_CHECK(hipGetLastError());
    g_xf16.valid = true;
    g_xf16.src_data = src1->data;
    g_xf16.src_bytes = ggml_nbytes(src1);
    g_xf16.f16_d = f16_d;
    g_xf16.nb = nb;
    g_xf16.ncols = N;
    g_xf16.x = out;
    return out;
}

void quantize_activation_q8p(const ggml_tensor *src1, bool f16_d, const int8_t **q8, const _Float16 **dx) {
    BK_ASSERT(src1->type == GGML_TYPE_F32 && src1->nb[0] == 4 && src1->ne[2] == 1 && src1->ne[3] == 1);
    const int64_t K = src1->ne[0], N = src1->ne[1], nb = K / 32;
    if (!(g_xq8.valid && g_xq8.src_data == src1->data && g_xq8.f16_d == f16_d && g_xq8.nb == nb && g_xq8.ncols == N)) {
        int8_t *o8 = (int8_t *)ws_alloc((size_t)N * K);
        _Float16 *od = (_Float16 *)ws_alloc((size_t)N * nb * 2);
        Timed tm(GGML_HIP_KCLASS_OTHER, (double)(K * N * 5));
        const int64_t threads = nb * N * 32;
        if (f16_d)
            hipLaunchKernelGGL(k_quant_act_q8p<true>, grid1(threads), dim3(256), 0, g.stream, dev_ptr(src1),
                               (int64_t)src1->nb[1], nb, N, o8, od);
        else
            hipLaunchKernelGGL(k_quant_act_q8p<false>, grid1(threads), dim3(256), 0, g.stream, dev_ptr(src1),
                               (int64_t)src1->nb[1], nb, N, o8, od);
        HIP_CHECK(hipGetLastError());
        g_xq8.valid = true;
        g_xq8.src_data = src1->data;
        g_xq8.src_bytes = ggml_nbytes(src1);
        g_xq8.f16_d = f16_d;
        g_xq8.nb = nb;
        g_xq8.ncols = N;
        g_xq8.q8 = o8;
        g_xq8.dx = od;
    }
    *q8 = g_xq8.q8;
    *dx = g_xq8.dx;
}

struct XI8Buf {  // int8 activations + f32 block scale + zero-point term of the integer prompt GEMM (kernels/mmq_i8.h)
    const void *src_data = nullptr;
    size_t src_bytes = 0;
    int qt = -1;
    int64_t nb = 0, ncols = 0;
    const int8_t *q8 = nullptr;
    const float *dx = nullptr, *xs = nullptr;
    bool valid = false;
} g_xi8_[GGML_HIP_MAX_BACKENDS];
#define g_xi8 (g_xi8_[g.slot])
void quantize_activation_i8(const ggml_tensor *src1, int qt, const int8_t **q8, const float **dx, const float **xs) {
    BK_ASSERT(src1->type == GGML_TYPE_F32 && src1->nb[0] == 4 && src1->ne[2] == 1 && src1->ne[3] == 1);
    const int64_t K = src1->ne[0], N = src1->ne[1], nb = K / 32;
    const bool f16_d = qt == QT_Q4_0 || qt == QT_Q5_0 || qt == QT_Q8_0;
    const float zp = qt == QT_Q4_0 ? 8.0f : qt == QT_Q5_0 ? 16.0f : 0.0f;
    // Q4_0 and Q5_0 differ only in the zero point, Q4_1 and Q5_1 not at all: the cache is keyed on what was produced
    const int key = f16_d ? (int)zp : 1000;
    if (!(g_xi8.valid && g_xi8.src_data == src1->data && g_xi8.qt == key && g_xi8.nb == nb && g_xi8.ncols == N)) {
        int8_t *o8 = (int8_t *)ws_alloc((size_t)N * K);
        float *od = (float *)ws_alloc((size_t)N * nb * 4);
        float *os = (float *)ws_alloc((size_t)N * nb * 4);
        Timed tm(GGML_HIP_KCLASS_OTHER, (double)(K * N * 5));
        const int64_t threads = nb * N * 32;
        if (f16_d)
            hipLaunchKernelGGL(k_quant_act_i8<true>, grid1(threads), dim3(256), 0, g.stream, dev_ptr(src1), (int64_t)src1->nb[1], nb,
                               N, zp, o8, od, os);
        else
            hipLaunchKernelGGL(k_quant_act_i8<false>, grid1(threads), dim3(256), 0, g.stream, dev_ptr(src1), (int64_t)src1->nb[1], nb,
                               N, zp, o8, od, os);
        HIP_CHECK(hipGetLastError());
        g_xi8.valid = true;
        g_xi8.src_data = src1->data;
        g_xi8.src_bytes = ggml_nbytes(src1);
        g_xi8.qt = key;
        g_xi8.nb = nb;
        g_xi8.ncols = N;
        g_xi8.q8 = o8;
        g_xi8.dx = od;
        g_xi8.xs = os;
    }
    *q8 = g_xi8.q8;
    *dx = g_xi8.dx;
    *xs = g_xi8.xs;
}
template <int QT>
void launch_mmq_i8(const MmqI8Args &a, dim3 grid) {
    static DevOnce attr;
    if (attr.first()) {
        HIP_CHECK(hipFuncSetAttribute((const void *)k_mmq_i8<QT>, hipFuncAttributeMaxDynamicSharedMemorySize, I8_LDS));
    }
    g.stat_mmq[Backend::MMQ_K_I8]++;
    hipLaunchKernelGGL(k_mmq_i8<QT>, grid, dim3(256), I8_LDS, g.stream, a);
}

// ---- K-quant mat-vec (kernels/kquant.h): Q8_K activations (ggml's vec_dot_type of every K-quant), one launch per
// chunk of up to 8 columns.  Prompt batches take the same kernel (weights streamed once per 8 tokens): there is no
// K-quant GEMM yet.
struct XKBuf {
    const void *src_data = nullptr;
    size_t src_bytes = 0;
    int64_t nsb = 0, ncols = 0;
    KAct act{};
    bool valid = false;
} g_xk_[GGML_HIP_MAX_BACKENDS];
#define g_xk (g_xk_[g.slot])
KAct quantize_activation_k(const ggml_tensor *src1) {
    BK_ASSERT(src1->type == GGML_TYPE_F32 && src1->nb[0] == 4 && src1->ne[2] == 1 && src1->ne[3] == 1);
    const int64_t K = src1->ne[0], N = src1->ne[1], nsb = K / 256;
    if (!(g_xk.valid && g_xk.src_data == src1->data && g_xk.nsb == nsb && g_xk.ncols == N)) {
        int8_t *q8 = (int8_t *)ws_alloc((size_t)N * K);
        float *d8 = (float *)ws_alloc((size_t)N * nsb * 4);
        int16_t *bs = (int16_t *)ws_alloc((size_t)N * nsb * 32);
        Timed tm(GGML_HIP_KCLASS_OTHER, (double)(K * N * 5));
        hipLaunchKernelGGL(k_quant_q8k, dim3((unsigned)nsb, (unsigned)N), dim3(256), 0, g.stream, dev_ptr(src1),
                           (int64_t)src1->nb[1], nsb, q8, d8, bs);
        HIP_CHECK(hipGetLastError());
        g_xk.valid = true;
        g_xk.src_data = src1->data;
        g_xk.src_bytes = ggml_nbytes(src1);
        g_xk.nsb = nsb;
        g_xk.ncols = N;
        g_xk.act = KAct{q8, d8, bs};
    }
    return g_xk.act;
}
template <int KT, int NCOLS>
void launch_mmvq_k(const MmvqKArgs &a, int nwg, size_t lds) {
    static DevOnce opted;  // more than 64 KB of dynamic LDS needs the attribute, once per device and instantiation
    if constexpr (KT == KT_Q4_K || KT == KT_Q6_K) {
        if (lds > 64 * 1024 && opted.first())
            HIP_CHECK(hipFuncSetAttribute((const void *)k_mmvq_k<KT, NCOLS>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
        hipLaunchKernelGGL((k_mmvq_k<KT, NCOLS>), dim3(nwg), dim3(256), lds, g.stream, a);
    } else {
        if (lds > 64 * 1024 && opted.first())
            HIP_CHECK(hipFuncSetAttribute((const void *)k_mmvq_k2<KT, NCOLS>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
        hipLaunchKernelGGL((k_mmvq_k2<KT, NCOLS>), dim3(nwg), dim3(256), lds, g.stream, a);
    }
}
template <int KT>
void launch_mmvq_k_c(const MmvqKArgs &a, int ncols, int nwg, size_t lds) {
    switch (ncols) {
        case 1: launch_mmvq_k<KT, 1>(a, nwg, lds); break;
        case 2: launch_mmvq_k<KT, 2>(a, nwg, lds); break;
        case 4: launch_mmvq_k<KT, 4>(a, nwg, lds); break;
        default: launch_mmvq_k<KT, 8>(a, nwg, lds); break;
    }
}
bool mul_mat_k_gemm(const ggml_tensor *src0, const ggml_tensor *src1, ggml_tensor *dst);
void mul_mat_k(const ggml_tensor *src0, const ggml_tensor *src1, ggml_tensor *dst) {
    const int kt = kt_of(src0->type);
    const int64_t K = src1->ne[0], N = src1->ne[1], nsb = K / 256;
    BK_ASSERT(K % 256 == 0 && src0->ne[0] == K && dst->type == GGML_TYPE_F32 && dst->nb[0] == 4);
    if (g.opt_mmq_min > 0 && N >= g.opt_mmq_min && mul_mat_k_gemm(src0, src1, dst)) return;  // prompt batch: f16 GEMM
    const KWeight w = kweight_of(src0);
    const KAct act = quantize_activation_k(src1);
    const size_t col_lds = (size_t)K + (size_t)nsb * (4 + 64);
    if (col_lds > 150 * 1024) die("mul_mat: K=%lld too large for the LDS-staged K-quant mat-vec", (long long)K);
    const double sb_bytes = k_block_bytes(kt);
    int64_t c0 = 0;
    while (c0 < N) {
        int ncols = 8;
        while (ncols > 1 && (ncols > N - c0 || (size_t)ncols * col_lds > 150 * 1024)) ncols >>= 1;
        MmvqKArgs a;
        a.w = w;
        a.x.q8 = act.q8 + c0 * K;
        a.x.d8 = act.d8 + c0 * nsb;
        a.x.bs = act.bs + c0 * nsb * 16;
        a.dst = (float *)(dev_ptr(dst) + c0 * dst->nb[1]);
        a.ldd = (int64_t)dst->nb[1] / 4;
        const size_t lds = (size_t)ncols * col_lds;
        // enough workgroups to fill the CUs' wave slots at this LDS footprint, never more than one row per wave
        const int per_cu = (int)std::max<size_t>(1, std::min<size_t>(8, (150 * 1024) / std::max<size_t>(lds, 1)));
        const int nwg = (int)std::min<int64_t>((w.M + 3) / 4, (int64_t)g.num_cus * per_cu);
        Timed tm(GGML_HIP_KCLASS_MMVQ, (double)w.M * nsb * sb_bytes + (double)w.M * ncols * 4 + (double)lds);
        switch (kt) {
            case KT_Q4_K: launch_mmvq_k_c<KT_Q4_K>(a, ncols, nwg, lds); break;
            case KT_Q6_K: launch_mmvq_k_c<KT_Q6_K>(a, ncols, nwg, lds); break;
            case KT_Q2_K: launch_mmvq_k_c<KT_Q2_K>(a, ncols, nwg, lds); break;
            case KT_Q3_K: launch_mmvq_k_c<KT_Q3_K>(a, ncols, nwg, lds); break;
            default: launch_mmvq_k_c<KT_Q5_K>(a, ncols, nwg, lds); break;
        }
        HIP_CHECK(hipGetLastError());
        c0 += ncols;
    }
}
// rows of a planar K weight dequantized to f32 (the decoders of get_rows): dst[r * ldd + k]
void dequant_k_rows(const KWeight &w, const int *ids, int64_t rows, float *dst, int64_t ldd) {
    const dim3 grid((unsigned)((w.nsb * (w.kt == KT_Q4_K || w.kt == KT_Q6_K ? 8 : 16) + 255) / 256), (unsigned)rows);
    switch (w.kt) {
        case KT_Q4_K:
        case KT_Q6_K: hipLaunchKernelGGL(k_get_rows_k, grid, dim3(256), 0, g.stream, w, ids, dst, ldd); break;
        case KT_Q2_K: hipLaunchKernelGGL(k_get_rows_k2<KT_Q2_K>, grid, dim3(256), 0, g.stream, w, ids, dst, ldd); break;
        case KT_Q3_K: hipLaunchKernelGGL(k_get_rows_k2<KT_Q3_K>, grid, dim3(256), 0, g.stream, w, ids, dst, ldd); break;
        default: hipLaunchKernelGGL(k_get_rows_k2<KT_Q5_K>, grid, dim3(256), 0, g.stream, w, ids, dst, ldd); break;
    }
    HIP_CHECK(hipGetLastError());
}

// Resident f16 copy of a quantized weight (kernels/mmq_w16.h): created on first use by a prompt batch and kept as a CACHE:
// released with the weight's record, when option mmq_w16 is switched off, and whenever another allocation of the library
// fails (dev_malloc).  Returns false (and the GEMM dequantizes in LDS as before) when the option is off, K / 32 is odd,
// or HBM would be left with less than the headroom (option w16_headroom_gb, default 16 GB) after the allocation.
size_t w16_headroom() { return (size_t)std::max(0, g.opt_w16_headroom_gb) << 30; }
size_t release_w16_copies() {
    size_t freed = 0;
    bool synced = false;
    for (auto *m : {&g.tensors, &g.auto_tensors})
        for (auto &kv : *m) {
            DevTensor *e = kv.second;
            if (!e->w16) continue;
            if (!synced && g.stream) {
                HIP_CHECK(hipStreamSynchronize(g.stream));
                synced = true;
            }
            HIP_CHECK(hipFree(e->w16));
            freed += e->w16_size;
            g.w16_bytes -= e->w16_size;
            e->w16 = nullptr;
            e->w16_size = 0;
            e->qw.w16 = nullptr;
        }
    if (freed) g.w16_gen++;  // prompt plans re-read their weights' w16 pointers at the next launch (llama_plan.inc)
    return freed;
}
bool ensure_w16(DevTensor *e) {
    if (!e || !e->soa || !g.opt_mmq_w16 || e->qw.nb % 2 != 0) return false;
    if (e->w16) return true;
    const size_t bytes = (size_t)e->qw.M * e->qw.nb * 64;
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess || free_b < bytes + w16_headroom()) return false;
    if (hipMalloc((void **)&e->w16, bytes) != hipSuccess) {
        (void)hipGetLastError();
        e->w16 = nullptr;
        return false;
    }
    g.w16_bytes += bytes;
    e->w16_size = bytes;
    const unsigned nblk = (unsigned)((e->qw.M * e->qw.nb + 255) / 256);
    switch (e->qw.qt) {
        case QT_Q4_0: hipLaunchKernelGGL(k_dequant_w16<QT_Q4_0>, dim3(nblk), dim3(256), 0, g.stream, e->qw, (_Float16 *)e->w16); break;
        case QT_Q4_1: hipLaunchKernelGGL(k_dequant_w16<QT_Q4_1>, dim3(nblk), dim3(256), 0, g.stream, e->qw, (_Float16 *)e->w16); break;
        case QT_Q5_0: hipLaunchKernelGGL(k_dequant_w16<QT_Q5_0>, dim3(nblk), dim3(256), 0, g.stream, e->qw, (_Float16 *)e->w16); break;
        case QT_Q5_1: hipLaunchKernelGGL(k_dequant_w16<QT_Q5_1>, dim3(nblk), dim3(256), 0, g.stream, e->qw, (_Float16 *)e->w16); break;
        case QT_Q8_0: hipLaunchKernelGGL(k_dequant_w16<QT_Q8_0>, dim3(nblk), dim3(256), 0, g.stream, e->qw, (_Float16 *)e->w16); break;
        default: die("w16: bad weight type");
    }
    HIP_CHECK(hipGetLastError());
    e->qw.w16 = e->w16;
    return true;
}

// The same for a planar K-quant weight (kernels/kquant2.h): its rows dequantized by the get_rows decoders (f32, 8192 rows
// at a time through a temporary), rounded to f16 in the GEMM's k order.
bool ensure_w16_k(DevTensor *e) {
    if (!e || !e->ksoa || !g.opt_mmq_w16) return false;
    if (e->w16) return true;
    const int64_t M = e->kw.M, K = e->kw.nsb * 256, chunk = std::min<int64_t>(M, 8192);
    const size_t bytes = (size_t)M * K * 2, tmp_bytes = (size_t)chunk * K * 4;
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess || free_b < bytes + tmp_bytes + w16_headroom()) return false;
    float *tmp = nullptr;
    if (hipMalloc((void **)&e->w16, bytes) != hipSuccess || hipMalloc((void **)&tmp, tmp_bytes) != hipSuccess) {
        (void)hipGetLastError();
        if (e->w16) HIP_CHECK(hipFree(e->w16));
        e->w16 = nullptr;
        return false;
    }
    for (int64_t r0 = 0; r0 < M; r0 += chunk) {
        const int64_t n = std::min(chunk, M - r0);
        dequant_k_rows(kw_rows(e->kw, r0, n), nullptr, n, tmp, K);
        hipLaunchKernelGGL(k_f32_to_w16, grid1(n * K), dim3(256), 0, g.stream, tmp, n * K / 32, (_Float16 *)e->w16 + r0 * K);
        HIP_CHECK(hipGetLastError());
    }
    HIP_CHECK(hipStreamSynchronize(g.stream));
    HIP_CHECK(hipFree(tmp));
    g.w16_bytes += bytes;
    e->w16_size = bytes;
    return true;
}

// The default prompt GEMM launch (f16 matrix cores, k_mmq_dma / k_mmq): x16 (or x8 + dx for option mmq_dma = 2) are the
// activations after the Q8 pre-pass; dst[n * ldd + m].  Shared by the generic executor and the fused prompt plan.
// nseg > 1: up to three matrices with the same K in one launch (MmqArgs: nseg); `splits` 0 = chosen here.
struct MmqSegHost {
    QWeight w;
    float *dst;
    int64_t ldd;
};
int mmq_auto_splits(int tiles, int64_t nb, bool dst_contig) {
    // too few tiles to fill the chip (E x E at 512 tokens: 128 tiles for 256 CUs): split K in two, combined with
    // commutative (2-addend) f32 atomic adds into a zeroed dst
    const int nstage = (int)((nb + 1) / 2);
    return (g.opt_mmq_splitk && tiles * 4 <= g.num_cus * 3 && nstage >= 16 && dst_contig) ? 2 : 1;
}
// Whether a prompt GEMM launch runs on the 256 x 256 kernel (kernels/mmq_w16_256.h): it needs every weight's resident f16
// copy, at least 160 tokens (62 % of a token tile) and enough (tile x split) items to keep >= 60 % of the CUs busy in every
// round.  The K split itself is NOT chosen here: it follows the 128-tile rule (mmq_auto_splits, per matrix) whatever the
// kernel, so that a fused launch of the prompt plan and the per-matrix launches of the node-by-node executor add the same
// partial sums (the kernels are bit-identical for equal splits).
bool mmq_use_t256(int nseg, const MmqSegHost *segs, int64_t N, int64_t nb, int splits) {
    if (!g.opt_mmq_t256 || !g.opt_mmq_w16 || !g.opt_mmq_persist || g.opt_mmq_dma != 1 || nb % 2 != 0) return false;
    int tiles = 0;
    for (int i = 0; i < nseg; i++) {
        if (!segs[i].w.w16) return false;
        tiles += (int)((segs[i].w.M + T256_TM - 1) / T256_TM);
    }
    if (g.opt_mmq_t256 == 2) return true;  // tests: every launch that can run on it does
    if (N < 160) return false;
    tiles *= (int)((N + T256_TN - 1) / T256_TN);
    const int items = tiles * splits, rounds = (items + g.num_cus - 1) / g.num_cus;
    return items * 5 >= rounds * g.num_cus * 3;
}
void mmq_w16_256_launch(int nseg, const MmqSegHost *segs, const _Float16 *x16, int64_t N, int64_t nb, int splits, bool zero_dst,
                        int64_t split_stride) {
    MmqArgs a;
    memset(&a, 0, sizeof(a));
    a.w = segs[0].w;
    a.x = x16;
    a.dst = segs[0].dst;
    a.ldd = segs[0].ldd;
    a.M = a.w.M;
    a.N = N;
    a.nb = nb;
    a.nseg = nseg;
    int tiles_m = 0;
    double rows = 0;
    for (int i = 0; i < nseg; i++) {
        tiles_m += (int)((segs[i].w.M + T256_TM - 1) / T256_TM);
        rows += (double)segs[i].w.M;
        if (i < 2) a.tile_end[i] = tiles_m;
    }
    if (nseg > 1) { a.wb = segs[1].w; a.dst_b = segs[1].dst; a.ldd_b = segs[1].ldd; }
    if (nseg > 2) { a.wc = segs[2].w; a.dst_c = segs[2].dst; a.ldd_c = segs[2].ldd; }
    a.tiles_n = (int)((N + T256_TN - 1) / T256_TN);
    a.split_stride = splits > 1 ? split_stride : 0;
    if (splits > 1 && zero_dst && !split_stride)
        for (int i = 0; i < nseg; i++) HIP_CHECK(hipMemsetAsync(segs[i].dst, 0, (size_t)segs[i].w.M * N * 4, g.stream));
    static DevOnce attr_set;
    if (attr_set.first()) {
        HIP_CHECK(hipFuncSetAttribute((const void *)k_mmq_w16_256<0>, hipFuncAttributeMaxDynamicSharedMemorySize, T256_LDS));
        HIP_CHECK(hipFuncSetAttribute((const void *)k_mmq_w16_256<1>, hipFuncAttributeMaxDynamicSharedMemorySize, T256_LDS));
        HIP_CHECK(hipFuncSetAttribute((const void *)k_mmq_w16_256<2>, hipFuncAttributeMaxDynamicSharedMemorySize, T256_LDS));
        HIP_CHECK(hipFuncSetAttribute((const void *)k_mmq_w16_256<3>, hipFuncAttributeMaxDynamicSharedMemorySize, T256_LDS));
    }
    const int tiles_total = tiles_m * a.tiles_n, n_items = tiles_total * splits;
    Timed tm(GGML_HIP_KCLASS_MMQ_MFMA, 2.0 * rows * (double)N * (double)(nb * 32));
    g.stat_mmq[Backend::MMQ_K_W16_256]++;
    const dim3 grid((unsigned)std::min(n_items, g.num_cus));
    switch (g.opt_mmq_t256_var & 3) {
        case 1: hipLaunchKernelGGL(k_mmq_w16_256<1>, grid, dim3(512), T256_LDS, g.stream, a, n_items, tiles_total, splits); break;
        case 2: hipLaunchKernelGGL(k_mmq_w16_256<2>, grid, dim3(512), T256_LDS, g.stream, a, n_items, tiles_total, splits); break;
        case 3: hipLaunchKernelGGL(k_mmq_w16_256<3>, grid, dim3(512), T256_LDS, g.stream, a, n_items, tiles_total, splits); break;
        default: hipLaunchKernelGGL(k_mmq_w16_256<0>, grid, dim3(512), T256_LDS, g.stream, a, n_items, tiles_total, splits); break;
    }
    HIP_CHECK(hipGetLastError());
}
void mmq_f16_launch_multi(int qt, int nseg, const MmqSegHost *segs, const _Float16 *x16, const int8_t *x8, const _Float16 *dx,
                          int64_t N, int64_t nb, bool dst_contig, int splits, bool zero_dst, int64_t split_stride = 0) {
    if (splits <= 0) {
        int t128 = 0;
        for (int i = 0; i < nseg; i++) t128 += (int)((segs[i].w.M + MMQ_TM - 1) / MMQ_TM);
        splits = mmq_auto_splits(t128 * (int)((N + MMQ_TN - 1) / MMQ_TN), nb, dst_contig);
        if (g.opt_mmq_splits > 0 && dst_contig && nb / 2 >= 2 * g.opt_mmq_splits) splits = std::min(g.opt_mmq_splits, 2);  // probes
    }
    if (x16 && mmq_use_t256(nseg, segs, N, nb, splits)) {
        mmq_w16_256_launch(nseg, segs, x16, N, nb, splits, zero_dst, split_stride);
        return;
    }
    MmqArgs a;
    memset(&a, 0, sizeof(a));
    a.w = segs[0].w;
    const bool use_dma = g.opt_mmq_dma && nb % 2 == 0, use_x8 = use_dma && g.opt_mmq_dma >= 2;
    a.x = x16;
    a.x8 = x8;
    a.dx = dx;
    a.dst = segs[0].dst;
    a.ldd = segs[0].ldd;
    a.M = a.w.M;
    a.N = N;
    a.nb = nb;
    a.nseg = nseg;
    int tiles_m = 0;
    double rows = 0;
    for (int i = 0; i < nseg; i++) {
        tiles_m += (int)((segs[i].w.M + MMQ_TM - 1) / MMQ_TM);
        rows += (double)segs[i].w.M;
        if (i < 2) a.tile_end[i] = tiles_m;
    }
    if (nseg > 1) { a.wb = segs[1].w; a.dst_b = segs[1].dst; a.ldd_b = segs[1].ldd; }
    if (nseg > 2) { a.wc = segs[2].w; a.dst_c = segs[2].dst; a.ldd_c = segs[2].ldd; }
    a.tiles_n = (int)((N + MMQ_TN - 1) / MMQ_TN);
    if (splits <= 0) splits = mmq_auto_splits(tiles_m * a.tiles_n, nb, dst_contig);
    a.split_stride = splits > 1 ? split_stride : 0;
    if (splits > 1 && zero_dst && !split_stride)
        for (int i = 0; i < nseg; i++) HIP_CHECK(hipMemsetAsync(segs[i].dst, 0, (size_t)segs[i].w.M * N * 4, g.stream));
    a.xcd_by_n = g.opt_mmq_xcdn == 2 ? 1 : g.opt_mmq_xcdn && (a.tiles_n == 1 || a.tiles_n == 2 || a.tiles_n == 4 || a.tiles_n == 8) && tiles_m % (8 / a.tiles_n) == 0;
    const dim3 grid((unsigned)(tiles_m * a.tiles_n), (unsigned)splits);
    static DevOnce lds_attr_set;
    if (lds_attr_set.first()) {  // 73.7 KB of dynamic LDS: above the 64 KB a kernel gets without opting in
        HIP_CHECK(hipFuncSetAttribute((const void *)k_mmq<QT_Q4_0>, hipFuncAttributeMaxDynamicSharedMemorySize, MMQ_LDS));
        HIP_CHECK(hipFuncSetAttribute((const void *)k_mmq<QT_Q4_1>, hipFuncAttributeMaxDynamicSharedMemorySize, MMQ_LDS));
        HIP_CHECK(hipFuncSetAttribute((const void *)k_mmq<QT_Q5_0>, hipFuncAttributeMaxDynamicSharedMemorySize, MMQ_LDS));
        HIP_CHECK(hipFuncSetAttribute((const void *)k_mmq<QT_Q5_1>, hipFuncAttributeMaxDynamicSharedMemorySize, MMQ_LDS));
        HIP_CHECK(hipFuncSetAttribute((const void *)k_mmq<QT_Q8_0>, hipFuncAttributeMaxDynamicSharedMemorySize, MMQ_LDS));
    }
    Timed tm(GGML_HIP_KCLASS_MMQ_MFMA, 2.0 * rows * (double)N * (double)(nb * 32));
    if (use_dma && !use_x8 && g.opt_mmq_persist) {  // one workgroup per CU walks the tiles (kernels/mmq_dmap.h)
        static DevOnce p_attr_set;
        if (p_attr_set.first()) {
            HIP_CHECK(hipFuncSetAttribute((const void *)k_mmq_dma_p<QT_Q4_0>, hipFuncAttributeMaxDynamicSharedMemorySize, DMA_LDS));
            HIP_CHECK(hipFuncSetAttribute((const void *)k_mmq_dma_p<QT_Q4_1>, hipFuncAttributeMaxDynamicSharedMemorySize, DMA_LDS));
            HIP_CHECK(hipFuncSetAttribute((const void *)k_mmq_dma_p<QT_Q5_0>, hipFuncAttributeMaxDynamicSharedMemorySize, DMA_LDS));
            HIP_CHECK(hipFuncSetAttribute((const void *)k_mmq_dma_p<QT_Q5_1>, hipFuncAttributeMaxDynamicSharedMemorySize, DMA_LDS));
            HIP_CHECK(hipFuncSetAttribute((const void *)k_mmq_dma_p<QT_Q8_0>, hipFuncAttributeMaxDynamicSharedMemorySize, DMA_LDS));
        }
        const int tiles_total = tiles_m * a.tiles_n, n_items = tiles_total * splits;
        const dim3 pgrid((unsigned)std::min(n_items, g.num_cus));
        bool all_w16 = g.opt_mmq_w16 != 0;
        for (int i = 0; i < nseg; i++) all_w16 = all_w16 && segs[i].w.w16 != nullptr;
        if (all_w16) {  // both operands by DMA from resident f16 copies (kernels/mmq_w16.h)
            static DevOnce w16_attr_set;
            if (w16_attr_set.first()) {
                HIP_CHECK(hipFuncSetAttribute((const void *)k_mmq_w16_p8, hipFuncAttributeMaxDynamicSharedMemorySize, W16_LDS));
            }
            g.stat_mmq[Backend::MMQ_K_W16_P8]++;
            hipLaunchKernelGGL(k_mmq_w16_p8, pgrid, dim3(512), W16_LDS, g.stream, a, n_items, tiles_total, splits);
            HIP_CHECK(hipGetLastError());
            return;
        }
        if (g.opt_mmq_waves == 8) {  // two waves per SIMD on the same tile (kernels/mmq_dmap8.h)
            static DevOnce p8_attr_set;
            if (p8_attr_set.first()) {
                HIP_CHECK(hipFuncSetAttribute((const void *)k_mmq_dma_p8<QT_Q4_0>, hipFuncAttributeMaxDynamicSharedMemorySize, Dma8<QT_Q4_0>::LDS));
                HIP_CHECK(hipFuncSetAttribute((const void *)k_mmq_dma_p8<QT_Q4_1>, hipFuncAttributeMaxDynamicSharedMemorySize, Dma8<QT_Q4_1>::LDS));
                HIP_CHECK(hipFuncSetAttribute((const void *)k_mmq_dma_p8<QT_Q5_0>, hipFuncAttributeMaxDynamicSharedMemorySize, Dma8<QT_Q5_0>::LDS));
                HIP_CHECK(hipFuncSetAttribute((const void *)k_mmq_dma_p8<QT_Q5_1>, hipFuncAttributeMaxDynamicSharedMemorySize, Dma8<QT_Q5_1>::LDS));
                HIP_CHECK(hipFuncSetAttribute((const void *)k_mmq_dma_p8<QT_Q8_0>, hipFuncAttributeMaxDynamicSharedMemorySize, Dma8<QT_Q8_0>::LDS));
            }
            g.stat_mmq[Backend::MMQ_K_DMA_P8]++;
            switch (qt) {
                case QT_Q4_0: hipLaunchKernelGGL(k_mmq_dma_p8<QT_Q4_0>, pgrid, dim3(512), Dma8<QT_Q4_0>::LDS, g.stream, a, n_items, tiles_total, splits); break;
                case QT_Q4_1: hipLaunchKernelGGL(k_mmq_dma_p8<QT_Q4_1>, pgrid, dim3(512), Dma8<QT_Q4_1>::LDS, g.stream, a, n_items, tiles_total, splits); break;
                case QT_Q5_0: hipLaunchKernelGGL(k_mmq_dma_p8<QT_Q5_0>, pgrid, dim3(512), Dma8<QT_Q5_0>::LDS, g.stream, a, n_items, tiles_total, splits); break;
                case QT_Q5_1: hipLaunchKernelGGL(k_mmq_dma_p8<QT_Q5_1>, pgrid, dim3(512), Dma8<QT_Q5_1>::LDS, g.stream, a, n_items, tiles_total, splits); break;
                case QT_Q8_0: hipLaunchKernelGGL(k_mmq_dma_p8<QT_Q8_0>, pgrid, dim3(512), Dma8<QT_Q8_0>::LDS, g.stream, a, n_items, tiles_total, splits); break;
                default: die("mmq: bad weight type");
            }
            HIP_CHECK(hipGetLastError());
            return;
        }
        g.stat_mmq[Backend::MMQ_K_DMA_P]++;
        switch (qt) {
            case QT_Q4_0: hipLaunchKernelGGL(k_mmq_dma_p<QT_Q4_0>, pgrid, dim3(256), DMA_LDS, g.stream, a, n_items, tiles_total, splits); break;
            case QT_Q4_1: hipLaunchKernelGGL(k_mmq_dma_p<QT_Q4_1>, pgrid, dim3(256), DMA_LDS, g.stream, a, n_items, tiles_total, splits); break;
            case QT_Q5_0: hipLaunchKernelGGL(k_mmq_dma_p<QT_Q5_0>, pgrid, dim3(256), DMA_LDS, g.stream, a, n_items, tiles_total, splits); break;
            case QT_Q5_1: hipLaunchKernelGGL(k_mmq_dma_p<QT_Q5_1>, pgrid, dim3(256), DMA_LDS, g.stream, a, n_items, tiles_total, splits); break;
            case QT_Q8_0: hipLaunchKernelGGL(k_mmq_dma_p<QT_Q8_0>, pgrid, dim3(256), DMA_LDS, g.stream, a, n_items, tiles_total, splits); break;
            default: die("mmq: bad weight type");
        }
        HIP_CHECK(hipGetLastError());
        return;
    }
    if (use_dma) {
        static DevOnce dma_attr_set;
        if (dma_attr_set.first()) {
            HIP_CHECK(hipFuncSetAttribute((const void *)k_mmq_dma<QT_Q4_0, false>, hipFuncAttributeMaxDynamicSharedMemorySize, DMA_LDS));
            HIP_CHECK(hipFuncSetAttribute((const void *)k_mmq_dma<QT_Q4_1, false>, hipFuncAttributeMaxDynamicSharedMemorySize, DMA_LDS));
            HIP_CHECK(hipFuncSetAttribute((const void *)k_mmq_dma<QT_Q5_0, false>, hipFuncAttributeMaxDynamicSharedMemorySize, DMA_LDS));
            HIP_CHECK(hipFuncSetAttribute((const void *)k_mmq_dma<QT_Q5_1, false>, hipFuncAttributeMaxDynamicSharedMemorySize, DMA_LDS));
            HIP_CHECK(hipFuncSetAttribute((const void *)k_mmq_dma<QT_Q8_0, false>, hipFuncAttributeMaxDynamicSharedMemorySize, DMA_LDS));
            HIP_CHECK(hipFuncSetAttribute((const void *)k_mmq_dma<QT_Q4_0, true>, hipFuncAttributeMaxDynamicSharedMemorySize, D8_LDS));
            HIP_CHECK(hipFuncSetAttribute((const void *)k_mmq_dma<QT_Q4_1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, D8_LDS));
            HIP_CHECK(hipFuncSetAttribute((const void *)k_mmq_dma<QT_Q5_0, true>, hipFuncAttributeMaxDynamicSharedMemorySize, D8_LDS));
            HIP_CHECK(hipFuncSetAttribute((const void *)k_mmq_dma<QT_Q5_1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, D8_LDS));
            HIP_CHECK(hipFuncSetAttribute((const void *)k_mmq_dma<QT_Q8_0, true>, hipFuncAttributeMaxDynamicSharedMemorySize, D8_LDS));
        }
        g.stat_mmq[Backend::MMQ_K_DMA]++;
#define LAUNCH_DMA(QT_)                                                                                             \
    if (use_x8)                                                                                                     \
        hipLaunchKernelGGL((k_mmq_dma<QT_, true>), grid, dim3(256), D8_LDS, g.stream, a);                           \
    else                                                                                                            \
        hipLaunchKernelGGL((k_mmq_dma<QT_, false>), grid, dim3(256), DMA_LDS, g.stream, a);
        switch (qt) {
            case QT_Q4_0: LAUNCH_DMA(QT_Q4_0) break;
            case QT_Q4_1: LAUNCH_DMA(QT_Q4_1) break;
            case QT_Q5_0: LAUNCH_DMA(QT_Q5_0) break;
            case QT_Q5_1: LAUNCH_DMA(QT_Q5_1) break;
            case QT_Q8_0: LAUNCH_DMA(QT_Q8_0) break;
            default: die("mmq: bad weight type");
        }
#undef LAUNCH_DMA
        HIP_CHECK(hipGetLastError());
        return;
    }
    g.stat_mmq[Backend::MMQ_K_PLAIN]++;
    switch (qt) {
        case QT_Q4_0: hipLaunchKernelGGL(k_mmq<QT_Q4_0>, grid, dim3(256), MMQ_LDS, g.stream, a); break;
        case QT_Q4_1: hipLaunchKernelGGL(k_mmq<QT_Q4_1>, grid, dim3(256), MMQ_LDS, g.stream, a); break;
        case QT_Q5_0: hipLaunchKernelGGL(k_mmq<QT_Q5_0>, grid, dim3(256), MMQ_LDS, g.stream, a); break;
        case QT_Q5_1: hipLaunchKernelGGL(k_mmq<QT_Q5_1>, grid, dim3(256), MMQ_LDS, g.stream, a); break;
        case QT_Q8_0: hipLaunchKernelGGL(k_mmq<QT_Q8_0>, grid, dim3(256), MMQ_LDS, g.stream, a); break;
        default: die("mmq: bad weight type");
    }
    HIP_CHECK(hipGetLastError());
}

void mmq_f16_launch(int qt, const QWeight &w, const _Float16 *x16, const int8_t *x8, const _Float16 *dx, float *dst,
                    int64_t ldd, int64_t N, int64_t nb, bool dst_contig) {
    const MmqSegHost seg{w, dst, ldd};
    mmq_f16_launch_multi(qt, 1, &seg, x16, x8, dx, N, nb, dst_contig, 0, true);
}

// Quantized GEMM on the f16 matrix cores (kernels/mmq.h); the `algo_bytes` slot of the MMQ_MFMA timing class
// carries FLOPs (2*M*N*K), the unit that class is bounded by.
void mul_mat_q_mfma(const ggml_tensor *src0, const ggml_tensor *src1, ggml_tensor *dst) {
    const int qt = qt_of(src0->type);
    const int64_t K = src1->ne[0], N = src1->ne[1], nb = K / 32;
    const bool f16_d = qt == QT_Q4_0 || qt == QT_Q5_0 || qt == QT_Q8_0;
    if (g.opt_mmq_i8 && nb % 2 == 0) {  // integer matrix cores: ggml's exact block dots (kernels/mmq_i8.h)
        MmqI8Args ia;
        ia.w = qweight_of(src0);
        quantize_activation_i8(src1, qt, &ia.x8, &ia.dx, &ia.xs);
        ia.dst = (float *)dev_ptr(dst);
        ia.ldd = (int64_t)dst->nb[1] / 4;
        ia.M = ia.w.M;
        ia.N = N;
        ia.nb = nb;
        const int tiles_m = (int)((ia.M + MMQ_TM - 1) / MMQ_TM);
        ia.tiles_n = (int)((N + MMQ_TN - 1) / MMQ_TN);
        const int nstage = (int)(nb / 2);
        const int splits = (g.opt_mmq_splitk && tiles_m * ia.tiles_n * 2 <= g.num_cus * 3 && nstage >= 16 && dst->nb[0] == 4 &&
                            ggml_is_contiguous(dst)) ? 2 : 1;  // two workgroups per CU: fewer than 1.5 tiles per slot -> split K
        if (splits > 1) HIP_CHECK(hipMemsetAsync(ia.dst, 0, (size_t)ia.M * N * 4, g.stream));
        const dim3 grid((unsigned)(tiles_m * ia.tiles_n), (unsigned)splits);
        Timed tm(GGML_HIP_KCLASS_MMQ_MFMA, 2.0 * (double)ia.M * (double)N * (double)K);
        switch (qt) {
            case QT_Q4_0: launch_mmq_i8<QT_Q4_0>(ia, grid); break;
            case QT_Q4_1: launch_mmq_i8<QT_Q4_1>(ia, grid); break;
            case QT_Q5_0: launch_mmq_i8<QT_Q5_0>(ia, grid); break;
            case QT_Q5_1: launch_mmq_i8<QT_Q5_1>(ia, grid); break;
            case QT_Q8_0: launch_mmq_i8<QT_Q8_0>(ia, grid); break;
            default: die("mmq: bad weight type");
        }
        HIP_CHECK(hipGetLastError());
        return;
    }
    const bool use_dma = g.opt_mmq_dma && nb % 2 == 0, use_x8 = use_dma && g.opt_mmq_dma >= 2;
    const _Float16 *x16 = nullptr, *dx = nullptr;
    const int8_t *x8 = nullptr;
    if (use_dma && !use_x8 && g.opt_mmq_persist && N >= W16_MIN_TOKENS) {  // a resident weight meeting a real prompt batch gets its
        DevTensor *e = extra_of(src0);                                      // f16 copy here too (the prompt plan makes them
        if (!e) e = find_tensor((uintptr_t)src0->data);                     // for a whole model at once, llama_plan.inc)
        if (e && e->soa && (uintptr_t)src0->data == e->host) ensure_w16(e);
    }
    if (use_x8)
        quantize_activation_q8p(src1, f16_d, &x8, &dx);
    else
        x16 = quantize_activation_f16(src1, f16_d);
    mmq_f16_launch(qt, qweight_of(src0), x16, x8, dx, (float *)dev_ptr(dst), (int64_t)dst->nb[1] / 4, N, nb,
                   dst->nb[0] == 4 && ggml_is_contiguous(dst));
}

// K-quant prompt batch on the f16 GEMM: the weight's resident f16 copy x the activations after their Q8_K round trip
// (f16(d8 * q): what ggml's K-quant dots see of src1).  Same kernels and tile rules as the other formats; the f16 rounding
// of both operands is the approximation those already make.  false = not applicable here (the caller streams the weight
// through the mat-vec kernel instead): options off, a view / workspace weight, fewer than 64 tokens without a copy yet, no HBM.
bool mul_mat_k_gemm(const ggml_tensor *src0, const ggml_tensor *src1, ggml_tensor *dst) {
    if (!(g.opt_mmq_w16 && g.opt_mmq_persist && g.opt_mmq_dma == 1) || g.opt_mmq_i8) return false;
    if (dst->type != GGML_TYPE_F32 || dst->nb[0] != 4 || src1->ne[2] != 1 || src1->ne[3] != 1) return false;
    DevTensor *e = extra_of(src0);
    if (!e) e = find_tensor((uintptr_t)src0->data);
    if (!(e && e->ksoa && (uintptr_t)src0->data == e->host)) return false;
    const int64_t K = src1->ne[0], N = src1->ne[1], nsb = K / 256;
    if (!e->w16 && N < W16_MIN_TOKENS) return false;
    if (!ensure_w16_k(e)) return false;
    _Float16 *x16 = (_Float16 *)ws_alloc((size_t)N * K * 2);
    {
        Timed tm(GGML_HIP_KCLASS_OTHER, (double)(K * N * 6));
        hipLaunchKernelGGL(k_quant_act_f16_k, dim3((unsigned)nsb, (unsigned)N), dim3(256), 0, g.stream, dev_ptr(src1),
                           (int64_t)src1->nb[1], nsb, x16);
        HIP_CHECK(hipGetLastError());
    }
    QWeight w;
    memset(&w, 0, sizeof(w));
    w.M = e->kw.M;
    w.nb = K / 32;
    w.qt = QT_Q8_0;  // never read: every kernel that takes a resident copy reads only w16
    w.w16 = e->w16;
    mmq_f16_launch(QT_Q8_0, w, x16, nullptr, nullptr, (float *)dev_ptr(dst), (int64_t)dst->nb[1] / 4, N, K / 32,
                   ggml_is_contiguous(dst));
    return true;
}

int pick_rows(int64_t M) {
    if (g.opt_mmvq_rows == 1 || g.opt_mmvq_rows == 2 || g.opt_mmvq_rows == 4) return g.opt_mmvq_rows;
    return M >= 16384 ? 2 : 1;
}

// Quantized mat-vec over up to 3 weight matrices sharing src1 (same type, same K).
void mul_mat_q(int nmat, const ggml_tensor *const *src0s, const ggml_tensor *src1, ggml_tensor *const *dsts) {
    const int qt = qt_of(src0s[0]->type);
    const int64_t K = src1->ne[0], N = src1->ne[1], nb = K / 32;
    BK_ASSERT(K % 32 == 0);
    QWeight ws[3];
    for (int i = 0; i < nmat; i++) {
        BK_ASSERT(src0s[i]->type == src0s[0]->type && src0s[i]->ne[0] == K);
        BK_ASSERT(dsts[i]->type == GGML_TYPE_F32 && dsts[i]->nb[0] == 4);
        ws[i] = qweight_of(src0s[i]);
    }
    if (g.opt_mmq_min > 0 && N >= g.opt_mmq_min) {  // prompt batch: MFMA GEMM
        for (int i = 0; i < nmat; i++) mul_mat_q_mfma(src0s[i], src1, dsts[i]);
        return;
    }
    const bool f16_d = qt == QT_Q4_0 || qt == QT_Q5_0 || qt == QT_Q8_0;
    const QAct act = quantize_activation(src1, f16_d);
    const int64_t max_cols_lds = (int64_t)(64 * 1024) / (nb * 40);
    if (max_cols_lds < 1) die("mul_mat: K=%lld too large for the LDS-staged mat-vec", (long long)K);
    int64_t c0 = 0;
    while (c0 < N) {
        int ncols = 8;
        while (ncols > 1 && (ncols > N - c0 || ncols > max_cols_lds)) ncols >>= 1;
        MmvqArgs a;
        memset(&a, 0, sizeof(a));
        a.nseg = nmat;
        a.nb = nb;
        a.x.lo = act.lo + c0 * nb;
        a.x.hi = act.hi + c0 * nb;
        a.x.d = act.d + c0 * nb;
        a.x.sum = act.sum + c0 * nb;
        int nwg = 0;
        int64_t Mmax = 0;
        for (int i = 0; i < nmat; i++) Mmax = std::max(Mmax, ws[i].M);
        const int R = pick_rows(Mmax);
        double bytes = 0;
        for (int i = 0; i < nmat; i++) {
            a.seg[i].w = ws[i];
            a.seg[i].dst = (float *)(dev_ptr(dsts[i]) + c0 * dsts[i]->nb[1]);
            a.seg[i].ldd = (int64_t)dsts[i]->nb[1] / 4;
            a.seg[i].wg_begin = nwg;
            nwg += (int)((ws[i].M + 4 * R - 1) / (4 * R));
            bytes += (double)ws[i].M * nb * blk_bytes(qt) + (double)ws[i].M * ncols * 4;
        }
        bytes += (double)ncols * nb * 40;
        Timed tm(GGML_HIP_KCLASS_MMVQ, bytes);
        launch_mmvq(qt, a, ncols, R, nwg, (size_t)ncols * nb * 40);
        c0 += ncols;
    }
}

void op_mul_mat(ggml_tensor *dst) {
    const ggml_tensor *a = dst->src[0], *b = dst->src[1];
    if (qt_of(a->type) >= 0) {
        const ggml_tensor *s0[1] = {a};
        ggml_tensor *d[1] = {dst};
        mul_mat_q(1, s0, b, d);
        return;
    }
    if (kt_of(a->type) >= 0) {
        mul_mat_k(a, b, dst);
        return;
    }
    BK_ASSERT(b->type == GGML_TYPE_F32 && dst->type == GGML_TYPE_F32);
    BK_ASSERT(b->nb[0] == 4);
    const TView va = view_of(a), vb = view_of(b), vd = view_of(dst);
    const dim3 grid((unsigned)((a->ne[1] + 3) / 4), (unsigned)b->ne[1], (unsigned)(b->ne[2] * b->ne[3]));
    const double bytes = (double)ggml_nelements(a) * ggml_element_size(a) * 1.0 + (double)ggml_nelements(b) * 4 +
                         (double)ggml_nelements(dst) * 4;
    Timed tm(GGML_HIP_KCLASS_ATTN, bytes);
    if (a->type == GGML_TYPE_F16) {
        BK_ASSERT(a->nb[0] == 2);
        const bool aligned = ((uintptr_t)va.p % 16 == 0) && a->nb[1] % 16 == 0 && a->nb[2] % 16 == 0 && a->nb[3] % 16 == 0;
        if (g.opt_mmq_min > 0 && b->ne[1] >= g.opt_mmq_min && aligned && dst->nb[0] == 4) {
            // prompt batch: both attention products on the f16 matrix cores (kernels/gemm_f16.h)
            GemmF16Args ga;
            ga.a = va.p; ga.a_nb1 = a->nb[1]; ga.a_nb2 = a->nb[2]; ga.a_nb3 = a->nb[3];
            ga.b = vb.p; ga.b_nb1 = b->nb[1]; ga.b_nb2 = b->nb[2]; ga.b_nb3 = b->nb[3];
            ga.d = vd.p; ga.d_nb1 = dst->nb[1]; ga.d_nb2 = dst->nb[2]; ga.d_nb3 = dst->nb[3];
            ga.M = a->ne[1]; ga.N = b->ne[1]; ga.K = a->ne[0];
            ga.ne12 = b->ne[2]; ga.r2 = b->ne[2] / a->ne[2]; ga.r3 = b->ne[3] / a->ne[3];
            ga.tiles_n = (int)((ga.N + 127) / 128);
            ga.causal = 0;
            ga.causal_past = 0;
            const int tiles_m = (int)((ga.M + 127) / 128);
            static DevOnce attr_set;
            if (attr_set.first()) {
                HIP_CHECK(hipFuncSetAttribute((const void *)k_gemm_f16, hipFuncAttributeMaxDynamicSharedMemorySize, MMQ_LDS));
            }
            hipLaunchKernelGGL(k_gemm_f16, dim3((unsigned)(tiles_m * ga.tiles_n), (unsigned)(b->ne[2] * b->ne[3])), dim3(256),
                               MMQ_LDS, g.stream, ga);
        } else
            hipLaunchKernelGGL(k_mul_mat_f16, grid, dim3(256), 0, g.stream, va, vb, vd);
    } else if (a->type == GGML_TYPE_F32) {
        BK_ASSERT(a->nb[0] == 4);
        hipLaunchKernelGGL(k_mul_mat_f32, grid, dim3(256), 0, g.stream, va, vb, vd);
    } else {
        die("mul_mat: unsupported src0 type %s", ggml_type_name(a->type));
    }
    HIP_CHECK(hipGetLastError());
}

void op_rms_norm(ggml_tensor *dst, const ggml_tensor *weight /* nullable: fused mul */, ggml_tensor *out) {
    const ggml_tensor *x = dst->src[0];
    BK_ASSERT(x->type == GGML_TYPE_F32 && x->nb[0] == 4 && out->nb[0] == 4);
    float eps;
    memcpy(&eps, dst->op_params, sizeof(float));
    const int64_t rows = ggml_nrows(x);
    Timed tm(GGML_HIP_KCLASS_OTHER, (double)ggml_nelements(x) * 8);
    if (weight) {
        BK_ASSERT(is_contig_f32(weight) && weight->ne[0] == x->ne[0] && ggml_nelements(weight) == x->ne[0]);
        hipLaunchKernelGGL(k_rms_norm<true>, dim3((unsigned)rows), dim3(256), 0, g.stream, view_of(x), view_of(out),
                           (const float *)dev_ptr(weight), eps);
    } else {
        hipLaunchKernelGGL(k_rms_norm<false>, dim3((unsigned)rows), dim3(256), 0, g.stream, view_of(x), view_of(out),
                           (const float *)nullptr, eps);
    }
    HIP_CHECK(hipGetLastError());
}

void op_norm(ggml_tensor *dst) {
    const ggml_tensor *x = dst->src[0];
    BK_ASSERT(x->type == GGML_TYPE_F32 && x->nb[0] == 4 && dst->nb[0] == 4);
    Timed tm(GGML_HIP_KCLASS_OTHER, (double)ggml_nelements(x) * 8);
    hipLaunchKernelGGL(k_norm, dim3((unsigned)ggml_nrows(x)), dim3(256), 0, g.stream, view_of(x), view_of(dst), 1e-5f);
    HIP_CHECK(hipGetLastError());
}

void op_bin(ggml_tensor *dst, int op) {
    const ggml_tensor *a = dst->src[0], *b = dst->src[1];
    BK_ASSERT(dst->type == GGML_TYPE_F32);
    const int64_t n = ggml_nelements(dst);
    Timed tm(GGML_HIP_KCLASS_OTHER, (double)n * 12);
    if (op == BIN_REPEAT) {
        BK_ASSERT(a->type == GGML_TYPE_F32);
        const TView va = view_of(a), vd = view_of(dst);
        hipLaunchKernelGGL(k_bin<BIN_REPEAT>, grid1(n), dim3(256), 0, g.stream, va, va, vd, n);
    } else {
        BK_ASSERT(a->type == GGML_TYPE_F32 && b->type == GGML_TYPE_F32);
        const TView va = view_of(a), vb = view_of(b), vd = view_of(dst);
        const bool same = is_contig_f32(a) && is_contig_f32(b) && is_contig_f32(dst) && ggml_nelements(a) == n &&
                          ggml_nelements(b) == n && n % 4 == 0 &&
                          (((uintptr_t)va.p | (uintptr_t)vb.p | (uintptr_t)vd.p) & 15) == 0;
        if (same) {
            if (op == BIN_ADD)
                hipLaunchKernelGGL(k_bin4<BIN_ADD>, grid1(n / 4), dim3(256), 0, g.stream, (const f32x4 *)va.p,
                                   (const f32x4 *)vb.p, (f32x4 *)vd.p, n / 4);
            else
                hipLaunchKernelGGL(k_bin4<BIN_MUL>, grid1(n / 4), dim3(256), 0, g.stream, (const f32x4 *)va.p,
                                   (const f32x4 *)vb.p, (f32x4 *)vd.p, n / 4);
        } else if (op == BIN_ADD)
            hipLaunchKernelGGL(k_bin<BIN_ADD>, grid1(n), dim3(256), 0, g.stream, va, vb, vd, n);
        else
            hipLaunchKernelGGL(k_bin<BIN_MUL>, grid1(n), dim3(256), 0, g.stream, va, vb, vd, n);
    }
    HIP_CHECK(hipGetLastError());
}

void op_unary(ggml_tensor *dst, const ggml_tensor *mul_b /* nullable: fused silu*b */, ggml_tensor *out) {
    const ggml_tensor *a = dst->src[0];
    const int32_t uop = dst->op_params[0];
    BK_ASSERT(is_contig_f32(a) && is_contig_f32(out));
    const int64_t n = ggml_nelements(a);
    Timed tm(GGML_HIP_KCLASS_OTHER, (double)n * (mul_b ? 12 : 8));
    const float *pa = (const float *)dev_ptr(a);
    float *pd = (float *)dev_ptr(out);
    if (uop == GGML_UNARY_OP_SILU) {
        if (mul_b) {
            BK_ASSERT(is_contig_f32(mul_b) && ggml_nelements(mul_b) == n);
            const float *pb = (const float *)dev_ptr(mul_b);
            if (n % 4 == 0 && (((uintptr_t)pa | (uintptr_t)pb | (uintptr_t)pd) & 15) == 0)
                hipLaunchKernelGGL((k_unary4<UN_SILU, true>), grid1(n / 4), dim3(256), 0, g.stream, (const f32x4 *)pa,
                                   (const f32x4 *)pb, (f32x4 *)pd, n / 4);
            else
                hipLaunchKernelGGL((k_unary<UN_SILU, true>), grid1(n), dim3(256), 0, g.stream, pa, pb, pd, n);
        } else {
            hipLaunchKernelGGL((k_unary<UN_SILU, false>), grid1(n), dim3(256), 0, g.stream, pa, (const float *)nullptr,
                               pd, n);
        }
    } else if (uop == GGML_UNARY_OP_GELU) {
        BK_ASSERT(!mul_b);
        hipLaunchKernelGGL((k_unary<UN_GELU, false>), grid1(n), dim3(256), 0, g.stream, pa, (const float *)nullptr, pd,
                           n);
    } else {
        die("unary op %d is outside the accelerated path", (int)uop);
    }
    HIP_CHECK(hipGetLastError());
}

void op_scale(ggml_tensor *dst) {
    const ggml_tensor *a = dst->src[0], *s = dst->src[1];
    BK_ASSERT(is_contig_f32(a) && is_contig_f32(dst) && s->type == GGML_TYPE_F32);
    const int64_t n = ggml_nelements(a);
    Timed tm(GGML_HIP_KCLASS_OTHER, (double)n * 8);
    hipLaunchKernelGGL(k_scale, grid1(n), dim3(256), 0, g.stream, (const float *)dev_ptr(a), (const float *)dev_ptr(s),
                       (float *)dev_ptr(dst), n);
    HIP_CHECK(hipGetLastError());
}

void op_diag_mask_inf(ggml_tensor *dst) {
    const ggml_tensor *a = dst->src[0];
    BK_ASSERT(is_contig_f32(a) && is_contig_f32(dst));
    const int64_t n = ggml_nelements(a);
    Timed tm(GGML_HIP_KCLASS_OTHER, (double)n * 8);
    hipLaunchKernelGGL(k_diag_mask_inf, grid1(n), dim3(256), 0, g.stream, (const float *)dev_ptr(a),
                       (float *)dev_ptr(dst), a->ne[0], a->ne[1], n, (int)dst->op_params[0]);
    HIP_CHECK(hipGetLastError());
}

void op_soft_max(ggml_tensor *dst) {
    const ggml_tensor *a = dst->src[0];
    BK_ASSERT(is_contig_f32(a) && is_contig_f32(dst));
    Timed tm(GGML_HIP_KCLASS_OTHER, (double)ggml_nelements(a) * 8);
    hipLaunchKernelGGL(k_soft_max<false>, dim3((unsigned)ggml_nrows(a)), dim3(256), 0, g.stream,
                       (const float *)dev_ptr(a), (float *)dev_ptr(dst), a->ne[0], a->ne[1], (const float *)nullptr, 0);
    HIP_CHECK(hipGetLastError());
}

// fused scale -> diag_mask_inf -> soft_max (all three in-place on the KQ tensor)
void op_scale_mask_softmax(const ggml_tensor *kq, const ggml_tensor *scale, int n_past, ggml_tensor *out) {
    BK_ASSERT(is_contig_f32(kq) && is_contig_f32(out));
    Timed tm(GGML_HIP_KCLASS_OTHER, (double)ggml_nelements(kq) * 8);
    hipLaunchKernelGGL(k_soft_max<true>, dim3((unsigned)ggml_nrows(kq)), dim3(256), 0, g.stream,
                       (const float *)dev_ptr(kq), (float *)dev_ptr(out), kq->ne[0], kq->ne[1],
                       (const float *)dev_ptr(scale), n_past);
    HIP_CHECK(hipGetLastError());
}

void op_rope(ggml_tensor *dst) {
    const ggml_tensor *a = dst->src[0];
    BK_ASSERT(a->type == GGML_TYPE_F32 && dst->type == GGML_TYPE_F32);
    const int n_past = dst->op_params[0], n_dims = dst->op_params[1], mode = dst->op_params[2];
    float freq_base, freq_scale;
    memcpy(&freq_base, dst->op_params + 4, 4);
    memcpy(&freq_scale, dst->op_params + 5, 4);
    if ((mode & ~1) != 0) die("rope mode %d (NeoX/GLM) is outside the accelerated LLaMA path", mode);
    BK_ASSERT(a->ne[0] % 2 == 0);
    const float theta_scale = powf(freq_base, -2.0f / n_dims);
    const int64_t total = (a->ne[0] / 2) * a->ne[1] * a->ne[2] * a->ne[3];
    Timed tm(GGML_HIP_KCLASS_OTHER, (double)ggml_nelements(a) * 8);
    hipLaunchKernelGGL(k_rope, grid1(total), dim3(256), 0, g.stream, view_of(a), view_of(dst), n_past, theta_scale,
                       freq_scale, mode);
    HIP_CHECK(hipGetLastError());
}

void op_cpy(const ggml_tensor *src, ggml_tensor *dst) {
    const int64_t n = ggml_nelements(src);
    BK_ASSERT(n == ggml_nelements(dst));
    const TView vs = view_of(src), vd = view_of(dst);
    Timed tm(GGML_HIP_KCLASS_OTHER, (double)n * (ggml_element_size(src) + ggml_element_size(dst)));
    const ggml_type ts = src->type, td = dst->type;
    if (ts == GGML_TYPE_F32 && td == GGML_TYPE_F32)
        hipLaunchKernelGGL((k_cpy<float, float>), grid1(n), dim3(256), 0, g.stream, vs, vd, n);
    else if (ts == GGML_TYPE_F32 && td == GGML_TYPE_F16)
        hipLaunchKernelGGL((k_cpy<float, __half>), grid1(n), dim3(256), 0, g.stream, vs, vd, n);
    else if (ts == GGML_TYPE_F16 && td == GGML_TYPE_F16)
        hipLaunchKernelGGL((k_cpy<__half, __half>), grid1(n), dim3(256), 0, g.stream, vs, vd, n);
    else if (ts == GGML_TYPE_F16 && td == GGML_TYPE_F32)
        hipLaunchKernelGGL((k_cpy<__half, float>), grid1(n), dim3(256), 0, g.stream, vs, vd, n);
    else if (ts == GGML_TYPE_I32 && td == GGML_TYPE_I32)
        hipLaunchKernelGGL((k_cpy<int, int>), grid1(n), dim3(256), 0, g.stream, vs, vd, n);
    else
        die("cpy %s -> %s is outside the accelerated path", ggml_type_name(ts), ggml_type_name(td));
    HIP_CHECK(hipGetLastError());
}

void op_get_rows(ggml_tensor *dst) {
    const ggml_tensor *tab = dst->src[0], *ids = dst->src[1];
    BK_ASSERT(ids->type == GGML_TYPE_I32 && is_contig_f32(dst));
    const int64_t N = ids->ne[0], ne0 = tab->ne[0];
    const int *pid = (const int *)dev_ptr(ids);
    float *pd = (float *)dev_ptr(dst);
    Timed tm(GGML_HIP_KCLASS_OTHER, (double)N * ne0 * 5);
    if (qt_of(tab->type) >= 0) {
        const QWeight w = qweight_of(tab);
        hipLaunchKernelGGL(k_get_rows_q, dim3((unsigned)((w.nb + 255) / 256), (unsigned)N), dim3(256), 0, g.stream, w,
                           pid, pd, ne0);
    } else if (kt_of(tab->type) >= 0) {
        dequant_k_rows(kweight_of(tab), pid, N, pd, ne0);
    } else if (tab->type == GGML_TYPE_F16) {
        hipLaunchKernelGGL(k_get_rows<__half>, dim3((unsigned)((ne0 + 255) / 256), (unsigned)N), dim3(256), 0, g.stream,
                           (const char *)dev_ptr(tab), (int64_t)tab->nb[1], pid, pd, ne0);
    } else if (tab->type == GGML_TYPE_F32) {
        hipLaunchKernelGGL(k_get_rows<float>, dim3((unsigned)((ne0 + 255) / 256), (unsigned)N), dim3(256), 0, g.stream,
                           (const char *)dev_ptr(tab), (int64_t)tab->nb[1], pid, pd, ne0);
    } else {
        die("get_rows on %s is outside the accelerated path", ggml_type_name(tab->type));
    }
    HIP_CHECK(hipGetLastError());
}

// ---------------------------------------------------------------------------------------------------
// graph executor
// ---------------------------------------------------------------------------------------------------
bool is_view_op(ggml_op op) {
    return op == GGML_OP_NONE || op == GGML_OP_RESHAPE || op == GGML_OP_VIEW || op == GGML_OP_PERMUTE ||
           op == GGML_OP_TRANSPOSE;
}

struct GraphInfo {
    std::vector<int> n_uses;  // consumers per node index
};

int node_index(const ggml_cgraph *gr, const ggml_tensor *t, const std::map<const ggml_tensor *, int> &idx) {
    (void)gr;
    auto it = idx.find(t);
    return it == idx.end() ? -1 : it->second;
}

void upload_inputs(ggml_cgraph *gr) {
    if (gr->n_nodes == 0) return;
    // the compute context = the arena holding the tensor headers of the graph's nodes
    Arena *r0 = find_arena((uintptr_t)gr->nodes[gr->n_nodes - 1]);
    for (int i = 0; i < gr->n_leafs; i++) {
        ggml_tensor *leaf = gr->leafs[i];
        if (leaf->data == nullptr) continue;
        if (extra_of(leaf)) continue;
        const uintptr_t p = (uintptr_t)leaf->data;
        if (DevTensor *e = find_tensor(p)) {
            // auto-uploaded earlier; make sure it still describes this tensor
            if (e->auto_uploaded && (e->host != p || e->nbytes != ggml_nbytes(leaf) || e->type != leaf->type ||
                                     e->owner_hdr != (uintptr_t)leaf)) {
                free_dev_tensor(e);
            } else {
                continue;
            }
        }
        Arena *hdr = find_arena((uintptr_t)leaf);
        const bool per_eval = r0 && hdr == r0;
        if (per_eval) {
            // inputs written by the host before every compute (token ids, scalar constants, test operands)
            const size_t nbytes = ggml_nbytes(leaf);
            if (nbytes == 0) continue;
            h2d_small(dev_ptr(leaf), leaf->data, nbytes);
        } else {
            // persistent tensor (weight / KV memory) that was never offloaded by the caller: upload once
            upload_tensor(leaf->data, leaf, false, /*is_auto=*/true)->owner_hdr = (uintptr_t)leaf;
        }
    }
}

void download_outputs(ggml_cgraph *gr) {
    bool any = false;
    for (int i = 0; i < gr->n_nodes; i++) {
        ggml_tensor *n = gr->nodes[i];
        if (n->backend != GGML_BACKEND_CPU || is_view_op(n->op) || n->op == GGML_OP_CPY) continue;
        if (!ggml_is_contiguous(n) || n->data == nullptr) continue;
        if (extra_of(n) || find_tensor((uintptr_t)n->data)) continue;  // result aliases a device-resident tensor
        d2h_queue(n->data, dev_ptr(n), ggml_nbytes(n));
        any = true;
    }
    (void)any;
    d2h_finish();
}

void invalidate_xf16_if_overwritten_impl(const ggml_tensor *n);
void invalidate_qact_if_overwritten(const ggml_tensor *n) {
    if (g_xi8.valid && n->data != nullptr) {
        const uintptr_t c0 = (uintptr_t)g_xi8.src_data, c1 = c0 + g_xi8.src_bytes;
        const uintptr_t b0 = (uintptr_t)n->data, b1 = b0 + ggml_nbytes(n);
        if (b0 < c1 && c0 < b1) g_xi8.valid = false;
    }
    if (g_xk.valid && n->data != nullptr) {
        const uintptr_t c0 = (uintptr_t)g_xk.src_data, c1 = c0 + g_xk.src_bytes;
        const uintptr_t b0 = (uintptr_t)n->data, b1 = b0 + ggml_nbytes(n);
        if (b0 < c1 && c0 < b1) g_xk.valid = false;
    }
    invalidate_xf16_if_overwritten_impl(n);
    if (!g_qact.valid || n->data == nullptr) return;
    const uintptr_t a0 = (uintptr_t)g_qact.src_data, a1 = a0 + g_qact.src_bytes;
    const uintptr_t b0 = (uintptr_t)n->data, b1 = b0 + ggml_nbytes(n);
    if (b0 < a1 && a0 < b1) g_qact.valid = false;
}
void invalidate_xf16_if_overwritten_impl(const ggml_tensor *n) {
    if (!g_xf16.valid || n->data == nullptr) return;
    const uintptr_t a0 = (uintptr_t)g_xf16.src_data, a1 = a0 + g_xf16.src_bytes;
    const uintptr_t b0 = (uintptr_t)n->data, b1 = b0 + ggml_nbytes(n);
    if (b0 < a1 && a0 < b1) g_xf16.valid = false;
    if (g_xq8.valid) {
        const uintptr_t c0 = (uintptr_t)g_xq8.src_data, c1 = c0 + g_xq8.src_bytes;
        if (b0 < c1 && c0 < b1) g_xq8.valid = false;
    }
}

void finish_pending();
#include "llama_plan.inc"

void finish_pending() {
    if (!g.pending_wait) return;
    const uint64_t t = now_ns();
    d2h_finish();
    g.ns_wait += now_ns() - t;
    g.pending_wait = false;
}

void execute_graph(ggml_cgraph *gr) {
    ensure_init();
    finish_pending();
    ws_reset();
    g_qact.valid = false;
    g_xf16.valid = false;
    g_xq8.valid = false;
    g_xi8.valid = false;
    g_xk.valid = false;
    if (try_decode_plan(gr)) return;  // single-token LLaMA decode: fused launches + hipGraph replay
    g.stat_generic_graphs++;
    upload_inputs(gr);

    std::map<const ggml_tensor *, int> idx;
    std::vector<int> uses(gr->n_nodes, 0);
    for (int i = 0; i < gr->n_nodes; i++) idx[gr->nodes[i]] = i;
    for (int i = 0; i < gr->n_nodes; i++)
        for (int s = 0; s < GGML_MAX_SRC; s++)
            if (gr->nodes[i]->src[s]) {
                int j = node_index(gr, gr->nodes[i]->src[s], idx);
                if (j >= 0) uses[j]++;
            }
    std::vector<char> done(gr->n_nodes, 0);
    const bool fuse = g.opt_fuse != 0;
    // silu(a) whose only consumer is a later mul(silu(a), b) — the FFN gate; the reference's build order puts the
    // w3 mat-mul between the two (nodes: w1·x, silu, w3·x, mul), so the pair is not adjacent: the silu is deferred
    // and executed fused when its mul comes up (nothing in between writes a's buffer: it is a live operand).
    std::vector<int> deferred_silu(gr->n_nodes, -1);  // index of the mul that will run it
    if (fuse) {
        for (int j = 0; j < gr->n_nodes; j++) {
            ggml_tensor *mu = gr->nodes[j];
            if (mu->op != GGML_OP_MUL || !mu->src[0] || mu->src[0]->op != GGML_OP_UNARY) continue;
            const int i = node_index(gr, mu->src[0], idx);
            ggml_tensor *un = mu->src[0];
            if (i < 0 || i >= j || uses[i] != 1 || un->op_params[0] != GGML_UNARY_OP_SILU) continue;
            if (!is_contig_f32(un->src[0]) || !is_contig_f32(mu->src[1]) || !is_contig_f32(mu) ||
                ggml_nelements(mu->src[1]) != ggml_nelements(un) || ggml_nelements(mu) != ggml_nelements(un))
                continue;
            deferred_silu[i] = j;
        }
    }

    for (int i = 0; i < gr->n_nodes; i++) {
        ggml_tensor *n = gr->nodes[i];
        if (done[i] || is_view_op(n->op)) continue;
        invalidate_qact_if_overwritten(n);
        ggml_tensor *next = i + 1 < gr->n_nodes ? gr->nodes[i + 1] : nullptr;
        switch (n->op) {
            case GGML_OP_GET_ROWS: op_get_rows(n); break;
            case GGML_OP_RMS_NORM: {
                // fuse the broadcast multiply by the norm weight that follows (llama lib.rs:183-186)
                if (fuse && next && next->op == GGML_OP_MUL && next->src[0] == n && uses[i] == 1 && !done[i + 1] &&
                    is_contig_f32(next->src[1]) && ggml_nelements(next->src[1]) == n->ne[0] && next->nb[0] == 4) {
                    invalidate_qact_if_overwritten(next);
                    op_rms_norm(n, next->src[1], next);
                    done[i + 1] = 1;
                } else {
                    op_rms_norm(n, nullptr, n);
                }
            } break;
            case GGML_OP_NORM: op_norm(n); break;
            case GGML_OP_ADD: op_bin(n, BIN_ADD); break;
            case GGML_OP_MUL: {
                const int si = n->src[0] && n->src[0]->op == GGML_OP_UNARY ? node_index(gr, n->src[0], idx) : -1;
                if (si >= 0 && deferred_silu[si] == i)
                    op_unary(n->src[0], n->src[1], n);  // silu(a) * b in one pass
                else
                    op_bin(n, BIN_MUL);
            } break;
            case GGML_OP_REPEAT: op_bin(n, BIN_REPEAT); break;
            case GGML_OP_UNARY: {
                if (deferred_silu[i] >= 0) break;  // runs fused with its mul
                if (fuse && n->op_params[0] == GGML_UNARY_OP_SILU && next && next->op == GGML_OP_MUL &&
                    next->src[0] == n && uses[i] == 1 && !done[i + 1] && is_contig_f32(next->src[1]) &&
                    ggml_nelements(next->src[1]) == ggml_nelements(n) && is_contig_f32(next)) {
                    invalidate_qact_if_overwritten(next);
                    op_unary(n, next->src[1], next);
                    done[i + 1] = 1;
                } else {
                    op_unary(n, nullptr, n);
                }
            } break;
            case GGML_OP_MUL_MAT: op_mul_mat(n); break;
            case GGML_OP_SCALE: {
                // fuse scale -> diag_mask_inf -> soft_max when chained in place (llama lib.rs:268-281)
                ggml_tensor *n1 = next, *n2 = i + 2 < gr->n_nodes ? gr->nodes[i + 2] : nullptr;
                if (fuse && n1 && n2 && n1->op == GGML_OP_DIAG_MASK_INF && n1->src[0] == n && n2->op == GGML_OP_SOFT_MAX &&
                    n2->src[0] == n1 && uses[i] == 1 && uses[i + 1] == 1 && n->data == n->src[0]->data &&
                    n1->data == n->data && n2->data == n->data && is_contig_f32(n)) {
                    op_scale_mask_softmax(n->src[0], n->src[1], (int)n1->op_params[0], n2);
                    done[i + 1] = done[i + 2] = 1;
                } else {
                    op_scale(n);
                }
            } break;
            case GGML_OP_DIAG_MASK_INF: op_diag_mask_inf(n); break;
            case GGML_OP_SOFT_MAX: op_soft_max(n); break;
            case GGML_OP_ROPE: op_rope(n); break;
            case GGML_OP_CPY: op_cpy(n->src[0], n->src[1]); break;
            case GGML_OP_CONT:
            case GGML_OP_DUP: op_cpy(n->src[0], n); break;
            default:
                die("op %s (node '%s') is outside the accelerated path and this library has no CPU fallback",
                    ggml_op_name(n->op), n->name);
        }
    }
    download_outputs(gr);
}

}  // namespace

// ===================================================================================================
// exported: internal seam
// ===================================================================================================
// Host arenas are known to every slot (a context made while one device is current may be used while another is): the
// registry is per slot, so that each slot owns its shadows, and both calls are applied to all of them.
static void register_arena_here(void *host_base, size_t size, int is_scratch);
static void unregister_arena_here(void *host_base);
extern "C" void ggml_hip_internal_register_arena(void *host_base, size_t size, int is_scratch) {
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    for_each_slot([&] { register_arena_here(host_base, size, is_scratch); });
}
extern "C" void ggml_hip_internal_unregister_arena(void *host_base) {
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    for_each_slot([&] { unregister_arena_here(host_base); });
}
static void register_arena_here(void *host_base, size_t size, int is_scratch) {
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

static void unregister_arena_here(void *host_base) {
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

extern "C" void ggml_hip_internal_graph_compute(struct ggml_cgraph *cgraph) {
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    const uint64_t t0 = now_ns();
    execute_graph(cgraph);
    g.ns_compute += now_ns() - t0;
}
// Split form of ggml_graph_compute for callers that have host work to overlap with the device (the session
// mirror builds the next token's graph meanwhile): begin() enqueues the graph and returns 1 if it is still running
// (fused decode plan), 0 if it was executed synchronously (any other graph); end() waits and finishes the
// read-back of the host-visible results.  Nothing else may read results before end().
extern "C" int ggml_hip_graph_compute_begin(struct ggml_cgraph *cgraph) {
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    const uint64_t t0 = now_ns();
    ensure_init();
    finish_pending();
    ws_reset();
    g_qact.valid = false;
    g_xf16.valid = false;
    g_xq8.valid = false;
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
extern "C" void ggml_hip_graph_compute_end(void) {
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    const uint64_t t0 = now_ns();
    finish_pending();
    g.ns_compute += now_ns() - t0;
}

// ===================================================================================================
// exported: the 19 accelerator hooks (crates/ggml/sys/src/cuda.rs:6-77)
// ===================================================================================================
extern "C" {

void ggml_init_hipblas(void) {
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    ensure_init();
}
void ggml_hip_set_main_device(int main_device) {
    // crates/ggml/sys/src/cuda.rs:62 (accelerator/mod.rs:72): the slot every following call acts on
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    if (main_device < 0 || main_device >= std::max(1, slot_count()))
        die("ggml_hip_set_main_device(%d): %d device slot(s) available", main_device, slot_count());
    g_cur = &g_backends[main_device];
    g.slot = main_device;
    if (g.inited) bind_device();
}
void ggml_hip_set_tensor_split(const float *tensor_split) {
    // crates/ggml/sys/src/cuda.rs:11.  The reference passes a single 1.0 (crates/ggml/src/accelerator/mod.rs:74-75).
    // Here the fractions (one per slot, ggml's convention: device i takes the share split[i] / sum) are kept for the host
    // side, which turns them into a LAYER split (llm_split_layers, host/llm_host.cpp): rows of one tensor are never split.
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    const int n = std::max(1, slot_count());
    for (int i = 0; i < GGML_HIP_MAX_BACKENDS; i++) g_tensor_split[i] = tensor_split && i < n ? tensor_split[i] : 0.0f;
}
int ggml_hip_get_tensor_split(float *out, int cap) {
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    const int n = std::min(cap, std::max(1, slot_count()));
    for (int i = 0; i < n; i++) out[i] = g_tensor_split[i];
    return n;
}
void ggml_hip_set_mul_mat_q(bool) {}  // quantized kernels are always used
void ggml_hip_set_scratch_size(size_t) {}  // activations live in arena shadows, there is no scratch pool
void ggml_hip_free_scratch(void) {
    std::lock_guard<std::recursive_mutex> lk(g_mu);
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
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    ensure_init();
    void *p = nullptr;
    if (hipHostMalloc(&p, size, hipHostMallocDefault) != hipSuccess) return nullptr;
    return p;
}
void ggml_hip_host_free(void *ptr) {
    if (ptr) HIP_CHECK(hipHostFree(ptr));
}
void ggml_hip_transform_tensor(void *data, struct ggml_tensor *tensor) {
    std::lock_guard<std::recursive_mutex> lk(g_mu);
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
    std::lock_guard<std::recursive_mutex> lk(g_mu);
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
    std::lock_guard<std::recursive_mutex> lk(g_mu);
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
    std::lock_guard<std::recursive_mutex> lk(g_mu);
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
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    if (!tensor || !tensor->extra) return;
    DevTensor *e = (DevTensor *)tensor->extra;
    if (e->magic != 0x48495054) return;  // scratch-assigned node: nothing to free (as in the reference)
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
    std::lock_guard<std::recursive_mutex> lk(g_mu);
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
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    ensure_init();
    BK_ASSERT(dst->src[0] == src0 && dst->src[1] == src1);
    op_mul_mat(dst);
}
void ggml_hip_mul(const struct ggml_tensor *src0, const struct ggml_tensor *src1, struct ggml_tensor *dst) {
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    ensure_init();
    BK_ASSERT(dst->src[0] == src0 && dst->src[1] == src1);
    op_bin(dst, BIN_MUL);
}
bool ggml_hip_compute_forward(struct ggml_compute_params *params, struct ggml_tensor *tensor) {
    // Per-node hook of the reference's CPU executor. This library executes whole graphs itself
    // (ggml_graph_compute), so the hook only has to answer for callers that drive nodes one by one.
    std::lock_guard<std::recursive_mutex> lk(g_mu);
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
int ggml_hip_get_main_device(void) {
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    return (int)(g_cur - g_backends);
}
// The residual of a layer split crossing from one slot to another: dst on slot dst_device's stream waits for what
// src_device's stream has enqueued so far, then copies (peer copy over xGMI between two GPUs, a device copy when both slots
// sit on one GPU).  Asynchronous; ordered with both slots' later work on their own streams.
void ggml_hip_copy_between_devices(int dst_device, void *dst, int src_device, const void *src, size_t nbytes) {
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    if (dst_device < 0 || src_device < 0 || dst_device >= GGML_HIP_MAX_BACKENDS || src_device >= GGML_HIP_MAX_BACKENDS)
        die("ggml_hip_copy_between_devices: bad slot");
    Backend &S = g_backends[src_device], &D = g_backends[dst_device];
    if (!S.inited || !D.inited) die("ggml_hip_copy_between_devices: slot not initialised");
    Backend *keep = g_cur;
    g_cur = &S;
    bind_device();
    hipEvent_t ev;
    HIP_CHECK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    HIP_CHECK(hipEventRecord(ev, S.stream));
    g_cur = &D;
    bind_device();
    HIP_CHECK(hipStreamWaitEvent(D.stream, ev, 0));
    if (S.device == D.device)
        HIP_CHECK(hipMemcpyAsync(dst, src, nbytes, hipMemcpyDeviceToDevice, D.stream));
    else
        HIP_CHECK(hipMemcpyPeerAsync(dst, D.device, src, S.device, nbytes, D.stream));
    HIP_CHECK(hipEventDestroy(ev));  // released once the recorded work completes
    g_cur = keep;
    if (g.inited) bind_device();
}
void ggml_hip_synchronize(void) {
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    if (g.inited) HIP_CHECK(hipStreamSynchronize(g.stream));
}
void ggml_hip_tensor_get(const struct ggml_tensor *tensor, void *host_dst, size_t offset, size_t nbytes) {
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    ensure_init();
    d2h_queue(host_dst, dev_ptr(tensor) + offset, nbytes);
    d2h_finish();
}
void ggml_hip_tensor_set(struct ggml_tensor *tensor, const void *host_src, size_t offset, size_t nbytes) {
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    ensure_init();
    h2d_bulk(dev_ptr(tensor) + offset, host_src, nbytes);
    HIP_CHECK(hipStreamSynchronize(g.stream));
}
// Raw copies on the backend stream, synchronous (layer-split driver: moving the residual between a stage's
// hand-off buffer and the communication library's buffers). kind: 0 = host→device, 1 = device→host, 2 = device→device.
void ggml_hip_memcpy(void *dst, const void *src, size_t nbytes, int kind) {
    std::lock_guard<std::recursive_mutex> lk(g_mu);
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
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    ensure_init();
    return dev_ptr(tensor);
}
void ggml_hip_timing_begin(void) {
    std::lock_guard<std::recursive_mutex> lk(g_mu);
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
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    g.timing.on = false;
    if (g.inited) HIP_CHECK(hipStreamSynchronize(g.stream));
}
void ggml_hip_timing_query(int kclass, double *ms, int64_t *launches, double *algo_bytes) {
    std::lock_guard<std::recursive_mutex> lk(g_mu);
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
static void set_option_here(const char *key, int value);
void ggml_hip_set_option(const char *key, int value) {  // options are process-wide: every slot gets them
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    for_each_slot([&] { set_option_here(key, value); });
}
static void set_option_here(const char *key, int value) {
    const std::string k(key);
    if (k == "fuse")
        g.opt_fuse = value;
    else if (k == "plan")
        g.opt_plan = value;
    else if (k == "graph")
        g.opt_graph = value;
    else if (k == "xsrc") {
        if (g.opt_xsrc != value) drop_all_plans();
        g.opt_xsrc = value;
    }
    else if (k == "timeline") {
        drop_all_plans();
        if (g.timeline && value && (value == 1 ? 4 : value) != g.timeline_wgs) {
            HIP_CHECK(hipStreamSynchronize(g.stream));
            (void)hipFree(g.timeline);
            g.timeline = nullptr;
        }
        if (value && !g.timeline) {
            ensure_init();
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
    else if (k == "attn_split") {
        if (g.opt_attn_split != value) drop_all_plans();
        g.opt_attn_split = value;
    }
    else if (k == "prefetch") {
        if (g.opt_prefetch != value) drop_all_plans();
        g.opt_prefetch = value;
    }
    else if (k == "prefetch_wo") {
        if (g.opt_prefetch_wo != value) drop_all_plans();
        g.opt_prefetch_wo = value;
    }
    else if (k == "prefetch_wgs") {
        if (g.opt_prefetch_wgs != value) drop_all_plans();
        g.opt_prefetch_wgs = value;
    }
    else if (k == "prefetch_delay") {
        if (g.opt_prefetch_delay != value) drop_all_plans();
        g.opt_prefetch_delay = value;
    }
    else if (k == "mmq_fuse")
        g.opt_mmq_fuse = value;
    else if (k == "mmq_persist")
        g.opt_mmq_persist = value;
    else if (k == "mmq_waves")
        g.opt_mmq_waves = value;
    else if (k == "mmq_t256_var")
        g.opt_mmq_t256_var = value;
    else if (k == "mmq_splits")
        g.opt_mmq_splits = value;
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
    else if (k == "plan_multi") {
        if (g.opt_plan_multi != value) drop_all_plans();
        g.opt_plan_multi = value;
    }
    else if (k == "big") {
        if (g.opt_big != value) drop_all_plans();
        g.opt_big = value;
    }
    else if (k == "probe") {
        if (g.opt_probe != value) drop_all_plans();
        g.opt_probe = value;
    }
    else if (k == "mmvq_rows")
        g.opt_mmvq_rows = value;
    else if (!strcmp(key, "mmq_min"))
        g.opt_mmq_min = value;
    else if (!strcmp(key, "mmq_splitk"))
        g.opt_mmq_splitk = value;
    else if (!strcmp(key, "mmq_dma"))
        g.opt_mmq_dma = value;
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
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    unsigned kind_mask = ~0u;
    if (kclass >= GGML_HIP_KKIND_BASE && kclass < GGML_HIP_KKIND_BASE + 5) {  // one kind of mat-vec launch alone
        kind_mask = 1u << (kclass - GGML_HIP_KKIND_BASE);
        kclass = GGML_HIP_KCLASS_MMVQ;
    }
    if (g_plans.empty() || kclass < 0 || kclass >= GGML_HIP_KCLASS_COUNT || replays < 1) return -1;
    DecodePlan *p = g_plans.back();
    HIP_CHECK(hipStreamSynchronize(g.stream));
    DecParams saved, park;
    HIP_CHECK(hipMemcpy(&saved, p->prm, sizeof(saved), hipMemcpyDeviceToHost));
    park = saved;
    if (kclass == GGML_HIP_KCLASS_MMVQ) park.n_past = (int)p->m.C - 1;  // K/V stores go to a scratch slot
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
        plan_launch_all(p, 1u << kclass, &st, kind_mask);
        HIP_CHECK(hipStreamEndCapture(g.stream, &gr));
        HIP_CHECK(hipGraphInstantiate(&ex, gr, nullptr, nullptr, 0));
        HIP_CHECK(hipGraphLaunch(ex, g.stream));  // warm
        HIP_CHECK(hipEventRecord(a, g.stream));
        for (int i = 0; i < replays; i++) HIP_CHECK(hipGraphLaunch(ex, g.stream));
        HIP_CHECK(hipEventRecord(b, g.stream));
    } else {  // GGML_HIP_GRAPH=0 (e.g. under rocprofv3, whose kernel tracing crashes on graph launches here)
        plan_launch_all(p, 1u << kclass, &st, kind_mask);
        HIP_CHECK(hipEventRecord(a, g.stream));
        for (int i = 0; i < replays; i++) plan_launch_all(p, 1u << kclass, nullptr, kind_mask);
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

// ===================================================================================================
// Layer split over RCCL: one process per GPU, the residual [n_embd x N] f32 crosses a stage
// boundary with ncclSend / ncclRecv on the backend's own stream — stream-ordered with the kernels on either side, no
// host synchronisation per hop, no torch tensor in the data path.  librccl.so (0.5 GB) is opened on first use, so
// single-GPU users neither need nor load it; a missing library or any RCCL error aborts with a message.
// ===================================================================================================
}  // extern "C"
#include <dlfcn.h>
#if __has_include(<rccl/rccl.h>)
#include <rccl/rccl.h>
#else
// Single-GPU builds need no RCCL development package: librccl is opened at run time (rccl_load), and these are the few
// declarations of its stable C API the hop uses (nccl.h: ncclUniqueId is 128 opaque bytes, ncclUint8 = 1).
extern "C" {
typedef struct ncclComm *ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef enum { ncclSuccess = 0 } ncclResult_t;
typedef enum { ncclInt8 = 0, ncclUint8 = 1, ncclFloat32 = 7 } ncclDataType_t;
}
#endif
namespace {
struct Rccl {
    void *dl = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int *) = nullptr;
    ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    ncclComm_t comm = nullptr;
    int rank = -1, world = 0;
} rccl;
void rccl_load() {
    if (rccl.dl) return;
    const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char *n : names)
        if ((rccl.dl = dlopen(n, RTLD_NOW | RTLD_LOCAL))) break;
    if (!rccl.dl) die("cannot open librccl.so (%s): the layer split across GPUs needs RCCL", dlerror());
    auto sym = [&](const char *n) {
        void *p = dlsym(rccl.dl, n);
        if (!p) die("librccl.so lacks %s", n);
        return p;
    };
    rccl.GetUniqueId = (decltype(rccl.GetUniqueId))sym("ncclGetUniqueId");
    rccl.CommInitRank = (decltype(rccl.CommInitRank))sym("ncclCommInitRank");
    rccl.CommDestroy = (decltype(rccl.CommDestroy))sym("ncclCommDestroy");
    rccl.CommCount = (decltype(rccl.CommCount))sym("ncclCommCount");
    rccl.Send = (decltype(rccl.Send))sym("ncclSend");
    rccl.Recv = (decltype(rccl.Recv))sym("ncclRecv");
    rccl.GroupStart = (decltype(rccl.GroupStart))sym("ncclGroupStart");
    rccl.GroupEnd = (decltype(rccl.GroupEnd))sym("ncclGroupEnd");
    rccl.GetErrorString = (decltype(rccl.GetErrorString))sym("ncclGetErrorString");
}
#define RCCL_CHECK(x)                                                                                  \
    do {                                                                                               \
        ncclResult_t r_ = (x);                                                                         \
        if (r_ != ncclSuccess) die("RCCL error %s at %s:%d (%s)", rccl.GetErrorString(r_), __FILE__, __LINE__, #x); \
    } while (0)
void comm_need() {
    if (!rccl.comm) die("ggml_hip_comm_*: no communicator (call ggml_hip_comm_init first)");
}
}  // namespace
extern "C" {
int ggml_hip_comm_unique_id(void *id_out) {
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    ensure_init();
    rccl_load();
    ncclUniqueId id;
    RCCL_CHECK(rccl.GetUniqueId(&id));
    memcpy(id_out, &id, sizeof(id));
    return (int)sizeof(id);
}
int ggml_hip_comm_init(int rank, int world, const void *id_in) {
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    ensure_init();
    rccl_load();
    if (rccl.comm) die("ggml_hip_comm_init: a communicator already exists");
    if (world < 1 || rank < 0 || rank >= world) die("ggml_hip_comm_init: bad rank %d of %d", rank, world);
    ncclUniqueId id;
    memcpy(&id, id_in, sizeof(id));
    HIP_CHECK(hipSetDevice(g.device));
    RCCL_CHECK(rccl.CommInitRank(&rccl.comm, world, id, rank));
    rccl.rank = rank;
    rccl.world = world;
    int n = 0;
    RCCL_CHECK(rccl.CommCount(rccl.comm, &n));
    return n;
}
void ggml_hip_comm_destroy(void) {
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    if (!rccl.comm) return;
    HIP_CHECK(hipStreamSynchronize(g.stream));
    RCCL_CHECK(rccl.CommDestroy(rccl.comm));
    rccl.comm = nullptr;
    rccl.rank = -1;
    rccl.world = 0;
}
int ggml_hip_comm_ranks(void) {
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    if (!rccl.comm) return 0;
    int n = 0;
    RCCL_CHECK(rccl.CommCount(rccl.comm, &n));
    return n;
}
void ggml_hip_comm_send(const void *dev_src, size_t nbytes, int peer) {
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    comm_need();
    RCCL_CHECK(rccl.Send(dev_src, nbytes, ncclUint8, peer, rccl.comm, g.stream));
}
void ggml_hip_comm_recv(void *dev_dst, size_t nbytes, int peer) {
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    comm_need();
    RCCL_CHECK(rccl.Recv(dev_dst, nbytes, ncclUint8, peer, rccl.comm, g.stream));
}
void ggml_hip_comm_sendrecv(const void *dev_src, int send_peer, void *dev_dst, int recv_peer, size_t nbytes) {
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    comm_need();
    RCCL_CHECK(rccl.GroupStart());
    RCCL_CHECK(rccl.Send(dev_src, nbytes, ncclUint8, send_peer, rccl.comm, g.stream));
    RCCL_CHECK(rccl.Recv(dev_dst, nbytes, ncclUint8, recv_peer, rccl.comm, g.stream));
    RCCL_CHECK(rccl.GroupEnd());
}

// Launch-floor probe (tests/tools/launch_probe.py): a linear hipGraph of `n_launch` launches of a kernel that does
// nothing but stamp the 100 MHz wall clock — at its very first instruction, again once its LAST kernel argument has
// arrived, and at its end — with the launch shape of the decode mat-vecs (threads per workgroup, dynamic LDS, size
// of the kernarg segment).  Separates what a kernel boundary costs by itself, and how long the kernel arguments
// take to arrive, from what the mat-vec kernels add.  out[0] = us per launch (HIP events around `replays` replays),
// out[1] = us from one launch's end stamp to the next launch's first instruction, out[2] = us first instruction ->
// last kernel argument usable (workgroup 0 of each launch).
}  // extern "C"
namespace {
template <int NARG>
struct EmptyArgs {
    long long *ts;
    int idx;
    int pad[(NARG - 12) / 4];
};
template <int NARG>
__global__ void __launch_bounds__(1024) k_empty(const EmptyArgs<NARG> a, int last_arg) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    long long t0;
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0));
    const int v = a.pad[(NARG - 12) / 4 - 1] + last_arg;  // the far end of the kernarg segment
    long long t1 = v != 0x7fffffff ? (long long)wall_clock64() : 0;
    if (v == 0x12345678) smem[threadIdx.x] = 1;  // keeps the dynamic LDS allocation referenced
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        a.ts[a.idx * 4] = t0;
        a.ts[a.idx * 4 + 1] = t1;
        a.ts[a.idx * 4 + 2] = (long long)wall_clock64();
    }
}
template <int NARG>
void empty_launch(int wgs, int threads, int lds, long long *ts, int idx) {
    EmptyArgs<NARG> a;
    memset(&a, 0, sizeof(a));
    a.ts = ts;
    a.idx = idx;
    static DevOnce attr;
    if (attr.first()) {
        HIP_CHECK(hipFuncSetAttribute((const void *)k_empty<NARG>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    }
    hipLaunchKernelGGL(k_empty<NARG>, dim3(wgs), dim3(threads), (size_t)lds, g.stream, a, 0);
}
}  // namespace
extern "C" {
int ggml_hip_bench_empty(int wgs, int threads, int lds_bytes, int kernarg_bytes, int n_launch, int replays, double *out) {
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    ensure_init();
    if (n_launch < 2 || n_launch > 512 || replays < 1 || threads < 64 || threads > 1024 || lds_bytes > 160 * 1024) return -1;
    long long *ts = nullptr;
    HIP_CHECK(hipMalloc((void **)&ts, (size_t)n_launch * 32));
    HIP_CHECK(hipMemsetAsync(ts, 0, (size_t)n_launch * 32, g.stream));
    hipGraph_t gr = nullptr;
    hipGraphExec_t ex = nullptr;
    HIP_CHECK(hipStreamBeginCapture(g.stream, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < n_launch; i++) {
        if (kernarg_bytes <= 64) empty_launch<64>(wgs, threads, lds_bytes, ts, i);
        else if (kernarg_bytes <= 192) empty_launch<192>(wgs, threads, lds_bytes, ts, i);
        else empty_launch<448>(wgs, threads, lds_bytes, ts, i);
    }
    HIP_CHECK(hipStreamEndCapture(g.stream, &gr));
    HIP_CHECK(hipGraphInstantiate(&ex, gr, nullptr, nullptr, 0));
    HIP_CHECK(hipGraphLaunch(ex, g.stream));
    hipEvent_t a, b;
    HIP_CHECK(hipEventCreate(&a));
    HIP_CHECK(hipEventCreate(&b));
    HIP_CHECK(hipEventRecord(a, g.stream));
    for (int i = 0; i < replays; i++) HIP_CHECK(hipGraphLaunch(ex, g.stream));
    HIP_CHECK(hipEventRecord(b, g.stream));
    HIP_CHECK(hipStreamSynchronize(g.stream));
    float ms = 0;
    HIP_CHECK(hipEventElapsedTime(&ms, a, b));
    std::vector<long long> h((size_t)n_launch * 4);
    HIP_CHECK(hipMemcpy(h.data(), ts, (size_t)n_launch * 32, hipMemcpyDeviceToHost));
    double gap = 0, karg = 0;
    for (int i = 1; i < n_launch; i++) {
        gap += (double)(h[i * 4] - h[(i - 1) * 4 + 2]) / 100.0;
        karg += (double)(h[i * 4 + 1] - h[i * 4]) / 100.0;
    }
    out[0] = (double)ms * 1e3 / ((double)n_launch * replays);
    out[1] = gap / (n_launch - 1);
    out[2] = karg / (n_launch - 1);
    HIP_CHECK(hipEventDestroy(a));
    HIP_CHECK(hipEventDestroy(b));
    HIP_CHECK(hipGraphExecDestroy(ex));
    HIP_CHECK(hipGraphDestroy(gr));
    HIP_CHECK(hipFree(ts));
    return 0;
}

// Test hook: the attention of a prompt batch on host arrays through either path of the prompt plan (llama_plan.inc
// prompt_attention): q [N][E] f32 with RoPE applied, mem_k [C][Egqa] / mem_v [Egqa][C] f16 of one layer, out [N][E] f32.
// Returns 0, or -1 when `fused` is asked for a shape the fused kernel does not take.
int ggml_hip_debug_prompt_attention(const float *q, const uint16_t *mem_k, const uint16_t *mem_v, float *out, int N, int E, int Egqa,
                                    int H, int n_past, int64_t C, float scale, int fused) {
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    ensure_init();
    finish_pending();
    const int64_t D = E / H, Hkv = Egqa / D, T = (int64_t)n_past + N, Tp = (T + 7) & ~(int64_t)7;
    if (T > C || (fused && !prompt_attn_fits(D, T))) return -1;
    static DevOnce attr;
    if (attr.first()) {
        HIP_CHECK(hipFuncSetAttribute((const void *)k_gemm_f16, hipFuncAttributeMaxDynamicSharedMemorySize, MMQ_LDS));
        HIP_CHECK(hipFuncSetAttribute((const void *)k_gemm_f16_b16, hipFuncAttributeMaxDynamicSharedMemorySize, MMQ_LDS));
    }
    char *dq, *dk, *dv, *dout, *dsc, *dp;
    const size_t nq = (size_t)N * E * 4, nkv = (size_t)C * Egqa * 2, nsc = (size_t)H * N * T * 4, np = (size_t)H * N * Tp * 2;
    dev_malloc((void **)&dq, nq, "debug q");
    dev_malloc((void **)&dk, nkv, "debug k");
    dev_malloc((void **)&dv, nkv, "debug v");
    dev_malloc((void **)&dout, nq, "debug out");
    dev_malloc((void **)&dsc, nsc, "debug scores");
    dev_malloc((void **)&dp, np, "debug probabilities");
    h2d_bulk(dq, q, nq);
    h2d_bulk(dk, mem_k, nkv);
    h2d_bulk(dv, mem_v, nkv);
    HIP_CHECK(hipMemsetAsync(dout, 0xFF, nq, g.stream));
    prompt_attention(fused != 0, (const float *)dq, (const __half *)dk, (const __half *)dv, (float *)dout, (float *)dsc, (_Float16 *)dp, N, E,
                     Egqa, H, Hkv, D, n_past, C, scale);
    d2h_queue(out, dout, nq);
    d2h_finish();
    for (char *b : {dq, dk, dv, dout, dsc, dp}) HIP_CHECK(hipFree(b));
    return 0;
}

// Test hook: one weight matrix times N (2..8) activation rows through k_mmq_cols exactly as the multi-token plan launches it
// (k_quant_row with its [block][8] tables, then the EPI_STORE launch).  w: a quantized 2-D weight with a device copy;
// x: host [N][K] f32; out: host [N][M] f32.  Returns 0, or -1 when the plan would not take this shape on k_mmq_cols.
int ggml_hip_debug_mul_mat_cols(const struct ggml_tensor *w, const float *x, float *out, int N) {
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    ensure_init();
    finish_pending();
    const int qt = qt_of(w->type);
    if (qt < 0 || N < 2 || N > 8) return -1;
    const QWeight qw = qweight_of(w);
    const int64_t K = w->ne[0], M = w->ne[1], nb = K / 32;
    if (!cols_ok((int)M, 1, nb, {M})) return -1;
    const bool f16d = qt == QT_Q4_0 || qt == QT_Q5_0 || qt == QT_Q8_0;
    char *dx, *dlo, *dhi, *dd, *ds, *ddT, *dsT, *dout;
    dev_malloc((void **)&dx, (size_t)N * K * 4, "debug x");
    dev_malloc((void **)&dlo, (size_t)N * K / 2, "debug lo");
    dev_malloc((void **)&dhi, (size_t)N * K / 2, "debug hi");
    dev_malloc((void **)&dd, (size_t)N * nb * 4, "debug d");
    dev_malloc((void **)&ds, (size_t)N * nb * 4, "debug s");
    dev_malloc((void **)&ddT, (size_t)nb * 32, "debug dT");
    dev_malloc((void **)&dsT, (size_t)nb * 32, "debug sT");
    dev_malloc((void **)&dout, (size_t)N * M * 4, "debug out");
    h2d_bulk(dx, x, (size_t)N * K * 4);
    HIP_CHECK(hipMemsetAsync(ddT, 0, (size_t)nb * 32, g.stream));
    HIP_CHECK(hipMemsetAsync(dsT, 0, (size_t)nb * 32, g.stream));
    HIP_CHECK(hipMemsetAsync(dout, 0xFF, (size_t)N * M * 4, g.stream));
    const dim3 grid((unsigned)((nb * 32 + 255) / 256), (unsigned)N);
    if (f16d)
        hipLaunchKernelGGL(k_quant_row<true>, grid, dim3(256), 0, g.stream, (const float *)dx, (int)nb, (int8_t *)dlo, (int8_t *)dhi,
                           (float *)dd, (int *)ds, (float *)ddT, (int *)dsT);
    else
        hipLaunchKernelGGL(k_quant_row<false>, grid, dim3(256), 0, g.stream, (const float *)dx, (int)nb, (int8_t *)dlo, (int8_t *)dhi,
                           (float *)dd, (int *)ds, (float *)ddT, (int *)dsT);
    HIP_CHECK(hipGetLastError());
    ColsArgs c;
    memset(&c, 0, sizeof(c));
    c.d.w[0] = qw;
    c.d.x = QAct{(const i32x4 *)dlo, (const i32x4 *)dhi, (const float *)dd, (const int *)ds};
    c.d.nb = nb;
    c.d.dst = (float *)dout;
    c.ncols = N;
    c.ldd = M;
    c.ldr = M;
    c.dxT = (const float *)ddT;
    c.sxT = (const int *)dsT;
    launch_cols_t<EPI_STORE>(qt, c, (int)M);
    d2h_queue(out, dout, (size_t)N * M * 4);
    d2h_finish();
    for (char *b : {dx, dlo, dhi, dd, ds, ddT, dsT, dout}) HIP_CHECK(hipFree(b));
    return 0;
}

int ggml_hip_decode_greedy_chain(struct ggml_cgraph *last, int n, int32_t *out_tokens, float *last_logits) {
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    return decode_greedy_chain(last, n, out_tokens, last_logits);
}

int64_t ggml_hip_get_stat(const char *key) {
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    const std::string k(key);
    if (k == "attn_split_tokens") return (int64_t)g.stat_split_tokens;  // tokens whose attention ran split over positions
    if (k == "w16_bytes") return (int64_t)g.w16_bytes;  // HBM held by resident f16 weight copies
    if (k.rfind("mmq_launches_", 0) == 0) {                 // prompt-GEMM launches by kernel since library load
        static const char *names[Backend::MMQ_K_COUNT] = {"plain", "dma", "dma_p", "dma_p8", "w16_p8", "w16_256", "i8"};
        for (int i = 0; i < Backend::MMQ_K_COUNT; i++)
            if (k.substr(13) == names[i]) return (int64_t)g.stat_mmq[i];
        return -1;
    }
    if (k == "prompt_plan_tokens") return (int64_t)g.stat_prompt_plan_tokens;  // tokens executed by the fused prompt plan
    if (k == "plan_tokens") return (int64_t)g.stat_plan_tokens;       // tokens executed by the fused decode plan
    if (k == "graph_replays") {
        int64_t n = 0;
        for (auto *p : g_plans) n += (int64_t)p->replays;
        return n;
    }
    if (k == "plans") return (int64_t)g_plans.size();
    if (k == "generic_graphs") return (int64_t)g.stat_generic_graphs;  // graphs run node by node
    if (k == "ns_match") return (int64_t)g.ns_match;      // host ns spent recognising decode graphs
    if (k == "ns_launch") return (int64_t)g.ns_launch;    // ... enqueueing (param upload, graph launch, read-back queue)
    if (k == "ns_wait") return (int64_t)g.ns_wait;        // ... waiting for the device + copying results out
    if (k == "ns_compute") return (int64_t)g.ns_compute;  // total inside ggml_graph_compute
    return -1;
}
size_t ggml_hip_read_timeline(int64_t *dst, size_t max_records) {
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    if (!g.timeline) return 0;
    const size_t n = std::min(max_records, g.timeline_bytes / 64);
    HIP_CHECK(hipStreamSynchronize(g.stream));
    HIP_CHECK(hipMemcpy(dst, g.timeline, n * 64, hipMemcpyDeviceToHost));
    return n;
}
const char *ggml_hip_version(void) { return "libggml_hip 0.1 (gfx950)"; }

}  // extern "C"
