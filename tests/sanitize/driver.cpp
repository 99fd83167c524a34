// driver.cpp — the workload of the sanitizer job (tests/test_sanitize.py): the call sequences of the reference's user
// (llm::load -> start_session -> feed_prompt -> infer_next_token -> rewind -> snapshot, crates/llm-base/src/
// inference_session.rs) and of its ggml wrapper (contexts, scratch buffers, views, graph build / plan, quantizers,
// crates/ggml/src/{context,tensor,lib}.rs) through the product's two host translation units, compiled with ASan + UBSan and
// linked against tests/sanitize/stub_backend.cpp.  Also feeds the container reader truncated and corrupted files: it must
// return NULL, never read out of the mapping.   usage: driver <model.ggjt>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <thread>
#include <vector>

#include "ggml_hip.h"
#include "host/llm_host.h"

#define CHECK(c)                                                              \
    do {                                                                      \
        if (!(c)) {                                                           \
            fprintf(stderr, "driver: CHECK failed at line %d: %s\n", __LINE__, #c); \
            exit(2);                                                          \
        }                                                                     \
    } while (0)

static std::vector<unsigned char> slurp(const char *path) {
    FILE *f = fopen(path, "rb");
    CHECK(f);
    fseek(f, 0, SEEK_END);
    long n = ftell(f);
    fseek(f, 0, SEEK_SET);
    std::vector<unsigned char> b((size_t)n);
    CHECK(fread(b.data(), 1, (size_t)n, f) == (size_t)n);
    fclose(f);
    return b;
}
static void spit(const std::string &path, const unsigned char *p, size_t n) {
    FILE *f = fopen(path.c_str(), "wb");
    CHECK(f);
    if (n) CHECK(fwrite(p, 1, n, f) == n);
    fclose(f);
}

static void session_walk(llm_model *m, int seed) {
    llm_session_config cfg = {GGML_TYPE_F16, GGML_TYPE_F16, 8, 4};
    llm_session *s = llm_start_session(m, &cfg);
    CHECK(s);
    const int V = llm_model_n_vocab(m);
    std::vector<int32_t> toks(13);
    for (size_t i = 0; i < toks.size(); i++) toks[i] = (int32_t)((i * 37 + seed) % V);
    llm_feed_prompt(m, s, toks.data(), (int)toks.size());  // chunks of 8 + 5
    CHECK(llm_session_n_past(s) == 13);
    for (int i = 0; i < 6; i++) {
        const int32_t t = llm_infer_next_token_greedy(m, s);
        CHECK(t >= 0 && t < V);
    }
    CHECK(llm_session_n_past(s) == 19);
    std::vector<float> all(3 * (size_t)V), emb(4096);
    llm_evaluate(m, s, toks.data(), 3, all.data(), emb.data());
    int nn = 0, nl = 0;
    llm_session_last_graph_stats(s, &nn, &nl);
    CHECK(nn > 40 && nl > 0);
    std::vector<float> tv(8 + 2);
    std::vector<int32_t> ti(8);
    const int32_t extra[2] = {0, V - 1};
    CHECK(llm_session_topk(s, 8, extra, 2, tv.data(), ti.data()) == 0);
    CHECK(llm_session_rewind(s, 3) == 0);
    CHECK(llm_session_n_past(s) == 19);
    // K/V memory out and back in
    for (int which = 0; which < 2; which++) {
        const size_t n = llm_session_kv(s, which, 0, nullptr, 0);
        std::vector<unsigned char> kv(n);
        CHECK(llm_session_kv(s, which, 0, kv.data(), n) == n);
        CHECK(llm_session_kv(s, which, 1, kv.data(), n) == n);
    }
    // snapshot -> a second session from it -> both continue
    const size_t sn = llm_session_snapshot(s, nullptr, 0);
    std::vector<unsigned char> snap(sn);
    CHECK(llm_session_snapshot(s, snap.data(), sn) == sn);
    llm_session *s2 = llm_session_from_snapshot(m, snap.data(), sn);
    CHECK(s2 && llm_session_n_past(s2) == 19);
    CHECK(llm_session_from_snapshot(m, snap.data(), sn / 2) == nullptr);  // a truncated snapshot is refused
    std::vector<int32_t> out(5);
    CHECK(llm_infer_tokens_greedy_device(m, s2, 5, out.data()) == 5);
    (void)llm_infer_next_token_greedy(m, s);
    CHECK(llm_session_last_logits(s) != nullptr);
    std::vector<float> node(1 << 16);
    (void)llm_session_read_node(s, 0, nullptr, 0, node.data(), node.size() * 4);
    // round-5 entry points: the reference's own call order per token (no graph built ahead), a session on a named device slot,
    // a seek behind K/V the caller has put in place, the host mirror of a node
    llm_session_set_speculate(s, 0);
    for (int i = 0; i < 3; i++) (void)llm_infer_next_token_greedy(m, s);
    llm_session_set_speculate(s, 1);
    (void)llm_infer_next_token_greedy(m, s);
    (void)llm_session_read_node_host(s, 1, node.data(), node.size() * 4);
    (void)llm_session_read_node_host(s, 1 << 20, node.data(), 16);  // far beyond the graph: nothing copied
    llm_session *s3 = llm_start_session_on(m, &cfg, 0);
    CHECK(s3);
    llm_feed_prompt(m, s3, toks.data(), 5);
    llm_session_seek(s3, 9);
    CHECK(llm_session_n_past(s3) == 9);
    (void)llm_infer_next_token_greedy(m, s3);
    llm_session_free(s3);
    llm_session_free(s2);
    llm_session_free(s);
}

static void ggml_walk() {
    // a context over a caller-owned buffer, a scratch buffer, every view / permute flavour the LLaMA graph uses, graph build
    // + plan (crates/ggml/src/context.rs:131-191, 277-590; lib.rs:332-378)
    std::vector<unsigned char> buf(8u << 20), scratch(4u << 20);
    ggml_init_params ip = {buf.size(), buf.data(), false};
    ggml_context *ctx = ggml_init(ip);
    CHECK(ctx);
    ggml_tensor *a = ggml_new_tensor_2d(ctx, GGML_TYPE_F32, 64, 8);
    ggml_tensor *w = ggml_new_tensor_2d(ctx, GGML_TYPE_Q4_0, 64, 32);
    ggml_set_name(a, "a");
    CHECK(!strcmp(ggml_get_name(a), "a"));
    ggml_scratch sc = {0, scratch.size(), scratch.data()};
    (void)ggml_set_scratch(ctx, sc);
    ggml_tensor *n1 = ggml_mul(ctx, ggml_rms_norm(ctx, a, 5e-6f), ggml_new_tensor_1d(ctx, GGML_TYPE_F32, 64));
    ggml_tensor *mm = ggml_mul_mat(ctx, w, n1);                               // [32, 8]
    ggml_tensor *r3 = ggml_reshape_3d(ctx, mm, 8, 4, 8);
    ggml_tensor *rp = ggml_rope_inplace(ctx, r3, 3, 8, 0, 0);
    ggml_tensor *pm = ggml_permute(ctx, rp, 0, 2, 1, 3);
    ggml_tensor *kv = ggml_new_tensor_1d(ctx, GGML_TYPE_F16, 32 * 64);
    ggml_tensor *v1 = ggml_view_1d(ctx, kv, 32 * 8, 3 * 32 * 2);
    ggml_tensor *cp = ggml_cpy(ctx, rp, v1);
    ggml_tensor *v2 = ggml_view_2d(ctx, kv, 8, 32, 64 * 2, 16);
    ggml_tensor *v3 = ggml_view_3d(ctx, kv, 8, 4, 8, 16, 64, 0);
    ggml_tensor *tr = ggml_transpose(ctx, ggml_reshape_2d(ctx, mm, 32, 8));
    ggml_tensor *sm = ggml_soft_max_inplace(ctx, ggml_diag_mask_inf_inplace(ctx, ggml_scale_inplace(ctx, pm, ggml_new_f32(ctx, 0.5f)), 3));
    ggml_scratch none = {0, 0, nullptr};
    (void)ggml_set_scratch(ctx, none);
    ggml_tensor *sl = ggml_add(ctx, ggml_silu(ctx, mm), mm);
    CHECK(ggml_is_contiguous(mm) && !ggml_is_contiguous(tr) && ggml_nelements(v2) == 256 && ggml_nbytes(v3) > 0);
    ggml_cgraph *gr = ggml_new_graph(ctx);
    ggml_build_forward_expand(gr, cp);
    ggml_build_forward_expand(gr, sm);
    ggml_build_forward_expand(gr, sl);
    ggml_cplan plan = ggml_graph_plan(gr, 4);
    std::vector<unsigned char> work(plan.work_size + 1);
    plan.work_data = work.data();
    (void)ggml_graph_compute(gr, &plan);
    CHECK(ggml_used_mem(ctx) > 0 && ggml_graph_overhead() > 0);
    ggml_free(ctx);
    // a no_alloc context whose tensors point into caller memory (the mmap loader's form, context.rs:131-159)
    ggml_init_params ip2 = {1u << 20, nullptr, true};
    ggml_context *c2 = ggml_init(ip2);
    ggml_tensor *t = ggml_new_tensor_2d(c2, GGML_TYPE_Q5_1, 64, 4);
    std::vector<unsigned char> blocks(ggml_nbytes(t));
    t->data = blocks.data();
    CHECK(ggml_nbytes(t) == 4 * 2 * 24);
    ggml_free(c2);
    // quantizers: rows of 64 … 256 values through every block format, histogram included (crates/ggml/src/lib.rs:419-483)
    std::vector<float> src(4 * 256);
    for (size_t i = 0; i < src.size(); i++) src[i] = 0.01f * (float)((int)(i * 2654435761u % 2001) - 1000);
    for (int type : {GGML_TYPE_Q4_0, GGML_TYPE_Q4_1, GGML_TYPE_Q5_0, GGML_TYPE_Q5_1, GGML_TYPE_Q8_0}) {
        std::vector<unsigned char> dst((size_t)(ggml_type_sizef((ggml_type)type) * src.size()) + 64);
        int64_t hist[16] = {0};
        const size_t n = ggml_quantize_chunk((ggml_type)type, src.data(), dst.data(), 0, (int)src.size(), hist);
        CHECK(n == src.size() / ggml_blck_size((ggml_type)type) * ggml_type_size((ggml_type)type));
    }
    CHECK(ggml_fp16_to_fp32(ggml_fp32_to_fp16(0.5f)) == 0.5f);
}

int main(int argc, char **argv) {
    CHECK(argc == 2);
    const char *path = argv[1];
    // ---- the container reader on the good file
    llm_ggml_file *f = llm_ggml_file_open(path);
    CHECK(f);
    int container = -1, version = -1, n_tensors = 0, n_vocab = 0;
    llm_llama_hparams hp;
    llm_ggml_file_info(f, &container, &version, &hp, &n_tensors, &n_vocab);
    CHECK(container == 2 && version == 3 && n_tensors > 0 && n_vocab == hp.n_vocab);
    for (int i = 0; i < n_tensors; i++) {
        llm_tensor_desc d;
        CHECK(llm_ggml_file_tensor(f, i, &d) == 0);
        volatile unsigned char acc = ((const unsigned char *)d.data)[0];  // first byte of the mapped tensor data
        (void)acc;
    }
    llm_tensor_desc dd;
    CHECK(llm_ggml_file_tensor(f, n_tensors, &dd) != 0 && llm_ggml_file_tensor(f, -1, &dd) != 0);
    char tok[64];
    float score = 0;
    for (int i = 0; i < n_vocab; i++) CHECK(llm_ggml_file_vocab(f, i, tok, (int)sizeof tok, &score) >= 0);
    (void)llm_ggml_file_vocab(f, 0, tok, 1, nullptr);  // a 1-byte buffer
    llm_ggml_file_close(f);
    // ---- truncated and corrupted copies: NULL, no out-of-bounds read
    const std::vector<unsigned char> good = slurp(path);
    const std::string tmp = std::string(path) + ".bad";
    const size_t cuts[] = {0, 3, 4, 7, 8, 20, 35, 36, 40, 64, 300, good.size() / 3, good.size() / 2, good.size() - 33, good.size() - 1};
    for (size_t c : cuts) {
        if (c >= good.size()) continue;
        spit(tmp, good.data(), c);
        llm_ggml_file *b = llm_ggml_file_open(tmp.c_str());
        if (b) llm_ggml_file_close(b);  // a cut that happens to end on a tensor boundary may still parse
        llm_model_params mp = {32, 1, -1, 0, 1.0f, 10000, 0, -1};
        llm_model *bm = llm_llama_load(tmp.c_str(), &mp);
        if (bm) llm_model_free(bm);
    }
    for (size_t pos : {(size_t)0, (size_t)4, (size_t)8, (size_t)12, (size_t)36, (size_t)37, (size_t)44, good.size() / 4, good.size() / 2}) {
        if (pos + 4 > good.size()) continue;
        std::vector<unsigned char> bad = good;
        bad[pos] = 0xFF; bad[pos + 1] = 0xFF; bad[pos + 2] = 0xFF; bad[pos + 3] = 0x7F;  // a huge length / count / dimension
        spit(tmp, bad.data(), bad.size());
        llm_ggml_file *b = llm_ggml_file_open(tmp.c_str());
        if (b) llm_ggml_file_close(b);
        llm_model_params mp = {32, 1, -1, 0, 1.0f, 10000, 0, -1};
        llm_model *bm = llm_llama_load(tmp.c_str(), &mp);
        if (bm) llm_model_free(bm);
    }
    remove(tmp.c_str());
    CHECK(llm_ggml_file_open("/nonexistent/file") == nullptr);
    // ---- llm::load + sessions: one, then two on two threads over the same model (Send + Sync model, Send sessions)
    llm_model_params mp = {64, 1, -1, 0, 1.0f, 10000, 0, -1};
    llm_model *m = llm_llama_load(path, &mp);
    CHECK(m && llm_model_n_vocab(m) == hp.n_vocab);
    int lb[4], le[4], dev[4];
    CHECK(llm_model_stages(m, lb, le, dev, 4) == 1);
    session_walk(m, 1);
    std::thread t1(session_walk, m, 2), t2(session_walk, m, 3);
    t1.join();
    t2.join();
    llm_model_free(m);
    // ---- with RoPE overrides and a partial offload (gpu_layers = 1)
    llm_model_params mp2 = {48, 1, 1, 1, 0.5f, 20000, 0, -1};
    llm_model *m2 = llm_llama_load(path, &mp2);
    CHECK(m2);
    session_walk(m2, 4);
    llm_model_free(m2);
    // ---- helpers
    int bounds[9];
    const float shares[3] = {1.0f, 2.0f, 1.0f};
    llm_split_layers(10, 3, shares, bounds);
    CHECK(bounds[0] == 0 && bounds[3] == 10);
    llm_split_layers(2, 8, nullptr, bounds);  // more slots than layers
    const float lg[7] = {0.1f, NAN, 0.7f, 0.7f, -1.0f, 0.2f, 0.69f};
    CHECK(llm_argmax_first(lg, 7, 0) == 2 && llm_argmax_first(lg, 7, 1) == 2);
    ggml_walk();
    printf("sanitize driver OK\n");
    return 0;
}
