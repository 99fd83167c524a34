// ggml_wrap.hpp — C++ mirror of the reference's safe wrapper crate `crates/ggml` (the operator surface
// BASELINE.json's north_star names: ggml::Context / ggml::Tensor / ComputationGraph /
// GraphExecutionPlan / accelerator).  The reference's host code is Rust; this image has no Rust
// toolchain, so the host side above the C ABI is written in C++ with the same names, argument meaning
// and error behaviour (panics → abort with message), one method per Rust method:
//   Context            crates/ggml/src/context.rs:18-662
//   Tensor             crates/ggml/src/tensor.rs:9-267
//   Buffer             crates/ggml/src/lib.rs:283-319
//   ComputationGraph   crates/ggml/src/lib.rs:322-336
//   GraphExecutionPlan crates/ggml/src/lib.rs:338-378
//   accelerator::*     crates/ggml/src/accelerator/mod.rs:41-94
// Everything here goes through the C ABI of include/ggml_hip.h, exactly like `ggml-sys` does.
#pragma once
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <string>
#include <vector>

#include "ggml_hip.h"

namespace ggml {

[[noreturn]] inline void panic(const char *msg) {
    fprintf(stderr, "ggml (host mirror) panic: %s\n", msg);
    abort();
}

enum class Backend { Cpu, Gpu, GpuSplit };  // accelerator/mod.rs:33-42
inline ggml_backend to_sys(Backend b) {
    return b == Backend::Cpu ? GGML_BACKEND_CPU : b == Backend::Gpu ? GGML_BACKEND_GPU : GGML_BACKEND_GPU_SPLIT;
}

using Type = ggml_type;  // crates/ggml/src/lib.rs:154-281 (same discriminants)

constexpr float DEFAULT_EPS = LLAMA_DEFAULT_RMS_EPS;  // lib.rs:131-132

struct RoPEOverrides {  // lib.rs:134-152
    float frequency_scale = 1.0f;
    size_t frequency_base = 10000;
};

// lib.rs:283-319 — 16 KiB-aligned host allocation used for scratch / eval buffers
class Buffer {
   public:
    explicit Buffer(size_t size) : size_(size) {
        if (posix_memalign(&data_, 16384, size ? size : 16384) != 0) panic("Buffer allocation failed");
    }
    ~Buffer() { free(data_); }
    Buffer(const Buffer &) = delete;
    Buffer &operator=(const Buffer &) = delete;
    size_t size() const { return size_; }
    void *data() const { return data_; }

   private:
    void *data_ = nullptr;
    size_t size_;
};

namespace accelerator {  // accelerator/mod.rs:66-94 (cublas arms → this library's hooks)
inline void initialize(int device) {
    ggml_init_hipblas();
    ggml_hip_set_main_device(device);
    const float split = 1.0f;  // mod.rs:74-75: the address of one stack float; the hook reads exactly one
    ggml_hip_set_tensor_split(&split);
}
inline void set_scratch_size(size_t size) { ggml_hip_set_scratch_size(size); }
inline void free_scratch() { ggml_hip_free_scratch(); }
}  // namespace accelerator

class Context;

// tensor.rs:9-12 — a tensor is a raw pointer plus a weak reference to its context
class Tensor {
   public:
    Tensor() = default;
    Tensor(ggml_tensor *p, std::weak_ptr<struct ContextInner> inner) : ptr_(p), inner_(std::move(inner)) {}
    ggml_tensor *ptr() const { return ptr_; }
    bool is_null() const { return ptr_ == nullptr; }

    Tensor set_name(const char *name);                 // tensor.rs:25-35
    std::string name() const;                          // :38-44
    Backend backend() const;                           // :47-53
    Tensor transfer_to(Backend backend);               // :56-80
    void offload() const;                              // :87-94
    void offload_no_scratch() const;                   // :104-112
    Tensor share() const { return *this; }             // :115-120
    size_t nbytes() const;                             // :123-128
    void *data() const;                                // :136-141
    void set_data(void *p);                            // :149-155
    size_t nelements() const;                          // :158-163
    size_t element_size() const;                       // :181-183
    Type get_type() const { alive(); return ptr_->type; }
    void write_data(const void *src, size_t n);        // :191-193
    void zero_data();                                  // :196-198
    void read_data(size_t offset, void *dst, size_t n) const;  // :206-209
    bool is_contiguous() const;                        // :225-227
    void free_accelerator();                           // :213-222

   private:
    friend class Context;
    void alive() const;
    void mark_as_offloaded() const;                    // :258-266
    ggml_tensor *ptr_ = nullptr;
    std::weak_ptr<struct ContextInner> inner_;
};

struct ContextInner {  // context.rs:33-50
    ggml_context *ptr = nullptr;
    std::map<std::string, Tensor> offloaded_tensors;
};

// lib.rs:322-336
class ComputationGraph {
   public:
    explicit ComputationGraph(ggml_cgraph *raw) : inner_(raw) {}
    void build_forward_expand(const Tensor &t) { ggml_build_forward_expand(inner_, t.ptr()); }
    ggml_cgraph *raw() const { return inner_; }

   private:
    ggml_cgraph *inner_;
};

class Context {
   public:
    enum class Storage { BufferStorage, Allocate, NoAllocMmap };

    // context.rs:131-176
    static Context new_with_buffer(std::shared_ptr<Buffer> buffer) {
        Context c;
        c.storage_ = Storage::BufferStorage;
        c.buffer_ = std::move(buffer);
        c.init();
        return c;
    }
    static Context new_with_allocate(size_t mem_size) {
        Context c;
        c.storage_ = Storage::Allocate;
        c.mem_size_ = mem_size;
        c.init();
        return c;
    }
    // mmap flavour: ggml owns only the tensor headers; data pointers are patched by the loader
    static Context new_with_mmap(size_t header_bytes) {
        Context c;
        c.storage_ = Storage::NoAllocMmap;
        c.mem_size_ = header_bytes;
        c.init();
        return c;
    }
    Context() = default;
    Context(Context &&) = default;
    Context &operator=(Context &&o) {
        if (this != &o) {
            drop();
            inner_ = std::move(o.inner_);
            storage_ = o.storage_;
            buffer_ = std::move(o.buffer_);
            mem_size_ = o.mem_size_;
            can_offload = o.can_offload;
        }
        return *this;
    }
    ~Context() { drop(); }

    // context.rs:179-182 — "Recreates this context using the same storage"
    void recreate() {
        drop();
        can_offload = false;
        init();
    }

    ComputationGraph create_compute_graph() const { return ComputationGraph(ggml_new_graph(as_ptr())); }  // :184-191
    void set_offloading(bool v) { can_offload = v; }                                                      // :200-202
    size_t used_mem() const { return ggml_used_mem(as_ptr()); }                                           // :205-207
    void use_scratch(const Buffer *scratch) const {                                                       // :212-232
        ggml_scratch s;
        s.offs = 0;
        s.size = scratch ? scratch->size() : 0;
        s.data = scratch ? scratch->data() : nullptr;
        ggml_set_scratch(as_ptr(), s);
    }

    Tensor new_tensor_1d(Type t, size_t ne0) const { return raw(ggml_new_tensor_1d(as_ptr(), t, (int64_t)ne0)); }
    Tensor new_tensor_2d(Type t, size_t ne0, size_t ne1) const {
        return raw(ggml_new_tensor_2d(as_ptr(), t, (int64_t)ne0, (int64_t)ne1));
    }
    Tensor new_tensor_3d(Type t, size_t ne0, size_t ne1, size_t ne2) const {
        return raw(ggml_new_tensor_3d(as_ptr(), t, (int64_t)ne0, (int64_t)ne1, (int64_t)ne2));
    }
    Tensor new_f32(float x) const { return raw(ggml_new_f32(as_ptr(), x)); }

    // op builders, context.rs:276-626
    Tensor op_transpose(const Tensor &a) const { return raw(ggml_transpose(as_ptr(), a.ptr())); }
    Tensor op_get_rows(const Tensor &a, const Tensor &b) const { return raw(ggml_get_rows(as_ptr(), a.ptr(), b.ptr())); }
    Tensor op_norm(const Tensor &a) const { return raw(ggml_norm(as_ptr(), a.ptr())); }
    Tensor op_rms_norm(const Tensor &a) const { return raw(ggml_rms_norm(as_ptr(), a.ptr(), DEFAULT_EPS)); }
    Tensor op_mul(const Tensor &a, const Tensor &b) const { return raw(ggml_mul(as_ptr(), a.ptr(), b.ptr())); }
    Tensor op_repeat(const Tensor &a, const Tensor &b) const { return raw(ggml_repeat(as_ptr(), a.ptr(), b.ptr())); }
    Tensor op_mul_mat(const Tensor &a, const Tensor &b) const { return raw(ggml_mul_mat(as_ptr(), a.ptr(), b.ptr())); }
    Tensor op_add(const Tensor &a, const Tensor &b) const { return raw(ggml_add(as_ptr(), a.ptr(), b.ptr())); }
    Tensor op_silu(const Tensor &a) const { return raw(ggml_silu(as_ptr(), a.ptr())); }
    Tensor op_gelu(const Tensor &a) const { return raw(ggml_gelu(as_ptr(), a.ptr())); }
    Tensor op_scale(const Tensor &a, const Tensor &b) const { return raw(ggml_scale(as_ptr(), a.ptr(), b.ptr())); }
    Tensor op_scale_inplace(const Tensor &a, const Tensor &b) const {
        return raw(ggml_scale_inplace(as_ptr(), a.ptr(), b.ptr()));
    }
    Tensor op_diag_mask_inf(const Tensor &a, size_t n_past) const {
        return raw(ggml_diag_mask_inf(as_ptr(), a.ptr(), (int)n_past));
    }
    Tensor op_diag_mask_inf_inplace(const Tensor &a, size_t n_past) const {
        return raw(ggml_diag_mask_inf_inplace(as_ptr(), a.ptr(), (int)n_past));
    }
    Tensor op_soft_max(const Tensor &a) const { return raw(ggml_soft_max(as_ptr(), a.ptr())); }
    Tensor op_soft_max_inplace(const Tensor &a) const { return raw(ggml_soft_max_inplace(as_ptr(), a.ptr())); }
    Tensor op_view_1d(const Tensor &a, size_t ne0, size_t offset) const {
        return raw(ggml_view_1d(as_ptr(), a.ptr(), (int64_t)ne0, offset));
    }
    Tensor op_view_2d(const Tensor &a, size_t ne0, size_t ne1, size_t nb1, size_t offset) const {
        return raw(ggml_view_2d(as_ptr(), a.ptr(), (int64_t)ne0, (int64_t)ne1, nb1, offset));
    }
    Tensor op_view_3d(const Tensor &a, size_t ne0, size_t ne1, size_t ne2, size_t nb1, size_t nb2,
                      size_t offset) const {
        return raw(ggml_view_3d(as_ptr(), a.ptr(), (int64_t)ne0, (int64_t)ne1, (int64_t)ne2, nb1, nb2, offset));
    }
    Tensor op_cpy(const Tensor &a, const Tensor &b) const { return raw(ggml_cpy(as_ptr(), a.ptr(), b.ptr())); }
    Tensor op_permute(const Tensor &a, int a0, int a1, int a2, int a3) const {
        return raw(ggml_permute(as_ptr(), a.ptr(), a0, a1, a2, a3));
    }
    Tensor op_reshape_2d(const Tensor &a, size_t ne0, size_t ne1) const {
        return raw(ggml_reshape_2d(as_ptr(), a.ptr(), (int64_t)ne0, (int64_t)ne1));
    }
    Tensor op_reshape_3d(const Tensor &a, size_t ne0, size_t ne1, size_t ne2) const {
        return raw(ggml_reshape_3d(as_ptr(), a.ptr(), (int64_t)ne0, (int64_t)ne1, (int64_t)ne2));
    }
    Tensor op_cont(const Tensor &a) const { return raw(ggml_cont(as_ptr(), a.ptr())); }
    Tensor op_rope_inplace(const Tensor &a, size_t npast, size_t ndims, int mode,
                           const RoPEOverrides *overrides) const {  // context.rs:557-590
        if (overrides)
            return raw(ggml_rope_custom_inplace(as_ptr(), a.ptr(), (int)npast, (int)ndims, mode, 1,
                                                (float)overrides->frequency_base, overrides->frequency_scale));
        return raw(ggml_rope_inplace(as_ptr(), a.ptr(), (int)npast, (int)ndims, mode, 0));
    }

    ggml_context *as_ptr() const {
        if (!inner_) panic("Context used after drop");
        return inner_->ptr;
    }
    std::shared_ptr<ContextInner> inner() const { return inner_; }
    bool can_offload = false;  // context.rs:29-30

   private:
    void init() {  // context.rs:131-159
        ggml_init_params p;
        switch (storage_) {
            case Storage::BufferStorage:
                p.mem_size = buffer_->size();
                p.mem_buffer = buffer_->data();
                p.no_alloc = false;
                break;
            case Storage::Allocate:
                p.mem_size = mem_size_;
                p.mem_buffer = nullptr;
                p.no_alloc = false;
                break;
            case Storage::NoAllocMmap:
                p.mem_size = mem_size_;
                p.mem_buffer = nullptr;
                p.no_alloc = true;
                break;
        }
        ggml_context *raw = ggml_init(p);
        if (!raw) panic("Should not be null");
        inner_ = std::make_shared<ContextInner>();
        inner_->ptr = raw;
    }
    void drop() {  // context.rs:649-661
        if (!inner_) return;
        for (auto &kv : inner_->offloaded_tensors)
            if (kv.second.ptr()->backend != GGML_BACKEND_CPU) ggml_hip_free_data(kv.second.ptr());
        inner_->offloaded_tensors.clear();
        ggml_free(inner_->ptr);
        inner_->ptr = nullptr;
        inner_.reset();
    }
    Tensor raw(ggml_tensor *t) const {  // context.rs:636-646 new_tensor_raw
        if (!t) panic("Should not be null");
        Tensor tensor(t, inner_);
        if (can_offload) tensor.offload();
        return tensor;
    }
    std::shared_ptr<ContextInner> inner_;
    Storage storage_ = Storage::Allocate;
    std::shared_ptr<Buffer> buffer_;
    size_t mem_size_ = 0;
};

// ---- Tensor methods ----------------------------------------------------------------------------
inline void Tensor::alive() const {
    auto c = inner_.lock();
    if (!c || !c->ptr) panic("Using a tensor after the context was dropped");
}
inline Tensor Tensor::set_name(const char *name) {
    if (strlen(name) > GGML_MAX_NAME) panic("Tensor name must be less than GGML_MAX_NAME bytes");
    alive();
    ggml_set_name(ptr_, name);
    return *this;
}
inline std::string Tensor::name() const {
    alive();
    return ggml_get_name(ptr_);
}
inline Backend Tensor::backend() const {
    alive();
    switch (ptr_->backend) {
        case GGML_BACKEND_CPU: return Backend::Cpu;
        case GGML_BACKEND_GPU: return Backend::Gpu;
        default: return Backend::GpuSplit;
    }
}
inline void Tensor::mark_as_offloaded() const {
    auto c = inner_.lock();
    if (!c) panic("Attempted to update a dropped context's offloaded tensors");
    c->offloaded_tensors[name()] = *this;
}
inline Tensor Tensor::transfer_to(Backend backend) {
    alive();
    if (this->backend() != Backend::Cpu && backend == Backend::Cpu)
        panic("Tensors cannot be moved from an accelerator to the CPU at present");
    if (backend == Backend::Cpu) return *this;
    ptr_->backend = to_sys(backend);                 // set_backend, tensor.rs:251-255
    ggml_hip_transform_tensor(ptr_->data, ptr_);     // tensor.rs:68-71 — the H2D boundary
    mark_as_offloaded();
    return *this;
}
inline void Tensor::offload() const {
    alive();
    ggml_hip_assign_buffers(ptr_);
}
inline void Tensor::offload_no_scratch() const {
    alive();
    ggml_hip_assign_buffers_no_scratch(ptr_);
    mark_as_offloaded();
}
inline size_t Tensor::nbytes() const {
    alive();
    return ggml_nbytes(ptr_);
}
inline void *Tensor::data() const {
    alive();
    return ptr_->data;
}
inline void Tensor::set_data(void *p) {
    alive();
    ptr_->data = p;
}
inline size_t Tensor::nelements() const {
    alive();
    return (size_t)ggml_nelements(ptr_);
}
inline size_t Tensor::element_size() const {
    alive();
    return ggml_element_size(ptr_);
}
inline void Tensor::write_data(const void *src, size_t n) { memcpy(data(), src, n); }
inline void Tensor::zero_data() { memset(data(), 0, nbytes()); }
inline void Tensor::read_data(size_t offset, void *dst, size_t n) const {
    // tensor.rs:206-209 reads host memory; a node that lives on the device (backend Gpu) is fetched through the backend
    // (what INTEGRATION.md tells a Rust maintainer to do for offloaded tensors)
    if (ptr_->backend != GGML_BACKEND_CPU)
        ggml_hip_tensor_get(ptr_, dst, offset, n);
    else
        memcpy(dst, (const char *)ggml_get_data(ptr_) + offset, n);
}
inline bool Tensor::is_contiguous() const { return ggml_is_contiguous(ptr_); }
inline void Tensor::free_accelerator() { ggml_hip_free_data(ptr_); }

// lib.rs:338-378
class GraphExecutionPlan {
   public:
    GraphExecutionPlan(ComputationGraph &graph, size_t n_threads)
        : inner_(ggml_graph_plan(graph.raw(), (int)n_threads)), inner_graph_(graph.raw()) {}
    void execute(const Context &context) {
        Tensor work = context.new_tensor_1d(GGML_TYPE_I8, inner_.work_size);  // create_work_buffer
        inner_.work_data = (uint8_t *)work.data();                            // assign_work_buffer
        ggml_graph_compute(inner_graph_, &inner_);
    }
    // split form (hip backend extension): returns true if the graph is still running on the device
    bool execute_begin(const Context &context) {
        Tensor work = context.new_tensor_1d(GGML_TYPE_I8, inner_.work_size);
        inner_.work_data = (uint8_t *)work.data();
        return ggml_hip_graph_compute_begin(inner_graph_) != 0;
    }
    static void execute_end() { ggml_hip_graph_compute_end(); }
    // the next token's graph, built while the device runs: let the backend match it against its decode plan now (hip backend extension)
    static bool prepare(const ComputationGraph &next) { return ggml_hip_graph_prepare(next.raw()) != 0; }

   private:
    ggml_cplan inner_;
    ggml_cgraph *inner_graph_;
};

}  // namespace ggml
