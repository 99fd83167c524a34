// internal.h — private seam between the host-side ggml core (ggml_core.cpp) and the MI355X backend
// (hip_backend.hip).  Not part of the C ABI.
#pragma once
#include <stddef.h>

#include "ggml_hip.h"

extern "C" {
// Every ggml context buffer and every scratch buffer handed to ggml_set_scratch is an "arena": host
// memory whose tensors get a device mirror at the same offset inside a lazily created device shadow.
void ggml_hip_internal_register_arena(void *host_base, size_t size, int is_scratch);
void ggml_hip_internal_unregister_arena(void *host_base);
// Executes a whole cgraph on the device (called by ggml_graph_compute).
void ggml_hip_internal_graph_compute(struct ggml_cgraph *cgraph);
}
