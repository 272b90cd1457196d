// TEST-ONLY stand-ins that let the HOST half of the C ABI (mini-flash-attention_amd/csrc/mfa_capi.cpp: argument
// validation, split-count choice, workspace sizing, kv-cache planning) link into a CPU-only shared object built with
// -fsanitize=address,undefined (tests/test_sanitizers_cpu.py).  The kernel launchers record that they were reached and
// return success; the four HIP runtime calls the host half makes report "no device".  Nothing here ships.
#include <hip/hip_runtime.h>

#include "mfa_launch.h"

extern "C" int mfa_test_launch_count = 0;

namespace mfa {
int launch_prefill(const mfa_forward_params&, hipStream_t, bool*) { ++mfa_test_launch_count; return 0; }
int launch_decode(const mfa_forward_params&, hipStream_t, bool* f) { ++mfa_test_launch_count; if (f) *f = false; return 0; }
int launch_kvcache_packed(const mfa_forward_params&, hipStream_t, bool* f) { ++mfa_test_launch_count; if (f) *f = false; return 0; }
int launch_decode_combine(const mfa_forward_params&, hipStream_t) { ++mfa_test_launch_count; return 0; }
int launch_kvcache_append(const mfa_kvcache_append_params&, hipStream_t) { ++mfa_test_launch_count; return 0; }
int xcd_premise_probe(int) { return 0; }
bool fused_merge_pays(int64_t units, int64_t workgroups, int64_t pbytes) { return !spread_splits(units, 2) && workgroups <= kFusedMergeMaxWorkgroups && pbytes <= kFusedMergeMaxPartialBytes; }
} // namespace mfa

extern "C" {
hipError_t hipGetDevice(int* d) { if (d) *d = 0; return hipErrorNoDevice; }
hipError_t hipDeviceGetAttribute(int* v, hipDeviceAttribute_t, int) { if (v) *v = 0; return hipErrorNoDevice; }
hipError_t hipGetLastError(void) { return hipSuccess; }
const char* hipGetErrorString(hipError_t) { return "stub"; }
hipError_t hipStreamIsCapturing(hipStream_t, hipStreamCaptureStatus* s) { if (s) *s = hipStreamCaptureStatusNone; return hipSuccess; }
}
