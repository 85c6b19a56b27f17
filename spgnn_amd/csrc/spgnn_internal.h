// Internal helpers shared by the translation units of libspgnn_hip.so (not part of the C ABI).
#pragma once
#include <hip/hip_runtime.h>

namespace spgnn_detail {
// record the message spgnn_last_error() returns (thread-local) and hand `code` back
int fail(int code, const char* msg);
// hipGetLastError() after a launch -> SPGNN_OK or -(1000 + hipError_t) with the runtime's message recorded
int check_launch(const char* what);
// argument check failed inside `func` (source line `line`): records "<func>: <kind of error> (line N)"
int fail_at(int code, const char* func, int line);
// hipFuncAttributeMaxDynamicSharedMemorySize for kernels that need more than 64 KB of LDS; the attribute is per
// device, so it is remembered per (function, device)
int ensure_dynamic_lds(const void* func, int bytes);
}  // namespace spgnn_detail
