// rt.hpp — thin runtime layer under the host code: HIP on the product build; for the CPU test build
// (GAZ_HOST_EMU, tests/emu only) the same calls map to malloc/memcpy and a kernel launch becomes a loop
// over blocks with a one-lane wave.
#pragma once
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include "wave.hpp"

#ifdef GAZ_HOST_EMU
typedef int hipError_t;
typedef int hipStream_t;
typedef int hipEvent_t;
#define hipSuccess 0
inline const char* hipGetErrorString(hipError_t) { return "emu"; }
inline hipError_t hipSetDevice(int) { return 0; }
inline hipError_t hipMalloc(void** p, size_t n) { *p = calloc(n ? n : 1, 1); return *p ? 0 : 1; }
inline hipError_t hipFree(void* p) { free(p); return 0; }
inline hipError_t hipMemset(void* p, int v, size_t n) { memset(p, v, n); return 0; }
inline hipError_t hipMemsetAsync(void* p, int v, size_t n, hipStream_t) { memset(p, v, n); return 0; }
enum hipMemcpyKind { hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice };
inline hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { memcpy(d, s, n); return 0; }
inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t) { memcpy(d, s, n); return 0; }
inline hipError_t hipStreamCreate(hipStream_t* s) { *s = 0; return 0; }
inline hipError_t hipStreamDestroy(hipStream_t) { return 0; }
inline hipError_t hipStreamSynchronize(hipStream_t) { return 0; }
inline hipError_t hipDeviceSynchronize() { return 0; }
inline hipError_t hipGetLastError() { return 0; }
inline hipError_t hipEventCreate(hipEvent_t* e) { *e = 0; return 0; }
inline hipError_t hipEventDestroy(hipEvent_t) { return 0; }
inline hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return 0; }
inline hipError_t hipEventSynchronize(hipEvent_t) { return 0; }
#define hipEventDisableTiming 0
inline hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { *e = 0; return 0; }
inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return 0; }
#define hipStreamNonBlocking 0
inline hipError_t hipStreamCreateWithPriority(hipStream_t* s, unsigned, int) { *s = 0; return 0; }
inline hipError_t hipDeviceGetStreamPriorityRange(int* lo, int* hi) { *lo = 0; *hi = 0; return 0; }
inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t, hipEvent_t) { *ms = 0.f; return 0; }
#define GAZ_LAUNCH(kernel, grid, block, stream, ...)                          \
    do { for (int _b = 0; _b < (int)(grid); ++_b) { gaz::emu_block_id = _b; kernel(__VA_ARGS__); } } while (0)
#define GAZ_EMU_THREADS 1
#else
#include <hip/hip_runtime.h>
#define GAZ_LAUNCH(kernel, grid, block, stream, ...) hipLaunchKernelGGL(kernel, dim3(grid), dim3(block), 0, stream, __VA_ARGS__)
#endif
