// A host-memory stand-in for the few HIP runtime entry points libfemfct's HOST code uses, so that the host side (argument
// checks, workspace sizing, graph keys, log buffers, budget logic) can run under AddressSanitizer on a box without a GPU:
//   make -C tools/asan && tools/asan/run_host_asan
// "Device" memory is host memory with ASan redzones, copies are memcpy (size errors trip the sanitizer), kernels are not
// run (their launches only validate that the argument array is readable), graph capture records nothing.  Test
// infrastructure: the product never links this.
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>
#include <sanitizer/common_interface_defs.h>

extern "C" {
hipError_t hipGetDeviceCount(int* c) { *c = 1; return hipSuccess; }
hipError_t hipSetDevice(int) { return hipSuccess; }
hipError_t hipDeviceGetAttribute(int* v, hipDeviceAttribute_t, int) { *v = 256; return hipSuccess; }
hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { *s = (hipStream_t)malloc(8); return hipSuccess; }
hipError_t hipStreamDestroy(hipStream_t s) { free(s); return hipSuccess; }
hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
hipError_t hipMalloc(void** p, size_t n) { *p = calloc(n ? n : 1, 1); return *p ? hipSuccess : hipErrorOutOfMemory; }
hipError_t hipFree(void* p) { free(p); return hipSuccess; }
hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { memmove(d, s, n); return hipSuccess; }
hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t) { memmove(d, s, n); return hipSuccess; }
hipError_t hipMemset(void* d, int v, size_t n) { memset(d, v, n); return hipSuccess; }
hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t) { memset(d, v, n); return hipSuccess; }
const char* hipGetErrorString(hipError_t) { return "fake hip"; }
hipError_t hipGetLastError() { return hipSuccess; }
hipError_t hipFuncSetAttribute(const void*, hipFuncAttribute, int) { return hipSuccess; }
hipError_t hipEventCreate(hipEvent_t* e) { *e = (hipEvent_t)malloc(8); return hipSuccess; }
hipError_t hipEventDestroy(hipEvent_t e) { free(e); return hipSuccess; }
hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }
hipError_t hipEventElapsedTime(float* ms, hipEvent_t, hipEvent_t) { *ms = 0.001f; return hipSuccess; }
hipError_t hipStreamBeginCapture(hipStream_t, hipStreamCaptureMode) { return hipSuccess; }
hipError_t hipStreamEndCapture(hipStream_t, hipGraph_t* g) { *g = (hipGraph_t)malloc(8); return hipSuccess; }
hipError_t hipGraphInstantiate(hipGraphExec_t* e, hipGraph_t, hipGraphNode_t*, char*, size_t) { *e = (hipGraphExec_t)malloc(8); return hipSuccess; }
hipError_t hipGraphDestroy(hipGraph_t g) { free(g); return hipSuccess; }
hipError_t hipGraphExecDestroy(hipGraphExec_t e) { free(e); return hipSuccess; }
hipError_t hipGraphLaunch(hipGraphExec_t, hipStream_t) { return hipSuccess; }
// hipLaunchKernelGGL -> hipLaunchKernel(function_address, grid, block, args, shared, stream): touch every argument slot
hipError_t hipLaunchKernel(const void* f, dim3 g, dim3 b, void** args, size_t, hipStream_t) {
    if (!f || g.x == 0 || g.y == 0 || g.z == 0 || b.x == 0 || b.x * b.y * b.z > 1024) {
        fprintf(stderr, "fake hip: bad launch geometry grid (%u, %u, %u) block (%u, %u, %u)\n", g.x, g.y, g.z, b.x, b.y, b.z);
        __sanitizer_print_stack_trace();
        return hipErrorInvalidConfiguration;
    }
    (void)args;
    return hipSuccess;
}
// registration stubs emitted by hipcc for every translation unit that defines kernels
void** __hipRegisterFatBinary(const void*) { static void* h; return &h; }
void __hipRegisterFunction(void**, const void*, char*, const char*, unsigned, void*, void*, void*, void*, int*) {}
void __hipRegisterVar(void**, void*, char*, const char*, int, size_t, int, int) {}
void __hipUnregisterFatBinary(void**) {}
static thread_local dim3 cfg_grid, cfg_block;
static thread_local size_t cfg_shared;
static thread_local hipStream_t cfg_stream;
hipError_t __hipPushCallConfiguration(dim3 g, dim3 b, size_t sh, hipStream_t st) { cfg_grid = g; cfg_block = b; cfg_shared = sh; cfg_stream = st; return hipSuccess; }
hipError_t __hipPopCallConfiguration(dim3* g, dim3* b, size_t* sh, hipStream_t* st) { *g = cfg_grid; *b = cfg_block; *sh = cfg_shared; *st = cfg_stream; return hipSuccess; }
}
