// Error plumbing and version of the C ABI.
#include "common.h"
#include <string.h>

static thread_local char g_err[512] = "";

void bg_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* bg_last_error(void) { return g_err; }
extern "C" int bg_abi_version(void) { return BG_ABI_VERSION; }

// Dedicated HIP streams for the host mirror's side work (weight gradients, the early generator forward, gradient
// all-reduce).  torch.cuda.Stream() hands streams out of a pool of 32 per device, round-robin: two "different" side
// streams of this package could be the same HIP stream, and a forked capture stream that waits on itself sends
// hip::Stream::EndCapture into unbounded recursion (round 3: a segmentation fault in the whole-step graph capture
// that depended on how many trainers earlier tests had created).  Streams made here are never shared.
extern "C" int bg_stream_create(void** out) {
    BG_CHECK_ARG(out != nullptr, "bg_stream_create: null output");
    hipStream_t s = nullptr;
    const hipError_t e = hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    if (e != hipSuccess) {
        bg_set_error("bg_stream_create: %s", hipGetErrorString(e));
        return BG_E_LAUNCH;
    }
    *out = (void*)s;
    return BG_OK;
}

extern "C" int bg_stream_destroy(void* stream) {
    if (stream && hipStreamDestroy((hipStream_t)stream) != hipSuccess) {
        bg_set_error("bg_stream_destroy: failed");
        return BG_E_LAUNCH;
    }
    return BG_OK;
}
