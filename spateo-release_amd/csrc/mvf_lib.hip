// Library-level entry points: error channel, version, device probe.
#include "mvf_common.h"
#include <atomic>
#include <cstring>

namespace mvf {

char* err_buf() {
    static thread_local char buf[512] = {0};
    return buf;
}

int set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(err_buf(), 512, fmt, ap);
    va_end(ap);
    return 1;
}

static std::atomic<long long> g_debug[DBG_COUNT];
static const char* const g_debug_names[DBG_COUNT] = {"conk_form", "slice_len", "solve_small_off", "lr_timing", "lr_no_deflate",
                                                     "defl_block", "defl_apps", "lr_no_direct", "direct_accept"};
long long debug_opt(DebugOpt which) { return g_debug[which].load(std::memory_order_relaxed); }

// Looked up per CURRENT device (the host binding makes the launch stream's device current) and remembered per device
// index; the idempotent lazy write needs no lock.
int device_cu_count() {
    constexpr int MAXDEV = 64;
    static int cus[MAXDEV] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAXDEV) {
        (void)hipGetLastError();
        return 256;
    }
    if (cus[dev] == 0) {
        int c = 256;
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) c = prop.multiProcessorCount;
        (void)hipGetLastError();
        cus[dev] = c;
    }
    return cus[dev];
}

}  // namespace mvf

extern "C" const char* mvf_last_error(void) { return mvf::err_buf(); }

extern "C" int mvf_version(void) { return 7; }

extern "C" int mvf_debug_option(const char* name, long long value) {
    if (!name) return mvf::set_error("mvf_debug_option: null name");
    for (int i = 0; i < mvf::DBG_COUNT; ++i)
        if (std::strcmp(name, mvf::g_debug_names[i]) == 0) {
            mvf::g_debug[i].store(value, std::memory_order_relaxed);
            return 0;
        }
    return mvf::set_error("mvf_debug_option: unknown option '%s'", name);
}

extern "C" long long mvf_debug_option_get(const char* name) {
    if (name)
        for (int i = 0; i < mvf::DBG_COUNT; ++i)
            if (std::strcmp(name, mvf::g_debug_names[i]) == 0) return mvf::g_debug[i].load(std::memory_order_relaxed);
    return -1;
}

extern "C" int mvf_device_count(int* count) {
    if (!count) return mvf::set_error("mvf_device_count: null pointer");
    int c = 0;
    hipError_t e = hipGetDeviceCount(&c);
    if (e != hipSuccess) {
        (void)hipGetLastError();  // no GPU / no driver: report zero devices, not an error
        c = 0;
    }
    *count = c;
    return 0;
}

// The one host read of an EM iteration: a blocking device -> host copy of a small status block ON `stream` (what precedes it on
// the stream is complete when it returns).  hipMemcpyWithStream: 1.3 us on top of the synchronisation itself for 64 bytes on this
// part, against 9 us for hipMemcpyAsync into pageable memory + hipStreamSynchronize (tools/readback_probe.hip).
extern "C" int mvf_read_back(void* dst_host, const void* src_device, size_t nbytes, void* stream) {
    if (nbytes == 0) return 0;
    if (!dst_host || !src_device) return mvf::set_error("mvf_read_back: null pointer");
    MVF_CHECK_HIP(hipMemcpyWithStream(dst_host, src_device, nbytes, hipMemcpyDeviceToHost, (hipStream_t)stream));
    return 0;
}
