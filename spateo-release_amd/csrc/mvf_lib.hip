// Library-level entry points: error channel, version, device probe.
#include "mvf_common.h"

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

}  // namespace mvf

extern "C" const char* mvf_last_error(void) { return mvf::err_buf(); }

extern "C" int mvf_version(void) { return 2; }

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
