// Library-wide state of libevstore_hip.so: error strings, the sticky index-error flag.
#include "evs_common.h"

#include <mutex>

namespace evs {

static thread_local char g_err[512] = "";

void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
}

int *index_error_flag() {
    // one flag per device, allocated on first use and never freed (process lifetime)
    static std::mutex mu;
    static int *flags[64] = {nullptr};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) {
        set_error("hipGetDevice failed (no usable GPU?)");
        return nullptr;
    }
    std::lock_guard<std::mutex> lk(mu);
    if (!flags[dev]) {
        int *p = nullptr;
        if (hipMalloc(&p, sizeof(int)) != hipSuccess || hipMemset(p, 0, sizeof(int)) != hipSuccess) {
            set_error("hipMalloc of the index-error flag failed");
            return nullptr;
        }
        flags[dev] = p;
    }
    return flags[dev];
}

const void *zero_page() {
    static std::mutex mu;
    static void *pages[64] = {nullptr};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) {
        set_error("hipGetDevice failed (no usable GPU?)");
        return nullptr;
    }
    std::lock_guard<std::mutex> lk(mu);
    if (!pages[dev]) {
        void *p = nullptr;
        if (hipMalloc(&p, 4096) != hipSuccess || hipMemset(p, 0, 4096) != hipSuccess) {
            set_error("hipMalloc of the zero page failed");
            return nullptr;
        }
        pages[dev] = p;
    }
    return pages[dev];
}

const void *zero_code_page(int codec) {
    static std::mutex mu;
    static void *pages[64][3] = {};
    const int k = codec == 16 ? 0 : codec == 8 ? 1 : codec == 4 ? 2 : -1;
    int dev = 0;
    if (k < 0 || hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) {
        set_error("zero_code_page: codec %d / no usable GPU", codec);
        return nullptr;
    }
    std::lock_guard<std::mutex> lk(mu);
    if (!pages[dev][k]) {
        void *p = nullptr;
        // u16: 32 500 = 0x7ef4 -> w = 0 -> fma(0, ch, 0) = +0; u8: 127 / 254 * 2 - 1 = 0 exactly; u4: table entry 7 = 0.0
        hipError_t e = hipMalloc(&p, 1024);
        if (e == hipSuccess) e = k == 0 ? hipMemsetD16(p, 0x7ef4, 512) : hipMemset(p, k == 1 ? 0x7f : 0x77, 1024);
        if (e == hipSuccess) e = hipDeviceSynchronize();
        if (e != hipSuccess) {
            set_error("hipMalloc of the zero-code page failed");
            return nullptr;
        }
        pages[dev][k] = p;
    }
    return pages[dev][k];
}

// Flag slot of one optimistic launch pair (evs_fused.hip): a ring of 1024 device ints per device and a
// process-wide id counter.  The bag-1 kernel writes the pair's id into its slot when the bet is lost; the
// general kernel runs only if it finds its id there.  Nothing is ever reset: an id is used once.
int *optimistic_slot(int *id_out) {
    static std::mutex mu;
    static int *rings[64] = {nullptr};
    static int next_id = 1;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) {
        set_error("hipGetDevice failed (no usable GPU?)");
        return nullptr;
    }
    std::lock_guard<std::mutex> lk(mu);
    if (!rings[dev]) {
        void *p = nullptr;
        if (hipMalloc(&p, 1024 * sizeof(int)) != hipSuccess || hipMemset(p, 0, 1024 * sizeof(int)) != hipSuccess) {
            set_error("hipMalloc of the optimistic-launch flags failed");
            return nullptr;
        }
        rings[dev] = static_cast<int *>(p);
    }
    if (next_id >= 0x7ffffff0) next_id = 1;
    const int id = next_id++;
    *id_out = id;
    return rings[dev] + (id & 1023);
}

}  // namespace evs

extern "C" int evs_abi_version(void) { return EVS_ABI_VERSION; }
extern "C" const char *evs_last_error(void) { return evs::g_err; }

extern "C" void *evs_host_device_pointer(void *host_ptr) {
    void *dev = nullptr;
    if (!host_ptr || hipHostGetDevicePointer(&dev, host_ptr, 0) != hipSuccess) {
        (void)hipGetLastError();
        return nullptr;
    }
    return dev;
}
