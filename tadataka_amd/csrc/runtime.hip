// runtime.hip -- device selection, stream, error string, scratch pools, the (optionally red-zoned) device allocator.
#define TDK_RUNTIME_IMPL
#include "tdk_runtime.h"

#include <stdarg.h>
#include <stdlib.h>
#include <string.h>

#include <map>
#include <mutex>
#include <string>
#include <vector>

namespace tdk {

static thread_local char g_err[512] = "";
static hipStream_t g_stream = nullptr;
static hipStream_t g_upload_stream = nullptr;   // uploads of caller-owned arrays: waited for on their own
static bool g_ready = false;
static std::mutex g_mu;

static void *g_scratch[kScratchSlots] = {};
static size_t g_scratch_bytes[kScratchSlots] = {};
static void *g_pinned[kScratchSlots] = {};
static size_t g_pinned_bytes[kScratchSlots] = {};

void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

std::recursive_mutex &api_mutex() {
    static std::recursive_mutex m;
    return m;
}

static int g_options[2] = {1, 1};   // TDK_OPT_PYRAMID_STREAM, TDK_OPT_SD_WARP_GATHER
int option(int which) { return g_options[which]; }

hipStream_t stream() { return g_stream; }
hipStream_t upload_stream() { return g_upload_stream; }

tdk_status ensure_device() {
    if (g_ready) return TDK_OK;
    std::lock_guard<std::mutex> lock(g_mu);
    if (g_ready) return TDK_OK;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        set_error("no HIP device available (%s)", e == hipSuccess ? "count = 0" : hipGetErrorString(e));
        return TDK_ERR_NO_DEVICE;
    }
    TDK_HIP(hipStreamCreateWithFlags(&g_stream, hipStreamNonBlocking));
    TDK_HIP(hipStreamCreateWithFlags(&g_upload_stream, hipStreamNonBlocking));
    g_ready = true;
    return TDK_OK;
}

// ---- device allocator ------------------------------------------------------------------------------------------
constexpr size_t kZone = 4096;
struct Allocation {
    void *base;
    size_t bytes;
    std::string name;
};
static std::map<void *, Allocation> g_allocs;     // user pointer -> allocation (canary mode only)
static std::mutex g_alloc_mu;
static bool g_violation = false;
static char g_violation_msg[400] = "";

bool canaries_enabled() {
    static const bool on = [] { const char *v = getenv("TDK_DEBUG_CANARY"); return v && atoi(v) != 0; }();
    return on;
}

hipError_t dev_malloc(void **ptr, size_t bytes, const char *what, const char *file, int line) {
    if (!canaries_enabled()) return hipMalloc(ptr, bytes);
    void *base = nullptr;
    hipError_t e = hipMalloc(&base, bytes + 2 * kZone);
    if (e != hipSuccess) return e;
    e = hipMemset(base, 0xFF, kZone);
    if (e == hipSuccess) e = hipMemset((char *)base + kZone + bytes, 0xFF, kZone);
    if (e != hipSuccess) { (void)hipFree(base); return e; }
    void *user = (char *)base + kZone;
    const char *slash = strrchr(file, '/');
    char name[256];
    snprintf(name, sizeof(name), "%s (%s:%d, %zu bytes)", what, slash ? slash + 1 : file, line, bytes);
    std::lock_guard<std::mutex> lock(g_alloc_mu);
    g_allocs[user] = Allocation{base, bytes, name};
    *ptr = user;
    return hipSuccess;
}

// both zones of one allocation; the first damaged byte is reported
static bool zones_intact(void *user, const Allocation &a) {
    std::vector<unsigned char> host(2 * kZone);
    if (hipMemcpy(host.data(), a.base, kZone, hipMemcpyDeviceToHost) != hipSuccess ||
        hipMemcpy(host.data() + kZone, (char *)user + a.bytes, kZone, hipMemcpyDeviceToHost) != hipSuccess) {
        snprintf(g_violation_msg, sizeof(g_violation_msg), "canary check: cannot read the red zones of %s", a.name.c_str());
        return false;
    }
    for (size_t i = 0; i < 2 * kZone; i++) {
        if (host[i] == 0xFF) continue;
        if (i < kZone)
            snprintf(g_violation_msg, sizeof(g_violation_msg), "canary: %zu bytes BEFORE the start of %s were overwritten",
                     kZone - i, a.name.c_str());
        else
            snprintf(g_violation_msg, sizeof(g_violation_msg), "canary: byte %zu BEHIND the end of %s was overwritten",
                     i - kZone, a.name.c_str());
        return false;
    }
    return true;
}

hipError_t dev_free(void *ptr) {
    if (!canaries_enabled() || ptr == nullptr) return hipFree(ptr);
    Allocation a;
    {
        std::lock_guard<std::mutex> lock(g_alloc_mu);
        auto it = g_allocs.find(ptr);
        if (it == g_allocs.end()) return hipFree(ptr);            // allocated before the mode was known: cannot happen
        a = it->second;
        g_allocs.erase(it);
    }
    (void)hipDeviceSynchronize();
    if (!zones_intact(ptr, a)) {
        g_violation = true;
        fprintf(stderr, "libtadataka_hip: %s\n", g_violation_msg);
    }
    return hipFree(a.base);
}

tdk_status check_canaries() {
    if (!canaries_enabled()) return TDK_OK;
    TDK_HIP(hipDeviceSynchronize());
    std::lock_guard<std::mutex> lock(g_alloc_mu);
    for (auto &kv : g_allocs) {
        if (!zones_intact(kv.first, kv.second)) {
            g_violation = true;
            break;
        }
    }
    if (g_violation) {
        set_error("%s", g_violation_msg);
        return TDK_ERR_HIP;
    }
    return TDK_OK;
}

static void (*g_release_hooks[8])() = {};
static int g_n_release_hooks = 0;

void on_device_release(void (*hook)()) {
    if (g_n_release_hooks < 8) g_release_hooks[g_n_release_hooks++] = hook;
}

static void release_pools() {
    for (int i = 0; i < g_n_release_hooks; i++) g_release_hooks[i]();
    for (int i = 0; i < kScratchSlots; i++) {
        if (g_scratch[i]) (void)dev_free(g_scratch[i]);
        if (g_pinned[i]) (void)hipHostFree(g_pinned[i]);
        g_scratch[i] = g_pinned[i] = nullptr;
        g_scratch_bytes[i] = g_pinned_bytes[i] = 0;
    }
}

tdk_status scratch(int slot, size_t bytes, void **ptr) {
    TDK_TRY(ensure_device());
    if (bytes == 0) bytes = 8;
    if (g_scratch_bytes[slot] < bytes) {
        if (g_scratch[slot]) {
            TDK_HIP(hipStreamSynchronize(g_stream));
            TDK_HIP(dev_free(g_scratch[slot]));
            g_scratch[slot] = nullptr;
            g_scratch_bytes[slot] = 0;
        }
        size_t cap = bytes + bytes / 4;
        TDK_HIP(dev_malloc(&g_scratch[slot], cap, "scratch pool slot", __FILE__, __LINE__));
        g_scratch_bytes[slot] = cap;
    }
    *ptr = g_scratch[slot];
    return TDK_OK;
}

tdk_status pinned(int slot, size_t bytes, void **ptr) {
    TDK_TRY(ensure_device());
    if (bytes == 0) bytes = 8;
    if (g_pinned_bytes[slot] < bytes) {
        if (g_pinned[slot]) {
            TDK_HIP(hipStreamSynchronize(g_stream));
            TDK_HIP(hipHostFree(g_pinned[slot]));
            g_pinned[slot] = nullptr;
            g_pinned_bytes[slot] = 0;
        }
        size_t cap = bytes + bytes / 4;
        TDK_HIP(hipHostMalloc(&g_pinned[slot], cap, hipHostMallocDefault));
        g_pinned_bytes[slot] = cap;
    }
    *ptr = g_pinned[slot];
    return TDK_OK;
}

}  // namespace tdk

extern "C" {

const char *tdk_version(void) { return "tadataka_hip 0.3 (gfx950)"; }

const char *tdk_last_error(void) { return tdk::g_err; }

tdk_status tdk_device_count(int *count) {
    TDK_REQUIRE(count != nullptr, "count is NULL");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    *count = (e == hipSuccess) ? n : 0;
    return TDK_OK;
}

tdk_status tdk_set_device(int device) {
    TDK_API_GUARD;
    // Pools and the stream belong to the current device: drop them first.
    if (tdk::g_ready) {
        TDK_HIP(hipStreamSynchronize(tdk::g_stream));
        tdk::release_pools();
        TDK_HIP(hipStreamDestroy(tdk::g_stream));
        tdk::g_stream = nullptr;
        TDK_HIP(hipStreamSynchronize(tdk::g_upload_stream));
        TDK_HIP(hipStreamDestroy(tdk::g_upload_stream));
        tdk::g_upload_stream = nullptr;
        tdk::g_ready = false;
    }
    TDK_HIP(hipSetDevice(device));
    return tdk::ensure_device();
}

tdk_status tdk_get_device(int *device) {
    TDK_API_GUARD;
    TDK_REQUIRE(device != nullptr, "device is NULL");
    TDK_HIP(hipGetDevice(device));
    return TDK_OK;
}

tdk_status tdk_sync(void) {
    TDK_API_GUARD;
    TDK_TRY(tdk::ensure_device());
    TDK_HIP(hipDeviceSynchronize());   // the library stream and every batch's own stream
    return tdk::check_canaries();      // (TDK_DEBUG_CANARY=1 only)
}

tdk_status tdk_set_option(int option, int value) {
    TDK_API_GUARD;
    TDK_REQUIRE(option == TDK_OPT_PYRAMID_STREAM || option == TDK_OPT_SD_WARP_GATHER, "unknown option");
    TDK_REQUIRE(value >= 0 && value <= (option == TDK_OPT_PYRAMID_STREAM ? 3 : 1), "value out of range");
    tdk::g_options[option] = value;
    return TDK_OK;
}

tdk_status tdk_debug_check_canaries(int *n_allocations) {
    TDK_API_GUARD;
    if (n_allocations) {
        std::lock_guard<std::mutex> lock(tdk::g_alloc_mu);
        *n_allocations = tdk::canaries_enabled() ? (int)tdk::g_allocs.size() : -1;
    }
    return tdk::check_canaries();
}

tdk_status tdk_pinned_alloc(size_t bytes, void **out) {
    TDK_API_GUARD;
    TDK_REQUIRE(out != nullptr && bytes > 0, "bad argument");
    TDK_TRY(tdk::ensure_device());
    TDK_HIP(hipHostMalloc(out, bytes, hipHostMallocDefault));
    return TDK_OK;
}

tdk_status tdk_pinned_free(void *ptr) {
    TDK_API_GUARD;
    if (ptr) TDK_HIP(hipHostFree(ptr));
    return TDK_OK;
}

tdk_status tdk_device_name(char *buf, int buflen) {
    TDK_API_GUARD;
    TDK_REQUIRE(buf != nullptr && buflen > 0, "bad buffer");
    TDK_TRY(tdk::ensure_device());
    int dev = 0;
    TDK_HIP(hipGetDevice(&dev));
    hipDeviceProp_t prop;
    TDK_HIP(hipGetDeviceProperties(&prop, dev));
    snprintf(buf, (size_t)buflen, "%s (%s, %d CUs)", prop.name, prop.gcnArchName,
             prop.multiProcessorCount);
    return TDK_OK;
}

}  // extern "C"
