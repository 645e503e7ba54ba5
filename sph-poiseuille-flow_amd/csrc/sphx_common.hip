// sphx_common.hip -- error state, device selection, version.
#include "sphx_common.hpp"

#include <map>
#include <mutex>
#include <unordered_map>
#include <utility>

namespace sphx {

static thread_local std::string g_err_msg;
static thread_local std::string g_err_id;

void set_last_error(int code, const std::string &id, const std::string &msg)
{
    (void)code;
    g_err_id = id;
    g_err_msg = msg;
}

int report(const Error &e)
{
    set_last_error(e.code, e.id, e.what());
    return e.code;
}

int report_unknown(const std::exception &e)
{
    set_last_error(SPHX_ERR_DEVICE, "SPHX:Internal", e.what());
    return SPHX_ERR_DEVICE;
}

namespace {

size_t pool_class(size_t bytes)
{  // power of two below 1 MiB, whole MiB above: few distinct sizes, at most 2x slack on small blocks
    size_t c = 256;
    if (bytes <= (1u << 20)) {
        while (c < bytes) c <<= 1;
        return c;
    }
    return (bytes + (1u << 20) - 1) & ~(size_t)((1u << 20) - 1);
}

struct Pool {
    std::mutex m;
    std::multimap<std::pair<int, size_t>, void *> free_blocks;  // (device, class) -> block
    std::unordered_map<void *, int> owner;                      // block -> the device it was allocated on
    size_t cached = 0;
    static constexpr size_t kMaxCached = (size_t)4 << 30;
    void trim()
    {
        for (auto &kv : free_blocks) { owner.erase(kv.second); (void)hipFree(kv.second); }
        free_blocks.clear();
        cached = 0;
    }
    ~Pool() { /* process exit: the runtime reclaims device memory; HIP may already be torn down */ }
};

Pool &pool()
{
    static Pool *p = new Pool();  // never destroyed: DevBufs in static storage may outlive any static pool
    return *p;
}

}  // namespace

void *pool_alloc(size_t bytes)
{
    const size_t cls = pool_class(bytes);
    int dev = 0;
    SPHX_HIP(hipGetDevice(&dev));
    Pool &P = pool();
    {
        std::lock_guard<std::mutex> lk(P.m);
        auto it = P.free_blocks.find({dev, cls});
        if (it != P.free_blocks.end()) {
            void *p = it->second;
            P.free_blocks.erase(it);
            P.cached -= cls;
            return p;
        }
    }
    void *p = nullptr;
    hipError_t e = hipMalloc(&p, cls);
    if (e != hipSuccess) {  // give the cache back and try once more
        (void)hipGetLastError();
        {
            std::lock_guard<std::mutex> lk(P.m);
            P.trim();
        }
        e = hipMalloc(&p, cls);
    }
    SPHX_HIP(e);
    {
        std::lock_guard<std::mutex> lk(P.m);
        P.owner[p] = dev;
    }
    return p;
}

void pool_free(void *p, size_t bytes)
{
    if (!p) return;
    const size_t cls = pool_class(bytes);
    Pool &P = pool();
    std::lock_guard<std::mutex> lk(P.m);
    // a block goes back under the device that allocated it, whatever device is current now
    auto it = P.owner.find(p);
    if (it == P.owner.end() || P.cached + cls > Pool::kMaxCached) {
        if (it != P.owner.end()) P.owner.erase(it);
        (void)hipFree(p);
        return;
    }
    P.free_blocks.insert({{it->second, cls}, p});
    P.cached += cls;
}

void ensure_device()
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        (void)hipGetLastError();
        throw Error(SPHX_ERR_DEVICE, "SPHX:NoDevice",
                    "libsphx: no HIP device available (this library has no CPU fallback)");
    }
}

}  // namespace sphx

SPHX_EXPORT const char *sphx_version(void) { return "sphx 0.1 (gfx950)"; }
SPHX_EXPORT const char *sphx_last_error(void) { return sphx::g_err_msg.c_str(); }
SPHX_EXPORT const char *sphx_last_error_id(void) { return sphx::g_err_id.c_str(); }

SPHX_EXPORT int sphx_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return n;
}

SPHX_EXPORT int sphx_set_device(int device)
{
    SPHX_TRY
    sphx::ensure_device();
    SPHX_HIP(hipSetDevice(device));
    return SPHX_OK;
    SPHX_CATCH
}
