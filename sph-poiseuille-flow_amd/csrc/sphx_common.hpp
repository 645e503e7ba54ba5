// sphx_common.hpp -- host-side plumbing shared by the C-ABI translation units: error reporting
// (message + MEX-style id, the analogue of mexErrMsgIdAndTxt in sph_physics_mex.c:41-46), HIP call
// checking and a small RAII device buffer.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/sphx.h"

#define SPHX_EXPORT extern "C" __attribute__((visibility("default")))

namespace sphx {

struct Error : std::runtime_error {
    int code;
    std::string id;
    Error(int c, std::string i, const std::string &msg) : std::runtime_error(msg), code(c), id(std::move(i)) {}
};

void set_last_error(int code, const std::string &id, const std::string &msg);
int report(const Error &e);
int report_unknown(const std::exception &e);

inline void require(bool cond, const char *id, const char *msg)
{
    if (!cond) throw Error(SPHX_ERR_ARG, id, msg);
}

#define SPHX_HIP(expr)                                                                             \
    do {                                                                                           \
        hipError_t _e = (expr);                                                                    \
        if (_e != hipSuccess)                                                                      \
            throw ::sphx::Error(SPHX_ERR_DEVICE, "SPHX:HIP",                                       \
                                std::string(#expr) + ": " + hipGetErrorString(_e));               \
    } while (0)

// Fails loudly when no HIP device is usable -- there is no CPU fallback by design.
void ensure_device();

// Device memory comes from a small caching pool: the stateless MEX-surface calls allocate a dozen arrays each and
// hipMalloc/hipFree (100+ us apiece, the free synchronising the device) dominated them -- 10.7 ms per step of the
// six-calls-per-step loop at 5 760 particles.  Whoever returns a block must have synchronised the work using it
// (every owner here does: the stateless calls before their scope ends, contexts before they are destroyed).
void *pool_alloc(size_t bytes);
void pool_free(void *p, size_t bytes);

template <typename T>
class DevBuf {
public:
    DevBuf() = default;
    explicit DevBuf(size_t n) { alloc(n); }
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
    DevBuf(DevBuf &&o) noexcept : p_(o.p_), n_(o.n_) { o.p_ = nullptr; o.n_ = 0; }
    DevBuf &operator=(DevBuf &&o) noexcept
    {
        if (this != &o) { release(); p_ = o.p_; n_ = o.n_; o.p_ = nullptr; o.n_ = 0; }
        return *this;
    }
    ~DevBuf() { release(); }
    void alloc(size_t n)
    {
        release();
        n_ = n;
        if (n) p_ = static_cast<T *>(pool_alloc(n * sizeof(T)));
    }
    void release()
    {
        if (p_) pool_free(p_, n_ * sizeof(T));
        p_ = nullptr;
        n_ = 0;
    }
    void zero(hipStream_t s = nullptr)
    {
        if (n_) SPHX_HIP(hipMemsetAsync(p_, 0, n_ * sizeof(T), s));
    }
    void upload(const T *host, size_t n, hipStream_t s = nullptr)
    {
        if (n) SPHX_HIP(hipMemcpyAsync(p_, host, n * sizeof(T), hipMemcpyHostToDevice, s));
    }
    void download(T *host, size_t n, hipStream_t s = nullptr) const
    {
        if (n) SPHX_HIP(hipMemcpyAsync(host, p_, n * sizeof(T), hipMemcpyDeviceToHost, s));
    }
    T *get() const { return p_; }
    size_t size() const { return n_; }

private:
    T *p_ = nullptr;
    size_t n_ = 0;
};

inline unsigned div_up(size_t a, size_t b) { return (unsigned)((a + b - 1) / b); }

}  // namespace sphx

#define SPHX_TRY try {
#define SPHX_CATCH                                                                                 \
    }                                                                                              \
    catch (const ::sphx::Error &e) { return ::sphx::report(e); }                                   \
    catch (const std::exception &e) { return ::sphx::report_unknown(e); }
