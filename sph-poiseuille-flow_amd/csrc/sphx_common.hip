// sphx_common.hip -- error state, device selection, version.
#include "sphx_common.hpp"

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
