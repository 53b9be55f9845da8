// Error reporting shared by every entry point of the engine library.
#include "common.h"

namespace rt {
namespace {
thread_local std::string lastError;
}

void setLastError(const std::string& message) {
    lastError = message;
}

}  // namespace rt

extern "C" {

const char* rt_last_error(void) {
    return rt::lastError.c_str();
}

const char* rt_status_string(rt_status status) {
    if (status == RT_SUCCESS)
        return "success";
    if (status == RT_ERROR_INVALID_VALUE)
        return "invalid value";
    if (status == RT_ERROR_OUT_OF_RANGE)
        return "out of range";
    if (status > 0)
        return hipGetErrorString(static_cast<hipError_t>(status));
    return "unknown error";
}

}  // extern "C"
