// Shared host/device helpers of the engine library.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <string>

#include "rt_engine.h"

namespace rt {

// Remembers the message of the most recent failure on this host thread
// (rt_last_error()).
void setLastError(const std::string& message);

inline rt_status fail(hipError_t error, const char* what) {
    setLastError(std::string(what) + ": " + hipGetErrorString(error));
    return static_cast<rt_status>(error);
}

inline rt_status failInvalid(const char* what) {
    setLastError(std::string("invalid argument: ") + what);
    return RT_ERROR_INVALID_VALUE;
}

#define RT_HIP(expr)                                              \
    do {                                                          \
        const hipError_t rtHipStatus_ = (expr);                   \
        if (rtHipStatus_ != hipSuccess)                           \
            return ::rt::fail(rtHipStatus_, #expr);               \
    } while (0)

#define RT_TRY(expr)                                              \
    do {                                                          \
        const rt_status rtStatus_ = (expr);                       \
        if (rtStatus_ != RT_SUCCESS)                              \
            return rtStatus_;                                     \
    } while (0)

constexpr unsigned divUp(unsigned long long a, unsigned long long b) {
    return static_cast<unsigned>((a + b - 1) / b);
}

constexpr size_t alignUp(size_t value, size_t alignment) {
    return (value + alignment - 1) / alignment * alignment;
}

// Grid size for a grid-stride kernel: enough workgroups to fill the 256 CUs
// several times over without launching one per tile of a huge input.
inline unsigned stridedGrid(unsigned long long items, unsigned block, unsigned maxBlocks = 256U * 16U) {
    const unsigned long long blocks = (items + block - 1) / block;
    return static_cast<unsigned>(blocks < 1 ? 1 : (blocks > maxBlocks ? maxBlocks : blocks));
}

}  // namespace rt
