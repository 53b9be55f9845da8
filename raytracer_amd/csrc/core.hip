// Error reporting shared by every entry point of the engine library.
#include "common.h"
#include "rt_math.h"

namespace rt {
namespace {
thread_local std::string lastError;
}

void setLastError(const std::string& message) {
    lastError = message;
}

namespace {
// Keeps one wave busy for `ticks` of the 100 MHz clock (bounded by an iteration
// count in case the clock does not advance).
__global__ void occupyKernel(unsigned long long ticks, unsigned int* sink) {
    const unsigned long long begin = wall_clock64();
    unsigned int spins = 0;
    while (wall_clock64() - begin < ticks && spins < (1U << 22))
        ++spins;
    if (sink && spins == 0xffffffffU)
        *sink = spins;
}

// The elementary functions the shading kernels use (rt_math.h), over arrays.
__global__ __launch_bounds__(256) void sharedMathKernel(int which, const float* x, const float* y, uint32_t n, float* out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n)
        return;
    switch (which) {
    case 0: out[i] = rtm::sine(x[i]); break;
    case 1: out[i] = rtm::cosine(x[i]); break;
    case 2: out[i] = rtm::power(x[i], y[i]); break;
    case 3: out[i] = rtm::arcTangent2(x[i], y[i]); break;
    case 4: out[i] = rtm::arcCosine(x[i]); break;
    default: out[i] = rtm::exp2(x[i]); break;
    }
}
}  // namespace

}  // namespace rt

extern "C" {

rt_status rt_shared_math(int which, const float* x, const float* y, uint32_t n, float* out, void* stream) {
    if (which < 0 || which > 5)
        return rt::failInvalid("which: 0 sine .. 5 exp2");
    if (n == 0)
        return RT_SUCCESS;
    if (!x || !out || ((which == 2 || which == 3) && !y))
        return rt::failInvalid("null pointer");
    hipLaunchKernelGGL(rt::sharedMathKernel, dim3(rt::divUp(n, 256)), dim3(256), 0, static_cast<hipStream_t>(stream), which, x,
                       y, n, out);
    RT_HIP(hipGetLastError());
    return RT_SUCCESS;
}

rt_status rt_streams_run_concurrently(void* const* streams, uint32_t count, int* concurrent) {
    if (!streams || !concurrent || count == 0 || count > 16)
        return rt::failInvalid("streams / concurrent / count");
    *concurrent = 1;
    if (count == 1)
        return RT_SUCCESS;
    // one short wave per stream, started together: side by side they take the
    // time of one, through a shared hardware queue the time of several
    constexpr unsigned long long kTicks = 20000;  // 200 us
    hipEvent_t start = nullptr, stop = nullptr;
    RT_HIP(hipDeviceSynchronize());
    RT_HIP(hipEventCreate(&start));
    RT_HIP(hipEventCreate(&stop));
    float best = 1e30f;
    for (int attempt = 0; attempt < 2; ++attempt) {
        RT_HIP(hipEventRecord(start, nullptr));  // the blocking default stream orders against all of them
        for (uint32_t k = 0; k < count; ++k)
            hipLaunchKernelGGL(rt::occupyKernel, dim3(1), dim3(64), 0, static_cast<hipStream_t>(streams[k]), kTicks,
                               static_cast<unsigned int*>(nullptr));
        RT_HIP(hipEventRecord(stop, nullptr));
        RT_HIP(hipEventSynchronize(stop));
        float ms = 0.0f;
        RT_HIP(hipEventElapsedTime(&ms, start, stop));
        best = ms < best ? ms : best;
    }
    (void)hipEventDestroy(start);
    (void)hipEventDestroy(stop);
    *concurrent = best < 0.2f * 1.6f ? 1 : 0;
    return RT_SUCCESS;
}

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
