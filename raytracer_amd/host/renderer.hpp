// rt::Renderer over the MI355X engine: the public interface of the reference's
// src/renderer.hpp:11-37 (same constructor, methods, defaults and ownership
// rules), plus the hooks a multi-GPU driver needs to shard one frame.
#pragma once

#include "scene.hpp"

#include <cstdint>
#include <memory>
#include <vector>

struct rt_launch_params;

namespace rt {

enum class Profile {
    Preview,  // 5 bounces, no photon map
    Final     // 32 bounces, caustic photon map rebuilt every 4 samples
};

class Renderer {
public:
    Renderer(const Scene& scene, std::uint32_t width, std::uint32_t height, Profile profile = Profile::Preview);
    ~Renderer();

    Renderer(const Renderer&) = delete;
    Renderer& operator=(const Renderer&) = delete;

    // One more sample for every pixel (of the pixel range).  `output`, when not
    // null, is a caller-owned DEVICE buffer of width * height RGBA8 pixels.
    // Returns before the GPU has finished.
    void render(void* output = nullptr);
    // The reference runs the OptiX AOV denoiser here; this engine has none and
    // throws std::runtime_error.
    void denoise();
    void setCamera(const Camera& camera);
    void setProfile(Profile profile);
    void copyOutput(void* output) const;        // device-to-device
    std::vector<std::uint8_t> pixels() const;   // RGBA8, row 0 = bottom
    std::vector<float> linearPixels() const;    // RGBA float32 running mean
    std::uint32_t samples() const;

    // ---- hooks that are not part of the reference ----------------------------
    // Photon map sizes (the reference hard-codes 2^24 emissions, 2^23 stored,
    // 2^22 buckets, src/renderer.cpp:533-535).  Resets the accumulation.
    void setPhotonBudget(std::uint32_t emissions, std::uint32_t capacity, std::uint32_t buckets);
    // Sample-index sharding: the n-th render() call draws the random streams of
    // sample index first + n * stride; the running mean still counts local samples.
    void setSampleSequence(std::uint32_t first, std::uint32_t stride);
    // Restricts render() to the pixels [begin, end) in row-major order.
    void setPixelRange(std::uint32_t begin, std::uint32_t end);
    // Restricts render() to every stride-th band of eight pixel rows, starting with band `first`
    // (rows 8 * first ..): the share of GPU `first` of `stride` GPUs that take the bands of one
    // frame in turn, which spreads rows that are dear to trace over all of them.  Replaces a pixel
    // range; (0, 1), the default, is the whole frame.
    void setRowInterleave(std::uint32_t first, std::uint32_t stride);
    // (The wavefront state is created with the first pass, for the pixels render() covers by then.)
    // render() cuts its pixel range into `count` parts (1..8, default 1) that run
    // concurrently on streams of their own; the image does not depend on it.
    // With more than one part the stage timings of the parts overlap in time.
    // samplesInFlight > 1 (at most 8) lets that many consecutive samples of a region run at once, each
    // on a stream of its own, their accumulation steps chained in sample order: for small frames or
    // small shares of one, where a single sample's launches cannot fill the GPU.  Restarts the
    // accumulation.
    // samplesInFlight 0 leaves the choice to the renderer: two passes in flight, or - without a photon map, where
    // two passes of 64 samples do not fit the state budget - one pass of twice the samples.  The default is one
    // region and that choice; samplesInFlight() is what is in use.
    void setConcurrentTiles(std::uint32_t count, std::uint32_t samplesInFlight = 1);
    std::uint32_t samplesInFlight() const;
    // render() calls are collected into passes of up to `samples` consecutive samples (1 to 64) that the GPU
    // takes in one go: the pixels' means take the samples in order as before, so the image does not change,
    // but a pass is as long as its slowest rays however few pixels it covers, and a pass over four samples
    // costs less than four passes (1080p, eight samples per pass: 2.76 -> 2.40 ms per sample; an eighth of the
    // frame, as one of eight GPUs renders it: 0.58 -> 0.35 ms).  0, the default, takes the most of 64, 32, 16, 8, 4, 2, 1 samples for which
    // the wavefront state (rt_wavefront_state_bytes: 356 bytes x pixels covered x samples per pass x passes in
    // flight) stays below the lesser of 24 GiB and half of the device memory that is free; with a photon map a
    // pass holds at most eight groups of four samples, each group gathering from the map that was current for
    // it.  A pass goes out when it is full, with a render() call that names a
    // target of the caller, and whenever the image, the counters or the GPU are asked for (pixels(),
    // linearPixels(), copyOutput(), accumulationBuffer(), outputBuffer(), statistics(), stageTimings(),
    // synchronize()).  Restarts the accumulation.
    void setSampleBatch(std::uint32_t samples);
    std::uint32_t sampleBatch() const;
    // Share of the GPU's wave slots a ray launch of one part takes (setConcurrentTiles
    // resets it: 1/2 for a split frame, 1/1 otherwise).
    void setRayShare(std::uint32_t numerator, std::uint32_t denominator);
    std::uint32_t concurrentTiles() const;
    // Splits a photon pass: render() then no longer runs it by itself; the
    // driver calls beginPhotonPass(), tracePhotonRange() for its share, places
    // the other ranks' photons behind its own in photonBuffer() and calls
    // finishPhotonPass() with the total.
    void setManualPhotonPasses(bool manual);
    bool photonPassDue() const;
    void beginPhotonPass();
    // Returns the photons stored so far in this pass.  With intoStaging they are written to
    // photonStagingBuffer() instead of photonBuffer(): the memory of the sorted photon list, which
    // holds nothing between beginPhotonPass() and finishPhotonPass(), so that a driver can receive
    // every GPU's photons, its own included, straight at their final place in photonBuffer().
    std::uint32_t tracePhotonRange(std::uint32_t firstEmission, std::uint32_t emissionCount, bool intoStaging = false);
    void* photonBuffer(std::uint32_t pass = 0) const;           // device, photonCapacity() * 36 bytes
    void* photonStagingBuffer(std::uint32_t pass = 0) const;    // device, photonCapacity() * 36 bytes
    // The same three steps for several consecutive passes at once - those of the photon maps one camera pass of
    // many samples gathers from - which then share every launch of the tracer: beginPhotonPasses() starts up to
    // `wanted` passes (at most photonPassesTogether(), one when the next pass is no multiple of the count) and
    // returns how many; tracePhotonRangePasses() traces the range for each and reports the stored count per pass;
    // photonBuffer(k) / photonStagingBuffer(k) are the buffers of pass k; finishPhotonPasses() takes the total
    // per pass.
    std::uint32_t photonPassesTogether() const;
    std::uint32_t beginPhotonPasses(std::uint32_t wanted);
    void tracePhotonRangePasses(std::uint32_t firstEmission, std::uint32_t emissionCount, bool intoStaging, std::uint32_t* stored);
    void finishPhotonPasses(const std::uint32_t* totals);
    std::uint32_t photonCapacity() const;
    std::uint32_t photonEmissions() const;
    void finishPhotonPass(std::uint32_t totalPhotons);
    void* accumulationBuffer(int index) const;  // device float4 buffers: 0 accumulation .. 6 normal guide
    void* outputBuffer() const;                 // device RGBA8 image render() writes when given no target
    // Counters since the last call: [0] extend rays, [1] shadow rays, [2] gather
    // queries, [3] photons read by gathers, [4] nodes visited, [5] triangles
    // tested, [6] instances entered (the last three only while traversal counting
    // is on).  Waits for the GPU.
    void statistics(std::uint64_t out[8]);
    // Measurement switches: event-time every stage launch / count traversal steps /
    // keep every photon query in the lane that drew it (same image, for comparison).
    void setMeasurement(bool timeStages, bool countTraversal, bool gatherPerLane = false);
    // Event-timed milliseconds and launch counts per stage since the last call:
    // generate, trace, shade, queue bookkeeping, accumulate.  Waits for the GPU.
    void stageTimings(float milliseconds[5], std::uint32_t launches[5]);
    void synchronize() const;
    const rt_launch_params& launchParams() const;
    float causticRadius() const;

private:
    struct Impl;
    std::unique_ptr<Impl> impl_;
};

}  // namespace rt
