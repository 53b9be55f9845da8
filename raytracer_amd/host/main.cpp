// Command-line front end of the engine.
//
//   raytracer <scene.gltf|scene.glb> [--profile preview|final] [--width pixels] [--height pixels]
//             [--samples count] [--denoise on|off] [--output image.png|image.exr]
//
// Accepts the command lines of the reference's executable (flags, defaults and error texts of
// src/main.cpp:100-170; the offline loop of :211-230): an output file without --profile means the
// final profile, final renders 256 samples and preview 32 unless --samples says otherwise.
// What differs, because two parts of the reference are outside this engine:
//   * there is no viewer, so a command line without --output is an error instead of a window;
//   * there is no denoiser.  The reference denoises a final render unless told --denoise off; here
//     the same command line renders, prints a notice in place of the denoising step and writes the
//     raw accumulation.  Asking for the denoiser by name (--denoise on) is an error.
// After the image is written the engine's ray statistics are printed (Mrays/s, photon gathers/s),
// which the reference does not have.
#include "image.hpp"
#include "renderer.hpp"
#include "scene.hpp"

#include <chrono>
#include <cinttypes>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <memory>
#include <stdexcept>
#include <string>
#include <unistd.h>

namespace {

// ---- time -------------------------------------------------------------------------------------

class Stopwatch {
public:
    Stopwatch() : begin_(std::chrono::steady_clock::now()) {}
    double seconds() const {
        return std::chrono::duration_cast<std::chrono::duration<double>>(std::chrono::steady_clock::now() - begin_).count();
    }

private:
    std::chrono::steady_clock::time_point begin_;
};

// 42s, 3m07s, 2h05m: the reference's spelling of a span (main.cpp:26-38), whole seconds.
std::string spanText(double seconds) {
    const unsigned long long whole = seconds > 0.0 ? static_cast<unsigned long long>(seconds) : 0ULL;
    char text[48];
    if (whole < 60ULL)
        std::snprintf(text, sizeof text, "%llus", whole);
    else if (whole < 3600ULL)
        std::snprintf(text, sizeof text, "%llum%02llus", whole / 60ULL, whole % 60ULL);
    else
        std::snprintf(text, sizeof text, "%lluh%02llum", whole / 3600ULL, whole / 60ULL % 60ULL);
    return text;
}

// Runs `work` between "<label>..." and " done in <span>".
void timedStep(const std::string& label, const std::function<void()>& work) {
    std::printf("%s...", label.c_str());
    std::fflush(stdout);
    const Stopwatch watch;
    work();
    std::printf(" done in %s\n", spanText(watch.seconds()).c_str());
}

// ---- command line -----------------------------------------------------------------------------

struct Request {
    std::string scenePath;
    std::string outputPath;          // empty: none given
    bool final = true;               // main.cpp:103
    bool profileGiven = false;
    std::uint32_t width = 1280, height = 720;
    std::uint32_t samples = 0;       // 0: not given, the profile decides
    enum class Denoise { Default, On, Off } denoise = Denoise::Default;
};

[[noreturn]] void refuse(const std::string& message) { throw std::runtime_error(message); }

std::uint32_t countFrom(const char* text, const char* what) {
    // digits only, no sign, no blanks, at least 1, fits 32 bits
    std::uint64_t value = 0;
    const char* p = text;
    for (; *p >= '0' && *p <= '9' && value <= 0xffffffffULL; ++p)
        value = value * 10U + static_cast<unsigned>(*p - '0');
    if (p == text || *p != '\0' || value == 0 || value > 0xffffffffULL)
        refuse(std::string("Invalid ") + what);
    return static_cast<std::uint32_t>(value);
}

Request readCommandLine(int argc, char** argv) {
    if (argc < 2)
        refuse("usage: raytracer <scene.gltf|scene.glb> [--profile preview|final] [--width pixels] [--height pixels] "
               "[--samples count] [--denoise on|off] [--output image.png]");
    Request request;
    request.scenePath = argv[1];
    struct Flag {
        const char* name;
        std::function<void(const char*)> take;
    };
    const Flag flags[] = {
        {"--output", [&](const char* v) { request.outputPath = v; }},
        {"--profile",
         [&](const char* v) {
             if (std::strcmp(v, "preview") != 0 && std::strcmp(v, "final") != 0)
                 refuse("Expected preview or final profile");
             request.final = v[0] == 'f';
             request.profileGiven = true;
         }},
        {"--width", [&](const char* v) { request.width = countFrom(v, "width"); }},
        {"--height", [&](const char* v) { request.height = countFrom(v, "height"); }},
        {"--samples", [&](const char* v) { request.samples = countFrom(v, "sample count"); }},
        {"--denoise",
         [&](const char* v) {
             if (std::strcmp(v, "on") == 0)
                 request.denoise = Request::Denoise::On;
             else if (std::strcmp(v, "off") == 0)
                 request.denoise = Request::Denoise::Off;
             else
                 refuse("Expected on or off for denoise");
         }},
    };
    for (int at = 2; at < argc; at += 2) {
        if (at + 1 == argc)
            refuse("Missing option value");
        const Flag* match = nullptr;
        for (const Flag& flag : flags)
            if (std::strcmp(flag.name, argv[at]) == 0)
                match = &flag;
        if (!match)
            refuse(std::string("Unknown option: ") + argv[at]);
        match->take(argv[at + 1]);
    }
    if (!request.outputPath.empty() && !request.profileGiven)
        request.final = true;                         // main.cpp:165-166
    if (request.samples == 0)
        request.samples = request.final ? 256U : 32U;  // main.cpp:167-168
    return request;
}

bool endsWith(const std::string& text, const char* suffix) {
    const std::size_t n = std::strlen(suffix);
    return text.size() >= n && text.compare(text.size() - n, n, suffix) == 0;
}

// ---- progress ---------------------------------------------------------------------------------

// "Rendering  37% | 12/32 spp | 4s | 2.81 spp/s | ETA 7s" (main.cpp:56-77): redrawn in place with a
// bar every 100 ms on a terminal, a new line every 5 s otherwise; the first and the last state always.
class SampleMeter {
public:
    explicit SampleMeter(std::uint32_t goal) : goal_(goal), onTerminal_(isatty(STDOUT_FILENO) == 1) { report(0); }

    void report(std::uint32_t done) {
        const double now = watch_.seconds();
        const bool edge = done == 0 || done == goal_;
        if (!edge && now - lastReport_ < (onTerminal_ ? 0.1 : 5.0))
            return;
        lastReport_ = now;
        std::string line = "Rendering ";
        if (onTerminal_) {
            const std::uint32_t cells = 24, full = static_cast<std::uint32_t>(std::uint64_t(cells) * done / goal_);
            line += '[' + std::string(full, '#') + std::string(cells - full, '.') + "] ";
        }
        char part[96];
        std::snprintf(part, sizeof part, "%3u%% | %u/%u spp | %s", static_cast<unsigned>(100ULL * done / goal_), done, goal_,
                      spanText(now).c_str());
        line += part;
        if (done) {
            const double rate = done / (now > 1e-6 ? now : 1e-6);
            std::snprintf(part, sizeof part, " | %.2f spp/s", rate);
            line += part;
            if (done != goal_)
                line += " | ETA " + spanText((goal_ - done) / rate);
        }
        std::printf(onTerminal_ ? "\r\033[2K%s" : "%s\n", line.c_str());
        std::fflush(stdout);
    }

    void close() const {
        if (onTerminal_)
            std::printf("\n");
    }

private:
    std::uint32_t goal_;
    bool onTerminal_;
    Stopwatch watch_;
    double lastReport_ = 0.0;
};

void describe(const Request& request, const rt::Scene& scene) {  // main.cpp:185-204
    std::printf("Loaded \"%s\"\n", request.scenePath.c_str());
    std::printf("  %zu geometries, %zu instances\n", scene.geometries.size(), scene.instances.size());
    std::printf("  %zu materials, %zu textures, %zu images\n", scene.materials.size(), scene.textures.size(),
                scene.images.size());
    std::printf("  %zu cameras, %zu lights\n", scene.cameras.size(), scene.lights.size());
    if (!scene.cameras.empty())
        std::printf("  camera aperture %g, focus %g m\n", scene.cameras.front().aperture, scene.cameras.front().focusDistance);
    if (!scene.environment.pixels.empty())
        std::printf("  HDRI %ux%u, strength %g\n", scene.environment.width, scene.environment.height, scene.environment.strength);
    std::printf("  profile %s, %ux%u\n", request.final ? "final" : "preview", request.width, request.height);
}

int run(int argc, char** argv) {
    const Request request = readCommandLine(argc, argv);
    if (request.outputPath.empty())
        refuse("this build has no interactive viewer: pass --output image.png or image.exr");
    const bool exr = endsWith(request.outputPath, ".exr");
    if (!exr && !endsWith(request.outputPath, ".png"))
        refuse("Output must use .png or .exr");
    if (request.denoise == Request::Denoise::On)
        refuse("the OptiX denoiser is not part of this build: use --denoise off");

    rt::Scene scene;
    timedStep("Loading \"" + request.scenePath + "\"", [&] { scene = rt::loadScene(request.scenePath); });
    describe(request, scene);

    const rt::Profile profile = request.final ? rt::Profile::Final : rt::Profile::Preview;
    std::unique_ptr<rt::Renderer> renderer;
    timedStep("Initializing renderer",
              [&] { renderer = std::make_unique<rt::Renderer>(scene, request.width, request.height, profile); });

    const Stopwatch renderWatch;
    SampleMeter meter(request.samples);
    while (renderer->samples() < request.samples) {
        renderer->render();
        // render() only queues work: wait now and then so that the rate shown is the rate achieved
        if (renderer->samples() % 8U == 0U || renderer->samples() == request.samples)
            renderer->synchronize();
        meter.report(renderer->samples());
    }
    meter.close();
    const double renderSeconds = renderWatch.seconds();

    if (request.final && request.denoise == Request::Denoise::Default)
        std::printf("Denoising with OptiX... skipped: this build has no denoiser, the raw accumulation is written\n");
    timedStep("Writing " + request.outputPath, [&] {
        if (exr)
            rt::writeExr(request.outputPath, request.width, request.height, renderer->linearPixels());
        else
            rt::writePng(request.outputPath, request.width, request.height, renderer->pixels());
    });
    std::printf("Done at %u spp\n", renderer->samples());

    std::uint64_t counters[8] = {};
    renderer->statistics(counters);
    const double span = renderSeconds > 1e-9 ? renderSeconds : 1e-9;
    std::printf("  %.1f Mrays/s (%" PRIu64 " closest-hit + %" PRIu64 " shadow rays in %.2f s)",
                static_cast<double>(counters[0] + counters[1]) / span / 1e6, counters[0], counters[1], renderSeconds);
    if (counters[2])
        std::printf(", %.1f M photon gathers/s", static_cast<double>(counters[2]) / span / 1e6);
    std::printf("\n");
    return 0;
}

}  // namespace

int main(int argc, char** argv) {
    try {
        return run(argc, argv);
    } catch (const std::exception& error) {
        std::fflush(stdout);
        std::fprintf(stderr, "error: %s\n", error.what());
        return 1;
    }
}
