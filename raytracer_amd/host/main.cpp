// Command-line renderer: raytracer <scene.gltf|scene.glb> [--profile preview|final] [--width pixels]
// [--height pixels] [--samples count] [--denoise on|off] [--output image.png|image.exr]
//
// Same options, defaults and messages as the reference's src/main.cpp:100-230 (an output file implies
// the final profile; final means 256 samples, preview 32).  Two differences: without --output the
// reference opens its interactive viewer, which this build does not have, so --output is required; and
// --denoise on is refused, the OptiX denoiser having no counterpart here.  After rendering, the ray
// statistics of the engine are printed.
#include "image.hpp"
#include "renderer.hpp"
#include "scene.hpp"

#include <algorithm>
#include <charconv>
#include <chrono>
#include <exception>
#include <filesystem>
#include <iomanip>
#include <iostream>
#include <optional>
#include <sstream>
#include <string_view>
#include <unistd.h>

namespace {

using Clock = std::chrono::steady_clock;

double secondsSince(Clock::time_point start) { return std::chrono::duration<double>(Clock::now() - start).count(); }

std::string duration(double seconds) {  // main.cpp:26-38
    const auto value = static_cast<std::uint64_t>(seconds);
    std::ostringstream out;
    if (value >= 3600)
        out << value / 3600 << 'h' << std::setw(2) << std::setfill('0') << value / 60 % 60 << 'm';
    else if (value >= 60)
        out << value / 60 << 'm' << std::setw(2) << std::setfill('0') << value % 60 << 's';
    else
        out << value << 's';
    return out.str();
}

// One line per update: percent, samples, elapsed, rate, estimate (main.cpp:40-91); a bar on a terminal.
class Progress {
public:
    explicit Progress(std::uint32_t total) : total_(total), terminal_(isatty(STDOUT_FILENO) != 0), start_(Clock::now()), last_(start_) {
        show(0);
    }
    void show(std::uint32_t current) {
        const auto now = Clock::now();
        const auto interval = terminal_ ? std::chrono::milliseconds(100) : std::chrono::milliseconds(5000);
        if (current != total_ && current && now - last_ < interval)
            return;
        last_ = now;
        const double elapsed = std::chrono::duration<double>(now - start_).count();
        const double rate = current / std::max(elapsed, 1e-6);
        std::ostringstream line;
        line << "Rendering ";
        if (terminal_) {
            constexpr std::uint32_t width = 24;
            const auto filled = width * current / total_;
            line << '[' << std::string(filled, '#') << std::string(width - filled, '.') << "] ";
        }
        line << std::setw(3) << 100U * current / total_ << "% | " << current << '/' << total_ << " spp | " << duration(elapsed);
        if (current)
            line << " | " << std::fixed << std::setprecision(2) << rate << " spp/s";
        if (current && current != total_)
            line << " | ETA " << duration((total_ - current) / rate);
        if (terminal_)
            std::cout << "\r\033[2K" << line.str() << std::flush;
        else
            std::cout << line.str() << '\n' << std::flush;
    }
    void finish() {
        if (terminal_)
            std::cout << '\n';
    }

private:
    std::uint32_t total_;
    bool terminal_;
    Clock::time_point start_, last_;
};

template <typename Function>
void phase(std::string_view label, Function&& function) {
    std::cout << label << "..." << std::flush;
    const auto start = Clock::now();
    function();
    std::cout << " done in " << duration(secondsSince(start)) << '\n';
}

struct Options {  // main.cpp:100-110
    std::filesystem::path scene;
    std::optional<std::filesystem::path> output;
    rt::Profile profile = rt::Profile::Final;
    std::uint32_t width = 1280;
    std::uint32_t height = 720;
    std::uint32_t samples = 32;
    bool profileSet = false;
    bool samplesSet = false;
    bool denoise = false;
    bool denoiseSet = false;
};

std::uint32_t positiveInteger(std::string_view value, std::string_view name) {
    std::uint32_t result = 0;
    const auto parsed = std::from_chars(value.data(), value.data() + value.size(), result);
    if (parsed.ec != std::errc{} || parsed.ptr != value.data() + value.size() || result == 0)
        throw std::runtime_error("Invalid " + std::string(name));
    return result;
}

Options options(int argc, char** argv) {  // main.cpp:122-170
    if (argc < 2)
        throw std::runtime_error("usage: raytracer <scene.gltf|scene.glb> [--profile preview|final] "
                                 "[--width pixels] [--height pixels] [--samples count] "
                                 "[--denoise on|off] [--output image.png]");
    Options result;
    result.scene = argv[1];
    for (int index = 2; index < argc; index += 2) {
        if (index + 1 >= argc)
            throw std::runtime_error("Missing option value");
        const std::string_view name = argv[index], value = argv[index + 1];
        if (name == "--output") {
            result.output = std::filesystem::path(value);
        } else if (name == "--profile") {
            if (value == "preview")
                result.profile = rt::Profile::Preview;
            else if (value == "final")
                result.profile = rt::Profile::Final;
            else
                throw std::runtime_error("Expected preview or final profile");
            result.profileSet = true;
        } else if (name == "--width") {
            result.width = positiveInteger(value, "width");
        } else if (name == "--height") {
            result.height = positiveInteger(value, "height");
        } else if (name == "--samples") {
            result.samples = positiveInteger(value, "sample count");
            result.samplesSet = true;
        } else if (name == "--denoise") {
            if (value == "on")
                result.denoise = true;
            else if (value == "off")
                result.denoise = false;
            else
                throw std::runtime_error("Expected on or off for denoise");
            result.denoiseSet = true;
        } else {
            throw std::runtime_error("Unknown option: " + std::string(name));
        }
    }
    if (result.output && !result.profileSet)
        result.profile = rt::Profile::Final;
    if (!result.samplesSet)
        result.samples = result.profile == rt::Profile::Final ? 256U : 32U;
    return result;
}

}  // namespace

int main(int argc, char** argv) {
    try {
        const auto arguments = options(argc, argv);
        if (!arguments.output)
            throw std::runtime_error("this build has no interactive viewer: pass --output image.png or image.exr");
        if (arguments.output->extension() != ".png" && arguments.output->extension() != ".exr")
            throw std::runtime_error("Output must use .png or .exr");
        if (arguments.denoise)
            throw std::runtime_error("the OptiX denoiser is not part of this build: use --denoise off");
        const auto width = arguments.width, height = arguments.height;
        const auto loadStart = Clock::now();
        std::cout << "Loading " << arguments.scene << "..." << std::flush;
        const auto scene = rt::loadScene(arguments.scene);
        std::cout << " done in " << duration(secondsSince(loadStart)) << '\n';
        std::cout << "Loaded " << arguments.scene << '\n'  // main.cpp:183-199
                  << "  " << scene.geometries.size() << " geometries, " << scene.instances.size() << " instances\n"
                  << "  " << scene.materials.size() << " materials, " << scene.textures.size() << " textures, "
                  << scene.images.size() << " images\n"
                  << "  " << scene.cameras.size() << " cameras, " << scene.lights.size() << " lights\n";
        if (!scene.cameras.empty())
            std::cout << "  camera aperture " << scene.cameras.front().aperture << ", focus "
                      << scene.cameras.front().focusDistance << " m\n";
        if (!scene.environment.pixels.empty())
            std::cout << "  HDRI " << scene.environment.width << 'x' << scene.environment.height << ", strength "
                      << scene.environment.strength << '\n';
        std::cout << "  profile " << (arguments.profile == rt::Profile::Final ? "final" : "preview") << ", " << width << 'x'
                  << height << '\n';

        const auto initializeStart = Clock::now();
        std::cout << "Initializing renderer..." << std::flush;
        rt::Renderer renderer(scene, width, height, arguments.profile);
        std::cout << " done in " << duration(secondsSince(initializeStart)) << '\n';

        const auto renderStart = Clock::now();
        Progress progress(arguments.samples);
        while (renderer.samples() < arguments.samples) {
            renderer.render();
            if (renderer.samples() % 8U == 0U || renderer.samples() == arguments.samples)
                renderer.synchronize();  // render() only queues work; keep the progress line honest
            progress.show(renderer.samples());
        }
        renderer.synchronize();
        progress.finish();
        const double renderSeconds = secondsSince(renderStart);
        phase("Writing " + arguments.output->string(), [&] {
            if (arguments.output->extension() == ".exr")
                rt::writeExr(*arguments.output, width, height, renderer.linearPixels());
            else
                rt::writePng(*arguments.output, width, height, renderer.pixels());
        });
        std::uint64_t statistics[8] = {};
        renderer.statistics(statistics);
        const double rays = static_cast<double>(statistics[0] + statistics[1]);
        std::cout << "Done at " << renderer.samples() << " spp\n"
                  << "  " << std::fixed << std::setprecision(1) << rays / std::max(renderSeconds, 1e-9) / 1e6 << " Mrays/s ("
                  << statistics[0] << " closest-hit + " << statistics[1] << " shadow rays in " << std::setprecision(2)
                  << renderSeconds << " s)";
        if (statistics[2])
            std::cout << ", " << std::setprecision(1) << static_cast<double>(statistics[2]) / std::max(renderSeconds, 1e-9) / 1e6
                      << " M photon gathers/s";
        std::cout << '\n';
        return 0;
    } catch (const std::exception& error) {
        std::cerr << "error: " << error.what() << '\n';
        return 1;
    }
}
