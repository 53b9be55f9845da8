// glTF 2.0 / GLB scene loader producing rt::Scene.
//
// Follows the reference's src/scene.cpp:130-406 field by field (which glTF
// members land where, the defaults, the material index shift, the luminous
// efficacy divide, the `raytracer_*` extras, the error messages), but stands on
// its own: a small JSON reader, the GLB container, base64 data URIs, accessor
// decoding (including KHR_mesh_quantization's normalised integers), node
// hierarchy walk, PNG decoding through zlib and a Radiance .hdr reader.  The
// reference leans on fastgltf, simdjson and stb_image for those parts.
//
// Not supported (an exception says so): sparse accessors, Draco / meshopt
// compression, KTX2 and DDS images (they stay encoded), orthographic cameras.
#include "scene.hpp"

#include <zlib.h>

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <memory>
#include <stdexcept>
#include <string_view>
#include <utility>

namespace rt {
namespace {

// scene.cpp:18-21
constexpr std::uint32_t DoubleSided = 1U << 0U;
constexpr std::uint32_t Unlit = 1U << 1U;
constexpr std::uint32_t AlphaMask = 1U << 2U;
constexpr std::uint32_t AlphaBlend = 1U << 3U;

// scene.cpp:23-28: punctual lights arrive in photometric units.
constexpr float LuminousEfficacy = 683.0f;

// Image::mimeType values (the numbering fastgltf's MimeType gives the reference).
constexpr std::uint32_t MimeNone = 0, MimeJpeg = 1, MimePng = 2, MimeKtx2 = 3, MimeDds = 4, MimeOctetStream = 6;

std::runtime_error error(std::string_view message, std::string_view name = {}) {  // scene.cpp:99-104
    auto text = std::string(message);
    if (!name.empty())
        text += ": " + std::string(name);
    return std::runtime_error(text);
}

// ---- JSON -------------------------------------------------------------------

struct Json {
    enum class Kind { Null, Bool, Number, String, Array, Object };
    Kind kind = Kind::Null;
    bool boolean = false;
    double number = 0.0;
    std::string text;
    std::vector<Json> items;
    std::vector<std::pair<std::string, Json>> members;

    const Json* find(std::string_view key) const {
        if (kind != Kind::Object)
            return nullptr;
        for (const auto& m : members)
            if (m.first == key)
                return &m.second;
        return nullptr;
    }
    bool isNumber() const { return kind == Kind::Number; }
    bool isArray() const { return kind == Kind::Array; }
    bool isObject() const { return kind == Kind::Object; }
    bool isString() const { return kind == Kind::String; }
};

class JsonParser {
public:
    explicit JsonParser(std::string_view source) : s(source) {}

    Json parse() {
        Json value = parseValue(0);
        skipSpace();
        if (at != s.size())
            fail("trailing characters");
        return value;
    }

private:
    std::string_view s;
    std::size_t at = 0;

    [[noreturn]] void fail(const char* what) const {
        throw std::runtime_error("Invalid glTF JSON (" + std::string(what) + " at byte " + std::to_string(at) + ")");
    }
    void skipSpace() {
        while (at < s.size() && (s[at] == ' ' || s[at] == '\t' || s[at] == '\n' || s[at] == '\r'))
            ++at;
    }
    bool take(char c) {
        skipSpace();
        if (at < s.size() && s[at] == c) {
            ++at;
            return true;
        }
        return false;
    }
    void literal(const char* word) {
        const std::size_t n = std::strlen(word);
        if (s.compare(at, n, word) != 0)
            fail("unknown literal");
        at += n;
    }
    static void appendUtf8(std::string& out, std::uint32_t code) {
        if (code < 0x80) {
            out += static_cast<char>(code);
        } else if (code < 0x800) {
            out += static_cast<char>(0xc0 | (code >> 6));
            out += static_cast<char>(0x80 | (code & 0x3f));
        } else if (code < 0x10000) {
            out += static_cast<char>(0xe0 | (code >> 12));
            out += static_cast<char>(0x80 | ((code >> 6) & 0x3f));
            out += static_cast<char>(0x80 | (code & 0x3f));
        } else {
            out += static_cast<char>(0xf0 | (code >> 18));
            out += static_cast<char>(0x80 | ((code >> 12) & 0x3f));
            out += static_cast<char>(0x80 | ((code >> 6) & 0x3f));
            out += static_cast<char>(0x80 | (code & 0x3f));
        }
    }
    std::uint32_t hex4() {
        if (at + 4 > s.size())
            fail("short \\u escape");
        std::uint32_t v = 0;
        for (int k = 0; k < 4; ++k) {
            const char c = s[at++];
            v <<= 4;
            if (c >= '0' && c <= '9')
                v |= static_cast<std::uint32_t>(c - '0');
            else if (c >= 'a' && c <= 'f')
                v |= static_cast<std::uint32_t>(c - 'a' + 10);
            else if (c >= 'A' && c <= 'F')
                v |= static_cast<std::uint32_t>(c - 'A' + 10);
            else
                fail("bad \\u escape");
        }
        return v;
    }
    std::string parseString() {
        std::string out;
        for (;;) {
            if (at >= s.size())
                fail("unterminated string");
            const char c = s[at++];
            if (c == '"')
                return out;
            if (c != '\\') {
                out += c;
                continue;
            }
            if (at >= s.size())
                fail("unterminated escape");
            const char e = s[at++];
            switch (e) {
            case '"': out += '"'; break;
            case '\\': out += '\\'; break;
            case '/': out += '/'; break;
            case 'b': out += '\b'; break;
            case 'f': out += '\f'; break;
            case 'n': out += '\n'; break;
            case 'r': out += '\r'; break;
            case 't': out += '\t'; break;
            case 'u': {
                std::uint32_t code = hex4();
                if (code >= 0xd800 && code < 0xdc00 && at + 1 < s.size() && s[at] == '\\' && s[at + 1] == 'u') {
                    at += 2;
                    const std::uint32_t low = hex4();
                    code = 0x10000 + ((code - 0xd800) << 10) + (low - 0xdc00);
                }
                appendUtf8(out, code);
                break;
            }
            default: fail("unknown escape");
            }
        }
    }
    Json parseValue(int depth) {
        if (depth > 256)
            fail("nesting too deep");
        skipSpace();
        if (at >= s.size())
            fail("unexpected end");
        Json value;
        const char c = s[at];
        if (c == '{') {
            ++at;
            value.kind = Json::Kind::Object;
            if (take('}'))
                return value;
            for (;;) {
                skipSpace();
                if (at >= s.size() || s[at] != '"')
                    fail("expected a member name");
                ++at;
                std::string key = parseString();
                if (!take(':'))
                    fail("expected ':'");
                value.members.emplace_back(std::move(key), parseValue(depth + 1));
                if (take(','))
                    continue;
                if (take('}'))
                    return value;
                fail("expected ',' or '}'");
            }
        }
        if (c == '[') {
            ++at;
            value.kind = Json::Kind::Array;
            if (take(']'))
                return value;
            for (;;) {
                value.items.push_back(parseValue(depth + 1));
                if (take(','))
                    continue;
                if (take(']'))
                    return value;
                fail("expected ',' or ']'");
            }
        }
        if (c == '"') {
            ++at;
            value.kind = Json::Kind::String;
            value.text = parseString();
            return value;
        }
        if (c == 't') {
            literal("true");
            value.kind = Json::Kind::Bool;
            value.boolean = true;
            return value;
        }
        if (c == 'f') {
            literal("false");
            value.kind = Json::Kind::Bool;
            return value;
        }
        if (c == 'n') {
            literal("null");
            return value;
        }
        const std::size_t start = at;
        while (at < s.size() && (std::strchr("+-.eE", s[at]) || (s[at] >= '0' && s[at] <= '9')))
            ++at;
        if (at == start)
            fail("unexpected character");
        const std::string token(s.substr(start, at - start));
        char* end = nullptr;
        value.number = std::strtod(token.c_str(), &end);
        if (end != token.c_str() + token.size())
            fail("malformed number");
        value.kind = Json::Kind::Number;
        return value;
    }
};

// Typed reads with glTF defaults.
float numberOr(const Json* object, std::string_view key, float fallback) {
    const Json* v = object ? object->find(key) : nullptr;
    return v && v->isNumber() ? static_cast<float>(v->number) : fallback;
}
std::int64_t indexOr(const Json* object, std::string_view key, std::int64_t fallback) {
    const Json* v = object ? object->find(key) : nullptr;
    return v && v->isNumber() ? static_cast<std::int64_t>(v->number) : fallback;
}
// A byte offset, length, stride or element count: a non-negative whole number below 2^53 (what a
// JSON number holds exactly).  Anything else - negative, fractional, huge, NaN - is refused rather
// than cast: a negative offset turned into a size_t would wrap and slip past every bounds check.
std::size_t sizeOr(const Json* object, std::string_view key, std::size_t fallback) {
    const Json* v = object ? object->find(key) : nullptr;
    if (!v || !v->isNumber())
        return fallback;
    const double n = v->number;
    if (!(n >= 0.0) || n > 9007199254740992.0 || n != std::floor(n))
        throw error("Expected a non-negative whole number", std::string(key));
    return static_cast<std::size_t>(n);
}
std::string stringOr(const Json* object, std::string_view key, std::string fallback = {}) {
    const Json* v = object ? object->find(key) : nullptr;
    return v && v->isString() ? v->text : fallback;
}
bool boolOr(const Json* object, std::string_view key, bool fallback) {
    const Json* v = object ? object->find(key) : nullptr;
    return v && v->kind == Json::Kind::Bool ? v->boolean : fallback;
}
template <std::size_t N>
std::array<float, N> floatsOr(const Json* object, std::string_view key, std::array<float, N> fallback) {
    const Json* v = object ? object->find(key) : nullptr;
    if (!v || !v->isArray() || v->items.size() != N)
        return fallback;
    for (std::size_t i = 0; i < N; ++i)
        if (v->items[i].isNumber())
            fallback[i] = static_cast<float>(v->items[i].number);
    return fallback;
}
const std::vector<Json>& arrayOf(const Json& root, std::string_view key) {
    static const std::vector<Json> empty;
    const Json* v = root.find(key);
    return v && v->isArray() ? v->items : empty;
}
const Json* extension(const Json& object, std::string_view name) {
    const Json* extensions = object.find("extensions");
    return extensions ? extensions->find(name) : nullptr;
}

// ---- bytes --------------------------------------------------------------------

std::vector<std::byte> readFile(const std::filesystem::path& path) {
    std::ifstream in(path, std::ios::binary | std::ios::ate);
    if (!in)
        throw error("Could not open file", path.string());
    const std::streamsize size = in.tellg();
    in.seekg(0);
    std::vector<std::byte> bytes(static_cast<std::size_t>(size));
    if (size > 0 && !in.read(reinterpret_cast<char*>(bytes.data()), size))
        throw error("Could not read file", path.string());
    return bytes;
}

std::vector<std::byte> decodeBase64(std::string_view text) {
    std::vector<std::byte> out;
    out.reserve(text.size() / 4 * 3);
    std::uint32_t accumulator = 0;
    int bits = 0;
    for (const char c : text) {
        int v;
        if (c >= 'A' && c <= 'Z')
            v = c - 'A';
        else if (c >= 'a' && c <= 'z')
            v = c - 'a' + 26;
        else if (c >= '0' && c <= '9')
            v = c - '0' + 52;
        else if (c == '+' || c == '-')
            v = 62;
        else if (c == '/' || c == '_')
            v = 63;
        else
            continue;  // padding and line breaks
        accumulator = accumulator << 6 | static_cast<std::uint32_t>(v);
        bits += 6;
        if (bits >= 8) {
            bits -= 8;
            out.push_back(static_cast<std::byte>((accumulator >> bits) & 0xffU));
        }
    }
    return out;
}

std::string percentDecode(std::string_view uri) {
    std::string out;
    for (std::size_t i = 0; i < uri.size(); ++i) {
        if (uri[i] == '%' && i + 2 < uri.size() + 0 && std::isxdigit(static_cast<unsigned char>(uri[i + 1])) &&
            std::isxdigit(static_cast<unsigned char>(uri[i + 2]))) {
            out += static_cast<char>(std::strtol(std::string(uri.substr(i + 1, 2)).c_str(), nullptr, 16));
            i += 2;
        } else {
            out += uri[i];
        }
    }
    return out;
}

// A `uri` member: data URI or a path relative to the scene file.  `mime` gets the data URI's type.
std::vector<std::byte> loadUri(const std::string& uri, const std::filesystem::path& directory, std::string* mime) {
    if (uri.rfind("data:", 0) == 0) {
        const std::size_t comma = uri.find(',');
        if (comma == std::string::npos)
            throw error("Malformed data URI");
        const std::string header = uri.substr(5, comma - 5);
        if (mime)
            *mime = header.substr(0, header.find(';'));
        if (header.find(";base64") == std::string::npos)
            throw error("Only base64 data URIs are supported");
        return decodeBase64(std::string_view(uri).substr(comma + 1));
    }
    return readFile(directory / percentDecode(uri));
}

// ---- images ---------------------------------------------------------------------

std::uint32_t be32(const std::byte* p) {
    return std::to_integer<std::uint32_t>(p[0]) << 24 | std::to_integer<std::uint32_t>(p[1]) << 16 |
           std::to_integer<std::uint32_t>(p[2]) << 8 | std::to_integer<std::uint32_t>(p[3]);
}

// PNG -> RGBA8 (row 0 = top row of the file).  8- and 16-bit samples, grey /
// grey+alpha / RGB / RGBA / palette (with tRNS), bit depths below 8 for grey and
// palette, no interlacing.  Returns false for anything else.
bool decodePng(const std::vector<std::byte>& file, std::vector<std::uint8_t>& rgba, std::uint32_t& width,
               std::uint32_t& height) {
    static const unsigned char signature[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
    if (file.size() < 8 || std::memcmp(file.data(), signature, 8) != 0)
        return false;
    std::vector<unsigned char> compressed;
    std::vector<unsigned char> palette, paletteAlpha;
    int depth = 0, colour = 0, interlace = 0;
    bool header = false;
    int transparentGrey = -1;
    for (std::size_t at = 8; at + 12 <= file.size();) {
        const std::uint32_t length = be32(&file[at]);
        const char* type = reinterpret_cast<const char*>(&file[at + 4]);
        const std::byte* data = &file[at + 8];
        if (at + 12 + static_cast<std::size_t>(length) > file.size())
            return false;
        if (std::memcmp(type, "IHDR", 4) == 0 && length >= 13) {
            width = be32(data);
            height = be32(data + 4);
            depth = std::to_integer<int>(data[8]);
            colour = std::to_integer<int>(data[9]);
            interlace = std::to_integer<int>(data[12]);
            header = true;
        } else if (std::memcmp(type, "PLTE", 4) == 0) {
            palette.assign(reinterpret_cast<const unsigned char*>(data), reinterpret_cast<const unsigned char*>(data) + length);
        } else if (std::memcmp(type, "tRNS", 4) == 0) {
            if (colour == 3)
                paletteAlpha.assign(reinterpret_cast<const unsigned char*>(data),
                                    reinterpret_cast<const unsigned char*>(data) + length);
            else if (colour == 0 && length >= 2)
                transparentGrey = std::to_integer<int>(data[0]) << 8 | std::to_integer<int>(data[1]);
        } else if (std::memcmp(type, "IDAT", 4) == 0) {
            compressed.insert(compressed.end(), reinterpret_cast<const unsigned char*>(data),
                              reinterpret_cast<const unsigned char*>(data) + length);
        } else if (std::memcmp(type, "IEND", 4) == 0) {
            break;
        }
        at += 12 + static_cast<std::size_t>(length);
    }
    if (!header || interlace != 0 || width == 0 || height == 0)
        return false;
    int channels;
    switch (colour) {
    case 0: channels = 1; break;
    case 2: channels = 3; break;
    case 3: channels = 1; break;
    case 4: channels = 2; break;
    case 6: channels = 4; break;
    default: return false;
    }
    if (!(depth == 8 || depth == 16 || ((colour == 0 || colour == 3) && (depth == 1 || depth == 2 || depth == 4))))
        return false;
    if (colour == 3 && (depth == 16 || palette.size() < 3))
        return false;
    const std::size_t bitsPerPixel = static_cast<std::size_t>(channels) * depth;
    const std::size_t stride = (static_cast<std::size_t>(width) * bitsPerPixel + 7) / 8;
    const std::size_t pixelBytes = std::max<std::size_t>(1, bitsPerPixel / 8);
    // deflate expands by a factor of 1032 at most: a header that promises more pixels than
    // the compressed data can hold is refused before anything of that size is allocated
    const std::size_t rawBytes = (stride + 1) * static_cast<std::size_t>(height);
    if (rawBytes / 1032 > compressed.size() + 1 || rawBytes > (std::size_t{1} << 32))
        return false;
    std::vector<unsigned char> raw(rawBytes);
    uLongf rawSize = static_cast<uLongf>(raw.size());
    if (uncompress(raw.data(), &rawSize, compressed.data(), static_cast<uLong>(compressed.size())) != Z_OK ||
        rawSize != raw.size())
        return false;
    // undo the row filters in place
    std::vector<unsigned char> previous(stride, 0);
    for (std::uint32_t y = 0; y < height; ++y) {
        unsigned char* row = raw.data() + (stride + 1) * y;
        const int filter = row[0];
        unsigned char* cur = row + 1;
        for (std::size_t x = 0; x < stride; ++x) {
            const int a = x >= pixelBytes ? cur[x - pixelBytes] : 0;
            const int b = previous[x];
            const int c = x >= pixelBytes ? previous[x - pixelBytes] : 0;
            int predicted = 0;
            switch (filter) {
            case 0: break;
            case 1: predicted = a; break;
            case 2: predicted = b; break;
            case 3: predicted = (a + b) / 2; break;
            case 4: {
                const int p = a + b - c, pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c);
                predicted = pa <= pb && pa <= pc ? a : (pb <= pc ? b : c);
                break;
            }
            default: return false;
            }
            cur[x] = static_cast<unsigned char>(cur[x] + predicted);
        }
        std::memcpy(previous.data(), cur, stride);
    }
    rgba.resize(static_cast<std::size_t>(width) * height * 4);
    for (std::uint32_t y = 0; y < height; ++y) {
        const unsigned char* cur = raw.data() + (stride + 1) * y + 1;
        for (std::uint32_t x = 0; x < width; ++x) {
            unsigned char* out = &rgba[(static_cast<std::size_t>(y) * width + x) * 4];
            auto sample = [&](int channel) -> int {  // 8-bit value of one sample (high byte of 16-bit ones)
                if (depth >= 8)
                    return cur[(static_cast<std::size_t>(x) * channels + channel) * (depth / 8)];
                const std::size_t bit = static_cast<std::size_t>(x) * depth;
                return (cur[bit / 8] >> (8 - depth - bit % 8)) & ((1 << depth) - 1);
            };
            if (colour == 3) {
                const std::size_t index = static_cast<std::size_t>(sample(0));
                if (index * 3 + 2 >= palette.size())
                    return false;
                out[0] = palette[index * 3];
                out[1] = palette[index * 3 + 1];
                out[2] = palette[index * 3 + 2];
                out[3] = index < paletteAlpha.size() ? paletteAlpha[index] : 255;
            } else if (colour == 0 || colour == 4) {
                int grey = sample(0);
                const int original = grey;
                if (depth < 8)
                    grey = grey * 255 / ((1 << depth) - 1);
                out[0] = out[1] = out[2] = static_cast<unsigned char>(grey);
                out[3] = colour == 4 ? static_cast<unsigned char>(sample(1)) : 255;
                if (colour == 0 && transparentGrey >= 0 && depth <= 8 && original == transparentGrey)
                    out[3] = 0;
            } else {
                out[0] = static_cast<unsigned char>(sample(0));
                out[1] = static_cast<unsigned char>(sample(1));
                out[2] = static_cast<unsigned char>(sample(2));
                out[3] = colour == 6 ? static_cast<unsigned char>(sample(3)) : 255;
            }
        }
    }
    return true;
}

// Radiance .hdr (RGBE, flat or run-length encoded scanlines, -Y +X) -> RGBA
// float, alpha 1, rows in file order: what stbi_loadf(path, .., 4) gives the
// reference (scene.cpp:318-326): value = mantissa * 2^(exponent - 136), and
// (0, 0, 0) for a zero exponent.
bool decodeHdr(const std::vector<std::byte>& file, std::vector<Vec4>& pixels, std::uint32_t& width,
               std::uint32_t& height) {
    const char* text = reinterpret_cast<const char*>(file.data());
    std::size_t at = 0;
    auto line = [&]() -> std::string {
        std::string out;
        while (at < file.size() && text[at] != '\n')
            out += text[at++];
        ++at;
        return out;
    };
    const std::string magic = line();
    if (magic != "#?RADIANCE" && magic != "#?RGBE")
        return false;
    bool format = false;
    for (;;) {
        if (at >= file.size())
            return false;
        const std::string l = line();
        if (l.empty())
            break;
        if (l == "FORMAT=32-bit_rle_rgbe")
            format = true;
    }
    if (!format)
        return false;
    const std::string resolution = line();
    int h = 0, w = 0;
    if (std::sscanf(resolution.c_str(), "-Y %d +X %d", &h, &w) != 2 || w <= 0 || h <= 0)
        return false;
    // a run of two bytes yields 127 channel values at most, a pixel has four: a header that promises
    // more than 16 pixels per byte left in the file is refused before anything is allocated
    if (at > file.size() || static_cast<std::uint64_t>(w) * static_cast<std::uint64_t>(h) > (file.size() - at) * 16ULL + 16ULL)
        return false;
    width = static_cast<std::uint32_t>(w);
    height = static_cast<std::uint32_t>(h);
    pixels.resize(static_cast<std::size_t>(w) * h);
    const auto* bytes = reinterpret_cast<const unsigned char*>(file.data());
    auto convert = [](const unsigned char* rgbe) -> Vec4 {
        if (rgbe[3] == 0)
            return {0.0f, 0.0f, 0.0f, 1.0f};
        const float scale = std::ldexp(1.0f, static_cast<int>(rgbe[3]) - (128 + 8));
        return {rgbe[0] * scale, rgbe[1] * scale, rgbe[2] * scale, 1.0f};
    };
    std::vector<unsigned char> scanline(static_cast<std::size_t>(w) * 4);
    for (int y = 0; y < h; ++y) {
        if (at + 4 > file.size())
            return false;
        const bool rle = w >= 8 && w < 32768 && bytes[at] == 2 && bytes[at + 1] == 2 && !(bytes[at + 2] & 0x80) &&
                         (bytes[at + 2] << 8 | bytes[at + 3]) == w;
        if (!rle) {
            // flat pixels for the rest of the file (a flat file has no per-line marker)
            const std::size_t remaining = (static_cast<std::size_t>(h) - y) * w;
            if (at + remaining * 4 > file.size())
                return false;
            for (std::size_t i = 0; i < remaining; ++i)
                pixels[static_cast<std::size_t>(y) * w + i] = convert(bytes + at + 4 * i);
            return true;
        }
        at += 4;
        for (int channel = 0; channel < 4; ++channel) {
            for (int x = 0; x < w;) {
                if (at >= file.size())
                    return false;
                int count = bytes[at++];
                if (count > 128) {
                    count -= 128;
                    if (count == 0 || x + count > w || at >= file.size())
                        return false;
                    const unsigned char value = bytes[at++];
                    for (int k = 0; k < count; ++k)
                        scanline[static_cast<std::size_t>(x++) * 4 + channel] = value;
                } else {
                    if (count == 0 || x + count > w || at + count > file.size())
                        return false;
                    for (int k = 0; k < count; ++k)
                        scanline[static_cast<std::size_t>(x++) * 4 + channel] = bytes[at++];
                }
            }
        }
        for (int x = 0; x < w; ++x)
            pixels[static_cast<std::size_t>(y) * w + x] = convert(&scanline[static_cast<std::size_t>(x) * 4]);
    }
    return true;
}

// ---- the asset ------------------------------------------------------------------

struct BufferView {
    std::size_t buffer = 0, offset = 0, length = 0, stride = 0;
};

struct Accessor {
    std::int64_t view = -1;
    std::size_t offset = 0, count = 0;
    int componentType = 5126;
    int components = 1;
    bool normalized = false;
};

struct Asset {
    Json root;
    std::filesystem::path directory;
    std::vector<std::vector<std::byte>> buffers;
    std::vector<BufferView> views;
    std::vector<Accessor> accessors;
};

int componentBytes(int type) {
    switch (type) {
    case 5120: case 5121: return 1;
    case 5122: case 5123: return 2;
    case 5125: case 5126: return 4;
    default: throw error("Unknown accessor component type");
    }
}

int componentsOf(const std::string& type) {
    if (type == "SCALAR") return 1;
    if (type == "VEC2") return 2;
    if (type == "VEC3") return 3;
    if (type == "VEC4") return 4;
    if (type == "MAT2") return 4;
    if (type == "MAT3") return 9;
    if (type == "MAT4") return 16;
    throw error("Unknown accessor type", type);
}

// Element i, component c as float: floats as they are, normalised integers by
// the KHR_mesh_quantization rules, other integers converted.
struct AccessorReader {
    const std::byte* base = nullptr;
    std::size_t stride = 0;
    Accessor accessor;

    AccessorReader(const Asset& asset, std::size_t index) {
        if (index >= asset.accessors.size())
            throw error("Accessor index out of range");
        accessor = asset.accessors[index];
        const std::size_t element = static_cast<std::size_t>(componentBytes(accessor.componentType)) * accessor.components;
        if (accessor.view < 0)
            return;  // all zeros
        if (static_cast<std::size_t>(accessor.view) >= asset.views.size())
            throw error("Buffer view index out of range");
        const BufferView& view = asset.views[static_cast<std::size_t>(accessor.view)];
        if (view.buffer >= asset.buffers.size())
            throw error("Buffer index out of range");
        const auto& buffer = asset.buffers[view.buffer];
        stride = view.stride ? view.stride : element;
        // no sum or product here may wrap: every term is compared against what is left of the buffer
        const std::size_t size = buffer.size();
        if (view.offset > size || accessor.offset > size - view.offset)
            throw error("Accessor reaches past the end of its buffer");
        const std::size_t begin = view.offset + accessor.offset;
        if (accessor.count > 0) {
            const std::size_t room = size - begin;
            if (element > room || (accessor.count - 1) > (room - element) / stride)
                throw error("Accessor reaches past the end of its buffer");
        }
        base = buffer.data() + begin;
    }

    float component(std::size_t i, int c) const {
        if (!base)
            return 0.0f;
        const std::byte* p = base + i * stride + static_cast<std::size_t>(c) * componentBytes(accessor.componentType);
        switch (accessor.componentType) {
        case 5126: {
            float v;
            std::memcpy(&v, p, 4);
            return v;
        }
        case 5120: {
            std::int8_t v;
            std::memcpy(&v, p, 1);
            return accessor.normalized ? std::max(static_cast<float>(v) / 127.0f, -1.0f) : static_cast<float>(v);
        }
        case 5121: {
            std::uint8_t v;
            std::memcpy(&v, p, 1);
            return accessor.normalized ? static_cast<float>(v) / 255.0f : static_cast<float>(v);
        }
        case 5122: {
            std::int16_t v;
            std::memcpy(&v, p, 2);
            return accessor.normalized ? std::max(static_cast<float>(v) / 32767.0f, -1.0f) : static_cast<float>(v);
        }
        case 5123: {
            std::uint16_t v;
            std::memcpy(&v, p, 2);
            return accessor.normalized ? static_cast<float>(v) / 65535.0f : static_cast<float>(v);
        }
        default: {
            std::uint32_t v;
            std::memcpy(&v, p, 4);
            return static_cast<float>(v);
        }
        }
    }

    std::uint32_t index(std::size_t i) const {
        if (!base)
            return 0;
        const std::byte* p = base + i * stride;
        switch (accessor.componentType) {
        case 5121: return std::to_integer<std::uint32_t>(p[0]);
        case 5123: {
            std::uint16_t v;
            std::memcpy(&v, p, 2);
            return v;
        }
        case 5125: {
            std::uint32_t v;
            std::memcpy(&v, p, 4);
            return v;
        }
        default: throw error("Index accessors must hold unsigned integers");
        }
    }
};

Asset openAsset(const std::filesystem::path& path) {
    Asset asset;
    asset.directory = path.parent_path();
    std::vector<std::byte> file = readFile(path);
    std::vector<std::byte> binaryChunk;
    bool hasBinaryChunk = false;
    std::string jsonText;
    if (path.extension() == ".glb") {
        if (file.size() < 20 || std::memcmp(file.data(), "glTF", 4) != 0)
            throw error("Not a GLB container", path.string());
        auto le32 = [&](std::size_t at) {
            std::uint32_t v;
            std::memcpy(&v, file.data() + at, 4);
            return v;
        };
        if (le32(4) != 2)
            throw error("Only glTF version 2 is supported", path.string());
        std::size_t at = 12;
        while (at + 8 <= file.size()) {
            const std::uint32_t length = le32(at), type = le32(at + 4);
            if (at + 8 + static_cast<std::size_t>(length) > file.size())
                throw error("Truncated GLB chunk", path.string());
            if (type == 0x4e4f534aU)  // "JSON"
                jsonText.assign(reinterpret_cast<const char*>(file.data() + at + 8), length);
            else if (type == 0x004e4942U && !hasBinaryChunk) {  // "BIN\0"
                binaryChunk.assign(file.begin() + static_cast<std::ptrdiff_t>(at + 8),
                                   file.begin() + static_cast<std::ptrdiff_t>(at + 8 + length));
                hasBinaryChunk = true;
            }
            at += 8 + static_cast<std::size_t>(length);
        }
        if (jsonText.empty())
            throw error("GLB holds no JSON chunk", path.string());
    } else {
        jsonText.assign(reinterpret_cast<const char*>(file.data()), file.size());
    }
    asset.root = JsonParser(jsonText).parse();
    if (!asset.root.isObject())
        throw error("glTF root is not an object", path.string());
    const Json* assetInfo = asset.root.find("asset");
    if (!assetInfo || stringOr(assetInfo, "version").rfind("2.", 0) != 0)
        throw error("Only glTF version 2 is supported", path.string());

    // extensions this loader cannot honour
    for (const Json& required : arrayOf(asset.root, "extensionsRequired")) {
        static const char* known[] = {"KHR_lights_punctual", "KHR_materials_dispersion", "KHR_materials_emissive_strength",
                                      "KHR_materials_ior", "KHR_materials_transmission", "KHR_materials_unlit",
                                      "KHR_materials_volume", "KHR_mesh_quantization", "KHR_texture_transform"};
        bool ok = false;
        for (const char* k : known)
            ok |= required.text == k;
        if (!ok)
            throw error("Required glTF extension is not supported", required.text);  // scene.cpp:336-345
    }

    std::size_t bufferIndex = 0;
    for (const Json& b : arrayOf(asset.root, "buffers")) {
        const std::string uri = stringOr(&b, "uri");
        if (uri.empty()) {
            if (bufferIndex != 0 || !hasBinaryChunk)
                throw error("Buffer without a uri outside a GLB container", path.string());
            asset.buffers.push_back(std::move(binaryChunk));
        } else {
            asset.buffers.push_back(loadUri(uri, asset.directory, nullptr));
        }
        const std::size_t declared = sizeOr(&b, "byteLength", 0);
        if (asset.buffers.back().size() < declared)
            throw error("Buffer is shorter than its byteLength", uri);
        ++bufferIndex;
    }
    for (const Json& v : arrayOf(asset.root, "bufferViews")) {
        BufferView view;
        view.buffer = sizeOr(&v, "buffer", 0);
        view.offset = sizeOr(&v, "byteOffset", 0);
        view.length = sizeOr(&v, "byteLength", 0);
        view.stride = sizeOr(&v, "byteStride", 0);
        if (view.buffer >= asset.buffers.size() || view.offset > asset.buffers[view.buffer].size() ||
            view.length > asset.buffers[view.buffer].size() - view.offset)
            throw error("Buffer view reaches past the end of its buffer", path.string());
        asset.views.push_back(view);
    }
    for (const Json& a : arrayOf(asset.root, "accessors")) {
        if (a.find("sparse"))
            throw error("Sparse accessors are not supported", path.string());
        Accessor accessor;
        accessor.view = indexOr(&a, "bufferView", -1);
        accessor.offset = sizeOr(&a, "byteOffset", 0);
        accessor.count = sizeOr(&a, "count", 0);
        if (accessor.count > 0xffffffffULL)  // vertices and indices are addressed with 32 bits
            throw error("Accessor has too many elements", path.string());
        accessor.componentType = static_cast<int>(indexOr(&a, "componentType", 5126));
        accessor.components = componentsOf(stringOr(&a, "type", "SCALAR"));
        accessor.normalized = boolOr(&a, "normalized", false);
        componentBytes(accessor.componentType);
        asset.accessors.push_back(accessor);
    }
    return asset;
}

// ---- glTF objects -> rt:: structs ----------------------------------------------

TextureRef textureRef(const Json* info) {  // scene.cpp:106-128
    TextureRef result;
    if (!info || !info->isObject())
        return result;
    result.texture = static_cast<std::int32_t>(indexOr(info, "index", -1));
    result.texCoord = static_cast<std::uint32_t>(indexOr(info, "texCoord", 0));
    if (const Json* transform = extension(*info, "KHR_texture_transform")) {
        const auto offset = floatsOr<2>(transform, "offset", {0.0f, 0.0f});
        const auto scale = floatsOr<2>(transform, "scale", {1.0f, 1.0f});
        result.offset = {offset[0], offset[1]};
        result.scale = {scale[0], scale[1]};
        result.rotation = numberOr(transform, "rotation", 0.0f);
        if (const Json* coordinate = transform->find("texCoord"); coordinate && coordinate->isNumber())
            result.texCoord = static_cast<std::uint32_t>(coordinate->number);
    }
    return result;
}

Material material(const Json& source) {  // scene.cpp:130-162
    Material result;
    result.name = stringOr(&source, "name");
    const Json* pbr = source.find("pbrMetallicRoughness");
    const auto baseColor = floatsOr<4>(pbr, "baseColorFactor", {1.0f, 1.0f, 1.0f, 1.0f});
    const auto emissive = floatsOr<3>(&source, "emissiveFactor", {0.0f, 0.0f, 0.0f});
    result.baseColor = {baseColor[0], baseColor[1], baseColor[2], baseColor[3]};
    result.emissive = {emissive[0], emissive[1], emissive[2]};
    result.metallic = numberOr(pbr, "metallicFactor", 1.0f);
    result.roughness = numberOr(pbr, "roughnessFactor", 1.0f);
    result.ior = numberOr(extension(source, "KHR_materials_ior"), "ior", 1.5f);
    result.dispersion = numberOr(extension(source, "KHR_materials_dispersion"), "dispersion", 0.0f);
    result.emissiveStrength =
        numberOr(extension(source, "KHR_materials_emissive_strength"), "emissiveStrength", 1.0f);
    result.alphaCutoff = numberOr(&source, "alphaCutoff", 0.5f);
    const std::string alphaMode = stringOr(&source, "alphaMode", "OPAQUE");
    result.flags = (boolOr(&source, "doubleSided", false) ? DoubleSided : 0U) |
                   (extension(source, "KHR_materials_unlit") ? Unlit : 0U) | (alphaMode == "MASK" ? AlphaMask : 0U) |
                   (alphaMode == "BLEND" ? AlphaBlend : 0U);
    if (const Json* transmission = extension(source, "KHR_materials_transmission")) {
        result.transmission = numberOr(transmission, "transmissionFactor", 0.0f);
        result.transmissionTexture = textureRef(transmission->find("transmissionTexture"));
    }
    if (const Json* volume = extension(source, "KHR_materials_volume")) {
        result.thickness = numberOr(volume, "thicknessFactor", 0.0f);
        const auto attenuation = floatsOr<3>(volume, "attenuationColor", {1.0f, 1.0f, 1.0f});
        result.attenuationColor = {attenuation[0], attenuation[1], attenuation[2]};
        result.attenuationDistance =
            numberOr(volume, "attenuationDistance", std::numeric_limits<float>::infinity());
        result.thicknessTexture = textureRef(volume->find("thicknessTexture"));
    }
    result.baseColorTexture = textureRef(pbr ? pbr->find("baseColorTexture") : nullptr);
    result.metallicRoughnessTexture = textureRef(pbr ? pbr->find("metallicRoughnessTexture") : nullptr);
    const Json* normal = source.find("normalTexture");
    result.normalTexture = textureRef(normal);
    if (normal)
        result.normalTexture.strength = numberOr(normal, "scale", 1.0f);  // scene.cpp:125-126
    result.emissiveTexture = textureRef(source.find("emissiveTexture"));
    return result;
}

Geometry geometry(const Asset& asset, const Json& mesh, const Json& primitive, std::size_t primitiveIndex) {
    const std::string meshName = stringOr(&mesh, "name");  // scene.cpp:164-227
    if (indexOr(&primitive, "mode", 4) != 4)
        throw error("Only triangle primitives are supported", meshName);
    const Json* attributes = primitive.find("attributes");
    const std::int64_t positions = indexOr(attributes, "POSITION", -1);
    if (positions < 0)
        throw error("Malformed mesh primitive", meshName);

    Geometry result;
    result.name = meshName.empty() ? "mesh." + std::to_string(primitiveIndex)
                                   : meshName + "." + std::to_string(primitiveIndex);
    const AccessorReader position(asset, static_cast<std::size_t>(positions));
    if (position.accessor.components != 3)
        throw error("Malformed mesh primitive", meshName);
    result.vertices.resize(position.accessor.count);
    for (std::size_t i = 0; i < result.vertices.size(); ++i) {
        Vertex vertex{};
        vertex.position = {position.component(i, 0), position.component(i, 1), position.component(i, 2)};
        result.vertices[i] = vertex;
    }
    auto attribute = [&](const char* name, int components, auto&& assign) {
        const std::int64_t index = indexOr(attributes, name, -1);
        if (index < 0)
            return;
        const AccessorReader reader(asset, static_cast<std::size_t>(index));
        if (reader.accessor.components != components)
            throw error("Malformed mesh primitive", meshName);
        if (reader.accessor.count > result.vertices.size())
            throw std::out_of_range("vertex attribute longer than POSITION");  // .at() in scene.cpp:188
        for (std::size_t i = 0; i < reader.accessor.count; ++i)
            assign(result.vertices[i], reader, i);
    };
    attribute("NORMAL", 3, [](Vertex& v, const AccessorReader& r, std::size_t i) {
        v.normal = {r.component(i, 0), r.component(i, 1), r.component(i, 2)};
    });
    attribute("TANGENT", 4, [](Vertex& v, const AccessorReader& r, std::size_t i) {
        v.tangent = {r.component(i, 0), r.component(i, 1), r.component(i, 2), r.component(i, 3)};
    });
    attribute("TEXCOORD_0", 2, [](Vertex& v, const AccessorReader& r, std::size_t i) {
        v.uv = {r.component(i, 0), r.component(i, 1)};
    });
    attribute("TEXCOORD_1", 2, [](Vertex& v, const AccessorReader& r, std::size_t i) {
        v.uv1 = {r.component(i, 0), r.component(i, 1)};
    });

    const std::int64_t indices = indexOr(&primitive, "indices", -1);
    if (indices >= 0) {
        const AccessorReader reader(asset, static_cast<std::size_t>(indices));
        result.indices.resize(reader.accessor.count);
        for (std::size_t i = 0; i < result.indices.size(); ++i)
            result.indices[i] = reader.index(i);
    } else {
        // non-indexed primitive: 0, 1, 2, ... (what GenerateMeshIndices does upstream, scene.cpp:349)
        result.indices.resize(result.vertices.size());
        for (std::size_t i = 0; i < result.indices.size(); ++i)
            result.indices[i] = static_cast<std::uint32_t>(i);
    }
    if (result.indices.size() % 3 != 0 ||
        std::any_of(result.indices.begin(), result.indices.end(),
                    [&](std::uint32_t index) { return index >= result.vertices.size(); }))
        throw error("Invalid triangle indices", result.name);

    const std::int64_t materialIndex = indexOr(&primitive, "material", -1);
    result.material = materialIndex >= 0 ? static_cast<std::uint32_t>(materialIndex + 1) : 0U;
    return result;
}

std::uint32_t mimeOf(const std::string& mime, const std::string& uri) {
    auto endsWith = [&](const char* suffix) {
        const std::size_t n = std::strlen(suffix);
        if (uri.size() < n)
            return false;
        for (std::size_t i = 0; i < n; ++i)
            if (std::tolower(static_cast<unsigned char>(uri[uri.size() - n + i])) != suffix[i])
                return false;
        return true;
    };
    if (mime == "image/png" || endsWith(".png"))
        return MimePng;
    if (mime == "image/jpeg" || endsWith(".jpg") || endsWith(".jpeg"))
        return MimeJpeg;
    if (mime == "image/ktx2" || endsWith(".ktx2"))
        return MimeKtx2;
    if (mime == "image/vnd-ms.dds" || endsWith(".dds"))
        return MimeDds;
    if (mime == "application/octet-stream")
        return MimeOctetStream;
    return MimeNone;
}

Image image(const Asset& asset, const Json& source) {  // scene.cpp:229-256
    Image result;
    result.name = stringOr(&source, "name");
    std::string mime = stringOr(&source, "mimeType");
    const std::string uri = stringOr(&source, "uri");
    if (!uri.empty()) {
        std::string dataMime;
        result.encoded = loadUri(uri, asset.directory, &dataMime);
        if (mime.empty())
            mime = dataMime;
        result.mimeType = mimeOf(mime, uri.rfind("data:", 0) == 0 ? std::string() : uri);
    } else {
        const std::int64_t view = indexOr(&source, "bufferView", -1);
        if (view < 0 || static_cast<std::size_t>(view) >= asset.views.size())
            throw error("Image data was not loaded", result.name);
        const BufferView& v = asset.views[static_cast<std::size_t>(view)];
        const auto& buffer = asset.buffers[v.buffer];
        result.encoded.assign(buffer.begin() + static_cast<std::ptrdiff_t>(v.offset),
                              buffer.begin() + static_cast<std::ptrdiff_t>(v.offset + v.length));
        result.mimeType = mimeOf(mime, {});
    }
    // Decoded here when the format is one this loader reads; other formats stay
    // encoded (the renderer then reports the texture as undecodable).
    if (!decodePng(result.encoded, result.rgba, result.width, result.height) &&
        !decodeJpegImage(result.encoded, result.rgba, result.width, result.height)) {
        result.rgba.clear();
        result.width = result.height = 0;
    }
    return result;
}

Texture texture(const Asset& asset, const Json& source) {  // scene.cpp:258-272
    Texture result;
    result.name = stringOr(&source, "name");
    const std::int64_t imageIndex = indexOr(&source, "source", -1);
    if (imageIndex < 0)
        throw error("Only core glTF image sources are supported", result.name);
    result.image = static_cast<std::int32_t>(imageIndex);
    const std::int64_t samplerIndex = indexOr(&source, "sampler", -1);
    if (samplerIndex >= 0) {
        const auto& samplers = arrayOf(asset.root, "samplers");
        if (static_cast<std::size_t>(samplerIndex) >= samplers.size())
            throw error("Sampler index out of range", result.name);
        const Json& sampler = samplers[static_cast<std::size_t>(samplerIndex)];
        result.magFilter = static_cast<std::uint32_t>(indexOr(&sampler, "magFilter", 0));
        result.minFilter = static_cast<std::uint32_t>(indexOr(&sampler, "minFilter", 0));
        result.wrapU = static_cast<std::uint32_t>(indexOr(&sampler, "wrapS", 10497));
        result.wrapV = static_cast<std::uint32_t>(indexOr(&sampler, "wrapT", 10497));
    }
    return result;
}

// 4x4 column-major float matrices, products evaluated in float.
using Matrix = std::array<float, 16>;

Matrix identity() {
    return {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
}

Matrix multiply(const Matrix& a, const Matrix& b) {
    Matrix r{};
    for (int column = 0; column < 4; ++column)
        for (int row = 0; row < 4; ++row) {
            float sum = 0.0f;
            for (int k = 0; k < 4; ++k)
                sum += a[static_cast<std::size_t>(k * 4 + row)] * b[static_cast<std::size_t>(column * 4 + k)];
            r[static_cast<std::size_t>(column * 4 + row)] = sum;
        }
    return r;
}

Matrix localTransform(const Json& node) {
    if (const Json* m = node.find("matrix"); m && m->isArray() && m->items.size() == 16) {
        Matrix result{};
        for (std::size_t i = 0; i < 16; ++i)
            result[i] = static_cast<float>(m->items[i].number);
        return result;
    }
    const auto t = floatsOr<3>(&node, "translation", {0.0f, 0.0f, 0.0f});
    const auto q = floatsOr<4>(&node, "rotation", {0.0f, 0.0f, 0.0f, 1.0f});
    const auto s = floatsOr<3>(&node, "scale", {1.0f, 1.0f, 1.0f});
    const float x = q[0], y = q[1], z = q[2], w = q[3];
    Matrix r = identity();  // T * R * S
    r[0] = (1.0f - 2.0f * (y * y + z * z)) * s[0];
    r[1] = (2.0f * (x * y + z * w)) * s[0];
    r[2] = (2.0f * (x * z - y * w)) * s[0];
    r[4] = (2.0f * (x * y - z * w)) * s[1];
    r[5] = (1.0f - 2.0f * (x * x + z * z)) * s[1];
    r[6] = (2.0f * (y * z + x * w)) * s[1];
    r[8] = (2.0f * (x * z + y * w)) * s[2];
    r[9] = (2.0f * (y * z - x * w)) * s[2];
    r[10] = (1.0f - 2.0f * (x * x + y * y)) * s[2];
    r[12] = t[0];
    r[13] = t[1];
    r[14] = t[2];
    return r;
}

Mat4 mat4(const Matrix& m) {
    Mat4 result{};
    result.values = m;
    return result;
}

struct CameraExtras {  // scene.cpp:30-33
    float aperture = 0.0f;
    float focusDistance = 10.0f;
};

struct Extras {  // scene.cpp:35-42
    std::string environment;
    float rotation = 0.0f;
    float strength = 1.0f;
    float exposure = 0.0f;
    float luminousEfficacy = LuminousEfficacy;
};

Camera camera(const Json& source, const Matrix& transform) {  // scene.cpp:274-290, 44-57
    const std::string name = stringOr(&source, "name");
    const Json* perspective = source.find("perspective");
    if (stringOr(&source, "type") != "perspective" || !perspective)
        throw error("Orthographic cameras are not supported", name);
    Camera result;
    result.name = name;
    result.transform = mat4(transform);
    result.verticalFov = numberOr(perspective, "yfov", 0.0f);
    result.aspectRatio = numberOr(perspective, "aspectRatio", 0.0f);
    result.nearPlane = numberOr(perspective, "znear", 0.0f);
    result.farPlane = numberOr(perspective, "zfar", std::numeric_limits<float>::infinity());
    const Json* extras = source.find("extras");
    CameraExtras values;
    result.aperture = numberOr(extras, "raytracer_aperture", values.aperture);
    result.focusDistance = numberOr(extras, "raytracer_focus_distance", values.focusDistance);
    return result;
}

Light light(const Json& source, const Matrix& transform, float luminousEfficacy) {  // scene.cpp:292-304
    const auto color = floatsOr<3>(&source, "color", {1.0f, 1.0f, 1.0f});
    const std::string type = stringOr(&source, "type");
    std::uint32_t kind;
    if (type == "directional")
        kind = 0;
    else if (type == "spot")
        kind = 1;
    else if (type == "point")
        kind = 2;
    else
        throw error("Unknown punctual light type", type);
    const Json* spot = source.find("spot");
    Light result;
    result.name = stringOr(&source, "name");
    result.transform = mat4(transform);
    result.color = {color[0], color[1], color[2]};
    result.intensity = numberOr(&source, "intensity", 1.0f) / luminousEfficacy;
    result.range = numberOr(&source, "range", std::numeric_limits<float>::infinity());
    result.innerCone = numberOr(spot, "innerConeAngle", 0.0f);
    result.outerCone = numberOr(spot, "outerConeAngle", 0.7853982f);
    result.type = kind;
    return result;
}

Environment environment(const std::filesystem::path& scenePath, const Extras& extras) {  // scene.cpp:306-328
    Environment result;
    result.rotation = extras.rotation;
    result.strength = extras.strength;
    result.exposure = extras.exposure;
    if (extras.environment.empty())
        return result;
    const auto path = scenePath.parent_path() / extras.environment;
    const std::vector<std::byte> file = readFile(path);
    if (!decodeHdr(file, result.pixels, result.width, result.height))
        throw error("Only Radiance .hdr environment images are supported", path.string());
    return result;
}

void walk(const Asset& asset, const std::vector<Json>& nodes, std::size_t index, const Matrix& parent, int depth,
          const std::vector<std::vector<std::uint32_t>>& meshGeometries, const Extras& extras, Scene& scene) {
    if (index >= nodes.size())
        throw error("Node index out of range");
    if (depth > 512)
        throw error("Node hierarchy too deep (a cycle?)");
    const Json& node = nodes[index];
    const Matrix transform = multiply(parent, localTransform(node));
    const std::string name = stringOr(&node, "name");
    if (const std::int64_t mesh = indexOr(&node, "mesh", -1); mesh >= 0) {  // scene.cpp:391-394
        if (static_cast<std::size_t>(mesh) >= meshGeometries.size())
            throw error("Mesh index out of range", name);
        for (const auto geometryIndex : meshGeometries[static_cast<std::size_t>(mesh)])
            scene.instances.push_back({name, mat4(transform), geometryIndex});
    }
    if (const std::int64_t cameraIndex = indexOr(&node, "camera", -1); cameraIndex >= 0) {  // scene.cpp:395-400
        const auto& cameras = arrayOf(asset.root, "cameras");
        if (static_cast<std::size_t>(cameraIndex) >= cameras.size())
            throw error("Camera index out of range", name);
        scene.cameras.push_back(camera(cameras[static_cast<std::size_t>(cameraIndex)], transform));
    }
    if (const Json* punctual = extension(node, "KHR_lights_punctual")) {  // scene.cpp:401-403
        if (const std::int64_t lightIndex = indexOr(punctual, "light", -1); lightIndex >= 0) {
            const Json* root = extension(asset.root, "KHR_lights_punctual");
            const Json* lights = root ? root->find("lights") : nullptr;
            if (!lights || !lights->isArray() || static_cast<std::size_t>(lightIndex) >= lights->items.size())
                throw error("Light index out of range", name);
            scene.lights.push_back(
                light(lights->items[static_cast<std::size_t>(lightIndex)], transform, extras.luminousEfficacy));
        }
    }
    if (const Json* children = node.find("children"); children && children->isArray())
        for (const Json& child : children->items)
            walk(asset, nodes, static_cast<std::size_t>(child.number), transform, depth + 1, meshGeometries, extras,
                 scene);
}

}  // namespace

bool decodePngImage(const std::vector<std::byte>& file, std::vector<std::uint8_t>& rgba, std::uint32_t& width,
                    std::uint32_t& height) {
    return decodePng(file, rgba, width, height);
}

bool decodeRadianceImage(const std::vector<std::byte>& file, std::vector<Vec4>& pixels, std::uint32_t& width,
                         std::uint32_t& height) {
    return decodeHdr(file, pixels, width, height);
}

Scene loadScene(const std::filesystem::path& path) {  // scene.cpp:332-406
    if (path.extension() != ".gltf" && path.extension() != ".glb")
        throw error("Expected a glTF 2.0 .gltf or .glb scene", path.string());
    const Asset asset = openAsset(path);

    const auto& scenes = arrayOf(asset.root, "scenes");
    if (scenes.empty())
        throw error("glTF contains no scene", path.string());
    const std::size_t sceneIndex = static_cast<std::size_t>(indexOr(&asset.root, "scene", 0));
    if (sceneIndex >= scenes.size())
        throw error("Default scene index out of range", path.string());

    Extras extras;  // scene.cpp:58-76 (extras of every scene object, the last one wins)
    for (const Json& s : scenes) {
        const Json* e = s.find("extras");
        if (!e)
            continue;
        extras.environment = stringOr(e, "raytracer_hdri", extras.environment);
        extras.rotation = numberOr(e, "raytracer_hdri_rotation", extras.rotation);
        extras.strength = numberOr(e, "raytracer_hdri_strength", extras.strength);
        extras.exposure = numberOr(e, "raytracer_exposure", extras.exposure);
        const float efficacy = numberOr(e, "raytracer_luminous_efficacy", 0.0f);
        if (efficacy > 0.0f)
            extras.luminousEfficacy = efficacy;
    }

    Scene scene;
    scene.environment = environment(path, extras);
    scene.materials.emplace_back();  // material 0 is the default; glTF material i becomes i + 1
    for (const Json& source : arrayOf(asset.root, "materials"))
        scene.materials.push_back(material(source));
    for (const Json& source : arrayOf(asset.root, "images"))
        scene.images.push_back(image(asset, source));
    for (const Json& source : arrayOf(asset.root, "textures"))
        scene.textures.push_back(texture(asset, source));

    const auto& meshes = arrayOf(asset.root, "meshes");
    std::vector<std::vector<std::uint32_t>> meshGeometries(meshes.size());
    for (std::size_t meshIndex = 0; meshIndex < meshes.size(); ++meshIndex) {
        const auto& primitives = arrayOf(meshes[meshIndex], "primitives");
        for (std::size_t primitiveIndex = 0; primitiveIndex < primitives.size(); ++primitiveIndex) {
            meshGeometries[meshIndex].push_back(static_cast<std::uint32_t>(scene.geometries.size()));
            scene.geometries.push_back(geometry(asset, meshes[meshIndex], primitives[primitiveIndex], primitiveIndex));
        }
    }

    const auto& nodes = arrayOf(asset.root, "nodes");
    for (const Json& root : arrayOf(scenes[sceneIndex], "nodes"))
        walk(asset, nodes, static_cast<std::size_t>(root.number), identity(), 0, meshGeometries, extras, scene);
    return scene;
}

}  // namespace rt
