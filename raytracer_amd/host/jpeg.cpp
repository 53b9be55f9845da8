// JPEG (ITU-T T.81) decoder for texture images: 8-bit baseline, extended
// sequential and progressive Huffman files with one (grey) or three (YCbCr or
// RGB) components, any integer sampling factors, restart intervals.
//
// The reference hands texture files to stb_image (renderer.cpp:264-299), which is
// not part of its tree; this decoder follows the standard and, where the
// standard leaves the arithmetic open, the choices of the IJG library that most
// JPEG files are made and checked with: its 13-bit integer inverse DCT, its
// triangle-filter chroma upsampling and its 16-bit fixed-point YCbCr -> RGB
// tables.  Arithmetic-coded, 12-bit, lossless and four-component files are refused.
#include "scene.hpp"

#include <algorithm>
#include <array>
#include <cstring>

namespace rt {
namespace {

constexpr std::uint8_t kZigzag[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,
                                      12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6,  7,  14, 21, 28,
                                      35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51,
                                      58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

struct Huffman {
    bool present = false;
    std::uint8_t values[256] = {};
    int minCode[17] = {}, maxCode[18] = {}, firstValue[17] = {};
    // codes of up to 9 bits resolved by one lookup: length << 8 | value, 0 = longer code
    std::uint16_t fast[512] = {};

    bool build(const std::uint8_t* counts /* 16 */, const std::uint8_t* symbols, int symbolCount) {
        int total = 0;
        for (int l = 0; l < 16; ++l)
            total += counts[l];
        if (total > 256 || total != symbolCount)
            return false;
        std::memcpy(values, symbols, static_cast<std::size_t>(total));
        std::memset(fast, 0, sizeof(fast));
        int code = 0, index = 0;
        for (int length = 1; length <= 16; ++length) {
            firstValue[length] = index;
            minCode[length] = code;
            if (code + counts[length - 1] > (1 << length))
                return false;  // more codes of this length than the prefix leaves room for
            for (int k = 0; k < counts[length - 1]; ++k, ++code, ++index) {
                if (length <= 9) {
                    const int shift = 9 - length;
                    for (int pad = 0; pad < (1 << shift); ++pad)
                        fast[(code << shift) | pad] = static_cast<std::uint16_t>(length << 8 | values[index]);
                }
            }
            maxCode[length] = counts[length - 1] ? code - 1 : -1;
            code <<= 1;
        }
        maxCode[17] = 0x7fffffff;
        present = true;
        return true;
    }
};

// Entropy-coded segment reader: removes the zero byte stuffed after 0xFF and
// stops at a marker, after which it delivers zero bits.
struct BitReader {
    const std::uint8_t* data = nullptr;
    std::size_t size = 0, position = 0;
    std::uint32_t buffer = 0;
    int bits = 0;
    bool atMarker = false;

    void fill() {
        while (bits <= 24) {
            std::uint32_t byte = 0;
            if (!atMarker && position < size) {
                byte = data[position];
                if (byte == 0xff) {
                    const std::uint8_t next = position + 1 < size ? data[position + 1] : 0xd9;
                    if (next == 0x00) {
                        position += 2;
                    } else {
                        atMarker = true;  // leave the marker in place
                        byte = 0;
                    }
                } else {
                    ++position;
                }
            }
            buffer |= byte << (24 - bits);
            bits += 8;
        }
    }
    int peek(int count) {
        if (bits < count)
            fill();
        return static_cast<int>(buffer >> (32 - count));
    }
    void skip(int count) {
        buffer <<= count;
        bits -= count;
    }
    int receive(int count) {
        if (count == 0)
            return 0;
        const int v = peek(count);
        skip(count);
        return v;
    }
    // Drops the bits left of the current byte and expects the restart marker next.
    bool restart() {
        buffer = 0;
        bits = 0;
        if (atMarker) {
            atMarker = false;
        } else {
            // a decoder that read exactly to the byte boundary has not seen the marker yet
            while (position + 1 < size && !(data[position] == 0xff && data[position + 1] != 0x00))
                ++position;
        }
        if (position + 1 >= size || data[position] != 0xff || data[position + 1] < 0xd0 || data[position + 1] > 0xd7)
            return false;
        position += 2;
        return true;
    }
};

inline int extend(int value, int bits) {
    return value < (1 << (bits - 1)) ? value - (1 << bits) + 1 : value;
}

int decodeSymbol(BitReader& in, const Huffman& table) {
    const int look = in.peek(9);
    const std::uint16_t entry = table.fast[look];
    if (entry) {
        in.skip(entry >> 8);
        return entry & 0xff;
    }
    int code = in.peek(16);
    for (int length = 10; length <= 16; ++length) {
        const int prefix = code >> (16 - length);
        if (table.maxCode[length] >= 0 && prefix <= table.maxCode[length] && prefix >= table.minCode[length]) {
            in.skip(length);
            return table.values[table.firstValue[length] + prefix - table.minCode[length]];
        }
    }
    return -1;
}

struct Component {
    int id = 0, h = 1, v = 1, quant = 0;
    int dcTable = 0, acTable = 0;
    int width = 0, height = 0;            // samples that belong to the image
    int blocksWide = 0, blocksHigh = 0;   // blocks that belong to the image
    int storedWide = 0, storedHigh = 0;   // blocks stored (whole MCUs)
    int prediction = 0;
    std::vector<std::int16_t> coefficients;  // storedWide * storedHigh blocks, natural order
    std::vector<std::uint8_t> samples;       // (storedWide * 8) x (storedHigh * 8)
};

constexpr int kConstBits = 13, kPass1Bits = 2;
inline std::int64_t descale(std::int64_t x, int n) { return (x + (std::int64_t{1} << (n - 1))) >> n; }
inline std::uint8_t clampByte(std::int64_t v) { return static_cast<std::uint8_t>(v < 0 ? 0 : (v > 255 ? 255 : v)); }

// Inverse DCT of one dequantised block (the "slow integer" method of the IJG
// library: Loeffler-Ligtenberg-Moschytz with 13-bit constants, columns first).
void inverseDct(const std::int16_t* block, const std::uint16_t* quant, std::uint8_t* out, int outStride) {
    constexpr std::int64_t F0298 = 2446, F0390 = 3196, F0541 = 4433, F0765 = 6270, F0899 = 7373, F1175 = 9633,
                           F1501 = 12299, F1847 = 15137, F1961 = 16069, F2053 = 16819, F2562 = 20995, F3072 = 25172;
    std::int64_t work[64];
    for (int pass = 0; pass < 2; ++pass) {
        for (int line = 0; line < 8; ++line) {
            std::int64_t in[8];
            if (pass == 0) {
                for (int k = 0; k < 8; ++k)
                    in[k] = static_cast<std::int64_t>(block[8 * k + line]) * quant[8 * k + line];
            } else {
                for (int k = 0; k < 8; ++k)
                    in[k] = work[8 * line + k];
            }
            std::int64_t z2 = in[2], z3 = in[6];
            std::int64_t z1 = (z2 + z3) * F0541;
            std::int64_t tmp2 = z1 + z3 * (-F1847);
            std::int64_t tmp3 = z1 + z2 * F0765;
            z2 = in[0];
            z3 = in[4];
            std::int64_t tmp0 = (z2 + z3) * (std::int64_t{1} << kConstBits);
            std::int64_t tmp1 = (z2 - z3) * (std::int64_t{1} << kConstBits);
            const std::int64_t tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
            tmp0 = in[7];
            tmp1 = in[5];
            tmp2 = in[3];
            tmp3 = in[1];
            z1 = tmp0 + tmp3;
            z2 = tmp1 + tmp2;
            z3 = tmp0 + tmp2;
            std::int64_t z4 = tmp1 + tmp3;
            const std::int64_t z5 = (z3 + z4) * F1175;
            tmp0 *= F0298;
            tmp1 *= F2053;
            tmp2 *= F3072;
            tmp3 *= F1501;
            z1 *= -F0899;
            z2 *= -F2562;
            z3 *= -F1961;
            z4 *= -F0390;
            z3 += z5;
            z4 += z5;
            tmp0 += z1 + z3;
            tmp1 += z2 + z4;
            tmp2 += z2 + z3;
            tmp3 += z1 + z4;
            const std::int64_t result[8] = {tmp10 + tmp3, tmp11 + tmp2, tmp12 + tmp1, tmp13 + tmp0,
                                            tmp13 - tmp0, tmp12 - tmp1, tmp11 - tmp2, tmp10 - tmp3};
            if (pass == 0) {
                for (int k = 0; k < 8; ++k)
                    work[8 * k + line] = descale(result[k], kConstBits - kPass1Bits);
            } else {
                for (int k = 0; k < 8; ++k)
                    out[line * outStride + k] = clampByte(descale(result[k], kConstBits + kPass1Bits + 3) + 128);
            }
        }
    }
}

struct Decoder {
    const std::uint8_t* data;
    std::size_t size;
    Decoder(const std::uint8_t* bytes, std::size_t count) : data(bytes), size(count) {}
    std::size_t position = 2;
    int width = 0, height = 0;
    bool progressive = false, haveFrame = false;
    int adobeTransform = -1;
    int restartInterval = 0;
    std::uint16_t quant[4][64] = {};  // natural order
    bool quantPresent[4] = {};
    Huffman dc[4], ac[4];
    std::vector<Component> components;
    int hMax = 1, vMax = 1, mcusWide = 0, mcusHigh = 0;

    std::uint32_t word(std::size_t at) const { return static_cast<std::uint32_t>(data[at]) << 8 | data[at + 1]; }

    bool readTables(std::uint8_t marker, std::size_t begin, std::size_t end) {
        std::size_t at = begin;
        if (marker == 0xdb) {  // quantisation tables
            while (at < end) {
                const int precision = data[at] >> 4, slot = data[at] & 15;
                ++at;
                if (slot > 3 || precision > 1 || at + (precision ? 128U : 64U) > end)
                    return false;
                for (int k = 0; k < 64; ++k) {
                    quant[slot][kZigzag[k]] = static_cast<std::uint16_t>(precision ? word(at + 2U * k) : data[at + k]);
                }
                at += precision ? 128 : 64;
                quantPresent[slot] = true;
            }
            return true;
        }
        if (marker == 0xc4) {  // Huffman tables
            while (at < end) {
                if (at + 17 > end)
                    return false;
                const int kind = data[at] >> 4, slot = data[at] & 15;
                if (kind > 1 || slot > 3)
                    return false;
                int total = 0;
                for (int k = 0; k < 16; ++k)
                    total += data[at + 1 + k];
                if (at + 17 + static_cast<std::size_t>(total) > end)
                    return false;
                if (!(kind ? ac : dc)[slot].build(data + at + 1, data + at + 17, total))
                    return false;
                at += 17 + static_cast<std::size_t>(total);
            }
            return true;
        }
        if (marker == 0xdd) {  // restart interval
            if (end - begin < 2)
                return false;
            restartInterval = static_cast<int>(word(begin));
            return true;
        }
        if (marker == 0xee) {  // Adobe: colour transform flag
            if (end - begin >= 12 && std::memcmp(data + begin, "Adobe", 5) == 0)
                adobeTransform = data[begin + 11];
            return true;
        }
        return true;  // other application and comment segments
    }

    bool readFrame(std::uint8_t marker, std::size_t begin, std::size_t end) {
        if (haveFrame || end - begin < 6)
            return false;
        progressive = marker == 0xc2;
        if (data[begin] != 8)
            return false;  // 12-bit samples
        height = static_cast<int>(word(begin + 1));
        width = static_cast<int>(word(begin + 3));
        const int count = data[begin + 5];
        if (width <= 0 || height <= 0 || (count != 1 && count != 3) || end - begin < 6U + 3U * count)
            return false;
        if (static_cast<std::uint64_t>(width) * static_cast<std::uint64_t>(height) > (1ULL << 26))
            return false;
        components.resize(static_cast<std::size_t>(count));
        for (int k = 0; k < count; ++k) {
            Component& c = components[static_cast<std::size_t>(k)];
            c.id = data[begin + 6 + 3 * k];
            c.h = data[begin + 7 + 3 * k] >> 4;
            c.v = data[begin + 7 + 3 * k] & 15;
            c.quant = data[begin + 8 + 3 * k];
            if (c.h < 1 || c.h > 4 || c.v < 1 || c.v > 4 || c.quant > 3)
                return false;
            hMax = std::max(hMax, c.h);
            vMax = std::max(vMax, c.v);
        }
        mcusWide = (width + 8 * hMax - 1) / (8 * hMax);
        mcusHigh = (height + 8 * vMax - 1) / (8 * vMax);
        for (Component& c : components) {
            c.width = (width * c.h + hMax - 1) / hMax;
            c.height = (height * c.v + vMax - 1) / vMax;
            c.blocksWide = (c.width + 7) / 8;
            c.blocksHigh = (c.height + 7) / 8;
            c.storedWide = mcusWide * c.h;
            c.storedHigh = mcusHigh * c.v;
            c.coefficients.assign(static_cast<std::size_t>(c.storedWide) * c.storedHigh * 64, 0);
        }
        haveFrame = true;
        return true;
    }

    // One scan: header at [begin, end), entropy-coded data from `end` on.  Leaves
    // `position` at the marker that ends the scan.
    bool readScan(std::size_t begin, std::size_t end) {
        if (!haveFrame || end - begin < 1)
            return false;
        const int count = data[begin];
        if (count < 1 || count > static_cast<int>(components.size()) || end - begin < 4U + 2U * count)
            return false;
        Component* scan[3] = {};
        for (int k = 0; k < count; ++k) {
            const int id = data[begin + 1 + 2 * k];
            for (Component& c : components)
                if (c.id == id)
                    scan[k] = &c;
            if (!scan[k])
                return false;
            scan[k]->dcTable = data[begin + 2 + 2 * k] >> 4;
            scan[k]->acTable = data[begin + 2 + 2 * k] & 15;
            if (scan[k]->dcTable > 3 || scan[k]->acTable > 3)
                return false;
        }
        int first = data[begin + 1 + 2 * count], last = data[begin + 2 + 2 * count];
        int high = data[begin + 3 + 2 * count] >> 4, low = data[begin + 3 + 2 * count] & 15;
        if (!progressive) {
            first = 0;
            last = 63;
            high = low = 0;
        } else if (first > last || last > 63 || (first == 0 && last != 0) || (first > 0 && count != 1) || low > 13) {
            return false;
        }
        for (int k = 0; k < count; ++k) {
            if (first == 0 && high == 0 && !dc[scan[k]->dcTable].present)
                return false;
            if (last > 0 && !ac[scan[k]->acTable].present)
                return false;
        }

        BitReader in;
        in.data = data;
        in.size = size;
        in.position = end;
        for (Component& c : components)
            c.prediction = 0;
        int endOfBandRun = 0;
        const bool interleaved = count > 1;
        const int unitsWide = interleaved ? mcusWide : scan[0]->blocksWide;
        const int unitsHigh = interleaved ? mcusHigh : scan[0]->blocksHigh;
        int untilRestart = restartInterval;
        for (int unitY = 0; unitY < unitsHigh; ++unitY) {
            for (int unitX = 0; unitX < unitsWide; ++unitX) {
                if (restartInterval && untilRestart == 0) {
                    if (!in.restart())
                        return false;
                    for (Component& c : components)
                        c.prediction = 0;
                    endOfBandRun = 0;
                    untilRestart = restartInterval;
                }
                for (int k = 0; k < count; ++k) {
                    Component& c = *scan[k];
                    const int across = interleaved ? c.h : 1, down = interleaved ? c.v : 1;
                    for (int by = 0; by < down; ++by)
                        for (int bx = 0; bx < across; ++bx) {
                            const int blockX = unitX * across + bx, blockY = unitY * down + by;
                            std::int16_t* block =
                                c.coefficients.data() + (static_cast<std::size_t>(blockY) * c.storedWide + blockX) * 64;
                            if (!decodeBlock(in, c, block, first, last, high, low, endOfBandRun))
                                return false;
                        }
                }
                --untilRestart;
            }
        }
        // the scan ends at the next marker that is not a restart marker
        std::size_t at = in.position;
        while (at + 1 < size && !(data[at] == 0xff && data[at + 1] != 0x00 && (data[at + 1] < 0xd0 || data[at + 1] > 0xd7)))
            ++at;
        position = at;
        return true;
    }

    bool decodeBlock(BitReader& in, Component& c, std::int16_t* block, int first, int last, int high, int low,
                     int& endOfBandRun) {
        if (!progressive) {
            const int t = decodeSymbol(in, dc[c.dcTable]);
            if (t < 0 || t > 15)
                return false;
            const int difference = t ? extend(in.receive(t), t) : 0;
            c.prediction += difference;
            block[0] = static_cast<std::int16_t>(c.prediction);
            for (int k = 1; k < 64;) {
                const int rs = decodeSymbol(in, ac[c.acTable]);
                if (rs < 0)
                    return false;
                const int run = rs >> 4, bits = rs & 15;
                if (bits == 0) {
                    if (run != 15)
                        break;
                    k += 16;
                    continue;
                }
                k += run;
                if (k > 63)
                    return false;
                block[kZigzag[k]] = static_cast<std::int16_t>(extend(in.receive(bits), bits));
                ++k;
            }
            return true;
        }
        if (first == 0) {
            if (high == 0) {  // first pass over the DC coefficient
                const int t = decodeSymbol(in, dc[c.dcTable]);
                if (t < 0 || t > 15)
                    return false;
                const int difference = t ? extend(in.receive(t), t) : 0;
                c.prediction += difference;
                block[0] = static_cast<std::int16_t>(c.prediction * (1 << low));
            } else if (in.receive(1)) {  // one more bit of it
                block[0] = static_cast<std::int16_t>(block[0] | (1 << low));
            }
            return true;
        }
        if (high == 0) {  // first pass over a band of AC coefficients
            if (endOfBandRun > 0) {
                --endOfBandRun;
                return true;
            }
            for (int k = first; k <= last;) {
                const int rs = decodeSymbol(in, ac[c.acTable]);
                if (rs < 0)
                    return false;
                const int run = rs >> 4, bits = rs & 15;
                if (bits == 0) {
                    if (run < 15) {
                        endOfBandRun = (1 << run) - 1;
                        if (run)
                            endOfBandRun += in.receive(run);
                        break;
                    }
                    k += 16;
                    continue;
                }
                k += run;
                if (k > 63)
                    return false;
                block[kZigzag[k]] = static_cast<std::int16_t>(extend(in.receive(bits), bits) * (1 << low));
                ++k;
            }
            return true;
        }
        // refinement of a band: one more bit for the coefficients already non-zero, new
        // coefficients of magnitude one in between
        const int plus = 1 << low, minus = -(1 << low);
        int k = first;
        auto refine = [&](std::int16_t& value) {
            if (in.receive(1) && (value & plus) == 0)
                value = static_cast<std::int16_t>(value + (value >= 0 ? plus : minus));
        };
        if (endOfBandRun == 0) {
            for (; k <= last; ++k) {
                const int rs = decodeSymbol(in, ac[c.acTable]);
                if (rs < 0)
                    return false;
                int run = rs >> 4;
                const int bits = rs & 15;
                int value = 0;
                if (bits) {
                    if (bits != 1)
                        return false;
                    value = in.receive(1) ? plus : minus;
                } else if (run != 15) {
                    endOfBandRun = 1 << run;
                    if (run)
                        endOfBandRun += in.receive(run);
                    break;
                }
                for (; k <= last; ++k) {
                    std::int16_t& coefficient = block[kZigzag[k]];
                    if (coefficient != 0) {
                        refine(coefficient);
                    } else if (--run < 0) {
                        break;
                    }
                }
                if (value) {
                    if (k > last)
                        return false;
                    block[kZigzag[k]] = static_cast<std::int16_t>(value);
                }
            }
        }
        if (endOfBandRun > 0) {
            for (; k <= last; ++k) {
                std::int16_t& coefficient = block[kZigzag[k]];
                if (coefficient != 0)
                    refine(coefficient);
            }
            --endOfBandRun;
        }
        return true;
    }

    bool parse() {
        if (size < 4 || data[0] != 0xff || data[1] != 0xd8)
            return false;
        bool sawScan = false;
        while (position + 4 <= size) {
            if (data[position] != 0xff)
                return false;
            while (position < size && data[position] == 0xff)
                ++position;  // fill bytes
            if (position >= size)
                return false;
            const std::uint8_t marker = data[position++];
            if (marker == 0xd9)
                break;
            if (marker == 0x01 || (marker >= 0xd0 && marker <= 0xd7))
                continue;
            if (position + 2 > size)
                return false;
            const std::size_t length = word(position);
            if (length < 2 || position + length > size)
                return false;
            const std::size_t begin = position + 2, end = position + length;
            if (marker == 0xc0 || marker == 0xc1 || marker == 0xc2) {
                if (!readFrame(marker, begin, end))
                    return false;
                position = end;
            } else if (marker == 0xc3 || (marker >= 0xc5 && marker <= 0xcf && marker != 0xc8 && marker != 0xcc)) {
                return false;  // lossless, hierarchical, arithmetic coding
            } else if (marker == 0xcc) {
                return false;
            } else if (marker == 0xda) {
                if (!readScan(begin, end))
                    return false;
                sawScan = true;
            } else {
                if (!readTables(marker, begin, end))
                    return false;
                position = end;
            }
        }
        return haveFrame && sawScan;
    }

    bool reconstruct(std::vector<std::uint8_t>& rgba) {
        for (Component& c : components) {
            if (!quantPresent[c.quant])
                return false;
            const int stride = c.storedWide * 8;
            c.samples.assign(static_cast<std::size_t>(stride) * c.storedHigh * 8, 0);
            for (int by = 0; by < c.storedHigh; ++by)
                for (int bx = 0; bx < c.storedWide; ++bx)
                    inverseDct(c.coefficients.data() + (static_cast<std::size_t>(by) * c.storedWide + bx) * 64,
                               quant[c.quant], c.samples.data() + static_cast<std::size_t>(by) * 8 * stride + bx * 8,
                               stride);
            c.coefficients.clear();
            c.coefficients.shrink_to_fit();
        }
        // full-resolution planes
        std::vector<std::vector<std::uint8_t>> planes(components.size());
        for (std::size_t k = 0; k < components.size(); ++k)
            if (!upsample(components[k], planes[k]))
                return false;

        rgba.resize(static_cast<std::size_t>(width) * height * 4);
        const bool grey = components.size() == 1;
        bool isRgb = false;
        if (!grey) {
            if (adobeTransform == 0)
                isRgb = true;
            else if (adobeTransform < 0 && components[0].id == 'R' && components[1].id == 'G' && components[2].id == 'B')
                isRgb = true;
        }
        constexpr int kScale = 16;
        constexpr std::int32_t kHalf = 1 << (kScale - 1);
        for (std::size_t i = 0; i < static_cast<std::size_t>(width) * height; ++i) {
            std::uint8_t* out = rgba.data() + 4 * i;
            if (grey) {
                out[0] = out[1] = out[2] = planes[0][i];
            } else if (isRgb) {
                out[0] = planes[0][i];
                out[1] = planes[1][i];
                out[2] = planes[2][i];
            } else {
                // the fixed-point tables of the IJG library: 1.402, 1.772, 0.71414, 0.34414 at 16 bits
                const std::int32_t y = planes[0][i], cb = planes[1][i] - 128, cr = planes[2][i] - 128;
                const std::int32_t red = y + ((91881 * cr + kHalf) >> kScale);
                const std::int32_t blue = y + ((116130 * cb + kHalf) >> kScale);
                const std::int32_t green = y + ((-22554 * cb + kHalf - 46802 * cr) >> kScale);
                out[0] = clampByte(red);
                out[1] = clampByte(green);
                out[2] = clampByte(blue);
            }
            out[3] = 255;
        }
        return true;
    }

    // Brings one component to the image's resolution.  Components at half
    // resolution are interpolated with the 3:1 triangle filter (3/4 of the nearer
    // sample, 1/4 of the farther, image edges replicated, the roundings alternate
    // between the two output positions); other ratios are replicated.
    bool upsample(const Component& c, std::vector<std::uint8_t>& plane) const {
        if (hMax % c.h != 0 || vMax % c.v != 0)
            return false;
        const int fx = hMax / c.h, fy = vMax / c.v;
        const int stride = c.storedWide * 8;
        plane.resize(static_cast<std::size_t>(width) * height);
        auto sample = [&](int x, int y) -> int {
            x = std::min(std::max(x, 0), c.width - 1);
            y = std::min(std::max(y, 0), c.height - 1);
            return c.samples[static_cast<std::size_t>(y) * stride + x];
        };
        for (int y = 0; y < height; ++y) {
            std::uint8_t* row = plane.data() + static_cast<std::size_t>(y) * width;
            if (fx == 1 && fy == 1) {
                for (int x = 0; x < width; ++x)
                    row[x] = static_cast<std::uint8_t>(sample(x, y));
            } else if (fx == 2 && fy == 1) {
                for (int x = 0; x < width; ++x) {
                    const int i = x >> 1;
                    const int near = sample(i, y);
                    row[x] = static_cast<std::uint8_t>((x & 1) ? (near * 3 + sample(i + 1, y) + 2) >> 2
                                                               : (near * 3 + sample(i - 1, y) + 1) >> 2);
                }
            } else if (fx == 1 && fy == 2) {
                const int j = y >> 1, other = (y & 1) ? j + 1 : j - 1, bias = (y & 1) ? 2 : 1;
                for (int x = 0; x < width; ++x)
                    row[x] = static_cast<std::uint8_t>((sample(x, j) * 3 + sample(x, other) + bias) >> 2);
            } else if (fx == 2 && fy == 2) {
                const int j = y >> 1, other = (y & 1) ? j + 1 : j - 1;
                auto columnSum = [&](int i) { return sample(i, j) * 3 + sample(i, other); };
                for (int x = 0; x < width; ++x) {
                    const int i = x >> 1;
                    const int near = columnSum(i);
                    row[x] = static_cast<std::uint8_t>((x & 1) ? (near * 3 + columnSum(i + 1) + 7) >> 4
                                                               : (near * 3 + columnSum(i - 1) + 8) >> 4);
                }
            } else {
                for (int x = 0; x < width; ++x)
                    row[x] = static_cast<std::uint8_t>(sample(x / fx, y / fy));
            }
        }
        return true;
    }
};

}  // namespace

bool decodeJpegImage(const std::vector<std::byte>& file, std::vector<std::uint8_t>& rgba, std::uint32_t& width,
                     std::uint32_t& height) {
    Decoder decoder(reinterpret_cast<const std::uint8_t*>(file.data()), file.size());
    if (!decoder.parse() || !decoder.reconstruct(rgba)) {
        rgba.clear();
        return false;
    }
    width = static_cast<std::uint32_t>(decoder.width);
    height = static_cast<std::uint32_t>(decoder.height);
    return true;
}

}  // namespace rt
