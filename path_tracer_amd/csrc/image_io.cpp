// image_io.cpp -- the few image formats the path tracer touches: Radiance .hdr in (environment
// probes, Image::load src/cs488.h:200-218), PNG in (map_Kd textures, src/cs488.h:839-846),
// PNG out (Image::save src/cs488.h:219-232) and PFM out (raw float dumps for comparisons).
// The reference delegates all of this to a third-party decoder; only the resulting texel values
// matter to the renderer: RGBE texels decode to mantissa * 2^(exponent-136) (zero when the
// exponent byte is zero), textures become 8-bit RGB with alpha dropped.
#include <zlib.h>

#include <cstdio>
#include <algorithm>
#include <cstring>

#include "scene.h"

namespace pt {

namespace {

bool read_file(const char* path, std::vector<uint8_t>& out) {
    FILE* fp = fopen(path, "rb");
    if (!fp) return false;
    fseek(fp, 0, SEEK_END);
    long n = ftell(fp);
    fseek(fp, 0, SEEK_SET);
    out.resize(n > 0 ? (size_t)n : 0);
    size_t got = out.empty() ? 0 : fread(out.data(), 1, out.size(), fp);
    fclose(fp);
    return got == out.size();
}

inline void rgbe_to_float(const uint8_t* p, float* dst) {
    if (p[3] != 0) {
        const float scale = ldexpf(1.0f, (int)p[3] - 136);
        dst[0] = p[0] * scale; dst[1] = p[1] * scale; dst[2] = p[2] * scale;
    } else {
        dst[0] = dst[1] = dst[2] = 0.0f;
    }
}

uint32_t be32(const uint8_t* p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }

void put_be32(std::vector<uint8_t>& v, uint32_t x) {
    v.push_back((uint8_t)(x >> 24)); v.push_back((uint8_t)(x >> 16)); v.push_back((uint8_t)(x >> 8)); v.push_back((uint8_t)x);
}

void put_chunk(std::vector<uint8_t>& file, const char tag[4], const uint8_t* data, size_t n) {
    put_be32(file, (uint32_t)n);
    size_t start = file.size();
    file.insert(file.end(), tag, tag + 4);
    if (n) file.insert(file.end(), data, data + n);
    uint32_t crc = (uint32_t)crc32(0L, file.data() + start, (uInt)(n + 4));
    put_be32(file, crc);
}

int paeth(int a, int b, int c) {
    int p = a + b - c;
    int pa = p > a ? p - a : a - p, pb = p > b ? p - b : b - p, pc = p > c ? p - c : c - p;
    if (pa <= pb && pa <= pc) return a;
    return pb <= pc ? b : c;
}

}  // namespace

bool load_hdr(const char* path, HostImage& img, std::string& err) {
    std::vector<uint8_t> buf;
    if (!read_file(path, buf)) { err = std::string("Unable to load: ") + path; return false; }
    size_t pos = 0;
    auto next_line = [&](std::string& s) -> bool {
        s.clear();
        if (pos >= buf.size()) return false;
        while (pos < buf.size() && buf[pos] != '\n') s.push_back((char)buf[pos++]);
        if (pos < buf.size()) pos++;
        return true;
    };
    std::string line;
    if (!next_line(line) || (line != "#?RADIANCE" && line != "#?RGBE")) { err = "not a Radiance file"; return false; }
    bool fmt = false;
    while (next_line(line)) {
        if (line.empty()) break;
        if (line == "FORMAT=32-bit_rle_rgbe") fmt = true;
    }
    if (!fmt) { err = "unsupported Radiance format"; return false; }
    int w = 0, h = 0;
    if (!next_line(line) || sscanf(line.c_str(), "-Y %d +X %d", &h, &w) != 2 || w <= 0 || h <= 0) {
        err = "unsupported Radiance orientation";
        return false;
    }
    if (w > 32768 || h > 32768) { err = "Radiance image larger than 32768 pixels a side"; return false; }
    std::vector<uint8_t> texels((size_t)w * h * 4);
    const uint8_t* p = buf.data() + pos;
    const uint8_t* end = buf.data() + buf.size();
    bool rle = w >= 8 && w < 32768 && end - p >= 4 && p[0] == 2 && p[1] == 2 && !(p[2] & 0x80);
    if (!rle) {
        if ((size_t)(end - p) < texels.size()) { err = "truncated Radiance file"; return false; }
        memcpy(texels.data(), p, texels.size());
    } else {
        std::vector<uint8_t> scan((size_t)w * 4);
        for (int y = 0; y < h; y++) {
            if (end - p < 4 || p[0] != 2 || p[1] != 2 || (((int)p[2] << 8) | p[3]) != w) { err = "corrupt Radiance scanline"; return false; }
            p += 4;
            for (int c = 0; c < 4; c++) {
                int x = 0;
                while (x < w) {
                    if (p >= end) { err = "truncated Radiance file"; return false; }
                    int count = *p++;
                    if (count > 128) {
                        count -= 128;
                        if (p >= end || x + count > w) { err = "corrupt Radiance run"; return false; }
                        uint8_t v = *p++;
                        for (int k = 0; k < count; k++) scan[(size_t)(x++) * 4 + c] = v;
                    } else {
                        if (count == 0 || end - p < count || x + count > w) { err = "corrupt Radiance run"; return false; }
                        for (int k = 0; k < count; k++) scan[(size_t)(x++) * 4 + c] = *p++;
                    }
                }
            }
            memcpy(texels.data() + (size_t)y * w * 4, scan.data(), scan.size());
        }
    }
    img.w = w; img.h = h;
    img.rgb.resize((size_t)w * h * 3);
    for (int y = 0; y < h; y++) {       // file row y (top first) lands in stored row h-1-y
        float* dst = img.rgb.data() + (size_t)(h - 1 - y) * w * 3;
        const uint8_t* src = texels.data() + (size_t)y * w * 4;
        for (int x = 0; x < w; x++) rgbe_to_float(src + 4 * x, dst + 3 * x);
    }
    return true;
}

bool load_png_rgb8(const char* path, std::vector<uint8_t>& rgb, int& w, int& h, std::string& err) {
    std::vector<uint8_t> buf;
    if (!read_file(path, buf)) { err = std::string("cannot read ") + path; return false; }
    static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
    if (buf.size() < 8 || memcmp(buf.data(), sig, 8) != 0) { err = "not a PNG file"; return false; }
    size_t pos = 8;
    int depth = 0, ctype = 0, interlace = 0;
    std::vector<uint8_t> idat, palette;
    bool have_ihdr = false;
    while (pos + 12 <= buf.size()) {
        uint32_t n = be32(&buf[pos]);
        const uint8_t* tag = &buf[pos + 4];
        const uint8_t* data = &buf[pos + 8];
        if (pos + 12 + (size_t)n > buf.size()) { err = "truncated PNG chunk"; return false; }
        if (!memcmp(tag, "IHDR", 4) && n >= 13) {
            w = (int)be32(data); h = (int)be32(data + 4);
            depth = data[8]; ctype = data[9]; interlace = data[12];
            have_ihdr = true;
        } else if (!memcmp(tag, "PLTE", 4)) {
            palette.assign(data, data + n);
        } else if (!memcmp(tag, "IDAT", 4)) {
            idat.insert(idat.end(), data, data + n);
        } else if (!memcmp(tag, "IEND", 4)) {
            break;
        }
        pos += 12 + (size_t)n;
    }
    if (!have_ihdr || w <= 0 || h <= 0) { err = "PNG without header"; return false; }
    if (w > 32768 || h > 32768) { err = "PNG larger than 32768 pixels a side"; return false; }
    if (interlace > 1) { err = "unknown PNG interlace method"; return false; }
    int channels = ctype == 0 ? 1 : ctype == 2 ? 3 : ctype == 3 ? 1 : ctype == 4 ? 2 : ctype == 6 ? 4 : 0;
    if (!channels || (depth != 1 && depth != 2 && depth != 4 && depth != 8 && depth != 16)) { err = "unsupported PNG layout"; return false; }
    const size_t bits_pp = (size_t)channels * depth;
    const size_t bpp = bits_pp >= 8 ? bits_pp / 8 : 1;
    // The image as one pass, or as the seven reduced images of Adam7 interlacing: pass p holds the pixels
    // (x0 + i dx, y0 + j dy), each pass filtered as an image of its own; empty passes have no bytes.
    struct Pass { int x0, y0, dx, dy; };
    static const Pass adam7[7] = {{0, 0, 8, 8}, {4, 0, 8, 8}, {0, 4, 4, 8}, {2, 0, 4, 4}, {0, 2, 2, 4}, {1, 0, 2, 2}, {0, 1, 1, 2}};
    static const Pass whole = {0, 0, 1, 1};
    const Pass* passes = interlace ? adam7 : &whole;
    const int n_passes = interlace ? 7 : 1;
    auto pass_w = [&](const Pass& ps) { return (w - ps.x0 + ps.dx - 1) / ps.dx; };
    auto pass_h = [&](const Pass& ps) { return (h - ps.y0 + ps.dy - 1) / ps.dy; };
    size_t raw_size = 0;
    for (int k = 0; k < n_passes; k++) {
        const int pw = pass_w(passes[k]), ph = pass_h(passes[k]);
        if (pw > 0 && ph > 0) raw_size += (((size_t)pw * bits_pp + 7) / 8 + 1) * (size_t)ph;
    }
    std::vector<uint8_t> raw(raw_size);
    uLongf raw_len = (uLongf)raw.size();
    if (uncompress(raw.data(), &raw_len, idat.data(), (uLong)idat.size()) != Z_OK || raw_len != raw.size()) {
        err = "PNG data does not inflate";
        return false;
    }
    rgb.resize((size_t)w * h * 3);
    const int scale = depth == 1 ? 255 : depth == 2 ? 85 : depth == 4 ? 17 : 1;
    size_t at = 0;
    for (int k = 0; k < n_passes; k++) {
        const Pass& ps = passes[k];
        const int pw = pass_w(ps), ph = pass_h(ps);
        if (pw <= 0 || ph <= 0) continue;
        const size_t stride = ((size_t)pw * bits_pp + 7) / 8;
        std::vector<uint8_t> prev(stride, 0);
        for (int j = 0; j < ph; j++) {
            uint8_t* row = raw.data() + at;
            at += stride + 1;
            const int ft = row[0];
            uint8_t* cur = row + 1;
            for (size_t i = 0; i < stride; i++) {  // undo the scanline filter in place
                int a = i >= bpp ? cur[i - bpp] : 0, b = prev[i], c = i >= bpp ? prev[i - bpp] : 0;
                int add = ft == 0 ? 0 : ft == 1 ? a : ft == 2 ? b : ft == 3 ? (a + b) / 2 : ft == 4 ? paeth(a, b, c) : -1;
                if (add < 0) { err = "bad PNG filter"; return false; }
                cur[i] = (uint8_t)(cur[i] + add);
            }
            memcpy(prev.data(), cur, stride);
            const int y = ps.y0 + j * ps.dy;
            for (int i = 0; i < pw; i++) {
                uint8_t* dst = rgb.data() + ((size_t)y * w + (size_t)(ps.x0 + i * ps.dx)) * 3;
                auto sample = [&](int ch) -> int {  // channel value; 16-bit samples keep their high byte
                    if (depth == 16) return cur[((size_t)i * channels + ch) * 2];
                    if (depth == 8) return cur[(size_t)i * channels + ch];
                    size_t bit = (size_t)i * depth;  // sub-byte depths have one channel
                    return (cur[bit / 8] >> (8 - depth - (bit % 8))) & ((1 << depth) - 1);
                };
                if (ctype == 3) {
                    size_t idx = (size_t)sample(0) * 3;
                    for (int c = 0; c < 3; c++) dst[c] = idx + c < palette.size() ? palette[idx + c] : 0;
                } else if (ctype == 0 || ctype == 4) {
                    uint8_t g = (uint8_t)(depth < 8 ? sample(0) * scale : sample(0));
                    dst[0] = dst[1] = dst[2] = g;
                } else {
                    for (int c = 0; c < 3; c++) dst[c] = (uint8_t)sample(c);
                }
            }
        }
    }
    return true;
}

bool save_png_rgb8(const char* path, const uint8_t* rgb, int w, int h, std::string& err) {
    std::vector<uint8_t> raw;
    raw.reserve(((size_t)w * 3 + 1) * h);
    for (int y = 0; y < h; y++) {
        raw.push_back(0);
        raw.insert(raw.end(), rgb + (size_t)y * w * 3, rgb + (size_t)(y + 1) * w * 3);
    }
    uLongf zlen = compressBound((uLong)raw.size());
    std::vector<uint8_t> z(zlen);
    if (compress2(z.data(), &zlen, raw.data(), (uLong)raw.size(), 6) != Z_OK) { err = "deflate failed"; return false; }
    std::vector<uint8_t> file;
    static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
    file.insert(file.end(), sig, sig + 8);
    std::vector<uint8_t> ihdr;
    put_be32(ihdr, (uint32_t)w); put_be32(ihdr, (uint32_t)h);
    ihdr.push_back(8); ihdr.push_back(2); ihdr.push_back(0); ihdr.push_back(0); ihdr.push_back(0);
    put_chunk(file, "IHDR", ihdr.data(), ihdr.size());
    put_chunk(file, "IDAT", z.data(), zlen);
    put_chunk(file, "IEND", nullptr, 0);
    FILE* fp = fopen(path, "wb");
    if (!fp) { err = std::string("cannot write ") + path; return false; }
    bool ok = fwrite(file.data(), 1, file.size(), fp) == file.size();
    fclose(fp);
    if (!ok) err = std::string("short write to ") + path;
    return ok;
}

bool save_pfm(const char* path, const float* rgb_bottom_up, int w, int h, std::string& err) {
    FILE* fp = fopen(path, "wb");
    if (!fp) { err = std::string("cannot write ") + path; return false; }
    fprintf(fp, "PF\n%d %d\n-1.0\n", w, h);  // negative scale = little endian; PFM rows go bottom-up
    bool ok = fwrite(rgb_bottom_up, sizeof(float), (size_t)w * h * 3, fp) == (size_t)w * h * 3;
    fclose(fp);
    if (!ok) err = std::string("short write to ") + path;
    return ok;
}

void quantise_image(const float* rgb_bottom_up, int w, int h, std::vector<uint8_t>& out) {
    out.resize((size_t)w * h * 3);
    size_t k = 0;
    for (int j = h - 1; j >= 0; j--) {
        for (int i = 0; i < w; i++) {
            for (int c = 0; c < 3; c++) {
                float r = rgb_bottom_up[((size_t)i + (size_t)j * w) * 3 + c];
                float clamped = r < 1.0f ? r : 1.0f;   // min(1, r) then max(.., 0): NaN fails r < 1 and ends as 255
                clamped = clamped > 0.0f ? clamped : 0.0f;
                out[k++] = (uint8_t)(255.0f * clamped);  // gamma 1.0 is the identity (src/cs488.h:160-164)
            }
        }
    }
}

// ---- animated GIF out (the `F` recording of the reference's window, src/cs488.h:255-266, 2297-2310) ----
// GIF89a, one global palette of 6 x 7 x 6 colour levels with a 4 x 4 ordered dither, every frame the
// whole image, LZW with the usual variable code width.  (The reference hands its frames to gif.h,
// which builds a palette per frame; only the container and the frame content are kept here.)
namespace {

const int GIF_R = 6, GIF_G = 7, GIF_B = 6;

void put_le16(std::vector<uint8_t>& v, int x) { v.push_back((uint8_t)(x & 255)); v.push_back((uint8_t)((x >> 8) & 255)); }

struct BitWriter {
    std::vector<uint8_t>& out;
    uint8_t block[255];
    int fill = 0;
    uint32_t acc = 0;
    int bits = 0;
    explicit BitWriter(std::vector<uint8_t>& o) : out(o) {}
    void byte(uint8_t b) {
        block[fill++] = b;
        if (fill == 255) flush_block();
    }
    void flush_block() {
        if (!fill) return;
        out.push_back((uint8_t)fill);
        out.insert(out.end(), block, block + fill);
        fill = 0;
    }
    void code(uint32_t c, int width) {
        acc |= c << bits;
        bits += width;
        while (bits >= 8) { byte((uint8_t)(acc & 255)); acc >>= 8; bits -= 8; }
    }
    void finish() {
        if (bits > 0) byte((uint8_t)(acc & 255));
        acc = 0; bits = 0;
        flush_block();
    }
};

// one image's indices -> LZW sub-blocks (minimum code size 8)
void gif_lzw(const std::vector<uint8_t>& idx, std::vector<uint8_t>& out) {
    const int min_size = 8, clear = 1 << min_size, eoi = clear + 1;
    std::vector<uint16_t> next((size_t)4096 * 256, 0);  // next[code * 256 + byte] = longer string's code, 0 = none
    BitWriter bw(out);
    out.push_back((uint8_t)min_size);
    int width = min_size + 1, last = eoi;
    bw.code((uint32_t)clear, width);
    int cur = -1;
    for (uint8_t px : idx) {
        if (cur < 0) { cur = px; continue; }
        const uint16_t longer = next[(size_t)cur * 256 + px];
        if (longer) { cur = longer; continue; }
        bw.code((uint32_t)cur, width);
        next[(size_t)cur * 256 + px] = (uint16_t)++last;
        if (last >= (1 << width)) width++;
        if (last == 4095) {  // table full: start over
            bw.code((uint32_t)clear, width);
            std::fill(next.begin(), next.end(), (uint16_t)0);
            width = min_size + 1;
            last = eoi;
        }
        cur = px;
    }
    if (cur >= 0) bw.code((uint32_t)cur, width);
    bw.code((uint32_t)clear, width);
    bw.code((uint32_t)eoi, min_size + 1);
    bw.finish();
    out.push_back(0);  // block terminator
}

}  // namespace

bool GifWriter::begin(const char* path, int w, int h, int delay_cs, std::string& err) {
    end();
    if (w <= 0 || h <= 0 || w > 65535 || h > 65535) { err = "image size does not fit a GIF"; return false; }
    fp_ = fopen(path, "wb");
    if (!fp_) { err = std::string("cannot write ") + path; return false; }
    w_ = w; h_ = h; delay_ = delay_cs;
    std::vector<uint8_t> hd;
    const char sig[] = "GIF89a";
    hd.insert(hd.end(), sig, sig + 6);
    put_le16(hd, w); put_le16(hd, h);
    hd.push_back(0xF7); hd.push_back(0); hd.push_back(0);  // global table of 256 entries, 8 bits per channel
    for (int i = 0; i < 256; i++) {
        const int r = i / (GIF_G * GIF_B), g = (i / GIF_B) % GIF_G, b = i % GIF_B;
        const bool used = i < GIF_R * GIF_G * GIF_B;
        hd.push_back(used ? (uint8_t)(r * 255 / (GIF_R - 1)) : 0);
        hd.push_back(used ? (uint8_t)(g * 255 / (GIF_G - 1)) : 0);
        hd.push_back(used ? (uint8_t)(b * 255 / (GIF_B - 1)) : 0);
    }
    const uint8_t loop[] = {0x21, 0xFF, 0x0B, 'N', 'E', 'T', 'S', 'C', 'A', 'P', 'E', '2', '.', '0', 3, 1, 0, 0, 0};
    hd.insert(hd.end(), loop, loop + sizeof(loop));
    if (fwrite(hd.data(), 1, hd.size(), fp_) != hd.size()) { err = "short write"; end(); return false; }
    return true;
}

bool GifWriter::frame(const uint8_t* rgb_top_down, std::string& err) {
    if (!fp_) { err = "no GIF is open"; return false; }
    static const int bayer[16] = {0, 8, 2, 10, 12, 4, 14, 6, 3, 11, 1, 9, 15, 7, 13, 5};
    std::vector<uint8_t> idx((size_t)w_ * h_);
    for (int y = 0; y < h_; y++) {
        for (int x = 0; x < w_; x++) {
            const uint8_t* p = rgb_top_down + ((size_t)y * w_ + x) * 3;
            const int t = bayer[(y & 3) * 4 + (x & 3)];  // 0..15: threshold of the ordered dither
            auto level = [&](int v, int n) {
                const int scaled = v * (n - 1) * 16 / 255;  // level in sixteenths
                int q = scaled / 16 + ((scaled % 16) > t ? 1 : 0);
                return q > n - 1 ? n - 1 : q;
            };
            idx[(size_t)y * w_ + x] = (uint8_t)((level(p[0], GIF_R) * GIF_G + level(p[1], GIF_G)) * GIF_B + level(p[2], GIF_B));
        }
    }
    std::vector<uint8_t> out;
    const uint8_t gce[] = {0x21, 0xF9, 4, 0x04, (uint8_t)(delay_ & 255), (uint8_t)((delay_ >> 8) & 255), 0, 0};
    out.insert(out.end(), gce, gce + sizeof(gce));
    out.push_back(0x2C);
    put_le16(out, 0); put_le16(out, 0); put_le16(out, w_); put_le16(out, h_);
    out.push_back(0);
    gif_lzw(idx, out);
    if (fwrite(out.data(), 1, out.size(), fp_) != out.size()) { err = "short write"; return false; }
    frames_++;
    return true;
}

void GifWriter::end() {
    if (!fp_) return;
    fputc(0x3B, fp_);
    fclose(fp_);
    fp_ = nullptr;
    frames_ = 0;
}

}  // namespace pt
