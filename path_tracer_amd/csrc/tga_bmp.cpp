// tga_bmp.cpp -- Truevision TGA, Windows BMP, GIF (first frame) and binary PNM textures as 8-bit RGB, rows top-down.
//
// The reference loads map_Kd files with stbi_load(..., 3) (src/cs488.h:839-846), which also reads these
// formats; scene files in circulation (the publicly distributed sponza.obj among them) name .tga and .bmp
// textures.  Alpha, where a file has it, is dropped without compositing, as a three-channel request to that
// loader does.  Written from the format descriptions; the reference holds no file of these kinds, so parity with
// its decoder is pinned only through files written by an independent encoder in the tests.
#include <cstdio>
#include <cstring>

#include "scene.h"

namespace pt {

namespace {

bool slurp(const char* path, std::vector<uint8_t>& buf) {
    FILE* fp = fopen(path, "rb");
    if (!fp) return false;
    fseek(fp, 0, SEEK_END);
    const long n = ftell(fp);
    fseek(fp, 0, SEEK_SET);
    if (n < 0 || n > (1L << 30)) { fclose(fp); return false; }
    buf.resize((size_t)n);
    const bool ok = fread(buf.data(), 1, buf.size(), fp) == buf.size();
    fclose(fp);
    return ok;
}

unsigned le16(const uint8_t* p) { return (unsigned)p[0] | ((unsigned)p[1] << 8); }
uint32_t le32(const uint8_t* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }

// one TGA pixel of `bytes` bytes (file order B, G, R[, A]; 2 bytes = 1-5-5-5; 1 byte = grey) as RGB
void tga_pixel(const uint8_t* p, int bytes, uint8_t out[3]) {
    if (bytes == 1) { out[0] = out[1] = out[2] = p[0]; return; }
    if (bytes == 2) {
        const unsigned px = le16(p);
        out[0] = (uint8_t)((((px >> 10) & 31) * 255) / 31);
        out[1] = (uint8_t)((((px >> 5) & 31) * 255) / 31);
        out[2] = (uint8_t)(((px & 31) * 255) / 31);
        return;
    }
    out[0] = p[2]; out[1] = p[1]; out[2] = p[0];
}

// Extracts the field a mask selects and scales it to 8 bits.
uint8_t masked(uint32_t v, uint32_t mask) {
    if (!mask) return 0;
    int shift = 0, bits = 0;
    while (!((mask >> shift) & 1u)) shift++;
    while (((mask >> (shift + bits)) & 1u) && shift + bits < 32) bits++;
    uint32_t x = (v & mask) >> shift;
    if (bits >= 8) return (uint8_t)(x >> (bits - 8));
    // replicate the bits down the byte, so that the field's largest value gives 255
    uint32_t out = x << (8 - bits);
    for (int have = bits; have < 8; have += bits) out |= out >> have;
    return (uint8_t)out;
}

}  // namespace

bool looks_like_tga(const uint8_t* h, size_t n) {
    if (n < 18) return false;
    const int cmap = h[1], type = h[2], bpp = h[16];
    if (cmap > 1) return false;
    if (!(type == 1 || type == 2 || type == 3 || type == 9 || type == 10 || type == 11)) return false;
    if ((type == 1 || type == 9) != (cmap == 1)) return false;
    if (cmap == 1 && !(h[7] == 15 || h[7] == 16 || h[7] == 24 || h[7] == 32)) return false;
    if (le16(h + 12) == 0 || le16(h + 14) == 0) return false;
    return bpp == 8 || bpp == 15 || bpp == 16 || bpp == 24 || bpp == 32;
}

bool load_tga_rgb8(const char* path, std::vector<uint8_t>& rgb, int& w, int& h, std::string& err) {
    std::vector<uint8_t> buf;
    if (!slurp(path, buf)) { err = std::string("cannot read ") + path; return false; }
    if (!looks_like_tga(buf.data(), buf.size())) { err = "not a TGA file"; return false; }
    const uint8_t* hd = buf.data();
    const int id_len = hd[0], type = hd[2] & 7, rle = (hd[2] & 8) != 0;
    const int cmap_first = (int)le16(hd + 3), cmap_len = (int)le16(hd + 5), cmap_bits = hd[7];
    w = (int)le16(hd + 12); h = (int)le16(hd + 14);
    const int bpp = hd[16];
    const bool top_down = (hd[17] & 0x20) != 0;
    const bool indexed = type == 1;
    if (indexed && bpp != 8 && bpp != 16) { err = "TGA: colour-mapped image with an index size other than 8 or 16 bits"; return false; }
    if (type == 3 && bpp != 8) { err = "TGA: grey image with other than 8 bits"; return false; }
    const int px_bytes = (bpp + 7) / 8;
    const int cmap_bytes = indexed ? (cmap_bits + 7) / 8 : 0;
    size_t pos = 18 + (size_t)id_len;
    const size_t cmap_size = (size_t)cmap_len * (size_t)((hd[1] ? (cmap_bits + 7) / 8 : 0));
    if (pos + cmap_size > buf.size()) { err = "TGA: truncated colour map"; return false; }
    const uint8_t* cmap = buf.data() + pos;
    pos += cmap_size;
    rgb.assign((size_t)w * h * 3, 0);
    auto put = [&](size_t i, const uint8_t* p) -> bool {  // pixel i in file order
        const size_t x = i % (size_t)w, y_file = i / (size_t)w;
        const size_t y = top_down ? y_file : (size_t)h - 1 - y_file;
        uint8_t* dst = &rgb[(y * (size_t)w + x) * 3];
        if (indexed) {
            const int idx = (px_bytes == 1 ? (int)p[0] : (int)le16(p)) - cmap_first;
            if (idx < 0 || idx >= cmap_len) return false;
            tga_pixel(cmap + (size_t)idx * cmap_bytes, cmap_bytes, dst);
        } else {
            tga_pixel(p, px_bytes, dst);
        }
        return true;
    };
    const size_t total = (size_t)w * h;
    if (!rle) {
        if (pos + total * px_bytes > buf.size()) { err = "TGA: truncated pixel data"; return false; }
        for (size_t i = 0; i < total; i++)
            if (!put(i, buf.data() + pos + i * px_bytes)) { err = "TGA: colour index outside the map"; return false; }
        return true;
    }
    size_t i = 0;
    while (i < total) {
        if (pos >= buf.size()) { err = "TGA: truncated run-length data"; return false; }
        const int head = buf[pos++];
        const size_t run = (size_t)(head & 127) + 1;
        if (i + run > total) { err = "TGA: a run crosses the end of the image"; return false; }
        if (head & 128) {  // one pixel repeated
            if (pos + px_bytes > buf.size()) { err = "TGA: truncated run-length data"; return false; }
            for (size_t k = 0; k < run; k++)
                if (!put(i + k, buf.data() + pos)) { err = "TGA: colour index outside the map"; return false; }
            pos += px_bytes;
        } else {           // literal pixels
            if (pos + run * px_bytes > buf.size()) { err = "TGA: truncated run-length data"; return false; }
            for (size_t k = 0; k < run; k++)
                if (!put(i + k, buf.data() + pos + k * px_bytes)) { err = "TGA: colour index outside the map"; return false; }
            pos += run * px_bytes;
        }
        i += run;
    }
    return true;
}

bool load_bmp_rgb8(const char* path, std::vector<uint8_t>& rgb, int& w, int& h, std::string& err) {
    std::vector<uint8_t> buf;
    if (!slurp(path, buf)) { err = std::string("cannot read ") + path; return false; }
    if (buf.size() < 26 || buf[0] != 'B' || buf[1] != 'M') { err = "not a BMP file"; return false; }
    const uint32_t data_off = le32(&buf[10]);
    const uint32_t hsize = le32(&buf[14]);
    int bpp = 0;
    uint32_t compression = 0;
    bool top_down = false;
    if (hsize == 12) {  // OS/2 1.x header: 16-bit sizes, 3-byte palette entries
        w = (int)le16(&buf[18]); h = (int)le16(&buf[20]);
        bpp = (int)le16(&buf[24]);
    } else if (hsize == 40 || hsize == 52 || hsize == 56 || hsize == 108 || hsize == 124) {
        if (buf.size() < 14 + (size_t)hsize) { err = "BMP: truncated header"; return false; }
        const int32_t sw = (int32_t)le32(&buf[18]), sh = (int32_t)le32(&buf[22]);
        w = sw; h = sh < 0 ? -sh : sh;
        top_down = sh < 0;
        bpp = (int)le16(&buf[28]);
        compression = le32(&buf[30]);
    } else { err = "BMP: unknown header size"; return false; }
    if (w <= 0 || h <= 0 || w > (1 << 16) || h > (1 << 16)) { err = "BMP: bad dimensions"; return false; }
    if (compression == 1 || compression == 2) { err = "BMP: run-length compressed files are not supported"; return false; }
    if (compression != 0 && compression != 3) { err = "BMP: unknown compression"; return false; }
    uint32_t mr = 0, mg = 0, mb = 0;
    if (bpp == 16 || bpp == 32) {
        if (compression == 3) {
            const size_t at = hsize == 40 ? 54 : 14 + 40;  // after a 40-byte header, inside the longer ones
            if (buf.size() < at + 12) { err = "BMP: truncated bit masks"; return false; }
            mr = le32(&buf[at]); mg = le32(&buf[at + 4]); mb = le32(&buf[at + 8]);
        } else if (bpp == 16) { mr = 31u << 10; mg = 31u << 5; mb = 31u; }
        else { mr = 0xffu << 16; mg = 0xffu << 8; mb = 0xffu; }
        if (!mr || !mg || !mb) { err = "BMP: empty bit mask"; return false; }
    } else if (compression != 0) { err = "BMP: bit masks with other than 16 or 32 bits per pixel"; return false; }
    if (!(bpp == 1 || bpp == 4 || bpp == 8 || bpp == 16 || bpp == 24 || bpp == 32)) { err = "BMP: unsupported bit depth"; return false; }
    // palette: between the headers and the pixel data
    const int entry = hsize == 12 ? 3 : 4;
    const size_t pal_at = 14 + (size_t)hsize + (hsize == 40 && compression == 3 ? 12 : 0);
    int pal_n = 0;
    if (bpp <= 8) {
        uint32_t used = hsize == 12 ? 0 : le32(&buf[46]);
        pal_n = used ? (int)used : (1 << bpp);
        if (pal_n > 256 || pal_at + (size_t)pal_n * entry > buf.size()) { err = "BMP: truncated palette"; return false; }
    }
    const size_t stride = (((size_t)w * bpp + 31) / 32) * 4;
    if ((size_t)data_off + stride * (size_t)h > buf.size()) { err = "BMP: truncated pixel data"; return false; }
    rgb.assign((size_t)w * h * 3, 0);
    for (int y = 0; y < h; y++) {
        const uint8_t* row = buf.data() + data_off + stride * (size_t)(top_down ? y : h - 1 - y);
        uint8_t* dst = &rgb[(size_t)y * w * 3];
        for (int x = 0; x < w; x++) {
            if (bpp <= 8) {
                int idx;
                if (bpp == 8) idx = row[x];
                else if (bpp == 4) idx = (row[x >> 1] >> ((x & 1) ? 0 : 4)) & 15;
                else idx = (row[x >> 3] >> (7 - (x & 7))) & 1;
                if (idx >= pal_n) { err = "BMP: colour index outside the palette"; return false; }
                const uint8_t* e = buf.data() + pal_at + (size_t)idx * entry;
                dst[3 * x] = e[2]; dst[3 * x + 1] = e[1]; dst[3 * x + 2] = e[0];
            } else if (bpp == 24) {
                dst[3 * x] = row[3 * x + 2]; dst[3 * x + 1] = row[3 * x + 1]; dst[3 * x + 2] = row[3 * x];
            } else {
                const uint32_t v = bpp == 16 ? le16(row + 2 * x) : le32(row + 4 * x);
                dst[3 * x] = masked(v, mr); dst[3 * x + 1] = masked(v, mg); dst[3 * x + 2] = masked(v, mb);
            }
        }
    }
    return true;
}

// GIF: the first image of the file on a black canvas of the logical screen's size.  Pixels outside the image's
// rectangle and pixels of the transparent colour stay black (a decoder that returns the first frame uncomposited
// leaves them at zero).  Interlaced images are put back in order.
bool load_gif_rgb8(const char* path, std::vector<uint8_t>& rgb, int& w, int& h, std::string& err) {
    std::vector<uint8_t> buf;
    if (!slurp(path, buf)) { err = std::string("cannot read ") + path; return false; }
    if (buf.size() < 13 || memcmp(buf.data(), "GIF8", 4) != 0 || (buf[4] != '7' && buf[4] != '9') || buf[5] != 'a') { err = "not a GIF file"; return false; }
    w = (int)le16(&buf[6]); h = (int)le16(&buf[8]);
    if (w <= 0 || h <= 0) { err = "GIF: empty logical screen"; return false; }
    size_t pos = 13;
    const uint8_t* global = nullptr;
    int global_n = 0;
    if (buf[10] & 0x80) {
        global_n = 2 << (buf[10] & 7);
        if (pos + (size_t)global_n * 3 > buf.size()) { err = "GIF: truncated colour table"; return false; }
        global = &buf[pos];
        pos += (size_t)global_n * 3;
    }
    int transparent = -1;
    while (pos < buf.size()) {
        const uint8_t tag = buf[pos++];
        if (tag == 0x21) {  // extension: label, then sub-blocks; the graphic control extension names the transparent colour
            if (pos >= buf.size()) break;
            const uint8_t label = buf[pos++];
            bool first = true;
            while (pos < buf.size()) {
                const size_t n = buf[pos++];
                if (n == 0) break;
                if (pos + n > buf.size()) { err = "GIF: truncated extension"; return false; }
                if (label == 0xF9 && first && n >= 4) transparent = (buf[pos] & 1) ? (int)buf[pos + 3] : -1;
                first = false;
                pos += n;
            }
        } else if (tag == 0x2C) {
            if (pos + 9 > buf.size()) { err = "GIF: truncated image descriptor"; return false; }
            const int ix = (int)le16(&buf[pos]), iy = (int)le16(&buf[pos + 2]), iw = (int)le16(&buf[pos + 4]), ih = (int)le16(&buf[pos + 6]);
            const uint8_t flags = buf[pos + 8];
            pos += 9;
            const uint8_t* table = global;
            int table_n = global_n;
            if (flags & 0x80) {
                table_n = 2 << (flags & 7);
                if (pos + (size_t)table_n * 3 > buf.size()) { err = "GIF: truncated colour table"; return false; }
                table = &buf[pos];
                pos += (size_t)table_n * 3;
            }
            if (!table) { err = "GIF: image without a colour table"; return false; }
            if (ix + iw > w || iy + ih > h) { err = "GIF: image outside the logical screen"; return false; }
            if (pos >= buf.size()) { err = "GIF: truncated image data"; return false; }
            const int min_bits = buf[pos++];
            if (min_bits < 1 || min_bits > 11) { err = "GIF: bad code size"; return false; }
            std::vector<uint8_t> data;  // the sub-blocks joined
            while (pos < buf.size()) {
                const size_t n = buf[pos++];
                if (n == 0) break;
                if (pos + n > buf.size()) { err = "GIF: truncated image data"; return false; }
                data.insert(data.end(), &buf[pos], &buf[pos] + n);
                pos += n;
            }
            // LZW with variable code width (least significant bit first), at most 4096 codes
            std::vector<uint8_t> idx;
            idx.reserve((size_t)iw * ih);
            const int clear = 1 << min_bits, stop = clear + 1;
            std::vector<int> prefix(4096, -1);
            std::vector<uint8_t> suffix(4096, 0), first_of(4096, 0), stack;
            for (int k = 0; k < clear; k++) { suffix[k] = (uint8_t)k; first_of[k] = (uint8_t)k; }
            int width = min_bits + 1, next = stop + 1, prev = -1;
            uint32_t acc = 0;
            int have = 0;
            size_t at = 0;
            const size_t total = (size_t)iw * ih;
            while (idx.size() < total) {
                while (have < width && at < data.size()) { acc |= (uint32_t)data[at++] << have; have += 8; }
                if (have < width) break;
                const int code = (int)(acc & ((1u << width) - 1));
                acc >>= width; have -= width;
                if (code == clear) { width = min_bits + 1; next = stop + 1; prev = -1; continue; }
                if (code == stop) break;
                int c = code;
                stack.clear();
                if (code >= next) {  // the code being defined by this very step: previous string + its own first pixel
                    if (prev < 0 || code > next) { err = "GIF: corrupt code stream"; return false; }
                    stack.push_back(first_of[prev]);
                    c = prev;
                }
                while (c >= 0 && stack.size() <= 4096) { stack.push_back(suffix[c]); c = prefix[c]; }
                const uint8_t head = stack.back();
                for (size_t k = stack.size(); k-- > 0 && idx.size() < total;) idx.push_back(stack[k]);
                if (prev >= 0 && next < 4096) {
                    prefix[next] = prev; suffix[next] = head; first_of[next] = first_of[prev];
                    next++;
                    if (next == (1 << width) && width < 12) width++;
                }
                prev = code;
            }
            rgb.assign((size_t)w * h * 3, 0);
            // rows of an interlaced image come in four passes: every 8th from 0, every 8th from 4, every 4th from 2, every 2nd from 1
            std::vector<int> row_of((size_t)ih);
            if (flags & 0x40) {
                int r = 0;
                static const int start[4] = {0, 4, 2, 1}, step[4] = {8, 8, 4, 2};
                for (int p = 0; p < 4; p++) for (int y = start[p]; y < ih; y += step[p]) row_of[(size_t)r++] = y;
            } else {
                for (int y = 0; y < ih; y++) row_of[(size_t)y] = y;
            }
            for (size_t k = 0; k < idx.size(); k++) {
                const int x = ix + (int)(k % (size_t)iw), y = iy + row_of[k / (size_t)iw];
                const int c = idx[k];
                if (c == transparent || c >= table_n) continue;
                uint8_t* dst = &rgb[((size_t)y * w + x) * 3];
                dst[0] = table[3 * c]; dst[1] = table[3 * c + 1]; dst[2] = table[3 * c + 2];
            }
            return true;
        } else if (tag == 0x3B) {
            break;
        } else {
            err = "GIF: unknown block"; return false;
        }
    }
    err = "GIF: no image";
    return false;
}

// Binary PGM (P5) and PPM (P6), 8 or 16 bits per sample (a 16-bit sample keeps its high byte).
bool load_pnm_rgb8(const char* path, std::vector<uint8_t>& rgb, int& w, int& h, std::string& err) {
    std::vector<uint8_t> buf;
    if (!slurp(path, buf)) { err = std::string("cannot read ") + path; return false; }
    if (buf.size() < 3 || buf[0] != 'P' || (buf[1] != '5' && buf[1] != '6')) { err = "not a binary PGM or PPM file"; return false; }
    const int channels = buf[1] == '6' ? 3 : 1;
    size_t pos = 2;
    auto number = [&](int& out) -> bool {  // white space and # comments, then decimal digits
        while (pos < buf.size()) {
            if (buf[pos] == '#') { while (pos < buf.size() && buf[pos] != '\n' && buf[pos] != '\r') pos++; }
            else if (buf[pos] == ' ' || buf[pos] == '\t' || buf[pos] == '\n' || buf[pos] == '\r' || buf[pos] == '\f' || buf[pos] == '\v') pos++;
            else break;
        }
        if (pos >= buf.size() || buf[pos] < '0' || buf[pos] > '9') return false;
        long v = 0;
        while (pos < buf.size() && buf[pos] >= '0' && buf[pos] <= '9' && v < (1L << 24)) v = v * 10 + (buf[pos++] - '0');
        out = (int)v;
        return true;
    };
    int maxval = 0;
    if (!number(w) || !number(h) || !number(maxval)) { err = "PNM: bad header"; return false; }
    if (w <= 0 || h <= 0 || w > (1 << 16) || h > (1 << 16) || maxval <= 0 || maxval > 65535) { err = "PNM: bad dimensions"; return false; }
    pos++;  // the single white-space byte after the maximum value
    const int bytes = maxval > 255 ? 2 : 1;
    const size_t need = (size_t)w * h * channels * bytes;
    if (pos + need > buf.size()) { err = "PNM: truncated pixel data"; return false; }
    rgb.resize((size_t)w * h * 3);
    const uint8_t* src = buf.data() + pos;
    for (size_t i = 0; i < (size_t)w * h; i++)
        for (int c = 0; c < 3; c++) rgb[3 * i + c] = src[(i * channels + (channels == 3 ? c : 0)) * bytes];
    return true;
}

}  // namespace pt
