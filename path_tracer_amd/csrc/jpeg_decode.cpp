// jpeg_decode.cpp -- JPEG in, for map_Kd textures (the reference loads them through stbi_load,
// src/cs488.h:839-846, which reads baseline and progressive files).  Sequential (SOF0/SOF1) and
// progressive (SOF2) Huffman-coded files with 8-bit samples, one component (grey) or three (YCbCr),
// any sampling factors up to 4, restart intervals.  Arithmetic follows ITU-T T.81: entropy decoding
// as in Annex F / G, dequantisation, a floating-point inverse DCT, chroma planes subsampled by two
// enlarged with the usual triangle filter, JFIF colour conversion.  (stb_image's integer IDCT differs
// from this in the last one or two levels, as decoders do among themselves.)
#include <cmath>
#include <cstdio>
#include <cstring>

#include "scene.h"

namespace pt {

namespace {

const uint8_t ZIGZAG[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                            41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                            30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

struct HuffTable {
    bool present = false;
    uint8_t values[256];
    int min_code[17], max_code[18], first_index[17];
    // T.81 Annex C: codes of each length are consecutive, lengths in increasing order
    bool build(const uint8_t counts[16], const uint8_t* vals, int n_vals) {
        if (n_vals > 256) return false;
        memcpy(values, vals, (size_t)n_vals);
        int code = 0, k = 0;
        for (int len = 1; len <= 16; len++) {
            first_index[len] = k;
            min_code[len] = code;
            code += counts[len - 1];
            k += counts[len - 1];
            max_code[len] = counts[len - 1] ? code - 1 : -1;
            if (code > (1 << len)) return false;
            code <<= 1;
        }
        max_code[17] = 0x7fffffff;
        present = k == n_vals;
        return present;
    }
};

struct Component {
    int id = 0, h = 1, v = 1, tq = 0;
    int td = 0, ta = 0;          // tables of the current scan
    int blocks_w = 0, blocks_h = 0;  // blocks that carry image data (non-interleaved scans cover these)
    int alloc_w = 0, alloc_h = 0;    // blocks allocated (whole MCUs)
    int dc_pred = 0;
    std::vector<int16_t> coef;
    std::vector<uint8_t> plane;  // alloc_w*8 x alloc_h*8 samples after the inverse transform
};

struct Decoder {
    const uint8_t* data;
    size_t size, pos = 0;
    std::string err;

    uint16_t quant[4][64];
    bool quant_present[4] = {false, false, false, false};
    HuffTable dc_tab[4], ac_tab[4];
    std::vector<Component> comps;
    int width = 0, height = 0, hmax = 1, vmax = 1;
    bool progressive = false, have_frame = false;
    int restart_interval = 0;

    // entropy-coded segment reader
    uint32_t bit_buf = 0;
    int bit_count = 0;
    bool hit_marker = false;
    int eob_run = 0;

    Decoder(const uint8_t* d, size_t n) : data(d), size(n) { memset(quant, 0, sizeof(quant)); }

    bool fail(const char* what) {
        if (err.empty()) err = what;
        return false;
    }

    int read_u8() { return pos < size ? data[pos++] : -1; }
    int read_u16() {
        if (pos + 2 > size) { pos = size; return -1; }
        const int v = (data[pos] << 8) | data[pos + 1];
        pos += 2;
        return v;
    }

    void fill_bits() {
        while (bit_count <= 24) {
            int b = 0;
            if (!hit_marker && pos < size) {
                b = data[pos];
                if (b == 0xFF) {
                    const int next = pos + 1 < size ? data[pos + 1] : 0xD9;
                    if (next == 0x00) {
                        pos += 2;  // a stuffed zero follows a data byte of 0xFF
                    } else {
                        hit_marker = true;  // zeros are fed from here on, the marker stays for the caller
                        b = 0;
                    }
                } else {
                    pos++;
                }
            }
            bit_buf |= (uint32_t)b << (24 - bit_count);
            bit_count += 8;
        }
    }
    int get_bits(int n) {
        if (n == 0) return 0;
        if (bit_count < n) fill_bits();
        const int v = (int)(bit_buf >> (32 - n));
        bit_buf <<= n;
        bit_count -= n;
        return v;
    }
    int get_bit() { return get_bits(1); }
    void reset_bits() { bit_buf = 0; bit_count = 0; hit_marker = false; }

    int decode_symbol(const HuffTable& t) {
        int code = 0;
        for (int len = 1; len <= 16; len++) {
            code = (code << 1) | get_bit();
            if (t.max_code[len] >= 0 && code <= t.max_code[len] && code >= t.min_code[len])
                return t.values[t.first_index[len] + code - t.min_code[len]];
        }
        fail("bad Huffman code");
        return 0;
    }
    // T.81 F.2.2.1: the s-bit pattern v stands for v if its top bit is set, else for v - 2^s + 1
    static int extend(int v, int s) { return s == 0 ? 0 : (v < (1 << (s - 1)) ? v - (1 << s) + 1 : v); }

    // ---- marker segments ----
    bool read_dqt(int len) {
        const size_t end = pos + (size_t)len;
        while (pos < end) {
            const int pq_tq = read_u8();
            const int pq = pq_tq >> 4, tq = pq_tq & 15;
            if (tq > 3 || pq > 1) return fail("bad quantisation table");
            for (int i = 0; i < 64; i++) {
                const int q = pq ? read_u16() : read_u8();
                if (q < 0) return fail("truncated quantisation table");
                quant[tq][ZIGZAG[i]] = (uint16_t)q;
            }
            quant_present[tq] = true;
        }
        return pos == end || fail("bad quantisation segment");
    }
    bool read_dht(int len) {
        const size_t end = pos + (size_t)len;
        while (pos < end) {
            const int tc_th = read_u8();
            const int tc = tc_th >> 4, th = tc_th & 15;
            if (tc > 1 || th > 3) return fail("bad Huffman table id");
            uint8_t counts[16];
            int total = 0;
            for (int i = 0; i < 16; i++) {
                const int c = read_u8();
                if (c < 0) return fail("truncated Huffman table");
                counts[i] = (uint8_t)c;
                total += c;
            }
            if (total > 256 || pos + (size_t)total > size) return fail("bad Huffman table");
            HuffTable& t = tc ? ac_tab[th] : dc_tab[th];
            if (!t.build(counts, data + pos, total)) return fail("inconsistent Huffman table");
            pos += (size_t)total;
        }
        return pos == end || fail("bad Huffman segment");
    }
    bool read_sof(int len, bool prog) {
        if (have_frame) return fail("more than one frame");
        const int precision = read_u8();
        height = read_u16();
        width = read_u16();
        const int n = read_u8();
        if (precision != 8) return fail("only 8-bit JPEG samples are supported");
        if (width <= 0 || height <= 0) return fail("JPEG without a size");
        if (width > 16384 || height > 16384) return fail("JPEG larger than 16384 pixels a side");
        if (n != 1 && n != 3) return fail("only grey and YCbCr JPEG files are supported");
        if (len != 6 + 3 * n) return fail("bad frame header");
        comps.resize((size_t)n);
        for (Component& c : comps) {
            c.id = read_u8();
            const int hv = read_u8();
            c.h = hv >> 4; c.v = hv & 15;
            c.tq = read_u8();
            if (c.h < 1 || c.h > 4 || c.v < 1 || c.v > 4 || c.tq > 3) return fail("bad component");
            hmax = c.h > hmax ? c.h : hmax;
            vmax = c.v > vmax ? c.v : vmax;
        }
        const int mcus_w = (width + 8 * hmax - 1) / (8 * hmax), mcus_h = (height + 8 * vmax - 1) / (8 * vmax);
        for (Component& c : comps) {
            const int cw = (width * c.h + hmax - 1) / hmax, ch = (height * c.v + vmax - 1) / vmax;
            c.blocks_w = (cw + 7) / 8; c.blocks_h = (ch + 7) / 8;
            c.alloc_w = mcus_w * c.h; c.alloc_h = mcus_h * c.v;
            c.coef.assign((size_t)c.alloc_w * c.alloc_h * 64, 0);
        }
        progressive = prog;
        have_frame = true;
        return true;
    }

    // ---- block decoders ----
    bool block_sequential(Component& c, int16_t* b) {
        const HuffTable& dt = dc_tab[c.td];
        const HuffTable& at = ac_tab[c.ta];
        const int s = decode_symbol(dt);
        if (s > 11) return fail("bad DC size");
        c.dc_pred += extend(get_bits(s), s);
        b[0] = (int16_t)c.dc_pred;
        for (int k = 1; k < 64;) {
            const int rs = decode_symbol(at);
            const int r = rs >> 4, sz = rs & 15;
            if (sz == 0) {
                if (r != 15) break;  // end of block
                k += 16;
                continue;
            }
            k += r;
            if (k > 63) return fail("coefficient index out of range");
            b[ZIGZAG[k]] = (int16_t)extend(get_bits(sz), sz);
            k++;
        }
        return err.empty();
    }
    bool block_dc_first(Component& c, int16_t* b, int al) {
        const int s = decode_symbol(dc_tab[c.td]);
        if (s > 11) return fail("bad DC size");
        c.dc_pred += extend(get_bits(s), s);
        b[0] = (int16_t)(c.dc_pred * (1 << al));
        return err.empty();
    }
    void block_dc_refine(int16_t* b, int al) {
        if (get_bit()) b[0] = (int16_t)(b[0] | (1 << al));
    }
    bool block_ac_first(Component& c, int16_t* b, int ss, int se, int al) {
        if (eob_run > 0) { eob_run--; return true; }
        const HuffTable& at = ac_tab[c.ta];
        for (int k = ss; k <= se;) {
            const int rs = decode_symbol(at);
            const int r = rs >> 4, sz = rs & 15;
            if (sz == 0) {
                if (r == 15) { k += 16; continue; }
                eob_run = (1 << r) - 1;
                if (r) eob_run += get_bits(r);
                break;
            }
            k += r;
            if (k > se) return fail("coefficient index out of range");
            b[ZIGZAG[k]] = (int16_t)(extend(get_bits(sz), sz) * (1 << al));
            k++;
        }
        return err.empty();
    }
    // T.81 G.1.2.3: correction bits for the coefficients already non-zero are interleaved with the
    // newly non-zero ones, whose run lengths count only the still-zero positions
    bool block_ac_refine(Component& c, int16_t* b, int ss, int se, int al) {
        const int p1 = 1 << al, m1 = -(1 << al);
        const HuffTable& at = ac_tab[c.ta];
        int k = ss;
        if (eob_run == 0) {
            for (; k <= se; k++) {
                const int rs = decode_symbol(at);
                int r = rs >> 4, sz = rs & 15;
                int value = 0;
                if (sz) {
                    if (sz != 1) return fail("bad refinement symbol");
                    value = get_bit() ? p1 : m1;
                } else if (r != 15) {
                    eob_run = 1 << r;
                    if (r) eob_run += get_bits(r);
                    break;
                }
                for (; k <= se; k++) {
                    int16_t& coef = b[ZIGZAG[k]];
                    if (coef != 0) {
                        if (get_bit() && (coef & p1) == 0) coef = (int16_t)(coef + (coef >= 0 ? p1 : m1));
                    } else if (--r < 0) {
                        break;
                    }
                }
                if (value && k <= se) b[ZIGZAG[k]] = (int16_t)value;
                if (!err.empty()) return false;
            }
        }
        if (eob_run > 0) {
            for (; k <= se; k++) {
                int16_t& coef = b[ZIGZAG[k]];
                if (coef != 0 && get_bit() && (coef & p1) == 0) coef = (int16_t)(coef + (coef >= 0 ? p1 : m1));
            }
            eob_run--;
        }
        return err.empty();
    }

    bool restart() {
        // skip to the marker the entropy reader stopped at (or the next one), which must be RSTn
        reset_bits();
        while (pos + 1 < size && !(data[pos] == 0xFF && data[pos + 1] != 0x00 && data[pos + 1] != 0xFF)) pos++;
        if (pos + 1 >= size) return fail("truncated scan");
        const int m = data[pos + 1];
        if (m < 0xD0 || m > 0xD7) return fail("restart marker expected");
        pos += 2;
        for (Component& c : comps) c.dc_pred = 0;
        eob_run = 0;
        return true;
    }

    bool read_scan(int len) {
        if (!have_frame) return fail("scan before frame header");
        const int n = read_u8();
        if (n < 1 || n > (int)comps.size() || len != 4 + 2 * n) return fail("bad scan header");
        Component* sc[3];
        for (int i = 0; i < n; i++) {
            const int id = read_u8(), tables = read_u8();
            sc[i] = nullptr;
            for (Component& c : comps) if (c.id == id) sc[i] = &c;
            if (!sc[i]) return fail("scan names an unknown component");
            sc[i]->td = tables >> 4; sc[i]->ta = tables & 15;
            if (sc[i]->td > 3 || sc[i]->ta > 3) return fail("bad table selector");
        }
        const int ss = read_u8(), se = read_u8(), ahal = read_u8();
        const int ah = ahal >> 4, al = ahal & 15;
        if (progressive) {
            if (ss > se || se > 63 || (ss == 0 && se != 0) || (ss > 0 && n != 1) || al > 13) return fail("bad progressive scan");
        } else if (ss != 0 || se != 63 || ah != 0 || al != 0) {
            return fail("bad sequential scan");
        }
        for (int i = 0; i < n; i++) {
            const bool need_dc = ss == 0 && !(progressive && ah != 0), need_ac = progressive ? ss > 0 : true;
            if (need_dc && !dc_tab[sc[i]->td].present) return fail("missing DC table");
            if (need_ac && !ac_tab[sc[i]->ta].present) return fail("missing AC table");
        }
        reset_bits();
        eob_run = 0;
        for (Component& c : comps) c.dc_pred = 0;

        auto one_block = [&](Component& c, int bx, int by) -> bool {
            int16_t* b = c.coef.data() + ((size_t)by * c.alloc_w + bx) * 64;
            if (!progressive) return block_sequential(c, b);
            if (ss == 0) {
                if (ah == 0) return block_dc_first(c, b, al);
                block_dc_refine(b, al);
                return true;
            }
            return ah == 0 ? block_ac_first(c, b, ss, se, al) : block_ac_refine(c, b, ss, se, al);
        };

        int until_restart = restart_interval;
        if (n == 1) {  // not interleaved: the component's own blocks, row by row
            Component& c = *sc[0];
            for (int by = 0; by < c.blocks_h; by++) {
                for (int bx = 0; bx < c.blocks_w; bx++) {
                    if (restart_interval && until_restart == 0) {
                        if (!restart()) return false;
                        until_restart = restart_interval;
                    }
                    if (!one_block(c, bx, by)) return false;
                    until_restart--;
                }
            }
        } else {
            const int mcus_w = comps[0].alloc_w / comps[0].h, mcus_h = comps[0].alloc_h / comps[0].v;
            for (int my = 0; my < mcus_h; my++) {
                for (int mx = 0; mx < mcus_w; mx++) {
                    if (restart_interval && until_restart == 0) {
                        if (!restart()) return false;
                        until_restart = restart_interval;
                    }
                    for (int i = 0; i < n; i++) {
                        Component& c = *sc[i];
                        for (int v = 0; v < c.v; v++)
                            for (int h = 0; h < c.h; h++)
                                if (!one_block(c, mx * c.h + h, my * c.v + v)) return false;
                    }
                    until_restart--;
                }
            }
        }
        // leave the position at the marker that ended the entropy-coded data
        reset_bits();
        while (pos + 1 < size && !(data[pos] == 0xFF && data[pos + 1] != 0x00 && data[pos + 1] != 0xFF)) pos++;
        return true;
    }

    // ---- reconstruction ----
    void inverse_transform() {
        float basis[8][8];  // basis[x][u] = C(u) cos((2x+1) u pi / 16) / 2
        for (int x = 0; x < 8; x++)
            for (int u = 0; u < 8; u++)
                basis[x][u] = (float)((u == 0 ? std::sqrt(0.5) : 1.0) * std::cos((2 * x + 1) * u * M_PI / 16.0) * 0.5);
        for (Component& c : comps) {
            const int stride = c.alloc_w * 8;
            c.plane.assign((size_t)stride * c.alloc_h * 8, 0);
            const uint16_t* q = quant[c.tq];
            for (int by = 0; by < c.alloc_h; by++) {
                for (int bx = 0; bx < c.alloc_w; bx++) {
                    const int16_t* b = c.coef.data() + ((size_t)by * c.alloc_w + bx) * 64;
                    float f[64], tmp[64];
                    for (int i = 0; i < 64; i++) f[i] = (float)(b[i] * (int)q[i]);
                    for (int v = 0; v < 8; v++)  // rows: over u
                        for (int x = 0; x < 8; x++) {
                            float s = 0.0f;
                            for (int u = 0; u < 8; u++) s += basis[x][u] * f[v * 8 + u];
                            tmp[v * 8 + x] = s;
                        }
                    for (int x = 0; x < 8; x++)  // columns: over v
                        for (int y = 0; y < 8; y++) {
                            float s = 0.0f;
                            for (int v = 0; v < 8; v++) s += basis[y][v] * tmp[v * 8 + x];
                            const int level = (int)std::lrintf(s + 128.0f);
                            c.plane[(size_t)(by * 8 + y) * stride + bx * 8 + x] = (uint8_t)(level < 0 ? 0 : (level > 255 ? 255 : level));
                        }
                }
            }
        }
    }

    // A component at full resolution.  Planes subsampled by two are enlarged with the triangle filter
    // (output samples sit a quarter and three quarters of the way between input samples: weights 3 : 1
    // along each subsampled axis, rounded once), other ratios by replication.
    void full_plane(const Component& c, std::vector<uint8_t>& out) {
        out.resize((size_t)width * height);
        const int stride = c.alloc_w * 8;
        const int cw = (width * c.h + hmax - 1) / hmax, ch = (height * c.v + vmax - 1) / vmax;
        const bool h2 = c.h * 2 == hmax, v2 = c.v * 2 == vmax, h1 = c.h == hmax, v1 = c.v == vmax;
        if ((h2 || h1) && (v2 || v1) && (h2 || v2)) {
            std::vector<int> t((size_t)cw);
            for (int y = 0; y < height; y++) {
                const int ny = v2 ? (y / 2 < ch ? y / 2 : ch - 1) : y;
                int fy = ny;
                if (v2) { fy = (y & 1) ? ny + 1 : ny - 1; fy = fy < 0 ? 0 : (fy > ch - 1 ? ch - 1 : fy); }
                const uint8_t* near = c.plane.data() + (size_t)ny * stride;
                const uint8_t* far = c.plane.data() + (size_t)fy * stride;
                for (int i = 0; i < cw; i++) t[(size_t)i] = 3 * near[i] + far[i];  // in quarters
                uint8_t* o = out.data() + (size_t)y * width;
                for (int x = 0; x < width; x++) {
                    if (!h2) { o[x] = (uint8_t)((t[(size_t)x] + 2) >> 2); continue; }
                    const int i = x / 2 < cw ? x / 2 : cw - 1;
                    int j = (x & 1) ? i + 1 : i - 1;
                    j = j < 0 ? 0 : (j > cw - 1 ? cw - 1 : j);
                    o[x] = (uint8_t)((3 * t[(size_t)i] + t[(size_t)j] + 8) >> 4);
                }
            }
            return;
        }
        for (int y = 0; y < height; y++)
            for (int x = 0; x < width; x++)
                out[(size_t)y * width + x] = c.plane[(size_t)(y * c.v / vmax) * stride + x * c.h / hmax];
    }

    void to_rgb(std::vector<uint8_t>& rgb) {
        rgb.resize((size_t)width * height * 3);
        std::vector<uint8_t> planes[3];
        for (size_t k = 0; k < comps.size(); k++) full_plane(comps[k], planes[k]);
        for (size_t p = 0; p < (size_t)width * height; p++) {
            uint8_t* o = rgb.data() + p * 3;
            const int Y = planes[0][p];
            if (comps.size() == 1) { o[0] = o[1] = o[2] = (uint8_t)Y; continue; }
            const float cb = (float)planes[1][p] - 128.0f, cr = (float)planes[2][p] - 128.0f;
            const float r = Y + 1.402f * cr, g = Y - 0.344136f * cb - 0.714136f * cr, b = Y + 1.772f * cb;
            auto clamp8 = [](float v) { const int i = (int)std::lrintf(v); return (uint8_t)(i < 0 ? 0 : (i > 255 ? 255 : i)); };
            o[0] = clamp8(r); o[1] = clamp8(g); o[2] = clamp8(b);
        }
    }

    bool run(std::vector<uint8_t>& rgb, int& w, int& h) {
        if (size < 4 || data[0] != 0xFF || data[1] != 0xD8) return fail("not a JPEG file");
        pos = 2;
        bool scans = false;
        while (true) {
            // markers may be preceded by fill bytes
            while (pos < size && data[pos] != 0xFF) pos++;
            while (pos < size && data[pos] == 0xFF) pos++;
            if (pos >= size) break;
            const int marker = data[pos++];
            if (marker == 0xD9) break;                               // EOI
            if (marker == 0x01 || (marker >= 0xD0 && marker <= 0xD7)) continue;  // no payload
            const int len = read_u16();
            if (len < 2 || pos + (size_t)(len - 2) > size) return fail("truncated JPEG segment");
            const size_t end = pos + (size_t)(len - 2);
            bool ok = true;
            switch (marker) {
                case 0xDB: ok = read_dqt(len - 2); break;
                case 0xC4: ok = read_dht(len - 2); break;
                case 0xC0: case 0xC1: ok = read_sof(len - 2, false); break;
                case 0xC2: ok = read_sof(len - 2, true); break;
                case 0xC3: case 0xC5: case 0xC6: case 0xC7: case 0xC9: case 0xCA: case 0xCB: case 0xCD: case 0xCE: case 0xCF:
                    return fail("lossless, hierarchical and arithmetic-coded JPEG files are not supported");
                case 0xDD: restart_interval = read_u16(); break;
                case 0xDA:
                    ok = read_scan(len - 2);
                    scans = true;
                    if (ok) continue;  // read_scan left the position at the next marker
                    break;
                default: break;  // APPn, COM and the rest carry nothing that is needed
            }
            if (!ok) return false;
            pos = end;
        }
        if (!have_frame || !scans) return fail("JPEG without image data");
        for (const Component& c : comps)
            if (!quant_present[c.tq]) return fail("missing quantisation table");
        inverse_transform();
        to_rgb(rgb);
        w = width; h = height;
        return true;
    }
};

bool read_file(const char* path, std::vector<uint8_t>& buf) {
    FILE* fp = fopen(path, "rb");
    if (!fp) return false;
    fseek(fp, 0, SEEK_END);
    const long n = ftell(fp);
    fseek(fp, 0, SEEK_SET);
    buf.resize(n > 0 ? (size_t)n : 0);
    const bool ok = n >= 0 && fread(buf.data(), 1, buf.size(), fp) == buf.size();
    fclose(fp);
    return ok;
}

}  // namespace

bool load_jpeg_rgb8(const char* path, std::vector<uint8_t>& rgb, int& w, int& h, std::string& err) {
    std::vector<uint8_t> buf;
    if (!read_file(path, buf)) { err = std::string("cannot read ") + path; return false; }
    Decoder d(buf.data(), buf.size());
    if (!d.run(rgb, w, h)) { err = d.err.empty() ? "JPEG decoding failed" : d.err; return false; }
    return true;
}

bool load_texture_rgb8(const char* path, std::vector<uint8_t>& rgb, int& w, int& h, std::string& err) {
    FILE* fp = fopen(path, "rb");
    if (!fp) { err = std::string("cannot read ") + path; return false; }
    uint8_t sig[18] = {0};
    const size_t got = fread(sig, 1, sizeof(sig), fp);
    fclose(fp);
    if (got >= 2 && sig[0] == 0xFF && sig[1] == 0xD8) return load_jpeg_rgb8(path, rgb, w, h, err);
    if (got >= 2 && sig[0] == 'B' && sig[1] == 'M') return load_bmp_rgb8(path, rgb, w, h, err);
    if (got >= 8 && sig[0] == 0x89 && sig[1] == 'P') return load_png_rgb8(path, rgb, w, h, err);
    if (got >= 6 && sig[0] == 'G' && sig[1] == 'I' && sig[2] == 'F' && sig[3] == '8') return load_gif_rgb8(path, rgb, w, h, err);
    if (got >= 3 && sig[0] == 'P' && (sig[1] == '5' || sig[1] == '6')) return load_pnm_rgb8(path, rgb, w, h, err);
    // a TGA file has no signature: it is tried last, on a plausible header
    if (looks_like_tga(sig, got)) return load_tga_rgb8(path, rgb, w, h, err);
    return load_png_rgb8(path, rgb, w, h, err);
}

}  // namespace pt
