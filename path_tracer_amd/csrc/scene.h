// scene.h -- host-side scene data of the path tracer: mesh, materials, images, accelerator trees.
#pragma once

#include <cstdint>
#include <cstdio>
#include <string>
#include <vector>

#include "pt_math.h"

namespace pt {

enum MaterialType { MAT_LAMBERTIAN = 0, MAT_METAL = 1, MAT_GLASS = 2 };

// Render-time material state; defaults of src/cs488.h:372-387.
struct HostMaterial {
    std::string name;  // keeps the trailing newline of its "newmtl" line (src/cs488.h:868-871)
    int type = MAT_LAMBERTIAN;
    float eta = 1.0f;
    vec3 Ka = {0.0f, 0.0f, 0.0f};
    vec3 Kd = {0.9f, 0.9f, 0.9f};
    vec3 Ks = {0.0f, 0.0f, 0.0f};
    vec3 Ke = {0.0f, 0.0f, 0.0f};
    float Ns = 0.0f;
    float opacity = 1.0f;
    float roughness = 0.6f;
    bool textured = false;
    std::string tex_path;
    std::vector<uint8_t> tex;  // 3 bytes per texel, first image row first
    int tex_w = 0, tex_h = 0;
};

// Triangle soup: every triangle owns its three vertices (the reference expands indexed faces the
// same way, src/cs488.h:1138-1182).
struct HostMesh {
    std::vector<float> pos;  // 9 per triangle
    std::vector<float> nrm;  // 9 per triangle
    std::vector<float> uv;   // 6 per triangle
    std::vector<int32_t> mat;
    std::vector<HostMaterial> materials;
    int num_tris() const { return (int)mat.size(); }
    vec3 p(int tri, int k) const { return v3(pos[9 * tri + 3 * k], pos[9 * tri + 3 * k + 1], pos[9 * tri + 3 * k + 2]); }
    void clear() { pos.clear(); nrm.clear(); uv.clear(); mat.clear(); materials.clear(); }
};

// Float RGB image, row 0 = bottom row (src/cs488.h:195-198, 208-215).
struct HostImage {
    std::vector<float> rgb;
    int w = 0, h = 0;
};

// Accelerator as the CPU builders leave it, nodes in the reference's own order.
struct HostTree {
    struct Node {
        float box[6];  // min xyz, max xyz
        int32_t left = -1, right = -1;  // -1 on a leaf
        int32_t first = 0, count = 0;   // leaf: range in refs
    };
    std::vector<Node> nodes;
    std::vector<int32_t> refs;
    int kind = 0;
    int max_depth = 0;  // edges on the longest root-to-leaf path
};

// ---- loader.cpp ----
// TriangleMesh::load (src/cs488.h:703-788).  false: file missing or no triangle in it.
// extended = false is the reference's reader; true additionally accepts polygons (fan triangulation), negative
// indices and face lines longer than the reference's 79-byte reads
bool load_obj(const char* path, HostMesh& mesh, std::string& err, bool extended = false);
// TriangleMesh::createSingleTriangle (src/cs488.h:814-835)
void make_single_triangle(HostMesh& mesh);

// ---- image_io.cpp ----
// Radiance RGBE (.hdr), flat or new-style RLE scanlines; output rows flipped to bottom-up.
bool load_hdr(const char* path, HostImage& img, std::string& err);
// PNG of any colour type and bit depth, interlaced (Adam7) or not, converted to 8-bit RGB.
bool load_png_rgb8(const char* path, std::vector<uint8_t>& rgb, int& w, int& h, std::string& err);
// jpeg_decode.cpp: sequential and progressive Huffman-coded JPEG, 8-bit grey or YCbCr, to 8-bit RGB.
bool load_jpeg_rgb8(const char* path, std::vector<uint8_t>& rgb, int& w, int& h, std::string& err);
// tga_bmp.cpp: TGA (true colour, grey, colour-mapped; plain or run-length) and BMP (1/4/8-bit palette, 16/24/32-bit,
// bit masks; not run-length) to 8-bit RGB, alpha dropped.
bool looks_like_tga(const uint8_t* header, size_t n);
bool load_tga_rgb8(const char* path, std::vector<uint8_t>& rgb, int& w, int& h, std::string& err);
bool load_bmp_rgb8(const char* path, std::vector<uint8_t>& rgb, int& w, int& h, std::string& err);
// First image of a GIF on a black canvas; binary PGM / PPM with 8 or 16 bits per sample.
bool load_gif_rgb8(const char* path, std::vector<uint8_t>& rgb, int& w, int& h, std::string& err);
bool load_pnm_rgb8(const char* path, std::vector<uint8_t>& rgb, int& w, int& h, std::string& err);
// A map_Kd file by its signature: JPEG, PNG, BMP or (no signature: by its header) TGA -- the formats among those the
// reference's stbi_load reads (src/cs488.h:839-846) that scene files use.
bool load_texture_rgb8(const char* path, std::vector<uint8_t>& rgb, int& w, int& h, std::string& err);
bool save_png_rgb8(const char* path, const uint8_t* rgb, int w, int h, std::string& err);
bool save_pfm(const char* path, const float* rgb_bottom_up, int w, int h, std::string& err);
// Image::save quantisation (src/cs488.h:219-228): clamp to [0,1], scale by 255, truncate; rows top-down.
void quantise_image(const float* rgb_bottom_up, int w, int h, std::vector<uint8_t>& out);
// Animated GIF of 8-bit RGB frames (rows top-down), as the window's `F` recording produces
// (src/cs488.h:255-266, 2297-2310): fixed 252-colour palette, ordered dither, delay in 1/100 s.
class GifWriter {
public:
    ~GifWriter() { end(); }
    bool begin(const char* path, int w, int h, int delay_cs, std::string& err);
    bool frame(const uint8_t* rgb_top_down, std::string& err);
    void end();
    bool open() const { return fp_ != nullptr; }
    int width() const { return w_; }
    int height() const { return h_; }
private:
    FILE* fp_ = nullptr;
    int w_ = 0, h_ = 0, delay_ = 1, frames_ = 0;
};

// ---- accel_build.cpp ----
// threads: 0 = as many as the host offers, 1 = single-threaded; the tree does not depend on it
void build_bvh(const HostMesh& mesh, HostTree& tree, int threads = 0);  // src/cs488.h:1490-1667
void build_kd(const HostMesh& mesh, HostTree& tree, int threads = 0);   // src/cs488.h:1229-1320, 1388-1400

}  // namespace pt
