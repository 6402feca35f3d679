// loader.cpp -- Wavefront .obj / .mtl reader with the reference's semantics.
//
// Behaviour kept from TriangleMesh::load / ParseOBJ / LoadMTL (src/cs488.h:703-788, 917-1194,
// 854-915):
//   * text is consumed in pieces of at most 79 bytes (the reference reads through an 80-byte
//     fgets buffer), so an over-long line continues as a new "line";
//   * only v / vt / vn / f / usemtl / mtllib are understood; a face contributes its first three
//     vertex references and nothing else (no polygon triangulation);
//   * a reference is "v", "v/t", "v//n" or "v/t/n", each field read as a decimal integer that
//     stops at the first non-digit; a missing field is 0;
//   * one face without a normal (texcoord) index turns normals (texcoords) off for the whole
//     mesh; without normals every triangle gets its geometric normal;
//   * material names keep the newline of their line and are compared including it;
//   * .mtl keys: newmtl Ka Kd Ks Ns map_Kd Ke Op Ro Ri;
//   * typing: Ns == 100 -> metal; name starting with "glass" -> glass with eta forced to 1.5;
//   * texcoords survive only on triangles whose material has a texture;
//   * a mesh whose usemtl lines match nothing is given exactly one material.
// The reference indexes its arrays unchecked; here a reference outside the vertex, normal or
// texcoord tables fails the load instead.
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "scene.h"

namespace pt {

namespace {

// Parses up to `want` floats separated by blanks; values that fail to parse stay 0.
int scan_floats(const char* s, float* out, int want) {
    int got = 0;
    for (int i = 0; i < want; i++) out[i] = 0.0f;
    while (got < want) {
        char* end = nullptr;
        float v = strtof(s, &end);
        if (end == s) break;
        out[got++] = v;
        s = end;
    }
    return got;
}

bool starts_with(const char* line, const char* key) { return strncmp(line, key, strlen(key)) == 0; }

// Text after a keyword and one separator byte, as std::string::erase(0, len+1) leaves it.
std::string tail_of(const char* line, size_t key_len) {
    size_t n = strlen(line);
    return n > key_len + 1 ? std::string(line + key_len + 1) : std::string();
}

struct FaceRef {
    int v = 0, t = 0, n = 0;
};

FaceRef parse_ref(const char* word) {
    FaceRef r;
    const char* first = strchr(word, '/');
    r.v = atoi(word);
    if (first) {
        // the second field starts after the first slash, the third after the LAST further slash
        // (every later slash re-points the normal field, src/cs488.h:932-942)
        r.t = atoi(first + 1);
        const char* last = strrchr(word, '/');
        if (last != first) r.n = atoi(last + 1);
    }
    return r;
}

void read_mtl(const std::string& path, const std::string& dir, std::vector<HostMaterial>& out) {
    FILE* fp = fopen(path.c_str(), "r");
    if (!fp) return;
    HostMaterial cur;
    bool open = false;
    char line[81];
    float f[3];
    while (fgets(line, 80, fp)) {
        if (starts_with(line, "newmtl")) {
            if (open && !cur.name.empty()) out.push_back(cur);
            cur = HostMaterial();
            cur.name = tail_of(line, 6);
            open = true;
        } else if (starts_with(line, "Ka")) {
            scan_floats(line + 2, f, 3); cur.Ka = v3(f[0], f[1], f[2]);
        } else if (starts_with(line, "Kd")) {
            scan_floats(line + 2, f, 3); cur.Kd = v3(f[0], f[1], f[2]);
        } else if (starts_with(line, "Ks")) {
            scan_floats(line + 2, f, 3); cur.Ks = v3(f[0], f[1], f[2]);
        } else if (starts_with(line, "Ns")) {
            scan_floats(line + 2, f, 1); cur.Ns = f[0];
        } else if (starts_with(line, "map_Kd")) {
            std::string name = tail_of(line, 6);
            if (!name.empty()) name.erase(name.size() - 1);
            cur.textured = true;
            cur.tex_path = dir + name;
            std::string err;
            if (!load_texture_rgb8(cur.tex_path.c_str(), cur.tex, cur.tex_w, cur.tex_h, err)) {
                fprintf(stderr, "Unable to load texture: %s\n", cur.tex_path.c_str());  // src/cs488.h:843
                cur.tex.clear();
                cur.tex_w = cur.tex_h = 0;
            }
        } else if (starts_with(line, "Ke")) {
            scan_floats(line + 2, f, 3); cur.Ke = v3(f[0], f[1], f[2]);
        } else if (starts_with(line, "Op")) {
            scan_floats(line + 2, f, 1); cur.opacity = f[0];
        } else if (starts_with(line, "Ro")) {
            scan_floats(line + 2, f, 1); cur.roughness = f[0];
        } else if (starts_with(line, "Ri")) {
            scan_floats(line + 2, f, 1); cur.eta = f[0];
        }
    }
    if (open && !cur.name.empty()) out.push_back(cur);
    fclose(fp);
}

}  // namespace

bool load_obj(const char* path, HostMesh& mesh, std::string& err, bool extended) {
    mesh.clear();
    FILE* fp = fopen(path, "r");
    if (!fp) {
        err = std::string("Cannot open \"") + path + "\" for reading";
        return false;
    }
    std::string dir = path;
    size_t cut = dir.find_last_of("/\\");
    dir = (cut == std::string::npos ? std::string() : dir.substr(0, cut)) + "/";

    char line[81];
    // pass 1: material libraries, wherever their lines are (src/cs488.h:965-973)
    while (fgets(line, 80, fp)) {
        if (starts_with(line, "mtllib")) {
            std::string name = tail_of(line, 6);
            if (!name.empty()) name.erase(name.size() - 1);
            read_mtl(dir + name, dir, mesh.materials);
        }
    }
    rewind(fp);

    // pass 2: geometry
    std::vector<float> vs, vns, vts;
    std::vector<FaceRef> refs;  // 3 per face
    std::vector<int> face_mat;
    bool have_normals = true, have_uvs = true, any_material = false;
    int current = 0;
    float f[3];
    std::string long_line;  // extended mode: a face line of any length
    while (fgets(line, 80, fp)) {
        if (extended && line[0] == 'f') {
            // Beyond the reference: polygons become triangle fans (v0, vi, vi+1) and negative
            // indices count back from the end of the tables read so far.
            long_line = line;
            while (!long_line.empty() && long_line.back() != '\n' && fgets(line, 80, fp)) long_line += line;
            std::vector<FaceRef> corner;
            const int nv_now = (int)vs.size() / 3, nn_now = (int)vns.size() / 3, nt_now = (int)vts.size() / 2;
            size_t pos = 1;
            while (pos < long_line.size()) {
                while (pos < long_line.size() && isspace((unsigned char)long_line[pos])) pos++;
                size_t end = pos;
                while (end < long_line.size() && !isspace((unsigned char)long_line[end])) end++;
                if (end > pos) {
                    FaceRef r = parse_ref(long_line.substr(pos, end - pos).c_str());
                    if (r.v < 0) r.v += nv_now + 1;
                    if (r.t < 0) r.t += nt_now + 1;
                    if (r.n < 0) r.n += nn_now + 1;
                    corner.push_back(r);
                }
                pos = end;
            }
            for (size_t k = 1; k + 1 < corner.size(); k++) {
                face_mat.push_back(current);
                const FaceRef tri[3] = {corner[0], corner[k], corner[k + 1]};
                for (const FaceRef& r : tri) {
                    if (r.n == 0) have_normals = false;
                    if (r.t == 0) have_uvs = false;
                    refs.push_back(r);
                }
            }
            continue;
        }
        if (line[0] == 'v') {
            if (line[1] == 'n') {
                scan_floats(line + 2, f, 3);
                float len = sqrtf(f[0] * f[0] + f[1] * f[1] + f[2] * f[2]);  // src/cs488.h:1016-1019
                vns.push_back(f[0] / len); vns.push_back(f[1] / len); vns.push_back(f[2] / len);
            } else if (line[1] == 't') {
                scan_floats(line + 2, f, 2);
                vts.push_back(f[0]); vts.push_back(f[1]);
            } else {
                scan_floats(line + 1, f, 3);
                vs.push_back(f[0]); vs.push_back(f[1]); vs.push_back(f[2]);
            }
        }
        if (starts_with(line, "usemtl")) {
            std::string name = tail_of(line, 6);
            for (size_t i = 0; i < mesh.materials.size(); i++) {
                if (name == mesh.materials[i].name) { current = (int)i; any_material = true; break; }
            }
        } else if (line[0] == 'f') {
            char w[3][80];
            w[0][0] = w[1][0] = w[2][0] = '\0';
            sscanf(line + 1, "%79s %79s %79s", w[0], w[1], w[2]);
            face_mat.push_back(current);
            for (int k = 0; k < 3; k++) {
                FaceRef r = parse_ref(w[k]);
                if (r.n == 0) have_normals = false;
                if (r.t == 0) have_uvs = false;
                refs.push_back(r);
            }
        }
    }
    fclose(fp);

    const int ntri = (int)face_mat.size();
    if (ntri == 0) {
        err = "no faces";
        mesh.clear();
        return false;
    }
    const int nv = (int)vs.size() / 3, nn = (int)vns.size() / 3, nt = (int)vts.size() / 2;
    for (const FaceRef& r : refs) {
        if (r.v < 1 || r.v > nv || (have_normals && (r.n < 1 || r.n > nn))) {
            err = "face index outside the vertex or normal table";
            mesh.clear();
            return false;
        }
    }

    if (any_material) {
        for (HostMaterial& m : mesh.materials) {  // src/cs488.h:718-729
            m.type = MAT_LAMBERTIAN;
            if (m.Ns == 100.0f) m.type = MAT_METAL;
            if (m.name.compare(0, 5, "glass", 0, 5) == 0) { m.type = MAT_GLASS; m.eta = 1.5f; }
        }
    } else {
        mesh.materials.resize(1);  // src/cs488.h:732
    }

    mesh.pos.resize(9 * (size_t)ntri);
    mesh.nrm.resize(9 * (size_t)ntri);
    mesh.uv.assign(6 * (size_t)ntri, 0.0f);
    mesh.mat.assign(ntri, 0);
    for (int i = 0; i < ntri; i++) {
        for (int k = 0; k < 3; k++) {
            const FaceRef& r = refs[3 * i + k];
            for (int c = 0; c < 3; c++) mesh.pos[9 * i + 3 * k + c] = vs[3 * (r.v - 1) + c];
            if (have_normals)
                for (int c = 0; c < 3; c++) mesh.nrm[9 * i + 3 * k + c] = vns[3 * (r.n - 1) + c];
        }
        if (!have_normals) {  // src/cs488.h:748-757
            vec3 n = normalize(cross(mesh.p(i, 1) - mesh.p(i, 0), mesh.p(i, 2) - mesh.p(i, 0)));
            for (int k = 0; k < 3; k++) {
                mesh.nrm[9 * i + 3 * k] = n.x; mesh.nrm[9 * i + 3 * k + 1] = n.y; mesh.nrm[9 * i + 3 * k + 2] = n.z;
            }
        }
        if (any_material) {
            mesh.mat[i] = face_mat[i];
            if (have_uvs && mesh.materials[face_mat[i]].textured) {  // src/cs488.h:763-766
                for (int k = 0; k < 3; k++) {
                    const int t = refs[3 * i + k].t;
                    if (t < 1 || t > nt) {
                        err = "face index outside the texcoord table";
                        mesh.clear();
                        return false;
                    }
                    mesh.uv[6 * i + 2 * k] = vts[2 * (t - 1)];
                    mesh.uv[6 * i + 2 * k + 1] = vts[2 * (t - 1) + 1];
                }
            }
        }
    }
    return true;
}

void make_single_triangle(HostMesh& mesh) {
    mesh.clear();
    mesh.materials.resize(1);
    const float p[9] = {-0.5f, -0.5f, 0.0f, 0.5f, -0.5f, 0.0f, 0.0f, 0.5f, 0.0f};
    mesh.pos.assign(p, p + 9);
    mesh.mat.assign(1, 0);
    vec3 n = normalize(cross(mesh.p(0, 1) - mesh.p(0, 0), mesh.p(0, 2) - mesh.p(0, 0)));
    for (int k = 0; k < 3; k++) { mesh.nrm.push_back(n.x); mesh.nrm.push_back(n.y); mesh.nrm.push_back(n.z); }
    const float uv[6] = {0.0f, 0.0f, 0.0f, 1.0f, 1.0f, 0.0f};
    mesh.uv.assign(uv, uv + 6);
}

}  // namespace pt
