// orbit_gltf.cpp — see orbit_gltf.hpp.
#include "orbit_gltf.hpp"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <map>
#include <memory>
#include <unordered_map>
#include <unordered_set>

namespace orbit {

// ============================================================================ assets: MeshData, LOD chain
namespace assets {

size_t MeshData::add_submesh(const float *pos, size_t vertex_count, const uint32_t *idx, size_t index_count,
                             uint16_t material) { // mesh.rs:173-190
    const size_t submesh_index = submeshes.size();
    SubmeshData s;
    s.vertex_offset = positions.size() / 3;
    s.index_offset = indices.size();
    s.vertex_count = vertex_count;
    s.index_count = index_count;
    s.material = material;
    positions.insert(positions.end(), pos, pos + 3 * vertex_count);
    indices.insert(indices.end(), idx, idx + index_count);
    submeshes.push_back(s);
    return submesh_index;
}

void MeshData::compute_bounds() { // mesh.rs:192-215 (same arithmetic as gltf_loader.rs:498-506)
    const size_t n = positions.size() / 3;
    if (n == 0) return;
    compute_mesh_bounds(positions.data(), n, aabb_min, aabb_max, bounding_sphere);
}

namespace {

struct Grid {
    float origin[3];
    float inv_cell;
    uint32_t size;
    uint32_t cell_of(const float *p) const {
        uint32_t k[3];
        for (int a = 0; a < 3; a++) {
            const float f = (p[a] - origin[a]) * inv_cell;
            int32_t i = f > 0.0f ? (int32_t)f : 0;
            if (i >= (int32_t)size) i = (int32_t)size - 1;
            k[a] = (uint32_t)i;
        }
        return k[0] + size * (k[1] + size * k[2]); // size <= 1024: 30 bits
    }
};

// triangles that survive clustering with `g` (corners in three different cells), as sorted cell triples, deduplicated
size_t clustered_triangles(const Grid &g, const float *positions, const uint32_t *indices, size_t index_count,
                           std::vector<uint32_t> *cells_out) {
    struct Key {
        uint32_t a, b, c;
        bool operator==(const Key &o) const { return a == o.a && b == o.b && c == o.c; }
    };
    struct Hash {
        size_t operator()(const Key &k) const {
            uint64_t h = k.a * 0x9E3779B97F4A7C15ull;
            h ^= (h >> 29) + k.b * 0xBF58476D1CE4E5B9ull;
            h ^= (h >> 31) + k.c * 0x94D049BB133111EBull;
            return (size_t)(h ^ (h >> 32));
        }
    };
    std::unordered_set<Key, Hash> seen;
    seen.reserve(index_count / 3);
    size_t kept = 0;
    for (size_t t = 0; t + 2 < index_count; t += 3) {
        uint32_t c[3];
        for (int k = 0; k < 3; k++) c[k] = g.cell_of(positions + 3 * (size_t)indices[t + k]);
        if (c[0] == c[1] || c[1] == c[2] || c[0] == c[2]) continue; // collapsed
        Key key{c[0], c[1], c[2]};
        // rotate so that the smallest cell leads: the same triangle with the same winding has one key
        while (key.a > key.b || key.a > key.c) key = Key{key.b, key.c, key.a};
        if (!seen.insert(key).second) continue;
        kept++;
        if (cells_out) cells_out->insert(cells_out->end(), {c[0], c[1], c[2]});
    }
    return kept;
}

} // namespace

float simplify_clustered(const float *positions, size_t vertex_count, const uint32_t *indices, size_t index_count,
                         size_t target_index_count, std::vector<uint32_t> &out) {
    if (index_count < 3) return 0.0f;
    if (target_index_count >= index_count) { // nothing to do: like meshopt_simplify, the input comes back
        out.insert(out.end(), indices, indices + index_count / 3 * 3);
        return 0.0f;
    }
    // extent of the vertices the triangles use
    float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (size_t i = 0; i < index_count; i++)
        for (int a = 0; a < 3; a++) {
            const float v = positions[3 * (size_t)indices[i] + a];
            mn[a] = v < mn[a] ? v : mn[a];
            mx[a] = v > mx[a] ? v : mx[a];
        }
    float extent = 0.0f;
    for (int a = 0; a < 3; a++) extent = std::max(extent, mx[a] - mn[a]);
    if (!(extent > 0.0f)) return 0.0f; // a point: every triangle collapses
    auto grid = [&](uint32_t size) {
        Grid g;
        for (int a = 0; a < 3; a++) g.origin[a] = mn[a];
        g.size = size;
        g.inv_cell = (float)size / (extent * 1.0001f);
        return g;
    };
    const size_t target_triangles = target_index_count / 3;
    // the finest grid (<= 1024 cells per axis) whose result fits the target: the count grows with the resolution
    // (not strictly, so the bisection is followed by a downward walk until the result fits)
    uint32_t lo = 1, hi = 1024;
    while (lo < hi) {
        const uint32_t mid = (lo + hi + 1) / 2;
        if (clustered_triangles(grid(mid), positions, indices, index_count, nullptr) <= target_triangles) lo = mid;
        else hi = mid - 1;
    }
    uint32_t size = lo;
    std::vector<uint32_t> cells;
    for (;; size--) {
        cells.clear();
        if (clustered_triangles(grid(size), positions, indices, index_count, &cells) <= target_triangles || size == 1)
            break;
    }
    const Grid g = grid(size);
    // representative of a cell: the used vertex closest to the mean of the cell's used vertices
    struct Acc {
        double sum[3] = {0, 0, 0};
        uint32_t n = 0, best = 0;
        double best_d = INFINITY;
    };
    std::unordered_map<uint32_t, Acc> acc;
    std::vector<uint8_t> used(vertex_count, 0);
    for (size_t i = 0; i < index_count; i++) used[indices[i]] = 1;
    for (size_t v = 0; v < vertex_count; v++)
        if (used[v]) {
            Acc &a = acc[g.cell_of(positions + 3 * v)];
            for (int k = 0; k < 3; k++) a.sum[k] += positions[3 * v + k];
            a.n++;
        }
    for (size_t v = 0; v < vertex_count; v++)
        if (used[v]) {
            Acc &a = acc[g.cell_of(positions + 3 * v)];
            double d = 0;
            for (int k = 0; k < 3; k++) {
                const double e = positions[3 * v + k] - a.sum[k] / a.n;
                d += e * e;
            }
            if (d < a.best_d) a.best_d = d, a.best = (uint32_t)v;
        }
    for (uint32_t c : cells) out.push_back(acc[c].best);
    return 1.0f / (float)size;
}

uint32_t GpuAssets::add_material(const OrbitMaterialData &m) {
    materials.push_back(m);
    return (uint32_t)(materials.size() - 1);
}

uint32_t GpuAssets::add_mesh(const MeshData &mesh) { // assets/mod.rs:325-476
    std::vector<uint32_t> mesh_meshlet_data;
    std::vector<OrbitMeshlet> mesh_meshlets;
    uint32_t mesh_lod_count = 0;
    OrbitMeshLod mesh_lods[MAX_MESH_LODS] = {};
    std::vector<uint32_t> lod_indices;
    double index_count_scale = 1.0;

    for (size_t lod_index = 0; lod_index < MAX_MESH_LODS; lod_index++) { // :336
        const size_t meshlet_offset = mesh_meshlets.size();
        bool finish_lod = false;
        for (const SubmeshData &submesh : mesh.submeshes) {
            const float *submesh_vertices = mesh.positions.data() + 3 * submesh.vertex_offset;
            const uint32_t *submesh_indices = mesh.indices.data() + submesh.index_offset;
            lod_indices.clear();
            if (lod_index == 0) {
                lod_indices.assign(submesh_indices, submesh_indices + submesh.index_count);
            } else {
                const size_t target_index_count = (size_t)((double)submesh.index_count * index_count_scale); // :349
                simplify_clustered(submesh_vertices, submesh.vertex_count, submesh_indices, submesh.index_count,
                                   target_index_count, lod_indices);
                if ((target_index_count + 2) / 3 * 3 < lod_indices.size()) finish_lod = true; // :358-360
            }
            if (!lod_indices.empty())
                compute_meshlets(submesh_vertices, submesh.vertex_count, lod_indices.data(), lod_indices.size(),
                                 submesh.material, (uint32_t)submesh.vertex_offset, mesh_meshlet_data, mesh_meshlets);
            index_count_scale *= 0.8; // :391: per submesh, inside the LOD loop (the reference's placement)
        }
        mesh_lod_count += 1;
        mesh_lods[lod_index].meshlet_offset = (uint32_t)meshlet_offset;
        mesh_lods[lod_index].meshlet_count = (uint32_t)(mesh_meshlets.size() - meshlet_offset);
        if (finish_lod) break;
    }

    // the four arenas of shared_stuff (:404-411) as appends: ranges start at the current ends
    const uint32_t vertex_start = (uint32_t)(vertex_positions.size() / 3);
    const uint32_t data_start = (uint32_t)meshlet_data.size();
    const uint32_t meshlet_start = (uint32_t)meshlets.size();
    for (OrbitMeshlet &m : mesh_meshlets) { // :415-419
        m.vertex_offset += vertex_start;
        m.data_offset += data_start;
    }
    vertex_positions.insert(vertex_positions.end(), mesh.positions.begin(), mesh.positions.end());
    meshlet_data.insert(meshlet_data.end(), mesh_meshlet_data.begin(), mesh_meshlet_data.end());
    meshlets.insert(meshlets.end(), mesh_meshlets.begin(), mesh_meshlets.end());

    OrbitMeshInfo info; // MeshInfo::to_gpu, assets/mod.rs:60-96: LOD offsets become absolute meshlet indices
    memset(&info, 0, sizeof(info));
    memcpy(info.bounding_sphere, mesh.bounding_sphere, 16);
    memcpy(info.aabb_min, mesh.aabb_min, 12);
    memcpy(info.aabb_max, mesh.aabb_max, 12);
    info.vertex_offset = vertex_start;
    info.meshlet_data_offset = data_start;
    info.lod_count = mesh_lod_count;
    for (size_t i = 0; i < MAX_MESH_LODS; i++) {
        info.mesh_lods[i].meshlet_offset = i < mesh_lod_count ? meshlet_start + mesh_lods[i].meshlet_offset : 0u;
        info.mesh_lods[i].meshlet_count = i < mesh_lod_count ? mesh_lods[i].meshlet_count : 0u;
    }
    mesh_infos.push_back(info);
    return (uint32_t)(mesh_infos.size() - 1);
}

} // namespace assets

// ============================================================================ JSON (what a glTF document uses)
namespace gltf_loader {
namespace {

struct Json {
    enum Kind { Null, Bool, Number, String, Array, Object } kind = Null;
    bool b = false;
    double num = 0;
    std::string str;
    std::vector<Json> arr;
    std::vector<std::pair<std::string, Json>> obj;
    const Json *get(const char *key) const {
        if (kind != Object) return nullptr;
        for (const auto &kv : obj)
            if (kv.first == key) return &kv.second;
        return nullptr;
    }
    double number_or(const char *key, double dflt) const {
        const Json *j = get(key);
        return j && j->kind == Number ? j->num : dflt;
    }
    int64_t index_or(const char *key, int64_t dflt) const { return (int64_t)number_or(key, (double)dflt); }
};

struct JsonParser {
    const char *p, *end;
    std::string err;
    void ws() {
        while (p < end && (*p == ' ' || *p == '\t' || *p == '\n' || *p == '\r')) p++;
    }
    bool fail(const char *m) {
        if (err.empty()) err = m;
        return false;
    }
    bool parse_string(std::string &out) {
        if (p >= end || *p != '"') return fail("expected string");
        p++;
        while (p < end && *p != '"') {
            if (*p == '\\') {
                if (++p >= end) return fail("bad escape");
                switch (*p) {
                case 'n': out += '\n'; break;
                case 't': out += '\t'; break;
                case 'r': out += '\r'; break;
                case 'b': out += '\b'; break;
                case 'f': out += '\f'; break;
                case 'u': { // \uXXXX -> UTF-8 (surrogate pairs are kept as two 3-byte sequences; names only)
                    if (end - p < 5) return fail("bad \\u escape");
                    unsigned cp = 0;
                    for (int i = 1; i <= 4; i++) {
                        const char c = p[i];
                        cp = cp * 16 + (c >= '0' && c <= '9' ? c - '0' : (c | 32) >= 'a' && (c | 32) <= 'f' ? (c | 32) - 'a' + 10 : 0);
                    }
                    p += 4;
                    if (cp < 0x80) out += (char)cp;
                    else if (cp < 0x800) out += (char)(0xC0 | (cp >> 6)), out += (char)(0x80 | (cp & 63));
                    else out += (char)(0xE0 | (cp >> 12)), out += (char)(0x80 | ((cp >> 6) & 63)), out += (char)(0x80 | (cp & 63));
                    break;
                }
                default: out += *p; // \" \\ \/
                }
                p++;
            } else {
                out += *p++;
            }
        }
        if (p >= end) return fail("unterminated string");
        p++;
        return true;
    }
    bool parse(Json &out, int depth = 0) {
        if (depth > 64) return fail("nesting too deep");
        ws();
        if (p >= end) return fail("unexpected end");
        if (*p == '{') {
            out.kind = Json::Object;
            p++;
            ws();
            if (p < end && *p == '}') return p++, true;
            for (;;) {
                ws();
                std::string key;
                if (!parse_string(key)) return false;
                ws();
                if (p >= end || *p != ':') return fail("expected ':'");
                p++;
                out.obj.emplace_back(std::move(key), Json());
                if (!parse(out.obj.back().second, depth + 1)) return false;
                ws();
                if (p < end && *p == ',') { p++; continue; }
                if (p < end && *p == '}') return p++, true;
                return fail("expected ',' or '}'");
            }
        }
        if (*p == '[') {
            out.kind = Json::Array;
            p++;
            ws();
            if (p < end && *p == ']') return p++, true;
            for (;;) {
                out.arr.emplace_back();
                if (!parse(out.arr.back(), depth + 1)) return false;
                ws();
                if (p < end && *p == ',') { p++; continue; }
                if (p < end && *p == ']') return p++, true;
                return fail("expected ',' or ']'");
            }
        }
        if (*p == '"') {
            out.kind = Json::String;
            return parse_string(out.str);
        }
        if (end - p >= 4 && !strncmp(p, "true", 4)) return out.kind = Json::Bool, out.b = true, p += 4, true;
        if (end - p >= 5 && !strncmp(p, "false", 5)) return out.kind = Json::Bool, out.b = false, p += 5, true;
        if (end - p >= 4 && !strncmp(p, "null", 4)) return out.kind = Json::Null, p += 4, true;
        char *num_end = nullptr;
        const std::string tmp(p, std::min<size_t>((size_t)(end - p), 64));
        const double v = strtod(tmp.c_str(), &num_end);
        if (num_end == tmp.c_str()) return fail("unexpected character");
        out.kind = Json::Number;
        out.num = v;
        p += num_end - tmp.c_str();
        return true;
    }
};

bool read_file(const std::string &path, std::vector<uint8_t> &out) {
    FILE *f = fopen(path.c_str(), "rb");
    if (!f) return false;
    fseek(f, 0, SEEK_END);
    const long n = ftell(f);
    fseek(f, 0, SEEK_SET);
    out.resize(n > 0 ? (size_t)n : 0);
    const size_t got = out.empty() ? 0 : fread(out.data(), 1, out.size(), f);
    fclose(f);
    return got == out.size();
}

std::string dir_of(const std::string &path) {
    const size_t s = path.find_last_of('/');
    return s == std::string::npos ? std::string() : path.substr(0, s + 1);
}

// ---------------------------------------------------------------------------- accessors
struct Document {
    Json root;
    std::vector<std::vector<uint8_t>> buffers;
    std::string err;

    const Json *at(const char *array, int64_t i) const {
        const Json *a = root.get(array);
        return a && a->kind == Json::Array && i >= 0 && (size_t)i < a->arr.size() ? &a->arr[(size_t)i] : nullptr;
    }
    size_t count(const char *array) const {
        const Json *a = root.get(array);
        return a && a->kind == Json::Array ? a->arr.size() : 0;
    }

    // element `i`, component `c` of an accessor as double; component types 5120..5126; no sparse accessors
    struct View {
        const uint8_t *base = nullptr;
        size_t stride = 0, count = 0;
        int component_type = 0, components = 0;
        bool normalized = false;
        double get(size_t i, int c) const {
            const uint8_t *p = base + i * stride;
            switch (component_type) {
            case 5120: { const double v = ((const int8_t *)p)[c]; return normalized ? std::max(v / 127.0, -1.0) : v; }
            case 5121: { const double v = p[c]; return normalized ? v / 255.0 : v; }
            case 5122: { int16_t v; memcpy(&v, p + 2 * c, 2); return normalized ? std::max(v / 32767.0, -1.0) : v; }
            case 5123: { uint16_t v; memcpy(&v, p + 2 * c, 2); return normalized ? v / 65535.0 : v; }
            case 5125: { uint32_t v; memcpy(&v, p + 4 * c, 4); return v; }
            case 5126: { float v; memcpy(&v, p + 4 * c, 4); return v; }
            }
            return 0;
        }
    };

    bool view(int64_t accessor, View &v) {
        const Json *a = at("accessors", accessor);
        if (!a) return err = "accessor index out of range", false;
        if (a->get("sparse")) return err = "sparse accessors are not supported", false;
        const Json *type = a->get("type");
        static const std::pair<const char *, int> kTypes[] = {{"SCALAR", 1}, {"VEC2", 2}, {"VEC3", 3}, {"VEC4", 4},
                                                              {"MAT4", 16}};
        v.components = 0;
        for (const auto &t : kTypes)
            if (type && type->kind == Json::String && type->str == t.first) v.components = t.second;
        v.component_type = (int)a->index_or("componentType", 0);
        static const std::map<int, size_t> kSize = {{5120, 1}, {5121, 1}, {5122, 2}, {5123, 2}, {5125, 4}, {5126, 4}};
        if (!v.components || !kSize.count(v.component_type)) return err = "unsupported accessor type", false;
        // Every number below comes from the file: signed, unchecked, possibly huge.  They are range-checked as signed
        // 64-bit values BEFORE any cast, and the extent test is written so that nothing can wrap.
        const int64_t count = a->index_or("count", 0), a_off = a->index_or("byteOffset", 0);
        const Json *n = a->get("normalized");
        v.normalized = n && n->kind == Json::Bool && n->b;
        const Json *bv = at("bufferViews", a->index_or("bufferView", -1));
        if (!bv) return err = "accessor without bufferView", false;
        const int64_t buffer = bv->index_or("buffer", -1);
        if (buffer < 0 || (size_t)buffer >= buffers.size()) return err = "bufferView.buffer out of range", false;
        const auto &buf = buffers[(size_t)buffer];
        const int64_t size = (int64_t)buf.size();
        const int64_t v_off = bv->index_or("byteOffset", 0), v_len = bv->index_or("byteLength", -1);
        const int64_t stride_in = bv->index_or("byteStride", 0);
        const int64_t elem = (int64_t)(kSize.at(v.component_type) * (size_t)v.components);
        if (count < 0 || a_off < 0 || v_off < 0) return err = "negative count / byteOffset", false;
        // glTF 2.0 §5.11: byteStride in [4, 252], a multiple of 4; and it must hold an element
        if (stride_in != 0 && (stride_in < 4 || stride_in > 252 || stride_in % 4 != 0 || stride_in < elem))
            return err = "bufferView.byteStride out of range", false;
        const int64_t stride = stride_in ? stride_in : elem;
        // the view inside the buffer (byteLength is required by the format; honoured when present)
        if (v_off > size) return err = "bufferView exceeds its buffer", false;
        int64_t avail = size - v_off;
        if (v_len >= 0) {
            if (v_len > avail) return err = "bufferView exceeds its buffer", false;
            avail = v_len;
        }
        // the accessor inside the view: a_off + (count - 1) * stride + elem <= avail, without forming the products
        if (a_off > avail) return err = "accessor exceeds its bufferView", false;
        avail -= a_off;
        if (count > 0 && (elem > avail || (count - 1) > (avail - elem) / stride)) return err = "accessor exceeds its buffer", false;
        v.count = (size_t)count;
        v.stride = (size_t)stride;
        v.base = buf.data() + (size_t)(v_off + a_off);
        return true;
    }
};

bool open_document(const std::string &path, Document &doc, std::string &error) {
    std::vector<uint8_t> file;
    if (!read_file(path, file)) return error = "cannot read " + path, false;
    std::vector<uint8_t> bin_chunk;
    bool have_bin = false;
    const char *json_begin = (const char *)file.data(), *json_end = json_begin + file.size();
    if (file.size() >= 12 && !memcmp(file.data(), "glTF", 4)) { // binary container: header + chunks
        uint32_t version, length;
        memcpy(&version, file.data() + 4, 4);
        memcpy(&length, file.data() + 8, 4);
        if (version != 2 || length > file.size()) return error = "unsupported .glb header", false;
        size_t off = 12;
        json_begin = json_end = nullptr;
        while (off + 8 <= length) {
            uint32_t clen, ctype;
            memcpy(&clen, file.data() + off, 4);
            memcpy(&ctype, file.data() + off + 4, 4);
            if (off + 8 + clen > length) return error = "truncated .glb chunk", false;
            if (ctype == 0x4E4F534Au) json_begin = (const char *)file.data() + off + 8, json_end = json_begin + clen;
            else if (ctype == 0x004E4942u && !have_bin)
                bin_chunk.assign(file.begin() + off + 8, file.begin() + off + 8 + clen), have_bin = true;
            off += 8 + ((clen + 3u) & ~3u);
        }
        if (!json_begin) return error = ".glb without a JSON chunk", false;
    }
    JsonParser parser{json_begin, json_end, {}};
    if (!parser.parse(doc.root) || doc.root.kind != Json::Object) return error = "glTF JSON: " + parser.err, false;
    const std::string base = dir_of(path);
    for (size_t i = 0; i < doc.count("buffers"); i++) { // gltf_loader.rs:520-538
        const Json *b = doc.at("buffers", (int64_t)i);
        const Json *uri = b->get("uri");
        std::vector<uint8_t> data;
        if (!uri) {
            if (!have_bin) return error = "buffer without uri and no BIN chunk", false;
            data = bin_chunk; // gltf::buffer::Source::Bin
        } else if (uri->str.rfind("data:", 0) == 0) {
            const size_t comma = uri->str.find(',');
            if (comma == std::string::npos || uri->str.find(";base64") == std::string::npos)
                return error = "unsupported data: uri", false;
            static const char *kB64 = "ABCDEFGHIJKLMNOPQRSTUVWXYZabcdefghijklmnopqrstuvwxyz0123456789+/";
            uint32_t acc = 0;
            int bits = 0;
            for (size_t k = comma + 1; k < uri->str.size(); k++) {
                const char *q = strchr(kB64, uri->str[k]);
                if (!q || !uri->str[k]) continue; // '=' padding
                acc = (acc << 6) | (uint32_t)(q - kB64);
                bits += 6;
                if (bits >= 8) bits -= 8, data.push_back((uint8_t)(acc >> bits));
            }
        } else if (!read_file(uri->str[0] == '/' ? uri->str : base + uri->str, data)) {
            return error = "cannot read buffer " + uri->str, false;
        }
        const size_t want = (size_t)b->index_or("byteLength", 0);
        if (data.size() < want) return error = "buffer shorter than byteLength", false;
        doc.buffers.push_back(std::move(data));
    }
    return true;
}

// gltf::scene::Transform::matrix(): `matrix`, or T * R * S from translation / rotation (x, y, z, w) / scale
Mat4 node_matrix(const Json &node) {
    if (const Json *m = node.get("matrix")) {
        if (m->kind == Json::Array && m->arr.size() == 16) {
            float a[16];
            for (int i = 0; i < 16; i++) a[i] = (float)m->arr[(size_t)i].num;
            return Mat4::from_cols_array(a);
        }
    }
    auto vec = [&](const char *key, size_t n, const float *dflt, float *out) {
        const Json *v = node.get(key);
        for (size_t i = 0; i < n; i++)
            out[i] = v && v->kind == Json::Array && v->arr.size() == n ? (float)v->arr[i].num : dflt[i];
    };
    const float t0[3] = {0, 0, 0}, r0[4] = {0, 0, 0, 1}, s0[3] = {1, 1, 1};
    float t[3], r[4], s[3];
    vec("translation", 3, t0, t);
    vec("rotation", 4, r0, r);
    vec("scale", 3, s0, s);
    return math::mat4_from_scale_rotation_translation({s[0], s[1], s[2]}, {r[0], r[1], r[2], r[3]}, {t[0], t[1], t[2]});
}

} // namespace

passes::Transform transform_from_mat4(const Mat4 &m) { // glam Mat4::to_scale_rotation_translation
    const float *c = m.c;
    const float det = c[0] * (c[5] * c[10] - c[6] * c[9]) - c[4] * (c[1] * c[10] - c[2] * c[9]) +
                      c[8] * (c[1] * c[6] - c[2] * c[5]);
    auto len = [](const float *v) { return std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]); };
    passes::Transform t;
    t.scale = {len(c) * (det < 0.0f ? -1.0f : 1.0f), len(c + 4), len(c + 8)};
    const float inv[3] = {1.0f / t.scale.x, 1.0f / t.scale.y, 1.0f / t.scale.z};
    const float m00 = c[0] * inv[0], m01 = c[1] * inv[0], m02 = c[2] * inv[0];
    const float m10 = c[4] * inv[1], m11 = c[5] * inv[1], m12 = c[6] * inv[1];
    const float m20 = c[8] * inv[2], m21 = c[9] * inv[2], m22 = c[10] * inv[2];
    Quat q; // glam Quat::from_rotation_axes
    if (m22 <= 0.0f) {
        const float dif10 = m11 - m00, omm22 = 1.0f - m22;
        if (dif10 <= 0.0f) {
            const float four_xsq = omm22 - dif10, inv4x = 0.5f / std::sqrt(four_xsq);
            q = {four_xsq * inv4x, (m01 + m10) * inv4x, (m02 + m20) * inv4x, (m12 - m21) * inv4x};
        } else {
            const float four_ysq = omm22 + dif10, inv4y = 0.5f / std::sqrt(four_ysq);
            q = {(m01 + m10) * inv4y, four_ysq * inv4y, (m12 + m21) * inv4y, (m20 - m02) * inv4y};
        }
    } else {
        const float sum10 = m11 + m00, opm22 = 1.0f + m22;
        if (sum10 <= 0.0f) {
            const float four_zsq = opm22 - sum10, inv4z = 0.5f / std::sqrt(four_zsq);
            q = {(m02 + m20) * inv4z, (m12 + m21) * inv4z, four_zsq * inv4z, (m01 - m10) * inv4z};
        } else {
            const float four_wsq = opm22 + sum10, inv4w = 0.5f / std::sqrt(four_wsq);
            q = {(m12 - m21) * inv4w, (m20 - m02) * inv4w, (m01 - m10) * inv4w, four_wsq * inv4w};
        }
    }
    t.orientation = q;
    t.position = {c[12], c[13], c[14]};
    return t;
}

bool load_gltf(const std::string &path, LoadedScene &out, std::string &error) {
    Document doc;
    if (!open_document(path, doc, error)) return false;

    // materials (gltf_loader.rs:589-628): texture handles stay "none", the cull path reads alpha_mode only
    std::vector<uint32_t> material_lookup;
    for (size_t i = 0; i < doc.count("materials"); i++) {
        const Json &m = *doc.at("materials", (int64_t)i);
        OrbitMaterialData md;
        memset(&md, 0, sizeof(md));
        const Json *pbr = m.get("pbrMetallicRoughness");
        auto factor = [](const Json *parent, const char *key, size_t n, const float *dflt, float *o) {
            const Json *v = parent ? parent->get(key) : nullptr;
            for (size_t k = 0; k < n; k++)
                o[k] = v && v->kind == Json::Array && v->arr.size() == n ? (float)v->arr[k].num : dflt[k];
        };
        const float one4[4] = {1, 1, 1, 1}, zero3[3] = {0, 0, 0};
        factor(pbr, "baseColorFactor", 4, one4, md.base_color);
        factor(&m, "emissiveFactor", 3, zero3, md.emissive_factor);
        md.metallic_factor = pbr ? (float)pbr->number_or("metallicFactor", 1.0) : 1.0f;
        md.roughness_factor = pbr ? (float)pbr->number_or("roughnessFactor", 1.0) : 1.0f;
        const Json *occ = m.get("occlusionTexture");
        md.occlusion_factor = occ ? (float)occ->number_or("strength", 1.0) : 0.0f; // map_or(0.0, strength)
        md.alpha_cutoff = (float)m.number_or("alphaCutoff", 0.0);                  // alpha_cutoff().unwrap_or(0.0)
        md.base_texture_index = md.normal_texture_index = md.metallic_roughness_texture_index =
            md.occlusion_texture_index = md.emissive_texture_index = ORBIT_NONE;
        const Json *am = m.get("alphaMode"); // assets/mod.rs:142-150
        md.alpha_mode = am && am->kind == Json::String ? (am->str == "MASK" ? 1u : am->str == "BLEND" ? 2u : 0u) : 0u;
        material_lookup.push_back(out.assets.add_material(md));
    }

    // meshes (gltf_loader.rs:630-640 -> load_gltf_mesh :423-509)
    std::vector<int32_t> mesh_lookup;
    for (size_t mi = 0; mi < doc.count("meshes"); mi++) {
        const Json &mesh = *doc.at("meshes", (int64_t)mi);
        const Json *prims = mesh.get("primitives");
        assets::MeshData data;
        for (size_t pi = 0; prims && pi < prims->arr.size(); pi++) {
            const Json &prim = prims->arr[pi];
            if (prim.index_or("mode", 4) != 4) return error = "primitive mode is not TRIANGLES (assert_eq!, :441)", false;
            const int64_t mat = prim.index_or("material", -1);
            if (mat < 0 || (size_t)mat >= material_lookup.size())
                return error = "primitive without material (unwrap, :439)", false;
            const Json *attrs = prim.get("attributes");
            Document::View pos, idx;
            if (!attrs || !doc.view(attrs->index_or("POSITION", -1), pos) || pos.components != 3)
                return error = "primitive without POSITION: " + doc.err, false;
            if (!doc.view(prim.index_or("indices", -1), idx) || idx.components != 1)
                return error = "primitive without indices (unwrap, :466): " + doc.err, false;
            std::vector<float> positions(3 * pos.count);
            for (size_t v = 0; v < pos.count; v++)
                for (int c = 0; c < 3; c++) positions[3 * v + (size_t)c] = (float)pos.get(v, c);
            std::vector<uint32_t> indices(idx.count / 3 * 3);
            for (size_t k = 0; k < indices.size(); k++) {
                indices[k] = (uint32_t)idx.get(k, 0);
                if (indices[k] >= pos.count) return error = "index out of range", false;
            }
            data.add_submesh(positions.data(), pos.count, indices.data(), indices.size(),
                             (uint16_t)material_lookup[(size_t)mat]);
        }
        data.compute_bounds(); // :507 (the per-primitive bounds of :495-505 are overwritten by it)
        mesh_lookup.push_back((int32_t)out.assets.add_mesh(data));
    }

    // nodes of the first scene, depth first (gltf_loader.rs:642-673)
    const Json *scene = doc.at("scenes", doc.root.index_or("scene", 0));
    if (!scene) scene = doc.at("scenes", 0);
    if (!scene) return error = "document without a scene (unwrap, :671)", false;
    struct Frame {
        int64_t node;
        Mat4 parent;
        bool has_parent;
    };
    std::vector<Frame> stack;
    const Json *roots = scene->get("nodes");
    for (size_t i = roots ? roots->arr.size() : 0; i-- > 0;) stack.push_back({(int64_t)roots->arr[i].num, Mat4::identity(), false});
    size_t guard = 0;
    while (!stack.empty()) {
        const Frame f = stack.back();
        stack.pop_back();
        const Json *node = doc.at("nodes", f.node);
        if (!node || ++guard > 4u * (doc.count("nodes") + 1)) return error = "bad node graph", false;
        Mat4 m = node_matrix(*node);
        if (f.has_parent) m = f.parent.mul(m);
        LoadedEntity e;
        e.transform = transform_from_mat4(m);
        const int64_t mesh = node->index_or("mesh", -1);
        e.mesh = mesh >= 0 && (size_t)mesh < mesh_lookup.size() ? mesh_lookup[(size_t)mesh] : -1;
        if (const Json *n = node->get("name"))
            if (n->kind == Json::String) e.name = n->str;
        out.entities.push_back(std::move(e));
        if (const Json *ch = node->get("children"))
            for (size_t i = ch->arr.size(); i-- > 0;) stack.push_back({(int64_t)ch->arr[i].num, m, true});
    }
    return true;
}

} // namespace gltf_loader
} // namespace orbit
