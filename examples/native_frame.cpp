// native_frame.cpp — a renderer stand-in in plain C++/HIP: no Python, no torch in the process.
//
// Loads an `.orbitscene` dump (orbit_amd/orbitscene.py: the buffers the renderer uploads, verbatim), puts them in
// device memory with the HIP runtime, and drives the cull path through the C ABI exactly as the reference's frame does
// (src/passes/forward.rs:286-298 -> draw_gen.rs:239-325): entity_cull -> meshlet_cull with the dump's GpuCullInfo, and
// depth_reduce when the dump carries a depth buffer (draw_gen.rs:510-566).  The outputs are written to a file so that
// tests/test_native_example_gpu.py can compare them bit-exactly with the CPU oracle run on the same dump.
//
//   usage: native_frame <scene.orbitscene> <out.bin> [replays] [--stream]
//   --stream: cull from the derived meshlet streams (orbit_meshlet_stream_*), built once after the meshlet upload
//   out.bin: u64 sizes[3] {dispatch buffer, draw buffer, pyramid}, then the three blobs
//   replays > 0: the frame is also captured into a HIP graph (every entry point only enqueues on the caller's stream,
//   so stream capture just works) and replayed that many times; eager and replayed times per frame are printed.
//   The outputs written are those of the LAST execution, whichever way it ran.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "../include/orbit_abi_ext.h"

namespace {

struct Section {
    const uint8_t *data = nullptr;
    size_t size = 0;
};

[[noreturn]] void die(const char *what, const char *detail = "") {
    std::fprintf(stderr, "native_frame: %s%s\n", what, detail);
    std::exit(1);
}

void hip_check(hipError_t e, const char *what) {
    if (e != hipSuccess) die(what, hipGetErrorString(e));
}

void orbit_check(OrbitCtx *ctx, int32_t rc, const char *what) {
    if (rc != ORBIT_OK) {
        std::fprintf(stderr, "native_frame: %s: status %d: %s\n", what, rc, orbit_last_error(ctx));
        std::exit(1);
    }
}

std::vector<uint8_t> read_file(const char *path) {
    FILE *f = std::fopen(path, "rb");
    if (!f) die("cannot open ", path);
    std::fseek(f, 0, SEEK_END);
    const long n = std::ftell(f);
    std::fseek(f, 0, SEEK_SET);
    std::vector<uint8_t> buf((size_t)n);
    if (n && std::fread(buf.data(), 1, (size_t)n, f) != (size_t)n) die("short read on ", path);
    std::fclose(f);
    return buf;
}

// section table of orbit_amd/orbitscene.py: magic, u32 count, u32 0, then {char name[24]; u64 offset; u64 size}
std::map<std::string, Section> parse_scene(const std::vector<uint8_t> &raw) {
    if (raw.size() < 16 || std::memcmp(raw.data(), "ORBSCN01", 8) != 0) die("not an .orbitscene file");
    uint32_t n;
    std::memcpy(&n, raw.data() + 8, 4);
    std::map<std::string, Section> out;
    for (uint32_t i = 0; i < n; i++) {
        const uint8_t *e = raw.data() + 16 + 40 * (size_t)i;
        if (e + 40 > raw.data() + raw.size()) die("section table runs past the end of the file");
        char name[25] = {0};
        std::memcpy(name, e, 24);
        uint64_t off, size;
        std::memcpy(&off, e + 24, 8);
        std::memcpy(&size, e + 32, 8);
        if (off + size > raw.size()) die("section runs past the end of the file: ", name);
        out[name] = Section{raw.data() + off, (size_t)size};
    }
    return out;
}

void *to_device(const Section &s, size_t min_bytes = 16) {
    void *d = nullptr;
    const size_t bytes = s.size > min_bytes ? s.size : min_bytes;
    hip_check(hipMalloc(&d, bytes), "hipMalloc: ");
    hip_check(hipMemset(d, 0, bytes), "hipMemset: ");
    if (s.size) hip_check(hipMemcpy(d, s.data, s.size, hipMemcpyHostToDevice), "hipMemcpy H2D: ");
    return d;
}

std::vector<uint8_t> to_host(const void *d, size_t bytes) {
    std::vector<uint8_t> h(bytes);
    if (bytes) hip_check(hipMemcpy(h.data(), d, bytes, hipMemcpyDeviceToHost), "hipMemcpy D2H: ");
    return h;
}

} // namespace

int main(int argc, char **argv) {
    bool use_stream = false;
    if (argc > 3 && std::string(argv[argc - 1]) == "--stream") {
        use_stream = true;
        argc--;
    }
    if (argc != 3 && argc != 4) die("usage: native_frame <scene.orbitscene> <out.bin> [replays] [--stream]");
    const int replays = argc == 4 ? std::atoi(argv[3]) : 0;
    const std::vector<uint8_t> raw = read_file(argv[1]);
    const std::map<std::string, Section> sec = parse_scene(raw);
    for (const char *need : {"entity_draws", "mesh_infos", "entities", "meshlets", "materials", "cull_info"})
        if (!sec.count(need)) die("the dump lacks section ", need);
    if (sec.at("cull_info").size < sizeof(OrbitGpuCullInfo)) die("cull_info is shorter than 400 B");
    OrbitGpuCullInfo ci;
    std::memcpy(&ci, sec.at("cull_info").data, sizeof(ci));
    if (ci.occlusion_pass != 0) die("this stand-in runs the single-pass frame: the dump's cull_info must have occlusion_pass 0");

    uint32_t n_draws;
    std::memcpy(&n_draws, sec.at("entity_draws").data, 4);
    const size_t n_mesh = sec.at("mesh_infos").size / sizeof(OrbitMeshInfo);
    const OrbitMeshInfo *mesh_infos = reinterpret_cast<const OrbitMeshInfo *>(sec.at("mesh_infos").data);
    const uint8_t *draws = sec.at("entity_draws").data + ORBIT_ENTITY_DRAW_HEADER;
    // capacities the way the renderer sizes them from the scene: one record per 32 LOD-0 meshlets of every draw
    uint64_t max_records = 0, max_meshlets = 0;
    for (uint32_t i = 0; i < n_draws; i++) {
        OrbitEntityDraw d;
        std::memcpy(&d, draws + sizeof(OrbitEntityDraw) * (size_t)i, sizeof(d));
        if (d.mesh_index >= n_mesh) die("entity draw references a mesh past mesh_infos");
        uint32_t most = 0;
        for (uint32_t l = 0; l < mesh_infos[d.mesh_index].lod_count && l < ORBIT_MAX_MESH_LODS; l++)
            most = mesh_infos[d.mesh_index].mesh_lods[l].meshlet_count > most ? mesh_infos[d.mesh_index].mesh_lods[l].meshlet_count : most;
        max_records += (most + 31) / 32;
        max_meshlets += most;
    }
    OrbitCaps caps;
    orbit_default_caps(&caps);
    caps.max_entities = n_draws + 256;
    caps.max_dispatches = (uint32_t)max_records + 8;
    caps.max_draws = (uint32_t)max_meshlets + 8;
    OrbitCtx *ctx = nullptr;
    orbit_check(nullptr, orbit_ctx_create(0, &caps, &ctx), "orbit_ctx_create");

    hipStream_t stream;
    hip_check(hipStreamCreate(&stream), "hipStreamCreate: ");
    void *d_draws = to_device(sec.at("entity_draws")), *d_mesh = to_device(sec.at("mesh_infos"));
    void *d_ent = to_device(sec.at("entities")), *d_meshlets = to_device(sec.at("meshlets"));
    void *d_mat = to_device(sec.at("materials"));
    const size_t disp_bytes = ORBIT_DISPATCH_HEADER + 16 * (size_t)caps.max_dispatches;
    const size_t draw_bytes = ORBIT_DRAW_HEADER + 28 * (size_t)caps.max_draws;
    void *d_disp = nullptr, *d_cmds = nullptr;
    hip_check(hipMalloc(&d_disp, disp_bytes), "hipMalloc: ");
    hip_check(hipMalloc(&d_cmds, draw_bytes), "hipMalloc: ");
    hip_check(hipMemset(d_disp, 0, disp_bytes), "hipMemset: ");
    hip_check(hipMemset(d_cmds, 0, draw_bytes), "hipMemset: ");

    // depth pyramid of the dumped frame, if any
    void *d_depth = nullptr, *d_pyr = nullptr;
    size_t pyr_bytes = 0;
    uint32_t wh[2] = {0, 0};
    if (sec.count("depth") && sec.count("screen")) {
        std::memcpy(wh, sec.at("screen").data, 8);
        if ((size_t)wh[0] * wh[1] * 4 != sec.at("depth").size) die("depth section does not match screen");
        OrbitDepthPyramidDesc desc;
        orbit_check(ctx, orbit_depth_pyramid_desc(wh[0], wh[1], &desc), "orbit_depth_pyramid_desc");
        pyr_bytes = 4 * (size_t)desc.total_texels;
        d_depth = to_device(sec.at("depth"));
        hip_check(hipMalloc(&d_pyr, pyr_bytes), "hipMalloc: ");
    }

    OrbitEntityCullBufs eb;
    std::memset(&eb, 0, sizeof(eb));
    eb.entity_draw_buffer = d_draws;
    eb.mesh_info_buffer = d_mesh;
    eb.meshlet_dispatch_buffer = d_disp;
    eb.entity_buffer = d_ent;
    eb.dispatch_capacity = caps.max_dispatches;
    OrbitMeshletCullBufs mb;
    std::memset(&mb, 0, sizeof(mb));
    mb.meshlet_dispatch_buffer = d_disp;
    mb.meshlet_buffer = d_meshlets;
    mb.draw_commands_buffer = d_cmds;
    mb.entity_buffer = d_ent;
    mb.material_buffer = d_mat;
    mb.dispatch_capacity = caps.max_dispatches;
    mb.draw_capacity = caps.max_draws;
    mb.material_count = (uint32_t)(sec.at("materials").size / 80);
    // what GpuAssets::add_mesh does after uploading meshlets (assets/mod.rs:441-445), when the streams are in use
    OrbitMeshletStream *ms = nullptr;
    if (use_stream) {
        const uint64_t n_meshlets = sec.at("meshlets").size / sizeof(OrbitMeshlet);
        orbit_check(ctx, orbit_meshlet_stream_create(ctx, 0, n_meshlets, &ms), "orbit_meshlet_stream_create");
        orbit_check(ctx, orbit_meshlet_stream_update(ctx, ms, d_meshlets, 0, n_meshlets, stream),
                    "orbit_meshlet_stream_update");
        orbit_check(ctx, orbit_ctx_bind_meshlet_stream(ctx, ms), "orbit_ctx_bind_meshlet_stream");
    }
    auto frame = [&] { // the calls of one frame, in the reference's order; nothing here allocates or synchronises
        orbit_check(ctx, orbit_entity_cull(ctx, &ci, &eb, n_draws, stream), "orbit_entity_cull");
        orbit_check(ctx, orbit_meshlet_cull(ctx, &ci, &mb, stream), "orbit_meshlet_cull");
        if (d_pyr)
            orbit_check(ctx, orbit_depth_reduce(ctx, (const float *)d_depth, wh[0], wh[1], (float *)d_pyr, stream),
                        "orbit_depth_reduce");
    };
    frame();
    orbit_check(ctx, orbit_ctx_status(ctx, stream, 1), "orbit_ctx_status");
    if (replays > 0) {
        hipEvent_t e0, e1;
        hip_check(hipEventCreate(&e0), "hipEventCreate: ");
        hip_check(hipEventCreate(&e1), "hipEventCreate: ");
        float eager_ms = 0, graph_ms = 0;
        hip_check(hipEventRecord(e0, stream), "hipEventRecord: ");
        for (int i = 0; i < replays; i++) frame();
        hip_check(hipEventRecord(e1, stream), "hipEventRecord: ");
        hip_check(hipEventSynchronize(e1), "hipEventSynchronize: ");
        hip_check(hipEventElapsedTime(&eager_ms, e0, e1), "hipEventElapsedTime: ");
        hipGraph_t graph;
        hipGraphExec_t exec;
        hip_check(hipStreamBeginCapture(stream, hipStreamCaptureModeGlobal), "hipStreamBeginCapture: ");
        frame();
        hip_check(hipStreamEndCapture(stream, &graph), "hipStreamEndCapture: ");
        hip_check(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0), "hipGraphInstantiate: ");
        hip_check(hipGraphLaunch(exec, stream), "hipGraphLaunch: "); // first launch uploads the graph
        hip_check(hipEventRecord(e0, stream), "hipEventRecord: ");
        for (int i = 0; i < replays; i++) hip_check(hipGraphLaunch(exec, stream), "hipGraphLaunch: ");
        hip_check(hipEventRecord(e1, stream), "hipEventRecord: ");
        hip_check(hipEventSynchronize(e1), "hipEventSynchronize: ");
        hip_check(hipEventElapsedTime(&graph_ms, e0, e1), "hipEventElapsedTime: ");
        orbit_check(ctx, orbit_ctx_status(ctx, stream, 1), "orbit_ctx_status");
        std::printf("frame: eager %.1f us, graph replay %.1f us (x%d)\n", 1e3 * eager_ms / replays, 1e3 * graph_ms / replays,
                    replays);
        (void)hipGraphExecDestroy(exec);
        (void)hipGraphDestroy(graph);
        (void)hipEventDestroy(e0);
        (void)hipEventDestroy(e1);
    }

    const std::vector<uint8_t> h_disp = to_host(d_disp, disp_bytes), h_cmds = to_host(d_cmds, draw_bytes);
    const std::vector<uint8_t> h_pyr = to_host(d_pyr, pyr_bytes);
    uint32_t n_rec, n_cmd;
    std::memcpy(&n_rec, h_disp.data(), 4);
    std::memcpy(&n_cmd, h_cmds.data(), 4);
    std::printf("entity draws %u -> dispatch records %u -> draw commands %u; pyramid %zu B\n", n_draws, n_rec, n_cmd,
                pyr_bytes);
    std::printf("meshlet culls served from the derived streams: %llu\n",
                (unsigned long long)orbit_ctx_meshlet_stream_culls(ctx));
    FILE *f = std::fopen(argv[2], "wb");
    if (!f) die("cannot write ", argv[2]);
    const uint64_t sizes[3] = {h_disp.size(), h_cmds.size(), h_pyr.size()};
    std::fwrite(sizes, 8, 3, f);
    std::fwrite(h_disp.data(), 1, h_disp.size(), f);
    std::fwrite(h_cmds.data(), 1, h_cmds.size(), f);
    if (!h_pyr.empty()) std::fwrite(h_pyr.data(), 1, h_pyr.size(), f);
    std::fclose(f);

    orbit_ctx_bind_meshlet_stream(ctx, nullptr);
    orbit_meshlet_stream_destroy(ms);
    orbit_ctx_destroy(ctx);
    for (void *p : {d_draws, d_mesh, d_ent, d_meshlets, d_mat, d_disp, d_cmds, d_depth, d_pyr})
        if (p) (void)hipFree(p);
    (void)hipStreamDestroy(stream);
    return 0;
}
