/* orbit_oracle_mt.c — the CPU oracle over all host cores.
 *
 * TEST INFRASTRUCTURE ONLY (see orbit_oracle.c): used by bench.py's cpu_baseline leg and by
 * tests/.  Pinned like the oracle it wraps (oracle/orbit_oracle.c header).
 *
 * Nothing is restated here: the frame (entity_cull + meshlet_cull, any occlusion pass) is
 * the sequential oracle run on contiguous shards by OpenMP threads and concatenated in
 * shard order — the same argument as the multi-GPU path: appends are in
 * canonical order inside a shard, shards are contiguous in entity / record order, so the
 * concatenation is the sequential result.  Entity shards are cut at multiples of 32 (entity
 * bitset words stay thread-private); meshlet visibility words belong to one record each.
 * tests/test_oracle_cpu.py checks the output against the sequential oracle byte for byte. */
#include <omp.h>
#include <stdlib.h>
#include <string.h>

#include "../include/orbit_abi.h"

uint32_t oracle_entity_cull(const OrbitGpuCullInfo *ci, const uint8_t *entity_draw_buffer, uint32_t draw_first,
                            uint32_t entity_draw_count, uint32_t exact_range, const OrbitMeshInfo *mesh_infos,
                            const OrbitEntityData *entities, uint8_t *dispatch_buffer, uint32_t dispatch_capacity,
                            uint32_t *visibility, const float *pyr, uint32_t pw, uint32_t ph);
uint32_t oracle_meshlet_cull(const OrbitGpuCullInfo *ci, const uint8_t *dispatch_buffer, const OrbitMeshlet *meshlets,
                             uint8_t *draw_buffer, uint32_t draw_capacity, const OrbitEntityData *entities,
                             const OrbitMaterialData *materials, uint32_t *meshlet_visibility, const float *pyr,
                             uint32_t pw, uint32_t ph);

static uint32_t rd32(const uint8_t *p) {
    uint32_t v;
    memcpy(&v, p, 4);
    return v;
}
static void wr32(uint8_t *p, uint32_t v) { memcpy(p, &v, 4); }

int oracle_mt_max_threads(void) { return omp_get_max_threads(); }

/* Whole frame on `threads` threads.  Returns 0, or -1 if a thread could not allocate.
 * dropped[0] / dropped[1]: records / commands that did not fit the capacities. */
int32_t oracle_cull_frame_mt(const OrbitGpuCullInfo *ci, const uint8_t *entity_draw_buffer, uint32_t entity_draw_count,
                             const OrbitMeshInfo *mesh_infos, const OrbitEntityData *entities,
                             const OrbitMeshlet *meshlets, const OrbitMaterialData *materials,
                             uint8_t *dispatch_buffer, uint32_t dispatch_capacity, uint8_t *draw_buffer,
                             uint32_t draw_capacity, uint32_t *visibility, uint32_t *meshlet_visibility,
                             const float *pyr, uint32_t pw, uint32_t ph, int threads, uint32_t dropped[2]) {
    if (threads < 1) threads = 1;
    const uint32_t in_buffer = rd32(entity_draw_buffer);
    const uint32_t n = entity_draw_count < in_buffer ? entity_draw_count : in_buffer;
    const uint64_t words = ((uint64_t)n + 31u) / 32u;
    uint8_t **part = (uint8_t **)calloc((size_t)threads, sizeof(*part));
    uint64_t *cnt = (uint64_t *)calloc((size_t)threads + 1, sizeof(*cnt));
    int failed = 0;
    if (!part || !cnt) return -1;

    /* ---- entity stage: count (capacity 0 => everything "dropped"), then fill exactly ---- */
#pragma omp parallel for num_threads(threads) schedule(static, 1)
    for (int t = 0; t < threads; t++) {
        uint32_t b = (uint32_t)(words * (uint64_t)t / (uint64_t)threads * 32u);
        uint32_t e = (uint32_t)(words * (uint64_t)(t + 1) / (uint64_t)threads * 32u);
        if (b > n) b = n;
        if (e > n) e = n;
        uint8_t hdr[ORBIT_DISPATCH_HEADER];
        /* the counting pass must not touch the bitset: pass 2 rewrites it from the same inputs in the fill pass */
        uint32_t *scratch_vis = NULL;
        if (visibility && ci->occlusion_pass == 2) {
            scratch_vis = (uint32_t *)malloc(((size_t)(e - b) / 32 + 2) * 4);
            if (!scratch_vis) {
                failed = 1;
                continue;
            }
            memcpy(scratch_vis, visibility + b / 32, ((size_t)(e - b + 31) / 32) * 4);
        }
        const uint32_t *vis_for_count = scratch_vis ? scratch_vis - b / 32 : visibility;
        uint32_t need = oracle_entity_cull(ci, entity_draw_buffer, b, e - b, 1, mesh_infos, entities, hdr, 0,
                                           (uint32_t *)vis_for_count, pyr, pw, ph);
        free(scratch_vis);
        part[t] = (uint8_t *)malloc(ORBIT_DISPATCH_HEADER + (size_t)need * sizeof(OrbitMeshletDispatch) + 16);
        if (!part[t]) {
            failed = 1;
            continue;
        }
        (void)oracle_entity_cull(ci, entity_draw_buffer, b, e - b, 1, mesh_infos, entities, part[t], need, visibility,
                                 pyr, pw, ph);
        cnt[t + 1] = need;
    }
    if (failed) goto fail;
    for (int t = 0; t < threads; t++) cnt[t + 1] += cnt[t];
    {
        const uint64_t total = cnt[threads];
        const uint64_t kept = total < dispatch_capacity ? total : dispatch_capacity;
        uint8_t *out = dispatch_buffer + ORBIT_DISPATCH_HEADER;
#pragma omp parallel for num_threads(threads) schedule(static, 1)
        for (int t = 0; t < threads; t++) {
            uint64_t lo = cnt[t], hi = cnt[t + 1];
            if (hi > kept) hi = kept;
            if (lo < hi)
                memcpy(out + lo * sizeof(OrbitMeshletDispatch), part[t] + ORBIT_DISPATCH_HEADER,
                       (size_t)(hi - lo) * sizeof(OrbitMeshletDispatch));
            free(part[t]);
            part[t] = NULL;
        }
        wr32(dispatch_buffer + 0, (uint32_t)kept);
        wr32(dispatch_buffer + 4, 1);
        wr32(dispatch_buffer + 8, 1);
        dropped[0] = (uint32_t)(total - kept);
    }

    /* ---- meshlet stage: contiguous record ranges of the canonical dispatch list ---- */
    {
        const uint32_t nrec = rd32(dispatch_buffer);
        const OrbitMeshletDispatch *recs = (const OrbitMeshletDispatch *)(dispatch_buffer + ORBIT_DISPATCH_HEADER);
        memset(cnt, 0, ((size_t)threads + 1) * sizeof(*cnt));
#pragma omp parallel for num_threads(threads) schedule(static, 1)
        for (int t = 0; t < threads; t++) {
            const uint32_t b = (uint32_t)((uint64_t)nrec * (uint64_t)t / (uint64_t)threads);
            const uint32_t e = (uint32_t)((uint64_t)nrec * (uint64_t)(t + 1) / (uint64_t)threads);
            uint64_t meshlets_in = 0;
            for (uint32_t r = b; r < e; r++) meshlets_in += recs[r].meshlet_count;
            uint8_t *slice = (uint8_t *)malloc(ORBIT_DISPATCH_HEADER + (size_t)(e - b) * sizeof(*recs) + 16);
            part[t] = (uint8_t *)malloc(ORBIT_DRAW_HEADER + (size_t)meshlets_in * sizeof(OrbitMeshletDrawCommand) + 16);
            if (!slice || !part[t]) {
                free(slice);
                failed = 1;
                continue;
            }
            wr32(slice + 0, e - b);
            wr32(slice + 4, 1);
            wr32(slice + 8, 1);
            memcpy(slice + ORBIT_DISPATCH_HEADER, recs + b, (size_t)(e - b) * sizeof(*recs));
            (void)oracle_meshlet_cull(ci, slice, meshlets, part[t], (uint32_t)meshlets_in, entities, materials,
                                      meshlet_visibility, pyr, pw, ph);
            cnt[t + 1] = rd32(part[t]);
            free(slice);
        }
        if (failed) goto fail;
        for (int t = 0; t < threads; t++) cnt[t + 1] += cnt[t];
        const uint64_t total = cnt[threads];
        const uint64_t kept = total < draw_capacity ? total : draw_capacity;
        uint8_t *out = draw_buffer + ORBIT_DRAW_HEADER;
#pragma omp parallel for num_threads(threads) schedule(static, 1)
        for (int t = 0; t < threads; t++) {
            uint64_t lo = cnt[t], hi = cnt[t + 1];
            if (hi > kept) hi = kept;
            if (lo < hi)
                memcpy(out + lo * sizeof(OrbitMeshletDrawCommand), part[t] + ORBIT_DRAW_HEADER,
                       (size_t)(hi - lo) * sizeof(OrbitMeshletDrawCommand));
            free(part[t]);
            part[t] = NULL;
        }
        wr32(draw_buffer, (uint32_t)kept);
        dropped[1] = (uint32_t)(total - kept);
    }
    free(part);
    free(cnt);
    return 0;
fail:
    for (int t = 0; t < threads; t++) free(part[t]);
    free(part);
    free(cnt);
    return -1;
}
