// meshlet_lists.hip — the outputs of meshlet_cull that are not a MeshletDrawCommandBuffer, and the way back:
//   task_records_kernel            per-record task payloads of the mesh-shading path (orbit_meshlet_task_cull)
//   visible_records_expand_kernel  a gathered record list -> commands on the receiving GPU (shaders/meshlet_cull.comp:216-230)
//   exchange_scatter/wait_kernel   a rank's list into every peer's buffer over xGMI, signalled on the device (orbit_exchange_list)
//   compact_segments_kernel        the segments of an ncclAllGather -> one contiguous list (orbit_allgather_list)
#include "meshlet_common.h"

namespace orbit {

namespace {

// A survivor of a gathered list: {meshlet_index, entity_index}; lanes past the list read the zero page, so the code is
// straight-line and every vmcnt wait is exact.
struct ExpandChunk {
    uint2 item;
    bool valid;
};

// STREAM: the three command words come from the derived 12-B stream (`cmd`, orbit_meshlet_stream_*) instead of bytes
// 16..31 of the Meshlet: a third of the lines per survivor.
template <bool STREAM>
__device__ __forceinline__ uint4 expand_gather(const OrbitMeshlet *meshlets, const MeshletStreamView &ms,
                                               const ExpandChunk &c, const void *zero_page, int32_t *status) {
    if constexpr (STREAM) {
        const bool in = c.valid && ms.covers(c.item.x);
        if (c.valid && !in) latch_status(status, ORBIT_E_RANGE);
        const uint32_t *c3 = in ? ms.cmd + (size_t)c.item.x * 3u : reinterpret_cast<const uint32_t *>(zero_page);
        return make_uint4(0u, c3[0], c3[1], c3[2]);
    } else {
        return *(c.valid ? reinterpret_cast<const uint4 *>(meshlets + c.item.x) + 1
                         : reinterpret_cast<const uint4 *>(zero_page));
    }
}

constexpr uint32_t kVrBlock = 1024; // records per block of the record list's expansion: 256 threads x 4

// Commands from a (gathered) record list, in list order = the canonical order: orbit_expand_visible_records.  A count
// launch sums the survivors of every block of 1024 records; here a block adds up the blocks before it, a wave takes 64
// records at a time, lays their survivors out in LDS as (record lane, bit) codes at the records' prefix offsets and
// then writes them 64 commands at a time through LDS (every store covers 256 contiguous bytes).
__global__ __launch_bounds__(256) void visible_records_pop_kernel(const uint8_t *records, uint32_t *block_pop,
                                                                  uint32_t max_blocks) {
    __shared__ uint32_t smem[4];
    const uint32_t n = *reinterpret_cast<const uint32_t *>(records);
    if (blockIdx.x * kVrBlock >= n || blockIdx.x >= max_blocks) return;
    const uint32_t *recs = reinterpret_cast<const uint32_t *>(records + ORBIT_VISIBLE_HEADER);
    uint32_t pop = 0;
#pragma unroll
    for (uint32_t k = 0; k < 4; k++) {
        const uint32_t r = blockIdx.x * kVrBlock + k * 256u + threadIdx.x;
        pop += r < n ? (uint32_t)__popc(recs[3 * (size_t)r + 2]) : 0u;
    }
    pop = wave_reduce_add(pop);
    if ((threadIdx.x & 63) == 0) smem[threadIdx.x >> 6] = pop;
    __syncthreads();
    if (threadIdx.x == 0) block_pop[blockIdx.x] = smem[0] + smem[1] + smem[2] + smem[3];
}

template <bool STREAM>
__global__ __launch_bounds__(256) void visible_records_expand_kernel(const uint8_t *records, const uint32_t *block_pop,
                                                                     uint32_t max_blocks, const OrbitMeshlet *meshlets,
                                                                     const MeshletStreamView ms, uint8_t *draw_buffer,
                                                                     uint32_t draw_capacity, const void *zero_page,
                                                                     int32_t *status) {
    __shared__ uint32_t s_cmd[4][64 * 7];
    __shared__ uint16_t s_code[4][64 * 32];
    __shared__ uint32_t smem[8];
    __shared__ uint32_t s_chunk[kVrBlock / 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t n = *reinterpret_cast<const uint32_t *>(records);
    // the grid covers draw_capacity / 1024 blocks of records (every record holds a survivor: more records than that
    // cannot fit) and the scratch max_blocks: a longer list is cut there, and says so
    const uint32_t all_blocks = (n + kVrBlock - 1u) / kVrBlock;
    const uint32_t nblocks = min(min(all_blocks, max_blocks), gridDim.x);
    if (all_blocks > nblocks && blockIdx.x == 0 && threadIdx.x == 0) latch_status(status, ORBIT_E_CAPACITY);
    if (blockIdx.x >= nblocks && blockIdx.x != 0) return;
    const uint32_t upto = blockIdx.x == 0 ? nblocks : blockIdx.x;
    uint32_t part = 0;
    for (uint32_t i = threadIdx.x; i < upto; i += 256u) part += block_pop[i];
    part = wave_reduce_add(part);
    if (lane == 0) smem[wave] = part;
    __syncthreads();
    const uint32_t before = smem[0] + smem[1] + smem[2] + smem[3];
    if (blockIdx.x == 0 && threadIdx.x == 0) { // `count` header of the command list
        *reinterpret_cast<uint32_t *>(draw_buffer) = min(before, draw_capacity);
        if (before > draw_capacity) latch_status(status, ORBIT_E_CAPACITY);
    }
    if (blockIdx.x >= nblocks) return;
    const uint32_t *recs = reinterpret_cast<const uint32_t *>(records + ORBIT_VISIBLE_HEADER);
    // the block's 16 chunks of 64 records: wave w takes chunks 4w .. 4w+3; their survivor counts first
    uint32_t ent[4], off[4], mask[4];
#pragma unroll
    for (uint32_t k = 0; k < 4; k++) {
        const uint32_t r = blockIdx.x * kVrBlock + (wave * 4u + k) * 64u + (uint32_t)lane;
        const bool in = r < n;
        ent[k] = in ? recs[3 * (size_t)r + 0] : 0u;
        off[k] = in ? recs[3 * (size_t)r + 1] : 0u;
        mask[k] = in ? recs[3 * (size_t)r + 2] : 0u;
        const uint32_t cs = wave_reduce_add((uint32_t)__popc(mask[k]));
        if (lane == 0) s_chunk[wave * 4u + k] = cs;
    }
    __syncthreads();
    uint32_t chunk_base = blockIdx.x == 0 ? 0u : before;
    for (uint32_t c = 0; c < wave * 4u; c++) chunk_base += s_chunk[c];
    uint32_t *cmd = s_cmd[wave];
    uint16_t *code = s_code[wave];
    uint8_t *out = draw_buffer + ORBIT_DRAW_HEADER;
#pragma unroll 1
    for (uint32_t k = 0; k < 4; k++) {
        const uint32_t pc = (uint32_t)__popc(mask[k]);
        const uint32_t inc = wave_inclusive_scan(pc);
        const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)inc, 63);
        {   // lane = record: its survivors' codes at its prefix offset, ascending bits
            uint32_t m = mask[k], at = inc - pc;
            while (m != 0u) {
                code[at++] = (uint16_t)((uint32_t)lane << 5 | (uint32_t)__builtin_ctz(m));
                m &= m - 1u;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        for (uint32_t j0 = 0; j0 < total; j0 += 64u) {
            const uint32_t j = j0 + (uint32_t)lane;
            const bool valid = j < total;
            const uint32_t cd = valid ? code[j] : 0u;
            const uint32_t mi = (uint32_t)__shfl((int)off[k], (int)(cd >> 5), 64) + (cd & 31u);
            const uint32_t en = (uint32_t)__shfl((int)ent[k], (int)(cd >> 5), 64);
            ExpandChunk ch;
            ch.item = make_uint2(mi, en);
            ch.valid = valid;
            const uint4 b = expand_gather<STREAM>(meshlets, ms, ch, zero_page, status);
            uint32_t *c = cmd + lane * 7;
            c[0] = (b.w >> 24) * 3u;                    // cmd_index_count = triangle_count * 3
            c[1] = 1u;                                  // cmd_instance_count
            c[2] = (b.z + ((b.w >> 16) & 0xFFu)) * 4u;  // cmd_first_index = (data_offset + vertex_count) * 4
            c[3] = b.z;                                 // cmd_vertex_offset = int(data_offset)
            c[4] = en;                                  // cmd_first_instance = entity_index
            c[5] = b.y;                                 // meshlet_vertex_offset
            c[6] = mi;                                  // meshlet_index
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            // the chunk's commands as one range-checked window: words past the list or the capacity are dropped
            const uint32_t first = (uint32_t)__builtin_amdgcn_readfirstlane((int)(chunk_base + j0));
            const uint32_t room = first < draw_capacity ? draw_capacity - first : 0u;
            const uint32_t bytes = min(min(total - j0, 64u), room) * 28u;
            const __amdgpu_buffer_rsrc_t win = __builtin_amdgcn_make_buffer_rsrc(out + (size_t)first * 28u, 0, bytes, 0x00020000);
#pragma unroll
            for (uint32_t q = 0; q < 7; q++) {
                const uint32_t w = q * 64u + (uint32_t)lane;
                __builtin_amdgcn_raw_buffer_store_b32(cmd[w], win, w * 4u, 0, 0);
            }
            __builtin_amdgcn_wave_barrier(); // cmd is rewritten by the next trip
        }
        chunk_base += total;
        __builtin_amdgcn_wave_barrier(); // code is rewritten by the next chunk
    }
}

// orbit_compact_segments — behind the all-gather of north_star, ncclAllGather of the ranks' lists as FIXED-CAPACITY
// segments {count | header | segment_capacity items} (a collective of a fixed size needs no count on the host: no
// read-back, no stream synchronisation, capturable): the segments' items, in rank order, become the one contiguous list
// {total | header | items} that the p2p exchange delivers — byte for byte.  blockIdx.y = segment; every workgroup adds up
// the counts of the segments before its own (a handful of words), then a plain dword copy: list items start 8 bytes into
// their buffers and land at 12-byte multiples, so 4 B is the common alignment; a wave moves 4 x 256 contiguous bytes per
// trip, loads first.
struct CompactSegmentsParams {
    const uint8_t *segments;
    uint8_t *out;
    uint32_t world, segment_capacity, out_capacity, header_bytes, stride;
    int32_t *status;
};

__global__ __launch_bounds__(256) void compact_segments_kernel(const CompactSegmentsParams p) {
    const uint32_t seg = blockIdx.y;
    const uint64_t seg_bytes = (uint64_t)p.header_bytes + (uint64_t)p.stride * p.segment_capacity;
    uint32_t first = 0, total = 0, mine = 0;
    for (uint32_t r = 0; r < p.world; r++) {
        // a count beyond the segment's capacity is a sender's overflow (it latched its own error): cut here
        const uint32_t c = min(*reinterpret_cast<const uint32_t *>(p.segments + seg_bytes * r), p.segment_capacity);
        if (r < seg) first += c;
        if (r == seg) mine = c;
        total += c;
    }
    const uint32_t room = first < p.out_capacity ? p.out_capacity - first : 0u;
    const uint64_t words = (uint64_t)min(mine, room) * p.stride / 4u;
    const uint32_t *src = reinterpret_cast<const uint32_t *>(p.segments + seg_bytes * seg + p.header_bytes);
    uint32_t *dst = reinterpret_cast<uint32_t *>(p.out + p.header_bytes + (uint64_t)first * p.stride);
    const uint64_t step = (uint64_t)gridDim.x * 256u * 4u;
    for (uint64_t w = ((uint64_t)blockIdx.x * 256u + (threadIdx.x & ~63u)) * 4u + (threadIdx.x & 63u); w < words;
         w += step) {
        uint32_t v[4];
#pragma unroll
        for (int k = 0; k < 4; k++) v[k] = w + 64u * k < words ? src[w + 64u * k] : 0u;
#pragma unroll
        for (int k = 0; k < 4; k++)
            if (w + 64u * k < words) dst[w + 64u * k] = v[k];
    }
    if (seg == 0 && blockIdx.x == 0 && threadIdx.x == 0) {
        uint32_t *hdr = reinterpret_cast<uint32_t *>(p.out);
        hdr[0] = min(total, p.out_capacity);
        for (uint32_t k = 1; k < p.header_bytes / 4u; k++) hdr[k] = 0u;
        if (total > p.out_capacity) latch_status(p.status, ORBIT_E_CAPACITY);
    }
}

// Mesh-shading path: one OrbitMeshTaskRecord per dispatch record from the record's should-draw ballot
// (forward_depth_prepass.task:225-228,255-257).  Thread = record; the drawn lanes in ascending order.
__global__ __launch_bounds__(256) void task_records_kernel(const MeshletCullParams p) {
    const uint32_t nrec = min(*reinterpret_cast<const uint32_t *>(p.dispatch_buffer), p.dispatch_capacity);
    if (blockIdx.x == 0 && threadIdx.x < kTicketPools) p.tickets[threadIdx.x * kTicketStride] = 0u; // next evaluation
    const uint32_t r = blockIdx.x * 256u + threadIdx.x;
    if (r >= nrec) return;
    const uint4 rec = reinterpret_cast<const uint4 *>(p.dispatch_buffer + ORBIT_DISPATCH_HEADER)[r];
    uint32_t m = p.tile_masks[r]; // word r = tile r/16, record r%16
    const uint32_t count = (uint32_t)__popc(m);
    uint32_t w[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int k = 0; k < 32; k++) {
        if (m != 0u) {
            w[k / 4] |= (uint32_t)__builtin_ctz(m) << (8 * (k % 4));
            m &= m - 1u;
        }
    }
    uint32_t *out = reinterpret_cast<uint32_t *>(p.task_records + r);
    out[0] = count;
    out[1] = rec.x; // entity_index
    out[2] = rec.y; // meshlet_offset
#pragma unroll
    for (int k = 0; k < 8; k++) out[3 + k] = w[k];
}

// ---------------------------------------------------------------------------------------------------------------
// orbit_exchange_list: the rank-ordered all-gather of the ranks' lists with NO host in the step — the counts and the
// completion travel the way the lists do, as stores into the peers' IPC-mapped memory.  Every rank owns a control
// block (ExchangeCtrl, mapped by all peers):
//   count[e & 1][r]  {items of rank r, epoch e}: stored by rank r into EVERY rank's block at the start of its scatter;
//   done[r]          epoch e: stored by rank r into a rank's block once its items for epoch e have reached that rank's
//                    output buffer (behind a system-scope fence of every storing workgroup);
//   epoch            this rank's last finished exchange: all ranks count exchanges in step, so epochs agree without
//                    ever being communicated — and a replayed graph finds the epoch where the last replay left it.
// Scatter launch: count out to every peer; wait (bounded) for every rank's count — those of the ranks before this one
// fix the offset; copy the list into every peer's buffer (blockIdx.y = destination: all links side by side); the last workgroup
// of a destination to finish raises done there.  Wait launch (one workgroup): until every rank's count and done for
// this epoch have arrived here, then the header {total, 0 ..} and the epoch.  What runs behind it in stream order
// reads the complete list.  The count slots alternate by epoch parity: a peer may start exchange e + 1 (it has seen
// this rank's done for e) while this rank's wait launch still reads the counts of e; it cannot start e + 2 before this
// rank's count for e + 1, i.e. before that wait launch has finished.
struct ExchangeCtrl {
    unsigned long long count[2][ORBIT_P2P_MAX_WORLD]; // items | epoch << 32
    unsigned long long done[ORBIT_P2P_MAX_WORLD];     // epoch
    uint32_t epoch;
    uint32_t left[ORBIT_P2P_MAX_WORLD];               // scatter workgroups of this rank done with destination d
};
static_assert(sizeof(ExchangeCtrl) <= ORBIT_P2P_CTRL_BYTES, "control block size");

constexpr uint32_t kExchangeSpinLimit = 1u << 22; // polls (with a pause each) before a wait gives up: seconds

__device__ __forceinline__ unsigned long long sys_load(const unsigned long long *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ void sys_store(unsigned long long *p, unsigned long long v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

__global__ __launch_bounds__(256) void exchange_scatter_kernel(const ExchangeListParams p) {
    __shared__ uint32_t s_first, s_mine, s_timeout;
    ExchangeCtrl *mine_ctrl = reinterpret_cast<ExchangeCtrl *>(p.ctrl[p.rank]);
    const uint32_t dst_rank = blockIdx.y;
    const uint32_t e = __hip_atomic_load(&mine_ctrl->epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;
    const uint32_t mine = *reinterpret_cast<const uint32_t *>(p.local_list);
    if (blockIdx.x == 0 && threadIdx.x == 0) // this rank's count, to the destination this workgroup row serves
        sys_store(&reinterpret_cast<ExchangeCtrl *>(p.ctrl[dst_rank])->count[e & 1u][p.rank], (unsigned long long)e << 32 | mine);
    if (threadIdx.x == 0) s_first = 0u, s_mine = mine, s_timeout = 0u;
    __syncthreads();
    // The counts of ALL ranks (their stores into THIS rank's block); those of the ranks before this one fix the offset.
    // Waiting for every rank's count is also what orders this exchange behind the peers' use of the previous one: a
    // rank stores its count at the start of ITS scatter, i.e. (stream order) after whatever it enqueued to read the
    // list of the previous exchange out of the buffer this scatter is about to write into — the guarantee the
    // count all-gather gives the collective form.
    if (threadIdx.x < p.world) {
        unsigned long long v = sys_load(&mine_ctrl->count[e & 1u][threadIdx.x]);
        for (uint32_t spins = 0; (uint32_t)(v >> 32) != e; spins++) {
            if (spins == kExchangeSpinLimit) {
                latch_status(p.status, ORBIT_E_TIMEOUT);
                s_timeout = 1u;
                break;
            }
            __builtin_amdgcn_s_sleep(8);
            v = sys_load(&mine_ctrl->count[e & 1u][threadIdx.x]);
        }
        if (threadIdx.x < p.rank) atomicAdd(&s_first, (uint32_t)v);
    }
    __syncthreads();
    const uint32_t first = s_first;
    const uint32_t room = first < p.out_capacity ? p.out_capacity - first : 0u;
    // a count that never came: the offset is made of a stale slot — nothing of this rank is copied anywhere (a list in
    // the wrong place would corrupt the peers' otherwise valid lists); the epoch still advances, ranks stay in step
    const uint32_t n_items = s_timeout != 0u ? 0u : min(mine, room);
    const uint64_t words = (uint64_t)n_items * p.stride / 4u;
    const uint32_t *src = reinterpret_cast<const uint32_t *>(p.local_list + p.header_bytes);
    uint32_t *dst = reinterpret_cast<uint32_t *>(p.out[dst_rank] + p.header_bytes + (uint64_t)first * p.stride);
    const uint64_t step = (uint64_t)gridDim.x * 256u * 4u;
    for (uint64_t w = ((uint64_t)blockIdx.x * 256u + (threadIdx.x & ~63u)) * 4u + (threadIdx.x & 63u); w < words;
         w += step) {
        uint32_t v[4];
#pragma unroll
        for (int k = 0; k < 4; k++) v[k] = w + 64u * k < words ? src[w + 64u * k] : 0u;
#pragma unroll
        for (int k = 0; k < 4; k++)
            if (w + 64u * k < words) dst[w + 64u * k] = v[k];
    }
    // Every store of this workgroup has left for the destination before it counts itself out; the last one out raises
    // done there.  Every wave drains its own stores, the workgroup meets, ONE lane makes the system-scope release (a
    // fence by all 256 threads of 512 workgroups is 512 cache write-backs queued behind each other: the first version
    // of this kernel spent 70 us of a 90-us exchange there).
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const uint32_t before = __hip_atomic_fetch_add(&mine_ctrl->left[dst_rank], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (before == gridDim.x - 1u) {
            __hip_atomic_store(&mine_ctrl->left[dst_rank], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            sys_store(&reinterpret_cast<ExchangeCtrl *>(p.ctrl[dst_rank])->done[p.rank], (unsigned long long)e);
        }
    }
}

__global__ __launch_bounds__(64) void exchange_wait_kernel(const ExchangeListParams p) {
    ExchangeCtrl *c = reinterpret_cast<ExchangeCtrl *>(p.ctrl[p.rank]);
    const uint32_t e = __hip_atomic_load(&c->epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;
    uint32_t count = 0;
    bool timed_out = false;
    if (threadIdx.x < p.world) {
        unsigned long long v = sys_load(&c->count[e & 1u][threadIdx.x]), d = sys_load(&c->done[threadIdx.x]);
        for (uint32_t spins = 0; (uint32_t)(v >> 32) != e || (uint32_t)d != e; spins++) {
            if (spins == kExchangeSpinLimit) {
                latch_status(p.status, ORBIT_E_TIMEOUT);
                timed_out = true;
                break;
            }
            __builtin_amdgcn_s_sleep(8);
            v = sys_load(&c->count[e & 1u][threadIdx.x]);
            d = sys_load(&c->done[threadIdx.x]);
        }
        count = (uint32_t)v;
    }
    // a rank that never arrived: the list is incomplete or misplaced — consumers get an EMPTY list (and the latched
    // ORBIT_E_TIMEOUT), never one with a hole in it
    const uint32_t total = __ballot(timed_out) != 0ull ? 0u : wave_reduce_add(count);
    if (threadIdx.x == 0) {
        uint32_t *hdr = reinterpret_cast<uint32_t *>(p.out[p.rank]);
        hdr[0] = min(total, p.out_capacity);
        for (uint32_t k = 1; k < p.header_bytes / 4u; k++) hdr[k] = 0u;
        if (total > p.out_capacity) latch_status(p.status, ORBIT_E_CAPACITY);
        __hip_atomic_store(&c->epoch, e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

} // namespace

hipError_t launch_exchange_list(const ExchangeListParams &p, uint32_t num_cus, hipStream_t s) {
    // enough workgroups per destination to keep a link busy, few enough that `world` copies share the device — and no
    // more than the list can feed: a workgroup per 32 KB of a rank's share of the gathered capacity (the list's own
    // length is on the device), every workgroup being a fence and two atomics of fixed cost
    const uint64_t share = (uint64_t)p.out_capacity * p.stride / max(p.world, 1u);
    const uint64_t want = (share + 32767u) / 32768u;
    const uint32_t by_bytes = want > (1u << 20) ? (1u << 20) : (uint32_t)want;
    const uint32_t per_dst = max(min(num_cus * 2u / max(p.world, 1u), by_bytes), 8u);
    hipLaunchKernelGGL(exchange_scatter_kernel, dim3(per_dst, p.world), dim3(256), 0, s, p);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(exchange_wait_kernel, dim3(1), dim3(64), 0, s, p);
    return hipGetLastError();
}

// mesh-shading path: expand the per-record ballots, no draw list
hipError_t launch_task_records(const MeshletCullParams &p, hipStream_t s) {
    const uint32_t tblocks = (p.dispatch_capacity + 255u) / 256u;
    if (tblocks) hipLaunchKernelGGL(task_records_kernel, dim3(tblocks), dim3(256), 0, s, p);
    return hipGetLastError();
}

hipError_t launch_compact_segments(const uint8_t *segments, uint32_t world, uint32_t segment_capacity, uint8_t *out,
                                   uint32_t out_capacity, uint32_t header_bytes, uint32_t stride, uint32_t num_cus,
                                   int32_t *status, hipStream_t s) {
    CompactSegmentsParams p{segments, out, world, segment_capacity, out_capacity, header_bytes, stride, status};
    // a workgroup per 32 KB of a segment, the device's worth at most
    const uint64_t want = ((uint64_t)segment_capacity * stride + 32767u) / 32768u;
    const uint32_t per_seg = (uint32_t)max((uint64_t)1, min(want, (uint64_t)max(num_cus * 4u / max(world, 1u), 8u)));
    hipLaunchKernelGGL(compact_segments_kernel, dim3(per_seg, world), dim3(256), 0, s, p);
    return hipGetLastError();
}

hipError_t launch_visible_records_expand(const uint8_t *records, uint32_t *block_pop, uint32_t max_blocks,
                                         const OrbitMeshlet *meshlets, const MeshletStreamView &ms, uint8_t *draw_buffer,
                                         uint32_t draw_capacity, const void *zero_page, int32_t *status, hipStream_t s) {
    // a command list of draw_capacity entries has at most that many records behind it
    const uint32_t blocks = max(min((draw_capacity + kVrBlock - 1u) / kVrBlock, max_blocks), 1u);
    hipLaunchKernelGGL(visible_records_pop_kernel, dim3(blocks), dim3(256), 0, s, records, block_pop, max_blocks);
    if (ms.cmd)
        hipLaunchKernelGGL(visible_records_expand_kernel<true>, dim3(blocks), dim3(256), 0, s, records, block_pop,
                           max_blocks, meshlets, ms, draw_buffer, draw_capacity, zero_page, status);
    else
        hipLaunchKernelGGL(visible_records_expand_kernel<false>, dim3(blocks), dim3(256), 0, s, records, block_pop,
                           max_blocks, meshlets, ms, draw_buffer, draw_capacity, zero_page, status);
    return hipGetLastError();
}

} // namespace orbit
