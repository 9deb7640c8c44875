// hs_exchange.hip - packing / unpacking of the generic row exchange between ranks (round 3).
//
// The reference routes every row of a shuffle through files, one per (partition, writer) (tasks.py:347-375), and the
// reader concatenates what it finds (tasks.py:144-150).  Here a rank's rows for all peers travel in ONE byte buffer
// (all_to_all_single): destination-major, and inside a destination's share one slice per column piece (fixed-width
// values; length bytes and payload bytes of a STRING column).  Both directions are the same operation - a list of
// (source, destination, bytes) segments copied by one launch:
//   pack    column pieces, already in destination order (stable counting sort by destination + gathers), are cut at
//           the destination boundaries and laid into the send buffer          world x pieces segments
//   unpack  the received buffer is cut at (source, piece) boundaries and every piece's slices are laid end to end
//           into that piece's output column                                    world x pieces segments
// Round 2 did both with Python loops over torch slices and torch.cat.  HBM-bound byte copies, nothing else.
#include "hs_device.h"

extern thread_local char g_hs_err[256];
void hs_set_error(const char* fmt, ...);

// One workgroup walks whole segments (blockIdx.y) in 16-byte steps where source and destination are both aligned,
// byte by byte on the ragged head / tail or when their alignments differ.
__global__ void __launch_bounds__(256) k_copy_segments(const hs_segment* segs, int32_t n_segs) {
    for (int sgi = blockIdx.y; sgi < n_segs; sgi += gridDim.y) {
        const hs_segment sg = segs[sgi];
        const uint8_t* src = (const uint8_t*)sg.src;
        uint8_t* dst = (uint8_t*)sg.dst;
        const int64_t n = sg.bytes;
        if (n <= 0) continue;
        const int64_t t0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, nt = (int64_t)gridDim.x * blockDim.x;
        if ((((uintptr_t)src ^ (uintptr_t)dst) & 15) == 0) {
            int64_t head = (16 - ((uintptr_t)dst & 15)) & 15;
            if (head > n) head = n;
            const int64_t body = (n - head) / 16;
            for (int64_t i = t0; i < head; i += nt) dst[i] = src[i];
            const uint4* s4 = reinterpret_cast<const uint4*>(src + head);
            uint4* d4 = reinterpret_cast<uint4*>(dst + head);
            for (int64_t i = t0; i < body; i += nt) d4[i] = s4[i];
            for (int64_t i = head + body * 16 + t0; i < n; i += nt) dst[i] = src[i];
        } else {
            for (int64_t i = t0; i < n; i += nt) dst[i] = src[i];
        }
    }
}

extern "C" int hs_copy_segments(void* stream, const hs_segment* segments_dev, int32_t n_segments, int64_t max_bytes) {
    if (n_segments < 0 || (n_segments > 0 && !segments_dev) || max_bytes < 0) {
        hs_set_error("hs_copy_segments: bad arguments");
        return HS_E_ARG;
    }
    if (n_segments == 0 || max_bytes == 0) return HS_OK;
    // enough workgroups for the largest segment to run near the HBM rate, not more than the chip has use for
    int64_t bx = (max_bytes / 16 + 255) / 256;
    bx = bx < 1 ? 1 : (bx > 2048 ? 2048 : bx);
    int by = n_segments > 1024 ? 1024 : n_segments;
    while (bx * by > 8192 && bx > 1) bx = (bx + 1) / 2;
    hipLaunchKernelGGL(k_copy_segments, dim3((unsigned)bx, (unsigned)by), dim3(256), 0, (hipStream_t)stream, segments_dev, n_segments);
    if (hipGetLastError() != hipSuccess) {
        hs_set_error("hs_copy_segments: kernel launch failed");
        return HS_E_LAUNCH;
    }
    return HS_OK;
}


// ---- peer-to-peer exchange of the partial-row slabs (prototype, round 3; HIPSPARK_P2P_SLABS=1) -----------------------
// The short tail's one exchange step is an all-gather of a few KB per rank.  As a collective it
// costs a library call, stream hand-overs and a ring over xGMI for what is, per rank, one small store to every peer.
// Here every rank owns ONE buffer that its peers map (hipIpc handles, exchanged once by the host side:
// minispark_amd/distributed.py PeerSlabs):
//     [parity 0 | parity 1] x [world slots of slot_bytes]  |  flags: [2][world] uint64
// hs_slab_push   (one workgroup) stores the rank's slab into slot `rank` of EVERY peer's buffer - over xGMI these are
//                plain stores -, makes them visible system-wide and then sets flags[parity][rank] = epoch in every peer;
// hs_slab_wait   (one workgroup) spins until all `world` flags of the current parity carry this run's epoch, then copies
//                the slots into the gathered layout hs_agg_finish reads.
// No host involvement, no collective.  Epochs are device-side counters (both launches are replayable with unchanged
// arguments); a rank can be at most one run ahead of a peer that has not finished reading (its next push needs the
// peer's push of that run, which the peer issues after its own finish launch), hence two parities.  The wait is bounded:
// after timeout_ms it raises HS_FLAG_PEER_TIMEOUT and returns, so a lost peer becomes an error, not a hung GPU.
struct SlabPushArgs {
    const uint8_t* slab;
    int64_t slab_bytes, slot_bytes;
    uint8_t* const* peers;  // [world] every rank's buffer as mapped here (own buffer at [rank])
    uint64_t* epoch;        // [0]: pushes so far
    int32_t world, rank;
};

__device__ __forceinline__ uint64_t* slab_flags(uint8_t* buf, int world, int64_t slot_bytes) {
    return (uint64_t*)(buf + 2 * (int64_t)world * slot_bytes);
}

__global__ void __launch_bounds__(1024) k_slab_push(const SlabPushArgs A) {
    const uint64_t e = A.epoch[0] + 1;
    const int parity = (int)(e & 1);
    const int64_t words = (A.slab_bytes + 15) / 16;  // slabs and slots are multiples of 16 bytes
    for (int p = 0; p < A.world; ++p) {
        uint4* dst = (uint4*)(A.peers[p] + ((int64_t)parity * A.world + A.rank) * A.slot_bytes);
        const uint4* src = (const uint4*)A.slab;
        for (int64_t i = threadIdx.x; i < words; i += blockDim.x) dst[i] = src[i];
    }
    __threadfence_system();
    __syncthreads();
    if ((int)threadIdx.x < A.world) {
        uint64_t* flags = slab_flags(A.peers[threadIdx.x], A.world, A.slot_bytes);
        __hip_atomic_store(&flags[parity * A.world + A.rank], e, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    if (threadIdx.x == 0) A.epoch[0] = e;
}

struct SlabWaitArgs {
    uint8_t* own;           // this rank's buffer
    int64_t slab_bytes, slot_bytes, out_stride;
    uint8_t* gathered;      // [world] x out_stride
    uint64_t* epoch;        // [1]: waits so far
    uint32_t* flags;
    int64_t timeout_ticks;  // wall_clock64 ticks (100 MHz)
    int32_t world, pad;
};

__global__ void __launch_bounds__(1024) k_slab_wait(const SlabWaitArgs A) {
    __shared__ int s_late;
    const uint64_t e = A.epoch[1] + 1;
    const int parity = (int)(e & 1);
    if (threadIdx.x == 0) s_late = 0;
    __syncthreads();
    if ((int)threadIdx.x < A.world) {
        uint64_t* flag = slab_flags(A.own, A.world, A.slot_bytes) + parity * A.world + threadIdx.x;
        const long long t0 = (long long)wall_clock64();
        while (__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < e) {
            if ((long long)wall_clock64() - t0 > A.timeout_ticks) {  // every wave reaches the exit: a lost peer is an error, not a hang
                s_late = 1;
                break;
            }
            __builtin_amdgcn_s_sleep(8);
        }
    }
    __syncthreads();
    if (s_late) {
        if (threadIdx.x == 0) atomicOr(A.flags, HS_FLAG_PEER_TIMEOUT);
    } else {
        const int64_t words = (A.slab_bytes + 15) / 16;
        for (int r = 0; r < A.world; ++r) {
            const uint4* src = (const uint4*)(A.own + ((int64_t)parity * A.world + r) * A.slot_bytes);
            uint4* dst = (uint4*)(A.gathered + (int64_t)r * A.out_stride);
            for (int64_t i = threadIdx.x; i < words; i += blockDim.x) dst[i] = src[i];
        }
    }
    if (threadIdx.x == 0) A.epoch[1] = e;
}

extern "C" size_t hs_slab_p2p_bytes(int32_t world, int64_t slot_bytes) {
    return world < 1 || slot_bytes < 16 ? 0 : (size_t)(2 * (int64_t)world * slot_bytes + 2 * (int64_t)world * 8);
}

extern "C" int hs_slab_push(void* stream, const void* slab, int64_t slab_bytes, void* const* peers_dev, int32_t world, int32_t rank,
                            int64_t slot_bytes, uint64_t* epochs_dev) {
    if (!slab || !peers_dev || !epochs_dev || world < 1 || world > 1024 || rank < 0 || rank >= world || slab_bytes < 16 ||
        slab_bytes > slot_bytes || (slab_bytes & 15) || (slot_bytes & 15) || ((uintptr_t)slab & 15)) {
        hs_set_error("hs_slab_push: bad arguments (slab %lld B, slot %lld B, world %d)", (long long)slab_bytes, (long long)slot_bytes, (int)world);
        return HS_E_ARG;
    }
    SlabPushArgs A{(const uint8_t*)slab, slab_bytes, slot_bytes, (uint8_t* const*)peers_dev, epochs_dev, world, rank};
    hipLaunchKernelGGL(k_slab_push, dim3(1), dim3(1024), 0, (hipStream_t)stream, A);
    if (hipGetLastError() != hipSuccess) {
        hs_set_error("hs_slab_push: kernel launch failed");
        return HS_E_LAUNCH;
    }
    return HS_OK;
}

extern "C" int hs_slab_wait(void* stream, void* own_buf, int32_t world, int64_t slot_bytes, int64_t slab_bytes, uint64_t* epochs_dev,
                            void* gathered, int64_t out_stride, uint32_t* flags, int64_t timeout_ms) {
    if (!own_buf || !epochs_dev || !gathered || !flags || world < 1 || world > 1024 || slab_bytes < 16 || slab_bytes > slot_bytes ||
        (slab_bytes & 15) || (slot_bytes & 15) || out_stride < slab_bytes || (out_stride & 15) || ((uintptr_t)gathered & 15) || timeout_ms < 1) {
        hs_set_error("hs_slab_wait: bad arguments");
        return HS_E_ARG;
    }
    SlabWaitArgs A{(uint8_t*)own_buf, slab_bytes, slot_bytes, out_stride, (uint8_t*)gathered, epochs_dev, flags, timeout_ms * 100000, world, 0};
    hipLaunchKernelGGL(k_slab_wait, dim3(1), dim3(1024), 0, (hipStream_t)stream, A);
    if (hipGetLastError() != hipSuccess) {
        hs_set_error("hs_slab_wait: kernel launch failed");
        return HS_E_LAUNCH;
    }
    return HS_OK;
}
