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
