// hs_ops.hip - the non-aggregate operators of libhipspark on gfx950:
//   device-wide exclusive scans (string offsets, compaction, join output offsets),
//   WHERE -> row-index compaction, gathers, string concat, expression evaluation, quantisation,
//   hash partitioning (stable counting sort), hash join (build: dictionary + CSR row lists; probe:
//   count -> scan -> fill), and the counter-based synthetic lineitem generator.
//
// Everything that reorders or selects rows produces a ROW INDEX LIST; columns are then materialised by
// gathers.  That keeps one implementation per reference loop (cited at each kernel).
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include "hs_agg_kernel.h"

thread_local char g_hs_err[256] = {0};

void hs_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_hs_err, sizeof(g_hs_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* hs_last_error(void) { return g_hs_err; }
extern "C" size_t hs_sizeof(int32_t which) {
    switch (which) {
        case 0: return sizeof(hs_col);
        case 1: return sizeof(hs_program);
        case 2: return sizeof(hs_agg_spec);
        case 3: return sizeof(hs_agg_geom);
        case 4: return sizeof(hs_chunk);
        case 5: return sizeof(hs_slab_desc);
        case 6: return sizeof(hs_finish_out);
        case 7: return sizeof(hs_finish_spec);
        case 8: return sizeof(hs_stage_plan);
        case 9: return sizeof(hs_result_col);
        case 10: return sizeof(hs_join8);
        case 11: return sizeof(hs_join_stage_plan);
        case 12: return sizeof(hs_select_stage_plan);
        default: return 0;
    }
}
extern "C" int hs_version(void) { return HS_VERSION; }

// ---- launch capture (hs_capture.h) ---------------------------------------------------------------------------
extern "C" int hs_capture_begin(void) {
    if (g_hs_capture) {
        hs_set_error("hs_capture_begin: a capture is already open on this thread");
        return HS_E_ARG;
    }
    g_hs_capture = new HsCapture();
    return HS_OK;
}
extern "C" int hs_capture_end(void** handle, int32_t* n_ops) {
    if (!g_hs_capture || !handle) {
        hs_set_error("hs_capture_end: no open capture");
        return HS_E_ARG;
    }
    HsCapture* cap = g_hs_capture;
    g_hs_capture = nullptr;
    if (n_ops) *n_ops = (int32_t)cap->ops.size();
    *handle = cap;
    return HS_OK;
}
extern "C" int hs_capture_replay(void* handle, void* stream) {
    if (!handle) {
        hs_set_error("hs_capture_replay: null handle");
        return HS_E_ARG;
    }
    hipStream_t s = (hipStream_t)stream;
    for (HsCapOp& op : ((HsCapture*)handle)->ops) {
        hipError_t rc = hipSuccess;
        switch (op.kind) {
            case HsCapOp::KERNEL: {
                HsTraceLaunch* tr = hs_trace_open(op.fn, nullptr, s);
                rc = hipLaunchKernel(op.fn, op.grid, op.block, op.argv.data(), op.lds, s);
                hs_trace_close(tr, s);
                break;
            }
            case HsCapOp::MODULE: {
                size_t sz = op.blob.size();
                void* extra[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, op.blob.data(), HIP_LAUNCH_PARAM_BUFFER_SIZE, &sz, HIP_LAUNCH_PARAM_END};
                HsTraceLaunch* tr = hs_trace_open(nullptr, op.mfn, s);
                rc = hipModuleLaunchKernel(op.mfn, op.grid.x, 1, 1, op.block.x, 1, 1, (unsigned)op.lds, s, nullptr, extra);
                hs_trace_close(tr, s);
                break;
            }
            case HsCapOp::EVENT: rc = hipEventRecord(op.ev, s); break;
            case HsCapOp::MEMSET: rc = hipMemsetAsync(op.ptr, op.value, op.bytes, s); break;
        }
        if (rc != hipSuccess) {
            hs_set_error("hs_capture_replay: %s", hipGetErrorString(rc));
            return HS_E_LAUNCH;
        }
    }
    return HS_OK;
}
extern "C" void hs_capture_free(void* handle) { delete (HsCapture*)handle; }

// ---- per-launch GPU slices (N4: the trace bridge) -------------------------------------------------------------------
extern "C" int hs_trace_begin(void* stream) {
    if (g_hs_trace) {
        hs_set_error("hs_trace_begin: a trace is already open on this thread");
        return HS_E_ARG;
    }
    HsTrace* t = new HsTrace();
    if (hipEventCreate(&t->base) != hipSuccess || hipEventRecord(t->base, (hipStream_t)stream) != hipSuccess) {
        delete t;
        hs_set_error("hs_trace_begin: cannot record the base event");
        return HS_E_LAUNCH;
    }
    g_hs_trace = t;
    return HS_OK;
}
extern "C" int hs_trace_end(void* stream, hs_trace_slice* out, int32_t cap, int32_t* n) {
    if (!g_hs_trace || !n || (cap > 0 && !out)) {
        hs_set_error("hs_trace_end: no open trace");
        return HS_E_ARG;
    }
    HsTrace* t = g_hs_trace;
    g_hs_trace = nullptr;
    (void)hipStreamSynchronize((hipStream_t)stream);
    int32_t k = 0;
    for (HsTraceLaunch& l : t->launches) {
        float t0 = 0.f, t1 = 0.f;
        const bool ok = hipEventElapsedTime(&t0, t->base, l.begin) == hipSuccess && hipEventElapsedTime(&t1, t->base, l.end) == hipSuccess;
        if (ok && k < cap) {
            const char* name = l.fn ? hipKernelNameRefByPtr(l.fn, (hipStream_t)stream) : hipKernelNameRef(l.mfn);
            strncpy(out[k].name, name ? name : "?", sizeof(out[k].name) - 1);
            out[k].name[sizeof(out[k].name) - 1] = 0;
            out[k].start_us = (double)t0 * 1e3;
            out[k].dur_us = (double)(t1 - t0) * 1e3;
            ++k;
        }
        (void)hipEventDestroy(l.begin);
        (void)hipEventDestroy(l.end);
    }
    (void)hipEventDestroy(t->base);
    *n = k;
    delete t;
    return HS_OK;
}

#define HS_CHECK_LAUNCH(name)                                                              \
    do {                                                                                   \
        const hipError_t e_ = hipGetLastError();                                           \
        if (e_ != hipSuccess) {                                                            \
            hs_set_error("%s: kernel launch failed (%s)", name, hipGetErrorString(e_));    \
            return HS_E_LAUNCH;                                                            \
        }                                                                                  \
    } while (0)

static constexpr int SCAN_WG = 256;
static constexpr int SCAN_ITEMS = 8;
static constexpr int SCAN_TILE = SCAN_WG * SCAN_ITEMS;

// ---- block-level helpers ---------------------------------------------------------------------------
__device__ __forceinline__ int64_t wave_incl_scan_i64(int64_t v) {
    const int lane = threadIdx.x & (HS_WAVE - 1);
#pragma unroll
    for (int d = 1; d < HS_WAVE; d <<= 1) {
        const int64_t t = __shfl_up(v, d, HS_WAVE);
        if (lane >= d) v += t;
    }
    return v;
}
// exclusive scan of one int64 per thread across a 256-thread workgroup; returns the exclusive prefix and the
// total: wave shuffles + one LDS hop (2 barriers).  s_tmp: >= 4 cells.
__device__ __forceinline__ int64_t block_excl_scan(int64_t v, int64_t* s_tmp, int64_t& total) {
    const int tid = threadIdx.x, w = tid / HS_WAVE;
    const int64_t incl = wave_incl_scan_i64(v);
    if ((tid & (HS_WAVE - 1)) == HS_WAVE - 1) s_tmp[w] = incl;
    __syncthreads();
    const int64_t p0 = s_tmp[0], p1 = s_tmp[1], p2 = s_tmp[2], p3 = s_tmp[3];
    __syncthreads();  // s_tmp may be reused by the caller
    total = p0 + p1 + p2 + p3;
    const int64_t base = w == 0 ? 0 : w == 1 ? p0 : w == 2 ? p0 + p1 : p0 + p1 + p2;
    return base + incl - v;
}

struct InLens {
    const uint8_t* p;
    __device__ __forceinline__ int64_t operator()(int64_t i) const { return p[i]; }
};
struct InMask {
    const uint8_t* p;
    __device__ __forceinline__ int64_t operator()(int64_t i) const { return p[i] != 0; }
};
struct InI64 {
    const int64_t* p;
    __device__ __forceinline__ int64_t operator()(int64_t i) const { return p[i]; }
};

template <typename In>
__global__ void __launch_bounds__(SCAN_WG) k_scan_reduce(In in, int64_t n, int64_t* tile_sums) {
    __shared__ int64_t s_tmp[SCAN_WG];
    // a tile's SUM does not care which lane adds which element: consecutive lanes take consecutive elements
    // (k_scan_down needs a lane's elements to be consecutive; this pass does not)
    const int64_t base = (int64_t)blockIdx.x * SCAN_TILE + (int64_t)threadIdx.x;
    int64_t sum = 0;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k)
        if (base + (int64_t)k * SCAN_WG < n) sum += in(base + (int64_t)k * SCAN_WG);
    int64_t total;
    block_excl_scan(sum, s_tmp, total);
    if (threadIdx.x == 0) tile_sums[blockIdx.x] = total;
}

// single workgroup: in-place exclusive scan of tile sums; tile_sums[ntiles] = grand total.  Round 3: 2048 sums per round
// through LDS (coalesced loads, eight consecutive sums per lane, one block scan) - one sum per lane and round meant 128
// rounds of four barriers for a 64 Mi-row input: 57 us, half of hs_compact's whole time.
__global__ void __launch_bounds__(SCAN_WG) k_scan_tiles(int64_t* tile_sums, int64_t ntiles) {
    constexpr int PER = 8;
    __shared__ int64_t s_buf[SCAN_WG * PER];
    __shared__ int64_t s_tmp[4];
    __shared__ int64_t s_carry;
    const int tid = threadIdx.x;
    if (tid == 0) s_carry = 0;
    __syncthreads();
    for (int64_t t0 = 0; t0 < ntiles; t0 += SCAN_WG * PER) {
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const int64_t t = t0 + k * SCAN_WG + tid;
            s_buf[k * SCAN_WG + tid] = t < ntiles ? tile_sums[t] : 0;
        }
        __syncthreads();
        int64_t v[PER], sum = 0;
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            v[k] = s_buf[tid * PER + k];
            sum += v[k];
        }
        int64_t total;
        int64_t run = s_carry + block_excl_scan(sum, s_tmp, total);
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            s_buf[tid * PER + k] = run;
            run += v[k];
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const int64_t t = t0 + k * SCAN_WG + tid;
            if (t < ntiles) tile_sums[t] = s_buf[k * SCAN_WG + tid];
        }
        __syncthreads();
        if (tid == 0) s_carry += total;
        __syncthreads();
    }
    if (tid == 0) tile_sums[ntiles] = s_carry;
}

struct EmitOffsets {  // out[i] = exclusive prefix; out[n] = total
    int64_t* out;
    __device__ __forceinline__ void operator()(int64_t i, int64_t excl, int64_t) const { out[i] = excl; }
};
struct EmitSelected {  // sel[prefix] = i for flagged rows
    int64_t* sel;
    __device__ __forceinline__ void operator()(int64_t i, int64_t excl, int64_t v) const {
        if (v) sel[excl] = i;
    }
};

template <typename In, typename Emit>
__global__ void __launch_bounds__(SCAN_WG) k_scan_down(In in, int64_t n, const int64_t* tile_sums, Emit emit,
                                                       int64_t* total_out) {
    __shared__ int64_t s_tmp[SCAN_WG];
    // a lane scans SCAN_ITEMS consecutive elements, but the tile is READ with consecutive lanes on consecutive
    // elements and handed over through LDS (one pad word per lane segment keeps the strided read-back at two lanes
    // per bank): inputs that gather from two arrays per element (the join's slot counts) ran at a quarter of the
    // single-array rate when every lane walked its own 64-byte line
    __shared__ int64_t s_vals[SCAN_TILE + SCAN_WG];
    const int64_t tile0 = (int64_t)blockIdx.x * SCAN_TILE;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) {
        const int e = k * SCAN_WG + (int)threadIdx.x;
        s_vals[e + e / SCAN_ITEMS] = tile0 + e < n ? in(tile0 + e) : 0;
    }
    __syncthreads();
    const int64_t base = tile0 + (int64_t)threadIdx.x * SCAN_ITEMS;
    int64_t vals[SCAN_ITEMS];
    int64_t sum = 0;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) {
        vals[k] = s_vals[(int)threadIdx.x * (SCAN_ITEMS + 1) + k];
        sum += vals[k];
    }
    int64_t total;
    int64_t run = tile_sums[blockIdx.x] + block_excl_scan(sum, s_tmp, total);
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) {
        if (base + k < n) emit(base + k, run, vals[k]);
        run += vals[k];
    }
    if (blockIdx.x == 0 && threadIdx.x == 0 && total_out) *total_out = tile_sums[gridDim.x];
}

// ---- byte inputs (string lengths, filter masks): 16 rows per lane from one 16-byte load -------------------------
// Tile = 256 lanes x 16 bytes = 4096 rows.  A lane's rows are consecutive, so its serial prefix needs no
// exchange; across lanes a wave shuffle scan; results go through LDS so that the stores to HBM are contiguous
// per wave.  Requires a 16-byte aligned input (the launchers fall back to the generic kernels otherwise).
static constexpr int VT_U8 = 16;
static constexpr int TILE_U8 = SCAN_WG * VT_U8;

__device__ __forceinline__ void load_bytes16(const uint8_t* p, int64_t base, int64_t n, uint32_t (&w)[4]) {
    if (base + VT_U8 <= n) {
        const uint4 v = *reinterpret_cast<const uint4*>(p + base);
        w[0] = v.x; w[1] = v.y; w[2] = v.z; w[3] = v.w;
    } else {
        w[0] = w[1] = w[2] = w[3] = 0;
        for (int k = 0; k < VT_U8; ++k)
            if (base + k < n) w[k >> 2] |= (uint32_t)p[base + k] << (8 * (k & 3));
    }
}
__device__ __forceinline__ uint32_t bytes_sum4(uint32_t w) {
    w = (w & 0x00ff00ffu) + ((w >> 8) & 0x00ff00ffu);
    return (w & 0xffffu) + (w >> 16);
}
__device__ __forceinline__ uint32_t bytes_nonzero4(uint32_t w) {  // 0x80 in every byte that is not zero
    return (((w & 0x7f7f7f7fu) + 0x7f7f7f7fu) | w) & 0x80808080u;
}

// MASK: count of non-zero bytes; otherwise the byte sum (+ min / max of the bytes for fixed-length detection)
template <bool MASK>
__global__ void __launch_bounds__(SCAN_WG) k_bytes_reduce(const uint8_t* p, int64_t n, int64_t* tile_sums,
                                                          int32_t* minmax) {
    __shared__ int64_t s_tmp[4];
    const int64_t base = (int64_t)blockIdx.x * TILE_U8 + (int64_t)threadIdx.x * VT_U8;
    uint32_t w[4];
    load_bytes16(p, base, n, w);
    int64_t sum = 0;
    if constexpr (MASK) {
        sum = __popc(bytes_nonzero4(w[0])) + __popc(bytes_nonzero4(w[1])) + __popc(bytes_nonzero4(w[2])) +
              __popc(bytes_nonzero4(w[3]));
    } else {
        sum = bytes_sum4(w[0]) + bytes_sum4(w[1]) + bytes_sum4(w[2]) + bytes_sum4(w[3]);
        if (minmax) {
            int mn = 256, mx = -1;
            if (base + VT_U8 <= n) {
                // a full lane: the sixteen bytes as eight pairs of 16-bit fields (even bytes, odd bytes of every word), min and
                // max per field with the packed 16-bit instructions, then across the two fields (round 3; the byte loop below
                // cost more than the sums it rides along with)
                typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
                u16x2 lo2 = {255, 255}, hi2 = {0, 0};
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const uint32_t even = w[q] & 0x00ff00ffu, odd = (w[q] >> 8) & 0x00ff00ffu;
                    const u16x2 e = __builtin_bit_cast(u16x2, even), o = __builtin_bit_cast(u16x2, odd);
                    lo2 = __builtin_elementwise_min(lo2, __builtin_elementwise_min(e, o));
                    hi2 = __builtin_elementwise_max(hi2, __builtin_elementwise_max(e, o));
                }
                mn = lo2.x < lo2.y ? lo2.x : lo2.y;
                mx = hi2.x > hi2.y ? hi2.x : hi2.y;
            } else {
                for (int k = 0; k < VT_U8; ++k) {
                    if (base + k < n) {
                        const int v = (w[k >> 2] >> (8 * (k & 3))) & 0xff;
                        mn = v < mn ? v : mn;
                        mx = v > mx ? v : mx;
                    }
                }
            }
            for (int d = HS_WAVE / 2; d >= 1; d >>= 1) {
                const int omn = __shfl_down(mn, d, HS_WAVE), omx = __shfl_down(mx, d, HS_WAVE);
                mn = omn < mn ? omn : mn;
                mx = omx > mx ? omx : mx;
            }
            // round 3: the tile's min / max go into the tile's own word behind the tile sums (the workspace is sized for
            // 2048-row tiles, these are 4096-row ones) and block 0 of the down-sweep folds them - the two words of `minmax`
            // used to be read by every wave of the grid: one hot L2 line, 0.13 of this kernel's 0.15 ms at 64 Mi rows
            __shared__ int s_mn[SCAN_WG / HS_WAVE], s_mx[SCAN_WG / HS_WAVE];
            if ((threadIdx.x % HS_WAVE) == 0) {
                s_mn[threadIdx.x / HS_WAVE] = mn;
                s_mx[threadIdx.x / HS_WAVE] = mx;
            }
            __syncthreads();
            if (threadIdx.x == 0) {
                for (int k = 1; k < SCAN_WG / HS_WAVE; ++k) {
                    mn = s_mn[k] < mn ? s_mn[k] : mn;
                    mx = s_mx[k] > mx ? s_mx[k] : mx;
                }
                tile_sums[gridDim.x + 1 + blockIdx.x] = (int64_t)(uint32_t)(mn & 0xffff) | ((int64_t)(uint32_t)(mx & 0xffff) << 16);
            }
        }
    }
    int64_t total;
    block_excl_scan(sum, s_tmp, total);
    if (threadIdx.x == 0) tile_sums[blockIdx.x] = total;
}

// offs[i] = exclusive prefix of the lengths; staged through LDS (padded: lane * 17 + k) for contiguous stores
__global__ void __launch_bounds__(SCAN_WG) k_lens_offsets(const uint8_t* lens, int64_t n, const int64_t* tile_sums,
                                                          int64_t* offs, int64_t* total_out, int32_t* minmax) {
    __shared__ int64_t s_tmp[4];
    if (minmax && blockIdx.x == 0) {  // min / max of all lengths from the tiles' words (k_bytes_reduce)
        __shared__ int s_lo[SCAN_WG / HS_WAVE], s_hi[SCAN_WG / HS_WAVE];
        int mn = 256, mx = -1;
        for (int64_t t = threadIdx.x; t < (int64_t)gridDim.x; t += SCAN_WG) {
            const int64_t word = tile_sums[gridDim.x + 1 + t];
            const int a = (int)(word & 0xffff), b = (int)((word >> 16) & 0xffff);
            if (b != 0xffff) {  // 0xffff: the tile held no row (cannot happen for n > 0; kept for symmetry with the byte loop)
                mn = a < mn ? a : mn;
                mx = b > mx ? b : mx;
            }
        }
        for (int d = HS_WAVE / 2; d >= 1; d >>= 1) {
            const int omn = __shfl_down(mn, d, HS_WAVE), omx = __shfl_down(mx, d, HS_WAVE);
            mn = omn < mn ? omn : mn;
            mx = omx > mx ? omx : mx;
        }
        if ((threadIdx.x % HS_WAVE) == 0) {
            s_lo[threadIdx.x / HS_WAVE] = mn;
            s_hi[threadIdx.x / HS_WAVE] = mx;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            for (int k = 1; k < SCAN_WG / HS_WAVE; ++k) {
                mn = s_lo[k] < mn ? s_lo[k] : mn;
                mx = s_hi[k] > mx ? s_hi[k] : mx;
            }
            minmax[0] = mn;
            minmax[1] = mx;
        }
    }
    __shared__ int64_t s_out[TILE_U8 + TILE_U8 / VT_U8];
    const int tid = threadIdx.x;
    const int64_t tile0 = (int64_t)blockIdx.x * TILE_U8;
    const int64_t base = tile0 + (int64_t)tid * VT_U8;
    uint32_t w[4];
    load_bytes16(lens, base, n, w);
    const int64_t sum = bytes_sum4(w[0]) + bytes_sum4(w[1]) + bytes_sum4(w[2]) + bytes_sum4(w[3]);
    int64_t total;
    int64_t run = tile_sums[blockIdx.x] + block_excl_scan(sum, s_tmp, total);
#pragma unroll
    for (int k = 0; k < VT_U8; ++k) {
        s_out[tid * (VT_U8 + 1) + k] = run;
        run += (w[k >> 2] >> (8 * (k & 3))) & 0xff;
    }
    __syncthreads();
    const int64_t left = n - tile0;
    const int count = left < TILE_U8 ? (int)left : TILE_U8;
    for (int i = tid; i < count; i += SCAN_WG) offs[tile0 + i] = s_out[i + i / VT_U8];
    if (blockIdx.x == 0 && tid == 0 && total_out) *total_out = tile_sums[gridDim.x];
}

// sel[prefix] = row for every row whose mask byte is set: wave ballot-free variant of the classic compaction -
// per-lane popcount, shuffle prefix across the wave, selected rows staged in LDS as 16-bit tile offsets, then
// contiguous 8-byte stores
__global__ void __launch_bounds__(SCAN_WG) k_mask_select(const uint8_t* mask, int64_t n, const int64_t* tile_sums,
                                                         int64_t* sel, int64_t* total_out) {
    __shared__ int64_t s_tmp[4];
    __shared__ uint16_t s_sel[TILE_U8];
    const int tid = threadIdx.x;
    const int64_t tile0 = (int64_t)blockIdx.x * TILE_U8;
    const int64_t base = tile0 + (int64_t)tid * VT_U8;
    uint32_t w[4];
    load_bytes16(mask, base, n, w);
    uint32_t bits = 0;  // bit k = row base + k is selected
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const uint32_t t = bytes_nonzero4(w[q]);
        bits |= (((t >> 7) & 1u) | ((t >> 14) & 2u) | ((t >> 21) & 4u) | ((t >> 28) & 8u)) << (4 * q);
    }
    int64_t total;
    int pos = (int)block_excl_scan((int64_t)__popc(bits), s_tmp, total);
    while (bits) {
        const int k = __ffs((int)bits) - 1;
        bits &= bits - 1;
        s_sel[pos++] = (uint16_t)(tid * VT_U8 + k);
    }
    __syncthreads();
    const int64_t out0 = tile_sums[blockIdx.x];
    for (int i = tid; i < (int)total; i += SCAN_WG) sel[out0 + i] = tile0 + s_sel[i];
    if (blockIdx.x == 0 && tid == 0 && total_out) *total_out = tile_sums[gridDim.x];
}

extern "C" size_t hs_scan_ws_bytes(int64_t nrows) {
    const int64_t ntiles = (nrows + SCAN_TILE - 1) / SCAN_TILE;  // the smallest tile any scan uses
    return (size_t)(ntiles + 2) * 8;
}

template <typename In, typename Emit>
static int run_scan(hipStream_t s, In in, int64_t n, Emit emit, int64_t* total_out, void* ws, const char* name) {
    const int64_t ntiles = (n + SCAN_TILE - 1) / SCAN_TILE;
    int64_t* tiles = (int64_t*)ws;
    if (ntiles == 0) {
        // empty input: total = 0.  (Until round 4 this launched the two kernels over zero tiles, and the down-sweep read
        // its total from tile_sums[gridDim.x] = word 1 of a workspace only word 0 of which had been written: whatever the
        // allocator had left there came back as the row count - zero in a fresh process, garbage in a long-lived one.)
        if (total_out) hs_memset_async(total_out, 0, sizeof(int64_t), s);
        HS_CHECK_LAUNCH(name);
        return HS_OK;
    }
    if (ntiles > 0x7fffffffll) {
        hs_set_error("%s: input too large", name);
        return HS_E_LIMIT;
    }
    hipLaunchKernelGGL((k_scan_reduce<In>), dim3((unsigned)ntiles), dim3(SCAN_WG), 0, s, in, n, tiles);
    hipLaunchKernelGGL(k_scan_tiles, dim3(1), dim3(SCAN_WG), 0, s, tiles, ntiles);
    hipLaunchKernelGGL((k_scan_down<In, Emit>), dim3((unsigned)ntiles), dim3(SCAN_WG), 0, s, in, n, tiles, emit,
                       total_out);
    HS_CHECK_LAUNCH(name);
    return HS_OK;
}

// ---- A1: string offsets + fixed-length detection (reference io.py:143-149) -------------------------------
__global__ void k_minmax_init(int32_t* minmax) {
    minmax[0] = 256;
    minmax[1] = -1;
}
__global__ void __launch_bounds__(256) k_lens_minmax(const uint8_t* lens, int64_t n, int32_t* minmax) {
    int mn = 256, mx = -1;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int v = lens[i];
        mn = v < mn ? v : mn;
        mx = v > mx ? v : mx;
    }
    for (int d = HS_WAVE / 2; d >= 1; d >>= 1) {
        const int omn = __shfl_down(mn, d, HS_WAVE), omx = __shfl_down(mx, d, HS_WAVE);
        mn = omn < mn ? omn : mn;
        mx = omx > mx ? omx : mx;
    }
    if ((threadIdx.x % HS_WAVE) == 0) {
        atomicMin(&minmax[0], mn);
        atomicMax(&minmax[1], mx);
    }
}

static bool aligned16(const void* p) { return ((uintptr_t)p & 15) == 0; }

// reduce -> scan of the tile sums -> down-sweep, byte-input form.
// Round 4 tried ONE pass instead (decoupled look-back: 16384-row tiles taken in ticket order, a tile's sum and then its
// inclusive prefix published in one 64-bit word per tile, wave 0 looking back over 64 predecessors at a time) and measured
// it at 64 Mi rows on MI355X: with release / acquire at agent scope 0.53 ms (compaction) and 0.56 ms (offsets) - every
// publish wrote back, every poll invalidated the XCD's L2; with relaxed agent-scope words (the word is the whole message)
// and 1024 persistent workgroups 0.114 / 0.169 ms - against 0.099 / 0.164 ms for these three launches.  A poll of a word
// another XCD wrote goes through the memory-side cache (~1-2 us) and the four workgroups of a CU all stall in it; what
// the pass saves - the second read of 1 byte per row - is a fifth of the 8 bytes per kept row it has to write anyway.
// Not kept (profiles/r04_single_pass_scans_64M.txt).
template <bool MASK>
static int run_bytes_scan(hipStream_t s, const uint8_t* p, int64_t n, int64_t* out, int64_t* total_out, int32_t* minmax,
                          void* ws, const char* name) {
    const int64_t ntiles = (n + TILE_U8 - 1) / TILE_U8;
    int64_t* tiles = (int64_t*)ws;
    if (ntiles > 0x7fffffffll) {
        hs_set_error("%s: input too large", name);
        return HS_E_LIMIT;
    }
    hipLaunchKernelGGL((k_bytes_reduce<MASK>), dim3((unsigned)ntiles), dim3(SCAN_WG), 0, s, p, n, tiles, minmax);
    hipLaunchKernelGGL(k_scan_tiles, dim3(1), dim3(SCAN_WG), 0, s, tiles, ntiles);
    if constexpr (MASK)
        hipLaunchKernelGGL(k_mask_select, dim3((unsigned)ntiles), dim3(SCAN_WG), 0, s, p, n, tiles, out, total_out);
    else
        hipLaunchKernelGGL(k_lens_offsets, dim3((unsigned)ntiles), dim3(SCAN_WG), 0, s, p, n, tiles, out, total_out, minmax);
    HS_CHECK_LAUNCH(name);
    return HS_OK;
}

extern "C" int hs_str_offsets(void* stream, const uint8_t* lens, int64_t nrows, int64_t* offs, int32_t* minmax,
                              void* ws) {
    if ((!lens && nrows > 0) || !offs || !ws || nrows < 0) {
        hs_set_error("hs_str_offsets: bad arguments");
        return HS_E_ARG;
    }
    hipStream_t s = (hipStream_t)stream;
    if (minmax) hipLaunchKernelGGL(k_minmax_init, dim3(1), dim3(1), 0, s, minmax);
    if (nrows > 0 && aligned16(lens))  // min / max ride along with the reduce pass
        return run_bytes_scan<false>(s, lens, nrows, offs, offs + nrows, minmax, ws, "hs_str_offsets");
    if (minmax && nrows > 0) {
        int64_t blocks = (nrows + 256 * 16 - 1) / (256 * 16);
        if (blocks > 2048) blocks = 2048;
        hipLaunchKernelGGL(k_lens_minmax, dim3((unsigned)blocks), dim3(256), 0, s, lens, nrows, minmax);
    }
    return run_scan(s, InLens{lens}, nrows, EmitOffsets{offs}, offs + nrows, ws, "hs_str_offsets");
}

extern "C" int hs_exclusive_scan_i64(void* stream, const int64_t* counts, int64_t n, int64_t* start, void* ws) {
    if ((!counts && n > 0) || !start || !ws || n < 0) {
        hs_set_error("hs_exclusive_scan_i64: bad arguments");
        return HS_E_ARG;
    }
    return run_scan((hipStream_t)stream, InI64{counts}, n, EmitOffsets{start}, start + n, ws, "hs_exclusive_scan_i64");
}

// ---- A3: filter = stable compaction to a row index list (reference tasks.py:177) -----------------------
extern "C" int hs_compact(void* stream, const uint8_t* mask, int64_t nrows, int64_t* sel, int64_t* count, void* ws) {
    if ((!mask && nrows > 0) || (!sel && nrows > 0) || !count || !ws || nrows < 0) {
        hs_set_error("hs_compact: bad arguments");
        return HS_E_ARG;
    }
    if (nrows > 0 && aligned16(mask))
        return run_bytes_scan<true>((hipStream_t)stream, mask, nrows, sel, count, nullptr, ws, "hs_compact");
    return run_scan((hipStream_t)stream, InMask{mask}, nrows, EmitSelected{sel}, count, ws, "hs_compact");
}

// ---- gathers -------------------------------------------------------------------------------------
__device__ __forceinline__ int64_t capped(int64_t n, const int64_t* n_dev) {
    if (!n_dev) return n;
    const int64_t d = *n_dev;
    return d < n ? d : n;
}

// Row indices come out of other kernels (compaction, dictionaries, joins): an index outside the source is a bug
// upstream, and an unchecked read of it can fault the whole GPU - so it is checked: the row reads as zero and
// HS_FLAG_BAD_PROGRAM is raised (one compare per element, nothing next to the random read it guards).
template <typename T>
__global__ void __launch_bounds__(256) k_gather_fixed(const T* src, int64_t src_rows, const int64_t* idx, int64_t n,
                                                      const int64_t* n_dev, T* dst, uint32_t* flags) {
    n = capped(n, n_dev);
    bool bad = false;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = idx[i];
        if (r < 0 || r >= src_rows) {
            bad = true;
            dst[i] = T(0);
        } else {
            dst[i] = src[r];
        }
    }
    if (bad) atomicOr(flags, HS_FLAG_BAD_PROGRAM);
}

static unsigned grid_for(int64_t n, int per_block) {
    int64_t b = (n + per_block - 1) / per_block;
    if (b < 1) b = 1;
    if (b > 65536) b = 65536;
    return (unsigned)b;
}

extern "C" int hs_gather_fixed(void* stream, const void* src, int32_t elem_bytes, int64_t src_rows, const int64_t* idx,
                               int64_t n, const int64_t* n_dev, void* dst, uint32_t* flags) {
    if (n == 0) return HS_OK;
    if (!src || !idx || !dst || !flags || n < 0 || src_rows < 0) {
        hs_set_error("hs_gather_fixed: bad arguments");
        return HS_E_ARG;
    }
    hipStream_t s = (hipStream_t)stream;
    const dim3 g(grid_for(n, 256)), b(256);
    switch (elem_bytes) {
        case 1: hipLaunchKernelGGL(k_gather_fixed<uint8_t>, g, b, 0, s, (const uint8_t*)src, src_rows, idx, n, n_dev, (uint8_t*)dst, flags); break;
        case 2: hipLaunchKernelGGL(k_gather_fixed<uint16_t>, g, b, 0, s, (const uint16_t*)src, src_rows, idx, n, n_dev, (uint16_t*)dst, flags); break;
        case 4: hipLaunchKernelGGL(k_gather_fixed<uint32_t>, g, b, 0, s, (const uint32_t*)src, src_rows, idx, n, n_dev, (uint32_t*)dst, flags); break;
        case 8: hipLaunchKernelGGL(k_gather_fixed<uint64_t>, g, b, 0, s, (const uint64_t*)src, src_rows, idx, n, n_dev, (uint64_t*)dst, flags); break;
        default: hs_set_error("hs_gather_fixed: elem_bytes=%d", elem_bytes); return HS_E_ARG;
    }
    HS_CHECK_LAUNCH("hs_gather_fixed");
    return HS_OK;
}

// (row indices are checked like in k_gather_fixed: one outside [0, src_rows) gathers the empty string and raises
// HS_FLAG_BAD_PROGRAM)
__global__ void __launch_bounds__(256) k_gather_str_lens(const hs_col src, int64_t src_rows, const int64_t* idx, int64_t n,
                                                         uint8_t* out_lens, uint32_t* flags) {
    bool bad = false;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = idx ? idx[i] : i;
        if (r < 0 || r >= src_rows) {
            bad = true;
            out_lens[i] = 0;
        } else {
            out_lens[i] = src.fixed_len >= 0 ? (uint8_t)src.fixed_len : src.lens[r];
        }
    }
    if (bad) atomicOr(flags, HS_FLAG_BAD_PROGRAM);
}
__global__ void __launch_bounds__(256) k_gather_str_bytes(const hs_col src, int64_t src_rows, const int64_t* idx, int64_t n,
                                                          const int64_t* out_offs, uint8_t* out_data) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = idx ? idx[i] : i;
        if (r < 0 || r >= src_rows) continue;  // flagged (and given length 0) by k_gather_str_lens
        const HsStr s = hs_str_at(src, r);
        uint8_t* d = out_data + out_offs[i];
        if (s.len <= 16) {
            // the usual case: at most three aligned 8-byte loads instead of up to 16 byte loads (hs_str_words16)
            uint64_t w0, w1;
            hs_str_words16(s.p, s.len, w0, w1);
#pragma unroll
            for (uint32_t k = 0; k < 8; ++k)
                if (k < s.len) d[k] = (uint8_t)(w0 >> (8 * k));
#pragma unroll
            for (uint32_t k = 0; k < 8; ++k)
                if (k + 8 < s.len) d[k + 8] = (uint8_t)(w1 >> (8 * k));
        } else {
            for (uint32_t k = 0; k < s.len; ++k) d[k] = s.p[k];
        }
    }
}

extern "C" int hs_gather_str_lens(void* stream, const hs_col* src, int64_t src_rows, const int64_t* idx, int64_t n,
                                  uint8_t* out_lens, uint32_t* flags) {
    if (n == 0) return HS_OK;
    if (!src || src->kind != HS_STR || !out_lens || !flags || n < 0 || src_rows < 0) {
        hs_set_error("hs_gather_str_lens: bad arguments");
        return HS_E_ARG;
    }
    hipLaunchKernelGGL(k_gather_str_lens, dim3(grid_for(n, 256)), dim3(256), 0, (hipStream_t)stream, *src, src_rows, idx, n,
                       out_lens, flags);
    HS_CHECK_LAUNCH("hs_gather_str_lens");
    return HS_OK;
}
extern "C" int hs_gather_str_bytes(void* stream, const hs_col* src, int64_t src_rows, const int64_t* idx, int64_t n,
                                   const int64_t* out_offs, uint8_t* out_data) {
    if (n == 0) return HS_OK;
    if (!src || src->kind != HS_STR || !out_offs || n < 0 || src_rows < 0) {  // out_data may be NULL when every string is empty
        hs_set_error("hs_gather_str_bytes: bad arguments");
        return HS_E_ARG;
    }
    hipLaunchKernelGGL(k_gather_str_bytes, dim3(grid_for(n, 256)), dim3(256), 0, (hipStream_t)stream, *src, src_rows, idx, n,
                       out_offs, out_data);
    HS_CHECK_LAUNCH("hs_gather_str_bytes");
    return HS_OK;
}

// ---- string '+' (reference sql.py:262-266 operator.add on str; zig utils.zig:118-131) ------------------
#define HS_MAX_PARTS 8
struct ConcatArgs {
    int32_t n_parts;
    int32_t pad;
    hs_col parts[HS_MAX_PARTS];
};
__device__ __forceinline__ HsStr part_str(const hs_col& p, int64_t row) {
    if (p.kind == HS_STR) return hs_str_at(p, row);
    HsStr s;  // literal: data = device bytes, fixed_len = length
    s.p = (const uint8_t*)p.data;
    s.len = (uint32_t)p.fixed_len;
    return s;
}
__global__ void __launch_bounds__(256) k_concat_lens(const ConcatArgs A_kernarg, int64_t n, uint8_t* out_lens, uint32_t* flags) {
    HS_KERNARG(ConcatArgs, A);
    uint32_t err = 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        uint32_t len = 0;
        for (int p = 0; p < A.n_parts; ++p) len += part_str(A.parts[p], i).len;
        if (len > 255) {
            err |= HS_FLAG_STR_TOO_LONG;
            len = 255;
        }
        out_lens[i] = (uint8_t)len;
    }
    if (err && flags) atomicOr(flags, err);
}
__global__ void __launch_bounds__(256) k_concat_bytes(const ConcatArgs A_kernarg, int64_t n, const int64_t* out_offs,
                                                      uint8_t* out_data) {
    HS_KERNARG(ConcatArgs, A);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        uint8_t* d = out_data + out_offs[i];
        const int64_t room = out_offs[i + 1] - out_offs[i];
        int64_t w = 0;
        for (int p = 0; p < A.n_parts; ++p) {
            const HsStr s = part_str(A.parts[p], i);
            for (uint32_t k = 0; k < s.len && w < room; ++k) d[w++] = s.p[k];
        }
    }
}
static int fill_concat(ConcatArgs& A, const hs_col* parts, int32_t n_parts, const char* name) {
    if (!parts || n_parts < 1 || n_parts > HS_MAX_PARTS) {
        hs_set_error("%s: n_parts=%d out of range 1..%d", name, n_parts, HS_MAX_PARTS);
        return HS_E_LIMIT;
    }
    A.n_parts = n_parts;
    A.pad = 0;
    for (int i = 0; i < n_parts; ++i) A.parts[i] = parts[i];
    return HS_OK;
}
extern "C" int hs_concat_lens(void* stream, const hs_col* parts, int32_t n_parts, int64_t nrows, uint8_t* out_lens,
                              uint32_t* flags) {
    ConcatArgs A;
    int rc = fill_concat(A, parts, n_parts, "hs_concat_lens");
    if (rc) return rc;
    if (nrows == 0) return HS_OK;
    hipLaunchKernelGGL(k_concat_lens, dim3(grid_for(nrows, 256)), dim3(256), 0, (hipStream_t)stream, A, nrows, out_lens,
                       flags);
    HS_CHECK_LAUNCH("hs_concat_lens");
    return HS_OK;
}
extern "C" int hs_concat_bytes(void* stream, const hs_col* parts, int32_t n_parts, int64_t nrows,
                               const int64_t* out_offs, uint8_t* out_data) {
    ConcatArgs A;
    int rc = fill_concat(A, parts, n_parts, "hs_concat_bytes");
    if (rc) return rc;
    if (nrows == 0) return HS_OK;
    hipLaunchKernelGGL(k_concat_bytes, dim3(grid_for(nrows, 256)), dim3(256), 0, (hipStream_t)stream, A, nrows,
                       out_offs, out_data);
    HS_CHECK_LAUNCH("hs_concat_bytes");
    return HS_OK;
}

// ---- A4: expression evaluation, one row per lane (reference tasks.py:32-35, sql.py:262-266) ----------------
// hs_jit.hip: HS_OK = a kernel compiled for this program was launched
int hs_jit_launch_eval(const EvalArgs* args, unsigned grid, hipStream_t stream);

struct EvalSink {
    const EvalArgs& A;
    int64_t in_row, out_row;
    __device__ __forceinline__ EvalSink(const EvalArgs& a) : A(a) {}
    __device__ __forceinline__ uint64_t load(uint32_t s, int) const { return hs_load_cell(A.cols.c[s], in_row); }
    __device__ __forceinline__ bool live(int) const { return true; }
    __device__ __forceinline__ int64_t row(int) const { return in_row; }
    __device__ __forceinline__ void filter(int, bool) {}
    __device__ __forceinline__ void agg(uint32_t, int, uint64_t) {}
    __device__ __forceinline__ void key() {}
    __device__ __forceinline__ void out(uint32_t o, int, uint64_t cell) {
        switch (A.out_kinds[o]) {
            case HS_U8: ((uint8_t*)A.outs[o])[out_row] = cell != 0; break;
            default: ((uint64_t*)A.outs[o])[out_row] = cell; break;
        }
    }
};
__global__ void __launch_bounds__(256) k_eval(const EvalArgs A_kernarg) {
    HS_KERNARG(EvalArgs, A);
    uint32_t err = 0;
    EvalSink sink(A);
    const int64_t nrows = capped(A.nrows, A.nrows_dev);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nrows; i += (int64_t)gridDim.x * blockDim.x) {
        sink.out_row = i;
        sink.in_row = A.sel ? A.sel[i] : i;
        hs_run<HS_MAX_STACK, 1>(A.prog, A.cols, 0, A.prog.n_ins, sink, err);
    }
    if (err) atomicOr(A.flags, err);
}
extern "C" int hs_eval(void* stream, const hs_col* cols, int32_t n_cols, const hs_program* prog, const int64_t* sel,
                       int64_t nrows, const int64_t* nrows_dev, void* const* outs, const int32_t* out_kinds,
                       int32_t n_outs, uint32_t* flags) {
    if (!prog || !flags || (n_cols > 0 && !cols) || n_cols < 0 || n_cols > HS_MAX_COLS || n_outs < 0 ||
        n_outs > HS_MAX_OUTS || (n_outs > 0 && (!outs || !out_kinds)) || nrows < 0) {
        hs_set_error("hs_eval: bad arguments");
        return HS_E_ARG;
    }
    if (prog->n_ins > HS_MAX_INS || prog->n_lit > HS_MAX_LIT) {
        hs_set_error("hs_eval: program too long");
        return HS_E_LIMIT;
    }
    if (nrows == 0) return HS_OK;
    EvalArgs A;
    A.cols.n = n_cols;
    A.cols.pad = 0;
    for (int i = 0; i < HS_MAX_COLS; ++i) A.cols.c[i] = i < n_cols ? cols[i] : hs_col{HS_U8, -1, nullptr, nullptr, nullptr};
    A.prog = *prog;
    A.sel = sel;
    A.nrows = nrows;
    A.nrows_dev = nrows_dev;
    for (int i = 0; i < HS_MAX_OUTS; ++i) {
        A.outs[i] = i < n_outs ? outs[i] : nullptr;
        A.out_kinds[i] = i < n_outs ? out_kinds[i] : HS_F64;
    }
    A.flags = flags;
    // compiled form: four rows per lane with 16-byte loads / stores; needs whole-column inputs (no row list)
    // and 16-byte aligned numeric buffers.  Otherwise - or when hiprtc is not there - the interpreter kernel.
    bool vector_ok = sel == nullptr;
    for (int i = 0; i < n_cols && vector_ok; ++i)
        if (cols[i].kind != HS_STR && ((uintptr_t)cols[i].data & 15)) vector_ok = false;
    for (int i = 0; i < n_outs && vector_ok; ++i)
        if ((uintptr_t)outs[i] & 15) vector_ok = false;
    if (vector_ok && hs_jit_launch_eval(&A, grid_for((nrows + 3) / 4, 256), (hipStream_t)stream) == HS_OK) return HS_OK;
    hipLaunchKernelGGL(k_eval, dim3(grid_for(nrows, 256)), dim3(256), 0, (hipStream_t)stream, A);
    HS_CHECK_LAUNCH("hs_eval");
    return HS_OK;
}

// ---- quantisation at a file write (reference io.py:87-94) ---------------------------------------------
__global__ void __launch_bounds__(256) k_quantise(const void* src, int32_t kind, int64_t n, const int64_t* n_dev,
                                                  void* dst, uint32_t* flags) {
    uint32_t err = 0;
    n = capped(n, n_dev);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        if (kind == HS_F64) {
            const double d = ((const double*)src)[i];
            const float f = (float)d;
            if (isinf(f) && !isinf(d)) err |= HS_FLAG_FLT_OVERFLOW;
            ((float*)dst)[i] = f;
        } else {
            const int64_t v = ((const int64_t*)src)[i];
            if (v > 2147483647ll || v < -2147483648ll) err |= HS_FLAG_INT_OVERFLOW;
            ((int32_t*)dst)[i] = (int32_t)v;
        }
    }
    if (err) atomicOr(flags, err);
}
extern "C" int hs_quantise(void* stream, const void* src, int32_t src_kind, int64_t n, const int64_t* n_dev, void* dst,
                           uint32_t* flags) {
    if (n == 0) return HS_OK;
    if (!src || !dst || !flags || n < 0 || (src_kind != HS_F64 && src_kind != HS_I64)) {
        hs_set_error("hs_quantise: bad arguments");
        return HS_E_ARG;
    }
    hipLaunchKernelGGL(k_quantise, dim3(grid_for(n, 256)), dim3(256), 0, (hipStream_t)stream, src, src_kind, n, n_dev,
                       dst, flags);
    HS_CHECK_LAUNCH("hs_quantise");
    return HS_OK;
}

struct QuantManyArgs {
    int32_t n_cols;
    int32_t kinds[16];
    const void* srcs[16];
    void* dsts[16];
};
__global__ void __launch_bounds__(256) k_quantise_many(const QuantManyArgs A_kernarg, int64_t n, const int64_t* n_dev,
                                                       uint32_t* flags) {
    HS_KERNARG(QuantManyArgs, A);
    uint32_t err = 0;
    n = capped(n, n_dev);
    const int64_t total = n * A.n_cols;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(t / n);
        const int64_t i = t % n;
        if (A.kinds[c] == HS_F64) {
            const double d = ((const double*)A.srcs[c])[i];
            const float f = (float)d;
            if (isinf(f) && !isinf(d)) err |= HS_FLAG_FLT_OVERFLOW;
            ((float*)A.dsts[c])[i] = f;
        } else {
            const int64_t v = ((const int64_t*)A.srcs[c])[i];
            if (v > 2147483647ll || v < -2147483648ll) err |= HS_FLAG_INT_OVERFLOW;
            ((int32_t*)A.dsts[c])[i] = (int32_t)v;
        }
    }
    if (err) atomicOr(flags, err);
}
extern "C" int hs_quantise_many(void* stream, int32_t n_cols, void* const* srcs, const int32_t* src_kinds, int64_t n,
                                const int64_t* n_dev, void* const* dsts, uint32_t* flags) {
    if (n == 0 || n_cols == 0) return HS_OK;
    if (n_cols < 0 || n_cols > 16 || !srcs || !src_kinds || !dsts || !flags || n < 0) {
        hs_set_error("hs_quantise_many: bad arguments");
        return HS_E_ARG;
    }
    QuantManyArgs A;
    A.n_cols = n_cols;
    for (int i = 0; i < 16; ++i) {
        A.kinds[i] = i < n_cols ? src_kinds[i] : HS_I64;
        A.srcs[i] = i < n_cols ? srcs[i] : nullptr;
        A.dsts[i] = i < n_cols ? dsts[i] : nullptr;
        if (i < n_cols && A.kinds[i] != HS_F64 && A.kinds[i] != HS_I64) {
            hs_set_error("hs_quantise_many: column %d is not an in-flight kind", i);
            return HS_E_ARG;
        }
    }
    hipLaunchKernelGGL(k_quantise_many, dim3(grid_for(n * n_cols, 256)), dim3(256), 0, (hipStream_t)stream, A, n, n_dev,
                       flags);
    HS_CHECK_LAUNCH("hs_quantise_many");
    return HS_OK;
}

// ---- A6/A9: hash partitioning (reference tasks.py:353-365) ------------------------------------------------
__device__ __forceinline__ uint32_t py_int_partition(int64_t v, int32_t n_parts) {
    // hash(int) == int for |int| < 2**61-1, except hash(-1) == -2; then Python's floor-mod
    if (v == -1) v = -2;
    int64_t m = v % n_parts;
    if (m < 0) m += n_parts;
    return (uint32_t)m;
}
// CPython's hash of a finite float (Python/pyhash.c _Py_HashDouble; not randomised, unlike str): the mantissa in 28-bit
// chunks modulo the Mersenne prime 2^61 - 1, then a rotation by the exponent, the sign, and -1 -> -2.  The reference routes
// a FLOAT key to shuffle partition hash(key) % P (tasks.py:362), so which JoinJob sums a row - and with it the f32 rounding
// of that job's partial sums - depends on this function.  +-inf hash to +-314159; NaN (hashed by object identity since
// Python 3.10) gets 0 here.
__device__ __forceinline__ int64_t py_float_hash(double v) {
    constexpr uint64_t MOD = (1ull << 61) - 1;
    if (isinf(v)) return v > 0 ? 314159 : -314159;
    if (isnan(v)) return 0;
    int e;
    double m = frexp(v, &e);
    int sign = 1;
    if (m < 0) {
        sign = -1;
        m = -m;
    }
    uint64_t x = 0;
    while (m != 0.0) {
        x = ((x << 28) & MOD) | (x >> (61 - 28));
        m *= 268435456.0;  // 2^28
        e -= 28;
        const uint64_t y = (uint64_t)m;  // the integer part
        m -= (double)y;
        x += y;
        if (x >= MOD) x -= MOD;
    }
    e = e >= 0 ? e % 61 : 61 - 1 - ((-1 - e) % 61);
    x = ((x << e) & MOD) | (x >> (61 - e));
    int64_t h = sign * (int64_t)x;
    return h == -1 ? -2 : h;
}
__device__ __forceinline__ uint32_t py_hash_partition(int64_t h, int32_t n_parts) {  // Python's floor-mod
    int64_t r = h % (int64_t)n_parts;
    if (r < 0) r += n_parts;
    return (uint32_t)r;
}

__global__ void __launch_bounds__(256) k_partition_ids(const hs_col key, const int64_t* sel, int64_t n, int32_t n_parts,
                                                       uint8_t* part) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = sel ? sel[i] : i;
        uint32_t p;
        if (key.kind == HS_STR) {
            const HsStr s = hs_str_at(key, r);
            p = (uint32_t)(hs_fnv1a(s.p, s.len) % (uint64_t)n_parts);
        } else if (key.kind == HS_I32 || key.kind == HS_I64) {
            p = py_int_partition((int64_t)hs_load_cell(key, r), n_parts);
        } else {
            const double d = hs_u2d(hs_load_cell(key, r));
            p = py_hash_partition(py_float_hash(d), n_parts);  // (integral floats hash like the equal int: same function)
        }
        part[i] = (uint8_t)p;
    }
}
// INTEGER keys, every row (the shuffle of a table column): four keys per lane in one 16-byte load, the partition
// without an integer division (hs_py_partition_inv), four ids in one 4-byte store - 5 B/row at the HBM rate instead of
// 18 % of it (one key per lane, a 64-bit % per row, byte stores).
__global__ void __launch_bounds__(256) k_partition_ids_i32(const int32_t* keys, int64_t n, int32_t n_parts, uint8_t* part) {
    const double inv = 1.0 / (double)n_parts;
    const int64_t nquads = (n + 3) / 4;
    for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < nquads; q += (int64_t)gridDim.x * blockDim.x) {
        const int64_t base = q * 4;
        const int4 kv = *reinterpret_cast<const int4*>(keys + base);  // buffers carry slack past the last row
        const uint32_t p0 = hs_py_partition_inv(kv.x, n_parts, inv), p1 = hs_py_partition_inv(kv.y, n_parts, inv),
                       p2 = hs_py_partition_inv(kv.z, n_parts, inv), p3 = hs_py_partition_inv(kv.w, n_parts, inv);
        if (base + 4 <= n) {
            *reinterpret_cast<uint32_t*>(part + base) = p0 | (p1 << 8) | (p2 << 16) | (p3 << 24);
        } else {
            const uint32_t p[4] = {p0, p1, p2, p3};
            for (int j = 0; base + j < n; ++j) part[base + j] = (uint8_t)p[j];
        }
    }
}
extern "C" int hs_partition_ids(void* stream, const hs_col* key, const int64_t* sel, int64_t nrows, int32_t n_parts,
                                uint8_t* part) {
    if (nrows == 0) return HS_OK;
    if (!key || !part || nrows < 0 || n_parts < 1 || n_parts > 256) {
        hs_set_error("hs_partition_ids: bad arguments");
        return HS_E_ARG;
    }
    if (key->kind == HS_I32 && !sel && n_parts <= 127 && (((uintptr_t)key->data) & 15) == 0 && (((uintptr_t)part) & 3) == 0) {
        hipLaunchKernelGGL(k_partition_ids_i32, dim3(grid_for((nrows + 3) / 4, 256)), dim3(256), 0, (hipStream_t)stream,
                           (const int32_t*)key->data, nrows, n_parts, part);
        HS_CHECK_LAUNCH("hs_partition_ids");
        return HS_OK;
    }
    hipLaunchKernelGGL(k_partition_ids, dim3(grid_for(nrows, 256)), dim3(256), 0, (hipStream_t)stream, *key, sel, nrows,
                       n_parts, part);
    HS_CHECK_LAUNCH("hs_partition_ids");
    return HS_OK;
}

// stable counting sort by partition id: per-tile histograms -> (partition-major, tile-minor) scan ->
// per-tile stable scatter.  Tiles = SCAN_TILE rows, each lane owns SCAN_ITEMS consecutive rows.
#define HS_MAX_PARTS_SORT 32
__global__ void __launch_bounds__(SCAN_WG) k_part_hist(const uint8_t* part, int64_t n, int32_t n_parts, int64_t ntiles,
                                                       int64_t* hist /* [n_parts][ntiles] */) {
    // round 3: a lane takes its 8 ids as ONE 8-byte load and counts every partition with a byte-compare in the word
    // (round 2: eight byte loads and eight LDS atomics on n_parts hot addresses per lane); waves add up with shuffles
    __shared__ int s_hist[HS_MAX_PARTS_SORT];
    if (threadIdx.x < HS_MAX_PARTS_SORT) s_hist[threadIdx.x] = 0;
    __syncthreads();
    const int64_t base = (int64_t)blockIdx.x * SCAN_TILE + (int64_t)threadIdx.x * SCAN_ITEMS;
    static_assert(SCAN_ITEMS == 8, "one 64-bit word of ids per lane");
    uint64_t w = ~0ull;  // 0xff = no row
    if (base + SCAN_ITEMS <= n && (((uintptr_t)part) & 7) == 0) {
        w = *(const uint64_t*)(part + base);
    } else {
        for (int k = 0; k < SCAN_ITEMS; ++k)
            if (base + k < n) w = (w & ~(0xffull << (8 * k))) | ((uint64_t)part[base + k] << (8 * k));
    }
    const int lane = threadIdx.x & (HS_WAVE - 1);
    // a lane's counts (<= 8 each) go into 10-bit fields, six partitions per 64-bit word: ONE shuffle reduction per word sums
    // them over the wave (<= 512 per field)
    constexpr int PER = 6, WORDS = (HS_MAX_PARTS_SORT + PER - 1) / PER;
    uint64_t acc[WORDS] = {};
#pragma unroll
    for (int r = 0; r < WORDS; ++r) {
#pragma unroll
        for (int q = 0; q < PER; ++q) {
            const int p = r * PER + q;
            if (p < n_parts) {
                const uint64_t x = w ^ (0x0101010101010101ull * (uint64_t)p);  // bytes equal to p become zero
                const uint64_t zero = ~(((x & 0x7f7f7f7f7f7f7f7full) + 0x7f7f7f7f7f7f7f7full) | x) & 0x8080808080808080ull;
                acc[r] += (uint64_t)__popcll(zero) << (10 * q);
            }
        }
        if (r * PER < n_parts) {
            for (int d = HS_WAVE / 2; d >= 1; d >>= 1) acc[r] += __shfl_down(acc[r], d, HS_WAVE);
            if (lane == 0) {
                for (int q = 0; q < PER && r * PER + q < n_parts; ++q) {
                    const int c = (int)((acc[r] >> (10 * q)) & 1023u);
                    if (c) atomicAdd(&s_hist[r * PER + q], c);
                }
            }
        }
    }
    __syncthreads();
    if ((int)threadIdx.x < n_parts) hist[(int64_t)threadIdx.x * ntiles + blockIdx.x] = s_hist[threadIdx.x];
}
__global__ void __launch_bounds__(SCAN_WG) k_part_starts(const int64_t* hist_scanned, int64_t ntiles, int32_t n_parts,
                                                         int64_t total, int64_t* part_start) {
    if ((int)threadIdx.x < n_parts) part_start[threadIdx.x] = hist_scanned[(int64_t)threadIdx.x * ntiles];
    if (threadIdx.x == 0) part_start[n_parts] = total;
}
// Round 3: rows are taken STRIPED (step k of a wave = 64 consecutive rows), so a row's rank among the rows of its
// partition is "rows of the partition in earlier (step, wave) segments" + its rank among the lanes of its own step - one
// ballot per id bit, as in the radix tier's scatter - and the tile leaves through LDS in partition order, i.e. as one
// contiguous run of 8-byte stores per partition.  (Round 2: every lane owned 8 consecutive rows and kept its own counter per
// partition in LDS; the per-partition scan over 256 lanes ran on n_parts threads and the stores were scattered.)
__global__ void __launch_bounds__(SCAN_WG) k_part_scatter(const uint8_t* part, int64_t n, int32_t n_parts,
                                                          int64_t ntiles, const int64_t* hist_scanned, int64_t* perm) {
    constexpr int WAVES = SCAN_WG / HS_WAVE, SEGS = SCAN_ITEMS * WAVES;
    __shared__ int s_seg[SEGS + 1][HS_MAX_PARTS_SORT];   // rows of partition p in segment (step, wave); then exclusive over segments
    __shared__ int s_start[HS_MAX_PARTS_SORT + 1];        // tile-local start of every partition
    __shared__ int64_t s_gbase[HS_MAX_PARTS_SORT];        // global position of the partition's first row of this tile
    __shared__ uint16_t s_row[SCAN_TILE];                 // tile-local row ids in partition order
    const int tid = threadIdx.x, lane = tid & (HS_WAVE - 1), w = tid / HS_WAVE;
    const int64_t tile0 = (int64_t)blockIdx.x * SCAN_TILE;
    for (int i = tid; i < (SEGS + 1) * HS_MAX_PARTS_SORT; i += SCAN_WG) (&s_seg[0][0])[i] = 0;
    uint8_t mine[SCAN_ITEMS];
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) {
        const int64_t r = tile0 + k * SCAN_WG + tid;
        mine[k] = r < n ? part[r] : 255;
    }
    __syncthreads();
    const uint64_t below = (1ull << lane) - 1ull;
    int rank[SCAN_ITEMS];
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) {
        const bool valid = mine[k] != 255;
        const int p = valid ? mine[k] : 0;
        uint64_t peers = __ballot(valid);
#pragma unroll
        for (int bit = 0; bit < 5; ++bit) {  // ids < 32
            const bool on = (p >> bit) & 1;
            const uint64_t bal = __ballot(valid && on);
            peers &= on ? bal : ~bal;
        }
        rank[k] = __popcll(peers & below);
        if (valid && rank[k] == 0) s_seg[k * WAVES + w][p] = __popcll(peers);
    }
    __syncthreads();
    if (tid < n_parts) {  // exclusive scan over the segments, in (step, wave) order = row order
        int run = 0;
        for (int sg = 0; sg <= SEGS; ++sg) {
            const int c = s_seg[sg][tid];
            s_seg[sg][tid] = run;
            run += c;
        }
        s_gbase[tid] = hist_scanned[(int64_t)tid * ntiles + blockIdx.x];
    }
    __syncthreads();
    if (tid == 0) {
        int run = 0;
        for (int p = 0; p < n_parts; ++p) {
            s_start[p] = run;
            run += s_seg[SEGS][p];
        }
        s_start[n_parts] = run;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) {
        if (mine[k] == 255) continue;
        const int p = mine[k];
        s_row[s_start[p] + s_seg[k * WAVES + w][p] + rank[k]] = (uint16_t)(k * SCAN_WG + tid);
    }
    __syncthreads();
    const int rows = s_start[n_parts];
    for (int i = tid; i < rows; i += SCAN_WG) {
        int p = 0;
        while (s_start[p + 1] <= i) ++p;  // <= 32 partitions: a short walk
        perm[s_gbase[p] + (i - s_start[p])] = tile0 + s_row[i];
    }
}
extern "C" size_t hs_partition_ws_bytes(int64_t nrows, int32_t n_parts) {
    const int64_t ntiles = (nrows + SCAN_TILE - 1) / SCAN_TILE;
    const int64_t nhist = ntiles * n_parts;
    return (size_t)(2 * nhist + 2) * 8 + hs_scan_ws_bytes(nhist) + 64;
}
extern "C" int hs_partition_perm(void* stream, const uint8_t* part, int64_t nrows, int32_t n_parts, int64_t* perm,
                                 int64_t* part_start, void* ws) {
    if ((!part && nrows > 0) || (!perm && nrows > 0) || !part_start || !ws || nrows < 0 || n_parts < 1 ||
        n_parts > HS_MAX_PARTS_SORT) {
        hs_set_error("hs_partition_perm: bad arguments (n_parts <= %d)", HS_MAX_PARTS_SORT);
        return HS_E_ARG;
    }
    hipStream_t s = (hipStream_t)stream;
    const int64_t ntiles = (nrows + SCAN_TILE - 1) / SCAN_TILE;
    const int64_t nhist = ntiles * n_parts;
    int64_t* hist = (int64_t*)ws;
    int64_t* hist_scanned = hist + nhist + 1;
    void* scan_ws = hist_scanned + nhist + 1;
    if (ntiles > 0) {
        hipLaunchKernelGGL(k_part_hist, dim3((unsigned)ntiles), dim3(SCAN_WG), 0, s, part, nrows, n_parts, ntiles, hist);
    }
    int rc = run_scan(s, InI64{hist}, nhist, EmitOffsets{hist_scanned}, hist_scanned + nhist, scan_ws,
                      "hs_partition_perm");
    if (rc) return rc;
    if (ntiles > 0) {
        hipLaunchKernelGGL(k_part_starts, dim3(1), dim3(SCAN_WG), 0, s, hist_scanned, ntiles, n_parts, nrows, part_start);
        hipLaunchKernelGGL(k_part_scatter, dim3((unsigned)ntiles), dim3(SCAN_WG), 0, s, part, nrows, n_parts, ntiles,
                           hist_scanned, perm);
    } else {
        hs_memset_async(part_start, 0, (size_t)(n_parts + 1) * 8, s);
    }
    HS_CHECK_LAUNCH("hs_partition_perm");
    return HS_OK;
}

// ---- A8: hash join (reference tasks.py:201-240) ------------------------------------------------------------
// Build = global-memory dictionary over the left keys (slot per distinct key) + CSR lists of the left
// rows of every slot in ASCENDING row order (the reference appends row indices in row order,
// tasks.py:216-217).  Entries never change once claimed, so a stale read can only see "empty" and the
// CAS that follows arbitrates.
__global__ void __launch_bounds__(256) k_fill_u64(uint64_t* p, int64_t n, uint64_t v) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) p[i] = v;
}

// Returns the slot of `k`, claiming one if the key is new (`inserted`); -1 when the table is full.  Test, then
// test-and-set: scattered 64-bit atomics are the scarce resource of the build (the chip sustains ~20 G/s of them),
// so a key that is already in its slot costs a plain read, not a CAS.
__device__ __forceinline__ int64_t gdict_upsert(uint64_t* keys, int64_t* reps, uint64_t mask, const hs_col& c,
                                                bool hashed, uint64_t k, int64_t row, bool& inserted) {
    uint64_t h = hs_mix64(k) & mask;
    inserted = false;
    for (uint64_t probe = 0; probe <= mask; ++probe) {
        if (hashed) {
            long long cur = __hip_atomic_load(&reps[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (cur < 0) {
                cur = (long long)atomicCAS((unsigned long long*)&reps[h], (unsigned long long)(-1ll), (unsigned long long)row);
                if (cur < 0) {
                    inserted = true;
                    return (int64_t)h;
                }
            }
            if (hs_rows_equal(c, (int64_t)cur, row)) return (int64_t)h;
        } else {
            uint64_t cur = __hip_atomic_load(&keys[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (cur == HS_EMPTY_KEY) {
                cur = atomicCAS((unsigned long long*)&keys[h], (unsigned long long)HS_EMPTY_KEY, (unsigned long long)k);
                if (cur == HS_EMPTY_KEY) {
                    inserted = true;
                    return (int64_t)h;
                }
            }
            if (cur == k) return (int64_t)h;
        }
        h = (h + 1) & mask;
    }
    return -1;
}
// read-only lookup after the build kernel has completed
__device__ __forceinline__ int64_t gdict_find(const uint64_t* keys, const int64_t* reps, uint64_t mask,
                                              const hs_col& build_col, const hs_col& probe_col, bool hashed,
                                              uint64_t k, int64_t probe_row) {
    uint64_t h = hs_mix64(k) & mask;
    for (uint64_t probe = 0; probe <= mask; ++probe) {
        if (hashed) {
            const int64_t rep = reps[h];
            if (rep < 0) return -1;
            if (hs_str_cmp(hs_str_at(build_col, rep), hs_str_at(probe_col, probe_row)) == 0) return (int64_t)h;
        } else {
            const uint64_t cur = keys[h];
            if (cur == HS_EMPTY_KEY) return -1;
            if (cur == k) return (int64_t)h;
        }
        h = (h + 1) & mask;
    }
    return -1;
}

// position i of the build input is row sel[i] (or row0 + i) of the key column
__global__ void __launch_bounds__(256) k_join_slots(const hs_col key, const int64_t* sel, int64_t row0, int64_t n,
                                                    int64_t cap, uint64_t* tkeys, int64_t* treps, int64_t* slot_of_row,
                                                    int64_t* slot_extra, uint32_t* flags) {
    const bool hashed = !hs_col_packs(key);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = sel ? sel[i] : row0 + i;
        bool inserted;
        const int64_t s = gdict_upsert(tkeys, treps, (uint64_t)cap - 1, key, hashed, hs_key_at(key, row), row, inserted);
        slot_of_row[i] = s;
        if (s < 0) atomicOr(flags, HS_FLAG_DICT_FULL);
        else if (!inserted) atomicAdd((unsigned long long*)&slot_extra[s], 1ull);  // rows beyond a slot's first
    }
}
// The same with one table REGION per unit: position i belongs to the unit whose position range holds it (binary search
// over unit_bounds), its key is looked up inside that unit's region [region_base[u], region_base[u + 1]) (a power of
// two of slots) only - so every slot belongs to one unit, and walking the slots in order walks the groups unit by
// unit.  One build over a whole batch instead of one per unit (hs_group_build_units).
__global__ void __launch_bounds__(256) k_join_slots_units(const hs_col key, const int64_t* sel, int64_t row0, int64_t n,
                                                          const int64_t* unit_bounds, const int64_t* region_base,
                                                          int32_t n_units, uint64_t* tkeys, int64_t* treps,
                                                          int64_t* slot_of_row, int64_t* slot_extra, uint32_t* flags) {
    const bool hashed = !hs_col_packs(key);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        int lo = 0, hi = n_units;  // last unit whose first position is <= i
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (unit_bounds[mid] <= i) lo = mid;
            else hi = mid;
        }
        const int64_t base = region_base[lo], size = region_base[lo + 1] - base;
        const int64_t row = sel ? sel[i] : row0 + i;
        bool inserted;
        int64_t s = size > 0 ? gdict_upsert(tkeys + base, treps + base, (uint64_t)size - 1, key, hashed, hs_key_at(key, row), row, inserted) : -1;
        if (s >= 0) s += base;
        slot_of_row[i] = s;
        if (s < 0) atomicOr(flags, HS_FLAG_DICT_FULL);
        else if (!inserted) atomicAdd((unsigned long long*)&slot_extra[s], 1ull);
    }
}
// rows of slot s = 1 for the row that claimed it + slot_extra[s]
struct InSlotCount {
    const uint64_t* tkeys;
    const int64_t* treps;
    const int64_t* extra;
    bool hashed;
    __device__ __forceinline__ int64_t operator()(int64_t s) const {
        const bool occupied = hashed ? treps[s] >= 0 : tkeys[s] != HS_EMPTY_KEY;
        return (occupied ? 1 : 0) + extra[s];
    }
};
__global__ void __launch_bounds__(256) k_join_rows(const int64_t* slot_of_row, int64_t n, const int64_t* slot_start,
                                                   int64_t* cursor, int64_t* rows) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t s = slot_of_row[i];
        if (s < 0) continue;
        const int64_t b = slot_start[s];
        // a slot with a single row (unique build keys: the usual join) needs no cursor
        const int64_t pos = slot_start[s + 1] - b == 1 ? b : b + (int64_t)atomicAdd((unsigned long long*)&cursor[s], 1ull);
        rows[pos] = i;
    }
}
// The fill above lands the rows of a slot in arrival order; every list must end up ascending.  Short lists
// (the usual join: one or a few rows per key) are sorted by one lane - they arrive almost sorted, so insertion
// sort is near linear; longer ones (GROUP BY through this build: thousands of rows per group) are queued and
// sorted by a workgroup each - an insertion sort by one lane would cost len^2 dependent global round trips.
static constexpr int HS_SORT_SHORT = 24;
static constexpr int HS_SORT_LDS = 16384;  // elements of a list one workgroup sorts in LDS (128 KiB)
__global__ void __launch_bounds__(256) k_join_sort(const int64_t* slot_start, int64_t cap, int64_t* rows,
                                                   int64_t* long_list, int64_t* long_count) {
    for (int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; s < cap; s += (int64_t)gridDim.x * blockDim.x) {
        const int64_t lo = slot_start[s], hi = slot_start[s + 1];
        if (hi - lo > HS_SORT_SHORT) {
            long_list[atomicAdd((unsigned long long*)long_count, 1ull)] = s;
            continue;
        }
        for (int64_t i = lo + 1; i < hi; ++i) {
            const int64_t v = rows[i];
            int64_t j = i - 1;
            while (j >= lo && rows[j] > v) {
                rows[j + 1] = rows[j];
                --j;
            }
            rows[j + 1] = v;
        }
    }
}
// Bitonic sort, the variant whose every compare-exchange is ascending (the first stage of each merge pairs i with
// its mirror image in the block): positions >= len then act as +infinity without ever being touched.
template <typename T>
__device__ __forceinline__ void bitonic_sort_ascending(T* a, int64_t len) {
    const int tid = threadIdx.x, nthr = blockDim.x;
    int64_t p2 = 1;
    while (p2 < len) p2 <<= 1;
    for (int64_t size = 2; size <= p2; size <<= 1) {
        const int64_t half = size >> 1;
        for (int64_t t = tid; t < (p2 >> 1); t += nthr) {
            const int64_t blk = t / half, off = t - blk * half;
            const int64_t i = blk * size + off, j = blk * size + size - 1 - off;
            if (j < len) {
                const T x = a[i], y = a[j];
                if (x > y) { a[i] = y; a[j] = x; }
            }
        }
        __syncthreads();
        for (int64_t stride = size >> 2; stride >= 1; stride >>= 1) {
            for (int64_t t = tid; t < (p2 >> 1); t += nthr) {
                const int64_t i = (t / stride) * (stride << 1) + (t % stride), j = i + stride;
                if (j < len) {
                    const T x = a[i], y = a[j];
                    if (x > y) { a[i] = y; a[j] = x; }
                }
            }
            __syncthreads();
        }
    }
}
__global__ void __launch_bounds__(256) k_join_sort_long(const int64_t* slot_start, int64_t* rows, const int64_t* long_list,
                                                        const int64_t* long_count) {
    extern __shared__ __align__(16) int64_t s_list[];
    const int64_t n_long = *long_count;
    for (int64_t w = blockIdx.x; w < n_long; w += gridDim.x) {
        const int64_t s = long_list[w];
        const int64_t lo = slot_start[s], len = slot_start[s + 1] - lo;
        if (len <= HS_SORT_LDS) {
            for (int64_t i = threadIdx.x; i < len; i += blockDim.x) s_list[i] = rows[lo + i];
            __syncthreads();
            bitonic_sort_ascending(s_list, len);
            for (int64_t i = threadIdx.x; i < len; i += blockDim.x) rows[lo + i] = s_list[i];
            __syncthreads();
        } else {
            bitonic_sort_ascending(rows + lo, len);  // in global memory: rare (one key with > 16 Ki rows)
        }
    }
}

extern "C" size_t hs_join_build_ws_bytes(int64_t n_left, int64_t table_cap) {
    // slot_of_row[n_left] + slot_count[cap] + cursor[cap] + scan ws + queue of the long row lists (+ its counter)
    return (size_t)(n_left + 2 * table_cap + 2) * 8 + hs_scan_ws_bytes(table_cap) + 64 +
           (size_t)(n_left / HS_SORT_SHORT + 8) * 8;
}

static int group_build(void* stream, const hs_col* left_key, const int64_t* sel, int64_t row0, int64_t n_left,
                       int64_t table_cap, uint64_t* table_keys, int64_t* table_reps, int64_t* slot_start, int64_t* rows,
                       void* ws, uint32_t* flags, const int64_t* unit_bounds = nullptr, const int64_t* region_base = nullptr,
                       int32_t n_units = 0);

extern "C" int hs_join_build(void* stream, const hs_col* left_key, int64_t n_left, int64_t table_cap,
                             uint64_t* table_keys, int64_t* table_reps, int64_t* slot_start, int64_t* rows, void* ws,
                             uint32_t* flags) {
    return group_build(stream, left_key, nullptr, 0, n_left, table_cap, table_keys, table_reps, slot_start, rows, ws,
                       flags);
}

// Group dictionary over positions 0..n-1 of an input whose position i is row sel[i] (or row0 + i): the same
// build as the join's.  rows[] receives POSITIONS (ascending inside every slot); table_reps[] holds ROW ids.
extern "C" int hs_group_build(void* stream, const hs_col* key, const int64_t* sel, int64_t row0, int64_t n,
                              int64_t table_cap, uint64_t* table_keys, int64_t* table_reps, int64_t* slot_start,
                              int64_t* positions, void* ws, uint32_t* flags) {
    return group_build(stream, key, sel, row0, n, table_cap, table_keys, table_reps, slot_start, positions, ws, flags);
}

// One build over every unit of a batch (the HBM tier of GROUP BY in a single pass): unit u owns positions
// [unit_bounds[u], unit_bounds[u + 1]) and the table region [region_base[u], region_base[u + 1]) - a power of two of
// slots, at least twice the unit's rows; table_cap = region_base[n_units].  Slots in ascending order are groups in
// unit order.
extern "C" int hs_group_build_units(void* stream, const hs_col* key, const int64_t* sel, int64_t row0, int64_t n,
                                    const int64_t* unit_bounds, const int64_t* region_base, int32_t n_units,
                                    int64_t table_cap, uint64_t* table_keys, int64_t* table_reps, int64_t* slot_start,
                                    int64_t* positions, void* ws, uint32_t* flags) {
    if (!unit_bounds || !region_base || n_units < 1) {
        hs_set_error("hs_group_build_units: bad unit description");
        return HS_E_ARG;
    }
    return group_build(stream, key, sel, row0, n, table_cap, table_keys, table_reps, slot_start, positions, ws, flags,
                       unit_bounds, region_base, n_units);
}

static int group_build(void* stream, const hs_col* left_key, const int64_t* sel, int64_t row0, int64_t n_left,
                       int64_t table_cap, uint64_t* table_keys, int64_t* table_reps, int64_t* slot_start, int64_t* rows,
                       void* ws, uint32_t* flags, const int64_t* unit_bounds, const int64_t* region_base, int32_t n_units) {
    if (!left_key || !table_keys || !table_reps || !slot_start || (!rows && n_left > 0) || !ws || !flags ||
        n_left < 0 || table_cap < 1 || (!unit_bounds && (table_cap & (table_cap - 1)))) {
        hs_set_error("hs_join_build: bad arguments (table_cap must be a power of two)");
        return HS_E_ARG;
    }
    hipStream_t s = (hipStream_t)stream;
    int64_t* slot_of_row = (int64_t*)ws;
    int64_t* slot_count = slot_of_row + n_left;
    int64_t* cursor = slot_count + table_cap;
    void* scan_ws = cursor + table_cap + 1;
    hipLaunchKernelGGL(k_fill_u64, dim3(grid_for(table_cap, 256)), dim3(256), 0, s, table_keys, table_cap, HS_EMPTY_KEY);
    hipLaunchKernelGGL(k_fill_u64, dim3(grid_for(table_cap, 256)), dim3(256), 0, s, (uint64_t*)table_reps, table_cap,
                       ~0ull);
    hs_memset_async(slot_count, 0, (size_t)table_cap * 16, s);  // slot_count + cursor
    if (n_left > 0 && unit_bounds)
        hipLaunchKernelGGL(k_join_slots_units, dim3(grid_for(n_left, 256)), dim3(256), 0, s, *left_key, sel, row0, n_left,
                           unit_bounds, region_base, n_units, table_keys, table_reps, slot_of_row, slot_count, flags);
    else if (n_left > 0)
        hipLaunchKernelGGL(k_join_slots, dim3(grid_for(n_left, 256)), dim3(256), 0, s, *left_key, sel, row0, n_left,
                           table_cap, table_keys, table_reps, slot_of_row, slot_count, flags);
    int rc = run_scan(s, InSlotCount{table_keys, table_reps, slot_count, !hs_col_packs(*left_key)}, table_cap,
                      EmitOffsets{slot_start}, slot_start + table_cap, scan_ws, "hs_join_build");
    if (rc) return rc;
    if (n_left > 0) {
        hipLaunchKernelGGL(k_join_rows, dim3(grid_for(n_left, 256)), dim3(256), 0, s, slot_of_row, n_left, slot_start,
                           cursor, rows);
        int64_t* long_count = (int64_t*)((char*)scan_ws + ((hs_scan_ws_bytes(table_cap) + 63) & ~(size_t)63));
        int64_t* long_list = long_count + 1;
        hs_memset_async(long_count, 0, 8, s);
        hipLaunchKernelGGL(k_join_sort, dim3(grid_for(table_cap, 256)), dim3(256), 0, s, slot_start, table_cap, rows,
                           long_list, long_count);
        static unsigned long long attr_set = 0;
        if (hs_first_on_device(attr_set)) {
            hipFuncSetAttribute((const void*)k_join_sort_long, hipFuncAttributeMaxDynamicSharedMemorySize, HS_SORT_LDS * 8);
        }
        int64_t wgs = n_left / HS_SORT_SHORT + 1;  // at most this many long lists; every workgroup exits on the count
        if (wgs > 1024) wgs = 1024;
        hipLaunchKernelGGL(k_join_sort_long, dim3((unsigned)wgs), dim3(256), HS_SORT_LDS * 8, s, slot_start, rows, long_list,
                           long_count);
    }
    HS_CHECK_LAUNCH("hs_join_build");
    return HS_OK;
}

// out[q] = number of elements of the ascending list sorted[0 .. n) that are < queries[q] (lower bound): unit
// boundaries inside a row-index list, groups per unit inside a slot list.  n_dev (optional) caps n.
__global__ void __launch_bounds__(256) k_lower_bound_i64(const int64_t* sorted, int64_t n, const int64_t* n_dev,
                                                         const int64_t* queries, int64_t nq, int64_t* out) {
    if (n_dev) {
        const int64_t d = *n_dev;
        n = d < n ? d : n;
    }
    for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < nq; q += (int64_t)gridDim.x * blockDim.x) {
        const int64_t x = queries[q];
        int64_t lo = 0, hi = n;
        while (lo < hi) {
            const int64_t mid = (lo + hi) >> 1;
            if (sorted[mid] < x) lo = mid + 1;
            else hi = mid;
        }
        out[q] = lo;
    }
}
extern "C" int hs_lower_bound_i64(void* stream, const int64_t* sorted, int64_t n, const int64_t* n_dev,
                                  const int64_t* queries, int64_t n_queries, int64_t* out) {
    if ((!sorted && n > 0) || !queries || !out || n < 0 || n_queries < 0) {
        hs_set_error("hs_lower_bound_i64: bad arguments");
        return HS_E_ARG;
    }
    if (n_queries > 0)
        hipLaunchKernelGGL(k_lower_bound_i64, dim3(grid_for(n_queries, 256)), dim3(256), 0, (hipStream_t)stream, sorted, n, n_dev,
                           queries, n_queries, out);
    HS_CHECK_LAUNCH("hs_lower_bound_i64");
    return HS_OK;
}

// ---- A5/A7 global-memory tier: fold the value columns of every group in ascending position order --------------
// (reference fill_aggregators tasks.py:295-310: counter[key] = counter.get(key, identity) op x, row by row - one
// lane per group walks its row list front to back, so the fp64 additions happen in EXACTLY the reference's order)
__global__ void __launch_bounds__(256) k_group_mask(const int64_t* slot_start, int64_t cap, uint8_t* mask) {
    for (int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; s < cap; s += (int64_t)gridDim.x * blockDim.x)
        mask[s] = slot_start[s + 1] > slot_start[s];
}
struct GroupFoldArgs {
    hs_col vals[HS_MAX_ACC];
    hs_agg_spec spec;
    const int64_t* slot_list;   // dense list of non-empty slots
    int64_t n_groups_max;
    const int64_t* n_groups_dev;
    const int64_t* slot_start;
    const int64_t* positions;
    const int64_t* sel;
    int64_t row0;
    int32_t quantise;
    int32_t pad;
    int64_t* out_rep_row;  // [n_groups_max] row id of the group's first row
    uint64_t* out_acc;     // [n_acc][n_groups_max]
    uint32_t* flags;
};
__global__ void __launch_bounds__(256) k_group_fold(const GroupFoldArgs A_kernarg) {
    HS_KERNARG(GroupFoldArgs, A);
    uint32_t err = 0;
    const int64_t ng = capped(A.n_groups_max, A.n_groups_dev);
    const int NA = A.spec.n_acc;
    for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < ng; g += (int64_t)gridDim.x * blockDim.x) {
        const int64_t s = A.slot_list[g];
        const int64_t b = A.slot_start[s], e = A.slot_start[s + 1];
        const int64_t first = A.positions[b];
        A.out_rep_row[g] = A.sel ? A.sel[first] : A.row0 + first;
        for (int a = 0; a < NA; ++a) {
            const uint32_t op = A.spec.op[a];
            const bool is_int = A.spec.is_int[a] != 0;
            uint64_t v = hs_acc_identity(op, is_int);
            for (int64_t q = b; q < e; ++q) v = hs_acc_fold(op, is_int, v, hs_load_cell(A.vals[a], A.positions[q]));
            if (hs_float_identity_left(op, is_int, v)) err |= HS_FLAG_TYPE_ASSERT;  // both phases write a file next
            if (A.quantise) v = hs_quantise_cell(is_int, v, err);
            A.out_acc[(int64_t)a * A.n_groups_max + g] = v;
        }
    }
    if (err) atomicOr(A.flags, err);
}
extern "C" int hs_group_mask(void* stream, const int64_t* slot_start, int64_t table_cap, uint8_t* mask) {
    if (!slot_start || !mask || table_cap < 1) {
        hs_set_error("hs_group_mask: bad arguments");
        return HS_E_ARG;
    }
    hipLaunchKernelGGL(k_group_mask, dim3(grid_for(table_cap, 256)), dim3(256), 0, (hipStream_t)stream, slot_start,
                       table_cap, mask);
    HS_CHECK_LAUNCH("hs_group_mask");
    return HS_OK;
}
extern "C" int hs_group_fold(void* stream, const hs_col* val_cols, const hs_agg_spec* spec, const int64_t* slot_list,
                             int64_t n_groups_max, const int64_t* n_groups_dev, const int64_t* slot_start,
                             const int64_t* positions, const int64_t* sel, int64_t row0, int32_t quantise,
                             int64_t* out_rep_row, uint64_t* out_acc, uint32_t* flags) {
    if (n_groups_max == 0) return HS_OK;
    if (!spec || !slot_list || !slot_start || !positions || !out_rep_row || !out_acc || !flags || n_groups_max < 0 ||
        spec->n_acc < 0 || spec->n_acc > HS_MAX_ACC || (spec->n_acc > 0 && !val_cols)) {
        hs_set_error("hs_group_fold: bad arguments");
        return HS_E_ARG;
    }
    GroupFoldArgs A;
    for (int a = 0; a < HS_MAX_ACC; ++a)
        A.vals[a] = a < spec->n_acc ? val_cols[a] : hs_col{HS_U8, -1, nullptr, nullptr, nullptr};
    A.spec = *spec;
    A.slot_list = slot_list;
    A.n_groups_max = n_groups_max;
    A.n_groups_dev = n_groups_dev;
    A.slot_start = slot_start;
    A.positions = positions;
    A.sel = sel;
    A.row0 = row0;
    A.quantise = quantise;
    A.pad = 0;
    A.out_rep_row = out_rep_row;
    A.out_acc = out_acc;
    A.flags = flags;
    hipLaunchKernelGGL(k_group_fold, dim3(grid_for(n_groups_max, 256)), dim3(256), 0, (hipStream_t)stream, A);
    HS_CHECK_LAUNCH("hs_group_fold");
    return HS_OK;
}

struct JoinProbeArgs {
    hs_col left_key, right_key;
    int64_t n_right, cap;
    const uint64_t* tkeys;
    const int64_t* treps;
    const int64_t* slot_start;
    const int64_t* rows;
};
__global__ void __launch_bounds__(256) k_join_count(const JoinProbeArgs A, int64_t* counts) {
    const bool hashed = !hs_col_packs(A.left_key);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < A.n_right; i += (int64_t)gridDim.x * blockDim.x) {
        // the key word of a probe row must be computed like a build row's: same kind rules
        const int64_t s = gdict_find(A.tkeys, A.treps, (uint64_t)A.cap - 1, A.left_key, A.right_key, hashed,
                                     hs_key_at(A.right_key, i), i);
        counts[i] = s < 0 ? 0 : A.slot_start[s + 1] - A.slot_start[s];
    }
}
__global__ void __launch_bounds__(256) k_join_fill(const JoinProbeArgs A, const int64_t* out_start, int64_t* out_left,
                                                   int64_t* out_right) {
    const bool hashed = !hs_col_packs(A.left_key);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < A.n_right; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t n = out_start[i + 1] - out_start[i];
        if (n == 0) continue;
        const int64_t s = gdict_find(A.tkeys, A.treps, (uint64_t)A.cap - 1, A.left_key, A.right_key, hashed,
                                     hs_key_at(A.right_key, i), i);
        const int64_t lo = A.slot_start[s];
        int64_t o = out_start[i];
        for (int64_t k = 0; k < n; ++k, ++o) {
            out_left[o] = A.rows[lo + k];
            out_right[o] = i;
        }
    }
}
static int fill_probe(JoinProbeArgs& A, const hs_col* lk, const hs_col* rk, int64_t n_right, int64_t cap,
                      const uint64_t* tkeys, const int64_t* treps, const int64_t* slot_start, const int64_t* rows,
                      const char* name) {
    if (!lk || !rk || !tkeys || !treps || !slot_start || n_right < 0 || cap < 1 || (cap & (cap - 1))) {
        hs_set_error("%s: bad arguments", name);
        return HS_E_ARG;
    }
    const bool ls = lk->kind == HS_STR, rs = rk->kind == HS_STR;
    if (ls != rs) {
        hs_set_error("%s: join keys must both be strings or both be numeric", name);
        return HS_E_ARG;
    }
    A.left_key = *lk;
    A.right_key = *rk;
    A.n_right = n_right;
    A.cap = cap;
    A.tkeys = tkeys;
    A.treps = treps;
    A.slot_start = slot_start;
    A.rows = rows;
    return HS_OK;
}
extern "C" int hs_join_count(void* stream, const hs_col* left_key, const hs_col* right_key, int64_t n_right,
                             int64_t table_cap, const uint64_t* table_keys, const int64_t* table_reps,
                             const int64_t* slot_start, int64_t* counts) {
    JoinProbeArgs A;
    int rc = fill_probe(A, left_key, right_key, n_right, table_cap, table_keys, table_reps, slot_start, nullptr,
                        "hs_join_count");
    if (rc) return rc;
    if (n_right == 0) return HS_OK;
    hipLaunchKernelGGL(k_join_count, dim3(grid_for(n_right, 256)), dim3(256), 0, (hipStream_t)stream, A, counts);
    HS_CHECK_LAUNCH("hs_join_count");
    return HS_OK;
}
extern "C" int hs_join_fill(void* stream, const hs_col* left_key, const hs_col* right_key, int64_t n_right,
                            int64_t table_cap, const uint64_t* table_keys, const int64_t* table_reps,
                            const int64_t* slot_start, const int64_t* rows, const int64_t* out_start,
                            int64_t* out_left, int64_t* out_right) {
    JoinProbeArgs A;
    int rc = fill_probe(A, left_key, right_key, n_right, table_cap, table_keys, table_reps, slot_start, rows,
                        "hs_join_fill");
    if (rc) return rc;
    if (n_right == 0) return HS_OK;
    hipLaunchKernelGGL(k_join_fill, dim3(grid_for(n_right, 256)), dim3(256), 0, (hipStream_t)stream, A, out_start,
                       out_left, out_right);
    HS_CHECK_LAUNCH("hs_join_fill");
    return HS_OK;
}

// ---- multi-GPU: un-interleave all-gathered exchange slabs (minispark_amd/distributed.py SlabLayout) --------------
// slab := [flags u32][pad u32][row count i64][order key i64 x M][column 0: M x row_bytes][column 1] ...
#define HS_MAX_SLAB_COLS 20
struct SlabUnpackArgs {
    const uint8_t* gathered;  // [world][slab_bytes]
    int32_t world;
    int32_t n_cols;
    int64_t slab_bytes, slab_rows, order_offset;
    int64_t col_offset[HS_MAX_SLAB_COLS];
    int32_t col_row_bytes[HS_MAX_SLAB_COLS];
    void* col_dst[HS_MAX_SLAB_COLS];  // contiguous over world * slab_rows rows, rank-major
    int32_t* flags_out;               // [world]
    int64_t* order_out;               // [world * slab_rows], -1 on padding rows
};
__global__ void __launch_bounds__(256) k_slab_unpack(const SlabUnpackArgs A_kernarg) {
    HS_KERNARG(SlabUnpackArgs, A);
    const int64_t total = (int64_t)A.world * A.slab_rows;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t rank = i / A.slab_rows, row = i % A.slab_rows;
        const uint8_t* slab = A.gathered + rank * A.slab_bytes;
        const int64_t count = *reinterpret_cast<const int64_t*>(slab + 8);
        A.order_out[i] = row < count ? reinterpret_cast<const int64_t*>(slab + A.order_offset)[row] : -1;
        if (row == 0) A.flags_out[rank] = *reinterpret_cast<const int32_t*>(slab);
        for (int c = 0; c < A.n_cols; ++c) {
            const int rb = A.col_row_bytes[c];
            const uint8_t* src = slab + A.col_offset[c] + row * rb;
            uint8_t* dst = (uint8_t*)A.col_dst[c] + i * rb;
            switch (rb) {
                case 8: *reinterpret_cast<uint64_t*>(dst) = *reinterpret_cast<const uint64_t*>(src); break;
                case 4: *reinterpret_cast<uint32_t*>(dst) = *reinterpret_cast<const uint32_t*>(src); break;
                case 2: *reinterpret_cast<uint16_t*>(dst) = *reinterpret_cast<const uint16_t*>(src); break;
                default:
                    for (int b = 0; b < rb; ++b) dst[b] = src[b];
                    break;
            }
        }
    }
}
extern "C" int hs_slab_unpack(void* stream, const uint8_t* gathered, int32_t world, int64_t slab_bytes,
                              int64_t slab_rows, int64_t order_offset, int32_t n_cols, const int64_t* col_offsets,
                              const int32_t* col_row_bytes, void* const* col_dsts, int32_t* flags_out,
                              int64_t* order_out) {
    if (!gathered || world < 1 || slab_rows < 0 || n_cols < 0 || n_cols > HS_MAX_SLAB_COLS || !flags_out ||
        (slab_rows > 0 && !order_out) || (n_cols > 0 && (!col_offsets || !col_row_bytes || !col_dsts))) {
        hs_set_error("hs_slab_unpack: bad arguments");
        return HS_E_ARG;
    }
    SlabUnpackArgs A;
    A.gathered = gathered;
    A.world = world;
    A.n_cols = n_cols;
    A.slab_bytes = slab_bytes;
    A.slab_rows = slab_rows;
    A.order_offset = order_offset;
    for (int c = 0; c < HS_MAX_SLAB_COLS; ++c) {
        A.col_offset[c] = c < n_cols ? col_offsets[c] : 0;
        A.col_row_bytes[c] = c < n_cols ? col_row_bytes[c] : 0;
        A.col_dst[c] = c < n_cols ? col_dsts[c] : nullptr;
    }
    A.flags_out = flags_out;
    A.order_out = order_out;
    const int64_t total = (int64_t)world * (slab_rows > 0 ? slab_rows : 1);
    hipLaunchKernelGGL(k_slab_unpack, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, A);
    HS_CHECK_LAUNCH("hs_slab_unpack");
    return HS_OK;
}

// ---- synthetic TPC-H-shaped lineitem --------------------------------------------------
// value(row i, column c) = f(splitmix64(seed ^ c*GOLDEN + i)).  oracle/q1_oracle.c holds the CPU twin.
__host__ __device__ __forceinline__ uint64_t hs_splitmix(uint64_t x) {
    x += 0x9e3779b97f4a7c15ull;
    x = (x ^ (x >> 30)) * 0xbf58476d1ce4e5b9ull;
    x = (x ^ (x >> 27)) * 0x94d049bb133111ebull;
    return x ^ (x >> 31);
}
__host__ __device__ __forceinline__ uint64_t hs_rand(uint64_t seed, uint32_t col, int64_t i) {
    return hs_splitmix(hs_splitmix(seed + 0x632be59bd9b4e019ull * (col + 1)) + (uint64_t)i);
}
struct GenArgs {
    uint64_t seed;
    int64_t row0, nrows;
    float *quantity, *extendedprice, *discount, *tax;
    int64_t* shipdate;
    uint8_t *returnflag, *flag_lens;
    int32_t* orderkey;
    uint8_t* shipmode_code;
};
__global__ void __launch_bounds__(256) k_gen_lineitem(const GenArgs A) {
    for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < A.nrows; k += (int64_t)gridDim.x * blockDim.x) {
        const int64_t i = A.row0 + k;
        const uint32_t qty = 1 + (uint32_t)(hs_rand(A.seed, 0, i) % 50);
        if (A.quantity) A.quantity[k] = (float)qty;
        if (A.extendedprice) {
            const uint32_t cents = 90000 + (uint32_t)(hs_rand(A.seed, 1, i) % 110001);
            A.extendedprice[k] = (float)((double)qty * (double)cents / 100.0);
        }
        if (A.discount) A.discount[k] = (float)((double)(hs_rand(A.seed, 2, i) % 11) / 100.0);
        if (A.tax) A.tax[k] = (float)((double)(hs_rand(A.seed, 3, i) % 9) / 100.0);
        if (A.shipdate) {
            // 1992-01-02 00:00:00 UTC = 694310400 s; + U{0..2525} days  (max = 1998-12-01)
            const int64_t days = (int64_t)(hs_rand(A.seed, 4, i) % 2526);
            A.shipdate[k] = (694310400ll + days * 86400ll) * 1000000ll;
        }
        if (A.returnflag) {
            const uint32_t r = (uint32_t)(hs_rand(A.seed, 5, i) % 4);
            A.returnflag[k] = r == 0 ? 'A' : (r == 3 ? 'R' : 'N');
        }
        if (A.flag_lens) A.flag_lens[k] = 1;
        if (A.orderkey) {
            const int64_t o = i / 4;
            A.orderkey[k] = (int32_t)(32 * (o / 8) + (o % 8) + 1);
        }
        if (A.shipmode_code) A.shipmode_code[k] = (uint8_t)(hs_rand(A.seed, 6, i) % 7);
    }
}
extern "C" int hs_gen_lineitem(void* stream, uint64_t seed, int64_t row0, int64_t nrows, float* quantity,
                               float* extendedprice, float* discount, float* tax, int64_t* shipdate,
                               uint8_t* returnflag, uint8_t* flag_lens, int32_t* orderkey, uint8_t* shipmode_code) {
    if (nrows <= 0) return HS_OK;
    GenArgs A{seed, row0, nrows, quantity, extendedprice, discount, tax, shipdate, returnflag, flag_lens, orderkey,
              shipmode_code};
    hipLaunchKernelGGL(k_gen_lineitem, dim3(grid_for(nrows, 1024)), dim3(256), 0, (hipStream_t)stream, A);
    HS_CHECK_LAUNCH("hs_gen_lineitem");
    return HS_OK;
}
