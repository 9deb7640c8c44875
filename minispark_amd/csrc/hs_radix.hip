// hs_radix.hip - the HBM tier of GROUP BY as a radix partition + on-chip ordered fold (round 2).
//
// Reference loop replaced: AggregateTask over a block, tasks.py:284-310 - a Python dict per block, every row folded
// into its group's accumulators in row order (fp64 / int), written to the shuffle file at the block's end.
//
// Round 1 kept one hash table for all rows in HBM: three passes of random 8-16 B accesses (insert, count, place) over
// a table twice the size of the input, then one lane per group walking its row list - bound by the rate of random
// HBM transactions, ~3 G rows/s.  Here rows are MOVED instead, in streams:
//
//   pass 1, pass 2   stable radix partition of the (key, value...) tuples on hash bits of the key, inside every unit
//                    (file block): histogram per 8192-row tile, one exclusive scan over (segment, bin, tile), scatter.
//                    After the passes every final partition holds <= ~cap/2 rows of ONE unit, still in row order.
//   fold             ONE WAVE per partition: a dictionary + accumulators private to the wave in LDS, rows taken 64 at a
//                    time in order; rows of one step that share a group are ranked (ballots over the slot bits) and
//                    folded in rank order - each group's values are added in ascending row order, i.e. exactly the
//                    reference's sequential fold, bit for bit.  The partition's groups go to a provisional place (the
//                    partition's own start), a scan over the partitions' group counts and a copy make them dense.
//
// Everything is sized on the host from the row count and the largest unit; no host round trip inside
// hs_group_radix_run.  A partition that meets more distinct keys than its dictionary holds raises HS_FLAG_DICT_FULL
// (the caller takes the round-1 path).  Keys: INTEGER / TIMESTAMP (the key word is the value).
#include <type_traits>
#include "hs_device.h"

#include <cstdlib>
#include <cstring>

extern thread_local char g_hs_err[256];
void hs_set_error(const char* fmt, ...);
extern "C" size_t hs_scan_ws_bytes(int64_t nrows);
extern "C" int hs_exclusive_scan_i64(void* stream, const int64_t* counts, int64_t n, int64_t* start, void* ws);

#define RX_CHECK_LAUNCH(name)                        \
    if (hipGetLastError() != hipSuccess) {           \
        hs_set_error(name ": kernel launch failed"); \
        return HS_E_LAUNCH;                          \
    }

constexpr int RX_THREADS = 1024;               // a partition-pass workgroup
constexpr int RX_PER = 8;                      // rows per thread and tile
constexpr int RX_TILE = RX_THREADS * RX_PER;   // 8192 rows
constexpr int RX_WAVES = RX_THREADS / HS_WAVE;
constexpr int RX_SUB = RX_TILE / RX_WAVES;     // rows of a tile one wave ranks (512, contiguous)
constexpr int RX_MAX_BITS = 8;                 // fan-out of one pass <= 256
constexpr int RX_COLS = 1 + HS_MAX_ACC;        // key + value columns

static_assert(sizeof(hs_radix_plan) % 8 == 0, "hs_radix_plan");

// ---- segments -> tiles ------------------------------------------------------------------------------------------
// tile_base[s] = number of tiles of the segments before s (a tile never straddles two segments)
__global__ void __launch_bounds__(1024) k_rx_tiles(const int64_t* seg_start, int64_t n_seg, int64_t* tile_base) {
    __shared__ int64_t s_part[16];
    const int tid = threadIdx.x, lane = tid & (HS_WAVE - 1), w = tid / HS_WAVE;
    const int64_t per = (n_seg + blockDim.x - 1) / blockDim.x;
    const int64_t s0 = tid * per < n_seg ? tid * per : n_seg, s1 = (s0 + per) < n_seg ? (s0 + per) : n_seg;
    int64_t mine = 0;
    for (int64_t s = s0; s < s1; ++s) mine += (seg_start[s + 1] - seg_start[s] + RX_TILE - 1) / RX_TILE;
    int64_t x = mine;
    for (int d = 1; d < HS_WAVE; d <<= 1) {
        const int64_t t = __shfl_up(x, d, HS_WAVE);
        if (lane >= d) x += t;
    }
    if (lane == HS_WAVE - 1) s_part[w] = x;
    __syncthreads();
    int64_t base = 0, all = 0;
    for (int k = 0; k < (int)(blockDim.x / HS_WAVE); ++k) {
        if (k < w) base += s_part[k];
        all += s_part[k];
    }
    int64_t run = x + base - mine;
    for (int64_t s = s0; s < s1; ++s) {
        tile_base[s] = run;
        run += (seg_start[s + 1] - seg_start[s] + RX_TILE - 1) / RX_TILE;
    }
    if (tid == 0) tile_base[n_seg] = all;
}

// tile -> (segment, tile inside the segment); false past the last tile.  Uniform over the workgroup.
__device__ __forceinline__ bool rx_find_tile(const int64_t* tile_base, int64_t n_seg, int64_t tile, int64_t& seg, int64_t& t) {
    if (tile >= tile_base[n_seg]) return false;
    int64_t lo = 0, hi = n_seg;  // the last s with tile_base[s] <= tile owns it (empty segments share a later base)
    while (hi - lo > 1) {
        const int64_t mid = (lo + hi) >> 1;
        if (tile_base[mid] <= tile) lo = mid;
        else hi = mid;
    }
    seg = lo;
    t = tile - tile_base[lo];
    return true;
}

struct RxPass {
    const int64_t* seg_start;  // [n_seg + 1] positions in the row list
    const int64_t* tile_base;  // [n_seg + 1]
    int64_t n_seg;
    int32_t shift, bits;       // bin = (mix(key word) >> shift) & (2^bits - 1); raw: ((key + 1) >> shift) & ... (hs_sort_by_order)
    int32_t n_cols, first;     // columns that travel (0 = key); first pass reads the key through `key` / `sel`
    int32_t range, range_bias; // range (4-byte keys only): 1: bin = ((key - range_bias) >> shift) & (2^bits - 1) - partitions are KEY RANGES,
                               // most significant bits first (the dense join build); 2: bin = bits of the key's hash WINDOW
                               // jh_window(key, range_bias) (the hashed join build); 0: a mix of the key's bits
    int32_t wide, pad_w;       // wide: a STRING key of one fixed length 8 .. 16 travels as this many 4-byte word columns (2 .. 4), round 3
    int32_t raw, key4;         // key4: the key is a 4-byte integer (bins from a 32-bit mix: a quarter of hs_mix64's multiplies)
    hs_col key;
    const int64_t* sel;
    int64_t row0;
    const void* src[RX_COLS];  // position-indexed raw arrays (src[0] unused in the first pass)
    void* dst[RX_COLS];
    int32_t esize[RX_COLS];    // bytes per element: 1, 4 or 8
    int64_t* counters;         // [(tiles) << bits] laid out (segment, bin, tile): counts, then their exclusive scan
};

// the value of column 0 at position i
__device__ __forceinline__ uint64_t rx_key(const RxPass& A, int64_t i) {
    if (A.first) return hs_key_at(A.key, A.sel ? A.sel[i] : A.row0 + i);
    return A.esize[0] == 4 ? (uint64_t)(int64_t)((const int32_t*)A.src[0])[i] : ((const uint64_t*)A.src[0])[i];
}
__device__ __forceinline__ uint32_t rx_bin(uint64_t word, int shift, int bits, int raw) {
    return (uint32_t)((raw ? word + 1 : hs_mix64(word)) >> shift) & ((1u << bits) - 1u);
}
// 4-byte keys: which partition a key lands in only has to be a function of the key that spreads well - one 32-bit multiply
// and a fold of the high half into the low (the passes take bits 0 .. 15) instead of hs_mix64's 64-bit multiplies, in the
// histogram and the scatter of either pass alike
__device__ __forceinline__ uint32_t rx_mix32(uint32_t k) {
    uint32_t h = k * 0x9E3779B1u;
    h ^= h >> 15;
    h *= 0x85EBCA77u;
    return h ^ (h >> 16);
}
__device__ __forceinline__ uint32_t rx_bin4(uint32_t key, int shift, int bits) { return (rx_mix32(key) >> shift) & ((1u << bits) - 1u); }
// ... or, for range partitions (RxPass.range; wave-uniform choice), bits of the key's offset itself
// the hashed join (hs_join_hash_*): a key's window of the table, 0 .. windows - 1 (multiply-shift: no power of two needed)
__device__ __forceinline__ uint32_t jh_window(uint32_t key, uint32_t windows) {
    return (uint32_t)(((uint64_t)rx_mix32(key) * windows) >> 32);
}
__device__ __forceinline__ uint32_t rx_bin4r(const RxPass& A, uint32_t key, int shift, int bits) {
    if (A.range == 2) return (jh_window(key, (uint32_t)A.range_bias) >> shift) & ((1u << bits) - 1u);
    return A.range ? ((key - (uint32_t)A.range_bias) >> shift) & ((1u << bits) - 1u) : rx_bin4(key, shift, bits);
}
__device__ __forceinline__ uint32_t rx_bin_of(const RxPass& A, uint64_t word) {
    return A.key4 ? rx_bin4r(A, (uint32_t)word, A.shift, A.bits) : rx_bin(word, A.shift, A.bits, A.raw);
}
__device__ __forceinline__ void rx_move(const void* src, void* dst, int esize, int64_t from, int64_t to) {
    if (esize == 4) ((uint32_t*)dst)[to] = ((const uint32_t*)src)[from];
    else if (esize == 8) ((uint64_t*)dst)[to] = ((const uint64_t*)src)[from];
    else ((uint8_t*)dst)[to] = ((const uint8_t*)src)[from];
}

__global__ void __launch_bounds__(RX_THREADS) k_rx_hist(const RxPass A_kernarg) {
    HS_KERNARG(RxPass, A);
    __shared__ uint32_t hist[RX_WAVES][1 << RX_MAX_BITS];  // one per wave: LDS atomics of different waves never meet
    int64_t seg, t;
    if (!rx_find_tile(A.tile_base, A.n_seg, blockIdx.x, seg, t)) return;
    const int tid = threadIdx.x, w = tid / HS_WAVE, F = 1 << A.bits;
    for (int i = tid; i < RX_WAVES * F; i += RX_THREADS) hist[i / F][i % F] = 0;
    __syncthreads();
    const int64_t b = A.seg_start[seg] + t * RX_TILE;
    const int64_t seg_end = A.seg_start[seg + 1];
    const int64_t e = (b + RX_TILE) < seg_end ? (b + RX_TILE) : seg_end;
    uint64_t word[RX_PER];
#pragma unroll
    for (int j = 0; j < RX_PER; ++j) {
        const int64_t i = b + tid + (int64_t)j * RX_THREADS;
        word[j] = i < e ? rx_key(A, i) : 0;
    }
#pragma unroll
    for (int j = 0; j < RX_PER; ++j)
        if (b + tid + (int64_t)j * RX_THREADS < e) atomicAdd(&hist[w][rx_bin_of(A, word[j])], 1u);
    __syncthreads();
    const int64_t nt = A.tile_base[seg + 1] - A.tile_base[seg];
    if (tid < F) {
        uint32_t total = 0;
        for (int k = 0; k < RX_WAVES; ++k) total += hist[k][tid];
        A.counters[(A.tile_base[seg] << A.bits) + (int64_t)tid * nt + t] = total;
    }
}

// Stable scatter of a tile.  Wave w ranks rows [w * 512, (w + 1) * 512) of the tile, 64 at a time in order: a row's
// rank among the rows of its bin = the wave's running count of the bin + its rank among this step's equal-bin lanes
// (the AND of one ballot per bin bit).  A scan over the waves per bin and the tile's scanned counter finish the address.
//
// The tile is first put in bin order in LDS, one column at a time, and written out by consecutive threads: the rows of
// a bin (32 on average at fan-out 256) leave as one or two contiguous segments instead of one 4-8 B store per row and
// bin - the store path of a CU takes a request per distinct line, not per byte.
// two workgroups per CU (<= 64 VGPRs, a few spilled) beat one at 94 VGPRs: 560 us against 840 us per pass of 64 M rows
__global__ void __launch_bounds__(RX_THREADS) __attribute__((amdgpu_waves_per_eu(8, 8))) k_rx_scatter(const RxPass A_kernarg) {
    HS_KERNARG(RxPass, A);
    __shared__ uint32_t whist[RX_WAVES][1 << RX_MAX_BITS];
    __shared__ int64_t gbase[1 << RX_MAX_BITS];
    __shared__ uint32_t s_wave_tot[4];
    extern __shared__ __align__(16) uint8_t rx_stage[];  // sbin[RX_TILE] u8, then stage[RX_TILE] of the widest column
    int64_t seg, t;
    if (!rx_find_tile(A.tile_base, A.n_seg, blockIdx.x, seg, t)) return;
    const int tid = threadIdx.x, lane = tid & (HS_WAVE - 1), w = tid / HS_WAVE, F = 1 << A.bits;
    for (int i = tid; i < RX_WAVES * F; i += RX_THREADS) whist[i / F][i % F] = 0;
    __syncthreads();
    const int64_t b = A.seg_start[seg] + t * RX_TILE;
    const int64_t seg_end = A.seg_start[seg + 1];
    const int64_t e = (b + RX_TILE) < seg_end ? (b + RX_TILE) : seg_end;
    // all loads of a phase are issued before the first dependent use: the kernel would otherwise pay one global
    // round trip per row step (the ranking's LDS updates and the scatter's stores fence the loads behind them)
    uint64_t word[RX_PER];
    uint32_t bin[RX_PER], local[RX_PER];
    const uint64_t below = (1ull << lane) - 1ull;
    const int64_t first = b + (int64_t)w * RX_SUB + lane;
#pragma unroll
    for (int j = 0; j < RX_PER; ++j) {
        const int64_t i = first + j * HS_WAVE;
        word[j] = i < e ? rx_key(A, i) : 0;
    }
#pragma unroll
    for (int j = 0; j < RX_PER; ++j) {
        const bool valid = first + j * HS_WAVE < e;
        bin[j] = valid ? rx_bin_of(A, word[j]) : 0u;
        uint64_t peers = __ballot(valid);
        for (int bit = 0; bit < A.bits; ++bit) {
            const bool on = (bin[j] >> bit) & 1u;
            const uint64_t bal = __ballot(valid && on);
            peers &= on ? bal : ~bal;
        }
        uint32_t prior = 0;
        if (valid) prior = whist[w][bin[j]];
        const uint32_t rank = (uint32_t)__popcll(peers & below);
        local[j] = prior + rank;
        if (valid && rank == 0) whist[w][bin[j]] = prior + (uint32_t)__popcll(peers);  // the bin's first lane of the step
    }
    __syncthreads();
    const int64_t nt = A.tile_base[seg + 1] - A.tile_base[seg];
    uint32_t bin_total = 0;
    if (tid < F) {
        uint32_t run = 0;
        for (int k = 0; k < RX_WAVES; ++k) {
            const uint32_t c = whist[k][tid];
            whist[k][tid] = run;
            run += c;
        }
        bin_total = run;
        gbase[tid] = A.counters[(A.tile_base[seg] << A.bits) + (int64_t)tid * nt + t];
    }
    {
        // tile-local start of every bin: exclusive scan of the bin totals (threads 0 .. 255 = waves 0 .. 3);
        // gbase becomes "global position minus tile-local position"
        uint32_t x = bin_total;
        for (int d = 1; d < HS_WAVE; d <<= 1) {
            const uint32_t up = __shfl_up(x, d, HS_WAVE);
            if (lane >= d) x += up;
        }
        if (w < 4 && lane == HS_WAVE - 1) s_wave_tot[w] = x;
        __syncthreads();
        if (tid < F) {
            uint32_t before = 0;
            for (int k = 0; k < w; ++k) before += s_wave_tot[k];
            const uint32_t bin_start = before + x - bin_total;
            gbase[tid] -= bin_start;
            for (int k = 0; k < RX_WAVES; ++k) whist[k][tid] += bin_start;  // now: tile-local start of (wave, bin)
        }
        __syncthreads();
        uint8_t* sbin = rx_stage;
        uint8_t* stage = rx_stage + RX_TILE;
        const int rows = (int)(e - b);
        uint32_t lpos2[RX_PER / 2];  // two 16-bit tile-local positions per register; 0xffff: no row
#pragma unroll
        for (int j = 0; j < RX_PER; ++j) {
            const bool valid = first + j * HS_WAVE < e;
            const uint32_t at = valid ? whist[w][bin[j]] + local[j] : 0xffffu;
            if (valid) sbin[at] = (uint8_t)bin[j];
            lpos2[j / 2] = (j & 1) ? (lpos2[j / 2] | (at << 16)) : at;
        }
        auto lpos = [&](int j) -> uint32_t { return (lpos2[j / 2] >> ((j & 1) * 16)) & 0xffffu; };
        auto write_out = [&](void* dst, int es) {
            __syncthreads();
#pragma unroll
            for (int k = 0; k < RX_PER; ++k) {
                const int i = tid + k * RX_THREADS;
                if (i >= rows) continue;
                const int64_t to = gbase[sbin[i]] + i;
                if (es == 4) ((uint32_t*)dst)[to] = ((const uint32_t*)stage)[i];
                else if (es == 8) ((uint64_t*)dst)[to] = ((const uint64_t*)stage)[i];
                else ((uint8_t*)dst)[to] = stage[i];
            }
            __syncthreads();
        };
        {   // the key column, from registers
            const int es = A.esize[0];
#pragma unroll
            for (int j = 0; j < RX_PER; ++j) {
                if (lpos(j) == 0xffffu) continue;
                if (es == 4) ((uint32_t*)stage)[lpos(j)] = (uint32_t)word[j];
                else ((uint64_t*)stage)[lpos(j)] = word[j];
            }
            write_out(A.dst[0], es);
        }
        for (int c = 1; c < A.n_cols; ++c) {
            const int es = A.esize[c];
            const void* src = A.src[c];
            uint64_t v[RX_PER];
#pragma unroll
            for (int j = 0; j < RX_PER; ++j) {
                const int64_t i = first + j * HS_WAVE;
                v[j] = lpos(j) == 0xffffu ? 0 : (es == 4 ? (uint64_t)((const uint32_t*)src)[i] : es == 8 ? ((const uint64_t*)src)[i] : (uint64_t)((const uint8_t*)src)[i]);
            }
#pragma unroll
            for (int j = 0; j < RX_PER; ++j) {
                if (lpos(j) == 0xffffu) continue;
                if (es == 4) ((uint32_t*)stage)[lpos(j)] = (uint32_t)v[j];
                else if (es == 8) ((uint64_t*)stage)[lpos(j)] = v[j];
                else stage[lpos(j)] = (uint8_t)v[j];
            }
            write_out(A.dst[c], es);
        }
    }
}

// ---- the pass for 4-byte tuples (round 3) --------------------------------------------------------------------------
// INTEGER key + up to three 4-byte value columns (f32 / i32: what SUM, AVG, COUNT over stored columns carry): the same
// tile, ranking and staging as k_rx_hist / k_rx_scatter with everything that was decided per element at run time
// (element size, first pass or later, selection, key kind) decided at compile time, tile-local 32-bit indexing and the
// 32-bit bin mix.  FIRST: the key is read from the table's INTEGER column at row0 + position, or through the row list.
template <bool FIRST>
__device__ __forceinline__ const int32_t* rx4_keys(const RxPass& A) {
    return FIRST ? (const int32_t*)A.key.data + A.row0 : (const int32_t*)A.src[0];
}

constexpr int RX_H4_THREADS = 256;                   // the histogram of a tile needs no ranking: four waves, 32 keys per lane
constexpr int RX_H4_PER = RX_TILE / RX_H4_THREADS;
template <bool FIRST>
__global__ void __launch_bounds__(RX_H4_THREADS) k_rx_hist4(const RxPass A_kernarg) {
    HS_KERNARG(RxPass, A);
    constexpr int WAVES = RX_H4_THREADS / HS_WAVE;
    __shared__ uint32_t hist[WAVES][1 << RX_MAX_BITS];
    int64_t seg, t;
    if (!rx_find_tile(A.tile_base, A.n_seg, blockIdx.x, seg, t)) return;
    const int tid = threadIdx.x, w = tid / HS_WAVE, F = 1 << A.bits;
    for (int i = tid; i < WAVES * F; i += RX_H4_THREADS) hist[i / F][i % F] = 0;
    const int64_t b = A.seg_start[seg] + t * RX_TILE;
    const int64_t left = A.seg_start[seg + 1] - b;
    const int rows = left < RX_TILE ? (int)left : RX_TILE;
    const int32_t* keys = rx4_keys<FIRST>(A) + b;
    const int64_t* sel = FIRST && A.sel ? A.sel + b : nullptr;  // behind a WHERE the first pass reads the key through the row list
    uint32_t key[RX_H4_PER];
#pragma unroll
    for (int j = 0; j < RX_H4_PER; ++j) {
        const int i = tid + j * RX_H4_THREADS;
        key[j] = i >= rows ? 0u : (uint32_t)(sel ? ((const int32_t*)A.key.data)[sel[i]] : keys[i]);
    }
    __syncthreads();
    const int shift = A.shift, bits = A.bits;
#pragma unroll
    for (int j = 0; j < RX_H4_PER; ++j)
        if (tid + j * RX_H4_THREADS < rows) atomicAdd(&hist[w][rx_bin4r(A, key[j], shift, bits)], 1u);
    __syncthreads();
    const int64_t nt = A.tile_base[seg + 1] - A.tile_base[seg];
    if (tid < F) {
        uint32_t total = 0;
#pragma unroll
        for (int k = 0; k < WAVES; ++k) total += hist[k][tid];
        A.counters[(A.tile_base[seg] << A.bits) + (int64_t)tid * nt + t] = total;
    }
}

template <int NV, bool FIRST>
__global__ void __launch_bounds__(RX_THREADS) __attribute__((amdgpu_waves_per_eu(8, 8))) k_rx_scatter4(const RxPass A_kernarg) {
    HS_KERNARG(RxPass, A);
    __shared__ uint32_t whist[RX_WAVES][1 << RX_MAX_BITS];
    __shared__ int64_t gbase[1 << RX_MAX_BITS];
    __shared__ uint32_t s_wave_tot[4];
    __shared__ uint8_t sbin[RX_TILE];
    __shared__ uint32_t stage[RX_TILE];
    int64_t seg, t;
    if (!rx_find_tile(A.tile_base, A.n_seg, blockIdx.x, seg, t)) return;
    const int tid = threadIdx.x, lane = tid & (HS_WAVE - 1), w = tid / HS_WAVE, F = 1 << A.bits;
    const int64_t b = A.seg_start[seg] + t * RX_TILE;
    const int64_t left = A.seg_start[seg + 1] - b;
    const int rows = left < RX_TILE ? (int)left : RX_TILE;
    const int32_t* keys = rx4_keys<FIRST>(A) + b;
    constexpr int NVR = NV > 0 ? NV : 1;
    uint32_t key[RX_PER], val[NVR][RX_PER];
    const int first = w * RX_SUB + lane;  // tile-local row of step 0
    const int64_t* sel = FIRST && A.sel ? A.sel + b : nullptr;
#pragma unroll
    for (int j = 0; j < RX_PER; ++j) {
        const int i = first + j * HS_WAVE;
        key[j] = i >= rows ? 0u : (uint32_t)(sel ? ((const int32_t*)A.key.data)[sel[i]] : keys[i]);
    }
#pragma unroll
    for (int c = 0; c < NV; ++c) {
        if (FIRST && A.src[1 + c] == nullptr) {  // (uniform) a first pass without this column: it carries the row's POSITION
#pragma unroll
            for (int j = 0; j < RX_PER; ++j) val[c][j] = (uint32_t)(b + first + j * HS_WAVE);
            continue;
        }
        const uint32_t* src = (const uint32_t*)A.src[1 + c] + b;
#pragma unroll
        for (int j = 0; j < RX_PER; ++j) val[c][j] = first + j * HS_WAVE < rows ? src[first + j * HS_WAVE] : 0u;
    }
    for (int i = tid; i < RX_WAVES * F; i += RX_THREADS) whist[i / F][i % F] = 0;
    __syncthreads();
    const int shift = A.shift, bits = A.bits;
    const uint64_t below = (1ull << lane) - 1ull;
    uint32_t bl[RX_PER];  // bin | rank inside (wave, bin) << 8
#pragma unroll
    for (int j = 0; j < RX_PER; ++j) {
        const bool valid = first + j * HS_WAVE < rows;
        const uint32_t bin = valid ? rx_bin4r(A, key[j], shift, bits) : 0u;
        uint64_t peers = __ballot(valid);
        for (int bit = 0; bit < bits; ++bit) {
            const bool on = (bin >> bit) & 1u;
            const uint64_t bal = __ballot(valid && on);
            peers &= on ? bal : ~bal;
        }
        const uint32_t prior = valid ? whist[w][bin] : 0u;
        const uint32_t rank = (uint32_t)__popcll(peers & below);
        bl[j] = bin | ((prior + rank) << 8);
        if (valid && rank == 0) whist[w][bin] = prior + (uint32_t)__popcll(peers);
    }
    __syncthreads();
    const int64_t nt = A.tile_base[seg + 1] - A.tile_base[seg];
    uint32_t bin_total = 0;
    if (tid < F) {
        uint32_t run = 0;
#pragma unroll
        for (int k = 0; k < RX_WAVES; ++k) {
            const uint32_t c = whist[k][tid];
            whist[k][tid] = run;
            run += c;
        }
        bin_total = run;
        gbase[tid] = A.counters[(A.tile_base[seg] << A.bits) + (int64_t)tid * nt + t];
    }
    uint32_t x = bin_total;
    for (int d = 1; d < HS_WAVE; d <<= 1) {
        const uint32_t up = __shfl_up(x, d, HS_WAVE);
        if (lane >= d) x += up;
    }
    if (w < 4 && lane == HS_WAVE - 1) s_wave_tot[w] = x;
    __syncthreads();
    if (tid < F) {
        const uint32_t before = (w > 0 ? s_wave_tot[0] : 0u) + (w > 1 ? s_wave_tot[1] : 0u) + (w > 2 ? s_wave_tot[2] : 0u);
        const uint32_t bin_start = before + x - bin_total;
        gbase[tid] -= bin_start;  // global position minus tile-local position
#pragma unroll
        for (int k = 0; k < RX_WAVES; ++k) whist[k][tid] += bin_start;
    }
    __syncthreads();
    uint32_t at[RX_PER];  // tile-local position of my rows in bin order
#pragma unroll
    for (int j = 0; j < RX_PER; ++j) {
        const bool valid = first + j * HS_WAVE < rows;
        at[j] = valid ? whist[w][bl[j] & 0xffu] + (bl[j] >> 8) : 0xffffffffu;
        if (valid) {
            sbin[at[j]] = (uint8_t)bl[j];
            stage[at[j]] = key[j];
        }
    }
    __syncthreads();
    {
        int32_t* dst = (int32_t*)A.dst[0];
#pragma unroll
        for (int k = 0; k < RX_PER; ++k) {
            const int i = tid + k * RX_THREADS;
            if (i < rows) dst[gbase[sbin[i]] + i] = (int32_t)stage[i];
        }
    }
#pragma unroll
    for (int c = 0; c < NV; ++c) {
        __syncthreads();
#pragma unroll
        for (int j = 0; j < RX_PER; ++j)
            if (at[j] != 0xffffffffu) stage[at[j]] = val[c][j];
        __syncthreads();
        uint32_t* dst = (uint32_t*)A.dst[1 + c];
#pragma unroll
        for (int k = 0; k < RX_PER; ++k) {
            const int i = tid + k * RX_THREADS;
            if (i < rows) dst[gbase[sbin[i]] + i] = stage[i];
        }
    }
}

// ---- the pass for wide STRING keys (round 3) -------------------------------------------------------------------------
// A STRING key of one fixed length 8 .. 16 bytes travels as KW = ceil(length / 4) four-byte word columns (columns 0 .. KW - 1;
// zero-padded; the value columns follow), so its tuples are 4-byte columns like an INTEGER key's and take the same tile
// shape, ranking and 32 KB staging at two workgroups per CU; the bin is cut from a mix of all KW words.  FIRST: the words
// come from the string column itself (any alignment: hs_str_words16), through the row list when there is one.
template <bool FIRST, int KW>
__device__ __forceinline__ void rx_words(const RxPass& A, int64_t pos, uint32_t* w) {
    if constexpr (FIRST) {
        const int64_t row = A.sel ? A.sel[pos] : A.row0 + pos;
        uint64_t w0, w1;
        hs_str_words16((const uint8_t*)A.key.data + row * A.key.fixed_len, (uint32_t)A.key.fixed_len, w0, w1);
        w[0] = (uint32_t)w0;
        w[1] = (uint32_t)(w0 >> 32);
        if constexpr (KW > 2) w[2] = (uint32_t)w1;
        if constexpr (KW > 3) w[3] = (uint32_t)(w1 >> 32);
    } else {
#pragma unroll
        for (int k = 0; k < KW; ++k) w[k] = ((const uint32_t*)A.src[k])[pos];
    }
}
template <int KW>
__device__ __forceinline__ uint32_t rx_binw(const uint32_t* w, int shift, int bits) {
    uint32_t h = rx_mix32(w[0]);
#pragma unroll
    for (int k = 1; k < KW; ++k) h = rx_mix32(h ^ w[k]);
    return (h >> shift) & ((1u << bits) - 1u);
}

template <bool FIRST, int KW>
__global__ void __launch_bounds__(RX_H4_THREADS) k_rx_histw(const RxPass A_kernarg) {
    HS_KERNARG(RxPass, A);
    constexpr int WAVES = RX_H4_THREADS / HS_WAVE;
    __shared__ uint32_t hist[WAVES][1 << RX_MAX_BITS];
    int64_t seg, t;
    if (!rx_find_tile(A.tile_base, A.n_seg, blockIdx.x, seg, t)) return;
    const int tid = threadIdx.x, w = tid / HS_WAVE, F = 1 << A.bits;
    for (int i = tid; i < WAVES * F; i += RX_H4_THREADS) hist[i / F][i % F] = 0;
    const int64_t b = A.seg_start[seg] + t * RX_TILE;
    const int64_t left = A.seg_start[seg + 1] - b;
    const int rows = left < RX_TILE ? (int)left : RX_TILE;
    const int shift = A.shift, bits = A.bits;
    __syncthreads();
    for (int j0 = 0; j0 < RX_H4_PER; j0 += 8) {
        uint32_t words[8][KW];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int i = tid + (j0 + j) * RX_H4_THREADS;
            if (i < rows) rx_words<FIRST, KW>(A, b + i, words[j]);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j)
            if (tid + (j0 + j) * RX_H4_THREADS < rows) atomicAdd(&hist[w][rx_binw<KW>(words[j], shift, bits)], 1u);
    }
    __syncthreads();
    const int64_t nt = A.tile_base[seg + 1] - A.tile_base[seg];
    if (tid < F) {
        uint32_t total = 0;
#pragma unroll
        for (int k = 0; k < WAVES; ++k) total += hist[k][tid];
        A.counters[(A.tile_base[seg] << A.bits) + (int64_t)tid * nt + t] = total;
    }
}

template <bool FIRST, int KW>
__global__ void __launch_bounds__(RX_THREADS) __attribute__((amdgpu_waves_per_eu(8, 8))) k_rx_scatterw(const RxPass A_kernarg) {
    HS_KERNARG(RxPass, A);
    __shared__ uint32_t whist[RX_WAVES][1 << RX_MAX_BITS];
    __shared__ int64_t gbase[1 << RX_MAX_BITS];
    __shared__ uint32_t s_wave_tot[4];
    __shared__ uint8_t sbin[RX_TILE];
    __shared__ uint32_t stage[RX_TILE];
    int64_t seg, t;
    if (!rx_find_tile(A.tile_base, A.n_seg, blockIdx.x, seg, t)) return;
    const int tid = threadIdx.x, lane = tid & (HS_WAVE - 1), w = tid / HS_WAVE, F = 1 << A.bits;
    const int64_t b = A.seg_start[seg] + t * RX_TILE;
    const int64_t left = A.seg_start[seg + 1] - b;
    const int rows = left < RX_TILE ? (int)left : RX_TILE;
    const int first = w * RX_SUB + lane;
    for (int i = tid; i < RX_WAVES * F; i += RX_THREADS) whist[i / F][i % F] = 0;
    const int shift = A.shift, bits = A.bits;
    uint32_t bin8[RX_PER];
    {
        uint32_t words[RX_PER][KW];
#pragma unroll
        for (int j = 0; j < RX_PER; ++j)
            if (first + j * HS_WAVE < rows) rx_words<FIRST, KW>(A, b + first + j * HS_WAVE, words[j]);
#pragma unroll
        for (int j = 0; j < RX_PER; ++j) bin8[j] = first + j * HS_WAVE < rows ? rx_binw<KW>(words[j], shift, bits) : 0u;
    }
    __syncthreads();
    const uint64_t below = (1ull << lane) - 1ull;
    uint32_t bl[RX_PER];  // bin | rank inside (wave, bin) << 8
#pragma unroll
    for (int j = 0; j < RX_PER; ++j) {
        const bool valid = first + j * HS_WAVE < rows;
        const uint32_t bin = bin8[j];
        uint64_t peers = __ballot(valid);
        for (int bit = 0; bit < bits; ++bit) {
            const bool on = (bin >> bit) & 1u;
            const uint64_t bal = __ballot(valid && on);
            peers &= on ? bal : ~bal;
        }
        const uint32_t prior = valid ? whist[w][bin] : 0u;
        const uint32_t rank = (uint32_t)__popcll(peers & below);
        bl[j] = bin | ((prior + rank) << 8);
        if (valid && rank == 0) whist[w][bin] = prior + (uint32_t)__popcll(peers);
    }
    __syncthreads();
    const int64_t nt = A.tile_base[seg + 1] - A.tile_base[seg];
    uint32_t bin_total = 0;
    if (tid < F) {
        uint32_t run = 0;
#pragma unroll
        for (int k = 0; k < RX_WAVES; ++k) {
            const uint32_t c = whist[k][tid];
            whist[k][tid] = run;
            run += c;
        }
        bin_total = run;
        gbase[tid] = A.counters[(A.tile_base[seg] << A.bits) + (int64_t)tid * nt + t];
    }
    uint32_t x = bin_total;
    for (int d = 1; d < HS_WAVE; d <<= 1) {
        const uint32_t up = __shfl_up(x, d, HS_WAVE);
        if (lane >= d) x += up;
    }
    if (w < 4 && lane == HS_WAVE - 1) s_wave_tot[w] = x;
    __syncthreads();
    if (tid < F) {
        const uint32_t before = (w > 0 ? s_wave_tot[0] : 0u) + (w > 1 ? s_wave_tot[1] : 0u) + (w > 2 ? s_wave_tot[2] : 0u);
        const uint32_t bin_start = before + x - bin_total;
        gbase[tid] -= bin_start;
#pragma unroll
        for (int k = 0; k < RX_WAVES; ++k) whist[k][tid] += bin_start;
    }
    __syncthreads();
    uint32_t at[RX_PER];
#pragma unroll
    for (int j = 0; j < RX_PER; ++j) {
        const bool valid = first + j * HS_WAVE < rows;
        at[j] = valid ? whist[w][bl[j] & 0xffu] + (bl[j] >> 8) : 0xffffffffu;
        if (valid) sbin[at[j]] = (uint8_t)bl[j];
    }
    // one column at a time through the 32 KB stage: key words first (FIRST: cut from the string again - the bytes are in L2),
    // then the value columns
    for (int c = 0; c < A.n_cols; ++c) {
        uint32_t v[RX_PER];
        if (FIRST && c < KW) {
#pragma unroll
            for (int j = 0; j < RX_PER; ++j) {
                uint32_t words[KW] = {};
                if (at[j] != 0xffffffffu) rx_words<FIRST, KW>(A, b + first + j * HS_WAVE, words);
                uint32_t pick = words[0];
#pragma unroll
                for (int k = 1; k < KW; ++k) pick = c == k ? words[k] : pick;
                v[j] = pick;
            }
        } else {
            const uint32_t* src = (const uint32_t*)A.src[c] + b;
#pragma unroll
            for (int j = 0; j < RX_PER; ++j) v[j] = at[j] != 0xffffffffu ? src[first + j * HS_WAVE] : 0u;
        }
        __syncthreads();  // the previous column has left the stage (first column: sbin is complete)
#pragma unroll
        for (int j = 0; j < RX_PER; ++j)
            if (at[j] != 0xffffffffu) stage[at[j]] = v[j];
        __syncthreads();
        uint32_t* dst = (uint32_t*)A.dst[c];
#pragma unroll
        for (int k = 0; k < RX_PER; ++k) {
            const int i = tid + k * RX_THREADS;
            if (i < rows) dst[gbase[sbin[i]] + i] = stage[i];
        }
    }
}

// the partition pass's output segments: (segment, bin) starts where the first tile's counter of that bin points
__global__ void __launch_bounds__(256) k_rx_next(const int64_t* seg_start, const int64_t* tile_base, int64_t n_seg, int32_t bits,
                                                 const int64_t* scanned, int64_t n, int64_t* out) {
    const int64_t total = n_seg << bits;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx <= total; idx += (int64_t)gridDim.x * blockDim.x) {
        if (idx == total) {
            out[idx] = n;
            continue;
        }
        const int64_t s = idx >> bits, bin = idx & ((1ll << bits) - 1);
        const int64_t nt = tile_base[s + 1] - tile_base[s];
        out[idx] = nt ? scanned[(tile_base[s] << bits) + bin * nt] : seg_start[s];
    }
}

// ---- the fold ---------------------------------------------------------------------------------------------------
struct RxAgg {
    const int64_t* seg_start;  // [n_parts + 1]
    int64_t n_parts;
    const void* src[RX_COLS];      // the partitioned tuples
    int32_t esize[RX_COLS];
    int32_t val_kind[HS_MAX_ACC];  // kind of the carried column behind aggregate a (HS_I32 / F32 / I64 / F64 / U8)
    int32_t carried[HS_MAX_ACC];   // aggregate a -> column 1.. of src, 0: the constant const_cell[a]
    uint64_t const_cell[HS_MAX_ACC];
    hs_agg_spec spec;
    int32_t cap, quantise;
    int32_t debug, last;           // last: no later launch takes what this one cannot hold
    uint64_t* prov_key;            // [n] a partition's groups, from the partition's own start
    uint64_t* prov_key1;           // [n] wide keys: the second word
    void* prov_acc[HS_MAX_ACC];    // [n] each: f32 / i32 when quantising, f64 / i64 bits otherwise
    int64_t* pcount;               // [n_parts] groups of the partition
    uint32_t* flags;
    const int64_t* list;           // second launch: the partitions the small table could not hold ...
    const int64_t* list_count;     // ... and their number (device)
    int64_t* overflow;             // first launch: where such partitions are noted
    int64_t* overflow_count;
};

__device__ __forceinline__ uint64_t rx_cell(const void* src, int kind, int64_t i) {
    switch (kind) {
        case HS_I32: return (uint64_t)(int64_t)((const int32_t*)src)[i];
        case HS_F32: return hs_d2u((double)((const float*)src)[i]);
        case HS_U8: return (uint64_t)((const uint8_t*)src)[i];
        default: return ((const uint64_t*)src)[i];  // HS_I64 / HS_F64: the cell itself
    }
}

__device__ __forceinline__ uint64_t rx_widen(uint64_t raw, int kind) {
    switch (kind) {
        case HS_I32: return (uint64_t)(int64_t)(int32_t)(uint32_t)raw;
        case HS_F32: return hs_d2u((double)__uint_as_float((uint32_t)raw));
        default: return raw;  // HS_I64 / HS_F64 cells, HS_U8 zero-extended
    }
}
__device__ __forceinline__ uint64_t rx_raw(const void* src, int esize, int64_t i) {
    return esize == 4 ? (uint64_t)((const uint32_t*)src)[i] : esize == 8 ? ((const uint64_t*)src)[i] : (uint64_t)((const uint8_t*)src)[i];
}

// HIPSPARK_RADIX_STAMPS=1: cycles (s_memtime) a wave spends per phase of k_rx_fold, summed over all waves
__device__ unsigned long long rx_stamp_acc[8];
#define RX_T(var) const long long var = A.debug ? (long long)clock64() : 0

constexpr int RX_CHUNK = 4;  // 64-row steps loaded together; the next chunk is in flight while this one is folded

// NC = value columns that travel with the rows (0 .. 4 specialised: the tuples of a chunk sit in registers;
// -1: any number, one global round trip per step and column).
//
// A wave's tables: keys[cap], acc[NA][cap] and order[cap] - the slots in the order their keys first appeared, which is
// also the order the partition's groups are written in.  The tables are cleared once per wave; after a partition only
// the slots on its list are reset, so a partition costs its rows and its groups, not the table size.
//
// Two launches share the kernel: the first with a SMALL table (512 slots: many waves per CU), where a partition that
// meets more keys than 3/4 of it is put on the overflow list instead of being finished; the second (A.list set) takes
// the listed partitions with the big table the plan sized for the worst case (every row its own group).
template <int NC, int SLOT_BITS>
__global__ void __launch_bounds__(256) k_rx_fold(const RxAgg A_kernarg) {
    HS_KERNARG(RxAgg, A);
    extern __shared__ __align__(16) uint64_t rx_lds[];
    const int lane = threadIdx.x & (HS_WAVE - 1), w = threadIdx.x / HS_WAVE, wpb = blockDim.x / HS_WAVE;
    const int NA = A.spec.n_acc, cap = A.cap;
    const size_t per_wave = (size_t)cap * (1 + NA) + (size_t)cap / 4;  // in 8-byte words; order[] is u16
    uint64_t* keys = rx_lds + (size_t)w * per_wave;  // [cap]
    uint64_t* acc = keys + cap;                       // [NA][cap]
    uint16_t* order = (uint16_t*)(acc + (size_t)NA * cap);
    const uint32_t mask = (uint32_t)cap - 1u;
    const int limit = A.last ? cap : cap - cap / 4;   // groups a partition may hold in this launch
    const uint64_t below = (1ull << lane) - 1ull;
    constexpr int NCR = NC > 0 ? NC : 1;
    uint32_t err = 0;
    long long t_init = 0, t_slot = 0, t_rank = 0, t_fold = 0, t_emit = 0, t_wait = 0;
    if ((int64_t)blockIdx.x * wpb + w >= (A.list ? *A.list_count : A.n_parts)) return;  // nothing for this wave: no table to clear
    for (int s = lane; s < cap; s += HS_WAVE) keys[s] = HS_EMPTY_KEY;
    for (int a = 0; a < NA; ++a) {
        const uint64_t id = hs_acc_identity(A.spec.op[a], A.spec.is_int[a] != 0);
        for (int s = lane; s < cap; s += HS_WAVE) acc[a * cap + s] = id;
    }
    const int64_t n_todo = A.list ? *A.list_count : A.n_parts;
    for (int64_t q = (int64_t)blockIdx.x * wpb + w; q < n_todo; q += (int64_t)gridDim.x * wpb) {
        const int64_t p = A.list ? A.list[q] : q;
        const int64_t b = A.seg_start[p], e = A.seg_start[p + 1];
        if (b >= e) {
            if (lane == 0) A.pcount[p] = 0;
            continue;
        }
        RX_T(c0);
        uint64_t nk[RX_CHUNK], nx[RX_CHUNK][NCR];
        auto load_chunk = [&](int64_t base) {
#pragma unroll
            for (int j = 0; j < RX_CHUNK; ++j) {
                const int64_t i = base + j * HS_WAVE + lane;
                const bool valid = i < e;
                nk[j] = !valid ? 0 : (A.esize[0] == 4 ? (uint64_t)(int64_t)((const int32_t*)A.src[0])[i] : ((const uint64_t*)A.src[0])[i]);
                if constexpr (NC > 0) {
#pragma unroll
                    for (int c = 0; c < NC; ++c) nx[j][c] = valid ? rx_raw(A.src[1 + c], A.esize[1 + c], i) : 0;
                }
            }
        };
        load_chunk(b);
        int ngroups = 0;
        bool full = false;
        RX_T(c1);
        t_init += c1 - c0;
        for (int64_t base = b; base < e && !full; base += RX_CHUNK * HS_WAVE) {
            RX_T(w0);
            uint64_t ck[RX_CHUNK], cx[RX_CHUNK][NCR];
#pragma unroll
            for (int j = 0; j < RX_CHUNK; ++j) {
                ck[j] = nk[j];
#pragma unroll
                for (int c = 0; c < NCR; ++c) cx[j][c] = NC > 0 ? nx[j][c] : 0;
            }
            if (A.debug) t_wait += (long long)(ck[0] & 1) * 0 + (long long)clock64() - w0;  // the chunk's loads have landed
            if (base + RX_CHUNK * HS_WAVE < e) load_chunk(base + RX_CHUNK * HS_WAVE);
#pragma unroll
            for (int j = 0; j < RX_CHUNK; ++j) {
                const int64_t i = base + j * HS_WAVE + lane;
                if (base + j * HS_WAVE >= e || full) break;
                const bool valid = i < e;
                RX_T(s0);
                int slot = 0;
                bool inserted = false;
                if (valid) {
                    const uint64_t k = ck[j];
                    uint32_t h = (uint32_t)(hs_mix64(k) >> 36) & mask;
                    slot = -1;
                    for (uint32_t probe = 0; probe <= mask; ++probe) {
                        uint64_t cur = __hip_atomic_load(&keys[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        if (cur == HS_EMPTY_KEY) {
                            cur = atomicCAS((unsigned long long*)&keys[h], (unsigned long long)HS_EMPTY_KEY, (unsigned long long)k);
                            if (cur == HS_EMPTY_KEY) {
                                cur = k;
                                inserted = true;
                            }
                        }
                        if (cur == k) {
                            slot = (int)h;
                            break;
                        }
                        h = (h + 1) & mask;
                    }
                }
                const uint64_t fresh = __ballot(inserted);
                if (inserted) order[ngroups + __popcll(fresh & below)] = (uint16_t)slot;
                ngroups += __popcll(fresh);
                if (ngroups > limit || __ballot(valid && slot < 0) != 0ull) {  // too many keys for this table
                    full = true;
                    break;
                }
                // lanes of this step in the same group (one ballot per slot bit, SLOT_BITS >= log2(cap)), my rank among
                // them: row order = lane order
                RX_T(s1);
                t_slot += s1 - s0;
                uint64_t peers = __ballot(valid);
#pragma unroll
                for (int bit = 0; bit < SLOT_BITS; ++bit) {
                    const bool on = (slot >> bit) & 1;
                    peers &= ~(__ballot(valid && on) ^ (on ? ~0ull : 0ull));
                }
                const int rank = __popcll(peers & below);
                const int most = __popcll(peers);
                uint64_t crowded = __ballot(valid && most > 1);
                RX_T(s2);
                t_rank += s2 - s1;
                for (int a = 0; a < NA; ++a) {
                    const uint32_t op = A.spec.op[a];
                    const bool is_int = A.spec.is_int[a] != 0;
                    const int c = A.carried[a];
                    uint64_t x = A.const_cell[a];
                    if (c) {
                        uint64_t raw = 0;
                        if constexpr (NC > 0) {
#pragma unroll
                            for (int cc = 0; cc < NC; ++cc)
                                if (c - 1 == cc) raw = cx[j][cc];
                        } else {
                            raw = valid ? rx_raw(A.src[c], A.esize[c], i) : 0;
                        }
                        x = rx_widen(raw, A.val_kind[a]);
                    }
                    // a group's first lane of the step folds the values of all its lanes, in lane order, from registers
                    uint64_t v = 0;
                    const bool leader = valid && rank == 0;
                    if (leader) v = hs_acc_fold(op, is_int, acc[a * cap + slot], x);
                    if (crowded) {
                        uint64_t rest = leader ? peers & ~(1ull << lane) : 0ull;
                        while (__ballot(rest != 0ull) != 0ull) {
                            const int from = rest ? __ffsll((long long)rest) - 1 : lane;
                            const uint64_t xv = __shfl(x, from, HS_WAVE);
                            if (rest) {
                                v = hs_acc_fold(op, is_int, v, xv);
                                rest &= rest - 1;
                            }
                        }
                    }
                    if (leader) acc[a * cap + slot] = v;
                }
                if (A.debug) t_fold += (long long)clock64() - s2;
            }
        }
        RX_T(e0);
        if (full) {
            // leave the partition to the launch with the big table (or, from that one, to the caller's other path)
            if (A.last) err |= HS_FLAG_DICT_FULL;
            else if (lane == 0) A.overflow[atomicAdd((unsigned long long*)A.overflow_count, 1ull)] = p;
            if (lane == 0) A.pcount[p] = 0;
        }
        // the partition's groups, in the order their keys first appeared, from the partition's own start
        for (int g0 = 0; g0 < ngroups; g0 += HS_WAVE) {
            const int g = g0 + lane;
            if (g >= ngroups) continue;
            const int s = order[g];
            const int64_t at = b + g;
            if (!full) A.prov_key[at] = keys[s];
            keys[s] = HS_EMPTY_KEY;
            for (int a = 0; a < NA; ++a) {
                const bool is_int = A.spec.is_int[a] != 0;
                uint64_t v = acc[a * cap + s];
                acc[a * cap + s] = hs_acc_identity(A.spec.op[a], is_int);
                if (full) continue;
                if (hs_float_identity_left(A.spec.op[a], is_int, v)) err |= HS_FLAG_TYPE_ASSERT;
                if (A.quantise) {
                    v = hs_quantise_cell(is_int, v, err);
                    if (is_int) ((int32_t*)A.prov_acc[a])[at] = (int32_t)(int64_t)v;
                    else ((float*)A.prov_acc[a])[at] = (float)hs_u2d(v);
                } else {
                    ((uint64_t*)A.prov_acc[a])[at] = v;
                }
            }
        }
        if (!full && lane == 0) A.pcount[p] = ngroups;
        if (A.debug) t_emit += (long long)clock64() - e0;
    }
    if (err) atomicOr(A.flags, err);
    if (A.debug && lane == 0) {
        atomicAdd(&rx_stamp_acc[0], (unsigned long long)t_init);
        atomicAdd(&rx_stamp_acc[1], (unsigned long long)t_wait);
        atomicAdd(&rx_stamp_acc[2], (unsigned long long)t_slot);
        atomicAdd(&rx_stamp_acc[3], (unsigned long long)t_rank);
        atomicAdd(&rx_stamp_acc[4], (unsigned long long)t_fold);
        atomicAdd(&rx_stamp_acc[5], (unsigned long long)t_emit);
        atomicAdd(&rx_stamp_acc[6], 1ull);
    }
}

// Cross-lane read-after-write through LDS inside ONE wave (the rank rounds and the wide key's second word below): the
// hardware runs a wave's LDS instructions in order, but the COMPILER must be told not to move the plain loads / stores
// across the hand-over - __builtin_amdgcn_wave_barrier alone is a scheduling barrier, not a memory fence.  Wavefront-scope
// fences cost no instruction (no s_waitcnt at this scope); they pin the order in the source instead of in today's codegen.
__device__ __forceinline__ void rx_wave_handover() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// ---- the fold, specialised for SUMs (round 3) ---------------------------------------------------------------------
// What SUM / AVG / COUNT lower to: every aggregate is a SUM whose argument is an f32 column (class 0: f64 cell), an i32
// column (class 1: i64 cell) or an integer constant (class 2: COUNT).  Same tables, same launches (small table first,
// overflow list, big table) and the same order of additions as k_rx_fold - a group's f64 cell takes its rows' values in
// row order - at about a quarter of the instructions: the classes are compile-time (no per-value switches on kind and
// operator), the dictionary probe is the compare-and-swap itself, values travel as the 32 bits they are, and the rows of
// a step that share a group are folded in RANK ROUNDS - round r: the lanes whose rank among their group's lanes is r add
// their value to the group's cell in LDS (distinct cells within a round; LDS executes a wave's instructions in order, so
// round r + 1 reads what round r wrote) - instead of a leader lane collecting its peers' values with shuffles.  Integer
// cells are order-free: one LDS atomic per lane (COUNT: one per group and step, with the group's lane count).
// KM: 0 = 4-byte key column, 1 = 8-byte key words, 2 .. 4 = a wide STRING key in that many 4-byte word columns (compared as
// two 64-bit words; the value columns follow the key's)
template <int NA, int CLS, int KM>
__global__ void __launch_bounds__(256) k_rx_fold_sum(const RxAgg A_kernarg) {
    constexpr bool WIDE = KM >= 2, KEY4 = KM == 0 || WIDE;
    constexpr int KW = WIDE ? KM : 1;
    HS_KERNARG(RxAgg, A);
    extern __shared__ __align__(16) uint64_t rx_lds[];
    const int lane = threadIdx.x & (HS_WAVE - 1), w = threadIdx.x / HS_WAVE, wpb = blockDim.x / HS_WAVE;
    const int cap = A.cap;
    const size_t per_wave = (size_t)cap * (1 + (WIDE ? 1 : 0) + NA) + (size_t)cap / 4;  // the host sizes LDS by the same rule
    uint64_t* keys = rx_lds + (size_t)w * per_wave;
    uint64_t* keys1 = keys + cap;  // WIDE: the second key word of every slot
    uint64_t* acc = keys + (WIDE ? 2 : 1) * cap;
    uint16_t* order = (uint16_t*)(acc + (size_t)NA * cap);
    const uint32_t mask = (uint32_t)cap - 1u;
    const int limit = A.last ? cap : cap - cap / 4;
    const uint64_t below = (1ull << lane) - 1ull;
    constexpr auto cls = [](int a) { return (CLS >> (2 * a)) & 3; };
    constexpr int NC = (NA > 0 && cls(0) < 2 ? 1 : 0) + (NA > 1 && cls(1) < 2 ? 1 : 0) + (NA > 2 && cls(2) < 2 ? 1 : 0);
    constexpr int NCR = NC > 0 ? NC : 1;
    constexpr auto col_of = [](int a) {  // carried columns are numbered in aggregate order (hs_group_radix_run)
        int c = 0;
        for (int k = 0; k < a; ++k) c += ((CLS >> (2 * k)) & 3) < 2 ? 1 : 0;
        return c;
    };
    constexpr bool any_float = (NA > 0 && cls(0) == 0) || (NA > 1 && cls(1) == 0) || (NA > 2 && cls(2) == 0);
    int slot_bits = 0;
    while ((1 << slot_bits) < cap) ++slot_bits;
    uint32_t err = 0;
    const int64_t n_todo = A.list ? *A.list_count : A.n_parts;
    if ((int64_t)blockIdx.x * wpb + w >= n_todo) return;  // nothing listed for this wave: no table to clear either
    for (int s = lane; s < cap; s += HS_WAVE) keys[s] = HS_EMPTY_KEY;
    for (int s = lane; s < NA * cap; s += HS_WAVE) acc[s] = 0;  // the identity of SUM, f64 and i64 alike
    for (int64_t q = (int64_t)blockIdx.x * wpb + w; q < n_todo; q += (int64_t)gridDim.x * wpb) {
        const int64_t p = A.list ? A.list[q] : q;
        const int64_t b = A.seg_start[p], e = A.seg_start[p + 1];
        if (b >= e) {
            if (lane == 0) A.pcount[p] = 0;
            continue;
        }
        using KeyReg = std::conditional_t<KEY4, uint32_t, uint64_t>;  // as loaded: widening a key right after its load would wait for it
        KeyReg nk[RX_CHUNK];
        uint32_t nw[RX_CHUNK][KW];  // WIDE: key words 1 .. KW - 1 ([0] unused)
        uint32_t nx[RX_CHUNK][NCR];
        auto load_chunk = [&](int64_t base) {
#pragma unroll
            for (int j = 0; j < RX_CHUNK; ++j) {
                const int64_t i = base + j * HS_WAVE + lane;
                const bool valid = i < e;
                nk[j] = valid ? ((const KeyReg*)A.src[0])[i] : 0;
#pragma unroll
                for (int q = 1; q < KW; ++q) nw[j][q] = valid ? ((const uint32_t*)A.src[q])[i] : 0u;
#pragma unroll
                for (int c = 0; c < NC; ++c) nx[j][c] = valid ? ((const uint32_t*)A.src[KW + c])[i] : 0u;
            }
        };
        load_chunk(b);
        int ngroups = 0;
        bool full = false;
        for (int64_t base = b; base < e && !full; base += RX_CHUNK * HS_WAVE) {
            KeyReg ck[RX_CHUNK];
            uint32_t cw[RX_CHUNK][KW];
            uint32_t cx[RX_CHUNK][NCR];
#pragma unroll
            for (int j = 0; j < RX_CHUNK; ++j) {
                ck[j] = nk[j];
#pragma unroll
                for (int q = 1; q < KW; ++q) cw[j][q] = nw[j][q];
#pragma unroll
                for (int c = 0; c < NCR; ++c) cx[j][c] = NC > 0 ? nx[j][c] : 0u;
            }
            if (base + RX_CHUNK * HS_WAVE < e) load_chunk(base + RX_CHUNK * HS_WAVE);
#pragma unroll
            for (int j = 0; j < RX_CHUNK; ++j) {
                if (base + j * HS_WAVE >= e || full) break;
                const bool valid = base + j * HS_WAVE + lane < e;
                int slot = valid ? -1 : 0;
                bool inserted = false;
                if (valid) {
                    uint64_t k, k1 = 0;
                    if constexpr (WIDE) {
                        k = (uint64_t)ck[j] | ((uint64_t)cw[j][1] << 32);
                        if constexpr (KW > 2) k1 = (uint64_t)cw[j][2];
                        if constexpr (KW > 3) k1 |= (uint64_t)cw[j][3] << 32;
                    } else {
                        k = KEY4 ? (uint64_t)(int64_t)(int32_t)ck[j] : (uint64_t)ck[j];
                    }
                    uint32_t h;
                    if constexpr (KEY4 && !WIDE) {  // two 32-bit multiplies; bits independent of the ones the partitions were cut on
                        uint32_t m = (uint32_t)k * 0xCC9E2D51u;
                        m ^= m >> 17;
                        h = ((m * 0x1B873593u) >> 12) & mask;
                    } else if constexpr (WIDE) {
                        h = (uint32_t)(hs_mix64(k ^ hs_mix64(k1)) >> 36) & mask;
                    } else {
                        h = (uint32_t)(hs_mix64(k) >> 36) & mask;
                    }
                    for (uint32_t probe = 0; probe <= mask; ++probe) {
                        const uint64_t cur = atomicCAS((unsigned long long*)&keys[h], (unsigned long long)HS_EMPTY_KEY, (unsigned long long)k);
                        if constexpr (WIDE) {
                            // the lane that claimed the slot completes it before any lane of the wave - this probe round or a
                            // later one - compares the second word (LDS executes the wave's instructions in order)
                            if (cur == HS_EMPTY_KEY) keys1[h] = k1;
                            rx_wave_handover();
                            if (cur == HS_EMPTY_KEY || (cur == k && keys1[h] == k1)) {
                                inserted = cur == HS_EMPTY_KEY;
                                slot = (int)h;
                                break;
                            }
                        } else if (cur == HS_EMPTY_KEY || cur == k) {
                            inserted = cur == HS_EMPTY_KEY;
                            slot = (int)h;
                            break;
                        }
                        h = (h + 1) & mask;
                    }
                }
                const uint64_t fresh = __ballot(inserted);
                if (inserted) order[ngroups + __popcll(fresh & below)] = (uint16_t)slot;
                ngroups += __popcll(fresh);
                if (ngroups > limit || __ballot(valid && slot < 0) != 0ull) {
                    full = true;
                    break;
                }
                uint64_t peers = __ballot(valid);
                for (int bit = 0; bit < slot_bits; ++bit) {
                    const bool on = (slot >> bit) & 1;
                    peers &= ~(__ballot(valid && on) ^ (on ? ~0ull : 0ull));
                }
                const int rank = valid ? __popcll(peers & below) : -1;
#pragma unroll
                for (int a = 0; a < NA; ++a) {  // integer cells: any order
                    if (cls(a) == 1) {
                        if (valid) atomicAdd((unsigned long long*)&acc[a * cap + slot], (unsigned long long)(int64_t)(int32_t)cx[j][col_of(a)]);
                    } else if (cls(a) == 2) {
                        if (rank == 0) atomicAdd((unsigned long long*)&acc[a * cap + slot], A.const_cell[a] * (uint64_t)__popcll(peers));
                    }
                }
                if constexpr (any_float) {
                    for (int r = 0; __ballot(rank == r) != 0ull; ++r) {
                        if (rank == r) {
#pragma unroll
                            for (int a = 0; a < NA; ++a) {
                                if (cls(a) != 0) continue;
                                uint64_t* cell = &acc[a * cap + slot];
                                *cell = hs_d2u(hs_u2d(*cell) + (double)__uint_as_float(cx[j][col_of(a)]));
                            }
                        }
                        rx_wave_handover();
                    }
                }
            }
        }
        if (full) {
            if (A.last) err |= HS_FLAG_DICT_FULL;
            else if (lane == 0) A.overflow[atomicAdd((unsigned long long*)A.overflow_count, 1ull)] = p;
            if (lane == 0) A.pcount[p] = 0;
        }
        for (int g0 = 0; g0 < ngroups; g0 += HS_WAVE) {
            const int g = g0 + lane;
            if (g >= ngroups) continue;
            const int s = order[g];
            const int64_t at = b + g;
            if (!full) A.prov_key[at] = keys[s];
            if (WIDE && !full) A.prov_key1[at] = keys1[s];
            keys[s] = HS_EMPTY_KEY;
#pragma unroll
            for (int a = 0; a < NA; ++a) {
                const bool is_int = cls(a) != 0;
                uint64_t v = acc[a * cap + s];
                acc[a * cap + s] = 0;
                if (full) continue;
                if (A.quantise) {
                    v = hs_quantise_cell(is_int, v, err);
                    if (is_int) ((int32_t*)A.prov_acc[a])[at] = (int32_t)(int64_t)v;
                    else ((float*)A.prov_acc[a])[at] = (float)hs_u2d(v);
                } else {
                    ((uint64_t*)A.prov_acc[a])[at] = v;
                }
            }
        }
        if (!full && lane == 0) A.pcount[p] = ngroups;
    }
    if (err) atomicOr(A.flags, err);
}

// ---- dense output -----------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_rx_unit_groups(const int64_t* pscan, int64_t parts_per_unit, int32_t n_units,
                                                        int64_t* out) {
    const int u = blockIdx.x * blockDim.x + threadIdx.x;
    if (u <= n_units) out[u] = pscan[(int64_t)u * parts_per_unit];
}

struct RxEmit {
    const int64_t* seg_start;
    const int64_t* pscan;
    int64_t n_parts;
    const uint64_t* prov_key;
    const uint64_t* prov_key1;
    const void* prov_acc[HS_MAX_ACC];
    int32_t n_acc, osize, key_kind, pad;
    void* out_key;
    void* out_acc[HS_MAX_ACC];
};
__global__ void __launch_bounds__(256) k_rx_emit(const RxEmit A_kernarg) {
    HS_KERNARG(RxEmit, A);
    const int lane = threadIdx.x & (HS_WAVE - 1), wpb = blockDim.x / HS_WAVE;
    for (int64_t p = (int64_t)blockIdx.x * wpb + threadIdx.x / HS_WAVE; p < A.n_parts; p += (int64_t)gridDim.x * wpb) {
        const int64_t from = A.seg_start[p], to = A.pscan[p], cnt = A.pscan[p + 1] - to;
        for (int64_t q = lane; q < cnt; q += HS_WAVE) {
            const uint64_t k = A.prov_key[from + q];
            const int base = A.key_kind & 0xff;
            if (base == HS_I32) ((int32_t*)A.out_key)[to + q] = (int32_t)(int64_t)k;
            else if (base == HS_F32) ((float*)A.out_key)[to + q] = (float)hs_u2d(k);  // the word: the f32 widened, exactly
            else if (base == HS_STR) {
                const int len = A.key_kind >> 8;  // hs_pack_str / hs_str_words16: byte i of the string = byte i of the word(s)
                const uint64_t k1 = len > 8 ? A.prov_key1[from + q] : 0ull;
                for (int i = 0; i < len; ++i)
                    ((uint8_t*)A.out_key)[(to + q) * len + i] = (uint8_t)(i < 8 ? k >> (8 * i) : k1 >> (8 * (i - 8)));
            } else ((uint64_t*)A.out_key)[to + q] = k;
            for (int a = 0; a < A.n_acc; ++a) rx_move(A.prov_acc[a], A.out_acc[a], A.osize, from + q, to + q);
        }
    }
}

// ---- host -------------------------------------------------------------------------------------------------------
static int rx_esize(int32_t kind) {
    switch (kind) {
        case HS_I32:
        case HS_F32: return 4;
        case HS_I64:
        case HS_F64: return 8;
        case HS_U8: return 1;
        default: return 0;
    }
}
static size_t rx_align(size_t x) { return (x + 255) & ~(size_t)255; }

// key_kind of a plan: a STRING key of one fixed length 8 .. 16 bytes = two key words per tuple
static int rx_plan_wide(int64_t key_kind) {  // -> number of 4-byte key word columns (2 .. 4), 0: not a wide key
    return (key_kind & 0xff) == HS_STR && (key_kind >> 8) > 7 ? (int)(((key_kind >> 8) + 3) / 4) : 0;
}

// field use of the public plan (include/hipspark.h keeps it opaque: int64 f[48])
enum {
    PL_N, PL_UNITS, PL_BITS1, PL_BITS2, PL_CAP, PL_NA, PL_NCARRIED, PL_QUANTISE, PL_KEYKIND, PL_NSEG1, PL_PARTS, PL_TILES1,
    PL_TILES2, PL_COUNTERS, PL_OSIZE, PL_WS, PL_OFF_BUF_A, PL_OFF_BUF_B, PL_OFF_SEG1, PL_OFF_SEG2, PL_OFF_TB0, PL_OFF_TB1,
    PL_OFF_CNT, PL_OFF_SCAN, PL_OFF_SCANWS, PL_OFF_PKEY, PL_OFF_PACC, PL_OFF_PCOUNT, PL_OFF_PSCAN, PL_OFF_OVERFLOW, PL_TUPLE, PL_ESIZE0 /* .. +16 */
};

// one partition pass over the segments of P: tiles, histogram, scan, scatter, the next level's segments
static int rx_pass(hipStream_t stream, RxPass& P, int64_t max_tiles, int64_t* counters, int64_t* scanned, void* scan_ws, int64_t n,
                   int64_t* next_seg) {
    hipLaunchKernelGGL(k_rx_tiles, dim3(1), dim3(1024), 0, stream, P.seg_start, P.n_seg, (int64_t*)P.tile_base);
    RX_CHECK_LAUNCH("radix pass (tiles)");
    const int64_t ncnt = max_tiles << P.bits;
    hs_memset_async(counters, 0, (size_t)ncnt * 8, stream);
    P.counters = counters;
    // 4-byte tuples (INTEGER key, f32 / i32 values): the specialised kernels
    bool four = !P.raw && P.n_cols <= 4 && (P.first ? P.key.kind == HS_I32 : P.esize[0] == 4);
    for (int c = 1; c < P.n_cols; ++c) four = four && P.esize[c] == 4;
    P.key4 = four ? 1 : 0;
#define RX_WIDE(KERNEL, THREADS)                                                                                          \
    switch (P.wide * 2 + (P.first ? 1 : 0)) {                                                                             \
        case 4: hipLaunchKernelGGL((KERNEL<false, 2>), dim3((unsigned)max_tiles), dim3(THREADS), 0, stream, P); break;    \
        case 5: hipLaunchKernelGGL((KERNEL<true, 2>), dim3((unsigned)max_tiles), dim3(THREADS), 0, stream, P); break;     \
        case 6: hipLaunchKernelGGL((KERNEL<false, 3>), dim3((unsigned)max_tiles), dim3(THREADS), 0, stream, P); break;    \
        case 7: hipLaunchKernelGGL((KERNEL<true, 3>), dim3((unsigned)max_tiles), dim3(THREADS), 0, stream, P); break;     \
        case 8: hipLaunchKernelGGL((KERNEL<false, 4>), dim3((unsigned)max_tiles), dim3(THREADS), 0, stream, P); break;    \
        default: hipLaunchKernelGGL((KERNEL<true, 4>), dim3((unsigned)max_tiles), dim3(THREADS), 0, stream, P); break;    \
    }
    if (P.wide) {
        RX_WIDE(k_rx_histw, RX_H4_THREADS)
    } else if (four) {
        if (P.first) hipLaunchKernelGGL(k_rx_hist4<true>, dim3((unsigned)max_tiles), dim3(RX_H4_THREADS), 0, stream, P);
        else hipLaunchKernelGGL(k_rx_hist4<false>, dim3((unsigned)max_tiles), dim3(RX_H4_THREADS), 0, stream, P);
    } else {
        hipLaunchKernelGGL(k_rx_hist, dim3((unsigned)max_tiles), dim3(RX_THREADS), 0, stream, P);
    }
    RX_CHECK_LAUNCH("radix pass (histogram)");
    const int rc = hs_exclusive_scan_i64(stream, counters, ncnt, scanned, scan_ws);
    if (rc != HS_OK) return rc;
    P.counters = scanned;
    int widest = 1;
    for (int c = 0; c < P.n_cols; ++c) widest = P.esize[c] > widest ? P.esize[c] : widest;
    static unsigned long long attr_set = 0;
    if (hs_first_on_device(attr_set))  // 8-byte columns stage 72 KB + 18 KB static: above the 64 KB a launch gets unasked
        (void)hipFuncSetAttribute((const void*)k_rx_scatter, hipFuncAttributeMaxDynamicSharedMemorySize, RX_TILE * 9);
    if (P.wide) {
        RX_WIDE(k_rx_scatterw, RX_THREADS)
    } else if (four) {
#define RX_S4(NV)                                                                                                         \
    if (P.first) hipLaunchKernelGGL((k_rx_scatter4<NV, true>), dim3((unsigned)max_tiles), dim3(RX_THREADS), 0, stream, P); \
    else hipLaunchKernelGGL((k_rx_scatter4<NV, false>), dim3((unsigned)max_tiles), dim3(RX_THREADS), 0, stream, P)
        switch (P.n_cols - 1) {
            case 0: RX_S4(0); break;
            case 1: RX_S4(1); break;
            case 2: RX_S4(2); break;
            default: RX_S4(3); break;
        }
#undef RX_S4
#undef RX_WIDE
    } else hipLaunchKernelGGL(k_rx_scatter, dim3((unsigned)max_tiles), dim3(RX_THREADS), (size_t)RX_TILE * (1 + widest), stream, P);
    RX_CHECK_LAUNCH("radix pass (scatter)");
    if (next_seg) {
        const int64_t nout = (P.n_seg << P.bits) + 1;
        hipLaunchKernelGGL(k_rx_next, dim3((unsigned)((nout + 255) / 256 > 4096 ? 4096 : (nout + 255) / 256)), dim3(256), 0, stream,
                           P.seg_start, P.tile_base, P.n_seg, (int32_t)P.bits, (const int64_t*)scanned, n, next_seg);
        RX_CHECK_LAUNCH("radix pass (segments)");
    }
    return HS_OK;
}

extern "C" int hs_group_radix_plan(int32_t key_kind, int64_t n, int32_t n_units, int64_t max_unit_rows, const int32_t* val_kinds,
                                   const hs_agg_spec* spec, int32_t quantise, hs_radix_plan* plan) {
    if (!plan || !spec || n < 1 || n_units < 1 || max_unit_rows < 1 || spec->n_acc < 0 || spec->n_acc > HS_MAX_ACC ||
        (spec->n_acc > 0 && !val_kinds)) {
        hs_set_error("hs_group_radix_plan: bad arguments");
        return HS_E_ARG;
    }
    {   // keys whose 64-bit key word IS the key: integers, floats (0.0 == -0.0, as one Python dict key), strings of a fixed
        // length <= 7 bytes (key_kind = HS_STR + 256 x length)
        const int base = key_kind & 0xff, len = key_kind >> 8;
        const bool ok = base == HS_STR ? (len >= 1 && len <= 16) : (len == 0 && (base == HS_I32 || base == HS_I64 || base == HS_F32 || base == HS_F64));
        if (!ok) {
            hs_set_error("hs_group_radix_plan: key kind %d (HS_I32 / I64 / F32 / F64, or HS_STR + 256 x fixed length <= 16)", (int)key_kind);
            return HS_E_LIMIT;
        }
    }
    // strings of 8 .. 16 bytes travel as 2 .. 4 four-byte key word columns and are compared on all of them: exact, no
    // hashing.  Only the SUM-specialised fold knows them: SUM / AVG / COUNT over f32 / i32 columns and integer constants,
    // <= 3 aggregates
    const int wide = rx_plan_wide(key_kind);
    if (wide) {
        bool sums = spec->n_acc >= 1 && spec->n_acc <= 3;
        for (int a = 0; a < spec->n_acc && sums; ++a) {
            const bool is_int = spec->is_int[a] != 0;
            sums = spec->op[a] == HS_AGG_SUM && ((val_kinds[a] == HS_F32 && !is_int) || (val_kinds[a] == HS_I32 && is_int) || (val_kinds[a] < 0 && is_int));
        }
        if (!sums) {
            hs_set_error("hs_group_radix_plan: a STRING key of 8 .. 16 bytes needs SUM / COUNT aggregates over f32 / i32 columns (<= 3)");
            return HS_E_LIMIT;
        }
    }
    int64_t* f = plan->f;
    for (int i = 0; i < 48; ++i) f[i] = 0;
    const int NA = spec->n_acc;
    int cap = 2048;
    while (cap > 64 && ((size_t)cap * (1 + (wide ? 1 : 0) + NA) + (size_t)cap / 4) * 8 > 65536) cap >>= 1;
    // rows per partition: half the slots of the LARGE table.  Partitions that turn out to hold few distinct keys - the
    // usual case - are folded by the 512-slot launch anyway.  (Cutting down to half the slots of the SMALL table where two
    // passes can was measured at 600 M rows / 287 units: all-distinct keys 2.2x faster - no partition needs the large-table
    // launch -, but ~100 rows per group 65 % slower - a finer cut puts more rows of a group into every 64-row step, and
    // those fold one after the other - and 4 rows per group 10 % slower: not taken.)
    int bits = 0;
    while (bits < 2 * RX_MAX_BITS && ((int64_t)(cap / 2) << bits) < max_unit_rows) ++bits;
    const int bits1 = bits <= RX_MAX_BITS ? bits : (bits + 1) / 2, bits2 = bits - bits1;
    f[PL_N] = n;
    f[PL_UNITS] = n_units;
    f[PL_BITS1] = bits1;
    f[PL_BITS2] = bits2;
    f[PL_CAP] = cap;
    f[PL_NA] = NA;
    f[PL_QUANTISE] = quantise ? 1 : 0;
    f[PL_KEYKIND] = key_kind;
    f[PL_NSEG1] = (int64_t)n_units << bits1;
    f[PL_PARTS] = (int64_t)n_units << bits;
    f[PL_TILES1] = n / RX_TILE + n_units + 1;
    f[PL_TILES2] = bits2 ? n / RX_TILE + f[PL_NSEG1] + 1 : 0;
    const int64_t c1 = f[PL_TILES1] << bits1, c2 = f[PL_TILES2] << bits2;
    f[PL_COUNTERS] = c1 > c2 ? c1 : c2;
    f[PL_OSIZE] = quantise ? 4 : 8;
    f[PL_ESIZE0] = key_kind == HS_I32 || wide ? 4 : 8;
    const int kcols = wide ? wide : 1;  // key columns of a tuple; the value columns follow
    for (int q = 1; q < kcols; ++q) f[PL_ESIZE0 + q] = 4;
    int64_t tuple = f[PL_ESIZE0] * kcols;
    int carried = 0;
    for (int a = 0; a < NA; ++a) {
        if (val_kinds[a] < 0) continue;  // a constant: does not travel
        const int es = rx_esize(val_kinds[a]);
        if (!es) {
            hs_set_error("hs_group_radix_plan: value column %d has kind %d", a, (int)val_kinds[a]);
            return HS_E_ARG;
        }
        f[PL_ESIZE0 + kcols + carried] = es;
        tuple += es;
        ++carried;
    }
    f[PL_NCARRIED] = carried;
    f[PL_TUPLE] = tuple;
    size_t off = 0;
    auto take = [&](int field, size_t bytes) {
        f[field] = (int64_t)off;
        off += rx_align(bytes);
    };
    // a buffer set = the columns back to back, each n elements
    size_t set_bytes = 0;
    for (int c = 0; c < kcols + carried; ++c) set_bytes += rx_align((size_t)n * f[PL_ESIZE0 + c]);
    take(PL_OFF_BUF_A, set_bytes);
    take(PL_OFF_BUF_B, bits2 ? set_bytes : 0);
    take(PL_OFF_SEG1, (size_t)(f[PL_NSEG1] + 1) * 8);
    take(PL_OFF_SEG2, bits2 ? (size_t)(f[PL_PARTS] + 1) * 8 : 0);
    take(PL_OFF_TB0, (size_t)(n_units + 1) * 8);
    take(PL_OFF_TB1, bits2 ? (size_t)(f[PL_NSEG1] + 1) * 8 : 0);
    take(PL_OFF_CNT, (size_t)f[PL_COUNTERS] * 8);
    take(PL_OFF_SCAN, (size_t)(f[PL_COUNTERS] + 1) * 8);
    const int64_t scan_n = f[PL_COUNTERS] > f[PL_PARTS] ? f[PL_COUNTERS] : f[PL_PARTS];
    take(PL_OFF_SCANWS, hs_scan_ws_bytes(scan_n));
    take(PL_OFF_PKEY, (size_t)n * 8 * (wide ? 2 : 1));
    take(PL_OFF_PACC, (size_t)NA * rx_align((size_t)n * f[PL_OSIZE]));
    take(PL_OFF_PCOUNT, (size_t)f[PL_PARTS] * 8);
    take(PL_OFF_PSCAN, (size_t)(f[PL_PARTS] + 1) * 8);
    take(PL_OFF_OVERFLOW, (size_t)(f[PL_PARTS] + 1) * 16);  // two lists: past the 256-slot tables, past the 512-slot tables
    f[PL_WS] = (int64_t)off;
    return HS_OK;
}

extern "C" size_t hs_group_radix_ws_bytes(const hs_radix_plan* plan) { return plan ? (size_t)plan->f[PL_WS] : 0; }

static void rx_set_cols(const int64_t* f, uint8_t* ws, int field, void** cols) {
    size_t off = (size_t)f[field];
    for (int c = 0; c < (rx_plan_wide(f[PL_KEYKIND]) ? rx_plan_wide(f[PL_KEYKIND]) : 1) + (int)f[PL_NCARRIED]; ++c) {
        cols[c] = ws + off;
        off += rx_align((size_t)f[PL_N] * f[PL_ESIZE0 + c]);
    }
}

extern "C" int hs_group_radix_run(void* stream_, const hs_radix_plan* plan, const hs_col* key, const int64_t* sel, int64_t row0,
                                  const int64_t* unit_bounds, const hs_col* val_cols, const uint64_t* const_cells,
                                  const hs_agg_spec* spec, void* ws_, int64_t* out_unit_groups, uint32_t* flags) {
    if (!plan || !key || !unit_bounds || !spec || !ws_ || !out_unit_groups || !flags || spec->n_acc != (int)plan->f[PL_NA] ||
        key->kind != (int32_t)(plan->f[PL_KEYKIND] & 0xff) || (key->kind == HS_STR && key->fixed_len != (int32_t)(plan->f[PL_KEYKIND] >> 8)) ||
        (spec->n_acc > 0 && (!val_cols || !const_cells))) {
        hs_set_error("hs_group_radix_run: bad arguments");
        return HS_E_ARG;
    }
    const int64_t* f = plan->f;
    hipStream_t stream = (hipStream_t)stream_;
    uint8_t* ws = (uint8_t*)ws_;
    const int NA = (int)f[PL_NA], bits1 = (int)f[PL_BITS1], bits2 = (int)f[PL_BITS2], n_units = (int)f[PL_UNITS];
    const int64_t n = f[PL_N];
    void* buf_a[RX_COLS];
    void* buf_b[RX_COLS];
    rx_set_cols(f, ws, PL_OFF_BUF_A, buf_a);
    if (bits2) rx_set_cols(f, ws, PL_OFF_BUF_B, buf_b);
    int64_t* seg1 = (int64_t*)(ws + f[PL_OFF_SEG1]);
    int64_t* seg2 = bits2 ? (int64_t*)(ws + f[PL_OFF_SEG2]) : seg1;
    int64_t* tb0 = (int64_t*)(ws + f[PL_OFF_TB0]);
    int64_t* tb1 = (int64_t*)(ws + f[PL_OFF_TB1]);
    int64_t* counters = (int64_t*)(ws + f[PL_OFF_CNT]);
    int64_t* scanned = (int64_t*)(ws + f[PL_OFF_SCAN]);
    void* scan_ws = ws + f[PL_OFF_SCANWS];

    RxAgg G;
    std::memset(&G, 0, sizeof(G));
    RxPass P;
    std::memset(&P, 0, sizeof(P));
    const int wide = rx_plan_wide(f[PL_KEYKIND]);
    const int kcols = wide ? wide : 1;
    P.n_cols = kcols + (int)f[PL_NCARRIED];
    P.wide = wide;
    P.key = *key;
    P.sel = sel;
    P.row0 = row0;
    for (int c = 0; c < P.n_cols; ++c) P.esize[c] = (int)f[PL_ESIZE0 + c];
    int carried = 0;
    for (int a = 0; a < NA; ++a) {
        if (val_cols[a].data == nullptr) {
            G.carried[a] = 0;
            G.const_cell[a] = const_cells[a];
            continue;
        }
        ++carried;
        if (carried > (int)f[PL_NCARRIED] || rx_esize(val_cols[a].kind) != (int)f[PL_ESIZE0 + kcols - 1 + carried]) {
            hs_set_error("hs_group_radix_run: value column %d does not match the plan", a);
            return HS_E_ARG;
        }
        P.src[kcols - 1 + carried] = val_cols[a].data;
        G.carried[a] = kcols - 1 + carried;
        G.val_kind[a] = val_cols[a].kind;
    }
    if (carried != (int)f[PL_NCARRIED]) {
        hs_set_error("hs_group_radix_run: %d travelling value columns, the plan has %d", carried, (int)f[PL_NCARRIED]);
        return HS_E_ARG;
    }

    auto pass = [&](const int64_t* seg_start, int64_t n_seg, int64_t* tile_base, int64_t max_tiles, int shift, int bits, bool first,
                    void* const* src, void* const* dst, int64_t* next_seg) -> int {
        P.seg_start = seg_start;
        P.tile_base = tile_base;
        P.n_seg = n_seg;
        P.shift = shift;
        P.bits = bits;
        P.first = first ? 1 : 0;
        if (!first)
            for (int c = 0; c < P.n_cols; ++c) P.src[c] = src[c];
        for (int c = 0; c < P.n_cols; ++c) P.dst[c] = dst[c];
        return rx_pass(stream, P, max_tiles, counters, scanned, scan_ws, n, next_seg);
    };
    int rc = pass(unit_bounds, n_units, tb0, f[PL_TILES1], 0, bits1, true, nullptr, buf_a, seg1);
    if (rc != HS_OK) return rc;
    void* const* tuples = buf_a;
    if (bits2) {
        rc = pass(seg1, f[PL_NSEG1], tb1, f[PL_TILES2], bits1, bits2, false, buf_a, buf_b, seg2);
        if (rc != HS_OK) return rc;
        tuples = buf_b;
    }

    const int64_t parts = f[PL_PARTS];
    G.seg_start = seg2;
    G.n_parts = parts;
    for (int c = 0; c < P.n_cols; ++c) {
        G.src[c] = tuples[c];
        G.esize[c] = P.esize[c];
    }
    G.spec = *spec;
    G.cap = (int)f[PL_CAP];
    G.quantise = (int)f[PL_QUANTISE];
    G.prov_key = (uint64_t*)(ws + f[PL_OFF_PKEY]);
    G.prov_key1 = G.prov_key + n;
    for (int a = 0; a < NA; ++a) G.prov_acc[a] = ws + f[PL_OFF_PACC] + (size_t)a * rx_align((size_t)n * f[PL_OSIZE]);
    G.pcount = (int64_t*)(ws + f[PL_OFF_PCOUNT]);
    G.flags = flags;
    static const bool stamps = getenv("HIPSPARK_RADIX_STAMPS") != nullptr;
    G.debug = stamps ? 1 : 0;
    int64_t* overflow = (int64_t*)(ws + f[PL_OFF_OVERFLOW]);  // [0] = count, [1 ..] = partitions
    int64_t* overflow2 = overflow + f[PL_PARTS] + 1;           // the same for the next table size
    hs_memset_async(overflow, 0, 8, stream);
    hs_memset_async(overflow2, 0, 8, stream);
    // every aggregate a SUM over an f32 / i32 column or an integer constant: the specialised fold (k_rx_fold_sum)
    int sum_cls = NA >= 1 && NA <= 3 ? 0 : -1;
    for (int a = 0; a < NA && sum_cls >= 0; ++a) {
        const bool is_int = spec->is_int[a] != 0;
        int c = -1;
        if (spec->op[a] != HS_AGG_SUM) c = -1;
        else if (G.carried[a] && G.val_kind[a] == HS_F32 && !is_int) c = 0;
        else if (G.carried[a] && G.val_kind[a] == HS_I32 && is_int) c = 1;
        else if (!G.carried[a] && is_int) c = 2;
        sum_cls = c < 0 ? -1 : (sum_cls | (c << (2 * a)));
    }
    if (stamps && !wide) sum_cls = -1;  // (the phase stamps live in the general fold)
    if (wide && sum_cls < 0) {
        hs_set_error("hs_group_radix_run: a wide STRING key needs the SUM-specialised fold");
        return HS_E_LIMIT;
    }
    // todo: the partitions this launch works on (NULL: all), left: where it notes the ones that outgrow its tables
    auto fold = [&](int cap, int64_t* todo, int64_t* left) {
        G.cap = cap;
        G.last = cap == (int)f[PL_CAP] ? 1 : 0;
        G.list = todo ? todo + 1 : nullptr;
        G.list_count = todo;
        G.overflow = left + 1;
        G.overflow_count = left;
        const size_t per_wave = ((size_t)cap * ((wide ? 2 : 1) + NA) + (size_t)cap / 4) * 8;
        int wpb = (int)(65536 / per_wave);
        wpb = wpb < 1 ? 1 : (wpb > 4 ? 4 : wpb);
        int64_t grid = (parts + wpb - 1) / wpb;
        if (grid > 256 * 32) grid = 256 * 32;
        const dim3 g((unsigned)grid), t(HS_WAVE * wpb);
#define RX_FOLD(NC)                                                                                    \
    if (cap <= 512) hipLaunchKernelGGL((k_rx_fold<NC, 9>), g, t, per_wave * wpb, stream, G);           \
    else hipLaunchKernelGGL((k_rx_fold<NC, 12>), g, t, per_wave * wpb, stream, G)
#define RX_SUM1(C0) \
    case (C0):                                                                                              \
        switch (km) { \
            case 0: hipLaunchKernelGGL((k_rx_fold_sum<1, (C0), 0>), g, t, per_wave * wpb, stream, G); break; \
            case 1: hipLaunchKernelGGL((k_rx_fold_sum<1, (C0), 1>), g, t, per_wave * wpb, stream, G); break; \
            case 2: hipLaunchKernelGGL((k_rx_fold_sum<1, (C0), 2>), g, t, per_wave * wpb, stream, G); break; \
            case 3: hipLaunchKernelGGL((k_rx_fold_sum<1, (C0), 3>), g, t, per_wave * wpb, stream, G); break; \
            case 4: hipLaunchKernelGGL((k_rx_fold_sum<1, (C0), 4>), g, t, per_wave * wpb, stream, G); break; \
        } \
        return;
#define RX_SUM2(C0, C1) \
    case ((C0) | ((C1) << 2)):                                                                                             \
        switch (km) { \
            case 0: hipLaunchKernelGGL((k_rx_fold_sum<2, ((C0) | ((C1) << 2)), 0>), g, t, per_wave * wpb, stream, G); break; \
            case 1: hipLaunchKernelGGL((k_rx_fold_sum<2, ((C0) | ((C1) << 2)), 1>), g, t, per_wave * wpb, stream, G); break; \
            case 2: hipLaunchKernelGGL((k_rx_fold_sum<2, ((C0) | ((C1) << 2)), 2>), g, t, per_wave * wpb, stream, G); break; \
            case 3: hipLaunchKernelGGL((k_rx_fold_sum<2, ((C0) | ((C1) << 2)), 3>), g, t, per_wave * wpb, stream, G); break; \
            case 4: hipLaunchKernelGGL((k_rx_fold_sum<2, ((C0) | ((C1) << 2)), 4>), g, t, per_wave * wpb, stream, G); break; \
        } \
        return;
#define RX_SUM3(C0, C1, C2) \
    case ((C0) | ((C1) << 2) | ((C2) << 4)): \
        switch (km) { \
            case 0: hipLaunchKernelGGL((k_rx_fold_sum<3, ((C0) | ((C1) << 2) | ((C2) << 4)), 0>), g, t, per_wave * wpb, stream, G); break; \
            case 1: hipLaunchKernelGGL((k_rx_fold_sum<3, ((C0) | ((C1) << 2) | ((C2) << 4)), 1>), g, t, per_wave * wpb, stream, G); break; \
            case 2: hipLaunchKernelGGL((k_rx_fold_sum<3, ((C0) | ((C1) << 2) | ((C2) << 4)), 2>), g, t, per_wave * wpb, stream, G); break; \
            case 3: hipLaunchKernelGGL((k_rx_fold_sum<3, ((C0) | ((C1) << 2) | ((C2) << 4)), 3>), g, t, per_wave * wpb, stream, G); break; \
            case 4: hipLaunchKernelGGL((k_rx_fold_sum<3, ((C0) | ((C1) << 2) | ((C2) << 4)), 4>), g, t, per_wave * wpb, stream, G); break; \
        } \
        return;
#define RX_SUM3_LAST(C0, C1) RX_SUM3(C0, C1, 0) RX_SUM3(C0, C1, 1) RX_SUM3(C0, C1, 2)
#define RX_SUM3_MID(C0) RX_SUM3_LAST(C0, 0) RX_SUM3_LAST(C0, 1) RX_SUM3_LAST(C0, 2)
        const int km = wide ? kcols : (G.esize[0] == 4 ? 0 : 1);
        if (sum_cls >= 0 && NA == 1) {
            switch (sum_cls) { RX_SUM1(0) RX_SUM1(1) RX_SUM1(2) default: break; }
        } else if (sum_cls >= 0 && NA == 2) {
            switch (sum_cls) {
                RX_SUM2(0, 0) RX_SUM2(1, 0) RX_SUM2(2, 0) RX_SUM2(0, 1) RX_SUM2(1, 1) RX_SUM2(2, 1) RX_SUM2(0, 2) RX_SUM2(1, 2) RX_SUM2(2, 2)
                default: break;
            }
        } else if (sum_cls >= 0 && NA == 3) {
            switch (sum_cls) { RX_SUM3_MID(0) RX_SUM3_MID(1) RX_SUM3_MID(2) default: break; }
        }
#undef RX_SUM1
#undef RX_SUM2
#undef RX_SUM3
#undef RX_SUM3_LAST
#undef RX_SUM3_MID
        switch ((int)f[PL_NCARRIED]) {
            case 0: RX_FOLD(0); break;
            case 1: RX_FOLD(1); break;
            case 2: RX_FOLD(2); break;
            case 3: RX_FOLD(3); break;
            case 4: RX_FOLD(4); break;
            default: RX_FOLD(-1); break;
        }
#undef RX_FOLD
    };
    // Three table sizes, smallest first: a partition is cut to hold ~half the LARGE table's slots in ROWS, and usually holds far
    // fewer distinct keys.  The fold is a chain of LDS round trips per 64-row step, one wave per partition - what it needs is
    // waves: 256-slot tables leave room for 28 per CU (512: 16), and a partition that outgrows 3/4 of a table moves to the next
    // size's list (round 4; 64 Mi rows / 4 Mi groups: fold 0.52 -> see profiles/r04_radix_tier_64M.txt).
    const int big = (int)f[PL_CAP], small = big > 512 ? 512 : big, tiny = small > 256 ? 256 : small;
    if (tiny < small) {
        fold(tiny, nullptr, overflow);
        RX_CHECK_LAUNCH("hs_group_radix_run (fold)");
        fold(small, overflow, overflow2);
        if (small < big) {
            RX_CHECK_LAUNCH("hs_group_radix_run (fold)");
            fold(big, overflow2, overflow2);
        }
    } else if (small < big) {
        fold(small, nullptr, overflow);
        RX_CHECK_LAUNCH("hs_group_radix_run (fold)");
        fold(big, overflow, overflow);
    } else {
        fold(big, nullptr, overflow);  // nothing can overflow into a list: a full table raises HS_FLAG_DICT_FULL ...
    }
    RX_CHECK_LAUNCH("hs_group_radix_run (fold)");
    int64_t* pscan = (int64_t*)(ws + f[PL_OFF_PSCAN]);
    rc = hs_exclusive_scan_i64(stream, G.pcount, parts, pscan, scan_ws);
    if (rc != HS_OK) return rc;
    hipLaunchKernelGGL(k_rx_unit_groups, dim3((unsigned)((n_units + 256) / 256)), dim3(256), 0, stream, (const int64_t*)pscan,
                       (int64_t)1 << (bits1 + bits2), (int32_t)n_units, out_unit_groups);
    RX_CHECK_LAUNCH("hs_group_radix_run (unit groups)");
    return HS_OK;
}

extern "C" int hs_group_radix_emit(void* stream_, const hs_radix_plan* plan, void* ws_, void* out_key, void* const* out_acc) {
    if (!plan || !ws_ || !out_key || (plan->f[PL_NA] > 0 && !out_acc)) {
        hs_set_error("hs_group_radix_emit: bad arguments");
        return HS_E_ARG;
    }
    const int64_t* f = plan->f;
    uint8_t* ws = (uint8_t*)ws_;
    RxEmit E;
    std::memset(&E, 0, sizeof(E));
    E.seg_start = (const int64_t*)(ws + (f[PL_BITS2] ? f[PL_OFF_SEG2] : f[PL_OFF_SEG1]));
    E.pscan = (const int64_t*)(ws + f[PL_OFF_PSCAN]);
    E.n_parts = f[PL_PARTS];
    E.prov_key = (const uint64_t*)(ws + f[PL_OFF_PKEY]);
    E.prov_key1 = E.prov_key + f[PL_N];
    E.n_acc = (int)f[PL_NA];
    E.osize = (int)f[PL_OSIZE];
    E.key_kind = (int)f[PL_KEYKIND];
    E.out_key = out_key;
    for (int a = 0; a < E.n_acc; ++a) {
        E.prov_acc[a] = ws + f[PL_OFF_PACC] + (size_t)a * rx_align((size_t)f[PL_N] * f[PL_OSIZE]);
        E.out_acc[a] = out_acc[a];
    }
    int64_t grid = (E.n_parts + 3) / 4;
    if (grid > 256 * 64) grid = 256 * 64;
    hipLaunchKernelGGL(k_rx_emit, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream_, E);
    RX_CHECK_LAUNCH("hs_group_radix_emit");
    return HS_OK;
}

/* debug (HIPSPARK_RADIX_STAMPS=1): cycles per phase of the fold kernel summed over waves since the last call:
 * [0] clear tables [1] wait for a chunk's loads [2] slot lookup [3] ranking [4] fold [5] emit [6] waves */
extern "C" int hs_group_radix_debug_stamps(uint64_t* out8) {
    unsigned long long zero[8] = {0};
    if (!out8 || hipDeviceSynchronize() != hipSuccess ||
        hipMemcpyFromSymbol(out8, HIP_SYMBOL(rx_stamp_acc), sizeof(zero)) != hipSuccess ||
        hipMemcpyToSymbol(HIP_SYMBOL(rx_stamp_acc), zero, sizeof(zero)) != hipSuccess) {
        hs_set_error("hs_group_radix_debug_stamps failed");
        return HS_E_ARG;
    }
    return HS_OK;
}

// =====================================================================================================
// Merge order of a multi-rank final aggregate (reference: the shuffle files of a partition are read in block order,
// tasks.py:117-133): a STABLE sort of the partial rows by their order key (global block id, -1 = padding) with the
// same partition passes, least significant byte first.  Replaces torch.argsort on this path.
// =====================================================================================================
__global__ void __launch_bounds__(256) k_rx_iota(int64_t* out, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) out[i] = i;
    if (blockIdx.x == 0 && threadIdx.x == 0) {  // the one segment of the passes: [0, n)
        out[n] = 0;
        out[n + 1] = n;
    }
}

extern "C" size_t hs_sort_by_order_ws_bytes(int64_t n) {
    const int64_t tiles = n / RX_TILE + 2;
    return rx_align((size_t)n * 8) * 2 + rx_align((size_t)(n + 2) * 8) + 256 + rx_align((size_t)(tiles << 8) * 8) +
           rx_align((size_t)((tiles << 8) + 1) * 8) + rx_align(hs_scan_ws_bytes(tiles << 8));
}

extern "C" int hs_sort_by_order(void* stream_, const int64_t* order, int64_t n, int64_t n_order, int64_t* out_perm,
                                int64_t* out_sorted, void* ws_) {
    if (n == 0) return HS_OK;
    if (!order || !out_perm || !out_sorted || !ws_ || n < 0 || n_order < 0) {
        hs_set_error("hs_sort_by_order: bad arguments");
        return HS_E_ARG;
    }
    hipStream_t stream = (hipStream_t)stream_;
    uint8_t* ws = (uint8_t*)ws_;
    const int64_t tiles = n / RX_TILE + 2;
    size_t off = 0;
    auto take = [&](size_t bytes) {
        uint8_t* p = ws + off;
        off += rx_align(bytes);
        return p;
    };
    int64_t* key_b = (int64_t*)take((size_t)n * 8);
    int64_t* perm_b = (int64_t*)take((size_t)n * 8);
    int64_t* iota = (int64_t*)take((size_t)(n + 2) * 8);  // [n], then the segment bounds {0, n}
    int64_t* tile_base = (int64_t*)take(256);
    int64_t* counters = (int64_t*)take((size_t)(tiles << 8) * 8);
    int64_t* scanned = (int64_t*)take((size_t)((tiles << 8) + 1) * 8);
    void* scan_ws = take(hs_scan_ws_bytes(tiles << 8));
    hipLaunchKernelGGL(k_rx_iota, dim3((unsigned)((n + 255) / 256 > 4096 ? 4096 : (n + 255) / 256)), dim3(256), 0, stream, iota, n);
    RX_CHECK_LAUNCH("hs_sort_by_order (iota)");
    int bits = 1;
    while (bits < 64 && ((int64_t)1 << bits) <= n_order) ++bits;  // values + 1 lie in [0, n_order]
    const int passes = (bits + 7) / 8;
    RxPass P;
    std::memset(&P, 0, sizeof(P));
    P.seg_start = iota + n;
    P.tile_base = tile_base;
    P.n_seg = 1;
    P.n_cols = 2;
    P.esize[0] = P.esize[1] = 8;
    P.raw = 1;
    for (int k = 0; k < passes; ++k) {
        const bool to_out = ((passes - 1 - k) & 1) == 0;  // the last pass lands in the caller's arrays
        P.first = k == 0 ? 1 : 0;
        P.key = hs_col{HS_I64, -1, order, nullptr, nullptr};
        P.sel = nullptr;
        P.row0 = 0;
        P.src[0] = k == 0 ? nullptr : (to_out ? (const void*)key_b : (const void*)out_sorted);
        P.src[1] = k == 0 ? (const void*)iota : (to_out ? (const void*)perm_b : (const void*)out_perm);
        P.dst[0] = to_out ? (void*)out_sorted : (void*)key_b;
        P.dst[1] = to_out ? (void*)out_perm : (void*)perm_b;
        P.shift = 8 * k;
        P.bits = bits - 8 * k < 8 ? bits - 8 * k : 8;
        const int rc = rx_pass(stream, P, tiles, counters, scanned, scan_ws, n, nullptr);
        if (rc != HS_OK) return rc;
    }
    return HS_OK;
}

// out[i] = values[s] for bounds[s] <= i < bounds[s + 1]: a value per segment spread over the segment's rows (the global
// block id of every partial row, multi-rank partial aggregate).  Replaces torch.repeat_interleave.
__global__ void __launch_bounds__(256) k_expand_by_bounds(const int64_t* bounds, const int64_t* values, int64_t n_seg, int64_t n,
                                                          int64_t* out) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        int64_t lo = 0, hi = n_seg;  // last s with bounds[s] <= i
        while (hi - lo > 1) {
            const int64_t mid = (lo + hi) >> 1;
            if (bounds[mid] <= i) lo = mid;
            else hi = mid;
        }
        out[i] = values[lo];
    }
}
extern "C" int hs_expand_by_bounds(void* stream, const int64_t* bounds, const int64_t* values, int64_t n_seg, int64_t n, int64_t* out) {
    if (n == 0) return HS_OK;
    if (!bounds || !values || !out || n_seg < 1 || n < 0) {
        hs_set_error("hs_expand_by_bounds: bad arguments");
        return HS_E_ARG;
    }
    hipLaunchKernelGGL(k_expand_by_bounds, dim3((unsigned)((n + 255) / 256 > 4096 ? 4096 : (n + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, bounds, values, n_seg, n, out);
    RX_CHECK_LAUNCH("hs_expand_by_bounds");
    return HS_OK;
}

// =====================================================================================================
// The general inner join on INTEGER keys over a dense key range (round 4; include/hipspark.h hs_join_dense_*)
// =====================================================================================================
// Reference: BroadcastHashJoinTask.generate_chunks tasks.py:201-240 - a hash map key -> build rows (in row order), every
// probe row emits its matches in that order; duplicates on either side multiply (zig twin: tasks.zig:70-194, 258-326).
// Round 1-3 built one open-addressing table over all build keys with a 64-bit atomic per row scattered over 800 MB
// (8.6 G keys/s, 5.6 % of the HBM roofline).  Here the build side is MOVED instead, like the radix tier of GROUP BY:
//   1. two stable partition passes (the kernels above, RxPass.range) on the most significant bits of slot = key - key_min
//      bring the (key, row) tuples into partitions of 2^L consecutive slots, still in row order;
//   2. one wave per partition (k_jd_assemble): slot counts in LDS, a scan, then the rows are placed IN ORDER - lanes of a
//      64-tuple step that share a slot rank themselves with ballots - so every slot's rows come out ascending without a
//      sort and without a global atomic; the partition's slice of starts[] (CSR offsets, one per SLOT, absent keys
//      included) leaves with coalesced stores.
// The probe is then one or two adjacent 4-byte reads per row (starts[s], starts[s + 1]) instead of a hash probe over
// three arrays: count -> exclusive scan -> fill, pairs ordered by probe row, then build row (tasks.py:224-240).
constexpr int JD_MAX_L = 13;  // slots of one partition: 2^L cursors + 2^L first rows (uint32) + 2^L tag bytes in LDS per wave (72 KB at most)
constexpr uint32_t JD_EMPTY = 0xffffffffu;  // slot word: no build row has this key
constexpr uint32_t JD_MULTI = 0x80000000u;  // slot word: several - the low 31 bits are the start of the key's list in rows[]

struct JdAssemble {
    const int64_t* seg_start;  // [parts + 1] tuple ranges of the partitions
    int64_t parts;
    const int32_t* keys;       // tuples, partitioned
    const uint32_t* rows;
    int64_t slots;
    int32_t key_min, L;
    uint32_t* words;           // [slots]: JD_EMPTY | the one build row | JD_MULTI + list start
    uint32_t* out_rows;        // [n]: build rows in slot order, ascending within a slot
    uint32_t* list_count;      // [n]: at the start of a list of several rows, its length
    int64_t n;
    uint32_t* flags;
};

__global__ void __launch_bounds__(256) k_jd_assemble(const JdAssemble A) {
    extern __shared__ __align__(16) uint32_t jd_lds[];
    const int lane = threadIdx.x & (HS_WAVE - 1), w = threadIdx.x / HS_WAVE, wpb = blockDim.x / HS_WAVE;
    const int W = 1 << A.L;
    uint32_t* cur = jd_lds + (size_t)w * (2 * W + W / 4);  // counts, then list cursors
    uint32_t* head = cur + W;                    // the first (= lowest) build row of every slot
    uint8_t* tag = (uint8_t*)(head + W);         // placement: the lane that came by last in this step
    const uint64_t below = (1ull << lane) - 1ull;
    const int per = W / HS_WAVE;  // consecutive slots of a lane in the scan (W >= 64)
    const uint32_t wmask = (uint32_t)(W - 1), kmin = (uint32_t)A.key_min;
    uint32_t err = 0;
    for (int64_t p = (int64_t)blockIdx.x * wpb + w; p < A.parts; p += (int64_t)gridDim.x * wpb) {
        const int64_t b = A.seg_start[p], e = A.seg_start[p + 1];
        const int64_t slot0 = p << A.L;
        if (slot0 >= A.slots) {
            if (e > b) err |= HS_FLAG_BAD_PROGRAM;  // a key past the declared range
            continue;
        }
        // the first step's tuples are asked for before the tables are cleared; every later step's before the current one
        // is worked on (a wave walks its partition in order: without this it paid one global round trip per step)
        int32_t nkey = b + lane < e ? A.keys[b + lane] : 0;
        for (int s = lane; s < W; s += HS_WAVE) {
            cur[s] = 0;
            head[s] = JD_EMPTY;
        }
        rx_wave_handover();
        for (int64_t base = b; base < e; base += HS_WAVE) {
            const int32_t key = nkey;
            const bool valid = base + lane < e;
            if (base + HS_WAVE + lane < e) nkey = A.keys[base + HS_WAVE + lane];
            if (valid) atomicAdd(&cur[((uint32_t)key - kmin) & wmask], 1u);  // LDS; counting is order-free
        }
        rx_wave_handover();
        // exclusive scan of the W counts: a lane's consecutive slots, then a scan over the lanes
        uint32_t sum = 0;
        for (int k = 0; k < per; ++k) sum += cur[lane * per + k];
        uint32_t x = sum;
        for (int d = 1; d < HS_WAVE; d <<= 1) {
            const uint32_t up = __shfl_up(x, d, HS_WAVE);
            if (lane >= d) x += up;
        }
        uint32_t run = x - sum;
        for (int k = 0; k < per; ++k) {
            const uint32_t c = cur[lane * per + k];
            cur[lane * per + k] = run;
            run += c;
        }
        rx_wave_handover();
        // ordered placement: tuples arrive in row order; within a step, equal slots take consecutive places in lane order.
        // Usually the 64 tuples of a step fall into 64 DIFFERENT slots (unique or nearly unique build keys): every lane
        // leaves its number in a tag byte of its slot and reads it back - all lanes find their own: no ranking needed (the
        // ten ballots per step were a third of this kernel).
        nkey = b + lane < e ? A.keys[b + lane] : 0;
        uint32_t nrow = b + lane < e ? A.rows[b + lane] : 0u;
        for (int64_t base = b; base < e; base += HS_WAVE) {
            const bool valid = base + lane < e;
            const uint32_t s = ((uint32_t)nkey - kmin) & wmask;
            const uint32_t row = nrow;
            if (base + HS_WAVE + lane < e) {
                nkey = A.keys[base + HS_WAVE + lane];
                nrow = A.rows[base + HS_WAVE + lane];
            }
            if (valid) tag[s] = (uint8_t)lane;
            rx_wave_handover();
            const bool shared_slot = valid && tag[s] != (uint8_t)lane;
            const uint32_t at = valid ? cur[s] : 0u;
            const uint32_t first = valid ? head[s] : 0u;
            if (__ballot(shared_slot) == 0) {  // wave-uniform
                if (valid) {
                    A.out_rows[b + at] = row;
                    cur[s] = at + 1u;
                    if (first == JD_EMPTY) head[s] = row;
                }
                rx_wave_handover();
                continue;
            }
            uint64_t peers = __ballot(valid);
            for (int bit = 0; bit < A.L; ++bit) {
                const bool on = (s >> bit) & 1u;
                const uint64_t bal = __ballot(valid && on);
                peers &= on ? bal : ~bal;
            }
            const uint32_t rank = (uint32_t)__popcll(peers & below);
            rx_wave_handover();  // every lane has read its slot's cursor before a leader moves it
            if (valid) {
                A.out_rows[b + at + rank] = row;
                if (rank == 0) {
                    cur[s] = at + (uint32_t)__popcll(peers);
                    if (first == JD_EMPTY) head[s] = row;  // the step's lowest lane of the slot holds its lowest row
                }
            }
            rx_wave_handover();
        }
        // the partition's slice of the slot words (coalesced): lists are contiguous, so list s starts where list s - 1 ends
        const int64_t live = A.slots - slot0 < W ? A.slots - slot0 : W;
        for (int s = lane; s < live; s += HS_WAVE) {
            const uint32_t end = cur[s], start = s ? cur[s - 1] : 0u;
            const uint32_t c = end - start;
            uint32_t word = JD_EMPTY;
            if (c == 1) word = head[s];
            else if (c > 1) {
                word = JD_MULTI | (uint32_t)(b + start);
                A.list_count[b + start] = c;
            }
            A.words[slot0 + s] = word;
        }
        rx_wave_handover();
    }
    if (err) atomicOr(A.flags, err);
}

// the one segment [0, n) of the first pass; without passes (a key range of one partition) the tuples are the build column
// itself and the row ids 0 .. n-1
__global__ void __launch_bounds__(256) k_jd_setup(uint32_t* iota, int64_t n_iota, int64_t n, int64_t* seg0) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_iota; i += (int64_t)gridDim.x * blockDim.x) iota[i] = (uint32_t)i;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        seg0[0] = 0;
        seg0[1] = n;
    }
}

// the passes' geometry for a key range of `slots` slots: L = slots per partition (log2), bits1 + bits2 = partition bits
static void jd_geometry(int64_t slots, int& L, int& bits1, int& bits2) {
    int S = 0;
    while (S < 31 && ((int64_t)1 << S) < slots) ++S;
    L = 9;  // 512 slots: 4.5 KB of LDS per wave in the assembly (it runs on the number of waves a CU holds) ...
    if (S - L > 2 * RX_MAX_BITS) L = S - 2 * RX_MAX_BITS;  // ... more when 65 536 partitions would not cover the range
    const int bits = S > L ? S - L : 0;
    bits1 = bits <= RX_MAX_BITS ? bits : (bits + 1) / 2;
    bits2 = bits - bits1;
}
struct JdLayout {
    size_t keys_a, rows_a, keys_b, rows_b, iota, seg0, seg1, seg2, tb0, tb1, cnt, scan, scan_ws, total;
    int64_t tiles1, tiles2, nseg1, parts, counters;
    int L, bits1, bits2;
};
static bool jd_layout(int64_t n, int64_t slots, JdLayout& Y) {
    if (n < 0 || n >= 0x7fffffffll || slots < 1 || slots > ((int64_t)1 << 29)) return false;  // a slot word holds a row in 31 bits
    jd_geometry(slots, Y.L, Y.bits1, Y.bits2);
    if (Y.L > JD_MAX_L) return false;  // (slots <= 2^29 with 16 partition bits)
    Y.nseg1 = (int64_t)1 << Y.bits1;
    Y.parts = (int64_t)1 << (Y.bits1 + Y.bits2);
    Y.tiles1 = n / RX_TILE + 2;
    Y.tiles2 = Y.bits2 ? n / RX_TILE + Y.nseg1 + 1 : 0;
    const int64_t c1 = Y.tiles1 << Y.bits1, c2 = Y.tiles2 << Y.bits2;
    Y.counters = c1 > c2 ? c1 : c2;
    size_t off = 0;
    auto take = [&](size_t bytes) {
        const size_t at = off;
        off += rx_align(bytes);
        return at;
    };
    Y.keys_a = take((size_t)n * 4 + 64);
    Y.rows_a = take((size_t)n * 4 + 64);
    Y.keys_b = take(Y.bits2 ? (size_t)n * 4 + 64 : 0);
    Y.rows_b = take(Y.bits2 ? (size_t)n * 4 + 64 : 0);
    Y.iota = take(Y.bits1 ? 64 : (size_t)n * 4 + 64);  // row ids travel from the first pass on; needed as an array only without passes
    Y.seg0 = take(16);
    Y.seg1 = take((size_t)(Y.nseg1 + 1) * 8);
    Y.seg2 = take((size_t)(Y.parts + 1) * 8);
    Y.tb0 = take(16);
    Y.tb1 = take((size_t)(Y.nseg1 + 1) * 8);
    Y.cnt = take((size_t)Y.counters * 8);
    Y.scan = take((size_t)(Y.counters + 1) * 8);
    Y.scan_ws = take(hs_scan_ws_bytes(Y.counters > 1 ? Y.counters : 1));
    Y.total = off;
    return true;
}

extern "C" size_t hs_join_dense_ws_bytes(int64_t n_build, int64_t slots) {
    JdLayout Y;
    return jd_layout(n_build, slots, Y) ? Y.total : 0;
}

extern "C" int hs_join_dense_build(void* stream_, const int32_t* build_keys, int64_t n_build, int32_t key_min, int64_t slots,
                                   uint32_t* words, uint32_t* rows, uint32_t* list_count, void* ws_, uint32_t* flags) {
    JdLayout Y;
    if ((!build_keys && n_build > 0) || !words || !rows || !list_count || !ws_ || !flags || !jd_layout(n_build, slots, Y)) {
        hs_set_error("hs_join_dense_build: bad arguments (n_build < 2^31, 1 <= slots <= 2^29)");
        return HS_E_ARG;
    }
    hipStream_t stream = (hipStream_t)stream_;
    uint8_t* ws = (uint8_t*)ws_;
    const int64_t n = n_build;
    uint32_t* iota = (uint32_t*)(ws + Y.iota);
    int64_t* seg0 = (int64_t*)(ws + Y.seg0);
    int64_t* seg1 = (int64_t*)(ws + Y.seg1);
    int64_t* seg2 = (int64_t*)(ws + Y.seg2);
    const int64_t n_iota = Y.bits1 ? 0 : n;
    int64_t grid = (n_iota + 255) / 256;
    grid = grid < 1 ? 1 : (grid > 4096 ? 4096 : grid);
    hipLaunchKernelGGL(k_jd_setup, dim3((unsigned)grid), dim3(256), 0, stream, iota, n_iota, n, seg0);
    RX_CHECK_LAUNCH("hs_join_dense_build (row ids)");
    const int32_t* t_keys = build_keys;
    const uint32_t* t_rows = iota;
    const int64_t* seg = seg0;
    if (Y.bits1 > 0 && n > 0) {
        RxPass P;
        std::memset(&P, 0, sizeof(P));
        P.n_cols = 2;
        P.esize[0] = P.esize[1] = 4;
        P.range = 1;
        P.range_bias = key_min;
        P.key = hs_col{HS_I32, -1, build_keys, nullptr, nullptr};
        P.row0 = 0;
        // pass 1: the top bits1 bits of the slot over the one segment [0, n)
        P.seg_start = seg0;
        P.tile_base = (int64_t*)(ws + Y.tb0);
        P.n_seg = 1;
        P.shift = Y.L + Y.bits2;
        P.bits = Y.bits1;
        P.first = 1;
        P.src[1] = nullptr;  // the row id column is the position (k_rx_scatter4)
        P.dst[0] = ws + Y.keys_a;
        P.dst[1] = ws + Y.rows_a;
        int rc = rx_pass(stream, P, Y.tiles1, (int64_t*)(ws + Y.cnt), (int64_t*)(ws + Y.scan), ws + Y.scan_ws, n, Y.bits2 ? seg1 : seg2);
        if (rc != HS_OK) return rc;
        t_keys = (const int32_t*)(ws + Y.keys_a);
        t_rows = (const uint32_t*)(ws + Y.rows_a);
        seg = seg2;
        if (Y.bits2) {  // pass 2: the next bits2 bits inside every segment of pass 1
            P.seg_start = seg1;
            P.tile_base = (int64_t*)(ws + Y.tb1);
            P.n_seg = Y.nseg1;
            P.shift = Y.L;
            P.bits = Y.bits2;
            P.first = 0;
            P.src[0] = ws + Y.keys_a;
            P.src[1] = ws + Y.rows_a;
            P.dst[0] = ws + Y.keys_b;
            P.dst[1] = ws + Y.rows_b;
            rc = rx_pass(stream, P, Y.tiles2, (int64_t*)(ws + Y.cnt), (int64_t*)(ws + Y.scan), ws + Y.scan_ws, n, seg2);
            if (rc != HS_OK) return rc;
            t_keys = (const int32_t*)(ws + Y.keys_b);
            t_rows = (const uint32_t*)(ws + Y.rows_b);
        }
    } else if (Y.bits1 > 0) {  // no rows: every partition is empty
        hs_memset_async(seg2, 0, (size_t)(Y.parts + 1) * 8, stream);
        seg = seg2;
    }
    JdAssemble A;
    A.seg_start = seg;
    A.parts = Y.bits1 > 0 ? Y.parts : 1;
    A.keys = t_keys;
    A.rows = t_rows;
    A.slots = slots;
    A.key_min = key_min;
    A.L = Y.L;
    A.words = words;
    A.out_rows = rows;
    A.list_count = list_count;
    A.n = n;
    A.flags = flags;
    const size_t per_wave = (size_t)9 << Y.L;  // cursors + first rows (uint32) + tag bytes
    int wpb = (int)(65536 / per_wave);
    wpb = wpb < 1 ? 1 : (wpb > 4 ? 4 : wpb);
    int64_t g = (A.parts + wpb - 1) / wpb;
    if (g > 256 * 32) g = 256 * 32;
    static unsigned long long attr_set = 0;
    if (hs_first_on_device(attr_set)) (void)hipFuncSetAttribute((const void*)k_jd_assemble, hipFuncAttributeMaxDynamicSharedMemorySize, 9 << JD_MAX_L);
    hipLaunchKernelGGL(k_jd_assemble, dim3((unsigned)g), dim3(HS_WAVE * wpb), per_wave * wpb, stream, A);
    RX_CHECK_LAUNCH("hs_join_dense_build (assemble)");
    return HS_OK;
}

struct JdProbe {
    const int32_t* keys;
    int64_t n, slots;
    int32_t key_min, pad;
    const uint32_t* words;
    const uint32_t* rows;
    const uint32_t* list_count;
    int64_t* counts;
    uint32_t* aux;  // [2 n]: first matching build row | start of the probe row's list in `rows`
    const int64_t* out_start;
    int64_t* out_left;
    int64_t* out_right;
};
typedef int jd_i32x4 __attribute__((ext_vector_type(4)));
typedef unsigned jd_u32x4 __attribute__((ext_vector_type(4)));
typedef long long jd_i64x2 __attribute__((ext_vector_type(2)));

// Probe, pass 1: four keys per lane (one 16-byte load; buffers carry slack past n), ONE scattered 4-byte read per key - its
// slot word says "no partner", names the one partner, or points at a list (then, and only then, two more reads: the
// list's length and its first row); the four lookups of a lane are in flight together.  What the fill pass needs is written
// down sequentially (the first build row and the list's start: 8 bytes per probe row), so it never returns to the
// scattered arrays for a key with one partner - the usual case.
__global__ void __launch_bounds__(256) k_jd_count(const JdProbe A) {
    const int64_t nq = (A.n + 3) / 4;
    const int64_t second = (A.n + 3) & ~(int64_t)3;
    for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < nq; q += (int64_t)gridDim.x * blockDim.x) {
        const jd_i32x4 kv = __builtin_nontemporal_load(reinterpret_cast<const jd_i32x4*>(A.keys) + q);
        const int32_t k[4] = {kv.x, kv.y, kv.z, kv.w};
        uint32_t word[4], cnt[4], first[4], st[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int64_t o = (int64_t)k[j] - (int64_t)A.key_min;
            const bool in = q * 4 + j < A.n && (uint64_t)o < (uint64_t)A.slots;
            word[j] = in ? A.words[o] : JD_EMPTY;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const bool multi = word[j] != JD_EMPTY && (word[j] & JD_MULTI);
            st[j] = multi ? word[j] & ~JD_MULTI : 0u;
            cnt[j] = word[j] == JD_EMPTY ? 0u : 1u;
            first[j] = word[j];
            if (multi) {
                cnt[j] = A.list_count[st[j]];
                first[j] = A.rows[st[j]];
            }
        }
        if (q * 4 + 3 < A.n) {
            int64_t* c = A.counts + q * 4;
            __builtin_nontemporal_store(jd_i64x2{(long long)cnt[0], (long long)cnt[1]}, reinterpret_cast<jd_i64x2*>(c));
            __builtin_nontemporal_store(jd_i64x2{(long long)cnt[2], (long long)cnt[3]}, reinterpret_cast<jd_i64x2*>(c + 2));
            __builtin_nontemporal_store(jd_u32x4{first[0], first[1], first[2], first[3]}, reinterpret_cast<jd_u32x4*>(A.aux) + q);
            __builtin_nontemporal_store(jd_u32x4{st[0], st[1], st[2], st[3]}, reinterpret_cast<jd_u32x4*>(A.aux + second) + q);
        } else {
            for (int j = 0; j < 4 && q * 4 + j < A.n; ++j) {
                A.counts[q * 4 + j] = (int64_t)cnt[j];
                A.aux[q * 4 + j] = first[j];
                A.aux[second + q * 4 + j] = st[j];
            }
        }
    }
}

// Probe, pass 2: pairs ordered by probe row, then build row.  A stream: output offsets (their differences are the match
// counts), the first build rows, the pairs; only a probe row with several partners reads the rest of its list.
__global__ void __launch_bounds__(256) k_jd_fill(const JdProbe A) {
    const int64_t nq = (A.n + 3) / 4;
    const int64_t second = (A.n + 3) & ~(int64_t)3;
    for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < nq; q += (int64_t)gridDim.x * blockDim.x) {
        int64_t at[5];
        uint32_t first[4];
        if (q * 4 + 3 < A.n) {
            const jd_u32x4 f = __builtin_nontemporal_load(reinterpret_cast<const jd_u32x4*>(A.aux) + q);
            first[0] = f.x; first[1] = f.y; first[2] = f.z; first[3] = f.w;
#pragma unroll
            for (int j = 0; j < 5; ++j) at[j] = A.out_start[q * 4 + j];  // out_start has n + 1 entries
        } else {
            for (int j = 0; j < 5; ++j) at[j] = q * 4 + j <= A.n ? A.out_start[q * 4 + j] : A.out_start[A.n];
            for (int j = 0; j < 4; ++j) first[j] = q * 4 + j < A.n ? A.aux[q * 4 + j] : 0u;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int64_t cnt = at[j + 1] - at[j];
            if (cnt <= 0) continue;
            const int64_t i = q * 4 + j;
            A.out_left[at[j]] = (int64_t)first[j];
            A.out_right[at[j]] = i;
            if (cnt > 1) {
                const uint32_t st = A.aux[second + i];
                for (int64_t r = 1; r < cnt; ++r) {
                    A.out_left[at[j] + r] = (int64_t)A.rows[st + r];
                    A.out_right[at[j] + r] = i;
                }
            }
        }
    }
}

static int jd_probe_args(const char* who, const int32_t* probe_keys, int64_t n_probe, int64_t slots, const uint32_t* words) {
    if ((!probe_keys && n_probe > 0) || !words || n_probe < 0 || slots < 1 || ((uintptr_t)probe_keys & 15)) {
        hs_set_error("%s: bad arguments (probe keys 16-byte aligned)", who);
        return HS_E_ARG;
    }
    return HS_OK;
}
extern "C" size_t hs_join_dense_aux_bytes(int64_t n_probe) { return n_probe < 0 ? 0 : (size_t)(((n_probe + 3) & ~(int64_t)3) * 2) * 4 + 64; }
extern "C" int hs_join_dense_count(void* stream, const int32_t* probe_keys, int64_t n_probe, int32_t key_min, int64_t slots,
                                   const uint32_t* words, const uint32_t* rows, const uint32_t* list_count, int64_t* counts, void* aux) {
    int rc = jd_probe_args("hs_join_dense_count", probe_keys, n_probe, slots, words);
    if (rc != HS_OK) return rc;
    if (!rows || !list_count || !counts || !aux || ((uintptr_t)counts & 15) || ((uintptr_t)aux & 15)) {
        hs_set_error("hs_join_dense_count: counts and aux must be 16-byte aligned");
        return HS_E_ARG;
    }
    if (n_probe == 0) return HS_OK;
    JdProbe A{probe_keys, n_probe, slots, key_min, 0, words, rows, list_count, counts, (uint32_t*)aux, nullptr, nullptr, nullptr};
    int64_t g = ((n_probe + 3) / 4 + 255) / 256;
    if (g > 256 * 64) g = 256 * 64;
    hipLaunchKernelGGL(k_jd_count, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, A);
    RX_CHECK_LAUNCH("hs_join_dense_count");
    return HS_OK;
}
extern "C" int hs_join_dense_fill(void* stream, int64_t n_probe, const uint32_t* rows, const void* aux, const int64_t* out_start,
                                  int64_t* out_left, int64_t* out_right) {
    if (n_probe < 0 || !rows || !aux || !out_start || !out_left || !out_right || ((uintptr_t)aux & 15)) {
        hs_set_error("hs_join_dense_fill: bad arguments");
        return HS_E_ARG;
    }
    if (n_probe == 0) return HS_OK;
    JdProbe A{nullptr, n_probe, 0, 0, 0, nullptr, rows, nullptr, nullptr, (uint32_t*)const_cast<void*>(aux), out_start, out_left, out_right};
    int64_t g = ((n_probe + 3) / 4 + 255) / 256;
    if (g > 256 * 64) g = 256 * 64;
    hipLaunchKernelGGL(k_jd_fill, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, A);
    RX_CHECK_LAUNCH("hs_join_dense_fill");
    return HS_OK;
}

// =====================================================================================================
// The general inner join on ANY INTEGER keys (round 4; include/hipspark.h hs_join_hash_*)
// =====================================================================================================
// Same reference loop as above (tasks.py:201-240), for keys the dense form does not hold: a sparse or huge key range,
// negative keys, a few hot values.  The table is an open-addressing table of 8-byte slots {key, word} cut into WINDOWS of
// 2^JH_L slots; a key belongs to window jh_window(key) and probes linearly INSIDE it (wrapping at the window's end), so
//   * the build never leaves LDS: the (key, row) tuples are brought into window order by the two stable partition passes
//     (RxPass.range = 2), one wave per window inserts its keys into an LDS copy of the window (ds_cmpst on a 64-bit cell),
//     counts, scans and places the rows in order exactly like the dense form, and stores the finished window with coalesced
//     8-byte stores - no global atomic, no scattered store, every list ascending without a sort;
//   * a probe is ONE scattered 8-byte read in the usual case (the slot holds key and word together; at a load of ~0.57 a
//     present key sits 1.7 slots from its start on average - the same 64-byte line nearly always).
// The slot word is the dense form's (JD_EMPTY / the one build row / JD_MULTI + list start), rows[] and list_count[] too, so
// the probe's second pass IS hs_join_dense_fill.
// slots per window, log2: 512 (8.5 KB of LDS per wave: key cells, cursors, first rows, tag bytes - 18 waves per CU) while
// 65 536 windows hold the build side (19 M rows), else 1024 (17 KB: 9 waves per CU; 38 M rows).  The assembly is a chain of LDS
// round trips per 64-tuple step, one wave per window: it runs on the number of waves a CU holds.
constexpr int JH_L_SMALL = 9, JH_L_LARGE = 10;
constexpr uint64_t JH_FREE = ~0ull;            // LDS key cell: nobody here yet (a key occupies the low 32 bits only)

static int64_t jh_windows(int64_t n_build, int L) {   // ~1.75 slots per build row: distinct keys <= rows
    const int64_t w = (n_build * 7 / 4 + ((int64_t)1 << L) - 1) >> L;
    return w < 1 ? 1 : w;
}

struct JhAssemble {
    const int64_t* seg_start;  // [parts + 1] tuple ranges of the partitions = windows (parts >= windows; the rest are empty)
    int64_t parts, windows;
    const int32_t* keys;       // tuples, window by window, in row order inside a window
    const uint32_t* rows;
    uint2* table;              // [windows << L] {key, word}
    uint32_t* out_rows;        // [n]: build rows window by window, slot by slot, ascending within a slot
    uint32_t* list_count;      // [n]: at the start of a list of several rows, its length
    uint16_t* slot_of;         // [n] scratch: the slot every tuple found in the counting pass (the placement pass reads it back
                               // with coalesced loads instead of walking the LDS window again)
    uint32_t* flags;
};

// slot of `key` in the wave's LDS window, claiming a free cell on the way when INSERT; -1: the window is full (more
// distinct keys than slots - jh_windows leaves 1.75x room over the AVERAGE window; the caller falls back)
template <bool INSERT, int JH_L>
__device__ __forceinline__ int jh_slot(uint64_t* cell, uint32_t key, uint32_t start) {
    constexpr uint32_t wmask = (1u << JH_L) - 1u;
    uint32_t s = start & wmask;
    for (int probe = 0; probe <= (int)wmask; ++probe) {
        uint64_t cur = __hip_atomic_load(&cell[s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (INSERT && cur == JH_FREE) {
            cur = atomicCAS((unsigned long long*)&cell[s], (unsigned long long)JH_FREE, (unsigned long long)key);
            if (cur == JH_FREE) return (int)s;
        }
        if (cur == (uint64_t)key) return (int)s;
        if (!INSERT && cur == JH_FREE) return -1;
        s = (s + 1) & wmask;
    }
    return -1;
}

template <int JH_L>
__global__ void __launch_bounds__(256) k_jh_assemble(const JhAssemble A) {
    extern __shared__ __align__(16) uint64_t jh_lds[];
    constexpr int W = 1 << JH_L, per = W / HS_WAVE;
    const int lane = threadIdx.x & (HS_WAVE - 1), w = threadIdx.x / HS_WAVE, wpb = blockDim.x / HS_WAVE;
    uint64_t* cell = jh_lds + (size_t)w * (2 * W + W / 8);   // key cells
    uint32_t* cur = (uint32_t*)(cell + W);         // counts, then list cursors
    uint32_t* head = cur + W;                      // the first (= lowest) build row of every slot
    uint8_t* tag = (uint8_t*)(head + W);           // placement: the lane that came by last in this step
    const uint64_t below = (1ull << lane) - 1ull;
    uint32_t err = 0;
    for (int64_t p = (int64_t)blockIdx.x * wpb + w; p < A.parts; p += (int64_t)gridDim.x * wpb) {
        const int64_t b = A.seg_start[p], e = A.seg_start[p + 1];
        if (p >= A.windows) {
            if (e > b) err |= HS_FLAG_BAD_PROGRAM;  // a tuple past the last window: the passes and the table disagree
            continue;
        }
        int32_t nkey = b + lane < e ? A.keys[b + lane] : 0;
        for (int s = lane; s < W; s += HS_WAVE) {
            cell[s] = JH_FREE;
            cur[s] = 0;
            head[s] = JD_EMPTY;
        }
        rx_wave_handover();
        bool full = false;
        for (int64_t base = b; base < e; base += HS_WAVE) {  // claim slots, count rows per slot
            const uint32_t key = (uint32_t)nkey;
            const bool valid = base + lane < e;
            if (base + HS_WAVE + lane < e) nkey = A.keys[base + HS_WAVE + lane];
            if (valid) {
                const int s = jh_slot<true, JH_L>(cell, key, rx_mix32(key));
                if (s < 0) full = true;
                else atomicAdd(&cur[s], 1u);
                A.slot_of[base + lane] = (uint16_t)s;
            }
        }
        rx_wave_handover();
        if (__ballot(full)) {  // (wave-uniform) leave the window empty and say so
            err |= HS_FLAG_DICT_FULL;
            for (int s = lane; s < W; s += HS_WAVE) A.table[(p << JH_L) + s] = make_uint2(0u, JD_EMPTY);
            continue;
        }
        uint32_t sum = 0;
        for (int k = 0; k < per; ++k) sum += cur[lane * per + k];
        uint32_t x = sum;
        for (int d = 1; d < HS_WAVE; d <<= 1) {
            const uint32_t up = __shfl_up(x, d, HS_WAVE);
            if (lane >= d) x += up;
        }
        uint32_t run = x - sum;
        for (int k = 0; k < per; ++k) {
            const uint32_t c = cur[lane * per + k];
            cur[lane * per + k] = run;
            run += c;
        }
        rx_wave_handover();
        uint32_t nslot = b + lane < e ? A.slot_of[b + lane] : 0u;
        uint32_t nrow = b + lane < e ? A.rows[b + lane] : 0u;
        for (int64_t base = b; base < e; base += HS_WAVE) {  // ordered placement (see k_jd_assemble)
            const bool valid = base + lane < e;
            const uint32_t s = nslot, row = nrow;
            if (base + HS_WAVE + lane < e) {
                nslot = A.slot_of[base + HS_WAVE + lane];
                nrow = A.rows[base + HS_WAVE + lane];
            }
            if (valid) tag[s] = (uint8_t)lane;
            rx_wave_handover();
            const bool shared_slot = valid && tag[s] != (uint8_t)lane;
            const uint32_t at = valid ? cur[s] : 0u;
            const uint32_t first = valid ? head[s] : 0u;
            if (__ballot(shared_slot) == 0) {  // (wave-uniform) 64 different slots: nothing to rank
                if (valid) {
                    A.out_rows[b + at] = row;
                    cur[s] = at + 1u;
                    if (first == JD_EMPTY) head[s] = row;
                }
                rx_wave_handover();
                continue;
            }
            uint64_t peers = __ballot(valid);
            for (int bit = 0; bit < JH_L; ++bit) {
                const bool on = (s >> bit) & 1u;
                const uint64_t bal = __ballot(valid && on);
                peers &= on ? bal : ~bal;
            }
            const uint32_t rank = (uint32_t)__popcll(peers & below);
            rx_wave_handover();
            if (valid) {
                A.out_rows[b + at + rank] = row;
                if (rank == 0) {
                    cur[s] = at + (uint32_t)__popcll(peers);
                    if (first == JD_EMPTY) head[s] = row;
                }
            }
            rx_wave_handover();
        }
        // the finished window (coalesced).  Lists lie in SLOT order, so list s starts where the previous slot's ends.
        for (int s = lane; s < W; s += HS_WAVE) {
            const uint32_t end = cur[s], start = s ? cur[s - 1] : 0u;
            const uint32_t c = end - start;
            uint32_t word = JD_EMPTY;
            if (c == 1) word = head[s];
            else if (c > 1) {
                word = JD_MULTI | (uint32_t)(b + start);
                A.list_count[b + start] = c;
            }
            A.table[(p << JH_L) + s] = make_uint2((uint32_t)cell[s], word);
        }
        rx_wave_handover();
    }
    if (err) atomicOr(A.flags, err);
}

struct JhLayout {
    size_t keys_a, rows_a, keys_b, rows_b, iota, slot_of, seg0, seg1, seg2, tb0, tb1, cnt, scan, scan_ws, total;
    int64_t tiles1, tiles2, nseg1, parts, counters, windows;
    int bits1, bits2, L;
};
static bool jh_layout(int64_t n, JhLayout& Y) {
    if (n < 0 || n >= 0x7fffffffll) return false;  // a slot word holds a row in 31 bits
    Y.L = jh_windows(n, JH_L_SMALL) <= ((int64_t)1 << (2 * RX_MAX_BITS)) ? JH_L_SMALL : JH_L_LARGE;
    Y.windows = jh_windows(n, Y.L);
    int bits = 0;
    while (((int64_t)1 << bits) < Y.windows) ++bits;
    if (bits > 2 * RX_MAX_BITS) return false;  // 65 536 windows = 38 M build rows per call
    Y.bits1 = bits <= RX_MAX_BITS ? bits : (bits + 1) / 2;
    Y.bits2 = bits - Y.bits1;
    Y.nseg1 = (int64_t)1 << Y.bits1;
    Y.parts = (int64_t)1 << bits;
    Y.tiles1 = n / RX_TILE + 2;
    Y.tiles2 = Y.bits2 ? n / RX_TILE + Y.nseg1 + 1 : 0;
    const int64_t c1 = Y.tiles1 << Y.bits1, c2 = Y.tiles2 << Y.bits2;
    Y.counters = c1 > c2 ? c1 : c2;
    size_t off = 0;
    auto take = [&](size_t bytes) {
        const size_t at = off;
        off += rx_align(bytes);
        return at;
    };
    Y.keys_a = take((size_t)n * 4 + 64);
    Y.rows_a = take((size_t)n * 4 + 64);
    Y.keys_b = take(Y.bits2 ? (size_t)n * 4 + 64 : 0);
    Y.rows_b = take(Y.bits2 ? (size_t)n * 4 + 64 : 0);
    Y.iota = take(Y.bits1 ? 64 : (size_t)n * 4 + 64);
    Y.slot_of = take((size_t)n * 2 + 64);
    Y.seg0 = take(16);
    Y.seg1 = take((size_t)(Y.nseg1 + 1) * 8);
    Y.seg2 = take((size_t)(Y.parts + 1) * 8);
    Y.tb0 = take(16);
    Y.tb1 = take((size_t)(Y.nseg1 + 1) * 8);
    Y.cnt = take((size_t)Y.counters * 8);
    Y.scan = take((size_t)(Y.counters + 1) * 8);
    Y.scan_ws = take(hs_scan_ws_bytes(Y.counters > 1 ? Y.counters : 1));
    Y.total = off;
    return true;
}

extern "C" size_t hs_join_hash_ws_bytes(int64_t n_build) {
    JhLayout Y;
    return jh_layout(n_build, Y) ? Y.total : 0;
}
extern "C" int64_t hs_join_hash_slots(int64_t n_build) {
    JhLayout Y;
    return jh_layout(n_build, Y) ? Y.windows << Y.L : 0;
}

extern "C" int hs_join_hash_build(void* stream_, const int32_t* build_keys, int64_t n_build, void* table, uint32_t* rows,
                                  uint32_t* list_count, void* ws_, uint32_t* flags) {
    JhLayout Y;
    if ((!build_keys && n_build > 0) || !table || !rows || !list_count || !ws_ || !flags || ((uintptr_t)table & 7) || !jh_layout(n_build, Y)) {
        hs_set_error("hs_join_hash_build: bad arguments (n_build <= 38 M rows, table 8-byte aligned)");
        return HS_E_ARG;
    }
    hipStream_t stream = (hipStream_t)stream_;
    uint8_t* ws = (uint8_t*)ws_;
    const int64_t n = n_build;
    uint32_t* iota = (uint32_t*)(ws + Y.iota);
    int64_t* seg0 = (int64_t*)(ws + Y.seg0);
    int64_t* seg1 = (int64_t*)(ws + Y.seg1);
    int64_t* seg2 = (int64_t*)(ws + Y.seg2);
    const int64_t n_iota = Y.bits1 ? 0 : n;
    int64_t grid = (n_iota + 255) / 256;
    grid = grid < 1 ? 1 : (grid > 4096 ? 4096 : grid);
    hipLaunchKernelGGL(k_jd_setup, dim3((unsigned)grid), dim3(256), 0, stream, iota, n_iota, n, seg0);
    RX_CHECK_LAUNCH("hs_join_hash_build (row ids)");
    const int32_t* t_keys = build_keys;
    const uint32_t* t_rows = iota;
    const int64_t* seg = seg0;
    if (Y.bits1 > 0 && n > 0) {
        RxPass P;
        std::memset(&P, 0, sizeof(P));
        P.n_cols = 2;
        P.esize[0] = P.esize[1] = 4;
        P.range = 2;
        P.range_bias = (int32_t)Y.windows;
        P.key = hs_col{HS_I32, -1, build_keys, nullptr, nullptr};
        P.row0 = 0;
        P.seg_start = seg0;  // pass 1: the top bits1 bits of the window number over the one segment [0, n)
        P.tile_base = (int64_t*)(ws + Y.tb0);
        P.n_seg = 1;
        P.shift = Y.bits2;
        P.bits = Y.bits1;
        P.first = 1;
        P.src[1] = nullptr;  // the row id column is the position (k_rx_scatter4)
        P.dst[0] = ws + Y.keys_a;
        P.dst[1] = ws + Y.rows_a;
        int rc = rx_pass(stream, P, Y.tiles1, (int64_t*)(ws + Y.cnt), (int64_t*)(ws + Y.scan), ws + Y.scan_ws, n, Y.bits2 ? seg1 : seg2);
        if (rc != HS_OK) return rc;
        t_keys = (const int32_t*)(ws + Y.keys_a);
        t_rows = (const uint32_t*)(ws + Y.rows_a);
        seg = seg2;
        if (Y.bits2) {  // pass 2: the low bits2 bits inside every segment of pass 1
            P.seg_start = seg1;
            P.tile_base = (int64_t*)(ws + Y.tb1);
            P.n_seg = Y.nseg1;
            P.shift = 0;
            P.bits = Y.bits2;
            P.first = 0;
            P.src[0] = ws + Y.keys_a;
            P.src[1] = ws + Y.rows_a;
            P.dst[0] = ws + Y.keys_b;
            P.dst[1] = ws + Y.rows_b;
            rc = rx_pass(stream, P, Y.tiles2, (int64_t*)(ws + Y.cnt), (int64_t*)(ws + Y.scan), ws + Y.scan_ws, n, seg2);
            if (rc != HS_OK) return rc;
            t_keys = (const int32_t*)(ws + Y.keys_b);
            t_rows = (const uint32_t*)(ws + Y.rows_b);
        }
    } else if (Y.bits1 > 0) {  // no rows: every window is empty
        hs_memset_async(seg2, 0, (size_t)(Y.parts + 1) * 8, stream);
        seg = seg2;
    }
    JhAssemble A;
    A.seg_start = seg;
    A.parts = Y.bits1 > 0 ? Y.parts : 1;
    A.windows = Y.windows;
    A.keys = t_keys;
    A.rows = t_rows;
    A.table = (uint2*)table;
    A.out_rows = rows;
    A.list_count = list_count;
    A.slot_of = (uint16_t*)(ws + Y.slot_of);
    A.flags = flags;
    const size_t per_wave = (size_t)17 << Y.L;  // key cells (8 B) + cursors + first rows (4 B each) + tag bytes
    constexpr int wpb = 4;
    int64_t g = (A.parts + wpb - 1) / wpb;
    if (g > 256 * 32) g = 256 * 32;
    static unsigned long long attr_set = 0;
    if (hs_first_on_device(attr_set))
        (void)hipFuncSetAttribute((const void*)k_jh_assemble<JH_L_LARGE>, hipFuncAttributeMaxDynamicSharedMemorySize, (17 << JH_L_LARGE) * wpb);
    if (Y.L == JH_L_SMALL) hipLaunchKernelGGL(k_jh_assemble<JH_L_SMALL>, dim3((unsigned)g), dim3(HS_WAVE * wpb), per_wave * wpb, stream, A);
    else hipLaunchKernelGGL(k_jh_assemble<JH_L_LARGE>, dim3((unsigned)g), dim3(HS_WAVE * wpb), per_wave * wpb, stream, A);
    RX_CHECK_LAUNCH("hs_join_hash_build (assemble)");
    return HS_OK;
}

struct JhProbe {
    const int32_t* keys;
    int64_t n;
    uint32_t windows, pad;
    const uint2* table;
    const uint32_t* rows;
    const uint32_t* list_count;
    int64_t* counts;
    uint32_t* aux;
};

// Probe, pass 1 (the hashed twin of k_jd_count): four keys per lane, their first slots in flight together; a slot that
// holds another key sends the lane to the next one of the window.  counts / aux as hs_join_dense_count writes them.
template <int JH_L>
__global__ void __launch_bounds__(256) k_jh_count(const JhProbe A) {
    constexpr uint32_t wmask = (1u << JH_L) - 1u;
    const int64_t nq = (A.n + 3) / 4;
    const int64_t second = (A.n + 3) & ~(int64_t)3;
    for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < nq; q += (int64_t)gridDim.x * blockDim.x) {
        const jd_i32x4 kv = __builtin_nontemporal_load(reinterpret_cast<const jd_i32x4*>(A.keys) + q);
        const uint32_t k[4] = {(uint32_t)kv.x, (uint32_t)kv.y, (uint32_t)kv.z, (uint32_t)kv.w};
        uint32_t word[4], cnt[4], first[4], st[4], at[4];
        const uint2* win[4];
        uint2 slot[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint32_t h = rx_mix32(k[j]);
            win[j] = A.table + ((int64_t)(((uint64_t)h * A.windows) >> 32) << JH_L);
            at[j] = h & wmask;
            slot[j] = win[j][at[j]];
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            for (int probe = 0; probe < (int)wmask && slot[j].y != JD_EMPTY && slot[j].x != k[j]; ++probe) {
                at[j] = (at[j] + 1) & wmask;
                slot[j] = win[j][at[j]];
            }
            word[j] = q * 4 + j < A.n && slot[j].x == k[j] ? slot[j].y : JD_EMPTY;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const bool multi = word[j] != JD_EMPTY && (word[j] & JD_MULTI);
            st[j] = multi ? word[j] & ~JD_MULTI : 0u;
            cnt[j] = word[j] == JD_EMPTY ? 0u : 1u;
            first[j] = word[j];
            if (multi) {
                cnt[j] = A.list_count[st[j]];
                first[j] = A.rows[st[j]];
            }
        }
        if (q * 4 + 3 < A.n) {
            int64_t* c = A.counts + q * 4;
            __builtin_nontemporal_store(jd_i64x2{(long long)cnt[0], (long long)cnt[1]}, reinterpret_cast<jd_i64x2*>(c));
            __builtin_nontemporal_store(jd_i64x2{(long long)cnt[2], (long long)cnt[3]}, reinterpret_cast<jd_i64x2*>(c + 2));
            __builtin_nontemporal_store(jd_u32x4{first[0], first[1], first[2], first[3]}, reinterpret_cast<jd_u32x4*>(A.aux) + q);
            __builtin_nontemporal_store(jd_u32x4{st[0], st[1], st[2], st[3]}, reinterpret_cast<jd_u32x4*>(A.aux + second) + q);
        } else {
            for (int j = 0; j < 4 && q * 4 + j < A.n; ++j) {
                A.counts[q * 4 + j] = (int64_t)cnt[j];
                A.aux[q * 4 + j] = first[j];
                A.aux[second + q * 4 + j] = st[j];
            }
        }
    }
}

extern "C" int hs_join_hash_count(void* stream, const int32_t* probe_keys, int64_t n_probe, int64_t n_build, const void* table,
                                  const uint32_t* rows, const uint32_t* list_count, int64_t* counts, void* aux) {
    JhLayout Y;
    if ((!probe_keys && n_probe > 0) || n_probe < 0 || !table || !rows || !list_count || !counts || !aux || ((uintptr_t)probe_keys & 15) ||
        ((uintptr_t)counts & 15) || ((uintptr_t)aux & 15) || !jh_layout(n_build, Y)) {
        hs_set_error("hs_join_hash_count: bad arguments (probe keys, counts and aux 16-byte aligned; n_build as given to the build)");
        return HS_E_ARG;
    }
    if (n_probe == 0) return HS_OK;
    JhProbe A{probe_keys, n_probe, (uint32_t)Y.windows, 0, (const uint2*)table, rows, list_count, counts, (uint32_t*)aux};
    int64_t g = ((n_probe + 3) / 4 + 255) / 256;
    if (g > 256 * 64) g = 256 * 64;
    if (Y.L == JH_L_SMALL) hipLaunchKernelGGL(k_jh_count<JH_L_SMALL>, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, A);
    else hipLaunchKernelGGL(k_jh_count<JH_L_LARGE>, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, A);
    RX_CHECK_LAUNCH("hs_join_hash_count");
    return HS_OK;
}
