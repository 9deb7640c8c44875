// LDS atomic throughput probe (gfx950): lane-ops per cycle per CU for 64-bit / 32-bit atomics under different
// address patterns.  build: hipcc --offload-arch=gfx950 -O3 -o lds_atomic_probe lds_atomic_probe.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

enum { P_LINEAR = 0, P_STRIDE2 = 1, P_RANDOM = 2, P_SAME = 3, P_RANDOM50 = 4 };

template <int PAT>
__device__ __forceinline__ uint32_t cell_of(uint32_t lane, uint32_t wave, uint32_t it, uint32_t ncell) {
    if (PAT == P_LINEAR) return (lane + 64u * ((it + wave) & 63u)) % ncell;
    if (PAT == P_STRIDE2) return (2u * lane + 128u * ((it + wave) & 31u)) % ncell;
    if (PAT == P_SAME) return (it + wave) % ncell;
    uint32_t h = (lane * 2654435761u) ^ ((it + 977u * wave) * 40503u);
    h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
    if (PAT == P_RANDOM50) return ((h % 50u) * 64u + lane) % ncell;  // 50 keys x 64 replicas (replica = lane)
    return h % ncell;
}

template <int OP, int PAT>
__global__ __launch_bounds__(1024) void k_probe(int iters, uint64_t* out) {
    extern __shared__ uint64_t cells[];
    const uint32_t ncell = 8192;
    for (uint32_t i = threadIdx.x; i < ncell; i += blockDim.x) cells[i] = 0;
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint64_t sink = 0;
    for (int it = 0; it < iters; ++it) {
        const uint32_t c = cell_of<PAT>(lane, wave, (uint32_t)it, ncell);
        if (OP == 0) atomicAdd((double*)&cells[c], 1.0);
        else if (OP == 1) atomicAdd((unsigned long long*)&cells[c], 1ull);
        else if (OP == 2) atomicAdd((uint32_t*)&cells[c], 1u);
        else if (OP == 3) atomicAdd((float*)&cells[c], 1.0f);
        else if (OP == 4) atomicMax((double*)&cells[c], (double)it);
        else if (OP == 5) { double v = ((volatile double*)cells)[c]; ((volatile double*)cells)[c] = v + 1.0; }  // plain RMW
        else if (OP == 6) sink += ((volatile uint64_t*)cells)[c];  // plain read
    }
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = cells[0] + sink;
}

template <int OP, int PAT>
static void run(const char* name, int waves_per_cu) {
    const int iters = 4096, grid = 256;
    uint64_t* out;
    hipMalloc(&out, grid * sizeof(uint64_t));
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    const int threads = waves_per_cu * 64;
    k_probe<OP, PAT><<<grid, threads, 65536>>>(16, out);
    hipDeviceSynchronize();
    hipEventRecord(a);
    k_probe<OP, PAT><<<grid, threads, 65536>>>(iters, out);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    const double cycles = ms * 1e-3 * 2.4e9;  // nominal 2.4 GHz
    const double lane_ops = (double)iters * threads;
    printf("%-34s waves/CU %2d: %8.3f ms  %6.2f lane-ops/cycle/CU  (%6.1f cycles per wave-op)\n", name, waves_per_cu, ms,
           lane_ops / cycles, cycles / ((double)iters * waves_per_cu));
    hipFree(out);
}

#define ALLPAT(OP, NAME)                                   \
    run<OP, P_LINEAR>(NAME " linear (conflict-free)", 16); \
    run<OP, P_STRIDE2>(NAME " stride 16 B", 16);           \
    run<OP, P_RANDOM>(NAME " random cell", 16);            \
    run<OP, P_RANDOM50>(NAME " 50 keys x 64 replicas", 16); \
    run<OP, P_SAME>(NAME " one cell per wave", 16);

int main() {
    hipFuncSetAttribute((const void*)k_probe<0, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    ALLPAT(0, "ds_add_f64")
    ALLPAT(1, "ds_add_u64")
    ALLPAT(2, "ds_add_u32")
    ALLPAT(3, "ds_add_f32")
    ALLPAT(4, "ds_max_f64")
    ALLPAT(5, "read+add+write f64")
    ALLPAT(6, "ds_read_b64")
    run<0, P_LINEAR>("ds_add_f64 linear", 4);
    run<0, P_LINEAR>("ds_add_f64 linear", 8);
    run<1, P_RANDOM>("ds_add_u64 random", 4);
    run<1, P_RANDOM>("ds_add_u64 random", 8);
    return 0;
}
