// read_ceiling.hip - what can a read-only stream over Q1's seven columns reach on one MI355X, as a function of the
// bytes a wave keeps in flight?  Stand-alone probe (hipcc --offload-arch=gfx950 -O3 -o read_ceiling read_ceiling.hip).
//
//   ./read_ceiling [rows=600037902]
//
// Every variant reads 4 x f32 + 1 x i64 + 1 x u8 per row (25 B: the bytes the Q1 scan kernel moves) with 16-byte
// non-temporal loads, 4 rows per lane per step like the scan, and adds everything into one double per lane (no LDS
// tables, no dictionary: the load path alone).  DEPTH = steps whose loads are in flight ahead of the one being summed;
// WGS = workgroups per CU the LDS padding leaves room for; STEPS = steps per workgroup (chunk length).
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define CHECK(x)                                                                          \
    do {                                                                                  \
        hipError_t e_ = (x);                                                              \
        if (e_ != hipSuccess) {                                                           \
            fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
            exit(1);                                                                      \
        }                                                                                 \
    } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));

struct Cols {
    const float *q, *p, *d, *t;
    const unsigned long long* ship;
    const uint8_t* flag;
};

struct Quad {
    f32x4 q, p, d, t;
    u64x2 s0, s1;
    uint32_t f;
};

template <bool NT>
__device__ __forceinline__ void load_quad(const Cols& c, int64_t base, Quad& x) {
    if constexpr (NT) {
        x.q = __builtin_nontemporal_load((const f32x4*)(c.q + base));
        x.p = __builtin_nontemporal_load((const f32x4*)(c.p + base));
        x.d = __builtin_nontemporal_load((const f32x4*)(c.d + base));
        x.t = __builtin_nontemporal_load((const f32x4*)(c.t + base));
        x.s0 = __builtin_nontemporal_load((const u64x2*)(c.ship + base));
        x.s1 = __builtin_nontemporal_load((const u64x2*)(c.ship + base + 2));
        x.f = __builtin_nontemporal_load((const uint32_t*)(c.flag + base));
    } else {
        x.q = *(const f32x4*)(c.q + base);
        x.p = *(const f32x4*)(c.p + base);
        x.d = *(const f32x4*)(c.d + base);
        x.t = *(const f32x4*)(c.t + base);
        x.s0 = *(const u64x2*)(c.ship + base);
        x.s1 = *(const u64x2*)(c.ship + base + 2);
        x.f = *(const uint32_t*)(c.flag + base);
    }
}

__device__ __forceinline__ double eat(const Quad& x) {
    double s = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) s += (double)x.q[j] + (double)x.p[j] + (double)x.d[j] + (double)x.t[j];
    s += (double)(x.s0[0] ^ x.s0[1] ^ x.s1[0] ^ x.s1[1]) * 1e-30 + (double)x.f * 1e-30;
    return s;
}

// one workgroup per chunk of `steps` steps x 1024 rows
template <int DEPTH, bool NT>
__global__ __launch_bounds__(256) void k_read(Cols c, int64_t rows, int steps, double* out) {
    extern __shared__ uint64_t pad[];
    const int64_t stride = 256 * 4;
    const int64_t c0 = (int64_t)blockIdx.x * steps * stride;
    int64_t c1 = c0 + (int64_t)steps * stride;
    if (c1 > rows) c1 = rows & ~(int64_t)3;
    int64_t base = c0 + (int64_t)threadIdx.x * 4;
    Quad buf[DEPTH + 1];
#pragma unroll
    for (int k = 0; k < DEPTH; ++k)
        if (base + k * stride < c1) load_quad<NT>(c, base + k * stride, buf[k]);
    double acc = 0;
    // rotate through the buffers with a fully unrolled period of DEPTH + 1 steps so that indices stay compile-time
    while (base < c1) {
#pragma unroll
        for (int k = 0; k <= DEPTH; ++k) {
            if (base < c1) {
                const int64_t ahead = base + (int64_t)DEPTH * stride;
                if (ahead < c1) load_quad<NT>(c, ahead, buf[(k + DEPTH) % (DEPTH + 1)]);
                acc += eat(buf[k]);
                base += stride;
            }
        }
    }
    if (acc == 12345.678) pad[threadIdx.x] = 1;  // keeps the dynamic LDS block referenced
    // fixed-order reduce is not the point here
    for (int d = 32; d >= 1; d >>= 1) acc += __shfl_down(acc, d);
    if ((threadIdx.x & 63) == 0) atomicAdd(&out[blockIdx.x & 1023], acc);
}

// ONE stream of the same number of bytes: does reading seven columns side by side cost anything against the plainest read?
__global__ __launch_bounds__(256) void k_read_one(const f32x4* p, int64_t n16, int steps, double* out) {
    const int64_t stride = 256;
    const int64_t c0 = (int64_t)blockIdx.x * steps * stride * 6;
    int64_t c1 = c0 + (int64_t)steps * stride * 6;
    if (c1 > n16) c1 = n16;
    double acc = 0;
    for (int64_t base = c0 + threadIdx.x; base < c1; base += stride * 6) {
        f32x4 v[6];
#pragma unroll
        for (int k = 0; k < 6; ++k) v[k] = base + k * stride < c1 ? __builtin_nontemporal_load(p + base + k * stride) : f32x4{0, 0, 0, 0};
#pragma unroll
        for (int k = 0; k < 6; ++k) acc += (double)v[k][0] + (double)v[k][1] + (double)v[k][2] + (double)v[k][3];
    }
    for (int d = 32; d >= 1; d >>= 1) acc += __shfl_down(acc, d);
    if ((threadIdx.x & 63) == 0) atomicAdd(&out[blockIdx.x & 1023], acc);
}

template <int DEPTH, bool NT>
static double run(const Cols& c, int64_t rows, int steps, size_t lds, double* out, hipStream_t s, int reps = 7) {
    CHECK(hipFuncSetAttribute((const void*)k_read<DEPTH, NT>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
    const int64_t per = (int64_t)steps * 1024;
    const unsigned grid = (unsigned)((rows + per - 1) / per);
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a));
    CHECK(hipEventCreate(&b));
    std::vector<float> ms;
    for (int r = 0; r < reps + 2; ++r) {
        CHECK(hipEventRecord(a, s));
        hipLaunchKernelGGL((k_read<DEPTH, NT>), dim3(grid), dim3(256), lds, s, c, rows, steps, out);
        CHECK(hipEventRecord(b, s));
        CHECK(hipStreamSynchronize(s));
        float t;
        CHECK(hipEventElapsedTime(&t, a, b));
        if (r >= 2) ms.push_back(t);
    }
    std::sort(ms.begin(), ms.end());
    return ms[ms.size() / 2];
}

int main(int argc, char** argv) {
    const int64_t rows = argc > 1 ? atoll(argv[1]) : 600037902ll;
    const int64_t padded = (rows + 4095) & ~(int64_t)4095;
    Cols c;
    float* f[4];
    for (int i = 0; i < 4; ++i) {
        CHECK(hipMalloc(&f[i], padded * 4));
        CHECK(hipMemset(f[i], 0, padded * 4));
    }
    unsigned long long* ship;
    uint8_t* flag;
    CHECK(hipMalloc(&ship, padded * 8));
    CHECK(hipMemset(ship, 0, padded * 8));
    CHECK(hipMalloc(&flag, padded));
    CHECK(hipMemset(flag, 65, padded));
    c.q = f[0]; c.p = f[1]; c.d = f[2]; c.t = f[3]; c.ship = ship; c.flag = flag;
    double* out;
    CHECK(hipMalloc(&out, 1024 * 8));
    CHECK(hipMemset(out, 0, 1024 * 8));
    hipStream_t s;
    CHECK(hipStreamCreate(&s));
    const double gb = 25.0 * (double)rows / 1e9, gb26 = 26.0 * (double)rows / 1e9;
    printf("rows %lld: %.3f GB moved (25 B/row), %.3f GB algorithmic (26 B/row)\n", (long long)rows, gb, gb26);
    printf("%-8s %-4s %-6s %-6s %9s %10s %10s\n", "variant", "nt", "wg/CU", "steps", "ms", "moved GB/s", "frac(26B)");
    const size_t lds_for[] = {150 * 1024, 76 * 1024, 49 * 1024, 38 * 1024, 30 * 1024, 24 * 1024, 19 * 1024};  // 1,2,3,4,5,6,8 WG/CU
    const int wgs[] = {1, 2, 3, 4, 5, 6, 8};
    const int step_list[] = {128, 98, 32};
    for (int si = 0; si < 3; ++si) {
        for (int w = 0; w < 7; ++w) {
            const int steps = step_list[si];
            if (si > 0 && wgs[w] != 3 && wgs[w] != 4 && wgs[w] != 6) continue;
#define ROW(D, NT)                                                                                                 \
    {                                                                                                              \
        const double ms = run<D, NT>(c, rows, steps, lds_for[w], out, s);                                         \
        printf("depth%-3d %-4d %-6d %-6d %9.4f %10.1f %10.4f\n", D, (int)NT, wgs[w], steps, ms, gb / ms * 1e3,     \
               gb26 / ms * 1e3 / 8000.0);                                                                          \
        fflush(stdout);                                                                                            \
    }
            ROW(1, true)
            ROW(2, true)
            ROW(3, true)
            if (si == 0 && wgs[w] == 3) ROW(1, false)
            if (si == 0 && wgs[w] == 3) ROW(2, false)
#undef ROW
        }
    }
    {   // one stream: the four f32 columns are allocated one after the other? no - use the biggest single buffer: ship (8 B/row)
        const int64_t n16 = padded * 8 / 16;  // 16-byte words of the shipdate column
        const double gb1 = (double)n16 * 16 / 1e9;
        for (int steps : {32, 128}) {
            const int64_t per = (int64_t)steps * 256 * 6;
            const unsigned grid = (unsigned)((n16 + per - 1) / per);
            hipEvent_t a, b;
            CHECK(hipEventCreate(&a));
            CHECK(hipEventCreate(&b));
            std::vector<float> ms;
            for (int r = 0; r < 9; ++r) {
                CHECK(hipEventRecord(a, s));
                hipLaunchKernelGGL(k_read_one, dim3(grid), dim3(256), 0, s, (const f32x4*)ship, n16, steps, out);
                CHECK(hipEventRecord(b, s));
                CHECK(hipStreamSynchronize(s));
                float t;
                CHECK(hipEventElapsedTime(&t, a, b));
                if (r >= 2) ms.push_back(t);
            }
            std::sort(ms.begin(), ms.end());
            printf("one stream of %.3f GB, 6 x 16 B per lane in flight, %d steps per workgroup: %9.4f ms %10.1f GB/s (%.4f of 8 TB/s)\n", gb1, steps,
                   ms[ms.size() / 2], gb1 / ms[ms.size() / 2] * 1e3, gb1 / ms[ms.size() / 2] * 1e3 / 8000.0);
        }
    }
    return 0;
}
