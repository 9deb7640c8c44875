// hs_device.h - device-side building blocks shared by the gfx950 kernels of libhipspark:
//   * raw column access (BlockFile storage kinds -> 64-bit in-flight cells),
//   * the expression interpreter (stack program, see include/hipspark.h),
//   * group-key words and the LDS dictionary.
//
// Semantics follow the reference's row evaluator: values are widened to Python float (fp64) /
// Python int (i64 here) when read (src/mini_spark/io.py:129-149), operators are Python's
// (src/mini_spark/sql.py:215-231,262-266), booleans are 0/1.
#pragma once

#include <hip/hip_runtime.h>
#ifndef HS_JIT_BUILD
#include <stdint.h>
#endif

#ifdef HS_JIT_BUILD
/* hiprtc keeps its fixed-width types in a namespace: publish the ones the headers use */
typedef signed char int8_t;
typedef unsigned char uint8_t;
typedef short int16_t;
typedef unsigned short uint16_t;
typedef int int32_t;
typedef unsigned int uint32_t;
typedef long long int64_t;
typedef unsigned long long uint64_t;
#include "hipspark.h" /* supplied to hiprtc as an in-memory header */
#else
#include "../../include/hipspark.h"
#endif

#ifndef HS_JIT_BUILD
#include "hs_capture.h" /* host builds: every launch of the library is capturable */
#endif

#define HS_WAVE 64
#define HS_V 4 /* rows per lane per step in the vectorised kernels */

static constexpr uint64_t HS_EMPTY_KEY = 0x8000000000000000ull;

// Big by-value kernel arguments (programs, column tables) are indexed dynamically.  hipcc copies a
// by-value struct argument into private (scratch) memory as soon as it is indexed with a run-time
// value, so kernels read their single struct argument straight from the kernarg segment instead:
// explicit arguments start at offset 0 of that segment, and uniform reads of it become s_load.
#define HS_KERNARG(T, name) \
    (void)name##_kernarg;   \
    const T& name = *(const T*)__builtin_amdgcn_kernarg_segment_ptr()

// Latency-bound single-workgroup kernels: every first touch of a kernarg cache line is a serialised miss
// (a 2.6 KB argument block = 40 of them).  This variant copies the block into LDS with one cooperative
// vector load at kernel start; afterwards all argument reads are LDS reads.  sizeof(T) % 8 == 0.
#define HS_KERNARG_LDS(T, name)                                                                        \
    (void)name##_kernarg;                                                                              \
    __shared__ __align__(16) uint64_t name##_lds[(sizeof(T) + 7) / 8];                                 \
    {                                                                                                  \
        const uint64_t* name##_src = (const uint64_t*)__builtin_amdgcn_kernarg_segment_ptr();          \
        for (unsigned i_ = threadIdx.x; i_ < (sizeof(T) + 7) / 8; i_ += blockDim.x) name##_lds[i_] = name##_src[i_]; \
    }                                                                                                  \
    __syncthreads();                                                                                   \
    const T& name = *(const T*)name##_lds

#ifndef HS_JIT_BUILD
// Host side: one-time per-DEVICE set-up (function attributes live per device; a process may drive several GPUs).
// `seen` = a static bit set owned by the call site; true the first time the current device comes by.
static inline bool hs_first_on_device(unsigned long long& seen) {
    int d = 0;
    (void)hipGetDevice(&d);
    const unsigned long long bit = 1ull << (d & 63);
    if (seen & bit) return false;
    seen |= bit;
    return true;
}
#endif

struct HsCols {
    int32_t n;
    int32_t pad;
    hs_col c[HS_MAX_COLS];
};

__device__ __forceinline__ double hs_u2d(uint64_t u) { return __longlong_as_double((long long)u); }
__device__ __forceinline__ uint64_t hs_d2u(double d) { return (uint64_t)__double_as_longlong(d); }

// ---- instruction fields --------------------------------------------------------------------------
__device__ __forceinline__ uint32_t hs_ins_op(uint64_t w) { return (uint32_t)(w & 0xff); }
__device__ __forceinline__ uint32_t hs_ins_sp(uint64_t w) { return (uint32_t)((w >> 8) & 0xff); }
__device__ __forceinline__ uint32_t hs_ins_a(uint64_t w) { return (uint32_t)((w >> 16) & 0xffff); }
__device__ __forceinline__ uint32_t hs_ins_b(uint64_t w) { return (uint32_t)((w >> 32) & 0xffff); }
__device__ __forceinline__ uint32_t hs_ins_c(uint64_t w) { return (uint32_t)((w >> 48) & 0xffff); }

// ---- one row of a fixed-width column, widened to a 64-bit cell ---------------------------------------
__device__ __forceinline__ uint64_t hs_load_cell(const hs_col& c, int64_t row) {
    switch (c.kind) {
        case HS_I32: return (uint64_t)(int64_t)((const int32_t*)c.data)[row];
        case HS_F32: return hs_d2u((double)((const float*)c.data)[row]);
        case HS_I64: return (uint64_t)((const int64_t*)c.data)[row];
        case HS_F64: return hs_d2u(((const double*)c.data)[row]);
        case HS_U8: return (uint64_t)((const uint8_t*)c.data)[row];
        default: return 0;
    }
}

// ---- strings -------------------------------------------------------------------------------------
struct HsStr {
    const uint8_t* p;
    uint32_t len;
};

__device__ __forceinline__ HsStr hs_str_at(const hs_col& c, int64_t row) {
    HsStr s;
    if (c.fixed_len >= 0) {
        s.len = (uint32_t)c.fixed_len;
        s.p = (const uint8_t*)c.data + row * (int64_t)c.fixed_len;
    } else {
        s.len = c.lens[row];
        s.p = (const uint8_t*)c.data + c.offs[row];
    }
    return s;
}

// memcmp-style three-way compare (Python str ordering == byte ordering for ASCII)
// The first `len` (<= 16) bytes at p as two little-endian words, zero-padded.  Only ALIGNED 8-byte words that hold at
// least one byte of the string are loaded: such a word lies inside the buffer's allocation (whose start and size are
// multiples of 8), so this touches nothing a byte loop would not be entitled to at word granularity - and replaces up
// to 16 single-byte loads by at most three loads and a funnel shift.
__device__ __forceinline__ void hs_str_words16(const uint8_t* p, uint32_t len, uint64_t& w0, uint64_t& w1) {
    const unsigned long long a = (unsigned long long)p;
    const uint64_t* q = (const uint64_t*)(a & ~7ull);
    const uint32_t lead = (uint32_t)(a & 7ull);
    const uint32_t span = lead + len;  // bytes from q to the end of the string
    const uint64_t x0 = len ? q[0] : 0ull;
    const uint64_t x1 = span > 8 ? q[1] : 0ull;
    const uint64_t x2 = span > 16 ? q[2] : 0ull;
    const uint32_t sh = lead * 8;
    w0 = sh ? (x0 >> sh) | (x1 << (64 - sh)) : x0;
    w1 = sh ? (x1 >> sh) | (x2 << (64 - sh)) : x1;
    if (len < 8) {
        w0 = len ? w0 & (~0ull >> (64 - 8 * len)) : 0ull;
        w1 = 0ull;
    } else if (len < 16) {
        w1 = len > 8 ? w1 & (~0ull >> (64 - 8 * (len - 8))) : 0ull;
    }
}

__device__ __forceinline__ int hs_str_cmp(HsStr a, HsStr b) {
    uint32_t n = a.len < b.len ? a.len : b.len;
    for (uint32_t i = 0; i < n; ++i) {
        int d = (int)a.p[i] - (int)b.p[i];
        if (d) return d;
    }
    return (int)a.len - (int)b.len;
}

__device__ __forceinline__ bool hs_cmp_result(int d, uint32_t cmp) {
    switch (cmp) {
        case 0: return d < 0;
        case 1: return d <= 0;
        case 2: return d > 0;
        case 3: return d >= 0;
        case 4: return d == 0;
        default: return d != 0;
    }
}

// SQL LIKE with % (any run) and _ (any one byte), anchored both ends; iterative with one
// backtrack point (classic wildcard matcher).  Reference: re.match("^...$") on the translated
// pattern, sql.py:178-179,192-194.  '.' does not match '\n' in the reference's regex.
struct HsBytesMem {  // the string's bytes where they lie
    const uint8_t* p;
    __device__ __forceinline__ uint8_t operator[](uint32_t i) const { return p[i]; }
};
struct HsBytesReg {  // a string of <= 16 bytes held in two registers (hs_str_words16)
    uint64_t w0, w1;
    __device__ __forceinline__ uint8_t operator[](uint32_t i) const {
        return (uint8_t)(i < 8 ? w0 >> (8 * i) : w1 >> (8 * (i - 8)));
    }
};
template <class Bytes>
__device__ __forceinline__ bool hs_like_impl(const Bytes& b, uint32_t slen, const uint8_t* pat, uint32_t plen) {
    uint32_t si = 0, pi = 0;
    int64_t star_p = -1;
    uint32_t star_s = 0;
    while (si < slen) {
        if (pi < plen && pat[pi] == '%') {
            star_p = pi++;
            star_s = si;
        } else if (pi < plen && ((pat[pi] == '_' && b[si] != '\n') || (pat[pi] != '_' && pat[pi] == b[si]))) {
            ++pi;
            ++si;
        } else if (star_p >= 0 && b[star_s] != '\n') {
            pi = (uint32_t)star_p + 1;
            si = ++star_s;
        } else {
            return false;
        }
    }
    while (pi < plen && pat[pi] == '%') ++pi;
    return pi == plen;
}
__device__ __forceinline__ bool hs_like(HsStr s, const uint8_t* pat, uint32_t plen) {
    if (s.len <= 16) {  // the matcher re-reads bytes while backtracking: keep short strings in registers
        HsBytesReg r;
        hs_str_words16(s.p, s.len, r.w0, r.w1);
        return hs_like_impl(r, s.len, pat, plen);
    }
    return hs_like_impl(HsBytesMem{s.p}, s.len, pat, plen);
}

// ---- Python arithmetic -----------------------------------------------------------------------------
// b == -1 is answered without dividing: the compiler narrows a 64-bit division whose operands are sign-extended 32-bit
// values to a 32-bit one (so GROUP BY k % 97 costs no 64-bit division), and that narrowing turns INT32_MIN // -1 into
// INT32_MIN - Python's answer is 2147483648 (which then fails the i32 write with OverflowError, io.py:90).
__device__ __forceinline__ int64_t hs_floordiv_i(int64_t a, int64_t b) {
    if (b == -1) return -a;
    int64_t q = a / b;
    if ((a % b != 0) && ((a < 0) != (b < 0))) --q;
    return q;
}
__device__ __forceinline__ int64_t hs_mod_i(int64_t a, int64_t b) {
    if (b == -1) return 0;
    int64_t m = a % b;
    if (m != 0 && ((m < 0) != (b < 0))) m += b;
    return m;
}
// CPython float_divmod (Objects/floatobject.c)
__device__ __forceinline__ void hs_divmod_f(double vx, double wx, double& floordiv, double& mod) {
    mod = fmod(vx, wx);
    double div = (vx - mod) / wx;
    if (mod != 0.0) {
        if ((wx < 0) != (mod < 0)) {
            mod += wx;
            div -= 1.0;
        }
    } else {
        mod = copysign(0.0, wx);
    }
    if (div != 0.0) {
        floordiv = floor(div);
        if (div - floordiv > 0.5) floordiv += 1.0;
    } else {
        floordiv = copysign(0.0, vx / wx);
    }
}

// ---- binary operators: ONE definition shared by the interpreter and by JIT-generated code -------------------
// x = second-from-top cell, y = top cell; `live` gates data-dependent error reporting (rows that failed
// the WHERE clause are never evaluated by the reference and must not raise).
template <int OP>
__device__ __forceinline__ uint64_t hs_bin(uint64_t xs, uint64_t ys, bool live, uint32_t& err) {
    if constexpr (OP == HS_OP_ADD_F) return hs_d2u(hs_u2d(xs) + hs_u2d(ys));
    else if constexpr (OP == HS_OP_SUB_F) return hs_d2u(hs_u2d(xs) - hs_u2d(ys));
    else if constexpr (OP == HS_OP_MUL_F) return hs_d2u(hs_u2d(xs) * hs_u2d(ys));
    else if constexpr (OP == HS_OP_DIV_F) {
        if (hs_u2d(ys) == 0.0 && live) err |= HS_FLAG_DIV_ZERO;
        return hs_d2u(hs_u2d(xs) / hs_u2d(ys));
    } else if constexpr (OP == HS_OP_FLOORDIV_F || OP == HS_OP_MOD_F) {
        double fd = 0.0, md = 0.0;
        if (hs_u2d(ys) == 0.0) {
            if (live) err |= HS_FLAG_DIV_ZERO;
        } else {
            hs_divmod_f(hs_u2d(xs), hs_u2d(ys), fd, md);
        }
        return hs_d2u(OP == HS_OP_FLOORDIV_F ? fd : md);
    } else if constexpr (OP == HS_OP_ADD_I) return (uint64_t)((int64_t)xs + (int64_t)ys);
    else if constexpr (OP == HS_OP_SUB_I) return (uint64_t)((int64_t)xs - (int64_t)ys);
    else if constexpr (OP == HS_OP_MUL_I) return (uint64_t)((int64_t)xs * (int64_t)ys);
    else if constexpr (OP == HS_OP_FLOORDIV_I || OP == HS_OP_MOD_I) {
        int64_t r = 0;
        if ((int64_t)ys == 0) {
            if (live) err |= HS_FLAG_DIV_ZERO;
        } else {
            r = OP == HS_OP_FLOORDIV_I ? hs_floordiv_i((int64_t)xs, (int64_t)ys) : hs_mod_i((int64_t)xs, (int64_t)ys);
        }
        return (uint64_t)r;
    } else if constexpr (OP == HS_OP_LT_F) return (uint64_t)(hs_u2d(xs) < hs_u2d(ys));
    else if constexpr (OP == HS_OP_LE_F) return (uint64_t)(hs_u2d(xs) <= hs_u2d(ys));
    else if constexpr (OP == HS_OP_GT_F) return (uint64_t)(hs_u2d(xs) > hs_u2d(ys));
    else if constexpr (OP == HS_OP_GE_F) return (uint64_t)(hs_u2d(xs) >= hs_u2d(ys));
    else if constexpr (OP == HS_OP_EQ_F) return (uint64_t)(hs_u2d(xs) == hs_u2d(ys));
    else if constexpr (OP == HS_OP_NE_F) return (uint64_t)(hs_u2d(xs) != hs_u2d(ys));
    else if constexpr (OP == HS_OP_LT_I) return (uint64_t)((int64_t)xs < (int64_t)ys);
    else if constexpr (OP == HS_OP_LE_I) return (uint64_t)((int64_t)xs <= (int64_t)ys);
    else if constexpr (OP == HS_OP_GT_I) return (uint64_t)((int64_t)xs > (int64_t)ys);
    else if constexpr (OP == HS_OP_GE_I) return (uint64_t)((int64_t)xs >= (int64_t)ys);
    else if constexpr (OP == HS_OP_EQ_I) return (uint64_t)(xs == ys);
    else if constexpr (OP == HS_OP_NE_I) return (uint64_t)(xs != ys);
    else if constexpr (OP == HS_OP_AND) return xs & ys;
    else if constexpr (OP == HS_OP_OR) return xs | ys;
    else return 0;
}

// string predicates, shared likewise
__device__ __forceinline__ uint64_t hs_strcmp_lit(const hs_program& P, const hs_col& c, int64_t row, uint32_t lit_idx,
                                                  uint32_t cmp) {
    const uint64_t ref = P.lit[lit_idx];
    HsStr lit;
    lit.p = P.pool + (uint32_t)(ref >> 32);
    lit.len = (uint32_t)ref;
    return (uint64_t)hs_cmp_result(hs_str_cmp(hs_str_at(c, row), lit), cmp);
}
__device__ __forceinline__ uint64_t hs_strcmp_col(const hs_col& a, const hs_col& b, int64_t row, uint32_t cmp) {
    return (uint64_t)hs_cmp_result(hs_str_cmp(hs_str_at(a, row), hs_str_at(b, row)), cmp);
}
__device__ __forceinline__ uint64_t hs_like_lit(const hs_program& P, const hs_col& c, int64_t row, uint32_t lit_idx) {
    const uint64_t ref = P.lit[lit_idx];
    return (uint64_t)hs_like(hs_str_at(c, row), P.pool + (uint32_t)(ref >> 32), (uint32_t)ref);
}

// bit[code] of a dictionary-coded column: the literal words lit[first .. first + n_words) hold one bit per entry
__device__ __forceinline__ uint64_t hs_dictbit_code(const hs_program& P, uint32_t code, uint32_t first, uint32_t n_words) {
    const uint32_t w = code >> 6;
    return w < n_words ? (P.lit[first + w] >> (code & 63u)) & 1ull : 0ull;
}
__device__ __forceinline__ uint64_t hs_dictbit(const hs_program& P, const hs_col& c, int64_t row, uint32_t first,
                                               uint32_t n_words) {
    return hs_dictbit_code(P, ((const uint8_t*)c.data)[row], first, n_words);
}

// ---- interpreter ---------------------------------------------------------------------------------
// The program is straight-line, so the stack depth before every instruction is known when it is
// lowered and travels in the instruction (sp).  Dispatch is two wave-uniform switches (sp, then op)
// whose bodies index the stack with compile-time constants: the stack lives in VGPRs, nothing is
// spilled and nothing diverges.
//
// Sink concept:
//   uint64_t load(uint32_t slot, int j)               value of column slot for row j (already widened)
//   bool     live(int j)                              row j still takes part (valid & passed filters)
//   void     filter(int j, bool keep)
//   void     agg(uint32_t acc, int j, uint64_t cell)
//   void     out(uint32_t o, int j, uint64_t cell)
//   int64_t  row(int j)                               row index (string ops read memory with it)
//   void     key()                                    HS_OP_KEY: resolve the group slot of live rows


template <int SP, int D, int V, typename Sink>
__device__ __forceinline__ void hs_exec_at(uint64_t w, const hs_program& P, const HsCols& C, uint64_t (&st)[D][V],
                                           Sink& sink, uint32_t& err) {
    const uint32_t op = hs_ins_op(w);
    const uint32_t a = hs_ins_a(w);
    constexpr int T = SP >= 1 ? SP - 1 : 0;  // top
    constexpr int S = SP >= 2 ? SP - 2 : 0;  // second
    constexpr int N = SP < D ? SP : D - 1;   // next free

#define HS_BIN(OPC)                                                                  \
    case OPC:                                                                        \
        if constexpr (SP >= 2) {                                                     \
            _Pragma("unroll") for (int j = 0; j < V; ++j)                            \
                st[S][j] = hs_bin<OPC>(st[S][j], st[T][j], sink.live(j), err);       \
        }                                                                            \
        break;

    switch (op) {
        case HS_OP_LD:
            if constexpr (SP < D) {
#pragma unroll
                for (int j = 0; j < V; ++j) st[N][j] = sink.load(a, j);
            }
            break;
        case HS_OP_LIT:
            if constexpr (SP < D) {
                const uint64_t lit = P.lit[a];
#pragma unroll
                for (int j = 0; j < V; ++j) st[N][j] = lit;
            }
            break;
        HS_BIN(HS_OP_ADD_F) HS_BIN(HS_OP_SUB_F) HS_BIN(HS_OP_MUL_F) HS_BIN(HS_OP_DIV_F)
        HS_BIN(HS_OP_FLOORDIV_F) HS_BIN(HS_OP_MOD_F)
        HS_BIN(HS_OP_ADD_I) HS_BIN(HS_OP_SUB_I) HS_BIN(HS_OP_MUL_I) HS_BIN(HS_OP_FLOORDIV_I) HS_BIN(HS_OP_MOD_I)
        HS_BIN(HS_OP_LT_F) HS_BIN(HS_OP_LE_F) HS_BIN(HS_OP_GT_F) HS_BIN(HS_OP_GE_F) HS_BIN(HS_OP_EQ_F) HS_BIN(HS_OP_NE_F)
        HS_BIN(HS_OP_LT_I) HS_BIN(HS_OP_LE_I) HS_BIN(HS_OP_GT_I) HS_BIN(HS_OP_GE_I) HS_BIN(HS_OP_EQ_I) HS_BIN(HS_OP_NE_I)
        HS_BIN(HS_OP_AND) HS_BIN(HS_OP_OR)
        case HS_OP_I2F:
            if constexpr (SP >= 1) {
                if (a == 0) {
#pragma unroll
                    for (int j = 0; j < V; ++j) st[T][j] = hs_d2u((double)(int64_t)st[T][j]);
                } else if constexpr (SP >= 2) {
#pragma unroll
                    for (int j = 0; j < V; ++j) st[S][j] = hs_d2u((double)(int64_t)st[S][j]);
                }
            }
            break;
        case HS_OP_STRCMP_LIT:
            if constexpr (SP < D) {
#pragma unroll
                for (int j = 0; j < V; ++j)
                    st[N][j] = sink.live(j) ? hs_strcmp_lit(P, C.c[a], sink.row(j), hs_ins_b(w), hs_ins_c(w)) : 0;
            }
            break;
        case HS_OP_STRCMP_COL:
            if constexpr (SP < D) {
#pragma unroll
                for (int j = 0; j < V; ++j)
                    st[N][j] = sink.live(j) ? hs_strcmp_col(C.c[a], C.c[hs_ins_b(w)], sink.row(j), hs_ins_c(w)) : 0;
            }
            break;
        case HS_OP_LIKE:
            if constexpr (SP < D) {
#pragma unroll
                for (int j = 0; j < V; ++j)
                    st[N][j] = sink.live(j) ? hs_like_lit(P, C.c[a], sink.row(j), hs_ins_b(w)) : 0;
            }
            break;
        case HS_OP_DICTBIT:
            if constexpr (SP < D) {
#pragma unroll
                for (int j = 0; j < V; ++j)
                    st[N][j] = sink.live(j) ? hs_dictbit(P, C.c[a], sink.row(j), hs_ins_b(w), hs_ins_c(w)) : 0;
            }
            break;
        case HS_OP_FILTER:
            if constexpr (SP >= 1) {
#pragma unroll
                for (int j = 0; j < V; ++j) sink.filter(j, st[T][j] != 0);
            }
            break;
        case HS_OP_AGG:
            if constexpr (SP >= 1) {
#pragma unroll
                for (int j = 0; j < V; ++j) sink.agg(a, j, st[T][j]);
            }
            break;
        case HS_OP_OUT:
            if constexpr (SP >= 1) {
#pragma unroll
                for (int j = 0; j < V; ++j) sink.out(a, j, st[T][j]);
            }
            break;
        case HS_OP_KEY: sink.key(); break;
        default: err |= HS_FLAG_BAD_PROGRAM; break;
    }
#undef HS_BIN
}

template <int D, int V, typename Sink>
__device__ __forceinline__ void hs_run(const hs_program& P, const HsCols& C, uint32_t first, uint32_t last, Sink& sink,
                                       uint32_t& err) {
    uint64_t st[D][V];
#pragma unroll
    for (int d = 0; d < D; ++d)
#pragma unroll
        for (int j = 0; j < V; ++j) st[d][j] = 0;
    for (uint32_t pc = first; pc < last; ++pc) {
        const uint64_t w = P.ins[pc];
        const uint32_t sp = hs_ins_sp(w);
        switch (sp) {
            case 0: hs_exec_at<0, D, V>(w, P, C, st, sink, err); break;
            case 1: hs_exec_at<1, D, V>(w, P, C, st, sink, err); break;
            case 2: hs_exec_at<2, D, V>(w, P, C, st, sink, err); break;
            case 3: hs_exec_at<3, D, V>(w, P, C, st, sink, err); break;
            case 4: hs_exec_at<4, D, V>(w, P, C, st, sink, err); break;
            default:
                if constexpr (D > 4) {
                    switch (sp) {
                        case 5: hs_exec_at<5, D, V>(w, P, C, st, sink, err); break;
                        case 6: hs_exec_at<6, D, V>(w, P, C, st, sink, err); break;
                        case 7: hs_exec_at<7, D, V>(w, P, C, st, sink, err); break;
                        case 8: hs_exec_at<8, D, V>(w, P, C, st, sink, err); break;
                        default: err |= HS_FLAG_BAD_PROGRAM; break;
                    }
                } else {
                    err |= HS_FLAG_BAD_PROGRAM;
                }
                break;
        }
    }
}

// Compact interpreter, one row: a single op switch over a run-time indexed stack.  A fraction of the code
// of hs_run (which instantiates every operator per stack depth to keep the stack in registers): for the
// latency-bound one-workgroup kernels, where each first visit of an instruction-cache line is a serialised
// miss and throughput is irrelevant.  Same operators (hs_bin<>), same sink concept (j = 0).
// The stack lives in LDS (a dynamically indexed private array would go to scratch = global memory latency per
// access): `stack` = this lane's first cell, entries `stride` cells apart, HS_MAX_STACK + 1 of them.
template <typename Sink>
__device__ __forceinline__ void hs_run_compact(const hs_program& P, const HsCols& C, Sink& sink, uint32_t& err,
                                               uint64_t* stack, int stride) {
#define st(d) stack[(int)(d) * stride]
    const uint32_t n_ins = P.n_ins;
    for (uint32_t pc = 0; pc < n_ins; ++pc) {
        const uint64_t w = P.ins[pc];
        const uint32_t op = hs_ins_op(w), sp = hs_ins_sp(w), a = hs_ins_a(w);
        if (sp > HS_MAX_STACK) {
            err |= HS_FLAG_BAD_PROGRAM;
            return;
        }
        const uint32_t t = sp >= 1 ? sp - 1 : 0, s2 = sp >= 2 ? sp - 2 : 0, nx = sp < HS_MAX_STACK ? sp : HS_MAX_STACK;
        const uint64_t x = st(s2), y = st(t);
        const bool live = sink.live(0);
#define HS_CBIN(OPC) case OPC: st(s2) = hs_bin<OPC>(x, y, live, err); break;
        switch (op) {
            case HS_OP_LD: st(nx) = sink.load(a, 0); break;
            case HS_OP_LIT: st(nx) = P.lit[a]; break;
            HS_CBIN(HS_OP_ADD_F) HS_CBIN(HS_OP_SUB_F) HS_CBIN(HS_OP_MUL_F) HS_CBIN(HS_OP_DIV_F)
            HS_CBIN(HS_OP_FLOORDIV_F) HS_CBIN(HS_OP_MOD_F)
            HS_CBIN(HS_OP_ADD_I) HS_CBIN(HS_OP_SUB_I) HS_CBIN(HS_OP_MUL_I) HS_CBIN(HS_OP_FLOORDIV_I) HS_CBIN(HS_OP_MOD_I)
            HS_CBIN(HS_OP_LT_F) HS_CBIN(HS_OP_LE_F) HS_CBIN(HS_OP_GT_F) HS_CBIN(HS_OP_GE_F) HS_CBIN(HS_OP_EQ_F) HS_CBIN(HS_OP_NE_F)
            HS_CBIN(HS_OP_LT_I) HS_CBIN(HS_OP_LE_I) HS_CBIN(HS_OP_GT_I) HS_CBIN(HS_OP_GE_I) HS_CBIN(HS_OP_EQ_I) HS_CBIN(HS_OP_NE_I)
            HS_CBIN(HS_OP_AND) HS_CBIN(HS_OP_OR)
            case HS_OP_I2F:
                if (a == 0) st(t) = hs_d2u((double)(int64_t)y);
                else st(s2) = hs_d2u((double)(int64_t)x);
                break;
            case HS_OP_STRCMP_LIT: st(nx) = live ? hs_strcmp_lit(P, C.c[a], sink.row(0), hs_ins_b(w), hs_ins_c(w)) : 0; break;
            case HS_OP_STRCMP_COL: st(nx) = live ? hs_strcmp_col(C.c[a], C.c[hs_ins_b(w)], sink.row(0), hs_ins_c(w)) : 0; break;
            case HS_OP_LIKE: st(nx) = live ? hs_like_lit(P, C.c[a], sink.row(0), hs_ins_b(w)) : 0; break;
            case HS_OP_DICTBIT: st(nx) = live ? hs_dictbit(P, C.c[a], sink.row(0), hs_ins_b(w), hs_ins_c(w)) : 0; break;
            case HS_OP_FILTER: sink.filter(0, y != 0); break;
            case HS_OP_AGG: sink.agg(a, 0, y); break;
            case HS_OP_OUT: sink.out(a, 0, y); break;
            case HS_OP_KEY: sink.key(); break;
            default: err |= HS_FLAG_BAD_PROGRAM; break;
        }
#undef HS_CBIN
    }
#undef st
}

// ---- group keys ------------------------------------------------------------------------------------
// A key word identifies a group within one launch.  Numeric keys are their own word; a string of <= 7
// bytes is packed (bytes little-endian, length in the top byte) and so is exact.  Longer strings use
// "hashed mode": the word only picks the probe start, equality is decided by comparing the bytes of
// the candidate row with the slot's representative row.
__device__ __forceinline__ uint64_t hs_mix64(uint64_t x) {
    x ^= x >> 33;
    x *= 0xff51afd7ed558ccdull;
    x ^= x >> 33;
    x *= 0xc4ceb9fe1a85ec53ull;
    x ^= x >> 33;
    return x;
}

__device__ __forceinline__ uint64_t hs_fnv1a(const uint8_t* p, uint32_t n) {
    uint64_t h = 0xcbf29ce484222325ull;
    for (uint32_t i = 0; i < n; ++i) {
        h ^= p[i];
        h *= 0x100000001b3ull;
    }
    return h;
}

// numeric cell (already widened) -> key word.  Python: 0.0 == -0.0 are one dict key.
__device__ __forceinline__ uint64_t hs_key_from_cell(int32_t kind, uint64_t cell) {
    if (kind == HS_F32 || kind == HS_F64) {
        if (hs_u2d(cell) == 0.0) cell = 0;
    }
    return cell == HS_EMPTY_KEY ? HS_EMPTY_KEY + 1 : cell;
}

__device__ __forceinline__ uint64_t hs_pack_str(const uint8_t* p, uint32_t len) {
    uint64_t k = (uint64_t)len << 56;
    for (uint32_t i = 0; i < len; ++i) k |= (uint64_t)p[i] << (8 * i);
    return k;
}

// true when every string of the column packs exactly (<= 7 bytes)
__host__ __device__ __forceinline__ bool hs_col_packs(const hs_col& c) {
    return c.kind != HS_STR || (c.fixed_len >= 0 && c.fixed_len <= 7);
}

__device__ __forceinline__ uint64_t hs_key_at(const hs_col& c, int64_t row) {
    if (c.kind == HS_STR) {
        HsStr s = hs_str_at(c, row);
        if (s.len <= 7) return hs_pack_str(s.p, s.len);
        uint64_t h;
        if (s.len <= 16) {
            uint64_t w0, w1;
            hs_str_words16(s.p, s.len, w0, w1);
            h = hs_mix64(w0 ^ hs_mix64(w1 + s.len));
        } else {
            h = hs_fnv1a(s.p, s.len);
        }
        return (h & 0x3fffffffffffffffull) | 0x4000000000000000ull;
    }
    return hs_key_from_cell(c.kind, hs_load_cell(c, row));
}

__device__ __forceinline__ bool hs_rows_equal(const hs_col& c, int64_t r0, int64_t r1) {
    if (c.kind == HS_STR) {
        const HsStr a = hs_str_at(c, r0), b = hs_str_at(c, r1);
        if (a.len != b.len) return false;
        if (a.len <= 16) {
            uint64_t a0, a1, b0, b1;
            hs_str_words16(a.p, a.len, a0, a1);
            hs_str_words16(b.p, b.len, b0, b1);
            return a0 == b0 && a1 == b1;
        }
        return hs_str_cmp(a, b) == 0;
    }
    return hs_key_at(c, r0) == hs_key_at(c, r1);
}

// probe start for the small LDS dictionaries: one 32-bit multiply (the 64-bit mixer above is for the
// big global tables)
__device__ __forceinline__ uint32_t hs_slot_hash(uint64_t k) {
    return ((uint32_t)k ^ (uint32_t)(k >> 32)) * 0x9E3779B1u >> 12;
}
// the same for dictionaries with hundreds of slots (shared-dictionary tier): key words often carry their
// information in few bits (the fp64 patterns of 1.0 .. 50.0 differ only in bits 48-62) and a single multiply
// clusters them - with ~25 probes per lookup the kernel ran 8x slower; here the full 64-bit mixer
__device__ __forceinline__ uint32_t hs_slot_hash_strong(uint64_t k) { return (uint32_t)(hs_mix64(k) >> 20); }

// LDS dictionary, exact-word mode.  keys[] initialised to HS_EMPTY_KEY, reps[] to -1.
// Returns the slot of `k`, inserting it if absent; -1 when the table is full.
__device__ __forceinline__ int hs_dict_upsert_word_at(uint64_t* keys, int64_t* reps, uint32_t mask, uint64_t k,
                                                      int64_t row, uint32_t h, bool& inserted,
                                                      uint32_t step = 1) {
    h &= mask;
    inserted = false;
    for (uint32_t probe = 0; probe <= mask; ++probe) {
        // relaxed atomic load, NOT a volatile read: hipcc leaves volatile accesses on the generic address space
        // (flat_load + s_waitcnt vmcnt(0), which also drains the prefetched column loads); this one becomes ds_read_b64
        uint64_t cur = __hip_atomic_load(&keys[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (cur == HS_EMPTY_KEY) {
            cur = atomicCAS((unsigned long long*)&keys[h], (unsigned long long)HS_EMPTY_KEY, (unsigned long long)k);
            if (cur == HS_EMPTY_KEY) {
                reps[h] = row;
                inserted = true;
                return (int)h;
            }
        }
        if (cur == k) return (int)h;
        h = (h + step) & mask;  // step odd: every slot of the power-of-two table is visited
    }
    return -1;
}
__device__ __forceinline__ int hs_dict_upsert_word_at(uint64_t* keys, int64_t* reps, uint32_t mask, uint64_t k,
                                                      int64_t row, uint32_t h) {
    bool inserted;
    return hs_dict_upsert_word_at(keys, reps, mask, k, row, h, inserted);
}

__device__ __forceinline__ int hs_dict_upsert_word(uint64_t* keys, int64_t* reps, uint32_t mask, uint64_t k,
                                                   int64_t row) {
    return hs_dict_upsert_word_at(keys, reps, mask, k, row, hs_slot_hash(k));
}

// LDS dictionary, hashed mode: the slot is claimed by CAS on its representative row; equality is a
// byte compare against that row (immutable global memory, so no ordering hazard).
__device__ __forceinline__ int hs_dict_upsert_rows_at(int64_t* reps, uint32_t mask, const hs_col& c, uint64_t k,
                                                      int64_t row, uint32_t h, bool& inserted,
                                                      uint32_t step = 1) {
    (void)k;
    h &= mask;
    inserted = false;
    for (uint32_t probe = 0; probe <= mask; ++probe) {
        long long cur = __hip_atomic_load(&reps[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (cur < 0) {
            cur = (long long)atomicCAS((unsigned long long*)&reps[h], (unsigned long long)(-1ll),
                                       (unsigned long long)row);
            if (cur < 0) {
                inserted = true;
                return (int)h;
            }
        }
        if (hs_rows_equal(c, (int64_t)cur, row)) return (int)h;
        h = (h + step) & mask;
    }
    return -1;
}
__device__ __forceinline__ int hs_dict_upsert_rows_at(int64_t* reps, uint32_t mask, const hs_col& c, uint64_t k,
                                                      int64_t row, uint32_t h) {
    bool inserted;
    return hs_dict_upsert_rows_at(reps, mask, c, k, row, h, inserted);
}
__device__ __forceinline__ int hs_dict_upsert_rows(int64_t* reps, uint32_t mask, const hs_col& c, uint64_t k,
                                                   int64_t row) {
    return hs_dict_upsert_rows_at(reps, mask, c, k, row, hs_slot_hash(k));
}

// ---- accumulator folding -----------------------------------------------------------------------------
// The aggregate description as two scalars: an op nibble and an "integer accumulator" bit per accumulator.  Kernels index
// the description with run-time values, and gfx9 has no scalar byte load: `spec.op[a]` read from the kernarg segment is a
// VECTOR load + s_waitcnt vmcnt(0) at every use.  In the scan's table initialisation that wait drained the first step's
// column loads 24 times per workgroup (5-8 us on an idle chip, 40+ us next to streaming workgroups:
// profiles/r04_scan_stamps_before.txt).  Built once per kernel from nine s_load_dword; every later use is SALU / VALU.
struct HsSpecBits {
    uint64_t ops;   // 4 bits per accumulator: HS_AGG_*
    uint32_t ints;  // bit a: accumulator a is an i64
};
__device__ __forceinline__ HsSpecBits hs_spec_bits(const hs_agg_spec& s) {
    static_assert(HS_MAX_ACC == 16 && sizeof(hs_agg_spec) == 36, "hs_spec_bits packs 16 accumulators");
    const uint32_t* w = reinterpret_cast<const uint32_t*>(&s);  // [0] n_acc, [1..4] op bytes, [5..8] is_int bytes
    HsSpecBits b{0, 0};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const uint32_t o = w[1 + i], t = w[5 + i];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            b.ops |= (uint64_t)((o >> (8 * k)) & 0xfu) << (4 * (4 * i + k));
            b.ints |= (((t >> (8 * k)) & 0xffu) != 0 ? 1u : 0u) << (4 * i + k);
        }
    }
    return b;
}
__device__ __forceinline__ uint32_t hs_spec_op(const HsSpecBits& b, uint32_t a) { return (uint32_t)(b.ops >> (4 * a)) & 0xfu; }
__device__ __forceinline__ bool hs_spec_int(const HsSpecBits& b, uint32_t a) { return (b.ints >> a) & 1u; }

__device__ __forceinline__ uint64_t hs_acc_identity(uint32_t op, bool is_int) {
    if (op == HS_AGG_SUM) return is_int ? 0ull : hs_d2u(0.0);
    const int64_t id = (op == HS_AGG_MIN) ? 2147483647ll : -2147483648ll;  // MAX_INT / MIN_INT, constants.py:14-15
    return is_int ? (uint64_t)id : hs_d2u((double)id);
}

__device__ __forceinline__ uint64_t hs_acc_fold(uint32_t op, bool is_int, uint64_t acc, uint64_t x) {
    if (is_int) {
        int64_t a = (int64_t)acc, b = (int64_t)x;
        if (op == HS_AGG_SUM) return (uint64_t)(a + b);
        if (op == HS_AGG_MIN) return (uint64_t)(b < a ? b : a);
        return (uint64_t)(b > a ? b : a);
    }
    double a = hs_u2d(acc), b = hs_u2d(x);
    if (op == HS_AGG_SUM) return hs_d2u(a + b);
    if (op == HS_AGG_MIN) return hs_d2u(b < a ? b : a);  // Python min(acc, x): x only if strictly smaller
    return hs_d2u(b > a ? b : a);
}

// Python: min(MAX_INT, x) keeps the int MAX_INT unless x < MAX_INT; a FLOAT aggregate that still holds the int
// identity makes the reference's writer fail its type assertion (io.py:93).  True when `cell` is that case.
__device__ __forceinline__ bool hs_float_identity_left(uint32_t op, bool is_int, uint64_t cell) {
    return !is_int && op != HS_AGG_SUM && cell == hs_acc_identity(op, false);
}

// fp64 -> "what a shuffle/result file holds": f32 rounding (RNE) widened back; finite overflow flagged
__device__ __forceinline__ uint64_t hs_quantise_cell(bool is_int, uint64_t cell, uint32_t& err) {
    if (is_int) {
        int64_t v = (int64_t)cell;
        if (v > 2147483647ll || v < -2147483648ll) err |= HS_FLAG_INT_OVERFLOW;
        return cell;
    }
    double d = hs_u2d(cell);
    float f = (float)d;
    if (isinf(f) && !isinf(d)) err |= HS_FLAG_FLT_OVERFLOW;
    return hs_d2u((double)f);
}

// ---- the join's byte table (include/hipspark.h hs_join8): used by the build kernels and by the fused probe that the
// run-time compiler splices into the aggregate scan -----------------------------------------------------------------
// Python: hash(int) % n (hash(-1) = -2, floor-mod) - the reference's shuffle partition of a row, tasks.py:362
__device__ __forceinline__ uint32_t hs_py_partition(int32_t key, int32_t n_parts) {
    const int32_t h = key == -1 ? -2 : key;
    int32_t m = h % n_parts;
    if (m < 0) m += n_parts;
    return (uint32_t)m;
}
// The same without an integer division (a 32-bit signed % by a run-time divisor is ~40 instructions, several of them
// quarter-rate multiplies; the fused probe computes a partition per distinct key of every lane): the quotient comes from
// a double-precision multiply by inv = 1.0 / n_parts - |h| < 2^32 and a relative error of 2^-52 leave it off by at most
// one, which the remainder shows.
__device__ __forceinline__ uint32_t hs_py_partition_inv(int32_t key, int32_t n_parts, double inv) {
    const int32_t h = key == -1 ? -2 : key;
    const uint32_t a = h < 0 ? 0u - (uint32_t)h : (uint32_t)h;
    const uint32_t q = (uint32_t)((double)a * inv);
    uint32_t r = a - q * (uint32_t)n_parts;
    if ((int32_t)r < 0) r += (uint32_t)n_parts;
    else if (r >= (uint32_t)n_parts) r -= (uint32_t)n_parts;
    return (h < 0 && r != 0) ? (uint32_t)n_parts - r : r;  // Python's % takes the divisor's sign
}
// table byte of `key`: the build side's payload, 0xff = no build row has this key
__device__ __forceinline__ uint32_t hs_join8_lookup(const hs_join8& J, int32_t key) {
    const int64_t off = (int64_t)key - (int64_t)J.key_min;
    return (uint64_t)off < (uint64_t)J.slots ? (uint32_t)J.table[off] : 0xffu;
}
