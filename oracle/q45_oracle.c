/* ORACLE (test infrastructure - never linked into or called by the product).
 *
 * Plain-C restatements of what the reference's PythonExecutionEngine computes for BASELINE.json configs 4 and 5,
 * row at a time, with the reference's quantisation points - the checkers of the GPU results at sf=10 and the
 * `cpu_baseline` ("kind": "port") legs of `bench.py --config join|strkey`.  Pinned against the reference by
 * tests/test_oracle_golden.py (golden fixtures join_group / concat_like made by the real reference).
 *
 * Config 4:  orders JOIN lineitem ON o_orderkey = l_orderkey GROUP BY o_orderpriority
 *            -> COUNT(), SUM(l_quantity), SUM(l_extendedprice), MAX(l_extendedprice)     (workloads.join_group)
 *   stage 1/2  WriteToShufflePartitions.write   tasks.py:347-375   row -> partition hash(key) % 10 (hash(int) = int,
 *                                                                  hash(-1) = -2, Python floor-mod)
 *   stage 3    one JoinJob per partition         plan.py:99-109
 *     BroadcastHashJoinTask.generate_chunks      tasks.py:201-240   dict key -> left row list (row order); for every
 *                                                                  right row in row order, its left matches ascending
 *     AggregateTask.execute (before_shuffle)     tasks.py:284-289   aggregate arguments per joined row
 *     fill_aggregators                           tasks.py:295-310   SUM from 0, MAX from MIN_INT, in emission order
 *     WriteToShufflePartitions.write             tasks.py:373 -> io.py:87-94   partials stored as f32 / i32
 *   stage 4    AggregateTask.execute (after)     tasks.py:290-292   partials of a key merged in JoinJob order
 *     WriteToLocalFileTask.write                 tasks.py:400-410 -> io.py:94
 *
 * Config 5:  lineitem WHERE l_shipmode LIKE '%AIR%' -> key = l_returnflag + '-' + l_shipmode
 *            GROUP BY key -> SUM(l_quantity), AVG(l_discount), COUNT()                   (workloads.strkey_like)
 *   per block (= ScanJob, plan.py:90-93):
 *     FilterTask.execute                         tasks.py:167-177   LIKE = re.match('^' + escaped pattern + '$'),
 *                                                                  % -> .*, _ -> .          sql.py:178-194
 *     ProjectTask / BinaryOperatorColumn         sql.py:262-266     string '+'
 *     AggregateTask.execute + fill_aggregators   tasks.py:284-310   AVG -> sum + count (sql.py:436-441)
 *     shuffle write f32 / i32, final merge in block order, AVG = sum / count on the un-rounded merged sum
 *                                                plan.py:200-203, then the result write rounds to f32
 *
 * String columns arrive dictionary-coded (a u8 code per row + the dictionary's strings): both synthetic columns
 * have a handful of distinct values; the LIKE matcher runs on the row's actual string.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define Q45_PARTS 10
#define Q45_MAX_CODES 256

/* Python: hash(int) % n for a 32-bit key */
static int py_partition(int64_t key, int n) {
    int64_t h = key == -1 ? -2 : key; /* |key| < 2^61 - 1: hash(int) is the int itself */
    int64_t m = h % n;
    if (m < 0) m += n;
    return (int)m;
}

/* ---- config 4 ------------------------------------------------------------------------------------------------- */
typedef struct q4_row {
    int32_t code; /* dictionary code of o_orderpriority */
    int32_t n;
    double qty, revenue, max_price; /* f32 values widened */
} q4_row;

typedef struct q4_partial {
    uint8_t present[Q45_MAX_CODES];
    int64_t n[Q45_MAX_CODES];
    double qty[Q45_MAX_CODES], revenue[Q45_MAX_CODES], maxp[Q45_MAX_CODES];
    uint8_t max_is_identity[Q45_MAX_CODES];
} q4_partial;

static uint64_t mix64(uint64_t x) {
    x ^= x >> 33;
    x *= 0xff51afd7ed558ccdull;
    x ^= x >> 33;
    x *= 0xc4ceb9fe1a85ec53ull;
    x ^= x >> 33;
    return x;
}

/* one JoinJob: the rows of both inputs whose key falls into partition p */
static void q4_join_job(int p, const int32_t* o_key, const uint8_t* o_code, int64_t n_orders, const int32_t* l_key,
                        const float* l_qty, const float* l_price, int64_t n_li, q4_partial* out) {
    memset(out, 0, sizeof(*out));
    for (int c = 0; c < Q45_MAX_CODES; ++c) out->max_is_identity[c] = 1;
    /* build: key -> chain of left rows, ascending (rows are pushed in descending order) */
    int64_t n_left = 0;
    for (int64_t i = 0; i < n_orders; ++i) n_left += py_partition(o_key[i], Q45_PARTS) == p;
    uint64_t cap = 16;
    while (cap < (uint64_t)n_left * 2 + 2) cap <<= 1;
    int64_t* head = (int64_t*)malloc(sizeof(int64_t) * cap);
    int32_t* slot_key = (int32_t*)malloc(sizeof(int32_t) * cap);
    int64_t* next = (int64_t*)malloc(sizeof(int64_t) * (size_t)(n_orders > 0 ? n_orders : 1));
    for (uint64_t s = 0; s < cap; ++s) head[s] = -1;
    for (int64_t i = n_orders - 1; i >= 0; --i) {
        if (py_partition(o_key[i], Q45_PARTS) != p) continue;
        uint64_t s = mix64((uint64_t)(uint32_t)o_key[i]) & (cap - 1);
        while (head[s] >= 0 && slot_key[s] != o_key[i]) s = (s + 1) & (cap - 1);
        slot_key[s] = o_key[i];
        next[i] = head[s];
        head[s] = i;
    }
    /* probe in right-row order; aggregate the joined rows in emission order */
    for (int64_t r = 0; r < n_li; ++r) {
        const int32_t k = l_key[r];
        if (py_partition(k, Q45_PARTS) != p) continue;
        uint64_t s = mix64((uint64_t)(uint32_t)k) & (cap - 1);
        while (head[s] >= 0 && slot_key[s] != k) s = (s + 1) & (cap - 1);
        for (int64_t l = head[s]; l >= 0; l = next[l]) {
            const int c = o_code[l];
            const double qty = (double)l_qty[r], price = (double)l_price[r];
            out->present[c] = 1;
            out->n[c] = out->n[c] + 1;                 /* COUNT() = SUM(Lit(1)) */
            out->qty[c] = out->qty[c] + qty;           /* SUM(l_quantity) */
            out->revenue[c] = out->revenue[c] + price; /* SUM(l_extendedprice) */
            if (out->max_is_identity[c] ? price > -2147483648.0 : price > out->maxp[c]) { /* max(acc, x), acc from MIN_INT */
                out->maxp[c] = price;
                out->max_is_identity[c] = 0;
            }
        }
    }
    free(head);
    free(slot_key);
    free(next);
}

/* returns the number of groups (ascending code), -1 on i32 overflow at a quantisation point, -3 when a FLOAT MAX never
 * left its int identity (the reference's writer asserts, io.py:93) */
int q4_run(const int32_t* o_key, const uint8_t* o_code, int64_t n_orders, const int32_t* l_key, const float* l_qty,
           const float* l_price, int64_t n_li, int32_t nthreads, q4_row* out) {
    q4_partial* parts = (q4_partial*)malloc(sizeof(q4_partial) * Q45_PARTS);
#pragma omp parallel for schedule(dynamic, 1) num_threads(nthreads) if (nthreads > 1)
    for (int p = 0; p < Q45_PARTS; ++p) q4_join_job(p, o_key, o_code, n_orders, l_key, l_qty, l_price, n_li, &parts[p]);
    int overflow = 0, type_assert = 0, n = 0;
    for (int c = 0; c < Q45_MAX_CODES; ++c) {
        int present = 0, max_identity = 1;
        int64_t cnt = 0;
        double qty = 0.0, rev = 0.0, maxp = 0.0;
        for (int p = 0; p < Q45_PARTS; ++p) { /* shuffle write (f32 / i32) then the merge in JoinJob order */
            if (!parts[p].present[c]) continue;
            present = 1;
            if (parts[p].n[c] > 2147483647ll) overflow = 1;
            if (parts[p].max_is_identity[c]) type_assert = 1;
            cnt = cnt + parts[p].n[c];
            qty = qty + (double)(float)parts[p].qty[c];
            rev = rev + (double)(float)parts[p].revenue[c];
            const double m = (double)(float)parts[p].maxp[c];
            if (max_identity ? m > -2147483648.0 : m > maxp) {
                maxp = m;
                max_identity = 0;
            }
        }
        if (!present) continue;
        if (cnt > 2147483647ll) overflow = 1;
        if (max_identity) type_assert = 1;
        q4_row* r = &out[n++];
        r->code = c;
        r->n = (int32_t)cnt;
        r->qty = (double)(float)qty;
        r->revenue = (double)(float)rev;
        r->max_price = (double)(float)maxp;
    }
    free(parts);
    return overflow ? -1 : (type_assert ? -3 : n);
}

/* ---- config 5 ------------------------------------------------------------------------------------------------- */
/* SQL LIKE as the reference's regex reads it: % = any run not crossing '\n', _ = any one char but '\n' */
static int like_match(const uint8_t* s, int slen, const uint8_t* pat, int plen) {
    int si = 0, pi = 0, star_p = -1, star_s = 0;
    while (si < slen) {
        if (pi < plen && pat[pi] == '%') {
            star_p = pi++;
            star_s = si;
        } else if (pi < plen && ((pat[pi] == '_' && s[si] != '\n') || (pat[pi] != '_' && pat[pi] == s[si]))) {
            ++pi;
            ++si;
        } else if (star_p >= 0 && s[star_s] != '\n') {
            pi = star_p + 1;
            si = ++star_s;
        } else {
            return 0;
        }
    }
    while (pi < plen && pat[pi] == '%') ++pi;
    return pi == plen;
}

typedef struct q5_row {
    int32_t flag; /* the l_returnflag byte */
    int32_t mode; /* dictionary code of l_shipmode: key = flag + "-" + mode string */
    int32_t count;
    int32_t pad;
    double qty, avg_disc; /* f32 values widened */
} q5_row;

typedef struct q5_partial {
    uint8_t present[256][8];
    double qty[256][8], disc[256][8];
    int64_t disc_n[256][8], n[256][8];
} q5_partial;

/* mode_strs: n_modes (<= 8) strings, mode_off[n_modes + 1] offsets into mode_bytes */
int q5_run(const uint8_t* flag, const uint8_t* mode, const float* qty, const float* disc, const int64_t* block_rows,
           int32_t nblocks, const uint8_t* mode_bytes, const int32_t* mode_off, int32_t n_modes, const uint8_t* pat,
           int32_t plen, int32_t nthreads, q5_row* out) {
    if (n_modes > 8) return -2;
    q5_partial* parts = (q5_partial*)malloc(sizeof(q5_partial) * (size_t)(nblocks > 0 ? nblocks : 1));
    int64_t* starts = (int64_t*)malloc(sizeof(int64_t) * (size_t)(nblocks + 1));
    starts[0] = 0;
    for (int b = 0; b < nblocks; ++b) starts[b + 1] = starts[b] + block_rows[b];
#pragma omp parallel for schedule(dynamic, 1) num_threads(nthreads) if (nthreads > 1)
    for (int b = 0; b < nblocks; ++b) {
        q5_partial* P = &parts[b];
        memset(P, 0, sizeof(*P));
        for (int64_t i = starts[b]; i < starts[b + 1]; ++i) {
            const int m = mode[i];
            if (!like_match(mode_bytes + mode_off[m], mode_off[m + 1] - mode_off[m], pat, plen)) continue; /* FilterTask */
            const int f = flag[i];
            P->present[f][m] = 1;
            P->qty[f][m] = P->qty[f][m] + (double)qty[i];
            P->disc[f][m] = P->disc[f][m] + (double)disc[i];
            P->disc_n[f][m] = P->disc_n[f][m] + 1;
            P->n[f][m] = P->n[f][m] + 1;
        }
    }
    int n = 0, overflow = 0;
    for (int f = 0; f < 256; ++f) {
        for (int m = 0; m < n_modes; ++m) {
            int present = 0;
            double q = 0.0, d = 0.0;
            int64_t dn = 0, cnt = 0;
            for (int b = 0; b < nblocks; ++b) {
                if (!parts[b].present[f][m]) continue;
                present = 1;
                if (parts[b].n[f][m] > 2147483647ll) overflow = 1;
                q = q + (double)(float)parts[b].qty[f][m];
                d = d + (double)(float)parts[b].disc[f][m];
                dn = dn + parts[b].disc_n[f][m];
                cnt = cnt + parts[b].n[f][m];
            }
            if (!present) continue;
            if (cnt > 2147483647ll) overflow = 1;
            q5_row* r = &out[n++];
            r->flag = f;
            r->mode = m;
            r->count = (int32_t)cnt;
            r->pad = 0;
            r->qty = (double)(float)q;
            r->avg_disc = (double)(float)(d / (double)dn);
        }
    }
    free(parts);
    free(starts);
    return overflow ? -1 : n;
}

/* CPU twin of the synthetic orders table of config 4 (csrc hs_gen_orders): row j holds key(perm(j)) for a bijection
 * perm of [0, n) (an affine map modulo n with a multiplier coprime to n - build order != key order) and a priority
 * code in [0, 5).  key(o) = 32 * (o / 8) + o % 8 + 1 (sparse like TPC-H: 8 used of every 32). */
static uint64_t splitmix45(uint64_t x) {
    x += 0x9e3779b97f4a7c15ull;
    x = (x ^ (x >> 30)) * 0xbf58476d1ce4e5b9ull;
    x = (x ^ (x >> 27)) * 0x94d049bb133111ebull;
    return x ^ (x >> 31);
}
static uint64_t gcd64(uint64_t a, uint64_t b) {
    while (b) {
        const uint64_t t = a % b;
        a = b;
        b = t;
    }
    return a;
}
uint64_t q4_orders_multiplier(uint64_t seed, int64_t n) {
    if (n <= 1) return 1;
    uint64_t a = splitmix45(seed ^ 0x6f7264657273ull) % (uint64_t)n;
    if (a < 2) a = 2;
    while (gcd64(a, (uint64_t)n) != 1) ++a;
    return a % (uint64_t)n ? a % (uint64_t)n : 1;
}
void q4_gen_orders(uint64_t seed, int64_t row0, int64_t n_rows, int64_t n_total, int32_t* okey, uint8_t* prio_code) {
    const uint64_t a = q4_orders_multiplier(seed, n_total);
    const uint64_t c = splitmix45(seed + 77) % (uint64_t)(n_total > 0 ? n_total : 1);
    for (int64_t k = 0; k < n_rows; ++k) {
        const uint64_t j = (uint64_t)(row0 + k);
        const uint64_t o = (a * j + c) % (uint64_t)n_total; /* a, j < 2^31 (n_total < 2^31): no overflow */
        if (okey) okey[k] = (int32_t)(32 * (o / 8) + (o % 8) + 1);
        if (prio_code) prio_code[k] = (uint8_t)(splitmix45(splitmix45(seed + 0x7072696full) + j) % 5);
    }
}
