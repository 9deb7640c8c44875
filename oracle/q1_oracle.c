/* ORACLE (test infrastructure - never linked into or called by the product).
 *
 * Plain-C restatement of what the reference's PythonExecutionEngine computes for its benchmark query
 * (TPC-H Q1 variant, /root/reference/README.md:141-158, examples/benchmark.py:51-68), row at a time,
 * with the reference's quantisation points.  Two uses: (1) the checker for GPU results at sizes the
 * pure-Python oracle cannot finish in seconds, (2) the `cpu_baseline` ("kind": "port") leg of bench.py,
 * timed on the GPU box's host cores.  Pinned against the reference by tests/test_oracle_golden.py
 * (golden fixtures q1_multiblock / q1_selective / q1_ragged_blocks made by the real reference).
 *
 * Followed reference code:
 *   per block (= ScanJob, plan.py:90-93):
 *     FilterTask.execute            tasks.py:167-177   keep rows with l_shipdate <= cutoff
 *     AggregateTask.execute         tasks.py:284-289   evaluate the 11 expanded aggregate arguments
 *                                                      (AVG -> sum + count, sql.py:436-441) per row
 *     fill_aggregators              tasks.py:295-310   counter[key] = counter.get(key, 0) + x, row order
 *     BinaryOperatorColumn.execute_row sql.py:262-266  Python float (fp64) arithmetic on f32-decoded values
 *     WriteToShufflePartitions.write tasks.py:373 -> io.py:87-94  partials stored as f32 / i32
 *   final stage:
 *     AggregateTask.execute (after) tasks.py:290-292   fp64 sum of the f32 partials in block order
 *     ProjectTask AVG               plan.py:200-203, sql.py:443-446  sum / count on un-rounded sums
 *     WriteToLocalFileTask.write    tasks.py:400-410 -> io.py:94     result stored as f32 / i32
 *
 * Also holds the CPU twin of the counter-based synthetic lineitem generator (csrc/hs_ops.hip
 * k_gen_lineitem) so host and device produce identical rows from (seed, row index).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define Q1_MAX_GROUPS 256
#define Q1_NACC 11

typedef struct q1_row {
    int32_t key;            /* the 1-byte l_returnflag value */
    int32_t count_order;
    double sum_qty, sum_base_price, sum_disc_price, sum_charge, avg_qty, avg_price, avg_disc; /* f32 values */
    double raw[7];          /* the same 7 before the final f32 rounding (diagnostics) */
} q1_row;

static uint64_t splitmix(uint64_t x) {
    x += 0x9e3779b97f4a7c15ull;
    x = (x ^ (x >> 30)) * 0xbf58476d1ce4e5b9ull;
    x = (x ^ (x >> 27)) * 0x94d049bb133111ebull;
    return x ^ (x >> 31);
}
static uint64_t hs_rand(uint64_t seed, uint32_t col, int64_t i) {
    return splitmix(splitmix(seed + 0x632be59bd9b4e019ull * (col + 1)) + (uint64_t)i);
}

void q1_gen(uint64_t seed, int64_t row0, int64_t n, float* quantity, float* extendedprice, float* discount,
            float* tax, int64_t* shipdate, uint8_t* returnflag, int32_t* orderkey, uint8_t* shipmode_code) {
    for (int64_t k = 0; k < n; ++k) {
        const int64_t i = row0 + k;
        const uint32_t qty = 1 + (uint32_t)(hs_rand(seed, 0, i) % 50);
        if (quantity) quantity[k] = (float)qty;
        if (extendedprice) {
            const uint32_t cents = 90000 + (uint32_t)(hs_rand(seed, 1, i) % 110001);
            extendedprice[k] = (float)((double)qty * (double)cents / 100.0);
        }
        if (discount) discount[k] = (float)((double)(hs_rand(seed, 2, i) % 11) / 100.0);
        if (tax) tax[k] = (float)((double)(hs_rand(seed, 3, i) % 9) / 100.0);
        if (shipdate) {
            const int64_t days = (int64_t)(hs_rand(seed, 4, i) % 2526);
            shipdate[k] = (694310400ll + days * 86400ll) * 1000000ll;
        }
        if (returnflag) {
            const uint32_t r = (uint32_t)(hs_rand(seed, 5, i) % 4);
            returnflag[k] = r == 0 ? 'A' : (r == 3 ? 'R' : 'N');
        }
        if (orderkey) {
            const int64_t o = i / 4;
            orderkey[k] = (int32_t)(32 * (o / 8) + (o % 8) + 1);
        }
        if (shipmode_code) shipmode_code[k] = (uint8_t)(hs_rand(seed, 6, i) % 7);
    }
}

/* per-job dictionaries: one per aggregate in the reference (tasks.py:282-283); the key is one byte,
 * so a direct 256-entry table plays the role of Python's dict (presence tracked separately) */
typedef struct block_partial {
    uint8_t present[Q1_MAX_GROUPS];
    double f[Q1_MAX_GROUPS][7];  /* sum_qty, sum_base, sum_disc_price, sum_charge, avg_qty_sum, avg_price_sum, avg_disc_sum */
    int64_t c[Q1_MAX_GROUPS][4]; /* avg_qty_count, avg_price_count, avg_disc_count, count */
} block_partial;

static void q1_block(const float* q, const float* p, const float* d, const float* t, const int64_t* ship,
                     const uint8_t* flag, int64_t n, int64_t cutoff_us, block_partial* out) {
    memset(out, 0, sizeof(*out));
    for (int64_t i = 0; i < n; ++i) {
        if (!(ship[i] <= cutoff_us)) continue;            /* FilterTask */
        const uint8_t k = flag[i];
        const double qty = (double)q[i], price = (double)p[i], disc = (double)d[i], tx = (double)t[i];
        out->present[k] = 1;
        /* each aggregate evaluates its own argument tree per row, left-associated as parsed */
        out->f[k][0] = out->f[k][0] + qty;                               /* SUM(l_quantity) */
        out->f[k][1] = out->f[k][1] + price;                             /* SUM(l_extendedprice) */
        out->f[k][2] = out->f[k][2] + price * (1 - disc);                /* SUM(p * (1 - d)) */
        out->f[k][3] = out->f[k][3] + (price * (1 - disc)) * (1 + tx);   /* SUM(p * (1 - d) * (1 + t)) */
        out->f[k][4] = out->f[k][4] + qty;                               /* AVG(l_quantity)._sum */
        out->c[k][0] = out->c[k][0] + 1;                                 /* AVG(l_quantity)._count */
        out->f[k][5] = out->f[k][5] + price;
        out->c[k][1] = out->c[k][1] + 1;
        out->f[k][6] = out->f[k][6] + disc;
        out->c[k][2] = out->c[k][2] + 1;
        out->c[k][3] = out->c[k][3] + 1;                                 /* COUNT() = SUM(Lit(1)) */
    }
}

/* shuffle write + final stage over the per-block partials (see the header): returns the number of groups or -1 */
static int q1_merge(const block_partial* parts, int32_t nblocks, q1_row* out) {
    /* shuffle write: FLOAT -> f32, INTEGER -> i32; final stage: fp64 merge in block order */
    double fsum[Q1_MAX_GROUPS][7];
    int64_t csum[Q1_MAX_GROUPS][4];
    uint8_t present[Q1_MAX_GROUPS];
    memset(fsum, 0, sizeof(fsum));
    memset(csum, 0, sizeof(csum));
    memset(present, 0, sizeof(present));
    int overflow = 0;
    for (int b = 0; b < nblocks; ++b) {
        for (int k = 0; k < Q1_MAX_GROUPS; ++k) {
            if (!parts[b].present[k]) continue;
            present[k] = 1;
            for (int a = 0; a < 7; ++a) fsum[k][a] = fsum[k][a] + (double)(float)parts[b].f[k][a];
            for (int a = 0; a < 4; ++a) {
                if (parts[b].c[k][a] > 2147483647ll) overflow = 1;
                csum[k][a] = csum[k][a] + parts[b].c[k][a];
            }
        }
    }
    int n = 0;
    for (int k = 0; k < Q1_MAX_GROUPS; ++k) {
        if (!present[k]) continue;
        q1_row* r = &out[n++];
        r->key = k;
        if (csum[k][3] > 2147483647ll) overflow = 1;
        r->count_order = (int32_t)csum[k][3];
        r->raw[0] = fsum[k][0];
        r->raw[1] = fsum[k][1];
        r->raw[2] = fsum[k][2];
        r->raw[3] = fsum[k][3];
        r->raw[4] = fsum[k][4] / (double)csum[k][0]; /* AVG projection: sum / count in fp64 */
        r->raw[5] = fsum[k][5] / (double)csum[k][1];
        r->raw[6] = fsum[k][6] / (double)csum[k][2];
        r->sum_qty = (double)(float)r->raw[0];
        r->sum_base_price = (double)(float)r->raw[1];
        r->sum_disc_price = (double)(float)r->raw[2];
        r->sum_charge = (double)(float)r->raw[3];
        r->avg_qty = (double)(float)r->raw[4];
        r->avg_price = (double)(float)r->raw[5];
        r->avg_disc = (double)(float)r->raw[6];
    }
    return overflow ? -1 : n;
}

/* returns the number of groups written to out (ascending key byte), or -1 on an i32 overflow at a
 * quantisation point (the reference raises OverflowError, io.py:90) */
int q1_run_threads(const float* q, const float* p, const float* d, const float* t, const int64_t* ship,
                   const uint8_t* flag, const int64_t* block_rows, int32_t nblocks, int64_t cutoff_us,
                   int32_t nthreads, q1_row* out) {
    block_partial* parts = (block_partial*)malloc(sizeof(block_partial) * (size_t)(nblocks > 0 ? nblocks : 1));
    int64_t* starts = (int64_t*)malloc(sizeof(int64_t) * (size_t)(nblocks + 1));
    starts[0] = 0;
    for (int b = 0; b < nblocks; ++b) starts[b + 1] = starts[b] + block_rows[b];
    /* stage 0: one job per block; jobs are independent (the reference runs them one after another) */
#pragma omp parallel for schedule(dynamic, 1) num_threads(nthreads) if (nthreads > 1)
    for (int b = 0; b < nblocks; ++b) {
        const int64_t s = starts[b];
        q1_block(q + s, p + s, d + s, t + s, ship + s, flag + s, block_rows[b], cutoff_us, &parts[b]);
    }
    const int n = q1_merge(parts, nblocks, out);
    free(parts);
    free(starts);
    return n;
}

int q1_run(const float* q, const float* p, const float* d, const float* t, const int64_t* ship, const uint8_t* flag,
           const int64_t* block_rows, int32_t nblocks, int64_t cutoff_us, q1_row* out) {
    return q1_run_threads(q, p, d, t, ship, flag, block_rows, nblocks, cutoff_us, 1, out);
}

/* The same query over the SYNTHETIC table itself: block b = rows [b * rows_per_block, ...) of the counter-based
 * generator above, produced block by block into per-thread buffers (nothing of the 15.6 GB sf=100 table is ever
 * resident), one ScanJob per block, then the same merge.  bench.py checks the timed GPU result of the FULL table
 * against this; block ownership on N GPUs does not change it (the merge runs in block order). */
int q1_run_synth(uint64_t seed, int64_t total_rows, int64_t rows_per_block, int64_t cutoff_us, int32_t nthreads,
                 q1_row* out) {
    if (total_rows < 0 || rows_per_block < 1) return -2;
    const int64_t nb64 = (total_rows + rows_per_block - 1) / rows_per_block;
    if (nb64 > 1 << 24) return -2;
    const int32_t nblocks = (int32_t)nb64;
    block_partial* parts = (block_partial*)malloc(sizeof(block_partial) * (size_t)(nblocks > 0 ? nblocks : 1));
#pragma omp parallel num_threads(nthreads) if (nthreads > 1)
    {
        const size_t cap = (size_t)rows_per_block;
        float* q = (float*)malloc(cap * 4);
        float* p = (float*)malloc(cap * 4);
        float* d = (float*)malloc(cap * 4);
        float* t = (float*)malloc(cap * 4);
        int64_t* ship = (int64_t*)malloc(cap * 8);
        uint8_t* flag = (uint8_t*)malloc(cap);
#pragma omp for schedule(dynamic, 1)
        for (int b = 0; b < nblocks; ++b) {
            const int64_t row0 = (int64_t)b * rows_per_block;
            const int64_t n = total_rows - row0 < rows_per_block ? total_rows - row0 : rows_per_block;
            q1_gen(seed, row0, n, q, p, d, t, ship, flag, NULL, NULL);
            q1_block(q, p, d, t, ship, flag, n, cutoff_us, &parts[b]);
        }
        free(q); free(p); free(d); free(t); free(ship); free(flag);
    }
    const int n = q1_merge(parts, nblocks, out);
    free(parts);
    return n;
}
