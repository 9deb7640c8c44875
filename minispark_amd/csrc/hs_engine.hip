// hs_engine.hip - the stage-level C ABI: a GROUP BY query over one BlockFile table, end to end, without Python.
//
// What the reference does per query: the driver turns the physical plan into jobs, ships them to a worker process
// over stdin (src/mini_spark/execution.py:182-219, jobs.py:45-79) and the worker (zig-src/src/job.zig:3-57) reads
// BlockFile blocks (block_file.zig:225-306), runs the stage's task chain and writes shuffle / result files.  Here a
// host - the Python engine, or a cgo / JNI / FFI binding - hands over a PLAN BLOB (hs_stage_plan:
// the lowered programs of [scan -> WHERE -> partial aggregate] and [final merge -> projection -> result]) and gets the
// result columns back:
//
//   hs_engine_create -> hs_table_open (native BlockFile reader: header / footer / column spans, column pruning,
//   parallel pread into pinned staging, async H2D) -> hs_stage_prepare -> hs_stage_run -> hs_result_columns /
//   hs_result_write_blockfile.
//
// Everything that lived in minispark_amd/device.py for this path lives here too: slab and result-image layout, chunk
// geometry, workspace management, the capacity retry (a dictionary overflow grows the capacities and re-runs), the
// steady-state replay (the launches of a run are captured, hs_capture.h, and re-issued by later runs), and the
// zero-copy hand-over (result image in mapped pinned memory, "done" word polled by the host).  Queries that do not
// fit the on-chip tiers return HS_E_LIMIT: the caller takes the general operator sequence.
#include <fcntl.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "hs_device.h"

void hs_set_error(const char* fmt, ...);

namespace {

constexpr size_t kPad = 64;  // slack behind every device buffer (16-byte row-quad loads run past the last row)

struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;
    DevBuf() = default;
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    DevBuf(DevBuf&& o) noexcept : p(o.p), bytes(o.bytes) {
        o.p = nullptr;
        o.bytes = 0;
    }
    DevBuf& operator=(DevBuf&& o) noexcept {
        if (this != &o) {
            release();
            p = o.p;
            bytes = o.bytes;
            o.p = nullptr;
            o.bytes = 0;
        }
        return *this;
    }
    ~DevBuf() { release(); }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        bytes = 0;
    }
    bool alloc(size_t n, bool zero = false) {
        release();
        if (hipMalloc(&p, n + kPad) != hipSuccess) {
            p = nullptr;
            return false;
        }
        bytes = n;
        if (zero && hipMemset(p, 0, n + kPad) != hipSuccess) return false;
        return true;
    }
};

struct Column {
    int32_t type = 0;  // BlockFile type code: 0 INTEGER, 1 STRING, 2 FLOAT, 3 TIMESTAMP (constants.py:19-22)
    std::string name;
    bool loaded = false;
    hs_col col{};
    DevBuf data, lens, offs;
};

struct Span {
    int64_t off = 0, bytes = 0;
};

}  // namespace

// Pinned staging of the reader (hs_table_load): allocated once per engine - pinning host memory costs milliseconds per
// allocation, round 2 paid 2 x 16 MiB x 8 threads of it on every load.
struct PinnedPool {
    void* base = nullptr;
    size_t slot_bytes = 0;
    int n_slots = 0;
    ~PinnedPool() {
        if (base) (void)hipHostFree(base);
    }
    bool ensure(size_t bytes_per_slot, int slots) {
        if (base && slot_bytes >= bytes_per_slot && n_slots >= slots) return true;
        if (base) (void)hipHostFree(base);
        base = nullptr;
        if (hipHostMalloc(&base, bytes_per_slot * (size_t)slots, hipHostMallocMapped | hipHostMallocPortable) != hipSuccess) {
            base = nullptr;
            return false;
        }
        slot_bytes = bytes_per_slot;
        n_slots = slots;
        return true;
    }
    char* slot(int i) const { return (char*)base + (size_t)i * slot_bytes; }
};

struct ReaderPool;
void hs_reader_pool_free(ReaderPool* p);

struct hs_engine {
    int device = 0;
    DevBuf flags;  // one status word
    PinnedPool staging;
    ReaderPool* readers = nullptr;   // persistent reader threads (created by the first load)
    double last_load_seconds = 0.0;  // wall time of the last hs_table_load's read + copy pipeline
    int64_t last_load_bytes = 0;
    ~hs_engine() {
        if (readers) hs_reader_pool_free(readers);
    }
};

struct hs_table {
    hs_engine* engine = nullptr;
    std::string path;
    std::vector<Column> cols;
    std::vector<int64_t> block_rows;     // local blocks
    std::vector<int32_t> file_blocks;    // their global ids
    std::vector<std::vector<Span>> spans;  // [local block][column] byte span of the payload
    int32_t total_blocks = 0;
    int64_t nrows = 0;
    bool attached = false;  // columns are caller-owned device memory
};

namespace {

bool read_exact(int fd, void* dst, size_t n, int64_t off) {
    size_t done = 0;
    while (done < n) {
        const ssize_t got = pread(fd, (char*)dst + done, n - done, off + (int64_t)done);
        if (got <= 0) return false;
        done += (size_t)got;
    }
    return true;
}

int kind_of_type(int32_t type) { return type == 0 ? HS_I32 : type == 2 ? HS_F32 : type == 3 ? HS_I64 : HS_STR; }
int elem_bytes(int kind) { return kind == HS_I64 || kind == HS_F64 ? 8 : kind == HS_U8 ? 1 : 4; }

}  // namespace

extern "C" int hs_engine_create(int32_t device, hs_engine** out) {
    if (!out) {
        hs_set_error("hs_engine_create: null argument");
        return HS_E_ARG;
    }
    if (hipSetDevice(device) != hipSuccess) {
        hs_set_error("hs_engine_create: no such GPU (%d) - there is no CPU execution path", device);
        return HS_E_LAUNCH;
    }
    hs_engine* e = new hs_engine();
    e->device = device;
    if (!e->flags.alloc(16, true)) {
        delete e;
        hs_set_error("hs_engine_create: out of device memory");
        return HS_E_LAUNCH;
    }
    *out = e;
    return HS_OK;
}

extern "C" void hs_engine_destroy(hs_engine* e) { delete e; }

extern "C" int hs_engine_load_stats(const hs_engine* e, double* seconds, int64_t* bytes) {
    if (!e) {
        hs_set_error("hs_engine_load_stats: null engine");
        return HS_E_ARG;
    }
    if (seconds) *seconds = e->last_load_seconds;
    if (bytes) *bytes = e->last_load_bytes;
    return HS_OK;
}

// ---- BlockFile reader (format: SURVEY.md appendix A; reference io.py:47-170, zig block_file.zig:225-306) --------------
extern "C" int hs_table_open(hs_engine* e, const char* path, int32_t rank, int32_t world, hs_table** out) {
    if (!e || !path || !out || world < 1 || rank < 0 || rank >= world) {
        hs_set_error("hs_table_open: bad arguments");
        return HS_E_ARG;
    }
    const int fd = open(path, O_RDONLY);
    if (fd < 0) {
        hs_set_error("hs_table_open: cannot open %s", path);
        return HS_E_ARG;
    }
    struct stat st;
    fstat(fd, &st);
    const int64_t size = st.st_size;
    auto fail = [&](const char* what) {
        close(fd);
        hs_set_error("hs_table_open: %s: %s", path, what);
        return HS_E_ARG;
    };
    uint8_t ncols = 0;
    if (size < 5 || !read_exact(fd, &ncols, 1, 0)) return fail("not a BlockFile (too short)");
    hs_table* t = new hs_table();
    t->engine = e;
    t->path = path;
    int64_t pos = 1;
    for (int c = 0; c < ncols; ++c) {
        uint8_t hdr[2];
        if (!read_exact(fd, hdr, 2, pos)) {
            delete t;
            return fail("truncated header");
        }
        std::string name(hdr[1], '\0');
        if (hdr[1] && !read_exact(fd, &name[0], hdr[1], pos + 2)) {
            delete t;
            return fail("truncated header");
        }
        pos += 2 + hdr[1];
        Column col;
        col.type = hdr[0];
        col.name = name;
        if (hdr[0] > 3) {
            delete t;
            return fail("unknown column type");
        }
        t->cols.push_back(std::move(col));
    }
    uint32_t nblocks = 0;
    if (!read_exact(fd, &nblocks, 4, size - 4) || (int64_t)nblocks * 8 + 4 + pos > size) {
        delete t;
        return fail("bad footer");
    }
    std::vector<uint64_t> starts(nblocks);
    if (nblocks && !read_exact(fd, starts.data(), (size_t)nblocks * 8, size - 4 - (int64_t)nblocks * 8)) {
        delete t;
        return fail("bad footer");
    }
    t->total_blocks = (int32_t)nblocks;
    for (uint32_t b = 0; b < nblocks; ++b) {
        if ((int32_t)(b % (uint32_t)world) != rank) continue;  // block b lives on rank b % world (plan.py:90-93: independent jobs)
        uint32_t rows = 0;
        int64_t p = (int64_t)starts[b];
        if (!read_exact(fd, &rows, 4, p)) {
            delete t;
            return fail("truncated block");
        }
        p += 4;
        std::vector<Span> spans(ncols);
        for (int c = 0; c < ncols; ++c) {
            uint64_t bytes = 0;
            if (!read_exact(fd, &bytes, 8, p)) {
                delete t;
                return fail("truncated block");
            }
            spans[c] = Span{p + 8, (int64_t)bytes};
            p += 8 + (int64_t)bytes;
            const int kind = kind_of_type(t->cols[c].type);
            if ((kind != HS_STR && (int64_t)bytes != (int64_t)rows * elem_bytes(kind)) || (kind == HS_STR && (int64_t)bytes < rows) ||
                p > size) {
                delete t;
                return fail("column payload size does not match the block's row count");
            }
        }
        t->block_rows.push_back(rows);
        t->file_blocks.push_back((int32_t)b);
        t->spans.push_back(std::move(spans));
        t->nrows += rows;
    }
    close(fd);
    *out = t;
    return HS_OK;
}

extern "C" void hs_table_close(hs_table* t) { delete t; }

extern "C" int hs_table_info(const hs_table* t, int32_t* n_cols, int64_t* n_rows, int32_t* n_blocks, int32_t* total_blocks) {
    if (!t) {
        hs_set_error("hs_table_info: null table");
        return HS_E_ARG;
    }
    if (n_cols) *n_cols = (int32_t)t->cols.size();
    if (n_rows) *n_rows = t->nrows;
    if (n_blocks) *n_blocks = (int32_t)t->block_rows.size();
    if (total_blocks) *total_blocks = t->total_blocks;
    return HS_OK;
}

extern "C" int hs_table_schema(const hs_table* t, int32_t col, int32_t* type, char* name, int32_t name_cap) {
    if (!t || col < 0 || col >= (int32_t)t->cols.size()) {
        hs_set_error("hs_table_schema: no such column");
        return HS_E_ARG;
    }
    if (type) *type = t->cols[col].type;
    if (name && name_cap > 0) {
        strncpy(name, t->cols[col].name.c_str(), (size_t)name_cap - 1);
        name[name_cap - 1] = 0;
    }
    return HS_OK;
}

extern "C" int hs_table_column(const hs_table* t, int32_t col, hs_col* out, int64_t* n_rows) {
    if (!t || !out || col < 0 || col >= (int32_t)t->cols.size() || !t->cols[col].loaded) {
        hs_set_error("hs_table_column: column not loaded");
        return HS_E_ARG;
    }
    *out = t->cols[col].col;
    if (n_rows) *n_rows = t->nrows;
    return HS_OK;
}

// Caller-owned device columns as a table (synthetic data, or columns another reader placed in HBM).
extern "C" int hs_table_attach(hs_engine* e, int32_t n_cols, const hs_col* cols, const int32_t* types, const int64_t* block_rows,
                               int32_t n_blocks, hs_table** out) {
    if (!e || !cols || !types || !block_rows || !out || n_cols < 1 || n_blocks < 0) {
        hs_set_error("hs_table_attach: bad arguments");
        return HS_E_ARG;
    }
    hs_table* t = new hs_table();
    t->engine = e;
    t->attached = true;
    for (int c = 0; c < n_cols; ++c) {
        Column col;
        col.type = types[c];
        col.name = "c" + std::to_string(c);
        col.col = cols[c];
        col.loaded = cols[c].data != nullptr;
        t->cols.push_back(std::move(col));
    }
    for (int b = 0; b < n_blocks; ++b) {
        t->block_rows.push_back(block_rows[b]);
        t->file_blocks.push_back(b);
        t->nrows += block_rows[b];
    }
    t->total_blocks = n_blocks;
    *out = t;
    return HS_OK;
}

namespace {

struct Piece {
    int64_t file_off, bytes;
    char* dst;  // device address
};

// Reader threads (round 3): the column spans are cut into chunks of HS_READ_CHUNK bytes; every thread owns HS_READ_SLOTS
// pinned slots of the engine's pool and a stream, claims the next chunk, preads it into a free slot (page cache ->
// pinned memory: the one host-side copy) and queues its H2D copy; a slot is reused once its copy's event has fired.
// With chunks of a few MiB, up to 16 threads and 4 slots each, the preads of all threads and the DMA of earlier
// chunks overlap from the first millisecond on (round 2: 8 threads x 2 slots of whole 8-16 MiB spans, pinned memory
// allocated per call: 27 GB/s of the link's 63).
constexpr int64_t HS_READ_CHUNK = 4ll << 20;
constexpr int HS_READ_SLOTS = 4;
constexpr int HS_READ_THREADS_MAX = 16;

int reader_threads() {
    // 8 by default: measured on the test box (16 host threads granted) 6-10 readers all reach 42-45 GB/s at sf=10,
    // 16 fall back to 37-41 (they contend for the page cache / memory bandwidth, not for the link)
    int n = 8;
    if (const char* env = getenv("HIPSPARK_INGEST_READERS")) n = atoi(env);
    const unsigned hw = std::thread::hardware_concurrency();
    if (hw && n > (int)hw) n = (int)hw;
    return n < 1 ? 1 : (n > HS_READ_THREADS_MAX ? HS_READ_THREADS_MAX : n);
}

// Host-to-device copy of one staged chunk by a KERNEL that reads the pinned slot over PCIe (the slot is mapped into
// the device's address space): 16-byte loads from host memory, 16-byte stores to HBM.  HIPSPARK_INGEST_COPY=sdma uses
// hipMemcpyAsync (the DMA engines) instead; the two measure alike (42-45 GB/s at sf=10, ~50 GB/s at sf=30).
__global__ void __launch_bounds__(256) k_pull_chunk(const uint4* __restrict__ src, uint4* __restrict__ dst, int64_t n16,
                                                    const uint8_t* __restrict__ src_tail, uint8_t* __restrict__ dst_tail, int tail) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (int64_t)gridDim.x * blockDim.x)
        dst[i] = src[i];
    if (blockIdx.x == 0 && (int)threadIdx.x < tail) dst_tail[threadIdx.x] = src_tail[threadIdx.x];
}

bool copy_by_kernel() {
    static const bool k = !(getenv("HIPSPARK_INGEST_COPY") && getenv("HIPSPARK_INGEST_COPY")[0] == 's');
    return k;
}

}  // namespace

// The engine's reader threads live as long as the engine: a thread's first HIP call, its stream and its events cost
// milliseconds - at sf=10 (1.56 GB, ~31 ms at the link's practical rate) that was 5 ms of every load.
struct ReaderPool {
    struct Job {
        const std::string* path = nullptr;
        const std::vector<Piece>* pieces = nullptr;
        char* pool_dev = nullptr;
        bool by_kernel = false;
        std::atomic<size_t> next{0};
        std::atomic<bool> failed{false};
    };
    hs_engine* engine = nullptr;
    std::vector<std::thread> threads;
    std::mutex m;
    std::condition_variable cv_job, cv_done;
    Job* job = nullptr;
    uint64_t job_id = 0;
    int working = 0;
    bool stop = false;

    ~ReaderPool() {
        {
            std::lock_guard<std::mutex> lock(m);
            stop = true;
        }
        cv_job.notify_all();
        for (std::thread& th : threads) th.join();
    }

    // `seen0`: the id of the last job posted before this thread existed - a thread added to a pool that has already run
    // jobs must wait for the NEXT post, not wake on the stale id with `job` reset to null
    void worker(int w, uint64_t seen0) {
        const bool dev_ok = hipSetDevice(engine->device) == hipSuccess;
        hipStream_t stream = nullptr;
        hipEvent_t ev[HS_READ_SLOTS] = {};
        bool ok_setup = dev_ok && hipStreamCreateWithFlags(&stream, hipStreamNonBlocking) == hipSuccess;
        for (int k = 0; ok_setup && k < HS_READ_SLOTS; ++k) ok_setup = hipEventCreateWithFlags(&ev[k], hipEventDisableTiming) == hipSuccess;
        uint64_t seen = seen0;
        for (;;) {
            Job* j = nullptr;
            {
                std::unique_lock<std::mutex> lock(m);
                cv_job.wait(lock, [&] { return stop || job_id != seen; });
                if (stop) break;
                seen = job_id;
                j = job;
            }
            if (!j) continue;  // (never counted in `working`: nothing to report)
            bool ok = ok_setup;
            const int fd = ok ? open(j->path->c_str(), O_RDONLY) : -1;
            ok = ok && fd >= 0;
            bool used[HS_READ_SLOTS] = {};
            const std::vector<Piece>& pieces = *j->pieces;
            for (int turn = 0; ok && !j->failed; turn = (turn + 1) % HS_READ_SLOTS) {
                const size_t i = j->next.fetch_add(1);
                if (i >= pieces.size()) break;
                const Piece& p = pieces[i];
                char* slot = engine->staging.slot(w * HS_READ_SLOTS + turn);
                if (used[turn]) ok = hipEventSynchronize(ev[turn]) == hipSuccess;  // the slot's previous copy has left it
                ok = ok && read_exact(fd, slot, (size_t)p.bytes, p.file_off);
                if (ok && j->by_kernel && (((uintptr_t)p.dst) & 15) == 0) {
                    // (a destination inside a column buffer is 16-byte aligned whenever the span starts on a multiple of
                    // 16 bytes of the column: always for the fixed-width columns of 2 Mi-row blocks)
                    char* mapped = j->pool_dev + (slot - (char*)engine->staging.base);
                    const int64_t n16 = p.bytes / 16;
                    const int tail = (int)(p.bytes - n16 * 16);
                    hipLaunchKernelGGL(k_pull_chunk, dim3(64), dim3(256), 0, stream, (const uint4*)mapped, (uint4*)p.dst, n16,
                                       (const uint8_t*)mapped + n16 * 16, (uint8_t*)p.dst + n16 * 16, tail);
                    ok = hipGetLastError() == hipSuccess;
                } else {
                    ok = ok && hipMemcpyAsync(p.dst, slot, (size_t)p.bytes, hipMemcpyHostToDevice, stream) == hipSuccess;
                }
                ok = ok && hipEventRecord(ev[turn], stream) == hipSuccess;
                used[turn] = true;
            }
            if (stream) ok = (hipStreamSynchronize(stream) == hipSuccess) && ok;
            if (fd >= 0) close(fd);
            if (!ok) j->failed = true;
            {
                std::lock_guard<std::mutex> lock(m);
                --working;
            }
            cv_done.notify_all();
        }
        for (int k = 0; k < HS_READ_SLOTS; ++k)
            if (ev[k]) (void)hipEventDestroy(ev[k]);
        if (stream) (void)hipStreamDestroy(stream);
    }

    bool run(hs_engine* e, Job& j, int n_threads) {
        engine = e;
        while ((int)threads.size() < n_threads) {
            const int w = (int)threads.size();
            const uint64_t posted = job_id;  // run() is the only writer and is not re-entered
            threads.emplace_back([this, w, posted] { worker(w, posted); });
        }
        {
            std::lock_guard<std::mutex> lock(m);
            job = &j;
            ++job_id;
            working = (int)threads.size();
        }
        cv_job.notify_all();
        std::unique_lock<std::mutex> lock(m);
        cv_done.wait(lock, [&] { return working == 0; });
        job = nullptr;
        return !j.failed;
    }
};

void hs_reader_pool_free(ReaderPool* p) { delete p; }

namespace {

bool run_pieces(hs_engine* e, const std::string& path, const std::vector<Piece>& spans, std::string& err) {
    if (spans.empty()) return true;
    std::vector<Piece> pieces;
    int64_t total = 0;
    for (const Piece& p : spans) {
        for (int64_t at = 0; at < p.bytes; at += HS_READ_CHUNK) {
            const int64_t n = p.bytes - at < HS_READ_CHUNK ? p.bytes - at : HS_READ_CHUNK;
            pieces.push_back(Piece{p.file_off + at, n, p.dst + at});
        }
        total += p.bytes;
    }
    if (!e->staging.ensure((size_t)HS_READ_CHUNK, HS_READ_THREADS_MAX * HS_READ_SLOTS)) {
        err = "cannot pin host staging memory";
        return false;
    }
    if (!e->readers) e->readers = new ReaderPool();
    ReaderPool::Job job;
    job.path = &path;
    job.pieces = &pieces;
    job.by_kernel = copy_by_kernel() && hipHostGetDevicePointer((void**)&job.pool_dev, e->staging.base, 0) == hipSuccess && job.pool_dev;
    const auto t0 = std::chrono::steady_clock::now();
    const bool ok = e->readers->run(e, job, reader_threads());
    e->last_load_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    e->last_load_bytes = total;
    if (!ok) err = "read or host-to-device copy failed";
    return ok;
}

}  // namespace

// The reader's pipeline for a caller that owns the destination buffers (the Python engine's table loader): spans of the
// file -> device addresses.  Same threads, pinned pool and chunking as hs_table_load.
extern "C" int hs_read_spans(hs_engine* e, const char* path, const hs_span* spans, int32_t n_spans) {
    if (!e || !path || n_spans < 0 || (n_spans > 0 && !spans)) {
        hs_set_error("hs_read_spans: bad arguments");
        return HS_E_ARG;
    }
    if (hipSetDevice(e->device) != hipSuccess) return HS_E_LAUNCH;
    std::vector<Piece> pieces;
    for (int i = 0; i < n_spans; ++i) {
        if (spans[i].bytes < 0 || spans[i].file_offset < 0 || (spans[i].bytes > 0 && !spans[i].dst)) {
            hs_set_error("hs_read_spans: span %d is malformed", i);
            return HS_E_ARG;
        }
        if (spans[i].bytes) pieces.push_back(Piece{spans[i].file_offset, spans[i].bytes, (char*)spans[i].dst});
    }
    std::string err;
    if (!run_pieces(e, path, pieces, err)) {
        hs_set_error("hs_read_spans: %s: %s", path, err.c_str());
        return HS_E_LAUNCH;
    }
    return HS_OK;
}

// Load the byte spans of the listed columns (column pruning: nothing else is read) into one contiguous device buffer
// per column across all local blocks; STRING columns get their offsets from a device prefix sum and fixed_len when every
// row has the same length.  Idempotent per column.
extern "C" int hs_table_load(hs_engine* e, hs_table* t, const int32_t* col_ids, int32_t n) {
    if (!e || !t || !col_ids || n < 0) {
        hs_set_error("hs_table_load: bad arguments");
        return HS_E_ARG;
    }
    if (hipSetDevice(e->device) != hipSuccess) return HS_E_LAUNCH;
    std::vector<Piece> pieces;
    std::vector<int> fresh;
    for (int k = 0; k < n; ++k) {
        const int c = col_ids[k];
        if (c < 0 || c >= (int)t->cols.size()) {
            hs_set_error("hs_table_load: no such column %d", c);
            return HS_E_ARG;
        }
        Column& col = t->cols[c];
        if (col.loaded) continue;
        if (t->attached) {
            hs_set_error("hs_table_load: column %d of an attached table has no device data", c);
            return HS_E_ARG;
        }
        const int kind = kind_of_type(col.type);
        const size_t nb = t->block_rows.size();
        bool ok = true;
        if (kind == HS_STR) {
            int64_t payload = 0;
            for (size_t b = 0; b < nb; ++b) payload += t->spans[b][c].bytes - t->block_rows[b];
            ok = col.lens.alloc((size_t)t->nrows) && col.data.alloc((size_t)payload);
            int64_t row = 0, byte = 0;
            for (size_t b = 0; ok && b < nb; ++b) {
                const Span& s = t->spans[b][c];
                const int64_t rows = t->block_rows[b];
                if (rows) pieces.push_back(Piece{s.off, rows, (char*)col.lens.p + row});
                if (s.bytes > rows) pieces.push_back(Piece{s.off + rows, s.bytes - rows, (char*)col.data.p + byte});
                row += rows;
                byte += s.bytes - rows;
            }
        } else {
            ok = col.data.alloc((size_t)t->nrows * (size_t)elem_bytes(kind));
            int64_t byte = 0;
            for (size_t b = 0; ok && b < nb; ++b) {
                const Span& s = t->spans[b][c];
                if (s.bytes) pieces.push_back(Piece{s.off, s.bytes, (char*)col.data.p + byte});
                byte += s.bytes;
            }
        }
        if (!ok) {
            hs_set_error("hs_table_load: out of device memory");
            return HS_E_LAUNCH;
        }
        fresh.push_back(c);
    }
    std::string err;
    if (!run_pieces(e, t->path, pieces, err)) {
        hs_set_error("hs_table_load: %s: %s", t->path.c_str(), err.c_str());
        return HS_E_LAUNCH;
    }
    for (int c : fresh) {
        Column& col = t->cols[c];
        const int kind = kind_of_type(col.type);
        col.col = hs_col{kind, -1, col.data.p, nullptr, nullptr};
        if (kind == HS_STR) {
            col.col.lens = (const uint8_t*)col.lens.p;
            int32_t minmax[2] = {0, 0};
            if (t->nrows > 0) {
                DevBuf ws, mm;
                if (!col.offs.alloc((size_t)(t->nrows + 1) * 8) || !ws.alloc(hs_scan_ws_bytes(t->nrows)) || !mm.alloc(8)) {
                    hs_set_error("hs_table_load: out of device memory");
                    return HS_E_LAUNCH;
                }
                const int rc = hs_str_offsets(nullptr, (const uint8_t*)col.lens.p, t->nrows, (int64_t*)col.offs.p, (int32_t*)mm.p, ws.p);
                if (rc) return rc;
                if (hipMemcpy(minmax, mm.p, 8, hipMemcpyDeviceToHost) != hipSuccess) return HS_E_LAUNCH;
            }
            if (t->nrows == 0 || minmax[0] == minmax[1]) {  // every row has the same length: no offsets needed
                col.col.fixed_len = t->nrows == 0 ? 0 : minmax[0];
                col.offs.release();
            } else {
                col.col.offs = (const int64_t*)col.offs.p;
            }
        }
        col.loaded = true;
    }
    return HS_OK;
}

// =====================================================================================================
// Stages
// =====================================================================================================
struct hs_stage {
    hs_engine* engine = nullptr;
    hs_table* table = nullptr;
    hs_stage_plan plan{};
    int32_t group_cap = 4, merge_cap = 16;
    int32_t world = 1, n_order = 1;
    // prepared state (rebuilt when a capacity grows)
    bool ready = false;
    hs_col cols[HS_MAX_COLS];
    hs_agg_geom geom{};
    hs_slab_desc desc{};
    hs_finish_spec fin{};
    int32_t key_bytes = 0;
    int64_t n_units = 0, slab_bytes = 0, image_bytes = 0;
    DevBuf chunks, chunk0, unit_ids, slab, ws, scratch;
    // round 3: tens to thousands of groups per block - the shared-dictionary tier + the general operator sequence after it
    // (pack -> key gather -> merge -> key gather / projection -> rounding), all behind hs_stage_run
    int32_t tier = 0;  // 0: per-lane tables + the one-launch finish; 1: shared dictionary + general tail
    DevBuf key_col, key_wide;  // a computed GROUP BY key (plan version 2): the 4-byte column the scan reads + its i64 evaluation
    hs_col kcols[HS_MAX_COLS];
    DevBuf sh_rep, sh_acc, sh_ngroups, sh_pack_start, sh_dense_rep, sh_order, sh_key, sh_accs, sh_mrep, sh_macc, sh_mgroups, sh_mkey,
        sh_prog_out, sh_image;
    int64_t sh_slots = 0;
    void* image_host = nullptr;  // pinned, mapped
    void* image_dev = nullptr;
    void* capture = nullptr;     // steady state: the launches of one run
    int64_t runs = 0, replays = 0, grows = 0;
    // last result
    uint32_t last_flags = 0;
    int64_t last_rows = 0;
    ~hs_stage() {
        if (capture) hs_capture_free(capture);
        if (image_host) (void)hipHostFree(image_host);
    }
};

namespace {

// program slots -> the table's columns; a computed key (plan version 2) gets its own 4-byte column next to them
int bind_columns(hs_stage* s) {
    hs_table* t = s->table;
    const hs_stage_plan& P = s->plan;
    for (int i = 0; i < P.n_cols; ++i)
        if (i != P.key_slot || !P.key_computed) s->cols[i] = t->cols[P.col_ids[i]].col;
    if (!P.key_computed) return HS_OK;
    for (int i = 0; i < P.n_kcols; ++i) s->kcols[i] = t->cols[P.kcol_ids[i]].col;
    const size_t rows = (size_t)(t->nrows > 0 ? t->nrows : 1);
    if (!s->key_col.p && !(s->key_col.alloc(rows * 4 + kPad, true) && s->key_wide.alloc(rows * 8 + kPad))) {
        hs_set_error("hs_stage: out of device memory for the computed key column");
        return HS_E_LAUNCH;
    }
    s->cols[P.key_slot] = hs_col{HS_I32, -1, s->key_col.p, nullptr, nullptr};
    return HS_OK;
}

// ProjectTask in front of the aggregate (reference tasks.py:32-35), for the key column alone: one evaluation per run
int compute_key(hs_stage* s, void* stream) {
    const hs_stage_plan& P = s->plan;
    if (!P.key_computed || s->table->nrows == 0) return HS_OK;
    void* outs[1] = {s->key_wide.p};
    const int32_t kinds[1] = {HS_I64};
    uint32_t* flags = (uint32_t*)s->engine->flags.p;
    int rc = hs_eval(stream, s->kcols, P.n_kcols, &P.key_prog, nullptr, s->table->nrows, nullptr, outs, kinds, 1, flags);
    if (!rc) rc = hs_quantise(stream, s->key_wide.p, HS_I64, s->table->nrows, nullptr, s->key_col.p, flags);
    return rc;
}

int stage_prepare(hs_stage* s) {
    hs_table* t = s->table;
    const hs_stage_plan& P = s->plan;
    if (s->capture) {
        hs_capture_free(s->capture);
        s->capture = nullptr;
    }
    s->ready = false;
    if (const int rc0 = bind_columns(s)) return rc0;
    const hs_col& kc = s->cols[P.key_slot];
    // the exchange slab holds keys in their stored kinds, packed into the 64-bit key word by the finish launch
    if (kc.kind == HS_STR) {
        if (kc.fixed_len != 1 && kc.fixed_len != 2 && kc.fixed_len != 4) {
            hs_set_error("hs_stage: a string GROUP BY key needs a fixed length of 1, 2 or 4 bytes on this path");
            return HS_E_LIMIT;
        }
        s->key_bytes = kc.fixed_len;
    } else if (kc.kind == HS_I32 || kc.kind == HS_F32 || kc.kind == HS_I64) {
        s->key_bytes = elem_bytes(kc.kind);
    } else {
        hs_set_error("hs_stage: GROUP BY key is not a stored column kind");
        return HS_E_LIMIT;
    }
    // units = the table's local blocks (reference: one ScanJob per block, plan.py:90-93)
    s->n_units = (int64_t)t->block_rows.size();
    std::vector<int64_t> unit_rows(s->n_units + 1, 0);
    for (int64_t u = 0; u < s->n_units; ++u) unit_rows[u + 1] = unit_rows[u] + t->block_rows[u];
    int rc = hs_agg_partial_geom(unit_rows.data(), s->n_units, P.spec.n_acc, s->group_cap, &s->geom);
    if (rc) return rc;  // HS_E_LIMIT: the private-table tier does not hold this query
    std::vector<hs_chunk> chunks((size_t)(s->geom.n_chunks > 0 ? s->geom.n_chunks : 1));
    std::vector<int64_t> chunk0((size_t)s->n_units + 1, 0);
    rc = hs_agg_partial_chunks(unit_rows.data(), s->n_units, &s->geom, chunks.data(), chunk0.data());
    if (rc) return rc;
    std::vector<int64_t> ids(t->file_blocks.begin(), t->file_blocks.end());
    bool ok = s->chunks.alloc(chunks.size() * sizeof(hs_chunk)) && s->chunk0.alloc(chunk0.size() * 8) &&
              s->unit_ids.alloc((ids.size() ? ids.size() : 1) * 8) && s->ws.alloc(s->geom.ws_bytes, true);
    ok = ok && hipMemcpy(s->chunks.p, chunks.data(), chunks.size() * sizeof(hs_chunk), hipMemcpyHostToDevice) == hipSuccess &&
         hipMemcpy(s->chunk0.p, chunk0.data(), chunk0.size() * 8, hipMemcpyHostToDevice) == hipSuccess &&
         (ids.empty() || hipMemcpy(s->unit_ids.p, ids.data(), ids.size() * 8, hipMemcpyHostToDevice) == hipSuccess);
    // slab: [flags u32][pad][row count i64] | order key i64 x M | key column | accumulator columns (4 bytes per row)
    const int64_t max_local = (t->total_blocks + s->world - 1) / s->world;
    const int64_t M = (s->n_units > max_local ? s->n_units : max_local) * s->group_cap;
    hs_slab_desc& d = s->desc;
    memset(&d, 0, sizeof(d));
    d.slab_rows = M > 0 ? M : s->group_cap;
    int64_t pos = 16;
    d.order_off = pos;
    pos += 8 * d.slab_rows;
    pos = (pos + 15) & ~(int64_t)15;
    d.key_off = pos;
    pos += (int64_t)s->key_bytes * d.slab_rows;
    d.n_acc = P.spec.n_acc;
    for (int a = 0; a < P.spec.n_acc; ++a) {
        pos = (pos + 15) & ~(int64_t)15;
        d.acc_off[a] = pos;
        d.acc_kind[a] = P.spec.is_int[a] ? HS_I32 : HS_F32;
        pos += 4 * d.slab_rows;
    }
    s->slab_bytes = (pos + 15) & ~(int64_t)15;
    d.stride = s->slab_bytes;
    d.key_kind = kc.kind;
    d.key_len = kc.kind == HS_STR ? kc.fixed_len : 0;
    std::vector<uint8_t> slab_image((size_t)s->slab_bytes, 0);
    for (int64_t r = 0; r < d.slab_rows; ++r) ((int64_t*)(slab_image.data() + d.order_off))[r] = -1;
    ok = ok && s->slab.alloc((size_t)s->slab_bytes) &&
         hipMemcpy(s->slab.p, slab_image.data(), slab_image.size(), hipMemcpyHostToDevice) == hipSuccess;
    // result image: header 16 bytes, then every column at a 16-byte aligned offset, cap elements each
    s->fin = P.fin;
    pos = 16;
    for (int o = 0; o < s->fin.n_out; ++o) {
        hs_finish_out& out = s->fin.outs[o];
        const int width = out.src == 0 ? s->key_bytes : (out.kind == HS_I64 ? 8 : 4);
        out.offset = pos;
        pos = (pos + (int64_t)s->merge_cap * width + 15) & ~(int64_t)15;
    }
    s->image_bytes = pos;
    if (s->image_host) (void)hipHostFree(s->image_host);
    s->image_host = s->image_dev = nullptr;
    ok = ok && hipHostMalloc(&s->image_host, (size_t)s->image_bytes + kPad, hipHostMallocMapped) == hipSuccess &&
         hipHostGetDevicePointer(&s->image_dev, s->image_host, 0) == hipSuccess && s->image_dev;
    if (ok) memset(s->image_host, 0, (size_t)s->image_bytes + kPad);
    ok = ok && s->scratch.alloc(hs_agg_finish_scratch_bytes(s->merge_cap, s->fin.n_fold), true);
    if (!ok) {
        hs_set_error("hs_stage: out of device / pinned memory");
        return HS_E_LAUNCH;
    }
    s->n_order = t->total_blocks > 0 ? t->total_blocks : 1;
    s->ready = true;
    return HS_OK;
}

uint32_t* engine_flags(hs_stage* s) { return (uint32_t*)s->engine->flags.p; }

int launch_partial(hs_stage* s, void* stream) {
    const hs_stage_plan& P = s->plan;
    if (s->n_units == 0) return HS_OK;
    if (const int rc = compute_key(s, stream)) return rc;
    return hs_agg_partial_slab(stream, s->cols, P.n_cols, P.key_slot, &P.prog, &P.spec, (const hs_chunk*)s->chunks.p,
                               (const int64_t*)s->chunk0.p, s->n_units, &s->geom, (const int64_t*)s->unit_ids.p,
                               (uint8_t*)s->slab.p, &s->desc, s->ws.p, engine_flags(s), nullptr, nullptr);
}

int launch_finish(hs_stage* s, void* stream, const void* gathered, int32_t world) {
    const hs_stage_plan& P = s->plan;
    return hs_agg_finish(stream, (const uint8_t*)(gathered ? gathered : s->slab.p), world, &s->desc, &s->fin,
                         P.fin_prog.n_ins ? &P.fin_prog : nullptr, s->n_order, s->merge_cap, (uint8_t*)s->image_dev,
                         s->scratch.p, engine_flags(s), (uint32_t*)s->slab.p);
}

int wait_result(hs_stage* s, void* stream) {
    volatile uint32_t* done = (volatile uint32_t*)s->image_host + 1;
    for (int64_t spins = 0; *done == 0; ++spins) {
        if (spins > 2000000) {  // a long scan: let the runtime wait instead of this core
            if (hipStreamSynchronize((hipStream_t)stream) != hipSuccess || *done == 0) {
                hs_set_error("hs_stage_run: the finish launch did not hand its result over");
                return HS_E_LAUNCH;
            }
        }
    }
    std::atomic_thread_fence(std::memory_order_acquire);
    s->last_flags = *(volatile uint32_t*)s->image_host;
    const int64_t n = *(volatile int64_t*)((char*)s->image_host + 8);
    s->last_rows = n < s->merge_cap ? n : s->merge_cap;
    *done = 0;  // ready for the next launch into this image
    return HS_OK;
}

}  // namespace

namespace {

// ---- the shared-dictionary tier behind hs_stage_run ------------------------------------------------------------------------
int shared_prepare(hs_stage* s) {
    hs_table* t = s->table;
    const hs_stage_plan& P = s->plan;
    s->ready = false;
    if (s->world != 1) {
        hs_set_error("hs_stage: more than 16 groups per block on several ranks belongs to the per-operator ABI");
        return HS_E_LIMIT;
    }
    if (const int rc0 = bind_columns(s)) return rc0;
    const hs_col& kc = s->cols[P.key_slot];
    if (kc.kind == HS_STR) {
        if (kc.fixed_len != 1 && kc.fixed_len != 2 && kc.fixed_len != 4) {
            hs_set_error("hs_stage: a string GROUP BY key needs a fixed length of 1, 2 or 4 bytes on this path");
            return HS_E_LIMIT;
        }
        s->key_bytes = kc.fixed_len;
    } else if (kc.kind == HS_I32 || kc.kind == HS_F32 || kc.kind == HS_I64) {
        s->key_bytes = elem_bytes(kc.kind);
    } else {
        hs_set_error("hs_stage: GROUP BY key is not a stored column kind");
        return HS_E_LIMIT;
    }
    s->n_units = (int64_t)t->block_rows.size();
    std::vector<int64_t> unit_rows(s->n_units + 1, 0);
    for (int64_t u = 0; u < s->n_units; ++u) unit_rows[u + 1] = unit_rows[u] + t->block_rows[u];
    int rc = hs_agg_shared_geom(unit_rows.data(), s->n_units, P.spec.n_acc, s->group_cap, &s->geom);
    if (rc) return rc;
    std::vector<hs_chunk> chunks((size_t)(s->geom.n_chunks > 0 ? s->geom.n_chunks : 1));
    std::vector<int64_t> chunk0((size_t)s->n_units + 1, 0);
    rc = hs_agg_partial_chunks(unit_rows.data(), s->n_units, &s->geom, chunks.data(), chunk0.data());
    if (rc) return rc;
    const int unit_cap = s->geom.pad, n_acc = P.spec.n_acc, nf = s->plan.fin.n_fold;
    s->sh_slots = s->n_units * (int64_t)unit_cap;
    const int64_t slots = s->sh_slots > 0 ? s->sh_slots : 1;
    bool ok = s->chunks.alloc(chunks.size() * sizeof(hs_chunk)) &&
              hipMemcpy(s->chunks.p, chunks.data(), chunks.size() * sizeof(hs_chunk), hipMemcpyHostToDevice) == hipSuccess &&
              s->ws.alloc(s->geom.ws_bytes, true) && s->sh_rep.alloc((size_t)slots * 8) &&
              s->sh_acc.alloc((size_t)slots * (size_t)(n_acc > 0 ? n_acc : 1) * 8) && s->sh_ngroups.alloc((size_t)(s->n_units + 1) * 4) &&
              s->sh_pack_start.alloc((size_t)(s->n_units + 1) * 8) && s->sh_dense_rep.alloc((size_t)slots * 8) &&
              s->sh_order.alloc((size_t)slots * 8) && s->sh_key.alloc((size_t)slots * 8) &&
              s->sh_accs.alloc((size_t)slots * 4 * (size_t)(n_acc > 0 ? n_acc : 1)) && s->sh_mrep.alloc((size_t)s->merge_cap * 8) &&
              s->sh_macc.alloc((size_t)s->merge_cap * 8 * (size_t)(nf > 0 ? nf : 1)) && s->sh_mgroups.alloc(8) &&
              s->sh_mkey.alloc((size_t)s->merge_cap * 8) && s->sh_prog_out.alloc((size_t)s->merge_cap * 8 * HS_MAX_OUTS);
    // result image: the layout of the on-chip path (hs_result_columns / hs_result_write_blockfile read it)
    s->fin = P.fin;
    int64_t pos = 16;
    for (int o = 0; o < s->fin.n_out; ++o) {
        hs_finish_out& out = s->fin.outs[o];
        const int width = out.src == 0 ? s->key_bytes : (out.kind == HS_I64 ? 8 : 4);
        out.offset = pos;
        pos = (pos + (int64_t)s->merge_cap * width + 15) & ~(int64_t)15;
    }
    s->image_bytes = pos;
    if (s->image_host) (void)hipHostFree(s->image_host);
    s->image_host = s->image_dev = nullptr;
    ok = ok && hipHostMalloc(&s->image_host, (size_t)s->image_bytes + kPad, hipHostMallocDefault) == hipSuccess &&
         s->sh_image.alloc((size_t)s->image_bytes, true);
    if (!ok) {
        hs_set_error("hs_stage: out of device / pinned memory");
        return HS_E_LAUNCH;
    }
    memset(s->image_host, 0, (size_t)s->image_bytes + kPad);
    memset(&s->desc, 0, sizeof(s->desc));
    s->desc.key_kind = kc.kind;
    s->desc.key_len = kc.kind == HS_STR ? kc.fixed_len : 0;
    s->ready = true;
    return HS_OK;
}

// scan with one LDS dictionary per workgroup -> dense partial rows -> merge in unit order -> result columns in the image
int shared_run(hs_stage* s, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    const hs_stage_plan& P = s->plan;
    uint32_t* flags = (uint32_t*)s->engine->flags.p;
    const int n_acc = P.spec.n_acc, nf = s->fin.n_fold, unit_cap = s->geom.pad, cap = s->merge_cap;
    const int64_t slots = s->sh_slots;
    s->last_rows = 0;
    if (hipMemsetAsync(flags, 0, 4, stream) != hipSuccess) return HS_E_LAUNCH;
    if (s->n_units == 0 || slots == 0) {
        s->last_flags = 0;
        return HS_OK;
    }
    int rc = compute_key(s, stream);
    if (rc) return rc;
    rc = hs_agg_shared(stream, s->cols, P.n_cols, P.key_slot, &P.prog, &P.spec, (const hs_chunk*)s->chunks.p, s->n_units, &s->geom,
                           (int64_t*)s->sh_rep.p, (uint64_t*)s->sh_acc.p, (int32_t*)s->sh_ngroups.p, s->ws.p, flags, nullptr, nullptr);
    if (rc) return rc;
    // dense partial rows = the reference's shuffle-file content: accumulators in their stored kinds, unit of every row
    void* acc_ptrs[HS_MAX_ACC] = {};
    int32_t acc_kinds[HS_MAX_ACC] = {};
    for (int a = 0; a < n_acc; ++a) {
        acc_ptrs[a] = (char*)s->sh_accs.p + (size_t)a * (size_t)slots * 4;
        acc_kinds[a] = P.spec.is_int[a] ? HS_I32 : HS_F32;
    }
    rc = hs_agg_pack(stream, (const int64_t*)s->sh_rep.p, (const uint64_t*)s->sh_acc.p, (const int32_t*)s->sh_ngroups.p, s->n_units, unit_cap,
                     &P.spec, (int64_t*)s->sh_pack_start.p, (int64_t*)s->sh_dense_rep.p, acc_ptrs, acc_kinds, nullptr, (int64_t*)s->sh_order.p);
    if (rc) return rc;
    const int64_t* n_dense = (const int64_t*)s->sh_pack_start.p + s->n_units;
    const hs_col& kc = s->cols[P.key_slot];
    rc = hs_gather_fixed(stream, kc.data, s->key_bytes, s->table->nrows, (const int64_t*)s->sh_dense_rep.p, slots, n_dense, s->sh_key.p, flags);
    if (rc) return rc;
    // final merge (tasks.py:290-292): fold j = fold_op[j] over the dense accumulator column fold_src[j], partials in unit order
    hs_col key_dense{kc.kind, kc.kind == HS_STR ? kc.fixed_len : -1, s->sh_key.p, nullptr, nullptr};
    hs_col fold_cols[HS_MAX_ACC];
    hs_agg_spec mspec{};
    mspec.n_acc = nf;
    for (int j = 0; j < nf; ++j) {
        const int src = s->fin.fold_src[j];
        fold_cols[j] = hs_col{acc_kinds[src], -1, acc_ptrs[src], nullptr, nullptr};
        mspec.op[j] = (uint8_t)s->fin.fold_op[j];
        mspec.is_int[j] = P.spec.is_int[src];
    }
    rc = hs_agg_merge(stream, &key_dense, fold_cols, &mspec, (const int64_t*)s->sh_order.p, s->n_units, slots, n_dense, cap,
                      (int64_t*)s->sh_mrep.p, (uint64_t*)s->sh_macc.p, (int64_t*)s->sh_mgroups.p, flags);
    if (rc) return rc;  // HS_E_LIMIT: more partial rows than the on-chip merge holds
    const int64_t* ng = (const int64_t*)s->sh_mgroups.p;
    rc = hs_gather_fixed(stream, s->sh_key.p, s->key_bytes, slots, (const int64_t*)s->sh_mrep.p, cap, ng, s->sh_mkey.p, flags);
    if (rc) return rc;
    // projection after the merge (AVG = sum / count ...): the merged cells are its columns (slot -> key / fold j)
    int n_prog_out = 0;
    for (int o = 0; o < s->fin.n_out; ++o)
        if (s->fin.outs[o].src == 2 && s->fin.outs[o].index + 1 > n_prog_out) n_prog_out = s->fin.outs[o].index + 1;
    int32_t prog_kinds[HS_MAX_OUTS] = {};
    if (n_prog_out > 0) {
        hs_col pcols[HS_MAX_COLS];
        for (int i = 0; i < HS_MAX_COLS; ++i) {
            const int j = s->fin.prog_src[i];
            if (j >= 0 && j < nf) pcols[i] = hs_col{mspec.is_int[j] ? HS_I64 : HS_F64, -1, (char*)s->sh_macc.p + (size_t)j * (size_t)cap * 8, nullptr, nullptr};
            else pcols[i] = hs_col{kc.kind == HS_STR ? HS_U8 : kc.kind, -1, s->sh_mkey.p, nullptr, nullptr};
        }
        void* outs[HS_MAX_OUTS] = {};
        for (int o = 0; o < s->fin.n_out; ++o) {
            const hs_finish_out& d = s->fin.outs[o];
            if (d.src == 2) prog_kinds[d.index] = d.kind == HS_F32 ? HS_F64 : HS_I64;
        }
        for (int k = 0; k < n_prog_out; ++k) outs[k] = (char*)s->sh_prog_out.p + (size_t)k * (size_t)cap * 8;
        rc = hs_eval(stream, pcols, HS_MAX_COLS, &P.fin_prog, nullptr, cap, ng, outs, prog_kinds, n_prog_out, flags);
        if (rc) return rc;
    }
    // result columns in their stored kinds, at the image's offsets
    for (int o = 0; o < s->fin.n_out; ++o) {
        const hs_finish_out& d = s->fin.outs[o];
        char* dst = (char*)s->sh_image.p + d.offset;
        if (d.src == 0) {
            if (hipMemcpyAsync(dst, s->sh_mkey.p, (size_t)cap * (size_t)s->key_bytes, hipMemcpyDeviceToDevice, stream) != hipSuccess) return HS_E_LAUNCH;
            continue;
        }
        const void* cells = d.src == 1 ? (const char*)s->sh_macc.p + (size_t)d.index * (size_t)cap * 8
                                       : (const char*)s->sh_prog_out.p + (size_t)d.index * (size_t)cap * 8;
        const bool is_int = d.src == 1 ? mspec.is_int[d.index] != 0 : prog_kinds[d.index] == HS_I64;
        if (d.kind == HS_I64) {
            if (hipMemcpyAsync(dst, cells, (size_t)cap * 8, hipMemcpyDeviceToDevice, stream) != hipSuccess) return HS_E_LAUNCH;
        } else {
            rc = hs_quantise(stream, cells, is_int ? HS_I64 : HS_F64, cap, ng, dst, flags);
            if (rc) return rc;
        }
    }
    int64_t n_groups = 0;
    uint32_t f = 0;
    if (hipMemcpyAsync((char*)s->image_host, s->sh_image.p, (size_t)s->image_bytes, hipMemcpyDeviceToHost, stream) != hipSuccess ||
        hipMemcpyAsync(&n_groups, ng, 8, hipMemcpyDeviceToHost, stream) != hipSuccess ||
        hipMemcpyAsync(&f, flags, 4, hipMemcpyDeviceToHost, stream) != hipSuccess || hipStreamSynchronize(stream) != hipSuccess)
        return HS_E_LAUNCH;
    s->last_flags = f;
    s->last_rows = n_groups < cap ? n_groups : cap;
    return HS_OK;
}

}  // namespace

extern "C" int hs_stage_prepare(hs_engine* e, hs_table* t, const hs_stage_plan* plan, size_t plan_bytes, int32_t world,
                                hs_stage** out) {
    if (!e || !t || !plan || !out || plan_bytes != sizeof(hs_stage_plan) || plan->version != HS_STAGE_PLAN_VERSION ||
        plan->n_cols < 1 || plan->n_cols > HS_MAX_COLS || plan->key_slot < 0 || plan->key_slot >= plan->n_cols || world < 1) {
        hs_set_error("hs_stage_prepare: bad plan blob (size %zu, expected %zu)", plan_bytes, sizeof(hs_stage_plan));
        return HS_E_ARG;
    }
    if (hipSetDevice(e->device) != hipSuccess) return HS_E_LAUNCH;
    if (plan->key_computed && (plan->n_kcols < 1 || plan->n_kcols > HS_MAX_COLS || plan->key_prog.n_ins < 1)) {
        hs_set_error("hs_stage_prepare: a computed key needs its program and columns");
        return HS_E_ARG;
    }
    // only the columns the programs name are read from the file
    int32_t want[2 * HS_MAX_COLS];
    int32_t n_want = 0;
    auto add = [&](int32_t c) {
        for (int k = 0; k < n_want; ++k)
            if (want[k] == c) return;
        want[n_want++] = c;
    };
    for (int i = 0; i < plan->n_cols; ++i)
        if (i != plan->key_slot || !plan->key_computed) add(plan->col_ids[i]);
    for (int i = 0; plan->key_computed && i < plan->n_kcols; ++i) add(plan->kcol_ids[i]);
    int rc = n_want ? hs_table_load(e, t, want, n_want) : HS_OK;
    if (rc) return rc;
    hs_stage* s = new hs_stage();
    s->engine = e;
    s->table = t;
    s->plan = *plan;
    s->world = world;
    s->group_cap = plan->group_cap > 0 ? plan->group_cap : 4;
    s->merge_cap = plan->merge_cap > 0 ? plan->merge_cap : 16;
    rc = stage_prepare(s);
    if (rc) {
        delete s;
        return rc;
    }
    *out = s;
    return HS_OK;
}

extern "C" void hs_stage_destroy(hs_stage* s) { delete s; }

// One query on one GPU: scan + partial aggregate -> final merge + projection -> result image on the host.  A dictionary
// overflow (HS_FLAG_DICT_FULL) grows the capacities and runs again; capacities beyond the on-chip tiers: HS_E_LIMIT.
extern "C" int hs_stage_run(hs_stage* s, void* stream, uint32_t* flags_out, int64_t* n_rows_out) {
    if (!s) {
        hs_set_error("hs_stage_run: null stage");
        return HS_E_ARG;
    }
    if (s->world != 1) {
        hs_set_error("hs_stage_run: a multi-rank stage runs as hs_stage_launch_partial / collective / hs_stage_launch_finish");
        return HS_E_ARG;
    }
    for (int attempt = 0; attempt < 14; ++attempt) {
        int rc = HS_OK;
        if (s->tier == 1) {
            // tens to thousands of groups per block: shared-dictionary scan + the general operator sequence
            if (!s->ready) rc = shared_prepare(s);
            if (!rc) rc = shared_run(s, stream);
            if (rc) return rc;
            ++s->runs;
            if (s->last_flags & HS_FLAG_MERGE_ROWS) {
                hs_set_error("hs_stage_run: more partial rows than the on-chip final merge holds (the HBM tier belongs to the per-operator ABI)");
                return HS_E_LIMIT;
            }
            if (s->last_flags & (HS_FLAG_DICT_FULL | HS_FLAG_MERGE_FULL)) {
                const bool unit_full = s->last_flags & HS_FLAG_DICT_FULL, merge_full = s->last_flags & HS_FLAG_MERGE_FULL;
                if ((unit_full && s->group_cap >= 4096) || (merge_full && s->merge_cap >= 4096)) {
                    hs_set_error("hs_stage_run: GROUP BY cardinality exceeds the on-chip tiers of this path");
                    return HS_E_LIMIT;
                }
                if (unit_full) s->group_cap *= 4;
                if (merge_full) s->merge_cap *= 4;
                if (s->merge_cap < s->group_cap) s->merge_cap = s->group_cap;
                if (s->merge_cap > 4096) s->merge_cap = 4096;
                s->ready = false;
                ++s->grows;
                continue;
            }
            if (flags_out) *flags_out = s->last_flags;
            if (n_rows_out) *n_rows_out = s->last_rows;
            return HS_OK;
        }
        if (!s->ready) rc = stage_prepare(s);
        if (rc == HS_E_LIMIT && s->world == 1) {  // the per-lane tables do not hold this query: the shared dictionary may
            s->tier = 1;
            if (s->group_cap < 16) s->group_cap = 16;
            if (s->merge_cap < 64) s->merge_cap = 64;
            continue;
        }
        if (rc) return rc;
        if (s->capture) {
            rc = hs_capture_replay(s->capture, stream);
            ++s->replays;
        } else {
            const bool record = s->runs >= 1;  // the second run with these capacities is the one that is kept
            if (record) rc = hs_capture_begin();
            if (!rc) rc = launch_partial(s, stream);
            if (!rc) rc = launch_finish(s, stream, nullptr, 1);
            if (record) {
                int32_t n_ops = 0;
                void* handle = nullptr;
                const int rc2 = hs_capture_end(&handle, &n_ops);
                if (!rc && !rc2 && n_ops > 0) s->capture = handle;
                else if (handle) hs_capture_free(handle);
            }
        }
        if (rc) return rc;
        rc = wait_result(s, stream);
        if (rc) return rc;
        ++s->runs;
        if (s->last_flags & (HS_FLAG_DICT_FULL | HS_FLAG_MERGE_FULL)) {
            // more groups than a dictionary was sized for: x2 per workgroup (per-lane tables), x4 for the merge - each
            // grows on its own flag; the merge also keeps up with the per-unit capacity (it holds at least as many keys)
            const bool unit_full = s->last_flags & HS_FLAG_DICT_FULL, merge_full = s->last_flags & HS_FLAG_MERGE_FULL;
            if (merge_full && s->merge_cap >= 4096) {
                hs_set_error("hs_stage_run: GROUP BY cardinality exceeds the on-chip tiers of this path");
                return HS_E_LIMIT;
            }
            if (unit_full && s->group_cap >= 16) {
                // past the per-lane tables: the shared-dictionary tier takes over (round 3; round 2 answered HS_E_LIMIT here)
                if (s->capture) {
                    hs_capture_free(s->capture);
                    s->capture = nullptr;
                }
                s->tier = 1;
                s->group_cap = 64;
                if (s->merge_cap < 256) s->merge_cap = 256;
                s->ready = false;
                s->runs = 0;
                ++s->grows;
                continue;
            }
            if (unit_full) s->group_cap *= 2;
            if (merge_full) s->merge_cap *= 4;
            if (s->merge_cap < 4 * s->group_cap) s->merge_cap = 4 * s->group_cap;
            s->ready = false;
            s->runs = 0;
            ++s->grows;
            continue;
        }
        if (flags_out) *flags_out = s->last_flags;
        if (n_rows_out) *n_rows_out = s->last_rows;
        return HS_OK;
    }
    hs_set_error("hs_stage_run: capacities did not settle");
    return HS_E_LIMIT;
}

// Multi-rank form: launch the rank's scan into its slab, let the caller all-gather the slabs (hs_stage_slab: device
// pointer + bytes), then launch the finish over the gathered slabs and wait.
extern "C" int hs_stage_launch_partial(hs_stage* s, void* stream) {
    if (!s) return HS_E_ARG;
    if (!s->ready) {
        const int rc = stage_prepare(s);
        if (rc) return rc;
    }
    return launch_partial(s, stream);
}
extern "C" void* hs_stage_slab(hs_stage* s, int64_t* bytes) {
    if (!s || !s->ready) return nullptr;
    if (bytes) *bytes = s->slab_bytes;
    return s->slab.p;
}
extern "C" int hs_stage_launch_finish(hs_stage* s, void* stream, const void* gathered, int32_t world) {
    if (!s || !s->ready || world < 1) return HS_E_ARG;
    return launch_finish(s, stream, gathered, world);
}
extern "C" int hs_stage_wait(hs_stage* s, void* stream, uint32_t* flags_out, int64_t* n_rows_out) {
    if (!s || !s->ready) return HS_E_ARG;
    const int rc = wait_result(s, stream);
    if (rc) return rc;
    if (flags_out) *flags_out = s->last_flags;
    if (n_rows_out) *n_rows_out = s->last_rows;
    return HS_OK;
}
// After HS_FLAG_DICT_FULL / HS_FLAG_MERGE_FULL in the multi-rank form (every rank sees the same flags): grow and prepare again.
extern "C" int hs_stage_grow(hs_stage* s) {
    if (!s) return HS_E_ARG;
    const bool unit_full = s->last_flags & HS_FLAG_DICT_FULL, merge_full = s->last_flags & HS_FLAG_MERGE_FULL;
    if ((unit_full && s->group_cap >= 16) || (merge_full && s->merge_cap >= 4096)) return HS_E_LIMIT;
    if (unit_full) s->group_cap *= 2;
    if (merge_full || !unit_full) s->merge_cap = s->merge_cap < 4096 ? s->merge_cap * 4 : s->merge_cap;
    if (s->merge_cap < 4 * s->group_cap) s->merge_cap = 4 * s->group_cap;
    s->ready = false;
    ++s->grows;
    return stage_prepare(s);
}

extern "C" int hs_stage_stats(const hs_stage* s, int64_t* stats) {
    if (!s || !stats) return HS_E_ARG;
    stats[0] = s->runs;
    stats[1] = s->replays;
    stats[2] = s->grows;
    stats[3] = s->group_cap;
    stats[4] = s->merge_cap;
    stats[5] = s->geom.n_chunks;
    return HS_OK;
}

// ---- results -----------------------------------------------------------------------------------------------
extern "C" int hs_result_columns(const hs_stage* s, hs_result_col* out, int32_t cap, int32_t* n) {
    if (!s || !s->ready || !out || !n) {
        hs_set_error("hs_result_columns: bad arguments");
        return HS_E_ARG;
    }
    *n = s->fin.n_out;
    for (int o = 0; o < s->fin.n_out && o < cap; ++o) {
        const hs_finish_out& d = s->fin.outs[o];
        out[o].kind = d.src == 0 ? s->desc.key_kind : d.kind;
        out[o].width = d.src == 0 ? s->key_bytes : (d.kind == HS_I64 ? 8 : 4);
        out[o].data = (const uint8_t*)s->image_host + d.offset;
        out[o].n_rows = s->last_rows;
    }
    return HS_OK;
}

// The result as a one-block BlockFile (reference: WriteToLocalFileTask.write tasks.py:400-410, io.py:47-109): what
// `collect_results` reads back.  Column names / types come from the plan blob.
extern "C" int hs_result_write_blockfile(const hs_stage* s, const char* path) {
    if (!s || !s->ready || !path) {
        hs_set_error("hs_result_write_blockfile: bad arguments");
        return HS_E_ARG;
    }
    if (s->last_rows == 0) return HS_OK;  // empty result: the reference writes no file (tasks.py:405)
    FILE* f = fopen(path, "wb");
    if (!f) {
        hs_set_error("hs_result_write_blockfile: cannot create %s", path);
        return HS_E_ARG;
    }
    const int n_out = s->fin.n_out;
    const uint8_t nc = (uint8_t)n_out;
    fwrite(&nc, 1, 1, f);
    for (int o = 0; o < n_out; ++o) {
        const uint8_t type = (uint8_t)s->plan.out_types[o];
        const uint8_t len = (uint8_t)strnlen(s->plan.out_names[o], sizeof(s->plan.out_names[o]));
        fwrite(&type, 1, 1, f);
        fwrite(&len, 1, 1, f);
        fwrite(s->plan.out_names[o], 1, len, f);
    }
    const uint64_t block_start = (uint64_t)ftell(f);
    const uint32_t rows = (uint32_t)s->last_rows;
    fwrite(&rows, 4, 1, f);
    for (int o = 0; o < n_out; ++o) {
        const hs_finish_out& d = s->fin.outs[o];
        const uint8_t* col = (const uint8_t*)s->image_host + d.offset;
        const bool is_key_string = d.src == 0 && s->desc.key_kind == HS_STR;
        const int width = d.src == 0 ? s->key_bytes : (d.kind == HS_I64 ? 8 : 4);
        const uint64_t bytes = (uint64_t)rows * (uint64_t)width + (is_key_string ? rows : 0);
        fwrite(&bytes, 8, 1, f);
        if (is_key_string) {  // STRING payload: the length bytes, then the strings
            const uint8_t w = (uint8_t)width;
            for (uint32_t r = 0; r < rows; ++r) fwrite(&w, 1, 1, f);
        }
        fwrite(col, 1, (size_t)rows * (size_t)width, f);
    }
    fwrite(&block_start, 8, 1, f);
    const uint32_t nblocks = 1;
    fwrite(&nblocks, 4, 1, f);
    const bool ok = fclose(f) == 0;
    if (!ok) {
        hs_set_error("hs_result_write_blockfile: write to %s failed", path);
        return HS_E_ARG;
    }
    return HS_OK;
}

// =====================================================================================================================
// Round 3: the JOIN stage behind the same boundary - the reference's JoinJob (jobs.py:45-79, plan.py:99-109: one per
// shuffle partition; tasks.py:201-240 build + probe, tasks.py:284-289 partial aggregate) and the final stage after it,
// end to end without Python: BlockFile reader for BOTH tables, dictionary coding of the one build-side column the
// aggregate reads, key range, byte table (hs_join8_build), probe inside the aggregate scan (hs_agg_shared_join8), raw unit
// tables -> exchange slab (hs_agg_units_to_slab), finish launch, result image, result BlockFile.  Covers the primary-key /
// foreign-key case of DESIGN.md 4.6 (INTEGER keys, dense key range, unique build keys, GROUP BY the build-side column or a
// probe-side column of at most 4 bytes); anything else returns HS_E_LIMIT and belongs to the per-operator ABI.
// =====================================================================================================================
extern "C" int hs_dict_build(void* stream, const hs_col* col, int64_t nrows, int32_t cap, uint64_t* slot_words, int64_t* slot_reps,
                             int32_t* count, uint32_t* flags);
extern "C" int hs_dict_assign(void* stream, const hs_col* col, int64_t nrows, int32_t cap, uint64_t* slot_words, int64_t* slot_reps,
                              const uint8_t* slot_code, uint8_t* out_codes, uint32_t* flags);
extern "C" int hs_minmax_i32(void* stream, const int32_t* values, int64_t n, int32_t* minmax);

struct hs_join_stage {
    hs_engine* engine = nullptr;
    hs_table *build = nullptr, *probe = nullptr;
    hs_join_stage_plan plan{};
    std::vector<std::string> dict;  // the payload column's distinct strings, sorted: code = index
    DevBuf codes;                   // one code byte per build row
    int32_t key_min = 0;
    int64_t slots = 0;
    DevBuf table, build_ws;
    hs_col cols[HS_MAX_COLS + 1]{};
    hs_agg_geom geom{};
    DevBuf chunks, out_rep, xbuf, ws, slab, scratch;
    int32_t group_cap = 4, merge_cap = 16, unit_cap = 0, n_units = 0, key_bytes = 1, key_kind = HS_STR;
    bool key_is_payload = false, ready = false;
    hs_slab_desc desc{};
    hs_finish_spec fin{};
    int64_t image_bytes = 0;
    void *image_host = nullptr, *image_dev = nullptr, *capture = nullptr;
    int64_t runs = 0, replays = 0, grows = 0;
    uint32_t last_flags = 0;
    int64_t last_rows = 0;
    ~hs_join_stage() {
        if (capture) hs_capture_free(capture);
        if (image_host) (void)hipHostFree(image_host);
    }
};

namespace {

// bytes of row `row` of a STRING column (device) -> host; false on a copy error
bool fetch_string(const hs_col& c, int64_t row, std::string& out) {
    uint8_t len = 0;
    int64_t off = 0;
    if (c.fixed_len >= 0) {
        len = (uint8_t)c.fixed_len;
        off = row * (int64_t)c.fixed_len;
    } else {
        if (hipMemcpy(&len, c.lens + row, 1, hipMemcpyDeviceToHost) != hipSuccess) return false;
        if (hipMemcpy(&off, c.offs + row, 8, hipMemcpyDeviceToHost) != hipSuccess) return false;
    }
    out.assign((size_t)len, '\0');
    return len == 0 || hipMemcpy(&out[0], (const char*)c.data + off, len, hipMemcpyDeviceToHost) == hipSuccess;
}

// Device.dict_encode natively: distinct strings of the column (device set, representative rows read back), sorted, codes
// named BY STRING, one code byte per row.  HS_E_LIMIT when the column has more than 255 distinct values.
int join_encode_payload(hs_join_stage* s, const hs_col& col, int64_t n) {
    const int32_t cap = 4096;
    DevBuf words, reps, state, slot_code;
    if (!words.alloc((size_t)cap * 8) || !reps.alloc((size_t)cap * 8) || !state.alloc(8, true) || !slot_code.alloc(cap, true) ||
        !s->codes.alloc((size_t)(n > 0 ? n : 1))) {
        hs_set_error("hs_join_stage: out of device memory");
        return HS_E_LAUNCH;
    }
    int rc = hs_dict_build(nullptr, &col, n, cap, (uint64_t*)words.p, (int64_t*)reps.p, (int32_t*)state.p, (uint32_t*)state.p + 1);
    if (rc) return rc;
    int32_t st[2] = {0, 0};
    std::vector<int64_t> host_reps((size_t)cap);
    if (hipMemcpy(st, state.p, 8, hipMemcpyDeviceToHost) != hipSuccess ||
        hipMemcpy(host_reps.data(), reps.p, (size_t)cap * 8, hipMemcpyDeviceToHost) != hipSuccess)
        return HS_E_LAUNCH;
    if (st[1]) {
        hs_set_error("hs_join_stage: the build-side column has too many distinct values for a code byte");
        return HS_E_LIMIT;
    }
    std::vector<std::pair<int, std::string>> found;  // (slot, string)
    for (int sl = 0; sl < cap; ++sl) {
        if (host_reps[(size_t)sl] < 0) continue;
        std::string text;
        if (!fetch_string(col, host_reps[(size_t)sl], text)) return HS_E_LAUNCH;
        found.emplace_back(sl, std::move(text));
    }
    s->dict.clear();
    for (const auto& f : found) s->dict.push_back(f.second);
    std::sort(s->dict.begin(), s->dict.end());
    s->dict.erase(std::unique(s->dict.begin(), s->dict.end()), s->dict.end());
    if (s->dict.size() > 255) {
        hs_set_error("hs_join_stage: the build-side column has %zu distinct values (> 255)", s->dict.size());
        return HS_E_LIMIT;
    }
    std::vector<uint8_t> codes_of_slot((size_t)cap, 0);
    for (const auto& f : found)
        codes_of_slot[(size_t)f.first] = (uint8_t)(std::lower_bound(s->dict.begin(), s->dict.end(), f.second) - s->dict.begin());
    if (hipMemcpy(slot_code.p, codes_of_slot.data(), (size_t)cap, hipMemcpyHostToDevice) != hipSuccess) return HS_E_LAUNCH;
    if (n > 0) {
        rc = hs_dict_assign(nullptr, &col, n, cap, (uint64_t*)words.p, (int64_t*)reps.p, (const uint8_t*)slot_code.p, (uint8_t*)s->codes.p,
                            (uint32_t*)state.p + 1);
        if (rc) return rc;
    }
    if (hipDeviceSynchronize() != hipSuccess) return HS_E_LAUNCH;  // the temporaries above go out of scope
    return HS_OK;
}

int join_prepare_aggregate(hs_join_stage* s) {
    const hs_join_stage_plan& P = s->plan;
    if (s->capture) {
        hs_capture_free(s->capture);
        s->capture = nullptr;
    }
    s->ready = false;
    s->n_units = P.n_parts;
    int cap = 16;
    while (cap < (s->group_cap > 4 ? s->group_cap : 4) * s->n_units) cap *= 2;
    if (cap > 4096) cap = 4096;
    const int64_t unit_rows[2] = {0, s->probe->nrows};
    int rc = hs_agg_shared_geom(unit_rows, 1, P.spec.n_acc, cap, &s->geom);
    if (rc) return rc;
    std::vector<hs_chunk> chunks((size_t)(s->geom.n_chunks > 0 ? s->geom.n_chunks : 1));
    int64_t chunk0[2] = {0, 0};
    rc = hs_agg_partial_chunks(unit_rows, 1, &s->geom, chunks.data(), chunk0);
    if (rc) return rc;
    int per_unit = cap / s->n_units, small = 16;
    if (per_unit < 4) per_unit = 4;
    while (small < 4 * per_unit) small *= 2;
    if (s->geom.pad > small) s->geom.pad = small;  // slots of ONE unit's table
    s->unit_cap = s->geom.pad;
    const int64_t slots = (int64_t)s->n_units * s->unit_cap;
    const int n_acc = P.spec.n_acc;
    bool ok = s->chunks.alloc(chunks.size() * sizeof(hs_chunk)) &&
              hipMemcpy(s->chunks.p, chunks.data(), chunks.size() * sizeof(hs_chunk), hipMemcpyHostToDevice) == hipSuccess &&
              s->out_rep.alloc((size_t)slots * 8) && s->xbuf.alloc((size_t)(16 + slots * 8 * (1 + n_acc)), true) &&
              s->ws.alloc((size_t)(s->geom.n_chunks > 0 ? s->geom.n_chunks : 1) * (size_t)slots * (size_t)(n_acc > 0 ? n_acc : 1) * 8, true);
    // slab: header | order key i64 x M | key column | accumulator columns (4 bytes per row); unit u owns rows [u * cap, (u + 1) * cap)
    hs_slab_desc& d = s->desc;
    memset(&d, 0, sizeof(d));
    d.slab_rows = slots;
    int64_t pos = 16;
    d.order_off = pos;
    pos += 8 * d.slab_rows;
    pos = (pos + 15) & ~(int64_t)15;
    d.key_off = pos;
    pos += (int64_t)s->key_bytes * d.slab_rows;
    d.n_acc = n_acc;
    for (int a = 0; a < n_acc; ++a) {
        pos = (pos + 15) & ~(int64_t)15;
        d.acc_off[a] = pos;
        d.acc_kind[a] = P.spec.is_int[a] ? HS_I32 : HS_F32;
        pos += 4 * d.slab_rows;
    }
    d.stride = (pos + 15) & ~(int64_t)15;
    d.key_kind = s->key_kind;
    d.key_len = s->key_kind == HS_STR ? s->key_bytes : 0;
    ok = ok && s->slab.alloc((size_t)d.stride, true);
    s->fin = P.fin;
    pos = 16;
    for (int o = 0; o < s->fin.n_out; ++o) {
        hs_finish_out& out = s->fin.outs[o];
        const int width = out.src == 0 ? s->key_bytes : (out.kind == HS_I64 ? 8 : 4);
        out.offset = pos;
        pos = (pos + (int64_t)s->merge_cap * width + 15) & ~(int64_t)15;
    }
    s->image_bytes = pos;
    if (s->image_host) (void)hipHostFree(s->image_host);
    s->image_host = s->image_dev = nullptr;
    ok = ok && hipHostMalloc(&s->image_host, (size_t)s->image_bytes + kPad, hipHostMallocMapped) == hipSuccess &&
         hipHostGetDevicePointer(&s->image_dev, s->image_host, 0) == hipSuccess && s->image_dev;
    if (ok) memset(s->image_host, 0, (size_t)s->image_bytes + kPad);
    ok = ok && s->scratch.alloc(hs_agg_finish_scratch_bytes(s->merge_cap, s->fin.n_fold), true);
    if (!ok) {
        hs_set_error("hs_join_stage: out of device / pinned memory");
        return HS_E_LAUNCH;
    }
    s->ready = true;
    return HS_OK;
}

int join_launch(hs_join_stage* s, void* stream) {
    const hs_join_stage_plan& P = s->plan;
    uint32_t* flags = (uint32_t*)s->engine->flags.p;
    const hs_col& bk = s->build->cols[P.build_key_col].col;
    int rc = hs_join8_build(stream, (const int32_t*)bk.data, P.build_payload_col >= 0 ? (const uint8_t*)s->codes.p : nullptr,
                            s->build->nrows, 0, nullptr, s->key_min, s->slots, (uint8_t*)s->table.p, s->build_ws.p, flags);
    if (rc) return rc;
    const hs_join8 J{(const uint8_t*)s->table.p, s->slots, s->key_min, P.n_parts};
    const int64_t slots = (int64_t)s->n_units * s->unit_cap;
    uint64_t* keys = (uint64_t*)((char*)s->xbuf.p + 16);
    uint64_t* acc = keys + slots;
    rc = hs_agg_shared_join8(stream, s->cols, P.n_cols + 1, P.key_slot, P.n_cols, &J, s->n_units, &P.prog, &P.spec,
                             (const hs_chunk*)s->chunks.p, &s->geom, (int64_t*)s->out_rep.p, keys, acc, s->ws.p, flags, nullptr, nullptr);
    if (rc) return rc;
    rc = hs_agg_units_to_slab(stream, keys, acc, s->n_units, s->unit_cap, &P.spec, (uint8_t*)s->slab.p, &s->desc, flags);
    if (rc) return rc;
    return hs_agg_finish(stream, (const uint8_t*)s->slab.p, 1, &s->desc, &s->fin, P.fin_prog.n_ins ? &P.fin_prog : nullptr,
                         s->n_units, s->merge_cap, (uint8_t*)s->image_dev, s->scratch.p, flags, (uint32_t*)s->slab.p);
}

}  // namespace

extern "C" int hs_join_stage_prepare(hs_engine* e, hs_table* build, hs_table* probe, const hs_join_stage_plan* plan,
                                     size_t plan_bytes, hs_join_stage** out) {
    if (!e || !build || !probe || !plan || !out || plan_bytes != sizeof(hs_join_stage_plan) ||
        plan->version != HS_JOIN_STAGE_PLAN_VERSION || plan->n_cols < 1 || plan->n_cols >= 8 /* HS_FUSED_COLS: preloaded slots, one is the unit column */ ||
        plan->key_slot < 0 || plan->key_slot >= plan->n_cols || plan->n_parts < 1 || plan->n_parts > 127 ||
        plan->build_key_col < 0 || plan->build_key_col >= (int)build->cols.size() || plan->probe_key_col < 0 ||
        plan->probe_key_col >= (int)probe->cols.size() || plan->build_payload_col >= (int)build->cols.size()) {
        hs_set_error("hs_join_stage_prepare: bad plan blob (size %zu, expected %zu)", plan_bytes, sizeof(hs_join_stage_plan));
        return HS_E_ARG;
    }
    if (hipSetDevice(e->device) != hipSuccess) return HS_E_LAUNCH;
    if (kind_of_type(build->cols[plan->build_key_col].type) != HS_I32 || kind_of_type(probe->cols[plan->probe_key_col].type) != HS_I32) {
        hs_set_error("hs_join_stage_prepare: join keys must be INTEGER columns");
        return HS_E_LIMIT;
    }
    // read what the query references: build key (+ payload), probe key + the program's probe-side columns
    std::vector<int32_t> bcols{plan->build_key_col}, pcols{plan->probe_key_col};
    if (plan->build_payload_col >= 0) bcols.push_back(plan->build_payload_col);
    for (int i = 0; i < plan->n_cols; ++i) {
        if (plan->col_ids[i] >= (int)probe->cols.size()) {
            hs_set_error("hs_join_stage_prepare: slot %d names probe column %d", i, plan->col_ids[i]);
            return HS_E_ARG;
        }
        if (plan->col_ids[i] >= 0) pcols.push_back(plan->col_ids[i]);
    }
    int rc = hs_table_load(e, build, bcols.data(), (int32_t)bcols.size());
    if (!rc) rc = hs_table_load(e, probe, pcols.data(), (int32_t)pcols.size());
    if (rc) return rc;
    if (build->nrows <= 0 || build->nrows >= 0xffffffffll || probe->nrows >= 0xffffffffll) {
        hs_set_error("hs_join_stage_prepare: empty build side or more than 2^32 rows");
        return HS_E_LIMIT;
    }
    hs_join_stage* s = new hs_join_stage();
    s->engine = e;
    s->build = build;
    s->probe = probe;
    s->plan = *plan;
    s->group_cap = plan->group_cap > 0 ? plan->group_cap : 4;
    s->merge_cap = plan->merge_cap > 0 ? plan->merge_cap : 16;
    auto fail = [&](int code) {
        delete s;
        return code;
    };
    if (plan->build_payload_col >= 0) {
        const hs_col& pc = build->cols[plan->build_payload_col].col;
        if (pc.kind != HS_STR) {
            hs_set_error("hs_join_stage_prepare: the build-side column must be a STRING column");
            return fail(HS_E_LIMIT);
        }
        rc = join_encode_payload(s, pc, build->nrows);
        if (rc) return fail(rc);
    }
    // key range of the build side -> direct addressing
    DevBuf mm;
    int32_t minmax[2] = {0, 0};
    const hs_col& bk = build->cols[plan->build_key_col].col;
    if (!mm.alloc(8) || hs_minmax_i32(nullptr, (const int32_t*)bk.data, build->nrows, (int32_t*)mm.p) != HS_OK ||
        hipMemcpy(minmax, mm.p, 8, hipMemcpyDeviceToHost) != hipSuccess)
        return fail(HS_E_LAUNCH);
    s->key_min = minmax[0];
    s->slots = (int64_t)minmax[1] - (int64_t)minmax[0] + 1;
    if (s->slots > (1ll << 30) || s->slots > 32 * build->nrows) {
        hs_set_error("hs_join_stage_prepare: the build side's key range (%lld slots for %lld keys) is too sparse for the byte table",
                     (long long)s->slots, (long long)build->nrows);
        return fail(HS_E_LIMIT);
    }
    if (!s->table.alloc(hs_join8_table_bytes(s->slots)) || !s->build_ws.alloc(hs_join8_ws_bytes(build->nrows, s->slots))) {
        hs_set_error("hs_join_stage_prepare: out of device memory");
        return fail(HS_E_LAUNCH);
    }
    // column slots of the program: probe-side columns as loaded, the payload as the virtual code column, the unit column last
    const hs_col& pk = probe->cols[plan->probe_key_col].col;
    for (int i = 0; i < plan->n_cols; ++i) {
        if (plan->col_ids[i] >= 0) s->cols[i] = probe->cols[plan->col_ids[i]].col;
        else s->cols[i] = hs_col{HS_JOIN8_CODE, 1, pk.data, nullptr, nullptr};
    }
    s->cols[plan->n_cols] = hs_col{HS_JOIN8_UNIT, -1, pk.data, nullptr, nullptr};
    const hs_col& kc = s->cols[plan->key_slot];
    s->key_is_payload = kc.kind == HS_JOIN8_CODE;
    if (s->key_is_payload) {
        s->key_kind = HS_STR;
        s->key_bytes = 1;
    } else if (kc.kind == HS_I32) {
        s->key_kind = HS_I32;
        s->key_bytes = 4;
    } else if (kc.kind == HS_STR && (kc.fixed_len == 1 || kc.fixed_len == 2 || kc.fixed_len == 4)) {
        s->key_kind = HS_STR;
        s->key_bytes = kc.fixed_len;
    } else {
        hs_set_error("hs_join_stage_prepare: the GROUP BY key must be the build-side column, an INTEGER or a short fixed string");
        return fail(HS_E_LIMIT);
    }
    rc = join_prepare_aggregate(s);
    if (rc) return fail(rc);
    *out = s;
    return HS_OK;
}

extern "C" void hs_join_stage_destroy(hs_join_stage* s) { delete s; }

extern "C" int hs_join_stage_run(hs_join_stage* s, void* stream, uint32_t* flags_out, int64_t* n_rows_out) {
    if (!s) {
        hs_set_error("hs_join_stage_run: null stage");
        return HS_E_ARG;
    }
    for (int attempt = 0; attempt < 12; ++attempt) {
        int rc = HS_OK;
        if (!s->ready) rc = join_prepare_aggregate(s);
        if (rc) return rc;
        if (s->capture) {
            rc = hs_capture_replay(s->capture, stream);
            ++s->replays;
        } else {
            const bool record = s->runs >= 1;
            if (record) rc = hs_capture_begin();
            if (!rc) rc = join_launch(s, stream);
            if (record) {
                int32_t n_ops = 0;
                void* handle = nullptr;
                const int rc2 = hs_capture_end(&handle, &n_ops);
                if (!rc && !rc2 && n_ops > 0) s->capture = handle;
                else if (handle) hs_capture_free(handle);
            }
        }
        if (rc) return rc;
        volatile uint32_t* done = (volatile uint32_t*)s->image_host + 1;
        for (int64_t spins = 0; *done == 0; ++spins) {
            if (spins > 2000000) {
                if (hipStreamSynchronize((hipStream_t)stream) != hipSuccess || *done == 0) {
                    hs_set_error("hs_join_stage_run: the finish launch did not hand its result over");
                    return HS_E_LAUNCH;
                }
            }
        }
        std::atomic_thread_fence(std::memory_order_acquire);
        s->last_flags = *(volatile uint32_t*)s->image_host;
        const int64_t n = *(volatile int64_t*)((char*)s->image_host + 8);
        s->last_rows = n < s->merge_cap ? n : s->merge_cap;
        *done = 0;
        ++s->runs;
        if (s->last_flags & HS_FLAG_JOIN_DUP) {
            hs_set_error("hs_join_stage_run: the build side holds a key twice: not a primary-key / foreign-key join (use hs_join_build / count / fill)");
            return HS_E_LIMIT;
        }
        if (s->last_flags & (HS_FLAG_DICT_FULL | HS_FLAG_MERGE_FULL)) {
            const bool unit_full = s->last_flags & HS_FLAG_DICT_FULL, merge_full = s->last_flags & HS_FLAG_MERGE_FULL;
            if ((unit_full && s->group_cap * s->n_units >= 4096) || (merge_full && s->merge_cap >= 4096)) {
                hs_set_error("hs_join_stage_run: GROUP BY cardinality exceeds the on-chip tiers of this path");
                return HS_E_LIMIT;
            }
            if (unit_full) s->group_cap *= 4;
            if (merge_full) s->merge_cap *= 4;
            if (s->merge_cap < 4 * s->group_cap) s->merge_cap = 4 * s->group_cap;
            if (s->merge_cap > 4096) s->merge_cap = 4096;
            s->ready = false;
            s->runs = 0;
            ++s->grows;
            continue;
        }
        if (flags_out) *flags_out = s->last_flags;
        if (n_rows_out) *n_rows_out = s->last_rows;
        return HS_OK;
    }
    hs_set_error("hs_join_stage_run: capacities did not settle");
    return HS_E_LIMIT;
}

extern "C" int hs_join_stage_stats(const hs_join_stage* s, int64_t* stats) {
    if (!s || !stats) return HS_E_ARG;
    stats[0] = s->runs;
    stats[1] = s->replays;
    stats[2] = s->grows;
    stats[3] = s->group_cap;
    stats[4] = s->merge_cap;
    stats[5] = (int64_t)s->dict.size();
    stats[6] = s->slots;
    stats[7] = s->unit_cap;
    return HS_OK;
}

// The result as a one-block BlockFile (tasks.py:400-410, io.py:47-109).  A key that is the build-side column arrives as
// code bytes: decoded through the stage's dictionary here.
extern "C" int hs_join_result_write_blockfile(const hs_join_stage* s, const char* path) {
    if (!s || !s->ready || !path) {
        hs_set_error("hs_join_result_write_blockfile: bad arguments");
        return HS_E_ARG;
    }
    if (s->last_rows == 0) return HS_OK;  // empty result: the reference writes no file (tasks.py:405)
    FILE* f = fopen(path, "wb");
    if (!f) {
        hs_set_error("hs_join_result_write_blockfile: cannot create %s", path);
        return HS_E_ARG;
    }
    const int n_out = s->fin.n_out;
    const uint8_t nc = (uint8_t)n_out;
    fwrite(&nc, 1, 1, f);
    for (int o = 0; o < n_out; ++o) {
        const uint8_t type = (uint8_t)s->plan.out_types[o];
        const uint8_t len = (uint8_t)strnlen(s->plan.out_names[o], sizeof(s->plan.out_names[o]));
        fwrite(&type, 1, 1, f);
        fwrite(&len, 1, 1, f);
        fwrite(s->plan.out_names[o], 1, len, f);
    }
    const uint64_t block_start = (uint64_t)ftell(f);
    const uint32_t rows = (uint32_t)s->last_rows;
    fwrite(&rows, 4, 1, f);
    bool ok = true;
    for (int o = 0; o < n_out; ++o) {
        const hs_finish_out& d = s->fin.outs[o];
        const uint8_t* col = (const uint8_t*)s->image_host + d.offset;
        if (d.src == 0 && s->key_is_payload) {  // code bytes -> the strings they stand for
            uint64_t bytes = rows;
            for (uint32_t r = 0; r < rows; ++r) {
                if (col[r] >= s->dict.size()) ok = false;
                else bytes += s->dict[col[r]].size();
            }
            fwrite(&bytes, 8, 1, f);
            for (uint32_t r = 0; ok && r < rows; ++r) {
                const uint8_t len = (uint8_t)s->dict[col[r]].size();
                fwrite(&len, 1, 1, f);
            }
            for (uint32_t r = 0; ok && r < rows; ++r) fwrite(s->dict[col[r]].data(), 1, s->dict[col[r]].size(), f);
            continue;
        }
        const bool is_key_string = d.src == 0 && s->key_kind == HS_STR;
        const int width = d.src == 0 ? s->key_bytes : (d.kind == HS_I64 ? 8 : 4);
        const uint64_t bytes = (uint64_t)rows * (uint64_t)width + (is_key_string ? rows : 0);
        fwrite(&bytes, 8, 1, f);
        if (is_key_string) {
            const uint8_t w = (uint8_t)width;
            for (uint32_t r = 0; r < rows; ++r) fwrite(&w, 1, 1, f);
        }
        fwrite(col, 1, (size_t)rows * (size_t)width, f);
    }
    fwrite(&block_start, 8, 1, f);
    const uint32_t nblocks = 1;
    fwrite(&nblocks, 4, 1, f);
    ok = (fclose(f) == 0) && ok;
    if (!ok) {
        hs_set_error("hs_join_result_write_blockfile: write to %s failed (or a key code outside the dictionary)", path);
        return HS_E_ARG;
    }
    return HS_OK;
}

// =====================================================================================================================
// Round 3: the SELECT / WHERE stage behind the same boundary - a ScanJob whose rows go to the result file
// (jobs.py:45-60; FilterTask tasks.py:167-177, ProjectTask tasks.py:32-35, WriteToLocalFileTask tasks.py:391-410): native
// reader -> predicate (hs_eval) -> stable compaction (hs_compact) -> gathers of the passed-through columns / evaluation of
// the computed ones over the surviving rows (hs_eval with a row list) -> rounding to the stored kinds (hs_quantise) ->
// one device->host copy per column -> BlockFile blocks of ROWS_PER_BLOCK rows, appended in table order.
// =====================================================================================================================
struct hs_select_stage {
    hs_engine* engine = nullptr;
    hs_table* table = nullptr;
    hs_select_stage_plan plan{};
    hs_col cols[HS_MAX_COLS]{}, pcols[HS_MAX_COLS]{};
    // last result, host side: per output column the stored values (strings: lens + payload)
    struct Out {
        std::vector<uint8_t> data, lens;
        int width = 0;
    };
    std::vector<Out> outs;
    int64_t last_rows = 0;
    uint32_t last_flags = 0;
};

extern "C" int hs_select_stage_prepare(hs_engine* e, hs_table* t, const hs_select_stage_plan* plan, size_t plan_bytes,
                                       hs_select_stage** out) {
    if (!e || !t || !plan || !out || plan_bytes != sizeof(hs_select_stage_plan) || plan->version != HS_SELECT_STAGE_PLAN_VERSION ||
        plan->n_cols < 0 || plan->n_cols > HS_MAX_COLS || plan->n_pcols < 0 || plan->n_pcols > HS_MAX_COLS || plan->n_out < 1 ||
        plan->n_out > HS_FINISH_MAX_OUT) {
        hs_set_error("hs_select_stage_prepare: bad plan blob (size %zu, expected %zu)", plan_bytes, sizeof(hs_select_stage_plan));
        return HS_E_ARG;
    }
    if (hipSetDevice(e->device) != hipSuccess) return HS_E_LAUNCH;
    std::vector<int32_t> need(plan->col_ids, plan->col_ids + plan->n_cols);
    need.insert(need.end(), plan->pcol_ids, plan->pcol_ids + plan->n_pcols);
    for (int o = 0; o < plan->n_out; ++o) {
        const int src = plan->out_src[o];
        if (src >= (int)t->cols.size() || (src < 0 && (-1 - src) >= HS_MAX_OUTS)) {
            hs_set_error("hs_select_stage_prepare: output %d is malformed", o);
            return HS_E_ARG;
        }
        if (src >= 0) need.push_back(src);
    }
    for (int32_t c : need) {
        if (c < 0 || c >= (int)t->cols.size()) {
            hs_set_error("hs_select_stage_prepare: no such column %d", c);
            return HS_E_ARG;
        }
    }
    const int rc = hs_table_load(e, t, need.data(), (int32_t)need.size());
    if (rc) return rc;
    hs_select_stage* s = new hs_select_stage();
    s->engine = e;
    s->table = t;
    s->plan = *plan;
    for (int i = 0; i < plan->n_cols; ++i) s->cols[i] = t->cols[plan->col_ids[i]].col;
    for (int i = 0; i < plan->n_pcols; ++i) s->pcols[i] = t->cols[plan->pcol_ids[i]].col;
    *out = s;
    return HS_OK;
}

extern "C" void hs_select_stage_destroy(hs_select_stage* s) { delete s; }

extern "C" int hs_select_stage_run(hs_select_stage* s, void* stream_, uint32_t* flags_out, int64_t* n_rows_out) {
    if (!s) {
        hs_set_error("hs_select_stage_run: null stage");
        return HS_E_ARG;
    }
    hipStream_t stream = (hipStream_t)stream_;
    const hs_select_stage_plan& P = s->plan;
    hs_table* t = s->table;
    const int64_t n = t->nrows;
    uint32_t* flags = (uint32_t*)s->engine->flags.p;
    if (hipMemsetAsync(flags, 0, 4, stream) != hipSuccess) return HS_E_LAUNCH;
    int rc = HS_OK;
    // WHERE: mask -> ascending list of the surviving rows
    DevBuf mask, sel, count, scan_ws;
    int64_t kept = n;
    const bool filtered = P.filter.n_ins > 0;
    if (filtered && n > 0) {
        if (!mask.alloc((size_t)n) || !sel.alloc((size_t)n * 8) || !count.alloc(8) || !scan_ws.alloc(hs_scan_ws_bytes(n))) {
            hs_set_error("hs_select_stage_run: out of device memory");
            return HS_E_LAUNCH;
        }
        void* outs[1] = {mask.p};
        const int32_t kinds[1] = {HS_U8};
        rc = hs_eval(stream, s->cols, P.n_cols, &P.filter, nullptr, n, nullptr, outs, kinds, 1, flags);
        if (!rc) rc = hs_compact(stream, (const uint8_t*)mask.p, n, (int64_t*)sel.p, (int64_t*)count.p, scan_ws.p);
        if (rc) return rc;
        if (hipMemcpyAsync(&kept, count.p, 8, hipMemcpyDeviceToHost, stream) != hipSuccess || hipStreamSynchronize(stream) != hipSuccess)
            return HS_E_LAUNCH;
    }
    const int64_t* rows = filtered && n > 0 ? (const int64_t*)sel.p : nullptr;
    // computed columns over the surviving rows (in-flight f64 / i64), then rounded to what the file stores
    DevBuf computed[HS_MAX_OUTS], stored[HS_MAX_OUTS];
    int n_prog_out = 0;
    for (int o = 0; o < P.n_out; ++o)
        if (P.out_src[o] < 0 && -1 - P.out_src[o] + 1 > n_prog_out) n_prog_out = -1 - P.out_src[o] + 1;
    if (n_prog_out > 0 && kept > 0) {
        void* outs[HS_MAX_OUTS] = {};
        int32_t kinds[HS_MAX_OUTS] = {};
        for (int k = 0; k < n_prog_out; ++k) {
            if (!computed[k].alloc((size_t)kept * 8)) return HS_E_LAUNCH;
            outs[k] = computed[k].p;
            kinds[k] = P.project_kinds[k];
        }
        rc = hs_eval(stream, s->pcols, P.n_pcols, &P.project, rows, kept, nullptr, outs, kinds, n_prog_out, flags);
        for (int k = 0; !rc && k < n_prog_out; ++k) {
            if (!stored[k].alloc((size_t)kept * 4)) return HS_E_LAUNCH;
            rc = hs_quantise(stream, computed[k].p, kinds[k], kept, nullptr, stored[k].p, flags);
        }
        if (rc) return rc;
    }
    // every output column -> host
    s->outs.assign((size_t)P.n_out, hs_select_stage::Out());
    for (int o = 0; o < P.n_out && kept > 0; ++o) {
        hs_select_stage::Out& out = s->outs[(size_t)o];
        const int src = P.out_src[o];
        if (src < 0) {
            out.width = 4;
            out.data.resize((size_t)kept * 4);
            if (hipMemcpyAsync(out.data.data(), stored[-1 - src].p, out.data.size(), hipMemcpyDeviceToHost, stream) != hipSuccess) return HS_E_LAUNCH;
            continue;
        }
        const hs_col& c = t->cols[src].col;
        if (c.kind != HS_STR) {
            const int w = elem_bytes(c.kind);
            out.width = w;
            out.data.resize((size_t)kept * (size_t)w);
            if (!rows) {
                if (hipMemcpyAsync(out.data.data(), c.data, out.data.size(), hipMemcpyDeviceToHost, stream) != hipSuccess) return HS_E_LAUNCH;
            } else {
                DevBuf g;
                if (!g.alloc(out.data.size())) return HS_E_LAUNCH;
                rc = hs_gather_fixed(stream, c.data, w, n, rows, kept, nullptr, g.p, flags);
                if (rc) return rc;
                if (hipMemcpyAsync(out.data.data(), g.p, out.data.size(), hipMemcpyDeviceToHost, stream) != hipSuccess ||
                    hipStreamSynchronize(stream) != hipSuccess)
                    return HS_E_LAUNCH;
            }
            continue;
        }
        // STRING: lengths, offsets, bytes of the surviving rows
        DevBuf lens, offs, mm, ws, data;
        if (!lens.alloc((size_t)kept) || !offs.alloc((size_t)(kept + 1) * 8) || !mm.alloc(8) || !ws.alloc(hs_scan_ws_bytes(kept))) return HS_E_LAUNCH;
        std::vector<int64_t> iota;
        DevBuf all_rows;
        const int64_t* idx = rows;
        if (!idx) {  // no WHERE: the gathers still want a row list
            iota.resize((size_t)kept);
            for (int64_t i = 0; i < kept; ++i) iota[(size_t)i] = i;
            if (!all_rows.alloc((size_t)kept * 8) || hipMemcpy(all_rows.p, iota.data(), (size_t)kept * 8, hipMemcpyHostToDevice) != hipSuccess)
                return HS_E_LAUNCH;
            idx = (const int64_t*)all_rows.p;
        }
        rc = hs_gather_str_lens(stream, &c, n, idx, kept, (uint8_t*)lens.p, flags);
        if (!rc) rc = hs_str_offsets(stream, (const uint8_t*)lens.p, kept, (int64_t*)offs.p, (int32_t*)mm.p, ws.p);
        if (rc) return rc;
        int64_t total = 0;
        if (hipMemcpyAsync(&total, (const int64_t*)offs.p + kept, 8, hipMemcpyDeviceToHost, stream) != hipSuccess ||
            hipStreamSynchronize(stream) != hipSuccess)
            return HS_E_LAUNCH;
        if (!data.alloc((size_t)(total > 0 ? total : 1))) return HS_E_LAUNCH;
        rc = hs_gather_str_bytes(stream, &c, n, idx, kept, (const int64_t*)offs.p, (uint8_t*)data.p);
        if (rc) return rc;
        out.width = -1;
        out.lens.resize((size_t)kept);
        out.data.resize((size_t)total);
        if (hipMemcpyAsync(out.lens.data(), lens.p, (size_t)kept, hipMemcpyDeviceToHost, stream) != hipSuccess ||
            (total > 0 && hipMemcpyAsync(out.data.data(), data.p, (size_t)total, hipMemcpyDeviceToHost, stream) != hipSuccess) ||
            hipStreamSynchronize(stream) != hipSuccess)
            return HS_E_LAUNCH;
    }
    uint32_t f = 0;
    if (hipMemcpyAsync(&f, flags, 4, hipMemcpyDeviceToHost, stream) != hipSuccess || hipStreamSynchronize(stream) != hipSuccess) return HS_E_LAUNCH;
    s->last_flags = f;
    s->last_rows = kept;
    if (flags_out) *flags_out = f;
    if (n_rows_out) *n_rows_out = kept;
    return HS_OK;
}

// The rows of the last run as a BlockFile of rows_per_block-row blocks (reference tasks.py:391-410 + io.py:217-252: a
// result larger than a block continues in further blocks; an empty result writes no file).
extern "C" int hs_select_result_write_blockfile(const hs_select_stage* s, const char* path, int64_t rows_per_block) {
    if (!s || !path || rows_per_block < 1) {
        hs_set_error("hs_select_result_write_blockfile: bad arguments");
        return HS_E_ARG;
    }
    if (s->last_rows == 0) return HS_OK;
    FILE* f = fopen(path, "wb");
    if (!f) {
        hs_set_error("hs_select_result_write_blockfile: cannot create %s", path);
        return HS_E_ARG;
    }
    const hs_select_stage_plan& P = s->plan;
    const uint8_t nc = (uint8_t)P.n_out;
    fwrite(&nc, 1, 1, f);
    for (int o = 0; o < P.n_out; ++o) {
        const uint8_t type = (uint8_t)P.out_types[o];
        const uint8_t len = (uint8_t)strnlen(P.out_names[o], sizeof(P.out_names[o]));
        fwrite(&type, 1, 1, f);
        fwrite(&len, 1, 1, f);
        fwrite(P.out_names[o], 1, len, f);
    }
    std::vector<uint64_t> starts;
    std::vector<int64_t> str_pos((size_t)P.n_out, 0);  // byte position inside a string column's payload
    for (int64_t lo = 0; lo < s->last_rows; lo += rows_per_block) {
        const int64_t hi = lo + rows_per_block < s->last_rows ? lo + rows_per_block : s->last_rows;
        const uint32_t rows = (uint32_t)(hi - lo);
        starts.push_back((uint64_t)ftell(f));
        fwrite(&rows, 4, 1, f);
        for (int o = 0; o < P.n_out; ++o) {
            const hs_select_stage::Out& out = s->outs[(size_t)o];
            if (out.width > 0) {
                const uint64_t bytes = (uint64_t)rows * (uint64_t)out.width;
                fwrite(&bytes, 8, 1, f);
                fwrite(out.data.data() + (size_t)lo * (size_t)out.width, 1, (size_t)bytes, f);
            } else {
                uint64_t payload = 0;
                for (int64_t r = lo; r < hi; ++r) payload += out.lens[(size_t)r];
                const uint64_t bytes = rows + payload;
                fwrite(&bytes, 8, 1, f);
                fwrite(out.lens.data() + lo, 1, rows, f);
                fwrite(out.data.data() + str_pos[(size_t)o], 1, (size_t)payload, f);
                str_pos[(size_t)o] += (int64_t)payload;
            }
        }
    }
    fwrite(starts.data(), 8, starts.size(), f);
    const uint32_t nblocks = (uint32_t)starts.size();
    fwrite(&nblocks, 4, 1, f);
    if (fclose(f) != 0) {
        hs_set_error("hs_select_result_write_blockfile: write to %s failed", path);
        return HS_E_ARG;
    }
    return HS_OK;
}
