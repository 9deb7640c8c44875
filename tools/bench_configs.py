"""BASELINE.json configs 4 and 5 as bench.py workloads (``bench.py --config join|strkey``), plus a stand-alone runner:

    python tools/bench_configs.py [sf]        both configs once on one GPU, checked against the C oracle ports

Config 4: orders JOIN lineitem ON l_orderkey = o_orderkey GROUP BY o_orderpriority (sf=10: 14 996 513 x 59 986 052).
Config 5: lineitem WHERE l_shipmode LIKE '%AIR%' GROUP BY l_returnflag + '-' + l_shipmode.
Tables are synthetic (counter-based generators with CPU twins in oracle/), resident in HBM before the timed region;
the checkers are oracle/q45_oracle.c (pinned to the reference's goldens by tests/test_oracle_golden.py)."""

from __future__ import annotations

import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

HBM_PEAK_GBPS = 8000.0


def _with_traffic(out: dict, config: str, world: int, sf: float, kernel_ms: float) -> dict:
    """`traffic` (PMC bytes per launch of the dominant kernel, stamped with the kernel sources' hash like Q1's) and the HBM
    utilisation on those bytes - only for the configuration the passes were collected on (one GPU, sf=10)."""
    from bench import pmc_traffic  # noqa: PLC0415

    if world == 1 and sf == 10.0:
        out.update(pmc_traffic(config))
        if out.get("traffic"):
            out["hbm_frac"] = out["traffic"] / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS
    return out


class _EventPair:
    """HIP events on the launch stream around one operator call (the library launches on torch's current stream)."""

    def __init__(self) -> None:
        import torch

        self.begin, self.end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        self.begin.record()
        self.end.record()

    def ms(self) -> float:
        return self.begin.elapsed_time(self.end)


class _Workload:
    metric = ""
    dominant = ""

    def __init__(self, engine, scratch: Path, args, rank: int, world: int) -> None:
        from minispark_amd import synth

        self.engine, self.scratch, self.args, self.rank, self.world = engine, scratch, args, rank, world
        self.n_li = synth.lineitem_rows(args.sf)
        self.total_units = self.n_li
        self.timer = None

    def exchange_ms(self) -> float:
        return self.engine.dev.exchange_ms() if self.engine.dist is not None else 0.0

    def exchange_text(self, backend: str) -> str:
        return "none"

    def extra_split_ms(self) -> dict:
        return {}

    def dominant_kernel_ms(self) -> float:
        return self.engine.dev.scan_kernel_ms()

    def _time_dominant(self, method_name: str) -> None:
        """Bracket every call of Device.<method_name> with an event pair (read after the step's host sync)."""
        dev = self.engine.dev
        inner = getattr(dev, method_name)
        self.timer = _EventPair()
        timer = self.timer

        def timed(*a, **k):
            timer.begin.record()
            out = inner(*a, **k)
            timer.end.record()
            return out

        setattr(dev, method_name, timed)


class JoinWorkload(_Workload):
    """Config 4 on 1..N GPUs.  Every rank holds its blocks (b % N) of BOTH tables; the build side (orders: key + priority
    code) is all-gathered, every rank builds the whole byte table and probes its own lineitem blocks inside the aggregate
    scan; the units' raw tables meet through one small all-gather.  Byte accounting: orders
    (4 + 1 + len(priority)) + lineitem (4 + 4 + 4 referenced payload), read once, + one random access per probe row."""

    metric = "orders JOIN lineitem GROUP BY o_orderpriority (BASELINE config 4): lineitem probe rows/sec"

    def __init__(self, engine, scratch, args, rank, world) -> None:
        super().__init__(engine, scratch, args, rank, world)
        from minispark_amd import synth, workloads

        self.n_ord = synth.orders_rows(self.n_li)
        self.li_path, self.ord_path = scratch / "lineitem.bin", scratch / "orders.bin"
        self.li = synth.make_lineitem(engine.dev, self.li_path, self.n_li, with_orderkey=True, rank=rank, world=world)
        engine.attach_device_table(self.li_path, self.li)
        orders = synth.make_orders(engine.dev, self.ord_path, self.n_ord, rank=rank, world=world)
        engine.attach_device_table(self.ord_path, orders)
        self.prio_bytes = sum(len(p) for p in workloads.PRIORITIES) / len(workloads.PRIORITIES)
        self.frame = workloads.join_group(workloads.engine_api(engine), str(self.ord_path), str(self.li_path))
        engine.dev.time_join(True)  # events around the table build, part of the recorded run

    def fused(self) -> bool:
        return (getattr(self.engine.dev, "last_join", None) or {}).get("mode") == "byte table"

    def dominant_kernel_ms(self) -> float:
        # fused probe: the aggregate scan with the probe inside (the library's event pair around it); the materialising
        # form of round 2: table build + probe kernels (events around the operator)
        return self.engine.dev.scan_kernel_ms() if self.fused() else self.engine.dev.join_ms()

    def extra_split_ms(self) -> dict:
        return {"join_build": self.engine.dev.join_ms()} if self.fused() else {}

    def exchange_text(self, backend: str) -> str:
        return (f"all_gather of the build side's key + code columns, one all_gather of the raw unit tables per query ({backend}); "
                "probe rows never leave their rank")

    def algorithmic_bytes_per_launch(self) -> float:
        return self.n_ord * (4 + 1 + self.prio_bytes) + self.n_li * (4 + 4 + 4) + self.n_li * 64.0

    def config(self, rows) -> dict:
        return {"workload": f"SELECT o_orderpriority, COUNT(), SUM(l_quantity), SUM(l_extendedprice), MAX(l_extendedprice) "
                            f"FROM orders JOIN lineitem ON o_orderkey = l_orderkey GROUP BY o_orderpriority, synthetic sf={self.args.sf:g}",
                "rows": self.n_li, "orders": self.n_ord, "groups": len(rows or []),
                "join": getattr(self.engine.dev, "last_join", None), "fused_joins": self.engine.fused_joins,
                "probe_inside_the_aggregate": self.engine.fused_probes > 0}

    def roofline(self, kernel_avg_ms: float) -> dict:
        last = self.engine.dev.last_join or {}
        if self.fused():
            # The scan reads the rank's probe-side columns once (key + the two aggregated f32 columns) and one table byte
            # per probe key; lineitem is clustered on the order key, so the table is in effect streamed: its bytes once.
            n_local = self.li.nrows
            table_bytes = last.get("slots", 4 * self.n_ord)
            algo = n_local * 12 + table_bytes
            achieved = algo / (kernel_avg_ms * 1e-3) / 1e9
            build_ms = self.engine.dev.join_ms()
            build_algo = self.n_ord * 5 + self.n_ord * 4 * 2 + table_bytes  # keys + codes read, tuples written + read, table written
            out = {"bound": "hbm", "kernel": "k_agg_shared_jit with the join's probe inside (byte table lookup per key, unit = "
                                              "python_hash(key) % 10, LDS dictionary of (unit, code) cells)",
                    "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS, "traffic": None,
                    "kernel_ms": kernel_avg_ms, "rows_per_launch": n_local, "algorithmic_bytes_per_launch": algo,
                    "launch": self.engine.dev.last_scan,
                    "build": {"kernels": "k_join8_hist + scans + k_join8_scatter + k_join8_fill", "ms": build_ms,
                              "algorithmic_bytes": build_algo, "GBps": build_algo / (build_ms * 1e-3) / 1e9 if build_ms > 0 else None},
                    "sector_accounting_GBps": (n_local * (12 + 64.0)) / (kernel_avg_ms * 1e-3) / 1e9,
                    "accounting": "achieved = 12 B per local lineitem row (key + quantity + extendedprice) + the byte table once "
                                  "(probe keys arrive clustered: the table is streamed, not sampled); SURVEY 8d's own accounting "
                                  "(64 B per probe as a random access) is the sector_accounting figure; build: keys + codes in, "
                                  "4-byte tuples out and in, table out"}
            return _with_traffic(out, "join", self.world, self.args.sf, kernel_avg_ms)
        # round 2's materialising form (table fill + scatter + occupied-slot count + probe), priced on the bytes it must
        # move when every access were perfectly coalesced
        slots = last.get("slots", 4 * self.n_ord)
        algo = self.n_ord * (5 + 4) + slots * 8 + self.n_li * (4 + 4 + 2)
        achieved = algo / (kernel_avg_ms * 1e-3) / 1e9
        sector = self.n_ord * (5 + 64.0) + slots * 8 + self.n_li * (4 + 64.0 + 2)
        return {"bound": "hbm", "kernel": "in-place join: k_fill_u32 + k_join_scatter_direct + k_join_count_occupied + k_join_probe_unique",
                "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS, "traffic": None,
                "kernel_ms": kernel_avg_ms, "rows_per_launch": self.n_li, "algorithmic_bytes_per_launch": algo,
                "sector_accounting_GBps": sector / (kernel_avg_ms * 1e-3) / 1e9,
                "accounting": "achieved = coalesced-minimum bytes (5 B per order + 4 B table word per order and per probe + 8 B per "
                              "table slot for fill and count + 4 B key and 2 B out per lineitem)"}

    def _host_columns(self):
        from minispark_amd import synth
        from oracle import q1_native, q45_native

        cols = q1_native.gen(synth.SEED, 0, self.n_li, orderkey=True)
        okey, ocode = q45_native.gen_orders(synth.SEED, 0, self.n_ord, self.n_ord)
        return cols, okey, ocode

    def full_check(self, rows) -> dict:
        from minispark_amd import workloads
        from oracle import q1_native, q45_native
        from oracle.compare import assert_rows_match

        t0 = time.perf_counter()
        cols, okey, ocode = self._host_columns()
        threads = min(q1_native.host_threads(), 10)
        want = q45_native.run_join_group(okey, ocode, workloads.PRIORITIES, cols["l_orderkey"], cols["l_quantity"],
                                         cols["l_extendedprice"], threads=threads)
        dt = time.perf_counter() - t0
        try:
            flips, ok, why = assert_rows_match(rows, want, max_ulps=1), True, None
        except AssertionError as e:
            flips, ok, why = None, False, str(e)[:400]
        out = {"gpu_matches_oracle_full": ok, "f32_ulp_flips_full": flips, "rows_checked": self.n_li,
               "oracle": f"oracle/q45_oracle.c q4_run on {threads} host threads, {dt:.1f} s (table generation included)"}
        if why:
            out["mismatch"] = why
        return out

    def cpu_baseline(self) -> dict:
        from minispark_amd import workloads
        from oracle import q1_native, q45_native

        cols, okey, ocode = self._host_columns()
        threads = min(q1_native.host_threads(), 10)  # one JoinJob per shuffle partition
        t0 = time.perf_counter()
        q45_native.run_join_group(okey, ocode, workloads.PRIORITIES, cols["l_orderkey"], cols["l_quantity"],
                                  cols["l_extendedprice"], threads=threads)
        dt = time.perf_counter() - t0
        return {"value": self.n_li / dt, "unit": "rows/s", "cores": threads, "kind": "port",
                "sample": f"the whole sf={self.args.sf:g} tables, one pass ({dt:.1f} s) of oracle/q45_oracle.c q4_run, "
                          f"{threads} threads (one per JoinJob)"}


class StrKeyWorkload(_Workload):
    """Config 5.  Byte accounting: 4 (qty) + 4 (discount) + (1 + len) for l_returnflag and l_shipmode."""

    metric = "LIKE + CONCAT-key GROUP BY on lineitem (BASELINE config 5): lineitem rows/sec"

    def __init__(self, engine, scratch, args, rank, world) -> None:
        super().__init__(engine, scratch, args, rank, world)
        from minispark_amd import synth, workloads

        self.li_path = scratch / "lineitem.bin"
        self.table = synth.make_lineitem(engine.dev, self.li_path, self.n_li, with_shipmode=True, rank=rank, world=world)
        engine.attach_device_table(self.li_path, self.table)
        self.mode_bytes = sum(len(m) for m in workloads.SHIPMODES) / len(workloads.SHIPMODES)
        self.frame = workloads.strkey_like(workloads.engine_api(engine), str(self.li_path))
        # dominant kernel = the fused scan (WHERE on the code column + GROUP BY the coded key): the library's own event
        # pair around it (Device.scan_kernel_ms), like Q1

    def exchange_text(self, backend: str) -> str:
        return f"one all_gather of partial-row slabs per query ({backend}); string columns coded in one dictionary agreed by all ranks"

    def algorithmic_bytes_per_launch(self) -> float:
        return self.n_li * (4 + 4 + 2 + 1 + self.mode_bytes)

    def config(self, rows) -> dict:
        coded = {self.table.schema[c][0]: (col.dict is not None) for c, col in self.table.columns.items()
                 if self.table.schema[c][1].name == "STRING"}
        return {"workload": f"SELECT k, SUM(l_quantity), AVG(l_discount), COUNT() FROM (SELECT l_returnflag + '-' + l_shipmode AS k, "
                            f"... FROM lineitem WHERE l_shipmode LIKE '%AIR%') GROUP BY k, synthetic sf={self.args.sf:g}",
                "rows": self.n_li, "groups": len(rows or []), "dictionary_coded": coded}

    def roofline(self, kernel_avg_ms: float) -> dict:
        n_local = self.table.nrows
        algo = n_local * (4 + 4 + 2 + 1 + self.mode_bytes)
        achieved = algo / (kernel_avg_ms * 1e-3) / 1e9
        moved = n_local * (4 + 4 + 1 + 1)
        out = {"bound": "hbm", "kernel": "k_agg_jit: fused scan + WHERE (bit test on the l_shipmode code) + GROUP BY the coded "
                                          "CONCAT key + partial aggregate (after one pass over code bytes that builds the key)",
                # `frac` prices SURVEY 8d's ALGORITHMIC bytes (strings as stored); the kernel moves the dictionary-coded form:
                # `hbm_frac_moved` is the utilisation of the bytes that really leave HBM
                "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS, "traffic": None,
                "hbm_frac_moved": moved / (kernel_avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                "kernel_ms": kernel_avg_ms, "rows_per_launch": n_local,
                "algorithmic_bytes_per_row": algo / max(n_local, 1), "moved_bytes_per_row": moved / max(n_local, 1),
                "moved_GBps": moved / (kernel_avg_ms * 1e-3) / 1e9, "launch": self.engine.dev.last_scan,
                "accounting": "algorithmic (SURVEY 8d) = 4 (qty) + 4 (discount) + (1 + len) of l_returnflag and l_shipmode; moved = "
                              "the two f32 columns + one key-code byte + one shipmode-code byte (both string columns are "
                              "dictionary-coded at table open); LIKE is evaluated once per dictionary entry (<= 256), never per "
                              "row, in the timed step"}
        return _with_traffic(out, "strkey", self.world, self.args.sf, kernel_avg_ms)

    def _host_columns(self):
        from minispark_amd import synth
        from oracle import q1_native

        return q1_native.gen(synth.SEED, 0, self.n_li, shipmode=True)

    def _oracle(self, cols, threads: int):
        from minispark_amd import synth, workloads
        from oracle import q45_native

        return q45_native.run_strkey_like(cols["l_returnflag"], cols["l_shipmode_code"], workloads.SHIPMODES, cols["l_quantity"],
                                          cols["l_discount"], synth.block_sizes(self.n_li), threads=threads)

    def full_check(self, rows) -> dict:
        from oracle import q1_native
        from oracle.compare import assert_rows_match

        t0 = time.perf_counter()
        threads = q1_native.host_threads()
        want = self._oracle(self._host_columns(), threads)
        dt = time.perf_counter() - t0
        try:
            flips, ok, why = assert_rows_match(rows, want, max_ulps=1), True, None
        except AssertionError as e:
            flips, ok, why = None, False, str(e)[:400]
        out = {"gpu_matches_oracle_full": ok, "f32_ulp_flips_full": flips, "rows_checked": self.n_li,
               "oracle": f"oracle/q45_oracle.c q5_run on {threads} host threads, {dt:.1f} s (table generation included)"}
        if why:
            out["mismatch"] = why
        return out

    def cpu_baseline(self) -> dict:
        from oracle import q1_native

        cols = self._host_columns()
        threads = q1_native.host_threads()
        t0 = time.perf_counter()
        self._oracle(cols, threads)
        dt = time.perf_counter() - t0
        return {"value": self.n_li / dt, "unit": "rows/s", "cores": threads, "kind": "port",
                "sample": f"the whole sf={self.args.sf:g} table, one pass ({dt:.2f} s) of oracle/q45_oracle.c q5_run, {threads} threads"}


def main() -> None:
    import tempfile
    from types import SimpleNamespace

    os.environ.setdefault("TZ", "UTC")
    time.tzset()
    import torch

    from minispark_amd import constants
    from minispark_amd.execution import HipExecutionEngine

    sf = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
    only = sys.argv[2] if len(sys.argv) > 2 else None
    scratch = Path(tempfile.mkdtemp(prefix="hs_cfg_", dir="/dev/shm"))
    constants.SHUFFLE_FOLDER = scratch / "shuffle"
    for name, cls in (("join", JoinWorkload), ("strkey", StrKeyWorkload)):
        if only and name != only:
            continue
        engine = HipExecutionEngine(0)
        wl = cls(engine, scratch / name, SimpleNamespace(sf=sf, config=name), 0, 1)
        engine.dev.time_scan_kernel(True)
        rows = None
        times = []
        for _ in range(8):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            rows = wl.frame.collect()
            times.append(time.perf_counter() - t0)
        if os.environ.get("HS_CPROFILE"):
            import cProfile
            import pstats

            pr = cProfile.Profile()
            pr.enable()
            for _ in range(20):
                wl.frame.collect()
            pr.disable()
            stats = pstats.Stats(pr)
            top = sorted(((ct / 20 * 1e6, tt / 20 * 1e6, nc / 20, f"{Path(fn).name}:{line}({fname})")
                          for (fn, line, fname), (cc, nc, tt, ct, _) in stats.stats.items()), reverse=True)[:25]
            for ct, tt, nc, what in top:
                print(f"{ct:10.1f} us cum {tt:10.1f} us own {nc:6.1f} calls  {what}")
        check = wl.full_check(rows)
        print(f"{name}: sf={sf:g}  best {min(times) * 1e3:.2f} ms  runs {[round(t * 1e3, 2) for t in times]}  "
              f"dominant {wl.dominant_kernel_ms():.3f} ms  {wl.config(rows)}\n   check {check}", flush=True)
        engine.__exit__(None, None, None)


if __name__ == "__main__":
    main()
