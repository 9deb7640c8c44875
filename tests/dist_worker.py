"""Worker for the multi-process tests: one rank of a world running a catalogue query through
HipExecutionEngine.enable_distributed (backend gloo = rehearsal on one GPU / CPU plumbing; nccl = RCCL)."""

from __future__ import annotations

import json
import os
import sys
import time
from pathlib import Path

os.environ["TZ"] = "UTC"
time.tzset()
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


def main() -> None:
    """`dist_worker.py <case> <out> <backend>` - or `batch:<jobs.json>` as the case: a list of [case, out] pairs run one
    after the other inside ONE process group (one interpreter + torch import + rendezvous per world instead of per case:
    the spawns were half of the GPU suite's wall time).  A job that raises ends the batch: its error is written where its
    rows would have gone, the jobs behind it are not run (a rank that went on alone would hang in the next collective)."""
    case_name, out_path, backend = sys.argv[1], sys.argv[2], sys.argv[3]
    if os.environ.get("HIPSPARK_WORKER_DUMP_AFTER"):  # a hung collective: every thread's stack after N seconds, then exit
        import faulthandler

        faulthandler.dump_traceback_later(int(os.environ["HIPSPARK_WORKER_DUMP_AFTER"]), exit=True)
    import torch
    import torch.distributed as dist

    rank = int(os.environ["RANK"])
    world = int(os.environ["WORLD_SIZE"])
    if backend == "nccl":  # RCCL: one rank per device, communicator bound to it eagerly
        local = int(os.environ.get("LOCAL_RANK", "0"))
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
    else:
        dist.init_process_group(backend, rank=rank, world_size=world)
    jobs = json.loads(Path(case_name[len("batch:"):]).read_text()) if case_name.startswith("batch:") else [[case_name, out_path]]
    failed = None
    for job_case, job_out in jobs:
        if failed is not None:
            if rank == 0:
                Path(job_out).write_text(json.dumps({"error": f"not run: the batch ended at {failed}"}))
            continue
        try:
            run_case(job_case, job_out, dist, rank, world)
        except BaseException:  # noqa: BLE001
            import traceback

            failed = job_case
            if len(jobs) == 1:
                raise
            print(f"[rank {rank}] job {job_case} failed:\n{traceback.format_exc()}", file=sys.stderr, flush=True)
            Path(str(job_out) + f".rank{rank}.error").write_text(traceback.format_exc())
            if rank == 0:
                Path(job_out).write_text(json.dumps({"error": traceback.format_exc()[-3000:]}))
    if failed is not None:
        os._exit(3)  # peers may be parked in a collective of the failed job: no barrier, no teardown
    dist.barrier()
    dist.destroy_process_group()


def run_case(case_name: str, out_path: str, dist, rank: int, world: int) -> None:
    from minispark_amd import constants
    from minispark_amd.dataframe import DataFrame
    from minispark_amd.execution import HipExecutionEngine
    from minispark_amd.sql import Col, Functions, Lit
    from tests.conftest import load_golden
    from minispark_amd.workloads import api_namespace
    from tests.queries import case_by_name

    constants.SHUFFLE_FOLDER = Path(out_path).parent / f"shuffle_r{rank}"
    if case_name.startswith("fuzz:"):
        # a random query of tests/test_gpu_fuzz.py over tables the parent test wrote next to out_path
        import random

        from tests.test_gpu_fuzz import random_query

        seed = int(case_name.split(":")[1])
        t1, t2 = Path(out_path).parent / "a.bin", Path(out_path).parent / "b.bin"
        build = lambda api: random_query(random.Random(seed), api, str(t1), str(t2))  # noqa: E731
    elif case_name.startswith("wide:"):
        # a random many-group GROUP BY of tests/test_gpu_shared_tier.py over the table the parent test wrote
        import random

        from tests.test_gpu_shared_tier import _wide_query

        seed = int(case_name.split(":")[1])
        build = lambda api: _wide_query(random.Random(seed), api, str(Path(out_path).parent / "w.bin"))  # noqa: E731
    elif case_name.startswith(("config4:", "config5:")):
        # BASELINE configs 4 / 5 at size (tools/bench_configs.py): every rank generates its blocks (b % world) of the
        # synthetic tables, rank 0 checks the rows against the C port of the reference's algorithm over the WHOLE tables
        from types import SimpleNamespace

        from tools.bench_configs import JoinWorkload, StrKeyWorkload

        sf = float(case_name.split(":")[1])
        with HipExecutionEngine(device=int(os.environ.get("LOCAL_RANK", "0"))) as engine:
            engine.enable_distributed(dist)
            cls, name = (JoinWorkload, "join") if case_name.startswith("config4:") else (StrKeyWorkload, "strkey")
            wl = cls(engine, Path(out_path).parent / f"tables_r{rank}", SimpleNamespace(sf=sf, config=name), rank, world)
            rows = None
            for _ in range(5):
                rows = wl.frame.collect()
            report = {"replays": engine.replays, "fused_probes": engine.fused_probes, "rows": len(rows),
                      "join": getattr(engine.dev, "last_join", None), "sharded_builds": engine.sharded_builds}
            joins: list = [None] * world
            dist.all_gather_object(joins, report["join"])
            report["joins"] = joins
            if rank == 0:
                report["check"] = wl.full_check(rows)
                report["n"] = sum(r.get("n", r.get("count", 0)) for r in rows)
                Path(out_path).write_text(json.dumps(report))
            else:
                assert rows == [], f"rank {rank} must not own result rows"
        return
    elif case_name.startswith("width:"):
        # GROUP BY a string key whose fixed width the ranks see differently (tests/test_gpu_distributed.py)
        from tests.test_gpu_distributed import width_query

        build = lambda api: width_query(api, str(Path(out_path).parent / "w.bin"))  # noqa: E731
    else:
        golden = load_golden(case_name)
        case = case_by_name(case_name)
        build = lambda api: case.build(api, golden["paths"])  # noqa: E731
    with HipExecutionEngine(device=int(os.environ.get("LOCAL_RANK", "0"))) as engine:
        engine.enable_distributed(dist)
        api = api_namespace(lambda: DataFrame(engine), Col, Functions, Lit)
        frame = build(api)
        rows = None
        for _ in range(4):  # later runs exercise the plan / launch caches and the recorded replay
            rows = frame.collect()
        if os.environ.get("HIPSPARK_WORKER_EXPECT_P2P"):
            assert engine.p2p_exchanges >= 1 and engine.replays >= 1, (engine.p2p_exchanges, engine.replays)
        if case_name.startswith("q1"):
            assert engine.replays >= 1, "the recorded replay path must have been exercised"
        if rank == 0:
            enc = [{k: (v.hex() if type(v) is float else (v.isoformat() if hasattr(v, "isoformat") else v))
                    for k, v in r.items()} for r in rows]
            Path(out_path).write_text(json.dumps(enc))
        else:
            assert rows == [], f"rank {rank} must not own result rows"


if __name__ == "__main__":
    main()
