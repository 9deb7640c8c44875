"""N>1 exchange path on CPU (gloo, world_size 2): the slab layout, the all-gather and the un-interleave of
minispark_amd/distributed.py are device-agnostic torch plumbing, so they are exercised here without a GPU;
the kernels that consume the result are covered by the GPU tests (tests/test_gpu_distributed.py)."""

from __future__ import annotations

import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from minispark_amd.distributed import SlabLayout, all_gather_slabs, local_blocks, max_local_units, unpack_gathered


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _make_slab(rank: int, layout: SlabLayout, rows: int) -> torch.Tensor:
    slab = torch.zeros(layout.nbytes, dtype=torch.uint8)
    layout.flags_view(slab)[0] = 0x2 if rank == 1 else 0
    layout.count_view(slab)[0] = rows
    layout.order_view(slab)[:rows] = torch.arange(rows) * 2 + rank  # block b = 2*i + rank
    layout.order_view(slab)[rows:] = 777  # garbage beyond the count: must come out as -1
    layout.column_view(slab, 0)[:rows] = torch.arange(rows, dtype=torch.uint8) + 65 + rank  # 1-byte string key
    layout.column_view(slab, 1)[:rows] = torch.arange(rows, dtype=torch.float32) + 100 * rank
    layout.column_view(slab, 2)[:rows] = torch.arange(rows, dtype=torch.int32) - 5 * rank
    return slab


def _worker(rank: int, world: int, port: int, result):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    layout = SlabLayout.build(6, [(1, torch.uint8), (4, torch.float32), (4, torch.int32)])
    rows = 4 if rank == 0 else 3
    gathered = all_gather_slabs(dist, _make_slab(rank, layout, rows), world)
    flags, order, cols = unpack_gathered(gathered, layout)
    result[rank] = (flags.tolist(), order.tolist(), [c.tolist() for c in cols])
    dist.barrier()
    dist.destroy_process_group()


def test_slab_all_gather_world2():
    world = 2
    manager = mp.Manager()
    result = manager.dict()
    mp.spawn(_worker, args=(world, _free_port(), result), nprocs=world, join=True)
    assert result[0] == result[1], "every rank must see the same gathered partial rows"
    flags, order, cols = result[0]
    assert flags == [0, 2]
    #            rank 0: 4 rows, 2 padding       rank 1: 3 rows, 3 padding
    assert order == [0, 2, 4, 6, -1, -1, 1, 3, 5, -1, -1, -1]
    assert cols[0][:4] == [65, 66, 67, 68] and cols[0][6:9] == [66, 67, 68]
    assert cols[1][:4] == [0.0, 1.0, 2.0, 3.0] and cols[1][6:9] == [100.0, 101.0, 102.0]
    assert cols[2][:4] == [0, 1, 2, 3] and cols[2][6:9] == [-5, -4, -3]


def test_slab_layout_alignment_and_views():
    layout = SlabLayout.build(5, [(2, torch.uint8), (8, torch.int64), (4, torch.float32)])
    assert layout.nbytes % 16 == 0
    assert all(c.offset % 16 == 0 for c in layout.columns)
    slab = torch.zeros(layout.nbytes, dtype=torch.uint8)
    assert layout.order_view(slab).numel() == 5
    assert layout.column_view(slab, 0).numel() == 10
    assert layout.column_view(slab, 1).dtype == torch.int64 and layout.column_view(slab, 1).numel() == 5
    layout.column_view(slab, 2)[:] = 1.5  # views alias the slab
    assert slab[layout.columns[2].offset: layout.columns[2].offset + 4].view(torch.float32)[0] == 1.5


def test_block_ownership():
    assert local_blocks(7, 0, 2) == [0, 2, 4, 6]
    assert local_blocks(7, 1, 2) == [1, 3, 5]
    assert sorted(b for r in range(8) for b in local_blocks(287, r, 8)) == list(range(287))
    assert max_local_units(287, 8) == 36 and max_local_units(6, 2) == 3 and max_local_units(1, 8) == 1


def _matrix_worker(rank: int, world: int, port: int, result):
    from minispark_amd.distributed import all_to_all_rows, exchange_size_matrix

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    # rank r sends to rank d: (10 * r + d) rows and (100 * r + d) payload bytes
    mine = [[10 * rank + d, 100 * rank + d] for d in range(world)]
    theirs = exchange_size_matrix(dist, mine, torch.device("cpu"))
    # the data collective with those split sizes: one byte buffer, destination-major
    send_counts = [m[0] for m in mine]
    recv_counts = [t[0] for t in theirs]
    send = torch.cat([torch.full((c,), 16 * rank + d, dtype=torch.uint8) for d, c in enumerate(send_counts)])
    got = all_to_all_rows(dist, send, send_counts, recv_counts, 1)
    result[rank] = (theirs, got.tolist())
    dist.barrier()
    dist.destroy_process_group()


def test_size_matrix_and_single_data_collective_world3():
    """The generic row exchange's two collectives (execution._exchange_rows): a size matrix, then ONE all-to-all of
    bytes with those split sizes - under gloo on CPU, world 3."""
    world = 3
    manager = mp.Manager()
    result = manager.dict()
    mp.spawn(_matrix_worker, args=(world, _free_port(), result), nprocs=world, join=True)
    for r in range(world):
        theirs, got = result[r]
        assert theirs == [[10 * s + r, 100 * s + r] for s in range(world)]
        want = [v for s in range(world) for v in [16 * s + r] * (10 * s + r)]
        assert got == want


def _flags_worker(rank: int, world: int, port: int, result):
    from minispark_amd.distributed import or_flags

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    cpu = torch.device("cpu")
    mine = [0x300, 0x100 if rank == 0 else 0x0, 0x308 if rank == 1 else 0x1, 0x80000000 if rank == 0 else 0x0, 0]
    result[rank] = [or_flags(dist, f, cpu) for f in mine]
    dist.barrier()
    dist.destroy_process_group()


def test_or_flags_reduces_the_whole_status_word_world2():
    """Round 2 reduced bits 0-7 only: HS_FLAG_MERGE_FULL (0x100) / HS_FLAG_MERGE_ROWS (0x200) - tested by the engine
    AFTER or_flags - were erased on every multi-rank generic-exchange query (even the rank's own bits), so an
    overflowed final merge came back short, silently."""
    from minispark_amd import hipspark as hs

    assert hs.FLAG_MERGE_FULL == 0x100 and hs.FLAG_MERGE_ROWS == 0x200
    world = 2
    manager = mp.Manager()
    result = manager.dict()
    mp.spawn(_flags_worker, args=(world, _free_port(), result), nprocs=world, join=True)
    assert result[0] == result[1] == [0x300, 0x100, 0x309, 0x80000000, 0]


def test_bench_gpus_n_launches_its_own_ranks():
    """`python bench.py --gpus 2` with no WORLD_SIZE in the environment (the shape of the driver's N=1 command) starts
    its ranks itself as a child `torch.distributed.run` before anything touches a GPU, relays rank 0's line and its exit
    code.  HIPSPARK_BENCH_LAUNCH_ONLY stops the ranks right after the rendezvous (no GPU here)."""
    import json
    import subprocess
    import sys
    from pathlib import Path

    root = Path(__file__).resolve().parent.parent
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(HIPSPARK_DIST_BACKEND="gloo", HIPSPARK_BENCH_LAUNCH_ONLY="1")
    run = subprocess.run([sys.executable, str(root / "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                         env=env, capture_output=True, text=True, timeout=300)
    assert run.returncode == 0, run.stderr[-2000:]
    lines = [ln for ln in run.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1 and json.loads(lines[0]) == {"launched_ranks": 2, "world": 2}
    # a failing rank's exit code comes back through the parent
    env["HIPSPARK_DIST_BACKEND"] = "no-such-backend"
    run = subprocess.run([sys.executable, str(root / "bench.py"), "--gpus", "2"], env=env, capture_output=True,
                         text=True, timeout=300)
    assert run.returncode != 0
