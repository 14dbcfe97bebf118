"""CPU, world_size 2, gloo: the multi-GPU plumbing of bench.py (shard arithmetic + max-over-ranks timing).
The data path itself has no collective to test: shards are disjoint stream sets."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from foo_dsp_resampler_amd.sharding import job_throughput, max_over_ranks, shard_range


def test_shard_range_covers_everything():
    for total in (0, 1, 7, 8, 1024, 1025):
        for world in (1, 2, 3, 8):
            got = [shard_range(total, world, r) for r in range(world)]
            assert sum(n for _, n in got) == total
            pos = 0
            for first, n in got:
                assert first == pos
                pos += n
            assert max(n for _, n in got) - min(n for _, n in got) <= 1
    assert shard_range(1024, 8, 3) == (384, 128)      # BASELINE configs[4]: 128 streams per GPU
    with pytest.raises(ValueError):
        shard_range(8, 2, 2)


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    first, n = shard_range(1024, world, rank)
    elapsed = 1.0 + 0.5 * rank                      # rank 1 is the slow one
    mx = max_over_ranks([elapsed, 10.0 * (rank + 1)], dist)
    counts = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]
    dist.all_gather(counts, torch.tensor([n], dtype=torch.int64))
    dist.barrier()
    q.put((rank, first, n, mx, [int(c) for c in counts]))
    dist.destroy_process_group()


def test_two_ranks_gloo():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert [r[1:3] for r in res] == [(0, 512), (512, 512)]
    for r in res:
        assert r[3] == [1.5, 20.0]                  # every rank sees the slowest rank's time
        assert r[4] == [512, 512]
    assert job_throughput([512 * 10, 512 * 10], res[0][3][0]) == pytest.approx(10240 / 1.5)


def _run_bench(args, extra_env, timeout=300):
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, **extra_env)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
        env.pop(k, None)  # no launcher: bench.py must start its ranks itself
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py")] + args, env=env, capture_output=True, text=True, timeout=timeout)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout  # ONE JSON line, from rank 0
    return json.loads(lines[0])


def test_bench_gpus_flag_spawns_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher around it starts two ranks (fresh child processes, gloo rendezvous on
    127.0.0.1) and rank 0 reports n_gpus 2.  BENCH_DRY_RUN stops before anything needs a GPU (the real run of the same
    path on a one-GPU box is tests/test_gpu_round3.py::test_bench_two_ranks_on_one_gpu)."""
    out = _run_bench(["--gpus", "2", "--config", "4"], {"BENCH_SHARE_GPU": "1", "BENCH_BACKEND": "gloo", "BENCH_DRY_RUN": "1"})
    assert out["n_gpus"] == 2 and out["dry_run"] is True and out["value"] is None
    assert [(r["rank"], r["local_rank"], r["first_stream"], r["streams"]) for r in out["ranks"]] == [(0, 0, 0, 512), (1, 1, 512, 512)]
    one = _run_bench(["--gpus", "1"], {"BENCH_DRY_RUN": "1"})
    assert one["n_gpus"] == 1 and len(one["ranks"]) == 1


def test_bench_refuses_more_gpus_than_the_node_has():
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "BENCH_SHARE_GPU")}
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "64"], env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode != 0 and "GPU(s)" in p.stderr
