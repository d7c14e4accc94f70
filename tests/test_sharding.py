"""CPU-only: the N > 1 path (restart sharding + host-side gather + global minimum in
ScoredSolution order) on torch.distributed/gloo with world size 2.  The per-restart
"evaluation" is done by the CPU oracle here (tests may use it); on the GPUs the same
sharding code wraps the HIP path (bench.py)."""
import os
import socket
import subprocess
import sys
import textwrap

import numpy as np

import oracle as orc
from ls_amd import sharding

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_restarts_partition():
    for world in (1, 2, 3, 8):
        for total in (0, 1, 8, 64, 65):
            parts = [sharding.shard_restarts(total, world, r) for r in range(world)]
            flat = sorted(x for p in parts for x in p)
            assert flat == list(range(total))
            for r, p in enumerate(parts):
                assert all(x % world == r for x in p)          # restart r -> GPU r mod N
                assert len(p) in (total // world, total // world + 1)


def test_record_order_is_scored_solution_order():
    # (score, solution vector): score first, then lexicographic solution (local_search.rs:29-37)
    a = sharding.record(5, 10, [3, 1, 2])
    b = sharding.record(2, 10, [3, 0, 9])
    c = sharding.record(9, 8, [9, 9, 9])
    assert sharding.best_of([a, b, c]) == c
    assert sharding.best_of([a, b]) == b
    # two-level scores (ScheduleScore: hard then soft)
    d = sharding.record(0, (1, 50), [0])
    e = sharding.record(1, (2, 0), [0])
    assert sharding.best_of([e, d]) == d
    assert sharding.best_of([]) is None


_WORKER = textwrap.dedent("""
    import json, os, sys
    import numpy as np
    import torch.distributed as dist
    sys.path.insert(0, os.path.join({root!r}, "constraint-solver_amd"))
    sys.path.insert(0, os.path.join({root!r}, "oracle"))
    import oracle as orc
    from ls_amd import sharding
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:{port}", rank=int(sys.argv[1]), world_size=2)
    rank = dist.get_rank()
    n, total = 64, 7
    recs = []
    for rid in sharding.shard_restarts(total, 2, rank):
        rows = np.random.default_rng(rid).integers(0, n, size=n)
        res = orc.nq_neighbourhood(rows, np.arange(n), n * (n - 1), "table", want_scores=False)
        b = rows.copy(); b[res["best_index"] // n] = res["best_index"] % n
        recs.append(sharding.record(rid, res["best_score"], b))
    best, allr = sharding.gather_best(recs)
    dist.barrier()
    print(json.dumps(dict(rank=rank, n_local=len(recs), n_all=len(allr), best_score=best[0][0], best_restart=best[2],
                          best_board=list(best[1]))))
    dist.destroy_process_group()
""")


def test_two_rank_gloo_gather_matches_single_process(tmp_path):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    script = tmp_path / "worker.py"
    script.write_text(_WORKER.format(root=ROOT, port=port))
    env = dict(os.environ, OMP_NUM_THREADS="1")
    procs = [subprocess.Popen([sys.executable, str(script), str(r)], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                              env=env, text=True) for r in range(2)]
    outs = []
    for p in procs:
        o, e = p.communicate(timeout=180)
        assert p.returncode == 0, e[-2000:]
        outs.append(__import__("json").loads(o.strip().splitlines()[-1]))
    # single-process expectation
    n, total = 64, 7
    recs = []
    for rid in range(total):
        rows = np.random.default_rng(rid).integers(0, n, size=n)
        res = orc.nq_neighbourhood(rows, np.arange(n), n * (n - 1), "table", want_scores=False)
        b = rows.copy(); b[res["best_index"] // n] = res["best_index"] % n
        recs.append(sharding.record(rid, res["best_score"], b))
    exp = sharding.best_of(recs)
    assert sorted(o["n_local"] for o in outs) == [3, 4]
    for o in outs:
        assert o["n_all"] == total
        assert o["best_score"] == exp[0][0] and o["best_restart"] == exp[2]
        assert tuple(o["best_board"]) == exp[1]


def test_thread_ranks_barrier_and_gather():
    # bench.py without a launcher: ranks are host threads of one process (one context + stream per GPU); they meet at
    # a barrier and at the host-side gather only (no collective on the data path)
    import importlib.util
    import threading
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    world = 4
    sync = bench._ThreadSync(world)
    got = [None] * world

    def rank(r):
        recs = [sharding.record(rid, 10 * rid + 3, [rid, 0, 1]) for rid in sharding.shard_restarts(10, world, r)]
        sync.barrier()
        parts = sync.all_gather(r, recs)
        sync.barrier()
        got[r] = sharding.best_of([x for p in parts for x in p]), sorted(x[2] for p in parts for x in p)

    ts = [threading.Thread(target=rank, args=(r,)) for r in range(world)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert all(g == got[0] for g in got)
    assert got[0][1] == list(range(10)) and got[0][0][2] == 0
