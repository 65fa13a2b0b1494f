"""CPU, world_size 2 over gloo: the bench's multi-process scaffolding (barrier, exactly-K
timed steps, MAX over ranks) and the frame-shard helper.  The data path itself has no
collective (frames are independent), so this is all the N>1 logic there is."""
import os
import subprocess
import sys
import textwrap

import bench

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_frame_shard_partitions_exactly():
    for total in (0, 1, 7, 8, 1000, 1001):
        for world in (1, 2, 3, 8):
            spans = [bench.frame_shard(total, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


def test_timed_steps_max_over_ranks_gloo_world2(tmp_path):
    code = textwrap.dedent("""
        import os, sys, time, json
        sys.path.insert(0, %r)
        import torch, torch.distributed as dist
        import bench
        rank, local, world = bench.dist_env()
        dist.init_process_group("gloo", rank=rank, world_size=world)
        calls = []
        def step():
            calls.append(1); time.sleep(0.02 * (rank + 1))      # rank 1 is twice as slow
        el = bench.timed_steps(step, lambda: None, steps=5, warmup=2, world=world, backend_ready=True)
        assert len(calls) == 7
        lo, hi = bench.frame_shard(101, world, rank)
        t = torch.tensor([hi - lo]); dist.all_reduce(t)
        assert int(t) == 101
        if rank == 0:
            print(json.dumps({"el": el}))
        dist.destroy_process_group()
    """ % ROOT)
    script = tmp_path / "w.py"
    script.write_text(code)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29611")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", "29611", str(script)],
                       capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    import json
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    el = json.loads(line)["el"]
    assert 0.19 <= el <= 1.0, el     # 5 steps x 40 ms of the slow rank, not 5 x 20 ms
