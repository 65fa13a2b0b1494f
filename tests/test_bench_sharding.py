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


def test_control_plane_world8_gloo(tmp_path):
    """The N = 8 line's control plane, eight ranks over gloo on the CPU (a GPU box admits six processes on its card, so the
    eight-rank case cannot be rehearsed there): rendezvous, every rank's device ordinal gathered in rank order, the barrier
    + max-over-ranks timing with every rank's own time, the thread clamp of self_launch."""
    code = textwrap.dedent("""
        import os, sys, time, json
        sys.path.insert(0, %r)
        import torch, torch.distributed as dist
        import bench
        rank, local, world = bench.dist_env()
        assert world == 8
        torch.set_num_threads(max(1, bench.effective_cores() // world))
        dist.init_process_group("gloo", rank=rank, world_size=world)
        devs = bench.gather_device_ordinals(local, world)
        assert devs == list(range(8)), devs
        per = []
        el = bench.timed_steps(lambda: time.sleep(0.01 if rank != 5 else 0.03), lambda: None, steps=4, warmup=1, world=world,
                               backend_ready=True, per_rank=per)
        assert len(per) == 8 and max(per) == per[5] and el >= per[5]
        if rank == 0:
            print(json.dumps({"el": el, "per": per, "threads": torch.get_num_threads()}))
        dist.destroy_process_group()
    """ % ROOT)
    script = tmp_path / "w8.py"
    script.write_text(code)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29613", OMP_NUM_THREADS="1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8",
                        "--master-addr", "127.0.0.1", "--master-port", "29613", str(script)],
                       capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    import json
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert 0.11 <= d["el"] <= 2.0 and d["threads"] >= 1
