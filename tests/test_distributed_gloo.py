"""N > 1 path on CPU: two gloo ranks shard a batch, run a stand-in forward on their shard and
all-gather the detections; covers equal and ragged shards."""
import os
import socket

import pytest

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import msod_amd  # noqa: F401
from msod_amd import distributed as D


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _worker(rank, world, port, n_pairs, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    r, w, _ = D.init_from_env("gloo")
    assert (r, w) == (rank, world)
    g = torch.Generator().manual_seed(0)
    rgb = torch.rand(n_pairs, 3, 8, 8, generator=g)
    ir = torch.rand(n_pairs, 3, 8, 8, generator=g)

    def fake_model(a, b):   # per-pair independent, like the real forward in eval mode
        return (a.mean((2, 3)) + 2 * b.mean((2, 3))).unsqueeze(1).repeat(1, 5, 2), None

    out = D.sharded_forward(fake_model, rgb, ir, rank, world)
    full, _ = fake_model(rgb, ir)
    ok = torch.equal(out, full)
    def fake_nms(pred):    # keeps the first 3 rows of every image
        return pred[:, :3, :6].contiguous() if pred.shape[2] >= 6 else pred[:, :3].repeat(1, 1, 3), torch.full((pred.shape[0],), 3, dtype=torch.int32)

    dets, counts = D.sharded_detect(fake_model, rgb, ir, rank, world, fake_nms)
    fd, fc = fake_nms(full)
    ok = ok and torch.equal(dets, fd) and torch.equal(counts, fc)
    # the bench's overlapped gather: the source tensor is overwritten every step, results must still be per-step
    src = torch.zeros(2, 3)
    og = D.OverlappedGather(src, world)
    outs = []
    for step in range(5):
        src.fill_(10.0 * step + rank)
        outs.append((step, og.submit(src)))
        if step >= 1:                      # result of the previous step must be intact once its buffer is drained later
            pass
    last = og.drain()
    ok = ok and torch.equal(last, torch.cat([torch.full((2, 3), 40.0 + r) for r in range(world)]))
    ok = ok and torch.equal(outs[3][1], torch.cat([torch.full((2, 3), 30.0 + r) for r in range(world)]))
    same = D.gather_equal(torch.full((2, 3, 4), float(rank)))
    ok = ok and torch.equal(same, torch.cat([torch.full((2, 3, 4), 0.0), torch.full((2, 3, 4), 1.0)]))
    # bench.py's measurement loop (timed_steps) with a stub step: rank 1 is the slow rank, both ranks must report ITS time;
    # the overlapped gather must have delivered the last timed step's tensor from both ranks
    import time
    calls = []
    static = torch.zeros(2, 3)

    def step_fn():
        calls.append(len(calls))
        time.sleep(0.02 if rank == 1 else 0.002)
        static.fill_(100.0 * len(calls) + rank)
        return static

    og2 = D.OverlappedGather(static, world)
    elapsed = D.timed_steps(step_fn, steps=4, warmup=2, world=world, gather=og2)
    ok = ok and len(calls) == 6 and elapsed >= 4 * 0.02 * 0.9
    ok = ok and torch.equal(og2.drain(), torch.cat([torch.full((2, 3), 600.0 + r) for r in range(world)]))
    t = torch.tensor([elapsed], dtype=torch.float64)
    both = [torch.zeros_like(t) for _ in range(world)]
    dist.all_gather(both, t)
    ok = ok and float(both[0]) == float(both[1])           # MAX over ranks: identical on every rank
    slow_local = D.timed_steps.last_local_elapsed
    # bench.py's step mode: two forwards in flight (ForwardPipeline), each step's output submitted to the overlapped gather
    bufs = [torch.zeros(2, 3), torch.zeros(2, 3)]
    n_run = [0, 0]

    def runner(i):
        def run():
            n_run[i] += 1
            bufs[i].fill_(1000.0 * (n_run[0] + n_run[1]) + rank)
            return bufs[i]
        return run

    og3 = D.OverlappedGather(bufs[0], world)
    pipe = D.ForwardPipeline([runner(0), runner(1)], None, og3)
    D.timed_steps(pipe.step, steps=5, warmup=2, world=world, gather=og3)
    ok = ok and n_run == [4, 3] and pipe.tick == 7                       # steps alternate between the two runners
    ok = ok and torch.equal(og3.drain(), torch.cat([torch.full((2, 3), 7000.0 + r) for r in range(world)]))
    # the self-check that an N > 1 bench line carries: rank count, per-rank clocks, gathered rows == each rank's local rows
    local = torch.full((2, 3), 600.0 + rank)
    chk = D.gather_selfcheck(local, og2.drain(), rank, world, elapsed_local=slow_local)
    ok = ok and chk["n_ranks_seen"] == world and chk["rows_ok"] and len(chk["per_rank_elapsed_s"]) == world
    ok = ok and chk["per_rank_elapsed_s"][1] >= 4 * 0.02 * 0.9 and chk["gather_ms"] >= 0.0
    bad = D.gather_selfcheck(local + (1.0 if rank == 1 else 0.0), og2.drain(), rank, world)     # a rank whose rows did NOT arrive
    ok = ok and not bad["rows_ok"]
    q.put((rank, ok, tuple(out.shape)))
    dist.barrier()
    dist.destroy_process_group()


def _run(n_pairs):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_worker, args=(r, 2, port, n_pairs, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = [q.get(timeout=120) for _ in ps]
    for p in ps:
        p.join(60)
        assert p.exitcode == 0
    for rank, ok, shape in res:
        assert ok and shape[0] == n_pairs, (rank, ok, shape)


def test_shard_bounds():
    assert [D.shard_bounds(10, r, 4) for r in range(4)] == [(0, 3), (3, 6), (6, 8), (8, 10)]
    assert [D.shard_bounds(512, r, 8) for r in range(8)][-1] == (448, 512)
    assert D.shard_bounds(1, 1, 2) == (1, 1)


def test_two_rank_equal_shards():
    _run(6)


def test_two_rank_ragged_shards():
    _run(5)


def test_two_rank_one_empty_shard():
    """Global batch 1 on 2 ranks: rank 1 has no pair, skips the forward and still joins the collectives."""
    _run(1)


# ----------------------------------------------------------------------------- bench.py --gpus N without a launcher
def test_launch_command_control_flow():
    """``bench.py --gpus N`` started as a plain process must become N ranks under torch.distributed.run (or refuse), never time one
    rank and report it as N: no re-launch at N = 1 or under a launcher, refusal with fewer visible devices, else the torchrun argv."""
    assert D.launch_command("bench.py", 1, ["--gpus", "1"], env={}) is None
    assert D.launch_command("bench.py", 4, ["--gpus", "4"], env={"WORLD_SIZE": "4"}, visible_devices=0) is None      # already launched
    with pytest.raises(SystemExit) as ei:
        D.launch_command("bench.py", 4, ["--gpus", "4"], env={}, visible_devices=1)
    assert "only 1 GPU(s) visible" in str(ei.value)
    cmd = D.launch_command("bench.py", 4, ["--gpus", "4", "--steps", "5"], env={}, visible_devices=8, python="python")
    assert cmd[:4] == ["python", "-m", "torch.distributed.run", "--nnodes=1"] and "--nproc-per-node=4" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and int(cmd[cmd.index("--master-port") + 1]) > 0
    assert cmd[-5:] == ["bench.py", "--gpus", "4", "--steps", "5"]


def test_relaunched_ranks_form_a_process_group(tmp_path):
    """The argv ``launch_command`` returns, executed for real with two CPU ranks (gloo): the re-launched script sees WORLD_SIZE = 2
    (so it does not re-launch again), both ranks rendezvous on 127.0.0.1 and an all-gather of their ranks sees {0, 1}."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "relaunch_probe.py"
    script.write_text(
        "import os, sys\n"
        f"sys.path.insert(0, {root!r})\n"
        "import torch, torch.distributed as dist\n"
        "import msod_amd\n"
        "from msod_amd import distributed as D\n"
        "cmd = D.launch_command(os.path.abspath(__file__), int(sys.argv[1]), sys.argv[1:], visible_devices=2)\n"
        "if cmd is not None:\n"
        "    os.execvp(cmd[0], cmd)\n"
        "rank, world, local = D.init_from_env(backend='gloo')\n"
        "seen = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]\n"
        "dist.all_gather(seen, torch.tensor([rank]))\n"
        "if rank == 0:\n"
        "    print('RANKS', world, sorted(int(t) for t in seen), flush=True)\n"
        "dist.barrier(); dist.destroy_process_group()\n")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, str(script), "2"], capture_output=True, text=True, timeout=240, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "RANKS 2 [0, 1]" in out.stdout, out.stdout + out.stderr[-2000:]
