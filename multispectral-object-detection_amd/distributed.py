"""Batch-sharded multi-GPU inference: one process per GPU, full weight replica per rank, image
pairs split across ranks, one all-gather of the pre-NMS detections (RCCL over xGMI; the
torch.distributed backend string "nccl" IS RCCL on ROCm).

The reference has no multi-GPU inference at all (only DDP training,
train.py:654-658,993); this is the new collective of BASELINE.json's north_star.  Each pair's
forward is independent in eval mode, so there is no other communication.
"""
import os
import time

import torch
import torch.distributed as dist


def init_from_env(backend=None, force=False):
    """Initialise torch.distributed from RANK/WORLD_SIZE/MASTER_* (torchrun convention, same env
    contract as reference train.py:960-961,989-995).  Returns (rank, world_size, local_rank).
    ``force``: also build a process group when WORLD_SIZE is 1 (rendezvous on 127.0.0.1), so that the collective path
    - RCCL communicator, its stream, ``OverlappedGather`` - can be exercised on the one GPU a box has."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", str(rank)))
    if force and world == 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:
            import socket
            with socket.socket() as sock:
                sock.bind(("127.0.0.1", 0))
                os.environ["MASTER_PORT"] = str(sock.getsockname()[1])
    if (world > 1 or force) and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, init_method="env://", rank=rank, world_size=world)
    return rank, world, local


def launch_command(script, gpus, argv, env=None, visible_devices=None, python=None):
    """What ``bench.py --gpus N`` does when it is started WITHOUT a launcher: N > 1 ranks cannot be one process,
    so the script re-executes itself under ``python -m torch.distributed.run`` - the reference's launch convention
    (utils/aws/resume.py:31 ``torch.distributed.launch``, train.py:989-995: one process per GPU, RANK / WORLD_SIZE from the env) -
    instead of silently timing one rank and reporting ``n_gpus: 1``.

    Returns None when no re-launch is needed (N == 1, or WORLD_SIZE already set by a launcher), else the argv list to exec.
    Raises SystemExit when fewer than N devices are visible."""
    import sys
    env = os.environ if env is None else env
    if gpus <= 1 or "WORLD_SIZE" in env:
        return None
    if visible_devices is None:
        visible_devices = torch.cuda.device_count()
    if visible_devices < gpus:
        raise SystemExit(f"--gpus {gpus}: only {visible_devices} GPU(s) visible to this process; refusing to report a {gpus}-GPU number "
                         "from fewer devices")
    import socket
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    return [python or sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={gpus}", "--master-addr", "127.0.0.1",
            "--master-port", str(port), script] + list(argv)


def shard_bounds(n, rank, world):
    """Contiguous shard [lo, hi) of ``n`` items for ``rank``; the first n % world ranks get one extra."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_batch(x, rank, world):
    lo, hi = shard_bounds(x.shape[0], rank, world)
    return x[lo:hi]


def all_gather_detections(pred, world=None, group=None, like=None):
    """pred [B_local, rows, no] -> [sum(B_local), rows, no] on every rank, in rank order.
    Equal local batches use one all_gather_into_tensor; ragged ones pad to the max batch.  A rank whose shard is
    EMPTY (global batch < world size) passes ``pred=None`` and ``like=(ndim, dtype, device)``: it learns the row
    shape from the other ranks and still takes part in every collective (no rank may skip one - ADVICE r1)."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return pred
    world = world or dist.get_world_size(group)
    if pred is not None:
        pred = pred.contiguous()
        ndim, dtype, device = pred.dim(), pred.dtype, pred.device
        shape = list(pred.shape)
    else:
        ndim, dtype, device = like
        shape = [0] * ndim
    mine = torch.tensor(shape, device=device, dtype=torch.int64)
    shapes = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(shapes, mine, group=group)
    shapes = [[int(v) for v in t.tolist()] for t in shapes]
    sizes = [t[0] for t in shapes]
    trail = tuple(max(t[d] for t in shapes) for d in range(1, ndim))
    bmax = max(sizes)
    if bmax == 0:
        return torch.empty((0,) + trail, dtype=dtype, device=device)
    if min(sizes) == bmax:
        out = torch.empty((world * bmax,) + trail, dtype=dtype, device=device)
        dist.all_gather_into_tensor(out, pred, group=group)
        return out
    padded = torch.zeros((bmax,) + trail, dtype=dtype, device=device)
    if pred is not None and pred.shape[0]:
        padded[:pred.shape[0]] = pred
    chunks = [torch.empty_like(padded) for _ in range(world)]
    dist.all_gather(chunks, padded, group=group)
    return torch.cat([c[:n] for c, n in zip(chunks, sizes)], 0)


def gather_equal(pred, out=None, group=None):
    """Fast path for the benchmark loop: every rank holds the same local batch, no size exchange."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return pred
    world = dist.get_world_size(group)
    if out is None:
        out = torch.empty((world * pred.shape[0],) + tuple(pred.shape[1:]), dtype=pred.dtype, device=pred.device)
    dist.all_gather_into_tensor(out, pred.contiguous(), group=group)
    return out


def sharded_forward(model, rgb, ir, rank, world, group=None):
    """Run ``model`` on this rank's shard of the global batch and gather every rank's detections.  A rank with an
    empty shard (batch < world) skips the forward but not the collective."""
    a, b = shard_batch(rgb, rank, world), shard_batch(ir, rank, world)
    if a.shape[0] == 0:
        return all_gather_detections(None, world, group, like=(3, torch.float32, rgb.device))
    pred, _ = model(a, b)
    return all_gather_detections(pred, world, group)


def sharded_detect(model, rgb, ir, rank, world, nms, group=None):
    """Forward + on-device NMS on this rank's shard, then gather only the survivors: the collective moves
    [B_local, max_det, 6] + counts (0.46 MB per rank at 64 pairs) instead of the 25200-row prediction tensor
    (51.6 MB) - SURVEY.md section 8f rank 1.  ``nms(pred) -> (dets, counts)`` is
    ``utils.general.batched_nms`` (or a partial of it).  Returns (dets [B, max_det, 6], counts [B]) on all ranks."""
    a, b = shard_batch(rgb, rank, world), shard_batch(ir, rank, world)
    multi = dist.is_initialized() and dist.get_world_size(group) > 1
    if a.shape[0] == 0 and multi:
        dets = all_gather_detections(None, world, group, like=(3, torch.float32, rgb.device))
        counts = all_gather_detections(None, world, group, like=(2, torch.float32, rgb.device))
        return dets, counts.view(-1).to(torch.int32)
    pred, _ = model(a, b)
    dets, counts = nms(pred)
    if not multi:
        return dets, counts
    return all_gather_detections(dets, world, group), all_gather_detections(counts.view(-1, 1).to(torch.float32), world, group).view(-1).to(counts.dtype)


def timed_steps(step_fn, steps, warmup, world=1, gather=None, sync=None, group=None):
    """The measurement loop of bench.py, factored out so that its N > 1 control flow runs under the CPU (gloo)
    tests too: ``warmup`` untimed steps, then EXACTLY ``steps`` timed steps bracketed by barrier + device sync on
    both sides; returns the MAX elapsed seconds over ranks.  ``step_fn()`` runs one step and returns the tensor to
    gather (or None); ``gather`` is an ``OverlappedGather`` (its collective of step i overlaps step i+1 and all of
    them are drained inside the timed region); ``sync`` is ``torch.cuda.synchronize`` on a GPU, a no-op on CPU."""
    sync = sync or (lambda: None)
    multi = world > 1 and dist.is_initialized()

    def one():
        out = step_fn()
        if gather is not None and out is not None:
            gather.submit(out)

    for _ in range(warmup):
        one()
    if gather is not None:
        gather.drain()
    if multi:
        dist.barrier(group=group)
    sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        one()
    if gather is not None:
        gather.drain()                      # every gather of the K timed steps has completed
    sync()
    if multi:
        dist.barrier(group=group)
    elapsed = time.perf_counter() - t0
    timed_steps.last_local_elapsed = elapsed          # this rank's own clock (the self-check gathers one per rank)
    if multi:
        t = torch.tensor([elapsed], dtype=torch.float64, device=gather.out[0].device if gather is not None else "cpu")
        if t.device.type == "cpu" and dist.get_backend(group) == "nccl":
            t = t.cuda()
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
        elapsed = float(t.item())
    return elapsed


class OverlappedGather:
    """Double-buffered, asynchronous all-gather of a tensor that is overwritten every step (the static output of
    a captured HIP graph): ``submit(x)`` copies ``x`` into one of two staging buffers and launches
    ``all_gather_into_tensor`` with ``async_op=True``, so the collective of step i runs on RCCL's stream while
    the forward of step i+1 runs on the compute stream; a buffer is only reused after its previous collective
    has been waited for.  ``drain()`` waits for everything in flight and returns the most recent gathered tensor."""

    def __init__(self, like, world, group=None):
        self.group = group
        self.stage = [torch.empty_like(like) for _ in range(2)]
        self.out = [torch.empty((world * like.shape[0],) + tuple(like.shape[1:]), dtype=like.dtype, device=like.device)
                    for _ in range(2)]
        self.pending = [None, None]
        self.tick = 0

    def submit(self, x):
        i = self.tick & 1
        self.tick += 1
        if self.pending[i] is not None:
            self.pending[i].wait()
        self.stage[i].copy_(x, non_blocking=True)
        self.pending[i] = dist.all_gather_into_tensor(self.out[i], self.stage[i], group=self.group, async_op=True)
        return self.out[i]

    def drain(self):
        for i in range(2):
            if self.pending[i] is not None:
                self.pending[i].wait()
                self.pending[i] = None
        return self.out[(self.tick - 1) & 1] if self.tick else None


def gather_selfcheck(local, gathered, rank, world, elapsed_local=None, group=None, gather_probe_steps=3, force=False):
    """Evidence for an N > 1 bench line, computed from collectives so that a stub or a silently
    single-rank run cannot produce it: ``n_ranks_seen`` = ranks that contributed to an all-gather of their rank ids;
    ``rows_ok`` = rank 0's gathered rows [B r, B (r+1)) equal rank r's local tensor (compared through a float64
    sum, a float64 sum of squares and 16 strided samples that every rank all-gathers); ``per_rank_elapsed_s`` = every
    rank's own timed-region seconds; ``gather_ms`` = median wall time of ``gather_probe_steps`` synchronous gathers.
    Returns a dict on every rank (identical content).  ``force``: run the collectives with a one-rank group too
    (``init_from_env(force=True)``: they then go through RCCL with world size 1 instead of being skipped)."""
    import statistics
    multi = dist.is_initialized() and (dist.get_world_size(group) > 1 or force)
    if not multi:
        return {"n_ranks_seen": 1, "rows_ok": True, "per_rank_elapsed_s": [elapsed_local], "gather_ms": 0.0}
    dev = local.device
    ids = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(ids, torch.tensor([rank], dtype=torch.int64, device=dev), group=group)
    seen = sorted({int(t.item()) for t in ids})

    def fingerprint(t):
        f = t.reshape(-1).to(torch.float64)
        idx = torch.linspace(0, f.numel() - 1, 16, device=f.device).long()
        return torch.cat([f.sum().view(1), (f * f).sum().view(1), f[idx]])

    mine = fingerprint(local)
    fps = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(fps, mine, group=group)
    B = local.shape[0]
    rows_ok = all(torch.equal(fingerprint(gathered[r * B:(r + 1) * B]), fps[r]) for r in range(world))
    el = torch.tensor([elapsed_local if elapsed_local is not None else 0.0], dtype=torch.float64, device=dev)
    els = [torch.zeros_like(el) for _ in range(world)]
    dist.all_gather(els, el, group=group)
    sync = torch.cuda.synchronize if dev.type == "cuda" else (lambda: None)
    times = []
    out = torch.empty((world * B,) + tuple(local.shape[1:]), dtype=local.dtype, device=dev)
    for _ in range(gather_probe_steps):
        dist.barrier(group=group)
        sync()
        t0 = time.perf_counter()
        dist.all_gather_into_tensor(out, local.contiguous(), group=group)
        sync()
        times.append((time.perf_counter() - t0) * 1e3)
    return {"n_ranks_seen": len(seen), "rows_ok": bool(rows_ok), "per_rank_elapsed_s": [round(float(t.item()), 6) for t in els],
            "gather_ms": round(statistics.median(times), 3)}


class ForwardPipeline:
    """Several forwards in flight: ``runners[i]()`` enqueues forward i (e.g. the replay of a captured HIP graph with its
    OWN static input / output buffers) and returns its static output tensor; step t runs runner t % K on stream t % K,
    so consecutive steps overlap on the GPU while the steps of one runner stay ordered on its stream.  Consecutive
    inference steps are independent (different batches in deployment), which is what makes this legal; it is the compute
    counterpart of ``OverlappedGather`` (step t's collective under step t + 1's forward).  ``gather`` (optional): every
    step's output is submitted to it on the step's stream.
    Ordering contract: ``step()`` does NOT order its stream after the caller's current stream - whoever writes a runner's
    input buffers must make those writes complete (or make ``streams[i]`` wait for them) before the step that reads them."""

    def __init__(self, runners, streams=None, gather=None):
        self.runners, self.streams, self.gather, self.tick = list(runners), streams, gather, 0
        if streams is not None and len(streams) != len(self.runners):
            raise ValueError("one stream per runner")

    @staticmethod
    def pick_streams(runners, device, groups=4, probe_steps=6, priorities=None):
        """HIP maps streams onto a few hardware queues, and a captured forward brings internal branch streams of its own: which
        streams the forwards in flight are replayed on changes the steady-state rate by up to 8 % (two attractors, measured:
        profiles/r03_forwards_in_flight.txt).  This draws ``groups`` candidate groups of K streams, times ``probe_steps`` steps per
        forward on each and returns the fastest group (plus the measured table).  ``priorities``: optional HIP stream priority per
        forward; round 4 measured (-1, 0) and (-1, -1) SLOWER than the default equal priorities (profiles/r04_forwards_in_flight.txt)."""
        k = len(runners)
        torch.cuda.synchronize(device)      # the runners' inputs (written on the caller's stream) are complete before any probe replays them
        pr = list(priorities) if priorities is not None else [0] * k      # optional HIP stream priority per forward in flight (-1 = high)
        cands = [[torch.cuda.Stream(device=device, priority=pr[i % len(pr)]) for i in range(k)] for _ in range(groups)]
        table = []
        for streams in cands:
            pipe = ForwardPipeline(runners, streams)
            for _ in range(k):
                pipe.step()
            torch.cuda.synchronize(device)
            t0 = time.perf_counter()
            for _ in range(probe_steps * k):
                pipe.step()
            torch.cuda.synchronize(device)
            table.append((time.perf_counter() - t0) / (probe_steps * k))
        best = min(range(groups), key=lambda i: table[i])
        return cands[best], [round(t * 1e3, 3) for t in table]

    def step(self):
        import contextlib
        i = self.tick % len(self.runners)
        self.tick += 1
        ctx = torch.cuda.stream(self.streams[i]) if self.streams is not None else contextlib.nullcontext()
        with ctx:
            out = self.runners[i]()
            if self.gather is not None and out is not None:
                self.gather.submit(out)
        return None        # (timed_steps must not submit a second time)
