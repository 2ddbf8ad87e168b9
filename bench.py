#!/usr/bin/env python3
"""bench.py -- training images/sec of the hypernet-conditioned captioning step on MI355X.

  python bench.py --gpus N --steps K --warmup W
  (N > 1: under `python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...` it is one rank;
   started bare, it launches those N ranks itself before touching the GPU and passes rank 0's line through)

One "step" = hypernet forward -> decoder forward -> cross entropy -> backward -> gradient exchange
-> global-norm clip -> Adam on one synthetic Flickr30k-shaped minibatch already resident in HBM
(BASELINE.json configs[1]: GRU decoder + hypernet, 3 style domains, bs=128 per GPU).  Weak scaling:
the per-GPU batch is fixed.  Rank 0 prints ONE JSON line.

Extra objects on that line:
  roofline     the dominant kernel (adam_rank_kernel on hn_heads.0.2.weight, 115.2 M parameters):
               algorithmic bytes = 24 B/parameter (read W,m,v + write W,m,v) / its measured duration
               (HIP events on the launch stream), against 8 TB/s HBM peak.
               copy_ceiling_gbps: this box's device-to-device copy rate, measured in the same run.
  module_api   SIDE figure (never `value`): the unchanged-driver loop training_step -> backward -> optimizer.step on the
               module API with the optimiser configure_optimizers() returns (caphn.optim.FusedAdam).
  collectives  what the data-parallel exchange issued: backend, world size, collectives per step, the time the main stream
               waited for them in front of the optimiser (exposed_us).
  cpu_baseline the CPU oracle (oracle/caphn_oracle.py, a port of the reference's PyTorch path) timed
               on this box's host cores on a bounded sample of the same workload (rank 0, N = 1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "hypernet-image-captioning_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import numpy as np  # noqa: E402
import torch  # noqa: E402

METRIC = "training images/sec at 1/2/4/8 MI355X, Flickr30k GRU+hypernet bs=128"
HBM_PEAK = 8.0e12           # B/s, MI355X_MICROARCH.md chip table
STEP_ALGO_BYTES = 5.228e9   # whole-step algorithmic HBM bytes at B=128
DOMINANT_KERNEL_SYMBOL = "adam_rank_kernel<2, true, true, 2, false>"   # instantiation that runs hn_heads.0.2.weight (k = 480) in the step


class _Vocab:
    w2i = {"<pad>": 0, "<s>": 1, "</s>": 2, "<unk>": 3, "factual": 4, "humorous": 5, "romantic": 6}

    def __call__(self, w):
        return self.w2i.get(w, 3)


def synth_batches(n, B, T, P, D, V, device, seed):
    """SURVEY.md 8d synthetic inputs, generated on the device."""
    g = torch.Generator(device=device).manual_seed(seed)
    out = []
    for _ in range(n):
        feats = torch.relu(torch.randn(B, P, D, generator=g, device=device)) * 0.45
        L = torch.clamp(torch.round(torch.randn(B, generator=g, device=device) * 4.0 + 12.9), 5, T).long()
        toks = torch.randint(7, V, (B, T), generator=g, device=device)
        pos = torch.arange(T, device=device)[None, :]
        caps = torch.where(pos < (L[:, None] - 1), toks, torch.zeros_like(toks))
        caps[:, 0] = 1
        caps.scatter_(1, (L - 1)[:, None], 2)
        out.append((feats.contiguous(), caps.contiguous()))
    return out


def usable_cores():
    """Cores this process may really use: affinity mask capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                q = int(txt[0])
                per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if q > 0:
                    n = min(n, max(1, q // per))
            break
        except Exception:  # noqa: BLE001
            continue
    return max(1, min(n, 64))


def cpu_baseline(B, T, P, budget_s=25.0, max_steps=5):
    """Oracle (port of the reference's path) on the host cores: 2 warm-up steps, then up to
    max_steps timed full steps (forward, backward, clip, Adam; the median is `value`), then 3 steps without the
    optimiser (forward + backward only, reported beside it).  Bounded by budget_s seconds."""
    from oracle import caphn_oracle as O
    cores = usable_cores()
    torch.set_num_threads(cores)
    dims = O.Dims()
    p = O.init_params(dims, seed=1)
    batch = O.synth_batch(dims, B, T, P, seed=2)
    state = {}
    t_start = time.perf_counter()
    for w in range(2):
        O.train_step(dims, p, state, w + 1, None, batch["features"], batch["captions"], lr=1e-3, style_token=4)
    ts = []
    for s in range(max_steps):
        t0 = time.perf_counter()
        O.train_step(dims, p, state, s + 3, None, batch["features"], batch["captions"], lr=1e-3, style_token=4)
        ts.append(time.perf_counter() - t0)
        if time.perf_counter() - t_start > budget_s:
            break
    tn = []
    for s in range(3):
        t0 = time.perf_counter()
        O.forward_backward(dims, p, None, batch["features"], batch["captions"], style_token=4)
        tn.append(time.perf_counter() - t0)
        if time.perf_counter() - t_start > budget_s + 8.0:
            break
    med, medn = float(np.median(ts)), float(np.median(tn))
    return {"value": B / med, "unit": "images/s", "cores": cores, "kind": "port",
            "value_without_adam": B / medn,
            "sample": f"{len(ts)} full steps (fwd+bwd+clip+Adam) at B={B}, T={T}, fp32, after 2 warm-ups; "
                      f"median {med * 1e3:.0f} ms/step; {len(tn)} steps without the optimiser: median {medn * 1e3:.0f} ms"}


def copy_ceiling(dev, nbytes=1_382_400_000, reps=6):
    """Device-to-device copy rate of this box, read + write bytes per second: what a kernel that
    streams as many bytes in as out -- the rank-1 Adam pass reads 12 B and writes 12 B per parameter -- can reach at best.
    Boxes differ by up to 16 % in it; reported beside the 8 TB/s data-sheet peak so that a slow HBM is told apart from a slow kernel."""
    n = nbytes // 4
    a = torch.empty(n, dtype=torch.float32, device=dev).normal_()
    b = torch.empty_like(a)
    from caphn import _lib
    lib = _lib.load()

    def cp():
        _lib.check(lib.caphn_stream_copy_f32(n, _lib.ptr(a), _lib.ptr(b), _lib.stream_ptr()), "caphn_stream_copy_f32")
    cp()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(reps):
        s.record(); cp(); e.record()
        torch.cuda.synchronize()
        ts.append(s.elapsed_time(e))
    del a, b
    return 2.0 * nbytes / (min(ts) * 1e-3) / 1e9


def module_api_loop(dev, dims, batches, steps=10, warmup=3):
    """SIDE measurement, never `value`: the loop an UNCHANGED reference driver runs (hypernet_attention.py:136-204 training_step,
    Lightning's backward / gradient_clip_val / optimizer.step, cc_train_hypernet.py:405, :120) on this package's module API --
    HyperNet.training_step() -> loss.backward() -> optimizer.step(), with the optimiser configure_optimizers() returns
    (caphn.optim.FusedAdam: torch.optim.Optimizer subclass, clip inside step, hypernet second-layer weights from rank-1 factors)."""
    from hypernet_attention import HyperNet
    B, T, P, D, F, E, H, V = dims
    torch.manual_seed(4321)
    net = HyperNet(F, E, H, V, _Vocab()).to(dev)
    (opt,), _ = net.configure_optimizers()
    for g in opt.param_groups:
        g["lr"] = 1e-3
    net.configure_gradient_clipping(opt, gradient_clip_val=5.0, gradient_clip_algorithm="norm")
    styles = ["factual", "humorous", "romantic"]

    def one(i):
        f, c = batches[i % len(batches)]
        opt.zero_grad()
        loss = net.training_step((f, (styles[i % 3], (c, None))), i)
        loss.backward()
        opt.step()
        return loss
    for i in range(warmup):
        one(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        loss = one(warmup + i)
    t_host = (time.perf_counter() - t0) / steps          # the host has enqueued the steps
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    out = {"ms_per_step": dt * 1e3, "images_s": B / dt, "steps": steps, "host_enqueue_ms_per_step": round(t_host * 1e3, 4),
           "final_loss": float(loss.detach()),
           "optimizer": type(opt).__module__ + "." + type(opt).__name__,
           "loop": "HyperNet.training_step -> loss.backward -> optimizer.step (clip 5.0 inside step), one style per step"}
    del net, opt
    torch.cuda.empty_cache()
    return out


def bind_to_gpu_numa(local_rank):
    """Pin this rank's host threads to the CPUs of its GPU's NUMA node -- by sched_setaffinity in THIS process, before anything
    touches the GPU (no re-exec, no numactl wrapper: a process that has initialised the GPU must not exec).  The GPU of local rank
    r is the r-th GPU node of the KFD topology; its CPU list is the PCI device's local_cpulist.  Best effort: any surprise leaves
    the affinity as it was.  CAPHN_BIND_NUMA=0 switches it off.  Returns a short description for the bench line."""
    if os.environ.get("CAPHN_BIND_NUMA", "1") != "1" or not hasattr(os, "sched_setaffinity"):
        return None
    try:
        base = "/sys/class/kfd/kfd/topology/nodes"
        gpus = []
        for n in sorted(os.listdir(base), key=int):
            props = dict(l.split() for l in open(os.path.join(base, n, "properties")) if len(l.split()) == 2)
            if int(props.get("simd_count", "0")) > 0:
                gpus.append(int(props["drm_render_minor"]))
        vis = os.environ.get("HIP_VISIBLE_DEVICES") or os.environ.get("ROCR_VISIBLE_DEVICES")
        if vis:
            gpus = [gpus[int(i)] for i in vis.split(",") if i.strip().isdigit() and int(i) < len(gpus)]
        minor = gpus[local_rank]
        cpus = set()
        for part in open(f"/sys/class/drm/renderD{minor}/device/local_cpulist").read().strip().split(","):
            lo, _, hi = part.partition("-")
            cpus.update(range(int(lo), int(hi or lo) + 1))
        allowed = cpus & os.sched_getaffinity(0)
        if not allowed or allowed == os.sched_getaffinity(0):
            return None
        os.sched_setaffinity(0, allowed)
        return f"renderD{minor}: {len(allowed)} cpus"
    except Exception:  # noqa: BLE001
        return None


def launch_ranks(n, argv):
    """`python bench.py --gpus N` without a launcher: this process -- which has made NO GPU call (counting devices does not
    initialise the GPU) and makes none -- starts N ranks through torch.distributed.run as a child process, passes their
    output through (rank 0 prints the JSON line) and returns the child's exit code."""
    import socket
    import subprocess
    dry = os.environ.get("CAPHN_BENCH_DRYRUN") == "1"
    if not dry and os.environ.get("CAPHN_BENCH_REHEARSAL") != "1":
        have = torch.cuda.device_count()
        if have < n:
            print(f"bench.py: --gpus {n} needs {n} GPUs on this node, {have} visible", file=sys.stderr)
            return 2
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)
    lines = 0
    for ln in proc.stdout:                       # only the result line goes to stdout; anything else is noise
        if ln.startswith('{"metric"'):
            sys.stdout.write(ln); sys.stdout.flush(); lines += 1
        else:
            sys.stderr.write(ln)
    rc = proc.wait()
    if rc == 0 and lines != 1:
        print(f"bench.py: the ranks printed {lines} result lines, expected 1", file=sys.stderr)
        return 3
    return rc


def dryrun(world, rank, args):
    """CAPHN_BENCH_DRYRUN=1: the launcher / rendezvous / max-over-ranks / one-line control flow with no GPU in it
    (tests/test_bench_launcher.py runs it here on CPU).  Never a measurement."""
    import torch.distributed as dist
    if os.environ.get("CAPHN_BENCH_DRYRUN_FAIL_RANK") == str(rank):
        raise SystemExit(7)
    if world > 1:
        dist.init_process_group("gloo")
        dist.barrier()
    t0 = time.perf_counter()
    time.sleep(0.01 * (rank + 1))
    dt = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    if world > 1:
        dist.barrier()
        dist.all_reduce(dt, op=dist.ReduceOp.MAX)
    if rank == 0:
        print(json.dumps({"metric": METRIC, "value": 0.0, "unit": "images/s", "n_gpus": world, "steps": args.steps,
                          "warmup": args.warmup, "dryrun": True, "max_over_ranks_s": float(dt)}), flush=True)
    if world > 1:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=128, help="per-GPU batch")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cell", choices=["gru", "lstm"], default="gru",
                    help="gru: the metric's workload; lstm: BASELINE config 3 (hypernet-generated LSTMCell, side measurement)")
    ap.add_argument("--dtype", choices=["f32", "bf16", "bf16x2"], default="f32",
                    help="f32: the metric's arithmetic (the reference trains with precision=32).  bf16: SIDE measurement, never the "
                         "headline -- every dense contraction as one bf16 MFMA product on operands rounded to bf16, fp32 accumulate; "
                         "recurrent kernels, softmax, loss, Adam and master weights stay fp32 (caphn_tune key 11).  bf16x2: SIDE measurement too -- "
                         "operands as two bf16 planes (16 significand bits), three products: logits within 1e-4 of the fp32 vectors")
    ap.add_argument("--phases", action="store_true", help="also print per-phase timings to stderr")
    ap.add_argument("--no-prefetch", action="store_true",
                    help="do not tell the optimiser pass the next minibatch's style (disables the fused next-theta GEMV)")
    ap.add_argument("--tune", action="append", default=[], help="kernel-variant knob key=value (caphn_tune), for A/B runs")
    ap.add_argument("--graph", action="store_true",
                    help="replay the step from a hipGraph (single GPU).  Default is host-launched kernels: the step is "
                         "GPU-bound either way (measured 2.958 vs 2.977 ms) and host launches let the HIP events around "
                         "the dominant kernel sit inside the timed region")
    ap.add_argument("--eager", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--no-spinup", action="store_true", help="skip the untimed spin-up windows after --warmup")
    ap.add_argument("--pretouch", type=int, default=0, help="issue this many trivial kernel launches (and a synchronise) after "
                    "--warmup: experiment on the one-time host stall of the first process on a fresh box (DESIGN.md section 6)")
    ap.add_argument("--from-host", action="store_true", help="side measurement (DESIGN.md): every minibatch starts in pinned HOST "
                    "memory and crosses PCIe on a copy stream, two batches ahead, into one of three device slots")
    ap.add_argument("--no-module-api", action="store_true", help="skip the side measurement of the unchanged-driver loop "
                    "(training_step -> backward -> clip -> optimizer.step with the optimiser configure_optimizers() returns)")
    ap.add_argument("--fixed-style", action="store_true", help="one style token per rank for the whole run (rounds 1-2); default: "
                    "the style rotates over the three domains step by step, so the fused next-theta GEMV sees a changing input row")
    ap.add_argument("--no-overlap", action="store_true", help="do not issue the next batch's caption-independent "
                    "precompute beside the optimiser")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # not under a launcher: start the ranks ourselves, before anything touches the GPU in this process
        raise SystemExit(launch_ranks(args.gpus, sys.argv[1:]))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but the launcher started {world} ranks (WORLD_SIZE={world})")
    if os.environ.get("CAPHN_BENCH_DRYRUN") == "1":
        return dryrun(world, rank, args)
    # stdout carries ONE line, the result.  Native libraries write there too (RCCL prints a version banner from C when
    # its communicator is created): until the result is ready, file descriptor 1 points at stderr.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    # rehearsal on a one-GPU box (never the measured configuration): CAPHN_BENCH_REHEARSAL=1 puts every rank on
    # device 0 and moves the collectives over gloo, so the N>1 control flow can be exercised without a second GPU
    rehearsal = os.environ.get("CAPHN_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local = 0
    numa = bind_to_gpu_numa(local) if world > 1 and not rehearsal else None      # before the first GPU call of this process
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    import torch.distributed as dist
    # CAPHN_FORCE_COLLECTIVES=1: a one-rank RCCL group still goes through every collective call of the exchange (side
    # measurement / test of the nccl code path on a one-GPU box)
    forced = world == 1 and os.environ.get("CAPHN_FORCE_COLLECTIVES") == "1"
    if world > 1 or forced:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if forced and "MASTER_PORT" not in os.environ:
            import socket
            sk = socket.socket(); sk.bind(("127.0.0.1", 0)); os.environ["MASTER_PORT"] = str(sk.getsockname()[1]); sk.close()
        if rehearsal:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    from hypernet_attention import HyperNet
    from caphn.engine import FusedTrainer
    from caphn import _lib
    from caphn import ops as ops_mod
    for kv in list(args.tune) + [t for t in os.environ.get("CAPHN_TUNE", "").split(",") if t]:
        k, v = kv.split("=")
        assert _lib.load().caphn_tune(int(k), int(v)) == 0
    if args.dtype != "f32":
        assert _lib.load().caphn_tune(11, {"bf16": 1, "bf16x2": 2}[args.dtype]) == 0

    B, T, P, D, F, E, H, V = args.batch, 20, 49, 2048, 200, 200, 200, 9684
    torch.manual_seed(1234)                       # identical replicas on every rank
    net = HyperNet(F, E, H, V, _Vocab(), cell=args.cell).to(dev)
    tr = FusedTrainer(net, lr=1e-3, max_norm=5.0)
    if os.environ.get("CAPHN_OVERLAP_LEVEL"):
        tr.overlap_level = int(os.environ["CAPHN_OVERLAP_LEVEL"])
    if os.environ.get("CAPHN_OVERLAP_AFTER_HEAD"):
        tr.overlap_after_head = int(os.environ["CAPHN_OVERLAP_AFTER_HEAD"])
    batches = synth_batches(4, B, T, P, D, V, dev, seed=1234 + rank)
    # one style domain per rank-batch; it rotates over the three domains step by step (every rank starts at its own), so the
    # next-theta GEMV fused into the Adam pass and the style row's embedding gradient see a changing row
    def style_at(j):
        return 4 + ((rank + (0 if args.fixed_style else j)) % 3)
    style = style_at(0)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    class HostFeed:
        """--from-host: batch j lives in pinned host memory and is copied into device slot j % 3 on a copy stream while
        step j-2 .. j-1 run (step j uses slot j % 3 and, for the next-batch precompute, slot (j+1) % 3)."""

        def __init__(self, dev_batches):
            self.host = [(f.cpu().pin_memory(), c.cpu().pin_memory()) for f, c in dev_batches]
            self.slots = [(torch.empty_like(dev_batches[0][0]), torch.empty_like(dev_batches[0][1])) for _ in range(3)]
            self.copy = torch.cuda.Stream(device=dev)
            self.copied = [torch.cuda.Event() for _ in range(3)]
            self.done = [None, None, None]
            self.next_j = 0

        def _issue(self, j):
            s = j % 3
            with torch.cuda.stream(self.copy):
                if self.done[s] is not None:
                    self.copy.wait_event(self.done[s])          # the step that last read this slot has finished
                hf, hc = self.host[j % len(self.host)]
                self.slots[s][0].copy_(hf, non_blocking=True)
                self.slots[s][1].copy_(hc, non_blocking=True)
                self.copied[s].record(self.copy)

        def get(self, j):
            while self.next_j <= j + 1:                          # batches j and j+1 are on their way
                self._issue(self.next_j)
                self.next_j += 1
            main = torch.cuda.current_stream()
            main.wait_event(self.copied[j % 3])
            main.wait_event(self.copied[(j + 1) % 3])
            return self.slots[j % 3]

        def after(self, j):
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream())
            self.done[j % 3] = ev
            self._issue(j + 2)                                   # slot (j+2) % 3 == (j-1) % 3: free once step j-1 is done
            self.next_j = max(self.next_j, j + 3)

    feed = HostFeed(batches) if args.from_host else None

    def batch_at(j):
        return feed.get(j) if feed else batches[j % len(batches)]

    def after_step(j):
        if feed:
            feed.after(j)

    use_graph = (world == 1) and args.graph
    if use_graph or args.no_prefetch:
        do_step = tr.step_graphed if use_graph else tr.step
    else:
        # the loader is one batch ahead, so the next batch's style is known when the optimiser runs
        seq = feed.slots if feed else batches
        nxt = {seq[i][0].data_ptr(): seq[(i + 1) % len(seq)] for i in range(len(seq))}

        def do_step(f, c, style_token, next_style=None):
            nf, nc = (None, None) if args.no_overlap else nxt[f.data_ptr()]
            return tr.step(f, c, style_token=style_token, next_style_token=style_token if next_style is None else next_style,
                           next_features=nf, next_captions=nc)
    if use_graph:                                  # two passes over the batch buffers: eager, then capture
        for _ in range(2):
            for f, c in batches:
                do_step(f, c, style_token=style)
    rot = not (use_graph or args.no_prefetch)      # (those two step functions take no announced next style)

    def run_step(j):
        f, c = batch_at(j)
        if rot:
            out = do_step(f, c, style_token=style_at(j), next_style=style_at(j + 1))
        else:
            out = do_step(f, c, style_token=style)
        after_step(j)
        return out
    for i in range(args.warmup):
        run_step(i)
    # Spin-up (untimed, additional to --warmup; reported as config.spinup_steps).  The first process on a freshly
    # acquired box stalls ONCE on the host for ~37 ms around its 1200th kernel launch (step 12-13 here; the GPU idles,
    # every kernel keeps its normal duration, a second process on the same box never shows it --
    # tools/first_steps.py).  Inside a 50-step timed region that one stall reads as 2.8 instead of 2.05 ms/step.
    # Keep stepping in windows of 20 until two consecutive windows agree within 2 % and the last one is within 3 % of
    # the fastest seen, or 4 s have passed.
    spin = {"steps": 0}
    if args.pretouch > 0:
        tiny = torch.zeros(4, dtype=torch.float32, device=dev)
        t_pt = time.perf_counter()
        worst = 0.0
        for j in range(args.pretouch):
            t1 = time.perf_counter()
            ops_mod.zero_(tiny)
            worst = max(worst, time.perf_counter() - t1)
        torch.cuda.synchronize()
        if rank == 0:
            print(f"[pretouch] {args.pretouch} launches in {(time.perf_counter() - t_pt) * 1e3:.1f} ms, slowest single call {worst * 1e3:.2f} ms", file=sys.stderr)
    if not args.no_spinup:
        def window(n=20):
            torch.cuda.synchronize()
            t = time.perf_counter()
            for j in range(n):
                run_step(args.warmup + spin["steps"] + j)
            torch.cuda.synchronize()
            spin["steps"] += n
            dtw = (time.perf_counter() - t) / n
            if rank == 0:
                print(f"[spin-up] window of {n}: {dtw * 1e3:.3f} ms/step", file=sys.stderr)
            return dtw
        t_spin = time.perf_counter()
        prev = window(); best = prev
        while True:
            cur = window(); best = min(best, cur)
            done = (abs(cur - prev) <= 0.02 * prev and cur <= 1.03 * best) or time.perf_counter() - t_spin >= 4.0
            if world > 1:       # steps are collective: every rank must take the same number of them
                flag = torch.tensor([1 if done else 0], device=dev, dtype=torch.int32)
                dist.all_reduce(flag, op=dist.ReduceOp.MIN)
                done = bool(int(flag))
            if done:
                break
            prev = cur
    # dominant-kernel timing: HIP events on the launch stream around the adam_rank launch of head 0
    from caphn import ops
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    orig = ops.adam_rank
    cnt = {"i": 0}

    # (timing events are not free: each pair costs the launch stream ~40 us of queue bubble around the kernel, 2 % of a
    #  step -- so every 4th launch is timed, at least 5 of them)
    every = 4 if args.steps >= 20 else 1
    cnt["seen"] = 0

    def timed_adam_rank(W, *a, **k):
        if W.shape[0] * W.shape[1] >= 100_000_000 and cnt["i"] < len(ev):
            cnt["seen"] += 1
            if cnt["seen"] % every == 0:
                ev[cnt["i"]][0].record()
                orig(W, *a, **k)
                ev[cnt["i"]][1].record()
                cnt["i"] += 1
                return
        orig(W, *a, **k)
    ops.adam_rank = timed_adam_rank

    barrier()
    t0 = time.perf_counter()
    off = args.warmup + spin["steps"]          # continue the batch cycle, so the announced next batch is the one that comes
    for i in range(args.steps):
        loss = run_step(off + i)
    t_host = time.perf_counter() - t0          # the host has ENQUEUED the timed steps (how far it runs ahead of the GPU)
    barrier()
    dt = time.perf_counter() - t0
    ops.adam_rank = orig
    if world > 1:
        tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax)

    final_loss = float(loss[0])
    if dist.is_initialized() and (world > 1 or forced):
        # per-collective device time: four extra steps OUTSIDE the timed region (every rank takes them) with every collective
        # waited for at once between two events on the communication stream
        tr.time_collectives = True
        for i in range(4):
            run_step(off + args.steps + i)
        tr.time_collectives = False
        barrier()
    coll = tr.collective_report()

    # The dominant kernel ALONE on the chip, same box, same process: in the timed region the theta-independent front of the next
    # forward runs beside it (FusedTrainer.overlap_level 4: a shorter step, a longer pass); eight extra steps OUTSIDE the timed
    # region with that front forked behind the pass instead (overlap_level 1) time it undisturbed -- reported as roofline.alone,
    # never as roofline.frac.  (Every rank takes the steps: they are collective.)
    alone_ms = None
    if tr.overlap_level >= 4 and not (use_graph or args.no_prefetch or args.no_overlap):
        lvl = tr.overlap_level
        tr.overlap_level = 1
        ev2 = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(8)]
        c2 = {"i": -4}

        def timed2(W, *a, **k):
            if W.shape[0] * W.shape[1] >= 100_000_000:
                c2["i"] += 1
                if 0 < c2["i"] <= len(ev2):
                    ev2[c2["i"] - 1][0].record()
                    orig(W, *a, **k)
                    ev2[c2["i"] - 1][1].record()
                    return
            orig(W, *a, **k)
        ops.adam_rank = timed2
        base = off + args.steps + 4
        for i in range(4 + len(ev2)):
            run_step(base + i)
        barrier()
        ops.adam_rank = orig
        tr.overlap_level = lvl
        run_step(base + 4 + len(ev2))        # (back in the default schedule for whatever follows)
        barrier()
        alone_ms = float(np.mean([a.elapsed_time(b) for a, b in ev2]))

    if args.phases and rank == 0:
        phase_report(tr, batches, style)

    if rank == 0:
        kts = [a.elapsed_time(b) for a, b in ev[:cnt["i"]]]
        kern_ms = float(np.mean(kts)) if kts else float("nan")
        k0, w0 = tr.shape.heads[0]
        kbytes = 24.0 * k0 * w0
        achieved = kbytes / (kern_ms * 1e-3) / 1e9 if cnt["i"] else None
        # HBM bytes of one launch of that kernel from the PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE,
        # gfx950 corrections applied) -- collected separately, summary committed under profiles/
        traffic = None
        # (the file names the kernel symbol it was collected on: a figure for another instantiation is not reported)
        try:
            pmc = json.load(open(os.path.join(ROOT, "profiles", "r03_pmc_adam_rank.json")))
            if int(pmc["algorithmic_bytes"]) == int(kbytes) and pmc.get("kernel", "").startswith(DOMINANT_KERNEL_SYMBOL):
                traffic = pmc["traffic_bytes"]
        except Exception:  # noqa: BLE001
            pass
        ceiling = copy_ceiling(dev) if B == 128 else None
        ms_step = dt / args.steps * 1e3
        line = {
            "metric": METRIC, "value": B * world * args.steps / dt, "unit": "images/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_step,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype,
            "data": "synthetic" + (" (each minibatch copied from pinned host memory inside the timed region)" if args.from_host else ""),
            "config": {"workload": f"Flickr30k-shaped {args.cell.upper()}+additive-attention decoder + hypernet (3 style domains), "
                                   "full training step (fwd, CE, bwd, clip 5.0, Adam)" +
                                   (" -- SIDE MEASUREMENT: single-product bf16 contractions, not the metric's fp32 arithmetic" if args.dtype == "bf16" else
                                    " -- SIDE MEASUREMENT: two-plane bf16 contractions (3 products), not the metric's fp32 arithmetic" if args.dtype == "bf16x2" else ""),
                       "per_gpu_batch": B, "global_batch": B * world, "T": T, "P": P, "D": D, "F": F, "E": E, "H": H,
                       "V": V, "hypernet_params": int(sum(q.numel() for q in net.hn_base.parameters()) +
                                                      sum(q.numel() for q in net.hn_heads.parameters())),
                       "parallelism": f"dp{world}" + (" (REHEARSAL: all ranks on one GPU over gloo)" if rehearsal else "") +
                                      (" (one-rank RCCL group, collectives forced)" if forced else ""), "launch": "hipGraph" if use_graph else "eager",
                       "next_theta_in_adam_pass": not (use_graph or args.no_prefetch),
                       "next_precompute_beside_adam": not (use_graph or args.no_prefetch or args.no_overlap),
                       "spinup_steps": spin["steps"], "host_enqueue_ms_per_step": round(t_host / args.steps * 1e3, 4), "cpu_affinity": numa,
                       "final_loss": final_loss},
            "roofline": {"bound": "hbm", "kernel": "adam_rank_kernel(hn_heads.0.2.weight)",
                         "achieved": achieved, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                         "frac": (achieved / (HBM_PEAK / 1e9)) if achieved else None, "traffic": traffic,
                         "kernel_launches_timed": len(kts), "kernel_ms": kern_ms, "kernel_ms_median": float(np.median(kts)) if kts else None,
                         "kernel_ms_min": float(np.min(kts)) if kts else None, "algorithmic_bytes": kbytes,
                         "copy_ceiling_gbps": ceiling, "frac_of_copy_ceiling": (achieved / ceiling) if (achieved and ceiling) else None,
                         "step_frac": STEP_ALGO_BYTES / (ms_step * 1e-3) / HBM_PEAK if B == 128 else None},
        }
        if alone_ms:
            line["roofline"]["beside"] = ("in the timed region the next forward's feature_fc / init_hidden / W_a f GEMMs run beside this kernel "
                                          "(FusedTrainer.overlap_level 4); `alone` = the same launch with that front forked behind it, 8 extra "
                                          "steps outside the timed region")
            line["roofline"]["alone"] = {"kernel_ms": alone_ms, "achieved": kbytes / (alone_ms * 1e-3) / 1e9,
                                         "frac": kbytes / (alone_ms * 1e-3) / HBM_PEAK,
                                         "frac_of_copy_ceiling": (kbytes / (alone_ms * 1e-3) / 1e9 / ceiling) if ceiling else None}
        line["collectives"] = coll
        if world == 1 and not forced and not args.no_module_api and args.cell == "gru" and args.dtype == "f32":
            line["module_api"] = module_api_loop(dev, (B, T, P, D, F, E, H, V), batches)
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(B, T, P)
        sys.stdout.flush()
        os.dup2(real_stdout, 1)
        print(json.dumps(line), flush=True)
        os.dup2(2, 1)
    if dist.is_initialized():
        dist.destroy_process_group()


def phase_report(tr, batches, style):
    """Coarse per-phase GPU times (events around the composite calls) -- a tuning aid, stderr only."""
    from caphn import ops
    names = ["hyper_forward", "decoder_forward", "cross_entropy_fwd_bwd", "decoder_backward", "hyper_backward",
             "sumsq_partials", "rank_sumsq", "clip_coef", "adam_dense", "adam_rank"]
    acc = {n: [] for n in names}
    origs = {n: getattr(ops, n) for n in names}

    def wrap(n):
        def f(*a, **k):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record(); r = origs[n](*a, **k); e.record()
            acc[n].append((s, e))
            return r
        return f
    for n in names:
        setattr(ops, n, wrap(n))
    t_all = []
    for i in range(5):
        f, c = batches[i % len(batches)]
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); tr.step(f, c, style_token=style); e.record()
        t_all.append((s, e))
    torch.cuda.synchronize()
    for n in names:
        setattr(ops, n, origs[n])
    tot = np.mean([a.elapsed_time(b) for a, b in t_all])
    print(f"[phases] step {tot:.3f} ms", file=sys.stderr)
    for n in names:
        ms = sum(a.elapsed_time(b) for a, b in acc[n]) / 5
        print(f"[phases]   {n:24s} {ms:8.3f} ms", file=sys.stderr)


if __name__ == "__main__":
    main()
