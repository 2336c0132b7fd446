#!/usr/bin/env python
"""Headline benchmark of the HIP hot path: PLE 3-domain training step (BASELINE.json configs[1]).

    python bench.py --gpus N --steps K --warmup W          (N > 1: launched by torch.distributed.run, one rank per GPU)

One step = one full training step of run.py:481-493 (forward, BCE, L2 term over every registered tensor incl. the
whole embedding table, backward, Adam on every parameter) on one resident synthetic batch.  Prints ONE JSON line.
`roofline` is for the dominant kernel of the step, timed with HIP events on the launch stream; `cpu_baseline` is the
CPU oracle (reference semantics, stock torch CPU ops) timed on this host's cores on a bounded sample (rank 0, N=1).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

HBM_PEAK_GBPS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
MFMA_BF16_PEAK_TFLOPS = 2500.0  # dense bf16 MFMA peak
METRIC = "train samples/sec + AUC parity, PLE 3-domain batch 4096 at 1/2/4/8 MI355X"


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20, help="untimed steps right before the timed region")
    ap.add_argument("--preroll", type=int, default=-1,
                    help="untimed steps run BEFORE --warmup, whatever --warmup is, so that the timed region is in steady state: the "
                         "lazy table needs flush_every steps until every slice owes a whole period of replay, and the Adam bias "
                         "corrections move for ~1700 steps (their table look-ups stop afterwards).  -1 = flush_every + the length of "
                         "the step-scalar table + 64")
    ap.add_argument("--batch", type=int, default=4096, help="per-GPU batch (weak scaling)")
    ap.add_argument("--fields", type=int, default=26)
    ap.add_argument("--vocab", type=int, default=1_000_000)
    ap.add_argument("--embed-dim", type=int, default=16)
    ap.add_argument("--dropout", type=float, default=0.2, help="reference default (ple.py:17)")
    ap.add_argument("--precision", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--table-mode", default="lazy", choices=["dense", "lazy"])
    ap.add_argument("--graph", type=int, default=1, help="replay the step as one hipGraph (single GPU)")
    ap.add_argument("--id-dist", default="uniform", choices=["uniform", "zipf"])
    ap.add_argument("--cpu-baseline", type=int, default=1)
    ap.add_argument("--cpu-steps", type=int, default=4)
    ap.add_argument("--sync-bn", type=int, default=1,
                    help="N>1: 1 (default, the parity semantic of SURVEY 8e) = BatchNorm statistics over the global batch, bit for bit "
                         "the single-process step on the concatenated batch; 0 = per-rank statistics over the local rows.  With 1 "
                         "the per-rank-statistics throughput is measured as well and reported as `value_per_rank_bn`")
    ap.add_argument("--table-dist", default=None, choices=["sharded", "replicated"], help="N>1: default sharded (lazy table)")
    ap.add_argument("--simulate-world", type=int, default=0,
                    help="diagnostic, single process: run ONE rank's launch work of a W-rank step with the collectives replaced "
                         "by local copies (timing of the compute side only; the numbers it trains on are meaningless)")
    ap.add_argument("--atten", type=int, default=0,
                    help="1: the reference's default attention branch (config.py:24-28: 3 x MultiheadAttention(64, 2 heads) + residual) "
                         "on top of the BASELINE configuration, which is defined without it")
    ap.add_argument("--flush-every", type=int, default=64, help="lazy table: the whole table is replayed once per this many steps")
    ap.add_argument("--fast-replay", type=int, default=-1, help="lazy table: 1 scaled-state replay of untouched rows (hardware rcp / sqrt), 0 the exact recurrence; -1 = FusedAdam's default")
    ap.add_argument("--overlap-waves", type=int, default=2, help="waves per SIMD of the background replay slice's capped grid (A/B only)")
    ap.add_argument("--rows-dense-one-launch", type=int, default=1, help="0: the step's row update and the dense Adam as two launches (A/B only)")
    ap.add_argument("--tower-one-launch", type=int, default=1, help="0: the fused towers as two launches, forward and backward (A/B only)")
    ap.add_argument("--fuse-towers", type=int, default=1, help="0: the towers as the five launches per direction the fused launch replaces (A/B only)")
    ap.add_argument("--pool", type=int, default=0,
                    help="resident synthetic batches cycled through; 0 = warmup+steps (max 1024), so that no batch repeats and "
                         "the lazy table replay sees realistic gaps between two look-ups of a row")
    return ap.parse_args()


def build_model(args, device):
    from cdcmdr_amd.model.ple import PLE
    torch.manual_seed(2000)
    field_dims = [args.vocab] * args.fields
    cfg = None
    if args.atten:
        import types
        cfg = types.SimpleNamespace(use_atten=True, atten_embed_dim=64, att_layer_num=3, att_head_num=2, att_res=True, use_dcn=False)
    with torch.device(device):
        model = PLE(field_dims, args.embed_dim, 3, 2, 2, ((256, 128), (64,)), (64, 32), args.dropout, cfg)
    model.set_precision(args.precision)
    return model, field_dims


def cpu_baseline(args, model, field_dims, Xc, yc, gc):
    """The oracle restatement of the reference step on the host cores: dense table gradient, whole-table L2,
    dense torch.optim.Adam — exactly what run.py:481-493 makes torch do."""
    from oracle import cdc_oracle as O
    # the cores this process may actually run on (a container's CPU share), not the host's core count
    try:
        n_threads = len(os.sched_getaffinity(0))
    except AttributeError:
        n_threads = os.cpu_count() or 1
    n_threads = max(1, min(n_threads, int(os.environ.get("CDC_CPU_THREADS", "64"))))
    torch.set_num_threads(n_threads)
    sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    leaves = {k: v.requires_grad_(True) for k, v in sd.items() if v.dtype.is_floating_point and "running_" not in k}
    sd.update(leaves)
    l2 = {n: 1e-5 for n in O.reg_names(list(sd), "ple")}
    opt = torch.optim.Adam(list(leaves.values()), lr=1e-3, betas=(0.9, 0.99), eps=1e-8, weight_decay=1e-8)
    O.DROPOUT_P = args.dropout
    B = args.batch
    times = []
    t_begin = time.perf_counter()
    for s in range(1 + args.cpu_steps):
        if s >= 2 and time.perf_counter() - t_begin > 40.0:      # bounded sample: stop after ~40 s of CPU work
            break
        x = Xc[s % len(Xc)]
        y = torch.from_numpy(yc[s % len(yc)])
        g = torch.from_numpy(gc[s % len(gc)]).reshape(-1, 1)
        t0 = time.perf_counter()
        stats = {}
        p = O.ple_forward(sd, x, field_dims, 3, training=True, stats_out=stats).gather(1, g).squeeze(1)
        loss = O.bce_mean(p, y) + O.reg_loss(sd, l2)
        opt.zero_grad()
        loss.sum().backward()
        opt.step()
        float(loss.sum().detach())                           # loss.item() as run.py:493
        for k, v in stats.items():
            sd[k] = v
        times.append(time.perf_counter() - t0)
    O.DROPOUT_P = 0.0
    t = float(np.mean(times[1:]))
    return {"value": B / t, "unit": "samples/s", "cores": n_threads, "kind": "port",
            "sample": f"{len(times) - 1} steps of batch {B} after 1 warm-up, same shapes, weights copied from the GPU model "
                      f"({t * 1e3:.0f} ms/step)"}


class Loopback:
    """--simulate-world: the DataParallel interface with every collective a local copy."""

    capturable = True                 # local copies: the whole step is captured as ONE graph, as over RCCL

    def __init__(self, world):
        self.world_size, self.rank, self.backend = world, 0, "loopback"
        self._on_close = []

    def all_reduce_sum(self, t):
        return t

    all_reduce_max = all_reduce_sum

    def all_to_all(self, out, inp):
        if inp.dtype == torch.int32:     # row ids: fold them onto rows this rank owns, so that its rows see the look-up
            w = self.world_size          # rate (and the ownership-filtered flush) of a real W-rank run
            out.copy_(torch.where(inp >= 0, inp // w * w, inp))
        else:
            out.copy_(inp)
        return out

    def all_to_all_start(self, out, inp):
        self.all_to_all(out, inp)
        return None

    @staticmethod
    def wait(handle):
        pass

    def all_gather_rows(self, out, local):
        out.view(self.world_size, *local.shape).copy_(local.unsqueeze(0).expand(self.world_size, *local.shape))
        return out

    def barrier(self):
        pass

    def close(self):
        pass


def self_launch(args):
    """--gpus N > 1 without a launcher around us: start the N ranks ourselves (torch.distributed.run as a CHILD process,
    before this process has touched a GPU — no re-exec) and relay rank 0's JSON line."""
    import socket
    import subprocess
    rehearsal = os.environ.get("CDC_BENCH_REHEARSAL") == "1"
    have = torch.cuda.device_count()            # counts devices without initialising the runtime
    if not rehearsal and have < args.gpus:
        print(f"bench.py: --gpus {args.gpus} but this node shows {have} GPU(s)", file=sys.stderr)
        return 2
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env["CDC_BENCH_CHILD"] = "1"
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for out in proc.stdout:
        if out.lstrip().startswith("{") and '"metric"' in out:
            line = out.strip()
        else:
            sys.stderr.write(out)
    rc = proc.wait()
    if line is not None:
        print(line, flush=True)
    return rc if rc else (0 if line is not None else 1)


def replay_depth(opt):
    """Mean number of L2-only steps the NEXT slice flush replays per row (host-synchronising; called outside the timed
    region).  In steady state this is flush_every for rows not looked up since their slice's last flush and less for the
    others; right after start-up (last[] = 0 everywhere) it is only the step count — the state BENCH_r01 was timed in."""
    if opt.table_mode != "lazy" or opt.flush_every <= 1:
        return None
    t = int(opt.step_dev.item())                 # the next step is t+1; its slice flush brings slice (t mod P) up to step t
    R, P = opt.table.shape[0], opt.flush_every
    rps = -(-R // P)
    lo = (t % P) * rps
    hi = min(lo + rps, R)
    if lo >= hi:
        return None
    last = opt.table_last[lo:hi]
    if opt.own_mod > 1:
        rows = torch.arange(lo, hi, device=last.device)
        last = last[rows % opt.own_mod == opt.own_rem]
    return float((t - last.double()).clamp_(min=0).mean().item())


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world == 1 and args.gpus > 1 and os.environ.get("CDC_BENCH_CHILD") != "1":
        raise SystemExit(self_launch(args))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    from cdcmdr_amd.dist import DataParallel
    from cdcmdr_amd.optim import FusedAdam
    from cdcmdr_amd.synth import make_dataset
    from cdcmdr_amd.trainer import TrainStep
    # CDC_BENCH_REHEARSAL=1: all ranks on cuda:0 with gloo (host-staged) collectives — exercises the N>1 code path of this
    # script on a one-GPU box; the throughput it prints is meaningless
    rehearsal = os.environ.get("CDC_BENCH_REHEARSAL") == "1"
    forced = bool(int(os.environ.get("CDC_FORCE_COLLECTIVES", "0")))      # the RCCL call path with a single rank
    dp = (DataParallel(backend="gloo") if rehearsal else DataParallel()) if (world > 1 or forced) else None
    sim = None
    if args.simulate_world > 1 and world == 1:
        sim = Loopback(args.simulate_world)
    local_rank = 0 if rehearsal else int(os.environ.get("LOCAL_RANK", "0"))
    rank = int(os.environ.get("RANK", "0"))
    device = torch.device("cuda", local_rank)
    torch.cuda.set_device(device)

    if not args.fuse_towers:
        from cdcmdr_amd import plan as _plan
        _plan.TowerChain.enabled = False
    model, field_dims = build_model(args, device)
    table_mode = args.table_mode
    use_graph = bool(args.graph)        # under DP the launch stages between the collectives are graphs
    opt_kw = {} if args.fast_replay < 0 else {"fast_replay": bool(args.fast_replay)}
    opt = FusedAdam(model, lr=1e-3, betas=(0.9, 0.99), eps=1e-8, weight_decay=1e-8, table_mode=table_mode,
                    flush_every=args.flush_every, **opt_kw)
    dist_obj = dp or sim
    ts = TrainStep(model, opt, args.batch, mode="multi", use_graph=use_graph, dist=dist_obj, sync_bn=bool(args.sync_bn),
                   table_dist=args.table_dist, tower_one_launch=bool(args.tower_one_launch), rows_dense_one_launch=bool(args.rows_dense_one_launch), overlap_waves=args.overlap_waves)
    # N>1: the headline runs with global-batch BatchNorm statistics (the parity semantic); the per-rank-statistics step
    # (torch DDP without SyncBatchNorm) is timed beside it on the same model and optimiser state
    ts_local = None
    if dp is not None and world > 1 and args.sync_bn:
        ts_local = TrainStep(model, opt, args.batch, mode="multi", use_graph=use_graph, dist=dp, sync_bn=False,
                             table_dist=args.table_dist)

    B = args.batch
    if args.preroll < 0:
        args.preroll = (args.flush_every + int(opt.scalars.shape[0]) + 64) if table_mode == "lazy" else 8
    if args.pool <= 0:
        args.pool = max(8, min(args.preroll + args.warmup + args.steps, 1024))
    # every rank draws its own shard of the synthetic stream (same generator, rank-offset seed)
    Xr, yr = make_dataset(B * args.pool, field_dims, n_domain=3, domain_idx=10, seed=2000 + rank, dist=args.id_dist)
    gr = Xr[:, 10].astype(np.int64)                            # identity domain -> tower map (3-domain PLE)
    Xd = torch.from_numpy(Xr).to(device).view(args.pool, B, -1)
    yd = torch.from_numpy(yr).to(device).view(args.pool, B)
    gd = torch.from_numpy(gr).to(device).view(args.pool, B)
    cursor = [0]

    def run(step_obj, n):
        for _ in range(n):
            j = cursor[0] % args.pool
            cursor[0] += 1
            # the next batch is named like a data loader with one batch of look-ahead does (data.train_epoch): on one GPU its row
            # sort runs beside this step's forward/backward — every step still sorts exactly one batch
            step_obj.step(Xd[j], yd[j], gd[j], next_X=Xd[cursor[0] % args.pool])

    def timed(step_obj, n):
        if dp:
            dp.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run(step_obj, n)
        torch.cuda.synchronize()
        if dp:
            dp.barrier()
        el = time.perf_counter() - t0
        if dp:
            tmax = torch.tensor([el], dtype=torch.float64, device=device)
            dp.all_reduce_max(tmax)
            el = float(tmax.item())
        return el

    run(ts, args.preroll)                        # untimed, independent of --warmup: reach the steady state
    run(ts, max(args.warmup, 1))
    elapsed = timed(ts, args.steps)
    # (host-synchronising and the first use of a few torch kernels: behind the timed region — in front of the warm-up it left the
    # first ~100 steps 3 % slower, which is what a --steps 20 --warmup 5 run then measured; the replay depth is a periodic function
    # of the step count in steady state)
    depth = replay_depth(opt)
    if os.environ.get("CDC_BENCH_REPEAT"):                     # development: the same bracket again (is the first one special?)
        print("brackets ms/step:", [round(elapsed / args.steps * 1e3, 4)] + [round(timed(ts, args.steps) / args.steps * 1e3, 4) for _ in range(4)], file=sys.stderr)
    loss_val = float(ts.loss.item())
    elapsed_local = None
    if ts_local is not None:
        run(ts_local, max(args.warmup, 4))
        elapsed_local = timed(ts_local, args.steps)

    # ---- roofline of the dominant kernel, HIP events on the launch stream (instrumented eager steps) ----------
    roof = measure_roofline(args, ts, opt, Xd, yd, gd, depth, elapsed / args.steps * 1e3)

    cpu = None
    if args.cpu_baseline and world == 1 and rank == 0 and sim is None:
        Xc = [Xr[i * B:(i + 1) * B] for i in range(min(args.pool, 1 + args.cpu_steps))]
        yc = [yr[i * B:(i + 1) * B].astype(np.float32) for i in range(len(Xc))]
        gc = [gr[i * B:(i + 1) * B] for i in range(len(Xc))]
        opt.flush_table()
        cpu = cpu_baseline(args, model, field_dims, Xc, yc, gc)

    if sim is not None:
        top = sorted(roof["breakdown_all"].items(), key=lambda kv: -kv[1])
        print(json.dumps({"simulated_world": sim.world_size, "table_dist": ts.table_dist, "sync_bn": bool(args.sync_bn),
                          "one_rank_compute_ms_per_step": elapsed / args.steps * 1e3,
                          "kernel_ms_per_step_sum": roof["kernel_ms_per_step_sum"], "launch_ms_per_step": dict(top)}), flush=True)
        return
    if os.environ.get("CDC_BENCH_BREAKDOWN_ALL") != "1":
        roof.pop("breakdown_all", None)
    if rank == 0:
        ms = elapsed / args.steps * 1e3
        steady = table_mode != "lazy" or args.preroll >= args.flush_every + int(opt.scalars.shape[0])
        out = {
            "metric": METRIC, "value": B * world * args.steps / elapsed, "unit": "samples/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.precision, "data": "synthetic",
            "config": {"workload": "PLE 3-domain full training step (fwd + BCE + whole-table L2 + bwd + Adam), "
                                   f"{args.fields} fields x vocab {args.vocab}, emb_dim={args.embed_dim}, batch {B}/GPU",
                       "global_batch": B * world, "dropout": args.dropout, "table_mode": table_mode,
                       "hip_graph": use_graph, "row_sort_one_batch_ahead": bool(getattr(ts, "_ahead_ok", False)), "id_dist": args.id_dist, "parallelism": f"dp{world}", "attention_branch": bool(args.atten),
                       "table_dist": ts.table_dist if world > 1 else None,
                       # N > 1: how the step's collectives are issued — "one_graph": captured with the launches in one hipGraph per
                       # step (RCCL); "segments": launch segments replayed, collectives issued by the host in between (fallback)
                       "exchange": None if not ts.dp_on else ("rccl, one_graph" if ts._one_graph_ok else "rccl, segments"),
                       "bn_stats": None if world == 1 else ("global batch (sync)" if args.sync_bn else "per rank"),
                       "steady_state": bool(steady), "preroll_steps": args.preroll,
                       "mean_replay_depth_of_next_slice_flush": depth, "flush_every": args.flush_every if table_mode == "lazy" else None,
                       "last_bce_loss": loss_val},
            "roofline": roof, "cpu_baseline": cpu,
        }
        if elapsed_local is not None:
            out["value_per_rank_bn"] = B * world * args.steps / elapsed_local
            out["ms_per_step_per_rank_bn"] = elapsed_local / args.steps * 1e3
        print(json.dumps(out), flush=True)
    if dp:
        dp.close()


# C-ABI call -> kernel name in the rocprofv3 summaries
_KERNEL_OF = {"cdc_embed_lazy_flush(slice)": "k_lazy_flush", "cdc_embed_adam_dense_pass": "k_adam_dense_pass",
              "cdc_glinear_fwd": "k_g2_nt", "cdc_glinear_bwd_x": "k_g2_nt", "cdc_glinear_pair_fwd": "k_pair_fwd",
              "cdc_embed_gather_fwd": "k_gather_fwd"}


def profiled_traffic(call_name):
    """HBM bytes per launch of the dominant kernel.  NOT measured in this run (counters need the profiler around the
    process): read from the newest committed PMC summary of this same command (profiles/roundN/pmc_default_fetch_write.txt:
    separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, KB per dispatch; gfx950 reports half of a wide coalesced read,
    hence fetch x 2).  The source file is named in the line so that a stale figure can be told from a fresh one."""
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles")
    kern = _KERNEL_OF.get(call_name)
    if kern is None or not os.path.isdir(root):
        return None, None
    for rnd in sorted((d for d in os.listdir(root) if d.startswith("round")), reverse=True):
        path = os.path.join(root, rnd, "pmc_default_fetch_write.txt")
        if not os.path.exists(path):
            continue
        fetch = write = None
        cur = None
        for line in open(path):
            if not line.startswith(" "):
                cur = line
            elif cur is not None and kern in cur:
                parts = line.split()
                if parts[0] == "FETCH_SIZE":
                    fetch = float(parts[1])
                elif parts[0] == "WRITE_SIZE":
                    write = float(parts[1])
        if fetch is not None and write is not None:
            return fetch * 1024 * 2 + write * 1024, f"profiles/{rnd}/pmc_default_fetch_write.txt (committed rocprofv3 --pmc passes, per dispatch; not live)"
    return None, None


def event_pair_floor_ms(ts, n=64):
    """median HIP-event time around a one-element cdc_fill_f32 launch, taken exactly like the per-launch times of ts.profile()"""
    import ctypes as C
    from cdcmdr_amd import _lib as L
    buf = torch.zeros(4, dtype=torch.float32, device=ts.device)
    rec = []
    L.PROFILE = rec
    try:
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        for _ in range(n):
            L.launch("floor", ts.lib.cdc_fill_f32, (buf.data_ptr(), 0.0, 1), st)
        torch.cuda.synchronize()
        times = sorted(e0.elapsed_time(e1) for _, e0, e1, _, _ in rec)
    finally:
        L.PROFILE = None
    return times[len(times) // 2]


def back_to_back(ts, reps=100):
    """Every launch of the plan's forward and backward sequences issued `reps` times in a row on the idle chip, ONE event pair around
    the run: microseconds per launch without an event pair's own latency in it (tools/step_probe.py's method; what
    profiles/roundN/asymptote.txt lists).  Returns [(name, us, flops)].  The launches are re-runs of the last step's (idempotent
    up to the BatchNorm running statistics, which are put back afterwards)."""
    import ctypes as C
    from cdcmdr_amd import _lib as L
    plan = ts.plan
    keep = {k: v.clone() for k, v in ts.model.state_dict().items() if "running_" in k or "num_batches" in k}
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    out = []
    try:
        for fn in list(plan.fwd_steps) + list(plan.bwd_steps):
            if getattr(fn, "is_comm", False):
                continue
            rec = []
            L.PROFILE = rec
            fn(st)
            torch.cuda.synchronize()
            L.PROFILE = None
            if not rec:
                continue
            name, _, _, fl, _ = rec[0]
            for _ in range(5):
                fn(st)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            for _ in range(reps):
                fn(st)
            e1.record()
            torch.cuda.synchronize()
            out.append((name, e0.elapsed_time(e1) * 1e3 / reps, fl))
    finally:
        L.PROFILE = None
        sd = ts.model.state_dict()
        for k, v in keep.items():
            sd[k].copy_(v)
    return out


def valu_issue_prices():
    """ns a SIMD needs per wave instruction (packed fp32, transcendental) at the slice's occupancy, from the newest committed
    profiles/roundN/valu_issue_probe.txt (tools/valu_issue_probe.hip: chains of one instruction kind on every SIMD, the whole launch
    timed by events — the shader clock moves with the instruction mix, so the prices are kept in ns, not cycles).  Fallback: the
    guide's one-wave issue costs at 2.4 GHz."""
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles")
    if os.path.isdir(root):
        for rnd in sorted((d for d in os.listdir(root) if d.startswith("round")), reverse=True):
            path = os.path.join(root, rnd, "valu_issue_probe.txt")
            if not os.path.exists(path):
                continue
            ns = {}
            for line in open(path):
                parts = line.split()
                if len(parts) > 3 and parts[1] == "W=8" and "by the launch's event time" in line:
                    ns[parts[0]] = float(line.split(";")[1].split("ns")[0])
            if all(k in ns for k in ("v_pk_fma_f32", "v_pk_mul_f32", "v_sqrt_f32", "v_rcp_f32")):
                return {"pk_fma": ns["v_pk_fma_f32"], "pk_mul": ns["v_pk_mul_f32"], "sqrt": ns["v_sqrt_f32"], "rcp": ns["v_rcp_f32"],
                        "source": f"profiles/{rnd}/valu_issue_probe.txt (8 waves per SIMD, ns per wave instruction and SIMD by the launch's event time)"}
    c = 1.0 / 2.4
    return {"pk_fma": 8 * c, "pk_mul": 8 * c, "sqrt": 8 * c, "rcp": 8 * c, "source": "MI355X_MICROARCH.md one-wave issue costs at 2.4 GHz (no probe table committed)"}


def in_step_figures():
    """what the newest committed rocprofv3 kernel trace of this command says about one REPLAYED step (profiles/roundN/step_timeline.txt):
    durations beside the background slice, which no event inside the timed region can give.  Not measured in this run."""
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles")
    if not os.path.isdir(root):
        return None
    for rnd in sorted((d for d in os.listdir(root) if d.startswith("round")), reverse=True):
        path = os.path.join(root, rnd, "step_timeline.txt")
        if not os.path.exists(path):
            continue
        slice_us = gather_us = None
        gemm_us, tower_us, n_kernels = 0.0, 0.0, 0
        for line in open(path):
            if line.startswith("#"):
                continue
            parts = line.split()
            if len(parts) < 6:
                continue
            dur, name = float(parts[4]), " ".join(parts[5:])
            n_kernels += 1
            if "k_lazy_flush" in name:
                slice_us = dur
            elif "k_gather_fwd" in name:
                gather_us = dur
            if any(t in name for t in ("k_g2_", "k_pair_fwd", "k_cgc_mid", "k_tower")):
                gemm_us += dur
            if "k_tower" in name:
                tower_us += dur
        return {"slice_us": slice_us, "gather_us": gather_us, "contraction_launches_us": gemm_us, "tower_launches_us": tower_us,
                "kernels_per_step": n_kernels, "source": f"profiles/{rnd}/step_timeline.txt", "measured_in_run": False}
    return None


def measure_roofline(args, ts, opt, Xd, yd, gd, depth, ms_per_step):
    """`roofline` = the step's dominant kernel (the lazy table's replay slice) from HIP events around every launch of 64 instrumented
    eager steps, each launch alone on the chip; the north-star figures (contractions against the bf16 MFMA peak, gather against the
    HBM peak) from back-to-back re-runs of each launch (no event latency in them); `roofline.step` = the whole step against its
    algorithmic HBM/MFMA time; the in-step figures of the committed kernel trace ride along, marked as not measured here."""
    batches = [(Xd[i], yd[i], gd[i]) for i in range(len(Xd))]
    prof = ts.profile(batches, n_steps=2 + 64, skip=2, overlap=False)      # 64 steps: exactly one period of the lazy table's whole-table flush
    floor_ms = event_pair_floor_ms(ts)
    floor_net = max(floor_ms - 0.002, 0.0)                                  # the fill launch inside the pair is ~2 us of kernel itself

    def net_ms(v):                                                          # a launch family's ms/step without the event pairs' latency
        return max(v["ms_per_step"] - floor_net * v["launches_per_step"], 1e-9)

    total = sum(v["ms_per_step"] for v in prof.values())
    top = sorted(prof.items(), key=lambda kv: -kv[1]["ms_per_step"])
    name, d = top[0]
    breakdown = {k: round(v["ms_per_step"], 4) for k, v in top[:8]}
    breakdown_all = {k: round(v["ms_per_step"], 4) for k, v in top}
    per_launch_ms = d["ms_per_step"] / max(d["launches_per_step"], 1e-9)
    traffic, traffic_src = profiled_traffic(name)
    if d["flops_per_step"] > 0:
        achieved = d["flops_per_step"] / (net_ms(d) * 1e-3) / 1e12
        roof = {"bound": "mfma", "kernel": name, "achieved": achieved, "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": achieved / MFMA_BF16_PEAK_TFLOPS, "algorithmic_flops_per_step": d["flops_per_step"], "traffic": traffic}
    elif "lazy_flush" in name:
        # The replay slice reads and writes each of its elements once and advances it up to flush_every Adam steps in between: its bound
        # is VALU ISSUE, not HBM.  Per element PAIR and step the scaled replay (csrc/common.h adam_scaled_step_pk) issues 4 v_pk_fma_f32,
        # 1 v_pk_mul_f32, 2 v_sqrt_f32 and 2 v_rcp_f32; a wave instruction covers 64 pairs = 128 element-steps.  The prices are MEASURED
        # (tools/valu_issue_probe.hip; ns per wave instruction and SIMD with the SIMD kept busy).  Counted are the EXECUTED element-steps:
        # a row looked up since its slice's last flush replays fewer than flush_every steps (mean_replay_depth).
        price = valu_issue_prices()
        ns_per_128 = 4 * price["pk_fma"] + price["pk_mul"] + 2 * price["sqrt"] + 2 * price["rcp"]
        es_nominal = opt.table.numel() * 1.0 / max(opt.own_mod, 1)          # every owned element, one step per training step
        share = min(1.0, depth / args.flush_every) if (depth and args.flush_every) else 1.0
        es = es_nominal * share
        peak = 256 * 4 * 128 / ns_per_128                                   # G element-steps/s
        achieved = es / (net_ms(d) * 1e-3) / 1e9
        nbytes = d["bytes_per_step"]
        hbm = nbytes / (net_ms(d) * 1e-3) / 1e9 if nbytes else None
        roof = {"bound": "valu", "kernel": name, "achieved": achieved, "peak": peak, "unit": "G element-steps/s", "frac": achieved / peak,
                "algorithmic_element_steps_per_step": es, "nominal_element_steps_per_step": es_nominal, "executed_share": share,
                "valu_ns_per_128_element_steps_per_simd": ns_per_128, "valu_cycles_per_64_element_steps_at_2p4GHz": ns_per_128 * 2.4 / 2,
                "valu_prices_ns": {k: v for k, v in price.items() if k != "source"}, "valu_prices_source": price["source"],
                "hbm": {"achieved_GBps": hbm, "peak_GBps": HBM_PEAK_GBPS, "frac": None if hbm is None else hbm / HBM_PEAK_GBPS,
                        "algorithmic_bytes_per_step": nbytes, "traffic": traffic, "traffic_source": traffic_src,
                        "measured_in_run": False if traffic is not None else None},
                "traffic": traffic,
                "note": "stand-alone launch (instrumented eager steps).  In the timed region this launch is NOT on the critical path: it runs in "
                        "the background beside the forward/backward (two waves per SIMD, lowest issue priority) — see in_step and step below"}
    else:
        nbytes = d["bytes_per_step"]
        achieved = nbytes / (net_ms(d) * 1e-3) / 1e9 if nbytes else None
        roof = {"bound": "hbm", "kernel": name, "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": None if achieved is None else achieved / HBM_PEAK_GBPS, "algorithmic_bytes_per_step": nbytes, "traffic": traffic}
    roof.update({"avg_launch_ms": per_launch_ms, "launches_per_step": d["launches_per_step"],
                 "traffic_source": traffic_src, "traffic_measured_in_run": False if traffic is not None else None,
                 "event_pair_floor_ms": floor_ms, "avg_launch_ms_less_event_floor": max(per_launch_ms - floor_net, 0.0),
                 "kernel_ms_per_step_sum": total, "breakdown_ms_per_step": breakdown, "breakdown_all": breakdown_all,
                 "timing": "dominant kernel: HIP events around every launch of 64 eager steps, each launch alone on the chip; "
                           "contraction / gather figures: the launch re-issued 100 times back to back, one event pair around the run"})
    # ---- the north star's two figures, from back-to-back re-runs (no event latency inside; = the B = 4096 rows of asymptote.txt)
    b2b = back_to_back(ts)
    roof["back_to_back_us"] = {f"{i:02d} {n}": round(us, 2) for i, (n, us, _) in enumerate(b2b)}
    gm = [(n, us, fl) for n, us, fl in b2b if fl > 0]
    if gm:
        fl = sum(f for _, _, f in gm)
        us = sum(u for _, u, _ in gm)
        roof["all_gemm_tflops"] = fl / (us * 1e-6) / 1e12
        roof["all_gemm_ms_per_step"] = us * 1e-3
        roof["all_gemm_flops_per_step"] = fl
        roof["gemm_frac_of_mfma_peak"] = roof["all_gemm_tflops"] / MFMA_BF16_PEAK_TFLOPS
        big = [(u, f) for n, u, f in gm if "glinear" in n]
        if big:
            roof["grouped_linear_tflops"] = sum(f for _, f in big) / (sum(u for u, _ in big) * 1e-6) / 1e12
    B, F, D = ts.B, ts.emb.F, ts.emb.D
    gb = B * F * (D * 4 + 4 + D * 4)                                        # row read + id + fp32 row written (the bf16 shadow not counted)
    g = [us for n, us, _ in b2b if n == "cdc_embed_gather_fwd"]
    if g:
        roof["gather_GBps"] = gb / (g[0] * 1e-6) / 1e9
        roof["gather_frac_of_hbm_peak"] = roof["gather_GBps"] / HBM_PEAK_GBPS
    # ---- the step as a whole against its algorithmic HBM / MFMA time
    flops = roof.get("all_gemm_flops_per_step", 0.0)
    n_dense = sum(p.numel() for p in ts.model.parameters() if p is not opt.table)
    touched = B * F * 7 * D * 4                                            # w, m, v read + written and the row gradient, per looked-up row (upper bound: all distinct)
    alg_us = flops / (MFMA_BF16_PEAK_TFLOPS * 1e12) * 1e6 + (gb + touched + 28.0 * n_dense) / (HBM_PEAK_GBPS * 1e9) * 1e6
    roof["step"] = {"algorithmic_us": alg_us, "ms_per_step": ms_per_step, "frac": alg_us * 1e-3 / ms_per_step,
                    "terms": {"contraction_flops": flops, "gather_bytes": gb, "touched_row_bytes": touched, "dense_param_bytes": 28.0 * n_dense},
                    "note": "sum of (flops / 2.5 PFLOP/s) and (bytes / 8 TB/s) over the step's algorithmic work, divided by the timed ms_per_step; "
                            "the replay arithmetic of the lazy table (VALU work, hidden in the background) is not in the numerator"}
    ins = in_step_figures()
    if ins:
        if ins.get("contraction_launches_us") and flops:
            ins["gemm_frac_in_step"] = flops / (ins["contraction_launches_us"] * 1e-6) / 1e12 / MFMA_BF16_PEAK_TFLOPS
        if ins.get("gather_us"):
            ins["gather_frac_in_step"] = gb / (ins["gather_us"] * 1e-6) / 1e9 / HBM_PEAK_GBPS
        roof["in_step"] = ins
    return roof


if __name__ == "__main__":
    main()
