#!/usr/bin/env python3
"""Benchmark of the PPCA EM hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W        (N > 1: launched by torch.distributed.run)

One "step" = one full EM iteration over the synthetic dataset (fused E-step + M-step
statistics pass, one RCCL all-reduce of the packed statistics when N > 1, on-device
finalisation into the next model), inputs resident in HBM.  Workload = BASELINE.json's
metric configuration: N = 10M samples x d = 256 x state_size = 10, 30 % iid masking,
sharded by contiguous row blocks over the ranks (total work fixed => strong scaling).
Rank 0 prints ONE JSON line.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
# HBM bytes per sample of the dominant kernel from rocprofv3 PMC passes (2*FETCH_SIZE + WRITE_SIZE,
# gfx950 correction calibrated on a known byte count): profiles/r01/README.md
PMC_BYTES_PER_SAMPLE = 2077.5
FP64_PEAK_TFLOPS = 78.6  # fp64 vector = fp64 matrix spec (dense MFMA peak for f64)
KERNEL_NAME = "ppca::pass_kernel<10, true, 4, true, false>"


def algorithmic_bytes_per_sample(d: int) -> float:
    return 8.0 * d + d / 8.0 + 8.0  # X row (f64) + bitmask + weight


def algorithmic_flops_per_sample(d: int, k: int, m: float) -> float:
    kp = k * (k + 1) / 2
    return 4 * m * kp + 4 * m * k + 2 * d * k + 2 * k ** 3


def cpu_baseline(ds, start_model, n_total: int, d: int, k: int, rows: int):
    """Times the oracle (literal restatement of the reference's rayon path, OpenMP) on a
    bounded sample of the same workload, on this box's host cores."""
    from oracle import ppca_oracle as o

    x = ds._slice(0, rows).numpy()
    c, mu, s = start_model.transform, start_model.mean, start_model.isotropic_noise
    o.iterate(x[:256], s, c, mu)  # warm up (thread pool, page faults)
    times = []
    t_end = time.time() + 25.0
    while len(times) < 3 and (not times or time.time() < t_end):
        t0 = time.perf_counter()
        o.iterate(x, s, c, mu)
        times.append(time.perf_counter() - t0)
    t = float(np.median(times))
    # second CPU number: the honestly optimised one-sweep form of the statistics pass
    o.fused_stats(x[:256], s, c, mu)
    tf = []
    while len(tf) < 3:
        t0 = time.perf_counter()
        o.fused_stats(x, s, c, mu)
        tf.append(time.perf_counter() - t0)
    tfm = float(np.median(tf))
    return {
        "value": 1.0 / (t * n_total / rows),
        "unit": "EM iters/sec",
        "cores": o.num_threads(),
        "kind": "port",
        "sample": f"{rows} of {n_total} rows of the same dataset, {len(times)} timed iterate() calls "
                  f"(median {t:.3f} s), scaled linearly in N",
        "samples_per_sec": rows / t,
        "optimised_port": {"value": 1.0 / (tfm * n_total / rows), "unit": "EM iters/sec", "cores": o.num_threads(),
                           "what": "oracle.fused_stats: one OpenMP sweep, table Gram + Cholesky, thread-private statistics "
                                   f"(median {tfm:.3f} s on the same {rows} rows, scaled linearly in N; finalisation excluded)"},
    }


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--n", "--samples", dest="n", type=int, default=10_000_000)
    ap.add_argument("--d", type=int, default=256)
    ap.add_argument("--k", type=int, default=10)
    ap.add_argument("--mask", type=float, default=0.3)
    ap.add_argument("--cpu-rows", type=int, default=100_000)
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo for 1-GPU tests)")
    ap.add_argument("--collective", default="auto", choices=["auto", "capi", "torch"],
                    help="who runs the all-reduce of the statistics: the library's own RCCL communicator behind the "
                         "C-ABI (capi), torch.distributed (torch), or capi with torch as the fallback if the "
                         "communicator cannot be created (auto)")
    ap.add_argument("--dump-model", default=None, help="rank 0 writes the final model to this .npz (tests)")
    ap.add_argument("--config", type=int, default=0, choices=[0, 1, 2, 3, 4],
                    help="a BASELINE.json configuration by number (1: toy 10k x 32 x 4 unmasked; 2: 1M x 256 x 10; 3: the "
                         "headline 10M x 256 x 10 = the default; 4: 2M x 1024 x 64, 50%% block-masked, generic pipeline); "
                         "0 = take --n/--d/--k/--mask as given")
    args = ap.parse_args()
    mask_kind, mask_run = 0, 0
    if args.config == 1:
        args.n, args.d, args.k, args.mask = 10_000, 32, 4, 0.0
    elif args.config == 2:
        args.n, args.d, args.k, args.mask = 1_000_000, 256, 10, 0.3
    elif args.config == 3:
        args.n, args.d, args.k, args.mask = 10_000_000, 256, 10, 0.3
    elif args.config == 4:
        args.n, args.d, args.k, args.mask = 2_000_000, 1024, 64, 0.5
        mask_kind, mask_run = 1, 512  # one cyclic run of d/2 masked dims per sample
        args.cpu_rows = min(args.cpu_rows, 4000)  # (the literal port needs ~3 ms per sample at this shape)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # Plain `python bench.py --gpus N`: start N fresh ranks (one per GPU) as CHILD processes and relay rank 0's
        # JSON line.  Nothing in this process has touched torch or the GPU yet, and it never will.
        import socket
        import subprocess

        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        raise SystemExit(subprocess.run(cmd, env=env).returncode)
    if world != args.gpus:
        args.gpus = world

    import torch
    import torch.distributed as dist

    n_dev = torch.cuda.device_count()
    dev_index = local_rank % max(n_dev, 1)  # several ranks may share a GPU only with --backend gloo (tests)
    torch.cuda.set_device(dev_index)
    if world > 1:
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world)

    import ppca_rs_amd as P
    from ppca_rs_amd import _lib
    from ppca_rs_amd.distributed import ShardedEM, shard_bounds

    ctx = _lib.Context(dev_index)
    _lib.set_default_context(ctx)

    n, d, k = args.n, args.d, args.k
    # SURVEY.md 8(d) seeds: C_true 1011, mean_true 1012, data 1013; start model 2011
    c_true = np.random.default_rng(1011).standard_normal((d, k))
    mean_true = np.random.default_rng(1012).standard_normal(d)
    truth = P.PPCAModel(0.1, c_true, mean_true)
    a, b = shard_bounds(n, world, rank)
    spec = _lib.SynthSpec(a, b - a, d, k, 0.1, args.mask, mask_kind, mask_run, 1013,
                          truth._c.ctypes.data_as(_lib.c_double_p), truth._mean.ctypes.data_as(_lib.c_double_p))
    import ctypes as C

    h = C.c_void_p()
    _lib.check(_lib.lib().ppca_dataset_generate(ctx.handle, C.byref(spec), C.byref(h)))
    shard = P.Dataset._wrap(h, ctx)
    c0 = np.random.default_rng(2011).standard_normal(d * k).reshape((k, d)).T.copy()
    start = P.PPCAModel(1.0, c0, np.zeros(d))  # as PPCAModel::init (ppca_model.rs:51-70)

    # The one collective of the path.  Default: the library's own RCCL communicator (ppca_comm, behind the C-ABI);
    # torch.distributed carries only the rendezvous (unique id), the barrier and the max-over-ranks of the clock.
    comm, collective = None, "none (single rank)"
    if world > 1:
        collective = f"torch.distributed all_reduce ({args.backend})"
        if args.collective in ("auto", "capi") and args.backend == "nccl":
            from ppca_rs_amd.distributed import Communicator

            try:
                comm = Communicator.from_torch(ctx)
                collective = "ppca_em_step_sharded: " + Communicator.backend()
            except Exception as e:  # noqa: BLE001
                if args.collective == "capi":
                    raise
                print(f"[bench rank {rank}] C-ABI communicator unavailable ({e}); using torch.distributed", file=sys.stderr)
            # all ranks must agree on the path
            flag = torch.tensor([1 if comm is not None else 0], device="cuda")
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            if int(flag.item()) == 0 and comm is not None:
                comm.close()
                comm, collective = None, f"torch.distributed all_reduce ({args.backend})"
    em = ShardedEM(shard, start, comm=comm)

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        em.step()
    sync()
    ctx.enable_timing(True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        em.step()
    sync()
    elapsed = time.perf_counter() - t0
    kern_ms, launches = ctx.kernel_time(reset=True)
    ctx.enable_timing(False)
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    llk_last = em.llk_of_previous()
    if args.dump_model and rank == 0:
        fm = em.model()
        np.savez(args.dump_model, sigma=fm.isotropic_noise, transform=fm.transform, mean=fm.mean, llk=llk_last)

    if rank == 0:
        ms_per_step = 1e3 * elapsed / args.steps
        iters_per_s = args.steps / elapsed
        rows_local = b - a
        kern_avg_ms = kern_ms / max(launches, 1)
        bytes_launch = rows_local * algorithmic_bytes_per_sample(d)
        flops_launch = rows_local * algorithmic_flops_per_sample(d, k, d * (1.0 - args.mask))
        achieved = bytes_launch / (kern_avg_ms * 1e-3) / 1e9
        t_kernel = kern_avg_ms * 1e-3
        tflops = flops_launch / t_kernel / 1e12
        gbs = achieved
        fp64_bound = flops_launch / (FP64_PEAK_TFLOPS * 1e12) >= bytes_launch / (HBM_PEAK_GBS * 1e9)
        traffic, traffic_src = None, None
        if (d, k) == (256, 10) and mask_kind == 0:
            # HBM bytes per sample of the dominant kernel by rocprofv3 PMC passes (2 x FETCH_SIZE + WRITE_SIZE, gfx950
            # correction), measured at the headline size and committed with the commit it was measured on
            try:
                with open(os.path.join(ROOT, "profiles", "r02", "traffic.json")) as fh:
                    tj = json.load(fh)
                traffic = tj["hbm_bytes_per_sample"] * rows_local
                traffic_src = f"profiles/r02/traffic.json (commit {tj.get('commit')}, N = {tj.get('n_samples')})"
            except (OSError, KeyError, ValueError):
                traffic, traffic_src = PMC_BYTES_PER_SAMPLE * rows_local, "profiles/r01/README.md (N = 2 M, round-1 build)"
        roofline = {
            "bound": "mfma" if fp64_bound else "hbm",
            "achieved": tflops if fp64_bound else gbs,
            "peak": FP64_PEAK_TFLOPS if fp64_bound else HBM_PEAK_GBS,
            "unit": "TFLOP/s" if fp64_bound else "GB/s",
            "frac": (tflops / FP64_PEAK_TFLOPS) if fp64_bound else (gbs / HBM_PEAK_GBS),
            "traffic": traffic,
            "traffic_unit": "HBM bytes per launch (PMC: 2*FETCH_SIZE + WRITE_SIZE)",
            "traffic_source": traffic_src,
            "kernel": KERNEL_NAME if (d, k) == (256, 10) else ("ppca::pass_kernel<k, true, 4, true>" if _lib.lib().ppca_path_kind(d, k) == 1
                                                               else "generic split pipeline (all kernels of one pass, timed as one region)"),
            "kernel_avg_ms": kern_avg_ms,
            "kernel_launches": launches,
            "algorithmic_flops_per_launch": flops_launch,
            "algorithmic_bytes_per_launch": bytes_launch,
            "fp64_achieved_tflops": tflops,
            "fp64_peak_tflops": FP64_PEAK_TFLOPS,
            "fp64_frac": tflops / FP64_PEAK_TFLOPS,
            "hbm_achieved_gbs": gbs,
            "hbm_peak_gbs": HBM_PEAK_GBS,
            "hbm_frac": gbs / HBM_PEAK_GBS,
            "note": "bound = the larger of (algorithmic bytes / HBM peak) and (algorithmic fp64 flops / dense fp64 MFMA peak) "
                    "per sample (SURVEY.md 8d): 0.261 ns vs 0.687 ns at d=256, k=10, so the fp64 pipe; hbm_* = the other one",
        }
        out = {
            "metric": "EM iters/sec (and samples/sec/iter) at N=10M d=256 k=10, 30% masked",
            "value": iters_per_s,
            "unit": "EM iters/sec",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": f"PPCA EM, N={n} samples x d={d}, state_size={k}, {int(100 * args.mask)}% "
                                   f"{'block' if mask_kind else 'iid'} masked, "
                                   f"{world} contiguous row shard(s), one all-reduce of {_lib.lib().ppca_stats_len(d, k)} f64 per step",
                       "collective": collective,
                       "n_samples": n, "d": d, "state_size": k, "mask_prob": args.mask, "parallelism": f"dp{world}"},
            "samples_per_sec": n * iters_per_s,
            "llk_per_sample_last_input_model": llk_last / n,
            # the bound is max(bytes / HBM peak, flops / fp64 peak) per sample; at d = 256,
            # k = 10 that is the fp64 pipe (0.687 ns vs 0.261 ns), so the headline fraction is the fp64 one
            # and the HBM figures ride along.
            "roofline": roofline,
        }
        if world == 1 and not args.no_cpu:
            out["cpu_baseline"] = cpu_baseline(shard, start, n, d, k, min(args.cpu_rows, rows_local))
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)

    em.close()
    if comm is not None:
        comm.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
