#!/usr/bin/env python3
"""Benchmark of the PPCA EM hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W        (N > 1: launched by torch.distributed.run, or self-launching)

One "step" = one full EM iteration over the synthetic dataset (fused E-step + M-step
statistics pass, one RCCL all-reduce of the packed statistics when N > 1, on-device
finalisation into the next model), inputs resident in HBM.  Workload = BASELINE.json's
metric configuration: N = 10M samples x d = 256 x state_size = 10, 30 % iid masking,
sharded by contiguous row blocks over the ranks (total work fixed => strong scaling).
`--config 5` runs BASELINE configuration 5 instead (mixture of K = 8 components, N = 5 M): one step = one
PPCAMix EM iteration = ONE C-ABI call per rank (ppca_mix_em_step_sharded / ppca_mix_em_step).
Rank 0 prints ONE JSON line.
"""
from __future__ import annotations

import argparse
import json
import os
import shutil
import subprocess
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
TRAFFIC_FILES = ("profiles/r06/traffic.json", "profiles/r05/traffic.json", "profiles/r04/traffic.json", "profiles/r03/traffic.json", "profiles/r02/traffic.json")


def algorithmic_bytes_per_sample(d: int) -> float:
    return 8.0 * d + d / 8.0 + 8.0  # X row (f64) + bitmask + weight


def algorithmic_flops_per_sample(d: int, k: int, m: float) -> float:
    kp = k * (k + 1) / 2
    return 4 * m * kp + 4 * m * k + 2 * d * k + 2 * k ** 3


def algorithmic_flops_llk_per_sample(k: int, m: float) -> float:
    """One log-likelihood evaluation (ppca_model.rs:124-139): Gram 2 m k', b 2 m k, Cholesky k^3 / 3, forward substitution k^2."""
    kp = k * (k + 1) / 2
    return 2 * m * kp + 2 * m * k + k ** 3 / 3 + k ** 2


def cargo_probe() -> dict:
    """record whether the reference's own toolchain exists on this box (it cannot build
    here either way: no network for the crates)."""
    path = shutil.which("cargo")
    ver = None
    if path:
        try:
            ver = subprocess.run([path, "--version"], capture_output=True, text=True, timeout=10).stdout.strip()
        except Exception as e:  # noqa: BLE001
            ver = f"error: {e}"
    return {"cargo": path, "cargo_version": ver}


def timed_calls(fn, warmup_left: int, min_calls: int = 3, budget_s: float = 30.0):
    """Warm-up calls (the first one is also the feasibility probe of the caller), then >= min_calls timed calls
    within the budget; returns the list of timed durations."""
    for _ in range(warmup_left):
        fn()
    times = []
    t_end = time.time() + budget_s
    while len(times) < min_calls and (not times or time.time() < t_end):
        t0 = time.perf_counter()
        fn()
        times.append(time.perf_counter() - t0)
    return times


def cpu_baseline(ds, start_model, n_total: int, d: int, k: int, rows: int):
    """Times the oracle (literal restatement of the reference's rayon path, OpenMP, all host cores) on a bounded
    sample of the same workload.  Protocol of BASELINE.md section 4: N_cpu = 1 M rows unless one call takes more
    than 20 s (then 100 000 rows, said so), 2 full warm-up iterations, >= 3 timed, median, scaled linearly in N."""
    from oracle import ppca_oracle as o

    c, mu, s = start_model.transform, start_model.mean, start_model.isotropic_noise
    note = ""
    x = ds._slice(0, rows).numpy()
    t0 = time.perf_counter()
    o.iterate(x, s, c, mu)  # warm-up 1 of 2 = the feasibility probe
    probe = time.perf_counter() - t0
    if probe > 20.0 and rows > 100_000:
        note = f"; one iterate() on {rows} rows took {probe:.1f} s (> 20 s): fell back to 100000 rows"
        rows = 100_000
        x = x[:rows]
        o.iterate(x, s, c, mu)
    times = timed_calls(lambda: o.iterate(x, s, c, mu), warmup_left=1)
    t = float(np.median(times))
    # second CPU number: the honestly optimised one-sweep form of the statistics pass
    tf = timed_calls(lambda: o.fused_stats(x, s, c, mu), warmup_left=2, budget_s=15.0)
    tfm = float(np.median(tf))
    out = {
        "value": 1.0 / (t * n_total / rows),
        "unit": "EM iters/sec",
        "cores": o.num_threads(),
        "kind": "port",
        "sample": f"{rows} of {n_total} rows of the same dataset, 2 warm-up + {len(times)} timed iterate() calls "
                  f"(median {t:.3f} s), scaled linearly in N{note}",
        "samples_per_sec": rows / t,
        "optimised_port": {"value": 1.0 / (tfm * n_total / rows), "unit": "EM iters/sec", "cores": o.num_threads(),
                           "what": "oracle.fused_stats: one OpenMP sweep, table Gram + Cholesky, thread-private statistics "
                                   f"(median {tfm:.3f} s on the same {rows} rows, scaled linearly in N; finalisation excluded)"},
    }
    out.update(cargo_probe())
    return out


def cpu_baseline_mix(x: np.ndarray, start, n_total: int):
    """Config 5: oracle.mix_iterate (literal PPCAMix::iterate_with_prior, mix.rs:281-337) on a bounded sample."""
    from oracle import ppca_oracle as o

    sig = [m.isotropic_noise for m in start.models]
    cs = [m.transform for m in start.models]
    ms = [m.mean for m in start.models]
    lw = start.log_weights
    rows = x.shape[0]
    times = timed_calls(lambda: o.mix_iterate(x, sig, cs, ms, lw), warmup_left=1, min_calls=2, budget_s=30.0)
    t = float(np.median(times))
    out = {
        "value": 1.0 / (t * n_total / rows),
        "unit": "mixture EM iters/sec",
        "cores": o.num_threads(),
        "kind": "port",
        "sample": f"{rows} of {n_total} rows of the same dataset, 1 warm-up + {len(times)} timed mix_iterate() calls "
                  f"(median {t:.3f} s), scaled linearly in N",
        "samples_per_sec": rows / t,
    }
    out.update(cargo_probe())
    return out


def self_launch(args) -> int:
    """Plain `python bench.py --gpus N`: start N fresh ranks (one per GPU) as CHILD processes and relay rank 0's JSON
    line.  Nothing in this process has touched torch or the GPU yet, and it never will (a process that has
    initialised the GPU must not exec another program on this pool).  If any rank fails, the others are stopped and
    that rank's stderr tail is what the caller sees, with a non-zero exit code."""
    import socket
    import tempfile

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    tmp = tempfile.mkdtemp(prefix="ppca_bench_")
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        err = open(os.path.join(tmp, f"rank{r}.err"), "w")
        # (rank 0's stdout goes to a FILE, not a pipe: nobody reads while the ranks run, and a pipe would block a chatty rank at 64 KB)
        out = open(os.path.join(tmp, "rank0.out"), "w") if r == 0 else subprocess.DEVNULL
        procs.append((subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env, stdout=out, stderr=err), err))
        if r == 0:
            out.close()
    def stop_children(signum=None, frame=None):  # never leave ranks behind when the launcher itself is stopped
        for p, _ in procs:
            if p.poll() is None:
                p.terminate()
        for p, _ in procs:
            try:
                p.wait(timeout=20)
            except subprocess.TimeoutExpired:
                p.kill()
        if signum is not None:
            raise SystemExit(128 + signum)

    import signal

    signal.signal(signal.SIGTERM, stop_children)
    signal.signal(signal.SIGINT, stop_children)
    failed = None
    pending = set(range(args.gpus))
    out0 = b""
    while pending and failed is None:
        for r in sorted(pending):
            p, _ = procs[r]
            rc = p.poll()
            if rc is not None:
                pending.discard(r)
                if rc != 0:
                    failed = (r, rc)
                    break
        time.sleep(0.05)
    if failed is not None:
        for r in pending:
            procs[r][0].terminate()
        for r in pending:
            try:
                procs[r][0].wait(timeout=20)
            except subprocess.TimeoutExpired:
                procs[r][0].kill()
    with open(os.path.join(tmp, "rank0.out"), "rb") as fh:
        out0 = fh.read()
    for _, err in procs:
        err.close()
    if failed is not None:
        r, rc = failed
        with open(os.path.join(tmp, f"rank{r}.err")) as fh:
            tail = fh.read()[-4000:]
        print(f"[bench] rank {r} of {args.gpus} exited with code {rc}; its stderr tail:\n{tail}", file=sys.stderr)
        return rc if rc > 0 else 1
    sys.stdout.write(out0.decode())
    sys.stdout.flush()
    for r in range(args.gpus):  # the ranks' stderr (library notes) goes to ours
        with open(os.path.join(tmp, f"rank{r}.err")) as fh:
            txt = fh.read()
        if txt.strip():
            sys.stderr.write(txt)
    shutil.rmtree(tmp, ignore_errors=True)
    return 0


def traffic_from_profiles(rows_local: int):
    for rel in TRAFFIC_FILES:
        try:
            with open(os.path.join(ROOT, rel)) as fh:
                tj = json.load(fh)
            return tj["hbm_bytes_per_sample"] * rows_local, f"{rel} (commit {tj.get('commit')}, N = {tj.get('n_samples')}, kernel {tj.get('kernel')})"
        except (OSError, KeyError, ValueError):
            continue
    return PMC_BYTES_PER_SAMPLE * rows_local, "profiles/r01/README.md (N = 2 M, round-1 build)"


def sustained_mfma_from_profiles():
    """What a loop of nothing but independent MFMAs sustains on this part (tools/mfma_peak, run on the GPU box before the
    bench line; reported beside the guide's nominal peak, never instead of it)."""
    for rel in ("profiles/r06/mfma_peak.json", "profiles/r05/mfma_peak.json", "profiles/r04/mfma_peak.json"):
        try:
            with open(os.path.join(ROOT, rel)) as fh:
                pj = json.load(fh)
            pj["file"] = rel
            return pj
        except (OSError, ValueError):
            continue
    return None


SECONDARY = {
    # BASELINE.json configurations 2, 4 and 5 as SHORT legs of the default run, so that the driver's clock sees them too (round 5's
    # review: five rounds of config-4 / config-5 numbers were only ever on the builder's clock).  No CPU leg for them.
    "cfg2": dict(n=1_000_000, d=256, k=10, mask=0.3, mask_kind=0, mask_run=0, warmup=10, steps=50, mixture=0),
    "cfg4": dict(n=2_000_000, d=1024, k=64, mask=0.5, mask_kind=1, mask_run=512, warmup=1, steps=3, mixture=0),
    "cfg5": dict(n=5_000_000, d=256, k=10, mask=0.3, mask_kind=0, mask_run=0, warmup=9, steps=11, mixture=8),
}


def secondary_leg(name: str, ctx, sync) -> dict:
    """One BASELINE configuration other than the headline, on one GPU, in this process: value, ms per step and the roofline fraction
    bench.py --config N reports for it (same data seeds, same start models, same timed regions: EM configurations = kernel time by
    HIP events over the timed steps against the algorithmic fp64 flops; the mixture = the one-pass bytes over the step time)."""
    import ctypes as C

    import ppca_rs_amd as P
    from ppca_rs_amd import _lib
    from ppca_rs_amd.distributed import ShardedEM, ShardedMixEM

    sp = SECONDARY[name]
    n, d, k = sp["n"], sp["d"], sp["k"]

    def generate(c_true, mean_true, row_offset, n_rows, seed):
        truth = P.PPCAModel(0.1, c_true, mean_true)
        spec = _lib.SynthSpec(row_offset, n_rows, d, k, 0.1, sp["mask"], sp["mask_kind"], sp["mask_run"], seed,
                              truth._c.ctypes.data_as(_lib.c_double_p), truth._mean.ctypes.data_as(_lib.c_double_p))
        h = C.c_void_p()
        _lib.check(_lib.lib().ppca_dataset_generate(ctx.handle, C.byref(spec), C.byref(h)))
        return P.Dataset._wrap(h, ctx)

    nm = sp["mixture"]
    if nm:
        blk = 65536
        comps = [(np.random.default_rng(1051 + 10 * c).standard_normal((d, k)), 3.0 * np.random.default_rng(1052 + 10 * c).standard_normal(d))
                 for c in range(nm)]
        parts = [generate(comps[(b0 // blk) % nm][0], comps[(b0 // blk) % nm][1], b0, min(n, b0 + blk) - b0, 1053) for b0 in range(0, n, blk)]
        data = P.Dataset.concat(parts)
        del parts
        ctx.trim()
        start = P.PPCAMix([P.PPCAModel(1.0, np.random.default_rng(2051 + c).standard_normal(d * k).reshape((k, d)).T.copy(), np.zeros(d))
                           for c in range(nm)], np.zeros(nm))
        em = ShardedMixEM(data, start)
    else:
        data = generate(np.random.default_rng(1011).standard_normal((d, k)), np.random.default_rng(1012).standard_normal(d), 0, n, 1013)
        start = P.PPCAModel(1.0, np.random.default_rng(2011).standard_normal(d * k).reshape((k, d)).T.copy(), np.zeros(d))
        em = ShardedEM(data, start)
    for _ in range(sp["warmup"]):
        em.step()
    sync()
    ctx.enable_timing(True)
    t0 = time.perf_counter()
    for _ in range(sp["steps"]):
        em.step()
    sync()
    elapsed = time.perf_counter() - t0
    kern_ms, launches = ctx.kernel_time(reset=True)
    ctx.last_fallback()  # (drops the second stages' events)
    ctx.enable_timing(False)
    t_step = elapsed / sp["steps"]
    m_obs = d * (1.0 - sp["mask"])
    leg = {"workload": f"N={n} x d={d}, state_size={k}, {int(100 * sp['mask'])}% {'block' if sp['mask_kind'] else 'iid'} masked"
                       + (f", mixture of {nm} components" if nm else ""),
           "value": 1.0 / t_step, "unit": "mixture EM iters/sec" if nm else "EM iters/sec", "ms_per_step": 1e3 * t_step,
           "steps": sp["steps"], "warmup": sp["warmup"]}
    if nm:
        used = (C.c_int64 * nm)()
        _lib.check(_lib.lib().ppca_mix_last_rows_used(ctx.handle, used, nm))
        rows_gathered = [int(v) for v in used]
        bytes_step = n * algorithmic_bytes_per_sample(d)
        flops_step = nm * n * algorithmic_flops_llk_per_sample(k, m_obs) + sum(rows_gathered) * algorithmic_flops_per_sample(d, k, m_obs)
        leg["roofline"] = {"bound": "hbm", "achieved": bytes_step / t_step / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                           "frac": bytes_step / t_step / 1e9 / HBM_PEAK_GBS,
                           "fp64_frac_executed": flops_step / t_step / 1e12 / FP64_PEAK_TFLOPS,
                           "rows_gathered_fraction_of_K_x_N": sum(rows_gathered) / float(nm * n),
                           "note": "frac = ONE-pass bytes (N x 2088 B) / step time / HBM peak, as bench.py --config 5"}
        traffic_file = "profiles/r06/traffic_cfg5.json"
    else:
        kern_avg = kern_ms / max(launches, 1) * 1e-3
        flops_step = n * algorithmic_flops_per_sample(d, k, m_obs)
        leg["roofline"] = {"bound": "mfma", "achieved": flops_step / kern_avg / 1e12, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s",
                           "frac": flops_step / kern_avg / 1e12 / FP64_PEAK_TFLOPS, "kernel_avg_ms": 1e3 * kern_avg, "kernel_launches": launches,
                           "note": "algorithmic fp64 flops / the pass's HIP-event time (the EM kernel; config 4: every kernel of the split "
                                   "pipeline as one region), as bench.py --config N"}
        traffic_file = {"cfg2": "profiles/r06/traffic.json", "cfg4": "profiles/r06/traffic_cfg4.json"}[name]
    leg["traffic_source"] = None
    for rel in (traffic_file, traffic_file.replace("r06", "r05")):
        try:
            with open(os.path.join(ROOT, rel)) as fh:
                tj = json.load(fh)
            leg["traffic_bytes_per_sample"] = tj["hbm_bytes_per_sample"]
            leg["traffic_source"] = f"{rel} (commit {tj.get('commit')}; PMC FETCH_SIZE / WRITE_SIZE passes, not measured in this run)"
            break
        except (OSError, KeyError, ValueError):
            continue
    if not nm:
        em.close()
    del em, data
    ctx.trim()
    return leg


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--n", "--samples", dest="n", type=int, default=10_000_000)
    ap.add_argument("--d", type=int, default=256)
    ap.add_argument("--k", type=int, default=10)
    ap.add_argument("--mask", type=float, default=0.3)
    ap.add_argument("--cpu-rows", type=int, default=1_000_000)
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo for 1-GPU tests)")
    ap.add_argument("--collective", default="auto", choices=["auto", "capi", "torch"],
                    help="who runs the all-reduce of the statistics: the library's own RCCL communicator behind the "
                         "C-ABI (capi), torch.distributed (torch), or capi with torch as the fallback if the "
                         "communicator cannot be created (auto)")
    ap.add_argument("--gram", default="auto", choices=["auto", "fp64"],
                    help="Gram engine of the fused passes: the int8-sliced MFMA behind its dynamic-range guard (auto), or "
                         "the guard's fallback engine, the fp64 MFMA, always (fp64; = PPCA_GRAM_FP64=1)")
    ap.add_argument("--dump-model", default=None, help="rank 0 writes the final model to this .npz (tests)")
    ap.add_argument("--outliers", type=float, default=0.0,
                    help="heavy-tailed data: this many rows per million (at fixed global positions) are multiplied by 1e6 -- the regime of "
                         "the W-side guard: the flagged workgroups' slices are recomputed on the fp64 engine (roofline.fallback)")
    ap.add_argument("--config", type=int, default=0, choices=[0, 1, 2, 3, 4, 5],
                    help="a BASELINE.json configuration by number (1: toy 10k x 32 x 4 unmasked; 2: 1M x 256 x 10; 3: the "
                         "headline 10M x 256 x 10 = the default; 4: 2M x 1024 x 64, 50%% block-masked, generic pipeline; "
                         "5: mixture of 8 components, 5M x 256 x 10); 0 = take --n/--d/--k/--mask as given")
    ap.add_argument("--components", type=int, default=8, help="config 5: number of mixture components")
    ap.add_argument("--no-secondary", action="store_true",
                    help="the default run (one GPU, the headline configuration) also times short legs of BASELINE configurations 2, 4 and "
                         "5 after the headline region and reports them under \"secondary\"; this switch leaves them out")
    args = ap.parse_args()
    default_run = args.config in (0, 3) and (args.n, args.d, args.k, args.mask) == (10_000_000, 256, 10, 0.3) and args.outliers == 0 and args.gram == "auto"
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
    elif args.config == 5:
        if args.n == 10_000_000:
            args.n = 5_000_000
        args.d, args.k, args.mask = 256, 10, 0.3
    mixture = args.config == 5
    if args.steps is None:
        args.steps = 11 if mixture else 20   # config 5: iterations 10..20 (the converged regime) are the timed ones
    if args.warmup is None:
        args.warmup = 9 if mixture else 3    # ... after iterations 1..9, of which the first three are reported too
    if args.gram == "fp64":
        os.environ["PPCA_GRAM_FP64"] = "1"  # read once by the library when it loads (below)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(self_launch(args))
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

    import ctypes as C

    import ppca_rs_amd as P
    from ppca_rs_amd import _lib
    from ppca_rs_amd.distributed import ShardedEM, ShardedMixEM, shard_bounds

    ctx = _lib.Context(dev_index)
    _lib.set_default_context(ctx)

    n, d, k = args.n, args.d, args.k
    a, b = shard_bounds(n, world, rank)

    def generate(c_true, mean_true, row_offset, n_rows, seed):
        truth = P.PPCAModel(0.1, c_true, mean_true)
        spec = _lib.SynthSpec(row_offset, n_rows, d, k, 0.1, args.mask, mask_kind, mask_run, seed,
                              truth._c.ctypes.data_as(_lib.c_double_p), truth._mean.ctypes.data_as(_lib.c_double_p))
        h = C.c_void_p()
        _lib.check(_lib.lib().ppca_dataset_generate(ctx.handle, C.byref(spec), C.byref(h)))
        return P.Dataset._wrap(h, ctx)

    if mixture:
        # K ground-truth components, mean_c ~ N(0, 3^2), C_c ~ N(0, 1), sigma 0.1, uniform
        # weights; rows keyed by GLOBAL index (block b of 65536 rows belongs to component b % K), so every rank's
        # contiguous shard holds all components and the data do not depend on the number of ranks
        nm, blk = args.components, 65536
        comps = [(np.random.default_rng(1051 + 10 * c).standard_normal((d, k)), 3.0 * np.random.default_rng(1052 + 10 * c).standard_normal(d))
                 for c in range(nm)]
        parts = []
        for b0 in range((a // blk) * blk, b, blk):
            lo, hi = max(a, b0), min(b, b0 + blk)
            if hi > lo:
                c = (b0 // blk) % nm
                parts.append(generate(comps[c][0], comps[c][1], lo, hi - lo, 1053))
        shard = P.Dataset.concat(parts) if len(parts) != 1 else parts[0]
        del parts
        ctx.trim()
        # start = K random models, log-weights 0 (PPCAMix::init mix.rs:76-83); seeded, identical on every rank
        start = P.PPCAMix([P.PPCAModel(1.0, np.random.default_rng(2051 + c).standard_normal(d * k).reshape((k, d)).T.copy(), np.zeros(d))
                           for c in range(nm)], np.zeros(nm))
    else:
        # SURVEY.md 8(d) seeds: C_true 1011, mean_true 1012, data 1013; start model 2011
        c_true = np.random.default_rng(1011).standard_normal((d, k))
        mean_true = np.random.default_rng(1012).standard_normal(d)
        shard = generate(c_true, mean_true, a, b - a, 1013)
        if args.outliers > 0:
            # global rows 137 + i * stride (independent of the number of ranks): this shard's are scaled in place
            stride = max(1, int(round(1e6 / args.outliers)))
            rows = np.arange(137, n, stride, dtype=np.int64)
            rows = rows[(rows >= a) & (rows < b)] - a
            _lib.check(_lib.lib().ppca_dataset_scale_rows(shard._h, rows.ctypes.data_as(C.POINTER(C.c_int64)), len(rows), 1e6))
        c0 = np.random.default_rng(2011).standard_normal(d * k).reshape((k, d)).T.copy()
        start = P.PPCAModel(1.0, c0, np.zeros(d))  # as PPCAModel::init (ppca_model.rs:51-70)

    # The collective(s) of the path.  Default: the library's own RCCL communicator (ppca_comm, behind the C-ABI);
    # torch.distributed carries only the rendezvous (unique id), the barrier and the max-over-ranks of the clock.
    comm, collective = None, "none (single rank)"
    if world > 1:
        collective = f"torch.distributed all_reduce ({args.backend})"
        if args.collective in ("auto", "capi") and args.backend == "nccl":
            from ppca_rs_amd.distributed import Communicator

            try:
                comm = Communicator.from_torch(ctx)  # collective, failures included: every rank takes the same branch
                n_ranks = int(_lib.lib().ppca_comm_n_ranks(comm.h))
                assert n_ranks == world, f"rank {rank}: communicator spans {n_ranks} ranks, WORLD_SIZE is {world}"
                collective = ("ppca_mix_em_step_sharded: " if mixture else "ppca_em_step_sharded: ") + Communicator.backend() + f", {n_ranks} ranks"
            except RuntimeError as e:
                if args.collective == "capi":
                    raise
                print(f"[bench rank {rank}] C-ABI communicator unavailable ({e}); using torch.distributed", file=sys.stderr)
    em = ShardedMixEM(shard, start, comm=comm) if mixture else ShardedEM(shard, start, comm=comm)

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    first_iters = []
    llk_trace = []
    for i in range(args.warmup):
        if mixture and i < 3:  # the soft-responsibility regime right after PPCAMix.init: timed on the side
            sync()
            t0 = time.perf_counter()
            llk_trace.append(em.step())
            sync()
            first_iters.append(time.perf_counter() - t0)
        else:
            v = em.step()
            if mixture:
                llk_trace.append(v)
    sync()
    ctx.enable_timing(True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        v = em.step()
        if mixture:
            llk_trace.append(v)
    sync()
    elapsed = time.perf_counter() - t0
    kern_ms, launches = ctx.kernel_time(reset=True)
    fb_mode, fb_wgs, fb_rows, fb_ms = ctx.last_fallback()  # second stage of the guarded EM passes (the last step's verdict; the timed steps' ms)
    ctx.enable_timing(False)
    if world > 1:
        t = torch.tensor([elapsed] + first_iters, dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t[0].item())
        first_iters = [float(x) for x in t[1:].tolist()]
    if mixture:
        llk_last = llk_trace[-1] if llk_trace else float("nan")
    else:
        llk_last = em.llk_of_previous()
    if args.dump_model and rank == 0 and not mixture:
        fm = em.model()
        np.savez(args.dump_model, sigma=fm.isotropic_noise, transform=fm.transform, mean=fm.mean, llk=llk_last)

    if rank == 0:
        ms_per_step = 1e3 * elapsed / args.steps
        iters_per_s = args.steps / elapsed
        rows_local = b - a
        m_obs = d * (1.0 - args.mask)
        # which Gram engine the fused passes ran on (decided on the device per model; the start model is representative
        # of the run's: unit-scale C, sigma 1 -> 0.1)
        eng = C.c_int32(-1)
        probe = start.models[0] if mixture else start
        _lib.check(_lib.lib().ppca_gram_engine(ctx.handle, probe._device(ctx).h, C.byref(eng)))
        gram_engine = {0: "int8-sliced MFMA (exact integer accumulation) behind the dynamic-range guard", 1: "fp64 MFMA"}.get(eng.value, "?")
        fused = _lib.lib().ppca_path_kind(d, k) == 1
        if mixture:
            nm = args.components
            # the mixture is reported against the ONE-pass byte figure (X read once per iteration) and
            # K x the flops of a k = 10 EM iteration (what the reference's K weighted iterate() calls perform; the
            # responsibility-sparse component passes skip the rows whose weight is below 2^-200 of the component's largest)
            bytes_step = rows_local * algorithmic_bytes_per_sample(d)
            t_step = elapsed / args.steps
            # What the step EXECUTED (never K x N x the EM flops: the component passes gather only the rows whose weight is not
            # negligible -- the device-side row counts of the last timed step): K log-likelihood sweeps over every row + the
            # gathered EM passes
            if em._one_call:
                used = (C.c_int64 * nm)()
                _lib.check(_lib.lib().ppca_mix_last_rows_used(ctx.handle, used, nm))
                rows_gathered = [int(v) for v in used]
            else:  # the step composed from the building blocks (--collective torch / gloo): ppca_mix_component_stats' own counts
                rows_gathered = [int(v) for v in em.backend.rows_used]
            flops_step = (nm * rows_local * algorithmic_flops_llk_per_sample(k, m_obs)
                          + sum(rows_gathered) * algorithmic_flops_per_sample(d, k, m_obs))
            flops_reference = nm * rows_local * (algorithmic_flops_llk_per_sample(k, m_obs) + algorithmic_flops_per_sample(d, k, m_obs))
            tflops, gbs = flops_step / t_step / 1e12, bytes_step / t_step / 1e9
            fp64_bound = False  # the mixture is reported against the ONE-pass byte figure (X read once per iteration)
            llk_kernel = "llk2_kernel" if os.environ.get("PPCA_LLK8") == "0" else "llk8_kernel"
            one_launch = os.environ.get("PPCA_MIX_MULTI") != "0" and llk_kernel == "llk8_kernel" and nm <= 16
            em_kernel = ("pass_kernel" if os.environ.get("PPCA_EM8") == "0" else
                         "em8_kernel" if os.environ.get("PPCA_EM9") == "0" else "em9_kernel")
            kernel_name = ((f"one mixture EM iteration = ONE mix_llk8_kernel<{k}> launch for the {nm} llk sweeps (X read once)" if one_launch
                            else f"one mixture EM iteration = {nm} {llk_kernel}<{k}> sweeps") +
                           f" + {nm} gathered {em_kernel}<{k}, true> passes + finalisations (timed as one region" + (": ONE C-ABI call)" if em._one_call else ", composed from the C-ABI building blocks)"))
            kern_avg_ms, launches_rep = 1e3 * t_step, args.steps
            traffic, traffic_src = None, None
            if (d, k, nm) == (256, 10, 8):
                # HBM bytes per sample of ONE steady-state mixture iteration over every kernel it launches (tools/pmc_mix.py under
                # separate rocprofv3 FETCH_SIZE / WRITE_SIZE passes, tools/make_traffic_all.py), committed with its commit
                multi = os.environ.get("PPCA_MIX_MULTI") != "0"
                for rel in (("profiles/r06/traffic_cfg5.json",) if multi else ()) + ("profiles/r05/traffic_cfg5.json",):
                    try:
                        with open(os.path.join(ROOT, rel)) as fh:
                            tj = json.load(fh)
                        traffic = tj["hbm_bytes_per_sample"] * rows_local
                        traffic_src = (f"{rel} (commit {tj.get('commit')}, {tj.get('n_samples')} rows x {tj.get('steps')} steady-state "
                                       "mixture iterations, all kernels; fabric bytes, Infinity-Cache hits included)")
                        break
                    except (OSError, KeyError, ValueError):
                        continue
        else:
            kern_avg_ms = kern_ms / max(launches, 1)
            launches_rep = launches
            bytes_step = rows_local * algorithmic_bytes_per_sample(d)
            flops_step = rows_local * algorithmic_flops_per_sample(d, k, m_obs)
            t_kernel = kern_avg_ms * 1e-3
            tflops, gbs = flops_step / t_kernel / 1e12, bytes_step / t_kernel / 1e9
            fp64_bound = flops_step / (FP64_PEAK_TFLOPS * 1e12) >= bytes_step / (HBM_PEAK_GBS * 1e9)
            traffic, traffic_src = None, None
            if (d, k) == (256, 10) and mask_kind == 0:
                # HBM bytes per sample of the dominant kernel by rocprofv3 PMC passes (2 x FETCH_SIZE + WRITE_SIZE, gfx950
                # correction), measured at the headline size and committed with the commit it was measured on
                traffic, traffic_src = traffic_from_profiles(rows_local)
            if (d, k) == (1024, 64):
                # config 4: HBM bytes per sample and EM step over EVERY kernel of the split pipeline (2 x FETCH_SIZE + WRITE_SIZE
                # by separate rocprofv3 PMC passes of tools/pmc_generic.py), committed with the commit it was measured on
                for rel in ("profiles/r06/traffic_cfg4.json", "profiles/r05/traffic_cfg4.json", "profiles/r04/traffic_cfg4.json"):
                    try:
                        with open(os.path.join(ROOT, rel)) as fh:
                            tj = json.load(fh)
                        traffic = tj["hbm_bytes_per_sample"] * rows_local
                        traffic_src = (f"{rel} (commit {tj.get('commit')}, {tj.get('n_samples')} rows x {tj.get('steps', tj.get('em_steps'))} EM steps, "
                                       "all kernels of one pass; fabric bytes, Infinity-Cache hits included)")
                        break
                    except (OSError, KeyError, ValueError):
                        continue
            if fused:
                if eng.value == 1:
                    kernel_name = f"ppca::pass_kernel<{k}, true, 4, false, false>"  # the guard's fallback engine
                elif os.environ.get("PPCA_EM8") == "0":
                    kernel_name = f"ppca::pass_kernel<{k}, true, 4, true, false>"
                elif os.environ.get("PPCA_EM9") == "0":
                    kernel_name = f"ppca::em8_kernel<{k}, false>"
                else:
                    kernel_name = f"ppca::em9_kernel<{k}, false>"
            elif d <= 256 and 11 <= k <= 16 and os.environ.get("PPCA_EM16") != "0":
                kernel_name = (f"ppca::estep16_kernel<{k}> + ppca::sstat16_kernel<{k}> per chunk of 2^20 rows (the two fused kernels of "
                               "one pass and their partial reduction, timed as one region)")
            else:
                kernel_name = "generic split pipeline (all kernels of one pass, timed as one region)"
        roofline = {
            "bound": "mfma" if fp64_bound else "hbm",
            "achieved": tflops if fp64_bound else gbs,
            "peak": FP64_PEAK_TFLOPS if fp64_bound else HBM_PEAK_GBS,
            "unit": "TFLOP/s" if fp64_bound else "GB/s",
            "frac": (tflops / FP64_PEAK_TFLOPS) if fp64_bound else (gbs / HBM_PEAK_GBS),
            "traffic": traffic,
            "traffic_unit": "HBM bytes per launch (PMC: 2*FETCH_SIZE + WRITE_SIZE)",
            "traffic_source": traffic_src,
            "kernel": kernel_name,
            "kernel_avg_ms": kern_avg_ms,
            "kernel_launches": launches_rep,
            "algorithmic_flops_per_launch": flops_step,
            "algorithmic_bytes_per_launch": bytes_step,
            "fp64_achieved_tflops": tflops,
            "fp64_peak_tflops": FP64_PEAK_TFLOPS,
            "fp64_frac": tflops / FP64_PEAK_TFLOPS,
            "hbm_achieved_gbs": gbs,
            "hbm_peak_gbs": HBM_PEAK_GBS,
            "hbm_frac": gbs / HBM_PEAK_GBS,
            "note": "bound = the larger of (algorithmic bytes / HBM peak) and (algorithmic fp64 flops / dense fp64 MFMA peak) "
                    "per sample (SURVEY.md 8d): 0.261 ns vs 0.687 ns at d=256, k=10, so the fp64 pipe; hbm_* = the other one",
        }
        if not mixture and fused:
            # second stage of the guarded pass (reduce_wguard_kernel's verdict): what a tripped W-side guard costs -- 0 nothing,
            # 1 the whole pass again on the fp64 engine, 2 only the flagged workgroups' slices (spread over the grid)
            roofline["fallback"] = {"mode_last_pass": fb_mode, "workgroups_recomputed": fb_wgs, "rows_recomputed": fb_rows,
                                    "second_stage_ms_per_step": fb_ms / max(args.steps, 1),
                                    "note": "HIP events around the fp64 fallback pass + second reduction of every timed step (two launches "
                                            "that return at once when no guard trips); not part of kernel_avg_ms"}
        sus = sustained_mfma_from_profiles()
        if sus and fp64_bound:
            roofline["sustained_mfma"] = {
                "fp64_tflops": sus["fp64_16x16x4_tflops_best"], "int8_pops": sus["int8_16x16x64_pops_best"],
                "frac_of_sustained_fp64": tflops / sus["fp64_16x16x4_tflops_best"],
                "source": "%s (tools/mfma_peak: back-to-back independent MFMAs, no memory traffic; commit %s)" % (sus.get("file"), sus.get("commit")),
            }
        if mixture:
            roofline["note"] = ("frac = SURVEY.md 8(d)'s ONE-pass bytes (N x 2088 B: X read once per iteration) / step time / HBM peak; "
                                "fp64_* = the flops the step EXECUTED (K llk sweeps over all rows + the EM flops of the rows each "
                                "component pass gathered), not K x N x the EM flops the reference's K weighted iterate() calls perform")
            roofline["rows_gathered_last_step"] = rows_gathered
            roofline["rows_gathered_fraction_of_K_x_N"] = sum(rows_gathered) / float(nm * rows_local)
            roofline["reference_equivalent_flops_per_step"] = flops_reference
        if mixture:
            workload = (f"PPCA mixture EM, {args.components} components, N={n} samples x d={d}, state_size={k}, "
                        f"{int(100 * args.mask)}% iid masked, {world} contiguous row shard(s); per step one all-reduce(MAX) of "
                        f"{args.components} f64 and one all-reduce(SUM) of {args.components * _lib.lib().ppca_stats_len(d, k) + args.components + 1} f64")
            metric = "mixture EM iters/sec at K=8 components, N=5M d=256 k=10, 30% masked (BASELINE config 5)"
            unit = "mixture EM iters/sec"
        else:
            workload = (f"PPCA EM, N={n} samples x d={d}, state_size={k}, {int(100 * args.mask)}% "
                        f"{'block' if mask_kind else 'iid'} masked, " + (f"{args.outliers:g} outlier rows (x 1e6) per million, " if args.outliers > 0 else "") +
                        f"{world} contiguous row shard(s), one all-reduce of {_lib.lib().ppca_stats_len(d, k)} f64 per step")
            metric = "EM iters/sec (and samples/sec/iter) at N=10M d=256 k=10, 30% masked"
            unit = "EM iters/sec"
        out = {
            "metric": metric,
            "value": iters_per_s,
            "unit": unit,
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": workload,
                       "collective": collective,
                       "n_samples": n, "d": d, "state_size": k, "mask_prob": args.mask, "parallelism": f"dp{world}"},
            "samples_per_sec": n * iters_per_s,
            "llk_per_sample_last_input_model": llk_last / n,
            "gram_engine": gram_engine,
            # the two device-side guards of the LAST EM pass of the timed region: did the model send the pass to the fp64 Gram,
            # did the reduced statistics send it to the fp64 mask-side contraction (ppca_em_last_guard; fused shapes only)
            "guards_last_pass": dict(zip(("gram_unsafe", "stats_unsafe"), ctx.last_guard())) if fused else None,
            # the bound is max(bytes / HBM peak, flops / fp64 peak) per sample; at d = 256,
            # k = 10 that is the fp64 pipe (0.687 ns vs 0.261 ns), so the headline fraction is the fp64 one
            # and the HBM figures ride along.
            "roofline": roofline,
        }
        if mixture:
            out["config"]["n_components"] = args.components
            out["regimes"] = {
                "first_iterations": {"what": "iterations 1-3 from PPCAMix.init (soft responsibilities: every component pass sees every row)",
                                     "ms_per_step": [1e3 * x for x in first_iters],
                                     "value": (len(first_iters) / sum(first_iters)) if first_iters else None, "unit": unit},
                "timed": {"what": f"iterations {args.warmup + 1}-{args.warmup + args.steps} (= `value`; from about iteration 10 a sample keeps weight in one component)",
                          "ms_per_step": ms_per_step, "value": iters_per_s, "unit": unit},
            }
            out["llk_per_sample_trace"] = [v / n for v in llk_trace]
        if world == 1 and not args.no_cpu:
            if mixture:
                rows = min(20_000, rows_local)
                out["cpu_baseline"] = cpu_baseline_mix(shard._slice(0, rows).numpy(), start, n)
            else:
                out["cpu_baseline"] = cpu_baseline(shard, start, n, d, k, min(args.cpu_rows, rows_local))
        else:
            out["cpu_baseline"] = None
        if world == 1 and default_run and not args.no_secondary:
            # BASELINE configurations 2, 4, 5 on the same clock as the headline: short legs, after the headline's dataset is released
            em.close()
            del em, shard
            em = None
            ctx.trim()
            out["secondary"] = {}
            for name in ("cfg2", "cfg4", "cfg5"):
                t_leg = time.perf_counter()
                try:
                    out["secondary"][name] = secondary_leg(name, ctx, sync)
                except Exception as e:  # noqa: BLE001  (a failed leg must not cost the headline line)
                    out["secondary"][name] = {"error": f"{type(e).__name__}: {e}"}
                out["secondary"][name]["leg_wall_s"] = time.perf_counter() - t_leg
        print(json.dumps(out), flush=True)

    if not mixture and em is not None:
        em.close()
    if comm is not None:
        comm.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
