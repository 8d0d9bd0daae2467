"""Two (or more) ranks run ShardedMixEM on row shards of one dataset and compare with the single-process
PPCAMix.iterate on the whole dataset (launched by tests/test_gpu_parity.py through torch.distributed.run,
gloo backend so that the ranks may share one GPU)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import torch.distributed as dist

rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
dev_index = int(os.environ.get("LOCAL_RANK", "0")) % max(torch.cuda.device_count(), 1)  # ranks may share a GPU (gloo)
torch.cuda.set_device(dev_index)
dist.init_process_group("gloo", rank=rank, world_size=world)
import ppca_rs_amd as P
from ppca_rs_amd import _lib
_lib.set_default_context(_lib.Context(dev_index))
from ppca_rs_amd.distributed import ShardedMixEM, shard_bounds

STEPS = 3
if len(sys.argv) > 1 and sys.argv[1] == "cfg5":
    # BASELINE config 5 at its own shape (8 components, d = 256, state_size = 10, 30 % masked, weighted): the seeded
    # inputs of the committed fixture; the sharded result is also held against the fixture's (oracle) iterations
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden"))
    from inputs import cfg5_inputs

    x, w, sig, cs, ms, lw = cfg5_inputs()
    start = P.PPCAMix([P.PPCAModel(sig[c], cs[c], ms[c]) for c in range(len(sig))], lw)
    STEPS = 2
else:
    rng = np.random.default_rng(33)
    d, k, nm, n = 40, 3, 3, 3001
    truth = [P.PPCAModel(0.2, rng.standard_normal((d, k)), 2.5 * rng.standard_normal(d)) for _ in range(nm)]
    x = np.concatenate([t.sample(n // nm + (1 if c == 0 else 0), 0.3, seed=50 + c).numpy() for c, t in enumerate(truth)])
    perm = np.random.default_rng(1).permutation(x.shape[0])
    x = x[perm]
    w = np.random.default_rng(2).uniform(0.5, 2.0, x.shape[0])
    start = P.PPCAMix([P.PPCAModel(1.0, rng.standard_normal((d, k)), rng.standard_normal(d)) for _ in range(nm)], np.log([0.3, 0.3, 0.4]))
a, b = shard_bounds(x.shape[0], world, rank)
em = ShardedMixEM(P.Dataset(x[a:b], w[a:b]), start)
llks = [em.step() for _ in range(STEPS)]
got = em.mixture()
if rank == 0:
    ref, want = start, []
    full = P.Dataset(x, w)
    for _ in range(STEPS):
        ref, llk = ref.iterate_with_llk(full)
        want.append(llk)
    rel = lambda u, v: float(np.abs(np.asarray(u) - np.asarray(v)).max() / max(np.abs(np.asarray(v)).max(), 1e-300))
    assert max(abs(p - q) / abs(q) for p, q in zip(llks, want)) < 1e-10, (llks, want)
    for g, r in zip(got.models, ref.models):
        assert abs(g.isotropic_noise - r.isotropic_noise) < 1e-9 * r.isotropic_noise
        assert rel(g.transform, r.transform) < 1e-8 and rel(g.mean, r.mean) < 1e-8
    assert rel(got.log_weights, ref.log_weights) < 1e-9
    if len(sys.argv) > 1 and sys.argv[1] == "cfg5":
        g5 = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "cfg5_d256_k10_m8.npz"))
        assert max(abs(p - q) / abs(q) for p, q in zip(llks, g5["it_llk"])) < 1e-9
        for c, g in enumerate(got.models):
            assert rel(g.transform, g5["it_c"][STEPS - 1][c]) < 1e-6 and abs(g.isotropic_noise - g5["it_sigma"][STEPS - 1][c]) < 1e-7
    print("sharded mixture OK", world, "ranks; llk", llks[-1])
dist.barrier()
dist.destroy_process_group()
