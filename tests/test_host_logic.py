"""CPU-only: host-side logic of the product (finalisation from statistics, sharding,
the world-size-2 all-reduce path over gloo, trainer metrics, model accessors).
The oracle supplies per-shard statistics and the expected results."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _case(oracle, n=240, d=12, k=3, seed=3):
    x, _, _ = oracle.synth(n, d, k, 0.3, seed)
    x[5] = np.nan
    rng = np.random.default_rng(seed + 1)
    return x, 0.8, rng.standard_normal((d, k)), 0.1 * rng.standard_normal(d), rng.uniform(0.5, 2.0, n)


def test_finalize_host_matches_oracle_iterate(oracle, hiplib):
    from ppca_rs_amd import PPCAModel
    from ppca_rs_amd.distributed import finalize_host

    x, s, c, mu, w = _case(oracle)
    new = finalize_host(PPCAModel(s, c, mu), oracle.stats(x, s, c, mu, w))
    s1, c1, m1 = oracle.iterate(x, s, c, mu, w)
    np.testing.assert_allclose(new.isotropic_noise, s1, rtol=1e-10)
    np.testing.assert_allclose(new.transform, c1, rtol=1e-8, atol=1e-12)
    np.testing.assert_allclose(new.mean, m1, rtol=1e-9, atol=1e-12)


def test_finalize_host_with_priors(oracle, hiplib):
    from ppca_rs_amd import PPCAModel, Prior
    from ppca_rs_amd.distributed import finalize_host

    x, s, c, mu, w = _case(oracle, d=8, k=2)
    d = 8
    pm, pc = np.linspace(-1, 1, d), 0.5 * np.eye(d) + 0.1
    prior = Prior().with_mean_prior(pm, pc).with_isotropic_noise_prior(3.0, 2.0).with_transformation_precision(0.7)
    op = oracle.Prior(mean=pm, mean_covariance=pc, isotropic_noise_alpha=3.0, isotropic_noise_beta=2.0,
                      transformation_precision=0.7)
    new = finalize_host(PPCAModel(s, c, mu), oracle.stats(x, s, c, mu, w), prior)
    s1, c1, m1 = oracle.iterate(x, s, c, mu, w, op)
    np.testing.assert_allclose(new.isotropic_noise, s1, rtol=1e-10)
    np.testing.assert_allclose(new.transform, c1, rtol=1e-8, atol=1e-12)
    np.testing.assert_allclose(new.mean, m1, rtol=1e-8, atol=1e-12)


def test_empty_dimension_keeps_old_row(oracle, hiplib):
    from ppca_rs_amd import PPCAModel
    from ppca_rs_amd.distributed import finalize_host

    x, s, c, mu, w = _case(oracle)
    x[:, 4] = np.nan  # ppca_model.rs:313-321 / :373-377
    new = finalize_host(PPCAModel(s, c, mu), oracle.stats(x, s, c, mu, w))
    s1, c1, m1 = oracle.iterate(x, s, c, mu, w)
    np.testing.assert_array_equal(new.transform[4], c[4])
    np.testing.assert_array_equal(c1[4], c[4])
    assert new.mean[4] == mu[4] == m1[4]
    np.testing.assert_allclose(new.transform, c1, rtol=1e-8, atol=1e-12)


def test_shard_bounds_follow_chunks_rule():
    from ppca_rs_amd.distributed import shard_bounds

    for n, world in [(10, 3), (1_000_000, 8), (7, 8), (0, 2), (16, 4)]:
        got = [shard_bounds(n, world, r) for r in range(world)]
        stride = -(-n // world) if world else n
        assert got[0][0] == 0 and got[-1][1] == n
        for r, (a, b) in enumerate(got):
            assert a == min(n, r * stride) and b == min(n, a + stride)
        assert sum(b - a for a, b in got) == n


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch.distributed as dist

    from oracle import ppca_oracle as o
    from ppca_rs_amd import PPCAModel
    from ppca_rs_amd.distributed import allreduce_finalize_host, shard_bounds

    dist.init_process_group("gloo", rank=rank, world_size=world)
    x, s, c, mu, w = _case(o, n=203)
    a, b = shard_bounds(len(x), world, rank)
    model = PPCAModel(s, c, mu)
    llks = []
    for _ in range(3):  # three EM iterations, one all-reduce each
        st = o.stats(x[a:b], model.isotropic_noise, model.transform, model.mean, w[a:b])
        model, llk = allreduce_finalize_host(model, st)
        llks.append(llk)
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, model.isotropic_noise, model.transform, model.mean, llks))


def test_world_size_2_gloo_matches_single_process(oracle, hiplib):
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=180) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    x, s, c, mu, w = _case(oracle, n=203)
    want_llk = []
    for _ in range(3):
        want_llk.append(oracle.llk(x, s, c, mu, w))
        s, c, mu = oracle.iterate(x, s, c, mu, w)
    for _, sig, tr, mean, llks in res:
        np.testing.assert_allclose(sig, s, rtol=1e-9)
        np.testing.assert_allclose(tr, c, rtol=1e-7, atol=1e-11)
        np.testing.assert_allclose(mean, mu, rtol=1e-8, atol=1e-11)
        np.testing.assert_allclose(llks, want_llk, rtol=1e-10)
    np.testing.assert_array_equal(res[0][2], res[1][2])  # ranks agree bit for bit: no broadcast needed


def test_model_accessors_and_canonical(oracle):
    from ppca_rs_amd import PPCAModel
    from ppca_rs_amd.api import _metrics

    rng = np.random.default_rng(0)
    c = rng.standard_normal((7, 3))
    m = PPCAModel(0.3, c, rng.standard_normal((7, 1)))
    assert (m.output_size, m.state_size, m.n_parameters) == (7, 3, 1 + 21 + 7)
    np.testing.assert_allclose(m.singular_values, oracle.singular_values(c))
    np.testing.assert_allclose(m.to_canonical().transform, oracle.to_canonical(c), atol=1e-12)
    with pytest.raises(ValueError, match="column- or row- vector"):
        PPCAModel(0.3, c, np.zeros((7, 2)))
    tm = _metrics(-1234.5, 29, 100)  # python/ppca_rs/__init__.py:52-57
    assert tm.llk == -12.345 and tm.aic == 2.0 * (29 + 1234.5) / 100
    assert tm.bic == (-1234.5 - 29 * np.log(100)) / 100
    import pickle

    m2 = pickle.loads(pickle.dumps(m))
    np.testing.assert_array_equal(m2.transform, m.transform)
    assert m2.isotropic_noise == 0.3


def test_prior_validation():
    from ppca_rs_amd import Prior

    with pytest.raises(ValueError):
        Prior().with_isotropic_noise_prior(-1.0, 1.0)
    with pytest.raises(ValueError):
        Prior().with_transformation_precision(-0.1)
    with pytest.raises(ValueError):
        Prior().with_mean_prior(np.zeros(3), np.zeros((3, 3)))
    p = Prior().with_transformation_precision(2.0)
    assert p.transformation_precision == 2.0 and Prior().transformation_precision == 0.0


def test_dataframe_pivot_matches_reference_semantics():
    """the long-format pivot behind DataFrameAdapter.from_pandas (python/ppca_rs/__init__.py:145-206)
    against a direct per-group fill, the way the reference populates its matrix."""
    import pandas as pd

    from ppca_rs_amd import frames

    rng = np.random.default_rng(3)
    rows = []
    for day in ["2024-01-03", "2024-01-01", "2024-01-02"]:
        for shop in ["b", "a"]:
            for prod, size in [("x", 1), ("y", 2), ("x", 2)]:
                if rng.random() < 0.7:
                    rows.append({"day": day, "shop": shop, "product": prod, "size": size, "sales": rng.normal()})
    df = pd.DataFrame(rows)
    mat, dim_idx, smp_idx, dims = frames.pivot_pandas(df, keys=["day", "shop"], dimensions=["product", "size"],
                                                       dimension_idx=None, metric="sales")
    assert dims == ["product", "size"]
    assert [tuple(r) for r in dim_idx[["product", "size"]].values.tolist()] == \
        sorted(set(map(tuple, df[["product", "size"]].values.tolist())))
    want = np.full((len(smp_idx), len(dim_idx)), np.nan)
    lookup = {tuple(r[:2]): int(r[2]) for r in dim_idx[["product", "size", frames.DIM]].values.tolist()}
    for i, (_, chunk) in enumerate(df.groupby(["day", "shop"])):
        for _, r in chunk.iterrows():
            want[i, lookup[(r["product"], r["size"])]] = r["sales"]
    assert np.array_equal(np.isnan(mat), np.isnan(want)) and np.allclose(np.nan_to_num(mat), np.nan_to_num(want))
    assert smp_idx[["day", "shop"]].values.tolist() == sorted(map(list, {tuple(v) for v in df[["day", "shop"]].values.tolist()}))
    # a given dimension index restricts (inner join) and fixes the dimension order
    sub = dim_idx.iloc[[2, 0]].reset_index(drop=True).copy()
    sub[frames.DIM] = [0, 1]
    mat2, _, _, dims2 = frames.pivot_pandas(df, keys=["day", "shop"], dimensions=None, dimension_idx=sub, metric="sales")
    assert dims2 == ["product", "size"] and mat2.shape[1] == 2
    # description <-> json round trip
    d = frames.DataFrameAdapterDescription(["day", "shop"], dims, "sales", dim_idx[dims].values.tolist())
    back = frames.DataFrameAdapterDescription.from_json(d.to_json())
    assert back == d and list(back.dimension_idx_pandas.columns) == [frames.DIM, "product", "size"]


def test_bincode_layouts_against_hand_built_bytes():
    """SURVEY 8f-3 (parity unpinned: no real artefact exists to compare with).  The layouts of wire.py against
    byte strings assembled field by field from the crates' serde rules for the reference's toy model
    (ppca_model.rs:647-656: C = [[1,1],[1,0],[0,1]]), plus round trips."""
    import struct

    from ppca_rs_amd import wire

    c = np.array([[1.0, 1.0], [1.0, 0.0], [0.0, 1.0]])
    mean = np.array([0.5, -1.0, 2.0])
    u64, f = (lambda v: struct.pack("<Q", v)), (lambda *v: struct.pack("<%dd" % len(v), *v))
    want = (f(0.1)                                    # OutputCovariance.isotropic_noise
            + u64(6) + f(1, 1, 0, 1, 0, 1)            # transform: VecStorage.data, column-major
            + u64(3) + u64(2)                         # nrows (Dyn), ncols (Dyn)
            + u64(3) + f(0.5, -1.0, 2.0) + u64(3))    # mean: data, nrows (Dyn); ncols Const<1> = unit
    got = wire.dump_model(0.1, c, mean)
    assert got == want and len(got) == 8 + 8 + 48 + 16 + 8 + 24 + 8
    s2, c2, m2 = wire.load_model(got)
    assert s2 == 0.1 and np.array_equal(c2, c) and np.array_equal(m2, mean)
    # Dataset: two samples of d = 3, one masked entry; BitVec<u32> = [n blocks][blocks][nbits], bit i of block i / 32
    x = np.array([[1.0, np.nan, 3.0], [4.0, 5.0, 6.0]])
    sample0 = u64(3) + struct.pack("<3d", 1.0, float("nan"), 3.0) + u64(3) + u64(1) + struct.pack("<I", 0b101) + u64(3)
    sample1 = u64(3) + f(4, 5, 6) + u64(3) + u64(1) + struct.pack("<I", 0b111) + u64(3)
    want_ds = u64(2) + sample0 + sample1 + u64(2) + f(1.0, 2.0)
    got_ds = wire.dump_dataset(x, np.array([1.0, 2.0]))
    assert got_ds == want_ds
    x2, w2 = wire.load_dataset(got_ds)
    assert np.array_equal(np.isnan(x2), np.isnan(x)) and np.array_equal(np.nan_to_num(x2), np.nan_to_num(x))
    assert np.array_equal(w2, [1.0, 2.0])
    # wide rows (two mask blocks) and the mixture container round-trip
    rng = np.random.default_rng(2)
    xx = rng.standard_normal((7, 40))
    xx[rng.random(xx.shape) < 0.3] = np.nan
    x3, _ = wire.load_dataset(wire.dump_dataset(xx, np.ones(7)))
    assert np.array_equal(np.isnan(x3), np.isnan(xx)) and np.array_equal(np.nan_to_num(x3), np.nan_to_num(xx))
    models, lw = wire.load_mix(wire.dump_mix([(0.1, c, mean), (0.2, 2 * c, -mean)], np.log([0.25, 0.75])))
    assert len(models) == 2 and models[1][0] == 0.2 and np.array_equal(models[1][1], 2 * c) and np.allclose(np.exp(lw), [0.25, 0.75])
    with pytest.raises(ValueError):
        wire.load_model(got[:-3])


def test_property_sharded_finalisation_equals_whole(oracle, hiplib):
    """Property (hypothesis): for random shapes, mask rates, weights and shard counts, summing the per-shard
    statistics (the multi-GPU invariant: every statistic is additive over samples) and finalising on the host
    gives the oracle's iterate() of the whole dataset; the optimised CPU pass gives the same statistics."""
    from hypothesis import given, settings, strategies as st

    from ppca_rs_amd import PPCAModel
    from ppca_rs_amd.distributed import finalize_host, shard_bounds, stats_len

    @settings(max_examples=25, deadline=None)
    @given(n=st.integers(2, 90), d=st.integers(1, 14), k=st.integers(1, 4), mp=st.floats(0.0, 0.7),
           world=st.integers(1, 5), weighted=st.booleans(), seed=st.integers(0, 10_000))
    def run(n, d, k, mp, world, weighted, seed):
        k = min(k, d)
        x, _, _ = oracle.synth(n, d, k, mp, seed)
        rng = np.random.default_rng(seed)
        c, mu, s = rng.standard_normal((d, k)), 0.3 * rng.standard_normal(d), float(rng.uniform(0.2, 2.0))
        w = rng.uniform(0.1, 3.0, n) if weighted else None
        total = np.zeros(stats_len(d, k))
        for r in range(world):
            a, b = shard_bounds(n, world, r)
            if b > a:
                total += oracle.stats(x[a:b], s, c, mu, None if w is None else w[a:b])
        whole = oracle.stats(x, s, c, mu, w)
        assert np.allclose(total, whole, rtol=1e-10, atol=1e-10 * max(1.0, np.abs(whole).max()))
        fused = oracle.fused_stats(x, s, c, mu, w)
        assert np.allclose(fused, whole, rtol=1e-8, atol=1e-9 * max(1.0, np.abs(whole).max()))
        if whole[-5] <= 0 or whole[stats_len(d, k) - 8 - d:stats_len(d, k) - 8].sum() <= 0:
            return  # nothing observed: the reference panics (ppca_model.rs:358); no model to compare
        new = finalize_host(PPCAModel(s, c, mu), total)
        s1, c1, m1 = oracle.iterate(x, s, c, mu, w)
        assert abs(new.isotropic_noise - s1) <= 1e-7 * max(s1, 1e-12)
        assert np.allclose(new.transform, c1, rtol=1e-6, atol=1e-8) and np.allclose(new.mean, m1, rtol=1e-7, atol=1e-9)

    run()


class _OracleMixBackend:
    """CPU stand-in for the per-shard pieces of ShardedMixEM (the device backend runs them on the GPU): the oracle
    computes responsibilities and weighted statistics of this rank's rows; finalisation is the product's host
    finalisation.  Exercises the collective logic (MAX of maxima, shifted weights, ONE SUM, log-weights)."""

    def __init__(self, oracle, x, w, sig, cs, ms):
        import torch

        self.o, self.x, self.w, self.torch = oracle, x, w, torch
        self.sig, self.cs, self.ms = np.array(sig, float), np.array(cs, float), np.array(ms, float)
        self.nm, self.d, self.k = self.cs.shape
        from ppca_rs_amd.distributed import stats_len

        self.L = stats_len(self.d, self.k)

    def responsibilities(self, lw):
        lp = self.o.mix_infer_cluster(self.x, self.sig, self.cs, self.ms, lw)
        with np.errstate(divide="ignore"):
            self.u = (np.log(np.where(self.w > 0, self.w, 0.0))[:, None] + lp).T.copy()  # mix.rs:304-309
        return float((self.o.mix_llks(self.x, self.sig, self.cs, self.ms, lw) * self.w).sum())

    def local_max(self, c):
        return float(np.nanmax(self.u[c])) if self.u.shape[1] else -np.inf

    def accumulate(self, c, shift):
        wc = np.exp(self.u[c] - shift)
        self._stats = getattr(self, "_stats", np.zeros(self.nm * self.L))
        self._stats[c * self.L:(c + 1) * self.L] = self.o.stats(self.x, self.sig[c], self.cs[c], self.ms[c], wc)
        return float(wc.sum()), None

    def pack(self, extras):
        return self.torch.from_numpy(np.concatenate([self._stats, extras]))

    def unpack_extras(self, packed, count):
        return packed[-count:].numpy()

    def finalize(self, c, packed):
        from ppca_rs_amd import PPCAModel
        from ppca_rs_amd.distributed import finalize_host

        new = finalize_host(PPCAModel(self.sig[c], self.cs[c], self.ms[c]), packed[c * self.L:(c + 1) * self.L].numpy())
        self.sig[c], self.cs[c], self.ms[c] = new.isotropic_noise, new.transform, new.mean

    def max_tensor(self, values):
        return self.torch.tensor(values, dtype=self.torch.float64)

    def models(self):
        from ppca_rs_amd import PPCAModel

        return [PPCAModel(self.sig[c], self.cs[c], self.ms[c]) for c in range(self.nm)]


def _mix_case(oracle):
    rng = np.random.default_rng(21)
    d, k, nm = 10, 2, 3
    x = np.concatenate([oracle.synth(70, d, k, 0.3, 300 + c, mean_scale=2.5)[0] for c in range(nm)])
    rng.shuffle(x)
    w = rng.uniform(0.5, 2.0, x.shape[0])
    return x, w, np.array([1.0, 0.8, 1.2]), rng.standard_normal((nm, d, k)), rng.standard_normal((nm, d)), np.log([0.2, 0.5, 0.3])


def _mix_worker(rank, world, port, queue):
    import torch.distributed as dist

    sys.path.insert(0, ROOT)
    from oracle import ppca_oracle as o
    from ppca_rs_amd import PPCAMix, PPCAModel
    from ppca_rs_amd.distributed import ShardedMixEM, shard_bounds

    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    x, w, sig, cs, ms, lw = _mix_case(o)
    a, b = shard_bounds(x.shape[0], world, rank)
    start = PPCAMix([PPCAModel(sig[c], cs[c], ms[c]) for c in range(3)], lw)
    em = ShardedMixEM(None, start, backend=_OracleMixBackend(o, x[a:b], w[a:b], sig, cs, ms))
    llks = [em.step() for _ in range(3)]
    be = em.backend
    queue.put((rank, be.sig.copy(), be.cs.copy(), be.ms.copy(), em.log_weights.copy(), llks))
    dist.destroy_process_group()


def test_sharded_mixture_world_size_2_gloo(oracle, hiplib):
    """BASELINE config 5's N > 1 path on CPU: two gloo ranks, each holding half the rows, against the oracle's
    mixture iteration over the whole dataset (mix.rs:281-337)."""
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_mix_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=240) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    x, w, sig, cs, ms, lw = _mix_case(oracle)
    want_llk = []
    for _ in range(3):
        want_llk.append(float((oracle.mix_llks(x, sig, cs, ms, lw) * w).sum()))
        sig, cs, ms, lw = oracle.mix_iterate(x, sig, cs, ms, lw, w)
    for _, s2, c2, m2, lw2, llks in res:
        np.testing.assert_allclose(llks, want_llk, rtol=1e-10)
        np.testing.assert_allclose(s2, sig, rtol=1e-8)
        np.testing.assert_allclose(c2, cs, rtol=1e-6, atol=1e-9)
        np.testing.assert_allclose(m2, ms, rtol=1e-7, atol=1e-9)
        np.testing.assert_allclose(lw2, lw, rtol=1e-9, atol=1e-12)
    np.testing.assert_array_equal(res[0][2], res[1][2])
