"""world_size-2 gloo test of the one-camera-per-rank step (gaussian_transformer_amd/dist.py).
The rasterizer is the oracle-backed stand-in: what is under test is the sharding, the flat bucket
and the all-reduce, not the kernels."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _scene_and_cams():
    from gaussian_transformer_amd import synth
    from gaussian_transformer_amd.camera import look_at_camera
    sc = synth.make_scene(P=400, width=64, height=48, sh_degree=1, s0=0.08, seed=5, max_sh_degree=1)
    cams = [look_at_camera((0.6 * (i - 0.5), 0.1 * i, 0.0), (0.0, 0.0, 6.0), (0.0, -1.0, 0.0), sc.camera.FoVx, 64, 48) for i in range(2)]
    return sc, cams


def _run_step(rank, world, cam_index, sc, cams, densify=False):
    from gaussian_transformer_amd import rasterizer
    from gaussian_transformer_amd.dist import data_parallel_step
    from gaussian_transformer_amd.loss import training_loss
    from gaussian_transformer_amd.model import GaussianParams
    from gaussian_transformer_amd.render import PipelineParams, TorchCamera, render
    from tests.oracle_backend import OracleBackend
    rasterizer._set_backend_for_tests(OracleBackend())
    pc = GaussianParams.from_synthetic(sc, "cpu")
    ctl = None
    if densify:
        from gaussian_transformer_amd.densify import DensityController
        ctl = DensityController(pc)
    gt = torch.tensor(np.random.default_rng(7 + cam_index).uniform(0, 1, size=(3, 48, 64)).astype(np.float32))
    out = data_parallel_step(pc, TorchCamera(cams[cam_index], "cpu"), PipelineParams(), torch.tensor(sc.bg), gt,
                             render, training_loss)
    if densify:
        with torch.no_grad():
            torch.manual_seed(1000 + rank)                  # ranks do NOT share the global RNG state ...
            n = ctl.densify_and_prune(1e-7, 0.005, 3.0, 20, generator=torch.Generator().manual_seed(42))   # ... only this seed
        out["densified"] = n
    return pc, out


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.set_num_threads(1)
        sc, cams = _scene_and_cams()
        pc, out = _run_step(rank, world, rank, sc, cams)
        q.put((rank, [p.grad.numpy().copy() for p in pc.parameters()], pc.xyz_gradient_accum.numpy().copy(),
               pc.denom.numpy().copy(), out["bucket"].nbytes))
    finally:
        dist.destroy_process_group()


def _worker_direct(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from gaussian_transformer_amd.dist import direct_all_reduce
        out = []
        for n in (1000, 7, 2, 4099):                      # slices that divide evenly, ragged, and ranks that own nothing
            x = torch.arange(n, dtype=torch.float32) * (rank + 1) + rank
            ref = x.clone(); dist.all_reduce(ref)
            got = direct_all_reduce(x.clone())
            out.append((got.numpy().copy(), ref.numpy().copy()))
        q.put((rank, out))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
@pytest.mark.parametrize("world", [2, 3])
def test_direct_all_reduce_matches_all_reduce(world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_direct, args=(r, world, port, q)) for r in range(world)]
    for p in procs: p.start()
    res = [q.get(timeout=240) for _ in range(world)]
    for p in procs:
        p.join(60); assert p.exitcode == 0
    for rank, out in res:
        for got, ref in out:
            np.testing.assert_allclose(got, ref, rtol=1e-6, atol=0)


def _worker_densify(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.set_num_threads(1)
        sc, cams = _scene_and_cams()
        pc, out = _run_step(rank, world, rank, sc, cams, densify=True)
        q.put((rank, pc._xyz.detach().numpy().copy(), pc._scaling.detach().numpy().copy(), out["densified"]))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_ranks_densify_identically():
    """Different cameras per rank, reduced statistics, one shared split seed: both ranks end with the same Gaussians."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_densify, args=(r, 2, port, q)) for r in range(2)]
    for p in procs: p.start()
    res = {}
    for _ in range(2):
        r = q.get(timeout=240); res[r[0]] = r
    for p in procs:
        p.join(60); assert p.exitcode == 0
    assert res[0][3] == res[1][3] and res[0][3]["cloned"] + res[0][3]["split"] > 0, res[0][3]
    assert res[0][1].shape[0] != 400
    np.testing.assert_array_equal(res[0][1], res[1][1])
    np.testing.assert_array_equal(res[0][2], res[1][2])


@pytest.mark.timeout(300)
def test_two_ranks_one_camera_each_allreduce_matches_serial_sum():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs: p.start()
    res = {}
    for _ in range(2):
        r = q.get(timeout=240); res[r[0]] = r
    for p in procs:
        p.join(60); assert p.exitcode == 0
    # serial reference on this process: each camera on its own (world size 1), then averaged by hand
    sc, cams = _scene_and_cams()
    serial = []
    for i in range(2):
        pc, _ = _run_step(0, 1, i, sc, cams)
        serial.append(([p.grad.numpy().copy() for p in pc.parameters()], pc.xyz_gradient_accum.numpy().copy(), pc.denom.numpy().copy()))
    from gaussian_transformer_amd import rasterizer
    rasterizer._set_backend_for_tests(None)
    for k in range(6):
        expect = 0.5 * (serial[0][0][k] + serial[1][0][k])
        np.testing.assert_allclose(res[0][1][k], expect, rtol=1e-5, atol=1e-7)
        np.testing.assert_array_equal(res[0][1][k], res[1][1][k])          # bitwise equal across ranks
    np.testing.assert_allclose(res[0][2], serial[0][1] + serial[1][1], rtol=1e-5, atol=1e-8)   # per-view norms summed
    np.testing.assert_array_equal(res[0][3], serial[0][2] + serial[1][2])
    P, M = 400, 4
    assert res[0][4] == 4 * (P * (3 + 3 + 3 * (M - 1) + 1 + 3 + 4) + 2 * P)      # 23 + 2 floats per Gaussian at M = 4


# ---- GradientExchange: overlap / buckets / active SH / sparse against the dense blocking sum ----------------------
def _fake_grads(rank, P, M, seed=0):
    """Deterministic per-rank 'gradients' with per-rank visibility: rows a rank does not see are exactly zero."""
    rng = np.random.default_rng(100 * seed + rank)
    vis = rng.uniform(size=P) < 0.6
    vis[:3] = False                                     # some rows no rank sees
    g = rng.normal(size=(P, 3 + 3 * M + 1 + 3 + 4)).astype(np.float32) * vis[:, None]
    return vis, g


def _fill(ex, g, P, M):
    v = ex.views(ex.arena())
    off = 0
    for n, w in (("means3D", 3), ("shs", 3 * M), ("opacities", 1), ("scales", 3), ("rotations", 4)):
        v[n].reshape(P, -1).copy_(torch.from_numpy(g[:, off:off + w])); off += w


def _read(ex, arena, P):
    v = ex.views(arena)
    return np.concatenate([v[n].reshape(P, -1).numpy() for n in ex.names], axis=1).copy()


def _worker_exchange(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from gaussian_transformer_amd.dist import GradientExchange
        P, M = 257, 4
        out = {}
        for algo in ("allreduce", "direct"):
            # dense, blocking, one piece: the reference result
            ex = GradientExchange(P, M, "cpu", mode="sync", algo=algo, bucket_bytes=1 << 30)
            vis, g = _fake_grads(rank, P, M)
            _fill(ex, g, P, M); a = ex.arenas[0]; ex.launch()
            out[algo, "dense"] = _read(ex, a, P); out[algo, "dense_bytes"] = ex.bytes_last
            # small buckets + overlap over three consecutive cameras (double buffering)
            ex = GradientExchange(P, M, "cpu", mode="overlap", algo=algo, bucket_bytes=1000)
            res = []
            for cam in range(3):
                vis_c, g_c = _fake_grads(rank, P, M, seed=cam)
                _fill(ex, g_c, P, M); arena = ex.arenas[ex.cur]; ex.launch()
                res.append(arena)
                if cam >= 1:
                    pass                                 # arena of camera cam-1 may still be in flight: only read after wait
            ex.finish()
            out[algo, "overlap_cam2"] = _read(ex, res[2], P)
            out[algo, "overlap_cam0_reused"] = res[0] is res[2]
            # the same three cameras, blocking
            ex = GradientExchange(P, M, "cpu", mode="sync", algo=algo, bucket_bytes=1 << 30)
            vis_c, g_c = _fake_grads(rank, P, M, seed=2)
            _fill(ex, g_c, P, M); a = ex.arenas[0]; ex.launch()
            out[algo, "sync_cam2"] = _read(ex, a, P)
            # sparse: union of the visibility masks, compacted rows
            # (the first sparse exchange does not know the union's size yet and is as large as the dense one; the second is compact;
            #  a union that outgrew the capacity is exchanged densely when it is waited for)
            ex = GradientExchange(P, M, "cpu", mode="sync", algo=algo, bucket_bytes=4096, sparse_slack=(1.0, 0))
            mask = torch.from_numpy(vis)
            _fill(ex, g, P, M); a = ex.arenas[0]; ex.launch(visible=mask)
            out[algo, "sparse_first"] = _read(ex, a, P); out[algo, "sparse_first_bytes"] = ex.bytes_last
            assert bool((mask == torch.from_numpy(vis)).all())            # the caller's mask is not overwritten by the union
            _fill(ex, g, P, M); ex.launch(visible=mask)
            out[algo, "sparse"] = _read(ex, a, P); out[algo, "sparse_bytes"] = ex.bytes_last; out[algo, "union"] = ex.union_rows
            ex._union_known = 5                                            # far too small: overflow -> dense fall-back at wait time
            _fill(ex, g, P, M); ex.launch(visible=mask)
            out[algo, "sparse_overflow"] = _read(ex, a, P); out[algo, "overflows"] = ex.sparse_overflows
            # active SH degree 0 of M = 4: the other coefficient columns are zero on every rank
            g0 = g.copy(); g0.reshape(P, -1)[:, 3 + 3:3 + 3 * M] = 0.0
            ex = GradientExchange(P, M, "cpu", mode="sync", algo=algo); _fill(ex, g0, P, M); a = ex.arenas[0]; ex.launch()
            out[algo, "sh_dense"] = _read(ex, a, P); full_bytes = ex.bytes_last
            ex = GradientExchange(P, M, "cpu", mode="sync", algo=algo); ex.sh_active = 1
            _fill(ex, g0, P, M); a = ex.arenas[0]; ex.launch()
            out[algo, "sh_active"] = _read(ex, a, P); out[algo, "sh_bytes"] = (ex.bytes_last, full_bytes)
        q.put((rank, out))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
@pytest.mark.parametrize("world", [2, 3])
def test_gradient_exchange_variants_equal_the_dense_sum(world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_exchange, args=(r, world, port, q)) for r in range(world)]
    for p in procs: p.start()
    res = {}
    for _ in range(world):
        r = q.get(timeout=240); res[r[0]] = r[1]
    for p in procs:
        p.join(60); assert p.exitcode == 0
    P, M = 257, 4
    serial = np.zeros((P, 3 + 3 * M + 8), np.float32)                   # rank-order sum ((g0 + g1) + g2)
    masks = []
    for r in range(world):
        vis, g = _fake_grads(r, P, M); masks.append(vis)
        serial = (serial + g).astype(np.float32) if r else g.copy()
    union = np.logical_or.reduce(masks)
    for algo in ("allreduce", "direct"):
        for r in range(world):
            o = res[r]
            np.testing.assert_allclose(o[algo, "dense"], serial, rtol=1e-5, atol=1e-6)      # a ring adds in another order
            np.testing.assert_array_equal(o[algo, "dense"], res[0][algo, "dense"])            # every rank holds the same bits
            exact = algo == "direct" or world == 2      # a ring's summation order depends on where an element sits in
            same = np.testing.assert_array_equal if exact else (lambda a, b: np.testing.assert_allclose(a, b, rtol=1e-5, atol=1e-6))
            same(o[algo, "overlap_cam2"], o[algo, "sync_cam2"])                              # overlap + buckets change no bit ...
            assert o[algo, "overlap_cam0_reused"]                                             # ... with two arenas reused in turn
            assert o[algo, "union"] == int(union.sum()) and (o[algo, "sparse"][~union] == 0).all()
            assert o[algo, "sparse_bytes"] == P + 4 * int(union.sum()) * (3 + 3 * M + 8) < o[algo, "dense_bytes"]
            same(o[algo, "sh_active"], o[algo, "sh_dense"])
            assert o[algo, "sh_bytes"][0] == o[algo, "sh_bytes"][1] - 4 * P * 3 * (M - 1)
            same(o[algo, "sparse"], o[algo, "dense"])    # bitwise with rank-order sums (direct) or two ranks; a ring of three
                                                         # differs in the last bit only (buffer position decides the order)
            same(o[algo, "sparse_first"], o[algo, "dense"])
            assert o[algo, "sparse_first_bytes"] == P + 4 * P * (3 + 3 * M + 8)
            same(o[algo, "sparse_overflow"], o[algo, "dense"])
            assert o[algo, "overflows"] == 1
        np.testing.assert_array_equal(res[0]["direct", "dense"], serial)                      # direct = the serial rank-order sum, bitwise
