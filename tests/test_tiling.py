"""The z-slab decomposition (patchperpix_amd.tiling) reproduces the whole-volume result.

CPU: the slab / halo / ownership / collective logic is run with the oracle standing in for the
HIP kernels (tests/oracle_ops.py) -- in one process with several slabs, and in TWO processes
over torch.distributed (gloo) -- and compared with the oracle's whole-volume result.
GPU (-m gpu): the same comparison with the real kernels, slabs processed sequentially."""
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import REPO
from patchperpix_amd import synth, tiling
from patchperpix_amd.flags import FLYLIGHT_NOTHIN_CC as FLYLIGHT


def _free_port():
    """a TCP port nobody listens on right now (fixed rendezvous ports collide when two launches
    overlap: parametrised cases under pytest-xdist, two suites on one box)"""
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return str(sk.getsockname()[1])


def make_case(seed=61, shape=(44, 12, 13), ps=(3, 3, 3)):
    c = synth.make_case(shape, ps, seed=seed, cell=[5, 5, 5], overlap_frac=0.02)
    if os.environ.get("PPP_TEST_EMPTY_TOP") == "1":
        # nothing above slice 24: a rank whose slab holds no patch at all
        z = 24
        c["pred"][:, z:] = 0.05
        c["foreground"][z:] = False
        c["numinst"][z:] = 0
    return c, list(ps), dict(FLYLIGHT)


def whole_volume(c, ps, kw):
    from oracle import ppp_oracle as orc
    return orc.to_instance_seg(c["pred"], c["foreground"], c["foreground"].copy(), c["numinst"],
                               ps, **kw)


def test_plan_and_halo():
    slabs = tiling.plan_slabs(44, 4)
    assert slabs[0][0] == 0 and slabs[-1][1] == 44
    assert all(a[1] == b[0] for a, b in zip(slabs, slabs[1:]))
    assert tiling.slabs_of_rank(slabs, 0, 2) + tiling.slabs_of_rank(slabs, 1, 2) == slabs
    assert tiling.halo((7, 7, 7)) == 17
    assert tiling.local_range(slabs[1:2], 44, (3, 3, 3)) == (max(0, slabs[1][0] - 7), min(44, slabs[1][1] + 7))


@pytest.mark.parametrize("n_slabs,thin", [(1, False), (3, False), (4, True)])
def test_sequential_slabs_equal_whole_volume_cpu(n_slabs, thin):
    import torch
    from oracle_ops import OracleOps
    c, ps, kw = make_case()
    kw["skipThinCover"] = not thin
    ref = whole_volume(c, ps, kw)
    ops = OracleOps(**kw)
    slabs = tiling.plan_slabs(c["pred"].shape[1], n_slabs)
    inst, fg = tiling.assemble(torch.from_numpy(c["pred"]), 0, c["foreground"].shape,
                               c["foreground"].copy(), c["foreground"].copy(), c["numinst"], ps,
                               slabs, ops=ops, _cover_chunk=500, **kw)
    assert np.array_equal(inst, ref["instances"])
    pairs, aff = tiling.assemble(torch.from_numpy(c["pred"]), 0, c["foreground"].shape,
                                 c["foreground"].copy(), c["foreground"].copy(), c["numinst"], ps,
                                 slabs, ops=ops, return_intermediates=True, **kw)
    assert np.array_equal(pairs, ref["pairs"])
    assert np.array_equal(aff.view(np.uint32), ref["aff"].view(np.uint32))


@pytest.mark.parametrize("n_slabs,yx", [(2, (2, 2)), (1, (3, 2)), (3, (1, 2))])
def test_yx_tiles_equal_whole_volume_cpu(n_slabs, yx):
    """z-slabs cut further into y/x tiles (the 512^3 / 9^3 single-GPU case): same result."""
    import torch
    from oracle_ops import OracleOps
    c, ps, kw = make_case(seed=63, shape=(20, 17, 19))
    ref = whole_volume(c, ps, kw)
    ops = OracleOps(**kw)
    slabs = tiling.plan_slabs(c["pred"].shape[1], n_slabs)
    pairs, aff = tiling.assemble(torch.from_numpy(c["pred"]), 0, c["foreground"].shape,
                                 c["foreground"].copy(), c["foreground"].copy(), c["numinst"], ps,
                                 slabs, ops=ops, return_intermediates=True, _yx_tiles=yx, **kw)
    assert np.array_equal(pairs, ref["pairs"])
    assert np.array_equal(aff.view(np.uint32), ref["aff"].view(np.uint32))
    inst, fg = tiling.assemble(torch.from_numpy(c["pred"]), 0, c["foreground"].shape,
                               c["foreground"].copy(), c["foreground"].copy(), c["numinst"], ps,
                               slabs, ops=ops, _yx_tiles=yx, **kw)
    assert np.array_equal(inst, ref["instances"]) and inst.any()


@pytest.mark.parametrize("n_slabs,yx,thin", [(2, (2, 2), False), (3, (1, 2), True), (4, (1, 1), False)])
def test_consensus_cache_equals_whole_volume_cpu(n_slabs, yx, thin):
    """`_cons_cache`: every base voxel's consensus computed once into a cache over the block (each
    voxel filled exactly once -- the stand-in ops check that), the tiles' shares cut from it in
    both passes: same pair affinities, same instance map."""
    import torch
    from oracle_ops import OracleOps
    from patchperpix_amd import backend
    c, ps, kw = make_case(seed=64, shape=(20, 17, 19))
    kw["skipThinCover"] = not thin
    ref = whole_volume(c, ps, kw)
    ops = OracleOps(**kw)
    slabs = tiling.plan_slabs(c["pred"].shape[1], n_slabs)
    backend.NOTES.pop("cons_cache_gb", None)
    pairs, aff = tiling.assemble(torch.from_numpy(c["pred"]), 0, c["foreground"].shape,
                                 c["foreground"].copy(), c["foreground"].copy(), c["numinst"], ps,
                                 slabs, ops=ops, return_intermediates=True, _yx_tiles=yx, _cons_cache=True, **kw)
    assert "cons_cache_gb" in backend.NOTES          # the cache path ran
    assert np.array_equal(pairs, ref["pairs"])
    assert np.array_equal(aff.view(np.uint32), ref["aff"].view(np.uint32))
    inst, fg = tiling.assemble(torch.from_numpy(c["pred"]), 0, c["foreground"].shape,
                               c["foreground"].copy(), c["foreground"].copy(), c["numinst"], ps,
                               slabs, ops=OracleOps(**kw), _yx_tiles=yx, _cons_cache=True, **kw)
    assert np.array_equal(inst, ref["instances"]) and inst.any()


@pytest.mark.parametrize("name", ["c2d_p5_mark", "c3d_p3_mark_nosparse", "c3d_p3_near_overlap"])
def test_marked_cover_options_in_the_tiled_assembly_cpu(name):
    """`mark_close_neighboorhood` / `select_patches_overlap_neighborhood` (foreground_cover.py:53-85,
    141-168) in the tiled assembly on one rank: the sequential native cover walks the ranked list
    with the patch bits taken from the tiles' frames -- the reference's own result (goldens)."""
    import torch
    from conftest import Golden
    from oracle_ops import OracleOps
    g = Golden(name)
    kw = dict(g.kw)
    shape = g.foreground.shape
    slabs = tiling.plan_slabs(shape[0], 2 if shape[0] > 1 else 1)
    inst, fg = tiling.assemble(torch.from_numpy(g.pred), 0, shape, g.foreground.copy(), g.foreground.copy(),
                               g.numinst.copy(), g.patchshape, slabs, ops=OracleOps(**kw), _yx_tiles=(2, 2), **kw)
    assert np.array_equal(inst, g["instances"]) and inst.any()
    with pytest.raises(NotImplementedError, match="one rank only"):
        class Two(tiling.LocalComm):
            world = 2
        tiling.assemble(torch.from_numpy(g.pred), 0, shape, g.foreground.copy(), g.foreground.copy(),
                        g.numinst.copy(), g.patchshape, slabs, ops=OracleOps(**kw), comm=Two(), **kw)


def test_plan_tiles_takes_the_cache_when_it_fits(monkeypatch):
    monkeypatch.delenv("PPP_CONS_CACHE", raising=False)
    ps = (9, 9, 9)
    # 256^3 / 9^3 next to its prediction on a 288 GB device: the planes (165 GB) fit beside a tile
    n, ny, nx, cache = tiling.plan_tiles((256, 256, 256), ps, 258e9, safety=0.92, copies=2.0, cache_shape=(256, 256, 256))
    assert cache and n * ny * nx > 1
    left = 0.92 * 258e9 - tiling.cons_cache_bytes((256, 256, 256), ps)
    assert 2.0 * ((17 ** 3 - 1) // 2) * 4 * tiling.pairs_box_voxels((256, 256, 256), ps, n, ny, nx) <= left
    # 512^3: 1.3 TB of planes do not fit -- no cache, the usual grid
    n5 = tiling.plan_tiles((512, 512, 512), ps, 90e9, safety=0.92, copies=2.0, cache_shape=(512, 512, 512))
    assert n5 == tiling.tiles_needed((512, 512, 512), ps, 90e9, safety=0.92, copies=2.0) + (False,)
    # a rank's 64 slices of 512^3 (+ 16 halo slices): 206 GB of planes, 12 GB left for tiles
    n8 = tiling.plan_tiles((64, 512, 512), ps, 237e9, safety=0.92, copies=2.0, cache_shape=(80, 512, 512))
    assert n8[3]
    # a volume that fits whole needs none; PPP_CONS_CACHE=0 switches it off
    assert tiling.plan_tiles((64, 64, 64), ps, 200e9, cache_shape=(64, 64, 64)) == (1, 1, 1, False)
    monkeypatch.setenv("PPP_CONS_CACHE", "0")
    assert not tiling.plan_tiles((256, 256, 256), ps, 258e9, safety=0.92, copies=2.0, cache_shape=(256, 256, 256))[3]


def test_plan_ring(monkeypatch):
    monkeypatch.delenv("PPP_RING", raising=False)
    ps = (9, 9, 9)
    # 512^3 / 9^3 next to the resident prediction: columns of 2 x 2, tiles 16 thick, less S1 work
    r = tiling.plan_ring((512, 512, 512), ps, 80e9, safety=0.92, copies=2.0)
    plain = tiling.tiles_needed((512, 512, 512), ps, 80e9, safety=0.92, copies=2.0)
    assert r is not None
    n, ny, nx, ring = r
    thick = max(b - a for a, b in tiling.plan_slabs(512, n))
    assert ring >= thick + 28 and thick % 8 == 0 and tiling.ring_margin(9) == 28
    # wider patches: the margin follows the patch (a fixed 28 failed assemble()'s
    # own check `ring_z >= thick + 2 rad + 2 (pz - 1) + 4` for 11^3 and 13^3)
    for p in (11, 13):
        rr = tiling.plan_ring((384, 384, 384), (p, p, p), 120e9, safety=0.92, copies=2.0)
        if rr is not None:
            t = max(b - a for a, b in tiling.plan_slabs(384, rr[0]))
            assert rr[3] >= t + 2 * (p // 2) + 2 * (p - 1) + 4
    # the ring of the column's pairs box fits the budget
    assert 2.0 * tiling.cons_cache_bytes((1, 1, 1), ps) * ring * tiling.pairs_box_voxels((1, 512, 512), (1, 9, 9), 1, ny, nx) <= 0.92 * 80e9
    assert tiling.consensus_work((512,) * 3, ps, n, ny, nx, ring=True) < 0.9 * tiling.consensus_work((512,) * 3, ps, *plain)
    # a volume that fits whole, or one whose plain grid is as good, takes none; PPP_RING=0 switches it off
    assert tiling.plan_ring((140, 140, 140), (7, 7, 7), 250e9, safety=0.92, copies=2.0) is None
    assert tiling.plan_ring((256, 256, 256), ps, 240e9, safety=0.92, copies=2.0) is None
    monkeypatch.setenv("PPP_RING", "0")
    assert tiling.plan_ring((512, 512, 512), ps, 80e9, safety=0.92, copies=2.0) is None


def test_dry_run_plan():
    """bench.py --dry-run-plan (tiling.dry_run_plan): no GPU.  At N = 1 it reproduces the plan the real
    512^3 / 9^3 run takes (16 x 2 x 2 tiles, a ring of 60 slices: profiles/r06_g_bench_default.json,
    config.plan); at N = 8 every rank owns 64 slices, holds 22 halo slices per inner side, takes the
    consensus cache, and the collectives' byte counts are there."""
    one = tiling.dry_run_plan((512, 512, 512), (9, 9, 9), 1)["ranks_plan"][0]
    assert one["tiles"] == [16, 2, 2] and one["ring_z"] == 60 and not one["cons_cache"]
    assert 2.0 < one["s1_work_per_owned_voxel"] < 2.2 and one["received_gb_per_step_total"] == 0
    eight = tiling.dry_run_plan((512, 512, 512), (9, 9, 9), 8)
    assert eight["halo_slices"] == 22 and len(eight["ranks_plan"]) == 8
    for r in eight["ranks_plan"]:
        assert r["own_z"][1] - r["own_z"][0] == 64 and r["cons_cache"] and r["s1_work_per_owned_voxel"] < 1.3
        inner = (r["own_z"][0] > 0) + (r["own_z"][1] < 512)
        assert r["held_z"][1] - r["held_z"][0] == 64 + 22 * inner
        got = r["received_gb_per_step"]
        assert abs(got["prediction_halo"] - inner * 22 * 729 * 512 * 512 * 2 / 1e9) < 0.01
        assert got["cover_zones_point_to_point"] > got["thinning_zones_point_to_point"] > 0 and got["instances_all_gather"] == 0
    big = tiling.dry_run_plan((1024, 1024, 1024), (9, 9, 9), 8, provider=True)
    assert all(r["received_gb_per_step"]["prediction_halo"] == 0 and r["prediction_resident_gb"] == 0 for r in big["ranks_plan"])


def test_tiles_needed():
    # fits whole: one tile
    assert tiling.tiles_needed((140, 140, 140), (7, 7, 7), 250e9) == (1, 1, 1)
    # 256^3 / 7^3: z-slabs suffice
    n, ny, nx = tiling.tiles_needed((256, 256, 256), (7, 7, 7), 200e9)
    assert n > 1 and (ny, nx) == (1, 1)
    # 512^3 / 9^3 next to a 196 GB prediction: needs y/x tiles, and every tile fits
    n, ny, nx = tiling.tiles_needed((512, 512, 512), (9, 9, 9), 90e9)
    assert ny > 1 and nx > 1
    planes = (17 ** 3 - 1) // 2
    box = [min(512, -(-512 // k) + g) for k, g in ((n, 16), (ny, 24), (nx, 24))]
    assert 3.0 * planes * 4 * box[0] * box[1] * box[2] <= 0.6 * 90e9


WORKER = r"""
import os, sys
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, {repo!r}); sys.path.insert(0, os.path.join({repo!r}, "tests"))
from patchperpix_amd import tiling
from test_tiling import make_case
from oracle_ops import OracleOps
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
c, ps, kw = make_case()
import json
kw.update(json.loads(os.environ.get("PPP_TEST_KW", "{{}}")))
Z = c["pred"].shape[1]
slabs = tiling.plan_slabs(Z, int(os.environ.get("PPP_TEST_SLABS", "4")))
mine = tiling.slabs_of_rank(slabs, rank, world)
lo, hi = tiling.local_range(mine, Z, ps)
fields = [c["foreground"].copy(), c["foreground"].copy(), c["numinst"]]
no_halo = os.environ.get("PPP_TEST_NO_HALO", "0")
if no_halo in ("1", "2"):
    # the U-Net's output as it stands on each rank: the OWN slices only; assemble() fetches the
    # halo from the neighbours (tiling.exchange_halo).  "2": the per-voxel fields are local too
    lo, hi = mine[0][0], mine[-1][1]
    if no_halo == "2":
        fields = [f[lo:hi].copy() for f in fields]
pred_local = torch.from_numpy(np.ascontiguousarray(c["pred"][:, lo:hi]))   # halo'd slab only
if no_halo == "3":
    # a halo-sized buffer whose halo slices are stale (here: zeros): refreshed in place
    pred_local[:, :mine[0][0] - lo] = 0
    pred_local[:, mine[-1][1] - lo:] = 0
    kw["_refresh_halo"] = True
inst, fg = tiling.assemble(pred_local, lo, c["foreground"].shape, fields[0], fields[1], fields[2], ps, mine,
                           comm=tiling.TorchDistComm(), ops=OracleOps(**kw), _cover_chunk=700, **kw)
np.save(os.path.join({out!r}, "inst_rank%d.npy" % rank), inst)
from patchperpix_amd import backend
np.save(os.path.join({out!r}, "notes_rank%d.npy" % rank),
        np.array([backend.NOTES.get("cover_sharded", 0), backend.NOTES.get("cover_rounds", 0),
                  backend.NOTES.get("cover_p2p", 0), 1 if "cons_cache_gb" in backend.NOTES else 0,
                  backend.NOTES.get("thin_sharded", 0), backend.NOTES.get("thin_rounds", 0)]))
dist.destroy_process_group()
"""


# 2 slabs per rank / one slab per rank (kept consensus) / three ranks (a rank with two
# neighbours) / several cover passes (pixel thresholds 10, 0) with thinning / consensus cache
@pytest.mark.parametrize("n_slabs,world,extra", [
    (4, 2, {}), (2, 2, {}), (3, 3, {}),
    (2, 2, {"select_patches_for_sparse_data": False, "skipThinCover": False}),
    (2, 2, {"_empty_top": True}),
    # every rank fills a consensus cache over its own block + halo (two tiles per rank)
    (4, 2, {"_cons_cache": True}),
    # round 6, the north star's halo exchange: every rank holds its OWN slices of the prediction
    # only and receives the patch-radius halo from its neighbours point to point; three ranks
    # (the middle one has two neighbours); local per-voxel fields as well
    # "3": a halo-sized buffer with stale halo slices, refreshed in place (`_refresh_halo`)
    (2, 2, {"_no_halo": "1"}), (3, 3, {"_no_halo": "1"}), (3, 3, {"_no_halo": "2"}), (3, 3, {"_no_halo": "3"})])
def test_ranks_gloo_equal_whole_volume(tmp_path, n_slabs, world, extra, monkeypatch):
    import json
    extra = dict(extra)
    if extra.pop("_empty_top", False):
        monkeypatch.setenv("PPP_TEST_EMPTY_TOP", "1")     # (inherited by the workers)
    no_halo = extra.pop("_no_halo", "0")
    monkeypatch.setenv("PPP_TEST_NO_HALO", no_halo)
    c, ps, kw = make_case()
    kw.update(extra)
    ref = whole_volume(c, ps, kw)
    script = tmp_path / "worker.py"
    script.write_text(WORKER.format(repo=REPO, out=str(tmp_path)))
    _port = _free_port()
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=_port, OMP_NUM_THREADS="1",
               PPP_TEST_SLABS=str(n_slabs), PPP_TEST_KW=json.dumps(extra))
    subprocess.check_call([sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
                           "--nproc-per-node=%d" % world, "--master-addr", "127.0.0.1",
                           "--master-port", _port, str(script)], env=env, timeout=900)
    for r in range(world):
        inst = np.load(tmp_path / ("inst_rank%d.npy" % r))
        assert np.array_equal(inst, ref["instances"]), "rank %d differs" % r
        # the cover ran sharded (z-halo exchange per round), the labels were merged
        notes = np.load(tmp_path / ("notes_rank%d.npy" % r))
        assert notes[0] == world and notes[1] > 0
        assert notes[3] == (1 if extra.get("_cons_cache") else 0)
        # the set-cover thinning ran sharded too (round 6), when the flags ask for it at all
        thin_on = not dict(kw).get("skipThinCover", False)
        assert notes[4] == (world if thin_on else 0) and (notes[5] > 0) == thin_on


HALO_WORKER = r"""
import os, sys
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, {repo!r})
from patchperpix_amd import tiling
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
comm = tiling.TorchDistComm()
Z, H = 23, 7
cuts = [0, 3, 9, 12, 23][:world] + [Z]
a, b = cuts[rank], cuts[rank + 1]
na, nb = max(0, a - H), min(Z, b + H)
vol = torch.arange(5 * Z * 4 * 6, dtype=torch.float32).reshape(5, Z, 4, 6).to(torch.float16)
got = tiling.exchange_halo(vol[:, a:b].contiguous(), (a, b), (na, nb), comm, z_axis=1, chunk_bytes=600)
assert got.shape == (5, nb - na, 4, 6) and torch.equal(got, vol[:, na:nb]), rank
fld = (torch.arange(Z * 4 * 6) % 251).to(torch.int16).reshape(Z, 4, 6)          # (no wire type: bytes)
got = tiling.exchange_halo(fld[a:b].contiguous(), (a, b), (na, nb), comm, z_axis=0)
assert torch.equal(got, fld[na:nb]), rank
# nothing missing anywhere: the tensor itself comes back
same = tiling.exchange_halo(vol[:, na:nb].contiguous(), (na, nb), (na, nb), comm, z_axis=1)
assert same.shape[1] == nb - na
# in place: a halo-sized buffer whose own slices are current and whose halo slices are stale
buf = torch.full((5, nb - na, 4, 6), -1.0, dtype=torch.float16)
buf[:, a - na:b - na] = vol[:, a:b]
own_view = buf.narrow(1, a - na, b - a)
got = tiling.exchange_halo(own_view, (a, b), (na, nb), comm, z_axis=1, out=buf)
assert got.data_ptr() == buf.data_ptr() and torch.equal(buf, vol[:, na:nb]), rank
open(os.path.join({out!r}, "ok%d" % rank), "w").write("ok")
dist.destroy_process_group()
"""


def test_exchange_halo_gloo(tmp_path):
    """tiling.exchange_halo on four ranks whose slabs (3, 6, 3 and 11 slices) are thinner than the
    halo of 7: a rank receives from up to three others, channel-chunked (600-byte budget: one
    channel per step), for a float16 (C, z, Y, X) block and an int16 field (moved as bytes)."""
    script = tmp_path / "halo_worker.py"
    script.write_text(HALO_WORKER.format(repo=REPO, out=str(tmp_path)))
    _port = _free_port()
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=_port, OMP_NUM_THREADS="1")
    subprocess.check_call([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=4",
                           "--master-addr", "127.0.0.1", "--master-port", _port, str(script)], env=env, timeout=600)
    assert all((tmp_path / ("ok%d" % r)).exists() for r in range(4))


@pytest.mark.gpu
@pytest.mark.parametrize("n_slabs", [1, 3])
def test_sequential_slabs_equal_whole_volume_gpu(n_slabs):
    from patchperpix_amd.vote_instances import vote_instances as vi
    c = synth.make_case((48, 20, 22), (5, 5, 5), seed=62, cell=[8, 8, 8], overlap_frac=0.02)
    ps, kw = [5, 5, 5], dict(FLYLIGHT)
    want, _ = vi.to_instance_seg(c["pred"].copy(), c["foreground"].copy(), c["foreground"].copy(),
                                 c["numinst"].copy(), ps, **kw)
    got, _ = tiling.to_instance_seg_tiled(c["pred"].copy(), c["foreground"].copy(),
                                          c["foreground"].copy(), c["numinst"].copy(), ps,
                                          n_slabs, **kw)
    assert np.array_equal(got, want)
    assert got.any()


@pytest.mark.gpu
@pytest.mark.parametrize("n_slabs,yx", [(2, (2, 2)), (1, (2, 3))])
def test_yx_tiles_equal_whole_volume_gpu(n_slabs, yx):
    from patchperpix_amd.vote_instances import vote_instances as vi
    c = synth.make_case((30, 34, 38), (5, 5, 5), seed=64, cell=[8, 8, 8], overlap_frac=0.02)
    ps, kw = [5, 5, 5], dict(FLYLIGHT)
    want_p, want_a = vi.to_instance_seg(c["pred"].copy(), c["foreground"].copy(),
                                        c["foreground"].copy(), c["numinst"].copy(), ps,
                                        **dict(kw, return_intermediates=True, _n_slabs=1))
    got_p, got_a = tiling.to_instance_seg_tiled(c["pred"].copy(), c["foreground"].copy(),
                                                c["foreground"].copy(), c["numinst"].copy(), ps,
                                                n_slabs, return_intermediates=True, _yx_tiles=yx, **kw)
    assert np.array_equal(got_p, want_p)
    assert np.array_equal(got_a.view(np.uint32), want_a.view(np.uint32))
    want, _ = vi.to_instance_seg(c["pred"].copy(), c["foreground"].copy(), c["foreground"].copy(),
                                 c["numinst"].copy(), ps, **dict(kw, _n_slabs=1))
    got, _ = tiling.to_instance_seg_tiled(c["pred"].copy(), c["foreground"].copy(),
                                          c["foreground"].copy(), c["numinst"].copy(), ps,
                                          n_slabs, _yx_tiles=yx, **kw)
    assert np.array_equal(got, want) and got.any()


@pytest.mark.gpu
@pytest.mark.parametrize("name,n_slabs,yx", [("c3d_p7_cells", 2, (2, 2)), ("c3d_p9_cells", 2, (2, 2)),
                                             ("c3d_p9_cells", 1, (3, 2)), ("c3d_p7_thin_mws", 2, (2, 2)),
                                             ("c3d_p5_thin_mws", 3, (2, 2))])
def test_tiles_reproduce_goldens_gpu(name, n_slabs, yx):
    """z-slabs x y/x tiles at 7^3 and 9^3 (two-chunk 81-bit candidate planes, pairs halo on both
    sides in y / x) against the reference's own outputs: pair rows, affinities (bit patterns)
    and the instance map of the golden cases, thinning + mutex watershed included."""
    from conftest import Golden
    g = Golden(name)
    ps, kw = g.patchshape, dict(g.kw, cuda=True, save_no_intermediates=True, sample=1.0,
                                debug=False, isbiHack=False, result_folder="/tmp",
                                affinities="golden.zarr")
    got_p, got_a = tiling.to_instance_seg_tiled(
        g.pred.copy(), g.foreground.copy(), g.foreground.copy(), g.numinst.copy(), ps, n_slabs,
        return_intermediates=True, _yx_tiles=yx, **kw)
    assert np.array_equal(got_p, g["pairs"])
    assert np.array_equal(got_a.view(np.uint32), g["aff"].view(np.uint32))
    got, _ = tiling.to_instance_seg_tiled(
        g.pred.copy(), g.foreground.copy(), g.foreground.copy(), g.numinst.copy(), ps, n_slabs,
        _yx_tiles=yx, **kw)
    assert np.array_equal(got, g["instances"]) and got.any()


@pytest.mark.gpu
@pytest.mark.parametrize("ps,shape,cell,n_slabs,yx,flagset", [
    ((7, 7, 7), (30, 34, 38), 10, 2, (2, 2), "nothin_cc"),
    ((7, 7, 7), (30, 34, 38), 10, 3, (2, 3), "shipped"),
    ((9, 9, 9), (28, 30, 34), 11, 2, (2, 2), "nothin_cc"),
    ((9, 9, 9), (28, 30, 34), 11, 1, (3, 2), "shipped"),
    ((9, 9, 9), (40, 26, 28), 11, 4, None, "cc"),
])
def test_tiles_equal_whole_volume_p7_p9_gpu(ps, shape, cell, n_slabs, yx, flagset):
    """The tile grids BASELINE configs [2]/[3] run on (9^3: 64-bit row ids, two-chunk planes),
    forced on volumes small enough to assemble untiled as well: same pair rows, same affinity
    bits, same instance map -- with the kernels-only, the connected-components and the shipped
    (thinning + mutex watershed) flag sets."""
    from patchperpix_amd import flags as flagsets
    from patchperpix_amd.vote_instances import vote_instances as vi
    c = synth.make_case(shape, ps, seed=71, cell=[cell] * 3, overlap_frac=0.02)
    kw = dict(flagsets.FLAG_SETS[flagset])
    extra = dict(_yx_tiles=yx) if yx else {}
    want_p, want_a = vi.to_instance_seg(c["pred"].copy(), c["foreground"].copy(),
                                        c["foreground"].copy(), c["numinst"].copy(), list(ps),
                                        **dict(kw, return_intermediates=True, _n_slabs=1))
    got_p, got_a = tiling.to_instance_seg_tiled(c["pred"].copy(), c["foreground"].copy(),
                                                c["foreground"].copy(), c["numinst"].copy(), list(ps),
                                                n_slabs, return_intermediates=True, **extra, **kw)
    assert np.array_equal(got_p, want_p)
    assert np.array_equal(got_a.view(np.uint32), want_a.view(np.uint32))
    want, _ = vi.to_instance_seg(c["pred"].copy(), c["foreground"].copy(), c["foreground"].copy(),
                                 c["numinst"].copy(), list(ps), **dict(kw, _n_slabs=1))
    got, _ = tiling.to_instance_seg_tiled(c["pred"].copy(), c["foreground"].copy(),
                                          c["foreground"].copy(), c["numinst"].copy(), list(ps),
                                          n_slabs, **extra, **kw)
    assert np.array_equal(got, want) and got.any()


@pytest.mark.gpu
def test_tiles_vs_oracle_p7_gpu():
    """Tiled 7^3 assembly against the CPU oracle's whole-volume result (shipped flags)."""
    from oracle import ppp_oracle as orc
    from patchperpix_amd.flags import FLYLIGHT as SHIPPED
    shape, ps = (20, 22, 24), (7, 7, 7)
    c = synth.make_case(shape, ps, seed=72, cell=[9, 9, 9], overlap_frac=0.02)
    kw = dict(SHIPPED)
    ref = orc.to_instance_seg(c["pred"], c["foreground"], c["foreground"].copy(), c["numinst"],
                              list(ps), **kw)
    got, _ = tiling.to_instance_seg_tiled(c["pred"].copy(), c["foreground"].copy(),
                                          c["foreground"].copy(), c["numinst"].copy(), list(ps),
                                          2, _yx_tiles=(2, 2), **kw)
    assert np.array_equal(got, ref["instances"]) and got.any()


def test_fused_path_asserts_like_the_references_consensus_stage():
    """consensus_array.py:131-133: counts interleaved with values that are not normalised -- the reference
    asserts where it launches the consensus kernel; the fused assembly (which never calls that stage function)
    must refuse too.  (35 of 300 random flag sets of tests/golden/fuzz_oracle_vs_reference.py were refused by the
    reference; the oracle refused all of them, the fused path missed this assertion.)"""
    import torch
    from oracle_ops import OracleOps
    c, ps, kw = make_case(seed=63, shape=(12, 9, 10))
    kw.update(consensus_interleaved_cnt=True, consensus_norm_aff=False)
    with pytest.raises(AssertionError, match="not normalized"):
        tiling.assemble(torch.from_numpy(c["pred"]), 0, c["foreground"].shape, c["foreground"].copy(), c["foreground"].copy(),
                        c["numinst"], ps, tiling.plan_slabs(12, 1), ops=OracleOps(**dict(kw, consensus_interleaved_cnt=False)), **kw)


def test_tiled_path_refuses_flags_it_does_not_honour():
    """the tiled dispatch must not silently drop flags."""
    import torch
    c, ps, kw = make_case()
    for flag in ("skipRanking", "termAfterThinCover", "one_instance_per_channel"):
        with pytest.raises(NotImplementedError):
            tiling.assemble(torch.from_numpy(c["pred"]), 0, c["foreground"].shape,
                            c["foreground"].copy(), c["foreground"].copy(), c["numinst"], ps,
                            tiling.plan_slabs(c["pred"].shape[1], 2), ops=object(),
                            **dict(kw, **{flag: True}))


GPU_WORKER = r"""
import os, sys
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, {repo!r})
from patchperpix_amd import synth, tiling, backend
from patchperpix_amd import flags as flagsets
FLYLIGHT = flagsets.FLAG_SETS[os.environ.get("PPP_TEST_FLAGSET", "nothin_cc")]
torch.cuda.set_device(0)                      # every rank on the one GPU of the box
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
shape, ps = {shape!r}, {ps!r}
c = synth.make_case(shape, ps, seed=66, cell=[9, 9, 9], overlap_frac=0.02)
if os.environ.get("PPP_TEST_EMPTY_TOP") == "1":      # the last rank's slab holds nothing
    c["pred"][:, 36:] = 0.05; c["foreground"][36:] = False; c["numinst"][36:] = 0
kw = dict(FLYLIGHT)
Z = shape[0]
slabs = tiling.plan_slabs(Z, world)
mine = tiling.slabs_of_rank(slabs, rank, world)
lo, hi = tiling.local_range(mine, Z, ps)
import json
sub = int(os.environ.get("PPP_TEST_SUBSLABS", "1"))        # the rank's range cut into several tiles
if sub > 1:
    a0 = mine[0][0]
    mine = [(a0 + a, a0 + b) for a, b in tiling.plan_slabs(mine[-1][1] - a0, sub)]
kw.update(json.loads(os.environ.get("PPP_TEST_KW", "{{}}")))
if os.environ.get("PPP_TEST_NO_HALO", "0") == "1":    # own slices only: assemble() exchanges the halo
    lo, hi = mine[0][0], mine[-1][1]
pred_local = torch.from_numpy(np.ascontiguousarray(c["pred"][:, lo:hi])).cuda()
if os.environ.get("PPP_TEST_NO_HALO", "0") == "3":    # halo-sized buffer, stale halo: refreshed in place
    pred_local[:, :mine[0][0] - lo] = 0
    pred_local[:, mine[-1][1] - lo:] = 0
    kw["_refresh_halo"] = True
inst, fg = tiling.assemble(pred_local, lo, shape, c["foreground"].copy(), c["foreground"].copy(),
                           c["numinst"], list(ps), mine, comm=tiling.TorchDistComm(), **kw)
np.save(os.path.join({out!r}, "inst_rank%d.npy" % rank), inst)
np.save(os.path.join({out!r}, "notes_rank%d.npy" % rank),
        np.array([backend.NOTES.get("cover_sharded", 0), backend.NOTES.get("cover_rounds", 0),
                  backend.NOTES.get("cover_p2p", 0), backend.NOTES.get("ring_z", 0),
                  1 if "cons_cache_gb" in backend.NOTES else 0,
                  backend.NOTES.get("halo_exchange_bytes_received", 0), backend.NOTES.get("thin_sharded", 0)]))
dist.destroy_process_group()
"""


@pytest.mark.gpu
@pytest.mark.parametrize("world,empty_top,ps,flagset,p2p,sub,extra", [
    (2, False, (5, 5, 5), "nothin_cc", "1", 1, {}), (3, False, (5, 5, 5), "nothin_cc", "1", 1, {}),
    (2, True, (5, 5, 5), "nothin_cc", "1", 1, {}), (2, False, (7, 7, 7), "shipped", "1", 1, {}),
    (3, False, (7, 7, 7), "cc", "1", 1, {}), (3, False, (5, 5, 5), "nothin_cc", "0", 1, {}),
    # a rank's range cut into three tiles: rows in a ring / a consensus cache over the rank's block
    (2, False, (5, 5, 5), "shipped", "1", 3, {"_ring_z": 40}), (2, False, (5, 5, 5), "cc", "1", 3, {"_cons_cache": True}),
    (2, False, (7, 7, 7), "shipped", "1", 3, {"_ring_z": 40, "_yx_tiles": [1, 2]}),
    # round 6: every rank holds its OWN slices of the prediction; the halo comes from the neighbours
    (3, False, (7, 7, 7), "shipped", "1", 1, {"_no_halo": "1"}), (2, False, (5, 5, 5), "cc", "1", 3, {"_no_halo": "1", "_ring_z": 40}),
    (3, False, (5, 5, 5), "shipped", "1", 1, {"_no_halo": "3"})])
def test_ranks_sharing_one_gpu_equal_whole_volume(tmp_path, world, empty_top, ps, flagset, p2p, sub, extra):
    """The multi-rank path with the REAL kernels: `world` processes on the one GPU of the box,
    gloo as the transport (RCCL needs one device per rank): sharded cover with z-halo exchange,
    per-rank pair rows, merged label forests -- same instance map as one process.  The slab
    boundary zones of the cover travel point to point between the two neighbours (p2p = "1") or
    through an all-reduce over all ranks (the form for slabs thinner than the zones)."""
    from patchperpix_amd.vote_instances import vote_instances as vi
    from patchperpix_amd import flags as flagsets
    FLYLIGHT = flagsets.FLAG_SETS[flagset]
    shape = (72, 26, 30)
    c = synth.make_case(shape, ps, seed=66, cell=[9, 9, 9], overlap_frac=0.02)
    if empty_top:
        c["pred"][:, 36:] = 0.05
        c["foreground"][36:] = False
        c["numinst"][36:] = 0
    want, _ = vi.to_instance_seg(c["pred"].copy(), c["foreground"].copy(), c["foreground"].copy(),
                                 c["numinst"].copy(), list(ps), **dict(FLYLIGHT, _n_slabs=1))
    script = tmp_path / "gpu_worker.py"
    script.write_text(GPU_WORKER.format(repo=REPO, out=str(tmp_path), shape=shape, ps=ps))
    _port = _free_port()
    extra = dict(extra)
    no_halo = str(extra.pop("_no_halo", "0"))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=_port, OMP_NUM_THREADS="1",
               PPP_TEST_NO_HALO=no_halo,
               PPP_TEST_EMPTY_TOP="1" if empty_top else "0", PPP_TEST_FLAGSET=flagset,
               PPP_COVER_P2P=p2p, PPP_TEST_SUBSLABS=str(sub), PPP_TEST_KW=__import__("json").dumps(extra))
    subprocess.check_call([sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
                           "--nproc-per-node=%d" % world, "--master-addr", "127.0.0.1",
                           "--master-port", _port, str(script)], env=env, timeout=900)
    assert want.any()
    for r in range(world):
        inst = np.load(tmp_path / ("inst_rank%d.npy" % r))
        assert np.array_equal(inst, want), "rank %d differs" % r
        notes = np.load(tmp_path / ("notes_rank%d.npy" % r))
        assert notes[0] == world and notes[1] > 0 and notes[2] == int(p2p)
        assert notes[3] == extra.get("_ring_z", 0) and notes[4] == (1 if extra.get("_cons_cache") else 0)
        assert (notes[5] > 0) == (no_halo != "0")
        assert notes[6] == (world if not FLYLIGHT.get("skipThinCover", False) else 0)      # thinning sharded too


RCCL_WORKER = r"""
import os, sys
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, {repo!r})
from patchperpix_amd import synth, tiling, backend, flags as flagsets
local = int(os.environ["LOCAL_RANK"])
torch.cuda.set_device(local)                  # one rank per GPU
dist.init_process_group("nccl")               # "nccl" is RCCL on ROCm
rank, world = dist.get_rank(), dist.get_world_size()
shape, ps = (72, 26, 30), (7, 7, 7)
c = synth.make_case(shape, ps, seed=66, cell=[9, 9, 9], overlap_frac=0.02)
kw = dict(flagsets.FLYLIGHT)
slabs = tiling.plan_slabs(shape[0], world)
mine = tiling.slabs_of_rank(slabs, rank, world)
lo, hi = tiling.local_range(mine, shape[0], ps)
if os.environ.get("PPP_TEST_NO_HALO", "0") == "1":    # own slices only: the halo travels over RCCL
    lo, hi = mine[0][0], mine[-1][1]
pred_local = torch.from_numpy(np.ascontiguousarray(c["pred"][:, lo:hi])).cuda()
inst, fg = tiling.assemble(pred_local, lo, shape, c["foreground"].copy(), c["foreground"].copy(),
                           c["numinst"], list(ps), mine, comm=tiling.TorchDistComm(), **kw)
np.save(os.path.join({out!r}, "inst_rank%d.npy" % rank), inst)
assert (backend.NOTES.get("halo_exchange_bytes_received", 0) > 0) == (os.environ.get("PPP_TEST_NO_HALO", "0") == "1")
dist.destroy_process_group()
"""


@pytest.mark.gpu
@pytest.mark.parametrize("no_halo", [False, True])
def test_two_ranks_over_rccl(tmp_path, no_halo):
    """One rank per GPU over RCCL (backend "nccl"): z-slabs with halos, sharded cover with zone
    MIN all-reduces, all-gather of owned score / instance slabs, merged label forests, replicated
    thinning and mutex watershed -- same instance map as one process.  Needs two GPUs in the box
    (skipped on the single-GPU development boxes; the same path runs there over gloo, above).
    no_halo: every rank holds its own slices of the prediction only and the patch-radius halo is
    exchanged over RCCL (tiling.exchange_halo: grouped point-to-point sends / receives)."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    from patchperpix_amd import flags as flagsets
    from patchperpix_amd.vote_instances import vote_instances as vi
    shape, ps = (72, 26, 30), (7, 7, 7)
    c = synth.make_case(shape, ps, seed=66, cell=[9, 9, 9], overlap_frac=0.02)
    want, _ = vi.to_instance_seg(c["pred"].copy(), c["foreground"].copy(), c["foreground"].copy(),
                                 c["numinst"].copy(), list(ps), **dict(flagsets.FLYLIGHT, _n_slabs=1))
    script = tmp_path / "rccl_worker.py"
    script.write_text(RCCL_WORKER.format(repo=REPO, out=str(tmp_path)))
    _port = _free_port()
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=_port, OMP_NUM_THREADS="1",
               HSA_ENABLE_IPC_MODE_LEGACY="0", PPP_TEST_NO_HALO="1" if no_halo else "0")
    subprocess.check_call([sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
                           "--nproc-per-node=2", "--master-addr", "127.0.0.1", "--master-port", _port, str(script)], env=env, timeout=900)
    assert want.any()
    for r in range(2):
        assert np.array_equal(np.load(tmp_path / ("inst_rank%d.npy" % r)), want), "rank %d differs" % r


# ------------------------------------------------------------------------------------------
# prediction PROVIDER (a rank holds one tile + halo at a time) and the SHARDED global stage
# (local fields, sharded sort and cover): BASELINE config [3]'s execution mode
# ------------------------------------------------------------------------------------------
class ArrayProvider:
    """pred_box() from a host array: stands in for a reader / generator of prediction tiles"""

    def __init__(self, pred, device="cpu"):
        self.pred, self.device, self.calls = pred, device, 0

    def pred_box(self, box):
        import torch
        z0, z1, y0, y1, x0, x1 = box
        self.calls += 1
        return torch.from_numpy(np.ascontiguousarray(self.pred[:, z0:z1, y0:y1, x0:x1])).to(self.device)


@pytest.mark.parametrize("n_slabs,yx,thin,mws", [(3, (2, 2), False, False), (2, (1, 2), True, True)])
def test_provider_and_sharded_stage_equal_whole_volume_cpu(n_slabs, yx, thin, mws):
    """One process: prediction through a provider (frames per tile), and the sharded global
    stage (local sort + merge ranks, sharded cover) forced on: same pair rows / instance map."""
    import torch
    from oracle_ops import OracleOps
    c, ps, kw = make_case(seed=63, shape=(24, 17, 19))
    kw.update(skipThinCover=not thin, mws=mws)
    ref = whole_volume(c, ps, kw)
    ops = OracleOps(**kw)
    slabs = tiling.plan_slabs(c["pred"].shape[1], n_slabs)
    shape = c["foreground"].shape
    prov = ArrayProvider(c["pred"])
    for sharded in (False, True):
        extra = dict(_yx_tiles=yx, _sharded_global=sharded)
        pairs, aff = tiling.assemble(prov, 0, shape, c["foreground"].copy(), c["foreground"].copy(),
                                     c["numinst"], ps, slabs, ops=ops, return_intermediates=True,
                                     **extra, **kw)
        assert np.array_equal(pairs, ref["pairs"])
        assert np.array_equal(aff.view(np.uint32), ref["aff"].view(np.uint32))
        inst, fg = tiling.assemble(prov, 0, shape, c["foreground"].copy(), c["foreground"].copy(),
                                   c["numinst"], ps, slabs, ops=ops, **extra, **kw)
        assert np.array_equal(inst, ref["instances"]) and inst.any()
    assert prov.calls > 2 * len(slabs) * yx[0] * yx[1]


WORKER_SHARDED = r"""
import os, sys, json
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, {repo!r}); sys.path.insert(0, os.path.join({repo!r}, "tests"))
from patchperpix_amd import tiling, backend
from test_tiling import make_case, ArrayProvider
from oracle_ops import OracleOps
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
c, ps, kw = make_case()
kw.update(json.loads(os.environ.get("PPP_TEST_KW", "{{}}")))
Z = c["pred"].shape[1]
slabs = tiling.plan_slabs(Z, world)
mine = tiling.slabs_of_rank(slabs, rank, world)
lo, hi = tiling.local_range(mine, Z, ps)
# the rank sees ONLY its slices of the fields, and the prediction through a provider
loc = lambda a: np.ascontiguousarray(a[lo:hi])
gather = os.environ.get("PPP_TEST_GATHER", "1") == "1"
inst, fg = tiling.assemble(ArrayProvider(c["pred"]), lo, c["foreground"].shape, loc(c["foreground"]),
                           loc(c["foreground"]).copy(), loc(c["numinst"]), ps, mine,
                           comm=tiling.TorchDistComm(), ops=OracleOps(**kw), _yx_tiles=(1, 2),
                           _gather_result=gather, **kw)
np.save(os.path.join({out!r}, "inst_rank%d.npy" % rank), inst)
np.save(os.path.join({out!r}, "fg_rank%d.npy" % rank), fg)
np.save(os.path.join({out!r}, "notes_rank%d.npy" % rank),
        np.array([backend.NOTES.get("cover_sharded", 0), backend.NOTES.get("ranked_own", 0), mine[0][0], mine[-1][1]]))
dist.destroy_process_group()
"""


@pytest.mark.parametrize("world,extra,gather", [
    (2, {}, True), (3, {}, False),
    (2, {"skipThinCover": False, "mws": True}, False),
    (2, {"select_patches_for_sparse_data": False, "skipThinCover": False}, True),
    (2, {"_empty_top": True}, True)])
def test_ranks_gloo_provider_local_fields(tmp_path, world, extra, gather, monkeypatch):
    """Two / three processes over gloo, each with its own slices of the fields only and the
    prediction through a provider: sharded sort (merge ranks from the gathered sorted scores),
    sharded cover, gathered selection, replicated thinning on the gathered mask -- the own slab
    of the result (or the gathered whole) equals the whole-volume result of one process."""
    import json
    extra = dict(extra)
    if extra.pop("_empty_top", False):
        monkeypatch.setenv("PPP_TEST_EMPTY_TOP", "1")
    c, ps, kw = make_case()
    kw.update(extra)
    ref = whole_volume(c, ps, kw)
    script = tmp_path / "worker.py"
    script.write_text(WORKER_SHARDED.format(repo=REPO, out=str(tmp_path)))
    _port = _free_port()
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=_port, OMP_NUM_THREADS="1",
               PPP_TEST_KW=json.dumps(extra), PPP_TEST_GATHER="1" if gather else "0")
    subprocess.check_call([sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
                           "--nproc-per-node=%d" % world, "--master-addr", "127.0.0.1",
                           "--master-port", _port, str(script)], env=env, timeout=900)
    assert ref["instances"].any()
    n_ranked = 0
    for r in range(world):
        inst = np.load(tmp_path / ("inst_rank%d.npy" % r))
        fg = np.load(tmp_path / ("fg_rank%d.npy" % r))
        notes = np.load(tmp_path / ("notes_rank%d.npy" % r))
        z0, z1 = int(notes[2]), int(notes[3])
        want = ref["instances"] if gather else ref["instances"][z0:z1]
        assert np.array_equal(inst, want), "rank %d differs" % r
        assert np.array_equal(fg != 0, c["foreground"] if gather else c["foreground"][z0:z1])
        assert notes[0] == world
        n_ranked += int(notes[1])
    assert n_ranked == len(ref["ranked_coords"])       # every patch was ranked by exactly one rank


@pytest.mark.gpu
@pytest.mark.parametrize("ps,shape,cell,n_slabs,yx,flagset", [
    ((5, 5, 5), (30, 34, 38), 8, 2, (2, 2), "nothin_cc"),
    ((7, 7, 7), (30, 34, 38), 10, 2, (2, 2), "shipped"),
    ((9, 9, 9), (40, 30, 34), 11, 2, (1, 2), "shipped"),
])
def test_provider_and_sharded_stage_gpu(ps, shape, cell, n_slabs, yx, flagset):
    """The real kernels through a prediction provider (a frame per tile and pass, origins in all
    three axes) and with the sharded global stage forced on: same result as the untiled path."""
    from patchperpix_amd import flags as flagsets
    from patchperpix_amd.vote_instances import vote_instances as vi
    c = synth.make_case(shape, ps, seed=73, cell=[cell] * 3, overlap_frac=0.02)
    kw = dict(flagsets.FLAG_SETS[flagset])
    want, _ = vi.to_instance_seg(c["pred"].copy(), c["foreground"].copy(), c["foreground"].copy(),
                                 c["numinst"].copy(), list(ps), **dict(kw, _n_slabs=1))
    assert want.any()
    slabs = tiling.plan_slabs(shape[0], n_slabs)
    prov = ArrayProvider(c["pred"].astype(np.float16), device="cuda")
    for sharded in (False, True):
        got, _ = tiling.assemble(prov, 0, shape, c["foreground"].copy(), c["foreground"].copy(),
                                 c["numinst"].copy(), list(ps), slabs, _yx_tiles=yx,
                                 _sharded_global=sharded, **kw)
        assert np.array_equal(got, want), "sharded=%s" % sharded


GPU_WORKER_SHARDED = r"""
import os, sys
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, {repo!r}); sys.path.insert(0, os.path.join({repo!r}, "tests"))
from patchperpix_amd import synth, tiling, backend
from patchperpix_amd import flags as flagsets
from test_tiling import ArrayProvider
torch.cuda.set_device(0)                      # every rank on the one GPU of the box
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
shape, ps = {shape!r}, {ps!r}
c = synth.make_case(shape, ps, seed=66, cell=[9, 9, 9], overlap_frac=0.02)
kw = dict(flagsets.FLAG_SETS[os.environ.get("PPP_TEST_FLAGSET", "shipped")])
Z = shape[0]
slabs = tiling.plan_slabs(Z, world)
mine = tiling.slabs_of_rank(slabs, rank, world)
lo, hi = tiling.local_range(mine, Z, ps)
loc = lambda a: np.ascontiguousarray(a[lo:hi])
inst, fg = tiling.assemble(ArrayProvider(c["pred"].astype(np.float16), device="cuda"), lo, shape,
                           loc(c["foreground"]), loc(c["foreground"]).copy(), loc(c["numinst"]),
                           list(ps), mine, comm=tiling.TorchDistComm(), _yx_tiles=(2, 1),
                           _gather_result=False, _instances_dtype=np.uint32, **kw)
np.save(os.path.join({out!r}, "inst_rank%d.npy" % rank), inst)
np.save(os.path.join({out!r}, "range_rank%d.npy" % rank), np.array([mine[0][0], mine[-1][1]]))
dist.destroy_process_group()
"""


@pytest.mark.gpu
@pytest.mark.parametrize("world,ps,flagset", [(2, (7, 7, 7), "shipped"), (3, (5, 5, 5), "cc")])
def test_ranks_provider_local_fields_share_one_gpu(tmp_path, world, ps, flagset):
    """BASELINE config [3]'s execution mode with the real kernels: `world` processes (sharing the
    one GPU, gloo transport), each with its own slices of the fields, the prediction through a
    provider, sharded sort / cover, uint32 ids, own-slab results: stacked, they equal the
    instance map of one process."""
    from patchperpix_amd.vote_instances import vote_instances as vi
    from patchperpix_amd import flags as flagsets
    shape = (72, 26, 30)
    c = synth.make_case(shape, ps, seed=66, cell=[9, 9, 9], overlap_frac=0.02)
    want, _ = vi.to_instance_seg(c["pred"].copy(), c["foreground"].copy(), c["foreground"].copy(),
                                 c["numinst"].copy(), list(ps),
                                 **dict(flagsets.FLAG_SETS[flagset], _n_slabs=1))
    script = tmp_path / "gpu_worker_sharded.py"
    script.write_text(GPU_WORKER_SHARDED.format(repo=REPO, out=str(tmp_path), shape=shape, ps=ps))
    _port = _free_port()
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=_port, OMP_NUM_THREADS="1",
               PPP_TEST_FLAGSET=flagset)
    subprocess.check_call([sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
                           "--nproc-per-node=%d" % world, "--master-addr", "127.0.0.1",
                           "--master-port", _port, str(script)], env=env, timeout=900)
    assert want.any()
    for r in range(world):
        z0, z1 = [int(v) for v in np.load(tmp_path / ("range_rank%d.npy" % r))]
        inst = np.load(tmp_path / ("inst_rank%d.npy" % r))
        assert inst.dtype == np.uint32 and np.array_equal(inst, want[z0:z1]), "rank %d differs" % r


COMM_WORKER = r"""
import os, sys
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, {repo!r})
from patchperpix_amd import tiling
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
comm = tiling.TorchDistComm()
ok = []
# 16-bit integers and booleans are not element types RCCL moves: they travel as int32 / bytes
a = torch.tensor([1, -2, 300 * (rank + 1)], dtype=torch.int16)
ok.append(comm.all_reduce_sum(a.clone()).tolist() == [world, -2 * world, 300 * world * (world + 1) // 2])
ok.append(comm.all_reduce_max(a.clone()).tolist() == [1, -2, 300 * world])
b = torch.tensor([rank == 0, True, False])
g = comm.all_gather(b)
ok.append(g.dtype == torch.bool and g.shape == (world, 3) and g[:, 0].tolist() == [r == 0 for r in range(world)])
h = comm.all_gather(torch.arange(4, dtype=torch.int16) + 10 * rank)
ok.append(h.dtype == torch.int16 and h.tolist() == [[10 * r + i for i in range(4)] for r in range(world)])
peer = (rank + 1) % world
t = torch.tensor([5 + rank, 7 - rank], dtype=torch.int16)
comm.neighbour_min([(peer, t)])
ok.append(t.tolist() == [5, 7 - (world - 1)])
vol = torch.zeros((2 * world, 3), dtype=torch.int16)
vol[2 * rank:2 * rank + 2] = rank + 1
comm.all_gather_slabs(vol, [(2 * r, 2 * r + 2) for r in range(world)])
ok.append(vol[:, 0].tolist() == [r + 1 for r in range(world) for _ in range(2)])
np.save(os.path.join({out!r}, "comm_rank%d.npy" % rank), np.array(ok))
dist.destroy_process_group()
"""


def test_communicator_moves_types_rccl_has_no_element_type_for(tmp_path):
    """TorchDistComm: int16 / bool tensors are reduced as int32 and gathered / exchanged as bytes
    (torch's NCCL = RCCL backend maps no 16-bit integer type); two gloo ranks."""
    script = tmp_path / "comm_worker.py"
    script.write_text(COMM_WORKER.format(repo=REPO, out=str(tmp_path)))
    _port = _free_port()
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=_port, OMP_NUM_THREADS="1")
    subprocess.check_call([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                           "--master-addr", "127.0.0.1", "--master-port", _port, str(script)], env=env, timeout=300)
    for r in range(2):
        assert np.load(tmp_path / ("comm_rank%d.npy" % r)).all(), "rank %d" % r


def test_rows_by_tile_equals_a_scan_per_tile():
    """tiling.rows_by_tile: one sort instead of a scan of the pair list per tile -- same indices,
    ascending, for a slab x (y, x) grid; None for tiles with gaps (the caller scans then)."""
    import torch
    from patchperpix_amd import tiling
    rng = np.random.default_rng(0)
    rows = torch.from_numpy(rng.integers(0, [20, 30, 40, 20, 30, 40], size=(5000, 6)).astype(np.int32))
    tiles = [(z0, z1) + t for (z0, z1) in [(4, 8), (8, 20)] for t in tiling.plan_yx(30, 40, 2, 3)]
    m = tiling.rows_by_tile(rows, tiles)
    for n, (z0, z1, y0, y1, x0, x1) in enumerate(tiles):
        own = (rows[:, 0] >= z0) & (rows[:, 0] < z1) & (rows[:, 1] >= y0) & (rows[:, 1] < y1) & \
              (rows[:, 2] >= x0) & (rows[:, 2] < x1)
        assert torch.equal(torch.nonzero(own).reshape(-1), m[n])
    assert tiling.rows_by_tile(rows, [(0, 8, 0, 30, 0, 40), (10, 20, 0, 30, 0, 40)]) is None


# configurations the random search of tests/fuzz_ranks_cpu.py found failing (round 6), kept as fixed cases:
FUZZ_REGRESSIONS = [
    # a middle rank whose slices + halo are the WHOLE volume next to neighbours with local fields: the ranks
    # took different paths (sharded / replicated global stage) and their collectives did not match
    (3, {"shape": [21, 8, 14], "ps": [3, 3, 3], "seed": 6830, "cell": [4, 4, 4], "overlap": 0.02,
         "flags": {"skipThinCover": False, "mws": True}, "n_slabs": 3, "mode": "provider_local_fields", "extra": {"_yx_tiles": [1, 2]}}),
    # a provider and ONE tile per rank (the consensus is kept between the passes): the kept frame ended where the
    # scores pass stops reading, short of the windows of the partner patches the patch-graph stage looks at
    (4, {"shape": [29, 15, 11], "ps": [3, 3, 3], "seed": 5582, "cell": [4, 4, 4], "overlap": 0.0,
         "flags": {"skipThinCover": True, "mws": False}, "n_slabs": 4, "mode": "provider", "extra": {}}),
    (4, {"shape": [29, 15, 11], "ps": [3, 5, 3], "seed": 5582, "cell": [4, 4, 4], "overlap": 0.0,
         "flags": {"skipThinCover": False, "mws": False}, "n_slabs": 4, "mode": "provider",
         "extra": {"_cover_chunk": 804, "_gather_result": False}}),
]


@pytest.mark.parametrize("world,cfg", FUZZ_REGRESSIONS)
def test_ranks_gloo_fuzz_regressions(world, cfg):
    import json
    r = subprocess.run([sys.executable, os.path.join(REPO, "tests", "fuzz_ranks_cpu.py"), "--world", str(world), "--cfg", json.dumps(cfg)],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900)
    out = r.stdout.decode(errors="replace")
    assert r.returncode == 0 and ": ok" in out, out[-3000:]
