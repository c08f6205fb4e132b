"""Multi-rank Problem schedule on CPU (gloo, world_size 2) with the oracle standing
in for the device: sharding, the all-gather of depth maps between passes, and the
Jacobi semantics (results independent of the number of ranks)."""
import importlib
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
W, H, NX, NY = 48, 32, 3, 2


def _scene():
    pm = importlib.import_module("mp-mvs_amd")
    sc, neigh = pm.synth.make_grid_scene(W, H, NX, NY, spacing=0.5, rot_deg=1.0)
    cams = [v.cam for v in sc.views]
    imgs = [v.image for v in sc.views]
    return cams, imgs, neigh


def _run(rank, world, dist, kw):
    from oracle import binding as ob
    sched = importlib.import_module("mp-mvs_amd.schedule")
    ob.set_num_threads(2)
    cams, imgs, neigh = _scene()
    s = sched.SceneScheduler(cams, imgs, neigh, ob.create, rank=rank, world=world, dist=dist, max_scale=0, workers=2 if world > 1 else 1)
    res = s.run(**kw)
    return {i: (r[0], r[1]) for i, r in res.items()}, s.depth_maps()


def _worker(rank, world, port, kw, outdir):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    res, all_depths = _run(rank, world, dist, kw)
    np.savez(os.path.join(outdir, f"rank{rank}.npz"), all_depths=all_depths,
             **{f"planes{i}": r[0] for i, r in res.items()}, **{f"costs{i}": r[1] for i, r in res.items()})
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,kw", [(2, dict(geom_iterations=2, planar_prior=True, geom_planar_prior=True, seed=777)),
                                      (2, dict(geom_iterations=1, planar_prior=False, geom_planar_prior=False, seed=5)),
                                      # 6 Problems over 4 ranks: uneven shards (2, 2, 1, 1), padded all-gather slots
                                      (4, dict(geom_iterations=2, planar_prior=True, geom_planar_prior=True, seed=31))])
def test_ranks_equal_one_rank(tmp_path, world, kw):
    import torch.multiprocessing as mp
    sched = importlib.import_module("mp-mvs_amd.schedule")
    assert sched.owned_problems(6, 0, 2) == [0, 2, 4] and sched.owned_problems(6, 1, 2) == [1, 3, 5]
    assert [sched.owned_problems(6, r, 4) for r in range(4)] == [[0, 4], [1, 5], [2], [3]]
    ref, ref_depths = _run(0, 1, None, kw)
    assert sorted(ref) == list(range(NX * NY))
    port = 29500 + (os.getpid() % 2000) + world
    mp.spawn(_worker, args=(world, port, kw, str(tmp_path)), nprocs=world, join=True)
    seen = set()
    for rank in range(world):
        z = np.load(os.path.join(str(tmp_path), f"rank{rank}.npz"))
        assert np.array_equal(z["all_depths"], ref_depths), "every rank holds every depth map after the barrier"
        for i in sched.owned_problems(NX * NY, rank, world):
            assert np.array_equal(z[f"planes{i}"], ref[i][0]) and np.array_equal(z[f"costs{i}"], ref[i][1]), f"problem {i}"
            seen.add(i)
    assert seen == set(range(NX * NY))
    # the geometric passes did something sensible: depths near ground truth
    pm = importlib.import_module("mp-mvs_amd")
    sc, _ = pm.synth.make_grid_scene(W, H, NX, NY, spacing=0.5, rot_deg=1.0)
    gt = sc.views[0].gt_depth
    assert (np.abs(ref[0][0][..., 3] - gt) / gt < 0.1).mean() > 0.6


# ---------------------------------------------------------------------------------------------------------------------
# bench.py started plainly with --gpus N > 1 launches its own ranks (no torch.distributed.run, no exec after GPU init)
# ---------------------------------------------------------------------------------------------------------------------
def _bench_env():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["OMP_NUM_THREADS"] = "1"
    return env


def test_bench_self_launch_two_ranks():
    import json
    import subprocess
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--launch-check"], env=_bench_env(),
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["max_rank_plus_1"] == 2.0 and out["local_rank"] == 0 and out["master"].startswith("127.0.0.1:")


def test_bench_self_launch_propagates_a_failed_rank():
    import subprocess
    # rank 1 dies before the collective: rank 0 would wait in it forever; the launcher must stop it and report failure
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--launch-check", "--launch-check-fail-rank", "1"],
                       env=_bench_env(), capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert "rank 1 exited with code 3" in r.stderr
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]


def test_bench_rank_count_mismatch_is_refused():
    import subprocess
    env = dict(_bench_env(), RANK="0", LOCAL_RANK="0", WORLD_SIZE="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE=1" in r.stderr
