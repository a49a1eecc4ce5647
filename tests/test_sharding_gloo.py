"""The multi-GPU path is file sharding with a gather of small result rows.  Here it runs at world size 2
over gloo on CPU (torch.distributed.run, one process per rank, as bench.py is launched on a GPU node)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from flac_amd import sharding

HERE = os.path.dirname(os.path.abspath(__file__))


def test_shard_indices_partition():
    for n in (0, 1, 7, 8, 10000):
        for world in (1, 2, 3, 8):
            seen = sorted(i for r in range(world) for i in sharding.shard_indices(n, r, world))
            assert seen == list(range(n))
            sizes = [len(sharding.shard_indices(n, r, world)) for r in range(world)]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        sharding.shard_indices(4, 2, 2)


def _run(world, out, port, worker="_shard_worker.py", *extra):
    env = dict(os.environ, OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(HERE, worker), out, *extra]
    subprocess.run(cmd, check=True, env=env, timeout=240, cwd=os.path.dirname(HERE))
    return json.load(open(out))


def test_two_ranks_equal_one(tmp_path):
    one = _run(1, str(tmp_path / "w1.json"), 29611)
    two = _run(2, str(tmp_path / "w2.json"), 29612)
    assert two["world"] == 2 and two["mine"] == list(range(0, len(one["table"]), 2))
    t1, t2 = np.array(one["table"]), np.array(two["table"])
    assert np.array_equal(t1, t2)
    assert (t1[:, 2] == t1[:, 1]).all() and (t1[:, 1] > 0).all()   # every frame of every stream decoded


def test_range_shards_cover_the_stream():
    class P:
        def __init__(self, s, o):
            self.sample_number, self.offset = s, o
    pts = [P(i * 4096, i * 1000) for i in range(1, 40)] + [P(0xFFFFFFFFFFFFFFFF, 0)]
    for parts in (1, 2, 3, 8, 100):
        sh = sharding.range_shards(pts, 40000, parts)
        assert sh[0][0] == 0 and sum(n for _, n in sh) == 40000
        assert all(a + n == b for (a, n), (b, _) in zip(sh[:-1], sh[1:]))
        assert all(a % 1000 == 0 for a, _ in sh)
        assert len(sh) <= parts
    assert sharding.range_shards([], 500, 4) == [(0, 500)]
    assert len(sharding.range_shards(pts, 40000, 8)) == 8


def check_range_run(r, world):
    """What a frame-range run must show: the ranges tile the stream, every rank decoded every frame of its range,
    the samples add up, and the MD5 finished on rank 0 is the stream's."""
    shards = r["shards"]
    assert r["world"] == world and len(shards) == world
    assert shards[0][0] == 0 and shards[0][2] == 0
    assert all(a[0] + a[1] == b[0] and a[2] + a[3] == b[2] for a, b in zip(shards[:-1], shards[1:]))
    assert sum(s[3] for s in shards) == r["total_samples"]
    t = np.array(r["table"])
    assert (t[:, 1] == t[:, 2]).all() and (t[:, 1] > 0).all() and (t[:, 4] == 0).all()
    assert [int(v) for v in t[:, 3]] == [s[3] for s in shards]
    assert r["md5"] == r["streaminfo_md5"]


def test_one_stream_cut_by_frame_range_over_two_ranks(tmp_path):
    """One 24-bit 8-channel stream, split at seek points (flacmeta.go:612-641) over the ranks; the chained MD5 of
    the stream is finished on rank 0 from the ranges' PCM and equals STREAMINFO's."""
    check_range_run(_run(1, str(tmp_path / "r1.json"), 29613, "_range_worker.py"), 1)
    check_range_run(_run(2, str(tmp_path / "r2.json"), 29614, "_range_worker.py"), 2)
    check_range_run(_run(3, str(tmp_path / "r3.json"), 29615, "_range_worker.py"), 3)


def test_range_shards_with_samples():
    class P:
        def __init__(self, s, o):
            self.sample_number, self.offset = s, o
    pts = [P(i * 4096, i * 1000) for i in range(0, 40)] + [P(0xFFFFFFFFFFFFFFFF, 0)]
    sh = sharding.range_shards_with_samples(pts, 40000, 40 * 4096 - 7, 4)
    assert len(sh) == 4 and sh[0][:3] == (0, 10000, 0)
    assert all(b == o // 1000 * 4096 for o, _, b, _ in sh)
    assert sum(n for _, _, _, n in sh) == 40 * 4096 - 7
    assert sharding.my_range(sh, 3, 4) == sh[3] and sharding.my_range(sh[:2], 3, 4) is None
