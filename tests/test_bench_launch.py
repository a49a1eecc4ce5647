"""bench.py --gpus N starts its own ranks: the command it builds, its refusal on a node with fewer GPUs (CPU),
and a two-rank run through the whole bench on one GPU in rehearsal mode (GPU)."""
import json
import os
import subprocess
import sys

import pytest

from util import ROOT

sys.path.insert(0, ROOT)
import bench  # noqa: E402


def test_launch_command_is_what_the_driver_runs():
    cmd = bench.launch_command(4, ["--gpus", "4", "--steps", "3"], 29512)
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"]
    assert "--nproc-per-node=4" in cmd and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[cmd.index("--master-port") + 1] == "29512"
    assert cmd[-5:] == [os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "3"]


def test_presets_name_the_baseline_configurations():
    assert bench.PRESETS["cfg3"] == {}
    c = bench.PRESETS["cfg4"]
    assert (c["rate"], c["channels"], c["bps"], c["blocksize"], c["force_lpc_order"], c["force_partition_order"]) == \
        (192000, 8, 24, 16384, 32, 8)
    assert c["files"] * c["seconds"] * c["rate"] * c["channels"] * 3 >= 8e9      # at least 8 GB of PCM per GPU


def test_committed_profiles_feed_the_roofline_fields():
    """The bench line's traffic, instruction and clock figures come from committed counter profiles keyed by the
    workload: the default workload must find all of them, and an unknown workload none (null fields, not stale ones)."""
    tag = "44100Hz_16bit_2ch_bs4096_lpc12_force0_po6_fpo-1_60s"
    kernel = "k_decode<12, 3, 1>"
    traffic, detail = bench.lookup_traffic(tag, kernel, 10000)
    assert traffic and detail["read_side_used"].startswith("32 x") and detail["write_bytes"] > 1.05e11
    assert 1.0 < traffic / (detail["write_bytes"] + 6.1e10) < 1.5          # reads: between once and twice the input
    per_step, src = bench.lookup_valu(tag, kernel)
    assert 25 < per_step < 40 and src.startswith("profiles/")
    clock, csrc = bench.lookup_clock(tag, kernel)
    assert 1500 < clock < 2400 and csrc.startswith("profiles/")
    v = bench.valu_issue(tag, kernel, 5.29e10, 0.0521, {"compute_units": 256, "clock_MHz": None})
    assert 0.7 < v["frac"] < 0.9 and 0.8 < v["frac_at_measured_clock"] < 1.1
    enc, esrc = bench.lookup_encode_traffic(tag, 10000)
    assert set(enc) >= {"k_enc_pick", "k_enc_stats", "k_enc_choose", "k_enc_pack"} and esrc.startswith("profiles/")
    assert all(0.95 < enc[k]["read"] / 1.0584e11 < 1.1 for k in ("k_enc_pick", "k_enc_stats", "k_enc_choose", "k_enc_pack"))
    assert bench.lookup_traffic("no such workload", kernel, 10)[0] is None
    assert bench.lookup_valu("no such workload", kernel) == (None, None)
    assert bench.lookup_clock("no such workload", kernel) == (None, None)
    assert bench.lookup_encode_traffic("no such workload", 10) == (None, None)


def test_more_gpus_than_the_node_has_is_refused():
    import torch
    have = torch.cuda.device_count()
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "FLAC_BENCH_REHEARSE")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(have + 2)], env=env,
                       capture_output=True, text=True, timeout=300)
    assert p.returncode == 2 and "not measurable here" in p.stderr and not p.stdout.strip()


def test_world_size_must_match_gpus():
    env = dict(os.environ, WORLD_SIZE="3", RANK="0", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert p.returncode == 2 and "WORLD_SIZE is 3" in p.stderr


def test_launcher_world_size_is_adopted_when_gpus_is_not_given():
    assert bench.resolve_gpus(None, {}) == (1, None)
    assert bench.resolve_gpus(None, {"WORLD_SIZE": "8"}) == (8, None)       # plain torchrun --nproc-per-node=8 bench.py
    assert bench.resolve_gpus(4, {"WORLD_SIZE": "4"}) == (4, None)
    assert bench.resolve_gpus(2, {}) == (2, None)                           # bench.py starts its own ranks
    n, why = bench.resolve_gpus(2, {"WORLD_SIZE": "3"})
    assert n == 2 and "WORLD_SIZE is 3" in why


@pytest.mark.gpu
def test_two_ranks_started_by_bench_itself():
    """`bench.py --gpus 2` with no launcher around it: two ranks (both on GPU 0, gloo for the timing scalars: a
    one-GPU box cannot run two RCCL ranks), one JSON line, n_gpus 2, one timing per rank."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["FLAC_BENCH_REHEARSE"] = "1"
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--files", "24", "--unique", "4",
                        "--seconds", "2", "--steps", "2", "--warmup", "1", "--cpu-files", "2", "--cpu-seconds", "0.2",
                        "--encode-steps", "1", "--host-files", "6", "--cfg4-files", "0"], env=env, capture_output=True, text=True,
                       timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and len(out["rank_ms_per_step"]) == 2 and out["scaling"] == "weak"
    assert out["config"]["files_per_gpu"] == 24 and out["value"] > 0
    assert 0 < out["roofline"]["pipeline_frac"] <= out["roofline"]["frac"] < 1
    assert out["pcie_inclusive"]["files"] == 6 and 0 < out["pcie_inclusive"]["value"] < out["value"]
    assert out["cpu_baseline"]["cores"] >= 1 and out["host"]["cpu_count"] >= 1 and out["device"]["compute_units"] > 0
    assert "configs" not in out          # --cfg4-files 0


@pytest.mark.gpu
def test_cfg4_leg_rides_on_the_headline_line():
    """The default command appends the 24-bit 8-channel configuration as configs.cfg4 with its own roofline,
    CPU baseline and MD5 check (here at a few files)."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "FLAC_BENCH_REHEARSE")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--files", "24", "--unique", "4", "--seconds", "2",
                        "--steps", "2", "--warmup", "1", "--cpu-files", "2", "--cpu-seconds", "0.2", "--encode-steps", "0",
                        "--host-files", "0", "--cfg4-files", "6"], env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    out = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][0])
    c4 = out["configs"]["cfg4"]
    assert out["config"]["name"] == "cfg3" and c4["config"]["name"] == "cfg4" and c4["config"]["files_per_gpu"] == 6
    assert c4["value"] > 0 and c4["md5_verified"]["streams"] == 6 and c4["cpu_baseline"]["value"] > 0
    assert c4["roofline"]["kernel"].startswith("k_decode<32, 2") and 0 < c4["roofline"]["frac"] < 1


@pytest.mark.gpu
def test_one_stream_cut_over_two_ranks_by_the_bench():
    """`bench.py --shard range --gpus 2`: one 24-bit 8-channel stream, a frame range per rank, MD5 finished on rank 0."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["FLAC_BENCH_REHEARSE"] = "1"
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--config", "cfg4", "--shard", "range", "--gpus", "2",
                        "--seconds", "6", "--steps", "2", "--warmup", "1", "--seek-frames", "4"],
                       env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    out = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][0])
    assert out["n_gpus"] == 2 and out["scaling"] == "strong" and len(out["config"]["ranges"]) == 2
    assert out["md5_verified"]["ranges"] == 2 and out["value"] > 0 and len(out["rank_ms_per_step"]) == 2
