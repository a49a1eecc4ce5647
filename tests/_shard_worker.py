"""Rank body of test_sharding_gloo.py: decodes this rank's shard (the oracle stands in for the GPU, which
this container lacks), gathers the result rows over gloo, rank 0 writes the table."""
import hashlib
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from cases import CASES, encode_case  # noqa: E402
from util import oracle_decode, split_file  # noqa: E402

import torch.distributed as dist  # noqa: E402
from flac_amd import sharding  # noqa: E402


def main():
    out = sys.argv[1]
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    picks = [i for i, c in enumerate(CASES) if c["n"] <= 9000][:9]
    mine = sharding.shard_indices(len(picks), rank, world)
    results, digests = [], []
    for j in mine:
        _, f, _ = encode_case(picks[j], CASES[picks[j]])
        si, a = split_file(f)
        pcm, st = oracle_decode(si, a)
        results.append(st)
        digests.append(hashlib.md5(pcm.tobytes()).digest())
    table = sharding.gather_rows(sharding.result_rows(list(mine), results, digests), len(picks))
    if rank == 0:
        json.dump({"world": world, "table": table.tolist(), "mine": list(mine)}, open(out, "w"))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
