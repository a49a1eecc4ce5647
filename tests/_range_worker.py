"""Rank body of the frame-range sharding tests: ONE stream, cut at its seek points into one range per rank; every
rank decodes its range into a region of its own, the result rows are gathered, and the stream's MD5 is finished on
rank 0 from the ranges' PCM.  argv: out.json [gpu].  Without "gpu" the oracle stands in for the GPU (this container
has none); with it every rank decodes on cuda:0 through the C ABI (a rehearsal of the N > 1 path on a one-GPU box;
the ranks still talk over gloo, which needs CPU tensors)."""
import ctypes
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from util import flac_amd, oracle, split_file, synth, to_oracle_si  # noqa: E402

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402
from flac_amd import sharding  # noqa: E402


def main():
    out, use_gpu = sys.argv[1], len(sys.argv) > 2 and sys.argv[2] == "gpu"
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    ch, bps = 8, 24
    x = synth.pcm(77, 2048 * 41 + 123, ch, bps, 192000, 2048)
    f, _ = synth.encode(x, bps, 192000, synth.options(2048, 12, seek_every_frames=2))
    f = bytes(f)
    md = flac_amd.parse_metadata(f)
    seek = [b for b in md.blocks[: md.n_blocks] if b.type == 4][0]
    pts = flac_amd.parse_seektable(f[seek.payload_offset: seek.payload_offset + seek.length])
    si, audio = split_file(f)
    shards = sharding.range_shards_with_samples(pts, len(audio), si.total_samples, world)
    mine = sharding.my_range(shards, rank, world)
    frame_bytes = ch * ((bps + 7) // 8)
    if mine is None:
        pcm, row = np.zeros(0, dtype=np.uint8), None
    else:
        off, n, base, nsamp = mine
        cap = nsamp * frame_bytes
        if use_gpu:
            dec = flac_amd.Decoder(0)
            d_in = torch.frombuffer(bytearray(audio) + bytearray(64), dtype=torch.uint8).cuda()
            d_out = torch.zeros(cap + 64, dtype=torch.uint8, device="cuda")
            res = dec.decode_batch_device(d_in, len(audio), d_out, cap + 64, dec.make_streams([(off, n, 0, cap, si, base)]))
            torch.cuda.synchronize()
            pcm, row = d_out[:cap].cpu().numpy(), res[0]
            dec.close()
        else:
            pcm, row, _ = oracle.decode_range(audio[off: off + n], to_oracle_si(si), sample_base=base, pcm_cap=cap)
            assert pcm.size == cap
    rows = sharding.result_rows([rank] if row is not None else [], [row] if row is not None else [])
    table = sharding.gather_rows(rows, min(world, len(shards)))
    digest = sharding.chained_md5(torch.from_numpy(np.ascontiguousarray(pcm)))
    if rank == 0:
        json.dump({"world": world, "shards": shards, "table": table.tolist(), "md5": digest.hex(),
                   "streaminfo_md5": si.md5_signature, "total_samples": int(si.total_samples)}, open(out, "w"))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
