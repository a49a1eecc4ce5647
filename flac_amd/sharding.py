"""Sharding of the decode path over the GPUs of one node: by file, and by frame range of one long stream.

FLAC files are independent and every frame is self-contained, so the decode path has no exchange
step: rank r of W takes files r, r+W, r+2W, ... (round robin keeps shards balanced when file sizes
drift along the list), decodes them on its own GPU with its own handle, and only the small per-file
result rows travel afterwards.  One process per GPU, torch.distributed for the rows (RCCL on GPUs,
gloo on CPU); nothing on the data path is a collective.

A single long stream is cut at seek points instead (range_shards_with_samples): rank r decodes the r-th
frame range into an output region of its own (flacgpu_stream.sample_base = the range's first sample).
The one thing that cannot be done per range is the stream's MD5, a chain over all its bytes in order:
chained_md5 moves the ranges' PCM to rank 0 (point-to-point, rank order = stream order) and finishes the
digest there.
"""
import numpy as np

ROW_FIELDS = ("index", "frames", "frames_ok", "samples", "first_error", "digest_lo", "digest_hi")


def shard_indices(n_items: int, rank: int, world: int) -> range:
    if world < 1 or not 0 <= rank < world:
        raise ValueError(f"rank {rank} of {world}")
    return range(rank, n_items, world)


def range_shards(seekpoints, audio_len: int, parts: int):
    """Cuts ONE stream into `parts` contiguous frame ranges at seek points (SeekpointBlock.Offset, bytes from
    frame 0, flacmeta.go:256-260: every seek point is the first byte of a frame), as evenly by bytes as the
    points allow.  Returns [(offset, length)] covering [0, audio_len) in order; ranges may be fewer than
    `parts` when the table is short.  Placeholder points (sample number 2^64 - 1) are ignored.  Every frame
    carries its own frame or sample number, so each range decodes on its own, on any GPU, into the same
    output region as the others."""
    if parts < 1:
        raise ValueError("parts must be at least 1")
    offs = sorted({int(p.offset) for p in seekpoints
                   if int(p.sample_number) != 0xFFFFFFFFFFFFFFFF and 0 < int(p.offset) < audio_len})
    cuts = [0]
    for k in range(1, parts):
        want = audio_len * k // parts
        best = min(offs, key=lambda o: abs(o - want), default=None)
        if best is not None and best > cuts[-1]:
            cuts.append(best)
    cuts.append(audio_len)
    return [(a, b - a) for a, b in zip(cuts[:-1], cuts[1:]) if b > a]


def result_rows(indices, results, digests=None) -> np.ndarray:
    """One int64 row per stream: global index, the counters of flacgpu_stream_result, and (optionally)
    a 128-bit digest of its PCM split in two words (e.g. the MD5 the caller computed)."""
    rows = np.zeros((len(indices), len(ROW_FIELDS)), dtype=np.int64)
    for j, (i, r) in enumerate(zip(indices, results)):
        rows[j, :5] = (i, r.frames, r.frames_ok, r.samples, r.first_error)
        if digests is not None:
            d = digests[j]
            rows[j, 5] = int.from_bytes(d[:8], "little", signed=True)
            rows[j, 6] = int.from_bytes(d[8:16], "little", signed=True)
    return rows


def gather_rows(rows: np.ndarray, n_items: int, device=None):
    """All ranks call this with their own rows; every rank gets the (n_items, len(ROW_FIELDS)) table
    ordered by global index.  Works on any backend: rows travel as one padded int64 tensor."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        table = np.zeros((n_items, len(ROW_FIELDS)), dtype=np.int64)
        table[rows[:, 0]] = rows
        return table
    world = dist.get_world_size()
    per = (n_items + world - 1) // world
    buf = torch.full((per, len(ROW_FIELDS)), -1, dtype=torch.int64)
    buf[: rows.shape[0]] = torch.from_numpy(rows)
    if device is not None:
        buf = buf.to(device)
    parts = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(parts, buf)
    table = np.zeros((n_items, len(ROW_FIELDS)), dtype=np.int64)
    seen = np.zeros(n_items, dtype=bool)
    for p in parts:
        a = p.cpu().numpy()
        a = a[a[:, 0] >= 0]
        if seen[a[:, 0]].any():
            raise RuntimeError("a stream was decoded by two ranks")
        seen[a[:, 0]] = True
        table[a[:, 0]] = a
    if not seen.all():
        raise RuntimeError(f"{int((~seen).sum())} streams were decoded by no rank")
    return table


def range_shards_with_samples(seekpoints, audio_len: int, total_samples: int, parts: int):
    """range_shards plus what a rank needs to decode its range into a region of its own:
    [(offset, length, sample_base, n_samples)], n_samples = inter-channel samples the range holds (from the seek
    points' SampleNumber, flacmeta.go:256-260; the last range ends at total_samples)."""
    first_sample = {int(p.offset): int(p.sample_number) for p in seekpoints
                    if int(p.sample_number) != 0xFFFFFFFFFFFFFFFF}
    first_sample[0] = 0
    shards = range_shards(seekpoints, audio_len, parts)
    bases = [first_sample[o] for o, _ in shards] + [int(total_samples)]
    return [(o, n, bases[i], bases[i + 1] - bases[i]) for i, (o, n) in enumerate(shards)]


def my_range(shards, rank: int, world: int):
    """The range rank `rank` decodes, or None when the seek table gave fewer ranges than ranks."""
    if world < 1 or not 0 <= rank < world:
        raise ValueError(f"rank {rank} of {world}")
    return shards[rank] if rank < len(shards) else None


def chained_md5(local_pcm, chunk_bytes: int = 64 << 20):
    """MD5 of a stream whose PCM lies in rank order across the ranks (each rank passes its range's PCM as a
    1-D torch.uint8 tensor on any device, possibly empty).  MD5 is a chain over the bytes in order, so one agent
    has to see them all: ranks 1.. send theirs to rank 0 in chunks, rank 0 hashes on the host as they arrive.
    Returns the 16-byte digest on rank 0, None elsewhere.  Point-to-point only; works over RCCL and gloo."""
    import hashlib
    import torch
    import torch.distributed as dist
    md = hashlib.md5()
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        md.update(local_pcm.cpu().numpy().tobytes())
        return md.digest()
    rank, world = dist.get_rank(), dist.get_world_size()
    size = torch.tensor([local_pcm.numel()], dtype=torch.int64, device=local_pcm.device)
    sizes = [torch.zeros_like(size) for _ in range(world)]
    dist.all_gather(sizes, size)
    sizes = [int(x.item()) for x in sizes]
    if rank == 0:
        for off in range(0, sizes[0], chunk_bytes):
            md.update(local_pcm[off: off + chunk_bytes].cpu().numpy().tobytes())
        for src in range(1, world):
            for off in range(0, sizes[src], chunk_bytes):
                buf = torch.empty(min(chunk_bytes, sizes[src] - off), dtype=torch.uint8, device=local_pcm.device)
                dist.recv(buf, src=src)
                md.update(buf.cpu().numpy().tobytes())
        return md.digest()
    for off in range(0, sizes[rank], chunk_bytes):
        dist.send(local_pcm[off: off + chunk_bytes].contiguous(), dst=0)
    return None
