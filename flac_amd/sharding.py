"""File-level sharding of a batch over the GPUs of one node.

FLAC files are independent and every frame is self-contained, so the decode path has no exchange
step: rank r of W takes files r, r+W, r+2W, ... (round robin keeps shards balanced when file sizes
drift along the list), decodes them on its own GPU with its own handle, and only the small per-file
result rows travel afterwards.  One process per GPU, torch.distributed for the rows (RCCL on GPUs,
gloo on CPU); nothing on the data path is a collective.
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
        rows[j, :5] = (i, r.frames, r.frames_ok, r.samples, getattr(r, "first_error", 0))
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
