"""Multi-GPU sample splitting: one process per GPU, every rank renders a contiguous range of the sample indices
of every pixel, and the per-rank fp32 accumulation buffers are summed onto rank 0 (RCCL reduce over xGMI).

Pixel samples are independent (camera.rs:112-117 sums them per pixel) and the random stream is a pure function of
(seed, pixel, sample index), so the summed buffer equals the single-GPU buffer up to fp32 addition order.
`render_fn(begin, end)` is whatever fills a rank's buffer: the HIP path in production, any stand-in in tests.
"""
from __future__ import annotations


def sample_range(spp: int, rank: int, world: int):
    """Samples [begin, end) of rank `rank` out of `world`: floor(rank*spp/world) .. floor((rank+1)*spp/world)."""
    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    return (rank * spp) // world, ((rank + 1) * spp) // world


def render_split(render_fn, spp: int, rank: int, world: int, dist=None, dst: int = 0, split: str = "spp", timings=None):
    """Render this rank's share with render_fn(begin, end) -> tensor and sum all shares onto `dst`.

    split "spp": the rank's contiguous range of sample indices; "tile": every rank renders all samples (of the tiles it
    owns -- render_fn knows which), so the sum only puts the tiles together.  `timings` (a dict, optional) receives the
    wall seconds this rank spent in render_fn ("render_s") and in the reduce ("reduce_s"): both are host times around
    asynchronous work unless render_fn / the backend synchronise.
    Returns the summed tensor on rank `dst` (other ranks get their own partial back)."""
    import time

    begin, end = sample_range(spp, rank, world) if split == "spp" else (0, spp)
    t0 = time.perf_counter()
    accum = render_fn(begin, end)
    t1 = time.perf_counter()
    if world > 1:
        import torch.distributed as td

        d = dist or td
        d.reduce(accum, dst=dst, op=td.ReduceOp.SUM)
    if timings is not None:
        timings["render_s"] = timings.get("render_s", 0.0) + (t1 - t0)
        timings["reduce_s"] = timings.get("reduce_s", 0.0) + (time.perf_counter() - t1)
    return accum
