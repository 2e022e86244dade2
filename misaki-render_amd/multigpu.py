"""Multi-GPU decomposition of the hot path: one process per GPU, pixel-tile or sample shard, one film reduce.

Samples are independent; the only coupling is the linear film sum (imageblock.cpp:100-110,
157-173).  Rank r of G renders the spiral blocks (imageblock.cpp:187-247) whose id is r mod G —
the interleave balances cheap and expensive tiles — with the full sample count, into its own
H x W x 5 film of weighted sums; the films are then summed onto rank 0 with ONE collective
(RCCL `reduce` over xGMI on the GPU box, gloo in the CPU tests).  A pixel further than the filter
border (2 px) from a tile edge receives a non-zero value from exactly one rank, so the reduced film
equals the single-GPU film bit for bit there; border pixels differ by fp32 re-association of <= 4 terms.

mode="samples" (SURVEY §8e's primary scheme, bench.py's default) shards the other axis: rank r renders EVERY tile for
the sample indices s = r (mod G).  Every rank then runs the single-GPU workload shape (all 256 tiles of a 512^2 film,
spp/G samples each), which matters for the ordered film replay: its parallelism is tiles x pixels and its critical path
is one pixel's spp-long chain, so 1/G of the tiles at the full spp (tile shard) starves it on a small film.  The reduced
film then differs from the single-GPU one by fp32 re-association of G partial sums per pixel (<< the 1e-4 tolerance).
"""


def shard_params(abi, spp_total, rank, world, mode="tiles", shares=None, **kw):
    """Render parameters of rank `rank`.  mode "tiles": blocks id % world == rank, every sample of those blocks;
    mode "samples": every block, sample indices s % world == rank; mode "range": every block, the contiguous sample
    indices [sum(shares[:rank]), sum(shares[:rank + 1])) — shares of unequal size for GPUs of unequal speed."""
    if mode == "tiles":
        return abi.render_params(spp=spp_total, block_first=rank, block_stride=world, **kw)
    if mode == "samples":
        return abi.render_params(spp=spp_total, sample_first=rank, sample_stride=world, **kw)
    if mode == "range":
        if shares is None or len(shares) != world or sum(shares) != spp_total or min(shares) < 1:
            raise ValueError("shares must be %d positive integers that sum to %d" % (world, spp_total))
        begin = sum(shares[:rank])
        return abi.render_params(spp=begin + shares[rank], sample_first=begin, sample_stride=1, **kw)
    raise ValueError(mode)


def balanced_shares(spp_total, step_ms):
    """Split spp_total samples per pixel over the ranks in proportion to their measured speed (1 / step_ms of an equal
    split), largest-remainder rounding, at least one sample each.  Same input on every rank -> same shares on every rank."""
    world = len(step_ms)
    if spp_total < world or min(step_ms) <= 0:
        raise ValueError("cannot balance %d samples over %d ranks" % (spp_total, world))
    speed = [1.0 / t for t in step_ms]
    total = sum(speed)
    exact = [spp_total * v / total for v in speed]
    shares = [max(1, int(x)) for x in exact]
    order = sorted(range(world), key=lambda r: (exact[r] - int(exact[r]), -r), reverse=True)
    k = 0
    while sum(shares) < spp_total:
        shares[order[k % world]] += 1
        k += 1
    while sum(shares) > spp_total:                       # only after the max(1, .) clamp
        r = max(range(world), key=lambda q: shares[q])
        shares[r] -= 1
    return shares


def reduce_film(film, dist=None, dst=0, force=False):
    """Sum the per-rank films onto rank `dst` (in place).  film: torch tensor [H, W, 5] float32.

    Returns when the reduce has finished with `film` on this rank.  The back end renders on its own HIP stream
    (msk_gpu_render_device with hip_stream = NULL returns when the film is complete), the collective runs on the process
    group's stream behind torch's current stream: without the wait below the next render could overwrite `film` while
    the reduce still reads it."""
    # (force: issue the collective in a group of ONE rank too — bench.py --rccl-world1, the RCCL step on a one-GPU box)
    if dist is not None and dist.is_initialized() and (dist.get_world_size() > 1 or force):
        dist.reduce(film, dst=dst, op=dist.ReduceOp.SUM)
        if film.is_cuda:
            import torch
            torch.cuda.current_stream(film.device).synchronize()
    return film


def weak_scaling_spp(spp_per_gpu, world):
    """Weak scaling: per-GPU work constant.  Each rank owns 1/world of the tiles, so the whole job
    renders spp_per_gpu * world samples per pixel."""
    return spp_per_gpu * world


def speed_proportional_shares(dist, kernel_ms, spp_total, device="cpu", threshold=1.04):
    """One all_gather of this rank's kernel time for an equal split -> the same decision on every rank: None (keep the
    equal split: the ranks are within `threshold` of each other, or a time is missing) or the speed-proportional shares
    plus the gathered times."""
    import torch
    world = dist.get_world_size()
    mine = torch.tensor([float(kernel_ms)], dtype=torch.float64, device=device)
    gathered = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(gathered, mine)
    times = [float(x.item()) for x in gathered]
    if min(times) <= 0 or max(times) / min(times) <= threshold or spp_total < world:
        return None, times
    return balanced_shares(spp_total, times), times


class ReducePipeline:
    """Two films in flight per rank: the film reduce of step k (and rank 0's copy-back of the reduced film) overlap the render
    of step k + 1.  Every step still renders, reduces and copies back its own film — the K films of K steps are all on rank 0's
    host when `drain()` returns — but a rank no longer idles while the collective waits for the slowest peer, and rank 0's
    copy-back is off everybody's critical path.  Order of collectives: one reduce per step, in step order, on every rank.

        film = pipe.begin()              # the device film of the coming step (waits until its previous reduce / copy-back is done)
        scene.render_device(prm, film.data_ptr())      # returns when the film is complete (the library's own streams)
        pipe.submit()                    # async reduce onto rank `dst`; there: copy-back into pinned host memory behind it
        ...
        pipe.drain()                     # inside the timed region: everything outstanding has arrived

    CUDA films: the reduce is issued with async_op=True, a side stream waits for it (Work.wait is a stream-level wait), rank
    `dst` queues the device-to-host copy behind it, and an event marks the slot free.  CPU films (gloo, the dry run): the reduce
    is awaited at once — the sequence of calls is the same, nothing overlaps."""

    def __init__(self, films, host_films, dist, rank, dst=0, force=False):
        self.films, self.host_films, self.dist, self.rank, self.dst = films, host_films, dist, rank, dst
        self.active = dist is not None and dist.is_initialized() and (dist.get_world_size() > 1 or force)
        self.k = 0                                   # steps begun
        self.pending = [None] * len(films)           # per slot: a Work (CPU films) or an Event (CUDA films)
        self.side = None
        if films[0].is_cuda:
            import torch
            self.side = torch.cuda.Stream(device=films[0].device)

    def _wait_slot(self, i):
        p, self.pending[i] = self.pending[i], None
        if p is not None:
            p.synchronize() if hasattr(p, "synchronize") else p.wait()

    def begin(self):
        i = self.k % len(self.films)
        self._wait_slot(i)
        return self.films[i]

    def submit(self):
        import torch
        i = self.k % len(self.films)
        self.k += 1
        film = self.films[i]
        work = self.dist.reduce(film, dst=self.dst, op=self.dist.ReduceOp.SUM, async_op=True) if self.active else None
        if not film.is_cuda:
            if work is not None:
                work.wait()
            if self.rank == self.dst:
                self.host_films[i].copy_(film)
            return i
        with torch.cuda.stream(self.side):
            if work is not None:
                work.wait()                                      # the side stream waits for the collective; the host does not
            if self.rank == self.dst:
                self.host_films[i].copy_(film, non_blocking=True)    # pinned target: a DMA behind the reduce
            ev = torch.cuda.Event(blocking=True)
            ev.record(self.side)
        self.pending[i] = ev
        return i

    def drain(self):
        for i in range(len(self.films)):
            self._wait_slot(i)

    def last_host_film(self):
        """rank `dst`, after drain(): the reduced film of the most recent step"""
        return self.host_films[(self.k - 1) % len(self.films)]
