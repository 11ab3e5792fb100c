"""Multi-GPU decision update: one process per GPU, contigs partitioned across ranks.

The reference is a single process.  Every per-site and per-bin stage of the
update is contig-local (boss/runs/core.py:83-121 loops over contigs), so ranks own whole
contigs — longest-first packing or contiguous runs in FASTA order, whichever balances better
(`partition_contigs`); the row-drift halo of `_distribute_strategy` (core.py:125-155) travels for
every contig, so ownership is free.  What is global:

  * the read-length distribution and the read-start counts (host, 4000 values per batch):
    one all-gather of the per-rank batch summaries when reads are sharded over ranks;
  * "is any strategy switched on" (core.py:111): one MAX all-reduce until it is;
  * normaliser = max(benefit) (sequences.py:588): MAX all-reduce of one double;
  * the exponent histogram, its f-hat sums and ubar0 (sequences.py:593-629): SUM all-reduce of
    ~26 KB of exact integer limbs (order-free, so the result is bit-identical for any rank
    count);
  * the <= n_contigs halo rows per contig and the finished masks (gathered for boss.npz).

All of it is latency-bound; with the "nccl" backend (= RCCL over xGMI) the tensors live on
the GPU, with "gloo" (CPU tests) on the host.  There is no collective on the per-site path.
"""

import numpy as np

from . import _lib
from .runs import BossRuns, MULT, FX_SHIFT, choose_threshold, ubar_to_float


def _partition_linear(w, world):
    """Contiguous partition of `w` (FASTA order) into `world` groups minimising the heaviest group
    (classic linear partition, O(n^2 * world))."""
    n = len(w)
    k = min(world, n)
    pre = np.concatenate(([0.0], np.cumsum(w)))
    cost = np.full((k + 1, n + 1), np.inf)
    cut = np.zeros((k + 1, n + 1), dtype=np.int64)
    cost[0, 0] = 0.0
    for g in range(1, k + 1):
        for i in range(g, n + 1):
            best, arg = np.inf, g - 1
            for j in range(g - 1, i):
                c = max(cost[g - 1, j], pre[i] - pre[j])
                if c < best:
                    best, arg = c, j
            cost[g, i], cut[g, i] = best, arg
    owner = [0] * n
    i = n
    for g in range(k, 0, -1):
        j = int(cut[g, i])
        for t in range(j, i):
            owner[t] = g - 1
        i = j
    return owner


def _refine(w, owner, world):
    """Pairwise moves / swaps off the heaviest rank while they make the pair's heavier side lighter."""
    load = [0.0] * world
    for x, o in zip(w, owner):
        load[o] += x
    for _ in range(10 * len(w) + 10):
        h = max(range(world), key=lambda t: (load[t], -t))
        best = None                                   # (new max of the pair, i, j or None, other rank)
        for i in (t for t in range(len(w)) if owner[t] == h):
            for r in range(world):
                if r == h:
                    continue
                m = max(load[h] - w[i], load[r] + w[i])
                if m < load[h] - 1e-9 and (best is None or m < best[0]):
                    best = (m, i, None, r)
                for j in (t for t in range(len(w)) if owner[t] == r):
                    m = max(load[h] - w[i] + w[j], load[r] + w[i] - w[j])
                    if m < load[h] - 1e-9 and (best is None or m < best[0]):
                        best = (m, i, j, r)
        if best is None:
            break
        _, i, j, r = best
        owner[i] = r; load[h] -= w[i]; load[r] += w[i]
        if j is not None:
            owner[j] = h; load[r] -= w[j]; load[h] += w[j]
    return owner, max(load)


def _partition_lpt(w, world):
    """Longest-first greedy packing: contigs by descending weight, each onto the least loaded rank
    (ties: FASTA order, lowest rank), then pairwise refinement; a few seeded perturbations of the order are packed the
    same way and the lightest heaviest rank wins (deterministic: every rank computes the same owners).  GRCh38 on
    8 ranks: max / mean 1.01 against the linear partition's 1.20."""
    def pack(order):
        load = [0.0] * world
        owner = [0] * len(w)
        for i in order:
            r = min(range(world), key=lambda t: (load[t], t))
            owner[i] = r
            load[r] += w[i]
        return _refine(w, owner, world)
    base = sorted(range(len(w)), key=lambda t: (-w[t], t))
    best_owner, best_max = pack(base)
    if len(w) > world:
        rng = np.random.default_rng(12345)
        lower = max(max(w), sum(w) / world)
        for _ in range(200):
            if best_max <= lower * 1.002:
                break
            order = list(base)
            for _s in range(3):                          # a few adjacent transpositions of the sorted order
                k = int(rng.integers(0, len(order) - 1))
                order[k], order[k + 1] = order[k + 1], order[k]
            o, m = pack(order)
            if m < best_max - 1e-9:
                best_owner, best_max = o, m
    return best_owner


def shard_balance(weights, owner, world):
    """max / mean load of a partition (1.0 = perfect)."""
    load = [0.0] * max(1, int(world))
    for x, o in zip(weights, owner):
        load[o] += float(x)
    mean = sum(load) / len(load)
    return max(load) / mean if mean > 0 else 1.0


def partition_contigs(weights, world, method="auto"):
    """Owner rank per contig.  `linear`: contiguous runs in FASTA order; `lpt`: longest-first packing; `auto`
    (default; BOSSX_PARTITION overrides): whichever leaves the lighter heaviest rank, `linear` on a tie.  Nothing
    downstream needs contiguity — the row-drift halo of `_distribute_strategy` (core.py:125-155) travels for every
    contig through the MAX all-reduce (device) or `_patch_halo` (host), whoever owns its neighbours.  With fewer
    contigs than ranks the trailing ranks own nothing."""
    import os
    w = [float(x) for x in weights]
    world = max(1, int(world))
    if not w:
        return []
    method = os.environ.get("BOSSX_PARTITION", method)
    if method == "linear":
        return _partition_linear(w, world)
    if method == "lpt":
        return _partition_lpt(w, world)
    if method != "auto":
        raise ValueError("partition method must be auto, linear or lpt")
    lin, lpt = _partition_linear(w, world), _partition_lpt(w, world)
    return lpt if shard_balance(w, lpt, world) < shard_balance(w, lin, world) * (1.0 - 1e-9) else lin


class Comm:
    """Thin wrapper over torch.distributed for small numpy payloads."""

    def __init__(self):
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.on = dist.is_available() and dist.is_initialized()
        self.rank = dist.get_rank() if self.on else 0
        self.world = dist.get_world_size() if self.on else 1
        self.device = "cpu"
        if self.on and dist.get_backend() == "nccl":
            self.device = torch.device("cuda", torch.cuda.current_device())
        self.n_collectives = 0
        # exercise the collectives even with a single rank (validation of the nccl path)
        import os
        self.force = self.on and bool(os.environ.get("BOSSX_FORCE_COLLECTIVES"))

    def _t(self, arr):
        return self.torch.from_numpy(np.ascontiguousarray(arr)).to(self.device)

    def allreduce(self, arr, op="sum"):
        arr = np.ascontiguousarray(arr)
        if self.world == 1 and not self.force:
            return arr.copy()
        t = self._t(arr)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM if op == "sum" else self.dist.ReduceOp.MAX)
        self.n_collectives += 1
        return t.cpu().numpy()

    def allgather(self, arr):
        """Same-shape arrays from every rank, stacked on a new leading axis."""
        arr = np.ascontiguousarray(arr)
        if self.world == 1 and not self.force:
            return arr[np.newaxis].copy()
        torch = self.torch
        if self.device == "cpu":
            t = torch.from_numpy(arr)
            out = torch.empty((self.world,) + tuple(t.shape), dtype=t.dtype)
            self.dist.all_gather(list(out.unbind(0)), t)  # views of one tensor (gloo has no all_gather_into_tensor)
            self.n_collectives += 1
            return out.numpy()
        # device path: page-locked staging on both sides, asynchronous copies, ONE synchronisation
        key = (arr.shape, arr.dtype.str)
        buf = getattr(self, "_stage", {}).get(key)
        if buf is None:
            tin = torch.from_numpy(arr.copy()).pin_memory()
            tout = torch.empty((self.world,) + tuple(tin.shape), dtype=tin.dtype).pin_memory()
            din = torch.empty_like(tin, device=self.device)
            dout = torch.empty((self.world,) + tuple(tin.shape), dtype=tin.dtype, device=self.device)
            buf = (tin, tout, din, dout)
            self._stage = getattr(self, "_stage", {})
            self._stage[key] = buf
        tin, tout, din, dout = buf
        tin.numpy()[...] = arr
        din.copy_(tin, non_blocking=True)
        self.dist.all_gather(list(dout.unbind(0)), din)
        tout.copy_(dout, non_blocking=True)
        torch.cuda.current_stream().synchronize()
        self.n_collectives += 1
        return tout.numpy().copy()


def fx_to_limbs(fx):
    """uint64[..., 2] (lo, hi) -> int64[..., 4] 32-bit limbs (sums of <= 2^31 ranks cannot overflow)."""
    fx = np.asarray(fx, dtype=np.uint64)
    lo, hi = fx[..., 0], fx[..., 1]
    m = np.uint64(0xffffffff)
    return np.stack([lo & m, lo >> np.uint64(32), hi & m, hi >> np.uint64(32)], axis=-1).astype(np.int64)


def limbs_to_float(limbs):
    """int64[..., 4] limbs (possibly carrying) -> correctly rounded float64 of value / 2^100."""
    limbs = np.asarray(limbs, dtype=np.int64)
    flat = limbs.reshape(-1, 4)
    out = np.empty(flat.shape[0], dtype=np.float64)
    for i, (a, b, c, d) in enumerate(flat.tolist()):
        out[i] = (a + (b << 32) + (c << 64) + (d << 96)) / (1 << FX_SHIFT)
    return out.reshape(limbs.shape[:-1])


class _DeviceBuffer:
    """Zero-copy view of engine-owned HBM for torch (`__cuda_array_interface__`)."""

    def __init__(self, ptr, shape, typestr):
        self.__cuda_array_interface__ = {"data": (int(ptr), False), "shape": tuple(shape),
                                         "typestr": typestr, "version": 2}


class DistributedBossRuns(BossRuns):
    """`BossRuns` over several GPUs.  Construct on every rank with the same arguments, after
    `torch.distributed.init_process_group`; with world size 1 it degenerates to the stage-wise
    single-GPU update."""

    READ_CAP = 8192      # max reads per rank and batch in the sharded-reads exchange

    def init(self, contigs, engine=None, sharded_reads=True, gather_masks=True, comm=None) -> None:
        """`contigs`: list of (name, sequence) — every rank may pass the full list, or a bare
        length in place of the sequence for contigs it does not own.  `comm` replaces the
        torch.distributed wrapper (tests)."""
        self.comm = comm or Comm()
        self.sharded_reads = sharded_reads
        self.gather_masks = gather_masks
        contigs = list(contigs)
        rej = set(self.args.optional.reject_refs.split(',')) if self.args.optional.reject_refs else set()
        keep = [(n, s) for n, s in contigs
                if (int(s) if isinstance(s, (int, np.integer)) else len(s)) >= int(1e5) and n not in rej]
        nb = len(self.args.general.barcodes) if self.args.general.barcodes else 1
        lens = [int(s) if isinstance(s, (int, np.integer)) else len(s) for _, s in keep]
        self.owner = partition_contigs([l * nb for l in lens], self.comm.world)
        self.owner_of = {n: o for (n, _), o in zip(keep, self.owner)}
        rank = self.comm.rank
        # RCCL path: the engine shares torch's stream and its statistics buffers are wrapped as
        # tensors, so the update's collectives run in-stream (no host round trips)
        import os
        on_gpu = engine is None and self.comm.on and self.comm.device != "cpu" and not os.environ.get("BOSSX_HOST_COLLECTIVES")
        # native driver (default on GPUs): the LIBRARY issues the update's all-reduces (RCCL, on the engine's own
        # stream, between its kernels): one C call per update.  BOSSX_TORCH_COLLECTIVES=1 keeps the form in which
        # torch.distributed issues them in-stream on tensors that alias the engine's buffers.
        self.native = on_gpu and not os.environ.get("BOSSX_TORCH_COLLECTIVES")
        self.instream = on_gpu and not self.native
        if self.native:
            from .engine import Engine
            engine = Engine(nbarcodes=nb, device=self.comm.torch.cuda.current_device(),
                            track_entropy=self.args.gpu.track_entropy)
        if self.instream:
            from .engine import Engine
            torch = self.comm.torch
            # a dedicated (non-default) stream: its handle is non-zero, so the engine really
            # launches on it, and c10d orders every collective issued under
            # `torch.cuda.stream(self.tstream)` against the engine's kernels on that stream
            self.tstream = torch.cuda.Stream()
            assert self.tstream.cuda_stream != 0
            engine = Engine(nbarcodes=nb, device=torch.cuda.current_device(),
                            track_entropy=self.args.gpu.track_entropy,
                            stream=self.tstream.cuda_stream)
        super().init(contigs=contigs, engine=engine, is_local=lambda name, k: self.owner[k] == rank)
        if self.native:
            # rank 0 makes the RCCL id; it travels to the others as an all-reduce of zeros elsewhere
            uid = self.engine.dist_unique_id() if rank == 0 else np.zeros(128, dtype=np.uint8)
            uid = self.comm.allreduce(uid.astype(np.int64), "sum").astype(np.uint8)
            self.engine.dist_init(uid, rank, self.comm.world)
        if self.instream:
            torch = self.comm.torch
            eng = self.engine
            nfilt = len(self.contigs_filt)

            def wrap(which, shape, typestr):
                ptr, nbytes = eng.device_ptr(which)
                t = torch.as_tensor(_DeviceBuffer(ptr, shape, typestr), device=self.comm.device)
                assert t.data_ptr() == ptr and t.numel() * t.element_size() == nbytes
                return t
            self.t_armed = wrap(0, (1,), "<i4")
            self.t_norm = wrap(1, (1,), "<i8")            # bit pattern of a non-negative double
            self.t_limbs = wrap(2, ((_lib.HIST_BINS + 1) * 5,), "<i8")
            # halo rows + the normaliser (ctrl.max_bits lies right behind them): ONE MAX all-reduce
            self.t_tails = wrap(3, (nfilt * nfilt * 2 * nb + 1,), "<f8")
            self.t_tails_only = self.t_tails[:-1]          # long form: SUM over the rows alone
        if hasattr(self.engine, "set_overlap"):
            # the protocol consumes the chain through the stage-wise entry points, which have no
            # time-out fallback for a chain running next to the sweep: keep it after the sweep
            self.engine.set_overlap(False)
        self.local_filt = {n: c for n, c in self.contigs_filt.items() if not c.remote}
        self.armed = False
        self._begun = False
        self.filt_names = list(self.contigs_filt.keys())

    @property
    def n_collectives(self):
        """Collectives of this rank so far: torch.distributed's plus the library's own (native driver)."""
        return self.comm.n_collectives + (self.engine.dist_collectives if getattr(self, "native", False) else 0)

    # ---- batch -----------------------------------------------------------------------------
    def process_batch_paf(self, paf_text, new_reads, barcodes=None, min_len=200, read_lengths=None,
                          **kw) -> None:
        """Sharded reads: this rank's reads (they must map to its own contigs); the read-length
        and read-start summaries are all-gathered so every rank holds the global
        distributions.  Replicated reads (`sharded_reads=False`): every rank passes the whole
        batch and no exchange is needed."""
        summ = self._ingest_batch(paf_text, new_reads, barcodes, min_len)
        if read_lengths is None:
            read_lengths = np.array([len(s) for s in new_reads.values()], dtype=np.int64)
        self.begin_update()
        self.account_batch(summ, read_lengths, len(new_reads))      # (launches the chain once the global read lengths are known)
        la = kw.get("lookahead")
        if la is not None and getattr(self, "native", False):
            # the whole update, collectives included, is in the queue before the next batch is staged (runs.py)
            self._update_native(between=lambda: self._stage_ahead(la))
        else:
            self._stage_ahead(la)       # the next batch, staged while this one's chain runs
            self.update_wrapper()

    def account_batch(self, summ, read_lengths, n_reads):
        """Make the read-length distribution, read-start counts and abundance counts global:
        with sharded reads, one all-gather of this rank's per-mapping summary."""
        read_lengths = np.asarray(list(read_lengths.values()) if isinstance(read_lengths, dict)
                                  else read_lengths, dtype=np.int64)
        if self.sharded_reads and (self.comm.world > 1 or self.comm.force):
            cap = self.READ_CAP
            k, m = len(summ["contig_idx"]), len(read_lengths)
            if k > cap or m > cap:
                raise ValueError("batch larger than READ_CAP")
            # one int32 record: header (k, m, n_reads) + contig / strand / start / end columns of
            # the k chosen mappings + the m read lengths (positions and lengths are < 2^31)
            buf = getattr(self, "_xbuf", None)
            if buf is None:
                buf = self._xbuf = np.zeros(4 + 5 * cap, dtype=np.int32)
            buf[:3] = (k, m, n_reads)
            buf[4:4 + k] = summ["contig_idx"]
            buf[4 + cap:4 + cap + k] = summ["rev"]
            buf[4 + 2 * cap:4 + 2 * cap + k] = summ["tstart"]
            buf[4 + 3 * cap:4 + 3 * cap + k] = summ["tend"]
            buf[4 + 4 * cap:4 + 4 * cap + m] = np.minimum(read_lengths, 2 ** 31 - 1)
            # (native driver: the library's own communicator and stream; otherwise torch.distributed)
            allb = self.engine.dist_allgather(buf) if getattr(self, "native", False) else self.comm.allgather(buf)
            # the read lengths first: the move_sum windows (and with them the chain) wait for them
            read_lengths = np.concatenate([b[4 + 4 * cap:4 + 4 * cap + int(b[1])] for b in allb]).astype(np.int64)
            self.rl_dist.update(read_lengths)
            self._launch_chain_early()
            ks = [int(b[0]) for b in allb]
            col = lambda j: np.concatenate([b[4 + j * cap:4 + j * cap + kk] for b, kk in zip(allb, ks)]).astype(np.int64)
            n_reads = int(sum(int(b[2]) for b in allb))
            summ = dict(contig_idx=col(0), rev=col(1), tstart=col(2), tend=col(3))
        else:
            self.rl_dist.update(read_lengths)
            self._launch_chain_early()          # the chain needs only the read-length windows
        self.total_reads += n_reads
        ci = np.asarray(summ["contig_idx"])
        for i, n in enumerate(np.bincount(ci[ci >= 0], minlength=len(self.contig_names))):
            if n:
                self.read_counts[self.contig_names[i]] += int(n)
        self.read_starts.count_starts(self.contig_names, ci, summ["rev"], summ["tstart"], summ["tend"])

    # ---- update ----------------------------------------------------------------------------
    def begin_update(self):
        """Start the sweep (with the pending batch's increments) and the device-side bucket
        switches now, so that `account_batch`'s exchange and the host bookkeeping overlap with
        them.  Optional: `update_wrapper` does it if it was not called."""
        if hasattr(self.engine, "update_begin") and not self._begun:
            self.engine.update_begin(self.args.optional.bucket_threshold)
            self._begun = True

    def _launch_chain_early(self):
        """In-stream protocol, sweep already enqueued: as soon as the global read-length windows
        are known, exchange the "some strategy is on" flag (until it is) and enqueue the chain, so
        that it runs while the host still counts read starts and builds f-hat."""
        self._chain_early = False
        if getattr(self, "native", False) and self._begun and hasattr(self.rl_dist, "time_cost") and getattr(self, "early_chain", True):
            windows = np.concatenate(([400 // 100], self.rl_dist.approx_ccl // 100)).astype(np.int32)
            self.engine.dist_chain(windows, MULT)
            self._chain_early = True
            return
        if not (getattr(self, "instream", False) and self._begun and hasattr(self.rl_dist, "time_cost")
                and getattr(self, "early_chain", True) and self.comm.dist is not None):
            return
        dist, torch = self.comm.dist, self.comm.torch
        with torch.cuda.stream(self.tstream):
            if not self.armed:
                dist.all_reduce(self.t_armed, op=dist.ReduceOp.MAX)      # core.py:111 is a global decision
                self.comm.n_collectives += 1
            windows = np.concatenate(([400 // 100], self.rl_dist.approx_ccl // 100)).astype(np.int32)
            self.engine.update_benefit(windows, MULT)                    # gated on the (now global) flag
        self._chain_early = True

    def _update_instream(self) -> None:
        """The update with device-resident statistics: two in-stream RCCL all-reduces (three until
        some strategy is on) between
        asynchronous engine stages, one synchronisation at the end (bossx.h, bossx_device_ptr)."""
        eng, dist, torch = self.engine, self.comm.dist, self.comm.torch
        MAXOP, SUMOP = dist.ReduceOp.MAX, dist.ReduceOp.SUM
        self.begin_update()
        self._begun = False
        have_rl = hasattr(self.rl_dist, "time_cost")
        early = getattr(self, "_chain_early", False)
        self._chain_early = False
        with torch.cuda.stream(self.tstream):                    # collectives ordered on the engine's stream
            if not early and not self.armed:                     # sticky: once on everywhere, no exchange
                dist.all_reduce(self.t_armed, op=MAXOP)          # core.py:111 is a global decision
                self.comm.n_collectives += 1
            if have_rl:
                if not early:
                    windows = np.concatenate(([400 // 100], self.rl_dist.approx_ccl // 100)).astype(np.int32)
                    eng.update_benefit(windows, MULT)            # gated on the (now global) flag
                eng.dist_tails()                                 # halo rows + normaliser in one buffer
                dist.all_reduce(self.t_tails, op=MAXOP)          # non-negative, one contributor each: exact
                if self.read_starts._engine is not None:
                    # every rank holds the global read-start counts in HBM: the posterior is rebuilt there
                    fm = self.read_starts.fhat_model()
                    eng.fhat_build(fm)
                    eng.dist_hist(None, fm["target_rs"], self.ref.n_sites // 100, n_windows=fm["n_windows"])
                else:
                    fhat_c, target_rs = self.read_starts.fhat_compact()
                    eng.dist_hist(fhat_c, target_rs, self.ref.n_sites // 100)
                dist.all_reduce(self.t_limbs, op=SUMOP)
                eng.dist_pick(self.rl_dist.time_cost // 100)
                self.comm.n_collectives += 2
        res = eng.dist_finish()
        for cont in self.local_filt.values():
            if res["contig_on"][cont.index]:
                cont.switched_on[:] = True
        self.armed = res["any_on"]
        if not self.armed:
            return
        if not have_rl:
            raise AttributeError("'ReadlengthDist' object has no attribute 'time_cost'")
        self.threshold = res["threshold"]
        self.last_stats = dict(normaliser=res["normaliser"], ubar0=res["ubar0"],
                               strat_size=res["strat_size"], n_bins=res["n_bins"], argmax_margin=res.get("argmax_margin"))
        for cont in self.local_filt.values():
            cont.strat = eng.strat_view(cont.index)
        if self.gather_masks and (self.comm.world > 1 or self.comm.force):
            self._gather_masks()
        if self.write_masks and self.comm.rank == 0:
            self._write_contig_strategies(contig_strats=self.ref.get_strategy_dict())

    def _update_native(self, between=None) -> None:
        """The update as ONE library call (bossx_dist_update): RCCL all-reduces issued by the library between its
        own kernels — MAX of the armed flag (until some strategy is on), MAX over halo rows + normaliser, SUM of
        the exact histogram limbs — one synchronisation at the end."""
        eng = self.engine
        self.n_updates += 1
        self.begin_update()
        self._begun = False
        self._chain_early = False
        thr_b = self.args.optional.bucket_threshold
        if hasattr(self.rl_dist, "time_cost"):
            windows = np.concatenate(([400 // 100], self.rl_dist.approx_ccl // 100)).astype(np.int32)
            if self.read_starts._engine is not None:
                res = eng.update(thr_b, windows, MULT, tc=self.rl_dist.time_cost // 100,
                                 fhat_model=self.read_starts.fhat_model(), dist=True, between=between)
            else:
                fhat_c, target_rs = self.read_starts.fhat_compact()
                res = eng.update(thr_b, windows, MULT, tc=self.rl_dist.time_cost // 100, fhat_c=fhat_c, target_rs=target_rs, dist=True,
                                 between=between)
        else:
            res = eng.update(thr_b, dist=True, between=between)
        # An exception raised by `between` (staging the next batch while this update ran) was held back by Engine.update so that this
        # update's results could be collected: it is re-raised once they have been applied (KeyboardInterrupt / SystemExit
        # included — they travelled through the same hand-off).  If APPLYING the results fails, that failure is the primary one
        # (ADVICE r5): it is raised with the held error as its cause instead of being replaced by it.
        held = res.get("between_error")

        def apply():
            for cont in self.local_filt.values():
                if res["contig_on"][cont.index]:
                    cont.switched_on[:] = True
            self.armed = res["any_on"]
            if not self.armed:
                return
            if not hasattr(self.rl_dist, "time_cost"):
                raise AttributeError("'ReadlengthDist' object has no attribute 'time_cost'")
            self.threshold = res["threshold"]
            self.last_stats = dict(normaliser=res["normaliser"], ubar0=res["ubar0"],
                                   strat_size=res["strat_size"], n_bins=res["n_bins"], argmax_margin=res.get("argmax_margin"))
            for cont in self.local_filt.values():
                cont.strat = eng.strat_view(cont.index)
            if self.gather_masks and (self.comm.world > 1 or self.comm.force):
                self._gather_masks()
            if self.write_masks and self.comm.rank == 0:
                self._write_contig_strategies(contig_strats=self.ref.get_strategy_dict())
        try:
            apply()
        except BaseException as primary:
            if held is not None and primary is not held:
                raise primary from held
            raise
        if held is not None:
            raise held

    n_updates = 0            # updates run since init (bench: collectives per update), whichever entry point ran them

    def update_wrapper(self) -> None:
        return self._update_wrapper()

    def _update_wrapper(self) -> None:
        if getattr(self, "native", False):
            return self._update_native()
        self.n_updates += 1
        if getattr(self, "instream", False):
            return self._update_instream()
        eng, comm = self.engine, self.comm
        device_side = hasattr(eng, "update_begin")        # the HIP engine; test doubles lack it
        thr_b = self.args.optional.bucket_threshold
        if device_side:
            self.begin_update()
            self._begun = False
            windows = None
            if self.armed and hasattr(self.rl_dist, "time_cost"):
                windows = np.concatenate(([400 // 100], self.rl_dist.approx_ccl // 100)).astype(np.int32)
                eng.update_benefit(windows, MULT)          # chain runs while f-hat is built
            if not self.armed:
                res = eng.update(thr_b)                    # waits for sweep + buckets: local flags
                for cont in self.local_filt.values():
                    if res["contig_on"][cont.index]:
                        cont.switched_on[:] = True
                self.armed = bool(comm.allreduce(np.array([int(res["any_on"])], dtype=np.int64), "max")[0])
                if not self.armed:
                    return
                eng.arm()                                  # the decision is global (core.py:111)
            fhat_c, target_rs = self.read_starts.fhat_compact()
            if windows is None:
                windows = np.concatenate(([400 // 100], self.rl_dist.approx_ccl // 100)).astype(np.int32)
                eng.update_benefit(windows, MULT)
            local_max = eng.get_max()
        else:
            eng.sweep()
            for cont in self.local_filt.values():
                cont.check_buckets(eng.bucket_sums(cont.index), threshold=thr_b)
            if not self.armed:
                local_on = any(any(c.switched_on) for c in self.local_filt.values())
                self.armed = bool(comm.allreduce(np.array([int(local_on)], dtype=np.int64), "max")[0])
                if not self.armed:
                    return
            fhat_c, target_rs = self.read_starts.fhat_compact()
            windows = np.concatenate(([400 // 100], self.rl_dist.approx_ccl // 100)).astype(np.int32)
            local_max = eng.benefit(windows, MULT)
        normaliser = float(comm.allreduce(np.array([local_max], dtype=np.float64), "max")[0])
        target = self.ref.n_sites // 100
        counts, fg, ub = eng.histogram(normaliser, fhat_c, target_rs, target)
        packed = np.concatenate((counts.reshape(-1, 1), fx_to_limbs(fg)), axis=1)           # [bins, 5]
        packed = np.concatenate((packed, np.concatenate(([0], fx_to_limbs(ub)))[np.newaxis]))
        packed = comm.allreduce(packed, "sum")
        counts = packed[:-1, 0]
        fgrid = np.zeros(counts.shape[0], dtype=np.float64)
        occ = np.nonzero(counts)[0]                      # only occupied bins need the big-int conversion
        fgrid[occ] = limbs_to_float(packed[occ, 1:])
        ubar0 = ubar_to_float(float(limbs_to_float(packed[-1, 1:])), normaliser)
        threshold, size, uniq = choose_threshold(normaliser, counts, fgrid, ubar0, self.rl_dist.time_cost)
        self.threshold = threshold
        self.last_stats = dict(normaliser=normaliser, exponents=uniq, counts=counts[uniq],
                               f_grid=fgrid[uniq], ubar0=ubar0, strat_size=size)
        eng.apply_threshold(threshold)
        for cont in self.local_filt.values():
            cont.strat = eng.get_strat(cont.index)
        if comm.world > 1 or comm.force:
            self._patch_halo(threshold)
            if self.gather_masks:
                self._gather_masks()
        if self.write_masks and comm.rank == 0:
            self._write_contig_strategies(contig_strats=self.ref.get_strategy_dict())

    def _patch_halo(self, threshold):
        """Contig k takes merged rows [row_off_k, row_off_k + T_k) although its block starts k
        rows later (core.py:141-155): its first <= k rows come from the tails of the preceding
        blocks, which may live on another rank.  Owners publish `benefit >= threshold` for the
        last n_filt rows of their blocks; everyone patches."""
        nf = len(self.filt_names)
        nb = self.nbarcodes
        tails = np.zeros((nf, nf, 2, nb), dtype=np.int64)
        for j, name in enumerate(self.filt_names):
            c = self.contigs_filt[name]
            if c.remote:
                continue
            t = self.engine.export(c.index, "benefit_tail") >= threshold          # [K,2,nb]
            tails[j, nf - t.shape[0]:] = t
        tails = self.comm.allreduce(tails, "sum")
        T = [self.contigs_filt[n].length // 100 for n in self.filt_names]
        row_off = np.concatenate(([0], np.cumsum(T)))
        bin_off = row_off[:-1] + np.arange(nf)
        bin_end = bin_off + np.array(T) + 1
        for k, name in enumerate(self.filt_names):
            c = self.contigs_filt[name]
            if c.remote:
                continue
            sw = c.bucket_switches
            for r in range(min(k, T[k])):
                g = row_off[k] + r
                j = int(np.searchsorted(bin_off, g, side="right") - 1)
                if not self.contigs_filt[self.filt_names[j]].remote:
                    continue                                   # the mask kernel already wrote it
                row = tails[j, nf - (bin_end[j] - g)]
                on = sw[r // 200]
                c.strat[r][:, on] = row[:, on].astype(bool)

    def _gather_masks(self):
        """Every rank ends up with every contig's mask (rank 0 writes boss.npz)."""
        sizes = [self.contigs_filt[n].length // 100 * 2 * self.nbarcodes for n in self.filt_names]
        total = int(sum(sizes))
        buf = np.zeros(total, dtype=np.uint8)
        off = 0
        for n, sz in zip(self.filt_names, sizes):
            c = self.contigs_filt[n]
            if not c.remote:
                buf[off:off + sz] = c.strat.reshape(-1)
            off += sz
        allb = self.comm.allgather(buf)
        off = 0
        for n, sz in zip(self.filt_names, sizes):
            c = self.contigs_filt[n]
            if c.remote:
                c.strat = allb[self.owner_of[n], off:off + sz].view(np.bool_).reshape(-1, 2, self.nbarcodes).copy()
            off += sz

    update_strategy = update_wrapper
