"""numpy-friendly wrapper around one bossx engine handle (one GPU)."""
import ctypes as C

import numpy as np

from . import _lib

EXPORT = dict(coverage=0, scores=1, entropy=2, scores_ds=3, benefit=4, state=5, touched=6,
              bucket_switches=7, benefit_tail=8)


class _PinnedBlock:
    """Owner of one hipHostMalloc block, exposed to numpy through the array interface: arrays and
    views made from it keep it alive, and the block frees itself with the last of them."""

    def __init__(self, lib, ptr, nbytes):
        self._lib, self._ptr = lib, ptr
        self.__array_interface__ = {"data": (ptr, False), "shape": (nbytes,), "typestr": "|u1", "version": 3}

    def __del__(self):
        try:
            self._lib.bossx_host_free(self._ptr)
        except Exception:
            pass


class _LazyIds:
    """The read names of a staged batch in batch order, listed only if somebody asks (the simulation
    path does; the per-update path does not)."""

    def __init__(self, seqs):
        self._seqs, self._ids = seqs, None

    def _list(self):
        if self._ids is None:
            self._ids = list(self._seqs.keys())
        return self._ids

    def __getitem__(self, i):
        return self._list()[i]

    def __len__(self):
        return len(self._seqs)

    def __iter__(self):
        return iter(self._list())


class Engine:
    def __init__(self, nbarcodes=1, device=0, track_entropy=True, stream=None):
        self.lib = _lib.load()
        self.nb = int(nbarcodes)
        cfg = _lib.Config(device=int(device), nbarcodes=self.nb, track_entropy=int(bool(track_entropy)),
                          reserved=0, stream=stream)
        h = C.c_void_p()
        rc = self.lib.bossx_create(C.byref(cfg), C.byref(h))
        if rc != 0:
            raise _lib.BossxError(
                "bossx_create failed (%d): no usable HIP device. The decision-update path has no "
                "CPU fallback." % rc)
        self.h = h
        self.names = []
        self.lengths = []
        self.rejected = []
        self.remote = []

    def _host_buffer(self, nbytes, fill):
        """uint8 array over page-locked host memory (direct DMA target of the mask copies).  The
        memory belongs to the array: it is released when the last view of it is gone, which may be
        after the engine."""
        try:
            ptr = C.c_void_p()
            self._ck(self.lib.bossx_host_alloc(self.h, int(nbytes), C.byref(ptr)))
            arr = np.asarray(_PinnedBlock(self.lib, ptr.value, int(nbytes)))
        except Exception:          # pinning is an optimisation only
            arr = np.empty(int(nbytes), dtype=np.uint8)
        arr[:] = fill
        return arr

    def close(self):
        if getattr(self, "h", None):
            self.lib.bossx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ck(self, rc):
        _lib.check(self.lib, self.h, rc)

    # ---- set-up --------------------------------------------------------------------------
    def add_contig(self, name, seq, rejected=False, remote_length=None):
        """seq: str/bytes (ASCII); None for a rejected contig; `remote_length` registers a
        contig whose sites live on another GPU (geometry only)."""
        if rejected:
            self._ck(self.lib.bossx_add_contig(self.h, name.encode(), None, 0, 1))
            self.lengths.append(4)
        elif remote_length is not None:
            self._ck(self.lib.bossx_add_contig(self.h, name.encode(), None, int(remote_length), 2))
            self.lengths.append(int(remote_length))
        else:
            raw = seq.encode("ascii") if isinstance(seq, str) else bytes(seq)
            self._ck(self.lib.bossx_add_contig(self.h, name.encode(), raw, len(raw), 0))
            self.lengths.append(len(raw))
        self.names.append(name.strip().split(" ")[0])
        self.rejected.append(bool(rejected))
        self.remote.append(remote_length is not None and not rejected)
        return len(self.names) - 1

    def finalize(self, score0, ent0):
        self._ck(self.lib.bossx_finalize(self.h, float(score0), float(ent0)))

    def set_lut(self, score, entropy):
        score = np.ascontiguousarray(score, dtype=np.float64).reshape(-1)
        entropy = np.ascontiguousarray(entropy, dtype=np.float64).reshape(-1)
        self._ck(self.lib.bossx_set_lut(self.h, score.ctypes.data, entropy.ctypes.data, score.size))

    # ---- ingestion -----------------------------------------------------------------------
    @staticmethod
    def pack_reads(seqs):
        """{read id: sequence} -> (names blob, name offsets, seq blob, seq offsets, ids)."""
        ids = list(seqs.keys())
        n = len(ids)
        name_off = np.zeros(n + 1, dtype=np.int64)
        seq_off = np.zeros(n + 1, dtype=np.int64)
        if n:
            np.cumsum([len(i.encode()) for i in ids], out=name_off[1:])
            np.cumsum([len(s) for s in seqs.values()], out=seq_off[1:])
        names = "".join(ids).encode()
        blob = "".join(seqs.values()).encode("ascii")
        return names, name_off, blob, seq_off, ids

    _as_utf8 = None

    @classmethod
    def _str_pointers(cls, strings):
        """Pointers/lengths of the (ASCII) buffers inside Python str objects — no copies; the
        caller keeps the strings alive for the duration of the C call."""
        if cls._as_utf8 is None:
            cls._as_utf8 = (C.cast(C.pythonapi.PyList_GetItem, C.c_void_p).value,
                            C.cast(C.pythonapi.PyUnicode_AsUTF8AndSize, C.c_void_p).value)
        if not isinstance(strings, list):
            strings = list(strings)
        n = len(strings)
        ptrs = np.empty(max(n, 1), dtype=np.uint64)
        lens = np.empty(max(n, 1), dtype=np.int64)
        if n:
            rc = _lib.load_gil().bossx_py_str_pointers(strings, n, cls._as_utf8[0], cls._as_utf8[1],
                                                       ptrs.ctypes.data, lens.ctypes.data)
            if rc:
                raise TypeError("read names and sequences must be str")
        return ptrs, lens

    def stage_batch(self, paf_text, seqs, barcodes=None, min_len=200, ingest=False, packed=None):
        """Parse + upload one batch.  Returns dict of per-mapping summary arrays."""
        # (bossx_stage_stream: a batch consumed right away is staged on the main stream, one staged ahead on the staging stream)
        if self.__dict__.get("_stage_main") is not bool(ingest):
            self._stage_main = bool(ingest)
            self._ck(self.lib.bossx_stage_stream(self.h, 1 if ingest else 0))
        if packed is None and not ingest:
            return self._stage_batch_ptrs(paf_text, seqs, barcodes, min_len)
        if packed is None and ingest:
            out = self._stage_batch_ptrs(paf_text, seqs, barcodes, min_len)
            self.ingest_staged()
            return out
        names, name_off, blob, seq_off, ids = packed
        n = len(ids)
        paf = paf_text.encode() if isinstance(paf_text, str) else bytes(paf_text)
        bc = None
        if barcodes is not None:
            bc = np.ascontiguousarray([barcodes[i] for i in ids] if isinstance(barcodes, dict) else barcodes,
                                      dtype=np.int32)
        s = dict(read_idx=np.zeros(max(n, 1), np.int32), contig_idx=np.zeros(max(n, 1), np.int32),
                 rev=np.zeros(max(n, 1), np.uint8), tstart=np.zeros(max(n, 1), np.int64),
                 tend=np.zeros(max(n, 1), np.int64), qlen=np.zeros(max(n, 1), np.int64))
        summ = _lib.BatchSummary(*[s[k].ctypes.data for k in ("read_idx", "contig_idx", "rev", "tstart", "tend", "qlen")])
        n_rec = C.c_int32(0)
        aligned = C.c_int64(0)
        fn = self.lib.bossx_ingest_paf if ingest else self.lib.bossx_stage_batch
        self._ck(fn(self.h, paf, len(paf), names, name_off.ctypes.data, blob, seq_off.ctypes.data,
                    None if bc is None else bc.ctypes.data, n, int(min_len), C.byref(summ),
                    C.byref(n_rec), C.byref(aligned)))
        k = n_rec.value
        out = {key: v[:k] for key, v in s.items()}
        out["aligned"] = aligned.value
        out["ids"] = ids
        return out

    _dict_next = None

    def _dict_pointers(self, seqs):
        """Pointers / lengths of the names and sequences of a {read id: sequence} dict, one pass in C
        (PyDict_Next; the dict keeps every buffer alive for the call).  The arrays are the engine's
        own and are reused by the next call."""
        if Engine._dict_next is None:
            Engine._dict_next = (C.cast(C.pythonapi.PyDict_Next, C.c_void_p).value,
                                 C.cast(C.pythonapi.PyUnicode_AsUTF8AndSize, C.c_void_p).value)
        n = len(seqs)
        buf = self.__dict__.get("_ptr_buf")
        if buf is None or buf[0].size < n:
            cap = max(n, 1) * 5 // 4 + 16
            buf = self._ptr_buf = (np.empty(cap, np.uint64), np.empty(cap, np.int64), np.empty(cap, np.uint64), np.empty(cap, np.int64))
        got = _lib.load_gil().bossx_py_dict_pointers(seqs, n, Engine._dict_next[0], Engine._dict_next[1],
                                                     buf[0].ctypes.data, buf[1].ctypes.data, buf[2].ctypes.data, buf[3].ctypes.data)
        if got != n:
            raise TypeError("read names and sequences must be str")
        return buf

    def _summary_buffers(self, n):
        """Per-mapping summary arrays of a staging call (reused between calls: the caller gets views)."""
        s = self.__dict__.get("_summ_buf")
        if s is None or s["read_idx"].size < max(n, 1):
            cap = max(n, 1) * 5 // 4 + 16
            s = self._summ_buf = dict(read_idx=np.zeros(cap, np.int32), contig_idx=np.zeros(cap, np.int32),
                                      rev=np.zeros(cap, np.uint8), tstart=np.zeros(cap, np.int64),
                                      tend=np.zeros(cap, np.int64), qlen=np.zeros(cap, np.int64))
            self._summ_struct = _lib.BatchSummary(*[s[k].ctypes.data for k in ("read_idx", "contig_idx", "rev", "tstart", "tend", "qlen")])
        return s, self._summ_struct

    def _stage_batch_ptrs(self, paf_text, seqs, barcodes, min_len):
        if type(seqs) is dict:
            n = len(seqs)
            nptr, nlen, sptr, slen = self._dict_pointers(seqs)
            ids = None
        else:
            ids = list(seqs.keys())
            vals = list(seqs.values())
            n = len(ids)
            nptr, nlen = self._str_pointers(ids)
            sptr, slen = self._str_pointers(vals)
        if isinstance(paf_text, str):
            # the interpreter's own UTF-8 buffer of the str (for ASCII text: the string itself): no copy
            pp, pl = self._str_pointers([paf_text])
            paf, paf_len = int(pp[0]), int(pl[0])
        else:
            paf = bytes(paf_text)
            paf_len = len(paf)
        bc = None
        if barcodes is not None:
            if isinstance(barcodes, dict):
                bc = np.fromiter((barcodes[i] for i in (seqs.keys() if ids is None else ids)), dtype=np.int32, count=n)
            else:
                bc = np.ascontiguousarray(barcodes, dtype=np.int32)
        s, summ = self._summary_buffers(n)
        n_rec = C.c_int32(0)
        aligned = C.c_int64(0)
        self._ck(self.lib.bossx_stage_batch_ptrs(self.h, paf, paf_len, nptr.ctypes.data, nlen.ctypes.data,
                                                 sptr.ctypes.data, slen.ctypes.data,
                                                 None if bc is None else bc.ctypes.data, n, int(min_len),
                                                 C.byref(summ), C.byref(n_rec), C.byref(aligned)))
        k = n_rec.value
        out = {key: v[:k].copy() for key, v in s.items()}      # (the engine's buffers are reused by the next staging call)
        out["aligned"] = aligned.value
        out["ids"] = _LazyIds(seqs) if ids is None else ids
        return out

    def paf_summary(self, paf_text, read_ids, min_len=1):
        """Mapping choice only (filters + best mapper): per-mapping summary arrays, nothing is
        staged.  `read_ids`: iterable of read names present in the batch."""
        ids = list(read_ids)
        n = len(ids)
        nptr, nlen = self._str_pointers(ids)
        paf = paf_text.encode() if isinstance(paf_text, str) else bytes(paf_text)
        s = dict(read_idx=np.zeros(max(n, 1), np.int32), contig_idx=np.zeros(max(n, 1), np.int32),
                 rev=np.zeros(max(n, 1), np.uint8), tstart=np.zeros(max(n, 1), np.int64),
                 tend=np.zeros(max(n, 1), np.int64), qlen=np.zeros(max(n, 1), np.int64))
        summ = _lib.BatchSummary(*[s[k].ctypes.data for k in ("read_idx", "contig_idx", "rev", "tstart", "tend", "qlen")])
        n_rec = C.c_int32(0)
        self._ck(self.lib.bossx_paf_summary(self.h, paf, len(paf), nptr.ctypes.data, nlen.ctypes.data, n,
                                            int(min_len), C.byref(summ), C.byref(n_rec)))
        out = {key: v[:n_rec.value] for key, v in s.items()}
        out["ids"] = ids
        return out

    def ingest_paf(self, paf_text, seqs, barcodes=None, min_len=200):
        return self.stage_batch(paf_text, seqs, barcodes, min_len, ingest=True)

    def ingest_staged(self, slot=None):
        if slot is not None:
            self.select_batch(slot)
        self._ck(self.lib.bossx_ingest_staged(self.h))

    def select_batch(self, slot):
        self._ck(self.lib.bossx_select_batch(self.h, int(slot)))

    # ---- update stages -------------------------------------------------------------------
    def sweep(self):
        self._ck(self.lib.bossx_sweep(self.h))

    def bucket_sums(self, contig):
        nfull = self.lengths[contig] // 20000
        dst = np.zeros((self.nb, nfull), dtype=np.uint64)
        self._ck(self.lib.bossx_get_bucket_sums(self.h, contig, dst.ctypes.data))
        return dst

    def set_bucket_switches(self, contig, switches):
        sw = np.ascontiguousarray(switches, dtype=np.uint8)
        assert sw.shape == (self.lengths[contig] // 20000 + 1, self.nb)
        self._ck(self.lib.bossx_set_bucket_switches(self.h, contig, sw.ctypes.data))

    def benefit(self, windows, mult):
        w = np.ascontiguousarray(windows, dtype=np.int32)
        m = np.ascontiguousarray(mult, dtype=np.float64)
        assert w.shape == (_lib.NWIN,) and m.shape == (10,)
        mx = C.c_double(0.0)
        self._ck(self.lib.bossx_benefit(self.h, w.ctypes.data, m.ctypes.data, C.byref(mx)))
        return mx.value

    def histogram(self, normaliser, fhat_c, target_rs, target):
        f = np.ascontiguousarray(fhat_c, dtype=np.float64)
        desc = _lib.FhatDesc(f.ctypes.data, f.shape[0], 20, int(target_rs), int(target))
        counts = np.zeros(_lib.HIST_BINS, dtype=np.int64)
        fg = np.zeros((_lib.HIST_BINS, 2), dtype=np.uint64)
        ub = np.zeros(2, dtype=np.uint64)
        self._ck(self.lib.bossx_histogram(self.h, float(normaliser), C.byref(desc), counts.ctypes.data,
                                          fg.ctypes.data, ub.ctypes.data))
        return counts, fg, ub

    def update_begin(self, bucket_threshold):
        """Enqueue sweep + bucket switches now (asynchronous); the following `update` call only
        adds the strategy stages.  Lets host bookkeeping overlap with the sweep."""
        self._ck(self.lib.bossx_update_begin(self.h, float(bucket_threshold)))
        self._sweep_done = True

    def update_benefit(self, windows, mult):
        """Enqueue the move_sum chain now (asynchronous, gated on the device-side armed flag)."""
        w = np.ascontiguousarray(windows, dtype=np.int32)
        m = np.ascontiguousarray(mult, dtype=np.float64)
        assert w.shape == (_lib.NWIN,) and m.shape == (10,)
        self._ck(self.lib.bossx_update_benefit(self.h, w.ctypes.data, m.ctypes.data))
        self._benefit_done = tuple(w.tolist())

    def device_ptr(self, which):
        """(address, nbytes) of a device-resident statistics buffer (0 armed flag, 1 normaliser,
        2 limbs, 3 tails) for zero-copy wrapping by torch (multi-GPU in-stream collectives)."""
        ptr = C.c_void_p()
        nbytes = C.c_size_t()
        self._ck(self.lib.bossx_device_ptr(self.h, int(which), C.byref(ptr), C.byref(nbytes)))
        return ptr.value, nbytes.value

    def fhat_build(self, model):
        """bossx_fhat_build: the posterior from the resident counts, in-stream (`model`: ReadStartDist.fhat_model())."""
        self._ck(self.lib.bossx_fhat_build(self.h, int(model["n_windows"]), int(model["target_rs"]), float(model["alpha"]),
                                           float(model["den"]), float(model["expected"]), float(model["on_target"])))

    def dist_hist(self, fhat_c, target_rs, target, n_windows=None):
        """`fhat_c` None: the posterior built by fhat_build (n_windows required)."""
        if fhat_c is None:
            desc = _lib.FhatDesc(None, int(n_windows), 20, int(target_rs), int(target))
        else:
            f = np.ascontiguousarray(fhat_c, dtype=np.float64)
            desc = _lib.FhatDesc(f.ctypes.data, f.shape[0], 20, int(target_rs), int(target))
            self._keep = f                       # the upload is asynchronous
        self._ck(self.lib.bossx_dist_hist(self.h, C.byref(desc)))

    def dist_tails(self):
        """Publish the block tails and put the normaliser into the tails buffer's last slot: one
        MAX all-reduce over that buffer then replaces the normaliser and tails exchanges."""
        self._ck(self.lib.bossx_dist_tails(self.h))

    def dist_pick(self, tc):
        self._ck(self.lib.bossx_dist_pick(self.h, float(tc)))

    def dist_finish(self):
        if getattr(self, "strat_all", None) is None:
            self.strat_all = self._host_buffer(max(int(self.lib.bossx_strat_bytes(self.h)), 1), 1)
        on = np.zeros(len(self.names), dtype=np.uint8)
        res = _lib.UpdateResult()
        self._ck(self.lib.bossx_dist_finish(self.h, self.strat_all.ctypes.data, on.ctypes.data, C.byref(res)))
        self._sweep_done = False
        self._benefit_done = None
        return dict(updated=bool(res.updated), any_on=bool(res.any_on), threshold=res.threshold,
                    normaliser=res.normaliser, ubar0=res.ubar0, strat_size=res.strat_size,
                    n_bins=res.n_bins, argmax_margin=res.argmax_margin, thr_code=res.thr_code, contig_on=on.astype(bool))

    def set_overlap(self, on):
        """Allow / forbid the chain of update_benefit to run next to the sweep of update_begin."""
        self._ck(self.lib.bossx_set_overlap(self.h, int(bool(on))))

    def arm(self):
        self._ck(self.lib.bossx_arm(self.h))

    def get_max(self):
        mx = C.c_double(0.0)
        self._ck(self.lib.bossx_get_max(self.h, C.byref(mx)))
        return mx.value

    # ---- read-start counts resident in HBM (readstartdist.py:24-82) --------------------------------
    fhat_resident = True        # update() accepts `fhat_model` instead of the host-built posterior

    def fhat_reset(self, counts, n_windows):
        """bossx_fhat_reset: install counts float64[n_windows, 2] (None: zeros)."""
        c = None if counts is None else np.ascontiguousarray(counts, dtype=np.float64)
        self._ck(self.lib.bossx_fhat_reset(self.h, None if c is None else c.ctypes.data, int(n_windows)))

    def fhat_add(self, keys):
        """bossx_fhat_add: one read start at every flat index (window * 2 + strand) of `keys`."""
        k = np.ascontiguousarray(keys, dtype=np.int64)
        if k.size:
            self._ck(self.lib.bossx_fhat_add(self.h, k.ctypes.data, int(k.size)))

    # ---- native multi-GPU driver: RCCL called by the library on the engine's stream (bossx_dist_*) --------
    def dist_unique_id(self):
        """bossx_dist_unique_id: the communicator id rank 0 shares with the other ranks (uint8[128])."""
        buf = np.zeros(128, dtype=np.uint8)
        rc = self.lib.bossx_dist_unique_id(buf.ctypes.data)
        if rc:
            raise _lib.BossxError("bossx_dist_unique_id failed (%d): librccl could not be loaded" % rc)
        return buf

    def dist_init(self, uid, rank, world):
        uid = np.ascontiguousarray(uid, dtype=np.uint8)
        assert uid.size == 128
        self._ck(self.lib.bossx_dist_init(self.h, uid.ctypes.data, int(rank), int(world)))
        self.dist_native = True
        self.dist_world = int(world)

    def dist_chain(self, windows, mult):
        """bossx_dist_chain: the global "some strategy is on" exchange (until it is) + the move_sum chain."""
        w = np.ascontiguousarray(windows, dtype=np.int32)
        m = np.ascontiguousarray(mult, dtype=np.float64)
        assert w.shape == (_lib.NWIN,) and m.shape == (10,)
        self._ck(self.lib.bossx_dist_chain(self.h, w.ctypes.data, m.ctypes.data))
        self._benefit_done = tuple(w.tolist())

    def dist_allgather(self, arr):
        """bossx_dist_allgather: same-shape arrays from every rank, stacked on a new leading axis (through the
        engine's own communicator and stream)."""
        arr = np.ascontiguousarray(arr)
        out = np.empty((self.dist_world,) + arr.shape, dtype=arr.dtype)
        self._ck(self.lib.bossx_dist_allgather(self.h, arr.ctypes.data, out.ctypes.data, arr.nbytes))
        return out

    @property
    def dist_collectives(self):
        return int(self.lib.bossx_dist_collectives(self.h))

    def chain_stats(self):
        """Counters of the chunk-parallel benefit chain since finalize (include/bossx.h: bossx_chain_stats)."""
        out = (C.c_int64 * 4)()
        self._ck(self.lib.bossx_chain_stats(self.h, out))
        return dict(chunk_parallel_launches=int(out[0]), serial_launches_while_paused=int(out[1]),
                    chunks_added_plainly=int(out[2]), failed_checks=int(out[3]))

    def chain_counters(self):
        """Device-side counters of the chunk-parallel chain (include/bossx.h: bossx_chain_counters; BOSSX_SPEC_STATS must be set)."""
        out = (C.c_int64 * 8)()
        self._ck(self.lib.bossx_chain_counters(self.h, out))
        keys = ("rows_built", "rows_left_standing", "rows_without_table", "strided_rows_built", "strided_rows_looked_up",
                "strided_rows_undecided", "groups_stepped", "groups_walked")
        return dict(zip(keys, (int(v) for v in out)))

    def update(self, bucket_threshold, windows=None, mult=None, tc=0.0, fhat_c=None, target_rs=0,
               want_stats=False, bits=False, fhat_model=None, dist=False, between=None):
        """bossx_update (`dist`: bossx_dist_update): one fused decision update.  Without `fhat_c` only the sweep and the
        bucket switches run.  Returns dict(updated, any_on, threshold, normaliser, ubar0,
        strat_size, n_bins, contig_on[, counts, fgrid_fx, ubar_fx]); masks land in
        `self.strat_all` (bytes of every non-rejected contig, add order) or, with `bits`,
        packed 8:1 in `self.strat_bits` (bossx_get_strat_bits layout).  `between`: host work done while the
        enqueued update runs (bossx_update_launch -> between() -> bossx_update_collect), e.g. staging the next
        batch into the other slot; it must not touch this update's inputs or results."""
        up = _lib.UpdateParams()
        f = None
        if fhat_model is not None and fhat_c is None:
            # the posterior is rebuilt on the device from the resident counts (BOSSX_UPDATE_FHAT_RESIDENT)
            w = np.ascontiguousarray(windows, dtype=np.int32)
            m = np.ascontiguousarray(mult, dtype=np.float64)
            assert w.shape == (_lib.NWIN,) and m.shape == (10,)
            for i in range(_lib.NWIN):
                up.windows[i] = int(w[i])
            for i in range(10):
                up.mult[i] = float(m[i])
            up.n_windows = int(fhat_model["n_windows"])
            up.target_rs = int(fhat_model["target_rs"])
            up.fhat_alpha = float(fhat_model["alpha"]); up.fhat_den = float(fhat_model["den"])
            up.fhat_expected = float(fhat_model["expected"]); up.fhat_on_target = float(fhat_model["on_target"])
        if fhat_c is not None:
            w = np.ascontiguousarray(windows, dtype=np.int32)
            m = np.ascontiguousarray(mult, dtype=np.float64)
            assert w.shape == (_lib.NWIN,) and m.shape == (10,)
            for i in range(_lib.NWIN):
                up.windows[i] = int(w[i])
            for i in range(10):
                up.mult[i] = float(m[i])
            f = np.ascontiguousarray(fhat_c, dtype=np.float64)
            up.fhat_c = f.ctypes.data
            up.n_windows = f.shape[0]
            up.target_rs = int(target_rs)
        up.tc = float(tc)
        up.bucket_threshold = float(bucket_threshold)
        up.flags = 1 if getattr(self, "_sweep_done", False) else 0
        if fhat_model is not None and fhat_c is None:
            up.flags |= 8
        done = getattr(self, "_benefit_done", None)
        if done is not None and (fhat_c is not None or fhat_model is not None) and done == tuple(int(x) for x in windows):
            up.flags |= 2
        self._sweep_done = False
        self._benefit_done = None
        if bits:
            up.flags |= 4
            if getattr(self, "strat_bits", None) is None:
                self.strat_bits = self._host_buffer(max(int(self.lib.bossx_strat_bits_bytes(self.h)), 1), 0xFF)
            masks = self.strat_bits
        else:
            if getattr(self, "strat_all", None) is None:
                self.strat_all = self._host_buffer(max(int(self.lib.bossx_strat_bytes(self.h)), 1), 1)
            masks = self.strat_all
            # BOSSX_UPDATE_STRAT_DELTA: nobody but the engine writes into strat_all (the views handed out are read-only), so an update
            # need only write the masks that changed since the one before
            up.flags |= 16
        on = np.zeros(len(self.names), dtype=np.uint8)
        res = _lib.UpdateResult()
        counts = fg = ub = None
        between_error = None
        if want_stats:
            counts = np.zeros(_lib.HIST_BINS, dtype=np.int64)
            fg = np.zeros((_lib.HIST_BINS, 2), dtype=np.uint64)
            ub = np.zeros(2, dtype=np.uint64)
        if dist:        # bossx_dist_update: the same update with the library's own RCCL collectives between the stages
            a = (self.h, C.byref(up), masks.ctypes.data, on.ctypes.data, C.byref(res))
            if between is None:
                self._ck(self.lib.bossx_dist_update(*a))
            else:
                self._ck(self.lib.bossx_dist_update_launch(*a))
                try:
                    between()
                except BaseException as e:       # the update is collected and applied first (the engine's state has advanced)
                    between_error = e
                self._ck(self.lib.bossx_dist_update_collect(*a))
        else:
            a = (self.h, C.byref(up), masks.ctypes.data, on.ctypes.data, C.byref(res), None if counts is None else counts.ctypes.data,
                 None if fg is None else fg.ctypes.data, None if ub is None else ub.ctypes.data)
            if between is None:
                self._ck(self.lib.bossx_update(*a))
            else:
                self._ck(self.lib.bossx_update_launch(*a))
                try:
                    between()
                except BaseException as e:       # the update is collected and applied first (the engine's state has advanced)
                    between_error = e
                self._ck(self.lib.bossx_update_collect(*a))
        out = dict(updated=bool(res.updated), any_on=bool(res.any_on), threshold=res.threshold,
                   normaliser=res.normaliser, ubar0=res.ubar0, strat_size=res.strat_size,
                   n_bins=res.n_bins, argmax_margin=res.argmax_margin, thr_code=res.thr_code, contig_on=on.astype(bool))
        if want_stats:
            out.update(counts=counts, fgrid_fx=fg, ubar_fx=ub)
        if between_error is not None:
            out["between_error"] = between_error        # re-raised by the caller once it has applied this update's results
        return out

    def strat_view(self, contig):
        """Zero-copy bool view [T,2,nb] of one contig's mask inside `strat_all` (the same object
        every time: the buffer lives as long as the engine)."""
        cache = self.__dict__.setdefault("_strat_views", {})
        v = cache.get(contig)
        if v is None or v.base is None or self.strat_all is not cache.get("_buf"):
            if self.strat_all is not cache.get("_buf"):
                cache.clear()
                cache["_buf"] = self.strat_all
            off = int(self.lib.bossx_strat_offset(self.h, contig))
            T = self.lengths[contig] // 100
            v = cache[contig] = self.strat_all[off: off + T * 2 * self.nb].view(np.bool_).reshape(T, 2, self.nb)
            v.setflags(write=False)          # (the buffer mirrors the device's masks: see BOSSX_UPDATE_STRAT_DELTA)
        return v

    def strat_offset(self, contig):
        return int(self.lib.bossx_strat_offset(self.h, contig))

    def get_strat_bits(self, out=None):
        """bossx_get_strat_bits: every mask packed 8:1 (np.packbits order)."""
        n = max(int(self.lib.bossx_strat_bits_bytes(self.h)), 1)
        if out is None:
            out = np.empty(n, dtype=np.uint8)
        assert out.size >= n and out.flags.c_contiguous
        self._ck(self.lib.bossx_get_strat_bits(self.h, out.ctypes.data))
        return out

    def apply_threshold(self, threshold):
        self._ck(self.lib.bossx_apply_threshold(self.h, float(threshold)))

    def get_strat(self, contig, out=None):
        if self.rejected[contig]:
            return np.zeros(1, dtype=bool)
        shape = (self.lengths[contig] // 100, 2, self.nb)
        if out is None:
            out = np.empty(shape, dtype=bool)
        assert out.shape == shape and out.flags.c_contiguous
        self._ck(self.lib.bossx_get_strat(self.h, contig, out.ctypes.data))
        return out

    # ---- introspection -------------------------------------------------------------------
    @property
    def n_sites(self):
        return self.lib.bossx_n_sites(self.h)

    @property
    def matrix_chain(self):
        return bool(self.lib.bossx_matrix_chain(self.h))

    @property
    def merged_bins(self):
        return self.lib.bossx_merged_bins(self.h)

    def export(self, contig, which):
        """Device state of one contig in the REFERENCE's layout."""
        L, nb = self.lengths[contig], self.nb
        code = EXPORT[which]
        if which == "coverage":
            raw = np.empty((nb, 5, L), dtype=np.uint16)
        elif which in ("scores", "entropy"):
            raw = np.empty((nb, L), dtype=np.float64)
        elif which == "scores_ds":
            raw = np.empty((nb, L // 100 + 1), dtype=np.float64)
        elif which == "benefit":
            raw = np.empty((nb, 2, L // 100 + 1), dtype=np.float64)
        elif which == "state":
            raw = np.empty((nb, L), dtype=np.uint8)
        elif which == "bucket_switches":
            raw = np.empty((nb, L // 20000 + 1), dtype=np.uint8)
        elif which == "benefit_tail":
            k = min(L // 100 + 1, sum(1 for r in self.rejected if not r))
            raw = np.empty((nb, 2, k), dtype=np.float64)
        else:
            raw = np.empty(L, dtype=np.uint8)
        self._ck(self.lib.bossx_export(self.h, contig, code, raw.ctypes.data, raw.nbytes))
        if which == "coverage":
            return np.ascontiguousarray(raw.transpose(2, 1, 0))          # [L,5,nb]
        if which in ("scores", "entropy", "scores_ds", "state"):
            return np.ascontiguousarray(raw.T)                           # [L,nb]
        if which == "bucket_switches":
            return np.ascontiguousarray(raw.T).astype(bool)              # [n_buckets,nb]
        if which in ("benefit", "benefit_tail"):
            return np.ascontiguousarray(raw.transpose(2, 1, 0))          # [T+1 (or tail),2,nb]
        return raw

    def import_state(self, contig, which, arr):
        L, nb = self.lengths[contig], self.nb
        if which == "coverage":
            raw = np.ascontiguousarray(np.asarray(arr, dtype=np.uint16).transpose(2, 1, 0))
        elif which == "entropy":
            raw = np.ascontiguousarray(np.asarray(arr, dtype=np.float64).T)
        elif which == "scores_ds":
            raw = np.ascontiguousarray(np.asarray(arr, dtype=np.float64).T)
        elif which == "state":
            raw = np.ascontiguousarray(np.asarray(arr, dtype=np.uint8).T)
        elif which == "touched":
            raw = np.ascontiguousarray(arr, dtype=np.uint8)
        elif which == "bucket_switches":
            raw = np.ascontiguousarray(np.asarray(arr, dtype=np.uint8).T)
        elif which == "strat":
            raw = np.ascontiguousarray(arr, dtype=np.uint8)
        else:
            raise ValueError(which)
        code = 9 if which == "strat" else EXPORT[which]
        self._ck(self.lib.bossx_import(self.h, contig, code, raw.ctypes.data, raw.nbytes))
        if which == "strat" and getattr(self, "strat_all", None) is not None:
            # strat_all mirrors the device's masks (Contig.strat views point into it): masks that come in from outside go there too
            # — an update only writes the rows of switched-on buckets, and with BOSSX_UPDATE_STRAT_DELTA only the rows that change
            off = int(self.lib.bossx_strat_offset(self.h, contig))
            self.strat_all[off: off + raw.size] = raw.reshape(-1)

    def preload_coverage(self, depth, seed=1):
        self._ck(self.lib.bossx_preload_coverage(self.h, float(depth), int(seed)))

    # ---- measurement ---------------------------------------------------------------------
    def enable_timing(self, on=True, only=None):
        """HIP events around every kernel (`on`), or around kernel `only` (a name of _lib.K_NAMES) alone."""
        code = 0 if not on else (1 if only is None else 2 + _lib.K_NAMES.index(only))
        self._ck(self.lib.bossx_enable_timing(self.h, code))

    def kernel_stats(self):
        n = len(_lib.K_NAMES)
        last = np.zeros(n, np.float32)
        total = np.zeros(n, np.float64)
        launches = np.zeros(n, np.int64)
        nbytes = np.zeros(n, np.float64)
        self._ck(self.lib.bossx_kernel_ms(self.h, last.ctypes.data, total.ctypes.data, launches.ctypes.data))
        self._ck(self.lib.bossx_kernel_bytes(self.h, nbytes.ctypes.data))
        return {k: dict(ms_last=float(last[i]), ms_total=float(total[i]), launches=int(launches[i]),
                        bytes_last=float(nbytes[i])) for i, k in enumerate(_lib.K_NAMES)}

    def synchronize(self):
        self._ck(self.lib.bossx_synchronize(self.h))


def host_parse(contigs, paf_text, seqs, barcodes=None, nbarcodes=1, min_len=200, n_threads=0, expand=True):
    """bossx_host_parse: the native PAF/CIGAR front end without a device (CPU test hook).
    `contigs`: list of (name, length, flags).  Returns the per-mapping summary plus, with
    `expand`, one entry per aligned reference base: contig index, position, code (0..3 ACGT,
    4 deletion, 255 other), barcode."""
    lib = _lib.load()
    ids = list(seqs.keys())
    vals = list(seqs.values())
    n = len(ids)
    nptr, nlen = Engine._str_pointers(ids)
    sptr, slen = Engine._str_pointers(vals)
    cnames = [c[0].encode() for c in contigs]
    cptr = (C.c_char_p * max(len(contigs), 1))(*cnames)
    clen = np.ascontiguousarray([c[1] for c in contigs] or [0], dtype=np.int64)
    cflag = np.ascontiguousarray([c[2] for c in contigs] or [0], dtype=np.int32)
    paf = paf_text.encode() if isinstance(paf_text, str) else bytes(paf_text)
    bc = None
    if barcodes is not None:
        bc = np.ascontiguousarray([barcodes[i] for i in ids] if isinstance(barcodes, dict) else barcodes, dtype=np.int32)
    s = dict(read_idx=np.zeros(max(n, 1), np.int32), contig_idx=np.zeros(max(n, 1), np.int32),
             rev=np.zeros(max(n, 1), np.uint8), tstart=np.zeros(max(n, 1), np.int64),
             tend=np.zeros(max(n, 1), np.int64), qlen=np.zeros(max(n, 1), np.int64))
    summ = _lib.BatchSummary(*[s[k].ctypes.data for k in ("read_idx", "contig_idx", "rev", "tstart", "tend", "qlen")])
    n_rec = C.c_int32(0)
    aligned = C.c_int64(0)
    errbuf = C.create_string_buffer(512)

    def call(oc, op, ocode, obc, cap):
        rc = lib.bossx_host_parse(C.cast(cptr, C.c_void_p), clen.ctypes.data, cflag.ctypes.data, len(contigs),
                                  int(nbarcodes), paf, len(paf), nptr.ctypes.data, nlen.ctypes.data,
                                  sptr.ctypes.data, slen.ctypes.data, None if bc is None else bc.ctypes.data,
                                  n, int(min_len), int(n_threads), C.byref(summ), C.byref(n_rec), C.byref(aligned),
                                  oc, op, ocode, obc, cap, errbuf, len(errbuf))
        if rc:
            raise _lib.ERRORS.get(rc, _lib.BossxError)(errbuf.value.decode("utf-8", "replace"))
    call(None, None, None, None, 0)
    out = {key: v[:n_rec.value] for key, v in s.items()}
    out["aligned"] = aligned.value
    out["ids"] = ids
    if expand:
        m = max(aligned.value, 1)
        oc, op = np.zeros(m, np.int32), np.zeros(m, np.int64)
        ocode, obc = np.zeros(m, np.uint8), np.zeros(m, np.uint8)
        call(oc.ctypes.data, op.ctypes.data, ocode.ctypes.data, obc.ctypes.data, m)
        k = aligned.value
        out.update(contig=oc[:k], pos=op[:k], code=ocode[:k], barcode=obc[:k])
    return out
