"""Read-length distribution — host side, 4000 values per update.

Same interface and results as /root/reference/boss/readlengthdist.py:8-97
(`ReadlengthDist.update`, `ccl_approx_constant`, attributes `lam`, `L`, `ccl`, `approx_ccl`,
`longest_read`, `time_cost`), vectorised: the per-read Python loop becomes one bincount and the
10-step `while` scan becomes searchsorted on the (monotone) complementary CDF.
"""
import ctypes as C
import logging

import numpy as np


def _native():
    """libbossx.so if it is built (host-only entry point: no GPU needed), else None."""
    global _LIB
    if _LIB is False:
        try:
            from . import _lib
            _LIB = _lib.load()
        except Exception:
            _LIB = None
    return _LIB


_LIB = False


class ReadlengthDist:
    def __init__(self, mu=400, sd=4000, lam=6000, eta=11):
        self.sd, self.lam, self.eta, self.mu = sd, lam, eta, mu
        self.read_lengths = np.zeros(int(1e6), dtype='uint16')
        self._L = self._ccl = None
        x = np.arange(int(lam + 10 * sd), dtype='int')
        dens = np.exp(-((x - lam + 1) ** 2) / (2 * (sd ** 2))) / (sd * np.sqrt(2 * np.pi))
        # the reference normalises with Python's sequential sum(); cumsum adds in the same order
        dens /= np.cumsum(dens)[-1]
        self.L = dens
        self.approx_ccl = self.ccl_approx_constant()

    def update(self, read_lengths):
        """`read_lengths`: {read id: length} (as the reference) or an integer array.  Runs in the
        native library (bossx_rl_update) when it is built — this step sits on the critical path
        of an update — and in numpy otherwise; both give bit-identical results."""
        if isinstance(read_lengths, dict):
            lens = np.fromiter(read_lengths.values(), dtype=np.int64, count=len(read_lengths))
        else:
            lens = np.ascontiguousarray(read_lengths, dtype=np.int64)
        lib = _native()
        if lib is None:
            return self._update_numpy(lens)
        hi = C.c_int64(getattr(self, "_hi", 0))
        lam, longest, observed = C.c_double(0.0), C.c_int64(0), C.c_int32(0)
        approx = np.empty(self.eta - 1, dtype=np.int32)
        rc = lib.bossx_rl_update(self.read_lengths.ctypes.data, self.read_lengths.size, lens.ctypes.data, lens.size,
                                 int(self.mu * 2), int(self.eta), C.byref(hi), C.byref(lam), C.byref(longest),
                                 approx.ctypes.data, C.byref(observed))
        if rc:
            raise ValueError("bossx_rl_update failed (%d)" % rc)
        self._hi = hi.value
        if not observed.value:
            logging.info('Attempted update of read lengths before observing any reads')
            return
        self.longest_read = np.int64(longest.value)
        self.lam = np.float64(lam.value)
        self._counts = None                           # built from the histogram when L / ccl are asked for
        self._L = self._ccl = None
        self.approx_ccl = approx
        logging.info('rld: %s', self.approx_ccl)
        self.time_cost = self.lam - 400 - 300

    def _update_numpy(self, lens):
        lens = lens[lens > self.mu * 2]
        if lens.size:
            lens = np.minimum(lens, int(1e6) - 1)
            self._hi = max(getattr(self, "_hi", 0), int(lens.max()))
            # unbuffered in-place add: uint16 counters wrap exactly like the reference's
            # repeated `+= 1` (readlengthdist.py:23,48)
            np.add.at(self.read_lengths, lens, np.uint16(1))
        # only lengths up to the longest one seen so far can be non-zero
        counts = self.read_lengths[:getattr(self, "_hi", 0) + 1].astype(np.int64)
        total = int(counts.sum())
        if total == 0:
            logging.info('Attempted update of read lengths before observing any reads')
            return
        # uint16 wrap-around can zero the top counters: the reference takes the last non-zero one
        hi = counts.size - 1
        while counts[hi] == 0:
            hi -= 1
        counts = counts[:hi + 1]
        self.longest_read = np.int64(hi)
        self.lam = np.int64(np.dot(np.arange(hi + 1, dtype=np.int64), counts)) / np.uint64(total)
        self._counts = counts
        self._L = self._ccl = None                    # materialised on access
        self.approx_ccl = self._approx_lean()
        logging.info('rld: %s', self.approx_ccl)
        self.time_cost = self.lam - 400 - 300

    # `L` (pmf) and `ccl` are attributes of the reference object (readlengthdist.py:60,84); here
    # they are built from the integer histogram when somebody looks at them.
    @property
    def L(self):
        if self._L is None:
            if self._counts is None:
                self._counts = self.read_lengths[:int(self.longest_read) + 1].astype(np.int64)
            dens = self._counts.astype('float64')
            dens /= float(self._counts.sum())   # = the reference's sum(L): integer-valued, exact
            self._L = dens
        return self._L

    @L.setter
    def L(self, value):
        self._L = value

    @property
    def ccl(self):
        if self._ccl is None:
            self.ccl_approx_constant()
        return self._ccl

    @ccl.setter
    def ccl(self, value):
        self._ccl = value

    def _approx_lean(self):
        """`ccl_approx_constant` without materialising `ccl`: the same float cumsum of L[1:]
        (bit-identical partial sums), the crossing of each level located by bisection on the
        monotone partial sums and settled with the reference's own expression `1 - cs > prob`.
        The 1e-6 cut-off and the trailing-zero trim of the reference do not move any crossing:
        ccl is non-increasing, so the first entry <= prob is the same before and after them."""
        total = float(self._counts.sum())
        n = self._counts.size - 1                     # len(L[1:])
        probs = [1 - (part + 0.5) / (self.eta - 1) for part in range(self.eta - 1)]
        # the last level (0.05) is crossed near the previous update's last index: sum that far
        # first (a prefix of the cumsum is bit-identical to the cumsum's prefix), all of it if not
        m = min(n, int(self.approx_ccl[-1] * 1.25) + 64)
        while True:
            cs = np.cumsum(self._counts[1:m + 1] / total)     # = L[1:m+1].cumsum()
            if m == n or 1.0 - cs[-1] <= probs[-1]:
                break
            m = n
        j = np.searchsorted(cs, 1.0 - np.array(probs), side='left')
        out = np.empty(len(probs), dtype='int32')
        for k, prob in enumerate(probs):
            i = int(j[k])
            while i > 0 and not (1.0 - cs[i - 1] > prob):
                i -= 1
            while i < cs.size and 1.0 - cs[i] > prob:
                i += 1
            out[k] = i + 1            # ccl[i + 1] = 1 - cs[i]; i == n is the appended final zero
        return out

    def ccl_approx_constant(self):
        ccl = np.zeros(len(self.L) + 1)
        ccl[0] = 1
        ccl[1:] = 1 - np.concatenate((self.L[1:].cumsum(), np.ones(1)))
        ccl[ccl < 1e-6] = 0
        ccl = np.concatenate((np.trim_zeros(ccl, trim='b'), np.zeros(1)))
        self.ccl = ccl
        probs = np.array([1 - (part + 0.5) / (self.eta - 1) for part in range(self.eta - 1)])
        # first index with ccl <= prob; ccl is non-increasing, so this equals the reference's scan
        return np.searchsorted(-ccl, -probs, side='left').astype('int32')
