"""Read-length distribution — host side, 4000 values per update.

Same interface and results as /root/reference/boss/readlengthdist.py:8-97
(`ReadlengthDist.update`, `ccl_approx_constant`, attributes `lam`, `L`, `ccl`, `approx_ccl`,
`longest_read`, `time_cost`), vectorised: the per-read Python loop becomes one bincount and the
10-step `while` scan becomes searchsorted on the (monotone) complementary CDF.
"""
import logging

import numpy as np


class ReadlengthDist:
    def __init__(self, mu=400, sd=4000, lam=6000, eta=11):
        self.sd, self.lam, self.eta, self.mu = sd, lam, eta, mu
        self.read_lengths = np.zeros(int(1e6), dtype='uint16')
        x = np.arange(int(lam + 10 * sd), dtype='int')
        dens = np.exp(-((x - lam + 1) ** 2) / (2 * (sd ** 2))) / (sd * np.sqrt(2 * np.pi))
        # the reference normalises with Python's sequential sum(); cumsum adds in the same order
        dens /= np.cumsum(dens)[-1]
        self.L = dens
        self.approx_ccl = self.ccl_approx_constant()

    def update(self, read_lengths):
        """`read_lengths`: {read id: length} (as the reference) or an integer array."""
        if isinstance(read_lengths, dict):
            lens = np.fromiter(read_lengths.values(), dtype=np.int64, count=len(read_lengths))
        else:
            lens = np.asarray(read_lengths, dtype=np.int64)
        lens = lens[lens > self.mu * 2]
        if lens.size:
            lens = np.minimum(lens, int(1e6) - 1)
            self._hi = max(getattr(self, "_hi", 0), int(lens.max()))
            # unbuffered in-place add: uint16 counters wrap exactly like the reference's
            # repeated `+= 1` (readlengthdist.py:23,48)
            np.add.at(self.read_lengths, lens, np.uint16(1))
        # only lengths up to the longest one seen so far can be non-zero
        observed = np.nonzero(self.read_lengths[:getattr(self, "_hi", 0) + 1])[0]
        if observed.size == 0:
            logging.info('Attempted update of read lengths before observing any reads')
            return
        counts = self.read_lengths[observed].astype(np.int64)
        self.lam = np.sum(observed * counts) / np.sum(self.read_lengths[observed])
        self.longest_read = observed[-1]
        dens = self.read_lengths[:self.longest_read + 1].astype('float64')
        dens /= np.sum(dens)          # integer counts: any summation order is exact
        self.L = dens
        self.approx_ccl = self.ccl_approx_constant()
        logging.info(f'rld: {self.approx_ccl}')
        self.time_cost = self.lam - 400 - 300

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
