"""Bit-packed mask file and its memory-mapped reader.

The reference publishes the strategy as `masks/boss.npz` (runs/core.py:59-69) and readfish
re-reads the whole archive whenever its mtime changes (dynamic_readfish.py:87-110), then answers
every decision with `arr[:, reverse(, barcode)][start_pos // 100]` (dynamic_readfish.py:169-210).
At 3 Gb that is 62 MB of bool per update through zip + np.load.  This module keeps the same
answers with a file that is 8x smaller, written straight from the device's packed buffer
(bossx_get_strat_bits) and mapped, not parsed, by the consumer:

    boss.bits :=  MAGIC(8) | header_len u64 LE | header JSON | zero pad to 64 B | packed bits

The header lists, per contig in add order: name, rows (length // 100), rejected, and the bit
offset of its first element; `nbarcodes` and `scale` (100) are global.  Element
(row, strand, barcode) of a contig is bit  bit_off + (row*2 + strand)*nbarcodes + barcode,
stored in np.packbits ("big") order.  `boss.npz` stays available: `MaskFile.to_dict()` returns
exactly the arrays the reference would have saved.
"""
import json
import os
from pathlib import Path

import numpy as np

MAGIC = b"BOSSXM1\n"
ALIGN = 64


def write_mask_bits(path, contigs, bits, nbarcodes, scale=100) -> None:
    """`contigs`: iterable of (name, rows, rejected, bit_off) in add order; `bits`: the packed
    buffer.  Written as tmp + rename like the reference's npz (core.py:66-69)."""
    header = json.dumps({
        "nbarcodes": int(nbarcodes), "scale": int(scale), "n_bytes": int(len(bits)),
        "contigs": [{"name": n, "rows": int(r), "rejected": bool(rej), "bit_off": int(off)}
                    for n, r, rej, off in contigs]}).encode()
    pre = len(MAGIC) + 8 + len(header)
    pad = (-pre) % ALIGN
    path = Path(path)
    tmp = path.with_name(path.name + ".tmp")
    with open(tmp, "wb") as fh:
        fh.write(MAGIC)
        fh.write(len(header).to_bytes(8, "little"))
        fh.write(header)
        fh.write(b"\0" * pad)
        fh.write(memoryview(np.ascontiguousarray(bits, dtype=np.uint8)))
    os.replace(tmp, path)


class MaskFile:
    """One mapped snapshot of a bit-packed mask file."""

    def __init__(self, path):
        with open(path, "rb") as fh:
            if fh.read(len(MAGIC)) != MAGIC:
                raise ValueError("not a bossx mask file: %s" % path)
            hlen = int.from_bytes(fh.read(8), "little")
            hdr = json.loads(fh.read(hlen))
        pre = len(MAGIC) + 8 + hlen
        self.nbarcodes = hdr["nbarcodes"]
        self.scale = hdr["scale"]
        self.contigs = {c["name"]: (c["rows"], c["rejected"], c["bit_off"]) for c in hdr["contigs"]}
        n = hdr["n_bytes"]
        self.bits = np.memmap(path, dtype=np.uint8, mode="r", offset=pre + (-pre) % ALIGN, shape=(n,)) if n else np.zeros(0, np.uint8)

    def lookup(self, contig, row, strand, barcode=0):
        rows, rej, off = self.contigs[contig]
        if row < 0:                       # numpy indexing of the reference wraps negatives
            row += rows
        if not 0 <= row < rows or not 0 <= barcode < self.nbarcodes:
            raise IndexError(row)
        bit = off + (row * 2 + int(strand)) * self.nbarcodes + barcode
        return (int(self.bits[bit >> 3]) >> (7 - (bit & 7))) & 1

    def strat(self, contig):
        """Contig.strat as the reference stores it: bool[rows, 2, nb] (bool[1] zeros if rejected)."""
        rows, rej, off = self.contigs[contig]
        if rej:
            return np.zeros(1, dtype=bool)
        n = rows * 2 * self.nbarcodes
        b0, b1 = off >> 3, (off + n + 7) >> 3
        flat = np.unpackbits(np.asarray(self.bits[b0:b1]))[(off & 7):(off & 7) + n]
        return flat.view(np.bool_).reshape(rows, 2, self.nbarcodes)

    def to_dict(self):
        """What np.load('boss.npz') gives the reference's consumer (dynamic_readfish.py:70-73)."""
        return {name: self.strat(name) for name in self.contigs}


class MaskReader:
    """The decision lookup of dynamic_readfish.py:87-110,169-210 over the mapped file: reload
    when the mtime moves, accept everything if the file cannot be read, reject rejected
    contigs, accept unknown contigs and out-of-range coordinates."""

    def __init__(self, path, barcodes_index=None):
        self.path = Path(path)
        self.barcodes_index = barcodes_index or {}
        self.last_mask_mtime = 0.0
        self.masks = None
        self.exception = False

    def reload(self) -> int:
        if not self.path.is_file():
            raise FileNotFoundError("No mask files present")
        mtime = self.path.stat().st_mtime
        if not mtime > self.last_mask_mtime:
            return 0
        try:
            self.masks = MaskFile(self.path)
            self.exception = False
        except Exception:
            self.masks = None
            self.exception = True
        self.last_mask_mtime = mtime
        return 1

    def check_coord(self, contig, start_pos, reverse, barcode=None):
        """`barcode`: 'barcodeNN' (mapped through barcodes_index) or None."""
        if self.exception or self.masks is None:
            return 1
        if contig not in self.masks.contigs:
            return 1
        rows, rej, _ = self.masks.contigs[contig]
        if rej:
            return 0
        try:
            b = 0 if barcode is None else self.barcodes_index[int(barcode.split('barcode')[1])]
            return self.masks.lookup(contig, start_pos // self.masks.scale, int(reverse), b)
        except Exception:
            return 1
