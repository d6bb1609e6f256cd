"""Streaming writer of the reference's output file: ``np.savez(path, event_stream=events)`` (v2ce.py:371-372),
produced while the events are still arriving.

The reference concatenates the whole clip in host memory and writes it at the end (v2ce.py:363-372).  Here the
packed 13-byte records of every batch go to the file as soon as their D2H copy has landed (``pipeline.EventSink``
with a writer: a small ring of pinned staging buffers and one writer thread), so the disk works under the GPU and the
host never holds the clip.  The file is the one ``np.savez`` writes -- a ZIP archive (stored, ZIP64) with the single
member ``event_stream.npy`` (format 1.0 header + raw records) -- and ``np.load(path)["event_stream"]`` returns the same
array; only the archive's bookkeeping differs from numpy's byte for byte (timestamps, the fixed-size .npy header
padding, ZIP64 fields always present).

The record count is not known until the clip ends, so the .npy header (fixed 256 bytes) and the ZIP sizes / CRC are
patched when the file is closed; the CRC-32 of header + data is combined from the two parts (zlib's crc32_combine,
restated below: Python's zlib does not export it).
"""
from __future__ import annotations

import concurrent.futures
import functools
import os
import struct
import time
import zlib

import numpy as np

_NPY_HEADER_BYTES = 256                  # magic (6) + version (2) + length (2) + dict padded with spaces + "\n"


def _gf2_times(mat, vec):
    s, i = 0, 0
    while vec:
        if vec & 1:
            s ^= mat[i]
        vec >>= 1
        i += 1
    return s


def _gf2_square(mat):
    return [_gf2_times(mat, mat[n]) for n in range(32)]


def _gf2_compose(a, b):
    """The operator 'b, then a' (matrix product a b; matrices are lists of 32 column words)."""
    return [_gf2_times(a, b[n]) for n in range(32)]


@functools.lru_cache(maxsize=64)
def _zeros_operator(nbytes: int):
    """GF(2) operator that advances a CRC-32 register over `nbytes` zero bytes (zlib's crc32_combine builds the same
    powers of the one-zero-bit operator on the fly; cached here per length: the writer folds many equal-sized pieces)."""
    op = [0xEDB88320] + [1 << n for n in range(31)]         # one zero bit
    op = _gf2_square(_gf2_square(_gf2_square(op)))          # one zero byte
    acc = None
    while nbytes:
        if nbytes & 1:
            acc = op if acc is None else _gf2_compose(op, acc)
        nbytes >>= 1
        if nbytes:
            op = _gf2_square(op)
    return tuple(acc) if acc is not None else None


def crc32_combine(crc1: int, crc2: int, len2: int) -> int:
    """CRC-32 of A || B from crc32(A), crc32(B) and len(B) (zlib's crc32_combine)."""
    if len2 <= 0:
        return crc1
    return _gf2_times(_zeros_operator(len2), crc1) ^ crc2


def npy_header(dtype: np.dtype, count: int) -> bytes:
    """The fixed-size (256-byte) format-1.0 header of a 1-D array of `count` records."""
    d = "{'descr': %r, 'fortran_order': False, 'shape': (%d,), }" % (np.lib.format.dtype_to_descr(np.dtype(dtype)), count)
    pad = _NPY_HEADER_BYTES - 10 - len(d) - 1
    if pad < 0:
        raise ValueError("dtype description does not fit the fixed .npy header")
    body = (d + " " * pad + "\n").encode("latin1")
    return b"\x93NUMPY\x01\x00" + struct.pack("<H", len(body)) + body


class NpzStreamWriter:
    """``with NpzStreamWriter(path, "event_stream", EVENT_DTYPE) as w: w.write(buffer) ...`` -- `buffer`: any
    bytes-like object holding whole records."""

    def __init__(self, path: str, key: str, dtype):
        self.path, self.dtype = path, np.dtype(dtype)
        self.name = (key + ".npy").encode("ascii")
        # the file grows under a temporary name and takes its final name when it is complete: a clip that dies half-way
        # (or is repeated by the range guard) never leaves a truncated archive under the final name
        self.part_path = path + ".part"
        self.f = open(self.part_path, "wb")
        t = time.localtime()
        self.dostime = (t.tm_hour << 11) | (t.tm_min << 5) | (t.tm_sec // 2)
        self.dosdate = ((t.tm_year - 1980) << 9) | (t.tm_mon << 5) | t.tm_mday
        self.f.write(self._local_header(0, 0))
        self.data_offset = self.f.tell()
        self.f.write(npy_header(self.dtype, 0))             # placeholder, patched by close()
        self.crc_data, self.nbytes, self.closed = 0, 0, False

    def _local_header(self, crc: int, size: int) -> bytes:
        extra = struct.pack("<HHQQ", 1, 16, size, size)     # ZIP64: uncompressed, compressed size
        return struct.pack("<IHHHHHIIIHH", 0x04034B50, 45, 0, 0, self.dostime, self.dosdate, crc, 0xFFFFFFFF, 0xFFFFFFFF,
                           len(self.name), len(extra)) + self.name + extra

    _PIECE = 16 << 20                                       # CRC work unit (equal pieces share one cached combine operator)
    _pool = None

    def write(self, buf) -> None:
        mv = memoryview(buf).cast("B")
        if len(mv) % self.dtype.itemsize:
            raise ValueError("partial record")
        if len(mv) <= 2 * self._PIECE:
            self.f.write(mv)
            self.crc_data = zlib.crc32(mv, self.crc_data)
        else:
            # zlib.crc32 (~2 GB/s on one core) is the bottleneck of a multi-GB events file: the pieces' CRCs are computed on
            # a few threads (zlib releases the GIL) while this thread writes, then folded with crc32_combine
            if NpzStreamWriter._pool is None:
                NpzStreamWriter._pool = concurrent.futures.ThreadPoolExecutor(max_workers=8)
            parts = [mv[i:i + self._PIECE] for i in range(0, len(mv), self._PIECE)]
            futs = [NpzStreamWriter._pool.submit(zlib.crc32, p) for p in parts]
            self.f.write(mv)
            for p, fu in zip(parts, futs):
                self.crc_data = crc32_combine(self.crc_data, fu.result(), len(p))
        self.nbytes += len(mv)

    @property
    def count(self) -> int:
        return self.nbytes // self.dtype.itemsize

    def close(self) -> None:
        if self.closed:
            return
        self.closed = True
        f = self.f
        header = npy_header(self.dtype, self.count)
        size = len(header) + self.nbytes
        crc = crc32_combine(zlib.crc32(header), self.crc_data, self.nbytes) & 0xFFFFFFFF
        cd_offset = f.tell()
        extra = struct.pack("<HHQQQ", 1, 24, size, size, 0)  # ZIP64: sizes + offset of the local header
        cd = struct.pack("<IHHHHHHIIIHHHHHII", 0x02014B50, 45, 45, 0, 0, self.dostime, self.dosdate, crc, 0xFFFFFFFF, 0xFFFFFFFF,
                         len(self.name), len(extra), 0, 0, 0, 0o600 << 16, 0xFFFFFFFF) + self.name + extra
        f.write(cd)
        eocd64_offset = f.tell()
        f.write(struct.pack("<IQHHIIQQQQ", 0x06064B50, 44, 45, 45, 0, 0, 1, 1, len(cd), cd_offset))
        f.write(struct.pack("<IIQI", 0x07064B50, 0, eocd64_offset, 1))
        f.write(struct.pack("<IHHHHIIH", 0x06054B50, 0, 0, 1, 1, 0xFFFFFFFF, 0xFFFFFFFF, 0))
        f.seek(0)
        f.write(self._local_header(crc, size))
        f.write(header)
        f.close()
        os.replace(self.part_path, self.path)

    @property
    def records_start(self) -> int:
        """File offset of the first record (behind the ZIP local header and the fixed .npy header)."""
        return self.data_offset + _NPY_HEADER_BYTES

    def set_external(self, nbytes: int, crc_data: int) -> None:
        """The records were written into the file by others (``dist.HostDirectGather``: every rank writes its own
        pieces at their offsets behind ``records_start``): take over their total size and CRC-32 before ``close()``."""
        self.f.flush()
        self.f.seek(self.records_start + nbytes)
        self.nbytes, self.crc_data = int(nbytes), int(crc_data)

    def abort(self) -> None:
        """Drop the partial file (error path)."""
        if not self.closed:
            self.closed = True
            try:
                self.f.close()
            finally:
                if os.path.exists(self.part_path):
                    os.unlink(self.part_path)

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        if exc and exc[0] is not None:
            self.abort()
        else:
            self.close()
        return False
