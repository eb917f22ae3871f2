"""Blocked gzip (BGZF) writer.

A BGZF file is an ordinary multi-member gzip file -- `gzip -d`, Python's gzip module, zlib's gzread and the
reference's readers all take it as it is -- whose members hold at most 64 KB of text each and state their
compressed size in a 'BC' extra field (the SAM/BAM specification, section 4.1).  Members are independent, which
is what lets kv_inflate.hip decode them in parallel on the GPU; kevlar_amd.open(name + '.gz', 'w') writes this
format for that reason.
"""
import struct
import zlib

BLOCK_TEXT = 0xff00            # text bytes per member, as bgzip chooses
_EOF = bytes.fromhex('1f8b08040000000000ff0600424302001b0003000000000000000000')


def member(data, level=6):
    """One BGZF member holding `data` (at most 65536 bytes)."""
    assert len(data) <= 65536
    squeeze = zlib.compressobj(level, zlib.DEFLATED, -15)
    payload = squeeze.compress(data) + squeeze.flush()
    if len(payload) + 26 > 65536:          # incompressible: stored blocks always fit
        squeeze = zlib.compressobj(0, zlib.DEFLATED, -15)
        payload = squeeze.compress(data) + squeeze.flush()
    total = len(payload) + 26
    head = struct.pack('<BBBBIBBHBBHH', 0x1f, 0x8b, 8, 4, 0, 0, 0xff, 6, 0x42, 0x43, 2, total - 1)
    return head + payload + struct.pack('<II', zlib.crc32(data) & 0xffffffff, len(data))


class BgzfWriter(object):
    """File-like sink: write() takes str (encoded as latin-1/ASCII) or bytes."""

    def __init__(self, filename, level=6):
        self._fh = open(filename, 'wb')
        self._held = bytearray()
        self._level = level
        self.name = filename

    def write(self, data):
        if isinstance(data, str):
            data = data.encode('latin-1')
        self._held += data
        while len(self._held) >= BLOCK_TEXT:
            self._fh.write(member(bytes(self._held[:BLOCK_TEXT]), self._level))
            del self._held[:BLOCK_TEXT]
        return len(data)

    def flush(self):
        pass

    def close(self):
        if self._fh is None:
            return
        if self._held:
            self._fh.write(member(bytes(self._held), self._level))
        self._fh.write(_EOF)
        self._fh.close()
        self._fh = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def is_bgzf(filename):
    """True if the file starts with a BGZF member header."""
    with open(filename, 'rb') as fh:
        head = fh.read(18)
    return len(head) == 18 and head[:4] == b'\x1f\x8b\x08\x04' and head[12:14] == b'BC'


def write_file(filename, data, level=6, threads=1):
    """Write `data` (bytes) as a BGZF file; with threads > 1 the members are compressed concurrently (zlib releases
    the interpreter lock)."""
    pieces = [data[i:i + BLOCK_TEXT] for i in range(0, len(data), BLOCK_TEXT)]
    with open(filename, 'wb') as fh:
        if threads > 1 and len(pieces) > 1:
            from concurrent.futures import ThreadPoolExecutor
            with ThreadPoolExecutor(max_workers=threads) as pool:
                for blob in pool.map(lambda piece: member(piece, level), pieces, chunksize=16):
                    fh.write(blob)
        else:
            for piece in pieces:
                fh.write(member(piece, level))
        fh.write(_EOF)
