"""CPU checks of the bit-level helpers behind the super-k-mer front end (kevlar_amd/csrc/kv_skm_device.h).

The header compiles for the host too; tests/harness/skm_host.cpp wraps it in a C ABI and this file compares it
with string-level restatements (reverse complement, canonical form, rolling, packing).  No GPU involved."""
import ctypes
import os
import random
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, 'tests', 'harness', 'skm_host.cpp')
HDR = os.path.join(ROOT, 'kevlar_amd', 'csrc', 'kv_skm_device.h')
HDR2 = os.path.join(ROOT, 'kevlar_amd', 'csrc', 'kv_fastmod.h')
SO = os.path.join(ROOT, 'tests', 'harness', 'libskm_host.so')
CODE = {'A': 0, 'C': 1, 'G': 2, 'T': 3}
COMP = {'A': 'T', 'C': 'G', 'G': 'C', 'T': 'A'}


@pytest.fixture(scope='module')
def lib():
    clang = '/opt/rocm/lib/llvm/bin/clang++'
    if not os.path.exists(clang):
        pytest.skip('clang++ of the ROCm toolchain not found')
    if not os.path.exists(SO) or os.path.getmtime(SO) < max(os.path.getmtime(SRC), os.path.getmtime(HDR), os.path.getmtime(HDR2)):
        subprocess.check_call([clang, '-x', 'c++', '-std=c++17', '-O1', '-ffp-contract=off', '-fPIC', '-shared', '-o', SO, SRC])
    L = ctypes.CDLL(SO)
    L.h_mmer_value.restype = ctypes.c_uint32
    L.h_bases32.restype = ctypes.c_uint64
    L.h_ascii4.restype = ctypes.c_uint32
    L.h_header.restype = ctypes.c_uint64
    L.h_header.argtypes = [ctypes.c_uint64, ctypes.c_uint32, ctypes.c_uint32]
    L.h_hdr_pos.restype = ctypes.c_uint64
    L.h_hdr_pos.argtypes = [ctypes.c_uint64]
    L.h_hdr_n.argtypes = [ctypes.c_uint64]
    L.h_hdr_fine.argtypes = [ctypes.c_uint64]
    return L


def pack(seq):
    """2 bits per base, base p in bits 2p.. (the kv_reads layout), as a Python int"""
    v = 0
    for p, ch in enumerate(seq):
        v |= CODE[ch] << (2 * p)
    return v


def revcomp(seq):
    return ''.join(COMP[c] for c in reversed(seq))


def words64(v, n):
    return (ctypes.c_uint64 * n)(*[(v >> (64 * i)) & (2**64 - 1) for i in range(n)])


def test_revcomp_matches_strings(lib):
    rng = random.Random(1)
    for k in list(range(16, 65)) * 3:
        seq = ''.join(rng.choice('ACGT') for _ in range(k))
        kw = 1 if k <= 32 else 2
        out = (ctypes.c_uint64 * 2)()
        lib.h_revcomp(kw, words64(pack(seq), 2), k, out)
        assert out[0] | (out[1] << 64) == pack(revcomp(seq)), (k, seq)


def test_rolling_gives_the_canonical_kmers_of_a_record(lib):
    rng = random.Random(2)
    for k in (16, 21, 31, 32, 33, 47, 51, 63, 64):
        kw = 1 if k <= 32 else 2
        nbw = kw + 1
        ncap = 32 * nbw - k + 1
        for n in (1, 2, ncap // 2, ncap):
            seq = ''.join(rng.choice('ACGT') for _ in range(n + k - 1))
            out = (ctypes.c_uint64 * (2 * n))()
            lib.h_roll(kw, words64(pack(seq), 3), k, n, out)
            for j in range(n):
                kmer = seq[j:j + k]
                want = min(pack(kmer), pack(revcomp(kmer)))
                assert out[2 * j] | (out[2 * j + 1] << 64) == want, (k, n, j)


def test_canonical_form_is_strand_symmetric(lib):
    rng = random.Random(3)
    for k in (31, 51):
        kw = 1 if k <= 32 else 2
        seq = ''.join(rng.choice('ACGT') for _ in range(k))
        a = (ctypes.c_uint64 * 2)()
        b = (ctypes.c_uint64 * 2)()
        lib.h_roll(kw, words64(pack(seq), 3), k, 1, a)
        lib.h_roll(kw, words64(pack(revcomp(seq)), 3), k, 1, b)
        assert list(a) == list(b)


def test_mmer_value_is_strand_symmetric_and_spreads(lib):
    rng = random.Random(4)
    seen = set()
    for m in (8, 10, 12, 16):
        for _ in range(200):
            seq = ''.join(rng.choice('ACGT') for _ in range(m))
            v, vr = lib.h_mmer_value(pack(seq), m), lib.h_mmer_value(pack(revcomp(seq)), m)
            # the order of the canonical m-mer is the same on both strands; bit 0 says on which strand the m-mer is NOT canonical:
            # opposite on the two strands (unless the m-mer is its own reverse complement: then neither is "reversed")
            assert v >> 1 == vr >> 1
            assert (v & 1) + (vr & 1) == (0 if seq == revcomp(seq) else 1)
            assert (v & 1) == (1 if pack(revcomp(seq)) < pack(seq) else 0)
            seen.add(v >> 1)
    assert len(seen) > 700


def test_oriented_record_helpers(lib):
    """reverse complement of the bases of a record (what S1 stores for a run whose minimizer stands reversed), the header's
    orientation flag with the position arithmetic that goes with it, and the fields of a compact record"""
    rng = random.Random(11)
    lib.h_rc_bases.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_uint32]
    for nbw in (2, 3):
        for nb in list(range(1, 9)) + [31, 32, 33, 50, 63, 64] + ([65, 80, 95, 96] if nbw == 3 else []):
            if nb > 32 * nbw:
                continue
            seq = ''.join(rng.choice('ACGT') for _ in range(nb))
            junk = ''.join(rng.choice('ACGT') for _ in range(32 * nbw - nb))      # the read's next bases: what skm_bases32 brings along
            v = pack(seq + junk)
            bw = (ctypes.c_uint64 * 3)(*[(v >> (64 * i)) & (2**64 - 1) for i in range(3)])
            lib.h_rc_bases(bw, nbw, nb)
            got = sum(int(bw[i]) << (64 * i) for i in range(nbw))
            assert got == pack(revcomp(seq)), (nbw, nb)
    lib.h_header_rev.restype = ctypes.c_uint64
    lib.h_header_rev.argtypes = [ctypes.c_uint64, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32]
    lib.h_hdr_pos_of.restype = ctypes.c_uint64
    lib.h_hdr_pos_of.argtypes = [ctypes.c_uint64, ctypes.c_uint32]
    lib.h_hdr_rev.argtypes = [ctypes.c_uint64]
    for rev in (0, 1):
        h = lib.h_header_rev(123456789, 34, 4095, rev)
        assert (lib.h_hdr_pos(h), lib.h_hdr_n(h), lib.h_hdr_fine(h), lib.h_hdr_rev(h)) == (123456789, 34, 4095, rev)
        assert [lib.h_hdr_pos_of(h, j) for j in (0, 1, 33)] == ([123456789 + 33, 123456789 + 32, 123456789] if rev else [123456789, 123456790, 123456789 + 33])
    lib.h_c_pack1.restype = ctypes.c_uint64
    lib.h_c_pack1.argtypes = [ctypes.c_uint64, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32]
    lib.h_c_b1.restype = ctypes.c_uint64
    for f in (lib.h_c_n, lib.h_c_fine, lib.h_c_rev, lib.h_c_b1):
        f.argtypes = [ctypes.c_uint64]
    w1 = lib.h_c_pack1(2**40 - 3, 22, 4095, 1)
    assert (lib.h_c_b1(w1), lib.h_c_n(w1), lib.h_c_fine(w1), lib.h_c_rev(w1)) == (2**40 - 3, 22, 4095, 1)
    w1 = lib.h_c_pack1(2**64 - 1, 1, 0, 0)             # (bases beyond the 52nd are cut off)
    assert (lib.h_c_b1(w1), lib.h_c_n(w1), lib.h_c_fine(w1), lib.h_c_rev(w1)) == (2**40 - 1, 1, 0, 0)


def test_bases32_and_ascii4(lib):
    rng = random.Random(5)
    seq = ''.join(rng.choice('ACGT') for _ in range(200))
    v = pack(seq)
    words = (ctypes.c_uint32 * 16)(*[(v >> (32 * i)) & 0xffffffff for i in range(16)])
    for b in (0, 1, 15, 16, 17, 31, 33, 100):
        assert lib.h_bases32(words, b) == pack(seq[b:b + 32])
    for byte in range(256):
        s = ''.join('ACGT'[(byte >> (2 * i)) & 3] for i in range(4))
        assert lib.h_ascii4(byte) == int.from_bytes(s.encode(), 'little')


def test_header_fields_and_bucket_ranges(lib):
    h = lib.h_header(2**40 - 7, 46, 511)
    assert (lib.h_hdr_pos(h), lib.h_hdr_n(h), lib.h_hdr_fine(h)) == (2**40 - 7, 46, 511)
    rng = random.Random(6)
    c, f = ctypes.c_uint32(), ctypes.c_uint32()
    counts = {}
    for _ in range(20000):
        # minimizer values are minima of ~20 uniform draws: concentrated near zero
        minv = min(rng.getrandbits(32) for _ in range(20))
        lib.h_bucket_of(minv, 251, 8, ctypes.byref(c), ctypes.byref(f))
        assert c.value < 251 and f.value < 256
        counts[c.value] = counts.get(c.value, 0) + 1
    assert len(counts) == 251 and max(counts.values()) < 4 * 20000 / 251
    lib.h_bucket_of(12345, 1, 0, ctypes.byref(c), ctypes.byref(f))
    assert (c.value, f.value) == (0, 0)


def test_fastmod_is_the_remainder(lib):
    """h % size for sizes on both sides of the FP64 / Barrett switch, at the inputs where a quotient estimate is most
    likely to be off: multiples of the size and their neighbours, the top of the 64-bit range, and random hashes"""
    import numpy as np
    lib.h_fastmod.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_void_p]
    rng = np.random.default_rng(11)
    sizes = [1, 2, 3, 97, 65521, 65535, 65536, 65537, 499999999, 499999993, 499999931, 499999909, 124999987, 1999999973,
             2**31 - 1, 2**32 - 5, 2**32 - 1, 2**32, 2**32 + 15, 7999999967, 2**40 + 15, 2**63 + 9]
    sizes += [int(x) for x in rng.integers(65536, 2**32, 40, dtype=np.uint64)]
    for size in sizes:
        q = (2**64 - 1) // size
        edge = []
        for mult in [0, 1, 2, q // 3, q // 2, q - 1, q] + [int(x) for x in rng.integers(0, q + 1, 2000, dtype=np.uint64)]:
            for d in (-2, -1, 0, 1, 2):
                v = mult * size + d
                if 0 <= v < 2**64:
                    edge.append(v)
        edge += [2**64 - 1, 2**64 - 2, 2**63, 2**63 - 1, 2**53, 2**53 + 1, 2**52 - 1]
        h = np.concatenate([np.array(edge, dtype=np.uint64), rng.integers(0, 2**64, 200000, dtype=np.uint64)])
        out = np.empty_like(h)
        lib.h_fastmod(h.ctypes.data, len(h), size, out.ctypes.data)
        assert np.array_equal(out, h % np.uint64(size)), size
