"""Overlap validation for `kevlar partition --strict` (semantics of the reference's
kevlar/readpair.py:15-178, written as plain functions).

Two reads that share an interesting k-mer are *compatible* when, anchored on that k-mer, one
read's suffix matches the other's prefix exactly (or one contains the other).  Host-side Python:
strict mode is an optional, small-input refinement; the default
relaxed graph runs on the GPU (kv_readgraph_components).
"""
from collections import namedtuple

from kevlar_amd.sequence import revcom

# one read in one orientation, anchored on the shared k-mer
Anchored = namedtuple('Anchored', 'name sequence offset ksize kmerseq')


def _anchor(record, kmerseq):
    """The record with the offset of the shared k-mer (looked up in either orientation), or
    None if the k-mer does not occur exactly once in the read (readpair.py:19-23,174-175)."""
    ikmer = record.ikmers.get(kmerseq)
    if ikmer is None:
        return None
    occurrences = record.sequence.count(kmerseq) + record.sequence.count(revcom(kmerseq))
    if occurrences != 1:
        return None
    return Anchored(record.name, record.sequence, ikmer.offset, ikmer.ksize, record.ikmerseq(ikmer))


def _flip(a):
    seq = revcom(a.sequence)
    return Anchored(a.name, seq, len(seq) - a.offset - a.ksize, a.ksize, revcom(a.kmerseq))


def _head_and_tail(r1, r2):
    """The tail sits to the left and keeps its orientation (readpair.py:79-145): larger k-mer
    offset over both possible arrangements, then longer read, then smaller name."""
    same = r1.kmerseq == r2.kmerseq
    r1rc, r2rc = _flip(r1), _flip(r2)
    arrangements = [(r1, r2), (r1rc, r2rc)] if same else [(r1, r2rc), (r1rc, r2)]
    best = [max(a.offset for a in arr) for arr in arrangements]
    if best[0] != best[1]:
        arr = arrangements[0] if best[0] > best[1] else arrangements[1]
        tail = arr[0] if arr[0].offset >= arr[1].offset else arr[1]      # max(): first wins a tie
        head = arr[0] if arr[0].offset <= arr[1].offset else arr[1]      # min(): first wins a tie
        return tail, head
    other2 = r2 if same else r2rc
    if len(r1.sequence) != len(r2.sequence):
        return (r1, other2) if len(r1.sequence) > len(r2.sequence) else (other2, r1)
    return (r1, other2) if r1.name < r2.name else (other2, r1)


def validate(record1, record2, kmerseq):
    """(merged sequence, tail name, head name) of a compatible pair, or None (readpair.py:147-178).

    Quirk kept from the reference: when both arrangements tie on the larger k-mer offset inside the
    chosen arrangement, max() and min() return the SAME read as tail and head; the pair then counts as
    compatible but the edge it produces is a self-loop (readgraph.py:86-102 adds tail--head)."""
    r1, r2 = _anchor(record1, kmerseq), _anchor(record2, kmerseq)
    if r1 is None or r2 is None:
        return None
    tail, head = _head_and_tail(r1, r2)
    if tail.offset < head.offset:
        tail, head = head, tail
    offset = tail.offset - head.offset
    overlap = len(tail.sequence) - offset
    tailseq, headseq = tail.sequence, head.sequence
    if headseq in tailseq or tailseq in headseq:
        return tailseq, tail.name, head.name
    headindex = len(tailseq) - offset
    if tailseq[offset:offset + overlap] == headseq[:headindex]:
        return tailseq + headseq[headindex:], tail.name, head.name
    return None


def merged_sequence(record1, record2, kmerseq):
    result = validate(record1, record2, kmerseq)
    return None if result is None else result[0]


def compatible(record1, record2, kmerseq):
    return validate(record1, record2, kmerseq) is not None
