"""Abundance lookups behind `kevlar simlike` (call sites kevlar/simlike.py:51-97,349-363): for
the window spanning a candidate variant, how often each of its alternate-allele k-mers was seen in the case, in
every control and in the reference genome.  Each sketch answers for the whole window in one batched device query
(khmer's get_kmer_counts = kv_hash_kmers + kv_get_hashes here); the bookkeeping is numpy masks over those vectors.
The likelihood model and the VCF plumbing that consume the numbers are out of scope."""
import numpy as np


def _window_counts(sketch, sequence):
    return np.asarray(sketch.get_kmer_counts(sequence), dtype=np.int64)


def _near_mean(values, tolerance=20):
    """mask of the entries within `tolerance` of the mean of the vector"""
    return np.abs(values - values.mean()) < tolerance if len(values) else np.zeros(0, dtype=bool)


def discard_nonunique_kmers(altseq, case, controls, refr):
    """Counts of the alternate-allele k-mers that do not occur in the reference genome (those that do carry no
    signal): (case counts, [control counts ...], reference counts of ALL window k-mers)."""
    in_genome = _window_counts(refr, altseq)
    novel = in_genome == 0
    keep = lambda sketch: _window_counts(sketch, altseq)[novel].tolist()   # noqa: E731
    return keep(case), [keep(control) for control in controls], in_genome.tolist()


def discard_outlier_abunds(case_counts, ctrl_counts):
    """Drop, per sample, the counts 20 or more away from that sample's mean."""
    trim = lambda counts: np.asarray(counts, dtype=np.int64)[_near_mean(np.asarray(counts, dtype=np.float64))].tolist()   # noqa: E731
    return trim(case_counts), [trim(counts) for counts in ctrl_counts]


def spanning_kmer_abundances(altseq, refrseq, case, controls, refr, dropoutliers=False):
    """(abundances, refr_abunds, ndropped) for the k-mers spanning a variant: abundances[0] the case counts and
    abundances[1:] the control counts of the alternate-allele k-mers absent from the reference genome; refr_abunds
    the genomic frequency of the matching reference-allele k-mers (substitutions: same window length) or None per
    k-mer (indels); ndropped how many k-mers of the window were discarded."""
    nwindow = len(altseq) - case.ksize() + 1
    in_genome = _window_counts(refr, altseq)
    novel = in_genome == 0
    per_sample = [_window_counts(sketch, altseq)[novel] for sketch in [case] + list(controls)]
    if dropoutliers:
        per_sample = [counts[_near_mean(counts.astype(np.float64))] for counts in per_sample]
    kept = len(per_sample[0])
    if len(altseq) == len(refrseq):
        refr_abunds = _window_counts(refr, refrseq)[novel].tolist()
    else:
        refr_abunds = [None] * kept
    return [counts.tolist() for counts in per_sample], refr_abunds, nwindow - kept
