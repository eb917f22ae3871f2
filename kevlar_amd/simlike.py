"""The k-mer abundance lookups of `kevlar simlike` (the reference's kevlar/simlike.py:24-97): for the
window spanning a candidate variant, the counts of its alternate-allele k-mers in the case, control and
reference sketches -- batched `get` calls on the device (kv_hash_kmers + kv_get_hashes), SURVEY.md 8(f).4.
The likelihood model and the VCF plumbing that consume these numbers are outside this build (DESIGN.md
section 7)."""


def discard_nonunique_kmers(altseq, case, controls, refr):
    """Alternate-allele k-mers that also occur in the reference genome carry no signal: drop them."""
    case_counts = case.get_kmer_counts(altseq)
    alt_counts_refr = refr.get_kmer_counts(altseq)
    case_counts_valid = [c for c, r in zip(case_counts, alt_counts_refr) if r == 0]
    ctrl_counts_valid = list()
    for control in controls:
        ctrl_counts = control.get_kmer_counts(altseq)
        ctrl_counts_valid.append([c for c, r in zip(ctrl_counts, alt_counts_refr) if r == 0])
    return case_counts_valid, ctrl_counts_valid, alt_counts_refr


def discard_outlier_abunds(case_counts, ctrl_counts):
    meanabund = sum(case_counts) / len(case_counts)
    case_counts_valid = [a for a in case_counts if abs(a - meanabund) < 20]
    ctrl_counts_valid = list()
    for control in ctrl_counts:
        meanabund = sum(control) / len(control)
        ctrl_counts_valid.append([a for a in control if abs(a - meanabund) < 20])
    return case_counts_valid, ctrl_counts_valid


def spanning_kmer_abundances(altseq, refrseq, case, controls, refr, dropoutliers=False):
    """Aggregate the abundances of the k-mers spanning the variant.

    Returns (abundances, refr_abunds, ndropped): abundances[0] are the case counts and abundances[1:]
    the control counts of the alternate-allele k-mers absent from the reference genome; refr_abunds the
    genomic frequency of the corresponding reference-allele k-mers for SNVs/MNVs, None per k-mer for
    indels; ndropped the number of k-mers discarded."""
    orig_nkmers = len(altseq) - case.ksize() + 1
    case_counts, ctrl_counts, alt_counts_refr = discard_nonunique_kmers(altseq, case, controls, refr)
    if dropoutliers:
        case_counts, ctrl_counts = discard_outlier_abunds(case_counts, ctrl_counts)
    ndropped = orig_nkmers - len(case_counts)
    abundances = [case_counts] + ctrl_counts
    if len(altseq) == len(refrseq):  # SNV or MNV
        refr_counts = refr.get_kmer_counts(refrseq)
        refr_abunds = [c for c, r in zip(refr_counts, alt_counts_refr) if r == 0]
    else:  # INDEL
        refr_abunds = [None] * len(case_counts)
    return abundances, refr_abunds, ndropped
