// Host-side harness for kevlar_amd/csrc/kv_skm_device.h (compiled as plain C++ by tests/test_skm_host.py):
// exposes the bit-level helpers of the super-k-mer front end through a C ABI so that Python can compare them
// with string-level restatements.
#include <stdint.h>
#include "../../kevlar_amd/csrc/kv_skm_device.h"
#include "../../kevlar_amd/csrc/kv_fastmod.h"

template <int KW>
static void t_revcomp(const uint64_t *in, int k, uint64_t *out)
{
    SkmKey<KW> f;
    f.w[0] = in[0];
    if (KW == 2) f.w[KW - 1] = in[1];
    const SkmKey<KW> r = skm_revcomp<KW>(f, k);
    out[0] = r.w[0];
    out[1] = KW == 2 ? r.w[KW - 1] : 0;
}

// canonical keys of the n k-mers of a record (base words bw), by rolling: out[2 * j], out[2 * j + 1]
template <int KW>
static void t_roll(const uint64_t *bw, int k, int n, uint64_t *out)
{
    SkmKey<KW> fw = skm_first_kmer<KW>(bw, k);
    SkmKey<KW> rc = skm_revcomp<KW>(fw, k);
    for (int j = 0; j < n; ++j) {
        if (j) skm_roll<KW>(fw, rc, skm_base_at(bw, (uint32_t)(j + k - 1)), k);
        const SkmKey<KW> c = skm_canonical<KW>(fw, rc);
        out[2 * j] = c.w[0];
        out[2 * j + 1] = KW == 2 ? c.w[KW - 1] : 0;
    }
}

extern "C" {
void h_revcomp(int kw, const uint64_t *in, int k, uint64_t *out) { if (kw == 1) t_revcomp<1>(in, k, out); else t_revcomp<2>(in, k, out); }
void h_roll(int kw, const uint64_t *bw, int k, int n, uint64_t *out) { if (kw == 1) t_roll<1>(bw, k, n, out); else t_roll<2>(bw, k, n, out); }
uint32_t h_mmer_value(uint32_t f, int m) { return skm_mmer_value(f, m); }
uint64_t h_bases32(const uint32_t *words, uint32_t b) { return skm_bases32(words, b); }
uint32_t h_ascii4(uint32_t byte) { return skm_ascii4(byte); }
void h_bucket_of(uint32_t minv, uint32_t C1, uint32_t fbits, uint32_t *coarse, uint32_t *fine) { skm_bucket_of(minv, C1, fbits, *coarse, *fine); }
uint64_t h_header(uint64_t pos, uint32_t n, uint32_t fine) { return skm_header(pos, n, fine); }
uint64_t h_hdr_pos(uint64_t h) { return skm_hdr_pos(h); }
uint32_t h_hdr_n(uint64_t h) { return skm_hdr_n(h); }
uint32_t h_hdr_fine(uint64_t h) { return skm_hdr_fine(h); }
uint64_t h_header_rev(uint64_t pos, uint32_t n, uint32_t fine, uint32_t rev) { return skm_header(pos, n, fine, rev); }
uint32_t h_hdr_rev(uint64_t h) { return skm_hdr_rev(h); }
uint64_t h_hdr_pos_of(uint64_t h, uint32_t j) { return skm_hdr_pos_of(h, j); }
void h_rc_bases(uint64_t *bw, int nbw, uint32_t nb) { skm_rc_bases(bw, nbw, nb); }
uint64_t h_c_pack1(uint64_t b1, uint32_t n, uint32_t fine, uint32_t rev) { return skm_c_pack1(b1, n, fine, rev); }
uint32_t h_c_n(uint64_t w1) { return skm_c_n(w1); }
uint32_t h_c_fine(uint64_t w1) { return skm_c_fine(w1); }
uint32_t h_c_rev(uint64_t w1) { return skm_c_rev(w1); }
uint64_t h_c_b1(uint64_t w1) { return skm_c_b1(w1); }
uint64_t h_fastmod_magic(uint64_t size) { return kv_fastmod_magic(size); }
// out[i] = fastmod(h[i], size): the device's remainder arithmetic (FP64 quotient or Barrett, by the size), on the host's IEEE doubles
void h_fastmod(const uint64_t *h, uint64_t n, uint64_t size, uint64_t *out)
{
    const uint64_t magic = kv_fastmod_magic(size);
    for (uint64_t i = 0; i < n; ++i) out[i] = fastmod(h[i], size, magic);
}
}
