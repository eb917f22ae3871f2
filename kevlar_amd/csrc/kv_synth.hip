// kv_synth.hip -- the bench's synthetic family written straight into HBM ("synthetic inputs"; BASELINE.json
// config 4: a 3 Gb genome at 30x is 900 M reads per sample, 90 GB of text -- far beyond what a host generator feeds in
// benchmark time).  No reference counterpart: kevlar's gentrio (kevlar/gentrio.py:185-257) writes haplotype FASTA and its
// test reads came from wgsim.  The rule is the one kevlar_amd/synth.py applies at 25 Mb, restated so that everything is a
// pure function of (seed, position) or (seed, read index) -- no genome, haplotype or variant list is ever stored:
//
//   reference base at p            = mix(seed, p) & 3                                    (iid uniform)
//   inherited variant at p         : one position in 2500 (400 per Mb), carried by ONE of the four parental haplotypes
//   de novo variant at p           : one position in 5000 (200 per Mb), on one of the proband's two haplotypes
//   father = haplotypes 0, 1; mother = 2, 3; proband = father's 0 + mother's 2 (no recombination) + its de novo variants
//   a variant substitutes the base (gentrio's 10 % insertions and 10 % deletions are SNVs here: a read generator that
//   never builds a haplotype cannot shift coordinates; the k-mer statistics the count and the scan see are the same)
//   read i of a sample              : haplotype, strand, start from mix(sample seed, i); base errors 0.5 % (substitutions)
//
// kevlar_amd/synth.py device_family_reads() is the numpy restatement; tests/test_gpu_synth.py compares the two bit for bit.
#include "kv_internal.h"

namespace {

__host__ __device__ inline uint64_t syn_mix(uint64_t x)
{
    x += 0x9e3779b97f4a7c15ull;
    x = (x ^ (x >> 30)) * 0xbf58476d1ce4e5b9ull;
    x = (x ^ (x >> 27)) * 0x94d049bb133111ebull;
    return x ^ (x >> 31);
}

#define SYN_INH_ONE_IN 2500ull
#define SYN_DN_ONE_IN 5000ull

// base at position p of haplotype `hap` of the family (0..3 parental; 4, 5 the proband's two)
__device__ __forceinline__ uint32_t syn_base(uint64_t seed, uint64_t p, uint32_t hap)
{
    uint32_t b = (uint32_t)syn_mix(seed ^ (p * 0x2545f4914f6cdd1dull)) & 3u;
    const uint32_t parental = hap < 4u ? hap : (hap == 4u ? 0u : 2u);
    const uint64_t hi = syn_mix((seed + 1) ^ (p * 0x9fb21c651e98df25ull));
    if (hi % SYN_INH_ONE_IN == 0 && ((hi / SYN_INH_ONE_IN) & 3ull) == parental) b = (b + 1u + (uint32_t)((hi >> 40) % 3ull)) & 3u;
    if (hap >= 4u) {
        const uint64_t hd = syn_mix((seed + 2) ^ (p * 0xd6e8feb86659fd93ull));
        if (hd % SYN_DN_ONE_IN == 0 && ((hd / SYN_DN_ONE_IN) & 1ull) == (uint64_t)(hap - 4u)) b = (b + 1u + (uint32_t)((hd >> 40) % 3ull)) & 3u;
    }
    return b;
}

// one thread per packed word (16 bases)
__global__ __launch_bounds__(256) void k_synth_reads(uint32_t *words, uint64_t genome_len, uint64_t seed, int sample, uint64_t first_read,
                                                     uint64_t n_reads, uint32_t read_len, uint32_t wpr, uint32_t err_threshold)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    const uint64_t sseed = seed + 16 + 8 * (uint64_t)sample;
    for (uint64_t wi = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; wi < n_reads * wpr; wi += stride) {
        const uint64_t r = wi / wpr, i = first_read + r;
        const uint32_t j0 = (uint32_t)(wi - r * wpr) * 16u;
        const uint64_t rr = syn_mix(sseed ^ (i * 0xa0761d6478bd642full));
        const uint32_t which = (uint32_t)rr & 1u, flip = (uint32_t)(rr >> 1) & 1u;
        const uint64_t start = (rr >> 8) % (genome_len - read_len + 1);
        const uint32_t hap = sample == 0 ? 4u + which : (sample == 2 ? which : 2u + which);     // 0 proband, 1 mother, 2 father
        uint32_t word = 0;
        for (uint32_t jj = 0; jj < 16u && j0 + jj < read_len; ++jj) {
            const uint32_t j = j0 + jj;
            uint32_t b = flip ? 3u - syn_base(seed, start + (read_len - 1u - j), hap) : syn_base(seed, start + j, hap);
            const uint64_t e = syn_mix((sseed + 1) ^ ((i * 4096ull + j) * 0xe7037ed1a0b428dbull));
            if ((uint32_t)e < err_threshold) b = (b + 1u + (uint32_t)((e >> 32) % 3ull)) & 3u;
            word |= b << (2u * jj);
        }
        words[wi] = word;
    }
}

}  // namespace

void kv_synth_fill(uint32_t *d_words, uint64_t genome_len, uint64_t seed, int sample, uint64_t first_read, uint64_t n_reads,
                   uint32_t read_len, double error_rate, hipStream_t st)
{
    const uint32_t wpr = (read_len + 15u) / 16u;
    const uint32_t thr = (uint32_t)(error_rate * 4294967296.0);
    const uint64_t nw = n_reads * wpr;
    const unsigned grid = (unsigned)std::min<uint64_t>((nw + 255) / 256, 65536);
    if (nw) hipLaunchKernelGGL(k_synth_reads, dim3(grid), dim3(256), 0, st, d_words, genome_len, seed, sample, first_read, n_reads, read_len, wpr, thr);
}
