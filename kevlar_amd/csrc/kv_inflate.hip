// kv_inflate.hip -- DEFLATE on the device for blocked gzip (BGZF) input (ingest).
//
// A plain .gz file is one DEFLATE stream: nothing can start before everything in front of it has been decoded, so it
// inflates at the speed of one host core (~250 MB/s with zlib: ~1 M reads/s).  BGZF -- what bgzip, samtools and
// htslib write, and what kevlar_amd.open(..., 'w') writes for *.gz -- is a series of independent gzip members of at
// most 64 KB of text each, every member announcing its compressed size in an extra header field.  The host only hops
// from header to header (kv_bgzf_index); the compressed bytes go to HBM as they are (a quarter of the text crosses
// PCIe), and one wavefront inflates one member:
//
//   * the last 1 KB of the member's text stays in LDS, so the near LZ77 matches (the previous record's header, the '+' line)
//     are LDS copies; the ones that reach further back read the text the wave itself wrote to HBM;
//   * the Huffman codes are walked one after the other (a serial job by nature) through 10-bit / 8-bit lookup tables in
//     LDS, longer codes through the canonical count/symbol arrays; the walk is wave-uniform, so its arithmetic runs on the
//     scalar unit; literals collect in a vector register, one per lane, and reach LDS 64 at a time;
//   * a match is copied by all 64 lanes at once (an overlapping match is periodic in its distance, so every byte has a
//     source in front of the match start);
//   * text leaves for HBM in rows of up to 64 bytes as it is produced.
//
// Decoding one member is a chain of dependent lookups (a few hundred cycles per symbol whatever the code does), so the rate
// comes from members in flight: 32 workgroups per CU (8 waves per SIMD), ~8000 members on the device.  ISIZE and CRC-32 of every
// member are checked (k_gz_crc of kv_gunzip.hip over the text); tests compare the output with zlib's byte for byte.  Reference: the reader this replaces is
// khmer.ReadParser's gzip stream (kevlar/__init__.py:125-128 opens every *.gz through it).
#include <algorithm>
#include <cstring>
#include <vector>

#include "kv_binned.h"
#include "kv_internal.h"
#include "kv_inflate_device.h"

namespace {

struct InflateJob {
    uint64_t in_off;        // deflate payload of the member, relative to the compressed buffer
    uint32_t in_len;
    uint32_t isize;         // bytes the member inflates to
    uint64_t out_off;       // where they go in the text buffer
};

template <int RBITS>
struct InflateShared {
    uint8_t ring[1 << RBITS];      // the last 2^RBITS bytes of text (byte t of the member at t mod 2^RBITS)
    uint16_t ll_table[1 << INF_FAST_LL];
    uint16_t d_table[1 << INF_FAST_D];
    uint16_t ll_count[16], d_count[16];
    uint16_t ll_symbol[288], d_symbol[32];
    uint8_t lengths[352];          // 19 code-length codes, then up to 286 + 30 code lengths
};

template <int RBITS>
__global__ __launch_bounds__(64, (RBITS <= 10 ? 8 : RBITS == 11 ? 5 : RBITS == 12 ? 4 : RBITS == 13 ? 3 : RBITS == 14 ? 2 : 1)) void k_inflate(const uint8_t *__restrict__ comp, const InflateJob *__restrict__ jobs, uint32_t n_jobs,
                                                 uint8_t *text, unsigned long long *ctr)
{
    __shared__ InflateShared<RBITS> sh;
    constexpr uint32_t RMASK = (1u << RBITS) - 1u;
    const uint32_t lane = threadIdx.x;
    for (;;) {
        uint32_t job_id = 0;
        if (lane == 0) job_id = (uint32_t)atomicAdd(&ctr[0], 1ull);
        job_id = INF_UNI(job_id);
        if (job_id >= n_jobs) return;
        const InflateJob job = jobs[job_id];
        if (job.isize == 0) continue;                 // BGZF's end-of-file marker, or an empty member
        BitReader br;                                 // identical in every lane
        br_init(br, comp + job.in_off);
        const uint32_t limit_words = (job.in_len + 3u) / 4u + 4u;     // a reader further than this has left the payload (damaged data)
        uint32_t o = 0;                               // bytes of text produced
        uint32_t lit = 0, n_lit = 0;                  // literals not yet stored: lane i holds the i-th, n_lit of them
        bool failed = job.isize > INF_MAX_OUT;
        bool last_block = false;
        uint8_t *dst = text + job.out_off;
        // pending literals -> LDS window and HBM, one row store each for up to 64 of them
        auto flush = [&]() {
            if (lane < n_lit) { sh.ring[(o + lane) & RMASK] = (uint8_t)lit; dst[o + lane] = (uint8_t)lit; }
            o += n_lit;
            n_lit = 0;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        };
        while (!failed && !last_block) {
            // ---- block header and code tables
            last_block = br_bits(br, 1) != 0;
            const uint32_t btype = br_bits(br, 2);
            if (btype == 0) {
                const uint32_t drop = br.cnt & 7u;        // to the next byte boundary
                br.buf >>= drop; br.cnt -= drop;
                const uint32_t stored_len = br_bits(br, 16);
                const uint32_t inv = br_bits(br, 16);
                if ((stored_len ^ inv) != 0xffffu || o + stored_len > job.isize) { failed = true; break; }
                // (a stored block that claims more bytes than the member holds -- damaged or crafted input -- must not walk
                // the reader out of the buffer: the position is checked once per row of 64 bytes, and the caller's buffer
                // has KV_INFLATE_SLACK readable bytes behind the last member)
                for (uint32_t j = 0; j < stored_len; ++j) {
                    const uint32_t byte = br_bits(br, 8);
                    lit = lane == n_lit ? byte : lit;
                    if (++n_lit == 64) {
                        if (br.next > limit_words) { failed = true; break; }
                        flush();
                    }
                }
                if (failed || br.next > limit_words) { failed = true; break; }
                flush();
                continue;
            }
            if (btype == 3) { failed = true; break; }
            uint32_t ok = 1;
            if (btype == 1) {
                if (lane == 0) {
                    for (int s = 0; s < 144; ++s) sh.lengths[s] = 8;
                    for (int s = 144; s < 256; ++s) sh.lengths[s] = 9;
                    for (int s = 256; s < 280; ++s) sh.lengths[s] = 7;
                    for (int s = 280; s < 288; ++s) sh.lengths[s] = 8;
                    ok = build_code(sh.lengths, 288, sh.ll_count, sh.ll_symbol, sh.ll_table, INF_FAST_LL);
                    for (int s = 0; s < 30; ++s) sh.lengths[s] = 5;
                    ok = ok && build_code(sh.lengths, 30, sh.d_count, sh.d_symbol, sh.d_table, INF_FAST_D);
                }
            } else {
                const uint32_t nlen = br_bits(br, 5) + 257, ndist = br_bits(br, 5) + 1, ncode = br_bits(br, 4) + 4;
                if (nlen > 286 || ndist > 30) { failed = true; break; }
                if (lane < 19) sh.lengths[lane] = 0;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                for (uint32_t s = 0; s < ncode; ++s) {
                    const uint32_t v = br_bits(br, 3);
                    if (lane == 0) sh.lengths[inf_clen_order(s)] = (uint8_t)v;
                }
                // the code-length code lives in the distance arrays for a moment
                if (lane == 0) ok = build_code(sh.lengths, 19, sh.d_count, sh.d_symbol, sh.d_table, 7);
                ok = INF_UNI(ok);
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                __builtin_amdgcn_wave_barrier();
                uint32_t idx = 0, prev = 0;
                while (ok && idx < nlen + ndist) {
                    const int sym = decode_sym(br, sh.d_count, sh.d_symbol, sh.d_table, 7);
                    if (sym < 0) { ok = 0; break; }
                    uint32_t rep = 1, val = (uint32_t)sym;
                    if (sym == 16) {
                        if (idx == 0) { ok = 0; break; }
                        val = prev;
                        rep = 3 + br_bits(br, 2);
                    } else if (sym == 17) { val = 0; rep = 3 + br_bits(br, 3); }
                    else if (sym == 18) { val = 0; rep = 11 + br_bits(br, 7); }
                    if (idx + rep > nlen + ndist) { ok = 0; break; }
                    if (lane < rep) sh.lengths[19 + idx + lane] = (uint8_t)val;       // rep <= 138: at most three rows
                    if (lane + 64 < rep) sh.lengths[19 + idx + lane + 64] = (uint8_t)val;
                    if (lane + 128 < rep) sh.lengths[19 + idx + lane + 128] = (uint8_t)val;
                    idx += rep;
                    prev = val;
                }
                if (br.next > limit_words) ok = 0;      // the header alone can be ~560 bytes: a truncated member ends here
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                __builtin_amdgcn_wave_barrier();
                if (lane == 0 && ok) {
                    ok = sh.lengths[19 + 256] != 0;     // a block without an end code cannot end
                    ok = ok && build_code(sh.lengths + 19, (int)nlen, sh.ll_count, sh.ll_symbol, sh.ll_table, INF_FAST_LL);
                    ok = ok && build_code(sh.lengths + 19 + nlen, (int)ndist, sh.d_count, sh.d_symbol, sh.d_table, INF_FAST_D);
                }
            }
            ok = INF_UNI(ok);
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            if (!ok) { failed = true; break; }
            // ---- symbols
            // (the size and the end of the payload are checked when text is stored, not per literal: the loop is bound by the
            // number of instructions it issues)
            for (;;) {
                const int sym = decode_sym(br, sh.ll_count, sh.ll_symbol, sh.ll_table, INF_FAST_LL);
                if ((uint32_t)sym < 256u) {
                    lit = lane == n_lit ? (uint32_t)sym : lit;
                    if (++n_lit == 64) {
                        if (o + 64 > job.isize || br.next > limit_words) { failed = true; break; }
                        flush();
                    }
                    continue;
                }
                if (sym < 0 || o + n_lit > job.isize || br.next > limit_words) { failed = true; break; }
                flush();
                if (sym == 256) break;
                const int ls = sym - 257;
                if (ls >= 29) { failed = true; break; }
                const uint32_t len = inf_len_base((uint32_t)ls) + br_bits(br, inf_len_extra((uint32_t)ls));
                const int ds = decode_sym(br, sh.d_count, sh.d_symbol, sh.d_table, INF_FAST_D);
                if (ds < 0 || ds >= 30) { failed = true; break; }
                const uint32_t extra = inf_dist_extra((uint32_t)ds);
                br_need(br, 16);
                const uint32_t dist = inf_dist_base((uint32_t)ds) + ((uint32_t)br.buf & ((1u << extra) - 1u));
                br.buf >>= extra; br.cnt -= extra;
                if (dist > o || o + len > job.isize) { failed = true; break; }
                // all lanes copy; an overlapping match repeats its first `dist` bytes.  The source comes from the LDS window
                // unless it lies further back than the window reaches (or where this very copy is about to write): then
                // from the text in HBM, which this wave wrote itself -- fence, then loads that bypass the CU's L1
                const uint32_t from = o - dist;
                if (dist + 258u <= (1u << RBITS)) {
                    for (uint32_t j = lane; j < len; j += 64) {
                        const uint32_t src = dist >= len ? from + j : from + j % dist;
                        const uint8_t v = sh.ring[src & RMASK];
                        sh.ring[(o + j) & RMASK] = v;
                        dst[o + j] = v;
                    }
                } else {
                    // (workgroup scope: the bytes were stored by this very wave, through this CU's L1 into this XCD's L2;
                    // an agent-scope release would write the whole L2 back on a multi-XCD device)
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                    for (uint32_t j = lane; j < len; j += 64) {                        // dist > 258 >= len here: no overlap
                        const uint8_t v = __hip_atomic_load(dst + from + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        sh.ring[(o + j) & RMASK] = v;
                        dst[o + j] = v;
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                __builtin_amdgcn_wave_barrier();
                o += len;
            }
        }
        if (failed || o != job.isize) {
            if (lane == 0) { atomicAdd(&ctr[1], 1ull); atomicMin(&ctr[2], (unsigned long long)job_id); }
            continue;
        }
        __builtin_amdgcn_wave_barrier();           // the next member reuses the window
    }
}

}  // namespace

// Index of a BGZF file image: for every member the deflate payload and the size it inflates to.  Returns KV_OK and
// *is_bgzf = 0 for anything that is not BGZF from its first byte to its last (plain gzip, a truncated file, ...).
int kv_bgzf_index(const uint8_t *file, uint64_t size, std::vector<KvBgzfMember> *members, int *is_bgzf)
{
    members->clear();
    *is_bgzf = 0;
    uint64_t pos = 0;
    while (pos < size) {
        if (size - pos < 18 + 8) return KV_OK;
        const uint8_t *h = file + pos;
        if (h[0] != 0x1f || h[1] != 0x8b || h[2] != 8 || (h[3] & 4) == 0) return KV_OK;
        if (h[3] & ~4u) return KV_OK;                           // name / comment / header CRC fields: not written by bgzip
        const uint32_t xlen = h[10] | (h[11] << 8);
        if (size - pos < 12 + (uint64_t)xlen + 8) return KV_OK;
        uint32_t bsize = 0;
        bool found = false;
        for (uint32_t x = 0; x + 4 <= xlen;) {
            const uint8_t *sf = h + 12 + x;
            const uint32_t slen = sf[2] | (sf[3] << 8);
            if (sf[0] == 'B' && sf[1] == 'C' && slen == 2 && x + 6 <= xlen) { bsize = (sf[4] | (sf[5] << 8)) + 1u; found = true; }
            x += 4 + slen;
        }
        if (!found || bsize < 12 + xlen + 8 || pos + bsize > size) return KV_OK;
        KvBgzfMember m;
        m.in_off = pos + 12 + xlen;
        m.in_len = bsize - 12 - xlen - 8;
        const uint8_t *t = file + pos + bsize - 4;
        m.isize = t[0] | (t[1] << 8) | (t[2] << 16) | ((uint32_t)t[3] << 24);
        m.crc = t[-4] | (t[-3] << 8) | (t[-2] << 16) | ((uint32_t)t[-1] << 24);
        m.pad = 0;
        if (m.isize > INF_MAX_OUT) return KV_OK;
        members->push_back(m);
        pos += bsize;
    }
    *is_bgzf = members->empty() ? 0 : 1;
    return KV_OK;
}

// Inflate members [first, first + count) of a BGZF image whose bytes [comp_base, comp_base + comp_len) sit at d_comp (with
// 512 readable bytes behind them): member i's text goes to d_text + text_off[i].  Runs on the calling thread's stream and
// returns once the text is there.
int kv_bgzf_inflate(const uint8_t *d_comp, uint64_t comp_base, const KvBgzfMember *members, uint64_t count, const uint64_t *text_off,
                    uint8_t *d_text, KvArena &scratch)
{
    if (count == 0) return KV_OK;
    KV_REQUIRE(count < (1ull << 31), KV_ERR_ARG, "kv_bgzf_inflate: too many members in one call");
    hipStream_t st = kv_stream();
    std::vector<InflateJob> jobs(count);
    for (uint64_t i = 0; i < count; ++i) {
        jobs[i].in_off = members[i].in_off - comp_base;
        jobs[i].in_len = members[i].in_len;
        jobs[i].isize = members[i].isize;
        jobs[i].out_off = text_off[i];
    }
    const size_t b_jobs = kv_round_up(count * sizeof(InflateJob), 256);
    KV_HIP(scratch.need(b_jobs + 256));
    InflateJob *d_jobs = (InflateJob *)scratch.p;
    unsigned long long *d_ctr = (unsigned long long *)((unsigned char *)scratch.p + b_jobs);
    const unsigned long long init[3] = {0, 0, ~0ull};
    KV_HIP(hipMemcpyAsync(d_jobs, jobs.data(), count * sizeof(InflateJob), hipMemcpyHostToDevice, st));
    KV_HIP(hipMemcpyAsync(d_ctr, init, sizeof(init), hipMemcpyHostToDevice, st));
    {
        KvProfScope prof("k_inflate");
        // the LDS window sets how many members a CU holds: 1 KB -> 32 workgroups = 8 waves per SIMD, the most the hardware
        // takes (the kernel is then held to 64 VGPRs; what spills is the table construction, not the symbol loop); matches
        // that reach further back (up to 32 KB) read the text from HBM.  Measured on 846 MB of FASTQ text (4 M reads, bgzip
        // level 4), k_inflate alone, one box: 1 KB 27.2 ms, 2 KB 31.4 ms, 4 KB 35.5 ms; on another box 2 KB 42.4, 4 KB 45.9,
        // 8 KB 52.1, 16 KB ~78, 32 KB (no HBM reads at all) ~100 ms.  KV_INFLATE_WINDOW_BITS = 10 .. 15 for experiments.
        const char *wb = kv_knob("KV_INFLATE_WINDOW_BITS");
        const int bits = wb ? atoi(wb) : 10;
        const int per_cu = bits >= 15 ? 4 : bits == 14 ? 8 : bits == 13 ? 12 : bits == 12 ? 16 : bits == 11 ? 20 : 32;
        const unsigned grid = (unsigned)std::min<uint64_t>(count, (uint64_t)per_cu * (uint64_t)kv_device_cus());
#define KV_LAUNCH_INFLATE(B_) hipLaunchKernelGGL(k_inflate<B_>, dim3(grid), dim3(64), 0, st, d_comp, (const InflateJob *)d_jobs, (uint32_t)count, d_text, d_ctr)
        if (bits >= 15) KV_LAUNCH_INFLATE(15);
        else if (bits == 14) KV_LAUNCH_INFLATE(14);
        else if (bits == 13) KV_LAUNCH_INFLATE(13);
        else if (bits == 12) KV_LAUNCH_INFLATE(12);
        else if (bits == 11) KV_LAUNCH_INFLATE(11);
        else KV_LAUNCH_INFLATE(10);
#undef KV_LAUNCH_INFLATE
    }
    KV_HIP(hipGetLastError());
    unsigned long long ctr[3] = {0, 0, 0};
    KV_HIP(hipMemcpyAsync(ctr, d_ctr, sizeof(ctr), hipMemcpyDeviceToHost, st));
    KV_HIP(hipStreamSynchronize(st));
    if (ctr[1] != 0) {
        kv_set_error("corrupt BGZF data: %llu member(s) did not inflate to their stated size (first: member %llu of the batch)", ctr[1], ctr[2]);
        return KV_ERR_IO;
    }
    // the members' CRC-32, as zlib / htslib would check it: slices of 16 KB of every member's text, joined per member
    const char *crc_env = kv_knob("KV_GUNZIP_CRC");
    if (!(crc_env && !strcmp(crc_env, "0"))) {
        std::vector<uint64_t> r_start;
        std::vector<uint32_t> r_len;
        r_start.reserve(count * 8); r_len.reserve(count * 8);
        for (uint64_t i = 0; i < count; ++i)
            for (uint32_t at = 0; at < members[i].isize; at += KV_CRC_SLICE) {
                r_start.push_back(text_off[i] + at);
                r_len.push_back(std::min<uint32_t>(KV_CRC_SLICE, members[i].isize - at));
            }
        std::vector<uint32_t> crcs(r_start.size());
        { const int rc = kv_crc32_ranges(d_text, r_start.data(), r_len.data(), r_start.size(), crcs.data(), scratch); if (rc != KV_OK) return rc; }
        size_t r = 0;
        for (uint64_t i = 0; i < count; ++i) {
            uint32_t run = 0;
            for (uint32_t at = 0; at < members[i].isize; at += KV_CRC_SLICE, ++r) run = kv_crc32_join(run, crcs[r], r_len[r]);
            if (run != members[i].crc) {
                kv_set_error("corrupt BGZF data: the CRC-32 of member %llu of the batch does not match its text", (unsigned long long)i);
                return KV_ERR_IO;
            }
        }
    }
    return KV_OK;
}

// Whole-buffer form for tests and tools: inflate a BGZF image that sits in host memory; `out` must hold the sum of the
// members' sizes (kv_bgzf_text_size).
extern "C" int kv_bgzf_text_size(const void *file, uint64_t size, uint64_t *text_bytes, uint64_t *n_members)
{
    KV_REQUIRE(file && text_bytes, KV_ERR_ARG, "kv_bgzf_text_size: null argument");
    std::vector<KvBgzfMember> members;
    int yes = 0;
    kv_bgzf_index((const uint8_t *)file, size, &members, &yes);
    KV_REQUIRE(yes, KV_ERR_IO, "not a BGZF file image");
    uint64_t total = 0;
    for (const KvBgzfMember &m : members) total += m.isize;
    *text_bytes = total;
    if (n_members) *n_members = members.size();
    return KV_OK;
}

extern "C" int kv_bgzf_inflate_host(const void *file, uint64_t size, void *out, uint64_t out_cap, double *kernel_ms)
{
    KV_REQUIRE(file && out, KV_ERR_ARG, "kv_bgzf_inflate_host: null argument");
    std::vector<KvBgzfMember> members;
    int yes = 0;
    kv_bgzf_index((const uint8_t *)file, size, &members, &yes);
    KV_REQUIRE(yes, KV_ERR_IO, "not a BGZF file image");
    std::vector<uint64_t> text_off(members.size());
    uint64_t total = 0;
    for (size_t i = 0; i < members.size(); ++i) { text_off[i] = total; total += members[i].isize; }
    KV_REQUIRE(total <= out_cap, KV_ERR_CAPACITY, "kv_bgzf_inflate_host: the text needs %llu bytes", (unsigned long long)total);
    uint8_t *d_comp = nullptr, *d_text = nullptr;
    KvArena scratch;
    hipStream_t st = kv_stream();
    int rc = KV_OK;
    hipError_t e = kv_hip_malloc((void **)&d_comp, size + KV_INFLATE_SLACK);
    if (e == hipSuccess) e = kv_hip_malloc((void **)&d_text, total + 64);
    if (e == hipSuccess) e = hipMemcpyAsync(d_comp, file, size, hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = hipMemsetAsync(d_comp + size, 0, KV_INFLATE_SLACK, st);
    if (e == hipSuccess) {
        hipEvent_t a, b;
        (void)hipEventCreate(&a); (void)hipEventCreate(&b);
        (void)hipEventRecord(a, st);
        rc = kv_bgzf_inflate(d_comp, 0, members.data(), members.size(), text_off.data(), d_text, scratch);
        (void)hipEventRecord(b, st);
        (void)hipEventSynchronize(b);
        float ms = 0;
        (void)hipEventElapsedTime(&ms, a, b);
        if (kernel_ms) *kernel_ms = ms;
        (void)hipEventDestroy(a); (void)hipEventDestroy(b);
        if (rc == KV_OK) e = hipMemcpy(out, d_text, total, hipMemcpyDeviceToHost);
    }
    if (d_comp) (void)hipFree(d_comp);
    if (d_text) (void)hipFree(d_text);
    if (scratch.p) (void)hipFree(scratch.p);
    if (rc != KV_OK) return rc;
    if (e != hipSuccess) { kv_set_error("kv_bgzf_inflate_host: %s", hipGetErrorString(e)); return KV_ERR_HIP; }
    return KV_OK;
}
