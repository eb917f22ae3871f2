import numpy as np, collections
rng = np.random.default_rng(5)
G, L, k, m, cov, err = 400_000, 100, 31, 12, 30, 0.005
w = k - m + 1
genome = rng.integers(0, 4, G, dtype=np.uint8)
n = cov * G // L
starts = rng.integers(0, G - L, n)
reads = genome[starts[:, None] + np.arange(L)[None, :]]
rev = rng.integers(0, 2, n).astype(bool)
reads[rev] = 3 - reads[rev][:, ::-1]
e = rng.random((n, L)) < err
reads = np.where(e, (reads + rng.integers(1, 4, (n, L))) % 4, reads).astype(np.uint8)
# m-mer codes
nm = L - m + 1
f = np.zeros((n, nm), dtype=np.uint64); r = np.zeros((n, nm), dtype=np.uint64)
for i in range(m):
    f |= reads[:, i:i + nm].astype(np.uint64) << np.uint64(2 * i)
    r |= (3 - reads[:, i:i + nm]).astype(np.uint64) << np.uint64(2 * (m - 1 - i))
can = np.minimum(f, r)
order = (can * np.uint64(0x9e3779b1)) & np.uint64(0xffffffff)
order ^= order >> np.uint64(15)
nk = L - k + 1
win = np.lib.stride_tricks.sliding_window_view(order, w, axis=1)   # (n, nk, w)
minv = win.min(axis=2)
change = np.ones((n, nk), dtype=bool); change[:, 1:] = minv[:, 1:] != minv[:, :-1]
tot = 0; whole = 0
cnt = collections.Counter(); cnt_norm = collections.Counter(); kmers_tot = 0; 
def rc(b): return bytes(3 - x for x in reversed(b))
for i in range(n):
    s = np.flatnonzero(change[i]); e2 = np.append(s[1:], nk)
    row = reads[i].tobytes()
    for a, b in zip(s, e2):
        # cut at ncap = 34
        while a < b:
            ln = min(b - a, 34)
            rec = row[a:a + ln + k - 1]
            cnt[rec] += 1
            rr = rc(rec); cnt_norm[min(rec, rr)] += 1
            tot += 1; kmers_tot += ln
            a += ln
print('records', tot, 'per read', tot / n, 'k-mers per record', kmers_tot / tot)
print('distinct records (same strand only)', len(cnt), 'fraction', len(cnt) / tot)
print('distinct records (strand-normalised)', len(cnt_norm), 'fraction', len(cnt_norm) / tot)
kd = sum(len(rk) - k + 1 for rk in cnt); print('k-mer inserts after dedupe (same strand)', kd, 'of', kmers_tot, '=', kd / kmers_tot)
kd = sum(len(rk) - k + 1 for rk in cnt_norm); print('k-mer inserts after dedupe (normalised)', kd, 'of', kmers_tot, '=', kd / kmers_tot)
