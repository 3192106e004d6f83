"""CPU prototype of the three-threshold level-set recursion of kernels_tv.hip (round 3).

Same data flow as the device code - one state byte per sample, region records by head (hrec) and
parent records by end (erec), the four passes of a level (clip forward scan, decode backward
scan, boundary backward scan, record creation) - with every scan done sequentially.  It exists
to pin the mathematics of the 4-way split (any thresholds give valid level sets; tau = region
mean decides termination) against the DP oracle before the parallel kernels are trusted; it is a
development tool, not part of the product or of the test-suite's oracle.

    python tools_tv3_prototype.py        # random cases against oracle/c_oracle.tv1d
"""

import numpy as np

HEAD, END = 1, 2
DONE = 0xFC


def side_sign(code):
    return 1 if code == 1 else (-1 if code == 2 else 0)


def tv1d_levelsets3(y, lam, max_levels=10000, stats=None, kappa=0.5, root=0.67, mode='parent', nblk=16, bfac=0.67):
    y = np.asarray(y, dtype=np.float64)
    n = y.size
    if n == 1 or lam == 0:
        return y.copy()
    Pp = np.concatenate([[0.0], np.cumsum(y)])
    st = np.zeros(n, dtype=np.int64)
    x = np.zeros(n)
    h_tau, h_del, h_flags = np.zeros(n), np.zeros(n), np.zeros(n, dtype=np.int64)  # by head
    e_tau, e_del, e_cr = np.zeros(n), np.zeros(n), np.zeros(n, dtype=np.int64)    # by end
    st[0] |= HEAD
    st[n - 1] |= END
    nb = min(64, n)
    edges = (np.arange(nb + 1) * n) // nb
    means = (Pp[edges[1:]] - Pp[edges[:-1]]) / np.maximum(1, edges[1:] - edges[:-1])
    h_tau[0] = Pp[n] / n
    h_del[0] = root * means.std()
    h_flags[0] = 0
    level = 0
    while True:
        level += 1
        assert level < max_levels
        # ---- forward clip scan (3 thresholds)
        d = np.zeros(3)
        l = -1
        tau = dl = 0.0
        flags = 0
        for i in range(n):
            b = int(st[i])
            if b & HEAD:
                l = i
                tau, dl, flags = h_tau[l], h_del[l], int(h_flags[l])
            if (b & DONE) == DONE:
                d[:] = 0.0
                continue
            if flags & 16:  # region finished a level ago: write x now
                x[i] = tau
                st[i] = (b & 3) | DONE
                d[:] = 0.0
                continue
            yp = y[i]
            if b & HEAD:
                yp -= lam * side_sign(flags & 3)
            if b & END:
                yp -= lam * side_sign((flags >> 2) & 3)
                e_tau[i], e_del[i], e_cr[i] = tau, dl, (flags >> 2) & 3
            a = tau - yp
            nbyte = b & 3
            for j, off in enumerate((-dl, 0.0, dl)):
                aj = a + off
                if b & HEAD:
                    d[j] = aj
                else:
                    d[j] = aj + min(max(d[j], -lam), lam)
                if b & END:
                    c = 1 if d[j] < 0 else 0
                elif d[j] < -lam:
                    c = 1
                elif d[j] >= lam:
                    c = 0
                else:
                    c = 2
                nbyte |= c << (2 + 2 * j)
            st[i] = nbyte
        # ---- backward decode
        run = [2, 2, 2]
        for i in range(n - 1, -1, -1):
            b = int(st[i])
            if (b & DONE) == DONE:
                run = [0, 0, 0]
                continue
            nbyte = b & 3
            for j in range(3):
                c = (b >> (2 + 2 * j)) & 3
                if c != 2:
                    run[j] = c
                u = run[j] if run[j] != 2 else 0
                nbyte |= u << (2 + 2 * j)
            st[i] = nbyte

        def label(b):
            return ((b >> 2) & 1) + ((b >> 4) & 1) + ((b >> 6) & 1)

        # ---- boundaries + records
        st2 = np.zeros(n, dtype=np.int64)
        cuts = 0
        new_head = np.zeros(n, dtype=bool)
        new_end = np.zeros(n, dtype=bool)
        for i in range(n):
            b = int(st[i])
            active = (b & DONE) != DONE
            head, end = bool(b & HEAD), bool(b & END)
            cr = active and not end and label(int(st[i + 1])) != label(b)
            cl = active and not head and label(int(st[i - 1])) != label(b)
            new_head[i] = head or cl
            new_end[i] = end or cr
            cuts += int(cr)
        near_new = np.full(n, -1)
        near_old = np.full(n, -1)
        nn = no = -1
        for i in range(n - 1, -1, -1):
            if new_end[i]:
                nn = i
            if int(st[i]) & END:
                no = i
            near_new[i], near_old[i] = nn, no
        for i in range(n):
            b = int(st[i])
            active = (b & DONE) != DONE
            st2[i] = (HEAD if new_head[i] else 0) | (END if new_end[i] else 0) | (0 if active else DONE)
            if not (new_head[i] and active):
                continue
            r, eo = near_new[i], near_old[i]
            lab = label(b)
            if b & HEAD:
                clc = int(h_flags[i]) & 3
            else:
                clc = 1 if lab > label(int(st[i - 1])) else 2
            if r == eo:
                crc = int(e_cr[eo])
            else:
                crc = 1 if label(int(st[r])) > label(int(st[r + 1])) else 2
            tp, dp = e_tau[eo], e_del[eo]
            tot = Pp[r + 1] - Pp[i] - lam * (side_sign(clc) + side_sign(crc))
            t_new = tot / (r - i + 1)
            t1, t2, t3 = tp - dp, tp, tp + dp
            if lab == 0:
                d_new = kappa * (t1 - t_new)
            elif lab == 3:
                d_new = kappa * (t_new - t3)
            elif lab == 1:
                d_new = kappa * min(t_new - t1, t2 - t_new)
            else:
                d_new = kappa * min(t_new - t2, t3 - t_new)
            d_new = max(d_new, 0.0)
            if mode != 'parent':
                L = r - i + 1
                nbk = min(nblk, L)
                ed = i + (np.arange(nbk + 1) * L) // nbk
                bm = (Pp[ed[1:]] - Pp[ed[:-1]]) / np.maximum(1, ed[1:] - ed[:-1])
                d_blk = bfac * bm.std()
                d_new = d_blk if mode == 'blocks' else min(d_new, d_blk) if mode == 'min' else max(d_new, d_blk)
            fin = bool(b & HEAD) and r == eo
            h_tau[i], h_del[i] = t_new, np.float32(d_new)
            h_flags[i] = clc | (crc << 2) | (16 if fin else 0)
        if stats is not None:
            stats.setdefault('cuts', []).append(cuts)
            stats.setdefault('active', []).append(int(np.sum((st2 & DONE) != DONE)))
        st = st2
        if cuts == 0:
            break
    # flush: every remaining region is constant
    l = -1
    for i in range(n):
        b = int(st[i])
        if b & HEAD:
            l = i
        if (b & DONE) != DONE:
            x[i] = h_tau[l]
    if stats is not None:
        stats["levels"] = level
    return x


if __name__ == "__main__":
    import sys
    sys.path.insert(0, ".")
    from oracle import c_oracle
    rng = np.random.RandomState(3)
    worst = 0.0
    for trial in range(60):
        kind = trial % 6
        n = int(rng.randint(2, 900))
        if kind == 0:
            v, lam = rng.randn(n), float(rng.uniform(0.05, 3))
        elif kind == 1:
            v, lam = np.cumsum(rng.randn(n)) + rng.randn(n), float(rng.uniform(1, 30))
        elif kind == 2:
            v, lam = np.round(rng.randn(n) * 2), 1.0
        elif kind == 3:
            v, lam = np.repeat(rng.randn(max(1, n // 30) + 1), 30)[:n] + 0.1 * rng.randn(n), 3.0
        elif kind == 4:
            v, lam = np.zeros(n), 1.0
        else:
            v, lam = rng.randn(n) * 100, 1e-3
        st = {}
        got = tv1d_levelsets3(v, lam, stats=st)
        want = c_oracle.tv1d(v, lam)
        err = float(np.abs(got - want).max())
        worst = max(worst, err)
        assert err < 1e-8 * max(1.0, np.abs(v).max()), (trial, kind, n, err)
    # level count against the binary recursion's log2
    for n in (3000, 30000):
        v = np.repeat(rng.randn(n // 100) * 3, 100) + rng.randn(n)
        st = {}
        got = tv1d_levelsets3(v, 5.0, stats=st)
        want = c_oracle.tv1d(v, 5.0)
        pieces = 1 + int(np.sum(np.abs(np.diff(want)) > 1e-12))
        print("n=%d pieces=%d levels=%d err=%.2e" % (n, pieces, st["levels"], np.abs(got - want).max()))
    print("ok, worst error %.2e" % worst)
