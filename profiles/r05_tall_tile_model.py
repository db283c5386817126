#!/usr/bin/env python3
"""CPU model of the tall-tile runs of panel_mv_kernel (trd.hip, TileMap) and of the partial-sum slots its readers add up.

Tiles of HR = 256 rows x HC = 16 columns over the stored (upper) triangle, ordered row block by row block (row block i: column
blocks 16 i .. ntc - 1); run hb = tiles [start(hb), start(hb + 1)), start(hb) = hb L + min(hb, T mod G), L = T // G.  The map is made for the order n_map of a panel's first
column and used for every order n <= n_map of the panel (tiles whose columns start at or beyond n are skipped).  Writers:
  Y(hb - first run of row block i, r)  <- the row-direction sums of run hb over its tiles of row block i, once per (run, row block),
                                          zeros where the run's tiles of that row block are all skipped;
  C(i, c)                              <- the column-direction sums of tile (i, c / 16), per live tile.
Reader of row r (row block ib): Y(0 .. nr(ib) - 1, r) and C(0 .. ib, r).
This script checks, for many (n_map, n, G), that every element of the triangle is summed exactly once in each direction; it is
the specification the HIP code of profiles/r05_tall_tile_variant.patch was written from (variant not kept: r05_experiments.txt section 10);
while it was in the tree a CPU test ran this model and compared the library's host-side copy of the map against it."""
import sys

HR, HC = 256, 16
Q = HR // HC


class Map:
    def __init__(self, n_map, G):
        self.ntr = (n_map + HR - 1) // HR
        self.ntc = (n_map + HC - 1) // HC
        self.T = self.tbase(self.ntr)
        self.G = max(1, min(G, self.T))
        self.L = self.T // self.G
        self.rem = self.T - self.L * self.G

    def tbase(self, i):
        return i * self.ntc - (Q // 2) * i * (i - 1)

    def start(self, hb):
        return hb * self.L + min(hb, self.rem)      # the first `rem` runs have L + 1 tiles, the others L = T // G

    def run_of(self, t):
        cut = self.rem * (self.L + 1)
        return t // (self.L + 1) if t < cut else self.rem + (t - cut) // self.L

    def row_of(self, t):
        i = 0
        while i + 1 < self.ntr and self.tbase(i + 1) <= t:
            i += 1
        return i

    def nr(self, ib):
        return self.run_of(self.tbase(ib + 1) - 1) - self.run_of(self.tbase(ib)) + 1

    def items(self):
        return max(self.nr(ib) + ib + 1 for ib in range(self.ntr))


def simulate(n_map, n, G):
    """integer 'matrix': element (r, c), r <= c < n, contributes r * n + c + 1 to row sum r and to column sum c (r < c)."""
    m = Map(n_map, G)
    Y, C = {}, {}
    for hb in range(m.G):
        t0, t1 = m.start(hb), m.start(hb + 1)
        assert t1 > t0
        assert m.run_of(t0) == hb and m.run_of(t1 - 1) == hb
        i = m.row_of(t0)
        j = Q * i + (t0 - m.tbase(i))
        cur_i = i
        yacc = None

        def flush(ib):
            nonlocal yacc
            o = hb - m.run_of(m.tbase(ib))
            assert 0 <= o < m.nr(ib)
            for rr in range(HR):
                key = (o, ib * HR + rr)
                assert key not in Y
                Y[key] = 0 if yacc is None else yacc[rr]
            yacc = None

        for t in range(t0, t1):
            ti, tj = i, j
            j += 1
            if j == m.ntc:
                i += 1
                j = Q * i
            if tj * HC >= n:
                continue
            while cur_i < ti:
                flush(cur_i)
                cur_i += 1
            if yacc is None:
                yacc = [0] * HR
            for cc in range(tj * HC, tj * HC + HC):
                tsum = 0
                for rr in range(HR):
                    r = ti * HR + rr
                    if r < n and cc < n and r <= cc:
                        val = r * n + cc + 1
                        yacc[rr] += val
                        if r < cc:
                            tsum += val
                assert (ti, cc) not in C
                C[(ti, cc)] = tsum
        i_last = i - (1 if j == Q * i else 0)
        while cur_i <= i_last:
            flush(cur_i)
            cur_i += 1
    for r in range(n):
        ib = r // HR
        got = sum(Y[(o, r)] for o in range(m.nr(ib))) + sum(C[(ii, r)] for ii in range(ib + 1))
        want = sum(r * n + c + 1 for c in range(r, n)) + sum(q * n + r + 1 for q in range(r))
        assert got == want, (n_map, n, G, r, got, want)
    return m


def main():
    cases = 0
    for n_map in (1, 15, 16, 17, 255, 256, 257, 300, 511, 513, 777, 1030):
        for G in (1, 3, 8, 40, 256):
            for n in sorted({n_map, max(1, n_map - 1), max(1, n_map - 17), max(1, n_map - 33), max(1, n_map - 63)}):
                simulate(n_map, n, G)
                cases += 1
    print(f"{cases} (n_map, n, G) cases: every element summed exactly once in each direction")
    print("     n   ntr   ntc      T     G  rows of P  reader items per row (max)  [64 x 64 scheme: rows = items = n / 64]")
    for n in (1024, 2048, 4096, 6144, 8192, 12288, 16384):
        m = Map(n, 256)
        print(f"{n:6d} {m.ntr:5d} {m.ntc:5d} {m.T:6d} {m.G:5d} {m.ntr + max(m.nr(i) for i in range(m.ntr)):8d} {m.items():8d} {'':20s}{n // 64:5d}")
    return 0


if __name__ == "__main__":
    sys.exit(main())
