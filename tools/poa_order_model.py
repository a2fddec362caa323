"""Model of the device-resident POA engine's bookkeeping (K12, savont_amd/csrc/kernels_poa_graph.hip), checked against the
plain-Python oracle (oracle/poa_oracle.py).  Test / design tool only -- nothing in the product imports it.

What the kernel does differently from spoa's bookkeeping, and what this model checks on random clusters:
  1. ORDER.  spoa re-sorts the graph after every read (depth-first, aligned nodes adjacent).  The kernel keeps ANY valid
     topological order in which every aligned set is a contiguous block, and splices the new nodes of a read in:
         new order = for every old block B: B, then the read's new sibling of B (if any), then the read's inserted bases
                     that follow B on the path, in sequence order;  an unaligned prefix goes first.
     Claim: that is again a valid order with contiguous blocks (new edges between old nodes only shortcut existing paths).
  2. DP values do not depend on the topological order; the only order-dependent choice is the end cell among EQUAL maxima in
     different rows (spoa: first in its own order).  The kernel detects such ties; the model counts them and checks that without
     a tie the alignment equals the oracle's.
  3. FUSE, position-parallel: every sequence position decides on its own (old node / sibling / new node); new ids come from a
     prefix sum in the host's creation order (prefix chain, suffix chain, then path order); edges (cur[p-1] -> cur[p]).
     Claim: same graph as the serial add_alignment, node for node, in-edge order and aligned-list order included.
"""
import os
import random
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "oracle"))
import poa_oracle as po  # noqa: E402

M, X, G, NEG = po.M, po.X, po.G, po.NEG


class DevGraph(po.Graph):
    """same storage as the oracle's graph; rank is maintained incrementally"""

    def block_end_row(self, v, row_of):
        return max([row_of[v]] + [row_of[a] for a in self.aligned[v] if a in row_of])          # old members only

    def align_rows(self, seq, band_base, band_frac):
        """the oracle's DP on self.rank (any valid order) -> (aln_row per sequence position or None, tie flag)"""
        L = len(seq); N = len(self.rank)
        bw = band_base + int(band_frac * L) + 1
        row_of = {v: i + 1 for i, v in enumerate(self.rank)}
        lo = [0] * (N + 1); hi = [L] * (N + 1)
        for i in range(1, N + 1):
            c = self.band_column(self.rank[i - 1])
            lo[i] = min(L, max(0, c - bw)); hi[i] = min(L, c + bw)
        H = [None] * (N + 1); H[0] = {j: 0 for j in range(L + 1)}
        preds = [None] * (N + 1)
        best = NEG; bi = bj = 0; tie = False
        for i in range(1, N + 1):
            v = self.rank[i - 1]
            ps = [row_of[self.e_tail[e]] for e in self.inn[v]] or [0]
            assert all(p < i for p in ps), "order is not topological"
            preds[i] = ps
            row = {}
            for j in range(lo[i], hi[i] + 1):
                if j == 0:
                    row[0] = 0; continue
                sc = M if seq[j - 1] == self.code[v] else X
                val = NEG
                for p in ps:
                    a = H[p].get(j - 1, NEG)
                    if a > NEG and a + sc > val: val = a + sc
                    b = H[p].get(j, NEG)
                    if b > NEG and b + G > val: val = b + G
                left = row.get(j - 1, NEG)
                if left > NEG and left + G > val: val = left + G
                row[j] = val
            H[i] = row
            cand = list(range(lo[i], hi[i] + 1)) if not self.out[v] else ([L] if hi[i] == L else [])
            for j in cand:
                if row[j] > best:
                    best = row[j]; bi = i; bj = j; tie = False
                elif row[j] == best and i != bi:
                    tie = True
        if best <= NEG // 2:
            return None, 0, -1, False, row_of
        aln = {}
        i, j = bi, bj
        while i > 0 and j > 0:
            v = self.rank[i - 1]; val = H[i][j]
            sc = M if seq[j - 1] == self.code[v] else X
            moved = False
            for p in preds[i]:
                a = H[p].get(j - 1, NEG)
                if a > NEG and a + sc == val:
                    aln[j - 1] = i; i = p; j -= 1; moved = True; break
            if not moved:
                for p in preds[i]:
                    b = H[p].get(j, NEG)
                    if b > NEG and b + G == val:
                        i = p; moved = True; break
            if not moved:
                left = H[i].get(j - 1, NEG)
                if j - 1 >= lo[i] and left > NEG and left + G == val:
                    aln[j - 1] = 0; j -= 1
                else:
                    break
        if not aln:
            return None, 0, -1, tie, row_of
        fp, lp = min(aln), max(aln)
        assert sorted(aln) == list(range(fp, lp + 1))
        return aln, fp, lp, tie, row_of

    def fuse_parallel(self, aln, fp, lp, row_of, seq, w):
        """position-parallel fuse + order splice"""
        L = len(seq); N = len(self.rank); n0 = len(self.code)
        has = aln is not None
        kind = [None] * L; cur = [-1] * L; anchor_node = [-1] * L
        for p in range(L):                                           # E1 classify (independent per p)
            if not has or p < fp or p > lp:
                kind[p] = "chain"; continue
            i = aln[p]
            if i == 0:
                kind[p] = "ins"; continue
            nd = self.rank[i - 1]; anchor_node[p] = nd
            if self.code[nd] == seq[p]:
                kind[p] = "old"; cur[p] = nd; continue
            for a in self.aligned[nd]:
                if self.code[a] == seq[p]:
                    kind[p] = "old"; cur[p] = a; break
            else:
                kind[p] = "mm"
        isnew = [k != "old" for k in kind]
        # E2 ids in the host's creation order: prefix chain, suffix chain, then path order
        if has:
            n_pre = fp; n_suf = L - 1 - lp
            run = 0
            for p in range(L):
                if p < fp: cur[p] = n0 + p
                elif p > lp: cur[p] = n0 + n_pre + (p - lp - 1)
                elif isnew[p]:
                    cur[p] = n0 + n_pre + n_suf + run; run += 1
            total_new = n_pre + n_suf + run
        else:
            for p in range(L): cur[p] = n0 + p
            total_new = L
        # E3 create / note
        for _ in range(total_new):
            self.code.append(None); self.inn.append([]); self.out.append([]); self.aligned.append([]); self.pos_sum.append(0); self.pos_n.append(0)
        for p in range(L):
            c = cur[p]
            if isnew[p]:
                self.code[c] = seq[p]; self.pos_sum[c] = p + 1; self.pos_n[c] = 1
                if kind[p] == "mm":
                    x = anchor_node[p]
                    for a in self.aligned[x]:
                        self.aligned[c].append(a); self.aligned[a].append(c)
                    self.aligned[c].append(x); self.aligned[x].append(c)
            else:
                self.pos_sum[c] += p + 1; self.pos_n[c] += 1
        # E4 edges
        for p in range(1, L):
            self.add_edge(cur[p - 1], cur[p], w[p - 1] + w[p])
        # E5 splice: anchor row of every position = end row of the block of its path node (old or mm), carried forward
        A = [0] * L; run = 0
        for p in range(L):
            if has and fp <= p <= lp and aln[p] != 0:
                run = self.block_end_row(anchor_node[p], row_of)
            A[p] = run
        new_rank = [None] * (N + total_new)
        ncnt = [0] * (L + 1)
        for p in range(L): ncnt[p + 1] = ncnt[p] + (1 if isnew[p] else 0)
        for p in range(L):
            if isnew[p]:
                new_rank[A[p] + ncnt[p]] = cur[p]                      # 0-based index of 1-based row A + ncnt + 1
        for i in range(1, N + 1):
            # S(i) = number of new positions anchored before row i = ncnt[first p with A[p] >= i]
            l, h = 0, L
            while l < h:
                m = (l + h) // 2
                if A[m] >= i: h = m
                else: l = m + 1
            new_rank[i - 1 + ncnt[l]] = self.rank[i - 1]
        assert all(x is not None for x in new_rank)
        self.rank = new_rank

    def check_order(self):
        pos = {v: i for i, v in enumerate(self.rank)}
        assert len(pos) == len(self.code)
        for e in range(len(self.e_w)):
            assert pos[self.e_tail[e]] < pos[self.e_head[e]], "edge against the order"
        for v in range(len(self.code)):
            blk = sorted(pos[x] for x in [v] + self.aligned[v])
            assert blk[-1] - blk[0] == len(blk) - 1, "aligned block not contiguous"


def mutate(rng, hap, rate):
    out = bytearray()
    for b in hap:
        u = rng.random()
        if u < rate * 0.4: out.append(rng.choice(b"ACGT"))
        elif u < rate * 0.7: out.append(b); out.append(rng.choice(b"ACGT"))
        elif u < rate: pass
        else: out.append(b)
    return bytes(out)


def run_cluster(rng, L, n, rate, ragged):
    hap = bytes(rng.choice(b"ACGT") for _ in range(L))
    hap2 = bytearray(hap)
    for _ in range(3): hap2[rng.randrange(L)] = rng.choice(b"ACGT")
    hap2 = bytes(hap2[:L // 2] + hap2[L // 2 + 7:])
    seqs = []
    for k in range(n):
        s = mutate(rng, hap if k % 3 else hap2, rate)
        if ragged: s = s[rng.randrange(0, 12):len(s) - rng.randrange(0, 12)]
        seqs.append(s)
    quals = [bytes(rng.randrange(35, 80) for _ in s) for s in seqs]
    ref_len = sum(len(s) for s in seqs) // len(seqs)
    max_dev = max(abs(ref_len - len(s)) for s in seqs)
    g = po.Graph(); d = DevGraph()
    ties = 0
    for s, q in zip(seqs, quals):
        w = list(q)
        aln_o = g.align(s, max_dev, 0.1)
        if len(d.rank) == 0:
            aln, fp, lp, tie, row_of = None, 0, -1, False, {}
        else:
            aln, fp, lp, tie, row_of = d.align_rows(s, max_dev, 0.1)
        # the oracle's alignment in the same form: node per sequence position
        want = {p: nd for (nd, p) in aln_o if p != -1}
        got = {} if aln is None else {p: (d.rank[i - 1] if i else -1) for p, i in aln.items()}
        if tie:
            ties += 1
            if got != want:                                             # follow the oracle (the kernel hands such a cluster to the host)
                rowd = {v: i + 1 for i, v in enumerate(d.rank)}
                aln = {p: (rowd[nd] if nd >= 0 else 0) for p, nd in want.items()} or None
                fp, lp = (min(want), max(want)) if want else (0, -1)
        else:
            assert got == want, "alignment differs without a tie"
        g.add_alignment(aln_o, s, w)
        d.fuse_parallel(aln, fp, lp, row_of, s, w)
        d.check_order()
        # edge ids differ (the oracle creates the chain edges first); what the DP, the fuse and the consensus read is each node's
        # in-list / out-list ORDER with tails, heads and weights
        view = lambda x: ([[(x.e_tail[e], x.e_w[e]) for e in l] for l in x.inn], [[(x.e_head[e], x.e_w[e]) for e in l] for l in x.out])
        assert d.code == g.code and d.aligned == g.aligned and view(d) == view(g), "graphs differ"
        assert d.pos_sum == g.pos_sum and d.pos_n == g.pos_n
    return ties, len(seqs)


if __name__ == "__main__":
    rng = random.Random(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
    tot_t = tot_n = 0
    for it in range(int(sys.argv[2]) if len(sys.argv) > 2 else 12):
        L = rng.choice((60, 120, 200)); n = rng.randrange(3, 25); rate = rng.choice((0.02, 0.05, 0.1))
        t, m = run_cluster(rng, L, n, rate, ragged=bool(it % 2))
        tot_t += t; tot_n += m
        print("cluster %d: L %d, %d reads, rate %.2f: ok, %d end-cell ties" % (it, L, n, rate, t))
    print("all ok; %d ties in %d reads" % (tot_t, tot_n))
