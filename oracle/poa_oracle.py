"""TEST INFRASTRUCTURE ONLY (see oracle/savont_oracle.h): a plain-Python restatement of Stage 4a, `generate_consensus_poa`
(src/alignment.rs:193-231): partial order alignment of the reads of a cluster and the heaviest-bundle consensus.

The reference calls spoars 0.1.3 (a Rust port of spoa), a third-party crate that is NOT in the reference tree (Cargo.lock:
`spoars 0.1.3`), with `Scoring::new(3, -8, -6, -6, 0, 0)` (linear gaps: open == extend), `AlignmentType::Overlap` and
`BandConfig { base: max length deviation, frac: 0.1 }` ("Band = 0.1 * seq_len + |ref_len - seq_len|", :209-221).  What is restated
here is the published spoa algorithm (Vaser et al. 2017; Lee, Grasso & Sharlow 2002):

  * graph of nodes (one base each) and weighted edges; nodes of different bases aligned to each other form "aligned" sets;
  * sequence-to-graph DP over the nodes in topological order (aligned nodes adjacent), linear gap -6, match 3, mismatch -8;
    overlap mode: leading and trailing overhangs of the sequence AND of the graph are free;
  * traceback priority (mis)match, then deletion (a node without a base), then insertion; predecessors in in-edge order;
  * fusing an alignment: reuse the node (same base) or an aligned sibling, else a new node; edge weight += w[i-1] + w[i];
  * consensus = heaviest bundle with branch completion (spoa's rules).
Choices spoa / the call site leave open and this restatement makes (the product makes the same): the band is
centred on the rounded mean of the 1-based positions the bases fused into a node had in their own reads; the best end cell is the
FIRST maximum in (topological row, column) order over the sink rows and the last column.

Everything is plain Python integers, lists and dicts: no band-relative storage, no 16-bit cells, no SIMD, no "ramped" frame.  It is
what savont_amd/csrc/host/poa.hpp (AVX-512, 16-bit rows in a ramped frame relative to a per-row base) and the K11 kernel must
reproduce alignment by alignment.  PARITY PINNING: spoars cannot be executed here; beyond this restatement Stage 4a is pinned only
by the reference's end-to-end criterion (tests/integration_test.rs:91-160) -- "parity unpinned" for spoars itself.
"""

M, X, G = 3, -8, -6
NEG = -(10 ** 9)


class Graph:
    def __init__(self):
        self.code = []; self.inn = []; self.out = []; self.aligned = []; self.pos_sum = []; self.pos_n = []
        self.e_tail = []; self.e_head = []; self.e_w = []
        self.rank = []

    # ---- bookkeeping
    def add_node(self, code, seq_pos):
        self.code.append(code); self.inn.append([]); self.out.append([]); self.aligned.append([]); self.pos_sum.append(seq_pos + 1); self.pos_n.append(1)
        return len(self.code) - 1

    def note_position(self, v, seq_pos):
        self.pos_sum[v] += seq_pos + 1; self.pos_n[v] += 1

    def add_edge(self, tail, head, w):
        for e in self.out[tail]:
            if self.e_head[e] == head:
                self.e_w[e] += w
                return
        self.e_tail.append(tail); self.e_head.append(head); self.e_w.append(w)
        self.out[tail].append(len(self.e_w) - 1); self.inn[head].append(len(self.e_w) - 1)

    def add_chain(self, seq, w, begin, end):
        if begin >= end:
            return -1
        first = self.add_node(seq[begin], begin)
        for i in range(begin + 1, end):
            n = self.add_node(seq[i], i)
            self.add_edge(n - 1, n, w[i - 1] + w[i])
        return first

    def band_column(self, v):
        return (2 * self.pos_sum[v] + self.pos_n[v]) // (2 * self.pos_n[v]) if self.pos_n[v] else 1

    def topological_sort(self):
        """depth-first over in-edges; a node is emitted together with its aligned set, so aligned nodes are adjacent rows"""
        n = len(self.code)
        mark = [0] * n; chk = [0] * n; rank = []
        for s in range(n):
            if mark[s]:
                continue
            st = [s]
            while st:
                c = st[-1]
                valid = True
                if mark[c] != 2:
                    for e in self.inn[c]:
                        if mark[self.e_tail[e]] != 2:
                            st.append(self.e_tail[e]); valid = False
                    if not chk[c]:
                        for a in self.aligned[c]:
                            if mark[a] != 2:
                                st.append(a); chk[a] = 1; valid = False
                    if valid:
                        mark[c] = 2
                        if not chk[c]:
                            rank.append(c)
                            rank.extend(self.aligned[c])
                    else:
                        mark[c] = 1
                if valid:
                    st.pop()
        self.rank = rank

    # ---- alignment
    def align(self, seq, band_base, band_frac):
        """-> list of (node or -1, sequence position or -1), start to end"""
        L = len(seq); N = len(self.rank)
        if N == 0 or L == 0:
            return []
        bw = band_base + int(band_frac * L) + 1
        row_of = {v: i + 1 for i, v in enumerate(self.rank)}
        lo = [0] * (N + 1); hi = [L] * (N + 1)
        for i in range(1, N + 1):
            c = self.band_column(self.rank[i - 1])
            lo[i] = min(L, max(0, c - bw)); hi[i] = min(L, c + bw)
        H = [None] * (N + 1)
        H[0] = {j: 0 for j in range(0, L + 1)}                          # free sequence prefix
        preds = [None] * (N + 1)
        best = NEG; bi = bj = 0
        for i in range(1, N + 1):
            v = self.rank[i - 1]
            ps = [row_of[self.e_tail[e]] for e in self.inn[v]] or [0]     # no in-edges: the virtual source row
            preds[i] = ps
            row = {}
            code = self.code[v]
            for j in range(lo[i], hi[i] + 1):
                if j == 0:
                    row[0] = 0                                          # free graph prefix
                    continue
                sc = M if seq[j - 1] == code else X
                val = NEG
                for p in ps:
                    a = H[p].get(j - 1, NEG)
                    if a > NEG and a + sc > val:
                        val = a + sc
                    b = H[p].get(j, NEG)
                    if b > NEG and b + G > val:
                        val = b + G
                left = row.get(j - 1, NEG)
                if left > NEG and left + G > val:
                    val = left + G
                row[j] = val
            H[i] = row
            if not self.out[v]:                                         # sink: free trailing graph... every column of the row may end the alignment
                for j in range(lo[i], hi[i] + 1):
                    if row[j] > best:
                        best = row[j]; bi = i; bj = j
            elif hi[i] == L and row[L] > best:                          # last column: free trailing sequence overhang
                best = row[L]; bi = i; bj = L
        if best <= NEG // 2:
            return []
        out = []
        i, j = bi, bj
        while i > 0 and j > 0:
            v = self.rank[i - 1]
            val = H[i][j]
            sc = M if seq[j - 1] == self.code[v] else X
            moved = False
            for p in preds[i]:
                a = H[p].get(j - 1, NEG)
                if a > NEG and a + sc == val:
                    out.append((v, j - 1)); i = p; j -= 1; moved = True
                    break
            if not moved:
                for p in preds[i]:
                    b = H[p].get(j, NEG)
                    if b > NEG and b + G == val:
                        out.append((v, -1)); i = p; moved = True
                        break
            if not moved:
                left = H[i].get(j - 1, NEG)
                if j - 1 >= lo[i] and left > NEG and left + G == val:
                    out.append((-1, j - 1)); j -= 1
                else:
                    break                                               # a free start (value 0 at the band edge / column 0)
        out.reverse()
        return out

    def add_alignment(self, aln, seq, w):
        L = len(seq)
        if L == 0:
            return
        valid = [p for (_, p) in aln if p != -1]
        if not valid:
            self.add_chain(seq, w, 0, L); self.topological_sort()
            return
        prev = -1; prev_pos = valid[0] - 1
        if self.add_chain(seq, w, 0, valid[0]) >= 0:
            prev = len(self.code) - 1
        tail_first = self.add_chain(seq, w, valid[-1] + 1, L)
        for (node, p) in aln:
            if p == -1:
                continue
            letter = seq[p]
            if node == -1:
                cur = self.add_node(letter, p)
            elif self.code[node] == letter:
                cur = node; self.note_position(cur, p)
            else:
                cur = -1
                for a in self.aligned[node]:
                    if self.code[a] == letter:
                        cur = a; self.note_position(cur, p)
                        break
                if cur < 0:
                    cur = self.add_node(letter, p)
                    for a in self.aligned[node]:
                        self.aligned[cur].append(a); self.aligned[a].append(cur)
                    self.aligned[cur].append(node); self.aligned[node].append(cur)
            if prev >= 0:
                self.add_edge(prev, cur, w[prev_pos] + w[p])
            prev = cur; prev_pos = p
        if tail_first >= 0:
            self.add_edge(prev, tail_first, w[valid[-1]] + w[valid[-1] + 1])
        self.topological_sort()

    # ---- consensus: heaviest bundle + branch completion (spoa Graph::TraverseHeaviestBundle / BranchCompletion)
    def consensus(self):
        N = len(self.rank)
        if N == 0:
            return b""
        n = len(self.code)
        score = [0] * n; pred = [-1] * n
        mx = -1
        for v in self.rank:
            for e in self.inn[v]:
                t = self.e_tail[e]
                if score[t] < 0:
                    continue
                if score[v] < self.e_w[e] or (score[v] == self.e_w[e] and pred[v] >= 0 and score[pred[v]] <= score[t]):
                    score[v] = self.e_w[e]; pred[v] = t
            if pred[v] >= 0:
                score[v] += score[pred[v]]
            if mx < 0 or score[mx] < score[v]:
                mx = v
        pos = {v: i for i, v in enumerate(self.rank)}
        while self.out[mx]:
            for e in self.out[mx]:
                for e2 in self.inn[self.e_head[e]]:
                    if self.e_tail[e2] != mx:
                        score[self.e_tail[e2]] = -1
            nmx = -1; best = 0
            for i in range(pos[mx] + 1, N):
                v = self.rank[i]
                score[v] = -1; pred[v] = -1
                sv = -1; pv = -1
                for e in self.inn[v]:
                    t = self.e_tail[e]
                    if score[t] == -1:
                        continue
                    if sv < self.e_w[e] or (sv == self.e_w[e] and pv >= 0 and score[pv] <= score[t]):
                        sv = self.e_w[e]; pv = t
                if pv >= 0:
                    score[v] = sv + score[pv]; pred[v] = pv
                    if nmx < 0 or best < score[v]:
                        nmx = v; best = score[v]
            if nmx < 0:
                break
            mx = nmx
        out = []
        v = mx
        while v >= 0:
            out.append(self.code[v]); v = pred[v]
        return bytes(reversed(out))


def poa_consensus(seqs, quals=None, no_band=False):
    """generate_consensus_poa (src/alignment.rs:193-231): sequences (bytes) + per-base weights (quality bytes, 1 when absent) -> (consensus, #graph nodes).
    no_band: the hidden --no-band flag (:198,217): spoa's unbanded engine, here a band that holds every column of every row"""
    if not seqs:
        return b"", 0
    ref_len = sum(len(s) for s in seqs) // len(seqs)                    # :211
    max_dev = max(abs(ref_len - len(s)) for s in seqs)                   # :212-215
    if no_band:
        max_dev = max(max_dev, max(len(s) for s in seqs) + 1)
    g = Graph()
    for i, s in enumerate(seqs):
        w = list(quals[i]) if quals is not None else [1] * len(s)
        aln = g.align(s, max_dev, 0.1)                                  # BandConfig { base: max_deviation, frac: 0.1 } :220
        g.add_alignment(aln, s, w)
    return g.consensus(), len(g.code)
