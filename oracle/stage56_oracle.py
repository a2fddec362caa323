"""TEST INFRASTRUCTURE ONLY (see oracle/savont_oracle.h): plain-Python restatement of Stage 5 (merge similar consensuses) and
Stage 6 (chimera detection).  Only tests/ may import this.

  minimizer_seeds_positions      src/seeding.rs:99-186
  remove_similar_seqs_kmers      src/alignment.rs:1162-1208
  has_homopolymer_context        src/alignment.rs:75-96
  calculate_adjusted_errors      src/alignment.rs:101-188
  merge_similar_consensuses      src/alignment.rs:1213-1517
  calculate_match_lengths        src/chimera.rs:274-399
  detect_chimeras/filter         src/chimera.rs:37-269, :465-494

The reference's minimap2 calls are replaced by the K7/K8/K9 contracts of the C oracle (orc_strand_vote, orc_align_nm,
orc_align_pileup_row); `aligner(query, target)` below is injected by the test and returns None or
dict(rev, nm, query_start, query_end, target_start, target_end, cigar=[(len, op)]) with op 0 = M, 1 = I, 2 = D.
Two reference behaviours are kept on purpose (see savont_amd/csrc/host/merge_chimera.cpp header): appended_depth is reset by
the rebuild at :1491, and the similarity map of src/chimera.rs is never hit (keys stored as (j, i), looked up as (min, max)).
Parity pinning: no reference tests / golden vectors exist for these functions -> "parity unpinned"."""

M64 = (1 << 64) - 1
_B2S = {1: 1, 2: 2, 3: 3, ord("C"): 1, ord("c"): 1, ord("G"): 2, ord("g"): 2, ord("T"): 3, ord("t"): 3, ord("U"): 3, ord("u"): 3}
_COMP = bytes.maketrans(b"ACGT", b"TGCA")


def mm_hash64(key):
    key = (~key + (key << 21)) & M64
    key ^= key >> 24
    key = (key + (key << 3) + (key << 8)) & M64
    key ^= key >> 14
    key = (key + (key << 2) + (key << 4)) & M64
    key ^= key >> 28
    key = (key + (key << 31)) & M64
    return key


def _position_min(w):
    best = 0
    for i in range(1, len(w)):
        if w[i] <= w[best]:          # max_by with the reversed comparison returns the LAST minimum
            best = i
    return best


def minimizer_seeds(s, w, k):
    out = []
    n = len(s)
    if n < k + w - 1:
        return out
    f = 0
    r = 0
    canonical = 0
    rshift = 2 * (k - 1)
    max_mask = M64 >> (64 - 2 * k)
    rev_mask = ~(3 << (2 * k - 2)) & M64
    win = [M64] * w
    for i in range(k + w - 1):
        nf = _B2S.get(s[i], 0)
        nr = 3 - nf
        f = ((f << 2) | nf) & M64                 # NOT masked to 2k bits here (:126-127)
        r = (r >> 2) | (nr << rshift)
        if i >= k - 1:
            canonical = f if f < r else r
            win[i + 1 - k] = mm_hash64(canonical)
    min_pos = _position_min(win)
    min_val = win[min_pos]
    out.append(canonical)                         # a k-mer, not a hash (:145)
    for i in range(k + w - 1, n):
        nf = _B2S.get(s[i], 0)
        nr = 3 - nf
        f = ((f << 2) | nf) & max_mask
        r = ((r >> 2) & rev_mask) | (nr << rshift)
        h = mm_hash64(f if f < r else r)
        g = i + 1 - k
        win[g % w] = h
        if h < min_val:
            min_val = h
            min_pos = g % w
            out.append(h)
        elif min_pos == g % w:
            min_pos = _position_min(win)
            min_val = win[min_pos]
            out.append(min_val)
    return out


def remove_similar_seqs_kmers(cons):
    """cons: list of dict(seq=masked bytes, depth, ...) -> survivors in input order"""
    index = {}
    minis = {}
    for i, c in enumerate(cons):
        s = c["seq"]
        if len(s) < 100:
            continue
        m = minimizer_seeds(s[25:len(s) - 25], 10, 21)
        for x in m:
            index.setdefault(x, []).append(i)
        minis[i] = m
    out = []
    for e in sorted(minis):
        greater = set()
        first = True
        for x in minis[e]:
            ids = index.get(x)
            if first:
                if ids is not None:
                    greater = {i for i in ids if cons[i]["depth"] // 2 > cons[e]["depth"]}
            elif ids is not None:
                greater &= set(ids)
            first = False
        if not greater:
            out.append(cons[e])
    return out


def has_homopolymer_context(seq, pos, window):
    if not seq:
        return False
    start = max(pos - window, 0)
    end = min(pos + window + 1, len(seq))
    if end <= start + 2:
        return False
    for i in range(start, max(end - 3, 0) + 1):
        if i + 2 < len(seq) and seq[i] == seq[i + 1] == seq[i + 2]:
            return True
    return False


def calculate_adjusted_errors(cigar, q, t, query_start, target_start):
    err = 0
    buffer = 35
    qp = query_start
    tp = target_start
    N = ord("N")
    for ln, op in cigar:
        if op == 0:
            for _ in range(ln):
                if qp < len(q) and tp < len(t):
                    if q[qp] != t[tp] and q[qp] != N and t[tp] != N:
                        if qp > buffer and qp + buffer < len(q):
                            err += 1
                qp += 1
                tp += 1
        elif op == 1:
            hp = has_homopolymer_context(q, qp, 2) or has_homopolymer_context(t, tp, 2)
            if not hp and qp > buffer and qp + ln + buffer < len(q):
                err += 1 if ln < 10 else ln
            qp += ln
        elif op == 2:
            hp = has_homopolymer_context(q, qp, 2) or has_homopolymer_context(t, tp, 2)
            if not hp and tp > buffer and tp + ln + buffer < len(t):
                err += 1 if ln < 10 else ln
            tp += ln
    return err


def _decompress(c):
    """ConsensusSequence::decompress of the merged record (:1491-1492): homopolymer_decompress with its hp_lengths (--use-hpc), then the N trim"""
    hp = c.get("hp_lengths")
    s = c["seq"]
    if hp is not None and len(hp) == len(s):
        s = b"".join(bytes([b]) * int(l) for b, l in zip(s, hp))
    return _trim_n(s)


def _trim_n(s):
    a = 0
    b = len(s)
    while a < b and s[a] == ord("N"):
        a += 1
    while b > a and s[b - 1] == ord("N"):
        b -= 1
    return s[a:b] if a < b else s


def merge_similar_consensuses(cons, aligner):
    """cons: list of dict(seq, decompressed, depth, id, cluster[, hp_lengths]) -> merged list (same dict shape)"""
    if not cons:
        return cons
    cons = remove_similar_seqs_kmers(cons)
    n = len(cons)
    mappings = []
    for q in range(n):
        for t in range(n):
            if q == t:
                continue
            a = aligner(cons[q]["decompressed"], cons[t]["decompressed"])
            if a is None:
                continue
            qs = cons[q]["decompressed"]
            if a["query_end"] - a["query_start"] < len(qs) * 3 // 4 or a["nm"] > 30:
                continue
            qq = qs.translate(_COMP)[::-1] if a["rev"] else qs
            adj = calculate_adjusted_errors(a["cigar"], qq, cons[t]["decompressed"], a["query_start"], a["target_start"])
            adj = min(adj, a["nm"])
            mappings.append((q, t, adj, cons[t]["depth"]))
    merge_map = {}
    for q in range(n):
        valid = []
        for (qi, ti, nm, td) in mappings:
            if qi != q or qi == ti:
                continue
            qd = cons[q]["depth"]
            rel = qd / td
            thr = 0.5 ** (nm * 0.75 + 1.25)
            if nm == 0:
                thr = 0.999999
                if qd == td:
                    if q > ti:
                        valid.append((ti, nm, td))
                    continue
            if rel < thr or 1.0 / rel < thr:
                valid.append((ti, nm, td))
        if valid:
            q2r = []
            r2q = []
            for (ti, nm, td) in valid:
                if cons[ti]["depth"] == cons[q]["depth"]:
                    if nm == 0 and q > ti:
                        merge_map[q] = ti
                    continue
                elif cons[ti]["depth"] > cons[q]["depth"]:
                    q2r.append((ti, nm, td, q))
                else:
                    r2q.append((q, nm, cons[q]["depth"], ti))
            if q2r:
                q2r.sort(key=lambda x: -x[2])
                merge_map[q] = q2r[0][0]
            for (_, _, _, ti) in r2q:
                if ti not in merge_map:
                    merge_map[ti] = q
    new_clusters = [list(c["cluster"]) for c in cons]
    merged_into = {}
    for q in range(n):
        if q in merge_map:
            fin = merge_map[q]
            while fin in merge_map:
                fin = merge_map[fin]
            merged_into[q] = fin
    for q in sorted(merged_into):
        t = merged_into[q]
        new_clusters[t].extend(new_clusters[q])
        new_clusters[q] = []
    out = []
    for i, c in enumerate(cons):
        if new_clusters[i]:
            out.append(dict(seq=c["seq"], hp_lengths=c.get("hp_lengths"), decompressed=_decompress(c), depth=len(new_clusters[i]), id=c["id"], cluster=new_clusters[i]))
    out.sort(key=lambda c: -c["depth"])
    return out


def calculate_match_lengths(a, q, t, allow=1, min_read_length=1100, chimera_detect_length=None):
    left = 0
    right = 0
    pcr_slack = 15
    errs = 0
    qp = a["query_start"]
    tp = a["target_start"]
    for ln, op in a["cigar"]:
        if errs > allow:
            break
        if op == 0:
            for i in range(ln):
                if qp + i < len(q) and tp + i < len(t):
                    if q[qp + i] == t[tp + i]:
                        left += 1
                    else:
                        errs += 1
                        if errs > allow and qp + i >= pcr_slack:
                            break
            qp += ln
            tp += ln
        elif op == 1:
            qp += ln
        else:
            tp += ln
    errs = 0
    qp = a["query_end"]
    tp = a["target_end"]
    for ln, op in reversed(a["cigar"]):
        if errs > allow:
            break
        if op == 0:
            for i in range(ln):
                if q[qp - i - 1] == t[tp - i - 1]:
                    right += 1
                else:
                    errs += 1
                    if errs > allow and qp - i + pcr_slack <= len(q):
                        break
            qp -= ln
            tp -= ln
        elif op == 1:
            qp -= ln
        else:
            tp -= ln
    min_len = chimera_detect_length if chimera_detect_length else max(min_read_length // 10, 100)
    r = right
    l = left
    if right < min_len or left >= right:
        r = None
    if left < min_len or right >= left:
        l = None
    return (r, l) if a["rev"] else (l, r)


def detect_and_filter_chimeras(cons, aligner, **kw):
    """-> (kept list, removed debug ids)"""
    n = len(cons)
    chim = []
    for q in range(n):
        lefts = []
        rights = []
        qs = cons[q]["decompressed"]
        for r in range(n):
            if r == q or cons[r]["depth"] <= cons[q]["depth"] * 3:
                continue
            a = aligner(qs, cons[r]["decompressed"])
            if a is None:
                continue
            qq = qs.translate(_COMP)[::-1] if a["rev"] else qs
            l, rr = calculate_match_lengths(a, qq, cons[r]["decompressed"], **kw)
            if l is not None:
                lefts.append((r, l))
            if rr is not None:
                rights.append((r, rr))
        hit = False
        for (lr, ll) in lefts:
            for (rr, rl) in rights:
                if lr != rr:
                    ps = 0.0                                   # similarities.get(..).unwrap_or(0.0): the lookup never hits
                    cov = (ll + rl) / len(qs)
                    if cov >= min(0.9 * max(ps, 0.7), 0.8) and (cov < 1.5 or (ps < 0.99 and cov < 1.8)):
                        hit = True
                        break
        if hit:
            chim.append(q)
    return [c for i, c in enumerate(cons) if i not in chim], [cons[i]["id"] for i in chim]
