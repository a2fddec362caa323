"""TEST INFRASTRUCTURE ONLY (see oracle/savont_oracle.h): plain-Python restatement of the host statistics of Stage 4,
following the reference statement by statement.  Only tests/ may import this.

  estimate_quality_error_rates   src/alignment.rs:663-786
  log_sum_exp                    src/alignment.rs:789-795
  analyze_pileup_consensuses     src/alignment.rs:864-1155
  lq_criteria                    src/alignment.rs:1157-1160
  decompress                     src/types.rs:212-217, src/utils.rs:114-130
  median_hp_lengths (--use-hpc)  src/alignment.rs:586-656

A pile-up is a list (one per consensus position) of entries (kind, base, qual) with kind 0 = Base, 1 = Deletion,
2 = Insertion (first inserted base / quality), in the push order of src/alignment.rs:527-571.
Parity pinning: the reference has no unit tests or golden vectors for these functions and cannot be executed here (Rust);
the restatement is pinned only by reading -- "parity unpinned" for Stage 4."""
import math

DEFAULT_ERR_RATE = 0.02          # src/constants.rs:35


def estimate_quality_error_rates(pileups, consensuses, top_frac):
    """pileups[i] = list of columns; consensuses[i] = dict(seq=bytes, depth=int) -> {quality: rate}"""
    depths = sorted(((i, c["depth"]) for i, c in enumerate(consensuses)), key=lambda x: -x[1])      # :674 (stable)
    take = int(round_half_away(top_frac * len(depths)))                                             # :678 f64::round
    stats = {}
    for ci, _ in depths[:take]:
        if ci >= len(pileups):
            continue
        seq = consensuses[ci]["seq"]
        for pos, col in enumerate(pileups[ci]):
            ref = seq[pos]
            total = len(col)
            err = sum(1 for (k, b, q) in col if k != 0 or b != ref)                                  # :701-719
            if total > 0 and err / total < 0.05:                                                    # :722-724
                for (k, b, q) in col:
                    if k == 0:
                        e = stats.setdefault(q, [1, 1])                                             # prior (1,1) :687,:728
                        e[1] += 1
                        if b != ref:
                            e[0] += 1
    return {q: (e / t if t > 0 else 0.0) for q, (e, t) in stats.items()}                            # :782-785


def round_half_away(x):
    return math.floor(x + 0.5) if x >= 0 else -math.floor(-x + 0.5)


def log_sum_exp(a, b):
    mx = max(a, b)
    if math.isinf(mx) and mx < 0:
        return -math.inf
    return mx + math.log(math.exp(a - mx) + math.exp(b - mx))


def median_hp_lengths(hp_columns):
    """hp_columns[pos] = run lengths of the Base entries of that column (push order) -> consensus.hp_lengths (:586-625, :652-656)"""
    out = []
    for hp in hp_columns:
        if not hp:
            out.append(1)                                                                           # :623-625
            continue
        srt = sorted(hp)                                                                            # :611-612
        mid = len(srt) // 2
        out.append(((srt[mid - 1] + srt[mid]) // 2) & 0xFF if len(srt) % 2 == 0 else srt[mid])      # :614-618 (u16 sum, as u8)
    return out


def decompress(seq, hp_lengths=None):
    """ConsensusSequence::decompress: homopolymer_decompress (a length mismatch returns the sequence unchanged, src/utils.rs:115-118), then
    the leading / trailing N are cut (src/types.rs:214-216; all N: everything stays)"""
    if hp_lengths is not None and len(hp_lengths) == len(seq):
        s = b"".join(bytes([b]) * int(l) for b, l in zip(seq, hp_lengths))
    else:
        s = bytes(seq)
    a = 0
    b = len(s)
    while a < b and s[a] == ord("N"):
        a += 1
    while b > a and s[b - 1] == ord("N"):
        b -= 1
    if a >= b:
        a, b = 0, len(s)
    return s[a:b]


def analyze_pileup_consensuses(pileups, consensuses, qmap, min_cluster_size=12, posterior_threshold_ln=30.0,
                               mask_low_quality=False, n_depth_cutoff=250, hp_lengths=None):
    """-> (kept, low): lists of dict(seq (masked, bytes), depth, id, low_quality_positions, decompressed); hp_lengths (--use-hpc): per
    consensus the list median_hp_lengths gave (main.rs:103-110 decompresses with them)"""
    bad_length_threshold = 100                                                                      # :872
    min_coverage_abs = max(min_cluster_size * 3 // 4, 2)                                            # :873
    rate = lambda q: qmap.get(q, DEFAULT_ERR_RATE)
    indel_err = rate(48)                                                                            # :874-879
    out = []
    for ci, cons in enumerate(consensuses):
        seq = bytearray(cons["seq"])
        cols = pileups[ci]
        low_conf = []
        lo, hi = 0, len(cols)
        if cols:
            min_cov = max(max(len(c) for c in cols) // 3, min_coverage_abs)                         # :894
            start, end = 0, len(cols)
            for i, c in enumerate(cols):                                                            # :905-914
                if len(c) >= min_cov:
                    start = i
                    break
            for i in range(len(cols) - 1, -1, -1):                                                  # :917-926
                if len(cols[i]) >= min_cov:
                    end = i + 1
                    break
            if start < end:                                                                         # :928-938
                lo, hi = start, end
                thr = min(posterior_threshold_ln, float(min_cluster_size * 3))                      # :995
                for p in range(start, end):
                    ref = seq[p]
                    lr = 0.0
                    ln = 0.0
                    for (k, b, q) in cols[p]:
                        if k == 0:                                                                  # :954-967
                            er = rate(q)
                            acc = 1.0 - er
                            if b == ref:
                                lr += math.log(acc); ln += math.log(er)
                            else:
                                lr += math.log(er); ln += math.log(acc)
                        elif k == 1:                                                                # :968-972
                            lr += math.log(indel_err); ln += math.log(1.0 - indel_err)
                        else:                                                                       # :973-985
                            er = rate(q)
                            ln += math.log(1.0 - er); lr += math.log(er)
                    alt = ln - log_sum_exp(lr, ln)                                                  # :991-992
                    if alt > -thr:                                                                  # :996
                        low_conf.append(p)
            # second loop :1080-1128 (an untrimmed pile-up has left_start 0 / right_end len)
            left_start, right_end = lo, hi
            start_polish = bad_length_threshold + left_start
            end_polish = max(right_end - bad_length_threshold, 0)
            left = [p for p in low_conf if p < start_polish]
            right = [p for p in low_conf if p >= end_polish]
            lc_left = max(left) if left else left_start
            lc_right = min(right) if right else right_end
            for p in range(0, min(lc_left, len(seq))):
                seq[p] = ord("N")
            for p in range(lc_right, len(seq)):
                seq[p] = ord("N")
            lqp = []
            for p in low_conf:
                if mask_low_quality:
                    seq[p] = ord("N")
                if lc_left < p < lc_right:
                    lqp.append(p)
        else:
            lqp = []
        s = bytes(seq)
        out.append(dict(seq=s, depth=cons["depth"], id=cons["id"], low_quality_positions=lqp, hp_lengths=hp_lengths[ci] if hp_lengths is not None else None,
                        decompressed=decompress(s, hp_lengths[ci] if hp_lengths is not None else None)))
    lq = lambda c: len(c["low_quality_positions"]) > 0 and c["depth"] // (len(c["low_quality_positions"]) ** 2) < n_depth_cutoff   # :1157-1160
    return [c for c in out if not lq(c)], [c for c in out if lq(c)]
