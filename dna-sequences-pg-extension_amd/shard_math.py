"""Chunk and owner arithmetic of the multi-GPU count (no torch, no GPU): shared by sharded.py and the tests."""

OWNER_BITS = 10     # the level-0 digit: top min(2k, 10) key bits


def owner_key_range(k, owner, world):
    """[key_min, key_max] of the keys owner `owner` holds: digits d with (d * world) >> bits == owner."""
    bits = min(2 * k, OWNER_BITS)
    R = 1 << bits
    d_lo = (owner * R + world - 1) // world
    d_hi = ((owner + 1) * R + world - 1) // world       # exclusive
    shift = 2 * k - bits
    if d_hi <= d_lo:
        return 0, 0
    return d_lo << shift, (d_hi << shift) - 1


def word_chunks(n_bases, world):
    """Equal word chunks of the packed sequence: rank r is resident with words [r*per, (r+1)*per).
    Returns (per_words, [(word_lo, n_bases_in_chunk)] per rank)."""
    n_words = (n_bases + 31) // 32
    per = (n_words + world - 1) // world
    out = []
    for r in range(world):
        lo = min(r * per, n_words)
        hi = min((r + 1) * per, n_words)
        out.append((lo, max(min(hi * 32, n_bases) - lo * 32, 0)))
    return per, out


def shard_ranges(n_bases, k, world):
    """Position ranges of owned k-mer starts per rank for the key-exchange variant, cut on 32-base
    (word) boundaries.  Returns (first_kmer, n_kmers, base_lo, base_hi) per rank where
    [base_lo, base_hi) are the bases the rank holds (its range plus the k-1 base halo)."""
    n_kmers = n_bases - k + 1 if n_bases >= k else 0
    words = (n_kmers + 31) // 32
    per = (words + world - 1) // world
    out = []
    for r in range(world):
        lo = min(r * per * 32, n_kmers)
        hi = min((r + 1) * per * 32, n_kmers)
        base_lo = lo if hi > lo else (lo // 32) * 32      # empty shard: any word-aligned start
        base_hi = min(hi + k - 1, n_bases) if hi > lo else base_lo
        out.append((lo, hi - lo, base_lo, base_hi))
    return out


def bucket_owner_ranges(n_buckets, world):
    """Coarse buckets of the record exchange per owner: owner o holds buckets [lo, hi) with (b * world) // n_buckets == o
    (contiguous, so a rank's records, which are grouped by bucket, are already grouped by owner)."""
    out = []
    for o in range(world):
        lo = (o * n_buckets + world - 1) // world
        hi = ((o + 1) * n_buckets + world - 1) // world
        out.append((lo, max(hi, lo)))
    return out


def bucket_owner_ranges_weighted(weights, world):
    """The same contiguous split, balanced by weight (records per bucket, summed over all ranks): owner o's range ends at
    the bucket where the running total comes closest to (o + 1) / world of the whole.  Every rank computes it from the
    same all-gathered counts.  Falls back to the even split when there is nothing to weigh."""
    nb = len(weights)
    total = sum(int(w) for w in weights)
    if total == 0 or world == 1:
        return bucket_owner_ranges(nb, world)
    cuts, run, b = [0], 0, 0
    for o in range(1, world):
        target = total * o / world
        while b < nb and run + int(weights[b]) / 2 <= target:      # (the bucket goes to the side its middle falls on)
            run += int(weights[b])
            b += 1
        cuts.append(b)
    cuts.append(nb)
    return [(cuts[o], max(cuts[o + 1], cuts[o])) for o in range(world)]


def bucket_group_cuts(weights, lo, hi, parts):
    """An owner's buckets [lo, hi) cut into `parts` groups for the pipelined exchange: the groups grow geometrically
    (1 : 3 : 9 ... by records), so that the first one lands -- and its counting starts -- after a small share of the
    transfer (the rule of dnagpu_count_multi_unordered).  Returns parts + 1 cut points."""
    parts = max(1, int(parts))
    tot = sum(int(weights[b]) for b in range(lo, hi))
    wsum = sum(3 ** p for p in range(parts))
    cuts, run, b, acc = [lo], 0, lo, 0
    for p in range(1, parts):
        acc += 3 ** (p - 1)
        target = tot * acc / wsum
        while b < hi and run + int(weights[b]) / 2 <= target:
            run += int(weights[b])
            b += 1
        cuts.append(b)
    cuts.append(hi)
    return cuts
