"""hanabizero_amd.mlp_sync -- which synchronisation every job of a fused-MLP job table needs, and a proof that it has it.

The 16 waves x 2 tiles kernel (csrc/hz_mlp_dev.h) runs a table of passes; in pass p wave w executes job (p, w) or idles.
Jobs read and write column ranges of the workgroup's row image in LDS.  Between passes there used to be a workgroup
barrier wherever a layer boundary was; include/hz_mlp.h now has two cheaper forms for the 16 x 2 shape:

  HZ_MLP_BLOCKWISE  a job that reads a full-width layer's output waits, per block of 128 input columns, for the four waves
                    that produce that block (decided in model.py::_FusedChain.add_dense);
  HZ_MLP_WAITS      a job waits, before it starts, for up to four (pass, group of four waves) arrival counters -- exactly
                    the producers and earlier readers its own columns depend on (decided HERE, for the passes that would
                    otherwise begin with a barrier).

`plan(passes, ...)` derives the waits from the jobs' column ranges (read-after-write, write-after-read, write-after-write
against every earlier job), keeps the barrier where four tokens per job are not enough, and `verify` re-checks the final
table from scratch: every hazard pair must be ordered by the transitive closure of program order, barriers, blockwise
producers and tokens.  A table that passes cannot race; one that does not raises.  Pure Python, no torch: tests/test_model.py
runs it on the CPU.
"""

BARRIER, STORE_HIDDEN, SIGNAL, BLOCKWISE, WAITS = 4, 8, 16, 32, 64
MAX_TOKENS = 4


class Job:
    """One (pass, wave) entry: column intervals [lo, hi) it reads and writes (empty lists for an idle entry)."""

    def __init__(self, reads=(), writes=(), active=True):
        self.reads, self.writes, self.active = [r for r in reads if r[0] < r[1]], [w for w in writes if w[0] < w[1]], active
        self.tokens = []  # [(pass, group, count)]


def _overlap(a, b):
    return any(x[0] < y[1] and y[0] < x[1] for x in a for y in b)


def _closure(n_pass, waves, flags, jobs):
    """hb[(q, u)] = set of (p, w) that job (q, u) happens-before (q < p), from: program order; a barrier at the start of pass p;
    a blockwise pass p (all jobs of pass p - 1 precede every ACTIVE job of pass p -- the last block a job waits for is the last
    producer's; an idle entry of such a pass waits for nothing);
    tokens."""
    succ = {(p, w): set() for p in range(n_pass) for w in range(waves)}
    for p in range(n_pass):
        for w in range(waves):
            if p + 1 < n_pass:
                succ[(p, w)].add((p + 1, w))
            for (q, g, _c) in jobs[p][w].tokens:
                for u in range(4 * g, 4 * g + 4):
                    if jobs[q][u].active:
                        succ[(q, u)].add((p, w))
        if p > 0 and (flags[p] & BARRIER):
            for u in range(waves):
                for w in range(waves):
                    succ[(p - 1, u)].add((p, w))
        elif p > 0 and (flags[p] & BLOCKWISE):
            # only the ACTIVE entries of a blockwise pass poll their producers' counters; an idle entry waits for nothing (the
            # kernel has no barrier there and `continue`s before any poll) and gets program order only
            for u in range(waves):
                for w in range(waves):
                    if jobs[p][w].active:
                        succ[(p - 1, u)].add((p, w))
    hb = {}
    for p in reversed(range(n_pass)):  # successors only ever lie in later passes: one backward sweep closes the relation
        for w in range(waves):
            s = set()
            for n in succ[(p, w)]:
                s.add(n)
                s |= hb[n]
            hb[(p, w)] = s
    return hb


def _required(n_pass, waves, jobs, p, w):
    """Earlier jobs that must have finished before job (p, w) may run: writers of what it reads, readers and writers of what it
    writes."""
    me = jobs[p][w]
    need = []
    for q in range(p):
        for u in range(waves):
            o = jobs[q][u]
            if _overlap(o.writes, me.reads) or _overlap(o.reads, me.writes) or _overlap(o.writes, me.writes):
                need.append((q, u))
    return need


def plan(flags, jobs, waves=16):
    """flags[p]: the pass's flags as decided so far (BARRIER / BLOCKWISE / STORE_HIDDEN); jobs[p][w]: Job.  Replaces the barrier
    of every pass where each job gets by with <= MAX_TOKENS tokens; returns (flags, signal_passes).  Pass 0 follows the
    staging barrier of the kernel and needs nothing."""
    n_pass = len(jobs)
    flags = list(flags)
    for p in range(1, n_pass):
        if flags[p] & BLOCKWISE:
            continue
        # (also the passes that had no barrier of their own -- the later rows of a stage -- because they leaned on one that
        # may just have gone)
        saved = [list(j.tokens) for j in jobs[p]]
        flags[p] &= ~BARRIER
        hb = _closure(n_pass, waves, flags, jobs)
        ok = True
        for w in range(waves):
            missing = [(q, u) for (q, u) in _required(n_pass, waves, jobs, p, w) if (p, w) not in hb[(q, u)]]
            toks = set()
            latest = {}
            for (q, u) in missing:  # (program order: a wave's later job covers its earlier ones)
                latest[u] = max(q, latest.get(u, -1))
            for u, q in latest.items():  # (a wave that idles through pass q signals nothing there: take its next job before pass p)
                qa = next((x for x in range(q, p) if jobs[x][u].active), None)
                if qa is None:
                    ok = False
                    break
                toks.add((qa, u // 4))
            # (one group at several passes: the latest pass covers the earlier ones if every wave of the group is active there)
            for (qa, g) in sorted(toks):
                later = [x for (x, gg) in toks if gg == g and x > qa]
                if later and all(jobs[max(later)][u].active for u in range(4 * g, 4 * g + 4) if latest.get(u, -1) >= 0 and latest[u] <= qa):
                    toks.discard((qa, g))
            if not ok or len(toks) > MAX_TOKENS or any(q >= 16 for q, _ in toks):
                ok = False
                break
            jobs[p][w].tokens = [(q, g, sum(1 for u in range(4 * g, 4 * g + 4) if jobs[q][u].active)) for q, g in sorted(toks)]
        if ok:  # (the tokens order whole groups: check that nothing is left uncovered)
            hb = _closure(n_pass, waves, flags, jobs)
            ok = all((p, w) in hb[(q, u)] for w in range(waves) for (q, u) in _required(n_pass, waves, jobs, p, w))
        if ok:
            if any(j.tokens for j in jobs[p]):
                flags[p] |= WAITS
        else:
            for j, t in zip(jobs[p], saved):
                j.tokens = t
            flags[p] |= BARRIER
    signal = sorted({q for row in jobs for j in row for (q, _g, _c) in j.tokens})
    verify(flags, jobs, waves)
    return flags, signal


def verify(flags, jobs, waves=16):
    """Every hazard pair of the table is ordered; every token names a group with that many active jobs; jobs of one pass do not
    touch each other's columns (they run concurrently), except a job reading what it itself overwrites."""
    n_pass = len(jobs)
    hb = _closure(n_pass, waves, flags, jobs)
    for p in range(n_pass):
        for w in range(waves):
            for (q, u) in _required(n_pass, waves, jobs, p, w):
                if (p, w) not in hb[(q, u)]:
                    raise AssertionError("job (pass %d, wave %d) may run before job (pass %d, wave %d) it depends on" % (p, w, q, u))
            for (q, g, c) in jobs[p][w].tokens:
                assert q < p and q < 16 and c == sum(1 for u in range(4 * g, 4 * g + 4) if jobs[q][u].active) and 1 <= c <= 4
            assert len(jobs[p][w].tokens) <= MAX_TOKENS
            for u in range(w + 1, waves):
                a, b = jobs[p][w], jobs[p][u]
                if _overlap(a.writes, b.reads) or _overlap(a.reads, b.writes) or _overlap(a.writes, b.writes):
                    raise AssertionError("jobs (pass %d, waves %d and %d) run side by side on overlapping columns" % (p, w, u))
    return True


def pack_tokens(tokens):
    """-> (count, word): byte k of the word = pass << 4 | group << 2 | (count - 1), as csrc/hz_mlp_dev.h reads it."""
    word = 0
    for k, (q, g, c) in enumerate(tokens):
        assert 0 <= q < 16 and 0 <= g < 4 and 1 <= c <= 4
        word |= ((q << 4) | (g << 2) | (c - 1)) << (8 * k)
    return len(tokens), word - (1 << 32) if word >= (1 << 31) else word
