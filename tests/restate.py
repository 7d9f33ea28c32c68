"""Host-side restatements the GPU tests replay moves with; each is itself checked against fixtures recorded from the
reference's own Python (tests/test_reference_callers.py)."""
import numpy as np


def ref_select_action(visit_counts, legal, u, temperature=1.0, deterministic=False):
    """core/utils.py:280-295: illegal actions' counts zeroed, p ~ count^(1/T), arg-max or np.random.choice -- whose
    algorithm (numpy mtrand: cdf = cumsum(p); cdf /= cdf[-1]; searchsorted(u, 'right')) is applied to a GIVEN uniform u --
    and scipy.stats.entropy(p, base=2).  Returns (action, entropy, masked counts)."""
    visit_counts = list(visit_counts)
    for i in range(len(legal)):
        if legal[i] == 0 and visit_counts[i] >= 1:
            visit_counts[i] = 0
    probs = [float(v) ** (1.0 / temperature) for v in visit_counts]
    total = sum(probs)
    probs = [x / total for x in probs]
    if deterministic:
        a = int(np.argmax(visit_counts))
    else:
        cdf = np.cumsum(np.array(probs, dtype=np.float64))
        cdf /= cdf[-1]
        a = int(cdf.searchsorted(u, side="right"))
    pk = np.array(probs) / np.sum(probs)
    ent = -np.sum(np.where(pk > 0, pk * np.log(np.where(pk > 0, pk, 1.0)), 0.0)) / np.log(2.0)
    return a, ent, visit_counts
