"""TEST INFRASTRUCTURE (CPU oracle; never imported by the product path).

Restatement of the reference's permutation-invariant loss, src/loss.py:58-100
(UtterenceBaasedPermutationInvariantTraining): the S x S matrix of BATCH-mean losses of (estimated speaker i, target speaker j)
is filled without gradient (:67-71), the permutation with the smallest sum wins -- itertools.permutations order, strict '<'
(:73-86) -- and the value is the mean over the matched pairs of loss_function(enhance[:, i], target[:, j]) (:88-96).
Pinned by tests/golden/pit_cases.npz (oracle/gen_golden_pit.py runs the imported reference function)."""
from itertools import permutations

import torch


def pit(enhance, target, loss_function):
    assert enhance.shape == target.shape
    s = enhance.shape[1]
    with torch.no_grad():
        m = torch.zeros(s, s)
        for i in range(s):
            for j in range(s):
                m[i, j] = loss_function(enhance[:, i], target[:, j])
    comb, lmin = None, 1e9
    for pe in permutations(range(s)):
        l = sum(m[pe[j], j] for j in range(s))
        if lmin > l:
            comb, lmin = [(pe[j], j) for j in range(s)], l
    loss = sum(loss_function(enhance[:, i], target[:, j]) for i, j in comb) / s
    return loss, comb, m
