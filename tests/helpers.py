"""Shared test helpers (test infrastructure)."""
import itertools

import numpy as np


def solvation_respa_inputs(h, lambda_coul):
    """What SolvationSystem (systems.py:261-313) + RESPASystem (systems.py:62-82) hand to the near force
    and to the exceptions CustomBondForce for the HEAQ case (solute = residue 'aaa')."""
    solute = np.where(h['resname'] == 'aaa')[0]
    q = h['charge'].copy(); s = h['sigma'].copy(); e = h['epsilon'].copy()
    q[solute] *= lambda_coul     # charge offset: 0 + lambda_coul*q  (systems.py:303,311)
    s[solute] = 0.0; e[solute] = 0.0
    have = {(int(a), int(b)) for a, b in h['exc_pairs']}
    pairs = [tuple(p) for p in h['exc_pairs']]
    qq = list(h['exc_chargeprod']); sg = list(h['exc_sigma']); ep = list(h['exc_epsilon'])
    for i, j in itertools.combinations([int(a) for a in solute], 2):
        if (i, j) not in have:
            pairs.append((i, j)); qq.append(h['charge'][i] * h['charge'][j])
            sg.append(0.5 * (h['sigma'][i] + h['sigma'][j])); ep.append(np.sqrt(h['epsilon'][i] * h['epsilon'][j]))
    return q, s, e, np.array(pairs, dtype=np.int32), np.array(qq), np.array(sg), np.array(ep)
