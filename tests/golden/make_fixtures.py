#!/usr/bin/env python3
"""Generate the committed golden fixtures under tests/golden/ (run in the BUILD container only).

Input  : the reference's own test DATA files  /root/reference/tests/data/<case>.{pdb,xml}
         (data, not source: coordinates + force-field tables the reference's tests load with
         app.PDBFile / app.ForceField, e.g. tests/test_respa_forces.py:14-17).  They are also copied verbatim to
         tests/golden/data/ so that the drop-in tests (tests/test_gpu_dropin.py) can start from the same two calls.
Output : tests/golden/<case>.npz  -- plain arrays in OpenMM's unit system (nm, e, kJ/mol, dalton, rad)
         that describe what `ForceField.createSystem(topology, ...)` produces: per-particle (charge, sigma,
         epsilon, mass), harmonic bonds/angles, periodic torsions and the NonbondedForce exception list
         (1-2/1-3 exclusions, scaled 1-4 pairs).  The parsing is the product's own `atomsmm_amd.openmm.app`
         (PDBFile, ForceField.describe); tests/test_host_api.py checks that it still reproduces these files.

The expected values the fixtures are checked against are the literals in the reference's tests; they
live in tests/golden/goldens.json with file:line citations.  Nothing of /root/reference is read at
test time: only the .npz/.json/.pdb/.xml files travel.

Usage:  python tests/golden/make_fixtures.py [/root/reference/tests/data]
"""
import os
import shutil
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
CASES = ['q-SPC-FW', 'hydroxyethylaminoanthraquinone-in-water', 'emim_BCN4_Jiung2014',
         'phenol-in-water', 'methane-in-water']
KEYS = ['charge', 'sigma', 'epsilon', 'mass', 'bonds', 'bond_r0', 'bond_k', 'angles', 'angle_theta0', 'angle_k', 'torsions',
        'torsion_n', 'torsion_phase', 'torsion_k', 'exc_pairs', 'exc_chargeprod', 'exc_sigma', 'exc_epsilon']


def case_arrays(pdb_path, xml_path):
    from atomsmm_amd.openmm import app
    pdb = app.PDBFile(pdb_path)
    d = app.ForceField(xml_path).describe(pdb.topology)
    atoms = list(pdb.topology.atoms())
    out = {k: d[k] for k in KEYS}
    out['positions'] = np.array([list(v) for v in pdb.positions._value])
    out['box'] = np.array(list(pdb.topology.getUnitCellDimensions()._value))
    out['residue'] = np.array([a.residue.index for a in atoms], dtype=np.int32)
    out['resname'] = np.array([a.residue.name for a in atoms])
    out['atomname'] = np.array([a.name for a in atoms])
    return out, d['n_one_four']


def build(case, datadir):
    out, n14 = case_arrays(os.path.join(datadir, case + '.pdb'), os.path.join(datadir, case + '.xml'))
    np.savez_compressed(os.path.join(HERE, case + '.npz'), **out)
    os.makedirs(os.path.join(HERE, 'data'), exist_ok=True)
    for ext in ('.pdb', '.xml'):
        shutil.copyfile(os.path.join(datadir, case + ext), os.path.join(HERE, 'data', case + ext))
    print('%-45s N=%5d bonds=%5d angles=%5d tors=%4d exceptions=%5d (1-4: %d) box=%s' % (
        case, len(out['mass']), len(out['bonds']), len(out['angles']), len(out['torsions']), len(out['exc_pairs']), n14, out['box']))


if __name__ == '__main__':
    datadir = sys.argv[1] if len(sys.argv) > 1 else '/root/reference/tests/data'
    for case in CASES:
        build(case, datadir)
