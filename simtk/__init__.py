"""`from simtk import openmm, unit` / `from simtk.openmm import app` -- the import lines every AtomsMM script starts
with (/root/reference/tests/test_respa_forces.py:4-6).  OpenMM itself is not rebuilt: these names resolve to
`atomsmm_amd.openmm` (an OpenMM-shaped object model whose Context is the hand-written HIP path) and
`atomsmm_amd.unit`.  Only put this directory on `sys.path` where no real OpenMM is meant to be used."""
import sys

from atomsmm_amd import openmm, unit  # noqa: F401

sys.modules[__name__ + '.openmm'] = openmm
sys.modules[__name__ + '.openmm.app'] = openmm.app
sys.modules[__name__ + '.unit'] = unit
