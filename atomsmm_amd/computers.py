"""`PressureComputer` with the interface of `atomsmm.computers.PressureComputer` (reference: src/atomsmm/computers.py):
a Context over a `ComputingSystem` that returns the atomic and molecular internal virials and pressures of a
configuration.  The pair sums run in the HIP kernels (families AMM_LJ_VIRIAL / AMM_NONBONDED, bond kinds
AMM_BOND_VIRIAL_*); the molecule bookkeeping is numpy on the host, as in the reference.
"""
import numpy as np

from . import openmm, unit
from .systems import ComputingSystem


class _MoleculeTotalizer(object):
    """Molecule membership, masses and mass fractions (computers.py:22-43), dense numpy instead of scipy.sparse."""

    def __init__(self, context, topology):
        molecules = context.getMolecules()
        self.nmols = len(molecules)
        self.natoms = sum(len(m) for m in molecules)
        self.mol_of = np.empty(self.natoms, dtype=np.int64)
        for k, atoms in enumerate(molecules):
            self.mol_of[atoms] = k
        system = context.getSystem()
        self.mass = np.array([unit.md_value(system.getParticleMass(i)) for i in range(self.natoms)])
        self.molMass = np.bincount(self.mol_of, weights=self.mass, minlength=self.nmols)
        self.massFrac = self.mass / self.molMass[self.mol_of]

    def sum_by_molecule(self, per_atom):
        """selection.dot(a): [natoms][3] -> [nmols][3]."""
        out = np.zeros((self.nmols, per_atom.shape[1]))
        np.add.at(out, self.mol_of, per_atom)
        return out

    def centre_of_mass(self, per_atom):
        """massFrac.dot(a)."""
        return self.sum_by_molecule(per_atom * self.massFrac[:, None])


class PressureComputer(openmm.Context):
    """computers.py:46-246.  `temperature` = bath temperature for the equipartition kinetic terms; None = instantaneous
    kinetic energies."""

    def __init__(self, system, topology, platform, properties=dict(), temperature=None):
        self._computing_system = ComputingSystem(system)
        super().__init__(self._computing_system, openmm.CustomIntegrator(0), platform, properties)
        self._mols = _MoleculeTotalizer(self, topology)
        self._kT = None if temperature is None else unit.MOLAR_GAS_CONSTANT_R * temperature
        self._make_obsolete()

    def _get_potential(self, groups):
        return self.getState(getEnergy=True, groups=groups).getPotentialEnergy()

    def _get_volume(self):
        box = self.getState().getPeriodicBoxVectors()
        return box[0][0] * box[1][1] * box[2][2] * unit.AVOGADRO_CONSTANT_NA

    def _make_obsolete(self):
        self._bond_virial = self._coulomb_virial = self._dispersion_virial = None
        self._molecular_kinetic_energy = None

    def get_bond_virial(self):
        if self._bond_virial is None:
            self._bond_virial = self._get_potential(self._computing_system._bonded)
        return self._bond_virial

    def get_coulomb_virial(self):
        if self._coulomb_virial is None:
            self._coulomb_virial = self._get_potential(self._computing_system._coulomb)
        return self._coulomb_virial

    def get_dispersion_virial(self):
        if self._dispersion_virial is None:
            self._dispersion_virial = self._get_potential(self._computing_system._dispersion)
        return self._dispersion_virial

    def get_atomic_virial(self):
        """W = -sum r_ij E'(r_ij) over van der Waals, Coulomb and bond-stretching interactions (computers.py:140-160)."""
        return self.get_bond_virial() + self.get_coulomb_virial() + self.get_dispersion_virial()

    def get_atomic_pressure(self):
        """P = (2K + W)/(3V) (computers.py:104-138)."""
        if self._kT is None:
            velocities = self.getState(getVelocities=True).getVelocities(asNumpy=True)._value
            dNkT = float(np.sum(self._mols.mass * np.sum(velocities ** 2, axis=1))) * unit.kilojoules_per_mole
        else:
            dNkT = 3 * self._mols.natoms * self._kT
        pressure = (dNkT + self.get_atomic_virial()) / (3 * self._get_volume())
        return pressure.in_units_of(unit.atmospheres)

    def get_molecular_kinetic_energy(self):
        if self._molecular_kinetic_energy is None:
            velocities = self.getState(getVelocities=True).getVelocities(asNumpy=True)._value
            vcm = self._mols.centre_of_mass(velocities)
            self._molecular_kinetic_energy = 0.5 * float(np.sum(self._mols.molMass * np.sum(vcm ** 2, axis=1))) * unit.kilojoules_per_mole
        return self._molecular_kinetic_energy

    def get_molecular_virial(self, forces):
        """W_mol = W - sum_i (r_i - r_i^cm) . F_i (Hunenberger 2002; computers.py:199-228)."""
        f = np.asarray(forces.value_in_unit(unit.kilojoules_per_mole / unit.nanometers))
        r = self.getState(getPositions=True).getPositions(asNumpy=True)._value
        fcm = self._mols.sum_by_molecule(f)
        rcm = self._mols.centre_of_mass(r)
        W = unit.md_value(self.get_atomic_virial())
        return (W + float(np.sum(rcm * fcm)) - float(np.sum(r * f))) * unit.kilojoules_per_mole

    def get_molecular_pressure(self, forces):
        """P = (2 K_mol + W_mol)/(3V) (computers.py:171-197)."""
        if self._kT is None:
            dNkT = 2.0 * self.get_molecular_kinetic_energy()
        else:
            dNkT = 3 * self._mols.nmols * self._kT
        pressure = (dNkT + self.get_molecular_virial(forces)) / (3 * self._get_volume())
        return pressure.in_units_of(unit.atmospheres)

    def import_configuration(self, state):
        self.setPeriodicBoxVectors(*state.getPeriodicBoxVectors())
        self.setPositions(state.getPositions())
        self.setVelocities(state.getVelocities())
        self._make_obsolete()
