"""Synthetic periodic boxes of the sizes BASELINE.json names."""
from .synthetic import build_c5_system, lj_fluid, solvated_chain, system_from_arrays, tip3p_box  # noqa: F401
