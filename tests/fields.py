"""Seeded synthetic states for the parity tests: now part of the package (bench.py and tools/ use them too)."""
from gfdl_atmos_cubed_sphere_amd.synthetic import SEED, smooth_state  # noqa: F401
