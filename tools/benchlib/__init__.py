"""Pieces of bench.py (repo root): what is not the timed step itself.  bench.py keeps the contract line -- arguments, setup, the timed loop, the
roofline bookkeeping -- and imports: common (constants, shape tables, the kernel-family rule), ranks (starting N ranks / emulating them), stages (the per-stage
and denominator passes, the secondary configurations), cpu (the host baselines)."""
