"""Constants shared with the reference (nsol/definitions.py:11)."""
EPS = 1e-10
