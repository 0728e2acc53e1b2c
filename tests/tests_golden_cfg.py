"""The scene configuration of the golden fixtures (shared by make_golden.py and the tests)."""
from webdgs_amd import synth

GOLDEN_CFG = synth.SceneConfig(7, 700, 80, 56, 3, 90.0, 0.02, "golden")
