"""webdgs_amd -- MI355X-native hot path of krispy-kenay/WebDGS (differentiable 3D Gaussian splatting).

``csrc/`` holds the HIP kernels and the C ABI (``include/webdgs.h`` -> ``lib/libwebdgs_hip.so``); ``ops`` mirrors the
reference's operator classes over that ABI; ``trainer`` mirrors ``src/trainer.ts`` and ``viewer`` ``src/viewer.ts``; ``loaders``
and ``images`` are the PLY / COLMAP / camera / image ingest of ``src/utils``; ``parallel`` adds view-sharded data parallelism over
RCCL; ``synth`` fabricates the seeded scenes of SURVEY.md section 8(d).
"""
__all__ = ["ops", "trainer", "viewer", "loaders", "images", "parallel", "synth"]
