"""MI355X-native RawFormer inference path (drop-in for the reference's ``RawFormer.forward``).

``RawFormer`` and the operator functions in :mod:`.ops` call hand-written gfx950 kernels
through the C ABI declared in ``include/rawformer_hip.h``; :mod:`.synth` generates the
deterministic synthetic weights / Bayer frames used by the tests and the benchmark.
Importing the package does not load the HIP library; the first call that needs it does and
raises if it has not been built (``python -m bayer_low_light_image_enhancement_amd.build``).
"""
from .model import RawFormer, canonical_key  # noqa: F401
from . import ops, synth  # noqa: F401

__all__ = ["RawFormer", "canonical_key", "ops", "synth"]
