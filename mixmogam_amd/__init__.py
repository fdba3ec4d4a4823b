"""mixmogam_amd -- MI355X-native EMMAX mixed-model GWAS hot path behind mixmogam's call surface.

Modules mirror the reference's names: `kinship`, `linear_models`, `simulations`.
All arithmetic on the hot path runs in libmixmogam_hip.so (hand-written HIP for gfx950);
there is no CPU fallback.
"""
__version__ = "0.1.0"
