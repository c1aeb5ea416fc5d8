"""context_attentive_ir_amd -- MI355X-native encode-and-rank hot path of neuroir
(wasiahmad/context_attentive_ir): Python host code with the reference's model-class / args API over
hand-written HIP kernels for gfx950 behind a C-ABI (include/neuroir_hip.h).  No CPU fallback."""
__version__ = "0.1.0"
