"""Deterministic, counter-based parameter initialisation.

The reference initialises weights with torch's global RNG (seed 1013,
/root/reference/main/ranker.py:49), which cannot be reproduced bit-for-bit on
another machine without shipping the tensors.  Parity fixtures therefore use a
counter-based generator (splitmix64 -> uniform) keyed by the *state-dict key*:
the golden generator (tests/golden/generate.py) loads these values into the
real reference model, and the GPU box regenerates exactly the same values for
the HIP path from the key names alone.
"""
import numpy as np
import torch

_MASK = np.uint64(0xFFFFFFFFFFFFFFFF)


def _fnv1a64(text):
    h = 0xCBF29CE484222325
    for ch in text.encode("utf-8"):
        h ^= ch
        h = (h * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
    return h


def splitmix_uniform(seed, n):
    """n floats in [0,1) from splitmix64 counters seed+1 .. seed+n (float64 -> exact in fp32 grid of 2^-24)."""
    with np.errstate(over="ignore"):
        idx = np.arange(1, n + 1, dtype=np.uint64)
        x = (np.uint64(seed) + idx * np.uint64(0x9E3779B97F4A7C15)) & _MASK
        z = (x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9) & _MASK
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB) & _MASK
        z = z ^ (z >> np.uint64(31))
    return (z >> np.uint64(40)).astype(np.float64) / float(1 << 24)


def det_tensor(key, shape, seed=1013, scale=None):
    """Uniform(-a, a) tensor for state-dict entry `key`; a = scale or 1/sqrt(fan_in)-like default."""
    shape = tuple(int(s) for s in shape)
    n = int(np.prod(shape)) if len(shape) else 1
    u = splitmix_uniform(_fnv1a64(key) ^ (seed * 0x2545F4914F6CDD1D & 0xFFFFFFFFFFFFFFFF), n)
    if scale is None:
        if "emb_luts" in key:
            scale = 0.5
        elif len(shape) >= 2:
            fan_in = int(np.prod(shape[1:]))
            scale = 2.0 / np.sqrt(max(fan_in, 1))  # gain 2: keeps activations O(1) so errors stay visible
        elif key.endswith("alpha"):
            scale = 1.0
        else:
            scale = 0.3
    v = ((2.0 * u - 1.0) * scale).astype(np.float32).reshape(shape)
    return torch.from_numpy(v)


def det_state_dict(shapes, seed=1013, pad_row_keys=("emb_luts",)):
    """Build a full state dict {key: tensor} from {key: shape}; embedding PAD row (index 0) is zeroed
    (nn.Embedding(padding_idx=PAD), /root/reference/neuroir/modules/embeddings.py:166-167)."""
    out = {}
    for key, shape in shapes.items():
        t = det_tensor(key, shape, seed)
        if any(p in key for p in pad_row_keys) and t.dim() == 2:
            t[0].zero_()
        out[key] = t
    return out


def fill_module_(module, seed=1013):
    """In-place deterministic init of every parameter/buffer of `module` by its state-dict key."""
    sd = module.state_dict()
    new = det_state_dict({k: v.shape for k, v in sd.items()}, seed)
    module.load_state_dict(new)
    return module
