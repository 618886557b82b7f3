"""Closed-form, RNG-free recipe for weights and inputs used by the golden vectors.

Both the capture script (``make_golden.py``, which runs the reference in the build
container) and the parity tests (which run the oracle / the HIP path anywhere) call
these functions, so the tensors are bit-identical on every machine without shipping
them: values are evaluated in numpy float64 and rounded once to float32.
"""
import zlib

import numpy as np


def _phase(name: str) -> float:
    # crc32 is stable across processes/platforms (unlike hash()).
    return (zlib.crc32(name.encode()) % 10007) * 0.001


def wave(name: str, shape, scale: float = 1.0, offset: float = 0.0) -> np.ndarray:
    """v[i] = offset + scale * sin(0.37 * i + phase(name)) (float64 -> float32)."""
    n = int(np.prod(shape)) if len(shape) else 1
    i = np.arange(n, dtype=np.float64)
    v = offset + scale * np.sin(0.37 * i + _phase(name)) * np.cos(0.011 * i + 1.3 * _phase(name) + 0.5)
    return v.astype(np.float32).reshape(shape)


def param_value(name: str, shape) -> np.ndarray:
    """Deterministic value for a parameter / buffer, keyed by its state-dict name."""
    shape = tuple(shape)
    leaf = name.split('.')[-1]
    if leaf == 'num_batches_tracked':
        return np.zeros(shape, dtype=np.int64)
    if leaf == 'running_mean':
        return np.zeros(shape, dtype=np.float32)
    if leaf == 'running_var':
        return np.ones(shape, dtype=np.float32)
    if leaf in ('pos_embedding', 'space_token', 'temporal_token'):
        return wave(name, shape, 1.0)
    if leaf == 'bias':
        return wave(name, shape, 0.1)
    if leaf == 'weight' and len(shape) == 1:          # LayerNorm / BatchNorm gain
        return wave(name, shape, 0.2, 1.0)
    if leaf == 'weight':
        fan_in = int(np.prod(shape[1:]))
        # sin*cos has rms ~0.5, so 2.4/sqrt(fan_in) keeps activations O(1)
        return wave(name, shape, 2.4 / np.sqrt(max(fan_in, 1)))
    return wave(name, shape, 1.0)


def fill_state_dict(sd: dict, prefix: str = '') -> dict:
    """Return {name: np.ndarray} for every entry of a torch state_dict (shapes only are read)."""
    return {k: param_value(prefix + k, tuple(v.shape)) for k, v in sd.items()}


def input_value(name: str, shape, scale: float = 1.0) -> np.ndarray:
    return wave('input:' + name, tuple(shape), scale)


# ---------------------------------------------------------------------------------------------
# Pseudo-random recipe (round 2): the sin-wave weights above make a deep BatchNorm chain amplify fp32 noise ~2x per
# layer (fine for the 6-layer entry flow, hopeless for the 40-layer full Xception), so the goldens of deep stacks use
# default-init-like uniform weights from numpy's PCG64 (integer arithmetic: identical on every machine with this numpy).
def _rng(name: str):
    return np.random.Generator(np.random.PCG64(zlib.crc32(name.encode())))


def rand_param_value(name: str, shape) -> np.ndarray:
    shape = tuple(shape)
    leaf = name.split('.')[-1]
    if leaf in ('num_batches_tracked', 'running_mean', 'running_var'):
        return param_value(name, shape)
    g = _rng('param:' + name)
    if leaf == 'weight' and len(shape) == 1:           # BatchNorm / LayerNorm gain
        return (1.0 + 0.1 * g.uniform(-1.0, 1.0, shape)).astype(np.float32)
    if leaf == 'bias':
        return (0.1 * g.uniform(-1.0, 1.0, shape)).astype(np.float32)
    if leaf in ('pos_embedding', 'space_token', 'temporal_token', 'cls_token'):
        return g.standard_normal(shape).astype(np.float32)
    fan_in = int(np.prod(shape[1:])) if len(shape) > 1 else int(shape[0])
    b = np.sqrt(3.0 / max(fan_in, 1))                  # unit-gain uniform: activations stay O(1) through ReLU + BN
    return g.uniform(-b, b, shape).astype(np.float32)


def rand_fill_state_dict(sd: dict, prefix: str = '') -> dict:
    return {k: rand_param_value(prefix + k, tuple(v.shape)) for k, v in sd.items()}


def rand_input_value(name: str, shape, scale: float = 1.0) -> np.ndarray:
    return (scale * _rng('input:' + name).standard_normal(tuple(shape))).astype(np.float32)
