"""Inference-time parameters of the SA backbone: conv + bias + BatchNorm folding and packing into the
MFMA fragment layout of csrc/mlp.hip.

The reference keeps these as TF variables named <scope>/conv<i>_<j>/{weights,biases} and
<scope>/conv<i>_<j>/bn/{gamma,beta,moving_mean,moving_variance} (lib/utils/layers_util.py:175,
lib/utils/tf_util.py:96,111,439-442); VariableStore plays the role of the TF variable scope: layer
code asks for a scope name and gets device-resident, BN-folded, fragment-packed weights back.
"""
import numpy as np
import torch

BN_EPS = 1e-3  # tf.contrib.layers.batch_norm default epsilon (tf_util.py:424-444)


def fold_conv_bn(params, scope, bn=True):
    """(W'[cin,cout], b'[cout]) fp32 with W' = W*s, b' = (b-mean)*s+beta, s = gamma/sqrt(var+eps)."""
    w = np.asarray(params[scope + "/weights"], np.float64)
    w = w.reshape(-1, w.shape[-1])
    bias = np.asarray(params[scope + "/biases"], np.float64)
    if bn:
        g = np.asarray(params[scope + "/bn/gamma"], np.float64)
        beta = np.asarray(params[scope + "/bn/beta"], np.float64)
        mu = np.asarray(params[scope + "/bn/moving_mean"], np.float64)
        var = np.asarray(params[scope + "/bn/moving_variance"], np.float64)
        s = g / np.sqrt(var + BN_EPS)
        w = w * s[None, :]
        bias = (bias - mu) * s + beta
    return w.astype(np.float32), bias.astype(np.float32)


def bf16_rne(x):
    """fp32 array -> bf16 bit patterns (uint16), round to nearest even (finite inputs)."""
    u = np.ascontiguousarray(x, np.float32).view(np.uint32).astype(np.uint64)
    u = u + 0x7FFF + ((u >> 16) & 1)
    return ((u >> 16) & 0xFFFF).astype(np.uint16)


def bf16_to_f32(h):
    return (h.astype(np.uint32) << 16).view(np.float32)


def pack_layer(w, bias):
    """Fragment-order hi/lo bf16 packing of W'[K,N] (see csrc/mlp.hip header) + zero-padded bias.

    Returns (packed uint16 [NT, KS, 2, 64, 8], bias fp32 [NT*32])."""
    K, N = w.shape
    KS, NT = (K + 15) // 16, (N + 31) // 32
    wp = np.zeros((KS * 16, NT * 32), np.float32)
    wp[:K, :N] = w
    hi = bf16_rne(wp)
    lo = bf16_rne(wp - bf16_to_f32(hi))
    arr = np.zeros((NT, KS, 2, 64, 8), np.uint16)
    for plane, src in enumerate((hi, lo)):
        s = src.reshape(KS, 2, 8, NT, 32)                      # [ks][half][e][ct][col]
        arr[:, :, plane] = s.transpose(3, 0, 1, 4, 2).reshape(NT, KS, 64, 8)   # lane = 32*half + col
    bp = np.zeros(NT * 32, np.float32)
    bp[:N] = bias
    return arr, bp


class PackedLayer:
    __slots__ = ("K", "N", "w", "bias")

    def __init__(self, w, bias, device, _packed=None):
        self.K, self.N = int(w.shape[0]), int(w.shape[1])
        if _packed is not None:                      # views into a buffer shared with the other layers of a scale
            self.w, self.bias = _packed
            return
        arr, bp = pack_layer(w, bias)
        self.w = torch.from_numpy(arr.view(np.int16).reshape(-1)).to(device)
        self.bias = torch.from_numpy(bp).to(device)


def pack_scale(ws, bs, device):
    """The layers of one MLP scale packed back to back in ONE device buffer (layer l+1 starts where layer l
    ends): the streamed-weight kernel of csrc/mlp_rowwave.hip walks them as a single linear stream.  Every
    kernel accepts these layers; separately allocated PackedLayers simply never take the streamed path."""
    packed = [pack_layer(w, b) for w, b in zip(ws, bs)]
    flat = np.concatenate([arr.view(np.int16).reshape(-1) for arr, _ in packed])
    buf = torch.from_numpy(flat).to(device)
    out, off = [], 0
    for (arr, bp), w in zip(packed, ws):
        n = arr.size
        out.append(PackedLayer(w, None, device, _packed=(buf[off:off + n], torch.from_numpy(bp).to(device))))
        off += n
    return out


class VariableStore:
    """name -> numpy parameter dict plus a cache of folded + packed layers on `device`."""

    def __init__(self, params, device):
        self.params = params
        self.device = torch.device(device)
        self._cache = {}

    @classmethod
    def from_checkpoint(cls, path, device, verify=True):
        """Variables of a TensorFlow checkpoint written by the reference's trainer (`tf.train.Saver`,
        lib/core/trainer.py:157-174): `path` is a checkpoint prefix (`.../model-80000`) or a directory holding a
        `checkpoint` state file.  Read without TensorFlow (tf_checkpoint.py); optimizer slots are dropped."""
        import os
        from . import tf_checkpoint
        if os.path.isdir(path):
            prefix = tf_checkpoint.latest_checkpoint(path)
            if prefix is None:
                raise FileNotFoundError("no `checkpoint` state file in %s" % path)
        else:
            prefix = path
        return cls(tf_checkpoint.load_checkpoint(prefix, verify=verify), device)

    def layer(self, scope, bn=True):
        key = (scope, bool(bn))
        if key not in self._cache:
            w, b = fold_conv_bn(self.params, scope, bn)
            self._cache[key] = PackedLayer(w, b, self.device)
        return self._cache[key]

    def scale(self, scopes, bn=True):
        """The conv layers of one MLP scale, packed contiguously (pack_scale)."""
        key = (tuple(scopes), bool(bn))
        if key not in self._cache:
            folded = [fold_conv_bn(self.params, sc, bn) for sc in scopes]
            self._cache[key] = pack_scale([w for w, _ in folded], [b for _, b in folded], self.device)
        return self._cache[key]


_DEFAULT_STORE = None


def set_default_variables(store):
    """The analogue of building the TF graph under a variable scope: layer functions called without an
    explicit `variables=` argument read their weights from this store."""
    global _DEFAULT_STORE
    _DEFAULT_STORE = store


def default_variables():
    if _DEFAULT_STORE is None:
        raise RuntimeError("no VariableStore set: call set_default_variables(VariableStore(params, device))")
    return _DEFAULT_STORE
