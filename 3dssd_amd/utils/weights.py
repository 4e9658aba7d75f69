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


# Operand precision of a grouped-MLP scale (csrc/mlp.hip, "Operand precision"):
#   "bf16x3"  every fp32 operand as hi + lo bf16, three MFMA passes per k-step (~5e-6 of the fp32 oracle)
#   "fp16"    one fp16 plane, one pass.  Error of the pooled output against the fp32 oracle through the three stacked
#             layers, max |y - ref| / max |ref|, emulated in numpy over 3 weight seeds x 4 frames (and confirmed on the
#             GPU for the bench seed): layer3 / layer4 scales (every contraction >= 128 wide) 4.1e-4 .. 7.1e-4, layer2
#             (67/64/96-wide) up to 9.1e-4, layer1 (4/16/32-wide) up to 8.6e-4 against the 1e-3 bar -- hence the rule
#             below: the wide scales, which hold 85 % of the MLP arithmetic, take the one-pass form, the narrow ones keep
#             the three-pass form.  MLP_PRECISION (module attribute, or VariableStore(..., precision=)) = "auto" |
#             "bf16x3" | "fp16" overrides the rule; nothing is read from the environment.
# fp16 range guards: a scale takes the fp16 form only when every folded weight fits (|w| < FP16_MAX_ABS) and no output
# channel's weights sit in the fp16 subnormal range as a whole (column maximum >= FP16_MIN_COL_MAX = 2^-10, so that
# every weight within a factor 16 of the column's largest keeps its 11 significant bits; BN folding with a tiny
# gamma / sqrt(var) can scale a whole column down -- single tiny entries of a healthy column only add an absolute
# error far below the column's scale).  Activations are converted WITHOUT saturation (v_cvt_pk_f16_f32: a value beyond 65504
# becomes inf) and an inf / NaN half -- or a NaN / inf input feature -- raises the overflow flag of the call (csrc/mlp_act.h; VariableStore.overflow, checked by
# SABackbone.raise_if_overflow and by SAPipeline tickets).
FP16_MIN_K = 128
FP16_MAX_ABS = 6.0e4
FP16_MIN_COL_MAX = 2.0 ** -10
MLP_PRECISION = "auto"


def fp16_weights_fit(w):
    """True when the folded matrix [K, N] survives fp16: no overflow, no output column living in the subnormals."""
    a = np.abs(np.asarray(w, np.float64))
    if a.size == 0:
        return True
    if float(a.max()) >= FP16_MAX_ABS:
        return False
    colmax = a.max(axis=0)
    live = colmax > 0
    return bool((colmax[live] >= FP16_MIN_COL_MAX).all())


def scale_precision(ws, mode=None):
    """Precision of one scale from its folded weight matrices [K, N] (rule above); fp16 is refused when a weight
    would overflow or underflow it."""
    mode = mode or MLP_PRECISION
    if mode == "bf16x3":
        return "bf16x3"
    fits = all(fp16_weights_fit(w) for w in ws)
    if mode == "fp16":
        return "fp16" if fits else "bf16x3"
    return "fp16" if fits and all(w.shape[0] >= FP16_MIN_K for w in ws) else "bf16x3"


def pack_layer(w, bias, precision="bf16x3"):
    """Fragment-order packing of W'[K,N] (see csrc/mlp.hip header) + zero-padded bias.

    Returns (packed uint16 [NT, KS, P, 64, 8], bias fp32 [NT*32]); P = 2 (hi, lo bf16 planes) for "bf16x3",
    P = 1 (fp16, round to nearest even) for "fp16"."""
    K, N = w.shape
    KS, NT = (K + 15) // 16, (N + 31) // 32
    wp = np.zeros((KS * 16, NT * 32), np.float32)
    wp[:K, :N] = w
    if precision == "fp16":
        planes = (wp.astype(np.float16).view(np.uint16),)
    else:
        hi = bf16_rne(wp)
        planes = (hi, bf16_rne(wp - bf16_to_f32(hi)))
    arr = np.zeros((NT, KS, len(planes), 64, 8), np.uint16)
    for plane, src in enumerate(planes):
        s = src.reshape(KS, 2, 8, NT, 32)                      # [ks][half][e][ct][col]
        arr[:, :, plane] = s.transpose(3, 0, 1, 4, 2).reshape(NT, KS, 64, 8)   # lane = 32*half + col
    bp = np.zeros(NT * 32, np.float32)
    bp[:N] = bias
    return arr, bp


class PackedLayer:
    __slots__ = ("K", "N", "w", "bias", "precision")

    def __init__(self, w, bias, device, _packed=None, precision="bf16x3"):
        self.K, self.N = int(w.shape[0]), int(w.shape[1])
        self.precision = precision
        if _packed is not None:                      # views into a buffer shared with the other layers of a scale
            self.w, self.bias = _packed
            return
        arr, bp = pack_layer(w, bias, precision)
        self.w = torch.from_numpy(arr.view(np.int16).reshape(-1)).to(device)
        self.bias = torch.from_numpy(bp).to(device)


def pack_scale(ws, bs, device, precision=None):
    """The layers of one MLP scale packed back to back in ONE device buffer (layer l+1 starts where layer l
    ends): the streamed-weight kernel of csrc/mlp_rowwave.hip walks them as a single linear stream.  Every
    kernel accepts these layers; separately allocated PackedLayers simply never take the streamed path.
    precision: "bf16x3" | "fp16" | None (= scale_precision(ws)); one precision per scale."""
    precision = precision or scale_precision(ws)
    packed = [pack_layer(w, b, precision) for w, b in zip(ws, bs)]
    flat = np.concatenate([arr.view(np.int16).reshape(-1) for arr, _ in packed])
    buf = torch.from_numpy(flat).to(device)
    out, off = [], 0
    for (arr, bp), w in zip(packed, ws):
        n = arr.size
        out.append(PackedLayer(w, None, device, _packed=(buf[off:off + n], torch.from_numpy(bp).to(device)), precision=precision))
        off += n
    return out


def scale_flags(layers):
    """The sa_group_mlp_max flag bits that describe how `layers` (one scale) were packed: bit 2 = fp16 planes."""
    precs = {l.precision for l in layers}
    if len(precs) != 1:
        raise ValueError("the layers of a scale must share one operand precision, got %s" % sorted(precs))
    return 4 if precs == {"fp16"} else 0


class VariableStore:
    """name -> numpy parameter dict plus a cache of folded + packed layers on `device`."""

    def __init__(self, params, device, precision=None):
        self.params = params
        self.device = torch.device(device)
        self.precision = precision           # None: MLP_PRECISION rule; "bf16x3" / "fp16": forced for every scale
        self._cache = {}
        # sticky fp16 range flag of every grouped-MLP call made with these variables (csrc/mlp_act.h): the kernels OR 1
        # into it when an input feature or hidden activation left the fp16 range; read by raise_if_overflow()
        self.overflow = torch.zeros(1, dtype=torch.int32, device=self.device)

    def raise_if_overflow(self, what="grouped MLP"):
        """Synchronising check of the sticky flag (one 4-byte read): a result computed after an fp16 overflow is
        unspecified, so this raises instead of letting it through, and clears the flag."""
        if int(self.overflow.item()) != 0:
            self.overflow.zero_()
            raise FloatingPointError(
                "%s: an activation left the fp16 range (|x| > 65504) in a scale evaluated in fp16 -- results are invalid. "
                "Use VariableStore(..., precision='bf16x3') (or weights.MLP_PRECISION = 'bf16x3') for these weights." % what)

    @classmethod
    def from_checkpoint(cls, path, device, verify=True, precision=None):
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
        return cls(tf_checkpoint.load_checkpoint(prefix, verify=verify), device, precision)

    def layer(self, scope, bn=True):
        key = (scope, bool(bn))
        if key not in self._cache:
            w, b = fold_conv_bn(self.params, scope, bn)
            self._cache[key] = PackedLayer(w, b, self.device)
        return self._cache[key]

    def scale(self, scopes, bn=True, precision=None):
        """The conv layers of one MLP scale, packed contiguously (pack_scale) at the scale's operand precision."""
        precision = precision or self.precision
        key = (tuple(scopes), bool(bn), precision)
        if key not in self._cache:
            folded = [fold_conv_bn(self.params, sc, bn) for sc in scopes]
            self._cache[key] = pack_scale([w for w, _ in folded], [b for _, b in folded], self.device, precision)
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
