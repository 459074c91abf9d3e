"""The two denoising-autoencoder classes of `codae.model`, backed by the HIP engine.

Drop-in surface: same constructor signatures, public attributes,
sub-module names (`input_layer.0.weight` ... so state_dict keys match), `forward / encode /
decode / corrupt / to` as codae/model/embedding_denoising_autoencoder.py:11-239 and
codae/model/mixed_variable_denoising_autoencoder.py:10-262 of the reference.

The nn.Linear modules exist to own the parameters (initialised in the reference's order so
the same torch seed gives the same weights) and to make `print(model)` / `state_dict()`
identical; they are never called.  `forward` hands the whole Linear/ReLU chain to
libcodae_hip.so through one autograd Function; the parameters live as views of the
engine's flat fp32 vector.  A model on a CPU device cannot run: there is no CPU path.
"""
import os

import torch

from ..hip import HipError
from ..hip import lib as _hip_lib
from ..hip import check as _check, ptr as _ptr, current_stream as _stream
from .schedule import linear_stack


class _ChainFunction(torch.autograd.Function):
    """y = chain[lo:hi](x); backward fills the engine's gradient vector."""

    @staticmethod
    def forward(ctx, x, owner, lo, hi, *params):
        eng = owner._ensure_engine(x)
        y = eng.forward(x, lo, hi)
        owner._act_stamp[lo:hi] = [eng.generation] * (hi - lo)
        ctx.owner, ctx.lo, ctx.hi, ctx.stamp = owner, lo, hi, eng.generation
        ctx.n_params = len(params)
        return y

    @staticmethod
    def backward(ctx, dy):
        owner, lo, hi = ctx.owner, ctx.lo, ctx.hi
        eng = owner._engine
        if eng is None or any(s != ctx.stamp for s in owner._act_stamp[lo:hi]):
            raise HipError("backward through a forward whose activations were overwritten by a later forward; "
                           "call backward before running the model again")
        dx = eng.backward(dy, lo, hi, need_dx=ctx.needs_input_grad[0])
        # One copy of the engine's flat gradient vector, handed out as views: autograd keeps (or accumulates) them as
        # .grad, and the next backward overwrites the engine's own buffer, so the parameters must not alias it.  (Round 1
        # cloned the 2 L tensors one by one: 20 launches per backward.)
        eng.join()                           # (weight gradients run on the engine's side stream)
        flat = eng.grads.clone()
        grads = [None] * ctx.n_params
        for l in range(lo, hi):
            grads[2 * l] = eng._view(flat, l, False)
            grads[2 * l + 1] = eng._view(flat, l, True)
        return (dx, None, None, None, *grads)


class _HipDenoisingAutoencoder(torch.nn.Module):
    """Shared machinery; subclasses only differ in the width schedule and constructor surface."""

    _mixed = False

    def _build(self, io_size, z_size, nb_input_layer, nb_output_layer, steep_layer_size, activation):
        encoder, decoder = linear_stack(io_size, z_size, nb_input_layer, nb_output_layer, steep_layer_size,
                                        self._mixed)
        self._schedule = encoder + decoder
        self._n_enc = len(encoder)
        # Build and initialise in the reference's order (encoder modules, their Xavier draws, then the
        # decoder's) so that torch.manual_seed(s) yields bit-identical initial weights.
        self.input_layer = self._sequential(encoder, activation)
        self.output_layer = self._sequential(decoder, activation)
        self._engine = None
        self._act_stamp = [0] * len(self._schedule)
        self._synced_version = None
        self.precision = os.environ.get("CODAE_PRECISION", "f32")

    @staticmethod
    def _sequential(stack, activation):
        mods = []
        for (k, n, relu) in stack:
            mods.append(torch.nn.Linear(k, n))
            if relu:
                mods.append(activation(True))
        seq = torch.nn.Sequential(*mods)
        for m in seq:
            if isinstance(m, torch.nn.Linear):
                torch.nn.init.xavier_uniform_(m.weight)     # embedding_...py:188-197
        for m in seq:
            if isinstance(m, torch.nn.Linear):
                m.bias.data.fill_(0)                        # embedding_...py:200-211
        return seq

    # ---- engine plumbing --------------------------------------------------------------
    def _linears(self):
        return [m for seq in (self.input_layer, self.output_layer) for m in seq if isinstance(m, torch.nn.Linear)]

    def _ensure_engine(self, x):
        from ..hip.engine import DaeEngine, precision_code
        lins = self._linears()
        dev = lins[0].weight.device
        if dev.type != "cuda":
            raise HipError("this build of codae runs on a HIP device only: move the model with "
                           "model.to('cuda:0') (parameters are on %s)" % dev)
        if x.device != dev:
            raise HipError("input is on %s but the model is on %s" % (x.device, dev))
        eng = self._engine
        want_prec = precision_code(self.precision)
        B = int(x.shape[0])
        if eng is None or eng.device != dev or eng.precision != want_prec or B > eng.max_batch:
            cap = max(B, eng.max_batch if eng is not None else 0, 256)
            try:
                eng = DaeEngine(self._schedule, cap, want_prec, dev, with_optimizer_state=False)
            except HipError:
                if want_prec == precision_code("bf16"):
                    # widths the bf16 tiles cannot take (e.g. abalone's 11): fp32 kernels instead
                    eng = DaeEngine(self._schedule, cap, "f32", dev, with_optimizer_state=False)
                    self.precision = "f32"
                else:
                    raise
            self._engine = eng
            self._synced_version = None
        # parameters must be views of the engine's flat vector; adopt them if they are not (first
        # use, after .to(), after someone rebound .data)
        adopted = False
        with torch.no_grad():
            for l, lin in enumerate(lins):
                for p, view in ((lin.weight, eng.weight(l)), (lin.bias, eng.bias(l))):
                    if p.data_ptr() != view.data_ptr():
                        view.copy_(p.data)
                        p.data = view
                        adopted = True
        version = sum(p._version for lin in lins for p in (lin.weight, lin.bias))
        if adopted or version != self._synced_version:
            eng.sync_shadows()          # bf16 copies follow the optimizer's in-place updates
            self._synced_version = version
        return eng

    def _chain(self, x, lo, hi):
        if x.dim() != 2:
            raise HipError("expected a [batch, features] tensor, got shape %s" % (tuple(x.shape),))
        params = [p for lin in self._linears() for p in (lin.weight, lin.bias)]
        return _ChainFunction.apply(x, self, lo, hi, *params)

    # ---- reference surface ------------------------------------------------------------
    def forward(self, x):
        """decode(encode(x)) (embedding_...py:137-151) as one fused chain."""
        return self._chain(x, 0, len(self._schedule))

    def encode(self, x):
        """embedding_...py:155-168"""
        return self._chain(x, 0, self._n_enc)

    def decode(self, z):
        """embedding_...py:171-185"""
        return self._chain(z, self._n_enc, len(self._schedule))

    def to(self, *args, **kwargs):
        """embedding_...py:214-223; the engine is rebuilt on the new device at the next call."""
        out = super().to(*args, **kwargs)
        out._engine = None
        return out

    def _apply(self, fn, *a, **k):
        out = super()._apply(fn, *a, **k)
        self._engine = None
        return out

    def _corrupt(self, input_data, mask):
        """input_data.clone() * mask (embedding_...py:226-239) on the device."""
        if input_data.device.type != "cuda":
            raise HipError("corrupt: tensors must live on a HIP device (got %s)" % input_data.device)
        x = input_data.detach().to(torch.float32).contiguous()
        m = mask.to(device=x.device, dtype=torch.float32).expand_as(x).contiguous()
        out = torch.empty_like(x)
        with torch.cuda.device(x.device):
            _check(_hip_lib().codae_corrupt(_ptr(x), _ptr(m), _ptr(out), x.numel(), _stream()))
        return out


class EmbeddingDenoisingAutoencoder(_HipDenoisingAutoencoder):
    """codae/model/embedding_denoising_autoencoder.py:9-239"""

    def __init__(self, io_size, z_size, embedding_size, nb_input_layer=2, nb_output_layer=2,
                 steep_layer_size=True, activation=torch.nn.ReLU):
        super().__init__()
        if io_size % embedding_size != 0:
            raise Exception("Error: io_size must be a multiple of embedding_size")
        self.embedding_size = embedding_size
        self.nb_category = io_size / embedding_size
        self.io_size = io_size
        self.z_size = z_size
        self.nb_input_layer = nb_input_layer
        self.nb_output_layer = nb_output_layer
        self.steep_layer_size = steep_layer_size
        self.activation = activation
        self.mode = 0
        self._build(io_size, z_size, nb_input_layer, nb_output_layer, steep_layer_size, activation)

    def corrupt(self, input_data, mask):
        return self._corrupt(input_data, mask)


class MixedVariableDenoisingAutoencoder(_HipDenoisingAutoencoder):
    """codae/model/mixed_variable_denoising_autoencoder.py:8-262"""

    _mixed = True

    def __init__(self, arch, io_size, z_size, device, nb_input_layer=2, nb_output_layer=2,
                 steep_layer_size=True, activation=torch.nn.ReLU):
        super().__init__()
        self.arch = arch
        self.z_size = z_size
        self.io_size = io_size
        self.device = device
        self.nb_input_layer = nb_input_layer
        self.nb_output_layer = nb_output_layer
        self.steep_layer_size = steep_layer_size
        self.activation = activation
        self._build(io_size, z_size, nb_input_layer, nb_output_layer, steep_layer_size, activation)

    def corrupt(self, input_data, mask, corruption_type="zero_continuous"):
        if corruption_type == "zero_continuous":
            return self._corrupt_zero_continuous(input_data=input_data, mask=mask)
        raise Exception("Error: invalid corruption type requested (zero_continuous).")

    def _corrupt_zero_continuous(self, input_data, mask):
        return self._corrupt(input_data, mask)


class cnnAutoencoder(torch.nn.Module):
    """Name kept for `from codae.model import cnnAutoencoder`; upstream is an empty stub
    (codae/model/cnn_autoencoder.py) and is out of scope."""

    def __init__(self):
        super().__init__()
