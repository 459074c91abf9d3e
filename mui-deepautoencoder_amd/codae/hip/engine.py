"""Python owner of one codae_handle and of the device buffers it borrows.

torch is used here for what the boundary allows: allocating device memory and
naming the current HIP stream.  All arithmetic happens in libcodae_hip.so.
"""
import ctypes as C

import torch

from . import (PREC_BF16, PREC_F32, S_COUNT, S_GRAD_SQ, S_GRAD_SQ_SLOTS, S_LAST_LOSS, S_N_SLOTS, S_SQ_FULL, S_SQ_PARTIAL,
               Batch, Buffers,
               HipError, Hyper, Sizes, Spec, check, current_stream, lib, ptr)


def precision_code(precision):
    if precision in (PREC_F32, "f32", "fp32", "parity"):
        return PREC_F32
    if precision in (PREC_BF16, "bf16", "throughput"):
        return PREC_BF16
    raise ValueError("unknown precision %r (use 'f32' or 'bf16')" % (precision,))


class DaeEngine:
    """Handle + flat parameter / gradient / Adam / workspace buffers on one HIP device.

    schedule: list of (in_features, out_features, relu_after) per Linear, as built by
    the model classes (embedding_denoising_autoencoder.py:49-129 of the reference).
    """

    def __init__(self, schedule, max_batch, precision, device, with_optimizer_state=True):
        device = torch.device(device)
        if device.type != "cuda":
            raise HipError("the codae HIP engine needs a HIP device (got %s); there is no CPU path" % device)
        self._lib = lib()
        self.device = device
        self.schedule = [(int(a), int(b), bool(r)) for a, b, r in schedule]
        self.L = len(self.schedule)
        self.max_batch = int(max_batch)
        self.precision = precision_code(precision)
        ins = (C.c_int32 * self.L)(*[s[0] for s in self.schedule])
        outs = (C.c_int32 * self.L)(*[s[1] for s in self.schedule])
        relu = (C.c_uint8 * self.L)(*[1 if s[2] else 0 for s in self.schedule])
        spec = Spec(self.L, ins, outs, relu, self.max_batch, self.precision)
        h = C.c_void_p()
        check(self._lib.codae_create(C.byref(spec), C.byref(h)))
        self._h = h
        sz = Sizes()
        check(self._lib.codae_get_sizes(self._h, C.byref(sz)))
        self.n_param = int(sz.n_param)
        with torch.cuda.device(device):
            f32 = dict(dtype=torch.float32, device=device)
            self._params = torch.zeros(self.n_param, **f32)
            self.grads = torch.zeros(self.n_param, **f32)
            self.adam_m = torch.zeros(self.n_param, **f32) if with_optimizer_state else None
            self.adam_v = torch.zeros(self.n_param, **f32) if with_optimizer_state else None
            self.shadow = (torch.zeros(int(sz.n_weight), dtype=torch.bfloat16, device=device)
                           if sz.n_weight > 0 else None)
            # transposed bf16 weights [in][out] per layer: lets the data-gradient GEMM run in the forward form
            self.shadow_t = (torch.zeros(int(sz.n_weight), dtype=torch.bfloat16, device=device)
                             if sz.n_weight > 0 else None)
            self.acts = torch.zeros(max(int(sz.act_bytes), 16), dtype=torch.uint8, device=device)
            self.dacts = torch.zeros(max(int(sz.dact_bytes), 16), dtype=torch.uint8, device=device)
            self.slabs = (torch.zeros(int(sz.slab_bytes), dtype=torch.uint8, device=device)
                          if sz.slab_bytes > 0 else None)
            # partial column sums of the bias gradients (one small deterministic finish kernel per backward call)
            self.bias_parts = torch.zeros(max(int(sz.bias_part_bytes) // 4, 16), dtype=torch.float32, device=device)
            if int(sz.n_scalars) != S_COUNT:
                raise HipError("library reports %d scalars, binding expects %d" % (int(sz.n_scalars), S_COUNT))
            self.scalars = torch.zeros(int(sz.n_scalars), dtype=torch.float64, device=device)
        self.bufs = Buffers(ptr(self._params), ptr(self.grads), ptr(self.adam_m), ptr(self.adam_v), ptr(self.shadow),
                            ptr(self.acts), ptr(self.dacts), ptr(self.slabs), ptr(self.scalars), ptr(self.shadow_t),
                            ptr(self.bias_parts))
        self.w_off, self.b_off = [], []
        for l in range(self.L):
            w, b, s = C.c_int64(), C.c_int64(), C.c_int64()
            check(self._lib.codae_param_offsets(self._h, l, C.byref(w), C.byref(b), C.byref(s)))
            self.w_off.append(w.value)
            self.b_off.append(b.value)
        self.step_count = 0
        self.generation = 0   # bumped by every forward: guards stale backward calls

    def __del__(self):
        h = getattr(self, "_h", None)
        if h is not None and h.value:
            try:
                self._lib.codae_destroy(h)
            except Exception:
                pass
            self._h = None

    # ---- views into the flat vectors --------------------------------------------------
    def join(self):
        """The last update's per-layer Adam kernels may still run on the engine's side stream: make the
        current stream wait for them (call before reading parameters / Adam state / gradients)."""
        with torch.cuda.device(self.device):
            check(self._lib.codae_join(self._h, current_stream()))

    @property
    def params(self):
        self.join()
        return self._params

    def _view(self, flat, l, bias):
        self.join()
        k, n, _ = self.schedule[l]
        if bias:
            return flat[self.b_off[l]:self.b_off[l] + n]
        return flat[self.w_off[l]:self.w_off[l] + n * k].view(n, k)

    def weight(self, l): return self._view(self._params, l, False)
    def bias(self, l): return self._view(self._params, l, True)
    def weight_grad(self, l): return self._view(self.grads, l, False)
    def bias_grad(self, l): return self._view(self.grads, l, True)

    def load_params(self, params):
        """params: list of (W[out,in], b[out]) tensors/arrays."""
        with torch.no_grad():
            for l, (w, b) in enumerate(params):
                self.weight(l).copy_(torch.as_tensor(w, dtype=torch.float32))
                self.bias(l).copy_(torch.as_tensor(b, dtype=torch.float32))
        self.sync_shadows()

    def sync_shadows(self):
        with torch.cuda.device(self.device):
            check(self._lib.codae_sync_shadows(self._h, C.byref(self.bufs), current_stream()))

    # ---- drop-in path -----------------------------------------------------------------
    def forward(self, x, layer_lo=0, layer_hi=None):
        layer_hi = self.L if layer_hi is None else layer_hi
        if x.dim() != 2 or x.shape[1] != self.schedule[layer_lo][0]:
            raise HipError("forward: input shape %s does not match layer %d (%d features)"
                           % (tuple(x.shape), layer_lo, self.schedule[layer_lo][0]))
        if x.shape[0] > self.max_batch:
            raise HipError("forward: batch %d exceeds the engine's max_batch %d" % (x.shape[0], self.max_batch))
        x = x.detach().to(dtype=torch.float32).contiguous()
        y = torch.empty((x.shape[0], self.schedule[layer_hi - 1][1]), dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            check(self._lib.codae_forward(self._h, C.byref(self.bufs), ptr(x), ptr(y), x.shape[0], layer_lo, layer_hi,
                                          1, current_stream()))
        self.generation += 1
        return y

    def backward(self, dy, layer_lo=0, layer_hi=None, need_dx=False):
        layer_hi = self.L if layer_hi is None else layer_hi
        dy = dy.detach().to(dtype=torch.float32).contiguous()
        dx = (torch.empty((dy.shape[0], self.schedule[layer_lo][0]), dtype=torch.float32, device=self.device)
              if need_dx else None)
        with torch.cuda.device(self.device):
            check(self._lib.codae_backward(self._h, C.byref(self.bufs), ptr(dy), ptr(dx), dy.shape[0], layer_lo,
                                           layer_hi, current_stream()))
        return dx

    # ---- fused step path --------------------------------------------------------------
    def make_batch(self, data, row_idx, mask_id, mask_table, B=None, mask_to_use=None, run=0):
        """data [N,io] fp32; row_idx / mask_id int32 [B] or None; mask_table uint8 [n_masks, io] or None;
        mask_to_use int32 [N, nb_run] + run: device-side id lookup when mask_id is None."""
        if data.dtype != torch.float32 or not data.is_contiguous() or data.device != self.device:
            raise HipError("batch data must be a contiguous fp32 tensor on %s" % self.device)
        for name, t in (("row_idx", row_idx), ("mask_id", mask_id)):
            if t is not None and (t.dtype != torch.int32 or t.device != self.device or not t.is_contiguous()):
                raise HipError("%s must be a contiguous int32 tensor on %s" % (name, self.device))
        if mask_table is not None and (mask_table.dtype != torch.uint8 or mask_table.device != self.device):
            raise HipError("mask_table must be a uint8 tensor on %s" % self.device)
        if B is None:
            B = int(row_idx.numel()) if row_idx is not None else int(data.shape[0])
        nb_run = 0
        if mask_to_use is not None:
            if mask_to_use.dtype != torch.int32 or mask_to_use.device != self.device or not mask_to_use.is_contiguous() \
                    or mask_to_use.dim() != 2 or mask_to_use.shape[0] != data.shape[0]:
                raise HipError("mask_to_use must be a contiguous int32 [n_rows, nb_run] tensor on %s" % self.device)
            nb_run = int(mask_to_use.shape[1])
        b = Batch(ptr(data), ptr(row_idx), ptr(mask_id), ptr(mask_table), int(B), int(data.shape[1]),
                  ptr(mask_to_use), nb_run, int(run))
        # the struct only carries raw pointers: pin the tensors to it, or a temporary (mask_id) is
        # returned to the caching allocator and handed to the next torch.empty() while the kernels
        # that read it are still queued
        b._pinned = (data, row_idx, mask_id, mask_table, mask_to_use)
        return b

    def hyper(self, lr, weight_decay, clip=1.0, global_rows=0, betas=(0.9, 0.999), eps=1e-8, step=None):
        return Hyper(lr, weight_decay, betas[0], betas[1], eps, clip if clip else 0.0,
                     self.step_count + 1 if step is None else step, float(global_rows))

    def train_step(self, batch, hyper, graph=False):
        """graph=True: replay the step from a hipGraph (captured on first use; batch.row_idx / mask_id must be
        buffers whose contents, not addresses, change between calls)."""
        fn = self._lib.codae_train_step_graph if graph else self._lib.codae_train_step
        with torch.cuda.device(self.device):
            check(fn(self._h, C.byref(self.bufs), C.byref(batch), C.byref(hyper), current_stream()))
        self.step_count += 1

    # ---- data parallel with the library's own RCCL communicator (codae_dp_*) ----------------------------------------
    def dp_unique_id(self):
        """bytes of a fresh ncclUniqueId (call on rank 0, ship to the other ranks)."""
        buf = (C.c_char * 128)()
        check(self._lib.codae_dp_unique_id(buf, 128))
        return bytes(buf)

    def dp_init(self, unique_id, rank, world):
        """Collective: every rank calls it with rank 0's id.  Creates this engine's communicator and collective stream."""
        buf = (C.c_char * 128).from_buffer_copy(unique_id)
        with torch.cuda.device(self.device):
            check(self._lib.codae_dp_init(self._h, buf, int(rank), int(world)))

    def train_step_dp(self, batch, hyper, buckets):
        """codae_train_step_dp: forward + loss, bucketed backward with the all-reduces issued by the library on its own stream,
        clip + Adam - one call, no Python between the buckets."""
        n = len(buckets)
        lo = (C.c_int32 * n)(*[int(a) for a, _ in buckets])
        hi = (C.c_int32 * n)(*[int(b) for _, b in buckets])
        with torch.cuda.device(self.device):
            check(self._lib.codae_train_step_dp(self._h, C.byref(self.bufs), C.byref(batch), C.byref(hyper), n, lo, hi, current_stream()))
        self.step_count += 1

    def step_forward_loss(self, batch, hyper, out_y=None):
        with torch.cuda.device(self.device):
            check(self._lib.codae_step_forward_loss(self._h, C.byref(self.bufs), C.byref(batch), C.byref(hyper),
                                                    ptr(out_y), current_stream()))

    def step_backward(self, B, layer_lo, layer_hi, join=True):
        """join=False (data parallel): return without making the current stream wait for the side stream; the
        weight gradients of the range are then ordered on `side_stream()` (see codae_step_backward_async)."""
        fn = self._lib.codae_step_backward if join else self._lib.codae_step_backward_async
        with torch.cuda.device(self.device):
            check(fn(self._h, C.byref(self.bufs), B, layer_lo, layer_hi, current_stream()))

    def side_stream(self):
        """torch view of the engine's side stream (None when it runs everything on the caller's stream)."""
        if getattr(self, "_side_ext", None) is None:
            out = C.c_void_p()
            with torch.cuda.device(self.device):
                check(self._lib.codae_side_stream(self._h, C.byref(out)))
            self._side_ext = torch.cuda.ExternalStream(out.value, device=self.device) if out.value else False
        return self._side_ext or None

    def step_update(self, hyper):
        with torch.cuda.device(self.device):
            check(self._lib.codae_step_update(self._h, C.byref(self.bufs), C.byref(hyper), current_stream()))
        self.step_count += 1

    # ---- sharded data-parallel update (codae.train.DataParallel(sharded=True)) -------------------
    def span_sumsq(self, lo, hi, acc):
        """acc (1-element float64 device tensor) += sum grads[lo:hi]^2."""
        with torch.cuda.device(self.device):
            check(self._lib.codae_span_sumsq(C.c_void_p(self.grads.data_ptr() + 4 * lo), hi - lo, ptr(acc), current_stream()))

    def step_update_span(self, hyper, lo, hi, total_sq):
        """clip + Adam on flat elements [lo, hi) with the global sum g^2 in `total_sq` (float64 device tensor)."""
        with torch.cuda.device(self.device):
            check(self._lib.codae_step_update_span(self._h, C.byref(self.bufs), C.byref(hyper), lo, hi, ptr(total_sq),
                                                   current_stream()))

    def replica_tensors(self):
        """Flat tensors (parameter layout) every rank must hold in full for the next forward / backward: the bf16 weight
        shadow in BF16 mode (biases are updated on every rank), the fp32 parameters in F32 mode."""
        return [self.shadow] if self.precision == PREC_BF16 else [self._params]

    def after_replica_sync(self):
        """The shadows were all-gathered: rebuild the transposed copies the data-gradient GEMMs read."""
        with torch.cuda.device(self.device):
            check(self._lib.codae_sync_transposed(self._h, C.byref(self.bufs), current_stream()))

    def new_accumulator(self):
        return torch.zeros(1, dtype=torch.float64, device=self.device)

    def step_path(self, B):
        """'chain' if a fused step of B rows runs the persistent chain kernel, 'layers' for per-layer launches."""
        return "chain" if self._lib.codae_step_path(self._h, C.byref(self.bufs), int(B)) == 1 else "layers"

    def eval_step(self, batch, out_y=None):
        with torch.cuda.device(self.device):
            check(self._lib.codae_eval_step(self._h, C.byref(self.bufs), C.byref(batch), ptr(out_y), current_stream()))

    def profile_begin(self, classes=("gemm_fwd", "gemm_dgrad", "gemm_wgrad"), max_records=4096, every=1):
        from . import KERNEL_CLASSES
        check(self._lib.codae_profile_stride(self._h, int(every)))
        mask = 0
        for c in classes:
            mask |= 1 << KERNEL_CLASSES.index(c)
        self._prof_cap = int(max_records)
        check(self._lib.codae_profile_begin(self._h, mask, self._prof_cap))

    def profile_end(self):
        """{class name: [milliseconds per launch]} (synchronises on the recorded events)."""
        from . import KERNEL_CLASSES
        kinds = (C.c_int32 * self._prof_cap)()
        ms = (C.c_float * self._prof_cap)()
        n = C.c_int32()
        check(self._lib.codae_profile_end(self._h, kinds, ms, self._prof_cap, C.byref(n)))
        out = {}
        for i in range(n.value):
            out.setdefault(KERNEL_CLASSES[kinds[i]], []).append(float(ms[i]))
        return out

    def output_view(self, B):
        """fp32 [B, io] view of the reconstruction the last step left in the workspace."""
        n = self.schedule[-1][1]
        # y lives after the L activation buffers; recompute its offset like the C side does
        rows = (self.max_batch + 63) // 64 * 64
        es = 2 if self.precision == PREC_BF16 else 4
        off = 0
        for (k, _, _) in self.schedule:
            off += (rows * k * es + 255) // 256 * 256
        return self.acts[off:off + B * n * 4].view(torch.float32).view(B, n)

    def read_scalars(self):
        """(sq_full, sq_partial, grad_sq, last_loss) — synchronises."""
        s = self.scalars.cpu()
        gsq = float(s[S_GRAD_SQ]) + float(s[S_GRAD_SQ_SLOTS:S_GRAD_SQ_SLOTS + S_N_SLOTS].sum())
        return float(s[S_SQ_FULL]), float(s[S_SQ_PARTIAL]), gsq, float(s[S_LAST_LOSS])

    def record_grad_sq(self, acc):
        """The sharded data-parallel update gathers the global sum g^2 in its own accumulator: keep it where a fused
        step leaves it (scalars[GRAD_SQ]; the slots stay zero on that path)."""
        self.scalars[S_GRAD_SQ:S_GRAD_SQ + 1].copy_(acc.reshape(1))

    def zero_metric_sums(self):
        self.scalars[:2].zero_()
