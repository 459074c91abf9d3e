"""Device-resident training loop for the embedding DAE: the fast counterpart of the inner loop of
script/train_dae_on_embedding.py:194-223 (reference), one fused HIP step per minibatch.

What differs from the drop-in path (model(c_input) + torch loss/optimizer):
  - the dataset matrix stays in HBM and a batch is a vector of int32 row indices
    (DataLoader + collate_embedding, data_tool.py:96-103, become a gather inside the kernels);
  - the corruption mask is a per-sample mask id (Corrupter.mask_to_use[idx][run]) plus the small
    uint8 table, applied on load; the [B, io] fp32 fmask is never built;
  - loss, dL/dy, the epoch metrics (ftl / ptl sums), grad-norm clip and Adam are kernels of
    libcodae_hip.so; nothing is copied to the host per step (the reference moves a [B, io] fp32
    tensor to the host every step, :218-223).

Data parallel (one process per GPU, torch.distributed; backend "nccl" is RCCL over xGMI):
the minibatch is sharded by rows, parameters and Adam state are replicated, the loss is scaled
by the GLOBAL batch so that a SUM all-reduce of the gradients gives the global-batch mean
gradient; gradients are reduced in layer buckets, each launched as soon as its layers'
backward kernels are enqueued, so the reduction of late layers overlaps the backward GEMMs of
early ones; clip + Adam then run identically on every rank.

Sharded update (DataParallel(sharded=True)): the bucket collectives become reduce-scatters, every rank runs clip + Adam
on its 1/N of each bucket only (the bias block, 15 K floats, stays replicated), and the bf16 weight shadows - what the
next forward / backward actually read - are all-gathered: 25 % fewer bytes on xGMI than the all-reduce (4 + 2 instead of
4 + 4 bytes per parameter), Adam's 750 MB pass shrinks N-fold, and the fp32 master parameters / moments of a rank stay
valid for its own shards only until gather_params() is called (checkpoint, read-out).
"""
import torch

from .hostcpu import fit_host_threads, host_cpu_share  # noqa: F401  (re-exported: scripts and bench.py import them here)


class SubsetEpochSampler:
    """Batches of dataset row indices in the order
    DataLoader(dataset, batch_size, sampler=SubsetRandomSampler(indices)) yields them
    (script/train_dae_on_embedding.py:118-128 of the reference): one draw of the loader's base
    seed, then torch.randperm over the subset, both from torch's default generator — so a seeded
    run visits the same batches.  Indices come back as one int64 tensor per batch; nothing is
    gathered on the host."""

    def __init__(self, indices, batch_size):
        self.indices = torch.as_tensor(list(indices), dtype=torch.long)
        self.batch_size = int(batch_size)

    def __len__(self):
        return (len(self.indices) + self.batch_size - 1) // self.batch_size

    def _epoch_order(self):
        torch.empty((), dtype=torch.int64).random_()          # DataLoader iterator's base seed
        # torch.randperm on the CPU draws the same permutation whatever the thread count; with one thread it takes 0.4 ms for
        # 131 072 rows, with one per visible cpu of a GPU box 4-10 ms and the container's CPU quota (fit_host_threads above;
        # tools/abl/randperm_cost.py, tools/bench_script_loop.py)
        n_threads = torch.get_num_threads()
        if n_threads > 1:
            torch.set_num_threads(1)
        try:
            perm = torch.randperm(len(self.indices))
        finally:
            if n_threads > 1:
                torch.set_num_threads(n_threads)
        return self.indices[perm]

    def __iter__(self):
        order = self._epoch_order()
        for o in range(0, len(order), self.batch_size):
            yield order[o:o + self.batch_size]

    def device_batches(self, device, dtype=torch.int32):
        """The same batches as iterating the sampler (same draws from torch's default generator), as views of ONE device tensor
        holding the epoch's whole order: one host-to-device copy per epoch instead of one per step.  (A per-step `.to(device)`
        of a pageable host tensor is a synchronous copy ordered behind the previous step's kernels: the host can never run
        ahead of the device, which costs the step its enqueue time - tools/bench_script_loop.py.)"""
        order = self._epoch_order().to(dtype)
        if torch.device(device).type == "cuda":
            # one pinned staging buffer and one device buffer, both kept: the copy is ordered on the current stream behind the
            # steps that still read the previous epoch's order, and the host only waits for the PREVIOUS copy before reusing the
            # staging buffer (allocating / freeing pinned memory per epoch synchronises the device)
            st = getattr(self, "_stage", None)
            if st is None or st[0].numel() != order.numel() or st[0].dtype != order.dtype or st[1].device != torch.device(device):
                st = (torch.empty(order.numel(), dtype=order.dtype).pin_memory(), torch.empty(order.numel(), dtype=order.dtype, device=device),
                      torch.cuda.Event())
                self._stage = st
            else:
                st[2].synchronize()
            st[0].copy_(order)
            st[1].copy_(st[0], non_blocking=True)
            st[2].record()
            order = st[1]
        else:
            order = order.to(device)
        for o in range(0, len(order), self.batch_size):
            yield order[o:o + self.batch_size]


def shard_batch(batch_indices, rank, world):
    """This rank's rows of one global minibatch (strided: rank, rank + world, ...), or None when the batch has fewer
    rows than there are ranks (the ragged last batch of an epoch): EVERY rank then skips it - a rank with an empty shard
    would fail its launch while the others block forever in the gradient collectives."""
    if world <= 1:
        return batch_indices
    if len(batch_indices) < world:
        return None
    return batch_indices[rank::world]


def seed_all_ranks(seed):
    """Data parallel runs need the same sampler order, mask tables (Python's `random`, data_tool.py:222-226) and initial
    weights on every rank: seed the three generators the reference leaves unseeded (RNG)."""
    import random
    import numpy as np
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)


def default_buckets(n_layers, n_buckets=4):
    """[(lo, hi)] layer ranges in backward order, sizes decreasing: 10 layers -> (6,10) (3,6) (1,3) (0,1).
    The first buckets' all-reduces hide under the remaining backward GEMMs; the last bucket's cannot, so it
    is the smallest (one layer = 9.4 MB at 1536^2)."""
    n_buckets = max(1, min(n_buckets, n_layers))
    weights = list(range(n_buckets, 0, -1))                     # n, n-1, ..., 1
    total = sum(weights)
    sizes = [max(1, round(n_layers * w / total)) for w in weights]
    while sum(sizes) > n_layers:                                # trim from the biggest
        sizes[sizes.index(max(sizes))] -= 1
    while sum(sizes) < n_layers:                                # pad the first (best hidden) bucket
        sizes[0] += 1
    out, hi = [], n_layers
    for sz in sizes:
        if sz > 0:
            out.append((hi - sz, hi))
            hi -= sz
    return out


def init_rccl_process_group(device, **kw):
    """`torch.distributed.init_process_group("nccl")` (= RCCL on ROCm) with the collectives on HIGH-priority streams.

    Why it matters here: the ROCm runtime multiplexes the streams of one priority level onto a few hardware queues,
    and a `hipStreamWaitEvent` is a barrier packet that holds up EVERYTHING behind it on its hardware queue.  With
    torch's default (normal-priority) collective stream sharing the queue of the compute stream, each bucket
    all-reduce's wait for the engine's side stream stalled the dgrad chain for ~90 us (rocprofv3 trace, forced
    one-rank collectives).  High priority puts the collective stream in its own queue pool: compute stream normal,
    engine side stream low, collectives high."""
    import torch.distributed as dist
    opts = None
    try:
        opts = dist.ProcessGroupNCCL.Options(is_high_priority_stream=True)
    except Exception:                                   # builds without the NCCL/RCCL process group
        opts = None
    if opts is not None:
        kw.setdefault("pg_options", opts)
    dist.init_process_group(backend="nccl", device_id=device, **kw)


class DataParallel:
    """Bucketed gradient all-reduce around an engine's step_* phases.

    `engine` needs: L, grads (flat tensor), w_off (list), b_off (list), n_param,
    step_forward_loss(batch, hyper), step_backward(B, lo, hi), step_update(hyper).
    """

    def __init__(self, engine, process_group=None, n_buckets=None, sharded=False, native=False):
        """n_buckets: 1 .. engine.L; None = one bucket per layer (the first collective starts one layer into the
        backward, the exposed tail is one layer's gradient); sharded: see the module docstring."""
        if n_buckets is None:
            # (the sharded update issues five calls per bucket - reduce-scatter, span norm, span Adam, two shadow all-gathers - and
            #  becomes host-bound with ten of them: 1.86 ms/step in the one-rank rehearsal against 1.36 for the all-reduce path)
            n_buckets = 4 if sharded else engine.L
        import torch.distributed as dist
        self.dist = dist
        self.engine = engine
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(process_group) if dist.is_initialized() else 0
        self.buckets = default_buckets(engine.L, n_buckets)
        self.sharded = bool(sharded)
        # native: the library owns the RCCL communicator (codae_dp_init) and issues the bucket all-reduces itself, on its own
        # stream, inside ONE call per step (codae_train_step_dp) - no Python between the buckets.  torch.distributed is then
        # used only to ship rank 0's ncclUniqueId and for the per-epoch scalar reductions.  Opt-in (CODAE_DP_NATIVE=1 / native=
        # True): it has only ever run with one rank on this build's single-GPU test boxes.
        import os as _os
        self.native = bool(native) or _os.environ.get("CODAE_DP_NATIVE") == "1"
        if self.native:
            if self.sharded:
                raise ValueError("native RCCL data parallel: the sharded update goes through torch.distributed only")
            if not hasattr(engine, "dp_init"):
                raise ValueError("native RCCL data parallel needs the HIP engine")
            ids = [engine.dp_unique_id() if self.rank == 0 else None]
            if dist.is_initialized() and self.world > 1:
                dist.broadcast_object_list(ids, src=0, group=process_group)
            engine.dp_init(ids[0], self.rank, self.world)
        if self.sharded:
            for lo, hi in self.buckets:
                n = self._weight_span_bounds(lo, hi)
                if (n[1] - n[0]) % (4 * self.world) != 0:
                    raise ValueError("sharded update: bucket of %d elements does not split into %d shards of whole float4s"
                                     % (n[1] - n[0], self.world))
        # CODAE_DP_FORCE_ALLREDUCE=1: issue the bucketed collectives even with one rank (lets a
        # single-GPU box exercise the RCCL path end to end)
        import os
        self.always_reduce = dist.is_initialized() and os.environ.get("CODAE_DP_FORCE_ALLREDUCE") == "1"
        self._tw_every, self._tw_step, self._tw = 0, 0, []

    # ---- exposed-wait timing (bench.py --gpus N) ------------------------------------------
    def time_waits(self, every=4):
        """In every `every`-th step bracket each bucket's wait with a hipEvent pair on the compute stream: the
        elapsed time is how long the step was held up by that bucket's all-reduce (0 = fully hidden)."""
        self._tw_every, self._tw_step, self._tw = max(1, int(every)), 0, []

    def wait_report(self):
        """[mean exposed milliseconds per step] per bucket (last entry: the bias block); synchronises."""
        if not self._tw:
            return None
        import torch
        torch.cuda.synchronize()
        n = len(self._tw[0])
        return [sum(rec[i][0].elapsed_time(rec[i][1]) for rec in self._tw) / len(self._tw) for i in range(n)]

    def _weight_span_bounds(self, lo, hi):
        # weights of consecutive layers are contiguous in the flat vector
        end = self.engine.w_off[hi] if hi < self.engine.L else self.engine.b_off[0]
        return self.engine.w_off[lo], end

    def _weight_span(self, lo, hi):
        a, b = self._weight_span_bounds(lo, hi)
        return self.engine.grads[a:b]

    def _shard_bounds(self, lo, hi):
        """element range of THIS rank's shard of bucket (lo, hi)"""
        a, b = self._weight_span_bounds(lo, hi)
        n = (b - a) // self.world
        return a + self.rank * n, a + (self.rank + 1) * n

    def backward_and_reduce(self, B):
        eng = self.engine
        if self.world == 1 and not self.always_reduce:
            eng.step_backward(B, 0, eng.L)
            return
        works = []
        side = eng.side_stream() if hasattr(eng, "side_stream") else None
        if side is not None:
            # No join between buckets: the bucket's weight gradients are complete on the engine's side stream, so the
            # collective is ordered behind THAT stream while the main stream goes straight on with the next bucket.
            import torch
            for lo, hi in self.buckets:
                eng.step_backward(B, lo, hi, join=False)
                with torch.cuda.stream(side):
                    works.append(self.dist.all_reduce(self._weight_span(lo, hi), op=self.dist.ReduceOp.SUM,
                                                      group=self.group, async_op=True))
            eng.join()
        else:
            for lo, hi in self.buckets:
                eng.step_backward(B, lo, hi)
                works.append(self.dist.all_reduce(self._weight_span(lo, hi), op=self.dist.ReduceOp.SUM,
                                                  group=self.group, async_op=True))
        # bias gradients of layer l are finished by the dgrad of layer l+1: reduce the block last
        works.append(self.dist.all_reduce(eng.grads[eng.b_off[0]:eng.n_param], op=self.dist.ReduceOp.SUM,
                                          group=self.group, async_op=True))
        timed = False
        if self._tw_every:
            self._tw_step += 1
            timed = self._tw_step % self._tw_every == 0
        if not timed:
            # every collective of this process group runs in issue order on ONE internal stream, so the compute stream only has to
            # wait for the LAST one (the bias block): one cross-stream wait per step instead of one per bucket (~11 us of compute-
            # stream time each in the one-rank rehearsal: 0.12 ms per step with per-layer buckets)
            if self.dist.get_backend(self.group) == "nccl":
                works[-1].wait()
                self._keep_works = works      # (handles stay referenced until the next step replaces them)
            else:
                for w in works:               # (gloo, the CPU test backend: no stream semantics)
                    w.wait()
            return
        import torch
        rec = []
        for w in works:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            w.wait()
            e1.record()
            rec.append((e0, e1))
        self._tw.append(rec)

    def train_step(self, batch, hyper, B):
        """forward+loss -> bucketed backward with overlapped all-reduce -> clip+Adam.
        `hyper.loss_scale_rows` must hold the GLOBAL batch rows."""
        if self.native:
            self.engine.train_step_dp(batch, hyper, self.buckets)
            return
        self.engine.step_forward_loss(batch, hyper)
        if self.sharded and (self.world > 1 or self.always_reduce):
            self._backward_reduce_scatter(B)
            self._sharded_update(hyper)
            return
        self.backward_and_reduce(B)
        self.engine.step_update(hyper)

    # ---- sharded update ---------------------------------------------------------------------
    def _backward_reduce_scatter(self, B):
        """As backward_and_reduce, but every bucket is REDUCE-SCATTERED: afterwards this rank holds the summed gradient of
        its own shard of each bucket (in place, inside the flat gradient vector); the bias block is all-reduced."""
        eng = self.engine
        works = []
        side = eng.side_stream() if hasattr(eng, "side_stream") else None
        for lo, hi in self.buckets:
            if side is not None:
                eng.step_backward(B, lo, hi, join=False)
            else:
                eng.step_backward(B, lo, hi)
            span = self._weight_span(lo, hi)
            a, b = self._shard_bounds(lo, hi)
            shard = eng.grads[a:b]
            if side is not None:
                with torch.cuda.stream(side):
                    works.append(self.dist.reduce_scatter_tensor(shard, span, op=self.dist.ReduceOp.SUM, group=self.group, async_op=True))
            else:
                works.append(self.dist.reduce_scatter_tensor(shard, span, op=self.dist.ReduceOp.SUM, group=self.group, async_op=True))
        if side is not None:
            eng.join()
        works.append(self.dist.all_reduce(eng.grads[eng.b_off[0]:eng.n_param], op=self.dist.ReduceOp.SUM, group=self.group, async_op=True))
        if self.dist.get_backend(self.group) == "nccl":
            works[-1].wait()                  # (one in-order collective stream: see backward_and_reduce)
            self._keep_works = works
        else:
            for w in works:
                w.wait()

    def _sharded_update(self, hyper):
        eng = self.engine
        # clip_grad_norm_'s total norm: every rank's shards (+ the replicated bias block, counted once) -> one all-reduce
        acc = eng.new_accumulator()
        for lo, hi in self.buckets:
            a, b = self._shard_bounds(lo, hi)
            eng.span_sumsq(a, b, acc)
        if self.rank == 0:
            eng.span_sumsq(eng.b_off[0], eng.n_param, acc)
        self.dist.all_reduce(acc, op=self.dist.ReduceOp.SUM, group=self.group)
        self.last_grad_sq = acc
        if hasattr(eng, "record_grad_sq"):
            eng.record_grad_sq(acc)          # where read_scalars() / last_loss_and_grad_norm() look for the step's sum g^2
        for lo, hi in self.buckets:
            a, b = self._shard_bounds(lo, hi)
            eng.step_update_span(hyper, a, b, acc)
        eng.step_update_span(hyper, eng.b_off[0], eng.n_param, acc)
        # what the next step reads must be whole on every rank (RCCL gathers in place: a rank's input IS its slot of
        # the output; gloo, the CPU test backend, wants a separate input)
        in_place = self.dist.get_backend(self.group) == "nccl"
        works = []
        for t in eng.replica_tensors():
            for lo, hi in self.buckets:
                a, b = self._weight_span_bounds(lo, hi)
                sa, sb = self._shard_bounds(lo, hi)
                src = t[sa:sb] if in_place else t[sa:sb].clone()
                works.append(self.dist.all_gather_into_tensor(t[a:b], src, group=self.group, async_op=True))
        for w in works:
            w.wait()
        eng.after_replica_sync()
        eng.step_count += 1

    def gather_params(self):
        """Sharded mode: make the fp32 master parameters whole on every rank (each rank has only kept its own shards
        current).  Call before reading / saving parameters."""
        if not self.sharded or self.world == 1:
            return
        eng = self.engine
        p = eng.params if not hasattr(eng, "_params") else eng._params
        for lo, hi in self.buckets:
            a, b = self._weight_span_bounds(lo, hi)
            sa, sb = self._shard_bounds(lo, hi)
            self.dist.all_gather_into_tensor(p[a:b], p[sa:sb].clone(), group=self.group)

    def broadcast_params(self, params_flat):
        if self.world > 1:
            self.dist.broadcast(params_flat, src=0, group=self.group)

    def reduce_scalars(self, t):
        if self.world > 1:
            self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, group=self.group)
        return t


class HipEmbeddingTrainer:
    """Owns a DaeEngine, the resident dataset and the mask tables; runs train / eval steps."""

    def __init__(self, schedule, data, mask_table_u8, mask_to_use_i32, lr, weight_decay, clip=1.0,
                 max_batch=8192, precision="bf16", device="cuda:0", distributed=False, n_buckets=None, use_graph=False,
                 sharded_update=False, native_dp=False):
        """use_graph: replay the fused step from a hipGraph (codae_train_step_graph): for launch-bound shapes
        (small batches); single process only - the bucketed data-parallel step is not captured."""
        from .hip.engine import DaeEngine
        fit_host_threads()      # the loop that feeds this trainer must not get its container CPU-throttled (codae/hostcpu.py)
        self.device = torch.device(device)
        self.engine = DaeEngine(schedule, max_batch, precision, self.device)
        self.data = data.to(device=self.device, dtype=torch.float32).contiguous()
        self.mask_table = None if mask_table_u8 is None else mask_table_u8.to(self.device).contiguous()
        self.mask_to_use = None if mask_to_use_i32 is None else mask_to_use_i32.to(self.device).contiguous()
        self.lr, self.weight_decay, self.clip = lr, weight_decay, clip
        self.dp = DataParallel(self.engine, n_buckets=n_buckets, sharded=sharded_update, native=native_dp) if distributed else None
        self.world = self.dp.world if self.dp else 1
        self.use_graph = bool(use_graph) and self.dp is None
        # graph replay freezes kernel arguments: the step's row indices / mask ids are copied into these
        self._idx_buf = torch.zeros(max_batch, dtype=torch.int32, device=self.device) if self.use_graph else None
        self._mid_buf = torch.zeros(max_batch, dtype=torch.int32, device=self.device) if self.use_graph else None
        # (the default stream cannot be captured: graph steps run on a stream of their own, ordered after / before
        # the caller's current stream)
        self._graph_stream = torch.cuda.Stream(device=self.device) if self.use_graph else None

    # ---- parameters ---------------------------------------------------------------------
    def load_params(self, params):
        self.engine.load_params(params)
        if self.dp:
            self.dp.broadcast_params(self.engine.params)
            self.engine.sync_shadows()

    def init_params(self, seed=0):
        """Xavier-uniform weights / zero bias (embedding_...py:188-211), same on every rank."""
        g = torch.Generator(device="cpu")
        g.manual_seed(seed)
        ps = []
        for (k, n, _) in self.engine.schedule:
            a = (6.0 / (k + n)) ** 0.5
            ps.append(((torch.rand((n, k), generator=g) * 2 - 1) * a, torch.zeros(n)))
        self.load_params(ps)

    def params(self):
        return [(self.engine.weight(l), self.engine.bias(l)) for l in range(self.engine.L)]

    # ---- steps ---------------------------------------------------------------------------
    def _batch(self, row_idx, run):
        if self.mask_to_use is not None and run is not None:
            # id = mask_to_use[row][run] is looked up inside the kernels
            return self.engine.make_batch(self.data, row_idx, None, self.mask_table, mask_to_use=self.mask_to_use, run=run)
        return self.engine.make_batch(self.data, row_idx, None, None)

    def train_batch(self, row_idx, run=0, mask_id=None, global_rows=None):
        """One optimizer step on rows `row_idx` (int32 device tensor) of the resident dataset.
        global_rows: rows of the whole minibatch over all ranks (default: B * world)."""
        eng = self.engine
        if self.use_graph:
            caller = torch.cuda.current_stream(self.device)
            self._graph_stream.wait_stream(caller)
            with torch.cuda.stream(self._graph_stream):
                n = int(row_idx.numel())
                self._idx_buf[:n].copy_(row_idx)
                if mask_id is not None:
                    self._mid_buf[:n].copy_(mask_id)
                    batch = eng.make_batch(self.data, self._idx_buf[:n], self._mid_buf[:n], self.mask_table)
                else:
                    batch = self._batch(self._idx_buf[:n], run)
                B = batch.B
                hyper = eng.hyper(self.lr, self.weight_decay, self.clip,
                                  global_rows=B * self.world if global_rows is None else global_rows)
                eng.train_step(batch, hyper, graph=True)
            caller.wait_stream(self._graph_stream)
            self._keep = batch
            return B
        if mask_id is None:
            batch = self._batch(row_idx, run)
        else:
            batch = eng.make_batch(self.data, row_idx, mask_id, self.mask_table)
        B = batch.B
        hyper = eng.hyper(self.lr, self.weight_decay, self.clip,
                          global_rows=B * self.world if global_rows is None else global_rows)
        if self.dp is None:
            eng.train_step(batch, hyper)
        else:
            self.dp.train_step(batch, hyper, B)
        self._keep = batch  # keep the ctypes struct (and its tensors) alive until the next call
        return B

    def eval_batch(self, row_idx, run=0, want_y=False):
        batch = self._batch(row_idx, run)
        y = (torch.empty((batch.B, self.data.shape[1]), dtype=torch.float32, device=self.device)
             if want_y else None)
        self.engine.eval_step(batch, y)
        self._keep = batch
        return y

    def epoch_sums(self, reset=True, reduce=True):
        """(sum (x-y)^2, sum (1-fmask)(x-y)^2) accumulated since the last reset (summed over the
        ranks when `reduce`)."""
        s = self.engine.scalars[:2].clone()
        if self.dp and reduce:
            self.dp.reduce_scalars(s)
        if reset:
            self.engine.zero_metric_sums()
        s = s.cpu()
        return float(s[0]), float(s[1])

    def last_loss_and_grad_norm(self):
        _, _, gsq, loss = self.engine.read_scalars()
        return loss, gsq ** 0.5
