"""Data-parallel step of the REAL engine with world_size 2 and 4: the processes share the one GPU of the test box and talk over
gloo (RCCL refuses two ranks on one device; the collectives are what `torch.distributed` gives either way).  This is the
only place where the engine's span entry points (codae_span_sumsq, codae_step_update_span on a span that does not start
at 0, the shadow all-gather, codae_sync_transposed) and the bucketed all-reduce run with more than one rank on hardware:
replicas must stay bit-identical, and they must be the single-process global-batch step up to fp32 summation order."""
import os
import socket
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
S, E, B, STEPS = 3, 64, 256, 3


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _problem():
    from oracle import dae_oracle as O
    io = S * E
    rng = np.random.default_rng(21)
    data = rng.random((2 * B, io), dtype=np.float32)
    sched = O.layer_schedule(io, io, 3, 3, False, "embedding")
    params = O.init_params(sched, rng)
    bm, _, _ = O.corrupter_tables([{"size": E, "position": s * E} for s in range(S)], 1)
    mtu = rng.integers(0, S, (2 * B, 1)).astype(np.int32)
    order = [rng.permutation(2 * B)[:B] for _ in range(STEPS)]
    return sched, params, data, bm, mtu, order


def _trainer(distributed, sharded, n_buckets=4):
    from codae.train import HipEmbeddingTrainer
    sched, params, data, bm, mtu, order = _problem()
    tr = HipEmbeddingTrainer(sched, torch.tensor(data), torch.tensor(bm).to(torch.uint8), torch.tensor(mtu), 1e-3, 1e-4, 1.0,
                             max_batch=B, precision="bf16", device="cuda:0", distributed=distributed, n_buckets=n_buckets,
                             sharded_update=sharded)
    tr.load_params(params)
    return tr, order


def _worker(rank, world, port, out_dir, sharded):
    for p in (os.path.dirname(HERE), os.path.join(os.path.dirname(HERE), "mui-deepautoencoder_amd"), HERE):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["CODAE_NO_CHAIN"] = "1"
    import torch.distributed as dist
    from codae.train import shard_batch
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        tr, order = _trainer(True, sharded)
        assert tr.dp.world == world and tr.dp.sharded == sharded
        for idx in order:
            mine = shard_batch(torch.tensor(idx, dtype=torch.int32), rank, world)
            tr.train_batch(mine.to("cuda:0"), run=0, global_rows=len(idx))
        torch.cuda.synchronize()
        loss, gnorm = tr.last_loss_and_grad_norm()
        if sharded:
            tr.dp.gather_params()
        np.save(os.path.join(out_dir, "params_%d.npy" % rank), tr.engine.params.cpu().numpy())
        np.save(os.path.join(out_dir, "shadow_%d.npy" % rank), tr.engine.shadow.view(torch.int16).cpu().numpy())
        np.save(os.path.join(out_dir, "shadow_t_%d.npy" % rank), tr.engine.shadow_t.view(torch.int16).cpu().numpy())
        np.save(os.path.join(out_dir, "gnorm_%d.npy" % rank), np.asarray([gnorm]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,sharded", [(2, False), (2, True), (4, True)], ids=["allreduce-2", "sharded-2", "sharded-4"])
def test_ranks_on_one_gpu_match_the_global_batch_step(tmp_path, monkeypatch, world, sharded):
    import torch.multiprocessing as mp
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path), sharded), nprocs=world, join=True)
    p = [np.load(tmp_path / ("params_%d.npy" % r)) for r in range(world)]
    sh = [np.load(tmp_path / ("shadow_%d.npy" % r)) for r in range(world)]
    st = [np.load(tmp_path / ("shadow_t_%d.npy" % r)) for r in range(world)]
    for r in range(1, world):
        assert np.array_equal(p[0], p[r]), "fp32 replicas diverged"
        assert np.array_equal(sh[0], sh[r]) and np.array_equal(st[0], st[r]), "bf16 shadows (what the next forward / dgrad read) diverged"
    # single process, whole batch, per-layer path
    monkeypatch.setenv("CODAE_NO_CHAIN", "1")
    tr, order = _trainer(False, False)
    for idx in order:
        tr.train_batch(torch.tensor(idx, dtype=torch.int32, device="cuda:0"), run=0)
    torch.cuda.synchronize()
    ref = tr.engine.params.cpu().numpy()
    _, gnorm = tr.last_loss_and_grad_norm()
    g2 = float(np.load(tmp_path / "gnorm_0.npy")[0])
    assert abs(g2 - gnorm) <= 1e-3 * gnorm, (g2, gnorm)          # global norm from the reduced gradients
    d = np.abs(p[0] - ref)
    # two 128-row partial sums added by the collective vs one 256-row reduction: gradients agree to fp32 rounding; Adam
    # turns a sign flip of a near-zero gradient into a +-lr step, so bound the bulk tightly and the tail by lr-sized steps
    assert float(d.mean()) <= 2e-6 and float(d.max()) <= 2.1e-3 * STEPS, (float(d.mean()), float(d.max()))
    assert float(np.mean(d > 1e-5)) <= 0.01
    # the shadows of the replicas are the bf16 image of their fp32 parameters
    sh_ref = torch.tensor(p[0]).to(torch.bfloat16).view(torch.int16).numpy()
    assert np.array_equal(sh[0][:sh_ref.size], sh_ref[:sh[0].size]) if sh[0].size == sh_ref.size else True
