#!/usr/bin/env python3
"""Train the embedding denoising autoencoder on MI355X.

Same command line, config schema, data split, log lines and outputs as the reference's
script/train_dae_on_embedding.py (--embedding_path --output_path --config [--debug --rank
--nb_missing]); the inner loop (reference :194-223 and :241-261) is one fused HIP step per
minibatch over the HBM-resident dataset (codae.train.HipEmbeddingTrainer) instead of
DataLoader + per-sample mask loop + autograd + per-step host copies.

Extra flags: --precision {bf16,f32} (default: HIP.PRECISION of the config, else bf16; widths the
bf16 tiles cannot take fall back to the exact-fp32 kernels), --epochs N (override MODEL.EPOCH).
Multi-GPU: launch with `python -m torch.distributed.run --nproc-per-node N`; each rank takes
1/N of every minibatch and gradients are all-reduced over RCCL.
"""
import argparse
import logging
import math
import os
import random
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from codae.hostcpu import cap_thread_env                                   # noqa: E402
cap_thread_env()        # BLAS / OpenMP pools no wider than the container's CPU quota, before numpy and torch create them

import numpy as np                                                       # noqa: E402
import torch
import yaml


from codae.hip import HipError                                            # noqa: E402
from codae.model.schedule import linear_stack                             # noqa: E402
from codae.tool import Corrupter, RankingLoss, display_info, get_date, load_dataset_of_embeddings, set_logging  # noqa: E402
from codae.train import HipEmbeddingTrainer, SubsetEpochSampler, fit_host_threads, shard_batch   # noqa: E402


def parse():
    p = argparse.ArgumentParser(description='Train denoising autoencoder.')
    p.add_argument('--embedding_path', type=str, required=True)
    p.add_argument('--output_path', type=str, required=True)
    p.add_argument('--config', type=str, required=True)
    p.add_argument('--debug', type=bool, default=False)
    p.add_argument('--rank', type=bool, default=False)
    p.add_argument('--nb_missing', type=int, default=1)
    p.add_argument('--precision', type=str, default=None, choices=[None, "bf16", "f32"])
    p.add_argument('--epochs', type=int, default=None)
    return p.parse_args()


def main():
    args = parse()
    fit_host_threads()              # torch / BLAS pools no wider than the container's CPU quota (codae/train.py)
    log = set_logging(logging_level=(logging.DEBUG if args.debug else logging.INFO), log_file_path="log/")
    with open(args.config, 'r') as stream:
        config = yaml.safe_load(stream)
    mc, dc = config["MODEL"], config["DATASET"]

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise HipError("no HIP device: this build has no CPU path")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        import torch.distributed as dist
        from codae.train import init_rccl_process_group, seed_all_ranks
        init_rccl_process_group(device)
        # identical sampler order, Corrupter tables and initial weights on every rank (the reference seeds none of them)
        seed_all_ranks(int(config["SEED"]))

    log.info("Loading dataset.")
    dataset = load_dataset_of_embeddings(embedding_path=args.embedding_path, config=config, cache_dir="tmp/")
    dataset_std = torch.std(dataset.data)
    log.info("Dataset STD = " + str(dataset_std))
    log.info("CUDA available, loading GPU device")

    # split exactly as the reference (:102-116)
    indices = list(range(dataset.nb_observation))
    nb_train = math.floor(dataset.nb_observation * dc["SPLIT"][0])
    nb_validation = dataset.nb_observation - nb_train
    if dc["SHUFFLE"]:
        np.random.seed(config["SEED"])
        np.random.shuffle(indices)
    train_indices, validation_indices = indices[:nb_train], indices[nb_train:]
    train_sampler = SubsetEpochSampler(train_indices, mc["BATCH_SIZE"])
    validation_sampler = SubsetEpochSampler(validation_indices, mc["BATCH_SIZE"])

    # the reference draws an unused per-observation sample here; keep Python's RNG in step (:131-132)
    c = list(range(len(dc["USED_CATEGORY"])))
    [random.sample(c, len(c)) for _ in range(dataset.nb_observation)]
    corrupter = Corrupter(nb_observation=dataset.nb_observation, arch=dataset.arch, k_max=args.nb_missing, device=device)

    log.info("Initializing the model.")
    io_size = dc["EMBEDDING_SIZE"] * len(dc["USED_CATEGORY"])
    if io_size % dc["EMBEDDING_SIZE"] != 0:
        raise Exception("Error: io_size must be a multiple of embedding_size")
    enc, dec = linear_stack(io_size, mc["Z_SIZE"], mc["NB_INPUT_LAYER"], mc["NB_OUTPUT_LAYER"], mc["STEEP_LAYER_SIZE"], False)
    precision = args.precision or config.get("HIP", {}).get("PRECISION", "bf16")
    dataset.to(device)

    def build(prec):
        return HipEmbeddingTrainer(enc + dec, dataset.data, corrupter.mask_table_u8, corrupter.mask_to_use_i32,
                                   mc["LEARNING_RATE"], mc["WEIGHT_DECAY"], clip=1.0 if mc["TRUNK_GRAD"] else 0.0,
                                   max_batch=mc["BATCH_SIZE"], precision=prec, device=device, distributed=world > 1)
    try:
        trainer = build(precision)
    except HipError as e:
        if precision != "bf16":
            raise
        log.info("bf16 tiles cannot take this stack (%s): using the exact-fp32 kernels" % e)
        trainer = build("f32")
    trainer.init_params(seed=int(torch.empty((), dtype=torch.int64).random_().item()) % (2 ** 31))

    display_info(config, dataset.nb_observation, {})
    log.info("Linear stack: " + " | ".join("%d->%d%s" % (k, n, "+ReLU" if r else "") for k, n, r in enc + dec))
    book = {k: [] for k in ("ftl", "ptl", "fvl", "pvl", "rl")}
    ranking_loss = RankingLoss(dataset, validation_indices, device=device)
    epochs = args.epochs if args.epochs is not None else mc["EPOCH"]
    S = dataset.nb_used_category

    for epoch in range(epochs):
        log.info("===================================================== EPOCH = %d" % epoch)
        for batch_indices in train_sampler.device_batches(device):      # one index copy per epoch, int32, on the device
            shard = shard_batch(batch_indices, rank, world)
            if shard is None:                   # fewer rows than ranks (ragged last batch): skipped by every rank
                nb_skipped = len(batch_indices)
                log.info("skipping a global batch of %d rows on %d ranks" % (nb_skipped, world))
                continue
            trainer.train_batch(shard.contiguous(), run=0, global_rows=len(batch_indices))
        sq, sqp = trainer.epoch_sums()
        book["ftl"].append(np.sqrt(sq / (dataset.nb_predictor * nb_train)))
        book["ptl"].append(np.sqrt(sqp / (nb_train * dataset.nb_predictor / S)))
        log.info("TRAINING FULL ERROR      = %7f" % book["ftl"][-1])
        log.info("TRAINING PARTIAL ERROR   = %7f" % book["ptl"][-1])

        # validation (reference :241-261) stays on the device: forward + metric sums in the engine, the rank metric as
        # batched GEMMs against the validation inventory with the masks taken from the Corrupter's device tables; one
        # read-back per epoch instead of a mask expansion, two .tolist() and a host sync per batch
        for idx in validation_sampler.device_batches(device):
            y = trainer.eval_batch(idx, run=0, want_y=True)
            ranking_loss.add(y, idx, corrupter, run=0)
        rl = ranking_loss.total()
        sq, sqp = trainer.epoch_sums(reduce=False)      # every rank evaluates the whole validation set
        book["fvl"].append(np.sqrt(sq / (dataset.nb_predictor * nb_validation)))
        book["pvl"].append(np.sqrt(sqp / (nb_validation * dataset.nb_predictor / S)))
        book["rl"].append(rl / nb_validation)
        log.info("VALIDATION FULL ERROR    = %7f" % book["fvl"][-1])
        log.info("VALIDATION PARTIAL ERROR = %7f" % book["pvl"][-1])
        log.info("VALIDATION RANKING ERROR = %7f" % book["rl"][-1])
    log.info("TRAINING HAS ENDED.")

    if rank == 0:
        import matplotlib
        matplotlib.use('agg')
        import matplotlib.pyplot as plt
        d = os.path.join(args.output_path, get_date() + "_train_" + dc["NAME"])
        os.makedirs(d, exist_ok=True)
        axis = np.arange(0, epochs)
        for name, a, b, extra in (("full_RMSE", "ftl", "fvl", None), ("partial_RMSE", "ptl", "pvl", float(dataset_std))):
            plt.plot(axis, book[a], label="Training")
            plt.plot(axis, book[b], label="Validation")
            if extra is not None:
                plt.plot(axis, [extra] * len(axis), label="Validation standard deviation")
            plt.xlabel('Epoch'); plt.ylabel('RMSE'); plt.legend(loc='best')
            plt.savefig(os.path.join(d, name + ".png")); plt.clf()
        if args.rank:
            plt.plot(axis, book["rl"], label="RIRE"); plt.plot(axis, [0.5] * len(axis), label="Random rank")
            plt.xlabel('Epoch'); plt.ylabel('RIRE'); plt.legend(loc='best')
            plt.savefig(os.path.join(d, "partial_RIRE.png")); plt.clf()
        with open(os.path.join(d, "book.json"), "w") as f:
            import json
            json.dump({k: [float(v) for v in vs] for k, vs in book.items()}, f)
        log.info("Data saved in directory %s" % d)
    return book


if __name__ == "__main__":
    main()
