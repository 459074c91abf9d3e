#!/usr/bin/env python3
"""Train the mixed-variable denoising autoencoder on abalone-schema data (MI355X).

Same command line, config schema, split, augmentation-run loop, per-k tables and log lines as the
reference's script/train_dae_on_abalone.py (--dataset_path --output_path --config [--debug
--nb_missing]).  The model is this build's codae.model.MixedVariableDenoisingAutoencoder (11-wide
layers run on the exact-fp32 MFMA kernels), masks come from the Corrupter's HIP kernel, the
loss/optimizer are CombinedCriterion + torch.optim.Adam as upstream (scope row "next",
SURVEY.md 8f-2).
"""
import argparse
import json
import logging
import math
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from codae.hostcpu import cap_thread_env                                   # noqa: E402
cap_thread_env()        # BLAS / OpenMP pools no wider than the container's CPU quota, before numpy and torch create them

import numpy as np                                                       # noqa: E402
import pandas as pd
import torch
import yaml
from sklearn.preprocessing import MinMaxScaler


from codae.dataset import MixedVariableDataset                      # noqa: E402
from codae.hip import HipError                                      # noqa: E402
from codae.model import MixedVariableDenoisingAutoencoder           # noqa: E402
from codae.tool import CombinedCriterion, Corrupter, Normalizer, get_date, set_logging  # noqa: E402
from codae.train import SubsetEpochSampler, fit_host_threads        # noqa: E402


def parse():
    p = argparse.ArgumentParser(description='Train denoising autoencoder.')
    p.add_argument('--dataset_path', type=str, required=True)
    p.add_argument('--output_path', type=str, required=True)
    p.add_argument('--config', type=str, required=True)
    p.add_argument('--debug', type=bool, default=False)
    p.add_argument('--nb_missing', type=int, default=1)
    p.add_argument('--epochs', type=int, default=None)
    return p.parse_args()


def print_table(names, table, k_max):
    print("k ", end="")
    for name in names:
        print("%s " % str(name).rjust(12), end="")
    print()
    for i in range(k_max):
        print("%d     " % (i + 1), end="")
        for v in table[i]:
            print("%f     " % v, end="")
        print()
    print()


def main():
    print("===== Train DAE on Abalone data =====")
    args = parse()
    fit_host_threads()              # torch / BLAS pools no wider than the container's CPU quota (codae/train.py)
    log = set_logging(logging_level=(logging.DEBUG if args.debug else logging.INFO), log_file_path="log/")
    with open(args.config, 'r') as stream:
        config = yaml.safe_load(stream)
    mc = config["MODEL"]
    if not torch.cuda.is_available():
        raise HipError("no HIP device: this build has no CPU path")
    log.info("CUDA available, loading GPU device")
    device = torch.device("cuda:0")

    with open(os.path.join(args.dataset_path, "abalone.data"), 'r') as f:
        frame = pd.read_csv(f, sep=",")                        # first line becomes the header, as upstream
    scaler = MinMaxScaler()
    scaled = scaler.fit_transform(frame.iloc[:, 1:])
    for j, col in enumerate(frame.columns[1:]):
        frame[col] = scaled[:, j]
    tensor_normazer = Normalizer(normalizer=scaler, device=device)
    dataset = MixedVariableDataset(frame)

    indices = list(range(dataset.nb_observation))
    nb_train = math.floor(dataset.nb_observation * config["DATASET"]["SPLIT"][0])
    nb_validation = dataset.nb_observation - nb_train
    if config["DATASET"]["SHUFFLE"]:
        np.random.seed(config["SEED"])
        np.random.shuffle(indices)
    train_indices, validation_indices = indices[:nb_train], indices[nb_train:]
    train_sampler = SubsetEpochSampler(train_indices, mc["BATCH_SIZE"])
    validation_sampler = SubsetEpochSampler(validation_indices, mc["BATCH_SIZE"])

    corrupter = Corrupter(nb_observation=dataset.nb_observation, arch=dataset.arch, k_max=args.nb_missing, device=device)
    print(corrupter.nb_corruption_per_k)
    log.info("Initializing the model.")
    model = MixedVariableDenoisingAutoencoder(arch=dataset.arch, io_size=dataset.io_size, z_size=mc["Z_SIZE"],
                                              device=device, nb_input_layer=mc["NB_INPUT_LAYER"],
                                              nb_output_layer=mc["NB_OUTPUT_LAYER"],
                                              steep_layer_size=mc["STEEP_LAYER_SIZE"])
    print(model)
    model.to(device)
    dataset.to(device)
    optimizer = torch.optim.Adam(model.parameters(), lr=mc["LEARNING_RATE"], weight_decay=mc["WEIGHT_DECAY"])
    n_var = len(dataset.arch)
    full_criterion = CombinedCriterion(arch=dataset.arch, k_max=args.nb_missing, device=device,
                                       observation_mask=dataset.type_mask, weight=[0.4] + [1] * (n_var - 1),
                                       reduction="mean")
    monitor = CombinedCriterion(arch=dataset.arch, k_max=args.nb_missing, device=device,
                                observation_mask=dataset.type_mask, reduction="none")
    n_onehot = dataset.arch[0]["size"]
    per_k = corrupter.nb_corruption_per_k
    book = {k: [] for k in ("ftl_per_k", "ptl_per_k", "fvl_per_k", "pvl_per_k", "ftl", "ptl", "fvl", "pvl")}
    epochs = args.epochs if args.epochs is not None else mc["EPOCH"]

    def sweep(sampler, n_rows, train):
        """every observation under every C(n, <=k) corruption once (reference :200-236 / :276-314).  Nothing crosses to the host
        inside the loops: the batch is a device index vector, masks come from the Corrupter's device tables, and the per-step
        accounting of the reference (monitor criterion, get_partial, get_per_k: :227-236) is one kernel adding into fp64 tables
        that are read once per sweep."""
        for run in range(corrupter.nb_run):
            for rows in sampler.device_batches(device, dtype=torch.long):
                input_data = dataset.data[rows]
                ids = corrupter.mask_ids(rows, run)
                masks, fmask = corrupter.get_masks(rows, run)
                output_data = model(model.corrupt(input_data=input_data, mask=fmask))
                if train:
                    loss = full_criterion(x=input_data, y=output_data)
                    optimizer.zero_grad()
                    loss.backward()
                    if mc["TRUNK_GRAD"]:
                        torch.nn.utils.clip_grad_norm_(model.parameters(), 1)
                    optimizer.step()
                monitor.accumulate(input_data, output_data, ids, corrupter, normalizer=tensor_normazer, first_scaled_column=n_onehot)
        f, p, f_k, p_k = monitor.accumulated()
        for i in range(len(per_k)):
            f_k[i, :] /= n_rows * sum(per_k[:i + 1])
            p_k[i, :] /= n_rows * sum(per_k[:i + 1]) / dataset.nb_predictor
        f /= sum(per_k) * n_rows
        p /= sum(per_k) * n_rows / dataset.nb_predictor
        f_k[:, 1:] = np.sqrt(f_k[:, 1:])
        p_k[:, 1:] = np.sqrt(p_k[:, 1:])
        return np.sqrt(f), np.sqrt(p), f_k, p_k

    for epoch in range(epochs):
        log.info("===================================================== EPOCH = %d\n" % epoch)
        f, p, f_k, p_k = sweep(train_sampler, nb_train, True)
        book["ftl"].append(f); book["ptl"].append(p); book["ftl_per_k"].append(f_k); book["ptl_per_k"].append(p_k)
        log.info("TRAINING PARTIAL ERROR = %7f" % np.mean(p_k))
        print_table(dataset.variable_names, p_k, args.nb_missing)
        f, p, f_k, p_k = sweep(validation_sampler, nb_validation, False)
        book["fvl"].append(f); book["pvl"].append(p); book["fvl_per_k"].append(f_k); book["pvl_per_k"].append(p_k)
        log.info("VALIDATION PARTIAL ERROR = %7f" % np.mean(p_k))
        print_table(dataset.variable_names, p_k, args.nb_missing)

    d = os.path.join(args.output_path, get_date() + "_train_" + config["DATASET"]["NAME"])
    os.makedirs(d, exist_ok=True)
    with open(os.path.join(d, "book.json"), "w") as fjs:
        json.dump({k: np.asarray(v).tolist() for k, v in book.items()}, fjs)
    plot = config.get("PLOT", {})
    if any(plot.get(k) for k in ("TRAINING_ERROR_PER_K", "VALIDATION_ERROR_PER_K", "FULL_ERROR", "PARTIAL_ERROR")):
        import matplotlib
        matplotlib.use('agg')
        import matplotlib.pyplot as plt
        axis = np.arange(0, epochs)
        for name, a, b in (("full_error", "ftl", "fvl"), ("partial_error", "ptl", "pvl")):
            plt.plot(axis, book[a], label="Training")
            plt.plot(axis, book[b], label="Validation")
            plt.xlabel('Epoch'); plt.ylabel('RMSE'); plt.legend(loc='best')
            plt.savefig(os.path.join(d, name + ".png")); plt.clf()
    log.info("Data saved in directory %s" % d)
    return book


if __name__ == "__main__":
    main()
