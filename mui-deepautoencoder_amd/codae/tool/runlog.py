"""Logging helpers with the reference's names and output format (codae/tool/logger.py:15-81,
192-203).  Plot drawing is thesis tooling outside the training path; PlotDrawer only stores."""
import datetime
import json
import logging
import os
import sys


def set_logging(log_file_path="/mnt/ramdisk/", log_file_name=None, logging_level=logging.INFO):
    now = datetime.datetime.now()
    if log_file_name is None:
        log_file_name = "CODAE_" + now.strftime("%Y-%m-%d %H:%M") + ".log"
    root = logging.getLogger()
    root.setLevel(logging_level)
    sh = logging.StreamHandler(sys.stdout)
    sh.setLevel(logging_level)
    sh.setFormatter(logging.Formatter('%(asctime)s - %(levelname)s - %(message)s'))
    root.addHandler(sh)
    try:
        fh = logging.FileHandler(log_file_path + log_file_name)
        fh.setLevel(logging_level)
        root.addHandler(fh)
    except Exception:
        logging.error("Couln't create log file %s" % (log_file_path + log_file_name))
    return logging


def display_info(config, nb_observation, metric_log=None):
    m, d = config["MODEL"], config["DATASET"]
    rows = [("LEARNING RATE", "%f", m["LEARNING_RATE"], "LEARNING_RATE"),
            ("WEIGHT DECAY", "%f", m["WEIGHT_DECAY"], "WEIGHT_DECAY"),
            ("NB EPOCH", "%d", m["EPOCH"], "EPOCH"),
            ("BATCH SIZE", "%d", m["BATCH_SIZE"], "BATCH_SIZE"),
            ("NB IN LAYER", "%d", m["NB_INPUT_LAYER"], "NB_INPUT_LAYER"),
            ("NB OUT LAYER", "%d", m["NB_OUTPUT_LAYER"], "NB_OUTPUT_LAYER"),
            ("STEEP LAYER", "%d", m["STEEP_LAYER_SIZE"], "STEEP_LAYER_SIZE"),
            ("EMBEDDING SIZE", "%d", d["EMBEDDING_SIZE"], "EMBEDDING_SIZE"),
            ("Z SIZE", "%d", m["Z_SIZE"], "Z_SIZE"),
            ("IO SIZE", "%d", len(d["USED_CATEGORY"]) * d["EMBEDDING_SIZE"], "IO_SIZE"),
            ("NB CATEGORY", "%d", len(d["USED_CATEGORY"]), "NB_CATEGORY"),
            ("NB OBSERVATION", "%d", nb_observation, "NB_OBSERVATION")]
    print("")
    for i, (label, fmt, value, _) in enumerate(rows):
        logging.info(("### %-15s = " + fmt + ("\n" if i == len(rows) - 1 else "")) % (label, value))
    if metric_log is not None:
        for _, _, value, key in rows:
            metric_log[key] = value
        return metric_log


def get_date():
    d = datetime.date.today()
    t = str(datetime.datetime.now().time()).split('.')[0].replace(':', '')
    return '{:02d}{:02d}{:02d}_'.format(d.day, d.month, d.year) + t


def export_parameters_to_json(args, output_dir):
    d = dict(vars(args))
    d.pop('log', None)
    os.makedirs(output_dir, exist_ok=True)
    with open(output_dir + "/training_parameters.json", 'w+') as f:
        f.write(json.dumps(d))


class PlotDrawer:
    """Keeps curves in memory; export writes them as JSON next to where the PNG would go."""

    def __init__(self):
        self.graph_list = []

    def add(self, data, legend=None, title="", display=False):
        self.graph_list.append({"data": data, "legend": legend, "title": title})

    def export_to_png(self, data=None, legend=None, title=None, idx=None, export_path="out/"):
        import matplotlib
        matplotlib.use("agg")
        import matplotlib.pyplot as plt
        if idx is not None:
            g = self.graph_list[idx]
            data, legend, title = g["data"], g["legend"], g["title"]
        os.makedirs(export_path, exist_ok=True)
        fig = plt.figure()
        if title is not None:
            fig.suptitle(title)
        series = data if isinstance(legend, list) else [data]
        handles = [plt.plot(s)[0] for s in series]
        plt.legend(handles, legend if isinstance(legend, list) else [legend])
        fig.savefig(export_path + (title if title is not None else "figure") + ".png")
        plt.close(fig)
