"""Small helpers kept for import compatibility (codae/tool/dictionnary.py, codae/tool/parser.py)."""
import argparse


class Dict(dict):
    """dict with attribute access (codae/tool/dictionnary.py:5-10)."""

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.__dict__ = self


def parse():
    """Legacy argument parser (codae/tool/parser.py:5-27); the training scripts define their own."""
    p = argparse.ArgumentParser()
    p.add_argument('--nb_epoch', type=int, default=300)
    p.add_argument('--dataset', type=str, default="100k")
    p.add_argument('--batch_size', type=int, default=1)
    p.add_argument('--model', type=str, default="muidae")
    p.add_argument('--learning_rate', type=float, default=0.000004)
    p.add_argument('--regularization', type=float, default=0.001)
    p.add_argument('--nb_layer', type=int, default=0)
    p.add_argument('--redux', type=float, default=1.0)
    p.add_argument('--view', type=str, default='item')
    p.add_argument('--zsize', type=int, default=16)
    p.add_argument('--reload_dataset', type=bool, default=False)
    p.add_argument('--debug', action="store_true")
    p.add_argument('--normalize', action="store_true")
    p.add_argument('--max_increasing_cnt', type=int, default=2)
    p.add_argument('--max_nan_cnt', type=int, default=3)
    p.add_argument('--mode', type=int, default=0)
    return p.parse_args()
