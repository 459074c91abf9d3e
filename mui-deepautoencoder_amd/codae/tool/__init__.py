"""codae.tool: the names the training scripts import (same public surface as the reference package, so that
`from codae.tool import Corrupter, CombinedCriterion, ...` resolves here), gathered from this build's modules.
The reference's legacy argparse table (codae/tool/parser.py) and attr-dict (dictionnary.py) are out of scope
(SURVEY.md section 2): nothing on the path uses them, so they have no counterpart here."""
from . import batching, corruption, criteria, runlog

_PUBLIC = {
    runlog: ("set_logging", "display_info", "get_date", "PlotDrawer", "export_parameters_to_json"),
    batching: ("collate_embedding", "simple_collate", "load_dataset_of_embeddings", "Normalizer",
               "get_mask_transformation"),
    corruption: ("Corrupter",),
    criteria: ("get_rmse", "RankingLoss", "CombinedCriterion"),
}
__all__ = []
for _module, _names in _PUBLIC.items():
    for _name in _names:
        globals()[_name] = getattr(_module, _name)
        __all__.append(_name)
del _module, _names, _name
