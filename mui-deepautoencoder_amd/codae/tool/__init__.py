from .misc import Dict, parse
from .runlog import set_logging, display_info, get_date, PlotDrawer, export_parameters_to_json
from .batching import (collate_embedding, simple_collate, load_dataset_of_embeddings, Normalizer,
                       get_mask_transformation)
from .corruption import Corrupter
from .criteria import get_rmse, RankingLoss, CombinedCriterion
