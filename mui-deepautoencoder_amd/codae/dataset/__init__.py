from .embedding import ConcatenatedEmbeddingDataset
from .mixed import MixedVariableDataset


def df2_to_coco(*args, **kwargs):
    """DeepFashion2 -> COCO conversion is dataset preparation, outside the accelerated path
    (OUT OF SCOPE)."""
    raise NotImplementedError("df2_to_coco is not part of the MI355X training-path build")
