from .dae import EmbeddingDenoisingAutoencoder, MixedVariableDenoisingAutoencoder
from .dae import cnnAutoencoder
