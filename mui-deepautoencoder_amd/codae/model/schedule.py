"""Widths of the Linear stack of the two DAE classes.

Counterpart of the layer-building loops of the reference constructors
(codae/model/embedding_denoising_autoencoder.py:49-126 and
codae/model/mixed_variable_denoising_autoencoder.py:45-122): one (in, out, relu)
triple per Linear; ReLU follows every Linear except the code layer and the output
layer.  Note the flag is inverted upstream: the widths taper only when
`steep_layer_size` is False.
"""
import math


def _taper(io_size, z_size, nb_layer, steep_layer_size, round_up):
    if steep_layer_size:
        return 0
    q = (io_size - z_size) / nb_layer
    return math.ceil(q) if round_up else math.floor(q)


def linear_stack(io_size, z_size, nb_input_layer, nb_output_layer, steep_layer_size, mixed):
    """mixed=False: EmbeddingDenoisingAutoencoder (floor, last Linear fed by the previous width);
    mixed=True: MixedVariableDenoisingAutoencoder (ceil, last Linear io -> io)."""
    step_in = _taper(io_size, z_size, nb_input_layer, steep_layer_size, mixed)
    step_out = _taper(io_size, z_size, nb_output_layer, steep_layer_size, mixed)

    encoder = [(io_size, io_size, True)]
    width = None
    for i in range(1, nb_input_layer):
        if steep_layer_size:
            encoder.append((io_size, io_size, True))
        else:
            width = max(io_size - i * step_in, z_size)
            encoder.append((max(io_size - (i - 1) * step_in, z_size), width, True))
    if steep_layer_size:
        encoder.append((io_size, z_size, False))
    elif width is None:
        # upstream reads an unassigned local here (embedding_...py:89)
        raise UnboundLocalError("local variable 'next_layer_output_size' referenced before assignment")
    else:
        encoder.append((width, z_size, False))

    decoder = []
    width = None
    for i in range(nb_output_layer):
        if steep_layer_size:
            decoder.append((z_size if i == 0 else io_size, io_size, True))
        else:
            width = min(z_size + (i + 1) * step_out, io_size)
            decoder.append((min(z_size + i * step_out, io_size), width, True))
    if mixed:
        decoder.append((io_size, io_size, False))
    elif width is None:
        # upstream: steep_layer_size=True never binds the name used at embedding_...py:126
        raise UnboundLocalError("local variable 'next_layer_output_size' referenced before assignment")
    else:
        decoder.append((width, io_size, False))
    return encoder, decoder
