"""audiogan_amd -- MI355X-native G+D training path for raw-audio GANs (drop-in for the
Generator / Discriminator / dataset interface of BarclayII/audiogan).  Importing this package
loads libaudiogan_hip.so (hand-written gfx950 kernels) and fails loudly if it is missing."""
from . import _lib  # noqa: F401  (raises ImportError when the HIP library is not built)
from . import kernels, ops, losses, optim, train, convnets, dataset, timer, extras, checkpoint  # noqa: F401
from .convnets import Conv1DGenerator, Conv1DDiscriminator, ConvPoolCritic, wgan_gp_d_loss, wgan_g_loss  # noqa: F401
from .modules import Generator, GRUGenerator, Discriminator, Embedder, Residual, dense_res_bottleneck  # noqa: F401
from .losses import binary_cross_entropy_with_logits_per_sample, length_mask, masked_bce_mean  # noqa: F401

__all__ = ['Generator', 'Discriminator', 'Embedder', 'Residual', 'dense_res_bottleneck',
           'binary_cross_entropy_with_logits_per_sample', 'length_mask', 'masked_bce_mean',
           'kernels', 'ops', 'losses', 'optim', 'train']
