"""gan_ode_amd -- MI355X-native implementation of chechaohp/gan-ode's MoCoGAN + Neural-ODE hot path.

The arithmetic lives in libgode.so (hand-written HIP for gfx950, C ABI in include/gode.h); this package is the
Python host side that mirrors the reference's class surface.  Importing the package does not need a GPU; running
anything does, and there is no fallback path.
"""
from . import _lib
from .modules import (Noise, ODEFunc, PatchImageDiscriminator, VideoDiscriminator, VideoGenerator,
                      VideoGeneratorMNIST, VideoGeneratorMNISTODE, VideoGeneratorMNISTODERNN)
from .train import FusedAdam, GanTrainer, bce_with_logits_const, build_mnist, build_ucf, train_step

__all__ = ["Noise", "ODEFunc", "PatchImageDiscriminator", "VideoDiscriminator", "VideoGenerator",
           "VideoGeneratorMNIST", "VideoGeneratorMNISTODE", "VideoGeneratorMNISTODERNN", "FusedAdam", "GanTrainer", "bce_with_logits_const",
           "build_mnist", "build_ucf", "train_step", "_lib"]
