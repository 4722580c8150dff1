"""gan_ode_amd -- MI355X-native implementation of chechaohp/gan-ode's MoCoGAN + Neural-ODE hot path.

The arithmetic lives in libgode.so (hand-written HIP for gfx950, C ABI in include/gode.h); this package is the
Python host side that mirrors the reference's class surface.  Importing the package does not need a GPU; running
anything does, and there is no fallback path.
"""
from . import _lib
from .modules import (Noise, ODEFunc, PatchImageDiscriminator, VideoDiscriminator, VideoGenerator,
                      VideoGeneratorMNIST, VideoGeneratorMNISTODE, VideoGeneratorMNISTODERNN)
from .train import (FusedAdam, GanTrainer, bce_with_logits_const, build_mnist, build_ucf, freeze_host_gc,
                    host_cpu_quota, limit_host_threads, train_step)

__all__ = ["Noise", "ODEFunc", "PatchImageDiscriminator", "VideoDiscriminator", "VideoGenerator",
           "VideoGeneratorMNIST", "VideoGeneratorMNISTODE", "VideoGeneratorMNISTODERNN", "FusedAdam", "GanTrainer", "bce_with_logits_const",
           "build_mnist", "build_ucf", "train_step", "host_cpu_quota", "limit_host_threads", "freeze_host_gc", "_lib"]

import os as _os

if _os.environ.get("GODE_KEEP_TORCH_THREADS") != "1":
    # see train.limit_host_threads: an intra-op pool larger than the container's CPU quota gets the whole process
    # throttled (75-90 ms stalls); opt out with GODE_KEEP_TORCH_THREADS=1
    limit_host_threads()
