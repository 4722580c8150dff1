"""gan_ode_amd -- MI355X-native implementation of chechaohp/gan-ode's MoCoGAN + Neural-ODE hot path.

The arithmetic lives in libgode.so (hand-written HIP for gfx950, C ABI in include/gode.h); this package is the
Python host side that mirrors the reference's class surface.  Importing the package does not need a GPU; running
anything does, and there is no fallback path.
"""
from . import _lib
from .modules import (Noise, ODEFunc, PatchImageDiscriminator, VideoDiscriminator, VideoGenerator,
                      VideoGeneratorMNIST, VideoGeneratorMNISTODE, VideoGeneratorMNISTODERNN)
from .train import (FusedAdam, GanTrainer, bce_with_logits_const, bce_with_logits_halves, bce_with_logits_pair, build_mnist, build_ucf,
                    freeze_host_gc, host_cpu_quota, limit_host_threads, train_step, unit_grad)

__all__ = ["Noise", "ODEFunc", "PatchImageDiscriminator", "VideoDiscriminator", "VideoGenerator",
           "VideoGeneratorMNIST", "VideoGeneratorMNISTODE", "VideoGeneratorMNISTODERNN", "FusedAdam", "GanTrainer", "bce_with_logits_const",
           "bce_with_logits_pair", "bce_with_logits_halves", "unit_grad", "build_mnist", "build_ucf", "train_step", "host_cpu_quota", "limit_host_threads", "freeze_host_gc", "_lib"]

# Process-global host tuning is opt-in (nothing happens at import): call limit_host_threads() once at start-up when
# torch's intra-op pool is larger than the container's CPU quota (see its docstring; bench.py and tests/conftest.py
# do), and freeze_host_gc() after warm-up in long loops (GanTrainer(freeze_gc=True) does it after its 2nd iteration).
