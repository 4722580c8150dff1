"""Drop-in for the reference's models/mocogan_ode.py (mnist_moco_ode.py:6, ucf_moco_ode.py:6)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gan_ode_amd.modules import ODEFunc, VideoGenerator, VideoGeneratorMNIST, VideoGeneratorMNISTODE  # noqa: E402,F401
